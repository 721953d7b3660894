"""Network plumbing that runs without a GPU: checkpoint layout and BatchNorm folding of InferenceNet."""
import numpy as np
import torch

from librubiks.model import InferenceNet, Model, ModelConfig, make_inference_net, GenericNet


def _randomise_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.2)


def _one_hot(n, seed=0):
    rng = np.random.RandomState(seed)
    oh = np.zeros((n, 480), dtype=np.float32)
    oh[np.repeat(np.arange(n), 20), (24 * np.arange(20) + rng.randint(0, 24, (n, 20))).ravel()] = 1
    return torch.from_numpy(oh)


def test_state_dict_layout_matches_reference_names():
    torch.manual_seed(0)
    m = Model(ModelConfig())
    keys = list(m.state_dict().keys())
    # the reference's fc_small: Linear(0) ELU(1) BN(2) Linear(3) ELU(4) BN(5) | heads Linear(0) ELU BN(2) Linear(3)
    for k in ("shared_net.0.weight", "shared_net.2.running_mean", "shared_net.3.bias", "shared_net.5.num_batches_tracked",
              "policy_net.0.weight", "policy_net.2.weight", "policy_net.3.bias", "value_net.3.weight"):
        assert k in keys
    assert sum(p.numel() for p in m.parameters()) == 12_480_013
    assert m.shared_net[0].weight.shape == (4096, 480) and m.value_net[3].weight.shape == (1, 512)
    r = Model(ModelConfig(architecture="res_small"))
    assert "shared_net.resblock3.batchnorm2.running_var" in r.state_dict()
    assert ModelConfig(architecture="fc").architecture == "fc_small"   # legacy names (model.py:52-56)


def test_save_load_round_trip(tmp_path):
    torch.manual_seed(1)
    m = Model(ModelConfig(activation_function=torch.nn.ReLU(), batchnorm=True))
    _randomise_bn(m)
    m.save(str(tmp_path))
    m.save(str(tmp_path), is_min=True)
    m2 = Model.load(str(tmp_path), load_best=True)
    assert isinstance(m2.config.activation_function, torch.nn.ReLU)
    x = _one_hot(5)
    m.eval(), m2.eval()
    with torch.no_grad():
        for a, b in zip(m(x), m2(x)):
            assert torch.equal(a, b)


def test_inference_net_folding_fp32():
    """BN folding + head merging must be the same function as Model.eval() (fp32, tolerance 1e-4 relative)."""
    for act in (torch.nn.ELU(), torch.nn.ReLU()):
        for bn in (True, False):
            torch.manual_seed(2)
            m = Model(ModelConfig(activation_function=act, batchnorm=bn))
            _randomise_bn(m)
            m.eval()
            x = _one_hot(64, seed=3)
            with torch.no_grad():
                p_ref, v_ref = m(x)
            eng = InferenceNet(m, dtype=torch.float32)
            p, v = eng(x)
            assert p.shape == (64, 12) and v.shape == (64,)
            assert torch.allclose(p, p_ref, rtol=1e-4, atol=1e-4)
            assert torch.allclose(v, v_ref.reshape(-1), rtol=1e-4, atol=1e-4)
            assert eng.flops_per_state >= 2 * (480 * 4096 + 4096 * 2048 + 2 * 2048 * 512 + 512 * 13)
            assert m.training is False


def test_make_inference_net_dispatch():
    from standin_net import StandInNet
    assert isinstance(make_inference_net(StandInNet(0)), GenericNet)
    assert isinstance(make_inference_net(Model(ModelConfig()), torch.float32), InferenceNet)
    # residual architectures run on the same engines (skip connection + unfoldable BatchNorm as layer options), on the CPU too
    for arch in ("res_small", "res_big"):
        for bn in (True, False):
            m = Model(ModelConfig(architecture=arch, batchnorm=bn))
            _randomise_bn(m)
            m.eval()
            eng = make_inference_net(m, torch.float32)
            assert isinstance(eng, InferenceNet) and eng.residual
            x = _one_hot(32, seed=5)
            with torch.no_grad():
                p_ref, v_ref = m(x)
            p, v = eng(x)
            assert torch.allclose(p, p_ref, rtol=1e-4, atol=1e-4) and torch.allclose(v, v_ref.reshape(-1), rtol=1e-4, atol=1e-4)
            assert torch.allclose(eng.value(x), v_ref.reshape(-1), rtol=1e-4, atol=1e-4)


def test_reference_model_tests_restated(tmp_path):
    """tests/test_model.py:13-59 of the reference: every architecture runs in eval and train mode, every init works,
    ModelConfig survives its JSON form."""
    import json
    x = torch.randn(2, 480)
    for arch in ("fc_small", "res_big"):
        m = Model(ModelConfig(architecture=arch))
        m.eval()
        p, v = m(x)
        assert p.shape == (2, 12) and v.shape == (2, 1)
        m.train()
        p, v = m(x)
        assert p.shape == (2, 12) and v.shape == (2, 1) and torch.isfinite(p).all()
    for init in ("glorot", "he", 0, 1.123123123e-3):
        m = Model(ModelConfig(init=init))
        m(x)
        if not isinstance(init, str):
            w0 = m.shared_net[0].weight.detach()
            assert float(w0.min()) == float(w0.max()) == float(np.float32(init))
    cf = ModelConfig(torch.nn.ReLU())
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cf.as_json_dict()))
    cf2 = ModelConfig.from_json_dict(json.loads(path.read_text()))
    assert type(cf2.activation_function) is torch.nn.ReLU and cf2.architecture == cf.architecture and cf2.batchnorm == cf.batchnorm
    # value-only / policy-only calls (model.py:131-141)
    m = Model(ModelConfig()).eval()
    assert m(x, policy=True, value=False).shape == (2, 12) and m(x, policy=False, value=True).shape == (2, 1)


# ------------------------------------------------------------------------------------------------
# a11 pinned against the REFERENCE's Model (tests/golden/model_golden.npz, written by make_golden.py `model`
# from the imported reference, librubiks/model.py:106-161 and :250-264)
# ------------------------------------------------------------------------------------------------
import hashlib   # noqa: E402
import json      # noqa: E402
import os        # noqa: E402

import pytest    # noqa: E402

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = [(a, b) for a in ("fc_small", "fc_big", "res_small", "res_big") for b in (True, False)]


def golden_model_inputs():
    """One-hot of the 256 golden `oh_in` states and of the three train-mode batches (states 256..1023 of `mr_in`)."""
    g = np.load(os.path.join(_GOLDEN, "cube_golden.npz"))

    def oh(states):
        out = np.zeros((len(states), 480), dtype=np.float32)
        out[np.repeat(np.arange(len(states)), 20), (24 * np.arange(20) + states).ravel()] = 1   # cube.py:265-277
        return torch.from_numpy(out)
    assert np.array_equal(np.nonzero(oh(g["oh_in"]).numpy())[1].reshape(256, 20), g["oh_cols"])
    return oh(g["oh_in"]), [oh(g["mr_in"][256 * i:256 * (i + 1)]) for i in (1, 2, 3)]


def seeded_reference_model(arch, bn, fx=None, trained_stats=False):
    """The build's Model under the reference's seed; with `trained_stats` the BatchNorm statistics of the fixture loaded."""
    torch.manual_seed(0)
    net = Model.create(ModelConfig(architecture=arch, batchnorm=bn))
    if trained_stats:
        pre = f"{arch}_bn1_stat_"
        net.load_state_dict({k[len(pre):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(pre)}, strict=False)
    return net


@pytest.mark.parametrize("arch,bn", MODEL_CASES)
def test_model_is_the_references_module(arch, bn):
    """Same ordered state_dict (key, shape, dtype), same parameters bit for bit under the same seed, same module tree, and
    the same function: eval-mode outputs in float64 to 1e-10 and in fp32 to 2e-6 x |out| (fp32 summation order differs
    between BLAS builds and thread counts, nothing else may), fresh and behind three train-mode forwards."""
    fx = np.load(os.path.join(_GOLDEN, "model_golden.npz"))
    meta = json.loads(str(fx["meta_json"]))[f"{arch}_bn{int(bn)}"]
    assert str(fx["torch_version"]) == torch.__version__      # initialisers are only promised per torch build
    net = seeded_reference_model(arch, bn)
    assert net.training
    sd = net.state_dict()
    assert [[k, list(t.shape), str(t.dtype)] for k, t in sd.items()] == meta["keys"]
    assert sum(p.numel() for p in net.parameters()) == meta["n_params"]
    for k, t in sd.items():
        assert hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest() == meta["sha256"][k], k
    # module tree (layer order Linear -> ELU -> BatchNorm1d, model.py:150-159); the reference's residual class is `ResNet`
    assert repr(net).split("\n", 1)[1] == meta["repr"].split("\n", 1)[1]
    oh_eval, oh_train = golden_model_inputs()
    name = f"{arch}_bn{int(bn)}"

    def check(net, tag):
        net.eval()
        with torch.no_grad():
            p, v = net(oh_eval)
            p64, v64 = net.double()(oh_eval.double())
            net.float()
        assert p.dtype == torch.float32 and v.shape == (256, 1)
        for got, want in ((p64, fx[f"{tag}_p64"]), (v64, fx[f"{tag}_v64"])):
            assert np.abs(got.numpy() - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
        for got, want in ((p, fx[f"{tag}_p32"]), (v, fx[f"{tag}_v32"])):
            assert np.abs(got.numpy() - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
    check(net, f"{name}_fresh")
    if bn:
        net.train()
        with torch.no_grad():
            for x in oh_train:
                net(x)
        for k, t in net.state_dict().items():
            if "running_" in k or "num_batches" in k:
                assert np.allclose(t.numpy(), fx[f"{name}_stat_{k}"], rtol=1e-5, atol=1e-7), k
        check(seeded_reference_model(arch, bn, fx, trained_stats=True), f"{name}_trained_stats")
