"""
GPU tests of the network path: the fused input-layer kernel against the one-hot GEMM it replaces, and
the bf16 inference engine against the fp32 module.  Floating point: tolerances stated per test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cube as oc  # noqa: E402  (checker only)


def _states(n, seed):
    rng = np.random.RandomState(seed)
    s = np.tile(oc.get_solved(), (n, 1))
    for _ in range(25):
        s = oc.multi_rotate_actions(s, rng.randint(0, 12, n))
    return s


def _model(act=None, seed=0):
    from librubiks.model import Model, ModelConfig
    torch.manual_seed(seed)
    m = Model.create(ModelConfig(activation_function=act or torch.nn.ELU())).eval()
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_((torch.randn(mod.num_features, generator=g) * 0.3).cuda())
            mod.running_var.copy_((torch.rand(mod.num_features, generator=g) + 0.5).cuda())
    return m


@pytest.mark.parametrize("n", [1, 17, 512, 513, 4099, 12288])
@pytest.mark.parametrize("act", ["elu", "relu"])
@pytest.mark.parametrize("table", ["bf16", "f16", "f16pair", "mfma", "mfma16"])
def test_first_layer_matches_onehot_gemm(n, act, table):
    """
    rc_first_layer_bf16 == act(as_oh(s) @ W1^T + b1) computed in fp32 from the SAME bf16 weights,
    then rounded to bf16.  The sums have 20 terms in a different order: tolerance = 1 bf16 ulp
    (rtol 2^-7) plus 1e-3 absolute.
    """
    from librubiks.cube import DeviceCubes
    from librubiks.model import InferenceNet
    m = _model(torch.nn.ELU() if act == "elu" else torch.nn.ReLU())
    eng = InferenceNet(m, dtype=torch.bfloat16, first_layer_table=table)
    assert eng.supports_cubes and eng._fused_first[5] == {"bf16": 0, "f16": 1, "mfma": 2, "f16pair": 3, "mfma16": 4}[table]
    s = _states(n, seed=n)
    cubes = DeviceCubes.from_numpy(s)
    got = eng.first_layer(cubes).float()
    oh = torch.from_numpy(oc.as_oh(s)).cuda()
    # reference: the SAME 16-bit table and fp32 bias, accumulated in fp32 by a dense product
    table_kc = eng._fused_first[0].float() if not table.startswith("mfma") else eng._fused_first[0].float().t()   # [480][H]
    ref = oh @ table_kc + eng._fused_first[1]
    ref = torch.nn.functional.elu(ref) if act == "elu" else torch.relu(ref)
    ref = ref.to(torch.bfloat16).float()
    assert got.shape == (n, 4096)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=1e-3), float((got - ref).abs().max())
    # exactly equal on the overwhelming majority of elements ("f16pair" rounds ten pair sums to half precision
    # before the fp32 accumulation: a few more results land on the neighbouring bf16 value)
    assert float((got == ref).float().mean()) > (0.85 if table == "f16pair" else 0.98)


def test_engine_paths_agree_and_track_fp32():
    """bf16 engine: fused-input path vs one-hot path (atol 2e-2), and both vs the fp32 module (atol 3e-2)."""
    from librubiks.cube import DeviceCubes
    from librubiks.model import InferenceNet
    m = _model()
    eng = InferenceNet(m, dtype=torch.bfloat16)
    s = _states(3000, seed=3)
    cubes = DeviceCubes.from_numpy(s)
    p1, v1 = eng.forward_cubes(cubes)
    p2, v2 = eng(cubes.as_oh(torch.bfloat16))
    with torch.no_grad():
        p0, v0 = m(cubes.as_oh(torch.float32))
    assert torch.allclose(p1, p2, atol=2e-2) and torch.allclose(v1, v2, atol=2e-2)
    assert torch.allclose(p1, p0, atol=3e-2) and torch.allclose(v1, v0.reshape(-1), atol=3e-2)
    assert torch.allclose(eng.value_cubes(cubes), v1, atol=2e-2)
    assert torch.allclose(eng.value(cubes.as_oh(torch.bfloat16)), v2, atol=2e-2)
    eng32 = InferenceNet(m, dtype=torch.float32)
    assert not eng32.supports_cubes
    p3, v3 = eng32(cubes.as_oh(torch.float32))
    assert torch.allclose(p3, p0, rtol=1e-3, atol=1e-3) and torch.allclose(v3, v0.reshape(-1), rtol=1e-3, atol=1e-3)


def test_first_layer_argument_errors():
    from librubiks import _hip
    lib = _hip.lib()
    soa = torch.zeros(20 * 256, dtype=torch.int8, device="cuda")
    w = torch.zeros(480 * 128, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(128, device="cuda")
    out = torch.zeros(256 * 128, dtype=torch.bfloat16, device="cuda")
    args = (soa.data_ptr(), 100, 256, w.data_ptr(), b.data_ptr(), out.data_ptr())
    assert lib.rc_first_layer_bf16(*args, 100, 2, 1.0, 0, None) == -4    # H not a multiple of 128
    assert lib.rc_first_layer_bf16(*args, 128, 7, 1.0, 0, None) == -4    # unknown activation
    assert lib.rc_first_layer_bf16(*args, 128, 2, 1.0, 0, None) == 0
    assert lib.rc_first_layer_bf16(*args, 128, 1, 1.0, 1, None) == 0


@pytest.mark.parametrize("n", [1, 5, 1000, 11264])
def test_fused_head_matches_gemm_path(n):
    """
    rc_head_bf16 (last ELU + 1024 -> 13 layer in fp32 from the raw bf16 activations) vs the ELU pass + head GEMM it
    replaces.  The GEMM path rounds the activations to bf16 before the last layer, the kernel does not: the two
    agree within 3e-2 absolute on O(1) logits / values, and the kernel is the one closer to an fp32 evaluation.
    """
    from librubiks.cube import DeviceCubes
    from librubiks.model import InferenceNet
    m = _model()
    eng = InferenceNet(m, dtype=torch.bfloat16)
    assert eng._fused_head_ok()
    cubes = DeviceCubes.from_numpy(_states(n, seed=n + 3))
    head = eng.head_cubes(cubes)
    assert head.shape == (n, 16) and head.dtype == torch.float32
    ref = eng._run(eng.layers[1:], eng.first_layer(cubes)).float()
    assert torch.allclose(head[:, :13], ref, atol=3e-2)
    assert (head[:, 13:] == 0).all()
    # fp32 evaluation of the same last two layers from the same bf16 input of the last hidden layer
    x = eng._run(eng.layers[1:-2], eng.first_layer(cubes)).float()
    W3, b3, _ = eng.layers[-2]
    W4, b4, _ = eng.layers[-1]
    raw = (x @ W3.float().t() + b3.float()).to(torch.bfloat16).float()
    exact = torch.nn.functional.elu(raw) @ W4.float().t() + b4.float()
    assert float((head[:, :13] - exact).abs().max()) <= float((ref - exact).abs().max()) + 1e-3


@pytest.mark.parametrize("n", [8, 1000 * 8, 11264 * 2048])
def test_inplace_activation_kernel(n):
    """rc_act_bf16_inplace == torch's ELU / ReLU evaluated in fp32 and rounded to bf16 (1 bf16 ulp: exp differs)."""
    from librubiks import _hip
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.randn(n, device="cuda", generator=g) * 3).to(torch.bfloat16)
    for code, ref in ((2, lambda t: torch.nn.functional.elu(t.float(), alpha=1.0)), (1, lambda t: torch.relu(t.float()))):
        y = x.clone()
        _hip.check(_hip.lib().rc_act_bf16_inplace(y.data_ptr(), n, code, 1.0, _hip.stream_ptr()), "rc_act_bf16_inplace")
        want = ref(x).to(torch.bfloat16)
        assert torch.allclose(y.float(), want.float(), rtol=2 ** -7, atol=1e-6)
        assert float((y == want).float().mean()) > 0.99
    assert _hip.lib().rc_act_bf16_inplace(x.data_ptr(), 12, 2, 1.0, None) == -2      # n % 8 != 0
    assert _hip.lib().rc_act_bf16_inplace(x.data_ptr(), 8, 7, 1.0, None) == -4       # unknown activation
