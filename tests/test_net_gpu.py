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
@pytest.mark.parametrize("weights", ["half", "bf16"])
def test_first_layer_matches_onehot_gemm(n, act, weights):
    """
    rc_first_layer_mfma_bf16 == act(as_oh(s) @ W1^T + b1) computed in fp32 from the SAME 16-bit weights,
    then rounded to bf16.  The sums have 20 terms in a different order: tolerance = 1 bf16 ulp
    (rtol 2^-7) plus 1e-3 absolute.  "half": the default (IEEE-half weights); "bf16": what the engine falls back to when
    a folded input-layer weight lies outside half's range -- one weight (output 7, cubie 0 / code 0) is pushed to 5e4 here.
    """
    from librubiks.cube import DeviceCubes
    from librubiks.model import InferenceNet
    m = _model(torch.nn.ELU() if act == "elu" else torch.nn.ReLU())
    if weights == "bf16":
        with torch.no_grad():
            m.shared_net[0].weight[7, 0] = 5.0e4      # beyond 3e4: the engine keeps the input layer's weights in bf16
    eng = InferenceNet(m, dtype=torch.bfloat16)
    assert eng.supports_cubes and eng._fused_first[5] == (weights == "half")
    assert eng._fused_first[0].dtype == (torch.float16 if weights == "half" else torch.bfloat16)
    s = _states(n, seed=n)
    cubes = DeviceCubes.from_numpy(s)
    got = eng.first_layer(cubes).float()
    oh = torch.from_numpy(oc.as_oh(s)).cuda()
    # reference: the SAME 16-bit weights and fp32 bias, accumulated in fp32 by a dense product
    ref = oh @ eng._fused_first[0].float().t() + eng._fused_first[1]
    ref = torch.nn.functional.elu(ref) if act == "elu" else torch.relu(ref)
    ref = ref.to(torch.bfloat16).float()
    assert got.shape == (n, 4096)
    assert torch.allclose(got, ref, rtol=2 ** -7, atol=1e-3), float((got - ref).abs().max())
    assert float((got == ref).float().mean()) > 0.98      # exactly equal on the overwhelming majority of elements


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


@pytest.mark.parametrize("n", [1, 5, 352, 1000, 11267, 120003])
def test_input_layer_as_a_sum_of_rows(n):
    """
    rc_first_layer_gather_f16: y = act(bias + sum_j W^T[24 j + code_j]) in fp32, in the order j = 0 .. 19 behind the bias, written as
    [hi | lo] halves.  Without activation the kernel equals the same twenty fp32 additions in torch BIT FOR BIT (and so does its split
    into halves); with ELU it stays within fp32 rounding of float64; the one-hot MFMA kernel it replaces in the split engine
    (rc_first_layer_split_flag_f16, hi / lo tables of the same weights) agrees to the split format's 2^-22; rows past n are not written;
    the half-range flag rises with an output beyond 65504; malformed requests are refused.  The kernel has two forms by batch size (four
    columns per lane on sixteen waves / eight on eight: the last n of the list is in the second) -- the same additions in the same order.
    """
    from librubiks import _hip
    from librubiks.cube import DeviceCubes
    lib = _hip.lib()
    g = torch.Generator().manual_seed(100 + n)
    H = 192
    Wt = (torch.randn(480, H, generator=g) * 0.4).cuda().contiguous()        # W^T: one row per (cubie, code)
    b = torch.randn(H, generator=g).cuda()
    s = _states(n, seed=7 + n)
    cubes = DeviceCubes.from_numpy(s)
    idx = torch.from_numpy(s.astype(np.int64)).cuda() + 24 * torch.arange(20, device="cuda")      # [n, 20] rows of W^T
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = torch.full((n + 3, 2 * H), 7.0, dtype=torch.float16, device="cuda")

    def run(act, bias=b):
        _hip.check(lib.rc_first_layer_gather_f16(cubes.soa.data_ptr(), n, cubes.stride, Wt.data_ptr(), bias.data_ptr(), out.data_ptr(), H, act, 1.0,
                                                 flag.data_ptr(), None), "rc_first_layer_gather_f16")
        return out[:n, :H].clone(), out[:n, H:].clone()
    hi, lo = run(0)
    acc = b.expand(n, H).clone()
    for j in range(20):
        acc = acc + Wt[idx[:, j]]                                            # the same fp32 additions, in the same order
    ref_hi = acc.half()
    ref_lo = ((acc - ref_hi.float()) * 2048.0).half()
    assert torch.equal(hi, ref_hi) and torch.equal(lo, ref_lo)
    assert bool((out[n:] == 7.0).all()) and int(flag.item()) == 0          # nothing behind row n
    y64 = b.double() + Wt.double()[idx].sum(1)
    hi, lo = run(2)
    got = hi.double() + lo.double() / 2048
    want = torch.where(y64 > 0, y64, torch.expm1(y64))
    assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))
    # the MFMA form of the same layer: hi / lo tables of the same weights
    W = Wt.t().contiguous()
    Wh = W.half()
    Wl = ((W - Wh.float()) * 2048.0).half()
    out2 = torch.empty((n, 2 * H), dtype=torch.float16, device="cuda")
    _hip.check(lib.rc_first_layer_split_flag_f16(cubes.soa.data_ptr(), n, cubes.stride, Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), out2.data_ptr(), H, 2, 1.0,
                                                 flag.data_ptr(), None), "rc_first_layer_split_flag_f16")
    assert float((out2[:, :H].double() + out2[:, H:].double() / 2048 - got).abs().max()) < 4e-6 * max(1.0, float(want.abs().max()))
    assert int(flag.item()) == 0
    big = b.clone()
    big[H - 1] = 1.0e5                                                       # column H - 1 of every state leaves half's range
    run(2, big)
    assert int(flag.item()) == 1
    args = (cubes.soa.data_ptr(), n, cubes.stride, Wt.data_ptr(), b.data_ptr(), out.data_ptr())
    assert lib.rc_first_layer_gather_f16(*args, 100, 2, 1.0, None, None) == -4          # H not a multiple of 64
    assert lib.rc_first_layer_gather_f16(*args, H, 7, 1.0, None, None) == -4            # unknown activation
    assert lib.rc_first_layer_gather_f16(args[0], n, args[2], None, *args[4:], H, 2, 1.0, None, None) == -1   # no table
    assert lib.rc_first_layer_gather_f16(*args, H, 1, 1.0, None, None) == 0             # ReLU, no flag


def test_first_layer_argument_errors():
    from librubiks import _hip
    lib = _hip.lib()
    soa = torch.zeros(20 * 256, dtype=torch.int8, device="cuda")
    w = torch.zeros(480 * 128, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(128, device="cuda")
    out = torch.zeros(256 * 128, dtype=torch.bfloat16, device="cuda")
    args = (soa.data_ptr(), 100, 256, w.data_ptr(), b.data_ptr(), out.data_ptr())
    assert lib.rc_first_layer_mfma_bf16(*args, 100, 2, 1.0, 0, None) == -4    # H not a multiple of 128
    assert lib.rc_first_layer_mfma_bf16(*args, 128, 7, 1.0, 0, None) == -4    # unknown activation
    assert lib.rc_first_layer_mfma_bf16(*args, 128, 2, 1.0, 0, None) == 0
    assert lib.rc_first_layer_mfma_bf16(*args, 128, 1, 1.0, 1, None) == 0


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


def test_split_f32_engine_is_at_least_as_accurate_as_fp32():
    """
    SplitF32Net (three f16 MFMA products per layer, fp32 accumulation) against the float64 forward of the module,
    next to the fp32 forward the reference runs (librubiks/model.py:131-141): the split engine must not be further
    from float64 than fp32 itself is.  Tolerance stated: its max |error| <= 1.25 x that of the fp32 module, and
    < 2e-5 absolute on outputs of magnitude ~10.
    """
    import copy
    import os
    from conftest import ROOT
    from librubiks import cube
    from librubiks.model import F32_SPLIT, InferenceNet, Model, ModelConfig, SplitF32Net, make_inference_net
    torch.manual_seed(0)
    np.random.seed(0)
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    nets = [Model.create(ModelConfig()).eval()]
    if os.path.isdir(wdir):
        nets.append(Model.load(wdir).eval())
    cubes, _, _ = cube.scramble_batch(4096, 30, True)
    oh = cubes.as_oh(torch.float32)
    for net in nets:
        eng = make_inference_net(net, F32_SPLIT)
        assert isinstance(eng, SplitF32Net) and eng.supports_cubes
        ref64 = copy.deepcopy(net).double()
        with torch.no_grad():
            p64, v64 = ref64(oh.double())
            p32, v32 = net(oh)
        ps, vs = eng.forward_cubes(cubes)                 # input layer: the fused MFMA kernel from the cube states
        pe, ve = eng(oh)                                  # ... and as library GEMM on the split one-hot: same numbers up to summation order
        assert float((ps - pe).abs().max()) <= 4e-6 * max(1.0, float(pe.abs().max())) and float((vs - ve).abs().max()) <= 4e-6 * max(1.0, float(ve.abs().max()))
        eng.fused_input = False
        pg, vg = eng.forward_cubes(cubes)                 # the GEMM form from cubes (rc_oh_split_f16) equals the one-hot entry point exactly
        eng.fused_input = True
        assert torch.equal(pg, pe) and torch.equal(vg, ve)
        pi, vi = InferenceNet(net, torch.float32)(oh)     # BN-folded fp32 engine, for reference
        err = lambda a, b: float((a.double() - b).abs().max())   # noqa: E731
        e_split = max(err(ps, p64), err(vs, v64.reshape(-1)))
        e_f32 = max(err(p32, p64), err(v32.reshape(-1), v64.reshape(-1)))
        e_fold = max(err(pi, p64), err(vi, v64.reshape(-1)))
        scale = float(p64.abs().max())
        print(f"|out| <= {scale:.2f}: max |err| vs float64: split {e_split:.3e}, fp32 module {e_f32:.3e}, fp32 folded engine {e_fold:.3e}")
        assert e_split <= 1.25 * e_f32 + 1e-7 and e_split < 2e-5 * max(1.0, scale / 10)
        # value head alone (A*'s cost) and a column window of the SoA
        assert err(eng.value_cubes(cubes), v64.reshape(-1)) <= 1.25 * e_f32 + 1e-7
        # (a 512-row window runs on the library GEMMs, 4 096 rows on the own kernel with its K loop cut in two: same rows, results
        #  within fp32 rounding of each other -- the layer plan depends on the row count)
        win, full = eng.value_cubes(cubes, None, 1024, 512), eng.value_cubes(cubes)[1024:1536]
        assert float((win - full).abs().max()) <= 2e-6 * max(1.0, float(full.abs().max()))
        eng.fused_hidden = False
        assert torch.equal(eng.value_cubes(cubes, None, 1024, 512), eng.value_cubes(cubes)[1024:1536])   # one plan: bit for bit
        eng.fused_hidden = True


def test_split_f32_engine_drives_the_search():
    """MCTS and A* on the split engine: same solve decisions as the fp32 engine on easy scrambles, trees exact vs oracle replay."""
    import os
    from conftest import ROOT
    from librubiks.model import F32_SPLIT, Model, ModelConfig
    from librubiks.solving.agents import AStar, MCTS
    from oracle import cube as oc
    torch.manual_seed(0)
    np.random.seed(1)
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
    states = np.array([oc.scramble(6 + i % 6, True)[0] for i in range(48)])
    for use_graph in (False, True):
        agent = MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT, use_graph=use_graph)
        res = agent.search_batch(states, None, 6000, compact=False)
        assert agent.forest._fused and agent.forest.rows_per_tree == 11
        for t in range(48):
            if res.solved[t]:
                x = states[t]
                for a in res.queues[t]:
                    x = oc.rotate(x, *oc.ACTION_SPACE[a])
                assert oc.is_solved(x)
        if os.path.isdir(wdir):
            assert res.solved.mean() > 0.8
    ra = AStar(net, lambda_=0.2, expansions=20, net_dtype=F32_SPLIT).search_batch(states, None, 4000)
    rb = AStar(net, lambda_=0.2, expansions=20, net_dtype=torch.float32).search_batch(states, None, 4000)
    assert np.array_equal(ra.solved, rb.solved)
    if os.path.isdir(wdir):
        assert ra.solved.mean() > 0.9 and abs(ra.lengths[ra.solved].mean() - rb.lengths[rb.solved].mean()) < 0.5


# ---- own MFMA layer kernels (csrc/rubiks_gemm.hip) ------------------------------------------------------------------
def _split_halves(x64):
    hi = x64.half()
    return hi, ((x64 - hi.double()) * 2048.0).half()


@pytest.mark.parametrize("rows,k,n_out", [(11264, 512, 256), (1000, 256, 512), (353, 128, 256), (7, 64, 128)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_split_layer_kernel_matches_the_library_chain(rows, k, n_out, act):
    """rc_split_gemm_f16 (three f16 products in one accumulator + bias + activation + re-split) against the chain it replaces
    (two fp32-out library GEMMs + rc_split_act_f16) and against float64: |error| within fp32 rounding of the result's scale.
    Every tile shape walks K in the same order, so their outputs are bit-identical (a row's result does not depend on the tile
    or on how many rows share the launch)."""
    from librubiks import _hip
    lib = _hip.lib()
    g = torch.Generator().manual_seed(rows + k + act)
    x = (torch.randn(rows, k, generator=g, dtype=torch.float64) * 0.7).float().double()
    W = (torch.randn(n_out, k, generator=g, dtype=torch.float64) / np.sqrt(k)).float().double()
    b = torch.randn(n_out, generator=g, dtype=torch.float64).float().cuda()
    (xh, xl), (wh, wl) = _split_halves(x), _split_halves(W)
    a = torch.cat([xh, xl], 1).contiguous().cuda()
    W3 = torch.cat([wl, wh, wh], 1).contiguous().cuda()
    Wh, B2 = wh.contiguous().cuda(), torch.cat([wl, wh], 1).contiguous().cuda()
    y = (xh.double() + xl.double() / 2048.0) @ (wh.double() + wl.double() / 2048.0).t() + b.cpu().double()
    ref = torch.where(y > 0, y, torch.expm1(y)) if act == 2 else torch.relu(y) if act == 1 else y
    scale = max(1.0, float(ref.abs().max()))
    tiles = [t for t in (1, 2, 3) if t in (2, 3) or n_out % 256 == 0]
    for split_out in (True, False):
        c = torch.mm(a[:, :k], Wh.t(), out_dtype=torch.float32)
        corr = torch.mm(a, B2.t(), out_dtype=torch.float32)
        chain = torch.empty((rows, 2 * n_out), dtype=torch.float16, device="cuda") if split_out else torch.empty((rows, n_out), device="cuda")
        _hip.check(lib.rc_split_act_f16(c.data_ptr(), corr.data_ptr(), 1.0 / 2048.0, rows, n_out, b.data_ptr(), act, 1.0,
                                        chain.data_ptr() if split_out else None, None if split_out else chain.data_ptr(), None), "rc_split_act_f16")
        outs = []
        for tile in tiles:
            o = torch.full((rows + 1, 2 * n_out), float("nan"), dtype=torch.float16, device="cuda") if split_out else \
                torch.full((rows + 1, n_out), float("nan"), device="cuda")
            _hip.check(lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), b.data_ptr(), rows, n_out, k, act, 1.0,
                                             o.data_ptr() if split_out else None, None if split_out else o.data_ptr(), tile, None), "rc_split_gemm_f16")
            torch.cuda.synchronize()
            assert bool(torch.isnan(o[rows]).all()), "wrote past the last row"
            outs.append(o[:rows])
        for o in outs[1:]:
            assert torch.equal(o, outs[0])

        def value(t):
            t = t.cpu()
            return t[:, :n_out].double() + t[:, n_out:].double() / 2048.0 if split_out else t.double()
        assert float((value(outs[0]) - ref).abs().max()) <= 4e-6 * scale       # fp32-level accuracy against float64
        assert float((value(outs[0]) - value(chain)).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize("rows,k,n_out,act", [(11264, 512, 256, 2), (1000, 256, 512, 1), (353, 128, 256, 0), (7, 64, 128, 2)])
def test_bf16_layer_kernel_matches_the_library_chain(rows, k, n_out, act):
    """rc_gemm_bias_act_bf16 against torch.addmm + rc_act_bf16_inplace: both round an fp32 accumulation to bf16 (the library
    rounds once more before the activation), so they agree within one bf16 step, rtol 2^-7; tiles agree bit for bit."""
    from librubiks import _hip
    lib = _hip.lib()
    g = torch.Generator().manual_seed(rows + k)
    x = (torch.randn(rows, k, generator=g) * 0.7).bfloat16().cuda()
    W = (torch.randn(n_out, k, generator=g) / np.sqrt(k)).bfloat16().cuda()
    b = torch.randn(n_out, generator=g).cuda()
    chain = torch.addmm(b.bfloat16(), x, W.t())
    if act:
        _hip.check(lib.rc_act_bf16_inplace(chain.data_ptr(), chain.numel(), act, 1.0, None), "rc_act_bf16_inplace")
    outs = []
    for tile in ([1, 3] if n_out % 256 == 0 else [3]):
        o = torch.full((rows + 1, n_out), float("nan"), dtype=torch.bfloat16, device="cuda")
        _hip.check(lib.rc_gemm_bias_act_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), rows, n_out, k, act, 1.0, o.data_ptr(), tile, None),
                   "rc_gemm_bias_act_bf16")
        torch.cuda.synchronize()
        assert bool(torch.isnan(o[rows]).all())
        outs.append(o[:rows])
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    y = x.double() @ W.double().t() + b.double()
    ref = torch.where(y > 0, y, torch.expm1(y)) if act == 2 else torch.relu(y) if act == 1 else y
    assert torch.allclose(outs[0].double(), ref, rtol=2.0 ** -7, atol=2.0 ** -7)
    assert torch.allclose(outs[0].double(), chain.double(), rtol=2.0 ** -6, atol=2.0 ** -6)


def test_layer_kernel_argument_errors():
    from librubiks import _hip
    lib = _hip.lib()
    a = torch.zeros((16, 256), dtype=torch.float16, device="cuda")
    w = torch.zeros((128, 384), dtype=torch.float16, device="cuda")
    b = torch.zeros(128, device="cuda")
    o = torch.zeros((16, 256), dtype=torch.float16, device="cuda")
    f = torch.zeros((16, 128), device="cuda")
    ok = lambda **kw: lib.rc_split_gemm_f16(kw.get("a", a.data_ptr()), w.data_ptr(), b.data_ptr(), kw.get("rows", 16), kw.get("n", 128),   # noqa: E731
                                            kw.get("k", 128), kw.get("act", 2), 1.0, kw.get("o", o.data_ptr()), kw.get("f", None), kw.get("tile", 0), None)
    assert ok() == 0 and ok(rows=0) == 0
    assert ok(a=None) == -1 and ok(o=None) == -1 and ok(f=f.data_ptr()) == -1   # exactly one output
    assert ok(k=96) == -4 and ok(n=192) == -4 and ok(act=3) == -4 and ok(tile=4) == -4 and ok(tile=5) == -4
    assert ok(tile=1) == -4                                  # 352 x 256 tiles need n_out % 256 == 0
    assert ok(a=a.data_ptr() + 2) == -2
    bf = lambda **kw: lib.rc_gemm_bias_act_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), 16, kw.get("n", 128), kw.get("k", 128), 2, 1.0,   # noqa: E731
                                                kw.get("o", o.data_ptr()), kw.get("tile", 0), None)
    assert bf() == 0 and bf(o=None) == -1 and bf(k=32) == -4 and bf(tile=2) == -4
    assert bf(tile=1) == -4


@pytest.mark.parametrize("n,K,n_out,act", [(11264, 1024, 13, 2), (1000, 512, 1, 2), (17, 1024, 16, 1), (1, 512, 13, 0)])
def test_split_head_kernel_matches_float64(n, K, n_out, act):
    """rc_head_split_f32: act(c + 2^-11 corr + b) followed by the skinny fp32 output layer, against float64 (fp32 FMA chains:
    error within a few fp32 roundings of sum |w y|) and against the two-step path it replaces."""
    from librubiks import _hip
    g = torch.Generator().manual_seed(n + K)
    c = torch.randn(n, K, generator=g).cuda()
    corr = (torch.randn(n, K, generator=g) * 100).cuda()
    bh = torch.randn(K, generator=g).cuda()
    w = (torch.randn(n_out, K, generator=g) / np.sqrt(K)).cuda()
    bo = torch.randn(n_out, generator=g).cuda()
    out = torch.full((n + 1, 16), float("nan"), device="cuda")
    _hip.check(_hip.lib().rc_head_split_f32(c.data_ptr(), corr.data_ptr(), 1.0 / 2048.0, n, K, bh.data_ptr(), act, 1.0, w.data_ptr(),
                                            bo.data_ptr(), n_out, out.data_ptr(), None), "rc_head_split_f32")
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[n]).all()) and bool((out[:n, n_out:] == 0).all())
    y = c.double() + corr.double() / 2048.0 + bh.double()
    y = torch.where(y > 0, y, torch.expm1(y)) if act == 2 else torch.relu(y) if act == 1 else y
    ref = y @ w.double().t() + bo.double()
    bound = 8 * 2.0 ** -24 * ((y.abs() @ w.double().abs().t()) + bo.double().abs()) + 1e-6
    assert bool(((out[:n, :n_out].double() - ref).abs() <= bound).all())
    assert _hip.lib().rc_head_split_f32(c.data_ptr(), None, 0.0, n, 768, bh.data_ptr(), act, 1.0, w.data_ptr(), bo.data_ptr(), n_out,
                                        out.data_ptr(), None) == -4     # K must be 512 or 1024
    assert _hip.lib().rc_head_split_f32(None, None, 0.0, n, K, bh.data_ptr(), act, 1.0, w.data_ptr(), bo.data_ptr(), n_out,
                                        out.data_ptr(), None) == -1


@pytest.mark.parametrize("rows,k,n_out", [(11264, 256, 256), (1000, 128, 512), (353, 384, 256)])
def test_split_layer_partials_sum_to_the_whole_layer(rows, k, n_out):
    """rc_split_gemm_partials_f16 (the K loop cut in two): partials[1] + 2^-11 partials[0] equals the pre-activation the one-kernel
    form accumulates, up to the fp32 rounding of one more addition; against float64 within fp32 accuracy."""
    from librubiks import _hip
    lib = _hip.lib()
    g = torch.Generator().manual_seed(rows + k)
    x = (torch.randn(rows, k, generator=g, dtype=torch.float64) * 0.7).float().double()
    W = (torch.randn(n_out, k, generator=g, dtype=torch.float64) / np.sqrt(k)).float().double()
    (xh, xl), (wh, wl) = _split_halves(x), _split_halves(W)
    a = torch.cat([xh, xl], 1).contiguous().cuda()
    W3 = torch.cat([wl, wh, wh], 1).contiguous().cuda()
    part = torch.full((2 * rows + 1, n_out), float("nan"), device="cuda")
    _hip.check(lib.rc_split_gemm_partials_f16(a.data_ptr(), W3.data_ptr(), rows, n_out, k, part.data_ptr(), None), "rc_split_gemm_partials_f16")
    torch.cuda.synchronize()
    assert bool(torch.isnan(part[2 * rows]).all()) and not bool(torch.isnan(part[:2 * rows]).any())
    y = part[rows:2 * rows].double() + part[:rows].double() / 2048.0
    ref = (xh.double() + xl.double() / 2048.0) @ (wh.double() + wl.double() / 2048.0).t()
    assert float((y.cpu() - ref).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max()))
    whole = torch.empty((rows, n_out), device="cuda")
    zero = torch.zeros(n_out, device="cuda")
    _hip.check(lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), zero.data_ptr(), rows, n_out, k, 0, 1.0, None, whole.data_ptr(), 1, None),
               "rc_split_gemm_f16")
    assert float((y.float() - whole).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert lib.rc_split_gemm_partials_f16(a.data_ptr(), W3.data_ptr(), rows, n_out, 64, part.data_ptr(), None) == -4   # k % 128
    assert lib.rc_split_gemm_partials_f16(a.data_ptr(), W3.data_ptr(), rows, 128, k, part.data_ptr(), None) == -4      # n_out % 256


def test_split_engine_own_kernels_on_fc_big():
    """fc_big (480 -> 8192 -> 4096 -> 2048, heads 1024 -> 512): every hidden layer of the split engine runs on the own kernel at a
    full batch (K up to 8192, the last one in the K-cut form) and the outputs stay closer to float64 than the fp32 module's."""
    import copy
    from librubiks import cube
    from librubiks.model import F32_SPLIT, Model, ModelConfig, make_inference_net
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig(architecture="fc_big")).eval()
    cubes, _, _ = cube.scramble_batch(11264, 30, True)
    oh = cubes.as_oh(torch.float32)[:1024]
    eng = make_inference_net(net, F32_SPLIT)
    assert [eng._layer_plan(cubes.n, eng.layers, i) for i in range(1, len(eng.layers) - 1)] == ["fused", "fused", "fused", ("cut", 1, 2)]
    with torch.no_grad():
        p64, v64 = copy.deepcopy(net).double()(oh.double())
        p32, v32 = net(oh)
    ps, vs = eng.forward_cubes(cubes)
    err = lambda a, b: float((a.double() - b).abs().max())   # noqa: E731
    e_split = max(err(ps[:1024], p64), err(vs[:1024], v64.reshape(-1)))
    e_f32 = max(err(p32, p64), err(v32.reshape(-1), v64.reshape(-1)))
    assert e_split <= 1.25 * e_f32 + 1e-7


def _net_with_huge_activations(factor):
    """fc_small whose second shared layer is scaled so that its ELU outputs are of order `factor` (BatchNorm behind it has
    unit statistics, so the network stays a finite fp32 function)."""
    from librubiks.model import Model, ModelConfig
    torch.manual_seed(3)
    net = Model.create(ModelConfig()).eval()
    with torch.no_grad():
        net.shared_net[3].weight.mul_(factor)
    return net


def test_split_engine_flags_activations_beyond_half_range():
    """
    The f16x3 split carries a value as IEEE halves hi + lo 2^-11: an activation beyond +-65504 cannot be represented.  Every
    kernel that writes the format (input layer, own hidden-layer kernel at full batches, the reduce kernel behind library
    GEMMs / K-split partials) must raise the engine's device flag; ordinary networks must not.  Weights beyond half range are
    refused when the engine is built and make_inference_net falls back to fp32 with a warning.
    """
    from librubiks import cube
    from librubiks.model import F32_SPLIT, InferenceNet, SplitF32Net, make_inference_net
    np.random.seed(1)
    big, _, _ = cube.scramble_batch(11264, 20, True)      # full batch: the fused kernels
    small, _, _ = cube.scramble_batch(300, 20, True)      # small batch: library GEMMs + reduce kernel
    mid, _, _ = cube.scramble_batch(5632, 20, True)       # K cut in two + reduce kernel
    ok = SplitF32Net(_model())
    for cubes in (big, mid, small):
        ok.forward_cubes(cubes)
    assert not ok.overflowed()
    hot = SplitF32Net(_net_with_huge_activations(2.0e5))
    for cubes in (big, mid, small):
        assert not hot.overflowed()
        p, v = hot.forward_cubes(cubes)
        assert hot.overflowed() and not hot.overflowed()      # raised by this forward, cleared by the read
        # what the fallback computes is the fp32 engine's answer, finite
        pf, vf = hot.fallback().forward_cubes(cubes) if hot.fallback().supports_cubes else hot.fallback()(cubes.as_oh(torch.float32))
        assert bool(torch.isfinite(pf).all()) and bool(torch.isfinite(vf).all())
    # input layer alone: a first-layer bias beyond half range
    net = _model()
    with torch.no_grad():
        net.shared_net[0].bias.add_(1.0e5)
    eng = SplitF32Net(net)
    eng._first_from_cubes(small, eng.layers)
    assert eng.overflowed()
    # weights that do not fit IEEE half: refused at build time, fp32 instead
    with pytest.raises(Exception):
        SplitF32Net(_net_with_huge_activations(1.0e7))
    with pytest.warns(RuntimeWarning):
        e = make_inference_net(_net_with_huge_activations(1.0e7), F32_SPLIT)
    assert isinstance(e, InferenceNet) and e.dtype == torch.float32


def test_agents_repeat_the_search_in_fp32_when_the_split_engine_overflows():
    """The default engine of every deep agent is the split engine; when an activation leaves half range the agent warns and
    returns what the fp32 engine returns -- tree for tree, queue for queue."""
    import warnings
    from librubiks import cube
    from librubiks.model import F32_SPLIT, InferenceNet
    from librubiks.solving.agents import MCTS, AStar, ValueSearch
    np.random.seed(2)
    cubes, _, _ = cube.scramble_batch(48, 6, True)
    states = cubes.numpy()
    net = _net_with_huge_activations(2.0e5)
    for make in (lambda dt: MCTS(net, c=0.6, search_graph=True, **dt), lambda dt: AStar(net, lambda_=0.2, expansions=20, **dt),
                 lambda dt: ValueSearch(net, **dt)):
        agent = make({})
        assert agent.net_dtype == F32_SPLIT
        with pytest.warns(RuntimeWarning, match="half"):
            got = agent.search_batch(states, None, 600)
        with warnings.catch_warnings():
            warnings.simplefilter("error")           # the second search goes straight to fp32: no overflow, no warning
            again = agent.search_batch(states, None, 600)
            ref = make({"net_dtype": torch.float32}).search_batch(states, None, 600)
        for r in (got, again):
            assert np.array_equal(r.solved, ref.solved) and np.array_equal(r.nodes, ref.nodes) and np.array_equal(r.lengths, ref.lengths)
            assert all(list(a) == list(b) for a, b in zip(r.queues, ref.queues))
        assert isinstance(agent._fp32_for[1], InferenceNet)


def _res_model(arch="res_small", batchnorm=True, seed=0):
    from librubiks.model import Model, ModelConfig
    torch.manual_seed(seed)
    m = Model.create(ModelConfig(architecture=arch, batchnorm=batchnorm)).eval()
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_((torch.randn(mod.num_features, generator=g) * 0.3).cuda())
            mod.running_var.copy_((torch.rand(mod.num_features, generator=g) + 0.5).cuda())
            mod.weight.data.copy_((torch.rand(mod.num_features, generator=g) + 0.5).cuda())
            mod.bias.data.copy_((torch.randn(mod.num_features, generator=g) * 0.2).cuda())
    return m


@pytest.mark.parametrize("batchnorm", [True, False])
def test_residual_architectures_run_on_the_native_engines(batchnorm):
    """
    res_small (reference model.py:249-264: 480 -> 4096 -> 1024, four NonConvResBlocks of 1024 -> 1024 x 2, heads) on the engines
    the fc nets use: BatchNorm folded (a block's own into its Linear; the one in front of the first block as a post-activation
    affine, because the skip connection reads it too), skip connection added in the layer kernels' epilogue.  Tolerances: fp32
    engine vs the module rtol = atol = 1e-4; split engine vs float64 within 1.25 x the fp32 module's error; bf16 atol 4e-2 on
    outputs of magnitude ~1.  All row counts: own kernels (11 264 rows), K cut in two + reduce (5 632), K cut into 16-32 chunks (300).
    """
    import copy
    from librubiks import cube
    from librubiks.model import F32_SPLIT, GenericNet, InferenceNet, SplitF32Net, make_inference_net
    net = _res_model(batchnorm=batchnorm)
    np.random.seed(4)
    ref64 = copy.deepcopy(net).double()
    engines = {dt: make_inference_net(net, dt) for dt in (F32_SPLIT, torch.float32, torch.bfloat16)}
    assert isinstance(engines[F32_SPLIT], SplitF32Net) and isinstance(engines[torch.float32], InferenceNet)
    assert not any(isinstance(e, GenericNet) for e in engines.values()) and all(e.residual for e in engines.values())
    assert engines[torch.bfloat16].supports_cubes and engines[F32_SPLIT].supports_cubes          # fused input layer from the cube states
    for n in (11264, 5632, 300):
        cubes, _, _ = cube.scramble_batch(n, 25, True)
        oh = cubes.as_oh(torch.float32)
        with torch.no_grad():
            p64, v64 = ref64(oh.double())
            p32, v32 = net(oh)
        err = lambda a, b: float((a.double() - b).abs().max())   # noqa: E731
        e_f32 = max(err(p32, p64), err(v32.reshape(-1), v64.reshape(-1)))
        ps, vs = engines[F32_SPLIT].forward_cubes(cubes)
        assert not engines[F32_SPLIT].overflowed()
        e_split = max(err(ps, p64), err(vs, v64.reshape(-1)))
        pf, vf = engines[torch.float32](oh)
        e_fold = max(err(pf, p64), err(vf, v64.reshape(-1)))
        pb, vb = engines[torch.bfloat16].forward_cubes(cubes)
        e_bf = max(err(pb, p64), err(vb, v64.reshape(-1)))
        scale = max(1.0, float(p64.abs().max()), float(v64.abs().max()))
        print(f"res_small bn={batchnorm} n={n}: |out| <= {scale:.2f}; max |err| vs float64: split {e_split:.2e} fp32 module {e_f32:.2e} "
              f"fp32 engine {e_fold:.2e} bf16 {e_bf:.2e}")
        assert e_split <= 1.25 * e_f32 + 1e-7 * scale
        assert e_fold <= 1e-4 * scale
        assert e_bf <= 4e-2 * scale
        assert err(engines[F32_SPLIT].value_cubes(cubes), v64.reshape(-1)) <= 1.25 * e_f32 + 1e-7 * scale
    # the own bf16 layer kernel (opt-in) with the skip connection in its epilogue: same numbers as the library path up to bf16 rounding
    eng = engines[torch.bfloat16]
    cubes, _, _ = cube.scramble_batch(11264, 25, True)
    base = eng.forward_cubes(cubes)
    eng.fused_hidden = True
    own = eng.forward_cubes(cubes)
    eng.fused_hidden = False
    assert float((own[0] - base[0]).abs().max()) <= 4e-2 * scale and float((own[1] - base[1]).abs().max()) <= 4e-2 * scale


def test_mcts_on_a_residual_net_uses_the_fused_path_and_equals_the_oracle():
    """MCTS with res_small on the split engine: packed 11 rows per tree, fused input layer and head, trees exact vs the oracle replayed on
    the recorded network outputs."""
    from test_search_edge_gpu import _TableNet, _compare
    from librubiks.solving.agents import MCTS
    from oracle import agents as oa
    net = _res_model(seed=1)
    np.random.seed(8)
    states = np.array([oc.scramble(3 + i % 5, True)[0] for i in range(40)])
    agent = MCTS(net, c=0.6, search_graph=True)
    res = agent.search_batch(states, None, 400, compact=False)
    assert agent.forest._fused and agent.forest.rows_per_tree == 11 and agent.forest.engine.residual
    for t in range(40):
        tree = agent.forest.tree_arrays(t)
        n = tree["n"]
        table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
        ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=True)
        ok = ref.search(states[t], 400)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n and list(res.queues[t]) == list(ref.action_queue)
        _compare(tree, ref, n)


def test_layer_request_k_splits_and_reduce_match_the_whole_kernel():
    """rc_split_layer_f16 with out_partials and k_splits = 2 .. 18 in both tile shapes + rc_split_reduce_f16 (ordered sum, the first
    rc_split_layer_corr_chunks partials scaled by 2^-11, bias, skip connection, ELU, post-activation affine, re-split) against the
    whole-K kernel with the same epilogue options, and the C-ABI's argument checks for both."""
    from librubiks import _hip
    from librubiks.model import _layer_call
    lib = _hip.lib()
    g = torch.Generator().manual_seed(11)
    rows, k, n_out = 700, 768, 512                      # 3 k / 64 = 36 K-steps
    halves = lambda v: (v.half(), ((v - v.half().double()) * 2048.0).half())   # noqa: E731
    x = torch.randn(rows, k, generator=g).double() * 0.7
    W = torch.randn(n_out, k, generator=g).double() / np.sqrt(k)
    b = torch.randn(n_out, generator=g).cuda()
    skip = torch.randn(rows, n_out, generator=g).double()
    ps, pt = (torch.rand(n_out, generator=g) + 0.5).cuda(), torch.randn(n_out, generator=g).cuda()
    (xh, xl), (wh, wl), (sh, sl) = halves(x), halves(W), halves(skip)
    a, w3, res = torch.cat([xh, xl], 1).cuda(), torch.cat([wl, wh, wh], 1).cuda(), torch.cat([sh, sl], 1).cuda()
    whole = torch.empty((rows, 2 * n_out), dtype=torch.float16, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _layer_call("rc_split_layer_f16", a=a, w=w3, bias=b, residual=res, post_scale=ps, post_shift=pt, n_rows=rows, n_out=n_out, k=k,
                activation=2, alpha=1.0, out_hi_lo=whole, tile=0, k_splits=1, range_flag=flag)
    y_whole = whole[:, :n_out].double() + whole[:, n_out:].double() / 2048
    # float64 reference of the same function of the same (split) operands
    z = (xh.double() + xl.double() / 2048) @ (wh.double() + wl.double() / 2048).t() + b.cpu().double() + (sh.double() + sl.double() / 2048)
    ref = torch.where(z > 0, z, torch.expm1(z)) * ps.cpu().double() + pt.cpu().double()
    assert float((y_whole.cpu() - ref).abs().max()) < 2e-5 and int(flag.item()) == 0
    parts = {}
    for S, tile in ((2, 0), (3, 1), (4, 0), (6, 0), (4, 3), (12, 3), (18, 3), (4, 7), (9, 7), (18, 7), (2, 7)):   # tile 3: 352 x 128 tiles, 7: 352 x 64
        part = torch.full((S, rows, n_out), float("nan"), device="cuda")
        _layer_call("rc_split_layer_f16", a=a, w=w3, n_rows=rows, n_out=n_out, k=k, out_partials=part, k_splits=S, tile=tile)
        assert torch.equal(parts.setdefault(S, part), part), (S, tile)      # a chunk's partial sums do not depend on the tile's shape
        n_corr = lib.rc_split_layer_corr_chunks(k, S)
        assert n_corr == sum((p + 1) * (36 // S) <= 24 for p in range(S))
        out = torch.empty((rows, 2 * n_out), dtype=torch.float16, device="cuda")
        f32 = torch.empty((rows, n_out), device="cuda")
        _hip.check(lib.rc_split_reduce_f16(part.data_ptr(), rows * n_out, S, n_corr, rows, n_out, b.data_ptr(), res.data_ptr(), 2, 1.0,
                                           ps.data_ptr(), pt.data_ptr(), out.data_ptr(), f32.data_ptr(), flag.data_ptr(), None), "rc_split_reduce_f16")
        y = out[:, :n_out].double() + out[:, n_out:].double() / 2048
        assert float((y - y_whole).abs().max()) < 4e-6 * max(1.0, float(y_whole.abs().max())), S      # summation order differs, nothing else
        assert float((f32.double() - y).abs().max()) < 2e-6 * max(1.0, float(y.abs().max()))          # fp32 output vs its 22-bit split
    assert int(flag.item()) == 0
    # argument checks
    L = dict(a=a, w=w3, n_rows=rows, n_out=n_out, k=k)
    bad = lambda **kw: pytest.raises(_hip.RubiksHipError, _layer_call, "rc_split_layer_f16", **{**L, **kw})   # noqa: E731
    part = torch.empty((2, rows, n_out), device="cuda")
    bad(out_partials=part, k_splits=5)                                   # 36 K-steps are not divisible by 5
    bad(out_partials=part, k_splits=1)                                   # partials need at least two chunks
    bad(out_partials=part, k_splits=36)                                  # ... of at least two K-steps each
    bad(out_partials=part, k_splits=2, tile=2)                           # partials come in 352 x 256, 352 x 128 and 352 x 64 tiles only
    bad(out_partials=part, out_hi_lo=whole, k_splits=2)                  # exactly one output
    bad(bias=b, out_hi_lo=whole, post_scale=ps)                          # post_scale without post_shift
    bad(out_hi_lo=whole)                                                 # the fused epilogue needs a bias
    assert lib.rc_split_layer_corr_chunks(100, 2) == -1
    red = lambda **kw: lib.rc_split_reduce_f16(part.data_ptr(), kw.get("stride", rows * n_out), kw.get("P", 2), kw.get("nc", 1), rows,   # noqa: E731
                                               kw.get("cols", n_out), b.data_ptr(), None, kw.get("act", 2), 1.0, None, None,
                                               kw.get("o", whole.data_ptr()), None, None, None)
    assert red() == 0 and red(o=None) == -1 and red(nc=3) == -4 and red(P=0) == -4 and red(act=7) == -4 and red(cols=n_out + 4) == -2
    assert red(stride=rows * n_out - 4) == -4


def test_reduce_kernel_adds_the_partials_in_order_bit_for_bit():
    """rc_split_reduce_f16 with fp32 output and no activation against the same fp32 sum in torch, in the order p = 0, 1, ... with the
    factor 2^-11 applied once the first n_corr partials are in: equal bit for bit for every partial count the kernel's load groups
    (8, 4, 2, 1 partials in flight) split differently, and for every place of the scaling inside a group."""
    from librubiks import _hip
    lib = _hip.lib()
    g = torch.Generator().manual_seed(17)
    rows, cols = 37, 264
    bias = torch.randn(cols, generator=g).cuda()
    for P in (1, 2, 3, 5, 7, 8, 9, 13, 16, 23, 32):
        part = (torch.randn(P, rows, cols, generator=g) * 3.0).cuda()
        for n_corr in sorted({0, 1, P // 2, P - 1, P}):
            ref = torch.zeros(rows, cols, device="cuda")
            for p in range(P):
                if p == n_corr:
                    ref = ref * (1.0 / 2048.0)
                ref = ref + part[p]
            if n_corr >= P:
                ref = ref * (1.0 / 2048.0)
            ref = ref + bias
            out = torch.full((rows, cols), float("nan"), device="cuda")
            _hip.check(lib.rc_split_reduce_f16(part.data_ptr(), rows * cols, P, n_corr, rows, cols, bias.data_ptr(), None, 0, 1.0, None, None,
                                               None, out.data_ptr(), None, None), "rc_split_reduce_f16")
            assert torch.equal(out, ref), (P, n_corr)


def test_split_engine_small_batches_run_the_cut_kernel():
    """Below the whole-K tile's fill point the hidden layers run as the own kernel with its K loop cut into chunks (~256 workgroups)
    + the reduce kernel -- the launches of a narrowed forest.  Every plan the ladder reaches is checked against float64 (closer than
    the fp32 module), and the plans themselves are pinned."""
    import copy
    from librubiks import cube
    from librubiks.model import F32_SPLIT, Model, ModelConfig, SplitF32Net, make_inference_net
    torch.manual_seed(3)
    np.random.seed(3)
    net = Model.create(ModelConfig(architecture="fc_small")).eval().cuda()
    eng = make_inference_net(net, F32_SPLIT)
    assert SplitF32Net._k_split(352, 2048, 4096) == (3, 16) and SplitF32Net._k_split(352, 1024, 2048) == (7, 16)
    assert SplitF32Net._k_split(704, 1024, 2048) == (3, 16)
    assert SplitF32Net._k_split(2816, 2048, 4096) == (1, 4) and SplitF32Net._k_split(2816, 1024, 2048) == (3, 4)
    assert SplitF32Net._k_split(5632, 2048, 4096) == (1, 2) and SplitF32Net._k_split(7040, 2048, 4096) is None
    seen = set()
    for trees in (32, 64, 96, 128, 192, 256, 320, 384, 512, 640, 1024):
        rows = trees * 11
        cubes, _, _ = cube.scramble_batch(rows, 25, True)
        plans = tuple(eng._layer_plan(rows, eng.layers, i) for i in (1, 2))
        seen.update(plans)
        oh = cubes.as_oh(torch.float32)
        with torch.no_grad():
            p64, v64 = copy.deepcopy(net).double()(oh.double())
            p32, v32 = net(oh)
        ps, vs = eng.forward_cubes(cubes)
        e_split = max(float((ps.double() - p64).abs().max()), float((vs.double() - v64.reshape(-1)).abs().max()))
        e_f32 = max(float((p32.double() - p64).abs().max()), float((v32.reshape(-1).double() - v64.reshape(-1)).abs().max()))
        assert e_split < 2e-5 and e_split <= 2.0 * e_f32 + 1e-6, (trees, plans, e_split, e_f32)
        vo = eng.value_cubes(cubes)
        assert float((vo.double() - v64.reshape(-1)).abs().max()) < 2e-5
    assert "library" not in seen and "fused" in seen and sum(isinstance(p, tuple) for p in seen) >= 6, seen
    assert not eng.overflowed()


def test_layer_request_one_product_and_the_input_layer_on_it():
    """rc_split_layer_t.products = 1: the GEMM kernel as ONE f16 product of a [n][k] and w [n_out][k] (fp32 accumulation, the split
    epilogue) against float64; then the input layer of the split engine both ways -- fused one-hot kernel / explicit one-hot operand on
    the GEMM kernel (big batches) -- the same sums in another order."""
    from librubiks import _hip, cube
    from librubiks.model import F32_SPLIT, Model, ModelConfig, SplitF32Net, _layer_call, make_inference_net
    g = torch.Generator().manual_seed(5)
    rows, k, n_out = 1000, 960, 512
    a = (torch.randn(rows, k, generator=g) * 0.5).half().cuda()
    w = (torch.randn(n_out, k, generator=g) / 30).half().cuda()
    b = torch.randn(n_out, generator=g).cuda()
    ref = a.double() @ w.double().t() + b.double()
    ref = torch.where(ref > 0, ref, torch.expm1(ref))
    for tile in (0, 1, 2, 3):
        out = torch.empty((rows, 2 * n_out), dtype=torch.float16, device="cuda")
        _layer_call("rc_split_layer_f16", a=a, w=w, bias=b, n_rows=rows, n_out=n_out, k=k, activation=2, alpha=1.0, out_hi_lo=out, tile=tile, k_splits=1,
                    products=1)
        y = out[:, :n_out].double() + out[:, n_out:].double() / 2048
        assert float((y - ref).abs().max()) < 3e-6, tile
    f32 = torch.empty((rows, n_out), device="cuda")
    _layer_call("rc_split_layer_f16", a=a, w=w, bias=b, n_rows=rows, n_out=n_out, k=k, activation=2, alpha=1.0, out_f32=f32, tile=0, k_splits=1, products=1)
    assert float((f32.double() - ref).abs().max()) < 3e-6
    with pytest.raises(_hip.RubiksHipError):
        _layer_call("rc_split_layer_f16", a=a, w=w, bias=b, n_rows=rows, n_out=n_out, k=k, activation=2, alpha=1.0, out_f32=f32, products=2)
    # the engine's input layer
    torch.manual_seed(1)
    np.random.seed(1)
    net = Model.create(ModelConfig(architecture="fc_small")).eval().cuda()
    eng = make_inference_net(net, F32_SPLIT)
    cubes, _, _ = cube.scramble_batch(11264, 30, True)
    fused = eng._first_from_cubes(cubes, eng.layers)
    # the same layer as explicit one-hot operand [oh | 2^-11 oh] (rc_oh_split_f16) x [W_hi | W_lo] through the GEMM kernel as ONE product
    # with K = 960: an independent implementation of the fused kernel's function (the same sums in another order).  Measured 2 % slower
    # in whole searches (profiles/r3_input_gemm_ab.txt), so the engine does not use it.
    _, B, b0, code, alpha = eng.layers[0][:5]
    oh = torch.empty((cubes.n, 960), dtype=torch.float16, device="cuda")
    _hip.check(_hip.lib().rc_oh_split_f16(cubes.soa.data_ptr(), cubes.n, cubes.stride, oh.data_ptr(), _hip.stream_ptr()), "rc_oh_split_f16")
    on_gemm = torch.empty_like(fused)
    _layer_call("rc_split_layer_f16", a=oh, w=B, bias=b0, n_rows=cubes.n, n_out=B.shape[0], k=960, activation=code, alpha=alpha, out_hi_lo=on_gemm,
                tile=1, k_splits=1, products=1, range_flag=eng.range_flag)
    H = fused.shape[1] // 2
    val = lambda t: t[:, :H].double() + t[:, H:].double() / 2048   # noqa: E731
    assert float((val(on_gemm) - val(fused)).abs().max()) < 1e-6
    assert not eng.overflowed()


def test_engines_at_every_launch_size_of_a_narrowing_search():
    """
    The layer plan of the engines depends on the row count (own kernels on whole tiles, K cut in 2 .. 32 chunks + reduce for small
    batches, the library chain in between: DESIGN.md 3.3), and a narrowing search walks through all of them: every rung of a
    1 024-tree forest (11 rows per tree), plus the row counts either side of every plan boundary and a few odd ones.  At each size
    the split engine must stay within 1.25 x the fp32 module's own error against float64 (its contract), the bf16 engine within
    2 % of the largest output, and padding rows must not leak: the first n rows of a larger launch equal the rows of the
    n-row launch to fp32 rounding.
    """
    import copy
    import os
    from conftest import ROOT
    from librubiks import cube
    from librubiks.cube.device import DeviceCubes
    from librubiks.model import F32_SPLIT, Model, ModelConfig, make_inference_net
    from librubiks.solving.mcts_device import rungs
    torch.manual_seed(0)
    np.random.seed(0)
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
    ref64 = copy.deepcopy(net).double()
    split, bf16 = make_inference_net(net, F32_SPLIT), make_inference_net(net, torch.bfloat16)
    sizes = sorted({11 * r for r in rungs(1024)} | {1, 2, 11, 12, 351, 352, 353, 703, 704, 705, 1407, 1408, 1409, 2815, 2816, 2817,
                                                      4097, 5631, 5632, 5633, 11263, 11265, 12288, 20000})
    cubes, _, _ = cube.scramble_batch(max(sizes), 30, True)
    oh = cubes.as_oh(torch.float32)
    with torch.no_grad():
        p64, v64 = ref64(oh.double())
        p32, v32 = net(oh)
    v64, v32 = v64.reshape(-1), v32.reshape(-1)
    scale = max(1.0, float(p64.abs().max()), float(v64.abs().max()))
    worst = {"split": 0.0, "bf16": 0.0, "fp32": 0.0}
    for n in sizes:
        e_f32 = max(float((p32[:n].double() - p64[:n]).abs().max()), float((v32[:n].double() - v64[:n]).abs().max()))
        for name, eng in (("split", split), ("bf16", bf16)):
            p, v = eng.forward_cubes(DeviceCubes(cubes.soa, n))        # the first n states of the same SoA
            assert p.shape == (n, 12) and v.shape == (n,), (name, n)
            e = max(float((p.double() - p64[:n]).abs().max()), float((v.double() - v64[:n]).abs().max()))
            worst[name] = max(worst[name], e)
            if name == "split":
                assert not eng.overflowed() and e <= 1.25 * e_f32 + 2e-7 * scale, (n, e, e_f32)
            else:
                assert e <= 2e-2 * scale, (n, e)        # bf16: 2^-8 per rounding, five layers
        worst["fp32"] = max(worst["fp32"], e_f32)
        # a column window in the middle of the SoA (what a forest's rows are for A*): same rows, same numbers to fp32 rounding
        if 16 <= n <= max(sizes) // 2:
            off = n // 2 // 16 * 16                      # windows start on a multiple of 16 states
            vw = split.value_cubes(cubes, None, off, n)
            assert float((vw.double() - v64[off:off + n]).abs().max()) <= 1.25 * float((v32[off:off + n].double() - v64[off:off + n]).abs().max()) + 2e-7 * scale, n
    print(f"{len(sizes)} launch sizes, |out| <= {scale:.2f}: worst max |err| vs float64: split {worst['split']:.2e}, fp32 module {worst['fp32']:.2e}, bf16 {worst['bf16']:.2e}")


@pytest.mark.parametrize("arch,bn,stats", [("fc_small", True, "trained_stats"), ("fc_small", True, "fresh"), ("fc_small", False, "fresh"),
                                           ("res_small", True, "trained_stats"), ("fc_big", True, "trained_stats")])
def test_engines_against_outputs_of_the_references_model(arch, bn, stats):
    """
    a11 against the REFERENCE: tests/golden/model_golden.npz holds what the imported reference `Model` (librubiks/model.py:106-161)
    returns on the 256 golden `oh_in` states under torch.manual_seed(0) -- fp32 as the reference runs it, and the same module in
    float64.  The build's Model has those parameters bit for bit (tests/test_model.py checks the hashes on the CPU; here again on the
    state_dict that feeds the engines), and every engine is held to the fixture:
      SplitF32Net   max |err| vs the reference's float64 outputs <= 1.25 x the error of the reference's OWN fp32 outputs (+1e-7)
      fp32 chain    rtol = atol = 1e-4 against the reference's fp32 outputs
      bf16          atol 3e-2 on logits and values of magnitude <= 1 (glorot weights)
    """
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from test_model import seeded_reference_model
    from librubiks.cube import DeviceCubes
    from librubiks.model import F32_SPLIT, InferenceNet, SplitF32Net, make_inference_net
    fx = np.load(os.path.join(GOLDEN, "model_golden.npz"))
    name = f"{arch}_bn{int(bn)}"
    meta = json.loads(str(fx["meta_json"]))[name]
    net = seeded_reference_model(arch, bn, fx, trained_stats=(stats == "trained_stats"))
    for k, t in net.state_dict().items():
        if stats == "fresh" or not ("running_" in k or "num_batches" in k):
            assert hashlib.sha256(t.cpu().contiguous().numpy().tobytes()).hexdigest() == meta["sha256"][k], k
    net = net.cuda().eval()
    states = np.load(os.path.join(GOLDEN, "cube_golden.npz"))["oh_in"]
    cubes = DeviceCubes.from_numpy(states)
    oh = cubes.as_oh(torch.float32)
    tag = f"{name}_{stats}"
    p32, v32, p64, v64 = (fx[f"{tag}_{k}"] for k in ("p32", "v32", "p64", "v64"))
    e_ref32 = max(np.abs(p32 - p64).max(), np.abs(v32 - v64).max())
    err = lambda got, want: float(np.abs(got.double().cpu().numpy().reshape(want.shape) - want).max())   # noqa: E731
    # the module itself on the GPU (torch's fp32 GEMMs): the function the fixture pins
    with torch.no_grad():
        pm, vm = net(oh)
    assert err(pm, p32) <= 1e-4 and err(vm, v32) <= 1e-4
    split = make_inference_net(net, F32_SPLIT)
    assert isinstance(split, SplitF32Net)
    for p, v in (split.forward_cubes(cubes), split(oh)):
        e = max(err(p, p64), err(v, v64))
        print(f"{tag}: split engine {e:.3e}, the reference's fp32 forward {e_ref32:.3e} from its float64 forward")
        assert e <= 1.25 * e_ref32 + 1e-7
    assert err(split.value_cubes(cubes), v64) <= 1.25 * e_ref32 + 1e-7
    f32 = InferenceNet(net, torch.float32)
    for p, v in ((f32.forward_cubes(cubes), f32(oh)) if f32.supports_cubes else (f32(oh),)):
        assert np.allclose(p.cpu().numpy(), p32, rtol=1e-4, atol=1e-4) and np.allclose(v.cpu().numpy().reshape(-1, 1), v32, rtol=1e-4, atol=1e-4)
    bf = make_inference_net(net, torch.bfloat16)
    p, v = bf.forward_cubes(cubes)
    assert err(p.float(), p32) <= 3e-2 and err(v.float(), v32) <= 3e-2


def test_deterministic_split_engine_gives_a_state_the_same_outputs_in_every_launch():
    """SplitF32Net(deterministic=True): the K-cut form with a fixed chunk count at every row count.  The same 352 states evaluated
    inside launches of 352 ... 11 264 rows (every layer-plan boundary of the default engine lies in between), at different row
    offsets, as policy + value and as value only, return bit-identical numbers; and they are as close to float64 as the default
    engine's (<= 1.25 x the fp32 module's error)."""
    import copy
    import os
    from conftest import ROOT
    from librubiks import cube
    from librubiks.cube import DeviceCubes
    from librubiks.model import F32_SPLIT_DET, Model, ModelConfig, SplitF32Net, make_inference_net
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    torch.manual_seed(0)
    net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
    eng = make_inference_net(net, F32_SPLIT_DET)
    assert isinstance(eng, SplitF32Net) and eng.deterministic and eng.dtype == F32_SPLIT_DET
    np.random.seed(5)
    cubes, _, _ = cube.scramble_batch(11264, 25, True)
    states = cubes.numpy()
    base = states[:352]
    want_p, want_v = eng.forward_cubes(DeviceCubes.from_numpy(base))
    want_val = eng.value_cubes(DeviceCubes.from_numpy(base))     # the value head alone (A*): its own layers, its own -- equally fixed -- sums
    assert float((want_val - want_v).abs().max()) <= 4e-6 * float(want_v.abs().max())
    for rows, at in ((352, 0), (368, 16), (1056, 352), (1408, 1056), (2816, 0), (4224, 2464), (5632, 5280), (6336, 352), (11264, 10912), (11264, 0)):
        batch = states[:rows].copy()
        batch[at:at + 352] = base
        p, v = eng.forward_cubes(DeviceCubes.from_numpy(batch))
        assert torch.equal(p[at:at + 352], want_p) and torch.equal(v[at:at + 352], want_v), (rows, at)
        assert torch.equal(eng.value_cubes(DeviceCubes.from_numpy(batch))[at:at + 352], want_val), (rows, at)
    one_p, one_v = eng.forward_cubes(DeviceCubes.from_numpy(base[:1]))
    assert torch.equal(one_p, want_p[:1]) and torch.equal(one_v, want_v[:1])
    oh = DeviceCubes.from_numpy(states[:4096]).as_oh(torch.float32)
    with torch.no_grad():
        p64, v64 = copy.deepcopy(net).double()(oh.double())
        p32, v32 = net(oh)
    pd, vd = eng.forward_cubes(DeviceCubes.from_numpy(states[:4096]))
    err = lambda a, b: float((a.double() - b).abs().max())   # noqa: E731
    assert max(err(pd, p64), err(vd, v64.reshape(-1))) <= 1.25 * max(err(p32, p64), err(v32.reshape(-1), v64.reshape(-1))) + 1e-7


def test_deterministic_mode_has_no_silent_way_out():
    """deterministic=True is a promise (one summation order per output, whatever the batch).  It is checked when the engine is BUILT:
    a network with a layer that would fall to a library GEMM raises there -- not in the middle of a search or of a graph capture --,
    the one-hot entry points run the fused input kernel too, and the fp32 fallback behind a half-range overflow is refused."""
    from librubiks.cube import DeviceCubes
    from librubiks.model import F32_SPLIT_DET, Model, ModelConfig, SplitF32Net, SplitRangeError, make_inference_net
    torch.manual_seed(0)
    ok = Model.create(ModelConfig()).eval()
    eng = SplitF32Net(ok, deterministic=True)
    np.random.seed(3)
    from librubiks import cube
    cubes, _, _ = cube.scramble_batch(640, 20, True)
    p, v = eng.forward_cubes(cubes)
    oh = cubes.as_oh(torch.float32)
    p2, v2 = eng(oh)                                  # the one-hot API: decoded, then the same kernels
    assert torch.equal(p, p2) and torch.equal(v, v2) and torch.equal(eng.value(oh[:17]), eng.value_cubes(DeviceCubes.from_numpy(cubes.numpy()[:17])))
    with pytest.raises(SplitRangeError, match="deterministic"):
        eng.fallback()
    # a hidden width the K-cut kernel does not take (n_out % 128 != 0): refused at construction

    class Odd(SplitF32Net):                           # the first hidden layer cut to 2 000 outputs
        def _split(self, layers, ref_opts):
            out = super()._split(layers, ref_opts)
            kind, Wh, B2, b, code, alpha, W3 = out[1]
            out[1] = (kind, Wh[:2000].contiguous(), B2[:2000].contiguous(), b[:2000].contiguous(), code, alpha, W3[:2000].contiguous())
            return out
    with pytest.raises(SplitRangeError, match="deterministic"):
        Odd(ok, deterministic=True)
    assert isinstance(make_inference_net(ok, F32_SPLIT_DET), SplitF32Net)
