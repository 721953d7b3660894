"""Input layer of the split engine: the fused one-hot MFMA kernel (rc_first_layer_split_flag_f16) against the explicit one-hot operand
(rc_oh_split_f16) + the hidden layers' GEMM kernel run as ONE f16 product (rc_split_layer_t.products = 1, K = 960), alone and inside a
loop with the next layer (in-flow clocks).   python tools/first_layer_gemm_probe.py"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip, cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model, _layer_call, make_inference_net  # noqa: E402

lib = _hip.lib()
np.random.seed(0)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
eng = make_inference_net(model, F32_SPLIT)
_, B, b, code, alpha, Wh, Wl = eng.layers[0]
H = Wh.shape[0]
_, Wh1, B21, b1, code1, alpha1, W31 = eng.layers[1]


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for rows in (352, 1408, 2816, 5632, 11264):
    cubes, _, _ = cube.scramble_batch(rows, 25, True)
    out_f = torch.empty((rows, 2 * H), dtype=torch.float16, device="cuda")
    out_g = torch.empty_like(out_f)
    oh = torch.empty((rows, 960), dtype=torch.float16, device="cuda")
    nxt = torch.empty((rows, 2 * Wh1.shape[0]), dtype=torch.float16, device="cuda")

    def fused():
        _hip.check(lib.rc_first_layer_split_flag_f16(cubes.soa.data_ptr(), rows, cubes.stride, Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), out_f.data_ptr(), H, code, alpha,
                                                     None, _hip.stream_ptr()))

    def gemm(tile):
        _hip.check(lib.rc_oh_split_f16(cubes.soa.data_ptr(), rows, cubes.stride, oh.data_ptr(), _hip.stream_ptr()))
        _layer_call("rc_split_layer_f16", a=oh, w=B, bias=b, n_rows=rows, n_out=H, k=960, activation=code, alpha=alpha, out_hi_lo=out_g, tile=tile, k_splits=1, products=1)

    def layer1(x):
        _layer_call("rc_split_layer_f16", a=x, w=W31, bias=b1, n_rows=rows, n_out=Wh1.shape[0], k=Wh1.shape[1], activation=code1, alpha=alpha1, out_hi_lo=nxt, tile=1, k_splits=1)

    fused()
    gemm(1)
    yf = out_f[:, :H].double() + out_f[:, H:].double() / 2048
    yg = out_g[:, :H].double() + out_g[:, H:].double() / 2048
    line = f"rows {rows:6d}: max |fused - gemm| {float((yf - yg).abs().max()):.2e}  fused {t(fused):6.1f} us"
    for tile in (1, 3, 2):
        line += f"  gemm tile {tile}: {t(lambda: gemm(tile)):6.1f}"
    tl = t(lambda: layer1(out_f))
    line += f" | in flow with layer 1 ({tl:.0f} us): fused {t(lambda: (fused(), layer1(out_f))) - tl:6.1f}  gemm tile 1 {t(lambda: (gemm(1), layer1(out_g))) - tl:6.1f}  tile 3 {t(lambda: (gemm(3), layer1(out_g))) - tl:6.1f}"
    print(line, flush=True)
