"""Input layer of the split engine: fused MFMA kernel vs split one-hot + library GEMM + activation kernel (11 264 rows)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import Model, SplitF32Net  # noqa: E402


def ms(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


np.random.seed(0)
cubes, _, _ = cube.scramble_batch(11264, 20, True)
eng = SplitF32Net(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval())
L = eng.layers
a_f = eng._first_from_cubes(cubes, L)
_, B, b, code, alpha = L[0][:5]


def gemm_form():
    part = torch.empty((1, cubes.n, B.shape[0]), dtype=torch.float32, device="cuda")
    part[0] = torch.mm(eng._input_from_cubes(cubes), B.t(), out_dtype=torch.float32)
    return eng._act(part, 0, b, code, alpha, True)


a_g = gemm_form()
H = a_f.shape[1] // 2
y_f = a_f[:, :H].double() + a_f[:, H:].double() / 2048
y_g = a_g[:, :H].double() + a_g[:, H:].double() / 2048
print("max |fused - gemm| =", float((y_f - y_g).abs().max()), "max |y| =", float(y_g.abs().max()))
print("fused kernel ms", ms(lambda: eng._first_from_cubes(cubes, L)))
if "--quick" not in sys.argv:
    print("gemm form   ms", ms(gemm_form))
    print("whole net fused ms", ms(lambda: eng.head_cubes(cubes)))
    eng.fused_input = False
    print("whole net gemm  ms", ms(lambda: eng.head_cubes(cubes)))
