"""Cycles per stage of k_first_layer_split (a -DRC_FL_STAMPS build): python tools/first_layer_stamps.py"""
import ctypes
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip, cube  # noqa: E402
from librubiks.model import Model, SplitF32Net  # noqa: E402

np.random.seed(0)
cubes, _, _ = cube.scramble_batch(11264, 20, True)
eng = SplitF32Net(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval())
lib = _hip.lib()
out = (ctypes.c_ulonglong * 8)()
for _ in range(3):
    eng._first_from_cubes(cubes, eng.layers)
torch.cuda.synchronize()
lib.rc_debug_fl_stamps(out, 1)
reps = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng._first_from_cubes(cubes, eng.layers)
e1.record()
torch.cuda.synchronize()
lib.rc_debug_fl_stamps(out, 0)
us = e0.elapsed_time(e1) / reps * 1e3
passes = 11264 / 64 * 64   # wave passes per call: (rows / 64 states) x 64 column tiles
names = ["codes", "matrix stage", "epilogue"]
print(f"{us:.1f} us per call; cycles per wave pass (clock64 ticks, summed over lanes 0 / passes):")
for i, n in enumerate(names):
    print(f"  {n:14s} {out[i] / reps / passes:9.0f}")
print(f"  total          {sum(out[:3]) / reps / passes:9.0f}   ({passes:.0f} wave passes per call; 8 waves per CU, 2 per SIMD)")
