"""Ablation of the matrix-core input layer: the same launch with activation none / ReLU / ELU (HIP events)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model  # noqa: E402

np.random.seed(0)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11264
cubes, _, _ = cube.scramble_batch(n, 20, True)
eng = InferenceNet(model, torch.bfloat16, first_layer_table="mfma16")
out = torch.empty((n, 4096), dtype=torch.bfloat16, device="cuda")
base = eng._fused_first
res = {}
for rnd in range(6):
    for name, code in (("none", 0), ("relu", 1), ("elu", 2)):
        eng._fused_first = (base[0], base[1], code, base[3], base[4], base[5])
        eng.first_layer(cubes, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.first_layer(cubes, out)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) * 100)
for k, v in res.items():
    v.sort()
    print(f"activation {k}: median {v[len(v) // 2]:.1f} us", flush=True)
