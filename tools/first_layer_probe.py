"""
The split engine's input layer alone (rc_first_layer_split_f16: one-hot MFMA from the cube codes + bias + ELU + re-split):
time per call at the row counts of a narrowing forest, and the output's checksum (same-box A/B of two builds with RUBIKS_HIP_LIB:
the two must print identical checksums).
    python tools/first_layer_probe.py [rows ...]
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model, ModelConfig, SplitF32Net  # noqa: E402

bf16 = "--bf16" in sys.argv          # the bf16 engine's input layer (rc_first_layer_mfma_bf16) instead
rows_list = [int(a) for a in sys.argv[1:] if a != "--bf16"] or [352, 1056, 2816, 5632, 11264, 196608]
wdir = os.path.join(ROOT, "weights", "fc_small_r1")
torch.manual_seed(0)
net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
eng = InferenceNet(net, torch.bfloat16) if "--bf16" in sys.argv else SplitF32Net(net)
first = (lambda c: eng.first_layer(c)) if "--bf16" in sys.argv else (lambda c: eng._first_from_cubes(c, eng.layers))
np.random.seed(1)
print("library:", os.environ.get("RUBIKS_HIP_LIB", "in-tree"))
for rows in rows_list:
    cubes, _, _ = cube.scramble_batch(rows, 25, True)
    out = first(cubes)
    torch.cuda.synchronize()
    crc = zlib.crc32(out.view(torch.int16).cpu().numpy().tobytes())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        first(cubes)
    reps = 50
    a.record()
    for _ in range(reps):
        first(cubes)
    b.record()
    torch.cuda.synchronize()
    print(f"rows {rows:7d}: {a.elapsed_time(b) / reps * 1e3:8.1f} us per call, crc32 of the [hi | lo] output {crc:08x}, overflow flag {int(eng.range_flag.item()) if hasattr(eng, 'range_flag') else '-'}", flush=True)
