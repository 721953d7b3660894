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
from librubiks.model import Model, ModelConfig, SplitF32Net  # noqa: E402

rows_list = [int(a) for a in sys.argv[1:]] or [352, 1056, 2816, 5632, 11264, 196608]
wdir = os.path.join(ROOT, "weights", "fc_small_r1")
torch.manual_seed(0)
net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
eng = SplitF32Net(net)
np.random.seed(1)
print("library:", os.environ.get("RUBIKS_HIP_LIB", "in-tree"))
for rows in rows_list:
    cubes, _, _ = cube.scramble_batch(rows, 25, True)
    out = eng._first_from_cubes(cubes, eng.layers)
    torch.cuda.synchronize()
    crc = zlib.crc32(out.cpu().numpy().tobytes())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        eng._first_from_cubes(cubes, eng.layers)
    reps = 50
    a.record()
    for _ in range(reps):
        eng._first_from_cubes(cubes, eng.layers)
    b.record()
    torch.cuda.synchronize()
    print(f"rows {rows:7d}: {a.elapsed_time(b) / reps * 1e3:8.1f} us per call, crc32 of the [hi | lo] output {crc:08x}, overflow flag {int(eng.range_flag.item())}", flush=True)
