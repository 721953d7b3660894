"""
The input layer of the split network both ways, same box: rc_first_layer_gather_f16 (sum of 20 fp32 rows of W^T per state) against
rc_first_layer_split_flag_f16 (one-hot MFMA, hi / lo tables), microseconds per launch as a replayed graph of `reps` launches.

    python tools/input_layer_ab.py [rows ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip, cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model, ModelConfig, make_inference_net  # noqa: E402

rows_list = [int(a) for a in sys.argv[1:]] or [352, 1408, 5632, 11264, 196608]
torch.manual_seed(0)
np.random.seed(0)
eng = make_inference_net(Model.create(ModelConfig(architecture="fc_small")).eval().cuda(), F32_SPLIT)


def graph_us(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(5):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / (5 * reps)


for rows in rows_list:
    cubes, _, _ = cube.scramble_batch(rows, 30, True)
    out = {}
    for name, flag in (("gather", True), ("mfma", False)):
        eng.gather_input = flag
        out[name] = graph_us(lambda: eng._first_from_cubes(cubes, eng.layers))
    eng.gather_input = True
    print(f"rows {rows:7d}: gather {out['gather']:8.1f} us   one-hot MFMA {out['mfma']:8.1f} us", flush=True)
