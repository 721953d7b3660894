"""Steady-state pool: consecutive timed windows of K steps (sync on both sides), with the refills that fell into each."""
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

dt = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(8192, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=dt)
run = agent.start_batch(cubes, None, 50000, slots=1024)
while run.next_game < 2048:
    run.round()
for w in range(12):
    torch.cuda.synchronize()
    r0, n0 = run.stats["refills"], run.nodes_now()
    torch.cuda.synchronize()
    t = time.perf_counter()
    left = K
    while left > 0:
        b = run.it
        run.round(left)
        left -= run.it - b
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t
    n1 = run.nodes_now()
    print(f"window {w}: {dtm / K * 1e3:.4f} ms/step, {(n1 - n0) / dtm / 1e6:.2f} M/s, refills {run.stats['refills'] - r0}", flush=True)
