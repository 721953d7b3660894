"""Per-step GPU completion times right after the steady-state prep (events on the step stream, no syncs inside)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

dt = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(8192, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=dt)
run = agent.start_batch(cubes, None, 50000, slots=1024)
while run.next_game < 2048:
    run.round()
torch.cuda.synchronize()
N = 80
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
its, refills = [], []
ev[0].record()
for i in range(N):
    b = run.it
    run.round(1)
    its.append(run.it - b)
    refills.append(run.stats["refills"])
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
for i in range(0, N, 10):
    print(" ".join(f"{ms[j]:.2f}{'*' if j and refills[j] != refills[j - 1] else ''}" for j in range(i, min(i + 10, N))), "| its", sum(its[i:i + 10]))
