"""
The tree kernel in the steady-state pool (8 192 scrambles on 1 024 slots, 256 threads per tree): per tree and step, from
rc_mcts_t::select_stats -- path length, ticks (10 ns) of the parallel phases and of the walk; the kernel lasts as long as its slowest tree.

    python tools/select_window_probe.py [steps sampled, default 40]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving import mcts_device as md  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(8192, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=F32_SPLIT)
run = agent.start_batch(cubes, None, 175000, slots=1024)
while run.next_game < 2048 + 128:
    run.round()
per_step = []
for _ in range(N):
    run.round(1)
    torch.cuda.synchronize()
    f = run.forest
    st, s = f.status.cpu().numpy(), f.select_stats.cpu().numpy()
    live = (run.owner >= 0) & (st == md.RUNNING)
    a = s[live]
    tot = a[:, 2] + a[:, 3]
    i = int(np.argmax(tot))
    per_step.append((live.sum(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(), tot.mean(), tot.max(), a[i, 1], a[i, 1] - a[i, 0], a[i, 2], a[i, 3],
                     np.percentile(tot, 99), (tot > 0.5 * tot.max()).sum(), a[:, 5].mean(), a[:, 6].mean(), a[:, 7].mean(), a[i, 5], a[i, 6], a[i, 7]))
p = np.array(per_step, dtype=float)
names = ["running trees", "path length (mean)", "parallel phases, ticks (mean)", "walk, ticks (mean)", "both (mean)", "both (slowest tree)",
         "slowest tree: path length", "slowest tree: levels walked", "slowest tree: parallel phases", "slowest tree: walk", "both (99th percentile)",
         "trees above half the slowest"]
if "phases" in os.environ.get("RUBIKS_HIP_LIB", ""):   # a -DRUBIKS_SELECT_PHASES build (tools/build_ab_lib.sh WORK phases -DRUBIKS_SELECT_PHASES)
    names += ["children's backup + staging (mean)", "pass A (mean)", "pass B (mean)", "slowest tree: staging", "slowest tree: pass A", "slowest tree: pass B"]
for j, n in enumerate(names):
    print(f"{n:34s} mean over {N} steps {p[:, j].mean():9.1f}   min {p[:, j].min():9.1f}   max {p[:, j].max():9.1f}")
