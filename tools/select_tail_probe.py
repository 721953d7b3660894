"""
What the tree kernel does per step in the TAIL of BASELINE configs[1] searched to completion (tens of trees left): per running
tree and step, from rc_mcts_t::select_stats -- the level the descent became sequential at, the new path length, 10 ns ticks of
staging + re-validation and of the sequential walk, levels walked without a walk record.

    python tools/select_tail_probe.py [launch size to start sampling at, default 64] [rounds, default 40]

With a diagnostic build of the library (tools/build_ab_lib.sh WORK phases -DRUBIKS_SELECT_PHASES, RUBIKS_HIP_LIB=.../ab/phases.so) the
last three slots hold the ticks of the kernel's parallel phases instead: children's backup + staging, pass A, pass B
(-DRUBIKS_SELECT_PHASES=2: pass A's first levels, number of later levels; =3: walk loop, write-back, ring line; =4: shader cycles inside
line segments, segments, passes of the walk loop -- csrc/rubiks_mcts.hip; the three 'ticks:' lines keep the labels of the first mode).
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

G0, ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 40
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=F32_SPLIT)
agent.prepare(1024, 175000)
run = agent.start_batch(cubes, None, 175000)
while not run.done and run.forest.G > G0:
    run.round()
rows = []
PHASES = "phases" in os.environ.get("RUBIKS_HIP_LIB", "")
for _ in range(ROUNDS):
    if run.done:
        break
    run.round()
    torch.cuda.synchronize()
    f = run.forest
    st, stats = f.status.cpu().numpy(), f.select_stats.cpu().numpy()
    live = np.flatnonzero((run.owner >= 0) & (st == md.RUNNING))
    for t in live:
        s = stats[t]
        rows.append((f.G, len(live), s[0], s[1], s[2], s[3], s[4], s[5] & 0xFFFF, s[6], s[7] >> 16, s[7] & 0xFFFF) if not PHASES else
                    (f.G, len(live), s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], 0))
a = np.array(rows, dtype=np.int64)
print(f"samples {len(a)} (tree, step) pairs at launch sizes {sorted(set(a[:, 0].tolist()))}, running trees {a[:, 1].min()}..{a[:, 1].max()}")
names = ["first sequential level", "new path length", "staging + re-validation (10 ns ticks)", "sequential walk (ticks)", "walk shader cycles",
         "levels without a walk record", "revisits", "line rounds", "line levels"]
PHASES = "phases" in os.environ.get("RUBIKS_HIP_LIB", "")   # a -DRUBIKS_SELECT_PHASES build: the last three slots are phase ticks instead
if PHASES:
    names[5:] = ["ticks: children's backup + staging + chains", "ticks: pass A (+ late levels)", "ticks: pass B", "-"]
for i, n in enumerate(names):
    c = a[:, 2 + i]
    print(f"{n:40s} mean {c.mean():9.1f}  median {np.median(c):8.0f}  p90 {np.percentile(c, 90):8.0f}  max {c.max():8d}")
seq = a[:, 3] - a[:, 2]
print(f"{'levels walked sequentially':40s} mean {seq.mean():9.1f}  median {np.median(seq):8.0f}  p90 {np.percentile(seq, 90):8.0f}  max {seq.max():8d}")
# the step waits for its slowest tree: per sample group (one step's trees) the maximum
print("slowest tree of a step: staging+reval ticks", int(np.mean([a[a[:, 1] == k][:, 4].max() for k in set(a[:, 1])])),
      " walk ticks", int(np.mean([a[a[:, 1] == k][:, 5].max() for k in set(a[:, 1])])))
while not run.done:
    run.round()
res = run.finish()
print("solved", float(res.solved.mean()), "nodes", int(res.nodes.sum()))
