"""
Would a cache of the last K descent paths turn the long sequential tails of deep trees into parallel
re-validations?  Deep into a run to completion (trained weights), record the descent paths of the running trees
for a number of consecutive iterations and replay the splice rule on the host: at a level where the walk has to
start, continue along any of the last K paths that holds the same node AT THE SAME LEVEL for as long as it agrees
with the actual new path (a parallel validation round); levels no cached path covers are walked one by one.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model  # noqa: E402
from librubiks.solving.mcts_device import MCTSForest  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(B, 20, True)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
f = MCTSForest(B, 50000)
f.set_net(InferenceNet(model, torch.bfloat16))
f.reset(cubes)
KS = (1, 2, 4, 8)
for start_it in (600, 1500, 2500):
    while int(f.iterations.max().item()) < start_it:
        f.step(0.6, 50000, use_graph=False)
    run = np.flatnonzero((f.status == 0).cpu().numpy())
    if len(run) == 0:
        break
    trees = run[:24]
    history = {int(t): [] for t in trees}
    agg = {k: {"tail": 0, "seq": 0, "rounds": 0, "descents": 0, "worst": []} for k in KS}
    for it in range(80):
        f.step(0.6, 50000, use_graph=False)
        plen = f.path_len.cpu().numpy()
        pn = f.path_node.cpu().numpy()
        st = f.select_stats.cpu().numpy()
        status = f.status.cpu().numpy()
        worst = {k: 0 for k in KS}
        for t in trees:
            t = int(t)
            if status[t] != 0:
                continue
            path = pn[t, :plen[t]].copy()
            first = int(st[t, 0])
            if it >= 8:
                for k in KS:
                    j, rounds, seq = first + 1, 0, 0
                    while j < len(path):
                        best = 0
                        for old in history[t][-k:]:
                            n, i = min(len(old), len(path)), j
                            while i < n and old[i] == path[i]:
                                i += 1
                            best = max(best, i - j)
                        if best == 0:
                            seq += 1
                            j += 1
                        else:
                            rounds += 1
                            j += best
                    a = agg[k]
                    a["tail"] += len(path) - 1 - first
                    a["seq"] += seq
                    a["rounds"] += rounds
                    a["descents"] += 1
                    worst[k] = max(worst[k], seq + 4 * rounds)
            history[t].append(path)
            history[t] = history[t][-8:]
        if it >= 8:
            for k in KS:
                agg[k]["worst"].append(worst[k])
    print(f"--- from iteration {start_it}: {len(trees)} trees, mean path {np.mean([len(h[-1]) for h in history.values() if h]):.0f}")
    for k in KS:
        a = agg[k]
        if a["descents"]:
            print(f"last {k} paths: tail levels {a['tail'] / a['descents']:.1f} per descent -> sequential {a['seq'] / a['descents']:.1f} "
                  f"+ {a['rounds'] / a['descents']:.2f} parallel rounds; worst tree per step (seq + 4 x rounds): mean {np.mean(a['worst']):.0f}", flush=True)
