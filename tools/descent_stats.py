"""Diagnostics: distribution of sequential-tail lengths of the MCTS descents (trained weights, 1 024 depth-20 trees)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model  # noqa: E402
from librubiks.solving.mcts_device import MCTSForest  # noqa: E402

np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
f = MCTSForest(1024, 12 * 400 + 64)
f.set_net(InferenceNet(model, torch.bfloat16))
f.reset(cubes)
prev_paths = {}
hist = []
for it in range(330):
    f.step(0.6, f.C, use_graph=False)
    if it >= 200 and it % 10 == 0:
        st = f.select_stats.cpu().numpy()
        run = (f.status == 0).cpu().numpy()
        first, plen = st[run, 0], st[run, 1]
        tail = plen - 1 - first
        hist.append((it, int(run.sum()), float(plen.mean()), int(plen.max()), float(tail.mean()), int(np.percentile(tail, 90)),
                     int(np.percentile(tail, 99)), int(tail.max()), float((tail > 32).mean())))
for h in hist:
    print("it %d running %d | plen mean %.1f max %d | tail mean %.1f p90 %d p99 %d max %d | frac tail>32: %.3f" % h)

# ---- would a cache of the last K paths cover the sequential tails? (iterated matching) -----------------------
print("path-cache hypothesis", flush=True)
KMAX = 8
trees = np.flatnonzero((f.status == 0).cpu().numpy())[:128]
history = {int(t): [] for t in trees}
stats = {k: {"tails": 0, "levels": 0, "uncovered": 0, "rounds": [], "worst_unc": []} for k in (1, 2, 4, 8)}
for it in range(60):
    f.step(0.6, f.C, use_graph=False)
    plen = f.path_len.cpu().numpy()
    pn = f.path_node.cpu().numpy()
    st = f.select_stats.cpu().numpy()
    status = f.status.cpu().numpy()
    worst = {k: 0 for k in stats}
    for t in trees:
        t = int(t)
        if status[t] != 0:
            continue
        path = pn[t, :plen[t]].copy()
        first = int(st[t, 0])
        tail = len(path) - 1 - first
        if tail > 0 and len(history[t]) >= 1:
            for k, sk in stats.items():
                # walk the tail: at each uncovered level try to continue along one of the last k paths
                j, rounds, unc = first + 1, 0, 0
                while j < len(path):
                    best = 0
                    for old in history[t][-k:]:
                        n, i = min(len(old), len(path)), j
                        while i < n and old[i] == path[i]:
                            i += 1
                        best = max(best, i - j)
                    if best == 0:
                        unc += 1      # this level has to be walked sequentially
                        j += 1
                    else:
                        rounds += 1
                        j += best
                sk["tails"] += 1
                sk["levels"] += tail
                sk["uncovered"] += unc
                sk["rounds"].append(rounds)
                worst[k] = max(worst[k], unc + rounds)
        history[t].append(path)
        history[t] = history[t][-KMAX:]
    for k in stats:
        stats[k]["worst_unc"].append(worst[k])
for k, sk in stats.items():
    print(f"last {k} paths: sequential levels left {sk['uncovered'] / max(sk['levels'], 1):.3f} of tail levels, "
          f"mean rounds {np.mean(sk['rounds']):.2f}, per-step WORST tree (sequential levels + rounds): mean {np.mean(sk['worst_unc']):.1f} "
          f"max {np.max(sk['worst_unc'])}")
# ---- why do descents become sequential?  action change vs repeated node ---------------------------------------
f.step(0.6, f.C, use_graph=False)
plen_prev = f.path_len.cpu().numpy().copy()
pn_prev = f.path_node.cpu().numpy().copy()
pa_prev = f.path_act.cpu().numpy().copy()
f.step(0.6, f.C, use_graph=False)
plen = f.path_len.cpu().numpy(); pn = f.path_node.cpu().numpy(); pa = f.path_act.cpu().numpy()
st = f.select_stats.cpu().numpy(); status = f.status.cpu().numpy()
reasons = {"repeat": 0, "action": 0, "old_leaf": 0}
rep_level = []
for t in np.flatnonzero(status == 0):
    old = pn_prev[t, :plen_prev[t]]
    first = st[t, 0]
    seen, fr = set(), len(old)
    for k, node in enumerate(old):
        if node in seen:
            fr = k
            break
        seen.add(node)
    rep_level.append(fr / len(old))
    if first == len(old) - 1:
        reasons["old_leaf"] += 1
    elif fr == first:
        reasons["repeat"] += 1
    else:
        reasons["action"] += 1
print("reason for first sequential level:", reasons, " mean (first repeat level / path length):", float(np.mean(rep_level)))
