"""The tree kernel in the tail of a run to completion (configs[1], trained weights): per-tree validate / walk times, path length,
first changed level and line-following rounds from rc_mcts_t::select_stats, sampled whenever the listed trees fall to <= 64.
    python tools/tail_tree_stats.py [bf16|f32s]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

dt = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[sys.argv[1] if len(sys.argv) > 1 else "f32s"]
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=dt)
agent.prepare(1024, 175000)
run = agent.start_batch(cubes, None, 175000)
shown = 0
while not run.done:
    run.round()
    f = run.forest
    if f.G <= 64 and run.it % 160 < 16 and shown < 12:
        torch.cuda.synchronize()
        st = f.select_stats.cpu().numpy().astype(np.int64)
        live = (f.status == 0).cpu().numpy() & (st[:, 1] > 2)
        s = st[live]
        if not len(s):
            continue
        shown += 1
        tot = s[:, 2] + s[:, 3]
        w = int(np.argmax(tot))
        pc = lambda a: [round(float(np.percentile(a, q)) / 100, 1) for q in (50, 90, 100)]   # noqa: E731
        print(f"it {run.it} G {f.G} running {live.sum()} plen p50/90/max {[int(np.percentile(s[:, 1], q)) for q in (50, 90, 100)]} first-changed p50 {int(np.median(s[:, 0]))} | "
              f"us p50/90/max: validate {pc(s[:, 2])} walk {pc(s[:, 3])} total {pc(tot)} | f64 levels p50 {int(np.median(s[:, 5]))} revisits p50 {int(np.median(s[:, 6]))} "
              f"line rounds p50 {int(np.median(s[:, 7] >> 16))} line levels p50 {int(np.median(s[:, 7] & 0xFFFF))}")
        print(f"   slowest: plen {s[w, 1]} first {s[w, 0]} validate {s[w, 2] / 100:.1f} walk {s[w, 3] / 100:.1f} us, f64 {s[w, 5]} revisits {s[w, 6]} line rounds {s[w, 7] >> 16} levels {s[w, 7] & 0xFFFF}", flush=True)
res = run.finish()
print("solved", float(res.solved.mean()), "seconds", res.seconds)
