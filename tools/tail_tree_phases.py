"""Phases of the tree kernel in the tail of a run to completion (needs the stamp build: tools/select_stamps_patch.py).   python tools/tail_tree_phases.py [bf16|f32s]"""
import os, sys
import numpy as np, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube
from librubiks.model import F32_SPLIT, Model
from librubiks.solving.agents import MCTS
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
    if (f.G <= 64 or (f.G in (1024, 512, 256) and run.it % 320 < 16)) and run.it % 160 < 16 and shown < 16:
        torch.cuda.synchronize()
        st = f.select_stats.cpu().numpy().astype(np.int64)
        live = (f.status == 0).cpu().numpy() & (st[:, 1] > 2)
        s = st[live]
        if not len(s):
            continue
        shown += 1
        tot = s[:, 2] + s[:, 3] + s[:, 4]
        w = int(np.argmax(tot))
        pc = lambda a: [round(float(np.percentile(a, q)) / 100, 1) for q in (50, 90, 100)]
        print(f"it {run.it} G {f.G} running {live.sum()} plen p50 {int(np.median(s[:, 1]))} | us p50/90/max: top {pc(s[:, 5])} stage {pc(s[:, 6] - s[:, 5])} passA {pc(s[:, 7] - s[:, 6])} "
              f"passB+chains {pc(s[:, 2] - s[:, 7])} walk {pc(s[:, 3])} expand {pc(s[:, 4])} total {pc(tot)}")
        print(f"   slowest: plen {s[w, 1]} first {s[w, 0]} top {s[w, 5] / 100:.1f} stage {(s[w, 6] - s[w, 5]) / 100:.1f} passA {(s[w, 7] - s[w, 6]) / 100:.1f} passB+chains {(s[w, 2] - s[w, 7]) / 100:.1f} "
              f"walk {s[w, 3] / 100:.1f} expand {s[w, 4] / 100:.1f}", flush=True)
res = run.finish()
print("solved", float(res.solved.mean()), "seconds", res.seconds)
