"""k_mcts_select in the steady-state pool: per-tree time before the walk (staging + re-validation) and of the walk, from the kernel's own
10-ns stamps (select_stats[2], [3]): python tools/select_split_stats.py [bf16|f32s]"""
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
run = agent.start_batch(cubes, None, 175000, slots=1024)
while run.next_game < 3072:
    run.round()
f = run.forest
pc = lambda a: [round(float(np.percentile(a, q)) / 100, 1) for q in (50, 90, 99, 100)]   # noqa: E731
for rep in range(5):
    for _ in range(7):
        run.round()
    torch.cuda.synchronize()
    st = f.select_stats.cpu().numpy().astype(np.int64)
    live = (f.status == 0).cpu().numpy() & (st[:, 1] > 2)
    s = st[live]
    tot = s[:, 2] + s[:, 3]
    w = int(np.argmax(tot))
    print(f"trees {live.sum()} plen p50/90/99/max {[int(np.percentile(s[:, 1], q)) for q in (50, 90, 99, 100)]} | us p50/90/99/max: before the walk {pc(s[:, 2])} "
          f"walk {pc(s[:, 3])} both {pc(tot)} | slowest tree: plen {s[w, 1]} first changed level {s[w, 0]} before {s[w, 2] / 100:.1f} walk {s[w, 3] / 100:.1f} "
          f"({s[w, 7] >> 16} line rounds, {s[w, 7] & 0xFFFF} levels)", flush=True)
