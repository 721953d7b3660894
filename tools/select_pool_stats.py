"""Where k_mcts_select spends its time in the steady-state pool (needs the stamp build of rubiks_mcts.hip)."""
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
while run.next_game < 3072:
    run.round()
f = run.forest
for rep in range(6):
    for _ in range(10):
        run.round()
    torch.cuda.synchronize()
    st = f.select_stats.cpu().numpy().astype(np.int64)
    live = (f.status == 0).cpu().numpy() & (st[:, 1] > 2)
    s = st[live]
    tot = s[:, 2] + s[:, 3]
    w = int(np.argmax(tot))
    pc = lambda a: [round(float(np.percentile(a, q)) / 100, 1) for q in (50, 90, 99, 100)]
    print(f"trees {live.sum()} plen p50/90/99/max {[int(np.percentile(s[:, 1], q)) for q in (50, 90, 99, 100)]} | us p50/90/99/max: "
          f"backup+stage {pc(s[:, 5])} passA {pc(s[:, 6] - s[:, 5])} passB {pc(s[:, 7] - s[:, 6])} prefix-L {pc(s[:, 2] - s[:, 7])} walk {pc(s[:, 3])} total {pc(tot)}")
    print(f"   slowest raw: start {s[w, 0]} plen {s[w, 1]} validate {s[w, 2] / 100:.1f} us walk {s[w, 3] / 100:.1f} us cycles {s[w, 4]} f5 {s[w, 5]} f6 {s[w, 6]} f7 {s[w, 7] >> 16}/{s[w, 7] & 0xFFFF}  (unstamped build: f5 = float64 fallbacks, f6 = revisited levels, f7 = line rounds / levels)")
    print(f"   slowest: plen {s[w, 1]} first {s[w, 0]} backup+stage {s[w, 5] / 100:.1f} passA {(s[w, 6] - s[w, 5]) / 100:.1f} passB {(s[w, 7] - s[w, 6]) / 100:.1f} "
          f"prefix-L {(s[w, 2] - s[w, 7]) / 100:.1f} walk {s[w, 3] / 100:.1f} ({s[w, 1] - 1 - s[w, 0]} levels, {s[w, 4] / max(s[w, 1] - 1 - s[w, 0], 1):.0f} cyc/level)", flush=True)
