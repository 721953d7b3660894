"""Descent statistics deep into a run to completion (trained weights): path length and sequential tail by iteration."""
import os
import sys
import time

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
t_sel = 0.0
for it in range(1, 4400):
    f.step(0.6, 50000, use_graph=False)
    if it % 200 == 0:
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            f.step(0.6, 50000, use_graph=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 20
        st = f.select_stats.cpu().numpy()
        run = (f.status == 0).cpu().numpy()
        if not run.any():
            break
        first, plen = st[run, 0], st[run, 1]
        tail = plen - 1 - first
        w = int(np.argmax(st[run, 2] + st[run, 3]))   # the slowest tree of the step
        print(f"it {it} running {int(run.sum())} | plen mean {plen.mean():.0f} max {plen.max()} | tail mean {tail.mean():.1f} "
              f"p90 {np.percentile(tail, 90):.0f} max {tail.max()} | step {dt * 1e3:.3f} ms | slowest tree: plen {plen[w]} "
              f"validate {st[run, 2][w] / 100:.0f} us, walk {st[run, 3][w] / 100:.0f} us for {tail[w]} levels, "
              f"{st[run, 4][w] / max(tail[w], 1):.0f} cycles/level at {st[run, 4][w] / max(st[run, 3][w], 1) / 10:.2f} GHz, "
              f"{st[run, 5][w]} float64 fallbacks, {st[run, 6][w]} revisited levels, "
              f"{st[run, 7][w] >> 16} line rounds appended {st[run, 7][w] & 0xFFFF} levels", flush=True)
