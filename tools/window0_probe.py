"""Where the first timed window after the prep spends its time: GPU time per step (events around every forest.step) and host gaps."""
import os
import sys
import time
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
while run.next_game < 2048 + 128:
    run.round()
f = run.forest
orig = f.step
evs, host = [], []


def step(*a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    host.append(time.perf_counter())
    return orig(*a, **k)


f.step = step
for w in range(3):
    torch.cuda.synchronize()
    n0 = run.nodes_now() if len(sys.argv) > 2 else 0
    torch.cuda.synchronize()
    evs.clear(); host.clear()
    t = time.perf_counter()
    left = 20
    while left > 0:
        b = run.it
        run.round(left)
        left -= run.it - b
    e_end = torch.cuda.Event(enable_timing=True)
    e_end.record()
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t
    gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)] + [evs[-1].elapsed_time(e_end)]
    print(f"window {w}: wall {dtm * 1e3:.2f} ms; first launch after {1e3 * (host[0] - t):.3f} ms; host enqueue span {1e3 * (host[-1] - host[0]):.2f} ms")
    print("   gpu ms per step:", " ".join(f"{g:.2f}" for g in gpu), "| sum", f"{sum(gpu):.2f}", flush=True)
