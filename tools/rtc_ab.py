"""configs[1] to completion, several times on one box (same process set-up each time): python tools/rtc_ab.py f32s|bf16 [reps]"""
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

dt = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[sys.argv[1] if len(sys.argv) > 1 else "f32s"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=dt)
agent.prepare(1024, 175000)
agent.search_batch(cubes, None, 175000, max_iterations=30)
out = []
for _ in range(reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = agent.search_batch(cubes, None, 175000)
    torch.cuda.synchronize()
    out.append(time.perf_counter() - t)
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'f32s'} RUBIKS_STEP_THREADS={os.environ.get('RUBIKS_STEP_THREADS', 'auto')}: seconds {[round(x, 3) for x in out]}, "
      f"{res.nodes.sum() / min(out) / 1e6:.2f} M nodes/s best, iterations {int(res.iterations.max())}, solved {res.solved.mean():.3f}")
