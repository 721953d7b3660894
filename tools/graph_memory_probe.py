"""HBM held by the captured graph ladder (MCTS.prepare) next to the forest, and the time it takes.   python tools/graph_memory_probe.py"""
import os, sys
import numpy as np, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks.model import Model, F32_SPLIT
from librubiks.solving.agents import MCTS
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
for trees, cap, dt in ((1024, 175000, F32_SPLIT), (8192, 50000, F32_SPLIT), (8192, 50000, torch.bfloat16)):
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    agent = MCTS(model, c=0.6, search_graph=True, net_dtype=dt)
    import time
    t0 = time.perf_counter()
    forest = agent._forest_for(trees, cap)
    torch.cuda.synchronize()
    a1 = torch.cuda.memory_allocated()
    agent.prepare(trees, cap)
    torch.cuda.synchronize()
    a2, r2 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    print(f"{trees} trees cap {cap} {dt}: forest {(a1 - base) / 2**30:.1f} GiB, after capturing {len(forest.rungs)} launch sizes +{(a2 - a1) / 2**30:.2f} GiB allocated, reserved {r2 / 2**30:.1f} GiB, prepare {time.perf_counter() - t0:.1f} s", flush=True)
    del agent, forest
    import gc; gc.collect()
