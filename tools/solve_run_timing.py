"""Where does the time of a full 1 024-tree solve run go? (search loop vs post-processing, level budget 0 vs 32)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube
from librubiks.model import Model
from librubiks.solving.agents import MCTS

np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
for budget in (0, 32):
    agent = MCTS(model, c=0.6, search_graph=True, level_budget=budget)
    orig = agent._collect
    times = {}
    def timed_collect(forest, seconds, orig=orig, times=times):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = orig(forest, seconds)
        torch.cuda.synchronize(); times["collect"] = time.perf_counter() - t
        return r
    agent._collect = timed_collect
    for rep in range(2):
        t0 = time.perf_counter()
        res = agent.search_batch(cubes, None, 50000)
        total = time.perf_counter() - t0
        print(f"budget {budget} rep {rep}: total {total:.2f}s search {res.seconds:.2f}s collect {times['collect']:.2f}s "
              f"solved {res.solved.mean():.4f} iterations max {res.iterations.max()} nodes {res.nodes.sum()}", flush=True)
