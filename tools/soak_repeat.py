"""The same search again and again in one process: on one agent (forest, engine and HIP graphs reused) and on fresh agents (node-store
ranges parked and taken over, or retired when the shape changes in between).  Every repetition must return the first one's results,
game for game -- nodes, solution lengths, action queues.
    python tools/soak_repeat.py [repeats] [trees] [max_states]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving.agents import MCTS, AStar  # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 12
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 175000
net = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
np.random.seed(0)
states = np.array([cube.scramble(20, True)[0] for _ in range(trees)])
other = np.array([cube.scramble(14, True)[0] for _ in range(97)])


def same(a, b):
    return (np.array_equal(a.solved, b.solved) and np.array_equal(a.nodes, b.nodes) and np.array_equal(a.lengths, b.lengths)
            and all(list(x) == list(y) for x, y in zip(a.queues, b.queues)))


bad = 0
for dt, name in ((F32_SPLIT, "f32s"), (torch.bfloat16, "bf16")):
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=dt)
    first = agent.search_batch(states, None, cap)
    print(f"{name}: first run {first.seconds:.2f} s, solved {first.solved.mean():.3f}, nodes {int(first.nodes.sum())}", flush=True)
    for r in range(repeats):
        t0 = time.perf_counter()
        if r % 3 == 0:
            res, how = agent.search_batch(states, None, cap), "same agent"
        elif r % 3 == 1:
            res, how = MCTS(net, c=0.6, search_graph=True, net_dtype=dt).search_batch(states, None, cap), "fresh agent"
        else:   # a forest of another shape and an A* batch in between, then a fresh agent
            MCTS(net, c=0.6, search_graph=True, net_dtype=dt).search_batch(other, None, 30000, slots=40)
            AStar(net, lambda_=0.2, expansions=64, net_dtype=dt).search_batch(other, None, 20000)
            res, how = MCTS(net, c=0.6, search_graph=True, net_dtype=dt).search_batch(states, None, cap), "fresh agent after other shapes"
        ok = same(first, res)
        bad += not ok
        print(f"{name} repeat {r} ({how}): {'same' if ok else 'DIFFERENT'}; {time.perf_counter() - t0:.2f} s; HBM in use {(torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9:.0f} GB", flush=True)
print("different repetitions:", bad)
sys.exit(bad)
