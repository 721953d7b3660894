"""Does running the network in bf16 change what the search achieves?  Same 512 depth-20 scrambles, MCTS and A*, bf16 vs fp32."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import Model  # noqa: E402
from librubiks.solving.agents import MCTS, AStar  # noqa: E402

np.random.seed(0)
cubes, _, _ = cube.scramble_batch(512, 20, True)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
out = {}
for name, make in (("MCTS c=0.6", lambda dt: MCTS(model, c=0.6, search_graph=True, net_dtype=dt)),
                   ("AStar lambda=0.2 N=100", lambda dt: AStar(model, lambda_=0.2, expansions=100, net_dtype=dt))):
    row = {}
    for dt in (torch.bfloat16, torch.float32):
        agent = make(dt)
        res = agent.search_batch(cubes, None, 20000)
        row[str(dt).split(".")[-1]] = {"solve_rate": float(res.solved.mean()), "mean_length": float(res.lengths[res.solved].mean()),
                                      "mean_nodes": float(res.nodes.mean()), "seconds": res.seconds}
        del agent
        torch.cuda.empty_cache()
    both = None
    out[name] = row
    print(name, json.dumps(row), flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bf16_vs_fp32.json", "w"), indent=1)
