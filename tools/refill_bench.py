"""
Continuous batching vs plain batches on an evaluation-sized job: G depth-20 scrambles (trained weights, MCTS
c = 0.6, graph search, max_states 50 000) on `slots` concurrent trees.

    python tools/refill_bench.py --games 4096 --slots 1024 --out gpurun_out/refill_bench.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--max-states", type=int, default=50000)
    ap.add_argument("--out", default="gpurun_out/refill_bench.json")
    args = ap.parse_args()
    from librubiks import cube
    from librubiks.cube.device import DeviceCubes
    from librubiks.model import Model
    from librubiks.solving.agents import MCTS

    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(args.games, 20, True)
    model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
    agent = MCTS(model, c=0.6, search_graph=True)
    warm = DeviceCubes.empty(args.slots)
    warm.soa[:, :args.slots] = cubes.soa[:, :args.slots]
    agent.search_batch(warm, None, 2000)
    out = {"games": args.games, "slots": args.slots, "max_states": args.max_states}

    torch.cuda.synchronize()
    t = time.perf_counter()
    parts = []
    for lo in range(0, args.games, args.slots):
        n = min(args.slots, args.games - lo)
        b = DeviceCubes.empty(n)
        b.soa[:, :n] = cubes.soa[:, lo:lo + n]
        parts.append(agent.search_batch(b, None, args.max_states))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    solved = np.concatenate([p.solved for p in parts]); nodes = np.concatenate([p.nodes for p in parts])
    lengths = np.concatenate([p.lengths for p in parts])
    out["plain_batches"] = {"seconds": dt, "solved": float(solved.mean()), "nodes": int(nodes.sum()), "nodes_per_sec": float(nodes.sum() / dt)}
    print(json.dumps(out["plain_batches"]), flush=True)

    for budget in (0, 64):
        MCTS.refill_level_budget = budget
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = agent.search_batch(cubes, None, args.max_states, slots=args.slots)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        out[f"continuous_batching_budget{budget}"] = {"seconds": dt, "solved": float(r.solved.mean()), "nodes": int(r.nodes.sum()),
                                      "nodes_per_sec": float(r.nodes.sum() / dt),
                                      "identical_results": bool(np.array_equal(r.solved, solved) and np.array_equal(r.nodes, nodes)
                                                                and np.array_equal(r.lengths, lengths)),
                                      "stats": dict(agent.refill_stats)}
        print(budget, json.dumps(out[f"continuous_batching_budget{budget}"]), flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
