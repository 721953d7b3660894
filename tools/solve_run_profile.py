"""
Trajectory of a run to completion (config #2 with trained weights): running trees and wall time against the
iteration count, plus the step time of small forests (the straggler regime).

    python tools/solve_run_profile.py --out gpurun_out/solve_run_profile.json
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
    ap.add_argument("--out", default="gpurun_out/solve_run_profile.json")
    ap.add_argument("--trees", type=int, default=1024)
    ap.add_argument("--max-states", type=int, default=50000)
    ap.add_argument("--no-sizes", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32s"])
    args = ap.parse_args()
    from librubiks import cube
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS

    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(args.trees, 20, True)
    model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
    agent = MCTS(model, c=0.6, search_graph=True, net_dtype=F32_SPLIT if args.dtype == "f32s" else torch.bfloat16)
    out = {}
    # (1) trajectory: patch forest.step to log at the agent's own sync points
    agent.search_batch(cubes, None, 2000)   # warm-up
    log = []
    orig_step = md.MCTSForest.step
    state = {"it": 0, "t0": None}

    def step(self, *a, **k):
        r = orig_step(self, *a, **k)
        state["it"] += 1
        if state["it"] % agent.sync_every == 0:
            torch.cuda.synchronize()
            log.append((state["it"], self.G, int((self.status == md.RUNNING).sum().item()), time.perf_counter() - state["t0"]))
        return r

    md.MCTSForest.step = step
    state["t0"] = time.perf_counter()
    res = agent.search_batch(cubes, None, args.max_states)
    total = time.perf_counter() - state["t0"]
    md.MCTSForest.step = orig_step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res2 = agent.search_batch(cubes, None, args.max_states)   # the same search again: every launch size's graph is in the forest's cache
    torch.cuda.synchronize()
    again = time.perf_counter() - t0
    out["run"] = {"seconds_total": total, "seconds_search": res.seconds, "solved": float(res.solved.mean()),
                  "nodes": int(res.nodes.sum()), "iterations_max": int(res.iterations.max()),
                  "second_run_seconds": again, "second_run_nodes_per_sec": float(res2.nodes.sum()) / again,
                  "second_run_same_results": bool(np.array_equal(res.nodes, res2.nodes) and np.array_equal(res.lengths, res2.lengths)),
                  "stats": {k: v for k, v in agent.refill_stats.items()},
                  "trajectory_it_B_running_t": log[:: max(1, len(log) // 60)] + [log[-1]]}
    print(json.dumps(out["run"]), flush=True)
    # (2) step time of small forests
    sizes = {}
    for B in (() if args.no_sizes else (1, 4, 16, 64, 256, 1024)):
        a = MCTS(model, c=0.6, search_graph=True)
        sub = cube.scramble_batch(B, 20, True)[0]
        a.search_batch(sub, None, 4000, max_iterations=60)
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = a.search_batch(sub, None, 4000, max_iterations=260)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        sizes[B] = {"ms_per_iteration": dt / 260 * 1e3, "nodes_per_sec": float(r.nodes.sum() / dt)}
        print(B, sizes[B], flush=True)
    out["step_time_by_forest_size"] = sizes
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
