"""
The reference's evaluation protocol at its CLI defaults (runeval.py:32-73: 500 games per depth, max_states 175 000,
MCTS c = 0.6 with graph search / AStar lambda 0.2 N 100) on one MI355X, all depths pooled on 1 024 concurrent trees.
Writes the reference's result files (<out>/evaluation_results/*.npy, eval_settings.json) and a summary JSON.

    python tools/eval_default.py --out gpurun_out/eval_default
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
    ap.add_argument("--out", default="gpurun_out/eval_default")
    ap.add_argument("--games", type=int, default=500)
    ap.add_argument("--depths", default="10,15,20,25,30")
    ap.add_argument("--max-states", type=int, default=175_000)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--agents", default="mcts,astar")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32s"], help="network engine: bf16 (fast) or f32s (fp32 accuracy, the reference's precision)")
    args = ap.parse_args()
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS, AStar
    from librubiks.solving.evaluation import Evaluator
    from librubiks.utils import set_seeds
    model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
    depths = [int(d) for d in args.depths.split(",")]
    nd = F32_SPLIT if args.dtype == "f32s" else torch.bfloat16
    summary = {"games_per_depth": args.games, "depths": depths, "max_states": args.max_states, "slots": args.slots,
               "engine": args.dtype}
    for name in args.agents.split(","):
        set_seeds()
        if name == "mcts":
            agent, ev = MCTS(model, c=0.6, search_graph=True, net_dtype=nd), Evaluator(args.games, depths, None, args.max_states, slots=args.slots)
        else:
            agent, ev = AStar(model, lambda_=0.2, expansions=100, net_dtype=nd), Evaluator(args.games, depths, None, args.max_states)
        torch.cuda.synchronize()
        t = time.perf_counter()
        res, states, times = ev.eval(agent)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        ev.save(args.out, str(agent), res, states, times)
        per_depth = [ev.log_this_depth(res[i], states[i], times[i], d) for i, d in enumerate(depths)]
        summary[name] = {"agent": str(agent), "seconds": dt, "states": int(states.sum()), "states_per_sec": float(states.sum() / dt),
                         "per_depth": [{k: p[k] for k in ("depth", "share_completed", "ci95", "mean_turns", "states_per_game", "time_per_game", "states_per_sec")} for p in per_depth],
                         "note": "time_per_game / states_per_sec per depth are the reference's quantities (each game's own wall interval; the games of a batch "
                                 "overlap on the GPU), `states_per_sec` at this level is the throughput of the evaluation (all states / wall seconds)"}
        print(name, json.dumps(summary[name]), flush=True)
        del agent
        torch.cuda.empty_cache()
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
