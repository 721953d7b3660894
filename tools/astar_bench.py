"""BASELINE config #3: 4 096 depth-20 scrambles, batch weighted A* (lambda 0.2, N 100) on one MI355X.  Reports new states/s."""
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
    ap.add_argument("--problems", type=int, default=4096)
    ap.add_argument("--expansions", type=int, default=100)
    ap.add_argument("--lambda_", type=float, default=0.2)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--weights", default=os.path.join(ROOT, "weights", "fc_small_r1"))
    ap.add_argument("--solve-max-states", type=int, default=0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f32s"], help="network engine inside the agent")
    ap.add_argument("--fused-hidden", type=int, default=None, help="1 / 0: force the own bf16 layer kernel on / off (default: the agent's choice)")
    args = ap.parse_args()
    from librubiks import cube
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import AStar
    if args.fused_hidden is not None:
        import librubiks.model as _m
        import librubiks.solving.astar_device as _ad
        _m.InferenceNet.fused_hidden = bool(args.fused_hidden)
        _ad.FUSED_BF16_LAYER = bool(args.fused_hidden)
    np.random.seed(0)
    torch.manual_seed(0)
    cubes, _, _ = cube.scramble_batch(args.problems, args.depth, True)
    model = Model.load(args.weights).eval() if os.path.isdir(args.weights) else Model.create(ModelConfig()).eval()
    from librubiks.model import F32_SPLIT
    agent = AStar(model, args.lambda_, args.expansions, net_dtype={"bf16": torch.bfloat16, "f32": torch.float32, "f32s": F32_SPLIT}[args.dtype])
    cap = 12 * args.expansions * (args.steps + args.warmup + 2) + 16
    batch = agent._batch_for(cubes.n, cap)
    batch.reset(cubes)
    for _ in range(args.warmup):
        batch.iteration(args.lambda_, batch.C)
    torch.cuda.synchronize()
    n0 = int(batch.n_nodes.sum().item())
    t0 = time.perf_counter()
    per_iter = []
    for _ in range(args.steps):
        a = time.perf_counter()
        new = batch.iteration(args.lambda_, batch.C)
        torch.cuda.synchronize()
        per_iter.append((new, time.perf_counter() - a))
    dt = time.perf_counter() - t0
    nodes = int(batch.n_nodes.sum().item()) - n0
    out = {"metric": "A* node expansions/sec (new states added to the problems)", "value": nodes / dt, "unit": "states/s",
           "ms_per_iteration": dt / args.steps * 1e3, "new_states_per_iteration": nodes / args.steps,
           "child_rows_per_iteration": args.problems * args.expansions * 12,
           "config": {"workload": f"{args.problems} depth-{args.depth} scrambles, AStar lambda={args.lambda_} N={args.expansions}, fc_small",
                      "weights": args.weights if os.path.isdir(args.weights) else "random-init", "net": args.dtype},
           "solved_so_far": int((batch.status == 1).sum().item())}
    if args.solve_max_states:
        t = time.perf_counter()
        res = agent.search_batch(cubes, None, args.solve_max_states)
        out["solve_run"] = {"max_states": args.solve_max_states, "solve_rate": float(res.solved.mean()),
                            "mean_length": float(res.lengths[res.solved].mean()) if res.solved.any() else None,
                            "nodes": int(res.nodes.sum()), "seconds": time.perf_counter() - t,
                            "states_per_sec": float(res.nodes.sum() / (time.perf_counter() - t))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
