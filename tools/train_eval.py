"""
End-to-end experiment on one MI355X: train fc_small by Autodidactic Iteration with the device-resident
data path (reference settings of configs/main_train.ini, scaled by wall-clock budget), then evaluate
the batched agents on the trained weights.  (The game-by-game cross-check of the fp32 agents against the restated
reference agents on the trained weights is a test: tests/test_evaluation_gpu.py.)

    python tools/train_eval.py --minutes 12 --out gpurun_out/train_eval

Writes <out>/results.json (losses, timings, solve rates by depth and agent) and
<out>/model/{model.pt,config.json} in the reference's checkpoint layout.
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
    ap.add_argument("--minutes", type=float, default=10.0, help="training wall-clock budget")
    ap.add_argument("--max-rollouts", type=int, default=3000)
    ap.add_argument("--games", type=int, default=7500)
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--eval-games", type=int, default=256)
    ap.add_argument("--eval-depths", default="4,8,12,16,20")
    ap.add_argument("--max-states", type=int, default=20000)
    ap.add_argument("--out", default="gpurun_out/train_eval")
    ap.add_argument("--load", default=None, help="skip training and load this model directory")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS, AStar, PolicySearch, ValueSearch
    from librubiks.solving.evaluation import Evaluator
    from librubiks.train import Train
    from librubiks.utils import set_seeds

    set_seeds()
    results = {"settings": vars(args)}
    if args.load:
        net = Model.load(args.load)
    else:
        net = Model.create(ModelConfig())
        tr = Train(rollouts=1, batch_size=args.batch, rollout_games=args.games, rollout_depth=args.depth,
                   optim_fn=torch.optim.Adam, alpha_update=0, lr=2e-4, gamma=0.9, update_interval=100, agent=None,
                   evaluator=None, evaluation_interval=0, tau=0.3, reward_method="lapanfix",
                   adi_net_dtype=torch.bfloat16)
        # Train.train() runs a fixed number of rollouts; here the same loop body is driven by a wall-clock budget
        optimizer = torch.optim.Adam(net.parameters(), lr=2e-4)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 1, 0.9)
        generator = net.clone()
        losses, t_adi, t_opt = [], 0.0, 0.0
        t0 = time.perf_counter()
        rollout = 0
        while rollout < args.max_rollouts and time.perf_counter() - t0 < args.minutes * 60:
            generator = tr._update_gen_net(generator, net)
            torch.cuda.synchronize()
            a = time.perf_counter()
            data, ptar, vtar, w = tr.ADI_traindata(generator, 0)
            torch.cuda.synchronize()
            b = time.perf_counter()
            net.train()
            tot = torch.zeros((), dtype=torch.float64, device=data.device)
            batches = tr._get_batches(len(data), args.batch)
            for sl in batches:
                optimizer.zero_grad()
                pp, vp = net(data[sl])
                pl = tr.policy_criterion(pp, ptar[sl]) * w[sl]
                vl = tr.value_criterion(vp.squeeze(1), vtar[sl]) * w[sl]
                loss = torch.mean(pl + vl)
                loss.backward()
                optimizer.step()
                tot += loss.detach().double()
            torch.cuda.synchronize()
            c = time.perf_counter()
            t_adi += b - a
            t_opt += c - b
            losses.append(float(tot) / len(batches))
            if rollout and rollout % 100 == 0:
                scheduler.step()
            if rollout % 25 == 0:
                print(f"rollout {rollout}: loss {losses[-1]:.4f}  adi {b - a:.3f}s  opt {c - b:.3f}s  "
                      f"elapsed {time.perf_counter() - t0:.0f}s", flush=True)
            rollout += 1
        net.eval()
        net.save(os.path.join(args.out, "model"))
        states = rollout * args.games * args.depth
        results["training"] = {"rollouts": rollout, "states": states, "substates": 12 * states, "loss_first": losses[0],
                               "loss_last": losses[-1], "losses_every_25": losses[::25], "adi_seconds": t_adi,
                               "optim_seconds": t_opt, "adi_substates_per_sec": 12 * states / t_adi,
                               "train_states_per_sec": states / t_opt}
        print(json.dumps(results["training"]), flush=True)

    # ---- evaluation on the trained weights -------------------------------------------------------------
    net.eval()
    depths = [int(d) for d in args.eval_depths.split(",")]
    agents = {
        "MCTS c=0.6 graph": lambda: MCTS(net, c=0.6, search_graph=True),
        "AStar lambda=0.2 N=100": lambda: AStar(net, lambda_=0.2, expansions=100),
        "Greedy value": lambda: ValueSearch(net),
        "Greedy policy": lambda: PolicySearch(net),
    }
    results["evaluation"] = {}
    for name, make in agents.items():
        np.random.seed(0)
        agent = make()
        cap = args.max_states if "MCTS" in name or "AStar" in name else 200
        ev = Evaluator(args.eval_games, depths, max_time=None, max_states=cap)
        t = time.perf_counter()
        res, states_seen, times = ev.eval(agent)
        dt = time.perf_counter() - t
        per_depth = [ev.log_this_depth(res[i], states_seen[i], times[i], d) for i, d in enumerate(depths)]
        results["evaluation"][name] = {"max_states": cap, "seconds": dt, "per_depth": per_depth,
                                       "states_per_sec_overall": float(states_seen.sum() / dt)}
        print(name, [(p["depth"], round(p["share_completed"], 3)) for p in per_depth], f"{dt:.1f}s", flush=True)
        del agent
        torch.cuda.empty_cache()

    with open(os.path.join(args.out, "results.json"), "w") as f:
        json.dump(results, f, indent=1)
    print("wrote", os.path.join(args.out, "results.json"))


if __name__ == "__main__":
    main()
