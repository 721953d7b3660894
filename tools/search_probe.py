"""
MCTS search probes on one MI355X with the trained weights (the programs profiles are taken of, and same-box A/B runs):

  solve  [--dtype f32s|bf16] [--max-states N] [--trees B] [--out file.json]
         BASELINE configs[1] to completion: warm-up search, one search with the trajectory (iteration, launch size, running trees,
         wall time) logged at the agent's sync points, the same search again (every graph cached).  Under
         `rocprofv3 --kernel-trace --output-format csv` this is the trace `tools/rocprof_summary.py timeline` reads.
  window [bf16|f32s] [K]
         the steady-state pool (8 192 scrambles on 1 024 slots): 12 consecutive timed windows of K steps with the refills in each;
         the program for PMC passes on the tree kernel in a full forest.
  rtc    [f32s|bf16] [reps]
         configs[1] to completion `reps` times (prepared forest, 30 warm-up iterations): seconds per run, best M nodes/s.
  groups [f32s|bf16] [G] [games] [slots]
         the tree kernel of one group next to the network of another: the pool searched as G independent groups (slots / G each,
         every group on a HIP stream of its own, rounds alternating) against the one-group form, same box, a, b, a, b.  Per form:
         the steady-state window (ms per lock-step step of all 1 024 slots) and the whole pool.
  ab     <owner.attribute> <value_a> <value_b>
         same-box A/B of a module / class attribute (e.g. mcts_device.RUNG_RATIO 0.9 0.95, model.SplitF32Net.fused_head 1 0,
         agents.MCTS.sync_every via `sync_every 16 32`): variants alternate a, b, a, b in one process; per variant configs[1] to
         completion three times and a 4 096-game pool.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import F32_SPLIT, Model  # noqa: E402
from librubiks.solving import mcts_device as md  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

WEIGHTS = os.path.join(ROOT, "weights", "fc_small_r1")
DT = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}


def _timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, time.perf_counter() - t


def solve(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "solve_run.json"))
    ap.add_argument("--trees", type=int, default=1024)
    ap.add_argument("--max-states", type=int, default=175000)
    ap.add_argument("--dtype", default="f32s", choices=list(DT))
    args = ap.parse_args(argv)
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(args.trees, 20, True)
    agent = MCTS(Model.load(WEIGHTS).eval(), c=0.6, search_graph=True, net_dtype=DT[args.dtype])
    _, prep = _timed(lambda: agent.prepare(args.trees, args.max_states))
    agent.search_batch(cubes, None, args.max_states, max_iterations=30)
    log, state, orig = [], {"it": 0, "t0": None}, md.MCTSForest.step

    def step(self, *a, **k):
        r = orig(self, *a, **k)
        state["it"] += 1
        if state["it"] % agent.sync_every == 0:
            torch.cuda.synchronize()
            log.append((state["it"], self.G, int((self.status == md.RUNNING).sum().item()), round(time.perf_counter() - state["t0"], 4)))
        return r

    md.MCTSForest.step, graph_steps, md.MCTSForest.GRAPH_STEPS = step, md.MCTSForest.GRAPH_STEPS, 1   # (the logged run: one graph launch per iteration)
    state["t0"] = time.perf_counter()
    res, total = _timed(lambda: agent.search_batch(cubes, None, args.max_states))
    md.MCTSForest.step, md.MCTSForest.GRAPH_STEPS = orig, graph_steps
    res2, again = _timed(lambda: agent.search_batch(cubes, None, args.max_states))
    out = {"dtype": args.dtype, "prepare_seconds": prep, "seconds_logged_run": total, "solved": float(res.solved.mean()), "nodes": int(res.nodes.sum()),
           "iterations_max": int(res.iterations.max()), "second_run_seconds": again, "second_run_nodes_per_sec": float(res2.nodes.sum()) / again,
           "second_run_same_results": bool(np.array_equal(res.nodes, res2.nodes) and np.array_equal(res.lengths, res2.lengths)),
           "forest_gb": round(agent.forest.bytes_allocated() / 1e9, 2), "forest_mapped_on_demand": agent.forest.vmm,
           "stats": dict(agent.refill_stats), "trajectory_it_G_running_t": log[:: max(1, len(log) // 60)] + log[-1:]}
    print(json.dumps({k: v for k, v in out.items() if k != "trajectory_it_G_running_t"}), flush=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)


def window(argv):
    dt, K = DT[argv[0] if argv else "bf16"], int(argv[1]) if len(argv) > 1 else 20
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(8192, 20, True)
    agent = MCTS(Model.load(WEIGHTS).eval(), c=0.6, search_graph=True, net_dtype=dt)
    run = agent.start_batch(cubes, None, 175000, slots=1024)
    while run.next_game < 2048 + 128:
        run.round()
    for w in range(12):
        torch.cuda.synchronize()
        r0, n0 = run.stats["refills"], run.nodes_now()
        torch.cuda.synchronize()
        t, left = time.perf_counter(), K
        while left > 0:
            b = run.it
            run.round(left)
            left -= run.it - b
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t
        print(f"window {w}: {dtm / K * 1e3:.4f} ms/step, {(run.nodes_now() - n0) / dtm / 1e6:.2f} M/s, refills {run.stats['refills'] - r0}", flush=True)


def rtc(argv):
    name, reps = (argv[0] if argv else "f32s"), int(argv[1]) if len(argv) > 1 else 3
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(1024, 20, True)
    agent = MCTS(Model.load(WEIGHTS).eval(), c=0.6, search_graph=True, net_dtype=DT[name])
    _, prep = _timed(lambda: agent.prepare(1024, 175000))
    agent.search_batch(cubes, None, 175000, max_iterations=30)
    secs = []
    for _ in range(reps):
        res, s = _timed(lambda: agent.search_batch(cubes, None, 175000))
        secs.append(s)
    print(f"{name}: prepare {prep:.3f} s; seconds {[round(x, 3) for x in secs]}, {res.nodes.sum() / min(secs) / 1e6:.2f} M nodes/s best, iterations "
          f"{int(res.iterations.max())}, solved {res.solved.mean():.3f}, forest {agent.forest.bytes_allocated() / 1e9:.1f} GB "
          f"({'mapped on demand' if agent.forest.vmm else 'allocated up front'}), env {dict((k, v) for k, v in os.environ.items() if k.startswith('RUBIKS_'))}")


def groups(argv):
    name, G, games = (argv[0] if argv else "f32s"), int(argv[1]) if len(argv) > 1 else 2, int(argv[2]) if len(argv) > 2 else 8192
    CAP, SLOTS, K = 175000, int(argv[3]) if len(argv) > 3 else 1024, 40
    np.random.seed(0)
    pool, _, _ = cube.scramble_batch(games, 20, True)
    states = pool.numpy()
    model = Model.load(WEIGHTS).eval()

    def one_form(g):
        agents = [MCTS(model, c=0.6, search_graph=True, net_dtype=DT[name]) for _ in range(g)]
        streams = [torch.cuda.Stream() for _ in range(g)] if g > 1 else [torch.cuda.current_stream()]
        for a, st in zip(agents, streams):          # one after the other: a capture does not tolerate foreign launches
            with torch.cuda.stream(st):
                a.prepare(SLOTS // g, CAP)
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runs = []
        for i, (a, st) in enumerate(zip(agents, streams)):
            with torch.cuda.stream(st):
                runs.append(a.start_batch(states[i::g], None, CAP, slots=SLOTS // g))

        def rounds(pred, max_steps=None):
            while any(pred(r) for r in runs):
                for r, st in zip(runs, streams):
                    if pred(r):
                        with torch.cuda.stream(st):
                            r.round(max_steps)
        rounds(lambda r: not r.done and r.next_game < min(2 * SLOTS // g + 64, r.n_games))      # slots hold trees of every age
        torch.cuda.synchronize()
        n0 = sum(r.nodes_now() for r in runs)
        for r in runs:
            r._target = r.it + K
        torch.cuda.synchronize()
        tw = time.perf_counter()
        while any(not r.done and r.it < r._target for r in runs):
            for r, st in zip(runs, streams):
                if not r.done and r.it < r._target:
                    with torch.cuda.stream(st):
                        r.round(r._target - r.it)
        torch.cuda.synchronize()
        window_s = time.perf_counter() - tw
        n1 = sum(r.nodes_now() for r in runs)
        rounds(lambda r: not r.done)
        res = []
        for r, st in zip(runs, streams):
            with torch.cuda.stream(st):
                res.append(r.finish())
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        nodes = sum(int(x.nodes.sum()) for x in res)
        out = {"groups": g, "window_ms_per_step_of_all_slots": round(window_s / K * 1e3, 4), "window_nodes_per_sec": round((n1 - n0) / window_s),
               "pool_seconds": round(total, 3), "pool_nodes_per_sec": round(nodes / total), "solved": float(np.mean(np.concatenate([x.solved for x in res]))),
               "nodes": nodes}
        for a in agents:
            if a.forest is not None:
                a.forest.close()
        del agents, runs
        torch.cuda.empty_cache()
        return out

    for g in (1, G, 1, G):
        print(name, json.dumps(one_form(g)), flush=True)


def rtc_groups(argv):
    """BASELINE configs[1] (1 024 depth-20 scrambles as ONE batch, to completion) as G independent groups of 1 024 / G trees, each on a
    HIP stream of its own with rounds alternating: in the tail of the run (tens of trees left) a step is one tree kernel (~92 us on a
    few CUs) behind one network pass (~84 us of short kernels) -- two groups can run the one beside the other."""
    name, G, reps = (argv[0] if argv else "f32s"), int(argv[1]) if len(argv) > 1 else 2, int(argv[2]) if len(argv) > 2 else 2
    CAP, N = 175000, 1024
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(N, 20, True)
    states = cubes.numpy()
    model = Model.load(WEIGHTS).eval()

    def one_form(g):
        agents = [MCTS(model, c=0.6, search_graph=True, net_dtype=DT[name]) for _ in range(g)]
        streams = [torch.cuda.Stream() for _ in range(g)] if g > 1 else [torch.cuda.current_stream()]
        for a, st in zip(agents, streams):
            with torch.cuda.stream(st):
                a.prepare(N // g, CAP)
            torch.cuda.synchronize()
        secs = []
        for rep in range(reps + 1):      # the first pass warms up (library heuristics, clocks)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            runs = []
            for i, (a, st) in enumerate(zip(agents, streams)):
                with torch.cuda.stream(st):
                    runs.append(a.start_batch(states[i::g], None, CAP))
            while any(not r.done for r in runs):
                for r, st in zip(runs, streams):
                    if not r.done:
                        with torch.cuda.stream(st):
                            r.round()
            res = []
            for r, st in zip(runs, streams):
                with torch.cuda.stream(st):
                    res.append(r.finish())
            torch.cuda.synchronize()
            if rep:
                secs.append(time.perf_counter() - t0)
        nodes = sum(int(x.nodes.sum()) for x in res)
        out = {"groups": g, "seconds": [round(x, 3) for x in secs], "nodes_per_sec_best": round(nodes / min(secs)), "nodes": nodes,
               "solved": float(np.mean(np.concatenate([x.solved for x in res]))), "iterations": [int(x.iterations.max()) for x in res]}
        for a in agents:
            if a.forest is not None:
                a.forest.close()
        del agents, runs
        torch.cuda.empty_cache()
        return out

    for g in (1, G, 1, G):
        print(name, json.dumps(one_form(g)), flush=True)


def ab(argv):
    attr, vals = argv[0], [float(v) if "." in v else int(v) for v in argv[1:3]]
    CAP = 175000
    np.random.seed(0)
    batch, _, _ = cube.scramble_batch(1024, 20, True)
    pool, _, _ = cube.scramble_batch(4096, 20, True)
    model = Model.load(WEIGHTS).eval()
    owner = None
    if "." in attr:
        path, attr = attr.rsplit(".", 1)
        mod, _, cls = path.partition(".")
        owner = importlib.import_module("librubiks." + ("solving." if mod in ("mcts_device", "agents", "astar_device") else "") + mod)
        owner = getattr(owner, cls) if cls else owner
    for name, v in ((f"{attr}={vals[0]}", vals[0]), (f"{attr}={vals[1]}", vals[1]), (f"{attr}={vals[0]} again", vals[0]), (f"{attr}={vals[1]} again", vals[1])):
        if owner is None:
            agent = MCTS(model, c=0.6, search_graph=True, **{attr: int(v)})      # a constructor argument of the agent
        else:
            setattr(owner, attr, type(getattr(owner, attr))(v))
            agent = MCTS(model, c=0.6, search_graph=True)
        agent.prepare(1024, CAP)
        agent.search_batch(batch, None, CAP)
        runs = []
        for _ in range(3):
            r, s = _timed(lambda: agent.search_batch(batch, None, CAP))
            runs.append(s)
        rp, tp = _timed(lambda: agent.search_batch(pool, None, CAP, slots=1024))
        print(name, json.dumps({"batch_seconds": [round(x, 4) for x in runs], "batch_nodes_per_sec": round(float(r.nodes.sum()) / min(runs)),
                                "solved": float(r.solved.mean()), "nodes": int(r.nodes.sum()), "pool_seconds": round(tp, 4),
                                "pool_nodes_per_sec": round(float(rp.nodes.sum()) / tp)}), flush=True)
        del agent
        torch.cuda.empty_cache()


if __name__ == "__main__":
    {"solve": solve, "window": window, "rtc": rtc, "ab": ab, "groups": groups, "rtc_groups": rtc_groups}[sys.argv[1]](sys.argv[2:])
