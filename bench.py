"""
bench.py -- MCTS node expansions/sec on depth-20 scrambles (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1 024 concurrent depth-20 MCTS trees per GPU (np.random.seed(0), the
reference's scramble stream), MCTS c = 0.6 with graph search, max_states 175 000 (the reference's default,
runeval.py:42-44), fc_small policy/value net with the ADI-trained weights of weights/fc_small_r1 (glorot weights
from torch.manual_seed(0) if absent).
A "step" is one lock-step MCTS iteration of every tree on the rank: expand the leaf's 12 children (HIP), policy /
value network on the NEW children (11 packed rows per tree; PyTorch-ROCm GEMMs), backup + PUCT descent (HIP).

What is timed.  The 1 024 tree slots are fed from a pool of `--pool-factor` x 1 024 scrambles: a finished tree
hands its slot to the next waiting scramble (continuous batching), so the slots hold trees of every age.  The
pool is first advanced UNTIMED until as many scrambles again as there are slots have been started (the age mix
is then that of a long-running evaluation), then W warm-up steps, then EXACTLY K steps are timed between
barrier + synchronize on both sides, harvesting and refilling included:
  value = unique states inserted into the trees of all ranks during the K timed steps / max-over-ranks seconds.
This is done once per network precision (a "leg"):
  f32s  fp32 accuracy on the f16 matrix cores (the reference's net runs in fp32, librubiks/model.py:131-141)  ->  `value`, `dtype`
  f32   the fp32 MFMA GEMM chain as is (window only)          bf16  the fast engine  ->  `legs.*`
Each leg also reports (sum of len(agent) / wall seconds of the batched search):
  pool_run           the whole pool searched to completion / its wall time (prep, window and tail included)
  run_to_completion  BASELINE configs[1] itself: the first 1 024 scrambles as ONE batch, to completion, + solve rate
Ranks own disjoint scramble slices (weak scaling); the only collectives are the barrier, the max/sum reductions
of the result and one all_gather of per-game results.
Further legs on the same line (`--extra-legs`): `astar` = BASELINE configs[2] (4 096 depth-20 scrambles, AStar lambda 0.2, N 100:
K timed iterations + the solve run at max_states 175 000, phase times and the roofline of its dominant kernel) and `config5` =
one GPU's share of BASELINE configs[4] (8 192 concurrent depth-24 trees: timed window + the 8 192 as one batch to completion).
The scalars that summarise all of this are repeated in `config.results`.

Also printed on the same JSON line:
  roofline      dominant kernel of the headline leg's step = the first hidden GEMM (MFMA bound)
  roofline_env  the hand-written environment kernels in isolation (HBM bound; multi_rotate is the north_star's
                roofline target) at 2^14 .. 2^26 states
  phases        per-phase milliseconds of one MCTS step (HIP events, eager replay of the same step)
  cpu_baseline  the restated reference agent (oracle/, NumPy + torch CPU) on this box's host cores

This file owns the contract (arguments, what is timed, the JSON line's keys) and the cpu_baseline leg -- besides tests/ and smoke() the only
importer of oracle/.  The pieces live in tools/benchlib/: kernels.py (per-kernel timings, rooflines), legs.py (the MCTS / A* / ADI legs),
launch.py (ranks, preflight, the scaling figure), line.py (the compact stdout line + the detail file).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
from benchlib.common import *   # noqa: E402,F401,F403  (json, time, np, torch, dist, ROOT, roofline constants, progress, event_ms, LEG_DTYPE)
from benchlib.kernels import boundary_calls, env_roofline, step_rooflines   # noqa: E402,F401
from benchlib.launch import SCALE_REF, launch_ranks, preflight, scale_efficiency   # noqa: E402,F401
from benchlib.legs import adi_leg, astar_leg, draw_scrambles, release_node_stores, replay_solutions, run_leg   # noqa: E402,F401
from benchlib.line import LINE_LIMIT, STR_LIMIT, compact_line, emit   # noqa: E402,F401


def cpu_baseline(model, depth, budget_s=14.0, max_states=5000):
    """
    Restated reference MCTS (oracle/) with the same weights on the host cores, bounded sample.
    The reference leaves torch's thread count at its default; on a many-core host that is far from
    the best choice for 12-row batches, so a short calibration picks the fastest of a few thread
    counts and the reported number is the CPU's best.
    """
    import copy
    from oracle import agents as oa
    from oracle import cube as oc
    cpu_model = copy.deepcopy(model).cpu().float().eval()
    net = oa.TorchNet(cpu_model, device="cpu")

    def run(seconds, iters_cap):
        np.random.seed(0)
        nodes, games, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            s, _, _ = oc.scramble(depth, True)
            agent = oa.MCTS(net, c=0.6, search_graph=True)
            agent.search(s, max_states, max_iterations=iters_cap)
            nodes += len(agent)
            games += 1
        return nodes, games, time.perf_counter() - t0

    default_threads = torch.get_num_threads()
    best = (0.0, default_threads)
    for th in sorted({1, 4, 8, 16, min(32, default_threads)}):
        torch.set_num_threads(th)
        nodes, _, dt = run(1.0, 40)
        best = max(best, (nodes / dt, th))
    torch.set_num_threads(best[1])
    nodes, games, dt = run(budget_s, None)
    torch.set_num_threads(default_threads)
    return {"value": round(nodes / dt, 1), "unit": "node expansions/s", "cores": best[1],
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{games} depth-{depth} scrambles x max_states={max_states}, single-tree MCTS c=0.6 "
                      f"(oracle/agents.py on NumPy + torch CPU fp32, {best[1]} torch threads picked by calibration), "
                      f"{dt:.1f} s",
            "sample_short": f"{games} depth-{depth} scrambles, max_states {max_states}, single-tree MCTS (oracle port, torch CPU fp32, {best[1]} threads), {dt:.1f} s",
            "env_ops": cpu_env_ops(), "bfs_config1": cpu_bfs_config1(), "boundary_calls": boundary_calls()}


def cpu_env_ops(sizes=(10_000, 196_608), warm=5, reps=20):
    """The reference's NumPy cube expressions (restated in oracle/cube.py) on ONE host core, median of `reps`
    (N = 10 000 is the reference's own multi_op_size, 196 608 = 16 384 x 12 of config #4)."""
    from oracle import cube as oc
    out = {"cores": 1, "unit": "M states/s", "reps": reps}
    rng = np.random.RandomState(0)
    for n in sizes:
        base = np.tile(oc.get_solved(), (n, 1))
        for _ in range(30):
            base = oc.multi_rotate_actions(base, rng.randint(0, 12, n))
        acts = rng.randint(0, 12, n)
        faces, dirs = acts // 2, 1 - acts % 2
        row = {}
        for name, fn in (("multi_rotate", lambda: oc.multi_rotate(base, faces, dirs)),
                         ("multi_is_solved", lambda: oc.multi_is_solved(base)),
                         ("as_oh", lambda: oc.as_oh(base))):
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t)
            row[name] = round(n / float(np.median(ts)) / 1e6, 2)
        out[str(n)] = row
    return out


def cpu_bfs_config1():
    """BASELINE config #1 (10 depth-5 scrambles after set_seeds(), BFS) on the restated FIFO loop, one host core."""
    from oracle import agents as oa
    g = np.load(os.path.join(ROOT, "tests", "golden", "bfs_golden.npz"))
    agent, seen, t0 = oa.BFS(), 0, time.perf_counter()
    for s in g["states"]:
        agent.search(s, 10_000_000)
        seen += len(agent)
    dt = time.perf_counter() - t0
    return {"games": int(len(g["states"])), "states_seen": int(seen), "seconds": round(dt, 3),
            "states_per_sec": round(seen / dt, 1), "cores": 1}


def cpu_adi(model, games=64, depth=32):
    """The restated reference data generation (oracle/train.py, NumPy + torch CPU fp32 value network) on the host cores: a
    bounded sample of config #4's rollout (64 x 32 = 2 048 states, an eighth of the 16 384), torch's default thread count."""
    import copy
    from oracle import agents as oa
    from oracle import train as ot
    cpu_model = copy.deepcopy(model).cpu().float().eval()
    value = oa.TorchNet(cpu_model, device="cpu").value
    np.random.seed(7)
    ot.adi_traindata(value, 4, depth, "lapanfix", 0.5)          # warm-up
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 6.0 or reps < 1:
        ot.adi_traindata(value, games, depth, "lapanfix", 0.5)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": round(reps * games * depth / dt, 1), "unit": "states/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x oracle.train.adi_traindata({games} games x {depth} moves = {games * depth} states, {12 * games * depth} substates), "
                      f"NumPy cube ops + torch CPU fp32 value network, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--trees", type=int, default=1024, help="concurrent MCTS trees (slots) per GPU")
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--legs", default="f32s,f32:window,bf16",
                    help="network engines to measure; the FIRST one is the headline `value`: f32s = fp32 accuracy on the f16 matrix "
                         "cores (SplitF32Net), f32 = fp32 MFMA GEMMs (the reference's arithmetic as is), bf16 = the fast engine; "
                         "`:window` = timed window only (no pool tail, no run to completion)")
    ap.add_argument("--scale-ref-value", type=float, default=float(os.environ.get("RUBIKS_SCALE_REF", 0)) or None,
                    help="the one-GPU `value` of this workload (or RUBIKS_SCALE_REF): `efficiency` of an N-GPU run = value / (N x it).  Without "
                         "it the record a --gpus 1 run of the same checkout left next to bench.py is used, else efficiency is null -- the "
                         "line always carries value_per_gpu (= value / N), which is all a driver needs to compute the curve itself")
    ap.add_argument("--extra-legs", default="auto",
                    help="auto = astar,config5,adi on one GPU; config5 alone under --gpus N > 1 (the curve needs the MCTS window, the run to "
                         "completion and the config-5 share: A* and ADI at two precisions on every rank add a minute of wall time and "
                         "nothing to it).  astar = BASELINE configs[2] (4 096 depth-20 A* problems per GPU); config5 = one GPU's share of configs[4] "
                         "(8 192 concurrent depth-24 trees); adi = configs[3] (data generation of a 16 384-state ADI rollout); none = "
                         "none of them.  Each runs at f32s, then bf16")
    ap.add_argument("--adi-states", type=int, default=16384, help="states per ADI rollout of the `adi` leg (BASELINE configs[3]: 16 384 = 512 games x 32 moves)")
    ap.add_argument("--astar-problems", type=int, default=4096)
    ap.add_argument("--config5-trees", type=int, default=8192)
    ap.add_argument("--config5-max-states", type=int, default=175000,
                    help="per-tree cap of the config5 leg = the reference's max_states: 8 192 x 175 001 node rows (408 GB) are reserved address "
                         "space, memory is mapped behind the rows the trees reach (rounds 1-3 allocated up front and had to stop at 50 000)")
    ap.add_argument("--pool-factor", type=int, default=8, help="scrambles in the pool per tree slot")
    ap.add_argument("--prep-cap", type=int, default=4000, help="most untimed iterations before the timed window")
    ap.add_argument("--window-only", action="store_true", help="skip the pool's tail and the runs to completion")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-roofline", action="store_true")
    ap.add_argument("--phase-reps", type=int, default=20)
    ap.add_argument("--weights", default=os.path.join(ROOT, "weights", "fc_small_r1"),
                    help="checkpoint directory (model.pt + config.json); random-init weights if it does not exist")
    ap.add_argument("--architecture", default="fc_small", choices=["fc_small", "fc_big", "res_small", "res_big"],
                    help="network architecture when --weights does not name a checkpoint directory (random-init weights then)")
    ap.add_argument("--solve-max-states", type=int, default=175000,
                    help="per-tree / per-problem cap = the reference's max_states (default: its CLI default, runeval.py:42-44)")
    ap.add_argument("--level-budget", default="auto",
                    help="new tree levels a PUCT descent may walk per step before it is suspended (0 = strict lock step; "
                         "auto = the agent's default: a budget while scrambles are waiting for a slot, none for the tail)")
    ap.add_argument("--first-layer-table", default="auto", choices=["auto", "onehot"],
                    help="bf16 engine's input layer: the fused matrix-core kernel from the cube codes, or the explicit one-hot + library GEMM")
    ap.add_argument("--preflight", action="store_true",
                    help="run only the preflight (device binding, free HBM, the collectives the legs use) and print its one-line result; "
                         "with more than one rank the preflight always runs before the first leg")
    ap.add_argument("--as-rank", default=None, metavar="R/W",
                    help="single process, no process group: take rank R's share of a W-rank run's scrambles (tests compare "
                         "the ranks of a distributed run with these)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file that receives the full result (per-leg detail, phases, the roofline_env ladder); the stdout line names it")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and not args.as_rank:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))   # no launcher around us: start the ranks ourselves (before any HIP call)
    if args.level_budget != "auto":
        args.level_budget = int(args.level_budget)
    legs = [x.split(":")[0] for x in args.legs.split(",") if x]
    leg_window_only = {x.split(":")[0]: x.endswith(":window") for x in args.legs.split(",") if x}
    assert legs and all(x in LEG_DTYPE for x in legs)
    if args.extra_legs == "auto":
        args.extra_legs = "astar,config5,adi" if args.gpus == 1 else "config5"
    extra = [] if args.extra_legs in ("", "none") else [x for x in args.extra_legs.split(",") if x]
    assert all(x in ("astar", "config5", "adi") for x in extra)

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    from librubiks.solving.sharding import pick_backend
    # one process per GPU over RCCL; RUBIKS_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    backend, device_index, coll_device = pick_backend(os.environ, torch.cuda.device_count(), local_rank)
    torch.cuda.set_device(device_index)
    if world > 1:
        import datetime
        limit = datetime.timedelta(minutes=10)      # a rank that never arrives fails the others instead of hanging them
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    pre = None
    if world > 1 or args.preflight:
        need_gb = 190 if "config5" in extra and args.config5_max_states >= 100000 else 90      # HBM the largest leg maps (config-5 share: 146-165 GB)
        need_gb = float(os.environ.get("RUBIKS_PREFLIGHT_NEED_GB", need_gb))                    # (tests: an unmeetable requirement)
        pre = preflight(rank, world, local_rank, backend, device_index, coll_device, need_gb)
        if args.preflight:
            if rank == 0:
                print(json.dumps({"preflight": "ok", "n_gpus": world, **pre}), flush=True)
            if world > 1:
                dist.destroy_process_group()
            return

    from librubiks.model import Model, ModelConfig

    # ---- synthetic inputs: the reference's scramble stream for the configs' own games, a private stream for pool filler -----
    per_rank = args.trees * args.pool_factor
    slice_rank, slice_world = (rank, world) if not args.as_rank else tuple(int(x) for x in args.as_rank.split("/"))
    config_roots, pool_roots = draw_scrambles(args.trees, per_rank, args.depth, slice_rank, slice_world)

    torch.manual_seed(0)
    if os.path.isdir(args.weights):
        model = Model.load(args.weights).eval()
        weights_note = f"{os.path.relpath(args.weights, ROOT)} (ADI-trained on one MI355X by tools/train_eval.py; see weights/README.md)"
    else:
        model = Model.create(ModelConfig(architecture=args.architecture)).eval()
        weights_note = f"random-init {args.architecture} (glorot, torch.manual_seed(0))"

    results, extras = {}, {}
    for name in legs:
        leg, engine, agent = run_leg(name, model, pool_roots, config_roots, args, world, coll_device, args.trees, args.solve_max_states,
                                     window_only=args.window_only or leg_window_only[name], full_warm=False)
        results[name] = leg
        if rank == 0:
            progress(f"mcts leg {name} done")
        if rank == 0 and args.phase_reps:
            extras[name] = step_rooflines(engine, agent, config_roots, args, name)
        del engine, agent
        torch.cuda.empty_cache()
    del pool_roots
    release_node_stores()

    # ---- BASELINE configs[2]: A* ------------------------------------------------------------------------------------------
    astar = {}
    if "astar" in extra:
        a_roots, _ = draw_scrambles(args.astar_problems, args.astar_problems, 20, slice_rank, slice_world)
        for name in ("f32s", "bf16"):
            astar[name] = astar_leg(name, model, a_roots, args, world, coll_device)
            if rank == 0:
                progress(f"astar leg {name} done")
        del a_roots
    # ---- one GPU's share of BASELINE configs[4]: 8 192 concurrent depth-24 trees -------------------------------------------
    config5 = {}
    if "config5" in extra:
        c_roots, c_pool = draw_scrambles(args.config5_trees, 3 * args.config5_trees, 24, slice_rank, slice_world)
        for name in ("f32s", "bf16"):
            leg, engine, agent = run_leg(name, model, c_pool, c_roots, args, world, coll_device, args.config5_trees, args.config5_max_states,
                                         full_warm=False)
            leg.pop("pool_run", None)   # the pool only feeds the window here (3 x 8 192 scrambles), its total is not a result
            config5[name] = leg
            if rank == 0:
                progress(f"config5 leg {name} done")
            del engine, agent
            torch.cuda.empty_cache()
        del c_roots, c_pool
        release_node_stores()

    # ---- BASELINE configs[3]: data generation of an ADI rollout ---------------------------------------------------------------
    adi = {}
    if "adi" in extra:
        for name in ("f32s", "bf16"):
            adi[name] = adi_leg(name, model, args, world, coll_device)
            if rank == 0:
                progress(f"adi leg {name} done")
        torch.cuda.empty_cache()

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    head = results[legs[0]]
    rtc_of = lambda leg: (leg.get("run_to_completion") or {})   # noqa: E731
    summary = {   # the scalars of this line that matter, in one flat place (everything else is detail under `legs`, `astar`, `config5_share`)
        "max_states": args.solve_max_states,
        "value_pool_run": (head.get("pool_run") or {}).get("nodes_per_sec"),
        "value_run_to_completion": rtc_of(head).get("nodes_per_sec"),
        "run_to_completion_seconds": rtc_of(head).get("seconds"),
        "run_to_completion_seconds_incl_prepare": rtc_of(head).get("seconds_incl_prepare_rank0"),
        "prepare_seconds": head.get("prepare_seconds_rank0"), "forest_gb": (head.get("forest_rank0") or {}).get("hbm_behind_the_forest_gb"),
        "solve_rate": rtc_of(head).get("solve_rate"), "solve_rate_ci95": rtc_of(head).get("ci95"),
        "mean_solution_length": rtc_of(head).get("mean_solution_length"),
        "result_flushes_in_window": head.get("result_flushes_in_window"),
        # solutions walked through cube.multi_rotate / multi_is_solved outside the timed regions (rank 0's games; a mismatch aborts the run)
        "run_to_completion_solutions_replayed": rtc_of(head).get("solutions_replayed_to_solved_rank0"),
        "pool_solutions_replayed": (head.get("pool_run") or {}).get("solutions_replayed_to_solved"),
        # trees ended by a full path store: 0 -- the default store has no bound (descents of any length, as in the reference)
        "path_overflow_trees": (rtc_of(head).get("path_overflow_trees_rank0") or 0) + ((head.get("pool_run") or {}).get("path_overflow_trees") or 0),
    }
    for name in legs[1:]:
        summary[f"{name}_value"] = results[name]["value"]
        if rtc_of(results[name]):
            summary[f"{name}_value_pool_run"] = results[name]["pool_run"]["nodes_per_sec"]
            summary[f"{name}_value_run_to_completion"] = rtc_of(results[name])["nodes_per_sec"]
            summary[f"{name}_solve_rate"] = rtc_of(results[name])["solve_rate"]
    for name, leg in astar.items():
        summary[f"astar_{name}_states_per_sec"] = leg["value"]
        if "solve_run" in leg:
            summary[f"astar_{name}_solve_run_states_per_sec"] = leg["solve_run"]["states_per_sec"]
            summary[f"astar_{name}_solve_rate"] = leg["solve_run"]["solve_rate"]
            summary[f"astar_{name}_solutions_replayed"] = leg["solve_run"].get("solutions_replayed_to_solved_rank0")
        if "roofline" in leg:
            summary[f"astar_{name}_roofline_frac"] = leg["roofline"]["frac"]
    for name, leg in adi.items():
        summary[f"adi_{name}_states_per_sec"] = leg["value"]
        summary[f"adi_{name}_ms_per_rollout"] = leg["ms_per_rollout"]
        if "roofline" in leg:
            summary[f"adi_{name}_roofline_frac"] = leg["roofline"]["frac"]
    for name, leg in config5.items():
        summary["config5_share_max_states"] = args.config5_max_states
        summary[f"config5_share_{name}_value"] = leg["value"]
        if rtc_of(leg):
            summary[f"config5_share_{name}_run_to_completion"] = rtc_of(leg)["nodes_per_sec"]
            summary[f"config5_share_{name}_solve_rate"] = rtc_of(leg)["solve_rate"]
            summary[f"config5_share_{name}_forest_gb"] = leg["forest_rank0"]["hbm_behind_the_forest_gb"]
    # ---- the scaling curve, without post-processing: a one-GPU run leaves its value next to the script (SCALE_REF); an N-GPU run
    # of the same workload on the same checkout divides by it.  null when no such record exists (or the workload differs).
    workload_key = {"trees": args.trees, "depth": args.depth, "max_states": args.solve_max_states, "leg": legs[0], "pool_factor": args.pool_factor,
                    "steps": args.steps, "warmup": args.warmup}
    efficiency, efficiency_note = scale_efficiency(world, head["value"], workload_key, os.path.join(ROOT, SCALE_REF), write=not args.as_rank,
                                                   gpu=torch.cuda.get_device_name(device_index), ref_value=args.scale_ref_value)
    weights_short = os.path.relpath(args.weights, ROOT) if os.path.isdir(args.weights) else "random-init"
    result = {
        "metric": "MCTS node expansions/sec, depth-20 scrambles", "value": head["value"],
        "unit": "node expansions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "scaling_measured": world > 1,
        "rank_values": head.get("rank_values"), "value_per_gpu": round(head["value"] / world, 1), "value_spread": head.get("value_spread"),
        "efficiency": efficiency, "efficiency_note": efficiency_note,
        "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
        "config": {"workload": f"{args.trees} concurrent depth-{args.depth} MCTS trees per GPU (c=0.6, graph search, max_states "
                               f"{args.solve_max_states}), slots refilled from a pool of {args.pool_factor} x {args.trees} scrambles "
                               f"per GPU; {model.config.architecture} net, weights: {weights_note}",
                   # (the driver's record keeps 120 characters of a string: the line carries the *_short forms, the detail file both)
                   "workload_short": f"{args.trees} depth-{args.depth} MCTS trees/GPU, c=0.6, graph search, max_states {args.solve_max_states}, "
                                     f"pool {args.pool_factor}x{args.trees}, {model.config.architecture} {weights_short}",
                   "trees_per_gpu": args.trees, "pool_scrambles_per_gpu": per_rank, "select_level_budget": args.level_budget,
                   "scramble_depth": args.depth, "max_states": args.solve_max_states, "parallelism": f"scramble-sharded x{world}",
                   "scrambles": "configs' own games: the reference's stream (np.random.seed(0), scramble(depth, True)), rank r owns games "
                                "[r n, (r + 1) n); pool filler: a private stream per rank (seed 1 000 003 + rank)",
                   "timed_region": "K lock-step iterations of the stationary pool (harvest + refill included) between barrier + synchronize; "
                                   "prep (until 2 x trees scrambles have been started) and warm-up untimed; result flushes (graph completion + "
                                   "BFS of 256 finished trees on a side stream) fall where they fall: results.result_flushes_in_window",
                   "timed_region_short": "K lock-step steps of the stationary pool between barrier+synchronize; prep and warm-up untimed",
                   "results": summary},
        "value_note": f"headline = the '{legs[0]}' leg: the reference's network arithmetic is fp32 (librubiks/model.py:131-141); f32s reaches "
                      "fp32 accuracy with three f16 MFMA products per layer (error against float64 within 1.25 x the fp32 forward's, "
                      "in practice below it: tests/test_net_gpu.py), f32 is the fp32 MFMA GEMM chain as is, bf16 the fast engine; all under `legs`",
        "value_run_to_completion": summary["value_run_to_completion"],
        "value_pool_run": summary["value_pool_run"],
        "solve_rate": summary["solve_rate"],
        "scaling_note": ("this line is one point of the curve: value = all ranks' nodes / max-over-ranks seconds; rank_values = every rank's own rate"
                         if world > 1 else "no multi-GPU curve has been measured by the build (gpurun hands out one GPU); N > 1 is covered by gloo tests"),
        "preflight": pre,
        "legs": results,
    }
    if astar:
        result["astar"] = dict(astar, workload=f"BASELINE configs[2]: {args.astar_problems} depth-20 scrambles per GPU, AStar lambda=0.2 N=100, "
                                               f"max_states {args.solve_max_states}")
    if adi:
        result["adi"] = dict(adi, workload=f"BASELINE configs[3]: data generation of one ADI rollout, {args.adi_states} states ({args.adi_states // 32 // world} games x 32 moves "
                                           f"per GPU) -> {12 * args.adi_states} substates, reward method lapanfix (reference train.py:257-339)")
    if config5:
        result["config5_share"] = dict(config5, workload=f"one GPU's share of BASELINE configs[4]: {args.config5_trees} concurrent depth-24 MCTS "
                                                         f"trees, max_states {args.config5_max_states}")
    for name in legs:
        if name in extras:
            phases, roofline, group, roofline_input, rows = extras[name]
            results[name]["phases_ms"] = phases
            results[name]["roofline"] = roofline
            results[name]["roofline_net_group"] = group
            results[name]["net_rows_per_step"] = rows
            if roofline_input:
                results[name]["roofline_input_layer"] = roofline_input
    result["roofline"] = dict(results[legs[0]].get("roofline") or {})
    if not args.no_env_roofline and world == 1:
        torch.cuda.empty_cache()
        result["roofline_env"] = [r for log2n in (14, 20, 24, 26) for r in env_roofline(log2n)]
        mr = [r for r in result["roofline_env"] if r["kernel"] == "multi_rotate" and r["units"] == 1 << 24][0]
        # north_star's env target (>= 50 % of the HBM roofline for multi_rotate) next to the dominant kernel, where the driver keeps it
        result["roofline"]["env_multi_rotate_2p24"] = {k: mr[k] for k in ("bound", "achieved", "peak", "frac", "traffic", "algorithmic_bytes", "ms")}
        summary["multi_rotate_hbm_frac_2p24"] = mr["frac"]
    if astar and "roofline" in astar.get("f32s", {}):
        result["roofline"]["astar_dominant_kernel"] = {k: astar["f32s"]["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "ms_per_launch")}
    if adi and "roofline" in adi.get("f32s", {}):
        result["roofline"]["adi_dominant_kernel"] = {k: adi["f32s"]["roofline"][k] for k in ("kernel", "achieved", "peak", "frac", "ms_per_launch")}
        result["roofline"]["adi_env"] = {k: adi["f32s"]["roofline_env"][k] for k in ("bound", "achieved", "peak", "frac", "algorithmic_bytes", "ms")}
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(model, args.depth)
        if adi:
            result["cpu_baseline"]["adi"] = cpu_adi(model)
    emit(result, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
