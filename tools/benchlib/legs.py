"""The timed workloads of bench.py: the MCTS legs (BASELINE configs[1] and the config-5 share), A* (configs[2]) and the ADI rollout (configs[3])."""
from .common import *   # noqa: F401,F403  (json, os, sys, time, np, torch, dist, ROOT, the roofline constants, progress, event_ms)


def replay_solutions(roots_np, res, what):
    """
    Every game reported solved: its action queue has the reported length and, walked from the game's scramble through the
    library's own `cube.multi_rotate` (one call per move index over the games still moving), ends on the solved state
    (`cube.multi_is_solved`).  Outside every timed region.  A mismatch ends the benchmark: a solve rate is checked, not reported.
    """
    from librubiks import cube
    idx = np.flatnonzero(np.asarray(res.solved))
    if not len(idx):
        return {"games_reported_solved": 0, "solutions_replayed_to_solved": 0}
    if hasattr(res.queues, "padded"):
        acts, lens = res.queues.padded(idx)
    else:
        lens = np.array([len(res.queues[i]) for i in idx])
        acts = np.full((len(idx), int(lens.max())), 255, dtype=np.uint8)
        for o, i in enumerate(idx):
            acts[o, :lens[o]] = list(res.queues[i])
    if not np.array_equal(lens, np.asarray(res.lengths)[idx]):
        raise RuntimeError(f"{what}: a reported solution length is not its action queue's")
    order = np.argsort(-lens, kind="stable")           # longest first: the games still moving at move d are a prefix
    acts, lens, cur = acts[order], lens[order], np.ascontiguousarray(roots_np[idx][order]).copy()
    for d in range(int(lens.max())):
        n_live = int(np.searchsorted(-lens, -d, side="left"))      # games with more than d moves
        faces, dirs = cube.indices_to_actions(acts[:n_live, d].astype(np.int64))
        cur[:n_live] = cube.multi_rotate(cur[:n_live], faces, dirs)
    ok = int(np.asarray(cube.multi_is_solved(cur)).sum())
    if ok != len(idx):
        raise RuntimeError(f"{what}: {len(idx) - ok} of {len(idx)} reported solutions do not end on the solved state")
    return {"games_reported_solved": int(len(idx)), "solutions_replayed_to_solved": ok}


def run_leg(name, model, pool_roots, config_roots, args, world, coll_device, trees, cap, window_only=False, full_warm=True):
    """
    One network precision: steady-state window of K steps on the continuously refilled pool, the whole pool to
    completion, and the first `trees` scrambles as one batch to completion (BASELINE configs[1] when trees = 1 024).
    window_only: stop after the timed window.  full_warm: the untimed warm-up of the run to completion is the same search run
    once before (every launch size's HIP graph is then in the forest's cache, as in any evaluator that searches more than one
    batch); otherwise 30 iterations (the first graph only; the others are captured inside the timed run).
    Returns (dict for the JSON line, engine, agent).
    """
    from librubiks.model import F32_SPLIT, InferenceNet, SplitF32Net
    from librubiks.solving.agents import MCTS
    net_dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f32s": F32_SPLIT}[name]
    engine = SplitF32Net(model) if name == "f32s" else InferenceNet(model, dtype=net_dtype, first_layer_table=args.first_layer_table)
    agent = MCTS(engine, c=0.6, search_graph=True, net_dtype=net_dtype, level_budget=args.level_budget)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- pool: untimed prep, warm-up, timed window, rest of the pool ------------------------------------
    # one-off set-up, untimed: forest allocation (tens of GB of HBM, zero-filled), engine, one HIP graph per launch size
    t_prep = time.perf_counter()
    agent.prepare(trees, cap)
    barrier()
    prepare_seconds = time.perf_counter() - t_prep
    t_pool = time.perf_counter()
    run = agent.start_batch(pool_roots, None, cap, slots=trees)
    # Prep: until as many scrambles again as there are slots have been started (the slots then hold trees of every age).  Where
    # the window falls relative to the flushes of the results forest (graph completion + BFS of 256 finished trees on a side
    # stream, ~35 ms of kernels every 256 finished games that slow the concurrent steps by 10-20 %) is NOT chosen:
    # `result_flushes_in_window` says what it saw, pool_run contains all of them.
    prep_games = 2 * trees
    while not run.done and run.next_game < min(prep_games, run.n_games) and run.it < args.prep_cap:
        run.round()
    # Harvested trees are turned into host results lazily; doing that here (tens of ms of host work, the GPU idles and drops its
    # clocks) instead of inside nodes_now() right in front of the timed window, then two more untimed rounds to bring the clocks
    # back: the first ~10 steps after such a pause were measured 5-30 % slow (round 3 probe: profiles/README.md).
    run.nodes_now()
    for _ in range(2):
        if not run.done:
            run.round()
    prep_iters = run.it
    left = max(args.warmup, 1)
    while left > 0 and not run.done:
        before = run.it
        run.round(left)
        left -= run.it - before
    barrier()
    nodes0, refills0, it0, flushes0 = run.nodes_now(), run.stats["refills"], run.it, run.stats.get("flushes", 0)
    barrier()
    t0 = time.perf_counter()
    left = args.steps
    while left > 0 and not run.done:
        before = run.it
        run.round(left)
        left -= run.it - before
    barrier()
    seconds = time.perf_counter() - t0
    nodes = run.nodes_now() - nodes0
    steps_done = run.it - it0
    status = run.forest.status.cpu().numpy()
    running_in_window = int(((status == 0) & (run.owner >= 0)).sum())
    plen = run.forest.path_len.cpu().numpy()
    mean_path = float(plen[(status == 0) & (run.owner >= 0)].mean()) if running_in_window else 0.0
    refills_in_window = run.stats["refills"] - refills0
    flushes_in_window = run.stats.get("flushes", 0) - flushes0
    # Five more windows of K steps right behind the timed one (same bracket): how far one K-step window of this run is from the next,
    # so that a change of a few per cent between two runs can be told from the window's own scatter (value_spread: median, min, max).
    more = torch.zeros((SPREAD_WINDOWS, 2), dtype=torch.float64)
    for w in range(SPREAD_WINDOWS):
        over = torch.tensor([1.0 if run.done else 0.0], dtype=torch.float64, device=coll_device)
        if world > 1:      # every rank runs the same number of windows (each ends in a collective): stop together when any pool has run dry
            dist.all_reduce(over, op=dist.ReduceOp.MAX)
        if float(over.item()):
            break
        n_before = run.nodes_now()
        barrier()
        tw = time.perf_counter()
        left = args.steps
        while left > 0 and not run.done:
            before = run.it
            run.round(left)
            left -= run.it - before
        barrier()
        more[w, 0] = time.perf_counter() - tw
        more[w, 1] = run.nodes_now() - n_before
    pool = None
    if not window_only:
        while not run.done:
            run.round()
        res = run.finish()
        torch.cuda.synchronize()
        pool_seconds = time.perf_counter() - t_pool
        pool_check = replay_solutions(pool_roots.numpy(), res, f"{name} pool run")
        pool = {"games": int(run.n_games), "slots": trees, "nodes": int(res.nodes.sum()), "seconds": round(pool_seconds, 3), **pool_check,
                "nodes_per_sec": round(float(res.nodes.sum()) / pool_seconds, 1), "solve_rate": float(res.solved.mean()),
                "path_overflow_trees": res.path_overflow_trees, "iterations": int(run.it), **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in run.stats.items() if k != "iterations"}}
    del run
    # ---- the first `trees` scrambles as one batch, to completion (BASELINE configs[1]) ------------------------
    rtc = local = None
    if not window_only:
        if full_warm:
            agent.search_batch(config_roots, None, cap)                       # untimed: the same search once before
        else:
            agent.search_batch(config_roots, None, cap, max_iterations=30)   # untimed: 30 iterations (clocks, library heuristics)
        barrier()
        t1 = time.perf_counter()
        full = agent.search_batch(config_roots, None, cap)
        torch.cuda.synchronize()
        rtc_seconds = time.perf_counter() - t1
        local = {"nodes": full.nodes, "solved": full.solved, "lengths": full.lengths}
        rtc = {"seconds": rtc_seconds, "nodes": int(full.nodes.sum()), "iterations": int(full.iterations.max()),
               "path_overflow_trees": full.path_overflow_trees,
               "launch_sizes": int(agent.refill_stats.get("compactions", 0)) + 1,
               "check": replay_solutions(config_roots.numpy(), full, f"{name} run to completion")}
    forest_gb = {"hbm_behind_the_forest_gb": round(agent.forest.bytes_allocated() / 1e9, 2), "mapped_on_demand": bool(agent.forest.vmm),
                 "node_rows_reserved_gb": round(agent.forest.bytes_reserved() / 1e9, 2)}
    stats = torch.tensor([seconds, float(nodes), float(steps_done), rtc["seconds"] if rtc else 0.0,
                          float(pool["nodes"]) if pool else 0.0, float(pool["seconds"]) if pool else 0.0],
                         dtype=torch.float64, device=coll_device)
    rank_values = [round(nodes / seconds, 1)]           # every rank's own window: its nodes / its seconds
    more = more.to(coll_device)
    if world > 1:       # a window of the job: all ranks' nodes / the slowest rank's seconds
        sec, nod = more[:, 0].clone(), more[:, 1].clone()
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
        dist.all_reduce(nod, op=dist.ReduceOp.SUM)
        more = torch.stack([sec, nod], 1)
    windows = sorted(float(n / t) for t, n in more.cpu().tolist() if t > 0 and n > 0)
    spread = {"windows": len(windows), "median": round(windows[len(windows) // 2], 1), "min": round(windows[0], 1), "max": round(windows[-1], 1),
              "note": f"{len(windows)} further windows of K steps right behind the timed one"} if windows else None
    if world > 1:
        every = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(every, stats)
        rank_values = [round(float(e[1]) / float(e[0]), 1) for e in every]
        mx, sm = stats.clone(), stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        seconds, nodes, rtc_s = float(mx[0]), int(sm[1]), float(mx[3])
        pool_nodes, pool_s = int(sm[4]), float(mx[5])
    else:
        rtc_s, pool_nodes, pool_s = float(stats[3]), int(stats[4]), float(stats[5])
    out = {"dtype": LEG_DTYPE[name], "value": round(nodes / seconds, 1), "ms_per_step": round(seconds / max(steps_done, 1) * 1e3, 4),
           "nodes_in_window": nodes, "steps_timed": steps_done, "prep_iterations_untimed": prep_iters, "result_flushes_in_window": flushes_in_window,
           "refills_in_window": refills_in_window, "running_trees_rank0": running_in_window,
           "mean_descent_depth_rank0": round(mean_path, 1), "max_states_per_tree": cap,
           "prepare_seconds_rank0": round(prepare_seconds, 3), "forest_rank0": forest_gb, "rank_values": rank_values, "value_spread": spread}
    if pool:
        out["pool_run"] = dict(pool, nodes=pool_nodes, seconds=round(pool_s, 3), nodes_per_sec=round(pool_nodes / pool_s, 1),
                               games=int(pool["games"]) * world,
                               note="whole pool searched to completion on `slots` tree slots; wall time includes prep, window and tail "
                                    "(not the one-off set-up before: forest allocation, HIP graph capture per launch size)")
    if rtc:
        from librubiks.solving.sharding import gather_results
        total = trees * world
        g = gather_results(local, total, device=coll_device)
        p = float(np.mean(g["solved"]))
        out["run_to_completion"] = {
            "games": int(total), "max_states_per_tree": cap, "nodes": int(np.sum(g["nodes"])), "seconds": round(rtc_s, 3),
            "nodes_per_sec": round(float(np.sum(g["nodes"])) / rtc_s, 1), "solve_rate": p,
            "ci95": float(1.959963984540054 * np.sqrt(p * (1 - p) / total)),
            "mean_solution_length": float(np.mean(g["lengths"][g["solved"].astype(bool)])) if np.any(g["solved"]) else None,
            "lock_step_iterations_rank0": rtc["iterations"], "launch_sizes_rank0": rtc["launch_sizes"],
            "path_overflow_trees_rank0": rtc["path_overflow_trees"],
            **{k + "_rank0": v for k, v in rtc["check"].items()},
            "seconds_incl_prepare_rank0": round(rtc["seconds"] + prepare_seconds, 3),
            "warm_up": "forest allocated and HIP graphs of every launch size captured before (MCTS.prepare), then "
                       + ("the same search once, untimed" if full_warm else "30 iterations of it, untimed"),
            "note": "the scrambles as ONE batch: sum len(agent) / wall seconds of the batched search (SURVEY 8(d)(i))"}
    return out, engine, agent


def astar_leg(name, model, roots, args, world, coll_device):
    """
    BASELINE configs[2]: `roots.n` depth-20 scrambles per GPU, batch weighted A* with the reference's defaults lambda = 0.2,
    N = 100 (runeval.py:60,65).  Times K iterations of all problems (after W warm-up iterations of the same batch), the phases
    of one iteration (HIP events), the dominant kernel alone on the iteration's real operands, and the search to completion
    at max_states = `--solve-max-states`.
    """
    import ctypes
    from librubiks import _hip
    from librubiks.model import F32_SPLIT, SplitF32Net
    from librubiks.solving.agents import AStar
    net_dtype = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[name]
    lam, N, cap = 0.2, 100, args.solve_max_states
    agent = AStar(model, lam, N, net_dtype=net_dtype)
    K, W = max(1, min(args.steps, 12)), max(1, min(args.warmup, 3))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    batch = agent._batch_for(roots.n, max(cap, 12 * N * (K + W + 4) + 16))
    batch.reset(roots)
    for _ in range(W):
        batch.iteration(lam, batch.C)
    barrier()
    n0 = int(batch.n_nodes.sum().item())
    t0 = time.perf_counter()
    for _ in range(K):
        batch.iteration(lam, batch.C)
    barrier()
    seconds = time.perf_counter() - t0
    nodes = int(batch.n_nodes.sum().item()) - n0
    out = {"dtype": LEG_DTYPE[name], "problems_per_gpu": int(roots.n), "lambda": lam, "expansions": N, "iterations_timed": K,
           "warmup_iterations": W, "child_rows_per_iteration": int(roots.n) * N * 12}
    # ---- phases of one more iteration + its dominant kernel (rank 0's view) ------------------------------------
    phases = roof = None
    if args.phase_reps:
        m, st, eng = ctypes.byref(batch.struct), _hip.stream_ptr(), batch.engine
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        _hip.check(batch.lib.rc_astar_pop_expand(m, batch.C, st), "rc_astar_pop_expand")
        ev[1].record()
        torch.cumsum(batch.new_count, 0, dtype=torch.int32, out=batch.new_offset[1:])
        total = int(batch.new_offset[-1].item())
        _hip.check(batch.lib.rc_astar_gather_new(m, batch.new_offset.data_ptr(), batch.new_states.soa.data_ptr(),
                                                 batch.new_states.stride, st), "rc_astar_gather_new")
        ev[2].record()
        batch._values_of_new(total)
        ev[3].record()
        _hip.check(batch.lib.rc_astar_push_relax(m, batch.new_offset.data_ptr(), batch.values.data_ptr(), lam, st), "rc_astar_push_relax")
        ev[4].record()
        torch.cuda.synchronize()
        names = ["pop_expand", "prefix_sum+gather_new", "value_net", "push_relax"]
        phases = {k: round(ev[i].elapsed_time(ev[i + 1]), 4) for i, k in enumerate(names)}
        phases["new_states"] = total
        from librubiks.solving.astar_device import NET_CHUNK
        rows = min(total, NET_CHUNK)
        flops_state = 2 * sum(int(l[1].shape[0]) * int(l[1].shape[1]) for l in eng.value_layers) if isinstance(eng, SplitF32Net) \
            else 2 * sum(int(Wt.shape[0]) * int(Wt.shape[1]) for Wt, _, _ in eng.value_layers)
        mult = 3 if isinstance(eng, SplitF32Net) else 1
        peak = MFMA_BF16_PEAK_TFLOPS
        tf_net = mult * flops_state * total / (phases["value_net"] * 1e-3) / 1e12
        group = {"kernel": f"value network on the {total} new states of one iteration ({'three f16 products per layer' if mult == 3 else 'bf16'})",
                 "bound": "mfma", "achieved": round(tf_net, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf_net / peak, 4),
                 "flops_per_launch": mult * flops_state * total, "ms_per_launch": phases["value_net"], "traffic": None}
        # the dominant kernel alone: the first hidden layer on one chunk of the iteration's real input-layer activations
        if isinstance(eng, SplitF32Net):
            a = eng._first_from_cubes(batch.new_states, eng.value_layers, 0, rows)
            _, Wh, B2, b, code, alpha, W3 = eng.value_layers[1]
            Kd, Nd = int(Wh.shape[1]), int(Wh.shape[0])
            o = torch.empty((rows, 2 * Nd), dtype=torch.float16, device=Wh.device)
            ms = event_ms(lambda: _hip.check(batch.lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), b.data_ptr(), rows, Nd, Kd, code, alpha,
                                                                         o.data_ptr(), None, 0, _hip.stream_ptr()), "rc_split_gemm_f16"), 5)[0]
            fl = 3 * 2 * rows * Nd * Kd
            kname = f"rc_split_gemm_f16 [{rows}x{3 * Kd}]x[{3 * Kd}x{Nd}] f16 MFMA +bias+ELU+re-split: hidden layer 1 of A*'s value network"
        else:
            x1 = eng.first_layer(batch.new_states, None, 0, rows)
            Wt, bt, _ = eng.value_layers[1]
            Kd, Nd = int(Wt.shape[1]), int(Wt.shape[0])
            ms = event_ms(lambda: torch.addmm(bt, x1, Wt.t()), 5)[0]
            fl = 2 * rows * Nd * Kd
            kname = f"hidden GEMM [{rows}x{Kd}]x[{Kd}x{Nd}] + bias, bf16 MFMA via hipBLASLt: hidden layer 1 of A*'s value network"
        roof = {"kernel": kname, "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4), "flops_per_launch": fl, "ms_per_launch": round(ms, 4), "traffic": None,
                "note": "HIP events on the launch stream; the tree-side kernels (pop_expand: per-problem heap pops + 12 children + hash "
                        "dedup + first-occurrence election; push_relax: float64 cost, heap pushes, relaxation) are latency / atomic "
                        "bound, their times are in phases_ms"}
        out["phases_ms"], out["roofline"], out["roofline_net_group"] = phases, roof, group
    # ---- search to completion ---------------------------------------------------------------------------------
    local = None
    if not args.window_only:
        barrier()
        t1 = time.perf_counter()
        res = agent.search_batch(roots, None, cap)
        torch.cuda.synchronize()
        solve_s = time.perf_counter() - t1
        local = {"nodes": res.nodes, "solved": res.solved, "lengths": res.lengths}
        solve_check = replay_solutions(roots.numpy(), res, f"{name} A* solve run")
    stats = torch.tensor([seconds, float(nodes), solve_s if local else 0.0], dtype=torch.float64, device=coll_device)
    if world > 1:
        mx, sm = stats.clone(), stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        seconds, nodes, solve_s = float(mx[0]), int(sm[1]), float(mx[2])
    out.update({"value": round(nodes / seconds, 1), "unit": "new states/s", "ms_per_iteration": round(seconds / K * 1e3, 3),
                "new_states_per_iteration": round(nodes / K / world, 1)})
    if local:
        from librubiks.solving.sharding import gather_results
        total_games = roots.n * world
        g = gather_results(local, total_games, device=coll_device)
        p = float(np.mean(g["solved"]))
        out["solve_run"] = {"games": int(total_games), "max_states_per_problem": cap, "nodes": int(np.sum(g["nodes"])), "seconds": round(solve_s, 3),
                            "states_per_sec": round(float(np.sum(g["nodes"])) / solve_s, 1), "solve_rate": p,
                            "ci95": float(1.959963984540054 * np.sqrt(p * (1 - p) / total_games)),
                            "mean_solution_length": float(np.mean(g["lengths"][g["solved"].astype(bool)])) if np.any(g["solved"]) else None,
                            **{k + "_rank0": v for k, v in solve_check.items()}}
    del agent, batch
    torch.cuda.empty_cache()
    return out


def adi_leg(name, model, args, world, coll_device):
    """
    BASELINE configs[3]: the data generation of one Autodidactic-Iteration rollout (reference train.py:257-339) for a batch of
    `--adi-states` states (512 games x 32 moves = 16 384 -> 196 608 substates), device resident:
        sequence_scrambler -> expand12 -> is_solved (substates, states) -> value network on the substates -> rc_adi_targets -> one-hot of the states
    Timed: K calls of Train.ADI_traindata after W warm-up calls (K, W capped at 20 / 3), barrier + synchronize on both sides;
    value = states of all ranks / max-over-ranks seconds.  With N ranks every rank generates games / N games (the reference's
    data-parallel layout of config #4).  Phases: the same steps once more between HIP events; roofline: the dominant kernel
    (first hidden layer of the value network on the substates' real input-layer activations).
    """
    from librubiks import _hip, cube as pcube
    from librubiks.model import F32_SPLIT, SplitF32Net
    from librubiks.train import Train
    net_dtype = {"bf16": torch.bfloat16, "f32s": F32_SPLIT}[name]
    depth = 32
    games = max(1, args.adi_states // depth // world)
    tr = Train(rollouts=1, batch_size=1000, rollout_games=games, rollout_depth=depth, optim_fn=torch.optim.Adam, alpha_update=0, lr=1e-4,
               gamma=1, update_interval=0, agent=None, evaluator=None, evaluation_interval=0, tau=1, reward_method="lapanfix",
               adi_net_dtype=net_dtype)
    K, W = max(1, min(args.steps, 20)), max(1, min(args.warmup, 3))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    np.random.seed(1234 + int(os.environ.get("RANK", 0)))
    for _ in range(W):
        tr.ADI_traindata(model, 0.5)
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        out = tr.ADI_traindata(model, 0.5)
    barrier()
    seconds = time.perf_counter() - t0
    n = games * depth
    assert out[0].shape == (n, 480) and out[1].shape == (n,)
    stats = torch.tensor([seconds], dtype=torch.float64, device=coll_device)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    seconds = float(stats[0])
    res = {"dtype": LEG_DTYPE[name], "games_per_gpu": games, "depth": depth, "states_per_rollout": n * world, "substates_per_rollout": 12 * n * world,
           "rollouts_timed": K, "warmup_rollouts": W, "value": round(n * world * K / seconds, 1), "unit": "states/s",
           "ms_per_rollout": round(seconds / K * 1e3, 3), "substates_per_sec": round(12 * n * world * K / seconds, 1),
           "reward_method": "lapanfix"}
    if not args.phase_reps:
        return res
    # ---- the same steps between HIP events (rank 0's view), and the dominant kernel alone -------------------------------------
    lib, eng = _hip.lib(), tr._adi_engine(model)
    names = ["sequence_scrambler (host RNG + moves to the device + rc_sequence_states)", "expand12 + is_solved (substates + states), one launch", "flag views",
             "value_net", "rc_adi_targets", "as_oh(states, f32)"]
    acc = np.zeros(len(names))
    reps = 5
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        ev[0].record()
        states = pcube.sequence_scrambler_device(games, depth, with_solved=True)
        ev[1].record()
        kids, state_solved, kid_solved = states.expand12_flags()   # ONE launch: children + both solved tests (as Train.ADI_traindata does)
        ev[2].record()
        kid_solved, state_solved = kid_solved.view(torch.uint8), state_solved.view(torch.uint8)
        ev[3].record()
        values = eng.value_cubes(kids)
        ev[4].record()
        pol = torch.empty(n, dtype=torch.int64, device="cuda")
        val = torch.empty(n, dtype=torch.float32, device="cuda")
        _hip.check(lib.rc_adi_targets(values.data_ptr(), kid_solved.data_ptr(), state_solved.data_ptr(), n, depth, 1.0, 1, pol.data_ptr(),
                                      val.data_ptr(), _hip.stream_ptr()), "rc_adi_targets")
        ev[5].record()
        states.as_oh(torch.float32)
        ev[6].record()
        torch.cuda.synchronize()
        acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(len(names))]
    res["phases_ms"] = {k: round(float(v) / reps, 4) for k, v in zip(names, acc)}
    rows = 12 * n
    env_bytes = 260 * n + 13 * n + 1940 * n      # expand12 with the 13 solved flags per parent written by the same launch + one-hot f32
    env_ms = res["phases_ms"]["expand12 + is_solved (substates + states), one launch"] + res["phases_ms"]["as_oh(states, f32)"]
    # the same two launches as a captured HIP graph, replayed: what the kernels take on the device, without the host's launch path
    # (eager calls from Python cost ~10 us each before the GPU sees them: for 36 MB of traffic that IS the time measured above)
    g_states = pcube.sequence_scrambler_device(games, depth, with_solved=True)
    g_kids = pcube.DeviceCubes.empty(12 * n, g_states.soa.device)
    g_oh = torch.empty((n, 480), dtype=torch.float32, device=g_states.soa.device)
    g_states.expand12_flags(out=g_kids)
    g_states.as_oh(torch.float32, out=g_oh)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_states.expand12_flags(out=g_kids)
        g_states.as_oh(torch.float32, out=g_oh)
    env_graph_ms = event_ms(graph.replay, 20)[0]
    res["phases_ms"]["environment step as a replayed HIP graph (expand12 + flags, as_oh)"] = round(env_graph_ms, 4)
    eager_ms, env_ms = env_ms, env_graph_ms
    res["roofline_env"] = {"kernel": f"expand12 + solved flags of {13 * n} states in one launch ({n} parents) + as_oh f32 ({n} states): the rollout's environment kernels",
                           "ms_eager_from_python": round(eager_ms, 4),
                           "bound": "hbm", "algorithmic_bytes": int(env_bytes), "ms": round(env_ms, 4), "achieved": round(env_bytes / (env_ms * 1e-3) / 1e9, 1),
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(env_bytes / (env_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                           "note": "the two launches replayed as one HIP graph (device time); called eagerly from Python they take ms_eager_from_python, the host's launch path. "
                                   "A rollout's arrays are a few MB: far from the HBM roof either way (roofline_env of the main line has the kernels at 2^14 .. 2^26 states)"}
    if isinstance(eng, SplitF32Net):
        a = eng._first_from_cubes(kids, eng.value_layers, 0, rows)
        _, Wh, B2, b, code, alpha, W3 = eng.value_layers[1]
        Kd, Nd = int(Wh.shape[1]), int(Wh.shape[0])
        o = torch.empty((rows, 2 * Nd), dtype=torch.float16, device=Wh.device)
        ms = event_ms(lambda: _hip.check(lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), b.data_ptr(), rows, Nd, Kd, code, alpha, o.data_ptr(), None, 0,
                                                               _hip.stream_ptr()), "rc_split_gemm_f16"), 5)[0]
        fl = 3 * 2 * rows * Nd * Kd
        kname = f"rc_split_gemm_f16 [{rows}x{3 * Kd}]x[{3 * Kd}x{Nd}] f16 MFMA +bias+ELU+re-split: hidden layer 1 of the ADI value network"
        flops_state = 3 * 2 * sum(int(l[1].shape[0]) * int(l[1].shape[1]) for l in eng.value_layers)
    else:
        x1 = eng.first_layer(kids, None, 0, rows)
        Wt, bt, _ = eng.value_layers[1]
        Kd, Nd = int(Wt.shape[1]), int(Wt.shape[0])
        ms = event_ms(lambda: torch.addmm(bt, x1, Wt.t()), 5)[0]
        fl = 2 * rows * Nd * Kd
        kname = f"hidden GEMM [{rows}x{Kd}]x[{Kd}x{Nd}] + bias, bf16 MFMA via hipBLASLt: hidden layer 1 of the ADI value network"
        flops_state = 2 * sum(int(Wt.shape[0]) * int(Wt.shape[1]) for Wt, _, _ in eng.value_layers)
    tf = fl / (ms * 1e-3) / 1e12
    res["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "flops_per_launch": fl, "ms_per_launch": round(ms, 4), "traffic": None}
    tfn = flops_state * rows / (res["phases_ms"]["value_net"] * 1e-3) / 1e12
    res["roofline_net_group"] = {"kernel": f"value network on the {rows} substates of a rollout", "bound": "mfma", "achieved": round(tfn, 1),
                                 "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tfn / MFMA_BF16_PEAK_TFLOPS, 4),
                                 "flops_per_launch": flops_state * rows, "ms_per_launch": res["phases_ms"]["value_net"], "traffic": None}
    return res


def release_node_stores():
    """Between leg families: the address ranges and memory that finished forests have left for a successor of their shape
    (librubiks/_vmm.py) are given back, so the next family starts from an empty card."""
    from librubiks._vmm import VmmArray
    torch.cuda.synchronize()
    VmmArray.trim()
    torch.cuda.empty_cache()


def draw_scrambles(n_config, n_pool, depth, slice_rank, slice_world):
    """
    Synthetic inputs of one rank.  The first n_config * world games are the reference's scramble stream (np.random.seed(0),
    scramble(depth, True) game after game): rank r owns games [r n_config, (r + 1) n_config) and replays only
    those n_config * world draws.  The rest of a rank's pool (n_pool - n_config scrambles that merely keep the slots busy) comes
    from a stream of the rank's own (seed 1 000 003 + rank), so no rank draws another rank's pool.
    Returns (config_roots, pool_roots) as DeviceCubes; the pool starts with the rank's config scrambles.
    """
    from librubiks import cube
    from librubiks.cube import DeviceCubes
    from librubiks.solving.sharding import shard_range
    np.random.seed(0)
    all_cubes, _, _ = cube.scramble_batch(n_config * slice_world, depth, True)
    lo, hi = shard_range(n_config * slice_world, slice_rank, slice_world)
    config_roots = DeviceCubes.empty(hi - lo)
    config_roots.soa[:, :hi - lo] = all_cubes.soa[:, lo:hi]
    pool_roots = DeviceCubes.empty(n_pool)
    pool_roots.soa[:, :hi - lo] = config_roots.soa[:, :hi - lo]
    if n_pool > hi - lo:
        np.random.seed(1_000_003 + slice_rank)
        rest, _, _ = cube.scramble_batch(n_pool - (hi - lo), depth, True)
        pool_roots.soa[:, hi - lo:n_pool] = rest.soa[:, :n_pool - (hi - lo)]
    return config_roots, pool_roots
