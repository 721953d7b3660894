"""
bench.py -- MCTS node expansions/sec on depth-20 scrambles (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1 024 depth-20 scrambles per GPU (np.random.seed(0), the
reference's scramble stream), MCTS c = 0.6 with graph search, fc_small policy/value net with
glorot weights from torch.manual_seed(0) (no trained weights exist offline), bf16 inference engine.
A "step" is one lock-step MCTS iteration of every tree on the rank: expand 12 children per leaf
(HIP), input layer fused with the one-hot encoding (HIP), remaining network GEMMs (PyTorch-ROCm / hipBLASLt
MFMA) on the NEW children only (11 packed rows per tree), backup + PUCT descent (HIP), replayed as one HIP graph.  value = unique states inserted into the trees by all ranks during the K
timed steps / max-over-ranks wall time (inputs resident in HBM before the timed region).
Ranks own disjoint scramble slices (weak scaling); the only collectives are the barrier, the
max/sum reductions of the result and one all_gather of per-tree node counts.

Also printed on the same JSON line:
  roofline      dominant kernel group of the timed step = the network's GEMMs (MFMA bound)
  roofline_env  the hand-written environment kernels in isolation at 2^24 states (HBM bound;
                multi_rotate is the north_star's roofline target)
  phases        per-phase milliseconds of one MCTS step (HIP events, eager replay of the same step)
  cpu_baseline  the restated reference agent (oracle/, NumPy + torch CPU) on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3


def event_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    return float(np.mean(ts)), float(np.min(ts))


def env_roofline(log2n=24):
    """Environment kernels alone, HBM-resident inputs, HIP events on the launch stream."""
    from librubiks.cube import DeviceCubes
    n = 1 << log2n
    g = torch.Generator(device="cuda").manual_seed(0)
    cubes = DeviceCubes.solved(n)
    for _ in range(30):   # states 30 random moves from solved
        cubes = cubes.multi_rotate(torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g))
    act = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g)
    out = DeviceCubes.empty(n)
    res = []
    # HBM bytes per launch from the committed rocprofv3 PMC passes of these same launches (FETCH_SIZE and
    # WRITE_SIZE in separate runs, gfx950 corrections applied: tools/summarize_pmc.py); None if absent
    pmc_path = os.path.join(ROOT, "profiles", "r1b_env_pmc_traffic.json")
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) and log2n == 24 else {}

    def add(kernel, unit, unit_bytes, units, fn, reps=20):
        mean, best = event_ms(fn, reps)
        gbps = unit_bytes * units / (mean * 1e-3) / 1e9
        res.append({"kernel": kernel, "bound": "hbm", "units": units, "unit": unit, "bytes_per_unit": unit_bytes,
                    "ms": round(mean, 4), "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit_rate": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 4), "Munits_per_s": round(units / (mean * 1e-3) / 1e6, 1),
                    "algorithmic_bytes": int(unit_bytes * units),
                    "traffic": pmc.get(kernel.split("(")[0] if kernel.startswith("is_solved") else kernel, {}).get("traffic_bytes")})

    add("multi_rotate", "state", 41, n, lambda: cubes.multi_rotate(act, out=out))
    npar = n // 4
    parents = DeviceCubes(cubes.soa[:, :npar].contiguous(), npar)
    kids = DeviceCubes.empty(12 * npar)
    add("expand12", "parent", 260, npar, lambda: parents.expand12(out=kids))
    del kids
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    from librubiks import _hip
    lib = _hip.lib()
    add("is_solved(flags)", "state", 21, n,
        lambda: _hip.check(lib.rc_is_solved(cubes.soa.data_ptr(), flags.data_ptr(), None, None, n, cubes.stride,
                                            _hip.stream_ptr())))
    mask = torch.zeros(n // 64 + 2, dtype=torch.int64, device="cuda")
    add("is_solved(mask)", "state", 20.125, n,
        lambda: _hip.check(lib.rc_is_solved(cubes.soa.data_ptr(), None, mask.data_ptr(), None, n, cubes.stride,
                                            _hip.stream_ptr())))
    noh = n // 16
    small = DeviceCubes(cubes.soa[:, :noh].contiguous(), noh)
    oh = torch.empty((noh, 480), dtype=torch.float32, device="cuda")
    add("as_oh(f32)", "state", 1940, noh, lambda: small.as_oh(out=oh))
    oh = torch.empty((noh, 480), dtype=torch.bfloat16, device="cuda")
    add("as_oh(bf16)", "state", 980, noh, lambda: small.as_oh(out=oh))
    return res


def phase_times(forest, c, max_states, reps):
    """Per-phase HIP-event timing of the eager step (same launches the captured graph replays)."""
    import ctypes
    from librubiks import _hip
    lib, m = forest.lib, ctypes.byref(forest.struct)
    names = ["expand", "input_layer", "net_forward", "softmax+copy", "backup", "select"]
    acc = {k: 0.0 for k in names}
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        st = _hip.stream_ptr()
        ev[0].record()
        _hip.check(lib.rc_mcts_expand(m, max_states, st))
        ev[1].record()
        cubes, rows = forest._net_input()
        if forest._fused:
            x1 = forest.engine.first_layer(cubes, forest._x1[:rows])
        else:
            cubes.as_oh(out=forest._oh[:rows])
        ev[2].record()
        if forest._fused:   # same calls as MCTSForest._iteration -> InferenceNet.head_cubes, split at the input layer
            eng = forest.engine
            if eng._fused_head_ok():
                x = eng._run(eng.layers[1:-2], x1)
                raw = torch.addmm(eng.layers[-2][1], x, eng.layers[-2][0].t())
                ev_h = torch.cuda.Event(enable_timing=True)
                ev_h.record()
                head = eng.head_from_raw(raw)
            else:
                head = eng._run(eng.layers[1:], x1)
                ev_h = None
        else:
            logits, values = forest.engine(forest._oh[:rows])
            ev_h = None
        ev[3].record()
        if not forest._fused:
            torch.softmax(logits, dim=1, out=forest.probs[:rows])
            forest.values[:rows].copy_(values)
        ev[4].record()
        if forest._fused:   # softmax + value extraction happen inside the backup kernel
            _hip.check(lib.rc_mcts_backup_head(m, head.data_ptr(), head.stride(0), int(head.dtype == torch.bfloat16), st))
        else:
            _hip.check(lib.rc_mcts_backup(m, forest.probs.data_ptr(), forest.values.data_ptr(), st))
        ev[5].record()
        _hip.check(lib.rc_mcts_select(m, c, forest.level_budget, st))
        ev[6].record()
        torch.cuda.synchronize()
        for i, k in enumerate(names):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
        if ev_h is not None:
            acc["head_kernel"] = acc.get("head_kernel", 0.0) + ev_h.elapsed_time(ev[3])
    out = {k: round(v / reps, 4) for k, v in acc.items()}
    if forest._fused:   # the dominant single kernel by itself: the first hidden GEMM (hipBLASLt, bf16 MFMA)
        eng = forest.engine
        W, b, _ = eng.layers[1]
        cubes, rows = forest._net_input()
        x1 = forest._x1[:rows]
        torch.addmm(b, x1, W.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.addmm(b, x1, W.t())
        e1.record()
        torch.cuda.synchronize()
        out["gemm_hidden1"] = round(e0.elapsed_time(e1) / reps, 4)
    return out


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
            "env_ops": cpu_env_ops(), "bfs_config1": cpu_bfs_config1()}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--trees", type=int, default=1024, help="MCTS trees (scrambles) per GPU")
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-roofline", action="store_true")
    ap.add_argument("--phase-reps", type=int, default=20)
    ap.add_argument("--weights", default=os.path.join(ROOT, "weights", "fc_small_r1"),
                    help="checkpoint directory (model.pt + config.json); random-init weights if it does not exist")
    ap.add_argument("--solve-max-states", type=int, default=50000,
                    help="after the timed steps, search the same scrambles to completion with this per-tree cap and "
                         "report the solve rate (0 = skip)")
    ap.add_argument("--level-budget", type=int, default=0,
                    help="new tree levels a PUCT descent may walk per step before it is suspended (0 = strict lock step)")
    ap.add_argument("--first-layer-table", default="auto", choices=["auto", "f16", "f16pair", "bf16", "mfma", "mfma16", "onehot"],
                    help="input layer: fused gather-sum with an f16 / bf16 table, or the one-hot GEMM")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one process per GPU; RUBIKS_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    backend = os.environ.get("RUBIKS_DIST_BACKEND", "nccl")   # nccl == RCCL on ROCm
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from librubiks import cube
    from librubiks.cube import DeviceCubes
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    from librubiks.solving.sharding import gather_results, shard_range

    # ---- synthetic inputs: the reference's scramble stream, sliced by rank ------------------------
    np.random.seed(0)
    total = args.trees * world
    all_cubes, _, _ = cube.scramble_batch(total, args.depth, True)
    lo, hi = shard_range(total, rank, world)
    roots = DeviceCubes.empty(hi - lo)
    roots.soa[:, :hi - lo] = all_cubes.soa[:, lo:hi]
    del all_cubes

    torch.manual_seed(0)
    if os.path.isdir(args.weights):
        model = Model.load(args.weights).eval()
        weights_note = f"{os.path.relpath(args.weights, ROOT)} (ADI-trained on one MI355X by tools/train_eval.py; see weights/README.md)"
    else:
        model = Model.create(ModelConfig()).eval()
        weights_note = "random-init (glorot, torch.manual_seed(0))"
    net_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    c = 0.6
    capacity = 12 * (args.warmup + args.steps + args.phase_reps + 8) + 64
    from librubiks.model import InferenceNet
    engine = InferenceNet(model, dtype=net_dtype, first_layer_table=args.first_layer_table)
    agent = MCTS(engine, c=c, search_graph=True, net_dtype=net_dtype, level_budget=args.level_budget)
    forest = agent._forest_for(roots.n, capacity)
    max_states = forest.C
    forest.reset(roots)

    # ---- warm-up (first step runs eagerly and captures the HIP graph), then the timed region ------
    for _ in range(max(args.warmup, 1)):
        forest.step(c, max_states, use_graph=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    nodes0 = int(forest.n_nodes.sum().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        forest.step(c, max_states, use_graph=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    nodes = int(forest.n_nodes.sum().item()) - nodes0
    mean_path = float(forest.path_len.float().mean().item())

    stats = torch.tensor([seconds, float(nodes)], dtype=torch.float64, device=coll_device)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        nsum = stats[1:].clone()
        dist.all_reduce(nsum, op=dist.ReduceOp.SUM)
        seconds, nodes = float(tmax.item()), int(nsum.item())
    # ---- solve rate: the same scrambles searched to completion (untimed for `value`) -----------------------
    if args.solve_max_states:
        phases_early = phase_times(forest, c, max_states, args.phase_reps) if (args.phase_reps and rank == 0) else {}
        rows_early, eng_early, fused_early = forest.rows_per_tree * roots.n, forest.engine, forest._fused
        del forest
        agent.forest = None
        torch.cuda.empty_cache()
        t_solve = time.perf_counter()
        full = agent.search_batch(roots, None, args.solve_max_states)
        solve_seconds = time.perf_counter() - t_solve
        local = {"nodes": full.nodes, "solved": full.solved, "lengths": full.lengths}
    else:
        status = forest.status.cpu().numpy()
        local = {"nodes": forest.n_nodes.cpu().numpy(), "solved": status == 1, "lengths": np.full(hi - lo, -1)}
    # final aggregation of per-tree results: the one data collective of an evaluation run (RCCL all_gather)
    gathered = gather_results(local, total, device=coll_device)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.solve_max_states:
        rows, phases, eng, fused = rows_early, phases_early, eng_early, fused_early
    else:
        rows = forest.rows_per_tree * roots.n
        phases = phase_times(forest, c, max_states, args.phase_reps) if args.phase_reps else {}
        eng, fused = forest.engine, forest._fused
    peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    roofline = roofline_group = roofline_input = None
    if phases:
        # GEMM group actually executed on MFMA: every layer when the input is a one-hot matrix, layers 2..
        # when the input layer is the fused gather-sum kernel (which is HBM/LDS work, reported separately)
        gemm_layers = eng.layers[1:] if fused else eng.layers
        flops = 2 * sum(W.shape[0] * W.shape[1] for W, _, _ in gemm_layers) * rows
        tf = flops / (phases["net_forward"] * 1e-3) / 1e12
        roofline_group = {"kernel": f"policy/value net GEMMs on {rows} child rows ({len(gemm_layers)} hipBLASLt GEMMs + bias + "
                                    "ELU passes, BatchNorm folded, heads merged)",
                          "bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                          "frac": round(tf / peak, 4), "traffic": None,
                          "flops_per_launch": flops, "ms_per_launch": phases["net_forward"]}
        roofline = roofline_group
        if "gemm_hidden1" in phases:   # the dominant kernel of the step, alone
            W1 = eng.layers[1][0]
            f1 = 2 * W1.shape[0] * W1.shape[1] * rows
            tf1 = f1 / (phases["gemm_hidden1"] * 1e-3) / 1e12
            roofline = {"kernel": f"hidden GEMM [{rows} x {W1.shape[1]}] x [{W1.shape[1]} x {W1.shape[0]}] + bias, bf16 MFMA via hipBLASLt "
                                  "(Cijk_..._MT256x192x64 in the rocprof summary): the dominant kernel of a step",
                        "bound": "mfma", "achieved": round(tf1, 1), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(tf1 / peak, 4), "traffic": None, "flops_per_launch": f1,
                        "ms_per_launch": phases["gemm_hidden1"]}
        if fused:
            H = eng._fused_first[4]
            nbytes = (20 + 2 * H) * rows
            gbps = nbytes / (phases["input_layer"] * 1e-3) / 1e9
            mode = eng._fused_first[5]
            name = ("rc_first_layer_mfma_bf16 (one-hot x W1 on the matrix cores, one-hot fragments generated from the cube codes, "
                    "W1 slice in LDS, + bias + ELU)") if mode in (2, 4) else \
                "rc_first_layer_bf16 (one-hot x W1 as 20-row gather-sum from LDS + bias + ELU)"
            roofline_input = {"kernel": name, "table": {0: "bf16", 1: "f16", 2: "bf16", 3: "f16 pairs", 4: "f16"}[mode],
                              "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(gbps / HBM_PEAK_GBPS, 4), "algorithmic_bytes": nbytes,
                              "bytes_per_unit": 20 + 2 * H, "traffic": None, "ms_per_launch": phases["input_layer"],
                              "note": "algorithmic bytes: 20 B of cube codes in, 2 H B of activations out per row; the kernel is "
                                      "LDS-feed / issue bound, not HBM bound (DESIGN.md section 3)"}
    result = {
        "metric": "MCTS node expansions/sec, depth-20 scrambles", "value": round(nodes / seconds, 1),
        "unit": "node expansions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(seconds / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.trees} depth-{args.depth} scrambles per GPU, MCTS agent (c=0.6, graph search), "
                               f"fc_small net, weights: {weights_note}", "trees_per_gpu": args.trees, "select_level_budget": args.level_budget,
                   "scramble_depth": args.depth, "net_rows_per_step": rows, "parallelism": f"scramble-sharded x{world}"},
        "nodes_expanded": nodes, "mean_descent_depth": round(mean_path, 2), "solve_rate": float(np.mean(gathered["solved"])),
        "solve_run": ({"max_states_per_tree": args.solve_max_states, "games": int(total),
                       "ci95": float(1.959963984540054 * np.sqrt(np.mean(gathered["solved"]) * (1 - np.mean(gathered["solved"])) / total)),
                       "mean_solution_length": float(np.mean(gathered["lengths"][gathered["solved"].astype(bool)]))
                       if np.any(gathered["solved"]) else None,
                       "nodes": int(np.sum(gathered["nodes"])), "seconds_rank0": round(solve_seconds, 2),
                       "note": "same scrambles searched to completion after the timed steps; not part of `value`"}
                      if args.solve_max_states else None),
        "roofline": roofline, "roofline_net_group": roofline_group, "roofline_input_layer": roofline_input, "phases_ms": phases,
    }
    if not args.no_env_roofline and world == 1:
        agent.forest = None
        torch.cuda.empty_cache()
        result["roofline_env"] = env_roofline()
    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(model, args.depth)
    print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
