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
# `traffic` of roofline_env is NOT measured inside this run: it is the stored figure of separate rocprofv3 --pmc passes
# (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied by tools/rocprof_summary.py traffic) of the same launches at 2^24 states
_PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
# the most recent measurement of the kernels as they are in this tree (re-taken every round the kernels change: tools/r6_pmc_pass.sh)
PMC_FILE = next((f for f in ("r6_env_pmc_traffic.json", "r4k_env_pmc_traffic.json") if os.path.exists(os.path.join(_PROFILES, f))), "r6_env_pmc_traffic.json")
GEMM_PMC_FILE = next((f for f in ("r6_split_gemm_traffic.json", "r4_split_gemm_traffic.json") if os.path.exists(os.path.join(_PROFILES, f))),
                     "r6_split_gemm_traffic.json")   # the hidden-layer kernel that runs today (k_split_gemm<2, 4, 11, 4, 2, 0>)
PMC_SOURCE = f"stored PMC figure: profiles/{PMC_FILE} (separate rocprofv3 --pmc passes of these launches, not this run)"


_T0 = time.perf_counter()


def progress(what):
    """One short line per finished leg on stderr: where a run was when something went wrong, and a sign of life for the box's
    silence watchdog.  (The final stdout line stays the last thing printed.)"""
    free, total = torch.cuda.mem_get_info()
    print(f"[bench {time.perf_counter() - _T0:6.1f} s] {what}; HBM in use {(total - free) / 1e9:.0f} GB", file=sys.stderr, flush=True)


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
    """Environment kernels alone, HBM-resident inputs, HIP events on the launch stream (N = 2^14 .. 2^26)."""
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
    # WRITE_SIZE in separate runs, gfx950 corrections applied: tools/rocprof_summary.py traffic); None if absent
    pmc_path = os.path.join(ROOT, "profiles", PMC_FILE)
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) and log2n == 24 else {}

    def add(kernel, unit, unit_bytes, units, fn, reps=20):
        mean, best = event_ms(fn, reps)
        gbps = unit_bytes * units / (mean * 1e-3) / 1e9
        res.append({"kernel": kernel, "bound": "hbm", "units": units, "unit": unit, "bytes_per_unit": unit_bytes,
                    "ms": round(mean, 4), "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit_rate": "GB/s",
                    "frac": round(gbps / HBM_PEAK_GBPS, 4), "Munits_per_s": round(units / (mean * 1e-3) / 1e6, 1),
                    "algorithmic_bytes": int(unit_bytes * units),
                    "traffic": pmc.get(kernel.split("(")[0] if kernel.startswith("is_solved") else kernel, {}).get("traffic_bytes"),
                    "traffic_source": (PMC_SOURCE if pmc else None)})

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
    from librubiks.model import InferenceNet, SplitF32Net
    lib, m = forest.lib, ctypes.byref(forest.struct)
    if isinstance(forest.engine, SplitF32Net):
        return phase_times_split(forest, c, max_states, reps)
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
    if isinstance(forest.engine, InferenceNet):   # the dominant single kernel by itself: the first hidden GEMM (hipBLASLt MFMA)
        eng = forest.engine
        W, b, _ = eng.layers[1]
        cubes, rows = forest._net_input()
        x1 = forest._x1[:rows] if forest._fused else torch.randn((rows, W.shape[1]), dtype=W.dtype, device=W.device)
        out["gemm_hidden1_weight"] = (int(W.shape[0]), int(W.shape[1]))
        torch.addmm(b, x1, W.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.addmm(b, x1, W.t())
        e1.record()
        torch.cuda.synchronize()
        out["gemm_hidden1"] = round(e0.elapsed_time(e1) / reps, 4)
    return out


def phase_times_split(forest, c, max_states, reps):
    """phase_times for the f16x3 split engine: expand | network (operands + GEMMs + activation kernels) | backup | select,
    plus the two GEMMs of the first hidden layer alone (the dominant kernels of its step)."""
    import ctypes
    from librubiks import _hip
    from librubiks.model import _layer_call, _mm_f32
    lib, m, eng = forest.lib, ctypes.byref(forest.struct), forest.engine
    names = ["expand", "net_forward", "backup", "select"]
    acc = {k: 0.0 for k in names}
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        st = _hip.stream_ptr()
        ev[0].record()
        _hip.check(lib.rc_mcts_expand(m, max_states, st))
        ev[1].record()
        cubes, rows = forest._net_input()
        head = eng.head_cubes(cubes)
        ev[2].record()
        _hip.check(lib.rc_mcts_backup_head(m, head.data_ptr(), head.stride(0), 0, st))
        ev[3].record()
        _hip.check(lib.rc_mcts_select(m, c, forest.level_budget, st))
        ev[4].record()
        torch.cuda.synchronize()
        for i, k in enumerate(names):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
    out = {k: round(v / reps, 4) for k, v in acc.items()}
    cubes, rows = forest._net_input()
    hid = {}
    a = eng._first_from_cubes(cubes, eng.layers)   # the REAL activations of this step's children (MFMA time depends on the operand bits)
    if a is None:
        a = eng._forward(eng._input_from_cubes(cubes), eng.layers[:1] + eng.layers[-1:])   # not reached with fc_small (fused input layer)
    for li in (1, 2):   # the two hidden layers behind the input layer, as the engine runs them
        _, Wh, B2, b, code, alpha, W3 = eng.layers[li]
        K, N = Wh.shape[1], Wh.shape[0]
        plan = eng._layer_plan(rows, eng.layers, li)
        tile = eng._fused_tile(rows, N, K) if plan == "fused" else 0
        last = li == len(eng.layers) - 2
        if plan != "fused" and plan != "library":   # the own kernel with its K loop cut into chunks, raw fp32 partials
            _, cut_tile, chunks = plan
            part = torch.empty((chunks, rows, N), dtype=torch.float32, device=Wh.device)
            hid[f"gemm_hidden{li}"] = round(event_ms(lambda: _layer_call(
                "rc_split_layer_f16", a=a, w=W3, n_rows=rows, n_out=N, k=K, out_partials=part, k_splits=chunks, tile=cut_tile), reps)[0], 4)
            hid[f"gemm_hidden{li}_kernel"] = f"rc_split_layer_f16 (K loop in {chunks} chunks)"
            if not last:   # (behind the last hidden layer the fused head / the reduce kernel consumes the partials)
                a = eng._act(part, lib.rc_split_layer_corr_chunks(K, chunks), b, code, alpha, split=True)
        elif tile:   # one kernel: three f16 products + bias + activation + re-split (csrc/rubiks_gemm.hip)
            o = torch.empty((rows, N if last else 2 * N), dtype=torch.float32 if last else torch.float16, device=Wh.device)
            hid[f"gemm_hidden{li}"] = round(event_ms(lambda: _hip.check(lib.rc_split_gemm_f16(
                a.data_ptr(), W3.data_ptr(), b.data_ptr(), rows, N, K, code, alpha, None if last else o.data_ptr(),
                o.data_ptr() if last else None, tile, _hip.stream_ptr()), "rc_split_gemm_f16"), reps)[0], 4)
            hid[f"gemm_hidden{li}_kernel"] = "rc_split_gemm_f16"
            a = o
        else:      # hi x hi GEMM (K deep) + correction GEMM (2 K deep) through the library, + rc_split_reduce_f16
            part = torch.empty((2, rows, N), dtype=torch.float32, device=Wh.device)
            hid[f"gemm_hidden{li}_main"] = round(event_ms(lambda: _mm_f32(a[:, :K], Wh.t(), part[1]), reps)[0], 4)
            hid[f"gemm_hidden{li}_corr"] = round(event_ms(lambda: _mm_f32(a, B2.t(), part[0]), reps)[0], 4)
            hid[f"gemm_hidden{li}"] = round(hid[f"gemm_hidden{li}_main"] + hid[f"gemm_hidden{li}_corr"], 4)
            hid[f"gemm_hidden{li}_kernel"] = "hipBLASLt x2"
            a = eng._act(part, 1, b, code, alpha, split=not last)
    out.update(hid)
    Wh = eng.layers[1][1]
    out["gemm_hidden1_weight"] = (int(Wh.shape[0]), int(Wh.shape[1]))
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


def boundary_calls(reps=300):
    """
    The stateless drop-in functions at the sizes the reference's own callers use (agents.py:109,513: n = 1 and 12;
    ADI-sized 1 200), NumPy in / NumPy (or device tensor) out: microseconds per call, product (one HIP launch through
    pinned host memory + one stream synchronisation) next to the restated NumPy expression on one host core.
    """
    from librubiks import cube
    from oracle import cube as oc
    rng = np.random.RandomState(1)
    out = {"unit": "us per call (median)", "reps": reps}

    def med(fn):
        for _ in range(10):
            fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return round(float(np.median(ts)) * 1e6, 1)

    for n in (1, 12, 1200):
        states = np.tile(oc.get_solved(), (n, 1))
        for _ in range(20):
            states = oc.multi_rotate_actions(states, rng.randint(0, 12, n))
        acts = rng.randint(0, 12, n)
        faces, dirs = acts // 2, 1 - acts % 2
        row = {"multi_rotate": {"hip": med(lambda: cube.multi_rotate(states, faces, dirs)),
                                "numpy": med(lambda: oc.multi_rotate(states, faces, dirs))},
               "multi_is_solved": {"hip": med(lambda: cube.multi_is_solved(states)), "numpy": med(lambda: oc.multi_is_solved(states))},
               "as_oh": {"hip_to_device_tensor": med(lambda: cube.as_oh(states)),
                         "numpy_plus_copy_to_device": med(lambda: torch.from_numpy(oc.as_oh(states)).cuda())}}
        if n == 1:
            row["rotate"] = {"hip": med(lambda: cube.rotate(states[0], int(faces[0]), int(dirs[0]))),
                             "numpy": med(lambda: oc.rotate(states[0], int(faces[0]), int(dirs[0])))}
        out[f"n={n}"] = row
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


LEG_DTYPE = {   # leg name -> the arithmetic the network computes in (`dtype` of the JSON line)
    "f32s": "f32 (f16x3 split: three f16 MFMA products per layer, fp32 accumulate)",
    "f32": "f32",
    "bf16": "bf16",
}


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


SPREAD_WINDOWS = 5


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
        if run.done:
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
    res["roofline_env"] = {"kernel": f"expand12 + solved flags of {13 * n} states in one launch ({n} parents) + as_oh f32 ({n} states): the rollout's environment kernels",
                           "bound": "hbm", "algorithmic_bytes": int(env_bytes), "ms": round(env_ms, 4), "achieved": round(env_bytes / (env_ms * 1e-3) / 1e9, 1),
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(env_bytes / (env_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                           "note": "a rollout's arrays are a few MB: these launches are bound by launch latency, not by HBM (roofline_env of the main line has the kernels at 2^14 .. 2^26 states)"}
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


def preflight(rank, world, local_rank, backend, device_index, coll_device, need_gb):
    """
    Everything the N-rank run relies on, checked before any leg starts; a failure is ONE rank-tagged line on stderr and a
    non-zero exit of that rank (the launcher then ends the others): the device this rank is bound to (RCCL: GPU index =
    LOCAL_RANK), free HBM against the largest leg's reservation, and -- on the run's own process group and tensors of the sizes the
    legs use -- a float64 `all_reduce` (SUM and MAX) of a known vector, the `all_gather` of per-game results, and the
    `all_to_all_single` + `all_gather_into_tensor` pair of one 16 MB `GradBuckets` bucket.  Returns a dict for the result file.
    """
    def fail(what):
        print(f"[bench preflight] rank {rank}/{world} (local rank {local_rank}, device {device_index}, backend {backend}): {what}",
              file=sys.stderr, flush=True)
        sys.exit(3)

    out = {"backend": backend, "device_index": device_index}
    if backend == "nccl" and device_index != local_rank:
        fail(f"bound to GPU {device_index}, expected LOCAL_RANK {local_rank}")
    if torch.cuda.current_device() != device_index:
        fail(f"torch's current device is {torch.cuda.current_device()}")
    free, total = torch.cuda.mem_get_info(device_index)
    out["free_hbm_gb"], out["need_hbm_gb"] = round(free / 1e9, 1), need_gb
    if free < need_gb * 1e9:
        fail(f"{free / 1e9:.1f} GB of HBM free, the largest leg needs ~{need_gb} GB (another process on this GPU?)")
    from librubiks import _hip
    try:
        _hip.lib()
    except Exception as e:   # noqa: BLE001
        fail(f"librubiks_hip.so: {e!r}")
    if world > 1:
        try:
            v = torch.arange(4, dtype=torch.float64, device=coll_device) + rank
            sm, mx = v.clone(), v.clone()
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            want = torch.arange(4, dtype=torch.float64) * world + world * (world - 1) / 2
            if not torch.equal(sm.cpu(), want) or not torch.equal(mx.cpu(), torch.arange(4, dtype=torch.float64) + world - 1):
                fail(f"all_reduce of a known vector returned {sm.tolist()} / {mx.tolist()}")
            from librubiks.solving.sharding import gather_results
            games = 8
            g = gather_results({"nodes": np.full(games, rank, dtype=np.int64), "solved": np.ones(games, dtype=bool),
                                "lengths": np.full(games, 20 + rank, dtype=np.int64)}, games * world, device=coll_device)
            if not np.array_equal(g["nodes"], np.repeat(np.arange(world), games)) or not np.array_equal(g["lengths"], 20 + np.repeat(np.arange(world), games)):
                fail("gather_results did not return every rank's slice in rank order")
            n = (16 << 20) // 4 // world * world             # one GradBuckets bucket (fp32), whole shards
            send = torch.full((n,), float(rank + 1), device=coll_device)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send)
            shard = recv.view(world, -1).sum(0)
            back = torch.empty_like(send)
            dist.all_gather_into_tensor(back, shard)
            if float(back.min()) != world * (world + 1) / 2 or float(back.max()) != world * (world + 1) / 2:
                fail(f"all_to_all_single + all_gather_into_tensor of a bucket returned {float(back.min())} .. {float(back.max())}")
            torch.cuda.synchronize()
            dist.barrier()
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001
            fail(f"collective failed: {e!r}")
        out["collectives"] = "all_reduce SUM/MAX (f64), gather_results, all_to_all_single + all_gather_into_tensor (16 MB): ok"
    return out


def step_rooflines(engine, agent, roots, args, name):
    """Per-phase times of one lock-step iteration on a young forest of `trees` trees + the rooflines derived from them."""
    c = 0.6
    capacity = 12 * (args.phase_reps + 40) + 64
    agent.forest = None
    torch.cuda.empty_cache()
    forest = agent._forest_for(roots.n, capacity)
    forest.reset(roots)
    for _ in range(20):
        forest.step(c, forest.C, use_graph=False)
    torch.cuda.synchronize()
    phases = phase_times(forest, c, forest.C, args.phase_reps)
    # the production form of a step on the same young forest: network, then ONE tree kernel (backup + descent + next expansion)
    forest.reset(roots, forest.C)
    for _ in range(20):
        forest.step(c, forest.C, use_graph=False)
    if forest._one_launch:
        import ctypes
        from librubiks import _hip
        acc_net = acc_tree = 0.0
        for _ in range(args.phase_reps):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            if forest._fused:
                cubes, nrows = forest._net_input()
                head = forest.engine.head_cubes(cubes, None if forest._x1 is None else forest._x1[:nrows])
                ev[1].record()
                _hip.check(forest.lib.rc_mcts_step_head(ctypes.byref(forest.struct), head.data_ptr(), head.stride(0), int(head.dtype == torch.bfloat16),
                                                        c, forest.level_budget, forest.C, _hip.stream_ptr()), "rc_mcts_step_head")
            else:
                forest._evaluate_children()
                ev[1].record()
                _hip.check(forest.lib.rc_mcts_step(ctypes.byref(forest.struct), forest.probs.data_ptr(), forest.values.data_ptr(), c,
                                                   forest.level_budget, forest.C, _hip.stream_ptr()), "rc_mcts_step")
            ev[2].record()
            torch.cuda.synchronize()
            acc_net += ev[0].elapsed_time(ev[1])
            acc_tree += ev[1].elapsed_time(ev[2])
        phases["one_launch_form"] = {"net_forward": round(acc_net / args.phase_reps, 4), "tree_kernel": round(acc_tree / args.phase_reps, 4),
                                     "note": "the step the searches run: network, then rc_mcts_step* (the rows above are its three-phase form)"}
    rows, eng, fused = forest.rows_per_tree * roots.n, forest.engine, forest._fused
    if name == "f32s":
        W1 = phases.pop("gemm_hidden1_weight")
        f32_equiv = 2 * W1[0] * W1[1] * rows                      # the layer as an fp32 GEMM
        executed = 3 * f32_equiv                                    # three f16 products per element pair
        t = phases["gemm_hidden1"] * 1e-3
        own = phases.get("gemm_hidden1_kernel") == "rc_split_gemm_f16"
        gemm_traffic, gemm_traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", GEMM_PMC_FILE)
        if own and rows == 11264 and os.path.exists(tpath):   # a stored figure of exactly this kernel and launch shape, not measured in this run
            stored = json.load(open(tpath))
            gemm_traffic = stored["traffic_bytes"]
            gemm_traffic_src = (f"stored PMC figure: profiles/{GEMM_PMC_FILE} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                f"{stored['kernel']}, gfx950 corrections applied; algorithmic bytes 310 MB)")
        roofline = {"kernel_short": (f"rc_split_gemm_f16 352x256 tiles, hidden layer 1: [{rows}x{3 * W1[1]}]x[{3 * W1[1]}x{W1[0]}] f16 MFMA, f32 acc, +bias+ELU+re-split"
                                     if own else f"hidden layer 1 of the split engine via hipBLASLt: f16 GEMMs [{rows}x{3 * W1[1]}]x[{3 * W1[1]}x{W1[0]}], fp32 out"),
                    "kernel": (f"rc_split_gemm_f16 (own MFMA kernel, 352 x 256 tiles), first hidden layer: [{rows} x {3 * W1[1]}] x [{3 * W1[1]} x {W1[0]}] "
                               f"f16 products (hi.lo, lo.hi, hi.hi) in one fp32 accumulator + bias + ELU + re-split") if own else
                              (f"first hidden layer of the split engine via hipBLASLt: f16 GEMMs [{rows} x {W1[1]}] x [{W1[1]} x {W1[0]}] (hi.hi) and "
                               f"[{rows} x {2 * W1[1]}] x [{2 * W1[1]} x {W1[0]}] (hi.lo + lo.hi), fp32 out"),
                    "bound": "mfma", "achieved": round(executed / t / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(executed / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                    # `frac` prices the f16 flops the kernel EXECUTES (three products per fp32-equivalent product); SURVEY 8(d)'s algorithmic
                    # figure for the layer (2 x 4096 x 2048 flops per row = 189 GFLOP per launch) against the same peak is algorithmic_frac
                    "algorithmic_frac": round(f32_equiv / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": f32_equiv,
                    "traffic": gemm_traffic, "traffic_source": gemm_traffic_src,
                    "traffic_source_short": f"stored: profiles/{GEMM_PMC_FILE} (separate --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)" if gemm_traffic else None,
                    "flops_per_launch": executed,
                    "ms_per_launch": phases["gemm_hidden1"], "fp32_equivalent_tflops": round(f32_equiv / t / 1e12, 1),
                    "fp32_mfma_peak_tflops": MFMA_F32_PEAK_TFLOPS,
                    "note": "f16 MFMA flops executed (3 per fp32-equivalent flop) against the dense f16 peak; the same layer as an "
                            "fp32 MFMA GEMM is bounded by 157.3 TFLOP/s.  ms_per_launch is measured live with HIP events on the "
                            "launch stream (phases_ms.gemm_hidden1); the rocprofv3 average of the same kernel is in profiles/"}
        flops_net = eng.flops_per_state * rows
        group = {"kernel": f"whole split-engine forward on {rows} child rows (operand kernels + 5 f16 GEMMs + fp32 output layer)", "bound": "mfma",
                 "achieved": round(3 * flops_net / (phases["net_forward"] * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(3 * flops_net / (phases["net_forward"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                 "flops_per_launch": 3 * flops_net, "ms_per_launch": phases["net_forward"],
                 "fp32_equivalent_tflops": round(flops_net / (phases["net_forward"] * 1e-3) / 1e12, 1)}
        del forest
        agent.forest = None
        torch.cuda.empty_cache()
        return phases, roofline, group, None, rows
    peak = MFMA_BF16_PEAK_TFLOPS if name == "bf16" else MFMA_F32_PEAK_TFLOPS
    gemm_layers = eng.layers[1:] if fused else eng.layers
    flops = 2 * sum(W.shape[0] * W.shape[1] for W, _, _ in gemm_layers) * rows
    tf = flops / (phases["net_forward"] * 1e-3) / 1e12
    lib_name = "bf16 MFMA via hipBLASLt" if name == "bf16" else "fp32 MFMA via hipBLASLt (v_mfma_f32_*_f32, 1/16 of the bf16 rate)"
    group = {"kernel": f"policy/value net GEMMs on {rows} child rows ({len(gemm_layers)} GEMMs + bias + ELU passes, BatchNorm folded, "
                       f"heads merged), {lib_name}",
             "bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
             "flops_per_launch": flops, "ms_per_launch": phases["net_forward"]}
    roofline, roofline_input = group, None
    if "gemm_hidden1" in phases:   # the dominant kernel of the step, alone
        W1 = phases.pop("gemm_hidden1_weight")
        f1 = 2 * W1[0] * W1[1] * rows
        tf1 = f1 / (phases["gemm_hidden1"] * 1e-3) / 1e12
        roofline = {"kernel_short": f"hidden GEMM [{rows}x{W1[1]}]x[{W1[1]}x{W1[0]}] + bias, {'bf16' if name == 'bf16' else 'fp32'} MFMA via hipBLASLt: dominant kernel of a step",
                    "kernel": f"hidden GEMM [{rows} x {W1[1]}] x [{W1[1]} x {W1[0]}] + bias, {lib_name}: the dominant kernel of a step",
                    "bound": "mfma", "achieved": round(tf1, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf1 / peak, 4),
                    "traffic": None, "flops_per_launch": f1, "ms_per_launch": phases["gemm_hidden1"]}
    if fused:
        H = eng._fused_first[4]
        is_half = bool(eng._fused_first[5])
        kname = ("rc_first_layer_mfma_bf16 (one-hot x W1 on the matrix cores, one-hot fragments generated from the cube codes, "
                 "W1 slice in LDS, + bias + ELU)")
        f_in = 2 * 480 * H * rows
        tf_in = f_in / (phases["input_layer"] * 1e-3) / 1e12
        nbytes = (20 + 2 * H) * rows
        roofline_input = {"kernel": kname, "weights": "f16" if is_half else "bf16",
                          "bound": "mfma", "achieved": round(tf_in, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": round(tf_in / MFMA_BF16_PEAK_TFLOPS, 4), "flops_per_launch": f_in,
                          "traffic": None, "ms_per_launch": phases["input_layer"], "algorithmic_bytes": nbytes,
                          "note": "dense-equivalent flops of the 480-wide one-hot product; the kernel is bound by feeding the MFMAs "
                                  "from LDS (DESIGN.md section 3), its HBM traffic (20 B in, 2 H B out per row) is far from the HBM roof"}
    del forest
    agent.forest = None
    torch.cuda.empty_cache()
    return phases, roofline, group, roofline_input, rows


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


def scale_efficiency(world, value, workload_key, ref_path, write=True, gpu="", ref_value=None):
    """
    The scaling curve without post-processing: a one-GPU run leaves {value, workload} in `ref_path`; an N-GPU run of the SAME
    workload on the same checkout returns value / (N x that value).  (efficiency, note); efficiency is None for one GPU, when
    there is no record, or when the record is of another workload.
    """
    if world == 1:
        if write:
            try:
                with open(ref_path, "w") as f:
                    json.dump({"value": value, "workload": workload_key, "gpu": gpu}, f)
            except OSError:
                pass
        return None, "one GPU: this run IS the reference of the curve"
    if ref_value:   # handed in (--scale-ref-value / RUBIKS_SCALE_REF): no file of an earlier run is needed
        return round(value / (world * ref_value), 4), f"value / ({world} x {ref_value}), the one-GPU value given with --scale-ref-value / RUBIKS_SCALE_REF"
    name = os.path.basename(ref_path)
    if not os.path.exists(ref_path):
        return None, f"no one-GPU record of this workload ({name}) next to bench.py: run --gpus 1 first on this checkout"
    try:
        with open(ref_path) as f:
            ref = json.load(f)
    except (OSError, ValueError):
        return None, f"{name} is unreadable"
    if ref.get("workload") != workload_key or not ref.get("value"):
        return None, f"{name} holds another workload ({ref.get('workload')})"
    return round(value / (world * ref["value"]), 4), (f"value / ({world} x {ref['value']}), the one-GPU value this checkout's last --gpus 1 run of the "
                                                      f"same workload left in {name}")


def launch_ranks(n, argv, script=os.path.abspath(__file__)):
    """
    `python bench.py --gpus N` without a launcher (no RANK / WORLD_SIZE in the environment): this process starts the N ranks as
    children -- the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, i.e. what torch.distributed.run would have
    exported -- and relays rank 0's line as its own single stdout line.  It never touches the GPU itself (nothing here makes a
    HIP call, and no process that has initialised the GPU is ever exec'ed over).  A rank that dies takes the others with it
    and its exit code becomes ours.
    """
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # RCCL shares device buffers between the ranks of a node through IPC handles; the hosts of this pool only support the dmabuf
        # form (with the legacy mode hipIpcGetMemHandle fails with "invalid argument" and the first collective with it).  The image
        # exports the variable already; a rank started from an environment that lost it gets it back (tests/test_bench_line.py).
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drains rank 0's pipe while we wait
    reader.start()
    rc = 0
    try:
        while rc == 0 and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            rc = next((p.returncode for p in procs if p.poll() not in (None, 0)), 0)
        rc = rc or next((p.returncode for p in procs if p.returncode), 0)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        reader.join(timeout=20)
    out0 = "".join(chunks)
    if rc:
        print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
        return rc
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} result lines", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


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


SCALE_REF = "bench_scale_ref.json"   # left by a one-GPU run: the denominator of `efficiency` in the N-GPU runs that follow on the same checkout
LINE_LIMIT = 8000   # bytes of the final stdout line: the driver's record keeps the last 8 KB of output and parses the line from there


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(full, detail_name="bench_detail.json"):
    """
    The ONE stdout line of a run, built from the full result: what the bench contract names (metric ... config, roofline,
    cpu_baseline) and the flat scalars of `config.results`; the per-leg detail (`legs`, `astar`, `config5_share`, `adi`, the
    `roofline_env` ladder, boundary-call timings, notes) stays in `detail_name`, which the line names.  Always < LINE_LIMIT bytes:
    free text is clipped, and should the scalars ever outgrow the limit the least important groups are dropped (and listed).
    """
    cfg = full["config"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "scaling_measured", "rank_values", "value_per_gpu", "value_spread", "efficiency", "vs_baseline", "data") if k in full}
    if isinstance(line.get("value_spread"), dict):
        line["value_spread"] = {k: v for k, v in line["value_spread"].items() if k != "note"}
    line["dtype"] = _short(full["dtype"], 96)
    line["config"] = {"workload": _short(cfg.get("workload_short") or cfg["workload"], STR_LIMIT), "trees_per_gpu": cfg.get("trees_per_gpu"),
                      "max_states": cfg.get("max_states"), "scramble_depth": cfg.get("scramble_depth"), "parallelism": cfg.get("parallelism"),
                      "timed_region": _short(cfg.get("timed_region_short") or cfg.get("timed_region", ""), STR_LIMIT),
                      "results": dict(cfg.get("results") or {})}
    roof = full.get("roofline") or {}
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "algorithmic_frac", "traffic", "traffic_source", "algorithmic_bytes", "flops_per_launch",
            "algorithmic_flops_per_launch", "ms_per_launch", "ms", "fp32_equivalent_tflops")
    line["roofline"] = {k: (_short(roof.get("kernel_short") or roof[k], STR_LIMIT) if k == "kernel" else
                            _short(roof.get("traffic_source_short") or roof[k] or "", STR_LIMIT) if k == "traffic_source" else roof[k]) for k in keep if k in roof}
    for sub in ("env_multi_rotate_2p24", "astar_dominant_kernel", "adi_dominant_kernel", "adi_env"):
        if sub in roof:
            line["roofline"][sub] = {k: (_short(roof[sub].get("kernel_short") or v, STR_LIMIT) if isinstance(v, str) else v)
                                     for k, v in roof[sub].items() if k in keep}
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {k: (_short(cpu.get("sample_short") or cpu[k], STR_LIMIT) if k == "sample" else cpu[k])
                                for k in ("value", "unit", "cores", "host_cpus", "kind", "sample", "env_ops", "bfs_config1", "adi") if k in cpu}
    line["detail"] = detail_name
    dropped = []
    for victim in (("cpu_baseline", "env_ops"), ("cpu_baseline", "bfs_config1"), ("roofline", "adi_env"), ("roofline", "astar_dominant_kernel")):
        if len(json.dumps(line)) < LINE_LIMIT - 64:
            break
        if victim[1] in line.get(victim[0], {}):
            del line[victim[0]][victim[1]]
            dropped.append(".".join(victim))
    if len(json.dumps(line)) >= LINE_LIMIT - 64:   # last resort: keep the scalars in the order they were added until the line fits
        res, n_cut = line["config"]["results"], 0
        while res and len(json.dumps(line)) >= LINE_LIMIT - 160:
            res.popitem()
            n_cut += 1
        dropped.append(f"config.results: the last {n_cut} scalars")
    if dropped:
        line["dropped_to_detail"] = dropped
    return _clip_strings(line)


STR_LIMIT = 120   # characters of a string the driver's record keeps: longer descriptions live in the detail file


def _clip_strings(x):
    if isinstance(x, str):
        return _short(x, STR_LIMIT)
    if isinstance(x, dict):
        return {k: _clip_strings(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_clip_strings(v) for v in x]
    return x


def emit(full, detail_path):
    """Writes the full result to `detail_path` (and to gpurun_out/ when that exists) and prints the compact line, last, on stdout."""
    text = json.dumps(full)
    for path in {detail_path, *([os.path.join(ROOT, "gpurun_out", os.path.basename(detail_path))] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])}:
        try:
            with open(path, "w") as f:
                f.write(text + "\n")
        except OSError as e:   # a read-only checkout must not cost the run its line
            print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    line = json.dumps(compact_line(full, os.path.basename(detail_path)))
    assert len(line) < LINE_LIMIT and "\n" not in line
    sys.stdout.flush()
    print(line, flush=True)


if __name__ == "__main__":
    main()
