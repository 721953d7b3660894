"""Per-kernel timings of a lock-step iteration and of the environment kernels, and the rooflines derived from them."""
from .common import *   # noqa: F401,F403  (json, os, sys, time, np, torch, dist, ROOT, the roofline constants, progress, event_ms)


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
