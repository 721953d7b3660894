"""Starting the ranks of an N-GPU run, the preflight before their first leg, and the scaling figure of the line."""
from .common import *   # noqa: F401,F403  (json, os, sys, time, np, torch, dist, ROOT, the roofline constants, progress, event_ms)

SCALE_REF = "bench_scale_ref.json"   # left by a one-GPU run: the denominator of `efficiency` in the N-GPU runs that follow on the same checkout


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


def launch_ranks(n, argv, script=os.path.join(ROOT, "bench.py")):
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
