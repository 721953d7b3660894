"""
Multi-rank tests that fail FAST: `run_ranks` starts one process per rank (spawn), waits for one result per rank in one-second
slices while watching the children, and on the first dead child raises with that child's traceback / stderr instead of
waiting out a queue timeout; whatever happens, every rank still alive is terminated before the test returns (a rank blocked in
a collective whose peer has died would otherwise outlive the test -- and, on the GPU box, keep holding the card).
`init_gloo` gives the process group a timeout, so a surviving rank's collective fails by itself as well.
"""
import datetime
import os
import queue
import socket
import sys
import tempfile
import time
import traceback

_ERROR = "__rank_error__"


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_gloo(rank: int, world: int, port: int, seconds: int = 60):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=seconds))


def _entry(target, rank, args, q, err_path):
    err = open(err_path, "w", buffering=1)
    os.dup2(err.fileno(), 2)          # the interpreter's and the runtime libraries' stderr, for crashes that raise nothing
    sys.stderr = err
    try:
        target(*args)
    except BaseException:             # noqa: BLE001 -- reported to the parent, then the process ends with a non-zero code
        tb = traceback.format_exc()
        err.write(tb)
        try:
            q.put((_ERROR, rank, tb))
            q.close()
            q.join_thread()           # the queue's feeder thread has written the report before the process goes
        finally:
            os._exit(1)


def run_ranks(target, world: int, make_args, timeout: float = 300.0) -> list:
    """
    Runs target(*make_args(rank, port, q)) in `world` spawned processes; every rank puts exactly one result on q.
    Returns the sorted results.  Raises AssertionError as soon as a rank has died or reported an exception (with its
    traceback), or when `timeout` seconds have passed; all ranks are terminated before returning.
    """
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    tmp = tempfile.mkdtemp(prefix="ranks_")
    errs = [os.path.join(tmp, f"rank{r}.err") for r in range(world)]
    procs = [ctx.Process(target=_entry, args=(target, r, make_args(r, port, q), q, errs[r])) for r in range(world)]

    def stderr_of(r):
        try:
            with open(errs[r]) as f:
                return f.read()[-4000:]
        except OSError:
            return ""

    def report(headline):
        """The failure text: what was noticed first, then exit code and stderr of EVERY rank (the rank that is noticed first
        is often the one whose collective broke because another rank had died)."""
        time.sleep(0.5)
        parts = [headline]
        for r, p in enumerate(procs):
            parts.append(f"--- rank {r}: {'alive' if p.exitcode is None else f'exit code {p.exitcode}'}; stderr:\n{stderr_of(r)}")
        return "\n".join(parts)

    got, deadline = [], time.monotonic() + timeout
    try:
        for p in procs:
            p.start()
        while len(got) < world:
            try:
                item = q.get(timeout=1.0)
            except queue.Empty:
                item = None
            if isinstance(item, tuple) and len(item) == 3 and item[0] == _ERROR:
                raise AssertionError(report(f"rank {item[1]} raised:\n{item[2]}"))
            if item is not None:
                got.append(item)
                continue
            for r, p in enumerate(procs):
                if p.exitcode not in (None, 0):
                    raise AssertionError(report(f"rank {r} died with exit code {p.exitcode} before reporting"))
            if time.monotonic() > deadline:
                raise AssertionError(report(f"no result from {world - len(got)} of {world} ranks after {timeout:.0f} s"))
        for r, p in enumerate(procs):
            p.join(max(1.0, deadline - time.monotonic()))
            assert p.exitcode == 0, report(f"rank {r} ended with exit code {p.exitcode}")
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(5)
    return sorted(got)
