"""
Multi-GPU layout of a batched search: scrambles are independent (the reference runs them strictly
one after another, librubiks/solving/evaluation.py:71-80), so rank r simply owns a contiguous slice
of the games and nothing is exchanged while searching.  The only collective is the final gather of
the per-game results (RCCL all_gather over xGMI on GPUs, gloo in the CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) owned by `rank`; sizes differ by at most one and cover 0..n_items exactly."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pick_backend(env: dict, n_devices: int, local_rank: int):
    """
    (torch.distributed backend, CUDA device index of this rank, device of the tensors handed to collectives).
    Default: "nccl" (= RCCL on ROCm) with CUDA tensors and ONE GPU PER LOCAL RANK -- more local ranks than GPUs is an error,
    two RCCL ranks cannot share a device.  RUBIKS_DIST_BACKEND=gloo rehearses the N > 1 code path with several ranks on
    one GPU (or none): collectives on CPU tensors, ranks mapped onto the devices round robin.
    """
    backend = env.get("RUBIKS_DIST_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise ValueError(f"RUBIKS_DIST_BACKEND must be nccl or gloo, not {backend!r}")
    if backend == "nccl":
        if not 0 <= local_rank < n_devices:
            raise RuntimeError(f"local rank {local_rank} has no GPU of its own ({n_devices} visible): one process per GPU")
        return backend, local_rank, "cuda"
    return backend, (local_rank % n_devices if n_devices else 0), "cpu"


def gather_results(local: dict, n_total: int, device=None) -> dict:
    """
    all_gather of per-game result vectors.  `local` maps name -> 1-D array for this rank's games (in
    game order); returns name -> array of length n_total in global game order on every rank.
    Works for any world size, uneven shards included (vectors are padded to the largest shard).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return {k: np.asarray(v) for k, v in local.items()}
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    out = {}
    for name, vec in local.items():
        vec = np.asarray(vec)
        assert len(vec) == sizes[rank][1] - sizes[rank][0], f"{name}: shard length mismatch"
        t = torch.zeros(pad, dtype=torch.float64, device=device)
        t[:len(vec)] = torch.from_numpy(vec.astype(np.float64)).to(t.device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        full = np.concatenate([p[:hi - lo].cpu().numpy() for p, (lo, hi) in zip(parts, sizes)])
        out[name] = full.astype(vec.dtype)
    return out


def sharded_search_batch(agent, states: np.ndarray, time_limit=None, max_states=None, device=None, **kwargs) -> dict:
    """
    One batched search over `states` ((n, 20) int8, the SAME array on every rank) with the games split over the
    ranks of the default process group: rank r runs agent.search_batch on its contiguous slice, then the per-game
    vectors are all-gathered.  Returns {"solved", "lengths", "nodes", "seconds"} for all n games, in game order, on every rank
    ("seconds": every game's own wall interval on its rank, `BatchResult.game_seconds`).
    With a single process this is agent.search_batch on everything.
    """
    states = np.asarray(states)
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_range(len(states), rank, world)
    if hi > lo:
        out = agent.search_batch(states[lo:hi], time_limit, max_states, **kwargs)
        each = getattr(out, "game_seconds", None)
        local = {"solved": np.asarray(out.solved), "lengths": np.asarray(out.lengths), "nodes": np.asarray(out.nodes),
                 "seconds": np.asarray(each, dtype=np.float64) if each is not None else np.full(hi - lo, float(getattr(out, "seconds", 0.0)) / max(1, hi - lo))}
    else:   # more ranks than games
        local = {"solved": np.zeros(0, dtype=bool), "lengths": np.zeros(0, dtype=np.int64), "nodes": np.zeros(0, dtype=np.int64),
                 "seconds": np.zeros(0)}
    return gather_results(local, len(states), device=device)


def summarize(results: dict, seconds: float) -> dict:
    """Evaluator-style summary (librubiks/solving/evaluation.py:96-125): solve rate +/- 95 % half-width, nodes/s."""
    solved = np.asarray(results["solved"]).astype(bool)
    n = len(solved)
    p = float(solved.mean()) if n else 0.0
    lengths = np.asarray(results["lengths"])[solved]
    return {
        "games": n, "solve_rate": p, "solve_rate_ci95": float(1.959963984540054 * np.sqrt(p * (1 - p) / max(n, 1))),
        "mean_length": float(lengths.mean()) if len(lengths) else None,
        "median_length": float(np.median(lengths)) if len(lengths) else None,
        "nodes": int(np.asarray(results["nodes"]).sum()),
        "nodes_per_sec": float(np.asarray(results["nodes"]).sum() / max(seconds, 1e-12)),
    }
