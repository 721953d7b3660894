"""N > 1 path on CPU: two gloo ranks shard the games, gather per-game results, agree on the summary."""
import os
import time

import numpy as np
import pytest
import torch.distributed as dist

from librubiks.solving.sharding import gather_results, shard_range, summarize
from ranks import init_gloo, run_ranks


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 1024, 65536, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_backend_choice():
    """bench.py's process-group set-up: RCCL with one GPU per local rank by default, gloo only on request."""
    import pytest
    from librubiks.solving.sharding import pick_backend
    assert pick_backend({}, 8, 0) == ("nccl", 0, "cuda")
    assert pick_backend({}, 8, 7) == ("nccl", 7, "cuda")
    with pytest.raises(RuntimeError):
        pick_backend({}, 1, 1)          # two RCCL ranks cannot share one GPU
    with pytest.raises(RuntimeError):
        pick_backend({}, 0, 0)
    assert pick_backend({"RUBIKS_DIST_BACKEND": "gloo"}, 1, 1) == ("gloo", 0, "cpu")
    assert pick_backend({"RUBIKS_DIST_BACKEND": "gloo"}, 2, 3) == ("gloo", 1, "cpu")
    assert pick_backend({"RUBIKS_DIST_BACKEND": "gloo"}, 0, 1) == ("gloo", 0, "cpu")
    with pytest.raises(ValueError):
        pick_backend({"RUBIKS_DIST_BACKEND": "mpi"}, 1, 0)


def _worker(rank, world, port, n_games, q):
    init_gloo(rank, world, port)
    lo, hi = shard_range(n_games, rank, world)
    g = np.arange(lo, hi)
    local = {"solved": (g % 3 == 0), "lengths": np.where(g % 3 == 0, g % 11, -1), "nodes": 100 + g}
    full = gather_results(local, n_games)
    q.put((rank, {k: v.tolist() for k, v in full.items()}, summarize(full, 2.0)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_games,world", [(37, 2), (37, 8), (5, 8)])
def test_gather_of_uneven_shards(n_games, world):
    """Two ranks, the eight of a node (uneven shards), and more ranks than games (some shards are empty)."""
    got = run_ranks(_worker, world, lambda r, port, q: (r, world, port, n_games, q), timeout=120)
    g = np.arange(n_games)
    for rank, full, summary in got:
        assert full["nodes"] == (100 + g).tolist()
        assert full["solved"] == (g % 3 == 0).tolist()
        assert full["lengths"] == np.where(g % 3 == 0, g % 11, -1).tolist()
        assert summary["games"] == n_games and summary["nodes"] == int((100 + g).sum())
        assert abs(summary["nodes_per_sec"] - (100 + g).sum() / 2.0) < 1e-6
    assert all(g[2] == got[0][2] for g in got)


def _grad_worker(rank, world, port, q):
    import torch
    init_gloo(rank, world, port)
    from librubiks.train import average_gradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2))
    x = torch.full((3, 6), float(rank + 1))
    net(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    average_gradients(net)
    q.put((rank, [g.tolist() for g in local], [p.grad.tolist() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_average():
    """Config #4's only collective: the mean of per-rank gradients in one flat all_reduce."""
    world = 2
    got = run_ranks(_grad_worker, world, lambda r, port, q: (r, world, port, q), timeout=120)
    mean = [np.mean([np.array(got[r][1][i]) for r in range(world)], axis=0) for i in range(len(got[0][1]))]
    for r in range(world):
        for i, m in enumerate(mean):
            assert np.allclose(np.array(got[r][2][i]), m, atol=1e-6)
    assert not np.allclose(np.array(got[0][1][0]), np.array(got[1][1][0]))   # local gradients did differ


class _ToyAgent:
    """search_batch on host arrays with a deterministic per-game outcome (stands in for the GPU agents on the CPU)."""

    def search_batch(self, states, time_limit=None, max_states=None):
        from types import SimpleNamespace
        key = states.astype(np.int64).sum(axis=1)
        solved = key % 2 == 0
        return SimpleNamespace(solved=solved, lengths=np.where(solved, key % 17, -1), nodes=1000 + key)


def _search_worker(rank, world, port, states, q):
    init_gloo(rank, world, port)
    from librubiks.solving.sharding import sharded_search_batch
    got = sharded_search_batch(_ToyAgent(), states, None, 100)
    q.put((rank, {k: v.tolist() for k, v in got.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_games", [(2, 41), (8, 41), (8, 3)])
def test_sharded_search_batch_on_several_ranks(world, n_games):
    """Every rank ends with the results of ALL games, equal to the single-process search (eight ranks; more ranks than games)."""
    from librubiks.solving.sharding import sharded_search_batch
    rng = np.random.default_rng(0)
    states = rng.integers(0, 24, size=(n_games, 20)).astype(np.int8)
    whole = sharded_search_batch(_ToyAgent(), states, None, 100)      # no process group: plain search
    got = run_ranks(_search_worker, world, lambda r, port, q: (r, world, port, states, q), timeout=180)
    assert len(got) == world
    for _, full in got:
        for k in ("solved", "lengths", "nodes"):
            assert full[k] == whole[k].tolist()


def _bucket_worker(rank, world, port, exchange, q):
    import torch
    init_gloo(rank, world, port)
    from librubiks.train import GradBuckets
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(),
                              torch.nn.Linear(30, 2))
    buckets = GradBuckets(net, bucket_bytes=200, exchange=exchange)      # three buckets, one per Linear layer (last layer first)
    assert buckets.direct == (exchange == "direct") and all(f.numel() % world == 0 for f in buckets.flats)
    out = []
    for step in range(3):                             # a second step: zero() really resets, hooks fire again
        buckets.zero()
        x = torch.full((3, 6), float(rank + 1 + step))
        # third step: a loss that does not reach the last layer -- its bucket never fires from the hooks and wait() must reduce it
        (net(x) if step < 2 else net[:-1](x)).sum().backward()
        launched = len(buckets.works)
        local = [p.grad.clone() for p in net.parameters()]   # (already reduced where a bucket has finished: compare on rank sums)
        buckets.wait()
        out.append((launched, [p.grad.tolist() for p in net.parameters()]))
    views = all(p.grad.data_ptr() >= f.data_ptr() and p.grad.data_ptr() < f.data_ptr() + f.numel() * 4
                for p in net.parameters() for f in [buckets.flats[buckets.bucket_of[p]]])
    q.put((rank, len(buckets.flats), views, out))
    buckets.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "auto"), (2, "direct"), (3, "direct"), (8, "direct")])
def test_bucketed_async_gradient_average(world, exchange):
    """GradBuckets: gradients live in flat buckets, every bucket's exchange starts during backward, result = mean over ranks --
    by one all_reduce per bucket ("auto" on gloo = ring) or by the direct form (all_to_all_single of shards, local sum,
    all_gather_into_tensor: what uses all xGMI links at once; three ranks: buckets padded to whole shards; eight: the rank count of a node,
    where "auto" picks this form on RCCL)."""
    import torch
    got = run_ranks(_bucket_worker, world, lambda r, port, q: (r, world, port, exchange, q), timeout=120)
    assert got[0][1] == got[1][1] >= 3 and got[0][2] and got[1][2]
    for step in range(3):
        # reference: the same two local backward passes in this process, averaged by hand
        grads = []
        for rank in range(world):
            torch.manual_seed(0)
            net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(),
                                      torch.nn.Linear(30, 2))
            x = torch.full((3, 6), float(rank + 1 + step))
            (net(x) if step < 2 else net[:-1](x)).sum().backward()
            grads.append([p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32) for p in net.parameters()])
        mean = [np.mean([grads[r][i] for r in range(world)], axis=0) for i in range(len(grads[0]))]
        for r in range(world):
            launched, reduced = got[r][3][step]
            assert launched == got[r][1] - (step == 2)        # every bucket that got its gradients was reduced inside backward
            for i, m in enumerate(mean):
                assert np.allclose(np.array(reduced[i]), m, rtol=1e-6, atol=1e-7)
        assert all(got[r][3][step][1] == got[0][3][step][1] for r in range(world))     # bit-identical on every rank


def _dying_worker(rank, world, port, how, q):
    init_gloo(rank, world, port)
    if rank == 1:
        if how == "raise":
            raise RuntimeError("rank 1 gives up before the collective")
        os.write(2, b"rank 1 aborts without a traceback\n")
        os._exit(3)
    dist.barrier()                 # rank 0 waits for a peer that is gone
    q.put((rank, "unreachable"))


@pytest.mark.parametrize("how", ["raise", "exit"])
def test_a_dead_rank_fails_the_test_quickly_with_its_output(how):
    """What `run_ranks` is for (round 3: one rank died in an assert, the other sat in a gloo collective and the queue wait was
    longer than the box's 420-s silence limit): the failure comes within seconds, carries the dead rank's traceback or stderr,
    and no rank outlives the test."""
    import multiprocessing
    t0 = time.monotonic()
    with pytest.raises(AssertionError) as e:
        run_ranks(_dying_worker, 2, lambda r, port, q: (r, 2, port, how, q), timeout=120)
    assert time.monotonic() - t0 < 15
    text = str(e.value)
    assert ("rank 1 gives up before the collective" in text and "Traceback" in text) if how == "raise" else \
        ("exit code 3" in text and "rank 1 aborts without a traceback" in text)
    assert not multiprocessing.active_children()
