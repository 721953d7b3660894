"""
The N > 1 paths use torch.distributed's "nccl" backend (= RCCL on ROCm) with CUDA tensors; the CPU tests rehearse them on gloo.
A one-GPU box cannot hold two RCCL ranks, but it can hold ONE: a process group of world size 1 over RCCL goes through the same
initialisation (device binding, HSA_ENABLE_IPC_MODE_LEGACY=0 environment), the same collective entry points and dtypes as the
driver's N = 2..8 runs -- barrier, MAX / SUM all_reduce of float64 CUDA tensors (bench.py), all_gather (sharding.gather_results),
and GradBuckets' ring and direct exchanges (all_to_all_single, all_gather_into_tensor, asynchronous, on CUDA gradients).
"""
import numpy as np
import pytest

from ranks import run_ranks

pytestmark = pytest.mark.gpu


def _rccl_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from librubiks.solving.sharding import pick_backend
    backend, index, coll_device = pick_backend({}, torch.cuda.device_count(), rank)
    assert (backend, coll_device) == ("nccl", "cuda")
    torch.cuda.set_device(index)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", index))
    dist.barrier()
    stats = torch.tensor([1.5, 2.0 ** 40 + 3, 0.0], dtype=torch.float64, device=coll_device)     # bench.py's statistics vector
    mx, sm = stats.clone(), stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    t = torch.arange(7, dtype=torch.float64, device=coll_device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    # the three asynchronous forms GradBuckets issues on CUDA gradients (with one rank it issues none itself)
    flat = torch.arange(24, dtype=torch.float32, device="cuda")
    recv, back = torch.empty_like(flat), torch.empty_like(flat)
    w = dist.all_to_all_single(recv, flat, async_op=True)
    w.wait()
    shard = recv.view(world, -1).sum(0)
    g = dist.all_gather_into_tensor(back, shard, async_op=True)
    ring = flat.clone()
    r = dist.all_reduce(ring, op=dist.ReduceOp.SUM, async_op=True)
    g.wait()
    r.wait()
    torch.cuda.synchronize()
    assert torch.equal(back, flat) and torch.equal(ring, flat)
    from librubiks.train import GradBuckets
    grads = {}
    for exchange in ("ring", "direct"):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(),
                                  torch.nn.Linear(30, 2)).cuda()
        buckets = GradBuckets(net, bucket_bytes=200, exchange=exchange)
        for step in range(2):
            buckets.zero()
            net(torch.full((3, 6), float(step + 1), device="cuda")).sum().backward()
            launched = len(buckets.works)
            buckets.wait()
        grads[exchange] = (launched, len(buckets.flats), [p.grad.cpu().numpy().copy() for p in net.parameters()])
        buckets.close()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(), torch.nn.Linear(30, 2)).cuda()
    net(torch.full((3, 6), 2.0, device="cuda")).sum().backward()
    plain = [p.grad.cpu().numpy().copy() for p in net.parameters()]
    q.put((rank, mx.cpu().tolist(), sm.cpu().tolist(), parts[0].cpu().tolist(), grads, plain))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_process_group_over_rccl_runs_every_collective_the_n_gpu_paths_use():
    (rank, mx, sm, part, grads, plain), = run_ranks(_rccl_worker, 1, lambda r, port, q: (r, 1, port, q), timeout=240)
    assert mx == sm == [1.5, 2.0 ** 40 + 3, 0.0] and part == list(range(7))
    for exchange in ("ring", "direct"):
        launched, n_buckets, g = grads[exchange]
        assert launched == 0 and n_buckets >= 3             # one rank: nothing to exchange, the buckets still hold the gradients
        assert all(np.array_equal(a, b) for a, b in zip(g, plain)), exchange      # mean over one rank = the local gradient, exactly
