"""
The ends of the shape range: a forest of 65 536 small trees (one launch with 65 536 workgroups, 720 896 network rows; allocated up
front -- MCTSForest.on_demand_pays), and ONE tree with the largest capacity a forest accepts (2^24 - 2 nodes: a tree's 256-byte node
records are addressed by 32-bit byte offsets; 4.8 GB of address space mapped on demand).  Sampled trees are replayed through the oracle on the network outputs the search
recorded, node for node.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

WEIGHTS = os.path.join(ROOT, "weights", "fc_small_r1")


def _replay(agent, res, states, t, cap, graph):
    from test_search_edge_gpu import _TableNet, _compare
    tree = agent.forest.tree_arrays(t)
    n = tree["n"]
    table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
    ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=graph)
    ok = ref.search(states[t], cap)
    assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n, f"tree {t}"
    assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
    _compare(tree, ref, n)


def test_a_forest_of_65536_small_trees():
    from librubiks import cube
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    from librubiks.solving.mcts_device import MCTSForest
    if not os.path.isdir(WEIGHTS):
        pytest.skip("needs the trained weights")
    B, cap = 65536, 150
    assert not MCTSForest.on_demand_pays(B, cap) or MCTSForest.VMM_MIN_BYTES == 0
    np.random.seed(3)
    cubes, _, _ = cube.scramble_batch(B, 9, True)
    states = cubes.numpy()
    agent = MCTS(Model.load(WEIGHTS).eval(), c=0.6, search_graph=False, net_dtype=F32_SPLIT)
    res = agent.search_batch(cubes, None, cap, compact=False)
    assert res.nodes.shape == (B,) and (res.nodes >= 13).all() and (res.nodes <= cap).all()
    assert 0.05 < res.solved.mean() < 1.0 and ((res.nodes + 12 > cap) | res.solved).all()       # every tree ran to its end
    for t in (0, 1, 255, 256, 32767, 32768, 65534, 65535):
        _replay(agent, res, states, t, cap, False)


def test_one_tree_with_the_largest_capacity():
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    if not os.path.isdir(WEIGHTS):
        pytest.skip("needs the trained weights")
    from librubiks.solving.mcts_device import MAX_CAPACITY, MCTSForest
    cap = MAX_CAPACITY
    assert cap == (1 << 24) - 2
    with pytest.raises(AssertionError, match="capacity"):
        MCTSForest(1, cap + 1)
    np.random.seed(4)
    states = np.array([oc.scramble(30, True)[0]])
    agent = MCTS(Model.load(WEIGHTS).eval(), c=0.6, search_graph=True, net_dtype=F32_SPLIT)
    res = agent.search_batch(states, None, cap, max_iterations=200, compact=False)
    forest = agent.forest
    assert forest.vmm and forest.C == cap and forest.bytes_reserved() > 4e9 and forest.bytes_mapped() < 1e9
    assert res.iterations[0] == 200 and res.nodes[0] > 1500 and not res.solved[0]
    tree = forest.tree_arrays(0)
    n = tree["n"]
    from test_search_edge_gpu import _TableNet, _compare
    table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
    ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=True)
    ref.search(states[0], cap, max_iterations=200)
    assert len(ref) == n and ref.iterations == res.iterations[0]
    _compare(tree, ref, n)


def test_environment_kernels_beyond_2_to_the_31_states():
    """multi_rotate, is_solved and the AoS <-> SoA transposes on 2^31 + 4 096 + 5 cube states (43 GB per SoA: element indices pass
    2^32 inside a plane, state indices pass 2^31): windows at the start, either side of 2^31 and at the ragged end against the
    oracle; the whole array through a property -- every action followed by its inverse gives the input back."""
    from librubiks import cube
    from librubiks.cube.device import DeviceCubes
    n = (1 << 31) + 4096 + 5
    if torch.cuda.mem_get_info()[0] < 150e9:
        pytest.skip("needs 150 GB of free HBM")
    np.random.seed(6)
    block = 1 << 20
    base, _, _ = cube.scramble_batch(block, 30, True)
    big = DeviceCubes.empty(n)
    for lo in range(0, n, block):
        w = min(block, n - lo)
        big.soa[:, lo:lo + w] = base.soa[:, :w]
    g = torch.Generator(device="cuda").manual_seed(1)
    actions = torch.randint(0, 12, ((n + 15) // 16 * 16,), dtype=torch.uint8, device="cuda", generator=g)
    out = big.multi_rotate(actions)
    windows = [0, (1 << 31) - 2048, n - 4096 - 5]
    for lo in windows:
        w = min(4096 + 5, n - lo)
        src = big.soa[:, lo:lo + w].T.contiguous().cpu().numpy()
        got = out.soa[:, lo:lo + w].T.contiguous().cpu().numpy()
        assert np.array_equal(got, oc.multi_rotate_actions(src, actions[lo:lo + w].cpu().numpy())), lo
    back = out.multi_rotate(actions ^ 1)                       # the inverse of action a is a ^ 1
    for lo in range(0, n, 1 << 28):
        hi = min(n, lo + (1 << 28))
        assert torch.equal(back.soa[:, lo:hi], big.soa[:, lo:hi]), lo
    del back
    # is_solved: plant solved cubes at chosen places, the flags must find exactly those
    solved = torch.from_numpy(oc.get_solved().astype(np.int8)).cuda()
    where = [0, 12345, (1 << 31) - 1, 1 << 31, (1 << 31) + 17, n - 1]
    for i in where:
        out.soa[:, i] = solved
    flags = out.is_solved()
    mask, count = out.solved_mask()
    assert int(count.item()) == len(where)
    for i in where:
        assert (int(mask[i // 64].item()) >> (i % 64)) & 1
    assert sum(int(flags[lo:lo + (1 << 28)].sum().item()) for lo in range(0, n, 1 << 28)) == len(where) and all(bool(flags[i].item()) for i in where)
    # the ragged end through the transposes
    aos = out.to_aos()
    assert aos.shape == (n, 20) and np.array_equal(aos[n - 3:].cpu().numpy(), out.soa[:, n - 3:n].T.cpu().numpy())
    assert np.array_equal(aos[(1 << 31) - 2:(1 << 31) + 2].cpu().numpy(), out.soa[:, (1 << 31) - 2:(1 << 31) + 2].T.cpu().numpy())


def test_astar_on_32768_problems_and_on_one_problem_with_a_large_open_list(standin_net):
    """Batched A* at the ends of its shape range, stand-in net (exact arithmetic), sampled problems against the oracle:
    32 768 problems at once (8 expansions each per iteration), and ONE problem with 1 000 expansions per iteration and
    3 000 000 node slots."""
    from librubiks.solving.agents import AStar
    net = standin_net.cuda()
    onet = oa.TorchNet(net, device="cuda")
    np.random.seed(12)
    B = 32768
    states = np.array([oc.scramble(1 + i % 9, True)[0] for i in range(512)])
    states = np.tile(states, (B // 512, 1))
    res = AStar(net, lambda_=0.3, expansions=8, net_dtype=torch.float32).search_batch(states, None, 400)
    assert res.nodes.shape == (B,) and 0.2 < res.solved.mean() < 1.0
    for b in (0, 1, 511, 512, 16383, 16384, B - 2, B - 1):
        ref = oa.AStar(onet, lambda_=0.3, expansions=8)
        ok = ref.search(states[b], 400)
        assert bool(res.solved[b]) == ok and res.nodes[b] == len(ref) and list(res.queues[b]) == list(ref.action_queue), b
    assert np.array_equal(res.nodes[:512], res.nodes[512:1024]) and np.array_equal(res.lengths[:512], res.lengths[-512:])   # equal problems, equal ends
    one = np.array([oc.scramble(40, True)[0]])
    res = AStar(net, lambda_=0.1, expansions=1000, net_dtype=torch.float32).search_batch(one, None, 3_000_000)
    ref = oa.AStar(onet, lambda_=0.1, expansions=1000)
    ok = ref.search(one[0], 3_000_000, max_iterations=None)
    assert bool(res.solved[0]) == ok and res.nodes[0] == len(ref) and list(res.queues[0]) == list(ref.action_queue)
    assert res.nodes[0] > 100_000 or ok
