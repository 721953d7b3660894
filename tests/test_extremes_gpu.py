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
