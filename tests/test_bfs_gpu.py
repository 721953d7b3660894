"""
Device BFS (rc_bfs_* behind librubiks.solving.agents.BFS) against the reference run of BASELINE
config #1 (tests/golden/bfs_golden.npz) and against the restated FIFO loop (oracle/agents.py),
including the max_states cut and every chunking of a level.  Exact comparisons.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402


def _apply(state, queue):
    for a in queue:
        state = oc.rotate(state, *oc.ACTION_SPACE[a])
    return state


def test_bfs_config1_reference_run(bfs_golden):
    """10 depth-5 scrambles after set_seeds(): solution, its length and len(agent) of the reference's run."""
    from librubiks.solving.agents import BFS
    agent = BFS()
    assert str(agent) == "Breadth-first search"
    for s, length, seen, queue in zip(bfs_golden["states"], bfs_golden["lengths"], bfs_golden["seen"], bfs_golden["queues"]):
        assert agent.search(s, None, 10_000_000)
        assert len(agent) == seen
        assert list(agent.action_queue) == list(queue[:length])
        assert oc.is_solved(_apply(s, agent.action_queue))
    res = agent.search_batch(bfs_golden["states"], None, 10_000_000)
    assert res.solved.all() and np.array_equal(res.lengths, bfs_golden["lengths"]) and np.array_equal(res.nodes, bfs_golden["seen"])


@pytest.mark.parametrize("chunk", [1, 5, 64, 4096])
def test_bfs_vs_oracle_with_cut(chunk):
    """Solved and cut searches for several max_states; the chunking of a level never shows."""
    from librubiks.solving.agents import BFS
    np.random.seed(11 + chunk)
    agent = BFS(chunk=chunk)
    ref = oa.BFS()
    depths = [1, 2, 3, 3, 4, 4] if chunk < 64 else [1, 2, 3, 4, 4, 5]
    for d in depths:
        s = oc.scramble(d, True)[0]
        for cap in (1, 2, 13, 14, 150, 1500, 20_000):
            if chunk < 64 and cap > 1500:
                continue
            ok = ref.search(s, cap)
            assert agent.search(s, None, cap) == ok, (d, cap)
            assert len(agent) == len(ref), (d, cap)
            assert list(agent.action_queue) == list(ref.action_queue), (d, cap)


def test_bfs_node_store_matches_discovery_order():
    """Committed nodes are the oracle's dict in insertion order (states, parent link, action)."""
    from librubiks.solving.agents import BFS
    np.random.seed(3)
    s = oc.scramble(6, True)[0]
    cap = 2000
    ref = oa.BFS()
    assert not ref.search(s, cap)
    agent = BFS(chunk=37)
    assert not agent.search(s, None, cap)
    assert len(agent) == len(ref)
    arr = agent._dev.node_arrays()
    n = len(arr["states"])
    keys = list(ref.states.keys())
    assert 0 < n <= len(keys)
    index = {k: i for i, k in enumerate(keys)}
    for i in range(n):
        assert arr["states"][i].tobytes() == keys[i]
        pk, a = ref.states[keys[i]]
        if pk is not None:
            assert index[pk] == arr["parent"][i] and a == arr["action"][i]


def test_bfs_solved_start_and_asserts():
    from librubiks.solving.agents import BFS
    agent = BFS()
    assert agent.search(oc.get_solved(), None, 100) and len(agent.action_queue) == 0 and len(agent) == 0
    with pytest.raises(AssertionError):
        agent.search(oc.get_solved(), None, None)   # reference agents.py:54
    s = oc.rotate(oc.get_solved(), 0, 1)
    assert agent.search(s, 5.0, None) and list(agent.action_queue) == [1]


def test_bfs_and_select_argument_errors():
    """The C ABI refuses what would index out of its buffers (no launch happens)."""
    import ctypes
    import torch
    from librubiks import _hip
    from librubiks.solving.bfs_device import BFSDevice
    from librubiks.solving.mcts_device import MCTSForest
    dev = BFSDevice(1000, chunk=16)
    lib, m = _hip.lib(), ctypes.byref(dev.struct)
    assert lib.rc_bfs_expand(m, 0, 17, 1, 1000, None) == -4       # more parents than the chunk: RC_ERR_RANGE
    assert lib.rc_bfs_expand(m, 5, 4, 6, 1000, None) == -4        # parents beyond the stored nodes
    assert lib.rc_bfs_expand(m, 0, 16, dev.capacity - 10, 1000, None) == -4   # children would not fit
    assert lib.rc_bfs_init(m, None, None) == -1                   # RC_ERR_NULL
    assert lib.rc_bfs_path(m, dev.capacity, dev._path.data_ptr(), dev._path_len.data_ptr(), 64, None) == -4
    broken = type(dev.struct)()
    ctypes.memmove(ctypes.byref(broken), ctypes.byref(dev.struct), ctypes.sizeof(broken))
    broken.hash_size = dev.hash_size - 1                          # not a power of two
    assert lib.rc_bfs_init(ctypes.byref(broken), dev._root.data_ptr(), None) == -4
    forest = MCTSForest(4, 64)
    s = type(forest.struct)()
    ctypes.memmove(ctypes.byref(s), ctypes.byref(forest.struct), ctypes.sizeof(s))
    s.rec = None
    assert lib.rc_mcts_select(ctypes.byref(s), 0.6, 0, None) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("chunk", [None, 3, 100])
def test_bfs_max_states_cut_reference_runs(chunk):
    """The same 40 reference searches on the device BFS, for several chunkings of a level."""
    import os
    from conftest import GOLDEN
    from librubiks.solving.agents import BFS
    g = np.load(os.path.join(GOLDEN, "bfs_cut_golden.npz"))
    agent = BFS(chunk=chunk)
    for s, cap, solved, seen, queue in zip(g["states"], g["caps"], g["solved"], g["seen"], g["queues"]):
        assert agent.search(s, None, int(cap)) == bool(solved)
        assert len(agent) == seen
        assert list(agent.action_queue) == [a for a in queue if a >= 0]
