"""
GPU parity tests for the behaviours this build DEFINES where the reference raises, hangs or is unbounded
(DESIGN.md section 4), HIP path vs the oracle's identical definition, plus an exact-tree test of the
production bf16 path.

1. MCTS leaf whose 12 children are all in the tree already (reference: ValueError on v.max() of an empty array,
   librubiks/solving/agents.py:548-559): best value = max V over the existing neighbours.  Unreachable in small
   natural searches (the cube graph has no short cycles besides a a' and commuting faces), so the situation is
   planted into BOTH trees: the unseen children of the current leaf are appended as leaf nodes.
2. A caller-bounded path store (MCTS(max_path=...), a resource bound the reference does not have): exactly the trees whose
   reference search makes a descent that does not fit end with RC_MCTS_PATH_OVERFLOW, every other tree is the reference's.
   (The default store has no bound: tests/test_deep_paths_gpu.py.)
3. A* whose open list runs dry (reference: spins in agents.py:236-239): ends unsolved, RC_ASTAR_OPEN_EMPTY.
4. Searches bounded by time only: the trees' capacity is what the kernels can address / what the hash tables' budget allows
   (`agents.time_only_capacity`; the reference's arrays double without bound, agents.py:450-459).  A tree that does reach its
   capacity ends EXHAUSTED exactly like one that reaches max_states (growth past the old 2^18 cap: tests/test_deep_paths_gpu.py).
5. Production dtype: the trees the bf16 engine builds (packed 11 rows, fused input layer / head / backup, graph
   replay, line following) equal node-for-node what the oracle builds from the SAME network outputs, replayed
   through a table-lookup net recorded from the device trees.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

_M32 = 0xFFFFFFFF


def _pack_key(state: np.ndarray) -> np.ndarray:
    """int8[20] -> uint32[4]: 20 codes of 5 bits, 6 per dword (csrc/rubiks_common.h key_set)."""
    k = [0, 0, 0, 0]
    for j, code in enumerate(state):
        k[j // 6] |= (int(code) & 31) << (5 * (j % 6))
    return np.array(k, dtype=np.uint32)


def _key_hash(k) -> int:
    """csrc/rubiks_common.h key_hash."""
    h = (int(k[0]) * 0x9E3779B1) & _M32
    h ^= h >> 15
    h = (h + int(k[1]) * 0x85EBCA77) & _M32
    h ^= h >> 13
    h = (h + int(k[2]) * 0xC2B2AE3D) & _M32
    h ^= h >> 16
    h = (h + int(k[3]) * 0x27D4EB2F) & _M32
    h ^= h >> 15
    h = (h * 0x165667B1) & _M32
    h ^= h >> 16
    return h


def _compare(tree: dict, ref, n: int):
    assert tree["n"] == n == len(ref)
    assert np.array_equal(tree["states"][1:n + 1], ref.states[1:n + 1])
    assert np.array_equal(tree["neighbors"][:n + 1], ref.neighbors[:n + 1])
    assert np.array_equal(tree["leaves"][1:n + 1], ref.leaves[1:n + 1])
    assert np.array_equal(tree["N"][:n + 1], ref.N[:n + 1])
    assert np.array_equal(tree["L"][:n + 1], ref.L[:n + 1])
    assert np.array_equal(tree["V"][1:n + 1], ref.V[1:n + 1].astype(np.float64))
    assert np.array_equal(tree["P"][1:n + 1], ref.P[1:n + 1].astype(np.float64))
    expanded = ~ref.leaves[1:n + 1]   # W of a never-expanded node is v everywhere on both sides; compare all rows
    assert np.array_equal(tree["W"][1:n + 1], ref.W[1:n + 1]) and expanded.any()


@pytest.fixture(scope="module")
def net_gpu(standin_net):
    return standin_net.cuda()


def test_leaf_with_all_children_known(net_gpu):
    from librubiks.cube import DeviceCubes
    from librubiks.model import GenericNet
    from librubiks.solving.mcts_device import MCTSForest
    B, C, c, warm = 6, 400, 0.6, 7
    np.random.seed(21)
    states = np.array([oc.scramble(15 + i, True)[0] for i in range(B)])
    onet = oa.TorchNet(net_gpu, device="cuda")
    forest = MCTSForest(B, C)
    forest.set_net(GenericNet(net_gpu), torch.float32)
    forest.reset(DeviceCubes.from_numpy(states))
    for _ in range(warm + 1):       # a tree's first iteration (its root's) takes two steps
        forest.step(c, C, use_graph=False)
    torch.cuda.synchronize()
    refs, paths = [], []
    for t in range(B):
        ref = oa.MCTS(onet, c=c, search_graph=False)
        assert not ref.search(states[t], C, max_iterations=warm)
        actions = list(ref.action_queue)
        path = [1]
        for a in actions:
            path.append(int(ref.neighbors[path[-1], a]))
        refs.append(ref)
        paths.append((path, actions))
        assert forest.read_path("path_node", t, len(path)).tolist() == path   # same descent on the device
    # plant the unseen children of every tree's current leaf into both trees, as leaf nodes
    planted = 0
    for t, (ref, (path, actions)) in enumerate(zip(refs, paths)):
        leaf = path[-1]
        kids = oc.expand12(ref.states[leaf][None])
        unseen = [k for k in range(12) if kids[k].tobytes() not in ref.indices]
        assert len(unseen) >= 10
        p, v = onet(kids[unseen])
        base, n = t * (forest.C + 1), len(ref)      # forest.C: an on-demand forest rounds its rows per tree up
        table = forest.hash[t].cpu().numpy()
        for i, k in enumerate(unseen):
            idx = n + 1 + i
            ref.indices[kids[k].tobytes()] = idx
            ref.states[idx], ref.P[idx], ref.V[idx], ref.W[idx] = kids[k], p[i], v[i], v[i]
            key = _pack_key(kids[k])
            slot = _key_hash(key) & (forest.hash_size - 1)
            while table[slot] != 0:
                slot = (slot + 1) & (forest.hash_size - 1)
            table[slot] = idx
            forest.keys[base + idx] = torch.from_numpy(key.view(np.int32)).cuda()
            forest.P[base + idx] = torch.from_numpy(p[i].astype(np.float32)).cuda()
            forest.W[base + idx] = float(v[i])
            forest.V[base + idx] = float(v[i])
            forest.leaf[base + idx] = 1
            forest.rec[base + idx] = torch.tensor([0, 0, 1 << 16, 0], dtype=torch.int32).cuda()
        forest.hash[t] = torch.from_numpy(table).cuda()
        forest.n_nodes[t] = n + len(unseen)
        planted += len(unseen)
    assert planted >= 10 * B
    # the next iteration expands a leaf that has NO new child; then a few more on top of the planted nodes
    for extra in (1, 6):
        for _ in range(extra):
            forest.step(c, C, use_graph=False)
        torch.cuda.synchronize()
        for t, ref in enumerate(refs):
            path, actions = paths[t]
            for _ in range(extra):
                n_before = len(ref)
                solved_idx, _ = ref._expand_leaf(path, actions)
                assert solved_idx == -1
                if extra == 1:
                    assert len(ref) == n_before           # nothing new: the defined branch ran
                path, actions = ref._find_leaf()
            paths[t] = (path, actions)
            _compare(forest.tree_arrays(t), ref, len(ref))
            assert forest.read_path("path_node", t, len(path)).tolist() == path


@pytest.mark.parametrize("use_graph", [False, True])
def test_path_overflow(net_gpu, use_graph):
    """A bounded path store of 8 levels.  The oracle has no bound (it is the reference's unbounded descent): it only records its
    deepest descent, which says which trees the bound ends -- a descent that stands on a non-leaf with 8 nodes on its path."""
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    np.random.seed(4)
    states = np.array([oc.scramble(20, True)[0] for _ in range(48)])
    max_path, cap = 8, 1500
    agent = MCTS(net_gpu, c=0.2, search_graph=True, net_dtype=torch.float32, max_path=max_path, use_graph=use_graph)
    res = agent.search_batch(states, None, cap, compact=False)
    assert agent.forest.max_path == max_path and not agent.forest.path_vmm
    onet = oa.TorchNet(net_gpu, device="cuda")
    overflowed = 0
    for t, s in enumerate(states):
        ref = oa.MCTS(onet, c=0.2, search_graph=True)
        ok = ref.search(s, cap)
        too_deep = ref.deepest_path > max_path
        assert (res.status[t] == md.PATH_OVERFLOW) == too_deep, f"tree {t}"
        if too_deep:
            assert not res.solved[t] and len(res.queues[t]) == max_path - 1 and res.nodes[t] <= len(ref)
            overflowed += 1
        else:                      # the bound was never met: the reference's tree, node for node
            assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref), f"tree {t}"
            assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
            assert res.iterations[t] == ref.iterations
    assert overflowed >= 8 and res.path_overflow_trees == overflowed


def test_astar_open_list_runs_dry(net_gpu):
    """Both sides: an iteration that finds nothing on the open list ends the problem as unsolved."""
    from librubiks.cube import DeviceCubes
    from librubiks.model import GenericNet
    from librubiks.solving import astar_device as ad
    np.random.seed(8)
    states = np.array([oc.scramble(7, True)[0] for _ in range(5)])
    N, cap, lam = 3, 2000, 0.2
    batch = ad.AStarBatch(5, cap, N)
    batch.set_net(GenericNet(net_gpu), torch.float32)
    batch.reset(DeviceCubes.from_numpy(states))
    refs = []
    onet = oa.TorchNet(net_gpu, device="cuda")
    for s in states:
        ref = oa.AStar(onet, lam, N)
        assert not ref.search(s, cap, max_iterations=2)
        refs.append(ref)
    for _ in range(2):
        batch.iteration(lam, cap)
    torch.cuda.synchronize()
    nodes_before = batch.n_nodes.cpu().numpy().copy()
    assert [len(r) for r in refs] == nodes_before.tolist()
    # empty the open lists of problems 1 and 3 on both sides
    for b in (1, 3):
        batch.heap_size[b] = 0
        refs[b].open_queue.clear()
    batch.iteration(lam, cap)
    torch.cuda.synchronize()
    status = batch.status.cpu().numpy()
    for b, ref in enumerate(refs):
        cont = ref.resume(cap, max_iterations=1)
        if b in (1, 3):
            assert status[b] == ad.OPEN_EMPTY and cont is False and ref.open_empty
            assert batch.n_nodes[b].item() == nodes_before[b] == len(ref)
        else:
            assert (status[b] == ad.SOLVED) == bool(cont) and status[b] in (ad.RUNNING, ad.SOLVED) and not ref.open_empty
            assert batch.n_nodes[b].item() == len(ref) > nodes_before[b]


def test_time_limit_only_caps_the_tree(net_gpu, monkeypatch):
    from librubiks.solving import agents as pa
    from librubiks.solving import mcts_device as md
    np.random.seed(6)
    states = np.array([oc.scramble(20, True)[0] for _ in range(12)])
    monkeypatch.setattr(pa, "time_only_capacity", lambda n_trees: 700)   # as if the hash-table budget allowed 700 nodes per tree
    agent = pa.MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32)
    res = agent.search_batch(states, time_limit=120.0)
    assert res.seconds < 60
    onet = oa.TorchNet(net_gpu, device="cuda")
    for t, s in enumerate(states):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        ok = ref.search(s, 700)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) and list(res.queues[t]) == list(ref.action_queue)
        if not ok:
            assert res.status[t] == md.EXHAUSTED and res.nodes[t] + 12 > 700


class _TableNet:
    """The oracle's network interface answered from a record {state bytes: (P row, V)}."""

    def __init__(self, table):
        self.table = table

    def __call__(self, states):
        if len(states) == 0:
            return np.zeros((0, 12), dtype=np.float32), np.zeros(0, dtype=np.float32)
        rows = [self.table[s.tobytes()] for s in states]
        return np.stack([r[0] for r in rows]), np.array([r[1] for r in rows], dtype=np.float32)


def test_production_bf16_trees_equal_oracle_on_recorded_outputs():
    import os
    from conftest import ROOT
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    torch.manual_seed(0)
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    net = Model.load(wdir).eval() if os.path.isdir(wdir) else Model.create(ModelConfig()).eval()
    np.random.seed(17)
    B, cap = 96, 900
    states = np.array([oc.scramble(8 + i % 13, True)[0] for i in range(B)])
    for graph_search in (True, False):
        agent = MCTS(net, c=0.6, search_graph=graph_search, net_dtype=torch.bfloat16)   # the fast engine, HIP graph
        assert agent.use_graph
        res = agent.search_batch(states, None, cap, compact=False)    # the trees stay in agent.forest
        assert agent.forest._fused and agent.forest.rows_per_tree == 11
        n_solved = deep = 0
        for t in range(B):
            tree = agent.forest.tree_arrays(t)
            n = tree["n"]
            table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i]))
                     for i in range(1, n + 1)}
            ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=graph_search)
            ok = ref.search(states[t], cap)
            assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n, f"tree {t}"
            assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
            assert res.iterations[t] == ref.iterations
            _compare(tree, ref, n)
            n_solved += ok
            deep += ref.iterations > 40
        assert n_solved >= 10 and deep >= 10


@pytest.mark.parametrize("engine", ["bf16", "f32s"])
def test_deep_production_trees_equal_oracle_on_recorded_outputs(engine):
    """The same replay on DEEP trees (trained weights, depth-20 scrambles, up to 30 000 states: up to ~2 600 iterations, descents of
    several hundred levels): re-validation in several rounds per thread, line following over many 64-level segments, the ring of
    descent lines wrapping around, loops through transpositions -- node for node, incl. the virtual losses of the pending path."""
    import os
    from conftest import ROOT
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    if not os.path.isdir(wdir):
        pytest.skip("needs the trained weights")
    net = Model.load(wdir).eval()
    np.random.seed(5)
    B, cap = 6, 30000
    states = np.array([oc.scramble(20, True)[0] for _ in range(B)])
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=torch.bfloat16 if engine == "bf16" else F32_SPLIT)   # both production engines
    res = agent.search_batch(states, None, cap, compact=False)
    longest = 0
    for t in range(B):
        tree = agent.forest.tree_arrays(t)
        n = tree["n"]
        table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
        ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=True)
        ok = ref.search(states[t], cap)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n, f"tree {t}"
        assert list(res.queues[t]) == list(ref.action_queue) and res.iterations[t] == ref.iterations, f"tree {t}"
        _compare(tree, ref, n)
        longest = max(longest, int(agent.forest.path_len[t].item()))
    assert longest > 256 and int(res.iterations.max()) > 1000      # descents deeper than one re-validation round, ring wrapped many times


@pytest.mark.parametrize("n_trees,unc_cap,offset", [(6, 0, 0.0), (300, 0, 0.0), (600, 0, 0.0), (6, 0, 4096.0), (6, 128, 262144.0)])
def test_unsettled_levels_are_redecided_whatever_the_launch_shape(n_trees, unc_cap, offset):
    """
    Pass B of the tree kernel's re-validation (float64 re-decision of the levels float32 could not settle) has two forms: a dense
    list while at most `unc_list_cap` levels are flagged, a scan of the flag bitmap beyond -- NT / 16 levels per step, i.e. 16, 32
    or 64 at 256 / 512 / 1 024 threads per tree (full forest / <= 512 / <= 256 listed trees).  Round 3's scan looked at the first
    16 flags of a step only, so at 512 and 1 024 threads a step whose flagged levels all lay further back was skipped and a stale
    path prefix kept (the kernel built with that test fails the unc_cap = 0 cases here).
    unc_cap = 0 sends every flagged level through the bitmap: the ~1 % of levels the trained network leaves unsettled (sparse
    flags, what the old test missed), and with an offset of 2^12 on the value head (float32 cannot tell PUCT scores closer than
    ~0.008 apart) several times as many.  The last case needs no knob: with an offset of 2^18 (values keep a resolution of 2^-5,
    scores closer than ~0.5 are unsettled) a deep descent flags more levels than the list's 128.  Every tree must be what the
    oracle builds from the same (state -> P, V) pairs, node for node, in every launch shape.
    """
    import copy
    import os
    from conftest import ROOT
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    if not os.path.isdir(wdir):
        pytest.skip("needs the trained weights")
    net = copy.deepcopy(Model.load(wdir).eval())
    with torch.no_grad():
        net.value_net[-1].bias += offset
    np.random.seed(5)
    cap = 12000 if n_trees <= 6 else 2500
    states = np.array([oc.scramble(20, True)[0] for _ in range(n_trees)])
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT)
    forest = agent._forest_for(n_trees, cap)
    forest.struct.unc_list_cap = unc_cap           # before the first step: the captured graphs carry the struct
    res = agent.search_batch(states, None, cap, compact=False)
    assert agent.forest is forest
    stats = forest.select_stats.cpu().numpy()
    most_unsettled = int((stats[:, 5] >> 16).max())
    assert most_unsettled > (128 if unc_cap else 2), most_unsettled      # the bitmap form was reached
    check = range(n_trees) if n_trees <= 6 else range(0, n_trees, max(1, n_trees // 12))
    longest = 0
    for t in check:
        tree = forest.tree_arrays(t)
        n = tree["n"]
        table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
        ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=True)
        ok = ref.search(states[t], cap)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n, f"tree {t}"
        assert list(res.queues[t]) == list(ref.action_queue) and res.iterations[t] == ref.iterations, f"tree {t}"
        _compare(tree, ref, n)
        longest = max(longest, int(forest.path_len[t].item()))
    assert longest > 64


@pytest.mark.parametrize("engine", ["f32s", "bf16"])
def test_production_trees_through_refill_narrowing_and_results_forest_equal_oracle(engine):
    """
    The continuous-batching path at production precision (trained weights, both engines): 112 scrambles on 48 tree slots, so
    finished trees hand their slots to waiting scrambles (rc_mcts_plant into a running forest), are copied into the results
    forest (65 B per node) and post-processed there; when nobody is waiting the forest is narrowed (MCTSForest.set_active) and
    the last trees are post-processed where they lie.  Every tree is recorded as its search left it; the oracle is replayed on
    the tree's own (state -> P, V) pairs and must build the same tree -- nodes, neighbours, N, W, L, P, V, leaves -- and, after
    its own graph completion + BFS shortening (agents.py:597-633), return the same action queue the device pipeline returned.
    """
    import os
    from conftest import ROOT
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    if not os.path.isdir(wdir):
        pytest.skip("needs the trained weights")
    net = Model.load(wdir).eval()
    np.random.seed(23)
    G, S, cap = 112, 48, 5000
    states = np.array([oc.scramble(13 + i % 8, True)[0] for i in range(G)])
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=torch.bfloat16 if engine == "bf16" else F32_SPLIT, sync_every=8)
    agent.snapshot_trees = {}
    res = agent.search_batch(states, None, cap, slots=S)
    st = agent.refill_stats
    assert st["refills"] >= 2 and st["compactions"] >= 1 and st.get("flushes", 0) >= 2      # every path was taken
    assert sorted(agent.snapshot_trees) == list(range(G))
    solved = unsolved = 0
    for g in range(G):
        tree = agent.snapshot_trees[g]
        n = tree["n"]
        table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
        ref = oa.MCTS(_TableNet(table), c=0.6, search_graph=True)
        before = {}
        complete = ref._complete_graph

        def recording_complete():
            before["neighbors"] = ref.neighbors.copy()    # the device tree was recorded before its graph completion
            complete()

        ref._complete_graph = recording_complete
        ok = ref.search(states[g], cap)
        assert bool(res.solved[g]) == ok and res.nodes[g] == len(ref) == n, f"game {g}"
        assert res.iterations[g] == ref.iterations, f"game {g}"
        assert list(res.queues[g]) == list(ref.action_queue), f"game {g}"          # after completion + shortening in the results forest
        assert res.lengths[g] == (len(ref.action_queue) if ok else -1)
        after = ref.neighbors
        if "neighbors" in before:
            ref.neighbors = before["neighbors"]
        _compare(tree, ref, n)
        ref.neighbors = after
        solved += ok
        unsolved += not ok
    assert solved >= 40 and unsolved >= 1      # both outcomes occur (the cap stops the hardest scrambles)


@pytest.mark.parametrize("what", ["neighbour", "hash"])
def test_rows_that_are_not_the_trees_data_raise_instead_of_faulting(what, net_gpu, agents_golden):
    """rc_mcts_complete_graph / rc_mcts_shorten follow indices they read from memory (hash slots, neighbour rows).  An index that
    names no node of the tree -- rows that are not the tree's data, the round-4 fault class -- marks the tree RC_MCTS_CORRUPT and
    result extraction raises with the rows' addresses, instead of the kernel indexing with it."""
    from librubiks import _hip
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    cases = ["d4_s17_c20_graph", "d5_s1_c20_graph", "d7_s25_c20_graph"]          # recorded by the reference: all solved at c = 20
    states = np.array([agents_golden[f"mcts_{c}_state"] for c in cases])
    agent = MCTS(net_gpu, c=20.0, search_graph=True, net_dtype=torch.float32)
    run = agent.start_batch(states, None, 2500, compact=False)
    while not run.done:
        run.round()
    torch.cuda.synchronize()
    f = run.forest
    assert (f.status.cpu().numpy() == md.SOLVED).all()
    t = 1
    if what == "neighbour":
        f.nbr[t * (f.C + 1) + 1, 3] = 0x7FFFFF00      # the root's neighbour through action 3: far outside the tree
    else:
        filled = torch.nonzero(f.hash[t])[:, 0]
        f.hash[t, filled] = 0x7FFFFF00                # every occupied slot names a node that does not exist
    with pytest.raises(_hip.RubiksHipError, match="not theirs") as err:
        run.finish()
    assert "nbr @0x" in str(err.value) and "tree 1" in str(err.value)
    torch.cuda.synchronize()                          # the process is alive and the GPU answers
    assert int(f.status[t].item()) == md.CORRUPT and int(f.status[0].item()) == md.SOLVED
