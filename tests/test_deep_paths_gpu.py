"""
PUCT descents of any length (reference: librubiks/solving/agents.py:575-595 walks until it meets a leaf; its node arrays and
its Python lists grow as needed, agents.py:450-459).

rc_mcts_select works on the first `lds_levels` levels of a path in LDS and on deeper ones where they lie in the blocked path
arrays in HBM, which get memory a block at a time (`MCTSForest.ensure_path`); the first `ring_levels` levels of a path are kept
as a line.  In production the three numbers are 4 096 / 4 096 / 4 096, far beyond the trees the reference's recorded traces
reach, so the tests shrink them (module knobs RUBIKS_LDS_LEVELS / RUBIKS_PATH_BLOCK / RUBIKS_RING_LEVELS): with 8 levels in LDS
every recorded tree of the REFERENCE ITSELF (tests/golden/agents_golden.npz: deepest descents 358 levels in d20_graph, 95 in
d24_graph, 43 in d20_naive_c4) crosses the window, the block boundaries and the end of the ring lines -- and must still be
rebuilt node for node.  Then the production numbers on a tree whose descents really are deeper than 4 096 levels.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_cases

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

_G = np.load(f"{GOLDEN}/agents_golden.npz")

# (levels in LDS, levels per path block, levels per ring line)
KNOBS = [(8, 16, 32), (64, 64, 4096), (8, 4096, 4096), (1, 2, 1)]


@pytest.fixture(scope="module")
def net_gpu(standin_net):
    return standin_net.cuda()


@pytest.fixture
def knobs(monkeypatch):
    from librubiks.solving import mcts_device as md

    def set_(lds, block, ring):
        monkeypatch.setattr(md, "LDS_LEVELS", lds)
        monkeypatch.setattr(md, "PATH_BLOCK", block)
        monkeypatch.setattr(md, "RING_LEVELS", ring)
    return set_


def _compare_tree(tree: dict, ref: dict, n: int):
    assert tree["n"] == n
    assert np.array_equal(tree["states"][1:n + 1], ref["states"][1:n + 1])
    assert np.array_equal(tree["neighbors"][:n + 1], ref["neighbors"][:n + 1])
    assert np.array_equal(tree["leaves"][1:n + 1], ref["leaves"][1:n + 1])
    assert np.array_equal(tree["N"][:n + 1], ref["N"][:n + 1])
    assert np.array_equal(tree["L"][:n + 1], ref["L"][:n + 1])
    assert np.array_equal(tree["V"][1:n + 1], np.asarray(ref["V"][1:n + 1], dtype=np.float64))
    assert np.array_equal(tree["W"][1:n + 1], ref["W"][1:n + 1])
    assert np.allclose(tree["P"][1:n + 1], ref["P"][1:n + 1], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", golden_cases(_G, "mcts_"))
@pytest.mark.parametrize("lds,block,ring", KNOBS)
@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_traces_through_a_small_window(case, lds, block, ring, use_graph, agents_golden, net_gpu, knobs):
    """Every tree the reference recorded, rebuilt with most of its descents outside the LDS window / the first path block /
    the ring lines: node for node the reference's, including L (the pending path) and the action queue."""
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    knobs(lds, block, ring)
    g = lambda k: agents_golden[f"mcts_{case}_{k}"]   # noqa: E731
    depth, c, graph, max_states, solved, n = g("params")
    agent = MCTS(net_gpu, c=float(c), search_graph=bool(graph), net_dtype=torch.float32, use_graph=use_graph)
    assert agent.search(g("state"), None, int(max_states)) == bool(solved)
    f = agent._last_forest
    assert (f.lds_levels, f.path_block) == (lds, block) and f.ring_levels == min(ring, f.max_path) and f.path_vmm
    assert f.max_path >= md.MAX_PATH_LEVELS // 2, "the default path store is address space for descents of any length"
    assert len(agent) == int(n)
    assert list(agent.action_queue) == list(g("queue"))
    ref = {k: g(k) for k in ("states", "neighbors", "leaves", "N", "L", "V", "W", "P")}
    _compare_tree(agent._host_tree(), ref, int(n))
    deepest = {"d20_graph": 358, "d24_graph": 95, "d20_naive_c4": 43}.get(case, 0)   # levels of the reference's deepest descent there
    if deepest > block:   # ... which did leave the first block (and with it the LDS window, which is never larger here)
        assert int(f.path_rows_host.max()) > block


@pytest.mark.parametrize("lds,block,ring", [(8, 16, 32), (4, 8, 8)])
def test_batch_with_blocks_arriving_late(lds, block, ring, net_gpu, knobs):
    """A lock-step batch whose trees outgrow their path blocks all the time (16-level blocks, the host looks every few
    iterations): descents are suspended at the end of their memory and resume when the next block is there.  Every tree is the
    oracle's single-tree run -- continuous batching (slots < games) and narrowing included."""
    from librubiks.solving.agents import MCTS
    knobs(lds, block, ring)
    np.random.seed(11)
    states = np.array([oc.scramble(3 + i % 18, True)[0] for i in range(72)])
    max_states = 900
    onet = oa.TorchNet(net_gpu, device="cuda")
    refs = []
    for s in states:
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        refs.append((ref.search(s, max_states), ref))
    assert max(r.deepest_path for _, r in refs) > 4 * block
    for slots in (None, 24):
        agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=4)
        res = agent.search_batch(states, None, max_states, slots=slots)
        assert res.path_overflow_trees == 0
        for t, (ok, ref) in enumerate(refs):
            assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref), f"tree {t}"
            assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
            assert res.iterations[t] == ref.iterations, f"tree {t}"
        assert int(agent._last_forest.path_rows_host.max()) > 2 * block   # blocks did arrive while the search ran


class ClockNet(torch.nn.Module):
    """A network whose value output rises with every call (and, within a call, with the row): the newest leaf is always the best,
    so every iteration's descent goes one level deeper than the one before -- the deepest tree a search can build.  Not a model of
    anything: it makes descents of thousands of levels in thousands of iterations.  (The same launch sequence gives the same
    outputs, so two searches of the same batch see the same numbers.)"""

    def __init__(self):
        super().__init__()
        self.register_buffer("clock", torch.zeros((), dtype=torch.float32))

    def forward(self, x, policy=True, value=True):
        n = x.shape[0]
        self.clock += 1.0
        v = (self.clock + (torch.arange(n, device=x.device) % 11).float() / 16.0).reshape(-1, 1)
        p = torch.zeros((n, 12), dtype=torch.float32, device=x.device)
        out = ([p] if policy else []) + ([v] if value else [])
        return out if len(out) > 1 else out[0]


def _deep_search(knobs_, lds, block, ring, iterations, premap):
    from librubiks.solving.agents import MCTS
    knobs_(lds, block, ring)
    np.random.seed(2)
    states = np.array([oc.scramble(20, True)[0] for _ in range(3)])
    net = ClockNet().cuda()
    agent = MCTS(net, c=0.6, search_graph=False, net_dtype=torch.float32, use_graph=False)
    run = agent.start_batch(states, None, 12 * iterations + 64, compact=False, one_launch=True)
    run.forest.ensure_path(np.arange(3), np.full(3, premap))   # blocks up front: no tree ever waits, so both runs see the same clock
    while not run.done:
        run.round()
    res = run.finish()
    f = agent._last_forest
    trees = [f.tree_arrays(t) for t in range(3)]
    plen = f.path_len.cpu().numpy()
    paths = [(f.read_path("path_node", t, plen[t]), f.read_path("path_act", t, plen[t] - 1)) for t in range(3)]
    return res, trees, paths, f


def test_descents_deeper_than_the_lds_window_at_production_sizes(knobs):
    """Production numbers (4 096 levels in LDS, 4 096-level blocks and ring lines) on trees whose descents reach ~6 000 levels:
    levels beyond 4 096 are walked in HBM, the second path block is in use.  No oracle finishes a 6 000-level tree in seconds,
    so the check is the deep-path code against itself at the sizes the reference's own traces pin above (8 levels in LDS,
    64-level blocks): the same trees, paths and queues, bit for bit -- plus what must hold of any such tree."""
    iters = 6000
    res_a, trees_a, paths_a, fa = _deep_search(knobs, 4096, 4096, 4096, iters, 8192)
    assert (fa.lds_levels, fa.path_block, fa.ring_levels) == (4096, 4096, 4096)
    assert res_a.path_overflow_trees == 0
    deepest = max(len(p[0]) for p in paths_a)
    assert deepest > 5000, deepest                      # well beyond the LDS window and the first block
    res_b, trees_b, paths_b, fb = _deep_search(knobs, 8, 64, 64, iters, 8192)
    assert (fb.lds_levels, fb.path_block) == (8, 64)
    for t in range(3):
        assert np.array_equal(res_a.nodes, res_b.nodes) and np.array_equal(res_a.iterations, res_b.iterations)
        assert list(res_a.queues[t]) == list(res_b.queues[t]) and len(res_a.queues[t]) == len(paths_a[t][1])
        for k in ("states", "neighbors", "leaves", "N", "L", "V", "W", "P"):
            assert np.array_equal(trees_a[t][k], trees_b[t][k]), (t, k)
        assert np.array_equal(paths_a[t][0], paths_b[t][0]) and np.array_equal(paths_a[t][1], paths_b[t][1])
        # the pending path is a path of the tree: consecutive nodes are neighbours through the recorded actions, from the root to a leaf
        nodes, acts = paths_a[t]
        nb = trees_a[t]["neighbors"]
        assert nodes[0] == 1 and np.array_equal(nb[nodes[:-1], acts], nodes[1:]) and trees_a[t]["leaves"][nodes[-1]]
        assert not trees_a[t]["leaves"][nodes[:-1]].any()
        # the unsolved tree's queue is its last descent (agents.py:492), all of it
        assert list(res_a.queues[t]) == list(acts)


def test_a_search_bounded_by_time_alone_grows_past_the_old_node_cap(net_gpu):
    """The reference doubles its node arrays for as long as the time limit lets a tree grow (agents.py:450-459).  A search with a
    time limit only gets the capacity the kernels can address (node rows are address space with memory behind the rows in use),
    not the 2^18 nodes of earlier rounds: one tree, six seconds, more than 2^18 nodes -- and still a consistent tree."""
    from librubiks.solving import agents as ag
    from librubiks.solving import mcts_device as md
    assert ag.time_only_capacity(1) == md.MAX_CAPACITY and ag.time_only_capacity(1024) >= 4 * (1 << 18)
    np.random.seed(5)
    state = oc.scramble(30, True)[0]
    agent = ag.MCTS(net_gpu, c=0.6, search_graph=False, net_dtype=torch.float32)
    agent.prepare(1, None)
    solved = agent.search(state, time_limit=6.0)            # (one tree: ~280 k nodes in 2 s, ~700 k in 6 s on an MI355X)
    f = agent._last_forest
    assert f.C >= md.MAX_CAPACITY - 8192 and f.vmm
    n = len(agent)
    if solved:
        pytest.skip("the stand-in net solved this scramble before the tree reached 2^18 nodes")
    assert n > (1 << 18), n
    assert f.bytes_mapped() < 2 * n * 300 + (64 << 20)        # memory follows the tree, not the capacity
    tree = agent._host_tree()
    nb = tree["neighbors"][:n + 1]
    i, a = np.nonzero(nb[1:])
    i = i + 1
    assert (nb[i, a] <= n).all() and np.array_equal(nb[nb[i, a], a ^ 1], i)          # links both ways (agents.py:533-535)
    pick = np.random.RandomState(0).choice(len(i), 4096, replace=False)
    kids = oc.multi_rotate(tree["states"][i[pick]], *oc.indices_to_actions(a[pick]))
    assert np.array_equal(kids, tree["states"][nb[i[pick], a[pick]]])                 # ... between the right states
    assert len(np.unique(tree["states"][1:n + 1], axis=0)) == n                        # every state once (agents.py:517-529)
    assert (tree["leaves"][1:n + 1] == (nb[1:] == 0).any(axis=1)).all()


def test_time_only_search_with_continuous_batching_copies_trees_into_small_forests(net_gpu):
    """A pool of games bounded by time alone, on fewer slots than games: the forest's capacity is address-space sized (2^24 - 2 rows
    per tree), so finished trees are copied into results forests of the capacity they NEED (hash tables rebuilt from the keys,
    rc_mcts_copy_trees) -- not into gigabyte copies of the big one.  Shallow scrambles that the stand-in net solves: every game is the
    oracle's search, the inspectable tree of game 0 included."""
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    np.random.seed(31)
    onet = oa.TorchNet(net_gpu, device="cuda")
    states = []
    while len(states) < 40:                      # scrambles the stand-in net does solve (a search bounded by time alone ends no other way)
        cand = oc.scramble(1 + len(states) % 4, True)[0]
        if oa.MCTS(onet, c=20.0, search_graph=True).search(cand, 3000):
            states.append(cand)
    states = np.array(states)
    agent = MCTS(net_gpu, c=20.0, search_graph=True, net_dtype=torch.float32, sync_every=4)
    free0 = torch.cuda.mem_get_info()[0]
    res = agent.search_batch(states, time_limit=60.0, slots=8)
    assert agent.forest.C > 1 << 20 and agent.forest.vmm                      # the searching forest: capacity by address space
    assert free0 - torch.cuda.mem_get_info()[0] < 24 << 30                    # ... and nothing near 8 x 4.5 GB of rows, or a 256 x 1 GB grave
    assert res.solved.all() and res.seconds < 50
    for t, s in enumerate(states):
        ref = oa.MCTS(onet, c=20.0, search_graph=True)
        assert ref.search(s, 100000)
        assert res.nodes[t] == len(ref) and list(res.queues[t]) == list(ref.action_queue), f"game {t}"
    tree = agent._host_tree()                                                  # game 0, read from the small forest it was copied into
    ref0 = oa.MCTS(onet, c=20.0, search_graph=True)
    ref0.search(states[0], 100000)
    assert tree["n"] == len(ref0) and np.array_equal(tree["neighbors"][:tree["n"] + 1], ref0.neighbors[:len(ref0) + 1])
    assert agent._tree_src[0].C <= 1 << 14


def test_large_forest_form_of_the_deep_kernel(net_gpu, knobs):
    """Forests of more than 512 trees run the tree kernel with 256 threads per tree and ONE wave checking a line (smaller ones: 512 / 1 024
    threads, four waves): the same deep-path body in its other launch shape.  640 trees with 8 levels in LDS and 16-level path blocks, every
    fifth tree against the oracle's single-tree run."""
    from librubiks.solving.agents import MCTS
    knobs(8, 16, 32)
    np.random.seed(17)
    states = np.array([oc.scramble(2 + i % 19, True)[0] for i in range(640)])
    max_states = 420
    agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=4)
    res = agent.search_batch(states, None, max_states, compact=False)        # the forest keeps its 640-tree launch shape to the end
    assert res.path_overflow_trees == 0 and int(agent._last_forest.path_rows_host.max()) > 16
    onet = oa.TorchNet(net_gpu, device="cuda")
    deepest = 0
    for t in range(0, 640, 5):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        ok = ref.search(states[t], max_states)
        deepest = max(deepest, ref.deepest_path)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) and res.iterations[t] == ref.iterations, f"tree {t}"
        assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
    assert deepest > 16
