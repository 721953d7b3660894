"""
GPU parity tests of the batched MCTS (rc_mcts_* kernels behind librubiks.solving.agents.MCTS).

1. Golden traces: every tree recorded from the REFERENCE agent (tests/golden/agents_golden.npz) is
   rebuilt node-for-node by the HIP path (fp32 stand-in network whose outputs are exact integers/16).
2. Batched vs oracle: B independent scrambles in one lock-step batch, each compared with the
   restated single-tree agent (oracle/agents.py) fed by the same network on the same device.
3. The reference's own structural invariants (tests/test_agents.py:38-94) restated, with the real
   fc_small network on the bf16 engine.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_cases

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

_G = np.load(f"{GOLDEN}/agents_golden.npz")


@pytest.fixture(scope="module")
def net_gpu(standin_net):
    return standin_net.cuda()


def _compare_tree(tree: dict, ref: dict, n: int, exact_p: bool):
    assert tree["n"] == n
    assert np.array_equal(tree["states"][1:n + 1], ref["states"][1:n + 1])
    assert np.array_equal(tree["neighbors"][:n + 1], ref["neighbors"][:n + 1])
    assert np.array_equal(tree["leaves"][1:n + 1], ref["leaves"][1:n + 1])
    assert np.array_equal(tree["N"][:n + 1], ref["N"][:n + 1])
    assert np.array_equal(tree["L"][:n + 1], ref["L"][:n + 1])
    assert np.array_equal(tree["V"][1:n + 1], np.asarray(ref["V"][1:n + 1], dtype=np.float64))   # integers / 16: exact
    assert np.array_equal(tree["W"][1:n + 1], ref["W"][1:n + 1])
    if exact_p:
        assert np.array_equal(tree["P"][1:n + 1], np.asarray(ref["P"][1:n + 1], dtype=np.float64))
    else:   # softmax evaluated by torch on another device: float32 rounding only
        assert np.allclose(tree["P"][1:n + 1], ref["P"][1:n + 1], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", golden_cases(_G, "mcts_"))
@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trace(case, use_graph, agents_golden, net_gpu):
    from librubiks.solving.agents import MCTS
    g = lambda k: agents_golden[f"mcts_{case}_{k}"]   # noqa: E731
    depth, c, graph, max_states, solved, n = g("params")
    agent = MCTS(net_gpu, c=float(c), search_graph=bool(graph), net_dtype=torch.float32, use_graph=use_graph)
    assert agent.search(g("state"), None, int(max_states)) == bool(solved)
    assert len(agent) == int(n)
    assert list(agent.action_queue) == list(g("queue"))
    ref = {k: g(k) for k in ("states", "neighbors", "leaves", "N", "L", "V", "W", "P")}
    tree = agent._host_tree()
    _compare_tree(tree, ref, int(n), exact_p=False)
    assert agent.indices[g("state").tobytes()] == 1 and len(agent.indices) == int(n)


def test_batch_vs_oracle(net_gpu):
    """96 scrambles of depth 1..8 in one lock-step batch; every tree equals the oracle's single-tree run."""
    from librubiks.solving.agents import MCTS
    np.random.seed(5)
    states = np.array([oc.scramble(1 + i % 8, True)[0] for i in range(96)])
    states[17] = oc.get_solved()          # a solved root in the middle of the batch
    max_states = 700
    for c, graph in ((0.6, True), (4.13, False)):
        agent = MCTS(net_gpu, c=c, search_graph=graph, net_dtype=torch.float32)
        res = agent.search_batch(states, None, max_states, compact=False)   # the trees stay in agent.forest for inspection
        onet = oa.TorchNet(net_gpu, device="cuda")
        n_solved = 0
        for t, s in enumerate(states):
            ref = oa.MCTS(onet, c=c, search_graph=graph)
            ok = ref.search(s, max_states)
            assert bool(res.solved[t]) == ok, f"tree {t}"
            assert res.nodes[t] == len(ref), f"tree {t}"
            assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
            assert res.lengths[t] == (len(ref.action_queue) if ok else -1)
            if t % 8 == 3 and not oc.is_solved(s):
                tree = agent.forest.tree_arrays(t)   # solved graph-search trees were completed on the device
                refd = {k: getattr(ref, k) for k in ("states", "neighbors", "leaves", "N", "L", "V", "W", "P")}
                _compare_tree(tree, refd, len(ref), exact_p=True)
            if ok:
                n_solved += 1
                x = s
                for a in res.queues[t]:
                    x = oc.rotate(x, *oc.ACTION_SPACE[a])
                assert oc.is_solved(x)
        assert n_solved >= 20
        assert res.solved[17] and res.lengths[17] == 0 and res.nodes[17] == 1


@pytest.mark.parametrize("budget", [1, 3, 48])
def test_level_budget_does_not_change_trees(budget, net_gpu):
    """Suspended descents (rc_mcts_select level budget) only re-time iterations: every tree is still the oracle's."""
    from librubiks.solving.agents import MCTS
    np.random.seed(11)
    states = np.array([oc.scramble(2 + i % 9, True)[0] for i in range(72)])
    agent = MCTS(net_gpu, c=4.13, search_graph=True, net_dtype=torch.float32, level_budget=budget)
    res = agent.search_batch(states, None, 900, compact=False)
    assert agent.forest.level_budget == budget
    onet = oa.TorchNet(net_gpu, device="cuda")
    for t, s in enumerate(states):
        ref = oa.MCTS(onet, c=4.13, search_graph=True)
        ok = ref.search(s, 900)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref), f"tree {t}"
        assert list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
        assert res.iterations[t] == ref.iterations, f"tree {t}"
        if t % 9 == 4:
            refd = {k: getattr(ref, k) for k in ("states", "neighbors", "leaves", "N", "L", "V", "W", "P")}
            _compare_tree(agent.forest.tree_arrays(t), refd, len(ref), exact_p=True)


def test_max_iterations_and_budget(net_gpu):
    """Unsolved trees stop exactly where the reference's `len + 12 <= max_states` loop stops."""
    from librubiks.solving.agents import MCTS
    np.random.seed(9)
    states = np.array([oc.scramble(20, True)[0] for _ in range(40)])
    agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=7)
    res = agent.search_batch(states, None, 333)
    onet = oa.TorchNet(net_gpu, device="cuda")
    for t in (0, 7, 39):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        assert not ref.search(states[t], 333)
        assert res.nodes[t] == len(ref) and res.iterations[t] == ref.iterations
        assert list(res.queues[t]) == list(ref.action_queue)
    assert (res.nodes <= 333).all() and (res.nodes + 12 > 333).all()
    assert not res.solved.any() and (res.lengths == -1).all()


@pytest.mark.parametrize("graph_steps", [1, 3, 4])
def test_iterations_per_graph_launch_do_not_change_a_search(net_gpu, graph_steps, monkeypatch):
    """`MCTSForest.steps` sends the iterations of a round out several to a graph launch (GRAPH_STEPS, the remainder one by one): a
    packaging of the same launches.  With 1, 3 and 4 iterations per graph, rounds of 7 (never a multiple) and an iteration bound that
    ends a round early, every tree is the reference's: node count, iterations, solution, and the arrays of two trees."""
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    monkeypatch.setattr(md.MCTSForest, "GRAPH_STEPS", graph_steps)
    np.random.seed(21)
    states = np.array([oc.scramble(int(d), True)[0] for d in np.random.randint(3, 9, size=24)])
    agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=7)
    res = agent.search_batch(states, None, 600)
    assert graph_steps == 1 or any(len(k) == 6 and k[5] == graph_steps for k in agent.forest._graphs)   # the multi-iteration graph was used
    onet = oa.TorchNet(net_gpu, device="cuda")
    for t in range(0, 24, 5):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        assert ref.search(states[t], 600) == bool(res.solved[t])
        assert res.nodes[t] == len(ref) and res.iterations[t] == ref.iterations and list(res.queues[t]) == list(ref.action_queue), t
    bounded = agent.search_batch(states, None, 600, max_iterations=9)
    for t in (1, 13):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        ref.search(states[t], 600, max_iterations=9)
        assert bounded.nodes[t] == len(ref) and bounded.iterations[t] == ref.iterations, t


def test_structural_invariants_real_net():
    """tests/test_agents.py:38-94 of the reference, on fc_small with the bf16 inference engine."""
    from librubiks import cube
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig()).eval()
    for depth, graph in ((50, False), (3, False), (3, True)):
        state, _, _ = cube.scramble(depth)
        agent = MCTS(net, c=1, search_graph=graph)
        solved = agent.search(state, None, 1500)
        idx = agent.indices
        assert idx[state.tobytes()] == 1
        vals = sorted(idx.values())
        assert vals[0] == 1 and np.all(np.diff(vals) == 1) and len(vals) == len(agent)
        used = np.array(vals)
        states = agent.states
        for s, i in idx.items():
            assert states[i].tobytes() == s
        assert np.array_equal(states[1], state)
        if not graph:
            kids = oc.expand12(states[used]).reshape(len(used), 12, 20)
            nb = agent.neighbors[used]
            assert ((nb == 0) | (nb <= len(agent))).all()
            for r, c_ in zip(*np.nonzero(nb)):
                assert np.array_equal(states[nb[r, c_]], kids[r, c_])
            assert np.all(agent.neighbors[used].all(axis=1) != agent.leaves[used])
        # P and V are the network's outputs on the stored states; bf16 weights/activations vs the fp32
        # module: tolerance 3e-2 absolute (values are O(1) for a glorot-initialised net)
        with torch.no_grad():
            p, v = net(cube.as_oh(states[used]))
        p, v = p.softmax(dim=1).cpu().numpy(), v.squeeze().cpu().numpy()
        assert np.allclose(agent.P[used], p, atol=3e-2)
        assert np.allclose(agent.V[used], v, atol=3e-2)
        assert agent.W[used].all()
        x = state
        assert all(0 <= a < 12 for a in agent.action_queue)
        for a in agent.action_queue:
            x = cube.rotate(x, *cube.action_space[a])
        assert cube.is_solved(x) == solved


def test_time_limit_only():
    """A search bounded by wall time alone stops, reports unsolved trees and a best-guess queue."""
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    torch.manual_seed(0)
    np.random.seed(1)
    net = Model.create(ModelConfig()).eval()
    states = np.array([oc.scramble(20, True)[0] for _ in range(8)])
    agent = MCTS(net, c=0.6, search_graph=True)
    res = agent.search_batch(states, time_limit=0.5, max_states=20000)
    assert res.seconds < 30 and (res.nodes > 12).all()
    assert str(agent) == "BFS MCTS (c=0.6)"


def test_fused_head_backup_matches_generic_path():
    """
    rc_mcts_backup_select_head (one kernel: softmax inside, bf16 head output, path backup fused into the re-validation)
    vs the generic three-kernel path (torch softmax, rc_mcts_backup, rc_mcts_select): after three iterations from the
    same roots the stored P agree to 1e-6 and V / W / N / paths exactly.
    """
    import ctypes
    from librubiks import _hip, cube
    from librubiks.model import InferenceNet, Model, ModelConfig
    from librubiks.solving.mcts_device import MCTSForest
    torch.manual_seed(0)
    np.random.seed(3)
    net = Model.create(ModelConfig()).eval()
    cubes, _, _ = cube.scramble_batch(256, 12, True)
    eng = InferenceNet(net, torch.bfloat16)
    a, b = MCTSForest(256, 64), MCTSForest(256, 64)
    for f in (a, b):
        f.set_net(eng)
        f.reset(cubes)
    st, m = _hip.stream_ptr(), ctypes.byref(b.struct)
    for _ in range(3):                             # the root's two steps, then an ordinary iteration
        a._iteration(0.6, 64)                      # fused: head -> backup + select in one kernel
        _hip.check(b.lib.rc_mcts_expand(m, 64, st))
        head = eng.head_cubes(b.children, b._x1).float()
        probs, values = torch.softmax(head[:, :12], dim=1).contiguous(), head[:, 12].contiguous()
        _hip.check(b.lib.rc_mcts_backup(m, probs.data_ptr(), values.data_ptr(), st))
        _hip.check(b.lib.rc_mcts_select(m, 0.6, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(a.n_nodes, b.n_nodes) and torch.equal(a.nbr, b.nbr)
    assert torch.equal(a.V, b.V) and torch.equal(a.W, b.W) and torch.equal(a.N, b.N)
    assert torch.allclose(a.P, b.P, rtol=0, atol=1e-6)
    assert torch.equal(a.path_len, b.path_len)


def test_compaction_is_invisible(net_gpu):
    """Dropping finished trees from the launches (MCTSForest.set_active) must not change any per-game result."""
    from librubiks.solving.agents import MCTS
    np.random.seed(13)
    states = np.array([oc.scramble(1 + i % 10, True)[0] for i in range(600)])
    res = {}
    for compact in (False, True):
        agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=4)
        res[compact] = agent.search_batch(states, None, 1800, compact=compact)
        if compact:
            assert agent.refill_stats["compactions"] >= 2 and agent._last_forest.G < 600      # the batch really was narrowed
    a, b = res[False], res[True]
    assert np.array_equal(a.solved, b.solved) and np.array_equal(a.nodes, b.nodes)
    assert np.array_equal(a.lengths, b.lengths) and np.array_equal(a.iterations, b.iterations)
    assert np.array_equal(a.status, b.status)
    assert all(list(x) == list(y) for x, y in zip(a.queues, b.queues))
    assert 0.2 < a.solved.mean() < 1.0


def test_one_launch_iterations_equal_the_three_phase_form(net_gpu):
    """rc_mcts_plant_expanded + [network -> rc_mcts_step*] (the expansion of the next leaf at the END of a step, by the wave that
    walked to it) against [rc_mcts_expand -> network -> rc_mcts_backup_select*]: the same trees, game for game, with the exact
    stand-in net and with the production engine; slots refilled in a running forest included."""
    import os
    from conftest import ROOT
    from librubiks.model import Model
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    np.random.seed(31)
    states = np.array([oc.scramble(2 + i % 9, True)[0] for i in range(150)])
    nets = [(net_gpu, torch.float32, 900)]
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    if os.path.isdir(wdir):
        nets.append((Model.load(wdir).eval(), torch.bfloat16, 2500))
    for net, dt, cap in nets:
        out = {}
        for one in (True, False):
            md.MCTSForest.fused_step = one
            try:
                agent = MCTS(net, c=0.6, search_graph=True, net_dtype=dt, sync_every=4)
                out[one] = (agent.search_batch(states, None, cap, compact=False), agent.search_batch(states, None, cap, slots=40))
                assert agent.forest._one_launch == one
            finally:
                md.MCTSForest.fused_step = True
        for a, b in zip(out[True], out[False]):
            assert np.array_equal(a.solved, b.solved) and np.array_equal(a.nodes, b.nodes) and np.array_equal(a.lengths, b.lengths)
            assert np.array_equal(a.iterations, b.iterations) and np.array_equal(a.status, b.status)
            assert all(list(x) == list(y) for x, y in zip(a.queues, b.queues))
        assert 0.1 < out[True][0].solved.mean() < 1.0


def test_one_launch_search_stopped_midway_ends_on_completed_iterations(net_gpu):
    """A one-launch search that is stopped while trees are running (what a time limit does) has expanded one leaf more than it
    has backed up; `MCTSRun.finish` closes those iterations (network on the pending rows, backup, the descent that follows, no new
    expansion).  S one-launch steps + the closing step must leave every tree exactly where S iterations of the three-phase form
    leave it (`max_iterations=S`) -- nodes, N, W, P, V, neighbours, the pending path's virtual losses -- and both equal the oracle
    after S iterations."""
    from librubiks.solving.agents import MCTS
    np.random.seed(41)
    B, cap, S = 40, 600, 23
    states = np.array([oc.scramble(6 + i % 9, True)[0] for i in range(B)])
    for dt_net in ((net_gpu, torch.float32),):
        net, dt = dt_net
        a = MCTS(net, c=0.6, search_graph=False, net_dtype=dt, sync_every=4)
        run = a.start_batch(states, None, cap, compact=False, one_launch=True)
        while run.it < S:
            run.round(S - run.it)
        assert run.it == S and bool(run.forest.expanded.any().item())          # expansions are pending
        ra = run.finish()
        assert not bool(run.forest.expanded.any().item())
        b = MCTS(net, c=0.6, search_graph=False, net_dtype=dt, sync_every=4)
        rb = b.search_batch(states, None, cap, max_iterations=S, compact=False)
        assert not b.forest._one_launch and a.forest._one_launch
        assert np.array_equal(ra.nodes, rb.nodes) and np.array_equal(ra.iterations, rb.iterations) and np.array_equal(ra.status, rb.status)
        assert np.array_equal(ra.solved, rb.solved) and all(list(x) == list(y) for x, y in zip(ra.queues, rb.queues))
        assert (ra.status == 0).sum() >= B // 2 and ra.iterations.max() == S      # most trees were stopped mid-search
        onet = oa.TorchNet(net, device="cuda")
        for t in range(B):
            ta, tb = a.forest.tree_arrays(t), b.forest.tree_arrays(t)
            assert ta["n"] == tb["n"]
            for k in ("states", "neighbors", "P", "V", "W", "N", "L", "leaves"):
                assert np.array_equal(ta[k][1:], tb[k][1:]), (t, k)   # row 0 is no node: padded launch rows are written there
            if t % 8 == 0:
                ref = oa.MCTS(onet, c=0.6, search_graph=False)
                ref.search(states[t], cap, max_iterations=S)
                n = len(ref)
                assert ta["n"] == n and np.array_equal(ta["N"][:n + 1], ref.N[:n + 1]) and np.array_equal(ta["L"][:n + 1], ref.L[:n + 1])
                assert np.array_equal(ta["W"][1:n + 1], ref.W[1:n + 1]) and np.array_equal(ta["neighbors"][:n + 1], ref.neighbors[:n + 1])


def test_one_launch_entry_points_check_their_arguments(net_gpu):
    """rc_mcts_plant_expanded / rc_mcts_step*: max_states must be positive, the head pointer present, and roots can only be expanded
    while every tree is listed in order (their network rows are those of list position == tree index)."""
    import ctypes
    from librubiks import _hip
    from librubiks.cube import DeviceCubes
    from librubiks.solving import mcts_device as md
    forest = md.MCTSForest(64, 200)
    forest.set_net(net_gpu, torch.float32)
    roots = DeviceCubes.from_numpy(np.array([oc.scramble(4, True)[0] for _ in range(64)]))
    lib, m = forest.lib, ctypes.byref(forest.struct)
    head = torch.zeros((64 * 11, 16), device="cuda")
    assert lib.rc_mcts_plant_expanded(m, None, 64, roots.soa.data_ptr(), roots.stride, 0, 0, None) == -4          # max_states = 0
    assert lib.rc_mcts_plant_expanded(m, None, 64, None, roots.stride, 0, 200, None) == -1
    assert lib.rc_mcts_plant_expanded(m, None, 65, roots.soa.data_ptr(), roots.stride, 0, 200, None) == -4
    assert lib.rc_mcts_step_head(m, None, 16, 0, 0.6, 0, 200, None) == -1
    assert lib.rc_mcts_step_head(m, head.data_ptr(), 12, 0, 0.6, 0, 200, None) == -4                              # 13 values per row
    assert lib.rc_mcts_step_head(m, head.data_ptr(), 16, 0, 0.6, 0, 0, None) == -4
    assert lib.rc_mcts_step(m, None, None, 0.6, 0, 200, None) == -1
    forest.reset(roots, 200)                       # plants and expands: fine while all 64 trees are listed
    forest.set_active(np.arange(32))
    assert lib.rc_mcts_plant_expanded(ctypes.byref(forest.struct), None, 32, roots.soa.data_ptr(), roots.stride, 0, 200, None) == -4
    with pytest.raises(AssertionError):
        forest.plant(None, roots, 0, 200)
    s = forest.listed(torch.arange(4, dtype=torch.int32, device="cuda"))
    s.n_active = 65
    assert lib.rc_mcts_complete_graph(ctypes.byref(s), None) == -4                                                # list longer than the forest
    torch.cuda.synchronize()


def test_a_second_batch_does_not_take_the_forest_from_a_run_that_is_not_done(net_gpu):
    """`start_batch` hands out a run that steps the agent's forest.  Starting another batch of another shape on the same agent before the
    first is finished must not hand the first forest's memory on under it: the first run still ends with the oracle's trees."""
    from librubiks.solving.agents import MCTS
    np.random.seed(12)
    states = np.array([oc.scramble(3 + i % 6, True)[0] for i in range(24)])
    agent = MCTS(net_gpu, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=4)
    first = agent.start_batch(states, None, 500, compact=False)
    for _ in range(3):
        first.round()
    forest = first.forest
    other = agent.search_batch(states[:5], None, 300)             # another shape: the agent builds a new forest ...
    assert agent.forest is not forest and hasattr(forest, "keys")  # ... and the first one still has its arrays
    while not first.done:
        first.round()
    res = first.finish()
    onet = oa.TorchNet(net_gpu, device="cuda")
    for t in range(0, 24, 4):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        ok = ref.search(states[t], 500)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) and list(res.queues[t]) == list(ref.action_queue), f"tree {t}"
    assert len(other.nodes) == 5
