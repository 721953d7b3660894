"""
Pins oracle/agents.py (restated BFS / MCTS / A*) against traces recorded from the imported
reference agents (tests/golden/agents_golden.npz, bfs_golden.npz).  CPU only, exact comparisons:
same torch CPU arithmetic for the stand-in net on both sides, float64 bookkeeping as the reference.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_cases
from oracle import agents as oa
from oracle import cube as oc

_G = np.load(f"{GOLDEN}/agents_golden.npz")


def test_standin_net_pinned(agents_golden, standin_net):
    with torch.no_grad():
        p, v = standin_net(torch.from_numpy(oc.as_oh(agents_golden["net_probe_states"])))
    assert np.array_equal(p.numpy(), agents_golden["net_probe_p"])
    assert np.array_equal(v.numpy(), agents_golden["net_probe_v"])
    from standin_net import StandInNet
    regenerated = StandInNet(seed=0).numpy_weights()
    for k, w in regenerated.items():
        assert np.array_equal(w, agents_golden["net_" + k])


def test_bfs_config1(bfs_golden):
    """BASELINE config #1: 10 depth-5 scrambles, lengths and states-seen of the reference run."""
    agent = oa.BFS()
    for s, length, seen, queue in zip(bfs_golden["states"], bfs_golden["lengths"], bfs_golden["seen"],
                                      bfs_golden["queues"]):
        assert agent.search(s, 10_000_000)
        assert len(agent.action_queue) == length and len(agent) == seen
        assert list(agent.action_queue) == list(queue[:length])
        for a in agent.action_queue:
            s = oc.rotate(s, *oc.ACTION_SPACE[a])
        assert oc.is_solved(s)


@pytest.mark.parametrize("case", golden_cases(_G, "mcts_"))
def test_mcts_trace(case, agents_golden, standin_net):
    g = lambda k: agents_golden[f"mcts_{case}_{k}"]   # noqa: E731
    depth, c, graph, max_states, solved, n = g("params")
    agent = oa.MCTS(oa.TorchNet(standin_net), c=float(c), search_graph=bool(graph))
    assert agent.search(g("state"), int(max_states)) == bool(solved)
    n = int(n)
    assert len(agent) == n
    assert list(agent.action_queue) == list(g("queue"))
    assert np.array_equal(agent.states[1:n + 1], g("states")[1:])
    assert np.array_equal(agent.neighbors[:n + 1], g("neighbors"))
    assert np.array_equal(agent.leaves[1:n + 1], g("leaves")[1:])
    assert np.array_equal(agent.N[:n + 1], g("N"))
    assert np.array_equal(agent.L[:n + 1], g("L"))
    assert np.array_equal(agent.W[1:n + 1], g("W")[1:])
    assert np.array_equal(agent.P[1:n + 1].astype(np.float32), g("P")[1:])
    assert np.array_equal(agent.V[1:n + 1].astype(np.float32), g("V")[1:])
    if solved:
        s = g("state")
        for a in agent.action_queue:
            s = oc.rotate(s, *oc.ACTION_SPACE[a])
        assert oc.is_solved(s)


@pytest.mark.parametrize("case", golden_cases(_G, "astar_"))
def test_astar_trace(case, agents_golden, standin_net):
    g = lambda k: agents_golden[f"astar_{case}_{k}"]   # noqa: E731
    depth, lam, nexp, max_states, solved, n = g("params")
    agent = oa.AStar(oa.TorchNet(standin_net), lambda_=float(lam), expansions=int(nexp))
    assert agent.search(g("state"), int(max_states)) == bool(solved)
    n = int(n)
    assert len(agent) == n
    assert list(agent.action_queue) == list(g("queue"))
    assert np.array_equal(agent.states[1:n + 1], g("states")[1:])
    assert np.array_equal(agent.G[1:n + 1], g("G")[1:])
    assert np.array_equal(agent.parents[2:n + 1], g("parents")[2:])
    assert np.array_equal(agent.parent_actions[2:n + 1], g("parent_actions")[2:])
    oq = sorted((float(c), int(i)) for c, i in agent.open_queue)
    assert np.array_equal(np.array([i for _, i in oq]), g("open_idx"))
    assert np.array_equal(np.array([c for c, _ in oq]), g("open_cost"))


@pytest.mark.parametrize("method", ["paper", "lapanfix", "schultzfix", "reward0"])
def test_adi_targets(method, standin_net):
    """oracle/train.py against Train.ADI_traindata of the reference (tests/golden/adi_golden.npz)."""
    from oracle import train as ot
    g = np.load(f"{GOLDEN}/adi_golden.npz")
    net = oa.TorchNet(standin_net)
    for games, depth in ((6, 10), (5, 7)):
        pre = f"adi_{method}_{games}x{depth}_"
        np.random.seed(31)
        states, pol, val, w = ot.adi_traindata(net.value, games, depth, method, float(g[pre + "alpha"][0]))
        assert np.array_equal(oc.oh_indices(states), g[pre + "ohcols"])
        assert np.array_equal(pol, g[pre + "policy"])
        assert np.array_equal(val, g[pre + "value"])
        assert np.array_equal(w, g[pre + "weights"])


def test_value_search_and_egvm_traces(standin_net):
    """oracle ValueSearch / EGVM against traces of the reference agents (simple_agents_golden.npz)."""
    g = np.load(f"{GOLDEN}/simple_agents_golden.npz")
    net = oa.TorchNet(standin_net)
    assert len(g["value_states"]) > 50
    for s, q in zip(g["value_states"], g["value_queues"]):
        agent = oa.ValueSearch(net)
        assert agent.search(s, 64)
        assert list(agent.action_queue) == list(q[q >= 0])
    cases = sorted(k[:-len("params")] for k in g.files if k.startswith("egvm_") and k.endswith("params"))
    n_solved = 0
    for pre in cases:
        eps, workers, depth, max_states, solved, n, seed = g[pre + "params"]
        np.random.seed(int(seed))
        state, _, _ = oc.scramble({0: 3, 1: 4, 2: 2, 3: 5, 4: 20, 5: 1, 6: 2}[int(pre.split("_")[1])], True)
        assert np.array_equal(state, g[pre + "state"])
        agent = oa.EGVM(net, float(eps), int(workers), int(depth))
        assert agent.search(state, int(max_states)) == bool(solved)
        assert len(agent) == int(n)
        assert list(agent.action_queue) == list(g[pre + "queue"])
        n_solved += int(solved)
    assert n_solved >= 2


def test_bfs_max_states_cut_reference_runs():
    """40 BFS searches of the reference, 32 of them ended by its `len(self) < max_states` test (bfs_cut_golden.npz)."""
    g = np.load(f"{GOLDEN}/bfs_cut_golden.npz")
    agent = oa.BFS()
    for s, cap, solved, seen, queue in zip(g["states"], g["caps"], g["solved"], g["seen"], g["queues"]):
        assert agent.search(s, int(cap)) == bool(solved)
        assert len(agent) == seen
        assert list(agent.action_queue) == [a for a in queue if a >= 0]
    assert (g["solved"] == 0).sum() >= 30
