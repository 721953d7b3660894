"""
The search agents cache a BatchNorm-folded, cast copy of the network (InferenceNet) inside their forest / batch.
It must follow the module: training `agent.net` in place or assigning a new module between two searches has to
change what the next search evaluates (Train.train does exactly that before every evaluation rollout).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cube as oc  # noqa: E402


def _root_rows(agent, n):
    """(P, V) of node 1 of the first n trees of the agent's forest."""
    f = agent.forest
    rows = torch.arange(n, device=f.P.device) * (f.C + 1) + 1
    return f.P[rows].cpu().numpy().copy(), f.V[rows].cpu().numpy().copy()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_mcts_engine_follows_the_module(dtype):
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    np.random.seed(3)
    states = np.array([oc.scramble(12, True)[0] for _ in range(16)])
    torch.manual_seed(1)
    net_a = Model.create(ModelConfig()).eval()
    torch.manual_seed(2)
    net_b = Model.create(ModelConfig()).eval()
    agent = MCTS(net_a, c=0.6, search_graph=False, net_dtype=dtype)
    agent.search_batch(states, None, 200, compact=False)
    pa, va = _root_rows(agent, 16)
    agent.search_batch(states, None, 200, compact=False)          # same weights: the engine is reused, same outputs
    assert np.array_equal(_root_rows(agent, 16)[0], pa)
    engine = agent.forest.engine
    agent.search_batch(states, None, 200, compact=False)
    assert agent.forest.engine is engine
    # (1) a different module
    agent.net = net_b
    agent.search_batch(states, None, 200, compact=False)
    pb, vb = _root_rows(agent, 16)
    fresh = MCTS(net_b, c=0.6, search_graph=False, net_dtype=dtype)
    fresh.search_batch(states, None, 200, compact=False)
    pf, vf = _root_rows(fresh, 16)
    assert np.array_equal(pb, pf) and np.array_equal(vb, vf)
    assert not np.allclose(pb, pa, atol=1e-3)
    # (2) the same module trained in place (an optimizer step bumps the parameters' version counters)
    net_b.train()
    opt = torch.optim.SGD(net_b.parameters(), lr=0.5)
    x = torch.randn(64, 480, device="cuda")
    p, v = net_b(x)
    (p.square().mean() + v.square().mean()).backward()
    opt.step()
    net_b.eval()
    agent.search_batch(states, None, 200, compact=False)
    pc, vc = _root_rows(agent, 16)
    fresh = MCTS(net_b, c=0.6, search_graph=False, net_dtype=dtype)
    fresh.search_batch(states, None, 200, compact=False)
    pf, vf = _root_rows(fresh, 16)
    assert np.array_equal(pc, pf) and np.array_equal(vc, vf)
    assert not np.array_equal(pc, pb)


def test_astar_engine_follows_the_module():
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import AStar
    np.random.seed(5)
    states = np.array([oc.scramble(10, True)[0] for _ in range(8)])
    torch.manual_seed(1)
    net_a = Model.create(ModelConfig()).eval()
    torch.manual_seed(2)
    net_b = Model.create(ModelConfig()).eval()
    agent = AStar(net_a, lambda_=0.2, expansions=10)
    ra = agent.search_batch(states, None, 3000, max_iterations=6)
    cost_a = agent.batch.heap_cost[:8].cpu().numpy()
    agent.net = net_b
    rb = agent.search_batch(states, None, 3000, max_iterations=6)
    cost_b = agent.batch.heap_cost[:8].cpu().numpy()
    fresh = AStar(net_b, lambda_=0.2, expansions=10)
    rf = fresh.search_batch(states, None, 3000, max_iterations=6)
    assert np.array_equal(cost_b, fresh.batch.heap_cost[:8].cpu().numpy()) and np.array_equal(rb.nodes, rf.nodes)
    assert not np.array_equal(cost_a, cost_b)
    with torch.no_grad():
        for p in net_b.parameters():
            p.mul_(1.25)
    rc = agent.search_batch(states, None, 3000, max_iterations=6)
    fresh = AStar(net_b, lambda_=0.2, expansions=10)
    rf = fresh.search_batch(states, None, 3000, max_iterations=6)
    assert np.array_equal(agent.batch.heap_cost[:8].cpu().numpy(), fresh.batch.heap_cost[:8].cpu().numpy())
    assert np.array_equal(rc.nodes, rf.nodes)
    assert not np.array_equal(agent.batch.heap_cost[:8].cpu().numpy(), cost_b)
