"""ValueSearch / PolicySearch / RandomSearch / EGVM on the device vs reference traces and the oracle."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402


def test_value_search_reference_traces(standin_net):
    from librubiks.solving.agents import ValueSearch
    g = np.load(f"{GOLDEN}/simple_agents_golden.npz")
    agent = ValueSearch(standin_net.cuda(), net_dtype=torch.float32)
    res = agent.search_batch(g["value_states"], None, 64)        # all 107 games in one batch
    assert res.solved.all()
    for q, ref in zip(res.queues, g["value_queues"]):
        assert list(q) == list(ref[ref >= 0])
    # per-game wall intervals (what the Evaluator reports as `times`): a game with fewer moves is seen finished no later than one with more
    lens = np.array([len(q) for q in res.queues])
    assert res.game_seconds.shape == lens.shape and (res.game_seconds > 0).all() and res.game_seconds.max() <= res.seconds
    order = np.argsort(lens, kind="stable")
    assert (np.diff(res.game_seconds[order]) >= 0).all()
    assert agent.search(g["value_states"][3], None, 64) and list(agent.action_queue) == list(res.queues[3])
    assert str(agent) == "Greedy value"


def test_step_agents_vs_oracle(standin_net):
    from librubiks.solving.agents import PolicySearch, ValueSearch
    net = standin_net.cuda()
    onet = oa.TorchNet(net, device="cuda")
    np.random.seed(4)
    states = np.array([oc.scramble(1 + i % 6, True)[0] for i in range(48)])
    states[5] = oc.get_solved()
    for prod, ref_cls in ((PolicySearch(net, net_dtype=torch.float32), oa.PolicySearch),
                          (ValueSearch(net, net_dtype=torch.float32), oa.ValueSearch)):
        res = prod.search_batch(states, None, 30)
        for g, s in enumerate(states):
            ref = ref_cls(onet)
            ok = ref.search(s, 30)
            assert bool(res.solved[g]) == ok and list(res.queues[g]) == list(ref.action_queue), f"game {g}"
            assert res.lengths[g] == (len(ref.action_queue) if ok else -1)
            x = s
            for a in res.queues[g]:
                x = oc.rotate(x, *oc.ACTION_SPACE[a])
            assert oc.is_solved(x) == ok
        assert res.solved[5] and res.lengths[5] == 0


def test_random_search_single_game_stream():
    """With one game the draws are the reference's: one np.random.randint(12) per step (agents.py:84)."""
    from librubiks.solving.agents import RandomSearch
    np.random.seed(8)
    s, _, _ = oc.scramble(3, True)
    np.random.seed(99)
    expect = [int(np.random.randint(12)) for _ in range(25)]
    np.random.seed(99)
    agent = RandomSearch()
    ok = agent.search(s, None, 25)
    q = list(agent.action_queue)
    assert q == expect[:len(q)] and (ok or len(q) == 25)


def test_egvm_reference_traces(standin_net):
    from librubiks.solving.agents import EGVM
    g = np.load(f"{GOLDEN}/simple_agents_golden.npz")
    net = standin_net.cuda()
    cases = sorted(k[:-len("params")] for k in g.files if k.startswith("egvm_") and k.endswith("params"))
    sdepth = {0: 3, 1: 4, 2: 2, 3: 5, 4: 20, 5: 1, 6: 2}
    for pre in cases:
        eps, workers, depth, max_states, solved, n, seed = g[pre + "params"]
        np.random.seed(int(seed))
        state, _, _ = oc.scramble(sdepth[int(pre.split("_")[1])], True)   # consumes the same draws as the recording
        agent = EGVM(net, float(eps), int(workers), int(depth), net_dtype=torch.float32)
        assert agent.search(state, None, int(max_states)) == bool(solved), pre
        assert len(agent) == int(n), pre
        assert list(agent.action_queue) == list(g[pre + "queue"]), pre
    assert str(agent).startswith("EGVM (e=")


def test_reference_generic_agent_test_restated():
    """tests/test_agents.py:18-36 of the reference: every agent, a depth-4 scramble, 50 ms, queue replays to `solved`."""
    from librubiks import cube
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import BFS, EGVM, PolicySearch, RandomSearch, ValueSearch
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig()).eval()
    agents = [RandomSearch(), BFS(), PolicySearch(net, sample_policy=False), PolicySearch(net, sample_policy=True),
              ValueSearch(net), EGVM(net, 0.1, 4, 12)]
    for agent in agents:
        state, _, _ = cube.scramble(4)
        found = agent.search(state, .05)
        assert all(0 <= a < cube.action_dim for a in agent.action_queue)
        for a in agent.action_queue:
            state = cube.rotate(state, *cube.action_space[a])
        assert found == cube.is_solved(state), str(agent)
    assert str(agents[1]) == "Breadth-first search"
