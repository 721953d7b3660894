"""
The reference's own operating points at full size (no oracle at these sizes: structural checks and
replay of the returned solutions): AStar with the authors' lambda = 0.16, N = 700
(configs/main_eval.ini:8-9) and MCTS with the CLI default max_states = 175 000 (runeval.py:32-73).
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

from oracle import cube as oc  # noqa: E402  (checker only)

WEIGHTS = os.path.join(ROOT, "weights", "fc_small_r1")


def _net():
    from librubiks.model import Model, ModelConfig
    if os.path.isdir(WEIGHTS):
        return Model.load(WEIGHTS).eval(), True
    torch.manual_seed(0)
    return Model.create(ModelConfig()).eval(), False


def _replay_ok(state, queue):
    for a in queue:
        state = oc.rotate(state, *oc.ACTION_SPACE[a])
    return oc.is_solved(state)


def test_astar_authors_settings():
    from librubiks import cube
    from librubiks.solving.agents import AStar
    net, trained = _net()
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(64, 20, True)
    states = cubes.numpy()
    agent = AStar(net, lambda_=0.16, expansions=700)
    res = agent.search_batch(cubes, None, 175_000)
    assert (res.nodes <= 175_000).all()
    for b in np.flatnonzero(res.solved):
        assert _replay_ok(states[b], res.queues[b]) and res.lengths[b] == len(res.queues[b])
    for b in np.flatnonzero(~res.solved):
        assert res.nodes[b] + 700 * 12 > 175_000       # stopped by the budget rule (agents.py:236)
    if trained:
        assert res.solved.mean() > 0.8
    # per-problem structure of one problem: G of the root's children, parents consistent with G
    h = agent.batch.problem_arrays(0)
    n = h["n"]
    assert h["G"][1] == 0 and (h["G"][2:14] == 1).all() and (h["parents"][2:14] == 1).all()
    idx = np.arange(2, n + 1)
    assert (h["G"][idx] >= 1).all() and (h["parents"][idx] >= 1).all() and (h["parents"][idx] <= n).all()


def test_mcts_default_budget():
    from librubiks import cube
    from librubiks.solving.agents import MCTS
    net, trained = _net()
    np.random.seed(1)
    cubes, _, _ = cube.scramble_batch(8, 24, True)
    states = cubes.numpy()
    agent = MCTS(net, c=0.6, search_graph=True)
    res = agent.search_batch(cubes, None, 175_000)
    assert (res.nodes <= 175_000).all()
    for t in range(8):
        if res.solved[t]:
            assert _replay_ok(states[t], res.queues[t])
        else:
            assert res.nodes[t] + 12 > 175_000 or res.status[t] == 3
    assert agent.forest.C >= 175_000
    if trained:
        assert res.solved.sum() >= 6
