"""
A seeded sweep over batched A*'s knobs (batch size, lambda, expansions per iteration, node cap, scramble depths): every problem
of every configuration must end as the oracle's single-problem search ends -- solved flag, node count, action queue.
(RUBIKS_SWEEP_CONFIGS widens it for a soak run; see tests/test_mcts_sweep_gpu.py.)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

N_CONFIGS = int(os.environ.get("RUBIKS_SWEEP_CONFIGS", "16"))


@pytest.mark.parametrize("seed", range(N_CONFIGS))
def test_every_problem_of_a_random_configuration_ends_as_the_oracle_ends(seed, standin_net):
    from librubiks.solving.agents import AStar
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.choice([1, 3, 16, 40, 65]))
    lam = float(rng.choice([0.0, 0.05, 0.2, 0.6, 1.0, 3.0]))
    nexp = int(rng.choice([1, 2, 5, 16, 64, 100]))
    cap = int(rng.choice([12, 13, 40, 200, 700, 2000]))
    lo, span = int(rng.integers(0, 9)), int(rng.integers(1, 9))
    np.random.seed(4000 + seed)
    states = np.array([oc.scramble(lo + i % span, True)[0] for i in range(n)])
    net = standin_net.cuda()
    res = AStar(net, lambda_=lam, expansions=nexp, net_dtype=torch.float32).search_batch(states, None, cap)
    onet = oa.TorchNet(net, device="cuda")
    for b, s in enumerate(states):
        ref = oa.AStar(onet, lambda_=lam, expansions=nexp)
        ok = ref.search(s, cap)
        what = f"n={n} lambda={lam} expansions={nexp} cap={cap} depths {lo}+{span} problem {b}"
        assert bool(res.solved[b]) == ok and res.nodes[b] == len(ref), what
        assert list(res.queues[b]) == list(ref.action_queue) and res.lengths[b] == (len(ref.action_queue) if ok else -1), what
        if ok:
            x = s
            for a in res.queues[b]:
                x = oc.rotate(x, *oc.ACTION_SPACE[a])
            assert oc.is_solved(x), what
