"""
A seeded sweep over the batched MCTS's knobs: every configuration draws its own batch size, node cap, exploration constant,
scramble depths, slot count (continuous batching or not), narrowing on / off, round length and level budget, and every game of
it must end exactly as the oracle's single-tree search ends -- solved flag, node count, iterations, action queue.  The fixed
parity tests pin chosen corners; this walks the space between them (RUBIKS_SWEEP_CONFIGS=200 widens it for a soak run).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

N_CONFIGS = int(os.environ.get("RUBIKS_SWEEP_CONFIGS", "24"))


def _draw(rng):
    n = int(rng.choice([1, 2, 5, 17, 33, 48, 70]))
    cfg = {
        "n": n, "cap": int(rng.choice([13, 14, 25, 26, 60, 150, 400, 900, 1500])), "c": float(rng.choice([0.05, 0.6, 1.0, 4.13, 25.0])),
        "graph": bool(rng.integers(2)), "depth_lo": int(rng.integers(0, 12)), "depth_span": int(rng.integers(1, 12)),
        "slots": None if rng.integers(3) == 0 or n < 4 else int(rng.integers(2, n)), "compact": bool(rng.integers(2)),
        "sync_every": int(rng.choice([1, 2, 4, 8, 16, 50])), "budget": rng.choice(["auto", 1, 2, 7, 64]),
    }
    cfg["budget"] = cfg["budget"] if cfg["budget"] == "auto" else int(cfg["budget"])
    return cfg


@pytest.mark.parametrize("seed", range(N_CONFIGS))
def test_every_game_of_a_random_configuration_ends_as_the_oracle_ends(seed, standin_net):
    from librubiks.solving.agents import MCTS
    rng = np.random.default_rng(1000 + seed)
    cfg = _draw(rng)
    np.random.seed(2000 + seed)
    states = np.array([oc.scramble(cfg["depth_lo"] + i % cfg["depth_span"], True)[0] for i in range(cfg["n"])])
    net = standin_net.cuda()
    agent = MCTS(net, c=cfg["c"], search_graph=cfg["graph"], net_dtype=torch.float32, sync_every=cfg["sync_every"], level_budget=cfg["budget"])
    res = agent.search_batch(states, None, cfg["cap"], compact=cfg["compact"], slots=cfg["slots"])
    onet = oa.TorchNet(net, device="cuda")
    for t, s in enumerate(states):
        ref = oa.MCTS(onet, c=cfg["c"], search_graph=cfg["graph"])
        ok = ref.search(s, cfg["cap"])
        what = f"{cfg} game {t}"
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref), what
        assert list(res.queues[t]) == list(ref.action_queue) and res.lengths[t] == (len(ref.action_queue) if ok else -1), what
        assert res.iterations[t] == getattr(ref, "iterations", 0), what      # (a solved root: the oracle returns before it counts)
        if ok:
            x = s
            for a in res.queues[t]:
                x = oc.rotate(x, *oc.ACTION_SPACE[a])
            assert oc.is_solved(x), what
