"""
Search-agent traces recorded from the IMPORTED REFERENCE agents (librubiks/solving/agents.py MCTS,
AStar) driven by tests/standin_net.StandInNet.  See make_golden.py for how to run.
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from standin_net import StandInNet  # noqa: E402

MCTS_CASES = [
    # (name, scramble seed, depth, c, search_graph, max_states) -- seeds picked by scanning for
    # cases the reference solves, incl. ones where the BFS shortening changes the action queue
    ("d2_s3_graph", 3, 2, 0.6, True, 2000),
    ("d3_s8_naive", 8, 3, 0.6, False, 2000),
    ("d4_s17_c20_graph", 17, 4, 20.0, True, 2500),
    ("d4_s17_c20_naive", 17, 4, 20.0, False, 2500),
    ("d4_s24_c4_graph", 24, 4, 4.13, True, 2500),
    ("d5_s1_c20_graph", 1, 5, 20.0, True, 2500),
    ("d5_s7_c4_graph", 7, 5, 4.13, True, 2500),
    ("d7_s25_c20_graph", 25, 7, 20.0, True, 2500),
    ("d2_s1_unsolved", 1, 2, 0.6, True, 2000),
    ("d20_graph", 6, 20, 0.6, True, 5000),
    ("d20_naive_c4", 7, 20, 4.13, False, 2500),
    ("d24_graph", 8, 24, 0.6, True, 1500),
]
ASTAR_CASES = [
    # (name, scramble seed, depth, lambda, expansions, max_states)
    ("d2_l0_n10", 11, 2, 0.0, 10, 3000),
    ("d3_l05_n2", 12, 3, 0.5, 2, 3000),
    ("d3_l1_n1", 13, 3, 1.0, 1, 3000),
    ("d5_l02_n100", 14, 5, 0.2, 100, 6000),
    ("d20_l02_n100", 15, 20, 0.2, 100, 8000),
    ("d20_l016_n20", 16, 20, 0.16, 20, 4000),
    ("d24_l02_n7", 17, 24, 0.2, 7, 2500),
]


def make_agents():
    from librubiks import cube
    from librubiks.solving.agents import MCTS, AStar
    torch.set_num_threads(1)
    net = StandInNet(seed=0)
    fx = {f"net_{k}": v for k, v in net.numpy_weights().items()}
    # reference outputs of the stand-in net on a few states, to pin the net itself
    np.random.seed(99)
    probe = np.array([cube.scramble(12)[0] for _ in range(32)])
    with torch.no_grad():
        p, v = net(cube.as_oh(probe))
    fx["net_probe_states"], fx["net_probe_p"], fx["net_probe_v"] = probe, p.numpy(), v.numpy()

    for name, seed, depth, c, graph, max_states in MCTS_CASES:
        np.random.seed(seed)
        state, _, _ = cube.scramble(depth, True)
        agent = MCTS(net, c=c, search_graph=graph)
        solved = agent.search(state, None, max_states)
        n = len(agent)
        assert sorted(agent.indices.values()) == list(range(1, n + 1))
        pre = f"mcts_{name}_"
        fx[pre + "state"] = state
        fx[pre + "params"] = np.array([depth, c, int(graph), max_states, int(solved), n], dtype=np.float64)
        fx[pre + "queue"] = np.array(list(agent.action_queue), dtype=np.int16)
        fx[pre + "states"] = agent.states[:n + 1].copy()
        fx[pre + "neighbors"] = agent.neighbors[:n + 1].astype(np.int32)
        fx[pre + "leaves"] = agent.leaves[:n + 1].copy()
        fx[pre + "N"] = agent.N[:n + 1].astype(np.int32)
        fx[pre + "W"] = agent.W[:n + 1].copy()
        fx[pre + "P"] = agent.P[:n + 1].astype(np.float32)
        fx[pre + "V"] = agent.V[:n + 1].astype(np.float32)
        assert np.array_equal(fx[pre + "P"][1:].astype(np.float64), agent.P[1:n + 1])   # values are f32-exact
        fx[pre + "L"] = agent.L[:n + 1].astype(np.int32)
        for k in ("states", "P", "V"):      # row 0 of these is np.empty memory in the reference (agents.py:440-444): zeroed,
            fx[pre + k][0] = 0              # so that the fixture is reproducible byte for byte
        print(f"MCTS {name}: solved={solved} n={n} queue={list(agent.action_queue)[:12]}")

    for name, seed, depth, lam, nexp, max_states in ASTAR_CASES:
        np.random.seed(seed)
        state, _, _ = cube.scramble(depth, True)
        agent = AStar(net, lambda_=lam, expansions=nexp)
        solved = agent.search(state, None, max_states)
        n = len(agent)
        pre = f"astar_{name}_"
        fx[pre + "state"] = state
        fx[pre + "params"] = np.array([depth, lam, nexp, max_states, int(solved), n], dtype=np.float64)
        fx[pre + "queue"] = np.array(list(agent.action_queue), dtype=np.int16)
        fx[pre + "states"] = agent.states[:n + 1].copy()
        fx[pre + "G"] = agent.G[:n + 1].copy()
        fx[pre + "parents"] = agent.parents[:n + 1].astype(np.int32)
        fx[pre + "parent_actions"] = agent.parent_actions[:n + 1].astype(np.int8)
        fx[pre + "states"][0], fx[pre + "G"][0] = 0, 0        # np.empty rows of the reference (agents.py:390-393) ...
        fx[pre + "parents"][:2] = 0                            # ... incl. the root's parent, which it never sets
        oq = sorted((float(c), int(i)) for c, i in agent.open_queue)
        fx[pre + "open_cost"] = np.array([c for c, _ in oq])
        fx[pre + "open_idx"] = np.array([i for _, i in oq], dtype=np.int32)
        print(f"A* {name}: solved={solved} n={n} open={len(oq)} queue={list(agent.action_queue)[:12]}")

    np.savez_compressed(os.path.join(OUT, "agents_golden.npz"), **fx)
    print("agents_golden.npz", os.path.getsize(os.path.join(OUT, "agents_golden.npz")) // 1024, "KiB")


def make_adi():
    """ADI training targets of the reference's Train.ADI_traindata (librubiks/train.py:257-339)."""
    from librubiks.train import Train
    from librubiks.solving.agents import PolicySearch
    net = StandInNet(seed=0)
    fx = {}
    for method in ("paper", "lapanfix", "schultzfix", "reward0"):
        for games, depth, alpha in ((6, 10, 0.0), (5, 7, 0.4)):
            tr = Train(rollouts=1, batch_size=10, rollout_games=games, rollout_depth=depth, optim_fn=torch.optim.Adam,
                       alpha_update=0, lr=1e-3, gamma=1, update_interval=0, agent=PolicySearch(None), evaluator=None,
                       evaluation_interval=0, with_analysis=False, tau=1, reward_method=method)
            np.random.seed(31)
            oh, ptar, vtar, w = tr.ADI_traindata(net, alpha)
            pre = f"adi_{method}_{games}x{depth}_"
            fx[pre + "ohcols"] = np.nonzero(oh.cpu().numpy())[1].reshape(games * depth, 20).astype(np.int16)
            fx[pre + "policy"] = ptar.numpy()
            fx[pre + "value"] = vtar.numpy()
            fx[pre + "weights"] = w.numpy()
            fx[pre + "alpha"] = np.array([alpha])
            print(pre, ptar[:8].tolist(), vtar[:4].tolist())
    np.savez_compressed(os.path.join(OUT, "adi_golden.npz"), **fx)


def make_simple_agents():
    """ValueSearch / PolicySearch (solved games only: they are wall-time bounded) and EGVM (max_states bounded)."""
    from librubiks import cube
    from librubiks.solving.agents import ValueSearch, PolicySearch, EGVM
    torch.set_num_threads(1)
    net = StandInNet(seed=0)
    fx = {}
    for name, cls in (("value", ValueSearch), ("policy", PolicySearch)):
        states, queues = [], []
        for depth in (1, 2, 3, 4, 5):
            for seed in range(40):
                np.random.seed(1000 * depth + seed)
                s, _, _ = cube.scramble(depth, True)
                agent = cls(net)
                if agent.search(s, time_limit=0.15) and len(agent.action_queue) <= 64:
                    states.append(s)
                    queues.append(list(agent.action_queue) + [-1] * (64 - len(agent.action_queue)))
        fx[f"{name}_states"], fx[f"{name}_queues"] = np.array(states), np.array(queues, dtype=np.int16)
        print(name, "solved cases:", len(states), "max len", max((sum(a >= 0 for a in q) for q in queues), default=0))
    cases = []
    for i, (eps, workers, depth, sdepth, max_states) in enumerate(
            [(0.1, 4, 12, 3, 600), (0.3, 10, 5, 4, 1000), (0.0, 6, 8, 2, 480), (0.5, 16, 3, 5, 960), (0.2, 8, 10, 20, 800),
             (0.9, 32, 4, 1, 1280), (1.0, 64, 3, 2, 1920)]):
        for seed in (1, 2, 3):
            np.random.seed(100 * i + seed)
            s, _, _ = cube.scramble(sdepth, True)
            agent = EGVM(net, eps, workers, depth)
            ok = agent.search(s, None, max_states)
            pre = f"egvm_{i}_{seed}_"
            fx[pre + "state"] = s
            fx[pre + "params"] = np.array([eps, workers, depth, max_states, int(ok), len(agent), 100 * i + seed])
            fx[pre + "queue"] = np.array(list(agent.action_queue), dtype=np.int16)
            cases.append((ok, len(agent), len(agent.action_queue)))
    print("EGVM cases (solved, len, queue):", cases)
    np.savez_compressed(os.path.join(OUT, "simple_agents_golden.npz"), **fx)
