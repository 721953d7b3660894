"""
Batched Evaluator with the reference's interface and result shapes (librubiks/solving/evaluation.py:15-125,
result files as librubiks/jobs.py:283-300,313-322) -- plots excluded.

    res, states, times = Evaluator(n_games, scrambling_depths, max_time, max_states).eval(agent)

The reference plays the games one after another; here all `n_games` scrambles of a depth are searched
at once through `agent.search_batch` (one tree / one A* problem per scramble on the GPU).  The
scrambles are the reference's own (same np.random stream, `cube.scramble(depth, True)` per game), so
`res` / `states` are comparable game by game with a reference run on the same seed for agents that
do not draw random numbers themselves (MCTS, AStar).  `max_time` bounds every game's wall time as in
the reference, but the games of a depth run concurrently.

`times[d, g]` is the reference's quantity (evaluation.py:45-52): game g's own wall interval, from the moment its search starts
(its tree is planted, its problem enters the batch) to the moment the host sees it finished (`BatchResult.game_seconds`), and
"states per sec" is the mean over games of states / time as in evaluation.py:120-124.  Because the games of a batch share the
GPU these intervals overlap -- their sum is not the wall time of the evaluation; the throughput figure (all states of a depth /
the batch's wall seconds) is kept beside it as `states_per_sec_batch` in `log_this_depth`'s summary and in `batch_seconds`.
"""
import inspect
import json
import os

import numpy as np

from librubiks import cube
from librubiks.solving.sharding import sharded_search_batch
from librubiks.utils import NullLogger, TickTock, bernoulli_error


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class Evaluator:
    def __init__(self, n_games, scrambling_depths, max_time=None, max_states=None, logger=NullLogger(), slots: int = None):
        """
        slots: for agents whose `search_batch` takes `slots` (MCTS): the games of ALL depths form one pool that
        is searched with at most `slots` concurrent trees, finished trees handing their place to waiting games
        (continuous batching); `max_time`, if given, then bounds the whole pool at max_time x len(depths).
        """
        self.n_games, self.max_time, self.max_states, self.slots = n_games, max_time, max_states, slots
        self.tt = TickTock()
        self.log = logger
        # range(0) means "deep": every game draws its own depth uniformly in [100, 999] (evaluation.py:30,73-74)
        self.scrambling_depths = np.array(scrambling_depths) if scrambling_depths != range(0) else np.array([0])

    def _isdeep(self):
        return self.scrambling_depths.size == 1 and self.scrambling_depths[0] == 0

    def approximate_time(self):
        return self.max_time * len(self.scrambling_depths)   # games of a depth run concurrently

    def eval(self, agent):
        """(res, states, times), each (len(scrambling_depths), n_games); res = solution length or -1."""
        res, states, times = [], [], []
        self.batch_seconds = []     # wall seconds of each depth's batch (the pooled form: one entry for everything)
        if self.slots and hasattr(agent, "search_batch") and "slots" in inspect.signature(agent.search_batch).parameters:
            return self._eval_pooled(agent)
        for d in self.scrambling_depths:
            depth = (lambda: np.random.randint(100, 1000)) if self._isdeep() else int(d)
            if hasattr(agent, "search_batch"):
                cubes, _, _ = cube.scramble_batch(self.n_games, depth, True)   # every rank draws the same scrambles
                self.tt.profile(f"Evaluation of {agent}. Depth {'100 - 999' if self._isdeep() else d}")
                if _world_size() > 1:   # one process per GPU: each searches its slice of the games, results are all-gathered
                    got = sharded_search_batch(agent, cubes.numpy(), self.max_time, self.max_states, device=cubes.soa.device)
                    lengths, nodes, each = got["lengths"], got["nodes"], got["seconds"]
                else:
                    out = agent.search_batch(cubes, self.max_time, self.max_states)
                    lengths, nodes, each = out.lengths, out.nodes, getattr(out, "game_seconds", None)
                dt = self.tt.end_profile()
                res.append(lengths)
                states.append(nodes)
                times.append(np.asarray(each, dtype=float) if each is not None else np.full(self.n_games, dt / self.n_games))
                self.batch_seconds.append(dt)
            else:   # agents without a batched search: the reference's game-by-game loop
                r, s, t = [], [], []
                for _ in range(self.n_games):
                    dd = depth() if callable(depth) else depth
                    state, _, _ = cube.scramble(dd, True)
                    self.tt.profile(f"Evaluation of {agent}. Depth {d}")
                    found = agent.search(state, self.max_time, self.max_states)
                    t.append(self.tt.end_profile())
                    r.append(len(agent.action_queue) if found else -1)
                    s.append(len(agent))
                res.append(r), states.append(s), times.append(t)
        res, states, times = np.array(res, dtype=np.int64), np.array(states, dtype=np.int64), np.array(times, dtype=float)
        for i, d in enumerate(self.scrambling_depths):
            self.log_this_depth(res[i], states[i], times[i], d, self.batch_seconds[i] if i < len(self.batch_seconds) else None)
        return res, states, times

    def _eval_pooled(self, agent):
        """All depths' scrambles (drawn in the reference's order) as one pool with continuous batching."""
        from librubiks.cube.device import DeviceCubes
        D, G = len(self.scrambling_depths), self.n_games
        pool = DeviceCubes.empty(D * G)
        for i, d in enumerate(self.scrambling_depths):
            depth = (lambda: np.random.randint(100, 1000)) if self._isdeep() else int(d)
            cubes, _, _ = cube.scramble_batch(G, depth, True)
            pool.soa[:, i * G:(i + 1) * G] = cubes.soa[:, :G]
        self.tt.profile(f"Evaluation of {agent}. {D} depths pooled")
        out = agent.search_batch(pool, self.max_time * D if self.max_time else None, self.max_states, slots=self.slots)
        dt = self.tt.end_profile()
        res, states = out.lengths.reshape(D, G).astype(np.int64), out.nodes.reshape(D, G).astype(np.int64)
        each = getattr(out, "game_seconds", None)
        times = np.asarray(each, dtype=float).reshape(D, G) if each is not None else np.full((D, G), dt / (D * G))
        self.batch_seconds = [dt]
        for i, d in enumerate(self.scrambling_depths):
            self.log_this_depth(res[i], states[i], times[i], d)
        return res, states, times

    def log_this_depth(self, res, states, times, depth, batch_seconds=None) -> dict:
        """Summary statistics of one depth (evaluation.py:96-125); logged and returned.  batch_seconds: wall time of the depth's
        batch -> `states_per_sec_batch`, the throughput of the GPU (the reference's figure, the mean of per-game states / time,
        is `states_per_sec`)."""
        won = res[res != -1]
        share = len(won) / len(res)
        ok = times != 0
        sps = states[ok] / times[ok]
        summary = {
            "depth": int(depth), "share_completed": share, "ci95": float(bernoulli_error(share, len(res), 0.05)),
            "mean_turns": float(won.mean()) if won.size else None,
            "median_turns": float(np.median(won)) if won.size else None,
            "states_per_game": float(states.mean()), "states_per_sec": float(sps.mean()) if sps.size else 0.0,
            "time_per_game": float(times.mean()),
            "states_per_sec_batch": float(states.sum() / batch_seconds) if batch_seconds else None,
        }
        self.log(f"Scrambling depth {depth if depth else 'deep'}\n"
                 f"\tShare completed: {share * 100:.2f} % {bernoulli_error(share, len(res), 0.05, stringify=True)} (approx. 95 % CI)\n"
                 + (f"\tTurns to win: {won.mean():.2f} +/- {won.std():.1f} (std.), Median: {np.median(won):.0f}\n" if won.size else "")
                 + f"\tStates seen: Pr. game: {states.mean():.2f} +/- {states.std():.0f} (std.), Pr. sec.: {summary['states_per_sec']:.2f}\n"
                 f"\tTime:  {times.mean():.4f} +/- {times.std():.4f} (std.)")
        return summary

    def save(self, location: str, name: str, res, states, times) -> list:
        """The reference's result files: <name>_results.npy, _states_seen.npy, _playtimes.npy + eval_settings.json."""
        sub = os.path.join(location, "evaluation_results")
        os.makedirs(sub, exist_ok=True)
        paths = [os.path.join(sub, f"{name}_{kind}.npy") for kind in ("results", "states_seen", "playtimes")]
        for p, arr in zip(paths, (res, states, times)):
            np.save(p, arr)
        settings = {name: {"n_games": self.n_games, "max_time": self.max_time, "max_states": self.max_states,
                           "scrambling_depths": self.scrambling_depths.tolist()}}
        with open(os.path.join(location, "eval_settings.json"), "w", encoding="utf-8") as f:
            json.dump(settings, f, indent=4)
        return paths
