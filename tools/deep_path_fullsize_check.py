"""
BASELINE configs[1] (1 024 depth-20 scrambles, MCTS c 0.6, graph search, max_states 175 000, trained weights) searched twice: at the production sizes of the path
store (4 096 levels in LDS, 4 096-level blocks) and with RUBIKS_LDS_LEVELS / RUBIKS_PATH_BLOCK / RUBIKS_RING_LEVELS shrunk so that nearly every level of every descent
(up to ~1 300 levels, ~12 k nodes per tree) goes through the deep-path code and its blocks arrive while the search runs.  The network engine and the batch are the same,
so every game must end identically: nodes, iterations, solution.

    python tools/deep_path_fullsize_check.py            # runs both in child processes and compares
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, os, sys, time
sys.path[:0] = [{root!r}, os.path.join({root!r}, "rl-rubiks_amd")]
import numpy as np, torch
from librubiks import cube
from librubiks.model import Model
from librubiks.solving.agents import MCTS
np.random.seed(0)
cubes, _, _ = cube.scramble_batch(1024, 20, True)
agent = MCTS(Model.load(os.path.join({root!r}, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, deterministic=True)
t = time.perf_counter()
res = agent.search_batch(cubes, None, 175000)
dt = time.perf_counter() - t
h = hashlib.sha256()
for q in res.queues:
    h.update(bytes(q)); h.update(b"|")
f = agent._last_forest
print(json.dumps({{"seconds": round(dt, 2), "solved": float(res.solved.mean()), "nodes": int(res.nodes.sum()), "nodes_sha": hashlib.sha256(res.nodes.tobytes()).hexdigest()[:16],
                  "iterations_sha": hashlib.sha256(res.iterations.tobytes()).hexdigest()[:16], "queues_sha": h.hexdigest()[:16], "lds_levels": f.lds_levels,
                  "path_block": f.path_block, "deepest_mapped_levels": int(f.path_rows_host.max()), "path_overflow_trees": res.path_overflow_trees}}))
"""


def run(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    a = run({})
    print("production sizes:", json.dumps(a), flush=True)
    b = run({"RUBIKS_LDS_LEVELS": "8", "RUBIKS_PATH_BLOCK": "64", "RUBIKS_RING_LEVELS": "64"})
    print("8 levels in LDS, 64-level blocks and lines:", json.dumps(b), flush=True)
    same = all(a[k] == b[k] for k in ("solved", "nodes", "nodes_sha", "iterations_sha", "queues_sha"))
    print("identical games:", same)
    sys.exit(0 if same else 1)
