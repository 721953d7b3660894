"""Same-box A/B of a switch of the split engine (a class attribute of SplitF32Net, default small_batch_cut on / off): config #2 as one
batch to completion and as a pool of games at the reference's cap, fp32-accurate engine, variants alternating in one process.
    python tools/engine_ab.py [attribute value_a value_b]        e.g.  gemm_input_rows 8192 0"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import Model, SplitF32Net  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

CAP = 175000
np.random.seed(0)
batch, _, _ = cube.scramble_batch(1024, 20, True)
pool, _, _ = cube.scramble_batch(4096, 20, True)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
out = {}
ATTR = sys.argv[1] if len(sys.argv) > 1 else "small_batch_cut"
VALS = [float(v) if "." in v else int(v) for v in sys.argv[2:4]] if len(sys.argv) > 3 else [True, False]
OWNER = SplitF32Net
if "." in ATTR:      # e.g. mcts_device.RUNG_RATIO: a module constant of librubiks.solving
    import importlib
    mod, ATTR = ATTR.rsplit(".", 1)
    OWNER = importlib.import_module("librubiks.solving." + mod)
for name, on in ((f"{ATTR}={VALS[0]}", VALS[0]), (f"{ATTR}={VALS[1]}", VALS[1]), (f"{ATTR}={VALS[0]} again", VALS[0]), (f"{ATTR}={VALS[1]} again", VALS[1])):
    if ATTR == "sync_every":      # a constructor argument of the agent
        agent = MCTS(model, c=0.6, search_graph=True, sync_every=int(on))
    else:
        setattr(OWNER, ATTR, type(getattr(OWNER, ATTR))(on))
        agent = MCTS(model, c=0.6, search_graph=True)
    agent.prepare(1024, CAP)
    agent.search_batch(batch, None, CAP)
    runs = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = agent.search_batch(batch, None, CAP)
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rp = agent.search_batch(pool, None, CAP, slots=1024)
    torch.cuda.synchronize()
    tp = time.perf_counter() - t0
    out[name] = {"batch_seconds": [round(x, 4) for x in runs], "batch_nodes_per_sec": round(float(r.nodes.sum()) / min(runs)),
                 "solved": float(r.solved.mean()), "nodes": int(r.nodes.sum()),
                 "pool_seconds": round(tp, 4) if rp is not None else None,
                 "pool_nodes_per_sec": round(float(rp.nodes.sum()) / tp) if rp is not None else None}
    print(name, json.dumps(out[name]), flush=True)
    del agent
    torch.cuda.empty_cache()
