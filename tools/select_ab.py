"""
A/B of k_mcts_select builds in the steady-state pool (tools/window_probe.py): average duration of the 1 024-tree launches from a
rocprofv3 --kernel-trace csv, HBM fetch / write per launch from --pmc csvs, and the share of levels the one-line
re-validation settled (a -DRC_SELECT_FASTSTATS build).

    python tools/select_ab.py trace <kernel_trace.csv>      python tools/select_ab.py pmc <counter_collection.csv>
    python tools/select_ab.py faststats                     (runs the pool itself)
"""
import csv
import os
import sys

import numpy as np


def trace(path):
    rows = [r for r in csv.DictReader(open(path)) if "k_mcts_select<2>" in r["Kernel_Name"]]
    gx = "Grid_Size_X" if "Grid_Size_X" in rows[0] else "Grid_Size"
    big = [r for r in rows if int(r[gx]) == 1024 * 256]
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in big]) / 1e3
    tail = d[len(d) // 2:]     # the second half of the run: the stationary pool
    print(f"k_mcts_select<2> at 1 024 trees: {len(d)} launches, mean {d.mean():.1f} us, second half mean {tail.mean():.1f} us "
          f"p50 {np.percentile(tail, 50):.1f} p90 {np.percentile(tail, 90):.1f} max {tail.max():.1f}")


def pmc(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        if "k_mcts_select<2>" in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) == 1024 * 256:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = np.array(v)
        tail = v[len(v) // 2:]
        mult = 2 if k == "FETCH_SIZE" else 1
        print(f"{k}: {len(v)} launches, second half mean {tail.mean():.0f} KiB = {tail.mean() * 1024 * mult / 1e6:.1f} MB per launch"
              f"{' (x2: gfx950 tallies 16-byte-per-lane reads at half)' if mult == 2 else ''}; as counted {tail.mean() * 1024 / 1e6:.1f} MB")


def faststats():
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
    from librubiks import cube
    from librubiks.model import Model
    from librubiks.solving.agents import MCTS
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(8192, 20, True)
    agent = MCTS(Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval(), c=0.6, search_graph=True, net_dtype=torch.bfloat16)
    run = agent.start_batch(cubes, None, 50000, slots=1024)
    while run.next_game < 2048 + 128:
        run.round()
    for rep in range(4):
        for _ in range(5):
            run.round()
        torch.cuda.synchronize()
        st = run.forest.select_stats.cpu().numpy().astype(np.int64)
        live = (run.forest.status == 0).cpu().numpy() & (st[:, 1] > 2)
        ok, no = st[live, 5], st[live, 6]
        deep = live & (st[:, 1] > 600)
        print(f"levels re-decided per step: {int((ok + no).sum())}, from line 0 alone {ok.sum() / max((ok + no).sum(), 1):.1%}; "
              f"trees deeper than 600 levels ({int(deep.sum())}): {st[deep, 5].sum() / max((st[deep, 5] + st[deep, 6]).sum(), 1):.1%}", flush=True)


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}.get(sys.argv[1], lambda *_: faststats())(*sys.argv[2:])
