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


def _bench_line(args, env_extra, launcher=(), detail=None):
    """Runs bench.py; returns (the one stdout line, the full result it wrote to its detail file)."""
    import json
    import subprocess
    import sys
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra)
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "bench_detail.json")
        cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args, "--detail", detail]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        assert len(lines[0]) < 8000 and out.stdout.rstrip().endswith(lines[0])     # short, and the last thing on stdout
        line = json.loads(lines[0])
        assert line["detail"] == "bench_detail.json"
        with open(detail) as f:
            full = json.load(f)
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup", "scaling"):
        assert line[k] == full[k], k
    assert line["config"]["results"] == full["config"]["results"]
    return line, full


def test_bench_preflight_alone():
    """`bench.py --gpus 2 --preflight`: device binding, free HBM and every collective the legs use, on two gloo ranks sharing the
    GPU; prints one line and runs no leg.  A rank that cannot meet a requirement ends with a rank-tagged message and code 3."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["RUBIKS_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["preflight"] == "ok" and line["n_gpus"] == 2 and line["collectives"].endswith("ok")
    # a requirement that cannot be met: ONE rank-tagged line per failing rank on stderr, exit code 3, no result line
    env["RUBIKS_PREFLIGHT_NEED_GB"] = "100000"
    short = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=300)
    assert short.returncode == 3 and "[bench preflight] rank" in short.stderr and "GB of HBM free" in short.stderr, short.stderr[-1500:]
    assert not [ln for ln in short.stdout.splitlines() if ln.startswith("{")]
    del env["RUBIKS_PREFLIGHT_NEED_GB"]
    # RCCL with two ranks on ONE GPU: pick_backend refuses before anything hangs
    env["RUBIKS_DIST_BACKEND"] = "nccl"
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_two_ranks_aggregate_their_shares():
    """
    bench.py for N = 2, both ways it can be started: under torch.distributed.run (one process per rank, RANK / WORLD_SIZE in
    the environment) and as plain `python bench.py --gpus 2`, which starts its two ranks itself.  Both ranks share this one GPU
    (gloo backend): rank r searches its slice of the scrambles and the line aggregates them.
    Each rank's share is also run alone (`--as-rank r/2`: same slice, same forest sizes, hence the same bf16 GEMM
    shapes and bit-identical trees): the distributed results must be exactly the sum of the two.
    """
    import socket
    common = ["--steps", "6", "--warmup", "2", "--trees", "48", "--pool-factor", "2", "--legs", "bf16", "--solve-max-states",
              "1500", "--phase-reps", "0", "--no-cpu-baseline", "--no-env-roofline", "--prep-cap", "40", "--astar-problems", "24",
              "--config5-trees", "32", "--config5-max-states", "800", "--extra-legs", "astar,config5"]   # (N > 1 defaults to config5 alone)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line, two = _bench_line(["--gpus", "2", *common], {"RUBIKS_DIST_BACKEND": "gloo"},
                            launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                      "--master-port", str(port)))
    line_plain, two_plain = _bench_line(["--gpus", "2", *common], {"RUBIKS_DIST_BACKEND": "gloo"})
    shares = [_bench_line(["--gpus", "1", "--as-rank", f"{r}/2", *common], {})[1] for r in range(2)]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["dtype"] == "bf16"
    assert line_plain["n_gpus"] == 2 and line_plain["config"].keys() == line["config"].keys()
    a = two["legs"]["bf16"]["run_to_completion"]
    parts = [x["legs"]["bf16"]["run_to_completion"] for x in shares]
    assert a["games"] == 96 and [p["games"] for p in parts] == [48, 48]
    assert a["nodes"] == parts[0]["nodes"] + parts[1]["nodes"]
    assert abs(a["solve_rate"] - (parts[0]["solve_rate"] + parts[1]["solve_rate"]) / 2) < 1e-12
    pool = two["legs"]["bf16"]["pool_run"]
    assert pool["games"] == 2 * 96 and pool["nodes"] == sum(x["legs"]["bf16"]["pool_run"]["nodes"] for x in shares)
    assert two["legs"]["bf16"]["steps_timed"] == 6 and two["value"] > 0
    assert two["config"]["parallelism"] == "scramble-sharded x2" and two["scaling_measured"] is True and shares[0]["scaling_measured"] is False
    # every rank's own rate is on the line, the preflight ran before the first leg, and the efficiency field is there (a number
    # only when a one-GPU run of the same workload left its value next to bench.py)
    assert len(line["rank_values"]) == 2 and all(v > 0 for v in line["rank_values"]) and line["value"] <= sum(line["rank_values"]) * 1.001
    assert two["preflight"]["backend"] == "gloo" and two["preflight"]["collectives"].endswith("ok") and two["preflight"]["free_hbm_gb"] > 90
    assert "efficiency" in line and (line["efficiency"] is None or 0 < line["efficiency"] < 2) and shares[0]["efficiency"] is None
    assert line["value_per_gpu"] == round(line["value"] / 2, 1) and line["value_spread"]["min"] <= line["value_spread"]["median"] <= line["value_spread"]["max"]
    assert max(len(v) for v in (line["config"]["workload"], line["config"]["timed_region"], line["roofline"].get("kernel", ""))) <= 120
    # the self-launched run searched the same games to the same trees
    ap = two_plain["legs"]["bf16"]
    assert ap["run_to_completion"]["nodes"] == a["nodes"] and ap["run_to_completion"]["solve_rate"] == a["solve_rate"]
    assert ap["pool_run"]["nodes"] == pool["nodes"]
    # the extra legs (A*: BASELINE configs[2]; one GPU's share of configs[4]) aggregate over the ranks as well
    for name in ("f32s", "bf16"):
        a2, parts2 = two["astar"][name]["solve_run"], [x["astar"][name]["solve_run"] for x in shares]
        assert a2["games"] == 48 and a2["nodes"] == parts2[0]["nodes"] + parts2[1]["nodes"]
        assert two_plain["astar"][name]["solve_run"]["nodes"] == a2["nodes"]
        c2, cparts = two["config5_share"][name]["run_to_completion"], [x["config5_share"][name]["run_to_completion"] for x in shares]
        assert c2["games"] == 64 and c2["nodes"] == cparts[0]["nodes"] + cparts[1]["nodes"]
    r = line["config"]["results"]
    assert r["value_run_to_completion"] == a["nodes_per_sec"] and "astar_f32s_states_per_sec" in r and "config5_share_bf16_value" in r
