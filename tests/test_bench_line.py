"""
bench.py's contract with the driver, checked without a GPU:
  * the ONE stdout line is built from the full result by `compact_line`, is valid JSON, stays under 8 000 bytes (the driver's
    record keeps the last 8 KB of output: round 3's 26 KB line came back unparsed) and carries what the contract names;
  * `python bench.py --gpus N` without a launcher starts its N ranks itself (`launch_ranks`), relays rank 0's line and
    fails fast, with the rank's exit code, when one of them dies.
The canned full result is a real one: round 3's line (profiles/r3h_bench.json, 26 KB).
"""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r3h_bench.json")


def _full():
    with open(CANNED) as f:
        return json.load(f)


def test_line_is_small_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 20_000          # the thing that did not parse
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT <= 8000 and "\n" not in text
    back = json.loads(text)
    assert back == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert key in back, key
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert isinstance(back["config"]["workload"], str) and back["config"]["workload"]
    assert "model" not in back["config"]
    assert back["config"]["results"] == full["config"]["results"]          # every flat scalar survives
    roof = back["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["env_multi_rotate_2p24"]["bound"] == "hbm" and "astar_dominant_kernel" in roof
    cpu = back["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference")
    for bulky in ("legs", "astar", "config5_share", "roofline_env"):
        assert bulky not in back
    assert "boundary_calls" not in cpu
    assert "dropped_to_detail" not in back
    # the driver's record keeps 120 characters of a string: nothing in the line is longer (long forms live in the detail file)
    def strings(x):
        if isinstance(x, str):
            yield x
        elif isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
    assert max(len(t) for t in strings(back)) <= bench.STR_LIMIT == 120


def test_short_forms_are_what_the_line_carries():
    full = _full()
    full["config"]["workload_short"] = "1024 depth-20 MCTS trees/GPU, c=0.6, graph search, max_states 175000, pool 8x1024, fc_small weights/fc_small_r1"
    full["config"]["timed_region_short"] = "K lock-step steps of the stationary pool between barrier+synchronize; prep and warm-up untimed"
    full["roofline"]["kernel_short"] = "rc_split_gemm_f16 352x256 tiles, hidden layer 1: [11264x12288]x[12288x2048] f16 MFMA, f32 acc, +bias+ELU+re-split"
    full.update(rank_values=[1.0, 2.0], efficiency=0.97)
    line = bench.compact_line(full)
    assert line["config"]["workload"] == full["config"]["workload_short"] and len(line["config"]["workload"]) <= 120
    assert line["config"]["timed_region"] == full["config"]["timed_region_short"]
    assert line["roofline"]["kernel"] == full["roofline"]["kernel_short"] and not line["roofline"]["kernel"].endswith("...")
    assert line["rank_values"] == [1.0, 2.0] and line["efficiency"] == 0.97


def test_line_stays_under_the_limit_whatever_the_result_holds():
    full = _full()
    full["config"]["workload"] = "w" * 5000
    full["config"]["timed_region"] = "t" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["dtype"] = "d" * 500
    for i in range(400):
        full["config"]["results"][f"extra_scalar_{i}"] = 1234567.8 + i
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT
    assert line["dropped_to_detail"]                      # and it says what it left to the detail file
    assert line["value"] == full["value"] and "frac" in line["roofline"] and "value" in line["cpu_baseline"]
    assert "value_run_to_completion" in line["config"]["results"]     # the first scalars are the last to go


def test_emit_writes_the_detail_file_and_prints_one_line(tmp_path, capsys):
    full = _full()
    path = str(tmp_path / "bench_detail.json")
    bench.emit(full, path)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0]) < 8000
    assert json.loads(lines[0])["detail"] == "bench_detail.json"
    with open(path) as f:
        assert json.load(f) == full


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    mode = sys.argv[sys.argv.index("--mode") + 1]
    if mode == "die" and rank == 1:
        sys.exit(7)
    if mode == "die":
        time.sleep(120)            # a survivor that would wait for ever
    print("rank", rank, "chatter")
    if rank == 0:
        print(json.dumps({"n_gpus": world, "argv": sys.argv[1:]}))
""")


def test_gpus_n_without_a_launcher_starts_its_ranks_and_relays_rank0(tmp_path, capsys):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    env_before = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE")}
    rc = bench.launch_ranks(3, ["--gpus", "3", "--mode", "ok"], script=str(script))
    out = capsys.readouterr().out
    assert rc == 0
    lines = out.splitlines()
    assert len(lines) == 1                      # rank 0's result line alone, none of the chatter
    assert json.loads(lines[0]) == {"n_gpus": 3, "argv": ["--gpus", "3", "--mode", "ok"]}
    assert env_before == {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE")}


def test_a_dead_rank_ends_the_launch_quickly_with_its_code(tmp_path, capsys):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    t0 = time.perf_counter()
    rc = bench.launch_ranks(2, ["--mode", "die"], script=str(script))
    assert rc == 7 and time.perf_counter() - t0 < 15
    assert capsys.readouterr().out == ""


def test_main_takes_the_launcher_branch_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE must reach launch_ranks; under a launcher (RANK set) it must not."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    probe = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        bench.launch_ranks = lambda n, argv, **kw: (print("LAUNCH", n, argv), 0)[1]
        sys.argv = ["bench.py", "--gpus", "2", "--steps", "3"]
        bench.main()
    """)
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "LAUNCH 2 ['--gpus', '2', '--steps', '3']", out.stderr[-1500:]


def test_scale_efficiency_from_the_one_gpu_record(tmp_path):
    """`efficiency` of an N-GPU line = value / (N x the value the same checkout's one-GPU run of the same workload left behind);
    null for one GPU, without a record, or when the record is of another workload."""
    ref = str(tmp_path / "bench_scale_ref.json")
    key = {"trees": 1024, "depth": 20, "max_states": 175000, "leg": "f32s", "pool_factor": 8, "steps": 20, "warmup": 5}
    assert bench.scale_efficiency(8, 9.0e7, key, ref)[0] is None                       # nothing to divide by yet
    eff, note = bench.scale_efficiency(1, 1.25e7, key, ref, gpu="MI355X")
    assert eff is None and "reference" in note and json.load(open(ref))["value"] == 1.25e7
    assert bench.scale_efficiency(8, 9.5e7, key, ref)[0] == 0.95 and bench.scale_efficiency(2, 2.5e7, key, ref)[0] == 1.0
    assert bench.scale_efficiency(8, 9.5e7, dict(key, trees=512), ref)[0] is None      # another workload
    assert bench.scale_efficiency(1, 1.0e7, key, ref, write=False)[0] is None and json.load(open(ref))["value"] == 1.25e7   # --as-rank runs leave no record
    open(ref, "w").write("{not json")
    assert bench.scale_efficiency(4, 1.0, key, ref)[0] is None
    # a one-GPU value handed in (--scale-ref-value / RUBIKS_SCALE_REF) needs no record next to the script at all
    assert bench.scale_efficiency(8, 9.6e7, key, str(tmp_path / "absent.json"), ref_value=1.25e7)[0] == 0.96


def test_launched_ranks_get_the_ipc_mode_rccl_needs_on_this_pool(monkeypatch, tmp_path):
    """`bench.py --gpus N` without a launcher starts its ranks itself.  RCCL shares device buffers between the ranks of a node through
    IPC handles and the hosts of this pool only support the dmabuf form (HSA_ENABLE_IPC_MODE_LEGACY=0; with the legacy mode
    hipIpcGetMemHandle fails and so does the first collective): a rank started from an environment that lost the variable gets it
    back, one that sets it keeps its own value."""
    import subprocess
    import sys
    seen = []

    class FakeProc:
        returncode = 0

        def __init__(self, argv, env=None, stdout=None, text=None):
            seen.append(env)
            import io
            self.stdout = io.StringIO('{"ok": 1}\n') if stdout is not None and stdout != subprocess.DEVNULL else None

        def poll(self):
            return 0

        def wait(self, timeout=None):
            return 0

        def terminate(self):
            pass

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
    assert [e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in seen] == ["0", "0"] and [e["RANK"] for e in seen] == ["0", "1"]
    seen.clear()
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    assert bench.launch_ranks(2, ["--gpus", "2"]) == 0 and [e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in seen] == ["1", "1"]
