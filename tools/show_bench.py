"""Prints the scalars of a bench.py JSON line that matter.   python tools/show_bench.py gpurun_out/x.json"""
import json
import sys


def main(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "dtype")}))
    print(json.dumps(d["config"]["results"], indent=1))
    for n, leg in d["legs"].items():
        print(n, leg["value"], leg["ms_per_step"], "flushes", leg.get("result_flushes_in_window"), "pool s", (leg.get("pool_run") or {}).get("seconds"),
              "rtc", {k: v for k, v in (leg.get("run_to_completion") or {}).items() if k in ("seconds", "nodes_per_sec", "solve_rate", "lock_step_iterations_rank0")})
        print("  phases", leg.get("phases_ms"))
    print({k: v for k, v in d["roofline"].items() if k not in ("note", "kernel", "traffic_source")})
    for n, leg in (d.get("astar") or {}).items():
        if isinstance(leg, dict):
            print("astar", n, leg["value"], leg["ms_per_iteration"], leg.get("phases_ms"), leg.get("solve_run"), (leg.get("roofline") or {}).get("frac"))
    for n, leg in (d.get("config5_share") or {}).items():
        if isinstance(leg, dict):
            print("config5", n, leg["value"], leg["ms_per_step"], leg.get("run_to_completion"))
    print("cpu", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "kind")})


if __name__ == "__main__":
    main(sys.argv[1])
