"""Print the headline numbers of a bench.py JSON line: python tools/show_bench.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
for name, l in d.get("legs", {}).items():
    print(name, "window", l["value"], "pool", l["pool_run"]["nodes_per_sec"], "rtc", l["run_to_completion"]["nodes_per_sec"],
          l["run_to_completion"]["seconds"], "solved", l["run_to_completion"]["solve_rate"])
    print("   phases", l["phases_ms"])
