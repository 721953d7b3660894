"""Mean counter values per kernel from a rocprofv3 --pmc CSV: python tools/pmc_summary.py counter_collection.csv [name-substring]"""
import csv
import sys
from collections import defaultdict

acc, cnt = defaultdict(float), defaultdict(int)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if sub in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:60], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        cnt[key] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:32s} mean {v / cnt[(k, c)]:.4g}  (n={cnt[(k, c)]})")
