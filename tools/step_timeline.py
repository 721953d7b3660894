"""
Per-step timeline from a rocprofv3 --kernel-trace CSV: a step ends with its tree kernel (k_mcts_select; rounds 1-2: started at
k_mcts_expand); for the chosen steps print the period, the busy time per kernel (by short name) and the idle gaps on the step's own queue.

    python tools/step_timeline.py kernel_trace.csv [first_step] [n_steps]      (first_step < 0: counted from the end of the trace)
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
k.sort()


def short(name):
    for key in ("k_mcts_expand", "k_mcts_select", "k_mcts_backup", "k_split_gemm", "k_first_layer_split", "k_split_act", "k_split_reduce", "k_head_split",
                "k_first_layer_mfma", "k_head", "k_act_bf16", "k_mcts_plant",
                "k_mcts_complete_graph", "k_mcts_shorten", "Cijk"):
        if key in name:
            if key == "Cijk":
                return "Cijk_" + name.split("_MT")[1].split("_")[0] if "_MT" in name else "Cijk"
            return key
    return name[:40]


starts = [i + 1 for i, e in enumerate(k) if "k_mcts_select" in e[2]]   # a step = the launches behind the previous tree kernel up to and including this one
starts = [i for i in starts if i < len(k)]
print("steps in trace:", len(starts))
if first < 0:
    first += len(starts)
sel = starts[first:first + count + 1]
if len(sel) < 2:
    sys.exit("not enough steps")
main_q = k[sel[0]][3]
period = (k[sel[-1]][0] - k[sel[0]][0]) / (len(sel) - 1) / 1e3
busy, other, gaps = defaultdict(float), defaultdict(float), defaultdict(float)
for a, b in zip(sel[:-1], sel[1:]):
    prev_end, prev_name = None, None
    for e in k[a:b]:
        if e[3] == main_q:
            busy[short(e[2])] += (e[1] - e[0]) / 1e3
            if prev_end is not None and e[0] > prev_end:
                gaps[prev_name + " -> " + short(e[2])] += (e[0] - prev_end) / 1e3
            prev_end, prev_name = max(e[1], prev_end or 0), short(e[2])
        else:
            other[short(e[2])] += (e[1] - e[0]) / 1e3
    nxt = k[b]
    if prev_end is not None and nxt[0] > prev_end:
        gaps[prev_name + " -> next step"] += (nxt[0] - prev_end) / 1e3
n = len(sel) - 1
print(f"steps {first}..{first + n}: period {period:.1f} us")
print("busy on the step queue (us/step):")
for name, v in sorted(busy.items(), key=lambda x: -x[1]):
    print(f"  {v / n:8.1f}  {name}")
print(f"  {sum(busy.values()) / n:8.1f}  total")
print("gaps on the step queue (us/step):")
for name, v in sorted(gaps.items(), key=lambda x: -x[1])[:12]:
    print(f"  {v / n:8.1f}  {name}")
print(f"  {sum(gaps.values()) / n:8.1f}  total")
print("other queues (us/step):")
for name, v in sorted(other.items(), key=lambda x: -x[1])[:8]:
    print(f"  {v / n:8.1f}  {name}")
