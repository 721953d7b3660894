# Kernel trace of bench.py's timed window (fp32-accurate engine, steady-state pool) -> per-step timeline of the last steps before the window ends.
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/profw -- python3 /root/repo/bench.py --steps 20 --warmup 5 --legs f32s --extra-legs "" --window-only --no-cpu-baseline --no-env-roofline --phase-reps 1 > /root/repo/gpurun_out/r3g_window_bench.json 2> /root/repo/gpurun_out/r3g_window.log
cd /root/repo
f=$(find /tmp/profw -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/r3g_window_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
sel = [i for i, e in enumerate(k) if "k_mcts_select" in e[2]]
print("tree-kernel launches", len(sel))
# periods between consecutive tree kernels, whole trace, bucketed
import collections
per = [(k[b][1] - k[a][1]) / 1e3 for a, b in zip(sel, sel[1:])]
print("periods (us): n", len(per), "median", sorted(per)[len(per) // 2], "p10", sorted(per)[len(per) // 10], "p90", sorted(per)[9 * len(per) // 10])
# dump the 60 steps with full listing around 3/4 of the trace (inside the timed window's neighbourhood)
for idx in range(len(sel) - 40, len(sel) - 10):
    a, b = sel[idx], sel[idx + 1]
    parts = []
    for e in k[a + 1:b + 1]:
        nm = e[2]
        for key in ("k_mcts_select", "k_split_gemm", "k_first_layer_split", "k_split_reduce", "k_head_split", "k_mcts_plant", "k_mcts_harvest", "copyBuffer", "Cijk"):
            if key in nm:
                nm = key
                break
        parts.append(f"{nm[:28]}:{(e[1] - e[0]) / 1e3:.0f}@q{e[3]}")
    busy = sum(e[1] - e[0] for e in k[a + 1:b + 1] if e[3] == k[b][3]) / 1e3
    print(f"step {idx}: period {(k[b][1] - k[a][1]) / 1e3:.0f} us, busy on its queue {busy:.0f}:", " ".join(parts))
PY
