"""
Summaries of rocprofv3 output, as committed under profiles/ (run where the profile was taken):

  kernels  <results.db> <out.csv> [rows]       --kernel-trace (rocpd sqlite): one row per kernel name and grid -- launches, total ms,
                                               share, avg / min / max us (the `*_kernel_stats.csv` files)
  pmc      <counter_collection.csv> [substr]   --pmc: mean counter value per kernel and counter
  traffic  <fetch.csv> <write.csv> <out.json> [log2n]   the two --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/env_bench.py -> HBM bytes
                                               per launch of the environment kernels, gfx950 corrections applied
                                               (MI355X_MICROARCH.md, HBM: counters in KiB; FETCH_SIZE reports half the bytes of a
                                               wide coalesced streaming read -> doubled; WRITE_SIZE exact for 16-B-per-lane stores)
  timeline <kernel_trace.csv> [first_step] [n] [dump] --kernel-trace --output-format csv of a search: per lock-step iteration (a step ends
                                               with its tree kernel) the period, busy time per kernel and idle gaps on the step's
                                               queue (first_step < 0: counted from the end of the trace)
"""
import collections
import csv
import json
import sqlite3
import sys


def kernels(db_path, out_csv, rows=100):
    cur = sqlite3.connect(db_path).cursor()
    got = cur.execute("select name, grid_x, workgroup_x, count(*), sum(end - start) / 1e6, avg(end - start) / 1e3, "
                      "min(end - start) / 1e3, max(end - start) / 1e3 from kernels group by name, grid_x, workgroup_x "
                      "order by 5 desc").fetchall()
    total = sum(r[4] for r in got)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_x", "workgroup_x", "launches", "total_ms", "share_pct", "avg_us", "min_us", "max_us"])
        for name, gx, wx, n, tot, avg, mn, mx in got[:int(rows)]:
            w.writerow([name, gx, wx, n, f"{tot:.3f}", f"{100 * tot / total:.2f}", f"{avg:.2f}", f"{mn:.2f}", f"{mx:.2f}"])
    print(f"{len(got)} rows ({min(len(got), int(rows))} written), {total:.1f} ms of kernel time -> {out_csv}")


def pmc(path, sub=""):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:70], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"])
            cnt[key] += 1
    for (k, c), v in sorted(acc.items()):
        print(f"{k:70s} {c:32s} mean {v / cnt[(k, c)]:.6g}  (n={cnt[(k, c)]})")


def _median_per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}


def traffic(fetch_csv, write_csv, out_json, log2n=24):
    n = 1 << int(log2n)
    f, w = _median_per_kernel(fetch_csv, "FETCH_SIZE"), _median_per_kernel(write_csv, "WRITE_SIZE")
    # (kernel-name prefix of the large-batch variant, short name, algorithmic bytes per launch in tools/env_bench.py)
    wanted = [("rubiks::k_multi_rotate<4", "multi_rotate", 41 * n), ("rubiks::k_expand12<256", "expand12", 260 * n // 4),
              ("rubiks::k_is_solved", "is_solved", None), ("rubiks::k_as_oh<256, false", "as_oh(f32)", 1940 * n // 16),
              ("rubiks::k_as_oh<256, true", "as_oh(bf16)", 980 * n // 16)]
    out = {}
    for prefix, short, alg in wanted:
        fk = [k for k in f if k.startswith(prefix)]
        wk = [k for k in w if k.startswith(prefix)]
        if not fk or not wk:
            continue
        rd, wr = 2 * f[fk[0]] * 1024, w[wk[0]] * 1024
        out[short] = {"kernel": fk[0], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "algorithmic_bytes": alg,
                      "states": n, "corrections": "KiB->B; FETCH_SIZE x2 (gfx950 wide streaming reads)"}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


_SHORT = ("k_mcts_expand", "k_mcts_select", "k_mcts_backup", "k_split_gemm", "k_first_layer_split", "k_split_act", "k_split_reduce", "k_head_split",
          "k_first_layer_mfma", "k_head", "k_act_bf16", "k_mcts_plant", "k_mcts_copy_trees", "k_mcts_complete_graph", "k_mcts_shorten", "Cijk")


def _short(name):
    for key in _SHORT:
        if key in name:
            if key == "Cijk":
                return "Cijk_" + name.split("_MT")[1].split("_")[0] if "_MT" in name else "Cijk"
            return key
    return name[:40]


def timeline(path, first=2000, count=200, dump=0):
    first, count, dump = int(first), int(count), int(dump)
    rows = list(csv.DictReader(open(path)))
    k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows)
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
    busy, other, gaps = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(float)
    worst = collections.defaultdict(lambda: [0.0, 0])      # per edge: the largest single gap and how many exceeded 20 us
    for a, b in zip(sel[:-1], sel[1:]):
        prev_end, prev_name = None, None
        for e in k[a:b]:
            if e[3] == main_q:
                busy[_short(e[2])] += (e[1] - e[0]) / 1e3
                if prev_end is not None and e[0] > prev_end:
                    g = (e[0] - prev_end) / 1e3
                    gaps[prev_name + " -> " + _short(e[2])] += g
                    w = worst[prev_name + " -> " + _short(e[2])]
                    w[0], w[1] = max(w[0], g), w[1] + (g > 20.0)
                prev_end, prev_name = max(e[1], prev_end or 0), _short(e[2])
            else:
                other[_short(e[2])] += (e[1] - e[0]) / 1e3
        nxt = k[b]
        if prev_end is not None and nxt[0] > prev_end:
            gaps[prev_name + " -> next step"] += (nxt[0] - prev_end) / 1e3
    n = len(sel) - 1
    if dump:      # the launches of the first selected step, one per line: offset from the step's start, duration, queue, grid
        t0 = k[sel[0]][0]
        for e in k[sel[0]:sel[1] + 1]:
            print(f"    +{(e[0] - t0) / 1e3:8.1f} us  {(e[1] - e[0]) / 1e3:8.1f} us  queue {e[3]:>3}  {_short(e[2])}")
    print(f"steps {first}..{first + n}: period {period:.1f} us")
    for title, d, top in (("busy on the step queue", busy, 99), ("gaps on the step queue", gaps, 12), ("other queues", other, 8)):
        print(f"{title} (us/step):")
        for name, v in sorted(d.items(), key=lambda x: -x[1])[:top]:
            extra = f"   (largest {worst[name][0]:.0f} us, {worst[name][1]} of {n} steps above 20 us)" if d is gaps and name in worst else ""
            print(f"  {v / n:8.1f}  {name}{extra}")
        if d is not other:
            print(f"  {sum(d.values()) / n:8.1f}  total")


if __name__ == "__main__":
    {"kernels": kernels, "pmc": pmc, "traffic": traffic, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])
