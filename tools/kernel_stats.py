"""rocprofv3 --kernel-trace results (rocpd sqlite .db) -> per-kernel summary CSV: name, grid, launches, total ms, avg / min / max us.

    python tools/kernel_stats.py gpurun_out/prof/x_results.db profiles/r2_bench_kernel_stats.csv
Kernels of one name launched with different grids (the library GEMM of several shapes) get one row per grid.
"""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, count(*), sum(end - start) / 1e6, avg(end - start) / 1e3, "
                       "min(end - start) / 1e3, max(end - start) / 1e3 from kernels group by name, grid_x, workgroup_x "
                       "order by 5 desc").fetchall()
    total = sum(r[4] for r in rows)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_x", "workgroup_x", "launches", "total_ms", "share_pct", "avg_us", "min_us", "max_us"])
        for name, gx, wx, n, tot, avg, mn, mx in rows:
            w.writerow([name, gx, wx, n, f"{tot:.3f}", f"{100 * tot / total:.2f}", f"{avg:.2f}", f"{mn:.2f}", f"{mx:.2f}"])
    print(f"{len(rows)} rows, {total:.1f} ms of kernel time -> {out_csv}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
