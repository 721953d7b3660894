"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/env_bench.py into per-launch HBM bytes.

gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import json
import sys


def median_per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in d.items()}


def main(fetch_csv, write_csv, out_json, log2n=24):
    n = 1 << int(log2n)
    f, w = median_per_kernel(fetch_csv, "FETCH_SIZE"), median_per_kernel(write_csv, "WRITE_SIZE")
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
        out[short] = {"kernel": fk[0], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes": rd + wr,
                      "algorithmic_bytes": alg, "states": n,
                      "corrections": "KiB->B; FETCH_SIZE x2 (gfx950 wide streaming reads)"}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
