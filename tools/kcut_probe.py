"""
The K-cut hidden layers of a small forest (rc_split_layer_f16 with out_partials + rc_split_reduce_f16), one row count at a time:
every tile that takes the shape (1 = 352 x 256, 3 = 352 x 128, 7 = 352 x 64: fewer, deeper chunks) with every chunk count that keeps the launch within one round of the
chip, timed as GEMM + reduce; the plan `SplitF32Net._k_split` picks is marked.  (Round 5 also ran it on a build with tile 5 = 176 x 128
tiles and FOUR LDS stages, counted vmcnt waits: profiles/r5_kcut_deep_pipeline.txt.)

    python tools/kcut_probe.py [rows ...]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip  # noqa: E402
from librubiks.model import SplitF32Net, _layer_call  # noqa: E402

rows_list = [int(a) for a in sys.argv[1:]] or [352, 704, 1056, 1408, 2112, 2816, 4224, 5632]
lib = _hip.lib()
g = torch.Generator().manual_seed(0)


def ms(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, N in ((4096, 2048), (2048, 1024)):
    W3 = (torch.randn(N, 3 * K, generator=g) / 60).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    steps = 3 * K // 64
    for rows in rows_list:
        a = (torch.randn(rows, 2 * K, generator=g) * 0.5).half().cuda()
        picked = SplitF32Net._k_split(rows, N, K)
        out = torch.empty((rows, 2 * N), dtype=torch.float16, device="cuda")
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        results, keep = [], {}
        for tile, bm, bn in ((1, 352, 256), (3, 352, 128), (7, 352, 64)) + (((5, 176, 128),) if os.environ.get("KCUT_PROBE_TILE5") else ()):
            if N % bn:
                continue
            base = -(-rows // bm) * (N // bn)
            for chunks in (c for c in range(2, 33) if steps % c == 0 and steps // c >= 2 and base * c <= 256 and base * c >= 96):
                part = torch.empty((chunks, rows, N), dtype=torch.float32, device="cuda")
                n_corr = lib.rc_split_layer_corr_chunks(K, chunks)

                def run():
                    _layer_call("rc_split_layer_f16", a=a, w=W3, n_rows=rows, n_out=N, k=K, out_partials=part, k_splits=chunks, tile=tile)
                    _hip.check(lib.rc_split_reduce_f16(part.data_ptr(), rows * N, chunks, n_corr, rows, N, bias.data_ptr(), None, 2, 1.0, None, None,
                                                       out.data_ptr(), None, flag.data_ptr(), _hip.stream_ptr()), "rc_split_reduce_f16")
                t_all = ms(run)
                t_gemm = ms(lambda: _layer_call("rc_split_layer_f16", a=a, w=W3, n_rows=rows, n_out=N, k=K, out_partials=part, k_splits=chunks, tile=tile))
                results.append((t_all, t_gemm, tile, chunks, base * chunks))
                if tile in (3, 5):
                    keep[(tile, chunks)] = part.clone()
        same = [bool(torch.equal(keep[(5, c)], keep[(3, c)])) for (t, c) in keep if t == 5 and (3, c) in keep]
        results.sort()
        best = results[0]
        mine = [r for r in results if picked and (r[2], r[3]) == tuple(picked)]
        five = min((r for r in results if r[2] == 5), default=None)
        seven = min((r for r in results if r[2] == 7), default=None)
        line = f"{K}->{N} rows {rows:5d}: best tile {best[2]} x {best[3]:2d} chunks ({best[4]:3d} wgs) {best[0]:6.1f} us (gemm {best[1]:5.1f})"
        if mine:
            line += f" | plan in the tree: tile {mine[0][2]} x {mine[0][3]:2d} {mine[0][0]:6.1f} us (gemm {mine[0][1]:5.1f})"
        if five:
            line += f" | best tile 5: x {five[3]:2d} {five[0]:6.1f} us (gemm {five[1]:5.1f})"
        if seven:
            line += f" | best tile 7: x {seven[3]:2d} {seven[0]:6.1f} us (gemm {seven[1]:5.1f})"
        line += f" | tile 5 == tile 3 partials: {all(same) if same else 'n/a'}"
        print(line, flush=True)
