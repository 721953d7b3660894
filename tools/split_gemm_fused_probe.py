"""
One hidden layer of the split network: the fused kernel (rc_split_gemm_f16) against the library chain it replaces
(two torch.mm with fp32 output + rc_split_act_f16), for correctness (against float64 on a row sample) and time.

    python tools/split_gemm_fused_probe.py [--rows 11264] [--shapes 4096x2048,2048x512] [--reps 20] [--tile 0]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rl-rubiks_amd"))
from librubiks import _hip  # noqa: E402

SCALE = 2048.0


def split(x64: torch.Tensor):
    hi = x64.half()
    lo = ((x64 - hi.double()) * SCALE).half()
    return hi, lo


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=11264)
    ap.add_argument("--shapes", default="4096x2048,2048x512")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: what the kernel does when it is not limited by power")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    g = torch.Generator(device="cpu").manual_seed(1)
    results = []
    for shape in args.shapes.split(","):
        K, N = (int(v) for v in shape.split("x"))
        M = args.rows
        x = (torch.randn(M, K, generator=g, dtype=torch.float64) * 0.7).float().double()   # fp32-representable activations
        W = (torch.randn(N, K, generator=g, dtype=torch.float64) / np.sqrt(K)).float().double()
        b = torch.randn(N, generator=g, dtype=torch.float64).float()
        if args.zeros:
            x, W = x * 0, W * 0
        xh, xl = split(x)
        wh, wl = split(W)
        a = torch.cat([xh, xl], 1).contiguous().to(dev)
        Wh = wh.contiguous().to(dev)
        B2 = torch.cat([wl, wh], 1).contiguous().to(dev)
        W3 = torch.cat([wl, wh, wh], 1).contiguous().to(dev)
        bias = b.to(dev)
        for split_out in (True, False):
            def chain():
                c = torch.mm(a[:, :K], Wh.t(), out_dtype=torch.float32)
                corr = torch.mm(a, B2.t(), out_dtype=torch.float32)
                out = torch.empty((M, 2 * N), dtype=torch.float16, device=dev) if split_out else torch.empty((M, N), dtype=torch.float32, device=dev)
                _hip.check(lib.rc_split_act_f16(c.data_ptr(), corr.data_ptr(), 1.0 / SCALE, M, N, bias.data_ptr(), 2, 1.0,
                                                out.data_ptr() if split_out else None, None if split_out else out.data_ptr(),
                                                _hip.stream_ptr()), "rc_split_act_f16")
                return out

            def fused():
                out = torch.empty((M, 2 * N), dtype=torch.float16, device=dev) if split_out else torch.empty((M, N), dtype=torch.float32, device=dev)
                _hip.check(lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), bias.data_ptr(), M, N, K, 2, 1.0,
                                                 out.data_ptr() if split_out else None, None if split_out else out.data_ptr(),
                                                 args.tile, _hip.stream_ptr()), "rc_split_gemm_f16")
                return out

            def value(o):
                o = o.cpu()
                return o[:, :N].double() + o[:, N:].double() / SCALE if split_out else o.double()

            rows = torch.cat([torch.arange(0, 64), torch.randint(0, M, (448,), generator=g), torch.arange(M - 64, M)])
            xs = xh[rows].double() + xl[rows].double() / SCALE
            ws = wh.double() + wl.double() / SCALE
            y = xs @ ws.t() + b.double()
            ref = torch.where(y > 0, y, torch.expm1(y))
            oc, of = value(chain()), value(fused())
            err_c = (oc[rows] - ref).abs()
            err_f = (of[rows] - ref).abs()
            diff = (oc - of).abs().max().item()
            t_c, t_f = timed(chain, args.reps), timed(fused, args.reps)
            flops = 3 * 2.0 * M * N * K
            rec = {"rows": M, "k": K, "n_out": N, "out": "hi|lo halves" if split_out else "fp32",
                   "chain_ms": round(t_c, 4), "fused_ms": round(t_f, 4), "fused_tflops_f16": round(flops / t_f / 1e9, 1),
                   "chain_tflops_f16": round(flops / t_c / 1e9, 1),
                   "max_err_chain_vs_f64": float(err_c.max()), "max_err_fused_vs_f64": float(err_f.max()),
                   "mean_err_chain_vs_f64": float(err_c.mean()), "mean_err_fused_vs_f64": float(err_f.mean()),
                   "max_abs_fused_minus_chain": diff}
            print(json.dumps(rec), flush=True)
            results.append(rec)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
