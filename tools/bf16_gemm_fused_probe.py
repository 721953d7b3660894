"""
A hidden layer of the bf16 engine: the own kernel (rc_gemm_bias_act_bf16) against torch.addmm (hipBLASLt) + rc_act_bf16_inplace.

    python tools/bf16_gemm_fused_probe.py [--rows 11264] [--shapes 4096x2048:2,2048x1024:0] [--reps 20] [--tile 0]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rl-rubiks_amd"))
from librubiks import _hip  # noqa: E402


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
    ap.add_argument("--shapes", default="4096x2048:2,2048x1024:0")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    g = torch.Generator(device="cpu").manual_seed(1)
    results = []
    for spec in args.shapes.split(","):
        shape, act = spec.split(":")
        act = int(act)
        K, N = (int(v) for v in shape.split("x"))
        M = args.rows
        x = (torch.randn(M, K, generator=g) * 0.7).bfloat16().to(dev)
        W = (torch.randn(N, K, generator=g) / np.sqrt(K)).bfloat16().to(dev)
        b = torch.randn(N, generator=g).to(dev)
        bb = b.bfloat16()

        def chain():
            y = torch.addmm(bb, x, W.t())
            if act:
                _hip.check(lib.rc_act_bf16_inplace(y.data_ptr(), y.numel(), act, 1.0, _hip.stream_ptr()), "rc_act_bf16_inplace")
            return y

        def fused():
            y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            _hip.check(lib.rc_gemm_bias_act_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), M, N, K, act, 1.0, y.data_ptr(), args.tile,
                                                 _hip.stream_ptr()), "rc_gemm_bias_act_bf16")
            return y

        rows = torch.cat([torch.arange(0, 64), torch.randint(0, M, (448,), generator=g), torch.arange(M - 64, M)]).to(dev)
        y64 = x[rows].double() @ W.double().t() + b.double()
        ref = torch.where(y64 > 0, y64, torch.expm1(y64)) if act == 2 else torch.relu(y64) if act == 1 else y64
        oc, of = chain().double(), fused().double()
        t_c, t_f = timed(chain, args.reps), timed(fused, args.reps)
        flops = 2.0 * M * N * K
        rec = {"rows": M, "k": K, "n_out": N, "activation": act, "chain_ms": round(t_c, 4), "fused_ms": round(t_f, 4),
               "fused_tflops": round(flops / t_f / 1e9, 1), "chain_tflops": round(flops / t_c / 1e9, 1),
               "max_err_chain_vs_f64": float((oc[rows] - ref).abs().max()), "max_err_fused_vs_f64": float((of[rows] - ref).abs().max()),
               "max_rel_fused_vs_chain": float(((of - oc).abs() / (oc.abs() + 1.0)).max())}
        print(json.dumps(rec), flush=True)
        results.append(rec)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
