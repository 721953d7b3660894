"""
The hidden-layer kernel of the split network (rc_split_layer_f16) by tile, row count and operand data, against the library's
bf16 GEMM of the same shape as the yardstick:

    python tools/gemm_tile_ab.py [--rows 11264,196608,524288] [--tiles 1,4,6] [--k 4096] [--n 2048] [--reps 10] [--out f.json]

Per (rows, tile): milliseconds (HIP events over `reps` launches), TFLOP/s of executed f16 flops (3 products), fraction of the
2.5 PFLOP/s dense f16 peak, and whether the output equals tile 1's bit for bit (every tile walks K in the same order).
`hipblaslt_bf16`: torch.mm of [rows, k] x [k, n] bf16 (one product): what the library reaches on this box in this run.
RUBIKS_HIP_LIB selects another build of the library (tools/build_ab_lib.sh): the same command under two builds is the A/B.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rl-rubiks_amd"))
from librubiks import _hip  # noqa: E402

PEAK = 2500.0


def timed(fn, reps):
    for _ in range(2):
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
    ap.add_argument("--rows", default="11264,196608,524288")
    ap.add_argument("--tiles", default="1")
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--data", default="elu", choices=["elu", "zeros"], help="elu: ELU(N(0,1)) activations and N(0,1/sqrt k) weights, split exactly")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.lib()
    K, N = args.k, args.n
    g = torch.Generator(device=dev).manual_seed(1)
    W = torch.randn(N, K, generator=g, device=dev) / K ** 0.5
    if args.data == "zeros":
        W.zero_()
    wh = W.half()
    wl = ((W - wh.float()) * 2048.0).half()
    W3 = torch.cat([wl, wh, wh], 1).contiguous()
    Wb = W.bfloat16().contiguous()
    bias = torch.randn(N, generator=g, device=dev)
    out = {"lib": os.environ.get("RUBIKS_HIP_LIB", "tree"), "k": K, "n": N, "data": args.data, "runs": []}
    for M in (int(v) for v in args.rows.split(",")):
        x = torch.nn.functional.elu(torch.randn(M, K, generator=g, device=dev))
        if args.data == "zeros":
            x.zero_()
        xh = x.half()
        a = torch.cat([xh, ((x - xh.float()) * 2048.0).half()], 1).contiguous()
        xb = x.bfloat16().contiguous()
        del x, xh
        y = torch.empty((M, 2 * N), dtype=torch.float16, device=dev)
        first = None
        flops = 3 * 2.0 * M * N * K
        for tile in (int(v) for v in args.tiles.split(",")):
            def run(tile=tile):
                _hip.check(lib.rc_split_gemm_f16(a.data_ptr(), W3.data_ptr(), bias.data_ptr(), M, N, K, 2, 1.0, y.data_ptr(), None, tile,
                                                 _hip.stream_ptr()), "rc_split_gemm_f16")
            try:
                ms = timed(run, args.reps)
            except _hip.RubiksHipError as e:
                print(json.dumps({"rows": M, "tile": tile, "error": str(e)}), flush=True)
                continue
            same = None
            if first is None:
                first = y.clone()
            else:
                same = bool(torch.equal(first, y))
            v = y.view(torch.int32).to(torch.int64).reshape(-1)
            checksum = int((v * (torch.arange(v.numel(), device=dev, dtype=torch.int64) % 8191 + 1)).sum().item()) & ((1 << 62) - 1)   # equal across builds <=> same bits
            rec = {"rows": M, "tile": tile, "ms": round(ms, 4), "tflops_f16": round(flops / ms / 1e9, 1), "frac": round(flops / ms / 1e9 / PEAK, 4),
                   "bit_identical_to_first_tile": same, "checksum": checksum}
            print(json.dumps(rec), flush=True)
            out["runs"].append(rec)
        del first
        yb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ms = timed(lambda: torch.mm(xb, Wb.t(), out=yb), args.reps)
        rec = {"rows": M, "tile": "hipblaslt_bf16", "ms": round(ms, 4), "tflops_bf16": round(2.0 * M * N * K / ms / 1e9, 1),
               "frac": round(2.0 * M * N * K / ms / 1e9 / PEAK, 4)}
        print(json.dumps(rec), flush=True)
        out["runs"].append(rec)
        for tile in (1,):    # the own kernel as a plain bf16 layer (one product, K as is): the same schedule on bf16 operands
            yb2 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            try:
                ms = timed(lambda: _hip.check(lib.rc_gemm_bias_act_bf16(xb.data_ptr(), Wb.data_ptr(), bias.data_ptr(), M, N, K, 2, 1.0, yb2.data_ptr(), tile,
                                                                        _hip.stream_ptr()), "rc_gemm_bias_act_bf16"), args.reps)
                rec = {"rows": M, "tile": f"own_bf16_tile{tile}", "ms": round(ms, 4), "tflops_bf16": round(2.0 * M * N * K / ms / 1e9, 1),
                       "frac": round(2.0 * M * N * K / ms / 1e9 / PEAK, 4)}
                print(json.dumps(rec), flush=True)
                out["runs"].append(rec)
            except _hip.RubiksHipError as e:
                print(json.dumps({"rows": M, "tile": f"own_bf16_tile{tile}", "error": str(e)}), flush=True)
        del a, xb, y, yb, yb2
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
