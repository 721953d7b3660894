"""Interleaved A/B of ways to run the dominant GEMM [11264 x 4096] x [4096 x 2048] + bias in bf16 (one process)."""
import torch

M, K, N = 11264, 4096, 2048
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)      # nn.Linear layout [out, in]
Wt = W.t().contiguous()                                                # [in, out]
b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
xT = x.t().contiguous()


def variants():
    v = {
        "addmm(b, x, W.t())  [current]": lambda: torch.addmm(b, x, W.t()),
        "addmm(b, x, Wt)": lambda: torch.addmm(b, x, Wt),
        "addmm out=": lambda: torch.addmm(b, x, W.t(), out=out),
        "F.linear(x, W, b)": lambda: torch.nn.functional.linear(x, W, b),
        "mm(x, W.t()) no bias": lambda: torch.mm(x, W.t()),
        "(W @ xT) transposed problem": lambda: torch.mm(W, xT),
    }
    return v


def time_all(tag):
    v = variants()
    for f in v.values():
        f()
    torch.cuda.synchronize()
    res = {k: [] for k in v}
    for rnd in range(12):
        for k, f in v.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    for k, t in res.items():
        t.sort()
        print(f"{tag:10s} {k:34s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us  -> {2 * M * K * N / t[len(t) // 2] / 1e6:7.0f} TFLOP/s", flush=True)


time_all(str(torch.backends.cuda.preferred_blas_library()).split(".")[-1])
for lib in ("cublas", "cublaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        time_all(lib)
    except Exception as e:   # noqa: BLE001
        print(lib, "failed:", e)
