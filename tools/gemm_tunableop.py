"""Does PyTorch's TunableOp find a faster library kernel for the two hidden GEMM shapes of a step?  (bf16, random data)"""
import os
import time

os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "gpurun_out/tunableop_results.csv")
import torch  # noqa: E402

torch.manual_seed(0)
for (M, K, N) in ((11264, 4096, 2048), (11264, 2048, 1024)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    torch.cuda.tunable.enable(False)
    for _ in range(3):
        torch.addmm(b, x, W.t())
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.addmm(b, x, W.t())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    base = ts[len(ts) // 2]
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    t0 = time.perf_counter()
    torch.addmm(b, x, W.t())          # tunes this shape
    torch.cuda.synchronize()
    tune_s = time.perf_counter() - t0
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.addmm(b, x, W.t())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    ts.sort()
    print(f"[{M} x {K}] x [{K} x {N}]: default {base:.1f} us ({2 * M * K * N / base / 1e6:.0f} TFLOP/s), tuned {ts[len(ts) // 2]:.1f} us "
          f"({2 * M * K * N / ts[len(ts) // 2] / 1e6:.0f} TFLOP/s), tuning took {tune_s:.1f} s", flush=True)
print(torch.cuda.tunable.get_results()[-4:])
