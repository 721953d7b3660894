"""Probe: f16 x f16 -> f32 GEMMs through torch / hipBLASLt, and the accuracy of the f16x3 split against fp64."""
import time
import torch

torch.manual_seed(0)
M, K, N = 11264, 4096, 2048
dev = "cuda"
x = torch.randn(M, K, device=dev) * 0.7
W = torch.randn(N, K, device=dev) * 0.03
b = torch.randn(N, device=dev) * 0.1


def t_ms(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


S = 2.0 ** 11
xh = x.half()
xl = ((x - xh.float()) * S).half()
Wh = W.half()
Wl = ((W - Wh.float()) * S).half()
ok = {}
try:
    c1 = torch.mm(xh, Wh.t(), out_dtype=torch.float32)
    ok["mm_out_dtype"] = True
except Exception as e:   # noqa: BLE001
    ok["mm_out_dtype"] = repr(e)[:200]
print(ok)
if ok["mm_out_dtype"] is True:
    A2 = torch.cat([xh, xl], 1).contiguous()
    B2 = torch.cat([Wl, Wh], 1).contiguous()
    f_main = lambda: torch.mm(A2[:, :K], Wh.t(), out_dtype=torch.float32)   # noqa: E731
    f_corr = lambda: torch.mm(A2, B2.t(), out_dtype=torch.float32)          # noqa: E731
    print("main K   ms", t_ms(f_main), "TF", 2 * M * K * N / t_ms(f_main) / 1e9)
    print("corr 2K  ms", t_ms(f_corr), "TF", 2 * M * 2 * K * N / t_ms(f_corr) / 1e9)
    print("contig main ms", t_ms(lambda: torch.mm(xh, Wh.t(), out_dtype=torch.float32)))
    print("f16->f16 ms", t_ms(lambda: torch.mm(xh, Wh.t())))
    print("bf16 addmm ms", t_ms(lambda: torch.addmm(b.bfloat16(), x.bfloat16(), W.bfloat16().t())))
    print("f32 native ms", t_ms(lambda: torch.addmm(b, x, W.t()), 3))
    ref = (x.double() @ W.double().t())
    nat = x @ W.t()
    spl = f_main() + f_corr() / S
    bf = (x.bfloat16() @ W.bfloat16().t()).float()
    hf = torch.mm(xh, Wh.t(), out_dtype=torch.float32)
    sc = ref.abs().mean().item()
    for name, v in (("f32 native", nat), ("f16x3 split", spl), ("f16 plain", hf), ("bf16", bf)):
        err = (v.double() - ref).abs()
        print(f"{name:12s} max abs err {err.max().item():.3e}  mean abs err {err.mean().item():.3e}  (mean |ref| {sc:.3f})")
    # small magnitudes: activations near zero and tiny weights
    x2 = x * 1e-3
    xh2 = x2.half(); xl2 = ((x2 - xh2.float()) * S).half()
    ref2 = x2.double() @ W.double().t()
    spl2 = torch.mm(xh2, Wh.t(), out_dtype=torch.float32) + (torch.mm(xh2, Wl.t(), out_dtype=torch.float32) + torch.mm(xl2, Wh.t(), out_dtype=torch.float32)) / S
    nat2 = x2 @ W.t()
    for name, v in (("f32 native small", nat2), ("split small", spl2)):
        err = (v.double() - ref2).abs()
        print(f"{name:18s} max abs err {err.max().item():.3e} mean {err.mean().item():.3e} (mean |ref| {ref2.abs().mean().item():.3e})")
try:
    c = torch.zeros(M, N, device=dev)
    torch.addmm(c, xh, Wh.t(), out_dtype=torch.float32)
    print("addmm out_dtype ok")
except Exception as e:   # noqa: BLE001
    print("addmm out_dtype:", repr(e)[:200])
