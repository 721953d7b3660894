"""Probe of HIP virtual memory management on the GPU box (rc_vmm_*): granularity, cost of mapping, mapping next to running
kernels and under HIP-graph replay, torch views of a reserved range.   python tools/vmm_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip  # noqa: E402

P, SZ = ctypes.c_void_p, ctypes.c_size_t
_hip.register({"rc_vmm_granularity": [ctypes.POINTER(SZ)], "rc_vmm_reserve": [SZ, SZ, ctypes.POINTER(P)],
               "rc_vmm_map": [P, SZ, SZ, ctypes.POINTER(SZ)], "rc_vmm_mapped_bytes": [P, ctypes.POINTER(SZ)], "rc_vmm_release": [P]})
lib = _hip.lib()
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")


class CAI:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


g = SZ()
_hip.check(lib.rc_vmm_granularity(ctypes.byref(g)), "gran")
print("granularity", g.value, flush=True)
MB = 1 << 20
free0 = torch.cuda.mem_get_info()[0]
for chunk_mb in (2,):
    base = P()
    total = 64 << 30
    t0 = time.perf_counter()
    _hip.check(lib.rc_vmm_reserve(total, chunk_mb * MB, ctypes.byref(base)), "reserve")
    t_res = time.perf_counter() - t0
    # map 64 single chunks one call each
    ts = []
    new = SZ()
    for i in range(64):
        t0 = time.perf_counter()
        _hip.check(lib.rc_vmm_map(base, i * chunk_mb * MB * 2, chunk_mb * MB, ctypes.byref(new)), "map")
        ts.append(time.perf_counter() - t0)
    # one call mapping a run of 4 GB
    t0 = time.perf_counter()
    _hip.check(lib.rc_vmm_map(base, 32 << 30, 4 << 30, ctypes.byref(new)), "map 4 GB (runs of <= 32 MB inside)")
    t_run = time.perf_counter() - t0
    print(f"chunk {chunk_mb} MB: reserve 64 GB {t_res*1e3:.2f} ms; single-chunk map median {np.median(ts)*1e6:.0f} us max {np.max(ts)*1e6:.0f} us; "
          f"4 GB in one call {t_run*1e3:.1f} ms (new {new.value >> 20} MB); free now {torch.cuda.mem_get_info()[0] >> 20} MB", flush=True)
    # torch view of the first chunk
    t = torch.as_tensor(CAI(base.value, chunk_mb * MB), device="cuda")
    t.fill_(7)
    assert int(t[:1000].sum().item()) == 7000 and t.data_ptr() == base.value
    v32 = t.view(torch.int32)
    v32[:16] = torch.arange(16, dtype=torch.int32, device="cuda")
    assert v32[:16].cpu().tolist() == list(range(16))
    # integrity over many chunks: a 4 GB pattern written and read back through the view (chunks at 32 GB .. 36 GB)
    w = torch.as_tensor(CAI(base.value + (32 << 30), 4 << 30), device="cuda").view(torch.int64)
    ar = torch.arange(w.numel(), dtype=torch.int64, device="cuda")
    w.copy_(ar * 2654435761 % 1000003)
    torch.cuda.synchronize()
    assert bool((w == ar * 2654435761 % 1000003).all().item())
    assert int(w[(2 * MB) // 8 - 1].item()) == ((2 * MB) // 8 - 1) * 2654435761 % 1000003      # last element of a chunk
    print("   4 GB pattern across 2048 chunks: ok", flush=True)
    del t, v32, w, ar
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _hip.check(lib.rc_vmm_release(base), "release")
    print(f"   release {1e3*(time.perf_counter()-t0):.2f} ms; free now {torch.cuda.mem_get_info()[0] >> 20} MB (start {free0 >> 20})", flush=True)

# mapping while a kernel runs on the mapped part, and under graph replay
base = P()
chunk = 2 * MB
_hip.check(lib.rc_vmm_reserve(8 << 30, chunk, ctypes.byref(base)), "reserve")
new = SZ()
_hip.check(lib.rc_vmm_map(base, 0, 1 << 30, ctypes.byref(new)), "map")
a = torch.as_tensor(CAI(base.value, 1 << 30), device="cuda").view(torch.float32)
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)


def busy(n=40):
    for _ in range(n):
        torch.mm(x, x)
        a.add_(1.0)


busy(2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); busy(); e1.record(); torch.cuda.synchronize()
alone = e0.elapsed_time(e1)
e0.record(); busy(); e1.record()
ts = []
for i in range(200):
    t0 = time.perf_counter()
    _hip.check(lib.rc_vmm_map(base, (1 << 30) + i * chunk, chunk, ctypes.byref(new)), "map while busy")
    ts.append(time.perf_counter() - t0)
host_done_before_gpu = not e1.query()
torch.cuda.synchronize()
print(f"GPU work alone {alone:.1f} ms, with 200 maps next to it {e0.elapsed_time(e1):.1f} ms; map median {np.median(ts)*1e6:.0f} us max {np.max(ts)*1e6:.0f} us; "
      f"maps finished while the GPU was still busy: {host_done_before_gpu}", flush=True)
# graph: writes into a tensor view over [1 GB, 1 GB + 400 MB) -- mapped above
b = torch.as_tensor(CAI(base.value + (1 << 30), 200 * chunk), device="cuda").view(torch.float32)
b.zero_()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    b.add_(1.0)
    a.add_(1.0)
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
assert float(b[-1].item()) == 3.0 and float(b[0].item()) == 3.0
# now a graph over a range that gets mapped AFTER capture? (capture needs valid memory only at replay)
c_lo = (1 << 30) + 200 * chunk
_hip.check(lib.rc_vmm_map(base, c_lo, chunk, ctypes.byref(new)), "map")
c = torch.as_tensor(CAI(base.value + c_lo, 2 * chunk), device="cuda").view(torch.float32)   # second half not mapped yet
half = c[:chunk // 4]
half.zero_()
late = c[chunk // 4: chunk // 4 + 4096]      # lies in the chunk that is NOT mapped yet: only captured (a capture runs nothing)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    late.add_(5.0)
    half.add_(1.0)
_hip.check(lib.rc_vmm_map(base, c_lo + chunk, chunk, ctypes.byref(new)), "map later")
late.zero_()
g2.replay(); torch.cuda.synchronize()
assert float(late[7].item()) == 5.0 and float(half[0].item()) == 1.0
print("graph replay into a chunk mapped after capture: ok", flush=True)
m = SZ()
_hip.check(lib.rc_vmm_mapped_bytes(base, ctypes.byref(m)), "mapped")
print("mapped MB", m.value >> 20, "torch allocated MB", torch.cuda.memory_allocated() >> 20, "free MB", torch.cuda.mem_get_info()[0] >> 20, flush=True)
# big: how long do 100 GB in 2 MB chunks (single calls) take, and in one run
del a, b, c, half
torch.cuda.synchronize()
_hip.check(lib.rc_vmm_release(base), "release")
base = P()
_hip.check(lib.rc_vmm_reserve(200 << 30, chunk, ctypes.byref(base)), "reserve")
t0 = time.perf_counter()
for i in range(100):
    _hip.check(lib.rc_vmm_map(base, i << 30, 1 << 30, ctypes.byref(new)), "map 1 GB")
print(f"100 GB in 1 GB runs: {time.perf_counter()-t0:.3f} s", flush=True)
t0 = time.perf_counter()
for i in range(4096):
    _hip.check(lib.rc_vmm_map(base, (100 << 30) + 2 * i * chunk, chunk, ctypes.byref(new)), "map")
print(f"4096 separate 2 MB chunks: {time.perf_counter()-t0:.3f} s", flush=True)
big = torch.as_tensor(CAI(base.value, 100 << 30), device="cuda")
e0.record(); big.fill_(1); e1.record(); torch.cuda.synchronize()
print(f"fill 100 GB: {e0.elapsed_time(e1):.1f} ms = {100*1.073741824/e0.elapsed_time(e1)*1e3:.0f} GB/s", flush=True)
reg = torch.empty(20 << 30, dtype=torch.uint8, device="cuda")
e0.record(); reg.fill_(1); e1.record(); torch.cuda.synchronize()
print(f"fill 20 GB of a torch allocation: {20*1.073741824/e0.elapsed_time(e1)*1e3:.0f} GB/s", flush=True)
# random 256-byte rows (the tree kernels' access pattern) from a 64 GB region: VMM range vs a torch allocation
rows = (64 << 30) // 256
idx = torch.randint(0, rows, (1 << 24,), device="cuda")
out = torch.empty((1 << 24, 64), dtype=torch.int32, device="cuda")
v = big[:64 << 30].view(torch.int32).view(rows, 64)
for name, src in (("vmm", v), ("torch", None)):
    if src is None:
        del big, v
        reg64 = torch.empty(64 << 30, dtype=torch.uint8, device="cuda")
        reg64.fill_(1)
        src = reg64.view(torch.int32).view(rows, 64)
    torch.index_select(src, 0, idx, out=out)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        torch.index_select(src, 0, idx, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"gather of 2^24 random 256-B rows out of 64 GB, {name}: {ms:.2f} ms = {(1 << 24) * 256 / ms / 1e6:.0f} GB/s read", flush=True)
del reg, src, out
reg64 = None
t0 = time.perf_counter()
_hip.check(lib.rc_vmm_release(base), "release")
print(f"release: {time.perf_counter()-t0:.3f} s; free MB {torch.cuda.mem_get_info()[0] >> 20}", flush=True)
print("vmm probe ok")
