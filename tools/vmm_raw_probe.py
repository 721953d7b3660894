"""Which sequences of hipMemCreate / hipMemMap / hipMemSetAccess does this ROCm accept?  Raw ctypes calls on libamdhip64
(no torch GPU work besides initialisation).   python tools/vmm_raw_probe.py"""
import ctypes
from ctypes import POINTER, Structure, byref, c_int, c_size_t, c_ubyte, c_ulonglong, c_ushort, c_void_p

import torch

torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")


class Loc(Structure):
    _fields_ = [("type", c_int), ("id", c_int)]


class Flags(Structure):
    _fields_ = [("compressionType", c_ubyte), ("gpuDirectRDMACapable", c_ubyte), ("usage", c_ushort)]


class Prop(Structure):
    _fields_ = [("type", c_int), ("requestedHandleType", c_int), ("location", Loc), ("win32", c_void_p), ("allocFlags", Flags)]


class Access(Structure):
    _fields_ = [("location", Loc), ("flags", c_int)]


prop = Prop()
prop.type, prop.location.type, prop.location.id = 1, 1, 0      # pinned, device 0
acc = Access()
acc.location.type, acc.location.id, acc.flags = 1, 0, 3
hip.hipMemAddressReserve.argtypes = [POINTER(c_void_p), c_size_t, c_size_t, c_void_p, c_ulonglong]
hip.hipMemCreate.argtypes = [POINTER(c_void_p), c_size_t, POINTER(Prop), c_ulonglong]
hip.hipMemMap.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_ulonglong]
hip.hipMemSetAccess.argtypes = [c_void_p, c_size_t, POINTER(Access), c_size_t]
hip.hipMemUnmap.argtypes = [c_void_p, c_size_t]
hip.hipMemRelease.argtypes = [c_void_p]
hip.hipMemAddressFree.argtypes = [c_void_p, c_size_t]
hip.hipMemGetAllocationGranularity.argtypes = [POINTER(c_size_t), POINTER(Prop), c_int]
hip.hipGetLastError.restype = c_int
MB = 1 << 20
for opt in (0, 1):
    g = c_size_t()
    print("granularity opt", opt, hip.hipMemGetAllocationGranularity(byref(g), byref(prop), opt), g.value)


def reserve(n, align=0):
    p = c_void_p()
    rc = hip.hipMemAddressReserve(byref(p), n, align, None, 0)
    return rc, p.value


def create(n):
    h = c_void_p()
    rc = hip.hipMemCreate(byref(h), n, byref(prop), 0)
    return rc, h


def seq(title, base, pieces, access="each"):
    """pieces: list of (offset, size).  Reports the rc of every call."""
import time
import numpy as np


class CAI:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def maps_count():
    try:
        return sum(1 for _ in open("/proc/self/maps"))
    except OSError:
        return -1


print("vm.max_map_count", open("/proc/sys/vm/max_map_count").read().strip(), "maps now", maps_count(), flush=True)
# 1. pieces larger than 2 MB at addresses aligned to the piece size INSIDE a reservation that is only 2 MB aligned
for piece_mb in (4, 8, 16, 32, 64):
    size = piece_mb * MB
    rc, base = reserve(4 << 30, 2 * MB)
    al = (base + size - 1) // size * size
    out = []
    for i in range(6):
        rc_c, h = create(size)
        rc_m = hip.hipMemMap(al + i * size, size, 0, h, 0)
        rc_a = hip.hipMemSetAccess(al + i * size, size, byref(acc), 1)
        hip.hipGetLastError()
        out.append((rc_c, rc_m, rc_a))
    ok = all(o == (0, 0, 0) for o in out)
    res = None
    if ok:
        t = torch.as_tensor(CAI(al, 6 * size), device="cuda").view(torch.int64)
        ar = torch.arange(t.numel(), dtype=torch.int64, device="cuda")
        t.copy_(ar * 2654435761 % 1000003)
        torch.cuda.synchronize()
        res = bool((t == ar * 2654435761 % 1000003).all().item())
        del t, ar
    print(f"piece {piece_mb} MB at size-aligned addresses (reservation base {hex(base)}, first piece at +{(al - base) // MB} MB): calls {out[:2]} ... pattern ok: {res}", flush=True)
# 2. how many 2 MB mappings before something gives: count /proc/self/maps as we go, verify the LAST mapped chunk is usable
rc, base = reserve(200 << 30, 2 * MB)
n_ok, t0 = 0, time.perf_counter()
for i in range(80000):
    rc_c, h = create(2 * MB)
    if rc_c:
        print("hipMemCreate failed at", i, rc_c); break
    rc_m = hip.hipMemMap(base + i * 2 * MB, 2 * MB, 0, h, 0)
    rc_a = hip.hipMemSetAccess(base + i * 2 * MB, 2 * MB, byref(acc), 1) if rc_m == 0 else None
    if rc_m or rc_a:
        print("map failed at", i, rc_m, rc_a, "maps", maps_count()); hip.hipGetLastError(); break
    n_ok = i + 1
    if n_ok % 10000 == 0:
        t = torch.as_tensor(CAI(base + i * 2 * MB, 2 * MB), device="cuda")
        t.fill_(3)
        torch.cuda.synchronize()
        good = int(t[-1].item()) == 3
        del t
        print(f"{n_ok} chunks mapped in {time.perf_counter() - t0:.1f} s; /proc/self/maps lines {maps_count()}; last chunk usable {good}", flush=True)
print("mapped", n_ok, "maps", maps_count())
print("done")
