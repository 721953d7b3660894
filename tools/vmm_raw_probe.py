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
    out = []
    for off, size in pieces:
        rc_c, h = create(size)
        rc_m = hip.hipMemMap(base + off, size, 0, h, 0) if rc_c == 0 else None
        rc_a = hip.hipMemSetAccess(base + off, size, byref(acc), 1) if (rc_m == 0 and access == "each") else None
        out.append((off // MB, size // MB, rc_c, rc_m, rc_a))
        hip.hipGetLastError()
    if access == "whole":
        lo = min(o for o, _ in pieces)
        hi = max(o + s for o, s in pieces)
        out.append(("whole", hip.hipMemSetAccess(base + lo, hi - lo, byref(acc), 1)))
        hip.hipGetLastError()
    print(title, out, flush=True)


rc, base = reserve(8 << 30, 2 * MB)
print("reserve", rc, hex(base or 0))
seq("gaps, 2 MB each", base, [(i * 4 * MB, 2 * MB) for i in range(4)])
seq("adjacent, 2 MB each (ascending)", base, [(64 * MB + i * 2 * MB, 2 * MB) for i in range(4)])
seq("adjacent, 2 MB each (descending)", base, [(128 * MB + (3 - i) * 2 * MB, 2 * MB) for i in range(4)])
seq("adjacent, access once over the whole", base, [(192 * MB + i * 2 * MB, 2 * MB) for i in range(4)], access="whole")
seq("one 64 MB piece", base, [(256 * MB, 64 * MB)])
seq("one 256 MB piece", base, [(512 * MB, 256 * MB)])
seq("one 1 GB piece", base, [(1024 * MB, 1024 * MB)])
seq("adjacent 64 MB pieces", base, [(2048 * MB + i * 64 * MB, 64 * MB) for i in range(3)])
# a fresh reservation per experiment: is it the position inside the reservation?
rc, b2 = reserve(1 << 30, 2 * MB)
seq("fresh reservation: adjacent 2 MB pieces from offset 0", b2, [(i * 2 * MB, 2 * MB) for i in range(4)])
rc, b3 = reserve(1 << 30, 2 * MB)
seq("fresh reservation: first at 2 MB then at 0", b3, [(2 * MB, 2 * MB), (0, 2 * MB)])
rc, b4 = reserve(1 << 30, 0)
seq("fresh reservation, alignment 0: adjacent", b4, [(i * 2 * MB, 2 * MB) for i in range(4)])
print("done")
