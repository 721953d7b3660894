"""
Device arrays whose memory arrives on demand: a reserved address range of the array's full size (`rc_vmm_reserve`), backed by
HBM 2 MiB at a time where rows come into use (`rc_vmm_map`).  The reference's node arrays grow by doubling as a tree grows
(librubiks/solving/agents.py:450-459); here the address of every row is fixed from the start -- kernels, rc_mcts_t and captured
HIP graphs never notice -- and only the rows in use cost memory.  Measured on MI355X (profiles/r4_vmm_probe.txt): ~10 us per
chunk mapped, also next to running kernels; streaming and random-row bandwidth as for ordinary allocations.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint8, c_void_p

import numpy as np
import torch

from librubiks import _hip

CHUNK = 2 << 20
ADDR_KINDS = {0: "unknown", 1: "mapped", 2: "unmapped", 3: "slack", 4: "idle"}   # RC_VMM_ADDR_* (include/rubiks_hip.h)

_hip.register({
    "rc_vmm_granularity": [POINTER(c_size_t)],
    "rc_vmm_reserve": [c_size_t, c_size_t, POINTER(c_void_p)],
    "rc_vmm_map": [c_void_p, c_size_t, c_size_t, POINTER(c_size_t)],
    "rc_vmm_mapped_bytes": [c_void_p, POINTER(c_size_t)],
    "rc_vmm_chunk_map": [c_void_p, POINTER(c_uint8), c_size_t, POINTER(c_size_t)],
    "rc_vmm_release": [c_void_p],
    "rc_vmm_retired_bytes": [POINTER(c_size_t)],
    "rc_vmm_classify": [c_void_p, POINTER(c_int), POINTER(c_void_p), POINTER(c_size_t)],
    "rc_vmm_dump": [c_char_p, c_size_t, POINTER(c_size_t)],
})


def class_bytes(nbytes: int, chunk: int = CHUNK) -> int:
    """Usable bytes of the size class an array of `nbytes` lives in: reservations are whole powers of two (rc_vmm_reserve), of
    which one chunk is alignment slack when chunks are larger than 2 MiB.  Every array of a class reserves exactly this, so that
    parked arrays and idle address ranges are interchangeable within the class, whatever the forest shape they were made for."""
    slack = chunk if chunk > CHUNK else 0
    need, c = (int(nbytes) + chunk - 1) // chunk * chunk + slack, 4 << 20
    while c < need:
        c <<= 1
    return c - slack


def dump() -> str:
    """The node store's state as text: live ranges with their mapped chunk runs, idle ranges, the last 512 events."""
    need = c_size_t()
    _hip.check(_hip.lib().rc_vmm_dump(None, 0, ctypes.byref(need)), "rc_vmm_dump")
    buf = ctypes.create_string_buffer(int(need.value) + 4096)
    _hip.check(_hip.lib().rc_vmm_dump(buf, len(buf), ctypes.byref(need)), "rc_vmm_dump")
    return buf.value.decode()


def classify(addr: int):
    """(kind, base, offset) of a device address -- e.g. the one a GPU memory access fault names; kind is one of ADDR_KINDS' values."""
    kind, base, off = c_int(), c_void_p(), c_size_t()
    _hip.check(_hip.lib().rc_vmm_classify(c_void_p(addr), ctypes.byref(kind), ctypes.byref(base), ctypes.byref(off)), "rc_vmm_classify")
    return ADDR_KINDS[int(kind.value)], int(base.value or 0), int(off.value)


def zeros_or_trim(shape, dtype, device, fill=None) -> torch.Tensor:
    """torch.zeros / torch.full for the large up-front allocations of the search stores (forests allocated up front, A* batches):
    torch's caching allocator cannot see the HBM that parked node stores hold, so on an out-of-memory error that memory is given
    back (`VmmArray.trim`, after a synchronise) and the allocation is tried once more."""
    make = (lambda: torch.zeros(shape, dtype=dtype, device=device)) if fill is None else (lambda: torch.full(shape, fill, dtype=dtype, device=device))
    try:
        return make()
    except torch.OutOfMemoryError:
        if not VmmArray.parked_bytes():
            raise
        torch.cuda.synchronize(device)
        VmmArray.trim()
        torch.cuda.empty_cache()
        return make()


class _Span:
    """What torch.as_tensor reads a device pointer from."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class VmmArray:
    """`nbytes` of reserved device address space; `tensor` views it, `ensure` puts memory behind a byte range of it.
    chunk: the unit memory arrives in (a power of two >= 2 MiB).  A map call costs ~10 us plus ~15 us per 1 000 chunks the
    process has mapped already, and returns only when the GPU has finished the work queued before it
    (profiles/r4_vmm_raw_probe2.txt, r4_vmm_stress.txt): big arrays take bigger chunks, callers map in few large steps.
    The reservation is the array's whole SIZE CLASS (`class_bytes`), so that address ranges are interchangeable within a class:
    a finished owner parks the array with its memory for a successor of the same shape; released arrays leave their address
    range to the next reservation of the class, of whatever shape."""

    _parked = {}     # (class bytes, chunk, device index) -> parked arrays, memory still mapped, oldest first
    # HBM that parked arrays may hold between owners; what is beyond it is released oldest first when an array is parked
    # (RUBIKS_VMM_PARK_GB; torch's allocator cannot see or reclaim parked memory, `trim` gives all of it back)
    PARK_CAP_BYTES = int(float(os.environ.get("RUBIKS_VMM_PARK_GB", "64")) * 2 ** 30)
    _park_clock = 0
    before_trim = []   # callables run at the start of `trim` (owners that hold arrays outside the parking lot hand them in first)

    @classmethod
    def take(cls, nbytes: int, device, chunk: int = CHUNK) -> "VmmArray":
        """An array of at least `nbytes`: one that a finished owner of the same shape has parked (its chunks still mapped), else a
        new reservation (which the library serves from the idle address ranges of the size class before it reserves fresh ones)."""
        free = cls._parked.get((class_bytes(nbytes, chunk), chunk, torch.device(device).index)) or []
        for i in range(len(free) - 1, -1, -1):
            if free[i].asked == int(nbytes):      # the same shape: the memory behind it lies where the new owner's rows will be
                return free.pop(i)
        return cls(nbytes, device, chunk)

    def __init__(self, nbytes: int, device, chunk: int = CHUNK):
        self.lib = _hip.lib()
        self.device = torch.device(device)
        self.chunk = int(chunk)
        self.asked = int(nbytes)
        self.nbytes = class_bytes(nbytes, self.chunk)
        base = c_void_p()
        _hip.check(self.lib.rc_vmm_reserve(self.nbytes, self.chunk, ctypes.byref(base)), "rc_vmm_reserve")
        self.ptr = int(base.value)
        self.have = np.zeros(self.nbytes // self.chunk, dtype=bool)     # host mirror of what is mapped: most calls need no library call
        self.mapped_bytes = 0

    def tensor(self, dtype, shape) -> torch.Tensor:
        """The first prod(shape) elements of the range as a tensor.  Only rows with memory behind them may be touched -- by
        anything, torch ops included."""
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        assert n <= self.nbytes
        t = torch.as_tensor(_Span(self.ptr, n), device=self.device)
        assert t.data_ptr() == self.ptr
        return t.view(dtype).view(*shape)

    def _resync(self):
        """The host mirror from the library's own record (after a map call that failed part of the way)."""
        n = c_size_t()
        flags = np.zeros(len(self.have), dtype=np.uint8)
        _hip.check(self.lib.rc_vmm_chunk_map(self.ptr, flags.ctypes.data_as(POINTER(c_uint8)), len(flags), ctypes.byref(n)), "rc_vmm_chunk_map")
        assert int(n.value) == len(self.have)
        self.have[:] = flags != 0
        self.mapped_bytes = int(self.have.sum()) * self.chunk

    def ensure(self, lo: int, hi: int) -> int:
        """Memory behind bytes [lo, hi); returns the bytes newly mapped.  Host-synchronous (waits for the GPU to drain)."""
        if hi <= lo:
            return 0
        c0, c1 = lo // self.chunk, (min(hi, self.nbytes) - 1) // self.chunk
        if self.have[c0:c1 + 1].all():
            return 0
        new, total = c_size_t(), 0
        rc = self.lib.rc_vmm_map(self.ptr, c0 * self.chunk, (c1 - c0 + 1) * self.chunk, ctypes.byref(new))
        total += int(new.value)
        if rc != 0 and any(self._parked.values()):
            # out of memory while parked arrays still hold theirs.  Releasing unmaps: everything queued so far must have finished
            # with whatever it reads (parked arrays have no readers, but the release also flushes translations device-wide).
            torch.cuda.synchronize(self.device)
            self.trim()
            rc = self.lib.rc_vmm_map(self.ptr, c0 * self.chunk, (c1 - c0 + 1) * self.chunk, ctypes.byref(new))
            total += int(new.value)
        if rc != 0:
            self._resync()           # the chunks mapped before the failure stay mapped: keep the books right, then report
            _hip.check(rc, "rc_vmm_map")
        self.have[c0:c1 + 1] = True
        self.mapped_bytes += total
        return total

    @classmethod
    def retired_bytes(cls) -> int:
        """Address space of released arrays idle on the library's per-class lists (reused by the next reservation of the class)."""
        out = c_size_t()
        _hip.check(_hip.lib().rc_vmm_retired_bytes(ctypes.byref(out)), "rc_vmm_retired_bytes")
        return int(out.value)

    @classmethod
    def parked_bytes(cls) -> int:
        """HBM mapped behind parked arrays."""
        return sum(a.mapped_bytes for arrs in cls._parked.values() for a in arrs)

    @classmethod
    def has_parked(cls, nbytes: int, device, chunk: int = CHUNK) -> bool:
        return any(a.asked == int(nbytes) for a in cls._parked.get((class_bytes(nbytes, chunk), chunk, torch.device(device).index), []))

    def park(self, protect_from: int = None):
        """The owner is done with the array (it has synchronised and dropped its tensors).  Address range and memory are kept for
        the next array of the same shape -- successive forests of one benchmark or evaluation -- instead of being unmapped and
        mapped again.  Parked memory beyond PARK_CAP_BYTES is released, oldest arrays first -- except the arrays parked from
        `protect_from` on (`next_park_mark()` taken before a forest parks its arrays): the forest that has just finished always
        stays whole, whatever its size, because its successor is the likeliest next owner (releasing and re-mapping the 146 GB
        behind a config-5-size forest costs the next search tens of seconds).  `trim` releases everything."""
        if not self.ptr:
            return
        VmmArray._park_clock += 1
        self._parked_at = VmmArray._park_clock
        self._parked.setdefault((self.nbytes, self.chunk, self.device.index), []).append(self)
        while self.parked_bytes() > self.PARK_CAP_BYTES:
            old = [a for arrs in self._parked.values() for a in arrs if protect_from is None or a._parked_at < protect_from]
            if not old:
                break
            victim = min(old, key=lambda a: a._parked_at)
            self._parked[(victim.nbytes, victim.chunk, victim.device.index)].remove(victim)
            victim.close()

    @classmethod
    def next_park_mark(cls) -> int:
        """What `park(protect_from=...)` takes: arrays parked after this call belong to one owner and are not evicted on its behalf."""
        return cls._park_clock + 1

    @classmethod
    def trim(cls) -> int:
        """Releases every parked array (memory back to the device, address range to its class's idle list); returns how many.
        Call it before a large torch allocation in a process that has searched with node stores mapped on demand: torch's
        caching allocator cannot see or reclaim parked memory."""
        for hook in cls.before_trim:
            hook()
        n = 0
        for arrs in cls._parked.values():
            while arrs:
                arrs.pop().close()
                n += 1
        return n

    def close(self):
        """Gives the memory back.  The caller has synchronised with every kernel that used the array and holds no tensor of it
        any more.  The library unmaps, flushes the GPU's translations (memory mapped at an address that was mapped before is
        otherwise not coherent on this platform: rc_vmm_release, tools/vmm_remap_probe.hip) and keeps the address range for the
        next reservation of its size class."""
        if self.ptr:
            ptr, self.ptr = self.ptr, 0
            _hip.check(self.lib.rc_vmm_release(ptr), "rc_vmm_release")
