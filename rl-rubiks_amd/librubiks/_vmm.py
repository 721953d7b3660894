"""
Device arrays whose memory arrives on demand: a reserved address range of the array's full size (`rc_vmm_reserve`), backed by
HBM 2 MiB at a time where rows come into use (`rc_vmm_map`).  The reference's node arrays grow by doubling as a tree grows
(librubiks/solving/agents.py:450-459); here the address of every row is fixed from the start -- kernels, rc_mcts_t and captured
HIP graphs never notice -- and only the rows in use cost memory.  Measured on MI355X (profiles/r4_vmm_probe.txt): ~10 us per
chunk mapped, also next to running kernels; streaming and random-row bandwidth as for ordinary allocations.
"""
import ctypes
from ctypes import POINTER, c_size_t, c_void_p

import numpy as np
import torch

from librubiks import _hip

CHUNK = 2 << 20

_hip.register({
    "rc_vmm_granularity": [POINTER(c_size_t)],
    "rc_vmm_reserve": [c_size_t, c_size_t, POINTER(c_void_p)],
    "rc_vmm_map": [c_void_p, c_size_t, c_size_t, POINTER(c_size_t)],
    "rc_vmm_mapped_bytes": [c_void_p, POINTER(c_size_t)],
    "rc_vmm_release": [c_void_p],
    "rc_vmm_retired_bytes": [POINTER(c_size_t)],
})


class _Span:
    """What torch.as_tensor reads a device pointer from."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class VmmArray:
    """`nbytes` of reserved device address space; `tensor` views it, `ensure` puts memory behind a byte range of it.
    chunk: the unit memory arrives in (a power of two >= 2 MiB).  A map call costs ~10 us plus ~15 us per 1 000 chunks the
    process has mapped already, and returns only when the GPU has finished the work queued before it
    (profiles/r4_vmm_raw_probe2.txt, r4_vmm_stress.txt): big arrays take bigger chunks, callers map in few large steps."""

    _parked = {}   # (nbytes, chunk, device index) -> released arrays, memory still mapped: the next array of that shape takes one over

    @classmethod
    def take(cls, nbytes: int, device, chunk: int = CHUNK) -> "VmmArray":
        """An array of this shape: one that a finished owner has parked (same address range, its chunks still mapped), else a
        new reservation."""
        nb = (int(nbytes) + chunk - 1) // chunk * chunk
        free = cls._parked.get((nb, chunk, torch.device(device).index))
        return free.pop() if free else cls(nbytes, device, chunk)

    def __init__(self, nbytes: int, device, chunk: int = CHUNK):
        self.lib = _hip.lib()
        self.device = torch.device(device)
        self.chunk = int(chunk)
        self.nbytes = (int(nbytes) + chunk - 1) // chunk * chunk
        base = c_void_p()
        _hip.check(self.lib.rc_vmm_reserve(self.nbytes, self.chunk, ctypes.byref(base)), "rc_vmm_reserve")
        self.ptr = int(base.value)
        self.have = np.zeros(self.nbytes // self.chunk, dtype=bool)     # host mirror of what is mapped: most calls need no library call
        self.mapped_bytes = 0

    def tensor(self, dtype, shape) -> torch.Tensor:
        """The whole range as a tensor.  Only rows with memory behind them may be touched -- by anything, torch ops included."""
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        assert n <= self.nbytes
        t = torch.as_tensor(_Span(self.ptr, n), device=self.device)
        assert t.data_ptr() == self.ptr
        return t.view(dtype).view(*shape)

    def ensure(self, lo: int, hi: int) -> int:
        """Memory behind bytes [lo, hi); returns the bytes newly mapped.  Host-synchronous (waits for the GPU to drain)."""
        if hi <= lo:
            return 0
        c0, c1 = lo // self.chunk, (min(hi, self.nbytes) - 1) // self.chunk
        if self.have[c0:c1 + 1].all():
            return 0
        new = c_size_t()
        rc = self.lib.rc_vmm_map(self.ptr, c0 * self.chunk, (c1 - c0 + 1) * self.chunk, ctypes.byref(new))
        if rc != 0 and self.trim():          # out of memory with released arrays still holding theirs: give that back, try again
            rc = self.lib.rc_vmm_map(self.ptr, c0 * self.chunk, (c1 - c0 + 1) * self.chunk, ctypes.byref(new))
        _hip.check(rc, "rc_vmm_map")
        self.have[c0:c1 + 1] = True
        self.mapped_bytes += int(new.value)
        return int(new.value)

    @classmethod
    def retired_bytes(cls) -> int:
        """Address space of closed arrays that stays reserved so that nothing is mapped there again."""
        out = c_size_t()
        _hip.check(_hip.lib().rc_vmm_retired_bytes(ctypes.byref(out)), "rc_vmm_retired_bytes")
        return int(out.value)

    @classmethod
    def has_parked(cls, nbytes: int, device, chunk: int = CHUNK) -> bool:
        nb = (int(nbytes) + chunk - 1) // chunk * chunk
        return bool(cls._parked.get((nb, chunk, torch.device(device).index)))

    def park(self):
        """The owner is done with the array (it has synchronised and dropped its tensors).  Address range and memory are kept for
        the next array of the same shape -- successive forests of one benchmark or evaluation are that -- instead of being
        unmapped and mapped again; `trim` really releases them."""
        if self.ptr:
            self._parked.setdefault((self.nbytes, self.chunk, self.device.index), []).append(self)

    @classmethod
    def trim(cls) -> int:
        """Releases every parked array (memory and address range); returns how many."""
        n = 0
        for arrs in cls._parked.values():
            while arrs:
                arrs.pop().close()
                n += 1
        return n

    def close(self):
        """Gives the memory back.  The caller has synchronised with every kernel that used the array and holds no tensor of it
        any more.  The address range is retired, not reused: on this platform memory mapped at an address that was mapped before
        is not coherent (rc_vmm_release, tools/vmm_remap_probe.hip) -- which is also why arrays are parked rather than closed
        wherever a successor of the same shape is likely."""
        if self.ptr:
            ptr, self.ptr = self.ptr, 0
            _hip.check(self.lib.rc_vmm_release(ptr), "rc_vmm_release")
