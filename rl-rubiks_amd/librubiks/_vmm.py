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
})


class _Span:
    """What torch.as_tensor reads a device pointer from."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class VmmArray:
    """`nbytes` of reserved device address space; `tensor` views it, `ensure` puts memory behind a byte range of it."""

    def __init__(self, nbytes: int, device):
        self.lib = _hip.lib()
        self.device = device
        self.nbytes = (int(nbytes) + CHUNK - 1) // CHUNK * CHUNK
        base = c_void_p()
        _hip.check(self.lib.rc_vmm_reserve(self.nbytes, CHUNK, ctypes.byref(base)), "rc_vmm_reserve")
        self.ptr = int(base.value)
        self.have = np.zeros(self.nbytes // CHUNK, dtype=bool)     # host mirror of what is mapped: most calls need no library call
        self.mapped_bytes = 0

    def tensor(self, dtype, shape) -> torch.Tensor:
        """The whole range as a tensor.  Only rows with memory behind them may be touched -- by anything, torch ops included."""
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        assert n <= self.nbytes
        t = torch.as_tensor(_Span(self.ptr, n), device=self.device)
        assert t.data_ptr() == self.ptr
        return t.view(dtype).view(*shape)

    def ensure(self, lo: int, hi: int) -> int:
        """Memory behind bytes [lo, hi); returns the bytes newly mapped.  Host-synchronous, safe next to running kernels."""
        if hi <= lo:
            return 0
        c0, c1 = lo // CHUNK, (min(hi, self.nbytes) - 1) // CHUNK
        if self.have[c0:c1 + 1].all():
            return 0
        new = c_size_t()
        _hip.check(self.lib.rc_vmm_map(self.ptr, c0 * CHUNK, (c1 - c0 + 1) * CHUNK, ctypes.byref(new)), "rc_vmm_map")
        self.have[c0:c1 + 1] = True
        self.mapped_bytes += int(new.value)
        return int(new.value)

    def close(self):
        """Gives memory and address range back.  The caller has synchronised with every kernel that used the array and holds no
        tensor of it any more."""
        if self.ptr:
            _hip.check(self.lib.rc_vmm_release(self.ptr), "rc_vmm_release")
            self.ptr = 0

    def __del__(self):
        try:
            if self.ptr:
                torch.cuda.synchronize()
                self.close()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass
