"""
DeviceCubes: a batch of cube states resident in HBM as structure-of-arrays.

Plane j (0..19) of cube i is byte soa[j, i]; planes are `stride` bytes apart (a multiple of 16, and
256-byte aligned allocations from torch) so every kernel streams them with 16-byte lane accesses.
The reference keeps (n,20) row-major NumPy arrays on the host (librubiks/cube/cube.py:58-65); the
transposes at the API boundary are rc_aos_to_soa / rc_soa_to_aos.
"""
import numpy as np
import torch

from librubiks import _hip

N_PLANES = 20
OH_WIDTH = 480


def _stride_for(n: int) -> int:
    return max(256, (n + 255) // 256 * 256)


class DeviceCubes:
    __slots__ = ("soa", "n")

    def __init__(self, soa: torch.Tensor, n: int):
        assert soa.dtype == torch.int8 and soa.dim() == 2 and soa.shape[0] == N_PLANES and soa.is_contiguous()
        assert soa.shape[1] % 16 == 0 and soa.shape[1] >= (n + 15) // 16 * 16
        self.soa, self.n = soa, n

    # ---- construction -------------------------------------------------------------------------
    @property
    def stride(self) -> int:
        return self.soa.shape[1]

    def __len__(self):
        return self.n

    @classmethod
    def empty(cls, n: int, device=None) -> "DeviceCubes":
        _hip.lib()
        device = device or torch.device("cuda", torch.cuda.current_device())
        return cls(torch.zeros((N_PLANES, _stride_for(n)), dtype=torch.int8, device=device), n)

    @classmethod
    def from_numpy(cls, states: np.ndarray) -> "DeviceCubes":
        """(n,20) int8 row-major host array -> SoA in HBM."""
        _hip.lib()
        states = np.ascontiguousarray(states, dtype=np.int8)
        assert states.ndim == 2 and states.shape[1] == N_PLANES, f"expected (n,20) states, got {states.shape}"
        return cls.from_aos(torch.from_numpy(states).cuda())

    @classmethod
    def from_aos(cls, aos: torch.Tensor) -> "DeviceCubes":
        lib = _hip.lib()
        n = aos.shape[0]
        out = cls.empty(n, aos.device)
        if n:
            _hip.check(lib.rc_aos_to_soa(aos.data_ptr(), out.soa.data_ptr(), n, out.stride, _hip.stream_ptr()),
                       "rc_aos_to_soa")
        return out

    @classmethod
    def solved(cls, n: int) -> "DeviceCubes":
        lib = _hip.lib()
        buf = np.empty(N_PLANES, dtype=np.int8)
        _hip.check(lib.rc_get_solved(buf.ctypes.data), "rc_get_solved")
        out = cls.empty(n)
        out.soa[:] = torch.from_numpy(buf).cuda()[:, None]
        return out

    # ---- back to the host ---------------------------------------------------------------------
    def to_aos(self) -> torch.Tensor:
        lib = _hip.lib()
        aos = torch.empty((self.n, N_PLANES), dtype=torch.int8, device=self.soa.device)
        if self.n:
            _hip.check(lib.rc_soa_to_aos(self.soa.data_ptr(), aos.data_ptr(), self.n, self.stride, _hip.stream_ptr()),
                       "rc_soa_to_aos")
        return aos

    def numpy(self) -> np.ndarray:
        return self.to_aos().cpu().numpy()

    # ---- environment ops (all out of place, like the reference's functional API) --------------
    def multi_rotate(self, actions: torch.Tensor, out: "DeviceCubes" = None) -> "DeviceCubes":
        """actions: uint8 device tensor of action indices 0..11, length >= n (padded to 16)."""
        lib = _hip.lib()
        assert actions.dtype == torch.uint8 and actions.is_cuda and actions.numel() >= (self.n + 15) // 16 * 16
        out = out or DeviceCubes.empty(self.n, self.soa.device)
        _hip.check(lib.rc_multi_rotate(self.soa.data_ptr(), actions.data_ptr(), out.soa.data_ptr(), self.n,
                                       self.stride, out.stride, _hip.stream_ptr()), "rc_multi_rotate")
        return out

    def expand12(self, out: "DeviceCubes" = None) -> "DeviceCubes":
        """All 12 children, child k of parent p at column 12p+k."""
        lib = _hip.lib()
        out = out or DeviceCubes.empty(12 * self.n, self.soa.device)
        assert out.n == 12 * self.n
        _hip.check(lib.rc_expand12(self.soa.data_ptr(), out.soa.data_ptr(), self.n, self.stride, out.stride,
                                   _hip.stream_ptr()), "rc_expand12")
        return out

    def expand12_flags(self, out: "DeviceCubes" = None):
        """(children, parents solved bool[n], children solved bool[12 n]) in ONE launch: `expand12()`, `is_solved()` and
        `expand12().is_solved()` as a data-generation step of an ADI rollout asks for them (reference train.py:285-296)."""
        lib = _hip.lib()
        out = out or DeviceCubes.empty(12 * self.n, self.soa.device)
        assert out.n == 12 * self.n
        pad = (self.n + 15) // 16 * 16
        pflags = torch.empty(pad, dtype=torch.uint8, device=self.soa.device)
        cflags = torch.empty(12 * pad, dtype=torch.uint8, device=self.soa.device)
        _hip.check(lib.rc_expand12_flags(self.soa.data_ptr(), out.soa.data_ptr(), self.n, self.stride, out.stride, pflags.data_ptr(),
                                         cflags.data_ptr(), _hip.stream_ptr()), "rc_expand12_flags")
        return out, pflags[:self.n].view(torch.bool), cflags[:12 * self.n].view(torch.bool)

    def is_solved(self) -> torch.Tensor:
        """bool[n] device tensor."""
        lib = _hip.lib()
        flags = torch.empty((self.n + 15) // 16 * 16, dtype=torch.uint8, device=self.soa.device)
        _hip.check(lib.rc_is_solved(self.soa.data_ptr(), flags.data_ptr(), None, None, self.n, self.stride,
                                    _hip.stream_ptr()), "rc_is_solved")
        return flags[:self.n].view(torch.bool)

    def solved_mask(self):
        """(uint64-word bit mask as int64 tensor, count) -- bit i%64 of word i//64 set iff cube i is solved."""
        lib = _hip.lib()
        words = (self.n + 63) // 64
        mask = torch.zeros(words + 1, dtype=torch.int64, device=self.soa.device)
        count = torch.zeros(1, dtype=torch.int32, device=self.soa.device)
        _hip.check(lib.rc_is_solved(self.soa.data_ptr(), None, mask.data_ptr(), count.data_ptr(), self.n, self.stride,
                                    _hip.stream_ptr()), "rc_is_solved")
        return mask[:words], count

    def as_oh(self, dtype=torch.float32, out: torch.Tensor = None) -> torch.Tensor:
        """(n,480) one-hot on the device, float32 (the reference's dtype) or bfloat16."""
        lib = _hip.lib()
        if out is None:
            out = torch.empty((self.n, OH_WIDTH), dtype=dtype, device=self.soa.device)
        assert out.is_contiguous() and out.shape == (self.n, OH_WIDTH)
        fn = {torch.float32: lib.rc_as_oh_f32, torch.bfloat16: lib.rc_as_oh_bf16}[out.dtype]
        _hip.check(fn(self.soa.data_ptr(), out.data_ptr(), self.n, self.stride, _hip.stream_ptr()), "rc_as_oh")
        return out

    def apply_moves(self, moves: torch.Tensor) -> "DeviceCubes":
        """In place.  moves: uint8 (depth, stride) device tensor of action indices; row d is applied d-th."""
        lib = _hip.lib()
        assert moves.dtype == torch.uint8 and moves.is_cuda and moves.is_contiguous() and moves.shape[1] == self.stride
        _hip.check(lib.rc_apply_moves(self.soa.data_ptr(), moves.data_ptr(), self.stride, self.stride, moves.shape[0],
                                      _hip.stream_ptr()), "rc_apply_moves")
        return self
