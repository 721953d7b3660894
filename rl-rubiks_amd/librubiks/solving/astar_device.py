"""
Device-resident batch of weighted-A* problems and the iteration that drives the rc_astar_* kernels.
The host sequences launches, runs the value network on the compacted new states of ALL problems
and reads back one scalar per iteration (the total number of new states = the GEMM's row count);
no cube arithmetic happens on the host.

Per node ~50 B (packed state 16, G 4, parent 4, action 1, election scratch 4, 2 hash slots x 4,
open-list entry 12): BASELINE config #3 (4 096 problems) at the reference's default
max_states = 175 000 is ~36 GB of the 288 GB HBM3E.
"""
import ctypes
from ctypes import POINTER, Structure, c_double, c_size_t, c_uint32, c_void_p

import numpy as np
import torch

from librubiks import _hip
from librubiks.cube.device import DeviceCubes
from librubiks.model import make_inference_net, net_fingerprint
from librubiks.solving.mcts_device import unpack_keys

RUNNING, SOLVED, EXHAUSTED, OPEN_EMPTY, ROOT_SOLVED = 0, 1, 2, 3, 4
N_ACT = 12
INT_MAX = 2 ** 31 - 1


class _AsStruct(Structure):   # mirrors rc_astar_t (include/rubiks_hip.h)
    _fields_ = [("n_problems", c_uint32), ("capacity", c_uint32), ("hash_size", c_uint32), ("expansions", c_uint32)] + \
               [(name, c_void_p) for name in ("keys", "G", "parents", "parent_actions", "claim", "hash", "heap_cost",
                                              "heap_idx", "heap_size", "n_nodes", "status", "solved_idx", "iterations",
                                              "n_popped", "new_count", "popped", "child_keys", "child_node", "row_tmp",
                                              "row_flags")]


_hip.register({
    "rc_astar_init": [POINTER(_AsStruct), c_void_p, c_size_t, c_void_p],
    "rc_astar_pop_expand": [POINTER(_AsStruct), c_uint32, c_void_p],
    "rc_astar_gather_new": [POINTER(_AsStruct), c_void_p, c_void_p, c_size_t, c_void_p],
    "rc_astar_push_relax": [POINTER(_AsStruct), c_void_p, c_void_p, c_double, c_void_p],
})

NET_CHUNK = 1 << 19   # rows per network call (bounds the one-hot buffer to ~0.5 GB in bf16)


class AStarBatch:
    def __init__(self, n_problems: int, capacity: int, expansions: int, device=None):
        self.lib = _hip.lib()
        dev = device or torch.device("cuda", torch.cuda.current_device())
        B, C, N = int(n_problems), int(capacity), int(expansions)
        assert B > 0 and N > 0 and C >= 12 * N + 1
        self.B, self.C, self.N, self.device = B, C, N, dev
        self.hash_size = 1 << int(np.ceil(np.log2(2 * (C + 1))))
        from librubiks._vmm import zeros_or_trim
        z = lambda shape, dt: zeros_or_trim(shape, dt, dev)   # noqa: E731  (parked MCTS node stores are given back if HBM runs out)
        rows = B * (C + 1)
        self.keys = z((rows, 4), torch.int32)
        self.G = z((rows,), torch.int32)
        self.parents = z((rows,), torch.int32)
        self.parent_actions = z((rows,), torch.uint8)
        self.claim = zeros_or_trim((rows,), torch.int32, dev, fill=INT_MAX)
        self.hash = z((B, self.hash_size), torch.int32)
        self.heap_cost = z((rows,), torch.float64)
        self.heap_idx = z((rows,), torch.int32)
        for name in ("heap_size", "n_nodes", "status", "solved_idx", "iterations", "n_popped", "new_count"):
            setattr(self, name, z((B,), torch.int32))
        self.popped = z((B, N), torch.int32)
        self.child_keys = z((B * N * N_ACT, 4), torch.int32)
        self.child_node = z((B, N * N_ACT), torch.int32)
        self.row_tmp = z((B, N * N_ACT), torch.int32)
        self.row_flags = z((B, N * N_ACT), torch.uint8)
        self.new_offset = z((B + 1,), torch.int32)
        self.new_states = DeviceCubes.empty(B * N * N_ACT, dev)     # compacted network input (worst case size)
        self.values = z((B * N * N_ACT,), torch.float32)
        s = _AsStruct()
        s.n_problems, s.capacity, s.hash_size, s.expansions = B, C, self.hash_size, N
        for name, _ in _AsStruct._fields_[4:]:
            setattr(s, name, getattr(self, name).data_ptr())
        self.struct = s
        self.engine = None
        self._net_fp = None
        self._oh = None

    def set_net(self, net, dtype=torch.bfloat16):
        """Builds the inference engine for `net`; a no-op when the batch already runs exactly these weights."""
        fp = net_fingerprint(net, dtype)
        if self.engine is not None and fp == self._net_fp:
            return
        self._net_fp = fp
        self.engine = make_inference_net(net, dtype)
        rows = min(NET_CHUNK, self.B * self.N * N_ACT)
        if getattr(self.engine, "supports_cubes", False):
            self._oh = None
            self._x1 = self.engine.workspace(rows)
        else:
            self._oh = torch.empty((rows, 480), dtype=self.engine.input_dtype, device=self.device)

    def reset(self, roots: DeviceCubes):
        assert roots.n == self.B and self.engine is not None
        self.hash.zero_()
        self.claim.fill_(INT_MAX)
        _hip.check(self.lib.rc_astar_init(ctypes.byref(self.struct), roots.soa.data_ptr(), roots.stride, _hip.stream_ptr()),
                   "rc_astar_init")

    def _values_of_new(self, total: int):
        """Value head on the `total` compacted new states, chunked through the one-hot buffer."""
        lib, st = self.lib, _hip.stream_ptr()
        soa = self.new_states.soa
        if getattr(self.engine, "supports_cubes", False):   # input layer fused with the one-hot encoding: no (n, 480) matrix
            for lo in range(0, total, NET_CHUNK):
                n = min(NET_CHUNK, total - lo)
                self.values[lo:lo + n] = self.engine.value_cubes(self.new_states, None if self._x1 is None else self._x1[:n], lo, n)
            return
        for lo in range(0, total, NET_CHUNK):
            n = min(NET_CHUNK, total - lo)
            oh = self._oh[:n]
            fn = lib.rc_as_oh_bf16 if oh.dtype == torch.bfloat16 else lib.rc_as_oh_f32
            # column offset lo is a multiple of 16, so the shifted plane pointer stays 16-byte aligned
            _hip.check(fn(soa.data_ptr() + lo, oh.data_ptr(), n, self.new_states.stride, st), "rc_as_oh")
            self.values[lo:lo + n] = self.engine.value(oh)

    def iteration(self, lambda_: float, max_states: int) -> int:
        """pop N, expand, dedup, evaluate the new states, push, win check, relax.  Returns #new states."""
        m, st = ctypes.byref(self.struct), _hip.stream_ptr()
        _hip.check(self.lib.rc_astar_pop_expand(m, max_states, st), "rc_astar_pop_expand")
        torch.cumsum(self.new_count, 0, dtype=torch.int32, out=self.new_offset[1:])
        total = int(self.new_offset[-1].item())   # the one host sync per iteration: the network's row count
        if total:
            _hip.check(self.lib.rc_astar_gather_new(m, self.new_offset.data_ptr(), self.new_states.soa.data_ptr(),
                                                    self.new_states.stride, st), "rc_astar_gather_new")
            self._values_of_new(total)
        _hip.check(self.lib.rc_astar_push_relax(m, self.new_offset.data_ptr(), self.values.data_ptr(), float(lambda_), st),
                   "rc_astar_push_relax")
        return total

    def any_running(self) -> bool:
        return bool((self.status == RUNNING).any().item())

    def problem_arrays(self, b: int) -> dict:
        """Host copies of problem b's arrays, shaped like the reference agent's attributes."""
        n = int(self.n_nodes[b].item())
        lo, hi = b * (self.C + 1), b * (self.C + 1) + n + 1
        hs = int(self.heap_size[b].item())
        return {
            "n": n,
            "states": unpack_keys(self.keys[lo:hi].cpu().numpy().view(np.uint32)),
            "G": self.G[lo:hi].cpu().numpy().astype(np.float64),
            "parents": self.parents[lo:hi].cpu().numpy().astype(np.int64),
            "parent_actions": self.parent_actions[lo:hi].cpu().numpy().astype(np.int64),
            "open_queue": list(zip(self.heap_cost[lo:lo + hs].cpu().numpy().tolist(),
                                   self.heap_idx[lo:lo + hs].cpu().numpy().tolist())),
        }
