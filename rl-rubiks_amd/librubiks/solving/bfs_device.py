"""
Device-resident breadth-first search driving the rc_bfs_* kernels (csrc/rubiks_bfs.hip).

The reference's BFS (librubiks/solving/agents.py:92-131) pops one state at a time from a Python
deque.  Here a whole level is expanded per launch sequence, in chunks of at most `chunk` parents;
the bookkeeping that makes `len(agent)` and the action queue identical to the FIFO loop (discovery
order numbering, first-occurrence dedup, first solved row, the max_states test before each pop) is
done by the kernels -- see the header of rubiks_bfs.hip.  The host reads five words per chunk.

HBM per search: 21 B per node slot + 8 B per hash slot + 28 B per child row of a chunk; the
reference's config #1 (`--max_states 10000000`) takes ~1.2 GB.
"""
import ctypes
from ctypes import POINTER, Structure, c_size_t, c_uint32, c_void_p

import numpy as np
import torch

from librubiks import _hip

N_ACT = 12
NONE = 2 ** 64 - 1
MAX_DEPTH = 64   # longest action queue rc_bfs_path may return (God's number in quarter turns is 26)


class _BfsStruct(Structure):   # mirrors rc_bfs_t (include/rubiks_hip.h)
    _fields_ = [("capacity", c_uint32), ("hash_size", c_uint32), ("chunk", c_uint32), ("reserved", c_uint32)] + \
               [(name, c_void_p) for name in ("keys", "parent", "action", "hash", "child_keys", "child_slot", "flags",
                                              "prefix", "result", "scan_tmp")] + [("scan_tmp_bytes", c_size_t)]


_hip.register({
    "rc_bfs_scan_bytes": [c_uint32],
    "rc_bfs_init": [POINTER(_BfsStruct), c_void_p, c_void_p],
    "rc_bfs_expand": [POINTER(_BfsStruct), c_uint32, c_uint32, c_uint32, c_uint32, c_void_p],
    "rc_bfs_commit": [POINTER(_BfsStruct), c_uint32, c_uint32, c_uint32, c_void_p],
    "rc_bfs_path": [POINTER(_BfsStruct), c_uint32, c_void_p, c_void_p, c_uint32, c_void_p],
}, {"rc_bfs_scan_bytes": c_size_t})


def default_chunk(max_states: int) -> int:
    """Parents per launch sequence: large enough to fill the chip, small next to max_states so that
    little is generated past the cut (any value gives the same result)."""
    return int(min(1 << 20, max(1 << 10, max_states // 8)))


class BFSDevice:
    def __init__(self, max_states: int, chunk: int = None, device=None):
        self.lib = _hip.lib()
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self.max_states = int(max_states)
        self.chunk = int(chunk or default_chunk(self.max_states))
        assert self.max_states >= 1 and 1 <= self.chunk and self.chunk * N_ACT < 2 ** 31
        self.capacity = self.max_states + N_ACT * self.chunk + 1
        assert self.capacity < 2 ** 30, "max_states too large for 32-bit hash slots"
        self.hash_size = 1 << int(np.ceil(np.log2(2 * self.capacity)))
        rows = N_ACT * self.chunk
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)   # noqa: E731
        self.keys = e((self.capacity, 4), torch.int32)
        self.parent = e((self.capacity,), torch.int32)
        self.action = e((self.capacity,), torch.uint8)
        self.hash = e((self.hash_size,), torch.int32)
        self.child_keys = e((rows, 4), torch.int32)
        self.child_slot = e((rows,), torch.int32)
        self.flags = e((rows,), torch.int32)
        self.prefix = e((rows,), torch.int32)
        self.result = torch.zeros((5,), dtype=torch.int64, device=dev)
        scan_bytes = int(self.lib.rc_bfs_scan_bytes(self.chunk))
        if scan_bytes == 0:
            raise _hip.RubiksHipError("rc_bfs_scan_bytes failed")
        self.scan_tmp = e((scan_bytes,), torch.uint8)
        self._path = e((MAX_DEPTH,), torch.uint8)
        self._path_len = e((1,), torch.int32)
        self._root = e((20,), torch.int8)
        s = _BfsStruct()
        s.capacity, s.hash_size, s.chunk, s.reserved = self.capacity, self.hash_size, self.chunk, 0
        for name in ("keys", "parent", "action", "hash", "child_keys", "child_slot", "flags", "prefix", "result", "scan_tmp"):
            setattr(s, name, getattr(self, name).data_ptr())
        s.scan_tmp_bytes = scan_bytes
        self.struct = s
        self.n_nodes = 0      # len(agent) of the last search
        self.committed = 0    # nodes stored in keys / parent / action
        self.levels = 0

    def _result(self):
        r = self.result.cpu().numpy().view(np.uint64)   # the host sync of a chunk
        return [int(x) for x in r]

    def search(self, state: np.ndarray, max_states: int, deadline=None):
        """-> (solved, action list, states seen).  `deadline()` -> True stops between chunks (time limit)."""
        assert max_states <= self.max_states
        lib, m, st = self.lib, ctypes.byref(self.struct), _hip.stream_ptr()
        self._root.copy_(torch.from_numpy(np.ascontiguousarray(state, dtype=np.int8)))
        _hip.check(lib.rc_bfs_init(m, self._root.data_ptr(), st), "rc_bfs_init")
        self.levels, self.committed = 0, 1
        if self._result()[0] == 0:        # agents.py:100: solved start state, nothing stored
            self.n_nodes = 0
            return True, [], 0
        n_nodes, lo, hi = 1, 0, 1         # frontier = nodes lo .. hi-1
        while True:
            level_end = hi
            self.levels += 1
            while lo < level_end:
                if n_nodes >= max_states or (deadline is not None and deadline()):   # agents.py:105
                    self.n_nodes = n_nodes
                    return False, [], n_nodes
                n_par = min(self.chunk, level_end - lo)
                _hip.check(lib.rc_bfs_expand(m, lo, n_par, n_nodes, max_states, st), "rc_bfs_expand")
                first_solved, n_new, cut, new_before_solved, new_before_cut = self._result()
                if first_solved != NONE and first_solved < N_ACT * cut:
                    parent_node, last = lo + first_solved // N_ACT, first_solved % N_ACT
                    _hip.check(lib.rc_bfs_path(m, parent_node, self._path.data_ptr(), self._path_len.data_ptr(),
                                               MAX_DEPTH, st), "rc_bfs_path")
                    n = int(self._path_len.item())
                    if n < 0:
                        raise _hip.RubiksHipError("rc_bfs_path: parent chain longer than MAX_DEPTH")
                    self.n_nodes = n_nodes + new_before_solved
                    return True, self._path[:n].cpu().numpy()[::-1].tolist() + [last], self.n_nodes
                if cut < n_par:
                    self.n_nodes = n_nodes + new_before_cut
                    return False, [], self.n_nodes
                _hip.check(lib.rc_bfs_commit(m, lo, n_par, n_nodes, st), "rc_bfs_commit")
                n_nodes += n_new
                self.committed = n_nodes
                lo += n_par
            hi = n_nodes
            if hi == lo:   # nothing new in a whole level (the reference would pop from an empty deque)
                self.n_nodes = n_nodes
                return False, [], n_nodes

    def node_arrays(self, n: int = None) -> dict:
        """Host copies of the first n committed nodes (for tests)."""
        from librubiks.solving.mcts_device import unpack_keys
        n = self.committed if n is None else n
        return {"states": unpack_keys(self.keys[:n].cpu().numpy().view(np.uint32)),
                "parent": self.parent[:n].cpu().numpy(), "action": self.action[:n].cpu().numpy()}
