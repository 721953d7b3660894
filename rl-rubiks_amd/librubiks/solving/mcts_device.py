"""
Device-resident forest of MCTS trees (one per scramble) and the lock-step iteration that drives
the rc_mcts_* kernels of librubiks_hip.so.  Host code here only allocates HBM, sequences launches
on torch's current stream and reads results back; no cube arithmetic happens on the host.

Per node (tree-major, 1-based, row 0 = "no neighbour" sentinel like the reference's arrays,
librubiks/solving/agents.py:417-459):  packed state 16 B, neighbors 12 x i32, P / W 12 x f32,
N 12 x i32, virtual-loss count 12 x u16, V f32, leaf u8, a 16-byte walk record (what a PUCT descent does at the
node, see csrc/rubiks_mcts.hip) plus 2 hash slots x i32  ->  ~285 B per node.
Large forests reserve ADDRESS SPACE for capacity + 1 rows per tree and map memory behind the rows as the trees grow
(librubiks/_vmm.py, `MCTSForest.grow`): the reference's default max_states = 175 000 costs what the trees reach, not
1 024 x 175 000 x 285 B = 51 GB, and 8 192 trees at that cap (408 GB of rows) fit the 288 GB of HBM3E.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_int, c_size_t, c_uint32, c_void_p

import numpy as np
import torch

from librubiks import _hip
from librubiks._vmm import VmmArray, zeros_or_trim
from librubiks.cube.device import DeviceCubes
from librubiks.model import make_inference_net, net_fingerprint

RUNNING, SOLVED, EXHAUSTED, PATH_OVERFLOW, ROOT_SOLVED, CORRUPT = 0, 1, 2, 3, 4, 5
N_ACT = 12


class _McStruct(Structure):   # mirrors rc_mcts_t (include/rubiks_hip.h)
    _fields_ = [("n_trees", c_uint32), ("capacity", c_uint32), ("hash_size", c_uint32), ("max_path", c_uint32),
                ("rows_per_tree", c_uint32), ("node_words", c_uint32)] + \
               [(name, c_void_p) for name in ("keys", "nbr", "P", "W", "N", "V", "leaf", "hash", "n_nodes",
                                              "status", "solved_idx", "solved_action", "iterations", "path_len", "pending",
                                              "path_node", "path_act", "child_soa")] + \
               [("child_stride", c_size_t)] + \
               [(name, c_void_p) for name in ("child_idx", "new_mask", "expanded", "select_stats", "bfs", "short_act",
                                              "short_len", "rec")] + \
               [("ring_k", c_uint32)] + [(name, c_void_p) for name in ("ring_node", "ring_act", "ring_len", "phase", "active")] + \
               [("n_active", c_uint32), ("unc_list_cap", c_uint32), ("mapped_rows", c_void_p)] + \
               [("path_block_log2", c_uint32), ("lds_levels", c_uint32), ("ring_levels", c_uint32), ("path_next", c_void_p),
                ("path_rows", c_void_p)]


VmmArray.before_trim.append(lambda: MCTSForest.drain_deferred())   # (defined below; resolved when called)

_hip.register({
    "rc_mcts_struct_bytes": [],
    "rc_mcts_plant": [POINTER(_McStruct), c_void_p, c_uint32, c_void_p, c_size_t, c_size_t, c_void_p],
    "rc_mcts_expand": [POINTER(_McStruct), c_uint32, c_void_p],
    "rc_mcts_backup": [POINTER(_McStruct), c_void_p, c_void_p, c_void_p],
    "rc_mcts_backup_head": [POINTER(_McStruct), c_void_p, c_size_t, c_int, c_void_p],
    "rc_mcts_select": [POINTER(_McStruct), c_double, c_uint32, c_void_p],
    "rc_mcts_backup_select": [POINTER(_McStruct), c_void_p, c_void_p, c_double, c_uint32, c_void_p],
    "rc_mcts_backup_select_head": [POINTER(_McStruct), c_void_p, c_size_t, c_int, c_double, c_uint32, c_void_p],
    "rc_mcts_plant_expanded": [POINTER(_McStruct), c_void_p, c_uint32, c_void_p, c_size_t, c_size_t, c_uint32, c_void_p],
    "rc_mcts_step": [POINTER(_McStruct), c_void_p, c_void_p, c_double, c_uint32, c_uint32, c_void_p],
    "rc_mcts_step_head": [POINTER(_McStruct), c_void_p, c_size_t, c_int, c_double, c_uint32, c_uint32, c_void_p],
    "rc_mcts_complete_graph": [POINTER(_McStruct), c_void_p],
    "rc_mcts_shorten": [POINTER(_McStruct), c_void_p],
    "rc_mcts_copy_trees": [POINTER(_McStruct), POINTER(_McStruct), c_void_p, c_uint32, c_uint32, c_void_p],
}, restypes={"rc_mcts_struct_bytes": ctypes.c_size_t})


def unpack_keys(keys: np.ndarray) -> np.ndarray:
    """(n,4) uint32 packed states -> (n,20) int8 codes (6 codes of 5 bits per dword)."""
    keys = keys.astype(np.uint32).reshape(-1, 4)
    out = np.empty((len(keys), 20), dtype=np.int8)
    for j in range(20):
        out[:, j] = (keys[:, j // 6] >> np.uint32(5 * (j % 6))) & np.uint32(31)
    return out


_PER_NODE = ("keys", "node", "V", "leaf")
_PER_TREE = ("n_nodes", "status", "solved_idx", "solved_action", "iterations", "path_len", "pending", "ring_node", "ring_act", "ring_len",
             "phase")
_PATH_ARRAYS = {"path_node": torch.int32, "path_act": torch.uint8, "path_next": torch.int32, "short_act": torch.uint8}
_RESULT_NODE = ("keys", "nbr", "leaf")          # what rc_mcts_complete_graph / rc_mcts_shorten read of a tree (65 B per node)
_RESULT_TREE = ("n_nodes", "status", "solved_idx", "solved_action", "iterations", "path_len", "pending", "phase")
RING_K = 32   # descent paths kept per tree for line following (rc_mcts_t::ring_k)
# The descent path of a tree has no length limit in the reference (agents.py:575-595).  The path arrays are blocked
# (rc_mcts_t: [blocks][B][PATH_BLOCK]); block 0 always has memory, deeper blocks are address space that gets memory for the
# trees that go that deep (`MCTSForest.ensure_path`).  MAX_PATH_LEVELS is the address space a tree's path can grow into.
# RUBIKS_PATH_BLOCK / RUBIKS_LDS_LEVELS / RUBIKS_RING_LEVELS shrink the block, the levels rc_mcts_select works on in LDS and the
# levels kept per ring line: tests drive the deep-path code with the reference's recorded (shallow) trees that way; results
# do not depend on any of them.
PATH_BLOCK = int(os.environ.get("RUBIKS_PATH_BLOCK", "4096"))
LDS_LEVELS = int(os.environ.get("RUBIKS_LDS_LEVELS", "4096"))
RING_LEVELS = int(os.environ.get("RUBIKS_RING_LEVELS", "4096"))
MAX_PATH_LEVELS = 1 << 20
ROWS = 11    # network rows per tree and iteration (rc_mcts_t::rows_per_tree)
MIN_RUNG = 32     # smallest launch size a running forest is narrowed to (32 trees = 352 network rows = one GEMM row tile)
NODE_WORDS = 64   # 32-bit words per node record (RC_MCTS_NODE_WORDS): line 0 = N | W | walk record, line 1 = P | nbr
_NODE_FIELDS = {"N": (0, 12, torch.int32), "W": (12, 24, torch.float32), "rec": (24, 28, torch.int32),
                "P": (32, 44, torch.float32), "nbr": (44, 56, torch.int32)}


MAX_CAPACITY = (1 << 24) - 2   # nodes per tree: capacity + 1 rows of 256 bytes, addressed by 32-bit byte offsets (rc_mcts_t, tree_bufs)

RUNG_RATIO = 0.95   # measured on configs[1] to completion, same box: 0.8 1.772 s, 0.9 1.735, 0.93 1.715, 0.95 1.710, 0.97 1.703 (profiles/r3_rung_ratio_ab.txt)


def rungs(n_trees: int) -> list:
    """Launch sizes a forest of n_trees is narrowed to as its trees finish (`MCTSForest.set_active`), largest first: n_trees,
    then multiples of 32 trees (352 network rows, the row tile of the layer kernels) each at most RUNG_RATIO of the one before,
    down to MIN_RUNG (26 sizes for 1 024 trees).  One HIP graph is captured per size and kept for the forest's lifetime."""
    out = [int(n_trees)]
    while out[-1] > MIN_RUNG:
        nxt = max(MIN_RUNG, int(out[-1] * RUNG_RATIO) // 32 * 32)
        if nxt >= out[-1]:
            break
        out.append(nxt)
    return out


def growth_plan(have: np.ndarray, seen: np.ndarray, steps_ahead: int, max_rows: int, pregrow: bool, factor: float = 2.0,
                step: int = 32768):
    """
    Which trees get more rows, and how many (host logic of the on-demand node store, `MCTSForest.grow`).
    have[t]: rows of tree t with memory behind them; seen[t]: its node count at the host's last look; up to `steps_ahead`
    iterations (12 new nodes each at most) run before the next look.  Nothing happens until some tree could reach the end of its
    rows in that time; then that tree -- and, if `pregrow`, every tree past 70 % of its rows: a map call drains the GPU, so growth
    steps should be few -- goes to max(what it can reach, min(factor x its rows, its rows + step)), never beyond max_rows.
    Returns (tree indices, rows wanted).
    """
    have = np.asarray(have, dtype=np.int64)
    seen = np.asarray(seen, dtype=np.int64)
    need = np.minimum(max_rows, seen + N_ACT * (steps_ahead + 1) + 2)
    must = need > have
    if not must.any():
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    soon = (must | (seen > 0.7 * have)) if pregrow else must
    trees = np.flatnonzero(soon & (have < max_rows))
    grown = np.minimum((have[trees] * factor).astype(np.int64), have[trees] + step)
    return trees, np.minimum(max_rows, np.maximum(need[trees], grown))


def path_growth_plan(have: np.ndarray, seen: np.ndarray, block: int, max_path: int):
    """
    Which trees get more path blocks, and up to which level (host logic of the on-demand path store, `MCTSForest.grow_paths`).
    have[t]: levels of tree t's path arrays with memory behind them (whole blocks); seen[t]: its path length at the host's last look.
    A tree whose descents have come within a quarter block of the end of its blocks gets the next block -- or as many as one and a half
    times its path needs (deep trees get deeper: growth steps should be few) --, never beyond max_path.  Returns (tree indices, levels).
    """
    have = np.asarray(have, dtype=np.int64)
    seen = np.asarray(seen, dtype=np.int64)
    near = np.flatnonzero((seen + block // 4 >= have) & (have < max_path))
    want = np.maximum(have[near] + block, seen[near] + seen[near] // 2)
    return near, np.minimum(max_path, (want + block - 1) // block * block)


class MCTSForest:
    # node records of at least this many bytes: the per-node arrays are mapped on demand (None: never).  RUBIKS_VMM_MIN_GB
    # overrides it for a process (0 = every forest, "never" = none): A/B runs and diagnosis.
    VMM_MIN_BYTES = (lambda v: 1 << 30 if v is None else None if v == "never" else int(float(v) * (1 << 30)))(os.environ.get("RUBIKS_VMM_MIN_GB"))
    # Rows a planted tree starts with (a depth-20 tree ends at 12-14 k nodes on average and never grows at all), and how its
    # mapping grows when it gets near them: by GROW_FACTOR, at most GROW_STEP rows at a time.  A map call returns only when the
    # GPU has drained, so in forests whose iterations are short (<= PREGROW_TREES trees) every tree past 70 % of its rows grows
    # in the same step as the one that has to; in large forests an iteration takes milliseconds, a drained queue is nothing
    # against that, and only the trees that have to grow do (8 192 trees x one chunk too many would be 67 GB).
    GROW_ROWS = 16384          # (forests of <= PREGROW_TREES trees: twice that -- 8 GB for 1 024 trees -- so that few trees ever grow)
    GROW_FACTOR = 2.0
    GROW_STEP = 32768
    PREGROW_TREES = 2048
    # Chunks: 2 MiB (8 192 node records).  A map call costs ~15 us per 1 000 chunks the process has mapped already, so forests
    # whose node records reserve this many bytes and more take 4 MiB chunks for the records and 8 MiB for the keys.
    BIG_CHUNKS_FROM = 96 << 30

    @classmethod
    def on_demand_pays(cls, n_trees: int, capacity: int) -> bool:
        """Whether a forest of this shape is mapped on demand.  Memory arrives a chunk (2 MiB = 8 192 node records) at a time and
        every tree's records start on a chunk boundary, a planted tree starts with `_first_rows` rows, and a map call costs more
        the more chunks are mapped: trees whose whole capacity is not at least twice their first rows gain nothing (8 192 trees x
        10 000 nodes: 23 GB allocated up front, 36 GB mapped at the first plant; 65 536 trees x 200 nodes: 3.7 GB against one chunk
        per tree = 137 GB and half a minute of map calls) -- those, and forests under VMM_MIN_BYTES, are allocated up front.
        VMM_MIN_BYTES = 0 (tests, RUBIKS_VMM_MIN_GB=0) means every forest."""
        if cls.VMM_MIN_BYTES is None:
            return False
        if cls.VMM_MIN_BYTES == 0:
            return True
        first = cls.GROW_ROWS * (2 if n_trees <= cls.PREGROW_TREES else 1)
        return n_trees * (capacity + 1) * NODE_WORDS * 4 >= cls.VMM_MIN_BYTES and capacity + 1 >= 2 * first

    def __init__(self, n_trees: int, capacity: int, max_path: int = None, device=None, _results_only: bool = False, vmm: bool = None,
                 path_block: int = None, lds_levels: int = None, ring_levels: int = None):
        """vmm: per-node arrays as reserved address ranges with memory mapped behind the rows in use (`grow`); None = by size.
        max_path: None (the default) = descents of any length, as in the reference (agents.py:575-595): the path arrays reserve
        address space for MAX_PATH_LEVELS levels per tree and get memory a block at a time for the trees that go that deep
        (`ensure_path`).  A number = a fixed path store of that many levels (rounded up to whole blocks), allocated up front; a
        descent that fills it ends its tree with status PATH_OVERFLOW.  path_block / lds_levels / ring_levels: see PATH_BLOCK."""
        self.lib = _hip.lib()
        MCTSForest.drain_deferred()
        if self.lib.rc_mcts_struct_bytes() != ctypes.sizeof(_McStruct):
            raise _hip.RubiksHipError(f"rc_mcts_t is {self.lib.rc_mcts_struct_bytes()} bytes in librubiks_hip.so but {ctypes.sizeof(_McStruct)} "
                                      "here: rebuild the library (make -C rl-rubiks_amd)")
        dev = device or torch.device("cuda", torch.cuda.current_device())
        B, C = int(n_trees), int(capacity)
        self.C_asked = C            # C below may be rounded up to whole chunks per tree
        assert B > 0 and 13 <= C <= MAX_CAPACITY, f"capacity {C}: 13 .. {MAX_CAPACITY} (32-bit byte offsets inside a tree's node records)"
        pow2 = lambda x: 1 << max(1, int(np.ceil(np.log2(max(2, int(x))))))   # noqa: E731
        pb = int(path_block or PATH_BLOCK)
        assert pb == pow2(pb) and 2 <= pb <= 1 << 24, "path blocks are powers of two"
        self.path_vmm = max_path is None and self.VMM_MIN_BYTES is not None
        if max_path is None:   # address space for a 64 GB path_node array at most, whatever the number of trees
            levels = max(pb, min(MAX_PATH_LEVELS, (16 << 30) // B) // pb * pb) if self.path_vmm else pb
        else:
            assert max_path >= 2
            pb = min(pb, pow2(max_path))
            levels = (int(max_path) + pb - 1) // pb * pb
        self.path_block, self.path_blocks = pb, levels // pb
        self.lds_levels = max(1, min(4096, int(lds_levels or LDS_LEVELS)))
        self.ring_levels = max(1, min(4096, levels, int(ring_levels or RING_LEVELS)))
        max_path = levels
        if vmm is None:
            vmm = self.on_demand_pays(B, C)
        self.vmm = bool(vmm)
        if self.vmm and not _results_only:
            # every tree's node records start on a chunk boundary (rows per tree rounded up to whole chunks: address space, not
            # memory), so the first rows of a tree cost one chunk, not the two a straddling range would
            per_chunk = self._chunk_for(B * (C + 1), (0, NODE_WORDS), torch.int32) // (NODE_WORDS * 4)
            C = min(MAX_CAPACITY, (C + per_chunk) // per_chunk * per_chunk - 1)
        self.B, self.C, self.max_path, self.device = B, C, max_path, dev
        self.hash_size = 1 << int(np.ceil(np.log2(2 * (C + 1))))
        z = lambda shape, dt: zeros_or_trim(shape, dt, dev)   # noqa: E731  (parked node stores of other shapes are given back if HBM runs out)
        rows = B * (C + 1)
        self.results_only = _results_only
        per_node = {   # [B][capacity + 1] rows each
            "keys": ((rows, 4), torch.int32), "node": ((rows, NODE_WORDS), torch.int32), "nbr": ((rows, N_ACT), torch.int32),
            "V": ((rows,), torch.float32), "leaf": ((rows,), torch.uint8),
        }
        per_tree = {
            "hash": ((B, self.hash_size), torch.int32),
            "n_nodes": ((B,), torch.int32), "status": ((B,), torch.int32), "solved_idx": ((B,), torch.int32),
            "solved_action": ((B,), torch.int32), "iterations": ((B,), torch.int32), "path_len": ((B,), torch.int32),
            "pending": ((B,), torch.int32),
            "ring_node": ((B, RING_K, self.ring_levels), torch.int32), "ring_act": ((B, RING_K, self.ring_levels), torch.uint8),
            "ring_len": ((B, RING_K), torch.int32), "phase": ((B,), torch.int32),
        }
        self._ranges = {}     # name -> (VmmArray, bytes per row) of the arrays mapped on demand
        if self.vmm:
            # the arrays a finished forest of the same shape has left behind are taken over as they are (addresses and memory);
            # a forest of another shape first gives all such memory back
            sizes = [(rows * int(np.prod(shape[1:], dtype=np.int64)) * torch.empty(0, dtype=dt).element_size(), self._chunk_for(rows, shape, dt))
                     for name, (shape, dt) in per_node.items() if name != "nbr"]
            if not all(VmmArray.has_parked(nb, dev, ch) for nb, ch in sizes):
                torch.cuda.synchronize()
                VmmArray.trim()
            try:
                for name, (shape, dt) in per_node.items():
                    if name != ("node" if _results_only else "nbr") and not (_results_only and name not in _RESULT_NODE):
                        bpr = int(np.prod(shape[1:], dtype=np.int64)) * torch.empty(0, dtype=dt).element_size()
                        self._ranges[name] = (VmmArray.take(rows * bpr, dev, chunk=self._chunk_for(rows, shape, dt)), bpr)
            except _hip.RubiksHipError as e:
                # a runtime without HIP virtual memory management (or out of address space): the forest is allocated up front,
                # as in rounds 1-3 -- which fails by itself, loudly, if that does not fit
                import warnings
                warnings.warn(f"MCTSForest: node rows cannot be reserved on demand ({e}); allocating {rows * 285 / 1e9:.0f} GB up front", RuntimeWarning)
                for arr, _ in self._ranges.values():
                    arr.close()
                self._ranges, self.vmm = {}, False
        for name, (shape, dt) in per_node.items():
            if name == ("node" if _results_only else "nbr"):
                continue      # a search forest keeps nbr as a field of the node record, a results-only forest as a plain array
            if _results_only and name not in _RESULT_NODE:
                t = z((1,) + shape[1:], dt)   # arrays that turning finished trees into results never touches: one row, so that pointers are valid
            elif self.vmm:    # no memory yet, and none of these arrays needs clearing: a node's rows are written when it is created
                t = self._ranges[name][0].tensor(dt, shape)
            else:
                t = z(shape, dt)
            setattr(self, name, t)
        for name, (shape, dt) in per_tree.items():
            if _results_only and name not in _RESULT_TREE + ("hash",):
                t = z((1,) + shape[1:], dt)
            else:
                t = z(shape, dt)
            setattr(self, name, t)
        self._make_path_store(z)
        # the reference's per-action node arrays (agents.py:421-427) are strided views of the 256-byte node records: one or two
        # adjacent cache lines per node for the kernels, the same [rows, 12] tensors for everything that inspects a tree
        for name, (lo, hi, dt) in _NODE_FIELDS.items():
            if _results_only:
                if name != "nbr":
                    setattr(self, name, z((1, hi - lo), dt))
            else:
                setattr(self, name, self.node[:, lo:hi].view(dt))
        # rows 0 .. mapped_host[t] - 1 of tree t have memory behind them (every row, if the arrays are ordinary allocations)
        self.mapped_host = np.full(B, 0 if self.vmm else C + 1, dtype=np.int32)
        self.mapped_rows = z((B,), torch.int32) if self.vmm else None
        self.nodes_seen = np.zeros(B, dtype=np.int64)   # node counts as of the last look the host had (`grow`, `MCTSRun`)
        self._steps_covered = 0                         # iterations that may be queued before the mapping has to be looked at again
        # Network rows per tree: only the NEW children of the expanded leaf are evaluated, and a non-root leaf has at most
        # 11 of them (its parent is known): 11 row slots per tree.  A planted root takes two iterations (rc_mcts_t::phase).
        self.children = DeviceCubes.empty(ROWS * B, dev)
        self.child_idx = z((B, N_ACT), torch.int32)
        self.new_mask = z((B,), torch.int32)
        self.expanded = z((B,), torch.uint8)
        self.probs = z((ROWS * B, N_ACT), torch.float32)     # static network outputs (graph capture)
        self.values = z((ROWS * B,), torch.float32)
        s = _McStruct()
        s.n_trees, s.capacity, s.hash_size, s.max_path = B, C, self.hash_size, max_path
        s.rows_per_tree = ROWS
        s.node_words = N_ACT if _results_only else NODE_WORDS
        s.ring_k = RING_K
        for name in ("keys", "nbr", "P", "W", "N", "V", "leaf", "rec", "hash", "n_nodes", "status", "solved_idx",
                     "solved_action", "iterations", "path_len", "pending", "path_node", "path_act", "path_next", "short_act", "child_idx",
                     "new_mask", "expanded", "ring_node", "ring_act", "ring_len", "phase"):
            setattr(s, name, getattr(self, name).data_ptr())
        s.path_block_log2, s.lds_levels, s.ring_levels = int(np.log2(self.path_block)), self.lds_levels, self.ring_levels
        s.path_rows = self.path_rows.data_ptr() if self.path_rows is not None else None
        s.child_soa, s.child_stride = self.children.soa.data_ptr(), self.children.stride
        self.select_stats = z((B, 8), torch.int32)   # diagnostics: where each descent became sequential, its length, ticks
        s.select_stats = self.select_stats.data_ptr()
        self.bfs = None   # BFS scratch of rc_mcts_shorten: allocated on first use (graph search only)
        self.short_len = z((B,), torch.int32)
        s.short_len = self.short_len.data_ptr()
        # The trees the iteration kernels work on (rc_mcts_t::active): position i of the list = workgroup i = network rows
        # 11 i .. 11 i + 10.  Finished trees are dropped from a running batch by writing a shorter list (`set_active`); no tree
        # moves in memory.  The buffer's address is baked into the captured graphs, its content and the launch size G are not.
        self.active_buf = torch.arange(B, dtype=torch.int32, device=dev)
        self.rungs = rungs(B)
        self.G = B
        self._listed = None   # host copy of the list (None = every tree in order)
        s.active, s.n_active = self.active_buf.data_ptr(), B
        s.unc_list_cap = 128
        s.mapped_rows = self.mapped_rows.data_ptr() if self.vmm else None
        self.struct = s
        self.engine = None
        self._net_fp = None
        self._oh = None
        self._graphs = {}          # (G, c, max_states, level budget) -> captured iteration
        self._graph_pool = None    # one memory pool for all of them: they never run concurrently

    @classmethod
    def _chunk_for(cls, rows: int, shape, dt) -> int:
        bpr = int(np.prod(shape[1:], dtype=np.int64)) * torch.empty(0, dtype=dt).element_size()
        if rows * NODE_WORDS * 4 < cls.BIG_CHUNKS_FROM or bpr < 16:
            return 2 << 20
        return (4 << 20) if bpr >= NODE_WORDS * 4 else (8 << 20)

    # ---- the descent paths ------------------------------------------------------------------------------
    def _make_path_store(self, z):
        """path_node / path_act / path_next / short_act as [blocks][B][block] arrays (rc_mcts_t).  Unbounded paths: reserved address
        ranges, block 0 of every tree with memory from the start; a fixed store: ordinary allocations.  A results-only forest keeps
        the two arrays result extraction reads."""
        B, pb, nb = self.B, self.path_block, self.path_blocks
        names = [n for n in _PATH_ARRAYS if not (self.results_only and n in ("path_node", "path_next"))]
        self._path_ranges = {}
        if self.path_vmm:
            try:
                for name in names:
                    esz = torch.empty(0, dtype=_PATH_ARRAYS[name]).element_size()
                    arr = VmmArray.take(nb * B * pb * esz, self.device)
                    self._path_ranges[name] = (arr, esz)
                    arr.ensure(0, B * pb * esz)
            except _hip.RubiksHipError as e:
                import warnings
                warnings.warn(f"MCTSForest: no address space for descent paths of any length ({e}); paths are limited to {pb} levels", RuntimeWarning)
                for arr, _ in self._path_ranges.values():
                    arr.close()
                self._path_ranges, self.path_vmm, self.path_blocks, self.max_path, nb = {}, False, 1, pb, 1
                self.ring_levels = min(self.ring_levels, pb)
        for name, dt in _PATH_ARRAYS.items():
            if name not in names:
                t = z((1, 1, pb), dt)
            elif self.path_vmm:
                t = self._path_ranges[name][0].tensor(dt, (nb, B, pb))
            else:
                t = z((nb, B, pb), dt)
            setattr(self, name, t)
        # levels of tree t's path arrays with memory behind them (a whole number of blocks)
        self.path_rows_host = np.full(B, pb if self.path_vmm else self.max_path, dtype=np.int32)
        self.path_rows = torch.full((B,), pb, dtype=torch.int32, device=self.device) if self.path_vmm else None
        self.paths_seen = np.zeros(B, dtype=np.int64)   # path lengths as of the host's last look (`grow_paths`)

    def ensure_path(self, trees: np.ndarray, levels: np.ndarray) -> int:
        """Memory behind levels 0 .. levels[i] - 1 of tree trees[i]'s path arrays (whole blocks; never less than the tree has);
        the kernels learn of it in stream order.  Returns the bytes newly mapped.  No-op for a fixed path store."""
        if not self.path_vmm:
            return 0
        pb, B = self.path_block, self.B
        trees = np.asarray(trees, dtype=np.int64).reshape(-1)
        want = np.minimum(self.max_path, (np.asarray(levels, dtype=np.int64).reshape(-1) + pb - 1) // pb * pb)
        more = want > self.path_rows_host[trees]
        if not more.any():
            return 0
        new = 0
        for t, w in zip(trees[more], want[more]):
            for b in range(int(self.path_rows_host[t]) // pb, int(w) // pb):
                for arr, esz in self._path_ranges.values():
                    off = (b * B + int(t)) * pb * esz
                    new += arr.ensure(off, off + pb * esz)
            self.path_rows_host[t] = w
        self.path_rows.copy_(torch.from_numpy(self.path_rows_host.copy()).pin_memory(), non_blocking=True)
        return new

    def grow_paths(self, path_len: np.ndarray):
        """The host's look at the path lengths ([B], as of some point of the stream): a tree whose descents have come within a
        quarter block of the end of its path blocks gets the next one -- or as many as one and a half times its path needs (deep
        trees get deeper: growth steps should be few).  A descent that gets there first is suspended by the kernel
        (rc_mcts_t::path_rows) and resumes once the block is there: exact either way."""
        self.paths_seen = np.asarray(path_len, dtype=np.int64).copy()
        if not self.path_vmm:
            return
        near, levels = path_growth_plan(self.path_rows_host, self.paths_seen, self.path_block, self.max_path)
        if len(near):
            self.ensure_path(near, levels)

    def read_path(self, name: str, t: int, n: int) -> np.ndarray:
        """The first n entries of tree t in the blocked path array `name` ("path_node", "path_act", "short_act"), on the host."""
        arr, pb = getattr(self, name), self.path_block
        n = int(n)
        assert n <= int(self.path_rows_host[t]), "levels without memory behind them"
        parts = [arr[b, t, :min(pb, n - b * pb)] for b in range((n + pb - 1) // pb)]
        return (torch.cat(parts) if parts else arr[0, t, :0]).cpu().numpy()

    # ---- memory behind the rows (forests mapped on demand) ---------------------------------------------
    def ensure_rows(self, trees: np.ndarray, rows: np.ndarray) -> int:
        """Memory behind rows 0 .. rows[i] - 1 of tree trees[i] in every per-node array (never less than a tree already has);
        the kernels learn of it in stream order.  Returns the bytes newly mapped.  No-op without vmm."""
        if not self.vmm:
            return 0
        trees = np.asarray(trees, dtype=np.int64).reshape(-1)
        want = np.minimum(self.C + 1, np.asarray(rows, dtype=np.int64).reshape(-1))
        more = want > self.mapped_host[trees]
        if not more.any():
            return 0
        new = 0
        for t, r in zip(trees[more], want[more]):
            base = int(t) * (self.C + 1)
            for arr, bpr in self._ranges.values():
                new += arr.ensure((base + int(self.mapped_host[t])) * bpr, (base + int(r)) * bpr)
            self.mapped_host[t] = r
        self.mapped_rows.copy_(torch.from_numpy(self.mapped_host.copy()).pin_memory(), non_blocking=True)
        return new

    def grow(self, n_nodes: np.ndarray, steps_ahead: int):
        """The host's look at the trees: `n_nodes` (host array, [B]) are node counts at some point of the stream, and up to
        `steps_ahead` iterations may run beyond that point before the next look (a tree gains at most 12 nodes per iteration).
        A tree that can reach the end of its rows in that time gets more (see GROW_ROWS)."""
        self.nodes_seen = np.asarray(n_nodes, dtype=np.int64).copy()
        if not self.vmm:
            return
        trees, rows = growth_plan(self.mapped_host, self.nodes_seen, steps_ahead, self.C + 1, self.B <= self.PREGROW_TREES,
                                  self.GROW_FACTOR, self.GROW_STEP)
        if len(trees):
            self.ensure_rows(trees, rows)

    def _first_rows(self) -> int:
        """Rows a planted tree starts with."""
        return min(self.C + 1, self.GROW_ROWS * (2 if self.B <= self.PREGROW_TREES else 1))

    def _grow_now(self):
        """Direct steppers (tests, tools) have no MCTSRun looking after the mapping: a synchronising look, 256 iterations ahead."""
        self.grow(self.n_nodes.cpu().numpy(), 256)
        self.ensure_rows(np.arange(self.B), np.full(self.B, self._first_rows()))
        self.grow_paths(self.path_len.cpu().numpy())
        self._steps_covered = min(256, max(1, self.path_block // 16)) if self.path_vmm else 256

    def ensure_bfs(self, trees: np.ndarray = None):
        """rc_mcts_shorten's scratch ([rows][2] int32) behind the rows of `trees` (all if None) as counted by `nodes_seen`."""
        if self.bfs is None:
            rows = self.B * (self.C + 1)
            if self.vmm:
                self._ranges_bfs = VmmArray.take(rows * 8, self.device)
                self.bfs = self._ranges_bfs.tensor(torch.int32, (rows, 2))
            else:
                self.bfs = torch.zeros((rows, 2), dtype=torch.int32, device=self.device)
            self.struct.bfs = self.bfs.data_ptr()
        if self.vmm:
            for t in (range(self.B) if trees is None else np.asarray(trees).reshape(-1)):
                base = int(t) * (self.C + 1)
                self._ranges_bfs.ensure(base * 8, (base + int(self.nodes_seen[t]) + 2) * 8)

    _deferred = []   # ranges of forests collected while a HIP graph was being captured: parked by the next close() outside a capture

    @classmethod
    def drain_deferred(cls):
        """Parks the arrays of forests that were collected during a graph capture (they could not synchronise then).  Called where a
        forest is built and before parked memory is given back (`VmmArray.trim`), so that such memory never stays out of the books."""
        if not cls._deferred or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        torch.cuda.synchronize()
        arrays, cls._deferred = cls._deferred, []
        mark = VmmArray.next_park_mark()
        for arr in arrays:
            arr.park(protect_from=mark)

    def close(self):
        """Hands the node store on (forests mapped on demand; others free theirs with their tensors): the arrays are parked for the
        next forest of this shape (`VmmArray.park`; `VmmArray.trim()` releases parked memory).  Synchronises -- except while a HIP
        graph is being captured on this thread (a forest may be garbage-collected at any point, and a synchronisation would break
        the capture): the ranges then wait in `_deferred`.  Agents close the forests they drop themselves (`MCTS._forest_for`);
        `__del__` is the safety net."""
        ranges, pranges = getattr(self, "_ranges", None) or {}, getattr(self, "_path_ranges", None) or {}
        if not ranges and not pranges:
            return
        self._graphs, self._graph_pool = {}, None
        for name in list(ranges) + (list(_NODE_FIELDS) if ranges else []) + list(pranges):
            if hasattr(self, name):
                delattr(self, name)
        arrays = [arr for arr, _ in ranges.values()] + [arr for arr, _ in pranges.values()]
        if getattr(self, "_ranges_bfs", None) is not None:
            self.bfs = None
            arrays.append(self._ranges_bfs)
        self._ranges, self._ranges_bfs, self._path_ranges = None, None, None
        if torch.cuda.is_current_stream_capturing():
            MCTSForest._deferred.extend(arrays)
            return
        torch.cuda.synchronize(self.device)
        arrays, MCTSForest._deferred = MCTSForest._deferred + arrays, []
        mark = VmmArray.next_park_mark()      # this forest's arrays stay together; older parked memory beyond the cap goes first
        for arr in arrays:
            arr.park(protect_from=mark)

    def __del__(self):
        try:
            self.close()
        except Exception as e:   # noqa: BLE001 -- nothing to raise into; but a failed release must not pass unseen
            try:
                import sys
                print(f"MCTSForest: handing the node store on failed: {e!r}", file=sys.stderr)
            except Exception:   # noqa: BLE001 -- interpreter shutdown
                pass

    def subset(self, keep: np.ndarray, results_only: bool = False) -> "MCTSForest":
        """
        A new, smaller forest holding only the trees `keep` (host indices), copied on the device by rc_mcts_copy_trees: the
        rows that exist, nothing beyond.  `nodes_seen` must hold their final node counts (`grow`; the trees are finished).
        results_only: the trees only wait to be turned into results (graph completion, BFS shortening, paths): just the
        arrays those steps read are copied (65 of ~285 bytes per node), the forest cannot be stepped or inspected.
        """
        keep = np.asarray(keep, dtype=np.int64)
        # an ordinary allocation (the caching allocator hands the block of the previous harvest out again): forests that come and
        # go with every harvest are not worth reserving, mapping and unmapping address ranges for.  Its path store is a fixed one
        # that holds the deepest of the (finished) trees' last paths.
        levels = max(self.path_block, self.ring_levels, int(self.paths_seen[keep].max()) + 1) if self.path_vmm else self.max_path
        sub = MCTSForest(len(keep), self.copy_capacity(self.nodes_seen[keep]), levels, self.device, _results_only=results_only, vmm=False,
                         path_block=self.path_block, lds_levels=self.lds_levels, ring_levels=self.ring_levels)
        sub.level_budget, sub._one_launch = self.level_budget, self._one_launch
        sub.set_net(self.engine, self.engine.dtype if hasattr(self.engine, "dtype") else torch.bfloat16)
        sub.adopt(0, self, keep)
        return sub

    COPY_CAPACITY_MAX = (1 << 18) + 8192   # forests up to this capacity are copied into forests of the same capacity (hash tables travel as they are)

    def copy_capacity(self, nodes: np.ndarray) -> int:
        """Capacity of a forest that takes copies of (finished) trees of this one with `nodes` nodes: this forest's own, unless that is
        the address-space-sized capacity of a search bounded by time alone -- an up-front copy of that would be gigabytes per tree --
        then what the trees need (a power of two; the copy rebuilds their hash tables, rc_mcts_copy_trees)."""
        if self.C <= self.COPY_CAPACITY_MAX:
            return self.C
        need = int(np.max(nodes, initial=0)) + 14
        return int(min(self.C, max(1 << 14, 1 << int(np.ceil(np.log2(need))))))

    def adopt(self, pos: int, other: "MCTSForest", trees: np.ndarray):
        """Copies the (finished) trees `trees` of `other` into this forest's slots pos .. pos + len(trees) - 1: what this kind
        of forest keeps of a tree (results-only: keys, neighbours, leaf flags, hash table and the per-tree words result
        extraction reads), rows 0 .. n_nodes only."""
        trees = np.asarray(trees, dtype=np.int64)
        k = len(trees)
        assert other.path_block == self.path_block and (self.results_only or other.ring_levels == self.ring_levels)
        assert other.C == self.C or int(other.nodes_seen[np.asarray(trees, dtype=np.int64)].max(initial=0)) + 13 <= self.C, "the trees do not fit this forest's rows"
        assert pos + k <= self.B and not other.results_only
        plen = other.paths_seen[trees]          # the trees are finished: their last paths, as the host has seen them
        assert int(plen.max(initial=0)) <= self.max_path, "the destination's path store is too small for these trees"
        n = other.nodes_seen[trees]
        self.ensure_rows(np.arange(pos, pos + k), n + 2)
        self.nodes_seen[pos:pos + k] = n
        idx = torch.from_numpy(trees.astype(np.int32)).pin_memory().to(self.device, non_blocking=True)
        _hip.check(self.lib.rc_mcts_copy_trees(ctypes.byref(other.struct), ctypes.byref(self.struct), idx.data_ptr(), k, pos,
                                               _hip.stream_ptr()), "rc_mcts_copy_trees")
        pick = idx.long()
        for name in (_RESULT_TREE if self.results_only else _PER_TREE):
            getattr(self, name)[pos:pos + k] = getattr(other, name)[pick]
        for name in (("path_act",) if self.results_only else ("path_node", "path_act")):   # block by block, for the trees that reach it
            src, dst = getattr(other, name), getattr(self, name)
            dst[0, pos:pos + k] = src[0, pick]
            for b in range(1, self.path_blocks):
                deep = np.flatnonzero(plen > b * self.path_block)
                if len(deep) == 0:
                    break
                dst[b, torch.from_numpy(pos + deep).to(self.device)] = src[b, pick[torch.from_numpy(deep).to(self.device)]]
        self.paths_seen[pos:pos + k] = plen

    def bury(self, pos: int, other: "MCTSForest", trees: np.ndarray):
        """`adopt` into a results-only forest (the trees leave `other` for good)."""
        assert self.results_only
        self.adopt(pos, other, trees)

    def bytes_allocated(self) -> int:
        """HBM behind the forest's search state: mapped bytes of the arrays mapped on demand + the ordinary allocations."""
        plain = [t for name, t in (("keys", self.keys), ("node", getattr(self, "node", None)), ("V", self.V), ("leaf", self.leaf))
                 if t is not None and name not in (self._ranges or {})]
        plain += [self.hash, self.ring_node, self.ring_act, self.ring_len]
        if not self.path_vmm:
            plain += [self.path_node, self.path_act, self.path_next, self.short_act]
        return sum(t.numel() * t.element_size() for t in plain) + sum(arr.mapped_bytes for arr, _ in (self._ranges or {}).values()) + \
            sum(arr.mapped_bytes for arr, _ in (self._path_ranges or {}).values())

    def bytes_mapped(self) -> int:
        """HBM currently behind the arrays mapped on demand."""
        return sum(arr.mapped_bytes for arr, _ in (self._ranges or {}).values())

    def bytes_reserved(self) -> int:
        """Address space of the arrays mapped on demand (what an up-front allocation of the same forest would cost in HBM)."""
        return sum(arr.nbytes for arr, _ in (self._ranges or {}).values())

    # ---- which trees the iterations work on ---------------------------------------------------------
    def rung_for(self, n: int) -> int:
        """Smallest launch size of the ladder that holds n trees."""
        return min((g for g in self.rungs if g >= n), default=self.rungs[0])

    def set_active(self, trees: np.ndarray = None):
        """The iterations from now on work on `trees` (indices, any order; None = all): the list is padded with -1 to the
        ladder's next launch size G, the network runs on 11 G rows.  Stream-ordered (an asynchronous copy into the list the
        kernels read), so it is safe between two iterations of a running forest."""
        if trees is None:
            self.active_buf.copy_(torch.arange(self.B, dtype=torch.int32, device=self.device))
            self.G = self.B
            self._listed = None
        else:
            n = len(trees)
            G = self.rung_for(n)
            host = torch.full((G,), -1, dtype=torch.int32).pin_memory()
            host[:n] = torch.from_numpy(np.ascontiguousarray(trees, dtype=np.int32))
            if self._one_launch and n:
                # one-launch iterations: the rows of the NEXT network call were written by the previous step's expansion, at the
                # trees' old list positions -- they move with their trees
                old = np.arange(self.B) if self._listed is None else self._listed
                where = np.full(self.B, -1, dtype=np.int64)
                where[old[old >= 0]] = np.flatnonzero(old >= 0)
                src = where[np.asarray(trees, dtype=np.int64)]
                assert (src >= 0).all(), "a tree that was not listed cannot be listed again without being planted"
                if not np.array_equal(src, np.arange(n)):
                    cols = lambda slots: torch.from_numpy((slots[:, None] * ROWS + np.arange(ROWS)[None, :]).ravel()).pin_memory().to(  # noqa: E731
                        self.device, non_blocking=True)
                    moved = self.children.soa[:, cols(src)]
                    self.children.soa[:, cols(np.arange(n))] = moved
            self.active_buf[:G].copy_(host, non_blocking=True)
            self._active_host = host   # alive until the copy has run
            self._listed = host.numpy().astype(np.int64).copy()
            self.G = G
        self.struct.n_active = self.G

    def listed(self, trees: torch.Tensor) -> "_McStruct":
        """A copy of the forest's struct whose `active` list is `trees` (int32 device tensor): for the result kernels, which
        post-process the finished trees of a forest where they lie."""
        assert trees.dtype == torch.int32 and trees.is_cuda and trees.is_contiguous() and 0 < trees.numel() <= self.B
        s = _McStruct()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(self.struct), ctypes.sizeof(_McStruct))
        s.active, s.n_active = trees.data_ptr(), int(trees.numel())
        return s

    # ---- network ---------------------------------------------------------------------------------
    def set_net(self, net, dtype=torch.bfloat16):
        """Builds the inference engine for `net`; a no-op when the forest already runs exactly these weights."""
        fp = net_fingerprint(net, dtype)
        if self.engine is not None and fp == self._net_fp:
            return
        self._net_fp = fp
        self.engine = make_inference_net(net, dtype)
        self._fused = bool(getattr(self.engine, "supports_cubes", False))
        if self._fused:   # the input layer reads the child SoA directly: no one-hot matrix
            self._oh = None
            self._x1 = self.engine.workspace(ROWS * self.B)
        else:
            self._oh = torch.empty((ROWS * self.B, 480), dtype=self.engine.input_dtype, device=self.device)
        self._graphs = {}
        self._graph_pool = None   # (a pool whose graphs have all been dropped cannot be captured into again: a fresh one next time)

    rows_per_tree = ROWS

    def _net_input(self):
        """(device cubes holding this iteration's network input, number of rows): the rows of the G listed trees."""
        rows = ROWS * self.G
        return (self.children if rows == self.children.n else DeviceCubes(self.children.soa, rows)), rows

    def _evaluate_children(self):
        """child_soa -> one-hot (HIP kernel) -> network -> softmax -> static probs / values buffers."""
        cubes, rows = self._net_input()
        if self._fused:
            logits, values = self.engine.forward_cubes(cubes, None if self._x1 is None else self._x1[:rows])
        else:
            cubes.as_oh(out=self._oh[:rows])
            logits, values = self.engine(self._oh[:rows])
        torch.softmax(logits, dim=1, out=self.probs[:rows])   # agents.py:552 (`p.softmax(dim=1)`)
        self.values[:rows].copy_(values)

    # ---- search phases ---------------------------------------------------------------------------
    def reset(self, roots: DeviceCubes, max_states: int = None):
        """Empties every tree and plants root t = roots[t] as node 1 (agents.py:466-469); the roots are evaluated and
        expanded by the first two iterations (rc_mcts_t::phase).  max_states: see `plant`."""
        assert roots.n == self.B and self.engine is not None
        self.set_active(None)
        self.plant(None, roots, 0, max_states)

    fused_step = True   # iterations as [network -> rc_mcts_step*] (expansion at the END of a step) when the trees were planted for it
    _one_launch = False

    def plant(self, slots, roots: DeviceCubes, first: int = 0, max_states: int = None, slots_host: np.ndarray = None):
        """Trees `slots` (int32 device tensor, or None for all) restart from roots[first], roots[first + 1], ...: their
        hash tables are cleared by the kernel, nothing else needs clearing (a node's rows are initialised when it is
        created).  Safe between two iterations of a running forest: other trees are not touched.
        max_states (the search's per-tree cap): the roots are expanded right here (rc_mcts_plant_expanded) and the forest's
        iterations become [network -> one tree kernel that backs up, descends and expands the next leaf]; without it the
        three-phase form [expand -> network -> backup + descent] is used.  Planting all trees chooses the form, planting some
        (slots freed in a running forest) must keep it."""
        n = self.B if slots is None else int(slots.numel())
        assert slots is None or (slots.dtype == torch.int32 and slots.is_cuda and slots.is_contiguous())
        assert 0 <= first and first + n <= roots.n
        one = bool(self.fused_step and max_states is not None)
        if self.vmm:   # a root and its children need rows before the kernel runs; the first GROW_ROWS rows of every planted tree
            which = np.arange(self.B) if slots is None else (slots_host if slots_host is not None else slots.cpu().numpy())
            self.ensure_rows(which, np.full(len(which), self._first_rows()))
            self.nodes_seen[which] = 0
        if slots is None:
            self._one_launch = one
        assert one == self._one_launch, "trees planted into a running forest must use the form its iterations run in"
        if one:
            assert self.G == self.B, "roots are expanded into the rows of list position == tree index: plant before narrowing"
            _hip.check(self.lib.rc_mcts_plant_expanded(ctypes.byref(self.struct), None if slots is None else slots.data_ptr(), n,
                                                       roots.soa.data_ptr(), roots.stride, first, int(max_states), _hip.stream_ptr()),
                       "rc_mcts_plant_expanded")
            return
        _hip.check(self.lib.rc_mcts_plant(ctypes.byref(self.struct), None if slots is None else slots.data_ptr(), n,
                                          roots.soa.data_ptr(), roots.stride, first, _hip.stream_ptr()), "rc_mcts_plant")

    level_budget = 0   # new levels a tree may descend per iteration (0 = unlimited, strict lock step)

    def _iteration(self, c: float, max_states: int):
        st = _hip.stream_ptr()
        m = ctypes.byref(self.struct)
        if self._one_launch:   # the rows were left by the previous step's expansion (or by the plant): network, then ONE tree kernel
            if self._fused:
                cubes, rows = self._net_input()
                head = self.engine.head_cubes(cubes, None if self._x1 is None else self._x1[:rows])
                _hip.check(self.lib.rc_mcts_step_head(m, head.data_ptr(), head.stride(0), int(head.dtype == torch.bfloat16), c,
                                                      self.level_budget, max_states, st), "rc_mcts_step_head")
            else:
                self._evaluate_children()
                _hip.check(self.lib.rc_mcts_step(m, self.probs.data_ptr(), self.values.data_ptr(), c, self.level_budget, max_states, st),
                           "rc_mcts_step")
            return
        _hip.check(self.lib.rc_mcts_expand(m, max_states, st), "rc_mcts_expand")
        if self._fused:   # head GEMM output (12 logits + value per row) goes straight into the backup kernel
            cubes, rows = self._net_input()
            head = self.engine.head_cubes(cubes, None if self._x1 is None else self._x1[:rows])
            _hip.check(self.lib.rc_mcts_backup_select_head(m, head.data_ptr(), head.stride(0), int(head.dtype == torch.bfloat16),
                                                           c, self.level_budget, st), "rc_mcts_backup_select_head")
        else:
            self._evaluate_children()
            _hip.check(self.lib.rc_mcts_backup_select(m, self.probs.data_ptr(), self.values.data_ptr(), c, self.level_budget, st),
                       "rc_mcts_backup_select")

    def close_pending(self, c: float):
        """One-launch iterations end with the NEXT leaf's expansion, so a search that stops while trees are running (time limit,
        step count) -- or right after the step whose expansion solved a tree -- leaves expansions that were never backed up.
        This is the second half of the three-phase iteration for them: network on the rows the last step left, backup, and the
        descent that follows (agents.py:555-595), no new expansion.  Afterwards every tree is where the reference's loop
        (expand_leaf, then find_leaf, agents.py:476-490) leaves it after that many iterations.  Trees without a pending expansion
        are not touched."""
        assert self._one_launch and not self.results_only
        st, m = _hip.stream_ptr(), ctypes.byref(self.struct)
        if self._fused:
            cubes, rows = self._net_input()
            head = self.engine.head_cubes(cubes, None if self._x1 is None else self._x1[:rows])
            _hip.check(self.lib.rc_mcts_backup_select_head(m, head.data_ptr(), head.stride(0), int(head.dtype == torch.bfloat16),
                                                           c, 0, st), "rc_mcts_backup_select_head")
        else:
            self._evaluate_children()
            _hip.check(self.lib.rc_mcts_backup_select(m, self.probs.data_ptr(), self.values.data_ptr(), c, 0, st), "rc_mcts_backup_select")
        self.expanded.zero_()

    def step(self, c: float, max_states: int, use_graph: bool = True):
        """One lock-step iteration of every running tree: expand -> network -> backup + select (one kernel).
        A freshly planted tree spends its first two steps on its root (evaluation + expansion, then backup + first descent)."""
        assert not self.results_only
        if self.vmm or self.path_vmm:
            if self._steps_covered <= 0:
                self._grow_now()
            self._steps_covered -= 1
        if not use_graph:
            return self._iteration(c, max_states)
        key = (self.G, float(c), int(max_states), int(self.level_budget), self._one_launch)
        g = self._graphs.get(key)
        if g is None:
            # this call's iteration runs eagerly (hipBLASLt picks its kernels, the allocator settles);
            # the capture that follows only records launches, it does not advance the search.  One graph per launch size,
            # kept for the forest's lifetime: a later search on the same forest replays them from its first iteration on.
            self._iteration(c, max_states)
            torch.cuda.synchronize()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool):
                self._iteration(c, max_states)
            self._graphs[key] = g
            return
        g.replay()

    GRAPH_STEPS = max(1, int(os.environ.get("RUBIKS_GRAPH_STEPS", "4")))   # iterations per replayed graph in `steps` (1: one graph launch per iteration)

    def _graph_of_steps(self, c: float, max_states: int):
        """The HIP graph of GRAPH_STEPS consecutive iterations at the current launch size (captured on first use, once the
        one-iteration graph of that size exists: its eager run has settled the library's kernel choices and the allocator)."""
        key = (self.G, float(c), int(max_states), int(self.level_budget), self._one_launch)
        if key not in self._graphs:
            return None
        g = self._graphs.get(key + (self.GRAPH_STEPS,))
        if g is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool):
                for _ in range(self.GRAPH_STEPS):
                    self._iteration(c, max_states)
            self._graphs[key + (self.GRAPH_STEPS,)] = g
        return g

    def steps(self, n: int, c: float, max_states: int, use_graph: bool = True):
        """n lock-step iterations.  Between two graph launches the queue idles ~9 us (the kernels inside a graph follow each other
        without a gap: profiles/r6_step_timelines_f32s.txt), 5 % of a 32-tree step -- so the iterations go out GRAPH_STEPS to a graph
        launch, the remainder one by one.  The host looks at the trees between calls, never between the iterations of a call."""
        U = self.GRAPH_STEPS
        while n > 0:
            g = self._graph_of_steps(c, max_states) if (use_graph and U > 1 and n >= U) else None
            if g is None:
                self.step(c, max_states, use_graph)
                n -= 1
                continue
            if self.vmm or self.path_vmm:
                if self._steps_covered < U:
                    self._grow_now()
                self._steps_covered -= U
            g.replay()
            n -= U

    def capture_all(self, c: float, max_states: int):
        """Captures the iteration's HIP graph for every launch size of the ladder (`rungs`) ahead of time, on an idle forest (every
        tree marked finished: the tree kernels return at once, the network runs on whatever the row buffers hold).  One-off set-up
        per forest and (c, max_states): a search started afterwards replays graphs from its first iteration on, however it narrows."""
        assert not self.results_only and self.engine is not None
        self.status.fill_(EXHAUSTED)
        self.pending.zero_()
        self.expanded.zero_()
        self._one_launch = bool(self.fused_step)
        for G in self.rungs:
            self.set_active(np.arange(G))
            if (self.G, float(c), int(max_states), int(self.level_budget), self._one_launch) not in self._graphs:
                self.step(c, max_states, use_graph=True)
            if self.GRAPH_STEPS > 1:
                self._graph_of_steps(c, max_states)
        self.set_active(None)
        torch.cuda.synchronize()

    def any_running(self) -> bool:
        return bool((self.status == RUNNING).any().item())

    # ---- results ---------------------------------------------------------------------------------
    def tree_arrays(self, t: int) -> dict:
        """Host copies of tree t's node arrays, shaped like the reference agent's attributes."""
        assert not self.results_only, "this forest only holds what turning finished trees into results needs"
        n = int(self.n_nodes[t].item())
        lo, hi = t * (self.C + 1), t * (self.C + 1) + n + 1
        keys = self.keys[lo:hi].cpu().numpy().view(np.uint32)
        states = unpack_keys(keys)
        return {
            "n": n, "states": states,
            "neighbors": self.nbr[lo:hi].cpu().numpy().astype(np.int64),
            "P": self.P[lo:hi].cpu().numpy().astype(np.float64),
            "V": self.V[lo:hi].cpu().numpy().astype(np.float64),
            "W": self.W[lo:hi].cpu().numpy().astype(np.float64),
            "N": self.N[lo:hi].cpu().numpy().astype(np.int64),
            "L": self._virtual_losses(t, n),
            "leaves": self.leaf[lo:hi].cpu().numpy().astype(bool),
        }

    def _virtual_losses(self, t: int, n: int) -> np.ndarray:
        """The reference's L (agents.py:427) of tree t: every backup clears what the descent before it raised (agents.py:569-570),
        so L is nu times the number of times the PENDING descent path leaves a node by an action (agents.py:589) or arrives at
        one by its reverse (agents.py:591) -- and zero in a solved tree, whose last descent was backed up (agents.py:478-487)."""
        L = np.zeros((n + 1, N_ACT))
        if int(self.status[t].item()) == SOLVED:
            return L
        plen = int(self.path_len[t].item())
        if plen > 1:
            nodes = self.read_path("path_node", t, plen).astype(np.int64)
            acts = self.read_path("path_act", t, plen - 1).astype(np.int64)
            np.add.at(L, (nodes[:-1], acts), 100.0)
            np.add.at(L, (nodes[1:], acts ^ 1), 100.0)
        return L

    def _all_trees(self) -> "_McStruct":
        s = _McStruct()
        ctypes.memmove(ctypes.byref(s), ctypes.byref(self.struct), ctypes.sizeof(_McStruct))
        s.active, s.n_active = None, self.B
        return s

    def complete_graphs(self, trees: torch.Tensor = None):
        """_complete_graph of every solved tree (of `trees`, an int32 device list, if given), on the device (agents.py:597-611)."""
        s = self._all_trees() if trees is None else self.listed(trees)
        _hip.check(self.lib.rc_mcts_complete_graph(ctypes.byref(s), _hip.stream_ptr()), "rc_mcts_complete_graph")

    def shorten_launch(self, trees: torch.Tensor = None, trees_host: np.ndarray = None):
        """_shorten_action_queue of every solved tree (of `trees`) on the device -> short_len[B] (-1 = keep the naive queue),
        short_act[B, max_path]."""
        self.ensure_bfs(trees_host)
        s = self._all_trees() if trees is None else self.listed(trees)
        _hip.check(self.lib.rc_mcts_shorten(ctypes.byref(s), _hip.stream_ptr()), "rc_mcts_shorten")

    def shorten_queues(self):
        """shorten_launch + the two result arrays on the host (the queues' first path_block moves)."""
        self.shorten_launch()
        return self.short_len.cpu().numpy(), self.short_act[0].cpu().numpy()

    def status_snapshot(self):
        """(event, pinned int32[4, B]): per-tree status (row 0), node count (row 1) and path length (row 2) as of the work queued so
        far, readable once the event has passed; [3, 0] != 0: the split engine has written an activation beyond half range (the
        network outputs since then are not numbers: the search is to be repeated in fp32, see DeepAgent._overflowed)."""
        host = torch.zeros((4, self.B), dtype=torch.int32, pin_memory=True)
        host[0].copy_(self.status, non_blocking=True)
        host[1].copy_(self.n_nodes, non_blocking=True)
        host[2].copy_(self.path_len, non_blocking=True)
        flag = getattr(self.engine, "range_flag", None)
        if flag is not None:
            host[3, :1].copy_(flag.reshape(-1)[:1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev, host

    def describe_rows(self, t: int) -> str:
        """Where tree t's rows lie and what the node store says those addresses are (for the error raised when result extraction
        finds rows that are not the tree's data: RC_MCTS_CORRUPT)."""
        from librubiks import _vmm
        out = []
        for name, t_ in (("keys", self.keys), ("nbr", self.nbr), ("hash", self.hash)):
            per_tree = (self.hash_size * 4) if name == "hash" else (self.C + 1) * (16 if name == "keys" else self.struct.node_words * 4)
            addr = int(getattr(self.struct, name)) + int(t) * per_tree
            try:
                kind, base, off = _vmm.classify(addr)
            except Exception as e:   # noqa: BLE001 -- diagnosis must not hide the error it explains
                kind, base, off = f"? ({e})", 0, 0
            out.append(f"{name} @0x{addr:x}: {kind}" + (f" (range 0x{base:x} + {off})" if base else ""))
        return "; ".join(out)

    def neighbors_of(self, t: int, n: int) -> np.ndarray:
        lo = t * (self.C + 1)
        return self.nbr[lo:lo + n + 1].cpu().numpy().astype(np.int64)

    def paths(self):
        """(path_len[B], path_act[B, path_block]) on the host: the paths' first block (`read_path` for a deeper one)."""
        return self.path_len.cpu().numpy(), self.path_act[0].cpu().numpy()
