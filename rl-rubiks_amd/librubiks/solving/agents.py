"""
Search agents of the hot path with the reference's API (librubiks/solving/agents.py):

    MCTS(net, c, search_graph).search(state, time_limit=None, max_states=None) -> bool
    .action_queue (deque of action indices), len(agent) (states explored), str(agent), .tt

plus a batched entry point the reference lacks -- `search_batch(states, ...)` runs one tree per
scramble in lock step on one MI355X; `search` is `search_batch` with a single tree.  The trees are
built by the rc_mcts_* HIP kernels (csrc/rubiks_mcts.hip) with the reference's exact per-tree
semantics; the network runs through an engine of librubiks.model chosen by `net_dtype`: F32_SPLIT (the default: fp32
accuracy, the reference's precision, on the f16 matrix cores), torch.bfloat16 (the fast engine) or torch.float32.
"""
import warnings
import weakref
from collections import deque
from time import perf_counter

import numpy as np
import torch

from librubiks import _hip, gpu, no_grad
from librubiks.cube.device import DeviceCubes
from librubiks.model import F32_SPLIT, F32_SPLIT_DET, Model, net_fingerprint
from librubiks.solving import astar_device as ad
from librubiks.solving import bfs_device as bd
from librubiks.solving import mcts_device as md
from librubiks.utils import TickTock

DEFAULT_NODE_CAP = 1 << 18   # least per-tree / per-problem node capacity of a search bounded by time only
ASTAR_TIME_ONLY_BYTES = 32 << 30   # A* bounded by time only: what its per-node arrays (57 bytes per node, allocated up front) may take


def astar_time_only_capacity(n_problems: int) -> int:
    """Node capacity per problem of an A* search bounded by wall time only (the reference's arrays double without bound,
    agents.py:396-404): 2^18 nodes at least, 2^26 at most, and in between what 32 GB of node arrays allow for the batch --
    one problem gets 2^26 nodes (3.8 GB), 4 096 problems keep 2^18 each."""
    return int(max(DEFAULT_NODE_CAP, min(1 << 26, ASTAR_TIME_ONLY_BYTES // (57 * max(1, int(n_problems))))))
TIME_ONLY_HASH_BYTES = 32 << 30   # MCTS bounded by time only: what the trees' hash tables (the one per-node array that must exist up front) may take


def time_only_capacity(n_trees: int) -> int:
    """Node capacity per tree of an MCTS search bounded by wall time only.  The reference's arrays double for as long as the time
    limit lets the tree grow (agents.py:450-459): there is no node limit.  Here a tree's node rows are address space that gets
    memory as the tree grows (`MCTSForest.grow`), so the capacity is what the kernels can address -- 2^24 - 2 nodes per tree -- or,
    in large batches, what 32 GB of hash tables (8 bytes per node, cleared when a tree is planted) allow; what really ends such a
    search is its time limit or the HBM behind the rows running out (the trees then stop growing and the time limit ends it)."""
    if md.MCTSForest.VMM_MIN_BYTES is None:     # node rows allocated up front (RUBIKS_VMM_MIN_GB=never): the old fixed capacity
        return DEFAULT_NODE_CAP
    return int(max(DEFAULT_NODE_CAP, min(md.MAX_CAPACITY, TIME_ONLY_HASH_BYTES // (8 * max(1, int(n_trees))) - 1)))
bd_MAX_STATES = 1 << 26      # BFS bounded by time only: node slots for 2^26 states (~4 GB)


class Agent:
    _explored_states = 0

    def __init__(self):
        self.action_queue = deque()
        self.tt = TickTock()

    def reset(self, time_limit, max_states):
        self._explored_states = 0
        self.action_queue = deque()
        self.tt.reset()
        if hasattr(self, "net") and hasattr(self.net, "eval"):
            self.net.eval()
        assert time_limit or max_states   # reference agents.py:54
        return time_limit or 1e10, max_states or int(1e10)

    def __len__(self):
        return self._explored_states


class DeepAgent(Agent):
    """
    Agents that evaluate a network.  `net_dtype` selects the inference engine (librubiks.model.make_inference_net): the default
    F32_SPLIT is the reference's precision (its forward is fp32, agents.py:551-552) on the f16 matrix cores; torch.bfloat16 is the
    fast engine, torch.float32 the plain fp32 GEMM chain.
    """
    net_dtype = F32_SPLIT

    def __init__(self, net):
        super().__init__()
        self.net = net
        self._fp32_for = None   # (fingerprint of `net`, fp32 engine) once a search on the split engine left half range

    def _search_net(self):
        """What this search's engine is built from: `net`, or -- after a search on the f16x3 split engine saw an activation
        beyond IEEE half's range -- the fp32 GEMM chain of exactly those weights."""
        if self._fp32_for is not None and self._fp32_for[0] == net_fingerprint(self.net, self.net_dtype):
            return self._fp32_for[1]
        return self.net

    def _overflowed(self, engine) -> bool:
        """True if `engine` (the split engine) wrote an activation it cannot represent during the search just finished: the
        caller repeats the search, which `_search_net` then runs in fp32.  Synchronises (call where results are collected)."""
        if not (hasattr(engine, "overflowed") and engine.overflowed()):
            return False
        fallback = engine.fallback()    # (deterministic mode: raises -- the fp32 chain is not bit-reproducible across batch shapes)
        warnings.warn("SplitF32Net: a hidden activation left IEEE half's range (|x| > 65504) during this search; it is repeated "
                      "on the fp32 GEMM chain, and so are later searches with these weights", RuntimeWarning)
        self._fp32_for = (net_fingerprint(self.net, self.net_dtype), fallback)
        return True

    @classmethod
    def from_saved(cls, loc: str, use_best: bool, **kwargs):
        return cls(Model.load(loc, load_best=use_best).to(gpu), **kwargs)


class QueueTable:
    """
    Action queues of a batch kept as rows of a (games, max_len) uint8 array plus a length per game; a
    `deque` of ints is only built for the games somebody indexes (`table[g]`, iteration).  Behaves like
    the list of deques it replaces.
    """

    def __init__(self, acts: np.ndarray = None, lens: np.ndarray = None, n: int = None):
        if acts is not None:
            self._rows = [(acts, i) for i in range(len(lens))]
            self._lens = np.asarray(lens, dtype=np.int64).copy()
        else:
            self._rows = [None] * n
            self._lens = np.zeros(n, dtype=np.int64)

    def put(self, g: int, other: "QueueTable", i: int):
        self._rows[g], self._lens[g] = other._rows[i], other._lens[i]

    def set_row(self, g: int, acts: np.ndarray):
        """Game g's queue from an array of its own (a queue longer than the rows of the shared array)."""
        arr = np.ascontiguousarray(acts, dtype=np.uint8).reshape(1, -1)
        self._rows[g], self._lens[g] = (arr, 0), arr.shape[1]

    def lengths(self) -> np.ndarray:
        return self._lens

    def padded(self, games=None, fill: int = 255):
        """(uint8 [len(games), longest] array of the games' action queues padded with `fill`, their lengths): the queues of many
        games at once without building a deque per game (replaying / scoring whole result sets)."""
        games = np.arange(len(self)) if games is None else np.asarray(games, dtype=np.int64)
        lens = self._lens[games]
        out = np.full((len(games), int(lens.max()) if len(games) else 0), fill, dtype=np.uint8)
        for o, g in enumerate(games):
            src = self._rows[g]
            if src is not None and lens[o]:
                out[o, :lens[o]] = src[0][src[1], :lens[o]]
        return out, lens

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, g):
        if isinstance(g, slice):
            return [self[i] for i in range(*g.indices(len(self)))]
        src = self._rows[g]
        if src is None:
            return deque()
        acts, i = src
        return deque(int(a) for a in acts[i, :self._lens[g]])

    def __setitem__(self, g, q):
        if isinstance(g, slice):
            for i, qq in zip(range(*g.indices(len(self))), q):
                self[i] = qq
            return
        arr = np.fromiter(q, dtype=np.uint8, count=len(q)).reshape(1, -1)
        self._rows[g], self._lens[g] = (arr, 0), len(q)

    def __iter__(self):
        return (self[g] for g in range(len(self)))


class BatchResult:
    """Per-scramble outcome of a batched search (shapes (B,)); `queues[t]` is tree t's action queue."""

    def __init__(self, solved, lengths, nodes, queues, seconds, iterations, status, game_seconds=None):
        self.solved, self.lengths, self.nodes, self.queues = solved, lengths, nodes, queues
        self.seconds, self.iterations, self.status = seconds, iterations, status
        # Per-game wall interval (float64 [B]) where the agent keeps one: from the moment the game's search starts (its tree is
        # planted / its problem enters the batch) to the moment the host sees it finished -- what the reference's Evaluator times
        # around agent.search (evaluation.py:45-52).  Games of one batch share the GPU, so these intervals OVERLAP: their sum is not
        # the batch's wall time (`seconds`).  None: the agent does not record them.
        self.game_seconds = game_seconds

    @property
    def states_per_sec(self) -> float:
        return float(self.nodes.sum()) / max(self.seconds, 1e-12)

    @property
    def path_overflow_trees(self) -> int:
        """Trees that ended because a descent filled the path store (status PATH_OVERFLOW).  The reference has no such limit
        (agents.py:575-595): with the default store (MCTS(max_path=None)) this is HBM / address space running out and is 0 in
        every run on record; a caller who bounds the store (max_path=...) reads here what that bound cost."""
        return int((np.asarray(self.status) == md.PATH_OVERFLOW).sum())

    def select(self, mask: np.ndarray) -> "BatchResult":
        idx = np.flatnonzero(mask)
        if isinstance(self.queues, QueueTable):
            queues = QueueTable(n=len(idx))
            for o, i in enumerate(idx):
                queues.put(o, self.queues, int(i))
        else:
            queues = [self.queues[i] for i in idx]
        return BatchResult(self.solved[idx], self.lengths[idx], self.nodes[idx], queues,
                           self.seconds, self.iterations[idx], self.status[idx],
                           None if self.game_seconds is None else self.game_seconds[idx])

    @staticmethod
    def merge(n: int, parts, seconds: float) -> "BatchResult":
        """Reassembles per-game results from (original indices, BatchResult) pieces."""
        solved, lengths = np.zeros(n, dtype=bool), np.full(n, -1, dtype=np.int64)
        nodes, iters, status = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        queues = QueueTable(n=n)
        for owner, r in parts:
            solved[owner], lengths[owner], nodes[owner] = r.solved, r.lengths, r.nodes
            iters[owner], status[owner] = r.iterations, r.status
            for i, o in enumerate(owner):
                if isinstance(r.queues, QueueTable):
                    queues.put(int(o), r.queues, i)
                else:
                    queues[int(o)] = r.queues[i]
        return BatchResult(solved, lengths, nodes, queues, seconds, iters, status)


class BFS(Agent):
    """
    Breadth-first search (reference agents.py:92-131) with whole levels expanded on the GPU; solution,
    `len(agent)` and the stop rule are those of the reference's FIFO loop (csrc/rubiks_bfs.hip).
    `time_limit` is tested between launch sequences instead of before every pop.
    """

    def __init__(self, chunk: int = None):
        super().__init__()
        self.chunk = chunk
        self._dev = None

    def _device_for(self, max_states: int) -> bd.BFSDevice:
        cap = int(min(max_states, bd_MAX_STATES))
        if self._dev is None or self._dev.max_states < cap:
            self._dev = None   # frees the old buffers first
            self._dev = bd.BFSDevice(cap, self.chunk)
        return self._dev

    def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        time_limit, max_states = self.reset(time_limit, max_states)
        self.tt.tick()
        dev = self._device_for(max_states)
        deadline = (lambda: self.tt.tock() >= time_limit) if time_limit < 1e10 else None
        solved, actions, seen = dev.search(np.asarray(state), int(min(max_states, dev.max_states)), deadline)
        self._explored_states = seen
        self.action_queue = deque(actions)
        return solved

    def search_batch(self, states, time_limit: float = None, max_states: int = None) -> BatchResult:
        """One search after the other (each one fills the GPU by itself); same result layout as the deep agents."""
        states = states.numpy() if isinstance(states, DeviceCubes) else np.asarray(states)
        B = len(states)
        solved, lengths, nodes = np.zeros(B, dtype=bool), np.full(B, -1, dtype=np.int64), np.zeros(B, dtype=np.int64)
        queues, levels, each = [], np.zeros(B, dtype=np.int64), np.zeros(B)
        tt = TickTock()
        tt.tick()
        for g in range(B):
            t0 = tt.tock()
            solved[g] = self.search(states[g], time_limit, max_states)
            each[g] = tt.tock() - t0
            queues.append(self.action_queue)
            nodes[g], levels[g] = len(self), self._dev.levels
            if solved[g]:
                lengths[g] = len(self.action_queue)
        status = np.where(solved, np.where(nodes == 0, 4, 1), 2)
        return BatchResult(solved, lengths, nodes, queues, tt.tock(), levels, status, each)

    def __str__(self):
        return "Breadth-first search"


class _Harvest:
    """
    The trees of `forest` on their way to becoming per-game results: graph completion and BFS shortening of the
    solved trees (device kernels), then asynchronous copies of the per-tree words and paths into pinned host
    memory, all on the current stream; `result()` assembles the BatchResult once the copies have landed.
    """
    _pinned = {}

    @classmethod
    def _host_like(cls, t: torch.Tensor) -> torch.Tensor:
        """Pinned host tensor shaped like t, from a pool keyed by the row count rounded up to a power of two (harvests
        come in all sizes; a fresh hipHostMalloc per harvest would cost more than the harvest)."""
        rows = 1 << max(4, int(np.ceil(np.log2(max(1, t.shape[0])))))
        key = (rows, tuple(t.shape[1:]), t.dtype)
        free = cls._pinned.setdefault(key, [])
        buf = free.pop() if free else torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, pin_memory=True)
        return buf

    def __init__(self, agent, forest: md.MCTSForest, games: np.ndarray, n: int = None, trees: torch.Tensor = None,
                 trees_host: np.ndarray = None):
        """n: only the first n trees of the forest are real (a partly filled results forest).
        trees: int32 device list -- only these (finished) trees of a forest that may still be running are turned into
        results, where they lie; games[i] is the game of trees[i].  trees_host: the same list on the host (a forest whose rows
        are mapped on demand gives the BFS scratch of exactly these trees its memory)."""
        self.games, self.graph = games, agent.search_graph
        # (the first block of the blocked path arrays: a queue longer than that -- rare -- is fetched by `result`)
        src = {"status": forest.status, "nodes": forest.n_nodes, "iterations": forest.iterations,
               "plen": forest.path_len, "sol": forest.solved_action, "pact": forest.path_act[0]}
        if self.graph:
            forest.complete_graphs(trees)       # _complete_graph of all solved trees in one launch
            forest.shorten_launch(trees, trees_host)   # ... and their BFS shortening in another
            src["slen"], src["sact"] = forest.short_len, forest.short_act[0]
        self.host, self.n = {}, (forest.B if n is None else n) if trees is None else int(trees.numel())
        self.tree_ids = np.arange(self.n) if trees is None else np.asarray(trees_host, dtype=np.int64)
        pick = None if trees is None else trees.long()
        # the list is read by kernels queued on THIS (side) stream; it was allocated under another one, whose allocator would hand
        # the block out again the moment the caller drops it: it lives as long as this harvest, and the allocator is told as well
        self.trees = trees
        if trees is not None:
            trees.record_stream(torch.cuda.current_stream())
        for name, t in src.items():
            part = t[:self.n] if pick is None else t[pick]
            self.host[name] = self._host_like(part)
            self.host[name][:self.n].copy_(part, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.forest = forest   # keeps the buffers alive until the copies have landed

    def ready(self) -> bool:
        return self.event.query()

    def result(self) -> BatchResult:
        self.event.synchronize()
        h = {k: v.numpy()[:self.n] for k, v in self.host.items()}
        status, plen = h["status"].astype(np.int64), h["plen"].astype(np.int64)
        bad = np.flatnonzero(status == md.CORRUPT)
        if len(bad):   # graph completion / BFS shortening met indices that name no node of the tree: not the tree's rows (never seen since the
            # node store flushes translations; a loud error with the addresses instead of the GPU fault a wild index would be)
            t = int(self.tree_ids[bad[0]])
            raise _hip.RubiksHipError(f"MCTS result extraction: {len(bad)} tree(s) hold rows that are not theirs (first: tree {t}, game "
                                      f"{int(self.games[bad[0]])}, {int(h['nodes'][bad[0]])} nodes): {self.forest.describe_rows(t)}")
        acts = h["pact"].copy()
        width = acts.shape[1]                             # one path block
        lens = plen - 1                                   # the actions taken: the best guess of an unsolved tree (agents.py:492)
        won = np.flatnonzero(status == md.SOLVED)
        fits = won[plen[won] <= width]
        acts[fits, plen[fits] - 1] = h["sol"][fits]       # agents.py:483
        lens[won] = plen[won]
        shortened = np.zeros(len(status), dtype=bool)
        if self.graph and len(won):
            short = won[h["slen"][won] >= 0]
            acts[short] = h["sact"][short]
            lens[short] = h["slen"][short]
            shortened[short] = True
        lens[status == md.ROOT_SOLVED] = 0
        solved = (status == md.SOLVED) | (status == md.ROOT_SOLVED)
        queues = QueueTable(acts, lens)
        for i in np.flatnonzero(lens > width):            # a queue that goes beyond the first path block: read where it lies
            t = int(self.tree_ids[i])
            if shortened[i]:
                q = self.forest.read_path("short_act", t, lens[i])
            else:
                q = self.forest.read_path("path_act", t, plen[i] - 1)
                if status[i] == md.SOLVED:
                    q = np.append(q, np.uint8(h["sol"][i]))
            queues.set_row(int(i), q)
        out = BatchResult(solved, np.where(solved, lens, -1), h["nodes"].astype(np.int64), queues, 0.0,
                          h["iterations"].astype(np.int64), status)
        for name, t in self.host.items():
            self._pinned[(t.shape[0], tuple(t.shape[1:]), t.dtype)].append(t)
        self.host, self.forest, self.trees = None, None, None
        return out


class MCTS(DeepAgent):
    """Batched PUCT graph search with virtual loss and max-backup (reference agents.py:415-645)."""

    nu = 100
    snapshot_trees = None      # test hook: a dict -> {game: tree_arrays()} of every tree as its search left it (before graph completion)
    refill_level_budget = 0    # new levels per descent and iteration while scrambles wait for a slot (0 = no limit)

    def __init__(self, net, c: float, search_graph: bool, net_dtype=F32_SPLIT, use_graph: bool = True,
                 max_path: int = None, sync_every: int = 16, level_budget="auto", deterministic: bool = False):
        """
        deterministic: bit-reproducible searches.  The default engines pick a layer plan by row count (whole-K tiles, K cut into
        2..32 chunks), so a state's network outputs carry rounding that depends on how many states share its launch, and a near-tie
        in a PUCT argmax can fall the other way: per-game results agree across batch shapes on > 98 % of games, not on all.  With
        deterministic=True the split engine runs ONE plan for every row count (`SplitF32Net(deterministic=True)`): a game searched
        alone, in a batch, on fewer `slots` or in a narrowed forest builds the same tree, bit for bit (tests/test_full_size_gpu.py).
        It costs throughput at both ends of the row-count range: see DESIGN.md section 3.3 for the measured figures.
        max_path: None (default) = PUCT descents of any length, as in the reference (agents.py:575-595): the select kernel works
        on a path's first 4 096 levels in LDS and on deeper ones where they lie in HBM, and the path arrays get memory block by
        block for the trees that go that deep (`MCTSForest.ensure_path`).  A number bounds the path store instead (a resource
        bound the reference does not have: a tree whose descent fills it ends unsolved with status PATH_OVERFLOW, counted in
        `BatchResult.path_overflow_trees`).  With the ADI-trained net, 39 of 1 024 depth-20 trees needed more than 1 024
        levels and none more than 2 048 (longest solution found: 804 moves).
        level_budget: how many NEW tree levels a PUCT descent may walk per lock-step iteration before it is
        suspended until the next one (0 = unlimited).  Every tree still performs exactly the reference's
        sequence of iterations; a budget only stops the deepest descent of the batch from pacing all trees.
        Measured on 1 024 depth-20 trees with trained weights: over a fixed number of lock-step iterations a
        budget of 32 gains 4 % (23.9 vs 22.9 M nodes/s), but a run to completion gets 2.5x SLOWER (6.1 s vs
        2.4 s: the last stragglers are suspended over and over), so "auto" means 0 (strict lock step).
        """
        super().__init__(net)
        self.level_budget = level_budget
        self.c, self.search_graph = float(c), bool(search_graph)
        if deterministic:
            if net_dtype not in (F32_SPLIT, F32_SPLIT_DET):
                raise ValueError("deterministic=True runs on the split engine (net_dtype=F32_SPLIT): the library GEMMs of the other engines "
                                 "choose their kernels by batch shape")
            net_dtype = F32_SPLIT_DET
        self.deterministic = net_dtype == F32_SPLIT_DET
        self.net_dtype, self.use_graph, self.max_path, self.sync_every = net_dtype, use_graph, max_path, sync_every
        self.forest = None
        self._last_forest = None   # the forest the last search ended in (a compacted one after `compact`)
        self._tree = None      # host copy of game 0's tree, for the reference's inspectable attributes
        self._tree_src = None  # (forest, tree index) holding game 0's tree after the last search

    @classmethod
    def from_saved(cls, loc: str, use_best: bool, c: float, search_graph: bool, **kwargs):
        return cls(Model.load(loc, load_best=use_best).to(gpu), c=c, search_graph=search_graph, **kwargs)

    def __str__(self):
        return ("BFS" if self.search_graph else "Naive") + f" MCTS (c={self.c})"

    def __len__(self):
        return self._explored_states

    # ---- batched search --------------------------------------------------------------------------
    def _forest_for(self, n_trees: int, capacity: int) -> md.MCTSForest:
        f = self.forest
        if f is None or f.B != n_trees or f.C < capacity or f.C_asked > 4 * capacity:
            self.forest = None
            if f is not None:
                # what still points into the forest that is about to hand its memory on: game 0's tree is read out now (the
                # reference's inspectable attributes keep working), the pointers are dropped
                if self._tree_src is not None and self._tree_src[0] is f:
                    if self._tree is None:
                        try:
                            self._host_tree()
                        except Exception:   # noqa: BLE001 -- a forest whose search never finished has no tree to show
                            pass
                    self._tree_src = None
                if self._last_forest is f:
                    self._last_forest = None
                run = getattr(f, "_run", None)
                run = run() if run is not None else None
                if run is None or run.done:
                    f.close()         # node store mapped on demand: parked for the next forest of that shape (not left to __del__)
                # (a run started with `start_batch` and not finished still steps this forest: it keeps it -- and its memory -- until it
                # is finished or dropped; the forest is then collected like any object, `MCTSForest.__del__`)
                del f
            torch.cuda.empty_cache()
            f = self.forest = md.MCTSForest(n_trees, capacity, self.max_path)
        f.set_net(self._search_net(), self.net_dtype)   # every search: `net` may have been trained or replaced since the last one
        f.level_budget = 0 if self.level_budget == "auto" else int(self.level_budget)
        return f

    @no_grad
    def prepare(self, n_trees: int, max_states: int):
        """One-off set-up of a batched search with `n_trees` concurrent trees of capacity `max_states`, so that the search
        itself starts at full speed: allocates the forest (HBM, zero-filled), builds the inference engine and captures the HIP
        graph of every launch size the forest will be narrowed to (~0.1 s per size for the split engine: the allocations of
        its activations).  Optional: a search on an unprepared agent does the same work on the way."""
        cap_states = int(max_states) if max_states and max_states < int(1e10) else time_only_capacity(n_trees)
        forest = self._forest_for(int(n_trees), max(cap_states, 16))
        if self.use_graph:
            forest.capture_all(self.c, cap_states)

    @no_grad
    def search_batch(self, states, time_limit: float = None, max_states: int = None,
                     max_iterations: int = None, compact: bool = True, slots: int = None) -> BatchResult:
        """
        One MCTS tree per row of `states` ((G,20) int8 NumPy array or DeviceCubes).  `max_states` is the
        reference's per-tree cap (stop when len + 12 > max_states); `time_limit` bounds the wall time of the
        whole batch; `max_iterations` bounds every tree's number of expansions (lock-step batches only).
        See `MCTSRun` for how the batch is driven.
        slots: run at most this many trees at a time and give the places of finished trees to the scrambles
        still waiting (continuous batching): the GPU stays full until the last games instead of idling on the
        stragglers of every batch.  Per-game results are those of a plain batch (trees are independent).
            compact: once nobody is waiting, finished trees are dropped from the launches as the running ones become fewer
        (`MCTSForest.set_active`: a shorter list of trees, nothing moves in memory), so the stragglers continue on small
        network batches; the finished trees are turned into results where they lie, on a side stream.
        """
        # a bounded number of iterations must end on a completed one: the three-phase form (the one-launch form of an iteration
        # expands the NEXT leaf at its end)
        run = self.start_batch(states, time_limit, max_states, compact=compact, slots=slots, one_launch=max_iterations is None)
        assert max_iterations is None or run.S == run.n_games, "max_iterations applies to lock-step batches only"
        # a tree's first expansion (its root's) takes two lock-step iterations, every later one a single iteration
        steps = None if max_iterations is None else max_iterations + 1
        while not run.done and (steps is None or run.it < steps):
            run.round(None if steps is None else steps - run.it)
        if max_iterations is not None:
            run.catch_up(max_iterations)
        return run.finish()

    @no_grad
    def start_batch(self, states, time_limit: float = None, max_states: int = None, compact: bool = True,
                    slots: int = None, one_launch: bool = True) -> "MCTSRun":
        time_limit, max_states = self.reset(time_limit, max_states)
        roots = states if isinstance(states, DeviceCubes) else DeviceCubes.from_numpy(np.asarray(states))
        return MCTSRun(self, roots, time_limit, max_states, compact, slots, one_launch)

    # ---- the reference's single-state API ----------------------------------------------------------
    def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        res = self.search_batch(np.asarray(state)[None], time_limit, max_states)
        return bool(res.solved[0])

    def _host_tree(self):
        """Game 0's tree, read from the forest it was harvested in (after a solved graph search the device arrays
        hold the completed graph)."""
        if self._tree is None:
            forest, t = self._tree_src
            self._tree = forest.tree_arrays(t)
        return self._tree

    # inspectable attributes relied on by the reference's tests (tests/test_agents.py:54-92)
    states = property(lambda self: self._host_tree()["states"])
    neighbors = property(lambda self: self._host_tree()["neighbors"])
    leaves = property(lambda self: self._host_tree()["leaves"])
    P = property(lambda self: self._host_tree()["P"])
    V = property(lambda self: self._host_tree()["V"])
    W = property(lambda self: self._host_tree()["W"])
    N = property(lambda self: self._host_tree()["N"])
    L = property(lambda self: self._host_tree()["L"])

    @property
    def indices(self) -> dict:
        tree = self._host_tree()
        return {s.tobytes(): i for i, s in enumerate(tree["states"][1:tree["n"] + 1], start=1)}


def _to_device_async(arr: np.ndarray, device) -> torch.Tensor:
    """Host array -> device tensor without blocking the host: a copy from pageable memory makes the caller wait until the
    GPU has reached that point of the stream, i.e. until everything queued ahead of it has run."""
    return torch.from_numpy(np.ascontiguousarray(arr)).pin_memory().to(device, non_blocking=True)


class MCTSRun:
    """
    A batched MCTS search in progress: `round()` queues the next lock-step iterations, `finish()` returns the
    per-game BatchResult.  All trees of the forest advance together, `sync_every` iterations per round; the host
    never waits for the round it has just queued: it reads the tree states of the PREVIOUS round (an asynchronous
    copy into pinned memory) while the GPU works on the current one, so the launch queue never runs dry.
    Finished trees are turned into results on a side stream.  With `slots` < games their places go to the scrambles
    still waiting, so what result extraction reads of them is first copied into a results forest (`bury`), then `plant`
    clears the slots' hash tables and writes the new roots, which the next two iterations evaluate and expand in step
    with everybody else.  Once nobody is waiting the finished trees stay where they are: they are post-processed in place
    (`_Harvest(trees=...)`) and dropped from the iteration launches by a shorter list of trees (`MCTSForest.set_active`),
    so narrowing a forest costs nothing, whatever the per-tree capacity (copying the survivors of 1 024 trees of capacity
    175 000, the reference's default max_states, took 0.55 s per halving).  While games are waiting, descents may be cut
    at the agent's `refill_level_budget` new levels per iteration (0 = off, the default since descents follow lines: a
    budget then costs more iterations than it saves time per iteration).
    """

    def __init__(self, agent: "MCTS", roots: DeviceCubes, time_limit: float, max_states: int, compact: bool, slots, one_launch: bool = True):
        self.agent, self.roots, self.time_limit, self.compact = agent, roots, time_limit, compact
        self.max_states, self.slots, self.one_launch = max_states, slots, one_launch
        self.n_games = roots.n
        S = self.S = self.n_games if slots is None else max(1, min(int(slots), self.n_games))
        self.cap_states = int(max_states) if max_states < int(1e10) else time_only_capacity(S)
        forest = self.forest = agent._forest_for(S, max(self.cap_states, 16))
        forest._run = weakref.ref(self)     # who steps this forest: an agent does not hand its memory on under a run that is not done
        agent.tt.tick()
        forest.set_active(None)
        self.plant_states = self.cap_states if one_launch else None   # one-launch iterations: roots expanded by the plant itself
        forest.plant(None, roots, 0, self.plant_states)       # the first S scrambles; the others move in as trees finish
        # iterations that may be queued before the host has looked at the node counts: what a planted tree's first rows cover
        # (16 384 rows: its first ~1 360 iterations), so that a long `sync_every` cannot make trees sit out behind the kernel's guard
        forest._steps_covered = min(2 * agent.sync_every, max(1, (forest._first_rows() - 14) // 12))
        self.t_start = np.zeros(self.n_games)               # per game: seconds on the agent's clock when its tree was planted ...
        self.t_end = np.full(self.n_games, np.nan)          # ... and when the host first saw it finished (NaN: still running)
        self.owner = np.arange(S)          # game index of every slot; -1 once its result has been taken and nobody moved in
        self.stale_until = np.full(S, -1)  # snapshots up to this index predate the tree that now lives in the slot
        self.next_game = S
        self.base_budget = forest.level_budget
        if self.next_game < self.n_games and agent.level_budget == "auto":
            forest.level_budget = agent.refill_level_budget
        self.min_refill = max(8, S // 32)
        self.stats = agent.refill_stats = {"iterations": 0, "harvests": 0, "refills": 0, "compactions": 0,
                                           "host_enqueue_s": 0.0, "host_wait_s": 0.0, "host_process_s": 0.0}
        self.side = torch.cuda.Stream()
        self.harvests = []                 # _Harvest objects in flight
        self.grave, self.grave_fill, self.grave_event, self.grave_games = None, 0, None, None
        self.resting = []                  # finished trees left in the forest, waiting for their (batched) result extraction
        self.parts = []                    # (game ids, BatchResult)
        agent._tree, agent._tree_src = None, None
        self.snapshots = deque()           # (index, forest, event, pinned status) of rounds nobody has looked at yet
        self.it, self.q, self.done = 0, 0, False

    def _drain(self, block: bool):
        for h in list(self.harvests):
            if block or h.ready():
                self.parts.append((h.games, h.result()))
                self.harvests.remove(h)

    GRAVE = 256   # finished trees collected before their graph completion / BFS shortening is launched

    def _flush_grave(self):
        """Result extraction for the trees collected so far: ONE launch sequence over up to GRAVE trees on the side stream.
        Post-processing the ~30 trees of every refill by themselves keeps ~30 CUs busy for milliseconds each time, and the
        library GEMMs of the running forest (one workgroup per CU) then take two rounds instead of one."""
        if self.grave is None or self.grave_fill == 0:
            return
        g, n = self.grave, self.grave_fill
        self.stats["flushes"] = self.stats.get("flushes", 0) + 1
        g.status[n:] = md.RUNNING          # slots beyond the fill are not trees
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            h = _Harvest(self.agent, g, self.grave_games[:n].copy(), n=n, trees_host=np.arange(n))
        self.harvests.append(h)
        self.grave_event, self.grave_fill = h.event, 0

    def _flush_resting(self):
        """Result extraction of the finished trees left in the forest, where they lie: one launch sequence on the side stream."""
        if not self.resting:
            return
        idx_np = np.concatenate(self.resting)
        self.resting = []
        forest = self.forest
        trees = _to_device_async(idx_np.astype(np.int32), forest.status.device)
        games = self.games_of_resting[idx_np].copy()
        self.stats["flushes"] = self.stats.get("flushes", 0) + 1
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.harvests.append(_Harvest(self.agent, forest, games, trees=trees, trees_host=idx_np))

    def _snapshot(self, idx_np: np.ndarray):
        if self.agent.snapshot_trees is not None:
            for t in idx_np:
                self.agent.snapshot_trees[int(self.owner[t])] = self.forest.tree_arrays(int(t))

    def _rest(self, idx_np: np.ndarray):
        """The finished trees `idx_np` are done with iterations and nobody needs their slots: they stay in the forest and are
        turned into results GRAVE trees at a time (few, large launch sequences next to the running iterations: see _flush_grave)."""
        self._snapshot(idx_np)
        if not hasattr(self, "games_of_resting"):
            self.games_of_resting = np.full(self.forest.B, -1, dtype=np.int64)
        self.games_of_resting[idx_np] = self.owner[idx_np]
        if self.agent._tree_src is None and (self.owner[idx_np] == 0).any():   # game 0's tree stays inspectable, where it is
            self.agent._tree_src = (self.forest, int(idx_np[self.owner[idx_np] == 0][0]))
        self.resting.append(idx_np.copy())
        self.stats["harvests"] += 1

    def _harvest(self, idx_np: np.ndarray):
        """Copies the finished trees `idx_np` out of the forest (their slots are needed or dropped): what result extraction
        reads of them goes into the results forest, which is processed GRAVE trees at a time on the side stream."""
        forest, agent = self.forest, self.agent
        self._snapshot(idx_np)
        games = self.owner[idx_np].copy()
        if agent._tree_src is None and (games == 0).any():
            # game 0's tree stays inspectable (the reference's attributes): a search forest of its own, with this one tree
            first = idx_np[games == 0][:1]
            one = forest.subset(first, results_only=False)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self.harvests.append(_Harvest(agent, one, np.zeros(1, dtype=np.int64)))
            agent._tree_src = (one, 0)
            idx_np, games = idx_np[games != 0], games[games != 0]
            if len(idx_np) == 0:
                self.stats["harvests"] += 1
                return
        # a last path beyond the first block, or (searches bounded by time alone) more nodes than the grave's rows: a results forest of its own
        deep = (forest.paths_seen[idx_np] > forest.path_block) | (forest.nodes_seen[idx_np] + 13 > min(forest.C, forest.COPY_CAPACITY_MAX))
        if deep.any():
            sub = forest.subset(idx_np[deep], results_only=True)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self.harvests.append(_Harvest(agent, sub, games[deep]))
            idx_np, games = idx_np[~deep], games[~deep]
            if len(idx_np) == 0:
                self.stats["harvests"] += 1
                return
        if len(idx_np) < self.GRAVE // 2:
            if self.grave is None:
                self.grave = md.MCTSForest(self.GRAVE, min(forest.C, forest.COPY_CAPACITY_MAX), forest.path_block, forest.device, _results_only=True, vmm=False,
                                           path_block=forest.path_block, lds_levels=forest.lds_levels, ring_levels=forest.ring_levels)
                self.grave_games = np.zeros(self.GRAVE, dtype=np.int64)
            if self.grave_fill + len(idx_np) > self.GRAVE:
                self._flush_grave()
            if self.grave_event is not None:     # the previous batch must have been read before its slots are overwritten
                torch.cuda.current_stream().wait_event(self.grave_event)
                self.grave_event = None
            self.grave.bury(self.grave_fill, forest, idx_np)
            self.grave_games[self.grave_fill:self.grave_fill + len(idx_np)] = games
            self.grave_fill += len(idx_np)
            self.stats["harvests"] += 1
            return
        sub = forest.subset(idx_np, results_only=True)
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.harvests.append(_Harvest(agent, sub, games))
        self.stats["harvests"] += 1

    def round(self, max_steps: int = None):
        """Queues up to `sync_every` iterations, then acts on the tree states of the round before."""
        agent, forest = self.agent, self.forest
        n_steps = min(agent.sync_every, max(1, self.it))   # short first rounds: a batch that is solved at once ends at once
        if max_steps is not None:
            n_steps = min(n_steps, max_steps)
        t0 = perf_counter()
        forest.steps(n_steps, agent.c, self.cap_states, agent.use_graph)
        self.it += n_steps
        self.stats["iterations"] = self.it
        self.snapshots.append((self.q, forest, *forest.status_snapshot(), self.it))
        self.q += 1
        t1 = perf_counter()
        self.stats["host_enqueue_s"] += t1 - t0
        if len(self.snapshots) < 2 and self.it > 1:
            return            # look at round r - 1 while round r runs
        qi, f_snap, ev, st_host, it_then = self.snapshots.popleft()
        ev.synchronize()
        t2 = perf_counter()
        self.stats["host_wait_s"] += t2 - t1   # time the host had to spare: it waited for the GPU, not the other way round
        if f_snap is forest and int(st_host[3, 0]) != 0:
            # The split engine left half range: from then on its outputs are inf / NaN, and a PUCT descent over NaN scores need
            # never meet a leaf (nor does the reference's on such numbers).  The search stops here; `finish` repeats it in fp32.
            self.done = True
            return
        try:
            if f_snap is forest:
                # the host's look at the node counts: rows for everything the iterations queued since that snapshot, the next
                # round and one more can reach (forests mapped on demand; otherwise only the counts are noted)
                queued = self.it - it_then
                forest.paths_seen = st_host[2].numpy().astype(np.int64)   # (what `_harvest` sizes its copies by, whatever happens below)
                try:
                    forest.grow(st_host[1].numpy(), queued + 2 * agent.sync_every)
                    forest.grow_paths(st_host[2].numpy())      # ... and the next path block for the trees whose descents near the end of theirs
                except _hip.RubiksHipError as e:
                    # HBM has run out behind the node rows / path blocks.  The trees that need more sit out behind the kernels'
                    # guards (exact: nothing is half-written); a search with a time limit ends there, others are stopped here.
                    if not getattr(self, "_hbm_warned", False):
                        warnings.warn(f"MCTS: no more HBM behind the trees' node rows ({e}); the trees that need more stop growing", RuntimeWarning)
                        self._hbm_warned = True
                    if self.time_limit >= 1e10:
                        self.done = True
                        return
                forest._steps_covered = 2 * agent.sync_every
                self.stats["mapped_gb"] = round(forest.bytes_allocated() / 1e9, 2)
            self._act_on(qi, f_snap, st_host[0])
        finally:
            self.stats["host_process_s"] += perf_counter() - t2

    def _act_on(self, qi: int, f_snap, st_host):
        """Harvest / refill / compaction decisions from the tree states of round qi."""
        agent, forest = self.agent, self.forest
        if f_snap is not forest:
            return            # taken before a compaction
        status = st_host.numpy()
        owner = self.owner
        live = owner >= 0
        fresh = self.stale_until < qi
        running = live & ((status == md.RUNNING) | ~fresh)      # a slot refilled after the snapshot runs by definition
        done = np.flatnonzero(live & fresh & (status != md.RUNNING))
        if len(done):                                           # the host has just seen these games finished (first sighting counts)
            g = owner[done]
            first = np.isnan(self.t_end[g])
            self.t_end[g[first]] = agent.tt.tock()
        n_run = int(running.sum())
        out_of_time = agent.tt.tock() >= self.time_limit
        waiting = self.next_game < self.n_games
        self._drain(False)
        if out_of_time or (n_run == 0 and not waiting):
            self.done = True
            return
        if waiting and len(done) and (len(done) >= self.min_refill or n_run == 0):
            self._harvest(done)
            k = min(len(done), self.n_games - self.next_game)
            idx = _to_device_async(done[:k].astype(np.int32), forest.status.device)
            forest.plant(idx, self.roots, self.next_game, self.plant_states, slots_host=done[:k])   # the waiting scrambles move in: roots evaluated by the next two iterations
            owner[done] = -1
            owner[done[:k]] = np.arange(self.next_game, self.next_game + k)
            self.t_start[self.next_game:self.next_game + k] = agent.tt.tock()
            self.stale_until[done[:k]] = self.q - 1     # every snapshot queued so far predates the adoption
            self.next_game += k
            self.stats["refills"] += 1
            if self.next_game >= self.n_games:
                forest.level_budget = self.base_budget   # nobody is waiting any more: strict lock step for the tail
        elif not waiting and self.compact and forest.rung_for(n_run) < forest.G:
            # fewer running trees than the next smaller launch size: the finished ones rest where they are, the iterations
            # go on with a shorter list of trees (nothing is copied)
            if len(done):
                self._rest(done)
                owner[done] = -1
            forest.set_active(np.flatnonzero(owner >= 0))
            self.stats["compactions"] += 1
            if sum(len(r) for r in self.resting) >= self.GRAVE:
                self._flush_resting()

    def catch_up(self, iterations: int):
        """A search bounded by a number of iterations (lock-step batches, three-phase form) must leave every running tree after
        exactly that many expansions and the descent that follows the last one (agents.py:476-490).  A tree sits an iteration out
        when its next node rows or path block have no memory yet (the kernels' guards); such trees -- rare: the host maps ahead --
        are stepped on their own here until they are where the others are."""
        forest = self.forest
        listed, stepped = forest._listed, False
        for _ in range(4 * iterations + 64):
            torch.cuda.synchronize()
            it, st, pend = (x.cpu().numpy() for x in (forest.iterations, forest.status, forest.pending))
            lag = np.flatnonzero((self.owner >= 0) & (st == md.RUNNING) & ((it < iterations) | (pend != 0)))
            if len(lag) == 0:
                break
            forest.grow(forest.n_nodes.cpu().numpy(), 2)
            forest.grow_paths(forest.path_len.cpu().numpy())
            forest._steps_covered = 2
            forest.set_active(lag)
            forest.step(self.agent.c, self.cap_states, self.agent.use_graph)
            self.it += 1
            stepped = True
        if stepped:
            forest.set_active(None if listed is None else listed[listed >= 0])

    def nodes_now(self) -> int:
        """Nodes in the trees currently in the forest plus those of the trees already harvested (synchronises)."""
        self._drain(True)
        live = self.owner >= 0
        if self.resting:
            live = live.copy()
            live[np.concatenate(self.resting)] = True
        live = torch.from_numpy(live).to(self.forest.n_nodes.device)
        buried = int(self.grave.n_nodes[:self.grave_fill].sum().item()) if self.grave_fill else 0
        return int(self.forest.n_nodes[live].sum().item()) + buried + sum(int(r.nodes.sum()) for _, r in self.parts)

    def finish(self) -> BatchResult:
        agent, forest, owner = self.agent, self.forest, self.owner
        if self.one_launch and bool(forest.expanded.any().item()):
            # the search stopped on time or on a step count (or its last step solved a tree): the expansions that step left
            # pending are backed up and followed by their descent, as the reference's loop would have done (agents.py:476-490)
            forest.close_pending(agent.c)
        torch.cuda.synchronize()
        seconds = agent.tt.tock()
        forest.grow(forest.n_nodes.cpu().numpy(), 0)     # the final node counts (what result extraction reads of a tree)
        forest.grow_paths(forest.path_len.cpu().numpy())  # ... and path lengths
        left = np.flatnonzero(owner >= 0)
        if len(left) == forest.B:
            self._snapshot(left)
            self.harvests.append(_Harvest(agent, forest, owner.copy()))
            if agent._tree_src is None and (owner == 0).any():
                agent._tree_src = (forest, int(np.flatnonzero(owner == 0)[0]))
        elif len(left):
            self._rest(left)
        self._flush_resting()
        self._flush_grave()
        self._drain(True)
        result = BatchResult.merge(self.n_games, self.parts, seconds)
        if self.next_game < self.n_games:   # games that never got a slot before the time limit: unsolved, nothing explored
            result.status[self.next_game:] = md.EXHAUSTED
            self.t_start[self.next_game:] = seconds
        result.game_seconds = np.where(np.isnan(self.t_end), seconds, self.t_end) - self.t_start   # (still running at the end: until the end)
        self.done = True
        if agent._overflowed(forest.engine):   # the split engine could not represent an activation: the same search in fp32
            again = MCTSRun(agent, self.roots, self.time_limit, self.max_states, self.compact, self.slots, self.one_launch)
            while not again.done:
                again.round()
            return again.finish()
        agent._last_forest = forest
        agent._explored_states = int(result.nodes[0])
        agent.action_queue = result.queues[0]
        return result


class AStar(DeepAgent):
    """
    Batch weighted A* (DeepCubeA style; reference agents.py:171-413): every iteration expands the
    `expansions` open nodes of lowest cost  lambda * G(node) - value_net(node).
    `search_batch` runs one such search per scramble, all on one GPU; `search` is the batch of one.
    """

    def __init__(self, net, lambda_: float, expansions: int, net_dtype=F32_SPLIT, deterministic: bool = False):
        """deterministic: as for `MCTS` -- one layer plan of the split engine for every row count, so a problem's search does not depend
        on which other problems share its batch."""
        super().__init__(net)
        if deterministic:
            if net_dtype not in (F32_SPLIT, F32_SPLIT_DET):
                raise ValueError("deterministic=True runs on the split engine (net_dtype=F32_SPLIT)")
            net_dtype = F32_SPLIT_DET
        self.lambda_, self.expansions, self.net_dtype = float(lambda_), int(expansions), net_dtype
        self.batch = None
        self._arrays = None

    @classmethod
    def from_saved(cls, loc: str, use_best: bool, lambda_: float, expansions: int, **kwargs):
        return cls(Model.load(loc, load_best=use_best).to(gpu), lambda_=lambda_, expansions=expansions, **kwargs)

    def __str__(self):
        return f"AStar (lambda={self.lambda_}, N={self.expansions})"

    def __len__(self):
        return self._explored_states

    def _batch_for(self, n_problems: int, capacity: int):
        b = self.batch
        if b is None or b.B != n_problems or b.N != self.expansions or b.C < capacity or b.C > 4 * capacity:
            self.batch = None
            torch.cuda.empty_cache()
            b = self.batch = ad.AStarBatch(n_problems, capacity, self.expansions)
        b.set_net(self._search_net(), self.net_dtype)   # every search: `net` may have been trained or replaced since the last one
        return b

    @no_grad
    def search_batch(self, states, time_limit: float = None, max_states: int = None,
                     max_iterations: int = None) -> BatchResult:
        time_limit, max_states = self.reset(time_limit, max_states)
        roots = states if isinstance(states, DeviceCubes) else DeviceCubes.from_numpy(np.asarray(states))
        cap_states = int(max_states) if max_states < int(1e10) else astar_time_only_capacity(roots.n)
        batch = self._batch_for(roots.n, max(cap_states, 12 * self.expansions + 1))
        self.tt.tick()
        batch.reset(roots)
        it = 0
        t_end = np.full(roots.n, np.nan)
        while max_iterations is None or it < max_iterations:
            batch.iteration(self.lambda_, cap_states)
            it += 1
            running = (batch.status == ad.RUNNING).cpu().numpy()      # (the loop synchronises here anyway: any_running)
            now = self.tt.tock()
            t_end[np.isnan(t_end) & ~running] = now                   # first sighting of a finished problem
            if not running.any() or now >= time_limit:
                break
        torch.cuda.synchronize()
        if self._overflowed(batch.engine):   # the split engine could not represent an activation: the same search in fp32
            return self.search_batch(roots, time_limit if time_limit < 1e10 else None, max_states if max_states < int(1e10) else None,
                                     max_iterations)
        seconds = self.tt.tock()
        status = batch.status.cpu().numpy()
        nodes = batch.n_nodes.cpu().numpy().astype(np.int64)
        solved = (status == ad.SOLVED) | (status == ad.ROOT_SOLVED)
        sol_idx = batch.solved_idx.cpu().numpy()
        queues = []
        for b in range(batch.B):
            q = deque()
            if status[b] == ad.SOLVED:   # walk the parent pointers back to the root (agents.py:244-251)
                lo = b * (batch.C + 1)
                par = batch.parents[lo:lo + nodes[b] + 1].cpu().numpy()
                pact = batch.parent_actions[lo:lo + nodes[b] + 1].cpu().numpy()
                i = int(sol_idx[b])
                while i != 1:
                    q.appendleft(int(pact[i]))
                    i = int(par[i])
            queues.append(q)
        lengths = np.array([len(q) if s else -1 for q, s in zip(queues, solved)])
        self._explored_states = int(nodes[0])
        self.action_queue = queues[0]
        self._arrays = None
        return BatchResult(solved, lengths, nodes, queues, seconds, batch.iterations.cpu().numpy(), status,
                           np.where(np.isnan(t_end), seconds, t_end))

    def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        return bool(self.search_batch(np.asarray(state)[None], time_limit, max_states).solved[0])

    # ---- inspectable attributes relied on by the reference's tests (tests/test_agents.py:110-145) ----
    def _host(self):
        if self._arrays is None:
            self._arrays = self.batch.problem_arrays(0)
        return self._arrays

    states = property(lambda self: self._host()["states"])
    G = property(lambda self: self._host()["G"])
    parents = property(lambda self: self._host()["parents"])
    parent_actions = property(lambda self: self._host()["parent_actions"])
    open_queue = property(lambda self: self._host()["open_queue"])

    @property
    def indices(self) -> dict:
        h = self._host()
        return {s.tobytes(): i for i, s in enumerate(h["states"][1:h["n"] + 1], start=1)}

    @no_grad
    def cost(self, states: np.ndarray, indeces: np.ndarray) -> np.ndarray:
        """lambda * G[indeces] - value_net(states) (agents.py:369-383), evaluated on the device net."""
        from librubiks.model import make_inference_net
        eng = self.batch.engine if self.batch is not None else make_inference_net(self._search_net(), self.net_dtype)
        oh = DeviceCubes.from_numpy(np.asarray(states)).as_oh(eng.input_dtype)
        h = -eng.value(oh).cpu().numpy().astype(np.float64)
        return self.lambda_ * np.asarray(self.G)[indeces] + h


# =================================================================================================
# Rollout agents (reference agents.py:82-90, 132-169, 649-726), batched over games on the device
# =================================================================================================
DEFAULT_STEP_CAP = 1000   # steps per game when a rollout search is bounded by time only


def _evaluate(engine, cubes: DeviceCubes):
    """(policy logits float32[n,12], values float32[n]) of device-resident states."""
    if getattr(engine, "supports_cubes", False):
        return engine.forward_cubes(cubes)
    return engine(cubes.as_oh(engine.input_dtype))


def _values(engine, cubes: DeviceCubes):
    if getattr(engine, "supports_cubes", False):
        return engine.value_cubes(cubes)
    return engine.value(cubes.as_oh(engine.input_dtype))


class _StepAgent(Agent):
    """
    Agents that take one move per step.  The reference bounds them by wall time only (its
    `len(self) < max_states` test reads a counter that is updated after the loop, agents.py:30-38);
    here `max_states` additionally caps the number of moves per game, which keeps runs deterministic.
    All games of a batch step together; a finished game idles on the identity action.
    """

    def _actions(self, cubes: DeviceCubes, running: torch.Tensor):
        """-> (uint8 actions [n_padded], bool solved_after [n]) for the current states."""
        raise NotImplementedError

    @no_grad
    def search_batch(self, states, time_limit: float = None, max_states: int = None) -> BatchResult:
        time_limit, max_states = self.reset(time_limit, max_states)
        cubes = states if isinstance(states, DeviceCubes) else DeviceCubes.from_numpy(np.asarray(states))
        cubes = DeviceCubes(cubes.soa.clone(), cubes.n)
        B = cubes.n
        cap = int(max_states) if max_states < int(1e10) else DEFAULT_STEP_CAP
        self.tt.tick()
        solved = cubes.is_solved().clone()
        root_solved = solved.clone()
        running = ~solved
        history, stamps = [], []     # stamps[i]: seconds on the agent's clock when step i had been queued and the loop's test had waited for step i - 1
        steps = torch.zeros(B, dtype=torch.int64, device=solved.device)
        while len(history) < cap and bool(running.any()) and self.tt.tock() < time_limit:
            actions = self._actions(cubes, running)
            actions[:B][~running] = 12                        # identity padding of the move table
            cubes.multi_rotate(actions, out=cubes)
            history.append(actions[:B].clone())
            steps += running
            now = cubes.is_solved()
            solved |= now & running
            running &= ~now
            stamps.append(self.tt.tock())
        torch.cuda.synchronize()
        seconds = self.tt.tock()
        hist = torch.stack(history, 1).cpu().numpy() if history else np.zeros((B, 0), dtype=np.uint8)
        steps, solved_h = steps.cpu().numpy(), solved.cpu().numpy()
        queues = [deque(int(a) for a in hist[g, :steps[g]]) for g in range(B)]
        lengths = np.where(solved_h, steps, -1)
        self._explored_states = int(steps[0])
        self.action_queue = queues[0]
        status = np.where(root_solved.cpu().numpy(), 4, np.where(solved_h, 1, 2))
        # a game's wall interval: from the start of the batch to the host's first look after its last move (`running.any()` synchronises
        # every step, so the look behind step i sees it done); games that ran to the end: the batch's seconds
        after = np.array(stamps[1:] + [seconds]) if stamps else np.zeros(0)
        each = np.where(steps > 0, after[np.maximum(steps, 1) - 1] if len(after) else seconds, stamps[0] if stamps else seconds)
        return BatchResult(solved_h, lengths, steps.astype(np.int64), queues, seconds, steps, status, np.asarray(each, dtype=np.float64))

    def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        return bool(self.search_batch(np.asarray(state)[None], time_limit, max_states).solved[0])


def _pad16(t: torch.Tensor) -> torch.Tensor:
    n = t.numel()
    out = torch.zeros((n + 15) // 16 * 16, dtype=torch.uint8, device=t.device)
    out[:n] = t
    return out


class RandomSearch(_StepAgent):
    """Uniformly random moves from np.random (reference agents.py:82-90); one draw per running game per step."""

    def _actions(self, cubes, running):
        a = torch.from_numpy(np.random.randint(12, size=cubes.n).astype(np.uint8)).to(cubes.soa.device)
        return _pad16(a)

    def __str__(self):
        return "Random depth-first search"


class _DeepStepAgent(_StepAgent, DeepAgent):
    def __init__(self, net, net_dtype=F32_SPLIT):
        DeepAgent.__init__(self, net)
        self.net_dtype, self._engine = net_dtype, None

    def reset(self, time_limit, max_states):
        from librubiks.model import make_inference_net
        out = Agent.reset(self, time_limit, max_states)
        self._engine = make_inference_net(self._search_net(), self.net_dtype)   # rebuilt per search: weights may have been trained
        return out

    def search_batch(self, states, time_limit: float = None, max_states: int = None) -> BatchResult:
        res = _StepAgent.search_batch(self, states, time_limit, max_states)
        if self._overflowed(self._engine):   # the split engine could not represent an activation: the same search in fp32
            res = _StepAgent.search_batch(self, states, time_limit, max_states)
        return res


class PolicySearch(_DeepStepAgent):
    """Follows the policy head: greedy argmax of softmax(policy), or samples it (reference agents.py:132-151)."""

    def __init__(self, net, sample_policy=False, net_dtype=F32_SPLIT):
        super().__init__(net, net_dtype)
        self.sample_policy = sample_policy

    @classmethod
    def from_saved(cls, loc: str, use_best: bool, sample_policy=False, **kw):
        return cls(Model.load(loc, load_best=use_best).to(gpu), sample_policy, **kw)

    def _actions(self, cubes, running):
        logits, _ = _evaluate(self._engine, cubes)
        p = torch.softmax(logits, dim=1)
        if self.sample_policy:   # np.random.choice per game, in game order (agents.py:140)
            pn = p.double().cpu().numpy()
            a = np.array([np.random.choice(12, p=row / row.sum()) for row in pn], dtype=np.uint8)
            return _pad16(torch.from_numpy(a).to(p.device))
        return _pad16(p.argmax(dim=1).to(torch.uint8))

    def __str__(self):
        return f"{'Sampled' if self.sample_policy else 'Greedy'} policy"


class ValueSearch(_DeepStepAgent):
    """Moves to the child of highest value; a solved child is taken at once (reference agents.py:154-169)."""

    def _actions(self, cubes, running):
        kids = cubes.expand12()
        solved = kids.is_solved().view(cubes.n, 12)
        v = _values(self._engine, kids).view(cubes.n, 12)
        best = v.argmax(dim=1)
        first_solved = solved.to(torch.uint8).argmax(dim=1)   # first True (argmax returns the first maximum)
        return _pad16(torch.where(solved.any(dim=1), first_solved, best).to(torch.uint8))

    def __str__(self):
        return "Greedy value"


class EGVM(DeepAgent):
    """
    Epsilon-greedy value maximisation (reference agents.py:649-726): `workers` epsilon-greedy policy
    rollouts of `depth` moves from the current state, then jump to the visited state of highest value.
    The random draws follow the reference's np.random call order, so a game is reproducible against
    it; the workers of a game run in parallel on the device, games run one after another.
    """

    def __init__(self, net, epsilon: float, workers: int, depth: int, net_dtype=F32_SPLIT):
        super().__init__(net)
        self.epsilon, self.workers, self.depth, self.net_dtype = epsilon, workers, depth, net_dtype

    @classmethod
    def from_saved(cls, loc: str, use_best: bool, epsilon: float, workers: int, depth: int, **kw):
        return cls(Model.load(loc, load_best=use_best).to(gpu), epsilon, workers, depth, **kw)

    def __str__(self):
        return f"EGVM (e={self.epsilon}, w={self.workers}, d={self.depth})"

    def search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        rng = np.random.get_state()
        ok = self._search(state, time_limit, max_states)
        if self._overflowed(self._engine):   # the split engine could not represent an activation: the same search (same draws) in fp32
            np.random.set_state(rng)
            ok = self._search(state, time_limit, max_states)
        return ok

    @no_grad
    def _search(self, state: np.ndarray, time_limit: float = None, max_states: int = None) -> bool:
        from librubiks.model import make_inference_net
        time_limit, max_states = self.reset(time_limit, max_states)
        engine = self._engine = make_inference_net(self._search_net(), self.net_dtype)
        self.tt.tick()
        cur = DeviceCubes.from_numpy(np.asarray(state)[None])
        if bool(cur.is_solved()[0]):
            return True
        W, D = self.workers, self.depth
        while self.tt.tock() < time_limit and len(self) + W * D <= max_states:
            cubes = DeviceCubes.empty(W)
            cubes.soa[:, :W] = cur.soa[:, :1]
            paths = np.empty((W, D), dtype=int)
            visited = DeviceCubes.empty(W * D)
            hit = None
            for d in range(D):
                use_random = np.random.choice(2, W, p=[1 - self.epsilon, self.epsilon]).astype(bool)
                actions = np.empty(W, dtype=int)
                actions[use_random] = np.random.randint(0, 12, use_random.sum())
                if (~use_random).any():
                    logits, _ = _evaluate(engine, cubes)
                    actions[~use_random] = logits.argmax(dim=1).cpu().numpy()[~use_random]
                paths[:, d] = actions
                cubes.multi_rotate(_pad16(torch.from_numpy(actions.astype(np.uint8)).cuda()), out=cubes)
                solved = cubes.is_solved().cpu().numpy()
                if solved.any():
                    self._explored_states += (d + 1) * W
                    hit = (int(np.flatnonzero(solved)[0]), d + 1)
                    break
                visited.soa[:, d:W * D:D] = cubes.soa[:, :W]     # row w * depth + d (agents.py:681-682,714)
            if hit is not None:
                self.action_queue += deque(int(a) for a in paths[hit[0], :hit[1]])
                return True
            self._explored_states += W * D
            best = int(_values(engine, visited).argmax())
            cur = DeviceCubes.empty(1)
            cur.soa[:, :1] = visited.soa[:, best:best + 1]
            worker, depth = best // D, best % D
            self.action_queue += deque(int(a) for a in paths[worker, :depth + 1])
        return False

    def search_batch(self, states, time_limit: float = None, max_states: int = None) -> BatchResult:
        states = states.numpy() if isinstance(states, DeviceCubes) else np.asarray(states)
        solved, lengths, nodes, queues = [], [], [], []
        tt = TickTock()
        tt.tick()
        for s in states:
            ok = self.search(s, time_limit, max_states)
            solved.append(ok), lengths.append(len(self.action_queue) if ok else -1)
            nodes.append(len(self)), queues.append(self.action_queue)
        solved = np.array(solved)
        return BatchResult(solved, np.array(lengths), np.array(nodes, dtype=np.int64), queues, tt.tock(),
                           np.zeros(len(states), dtype=int), np.where(solved, 1, 2))
