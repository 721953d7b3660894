"""
Node storage on demand (rc_vmm_*, librubiks/_vmm.py, MCTSForest(vmm=True)): the reference's node arrays grow as a tree grows
(librubiks/solving/agents.py:450-459); here a forest reserves address space for capacity + 1 rows per tree and maps memory
behind the rows in use.  What must hold:
  * a mapped range behaves like any allocation (exact data, torch views, growth next to running work);
  * rc_mcts_copy_trees copies exactly what the whole-capacity tensor copies of rounds 1-3 copied of the rows that exist;
  * searches on forests mapped on demand -- also with so little mapped ahead that trees have to WAIT for their rows (the
    kernels' `mapped_rows` guard) -- build the reference's trees node for node: the exact-tree tests of the continuous
    batching path (refill, narrowing, results forest) and of the deep production trees are repeated in that mode;
  * an address is mapped again only behind a flush of the GPU's translations: on this platform memory mapped where other memory
    was mapped before is otherwise not coherent (tools/vmm_remap_probe.hip), so rc_vmm_release flushes and keeps the address
    range for the next reservation of its size class -- a forest of one shape followed by a forest of another builds exact
    trees (it did not while freed ranges were handed out again unflushed), data written in an address range's second and
    third life is read back exactly by other kernel shapes, and the idle address space stops growing after the first cycle
    through a set of shapes;
  * rc_vmm_classify / rc_vmm_dump / RUBIKS_VMM_LOG say what an address is to the node store (what a GPU fault report needs);
  * BASELINE configs[1] at the reference's max_states = 175 000 (1 024 x 175 001 rows reserved = 46 GB of node records)
    keeps less than 20 GB mapped.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402

WEIGHTS = os.path.join(ROOT, "weights", "fc_small_r1")


def test_vmm_array_maps_on_demand():
    from librubiks._vmm import CHUNK, VmmArray
    free0 = torch.cuda.mem_get_info()[0]
    arr = VmmArray(40 << 30, torch.device("cuda", 0))            # 40 GB of addresses, no memory yet
    assert arr.mapped_bytes == 0 and torch.cuda.mem_get_info()[0] > free0 - (1 << 30)
    t = arr.tensor(torch.int64, ((40 << 30) // 8,))
    assert t.data_ptr() == arr.ptr and t.is_cuda
    assert arr.ensure(0, 3 * CHUNK) == 3 * CHUNK and arr.ensure(CHUNK, 2 * CHUNK) == 0      # chunks are mapped once
    lo = (17 << 30) + 12345 * 8                                    # a range that starts and ends inside chunks
    assert arr.ensure(lo, lo + 5 * CHUNK) == 6 * CHUNK
    assert arr.mapped_bytes == 9 * CHUNK
    a, b = lo // 8, (lo + 5 * CHUNK) // 8
    t[a:b] = torch.arange(a, b, dtype=torch.int64, device="cuda") * 7
    t[:3 * CHUNK // 8].fill_(-5)
    torch.cuda.synchronize()
    assert torch.equal(t[a:b], torch.arange(a, b, dtype=torch.int64, device="cuda") * 7) and int(t[3 * CHUNK // 8 - 1].item()) == -5
    # growth next to running work that uses the mapped part
    x = torch.randn(4096, 4096, device="cuda")
    for i in range(20):
        y = x @ x
        t[a:b].add_(1)
        arr.ensure((20 << 30) + i * CHUNK, (20 << 30) + (i + 1) * CHUNK)
    torch.cuda.synchronize()
    assert torch.equal(t[a:b], torch.arange(a, b, dtype=torch.int64, device="cuda") * 7 + 20) and y.shape == x.shape
    assert arr.mapped_bytes == 29 * CHUNK
    del t
    arr.close()
    assert arr.ptr == 0 and torch.cuda.mem_get_info()[0] > free0 - (1 << 30)


def test_released_addresses_come_back_clean_and_idle_address_space_is_bounded(tmp_path):
    """hipMemUnmap leaves stale translations on the GPU (tools/vmm_remap_probe.hip: 9 of 9 rounds with wrong data at addresses
    mapped a second time, 0 with an ordinary hipMalloc + hipFree in between): rc_vmm_release flushes, then keeps the address range
    on the idle list of its size class, and the next reservation of the class maps memory there again.  A process cycling through
    20 array shapes three times: (1) every array's data -- written by one kernel shape, read by three others -- is exact in the
    range's first, second and third life; (2) from the second cycle on every reservation is a reused one; (3) the idle address
    space (rc_vmm_retired_bytes) does not grow after the first cycle and is bounded by the classes' sizes; (4) rc_vmm_classify
    names what an address is at each point of an array's life, rc_vmm_dump lists the range."""
    from librubiks import _vmm
    from librubiks._vmm import CHUNK, VmmArray, class_bytes
    dev = torch.device("cuda", 0)
    torch.cuda.synchronize()
    VmmArray.trim()
    g = torch.Generator(device="cuda").manual_seed(0)
    shapes = [(3 + 7 * i + (i % 3) * 40, CHUNK if i % 4 else 2 * CHUNK) for i in range(20)]      # (chunks asked, chunk size): 20 shapes, 8 classes
    retired0 = VmmArray.retired_bytes()
    after_cycle, lives = [], {}
    for cycle in range(3):
        for i, (n, chunk) in enumerate(shapes):
            idle_before = VmmArray.retired_bytes()
            arr = VmmArray(n * chunk, dev, chunk)
            assert arr.nbytes == class_bytes(n * chunk, chunk) >= n * chunk
            lives[arr.ptr] = lives.get(arr.ptr, 0) + 1
            if cycle:   # (an idle range of the class: this test's own from the cycle before, or one an earlier owner of the process left)
                assert VmmArray.retired_bytes() < idle_before, "a reservation after the first cycle did not reuse an idle range of its class"
            mapped = n * chunk                                       # the part the 'forest' uses; the rest of the class stays unmapped
            arr.ensure(0, mapped)
            assert _vmm.classify(arr.ptr + mapped - 1)[0] == "mapped" and _vmm.classify(arr.ptr + 5)[:2] == ("mapped", arr.ptr)
            if arr.nbytes > mapped:
                assert _vmm.classify(arr.ptr + mapped)[0] == "unmapped"     # reserved, no memory: what a guard miss would touch
            rows = mapped // 256
            t = arr.tensor(torch.int32, (rows, 64))
            want = (torch.arange(rows, device=dev, dtype=torch.int32) * 7 + 1000 * cycle + i).view(-1, 1).expand(rows, 64)
            t.copy_(want)
            t[1::3] += 1
            exp = want.clone()
            exp[1::3] += 1
            idx = torch.randint(0, rows, (1 << 16,), device=dev, generator=g)
            assert torch.equal(t.index_select(0, idx), exp.index_select(0, idx)) and torch.equal(t.flip(0), exp.flip(0))
            assert torch.equal(t.cpu(), exp.cpu())
            assert f"base=0x{arr.ptr:x}" in _vmm.dump()
            ptr = arr.ptr
            del t
            torch.cuda.synchronize()
            arr.close()
            assert _vmm.classify(ptr)[0] == "idle"
        after_cycle.append(VmmArray.retired_bytes() - retired0)
    # no growth after the first cycle (0 from the start when earlier work of this process already left an idle range in every class)
    assert after_cycle[1] == after_cycle[0] and after_cycle[2] == after_cycle[0], after_cycle
    classes = {(class_bytes(n * c, c) + (c if c > CHUNK else 0)) for n, c in shapes}
    assert 0 <= after_cycle[0] <= sum(classes)                     # at most one idle range per class (arrays lived one at a time)
    assert max(lives.values()) >= 3
    assert _vmm.classify(1 << 20)[0] == "unknown"
    text = _vmm.dump()
    assert "idle raw=" in text and "event" in text and " U 0x" in text      # reuse events are on record


def test_vmm_log_replays_to_the_librarys_own_classification(tmp_path):
    """RUBIKS_VMM_LOG: the event file of a process (flushed line by line: it survives the abort behind a GPU fault) replayed by
    tools/vmm_classify.py gives, for addresses in every state, what rc_vmm_classify said inside the process."""
    import json
    import subprocess
    import sys
    log = tmp_path / "vmm.log"
    code = f"""
import json, sys, torch
sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, "rl-rubiks_amd")!r}]
from librubiks import _vmm
from librubiks._vmm import CHUNK, VmmArray
dev = torch.device("cuda", 0)
a = VmmArray(40 * CHUNK, dev); a.ensure(3 * CHUNK, 9 * CHUNK)
b = VmmArray(10 * CHUNK, dev, 2 * CHUNK); b.ensure(0, 4 * CHUNK)
c = VmmArray(12 * CHUNK, dev); c.ensure(0, CHUNK); pc = c.ptr; torch.cuda.synchronize(); c.close()
d = VmmArray(5 * CHUNK, dev)
probes = [a.ptr, a.ptr + 3 * CHUNK, a.ptr + 9 * CHUNK - 1, a.ptr + 9 * CHUNK, a.ptr + 63 * CHUNK, b.ptr + CHUNK, b.ptr + 4 * CHUNK, b.ptr - 1,
          pc, pc + 11 * CHUNK, d.ptr, 1 << 20]
print(json.dumps([[p, _vmm.classify(p)[0]] for p in probes]))
"""
    env = dict(os.environ, RUBIKS_VMM_LOG=str(log))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    inside = json.loads(p.stdout.strip().splitlines()[-1])
    assert {k for _, k in inside} >= {"mapped", "unmapped", "idle", "unknown"}
    for addr, kind in inside:
        q = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vmm_classify.py"), str(log), hex(addr)], capture_output=True, text=True, timeout=60)
        assert q.returncode == 0, q.stderr[-2000:]
        assert q.stdout.split()[1] == kind, (hex(addr), kind, q.stdout)


def test_a_forest_after_a_forest_of_another_shape_builds_exact_trees(standin_net, monkeypatch):
    """The sequence that exposed the stale translations: a search on one forest, then a forest of another shape (the first
    one's arrays cannot be taken over: their memory is given back, and their ADDRESS RANGES are what the second forest's
    reservations of the same size classes get), stepped in the three-phase form -- planting, expansion and
    descent are different kernels with different grids, so they met different translations of the same address.  Every tree
    must be the oracle's after each of the first iterations."""
    from librubiks.cube import DeviceCubes
    from librubiks.model import GenericNet
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    monkeypatch.setattr(md.MCTSForest, "VMM_MIN_BYTES", 0)          # every forest on demand, whatever its size
    net = standin_net.cuda()
    onet = oa.TorchNet(net, device="cuda")
    for round_, (n_a, cap_a, n_b, cap_b) in enumerate([(8, 20000, 6, 400), (40, 9000, 5, 700), (3, 30000, 12, 300)]):
        first = MCTS(net, c=0.6, search_graph=True, net_dtype=torch.float32)
        first.search_batch(_roots(n_a, 20, 60 + round_), None, cap_a)          # to the cap: every tree's rows get memory
        assert first.forest.vmm
        del first
        states = _roots(n_b, 15, 70 + round_)
        forest = md.MCTSForest(n_b, cap_b)
        assert forest.vmm
        forest.set_net(GenericNet(net), torch.float32)
        forest.reset(DeviceCubes.from_numpy(states))
        warm = 9
        for _ in range(warm + 1):                                   # a tree's first iteration (its root's) takes two steps
            forest.step(0.6, cap_b, use_graph=False)
        torch.cuda.synchronize()
        for t in range(n_b):
            ref = oa.MCTS(onet, c=0.6, search_graph=False)
            assert not ref.search(states[t], cap_b, max_iterations=warm)
            tree, n = forest.tree_arrays(t), len(ref)
            assert tree["n"] == n, (round_, t)
            assert np.array_equal(tree["neighbors"][:n + 1], ref.neighbors[:n + 1]) and np.array_equal(tree["N"][:n + 1], ref.N[:n + 1])
            assert np.array_equal(tree["states"][1:n + 1], ref.states[1:n + 1]) and np.array_equal(tree["W"][1:n + 1], ref.W[1:n + 1])
        forest.close()


def _roots(n, depth, seed):
    np.random.seed(seed)
    return np.array([oc.scramble(depth, True)[0] for _ in range(n)])


@pytest.mark.parametrize("vmm", [False, True])
def test_copy_trees_keeps_exactly_the_rows_that_exist(standin_net, vmm):
    """Finished trees leave a forest through rc_mcts_copy_trees (MCTSForest.subset / bury): every array of the copy equals the
    source on rows 0 .. n_nodes, in both kinds of destination (search forest, results-only forest), whether or not the forests
    are mapped on demand; graph completion + BFS shortening on the copies give the source's own action queues."""
    from librubiks.cube import DeviceCubes
    from librubiks.solving import mcts_device as md
    from librubiks.solving.agents import MCTS
    states = _roots(24, 6, 3)
    agent = MCTS(standin_net.cuda(), c=0.6, search_graph=True, net_dtype=torch.float32)
    md_vmm = md.MCTSForest.VMM_MIN_BYTES
    md.MCTSForest.VMM_MIN_BYTES = 0 if vmm else None
    try:
        res = agent.search_batch(states, None, 700, compact=False)
        forest = agent.forest
        assert forest.vmm == vmm and res.solved.any()
        keep = np.array([5, 0, 17, 23, 11])
        full = forest.subset(keep)
        slim = forest.subset(keep, results_only=True)
        assert not full.vmm and not slim.vmm and slim.results_only      # forests that come and go with a harvest are ordinary allocations
        for i, t in enumerate(keep):
            n = int(forest.n_nodes[t].item())
            a, b = forest.tree_arrays(int(t)), full.tree_arrays(i)
            assert a["n"] == b["n"] == n
            for k in ("states", "neighbors", "P", "V", "W", "N", "L", "leaves"):
                assert np.array_equal(a[k], b[k]), (t, k)
            lo_s, lo_d = int(t) * (forest.C + 1), i * (slim.C + 1)
            assert torch.equal(slim.keys[lo_d:lo_d + n + 1], forest.keys[lo_s:lo_s + n + 1])
            assert torch.equal(slim.nbr[lo_d:lo_d + n + 1], forest.nbr[lo_s:lo_s + n + 1])
            assert torch.equal(slim.leaf[lo_d:lo_d + n + 1], forest.leaf[lo_s:lo_s + n + 1])
            assert torch.equal(slim.hash[i], forest.hash[t]) and torch.equal(full.hash[i], forest.hash[t])
            assert int(slim.n_nodes[i].item()) == n and int(slim.status[i].item()) == int(forest.status[t].item())
        for f in (full, slim):
            f.complete_graphs()
            lens, acts = f.shorten_queues()
            for i, t in enumerate(keep):
                if res.solved[t] and lens[i] >= 0:
                    assert list(acts[i, :lens[i]]) == list(res.queues[t])
        if vmm:
            assert 0 < forest.bytes_mapped() <= forest.bytes_reserved() and forest.bytes_mapped() <= forest.bytes_allocated()
    finally:
        md.MCTSForest.VMM_MIN_BYTES = md_vmm


@pytest.fixture
def scarce_rows(monkeypatch):
    """Every forest is mapped on demand, 16 rows at a time: the host's look-ahead is all a tree has, and right after a plant
    (16 rows: the root's expansion fits, the next one does not) trees wait for their rows."""
    from librubiks import _vmm
    from librubiks.solving import mcts_device as md
    made = []
    take = _vmm.VmmArray.take.__func__

    def counting(cls, *a, **k):          # every array a forest asks for, new or taken over from a finished forest of the same shape
        arr = take(cls, *a, **k)
        made.append(arr)
        return arr

    monkeypatch.setattr(_vmm.VmmArray, "take", classmethod(counting))
    monkeypatch.setattr(md.MCTSForest, "VMM_MIN_BYTES", 0)
    monkeypatch.setattr(md.MCTSForest, "GROW_ROWS", 16)
    return made


def test_trees_that_wait_for_their_rows_are_still_the_reference_trees(scarce_rows, standin_net):
    """Stand-in net (exact arithmetic): 64 trees searched on a forest that maps 16 rows at a time, against the restated reference
    agent tree by tree; `iterations` counts expansions, not steps, so the waits do not show in any result."""
    from librubiks.solving.agents import MCTS
    net = standin_net.cuda()
    states = _roots(64, 7, 12)
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=torch.float32, sync_every=4)
    res = agent.search_batch(states, None, 900, compact=False)
    assert agent.forest.vmm and len(scarce_rows) >= 4
    onet = oa.TorchNet(net, device="cuda")
    for t in range(0, 64, 5):
        ref = oa.MCTS(onet, c=0.6, search_graph=True)
        ok = ref.search(states[t], 900)
        tree = agent.forest.tree_arrays(t)
        n = len(ref)
        assert bool(res.solved[t]) == ok and res.nodes[t] == n == tree["n"] and res.iterations[t] == ref.iterations
        assert list(res.queues[t]) == list(ref.action_queue)
        assert np.array_equal(tree["N"][:n + 1], ref.N[:n + 1]) and np.array_equal(tree["W"][1:n + 1], ref.W[1:n + 1])
        assert np.array_equal(tree["states"][1:n + 1], ref.states[1:n + 1])


@pytest.mark.parametrize("engine", ["f32s", "bf16"])
def test_refill_narrowing_and_results_forest_on_rows_mapped_on_demand(scarce_rows, engine):
    import test_search_edge_gpu as edge
    edge.test_production_trees_through_refill_narrowing_and_results_forest_equal_oracle(engine)
    assert len(scarce_rows) >= 4          # the search forest's arrays


def test_deep_production_trees_on_rows_mapped_on_demand(scarce_rows):
    import test_search_edge_gpu as edge
    edge.test_deep_production_trees_equal_oracle_on_recorded_outputs("f32s")
    assert len(scarce_rows) >= 4


def test_config2_at_the_reference_cap_keeps_a_fraction_mapped():
    """BASELINE configs[1] at max_states = 175 000 (runeval.py:42-44): 1 024 x 175 001 rows are reserved, what the trees reach is
    mapped.  Every returned solution is replayed; the budget rule holds for whoever is unsolved."""
    from librubiks import cube
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    if os.path.isdir(WEIGHTS):
        net = Model.load(WEIGHTS).eval()
    else:
        torch.manual_seed(0)
        net = Model.create(ModelConfig()).eval()
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(1024, 20, True)
    states = cubes.numpy()
    agent = MCTS(net, c=0.6, search_graph=True)
    cap = 175_000
    res = agent.search_batch(cubes, None, cap)
    forest = agent.forest
    assert forest.vmm and forest.C >= cap and forest.bytes_reserved() > 45e9
    assert forest.bytes_allocated() < 20e9, forest.bytes_allocated()
    assert (res.nodes <= cap).all() and (res.nodes >= 13).all()
    for t in np.flatnonzero(res.solved):
        state = states[t]
        for a in res.queues[t]:
            state = oc.rotate(state, *oc.ACTION_SPACE[a])
        assert oc.is_solved(state) and res.lengths[t] == len(res.queues[t])
    for t in np.flatnonzero(~res.solved):
        assert res.lengths[t] == -1 and (res.nodes[t] + 12 > cap or res.status[t] == 3)
    if os.path.isdir(WEIGHTS):
        assert res.solved.mean() > 0.97
    # the rows of every tree that were mapped cover what the tree reached, and not the whole capacity
    assert (forest.mapped_host >= np.minimum(cap + 1, forest.nodes_seen + 1)).all() and forest.mapped_host.mean() < 0.25 * cap


def test_config5_share_on_large_chunks_equals_the_oracle_on_sampled_trees():
    """One GPU's share of BASELINE configs[4] -- 8 192 concurrent depth-24 trees -- on a forest whose node records reserve 105 GB
    (capacity 50 000): the store then takes 4 MiB chunks for the records and 8 MiB for the keys, at addresses aligned by hand
    inside the reservation (csrc/rubiks_vmm.hip).  Searched with max_states 17 000 at production precision; twelve trees spread
    over the forest (first, last, chunk neighbours) are replayed through the oracle on their own recorded network outputs: nodes,
    neighbours, N, W, L, P, V and the action queues must be the oracle's."""
    from librubiks import cube
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    from librubiks.solving import mcts_device as md
    import test_search_edge_gpu as edge
    if not os.path.isdir(WEIGHTS):
        pytest.skip("needs the trained weights")
    net = Model.load(WEIGHTS).eval()
    np.random.seed(11)
    B, cap = 8192, 17_000
    cubes, _, _ = cube.scramble_batch(B, 24, True)
    states = cubes.numpy()
    agent = MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT)
    agent.prepare(B, 50_000)                                    # the forest the search below runs in (capacity within 4 x its budget)
    forest = agent.forest
    chunks = {name: arr.chunk for name, (arr, _) in forest._ranges.items()}
    assert forest.vmm and forest.bytes_reserved() > 100e9 and chunks["node"] == 4 << 20 and chunks["keys"] == 8 << 20 and chunks["V"] == 2 << 20
    assert (forest.C + 1) % ((4 << 20) // 256) == 0            # every tree starts on a chunk boundary
    res = agent.search_batch(cubes, None, cap, compact=False)
    assert agent.forest is forest and (res.nodes <= cap).all() and res.solved.mean() > 0.5
    assert forest.bytes_mapped() < 0.5 * forest.bytes_reserved()
    for t in (0, 1, 2, 511, 512, 2048, 4095, 4096, 7000, 8189, 8190, 8191):
        tree = forest.tree_arrays(t)
        n = tree["n"]
        table = {tree["states"][i].tobytes(): (tree["P"][i].astype(np.float32), np.float32(tree["V"][i])) for i in range(1, n + 1)}
        ref = oa.MCTS(edge._TableNet(table), c=0.6, search_graph=True)
        before = {}
        complete = ref._complete_graph

        def recording_complete(ref=ref, before=before, complete=complete):
            before["neighbors"] = ref.neighbors.copy()          # the device tree is read after its graph completion ran in place
            complete()

        ok = ref.search(states[t], cap)
        assert bool(res.solved[t]) == ok and res.nodes[t] == len(ref) == n, f"tree {t}"
        assert list(res.queues[t]) == list(ref.action_queue) and res.iterations[t] == ref.iterations, f"tree {t}"
        edge._compare(tree, ref, n)


def test_parked_memory_is_capped_but_the_store_just_finished_stays_whole(monkeypatch):
    """VmmArray.park: HBM behind parked arrays is bounded by PARK_CAP_BYTES -- older owners' arrays are released first -- while the
    arrays one owner parks together (`protect_from` = the mark taken before its first park) all stay, whatever they hold: the next
    forest of that shape takes them over with their memory.  `trim` releases everything (and `take` hands out what is parked)."""
    from librubiks._vmm import CHUNK, VmmArray
    dev = torch.device("cuda", 0)
    torch.cuda.synchronize()
    VmmArray.trim()
    monkeypatch.setattr(VmmArray, "PARK_CAP_BYTES", 8 * CHUNK)

    def owner(sizes):
        arrs = [VmmArray(n * CHUNK, dev) for n in sizes]
        for a in arrs:
            a.ensure(0, a.asked)
            a.tensor(torch.int32, (a.asked // 4,)).fill_(7)
        torch.cuda.synchronize()
        mark = VmmArray.next_park_mark()
        for a in arrs:
            a.park(protect_from=mark)
        return arrs

    first = owner([6, 5])                                           # 11 chunks parked by one owner: above the cap, and kept
    assert all(a.ptr for a in first) and VmmArray.parked_bytes() == 11 * CHUNK
    assert VmmArray.has_parked(6 * CHUNK, dev) and VmmArray.has_parked(5 * CHUNK, dev)
    second = owner([7, 3])                                          # the next owner: the first one's arrays go (oldest first) ...
    assert all(a.ptr == 0 for a in first) and all(a.ptr for a in second)
    assert VmmArray.parked_bytes() == 10 * CHUNK and not VmmArray.has_parked(6 * CHUNK, dev)
    third = owner([2])                                              # ... until what is parked fits: 7 + 3 + 2 > 8 -> the 7 goes, 3 + 2 stay
    assert second[0].ptr == 0 and second[1].ptr and third[0].ptr and VmmArray.parked_bytes() == 5 * CHUNK
    again = VmmArray.take(3 * CHUNK, dev)                           # the same shape takes a parked array over, memory included
    assert again is second[1] and again.mapped_bytes == 3 * CHUNK and VmmArray.parked_bytes() == 2 * CHUNK
    t = again.tensor(torch.int32, (3 * CHUNK // 4,))
    assert int(t.min().item()) == 7 and int(t.max().item()) == 7
    del t
    torch.cuda.synchronize()
    again.close()
    assert VmmArray.trim() == 1 and VmmArray.parked_bytes() == 0 and third[0].ptr == 0
