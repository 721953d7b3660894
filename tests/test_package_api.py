"""
tests/test_rubiks.py of the reference restated for this build (CPU, no kernel is launched): the `no_grad` decorator and
the representation switches, which exist for API compatibility -- only the 20x24 representation is implemented, so
selecting the 6x8x6 one raises instead of silently computing something else.
"""
import pytest
import torch


def test_no_grad_decorator():
    from librubiks import no_grad

    class Probe:
        @no_grad
        def grad_enabled(self, extra=0):
            """doc"""
            return torch.is_grad_enabled(), extra

    torch.set_grad_enabled(True)
    assert torch.is_grad_enabled()
    assert Probe().grad_enabled(extra=3) == (False, 3)
    assert torch.is_grad_enabled()
    assert Probe.grad_enabled.__name__ == "grad_enabled" and Probe.grad_enabled.__doc__ == "doc"


def test_representation_switches():
    from librubiks import cube
    assert cube.get_is2024() is True
    cube.store_repr()
    cube.set_is2024(True)
    cube.restore_repr()
    assert cube.get_is2024() is True
    with pytest.raises(NotImplementedError):
        cube.set_is2024(False)
    assert cube.get_is2024() is True
    with pytest.raises(NotImplementedError):
        cube.as_correct(torch.zeros(1, 6, 8, 6))

    class User:
        is2024 = True

        @cube.with_used_repr
        def which(self):
            return cube.get_is2024()

    u = User()
    assert u.which() is True
    u.is2024 = False
    with pytest.raises(NotImplementedError):
        u.which()
    assert cube.shape() == (20,) and cube.get_oh_shape() == 480 and cube.action_dim == 12
    assert cube.action_names == ("F", "B", "T", "D", "L", "R") and len(cube.action_space) == 12


def test_devices_and_reset():
    import librubiks
    assert librubiks.cpu == torch.device("cpu")
    assert librubiks.gpu.type in ("cuda", "cpu")
    librubiks.reset_cuda()   # harmless without a GPU


def test_ticktock_restated():
    """tests/test_ticktock.py of the reference: nested sections sum their own wall time."""
    from time import sleep
    import numpy as np
    from librubiks.utils import TickTock
    tt = TickTock()
    tt.profile("test0")
    sleep(.01)
    tt.profile("test1")
    sleep(.01)
    tt.end_profile("test1")
    sleep(.01)
    tt.end_profile("test0")
    assert np.isclose(0.03, tt.profiles["test0"].sum(), 1)
    assert np.isclose(0.01, tt.profiles["test1"].sum(), 1)
    assert tt.profiles["test0"].sum() > tt.profiles["test1"].sum() > 0.009
    assert "test1" in str(tt) and len(tt.profiles["test0"]) == 1
    tt.tick()
    sleep(.005)
    assert tt.tock() >= 0.005
    tt.reset()
    assert tt.profiles == {}


def test_launch_size_ladder():
    """MCTSForest narrows a running forest along a fixed ladder of launch sizes (one HIP graph each): the forest's own size, then
    multiples of 32 trees (352 network rows = the layer kernels' row tile), each at most 0.95 of the one before, down to 32."""
    from librubiks.solving.mcts_device import MIN_RUNG, rungs
    for n in (1, 5, 31, 32, 33, 48, 600, 1024, 8192):
        r = rungs(n)
        assert r[0] == n and r == sorted(set(r), reverse=True)
        assert all(x % 32 == 0 for x in r[1:]) and all(b <= 0.95 * a or b == MIN_RUNG for a, b in zip(r, r[1:]))
        assert r[-1] == (MIN_RUNG if n > MIN_RUNG else n)
    assert rungs(1024) == [1024, 960, 896, 832, 768, 704, 640, 608, 576, 544, 512, 480, 448, 416, 384, 352, 320, 288, 256, 224, 192, 160, 128, 96,
                           64, 32]


def test_queue_table_padded_matches_the_per_game_deques():
    """QueueTable.padded: many games' action queues as one padded array -- what bench.py's solution replay walks -- equals the deques
    the table hands out game by game (rows from a shared array, rows set one by one, empty rows)."""
    import numpy as np
    from librubiks.solving.agents import QueueTable
    acts = np.arange(40, dtype=np.uint8).reshape(4, 10) % 12
    table = QueueTable(acts, np.array([3, 0, 10, 7]))
    table[1] = [5, 4, 3]
    merged = QueueTable(n=3)
    merged.put(0, table, 2)
    merged.put(2, table, 1)
    for t, games in ((table, None), (table, [3, 0]), (merged, None)):
        padded, lens = t.padded(games)
        ids = range(len(t)) if games is None else games
        assert padded.dtype == np.uint8 and padded.shape == (len(lens), max(lens))
        for row, g, n in zip(padded, ids, lens):
            assert list(row[:n]) == list(t[g]) and n == len(t[g]) and (row[n:] == 255).all()
    empty, lens = QueueTable(n=2).padded()
    assert empty.shape == (2, 0) and list(lens) == [0, 0]


def test_vmm_classify_replays_an_event_log(tmp_path):
    """tools/vmm_classify.py on a hand-written event file (the format rubiks_vmm.hip writes under RUBIKS_VMM_LOG): a live range with
    two mapped chunks, its alignment slack, a released range on the idle list, a reused one, and addresses that were never ours."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("vmm_classify", os.path.join(ROOT, "tools", "vmm_classify.py"))
    vc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vc)
    MB = 1 << 20
    log = tmp_path / "vmm.log"
    log.write_text("\n".join([
        "# rubiks vmm log: seq op base a b rc",
        f"0 A 0x{0x7000_0000_0000:x} {64 * MB} 0 0",                       # reservation of 64 MiB ...
        f"1 R 0x{0x7000_0000_0000 + 2 * MB:x} {40 * MB} {4 * MB} 1",         # ... handed out from +2 MiB on (4 MiB chunks: aligned base)
        f"2 M 0x{0x7000_0000_0000 + 2 * MB:x} {4 * MB} {8 * MB} 0",          # chunks 1 and 2 get memory
        f"3 A 0x{0x7100_0000_0000:x} {16 * MB} 0 0",
        f"4 R 0x{0x7100_0000_0000:x} {16 * MB} {2 * MB} 1",
        f"5 M 0x{0x7100_0000_0000:x} 0 {2 * MB} 0",
        f"6 X 0x{0x7100_0000_0000:x} {2 * MB} {16 * MB} 0",                  # released ...
        f"7 I 0x{0x7100_0000_0000:x} {16 * MB} 1 0",                         # ... onto the idle list
        f"8 A 0x{0x7200_0000_0000:x} {8 * MB} 0 0",
        f"9 R 0x{0x7200_0000_0000:x} {8 * MB} {2 * MB} 1",
        f"10 X 0x{0x7200_0000_0000:x} 0 {8 * MB} 0",
        f"11 F 0x{0x7200_0000_0000:x} {8 * MB} 1 0",                         # address space given back to the runtime
    ]) + "\n")
    state = vc.replay(str(log))
    kind = lambda a: vc.classify(a, *state)[0]   # noqa: E731
    base = 0x7000_0000_0000 + 2 * MB
    assert kind(base) == "unmapped" and kind(base + 4 * MB) == "mapped" and kind(base + 12 * MB - 1) == "mapped" and kind(base + 12 * MB) == "unmapped"
    assert kind(0x7000_0000_0000 + MB) == "slack" and kind(base + 40 * MB) == "slack"
    assert kind(0x7100_0000_0000 + 5 * MB) == "idle"
    assert kind(0x7200_0000_0000 + MB) == "unknown" and "no longer exists" in vc.classify(0x7200_0000_0000 + MB, *state)[1]
    assert kind(0x1234_0000) == "unknown" and "never inside" in vc.classify(0x1234_0000, *state)[1]
    # the idle range is taken again by a later reservation of its class
    with open(log, "a") as f:
        f.write(f"12 A 0x{0x7100_0000_0000:x} {16 * MB} 0 0\n13 U 0x{0x7100_0000_0000:x} {10 * MB} {2 * MB} 2\n14 M 0x{0x7100_0000_0000:x} {2 * MB} {2 * MB} 0\n")
    state = vc.replay(str(log))
    assert vc.classify(0x7100_0000_0000 + 3 * MB, *state)[0] == "mapped" and vc.classify(0x7100_0000_0000, *state)[0] == "unmapped"
    assert vc.classify(0x7100_0000_0000 + 12 * MB, *state)[0] == "slack"


def test_capacities_of_searches_bounded_by_time_alone():
    """A search with a time limit only has no node limit in the reference (its arrays double, agents.py:396-404,450-459).  The host
    logic that stands in for "unbounded": MCTS gets what the kernels can address (node rows are address space) or what 32 GB of hash
    tables allow for the batch; A* what 32 GB of its (up-front) node arrays allow; never less than the 2^18 nodes of earlier rounds."""
    from librubiks.solving import agents as ag
    from librubiks.solving import mcts_device as md
    assert ag.time_only_capacity(1) == md.MAX_CAPACITY == (1 << 24) - 2
    assert ag.time_only_capacity(1024) == (32 << 30) // (8 * 1024) - 1 and ag.time_only_capacity(10 ** 6) == 1 << 18
    assert [ag.astar_time_only_capacity(b) for b in (1, 8, 4096, 10 ** 6)] == [1 << 26, 1 << 26, 1 << 18, 1 << 18]
    assert ag.astar_time_only_capacity(64) == (32 << 30) // (57 * 64)


def test_queue_table_takes_rows_longer_than_the_shared_array():
    """A queue that goes beyond the first path block (descents of any length) is set from an array of its own (`QueueTable.set_row`)."""
    import numpy as np
    from librubiks.solving.agents import QueueTable
    table = QueueTable(np.zeros((2, 4), dtype=np.uint8), np.array([4, 2]))
    long = (np.arange(5000) % 12).astype(np.uint8)
    table.set_row(1, long)
    assert len(table[1]) == 5000 and list(table[1])[:13] == list(range(12)) + [0] and table.lengths()[1] == 5000
    padded, lens = table.padded()
    assert padded.shape == (2, 5000) and list(lens) == [4, 5000] and (padded[0, 4:] == 255).all() and np.array_equal(padded[1], long)


def test_blocked_path_index_is_dense_in_block_zero():
    """rc_mcts_t's path arrays are [block][tree][2^lg levels]: level k of tree t at ((k >> lg) B + t) << lg | (k & (2^lg - 1)).  Block 0 is
    the dense [B][block] array of rounds 1-5; a tensor shaped (blocks, B, block) indexes the same element (what `MCTSForest.read_path` relies on)."""
    import numpy as np
    B, lg, blocks = 5, 4, 3
    flat = np.arange(blocks * B << lg)
    view = flat.reshape(blocks, B, 1 << lg)
    for t in range(B):
        for k in range(blocks << lg):
            idx = (((k >> lg) * B + t) << lg) | (k & ((1 << lg) - 1))
            assert view[k >> lg, t, k & ((1 << lg) - 1)] == idx
            if k < (1 << lg):
                assert idx == t * (1 << lg) + k


def test_path_growth_plan():
    """Host logic of the on-demand path store: which trees get their next path blocks, and how many."""
    import numpy as np
    from librubiks.solving.mcts_device import path_growth_plan
    have = np.array([4096, 4096, 4096, 8192, 1 << 20])
    seen = np.array([100, 3071, 3072, 8000, (1 << 20) - 1])
    trees, levels = path_growth_plan(have, seen, 4096, 1 << 20)
    assert list(trees) == [2, 3] and list(levels) == [8192, 12288]         # within a quarter block of the end; the full store gets nothing
    trees, levels = path_growth_plan([16], [40000], 16, 1 << 20)           # far behind a deep tree: one and a half times its path, in whole blocks
    assert list(trees) == [0] and levels[0] == 60000 and levels[0] % 16 == 0
    trees, levels = path_growth_plan([1 << 20], [1 << 20], 4096, 1 << 20)  # the address space itself is the end
    assert len(trees) == 0
    trees, levels = path_growth_plan([4096], [5000], 4096, 8192)
    assert list(levels) == [8192]
