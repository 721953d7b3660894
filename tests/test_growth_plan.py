"""
Host logic of the on-demand node store (librubiks/solving/mcts_device.py::growth_plan), no GPU: which trees get more rows, when,
and how many.  The reference grows a tree's arrays by doubling when they are full (librubiks/solving/agents.py:450-459); here the
host has to be AHEAD of the trees (a kernel never waits for the host: an expansion without rows is skipped for an iteration), and
growth steps should be few because every map call drains the GPU's queue.
"""
import numpy as np

from librubiks.solving.mcts_device import growth_plan

C1 = 175_001            # rows per tree at the reference's max_states


def test_nothing_grows_while_every_tree_is_far_from_its_rows():
    have = np.full(8, 32768)
    seen = np.array([1, 500, 5000, 12000, 20000, 25000, 30000, 32000 - 12 * 49 - 3])
    trees, rows = growth_plan(have, seen, steps_ahead=48, max_rows=C1, pregrow=True)
    assert len(trees) == 0 and len(rows) == 0


def test_a_tree_that_can_reach_its_last_row_grows_before_it_does():
    have = np.full(4, 32768)
    seen = np.array([100, 32768 - 12 * 49 - 1, 100, 100])       # tree 1 could be one row short after 48 more iterations
    trees, rows = growth_plan(have, seen, steps_ahead=48, max_rows=C1, pregrow=False)
    assert trees.tolist() == [1] and rows.tolist() == [65536]   # doubling
    # a tree far beyond its rows (the host lagged): it gets what it can reach, not just the doubled rows
    trees, rows = growth_plan(np.array([16]), np.array([13]), steps_ahead=48, max_rows=C1, pregrow=False)
    assert trees.tolist() == [0] and rows[0] == 13 + 12 * 49 + 2


def test_small_forests_grow_every_tree_that_is_getting_close_in_the_same_step():
    have = np.array([32768, 32768, 32768, 65536, 175_001])
    seen = np.array([32500, 23000, 22000, 50000, 170_000])      # 0 must; 1 and 3 are past 70 %; 2 is not; 4 has every row already
    trees, rows = growth_plan(have, seen, 48, C1, pregrow=True)
    assert trees.tolist() == [0, 1, 3] and rows.tolist() == [65536, 65536, 65536 + 32768]   # doubling, at most 32 768 rows a step
    trees, rows = growth_plan(have, seen, 48, C1, pregrow=False)                              # large forests: only who must
    assert trees.tolist() == [0] and rows.tolist() == [65536]


def test_rows_never_exceed_the_capacity_and_full_trees_are_left_alone():
    have = np.array([163840, 175_001])
    seen = np.array([163600, 174_990])
    trees, rows = growth_plan(have, seen, 48, C1, pregrow=True)
    assert trees.tolist() == [0] and rows.tolist() == [C1]
    trees, rows = growth_plan(np.array([C1]), np.array([C1 - 1]), 48, C1, pregrow=True)
    assert len(trees) == 0


def test_the_plan_keeps_a_growing_tree_ahead_of_its_nodes():
    """A tree that gains 12 nodes per iteration, looked at every 16 iterations one round late: its rows stay ahead, in few steps."""
    have, n, steps = np.array([32768]), 1, 0
    look_every, grows = 16, 0
    while n + 12 < C1:
        seen_then = n                                   # the host sees the count of one round ago ...
        for _ in range(look_every):
            if n + 12 >= C1:
                break                                   # the tree has reached its capacity: it ends (EXHAUSTED), nothing waits
            assert n + 12 < have[0], "an expansion would have had to wait"
            n += 12
            steps += 1
        trees, rows = growth_plan(have, np.array([seen_then]), look_every + 2 * look_every, C1, pregrow=True)
        if len(trees):
            assert rows[0] > have[0]
            have[0] = rows[0]
            grows += 1
    assert have[0] == C1 and grows <= 6                 # 32 768 -> 65 536 -> 98 304 -> 131 072 -> 163 840 -> 175 001


def test_on_demand_only_where_it_pays():
    """The automatic choice (MCTSForest.on_demand_pays): large forests of trees much larger than their first rows; forests of many
    small trees are allocated up front (one chunk per tree would cost far more than their whole capacity)."""
    from librubiks.solving.mcts_device import MCTSForest as F
    if F.VMM_MIN_BYTES != 1 << 30:
        import pytest
        pytest.skip("RUBIKS_VMM_MIN_GB overrides the choice in this process")
    assert F.on_demand_pays(1024, 175000) and F.on_demand_pays(8192, 175000) and F.on_demand_pays(8192, 50000)
    assert not F.on_demand_pays(8192, 10000)        # 23 GB up front, 36 GB if every tree started with its 16 384 first rows
    assert not F.on_demand_pays(65536, 200)         # 3.7 GB up front, a chunk per tree would be 137 GB
    assert not F.on_demand_pays(1024, 50000)        # forests of <= 2 048 trees start with 32 768 rows per tree: most of 50 000
    assert not F.on_demand_pays(16, 175000)         # under 1 GB
    assert F.on_demand_pays(2048, 70000)
