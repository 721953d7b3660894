"""
Pins oracle/cube.py against (a) fixtures produced by importing the reference
(tests/golden/make_golden.py) and (b) the reference's own known-answer vectors
(reference tests/test_cube.py:33-92,116-139; frontend/src/assets/maps.json via the `maps` fixture).
CPU only.
"""
import numpy as np

from oracle import cube as oc


def test_move_tables_match_reference(golden):
    assert np.array_equal(oc.move_deltas(), golden["maps"])
    assert oc.move_deltas().dtype == np.int8
    lut = oc.move_lut()
    # every LUT row is a permutation of 0..23; a^1 is the inverse; 4x = identity (SURVEY 3.6 #2,#3)
    ident = np.arange(24)
    for a in range(12):
        for k in range(2):
            assert sorted(lut[a, k]) == list(ident)
            assert np.array_equal(lut[a ^ 1, k][lut[a, k]], ident)
            p = ident
            for _ in range(4):
                p = lut[a, k][p]
            assert np.array_equal(p, ident)
        assert (lut[a, 0] != ident).sum() == 12 and (lut[a, 1] != ident).sum() == 8


def test_solved_and_action_constants(golden):
    assert np.array_equal(oc.get_solved(), golden["solved"])
    assert oc.get_solved().dtype == np.int8
    assert np.array_equal(np.array(oc.ACTION_SPACE), golden["action_space"])
    assert np.array_equal(oc.iter_actions(2), golden["iter_actions_2"])
    assert oc.iter_actions(2).dtype == np.uint8
    f, d = oc.indices_to_actions(np.arange(12))
    assert np.array_equal(f, golden["i2a_faces"]) and np.array_equal(d, golden["i2a_dirs"])
    assert np.array_equal(oc.rev_actions(np.arange(12)), golden["rev_actions"])
    assert [oc.rev_action(a) for a in range(12)] == list(golden["rev_action_scalar"])
    # literal values from the reference's tests/test_cube.py:116-127
    assert list(f) == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]
    assert list(d) == [1, 0] * 6


def test_multi_rotate(golden):
    out = oc.multi_rotate(golden["mr_in"], golden["mr_faces"], golden["mr_dirs"])
    assert out.dtype == np.int8 and out.flags["C_CONTIGUOUS"]
    assert np.array_equal(out, golden["mr_out"])
    actions = 2 * golden["mr_faces"].astype(int) + (1 - golden["mr_dirs"].astype(int))
    assert np.array_equal(oc.multi_rotate_actions(golden["mr_in"], actions), golden["mr_out"])
    for i in range(100):
        assert np.array_equal(oc.rotate(golden["mr_in"][i], golden["mr_faces"][i], golden["mr_dirs"][i]),
                              golden["mr_out"][i])


def test_expand12(golden):
    assert np.array_equal(oc.expand12(golden["ex_parents"]), golden["ex_children"])


def test_is_solved(golden):
    assert np.array_equal(oc.multi_is_solved(golden["is_in"]), golden["is_out"])
    assert oc.is_solved(oc.get_solved())
    assert not oc.is_solved(oc.rotate(oc.get_solved(), 0, 1))


def test_as_oh(golden):
    oh = oc.as_oh(golden["oh_in"])
    assert oh.shape == (256, 480) and oh.dtype == np.float32
    assert np.array_equal(np.nonzero(oh)[1].reshape(256, 20), golden["oh_cols"])
    assert np.array_equal(oh[0], golden["oh_dense_row0"])
    assert np.array_equal(oc.oh_indices(golden["oh_in"]), golden["oh_cols"])
    assert np.array_equal(oc.as_oh(golden["oh_in"][5]), golden["oh_single"])


def test_scramble_rng_stream(golden):
    for seed in (0, 42):
        for depth in (20, 24):
            np.random.seed(seed)
            for g in range(16):
                s, f, d = oc.scramble(depth, True)
                assert np.array_equal(s, golden[f"scr_s{seed}_d{depth}_states"][g])
                assert np.array_equal(f, golden[f"scr_s{seed}_d{depth}_faces"][g])
                assert np.array_equal(d, golden[f"scr_s{seed}_d{depth}_dirs"][g])


def test_scramble_inverse_solves():
    # reference tests/test_cube.py:103-114
    np.random.seed(42)
    s, _, _ = oc.scramble(1)
    assert not oc.is_solved(s)
    s, faces, dirs = oc.scramble(20)
    assert not oc.is_solved(s)
    for f, d in zip(reversed(faces), reversed(dirs)):
        s = oc.rotate(s, f, 1 - d)
    assert oc.is_solved(s)


def test_sequence_scrambler(golden):
    for ws in (True, False):
        np.random.seed(0)
        s, oh = oc.sequence_scrambler(8, 20, ws)
        assert np.array_equal(s, golden[f"seq_ws{int(ws)}_states"])
        assert np.array_equal(np.nonzero(oh)[1].reshape(len(s), 20), golden[f"seq_ws{int(ws)}_ohcols"])


def test_as633(golden):
    for s, net in zip(golden["as633_in"], golden["as633_out"]):
        assert np.array_equal(oc.as633(s), net)


# ---- the reference's literal known-answer sticker nets (tests/test_cube.py:33-92) -----------------
SOLVED_NET = "\n".join([
    "      2 2 2            ",
    "      2 2 2            ",
    "      2 2 2            ",
    "4 4 4 0 0 0 5 5 5 1 1 1",
    "4 4 4 0 0 0 5 5 5 1 1 1",
    "4 4 4 0 0 0 5 5 5 1 1 1",
    "      3 3 3            ",
    "      3 3 3            ",
    "      3 3 3            ",
])
AFTER_F_NET = "\n".join([
    "      2 2 2            ",
    "      2 2 2            ",
    "      5 5 5            ",
    "4 4 2 0 0 0 3 5 5 1 1 1",
    "4 4 2 0 0 0 3 5 5 1 1 1",
    "4 4 2 0 0 0 3 5 5 1 1 1",
    "      4 4 4            ",
    "      3 3 3            ",
    "      3 3 3            ",
])
AFTER_ALL12_NET = "\n".join([
    "      2 0 2            ",
    "      5 2 4            ",
    "      2 1 2            ",
    "4 2 4 0 2 0 5 2 5 1 2 1",
    "4 4 4 0 0 0 5 5 5 1 1 1",
    "4 3 4 0 3 0 5 3 5 1 3 1",
    "      3 1 3            ",
    "      5 3 4            ",
    "      3 0 3            ",
])


def test_known_answer_nets():
    s = oc.get_solved()
    assert oc.stringify(s) == SOLVED_NET
    for (f, d), solved in zip(((0, 1), (0, 0), (0, 1), (1, 1), (2, 0), (3, 0)),
                              (False, True, False, False, False, False)):
        s = oc.rotate(s, f, d)
        assert oc.is_solved(s) == solved
    for (f, d), solved in zip(((3, 1), (2, 1), (1, 0), (0, 0)), (False, False, False, True)):
        s = oc.rotate(s, f, d)
        assert oc.is_solved(s) == solved
    assert oc.stringify(oc.rotate(oc.get_solved(), 0, 1)) == AFTER_F_NET
    s = oc.get_solved()
    for d in (0, 1):
        for f in range(6):
            s = oc.rotate(s, f, d)
            assert not oc.is_solved(s)
    assert oc.stringify(s) == AFTER_ALL12_NET


def test_state_bytes_stay_in_range():
    # 5 bits per cubie suffice (device hash keys rely on it)
    rng = np.random.RandomState(3)
    s = np.tile(oc.get_solved(), (5000, 1))
    for _ in range(50):
        s = oc.multi_rotate(s, rng.randint(0, 6, 5000), rng.randint(0, 2, 5000))
        assert s.min() >= 0 and s.max() <= 23
