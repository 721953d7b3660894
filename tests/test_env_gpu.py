"""
GPU parity tests of the cube-environment kernels (run on an MI355X with `-m gpu`).

Every check goes through the product's public API / DeviceCubes, i.e. through the C ABI of
librubiks_hip.so, and compares bit-for-bit with (a) the golden fixtures generated from the
reference and (b) the NumPy oracle on seeded inputs, including empty / ragged sizes on both sides
of every kernel-variant threshold.  Large sizes are covered by size-independent properties.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cube as oc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def cube():
    from librubiks import cube as c
    return c


def reachable(n, moves=30, seed=0):
    rng = np.random.RandomState(seed)
    s = np.tile(oc.get_solved(), (n, 1))
    for _ in range(moves):
        s = oc.multi_rotate_actions(s, rng.randint(0, 12, n))
    return s


SIZES = [1, 3, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 4099, 70001]


# ---------------------------------------------------------------------------------------------
# multi_rotate
# ---------------------------------------------------------------------------------------------
def test_multi_rotate_golden(cube, golden):
    out = cube.multi_rotate(golden["mr_in"], golden["mr_faces"], golden["mr_dirs"])
    assert out.dtype == np.int8 and out.shape == (4096, 20) and out.flags["C_CONTIGUOUS"]
    assert np.array_equal(out, golden["mr_out"])
    for i in range(8):
        assert np.array_equal(cube.rotate(golden["mr_in"][i], golden["mr_faces"][i], golden["mr_dirs"][i]),
                              golden["mr_out"][i])


@pytest.mark.parametrize("n", SIZES + [(1 << 20) + 5])
def test_multi_rotate_vs_oracle(cube, n):
    rng = np.random.RandomState(n)
    s = reachable(n, 12, seed=n)
    faces, dirs = rng.randint(0, 6, n), rng.randint(0, 2, n)   # BOTH directions (reference test draws only 0)
    before = s.copy()
    out = cube.multi_rotate(s, faces, dirs)
    assert np.array_equal(s, before), "inputs must not be mutated"
    assert np.array_equal(out, oc.multi_rotate(s, faces, dirs))


def test_multi_rotate_empty(cube):
    out = cube.multi_rotate(np.empty((0, 20), dtype=np.int8), np.empty(0, dtype=int), np.empty(0, dtype=int))
    assert out.shape == (0, 20) and out.dtype == np.int8


def test_multi_rotate_in_place_alias(cube):
    from librubiks.cube import DeviceCubes
    s = reachable(5000, 10, seed=5)
    a = np.random.RandomState(1).randint(0, 12, 5008).astype(np.uint8)
    cubes = DeviceCubes.from_numpy(s)
    cubes.multi_rotate(torch.from_numpy(a).cuda(), out=cubes)
    assert np.array_equal(cubes.numpy(), oc.multi_rotate_actions(s, a[:5000]))


def test_multi_rotate_properties_full_size(cube):
    """2^24 cubes (BASELINE micro-bench size): a then a^1 is the identity; a four times is the identity."""
    from librubiks.cube import DeviceCubes
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(0)
    cubes = DeviceCubes.solved(n)
    for _ in range(6):
        cubes = cubes.multi_rotate(torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g))
    a = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g)
    fwd = cubes.multi_rotate(a)
    assert not torch.equal(fwd.soa[:, :n], cubes.soa[:, :n])
    back = fwd.multi_rotate(a ^ 1)
    assert torch.equal(back.soa[:, :n], cubes.soa[:, :n])
    x = cubes
    for _ in range(4):
        x = x.multi_rotate(a)
    assert torch.equal(x.soa[:, :n], cubes.soa[:, :n])
    assert int(cubes.soa.min()) >= 0 and int(cubes.soa.max()) <= 23
    # spot-check a slice against the oracle
    sl = slice(12345, 12345 + 4096)
    host = cubes.soa[:, sl].T.contiguous().cpu().numpy()
    assert np.array_equal(fwd.soa[:, sl].T.contiguous().cpu().numpy(),
                          oc.multi_rotate_actions(host, a[sl].cpu().numpy()))


# ---------------------------------------------------------------------------------------------
# expand12
# ---------------------------------------------------------------------------------------------
def test_expand12_golden(cube, golden):
    from librubiks.cube import DeviceCubes
    kids = DeviceCubes.from_numpy(golden["ex_parents"]).expand12().numpy()
    assert np.array_equal(kids, golden["ex_children"])
    # the reference idiom through the drop-in API gives the same rows
    idiom = cube.multi_rotate(np.repeat(golden["ex_parents"], 12, axis=0), *cube.iter_actions(256))
    assert np.array_equal(idiom, golden["ex_children"])


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 85, 86, 255, 256, 257, 1000, 3001, (1 << 18) + 3])
def test_expand12_vs_oracle(n):
    from librubiks.cube import DeviceCubes
    s = reachable(n, 14, seed=100 + n)
    kids = DeviceCubes.from_numpy(s).expand12()
    assert kids.n == 12 * n
    assert np.array_equal(kids.numpy(), oc.expand12(s))


@pytest.mark.parametrize("n", [1, 5, 64, 85, 257, 1000, 3001, (1 << 18) + 3])
def test_expand12_flags_is_the_three_calls_in_one(n):
    """rc_expand12_flags = expand12 + is_solved(parents) + is_solved(children) in one launch (a data-generation step of an ADI rollout,
    train.py:285-296): the same children, and the oracle's flags -- with solved parents, parents one move from solved (whose child
    through the reverse move is solved) and ordinary ones in the batch."""
    from librubiks.cube import DeviceCubes
    s = reachable(n, 14, seed=300 + n)
    near = oc.expand12(oc.get_solved()[None])                    # the twelve states one move from solved
    for i in range(0, n, 7):
        s[i] = near[(i // 7) % 12] if i % 3 else oc.get_solved()
    kids, ps, ks = DeviceCubes.from_numpy(s).expand12_flags()
    want = oc.expand12(s)
    assert np.array_equal(kids.numpy(), want)
    assert np.array_equal(ps.cpu().numpy(), oc.multi_is_solved(s)) and np.array_equal(ks.cpu().numpy(), oc.multi_is_solved(want))
    assert ks.cpu().numpy().sum() >= min(n, 1) - 1 and ps.dtype == torch.bool


def test_expand12_property_full_size():
    """2^22 parents -> 50 M children: undoing action k on child 12p+k gives parent p back."""
    from librubiks.cube import DeviceCubes
    n = 1 << 22
    g = torch.Generator(device="cuda").manual_seed(1)
    parents = DeviceCubes.solved(n)
    for _ in range(5):
        parents = parents.multi_rotate(torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g))
    kids = parents.expand12()
    undo = (torch.arange(12 * n, device="cuda") % 12).to(torch.uint8) ^ 1
    back = kids.multi_rotate(undo)
    expect = parents.soa[:, :n].repeat_interleave(12, dim=1)
    assert torch.equal(back.soa[:, :12 * n], expect)


# ---------------------------------------------------------------------------------------------
# is_solved
# ---------------------------------------------------------------------------------------------
def test_is_solved_golden(cube, golden):
    out = cube.multi_is_solved(golden["is_in"])
    assert out.dtype == bool and np.array_equal(out, golden["is_out"])
    assert cube.is_solved(cube.get_solved())
    assert not cube.is_solved(golden["mr_in"][0])


@pytest.mark.parametrize("n", SIZES)
def test_is_solved_flags_mask_count(n):
    from librubiks.cube import DeviceCubes
    rng = np.random.RandomState(n)
    s = reachable(n, 2, seed=n)   # depth 2: some rows come out solved
    plant = rng.rand(n) < 0.2
    s[plant] = oc.get_solved()
    near = s.copy()               # differs from solved only in the LAST plane -> must not count as solved
    expect = oc.multi_is_solved(s)
    cubes = DeviceCubes.from_numpy(s)
    assert np.array_equal(cubes.is_solved().cpu().numpy(), expect)
    mask, count = cubes.solved_mask()
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.array_equal(bits, expect)
    assert int(count.item()) == int(expect.sum())
    near[:, 19] = (near[:, 19] + 1) % 24
    assert not DeviceCubes.from_numpy(near).is_solved().any()


def test_is_solved_empty(cube):
    assert cube.multi_is_solved(np.empty((0, 20), dtype=np.int8)).shape == (0,)


# ---------------------------------------------------------------------------------------------
# as_oh
# ---------------------------------------------------------------------------------------------
def test_as_oh_golden(cube, golden):
    oh = cube.as_oh(golden["oh_in"])
    assert oh.dtype == torch.float32 and oh.shape == (256, 480) and oh.is_cuda
    oh = oh.cpu().numpy()
    assert np.array_equal(np.nonzero(oh)[1].reshape(256, 20), golden["oh_cols"])
    assert np.array_equal(oh[0], golden["oh_dense_row0"])
    single = cube.as_oh(golden["oh_in"][5])
    assert single.shape == (1, 480) and np.array_equal(single.cpu().numpy(), golden["oh_single"])
    # reference tests/test_cube.py:129-139: solved cube -> ones at 24*i + state[i]
    solved_oh = cube.as_oh(cube.get_solved()).cpu().numpy()
    expect = np.zeros((1, 480), dtype=np.float32)
    expect[0, 24 * np.arange(20) + cube.get_solved()] = 1
    assert np.array_equal(solved_oh, expect)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_as_oh_vs_oracle(n, dt):
    from librubiks.cube import DeviceCubes
    s = reachable(n, 20, seed=7 * n)
    out = torch.full((n, 480), float("nan"), dtype=dt, device="cuda")   # every element must be overwritten
    DeviceCubes.from_numpy(s).as_oh(out=out)
    assert np.array_equal(out.float().cpu().numpy(), oc.as_oh(s))


# ---------------------------------------------------------------------------------------------
# scrambling
# ---------------------------------------------------------------------------------------------
def test_scramble_golden_stream(cube, golden):
    for seed in (0, 42):
        for depth in (20, 24):
            np.random.seed(seed)
            for g in range(16):
                s, f, d = cube.scramble(depth, True)
                assert s.dtype == np.int8
                assert np.array_equal(s, golden[f"scr_s{seed}_d{depth}_states"][g])
                assert np.array_equal(f, golden[f"scr_s{seed}_d{depth}_faces"][g])
                assert np.array_equal(d, golden[f"scr_s{seed}_d{depth}_dirs"][g])
            np.random.seed(seed)
            cubes, f, d = cube.scramble_batch(16, depth, True)
            assert np.array_equal(cubes.numpy(), golden[f"scr_s{seed}_d{depth}_states"])
            assert np.array_equal(f, golden[f"scr_s{seed}_d{depth}_faces"])


def test_scramble_batch_redraw_matches_sequential_reference_order(cube):
    """Depth 2: about 1 in 12 scrambles comes out solved and must be redrawn before the next game draws."""
    np.random.seed(3)
    expect = [oc.scramble(2, True) for _ in range(300)]
    end_state = np.random.get_state()[1].copy()
    np.random.seed(3)
    cubes, faces, dirs = cube.scramble_batch(300, 2, True)
    assert np.array_equal(np.random.get_state()[1], end_state), "RNG stream position differs"
    assert np.array_equal(cubes.numpy(), np.array([e[0] for e in expect]))
    assert np.array_equal(faces, np.array([e[1] for e in expect]))
    assert np.array_equal(dirs, np.array([e[2] for e in expect]))
    assert not cubes.is_solved().any()


def test_scramble_inverse_solves(cube):
    np.random.seed(42)
    s, _, _ = cube.scramble(1)
    assert not cube.is_solved(s)
    s, faces, dirs = cube.scramble(20)
    for f, d in zip(reversed(faces), reversed(dirs)):
        s = cube.rotate(s, f, 1 - d)
    assert cube.is_solved(s)


def test_sequence_scrambler_golden(cube, golden):
    for ws in (True, False):
        np.random.seed(0)
        s, oh = cube.sequence_scrambler(8, 20, ws)
        assert np.array_equal(s, golden[f"seq_ws{int(ws)}_states"])
        assert oh.shape == (160, 480) and oh.dtype == torch.float32
        assert np.array_equal(np.nonzero(oh.cpu().numpy())[1].reshape(160, 20), golden[f"seq_ws{int(ws)}_ohcols"])


def test_sequence_scrambler_vs_oracle(cube):
    for games, depth, ws in ((1, 1, True), (3, 7, False), (300, 31, True), (1000, 5, False)):
        np.random.seed(games)
        exp_s, exp_oh = oc.sequence_scrambler(games, depth, ws)
        np.random.seed(games)
        s, oh = cube.sequence_scrambler(games, depth, ws)
        assert np.array_equal(s, exp_s)
        assert np.array_equal(oh.cpu().numpy(), exp_oh)


# ---------------------------------------------------------------------------------------------
# reference's own rotation known-answer sequence, through the drop-in API (tests/test_cube.py:45-92)
# ---------------------------------------------------------------------------------------------
def test_reference_known_answers(cube):
    from test_oracle_golden import AFTER_ALL12_NET, AFTER_F_NET, SOLVED_NET
    s = cube.get_solved()
    assert cube.stringify(s) == SOLVED_NET
    for (f, d), solved in zip(((0, 1), (0, 0), (0, 1), (1, 1), (2, 0), (3, 0)),
                              (False, True, False, False, False, False)):
        s = cube.rotate(s, f, d)
        assert cube.is_solved(s) == solved
    for (f, d), solved in zip(((3, 1), (2, 1), (1, 0), (0, 0)), (False, False, False, True)):
        s = cube.rotate(s, f, d)
        assert cube.is_solved(s) == solved
    assert cube.stringify(cube.rotate(cube.get_solved(), 0, 1)) == AFTER_F_NET
    s = cube.get_solved()
    for d in (0, 1):
        for f in range(6):
            s = cube.rotate(s, f, d)
    assert cube.stringify(s) == AFTER_ALL12_NET


# ---------------------------------------------------------------------------------------------
# layout round trip + raw C-ABI argument checking
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", SIZES)
def test_aos_soa_round_trip(n):
    from librubiks.cube import DeviceCubes
    s = reachable(n, 9, seed=n + 1)
    cubes = DeviceCubes.from_numpy(s)
    assert np.array_equal(cubes.soa[:, :n].cpu().numpy(), s.T)
    assert np.array_equal(cubes.numpy(), s)


def test_c_abi_argument_errors():
    from librubiks import _hip
    lib = _hip.lib()
    buf = torch.zeros(20 * 256, dtype=torch.int8, device="cuda")
    act = torch.zeros(256, dtype=torch.uint8, device="cuda")
    p, a = buf.data_ptr(), act.data_ptr()
    assert lib.rc_multi_rotate(None, a, p, 100, 256, 256, None) == -1          # RC_ERR_NULL
    assert lib.rc_multi_rotate(p + 4, a, p, 100, 256, 256, None) == -2         # RC_ERR_ALIGN
    assert lib.rc_multi_rotate(p, a, p, 100, 250, 256, None) == -2             # stride not multiple of 16
    assert lib.rc_multi_rotate(p, a, p, 300, 256, 256, None) == -3             # RC_ERR_STRIDE
    assert lib.rc_multi_rotate(p, a, p, 0, 0, 0, None) == 0                    # empty is a no-op
    assert lib.rc_expand12(p, p, 100, 256, 256, None) == -3                    # children need 1200 columns
    assert lib.rc_init(99) == -4                                               # RC_ERR_RANGE
    assert b"aligned" in ctypes.c_char_p(lib.rc_error_string(-2)).value


@pytest.mark.parametrize("n", [1, 2, 12, 13, 1200, 4097])
def test_small_call_row_major_kernels(n):
    """The one-launch row-major path of the stateless API (pinned host memory in and out) against the oracle."""
    from librubiks import cube
    from librubiks.cube import cube as cube_mod
    assert n <= cube_mod.SMALL_CALL
    rng = np.random.RandomState(n)
    states = np.tile(oc.get_solved(), (n, 1))
    for _ in range(25):
        states = oc.multi_rotate_actions(states, rng.randint(0, 12, n))
    states[n // 2] = oc.get_solved()
    acts = rng.randint(0, 12, n)
    faces, dirs = acts // 2, 1 - acts % 2
    before = states.copy()
    out = cube.multi_rotate(states, faces, dirs)
    assert out.dtype == np.int8 and out.shape == (n, 20) and np.array_equal(out, oc.multi_rotate(states, faces, dirs))
    out2 = cube.multi_rotate(out, faces, 1 - dirs)       # a second call reuses the staging buffers: `out` must be a copy
    assert np.array_equal(out2, states) and np.array_equal(out, oc.multi_rotate(states, faces, dirs))
    solved = cube.multi_is_solved(states)
    assert solved.dtype == bool and np.array_equal(solved, oc.multi_is_solved(states)) and solved[n // 2]
    oh = cube.as_oh(states)
    assert oh.shape == (n, 480) and oh.dtype == torch.float32 and oh.is_cuda
    assert np.array_equal(oh.cpu().numpy(), oc.as_oh(states))
    assert np.array_equal(states, before)                 # inputs are never mutated
    assert np.array_equal(cube.rotate(states[0], int(faces[0]), int(dirs[0])), oc.rotate(states[0], int(faces[0]), int(dirs[0])))
