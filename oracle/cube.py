"""
NumPy restatement of the reference's 20x24 cube environment (TEST INFRASTRUCTURE, see oracle/__init__.py).

State: int8[20].  Entry i < 8 is corner cubie i coded 3*position + orientation, entry 8+i is edge
cubie i coded 2*position + orientation; every entry lies in 0..23 (librubiks/cube/cube.py:58-65,
librubiks/cube/maps.py:101-105).

Pinned against the imported reference by tests/golden/*.npz (tests/test_oracle_golden.py).
"""
import numpy as np

# ---------------------------------------------------------------------------------------------
# Action constants (librubiks/cube/cube.py:30-35)
# ---------------------------------------------------------------------------------------------
F, B, T, D, L, R = range(6)
FACE_NAMES = ("F", "B", "T", "D", "L", "R")
N_ACTIONS = 12
# action index a <-> (face, direction): a = 2*face + (1 - direction); even a = positive turn.
ACTION_SPACE = [(a // 2, 1 - a % 2) for a in range(N_ACTIONS)]
STATE_DTYPE = np.int8
N_CUBIES = 20
N_CORNERS = 8
OH_WIDTH = 480
# 0 for the eight corner slots, 1 for the twelve edge slots (librubiks/cube/cube.py:240)
KIND = np.array([0] * 8 + [1] * 12)

# ---------------------------------------------------------------------------------------------
# Move definition data (librubiks/cube/maps.py:74-98).  Per face, in positive revolution:
#   corner 4-cycle, edge 4-cycle, the corner orientation that is preserved (the other two swap),
#   whether edge orientation flips.
# ---------------------------------------------------------------------------------------------
_FACE_MOVES = {
    F: ((0, 1, 2, 3), (0, 1, 2, 3), 0, False),
    B: ((4, 7, 6, 5), (8, 11, 10, 9), 0, False),
    T: ((0, 3, 7, 4), (0, 7, 8, 4), 1, True),
    D: ((1, 5, 6, 2), (2, 5, 10, 6), 1, True),
    L: ((0, 4, 5, 1), (1, 4, 9, 5), 2, False),
    R: ((7, 3, 2, 6), (3, 6, 11, 7), 2, False),
}


def _positive_permutations():
    """perm[face, kind, code] = code after a positive quarter turn of `face` (maps.py:120-139)."""
    perm = np.tile(np.arange(24), (6, 2, 1))
    for face, (ccyc, ecyc, keep, flip) in _FACE_MOVES.items():
        for j in range(4):
            c_from, c_to = ccyc[j], ccyc[(j + 1) % 4]
            for ori in range(3):
                # the orientation equal to `keep` stays, the other two trade places (maps.py:128)
                new_ori = ori if ori == keep else 3 - keep - ori
                perm[face, 0, 3 * c_from + ori] = 3 * c_to + new_ori
            e_from, e_to = ecyc[j], ecyc[(j + 1) % 4]
            for ori in range(2):
                new_ori = ori ^ int(flip)  # maps.py:135
                perm[face, 1, 2 * e_from + ori] = 2 * e_to + new_ori
    return perm


def move_deltas() -> np.ndarray:
    """
    int8[2(dir), 6(face), 2(kind), 24] such that new = code + deltas[dir, face, kind, code];
    dir 1 = positive, dir 0 = its inverse (librubiks/cube/maps.py:107-145, `get_tensor_map`).
    """
    pos = _positive_permutations()
    neg = np.empty_like(pos)
    idx = np.arange(24)
    for f in range(6):
        for k in range(2):
            neg[f, k, pos[f, k]] = idx  # inverse permutation (maps.py:132,139)
    return np.stack([neg - idx, pos - idx]).astype(STATE_DTYPE)


_DELTAS = move_deltas()


def move_lut() -> np.ndarray:
    """uint8[12(action), 2(kind), 24]: LUT[a, kind, code] = code after action a."""
    lut = np.empty((N_ACTIONS, 2, 24), dtype=np.uint8)
    for a, (face, d) in enumerate(ACTION_SPACE):
        lut[a] = np.arange(24) + _DELTAS[d, face]
    return lut


_LUT = move_lut()

# ---------------------------------------------------------------------------------------------
# Solved state (librubiks/cube/cube.py:58-65,73-83)
# ---------------------------------------------------------------------------------------------
_SOLVED = np.concatenate([3 * np.arange(8), 2 * np.arange(12)]).astype(STATE_DTYPE)


def get_solved() -> np.ndarray:
    return _SOLVED.copy()


# ---------------------------------------------------------------------------------------------
# Rotation (librubiks/cube/cube.py:244-263)
# ---------------------------------------------------------------------------------------------
def rotate(state: np.ndarray, face: int, direction: int) -> np.ndarray:
    """One move on one state (cube.py:244-254): per-slot delta lookup, out of place."""
    d = _DELTAS[direction, face]
    return state + d[KIND, state]


def multi_rotate(states: np.ndarray, faces: np.ndarray, directions: np.ndarray) -> np.ndarray:
    """
    Move (faces[i], directions[i]) on states[i] (cube.py:256-263).  Same algorithmic form as the
    reference (materialise the per-state (2,24) delta block, then one fancy-index gather), so it is
    also what bench.py times as the CPU baseline of this op.
    """
    n = len(states)
    per_state = _DELTAS[directions, faces]                      # (n, 2, 24)
    rows = np.repeat(np.arange(n), N_CUBIES)
    kinds = np.tile(KIND, n)
    gathered = per_state[rows, kinds, states.ravel()].reshape(n, N_CUBIES)
    return states + gathered


def multi_rotate_actions(states: np.ndarray, actions: np.ndarray) -> np.ndarray:
    """LUT form: out[i, j] = LUT[a_i, kind(j), s[i, j]]; used to cross-check."""
    a = np.asarray(actions).astype(np.int64)
    return _LUT[a[:, None], KIND[None, :], states.astype(np.int64)].astype(STATE_DTYPE)


def expand12(states: np.ndarray) -> np.ndarray:
    """
    All 12 children of every state, parent-major / action-minor: child k of parent p is row 12p+k
    (librubiks/solving/agents.py:277-281, librubiks/train.py:285).
    """
    n = len(states)
    faces, dirs = iter_actions(n)
    return multi_rotate(np.repeat(states, N_ACTIONS, axis=0), faces, dirs)


# ---------------------------------------------------------------------------------------------
# Solved test (librubiks/cube/cube.py:85-89)
# ---------------------------------------------------------------------------------------------
def is_solved(state: np.ndarray) -> bool:
    return bool((state == _SOLVED).all())


def multi_is_solved(states: np.ndarray) -> np.ndarray:
    return (states == _SOLVED).all(axis=1)


# ---------------------------------------------------------------------------------------------
# One-hot (librubiks/cube/cube.py:265-277)
# ---------------------------------------------------------------------------------------------
def oh_indices(states: np.ndarray) -> np.ndarray:
    """Column of the single 1 of each 24-wide block: 24*j + s[.., j] (cube.py:242,270,274)."""
    return np.arange(N_CUBIES) * 24 + states


def as_oh(states: np.ndarray) -> np.ndarray:
    """float32[(1|n), 480] with exactly 20 ones per row (cube.py:265-277); NumPy, not torch."""
    states = np.atleast_2d(states)
    oh = np.zeros((len(states), OH_WIDTH), dtype=np.float32)
    oh[np.repeat(np.arange(len(states)), N_CUBIES), oh_indices(states).ravel()] = 1
    return oh


# ---------------------------------------------------------------------------------------------
# Action helpers (librubiks/cube/cube.py:142-147,179-200)
# ---------------------------------------------------------------------------------------------
def repeat_state(state: np.ndarray, n: int = N_ACTIONS) -> np.ndarray:
    return np.tile(state, (n, 1))


def iter_actions(n: int = 1) -> np.ndarray:
    """uint8[2, 12n]: row 0 faces, row 1 directions, the 12 actions tiled n times (cube.py:179-184)."""
    faces = np.tile(np.repeat(np.arange(6), 2), n)
    dirs = np.tile(np.array([1, 0] * 6), n)
    return np.stack([faces, dirs]).astype(np.uint8)


def indices_to_actions(indices: np.ndarray):
    """(faces, dirs) of action indices (cube.py:186-192); dirs computed the way the reference does."""
    return indices // 2, ~(indices % 2) + 2


def rev_action(action: int) -> int:
    return action ^ 1  # cube.py:194-195


def rev_actions(actions: np.ndarray) -> np.ndarray:
    return np.asarray(actions) ^ 1  # cube.py:197-200


# ---------------------------------------------------------------------------------------------
# Scrambling (librubiks/cube/cube.py:206-234).  RNG call order is part of the contract:
# faces are drawn before directions, from the legacy global np.random stream.
# ---------------------------------------------------------------------------------------------
def scramble(depth: int, force_not_solved: bool = False):
    faces = np.random.randint(6, size=(depth,))
    dirs = np.random.randint(2, size=(depth,))
    state = get_solved()
    for f, d in zip(faces, dirs):
        state = rotate(state, f, d)
    if force_not_solved and depth != 0 and is_solved(state):
        return scramble(depth, True)  # consumes further draws, like the reference (cube.py:213-214)
    return state, faces, dirs


def sequence_scrambler(games: int, depth: int, with_solved: bool):
    """
    States visited by `games` independent scrambles, game-major: row g*depth + d (cube.py:218-234).
    Returns (int8[games*depth, 20], float32 one-hot[games*depth, 480]).
    """
    cur = np.tile(_SOLVED, (games, 1))
    faces = np.random.randint(0, 6, (depth, games))
    dirs = np.random.randint(0, 2, (depth, games))
    seq = [cur] if with_solved else []
    for d in range(depth - int(with_solved)):
        cur = multi_rotate(cur, faces[d], dirs[d])
        seq.append(cur)
    states = np.stack(seq, axis=1).reshape(games * depth, N_CUBIES)
    return states, as_oh(states)


# ---------------------------------------------------------------------------------------------
# Sticker view, used only by the known-answer nets of the reference's tests
# (librubiks/cube/maps.py:26-51, librubiks/cube/cube.py:149-173,279-307)
# ---------------------------------------------------------------------------------------------
# (face, row, col) of the three stickers of each corner position / two stickers of each edge position
_CORNER_STICKERS = (
    ((F, 0, 0), (L, 0, 2), (T, 2, 0)), ((F, 2, 0), (D, 0, 0), (L, 2, 2)),
    ((F, 2, 2), (R, 2, 0), (D, 0, 2)), ((F, 0, 2), (T, 2, 2), (R, 0, 0)),
    ((B, 0, 2), (T, 0, 0), (L, 0, 0)), ((B, 2, 2), (L, 2, 0), (D, 2, 0)),
    ((B, 2, 0), (D, 2, 2), (R, 2, 2)), ((B, 0, 0), (R, 0, 2), (T, 0, 2)),
)
_EDGE_STICKERS = (
    ((F, 0, 1), (T, 2, 1)), ((F, 1, 0), (L, 1, 2)), ((F, 2, 1), (D, 0, 1)), ((F, 1, 2), (R, 1, 0)),
    ((T, 1, 0), (L, 0, 1)), ((D, 1, 0), (L, 2, 1)), ((D, 1, 2), (R, 2, 1)), ((T, 1, 2), (R, 0, 1)),
    ((B, 0, 1), (T, 0, 1)), ((B, 1, 2), (L, 1, 0)), ((B, 2, 1), (D, 2, 1)), ((B, 1, 0), (R, 1, 2)),
)


def as633(state: np.ndarray) -> np.ndarray:
    """int[6,3,3] sticker colours, face order F,B,T,D,L,R (cube.py:279-307)."""
    net = np.empty((6, 3, 3), dtype=int)
    for f in range(6):
        net[f] = f
    for i in range(8):
        pos, ori = divmod(int(state[i]), 3)
        if pos in (0, 2, 5, 7):  # these positions list their stickers in the other handedness (cube.py:292-293)
            ori = -ori
        colours = np.roll([s[0] for s in _CORNER_STICKERS[i]], ori)
        for sticker, colour in zip(_CORNER_STICKERS[pos], colours):
            net[sticker] = colour
    for i in range(12):
        pos, ori = divmod(int(state[8 + i]), 2)
        colours = np.roll([s[0] for s in _EDGE_STICKERS[i]], ori)
        for sticker, colour in zip(_EDGE_STICKERS[pos], colours):
            net[sticker] = colour
    return net


def stringify(state: np.ndarray) -> str:
    """9x12 character cross layout (cube.py:160-173)."""
    net = as633(state)
    canvas = np.full((9, 12), " ", dtype="<U1")
    # block row / block column of each face in the unfolded cross:  . T . . / L F R B / . D . .
    placement = {T: (0, 1), L: (1, 0), F: (1, 1), R: (1, 2), B: (1, 3), D: (2, 1)}
    for f, (br, bc) in placement.items():
        canvas[3 * br:3 * br + 3, 3 * bc:3 * bc + 3] = net[f].astype(str)
    return "\n".join(" ".join(row) for row in canvas)
