"""
The reference's stateless cube API (librubiks/cube/cube.py:41-234) on MI355X.

Same names, argument meaning, return types and ownership rules: inputs are never mutated, every
call returns fresh arrays, NumPy (n,20) int8 in / out.  Every function that computes cube states
runs as a HIP kernel behind include/rubiks_hip.h: the arrays are transposed into the SoA HBM layout
at the boundary (DeviceCubes), so callers that care about throughput should stay on the device
(`DeviceCubes`, `scramble_batch`, the batched agents) instead of bouncing through NumPy per call.

Only the 20x24 representation exists (north_star scope); `set_is2024(False)` raises.
"""
import numpy as np
import torch

from librubiks import _hip, gpu
from librubiks.cube.device import DeviceCubes, N_PLANES, OH_WIDTH
from librubiks.cube.maps import get_633maps, get_solved_state

####################
# Action constants #   reference cube.py:30-35
####################
F, B, T, D, L, R = 0, 1, 2, 3, 4, 5
action_names = ('F', 'B', 'T', 'D', 'L', 'R')
action_space = [(a // 2, 1 - a % 2) for a in range(12)]   # index a <-> (face, direction); even a = positive
action_dim = len(action_space)

dtype = np.int8   # reference cube.py:73


def _actions_of(faces, directions) -> np.ndarray:
    """(face, direction) pairs -> action indices 0..11 (inverse of indices_to_actions)."""
    faces = np.asarray(faces).astype(np.int64)
    directions = np.asarray(directions).astype(np.int64)
    return (2 * faces + 1 - directions).astype(np.uint8)


SMALL_CALL = 1 << 15   # largest n served by the one-launch row-major kernels; beyond it the SoA path is faster


class _Staging:
    """
    Pinned host buffers for the small calls of the stateless API (rotate / multi_rotate / multi_is_solved / as_oh
    with a handful of states, as the reference's agents call them): the kernel reads the (n,20) bytes and writes its
    result straight through the mapping of pinned host memory into the device, so a call is one memcpy into the
    staging buffer, one launch and one stream synchronisation -- no allocation, no transposition, no DMA copies.
    """

    def __init__(self):
        self.cap = 0

    def get(self, n: int):
        if n > self.cap:
            self.cap = max(1024, 1 << int(np.ceil(np.log2(n))))
            self.states = torch.empty((self.cap, N_PLANES), dtype=torch.int8, pin_memory=True)
            self.out = torch.empty((self.cap, N_PLANES), dtype=torch.int8, pin_memory=True)
            self.actions = torch.empty(self.cap, dtype=torch.uint8, pin_memory=True)
            self.flags = torch.empty(self.cap, dtype=torch.uint8, pin_memory=True)
            self.np_states, self.np_out = self.states.numpy(), self.out.numpy()
            self.np_actions, self.np_flags = self.actions.numpy(), self.flags.numpy()
        return self


_staging = _Staging()


def _padded_actions(actions: np.ndarray, n: int) -> torch.Tensor:
    buf = np.zeros((n + 15) // 16 * 16, dtype=np.uint8)
    buf[:n] = actions
    return torch.from_numpy(buf).cuda()


################
# Rotate logic #   reference cube.py:41-52,244-263
################
def rotate(state: np.ndarray, face: int, direction: int) -> np.ndarray:
    """One move on one state; runs the row-major kernel with n = 1."""
    return multi_rotate(np.asarray(state)[None], np.array([face]), np.array([direction]))[0]


def multi_rotate(states: np.ndarray, faces: np.ndarray, directions: np.ndarray) -> np.ndarray:
    """Performs action (faces[i], directions[i]) on states[i]."""
    states = np.asarray(states)
    n = len(states)
    if n == 0:
        _hip.lib()
        return np.empty((0, N_PLANES), dtype=dtype)
    assert len(faces) == n and len(directions) == n
    if n <= SMALL_CALL:
        lib, st = _hip.lib(), _staging.get(n)
        st.np_states[:n] = states
        st.np_actions[:n] = _actions_of(faces, directions)
        stream = torch.cuda.current_stream()
        _hip.check(lib.rc_multi_rotate_aos(st.states.data_ptr(), st.actions.data_ptr(), st.out.data_ptr(), n, stream.cuda_stream),
                   "rc_multi_rotate_aos")
        stream.synchronize()
        return st.np_out[:n].copy()
    cubes = DeviceCubes.from_numpy(states)
    return cubes.multi_rotate(_padded_actions(_actions_of(faces, directions), n)).numpy()


#################
# Solving logic #   reference cube.py:58-89
#################
_solved2024 = None


def get_solved_instance() -> np.ndarray:
    """The module's own solved array -- read-only by convention, like the reference's."""
    global _solved2024
    if _solved2024 is None:
        _solved2024 = get_solved_state(dtype)
    return _solved2024


def get_solved() -> np.ndarray:
    return get_solved_instance().copy()


def is_solved(state: np.ndarray) -> bool:
    return bool(multi_is_solved(np.asarray(state)[None])[0])


def multi_is_solved(states: np.ndarray) -> np.ndarray:
    states = np.asarray(states)
    n = len(states)
    if n == 0:
        _hip.lib()
        return np.zeros(0, dtype=bool)
    if n <= SMALL_CALL:
        lib, st = _hip.lib(), _staging.get(n)
        st.np_states[:n] = states
        stream = torch.cuda.current_stream()
        _hip.check(lib.rc_is_solved_aos(st.states.data_ptr(), st.flags.data_ptr(), n, stream.cuda_stream), "rc_is_solved_aos")
        stream.synchronize()
        return st.np_flags[:n].astype(bool)
    return DeviceCubes.from_numpy(states).is_solved().cpu().numpy()


########################
# Representation logic #   reference cube.py:96-140
########################
_is2024 = True
_stored_repr = True


def set_is2024(is2024: bool):
    assert type(is2024) is bool
    if not is2024:
        raise NotImplementedError("only the 20x24 (int8 corner/edge) representation is implemented on MI355X")


def get_is2024():
    return _is2024


def store_repr():
    global _stored_repr
    _stored_repr = _is2024


def restore_repr():
    pass


def with_used_repr(fun):
    def wrapper(self, *args, **kwargs):
        set_is2024(getattr(self, "is2024", True))
        return fun(self, *args, **kwargs)
    return wrapper


def shape():
    return (N_PLANES,)


def as_correct(t):
    """Reference cube.py:135-137: the conv net's input form, defined for the 6x8x6 representation only."""
    raise NotImplementedError("as_correct belongs to the 6x8x6 representation, which is out of scope on MI355X")


def get_oh_shape() -> int:
    return OH_WIDTH


def as_oh(states: np.ndarray) -> torch.Tensor:
    """n states -> (n,480) float32 one-hot on `librubiks.gpu`; a single state gives (1,480)."""
    states = np.asarray(states)
    if states.ndim == 1:
        states = states[None]
    n = len(states)
    if n == 0:
        _hip.lib()
        return torch.zeros((0, OH_WIDTH), device=gpu)
    if n <= SMALL_CALL:   # one launch: the kernel reads the states from pinned host memory and writes every output element
        lib, st = _hip.lib(), _staging.get(n)
        st.np_states[:n] = states
        out = torch.empty((n, OH_WIDTH), dtype=torch.float32, device=gpu)
        stream = torch.cuda.current_stream()
        _hip.check(lib.rc_as_oh_aos_f32(st.states.data_ptr(), out.data_ptr(), n, stream.cuda_stream), "rc_as_oh_aos_f32")
        stream.synchronize()   # the staging buffer is free for the next call once the kernel has read it
        return out
    return DeviceCubes.from_numpy(states).as_oh(torch.float32)


def repeat_state(state: np.ndarray, n: int = action_dim) -> np.ndarray:
    return np.tile(state, (n, 1))


################
# Action logic #   reference cube.py:179-200
################
def iter_actions(n: int = 1) -> np.ndarray:
    """uint8[2, 12n]: faces row and directions row of the 12 actions tiled n times."""
    faces = np.tile(np.repeat(np.arange(6, dtype=np.uint8), 2), n)
    dirs = np.tile(np.array([1, 0], dtype=np.uint8), 6 * n)
    return np.stack([faces, dirs])


def indices_to_actions(indices: np.ndarray):
    faces = indices // 2
    dirs = 1 - indices % 2
    return faces, dirs


def rev_action(action: int) -> int:
    return action ^ 1


def rev_actions(actions: np.ndarray) -> np.ndarray:
    return np.asarray(actions) ^ 1


##################
# Scramble logic #   reference cube.py:206-234
##################
_SCRAMBLE_PASS = 8192   # games drawn per device launch


def _moves_tensor(actions_dn: np.ndarray, stride: int) -> torch.Tensor:
    """(depth, n) action indices -> (depth, stride) uint8 device tensor (padding cubes get action 0)."""
    depth, n = actions_dn.shape
    buf = np.zeros((depth, stride), dtype=np.uint8)
    buf[:, :n] = actions_dn
    return torch.from_numpy(buf).cuda()


def scramble_batch(games: int, depth, force_not_solved: bool = False):
    """
    `games` scrambles, bit-identical to calling the reference's scramble(depth, force_not_solved)
    `games` times in a row: the draws come from the legacy global np.random stream in the reference's
    order (per game: faces, then directions; a scramble that comes out solved is redrawn before the
    next game draws, cube.py:213-214), while all moves are applied by one rc_apply_moves launch.
    `depth` is an int, or a callable drawing one game's depth from np.random right before that game's
    moves are drawn (the Evaluator's deep mode, evaluation.py:73-74).
    Returns (DeviceCubes, faces int[games, max depth], dirs int[games, max depth]); with per-game
    depths the unused tail of a row is -1 (faces) / 0 (dirs).
    """
    _hip.lib()
    sampler = depth if callable(depth) else None
    depths = np.zeros(games, dtype=np.int64) if sampler else np.full(games, int(depth), dtype=np.int64)
    width = 999 if sampler else int(depth)
    faces = np.full((games, width), -1 if sampler else 0, dtype=np.int64)
    dirs = np.zeros((games, width), dtype=np.int64)
    cubes = DeviceCubes.solved(games)
    if games == 0 or width == 0:
        return cubes, faces, dirs
    start, redraw = 0, False

    def draw(lo, hi, keep_first_depth):
        for g in range(lo, hi):
            if sampler and not (keep_first_depth and g == lo):   # a redrawn game keeps its depth (the recursion is inside scramble)
                depths[g] = sampler()
            d = depths[g]
            faces[g, :d] = np.random.randint(6, size=(d,))
            dirs[g, :d] = np.random.randint(2, size=(d,))

    while start < games:
        stop = min(games, start + _SCRAMBLE_PASS)
        # RNG state at the start of the pass: if game g must be redrawn, the stream position right after its first
        # draw is recovered by replaying the draws of start..g from here (rare), instead of snapshotting the
        # 624-word state after every game
        state0 = np.random.get_state() if force_not_solved else None
        draw(start, stop, redraw)
        # moves beyond a game's own depth are action 12: identity padding of the kernels' move table
        acts = np.where(np.arange(width)[None, :] < depths[start:stop, None],
                        _actions_of(np.maximum(faces[start:stop], 0), dirs[start:stop]), 12).astype(np.uint8)
        acts = acts[:, :max(1, int(depths[start:stop].max()))]
        part = DeviceCubes.solved(stop - start)
        part.apply_moves(_moves_tensor(acts.T, part.stride))
        solved = part.is_solved().cpu().numpy() & (depths[start:stop] != 0) if force_not_solved \
            else np.zeros(stop - start, dtype=bool)
        first = int(np.argmax(solved)) if solved.any() else stop - start
        cubes.soa[:, start:start + first] = part.soa[:, :first]
        was_redraw, redraw = redraw, first < stop - start
        if redraw:
            # game start+first came out solved: the reference redraws it from the stream position
            # right after its first draw, and every later game follows that (cube.py:213-214)
            np.random.set_state(state0)
            draw(start, start + first + 1, was_redraw)
        start += first
    if sampler:
        w = max(1, int(depths.max()))
        faces, dirs = faces[:, :w], dirs[:, :w]
    return cubes, faces, dirs


def scramble(depth: int, force_not_solved=False):
    """(state int8[20], faces, dirs) exactly as the reference returns them."""
    cubes, faces, dirs = scramble_batch(1, depth, force_not_solved)
    return cubes.numpy()[0], faces[0], dirs[0]


def sequence_scrambler_device(games: int, depth: int, with_solved: bool) -> DeviceCubes:
    """Device-resident trajectory: column g*depth + d = game g after its d-th recorded state."""
    lib = _hip.lib()
    faces = np.random.randint(0, 6, (depth, games))
    dirs = np.random.randint(0, 2, (depth, games))
    out = DeviceCubes.empty(games * depth)
    if games * depth:
        moves = torch.from_numpy(np.ascontiguousarray(_actions_of(faces, dirs))).cuda()
        _hip.check(lib.rc_sequence_states(moves.data_ptr(), out.soa.data_ptr(), games, depth, int(bool(with_solved)),
                                          out.stride, _hip.stream_ptr()), "rc_sequence_states")
    return out


def sequence_scrambler(games: int, depth: int, with_solved: bool):
    """(int8[games*depth, 20] game-major states, float32 one-hot[games*depth, 480] on gpu)."""
    cubes = sequence_scrambler_device(games, depth, with_solved)
    return cubes.numpy(), cubes.as_oh(torch.float32)


############
# Printing #   reference cube.py:149-173,279-307 (host only; not on the hot path)
############
_corner_633map, _side_633map = get_633maps(F, B, T, D, L, R)


def as633(state: np.ndarray) -> np.ndarray:
    """Sticker colours int[6,3,3], faces in order F, B, T, D, L, R."""
    net = np.repeat(np.arange(6), 9).reshape(6, 3, 3)
    for i in range(8):
        pos, ori = divmod(int(state[i]), 3)
        shift = -ori if pos in (0, 2, 5, 7) else ori   # these positions are listed with the other handedness
        for where, colour in zip(_corner_633map[pos], np.roll([s[0] for s in _corner_633map[i]], shift)):
            net[where] = colour
    for i in range(12):
        pos, ori = divmod(int(state[i + 8]), 2)
        for where, colour in zip(_side_633map[pos], np.roll([s[0] for s in _side_633map[i]], ori)):
            net[where] = colour
    return net


def as69(state: np.ndarray) -> np.ndarray:
    return as633(state).reshape((6, 9))


def stringify(state: np.ndarray) -> str:
    net = as633(state)
    canvas = np.full((9, 12), " ", dtype="<U1")
    for face, (br, bc) in {T: (0, 1), L: (1, 0), F: (1, 1), R: (1, 2), B: (1, 3), D: (2, 1)}.items():
        canvas[3 * br:3 * br + 3, 3 * bc:3 * bc + 3] = net[face].astype(str)
    return "\n".join(" ".join(row) for row in canvas)


class Cube:
    """Namespace alias: BASELINE's north_star calls the API `librubiks.cube.Cube`; every caller in the
    reference uses the module functions, which this class simply re-exports as static methods."""
    rotate = staticmethod(rotate)
    multi_rotate = staticmethod(multi_rotate)
    is_solved = staticmethod(is_solved)
    multi_is_solved = staticmethod(multi_is_solved)
    as_oh = staticmethod(as_oh)
    get_solved = staticmethod(get_solved)
    scramble = staticmethod(scramble)
    scramble_batch = staticmethod(scramble_batch)
    sequence_scrambler = staticmethod(sequence_scrambler)
    expand12 = staticmethod(lambda states: DeviceCubes.from_numpy(states).expand12().numpy())
    action_space = action_space
    action_dim = action_dim
