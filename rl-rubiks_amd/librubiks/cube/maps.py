"""
Host-side constant data of the 20x24 representation.

The move tables themselves are built at compile time inside librubiks_hip.so
(csrc/rubiks_tables.h) and only *read back* here (`get_tensor_map`, same shape and meaning as the
reference's librubiks/cube/maps.py:107-145).  The sticker layout (reference maps.py:26-51) is only
needed by the printing helpers as633/stringify.
"""
import numpy as np


def get_move_lut() -> np.ndarray:
    """uint8[12, 2, 24]: code after action a, straight from the library (no GPU needed)."""
    from librubiks import _hip
    buf = np.empty(12 * 2 * 24, dtype=np.uint8)
    _hip.check(_hip.load().rc_get_move_table(buf.ctypes.data), "rc_get_move_table")
    return buf.reshape(12, 2, 24)


def get_tensor_map(dtype=np.int8) -> np.ndarray:
    """[dir(2), face(6), kind(2), 24] additive deltas, dir 1 = positive revolution."""
    lut = get_move_lut().astype(int)
    maps = np.empty((2, 6, 2, 24), dtype=dtype)
    for a in range(12):
        maps[1 - a % 2, a // 2] = lut[a] - np.arange(24)
    return maps


def get_solved_state(dtype=np.int8) -> np.ndarray:
    from librubiks import _hip
    buf = np.empty(20, dtype=np.int8)
    _hip.check(_hip.load().rc_get_solved(buf.ctypes.data), "rc_get_solved")
    return buf.astype(dtype)


def get_633maps(F, B, T, D, L, R):
    """(face,row,col) of every sticker of each corner / edge position, in 'right turn' order."""
    corners = (
        ((F, 0, 0), (L, 0, 2), (T, 2, 0)), ((F, 2, 0), (D, 0, 0), (L, 2, 2)),
        ((F, 2, 2), (R, 2, 0), (D, 0, 2)), ((F, 0, 2), (T, 2, 2), (R, 0, 0)),
        ((B, 0, 2), (T, 0, 0), (L, 0, 0)), ((B, 2, 2), (L, 2, 0), (D, 2, 0)),
        ((B, 2, 0), (D, 2, 2), (R, 2, 2)), ((B, 0, 0), (R, 0, 2), (T, 0, 2)),
    )
    edges = (
        ((F, 0, 1), (T, 2, 1)), ((F, 1, 0), (L, 1, 2)), ((F, 2, 1), (D, 0, 1)), ((F, 1, 2), (R, 1, 0)),
        ((T, 1, 0), (L, 0, 1)), ((D, 1, 0), (L, 2, 1)), ((D, 1, 2), (R, 2, 1)), ((T, 1, 2), (R, 0, 1)),
        ((B, 0, 1), (T, 0, 1)), ((B, 1, 2), (L, 1, 0)), ((B, 2, 1), (D, 2, 1)), ((B, 1, 0), (R, 1, 2)),
    )
    return corners, edges
