/*
 * rubiks_hip.h -- C ABI of librubiks_hip.so, the MI355X (gfx950) implementation of the
 * rl-rubiks cube-environment + search hot path.
 *
 * The reference (peleiden/rl-rubiks) is pure Python/NumPy and has no FFI of its own; these entry
 * points are what the Python shim `rl-rubiks_amd/librubiks/` binds with ctypes to implement the
 * reference's module API (INTEGRATION.md shows the binding).  Each entry point cites the reference
 * expression it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - Every pointer named *_soa / out / flags / mask / count / actions / moves is a DEVICE pointer
 *     (hipMalloc'ed; e.g. torch.Tensor.data_ptr()), 16-byte aligned.  No entry point allocates.
 *   - State layout in HBM is structure-of-arrays: plane j (0..19) of cube i lives at
 *     soa[j * stride + i]; `stride` is in bytes, a multiple of 16 and >= round_up(n, 16).
 *     Bytes of a plane beyond n are padding: kernels may overwrite them with unspecified values.
 *   - State bytes are the reference's codes: corner cubie j<8 -> 3*pos+ori, edge cubie j-8 ->
 *     2*pos+ori, all in 0..23 (librubiks/cube/cube.py:58-65).  Bytes outside 0..23 give
 *     unspecified results but never out-of-bounds accesses.
 *   - Actions are the reference's action indices 0..11: a = 2*face + (1 - direction)
 *     (`action_space`, librubiks/cube/cube.py:33-34).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     on that stream and safe to capture into a hipGraph.
 *   - Return value: 0 on success; RC_ERR_* (< 0) on argument errors; -(hipError_t) - 1000 when a
 *     HIP call fails.  rc_error_string() describes any of them.
 *   - Thread-safe for distinct streams.
 */
#ifndef RUBIKS_HIP_H
#define RUBIKS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 1

#define RC_OK 0
#define RC_ERR_NULL (-1)      /* required pointer is NULL */
#define RC_ERR_ALIGN (-2)     /* pointer or stride not 16-byte aligned */
#define RC_ERR_STRIDE (-3)    /* stride < round_up(n, 16) */
#define RC_ERR_RANGE (-4)     /* size / index argument out of range */
#define RC_ERR_NODEVICE (-5)  /* no gfx950 device / wrong architecture */
#define RC_ERR_HIP_BASE (-1000)

typedef void *rc_stream_t;

/* ---- library / host-only helpers ------------------------------------------------------------ */

int rc_abi_version(void);
const char *rc_error_string(int code);

/* Selects `device`, verifies it is gfx950.  Must be called once per process before device calls. */
int rc_init(int device);

/* Host-side copies of the constant tables the kernels use (no GPU needed):
 * out576[a*48 + kind*24 + v] = code v after action a  (v + maps[dir,face,kind,v],
 * librubiks/cube/maps.py:107-145 + librubiks/cube/cube.py:239). */
int rc_get_move_table(uint8_t *out576);
/* out20 = solved state (librubiks/cube/cube.py:58-65,73-74). */
int rc_get_solved(int8_t *out20);

/* ---- layout: the reference's (n,20) row-major arrays <-> SoA -------------------------------- */

int rc_aos_to_soa(const int8_t *aos, int8_t *soa, size_t n, size_t stride, rc_stream_t stream);
int rc_soa_to_aos(const int8_t *soa, int8_t *aos, size_t n, size_t stride, rc_stream_t stream);

/* ---- cube environment ------------------------------------------------------------------------ */

/* out[i] = move actions[i] applied to in[i].  Replaces _Cube2024.multi_rotate
 * (librubiks/cube/cube.py:256-263, dispatcher :49-52).  in and out may alias exactly.
 * Algorithmic HBM bytes: 41 per state. */
int rc_multi_rotate(const int8_t *in_soa, const uint8_t *actions, int8_t *out_soa, size_t n,
                    size_t stride_in, size_t stride_out, rc_stream_t stream);

/* children[12p + k] = action k applied to parents[p]  (parent-major, action-minor).  Replaces the
 * repeat/tile + multi_rotate idiom (librubiks/solving/agents.py:277-281,513; librubiks/train.py:285).
 * stride_c >= round_up(12 n_parents, 16).  Algorithmic HBM bytes: 260 per parent. */
int rc_expand12(const int8_t *parents_soa, int8_t *children_soa, size_t n_parents, size_t stride_p,
                size_t stride_c, rc_stream_t stream);

/* Solved test.  Replaces multi_is_solved (librubiks/cube/cube.py:85-89).  Any of the three
 * outputs may be NULL:
 *   flags[i]  = 1 if state i is solved else 0                (n bytes, padded to round_up(n,16))
 *   mask      = bit (i % 64) of word (i / 64) set iff solved  (round_up(n,64)/64 words... written
 *               as 16-bit pieces: buffer must hold round_up(n,16)/8 bytes)
 *   *count   += number of solved states                       (caller zeroes it)
 * Algorithmic HBM bytes: 20.125 per state with mask output, 21 with flags. */
int rc_is_solved(const int8_t *soa, uint8_t *flags, uint64_t *mask, uint32_t *count, size_t n,
                 size_t stride, rc_stream_t stream);

/* One-hot network input: out[i][24 j + s[i][j]] = 1, everything else 0; row-major (n, 480).
 * Replaces _Cube2024.as_oh (librubiks/cube/cube.py:265-277).  Every element of all n rows is
 * written (no pre-zeroing needed).  Algorithmic HBM bytes: 1940 (f32) / 980 (bf16) per state. */
int rc_as_oh_f32(const int8_t *soa, float *out, size_t n, size_t stride, rc_stream_t stream);
int rc_as_oh_bf16(const int8_t *soa, uint16_t *out, size_t n, size_t stride, rc_stream_t stream);

/* In place: for d in 0..depth-1: cube i <- action moves[d * n + i] applied to cube i.
 * Replaces the sequential rotate loop of scramble (librubiks/cube/cube.py:206-211) for n cubes at
 * once; the random draws stay on the host so the reference's RNG stream is preserved. */
int rc_apply_moves(int8_t *soa, const uint8_t *moves, size_t n, size_t stride, size_t depth,
                   rc_stream_t stream);

/* sequence_scrambler's state trajectory (librubiks/cube/cube.py:218-234): starting from `games`
 * solved cubes, out row g*depth + d holds game g after (d + 1 - with_solved) moves
 * (row g*depth is the solved cube when with_solved != 0); moves[d * games + g] as above.
 * out_soa has games*depth columns. */
int rc_sequence_states(const uint8_t *moves, int8_t *out_soa, size_t games, size_t depth,
                       int with_solved, size_t stride_out, rc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RUBIKS_HIP_H */
