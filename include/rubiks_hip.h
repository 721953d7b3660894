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

#define RC_ABI_VERSION 10

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

/* ---- small calls on the reference's own layout ------------------------------------------------
 * The same three operations on (n,20) row-major int8 states, for the sizes at which the reference's callers use the
 * stateless functions (n = 1 .. a few thousand: librubiks/solving/agents.py:109,513).  One launch each, no SoA
 * staging; every pointer may be device memory OR pinned host memory mapped into the device (hipHostMalloc), so a
 * NumPy caller pays one launch and one stream synchronisation per call.  No alignment requirement.
 *   rc_multi_rotate_aos  out[i] = move actions[i] on in[i]      (librubiks/cube/cube.py:49-52,256-263)
 *   rc_is_solved_aos     flags[i] = in[i] == solved             (librubiks/cube/cube.py:85-89)
 *   rc_as_oh_aos_f32     out[i, 24 j + in[i, j]] = 1, else 0    (librubiks/cube/cube.py:130-133,265-277) */
int rc_multi_rotate_aos(const int8_t *in_aos, const uint8_t *actions, int8_t *out_aos, size_t n, rc_stream_t stream);
int rc_is_solved_aos(const int8_t *in_aos, uint8_t *flags, size_t n, rc_stream_t stream);
int rc_as_oh_aos_f32(const int8_t *in_aos, float *out, size_t n, rc_stream_t stream);

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
/* rc_expand12 that also answers multi_is_solved (cube.py:85-89) for the parents and for every child in the same launch -- the
 * three calls one data-generation step of an ADI rollout makes (train.py:285-296).  The flags come from the staged parents
 * (child k of p is solved iff p is the solved cube turned by rev(k)); no child is read back.
 *   parent_solved: uint8[round_up(n_parents, 16)]   child_solved: uint8[12 round_up(n_parents, 16)]   (1 = solved; 16-byte aligned;
 *   entries behind n_parents / 12 n_parents are padding).  Algorithmic HBM bytes: 273 per parent. */
int rc_expand12_flags(const int8_t *parents_soa, int8_t *children_soa, size_t n_parents, size_t stride_p, size_t stride_c,
                      uint8_t *parent_solved, uint8_t *child_solved, rc_stream_t stream);

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

/* ---- network input layer fused with the one-hot encoding ------------------------------------------
 * out[i][c] = act( bias[c] + sum_j W1[c][24 j + s[i][j]] )   (bf16 out, fp32 accumulation)
 * = activation(Linear(480, H)(as_oh(states))) of the reference (librubiks/cube/cube.py:265-277 feeding
 * the first nn.Linear of librubiks/model.py:123-127,150-157) on the matrix cores, without materialising the
 * one-hot matrix: the one-hot A-fragments of v_mfma_f32_32x32x16_{f16,bf16} are generated in registers from the
 * cube codes and a 128-column slice of W1 is resident in LDS as the B operand.
 *   w1  : [H][480] row-major = the nn.Linear weight as stored, IEEE half if table_is_f16 != 0 (11 mantissa bits:
 *         the default whenever every weight fits half's range), else bf16
 *   bias: float[H], out: bf16 [n][H];  H: multiple of 128;  activation: 0 = none, 1 = ReLU, 2 = ELU(alpha)
 * Algorithmic HBM bytes per state: 20 in + 2 H out (W1 is 0.96 H KB, L2-resident). */
#define RC_ACT_NONE 0
#define RC_ACT_RELU 1
#define RC_ACT_ELU 2
int rc_first_layer_mfma_bf16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w1, const float *bias,
                             uint16_t *out, size_t H, int activation, float alpha, int table_is_f16, rc_stream_t stream);

/* In-place ReLU / ELU(alpha) of a contiguous bf16 tensor of n elements (n % 8 == 0): the activation pass between
 * two library GEMMs (model.py:150-157: Linear -> activation), 16 bytes per lane.  4 B of HBM traffic per element. */
int rc_act_bf16_inplace(uint16_t *x, size_t n, int activation, float alpha, rc_stream_t stream);

/* A hidden layer of the bf16 engine as ONE kernel: out = act(a w^T + bias) in bf16, fp32 accumulation on MFMA
 * (model.py:123-127,150-157).  a: bf16 [n_rows][k], w: bf16 [n_out][k] (nn.Linear layout), bias: float[n_out], out: bf16
 * [n_rows][n_out].  The kernel of rc_split_gemm_f16 (below) with one product instead of three: 352 x 256 tiles (tile 1,
 * n_out % 256 == 0) or 352 x 128 (tile 3, n_out % 128 == 0), tile 0 = choose; k % 64 == 0.  Replaces the library GEMM +
 * rc_act_bf16_inplace where it is the faster of the two (librubiks/model.py::InferenceNet decides by shape). */
int rc_gemm_bias_act_bf16(const uint16_t *a, const uint16_t *w, const float *bias, size_t n_rows, size_t n_out, size_t k,
                          int activation, float alpha, uint16_t *out, int tile, rc_stream_t stream);

/* ---- fp32-accurate network on the f16 matrix cores (f16x3 split) -------------------------------------------------
 * A float x travels as two IEEE halves, x = hi + lo * 2^-11 with hi = half(x), lo = half((x - hi) * 2^11); a layer is
 *   y = act(hi_x W_hi^T + 2^-11 (hi_x W_lo^T + lo_x W_hi^T) + b)      (fp32 accumulation on MFMA, library GEMMs)
 * which is closer to the float64 result than an fp32 GEMM (librubiks/model.py::SplitF32Net; replaces the fp32 forward
 * of librubiks/model.py:131-141).  These two kernels build the GEMM operands:
 *   rc_oh_split_f16   out[r] = [onehot(r), onehot(r) * 2^-11] as IEEE half, row pitch 960 (the one-hot is exact in half:
 *                     its product with [W_hi | W_lo] IS the input layer)                      (cube.py:265-277)
 *   rc_split_act_f16  y = act(c + corr_scale * c_corr + bias) per element of the fp32 GEMM outputs c, c_corr
 *                     [n_rows][n_cols] (c_corr may be NULL); writes out_hi_lo[r] = [hi(y row), lo(y row)] (row pitch
 *                     2 n_cols, may be NULL) and / or y itself to out_f32 (may be NULL).  n_cols % 8 == 0; ELU uses expm1f. */
int rc_oh_split_f16(const int8_t *soa, size_t n, size_t stride, uint16_t *out, rc_stream_t stream);
/* The whole input layer of that network in one kernel on the matrix cores, straight from the cube states:
 *   out_hi_lo[r] = [hi(y), lo(y)],  y = act(onehot(r) w_hi^T + 2^-11 onehot(r) w_lo^T + bias)      (row pitch 2 H halves)
 * w_hi, w_lo: IEEE half [H][480] row-major (the nn.Linear weight split as above), H % 64 == 0.  Replaces rc_oh_split_f16 +
 * the K = 960 GEMM + rc_split_act_f16 for the first layer (cube.py:265-277 + model.py:123-127,150-157 at fp32 accuracy). */
int rc_first_layer_split_f16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w_hi, const uint16_t *w_lo,
                             const float *bias, uint16_t *out_hi_lo, size_t H, int activation, float alpha, rc_stream_t stream);
/* The same with the half-range flag of rc_split_layer_t: *range_flag |= 1 when an output is beyond +-65504 or not finite. */
int rc_first_layer_split_flag_f16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w_hi, const uint16_t *w_lo,
                                  const float *bias, uint16_t *out_hi_lo, size_t H, int activation, float alpha,
                                  int32_t *range_flag, rc_stream_t stream);
/* The input layer as a sum of rows: a one-hot state times W is the sum of the 20 rows of W^T its cubies select,
 *   out_hi_lo[r] = [hi(y), lo(y)],  y = act(bias + sum_{j < 20} w_rows[24 j + code_j(r)])          (row pitch 2 H halves)
 * in fp32, in the order j = 0 .. 19 behind the bias (one order whatever the batch).  w_rows: float [480][H] = the nn.Linear
 * weight transposed, H % 64 == 0.  20 additions per output instead of 960 multiply-adds: what librubiks.model.SplitF32Net
 * runs (cube.py:265-277 + model.py:123-127,150-157); range_flag as above (may be NULL). */
int rc_first_layer_gather_f16(const int8_t *soa, size_t n, size_t stride, const float *w_rows, const float *bias,
                              uint16_t *out_hi_lo, size_t H, int activation, float alpha, int32_t *range_flag, rc_stream_t stream);
int rc_split_act_f16(const float *c, const float *c_corr, float corr_scale, size_t n_rows, size_t n_cols, const float *bias,
                     int activation, float alpha, uint16_t *out_hi_lo, float *out_f32, rc_stream_t stream);

/* The same layer with its K loop cut in two for layers too narrow to fill the chip with 352 x 256 tiles: twice the workgroups,
 * raw fp32 accumulators out_partials[2][n_rows][n_out] with  y = act(out_partials[1] + 2^-11 * out_partials[0] + bias)  left to
 * the consumer (rc_head_split_f32 / rc_split_act_f16 take them as c = out_partials[1], c_corr = out_partials[0]).
 * n_out % 256 == 0, k % 128 == 0. */
int rc_split_gemm_partials_f16(const uint16_t *a_hi_lo, const uint16_t *w_lo_hi_hi, size_t n_rows, size_t n_out, size_t k,
                               float *out_partials, rc_stream_t stream);

/* Last hidden activation + output layer of that network in one pass (the fp32 counterpart of rc_head_bf16):
 *   y = act(c + corr_scale * c_corr + bias_h)  (never written),   out[i][o] = bias_o[o] + sum_k w[o][k] * y[i][k],  o < n_out <= 16
 * c, c_corr: the fp32 GEMM outputs [n][K] (c_corr may be NULL), K = 512 or 1024;  w: float [n_out][K];  out: float, row pitch 16
 * (columns >= n_out are written as 0): the layout rc_mcts_backup_head reads.  fp32 FMA chains on the exact fp32 MFMA.
 * Replaces rc_split_act_f16's fp32 output + the 13-wide fp32 library GEMM (model.py:124-129,150-159 at fp32 accuracy). */
int rc_head_split_f32(const float *c, const float *c_corr, float corr_scale, size_t n, size_t K, const float *bias_h,
                      int activation, float alpha, const float *w, const float *bias_o, size_t n_out, float *out, rc_stream_t stream);
/* One hidden layer of that network as ONE kernel (own MFMA GEMM, fused epilogue; csrc/rubiks_gemm.hip):
 *   y = act(hi_x W_hi^T + 2^-11 (hi_x W_lo^T + lo_x W_hi^T) + bias)          (model.py:123-127,150-157 at fp32 accuracy)
 * a_hi_lo: [n_rows][2 k] halves (hi | lo), as the kernels above write it;  w_lo_hi_hi: [n_out][3 k] halves, the layer's
 * weight split and laid out [W_lo | W_hi | W_hi] (the order the K loop walks: both correction products, a 2^-11 scaling of
 * the accumulator, the main product);  exactly one of out_hi_lo ([n_rows][2 n_out] halves) and out_f32 ([n_rows][n_out])
 * is non-NULL.  k % 64 == 0;  tile: 0 = choose, 1 = 352 x 256 (n_out % 256 == 0), 3 = 352 x 128, 2 = 176 x 128 (n_out % 128 == 0).
 * Every tile walks K in the same order: a row's result does not depend on the tile or on the other rows of the launch.
 * Replaces two library GEMMs + rc_split_act_f16; the fp32 partial matrices never reach HBM. */
int rc_split_gemm_f16(const uint16_t *a_hi_lo, const uint16_t *w_lo_hi_hi, const float *bias, size_t n_rows, size_t n_out,
                      size_t k, int activation, float alpha, uint16_t *out_hi_lo, float *out_f32, int tile, rc_stream_t stream);

/* The layer kernels above with every option, as one request (zero-initialise, then fill what applies):
 *   y = post_scale * act(acc + bias + residual) + post_shift,   acc = the layer's products in fp32 on the matrix cores
 * residual: the skip connection of the reference's NonConvResBlock (librubiks/model.py:221-247), in the format of the
 * layer's own input ([n_rows][2 n_out] halves hi | lo; [n_rows][n_out] bf16 for rc_gemm_layer_bf16).  post_scale / post_shift
 * ([n_out] floats, both or neither): an eval-mode BatchNorm1d BEHIND the activation that cannot be folded into the next
 * Linear because a skip connection reads its output too (the last shared layer of the res_* architectures, model.py:249-264).
 * range_flag (device int, optional): OR-ed with 1 when a value written to out_hi_lo leaves IEEE half's range (|y| > 65504, or
 * not finite) -- the f16x3 split cannot carry it; librubiks.model.SplitF32Net then falls back to the fp32 GEMM chain.
 * Exactly one output.  rc_split_layer_f16: out_hi_lo | out_f32 | out_partials; with out_partials the K loop (3 k / 64 steps) is
 * cut into k_splits chunks (a divisor of 3 k / 64 leaving >= 2 steps per chunk, 2 .. 32; tile 0 / 1: 352 x 256 tiles, n_out % 256
 * == 0; tile 3: 352 x 128 tiles for small batches, n_out % 128 == 0; tile 7: 352 x 64), one workgroup per tile and chunk storing raw
 * accumulators out_partials[k_splits][n_rows][n_out] (no bias / residual / activation): the first rc_split_layer_corr_chunks(k,
 * k_splits) of them hold correction products only and still carry the factor 2^11, the others are in units of y
 * (rc_split_reduce_f16 finishes the layer).  rc_gemm_layer_bf16: a, w, residual, out_bf16 in bf16, one product. */
typedef struct rc_split_layer {
    const uint16_t *a;          /* [n_rows][2 k] halves hi | lo        (bf16 layer: [n_rows][k]) */
    const uint16_t *w;          /* [n_out][3 k] halves lo | hi | hi    (bf16 layer: [n_out][k]) */
    const float *bias;          /* [n_out] */
    const uint16_t *residual;   /* optional */
    const float *post_scale, *post_shift;   /* optional */
    uint16_t *out_hi_lo;        /* [n_rows][2 n_out] halves */
    float *out_f32;             /* [n_rows][n_out] */
    float *out_partials;        /* [k_splits][n_rows][n_out] */
    uint16_t *out_bf16;         /* [n_rows][n_out] (rc_gemm_layer_bf16 only) */
    size_t n_rows, n_out, k;
    int activation;             /* RC_ACT_* */
    float alpha;
    int tile;                   /* 0 = choose (see rc_split_gemm_f16) */
    int k_splits;               /* 0 / 1 = whole K per workgroup */
    int32_t *range_flag;        /* optional */
    int products;               /* rc_split_layer_f16: 0 / 3 = the three products of the split layer; 1 = ONE f16 product of a [n_rows][k]
                                 * and w [n_out][k] as they are (the input layer: a = [onehot | 2^-11 onehot], w = [W_hi | W_lo], k = 960) */
} rc_split_layer_t;
size_t rc_split_layer_struct_bytes(void);
int rc_split_layer_f16(const rc_split_layer_t *layer, rc_stream_t stream);
int rc_gemm_layer_bf16(const rc_split_layer_t *layer, rc_stream_t stream);
int rc_split_layer_corr_chunks(size_t k, int k_splits);   /* host only; -1 on a malformed request */
/* Finishes a layer whose products came as partial sums (rc_split_layer_f16 with out_partials, or the two fp32 library GEMMs
 * c_corr, c as partials[0], partials[1] with n_corr = 1):
 *   y = post_scale * act(sum_{p >= n_corr} partials[p] + 2^-11 sum_{p < n_corr} partials[p] + bias + residual) + post_shift
 * summed in the order p = 0, 1, ... (deterministic); partial p starts at partials + p * partial_stride floats.  Writes
 * out_hi_lo ([n_rows][2 n_cols] halves) and / or out_f32; n_cols % 8 == 0.  residual, post_*, range_flag as above. */
int rc_split_reduce_f16(const float *partials, size_t partial_stride, int n_partials, int n_corr, size_t n_rows, size_t n_cols,
                        const float *bias, const uint16_t *residual_hi_lo, int activation, float alpha, const float *post_scale,
                        const float *post_shift, uint16_t *out_hi_lo, float *out_f32, int32_t *range_flag, rc_stream_t stream);

/* ---- network head: last activation + skinny output layer in one pass -----------------------------------------
 * out[i][o] = bias[o] + sum_k w[o][k] * act(x[i][k])   for o < n_out <= 16   (float out, row pitch 16)
 * Replaces the final activation pass and the 1024 -> 13 GEMM of the merged policy/value heads
 * (librubiks/model.py:124-129,150-159): the 2 KB activation row is read once, the weights live in registers.
 *   x: bf16 [n][K] raw pre-activations (bias already added), K = 1024;  w: bf16 [n_out][K];  bias: float[n_out]. */
int rc_head_bf16(const uint16_t *x, size_t n, size_t K, const uint16_t *w, const float *bias, size_t n_out,
                 float *out, int activation, float alpha, rc_stream_t stream);

/* ---- Autodidactic-iteration targets (librubiks/train.py:292-325) ----------------------------------
 * For state i with children 12 i .. 12 i + 11 (rc_expand12 order):
 *   q[k]   = values[12 i + k] + (child_solved[12 i + k] ? win_reward : -1)
 *   policy_target[i] = first argmax_k q[k];   value_target[i] = q[policy_target[i]]
 *   fix_mode 1 ("lapanfix")  : value_target[i] = 0 where state_solved[i]
 *   fix_mode 2 ("schultzfix"): value_target[i] = 0 where i % depth == 0
 * values: float[12 n]; child_solved: uint8[12 n]; state_solved: uint8[n] (may be NULL unless fix_mode 1). */
int rc_adi_targets(const float *values, const uint8_t *child_solved, const uint8_t *state_solved, size_t n,
                   size_t depth, float win_reward, int fix_mode, int64_t *policy_target, float *value_target,
                   rc_stream_t stream);

/* ---- batched MCTS: one independent tree per scramble, lock-step iterations -------------------
 *
 * Replaces the per-tree Python loop of librubiks/solving/agents.py:415-645 (class MCTS) for B
 * trees at once.  One iteration of every running tree =
 *     rc_mcts_expand  ->  [network on the 11 B child rows]  ->  rc_mcts_backup  ->  rc_mcts_select
 * Node indices are 1-based per tree, 0 = "no neighbour" (agents.py:419-421); node arrays are
 * tree-major with capacity + 1 rows per tree.  All pointers are device pointers owned by the
 * caller (zero-initialised once, before the first rc_mcts_plant); the struct itself lives in host memory.
 */
#define RC_MCTS_NODE_WORDS 64   /* 32-bit words per node record (see rc_mcts_t) */
#define RC_MCTS_RUNNING 0
#define RC_MCTS_SOLVED 1        /* a child of the expanded leaf is the solved cube (agents.py:540-543) */
#define RC_MCTS_EXHAUSTED 2     /* len + 12 > max_states (agents.py:476) or node capacity reached */
#define RC_MCTS_PATH_OVERFLOW 3 /* a PUCT descent reached max_path levels: the path store is exhausted (the reference has no limit,
                                 * agents.py:575-595; max_path is a resource bound the caller chooses, not a search parameter) */
#define RC_MCTS_ROOT_SOLVED 4   /* the scramble itself is solved (agents.py:468) */
#define RC_MCTS_CORRUPT 5       /* rc_mcts_complete_graph / rc_mcts_shorten met an index that does not name a node of the tree (1 .. n_nodes)
                                 * in its hash table or neighbour rows: the rows are not this tree's data.  The tree is left as it is. */

typedef struct rc_mcts {
    uint32_t n_trees;    /* B */
    uint32_t capacity;   /* largest node index per tree; capacity + 1 < 2^24 (32-bit byte offsets into a tree's 256-byte records) */
    uint32_t hash_size;  /* slots per tree, power of two, >= 2 * (capacity + 1) */
    uint32_t max_path;   /* levels the path store can hold per tree (>= 2; see "descent path" below): a whole number of path blocks */
    uint32_t rows_per_tree; /* network rows reserved per tree and iteration: 11 (see child_soa) */
    uint32_t node_words;    /* 32-bit words between consecutive nodes in N / W / P / nbr / rec: RC_MCTS_NODE_WORDS.  A forest that
                             * only waits for rc_mcts_complete_graph / rc_mcts_shorten (finished trees) may instead pass 12 with
                             * nbr as a plain [B][capacity + 1][12] array; every other entry point rejects that */
    /* per node, [B][capacity + 1] */
    void *keys;          /* uint32[4]: the 20 codes packed 5 bits each (6 codes per dword) */
    /* The per-action arrays of a node (the reference's agents.py:421-426 attributes) are the fields of ONE record of
     * RC_MCTS_NODE_WORDS 32-bit words = 256 bytes = two cache lines, [B][capacity + 1][64], 256-byte aligned:
     *   line 0: words 0-11 N | 12-23 W | 24-27 walk record (rec, below) | 28-31 spare       (what backup / re-validation write)
     *   line 1: words 32-43 P | 44-55 nbr | 56-63 spare                                      (read-only once the children exist)
     * Each pointer below addresses its field of node 0 of tree 0: entry a of node n is X[n * RC_MCTS_NODE_WORDS + a].
     * Re-deciding a level of the previous descent path reads one record and dirties one line of it.
     * The reference's L (virtual losses, agents.py:427) is NOT stored: every backup clears exactly the entries the descent
     * before it raised (agents.py:569-570, 589-591), so between iterations L == nu x (how often the pending descent path
     * path_node / path_act leaves a node by an action or arrives by its reverse), and zero once a tree is solved.  The
     * kernels take their loss counts from the path; MCTSForest.tree_arrays() reports L from it. */
    int32_t *nbr;        /* x12  neighbors   (agents.py:421) */
    float *P;            /* x12  policy      (agents.py:423); values are float32-exact in the reference too */
    float *W;            /* x12  max value   (agents.py:426) */
    int32_t *N;          /* x12  visit count (agents.py:425) */
    float *V;            /*      value       (agents.py:424) */
    uint8_t *leaf;       /*      is leaf     (agents.py:422) */
    int32_t *hash;       /* [B][hash_size] open addressing, slot = node index or 0; full-key compare via keys */
    /* per tree, [B] */
    int32_t *n_nodes;    /* len(agent) (agents.py:644-645) */
    int32_t *status;     /* RC_MCTS_* */
    int32_t *solved_idx; /* node index of the solved child, solve_action = its action */
    int32_t *solved_action;
    int32_t *iterations;
    int32_t *path_len;   /* number of nodes on the current descent path, root included */
    int32_t *pending;    /* 0, or (carried action + 2) of a descent that rc_mcts_select suspended at its level budget */
    /* The descent path (agents.py:581-582,592-593) has no length limit in the reference.  The path arrays path_node / path_act /
     * path_next / short_act are therefore BLOCKED: level k of tree t is element
     *     ((k >> path_block_log2) * B + t) << path_block_log2  |  (k & (2^path_block_log2 - 1)),
     * i.e. [max_path >> path_block_log2][B][2^path_block_log2] -- block 0 is the dense [B][block] array a shallow tree lives in,
     * and deeper blocks can be address space that gets memory only for the trees that go that deep (path_rows). */
    int32_t *path_node;  /* indices_visited (agents.py:581,592) */
    uint8_t *path_act;   /* actions_taken   (agents.py:582,593) */
    /* per iteration staging */
    int8_t *child_soa;   /* [20][child_stride]: network input: the NEW children of tree t's leaf, packed in child order at
                          * columns 11 t + rank.  A non-root leaf always has a known child (its parent), so 11 rows
                          * suffice; a root's iteration takes two steps (phase): [root, children 0..9], then [children 10, 11] */
    size_t child_stride;
    int32_t *child_idx;  /* [B][12] node index of every child of the expanded leaf */
    uint32_t *new_mask;  /* [B] bit k set iff child k was not in the tree before */
    uint8_t *expanded;   /* [B] 1 iff the tree expanded a leaf in the current iteration */
    int32_t *select_stats; /* optional (may be NULL): [B][8] = first sequentially walked level, new path length,
                              10-ns ticks spent re-validating the old path, ticks and shader cycles spent in the
                              sequential walk, revisited levels that fell back to float64, revisited levels,
                              line-following rounds << 16 | levels they appended */
    /* optional, only needed by rc_mcts_shorten (may be NULL otherwise) */
    int32_t *bfs;        /* [B][capacity + 1][2] scratch: {claim, parent << 4 | action} */
    uint8_t *short_act;  /* (blocked like path_act) shortened action queue of every solved tree; never longer than the tree's last path */
    int32_t *short_len;  /* [B] its length, -1 where no shortened queue was produced */
    /* per node: the 16-byte walk record, owned by the kernels (words 24-27 of the node record):
     *   uint32 {neighbour through best0, neighbour through best1, best0 | best1 << 8 | leaf << 16, 0}
     * best0 = the action a PUCT descent takes at the node while no virtual loss is pending there, best1 = the same
     * with one loss on best0 (the descent arrived through rev(best0)).  rc_mcts_init / rc_mcts_expand write leaf
     * records, rc_mcts_select refreshes the records of the path it re-validates and walks by them. */
    void *rec;
    /* per tree: the last ring_k descent paths ("lines"), used by rc_mcts_select to validate the likely continuation of
     * a descent many levels at a time.  Path number s (= the tree's iteration count when it was walked) lives in slot
     * s % ring_k; zero-initialised by the caller, owned by the kernels.  ring_k: a power of two, 1 .. 64. */
    uint32_t ring_k;
    int32_t *ring_node;  /* [B][ring_k][ring_levels]: the first ring_levels levels of a path (deeper levels are walked one by one) */
    uint8_t *ring_act;   /* [B][ring_k][ring_levels] action taken at each level, 15 at the path's leaf */
    int32_t *ring_len;   /* [B][ring_k] */
    /* per tree, [B]: 0 = ordinary iterations; 1 / 2 = first / second step of a planted root's own iteration (| 16: that
     * expansion found a solved child, reported as RC_MCTS_SOLVED when the second step has completed the tree) */
    int32_t *phase;
    /* Optional (NULL = every tree, in order): the trees the per-iteration entry points work on.  Workgroup i serves tree
     * active[i] (a negative entry = nobody) and that tree's network rows are 11 i .. 11 i + 10: rows are packed by POSITION in
     * the list, not by tree index.  Dropping finished trees from a running batch is therefore a new list -- no tree moves in
     * memory, whatever the capacity -- and the network runs on 11 n_active rows.  rc_mcts_complete_graph / rc_mcts_shorten
     * read it the same way (the finished trees to post-process).  Device pointer, n_active <= n_trees entries. */
    const int32_t *active;
    uint32_t n_active;
    /* rc_mcts_select's re-validation re-decides in float64 the levels float32 could not settle ("pass B"): from a list while
     * there are at most unc_list_cap of them (the kernel clamps it to 128, the production value), from a bitmap beyond.  Tests
     * lower it to drive the bitmap form with the few unsettled levels real networks produce; results do not depend on it. */
    uint32_t unc_list_cap;
    /* Optional (NULL = every row of the node arrays is backed by memory): [B] number of node rows (indices 0 .. mapped_rows[t] - 1)
     * of tree t that the per-node arrays keys / node records / V / leaf currently have memory behind -- for forests whose node
     * arrays are reserved address ranges mapped on demand (rc_vmm_*: the reference grows its arrays as a tree grows,
     * agents.py:450-459).  A tree whose next expansion would reach beyond it simply does not expand in that iteration (it stays
     * RUNNING, `expanded` stays 0, its iteration count does not advance) and tries again in the next one: the host maps ahead
     * of the trees' growth, the kernels never touch a row that is not there. */
    const int32_t *mapped_rows;
    /* ---- descent paths of any length (the reference walks until it meets a leaf, agents.py:575-595) ----
     * rc_mcts_select keeps the first lds_levels levels of the path it works on in LDS; deeper levels are read and written in
     * path_node / path_act in place, and chained by node through path_next (same blocked layout; only entries of levels >=
     * lds_levels are used).  Results do not depend on lds_levels (tests run 8 and 64 against the reference's traces). */
    uint32_t path_block_log2; /* log2 of the levels per path block (12 in production) */
    uint32_t lds_levels;      /* 1 .. 4096 */
    uint32_t ring_levels;     /* levels per ring line, 1 .. 4096 (the line tag holds 12 bits of level) */
    uint32_t *path_next;      /* scratch owned by the kernels */
    /* Optional (NULL = max_path for every tree): [B] levels of tree t's path blocks that have memory behind them (a multiple of
     * the block size).  A descent that reaches it is SUSPENDED (pending, as at a level budget) and resumes in the next call;
     * the host maps the next block in between.  Only a descent that reaches max_path itself ends the tree (PATH_OVERFLOW). */
    const int32_t *path_rows;
} rc_mcts_t;

/* sizeof(rc_mcts_t) as the library was compiled: a binding that mirrors the struct (ctypes, cgo, JNI) checks its own
 * layout against it before the first call.  Host-only, needs no device. */
size_t rc_mcts_struct_bytes(void);

/* (Re)starts trees: for i < n_slots, tree slots[i] (or tree i if slots is NULL) is emptied -- its hash table is cleared,
 * nothing else needs to be -- and gets roots_soa column first_col + i as node 1; a solved root gets RC_MCTS_ROOT_SOLVED.
 * The root is evaluated and expanded (agents.py:466-473 and the first expand_leaf) inside the next two ordinary
 * iterations, on the tree's own 11 network rows: see `phase`.  Other trees of the forest are not touched, so finished
 * trees of a running batch can hand their slots to waiting scrambles between two iterations. */
int rc_mcts_plant(const rc_mcts_t *m, const int32_t *slots, uint32_t n_slots, const int8_t *roots_soa, size_t stride,
                  size_t first_col, rc_stream_t stream);
/* expand_leaf part 1 (agents.py:505-544): 12 children of the path's leaf, dedup against the tree,
 * new indices in child order, links both ways, first solved child. */
int rc_mcts_expand(const rc_mcts_t *m, uint32_t max_states, rc_stream_t stream);
/* expand_leaf part 2 (agents.py:555-571): P, V of new children, W/N/L updates along the path.
 * probs = softmax(policy logits) rows, values = value head, both for the 11 B child rows. */
int rc_mcts_backup(const rc_mcts_t *m, const float *probs, const float *values, rc_stream_t stream);
/* Same as rc_mcts_backup, but straight from the network's head output: row r of `head` (bf16 when
 * head_is_bf16 != 0, else float; `ld` elements per row) holds the 12 policy logits followed by the value.
 * P = softmax(logits) is evaluated in float32 inside the kernel (max-subtracted, expf), which removes the
 * separate cast / slice / softmax / copy launches of the generic path. */
int rc_mcts_backup_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, rc_stream_t stream);
/* find_leaf (agents.py:575-595): PUCT descent with virtual loss, float64 arithmetic as NumPy's.
 * level_budget = 0: every running tree descends to its leaf (strict lock step).
 * level_budget > 0: a tree walks at most that many NEW levels per call; if it has not reached a leaf it is
 * suspended (pending) and resumes in the next call, and rc_mcts_expand / rc_mcts_backup skip it meanwhile.
 * Each tree still performs exactly the reference's sequence of iterations; only their timing changes, so the
 * slowest descent of the batch no longer sets the pace of every iteration. */
int rc_mcts_select(const rc_mcts_t *m, double c, uint32_t level_budget, rc_stream_t stream);
/* rc_mcts_backup (or rc_mcts_backup_head) followed by rc_mcts_select as ONE kernel with the same results: the path part of
 * the backup is applied by the lanes that re-decide the path's levels, to the rows they hold in registers anyway. */
int rc_mcts_backup_select(const rc_mcts_t *m, const float *probs, const float *values, double c, uint32_t level_budget,
                          rc_stream_t stream);
int rc_mcts_backup_select_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, double c, uint32_t level_budget,
                               rc_stream_t stream);
/* The tree side of an iteration as ONE launch: rc_mcts_backup_select (rc_mcts_step) / rc_mcts_backup_select_head
 * (rc_mcts_step_head) followed by the rc_mcts_expand of the NEXT iteration, done by the wave that walked to the leaf.  An
 * iteration is then  [network on the rows the previous step left] -> rc_mcts_step*;  trees enter through
 * rc_mcts_plant_expanded, which is rc_mcts_plant + the root's expansion (its rows use list position == tree index: every tree
 * listed in order, as while scrambles are still waiting for slots).  Results are those of the three-kernel form.
 * Launch shape: 256 threads per tree in a forest of more than 512 listed trees, 512 / 1 024 threads and four waves checking a
 * descent line together below that (idle CUs; same results).  RUBIKS_STEP_THREADS=256|512|1024 and RUBIKS_LINE_WAVES=1|4 in the
 * environment pin the two choices, for A/B measurements only (read once per process). */
int rc_mcts_plant_expanded(const rc_mcts_t *m, const int32_t *slots, uint32_t n_slots, const int8_t *roots_soa, size_t stride,
                           size_t first_col, uint32_t max_states, rc_stream_t stream);
int rc_mcts_step(const rc_mcts_t *m, const float *probs, const float *values, double c, uint32_t level_budget, uint32_t max_states,
                 rc_stream_t stream);
int rc_mcts_step_head(const rc_mcts_t *m, const void *head, size_t ld, int head_is_bf16, double c, uint32_t level_budget,
                      uint32_t max_states, rc_stream_t stream);
/* _complete_graph (agents.py:597-611) for every tree with status RC_MCTS_SOLVED: each leaf is linked, both
 * ways, to those of its 12 children that already exist in the tree (looked up in the tree's hash table). */
int rc_mcts_complete_graph(const rc_mcts_t *m, rc_stream_t stream);
/* _shorten_action_queue (agents.py:613-633) for every solved tree, after rc_mcts_complete_graph: breadth-first
 * search over the tree's neighbour table from the root to the solved node, discovering nodes in exactly the
 * order of the reference's FIFO queue (frontier order, then action order), so the returned queue is the
 * reference's.  Overwrites the tree's hash table (used as the two frontier arrays). */
int rc_mcts_shorten(const rc_mcts_t *m, rc_stream_t stream);

/* Copies the finished (or suspended) trees src_trees[0 .. n) of `src` into slots dst_first .. dst_first + n - 1 of `dst`: rows
 * 0 .. n_nodes of keys, leaf and -- into a search forest (dst->node_words == RC_MCTS_NODE_WORDS) -- the node records and V, or --
 * into a results-only forest (dst->node_words == 12) -- the neighbour rows alone; plus the tree's whole hash table.  Only rows
 * that exist are touched (both forests may be mapped on demand); the per-tree words are the caller's to copy.  Same capacity
 * and hash_size on both sides; src must be a search forest.  Replaces whole-capacity tensor copies (capacity x 285 B per
 * tree) where finished trees leave a running forest (MCTSForest.bury / subset). */
int rc_mcts_copy_trees(const rc_mcts_t *src, const rc_mcts_t *dst, const int32_t *src_trees, uint32_t n, uint32_t dst_first,
                       rc_stream_t stream);

/* ---- node storage on demand (HIP virtual memory management) ---------------------------------------------------------------
 * The reference's node arrays grow by doubling as a tree grows (librubiks/solving/agents.py:450-459): max_states = 175 000 costs
 * only the nodes a search creates.  rc_vmm_reserve reserves an address range of the array's full size, rc_vmm_map puts memory
 * behind [offset, offset + bytes) of it, chunk by chunk (chunk_bytes: a multiple of 2 MiB; a chunk already mapped is left
 * alone); addresses never change, so structs, kernels and captured graphs are unaffected.  rc_vmm_map is host-synchronous and
 * may be called while kernels work on other parts of the range (*out_new_bytes = bytes it added, also when it fails part of
 * the way).  rc_vmm_release: after the caller has synchronised; unmaps, flushes the GPU's translations and puts the address
 * range on the idle list of its size class (reservations are whole powers of two), where the next rc_vmm_reserve of that class
 * finds it: rc_vmm_retired_bytes = address space idle on those lists, bounded by one range per class and live overlap.
 * Diagnosis: every call is recorded (the last 512 events in memory; all of them, flushed line by line, in the file the
 * environment variable RUBIKS_VMM_LOG names, "stderr" = stderr); rc_vmm_dump writes the live ranges with their mapped chunk runs,
 * the idle ranges and the recent events as text (at most cap - 1 characters; *out_needed = full length incl. NUL);
 * rc_vmm_classify says what an address -- that of a GPU memory access fault, say -- is to the library at this moment;
 * rc_vmm_chunk_map copies the per-chunk "has memory" flags of a range. */
#define RC_VMM_ADDR_UNKNOWN 0    /* not inside any reservation of this library */
#define RC_VMM_ADDR_MAPPED 1     /* live range, memory behind the chunk */
#define RC_VMM_ADDR_UNMAPPED 2   /* live range, NO memory behind the chunk: a row touched before it was mapped */
#define RC_VMM_ADDR_SLACK 3      /* live reservation, outside the range handed out (alignment slack, rest of the size class) */
#define RC_VMM_ADDR_IDLE 4       /* a released range waiting for reuse: nothing mapped */
int rc_vmm_granularity(size_t *out_bytes);
int rc_vmm_reserve(size_t bytes, size_t chunk_bytes, void **out_base);
int rc_vmm_map(void *base, size_t offset, size_t bytes, size_t *out_new_bytes);
int rc_vmm_mapped_bytes(void *base, size_t *out_bytes);
int rc_vmm_chunk_map(void *base, uint8_t *out_flags, size_t cap, size_t *out_chunks);
int rc_vmm_release(void *base);
int rc_vmm_retired_bytes(size_t *out_bytes);
int rc_vmm_classify(const void *addr, int *out_kind, void **out_base, size_t *out_offset);
int rc_vmm_dump(char *out, size_t cap, size_t *out_needed);

/* ---- batched weighted A*: B independent problems, N expansions each per iteration --------------
 *
 * Replaces the per-problem Python loop of librubiks/solving/agents.py:171-413 (class AStar).
 * One iteration of every running problem =
 *     rc_astar_pop_expand -> [prefix sum of new_count on the caller's side] -> rc_astar_gather_new
 *     -> [value network on the compacted new states] -> rc_astar_push_relax
 * Node indices are 1-based per problem (index 0 unused, agents.py:189); arrays are problem-major
 * with capacity + 1 rows.  Device pointers, zero-initialised by the caller except `claim`, which
 * must be filled with INT32_MAX.
 */
#define RC_ASTAR_RUNNING 0
#define RC_ASTAR_SOLVED 1        /* a NEW state of the batch is the solved cube (agents.py:321-323) */
#define RC_ASTAR_EXHAUSTED 2     /* len + 12 N > max_states (agents.py:236) or capacity reached */
#define RC_ASTAR_OPEN_EMPTY 3    /* open list ran dry (the reference would spin; defined as unsolved) */
#define RC_ASTAR_ROOT_SOLVED 4   /* agents.py:230 */

typedef struct rc_astar {
    uint32_t n_problems; /* B */
    uint32_t capacity;   /* largest node index per problem */
    uint32_t hash_size;  /* slots per problem, power of two, >= 2 * (capacity + 1) */
    uint32_t expansions; /* N (agents.py:218) */
    /* per node, [B][capacity + 1] */
    void *keys;              /* uint32[4] packed state (as rc_mcts) */
    int32_t *G;              /* path cost; integer valued in the reference's float array (agents.py:203,311) */
    int32_t *parents;        /* agents.py:204 */
    uint8_t *parent_actions; /* agents.py:205 */
    int32_t *claim;          /* scratch for first-/last-occurrence election, INT32_MAX when idle */
    int32_t *hash;           /* [B][hash_size]: > 0 node index, < 0 pending child row -(row+1), 0 empty */
    /* open list: binary min-heap on (cost, index), [B][capacity + 1] */
    double *heap_cost;
    int32_t *heap_idx;
    /* per problem, [B] */
    int32_t *heap_size;
    int32_t *n_nodes;        /* len(agent) (agents.py:409-410) */
    int32_t *status;         /* RC_ASTAR_* */
    int32_t *solved_idx;
    int32_t *iterations;
    int32_t *n_popped;       /* parents expanded in the current iteration */
    int32_t *new_count;      /* states added in the current iteration */
    /* per iteration staging, [B][N] and [B][12 N] */
    int32_t *popped;         /* node indices in pop order (agents.py:238-239) */
    void *child_keys;        /* uint32[4] per child row 12 p + k */
    int32_t *child_node;     /* node index of a seen child / hash slot of an unseen one */
    int32_t *row_tmp;        /* scratch per row (relaxation values) */
    uint8_t *row_flags;      /* bit0 first occurrence & unseen (new state), bit1 first occurrence & seen */
} rc_astar_t;

/* Root = node 1 with G = 0 and cost 0 on the open list (agents.py:233-234). */
int rc_astar_init(const rc_astar_t *a, const int8_t *roots_soa, size_t stride, rc_stream_t stream);
/* Pops min(len(open), N) lowest (cost, index) nodes, expands their 12 N children, dedups against
 * the problem's states and inside the batch (first occurrence in row order, agents.py:286-295),
 * appends the new states with G, parent, parent_action (agents.py:299-313).  Sets new_count. */
int rc_astar_pop_expand(const rc_astar_t *a, uint32_t max_states, rc_stream_t stream);
/* Writes the new states of every problem into out_soa at columns new_offset[b] .. (exclusive prefix
 * sum of new_count, B + 1 entries) -- the compacted network input of this iteration. */
int rc_astar_gather_new(const rc_astar_t *a, const int32_t *new_offset, int8_t *out_soa, size_t stride,
                        rc_stream_t stream);
/* cost = lambda * G - value (float64, agents.py:383), push on the open list (agents.py:315-317),
 * win check on the new states (agents.py:321-323), else relaxation of the first-seen children
 * (agents.py:326-328,333-367).  values[new_offset[b] + i] belongs to new state i of problem b. */
int rc_astar_push_relax(const rc_astar_t *a, const int32_t *new_offset, const float *values, double lambda,
                        rc_stream_t stream);

/* ---- breadth-first search (one problem per call sequence, GPU-wide levels) -----------------------
 *
 * Replaces the FIFO loop of librubiks/solving/agents.py:92-131 (class BFS) with level-synchronous
 * chunks that keep its bookkeeping exactly: nodes are numbered in discovery order (node 0 = the
 * start state), child row 12 p + k = action k on the p-th parent of the chunk, a row is new iff its
 * state is neither stored nor produced by a lower row, the first solved row ends the search, and
 * the `len(self) < max_states` test before each pop (agents.py:105) becomes the "cutoff parent".
 * Per chunk:  rc_bfs_expand -> [host reads result[0..4]] -> rc_bfs_commit (if neither solved nor cut).
 * Device pointers; nothing needs initialising except through rc_bfs_init.
 */
typedef struct rc_bfs {
    uint32_t capacity;   /* node slots; rc_bfs_expand needs n_nodes + 12 * n_parents <= capacity */
    uint32_t hash_size;  /* power of two, >= 2 * capacity */
    uint32_t chunk;      /* most parents per rc_bfs_expand */
    uint32_t reserved;
    void *keys;          /* uint32[4] packed state per node (as rc_mcts) */
    uint32_t *parent;    /* agents.py:102,119: the state a node was reached from ... */
    uint8_t *action;     /* ... and the action taken */
    int32_t *hash;       /* [hash_size]: > 0 node index + 1, < 0 pending row -(row+1), 0 empty */
    void *child_keys;    /* uint32[4] per row, [12 * chunk] */
    int32_t *child_slot; /* [12 * chunk] */
    uint32_t *flags;     /* [12 * chunk] 1 = new state */
    uint32_t *prefix;    /* [12 * chunk] exclusive prefix sum of flags */
    uint64_t *result;    /* [5]: first solved row (or ~0), new rows, cutoff parent (n_parents = none),
                            new rows before the solved row, new rows before the cutoff parent */
    void *scan_tmp;      /* rc_bfs_scan_bytes(chunk) bytes */
    size_t scan_tmp_bytes;
} rc_bfs_t;

size_t rc_bfs_scan_bytes(uint32_t chunk);
/* Clears the table and stores root_state (int8[20], device) as node 0.  result[0] = 0 if it is solved
 * (agents.py:100), ~0 otherwise. */
int rc_bfs_init(const rc_bfs_t *b, const int8_t *root_state, rc_stream_t stream);
/* Expands parents lo .. lo + n_parents - 1 with n_nodes states stored so far; fills result. */
int rc_bfs_expand(const rc_bfs_t *b, uint32_t lo, uint32_t n_parents, uint32_t n_nodes, uint32_t max_states,
                  rc_stream_t stream);
/* Appends the new states of the last rc_bfs_expand (same lo, n_parents, n_nodes) in row order. */
int rc_bfs_commit(const rc_bfs_t *b, uint32_t lo, uint32_t n_parents, uint32_t n_nodes, rc_stream_t stream);
/* Actions from the start state to `node`, last action first (agents.py:112-115); *out_len = ~0 if
 * the chain is longer than max_len. */
int rc_bfs_path(const rc_bfs_t *b, uint32_t node, uint8_t *out_actions, uint32_t *out_len, uint32_t max_len,
                rc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RUBIKS_HIP_H */
