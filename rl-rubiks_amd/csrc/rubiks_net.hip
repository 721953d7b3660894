// Network-side kernels that touch cube states: the input layer of the policy/value MLP fused with
// the one-hot encoding (the one-hot A fragments of the MFMAs are generated from the cube codes, so the (n x 480) one-hot
// matrix, the K = 480 GEMM and the separate bias + activation pass never exist), the fused head kernels, the
// activation / reduce passes of the split engine and the ADI target kernel.
#include <atomic>

#include <type_traits>

#include "rubiks_common.h"
#include "rubiks_netmath.h"

namespace rubiks {

constexpr int kOH = 480;

__device__ __forceinline__ float act_apply(float x, int act, float alpha) {
    if (act == RC_ACT_RELU) return fmaxf(x, 0.f);
    if (act == RC_ACT_ELU) return x > 0.f ? x : alpha * (__expf(x) - 1.f);
    return x;
}


// =================================================================================================
// Input layer on the matrix cores.
//   out[32 states x 128 cols] per wave = OneHot[32 x 480] * W1slice^T[480 x 128], 30 k-steps of
//   v_mfma_f32_32x32x16_bf16 per 32-column tile.
//   A (32 x 16 per k-step): lane l (r = l & 31, h = l >> 5) holds A[state r][k = 16 ks + 8 h + j], j = 0..7.
//     8 | 24, so those eight one-hot positions lie inside ONE cubie's 24-wide block: cubie (2 ks + h) / 3,
//     offset 8 * ((2 ks + h) % 3).  The fragment is "1.0 at position code - offset if that is in 0..7".
//   B (16 x 32): lane holds B[k = 16 ks + 8 h + j][col r] = W1[col][k..k+7]: one 16-byte LDS read from the
//     slice stored column-major-in-k with a 976-byte pitch (61 x 16 B, odd -> the 16 lanes of a ds_read_b128
//     group hit 16 different bank quads).
//   D: lane holds column l & 31 of states (reg & 3) + 8 (reg >> 2) + 4 h.
// Epilogue: activation, bf16; a lane's four accumulator tiles are four ADJACENT columns, so a state's row segment
// leaves as one 8-byte store per lane, 256 contiguous bytes per 32 lanes -- no LDS transposition, which keeps the
// workgroup's LDS at the W1 slice alone and lets TWO waves share a SIMD: one wave's epilogue (VALU: ELU on 64
// values per lane) runs under the other's MFMAs.
// =================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kMfCols = 128;                        // output columns per workgroup
constexpr int kMfPitch = 976;                       // bytes per column of the LDS slice: 480 bf16 + 16 B pad
constexpr int kMfWaves = 8;                         // two per SIMD
constexpr int kMfTile = 32;                         // states per MFMA tile
constexpr int kMfSub = 1;                           // tiles a wave holds at once (they share the B fragments)

template <int ACT, bool F16>   // F16: W1 and the one-hot fragments in IEEE half (v_mfma_f32_32x32x16_f16), else bf16
__global__ __launch_bounds__(kMfWaves * kWave) void k_first_layer_mfma(const u8 *__restrict__ soa, size_t n, size_t stride,
                                                                     const uint4 *__restrict__ w1, const float *__restrict__ bias,
                                                                     uint4 *__restrict__ out, u32 H, u32 rows_per_block,
                                                                     float alpha) {
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    unsigned char *wslice = lds;                                                   // [128 slots][976 B]
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint4 *onehot = reinterpret_cast<uint4 *>(lds + kMfCols * kMfPitch);           // [9] A fragments
    const u32 col_tiles = H / kMfCols;
    const u32 ct = blockIdx.x % col_tiles, rg = blockIdx.x / col_tiles;
    const size_t row_lo = (size_t)rg * rows_per_block;
    if (row_lo >= n) return;
    const size_t row_hi = (row_lo + rows_per_block < n) ? row_lo + rows_per_block : n;

    // W1[128 ct .. +127][0..479] -> LDS, 60 chunks of 16 B per column.  Column g of the slice goes to slot
    // (g % 4) * 32 + g / 4: MFMA column tile c, lane r then owns column 4 r + c, i.e. a lane's four accumulators
    // are four ADJACENT output columns (one 8-byte store per state in the epilogue), while the B reads of a
    // tile still walk 32 consecutive slots (conflict-free with the odd 16-byte pitch).
    {   // 7 680 chunks / 512 threads = 15 per thread, requested five at a time before the first LDS write
        constexpr int kPer = kMfCols * 60 / (kMfWaves * kWave), kBatch = 5;
#pragma unroll
        for (int b0 = 0; b0 < kPer; b0 += kBatch) {
            uint4 tmp[kBatch];
#pragma unroll
            for (int t = 0; t < kBatch; ++t) {
                const u32 i = tid + (b0 + t) * (kMfWaves * kWave);
                tmp[t] = w1[(size_t)(ct * kMfCols + i / 60) * 60 + i % 60];
            }
#pragma unroll
            for (int t = 0; t < kBatch; ++t) {
                const u32 i = tid + (b0 + t) * (kMfWaves * kWave);
                const u32 g = i / 60, slot = (g & 3) * 32 + (g >> 2);
                *reinterpret_cast<uint4 *>(wslice + slot * kMfPitch + (i % 60) * 16) = tmp[t];
            }
        }
    }
    if (tid < 9) {   // A fragment "bf16 1.0 at position p" for p = 0..7, all zero for p = 8
        u32 w[4] = {0, 0, 0, 0};
        const u32 one = F16 ? 0x3c00u : 0x3f80u;   // 1.0 in half / bf16
        if (tid < 8) w[tid >> 1] = (tid & 1) ? one << 16 : one;
        onehot[tid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const u32 r = lane & 31, h = lane >> 5;
    float b[4];   // bias of this lane's four columns 4 r + c
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = bias[ct * kMfCols + 4 * r + c];
    __syncthreads();

    // A wave works on kMfSub sub-tiles of 32 states at once: every B fragment read from LDS feeds kMfSub MFMAs, which
    // halves the LDS traffic per state (with one sub-tile the k-loop is bound by the CU's 128 B/clk of LDS, not by
    // the matrix cores).  The 20 codes of a state stay packed in five dwords; rows past the end are clamped to the
    // last state (their results are never stored), which keeps the loads free of branches.
    const u32 off_a = h ? 8u : 0u, off_c = h ? 16u : 8u;   // this lane half's offsets in k-steps 3m and 3m + 2
    constexpr size_t kStep = (size_t)kMfWaves * kMfSub * kMfTile;
    for (size_t t0 = row_lo + (size_t)wave * kMfSub * kMfTile; t0 < row_hi; t0 += kStep) {
        u32 pk[kMfSub][5];
#pragma unroll
        for (int u = 0; u < kMfSub; ++u) {
            const size_t row = t0 + (size_t)u * kMfTile + r;
            const u8 *p = soa + (row < n ? row : n - 1);
            u32 raw[kPlanes];
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) raw[j] = p[(size_t)j * stride];
#pragma unroll
            for (int q = 0; q < 5; ++q)
                pk[u][q] = (raw[4 * q] & 31u) | (raw[4 * q + 1] & 31u) << 8 | (raw[4 * q + 2] & 31u) << 16 | (raw[4 * q + 3] & 31u) << 24;
        }
        auto code = [&](int u, int j) -> u32 { return (pk[u][j >> 2] >> (8 * (j & 3))) & 0xffu; };
        f32x16 acc[kMfSub][4];   // start from the bias: the epilogue needs no add
#pragma unroll
        for (int u = 0; u < kMfSub; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[u][c][i] = b[c];
        // Three k-steps (48 one-hot positions) cover two cubies.  Lane half h = 0 sees (cubie 2m, offset 0),
        // (2m, 16), (2m+1, 8); half h = 1 sees (2m, 8), (2m+1, 0), (2m+1, 16).
        // Software pipeline over the 30 k-steps: while the MFMAs of step ks run, the B fragments of step ks + 1 are
        // already on their way from LDS and so are its one-hot A fragments.
        auto a_frag = [&](int u, int ks) -> uint4 {
            const int m2 = ks / 3, ph = ks % 3;
            const u32 pos = ph == 0 ? code(u, 2 * m2) - off_a : ph == 1 ? (h ? code(u, 2 * m2 + 1) : code(u, 2 * m2) - 16u)
                                                                         : code(u, 2 * m2 + 1) - off_c;
            return onehot[min(pos, 8u)];                    // pos in 0..7 iff the 1 falls into this fragment
        };
        const unsigned char *bbase = wslice + r * kMfPitch + 16 * h;   // + c * 32 * pitch + 32 * ks
        uint4 a_cur[kMfSub], b_cur[4];
#pragma unroll
        for (int u = 0; u < kMfSub; ++u) a_cur[u] = a_frag(u, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) b_cur[c] = *reinterpret_cast<const uint4 *>(bbase + c * 32 * kMfPitch);
#pragma unroll
        for (int ks = 0; ks < 30; ++ks) {
            uint4 a_nxt[kMfSub], b_nxt[4] = {b_cur[0], b_cur[1], b_cur[2], b_cur[3]};
#pragma unroll
            for (int u = 0; u < kMfSub; ++u) a_nxt[u] = a_cur[u];
            if (ks + 1 < 30) {
#pragma unroll
                for (int c = 0; c < 4; ++c) b_nxt[c] = *reinterpret_cast<const uint4 *>(bbase + c * 32 * kMfPitch + 32 * (ks + 1));
#pragma unroll
                for (int u = 0; u < kMfSub; ++u) a_nxt[u] = a_frag(u, ks + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int u = 0; u < kMfSub; ++u) {
                    if (F16)
                        acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[u]),
                                                                           __builtin_bit_cast(f16x8, b_cur[c]), acc[u][c], 0, 0, 0);
                    else
                        acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur[u]),
                                                                            __builtin_bit_cast(bf16x8, b_cur[c]), acc[u][c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < kMfSub; ++u) a_cur[u] = a_nxt[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) b_cur[c] = b_nxt[c];
        }
        // epilogue: activation -> bf16; the lane's columns 4 r .. 4 r + 3 of one state are one 8-byte global store
        uint2 *out2 = reinterpret_cast<uint2 *>(out);
#pragma unroll
        for (int u = 0; u < kMfSub; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const size_t row = t0 + (size_t)u * kMfTile + (i & 3) + 8 * (i >> 2) + 4 * h;
                const uint2 v = make_uint2(pack_bf16(act_apply(acc[u][0][i], ACT, alpha), act_apply(acc[u][1][i], ACT, alpha)),
                                           pack_bf16(act_apply(acc[u][2][i], ACT, alpha), act_apply(acc[u][3][i], ACT, alpha)));
                if (row < n) out2[row * (H / 4) + ct * (kMfCols / 4) + r] = v;
            }
    }
}

// =================================================================================================
// Network head: out = W_head * act(x) + b for a skinny output layer (13 outputs of 1024 inputs).
// One workgroup per 16 rows on the matrix cores: 32 k-steps of v_mfma_f32_16x16x32_bf16, eight per wave.
//   A (16 rows x 32 k): lane l (r = l & 15, g = l >> 4) loads x[row r][32 ks + 8 g .. +7] (16 bytes), applies the
//     activation in fp32 and repacks to bf16 -- the last hidden layer's ELU costs no pass over memory.
//   B (32 k x 16 outputs): W_head[o = l & 15][32 ks + 8 g .. +7]; a wave's eight fragments (outputs >= 13 are zero)
//     are loaded once and stay in registers for every tile the workgroup processes.
//   D: lane holds output l & 15 of rows 4 g .. 4 g + 3.
// =================================================================================================
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kHeadK = 1024, kHeadSteps = kHeadK / 32, kHeadMaxOut = 16;

template <int ACT>
__global__ __launch_bounds__(kBlock) void k_head(const uint4 *__restrict__ x, size_t n, const uint4 *__restrict__ w,
                                                 const float *__restrict__ bias, float *__restrict__ out, u32 n_out, float alpha) {
    // One 16-row tile per workgroup pass; the tile's K = 1024 is split over the four waves (8 k-steps each), so the
    // ELU of the 16 384 inputs of a tile is spread over four SIMDs and a wave keeps only its quarter of the weights
    // in registers; the four partial 16 x 16 products meet in LDS.
    __shared__ float s_part[kBlock / kWave][16][17];
    constexpr int kSteps = kHeadSteps / (kBlock / kWave);   // 8
    const u32 tid = threadIdx.x, wv = tid / kWave, lane = tid & (kWave - 1), r = lane & 15, g = lane >> 4;
    uint4 bfrag[kSteps];
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks)
        bfrag[ks] = (r < n_out) ? w[(size_t)r * (kHeadK / 8) + (wv * kSteps + ks) * 4 + g] : make_uint4(0, 0, 0, 0);
    const size_t n_tiles = ceil_div(n, (size_t)16);
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t row = tile * 16 + r;
        const uint4 *xr = x + (row < n ? row : n - 1) * (kHeadK / 8) + wv * kSteps * 4 + g;   // rows past the end: clamped, never stored
        uint4 a[kSteps];
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) a[ks] = xr[ks * 4];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) {
            const u32 aw[4] = {a[ks].x, a[ks].y, a[ks].z, a[ks].w};
            u32 pk[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                pk[d] = pack_bf16(act_apply(__uint_as_float(aw[d] << 16), ACT, alpha),
                                  act_apply(__uint_as_float(aw[d] & 0xffff0000u), ACT, alpha));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3])),
                                                          __builtin_bit_cast(bf16x8, bfrag[ks]), acc, 0, 0, 0);
        }
        __syncthreads();   // the previous tile's partial sums have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wv][g * 4 + i][r] = acc[i];   // D: lane holds output r of rows 4 g .. 4 g + 3
        __syncthreads();
        {   // 256 threads = 16 rows x 16 outputs
            const u32 orow = tid >> 4, o = tid & 15;
            const float v = s_part[0][orow][o] + s_part[1][orow][o] + s_part[2][orow][o] + s_part[3][orow][o];
            const size_t grow = tile * 16 + orow;
            if (grow < n) out[grow * kHeadMaxOut + o] = (o < n_out) ? v + bias[o] : 0.f;
        }
    }
}

// ADI targets: 12-way segmented argmax of value + reward, with the goal-state fixes (train.py:292-325).
__global__ __launch_bounds__(kBlock) void k_adi_targets(const float *__restrict__ values, const u8 *__restrict__ child_solved,
                                                        const u8 *__restrict__ state_solved, size_t n, size_t depth,
                                                        float win_reward, int fix_mode, long long *__restrict__ policy,
                                                        float *__restrict__ value) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        float best = 0.f;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < kActions; ++k) {
            const float q = values[i * kActions + k] + (child_solved[i * kActions + k] ? win_reward : -1.f);
            if (k == 0 || q > best) { best = q; arg = k; }   // first maximum
        }
        if (fix_mode == 1 && state_solved[i]) best = 0.f;
        if (fix_mode == 2 && i % depth == 0) best = 0.f;
        policy[i] = arg;
        value[i] = best;
    }
}

}  // namespace rubiks

using namespace rubiks;

extern "C" int rc_first_layer_mfma_bf16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w1, const float *bias,
                                        uint16_t *out, size_t H, int activation, float alpha, int table_is_f16,
                                        rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(w1 && bias && out, RC_ERR_NULL);
    RC_REQUIRE(aligned16(w1) && aligned16(out), RC_ERR_ALIGN);
    RC_REQUIRE(H >= kMfCols && H % kMfCols == 0 && activation >= RC_ACT_NONE && activation <= RC_ACT_ELU, RC_ERR_RANGE);
    const u32 col_tiles = (u32)(H / kMfCols);
    u32 row_groups = (256 + col_tiles - 1) / col_tiles;   // one workgroup per CU (LDS), every W1 slice staged once per row group
    // whole passes of the eight waves per workgroup; a batch too small for that (a narrowed forest: 352 rows) is spread over more
    // workgroups in units of ONE wave's rows instead -- fewer busy waves per CU, the same work per wave, a shorter launch
    const size_t per_group = ceil_div(n, row_groups), pass_rows = kMfWaves * kMfSub * kMfTile;
    u32 rows_per_block = (u32)round_up(per_group, per_group < pass_rows ? (size_t)kMfSub * kMfTile : pass_rows);
    row_groups = (u32)ceil_div(n, rows_per_block);
    const size_t lds_bytes = (size_t)kMfCols * kMfPitch + 9 * 16;
    const dim3 grid(col_tiles * row_groups), block(kMfWaves * kWave);
    hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH_MF(ACT, F16)                                                                                     \
    do {                                                                                                           \
        /* per device and thread-safe: the attribute belongs to the function ON A DEVICE (one process may drive several) */ \
        static std::atomic<unsigned long long> attr_set{0};                                                        \
        int dev_ = 0;                                                                                              \
        (void)hipGetDevice(&dev_);                                                                                 \
        if (!((attr_set.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ull)) {                                  \
            hipError_t e = hipFuncSetAttribute((const void *)k_first_layer_mfma<ACT, F16>,                          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);        \
            if (e != hipSuccess) return hip_rc(e);                                                                 \
            attr_set.fetch_or(1ull << (dev_ & 63), std::memory_order_release);                                     \
        }                                                                                                          \
        hipLaunchKernelGGL((k_first_layer_mfma<ACT, F16>), grid, block, lds_bytes, s, (const u8 *)soa, n, stride,   \
                           (const uint4 *)w1, bias, (uint4 *)out, (u32)H, rows_per_block, alpha);                  \
    } while (0)
#define RC_LAUNCH_MF_ACT(F16)                                        \
    do {                                                             \
        if (activation == RC_ACT_ELU) RC_LAUNCH_MF(RC_ACT_ELU, F16);  \
        else if (activation == RC_ACT_RELU) RC_LAUNCH_MF(RC_ACT_RELU, F16); \
        else RC_LAUNCH_MF(RC_ACT_NONE, F16);                         \
    } while (0)
    if (table_is_f16) RC_LAUNCH_MF_ACT(true);
    else RC_LAUNCH_MF_ACT(false);
#undef RC_LAUNCH_MF_ACT
#undef RC_LAUNCH_MF
    return launch_status();
}

extern "C" int rc_head_bf16(const uint16_t *x, size_t n, size_t K, const uint16_t *w, const float *bias, size_t n_out,
                            float *out, int activation, float alpha, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(x && w && bias && out, RC_ERR_NULL);
    RC_REQUIRE(aligned16(x) && aligned16(out) && aligned16(w), RC_ERR_ALIGN);
    RC_REQUIRE(K == (size_t)kHeadK && n_out >= 1 && n_out <= kHeadMaxOut && activation >= RC_ACT_NONE &&
                   activation <= RC_ACT_ELU, RC_ERR_RANGE);
    // one 16-row tile per workgroup pass (four waves split its K); up to 8 workgroups per CU, grid-stride beyond
    const size_t tiles = ceil_div(n, (size_t)16);
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipStream_t s = (hipStream_t)stream;
    if (activation == RC_ACT_ELU)
        hipLaunchKernelGGL(k_head<RC_ACT_ELU>, dim3(grid), dim3(kBlock), 0, s, (const uint4 *)x, n, (const uint4 *)w, bias, out,
                           (u32)n_out, alpha);
    else if (activation == RC_ACT_RELU)
        hipLaunchKernelGGL(k_head<RC_ACT_RELU>, dim3(grid), dim3(kBlock), 0, s, (const uint4 *)x, n, (const uint4 *)w, bias, out,
                           (u32)n_out, alpha);
    else
        hipLaunchKernelGGL(k_head<RC_ACT_NONE>, dim3(grid), dim3(kBlock), 0, s, (const uint4 *)x, n, (const uint4 *)w, bias, out,
                           (u32)n_out, alpha);
    return launch_status();
}

// In-place activation of a bf16 tensor (the pass between two library GEMMs, which have no ELU epilogue):
// 16 bytes per lane, four independent chunks in flight per thread.
template <int ACT>
__global__ __launch_bounds__(kBlock) void k_act_bf16(uint4 *__restrict__ x, size_t n16, float alpha) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x; i0 < n16; i0 += 4 * stride) {
        uint4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (i0 + t * stride < n16) v[t] = x[i0 + t * stride];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (i0 + t * stride >= n16) continue;
            const u32 w[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
            u32 o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = pack_bf16(act_apply(__uint_as_float(w[d] << 16), ACT, alpha), act_apply(__uint_as_float(w[d] & 0xffff0000u), ACT, alpha));
            x[i0 + t * stride] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

extern "C" int rc_act_bf16_inplace(uint16_t *x, size_t n, int activation, float alpha, rc_stream_t stream) {
    if (n == 0 || activation == RC_ACT_NONE) return RC_OK;
    RC_REQUIRE(x != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(x) && n % 8 == 0, RC_ERR_ALIGN);
    RC_REQUIRE(activation == RC_ACT_RELU || activation == RC_ACT_ELU, RC_ERR_RANGE);
    const size_t n16 = n / 8;
    const unsigned grid = grid_for(ceil_div(n16, (size_t)4), kBlock, 256 * 16);
    if (activation == RC_ACT_ELU)
        hipLaunchKernelGGL(k_act_bf16<RC_ACT_ELU>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)x, n16, alpha);
    else
        hipLaunchKernelGGL(k_act_bf16<RC_ACT_RELU>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (uint4 *)x, n16, alpha);
    return launch_status();
}

// =================================================================================================
// fp32-accurate network on the f16 matrix cores ("f16x3 split", librubiks/model.py::SplitF32Net)
// =================================================================================================
// A float x is carried as two halves:  x = hi + lo * 2^-11,  hi = f16(x),  lo = f16((x - hi) * 2^11)  (22 significant
// bits; the 2^11 keeps lo out of f16's subnormal range).  A product of two such numbers is
//   hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b) + O(2^-22),
// three f16 MFMA products accumulated in fp32 (each f16 x f16 product is exact in fp32).  Measured against float64 on
// the hidden layer's shape, the result is CLOSER than the fp32 MFMA GEMM (mean |error| 4.0e-7 vs 1.1e-6 at |y| ~ 1)
// at 2.8 x its speed (round 2 probe).  The two kernels here produce the operands: the split one-hot
// input and the bias + activation + re-split between two layers.
constexpr u32 kHalfOne = 0x3C00u, kHalfScaleInv = 0x1000u;   // 1.0 and 2^-11 as IEEE half

// out[r][c] = onehot(r)[c] for c < 480 and onehot(r)[c - 480] * 2^-11 for c >= 480 (IEEE half, row pitch 960): the A operand
// whose product with [W_hi | W_lo] is the input layer in one GEMM.  One 16-byte chunk per thread, inside one cubie's block.
__global__ __launch_bounds__(kBlock) void k_oh_split_f16(const u8 *__restrict__ soa, size_t n, size_t stride, uint4 *__restrict__ out) {
    constexpr int kChunks = 2 * kOH / 8;   // 120 chunks of 8 halves per row
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < n * kChunks; idx += (size_t)gridDim.x * kBlock) {
        const size_t r = idx / kChunks;
        const u32 ch = (u32)(idx - r * kChunks);
        const u32 c = (ch * 8) % kOH;                      // column inside the 480-wide one-hot
        const u32 one = ch * 8 >= (u32)kOH ? kHalfScaleInv : kHalfOne;
        const u32 code = soa[(size_t)(c / 24) * stride + r];
        const u32 off = code - (c % 24);                   // position of the 1 inside this chunk, if < 8
        u32 w[4] = {0, 0, 0, 0};
        if (off < 8) w[off >> 1] = one << (16 * (off & 1));
        out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// y = act(c + corr_scale * c_corr + bias); hi/lo halves to out_hl[r][col] / out_hl[r][n_cols + col] (row pitch 2 n_cols), or
// y itself to out_f32.  The small correction product is added to the main one here, in fp32.
template <int ACT>
__global__ __launch_bounds__(kBlock) void k_split_act(const float4 *__restrict__ c, const float4 *__restrict__ c_corr, float corr_scale,
                                                      size_t n_rows, size_t n_cols, const float4 *__restrict__ bias, float alpha,
                                                      uint4 *__restrict__ out_hl, float4 *__restrict__ out_f32) {
    const size_t chunks = n_cols / 8;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < n_rows * chunks; idx += (size_t)gridDim.x * kBlock) {
        const size_t r = idx / chunks, ch = idx - r * chunks;
        float4 a0 = c[idx * 2], a1 = c[idx * 2 + 1];
        const float4 b0 = bias[ch * 2], b1 = bias[ch * 2 + 1];
        if (c_corr) {
            const float4 k0 = c_corr[idx * 2], k1 = c_corr[idx * 2 + 1];
            a0 = make_float4(a0.x + corr_scale * k0.x, a0.y + corr_scale * k0.y, a0.z + corr_scale * k0.z, a0.w + corr_scale * k0.w);
            a1 = make_float4(a1.x + corr_scale * k1.x, a1.y + corr_scale * k1.y, a1.z + corr_scale * k1.z, a1.w + corr_scale * k1.w);
        }
        float y[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
            y[e] = ACT == RC_ACT_RELU ? fmaxf(y[e], 0.f) : ACT == RC_ACT_ELU ? (y[e] > 0.f ? y[e] : alpha * expm1f(y[e])) : y[e];
        if (out_f32) {
            out_f32[idx * 2] = make_float4(y[0], y[1], y[2], y[3]);
            out_f32[idx * 2 + 1] = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (out_hl) {
            float hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hi[e] = round_to_half_f32(y[e]);
                lo[e] = (y[e] - hi[e]) * kSplitScale;
            }
            uint4 *row = out_hl + r * (2 * chunks);
            row[ch] = make_uint4(pack_half2(hi[0], hi[1]), pack_half2(hi[2], hi[3]), pack_half2(hi[4], hi[5]), pack_half2(hi[6], hi[7]));
            row[chunks + ch] = make_uint4(pack_half2(lo[0], lo[1]), pack_half2(lo[2], lo[3]), pack_half2(lo[4], lo[5]), pack_half2(lo[6], lo[7]));
        }
    }
}

// The general form: the layer's products arrive as n_partials partial sums (K chunks of rc_split_layer_f16, or the two library
// GEMMs), the first n_corr of which still carry the factor 2^11; optional skip connection and post-activation affine; the
// half-range flag of the split format (rc_split_layer_t in include/rubiks_hip.h).  Summation order p = 0, 1, ...: deterministic.
template <int ACT>
__global__ __launch_bounds__(kBlock) void k_split_reduce(const float4 *__restrict__ partials, size_t pstride4, int n_partials, int n_corr,
                                                         size_t n_rows, size_t n_cols, const float4 *__restrict__ bias,
                                                         const uint4 *__restrict__ res, float alpha, const float4 *__restrict__ post_scale,
                                                         const float4 *__restrict__ post_shift, uint4 *__restrict__ out_hl,
                                                         float4 *__restrict__ out_f32, int *__restrict__ flag) {
    const size_t chunks = n_cols / 8;
    bool out_of_range = false;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < n_rows * chunks; idx += (size_t)gridDim.x * kBlock) {
        const size_t r = idx / chunks, ch = idx - r * chunks;
        float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // up to eight partials' loads in flight before the first add (a loop of load-then-add pays one memory latency per partial: 32
        // partials of a 352-row layer took 13 us, latency-bound).  The adds keep the order p = 0, 1, ...; the factor 2^-11 that the sum
        // of the first n_corr partials takes is a multiplication by 2^-11 or by 1.0 (exact), so a group is one straight block of code.
        auto group = [&](auto width, int p0) {
            constexpr int G = decltype(width)::value;
            float4 a0[G], a1[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const size_t at = (size_t)(p0 + u) * pstride4 + idx * 2;
                a0[u] = partials[at], a1[u] = partials[at + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const float f = p0 + u == n_corr ? 1.0f / kSplitScale : 1.0f;
                y[0] = y[0] * f + a0[u].x, y[1] = y[1] * f + a0[u].y, y[2] = y[2] * f + a0[u].z, y[3] = y[3] * f + a0[u].w;
                y[4] = y[4] * f + a1[u].x, y[5] = y[5] * f + a1[u].y, y[6] = y[6] * f + a1[u].z, y[7] = y[7] * f + a1[u].w;
            }
        };
        int p = 0;
        for (; p + 8 <= n_partials; p += 8) group(std::integral_constant<int, 8>{}, p);
        if (p + 4 <= n_partials) group(std::integral_constant<int, 4>{}, p), p += 4;
        if (p + 2 <= n_partials) group(std::integral_constant<int, 2>{}, p), p += 2;
        if (p < n_partials) group(std::integral_constant<int, 1>{}, p);
        if (n_corr >= n_partials) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] *= (1.0f / kSplitScale);
        }
        const float4 b0 = bias[ch * 2], b1 = bias[ch * 2 + 1];
        y[0] += b0.x, y[1] += b0.y, y[2] += b0.z, y[3] += b0.w, y[4] += b1.x, y[5] += b1.y, y[6] += b1.z, y[7] += b1.w;
        if (res) {
            const uint4 rh = res[r * (2 * chunks) + ch], rl = res[r * (2 * chunks) + chunks + ch];
            const u32 hw[4] = {rh.x, rh.y, rh.z, rh.w}, lw[4] = {rl.x, rl.y, rl.z, rl.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[e >> 1] >> (16 * (e & 1))));
                const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(lw[e >> 1] >> (16 * (e & 1))));
                y[e] += hi + lo * (1.0f / kSplitScale);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            y[e] = ACT == RC_ACT_RELU ? fmaxf(y[e], 0.f) : ACT == RC_ACT_ELU ? (y[e] > 0.f ? y[e] : alpha * expm1f(y[e])) : y[e];
        if (post_scale) {
            const float4 s0 = post_scale[ch * 2], s1 = post_scale[ch * 2 + 1], t0 = post_shift[ch * 2], t1 = post_shift[ch * 2 + 1];
            y[0] = y[0] * s0.x + t0.x, y[1] = y[1] * s0.y + t0.y, y[2] = y[2] * s0.z + t0.z, y[3] = y[3] * s0.w + t0.w;
            y[4] = y[4] * s1.x + t1.x, y[5] = y[5] * s1.y + t1.y, y[6] = y[6] * s1.z + t1.z, y[7] = y[7] * s1.w + t1.w;
        }
        if (out_f32) {
            out_f32[idx * 2] = make_float4(y[0], y[1], y[2], y[3]);
            out_f32[idx * 2 + 1] = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (out_hl) {
            float hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                out_of_range |= !(fabsf(y[e]) <= 65504.0f);
                hi[e] = round_to_half_f32(y[e]);
                lo[e] = (y[e] - hi[e]) * kSplitScale;
            }
            uint4 *row = out_hl + r * (2 * chunks);
            row[ch] = make_uint4(pack_half2(hi[0], hi[1]), pack_half2(hi[2], hi[3]), pack_half2(hi[4], hi[5]), pack_half2(hi[6], hi[7]));
            row[chunks + ch] = make_uint4(pack_half2(lo[0], lo[1]), pack_half2(lo[2], lo[3]), pack_half2(lo[4], lo[5]), pack_half2(lo[6], lo[7]));
        }
    }
    if (flag && out_of_range) atomicOr(flag, 1);
}

// Last hidden activation + output layer of the split network in one pass (the fp32 counterpart of k_head):
//   y = act(c + corr_scale * c_corr + bias_h)   [n][K] fp32, never written;   out[i][o] = bias_o[o] + sum_k w[o][k] y[i][k]
// on the exact fp32 MFMA (v_mfma_f32_16x16x4_f32 == an fmaf chain).  One 16-row tile per workgroup pass, K split over the
// four waves, whose partial 16 x 16 products meet in LDS.  A lane loads 16-byte chunks: chunk (j, g) of a row holds
// k = 16 j + 4 g .. + 3, and MFMA step s of block j contracts the s-th element of every chunk -- a permutation of k that
// A (activations) and B (weights) share.  It replaces rc_split_act_f16's fp32 output + the library's 13-wide fp32 GEMM:
// the [n][K] activation matrix (46 MB at n = 11 264) is neither written nor read back.
template <int ACT, int KB>
__global__ __launch_bounds__(kBlock) void k_head_split(const float4 *__restrict__ c, const float4 *__restrict__ c_corr, float corr_scale,
                                                       size_t n, const float4 *__restrict__ bias_h, float alpha, const float4 *__restrict__ w,
                                                       const float *__restrict__ bias_o, u32 n_out, float *__restrict__ out) {
    __shared__ float s_part[kBlock / kWave][16][17];
    constexpr int K = 64 * KB, K4 = K / 4;          // floats, float4 chunks per row; a wave owns K / 4 consecutive floats = KB blocks of 16
    const u32 tid = threadIdx.x, wv = tid / kWave, lane = tid & (kWave - 1), r = lane & 15, g = lane >> 4;
    const u32 chunk0 = wv * (K4 / 4) + g;           // + 4 j: the lane's chunk of block j
    float4 wf[KB], bh[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        wf[j] = r < n_out ? w[(size_t)r * K4 + chunk0 + 4 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
        bh[j] = bias_h[chunk0 + 4 * j];
    }
    const size_t n_tiles = ceil_div(n, (size_t)16);
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t row = tile * 16 + r;
        const size_t off = (row < n ? row : n - 1) * K4 + chunk0;   // rows past the end: clamped, never stored
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j0 = 0; j0 < KB; j0 += 4) {
            float4 a[4], k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = c[off + 4 * (j0 + u)];
                k4[u] = c_corr ? c_corr[off + 4 * (j0 + u)] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 b = bh[j0 + u];
                float y[4] = {a[u].x + corr_scale * k4[u].x + b.x, a[u].y + corr_scale * k4[u].y + b.y, a[u].z + corr_scale * k4[u].z + b.z,
                              a[u].w + corr_scale * k4[u].w + b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = act_value<ACT>(y[e], alpha);   // branch-free; libm's expm1f cost a third of this kernel's time
                const float4 wv4 = wf[j0 + u];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[0], wv4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[1], wv4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[2], wv4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(y[3], wv4.w, acc, 0, 0, 0);
            }
        }
        __syncthreads();   // the previous tile's partial sums have been read
#pragma unroll
        for (int i = 0; i < 4; ++i) s_part[wv][g * 4 + i][r] = acc[i];   // D: lane holds output r of rows 4 g .. 4 g + 3
        __syncthreads();
        {
            const u32 orow = tid >> 4, o = tid & 15;
            const float v = (s_part[0][orow][o] + s_part[1][orow][o]) + (s_part[2][orow][o] + s_part[3][orow][o]);
            const size_t grow = tile * 16 + orow;
            if (grow < n) out[grow * kHeadMaxOut + o] = (o < n_out) ? v + bias_o[o] : 0.f;
        }
    }
}

// =================================================================================================
// Input layer of the fp32-accurate split network on the matrix cores (SplitF32Net, below):
//   y = act(onehot W_hi^T + 2^-11 onehot W_lo^T + b),  written as the two halves [hi(y) | lo(y)] the next layer's GEMMs read.
// Same structure as k_first_layer_mfma -- one-hot A fragments generated from the cube codes, the W1 slice in LDS, a lane's
// accumulators are adjacent output columns -- with three differences: (1) 60 k-steps instead of 30: the second 30 multiply the
// one-hot scaled by 2^-11 (exact in half) with the W_lo table, into the same fp32 accumulators; (2) a workgroup owns 64 columns
// (hi + lo slices = 122 KiB of LDS) and a wave holds TWO 32-state tiles, so every B fragment read from LDS feeds two MFMAs
// (the 128-column kernel is bound by its LDS reads, not by the matrix cores); (3) the epilogue re-splits into halves.
// It replaces rc_oh_split_f16 + the K = 960 library GEMM + rc_split_act_f16 of the input layer.
// =================================================================================================
constexpr int kSpCols = 64;
constexpr int kSpSub = 2;   // 32-state tiles a wave holds at once (they share every B fragment read from LDS); 3 / 4 measured slower

template <int ACT>
__global__ __launch_bounds__(kMfWaves * kWave) void k_first_layer_split(const u8 *__restrict__ soa, size_t n, size_t stride,
                                                                      const uint4 *__restrict__ w_hi, const uint4 *__restrict__ w_lo,
                                                                      const float *__restrict__ bias, u32 *__restrict__ out, u32 H,
                                                                      u32 rows_per_block, float alpha, int *__restrict__ range_flag) {
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    constexpr int kTables = 2;
    bool out_of_range = false;
    unsigned char *wslice = lds;                                                       // [kTables][64 slots][976 B]
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);                         // wave-uniform: row bases and output pointers stay scalar
    uint4 *onehot = reinterpret_cast<uint4 *>(lds + kTables * kSpCols * kMfPitch);     // [kTables][9] A fragments: 1.0 / 2^-11 at position p
    const u32 col_tiles = H / kSpCols;
    const u32 ct = blockIdx.x % col_tiles, rg = blockIdx.x / col_tiles;
    const size_t row_lo = (size_t)rg * rows_per_block;
    if (row_lo >= n) return;
    const size_t row_hi = (row_lo + rows_per_block < n) ? row_lo + rows_per_block : n;
    {   // 2 x 64 columns x 60 chunks of 16 B = 7 680 chunks / 512 threads = 15 per thread.  Column g of the slice goes to slot
        // (g % 2) * 32 + g / 2: MFMA column tile c, lane r owns column 2 r + c.
        constexpr int kBatch = 5, kPer = kTables * kSpCols * 60 / (kMfWaves * kWave) / kBatch * kBatch;   // 15
#pragma unroll
        for (int b0 = 0; b0 < kPer; b0 += kBatch) {
            uint4 tmp[kBatch];
#pragma unroll
            for (int t = 0; t < kBatch; ++t) {
                const u32 i = tid + (b0 + t) * (kMfWaves * kWave), tbl = i / (kSpCols * 60), j = i % (kSpCols * 60);
                tmp[t] = (tbl ? w_lo : w_hi)[(size_t)(ct * kSpCols + j / 60) * 60 + j % 60];
            }
#pragma unroll
            for (int t = 0; t < kBatch; ++t) {
                const u32 i = tid + (b0 + t) * (kMfWaves * kWave), tbl = i / (kSpCols * 60), j = i % (kSpCols * 60);
                const u32 g = j / 60, slot = (g & 1) * 32 + (g >> 1);
                *reinterpret_cast<uint4 *>(wslice + (tbl * kSpCols + slot) * kMfPitch + (j % 60) * 16) = tmp[t];
            }
        }
        static_assert(kPer * kMfWaves * kWave == kTables * kSpCols * 60, "the staging loop covers the slices exactly");
    }
    if (tid < 9 * kTables) {
        u32 w[4] = {0, 0, 0, 0};
        const u32 p = tid % 9, one = tid < 9 ? kHalfOne : kHalfScaleInv;
        if (p < 8) w[p >> 1] = (p & 1) ? one << 16 : one;
        onehot[tid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const u32 r = lane & 31, h = lane >> 5;
    float b[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) b[c] = bias[ct * kSpCols + 2 * r + c];
    __syncthreads();

    const u32 off_a = h ? 8u : 0u, off_c = h ? 16u : 8u;
    constexpr size_t kStep = (size_t)kMfWaves * kSpSub * kMfTile;
    // A pass of a wave = [60 k-steps of MFMAs] then [epilogue: ELU + re-split of 64 values per lane, ~1 600 VALU instructions].
    // PMC (profiles/r3_first_layer_split_pmc.txt): the matrix pipe is busy 40 % of the cycles, 9 VALU instructions per MFMA,
    // LDS wait negligible.  Measured and dropped in round 3 (profiles/r3_first_layer_variants.txt): the two waves of a SIMD
    // alternating between the two stages by barriers (135 against 119 us per call), 3 or 4 state tiles per wave (132 / 149 us),
    // the one-hot fragment built in registers instead of read from LDS (+ 43 %).
    const u32 n_pass = (u32)((row_hi - row_lo + kStep - 1) / kStep);
    for (u32 pass = 0; pass < n_pass; ++pass) {
        const size_t t0 = row_lo + (size_t)pass * kStep + (size_t)wave * kSpSub * kMfTile;
        if (t0 >= row_hi) break;
        u32 pk[kSpSub][5];
#pragma unroll
        for (int u = 0; u < kSpSub; ++u) {
            const size_t row = t0 + (size_t)u * kMfTile + r;
            const u8 *p = soa + (row < n ? row : n - 1);
            u32 raw[kPlanes];
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) raw[j] = p[(size_t)j * stride];
#pragma unroll
            for (int q = 0; q < 5; ++q)
                pk[u][q] = (raw[4 * q] & 31u) | (raw[4 * q + 1] & 31u) << 8 | (raw[4 * q + 2] & 31u) << 16 | (raw[4 * q + 3] & 31u) << 24;
        }
        auto code = [&](int u, int j) -> u32 { return (pk[u][j >> 2] >> (8 * (j & 3))) & 0xffu; };
        f32x16 acc[kSpSub][2];
#pragma unroll
        for (int u = 0; u < kSpSub; ++u)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[u][c][i] = b[c];
        // two passes of 30 k-steps: table 0 = W_hi with one-hot 1.0, table 1 = W_lo with one-hot 2^-11 (the pass loop is
        // NOT unrolled: unrolling all 60 steps overflows the 256 registers a wave has at two waves per SIMD)
#pragma unroll 1
        for (int tbl = 0; tbl < kTables; ++tbl) {
            const uint4 *frag = onehot + 9 * tbl;
            const unsigned char *bbase = wslice + ((size_t)tbl * kSpCols + r) * kMfPitch + 16 * h;   // + c * 32 * pitch + 32 * ks
            auto a_frag = [&](int u, int ks) -> uint4 {
                const int m2 = ks / 3, ph = ks % 3;
                const u32 pos = ph == 0 ? code(u, 2 * m2) - off_a : ph == 1 ? (h ? code(u, 2 * m2 + 1) : code(u, 2 * m2) - 16u)
                                                                             : code(u, 2 * m2 + 1) - off_c;
                return frag[min(pos, 8u)];
            };
            uint4 a_cur[kSpSub], b_cur[2];
#pragma unroll
            for (int u = 0; u < kSpSub; ++u) a_cur[u] = a_frag(u, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) b_cur[c] = *reinterpret_cast<const uint4 *>(bbase + c * 32 * kMfPitch);
#pragma unroll
            for (int ks = 0; ks < 30; ++ks) {
                uint4 a_nxt[kSpSub], b_nxt[2] = {b_cur[0], b_cur[1]};
#pragma unroll
                for (int u = 0; u < kSpSub; ++u) a_nxt[u] = a_cur[u];
                if (ks + 1 < 30) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) b_nxt[c] = *reinterpret_cast<const uint4 *>(bbase + c * 32 * kMfPitch + 32 * (ks + 1));
#pragma unroll
                    for (int u = 0; u < kSpSub; ++u) a_nxt[u] = a_frag(u, ks + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int u = 0; u < kSpSub; ++u)
                        acc[u][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_cur[u]),
                                                                           __builtin_bit_cast(f16x8, b_cur[c]), acc[u][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < kSpSub; ++u) a_cur[u] = a_nxt[u];
#pragma unroll
                for (int c = 0; c < 2; ++c) b_cur[c] = b_nxt[c];
            }
        }
        // epilogue: activation, then the lane's columns 2 r, 2 r + 1 of one state as two 4-byte stores: the pair of hi halves and
        // the pair of lo halves into the [hi | lo] row of the state (row pitch 2 H halves = H dwords).  The kernel is bound by its
        // VALU port, not by the matrix pipe (profiles/r3_first_layer_split_pmc.txt), so the epilogue is kept lean: wave-uniform
        // output pointers + ONE running 32-bit offset per lane, no per-row tests when all of the wave's rows exist (every pass but a
        // workgroup's last), branch-free ELU, the split in 5 operations per pair.
        const size_t lim = row_hi < n ? row_hi : n;
        u32 *o_hi = out + t0 * (size_t)H + (ct * kSpCols) / 2;    // row t0, the workgroup's first column pair
        u32 *o_lo = o_hi + H / 2;
        if (t0 + (size_t)kSpSub * kMfTile <= lim) {
            u32 off = 4 * h * H + r;                                // this lane's row 4 h of the first tile, its column pair
#pragma unroll
            for (int u = 0; u < kSpSub; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float y0 = act_value<ACT>(acc[u][0][i], alpha), y1 = act_value<ACT>(acc[u][1][i], alpha);
                    out_of_range |= !(fabsf(y0) <= 65504.0f) || !(fabsf(y1) <= 65504.0f);
                    const SplitPair sp = split_pair(y0, y1);
                    o_hi[off] = sp.hi;
                    o_lo[off] = sp.lo;
                    off += ((i & 3) == 3 ? 5u : 1u) * H;           // rows (i & 3) + 8 (i >> 2); the next tile starts 32 rows on: 4 + 8 * 3 + 5 = 33 - 1
                }
        } else {
#pragma unroll
            for (int u = 0; u < kSpSub; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const size_t row = t0 + (size_t)u * kMfTile + (i & 3) + 8 * (i >> 2) + 4 * h;
                    const float y0 = act_value<ACT>(acc[u][0][i], alpha), y1 = act_value<ACT>(acc[u][1][i], alpha);
                    if (row < lim) {
                        out_of_range |= !(fabsf(y0) <= 65504.0f) || !(fabsf(y1) <= 65504.0f);
                        const SplitPair sp = split_pair(y0, y1);
                        u32 *orow = out + row * H;
                        orow[(ct * kSpCols) / 2 + r] = sp.hi;
                        orow[H / 2 + (ct * kSpCols) / 2 + r] = sp.lo;
                    }
                }
        }
    }
    if (range_flag && out_of_range) atomicOr(range_flag, 1);
}

// =================================================================================================
// The input layer as what it is: a one-hot row times W1 is the SUM OF 20 ROWS of W1^T, one per cubie (480 = 20 x 24 codes),
//   y = act(bias + sum_j W1^T[24 j + code_j]),   written as the two halves [hi(y) | lo(y)]
// -- 20 fp32 additions per output instead of 960 multiply-adds on the matrix cores (k_first_layer_split: 60 k-steps of
// mostly-zero one-hot fragments, VALU-bound on building those fragments), in plain fp32 from the fp32 table (no hi / lo split of
// the weights: the sum is exact to fp32 rounding, in the fixed order j = 0 .. 19 behind the bias -- whatever the form below).
// A workgroup owns 64 columns: its slice of the table, [480][64] fp32 = 120 KiB, lives in LDS.  A lane takes C adjacent columns
// of one state (C / 4 ds_read_b128 per cubie; with C = 4 the 16 lanes of a state read the 256 bytes of a row, all 64 banks once),
// a wave 64 C / 64 states at a time; the codes of the next states are requested while these are summed.  Like every elementwise
// part of the network the loop is bound by instruction ISSUE (profiles/r6_input_layer_ab.txt: stores, LDS reads, activation each
// cost their share of instructions, no resource is full), so the two forms trade per-output overhead against granularity:
//   <16 waves, C = 4>: four waves per SIMD, units of 4 states -- small batches (352 rows: 9 us);
//   < 8 waves, C = 8>: half the code loads, address computations and stores per output -- large batches (11 264 rows: 95 us).
// =================================================================================================
constexpr int kGaCols = 64, kGaRows = 480;
#ifndef RUBIKS_GATHER_ABLATE   // diagnostic builds (WRONG RESULTS): 1 no stores, 2 no LDS reads, 4 no code loads, 8 no activation
#define RUBIKS_GATHER_ABLATE 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ACT, int WAVES, int C>
__global__ __launch_bounds__(WAVES * kWave) void k_first_layer_gather(const u8 *__restrict__ soa, size_t n, size_t stride,
                                                                     const float4 *__restrict__ w_rows, const float *__restrict__ bias,
                                                                     unsigned char *__restrict__ out, u32 H, u32 rows_per_block, float alpha,
                                                                     int *__restrict__ range_flag) {
    static_assert(C == 4 || C == 8, "4 or 8 columns per lane");
    constexpr int R = C / 4;                  // 16-byte reads per table row and lane
    constexpr u32 kLanes = kGaCols / C;       // lanes per state
    constexpr u32 kStates = kWave / kLanes;   // states per wave and pass
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];   // [480][64] fp32
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 col_tiles = H / kGaCols, ct = blockIdx.x % col_tiles, rg = blockIdx.x / col_tiles;
    const size_t row_lo = (size_t)rg * rows_per_block;
    if (row_lo >= n) return;
    const size_t row_hi = (row_lo + rows_per_block < n) ? row_lo + rows_per_block : n;
    {   // the slice: 480 rows x 16 chunks of 16 B = 7 680 chunks, four requests in flight per thread at a time
        constexpr u32 kChunks = kGaRows * (kGaCols / 4), kThreads = WAVES * kWave, kBatch = 4;
        for (u32 i0 = tid; i0 < kChunks; i0 += kBatch * kThreads) {
            float4 tmp[kBatch];
#pragma unroll
            for (u32 b = 0; b < kBatch; ++b) {
                const u32 i = i0 + b * kThreads;
                if (i < kChunks) tmp[b] = w_rows[(size_t)(i >> 4) * (H / 4) + ct * (kGaCols / 4) + (i & 15)];
            }
#pragma unroll
            for (u32 b = 0; b < kBatch; ++b) {
                const u32 i = i0 + b * kThreads;
                if (i < kChunks) reinterpret_cast<float4 *>(lds)[i] = tmp[b];
            }
        }
    }
    const u32 u = lane % kLanes, sub = lane / kLanes;   // the lane's columns of the tile: C u .. C u + C - 1; its state among the wave's
    const u32 col = ct * kGaCols + C * u;
    float4 b4[R];
#pragma unroll
    for (int r = 0; r < R; ++r) b4[r] = *reinterpret_cast<const float4 *>(bias + col + 4 * r);
    bool out_of_range = false;
    // The codes: plane j of state t is byte j stride + t of the SoA block -- a buffer load with the state in the vector offset and the
    // plane in the scalar offset, no 64-bit address arithmetic per byte.  Rows past the end are clamped (computed, not stored).
    const __amdgpu_buffer_rsrc_t planes = __builtin_amdgcn_make_buffer_rsrc((void *)soa, 0, (int)(u32)(kPlanes * stride), 0x00020000);
    const u32 last = (u32)(n - 1), pitch = (u32)stride;
    auto codes_of = [&](size_t t, u32 (&c)[kPlanes]) {
        const u32 tv = t < n ? (u32)t : last;
#pragma unroll
        for (int j = 0; j < kPlanes; ++j) c[j] = (RUBIKS_GATHER_ABLATE & 4) ? (tv + j) % 24u : (u32)__builtin_amdgcn_raw_buffer_load_b8(planes, tv, (u32)j * pitch, 0);
    };
    // row (24 j + code) of the slice at byte (24 j + code) 256 + 4 C u: the code shifted onto one of two bases, the cubie (and the
    // lane's second 16 bytes) in the instruction's offset field (16 bits: cubies 0 .. 10 on the first base, 11 .. 19 on the second)
    const u32 base0 = u * (4 * C), base1 = u * (4 * C) + 11 * 24 * (kGaCols * 4);
    __syncthreads();
    constexpr size_t kStep = (size_t)WAVES * kStates;   // states per pass of the workgroup
    size_t t = row_lo + (size_t)wave * kStates + sub;
    u32 code[kPlanes], code_next[kPlanes];
    codes_of(t, code);
    for (size_t t0 = row_lo + (size_t)wave * kStates; t0 < row_hi; t0 += kStep, t += kStep) {
        const bool more = t0 + kStep < row_hi;        // wave-uniform
        // the rows are requested a batch at a time before the first of them is added (left to itself the compiler waits for each read
        // in turn: an LDS latency per row), the next states' codes in between, the additions in the order j = 0 .. 19
        f32x2 acc[2 * R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[2 * r] = f32x2{b4[r].x, b4[r].y}, acc[2 * r + 1] = f32x2{b4[r].z, b4[r].w};
        constexpr int kBatch = 10 / R;
#pragma unroll
        for (int part = 0; part < kPlanes / kBatch; ++part) {
            float4 v[kBatch][R];
#pragma unroll
            for (int i = 0; i < kBatch; ++i) {
                const int j = part * kBatch + i;
                const u32 at = (code[j] << 8) + (j < 11 ? base0 : base1);   // (codes are 0 .. 23; a byte beyond would read zeros past the slice, never fault)
                const unsigned char *rowp = lds + at + (j < 11 ? j : j - 11) * 24 * (kGaCols * 4);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    v[i][r] = (RUBIKS_GATHER_ABLATE & 2) ? make_float4(__uint_as_float(at), 1.f, 2.f, 3.f) : *reinterpret_cast<const float4 *>(rowp + 16 * r);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (part == 0 && more) codes_of(t + kStep, code_next);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kBatch; ++i)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[2 * r] += f32x2{v[i][r].x, v[i][r].y};
                    acc[2 * r + 1] += f32x2{v[i][r].z, v[i][r].w};
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        bool bad = false;
        SplitPair sp[2 * R];
#pragma unroll
        for (int p = 0; p < 2 * R; ++p) {
            if (!(RUBIKS_GATHER_ABLATE & 8)) acc[p] = act_value2<ACT>(acc[p], alpha);
            bad |= !(fabsf(acc[p].x) <= 65504.0f);
            bad |= !(fabsf(acc[p].y) <= 65504.0f);
            sp[p] = split_pair(acc[p].x, acc[p].y);
        }
        // 16-byte stores (8-byte stores are bound by their issue, not by bandwidth).  C = 8: the lane's eight hi halves, its eight lo
        // halves.  C = 4: neighbouring lanes (columns 8 i .. 8 i + 3 and 8 i + 4 .. 8 i + 7 of one state) swap halves through DPP, the even
        // lane stores the hi halves of both, the odd lane the lo halves of both.
        const bool live = (RUBIKS_GATHER_ABLATE & 1) ? (t < row_hi && acc[0].x == 1.2345e-30f) : t < row_hi;
        unsigned char *orow = out + t * ((size_t)H * 4);   // row pitch 2 H halves
        if (C == 8) {
            if (live) {
                out_of_range |= bad;
                *reinterpret_cast<uint4 *>(orow + (size_t)col * 2) = make_uint4(sp[0].hi, sp[1].hi, sp[2 * R - 2].hi, sp[2 * R - 1].hi);
                *reinterpret_cast<uint4 *>(orow + ((size_t)H + col) * 2) = make_uint4(sp[0].lo, sp[1].lo, sp[2 * R - 2].lo, sp[2 * R - 1].lo);
            }
        } else {
            constexpr int kSwapPairs = 0xB1;   // quad_perm [1, 0, 3, 2]
            const u32 n_hi0 = (u32)__builtin_amdgcn_mov_dpp((int)sp[0].hi, kSwapPairs, 0xF, 0xF, true), n_hi1 = (u32)__builtin_amdgcn_mov_dpp((int)sp[1].hi, kSwapPairs, 0xF, 0xF, true);
            const u32 n_lo0 = (u32)__builtin_amdgcn_mov_dpp((int)sp[0].lo, kSwapPairs, 0xF, 0xF, true), n_lo1 = (u32)__builtin_amdgcn_mov_dpp((int)sp[1].lo, kSwapPairs, 0xF, 0xF, true);
            if (live) {
                out_of_range |= bad;
                const bool odd = lane & 1u;
                const uint4 both = odd ? make_uint4(n_lo0, n_lo1, sp[0].lo, sp[1].lo) : make_uint4(sp[0].hi, sp[1].hi, n_hi0, n_hi1);
                *reinterpret_cast<uint4 *>(orow + (odd ? (size_t)H + col - 4 : (size_t)col) * 2) = both;
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < kPlanes; ++j) code[j] = code_next[j];
        }
    }
    if (range_flag && out_of_range) atomicOr(range_flag, 1);
}

extern "C" int rc_oh_split_f16(const int8_t *soa, size_t n, size_t stride, uint16_t *out, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(out != nullptr, RC_ERR_NULL);
    RC_REQUIRE(aligned16(out), RC_ERR_ALIGN);
    hipLaunchKernelGGL(k_oh_split_f16, dim3(grid_for(n * (2 * kOH / 8), kBlock, 256 * 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const u8 *)soa, n, stride, (uint4 *)out);
    return launch_status();
}

extern "C" int rc_split_act_f16(const float *c, const float *c_corr, float corr_scale, size_t n_rows, size_t n_cols, const float *bias,
                                int activation, float alpha, uint16_t *out_hi_lo, float *out_f32, rc_stream_t stream) {
    if (n_rows == 0 || n_cols == 0) return RC_OK;
    RC_REQUIRE(c && bias && (out_hi_lo || out_f32), RC_ERR_NULL);
    RC_REQUIRE(aligned16(c) && aligned16(c_corr) && aligned16(bias) && aligned16(out_hi_lo) && aligned16(out_f32) && n_cols % 8 == 0, RC_ERR_ALIGN);
    RC_REQUIRE(activation >= RC_ACT_NONE && activation <= RC_ACT_ELU, RC_ERR_RANGE);
    const dim3 grid(grid_for(n_rows * (n_cols / 8), kBlock, 256 * 16)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH_SPLIT(ACT)                                                                                                  \
    hipLaunchKernelGGL(k_split_act<ACT>, grid, block, 0, s, (const float4 *)c, (const float4 *)c_corr, corr_scale, n_rows, n_cols, \
                       (const float4 *)bias, alpha,                                                                            \
                       (uint4 *)out_hi_lo, (float4 *)out_f32)
    if (activation == RC_ACT_ELU) RC_LAUNCH_SPLIT(RC_ACT_ELU);
    else if (activation == RC_ACT_RELU) RC_LAUNCH_SPLIT(RC_ACT_RELU);
    else RC_LAUNCH_SPLIT(RC_ACT_NONE);
#undef RC_LAUNCH_SPLIT
    return launch_status();
}

extern "C" int rc_split_reduce_f16(const float *partials, size_t partial_stride, int n_partials, int n_corr, size_t n_rows, size_t n_cols,
                                   const float *bias, const uint16_t *residual_hi_lo, int activation, float alpha, const float *post_scale,
                                   const float *post_shift, uint16_t *out_hi_lo, float *out_f32, int32_t *range_flag, rc_stream_t stream) {
    if (n_rows == 0 || n_cols == 0) return RC_OK;
    RC_REQUIRE(partials && bias && (out_hi_lo || out_f32), RC_ERR_NULL);
    RC_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), RC_ERR_NULL);
    RC_REQUIRE(aligned16(partials) && aligned16(bias) && aligned16(residual_hi_lo) && aligned16(post_scale) && aligned16(post_shift) &&
                   aligned16(out_hi_lo) && aligned16(out_f32) && n_cols % 8 == 0 && partial_stride % 4 == 0, RC_ERR_ALIGN);
    RC_REQUIRE(activation >= RC_ACT_NONE && activation <= RC_ACT_ELU && n_partials >= 1 && n_partials <= 64 && n_corr >= 0 &&
                   n_corr <= n_partials && (n_partials == 1 || partial_stride >= n_rows * n_cols), RC_ERR_RANGE);
    const dim3 grid(grid_for(n_rows * (n_cols / 8), kBlock, 256 * 16)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH_RED(ACT)                                                                                                       \
    hipLaunchKernelGGL(k_split_reduce<ACT>, grid, block, 0, s, (const float4 *)partials, partial_stride / 4, n_partials, n_corr, n_rows, \
                       n_cols, (const float4 *)bias, (const uint4 *)residual_hi_lo, alpha, (const float4 *)post_scale,                \
                       (const float4 *)post_shift, (uint4 *)out_hi_lo, (float4 *)out_f32, (int *)range_flag)
    if (activation == RC_ACT_ELU) RC_LAUNCH_RED(RC_ACT_ELU);
    else if (activation == RC_ACT_RELU) RC_LAUNCH_RED(RC_ACT_RELU);
    else RC_LAUNCH_RED(RC_ACT_NONE);
#undef RC_LAUNCH_RED
    return launch_status();
}

extern "C" int rc_head_split_f32(const float *c, const float *c_corr, float corr_scale, size_t n, size_t K, const float *bias_h,
                                 int activation, float alpha, const float *w, const float *bias_o, size_t n_out, float *out,
                                 rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(c && bias_h && w && bias_o && out, RC_ERR_NULL);
    RC_REQUIRE(aligned16(c) && aligned16(c_corr) && aligned16(bias_h) && aligned16(w), RC_ERR_ALIGN);
    RC_REQUIRE((K == 512 || K == 1024) && n_out >= 1 && n_out <= (size_t)kHeadMaxOut && activation >= RC_ACT_NONE && activation <= RC_ACT_ELU,
               RC_ERR_RANGE);
    const unsigned grid = grid_for(ceil_div(n, (size_t)16) * kBlock, kBlock, 256 * 4);
    hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH_HS(ACT, KB)                                                                                                  \
    hipLaunchKernelGGL((k_head_split<ACT, KB>), dim3(grid), dim3(kBlock), 0, s, (const float4 *)c, (const float4 *)c_corr, corr_scale, n, \
                       (const float4 *)bias_h, alpha, (const float4 *)w, bias_o, (u32)n_out, out)
#define RC_LAUNCH_HS_K(ACT)                   \
    do {                                      \
        if (K == 1024) RC_LAUNCH_HS(ACT, 16); \
        else RC_LAUNCH_HS(ACT, 8);            \
    } while (0)
    if (activation == RC_ACT_ELU) RC_LAUNCH_HS_K(RC_ACT_ELU);
    else if (activation == RC_ACT_RELU) RC_LAUNCH_HS_K(RC_ACT_RELU);
    else RC_LAUNCH_HS_K(RC_ACT_NONE);
#undef RC_LAUNCH_HS_K
#undef RC_LAUNCH_HS
    return launch_status();
}

extern "C" int rc_first_layer_split_f16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w_hi, const uint16_t *w_lo,
                                        const float *bias, uint16_t *out_hi_lo, size_t H, int activation, float alpha,
                                        rc_stream_t stream) {
    return rc_first_layer_split_flag_f16(soa, n, stride, w_hi, w_lo, bias, out_hi_lo, H, activation, alpha, nullptr, stream);
}

extern "C" int rc_first_layer_split_flag_f16(const int8_t *soa, size_t n, size_t stride, const uint16_t *w_hi, const uint16_t *w_lo,
                                             const float *bias, uint16_t *out_hi_lo, size_t H, int activation, float alpha,
                                             int32_t *range_flag, rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(w_hi && w_lo && bias && out_hi_lo, RC_ERR_NULL);
    RC_REQUIRE(aligned16(w_hi) && aligned16(w_lo) && aligned16(out_hi_lo), RC_ERR_ALIGN);
    RC_REQUIRE(H >= (size_t)kSpCols && H % kSpCols == 0 && activation >= RC_ACT_NONE && activation <= RC_ACT_ELU, RC_ERR_RANGE);
    const u32 col_tiles = (u32)(H / kSpCols);
    u32 row_groups = (256 + col_tiles - 1) / col_tiles;   // ~one workgroup per CU (the LDS slice allows no more)
    constexpr u32 kRowsPerPass = kMfWaves * kSpSub * kMfTile;
    const size_t per_group = ceil_div(n, row_groups);   // (small batches: units of one wave's rows, as in rc_first_layer_mfma_bf16)
    u32 rows_per_block = (u32)round_up(per_group, per_group < kRowsPerPass ? (size_t)kSpSub * kMfTile : (size_t)kRowsPerPass);
    row_groups = (u32)ceil_div(n, rows_per_block);
    const size_t lds_bytes = (size_t)2 * kSpCols * kMfPitch + 2 * 9 * 16;
    const dim3 grid(col_tiles * row_groups), block(kMfWaves * kWave);
    hipStream_t s = (hipStream_t)stream;
#define RC_LAUNCH_SP(ACT)                                                                                             \
    do {                                                                                                              \
        static std::atomic<unsigned long long> attr_set{0};                                                           \
        int dev_ = 0;                                                                                                 \
        (void)hipGetDevice(&dev_);                                                                                    \
        if (!((attr_set.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ull)) {                                     \
            hipError_t e = hipFuncSetAttribute((const void *)k_first_layer_split<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds_bytes);                                                       \
            if (e != hipSuccess) return hip_rc(e);                                                                    \
            attr_set.fetch_or(1ull << (dev_ & 63), std::memory_order_release);                                        \
        }                                                                                                             \
        hipLaunchKernelGGL((k_first_layer_split<ACT>), grid, block, lds_bytes, s, (const u8 *)soa, n, stride, (const uint4 *)w_hi, \
                           (const uint4 *)w_lo, bias, (u32 *)out_hi_lo, (u32)H, rows_per_block, alpha, (int *)range_flag); \
    } while (0)
    if (activation == RC_ACT_ELU) RC_LAUNCH_SP(RC_ACT_ELU);
    else if (activation == RC_ACT_RELU) RC_LAUNCH_SP(RC_ACT_RELU);
    else RC_LAUNCH_SP(RC_ACT_NONE);
#undef RC_LAUNCH_SP
    return launch_status();
}

extern "C" int rc_first_layer_gather_f16(const int8_t *soa, size_t n, size_t stride, const float *w_rows, const float *bias,
                                         uint16_t *out_hi_lo, size_t H, int activation, float alpha, int32_t *range_flag,
                                         rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_CHECK_SOA(soa, n, stride);
    RC_REQUIRE(w_rows && bias && out_hi_lo, RC_ERR_NULL);
    RC_REQUIRE(aligned16(w_rows) && aligned16(bias) && aligned16(out_hi_lo), RC_ERR_ALIGN);
    RC_REQUIRE(H >= (size_t)kGaCols && H % kGaCols == 0 && H < (1u << 24) && activation >= RC_ACT_NONE && activation <= RC_ACT_ELU, RC_ERR_RANGE);
    RC_REQUIRE(stride * kPlanes < (1ull << 32), RC_ERR_RANGE);   // the planes as ONE buffer resource: 32-bit offsets (214 M states per launch)
    const u32 col_tiles = (u32)(H / kGaCols);
    const size_t lds_bytes = (size_t)kGaRows * kGaCols * 4;
    hipStream_t s = (hipStream_t)stream;
    // small batches: sixteen waves, four columns per lane (units of 4 states); large ones: eight waves, eight columns per lane.
    // A state's outputs are the same bits either way (the same additions in the same order).
    const bool wide = n * col_tiles > (size_t)5000 * 64;
    const u32 waves = wide ? 8 : 16, states = wide ? 8 : 4;
    u32 row_groups = (256 + col_tiles - 1) / col_tiles;   // ~one workgroup per CU (the LDS slice allows no more)
    const u32 rows_per_pass = waves * states;
    const size_t per_group = ceil_div(n, (size_t)row_groups);   // (small batches: units of one wave's states, more workgroups)
    const u32 rows_per_block = (u32)round_up(per_group, per_group < rows_per_pass ? (size_t)states : (size_t)rows_per_pass);
    row_groups = (u32)ceil_div(n, (size_t)rows_per_block);
    const dim3 grid(col_tiles * row_groups);
#define RC_LAUNCH_GA(ACT, WAVES, C)                                                                                   \
    do {                                                                                                              \
        static std::atomic<unsigned long long> attr_set{0};                                                           \
        int dev_ = 0;                                                                                                 \
        (void)hipGetDevice(&dev_);                                                                                    \
        if (!((attr_set.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ull)) {                                     \
            hipError_t e = hipFuncSetAttribute((const void *)k_first_layer_gather<ACT, WAVES, C>,                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);           \
            if (e != hipSuccess) return hip_rc(e);                                                                    \
            attr_set.fetch_or(1ull << (dev_ & 63), std::memory_order_release);                                        \
        }                                                                                                             \
        hipLaunchKernelGGL((k_first_layer_gather<ACT, WAVES, C>), grid, dim3(WAVES * kWave), lds_bytes, s, (const u8 *)soa, n, stride, \
                           (const float4 *)w_rows, bias, (unsigned char *)out_hi_lo, (u32)H, rows_per_block, alpha, (int *)range_flag); \
    } while (0)
#define RC_LAUNCH_GA_ACT(ACT)            \
    do {                                 \
        if (wide) RC_LAUNCH_GA(ACT, 8, 8); \
        else RC_LAUNCH_GA(ACT, 16, 4);   \
    } while (0)
    if (activation == RC_ACT_ELU) RC_LAUNCH_GA_ACT(RC_ACT_ELU);
    else if (activation == RC_ACT_RELU) RC_LAUNCH_GA_ACT(RC_ACT_RELU);
    else RC_LAUNCH_GA_ACT(RC_ACT_NONE);
#undef RC_LAUNCH_GA_ACT
#undef RC_LAUNCH_GA
    return launch_status();
}

extern "C" int rc_adi_targets(const float *values, const uint8_t *child_solved, const uint8_t *state_solved, size_t n,
                              size_t depth, float win_reward, int fix_mode, int64_t *policy_target, float *value_target,
                              rc_stream_t stream) {
    if (n == 0) return RC_OK;
    RC_REQUIRE(values && child_solved && policy_target && value_target, RC_ERR_NULL);
    RC_REQUIRE(fix_mode >= 0 && fix_mode <= 2 && depth > 0, RC_ERR_RANGE);
    RC_REQUIRE(fix_mode != 1 || state_solved != nullptr, RC_ERR_NULL);
    hipLaunchKernelGGL(k_adi_targets, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, values, child_solved,
                       state_solved, n, depth, win_reward, fix_mode, (long long *)policy_target, value_target);
    return launch_status();
}
