// Hidden layers of the fp32-accurate split network (librubiks/model.py::SplitF32Net) as ONE kernel per layer:
//   y = act(hi_x W_hi^T + 2^-11 (hi_x W_lo^T + lo_x W_hi^T) + b),  written as the [hi(y) | lo(y)] halves the next layer reads
// (or as fp32 for the layer in front of the output layer).  It replaces two library GEMMs (K and 2K deep, fp32 out) and
// rc_split_act_f16: the three f16 products run through one fp32 accumulator -- the two correction products first, scaled by
// 2^-11 (exact), the main product on top -- so the fp32 partial matrices never reach HBM.
//
// Shape of the problem (BASELINE config #2: 11 264 = 32 x 352 child rows per step, 4096 -> 2048 -> 1024): the tile is
// 352 x 256 (8 waves, 2 x 4, 176 x 64 outputs per wave), so the 4096 -> 2048 layer is exactly 256 workgroups = one per CU
// with no tail.  The 2048 -> 1024 layer has only 128 such tiles: each is given to two workgroups, one per half of the K loop
// (kPartials, below).  352 x 128 and 176 x 128 tiles exist for other shapes; at 256 workgroups their K-step is shorter than
// the operand latency, so they are the slower choice.  Every tile walks K in the same order: a row's
// result does not depend on the tile or on the other rows of the launch.
//
// Schedule (k_split_gemm): one barrier per K-step; per 16-row fragment of activations two ds_read_b128 and eight MFMAs, and
// behind the MFMAs of the first fragments the wave issues its LDS-DMA pieces of the NEXT stage two at a time -- in the shadow
// of the matrix cores and of the SIMD's other wave, early enough to land before the next barrier.
//
// Data movement: both operands are K-contiguous ([rows][K] halves), staged global -> LDS by global_load_lds_dwordx4 in
// 64-deep K-steps (128-byte LDS rows, two stages).  LDS-DMA writes lane-linear, so the bank swizzle is applied to the
// SOURCE chunk: 16-byte chunk c of row r is stored at slot c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads
// of v_mfma_f32_16x16x32_f16 (16 rows x 2 chunks per 16-lane group) conflict-free.  The weights are the MFMA's A operand
// and the activations its B operand, so a lane's four accumulator registers are four ADJACENT output columns of one row:
// the epilogue stores 8 bytes of hi halves and 8 bytes of lo halves per fragment.
#include <atomic>
#include <type_traits>

#include "rubiks_common.h"
#include "rubiks_netmath.h"

namespace rubiks {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int kOutHalves = 0, kOutF32 = 1, kBf16 = 2, kPartials = 3;   // operand / output kinds of the layer kernels
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Diagnostic builds only (tools/build_ab_lib.sh WORK <name> -DRUBIKS_GEMM_ABLATE=n; WRONG RESULTS, timing experiments):
//   1 no LDS-DMA after the first stage (what memory costs)   2 no barrier / vmcnt wait after the first step (what synchronisation costs)
//   8 no epilogue (activation, split, stores)   16 the epilogue without its stores
//   3 both       (the builds with the fragment reads removed as well, -DRUBIKS_GEMM_ABLATE=4 / 5 / 7 of profiles/r6_gemm_ablation.txt: commit 0190886)
#ifndef RUBIKS_GEMM_ABLATE
#define RUBIKS_GEMM_ABLATE 0
#endif


constexpr int kGemmRowBytes = 128;   // one K-step: 64 halves per row
constexpr int kPiecesPerRow = 2;   // LDS-DMA pieces a wave issues behind the MFMAs of one 16-row fragment

struct GemmArgs {
    const unsigned char *a;      // activations [M][2K] halves: hi | lo                       (kBf16: [M][K] bf16)
    const unsigned char *w;      // weights [N][3K] halves: lo | hi | hi, as the K loop walks them  (kBf16: [N][K] bf16)
    const float *bias;
    void *out;                   // [M][2N] halves (hi | lo), [M][N] fp32, [M][N] bf16, or the partials [S][M][N] fp32
    u32 M, N, K;
    float alpha;
    // optional epilogue inputs (NULL = absent):  y = post_scale * act(acc + bias + residual) + post_shift
    const unsigned char *res;    // residual [M][2N] halves hi | lo (kBf16: [M][N] bf16): the skip connection of a NonConvResBlock (model.py:221-247)
    const float *post_scale, *post_shift;   // [N]: an eval-mode BatchNorm BEHIND the activation that cannot be folded into the next
                                            // Linear because a skip connection reads its output as well
    int *flag;                   // kOutHalves: OR-ed with 1 when an output leaves IEEE half's range (|y| > 65504 or not finite)
    u32 S;                       // kPartials: number of K chunks (workgroups per tile)
    u32 P;                       // products of the f16 kinds: 3 = the split layer (a hi | lo, w lo | hi | hi), 1 = a [M][K] x w [N][K] as they are
};
constexpr float kHalfMax = 65504.0f;

// The rows of a tile's two operands as buffer resources (wave-uniform, in scalar registers): an LDS-DMA piece is then one
// buffer_load ... lds with a loop-invariant VGPR offset and a scalar K offset.
struct StageBufs {
    __amdgpu_buffer_rsrc_t a, w;
};
__device__ __forceinline__ StageBufs stage_bufs(const void *a, const void *w) {
    StageBufs b;
    b.a = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, 0x7FFFFFFF, 0x00020000);   // raw buffer, offsets checked against 2 GB
    b.w = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 0x7FFFFFFF, 0x00020000);
    return b;
}

// (a plain function: used directly inside the kernel template, the builtin silently drops the template's host-side instantiation)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, lds_void *dst, u32 voffset, u32 soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voffset, soffset, 0, 0);
}

template <int WM, int WN, int MR, int NR> struct GemmTile {
    static constexpr int BM = WM * MR * 16, BN = WN * NR * 16, WAVES = WM * WN, THREADS = WAVES * 64;
    static constexpr int A_BYTES = BM * kGemmRowBytes, B_BYTES = BN * kGemmRowBytes, STAGE = A_BYTES + B_BYTES;
    static constexpr int PIECES = STAGE / 1024;                  // wave-level LDS-DMA instructions per stage (8 rows each)
    static constexpr int PPW = (PIECES + WAVES - 1) / WAVES;     // per wave
    static constexpr int LDS_BYTES = 2 * STAGE;
};

// KIND: kOutHalves / kOutF32 = the split layer (f16 operands, three products), kBf16 = a plain bf16 layer (one product, bf16 out),
// kPartials = the split layer with its K loop cut in two: twice the workgroups, each walks one half of the 3K / 64 K-steps and
// stores its raw fp32 accumulators into out[half][M][N] (no bias, no activation).  Half 0 holds correction products only, half
// 1 the rest of them (scaled by 2^-11 inside, as always) plus the main product, so y = act(out[1] + 2^-11 out[0] + bias) --
// exactly what rc_head_split_f32 / rc_split_act_f16 compute from (c, c_corr).  For layers too narrow to fill the chip with
// 352 x 256 tiles (the 2048 -> 1024 layer: 128 tiles).
template <int WM, int WN, int MR, int NR, int ACT, int KIND>
__global__ __launch_bounds__(WM *WN * 64) void k_split_gemm(GemmArgs g) {
    using T = GemmTile<WM, WN, MR, NR>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 wr = wave / WN, wc = wave % WN;

    // workgroups that share an XCD (blockIdx % 8) take consecutive tiles: column tiles fastest, so an XCD's L2 sees few
    // distinct activation rows and every weight column once
    const u32 nwg = gridDim.x, nn = g.N / T::BN;
    const u32 xcd = blockIdx.x % 8, q = nwg / 8, r8 = nwg % 8;
    const u32 wg_all = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + blockIdx.x / 8;
    const u32 n_tiles = KIND == kPartials ? nwg / g.S : nwg;
    const u32 half = wg_all / n_tiles, wg = wg_all % n_tiles;   // kPartials: K chunk `half` of S; the first n_tiles workgroups of the (XCD-ordered) grid walk chunk 0
    const u32 tm = wg / nn, tn = wg % nn;
    const size_t row0 = (size_t)tm * T::BM;
    const u32 col0 = tn * T::BN;
    const bool one = KIND == kBf16 || g.P == 1;   // one product: a is [M][K], w is [N][K]
    const u32 K = g.K, lda = (one ? 1 : 2) * K * 2, ldw = (one ? 1 : 3) * K * 2;   // bytes
    const u32 last_row = (u32)(g.M - 1 - row0);            // rows past M re-read the last row; their outputs are not stored

    // A wave's LDS-DMA pieces (8 rows x 128 bytes each): its activation pieces first (PA of them: piece p = i WAVES + wave of the
    // tile's BM / 8), then its weight pieces.  Each is ONE buffer_load ... lds: the tile's rows as a buffer resource (scalar
    // registers), the lane's place in it a loop-invariant VGPR, the K-step a scalar offset -- no vector address arithmetic per piece
    // (a 64-bit vector add in front of every piece held the wave's issue next to the MFMAs: profiles/r6_gemm_ablation.txt).
    constexpr int DW = T::WAVES;   // waves that issue LDS-DMA: all of them (one wave per SIMD issuing everything: 7.9 -> 9.2 ms, profiles/r6_gemm_ablation.txt)
    constexpr int PA = (T::BM / 8 + DW - 1) / DW, PW = (T::BN / 8 + DW - 1) / DW;
    u32 src_off[PA + PW];
#pragma unroll
    for (int i = 0; i < PA + PW; ++i) {
        const u32 p = (i < PA ? i : i - PA) * DW + wave, R = p * 8 + (lane >> 3), chunk = (lane & 7) ^ ((R >> 1) & 7);
        src_off[i] = i < PA ? min(R, last_row) * lda + chunk * 16 : R * ldw + chunk * 16;   // (the swizzle depends on R mod 16 only: BM is a multiple of 16)
    }
    const StageBufs sb = stage_bufs(g.a + row0 * lda, g.w + (size_t)col0 * ldw);

    auto stage = [&](u32 ks, u32 buf, int lo = 0, int hi = 64) {   // this wave's pieces lo .. hi - 1 of stage ks
        const u32 kk = ks * 64, a_col = (one || kk < 2 * K) ? kk : kk - 2 * K;
        unsigned char *dst = lds + buf * T::STAGE;
#pragma unroll
        for (int i = 0; i < PA + PW; ++i) {
            if (i < lo || i >= hi) continue;
            const u32 p = (i < PA ? i : i - PA) * DW + wave;
            if (wave >= (u32)DW) continue;
            if (i < PA) {
                if (p < (u32)(T::BM / 8)) lds_dma16(sb.a, (lds_void *)(dst + p * 1024), src_off[i], a_col * 2);
            } else if (p < (u32)(T::BN / 8)) {
                lds_dma16(sb.w, (lds_void *)(dst + T::A_BYTES + p * 1024), src_off[i], kk * 2);
            }
        }
    };

    // fragment addresses inside a stage: row (lane & 15) of a 16-row block, chunk ((lane >> 4) + 4 kh) ^ swizzle(row)
    const u32 fr = lane & 15, fq = lane >> 4, swz = fr >> 1;
    u32 x_off[2], w_off[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const u32 c = ((fq + 4 * kh) ^ swz) * 16;
        x_off[kh] = (wr * MR * 16 + fr) * kGemmRowBytes + c;
        w_off[kh] = T::A_BYTES + (wc * NR * 16 + fr) * kGemmRowBytes + c;
    }

    f32x4 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const u32 nk_all = (one ? 1 : 3) * K / 64, scale_step = one ? 0xFFFFFFFFu : 2 * K / 64;
    const u32 ks0 = KIND == kPartials ? half * (nk_all / g.S) : 0u, nk = KIND == kPartials ? ks0 + nk_all / g.S : nk_all;
    stage(ks0, 0);
    {
    // two loops over the K-steps with the scaling of the correction products between them (one loop with the scaling behind a
    // test costs the 512-register tiles their register allocation: the accumulators live in AGPRs, the scaling needs them in VGPRs)
    const u32 ks_mid = scale_step > ks0 && scale_step < nk ? scale_step : nk;
    auto k_step = [&](u32 ks) {
        if (!(RUBIKS_GEMM_ABLATE & 2) || ks == ks0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const bool more = (RUBIKS_GEMM_ABLATE & 1) ? false : ks + 1 < nk;
        const unsigned char *s = lds + ((ks - ks0) & 1) * T::STAGE;

        f16x8 wf[NR][2];
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) wf[n][kh] = *reinterpret_cast<const f16x8 *>(s + w_off[kh] + n * 16 * kGemmRowBytes);
        // the activation fragments of row m + 1 are requested BEFORE the MFMAs of row m (two registers sets, alternating): a wave's
        // matrix instructions then wait for LDS once per K-step, not once per row
        f16x8 xf[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) xf[0][kh] = *reinterpret_cast<const f16x8 *>(s + x_off[kh]);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m + 1 < MR) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) xf[(m + 1) & 1][kh] = *reinterpret_cast<const f16x8 *>(s + x_off[kh] + (m + 1) * 16 * kGemmRowBytes);
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int n = 0; n < NR; ++n)
                    acc[m][n] = KIND == kBf16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[n][kh]), __builtin_bit_cast(bf16x8, xf[m & 1][kh]), acc[m][n], 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[n][kh], xf[m & 1][kh], acc[m][n], 0, 0, 0);
            // the next stage, two LDS-DMA pieces behind each of the first rows' MFMAs: issued in the shadow of the matrix cores
            // (and of the SIMD's other wave) instead of all at once behind the barrier, early enough to land before the next one
            if (more && kPiecesPerRow * m < PA + PW) stage(ks + 1, (ks + 1 - ks0) & 1, kPiecesPerRow * m, kPiecesPerRow * (m + 1));
        }
    };
    for (u32 ks = ks0; ks < ks_mid; ++ks) k_step(ks);
    if (ks_mid < nk) {   // (scaling in front of a chunk's first step would scale zeros: left out)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) acc[m][n] *= (1.0f / kSplitScale);
    }
    for (u32 ks = ks_mid; ks < nk; ++ks) k_step(ks);
    }

    // epilogue: lane holds columns cbase + 16 n + 4 fq + {0..3} of row rbase + 16 m + fr
    const u32 cbase = col0 + wc * NR * 16 + 4 * fq;
    float4 b4[NR];
    bool out_of_range = false;
#pragma unroll
    for (int n = 0; n < NR; ++n) b4[n] = KIND == kPartials ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(g.bias + cbase + 16 * n);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const size_t row = row0 + wr * MR * 16 + m * 16 + fr;
        if (row >= g.M) continue;
        if ((RUBIKS_GEMM_ABLATE & 8) && acc[m][0][0] != 1.2345e-30f) continue;    // (8: no epilogue at all -- what the epilogue costs)

#pragma unroll
        for (int n = 0; n < NR; ++n) {
            float y[4] = {acc[m][n][0] + b4[n].x, acc[m][n][1] + b4[n].y, acc[m][n][2] + b4[n].z, acc[m][n][3] + b4[n].w};
            const u32 col = cbase + 16 * n;
            if (KIND != kPartials && g.res) {   // skip connection: the block's input, in the activations' own format
                if (KIND == kBf16) {
                    const uint2 r2 = *reinterpret_cast<const uint2 *>(g.res + (row * (size_t)g.N + col) * 2);
                    y[0] += __uint_as_float(r2.x << 16), y[1] += __uint_as_float(r2.x & 0xffff0000u);
                    y[2] += __uint_as_float(r2.y << 16), y[3] += __uint_as_float(r2.y & 0xffff0000u);
                } else {
                    const unsigned char *rrow = g.res + row * ((size_t)g.N * 4);
                    const uint2 rh = *reinterpret_cast<const uint2 *>(rrow + col * 2), rl = *reinterpret_cast<const uint2 *>(rrow + ((size_t)g.N + col) * 2);
                    const u32 hw[2] = {rh.x, rh.y}, lw[2] = {rl.x, rl.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[e >> 1] >> (16 * (e & 1))));
                        const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(lw[e >> 1] >> (16 * (e & 1))));
                        y[e] += hi + lo * (1.0f / kSplitScale);
                    }
                }
            }
            {   // branch-free, two values per packed instruction (rubiks_netmath.h)
                const netmath_f32x2 y01 = act_value2<ACT>(netmath_f32x2{y[0], y[1]}, g.alpha), y23 = act_value2<ACT>(netmath_f32x2{y[2], y[3]}, g.alpha);
                y[0] = y01.x, y[1] = y01.y, y[2] = y23.x, y[3] = y23.y;
            }
            if (KIND != kPartials && g.post_scale) {
                const float4 ps = *reinterpret_cast<const float4 *>(g.post_scale + col), pt = *reinterpret_cast<const float4 *>(g.post_shift + col);
                y[0] = y[0] * ps.x + pt.x, y[1] = y[1] * ps.y + pt.y, y[2] = y[2] * ps.z + pt.z, y[3] = y[3] * ps.w + pt.w;
            }
            if (KIND == kPartials) {
                float *orow = reinterpret_cast<float *>(g.out) + ((size_t)half * g.M + row) * (size_t)g.N;
                *reinterpret_cast<float4 *>(orow + col) = make_float4(y[0], y[1], y[2], y[3]);
            } else if (KIND == kBf16) {
                unsigned char *orow = reinterpret_cast<unsigned char *>(g.out) + row * ((size_t)g.N * 2);
                *reinterpret_cast<uint2 *>(orow + col * 2) = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
            } else if (KIND == kOutHalves) {
#pragma unroll
                for (int e = 0; e < 4; ++e) out_of_range |= !(fabsf(y[e]) <= kHalfMax);   // hi would be +-inf (or y is NaN): the caller falls back to fp32
                const SplitPair p01 = split_pair(y[0], y[1]), p23 = split_pair(y[2], y[3]);   // hi = half(y), lo = half((y - hi) 2^11): 5 operations per pair
                unsigned char *orow = reinterpret_cast<unsigned char *>(g.out) + row * ((size_t)g.N * 4);
                if ((RUBIKS_GEMM_ABLATE & 16) && p01.hi != 0x12345678u) continue;
                // ONE 16-byte store per lane instead of two of 8 (the epilogue's stores are bound by their issue: 15 us of the launch at
                // 11 264 rows, profiles/r6_gemm_epilogue.txt): the lanes of fragment column groups 4 fq and 4 (fq + 1) -- 16 lanes apart,
                // the same row -- trade halves (v_permlane16_swap: the odd 16-lane rows of the first operand against the even rows of the
                // second), the even group then stores the hi halves of both, the odd group the lo halves of both.
                const auto s0 = __builtin_amdgcn_permlane16_swap(p01.hi, p01.lo, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(p23.hi, p23.lo, false, false);
                const size_t at = (fq & 1u) ? (size_t)g.N + col - 4 : (size_t)col;
                *reinterpret_cast<uint4 *>(orow + at * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            } else {
                float *orow = reinterpret_cast<float *>(g.out) + row * (size_t)g.N;
                *reinterpret_cast<float4 *>(orow + col) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
    if (KIND == kOutHalves && g.flag && out_of_range) atomicOr(g.flag, 1);
}

template <int WM, int WN, int MR, int NR, int ACT, int KIND> static int launch_split_gemm(const GemmArgs &g, hipStream_t s) {
    using T = GemmTile<WM, WN, MR, NR>;
    static std::atomic<unsigned long long> attr_set{0};   // per device: the attribute belongs to the function ON A DEVICE
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_set.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_split_gemm<WM, WN, MR, NR, ACT, KIND>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return hip_rc(e);
        attr_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const u32 grid = (u32)(ceil_div((size_t)g.M, (size_t)T::BM) * (g.N / T::BN)) * (KIND == kPartials ? g.S : 1u);
    hipLaunchKernelGGL((k_split_gemm<WM, WN, MR, NR, ACT, KIND>), dim3(grid), dim3(T::THREADS), T::LDS_BYTES, s, g);
    return launch_status();
}

template <int WM, int WN, int MR, int NR, int KIND> static int dispatch_split_gemm(const GemmArgs &g, int act, hipStream_t s) {
#define RC_GEMM_ACT(ACT) launch_split_gemm<WM, WN, MR, NR, ACT, KIND>(g, s)
    if (act == RC_ACT_ELU) return RC_GEMM_ACT(RC_ACT_ELU);
    if (act == RC_ACT_RELU) return RC_GEMM_ACT(RC_ACT_RELU);
    return RC_GEMM_ACT(RC_ACT_NONE);
#undef RC_GEMM_ACT
}

}  // namespace rubiks

using namespace rubiks;

// Fills the epilogue options every entry point shares; returns RC_OK or the error of a malformed request.
static int layer_args(const rc_split_layer_t *L, GemmArgs &g) {
    RC_REQUIRE(L != nullptr, RC_ERR_NULL);
    RC_REQUIRE(L->a && L->w, RC_ERR_NULL);
    RC_REQUIRE(aligned16(L->a) && aligned16(L->w) && aligned16(L->bias) && aligned16(L->residual) && aligned16(L->post_scale) &&
                   aligned16(L->post_shift) && aligned16(L->out_hi_lo) && aligned16(L->out_f32) && aligned16(L->out_partials) &&
                   aligned16(L->out_bf16), RC_ERR_ALIGN);
    RC_REQUIRE(L->k >= 64 && L->k % 64 == 0 && L->k <= (1u << 16) && L->n_out % 128 == 0 && L->n_rows < (1ull << 31) && L->n_out < (1u << 20) &&
                   L->activation >= RC_ACT_NONE && L->activation <= RC_ACT_ELU, RC_ERR_RANGE);
    RC_REQUIRE((L->post_scale == nullptr) == (L->post_shift == nullptr), RC_ERR_NULL);
    g.a = (const unsigned char *)L->a;
    g.w = (const unsigned char *)L->w;
    g.bias = L->bias;
    g.M = (u32)L->n_rows;
    g.N = (u32)L->n_out;
    g.K = (u32)L->k;
    g.alpha = L->alpha;
    g.res = (const unsigned char *)L->residual;
    g.post_scale = L->post_scale;
    g.post_shift = L->post_shift;
    g.flag = L->range_flag;
    g.S = 1;
    RC_REQUIRE(L->products == 0 || L->products == 1 || L->products == 3, RC_ERR_RANGE);
    g.P = L->products == 1 ? 1u : 3u;
    return RC_OK;
}

extern "C" size_t rc_split_layer_struct_bytes(void) { return sizeof(rc_split_layer_t); }

extern "C" int rc_split_layer_f16(const rc_split_layer_t *L, rc_stream_t stream) {
    GemmArgs g;
    if (int rc = layer_args(L, g)) return rc;
    if (L->n_rows == 0) return RC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int outs = (L->out_hi_lo != nullptr) + (L->out_f32 != nullptr) + (L->out_partials != nullptr);
    RC_REQUIRE(outs == 1 && L->out_bf16 == nullptr, RC_ERR_NULL);
    if (L->out_partials) {   // K cut into k_splits chunks, raw accumulators: no epilogue inputs
        const u32 S = (u32)L->k_splits, nk_all = g.P * g.K / 64;
        RC_REQUIRE(L->k_splits >= 2 && L->k_splits <= 32 && nk_all % S == 0 && nk_all / S >= 2 && (L->tile == 0 || L->tile == 1 || L->tile == 3 || L->tile == 7) &&
                       L->n_out % (L->tile == 3 ? 128 : L->tile == 7 ? 64 : 256) == 0, RC_ERR_RANGE);
        g.out = (void *)L->out_partials;
        g.S = S;
        g.bias = nullptr;
        if (L->tile == 3) return launch_split_gemm<2, 4, 11, 2, RC_ACT_NONE, kPartials>(g, s);   // 352 x 128 tiles: small batches
        if (L->tile == 7) return launch_split_gemm<2, 4, 11, 1, RC_ACT_NONE, kPartials>(g, s);   // 352 x 64: fewer, deeper chunks for one row tile
        return launch_split_gemm<2, 4, 11, 4, RC_ACT_NONE, kPartials>(g, s);
    }
    RC_REQUIRE(L->bias != nullptr && L->k_splits <= 1 && L->tile >= 0 && L->tile <= 6, RC_ERR_RANGE);
    g.out = L->out_hi_lo ? (void *)L->out_hi_lo : (void *)L->out_f32;
    const bool split = L->out_hi_lo != nullptr;
    int tile = L->tile;
    // tile 0: choose -- the 352 x 256 tile when it fills the chip, else 352 x 128, else 176 x 128
    const size_t row_tiles = ceil_div(L->n_rows, (size_t)352);
    if (tile == 0) tile = (L->n_out % 256 == 0 && row_tiles * (L->n_out / 256) >= 192) ? 1 : (row_tiles * (L->n_out / 128) >= 192) ? 3 : 2;
    RC_REQUIRE((tile != 1 && tile < 4) || L->n_out % 256 == 0, RC_ERR_RANGE);
    const int activation = L->activation;
    if (tile == 6)   // 256 x 256 by four waves, 128 x 128 per wave: the accumulators are exactly the 256 AGPRs
        return split ? dispatch_split_gemm<2, 2, 8, 8, kOutHalves>(g, activation, s) : dispatch_split_gemm<2, 2, 8, 8, kOutF32>(g, activation, s);
    // 352 x 256 by FOUR waves, one per SIMD, accumulators in the upper half of the 512-register file: 176 x 128 per wave (tile 4:
    // 152 KB of fragment reads per K-step instead of 240 KB) or 352 x 64 per wave (tile 5).  Same K order: bit-identical outputs.
    if (tile == 4)
        return split ? dispatch_split_gemm<2, 2, 11, 8, kOutHalves>(g, activation, s) : dispatch_split_gemm<2, 2, 11, 8, kOutF32>(g, activation, s);
    if (tile == 5)
        return split ? dispatch_split_gemm<1, 4, 22, 4, kOutHalves>(g, activation, s) : dispatch_split_gemm<1, 4, 22, 4, kOutF32>(g, activation, s);
    if (tile == 1)   // 352 x 256
        return split ? dispatch_split_gemm<2, 4, 11, 4, kOutHalves>(g, activation, s) : dispatch_split_gemm<2, 4, 11, 4, kOutF32>(g, activation, s);
    if (tile == 3)   // 352 x 128
        return split ? dispatch_split_gemm<2, 4, 11, 2, kOutHalves>(g, activation, s) : dispatch_split_gemm<2, 4, 11, 2, kOutF32>(g, activation, s);
    return split ? dispatch_split_gemm<1, 4, 11, 2, kOutHalves>(g, activation, s) : dispatch_split_gemm<1, 4, 11, 2, kOutF32>(g, activation, s);   // 176 x 128
}

// number of leading K chunks of an S-way cut that hold correction products only (their sum still carries the factor 2^11)
extern "C" int rc_split_layer_corr_chunks(size_t k, int k_splits) {
    if (k_splits < 1 || k % 64) return -1;
    const size_t nk_all = 3 * k / 64, scale_step = 2 * k / 64, per = nk_all / (size_t)k_splits;
    int n = 0;
    for (int p = 0; p < k_splits; ++p) n += (p + 1) * per <= scale_step;
    return n;
}

extern "C" int rc_split_gemm_f16(const uint16_t *a_hi_lo, const uint16_t *w_lo_hi_hi, const float *bias, size_t n_rows, size_t n_out,
                                 size_t k, int activation, float alpha, uint16_t *out_hi_lo, float *out_f32, int tile,
                                 rc_stream_t stream) {
    RC_REQUIRE(a_hi_lo && w_lo_hi_hi && bias && ((out_hi_lo != nullptr) != (out_f32 != nullptr)), RC_ERR_NULL);
    rc_split_layer_t L = {};
    L.a = a_hi_lo, L.w = w_lo_hi_hi, L.bias = bias, L.n_rows = n_rows, L.n_out = n_out, L.k = k, L.activation = activation, L.alpha = alpha;
    L.out_hi_lo = out_hi_lo, L.out_f32 = out_f32, L.tile = tile, L.k_splits = 1;
    return rc_split_layer_f16(&L, stream);
}

extern "C" int rc_gemm_layer_bf16(const rc_split_layer_t *L, rc_stream_t stream) {
    GemmArgs g;
    if (int rc = layer_args(L, g)) return rc;
    if (L->n_rows == 0) return RC_OK;
    RC_REQUIRE(L->bias && L->out_bf16 && !L->out_hi_lo && !L->out_f32 && !L->out_partials, RC_ERR_NULL);
    RC_REQUIRE(L->tile == 0 || L->tile == 1 || L->tile == 3 || L->tile == 4, RC_ERR_RANGE);
    g.out = (void *)L->out_bf16;
    int tile = L->tile;
    const size_t row_tiles = ceil_div(L->n_rows, (size_t)352);
    if (tile == 0) tile = (L->n_out % 256 == 0 && row_tiles * (L->n_out / 256) >= 192) ? 1 : 3;
    RC_REQUIRE((tile != 1 && tile != 4) || L->n_out % 256 == 0, RC_ERR_RANGE);
    hipStream_t s = (hipStream_t)stream;
    if (tile == 4) return dispatch_split_gemm<2, 2, 11, 8, kBf16>(g, L->activation, s);
    if (tile == 1) return dispatch_split_gemm<2, 4, 11, 4, kBf16>(g, L->activation, s);
    return dispatch_split_gemm<2, 4, 11, 2, kBf16>(g, L->activation, s);
}

extern "C" int rc_gemm_bias_act_bf16(const uint16_t *a, const uint16_t *w, const float *bias, size_t n_rows, size_t n_out, size_t k,
                                     int activation, float alpha, uint16_t *out, int tile, rc_stream_t stream) {
    RC_REQUIRE(a && w && bias && out, RC_ERR_NULL);
    rc_split_layer_t L = {};
    L.a = a, L.w = w, L.bias = bias, L.n_rows = n_rows, L.n_out = n_out, L.k = k, L.activation = activation, L.alpha = alpha;
    L.out_bf16 = out, L.tile = tile, L.k_splits = 1;
    return rc_gemm_layer_bf16(&L, stream);
}

extern "C" int rc_split_gemm_partials_f16(const uint16_t *a_hi_lo, const uint16_t *w_lo_hi_hi, size_t n_rows, size_t n_out, size_t k,
                                          float *out_partials, rc_stream_t stream) {
    RC_REQUIRE(a_hi_lo && w_lo_hi_hi && out_partials, RC_ERR_NULL);
    RC_REQUIRE(k >= 128 && k % 128 == 0, RC_ERR_RANGE);
    rc_split_layer_t L = {};
    L.a = a_hi_lo, L.w = w_lo_hi_hi, L.n_rows = n_rows, L.n_out = n_out, L.k = k, L.out_partials = out_partials, L.k_splits = 2;
    return rc_split_layer_f16(&L, stream);
}
