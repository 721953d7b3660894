// Shared host/device helpers for the librubiks_hip translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rubiks_hip.h"
#include "rubiks_tables.h"

namespace rubiks {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // default workgroup: 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 workgroups: memory-bound kernels grid-stride beyond this

// Move tables in constant memory; every kernel that needs them copies its layout into LDS.
static __constant__ MoveTables c_tables = kTables;

__host__ __device__ inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }
__host__ __device__ inline size_t ceil_div(size_t x, size_t m) { return (x + m - 1) / m; }
__host__ __device__ inline u32 round_up_dev16(size_t x) { return (u32)((x + 15) & ~(size_t)15); }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int hip_rc(hipError_t e) { return e == hipSuccess ? RC_OK : RC_ERR_HIP_BASE - (int)e; }

#define RC_REQUIRE(cond, code) \
    do {                       \
        if (!(cond)) return (code); \
    } while (0)

#define RC_CHECK_SOA(ptr, n, stride)                                   \
    do {                                                               \
        RC_REQUIRE((ptr) != nullptr, RC_ERR_NULL);                     \
        RC_REQUIRE(aligned16(ptr) && ((stride) & 15u) == 0, RC_ERR_ALIGN); \
        RC_REQUIRE((stride) >= round_up((n), 16), RC_ERR_STRIDE);      \
    } while (0)

inline int launch_status() { return hip_rc(hipGetLastError()); }

inline unsigned grid_for(size_t work_items, int block = kBlock, int cap = kMaxGrid) {
    size_t g = ceil_div(work_items, (size_t)block);
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (unsigned)g;
}

// ---- device helpers ---------------------------------------------------------------------------

// Cooperative copy of `bytes` (multiple of 4) from constant memory to LDS.
__device__ __forceinline__ void stage_to_lds(u32 *dst, const void *src, int bytes) {
    const u32 *s = reinterpret_cast<const u32 *>(src);
    for (int i = threadIdx.x; i < bytes / 4; i += blockDim.x) dst[i] = s[i];
}

// 5-bit code of byte b of a packed dword (state codes are 0..23; masking keeps table reads in range).
__device__ __forceinline__ u32 code_of(u32 packed, int b) { return (packed >> (8 * b)) & 31u; }

// ---- packed node states shared by the search kernels ---------------------------------------------
// A node's state is stored packed: 20 codes of 5 bits, 6 per dword (cubies 0-5, 6-11, 12-17, 18-19).

__device__ __forceinline__ bool key_eq(const uint4 &a, const uint4 &b) {
    return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
}

__device__ __forceinline__ u32 key_hash(const uint4 &k) {
    u32 h = k.x * 0x9E3779B1u;
    h ^= h >> 15;
    h += k.y * 0x85EBCA77u;
    h ^= h >> 13;
    h += k.z * 0xC2B2AE3Du;
    h ^= h >> 16;
    h += k.w * 0x27D4EB2Fu;
    h ^= h >> 15;
    h *= 0x165667B1u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ u32 key_code(const uint4 &k, int j) {   // j compile-time after unrolling
    const u32 w = (j < 6) ? k.x : (j < 12) ? k.y : (j < 18) ? k.z : k.w;
    return (w >> (5 * (j % 6))) & 31u;
}

__device__ __forceinline__ void key_set(u32 (&w)[4], int j, u32 code) { w[j / 6] |= code << (5 * (j % 6)); }

// Makes this wave's earlier global stores visible to its later loads (same CU, so L1/L2 in order
// once the stores have left the wave).
__device__ __forceinline__ void wave_store_fence() { 
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
 }

// ---- reductions over a 16-lane row with DPP (VALU-speed cross-lane moves, no LDS crossbar) --------------
// The four steps (swap neighbours, swap pairs, mirror within 8, mirror within 16) pair every lane with
// partners that together cover the row, so after them EVERY lane of the row holds the row's result.
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror
constexpr int kDppMirror = 0x140;      // row_mirror

template <int CTRL> __device__ __forceinline__ int dpp_int(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ float dpp_float(float v) {
    return __int_as_float(dpp_int<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_double(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_int<CTRL>((int)(b & 0xffffffffll)), hi = dpp_int<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The same reductions with the DPP permutation fused into the arithmetic instruction (hipcc emits a
// v_mov_dpp + the operation + hazard nops per step): one instruction per step after a two-wait-state nop.
// For dependent chains that are paced by instruction count (the sequential walk of k_mcts_select).
#define RUBIKS_DPP_STEP(OP, CTRL, v) \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ int row16_sum_fused(int v) {
    RUBIKS_DPP_STEP("v_add_u32_dpp", "quad_perm:[1,0,3,2]", v);
    RUBIKS_DPP_STEP("v_add_u32_dpp", "quad_perm:[2,3,0,1]", v);
    RUBIKS_DPP_STEP("v_add_u32_dpp", "row_half_mirror", v);
    RUBIKS_DPP_STEP("v_add_u32_dpp", "row_mirror", v);
    return v;
}
__device__ __forceinline__ float row16_max_fused(float v) {
    RUBIKS_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]", v);
    RUBIKS_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]", v);
    RUBIKS_DPP_STEP("v_max_f32_dpp", "row_half_mirror", v);
    RUBIKS_DPP_STEP("v_max_f32_dpp", "row_mirror", v);
    return v;
}

__device__ __forceinline__ int row16_sum(int v) {
    v += dpp_int<kDppXor1>(v);
    v += dpp_int<kDppXor2>(v);
    v += dpp_int<kDppHalfMirror>(v);
    v += dpp_int<kDppMirror>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_float<kDppXor1>(v));
    v = fmaxf(v, dpp_float<kDppXor2>(v));
    v = fmaxf(v, dpp_float<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_float<kDppMirror>(v));
    return v;
}
// argmax with "first maximum wins": the larger score, and on equal scores the smaller index
template <int CTRL> __device__ __forceinline__ void argmax_step(double &score, int &arg) {
    const double os = dpp_double<CTRL>(score);
    const int oa = dpp_int<CTRL>(arg);
    if (os > score || (os == score && oa < arg)) { score = os; arg = oa; }
}
template <int CTRL> __device__ __forceinline__ void argmax_step_f32(float &score, int &arg) {
    const float os = dpp_float<CTRL>(score);
    const int oa = dpp_int<CTRL>(arg);
    if (os > score || (os == score && oa < arg)) { score = os; arg = oa; }
}
__device__ __forceinline__ int row16_argmax_first(double score, int arg) {
    argmax_step<kDppXor1>(score, arg);
    argmax_step<kDppXor2>(score, arg);
    argmax_step<kDppHalfMirror>(score, arg);
    argmax_step<kDppMirror>(score, arg);
    return arg;
}

}  // namespace rubiks
