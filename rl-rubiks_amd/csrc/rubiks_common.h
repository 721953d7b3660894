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

}  // namespace rubiks
