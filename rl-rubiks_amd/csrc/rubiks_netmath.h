// Small numeric helpers shared by the network kernels (rubiks_net.hip, rubiks_gemm.hip).
#pragma once
#include "rubiks_common.h"

namespace rubiks {

constexpr float kSplitScale = 2048.0f;        // 2^11: x = hi + lo * 2^-11 (see rubiks_net.hip, "f16x3 split")

__device__ __forceinline__ u32 pack_half2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float float2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2_;
    float2_ v = {lo, hi};
    half2_ r = __builtin_convertvector(v, half2_);   // round to nearest even
    return __builtin_bit_cast(u32, r);
}
__device__ __forceinline__ float round_to_half_f32(float x) { return (float)(_Float16)x; }

__device__ __forceinline__ u32 pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float float2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    float2_ v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);   // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(u32, r);
}

// expm1(x) for x <= 0 to ~1e-7 absolute (what ELU's negative branch needs next to fp32 activations of order 1) without
// libm's register appetite: the Taylor polynomial near zero (x^9 / 9! < 3e-10 at |x| = 0.35), exp(x) - 1 beyond.
__device__ __forceinline__ float expm1_neg(float x) {
    float p = 1.0f / 40320.0f;
    p = fmaf(p, x, 1.0f / 5040.0f);
    p = fmaf(p, x, 1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    return x < -0.35f ? __expf(x) - 1.0f : p * x;
}

}  // namespace rubiks
