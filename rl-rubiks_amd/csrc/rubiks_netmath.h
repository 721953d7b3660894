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

// ELU / ReLU / identity on one value, BRANCH-FREE: both sides of expm1_neg's split are evaluated and selected (left to itself the
// compiler wraps every __expf in a divergent branch with its own exec-mask save: 64 branch regions in an epilogue of 64 values).
// Same values as `x > 0 ? x : alpha * expm1_neg(x)`.
template <int ACT> __device__ __forceinline__ float act_value(float x, float alpha) {
    if (ACT == RC_ACT_RELU) return fmaxf(x, 0.f);
    if (ACT != RC_ACT_ELU) return x;
    float e = __expf(x);
    asm volatile("" : "+v"(e));   // keeps the exponential out of a conditional block
    float p = 1.0f / 40320.0f;
    p = fmaf(p, x, 1.0f / 5040.0f);
    p = fmaf(p, x, 1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, 1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    const float em1 = x < -0.35f ? e - 1.0f : p * x;
    return x > 0.f ? x : alpha * em1;
}

// act_value on two values at once: the polynomial as packed fused multiply-adds (v_pk_fma_f32) -- the same operations on each value,
// hence the same results, in half the instructions.
typedef float netmath_f32x2 __attribute__((ext_vector_type(2)));
template <int ACT> __device__ __forceinline__ netmath_f32x2 act_value2(netmath_f32x2 x, float alpha) {
    if (ACT != RC_ACT_ELU) return netmath_f32x2{act_value<ACT>(x.x, alpha), act_value<ACT>(x.y, alpha)};
    float e0 = __expf(x.x), e1 = __expf(x.y);
    asm volatile("" : "+v"(e0), "+v"(e1));   // keeps the exponentials out of conditional blocks
    const auto k = [](float c) { return netmath_f32x2{c, c}; };
    netmath_f32x2 p = k(1.0f / 40320.0f);
    p = __builtin_elementwise_fma(p, x, k(1.0f / 5040.0f));
    p = __builtin_elementwise_fma(p, x, k(1.0f / 720.0f));
    p = __builtin_elementwise_fma(p, x, k(1.0f / 120.0f));
    p = __builtin_elementwise_fma(p, x, k(1.0f / 24.0f));
    p = __builtin_elementwise_fma(p, x, k(1.0f / 6.0f));
    p = __builtin_elementwise_fma(p, x, k(0.5f));
    p = __builtin_elementwise_fma(p, x, k(1.0f));
    const netmath_f32x2 px = p * x;
    const float m0 = x.x < -0.35f ? e0 - 1.0f : px.x, m1 = x.y < -0.35f ? e1 - 1.0f : px.y;
    return netmath_f32x2{x.x > 0.f ? x.x : alpha * m0, x.y > 0.f ? x.y : alpha * m1};
}

// The [hi | lo] split of two adjacent values, x = hi + lo * 2^-11: hi = half(y) (round to nearest even), lo = half((y - hi) * 2^11).
// (y - hi) * 2^11 is formed as fma(hi, -2^11, y * 2^11): both products are exact and their difference is representable, so the single
// rounding returns exactly what the subtract-then-scale form returns; v_fma_mix_f32 reads hi straight from the packed pair (no
// conversion back to fp32).  5 VALU operations per pair instead of 9.
struct SplitPair {
    u32 hi, lo;
};
__device__ __forceinline__ SplitPair split_pair(float y0, float y1) {
    SplitPair s;
    s.hi = pack_half2(y0, y1);
    const float t0 = y0 * kSplitScale, t1 = y1 * kSplitScale, ms = -kSplitScale;
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(s.hi), "v"(ms), "v"(t0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(s.hi), "v"(ms), "v"(t1));
    s.lo = pack_half2(l0, l1);
    return s;
}

}  // namespace rubiks
