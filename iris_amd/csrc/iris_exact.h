// iris_exact.h -- correctly rounded f32 division and square root in fewer instructions.
//
// The film is compared bit for bit with a scalar CPU evaluation, so every `/` and sqrt of the reference must be the IEEE
// result.  hipcc's correctly rounded expansions are general: 11 VALU instructions per division (two v_div_scale, v_rcp,
// six fma/mul, v_div_fmas, v_div_fixup) and 16 per square root, most of it handling of operand ranges that a given call
// site never sees.  The sequences below are the same Newton/Markstein refinements WITHOUT the range handling; each
// states its domain, and tools/exact/verify_exact.hip proves on the GPU, exhaustively, that inside that domain the
// result equals the compiler's `/` / sqrtf for every operand (all 2^32 inputs for the one-operand functions; all
// 2^23 x 2^23 significand pairs for the quotient -- inside the domain nothing over- or underflows, so powers of two
// factor out of every intermediate exactly).  Call sites either prove their operands are inside the domain or test it
// and take the compiler's expansion otherwise (`*_guarded`), so no operand ever gets a different bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iris {
namespace exact {

#define IRIS_EX __device__ __forceinline__

IRIS_EX float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// (bits << 1) - 2: an unsigned key that orders |x| and sends +-0 to the top (0xfffffffe), so that "zero or not tiny"
// is one unsigned comparison against a threshold.
IRIS_EX uint32_t mag_key(float x) { return (__float_as_uint(x) << 1) - 2u; }
#define IRIS_EX_KEY(bits) (((uint32_t)(bits) << 1) - 2u)
#define IRIS_EX_BITS_2M100 0x0d800000u  // 2^-100
#define IRIS_EX_BITS_2M96 0x0f800000u   // 2^-96
#define IRIS_EX_BITS_2M60 0x21800000u   // 2^-60
#define IRIS_EX_BITS_2P60 0x5d800000u   // 2^60
#define IRIS_EX_BITS_2P100 0x71800000u  // 2^100
#define IRIS_EX_BITS_2P126 0x7e800000u  // 2^126

// ---------------------------------------------------------------- square root
// Domain: x == +-0, or 2^-96 <= x <= 2^126 (the residual x - s*s needs 46 bits below x: exact down to 2^-96 or so).
// v_rsq_f32 (1 ulp) -> one coupled Newton step on (s, h = 1/(2s)) -> Markstein's final correction with the exact
// residual.  A NaN comes out as NaN.  Zero: the reciprocal square root is taken of max(x, 2^-126), so s0 = 0 * 2^63 = 0
// and every later term is zero.
IRIS_EX bool sqrt_domain(float x) { return x == 0.0f || (x >= 0x1p-96f && x <= 0x1p126f); }
IRIS_EX float sqrt_fast(float x) {
    const float r = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 0x1p-126f));
    const float s0 = x * r;
    const float h0 = 0.5f * r;
    const float e = fma_(-h0, s0, 0.5f);
    const float s1 = fma_(s0, e, s0);
    const float h1 = fma_(h0, e, h0);
    const float d = fma_(-s1, s1, x);
    return fma_(d, h1, s1);
}
// The same for x known to be > 0 (no zero to protect).
IRIS_EX float sqrt_fast_pos(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    const float s0 = x * r;
    const float h0 = 0.5f * r;
    const float e = fma_(-h0, s0, 0.5f);
    const float s1 = fma_(s0, e, s0);
    const float h1 = fma_(h0, e, h0);
    const float d = fma_(-s1, s1, x);
    return fma_(d, h1, s1);
}

// ---------------------------------------------------------------- division by a positive denominator
// recip_for(d): v_rcp_f32 (1 ulp) + one Newton step.  Domain: 2^-100 <= d <= 2^100.
// div_by(n, d, y): q0 = n*y, then STEPS corrections q += (n - d*q) * y with the exact residual from one fma each.
// Domain: d as above and POSITIVE; n == +-0 or 2^-100 <= |n| <= 2^100; 2^-100 <= |n/d| <= 2^100.  The residual is
// formed as -(d*q - n): with a zero numerator that keeps IEEE's sign of the zero quotient (n - d*q would give +0 - 0).
struct Recip { float y; };
IRIS_EX bool rcp_domain(float d) { return d >= 0x1p-100f && d <= 0x1p100f; }
IRIS_EX float rcp_refined(float d) {
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = fma_(-d, y0, 1.0f);
    return fma_(e, y0, y0);
}
IRIS_EX Recip recip_for(float d) { Recip r = {rcp_refined(d)}; return r; }
#ifndef IRIS_EXACT_DIV_STEPS
#define IRIS_EXACT_DIV_STEPS 1
#endif
IRIS_EX float div_by(float n, float d, Recip r) {
    float q = n * r.y;
#pragma unroll
    for (int k = 0; k < IRIS_EXACT_DIV_STEPS; k++) {
        const float t = fma_(d, q, -n);
        q = fma_(-t, r.y, q);
    }
    return q;
}
IRIS_EX bool div_domain(float n, float d) {
    if (!(d >= 0x1p-60f && d <= 0x1p60f)) return false;
    const float an = __builtin_fabsf(n);
    return n == 0.0f || (an >= 0x1p-60f && an <= 0x1p60f);
}

// ---------------------------------------------------------------- x / PI_F
// Division by the constant 3.14159274101257324219f with c = RN(1/PI_F): q0 = x*c, one correction.
// Domain: x == +-0 or 2^-100 <= |x| <= 2^100.
#define IRIS_EX_PI 3.14159274101257324219f
#define IRIS_EX_INV_PI 0.318309873342514038086f  // RN(1 / PI_F)
IRIS_EX bool div_pi_domain(float x) { const float a = __builtin_fabsf(x); return x == 0.0f || (a >= 0x1p-100f && a <= 0x1p100f); }
IRIS_EX float div_pi_fast(float x) {
    const float q0 = x * IRIS_EX_INV_PI;
    const float t = fma_(IRIS_EX_PI, q0, -x);
    return fma_(-t, IRIS_EX_INV_PI, q0);
}

}  // namespace exact
}  // namespace iris
