// iris_device.h -- gfx950 device functions of the iris path-tracing hot path.
//
// Everything here runs per lane on a register-resident path state; the kernels
// in iris_kernels.hip move that state between HBM/L2-resident SoA pools, LDS
// queues and registers.  Compile with -ffp-contract=off: each f32 operation
// below is one IEEE-754 operation in the order the reference writes it
// (file:line citations are relative to the reference's src/), which is what
// makes the film bit-comparable with a scalar CPU evaluation.  The only fused
// operations are the explicit fmaf calls of the sigmoid spectrum
// (spectrum/upsample.rs:19-23) and the f64 fma chain of iris_sin/iris_cos
// (docs/SPEC.md), both fused in the reference/specification as well.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/iris_hip.h"
#include "iris_bvh_build.h"  // IRIS_BVH_STACK
#include "iris_exact.h"

#define IRIS_TABLE_QUAL __device__ __attribute__((aligned(16))) const  // 16 bytes: sobol_sample_4d reads the direction vectors four words at a time
#include "tables/cie_cmf.inc"
#include "tables/sobol_rev16.inc"
#define IRIS_SOBOL_SETS 64u

namespace iris {

#define IRIS_DEV __device__ __forceinline__
#define PI_F 3.14159274101257324219f

// ---------------------------------------------------------------- scalars
// f32::max / f32::min (IEEE maxNum/minNum: a NaN operand is ignored)
IRIS_DEV float rmax(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
IRIS_DEV float rmin(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
// f32::clamp (NaN stays NaN)
IRIS_DEV float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// Rust `as usize` on a value that is then used as a small index, and Rust `as i32`: saturating, NaN -> 0.
// That is exactly what the gfx9 conversion instructions do in hardware (a C cast would be undefined
// out of range, hence the explicit instruction).
IRIS_DEV uint32_t f_as_index(float f) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
IRIS_DEV int32_t f_as_i32(float f) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}
IRIS_DEV float fabs_(float x) { return __builtin_fabsf(x); }

// ---------------------------------------------------------------- correctly rounded sqrt and division (iris_exact.h)
// Every `/` and sqrt below is the IEEE result, as the reference computes it.  The compiler's general expansions cost 11
// and 16 instructions; the sequences of iris_exact.h give the same bits in 3 (+3 per denominator) and 8 for operands
// inside their domains (proved exhaustively on the GPU, tools/exact/verify_exact.hip).  The wrappers here test the domain
// with a couple of integer instructions and fall back to the compiler's expansion for anything outside (lanes that do
// so are rare enough never to be seen in a profile: zero-length vectors, grazing discriminants below 2^-96, NaNs).
#ifndef IRIS_EXACT_FAST
#define IRIS_EXACT_FAST 1   // 0: the compiler's expansions everywhere (the films must not change by a bit: tested)
#endif
// The domain tests are made wave-uniform: the fast sequence runs when EVERY active lane of the wave is inside the domain
// (one scalar branch, no exec-mask bookkeeping); otherwise all of them take the compiler's expansion, which is right for
// any operand.  Per-lane branches here cost more scalar instructions than the fast sequences save (measured).
#define IRIS_ALL_LANES(cond) (__ballot(!(cond)) == 0ull)
// lo <= x <= hi for POSITIVE finite bounds given as bit patterns; false for negative x, NaN and anything outside
IRIS_DEV bool in_range_bits(float x, uint32_t lo_bits, uint32_t hi_bits) { return (__float_as_uint(x) - lo_bits) <= (hi_bits - lo_bits); }
// every component is +-0 or at least 2^-100 in magnitude (the residual of the quotient stays exact).  Says nothing about large
// components: for callers whose numerators are bounded by the denominator (normalize: |a.x| <= |a|).
IRIS_DEV bool numerators_ok(float x, float y, float z) {
    const uint32_t kx = exact::mag_key(x), ky = exact::mag_key(y), kz = exact::mag_key(z);
    const uint32_t k = kx < ky ? (kx < kz ? kx : kz) : (ky < kz ? ky : kz);  // v_min3_u32
    return k >= IRIS_EX_KEY(IRIS_EX_BITS_2M100);
}
// every component is +-0 or between 2^-60 and 2^60 in magnitude: with a denominator in [2^-60, 2^60] the quotient is
// 0 or in [2^-120, 2^120] -- it neither overflows (an infinite or huge numerator would turn the residual
// into inf - inf) nor leaves the normal range
IRIS_DEV bool numerators_moderate(float x, float y, float z) {
    // integer tests on the magnitude bits, so that a signalling NaN beside an infinity cannot hide it (a float max would return
    // the quieted NaN, which compares false with everything): any NaN or infinity fails, and the compiler's expansion runs
    const uint32_t mx = __float_as_uint(x) << 1, my = __float_as_uint(y) << 1, mz = __float_as_uint(z) << 1;
    const uint32_t kx = mx - 2u, ky = my - 2u, kz = mz - 2u;  // exact::mag_key
    const uint32_t lo = kx < ky ? (kx < kz ? kx : kz) : (ky < kz ? ky : kz);  // v_min3_u32; +-0 is at the top
    const uint32_t hi = mx > my ? (mx > mz ? mx : mz) : (my > mz ? my : mz);  // v_max3_u32; +-0 is at the bottom
    return lo >= IRIS_EX_KEY(IRIS_EX_BITS_2M60) && hi <= (IRIS_EX_BITS_2P60 << 1);
}
// Opt-in relaxed arithmetic (a separate build of the library, libiris_hip_relaxed.so: python -m iris_amd.build): the RADIOMETRIC
// code -- light sampling, BSDFs, MIS weights, spectra, the environment map, i.e. everything between a hit and the next ray --
// uses v_rcp / v_sqrt / v_rsq results (1 ulp) instead of correctly rounded quotients and roots, and lets the compiler contract
// a * b + c.  The camera sample (pixel, wavelength, first ray), the sampler and every sphere and box test keep the exact
// arithmetic, so a given ray still finds the same hit at the same distance, bit for bit.  The film is then no longer the
// reference's bit for bit; it is held to north_star's statistical gate instead (tests/test_gpu_relaxed.py).
#ifndef IRIS_RELAXED
#define IRIS_RELAXED 0
#endif
#if IRIS_RELAXED
#define IRIS_RADIOMETRIC_BEGIN _Pragma("clang fp contract(fast)")
#define IRIS_RADIOMETRIC_END _Pragma("clang fp contract(off)")
#else
#define IRIS_RADIOMETRIC_BEGIN
#define IRIS_RADIOMETRIC_END
#endif
// sqrt for any argument, always the IEEE result (sphere tests, camera)
IRIS_DEV float sqrt_x(float x) {
#if IRIS_EXACT_FAST
    if (__builtin_expect(IRIS_ALL_LANES(in_range_bits(x, IRIS_EX_BITS_2M96, IRIS_EX_BITS_2P126)), 1)) return exact::sqrt_fast_pos(x);
#endif
    return __builtin_sqrtf(x);
}
#if IRIS_RELAXED
IRIS_DEV float sqrt_g(float x) { return __builtin_amdgcn_sqrtf(x); }
IRIS_DEV float sqrt_nz(float x) { return __builtin_amdgcn_sqrtf(x); }
IRIS_DEV float rcp_g(float d) { return __builtin_amdgcn_rcpf(d); }
IRIS_DEV float div_pi(float x) { return x * IRIS_EX_INV_PI; }
IRIS_DEV float fdiv(float n, float d) { return n * __builtin_amdgcn_rcpf(d); }
#else
// sqrt for any argument (radiometric code: relaxed in the relaxed build)
IRIS_DEV float sqrt_g(float x) { return sqrt_x(x); }
// sqrt for an argument that is +-0, NaN or in [2^-96, 2^126] BY CONSTRUCTION -- the call site says why.  The usual case:
// max(1 - y, 0) with y >= 0 is 0 or at least 2^-24, and a difference of two floats the larger of which is at least 2^-24
// is 0 or at least 2^-49.
IRIS_DEV float sqrt_nz(float x) {
#if IRIS_EXACT_FAST
    return exact::sqrt_fast(x);
#else
    return __builtin_sqrtf(x);
#endif
}
// 1 / d
IRIS_DEV float rcp_g(float d) {
#if IRIS_EXACT_FAST
    if (__builtin_expect(IRIS_ALL_LANES(in_range_bits(d, IRIS_EX_BITS_2M100, IRIS_EX_BITS_2P100)), 1)) return exact::rcp_refined(d);
#endif
    return 1.0f / d;
}
// x / PI_F
IRIS_DEV float div_pi(float x) {
#if IRIS_EXACT_FAST
    if (__builtin_expect(IRIS_ALL_LANES(exact::mag_key(x) - IRIS_EX_KEY(IRIS_EX_BITS_2M100) <= IRIS_EX_KEY(IRIS_EX_BITS_2P100) - IRIS_EX_KEY(IRIS_EX_BITS_2M100) || x == 0.0f), 1))
        return exact::div_pi_fast(x);
#endif
    return x / PI_F;
}
// n / d in the radiometric code (one correctly rounded division; the relaxed build multiplies by v_rcp)
IRIS_DEV float fdiv(float n, float d) { return n / d; }
#endif
IRIS_DEV float rsqrt_exact(float x) { return sqrt_g(x); }

// ---------------------------------------------------------------- sin / cos (docs/SPEC.md)
// f64 Cody-Waite reduction by pi/2 + fdlibm kernel polynomials, explicit fma chain,
// one rounding to f32.  Within 1 ulp of (and almost always equal to) a correctly
// rounded libm sinf/cosf, which is what the reference's f32::sin/cos call.
IRIS_DEV double trig_reduce(double xd, int &quadrant) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    double kd = __builtin_rint(xd * TWO_OVER_PI);
    double r = __builtin_fma(-kd, PIO2_HI, xd);
    r = __builtin_fma(-kd, PIO2_LO, r);
    quadrant = ((int)kd) & 3;
    return r;
}
// The polynomial coefficients are pinned to scalar registers where they are used: left alone, the compiler hoists all
// twelve binary64 constants into vector registers at the top of the kernel (24 VGPRs for the whole kernel), and in the
// 168-register kernels spills them and reloads them from scratch around every evaluation.
// pin (a compile-time constant at every call): false for the Lambertian list kernel, which has registers to spare (119 of
// 168 when pinned, no spills either way) and is 1 % faster with the constants left in vector registers; true elsewhere
// (general classes: 44 / 36 B of scratch per lane -> none; pooled hierarchy kernel: 108 -> 28 B, 142 -> 127 ms).
IRIS_DEV double trig_const(double c, bool pin) {
    if (pin) asm("" : "+s"(c));
    return c;
}
IRIS_DEV double trig_ksin(double r, bool pin) {
    const double S1 = trig_const(-1.66666666666666324348e-01, pin), S2 = trig_const(8.33333333332248946124e-03, pin),
                 S3 = trig_const(-1.98412698298579493134e-04, pin), S4 = trig_const(2.75573137070700676789e-06, pin),
                 S5 = trig_const(-2.50507602534068634195e-08, pin), S6 = trig_const(1.58969099521155010221e-10, pin);
    double z = r * r;
    double p = __builtin_fma(z, S6, S5);
    p = __builtin_fma(z, p, S4);
    p = __builtin_fma(z, p, S3);
    p = __builtin_fma(z, p, S2);
    p = __builtin_fma(z, p, S1);
    return __builtin_fma(r * z, p, r);
}
IRIS_DEV double trig_kcos(double r, bool pin) {
    const double C1 = trig_const(4.16666666666666019037e-02, pin), C2 = trig_const(-1.38888888888741095749e-03, pin),
                 C3 = trig_const(2.48015872894767294178e-05, pin), C4 = trig_const(-2.75573143513906633035e-07, pin),
                 C5 = trig_const(2.08757232129817482790e-09, pin), C6 = trig_const(-1.13596475577881948265e-11, pin);
    double z = r * r;
    double p = __builtin_fma(z, C6, C5);
    p = __builtin_fma(z, p, C4);
    p = __builtin_fma(z, p, C3);
    p = __builtin_fma(z, p, C2);
    p = __builtin_fma(z, p, C1);
    return __builtin_fma(z * z, p, __builtin_fma(z, -0.5, 1.0));
}
// sin and cos of the same argument share the reduction and both kernels.
IRIS_DEV void iris_sincos(float x, float &s_out, float &c_out, bool pin = true) {
#if IRIS_RELAXED || defined(IRIS_EXPERIMENT_HW_SINCOS)  // (the second: a timing experiment only -- what ALL of the binary64 evaluation costs)
    // relaxed build: the hardware's sine and cosine (arguments in revolutions; absolute error about 1e-6) instead of ~45 binary64
    // instructions.  Every caller passes an angle in [-pi, 2 pi] and normalises the direction it builds from the result.
    (void)pin;
    const float rev = x * 0.15915494309189535f;
    s_out = __builtin_amdgcn_sinf(rev);
    c_out = __builtin_amdgcn_cosf(rev);
    return;
#endif
    double xd = (double)x;
    if (!(__builtin_fabs(xd) < 1.0e5)) { s_out = x - x; c_out = x - x; return; }
    int q;
    double r = trig_reduce(xd, q);
    double ks = trig_ksin(r, pin), kc = trig_kcos(r, pin);
    double s = (q & 1) ? kc : ks;
    double c = (q & 1) ? ks : kc;
    s_out = (float)((q & 2) ? -s : s);
    c_out = (float)(((q + 1) & 2) ? -c : c);
}

// ---------------------------------------------------------------- vectors (math/vec3.rs, point3.rs, vec4.rs)
struct V3 { float x, y, z; };
struct F4 { float v[4]; };  // Vec4 / SpectralSample / PdfSet / Wavelength; always indexed with constants

IRIS_DEV V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
IRIS_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
IRIS_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
IRIS_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
IRIS_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
IRIS_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
IRIS_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
IRIS_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:45-48
IRIS_DEV V3 cross(V3 a, V3 b) {                                                   // vec3.rs:50-57
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
IRIS_DEV float len_squared(V3 a) { return dot(a, a); }
IRIS_DEV float len(V3 a) { return sqrt_g(len_squared(a)); }
// a / d with one reciprocal for the three quotients; any d, any numerators (outside the proven domain: three divisions)
IRIS_DEV V3 div3_x(V3 a, float d) {
#if IRIS_EXACT_FAST
    if (__builtin_expect(IRIS_ALL_LANES(in_range_bits(d, IRIS_EX_BITS_2M60, IRIS_EX_BITS_2P60) && numerators_moderate(a.x, a.y, a.z)), 1)) {
        const exact::Recip y = exact::recip_for(d);
        return v3(exact::div_by(a.x, d, y), exact::div_by(a.y, d, y), exact::div_by(a.z, d, y));
    }
#endif
    return a / d;
}
// vec3.rs:67-70: self / self.len() -- a square root and a true divide per component
IRIS_DEV V3 normalize_x(V3 a) {
    const float l2 = len_squared(a);
#if IRIS_EXACT_FAST
    // 2^-96 <= |a|^2 <= 2^60, components zero or >= 2^-100: the length is in [2^-48, 2^30], the quotients in [2^-130.., 1] --
    // no: a component of 2^-100 over a length of 2^30 would be denormal, hence the tighter upper bound 2^40 on |a|^2
    if (__builtin_expect(IRIS_ALL_LANES(in_range_bits(l2, IRIS_EX_BITS_2M96, 0x53800000u /* 2^40 */) && numerators_ok(a.x, a.y, a.z)), 1)) {
        const float l = exact::sqrt_fast_pos(l2);
        const exact::Recip y = exact::recip_for(l);
        return v3(exact::div_by(a.x, l, y), exact::div_by(a.y, l, y), exact::div_by(a.z, l, y));
    }
#endif
    return a / __builtin_sqrtf(l2);
}
// the same two in the radiometric code
#if IRIS_RELAXED
IRIS_DEV V3 div3(V3 a, float d) { return a * __builtin_amdgcn_rcpf(d); }
IRIS_DEV V3 normalize(V3 a) { return a * __builtin_amdgcn_rsqf(len_squared(a)); }
#else
IRIS_DEV V3 div3(V3 a, float d) { return div3_x(a, d); }
IRIS_DEV V3 normalize(V3 a) { return normalize_x(a); }
#endif
IRIS_DEV float distance_squared(V3 a, V3 b) { return len_squared(a - b); }
IRIS_DEV float distance(V3 a, V3 b) { return sqrt_g(distance_squared(a, b)); }
IRIS_DEV V3 face_forward(V3 s, V3 n) { return dot(s, n) >= 0.0f ? s : -s; }  // vec3.rs:105-111

// vec3.rs:77-85 (divides by x^2+z^2, not its square root -- as written in the reference)
IRIS_DEV void coordinate_system_from_unit(V3 s, V3 &v2, V3 &v3o) {
    if (fabs_(s.x) > fabs_(s.y)) v2 = div3(v3(-s.z, 0.0f, s.x), s.x * s.x + s.z * s.z);
    else v2 = div3(v3(0.0f, s.z, -s.y), s.y * s.y + s.z * s.z);
    v3o = cross(s, v2);
}

IRIS_RADIOMETRIC_BEGIN
// shading-space trigonometry, vec3.rs:301-355
IRIS_DEV float cos_theta(V3 w) { return w.z; }
IRIS_DEV float cos_2_theta(V3 w) { return w.z * w.z; }
IRIS_DEV float sin_2_theta(V3 w) { return rmax(1.0f - cos_2_theta(w), 0.0f); }
IRIS_DEV float sin_theta(V3 w) { return sqrt_nz(sin_2_theta(w)); }  // max(1 - z*z, 0)
IRIS_DEV float tan_theta(V3 w) { return fdiv(sin_theta(w), cos_theta(w)); }
IRIS_DEV float tan_2_theta(V3 w) { return fdiv(sin_2_theta(w), cos_2_theta(w)); }
IRIS_DEV float cos_phi(V3 w) { float st = sin_theta(w); return st == 0.0f ? 0.0f : rclamp(fdiv(w.x, st), -1.0f, 1.0f); }
IRIS_DEV float sin_phi(V3 w) { float st = sin_theta(w); return st == 0.0f ? 0.0f : rclamp(fdiv(w.y, st), -1.0f, 1.0f); }
IRIS_DEV float cos_2_phi(V3 w) { float c = cos_phi(w); return c * c; }
IRIS_DEV float sin_2_phi(V3 w) { float s = sin_phi(w); return s * s; }
IRIS_DEV bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }

IRIS_DEV F4 f4(float a, float b, float c, float d) { F4 r = {{a, b, c, d}}; return r; }
IRIS_DEV F4 splat(float x) { return f4(x, x, x, x); }
#define IRIS_F4_OP(name, expr)                                                  \
    IRIS_DEV F4 name(F4 a, F4 b) { F4 r; _Pragma("unroll") for (int i = 0; i < 4; i++) { float x = a.v[i], y = b.v[i]; r.v[i] = (expr); } return r; }
IRIS_F4_OP(operator+, x + y)
IRIS_F4_OP(operator*, x * y)
IRIS_F4_OP(operator/, fdiv(x, y))
IRIS_DEV F4 operator*(F4 a, float s) { return a * splat(s); }
IRIS_DEV F4 operator*(float s, F4 a) { return splat(s) * a; }
IRIS_DEV F4 operator/(F4 a, float s) { return a / splat(s); }
IRIS_DEV F4 operator/(float s, F4 a) { return splat(s) / a; }
IRIS_DEV F4 operator+(float s, F4 a) { return splat(s) + a; }
IRIS_DEV float hero(F4 a) { return a.v[0]; }
// a / d for the four wavelengths with ONE reciprocal (hwss_naive.rs divides a SpectralSample by a scalar pdf in several places):
// 3 + 4 x 3 instructions instead of 4 x 11, when every lane's denominator is in [2^-60, 2^60] and its numerators are zero or
// in [2^-60, 2^60] (wave-uniform test, as in div3); otherwise four IEEE divisions.  mono (a compile-time constant at every call
// site): the four components are one value by construction, and one plain division is cheaper than the test.
IRIS_DEV F4 div4(F4 a, float d, bool mono) {
    if (mono) return splat(fdiv(a.v[0], d));
#if IRIS_RELAXED
    const float y = __builtin_amdgcn_rcpf(d);
    return f4(a.v[0] * y, a.v[1] * y, a.v[2] * y, a.v[3] * y);
#else
#if IRIS_EXACT_FAST
    const uint32_t m0 = __float_as_uint(a.v[0]) << 1, m1 = __float_as_uint(a.v[1]) << 1, m2 = __float_as_uint(a.v[2]) << 1, m3 = __float_as_uint(a.v[3]) << 1;
    const uint32_t k0 = m0 - 2u, k1 = m1 - 2u, k2 = m2 - 2u, k3 = m3 - 2u;  // exact::mag_key: +-0 at the top
    const uint32_t k01 = k0 < k1 ? k0 : k1, lo = k01 < k2 ? (k01 < k3 ? k01 : k3) : (k2 < k3 ? k2 : k3);
    const uint32_t m01 = m0 > m1 ? m0 : m1, hi = m01 > m2 ? (m01 > m3 ? m01 : m3) : (m2 > m3 ? m2 : m3);
    if (__builtin_expect(IRIS_ALL_LANES(in_range_bits(d, IRIS_EX_BITS_2M60, IRIS_EX_BITS_2P60) && lo >= IRIS_EX_KEY(IRIS_EX_BITS_2M60) && hi <= (IRIS_EX_BITS_2P60 << 1)), 1)) {
        const exact::Recip y = exact::recip_for(d);
        return f4(exact::div_by(a.v[0], d, y), exact::div_by(a.v[1], d, y), exact::div_by(a.v[2], d, y), exact::div_by(a.v[3], d, y));
    }
#endif
    return f4(a.v[0] / d, a.v[1] / d, a.v[2] / d, a.v[3] / d);
#endif
}
// vec4.rs:55-57: f32x4::reduce_sum is the ordered reduction starting from -0.0
IRIS_DEV float sum(F4 a) { return ((((-0.0f) + a.v[0]) + a.v[1]) + a.v[2]) + a.v[3]; }

IRIS_RADIOMETRIC_END
// ---------------------------------------------------------------- rays (math/ray.rs)
#define IRIS_RAY_EPSILON 0.001f  // ray.rs:5
struct Ray { V3 o, d; };

// ray.rs:61-85: Waechter-Binder integer-ulp origin offset
IRIS_DEV float offset_component(float p, float n) {
    const float ORIGIN = 1.0f / 32.0f, FLOAT_SCALE = 1.0f / 65536.0f, INT_SCALE = 256.0f;
    uint32_t off = (uint32_t)f_as_i32(INT_SCALE * n);
    uint32_t bits = __float_as_uint(p) + (p < 0.0f ? (0u - off) : off);
    return fabs_(p) < ORIGIN ? p + FLOAT_SCALE * n : __uint_as_float(bits);
}
IRIS_DEV V3 offset_origin(V3 p, V3 n) { return v3(offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z)); }
IRIS_DEV Ray ray_new(V3 o, V3 d) { Ray r = {o, normalize(d)}; return r; }  // ray.rs:14-19
// offset_origin(p, flip ? -n : n): the reference's `if .. { offset_origin(p, -n) } else { offset_origin(p, n) }`
// with the sign chosen first, so that the offset arithmetic runs once.
IRIS_DEV V3 offset_origin_signed(V3 p, V3 n, bool flip) { return offset_origin(p, flip ? -n : n); }
IRIS_DEV Ray ray_spawn(V3 o, V3 d, V3 normal) {                             // ray.rs:21-32
    V3 o2 = offset_origin_signed(o, normal, dot(d, normal) < 0.0f);
    Ray r = {o2, normalize(d)};
    return r;
}
IRIS_DEV Ray ray_spawn_to(V3 o, V3 p, V3 normal) {  // ray.rs:34-45 (direction from the OFFSET origin)
    V3 o2 = offset_origin_signed(o, normal, dot(p - o, normal) < 0.0f);
    Ray r = {o2, normalize(p - o2)};
    return r;
}

IRIS_RADIOMETRIC_BEGIN
// ---------------------------------------------------------------- math/mod.rs
IRIS_DEV float fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t) {  // mod.rs:35-59
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    float ei, et, ci;
    if (cos_theta_i > 0.0f) { ei = eta_i; et = eta_t; ci = cos_theta_i; }
    else { ei = eta_t; et = eta_i; ci = fabs_(cos_theta_i); }
    float sin_theta_i = sqrt_nz(rmax(1.0f - ci * ci, 0.0f));
    float sin_theta_t = fdiv(ei, et) * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = sqrt_nz(rmax(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    float r_par = fdiv((et * ci) - (ei * cos_theta_t), (et * ci) + (ei * cos_theta_t));
    float r_perp = fdiv((ei * ci) - (et * cos_theta_t), (ei * ci) + (et * cos_theta_t));
    return (r_par * r_par + r_perp * r_perp) / 2.0f;
}
IRIS_DEV bool refract(V3 wi, V3 n, float eta, V3 &out) {  // mod.rs:61-72
    float cos_theta_i = dot(n, wi);
    float sin_2_theta_i = rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t >= 1.0f) return false;
    float cos_theta_t = sqrt_nz(1.0f - sin_2_theta_t);  // 0 <= sin_2_theta_t < 1 here (or NaN): the argument is at least 2^-24
    out = eta * (-wi) + n * (eta * cos_theta_i - cos_theta_t);
    return true;
}

IRIS_RADIOMETRIC_END
// ---------------------------------------------------------------- sampler (sampling/sampler.rs)
IRIS_DEV uint32_t hash_u32(uint32_t n, uint32_t seed) {  // sampler.rs:74-83
    uint32_t h = n;
#pragma unroll
    for (int i = 0; i < 3; i++) { h *= 1936502639u; h ^= h >> 16; h ^= seed; }
    return h;
}
IRIS_DEV float hash_u32_to_f32(uint32_t n, uint32_t seed) {  // sampler.rs:85-88
    return (float)hash_u32(n, seed) * (1.0f / 4294967296.0f);
}
IRIS_DEV uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
IRIS_DEV uint32_t lk_scramble(uint32_t n, uint32_t seed) {
    n ^= n * 0x3d20adeau; n += seed; n *= (seed >> 16) | 1u; n ^= n * 0x05526c56u; n ^= n * 0x53a22864u;
    return n;
}
IRIS_DEV float unit_float_from_bits(uint32_t v) { return __uint_as_float((v >> 9) | 0x3f800000u) - 1.0f; }

// Sobol direction vectors.  The XOR over the 16 index bits is done a nibble at a time from a table in
// LDS: lut[(set * 4 + j) * 16 + v] = XOR of the vectors of bits 4j .. 4j+3 of `set` selected by the 4-bit
// value v (its most significant bit selects bit 4j), the four 16-bit reversed vectors of a bit packed as
// {d0 | d1 << 16, d2 | d3 << 16}.  The table covers the first IRIS_SOBOL_LUT_SETS sets (128 draws: every path
// of 15 bounces); later sets read the direction vectors bit by bit from the constant table in global memory.
// Lanes of a wave are at the same set and differ in v, i.e. they read different banks or the same address.
#ifndef IRIS_SOBOL_LUT_SETS
#define IRIS_SOBOL_LUT_SETS 32u
#endif
#define IRIS_SOBOL_LUT_ENTRIES (IRIS_SOBOL_LUT_SETS * 4u * 16u)
typedef const uint2 *SobolLds;
IRIS_DEV uint2 sobol_packed_vectors(uint32_t set, uint32_t bit) {
    const unsigned short *v = IRIS_SOBOL_REV16[set][bit];
    return make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16));
}
IRIS_DEV uint2 sobol_lut_entry(uint32_t e) {
    const uint32_t set = e >> 6, j = (e >> 4) & 3u, v = e & 15u;
    uint2 x = make_uint2(0u, 0u);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        if (v & (8u >> i)) {
            uint2 d = sobol_packed_vectors(set, 4u * j + i);
            x.x ^= d.x; x.y ^= d.y;
        }
    }
    return x;
}

// This build's Owen-scrambled Sobol (docs/SPEC.md); stands where the reference calls the external
// crate: sobol::sample_4d(index, dimension_set, seed), sampler.rs:30-34.
IRIS_DEV void sobol_sample_4d(SobolLds lut, uint32_t index, uint32_t set, uint32_t seed, float out[4]) {
    uint32_t shuffled_rev = lk_scramble(__builtin_bitreverse32(index), lowbias32(seed));
    uint32_t x01 = 0u, x23 = 0u;
    if (lut != nullptr && set < IRIS_SOBOL_LUT_SETS) {  // lut == nullptr: the kernel keeps no nibble table (hierarchy variant)
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            uint2 v = lut[(set * 4u + j) * 16u + ((shuffled_rev >> (28u - 4u * j)) & 15u)];
            x01 ^= v.x;
            x23 ^= v.y;
        }
    } else {
        // No table in LDS: the 16 packed vectors of the set (128 contiguous bytes of the constant table) are fetched with eight
        // independent 16-byte loads, four at a time, and masked by the index bits -- one memory round trip per half instead of
        // one per set bit (the loop this replaces read a vector, waited, and went on to the next bit: up to sixteen dependent
        // L1/L2 round trips per set, three sets per bounce, in the kernel that is latency-bound to begin with).
        const uint4 *row = (const uint4 *)&IRIS_SOBOL_REV16[set][0][0];
#pragma unroll
        for (uint32_t h = 0; h < 2; h++) {
            uint4 q[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) q[k] = row[4u * h + k];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t b = 8u * h + 2u * k;  // index bit b after the shuffle = bit 31 - b of shuffled_rev
                const uint32_t m0 = (uint32_t)((int32_t)(shuffled_rev << b) >> 31), m1 = (uint32_t)((int32_t)(shuffled_rev << (b + 1u)) >> 31);
                x01 ^= (q[k].x & m0) ^ (q[k].z & m1);
                x23 ^= (q[k].y & m0) ^ (q[k].w & m1);
            }
        }
    }
    uint32_t xr[4] = {x01 & 0xffffu, x01 >> 16, x23 & 0xffffu, x23 >> 16};
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        uint32_t dseed = lowbias32(seed ^ (0x9e3779b9u * (4u * set + k + 1u)));
        out[k] = unit_float_from_bits(__builtin_bitreverse32(lk_scramble(xr[k], dseed)));
    }
}

// sampler.rs:4-10.  The reference keeps a 4-entry buffer that is refilled by whichever gen_0_1 call
// finds it empty.  On a wave that would run the (long) refill at every one of the nine call sites of a
// bounce for the quarter of the lanes that happen to need it there.  Here the draw STREAM is the same
// (value j of a path is element 3 - j%4 of set j/4: the reverse pop of sampler.rs:24-26), but it is
// produced ahead of use at wave-uniform points (ensure_draws: once per camera sample and once per
// bounce) into a 16-entry ring per lane in LDS, and gen_0_1 is one LDS read.
#define IRIS_BLOCK 256
#define IRIS_RING_SLOTS 16u
typedef float *DrawRing;  // LDS: ring[slot * IRIS_BLOCK], already offset by threadIdx.x
struct Sampler {
    uint32_t scramble, index;
    uint32_t draws;  // values consumed so far
    uint32_t gen;    // values generated so far (a multiple of 4; set k holds values 4k .. 4k+3)
};
IRIS_DEV Sampler sampler_new(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed) {  // sampler.rs:13-21
    Sampler s;
    s.scramble = hash_u32(x ^ (y << 16), seed);
    s.index = sample_index;
    s.draws = 0u; s.gen = 0u;
    return s;
}
// One 4-D set in pop order: out[0] is the value popped first (buf[3] of sampler.rs:27-46).
IRIS_DEV void sampler_set(SobolLds tab, uint32_t index, uint32_t dimension, uint32_t scramble, float out[4]) {
    float b[4];
    if (dimension < IRIS_SOBOL_SETS) {
        sobol_sample_4d(tab, index, dimension, hash_u32(dimension, scramble), b);
    } else {  // sampler.rs:36-41 (from set 64 on: docs/SPEC.md)
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) b[k] = hash_u32_to_f32((dimension + k) ^ (index << 16), scramble);
    }
    out[0] = b[3]; out[1] = b[2]; out[2] = b[1]; out[3] = b[0];
}
// Makes at least `need` (<= 12) unread values available to an active lane.  Wave-uniform loop: every
// lane that is short generates its next set in the same pass.
IRIS_DEV void ensure_draws(SobolLds tab, DrawRing ring, Sampler &s, uint32_t need, bool active) {
    for (;;) {
        // signed: after sampler_rewind `gen` sits up to 3 values BEHIND `draws`
        const bool want = active && (int32_t)(s.gen - s.draws) < (int32_t)need;
        if (__ballot(want) == 0ull) break;
        if (want) {
            float v[4];
            sampler_set(tab, s.index, s.gen >> 2, s.scramble, v);
            const uint32_t slot = s.gen & (IRIS_RING_SLOTS - 1u);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) ring[(slot + k) * IRIS_BLOCK] = v[k];
            s.gen += 4u;
        }
    }
}
// After a path state has been moved to another lane (pool kernels): the set in progress is generated again
// (ensure_draws then starts from the set that holds value `draws`; at most 3 + 12 < 16 ring slots are live).
IRIS_DEV void sampler_rewind(Sampler &s) { s.gen = s.draws & ~3u; }
IRIS_DEV float gen_0_1(DrawRing ring, Sampler &s) {  // sampler.rs:23-50; the caller has run ensure_draws
    float r = ring[(s.draws & (IRIS_RING_SLOTS - 1u)) * IRIS_BLOCK];
    s.draws++;
    return r;
}
IRIS_DEV float gen_golden_ratio(const Sampler &s) {  // sampler.rs:65-71 (u32::MAX as f32 == 2^32)
    uint32_t u = (s.index + s.scramble) * 2654435769u;
    return (float)u / 4294967296.0f;
}

IRIS_RADIOMETRIC_BEGIN
// ---------------------------------------------------------------- warps (sampling/mod.rs), MIS (mis.rs)
IRIS_DEV float pdf_cone(float cos_theta_max) { return rcp_g(2.0f * PI_F * (1.0f - cos_theta_max)); }  // mod.rs:22-24
IRIS_DEV V3 unit_sphere(float r1, float r2, bool pin = true) {                                                              // mod.rs:26-31
    float z = 1.0f - 2.0f * r1;
    float r = sqrt_nz(rmax(1.0f - z * z, 0.0f));
    float phi = 2.0f * PI_F * r2;
    float s, c;
    iris_sincos(phi, s, c, pin);
    return v3(r * c, r * s, z);
}
IRIS_DEV V3 cosine_unit_hemisphere(float r1, float r2, bool pin = true) {  // mod.rs:33-54
    float x_off = 2.0f * r1 - 1.0f, y_off = 2.0f * r2 - 1.0f;
    float x = 0.0f, y = 0.0f;
    if (!(x_off == 0.0f && y_off == 0.0f)) {
        float r, theta;
        if (fabs_(x_off) > fabs_(y_off)) { r = x_off; theta = PI_F / 4.0f * fdiv(y_off, x_off); }
        else { r = y_off; theta = (PI_F / 2.0f) - (PI_F / 4.0f) * fdiv(x_off, y_off); }
        float s, c;
        iris_sincos(theta, s, c, pin);
        x = r * c; y = r * s;
    }
    float z = sqrt_nz(rmax(1.0f - x * x - y * y, 0.0f));  // (1 - x*x) is <= 0 or >= 2^-24; minus y*y: <= 0 or >= 2^-49
    return v3(x, y, z);
}
IRIS_DEV float balance_heuristic_1(F4 f) { return fdiv(hero(f), sum(f)); }             // mis.rs:3-5
IRIS_DEV float balance_heuristic_2(F4 f, F4 g) { return fdiv(hero(f), sum(f + g)); }  // mis.rs:8-10

IRIS_RADIOMETRIC_END
// ---------------------------------------------------------------- spectra, colour
#define LAMBDA_MIN_NM 360.0f
#define LAMBDA_MAX_NM 830.0f
#define LAMBDA_RANGE_NM (LAMBDA_MAX_NM - LAMBDA_MIN_NM)

IRIS_DEV float rotate_n(float h, uint32_t n) {  // wavelength.rs:55-64
    float lambda = h + (LAMBDA_RANGE_NM / 4.0f) * (float)n;
    return lambda >= LAMBDA_MAX_NM ? lambda - LAMBDA_RANGE_NM : lambda;
}
IRIS_DEV F4 wavelength_new(float h) { return f4(h, rotate_n(h, 1), rotate_n(h, 2), rotate_n(h, 3)); }  // wavelength.rs:14-24

IRIS_RADIOMETRIC_BEGIN
IRIS_DEV float spectrum_evaluate_single(const iris_spectrum &s, float lambda) {
    if (s.kind == IRIS_SPEC_CONSTANT) return s.c0;  // constant.rs:15-23
    // upsample.rs:18-24 / :35-41 -- explicit fused multiply-adds in the reference
    float x = __builtin_fmaf(__builtin_fmaf(s.c0, lambda, s.c1), lambda, s.c2);
    float y = rcp_g(rsqrt_exact(__builtin_fmaf(x, x, 1.0f)));
    float v = __builtin_fmaf(0.5f * x, y, 0.5f);
    return s.kind == IRIS_SPEC_UPSAMPLED ? v : s.hdr * v;
}
// `mono` (a compile-time constant at every call site): the scene holds constant spectra only, so the four
// wavelengths carry the same value and the compiler merges the four copies of every operation on them.
IRIS_DEV F4 spectrum_evaluate(const iris_spectrum &s, F4 wl, bool mono) {  // spectrum/mod.rs:16-20
    if (mono) return splat(s.c0);
    return f4(spectrum_evaluate_single(s, wl.v[0]), spectrum_evaluate_single(s, wl.v[1]),
              spectrum_evaluate_single(s, wl.v[2]), spectrum_evaluate_single(s, wl.v[3]));
}
// sample.rs:45-52 + color.rs:21-34: truncating table lookup per wavelength, summed a+b+c+d
IRIS_DEV V3 spectral_to_xyz(F4 value, F4 wl) {
    const float CIE_Y_INTEGRAL = 116.661843131358f;
    const float scale = (LAMBDA_MAX_NM - LAMBDA_MIN_NM) / CIE_Y_INTEGRAL;
    float X[4], Y[4], Z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t index = f_as_index(wl.v[k]) - 360u;
        if (index > 470u) index = 470u;  // the reference indexes unchecked; a valid path never leaves 360..830
        float v = value.v[k] * scale;
        X[k] = IRIS_CIE_X[index] * v;
        Y[k] = IRIS_CIE_Y[index] * v;
        Z[k] = IRIS_CIE_Z[index] * v;
    }
    return v3(((X[0] + X[1]) + X[2]) + X[3], ((Y[0] + Y[1]) + Y[2]) + Y[3], ((Z[0] + Z[1]) + Z[2]) + Z[3]);
}

IRIS_RADIOMETRIC_END
// ---------------------------------------------------------------- scene on the device
struct SceneDev {
    const float4 *spheres;      // (cx, cy, cz, r), primitive order
    const int4 *prim_info;      // (material_index, light_index, bits of albedo / PI when the material is Lambertian over a constant spectrum, 0); -1 = None
    const unsigned char *prim_kind;  // the kind of the primitive's BSDF (IRIS_BSDF_*), 0 for Lambertian, none or absorbing: the register kernel sorts hits by it
    const iris_bsdf *bsdfs;
    const iris_spectrum *spectra;
    const float *spec_over_pi;  // per spectrum: c0 / PI for a constant spectrum (the IEEE quotient, formed once on the host)
    const iris_light *lights;
    // per light, three float4 read side by side instead of the chain lights -> (spectra, spheres): the light's sphere; (prim,
    // spectrum kind, c0, c1); (c2, hdr, 0, 0)
    const float4 *light_rec;
    uint32_t n_prims, n_lights;
    // Bounding-volume hierarchy over the spheres (long lists only; n_bvh_nodes == 0: none).  Node k is two
    // float4: (lo.xyz, first) and (hi.xyz, count); count > 0: leaf holding leaf-ordered spheres
    // [first, first + count); count == 0: inner node whose children are nodes first and first + 1.
    const float4 *bvh_nodes;
    // The same nodes as the traversal reads them (iris_bvh_build.h): 16 bytes per node, boxes on a 16-bit grid, both children of
    // a node in one aligned 32-byte pair -- half the bytes, and so half the vector-L1 requests, of the float records, which is
    // what bounds the traversal.  bvh_nodes keeps the float boxes; the root's is tested from there.
    const uint4 *bvh_q;
    float bvh_g0[3], bvh_s, bvh_inv_s;
    const float4 *bvh_spheres;   // (cx, cy, cz, r * r) in leaf order
    const uint32_t *bvh_prim;    // primitive index of each leaf-ordered sphere
    uint32_t n_bvh_nodes;
    uint32_t bvh_big_first, bvh_big_count;  // leaf-ordered spheres kept out of the tree (scene-sized ones): tested first
    // Environment map (docs/SPEC.md section 6; env_w == 0: none): octahedral layout, one float4 per texel
    // (c0, c1, c2, hdr_coefficient) = UpsampledHdrSpectrum; sampling tables built at upload.
    const float4 *env_texels;
    const float *env_func;      // importance x texel solid-angle weight
    // The CDFs as the device searches them (EnvSearch below): entries 0 .. n-1 of each CDF row padded to whole
    // blocks of 32 with a value no draw reaches, plus one coarser level per factor of 32 (every 32nd entry).
    const float *env_cond_lvl[3];
    const float *env_marg_lvl[3];
    uint32_t env_cond_stride[3];  // floats per row at each level (marginal: one row)
    uint32_t env_cond_levels, env_marg_levels;
    uint32_t env_w, env_h;
    float env_integral;
};

// the emission spectrum and the primitive of light i from its flattened record
IRIS_DEV iris_spectrum light_spectrum(const SceneDev &sc, uint32_t i, int &prim_out) {
    const float4 r1 = sc.light_rec[3u * i + 1u], r2 = sc.light_rec[3u * i + 2u];
    prim_out = __float_as_int(r1.x);
    iris_spectrum sp = {__float_as_uint(r1.y), r1.z, r1.w, r2.x, r2.y};
    return sp;
}
// shape/mod.rs:14-38
struct Hit { V3 point, normal, tangeant, bitangeant; bool back_face; };
IRIS_DEV V3 world_to_shading(const Hit &h, V3 w) { return v3(dot(h.bitangeant, w), dot(h.tangeant, w), dot(h.normal, w)); }
IRIS_DEV V3 shading_to_world(const Hit &h, V3 s) {
    return v3(h.bitangeant.x * s.x + h.tangeant.x * s.y + h.normal.x * s.z,
              h.bitangeant.y * s.x + h.tangeant.y * s.y + h.normal.y * s.z,
              h.bitangeant.z * s.x + h.tangeant.z * s.y + h.normal.z * s.z);
}

// sphere.rs:26-35 + :53: the ray parameter of the first root > 0, or a negative number for None.
// `sp` is a staged sphere: (cx, cy, cz, radius * radius) -- the product the reference forms at sphere.rs:29.
// sphere.rs:31-35, :53 take `temp = (-half_b - root) / a`, and `(-half_b + root) / a` if that is not > 0.
// For 0.5 < a < 1.5 (a = d.d of a normalised direction; `a_unit`) the sign test can be made on the numerators:
// num / a > 0 <=> num > 0, because a positive quotient is above half of the smallest denormal and cannot round
// to zero, and a non-positive (or NaN) numerator gives a quotient that is not > 0.  So the root is chosen first
// and only the chosen numerator is divided -- the same value the reference computes -- and a sphere wholly behind
// the origin (both numerators <= 0: the sphere a bounce ray starts on, in every scan) costs no division at all.
// Returns the first root > 0, or a value that is not > 0.
IRIS_DEV bool a_is_unit(float a) { return a > 0.5f && a < 1.5f; }
// `ya`: exact::recip_for(a), formed once per ray (used only when a_unit: then a is inside its domain).
IRIS_DEV float first_positive_root(float half_b, float root, float a, bool a_unit, exact::Recip ya) {
    const float n1 = -half_b - root, n2 = -half_b + root;
    if (a_unit) {
        const float num = n1 > 0.0f ? n1 : n2;
        float temp = -1.0f;
#if IRIS_EXACT_FAST
        // every lane's numerator is either not > 0 (no division at all) or inside the domain: 0.5 < a < 1.5 keeps the quotient there
        if (__builtin_expect(IRIS_ALL_LANES(!(num > 0.0f) || in_range_bits(num, IRIS_EX_BITS_2M100, IRIS_EX_BITS_2P100)), 1)) {
            if (num > 0.0f) temp = exact::div_by(num, a, ya);
            return temp;
        }
#endif
        if (num > 0.0f) temp = num / a;
        return temp;
    }
    float temp = n1 / a;
    if (!(temp > 0.0f)) temp = n2 / a;
    return temp;
}
IRIS_DEV float sphere_hit_t(float4 sp, const Ray &ray, float a) {
    const exact::Recip ya = exact::recip_for(a);
    V3 oc = ray.o - v3(sp.x, sp.y, sp.z);
    float half_b = dot(ray.d, oc);
    float c = len_squared(oc) - sp.w;
    float discrim = half_b * half_b - a * c;
    float t = -1.0f;
    if (discrim > 0.0f) {
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_is_unit(a), ya);
        if (temp > 0.0f) t = temp;
    }
    return t;
}
// The same test inside a closest-hit scan: `if t < closest_t && t > 0 { closest = .. }` (scene.rs:123-126).
// Almost every test of a long list ends at `discrim > 0` being false, so the bookkeeping of the closest
// hit lives inside that branch and the common path is the discriminant alone.
IRIS_DEV void sphere_test(float4 sp, const Ray &ray, float a, bool a_unit, exact::Recip ya, int index, float &closest_t, int &closest) {
    V3 oc = ray.o - v3(sp.x, sp.y, sp.z);
    float half_b = dot(ray.d, oc);
    float c = len_squared(oc) - sp.w;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_unit, ya);
        if (temp > 0.0f && temp < closest_t) { closest_t = temp; closest = index; }
    }
}
IRIS_DEV float4 stage_sphere(float4 sp) { return make_float4(sp.x, sp.y, sp.z, sp.w * sp.w); }
// Filler behind the last staged sphere (the scans read the window in batches): c = |oc|^2 + inf, so the
// discriminant is -inf (or NaN) and the test can never report a hit.
IRIS_DEV float4 stage_filler() { return make_float4(0.0f, 0.0f, 0.0f, -__builtin_inff()); }
#define IRIS_SCAN_PAD 8u
// sphere.rs:36-40: the shading frame of a hit at parameter t (built once, for the winner only)
IRIS_DEV Hit sphere_frame(float4 sp, const Ray &ray, float t) {
    Hit h;
    h.point = ray.o + ray.d * t;
    h.normal = div3(h.point - v3(sp.x, sp.y, sp.z), sp.w);
    h.tangeant = normalize(cross(v3(0.0f, 1.0f, 0.0f), h.normal));
    h.bitangeant = cross(h.normal, h.tangeant);
    h.back_face = dot(h.normal, ray.d) >= 0.0f;
    return h;
}

// The sphere list as the scans see it: the whole list in LDS when it fits, else streamed through an
// LDS window chunk by chunk by the whole workgroup.
struct SphereStage {
    float4 *lds;           // window
    uint32_t cap;          // window capacity in spheres (a multiple of IRIS_SCAN_PAD)
    bool resident;         // n_prims <= cap and the window already holds the list
    bool bvh;              // the scene has a hierarchy: the scans traverse it instead of the list
    uint32_t *stack;       // bvh: this lane's traversal stack in LDS, entry k at stack[k * IRIS_BLOCK]
    // Entries from stack_lds_n on live in global memory (entry k at stack_hi[(k - stack_lds_n) * stack_hi_stride]): the
    // far child is stacked only when both boxes are hit, so a ray rarely holds more than a few entries, and LDS --
    // 1 KiB per level and workgroup -- is what limits the resident waves of the latency-bound traversal.
    uint32_t stack_lds_n;
    uint32_t *stack_hi;
    uint32_t stack_hi_stride;
};
IRIS_DEV void bvh_push(const SphereStage &st, uint32_t sp, uint32_t ref) {
    if (sp < st.stack_lds_n) st.stack[sp * IRIS_BLOCK] = ref;
    else st.stack_hi[(size_t)(sp - st.stack_lds_n) * st.stack_hi_stride] = ref;
}
IRIS_DEV uint32_t bvh_pop(const SphereStage &st, uint32_t sp) {
    return sp < st.stack_lds_n ? st.stack[sp * IRIS_BLOCK] : st.stack_hi[(size_t)(sp - st.stack_lds_n) * st.stack_hi_stride];
}

// One pass over `n` staged spheres (n real ones, followed by fillers up to a multiple of IRIS_SCAN_PAD),
// B at a time: the B LDS reads of a batch are issued together, so their latency is paid once per batch.
// n is wave-uniform, so the bound check of the last batch is a scalar branch and a filler costs no test.
template <uint32_t B>
IRIS_DEV void scan_window(const float4 *win, uint32_t n, uint32_t base, const Ray &ray, float a, float &closest_t, int &closest) {
    const bool a_unit = a_is_unit(a);
    const exact::Recip ya = exact::recip_for(a);  // one reciprocal of a = d.d per scan (sphere.rs:33 divides every root by it)
    for (uint32_t i = 0; i < n; i += B) {
        float4 sp[B];
#pragma unroll
        for (uint32_t k = 0; k < B; k++) sp[k] = win[i + k];
#pragma unroll
        for (uint32_t k = 0; k < B; k++) {
            if (__builtin_amdgcn_readfirstlane(i + k) < __builtin_amdgcn_readfirstlane(n))
                sphere_test(sp[k], ray, a, a_unit, ya, (int)(base + i + k), closest_t, closest);
        }
    }
}

// ---- the same closest-hit query through the hierarchy ----
// The answer must be the linear scan's, bit for bit: the closest t > 0 with the lowest primitive index on
// ties (scene.rs:116-132).  Every sphere is tested with the very same sphere arithmetic; the hierarchy only
// decides which spheres are tested.  A subtree is skipped only if the ray misses its (host-inflated) box or
// enters it strictly behind the closest hit so far, so a sphere that could win or tie is always visited;
// visiting order is free, hence the explicit tie rule on the index.
// How far outside a sphere a ray can pass and still be reported as a hit by the f32 arithmetic of sphere.rs:26-35 -- which the
// list scan runs on every sphere, and the hierarchy must therefore never cull.  With u = 2^-24 and D = |o - c|: the computed
// half_b^2 is off by at most 7u D^2, |oc|^2 - r^2 by 6u D^2, a*c by another 4u D^2, the difference rounds once more:
// |error(discrim)| <= 18u D^2 < 1.1e-6 D^2.  A ray missing by m (discrim = r^2 - m^2 for a unit direction) can thus come out
// positive only if m < sqrt(r^2 + 1.1e-6 D^2) <= r + 1.05e-3 D, and the same error moves a computed root by at most 1.05e-3 D
// along the ray.  So every box is grown, per ray and per box, by 1.2e-3 x (an upper bound of the distance from the ray's origin
// to any point of the box, hence to any centre inside it: the 1-norm of the per-axis farthest plane distances, which the slab
// test has at hand): a sub-pixel sphere far away is "fat" for the reference's arithmetic, and is for the traversal too, while
// boxes near the origin stay tight.  (The host adds 16 ulps of the coordinates for the rounding of the slab arithmetic itself.)
#define IRIS_BVH_PAD 1.2e-3f
IRIS_DEV float box_entry_d(float dx0, float dx1, float dy0, float dy1, float dz0, float dz1, V3 inv, float closest_t) {
    const float pad = IRIS_BVH_PAD * ((__builtin_fmaxf(fabs_(dx0), fabs_(dx1)) + __builtin_fmaxf(fabs_(dy0), fabs_(dy1))) + __builtin_fmaxf(fabs_(dz0), fabs_(dz1)));
    float tx0 = (dx0 - pad) * inv.x, tx1 = (dx1 + pad) * inv.x;
    float ty0 = (dy0 - pad) * inv.y, ty1 = (dy1 + pad) * inv.y;
    float tz0 = (dz0 - pad) * inv.z, tz1 = (dz1 + pad) * inv.z;
    // fminf / fmaxf ignore a NaN operand (0 * inf when the origin lies on a slab plane of an axis-parallel ray)
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fmaxf(__builtin_fminf(tz0, tz1), 0.0f));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fmaxf(tz0, tz1));
    return (tmin <= tmax && tmin <= closest_t) ? tmin : __builtin_inff();  // inf: do not enter
}
IRIS_DEV float box_entry(float4 lo, float4 hi, V3 o, V3 inv, float closest_t) {
    return box_entry_d(lo.x - o.x, hi.x - o.x, lo.y - o.y, hi.y - o.y, lo.z - o.z, hi.z - o.z, inv, closest_t);
}
// The same test on a 16-bit grid record, in grid coordinates: the ray's origin is moved to the grid once per ray (og = (o - grid
// origin) / scale, inv_g = scale / d), the record's six integers convert exactly (one v_cvt_f32_u32 with a 16-bit source select
// each), and the distances come out in the ray's own parameter.  What the grid arithmetic can lose: og carries a relative error of
// 3 x 2^-24 (a subtraction, the rounded reciprocal of the scale, a product), i.e. at most 0.012 units for an origin inside the grid
// and 1.8e-7 |og| outside; the record is rounded outwards by up to a unit and then grown by one more (the host), which covers the
// first, and the per-ray, per-box pad of 1.2e-3 x distance, derived with 1.05e-3 needed, covers the second a hundred times over.
struct BvhRay { V3 og, inv_g; };
IRIS_DEV BvhRay bvh_ray(const SceneDev &sc, V3 o, V3 inv) {
    BvhRay r;
    r.og = v3((o.x - sc.bvh_g0[0]) * sc.bvh_inv_s, (o.y - sc.bvh_g0[1]) * sc.bvh_inv_s, (o.z - sc.bvh_g0[2]) * sc.bvh_inv_s);
    r.inv_g = v3(inv.x * sc.bvh_s, inv.y * sc.bvh_s, inv.z * sc.bvh_s);
    return r;
}
IRIS_DEV float box_entry_q(uint4 c, const BvhRay &r, float closest_t) {
    const float lx = (float)(c.x & 0xffffu), ly = (float)(c.x >> 16), lz = (float)(c.y & 0xffffu);
    const float hx = (float)(c.y >> 16), hy = (float)(c.z & 0xffffu), hz = (float)(c.z >> 16);
    return box_entry_d(lx - r.og.x, hx - r.og.x, ly - r.og.y, hy - r.og.y, lz - r.og.z, hz - r.og.z, r.inv_g, closest_t);
}
// one leaf-ordered sphere, with the tie rule on the primitive index (visiting order is not index order)
IRIS_DEV void bvh_sphere_test(const SceneDev &sc, uint32_t slot, const Ray &ray, float a, exact::Recip ya, float &closest_t, int &closest) {
    const float4 sp4 = sc.bvh_spheres[slot];
    const int index = (int)sc.bvh_prim[slot];
    V3 oc = ray.o - v3(sp4.x, sp4.y, sp4.z);
    float half_b = dot(ray.d, oc);
    float c = len_squared(oc) - sp4.w;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {  // sphere.rs:31-35, :53
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_is_unit(a), ya);
        if (temp > 0.0f && (temp < closest_t || (temp == closest_t && index < closest))) { closest_t = temp; closest = index; }
    }
}
// A node reference is its (first, count) pair packed into one word: count (0 = inner node) in the top three
// bits.  References come out of the fetch that tested the node's box, so a node's own record is never read
// again; a stacked reference whose subtree a later hit has made unreachable fails its children's box tests.
IRIS_DEV uint32_t bvh_ref(float4 lo, float4 hi) { return __float_as_uint(lo.w) | (__float_as_uint(hi.w) << 29); }
// ANY = false: the closest hit (closest_t = inf, closest = -1 on entry).
// ANY = true: is there a sphere that a closest-hit scan would prefer to a hit at distance closest_t by primitive `tie`,
// i.e. one hit at 0 < t < closest_t, or at t == closest_t with a lower index than `tie`?  (closest_t = the limit on entry,
// unchanged on return; closest = the index of one such sphere, or -1.)  The limit culls boxes like a hit found at that
// distance would, and the lane leaves the tree at the first such sphere: what a visibility test needs, and all it needs.
IRIS_DEV void bvh_sphere_test_any(const SceneDev &sc, uint32_t slot, const Ray &ray, float a, exact::Recip ya, float limit, int tie, int &closest) {
    const float4 sp4 = sc.bvh_spheres[slot];
    V3 oc = ray.o - v3(sp4.x, sp4.y, sp4.z);
    float half_b = dot(ray.d, oc);
    float c = len_squared(oc) - sp4.w;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {  // sphere.rs:31-35, :53
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_is_unit(a), ya);
        if (temp > 0.0f && (temp < limit || temp == limit)) {  // at the limit itself only spheres that would win the tie: index < tie
            const int index = (int)sc.bvh_prim[slot];
            if (temp < limit || index < tie) closest = index;
        }
    }
}
// 1 / d per axis for the slab tests.  v_rcp_f32 (1 ulp): the boxes are grown by 1.2e-3 of the distance, which is some ten
// thousand times the effect of that ulp; both traversals (bvh_closest, bvh_pooled) use the same value, so they enter the same
// boxes and count the same sphere tests.
IRIS_DEV V3 bvh_inv_dir(V3 d) { return v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }
template <bool ANY>
IRIS_DEV void bvh_closest(const SceneDev &sc, const SphereStage &st, const Ray &ray, float a, float &closest_t, int &closest, unsigned long long &tests, int tie = 0x7fffffff) {
    const V3 inv = bvh_inv_dir(ray.d);
    const BvhRay br = bvh_ray(sc, ray.o, inv);
    const exact::Recip ya = exact::recip_for(a);
    uint32_t n_tests = sc.bvh_big_count;
    for (uint32_t k = 0; k < sc.bvh_big_count; k++) {
        if (ANY) bvh_sphere_test_any(sc, sc.bvh_big_first + k, ray, a, ya, closest_t, tie, closest);
        else bvh_sphere_test(sc, sc.bvh_big_first + k, ray, a, ya, closest_t, closest);
    }
    if (ANY && closest >= 0) { tests += n_tests; return; }
    const float4 root_lo = sc.bvh_nodes[0], root_hi = sc.bvh_nodes[1];
    uint32_t ref = bvh_ref(root_lo, root_hi);
    uint32_t sp = 0u;
    bool have = box_entry(root_lo, root_hi, ray.o, inv, closest_t) < __builtin_inff();
    // "while-while": every lane first descends through inner nodes until it holds a leaf (or has nothing left),
    // then the lanes of the wave test their leaves together -- the two kinds of work are not interleaved lane by
    // lane, which is what keeps the lanes of a wave on the same instructions.
    for (;;) {
        while (have && (ref >> 29) == 0u) {
            const uint32_t first = ref & 0x1fffffffu;
            const uint4 cl = sc.bvh_q[first - 1u], cr = sc.bvh_q[first];
            const float tl = box_entry_q(cl, br, closest_t), tr = box_entry_q(cr, br, closest_t);
            const bool hl = tl < __builtin_inff(), hr = tr < __builtin_inff();
            if (hl || hr) {
                const bool left_first = hl && (!hr || tl <= tr);
                const uint32_t ref_l = cl.w, ref_r = cr.w;
                if (hl && hr) {  // the farther child waits on the stack (depth <= IRIS_BVH_STACK by construction)
                    bvh_push(st, sp, left_first ? ref_r : ref_l);
                    sp++;
                }
                ref = left_first ? ref_l : ref_r;
            } else if (sp != 0u) {
                sp--;
                ref = bvh_pop(st, sp);
            } else {
                have = false;
            }
        }
        if (!have) break;
        const uint32_t first = ref & 0x1fffffffu, count = ref >> 29;
        for (uint32_t k = 0; k < count; k++) {
            if (ANY) bvh_sphere_test_any(sc, first + k, ray, a, ya, closest_t, tie, closest);
            else bvh_sphere_test(sc, first + k, ray, a, ya, closest_t, closest);
        }
        n_tests += count;
        if (ANY && closest >= 0) {
            have = false;
        } else if (sp != 0u) {
            sp--;
            ref = bvh_pop(st, sp);
        } else {
            have = false;
        }
    }
    tests += n_tests;
}

// ---- the same queries for a whole wave at once, with the rays detached from the lanes that own them ----
// A ray's traversal takes between a handful and a few hundred steps, so a wave whose lanes each walk their own ray waits for
// its slowest lane: config D ran at a quarter of its lanes.  Here the wave's rays -- the extension ray of every path that goes
// on (closest hit) and the light-sample visibility query of every path that wants one (any hit up to a limit) -- live in a
// small pool in LDS (two fixed slots per lane: its extension ray and its query), and lanes WALK rays independently of who
// owns them: a lane that finishes a ray takes the next pending one, the result goes back to the slot, the owner reads it.
// A pass does not wait for its slowest ray: when nothing is pending and fewer than IRIS_POOL_SUSPEND lanes are still walking,
// the walkers park their rays (ray, node, stack) in global memory and the wave goes back to shading the paths whose rays are
// done; the next pass resumes the parked rays among the new ones.  A path waits until BOTH of its rays are back, so the
// additions to its radiance happen in the reference's order.  The lanes of a wave run in lockstep, so handing out slots
// needs no atomics: ballots and bit counts.  Per ray the traversal is the one of bvh_closest, step for step (same boxes
// entered, same spheres tested in the same order): results and test counts are the same bit for bit.
#define IRIS_POOL_CAP 128u      // slots per wave: slot l = lane l's extension ray, slot 64 + l = its visibility query
#define IRIS_POOL_FIELDS 8u     // o.xyz, d.xyz, limit / closest t so far, closest primitive so far (biased while the ray is out)
#define IRIS_POOL_OUT 0x40000000  // field 7 holds (closest - IRIS_POOL_OUT) until the ray is done, then closest (>= -1)
#ifndef IRIS_POOL_FETCH_MIN
#define IRIS_POOL_FETCH_MIN 8u  // idle lanes that make a fetch worth its instructions (it runs for the idle lanes only)
#endif
#ifndef IRIS_POOL_SUSPEND
#define IRIS_POOL_SUSPEND 16u   // a pass ends when nothing is pending and fewer lanes than this are walking (8..24 measure alike; 32 and more starve the shade phase)
#endif
#ifndef IRIS_POOL_MIN_ROUNDS
#define IRIS_POOL_MIN_ROUNDS 6u // ... but not before this many rounds, so that parked rays move on in every pass
#endif
#ifndef IRIS_POOL_INNER_PCT
#define IRIS_POOL_INNER_PCT 50u // a round steps through inner nodes when at least this share of the walking lanes is at one, else through leaves (35 .. 80 measure within 2 %)
#endif
#ifndef IRIS_POOL_INNER_REPS
#define IRIS_POOL_INNER_REPS 3  // inner-node steps per such round: the round's ballots, counts and branch are paid once for all of them (1: 154.7 ms, 2: 147.2, 3: 145.1, 4: 146.2, 8: 147.8 on config D)
#endif
#define IRIS_WALK_WORDS (10u + IRIS_BVH_STACK)  // parked walker: o, d, closest t, closest, node reference, (slot, kind, stack depth), stack
IRIS_DEV uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// position of the r-th set bit (r = 0: the lowest) of a 64-bit mask that has more than r bits set
IRIS_DEV uint32_t nth_set_bit(unsigned long long mask, uint32_t r) {
    const uint32_t lo = (uint32_t)mask, n_lo = (uint32_t)__popc(lo);
    const bool upper = r >= n_lo;
    uint32_t word = upper ? (uint32_t)(mask >> 32) : lo, base = upper ? 32u : 0u;
    r -= upper ? n_lo : 0u;
#pragma unroll
    for (uint32_t shift = 16u; shift != 0u; shift >>= 1) {
        const uint32_t c = (uint32_t)__popc(word & ((1u << shift) - 1u));
        if (r >= c) { word >>= shift; r -= c; base += shift; }
    }
    return base;
}
IRIS_DEV void pool_sphere_test(const SceneDev &sc, uint32_t slot, V3 o, V3 d, float a, exact::Recip ya, bool any, float &cull_t, int &closest) {
    const float4 sp4 = sc.bvh_spheres[slot];
    V3 oc = o - v3(sp4.x, sp4.y, sp4.z);
    float half_b = dot(d, oc);
    float c = len_squared(oc) - sp4.w;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {  // sphere.rs:31-35, :53
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_is_unit(a), ya);
        if (temp > 0.0f) {
            const int index = (int)sc.bvh_prim[slot];
            if (any) { if (temp < cull_t || temp == cull_t) closest = index; }  // bvh_sphere_test_any with tie = none
            else if (temp < cull_t || (temp == cull_t && index < closest)) { cull_t = temp; closest = index; }  // bvh_sphere_test
        }
    }
}
// The owner's part of a query, run with the whole wave busy: the spheres kept out of the tree (tested first, as bvh_closest
// does), then the record.  A visibility query that one of those spheres already answers takes no slot: returns false, and
// `closest` >= 0 says "occluded".
IRIS_DEV bool pool_put(const SceneDev &sc, float *rec, uint32_t slot, const Ray &ray, bool any, float limit, int &closest, unsigned long long &tests) {
    const float a = len_squared(ray.d);
    const exact::Recip ya = exact::recip_for(a);
    float cull_t = limit;
    closest = -1;
    for (uint32_t b = 0; b < sc.bvh_big_count; b++) pool_sphere_test(sc, sc.bvh_big_first + b, ray.o, ray.d, a, ya, any, cull_t, closest);
    tests += sc.bvh_big_count;
    if (any && closest >= 0) return false;
    rec[0u * IRIS_POOL_CAP + slot] = ray.o.x; rec[1u * IRIS_POOL_CAP + slot] = ray.o.y; rec[2u * IRIS_POOL_CAP + slot] = ray.o.z;
    rec[3u * IRIS_POOL_CAP + slot] = ray.d.x; rec[4u * IRIS_POOL_CAP + slot] = ray.d.y; rec[5u * IRIS_POOL_CAP + slot] = ray.d.z;
    rec[6u * IRIS_POOL_CAP + slot] = cull_t;
    rec[7u * IRIS_POOL_CAP + slot] = __int_as_float(closest - IRIS_POOL_OUT);
    return true;
}
// Has the ray of `slot` come back?  (t, primitive) if so.
IRIS_DEV bool pool_get(const float *rec, uint32_t slot, float &t, int &prim) {
    const int v = __float_as_int(rec[7u * IRIS_POOL_CAP + slot]);
    if (v < -1) return false;
    prim = v;
    t = rec[6u * IRIS_POOL_CAP + slot];
    return true;
}
#ifdef IRIS_PHASE_TIMERS  // diagnostic build: passes, rays put, fetch rounds / lanes, inner rounds / lanes, leaf rounds / lanes
__device__ unsigned long long g_pool_stats[8];
#define IRIS_POOL_STAT(k, v) do { ps_[k] += (v); } while (0)
#else
#define IRIS_POOL_STAT(k, v) do { } while (0)
#endif
// One pass over the wave's pool.  pend_e / pend_s (wave-uniform, kept by the caller between passes): slots that hold a ray no
// lane has taken yet; the caller has set the bits of the rays it has just put.  walking (per lane, kept by the caller): this lane
// has a ray parked in `park` (its own IRIS_WALK_WORDS words of global memory, word k at park[k * park_stride]).
IRIS_DEV void bvh_pool_pass(const SceneDev &sc, const SphereStage &st, float *rec, uint32_t *park, size_t park_stride,
                            unsigned long long &pend_e, unsigned long long &pend_s, bool &walking, unsigned long long &tests) {
#ifdef IRIS_PHASE_TIMERS
    unsigned long long ps_[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    IRIS_POOL_STAT(0, 1ull); IRIS_POOL_STAT(1, (unsigned long long)(__popcll(pend_e) + __popcll(pend_s)));
#endif
    bool have = false, any = false;
    uint32_t my = 0u, ref = 0u, sp = 0u, n_tests = 0u;
    V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
    BvhRay br = {v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 0.0f)};
    float a = 1.0f, cull_t = 0.0f;
    exact::Recip ya = {1.0f};
    int closest = -1;
    // ---- resume: a lane that parked a ray at the end of the last pass takes it up again ----
    if (walking) {
        o = v3(__uint_as_float(park[0 * park_stride]), __uint_as_float(park[1 * park_stride]), __uint_as_float(park[2 * park_stride]));
        d = v3(__uint_as_float(park[3 * park_stride]), __uint_as_float(park[4 * park_stride]), __uint_as_float(park[5 * park_stride]));
        cull_t = __uint_as_float(park[6 * park_stride]);
        closest = (int)park[7 * park_stride];
        ref = park[8 * park_stride];
        const uint32_t meta = park[9 * park_stride];
        my = meta & 0xffu; any = (meta & 0x100u) != 0u; sp = meta >> 16;
        for (uint32_t k = 0; k < sp; k++) bvh_push(st, k, park[(10u + k) * park_stride]);
        a = len_squared(d);
        ya = exact::recip_for(a);
        br = bvh_ray(sc, o, bvh_inv_dir(d));
        have = true;
        walking = false;
    }
    uint32_t rounds = 0u;
    for (;;) {
        const unsigned long long m_have = __ballot(have);
        const uint32_t n_have = (uint32_t)__popcll(m_have), n_pe = (uint32_t)__popcll(pend_e), n_pend = n_pe + (uint32_t)__popcll(pend_s);
        // ---- hand pending rays to the lanes that hold none: when enough of them are idle to pay for it, or nothing else is left to do ----
        if (n_pend != 0u && (n_have == 0u || 64u - n_have >= IRIS_POOL_FETCH_MIN)) {
            const uint32_t n_take = min(n_pend, 64u - n_have), r = lanes_below(~m_have);
            IRIS_POOL_STAT(2, 1ull); IRIS_POOL_STAT(3, (unsigned long long)n_take);
            if (!have && r < n_take) {
                const uint32_t k = r < n_pe ? nth_set_bit(pend_e, r) : 64u + nth_set_bit(pend_s, r - n_pe);
                my = k;
                any = k >= 64u;
                o = v3(rec[0u * IRIS_POOL_CAP + k], rec[1u * IRIS_POOL_CAP + k], rec[2u * IRIS_POOL_CAP + k]);
                d = v3(rec[3u * IRIS_POOL_CAP + k], rec[4u * IRIS_POOL_CAP + k], rec[5u * IRIS_POOL_CAP + k]);
                cull_t = rec[6u * IRIS_POOL_CAP + k];
                closest = __float_as_int(rec[7u * IRIS_POOL_CAP + k]) + IRIS_POOL_OUT;
                a = len_squared(d);  // sphere.rs:27
                ya = exact::recip_for(a);
                const V3 inv = bvh_inv_dir(d);
                br = bvh_ray(sc, o, inv);
                const float4 root_lo = sc.bvh_nodes[0], root_hi = sc.bvh_nodes[1];
                ref = bvh_ref(root_lo, root_hi);
                sp = 0u;
                have = box_entry(root_lo, root_hi, o, inv, cull_t) < __builtin_inff();
                if (!have) rec[7u * IRIS_POOL_CAP + k] = __int_as_float(closest);  // misses the tree: the big spheres' answer stands
            }
            // the first n_take pending slots are out: extension rays first, then queries, each in lane order
            const uint32_t take_e = min(n_take, n_pe), take_s = n_take - take_e, me = lanes_below(~0ull);
            const bool mine_e = ((pend_e >> me) & 1ull) != 0ull && (uint32_t)__popcll(pend_e & ((1ull << me) - 1ull)) < take_e;
            const bool mine_s = ((pend_s >> me) & 1ull) != 0ull && (uint32_t)__popcll(pend_s & ((1ull << me) - 1ull)) < take_s;
            pend_e &= ~__ballot(mine_e);
            pend_s &= ~__ballot(mine_s);
            continue;
        }
        if (n_have == 0u) break;  // nothing pending, nothing walking
        if (n_pend == 0u && n_have < IRIS_POOL_SUSPEND && rounds >= IRIS_POOL_MIN_ROUNDS) break;  // park the stragglers
        rounds++;
        // ---- one kind of step per round, the kind more lanes are waiting for: inner nodes or leaves ----
        const bool at_inner = have && (ref >> 29) == 0u;
        const uint32_t n_inner = (uint32_t)__popcll(__ballot(at_inner));
        if (100u * n_inner >= IRIS_POOL_INNER_PCT * n_have) {
            IRIS_POOL_STAT(4, 1ull); IRIS_POOL_STAT(5, (unsigned long long)n_inner);
#pragma unroll
            for (int rep = 0; rep < IRIS_POOL_INNER_REPS; rep++)
            if (have && (ref >> 29) == 0u) {
                const uint32_t first = ref & 0x1fffffffu;
                const uint4 cl = sc.bvh_q[first - 1u], cr = sc.bvh_q[first];
                const float tl = box_entry_q(cl, br, cull_t), tr = box_entry_q(cr, br, cull_t);
                const bool hl = tl < __builtin_inff(), hr = tr < __builtin_inff();
                if (hl || hr) {
                    const bool left_first = hl && (!hr || tl <= tr);
                    const uint32_t ref_l = cl.w, ref_r = cr.w;
                    if (hl && hr) {
                        bvh_push(st, sp, left_first ? ref_r : ref_l);
                        sp++;
                    }
                    ref = left_first ? ref_l : ref_r;
                } else if (sp != 0u) {
                    sp--;
                    ref = bvh_pop(st, sp);
                } else {
                    have = false;
                }
            }
        } else if (have && !at_inner) {
            IRIS_POOL_STAT(6, 1ull); IRIS_POOL_STAT(7, (unsigned long long)__popcll(__ballot(true)));
            const uint32_t first = ref & 0x1fffffffu, count = ref >> 29;
            for (uint32_t j = 0; j < count; j++) pool_sphere_test(sc, first + j, o, d, a, ya, any, cull_t, closest);
            n_tests += count;
            if (any && closest >= 0) {
                have = false;
            } else if (sp != 0u) {
                sp--;
                ref = bvh_pop(st, sp);
            } else {
                have = false;
            }
        }
        if (!have && ((m_have >> lanes_below(~0ull)) & 1ull) != 0ull) {  // this lane held a ray at the top of the round and is done with it: result to its slot
            rec[6u * IRIS_POOL_CAP + my] = cull_t;
            rec[7u * IRIS_POOL_CAP + my] = __int_as_float(closest);
        }
    }
    tests += n_tests;
    // ---- park what is still walking ----
    if (have) {
        park[0 * park_stride] = __float_as_uint(o.x); park[1 * park_stride] = __float_as_uint(o.y); park[2 * park_stride] = __float_as_uint(o.z);
        park[3 * park_stride] = __float_as_uint(d.x); park[4 * park_stride] = __float_as_uint(d.y); park[5 * park_stride] = __float_as_uint(d.z);
        park[6 * park_stride] = __float_as_uint(cull_t);
        park[7 * park_stride] = (uint32_t)closest;
        park[8 * park_stride] = ref;
        park[9 * park_stride] = my | (any ? 0x100u : 0u) | (sp << 16);
        for (uint32_t k = 0; k < sp; k++) park[(10u + k) * park_stride] = bvh_pop(st, k);
        walking = true;
    }
    __builtin_amdgcn_wave_barrier();
#ifdef IRIS_PHASE_TIMERS
    if (lanes_below(~0ull) == 0u)
        for (int k = 0; k < 8; k++) atomicAdd(&g_pool_stats[k], ps_[k]);
#endif
}

// ---- the same queries as a kernel of their own: rays in a queue in HBM, results in an array beside it (k_wfq_trace) ----
// The traversal is bound by the latency of dependent L2 reads and by lanes that leave the tree at different times; both want many
// more rays in flight than a kernel that also shades can hold (168 registers, three waves per SIMD).  So the queue kernels give it
// a kernel of its own: nothing but this loop, small enough for eight waves per SIMD, every lane taking the next ray of the queue
// when it has finished one.  A ray is two float4: (o, cull_t) and (d, word); word = WFQ_DONE: already answered by the producer (it
// tests the spheres outside the tree and the root box with its whole wave busy: wfq_prepare); else bit 31 = any-hit query, low bits
// = the tie primitive (any-hit) or 1 + the closest primitive so far (closest-hit).  Per ray the steps are bvh_closest's, one for
// one: same boxes entered, same spheres tested in the same order, so results and test counts are those of every other kernel.
#define IRIS_WFQ_DONE 0x7fffffffu
#define IRIS_WFQ_ANY 0x80000000u
#ifndef IRIS_WFQ_FETCH_MIN
#define IRIS_WFQ_FETCH_MIN 12u  // idle lanes that make a fetch worth its instructions
#endif
#ifndef IRIS_WFQ_INNER_PCT
#define IRIS_WFQ_INNER_PCT 50u
#endif
#ifndef IRIS_WFQ_INNER_REPS
#define IRIS_WFQ_INNER_REPS 3
#endif
IRIS_DEV void wfq_sphere_test(const SceneDev &sc, uint32_t slot, V3 o, V3 d, float a, exact::Recip ya, bool any, int tie, float &cull_t, int &closest) {
    const float4 sp4 = sc.bvh_spheres[slot];
    V3 oc = o - v3(sp4.x, sp4.y, sp4.z);
    float half_b = dot(d, oc);
    float c = len_squared(oc) - sp4.w;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {  // sphere.rs:31-35, :53
        float temp = first_positive_root(half_b, sqrt_x(discrim), a, a_is_unit(a), ya);
        if (temp > 0.0f) {
            const int index = (int)sc.bvh_prim[slot];
            if (any) { if (temp < cull_t || (temp == cull_t && index < tie)) closest = index; }  // bvh_sphere_test_any
            else if (temp < cull_t || (temp == cull_t && index < closest)) { cull_t = temp; closest = index; }  // bvh_sphere_test
        }
    }
}
// The producer's part of a ray: the spheres kept out of the tree (tested first, as bvh_closest does) and the root box.  Returns the
// two float4 of the record; `answered`: the queue kernel has nothing to do (an any-hit query that one of those spheres answers, or a
// ray that misses the tree), (cull_t, closest) is the result.
IRIS_DEV void wfq_prepare(const SceneDev &sc, const Ray &ray, bool any, float limit, int tie, float4 &rec_o, float4 &rec_d, bool &answered,
                          float &cull_t, int &closest, unsigned long long &tests) {
    const float a = len_squared(ray.d);
    const exact::Recip ya = exact::recip_for(a);
    cull_t = limit;
    closest = -1;
    for (uint32_t b = 0; b < sc.bvh_big_count; b++) wfq_sphere_test(sc, sc.bvh_big_first + b, ray.o, ray.d, a, ya, any, tie, cull_t, closest);
    tests += sc.bvh_big_count;
    answered = any && closest >= 0;
    if (!answered) {
        const float4 root_lo = sc.bvh_nodes[0], root_hi = sc.bvh_nodes[1];
        answered = !(box_entry(root_lo, root_hi, ray.o, bvh_inv_dir(ray.d), cull_t) < __builtin_inff());
    }
    rec_o = make_float4(ray.o.x, ray.o.y, ray.o.z, cull_t);
    const uint32_t word = answered ? IRIS_WFQ_DONE : (any ? (IRIS_WFQ_ANY | ((uint32_t)tie & 0x7fffffffu)) : (uint32_t)(closest + 1));
    rec_d = make_float4(ray.d.x, ray.d.y, ray.d.z, __uint_as_float(word));
}
// One wave's share of the ray queue.  The queue is cut into REGIONS of IRIS_WFQ_REGION positions, one per tile of 256 path slots
// (a slot sends at most three rays per pass), filled from their start: region r holds n_region[r] rays (so the producer needs no
// atomic to append).  A wave claims one region at a time from `cursor` and hands its rays to its lanes as they fall idle.
// (Tried and dropped: two lanes per ray, each fetching and testing one child and swapping results through DPP -- bit-exact, but
// 124 ms against 109 ms for the trace kernels of config D: the vector L1 serves about one 16-byte lane request per cycle whichever
// lane asks, so the requests per ray, not the lanes, are what has to shrink -- see the compressed nodes, box_entry_q.
// Also tried and dropped, with the compressed nodes: TWO RAYS PER LANE -- a lane whose working ray is at the other kind of node than
// the round's exchanges it with its second ray (twelve v_swap_b32; origin and direction re-read from the queue at the leaves), so that
// a lane idles only when neither ray can take part: bit-exact, 116.9 ms (72 VGPR, 7 waves) / 119.3 ms (80 VGPR, 6 waves) against
// 113.3 ms.  The lanes were not the limit: with the texture-data path 83 % and the VALU 60-70 % busy, more lanes per step only
// queue up behind the same L1.)
#define IRIS_WFQ_REGION 768u
#ifdef IRIS_WFQ_STATS  // diagnostic build: rounds and the lanes busy in them -- fetches, rays fetched, inner rounds, lanes at each inner step, leaf rounds, lanes, all rounds, lanes holding a ray
__device__ unsigned long long g_wfq_stats[8];
#define IRIS_WFQ_STAT(k, v) do { ws_[k] += (v); } while (0)
#else
#define IRIS_WFQ_STAT(k, v) do { } while (0)
#endif
IRIS_DEV void wfq_trace_wave(const SceneDev &sc, const SphereStage &st, const float4 *__restrict__ rq_o, const float4 *__restrict__ rq_d, uint2 *__restrict__ res,
                             const uint32_t *__restrict__ n_region, uint32_t n_regions, uint32_t *cursor, unsigned long long &tests) {
    bool have = false, any = false, more = true;
    uint32_t my = 0u, ref = 0u, sp = 0u, n_tests = 0u, w_next = 0u, w_lim = 0u;
    V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
    BvhRay br = {v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 0.0f)};
    float a = 1.0f, cull_t = 0.0f;
    exact::Recip ya = {1.0f};
    int closest = -1, tie = 0;
    const float4 root_lo = sc.bvh_nodes[0], root_hi = sc.bvh_nodes[1];
    const uint32_t root_ref = bvh_ref(root_lo, root_hi);
#ifdef IRIS_WFQ_STATS
    unsigned long long ws_[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    for (;;) {
        unsigned long long m_have = __ballot(have);
        uint32_t n_have = (uint32_t)__popcll(m_have);
        // ---- fetch: lanes without a ray take the next rays of the wave's chunk (a new chunk when that one is used up) ----
        if (more && (n_have == 0u || 64u - n_have >= IRIS_WFQ_FETCH_MIN)) {
            if (w_next >= w_lim) {
                uint32_t r = 0u;
                if (lanes_below(~0ull) == 0u) r = atomicAdd(cursor, 1u);
                r = __builtin_amdgcn_readfirstlane(r);
                if (r >= n_regions) more = false;
                else { w_next = r * IRIS_WFQ_REGION; w_lim = w_next + n_region[r]; }
            }
            if (more && w_next < w_lim) {
                const uint32_t n_take = min(64u - n_have, w_lim - w_next), r = lanes_below(~m_have);
                IRIS_WFQ_STAT(0, 1ull); IRIS_WFQ_STAT(1, (unsigned long long)n_take);
                if (!have && r < n_take) {
                    my = w_next + r;
                    const float4 ro = rq_o[my], rd = rq_d[my];
                    const uint32_t word = __float_as_uint(rd.w);
                    if (word != IRIS_WFQ_DONE) {
                        o = v3(ro.x, ro.y, ro.z);
                        d = v3(rd.x, rd.y, rd.z);
                        cull_t = ro.w;
                        any = (word & IRIS_WFQ_ANY) != 0u;
                        tie = any ? (int)(word & 0x7fffffffu) : 0;
                        closest = any ? -1 : (int)word - 1;
                        a = len_squared(d);  // sphere.rs:27
                        ya = exact::recip_for(a);
                        br = bvh_ray(sc, o, bvh_inv_dir(d));
                        ref = root_ref;
                        sp = 0u;
                        have = true;
                    }
                }
                w_next += n_take;
                m_have = __ballot(have);
                n_have = (uint32_t)__popcll(m_have);
            }
        }
        if (n_have == 0u) {
            if (!more) break;
            continue;
        }
        // ---- one kind of step per round, the kind more lanes are waiting for: inner nodes or leaves ----
        const bool at_inner = have && (ref >> 29) == 0u;
        const uint32_t n_inner = (uint32_t)__popcll(__ballot(at_inner));
        IRIS_WFQ_STAT(6, 1ull); IRIS_WFQ_STAT(7, (unsigned long long)n_have);
        if (100u * n_inner >= IRIS_WFQ_INNER_PCT * n_have) {
            IRIS_WFQ_STAT(2, 1ull);
#pragma unroll
            for (int rep = 0; rep < IRIS_WFQ_INNER_REPS; rep++)
            if (have && (ref >> 29) == 0u) {
                IRIS_WFQ_STAT(3, (unsigned long long)__popcll(__ballot(true)));
                const uint32_t first = ref & 0x1fffffffu;
                const uint4 cl = sc.bvh_q[first - 1u], cr = sc.bvh_q[first];
                const float tl = box_entry_q(cl, br, cull_t), tr = box_entry_q(cr, br, cull_t);
                const bool hl = tl < __builtin_inff(), hr = tr < __builtin_inff();
                if (hl || hr) {
                    const bool left_first = hl && (!hr || tl <= tr);
                    const uint32_t ref_l = cl.w, ref_r = cr.w;
                    if (hl && hr) {
                        bvh_push(st, sp, left_first ? ref_r : ref_l);
                        sp++;
                    }
                    ref = left_first ? ref_l : ref_r;
                } else if (sp != 0u) {
                    sp--;
                    ref = bvh_pop(st, sp);
                } else {
                    have = false;
                }
            }
        } else if (have && !at_inner) {
            IRIS_WFQ_STAT(4, 1ull); IRIS_WFQ_STAT(5, (unsigned long long)__popcll(__ballot(true)));
            const uint32_t first = ref & 0x1fffffffu, count = ref >> 29;
            for (uint32_t j = 0; j < count; j++) wfq_sphere_test(sc, first + j, o, d, a, ya, any, tie, cull_t, closest);
            n_tests += count;
            if (any && closest >= 0) {
                have = false;
            } else if (sp != 0u) {
                sp--;
                ref = bvh_pop(st, sp);
            } else {
                have = false;
            }
        }
        if (!have && ((m_have >> lanes_below(~0ull)) & 1ull) != 0ull)  // this lane held a ray at the top of the round and is done with it
            res[my] = make_uint2(__float_as_uint(cull_t), (uint32_t)closest);
    }
    tests += n_tests;
#ifdef IRIS_WFQ_STATS
    if (lanes_below(~0ull) == 0u)
        for (int k = 0; k < 8; k++) atomicAdd(&g_wfq_stats[k], ws_[k]);
#endif
}

// Scene::intersection / ray_hits_point / ray_hits_object all reduce to this scan (scene.rs:116-166):
// closest t > 0 over all primitives in index order, strict '<' so the lowest index wins ties.
// Must be called by every thread of the workgroup (inactive threads pass active=false) because a
// list larger than the LDS window is staged cooperatively.
// `tests` counts the Sphere::intersect evaluations made for active lanes (list scans: n_prims each).
IRIS_DEV void scan_closest(const SceneDev &sc, const SphereStage &st, bool active, const Ray &ray, float &t_out, int &prim_out, unsigned long long &tests) {
    float closest_t = __builtin_inff();
    int closest = -1;
    float a = len_squared(ray.d);  // sphere.rs:27 (loop-invariant)
    if (st.bvh) {
        if (active) bvh_closest<false>(sc, st, ray, a, closest_t, closest, tests);
    } else if (st.resident) {
        if (active) scan_window<4>(st.lds, sc.n_prims, 0u, ray, a, closest_t, closest);
        if (active) tests += sc.n_prims;
    } else {
        if (active) tests += sc.n_prims;
        for (uint32_t base = 0; base < sc.n_prims; base += st.cap) {
            uint32_t n = sc.n_prims - base < st.cap ? sc.n_prims - base : st.cap;
            uint32_t n_pad = (n + IRIS_SCAN_PAD - 1u) & ~(IRIS_SCAN_PAD - 1u);
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) st.lds[i] = i < n ? stage_sphere(sc.spheres[base + i]) : stage_filler();
            __syncthreads();
            if (active) scan_window<8>(st.lds, n, base, ray, a, closest_t, closest);
        }
    }
    t_out = closest_t;
    prim_out = closest;
}

IRIS_RADIOMETRIC_BEGIN
// ---------------------------------------------------------------- sphere light (shape/sphere.rs:77-162)
IRIS_DEV void sphere_sample(float4 sp, const Hit &hit, DrawRing ring, Sampler &smp, V3 &p_out, float &pdf_out, bool pin = true, bool *off_sphere = nullptr) {
    float u0 = gen_0_1(ring, smp);
    float u1 = gen_0_1(ring, smp);
    V3 position = v3(sp.x, sp.y, sp.z);
    float radius = sp.w;
    V3 point = offset_origin_signed(hit.point, hit.normal, hit.back_face);
    if (distance_squared(position, point) <= radius * radius) {
        V3 light_normal = unit_sphere(u0, u1, pin);
        V3 light_point = position + radius * light_normal;
        float area = 4.0f * PI_F * (radius * radius);
        float pdf = fdiv(distance_squared(light_point, point), dot(light_normal, normalize(light_point - point)) * area);
        p_out = light_point;
        pdf_out = rmax(pdf, 0.001f);
        return;
    }
    float dc = distance(position, hit.point);
    V3 wc = normalize(position - hit.point);
    V3 wc_x, wc_y;
    coordinate_system_from_unit(wc, wc_x, wc_y);
    float sin_theta_max = fdiv(radius, dc);
    float sin_theta_max_2 = sin_theta_max * sin_theta_max;
    float cos_theta_max = sqrt_nz(rmax(1.0f - sin_theta_max_2, 0.0f));
    float sin_theta_2, cos_theta_;
    if (sin_theta_max_2 < 0.00068523f) {
        sin_theta_2 = sin_theta_max_2 * u0;
        cos_theta_ = sqrt_nz(rmax(1.0f - sin_theta_2, 0.0f));
    } else {
        cos_theta_ = (cos_theta_max - 1.0f) * u0 + 1.0f;
        sin_theta_2 = 1.0f - cos_theta_ * cos_theta_;
    }
    float cos_alpha = fdiv(sin_theta_2, sin_theta_max) + cos_theta_ * sqrt_nz(rmax(1.0f - fdiv(sin_theta_2, sin_theta_max_2), 0.0f));
    float sin_alpha = sqrt_nz(rmax(1.0f - cos_alpha * cos_alpha, 0.0f));
    float phi = u1 * 2.0f * PI_F;
    float sphi, cphi;
    iris_sincos(phi, sphi, cphi, pin);
    // vec3.rs:87-96 with (x, y, z) = (-wc_x, -wc_y, -wc)
    V3 normal = ((sin_alpha * cphi) * (-wc_x) + (sin_alpha * sphi) * (-wc_y)) + cos_alpha * (-wc);
    V3 sampled_point_local = radius * normalize(normal);
    p_out = position + sampled_point_local;
#if IRIS_DEBUG_INVARIANTS
    if (off_sphere) {  // sphere.rs:131-132
        const float dist = __builtin_sqrtf(distance_squared(p_out, position));
        *off_sphere = !(dist >= radius * 0.98f) || !(dist <= radius * 1.02f);
    }
#endif
    pdf_out = pdf_cone(cos_theta_max);
}
IRIS_DEV float sphere_pdf(float4 sp, const Hit &hit, V3 wi, unsigned long long &sphere_tests) {
    V3 position = v3(sp.x, sp.y, sp.z);
    float radius = sp.w;
    V3 point = offset_origin_signed(hit.point, hit.normal, hit.back_face);
    if (distance_squared(position, point) <= radius * radius) {
        Ray ray = ray_spawn(hit.point, wi, hit.normal);
        sphere_tests++;
        float t = sphere_hit_t(stage_sphere(sp), ray, len_squared(ray.d));
        if (t > 0.0f) {
            V3 lp = ray.o + ray.d * t;
            V3 ln = div3(lp - position, radius);
            float area = 4.0f * PI_F * (radius * radius);
            float pdf = fdiv(distance_squared(lp, point), dot(ln, normalize(lp - point)) * area);
            return rmax(pdf, 0.001f);
        }
        return 0.0f;
    }
    float sin_theta_max_2 = fdiv(radius * radius, distance_squared(position, hit.point));
    float cos_theta_max = sqrt_nz(rmax(1.0f - sin_theta_max_2, 0.0f));
    return pdf_cone(cos_theta_max);
}

// ---------------------------------------------------------------- GGX (sampling/ggx.rs)
IRIS_DEV V3 ggx_sample(V3 wo_in, float ax, float ay, DrawRing ring, Sampler &smp) {  // ggx.rs:7-33
    V3 wo = normalize(v3(ax * wo_in.x, ay * wo_in.y, wo_in.z));
    float len_2 = wo.x * wo.x + wo.y * wo.y;
    V3 t1_v = len_2 > 0.0f ? div3(v3(-wo.y, wo.x, 0.0f), rsqrt_exact(len_2)) : v3(1.0f, 0.0f, 0.0f);
    V3 t2_v = cross(wo, t1_v);
    float r = sqrt_nz(gen_0_1(ring, smp));  // a draw is 0 or at least 2^-32
    float phi = 2.0f * PI_F * gen_0_1(ring, smp);
    float sphi, cphi;
    iris_sincos(phi, sphi, cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + wo.z);
    t2 = (1.0f - s) * sqrt_nz(1.0f - t1 * t1) + s * t2;  // |t1| = r |cos| <= 1: the argument is 0 or at least 2^-24
    V3 nh = (t1 * t1_v + t2 * t2_v) + wo * sqrt_nz(rmax(1.0f - t1 * t1 - t2 * t2, 0.0f));
    return normalize(v3(ax * nh.x, ay * nh.y, rmax(nh.z, 0.0f)));
}
IRIS_DEV float ggx_evaluate(V3 wh, float ax, float ay) {  // ggx.rs:35-43
    float tan2 = tan_2_theta(wh);
    if (__builtin_isinf(tan2)) return 0.0f;
    float e = (fdiv(cos_2_phi(wh), ax * ax) + fdiv(sin_2_phi(wh), ay * ay)) * tan2;
    float ct = cos_theta(wh);
    float ct2 = ct * ct;
    float ope = 1.0f + e;
    return rcp_g(PI_F * ax * ay * (ct2 * ct2) * (ope * ope));
}
IRIS_DEV float ggx_lambda(V3 w, float ax, float ay) {  // ggx.rs:45-55
    float tt = tan_theta(w);
    if (__builtin_isinf(tt)) return 0.0f;
    float alpha = rsqrt_exact(cos_2_phi(w) * (ax * ax) + sin_2_phi(w) * (ay * ay));
    float a = alpha * fabs_(tt);
    return (-1.0f + rsqrt_exact(1.0f + a * a)) / 2.0f;
}
IRIS_DEV float ggx_g(V3 wo, V3 wh, float ax, float ay) { return rcp_g(1.0f + ggx_lambda(wh, ax, ay) + ggx_lambda(wo, ax, ay)); }
IRIS_DEV float ggx_g1(V3 w, float ax, float ay) { return rcp_g(1.0f + ggx_lambda(w, ax, ay)); }
IRIS_DEV float ggx_pdf(V3 wo, V3 wh, float ax, float ay) {  // ggx.rs:61-64
    return fdiv(ggx_evaluate(wh, ax, ay) * ggx_g1(wo, ax, ay) * fabs_(dot(wo, wh)), fabs_(cos_theta(wo)));
}

// ---------------------------------------------------------------- BSDFs (bsdf/*.rs)
IRIS_DEV V3 reflect(V3 wo, V3 n) { return (-wo) + (2.0f * dot(wo, n)) * n; }  // microfacet.rs:91-93

// lambertian.rs:24-31, :43: albedo.evaluate(lambda) / PI.  For a constant spectrum the four quotients are one
// value that does not depend on the path: iris_scene_upload forms it once (the same correctly rounded f32
// division), so the three evaluations of a bounce cost a load instead of IEEE divisions.
IRIS_DEV F4 lambertian_value(const SceneDev &sc, const iris_bsdf &b, F4 wl, bool mono) {
    if (mono) return splat(b.p0);  // the Lambertian/constant class: shade_bounce has put albedo / PI there (from prim_info: no dependent loads)
    if (sc.spectra[b.spec_a].kind == IRIS_SPEC_CONSTANT) return splat(sc.spec_over_pi[b.spec_a]);
    const F4 v = spectrum_evaluate(sc.spectra[b.spec_a], wl, mono);
    return f4(div_pi(v.v[0]), div_pi(v.v[1]), div_pi(v.v[2]), div_pi(v.v[3]));
}
IRIS_DEV F4 microfacet_evaluate(const SceneDev &sc, const iris_bsdf &b, V3 wi, V3 wo, F4 wl, bool mono) {  // microfacet.rs:33-57
    float cos_theta_o = fabs_(cos_theta(wo));
    float cos_theta_i = fabs_(cos_theta(wi));
    V3 wh = wo + wi;
    if (!same_hemisphere(wi, wo) || cos_theta_o == 0.0f || cos_theta_i == 0.0f || (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f))
        return splat(0.0f);
    wh = normalize(wh);
    V3 wh_facing = face_forward(wh, v3(0.0f, 0.0f, 1.0f));
    float d = ggx_evaluate(wh, b.p0, b.p1);
    float f = fresnel_dielectric(dot(wi, wh_facing), 1.5f, 1.0f);
    float g = ggx_g(wo, wh, b.p0, b.p1);
    return div4(((spectrum_evaluate(sc.spectra[b.spec_a], wl, mono) * d) * g) * f, 4.0f * cos_theta_o * cos_theta_i, mono);
}
IRIS_DEV F4 bsdf_evaluate(const SceneDev &sc, const iris_bsdf &b, V3 wi, V3 wo, F4 wl, bool mono) {
    if (b.kind == IRIS_BSDF_LAMBERTIAN) return lambertian_value(sc, b, wl, mono);  // lambertian.rs:24-31
    if (b.kind == IRIS_BSDF_MICROFACET) return microfacet_evaluate(sc, b, wi, wo, wl, mono);
    return splat(0.0f);  // specular.rs:26-33, fresnel.rs:37-44
}
IRIS_DEV F4 bsdf_pdf(const iris_bsdf &b, V3 wi, V3 wo) {
    if (b.kind == IRIS_BSDF_LAMBERTIAN) return splat(div_pi(fabs_(cos_theta(wi))));  // lambertian.rs:33-35, mod.rs:56-59
    if (b.kind == IRIS_BSDF_MICROFACET) {                                            // microfacet.rs:59-63
        V3 wh = normalize(wi + wo);
        return splat(fdiv(ggx_pdf(wo, wh, b.p0, b.p1), 4.0f * dot(wo, wh)));
    }
    return splat(0.0f);  // specular.rs:35-37, fresnel.rs:46-48
}
IRIS_DEV void bsdf_sample(const SceneDev &sc, const iris_bsdf &b, V3 wo, F4 wl, DrawRing ring, Sampler &smp, bool mono, V3 &wi_out, F4 &val_out, F4 &pdf_out, bool pin = true) {
    wi_out = v3(0.0f, 0.0f, 0.0f);
    val_out = splat(0.0f);
    pdf_out = splat(0.0f);
    if (b.kind == IRIS_BSDF_LAMBERTIAN) {  // lambertian.rs:37-50
        float r1 = gen_0_1(ring, smp);
        float r2 = gen_0_1(ring, smp);
        V3 wi = cosine_unit_hemisphere(r1, r2, pin);
        wi = same_hemisphere(wo, wi) ? wi : -wi;
        wi_out = wi;
        val_out = lambertian_value(sc, b, wl, mono);
        pdf_out = splat(div_pi(fabs_(cos_theta(wi))));
    } else if (b.kind == IRIS_BSDF_MICROFACET) {  // microfacet.rs:65-88
        V3 wh = ggx_sample(wo, b.p0, b.p1, ring, smp);
        V3 wi = reflect(wo, wh);
        if (!(cos_theta(wo) == 0.0f || dot(wo, wh) < 0.0f || !same_hemisphere(wo, wi))) {
            float pdf = fdiv(ggx_pdf(wo, wh, b.p0, b.p1), 4.0f * dot(wo, wh));
            wi_out = wi;
            val_out = microfacet_evaluate(sc, b, wi, wo, wl, mono);
            pdf_out = splat(pdf);
        }
    } else if (b.kind == IRIS_BSDF_SPECULAR) {  // specular.rs:39-49
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        wi_out = wi;
        val_out = div4(spectrum_evaluate(sc.spectra[b.spec_a], wl, mono), fabs_(cos_theta(wi)), mono);
        pdf_out = splat(1.0f);
    } else if (b.kind == IRIS_BSDF_FRESNEL) {  // fresnel.rs:30-34, 50-102
        const float eta_a = 1.0f;
        F4 eta_b = b.p0 + (b.p1 / ((wl * wl) * 1e-6f));
        F4 fr;
#pragma unroll
        for (int k = 0; k < 4; k++) fr.v[k] = fresnel_dielectric(cos_theta(wo), eta_a, eta_b.v[k]);
        if (gen_0_1(ring, smp) < hero(fr)) {
            V3 wi = v3(-wo.x, -wo.y, wo.z);
            F4 bs = div4(spectrum_evaluate(sc.spectra[b.spec_a], wl, mono), fabs_(cos_theta(wi)), mono);
            wi_out = wi;
            val_out = bs * fr;
            pdf_out = fr;
        } else {
            float eta_i, eta_t;
            if (cos_theta(wo) > 0.0f) { eta_i = eta_a; eta_t = hero(eta_b); }
            else { eta_i = hero(eta_b); eta_t = eta_a; }
            V3 wi;
            if (refract(wo, face_forward(v3(0.0f, 0.0f, 1.0f), wo), fdiv(eta_i, eta_t), wi)) {
                F4 ft = div4((spectrum_evaluate(sc.spectra[b.spec_b], wl, mono) * (1.0f - hero(fr))) * (eta_i * eta_i), eta_t * eta_t, mono);
                float hero_value = fdiv(hero(ft), fabs_(cos_theta(wi)));
                wi_out = wi;
                val_out = f4(hero_value, 0.0f, 0.0f, 0.0f);
                pdf_out = f4(1.0f - hero(fr), 0.0f, 0.0f, 0.0f);
            }
        }
    }
    // IRIS_BSDF_NULL (null.rs: unreachable!()): absorbs
}

// ---------------------------------------------------------------- environment map (docs/SPEC.md section 6)
// The reference stubs it (Scene._env_map scene.rs:27, background_emission scene.rs:112-114, the clamp note of
// upsample.rs:53-54); layout, look-up and sampling are this build's definition.  Octahedral square, y = pole
// axis: only + - * / sqrt and compares, so GPU and CPU agree to the bit without an inverse-trigonometry spec.
IRIS_DEV float env_sgn(float a) { return a >= 0.0f ? 1.0f : -1.0f; }
IRIS_DEV V3 oct_point(float u, float v) {
    float y = (1.0f - fabs_(u)) - fabs_(v);
    if (y >= 0.0f) return v3(u, y, v);
    return v3((1.0f - fabs_(v)) * env_sgn(u), y, (1.0f - fabs_(u)) * env_sgn(v));
}
IRIS_DEV uint32_t env_texel_coord(float t, uint32_t n) {
    uint32_t i = f_as_index((t * 0.5f + 0.5f) * (float)n);
    return i < n ? i : n - 1u;
}
// Emission arriving along direction d (the four wavelengths) and the solid-angle density with which
// env_sample produces d.
IRIS_DEV void env_eval(const SceneDev &sc, V3 d, F4 wl, F4 &le, float &pdf) {
    float s = (fabs_(d.x) + fabs_(d.y)) + fabs_(d.z);
    V3 q = div3(d, s);
    float u, v;
    if (q.y >= 0.0f) { u = q.x; v = q.z; }
    else { u = (1.0f - fabs_(q.z)) * env_sgn(q.x); v = (1.0f - fabs_(q.x)) * env_sgn(q.z); }
    float l2 = len_squared(q);
    float jac = l2 * rsqrt_exact(l2);  // |q|^3: d(omega) = du dv / |q|^3
    uint32_t c = env_texel_coord(u, sc.env_w), r = env_texel_coord(v, sc.env_h);
    uint32_t idx = r * sc.env_w + c;
    float4 t = sc.env_texels[idx];
    iris_spectrum sp = {IRIS_SPEC_UPSAMPLED_HDR, t.x, t.y, t.z, t.w};
    le = spectrum_evaluate(sp, wl, false);
    pdf = sc.env_integral > 0.0f ? (fdiv(sc.env_func[idx], sc.env_integral) * 0.25f) * jac : 0.0f;
}
// Piecewise-constant 1-D distribution: the largest idx with cdf[idx] <= u ("CDF search"), then the continuous
// position inside that bin, in [0, 1].  docs/SPEC.md states the search as a bisection; on the GPU a bisection is
// ceil(log2 n) DEPENDENT loads per lane (20 L2 round trips for a 1024^2 map, with three waves per SIMD to hide
// them), so the same index is found 32 entries at a time instead: because the CDF is non-decreasing, the index is
// the number of entries 1 .. n-1 that are <= u, counted block by block from the coarsest level down -- two rounds of
// eight independent 16-byte loads for n = 1024.  Same index, hence the same sample, bit for bit (tested against the
// oracle's bisection).
#define IRIS_ENV_BLOCK 32u
#define IRIS_ENV_PAD 2.0f  // pads blocks: above every draw (u < 1)
IRIS_DEV float env_sample_1d(const float *const lvl[3], const uint32_t stride[3], uint32_t n_levels, uint32_t row, uint32_t n, float u, uint32_t &idx) {
    uint32_t b = 0u;  // the index found at the level above = the block to search at this level
    for (uint32_t l = n_levels; l-- > 0u;) {
        const float4 *p = (const float4 *)(lvl[l] + (size_t)row * stride[l] + (size_t)b * IRIS_ENV_BLOCK);
        float4 v[IRIS_ENV_BLOCK / 4u];
#pragma unroll
        for (uint32_t k = 0; k < IRIS_ENV_BLOCK / 4u; k++) v[k] = p[k];
        uint32_t count = 0u;  // entry 0 of the block is <= u by the level above (cdf[0] = 0 at the top)
#pragma unroll
        for (uint32_t k = 0; k < IRIS_ENV_BLOCK / 4u; k++) {
            if (k != 0u) count += v[k].x <= u ? 1u : 0u;
            count += v[k].y <= u ? 1u : 0u;
            count += v[k].z <= u ? 1u : 0u;
            count += v[k].w <= u ? 1u : 0u;
        }
        b = b * IRIS_ENV_BLOCK + count;
    }
    const float *cdf = lvl[0] + (size_t)row * stride[0];
    float c0 = cdf[b];
    float c1 = b + 1u < n ? cdf[b + 1u] : 1.0f;  // cdf[n] = 1 exactly (iris_envmap_tables)
    float den = c1 - c0;
    float du = u - c0;
    if (den > 0.0f) du = fdiv(du, den);
    idx = b;
    return fdiv((float)b + du, (float)n);
}
IRIS_DEV V3 env_sample(const SceneDev &sc, float u0, float u1) {
    uint32_t r, c;
    const uint32_t marg_stride[3] = {0u, 0u, 0u};
    float vv = env_sample_1d(sc.env_marg_lvl, marg_stride, sc.env_marg_levels, 0u, sc.env_h, u0, r);
    float uu = env_sample_1d(sc.env_cond_lvl, sc.env_cond_stride, sc.env_cond_levels, r, sc.env_w, u1, c);
    return normalize(oct_point(2.0f * uu - 1.0f, 2.0f * vv - 1.0f));
}

IRIS_RADIOMETRIC_END
// ---------------------------------------------------------------- path state and the two stages' per-lane work
// -DIRIS_DEBUG_INVARIANTS=1 (libiris_hip_debug.so): the debug_assert!s of the reference's hot path, counted instead of trapping, at a
// fixed list of check points per shaded bounce -- the list the CPU checker of the tests counts too, so the totals are comparable:
//   inv[0] Vec3 not finite (vec3.rs:24-31): the hit's frame; the next ray           inv[1] SpectralSample negative or NaN (sample.rs:64-73):
//   radiance after the bounce's light, throughput after its update   inv[2] PdfSet negative or NaN (pdf.rs:49-58): the sampled
//   continuation's pdfs, the path pdfs after the update (HWSS)       inv[3] cone-sampled light point off its sphere by > 2 % (sphere.rs:131-132)
#ifndef IRIS_DEBUG_INVARIANTS
#define IRIS_DEBUG_INVARIANTS 0
#endif
struct Counters {
    uint32_t samples, bounces, rays;
    unsigned long long sphere_tests;
#if IRIS_DEBUG_INVARIANTS
    uint32_t inv[4];
#endif
};
#if IRIS_DEBUG_INVARIANTS
#define IRIS_INV(cnt, k, cond) do { if (cond) (cnt).inv[k]++; } while (0)
#else
#define IRIS_INV(cnt, k, cond) do { } while (0)
#endif

IRIS_DEV bool v3_bad(V3 a) { return !(__builtin_isfinite(a.x) && __builtin_isfinite(a.y) && __builtin_isfinite(a.z)); }
IRIS_DEV bool f4_bad(F4 a) { return !(a.v[0] >= 0.0f && a.v[1] >= 0.0f && a.v[2] >= 0.0f && a.v[3] >= 0.0f); }

struct PathState {
    Ray ray;
    float hero_nm;
    F4 thr, ppdf, rad;
    Sampler smp;
    uint32_t bounce;
};

struct CamDev { float px, py, pz; float m[16]; };
struct FrameDev { uint32_t width, height, seed, max_depth, min_depth; };

// tile.rs:134-157: the camera sample for (x, y, sample_index).  It consumes the first two values of set 0
// (sampler.rs:23-50); the other two (u2, u3) are what the first bounce draws next.
struct CamSample { uint32_t scramble; float hero_nm; V3 d; float u2, u3; };
IRIS_DEV CamSample camera_sample(const CamDev &cam, const FrameDev &fr, SobolLds tab, uint32_t x, uint32_t y, uint32_t sample_index) {
    CamSample cs;
    // Four quotients by the film's width and height per sample (tile.rs:134-138, :149-153).  1 <= W, H <= 2^31 (check_params); the
    // numerators are x + 0.5 in [0.5, 2^31] and half a draw, which is 0 or at least 2^-33: all inside exact::div_by's domain.
    const float fw = (float)fr.width, fh = (float)fr.height;
#if IRIS_EXACT_FAST
    const exact::Recip yw = exact::recip_for(fw), yh = exact::recip_for(fh);
#define IRIS_DIV_W(n) exact::div_by((n), fw, yw)
#define IRIS_DIV_H(n) exact::div_by((n), fh, yh)
#else
#define IRIS_DIV_W(n) ((n) / fw)
#define IRIS_DIV_H(n) ((n) / fh)
#endif
    float cx = (IRIS_DIV_W((float)x + 0.5f) - 0.5f) * 2.0f;
    float cy = (IRIS_DIV_H((float)y + 0.5f) - 0.5f) * -2.0f;
    Sampler smp = sampler_new(x, y, sample_index, fr.seed);
    cs.scramble = smp.scramble;
    cs.hero_nm = gen_golden_ratio(smp) * LAMBDA_RANGE_NM + LAMBDA_MIN_NM;  // wavelength.rs:26-28
    float u[4];
    sampler_set(tab, sample_index, 0u, smp.scramble, u);
    cs.u2 = u[2]; cs.u3 = u[3];
    float j0 = IRIS_DIV_W(0.5f * u[0]);
    float j1 = IRIS_DIV_H(0.5f * u[1]);
#undef IRIS_DIV_W
#undef IRIS_DIV_H
    float tx = cx + j0, ty = cy + j1, tz = 0.0f + 0.0f;
    const float *m = cam.m;  // matrix.rs:168-196
    float w = m[12] * tx + m[13] * ty + m[14] * tz + m[15];
    V3 target = div3_x(v3(m[0] * tx + m[1] * ty + m[2] * tz + m[3], m[4] * tx + m[5] * ty + m[6] * tz + m[7], m[8] * tx + m[9] * ty + m[10] * tz + m[11]), w);
    V3 origin = v3(cam.px, cam.py, cam.pz);
    cs.d = normalize_x(target - origin);  // ray.rs:14-19
    return cs;
}
// Starts the path of a camera sample in `ps`; the unread half of set 0 goes to the lane's ring.
IRIS_DEV void path_begin(const CamDev &cam, const CamSample &cs, uint32_t sample_index, DrawRing ring, PathState &ps) {
    ps.smp.scramble = cs.scramble; ps.smp.index = sample_index; ps.smp.draws = 2u; ps.smp.gen = 4u;
    ring[2 * IRIS_BLOCK] = cs.u2;
    ring[3 * IRIS_BLOCK] = cs.u3;
    ps.hero_nm = cs.hero_nm;
    ps.ray.o = v3(cam.px, cam.py, cam.pz);
    ps.ray.d = cs.d;
    ps.thr = splat(1.0f);
    ps.ppdf = splat(1.0f);
    ps.rad = splat(0.0f);
    ps.bounce = 0u;
}
IRIS_DEV void raygen(const CamDev &cam, const FrameDev &fr, SobolLds tab, DrawRing ring, uint32_t x, uint32_t y, uint32_t sample_index, PathState &ps) {
    CamSample cs = camera_sample(cam, fr, tab, x, y, sample_index);
    path_begin(cam, cs, sample_index, ring, ps);
}

IRIS_RADIOMETRIC_BEGIN
// What `radiance` returns for a finished path, as XYZ (tile.rs:159-162)
template <int INTEG>
IRIS_DEV V3 path_xyz(const PathState &ps) {
    F4 wl = wavelength_new(ps.hero_nm);
    F4 rad = (INTEG < 2) ? ps.rad : f4(hero(ps.rad), 0.0f, 0.0f, 0.0f);  // swss_naive.rs:87, swss_slow.rs:81
    return spectral_to_xyz(rad, wl);
}

// One iteration of the bounce loop after a closest hit (prim, t) has been found:
// hwss_naive.rs:46-88 + direct_light :95-157, hwss_slow.rs:46-80, swss_naive.rs:45-84 + :92-153,
// swss_slow.rs:45-78.  Returns true while the path continues (ps.ray is then the next ray).
// Every thread of the workgroup must call it (active=false for idle threads): the shadow scans
// stage the sphere list cooperatively.
// CLS: scene class the kernel is compiled for (iris_scene_upload picks it):
//   IRIS_CLS_GENERAL           every BSDF and spectrum kind;
//   IRIS_CLS_LAMBERT_CONSTANT  only Lambertian BSDFs over constant spectra (the reference's default scene):
//                              no other BSDF code is compiled in, and the four wavelengths of throughput,
//                              pdfs and radiance are equal by construction, so each 4-wide operation is one.
// Same operations in the same order on the same values either way: the film does not change by a bit.
#define IRIS_CLS_GENERAL 0
#define IRIS_CLS_LAMBERT_CONSTANT 1
#define IRIS_CLS_GENERAL_ENV 2  // IRIS_CLS_GENERAL plus the environment-map code (used when sc.env_w != 0)
template <int CLS>
IRIS_DEV bool env_on(const SceneDev &sc) { return CLS == IRIS_CLS_GENERAL_ENV && sc.env_w != 0u; }

// A ray that left the scene: `None => break` (hwss_naive.rs:41-44).  With an environment map the map's
// emission is added under the rule the integrator has for emitters it hits (after the commented miss branch
// of scene.rs:351-353): at bounce 0 only with next-event estimation, at every bounce without.
template <int INTEG, int CLS>
IRIS_DEV void path_miss(const SceneDev &sc, PathState &ps) {
    constexpr bool HWSS = (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_HWSS_SLOW);
    constexpr bool NAIVE = (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_SWSS_NAIVE);
    if (!env_on<CLS>(sc)) return;
    if (NAIVE && ps.bounce != 0u) return;
    F4 le;
    float pdf_unused;
    env_eval(sc, ps.ray.d, wavelength_new(ps.hero_nm), le, pdf_unused);
    if (HWSS) ps.rad = ps.rad + (ps.thr * le) * balance_heuristic_1(ps.ppdf);
    else ps.rad = ps.rad + ps.thr * le;
}
// What shade_bounce<.., DEFER != 0> leaves undone (Lambertian/constant class under a hierarchy, where the four wavelengths
// carry one value): visibility queries of direct_light are handed back to the caller, who has them traced elsewhere and then
// calls apply_deferred, which performs the additions of hwss_naive.rs:131-133, :148-151, :155 and :61-62 in the reference's
// order on the values kept here.
//   DEFER == 1 (pooled traversal of k_regwave): the light-sample query (hwss_naive.rs:115-123) goes to the wave's ray pool; the
//              BSDF-sampled ray's question is still answered inside shade_bounce.
//   DEFER == 2 (queue kernels, k_wfq_*): both go out -- shade_bounce compiles no traversal code at all.  The BSDF-sampled ray's
//              question "is the chosen light the closest hit?" (scene.rs:151-166) becomes: the light's own sphere is hit at
//              limit2 > 0 (tested here), and no sphere is preferred to that hit (`ray2` up to `limit2`, ties broken against
//              primitive `tie2`); its contribution c2 is formed here on the assumption that the answer is yes.
struct Deferred {
    bool pending;      // there is a direct-light term to add (the hit has a material)
    bool want1;        // the light sample passed the pre-checks: its visibility query is `ray1` up to `limit1`
    bool has2;         // DEFER 1: the BSDF-sampled ray reached the chosen light: `c2` counts.  DEFER 2: it reached the light's sphere: `c2` counts unless the query finds an occluder
    Ray ray1;
    float limit1;
    float c1, c2;      // the two MIS-weighted contributions (each the same for the four wavelengths)
    float thr, weight; // throughput before this bounce's update, light pick weight
    // DEFER 2 only
    Ray ray2;
    float limit2;
    int tie2;
    uint32_t spec_tests2;  // sphere tests made while forming c2 (Sphere::pdf from inside the light): counted only if c2 counts
};
IRIS_DEV void apply_deferred(const Deferred &df, bool occluded1, PathState &ps, bool occluded2 = false) {
    if (!df.pending) return;
    const bool visible = df.want1 && !occluded1 && __builtin_inff() > df.limit1;  // scene.rs:147-148: closest_t > target_t - eps
    F4 direct = splat(0.0f);
    if (visible) direct = direct + splat(df.c1);
    if (df.has2 && !occluded2) direct = direct + splat(df.c2);
    ps.rad = ps.rad + splat(df.thr) * (direct * df.weight);
}
template <int INTEG, int CLS, int DEFER = 0>
IRIS_DEV bool shade_bounce(const SceneDev &sc, const SphereStage &st, const FrameDev &fr, SobolLds tab, DrawRing ring, bool active,
                           PathState &ps, int prim, float t, Counters &cnt, Deferred *df = nullptr) {
    static_assert(!DEFER || CLS == IRIS_CLS_LAMBERT_CONSTANT, "the deferred form keeps one value for the four wavelengths");
    // (IRIS_DEBUG_INVARIANTS: the counters check the radiance where the bounce's light has been added, so the debug build launches
    // no deferring kernel -- PoolGeom::ON and use_queue_kernels are off there)
    if (DEFER && active) { df->pending = false; df->want1 = false; df->has2 = false;  // (lanes that wait for their rays keep what they deferred)
        df->limit1 = __builtin_inff(); df->c1 = 0.0f; df->c2 = 0.0f; df->thr = 0.0f; df->weight = 0.0f;
        if (DEFER == 2) { df->limit2 = __builtin_inff(); df->tie2 = 0x7fffffff; df->spec_tests2 = 0u; } }
    constexpr bool HWSS = (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_HWSS_SLOW);
    constexpr bool NAIVE = (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_SWSS_NAIVE);
    // all draws of this bounce, generated for the whole wave at once: light pick 1 + light point 2 + BSDF 2 + BSDF 2 +
    // roulette 1 with next-event estimation, BSDF 2 + roulette 1 without
    ensure_draws(tab, ring, ps.smp, NAIVE ? 8u : 3u, active);
    constexpr bool MONO = CLS == IRIS_CLS_LAMBERT_CONSTANT;
    constexpr bool TRIG_PIN = !(MONO && !DEFER);  // trig_const: the Lambertian list kernel keeps its sin/cos constants in vector registers
    if (MONO) { ps.thr = splat(ps.thr.v[0]); ps.ppdf = splat(ps.ppdf.v[0]); ps.rad = splat(ps.rad.v[0]); }
    bool alive = active;
    Hit hit = {};
    F4 wl = splat(LAMBDA_MIN_NM);
    iris_bsdf bsdf = {IRIS_BSDF_NULL, 0u, 0u, 0.0f, 0.0f};
    int light_index = -1;
    if (alive) {
        hit = sphere_frame(sc.spheres[prim], ps.ray, t);
        wl = wavelength_new(ps.hero_nm);
        int4 info = sc.prim_info[prim];
        light_index = info.y;
        if (!NAIVE && light_index >= 0) {  // hwss_slow.rs:46-49, swss_slow.rs:45-48
            int lp_unused;
            F4 le = spectrum_evaluate(light_spectrum(sc, (uint32_t)light_index, lp_unused), wl, MONO);
            if (HWSS) ps.rad = ps.rad + (ps.thr * le) * balance_heuristic_1(ps.ppdf);
            else ps.rad = ps.rad + ps.thr * le;
        }
        if (info.x < 0) alive = false;  // no material: break
        else {
            // Lambertian/constant class: everything the bounce needs of the material came with prim_info (one 16-byte load beside
            // the sphere's) instead of the chain prim_info -> bsdfs -> spec_over_pi, three dependent round trips
            if (CLS == IRIS_CLS_LAMBERT_CONSTANT) bsdf = iris_bsdf{IRIS_BSDF_LAMBERTIAN, 0u, 0u, __int_as_float(info.z), 0.0f};
            else bsdf = sc.bsdfs[info.x];
        }
    }
    if (alive) cnt.bounces++;
    V3 shading_wo = world_to_shading(hit, -ps.ray.d);

    if (NAIVE) {
        if (alive && ps.bounce == 0u && light_index >= 0) {  // hwss_naive.rs:51-58, swss_naive.rs:50-55
            int lp_unused;
            F4 le = spectrum_evaluate(light_spectrum(sc, (uint32_t)light_index, lp_unused), wl, MONO);
            if (HWSS) ps.rad = ps.rad + (ps.thr * le) * balance_heuristic_1(ps.ppdf);
            else ps.rad = ps.rad + ps.thr * le;
        }
        // ---- direct_light ----
        F4 direct = splat(0.0f);
        F4 light_emission = splat(0.0f);
        float4 light_sphere = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        int light_prim = -1;
        float light_pick_weight = 0.0f;
        // light sampling
        Ray shadow1 = {v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f)};
        V3 light_pos = v3(0.0f, 0.0f, 0.0f);
        float light_pdf = 0.0f;
        bool want1 = false;
        // With an environment map the map is entry n_lights of the light list (docs/SPEC.md section 6); its two
        // strategies mirror the sphere light's: a direction from the map's CDFs (2 draws, as Sphere::sample)
        // that must reach nothing, and the BSDF-sampled ray counted when it leaves the scene.  light_prim
        // stays -1 for it, which is what a scan that hits nothing reports.
        const bool ENV = env_on<CLS>(sc);
        bool env_pick = false;
        if (alive) {
            // scene.rs:168-172 + sampler.rs:52-60
            const uint32_t n_total = sc.n_lights + (ENV ? 1u : 0u);
            float u = gen_0_1(ring, ps.smp);
            uint32_t light_idx = f_as_index(u * (((float)n_total - 0.5f) - 0.0f) + 0.0f);
            if (light_idx >= n_total) light_idx = n_total - 1u;  // never taken for u in [0, 1); keeps a bad draw from indexing out of bounds
            light_pick_weight = (float)n_total;
            bool facing_forward;
            if (ENV && light_idx >= sc.n_lights) {
                env_pick = true;
                float u0 = gen_0_1(ring, ps.smp);
                float u1 = gen_0_1(ring, ps.smp);
                V3 dir = env_sample(sc, u0, u1);
                shadow1 = ray_spawn(hit.point, dir, hit.normal);
                env_eval(sc, shadow1.d, wl, light_emission, light_pdf);
                facing_forward = dot(dir, hit.normal) > 0.0f;
                want1 = light_pdf > 0.0f && facing_forward != hit.back_face;
            } else {
                light_emission = spectrum_evaluate(light_spectrum(sc, light_idx, light_prim), wl, MONO);
                light_sphere = sc.light_rec[3u * light_idx];
#if IRIS_DEBUG_INVARIANTS
                bool off_sphere = false;
                sphere_sample(light_sphere, hit, ring, ps.smp, light_pos, light_pdf, TRIG_PIN, &off_sphere);
                IRIS_INV(cnt, 3, off_sphere);
#else
                sphere_sample(light_sphere, hit, ring, ps.smp, light_pos, light_pdf, TRIG_PIN);
#endif
                shadow1 = ray_spawn_to(hit.point, light_pos, hit.normal);
                facing_forward = dot(light_pos - hit.point, hit.normal) > 0.0f;
                want1 = light_pdf > 0.0f && facing_forward != hit.back_face && distance_squared(light_pos, hit.point) > 0.00001f;
            }
        }
        float st1 = __builtin_inff();
        int sp1 = -1;
        bool visible_bvh = false;
        if (DEFER) {
            // the query and what it would add go back to the caller; nothing is traversed here (want1 implies active)
            if (want1) {
                df->want1 = true;
                df->ray1 = shadow1;
                cnt.rays++;
                df->limit1 = fdiv(len(light_pos - shadow1.o), len(shadow1.d)) - IRIS_RAY_EPSILON;
                V3 shading_wi = world_to_shading(hit, shadow1.d);
                F4 bsdf_values = bsdf_evaluate(sc, bsdf, shading_wi, shading_wo, wl, MONO);
                F4 bsdf_pdfs = bsdf_pdf(bsdf, shading_wi, shading_wo);
                float cos_t = fabs_(cos_theta(shading_wi));
                float mis_weight = HWSS ? balance_heuristic_2(splat(light_pdf), bsdf_pdfs) : fdiv(light_pdf, hero(bsdf_pdfs) + light_pdf);
                df->c1 = div4(((mis_weight * light_emission) * bsdf_values) * cos_t, light_pdf, MONO).v[0];
            }
        } else if (st.bvh) {
            // Scene::ray_hits_point and the map's visibility test (scene.rs:134-149) ask only whether the closest hit lies
            // beyond a limit: `closest_t > limit` with closest_t = inf when nothing is hit (the map: nothing hit at all,
            // i.e. limit = inf with equality allowed).  Under a hierarchy that needs no closest hit: it is false iff some
            // sphere is hit at 0 < t <= limit, or inf > limit is false (limit inf or NaN) -- so the lane leaves the tree at
            // the first occluder, and the limit culls boxes from the start.
            const bool env_lane = ENV && env_pick;
            float limit = __builtin_inff();
            if (want1 && !env_lane) limit = fdiv(len(light_pos - shadow1.o), len(shadow1.d)) - IRIS_RAY_EPSILON;
            int occluder = -1;
            float lim = limit;
            if (want1) bvh_closest<true>(sc, st, shadow1, len_squared(shadow1.d), lim, occluder, cnt.sphere_tests);
            visible_bvh = want1 && occluder < 0 && (env_lane || __builtin_inff() > limit);
        } else {
            scan_closest(sc, st, want1, shadow1, st1, sp1, cnt.sphere_tests);  // scene.rs:134-149
        }
        if (want1 && !DEFER) {
            cnt.rays++;
            bool visible;
            if (st.bvh) {
                visible = visible_bvh;
            } else if (ENV && env_pick) {
                visible = sp1 < 0;
            } else {
                float target_t = fdiv(len(light_pos - shadow1.o), len(shadow1.d));
                visible = st1 > target_t - IRIS_RAY_EPSILON;
            }
            if (visible) {
                V3 shading_wi = world_to_shading(hit, shadow1.d);
                F4 bsdf_values = bsdf_evaluate(sc, bsdf, shading_wi, shading_wo, wl, MONO);
                F4 bsdf_pdfs = bsdf_pdf(bsdf, shading_wi, shading_wo);
                float cos_t = fabs_(cos_theta(shading_wi));
                float mis_weight = HWSS ? balance_heuristic_2(splat(light_pdf), bsdf_pdfs) : fdiv(light_pdf, hero(bsdf_pdfs) + light_pdf);
                direct = direct + div4(((mis_weight * light_emission) * bsdf_values) * cos_t, light_pdf, MONO);
            }
        }
        // BSDF sampling
        Ray shadow2 = {v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f)};
        bool want2 = false;
        F4 bs_values = splat(0.0f), bs_pdfs = splat(0.0f);
        float bs_cos = 0.0f;
        if (alive) {
            V3 wi;
            bsdf_sample(sc, bsdf, shading_wo, wl, ring, ps.smp, MONO, wi, bs_values, bs_pdfs, TRIG_PIN);
            bs_cos = fabs_(cos_theta(wi));
            shadow2 = ray_spawn(hit.point, shading_to_world(hit, wi), hit.normal);
            want2 = hero(bs_pdfs) > 0.0f;
        }
        float st2 = __builtin_inff();
        int sp2 = -1;
        bool on_light_bvh = false;  // the closest hit of the BSDF-sampled ray is the chosen light (the map: nothing is hit)
        if (DEFER == 2) {
            // the light's own sphere here (one test, which most rays fail and are done); whether anything is preferred to that hit is
            // the caller's query
            if (want2) {
                const float limit = sphere_hit_t(stage_sphere(light_sphere), shadow2, len_squared(shadow2.d));
                cnt.sphere_tests++;  // the test of the light's own sphere is made whether or not it hits
                if (limit > 0.0f) { on_light_bvh = true; df->ray2 = shadow2; df->limit2 = limit; df->tie2 = light_prim; }
            }
        } else if (st.bvh) {
            // Scene::ray_hits_object (scene.rs:151-166) is used for one thing: `closest primitive == the light's`.  That
            // needs the light's own sphere to be hit -- one test, and most rays fail it and are done -- and then no sphere
            // that the scan would prefer to it: the same visibility query as above, with the light's hit as the limit and
            // its index as the tie rule.  (The map: no sphere hit at all.)
            const bool env_lane = ENV && env_pick;
            const float a2 = len_squared(shadow2.d);
            float limit = __builtin_inff();
            bool ask = want2;
            if (want2 && !env_lane) {
                limit = sphere_hit_t(stage_sphere(light_sphere), shadow2, a2);
                ask = limit > 0.0f;
                cnt.sphere_tests++;  // the test of the light's own sphere is made whether or not it hits
            }
            int other = -1;
            float lim = limit;
            if (ask) bvh_closest<true>(sc, st, shadow2, a2, lim, other, cnt.sphere_tests, env_lane ? 0x7fffffff : light_prim);
            on_light_bvh = ask && other < 0;
        } else {
            scan_closest(sc, st, want2, shadow2, st2, sp2, cnt.sphere_tests);  // scene.rs:151-166
        }
        if (want2) {
            cnt.rays++;
            if ((DEFER == 2 || st.bvh) ? on_light_bvh : sp2 == light_prim) {
                float lpdf;
                unsigned long long pdf_tests = 0ull;
                if (ENV && env_pick) env_eval(sc, shadow2.d, wl, light_emission, lpdf);
                else lpdf = sphere_pdf(light_sphere, hit, shadow2.d, DEFER == 2 ? pdf_tests : cnt.sphere_tests);
                float mis_weight = HWSS ? balance_heuristic_2(bs_pdfs, splat(lpdf)) : fdiv(hero(bs_pdfs), hero(bs_pdfs) + lpdf);
                const F4 c2 = div4(((mis_weight * light_emission) * bs_values) * bs_cos, hero(bs_pdfs), MONO);
                if (DEFER) { df->has2 = true; df->c2 = c2.v[0]; if (DEFER == 2) df->spec_tests2 = (uint32_t)pdf_tests; }
                else direct = direct + c2;
            }
        }
        if (DEFER) {
            if (alive) {
                df->pending = true;
                df->thr = ps.thr.v[0];
                df->weight = light_pick_weight;
            }
        } else if (alive) ps.rad = ps.rad + ps.thr * (direct * light_pick_weight);  // hwss_naive.rs:61-62, :155
    }

    if (alive) {
        IRIS_INV(cnt, 0, v3_bad(hit.point) || v3_bad(hit.normal) || v3_bad(hit.tangeant) || v3_bad(hit.bitangeant));
        IRIS_INV(cnt, 1, f4_bad(ps.rad));
        // hwss_naive.rs:64-88
        V3 wi;
        F4 bsdf_values, bsdf_pdfs;
        bsdf_sample(sc, bsdf, shading_wo, wl, ring, ps.smp, MONO, wi, bsdf_values, bsdf_pdfs, TRIG_PIN);
        IRIS_INV(cnt, 2, f4_bad(bsdf_pdfs));
        float cos_t = fabs_(cos_theta(wi));
        if (hero(bsdf_pdfs) == 0.0f || cos_t == 0.0f) {
            alive = false;
        } else {
            ps.thr = ps.thr * div4(bsdf_values * cos_t, hero(bsdf_pdfs), MONO);
            if (HWSS) ps.ppdf = ps.ppdf * bsdf_pdfs;
            IRIS_INV(cnt, 1, f4_bad(ps.thr));
            IRIS_INV(cnt, 2, HWSS && f4_bad(ps.ppdf));
            if (ps.bounce >= fr.min_depth) {
                float p = rmin(sum(ps.thr), 0.95f);
                if (gen_0_1(ring, ps.smp) > p) alive = false;
                else ps.thr = div4(ps.thr, p, MONO);
            }
            if (alive) {
                ps.ray = ray_spawn(hit.point, shading_to_world(hit, wi), hit.normal);
                IRIS_INV(cnt, 0, v3_bad(ps.ray.o) || v3_bad(ps.ray.d));
                ps.bounce += 1u;
                if (ps.bounce >= fr.max_depth) alive = false;  // for bounces in 0..MAX_DEPTH
            }
        }
    }
    return alive;
}
IRIS_RADIOMETRIC_END

}  // namespace iris
