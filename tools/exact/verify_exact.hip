// verify_exact.hip -- exhaustive checks, on the GPU, that the cheap division / square-root sequences of
// iris_amd/csrc/iris_exact.h return the correctly rounded IEEE result (the compiler's `/` and sqrtf, i.e. hipcc's
// -fhip-fp32-correctly-rounded-divide-sqrt expansions) for EVERY operand in their documented domains.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o verify_exact tools/exact/verify_exact.hip
//   ./verify_exact sqrt            all 2^32 bit patterns
//   ./verify_exact divpi           x / PI_F for all 2^32 x
//   ./verify_exact rcp             the refined reciprocal for all 2^32 d
//   ./verify_exact div [lo hi]     n / d for every pair of significands (2^23 x 2^23 = 7e13 pairs; lo..hi of 256 slices)
//   ./verify_exact divexp          n / d over all exponent pairs of the domain with random significands, and its edges
//   ./verify_exact wrappers        the guarded wrappers of iris_device.h on operands of EVERY kind (any exponent, zeros,
//                                  denormals, infinities, NaNs, negative values): same bits as the plain IEEE expression
//
// Why significand pairs are enough for div: inside the domain no intermediate of the sequence overflows, underflows or
// is denormal, so scaling n or d by a power of two scales every intermediate by a power of two exactly; the rounding
// decisions depend on the significands alone.  `divexp` samples the exponents to confirm that claim on the hardware.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../iris_amd/csrc/iris_device.h"  // the guarded wrappers (sqrt_g, normalize, div3, div4, div_pi, rcp_g, first_positive_root); includes iris_exact.h

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); std::exit(2); } } while (0)

struct Report { unsigned long long tested, bad; uint32_t ex_a, ex_b, ex_got, ex_want; };

__device__ __forceinline__ bool same(float a, float b) {
    const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    return x == y || (a != a && b != b);  // NaN payloads are not compared
}
__device__ void record(Report *r, float a, float b, float got, float want) {
    if (atomicAdd(&r->bad, 1ull) == 0ull) { r->ex_a = __float_as_uint(a); r->ex_b = __float_as_uint(b); r->ex_got = __float_as_uint(got); r->ex_want = __float_as_uint(want); }
}

// ---- one operand: every bit pattern ----
template <int WHAT>
__global__ void k_unary(Report *rep) {
    unsigned long long bad_local = 0ull;
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long tested = 0ull;
    for (unsigned long long i = u; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        float got, want;
        bool in_domain;
        if (WHAT == 0) { in_domain = iris::exact::sqrt_domain(x); got = iris::exact::sqrt_fast(x); want = __builtin_sqrtf(x); }
        else if (WHAT == 1) { in_domain = iris::exact::div_pi_domain(x); got = iris::exact::div_pi_fast(x); want = x / 3.14159274101257324219f; }
        else { in_domain = iris::exact::rcp_domain(x); got = iris::exact::rcp_refined(x); want = 1.0f / x; }
        if (!in_domain) continue;
        tested++;
        if (!same(got, want)) { bad_local++; record(rep, x, 0.0f, got, want); }
    }
    atomicAdd(&rep->tested, tested);
    (void)bad_local;
}

// ---- division: every significand pair, n in [1, 2), d in [1, 2) (and a second exponent so that q lands in both binades) ----
__global__ void k_div_pairs(Report *rep, uint32_t d_begin, uint32_t d_end) {
    // grid: one block per denominator significand; threads stride over the 2^23 numerators
    unsigned long long tested = 0ull;
    for (uint32_t dm = d_begin + blockIdx.x; dm < d_end; dm += gridDim.x) {
        const float d = __uint_as_float(0x3f800000u | dm);
        const iris::exact::Recip y = iris::exact::recip_for(d);
        for (uint32_t nm = threadIdx.x; nm < (1u << 23); nm += blockDim.x) {
            const float n = __uint_as_float(0x3f800000u | nm);
            const float got = iris::exact::div_by(n, d, y), want = n / d;
            if (!same(got, want)) record(rep, n, d, got, want);
        }
        tested += (1ull << 23) / blockDim.x;
    }
    atomicAdd(&rep->tested, tested);
}

// ---- division: exponents of the whole domain x random significands, plus the edges of the domain ----
__device__ uint32_t pcg(uint32_t &s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }
__global__ void k_div_exponents(Report *rep, uint32_t rounds) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long tested = 0ull;
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t r0 = pcg(s), r1 = pcg(s), r2 = pcg(s);
        // exponents anywhere (also outside the domain: those pairs are skipped by the domain test, which is what is being tested)
        float n = __uint_as_float((r0 & 0x807fffffu) | ((r2 & 0xffu) << 23));
        float d = __uint_as_float((r1 & 0x807fffffu) | (((r2 >> 8) & 0xffu) << 23));
        if ((r2 >> 16) & 1u) n = __uint_as_float(__float_as_uint(n) & 0xff800000u);          // powers of two
        if ((r2 >> 17) & 1u) d = __uint_as_float(__float_as_uint(d) | 0x007fffffu);          // all-ones significands
        if (((r2 >> 18) & 63u) == 0u) n = 0.0f;
        if (((r2 >> 18) & 63u) == 1u) n = -0.0f;
        if (!iris::exact::div_domain(n, d)) continue;
        tested++;
        const float got = iris::exact::div_by(n, d, iris::exact::recip_for(d)), want = n / d;
        if (!same(got, want)) record(rep, n, d, got, want);
    }
    atomicAdd(&rep->tested, tested);
}

// ---- the guarded wrappers on arbitrary operands: inside the domain they take the short sequences, outside the compiler's ----
// mode (wave-uniform, so that whole waves take the short sequences): 0 every lane moderate; 1 moderate except one lane; 2 anything
__device__ float any_float(uint32_t &s, uint32_t mode, bool odd_lane) {
    const uint32_t r = pcg(s), kind = pcg(s) & 15u;
    const float moderate = __uint_as_float((r & 0x807fffffu) | ((96u + (pcg(s) % 64u)) << 23));  // 2^-31 .. 2^32
    const float zero = __uint_as_float(r & 0x80000000u);
    if (mode == 0u || (mode == 1u && !odd_lane)) return (kind & 7u) == 0u ? zero : moderate;   // zeros are inside every numerator domain
    if (kind == 0u) return zero;
    if (kind == 1u) return __uint_as_float((r & 0x807fffffu));                 // denormal
    if (kind == 2u) return __uint_as_float((r & 0x80000000u) | 0x7f800000u);   // +-inf
    if (kind == 3u) return __uint_as_float(r | 0x7fc00000u);                   // NaN
    if (kind < 8u) return moderate;
    return __uint_as_float(r);                                                 // anything
}
__global__ void k_wrappers(Report *rep, uint32_t rounds) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 777u;
    unsigned long long tested = 0ull;
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64u, mode = ((wave * 2654435761u) ^ (it * 40503u)) % 3u;
        const bool odd = (threadIdx.x & 63u) == (it & 63u);
        const float x = any_float(s, mode, odd), y = any_float(s, mode, odd), z = any_float(s, mode, odd);
        // modes 0 and 1: a positive denominator, as the domains ask, on every lane but (mode 1) the odd one -- so that in mode 1
        // the wave takes the short sequences unless the odd lane's own operands forbid it, infinities and huge numerators included
        const float d = (mode == 0u || (mode == 1u && !odd)) ? iris::fabs_(any_float(s, 0u, false)) + 0x1p-31f : any_float(s, mode, odd);
        const iris::V3 v = iris::v3(x, y, z);
        {   const float xs = (mode == 0u || (mode == 1u && !odd)) ? iris::fabs_(x) : x;  // non-negative wherever the short sequence is to run
            const float got = iris::sqrt_g(xs), want = __builtin_sqrtf(xs); if (!same(got, want)) record(rep, xs, 0.0f, got, want); }
        {   const float got = iris::rcp_g(d), want = 1.0f / d; if (!same(got, want)) record(rep, d, 1.0f, got, want); }
        {   const float got = iris::div_pi(x), want = x / 3.14159274101257324219f; if (!same(got, want)) record(rep, x, 2.0f, got, want); }
        {   const iris::V3 got = iris::div3(v, d);
            if (!same(got.x, x / d)) record(rep, x, d, got.x, x / d);
            if (!same(got.y, y / d)) record(rep, y, d, got.y, y / d);
            if (!same(got.z, z / d)) record(rep, z, d, got.z, z / d); }
        {   const float w = any_float(s, mode, odd);
            const iris::F4 got = iris::div4(iris::f4(x, y, z, w), d, false);
            if (!same(got.v[0], x / d)) record(rep, x, d, got.v[0], x / d);
            if (!same(got.v[1], y / d)) record(rep, y, d, got.v[1], y / d);
            if (!same(got.v[2], z / d)) record(rep, z, d, got.v[2], z / d);
            if (!same(got.v[3], w / d)) record(rep, w, d, got.v[3], w / d); }
        {   const iris::V3 got = iris::normalize(v);
            const float l = __builtin_sqrtf((x * x + y * y) + z * z);
            if (!same(got.x, x / l)) record(rep, x, l, got.x, x / l);
            if (!same(got.y, y / l)) record(rep, y, l, got.y, y / l);
            if (!same(got.z, z / l)) record(rep, z, l, got.z, z / l); }
        {   // sphere.rs:33-35, :53 with a = d.d of a (nearly) unit direction, as first_positive_root sees it
            const float a = 1.0f + (float)((int)(pcg(s) & 15u) - 8) * 0x1p-23f, half_b = x, root = iris::fabs_(y);
            const float got = iris::first_positive_root(half_b, root, a, iris::a_is_unit(a), iris::exact::recip_for(a));
            float want = (-half_b - root) / a;
            if (!(want > 0.0f)) want = (-half_b + root) / a;
            const bool hit_got = got > 0.0f, hit_want = want > 0.0f;
            if (hit_got != hit_want || (hit_want && !same(got, want))) record(rep, half_b, root, got, want); }
        tested += 16ull;
    }
    atomicAdd(&rep->tested, tested);
}

static int finish(const char *what, Report *d_rep) {
    Report h;
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, d_rep, sizeof h, hipMemcpyDeviceToHost));
    std::printf("%s: tested %llu operands, %llu wrong", what, h.tested, h.bad);
    if (h.bad) std::printf("  e.g. a=%08x b=%08x got=%08x want=%08x", h.ex_a, h.ex_b, h.ex_got, h.ex_want);
    std::printf("\n");
    std::fflush(stdout);
    return h.bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const char *what = argc > 1 ? argv[1] : "sqrt";
    Report *d_rep;
    CHECK(hipMalloc((void **)&d_rep, sizeof(Report)));
    CHECK(hipMemset(d_rep, 0, sizeof(Report)));
    int rc = 0;
    if (!std::strcmp(what, "sqrt")) { hipLaunchKernelGGL(k_unary<0>, dim3(4096), dim3(256), 0, nullptr, d_rep); rc = finish("sqrt_fast == sqrtf on its domain", d_rep); }
    else if (!std::strcmp(what, "divpi")) { hipLaunchKernelGGL(k_unary<1>, dim3(4096), dim3(256), 0, nullptr, d_rep); rc = finish("div_pi_fast(x) == x / PI_F on its domain", d_rep); }
    else if (!std::strcmp(what, "rcp")) { hipLaunchKernelGGL(k_unary<2>, dim3(4096), dim3(256), 0, nullptr, d_rep); rc = finish("rcp_refined(d) == 1 / d on its domain", d_rep); }
    else if (!std::strcmp(what, "divexp")) { hipLaunchKernelGGL(k_div_exponents, dim3(4096), dim3(256), 0, nullptr, d_rep, 1u << 14); rc = finish("div_by == n / d, random exponents and edge cases", d_rep); }
    else if (!std::strcmp(what, "wrappers")) { hipLaunchKernelGGL(k_wrappers, dim3(4096), dim3(256), 0, nullptr, d_rep, 1u << 10); rc = finish("guarded wrappers == plain IEEE expressions, operands of every kind", d_rep); }
    else if (!std::strcmp(what, "div")) {
        const uint32_t lo = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 0u, hi = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 256u;
        for (uint32_t slice = lo; slice < hi && rc == 0; slice++) {  // 2^15 denominators per launch: seconds each, a progress line per slice
            CHECK(hipMemset(d_rep, 0, sizeof(Report)));
            hipLaunchKernelGGL(k_div_pairs, dim3(4096), dim3(256), 0, nullptr, d_rep, slice << 15, (slice + 1u) << 15);
            char label[96];
            std::snprintf(label, sizeof label, "div_by == n / d, significand pairs, slice %u of 256", slice);
            rc = finish(label, d_rep);
        }
    } else { std::fprintf(stderr, "unknown check %s\n", what); rc = 2; }
    CHECK(hipFree(d_rep));
    return rc;
}
