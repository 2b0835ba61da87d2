// iris_kernels.hip -- wavefront path-tracing kernels for gfx950 and the C ABI of libiris_hip.so.
//
// Layout of the work (DESIGN.md has the long version):
//   * A render is a list of items (sample index, pixel), sample-major.  Every persistent workgroup
//     owns a contiguous range of items, a private pool of path-state slots (SoA arrays in HBM, sized to
//     stay L2 / Infinity-Cache resident) and a private queue, so workgroups never talk to each other and
//     the result does not depend on scheduling.
//   * Stage A (generate + extend): every slot whose path is dead is refilled with the next item of the
//     workgroup's range (wave64 ballot + prefix sum, one LDS atomic per wave), the camera ray is generated,
//     and every live ray is intersected with the sphere list staged in LDS.  Misses are finished (their
//     XYZ goes to the per-sample result buffer), hits are compacted into the shade queue (ballot + prefix
//     sum again).
//   * Stage B (shade): one lane per queue entry: emission, next-event estimation with both MIS
//     strategies (two shadow scans), BSDF sampling, Russian roulette, next ray.
//   * Resolve: per pixel, the per-sample XYZ values are summed in ascending sample order (the order of
//     tile.rs:144-163) and scaled by 1/spp.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "iris_device.h"
#include "iris_bvh_build.h"


using namespace iris;

namespace {

constexpr int BLOCK = 256;
#define IRIS_N_COUNTERS 24  // work counters 0-3, phase timers 4-9 and 11-16 (diagnostic build), sky samples 10, invariant counters 20-23 (debug build)
static_assert(BLOCK == IRIS_BLOCK, "the draw ring stride is the workgroup size");
constexpr uint32_t STATE_ALIVE = 0x80000000u;

// Exact n / d for n < 2^31 by one multiply-high (Granlund-Montgomery: m = ceil(2^(31+L) / d), L = ceil(log2 d)).
struct FastDiv { uint32_t mul, shift; };  // shift = L - 1; divisor 1 is flagged by shift = 0xffffffff
inline FastDiv make_fastdiv(uint32_t d) {
    if (d <= 1u) return FastDiv{0u, 0xffffffffu};
    uint32_t L = 0;
    while ((1ull << L) < d) L++;
    const unsigned long long m = ((1ull << (31 + L)) + d - 1) / d;
    return FastDiv{(uint32_t)m, L - 1u};
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv fd) { return fd.shift == 0xffffffffu ? n : __umulhi(n, fd.mul) >> fd.shift; }

struct XyzResult { float x, y, z; };

struct Pool {
    float4 *rayo;  // (ox, oy, oz, hero wavelength)
    float4 *rayd;  // (dx, dy, dz, state bits: alive << 31 | bounce)
    float4 *thr;   // throughput, 4 wavelengths
    float4 *ppdf;  // path pdfs, 4 wavelengths
    float4 *rad;   // radiance, 4 wavelengths
    uint4 *meta;   // sampler scramble, sample index, item id, sampler draws consumed
    uint2 *hit;    // closest hit of stage A: primitive, t (f32 bits)
};

struct WfArgs {
    SceneDev sc;
    CamDev cam;
    FrameDev fr;
    Pool pool;
    XyzResult *results;       // per item: XYZ of the finished path (12 bytes: the buffer is written once and read once, both HBM-bound)
    uint32_t *queue;          // per workgroup: slots_per_block entries
    uint32_t *blk_cursor;     // per workgroup: next item (multi-kernel mode)
    uint32_t *blk_qn;         // per workgroup: queue length (multi-kernel mode)
    uint32_t *live_blocks;    // multi-kernel mode: workgroups that still have work after stage A
    unsigned long long *counters;  // samples, bounces, rays, sphere_tests
    uint32_t npix;
    FastDiv div_npix, div_width;
    uint32_t s_begin;         // sample index of item 0 of this chunk
    uint32_t n_items;         // items in this chunk
    uint32_t items_per_block;
    uint32_t slots_per_block;
    uint32_t sphere_cap;      // LDS window, in spheres
    const unsigned char *sky;  // register kernel, short lists, no environment map: per pixel, 1 = no camera sample of this pixel
                               // can hit anything (k_sky_mask); nullptr: not used
    uint32_t *park;           // pooled traversal (hierarchy + Lambertian/constant class): IRIS_WALK_WORDS words per lane for parked rays, word-major
    uint32_t *stack_hi;       // register kernel with a hierarchy: traversal-stack entries beyond the LDS ones, entry-major
    uint32_t stack_lds_n;     // traversal-stack entries per lane kept in LDS
    uint4 *kq;                // register kernel, general classes, hits sorted by BSDF kind: per wave, KQ_KINDS stacks of KQ_CAP path records
    unsigned long long *work_cursor;  // register kernel: next unclaimed item of the launch (one counter in global memory, zeroed per launch)
    uint32_t work_run;        // register kernel: items a wave claims at a time (a multiple of 64)
};

struct BlockShared {
    uint32_t cursor, w_end, qn;
};

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// exclusive prefix count of `mask` below this lane
__device__ __forceinline__ uint32_t prefix_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Reserve `popcount(ballot(pred))` consecutive entries of an LDS counter for the wave; returns this lane's index.
__device__ __forceinline__ uint32_t wave_reserve(bool pred, uint32_t *lds_counter) {
    unsigned long long mask = __ballot(pred);
    uint32_t count = (uint32_t)__popcll(mask);
    uint32_t base = 0u;
    if (count != 0u && lane_id() == 0u) base = atomicAdd(lds_counter, count);
    base = __builtin_amdgcn_readfirstlane(base);
    return base + prefix_below(mask);
}

__device__ __forceinline__ void stage_tables(uint2 *sobol_lds) {
    for (uint32_t e = threadIdx.x; e < IRIS_SOBOL_LUT_ENTRIES; e += blockDim.x) sobol_lds[e] = sobol_lut_entry(e);
}
__device__ __forceinline__ SphereStage stage_spheres(const SceneDev &sc, float4 *win, uint32_t cap) {
    SphereStage st;
    st.lds = win;
    st.cap = cap;
    st.bvh = sc.n_bvh_nodes != 0u;
    st.stack = (uint32_t *)win + threadIdx.x;  // hierarchy: the window's LDS holds the traversal stacks instead
    st.stack_lds_n = 0xffffffffu;              // all of them (the register kernel keeps only the first levels there)
    st.stack_hi = nullptr;
    st.stack_hi_stride = 0u;
    st.resident = !st.bvh && sc.n_prims <= cap;
    if (st.resident) {
        const uint32_t n_pad = (sc.n_prims + IRIS_SCAN_PAD - 1u) & ~(IRIS_SCAN_PAD - 1u);
        for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) win[i] = i < sc.n_prims ? stage_sphere(sc.spheres[i]) : stage_filler();
    }
    return st;
}

template <typename Args>
__device__ __forceinline__ void store_result(const Args &a, uint32_t item, V3 xyz) {
    XyzResult r = {xyz.x, xyz.y, xyz.z};
    a.results[item] = r;
}

__device__ __forceinline__ void flush_counters(unsigned long long *counters, const Counters &c) {
    // wave reduction, then one atomic per wave and counter
    unsigned long long v[4] = {c.samples, c.bounces, c.rays, c.sphere_tests};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane_id() == 0u && x) atomicAdd(&counters[k], x);
    }
#if IRIS_DEBUG_INVARIANTS
#pragma unroll
    for (int k = 0; k < 4; k++) {
        unsigned long long x = c.inv[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane_id() == 0u && x) atomicAdd(&counters[20 + k], x);
    }
#endif
}
__device__ __forceinline__ void flush_counters(const WfArgs &a, const Counters &c) { flush_counters(a.counters, c); }

__device__ __forceinline__ unsigned long long item_of(const WfArgs &a, uint32_t k);

// ---- stage A: refill dead slots, generate camera rays, closest hit, compact hits into the queue ----
template <int INTEG, int CLS>
__device__ __forceinline__ void stage_gen_extend(const WfArgs &a, const SphereStage &st, SobolLds tab, DrawRing ring, BlockShared &bs, Counters &cnt) {
    const uint32_t slot0 = blockIdx.x * a.slots_per_block;
    for (uint32_t tile = 0; tile < a.slots_per_block; tile += BLOCK) {
        const uint32_t slot = slot0 + tile + threadIdx.x;
        float4 rd = a.pool.rayd[slot];
        const bool was_alive = (__float_as_uint(rd.w) & STATE_ALIVE) != 0u;
        // refill: dead lanes take the next items of this workgroup's range
        const uint32_t k = wave_reserve(!was_alive, &bs.cursor);
        const unsigned long long id = item_of(a, k);
        const bool regen = !was_alive && k < bs.w_end && id < a.n_items;
        PathState ps;
        uint32_t item = 0u;
        if (regen) {
            item = (uint32_t)id;
            uint32_t s_local = fast_div(item, a.div_npix);
            uint32_t pix = item - s_local * a.npix;
            uint32_t y = fast_div(pix, a.div_width);
            uint32_t x = pix - y * a.fr.width;
            raygen(a.cam, a.fr, tab, ring, x, y, a.s_begin + s_local, ps);
            cnt.samples++;
        } else if (was_alive) {
            float4 ro = a.pool.rayo[slot];
            ps.ray.o = v3(ro.x, ro.y, ro.z);
            ps.ray.d = v3(rd.x, rd.y, rd.z);
            ps.hero_nm = ro.w;
        } else {
            ps.ray.o = v3(0.0f, 0.0f, 0.0f);
            ps.ray.d = v3(0.0f, 0.0f, 1.0f);
        }
        const bool active = regen || was_alive;
        float t;
        int prim;
        scan_closest(a.sc, st, active, ps.ray, t, prim, cnt.sphere_tests);
        if (active) cnt.rays++;
        const bool hitp = active && prim >= 0;
        if (active && !hitp) {  // the path left the scene: `None => break` (hwss_naive.rs:41-44)
            if (regen) {
                if (env_on<CLS>(a.sc)) {
                    path_miss<INTEG, CLS>(a.sc, ps);
                    store_result(a, item, path_xyz<INTEG>(ps));
                } else {
                    store_result(a, item, v3(0.0f, 0.0f, 0.0f));  // radiance is still exactly zero
                }
            } else {
                float4 r = a.pool.rad[slot];
                ps.rad = f4(r.x, r.y, r.z, r.w);
                if (env_on<CLS>(a.sc)) {
                    float4 th = a.pool.thr[slot];
                    ps.thr = f4(th.x, th.y, th.z, th.w);
                    if (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_HWSS_SLOW) { float4 pp = a.pool.ppdf[slot]; ps.ppdf = f4(pp.x, pp.y, pp.z, pp.w); }
                    else ps.ppdf = splat(1.0f);
                    ps.bounce = __float_as_uint(rd.w) & 0xffffu;
                    path_miss<INTEG, CLS>(a.sc, ps);
                }
                store_result(a, a.pool.meta[slot].z, path_xyz<INTEG>(ps));
                rd.w = __uint_as_float(0u);
                a.pool.rayd[slot] = rd;
            }
        }
        const uint32_t qi = wave_reserve(hitp, &bs.qn);
        if (hitp) {
            if (regen) {
                a.pool.rayo[slot] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.hero_nm);
                a.pool.rayd[slot] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __uint_as_float(STATE_ALIVE));
                a.pool.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                if (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_HWSS_SLOW) a.pool.ppdf[slot] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                a.pool.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                a.pool.meta[slot] = make_uint4(ps.smp.scramble, ps.smp.index, item, ps.smp.draws);
            }
            a.pool.hit[slot] = make_uint2((uint32_t)prim, __float_as_uint(t));
            a.queue[slot0 + qi] = slot;
        }
    }
}

// ---- stage B: shade every queued hit ----
template <int INTEG, int CLS>
__device__ __forceinline__ void stage_shade(const WfArgs &a, const SphereStage &st, SobolLds tab, DrawRing ring, uint32_t qn, Counters &cnt) {
    constexpr bool HWSS = (INTEG == IRIS_INT_HWSS_NAIVE || INTEG == IRIS_INT_HWSS_SLOW);
    const uint32_t slot0 = blockIdx.x * a.slots_per_block;
    for (uint32_t base = 0; base < qn; base += BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < qn;
        PathState ps;
        uint32_t slot = slot0, item = 0u;
        int prim = 0;
        float t = 0.0f;
        if (active) {
            slot = a.queue[slot0 + i];
            float4 ro = a.pool.rayo[slot], rd = a.pool.rayd[slot], th = a.pool.thr[slot], ra = a.pool.rad[slot];
            uint4 me = a.pool.meta[slot];
            uint2 h = a.pool.hit[slot];
            ps.ray.o = v3(ro.x, ro.y, ro.z);
            ps.ray.d = v3(rd.x, rd.y, rd.z);
            ps.hero_nm = ro.w;
            ps.bounce = __float_as_uint(rd.w) & 0xffffu;
            ps.thr = f4(th.x, th.y, th.z, th.w);
            if (HWSS) { float4 pp = a.pool.ppdf[slot]; ps.ppdf = f4(pp.x, pp.y, pp.z, pp.w); }
            else ps.ppdf = splat(1.0f);
            ps.rad = f4(ra.x, ra.y, ra.z, ra.w);
            ps.smp.scramble = me.x; ps.smp.index = me.y; ps.smp.draws = me.w;
            sampler_rewind(ps.smp);  // the ring of this lane holds another path's values
            item = me.z;
            prim = (int)h.x;
            t = __uint_as_float(h.y);
        } else {
            ps.ray.o = v3(0.0f, 0.0f, 0.0f); ps.ray.d = v3(0.0f, 0.0f, 1.0f); ps.hero_nm = LAMBDA_MIN_NM; ps.bounce = 0u;
            ps.thr = splat(0.0f); ps.ppdf = splat(1.0f); ps.rad = splat(0.0f);
            ps.smp = sampler_new(0u, 0u, 0u, 0u);
        }
        const bool cont = shade_bounce<INTEG, CLS>(a.sc, st, a.fr, tab, ring, active, ps, prim, t, cnt);
        if (active) {
            if (cont) {
                a.pool.rayo[slot] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.hero_nm);
                a.pool.rayd[slot] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __uint_as_float(STATE_ALIVE | ps.bounce));
                a.pool.thr[slot] = make_float4(ps.thr.v[0], ps.thr.v[1], ps.thr.v[2], ps.thr.v[3]);
                if (HWSS) a.pool.ppdf[slot] = make_float4(ps.ppdf.v[0], ps.ppdf.v[1], ps.ppdf.v[2], ps.ppdf.v[3]);
                a.pool.rad[slot] = make_float4(ps.rad.v[0], ps.rad.v[1], ps.rad.v[2], ps.rad.v[3]);
                a.pool.meta[slot] = make_uint4(ps.smp.scramble, ps.smp.index, item, ps.smp.draws);
            } else {
                store_result(a, item, path_xyz<INTEG>(ps));
                a.pool.rayd[slot] = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(0u));
            }
        }
    }
}

// Items are dealt to the workgroups in runs of 64 consecutive items, round-robin, so that every
// workgroup sees a representative mix of the image (sky rows are cheap, ground rows expensive) and all
// of them finish together.  The k-th item of workgroup b is item_of(k); ids beyond n_items (last round
// only) are skipped.
__device__ __forceinline__ void block_range(const WfArgs &a, uint32_t &begin, uint32_t &end) {
    begin = 0u;
    end = a.items_per_block;
}
__device__ __forceinline__ unsigned long long item_of(const WfArgs &a, uint32_t k) {
    return (unsigned long long)(k >> 6) * ((unsigned long long)gridDim.x << 6) + ((unsigned long long)blockIdx.x << 6) + (k & 63u);
}

// ---- the persistent kernel: both stages in a loop, one workgroup = one independent wavefront tracer ----
// Register budget: the Lambertian/constant-spectrum variant fits three waves per SIMD without spilling; the
// general one (every BSDF compiled in) is held at two.
template <int INTEG, int CLS>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(CLS == IRIS_CLS_LAMBERT_CONSTANT ? 3 : 2)))
k_wavefront(WfArgs a) {
    extern __shared__ float4 sphere_win[];
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    __shared__ BlockShared bs;
    stage_tables(sobol_lds);
    DrawRing ring = ring_lds + threadIdx.x;
    SphereStage st = stage_spheres(a.sc, sphere_win, a.sphere_cap);
    if (threadIdx.x == 0) {
        block_range(a, bs.cursor, bs.w_end);
        bs.qn = 0u;
    }
    const uint32_t slot0 = blockIdx.x * a.slots_per_block;
    for (uint32_t s = threadIdx.x; s < a.slots_per_block; s += BLOCK) a.pool.rayd[slot0 + s] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    __syncthreads();
    Counters cnt = {0u, 0u, 0u, 0u};
    for (;;) {
        stage_gen_extend<INTEG, CLS>(a, st, sobol_lds, ring, bs, cnt);
        __syncthreads();
        const uint32_t qn = bs.qn;
        const bool more = bs.cursor < bs.w_end;
        if (qn == 0u && !more) break;
        stage_shade<INTEG, CLS>(a, st, sobol_lds, ring, qn, cnt);
        __syncthreads();
        if (threadIdx.x == 0) bs.qn = 0u;
        __syncthreads();
    }
    flush_counters(a, cnt);
}


// ---- the register-resident variant for sphere lists that fit the LDS window ----
// Same per-lane device code, but the state of a path never leaves the registers of the lane that shades
// it: there is no pool in HBM and no workgroup barrier in the loop, so waves run independently.  What
// keeps the lanes busy is a queue of PRIMARY HITS per wave in LDS:
//   generate:  all 64 lanes take the next 64 camera samples of the workgroup's list, generate the camera
//              ray and intersect it.  Misses are finished on the spot (their radiance is exactly zero);
//              hits are compacted (ballot + prefix sum) into the wave's queue as an 11-dword record.
//              Runs whenever fewer than PQ_LOW records are queued, always with a full wave.
//   extend:    lanes whose path survived the last bounce intersect their new ray; then every path that has finished -- a ray that
//              just missed, a path that ended in the last shade phase (LANE_DONE) -- is converted to XYZ and stored, in one step.
//   refill:    lanes without a path pop a record each and become the first bounce of that path (general classes: hits are
//              sorted by BSDF kind first, see below).
//   shade:     every lane that holds a hit shades one bounce.
// The expensive shade phase therefore runs with (nearly) every lane busy as long as work is left, and the
// camera-sample code never runs for a handful of lanes.  No other wave touches a wave's queue, so no
// synchronisation is needed.
enum : uint32_t { LANE_DEAD = 0u, LANE_RAY = 1u, LANE_HIT = 2u, LANE_HITX = 3u, LANE_FIN = 4u, LANE_DONE = 5u };  // HITX: holds a hit of a BSDF kind that is not this iteration's; FIN: the path has ended, its last visibility query is still out; DONE: finished, to be stored
constexpr uint32_t PQ_FIELDS = 11;  // item, sample index, scramble, hero, d.xyz, t, prim, u2, u3
// General classes (SORTED), hits sorted by BSDF kind: a wave that shades hits on four BSDFs in one iteration runs ALL of that code for
// a quarter of the lanes each, which is what halved the lane utilisation of such scenes (S3 52 %, config E 44 %).  So every wave owns
// one stack of whole path records per BSDF kind in global memory (L2-resident: only the tops are touched).  Before a shade iteration the wave picks ONE kind -- the one it can fill most
// lanes with -- moves the hits of the other kinds out of its lanes into their stacks and takes hits of the chosen kind in, so the
// iteration runs one BSDF's code with (nearly) every lane instead of four BSDFs' code with a quarter of the lanes each.  A path is a
// function of (pixel, sample index) only, so which lane and which iteration shade a bounce changes no value.  A record: ray origin +
// hero wavelength, direction + bounce, throughput, path pdfs, radiance, sampler (scramble, index, item, draws), hit (prim, t): 7 x 16 B.
// The lane's draw ring is not moved: the lane that takes a record generates the set in progress again (sampler_rewind), which costs no
// extra pass of ensure_draws in most iterations (some lane needs the third set anyway).  Primary hits on Lambertian surfaces keep
// their compact queue in LDS; primary hits on the other kinds go to their stack as whole records.
// Measured (1920x1080x256, depth 32, HwssNaive): S3 55.7 -> 41.6 ms, lanes 52 -> 77 %, 11.5 records out and 21.8 in per iteration of
// 63.2 lanes; config E 77.5 -> 61.9 ms, lanes 44 -> 64 %.  In flight per wave (KQ_LOW): 64 49.0 / 73.8 ms, 192 41.6 / 61.9, 448 42.1 /
// 62.4, 960 42.4 / 62.7.  (The scheme this replaces set hits on non-Lambertian surfaces aside in the lanes and in a second primary
// queue and shaded them together in "mixed" iterations: S3 71.5 -> 67.1 ms at the time, a loss under an environment map.)
#ifndef IRIS_KQ_CAP
#define IRIS_KQ_CAP 256u   // records per stack; at most KQ_CAP hits are in flight per wave (lanes + stacks + primary queue)
#endif
#ifndef IRIS_KQ_LOW
#define IRIS_KQ_LOW 192u   // camera samples are generated while fewer hits than this are in flight (64 more may come of one round)
#endif
constexpr uint32_t KQ_KINDS = 4u, KQ_FIELDS = 7u, KQ_CAP = IRIS_KQ_CAP, KQ_LOW = IRIS_KQ_LOW;
static_assert(KQ_LOW + 64u <= KQ_CAP, "a generate round adds up to 64 hits");
// BVH (compile-time): the scene has a hierarchy.  Traversal is bound by the latency of dependent L2 reads, so
// what counts is resident waves, and LDS is what limits them: the per-lane traversal stacks take 1 KiB per
// level.  That variant therefore does without the Sobol nibble table (direction vectors come from the constant
// table in global memory, bit by bit) and keeps a shorter primary-hit queue, which buys the third workgroup per
// CU; the list-scan variant compiles no traversal code at all.
template <int CLS, bool BVH> struct PqGeom {  // < CAP - 63 queued when a generate phase starts, <= 64 added by it
    static constexpr uint32_t CAP = BVH ? (CLS == IRIS_CLS_LAMBERT_CONSTANT ? 72u : 80u) : 104u;  // the pooled variant gives 1.4 KiB to its ray pool (three workgroups per CU)
};

// POOLED (hierarchy + Lambertian/constant class: the 10k-sphere configuration): the extension rays and the light-sample visibility
// queries of a wave are traversed together by bvh_pooled, detached from the lanes that own them (iris_device.h); the shade phase
// leaves its visibility query to the next loop top (shade_bounce<.., DEFER>), where the paths that ended wait for it as LANE_FIN.
template <int CLS, bool BVH> struct PoolGeom { static constexpr bool ON = BVH && CLS == IRIS_CLS_LAMBERT_CONSTANT && !IRIS_DEBUG_INVARIANTS; };  // (the debug build defers nothing)
// Three waves per SIMD (168 VGPR) for every class: the general classes would fit 181-189 VGPR without their 28-44 B of scratch at
// two waves, and lose more to the missing wave than the spills cost (S3 225 -> 279 ms, config E 320 -> 401 ms).
template <int INTEG, int CLS, bool BVH>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(3))) k_regwave(WfArgs a) {
    constexpr bool POOLED = PoolGeom<CLS, BVH>::ON;
    constexpr uint32_t PQ_CAP = PqGeom<CLS, BVH>::CAP;
    constexpr uint32_t PQ_LOW = PQ_CAP - 63u;  // generate while fewer records than this are queued
    extern __shared__ float4 sphere_win[];
    __shared__ uint2 sobol_tab[BVH ? 1 : IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    __shared__ uint32_t primq[BLOCK / 64][PQ_FIELDS][PQ_CAP];
    __shared__ float ray_pool[POOLED ? BLOCK / 64 : 1][POOLED ? IRIS_POOL_FIELDS * IRIS_POOL_CAP : 1];
    if (!BVH) stage_tables(sobol_tab);
    const SobolLds sobol_lds = BVH ? nullptr : sobol_tab;
    DrawRing ring = ring_lds + threadIdx.x;
    SphereStage st = stage_spheres(a.sc, sphere_win, a.sphere_cap);
    st.bvh = BVH;        // compile-time: the other scan's code is not compiled in
    if (BVH) {
        st.stack_lds_n = a.stack_lds_n;
        st.stack_hi = a.stack_hi + (size_t)blockIdx.x * BLOCK + threadIdx.x;
        st.stack_hi_stride = gridDim.x * BLOCK;
    }
    st.resident = !BVH;  // launched only when the whole list is in the window (or with a hierarchy): no barrier inside the scans
    __syncthreads();
    // Work distribution: a wave claims `work_run` consecutive items at a time from one counter in global memory (one
    // atomic per run, by one lane), so the waves of the whole chip run dry within one run of each other whatever
    // the pixels cost -- no workgroup is dealt a share in advance.  Which wave traces an item changes nothing: every
    // result is a function of the item and goes to the item's own slot.
    uint32_t w_next = 0u, w_lim = 0u;  // wave-uniform: the unclaimed rest of this wave's current run
    uint32_t(*q)[PQ_CAP] = primq[threadIdx.x / 64];
    constexpr bool SORTED = CLS != IRIS_CLS_LAMBERT_CONSTANT;  // (never POOLED: that is the Lambertian/constant class)
    // sky mask (k_sky_mask): short lists without an environment map only -- under a map a miss needs its ray
    const bool USE_SKY = !BVH && CLS != IRIS_CLS_GENERAL_ENV && a.sky != nullptr;
    uint32_t n_sky = 0u;
    // SORTED: this wave's stacks, their fill (wave-uniform), the BSDF kind of the hit a lane holds
    uint4 *const kq = SORTED ? a.kq + (size_t)(blockIdx.x * (BLOCK / 64) + threadIdx.x / 64) * (KQ_KINDS * KQ_FIELDS * KQ_CAP) : nullptr;
    uint32_t kc[KQ_KINDS] = {0u, 0u, 0u, 0u};
    // (no stacks -- the scene holds one kind only, option sort_kinds = 0, or the entry point that may not allocate: every hit counts as kind 0,
    // nothing is ever moved, and an iteration shades whatever the lanes hold)
    const bool sort_on = SORTED && a.kq != nullptr;
    auto kind_of = [&](int prim_) -> uint32_t { return sort_on ? (uint32_t)a.sc.prim_kind[prim_] : 0u; };
    auto kq_store = [&](uint32_t k_, uint32_t slot_, const PathState &p_, uint32_t item_, int prim_, float t_) {
        uint4 *r = kq + (size_t)k_ * (KQ_FIELDS * KQ_CAP) + slot_;
        r[0 * KQ_CAP] = make_uint4(__float_as_uint(p_.ray.o.x), __float_as_uint(p_.ray.o.y), __float_as_uint(p_.ray.o.z), __float_as_uint(p_.hero_nm));
        r[1 * KQ_CAP] = make_uint4(__float_as_uint(p_.ray.d.x), __float_as_uint(p_.ray.d.y), __float_as_uint(p_.ray.d.z), p_.bounce);
        r[2 * KQ_CAP] = make_uint4(__float_as_uint(p_.thr.v[0]), __float_as_uint(p_.thr.v[1]), __float_as_uint(p_.thr.v[2]), __float_as_uint(p_.thr.v[3]));
        if (INTEG < 2) r[3 * KQ_CAP] = make_uint4(__float_as_uint(p_.ppdf.v[0]), __float_as_uint(p_.ppdf.v[1]), __float_as_uint(p_.ppdf.v[2]), __float_as_uint(p_.ppdf.v[3]));
        r[4 * KQ_CAP] = make_uint4(__float_as_uint(p_.rad.v[0]), __float_as_uint(p_.rad.v[1]), __float_as_uint(p_.rad.v[2]), __float_as_uint(p_.rad.v[3]));
        r[5 * KQ_CAP] = make_uint4(p_.smp.scramble, p_.smp.index, item_, p_.smp.draws);
        r[6 * KQ_CAP] = make_uint4((uint32_t)prim_, __float_as_uint(t_), 0u, 0u);
    };
    Counters cnt = {0u, 0u, 0u, 0u};
    PathState ps;
    ps.ray.o = v3(0.0f, 0.0f, 0.0f); ps.ray.d = v3(0.0f, 0.0f, 1.0f); ps.hero_nm = LAMBDA_MIN_NM; ps.bounce = 0u;
    ps.thr = splat(0.0f); ps.ppdf = splat(1.0f); ps.rad = splat(0.0f);
    ps.smp = sampler_new(0u, 0u, 0u, 0u);
    uint32_t lane_state = LANE_DEAD, item = 0u;
    int prim = -1;
    float t = 0.0f;
    bool work_left = true;               // wave-uniform
    uint32_t q_head = 0u, q_count = 0u;  // wave-uniform
#ifdef IRIS_PHASE_TIMERS  // diagnostic build only: wave-cycles per phase into counters[4..7]
    unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull}, ph_t = __builtin_readcyclecounter(), ph_lanes = 0ull, ph_iters = 0ull, ph_mixed = 0ull, ph_mixed_lanes = 0ull, ph_mixed_x = 0ull, ph_kind[4] = {0ull, 0ull, 0ull, 0ull};
#define IRIS_PHASE(k) do { unsigned long long n_ = __builtin_readcyclecounter(); ph[k] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define IRIS_PHASE(k) do { } while (0)
#endif
    Deferred df = {};
    // POOLED: what this lane has in the wave's ray pool (0 nothing, 1 to be put at the next loop top, 2 out, 3 extension ray back);
    // the slots no walker has taken yet (wave-uniform); whether this lane, as a walker, has a ray parked in global memory
    uint32_t own_e = 0u, own_s = 0u;
    bool occluded1 = false, walking = false;
    unsigned long long pend_e = 0ull, pend_s = 0ull;
    for (;;) {
        // ---- extend: the next ray of the paths that survived the last bounce ----
        if (POOLED) {
            float *const rec = ray_pool[threadIdx.x / 64];
            const uint32_t me = lane_id();
            // owners put the rays of the bounce just shaded: the spheres outside the tree are tested here, with the whole wave busy
            {
                const bool put_e = own_e == 1u;
                int unused;
                if (put_e) { (void)pool_put(a.sc, rec, me, ps.ray, false, __builtin_inff(), unused, cnt.sphere_tests); own_e = 2u; }
                pend_e |= __ballot(put_e);
                bool put_s = false;
                if (own_s == 1u) {
                    int occluder;
                    put_s = pool_put(a.sc, rec, 64u + me, df.ray1, true, df.limit1, occluder, cnt.sphere_tests);
                    own_s = put_s ? 2u : 0u;
                    occluded1 = !put_s;
                }
                pend_s |= __ballot(put_s);
                __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order: the walkers see these records
            }
            if ((pend_e | pend_s) != 0ull || __ballot(walking) != 0ull)
                bvh_pool_pass(a.sc, st, rec, a.park + (size_t)blockIdx.x * BLOCK + threadIdx.x, (size_t)gridDim.x * BLOCK, pend_e, pend_s, walking, cnt.sphere_tests);
            // owners: what has come back?
            if (own_e == 2u) {
                float t_new;
                int prim_new;
                if (pool_get(rec, me, t_new, prim_new)) { t = t_new; prim = prim_new; own_e = 3u; }
            }
            if (own_s == 2u) {
                float t_unused;
                int occluder;
                if (pool_get(rec, 64u + me, t_unused, occluder)) { occluded1 = occluder >= 0; own_s = 0u; }
            }
            // a path moves on when both of its rays are back: first the direct light of the bounce just shaded (hwss_naive.rs:61-62),
            // then the end of the path or its next hit
            if ((lane_state == LANE_RAY || lane_state == LANE_FIN) && own_e != 2u && own_s != 2u) {
                apply_deferred(df, occluded1, ps);
                df.pending = false;
                if (lane_state == LANE_FIN) {
                    store_result(a, item, path_xyz<INTEG>(ps));
                    lane_state = LANE_DEAD;
                } else {
                    cnt.rays++;
                    own_e = 0u;
                    if (prim >= 0) {
                        lane_state = LANE_HIT;
                    } else {  // `None => break` (hwss_naive.rs:41-44)
                        store_result(a, item, path_xyz<INTEG>(ps));
                        lane_state = LANE_DEAD;
                    }
                }
            }
        } else {
            const bool active = lane_state == LANE_RAY;
            if (__ballot(active) != 0ull) {
                float t_new;
                int prim_new;
                scan_closest(a.sc, st, active, ps.ray, t_new, prim_new, cnt.sphere_tests);
                if (active) {
                    cnt.rays++;
                    t = t_new;
                    prim = prim_new;
                    if (prim >= 0) {
                        lane_state = LANE_HIT;
                    } else {  // `None => break` (hwss_naive.rs:41-44)
                        path_miss<INTEG, CLS>(a.sc, ps);
                        lane_state = LANE_DONE;
                    }
                }
            }
            // Finished paths -- the rays that just missed and the paths that ended in the last shade phase -- are converted to XYZ
            // and stored HERE, together: one execution of the conversion per iteration for all of them instead of one in each phase
            // for a part of the lanes each.
            if (__ballot(lane_state == LANE_DONE) != 0ull) {
                if (lane_state == LANE_DONE) {
                    store_result(a, item, path_xyz<INTEG>(ps));
                    lane_state = LANE_DEAD;
                }
            }
        }
        IRIS_PHASE(0);
        // ---- generate: camera samples, 64 at a time, until a wave's worth of primary hits is queued ----
        // (SORTED: hits in flight -- in lanes, in the stacks, in the primary queue -- stay below KQ_CAP, which is what bounds every stack)
        uint32_t in_flight = 0u;
        if (SORTED && sort_on) in_flight = (uint32_t)__popcll(__ballot(lane_state == LANE_HIT)) + q_count + kc[0] + kc[1] + kc[2] + kc[3];
        while (work_left && q_count < PQ_LOW && (!SORTED || in_flight < KQ_LOW)) {
            if (w_next >= w_lim) {
                unsigned long long claim = 0ull;  // 64 bits: every wave asks once too often, and grids and runs can be made large
                if (lane_id() == 0u) claim = atomicAdd(a.work_cursor, (unsigned long long)a.work_run);
                const uint32_t start = __builtin_amdgcn_readfirstlane((uint32_t)claim);
                if (__builtin_amdgcn_readfirstlane((uint32_t)(claim >> 32)) != 0u || start >= a.n_items) { work_left = false; break; }
                w_next = start;
                w_lim = min(start + a.work_run, a.n_items);
            }
            const uint32_t id = w_next + lane_id();
            const bool valid = id < w_lim;
            w_next += 64u;
            CamSample cs = {};
            Ray ray = {v3(a.cam.px, a.cam.py, a.cam.pz), v3(0.0f, 0.0f, 1.0f)};
            uint32_t new_item = 0u, sample_index = 0u, pix = 0u;
            bool live = valid;  // false for a sample of a pixel that k_sky_mask has marked: it is exactly zero and never stored
            if (valid) {
                new_item = (uint32_t)id;
                uint32_t s_local = fast_div(new_item, a.div_npix);
                pix = new_item - s_local * a.npix;
                sample_index = a.s_begin + s_local;
                cnt.samples++;
                if (USE_SKY && a.sky[pix] != 0) {  // counted as the reference executes it: one scan over the whole list
                    live = false;
                    cnt.rays++;
                    cnt.sphere_tests += a.sc.n_prims;
                    n_sky++;
                }
            }
            float t_new = 0.0f;
            int prim_new = -1;
            if (!USE_SKY || __ballot(live) != 0ull) {  // 64 consecutive pixels of a row: mostly all sky or none
                if (live) {
                    uint32_t y = fast_div(pix, a.div_width);
                    uint32_t x = pix - y * a.fr.width;
                    cs = camera_sample(a.cam, a.fr, sobol_lds, x, y, sample_index);
                    ray.d = cs.d;
                }
                scan_closest(a.sc, st, live, ray, t_new, prim_new, cnt.sphere_tests);
            }
            if (live) {
                cnt.rays++;
                if (prim_new < 0) {  // `None => break`
                    if (env_on<CLS>(a.sc)) {
                        PathState pm;
                        pm.ray = ray; pm.hero_nm = cs.hero_nm; pm.bounce = 0u;
                        pm.thr = splat(1.0f); pm.ppdf = splat(1.0f); pm.rad = splat(0.0f);
                        path_miss<INTEG, CLS>(a.sc, pm);
                        store_result(a, new_item, path_xyz<INTEG>(pm));
                    } else {
                        store_result(a, new_item, v3(0.0f, 0.0f, 0.0f));  // radiance is exactly zero
                    }
                }
            }
            const bool hitp = live && prim_new >= 0;
            uint32_t kind_new = 0u;
            if (SORTED && hitp) kind_new = kind_of(prim_new);
            const bool hitx = hitp && kind_new != 0u;
            const unsigned long long hmask = __ballot(hitp && !hitx);
            if (hitp && !hitx) {
                const uint32_t e = (q_head + q_count + prefix_below(hmask)) % PQ_CAP;
                q[0][e] = new_item; q[1][e] = sample_index; q[2][e] = cs.scramble; q[3][e] = __float_as_uint(cs.hero_nm);
                q[4][e] = __float_as_uint(cs.d.x); q[5][e] = __float_as_uint(cs.d.y); q[6][e] = __float_as_uint(cs.d.z);
                q[7][e] = __float_as_uint(t_new); q[8][e] = (uint32_t)prim_new;
                q[9][e] = __float_as_uint(cs.u2); q[10][e] = __float_as_uint(cs.u3);
            }
            q_count += (uint32_t)__popcll(hmask);
            if (SORTED) {
                // primary hits on other BSDFs start their path here and go to the stack of their kind as whole records
                in_flight += (uint32_t)__popcll(__ballot(hitp));
                if (__ballot(hitx) != 0ull) {
                    uint32_t slot = 0u;
#pragma unroll
                    for (uint32_t j = 1; j < KQ_KINDS; j++) {
                        const unsigned long long m = __ballot(hitx && kind_new == j);
                        if (hitx && kind_new == j) slot = kc[j] + prefix_below(m);
                        kc[j] += (uint32_t)__popcll(m);
                    }
                    if (hitx) {
                        PathState pn;
                        pn.smp.scramble = cs.scramble; pn.smp.index = sample_index; pn.smp.draws = 2u; pn.smp.gen = 4u;  // as path_begin leaves it; the ring is
                        pn.hero_nm = cs.hero_nm; pn.ray = ray; pn.bounce = 0u;                                           // filled again by whoever takes the record
                        pn.thr = splat(1.0f); pn.ppdf = splat(1.0f); pn.rad = splat(0.0f);
                        kq_store(kind_new, slot, pn, new_item, prim_new, t_new);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        IRIS_PHASE(1);
        // ---- SORTED: pick the BSDF kind of this iteration, move the other kinds' hits out and hits of that kind in ----
        if (SORTED && sort_on) {
            const uint32_t kind = lane_state == LANE_HIT ? kind_of(prim) : 0u;  // (looked up here, one byte: not worth a register across the shade phase)
            uint32_t best = 0u, best_score = 0u, sel_count = 0u;
#pragma unroll
            for (uint32_t j = 0; j < KQ_KINDS; j++) {
                const uint32_t n_lane = (uint32_t)__popcll(__ballot(lane_state == LANE_HIT && kind == j));
                const uint32_t avail = n_lane + kc[j] + (j == 0u ? q_count : 0u);
                const uint32_t score = (avail < 64u ? avail : 64u) * 128u + n_lane;  // most lanes filled; among equals, fewest records moved
                if (score > best_score) { best_score = score; best = j; sel_count = kc[j]; }
            }
            if (best_score != 0u) {
                const uint32_t supply = sel_count + (best == 0u ? q_count : 0u);
                const bool dead = lane_state == LANE_DEAD, other = lane_state == LANE_HIT && kind != best;
                const unsigned long long dmask = __ballot(dead), omask = __ballot(other);
                const uint32_t n_dead = (uint32_t)__popcll(dmask), n_other = (uint32_t)__popcll(omask);
                const uint32_t fill_dead = n_dead < supply ? n_dead : supply;
                const uint32_t n_evict = n_other < supply - fill_dead ? n_other : supply - fill_dead;  // a hit is moved out only if one of the chosen kind takes its lane
                const bool evict = other && prefix_below(omask) < n_evict;
#ifdef IRIS_PHASE_TIMERS
                ph_mixed += 1ull; ph_mixed_lanes += n_evict; ph_mixed_x += (fill_dead + n_evict < sel_count ? fill_dead + n_evict : sel_count); ph_kind[best & 3u] += 1ull;
#endif
                if (n_evict != 0u) {
                    uint32_t slot = 0u;
#pragma unroll
                    for (uint32_t j = 0; j < KQ_KINDS; j++) {
                        const unsigned long long m = __ballot(evict && kind == j);
                        if (evict && kind == j) slot = kc[j] + prefix_below(m);
                        kc[j] += (uint32_t)__popcll(m);
                    }
                    if (evict) kq_store(kind, slot, ps, item, prim, t);
                }
                const bool fill = evict || (dead && prefix_below(dmask) < fill_dead);
                const unsigned long long fmask = __ballot(fill);
                const uint32_t n_fill = fill_dead + n_evict, rank = prefix_below(fmask);
                const uint32_t n_pop = n_fill < sel_count ? n_fill : sel_count;
                if (fill && rank < n_pop) {
                    const uint4 *r = kq + (size_t)best * (KQ_FIELDS * KQ_CAP) + (sel_count - 1u - rank);
                    const uint4 f0 = r[0 * KQ_CAP], f1 = r[1 * KQ_CAP], f2 = r[2 * KQ_CAP], f4_ = r[4 * KQ_CAP], f5 = r[5 * KQ_CAP], f6 = r[6 * KQ_CAP];
                    ps.ray.o = v3(__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z)); ps.hero_nm = __uint_as_float(f0.w);
                    ps.ray.d = v3(__uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z)); ps.bounce = f1.w;
                    ps.thr = f4(__uint_as_float(f2.x), __uint_as_float(f2.y), __uint_as_float(f2.z), __uint_as_float(f2.w));
                    if (INTEG < 2) { const uint4 f3 = r[3 * KQ_CAP]; ps.ppdf = f4(__uint_as_float(f3.x), __uint_as_float(f3.y), __uint_as_float(f3.z), __uint_as_float(f3.w)); }
                    else ps.ppdf = splat(1.0f);
                    ps.rad = f4(__uint_as_float(f4_.x), __uint_as_float(f4_.y), __uint_as_float(f4_.z), __uint_as_float(f4_.w));
                    ps.smp.scramble = f5.x; ps.smp.index = f5.y; item = f5.z; ps.smp.draws = f5.w;
                    sampler_rewind(ps.smp);  // this lane's ring holds another path's values
                    prim = (int)f6.x; t = __uint_as_float(f6.y);
                    lane_state = LANE_HIT;
                }
#pragma unroll
                for (uint32_t j = 0; j < KQ_KINDS; j++) if (j == best) kc[j] -= n_pop;
                // the rest of the lanes to fill (Lambertian iterations only): primary hits from the queue
                const uint32_t n_q = n_fill - n_pop;
                if (fill && rank >= n_pop) {
                    const uint32_t e = (q_head + (rank - n_pop)) % PQ_CAP;
                    item = q[0][e];
                    CamSample cs;
                    cs.scramble = q[2][e]; cs.hero_nm = __uint_as_float(q[3][e]);
                    cs.d = v3(__uint_as_float(q[4][e]), __uint_as_float(q[5][e]), __uint_as_float(q[6][e]));
                    cs.u2 = __uint_as_float(q[9][e]); cs.u3 = __uint_as_float(q[10][e]);
                    path_begin(a.cam, cs, q[1][e], ring, ps);
                    t = __uint_as_float(q[7][e]);
                    prim = (int)q[8][e];
                    lane_state = LANE_HIT;
                }
                q_head = (q_head + n_q) % PQ_CAP;
                q_count -= n_q;
                // lanes that keep a hit of another kind (none of the chosen kind was left to take their place) sit this iteration out
                if (other && !evict) lane_state = LANE_HITX;
                // the records stored above are read by OTHER lanes of this wave in a later iteration: same wave, same L1, in program
                // order -- the fence keeps the compiler from moving memory operations across this point, the hardware needs nothing
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- refill: lanes without a path take a queued primary hit ----
        if (!SORTED || !sort_on) {
            const bool dead = lane_state == LANE_DEAD;
            const unsigned long long dmask = __ballot(dead);
            const uint32_t n_dead = (uint32_t)__popcll(dmask);
            const uint32_t n = n_dead < q_count ? n_dead : q_count;
            const uint32_t rank = prefix_below(dmask);
            if (dead && rank < n) {
                const uint32_t e = (q_head + rank) % PQ_CAP;
                item = q[0][e];
                CamSample cs;
                cs.scramble = q[2][e]; cs.hero_nm = __uint_as_float(q[3][e]);
                cs.d = v3(__uint_as_float(q[4][e]), __uint_as_float(q[5][e]), __uint_as_float(q[6][e]));
                cs.u2 = __uint_as_float(q[9][e]); cs.u3 = __uint_as_float(q[10][e]);
                path_begin(a.cam, cs, q[1][e], ring, ps);
                t = __uint_as_float(q[7][e]);
                prim = (int)q[8][e];
                lane_state = LANE_HIT;
            }
            q_head = (q_head + n) % PQ_CAP;
            q_count -= n;
            __builtin_amdgcn_wave_barrier();
        }
        IRIS_PHASE(2);
        // ---- shade ----
        const bool active = lane_state == LANE_HIT;
        if (SORTED && lane_state == LANE_HITX) lane_state = LANE_HIT;  // back in the count for the next iteration's choice
        if (__ballot(active) == 0ull) {
            // Nothing to shade.  Without work left that is the end: no lane holds a hit and the queue (SORTED: and every stack) is empty.
            // (POOLED: paths may still wait for rays that are out, walkers may hold parked rays: the next loop top moves them on.)
            if (!work_left && (!POOLED || (__ballot(lane_state != LANE_DEAD) == 0ull && __ballot(walking) == 0ull))) break;
            continue;  // every camera sample of the last generate phase missed: generate again
        }
#ifdef IRIS_PHASE_TIMERS
        ph_lanes += (unsigned long long)__popcll(__ballot(active));
        ph_iters += 1ull;
#endif
        const bool cont = shade_bounce<INTEG, CLS, POOLED>(a.sc, st, a.fr, sobol_lds, ring, active, ps, prim, t, cnt, &df);
        if (active) {
            if (POOLED) { own_e = cont ? 1u : 0u; own_s = (df.pending && df.want1) ? 1u : 0u; occluded1 = false; }
            if (cont) {
                lane_state = LANE_RAY;
            } else if (POOLED && df.pending) {
                lane_state = LANE_FIN;  // its direct-light term is added once its visibility query is back, then the result is stored
            } else if (POOLED) {
                store_result(a, item, path_xyz<INTEG>(ps));
                lane_state = LANE_DEAD;
            } else {
                lane_state = LANE_DONE;  // stored after the next extend phase, with the paths whose ray misses there
            }
        }
        IRIS_PHASE(3);
    }
#ifdef IRIS_PHASE_TIMERS
    if (lane_id() == 0u)
    {
        for (int k = 0; k < 4; k++) atomicAdd(&a.counters[4 + k], ph[k]);
        atomicAdd(&a.counters[8], ph_lanes);
        atomicAdd(&a.counters[9], ph_iters);
        atomicAdd(&a.counters[11], ph_mixed);
        atomicAdd(&a.counters[12], ph_mixed_lanes);
        atomicAdd(&a.counters[13], ph_mixed_x);
        for (int kk = 1; kk <= 3; kk++) atomicAdd(&a.counters[13 + kk], ph_kind[kk]);
    }
#endif
#undef IRIS_PHASE
    flush_counters(a, cnt);
    {
        unsigned long long x = n_sky;
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane_id() == 0u && x) atomicAdd(&a.counters[10], x);
    }
}

// ---- the queue kernels: shading and traversal as separate kernels over path-state and ray queues in HBM ----
// (hierarchy + Lambertian/constant class: the 10k-sphere configuration; north_star's "separate kernels over SoA ray/path-state
// queues in HBM").  k_regwave's pooled traversal shares 168 registers and three waves per SIMD with the shading code and runs at
// half of its lanes; here the traversal is a kernel of its own (k_wfq_trace: nothing but wfq_trace_wave, eight waves per SIMD,
// every lane taking the next ray of the queue when it has finished one) and so is everything else (k_wfq_shade):
//   * a path lives in a SLOT: 48 bytes of state (item, bounce + flags, draws, queue position of its extension ray | throughput,
//     path pdf, radiance, throughput before the last bounce | the two deferred direct-light contributions and the queue positions
//     of their visibility queries), three 16-byte arrays indexed by slot, read and written in slot order;
//   * k_wfq_shade visits every slot: a slot whose extension ray came back with a hit adds the deferred direct light of its last
//     bounce (apply_deferred: the reference's order of additions), shades the next bounce (shade_bounce<.., 2>: no traversal
//     code) and appends up to three rays to the ray queue -- extension ray, light-sample query, BSDF-sample query; a slot whose
//     path has ended (miss, absorption, roulette, depth) adds what was deferred, stores the sample's XYZ and starts the next camera
//     sample of the launch's item counter in the same lane; a tile's rays go, compacted, to the tile's own region of the queue;
//   * k_wfq_trace answers the queue: (t, primitive) per ray into the result array beside it, read by the next k_wfq_shade.
// Two ray queues alternate (pass p reads queue p & 1 and fills the other); the counters a pass needs at zero are cleared by the
// kernel before it.  The host launches trace/shade pairs in batches and reads the number of live paths after each batch; several
// pipelines (streams of their own, own slots and queues, one shared item counter) overlap each other's launch tails.
// Which slot traces an item changes nothing: every result is a function of the item and goes to the item's own result.
constexpr uint32_t WFQ_DEAD = 0u, WFQ_RAY = 1u, WFQ_FIN = 2u;  // slot state (bits 16-17 of word 1)
constexpr uint32_t WFQ_F_PENDING = 4u, WFQ_F_WANT1 = 8u, WFQ_F_LIM1BAD = 16u, WFQ_F_HAS2 = 32u, WFQ_F_SPEC2 = 64u;
constexpr uint32_t WFQ_NONE = 0xffffffffu;
enum : uint32_t { WFQ_C_NRAYS = 0u, WFQ_C_CURSOR = 2u, WFQ_C_LIVE = 4u, WFQ_C_WORDS = 8u };
static_assert(IRIS_WFQ_REGION == 3u * BLOCK, "a region of the ray queue holds the rays of one tile of slots: at most three per slot");
constexpr uint32_t WFQ_STACK_LDS = 8u;  // traversal-stack entries per lane kept in LDS by k_wfq_trace (8 KiB per workgroup)

struct WfqArgs {
    SceneDev sc;
    CamDev cam;
    FrameDev fr;
    uint4 *st_a;    // item, bounce | (state | flags) << 16, sampler draws consumed, queue position of the extension ray
    float4 *st_b;   // throughput, path pdf, radiance, throughput before the last bounce (one value for the four wavelengths)
    float4 *st_c;   // deferred contributions c1, c2; queue positions of the two visibility queries (bits; WFQ_NONE: none)
    float4 *rq_o[2], *rq_d[2];  // ray queues: (o, cull_t), (d, word) -- wfq_prepare
    uint2 *res[2];              // per queue position: (t bits, primitive or -1)
    uint32_t *n_region[2];      // rays in each region of IRIS_WFQ_REGION queue positions (region r = the rays of slots 256 r .. 256 r + 255)
    uint2 *wg_run;              // per k_wfq_shade workgroup: the unclaimed rest [next, end) of its current run of items, kept between passes
    uint32_t *ctr;              // WFQ_C_*: trace cursor [2], live paths [2]
    XyzResult *results;
    unsigned long long *counters, *work_cursor;
    uint32_t npix;
    FastDiv div_npix, div_width;
    uint32_t s_begin, n_items, n_slots, par, seed_pass, item_run;
    uint32_t *stack_hi;
    uint32_t stack_lds_n;
};

#ifndef IRIS_WFQ_TRACE_WAVES
#define IRIS_WFQ_TRACE_WAVES 8
#endif
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(IRIS_WFQ_TRACE_WAVES))) k_wfq_trace(WfqArgs a) {
    extern __shared__ float4 stack_win[];
    SphereStage st;
    st.lds = stack_win;
    st.cap = 0u;
    st.resident = false;
    st.bvh = true;
    st.stack = (uint32_t *)stack_win + threadIdx.x;
    st.stack_lds_n = a.stack_lds_n;
    st.stack_hi = a.stack_hi + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    st.stack_hi_stride = gridDim.x * BLOCK;
    const uint32_t par = a.par;
    if (blockIdx.x == 0u && threadIdx.x == 0u) {  // the cursor of the next k_wfq_trace, and what the next k_wfq_shade adds to
        a.ctr[WFQ_C_CURSOR + (par ^ 1u)] = 0u;
        a.ctr[WFQ_C_LIVE + (par ^ 1u)] = 0u;
    }
    unsigned long long tests = 0ull;
    wfq_trace_wave(a.sc, st, a.rq_o[par], a.rq_d[par], a.res[par], a.n_region[par], (a.n_slots + BLOCK - 1u) / BLOCK, &a.ctr[WFQ_C_CURSOR + par], tests);
    for (int off = 32; off > 0; off >>= 1) tests += __shfl_down(tests, off, 64);
    if (lane_id() == 0u && tests) atomicAdd(&a.counters[3], tests);
}

template <int INTEG>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(3))) k_wfq_shade(WfqArgs a) {
    __shared__ uint2 sobol_tab[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    __shared__ uint32_t wg_count[BLOCK / 64][4], wg_got[BLOCK / 64];
    __shared__ uint32_t run_next, run_lim, run_new;  // the workgroup's run of items
    stage_tables(sobol_tab);
    const SobolLds sobol_lds = sobol_tab;
    DrawRing ring = ring_lds + threadIdx.x;
    SphereStage st = {};
    st.bvh = true;  // shade_bounce<.., 2> makes no scan of its own
    const uint32_t par = a.par, nxt = a.par ^ 1u;
    const float4 *__restrict__ rq_o = a.rq_o[par], *__restrict__ rq_d = a.rq_d[par];
    const uint2 *__restrict__ res = a.res[par];
    Counters cnt = {0u, 0u, 0u, 0u};
    uint32_t n_live = 0u;
    const uint32_t n_tiles = (a.n_slots + BLOCK - 1u) / BLOCK, wave = threadIdx.x / 64u;
    const float inf = __builtin_inff();
    if (threadIdx.x == 0u) {
        const uint2 run = a.seed_pass ? make_uint2(0u, 0u) : a.wg_run[blockIdx.x];
        run_next = run.x; run_lim = run.y; run_new = 0u;
    }
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();  // the last tile's exchange has been read (and, the first time, the Sobol table and the run are in place)
        const uint32_t slot = tile * BLOCK + threadIdx.x;
        const bool in_range = slot < a.n_slots;
        uint4 A = make_uint4(0u, 0u, 0u, WFQ_NONE);
        float4 B = make_float4(0.0f, 1.0f, 0.0f, 0.0f), C = make_float4(0.0f, 0.0f, __uint_as_float(WFQ_NONE), __uint_as_float(WFQ_NONE));
        if (in_range && !a.seed_pass) { A = a.st_a[slot]; B = a.st_b[slot]; C = a.st_c[slot]; }  // three independent loads: one round trip
        const uint32_t flags = A.y >> 16, state = flags & 3u;
        const bool is_ray = state == WFQ_RAY, live_in = is_ray || state == WFQ_FIN;
        // ---- what the rays of the last pass found ----
        PathState ps;
        ps.ray.o = v3(0.0f, 0.0f, 0.0f); ps.ray.d = v3(0.0f, 0.0f, 1.0f); ps.hero_nm = LAMBDA_MIN_NM; ps.bounce = 0u;
        ps.thr = splat(0.0f); ps.ppdf = splat(1.0f); ps.rad = splat(0.0f);
        ps.smp = sampler_new(0u, 0u, 0u, 0u);
        float t = 0.0f;
        int prim = -1;
        bool occ1 = false, occ2 = false;
        uint32_t item = A.x;
        if (live_in) {
            if (is_ray) {
                const float4 ro = rq_o[A.w], rd = rq_d[A.w];
                const uint2 r = res[A.w];
                ps.ray.o = v3(ro.x, ro.y, ro.z);
                ps.ray.d = v3(rd.x, rd.y, rd.z);
                t = __uint_as_float(r.x);
                prim = (int)r.y;
            }
            const uint32_t k1 = __float_as_uint(C.z), k2 = __float_as_uint(C.w);
            if (k1 != WFQ_NONE) occ1 = (int)res[k1].y >= 0;
            if (k2 != WFQ_NONE) occ2 = (int)res[k2].y >= 0;
            // the path's identity: everything else about its sampler and wavelength follows from the item (tile.rs:145-147)
            const uint32_t s_local = fast_div(item, a.div_npix), pix = item - s_local * a.npix;
            const uint32_t y = fast_div(pix, a.div_width), x = pix - y * a.fr.width;
            ps.smp = sampler_new(x, y, a.s_begin + s_local, a.fr.seed);
            ps.hero_nm = gen_golden_ratio(ps.smp) * LAMBDA_RANGE_NM + LAMBDA_MIN_NM;  // wavelength.rs:26-28, as camera_sample
            ps.smp.draws = A.z;
            sampler_rewind(ps.smp);  // the ring of this lane holds another path's values
            ps.bounce = A.y & 0xffffu;
            ps.thr = splat(B.x); ps.ppdf = splat(B.y); ps.rad = splat(B.z);
            // the direct light of the bounce shaded in the last pass, now that its queries are back (hwss_naive.rs:61-62)
            Deferred prev = {};
            prev.pending = (flags & WFQ_F_PENDING) != 0u;
            prev.want1 = (flags & WFQ_F_WANT1) != 0u;
            prev.limit1 = (flags & WFQ_F_LIM1BAD) ? inf : 0.0f;
            prev.has2 = (flags & WFQ_F_HAS2) != 0u;
            prev.c1 = C.x; prev.c2 = C.y; prev.thr = B.w; prev.weight = (float)a.sc.n_lights;
            apply_deferred(prev, occ1, ps, occ2);
            if (prev.pending && prev.has2 && !occ2 && (flags & WFQ_F_SPEC2)) cnt.sphere_tests++;
        }
        // ---- shade the paths whose extension ray found a surface ----
        const bool hit = is_ray && prim >= 0;
        Deferred df = {};
        bool cont = false;
        if (__ballot(hit) != 0ull) cont = shade_bounce<INTEG, IRIS_CLS_LAMBERT_CONSTANT, 2>(a.sc, st, a.fr, sobol_lds, ring, hit, ps, prim, t, cnt, &df);
        const bool go_on = hit && cont;
        const bool ended_now = hit && !cont && !df.want1 && !df.has2;  // nothing of this bounce is out: its direct light is complete
        if (ended_now) apply_deferred(df, false, ps, false);
        // ---- finished paths: store the sample (tile.rs:159-162), then the lane starts the launch's next camera sample ----
        const bool ended = (live_in && !hit) || ended_now;
        if (ended) store_result(a, item, path_xyz<INTEG>(ps));
        const bool regen = ended || (in_range && a.seed_pass != 0u);
        const bool has_1 = hit && df.want1, has_2 = hit && df.has2;
        // ---- one exchange per tile through LDS: how many lanes of each wave want an item, and how many rays of each kind the
        // waves will send (the extension rays of the paths that go on are known; those of new paths are known once the items
        // are: a new path always sends one) ----
        const unsigned long long m_regen = __ballot(regen), m_on = __ballot(go_on), m_1 = __ballot(has_1), m_2 = __ballot(has_2);
        if (lane_id() == 0u) {
            wg_count[wave][0] = (uint32_t)__popcll(m_regen); wg_count[wave][1] = (uint32_t)__popcll(m_on);
            wg_count[wave][2] = (uint32_t)__popcll(m_1); wg_count[wave][3] = (uint32_t)__popcll(m_2);
        }
        __syncthreads();
        // items: the workgroup hands out a run of consecutive items claimed from the launch's counter with one atomic per run
        // (consecutive items = neighbouring pixels of one sample index); a tile that wants more than the run has left takes the
        // rest of it and starts the next
        uint32_t want_before = 0u, want_all = 0u;
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64u; w++) { if (w < wave) want_before += wg_count[w][0]; want_all += wg_count[w][0]; }
        const uint32_t old_next = run_next, old_left = run_lim - run_next;  // (LDS, stable between the barriers)
        if (want_all > old_left) {
            if (threadIdx.x == 0u) {
                const unsigned long long start = atomicAdd(a.work_cursor, (unsigned long long)a.item_run);
                run_new = start < (unsigned long long)a.n_items ? (uint32_t)start : a.n_items;  // past the end: an empty run at the end
            }
            __syncthreads();
        }
        bool got = false;
        Ray ext = ps.ray;
        if (regen) {
            const uint32_t r = want_before + prefix_below(m_regen);
            const uint32_t id = r < old_left ? old_next + r : run_new + (r - old_left);
            got = id < a.n_items && (r < old_left || r - old_left < a.item_run);
            if (got) {
                item = id;
                const uint32_t s_local = fast_div(item, a.div_npix), pix = item - s_local * a.npix;
                const uint32_t y = fast_div(pix, a.div_width), x = pix - y * a.fr.width;
                const CamSample cs = camera_sample(a.cam, a.fr, sobol_lds, x, y, a.s_begin + s_local);
                ext.o = v3(a.cam.px, a.cam.py, a.cam.pz);
                ext.d = cs.d;
                cnt.samples++;
                cnt.rays++;
            }
        }
        if (go_on) cnt.rays++;  // the extension ray (counted by the other kernels when it comes back)
        const unsigned long long m_got = __ballot(got);
        if (lane_id() == 0u) wg_got[wave] = (uint32_t)__popcll(m_got);
        __syncthreads();  // (also: every thread has read run_next / run_lim / run_new)
        if (threadIdx.x == 0u) {
            if (want_all > old_left) { run_next = min(run_new + (want_all - old_left), run_new + a.item_run); run_lim = run_new + a.item_run; if (run_new >= a.n_items) { run_next = run_lim = a.n_items; } }
            else run_next = old_next + want_all;
        }
        // ---- the rays of this pass: the spheres outside the tree and the root box here, with the whole wave busy ----
        const bool has_e = go_on || got;
        const unsigned long long m_e = m_on | m_got;
        float4 e_o = {}, e_d = {}, q1_o = {}, q1_d = {}, q2_o = {}, q2_d = {};
        bool e_ans = false, q1_ans = false, q2_ans = false;
        float e_t = 0.0f, q1_t = 0.0f, q2_t = 0.0f;
        int e_p = -1, q1_p = -1, q2_p = -1;
        if (has_e) wfq_prepare(a.sc, ext, false, inf, 0, e_o, e_d, e_ans, e_t, e_p, cnt.sphere_tests);
        if (has_1) wfq_prepare(a.sc, df.ray1, true, df.limit1, 0x7fffffff, q1_o, q1_d, q1_ans, q1_t, q1_p, cnt.sphere_tests);
        if (has_2) wfq_prepare(a.sc, df.ray2, true, df.limit2, df.tie2, q2_o, q2_d, q2_ans, q2_t, q2_p, cnt.sphere_tests);
        uint32_t k_e = WFQ_NONE, k_1 = WFQ_NONE, k_2 = WFQ_NONE;
        {   // queue order inside the tile's region: extension rays of waves 0..3, then the light-sample queries, then the others
            uint32_t pos = tile * IRIS_WFQ_REGION, mine[3] = {0u, 0u, 0u};
#pragma unroll
            for (uint32_t kind = 0; kind < 3; kind++)
#pragma unroll
                for (uint32_t w = 0; w < BLOCK / 64u; w++) {
                    if (w == wave) mine[kind] = pos;
                    pos += kind == 0u ? wg_count[w][1] + wg_got[w] : wg_count[w][kind + 1u];
                }
            if (has_e) k_e = mine[0] + prefix_below(m_e);
            if (has_1) k_1 = mine[1] + prefix_below(m_1);
            if (has_2) k_2 = mine[2] + prefix_below(m_2);
            if (threadIdx.x == 0u) a.n_region[nxt][tile] = pos - tile * IRIS_WFQ_REGION;
        }
        if (has_e) { a.rq_o[nxt][k_e] = e_o; a.rq_d[nxt][k_e] = e_d; if (e_ans) a.res[nxt][k_e] = make_uint2(__float_as_uint(e_t), (uint32_t)e_p); }
        if (has_1) { a.rq_o[nxt][k_1] = q1_o; a.rq_d[nxt][k_1] = q1_d; if (q1_ans) a.res[nxt][k_1] = make_uint2(__float_as_uint(q1_t), (uint32_t)q1_p); }
        if (has_2) { a.rq_o[nxt][k_2] = q2_o; a.rq_d[nxt][k_2] = q2_d; if (q2_ans) a.res[nxt][k_2] = make_uint2(__float_as_uint(q2_t), (uint32_t)q2_p); }
        // ---- the slot's new state ----
        if (live_in || regen) {
            uint32_t nstate = WFQ_DEAD, nflags = 0u, bounce = 0u, draws = 0u;
            float4 nB = make_float4(1.0f, 1.0f, 0.0f, 0.0f), nC = make_float4(0.0f, 0.0f, __uint_as_float(WFQ_NONE), __uint_as_float(WFQ_NONE));
            if (got) {
                nstate = WFQ_RAY;
                draws = 2u;  // the camera sample has used two values of set 0 (tile.rs:149-153); the ring is refilled from `draws`
            } else if (hit && !ended_now) {
                nstate = go_on ? WFQ_RAY : WFQ_FIN;
                nflags = (df.pending ? WFQ_F_PENDING : 0u) | (df.want1 ? WFQ_F_WANT1 : 0u) | (inf > df.limit1 ? 0u : WFQ_F_LIM1BAD) |
                         (df.has2 ? WFQ_F_HAS2 : 0u) | (df.spec_tests2 ? WFQ_F_SPEC2 : 0u);
                bounce = ps.bounce;
                draws = ps.smp.draws;
                nB = make_float4(ps.thr.v[0], ps.ppdf.v[0], ps.rad.v[0], df.thr);
                nC = make_float4(df.c1, df.c2, __uint_as_float(k_1), __uint_as_float(k_2));
            }
            a.st_a[slot] = make_uint4(item, bounce | ((nstate | nflags) << 16), draws, k_e);
            if (nstate != WFQ_DEAD) { a.st_b[slot] = nB; a.st_c[slot] = nC; n_live++; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0u) a.wg_run[blockIdx.x] = make_uint2(run_next, run_lim);
    flush_counters(a.counters, cnt);
    for (int off = 32; off > 0; off >>= 1) n_live += __shfl_down(n_live, off, 64);
    if (lane_id() == 0u && n_live) atomicAdd(&a.ctr[WFQ_C_LIVE + nxt], n_live);
}

// ---- the same two stages as separate kernels (profiling / north-star layout) ----
__global__ void __launch_bounds__(BLOCK) k_pool_init(WfArgs a) {
    const uint32_t slot0 = blockIdx.x * a.slots_per_block;
    for (uint32_t s = threadIdx.x; s < a.slots_per_block; s += BLOCK) a.pool.rayd[slot0 + s] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
    if (threadIdx.x == 0) {
        uint32_t b, e;
        block_range(a, b, e);
        a.blk_cursor[blockIdx.x] = b;
        a.blk_qn[blockIdx.x] = 0u;
    }
}
template <int INTEG>
__global__ void __launch_bounds__(BLOCK) k_stage_gen_extend(WfArgs a) {
    extern __shared__ float4 sphere_win[];
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    __shared__ BlockShared bs;
    stage_tables(sobol_lds);
    DrawRing ring = ring_lds + threadIdx.x;
    SphereStage st = stage_spheres(a.sc, sphere_win, a.sphere_cap);
    if (threadIdx.x == 0) {
        uint32_t b;
        block_range(a, b, bs.w_end);
        bs.cursor = a.blk_cursor[blockIdx.x];
        bs.qn = 0u;
    }
    __syncthreads();
    Counters cnt = {0u, 0u, 0u, 0u};
    stage_gen_extend<INTEG, IRIS_CLS_GENERAL_ENV>(a, st, sobol_lds, ring, bs, cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.blk_cursor[blockIdx.x] = bs.cursor < bs.w_end ? bs.cursor : bs.w_end;
        a.blk_qn[blockIdx.x] = bs.qn;
        if (bs.qn != 0u || bs.cursor < bs.w_end) atomicAdd(a.live_blocks, 1u);
    }
    flush_counters(a, cnt);
}
template <int INTEG>
__global__ void __launch_bounds__(BLOCK) k_stage_shade(WfArgs a) {
    extern __shared__ float4 sphere_win[];
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    const uint32_t qn = a.blk_qn[blockIdx.x];
    if (qn == 0u) return;
    DrawRing ring = ring_lds + threadIdx.x;
    stage_tables(sobol_lds);
    SphereStage st = stage_spheres(a.sc, sphere_win, a.sphere_cap);
    __syncthreads();
    Counters cnt = {0u, 0u, 0u, 0u};
    stage_shade<INTEG, IRIS_CLS_GENERAL_ENV>(a, st, sobol_lds, ring, qn, cnt);
    flush_counters(a, cnt);
}

// ---- sky mask: pixels none of whose camera samples can hit anything ----
// A camera sample that hits nothing returns exactly zero when the scene has no environment map (`None => break`,
// hwss_naive.rs:41-44, with radiance still 0), and a pixel all of whose samples do so is exactly black.  For the
// reference's default scene that is half of the film, and each of those samples still pays a Sobol set, the
// camera transform and a scan.  This pass decides per pixel, conservatively and in binary64, whether ANY ray
// through the pixel's jitter footprint (tile.rs:149-153: clip-space offsets in [0, 0.5/W) x [0, 0.5/H)) can touch
// any sphere: the footprint is bounded by a cone around its centre ray (half-angle = the largest angle to its four
// corners), each sphere by its cone of half-angle asin(r / dist), and the pixel is marked only if every pair of
// cones is separated by more than 2e-3 rad.  The f32 camera ray deviates from the exact one by ~1e-7 rad and the f32
// discriminant of sphere.rs:30 errs by ~2e-7 dist^2, against a miss margin of >= 4e-6 dist^2 implied by 2e-3 rad,
// so a marked pixel's samples all miss in the reference's arithmetic as well.  Anything unusual (camera inside or
// near a sphere, NaN) leaves the pixel unmarked, i.e. on the normal path.  Marked pixels cost nothing in the trace
// kernel and are written as zeros by the resolve; the work counters count their samples as the reference executes them.
__global__ void __launch_bounds__(BLOCK) k_sky_mask(CamDev cam, FrameDev fr, const float4 *spheres, uint32_t n_prims, unsigned char *sky) {
    const uint32_t pix = blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= fr.width * fr.height) return;
    const uint32_t y = pix / fr.width, x = pix - y * fr.width;
    // tile.rs:134-138 in the f32 arithmetic of camera_sample (exactly representable in binary64), then the footprint corners
    const float cxf = (((float)x + 0.5f) / (float)fr.width - 0.5f) * 2.0f;
    const float cyf = (((float)y + 0.5f) / (float)fr.height - 0.5f) * -2.0f;
    const double jx = 0.5 / (double)fr.width, jy = 0.5 / (double)fr.height;
    double dir[5][3];
    bool ok = true;
    for (int k = 0; k < 5; k++) {
        const double tx = (double)cxf + (k == 4 ? 0.5 * jx : ((k & 1) ? jx : 0.0));
        const double ty = (double)cyf + (k == 4 ? 0.5 * jy : ((k & 2) ? jy : 0.0));
        const double w = (double)cam.m[12] * tx + (double)cam.m[13] * ty + (double)cam.m[15];
        double v[3];
        for (int r = 0; r < 3; r++) v[r] = ((double)cam.m[4 * r] * tx + (double)cam.m[4 * r + 1] * ty + (double)cam.m[4 * r + 3]) / w;
        v[0] -= (double)cam.px; v[1] -= (double)cam.py; v[2] -= (double)cam.pz;
        const double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        ok = ok && len > 0.0 && len < 1e30 && w > 0.0;  // NaN fails every comparison
        for (int r = 0; r < 3; r++) dir[k][r] = v[r] / len;
    }
    double beta = 0.0;  // half-angle of the footprint cone around the centre ray dir[4]
    for (int k = 0; k < 4; k++) {
        const double c = dir[4][0] * dir[k][0] + dir[4][1] * dir[k][1] + dir[4][2] * dir[k][2];
        const double ang = acos(c > 1.0 ? 1.0 : c);
        ok = ok && ang >= 0.0 && ang < 0.2;
        beta = ang > beta ? ang : beta;
    }
    bool all_miss = ok;
    for (uint32_t i = 0; i < n_prims && all_miss; i++) {
        const float4 sp = spheres[i];
        const double vx = (double)sp.x - (double)cam.px, vy = (double)sp.y - (double)cam.py, vz = (double)sp.z - (double)cam.pz;
        const double dist = sqrt(vx * vx + vy * vy + vz * vz), r = fabs((double)sp.w);
        const bool outside = dist > r * 1.001 + 1e-6 && dist < 1e30;
        const double alpha = asin(outside ? r / dist : 1.0);
        const double c = (dir[4][0] * vx + dir[4][1] * vy + dir[4][2] * vz) / dist;
        const double theta = acos(c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c));
        all_miss = outside && theta > alpha + beta + 2e-3;
    }
    sky[pix] = all_miss ? 1 : 0;
}

// ---- film resolve: ordered per-pixel sum of the per-sample results ----
__global__ void __launch_bounds__(BLOCK) k_resolve(const XyzResult *__restrict__ results, float4 *__restrict__ film_acc, float *__restrict__ out,
                                                   uint32_t npix, uint32_t n_samples, int first, int last, float weight, const unsigned char *__restrict__ sky) {
    uint32_t pix = blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= npix) return;
    if (sky != nullptr && sky[pix] != 0) {  // every sample of the pixel is exactly zero and was never stored: 0 + 0 + ... = 0, 0 * weight = 0
        if (last) { out[3 * (size_t)pix + 0] = 0.0f; out[3 * (size_t)pix + 1] = 0.0f; out[3 * (size_t)pix + 2] = 0.0f; }
        else film_acc[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (!first) { float4 acc = film_acc[pix]; x = acc.x; y = acc.y; z = acc.z; }
#pragma unroll 8
    for (uint32_t s = 0; s < n_samples; s++) {  // tile.rs:159: xyz_sum += ..., ascending sample index
        XyzResult r = results[(size_t)s * npix + pix];
        x += r.x; y += r.y; z += r.z;
    }
    if (last) {  // tile.rs:165: xyz_sum * weight
        out[3 * (size_t)pix + 0] = x * weight;
        out[3 * (size_t)pix + 1] = y * weight;
        out[3 * (size_t)pix + 2] = z * weight;
    } else {
        film_acc[pix] = make_float4(x, y, z, 0.0f);
    }
}

// ---- stage-level debug kernels (parity tests) ----
__global__ void __launch_bounds__(BLOCK) k_dbg_sampler(const uint32_t *xys, uint32_t n, uint32_t seed, uint32_t ndraws, float *out) {
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    stage_tables(sobol_lds);
    __syncthreads();
    DrawRing ring = ring_lds + threadIdx.x;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    Sampler s = active ? sampler_new(xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], seed) : sampler_new(0u, 0u, 0u, 0u);
    for (uint32_t k = 0; k < ndraws; k++) {  // uneven `need`s exercise every ring phase
        ensure_draws(sobol_lds, ring, s, 1u + (k % 7u), active);
        if (active) out[(size_t)i * ndraws + k] = gen_0_1(ring, s);
    }
}
__global__ void __launch_bounds__(BLOCK) k_dbg_raygen(CamDev cam, FrameDev fr, const uint32_t *xys, uint32_t n, float *hero_out, float *o_out, float *d_out) {
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    stage_tables(sobol_lds);
    __syncthreads();
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    CamSample cs = camera_sample(cam, fr, sobol_lds, xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]);
    F4 wl = wavelength_new(cs.hero_nm);
    for (int k = 0; k < 4; k++) hero_out[4 * i + k] = wl.v[k];
    o_out[3 * i] = cam.px; o_out[3 * i + 1] = cam.py; o_out[3 * i + 2] = cam.pz;
    d_out[3 * i] = cs.d.x; d_out[3 * i + 1] = cs.d.y; d_out[3 * i + 2] = cs.d.z;
}
__global__ void __launch_bounds__(BLOCK) k_dbg_intersect(SceneDev sc, uint32_t cap, const float *o, const float *d, uint32_t n, int32_t *prim_out, float *t_out) {
    extern __shared__ float4 sphere_win[];
    SphereStage st = stage_spheres(sc, sphere_win, cap);
    __syncthreads();
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    bool active = i < n;
    Ray ray = {v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f)};
    if (active) { ray.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); ray.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]); }
    float t;
    int prim;
    unsigned long long tests = 0ull;
    scan_closest(sc, st, active, ray, t, prim, tests);
    if (active) { prim_out[i] = prim; t_out[i] = t; }
}
template <int INTEG>
__global__ void __launch_bounds__(BLOCK) k_dbg_trace(SceneDev sc, CamDev cam, FrameDev fr, uint32_t cap, const uint32_t *xys, uint32_t n, iris_trace *out) {
    extern __shared__ float4 sphere_win[];
    __shared__ uint2 sobol_lds[IRIS_SOBOL_LUT_ENTRIES];
    __shared__ float ring_lds[IRIS_RING_SLOTS * BLOCK];
    stage_tables(sobol_lds);
    DrawRing ring = ring_lds + threadIdx.x;
    SphereStage st = stage_spheres(sc, sphere_win, cap);
    __syncthreads();
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    bool alive = i < n;
    PathState ps;
    Counters cnt = {0u, 0u, 0u, 0u};
    iris_trace *tr = alive ? &out[i] : nullptr;
    if (alive) {
        raygen(cam, fr, sobol_lds, ring, xys[3 * i], xys[3 * i + 1], xys[3 * i + 2], ps);
        F4 wl = wavelength_new(ps.hero_nm);
        for (int k = 0; k < 4; k++) tr->hero[k] = wl.v[k];
        tr->ray_o[0] = ps.ray.o.x; tr->ray_o[1] = ps.ray.o.y; tr->ray_o[2] = ps.ray.o.z;
        tr->ray_d[0] = ps.ray.d.x; tr->ray_d[1] = ps.ray.d.y; tr->ray_d[2] = ps.ray.d.z;
        tr->n_hits = 0u;
    } else {
        ps.ray.o = v3(0.0f, 0.0f, 0.0f); ps.ray.d = v3(0.0f, 0.0f, 1.0f); ps.hero_nm = LAMBDA_MIN_NM; ps.bounce = 0u;
        ps.thr = splat(0.0f); ps.ppdf = splat(1.0f); ps.rad = splat(0.0f);
        ps.smp = sampler_new(0u, 0u, 0u, 0u);
    }
    uint32_t n_hits = 0u;
    if (fr.max_depth == 0u) alive = false;  // `for bounces in 0..0` never runs (hwss_naive.rs:40): no scan, no hit, zero radiance -- as iris_render
    while (__syncthreads_or(alive ? 1 : 0)) {
        float t;
        int prim;
        scan_closest(sc, st, alive, ps.ray, t, prim, cnt.sphere_tests);
        if (alive && prim < 0) {
            path_miss<INTEG, IRIS_CLS_GENERAL_ENV>(sc, ps);
            alive = false;
        }
        if (alive) {
            if (n_hits < IRIS_TRACE_MAX_DEPTH) { tr->hit_prim[n_hits] = prim; tr->hit_t[n_hits] = t; }
            n_hits++;
        }
        alive = shade_bounce<INTEG, IRIS_CLS_GENERAL_ENV>(sc, st, fr, sobol_lds, ring, alive, ps, prim, t, cnt);
    }
    if (tr) {
        V3 xyz = path_xyz<INTEG>(ps);
        F4 rad = (INTEG < 2) ? ps.rad : f4(hero(ps.rad), 0.0f, 0.0f, 0.0f);
        for (int k = 0; k < 4; k++) tr->radiance[k] = rad.v[k];
        tr->xyz[0] = xyz.x; tr->xyz[1] = xyz.y; tr->xyz[2] = xyz.z;
        tr->n_hits = n_hits < IRIS_TRACE_MAX_DEPTH ? n_hits : IRIS_TRACE_MAX_DEPTH;
        tr->draws = ps.smp.draws;
    }
}

}  // namespace

// =====================================================================================================
// Host side: the C ABI
// =====================================================================================================

// ---- the workspace of iris_render_device_noalloc (include/iris_hip.h) ----
// Device memory a render needs beyond the scene, its inputs and its output: the per-sample result buffer, the sky mask,
// the work counters and the launch's item counter, the second hit queue of the general scene class and the deep levels of
// the traversal stacks.  The allocating entry points hipMalloc these on demand and keep them in the context; the
// no-alloc entry point carves them out of this array instead, so that it never calls an allocator.  One copy of the array
// exists per GPU the library runs on (held until the process ends); in host code the name below is only a handle for
// hipGetSymbolAddress, never an address: everything uses the per-device pointer that scratch_for_device() returns.
#define IRIS_NOALLOC_SCRATCH_BYTES (64ull << 20)
static __device__ __align__(256) unsigned char scratch[IRIS_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {};
// The address of `scratch` on `device`, which must be the current device.  Looked up once per device.
hipError_t scratch_for_device(int device, unsigned char **out) {
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned char *)ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace


struct iris_ctx {
    int device = 0;
    int n_cus = 0;
    std::string err;
    // options
    int64_t blocks_per_cu = 0;    // 0: as many persistent workgroups per CU as are resident at once (occupancy query)
    int64_t slots_per_block = 1024;
    int64_t result_bytes = 64ll << 30;
    int64_t multi_kernel = 0;
    int64_t mode = 0;             // 0 auto, 1 pool/queue kernels, 2 register-resident kernel
    int64_t reg_blocks_per_cu = 8;   // register-resident kernel: at least as many workgroups as a CU can hold (8 waves per SIMD);
                                     // items are claimed from one global counter, so workgroups beyond the resident ones find nothing left
    int64_t bvh_min_prims = 65;   // build a hierarchy for scenes with at least this many spheres (0: never)
    int64_t bvh_leaf = 2;         // spheres per leaf of the hierarchy (1..7); config D: 2 -> 142.5 ms, 4 -> 145.1, 7 -> 151.2; the general classes measure alike
    int64_t specialize = 1;       // 1: use the kernel compiled for the scene's class; 0: always the general kernel
    int scene_class = IRIS_CLS_GENERAL;
    // scene
    bool have_scene = false;
    bool scene_is_lambert_constant = false;
    SceneDev sc = {};
    void *d_spheres = nullptr, *d_prim_info = nullptr, *d_bsdfs = nullptr, *d_spectra = nullptr, *d_lights = nullptr, *d_spec_over_pi = nullptr, *d_prim_kind = nullptr, *d_light_rec = nullptr;
    uint32_t *d_stack_hi = nullptr, *d_park = nullptr;
    uint4 *d_kq = nullptr;
    size_t kq_items = 0;
    int64_t sort_kinds = 1;   // general classes: sort hits by BSDF kind through per-wave stacks in global memory (k_regwave's SORTED)
    int n_bsdf_kinds = 0;     // distinct kinds among the uploaded scene's primitives
    size_t park_words = 0;
    unsigned char *d_sky = nullptr;
    size_t sky_cap = 0;
    unsigned long long *d_work_cursor = nullptr;  // register kernel: the launch's item counter
    int64_t work_run = 0;               // items per claim (0: chosen per launch)
    int64_t sky_mask = 1;
    size_t stack_hi_words = 0;
    void *d_bvh_nodes = nullptr, *d_bvh_spheres = nullptr, *d_bvh_prim = nullptr, *d_bvh_q = nullptr;
    void *d_env_texels = nullptr, *d_env_func = nullptr, *d_env_cond = nullptr, *d_env_marg = nullptr;
    float *d_up_coeff = nullptr, *d_up_scale = nullptr;  // UpsampleTable resident in HBM (iris_upsample_table_upload)
    uint32_t up_res = 0;
    uint32_t bvh_depth = 0;
    // pools
    uint32_t pool_blocks = 0, pool_slots_per_block = 0;
    Pool pool = {};
    uint32_t *d_queue = nullptr, *d_blk_cursor = nullptr, *d_blk_qn = nullptr, *d_live = nullptr;
    unsigned long long *d_counters = nullptr;
    XyzResult *d_results = nullptr;
    size_t results_cap = 0;  // items
    float4 *d_film_acc = nullptr;
    size_t film_cap = 0;
    uint32_t *h_live = nullptr;  // pinned
    unsigned long long *h_counters = nullptr;  // pinned: the work counters of a render are copied here
    unsigned char *scratch_ptr = nullptr;  // this device's copy of the static workspace (iris_render_device_noalloc)
    // queue kernels (k_wfq_*): up to WFQ_MAX_PIPES independent pipelines, each with its own slots, ray queues, counters and stream
    // Measured on config D (ms per launch): 2 pipelines x 2 Mi slots, 8 + 4 workgroups per CU 121.1; 3 x 4 Mi, 3 + 2 112.9 (3 x 8 Mi 120.6, 4 x 4 Mi
    // 133.1, shade 1 per CU 119.4, trace 4 per CU 121.5): fewer resident workgroups per kernel leave room for another pipeline's
    // kernel on the same CU, which is what overlaps the latency-bound shade passes with the traversal
    int64_t wfq_slots = 12 << 20, wfq_pipes = 3, wfq_batch = 8, wfq_shade_per_cu = 2, wfq_trace_per_cu = 3, wfq_item_run = 1024;
    struct WfqPipe {
        uint4 *st_a = nullptr;
        float4 *st_b = nullptr, *st_c = nullptr, *rq_o[2] = {nullptr, nullptr}, *rq_d[2] = {nullptr, nullptr};
        uint2 *res[2] = {nullptr, nullptr};
        uint32_t *ctr = nullptr, *stack_hi = nullptr, *h_ctr = nullptr, *n_region[2] = {nullptr, nullptr};  // h_ctr: pinned
        uint2 *wg_run = nullptr;
        size_t slots_cap = 0, stack_hi_words = 0;
        hipStream_t stream = nullptr;
        hipEvent_t ev = nullptr;
    } wfq[4];
    hipEvent_t wfq_ev_start = nullptr;
    bool wfq_now = false;  // this render goes through the queue kernels
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    iris_stats stats = {};
    uint64_t invariants[4] = {0, 0, 0, 0};  // debug build: the invariant counters of the last render
};

namespace {

constexpr int WFQ_MAX_PIPES = 4;
void free_wfq(iris_ctx *c);
constexpr uint32_t SPHERE_CAP = 1024;  // LDS window: 16 KiB of float4 (three pool-kernel workgroups per CU stay resident)

// The host-side hierarchy build lives in iris_bvh_build.cpp (plain C++: it is also compiled, with sanitizers, into the
// CPU test driver tests/host_san/).
using iris_host::Bvh;
using iris_host::build_bvh;
static_assert(sizeof(iris_host::Float4) == sizeof(float4), "the builder's records are uploaded as float4");

// one correctly rounded f32 division on the host (what the kernels' `/` is: -ffp-contract=off, IEEE divide)
float iris_f32_div(float a, float b) {
    volatile float x = a, y = b;
    return x / y;
}

int fail(iris_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                                     \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(ctx, IRIS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dev_alloc(iris_ctx *ctx, T **p, size_t n) {
    HIP_TRY(ctx, hipMalloc((void **)p, n * sizeof(T)));
    return IRIS_OK;
}
void dev_free(void *p) {
    if (p) (void)hipFree(p);
}

void free_pools(iris_ctx *c) {
    dev_free(c->pool.rayo); dev_free(c->pool.rayd); dev_free(c->pool.thr); dev_free(c->pool.ppdf); dev_free(c->pool.rad);
    dev_free(c->pool.meta); dev_free(c->pool.hit);
    dev_free(c->d_queue); dev_free(c->d_blk_cursor); dev_free(c->d_blk_qn);
    c->pool = Pool{};
    c->d_queue = c->d_blk_cursor = c->d_blk_qn = nullptr;
    c->pool_blocks = c->pool_slots_per_block = 0;
}

int ensure_pools(iris_ctx *c, uint32_t blocks, uint32_t spb) {
    if (c->pool_blocks == blocks && c->pool_slots_per_block == spb) return IRIS_OK;
    free_pools(c);
    size_t n = (size_t)blocks * spb;
    int rc;
    if ((rc = dev_alloc(c, &c->pool.rayo, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.rayd, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.thr, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.ppdf, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.rad, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.meta, n))) return rc;
    if ((rc = dev_alloc(c, &c->pool.hit, n))) return rc;
    if ((rc = dev_alloc(c, &c->d_queue, n))) return rc;
    if ((rc = dev_alloc(c, &c->d_blk_cursor, (size_t)blocks))) return rc;
    if ((rc = dev_alloc(c, &c->d_blk_qn, (size_t)blocks))) return rc;
    c->pool_blocks = blocks;
    c->pool_slots_per_block = spb;
    return IRIS_OK;
}

CamDev cam_dev(const iris_camera *cam) {
    CamDev c;
    c.px = cam->position[0]; c.py = cam->position[1]; c.pz = cam->position[2];
    std::memcpy(c.m, cam->clip_to_world, sizeof c.m);
    return c;
}

int check_params(iris_ctx *ctx, const iris_render_params *p) {
    if (!p) return fail(ctx, IRIS_E_ARG, "params is NULL");
    if (p->width == 0 || p->height == 0) return fail(ctx, IRIS_E_ARG, "width and height must be positive");
    if ((uint64_t)p->width * p->height > (1ull << 31)) return fail(ctx, IRIS_E_LIMIT, "more than 2^31 pixels");
    if (p->spp_total == 0) return fail(ctx, IRIS_E_ARG, "spp_total must be positive");
    if (p->integrator > IRIS_INT_SWSS_SLOW) return fail(ctx, IRIS_E_ARG, "unknown integrator");
    if (p->max_depth > 0xffffu) return fail(ctx, IRIS_E_LIMIT, "max_depth above 65535");
    if ((uint64_t)p->spp_begin + p->spp_count > 0xffffffffull) return fail(ctx, IRIS_E_ARG, "sample index overflows u32");
    return IRIS_OK;
}
int check_scene_for(iris_ctx *ctx, const iris_render_params *p) {
    if (!ctx->have_scene) return fail(ctx, IRIS_E_STATE, "no scene uploaded");
    bool naive = p->integrator == IRIS_INT_HWSS_NAIVE || p->integrator == IRIS_INT_SWSS_NAIVE;
    // pick_one_light on an empty light list indexes out of bounds in the reference (scene.rs:168-172)
    if (naive && ctx->sc.n_lights == 0 && ctx->sc.env_w == 0) return fail(ctx, IRIS_E_ARG, "next-event estimation needs at least one light");
    return IRIS_OK;
}
FrameDev frame_dev(const iris_render_params *p) {
    FrameDev f = {p->width, p->height, p->seed, p->max_depth, p->min_depth};
    return f;
}
// window capacity in spheres: the list rounded up to whole scan batches, at most SPHERE_CAP; with a hierarchy
// the same LDS holds the traversal stacks (IRIS_BVH_STACK dwords per lane)
uint32_t sphere_cap_for(const iris_ctx *c) {
    if (c->sc.n_bvh_nodes) return std::max<uint32_t>(c->bvh_depth, 1u) * BLOCK * sizeof(uint32_t) / sizeof(float4);  // stack entries <= depth - 1
    return std::min<uint32_t>((std::max<uint32_t>(c->sc.n_prims, 1u) + IRIS_SCAN_PAD - 1u) & ~(IRIS_SCAN_PAD - 1u), SPHERE_CAP);
}
static_assert(SPHERE_CAP % IRIS_SCAN_PAD == 0, "the LDS window holds whole scan batches");

// The register-resident kernel needs the whole sphere list in the LDS window or a hierarchy; it is the default
// for short lists (where shading, not the scans, is the cost) and for scenes with a hierarchy; the pool/queue
// kernels serve long lists scanned sphere by sphere (where compacting rays before every scan pays).
constexpr uint32_t REGISTER_KERNEL_MAX_PRIMS = 64;
constexpr uint32_t REG_STACK_LDS = 8;  // traversal-stack entries per lane the register kernel keeps in LDS
bool use_register_kernel(const iris_ctx *c) {
    if (c->multi_kernel || c->mode == 1) return false;
    if (c->sc.n_bvh_nodes) return true;  // no LDS window needed: the register-resident kernel is the faster one (measured)
    if (c->sc.n_prims > SPHERE_CAP) return false;
    return c->mode == 2 || c->sc.n_prims <= REGISTER_KERNEL_MAX_PRIMS;
}

// The queue kernels serve scenes with a hierarchy whose BSDFs are Lambertian over constant spectra (one value for the four
// wavelengths: 48 bytes of path state); mode 0 chooses them there, mode 2 keeps the register-resident kernel.
bool use_queue_kernels(const iris_ctx *c) {
    if (IRIS_DEBUG_INVARIANTS) return false;  // the invariant counters live in the kernels that shade without deferring
    if (c->multi_kernel || !(c->mode == 0 || c->mode == 3)) return false;  // mode 0: config D 112.9 ms against 118.8 ms for k_regwave's pooled traversal
    return c->sc.n_bvh_nodes != 0u && c->scene_class == IRIS_CLS_LAMBERT_CONSTANT;
}

void free_wfq(iris_ctx *c) {
    for (auto &p : c->wfq) {
        dev_free(p.st_a); dev_free(p.st_b); dev_free(p.st_c);
        for (int k = 0; k < 2; k++) { dev_free(p.rq_o[k]); dev_free(p.rq_d[k]); dev_free(p.res[k]); dev_free(p.n_region[k]); }
        dev_free(p.ctr); dev_free(p.stack_hi); dev_free(p.wg_run);
        if (p.h_ctr) (void)hipHostFree(p.h_ctr);
        if (p.stream) (void)hipStreamDestroy(p.stream);
        if (p.ev) (void)hipEventDestroy(p.ev);
        p = iris_ctx::WfqPipe{};
    }
    if (c->wfq_ev_start) (void)hipEventDestroy(c->wfq_ev_start);
    c->wfq_ev_start = nullptr;
}

int ensure_wfq(iris_ctx *c, int n_pipes, size_t slots, size_t stack_hi_words) {
    int rc;
    if (!c->wfq_ev_start) HIP_TRY(c, hipEventCreateWithFlags(&c->wfq_ev_start, hipEventDisableTiming));
    for (int i = 0; i < n_pipes; i++) {
        auto &p = c->wfq[i];
        if (!p.stream) HIP_TRY(c, hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        if (!p.ev) HIP_TRY(c, hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
        if (!p.ctr && (rc = dev_alloc(c, &p.ctr, (size_t)WFQ_C_WORDS))) return rc;
        if (!p.h_ctr) HIP_TRY(c, hipHostMalloc((void **)&p.h_ctr, WFQ_C_WORDS * sizeof(uint32_t), hipHostMallocDefault));
        if (p.slots_cap < slots) {
            dev_free(p.st_a); dev_free(p.st_b); dev_free(p.st_c);
            p.st_a = nullptr; p.st_b = p.st_c = nullptr;
            for (int k = 0; k < 2; k++) { dev_free(p.rq_o[k]); dev_free(p.rq_d[k]); dev_free(p.res[k]); dev_free(p.n_region[k]); p.rq_o[k] = p.rq_d[k] = nullptr; p.res[k] = nullptr; p.n_region[k] = nullptr; }
            dev_free(p.wg_run);
            p.wg_run = nullptr;
            p.slots_cap = 0;
            if ((rc = dev_alloc(c, &p.st_a, slots)) || (rc = dev_alloc(c, &p.st_b, slots)) || (rc = dev_alloc(c, &p.st_c, slots))) return rc;
            for (int k = 0; k < 2; k++)  // a slot sends at most three rays per pass
                if ((rc = dev_alloc(c, &p.rq_o[k], 3 * slots)) || (rc = dev_alloc(c, &p.rq_d[k], 3 * slots)) || (rc = dev_alloc(c, &p.res[k], 3 * slots)) ||
                    (rc = dev_alloc(c, &p.n_region[k], slots / BLOCK)))
                    return rc;
            if ((rc = dev_alloc(c, &p.wg_run, slots / BLOCK))) return rc;  // one per workgroup of k_wfq_shade: at most one per tile
            p.slots_cap = slots;
        }
        if (p.stack_hi_words < stack_hi_words) {
            dev_free(p.stack_hi);
            p.stack_hi = nullptr;
            p.stack_hi_words = 0;
            if ((rc = dev_alloc(c, &p.stack_hi, stack_hi_words))) return rc;
            p.stack_hi_words = stack_hi_words;
        }
    }
    return IRIS_OK;
}

// One chunk of a render through the queue kernels.  `a`: the chunk as launch_chunk gets it (scene, camera, items, result buffer).
template <int INTEG>
int run_wfq(iris_ctx *c, const WfArgs &a, hipStream_t stream) {
    const uint32_t tiles_needed = (a.n_items + BLOCK - 1u) / BLOCK;
    int n_pipes = (int)std::min<int64_t>(c->wfq_pipes, WFQ_MAX_PIPES);
    while (n_pipes > 1 && tiles_needed < (uint32_t)n_pipes * 64u) n_pipes--;  // small renders: one pipeline
    const size_t cap = std::max<size_t>(BLOCK, ((size_t)c->wfq_slots / n_pipes) & ~(size_t)(BLOCK - 1));
    const size_t slots = std::min<size_t>(cap, (size_t)((tiles_needed + n_pipes - 1) / n_pipes) * BLOCK);
    const uint32_t trace_blocks = (uint32_t)c->n_cus * (uint32_t)c->wfq_trace_per_cu, lds_n = std::min<uint32_t>(std::max<uint32_t>(c->bvh_depth, 1u), WFQ_STACK_LDS);
    int shade_per_cu = 0;  // the shade kernel deals its tiles in advance: exactly the workgroups that are resident at once
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&shade_per_cu, k_wfq_shade<INTEG>, BLOCK, 0) != hipSuccess || shade_per_cu < 1) shade_per_cu = 3;
    const uint32_t shade_blocks = std::min<uint32_t>((uint32_t)(slots / BLOCK), (uint32_t)c->n_cus * (uint32_t)std::min<int64_t>(shade_per_cu, c->wfq_shade_per_cu));
    const size_t stack_hi_words = c->bvh_depth > lds_n ? (size_t)trace_blocks * BLOCK * (c->bvh_depth - lds_n) : 1;
    int rc;
    if ((rc = ensure_wfq(c, n_pipes, slots, stack_hi_words))) return rc;
    WfqArgs q[WFQ_MAX_PIPES];
    HIP_TRY(c, hipEventRecord(c->wfq_ev_start, stream));
    for (int i = 0; i < n_pipes; i++) {
        auto &p = c->wfq[i];
        WfqArgs &w = q[i];
        w.sc = a.sc; w.cam = a.cam; w.fr = a.fr;
        w.st_a = p.st_a; w.st_b = p.st_b; w.st_c = p.st_c;
        for (int k = 0; k < 2; k++) { w.rq_o[k] = p.rq_o[k]; w.rq_d[k] = p.rq_d[k]; w.res[k] = p.res[k]; }
        w.ctr = p.ctr;
        w.results = a.results;
        w.counters = a.counters;
        w.work_cursor = a.work_cursor;
        w.npix = a.npix; w.div_npix = a.div_npix; w.div_width = a.div_width;
        w.s_begin = a.s_begin; w.n_items = a.n_items; w.n_slots = (uint32_t)slots;
        w.par = 0u; w.seed_pass = 1u; w.item_run = (uint32_t)c->wfq_item_run;
        w.n_region[0] = p.n_region[0]; w.n_region[1] = p.n_region[1]; w.wg_run = p.wg_run;
        w.stack_hi = p.stack_hi;
        w.stack_lds_n = lds_n;
        HIP_TRY(c, hipStreamWaitEvent(p.stream, c->wfq_ev_start, 0));
        HIP_TRY(c, hipMemsetAsync(p.ctr, 0, WFQ_C_WORDS * sizeof(uint32_t), p.stream));
    }
    uint32_t pass[WFQ_MAX_PIPES] = {0u, 0u, 0u, 0u};
    bool active[WFQ_MAX_PIPES] = {false, false, false, false};
    auto enqueue = [&](int i) -> int {  // one batch of passes on pipeline i, then its counters to the host
        auto &p = c->wfq[i];
        WfqArgs &w = q[i];
        for (int64_t b = 0; b < c->wfq_batch; b++) {
            w.par = pass[i] & 1u;
            w.seed_pass = pass[i] == 0u ? 1u : 0u;
            if (pass[i] != 0u) {
                hipLaunchKernelGGL(k_wfq_trace, dim3(trace_blocks), dim3(BLOCK), (size_t)lds_n * BLOCK * sizeof(uint32_t), p.stream, w);
                c->stats.launches += 1;
            }
            hipLaunchKernelGGL(k_wfq_shade<INTEG>, dim3(shade_blocks), dim3(BLOCK), 0, p.stream, w);
            c->stats.launches += 1;
            pass[i]++;
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(p.h_ctr, p.ctr, WFQ_C_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, p.stream));
        return IRIS_OK;
    };
    for (int i = 0; i < n_pipes; i++) {
        active[i] = true;
        if ((rc = enqueue(i))) return rc;
    }
    for (int left = n_pipes; left > 0;) {
        for (int i = 0; i < n_pipes; i++) {
            if (!active[i]) continue;
            HIP_TRY(c, hipStreamSynchronize(c->wfq[i].stream));
            if (c->wfq[i].h_ctr[WFQ_C_LIVE + (pass[i] & 1u)] == 0u) {  // no path left in this pipeline (and the item counter is past the end)
                active[i] = false;
                left--;
            } else if ((rc = enqueue(i))) {
                return rc;
            }
        }
    }
    for (int i = 0; i < n_pipes; i++) {
        HIP_TRY(c, hipEventRecord(c->wfq[i].ev, c->wfq[i].stream));
        HIP_TRY(c, hipStreamWaitEvent(stream, c->wfq[i].ev, 0));
    }
    return IRIS_OK;
}

// Workgroups of the fused pool/queue kernel that one CU holds at a time (registers, LDS).
template <int INTEG>
int pool_blocks_per_cu(const iris_ctx *c, size_t dyn_lds) {
    int n = 0;
    hipError_t e = c->scene_class == IRIS_CLS_LAMBERT_CONSTANT
                       ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wavefront<INTEG, IRIS_CLS_LAMBERT_CONSTANT>, BLOCK, dyn_lds)
                       : (c->scene_class == IRIS_CLS_GENERAL_ENV
                              ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wavefront<INTEG, IRIS_CLS_GENERAL_ENV>, BLOCK, dyn_lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wavefront<INTEG, IRIS_CLS_GENERAL>, BLOCK, dyn_lds));
    return (e == hipSuccess && n >= 1) ? n : 2;
}

template <int INTEG>
int launch_chunk(iris_ctx *c, const WfArgs &a, uint32_t blocks, hipStream_t stream) {
    size_t dyn = (size_t)a.sphere_cap * sizeof(float4);
    const bool lc = c->scene_class == IRIS_CLS_LAMBERT_CONSTANT, env = c->scene_class == IRIS_CLS_GENERAL_ENV;
    if (c->wfq_now) return run_wfq<INTEG>(c, a, stream);
    if (use_register_kernel(c)) {
        const bool bvh = c->sc.n_bvh_nodes != 0u;
#define IRIS_LAUNCH_REG(CLS_)                                                                                                   \
    do {                                                                                                                        \
        if (bvh) hipLaunchKernelGGL((k_regwave<INTEG, CLS_, true>), dim3(blocks), dim3(BLOCK), dyn, stream, a);                 \
        else hipLaunchKernelGGL((k_regwave<INTEG, CLS_, false>), dim3(blocks), dim3(BLOCK), dyn, stream, a);                    \
    } while (0)
        if (lc) IRIS_LAUNCH_REG(IRIS_CLS_LAMBERT_CONSTANT);
        else if (env) IRIS_LAUNCH_REG(IRIS_CLS_GENERAL_ENV);
        else IRIS_LAUNCH_REG(IRIS_CLS_GENERAL);
#undef IRIS_LAUNCH_REG
        HIP_TRY(c, hipGetLastError());
        c->stats.launches += 1;
        return IRIS_OK;
    }
    if (!c->multi_kernel) {
        if (lc) hipLaunchKernelGGL((k_wavefront<INTEG, IRIS_CLS_LAMBERT_CONSTANT>), dim3(blocks), dim3(BLOCK), dyn, stream, a);
        else if (env) hipLaunchKernelGGL((k_wavefront<INTEG, IRIS_CLS_GENERAL_ENV>), dim3(blocks), dim3(BLOCK), dyn, stream, a);
        else hipLaunchKernelGGL((k_wavefront<INTEG, IRIS_CLS_GENERAL>), dim3(blocks), dim3(BLOCK), dyn, stream, a);
        HIP_TRY(c, hipGetLastError());
        c->stats.launches += 1;
        return IRIS_OK;
    }
    hipLaunchKernelGGL(k_pool_init, dim3(blocks), dim3(BLOCK), 0, stream, a);
    HIP_TRY(c, hipGetLastError());
    // Launch stage pairs in batches; after each batch read back how many workgroups still had work.
    const int BATCH = 8;
    for (;;) {
        for (int it = 0; it < BATCH; it++) {
            HIP_TRY(c, hipMemsetAsync(c->d_live, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_stage_gen_extend<INTEG>, dim3(blocks), dim3(BLOCK), dyn, stream, a);
            hipLaunchKernelGGL(k_stage_shade<INTEG>, dim3(blocks), dim3(BLOCK), dyn, stream, a);
            c->stats.launches += 2;
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(c->h_live, c->d_live, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        if (*c->h_live == 0u) break;
    }
    return IRIS_OK;
}

// Device memory of one render beyond scene, inputs and output.
struct Workspace {
    XyzResult *results = nullptr;
    float4 *film_acc = nullptr;        // running per-pixel sums between sample-range chunks (chunked renders only)
    unsigned char *sky = nullptr;
    unsigned long long *counters = nullptr, *work_cursor = nullptr;
    uint32_t *stack_hi = nullptr, *park = nullptr;
    uint4 *kq = nullptr;
};

int render_impl(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, float *d_out, hipStream_t stream, bool no_alloc = false) {
    int rc;
    if ((rc = check_params(c, p))) return rc;
    if ((rc = check_scene_for(c, p))) return rc;
    if (!cam || !d_out) return fail(c, IRIS_E_ARG, "camera or output is NULL");
    for (int k = 0; k < 3; k++)
        if (!(std::fabs(cam->position[k]) <= 0x1p60f)) return fail(c, IRIS_E_LIMIT, "camera position must be finite and at most 2^60 in magnitude");
    for (int k = 0; k < 16; k++)
        if (!std::isfinite(cam->clip_to_world[k])) return fail(c, IRIS_E_ARG, "camera matrix has a non-finite entry");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t npix = p->width * p->height;
    c->stats = iris_stats{};
    const bool reg = use_register_kernel(c);
    if (no_alloc && !reg) return fail(c, IRIS_E_LIMIT, "iris_render_device_noalloc serves the scenes of the register-resident kernel only (up to 64 spheres, or a hierarchy)");
    const bool wfq = !no_alloc && use_queue_kernels(c);  // the queue kernels keep gigabytes of queues and synchronise with the host: not for the no-alloc entry
    if (c->mode == 3 && !wfq) return fail(c, IRIS_E_ARG, "mode 3 (queue kernels) needs a scene with a hierarchy whose BSDFs are all Lambertian over constant spectra");
    c->wfq_now = wfq;
    int64_t per_cu = reg ? c->reg_blocks_per_cu : c->blocks_per_cu;
    if (no_alloc) per_cu = std::min<int64_t>(per_cu, 4);  // a CU holds at most 4 of these workgroups; a smaller grid keeps the per-wave queues small
    const bool pooled = reg && c->sc.n_bvh_nodes && c->scene_class == IRIS_CLS_LAMBERT_CONSTANT;  // k_regwave<.., LAMBERT_CONSTANT, true>: the ray pool
    if (pooled) per_cu = std::min<int64_t>(per_cu, no_alloc ? 3 : 4);  // three are resident; every lane of the grid owns a parking area in global memory (the static workspace pays for residents only)
    // k_regwave's SORTED: stacks of path records per BSDF kind, when the scene holds more than one kind
    const bool sorted = reg && c->scene_class != IRIS_CLS_LAMBERT_CONSTANT && c->sort_kinds && c->n_bsdf_kinds > 1 && !no_alloc;
    if (sorted) per_cu = std::min<int64_t>(per_cu, 4);    // three are resident; every wave of the grid owns its stacks in global memory
    if (per_cu == 0) {
        const size_t dyn = (size_t)sphere_cap_for(c) * sizeof(float4);
        switch (p->integrator) {
        case IRIS_INT_HWSS_NAIVE: per_cu = pool_blocks_per_cu<IRIS_INT_HWSS_NAIVE>(c, dyn); break;
#ifndef IRIS_EXPERIMENT_INTEG0  // experiment builds (python -m iris_amd.build --integ0 -o ...): HwssNaive kernels only
        case IRIS_INT_HWSS_SLOW: per_cu = pool_blocks_per_cu<IRIS_INT_HWSS_SLOW>(c, dyn); break;
        case IRIS_INT_SWSS_NAIVE: per_cu = pool_blocks_per_cu<IRIS_INT_SWSS_NAIVE>(c, dyn); break;
        default: per_cu = pool_blocks_per_cu<IRIS_INT_SWSS_SLOW>(c, dyn); break;
#else
        default: return fail(c, IRIS_E_ARG, "this experiment build holds the HwssNaive kernels only");
#endif
        }
    }
    const uint32_t blocks = (uint32_t)(c->n_cus * per_cu);
    const uint32_t spb = (uint32_t)c->slots_per_block;
    const uint32_t reg_stack_lds = std::min<uint32_t>(std::max<uint32_t>(c->bvh_depth, 1u), REG_STACK_LDS);
    // sky mask: register kernel on a short list without an environment map
    const bool use_sky = c->sky_mask && reg && !c->sc.n_bvh_nodes && c->sc.env_w == 0u && p->spp_count != 0u && p->max_depth != 0u;
    const size_t stack_hi_words = (reg && !wfq && c->sc.n_bvh_nodes && c->bvh_depth > reg_stack_lds) ? (size_t)blocks * BLOCK * (c->bvh_depth - reg_stack_lds) : 0;
    const size_t kq_items = sorted ? (size_t)blocks * (BLOCK / 64) * KQ_KINDS * KQ_FIELDS * KQ_CAP : 0;  // uint4 each
    const size_t park_words = (pooled && !wfq) ? (size_t)blocks * BLOCK * IRIS_WALK_WORDS : 0;
    uint32_t spp_per_chunk;
    Workspace ws;
    if (no_alloc) {
        // everything comes out of this device's copy of the static array; one chunk, so no running sums are needed
        if (!c->scratch_ptr) {  // the lookup at iris_ctx_create failed: once more, and this time the failure is this call's error
            hipError_t e = scratch_for_device(c->device, &c->scratch_ptr);
            if (e != hipSuccess) return fail(c, IRIS_E_HIP, std::string("the static workspace of iris_render_device_noalloc was not found on this device: ") + hipGetErrorString(e));
        }
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
        const size_t o_counters = carve(IRIS_N_COUNTERS * sizeof(unsigned long long)), o_cursor = carve(sizeof(unsigned long long));
        const size_t o_sky = carve(use_sky ? npix : 0);
        const size_t o_stack = carve(stack_hi_words * sizeof(uint32_t));
        const size_t o_park = carve(park_words * sizeof(uint32_t));
        const size_t o_kq = carve(kq_items * sizeof(uint4));
        const size_t o_results = carve((size_t)p->spp_count * npix * sizeof(XyzResult));
        if (off > IRIS_NOALLOC_SCRATCH_BYTES || (uint64_t)p->spp_count * npix > (1ull << 31))
            return fail(c, IRIS_E_LIMIT, "iris_render_device_noalloc: this render needs " + std::to_string(off) + " bytes of workspace (12 per camera sample + " +
                                             std::to_string(o_results) + " fixed for this scene and film), the static workspace holds " +
                                             std::to_string((unsigned long long)IRIS_NOALLOC_SCRATCH_BYTES));
        unsigned char *base = c->scratch_ptr;
        ws.counters = (unsigned long long *)(base + o_counters);
        ws.work_cursor = (unsigned long long *)(base + o_cursor);
        ws.sky = base + o_sky;
        ws.stack_hi = (uint32_t *)(base + o_stack);
        ws.park = (uint32_t *)(base + o_park);
        ws.kq = (uint4 *)(base + o_kq);
        ws.results = (XyzResult *)(base + o_results);
        spp_per_chunk = std::max<uint32_t>(p->spp_count, 1u);
    } else {
        // chunking of the sample range so that the per-sample result buffer fits its budget
        uint64_t max_items = std::min<uint64_t>((uint64_t)c->result_bytes / sizeof(XyzResult), 1ull << 31);
        if (max_items < npix) return fail(c, IRIS_E_LIMIT, "result_bytes is smaller than one sample of the film");
        spp_per_chunk = (uint32_t)std::min<uint64_t>(max_items / npix, std::max<uint32_t>(p->spp_count, 1u));
        size_t need_items = (size_t)spp_per_chunk * npix;
        if (c->results_cap < need_items) {
            dev_free(c->d_results);
            c->d_results = nullptr;
            c->results_cap = 0;
            if ((rc = dev_alloc(c, &c->d_results, need_items))) return rc;
            c->results_cap = need_items;
        }
        if (c->film_cap < npix) {
            dev_free(c->d_film_acc);
            c->d_film_acc = nullptr;
            c->film_cap = 0;
            if ((rc = dev_alloc(c, &c->d_film_acc, (size_t)npix))) return rc;
            c->film_cap = npix;
        }
        if (!reg && (rc = ensure_pools(c, blocks, spb))) return rc;
        if (c->stack_hi_words < stack_hi_words) {  // deeper stack entries of the register kernel: global memory
            dev_free(c->d_stack_hi);
            c->d_stack_hi = nullptr;
            c->stack_hi_words = 0;
            if ((rc = dev_alloc(c, &c->d_stack_hi, stack_hi_words))) return rc;
            c->stack_hi_words = stack_hi_words;
        }
        if (c->park_words < park_words) {  // parking areas of the pooled traversal
            dev_free(c->d_park);
            c->d_park = nullptr;
            c->park_words = 0;
            if ((rc = dev_alloc(c, &c->d_park, park_words))) return rc;
            c->park_words = park_words;
        }
        if (c->kq_items < kq_items) {  // per-wave stacks of path records by BSDF kind (general classes)
            dev_free(c->d_kq);
            c->d_kq = nullptr;
            c->kq_items = 0;
            if ((rc = dev_alloc(c, &c->d_kq, kq_items))) return rc;
            c->kq_items = kq_items;
        }
        if (use_sky && c->sky_cap < npix) {
            dev_free(c->d_sky);
            c->d_sky = nullptr;
            c->sky_cap = 0;
            if ((rc = dev_alloc(c, &c->d_sky, (size_t)npix))) return rc;
            c->sky_cap = npix;
        }
        ws.results = c->d_results;
        ws.film_acc = c->d_film_acc;
        ws.sky = c->d_sky;
        ws.counters = c->d_counters;
        ws.work_cursor = c->d_work_cursor;
        ws.stack_hi = c->d_stack_hi;
        ws.park = c->d_park;
        ws.kq = c->d_kq;
    }
    HIP_TRY(c, hipMemsetAsync(ws.counters, 0, IRIS_N_COUNTERS * sizeof(unsigned long long), stream));
    if (use_sky) {
        hipLaunchKernelGGL(k_sky_mask, dim3((npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, cam_dev(cam), frame_dev(p), c->sc.spheres, c->sc.n_prims, ws.sky);
        HIP_TRY(c, hipGetLastError());
    }

    const float weight = 1.0f / (float)p->spp_total;  // tile.rs:140
    if (p->spp_count == 0 || p->max_depth == 0) {  // `for bounces in 0..0` never runs: radiance is zero
        HIP_TRY(c, hipMemsetAsync(d_out, 0, (size_t)npix * 3 * sizeof(float), stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        return IRIS_OK;
    }
    float trace_ms = 0.0f, resolve_ms = 0.0f;
    for (uint32_t done = 0; done < p->spp_count; done += spp_per_chunk) {
        uint32_t n_s = std::min(spp_per_chunk, p->spp_count - done);
        WfArgs a;
        a.sc = c->sc;
        a.cam = cam_dev(cam);
        a.fr = frame_dev(p);
        a.pool = c->pool;
        a.results = ws.results;
        a.queue = c->d_queue;
        a.blk_cursor = c->d_blk_cursor;
        a.blk_qn = c->d_blk_qn;
        a.live_blocks = c->d_live;
        a.counters = ws.counters;
        a.npix = npix;
        a.div_npix = make_fastdiv(npix);
        a.div_width = make_fastdiv(p->width);
        a.s_begin = p->spp_begin + done;
        a.n_items = n_s * npix;
        a.items_per_block = ((a.n_items + blocks * 64u - 1) / (blocks * 64u)) * 64u;  // whole runs of 64
        a.slots_per_block = spb;
        a.sphere_cap = sphere_cap_for(c);
        a.kq = sorted ? ws.kq : nullptr;  // nullptr: the kernel sorts nothing and never touches the stacks
        a.sky = use_sky ? ws.sky : nullptr;
        a.stack_hi = ws.stack_hi;
        a.park = ws.park;
        a.stack_lds_n = reg_stack_lds;
        if (reg && c->sc.n_bvh_nodes) a.sphere_cap = reg_stack_lds * BLOCK * sizeof(uint32_t) / sizeof(float4);  // LDS for the first levels only
        a.work_cursor = ws.work_cursor;
        // a run: 2048 items for the Lambertian list kernel (512 .. 4096 measure alike there), 512 for the general classes and 256 under
        // a hierarchy, where an item costs several times as much and the launch's tail -- one run per wave -- weighs accordingly
        // (config D: 4096 125.5 ms, 2048 119.8, 1024 117.1, 512 115.9, 256 115.4, 64 116.1); less for small launches so that every wave
        // still gets several
        const uint32_t run_cap = c->sc.n_bvh_nodes ? 256u : (c->scene_class == IRIS_CLS_LAMBERT_CONSTANT ? 2048u : 512u);
        a.work_run = c->work_run ? (uint32_t)c->work_run : std::min<uint32_t>(run_cap, std::max<uint32_t>(64u, (a.n_items / (blocks * (BLOCK / 64u) * 8u)) & ~63u));
        if (reg || wfq) HIP_TRY(c, hipMemsetAsync(ws.work_cursor, 0, sizeof(unsigned long long), stream));
        HIP_TRY(c, hipEventRecord(c->ev0, stream));
        switch (p->integrator) {
        case IRIS_INT_HWSS_NAIVE: rc = launch_chunk<IRIS_INT_HWSS_NAIVE>(c, a, blocks, stream); break;
#ifndef IRIS_EXPERIMENT_INTEG0  // experiment builds (python -m iris_amd.build --integ0 -o ...): HwssNaive kernels only
        case IRIS_INT_HWSS_SLOW: rc = launch_chunk<IRIS_INT_HWSS_SLOW>(c, a, blocks, stream); break;
        case IRIS_INT_SWSS_NAIVE: rc = launch_chunk<IRIS_INT_SWSS_NAIVE>(c, a, blocks, stream); break;
        default: rc = launch_chunk<IRIS_INT_SWSS_SLOW>(c, a, blocks, stream); break;
#else
        default: return fail(c, IRIS_E_ARG, "this experiment build holds the HwssNaive kernels only");
#endif
        }
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev1, stream));
        bool first = done == 0, last = done + n_s >= p->spp_count;
        hipLaunchKernelGGL(k_resolve, dim3((npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, ws.results, ws.film_acc, d_out, npix, n_s,
                           first ? 1 : 0, last ? 1 : 0, weight, (const unsigned char *)(use_sky ? ws.sky : nullptr));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev2, stream));
        HIP_TRY(c, hipEventSynchronize(c->ev2));
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        trace_ms += ms;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev1, c->ev2));
        resolve_ms += ms;
        c->stats.chunks += 1;
    }
    unsigned long long *h = c->h_counters;  // pinned: the copy needs no staging buffer
    HIP_TRY(c, hipMemcpyAsync(h, ws.counters, IRIS_N_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
#ifdef IRIS_PHASE_TIMERS
    {
        unsigned long long t[IRIS_N_COUNTERS];
        (void)hipMemcpy(t, ws.counters, sizeof t, hipMemcpyDeviceToHost);
        const double tot = (double)(t[4] + t[5] + t[6] + t[7]);
        std::fprintf(stderr, "[phase timers] extend %.1f%% generate %.1f%% refill %.1f%% shade %.1f%% (wave-cycles %.3g)\n", 100.0 * t[4] / tot,
                     100.0 * t[5] / tot, 100.0 * t[6] / tot, 100.0 * t[7] / tot, tot);
        std::fprintf(stderr, "[phase timers] shade iterations %llu, lanes holding a hit per iteration %.2f of 64\n", t[9], (double)t[8] / (double)std::max(1ull, t[9]));
        std::fprintf(stderr, "[phase timers] sorted iterations %llu: %.2f records moved out, %.2f taken from the stacks\n", t[11],
                     (double)t[12] / (double)std::max(1ull, t[11]), (double)t[13] / (double)std::max(1ull, t[11]));
        {
            unsigned long long g[8] = {};
            (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(g_pool_stats), sizeof g);
            std::fprintf(stderr, "[pool] passes %llu rays/pass %.1f | fetch rounds/pass %.2f lanes/round %.1f | inner rounds/pass %.1f lanes/round %.1f | leaf rounds/pass %.1f lanes/round %.1f\n",
                         g[0], (double)g[1] / std::max(1ull, g[0]), (double)g[2] / std::max(1ull, g[0]), (double)g[3] / std::max(1ull, g[2]),
                         (double)g[4] / std::max(1ull, g[0]), (double)g[5] / std::max(1ull, g[4]), (double)g[6] / std::max(1ull, g[0]), (double)g[7] / std::max(1ull, g[6]));
        }
        std::fprintf(stderr, "[phase timers] share of sorted iterations by kind: microfacet %.2f specular %.2f fresnel %.2f\n", (double)t[14] / (double)std::max(1ull, t[11]),
                     (double)t[15] / (double)std::max(1ull, t[11]), (double)t[16] / (double)std::max(1ull, t[11]));
    }
#endif
#ifdef IRIS_WFQ_STATS
    if (wfq) {
        unsigned long long g[8] = {};
        (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(g_wfq_stats), sizeof g);
        std::fprintf(stderr, "[wfq trace] rays %llu | fetches %llu (%.1f rays each) | rounds %llu, lanes holding a ray %.1f | inner rounds %llu, lanes per step %.1f (steps %.3g) | leaf rounds %llu, lanes %.1f\n",
                     g[1], g[0], (double)g[1] / std::max(1ull, g[0]), g[6], (double)g[7] / std::max(1ull, g[6]), g[2], (double)g[3] / std::max(1ull, g[2]) / IRIS_WFQ_INNER_REPS, (double)g[3],
                     g[4], (double)g[5] / std::max(1ull, g[4]));
        unsigned long long z[8] = {};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wfq_stats), z, sizeof z);
    }
#endif
    c->stats.samples = h[0];
    c->stats.bounces = h[1];
    c->stats.rays = h[2];
    c->stats.sphere_tests = h[3];
    c->stats.kernel_ms = trace_ms;
    c->stats.resolve_ms = resolve_ms;
    c->stats.bvh_depth = c->bvh_depth;
    c->stats.sky_samples = h[10];
    for (int k = 0; k < 4; k++) c->invariants[k] = h[20 + k];
    c->stats.kernel = wfq ? IRIS_KERNEL_QUEUES : (reg ? IRIS_KERNEL_REGISTER : (c->multi_kernel ? IRIS_KERNEL_POOL_STAGES : IRIS_KERNEL_POOL_FUSED));
    return IRIS_OK;
}

}  // namespace

extern "C" {

int iris_ctx_create(int device, iris_ctx **out) {
    if (!out) return IRIS_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return IRIS_E_HIP;
    iris_ctx *c = new (std::nothrow) iris_ctx();
    if (!c) return IRIS_E_HIP;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return IRIS_E_HIP; }
    c->n_cus = prop.multiProcessorCount;
    if (hipMalloc((void **)&c->d_counters, IRIS_N_COUNTERS * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->d_live, sizeof(uint32_t)) != hipSuccess || hipMalloc((void **)&c->d_work_cursor, sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc((void **)&c->h_live, sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->h_counters, IRIS_N_COUNTERS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess) {
        iris_ctx_destroy(c);
        return IRIS_E_HIP;
    }
    // only iris_render_device_noalloc needs the static workspace: a failed lookup is retried, and reported, there
    if (scratch_for_device(device, &c->scratch_ptr) != hipSuccess) { c->scratch_ptr = nullptr; (void)hipGetLastError(); }
    *out = c;
    return IRIS_OK;
}

void iris_ctx_destroy(iris_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_pools(c);
    free_wfq(c);
    dev_free(c->d_spheres); dev_free(c->d_prim_info); dev_free(c->d_bsdfs); dev_free(c->d_spectra); dev_free(c->d_lights); dev_free(c->d_spec_over_pi); dev_free(c->d_prim_kind); dev_free(c->d_light_rec);
    dev_free(c->d_bvh_nodes); dev_free(c->d_bvh_spheres); dev_free(c->d_bvh_prim); dev_free(c->d_bvh_q);
    dev_free(c->d_env_texels); dev_free(c->d_env_func); dev_free(c->d_env_cond); dev_free(c->d_env_marg);
    dev_free(c->d_up_coeff); dev_free(c->d_up_scale);
    dev_free(c->d_counters); dev_free(c->d_live); dev_free(c->d_results); dev_free(c->d_film_acc); dev_free(c->d_kq); dev_free(c->d_stack_hi); dev_free(c->d_park); dev_free(c->d_sky); dev_free(c->d_work_cursor);
    if (c->h_live) (void)hipHostFree(c->h_live);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    delete c;
}

const char *iris_last_error(const iris_ctx *ctx) { return ctx ? ctx->err.c_str() : "invalid context"; }

int iris_ctx_set_option(iris_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return IRIS_E_ARG;
    std::string k(key);
    if (k == "blocks_per_cu") {
        if (value < 0 || value > 64) return fail(c, IRIS_E_ARG, "blocks_per_cu out of range [0, 64]");
        c->blocks_per_cu = value;
    } else if (k == "slots_per_block") {
        if (value < BLOCK || value > 65536 || value % BLOCK) return fail(c, IRIS_E_ARG, "slots_per_block must be a multiple of 256 in [256, 65536]");
        c->slots_per_block = value;
    } else if (k == "result_bytes") {
        if (value < (1 << 20)) return fail(c, IRIS_E_ARG, "result_bytes below 1 MiB");
        c->result_bytes = value;
    } else if (k == "multi_kernel") {
        c->multi_kernel = value ? 1 : 0;
    } else if (k == "mode") {
        if (value < 0 || value > 3) return fail(c, IRIS_E_ARG, "mode must be 0 (auto), 1 (pool), 2 (register) or 3 (queues)");
        c->mode = value;
    } else if (k == "bvh_min_prims") {
        if (value < 0) return fail(c, IRIS_E_ARG, "bvh_min_prims must be >= 0");
        c->bvh_min_prims = value;  // takes effect at the next iris_scene_upload
    } else if (k == "sort_kinds") {
        if (value < 0 || value > 1) return fail(c, IRIS_E_ARG, "sort_kinds must be 0 or 1");
        c->sort_kinds = value;
    } else if (k == "sky_mask") {
        c->sky_mask = value ? 1 : 0;
    } else if (k == "bvh_leaf") {
        if (value < 1 || value > 7) return fail(c, IRIS_E_ARG, "bvh_leaf out of range [1, 7]");
        c->bvh_leaf = value;  // takes effect at the next iris_scene_upload
    } else if (k == "specialize") {
        c->specialize = value ? 1 : 0;
        if (c->scene_class != IRIS_CLS_GENERAL_ENV)
            c->scene_class = (c->specialize && c->scene_is_lambert_constant) ? IRIS_CLS_LAMBERT_CONSTANT : IRIS_CLS_GENERAL;
    } else if (k == "work_run") {
        if (value < 0 || value > (1 << 20) || value % 64) return fail(c, IRIS_E_ARG, "work_run must be 0 (auto) or a multiple of 64 up to 2^20");
        c->work_run = value;
    } else if (k == "wfq_slots") {
        if (value < BLOCK || value > (1ll << 28) || value % BLOCK) return fail(c, IRIS_E_ARG, "wfq_slots must be a multiple of 256 in [256, 2^28]");
        c->wfq_slots = value;
    } else if (k == "wfq_pipes") {
        if (value < 1 || value > WFQ_MAX_PIPES) return fail(c, IRIS_E_ARG, "wfq_pipes out of range [1, 4]");
        c->wfq_pipes = value;
    } else if (k == "wfq_shade_per_cu" || k == "wfq_trace_per_cu") {
        if (value < 1 || value > 16) return fail(c, IRIS_E_ARG, k + " out of range [1, 16]");
        (k == "wfq_shade_per_cu" ? c->wfq_shade_per_cu : c->wfq_trace_per_cu) = value;
    } else if (k == "wfq_item_run") {
        if (value < BLOCK || value > (1 << 20)) return fail(c, IRIS_E_ARG, "wfq_item_run out of range [256, 2^20]");
        c->wfq_item_run = value;
    } else if (k == "wfq_batch") {
        if (value < 1 || value > 1024) return fail(c, IRIS_E_ARG, "wfq_batch out of range [1, 1024]");
        c->wfq_batch = value;
    } else if (k == "reg_blocks_per_cu") {
        if (value < 1 || value > 1024) return fail(c, IRIS_E_ARG, "reg_blocks_per_cu out of range [1, 1024]");
        c->reg_blocks_per_cu = value;
    } else {
        return fail(c, IRIS_E_ARG, "unknown option " + k);
    }
    return IRIS_OK;
}

int iris_scene_upload(iris_ctx *c, const iris_scene_desc *s) {
    if (!c) return IRIS_E_ARG;
    if (!s) return fail(c, IRIS_E_ARG, "scene is NULL");
    if (s->n_prims == 0 || !s->spheres || !s->prim_material || !s->prim_light) return fail(c, IRIS_E_ARG, "scene has no primitives");
    if ((s->n_bsdfs && !s->bsdfs) || (s->n_spectra && !s->spectra) || (s->n_lights && !s->lights)) return fail(c, IRIS_E_ARG, "NULL table with non-zero count");
    // validate every index the kernels will follow, so that a malformed scene cannot fault the GPU
    for (uint32_t i = 0; i < s->n_prims; i++) {
        // finite geometry of bounded size: squared distances (up to 2^123) then stay finite, which the intersection arithmetic relies on
        const iris_sphere &q = s->spheres[i];
        if (!(std::fabs(q.cx) <= 0x1p60f && std::fabs(q.cy) <= 0x1p60f && std::fabs(q.cz) <= 0x1p60f && std::fabs(q.r) <= 0x1p60f))
            return fail(c, IRIS_E_LIMIT, "sphere " + std::to_string(i) + ": centre and radius must be finite and at most 2^60 in magnitude");
        if (s->prim_material[i] < -1 || s->prim_material[i] >= (int32_t)s->n_bsdfs) return fail(c, IRIS_E_ARG, "prim_material out of range");
        if (s->prim_light[i] < -1 || s->prim_light[i] >= (int32_t)s->n_lights) return fail(c, IRIS_E_ARG, "prim_light out of range");
    }
    for (uint32_t i = 0; i < s->n_bsdfs; i++) {
        const iris_bsdf &b = s->bsdfs[i];
        if (b.kind > IRIS_BSDF_NULL) return fail(c, IRIS_E_ARG, "unknown bsdf kind");
        if (b.kind != IRIS_BSDF_NULL && b.spec_a >= s->n_spectra) return fail(c, IRIS_E_ARG, "bsdf spec_a out of range");
        if (b.kind == IRIS_BSDF_FRESNEL && b.spec_b >= s->n_spectra) return fail(c, IRIS_E_ARG, "bsdf spec_b out of range");
        if (b.kind == IRIS_BSDF_MICROFACET && (b.p0 == 0.0f || b.p1 == 0.0f)) return fail(c, IRIS_E_ARG, "microfacet alpha must be non-zero");
    }
    for (uint32_t i = 0; i < s->n_spectra; i++)
        if (s->spectra[i].kind > IRIS_SPEC_UPSAMPLED_HDR) return fail(c, IRIS_E_ARG, "unknown spectrum kind");
    for (uint32_t i = 0; i < s->n_lights; i++) {
        if (s->lights[i].spectrum >= s->n_spectra) return fail(c, IRIS_E_ARG, "light spectrum out of range");
        if (s->lights[i].prim >= s->n_prims) return fail(c, IRIS_E_ARG, "light prim out of range");
    }
    const iris_envmap_desc *em = s->env_map;
    if (em && em->width != 0) {
        if (!em->texels || em->height == 0) return fail(c, IRIS_E_ARG, "env_map without texels or height");
        if ((uint64_t)em->width * em->height > (1ull << 28)) return fail(c, IRIS_E_LIMIT, "env_map above 2^28 texels");
    } else {
        em = nullptr;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    dev_free(c->d_spheres); dev_free(c->d_prim_info); dev_free(c->d_bsdfs); dev_free(c->d_spectra); dev_free(c->d_lights); dev_free(c->d_spec_over_pi); dev_free(c->d_prim_kind); dev_free(c->d_light_rec);
    c->d_spec_over_pi = nullptr;
    c->d_prim_kind = nullptr;
    dev_free(c->d_bvh_nodes); dev_free(c->d_bvh_spheres); dev_free(c->d_bvh_prim); dev_free(c->d_bvh_q);
    dev_free(c->d_env_texels); dev_free(c->d_env_func); dev_free(c->d_env_cond); dev_free(c->d_env_marg);
    c->d_env_texels = c->d_env_func = c->d_env_cond = c->d_env_marg = nullptr;
    c->d_spheres = c->d_prim_info = c->d_bsdfs = c->d_spectra = c->d_lights = c->d_light_rec = nullptr;
    c->d_bvh_nodes = c->d_bvh_spheres = c->d_bvh_prim = c->d_bvh_q = nullptr;
    c->sc = SceneDev{};
    c->have_scene = false;
    // per primitive: (material, light, albedo / PI of a Lambertian material over a constant spectrum -- lambertian.rs:30, one IEEE
    // division, formed here once -- else 0, 0)
    std::vector<int4> info(s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; i++) {
        float over_pi_i = 0.0f;
        const int m = s->prim_material[i];
        if (m >= 0 && s->bsdfs[m].kind == IRIS_BSDF_LAMBERTIAN && s->spectra[s->bsdfs[m].spec_a].kind == IRIS_SPEC_CONSTANT)
            over_pi_i = iris_f32_div(s->spectra[s->bsdfs[m].spec_a].c0, PI_F);
        int bits;
        std::memcpy(&bits, &over_pi_i, 4);
        info[i] = make_int4(m, s->prim_light[i], bits, 0);
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    HIP_TRY(c, up(&c->d_spheres, s->spheres, sizeof(iris_sphere) * s->n_prims));
    HIP_TRY(c, up(&c->d_prim_info, info.data(), sizeof(int4) * s->n_prims));
    HIP_TRY(c, up(&c->d_bsdfs, s->bsdfs, sizeof(iris_bsdf) * s->n_bsdfs));
    HIP_TRY(c, up(&c->d_spectra, s->spectra, sizeof(iris_spectrum) * s->n_spectra));
    HIP_TRY(c, up(&c->d_lights, s->lights, sizeof(iris_light) * s->n_lights));
    {
        std::vector<float4> rec(3 * (size_t)s->n_lights);
        auto as_float = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
        for (uint32_t i = 0; i < s->n_lights; i++) {
            const iris_sphere &sp = s->spheres[s->lights[i].prim];
            const iris_spectrum &e = s->spectra[s->lights[i].spectrum];
            rec[3 * i] = make_float4(sp.cx, sp.cy, sp.cz, sp.r);
            rec[3 * i + 1] = make_float4(as_float(s->lights[i].prim), as_float(e.kind), e.c0, e.c1);
            rec[3 * i + 2] = make_float4(e.c2, e.hdr, 0.0f, 0.0f);
        }
        HIP_TRY(c, up(&c->d_light_rec, rec.data(), sizeof(float4) * rec.size()));
        c->sc.light_rec = (const float4 *)c->d_light_rec;
    }
    std::vector<float> over_pi(std::max<uint32_t>(s->n_spectra, 1u), 0.0f);
    for (uint32_t i = 0; i < s->n_spectra; i++) over_pi[i] = iris_f32_div(s->spectra[i].c0, PI_F);  // lambertian.rs:30: albedo / PI, one IEEE division
    HIP_TRY(c, up(&c->d_spec_over_pi, over_pi.data(), sizeof(float) * over_pi.size()));
    c->sc.spec_over_pi = (const float *)c->d_spec_over_pi;
    std::vector<unsigned char> kind(s->n_prims, 0);
    for (uint32_t i = 0; i < s->n_prims; i++) {  // the BSDF's kind; no material and the absorbing BSDF count as Lambertian (0): the cheap group
        const uint32_t k = s->prim_material[i] >= 0 ? s->bsdfs[s->prim_material[i]].kind : 0u;
        kind[i] = (unsigned char)(k < KQ_KINDS ? k : 0u);
    }
    {
        bool seen[KQ_KINDS] = {};
        for (unsigned char k : kind) seen[k] = true;
        c->n_bsdf_kinds = 0;
        for (bool b : seen) c->n_bsdf_kinds += b ? 1 : 0;
    }
    HIP_TRY(c, up(&c->d_prim_kind, kind.data(), kind.size()));
    c->sc.prim_kind = (const unsigned char *)c->d_prim_kind;
    c->sc.spheres = (const float4 *)c->d_spheres;
    c->sc.prim_info = (const int4 *)c->d_prim_info;
    c->sc.bsdfs = (const iris_bsdf *)c->d_bsdfs;
    c->sc.spectra = (const iris_spectrum *)c->d_spectra;
    c->sc.lights = (const iris_light *)c->d_lights;
    c->sc.n_prims = s->n_prims;
    c->sc.n_lights = s->n_lights;
    // hierarchy for long lists (the scans then visit O(log n) spheres; the answer is the linear scan's bit for bit)
    c->bvh_depth = 0;
    if (c->bvh_min_prims > 0 && (int64_t)s->n_prims >= c->bvh_min_prims) {
        Bvh bvh;
        if (build_bvh(s->spheres, s->n_prims, (uint32_t)c->bvh_leaf, bvh)) {
            HIP_TRY(c, up(&c->d_bvh_nodes, bvh.nodes.data(), sizeof(float4) * bvh.nodes.size()));
            HIP_TRY(c, up(&c->d_bvh_q, bvh.qnodes.data(), sizeof(uint32_t) * bvh.qnodes.size()));
            HIP_TRY(c, up(&c->d_bvh_spheres, bvh.spheres.data(), sizeof(float4) * bvh.spheres.size()));
            HIP_TRY(c, up(&c->d_bvh_prim, bvh.prim.data(), sizeof(uint32_t) * bvh.prim.size()));
            c->sc.bvh_nodes = (const float4 *)c->d_bvh_nodes;
            c->sc.bvh_q = (const uint4 *)c->d_bvh_q;
            for (int k = 0; k < 3; k++) c->sc.bvh_g0[k] = bvh.grid_origin[k];
            c->sc.bvh_s = bvh.grid_scale;
            c->sc.bvh_inv_s = bvh.grid_inv_scale;
            c->sc.bvh_spheres = (const float4 *)c->d_bvh_spheres;
            c->sc.bvh_prim = (const uint32_t *)c->d_bvh_prim;
            c->sc.n_bvh_nodes = (uint32_t)(bvh.nodes.size() / 2);
            c->sc.bvh_big_first = s->n_prims - bvh.n_big;
            c->sc.bvh_big_count = bvh.n_big;
            c->bvh_depth = bvh.depth;
        }
    }
    // environment map: texels and the sampling tables (built on the host, iris_envmap_tables) go to HBM
    if (em) {
        const size_t n = (size_t)em->width * em->height;
        std::vector<float> func(n), cond((size_t)(em->width + 1) * em->height), marg(em->height + 1);
        float integral = 0.0f;
        if (iris_envmap_tables(em, func.data(), cond.data(), marg.data(), &integral) != IRIS_OK) return fail(c, IRIS_E_ARG, "env_map tables");
        HIP_TRY(c, up(&c->d_env_texels, em->texels, sizeof(float) * 4 * n));
        HIP_TRY(c, up(&c->d_env_func, func.data(), sizeof(float) * func.size()));
        // search levels (EnvSearch in iris_device.h): entries 0 .. n-1 of every CDF row in blocks of 32, padded with a
        // value above every draw, then every 32nd entry of that, until one block is left; all levels of all rows in
        // one allocation per table
        auto levels = [](const std::vector<float> &cdf, uint32_t rows, uint32_t n, std::vector<float> &out, uint32_t offs[3], uint32_t strides[3]) -> uint32_t {
            uint32_t n_levels = 0, entries = n;
            std::vector<float> cur((size_t)rows * n);
            for (uint32_t r = 0; r < rows; r++)
                for (uint32_t i = 0; i < n; i++) cur[(size_t)r * n + i] = cdf[(size_t)r * (n + 1) + i];
            out.clear();
            for (;;) {
                const uint32_t padded = (entries + IRIS_ENV_BLOCK - 1u) / IRIS_ENV_BLOCK * IRIS_ENV_BLOCK;
                offs[n_levels] = (uint32_t)out.size();
                strides[n_levels] = padded;
                for (uint32_t r = 0; r < rows; r++) {
                    out.insert(out.end(), cur.begin() + (size_t)r * entries, cur.begin() + (size_t)(r + 1) * entries);
                    out.insert(out.end(), padded - entries, IRIS_ENV_PAD);
                }
                n_levels++;
                if (entries <= IRIS_ENV_BLOCK || n_levels == 3) break;
                const uint32_t next = (entries + IRIS_ENV_BLOCK - 1u) / IRIS_ENV_BLOCK;
                std::vector<float> nxt((size_t)rows * next);
                for (uint32_t r = 0; r < rows; r++)
                    for (uint32_t j = 0; j < next; j++) nxt[(size_t)r * next + j] = cur[(size_t)r * entries + (size_t)j * IRIS_ENV_BLOCK];
                cur.swap(nxt);
                entries = next;
            }
            return n_levels;
        };
        if (em->width > IRIS_ENV_BLOCK * IRIS_ENV_BLOCK * IRIS_ENV_BLOCK || em->height > IRIS_ENV_BLOCK * IRIS_ENV_BLOCK * IRIS_ENV_BLOCK)
            return fail(c, IRIS_E_LIMIT, "env_map side above 32768 texels");
        std::vector<float> cond_lv, marg_lv;
        uint32_t co[3] = {0, 0, 0}, cs[3] = {0, 0, 0}, mo[3] = {0, 0, 0}, ms[3] = {0, 0, 0};
        const uint32_t cl = levels(cond, em->height, em->width, cond_lv, co, cs);
        const uint32_t ml = levels(marg, 1u, em->height, marg_lv, mo, ms);
        HIP_TRY(c, up(&c->d_env_cond, cond_lv.data(), sizeof(float) * cond_lv.size()));
        HIP_TRY(c, up(&c->d_env_marg, marg_lv.data(), sizeof(float) * marg_lv.size()));
        c->sc.env_texels = (const float4 *)c->d_env_texels;
        c->sc.env_func = (const float *)c->d_env_func;
        for (uint32_t l = 0; l < 3; l++) {
            c->sc.env_cond_lvl[l] = (const float *)c->d_env_cond + co[l];
            c->sc.env_marg_lvl[l] = (const float *)c->d_env_marg + mo[l];
            c->sc.env_cond_stride[l] = cs[l];
        }
        c->sc.env_cond_levels = cl;
        c->sc.env_marg_levels = ml;
        c->sc.env_w = em->width;
        c->sc.env_h = em->height;
        c->sc.env_integral = integral;
    }
    // scene class: only Lambertian BSDFs and only constant spectra -> the specialised kernels apply
    bool lc = em == nullptr;
    for (uint32_t i = 0; i < s->n_bsdfs; i++) lc = lc && s->bsdfs[i].kind == IRIS_BSDF_LAMBERTIAN;
    for (uint32_t i = 0; i < s->n_spectra; i++) lc = lc && s->spectra[i].kind == IRIS_SPEC_CONSTANT;
    c->scene_is_lambert_constant = lc;
    c->scene_class = em ? IRIS_CLS_GENERAL_ENV : ((c->specialize && lc) ? IRIS_CLS_LAMBERT_CONSTANT : IRIS_CLS_GENERAL);
    c->have_scene = true;
    return IRIS_OK;
}

int iris_render_device(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, float *d_out, void *stream) {
    if (!c) return IRIS_E_ARG;
    return render_impl(c, cam, p, d_out, (hipStream_t)stream);
}

int iris_render_device_noalloc(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, float *d_out, void *stream) {
    if (!c) return IRIS_E_ARG;
    return render_impl(c, cam, p, d_out, (hipStream_t)stream, true);
}

size_t iris_noalloc_workspace_bytes(void) { return (size_t)IRIS_NOALLOC_SCRATCH_BYTES; }

int iris_render(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, float *out_xyz) {
    if (!c) return IRIS_E_ARG;
    int rc;
    if ((rc = check_params(c, p))) return rc;
    if (!out_xyz) return fail(c, IRIS_E_ARG, "out_xyz is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)p->width * p->height * 3;
    float *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_out, n * sizeof(float)));
    rc = render_impl(c, cam, p, d_out, nullptr);
    if (rc == IRIS_OK) {
        hipError_t e = hipMemcpy(out_xyz, d_out, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, IRIS_E_HIP, std::string("hipMemcpy film: ") + hipGetErrorString(e));
    }
    dev_free(d_out);
    return rc;
}

int iris_get_invariant_counts(const iris_ctx *c, uint64_t out[4]) {
    if (!c || !out) return IRIS_E_ARG;
#if IRIS_DEBUG_INVARIANTS
    for (int k = 0; k < 4; k++) out[k] = c->invariants[k];
    return IRIS_OK;
#else
    for (int k = 0; k < 4; k++) out[k] = 0;
    return IRIS_E_STATE;  // this build of the library does not count them (libiris_hip_debug.so does)
#endif
}

int iris_get_stats(const iris_ctx *c, iris_stats *out) {
    if (!c || !out) return IRIS_E_ARG;
    *out = c->stats;
    return IRIS_OK;
}

// ---- the exchange step of a multi-GPU render: sum the partial films over xGMI (RCCL) ----
// One process drives n GPUs (one iris_ctx each, e.g. from n host threads); every GPU has rendered a disjoint
// range of sample indices into its own device film.  RCCL is loaded at first use (dlopen), so a single-GPU
// caller never needs it, and the communicators are created once per device list and kept.
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::vector<int> devices;
    std::vector<void *> comms;
    std::vector<hipStream_t> streams;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;  // ncclDataType_t / ncclRedOp_t of rccl.h

bool rccl_load(std::string &err) {
    if (g_rccl.lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { return dlsym(g_rccl.lib, n); };
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString) {
        err = "librccl.so lacks an expected symbol";
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
        return false;
    }
    return true;
}
void rccl_drop_comms() {
    for (size_t i = 0; i < g_rccl.comms.size(); i++) {
        (void)hipSetDevice(g_rccl.devices[i]);
        if (g_rccl.comms[i]) (void)g_rccl.CommDestroy(g_rccl.comms[i]);
        if (g_rccl.streams[i]) (void)hipStreamDestroy(g_rccl.streams[i]);
    }
    g_rccl.comms.clear(); g_rccl.streams.clear(); g_rccl.devices.clear();
}
}  // namespace

int iris_allreduce_films(iris_ctx *const *ctxs, int n, float *const *d_films, size_t n_floats) {
    if (!ctxs || n < 1 || !ctxs[0]) return IRIS_E_ARG;
    iris_ctx *c0 = ctxs[0];
    if (!d_films) return fail(c0, IRIS_E_ARG, "d_films is NULL");
    std::vector<int> devices((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !d_films[i]) return fail(c0, IRIS_E_ARG, "NULL context or film");
        devices[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return fail(c0, IRIS_E_ARG, "two contexts on one device: RCCL needs one rank per GPU (sum such films on the host)");
    }
    if (n_floats == 0) return IRIS_OK;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    std::string err;
    if (!rccl_load(err)) return fail(c0, IRIS_E_RCCL, err);
#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                             \
        int r_ = (expr);                                                                                             \
        if (r_ != 0) return fail(c0, IRIS_E_RCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));           \
    } while (0)
    if (g_rccl.devices != devices) {  // first call, or another set of GPUs
        rccl_drop_comms();
        g_rccl.comms.assign((size_t)n, nullptr);
        g_rccl.streams.assign((size_t)n, nullptr);
        g_rccl.devices = devices;
        int r = g_rccl.CommInitAll(g_rccl.comms.data(), n, devices.data());
        if (r != 0) { g_rccl.devices.clear(); g_rccl.comms.clear(); g_rccl.streams.clear(); return fail(c0, IRIS_E_RCCL, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
        for (int i = 0; i < n; i++) {
            hipError_t e = hipSetDevice(devices[i]);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&g_rccl.streams[i], hipStreamNonBlocking);
            if (e != hipSuccess) {  // leave no half-initialised state behind: the next call starts from scratch
                rccl_drop_comms();
                return fail(c0, IRIS_E_HIP, std::string("stream for the film exchange: ") + hipGetErrorString(e));
            }
        }
    }
    for (int i = 0; i < n; i++) {  // the films were written on other streams: make sure they are complete
        HIP_TRY(c0, hipSetDevice(devices[i]));
        HIP_TRY(c0, hipDeviceSynchronize());
    }
    RCCL_TRY(g_rccl.GroupStart());
    std::string group_err;  // every exit after GroupStart goes through GroupEnd
    int group_code = IRIS_OK;
    for (int i = 0; i < n && group_code == IRIS_OK; i++) {
        hipError_t e = hipSetDevice(devices[i]);
        if (e != hipSuccess) { group_code = IRIS_E_HIP; group_err = std::string("hipSetDevice: ") + hipGetErrorString(e); break; }
        int r = g_rccl.AllReduce(d_films[i], d_films[i], n_floats, NCCL_FLOAT32, NCCL_SUM, g_rccl.comms[i], g_rccl.streams[i]);
        if (r != 0) { group_code = IRIS_E_RCCL; group_err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r); }
    }
    {
        int r = g_rccl.GroupEnd();
        if (group_code == IRIS_OK && r != 0) { group_code = IRIS_E_RCCL; group_err = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r); }
    }
    if (group_code != IRIS_OK) {
        rccl_drop_comms();  // the communicators may hold a half-issued collective
        return fail(c0, group_code, group_err);
    }
#undef RCCL_TRY
    for (int i = 0; i < n; i++) {
        HIP_TRY(c0, hipSetDevice(devices[i]));
        HIP_TRY(c0, hipStreamSynchronize(g_rccl.streams[i]));
    }
    return IRIS_OK;
}

// The split of a multi-GPU render: share `part` of `n_parts` of spp_count sample indices, as a contiguous block.
int iris_sample_range(uint32_t spp_count, uint32_t part, uint32_t n_parts, uint32_t *begin, uint32_t *count) {
    if (!begin || !count || n_parts == 0 || part >= n_parts) return IRIS_E_ARG;
    const uint64_t b = (uint64_t)spp_count * part / n_parts, e = (uint64_t)spp_count * (part + 1) / n_parts;
    *begin = (uint32_t)b;
    *count = (uint32_t)(e - b);
    return IRIS_OK;
}

// The whole multi-GPU render in one call: what main.rs:87-107 does with worker threads and one shared film.
int iris_render_multi(iris_ctx *const *ctxs, int n, const iris_camera *cam, const iris_render_params *p, float *out_xyz) {
    if (!ctxs || n < 1 || !ctxs[0]) return IRIS_E_ARG;
    iris_ctx *c0 = ctxs[0];
    int rc;
    if ((rc = check_params(c0, p))) return rc;
    if (!cam || !out_xyz) return fail(c0, IRIS_E_ARG, "camera or out_xyz is NULL");
    bool distinct = true;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return fail(c0, IRIS_E_ARG, "NULL context");
        for (int j = 0; j < i; j++) distinct = distinct && ctxs[j]->device != ctxs[i]->device;
    }
    const size_t n_floats = (size_t)p->width * p->height * 3;
    std::vector<float *> d_films((size_t)n, nullptr);
    std::vector<int> rcs((size_t)n, IRIS_OK);
    auto cleanup = [&]() {
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(ctxs[i]->device);
            dev_free(d_films[i]);
        }
    };
    for (int i = 0; i < n; i++) {
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipMalloc((void **)&d_films[i], n_floats * sizeof(float)) != hipSuccess) {
            cleanup();
            return fail(c0, IRIS_E_HIP, "hipMalloc of a device film failed");
        }
    }
    {   // one host thread per GPU, each rendering its share of the sample indices of every pixel
        std::vector<std::thread> threads;
        for (int i = 0; i < n; i++) {
            threads.emplace_back([&, i]() {
                iris_render_params q = *p;
                uint32_t b = 0, cnt = 0;
                (void)iris_sample_range(p->spp_count, (uint32_t)i, (uint32_t)n, &b, &cnt);
                q.spp_begin = p->spp_begin + b;
                q.spp_count = cnt;
                rcs[i] = render_impl(ctxs[i], cam, &q, d_films[i], nullptr);
            });
        }
        for (auto &t : threads) t.join();
    }
    for (int i = 0; i < n; i++)
        if (rcs[i] != IRIS_OK) {
            if (i != 0) c0->err = "context " + std::to_string(i) + ": " + ctxs[i]->err;
            cleanup();
            return rcs[i];
        }
    rc = IRIS_OK;
    if (n > 1 && distinct) {
        rc = iris_allreduce_films(ctxs, n, d_films.data(), n_floats);  // the exchange step, over xGMI
        if (rc == IRIS_OK && (hipSetDevice(c0->device) != hipSuccess || hipMemcpy(out_xyz, d_films[0], n_floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess))
            rc = fail(c0, IRIS_E_HIP, "hipMemcpy of the film failed");
    } else {  // one GPU, or several contexts sharing a GPU (a rehearsal): the partial films are summed on the host, in context order
        std::vector<float> part(n_floats);
        for (int i = 0; i < n && rc == IRIS_OK; i++) {
            if (hipSetDevice(ctxs[i]->device) != hipSuccess ||
                hipMemcpy(i == 0 ? out_xyz : part.data(), d_films[i], n_floats * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
                rc = fail(c0, IRIS_E_HIP, "hipMemcpy of a partial film failed");
                break;
            }
            if (i != 0)
                for (size_t k = 0; k < n_floats; k++) out_xyz[k] += part[k];
        }
    }
    cleanup();
    return rc;
}

void iris_allreduce_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) rccl_drop_comms();
}

// ---- debug / parity entry points ----
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { dev_free(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
};
}  // namespace

// ---- UpsampleTable on the device: the table resident in HBM, get_spectrum_hdr as a batch kernel (one thread per RGB value) ----
// The arithmetic of upsample.rs:46-106 operation for operation in f32 (no contraction, IEEE division), so a texel computed
// here equals iris_upsample_get_spectrum_hdr's bit for bit.
__attribute__((visibility("hidden"))) void iris_host_upsample_arrays(const iris_upsample_table *t, const float **coefficients, size_t *n_coefficients, const float **scale);
namespace {
__device__ __forceinline__ float up_rust_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
__device__ __forceinline__ float up_rust_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
// min(x as usize, res - 2): Rust's float-to-integer cast saturates and sends NaN to 0
__device__ __forceinline__ uint32_t up_cell(float x, uint32_t res) { return !(x > 0.0f) ? 0u : (x >= (float)(res - 1u) ? res - 2u : (uint32_t)x); }
__global__ void __launch_bounds__(BLOCK) k_upsample_hdr(const float *__restrict__ coeff, const float *__restrict__ scale_tab, uint32_t res, const float *__restrict__ rgb_in,
                                                         size_t n, float max_hdr, float4 *__restrict__ texels, float *__restrict__ importance) {
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= n) return;
    const float p0 = rgb_in[3 * gid], p1 = rgb_in[3 * gid + 1], p2 = rgb_in[3 * gid + 2];
    // get_spectrum_hdr, upsample.rs:46-56
    const float mx = up_rust_max(up_rust_max(up_rust_max(p0, p1), p2), 1.0f);
    const float rgb[3] = {p0 / mx, p1 / mx, p2 / mx};
    // get_spectrum, upsample.rs:59-106
    uint32_t i = 0u;
    if (rgb[1] >= rgb[i]) i = 1u;
    if (rgb[2] >= (i == 1u ? rgb[1] : rgb[0])) i = 2u;
    const float z = i == 0u ? rgb[0] : (i == 1u ? rgb[1] : rgb[2]);
    const float xa = i == 0u ? rgb[1] : (i == 1u ? rgb[2] : rgb[0]);  // rgb[(i + 1) % 3]
    const float ya = i == 0u ? rgb[2] : (i == 1u ? rgb[0] : rgb[1]);  // rgb[(i + 2) % 3]
    const float sc = (float)(res - 1u) / z;
    const float x = xa * sc, y = ya * sc;
    const uint32_t xi = up_cell(x, res), yi = up_cell(y, res);
    // find_interval, upsample.rs:147-165
    uint32_t left = 0u;
    const uint32_t last_interval = res - 2u;
    uint32_t size = last_interval;
    while (size > 0u) {
        const uint32_t half = size / 2u, middle = left + half + 1u;
        if (scale_tab[middle] < z) { left = middle; size -= half + 1u; }
        else size = half;
    }
    const uint32_t zi = left < last_interval ? left : last_interval;
    size_t offset = ((((size_t)i * res + zi) * res + yi) * res + xi) * 3u;
    const size_t dx = 3u, dy = 3u * (size_t)res, dz = 3u * (size_t)res * res;
    const float x1 = x - (float)xi, x0 = 1.0f - x1, y1 = y - (float)yi, y0 = 1.0f - y1;
    const float z1 = (z - scale_tab[zi]) / (scale_tab[zi + 1u] - scale_tab[zi]), z0 = 1.0f - z1;
    float out[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        out[k] = ((coeff[offset] * x0 + coeff[offset + dx] * x1) * y0 + (coeff[offset + dy] * x0 + coeff[offset + dy + dx] * x1) * y1) * z0 +
                 ((coeff[offset + dz] * x0 + coeff[offset + dz + dx] * x1) * y0 + (coeff[offset + dz + dy] * x0 + coeff[offset + dz + dy + dx] * x1) * y1) * z1;
        offset += 1u;
    }
    texels[gid] = make_float4(out[0], out[1], out[2], up_rust_min(mx, max_hdr));
    if (importance) importance[gid] = 0.2126f * p0 + 0.7152f * p1 + 0.0722f * p2;
}
}  // namespace

int iris_upsample_table_upload(iris_ctx *c, const iris_upsample_table *table) {
    if (!c) return IRIS_E_ARG;
    if (!table) return fail(c, IRIS_E_ARG, "no table");
    HIP_TRY(c, hipSetDevice(c->device));
    const float *coeff, *scale;
    size_t n_coeff;
    iris_host_upsample_arrays(table, &coeff, &n_coeff, &scale);
    const uint32_t res = iris_upsample_table_resolution(table);
    dev_free(c->d_up_coeff); dev_free(c->d_up_scale);
    c->d_up_coeff = c->d_up_scale = nullptr;
    c->up_res = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_up_coeff, n_coeff * sizeof(float)));
    HIP_TRY(c, hipMalloc((void **)&c->d_up_scale, res * sizeof(float)));
    HIP_TRY(c, hipMemcpy(c->d_up_coeff, coeff, n_coeff * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_up_scale, scale, res * sizeof(float), hipMemcpyHostToDevice));
    c->up_res = res;
    return IRIS_OK;
}

int iris_upsample_get_spectrum_hdr_batch(iris_ctx *c, const float *rgb, size_t n, float max_hdr, float *texels, float *importance, int on_device, void *stream_) {
    if (!c) return IRIS_E_ARG;
    if (c->up_res == 0) return fail(c, IRIS_E_STATE, "no table on the device: call iris_upsample_table_upload first");
    if (!rgb || !texels) return fail(c, IRIS_E_ARG, "bad arguments");
    if (n == 0) return IRIS_OK;
    if (n > ((size_t)1 << 31) * BLOCK) return fail(c, IRIS_E_LIMIT, "more than 2^39 values in one batch");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK));
    if (on_device) {
        if ((uintptr_t)texels % 16 != 0) return fail(c, IRIS_E_ARG, "texels must be 16-byte aligned");
        hipLaunchKernelGGL(k_upsample_hdr, grid, dim3(BLOCK), 0, stream, (const float *)c->d_up_coeff, (const float *)c->d_up_scale, c->up_res, rgb, n, max_hdr, (float4 *)texels, importance);
        HIP_TRY(c, hipGetLastError());
        return IRIS_OK;
    }
    DevBuf d_rgb, d_tex, d_imp;
    HIP_TRY(c, d_rgb.alloc(n * 3 * sizeof(float)));
    HIP_TRY(c, d_tex.alloc(n * 4 * sizeof(float)));
    if (importance) HIP_TRY(c, d_imp.alloc(n * sizeof(float)));
    HIP_TRY(c, hipMemcpyAsync(d_rgb.p, rgb, n * 3 * sizeof(float), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_upsample_hdr, grid, dim3(BLOCK), 0, stream, (const float *)c->d_up_coeff, (const float *)c->d_up_scale, c->up_res, (const float *)d_rgb.p, n, max_hdr, (float4 *)d_tex.p,
                       importance ? (float *)d_imp.p : nullptr);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(texels, d_tex.p, n * 4 * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (importance) HIP_TRY(c, hipMemcpyAsync(importance, d_imp.p, n * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    return IRIS_OK;
}

int iris_dbg_sampler(iris_ctx *c, uint32_t n, const uint32_t *xys, uint32_t seed, uint32_t ndraws, float *out) {
    if (!c) return IRIS_E_ARG;
    if (!xys || !out || n == 0 || ndraws == 0) return fail(c, IRIS_E_ARG, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf dx, dout;
    HIP_TRY(c, dx.alloc(sizeof(uint32_t) * 3 * n));
    HIP_TRY(c, dout.alloc(sizeof(float) * (size_t)n * ndraws));
    HIP_TRY(c, hipMemcpy(dx.p, xys, sizeof(uint32_t) * 3 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg_sampler, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, nullptr, (const uint32_t *)dx.p, n, seed, ndraws, (float *)dout.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(out, dout.p, sizeof(float) * (size_t)n * ndraws, hipMemcpyDeviceToHost));
    return IRIS_OK;
}

int iris_dbg_raygen(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, uint32_t n, const uint32_t *xys, float *hero_out,
                    float *ray_o, float *ray_d) {
    if (!c) return IRIS_E_ARG;
    int rc;
    if ((rc = check_params(c, p))) return rc;
    if (!cam || !xys || !hero_out || !ray_o || !ray_d || n == 0) return fail(c, IRIS_E_ARG, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf dx, dh, dorig, ddir;
    HIP_TRY(c, dx.alloc(sizeof(uint32_t) * 3 * n));
    HIP_TRY(c, dh.alloc(sizeof(float) * 4 * n));
    HIP_TRY(c, dorig.alloc(sizeof(float) * 3 * n));
    HIP_TRY(c, ddir.alloc(sizeof(float) * 3 * n));
    HIP_TRY(c, hipMemcpy(dx.p, xys, sizeof(uint32_t) * 3 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg_raygen, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, nullptr, cam_dev(cam), frame_dev(p), (const uint32_t *)dx.p, n,
                       (float *)dh.p, (float *)dorig.p, (float *)ddir.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(hero_out, dh.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(ray_o, dorig.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(ray_d, ddir.p, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
    return IRIS_OK;
}

int iris_dbg_intersect(iris_ctx *c, uint32_t n, const float *ray_o, const float *ray_d, int32_t *prim, float *t) {
    if (!c) return IRIS_E_ARG;
    if (!c->have_scene) return fail(c, IRIS_E_STATE, "no scene uploaded");
    if (!ray_o || !ray_d || !prim || !t || n == 0) return fail(c, IRIS_E_ARG, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf dorig, ddir, dp, dt;
    HIP_TRY(c, dorig.alloc(sizeof(float) * 3 * n));
    HIP_TRY(c, ddir.alloc(sizeof(float) * 3 * n));
    HIP_TRY(c, dp.alloc(sizeof(int32_t) * n));
    HIP_TRY(c, dt.alloc(sizeof(float) * n));
    HIP_TRY(c, hipMemcpy(dorig.p, ray_o, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(ddir.p, ray_d, sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    uint32_t cap = sphere_cap_for(c);
    hipLaunchKernelGGL(k_dbg_intersect, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), (size_t)cap * sizeof(float4), nullptr, c->sc, cap,
                       (const float *)dorig.p, (const float *)ddir.p, n, (int32_t *)dp.p, (float *)dt.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(prim, dp.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(t, dt.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    return IRIS_OK;
}

int iris_dbg_trace(iris_ctx *c, const iris_camera *cam, const iris_render_params *p, uint32_t n, const uint32_t *xys, iris_trace *out) {
    if (!c) return IRIS_E_ARG;
    int rc;
    if ((rc = check_params(c, p))) return rc;
    if ((rc = check_scene_for(c, p))) return rc;
    if (!cam || !xys || !out || n == 0) return fail(c, IRIS_E_ARG, "bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf dx, dtr;
    HIP_TRY(c, dx.alloc(sizeof(uint32_t) * 3 * n));
    HIP_TRY(c, dtr.alloc(sizeof(iris_trace) * (size_t)n));
    HIP_TRY(c, hipMemcpy(dx.p, xys, sizeof(uint32_t) * 3 * n, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(dtr.p, 0, sizeof(iris_trace) * (size_t)n));
    uint32_t cap = sphere_cap_for(c);
    dim3 grid((n + BLOCK - 1) / BLOCK), block(BLOCK);
    size_t dyn = (size_t)cap * sizeof(float4);
    CamDev cd = cam_dev(cam);
    FrameDev fd = frame_dev(p);
    switch (p->integrator) {
    case IRIS_INT_HWSS_NAIVE: hipLaunchKernelGGL(k_dbg_trace<IRIS_INT_HWSS_NAIVE>, grid, block, dyn, nullptr, c->sc, cd, fd, cap, (const uint32_t *)dx.p, n, (iris_trace *)dtr.p); break;
#ifndef IRIS_EXPERIMENT_INTEG0  // experiment builds (python -m iris_amd.build --integ0 -o ...): HwssNaive kernels only
    case IRIS_INT_HWSS_SLOW: hipLaunchKernelGGL(k_dbg_trace<IRIS_INT_HWSS_SLOW>, grid, block, dyn, nullptr, c->sc, cd, fd, cap, (const uint32_t *)dx.p, n, (iris_trace *)dtr.p); break;
    case IRIS_INT_SWSS_NAIVE: hipLaunchKernelGGL(k_dbg_trace<IRIS_INT_SWSS_NAIVE>, grid, block, dyn, nullptr, c->sc, cd, fd, cap, (const uint32_t *)dx.p, n, (iris_trace *)dtr.p); break;
    default: hipLaunchKernelGGL(k_dbg_trace<IRIS_INT_SWSS_SLOW>, grid, block, dyn, nullptr, c->sc, cd, fd, cap, (const uint32_t *)dx.p, n, (iris_trace *)dtr.p); break;
#else
    default: return fail(c, IRIS_E_ARG, "this experiment build holds the HwssNaive kernels only");
#endif
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(out, dtr.p, sizeof(iris_trace) * (size_t)n, hipMemcpyDeviceToHost));
    return IRIS_OK;
}

}  // extern "C"
