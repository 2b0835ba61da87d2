/*
 * iris_hip.h -- C ABI of libiris_hip.so, the MI355X (gfx950) implementation of
 * the per-pixel path-tracing hot path of the iris spectral renderer.
 *
 * Where it sits in the reference (64/iris @ 2025-01-03, paths relative to
 * src/): the reference calls `Integrator::radiance` once per camera sample
 * (integrator/mod.rs:13-15) from the sample loop of `get_pixel_color`
 * (tile.rs:127-166), which `TileData::render` drives per pixel
 * (tile.rs:75-124) from the worker threads of main.rs:87-103.  A per-ray call
 * is the wrong granularity for a GPU, so the boundary moves one level up:
 * `iris_render*` replaces the whole `TileData::render` / `get_pixel_color` /
 * `radiance` stack for a film and a range of sample indices, and returns what
 * `get_pixel_color` returns per pixel (xyz_sum * 1/spp, tile.rs:165).
 * Scene construction (scene.rs:31-110), the camera matrices (camera.rs:10-21)
 * and image output (main.rs:116-123) stay on the host.
 *
 * Conventions: every function returns 0 on success or a negative IRIS_E_*
 * code; no exception or abort crosses the ABI; the message for the last
 * failure on a context is available from iris_last_error().  The caller owns
 * every buffer it passes; the library copies scene data at upload and owns
 * all device memory behind the opaque iris_ctx.  One iris_ctx drives one GPU;
 * calls on one context must be serialised by the caller, different contexts
 * may be used from different threads.
 */
#ifndef IRIS_HIP_H
#define IRIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRIS_OK 0
#define IRIS_E_ARG (-1)   /* invalid argument / scene description */
#define IRIS_E_HIP (-2)   /* a HIP runtime call failed */
#define IRIS_E_STATE (-3) /* call out of order (e.g. render before scene upload) */
#define IRIS_E_LIMIT (-4) /* a documented size limit was exceeded */
#define IRIS_E_RCCL (-5)  /* librccl.so could not be loaded or an RCCL call failed */

/* enum Bsdf, bsdf/mod.rs:49-55 (same order) */
enum iris_bsdf_kind {
    IRIS_BSDF_LAMBERTIAN = 0, /* lambertian.rs: spec_a = albedo */
    IRIS_BSDF_MICROFACET = 1, /* microfacet.rs: spec_a = reflectance, p0 = alpha_x, p1 = alpha_y
                                 (already mapped from roughness by iris_roughness_to_alpha) */
    IRIS_BSDF_SPECULAR = 2,   /* specular.rs: spec_a = reflected_color */
    IRIS_BSDF_FRESNEL = 3,    /* fresnel.rs: spec_a = reflected_color, spec_b = transmitted_color,
                                 p0 = base_ior, p1 = dispersion */
    IRIS_BSDF_NULL = 4        /* null.rs (unreachable!() in the reference): absorbs */
};

/* enum Spectrum, spectrum/mod.rs:25-28 (same order); HDR variant: upsample.rs:27-42 */
enum iris_spectrum_kind {
    IRIS_SPEC_UPSAMPLED = 0,    /* c0,c1,c2 = sigmoid-polynomial coefficients (upsample.rs:12-24) */
    IRIS_SPEC_CONSTANT = 1,     /* c0 = value (constant.rs) */
    IRIS_SPEC_UPSAMPLED_HDR = 2 /* c0,c1,c2 + hdr = hdr_coefficient */
};

/* integrator/mod.rs:8-11; the reference picks one at compile time (main.rs:31) */
enum iris_integrator {
    IRIS_INT_HWSS_NAIVE = 0, /* hwss_naive.rs (the reference's default) */
    IRIS_INT_HWSS_SLOW = 1,  /* hwss_slow.rs */
    IRIS_INT_SWSS_NAIVE = 2, /* swss_naive.rs */
    IRIS_INT_SWSS_SLOW = 3   /* swss_slow.rs */
};

typedef struct iris_ctx iris_ctx;

/* shape/sphere.rs:7-11 */
typedef struct { float cx, cy, cz, r; } iris_sphere;
typedef struct { uint32_t kind; float c0, c1, c2, hdr; } iris_spectrum;
typedef struct { uint32_t kind; uint32_t spec_a, spec_b; float p0, p1; } iris_bsdf;
/* scene.rs:24: lights: Vec<PrimIndex<Spectrum>> (types.rs:1-5) */
typedef struct { uint32_t spectrum; uint32_t prim; } iris_light;

/* Scene._env_map: Vec<UpsampledHdrSpectrum> (scene.rs:27).  The reference declares the field, never fills it
 * (background_emission returns 0, scene.rs:112-114) and notes that it is not importance-sampled yet
 * (upsample.rs:53-54); what the map looks like and how it is sampled is therefore this build's definition
 * (docs/SPEC.md section 6): width*height texels of UpsampledHdrSpectrum as (c0, c1, c2, hdr_coefficient),
 * row-major, in the OCTAHEDRAL layout (square [-1,1]^2 folded onto the sphere, y = pole axis: direction ->
 * texel needs one division and no inverse trigonometry, and texel solid angles vary by 5.2x at most).
 * importance: width*height non-negative weights (e.g. luminance) the sampling CDFs are built from, times
 * each texel's solid angle; NULL = the texel's hdr_coefficient. */
typedef struct {
    uint32_t width, height;
    const float *texels;
    const float *importance;
} iris_envmap_desc;

/* Flattened `Scene` (scene.rs:22-28).  prim_material[i] / prim_light[i] are
 * Primitive::material_index / light_index (shape/mod.rs:55-60), -1 for None.
 * Primitive order is the order of the reference's `primitives` Vec: the
 * closest-hit scan keeps the lowest index on ties (scene.rs:116-132). */
typedef struct {
    uint32_t n_prims;
    const iris_sphere *spheres;
    const int32_t *prim_material;
    const int32_t *prim_light;
    uint32_t n_bsdfs;
    const iris_bsdf *bsdfs;
    uint32_t n_spectra;
    const iris_spectrum *spectra;
    uint32_t n_lights;
    const iris_light *lights;
    const iris_envmap_desc *env_map; /* NULL: none (the reference's behaviour: a ray that leaves the scene carries nothing) */
} iris_scene_desc;

/* camera.rs:3-7; matrices row-major (m[i][j] at 4*i+j) */
typedef struct {
    float position[3];
    float clip_to_world[16];
} iris_camera;

/* The compile-time constants of the reference, as run-time parameters:
 * WIDTH/HEIGHT/TOTAL_SPP (main.rs:27-29), SEED (tile.rs:15),
 * MAX_DEPTH/MIN_DEPTH (hwss_naive.rs:16-17), CurrentIntegrator (main.rs:31).
 * spp_begin/spp_count are get_pixel_color's samples_so_far/samples_this_iter
 * (tile.rs:130-131): the call renders sample indices
 * [spp_begin, spp_begin+spp_count) of every pixel and scales the sum by
 * 1/spp_total, so films of disjoint ranges add up to the full render. */
typedef struct {
    uint32_t width, height;
    uint32_t spp_total;
    uint32_t spp_begin;
    uint32_t spp_count;
    uint32_t seed;
    uint32_t max_depth;
    uint32_t min_depth;
    uint32_t integrator; /* enum iris_integrator */
} iris_render_params;

/* Exact device counters of the last iris_render* call (the inputs of the
 * flop model in DESIGN.md) and its timing. */
typedef struct {
    uint64_t samples;      /* camera samples traced */
    uint64_t bounces;      /* surface interactions shaded */
    uint64_t rays;         /* Scene scans: intersection + ray_hits_point + ray_hits_object */
    uint64_t sphere_tests; /* Sphere::intersect evaluations */
    double kernel_ms;      /* HIP-event time of the path-tracing kernels (sum over launches) */
    double resolve_ms;     /* HIP-event time of the film resolve kernels */
    uint32_t launches;     /* path-tracing kernel launches */
    uint32_t chunks;       /* sample-range chunks the render was split into */
    uint32_t kernel;       /* which kernel traced the paths: enum iris_kernel_variant */
    uint32_t bvh_depth;    /* levels of the scene's hierarchy (0: none; the scans test every sphere) */
    uint64_t sky_samples;  /* of `samples`: camera samples of pixels that the sky mask proved black (counted in samples, rays
                              and sphere_tests as the reference executes them, but not traced) */
} iris_stats;

enum iris_kernel_variant {
    IRIS_KERNEL_POOL_FUSED = 0,   /* k_wavefront: pool/queue stages in one persistent kernel */
    IRIS_KERNEL_POOL_STAGES = 1,  /* k_stage_gen_extend + k_stage_shade launched alternately */
    IRIS_KERNEL_REGISTER = 2,     /* k_regwave: path state in registers, per-wave queue of primary hits in LDS */
    IRIS_KERNEL_QUEUES = 3        /* k_wfq_shade + k_wfq_trace launched alternately over path-state and ray queues in HBM */
};

/* One traced camera sample, for stage-level parity tests (iris_dbg_trace). */
#define IRIS_TRACE_MAX_DEPTH 64
typedef struct {
    float hero[4];
    float ray_o[3], ray_d[3];
    float radiance[4];
    float xyz[3];
    uint32_t n_hits;
    int32_t hit_prim[IRIS_TRACE_MAX_DEPTH];
    float hit_t[IRIS_TRACE_MAX_DEPTH];
    uint32_t draws;
} iris_trace;

/* ---- context ---------------------------------------------------------- */

/* Creates a context on HIP device `device` (replaces nothing in the reference;
 * the Rust side would hold it next to `Render`, main.rs:33-41). */
int iris_ctx_create(int device, iris_ctx **out);
void iris_ctx_destroy(iris_ctx *ctx);
const char *iris_last_error(const iris_ctx *ctx);

/* Tuning knobs, all optional.  Keys:
 *   "blocks_per_cu"   pool/queue kernels: persistent workgroups per CU (default 0 = as many as are resident at once)
 *   "slots_per_block" path-state slots per workgroup (default 1024)
 *   "result_bytes"    cap on the per-sample result buffer (default 64 GiB)
 *   "multi_kernel"    1: launch the two stages as separate kernels (for profiling);
 *                     0: one persistent kernel (default)
 *   "mode"            0: choose by scene (default: register-resident kernel for up to 64 spheres and for scenes
 *                     with a hierarchy, pool/queue kernels for longer lists scanned sphere by sphere);
 *                     1: pool/queue kernels (path state in HBM/L2, rays compacted before every scan);
 *                     2: register-resident kernel (needs the sphere list to fit the LDS window, or a hierarchy);
 *                     3: queue kernels (shading and traversal as separate kernels over path-state and ray queues in HBM;
 *                     needs a hierarchy and a scene of Lambertian BSDFs over constant spectra, for which mode 0 chooses it)
 *   "wfq_slots"       queue kernels: path-state slots in all (default 12582912; a multiple of 256)
 *   "wfq_pipes"       queue kernels: independent pipelines on streams of their own that share the slots (1..4, default 3)
 *   "wfq_batch"       queue kernels: shade/trace passes enqueued between two looks at the number of live paths (default 8)
 *   "wfq_trace_per_cu", "wfq_shade_per_cu"  queue kernels: workgroups per CU of the two kernels (defaults 3 and 2)
 *   "wfq_item_run"    queue kernels: items a workgroup claims at a time from the launch's counter (default 1024)
 *   "work_run"          register-resident kernel: items a wave claims at a time from the launch's counter (0 = auto, default)
 *   "reg_blocks_per_cu" register-resident kernel: workgroups per CU in the grid (default 8: at least what a CU holds; items are claimed from one global counter)
 *   "bvh_min_prims"   scenes with at least this many spheres get a bounding-volume hierarchy at upload and the
 *                     scans traverse it (default 65; 0 = never: every scan tests every sphere like scene.rs:116-166).
 *                     Closest hit and tie rule are the list scan's; iris_stats.sphere_tests then counts the tests made
 *   "bvh_leaf"        spheres per leaf of the hierarchy, 1..7 (default 2)
 *   "sort_kinds"      register-resident kernel, scenes with more than one kind of BSDF: 1 (default): every wave sorts its hits by
 *                     BSDF kind through stacks of path records in global memory (28 KiB per kind and wave, 448 MiB per GPU),
 *                     so that a shade iteration runs one BSDF's code with nearly all lanes; 0: an iteration shades whatever
 *                     the lanes hold (what iris_render_device_noalloc does: the stacks do not fit its workspace)
 *   "sky_mask"        1 (default): scenes of up to 64 spheres without an environment map: a pre-pass marks the pixels
 *                     whose camera samples provably hit nothing (conservative cone test in binary64); they are exactly
 *                     black in the reference and are neither traced nor stored; 0: every sample is traced
 *   "specialize"      1 (default): scenes with only Lambertian BSDFs over constant spectra (the reference's
 *                     default scene) run kernels compiled for that class; 0: always the general kernels
 * None of them changes a single bit of the result.
 */
int iris_ctx_set_option(iris_ctx *ctx, const char *key, int64_t value);

/* ---- scene / camera --------------------------------------------------- */

/* Copies the scene to the GPU.  Replaces holding `Render.scene` (main.rs:37) by reference. */
int iris_scene_upload(iris_ctx *ctx, const iris_scene_desc *scene);

/* Host-side restatement of Camera::new (camera.rs:10-21; matrix.rs:32-141,206-225). */
int iris_camera_new(const float position[3], float aspect_ratio, iris_camera *out);

/* ggx::roughness_to_alpha (sampling/ggx.rs:66-73), applied by MicrofacetBsdf::new
 * (bsdf/microfacet.rs:21-29) at scene construction. */
float iris_roughness_to_alpha(float roughness);

/* UpsampleTable (spectrum/upsample.rs:5-9): the RGB -> sigmoid-polynomial coefficient table of
 * data/srgb.coeff.  Host-side, used while a scene is built (the reference calls it from scene
 * construction only).  load = UpsampleTable::load (upsample.rs:108-144; file format in its comment and in
 * tools/rgb2spec_opt.c, which regenerates the file the reference tree lacks); get_spectrum =
 * upsample.rs:59-106 (trilinear look-up, find_interval :147-165), result kind IRIS_SPEC_UPSAMPLED;
 * get_spectrum_hdr = upsample.rs:46-56, result kind IRIS_SPEC_UPSAMPLED_HDR. */
typedef struct iris_upsample_table iris_upsample_table;
int iris_upsample_table_load(const char *path, iris_upsample_table **out);
void iris_upsample_table_free(iris_upsample_table *table);
uint32_t iris_upsample_table_resolution(const iris_upsample_table *table);
int iris_upsample_get_spectrum(const iris_upsample_table *table, const float rgb[3], iris_spectrum *out);
int iris_upsample_get_spectrum_hdr(const iris_upsample_table *table, const float rgb[3], iris_spectrum *out);

/* Environment-map helpers (host side, used while a scene is built; docs/SPEC.md section 6).
 * iris_envmap_tables: the sampling tables iris_scene_upload builds from a map -- func[w*h] (importance x
 *   texel solid-angle weight), cond_cdf[(w+1)*h] (per row), marg_cdf[h+1], *integral -- for inspection and
 *   parity tests.
 * iris_oct_to_dir / iris_dir_to_oct: the octahedral mapping itself ((u,v) in [-1,1]^2 <-> unit direction).
 * iris_envmap_from_latlong: resamples an equirectangular RGB image (row 0 = +y pole, column 0 = direction -z,
 *   turning towards +x) to the octahedral layout, bilinear, at texel centres.
 * iris_envmap_texels_from_rgb: per texel UpsampleTable::get_spectrum_hdr (upsample.rs:46-56) with its clamp
 *   of the HDR factor at `max_hdr` (the reference: 50, "until we importance sample the env map"; INFINITY:
 *   none) -> texels[4*n]; importance[n] (may be NULL) = Rec.709 luminance of the RGB value. */
int iris_envmap_tables(const iris_envmap_desc *map, float *func, float *cond_cdf, float *marg_cdf, float *integral);
void iris_oct_to_dir(float u, float v, float dir[3]);
void iris_dir_to_oct(const float dir[3], float uv[2]);
int iris_envmap_from_latlong(const float *rgb, uint32_t width, uint32_t height, uint32_t out_width, uint32_t out_height, float *out_rgb);
int iris_envmap_texels_from_rgb(const iris_upsample_table *table, const float *rgb, size_t n_texels, float max_hdr,
                                float *texels, float *importance);

/* The UpsampleTable resident in HBM and get_spectrum_hdr as a batch on the GPU (north_star: "table resident in HBM/L2").
 * The reference converts one RGB value at a time on the CPU while it builds a scene (upsample.rs:46-106); an HDR
 * environment map is a million of them.
 * iris_upsample_table_upload: copies the table's coefficients (9 * res^3 floats) and scale array to the context's GPU;
 *   they stay there until the context is destroyed or another table is uploaded.
 * iris_upsample_get_spectrum_hdr_batch: n RGB triples -> texels[4*n] = (c0, c1, c2, hdr_coefficient) per value, exactly
 *   the iris_spectrum iris_upsample_get_spectrum_hdr returns for it (bit for bit: same f32 operations in the same order,
 *   find_interval included), with the clamp of the HDR factor at `max_hdr` as in iris_envmap_texels_from_rgb;
 *   importance[n] (may be NULL) = Rec.709 luminance.  on_device != 0: rgb, texels (16-byte aligned) and importance are
 *   DEVICE pointers on the context's GPU, the kernel is enqueued on `stream` and the call returns without waiting;
 *   on_device == 0: they are HOST pointers, staged through device memory, and the call returns when texels is filled.
 *   IRIS_E_STATE without an uploaded table. */
int iris_upsample_table_upload(iris_ctx *ctx, const iris_upsample_table *table);
int iris_upsample_get_spectrum_hdr_batch(iris_ctx *ctx, const float *rgb, size_t n, float max_hdr, float *texels,
                                         float *importance, int on_device, void *stream);

/* Radiance RGBE picture (.hdr) -> linear RGB f32, rows top to bottom: the file format of the environment maps the
 * reference's authors keep beside the code (`.hdr` files under data/, .gitignore:7; its own reader is a README.md:36 TODO).
 * Call with rgb == NULL to get the size, then with a buffer of width*height*3 floats (rgb_floats = its length).
 * Reads plain, old and adaptive run-length scanlines in the standard "-Y h +X w" orientation.
 * IRIS_E_ARG: cannot open, not an RGBE picture, or damaged; IRIS_E_LIMIT: buffer too small. */
int iris_hdr_read(const char *path, uint32_t *width, uint32_t *height, float *rgb, size_t rgb_floats);

/* ---- the hot path ----------------------------------------------------- */

/* Renders sample indices [spp_begin, spp_begin+spp_count) of every pixel and
 * writes `xyz_sum * (1/spp_total)` per pixel -- the value get_pixel_color
 * returns (tile.rs:127-166) -- as 3 f32 per pixel at 3*(x + y*width)
 * (film indexing of main.rs:120-122).  Samples of a pixel are summed in
 * ascending sample-index order, as tile.rs:144-163 does.
 * iris_render: out_xyz is HOST memory (width*height*3 floats).
 * iris_render_device: d_out_xyz is DEVICE memory on the context's GPU; work is
 * enqueued on `stream` (a hipStream_t, NULL = default stream) and the call
 * returns after the stream has finished it. */
int iris_render(iris_ctx *ctx, const iris_camera *camera, const iris_render_params *params, float *out_xyz);
int iris_render_device(iris_ctx *ctx, const iris_camera *camera, const iris_render_params *params,
                       float *d_out_xyz, void *stream);

/* iris_render_device for callers that cannot allocate (harnesses and processes that forbid hipMalloc or whose
 * device memory is budgeted elsewhere).  It has no counterpart in the reference.  Like iris_render_device it waits for
 * `stream` before it returns (the stats are read back), so it cannot be called under HIP stream capture.  Same
 * arguments, same film bit for bit, same stats; but every byte of device memory the render needs beyond the scene, the inputs and d_out_xyz -- the
 * per-sample result buffer (12 bytes per camera sample of the call), the sky mask, the work counters and the per-wave
 * queues -- comes out of one static __device__ array of iris_noalloc_workspace_bytes() (64 MiB) that each GPU the
 * library runs on holds until the process ends, so the call makes no hipMalloc/hipFree (nor does it go through any
 * allocating library routine).  Limits: scenes the register-resident kernel serves (up to 64 spheres, or a
 * hierarchy); width*height*spp_count*12 bytes plus the fixed parts must fit the array, else IRIS_E_LIMIT with the
 * sizes in iris_last_error() -- render the film in several calls over disjoint sample ranges instead.  Every call on a
 * GPU shares that GPU's array (whatever the context) and finds the last call's contents in it: the call clears what
 * must start at zero itself, and calls that overlap in time on one GPU race -- serialise them. */
int iris_render_device_noalloc(iris_ctx *ctx, const iris_camera *camera, const iris_render_params *params,
                               float *d_out_xyz, void *stream);
size_t iris_noalloc_workspace_bytes(void);

/* The exchange step of a multi-GPU render.  The reference has one shared film behind a RwLock that the
 * worker threads copy their tiles into (main.rs:39, tile.rs:115-121); here each of n GPUs (one context each,
 * all in this process, all on different devices) has rendered a disjoint range of sample indices
 * (spp_begin/spp_count) into its own DEVICE film d_films[i] of n_floats floats, and this call sums them in place
 * with one ncclAllReduce(sum, f32) per GPU over xGMI: every film ends up holding the whole render (equal to the
 * single-GPU film up to f32 re-association of the per-pixel sum).  RCCL is loaded on first use; the
 * communicators are kept until iris_allreduce_shutdown().  Calls must not overlap. */
int iris_allreduce_films(iris_ctx *const *ctxs, int n, float *const *d_films, size_t n_floats);
void iris_allreduce_shutdown(void);
/* The whole multi-GPU render in one call -- the worker-thread pool and shared film of main.rs:87-107 across n
 * GPUs: context i (every one holding the same scene) renders sample indices
 * [spp_begin + i*spp_count/n, spp_begin + (i+1)*spp_count/n) of every pixel from its own host thread into a
 * device film, iris_allreduce_films sums them, and the total is copied to out_xyz (HOST, width*height*3).
 * Contexts that share a device (a rehearsal on a one-GPU machine) are summed on the host instead. */
int iris_render_multi(iris_ctx *const *ctxs, int n, const iris_camera *camera, const iris_render_params *params, float *out_xyz);
/* The split iris_render_multi uses (and iris_amd/distributed.py::sample_range, one process per GPU): share `part` of
 * `n_parts` is the contiguous block [begin, begin+count) of spp_count sample indices, begin = floor(spp_count*part/n_parts);
 * blocks differ by at most one sample and cover every index once.  Replaces the tile queue of main.rs:57-64 as the unit
 * of distribution.  Host only. */
int iris_sample_range(uint32_t spp_count, uint32_t part, uint32_t n_parts, uint32_t *begin, uint32_t *count);

/* XYZ -> linear RGB, clamped at 0: Xyz::to_rgb_hdr (color.rs:36-41) + the clamp of tile.rs:108-109.
 * Host arrays of 3*n_pixels floats. */
int iris_film_to_rgb(const float *xyz, size_t n_pixels, float *rgb);

int iris_get_stats(const iris_ctx *ctx, iris_stats *out);

/* Where the reference's DEBUG build would have panicked during the last iris_render* call: its debug_assert!s on the hot path
 * -- Vec3 components finite (math/vec3.rs:24-31), SpectralSample and PdfSet components non-negative and not NaN
 * (spectrum/sample.rs:64-73, math/pdf.rs:49-58), the cone-sampled light point within 2 % of its sphere (shape/sphere.rs:131-132) --
 * evaluated at a fixed list of check points per shaded bounce (the hit's frame and the next ray; the path's radiance after the
 * bounce's light and its throughput after the update; the sampled continuation's pdfs and the path pdfs; the light point) and
 * COUNTED instead of trapping: out[0] Vec3, out[1] SpectralSample, out[2] PdfSet, out[3] light point.  The oracle counts at the
 * same points (orc_invariant_counts), so the totals are comparable.  Only libiris_hip_debug.so (the same sources built with
 * -DIRIS_DEBUG_INVARIANTS=1: the checks cost registers, and that build runs the kernels that shade without deferring their
 * visibility queries) counts them; every other build returns IRIS_E_STATE and zeros. */
int iris_get_invariant_counts(const iris_ctx *ctx, uint64_t out[4]);

/* ---- stage-level entry points for parity tests ------------------------ */
/* xys: n triplets (x, y, sample_index). */

/* First `ndraws` values of Sampler::gen_0_1 (sampling/sampler.rs:13-50) for each triplet: out[n][ndraws]. */
int iris_dbg_sampler(iris_ctx *ctx, uint32_t n, const uint32_t *xys, uint32_t seed, uint32_t ndraws, float *out);
/* Camera sample of tile.rs:134-157: hero[n][4], ray_o[n][3], ray_d[n][3]. */
int iris_dbg_raygen(iris_ctx *ctx, const iris_camera *camera, const iris_render_params *params, uint32_t n,
                    const uint32_t *xys, float *hero, float *ray_o, float *ray_d);
/* Scene::intersection (scene.rs:116-132) on explicit rays: prim[n] (-1 = miss), t[n]. */
int iris_dbg_intersect(iris_ctx *ctx, uint32_t n, const float *ray_o, const float *ray_d, int32_t *prim, float *t);
/* One full camera sample per triplet through the same device code the render uses. */
int iris_dbg_trace(iris_ctx *ctx, const iris_camera *camera, const iris_render_params *params, uint32_t n,
                   const uint32_t *xys, iris_trace *out);

#ifdef __cplusplus
}
#endif
#endif
