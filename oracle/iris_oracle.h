/*
 * iris_oracle.h -- CPU ORACLE for the iris path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the `cpu_baseline` leg of bench.py may build, load
 * or call it.  The product (iris_amd/, libiris_hip.so) never includes, links
 * or falls back to anything in this directory.
 *
 * It is a plain-C, scalar-f32 restatement of the reference renderer's
 * per-pixel path (64/iris @ 2025-01-03, Rust), function by function; each
 * function in iris_oracle.c cites the reference file:line it follows.
 *
 * PINNING STATUS
 *   - The reference cannot be built or run in the build container (no Rust
 *     toolchain; nightly-only crate; data/srgb.coeff missing), so the oracle
 *     is pinned by restating the reference's own unit tests
 *     (math/vec4.rs:289-300, math/matrix.rs:232-250) and by audit against the
 *     Rust text.  The reference holds no golden vector for the sampler, the
 *     intersection, the BSDFs, the integrators or the image.
 *   - Sobol points: the reference calls the external crate sobol_burley 0.5.0
 *     (Cargo.lock:574-578), whose source is absent.  orc_sobol_sample_4d is
 *     this build's own Owen-scrambled Sobol (docs/SPEC.md); its direction
 *     vectors are pinned against scipy's Joe-Kuo table, its scrambling is
 *     **parity unpinned** against the crate.
 *   - sin/cos: the reference calls f32::sin/cos (platform libm).  Mode
 *     ORC_TRIG_LIBM uses this machine's libm (closest to the reference on
 *     Linux); mode ORC_TRIG_IRIS uses the f64-polynomial definition in
 *     docs/SPEC.md that the HIP kernels implement, which makes GPU-vs-oracle
 *     comparisons bit-exact end to end.
 */
#ifndef IRIS_ORACLE_H
#define IRIS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_TRIG_LIBM = 0, ORC_TRIG_IRIS = 1 };

/* enum Bsdf order, bsdf/mod.rs:49-55 */
enum { ORC_BSDF_LAMBERTIAN = 0, ORC_BSDF_MICROFACET = 1, ORC_BSDF_SPECULAR = 2, ORC_BSDF_FRESNEL = 3, ORC_BSDF_NULL = 4 };
/* enum Spectrum order, spectrum/mod.rs:25-28; the HDR variant (upsample.rs:27-42) is not in the enum */
enum { ORC_SPEC_UPSAMPLED = 0, ORC_SPEC_CONSTANT = 1, ORC_SPEC_UPSAMPLED_HDR = 2 };
/* integrator/mod.rs:8-11 */
enum { ORC_INT_HWSS_NAIVE = 0, ORC_INT_HWSS_SLOW = 1, ORC_INT_SWSS_NAIVE = 2, ORC_INT_SWSS_SLOW = 3 };

typedef struct orc_scene orc_scene;

typedef struct {
    float position[3];
    float world_to_clip[16]; /* row-major m[i][j] at 4*i+j */
    float clip_to_world[16];
} orc_camera;

typedef struct {
    uint32_t width, height;
    uint32_t spp_total;  /* render.spp: weight = 1/spp_total (tile.rs:140) */
    uint32_t spp_begin;  /* samples_so_far (tile.rs:145) */
    uint32_t spp_count;  /* samples_this_iter (tile.rs:144) */
    uint32_t seed;       /* tile.rs:15 */
    uint32_t max_depth;  /* hwss_naive.rs:16 */
    uint32_t min_depth;  /* hwss_naive.rs:17 */
    uint32_t integrator; /* ORC_INT_* */
} orc_params;

typedef struct {
    uint64_t samples;
    uint64_t bounces;       /* closest-hit scans that found a shadeable hit */
    uint64_t rays;          /* all Scene scans: intersection + ray_hits_point + ray_hits_object */
    uint64_t sphere_tests;  /* Sphere::intersect calls made by those scans */
} orc_stats;

/* One traced sample, for stage-level parity. */
#define ORC_TRACE_MAX_DEPTH 64
typedef struct {
    float hero[4];       /* the four wavelengths */
    float ray_o[3], ray_d[3]; /* primary ray */
    float radiance[4];
    float xyz[3];
    uint32_t n_hits;     /* closest-hit scans that returned a hit */
    int32_t hit_prim[ORC_TRACE_MAX_DEPTH];
    float hit_t[ORC_TRACE_MAX_DEPTH];
    uint32_t draws;      /* sampler draws consumed (gen_0_1 calls) */
} orc_trace;

void orc_set_trig_mode(int mode);
float orc_sinf(float x);
float orc_cosf(float x);

/* sampling/sampler.rs */
uint32_t orc_hash_u32(uint32_t n, uint32_t seed);
float orc_hash_u32_to_f32(uint32_t n, uint32_t seed);
void orc_sobol_sample_4d(uint32_t index, uint32_t dimension_set, uint32_t seed, float out[4]);
uint32_t orc_sobol_plain_u32(uint32_t index, uint32_t dimension); /* no scrambling; for the scipy fixture */
uint32_t orc_lk_scramble(uint32_t n, uint32_t seed);
void orc_sampler_draws(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed, uint32_t n, float *out);
float orc_golden_ratio(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed);
void orc_wavelength_new(float hero, float out[4]);

/* math */
void orc_offset_origin(const float p[3], const float n[3], float out[3]);
void orc_matrix_mul(const float a[16], const float b[16], float out[16]);
int orc_matrix_inverse(const float m[16], float out[16]);
void orc_matrix_mul_point(const float m[16], const float p[3], float out[3]);
void orc_matrix_mul_vec(const float m[16], const float v[3], float out[3]);
void orc_matrix_projection(float aspect, float z_near, float z_far, float fov, float out[16]);
void orc_matrix_translation(const float dir[3], float out[16]);
float orc_vec4_dot(const float a[4], const float b[4]);
float orc_vec4_sum(const float a[4]);
float orc_fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t);
float orc_roughness_to_alpha(float r);
/* UpsampleTable, spectrum/upsample.rs:44-165 */
typedef struct orc_upsample_table orc_upsample_table;
orc_upsample_table *orc_upsample_table_load(const char *path);
void orc_upsample_table_free(orc_upsample_table *t);
int orc_upsample_get_spectrum(const orc_upsample_table *t, const float rgb[3], float coefficients[3]);
int orc_upsample_get_spectrum_hdr(const orc_upsample_table *t, const float rgb[3], float coefficients[3], float *hdr_coefficient);

/* camera.rs */
void orc_camera_new(const float pos[3], float aspect_ratio, orc_camera *out);

/* spectrum / colour */
float orc_upsampled_evaluate_single(const float coeff[3], float lambda);
void orc_xyz_from_wavelength(float lambda, float value, float out[3]);
void orc_xyz_to_rgb_hdr(const float xyz[3], float rgb[3]);
const float *orc_cie_table(int which); /* 0=X 1=Y 2=Z, 471 entries */

/* scene.rs */
orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);
int orc_scene_add_spectrum(orc_scene *s, uint32_t kind, float c0, float c1, float c2, float hdr);
int orc_scene_add_bsdf(orc_scene *s, uint32_t kind, int spec_a, int spec_b, float p0, float p1);
/* material < 0 / light_spectrum < 0 mean None (add_light / add_material / add_emissive_material) */
int orc_scene_add_sphere(orc_scene *s, float cx, float cy, float cz, float r, int material, int light_spectrum);
void orc_scene_dummy(orc_scene *s); /* Scene::dummy, scene.rs:31-69 */
uint32_t orc_scene_num_prims(const orc_scene *s);
/* Environment map (docs/SPEC.md section 6; the reference only stubs it: scene.rs:27, :112-114): w*h texels
 * of (c0, c1, c2, hdr_coefficient) in the octahedral layout, importance (may be NULL: the hdr coefficient)
 * per texel.  w == 0 removes the map. */
int orc_scene_set_envmap(orc_scene *s, uint32_t w, uint32_t h, const float *texels, const float *importance);
void orc_envmap_tables(const orc_scene *s, const float **func, const float **cond_cdf, const float **marg_cdf, float *integral);
void orc_oct_point(float u, float v, float out[3]);
void orc_dir_to_uv(const float d[3], float uv[2], float *jac);
void orc_env_eval(const orc_scene *s, const float d[3], const float wl[4], float le[4], float *pdf);
void orc_env_sample(const orc_scene *s, float u0, float u1, float d[3]);

/* shape/sphere.rs + scene.rs scans, on explicit rays */
int orc_scene_intersect(const orc_scene *s, const float o[3], const float d[3], float *t_out,
                        float point[3], float normal[3], float tangent[3], float bitangent[3], int *back_face);

/* tile.rs:127-166, one sample */
void orc_trace_sample(const orc_scene *s, const orc_camera *cam, const orc_params *p,
                      uint32_t x, uint32_t y, uint32_t sample_index, orc_trace *out);
/* tile.rs:127-166, one pixel: xyz_sum * weight */
void orc_pixel_xyz(const orc_scene *s, const orc_camera *cam, const orc_params *p,
                   uint32_t x, uint32_t y, float out_xyz[3]);
/* main.rs:87-107 + tile.rs:75-124: whole film (all pixels), nthreads workers popping 64x64 tiles
 * centre-out.  film_xyz: W*H*3 (xyz_sum*weight); film_rgb (may be NULL): to_rgb_hdr + clamp>=0.
 * Returns wall seconds. */
double orc_render(const orc_scene *s, const orc_camera *cam, const orc_params *p, int nthreads,
                  float *film_xyz, float *film_rgb, orc_stats *stats);

/* How often, since the library was loaded or the counts were last reset, the reference's DEBUG build would have panicked on the
 * hot path (its debug_assert!s: vec3.rs:24-31, sample.rs:64-73, pdf.rs:49-58, sphere.rs:131-132), at the check points listed in
 * iris_oracle.c: out[0] non-finite Vec3, [1] negative/NaN SpectralSample, [2] negative/NaN PdfSet, [3] light point off its sphere. */
void orc_invariant_counts(uint64_t out[4], int reset);

#ifdef __cplusplus
}
#endif
#endif
