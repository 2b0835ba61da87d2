/*
 * iris_oracle.c -- CPU ORACLE (test infrastructure only; see iris_oracle.h).
 *
 * Scalar f32 restatement of the reference's per-pixel path.  Compile with
 *   gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math
 * so that every f32 operation is a single IEEE-754 operation in the order the
 * Rust source writes it (rustc never contracts a*b+c; the only fused
 * operations in the reference are the explicit mul_add calls of
 * spectrum/upsample.rs:19-23, restated with fmaf).
 *
 * All "file:line" citations are relative to /root/reference/src/.
 */
#define _GNU_SOURCE
#include "iris_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define IRIS_TABLE_QUAL const
#include "tables/cie_cmf.inc"
#include "tables/sobol_rev16.inc"

/* ------------------------------------------------------------------------ */
/* Rust scalar semantics                                                      */
/* ------------------------------------------------------------------------ */

#define PI_F 3.14159274101257324219f /* std::f32::consts::PI */

/* f32::max / f32::min: IEEE maxNum/minNum -- a NaN operand is ignored. */
static inline float rmax(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
static inline float rmin(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
/* f32::clamp: NaN stays NaN */
static inline float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* `f as usize` (saturating, NaN -> 0) */
static inline uint64_t f_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}
/* `f as i32` (saturating, NaN -> 0) */
static inline int32_t f_as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------ */
/* sin / cos                                                                  */
/* ------------------------------------------------------------------------ */

static int g_trig_mode = ORC_TRIG_LIBM;
void orc_set_trig_mode(int mode) { g_trig_mode = mode; }

/* docs/SPEC.md "sin/cos": f64 Cody-Waite reduction by pi/2 and the fdlibm
 * kernel polynomials, every step an explicit fma or a single f64 operation,
 * result rounded once to f32.  The HIP kernels implement the same sequence. */
static double trig_reduce(double xd, int *quadrant) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI */
    double kd = rint(xd * TWO_OVER_PI);
    double r = fma(-kd, PIO2_HI, xd);
    r = fma(-kd, PIO2_LO, r);
    *quadrant = (int)((int64_t)kd & 3);
    return r;
}
static double trig_ksin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = fma(z, S6, S5);
    p = fma(z, p, S4);
    p = fma(z, p, S3);
    p = fma(z, p, S2);
    p = fma(z, p, S1);
    return fma(r * z, p, r);
}
static double trig_kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = fma(z, C6, C5);
    p = fma(z, p, C4);
    p = fma(z, p, C3);
    p = fma(z, p, C2);
    p = fma(z, p, C1);
    return fma(z * z, p, fma(z, -0.5, 1.0));
}
static float iris_sinf(float x) {
    double xd = (double)x;
    if (!(fabs(xd) < 1.0e5)) return x - x; /* NaN for NaN/inf/huge: never reached by the path */
    int q;
    double r = trig_reduce(xd, &q);
    double s = (q & 1) ? trig_kcos(r) : trig_ksin(r);
    return (float)((q & 2) ? -s : s);
}
static float iris_cosf(float x) {
    double xd = (double)x;
    if (!(fabs(xd) < 1.0e5)) return x - x;
    int q;
    double r = trig_reduce(xd, &q);
    double c = (q & 1) ? trig_ksin(r) : trig_kcos(r);
    return (float)(((q + 1) & 2) ? -c : c);
}
float orc_sinf(float x) { return g_trig_mode == ORC_TRIG_IRIS ? iris_sinf(x) : sinf(x); }
float orc_cosf(float x) { return g_trig_mode == ORC_TRIG_IRIS ? iris_cosf(x) : cosf(x); }

/* ------------------------------------------------------------------------ */
/* math/vec4.rs, vec3.rs, point3.rs                                           */
/* ------------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;          /* Vec3<S> / Point3<S> (the w lane of Vec3's Vec4 is never read) */
typedef struct { float v[4]; } f4;             /* Vec4 / SpectralSample / PdfSet / Wavelength */

static inline v3 v3_new(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_new(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vec3.rs:146-155 */
static inline v3 v3_sub(v3 a, v3 b) { return v3_new(a.x - b.x, a.y - b.y, a.z - b.z); }   /* vec3.rs:183-193, point3.rs:87-97 */
static inline v3 v3_neg(v3 a) { return v3_new(-a.x, -a.y, -a.z); }                         /* vec3.rs:290-299 */
static inline v3 v3_muls(v3 a, float s) { return v3_new(a.x * s, a.y * s, a.z * s); }     /* vec3.rs:233-243 */
static inline v3 v3_smul(float s, v3 a) { return v3_new(s * a.x, s * a.y, s * a.z); }     /* vec3.rs:221-231 */
static inline v3 v3_divs(v3 a, float s) { return v3_new(a.x / s, a.y / s, a.z / s); }     /* vec3.rs:271-281 */
/* vec3.rs:45-48 */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec3.rs:50-57 */
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_new(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_len_squared(v3 a) { return v3_dot(a, a); }       /* vec3.rs:59-61 */
static inline float v3_len(v3 a) { return sqrtf(v3_len_squared(a)); }   /* vec3.rs:63-65 */
static inline v3 v3_normalize(v3 a) { return v3_divs(a, v3_len(a)); }   /* vec3.rs:67-70 */
static inline float p3_distance_squared(v3 a, v3 b) { return v3_len_squared(v3_sub(a, b)); } /* point3.rs:48-50 */
static inline float p3_distance(v3 a, v3 b) { return sqrtf(p3_distance_squared(a, b)); }     /* point3.rs:44-46 */

/* vec3.rs:77-85 (divides by x^2+z^2, not its square root -- as written) */
static void v3_coordinate_system_from_unit(v3 s, v3 *v2_out, v3 *v3_out) {
    v3 v2;
    if (fabsf(s.x) > fabsf(s.y)) {
        v2 = v3_divs(v3_new(-s.z, 0.0f, s.x), s.x * s.x + s.z * s.z);
    } else {
        v2 = v3_divs(v3_new(0.0f, s.z, -s.y), s.y * s.y + s.z * s.z);
    }
    *v2_out = v2;
    *v3_out = v3_cross(s, v2);
}
/* vec3.rs:87-96 */
static v3 v3_spherical_direction(float sin_theta, float cos_theta, float phi, v3 x, v3 y, v3 z) {
    v3 a = v3_smul(sin_theta * orc_cosf(phi), x);
    v3 b = v3_smul(sin_theta * orc_sinf(phi), y);
    v3 c = v3_smul(cos_theta, z);
    return v3_add(v3_add(a, b), c);
}
/* vec3.rs:105-111 */
static inline v3 v3_face_forward(v3 s, v3 normal) { return v3_dot(s, normal) >= 0.0f ? s : v3_neg(s); }

/* Shading-space helpers, vec3.rs:301-355 */
static inline float sh_cos_theta(v3 w) { return w.z; }
static inline float sh_cos_2_theta(v3 w) { return w.z * w.z; }
static inline float sh_sin_2_theta(v3 w) { return rmax(1.0f - sh_cos_2_theta(w), 0.0f); }
static inline float sh_sin_theta(v3 w) { return sqrtf(sh_sin_2_theta(w)); }
static inline float sh_tan_theta(v3 w) { return sh_sin_theta(w) / sh_cos_theta(w); }
static inline float sh_tan_2_theta(v3 w) { return sh_sin_2_theta(w) / sh_cos_2_theta(w); }
static inline float sh_cos_phi(v3 w) {
    float st = sh_sin_theta(w);
    return st == 0.0f ? 0.0f : rclamp(w.x / st, -1.0f, 1.0f);
}
static inline float sh_sin_phi(v3 w) {
    float st = sh_sin_theta(w);
    return st == 0.0f ? 0.0f : rclamp(w.y / st, -1.0f, 1.0f);
}
static inline float sh_cos_2_phi(v3 w) { float c = sh_cos_phi(w); return c * c; }
static inline float sh_sin_2_phi(v3 w) { float s = sh_sin_phi(w); return s * s; }
static inline int sh_same_hemisphere(v3 a, v3 b) { return a.z * b.z > 0.0f; }

/* Vec4 / SpectralSample / PdfSet element-wise ops (vec4.rs:75-272, sample.rs:107-192, pdf.rs:78-127) */
static inline f4 f4_splat(float x) { f4 r = {{x, x, x, x}}; return r; }
static inline f4 f4_new(float a, float b, float c, float d) { f4 r = {{a, b, c, d}}; return r; }
static inline f4 f4_add(f4 a, f4 b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline f4 f4_mul(f4 a, f4 b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
static inline f4 f4_div(f4 a, f4 b) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] / b.v[i]; return r; }
static inline f4 f4_muls(f4 a, float s) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * s; return r; }
static inline f4 f4_smul(float s, f4 a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = s * a.v[i]; return r; }
static inline f4 f4_divs(f4 a, float s) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] / s; return r; }
static inline f4 f4_sdiv(float s, f4 a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = s / a.v[i]; return r; }
static inline f4 f4_sadd(float s, f4 a) { f4 r; for (int i = 0; i < 4; i++) r.v[i] = s + a.v[i]; return r; }
static inline float f4_hero(f4 a) { return a.v[0]; }
/* vec4.rs:55-57: f32x4::reduce_sum == simd_reduce_add_ordered(self, -0.0): (((-0+x)+y)+z)+w */
static inline float f4_sum(f4 a) { return ((((-0.0f) + a.v[0]) + a.v[1]) + a.v[2]) + a.v[3]; }
/* vec4.rs:51-53 */
static inline float f4_dot(f4 a, f4 b) { return f4_sum(f4_mul(a, b)); }

float orc_vec4_dot(const float a[4], const float b[4]) { return f4_dot(f4_new(a[0], a[1], a[2], a[3]), f4_new(b[0], b[1], b[2], b[3])); }
float orc_vec4_sum(const float a[4]) { return f4_sum(f4_new(a[0], a[1], a[2], a[3])); }

/* ------------------------------------------------------------------------ */
/* math/ray.rs                                                                */
/* ------------------------------------------------------------------------ */

#define RAY_EPSILON 0.001f /* ray.rs:5 */

typedef struct { v3 o, d; } ray_t;

/* ray.rs:61-85 */
static v3 offset_origin(v3 p, v3 n) {
    const float ORIGIN = 1.0f / 32.0f;
    const float FLOAT_SCALE = 1.0f / 65536.0f;
    const float INT_SCALE = 256.0f;
    int32_t of_i[3] = {f_as_i32(INT_SCALE * n.x), f_as_i32(INT_SCALE * n.y), f_as_i32(INT_SCALE * n.z)};
    float pc[3] = {p.x, p.y, p.z};
    float nc[3] = {n.x, n.y, n.z};
    float out[3];
    for (int i = 0; i < 3; i++) {
        /* i32 wrapping add, done in u32 */
        uint32_t off = (uint32_t)of_i[i];
        uint32_t bits = f_bits(pc[i]) + (pc[i] < 0.0f ? (0u - off) : off);
        float p_i = f_from_bits(bits);
        out[i] = fabsf(pc[i]) < ORIGIN ? pc[i] + FLOAT_SCALE * nc[i] : p_i;
    }
    return v3_new(out[0], out[1], out[2]);
}
void orc_offset_origin(const float p[3], const float n[3], float out[3]) {
    v3 r = offset_origin(v3_new(p[0], p[1], p[2]), v3_new(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
/* ray.rs:14-19 */
static ray_t ray_new(v3 o, v3 d) { ray_t r = {o, v3_normalize(d)}; return r; }
/* ray.rs:21-32 */
static ray_t ray_spawn(v3 o, v3 d, v3 normal) {
    v3 o2 = v3_dot(d, normal) < 0.0f ? offset_origin(o, v3_neg(normal)) : offset_origin(o, normal);
    ray_t r = {o2, v3_normalize(d)};
    return r;
}
/* ray.rs:34-45 (direction is taken from the OFFSET origin) */
static ray_t ray_spawn_to(v3 o, v3 p, v3 normal) {
    v3 o2 = v3_dot(v3_sub(p, o), normal) < 0.0f ? offset_origin(o, v3_neg(normal)) : offset_origin(o, normal);
    ray_t r = {o2, v3_normalize(v3_sub(p, o2))};
    return r;
}
/* ray.rs:55-57 */
static inline v3 ray_point_at(const ray_t *r, float t) { return v3_add(r->o, v3_muls(r->d, t)); }

/* ------------------------------------------------------------------------ */
/* math/mod.rs                                                                */
/* ------------------------------------------------------------------------ */

/* mod.rs:35-59 */
static float fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t) {
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    float ei, et, ci;
    if (cos_theta_i > 0.0f) { ei = eta_i; et = eta_t; ci = cos_theta_i; }
    else { ei = eta_t; et = eta_i; ci = fabsf(cos_theta_i); }
    float sin_theta_i = sqrtf(rmax(1.0f - ci * ci, 0.0f));
    float sin_theta_t = ei / et * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = sqrtf(rmax(1.0f - sin_theta_t * sin_theta_t, 0.0f));
    float r_par = ((et * ci) - (ei * cos_theta_t)) / ((et * ci) + (ei * cos_theta_t));
    float r_perp = ((ei * ci) - (et * cos_theta_t)) / ((ei * ci) + (et * cos_theta_t));
    return (r_par * r_par + r_perp * r_perp) / 2.0f;
}
float orc_fresnel_dielectric(float c, float ei, float et) { return fresnel_dielectric(c, ei, et); }

/* mod.rs:61-72; returns 0 for None */
static int refract(v3 wi, v3 n, float eta, v3 *out) {
    float cos_theta_i = v3_dot(n, wi);
    float sin_2_theta_i = rmax(1.0f - cos_theta_i * cos_theta_i, 0.0f);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t >= 1.0f) return 0;
    float cos_theta_t = sqrtf(1.0f - sin_2_theta_t);
    *out = v3_add(v3_smul(eta, v3_neg(wi)), v3_muls(n, eta * cos_theta_i - cos_theta_t));
    return 1;
}

/* ------------------------------------------------------------------------ */
/* math/matrix.rs                                                             */
/* ------------------------------------------------------------------------ */

/* matrix.rs:32-50 */
void orc_matrix_projection(float aspect_ratio, float z_near, float z_far, float fov, float out[16]) {
    float z_range = z_near - z_far;
    const float RADS_PER_DEG = PI_F / 180.0f; /* f32::to_radians */
    float tan_half_fov = tanf((fov / 2.0f) * RADS_PER_DEG);
    float m[16] = {1.0f / (tan_half_fov * aspect_ratio), 0, 0, 0,
                   0, 1.0f / tan_half_fov, 0, 0,
                   0, 0, (-z_near - z_far) / z_range, 2.0f * z_far * z_near / z_range,
                   0, 0, 1.0f, 0};
    memcpy(out, m, sizeof m);
}
/* matrix.rs:52-62 */
void orc_matrix_translation(const float dir[3], float out[16]) {
    float m[16] = {1, 0, 0, dir[0], 0, 1, 0, dir[1], 0, 0, 1, dir[2], 0, 0, 0, 1};
    memcpy(out, m, sizeof m);
}
/* matrix.rs:206-225 */
void orc_matrix_mul(const float a[16], const float b[16], float out[16]) {
    float m[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += a[4 * i + k] * b[4 * k + j];
            m[4 * i + j] = acc;
        }
    memcpy(out, m, sizeof m);
}
/* matrix.rs:64-141 (Gauss-Jordan with full pivoting, after pbrt-v3); returns 0 on "singular" panic */
int orc_matrix_inverse(const float min[16], float out[16]) {
    int indxc[4] = {0}, indxr[4] = {0}, ipiv[4] = {0};
    float inv[4][4];
    memcpy(inv, min, sizeof inv);
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        float big = 0.0f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (fabsf(inv[j][k]) >= big) { big = fabsf(inv[j][k]); irow = j; icol = k; }
                    } else if (ipiv[k] > 1) {
                        return 0;
                    }
                }
            }
        }
        ipiv[icol] += 1;
        if (irow != icol)
            for (int k = 0; k < 4; k++) { float t = inv[irow][k]; inv[irow][k] = inv[icol][k]; inv[icol][k] = t; }
        indxr[i] = irow;
        indxc[i] = icol;
        if (inv[icol][icol] == 0.0f) return 0;
        float pivinv = 1.0f / inv[icol][icol];
        inv[icol][icol] = 1.0f;
        for (int j = 0; j < 4; j++) inv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                float save = inv[j][icol];
                inv[j][icol] = 0.0f;
                for (int k = 0; k < 4; k++) inv[j][k] -= inv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j])
            for (int k = 0; k < 4; k++) { float t = inv[k][indxr[j]]; inv[k][indxr[j]] = inv[k][indxc[j]]; inv[k][indxc[j]] = t; }
    }
    memcpy(out, inv, sizeof inv);
    return 1;
}
/* matrix.rs:156-166 */
void orc_matrix_mul_vec(const float m[16], const float v[3], float out[3]) {
    float r[3];
    for (int i = 0; i < 3; i++) r[i] = m[4 * i + 0] * v[0] + m[4 * i + 1] * v[1] + m[4 * i + 2] * v[2];
    memcpy(out, r, sizeof r);
}
/* matrix.rs:168-196 */
void orc_matrix_mul_point(const float m[16], const float p[3], float out[3]) {
    float w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
    float r[3];
    for (int i = 0; i < 3; i++)
        r[i] = (m[4 * i + 0] * p[0] + m[4 * i + 1] * p[1] + m[4 * i + 2] * p[2] + m[4 * i + 3]) / w;
    memcpy(out, r, sizeof r);
}

/* camera.rs:10-21 */
void orc_camera_new(const float pos[3], float aspect_ratio, orc_camera *out) {
    float camera_to_clip[16], world_to_camera[16];
    orc_matrix_projection(aspect_ratio, 0.1f, 100.0f, 90.0f, camera_to_clip);
    float neg[3] = {-pos[0], -pos[1], -pos[2]};
    orc_matrix_translation(neg, world_to_camera);
    orc_matrix_mul(camera_to_clip, world_to_camera, out->world_to_clip);
    memcpy(out->position, pos, 3 * sizeof(float));
    orc_matrix_inverse(out->world_to_clip, out->clip_to_world);
}

/* ------------------------------------------------------------------------ */
/* sampling/sampler.rs                                                        */
/* ------------------------------------------------------------------------ */

/* sampler.rs:74-83 */
uint32_t orc_hash_u32(uint32_t n, uint32_t seed) {
    uint32_t hash = n;
    for (int i = 0; i < 3; i++) {
        hash *= 1936502639u;
        hash ^= hash >> 16;
        hash ^= seed;
    }
    return hash;
}
/* sampler.rs:85-88 (1.0 / u32::MAX as f32 == 2^-32) */
float orc_hash_u32_to_f32(uint32_t n, uint32_t seed) {
    const float INV_MAX = 1.0f / 4294967296.0f;
    return (float)orc_hash_u32(n, seed) * INV_MAX;
}

/* --- Sobol: this build's own definition (docs/SPEC.md); replaces the external
 * crate sobol_burley 0.5.0 called at sampler.rs:29-34.  PARITY UNPINNED. --- */
#define SOBOL_NUM_DIMENSION_SETS 64u

static inline uint32_t reverse_bits32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}
static inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
/* Laine-Karras style nested-uniform (Owen) scramble on a bit-reversed integer */
uint32_t orc_lk_scramble(uint32_t n, uint32_t seed) {
    n ^= n * 0x3d20adeau;
    n += seed;
    n *= (seed >> 16) | 1u;
    n ^= n * 0x05526c56u;
    n ^= n * 0x53a22864u;
    return n;
}
uint32_t orc_sobol_plain_u32(uint32_t index, uint32_t dimension) {
    uint32_t rev = 0;
    for (uint32_t b = 0; b < 16; b++)
        if ((index >> b) & 1u) rev ^= IRIS_SOBOL_REV16[dimension >> 2][b][dimension & 3];
    return reverse_bits32(rev);
}
void orc_sobol_sample_4d(uint32_t index, uint32_t dimension_set, uint32_t seed, float out[4]) {
    /* shuffle the sample index (Owen scramble in the reversed domain), keep 16 bits */
    uint32_t shuffled_rev = orc_lk_scramble(reverse_bits32(index), lowbias32(seed)) & 0xffff0000u;
    uint32_t x_rev[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < 16; b++) {
        if ((shuffled_rev >> (31 - b)) & 1u) {
            for (int k = 0; k < 4; k++) x_rev[k] ^= IRIS_SOBOL_REV16[dimension_set][b][k];
        }
    }
    for (uint32_t k = 0; k < 4; k++) {
        uint32_t dseed = lowbias32(seed ^ (0x9e3779b9u * (4u * dimension_set + k + 1u)));
        uint32_t v = reverse_bits32(orc_lk_scramble(x_rev[k], dseed));
        out[k] = f_from_bits((v >> 9) | 0x3f800000u) - 1.0f;
    }
}

/* sampler.rs:4-10 */
typedef struct {
    uint32_t scramble, dimension, index;
    float sample_buffer[4];
    uint32_t samples_in_buffer;
    uint32_t draws; /* oracle-only bookkeeping */
} sampler_t;

/* sampler.rs:13-21 (`x as u32`, `y as u32`, `sample_index as u32` truncate) */
static sampler_t sampler_new(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed) {
    sampler_t s;
    s.scramble = orc_hash_u32(x ^ (y << 16), seed);
    s.dimension = 0;
    s.index = sample_index;
    memset(s.sample_buffer, 0, sizeof s.sample_buffer);
    s.samples_in_buffer = 0;
    s.draws = 0;
    return s;
}
/* sampler.rs:23-50.  The reference guards with `dimension < sobol::NUM_DIMENSIONS` (256) although
 * the counter is in units of 4-D sets and the crate holds 64 sets; sets 64..255 have no defined
 * reference behaviour, so the hash fallback of sampler.rs:36-41 starts at set 64 (docs/SPEC.md). */
static float sampler_gen_0_1(sampler_t *s) {
    s->draws++;
    if (s->samples_in_buffer > 0) {
        s->samples_in_buffer -= 1;
        return s->sample_buffer[s->samples_in_buffer];
    }
    if (s->dimension < SOBOL_NUM_DIMENSION_SETS) {
        orc_sobol_sample_4d(s->index, s->dimension, orc_hash_u32(s->dimension, s->scramble), s->sample_buffer);
    } else {
        for (uint32_t k = 0; k < 4; k++)
            s->sample_buffer[k] = orc_hash_u32_to_f32((s->dimension + k) ^ (s->index << 16), s->scramble);
    }
    s->dimension += 1;
    s->samples_in_buffer = 3;
    return s->sample_buffer[3];
}
/* sampler.rs:52-55 */
static float sampler_gen_range(sampler_t *s, float lower, float upper) {
    return sampler_gen_0_1(s) * (upper - lower) + lower;
}
/* sampler.rs:57-60 */
static uint64_t sampler_gen_array_index(sampler_t *s, uint64_t len) {
    return f_as_usize(sampler_gen_range(s, 0.0f, (float)len - 0.5f));
}
/* sampler.rs:65-71 (u32::MAX as f32 == 2^32) */
static float sampler_gen_golden_ratio(const sampler_t *s) {
    uint32_t uniform_integer = (s->index + s->scramble) * 2654435769u;
    return (float)uniform_integer / 4294967296.0f;
}

void orc_sampler_draws(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed, uint32_t n, float *out) {
    sampler_t s = sampler_new(x, y, sample_index, seed);
    for (uint32_t i = 0; i < n; i++) out[i] = sampler_gen_0_1(&s);
}
float orc_golden_ratio(uint32_t x, uint32_t y, uint32_t sample_index, uint32_t seed) {
    sampler_t s = sampler_new(x, y, sample_index, seed);
    return sampler_gen_golden_ratio(&s);
}

/* ------------------------------------------------------------------------ */
/* sampling/mod.rs warps, mis.rs                                              */
/* ------------------------------------------------------------------------ */

/* mod.rs:22-24 */
static inline float pdf_cone(float cos_theta_max) { return 1.0f / (2.0f * PI_F * (1.0f - cos_theta_max)); }
/* mod.rs:26-31 */
static v3 unit_sphere(float r1, float r2) {
    float z = 1.0f - 2.0f * r1;
    float r = sqrtf(rmax(1.0f - z * z, 0.0f));
    float phi = 2.0f * PI_F * r2;
    return v3_new(r * orc_cosf(phi), r * orc_sinf(phi), z);
}
/* mod.rs:33-48 */
static void concentric_disk(float r1, float r2, float *ox, float *oy) {
    float x_off = 2.0f * r1 - 1.0f;
    float y_off = 2.0f * r2 - 1.0f;
    if (x_off == 0.0f && y_off == 0.0f) { *ox = 0.0f; *oy = 0.0f; return; }
    float r, theta;
    if (fabsf(x_off) > fabsf(y_off)) { r = x_off; theta = PI_F / 4.0f * (y_off / x_off); }
    else { r = y_off; theta = (PI_F / 2.0f) - (PI_F / 4.0f) * (x_off / y_off); }
    *ox = r * orc_cosf(theta);
    *oy = r * orc_sinf(theta);
}
/* mod.rs:50-54 */
static v3 cosine_unit_hemisphere(float r1, float r2) {
    float x, y;
    concentric_disk(r1, r2, &x, &y);
    float z = sqrtf(rmax(1.0f - x * x - y * y, 0.0f));
    return v3_new(x, y, z);
}
/* mod.rs:56-59 */
static inline float pdf_cosine_unit_hemisphere(float cos_theta) { return cos_theta / PI_F; }

/* mis.rs:3-5 */
static inline float balance_heuristic_1(f4 f) { return f4_hero(f) / f4_sum(f); }
/* mis.rs:8-10 */
static inline float balance_heuristic_2(f4 f, f4 g) { return f4_hero(f) / f4_sum(f4_add(f, g)); }

/* ------------------------------------------------------------------------ */
/* spectrum/, color.rs                                                        */
/* ------------------------------------------------------------------------ */

#define LAMBDA_MIN_NM 360.0f
#define LAMBDA_MAX_NM 830.0f
#define LAMBDA_RANGE_NM (LAMBDA_MAX_NM - LAMBDA_MIN_NM)

/* wavelength.rs:55-64 */
static float rotate_n(float hero, uint32_t n) {
    float lambda = hero + (LAMBDA_RANGE_NM / 4.0f) * (float)n;
    return lambda >= LAMBDA_MAX_NM ? lambda - LAMBDA_RANGE_NM : lambda;
}
/* wavelength.rs:14-24 */
static f4 wavelength_new(float hero) { return f4_new(hero, rotate_n(hero, 1), rotate_n(hero, 2), rotate_n(hero, 3)); }
void orc_wavelength_new(float hero, float out[4]) { f4 w = wavelength_new(hero); memcpy(out, w.v, sizeof w.v); }
/* wavelength.rs:26-28 */
static f4 wavelength_sample(const sampler_t *s) { return wavelength_new(sampler_gen_golden_ratio(s) * LAMBDA_RANGE_NM + LAMBDA_MIN_NM); }

typedef struct { uint32_t kind; float c0, c1, c2, hdr; } spectrum_t;

/* upsample.rs:18-24 */
float orc_upsampled_evaluate_single(const float c[3], float lambda) {
    float x = fmaf(fmaf(c[0], lambda, c[1]), lambda, c[2]);
    float y = 1.0f / sqrtf(fmaf(x, x, 1.0f));
    return fmaf(0.5f * x, y, 0.5f);
}
/* spectrum/mod.rs:16-20, constant.rs:15-23, upsample.rs:16-42 */
static float spectrum_evaluate_single(const spectrum_t *s, float lambda) {
    float c[3] = {s->c0, s->c1, s->c2};
    switch (s->kind) {
    case ORC_SPEC_CONSTANT: return s->c0;
    case ORC_SPEC_UPSAMPLED: return orc_upsampled_evaluate_single(c, lambda);
    default: return s->hdr * orc_upsampled_evaluate_single(c, lambda); /* upsample.rs:35-41 */
    }
}
static f4 spectrum_evaluate(const spectrum_t *s, f4 wl) {
    return f4_new(spectrum_evaluate_single(s, wl.v[0]), spectrum_evaluate_single(s, wl.v[1]),
                  spectrum_evaluate_single(s, wl.v[2]), spectrum_evaluate_single(s, wl.v[3]));
}

/* ---- UpsampleTable, upsample.rs:5-9 and :44-165 (host side: used while a scene is built) ---- */
struct orc_upsample_table { float *coefficients; size_t n_coefficients; float *scale; size_t resolution; };

/* UpsampleTable::load, upsample.rs:108-144: "SPEC", u32 LE resolution, resolution f32 LE, then f32 LE to the end */
orc_upsample_table *orc_upsample_table_load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char sig[4], resb[4];
    if (fread(sig, 1, 4, f) != 4 || memcmp(sig, "SPEC", 4) != 0 || fread(resb, 1, 4, f) != 4) { fclose(f); return NULL; }
    orc_upsample_table *t = calloc(1, sizeof *t);
    t->resolution = (size_t)resb[0] | ((size_t)resb[1] << 8) | ((size_t)resb[2] << 16) | ((size_t)resb[3] << 24);
    t->scale = malloc(sizeof(float) * t->resolution);
    if (fread(t->scale, sizeof(float), t->resolution, f) != t->resolution) { fclose(f); free(t->scale); free(t); return NULL; }
    long at = ftell(f);
    fseek(f, 0, SEEK_END);
    long end = ftell(f);
    fseek(f, at, SEEK_SET);
    t->n_coefficients = (size_t)(end - at) / 4;
    t->coefficients = malloc(sizeof(float) * (t->n_coefficients ? t->n_coefficients : 1));
    if (fread(t->coefficients, sizeof(float), t->n_coefficients, f) != t->n_coefficients) { fclose(f); free(t->coefficients); free(t->scale); free(t); return NULL; }
    fclose(f);
    return t;
}
void orc_upsample_table_free(orc_upsample_table *t) {
    if (t) { free(t->coefficients); free(t->scale); free(t); }
}
/* upsample.rs:147-165 */
static size_t upsample_find_interval(const float *values, size_t size_in, float x) {
    size_t left = 0, last_interval = size_in - 2, size = last_interval;
    while (size > 0) {
        size_t half = size / 2, middle = left + half + 1;
        if (values[middle] < x) { left = middle; size -= half + 1; }
        else size = half;
    }
    return left < last_interval ? left : last_interval;
}
/* UpsampleTable::get_spectrum, upsample.rs:59-106.  Returns 0 where the reference would panic (index out of bounds). */
int orc_upsample_get_spectrum(const orc_upsample_table *t, const float rgb[3], float coefficients[3]) {
    size_t i = 0, res = t->resolution;
    for (size_t j = 1; j < 3; j++)
        if (rgb[j] >= rgb[i]) i = j;
    float z = rgb[i];
    float scale = (float)(res - 1) / z;
    float x = rgb[(i + 1) % 3] * scale;
    float y = rgb[(i + 2) % 3] * scale;
    size_t xi = f_as_usize(x), yi = f_as_usize(y);
    if (xi > res - 2) xi = res - 2;
    if (yi > res - 2) yi = res - 2;
    size_t zi = upsample_find_interval(t->scale, res, z);
    size_t offset = (((i * res + zi) * res + yi) * res + xi) * 3;
    size_t dx = 3, dy = 3 * res, dz = 3 * res * res;
    if (offset + dz + dy + dx + 2 >= t->n_coefficients) return 0;
    float x1 = x - (float)xi, x0 = 1.0f - x1;
    float y1 = y - (float)yi, y0 = 1.0f - y1;
    float z1 = (z - t->scale[zi]) / (t->scale[zi + 1] - t->scale[zi]), z0 = 1.0f - z1;
    const float *c = t->coefficients;
    for (int k = 0; k < 3; k++) {
        coefficients[k] = ((c[offset] * x0 + c[offset + dx] * x1) * y0 + (c[offset + dy] * x0 + c[offset + dy + dx] * x1) * y1) * z0
                        + ((c[offset + dz] * x0 + c[offset + dz + dx] * x1) * y0 + (c[offset + dz + dy] * x0 + c[offset + dz + dy + dx] * x1) * y1) * z1;
        offset += 1;
    }
    return 1;
}
/* UpsampleTable::get_spectrum_hdr, upsample.rs:46-56 */
int orc_upsample_get_spectrum_hdr(const orc_upsample_table *t, const float rgb[3], float coefficients[3], float *hdr_coefficient) {
    float max = rmax(rmax(rmax(rgb[0], rgb[1]), rgb[2]), 1.0f);
    float n[3] = {rgb[0] / max, rgb[1] / max, rgb[2] / max};
    *hdr_coefficient = rmin(max, 50.0f);
    return orc_upsample_get_spectrum(t, n, coefficients);
}

const float *orc_cie_table(int which) { return which == 0 ? IRIS_CIE_X : (which == 1 ? IRIS_CIE_Y : IRIS_CIE_Z); }

/* color.rs:21-34 */
void orc_xyz_from_wavelength(float lambda, float value, float out[3]) {
    const float CIE_Y_INTEGRAL = 116.661843131358f; /* color.rs:6 */
    uint64_t index = f_as_usize(lambda) - (uint64_t)360;
    value *= (LAMBDA_MAX_NM - LAMBDA_MIN_NM) / CIE_Y_INTEGRAL;
    out[0] = IRIS_CIE_X[index] * value;
    out[1] = IRIS_CIE_Y[index] * value;
    out[2] = IRIS_CIE_Z[index] * value;
}
/* sample.rs:45-52 */
static void spectral_to_xyz(f4 value, f4 wl, float out[3]) {
    float a[3], b[3], c[3], d[3];
    orc_xyz_from_wavelength(wl.v[0], value.v[0], a);
    orc_xyz_from_wavelength(wl.v[1], value.v[1], b);
    orc_xyz_from_wavelength(wl.v[2], value.v[2], c);
    orc_xyz_from_wavelength(wl.v[3], value.v[3], d);
    for (int i = 0; i < 3; i++) out[i] = ((a[i] + b[i]) + c[i]) + d[i];
}
/* color.rs:36-41 */
void orc_xyz_to_rgb_hdr(const float xyz[3], float rgb[3]) {
    float x = xyz[0], y = xyz[1], z = xyz[2];
    rgb[0] = 3.240479f * x - 1.537150f * y - 0.498535f * z;
    rgb[1] = -0.969256f * x + 1.875991f * y + 0.041556f * z;
    rgb[2] = 0.055648f * x - 0.204043f * y + 1.057311f * z;
}

/* ------------------------------------------------------------------------ */
/* shape/mod.rs, shape/sphere.rs                                              */
/* ------------------------------------------------------------------------ */

/* shape/mod.rs:14-21 */
typedef struct { v3 point, normal, tangeant, bitangeant; int back_face; } isect_t;
/* shape/mod.rs:24-30 */
static inline v3 world_to_shading(const isect_t *h, v3 w) {
    return v3_new(v3_dot(h->bitangeant, w), v3_dot(h->tangeant, w), v3_dot(h->normal, w));
}
/* shape/mod.rs:32-37 */
static inline v3 shading_to_world(const isect_t *h, v3 s) {
    float x = h->bitangeant.x * s.x + h->tangeant.x * s.y + h->normal.x * s.z;
    float y = h->bitangeant.y * s.x + h->tangeant.y * s.y + h->normal.y * s.z;
    float z = h->bitangeant.z * s.x + h->tangeant.z * s.y + h->normal.z * s.z;
    return v3_new(x, y, z);
}

typedef struct { v3 position; float radius; } sphere_t;
/* shape/mod.rs:55-60 */
typedef struct { sphere_t geometry; int light_index; int material_index; } primitive_t;

static __thread orc_stats *t_stats; /* per-thread counters for the flop model */

/* Where the reference's DEBUG build would have panicked: its debug_assert!s on the hot path, evaluated at a fixed
 * list of check points per shaded bounce and counted instead of aborting -- the same list the -DIRIS_DEBUG_INVARIANTS build of the
 * kernels counts (iris_device.h), so the two totals must be equal:
 *   [0] Vec3 not finite (vec3.rs:24-31): the hit's frame (point, normal, tangeant, bitangeant); the next ray (origin, direction)
 *   [1] SpectralSample negative or NaN (sample.rs:64-73): the path's radiance after the bounce's light; its throughput after the update
 *   [2] PdfSet negative or NaN (pdf.rs:49-58): the pdfs of the sampled continuation; the path pdfs after the update (HWSS)
 *   [3] the cone-sampled light point off its sphere by more than 2 % (sphere.rs:131-132) */
static uint64_t g_inv[4];
static inline void inv_count(int k) { __atomic_fetch_add(&g_inv[k], 1, __ATOMIC_RELAXED); }
static inline int v3_bad(v3 a) { return !(isfinite(a.x) && isfinite(a.y) && isfinite(a.z)); }
static inline int f4_bad(f4 a) { return !(a.v[0] >= 0.0f && a.v[1] >= 0.0f && a.v[2] >= 0.0f && a.v[3] >= 0.0f); }
void orc_invariant_counts(uint64_t out[4], int reset) {
    for (int k = 0; k < 4; k++) out[k] = reset ? __atomic_exchange_n(&g_inv[k], 0, __ATOMIC_RELAXED) : __atomic_load_n(&g_inv[k], __ATOMIC_RELAXED);
}

/* sphere.rs:25-74 */
static int sphere_intersect(const sphere_t *sp, const ray_t *ray, isect_t *hit, float *t_out) {
    if (t_stats) t_stats->sphere_tests++;
    v3 oc = v3_sub(ray->o, sp->position);
    float a = v3_len_squared(ray->d);
    float half_b = v3_dot(ray->d, oc);
    float c = v3_len_squared(oc) - sp->radius * sp->radius;
    float discrim = half_b * half_b - a * c;
    if (discrim > 0.0f) {
        float root = sqrtf(discrim);
        float temp = (-half_b - root) / a;
        if (!(temp > 0.0f)) {
            temp = (-half_b + root) / a;
            if (!(temp > 0.0f)) return 0;
        }
        v3 point = ray_point_at(ray, temp);
        v3 normal = v3_divs(v3_sub(point, sp->position), sp->radius);
        v3 tangeant = v3_normalize(v3_cross(v3_new(0.0f, 1.0f, 0.0f), normal));
        v3 bitangeant = v3_cross(normal, tangeant);
        hit->point = point;
        hit->normal = normal;
        hit->tangeant = tangeant;
        hit->bitangeant = bitangeant;
        hit->back_face = v3_dot(normal, ray->d) >= 0.0f;
        *t_out = temp;
        return 1;
    }
    return 0;
}

/* sphere.rs:77-135 */
static void sphere_sample(const sphere_t *sp, const isect_t *hit, sampler_t *sampler, v3 *p_out, float *pdf_out) {
    float u0 = sampler_gen_0_1(sampler);
    float u1 = sampler_gen_0_1(sampler);
    v3 point = hit->back_face ? offset_origin(hit->point, v3_neg(hit->normal)) : offset_origin(hit->point, hit->normal);
    if (p3_distance_squared(sp->position, point) <= sp->radius * sp->radius) {
        v3 light_normal = unit_sphere(u0, u1);
        v3 light_point = v3_add(sp->position, v3_smul(sp->radius, light_normal));
        float area = 4.0f * PI_F * (sp->radius * sp->radius);
        float pdf = p3_distance_squared(light_point, point) /
                    (v3_dot(light_normal, v3_normalize(v3_sub(light_point, point))) * area);
        *p_out = light_point;
        *pdf_out = rmax(pdf, 0.001f);
        return;
    }
    float dc = p3_distance(sp->position, hit->point);
    v3 wc = v3_normalize(v3_sub(sp->position, hit->point));
    v3 wc_x, wc_y;
    v3_coordinate_system_from_unit(wc, &wc_x, &wc_y);

    float sin_theta_max = sp->radius / dc;
    float sin_theta_max_2 = sin_theta_max * sin_theta_max;
    float cos_theta_max = sqrtf(rmax(1.0f - sin_theta_max_2, 0.0f));

    float sin_theta_2, cos_theta;
    if (sin_theta_max_2 < 0.00068523f) {
        sin_theta_2 = sin_theta_max_2 * u0;
        cos_theta = sqrtf(rmax(1.0f - sin_theta_2, 0.0f));
    } else {
        cos_theta = (cos_theta_max - 1.0f) * u0 + 1.0f;
        sin_theta_2 = 1.0f - cos_theta * cos_theta;
    }
    float cos_alpha = sin_theta_2 / sin_theta_max + cos_theta * sqrtf(rmax(1.0f - sin_theta_2 / sin_theta_max_2, 0.0f));
    float sin_alpha = sqrtf(rmax(1.0f - cos_alpha * cos_alpha, 0.0f));
    float phi = u1 * 2.0f * PI_F;

    v3 normal = v3_spherical_direction(sin_alpha, cos_alpha, phi, v3_neg(wc_x), v3_neg(wc_y), v3_neg(wc));
    v3 sampled_point_local = v3_smul(sp->radius, v3_normalize(normal));
    v3 sampled_point_world = v3_add(sp->position, sampled_point_local); /* local_to_world, sphere.rs:18-20 */
    {   /* sphere.rs:131-132 */
        const float dist = p3_distance(sampled_point_world, sp->position);
        if (!(dist >= sp->radius * 0.98f) || !(dist <= sp->radius * 1.02f)) inv_count(3);
    }
    *p_out = sampled_point_world;
    *pdf_out = pdf_cone(cos_theta_max);
}

/* sphere.rs:137-162 */
static float sphere_pdf(const sphere_t *sp, const isect_t *hit, v3 wi) {
    v3 point = hit->back_face ? offset_origin(hit->point, v3_neg(hit->normal)) : offset_origin(hit->point, hit->normal);
    if (p3_distance_squared(sp->position, point) <= sp->radius * sp->radius) {
        ray_t ray = ray_spawn(hit->point, wi, hit->normal);
        isect_t light_hit;
        float t;
        if (sphere_intersect(sp, &ray, &light_hit, &t)) {
            float area = 4.0f * PI_F * (sp->radius * sp->radius);
            float pdf = p3_distance_squared(light_hit.point, point) /
                        (v3_dot(light_hit.normal, v3_normalize(v3_sub(light_hit.point, point))) * area);
            return rmax(pdf, 0.001f);
        }
        return 0.0f;
    }
    float sin_theta_max_2 = sp->radius * sp->radius / p3_distance_squared(sp->position, hit->point);
    float cos_theta_max = sqrtf(rmax(1.0f - sin_theta_max_2, 0.0f));
    return pdf_cone(cos_theta_max);
}

/* ------------------------------------------------------------------------ */
/* sampling/ggx.rs                                                            */
/* ------------------------------------------------------------------------ */

/* ggx.rs:7-33 */
static v3 ggx_sample(v3 wo_in, float alpha_x, float alpha_y, sampler_t *sampler) {
    v3 wo = v3_normalize(v3_new(alpha_x * wo_in.x, alpha_y * wo_in.y, wo_in.z));
    float len_2 = wo.x * wo.x + wo.y * wo.y;
    v3 t1_v = len_2 > 0.0f ? v3_divs(v3_new(-wo.y, wo.x, 0.0f), sqrtf(len_2)) : v3_new(1.0f, 0.0f, 0.0f);
    v3 t2_v = v3_cross(wo, t1_v);
    float r = sqrtf(sampler_gen_0_1(sampler));
    float phi = 2.0f * PI_F * sampler_gen_0_1(sampler);
    float t1 = r * orc_cosf(phi);
    float t2 = r * orc_sinf(phi);
    float s = 0.5f * (1.0f + wo.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    v3 nh = v3_add(v3_add(v3_smul(t1, t1_v), v3_smul(t2, t2_v)),
                   v3_muls(wo, sqrtf(rmax(1.0f - t1 * t1 - t2 * t2, 0.0f))));
    return v3_normalize(v3_new(alpha_x * nh.x, alpha_y * nh.y, rmax(nh.z, 0.0f)));
}
/* ggx.rs:35-43 (powi(4) = (x*x)*(x*x), powi(2) = x*x) */
static float ggx_evaluate(v3 wh, float ax, float ay) {
    float tan2 = sh_tan_2_theta(wh);
    if (isinf(tan2)) return 0.0f;
    float e = (sh_cos_2_phi(wh) / (ax * ax) + sh_sin_2_phi(wh) / (ay * ay)) * tan2;
    float ct = sh_cos_theta(wh);
    float ct2 = ct * ct;
    float ope = 1.0f + e;
    return 1.0f / (PI_F * ax * ay * (ct2 * ct2) * (ope * ope));
}
/* ggx.rs:45-55 */
static float ggx_lambda(v3 w, float ax, float ay) {
    float tt = sh_tan_theta(w);
    if (isinf(tt)) return 0.0f;
    float alpha = sqrtf(sh_cos_2_phi(w) * (ax * ax) + sh_sin_2_phi(w) * (ay * ay));
    float a = alpha * fabsf(tt);
    float alpha_2_tan_2_theta = a * a;
    return (-1.0f + sqrtf(1.0f + alpha_2_tan_2_theta)) / 2.0f;
}
/* ggx.rs:57-59 */
static float ggx_g(v3 wo, v3 wh, float ax, float ay) { return 1.0f / (1.0f + ggx_lambda(wh, ax, ay) + ggx_lambda(wo, ax, ay)); }
/* ggx.rs:75-77 */
static float ggx_g1(v3 w, float ax, float ay) { return 1.0f / (1.0f + ggx_lambda(w, ax, ay)); }
/* ggx.rs:61-64 */
static float ggx_pdf(v3 wo, v3 wh, float ax, float ay) {
    return ggx_evaluate(wh, ax, ay) * ggx_g1(wo, ax, ay) * fabsf(v3_dot(wo, wh)) / fabsf(sh_cos_theta(wo));
}
/* ggx.rs:66-73 (host side, at material construction; powi(3) = x*(x*x) via __powisf2, powi(4) = (x*x)*(x*x)) */
float orc_roughness_to_alpha(float r) {
    float x = logf(rmax(r, 1e-3f));
    float x2 = x * x;
    return 1.62142f + 0.819955f * x + 0.1734f * x2 + 0.0171201f * (x * x2) + 0.000640711f * (x2 * x2);
}

/* ------------------------------------------------------------------------ */
/* bsdf/                                                                      */
/* ------------------------------------------------------------------------ */

typedef struct { uint32_t kind; int spec_a, spec_b; float p0, p1; } bsdf_t;

struct orc_scene {
    spectrum_t *spectra; uint32_t n_spectra;
    bsdf_t *bsdfs; uint32_t n_bsdfs;          /* scene.materials (PrimIndex data) */
    int *light_spectrum; int *light_prim; uint32_t n_lights; /* scene.lights (PrimIndex{data, prim_index}) */
    primitive_t *prims; uint32_t n_prims;
    /* Scene._env_map: Vec<UpsampledHdrSpectrum> (scene.rs:27), never filled by the reference.  Layout and
     * sampling tables are this build's definition (docs/SPEC.md section 6). */
    uint32_t env_w, env_h;
    float *env_texels;   /* 4 per texel: c0, c1, c2, hdr_coefficient */
    float *env_func;     /* importance * texel solid-angle weight */
    float *env_cond_cdf; /* (env_w + 1) per row */
    float *env_marg_cdf; /* env_h + 1 */
    float env_integral;
};

/* lambertian.rs:24-31 */
static f4 lambertian_evaluate(const orc_scene *sc, const bsdf_t *b, f4 wl) {
    return f4_divs(spectrum_evaluate(&sc->spectra[b->spec_a], wl), PI_F);
}
/* lambertian.rs:33-35 */
static f4 lambertian_pdf(v3 wi) { return f4_splat(pdf_cosine_unit_hemisphere(fabsf(sh_cos_theta(wi)))); }

/* microfacet.rs:91-93 */
static v3 reflect(v3 wo, v3 n) { return v3_add(v3_neg(wo), v3_smul(2.0f * v3_dot(wo, n), n)); }
/* microfacet.rs:33-57 */
static f4 microfacet_evaluate(const orc_scene *sc, const bsdf_t *b, v3 wi, v3 wo, f4 wl) {
    float cos_theta_o = fabsf(sh_cos_theta(wo));
    float cos_theta_i = fabsf(sh_cos_theta(wi));
    v3 wh = v3_add(wo, wi);
    if (!sh_same_hemisphere(wi, wo) || cos_theta_o == 0.0f || cos_theta_i == 0.0f ||
        (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f))
        return f4_splat(0.0f);
    wh = v3_normalize(wh);
    v3 wh_facing = v3_face_forward(wh, v3_new(0.0f, 0.0f, 1.0f));
    float d = ggx_evaluate(wh, b->p0, b->p1);
    float f = fresnel_dielectric(v3_dot(wi, wh_facing), 1.5f, 1.0f);
    float g = ggx_g(wo, wh, b->p0, b->p1);
    f4 r = spectrum_evaluate(&sc->spectra[b->spec_a], wl);
    return f4_divs(f4_muls(f4_muls(f4_muls(r, d), g), f), 4.0f * cos_theta_o * cos_theta_i);
}
/* microfacet.rs:59-63 */
static f4 microfacet_pdf(const bsdf_t *b, v3 wi, v3 wo) {
    v3 wh = v3_normalize(v3_add(wi, wo));
    return f4_splat(ggx_pdf(wo, wh, b->p0, b->p1) / (4.0f * v3_dot(wo, wh)));
}

/* SampleableBsdf::evaluate dispatch, bsdf/mod.rs:23-55 */
static f4 bsdf_evaluate(const orc_scene *sc, const bsdf_t *b, v3 wi, v3 wo, f4 wl) {
    switch (b->kind) {
    case ORC_BSDF_LAMBERTIAN: return lambertian_evaluate(sc, b, wl);
    case ORC_BSDF_MICROFACET: return microfacet_evaluate(sc, b, wi, wo, wl);
    default: return f4_splat(0.0f); /* specular.rs:26-33, fresnel.rs:37-44 */
    }
}
static f4 bsdf_pdf(const bsdf_t *b, v3 wi, v3 wo) {
    switch (b->kind) {
    case ORC_BSDF_LAMBERTIAN: return lambertian_pdf(wi);
    case ORC_BSDF_MICROFACET: return microfacet_pdf(b, wi, wo);
    default: return f4_splat(0.0f); /* specular.rs:35-37, fresnel.rs:46-48 */
    }
}
static void bsdf_sample(const orc_scene *sc, const bsdf_t *b, v3 wo, f4 wl, sampler_t *sampler,
                        v3 *wi_out, f4 *val_out, f4 *pdf_out) {
    switch (b->kind) {
    case ORC_BSDF_LAMBERTIAN: { /* lambertian.rs:37-50 */
        float r1 = sampler_gen_0_1(sampler);
        float r2 = sampler_gen_0_1(sampler);
        v3 wi = cosine_unit_hemisphere(r1, r2);
        wi = sh_same_hemisphere(wo, wi) ? wi : v3_neg(wi);
        *wi_out = wi;
        *val_out = lambertian_evaluate(sc, b, wl);
        *pdf_out = lambertian_pdf(wi);
        return;
    }
    case ORC_BSDF_MICROFACET: { /* microfacet.rs:65-88 */
        v3 wh = ggx_sample(wo, b->p0, b->p1, sampler);
        v3 wi = reflect(wo, wh);
        if (sh_cos_theta(wo) == 0.0f || v3_dot(wo, wh) < 0.0f || !sh_same_hemisphere(wo, wi)) {
            *wi_out = v3_new(0.0f, 0.0f, 0.0f);
            *val_out = f4_splat(0.0f);
            *pdf_out = f4_splat(0.0f);
            return;
        }
        float pdf = ggx_pdf(wo, wh, b->p0, b->p1) / (4.0f * v3_dot(wo, wh));
        *wi_out = wi;
        *val_out = microfacet_evaluate(sc, b, wi, wo, wl);
        *pdf_out = f4_splat(pdf);
        return;
    }
    case ORC_BSDF_SPECULAR: { /* specular.rs:39-49 */
        v3 wi = v3_new(-wo.x, -wo.y, wo.z);
        *wi_out = wi;
        *val_out = f4_divs(spectrum_evaluate(&sc->spectra[b->spec_a], wl), fabsf(sh_cos_theta(wi)));
        *pdf_out = f4_splat(1.0f);
        return;
    }
    case ORC_BSDF_FRESNEL: { /* fresnel.rs:30-34, 50-102 */
        float eta_a = 1.0f;
        f4 eta_b = f4_sadd(b->p0, f4_sdiv(b->p1, f4_muls(f4_mul(wl, wl), 1e-6f)));
        f4 fresnel;
        for (int k = 0; k < 4; k++) fresnel.v[k] = fresnel_dielectric(sh_cos_theta(wo), eta_a, eta_b.v[k]);
        if (sampler_gen_0_1(sampler) < f4_hero(fresnel)) {
            v3 wi = v3_new(-wo.x, -wo.y, wo.z);
            f4 bs = f4_divs(spectrum_evaluate(&sc->spectra[b->spec_a], wl), fabsf(sh_cos_theta(wi)));
            *wi_out = wi;
            *val_out = f4_mul(bs, fresnel);
            *pdf_out = fresnel;
            return;
        }
        float eta_i, eta_t;
        if (sh_cos_theta(wo) > 0.0f) { eta_i = eta_a; eta_t = f4_hero(eta_b); }
        else { eta_i = f4_hero(eta_b); eta_t = eta_a; }
        v3 wi;
        if (refract(wo, v3_face_forward(v3_new(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, &wi)) {
            f4 ft = f4_divs(f4_muls(f4_muls(spectrum_evaluate(&sc->spectra[b->spec_b], wl), 1.0f - f4_hero(fresnel)),
                                    eta_i * eta_i), eta_t * eta_t);
            float hero_value = f4_hero(ft) / fabsf(sh_cos_theta(wi));
            *wi_out = wi;
            *val_out = f4_new(hero_value, 0.0f, 0.0f, 0.0f);
            *pdf_out = f4_new(1.0f - f4_hero(fresnel), 0.0f, 0.0f, 0.0f);
            return;
        }
        *wi_out = v3_new(0.0f, 0.0f, 0.0f);
        *val_out = f4_splat(0.0f);
        *pdf_out = f4_splat(0.0f);
        return;
    }
    default: /* null.rs: unreachable!() -- treated as an absorbing surface */
        *wi_out = v3_new(0.0f, 0.0f, 0.0f);
        *val_out = f4_splat(0.0f);
        *pdf_out = f4_splat(0.0f);
        return;
    }
}

/* ------------------------------------------------------------------------ */
/* scene.rs                                                                   */
/* ------------------------------------------------------------------------ */

orc_scene *orc_scene_new(void) { return (orc_scene *)calloc(1, sizeof(orc_scene)); }
void orc_scene_free(orc_scene *s) {
    if (!s) return;
    free(s->spectra); free(s->bsdfs); free(s->light_spectrum); free(s->light_prim); free(s->prims);
    free(s->env_texels); free(s->env_func); free(s->env_cond_cdf); free(s->env_marg_cdf); free(s);
}
uint32_t orc_scene_num_prims(const orc_scene *s) { return s->n_prims; }
int orc_scene_add_spectrum(orc_scene *s, uint32_t kind, float c0, float c1, float c2, float hdr) {
    s->spectra = (spectrum_t *)realloc(s->spectra, (s->n_spectra + 1) * sizeof(spectrum_t));
    spectrum_t sp = {kind, c0, c1, c2, hdr};
    s->spectra[s->n_spectra] = sp;
    return (int)s->n_spectra++;
}
int orc_scene_add_bsdf(orc_scene *s, uint32_t kind, int spec_a, int spec_b, float p0, float p1) {
    s->bsdfs = (bsdf_t *)realloc(s->bsdfs, (s->n_bsdfs + 1) * sizeof(bsdf_t));
    bsdf_t b = {kind, spec_a, spec_b, p0, p1};
    s->bsdfs[s->n_bsdfs] = b;
    return (int)s->n_bsdfs++;
}
/* scene.rs:71-110: add_light / add_material / add_emissive_material */
int orc_scene_add_sphere(orc_scene *s, float cx, float cy, float cz, float r, int material, int light_spectrum) {
    primitive_t p;
    p.geometry.position = v3_new(cx, cy, cz);
    p.geometry.radius = r;
    p.material_index = material;
    p.light_index = -1;
    if (light_spectrum >= 0) {
        s->light_spectrum = (int *)realloc(s->light_spectrum, (s->n_lights + 1) * sizeof(int));
        s->light_prim = (int *)realloc(s->light_prim, (s->n_lights + 1) * sizeof(int));
        s->light_spectrum[s->n_lights] = light_spectrum;
        s->light_prim[s->n_lights] = (int)s->n_prims;
        p.light_index = (int)s->n_lights++;
    }
    s->prims = (primitive_t *)realloc(s->prims, (s->n_prims + 1) * sizeof(primitive_t));
    s->prims[s->n_prims] = p;
    return (int)s->n_prims++;
}
/* scene.rs:31-69 */
void orc_scene_dummy(orc_scene *s) {
    int a05 = orc_scene_add_spectrum(s, ORC_SPEC_CONSTANT, 0.5f, 0, 0, 0);
    int e3 = orc_scene_add_spectrum(s, ORC_SPEC_CONSTANT, 3.0f, 0, 0, 0);
    int b05 = orc_scene_add_spectrum(s, ORC_SPEC_CONSTANT, 0.5f, 0, 0, 0);
    int c08 = orc_scene_add_spectrum(s, ORC_SPEC_CONSTANT, 0.8f, 0, 0, 0);
    int m0 = orc_scene_add_bsdf(s, ORC_BSDF_LAMBERTIAN, a05, -1, 0, 0);
    orc_scene_add_sphere(s, 0.0f, 2.3f, 3.0f, 1.0f, m0, e3);
    int m1 = orc_scene_add_bsdf(s, ORC_BSDF_LAMBERTIAN, b05, -1, 0, 0);
    orc_scene_add_sphere(s, 0.0f, -0.2f, 3.0f, 1.0f, m1, -1);
    int m2 = orc_scene_add_bsdf(s, ORC_BSDF_LAMBERTIAN, c08, -1, 0, 0);
    orc_scene_add_sphere(s, 0.0f, -101.5f, 2.0f, 100.0f, m2, -1);
}

/* scene.rs:116-132: closest hit, lowest index wins ties (strict <), t > 0 */
static int scene_intersection(const orc_scene *s, const ray_t *ray, isect_t *hit_out, float *t_out) {
    if (t_stats) t_stats->rays++;
    float closest_t = INFINITY;
    int closest = -1;
    for (uint32_t i = 0; i < s->n_prims; i++) {
        isect_t hit;
        float t;
        if (sphere_intersect(&s->prims[i].geometry, ray, &hit, &t) && t < closest_t && t > 0.0f) {
            closest_t = t;
            closest = (int)i;
            *hit_out = hit;
        }
    }
    *t_out = closest_t;
    return closest;
}
/* scene.rs:134-149 */
static int scene_ray_hits_point(const orc_scene *s, const ray_t *ray, v3 pos) {
    if (t_stats) t_stats->rays++;
    float closest_t = INFINITY;
    for (uint32_t i = 0; i < s->n_prims; i++) {
        isect_t hit;
        float t;
        if (sphere_intersect(&s->prims[i].geometry, ray, &hit, &t) && t < closest_t && t > 0.0f) closest_t = t;
    }
    float target_t = v3_len(v3_sub(pos, ray->o)) / v3_len(ray->d);
    return closest_t > target_t - RAY_EPSILON;
}
/* scene.rs:151-166 (pointer identity == primitive index identity) */
static int scene_ray_hits_object(const orc_scene *s, const ray_t *ray, int light_prim) {
    if (t_stats) t_stats->rays++;
    float closest_t = INFINITY;
    int closest_hit_is_obj = 0;
    for (uint32_t i = 0; i < s->n_prims; i++) {
        isect_t hit;
        float t;
        if (sphere_intersect(&s->prims[i].geometry, ray, &hit, &t) && t < closest_t && t > 0.0f) {
            closest_t = t;
            closest_hit_is_obj = ((int)i == light_prim);
        }
    }
    return closest_hit_is_obj;
}
int orc_scene_intersect(const orc_scene *s, const float o[3], const float d[3], float *t_out,
                        float point[3], float normal[3], float tangent[3], float bitangent[3], int *back_face) {
    ray_t ray = {v3_new(o[0], o[1], o[2]), v3_new(d[0], d[1], d[2])};
    isect_t hit;
    memset(&hit, 0, sizeof hit);
    float t;
    int prim = scene_intersection(s, &ray, &hit, &t);
    *t_out = t;
    if (prim >= 0) {
        point[0] = hit.point.x; point[1] = hit.point.y; point[2] = hit.point.z;
        normal[0] = hit.normal.x; normal[1] = hit.normal.y; normal[2] = hit.normal.z;
        tangent[0] = hit.tangeant.x; tangent[1] = hit.tangeant.y; tangent[2] = hit.tangeant.z;
        bitangent[0] = hit.bitangeant.x; bitangent[1] = hit.bitangeant.y; bitangent[2] = hit.bitangeant.z;
        *back_face = hit.back_face;
    }
    return prim;
}


/* ------------------------------------------------------------------------ */
/* environment map (docs/SPEC.md section 6).  The reference only stubs it:  */
/* Scene._env_map (scene.rs:27), background_emission returning 0            */
/* (scene.rs:112-114), get_spectrum_hdr's clamp "until we importance sample */
/* the env map" (upsample.rs:46-56).  No reference behaviour to match: the  */
/* definition is this build's, the oracle restates it for the GPU parity.   */
/* ------------------------------------------------------------------------ */

static inline float env_sgn(float a) { return a >= 0.0f ? 1.0f : -1.0f; }
/* octahedral square [-1,1]^2 -> point on the octahedron |x|+|y|+|z| = 1, y is the pole axis */
static v3 oct_point(float u, float v) {
    float y = (1.0f - fabsf(u)) - fabsf(v);
    if (y >= 0.0f) return v3_new(u, y, v);
    return v3_new((1.0f - fabsf(v)) * env_sgn(u), y, (1.0f - fabsf(u)) * env_sgn(v));
}
/* direction -> (u, v) and |q|^3, the Jacobian of d(omega) = du dv / |q|^3 */
static void dir_to_uv(v3 d, float *u, float *v, float *jac) {
    float s = (fabsf(d.x) + fabsf(d.y)) + fabsf(d.z);
    v3 q = v3_new(d.x / s, d.y / s, d.z / s);
    if (q.y >= 0.0f) { *u = q.x; *v = q.z; }
    else { *u = (1.0f - fabsf(q.z)) * env_sgn(q.x); *v = (1.0f - fabsf(q.x)) * env_sgn(q.z); }
    float l2 = v3_len_squared(q);
    *jac = l2 * sqrtf(l2);
}
static inline uint32_t env_texel_coord(float t, uint32_t n) {
    uint64_t i = f_as_usize((t * 0.5f + 0.5f) * (float)n);
    return i < n ? (uint32_t)i : n - 1u;
}
/* emission towards -d and the solid-angle density with which env_sample produces d */
static void env_eval(const orc_scene *sc, v3 d, f4 wl, f4 *le, float *pdf) {
    float u, v, jac;
    dir_to_uv(d, &u, &v, &jac);
    uint32_t c = env_texel_coord(u, sc->env_w), r = env_texel_coord(v, sc->env_h);
    size_t idx = (size_t)r * sc->env_w + c;
    const float *t = &sc->env_texels[4 * idx];
    spectrum_t sp = {ORC_SPEC_UPSAMPLED_HDR, t[0], t[1], t[2], t[3]};
    *le = spectrum_evaluate(&sp, wl);
    *pdf = sc->env_integral > 0.0f ? ((sc->env_func[idx] / sc->env_integral) * 0.25f) * jac : 0.0f;
}
/* piecewise-constant 1-D distribution: largest idx with cdf[idx] <= u, continuous position in [0, 1] */
static float env_sample_1d(const float *cdf, uint32_t n, float u, uint32_t *idx) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid; else hi = mid;
    }
    float den = cdf[lo + 1] - cdf[lo];
    float du = u - cdf[lo];
    if (den > 0.0f) du = du / den;
    *idx = lo;
    return ((float)lo + du) / (float)n;
}
static v3 env_sample(const orc_scene *sc, float u0, float u1) {
    uint32_t r, c;
    float vv = env_sample_1d(sc->env_marg_cdf, sc->env_h, u0, &r);
    float uu = env_sample_1d(sc->env_cond_cdf + (size_t)r * (sc->env_w + 1u), sc->env_w, u1, &c);
    return v3_normalize(oct_point(2.0f * uu - 1.0f, 2.0f * vv - 1.0f));
}
/* cdf[0..n] of n non-negative values, accumulated in binary64; returns their mean */
static double env_build_cdf(const float *func, uint32_t n, float *cdf) {
    double acc = 0.0;
    for (uint32_t i = 0; i < n; i++) acc += (double)func[i];
    double run = 0.0;
    cdf[0] = 0.0f;
    for (uint32_t i = 1; i <= n; i++) {
        run += (double)func[i - 1];
        cdf[i] = acc > 0.0 ? (float)(run / acc) : (float)i / (float)n;
    }
    cdf[n] = 1.0f;
    return acc / (double)n;
}
int orc_scene_set_envmap(orc_scene *s, uint32_t w, uint32_t h, const float *texels, const float *importance) {
    free(s->env_texels); free(s->env_func); free(s->env_cond_cdf); free(s->env_marg_cdf);
    s->env_texels = s->env_func = s->env_cond_cdf = s->env_marg_cdf = NULL;
    s->env_w = s->env_h = 0;
    s->env_integral = 0.0f;
    if (w == 0 || h == 0 || !texels) return 0;
    size_t n = (size_t)w * h;
    s->env_texels = (float *)malloc(n * 4 * sizeof(float));
    s->env_func = (float *)malloc(n * sizeof(float));
    s->env_cond_cdf = (float *)malloc((size_t)(w + 1) * h * sizeof(float));
    s->env_marg_cdf = (float *)malloc((size_t)(h + 1) * sizeof(float));
    float *row_mean = (float *)malloc(h * sizeof(float));
    memcpy(s->env_texels, texels, n * 4 * sizeof(float));
    for (uint32_t r = 0; r < h; r++) {
        for (uint32_t c = 0; c < w; c++) {
            float u = (((float)c + 0.5f) / (float)w) * 2.0f - 1.0f;
            float v = (((float)r + 0.5f) / (float)h) * 2.0f - 1.0f;
            v3 q = oct_point(u, v);
            float l2 = v3_len_squared(q);
            float imp = importance ? importance[(size_t)r * w + c] : texels[4 * ((size_t)r * w + c) + 3];
            if (!(imp > 0.0f)) imp = 0.0f; /* negative or NaN importance: never sampled */
            s->env_func[(size_t)r * w + c] = imp / (l2 * sqrtf(l2));
        }
        row_mean[r] = (float)env_build_cdf(s->env_func + (size_t)r * w, w, s->env_cond_cdf + (size_t)r * (w + 1));
    }
    s->env_integral = (float)env_build_cdf(row_mean, h, s->env_marg_cdf);
    free(row_mean);
    s->env_w = w;
    s->env_h = h;
    return 1;
}
void orc_envmap_tables(const orc_scene *s, const float **func, const float **cond_cdf, const float **marg_cdf, float *integral) {
    *func = s->env_func; *cond_cdf = s->env_cond_cdf; *marg_cdf = s->env_marg_cdf; *integral = s->env_integral;
}
void orc_oct_point(float u, float v, float out[3]) { v3 q = oct_point(u, v); out[0] = q.x; out[1] = q.y; out[2] = q.z; }
void orc_dir_to_uv(const float d[3], float uv[2], float *jac) { dir_to_uv(v3_new(d[0], d[1], d[2]), &uv[0], &uv[1], jac); }
void orc_env_eval(const orc_scene *s, const float d[3], const float wl[4], float le[4], float *pdf) {
    f4 l;
    env_eval(s, v3_new(d[0], d[1], d[2]), f4_new(wl[0], wl[1], wl[2], wl[3]), &l, pdf);
    memcpy(le, l.v, sizeof l.v);
}
void orc_env_sample(const orc_scene *s, float u0, float u1, float d[3]) { v3 r = env_sample(s, u0, u1); d[0] = r.x; d[1] = r.y; d[2] = r.z; }

/* ------------------------------------------------------------------------ */
/* integrator/                                                                */
/* ------------------------------------------------------------------------ */

typedef struct { orc_trace *trace; uint32_t max_depth, min_depth; } int_ctx;

/* hwss_naive.rs:95-157 (hwss != 0) and swss_naive.rs:92-153 (hwss == 0).
 * With an environment map (docs/SPEC.md section 6) the map is one more entry of the light list: index
 * n_lights of pick_one_light.  Its two strategies mirror the sphere light's: a direction drawn from the map's
 * CDFs (2 draws, as Sphere::sample) that must reach nothing, and the BSDF-sampled ray counted when it leaves
 * the scene; emission and density of both come from env_eval on the ray's direction. */
static f4 direct_light(const orc_scene *sc, int hwss, const bsdf_t *bsdf, const isect_t *hit, const ray_t *ray,
                       f4 wl, sampler_t *sampler) {
    f4 radiance = f4_splat(0.0f);
    v3 shading_wo = world_to_shading(hit, v3_neg(ray->d));
    const uint32_t n_total = sc->n_lights + (sc->env_w ? 1u : 0u);
    uint64_t light_idx = sampler_gen_array_index(sampler, n_total); /* scene.rs:168-172 */
    float light_pick_weight = (float)n_total;
    if (light_idx >= sc->n_lights) { /* the environment map */
        {
            float u0 = sampler_gen_0_1(sampler);
            float u1 = sampler_gen_0_1(sampler);
            v3 dir = env_sample(sc, u0, u1);
            ray_t ray_to_light = ray_spawn(hit->point, dir, hit->normal);
            f4 light_emission;
            float light_pdf;
            env_eval(sc, ray_to_light.d, wl, &light_emission, &light_pdf);
            int facing_forward = v3_dot(dir, hit->normal) > 0.0f;
            if (light_pdf > 0.0f && facing_forward != hit->back_face) {
                isect_t h2;
                float t2;
                if (scene_intersection(sc, &ray_to_light, &h2, &t2) < 0) {
                    v3 shading_wi = world_to_shading(hit, ray_to_light.d);
                    f4 bsdf_values = bsdf_evaluate(sc, bsdf, shading_wi, shading_wo, wl);
                    f4 bsdf_pdfs = bsdf_pdf(bsdf, shading_wi, shading_wo);
                    float cos_theta = fabsf(sh_cos_theta(shading_wi));
                    float mis_weight = hwss ? balance_heuristic_2(f4_splat(light_pdf), bsdf_pdfs)
                                            : light_pdf / (f4_hero(bsdf_pdfs) + light_pdf);
                    radiance = f4_add(radiance, f4_divs(f4_muls(f4_mul(f4_smul(mis_weight, light_emission), bsdf_values), cos_theta), light_pdf));
                }
            }
        }
        {
            v3 wi;
            f4 bsdf_values, bsdf_pdfs;
            bsdf_sample(sc, bsdf, shading_wo, wl, sampler, &wi, &bsdf_values, &bsdf_pdfs);
            float cos_theta = fabsf(sh_cos_theta(wi));
            ray_t ray_to_light = ray_spawn(hit->point, shading_to_world(hit, wi), hit->normal);
            if (f4_hero(bsdf_pdfs) > 0.0f) {
                isect_t h2;
                float t2;
                if (scene_intersection(sc, &ray_to_light, &h2, &t2) < 0) {
                    f4 light_emission;
                    float light_pdf;
                    env_eval(sc, ray_to_light.d, wl, &light_emission, &light_pdf);
                    float mis_weight = hwss ? balance_heuristic_2(bsdf_pdfs, f4_splat(light_pdf))
                                            : f4_hero(bsdf_pdfs) / (f4_hero(bsdf_pdfs) + light_pdf);
                    radiance = f4_add(radiance, f4_divs(f4_muls(f4_mul(f4_smul(mis_weight, light_emission), bsdf_values), cos_theta), f4_hero(bsdf_pdfs)));
                }
            }
        }
        return f4_muls(radiance, light_pick_weight);
    }
    int light_spec = sc->light_spectrum[light_idx], light_prim = sc->light_prim[light_idx];
    f4 light_emission = spectrum_evaluate(&sc->spectra[light_spec], wl);
    const sphere_t *light_geom = &sc->prims[light_prim].geometry;

    { /* sample light */
        v3 light_pos;
        float light_pdf;
        sphere_sample(light_geom, hit, sampler, &light_pos, &light_pdf);
        ray_t ray_to_light = ray_spawn_to(hit->point, light_pos, hit->normal);
        int facing_forward = v3_dot(v3_sub(light_pos, hit->point), hit->normal) > 0.0f;
        if (light_pdf > 0.0f && facing_forward != hit->back_face &&
            p3_distance_squared(light_pos, hit->point) > 0.00001f &&
            scene_ray_hits_point(sc, &ray_to_light, light_pos)) {
            v3 shading_wi = world_to_shading(hit, ray_to_light.d);
            f4 bsdf_values = bsdf_evaluate(sc, bsdf, shading_wi, shading_wo, wl);
            f4 bsdf_pdfs = bsdf_pdf(bsdf, shading_wi, shading_wo);
            float cos_theta = fabsf(sh_cos_theta(shading_wi));
            float mis_weight = hwss ? balance_heuristic_2(f4_splat(light_pdf), bsdf_pdfs)
                                    : light_pdf / (f4_hero(bsdf_pdfs) + light_pdf);
            radiance = f4_add(radiance, f4_divs(f4_muls(f4_mul(f4_smul(mis_weight, light_emission), bsdf_values), cos_theta), light_pdf));
        }
    }
    { /* sample BSDF */
        v3 wi;
        f4 bsdf_values, bsdf_pdfs;
        bsdf_sample(sc, bsdf, shading_wo, wl, sampler, &wi, &bsdf_values, &bsdf_pdfs);
        float cos_theta = fabsf(sh_cos_theta(wi));
        ray_t ray_to_light = ray_spawn(hit->point, shading_to_world(hit, wi), hit->normal);
        if (f4_hero(bsdf_pdfs) > 0.0f && scene_ray_hits_object(sc, &ray_to_light, light_prim)) {
            float light_pdf = sphere_pdf(light_geom, hit, ray_to_light.d);
            float mis_weight = hwss ? balance_heuristic_2(bsdf_pdfs, f4_splat(light_pdf))
                                    : f4_hero(bsdf_pdfs) / (f4_hero(bsdf_pdfs) + light_pdf);
            radiance = f4_add(radiance, f4_divs(f4_muls(f4_mul(f4_smul(mis_weight, light_emission), bsdf_values), cos_theta), f4_hero(bsdf_pdfs)));
        }
    }
    return f4_muls(radiance, light_pick_weight);
}

/* hwss_naive.rs:28-93, hwss_slow.rs:28-85, swss_naive.rs:28-89, swss_slow.rs:28-83 */
static f4 integrator_radiance(const orc_scene *sc, uint32_t integrator, const int_ctx *ic, ray_t ray, f4 wl, sampler_t *sampler) {
    const int hwss = (integrator == ORC_INT_HWSS_NAIVE || integrator == ORC_INT_HWSS_SLOW);
    const int naive = (integrator == ORC_INT_HWSS_NAIVE || integrator == ORC_INT_SWSS_NAIVE);
    f4 radiance = f4_splat(0.0f);
    f4 throughput = f4_splat(1.0f);
    f4 path_pdfs = f4_splat(1.0f);

    for (uint32_t bounces = 0; bounces < ic->max_depth; bounces++) {
        isect_t hit;
        float t;
        int prim_idx = scene_intersection(sc, &ray, &hit, &t);
        if (prim_idx < 0) {
            /* `None => break` (hwss_naive.rs:41-44).  With an environment map (docs/SPEC.md section 6, after the
             * commented miss branch of scene.rs:351-353) the map's emission is added under the rule the
             * integrator has for emitters it hits: at bounce 0 only with next-event estimation, always without. */
            if (sc->env_w && (!naive || bounces == 0)) {
                f4 le;
                float pdf_unused;
                env_eval(sc, ray.d, wl, &le, &pdf_unused);
                if (hwss) radiance = f4_add(radiance, f4_muls(f4_mul(throughput, le), balance_heuristic_1(path_pdfs)));
                else radiance = f4_add(radiance, f4_mul(throughput, le));
            }
            break;
        }
        const primitive_t *prim = &sc->prims[prim_idx];
        if (ic->trace && ic->trace->n_hits < ORC_TRACE_MAX_DEPTH) {
            ic->trace->hit_prim[ic->trace->n_hits] = prim_idx;
            ic->trace->hit_t[ic->trace->n_hits] = t;
            ic->trace->n_hits++;
        }
        const bsdf_t *bsdf;
        if (naive) {
            if (prim->material_index < 0) break;
            bsdf = &sc->bsdfs[prim->material_index];
            if (t_stats) t_stats->bounces++;
            if (bounces == 0 && prim->light_index >= 0) {
                f4 le = spectrum_evaluate(&sc->spectra[sc->light_spectrum[prim->light_index]], wl);
                if (hwss) radiance = f4_add(radiance, f4_muls(f4_mul(throughput, le), balance_heuristic_1(path_pdfs)));
                else radiance = f4_add(radiance, f4_mul(throughput, le));
            }
            radiance = f4_add(radiance, f4_mul(throughput, direct_light(sc, hwss, bsdf, &hit, &ray, wl, sampler)));
        } else {
            if (prim->light_index >= 0) {
                f4 le = spectrum_evaluate(&sc->spectra[sc->light_spectrum[prim->light_index]], wl);
                if (hwss) radiance = f4_add(radiance, f4_muls(f4_mul(throughput, le), balance_heuristic_1(path_pdfs)));
                else radiance = f4_add(radiance, f4_mul(throughput, le));
            }
            if (prim->material_index < 0) break;
            bsdf = &sc->bsdfs[prim->material_index];
            if (t_stats) t_stats->bounces++;
        }

        if (v3_bad(hit.point) || v3_bad(hit.normal) || v3_bad(hit.tangeant) || v3_bad(hit.bitangeant)) inv_count(0);
        if (f4_bad(radiance)) inv_count(1);

        v3 shading_wo = world_to_shading(&hit, v3_neg(ray.d));
        v3 wi;
        f4 bsdf_values, bsdf_pdfs;
        bsdf_sample(sc, bsdf, shading_wo, wl, sampler, &wi, &bsdf_values, &bsdf_pdfs);
        if (f4_bad(bsdf_pdfs)) inv_count(2);
        float cos_theta = fabsf(sh_cos_theta(wi));
        if (f4_hero(bsdf_pdfs) == 0.0f || cos_theta == 0.0f) break;

        throughput = f4_mul(throughput, f4_divs(f4_muls(bsdf_values, cos_theta), f4_hero(bsdf_pdfs)));
        if (hwss) path_pdfs = f4_mul(path_pdfs, bsdf_pdfs);
        if (f4_bad(throughput)) inv_count(1);
        if (hwss && f4_bad(path_pdfs)) inv_count(2);

        if (bounces >= ic->min_depth) {
            float p = rmin(f4_sum(throughput), 0.95f);
            if (sampler_gen_0_1(sampler) > p) break;
            throughput = f4_div(throughput, f4_splat(p));
        }
        v3 world_wi = shading_to_world(&hit, wi);
        ray = ray_spawn(hit.point, world_wi, hit.normal);
        if (v3_bad(ray.o) || v3_bad(ray.d)) inv_count(0);
    }
    if (hwss) return radiance;
    return f4_new(f4_hero(radiance), 0.0f, 0.0f, 0.0f);
}

/* ------------------------------------------------------------------------ */
/* tile.rs                                                                    */
/* ------------------------------------------------------------------------ */

/* tile.rs:134-138 */
static v3 pixel_center_clip(const orc_params *p, uint32_t x_abs, uint32_t y_abs) {
    return v3_new((((float)x_abs + 0.5f) / (float)p->width - 0.5f) * 2.0f,
                  (((float)y_abs + 0.5f) / (float)p->height - 0.5f) * -2.0f, 0.0f);
}

/* tile.rs:145-162, one iteration of the sample loop */
static void trace_one(const orc_scene *sc, const orc_camera *cam, const orc_params *p, uint32_t x, uint32_t y,
                      uint32_t sample_index, float xyz[3], orc_trace *trace) {
    v3 center = pixel_center_clip(p, x, y);
    sampler_t sampler = sampler_new(x, y, sample_index, p->seed);
    f4 hero_wavelength = wavelength_sample(&sampler);
    float j0 = 0.5f * sampler_gen_0_1(&sampler) / (float)p->width;
    float j1 = 0.5f * sampler_gen_0_1(&sampler) / (float)p->height;
    v3 target_clip = v3_add(center, v3_new(j0, j1, 0.0f));
    float tc[3] = {target_clip.x, target_clip.y, target_clip.z}, tw[3];
    orc_matrix_mul_point(cam->clip_to_world, tc, tw);
    v3 origin_world = v3_new(cam->position[0], cam->position[1], cam->position[2]);
    ray_t ray = ray_new(origin_world, v3_sub(v3_new(tw[0], tw[1], tw[2]), origin_world));
    if (trace) {
        memset(trace, 0, sizeof *trace);
        memcpy(trace->hero, hero_wavelength.v, sizeof trace->hero);
        trace->ray_o[0] = ray.o.x; trace->ray_o[1] = ray.o.y; trace->ray_o[2] = ray.o.z;
        trace->ray_d[0] = ray.d.x; trace->ray_d[1] = ray.d.y; trace->ray_d[2] = ray.d.z;
    }
    int_ctx ic = {trace, p->max_depth, p->min_depth};
    f4 radiance = integrator_radiance(sc, p->integrator, &ic, ray, hero_wavelength, &sampler);
    spectral_to_xyz(radiance, hero_wavelength, xyz);
    if (t_stats) t_stats->samples++;
    if (trace) {
        memcpy(trace->radiance, radiance.v, sizeof trace->radiance);
        memcpy(trace->xyz, xyz, 3 * sizeof(float));
        trace->draws = sampler.draws;
    }
}

void orc_trace_sample(const orc_scene *s, const orc_camera *cam, const orc_params *p,
                      uint32_t x, uint32_t y, uint32_t sample_index, orc_trace *out) {
    float xyz[3];
    trace_one(s, cam, p, x, y, sample_index, xyz, out);
}

/* tile.rs:127-166 */
void orc_pixel_xyz(const orc_scene *s, const orc_camera *cam, const orc_params *p, uint32_t x, uint32_t y, float out[3]) {
    float weight = 1.0f / (float)p->spp_total;
    float sum[3] = {0.0f, 0.0f, 0.0f};
    for (uint32_t i = 0; i < p->spp_count; i++) {
        float xyz[3];
        trace_one(s, cam, p, x, y, i + p->spp_begin, xyz, NULL);
        sum[0] += xyz[0]; sum[1] += xyz[1]; sum[2] += xyz[2];
    }
    out[0] = sum[0] * weight; out[1] = sum[1] * weight; out[2] = sum[2] * weight;
}

/* ------------------------------------------------------------------------ */
/* main.rs:57-107 worker pool over tiles, tile.rs:33-73 tile order            */
/* ------------------------------------------------------------------------ */

typedef struct { uint32_t x0, y0, w, h; float dist; } tile_t;
typedef struct {
    const orc_scene *s; const orc_camera *cam; const orc_params *p;
    tile_t *tiles; uint32_t n_tiles; uint32_t next; pthread_mutex_t mu;
    float *film_xyz, *film_rgb;
    orc_stats total;
} pool_t;

static int tile_cmp(const void *a, const void *b) {
    float da = ((const tile_t *)a)->dist, db = ((const tile_t *)b)->dist;
    return (da > db) - (da < db); /* tile.rs:182-191: render from the inside out */
}

static void *worker(void *arg) {
    pool_t *pool = (pool_t *)arg;
    orc_stats local;
    memset(&local, 0, sizeof local);
    t_stats = &local;
    for (;;) {
        pthread_mutex_lock(&pool->mu); /* main.rs:92 */
        uint32_t i = pool->next < pool->n_tiles ? pool->next++ : UINT32_MAX;
        pthread_mutex_unlock(&pool->mu);
        if (i == UINT32_MAX) break;
        const tile_t *t = &pool->tiles[i];
        for (uint32_t yy = 0; yy < t->h; yy++)
            for (uint32_t xx = 0; xx < t->w; xx++) { /* tile.rs:92-110 */
                uint32_t x = t->x0 + xx, y = t->y0 + yy;
                float xyz[3];
                orc_pixel_xyz(pool->s, pool->cam, pool->p, x, y, xyz);
                size_t idx = (size_t)x + (size_t)y * pool->p->width;
                memcpy(&pool->film_xyz[3 * idx], xyz, sizeof xyz);
                if (pool->film_rgb) {
                    float rgb[3];
                    orc_xyz_to_rgb_hdr(xyz, rgb);
                    for (int c = 0; c < 3; c++) pool->film_rgb[3 * idx + c] = rmax(rgb[c], 0.0f);
                }
            }
    }
    t_stats = NULL;
    pthread_mutex_lock(&pool->mu);
    pool->total.samples += local.samples;
    pool->total.bounces += local.bounces;
    pool->total.rays += local.rays;
    pool->total.sphere_tests += local.sphere_tests;
    pthread_mutex_unlock(&pool->mu);
    return NULL;
}

double orc_render(const orc_scene *s, const orc_camera *cam, const orc_params *p, int nthreads,
                  float *film_xyz, float *film_rgb, orc_stats *stats) {
    /* tile.rs:34-39 tile size; unlike tile.rs:45-53 every tile is clipped to the image individually,
     * so all pixels are rendered (the reference drops rows/columns when W or H is not a multiple of the tile). */
    uint32_t tw = p->width / 4 < 64 ? p->width / 4 : 64;
    uint32_t th = p->height / 4 < 64 ? p->height / 4 : 64;
    if (tw == 0) tw = 1;
    if (th == 0) th = 1;
    uint32_t nx = (p->width + tw - 1) / tw, ny = (p->height + th - 1) / th;
    pool_t pool;
    memset(&pool, 0, sizeof pool);
    pool.s = s; pool.cam = cam; pool.p = p;
    pool.n_tiles = nx * ny;
    pool.tiles = (tile_t *)malloc(sizeof(tile_t) * pool.n_tiles);
    for (uint32_t i = 0; i < pool.n_tiles; i++) {
        tile_t *t = &pool.tiles[i];
        t->x0 = (i % nx) * tw;
        t->y0 = (i / nx) * th;
        t->w = t->x0 + tw <= p->width ? tw : p->width - t->x0;
        t->h = t->y0 + th <= p->height ? th : p->height - t->y0;
        float dx = (float)((int)(t->x0 + t->w / 2) - (int)p->width / 2);
        float dy = (float)((int)(t->y0 + t->h / 2) - (int)p->height / 2);
        t->dist = sqrtf(dx * dx + dy * dy);
    }
    qsort(pool.tiles, pool.n_tiles, sizeof(tile_t), tile_cmp);
    pthread_mutex_init(&pool.mu, NULL);
    pool.film_xyz = film_xyz;
    pool.film_rgb = film_rgb;
    if (nthreads < 1) nthreads = 1;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_t *th_ids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) pthread_create(&th_ids[i], NULL, worker, &pool);
    for (int i = 0; i < nthreads; i++) pthread_join(th_ids[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th_ids);
    free(pool.tiles);
    pthread_mutex_destroy(&pool.mu);
    if (stats) *stats = pool.total;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
