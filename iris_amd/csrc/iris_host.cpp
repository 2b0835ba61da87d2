// iris_host.cpp -- host-side helpers of libiris_hip.so: the scalar f32 set-up code the reference runs
// once per render on the CPU (camera matrices, roughness mapping, XYZ -> RGB).  Plain C++, no HIP.
// Compiled with -ffp-contract=off: one IEEE-754 operation per source operation, as in the reference.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>

#include "../../include/iris_hip.h"

namespace {

constexpr float PI_F32 = 3.14159274101257324219f;

// Matrix::projection, math/matrix.rs:32-50
void projection(float aspect_ratio, float z_near, float z_far, float fov, float m[16]) {
    const float z_range = z_near - z_far;
    const float rads_per_deg = PI_F32 / 180.0f;  // f32::to_radians
    const float tan_half_fov = std::tan((fov / 2.0f) * rads_per_deg);
    const float r[16] = {1.0f / (tan_half_fov * aspect_ratio), 0.0f, 0.0f, 0.0f,
                         0.0f, 1.0f / tan_half_fov, 0.0f, 0.0f,
                         0.0f, 0.0f, (-z_near - z_far) / z_range, 2.0f * z_far * z_near / z_range,
                         0.0f, 0.0f, 1.0f, 0.0f};
    std::memcpy(m, r, sizeof r);
}

// Matrix::translation, math/matrix.rs:52-62
void translation(float x, float y, float z, float m[16]) {
    const float r[16] = {1.0f, 0.0f, 0.0f, x, 0.0f, 1.0f, 0.0f, y, 0.0f, 0.0f, 1.0f, z, 0.0f, 0.0f, 0.0f, 1.0f};
    std::memcpy(m, r, sizeof r);
}

// &Matrix * &Matrix, math/matrix.rs:206-225
void multiply(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = acc;
        }
    std::memcpy(out, r, sizeof r);
}

// Matrix::inverse, math/matrix.rs:64-141: Gauss-Jordan elimination with full pivoting.
bool inverse(const float in[16], float out[16]) {
    int col_of[4] = {0, 0, 0, 0}, row_of[4] = {0, 0, 0, 0}, pivoted[4] = {0, 0, 0, 0};
    float m[4][4];
    std::memcpy(m, in, sizeof m);
    for (int step = 0; step < 4; step++) {
        int prow = 0, pcol = 0;
        float best = 0.0f;
        for (int r = 0; r < 4; r++) {
            if (pivoted[r] == 1) continue;
            for (int c = 0; c < 4; c++) {
                if (pivoted[c] == 0) {
                    if (std::fabs(m[r][c]) >= best) { best = std::fabs(m[r][c]); prow = r; pcol = c; }
                } else if (pivoted[c] > 1) {
                    return false;
                }
            }
        }
        pivoted[pcol] += 1;
        if (prow != pcol)
            for (int c = 0; c < 4; c++) { float tmp = m[prow][c]; m[prow][c] = m[pcol][c]; m[pcol][c] = tmp; }
        row_of[step] = prow;
        col_of[step] = pcol;
        if (m[pcol][pcol] == 0.0f) return false;
        const float pivinv = 1.0f / m[pcol][pcol];
        m[pcol][pcol] = 1.0f;
        for (int c = 0; c < 4; c++) m[pcol][c] *= pivinv;
        for (int r = 0; r < 4; r++) {
            if (r == pcol) continue;
            const float save = m[r][pcol];
            m[r][pcol] = 0.0f;
            for (int c = 0; c < 4; c++) m[r][c] -= m[pcol][c] * save;
        }
    }
    for (int step = 3; step >= 0; step--) {
        if (row_of[step] == col_of[step]) continue;
        for (int r = 0; r < 4; r++) { float tmp = m[r][row_of[step]]; m[r][row_of[step]] = m[r][col_of[step]]; m[r][col_of[step]] = tmp; }
    }
    std::memcpy(out, m, sizeof m);
    return true;
}

}  // namespace

extern "C" {

// Camera::new, camera.rs:10-21
int iris_camera_new(const float position[3], float aspect_ratio, iris_camera *out) {
    if (!position || !out || !(aspect_ratio > 0.0f)) return IRIS_E_ARG;
    float camera_to_clip[16], world_to_camera[16], world_to_clip[16];
    projection(aspect_ratio, 0.1f, 100.0f, 90.0f, camera_to_clip);
    translation(-position[0], -position[1], -position[2], world_to_camera);
    multiply(camera_to_clip, world_to_camera, world_to_clip);
    std::memcpy(out->position, position, 3 * sizeof(float));
    return inverse(world_to_clip, out->clip_to_world) ? IRIS_OK : IRIS_E_ARG;
}

// ggx::roughness_to_alpha, sampling/ggx.rs:66-73 (powi(3) = x*(x*x), powi(4) = (x*x)*(x*x))
float iris_roughness_to_alpha(float roughness) {
    const float r = (roughness != roughness) ? 1e-3f : (roughness < 1e-3f ? 1e-3f : roughness);  // f32::max(1e-3)
    const float x = std::log(r);
    const float x2 = x * x;
    return 1.62142f + 0.819955f * x + 0.1734f * x2 + 0.0171201f * (x * x2) + 0.000640711f * (x2 * x2);
}

// Xyz::to_rgb_hdr (color.rs:36-41) followed by the clamp of tile.rs:108-109
int iris_film_to_rgb(const float *xyz, size_t n_pixels, float *rgb) {
    if (!xyz || !rgb) return IRIS_E_ARG;
    for (size_t i = 0; i < n_pixels; i++) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float r = 3.240479f * x - 1.537150f * y - 0.498535f * z;
        const float g = -0.969256f * x + 1.875991f * y + 0.041556f * z;
        const float b = 0.055648f * x - 0.204043f * y + 1.057311f * z;
        // f32::max(0.0): a NaN operand is ignored
        rgb[3 * i] = (r != r) ? 0.0f : (r < 0.0f ? 0.0f : r);
        rgb[3 * i + 1] = (g != g) ? 0.0f : (g < 0.0f ? 0.0f : g);
        rgb[3 * i + 2] = (b != b) ? 0.0f : (b < 0.0f ? 0.0f : b);
    }
    return IRIS_OK;
}

// ---- UpsampleTable (spectrum/upsample.rs:5-9, :44-165): host-side, used while a scene is built ----
}  // extern "C"

struct iris_upsample_table {
    std::vector<float> coefficients, scale;
    size_t resolution = 0;
};

namespace {

// Rust `f32 as usize`: saturating, NaN -> 0
size_t f_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return ~(size_t)0;
    return (size_t)f;
}
float rust_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
float rust_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }

// find_interval, upsample.rs:147-165
size_t find_interval(const std::vector<float> &values, size_t size_in, float x) {
    size_t left = 0;
    const size_t last_interval = size_in - 2;
    size_t size = last_interval;
    while (size > 0) {
        const size_t half = size / 2, middle = left + half + 1;
        if (values[middle] < x) {
            left = middle;
            size -= half + 1;
        } else {
            size = half;
        }
    }
    return left < last_interval ? left : last_interval;
}

// UpsampleTable::get_spectrum, upsample.rs:59-106
void get_spectrum(const iris_upsample_table &t, const float rgb[3], float out[3]) {
    size_t i = 0;
    const size_t res = t.resolution;
    for (size_t j = 1; j < 3; j++)
        if (rgb[j] >= rgb[i]) i = j;
    const float z = rgb[i];
    const float scale = (float)(res - 1) / z;
    const float x = rgb[(i + 1) % 3] * scale;
    const float y = rgb[(i + 2) % 3] * scale;
    const size_t xi = std::min(f_as_usize(x), res - 2), yi = std::min(f_as_usize(y), res - 2);
    const size_t zi = find_interval(t.scale, res, z);
    size_t offset = (((i * res + zi) * res + yi) * res + xi) * 3;
    const size_t dx = 3, dy = 3 * res, dz = 3 * res * res;
    const float x1 = x - (float)xi, x0 = 1.0f - x1, y1 = y - (float)yi, y0 = 1.0f - y1;
    const float z1 = (z - t.scale[zi]) / (t.scale[zi + 1] - t.scale[zi]), z0 = 1.0f - z1;
    const float *c = t.coefficients.data();
    for (int k = 0; k < 3; k++) {
        out[k] = ((c[offset] * x0 + c[offset + dx] * x1) * y0 + (c[offset + dy] * x0 + c[offset + dy + dx] * x1) * y1) * z0 +
                 ((c[offset + dz] * x0 + c[offset + dz + dx] * x1) * y0 + (c[offset + dz + dy] * x0 + c[offset + dz + dy + dx] * x1) * y1) * z1;
        offset += 1;
    }
}

}  // namespace

extern "C" {

// UpsampleTable::load, upsample.rs:108-144.  The reference panics on a bad header and would index out of
// bounds later on a short file; here both are IRIS_E_ARG at load time.
int iris_upsample_table_load(const char *path, iris_upsample_table **out) {
    if (!path || !out) return IRIS_E_ARG;
    *out = nullptr;
    FILE *f = std::fopen(path, "rb");
    if (!f) return IRIS_E_ARG;
    unsigned char head[8];
    iris_upsample_table *t = new (std::nothrow) iris_upsample_table();
    bool ok = t && std::fread(head, 1, 8, f) == 8 && std::memcmp(head, "SPEC", 4) == 0;
    if (ok) {
        const uint32_t res = (uint32_t)head[4] | ((uint32_t)head[5] << 8) | ((uint32_t)head[6] << 16) | ((uint32_t)head[7] << 24);
        ok = res >= 2 && res <= 1024;
        if (ok) {
            t->resolution = res;
            t->scale.resize(res);
            t->coefficients.resize((size_t)9 * res * res * res);
            ok = std::fread(t->scale.data(), sizeof(float), res, f) == res &&  // little-endian host
                 std::fread(t->coefficients.data(), sizeof(float), t->coefficients.size(), f) == t->coefficients.size();
        }
    }
    std::fclose(f);
    if (!ok) {
        delete t;
        return IRIS_E_ARG;
    }
    *out = t;
    return IRIS_OK;
}

void iris_upsample_table_free(iris_upsample_table *t) { delete t; }

uint32_t iris_upsample_table_resolution(const iris_upsample_table *t) { return t ? (uint32_t)t->resolution : 0u; }

// for iris_upsample_table_upload (iris_kernels.hip): the table's arrays.  Internal: not exported, not in the header.
__attribute__((visibility("hidden"))) void iris_host_upsample_arrays(const iris_upsample_table *t, const float **coefficients, size_t *n_coefficients, const float **scale) {
    *coefficients = t->coefficients.data();
    *n_coefficients = t->coefficients.size();
    *scale = t->scale.data();
}

int iris_upsample_get_spectrum(const iris_upsample_table *t, const float rgb[3], iris_spectrum *out) {
    if (!t || !rgb || !out) return IRIS_E_ARG;
    float c[3];
    get_spectrum(*t, rgb, c);
    *out = iris_spectrum{IRIS_SPEC_UPSAMPLED, c[0], c[1], c[2], 0.0f};
    return IRIS_OK;
}

// UpsampleTable::get_spectrum_hdr, upsample.rs:46-56 (including its clamp of the HDR factor at 50)
int iris_upsample_get_spectrum_hdr(const iris_upsample_table *t, const float rgb[3], iris_spectrum *out) {
    if (!t || !rgb || !out) return IRIS_E_ARG;
    const float max = rust_max(rust_max(rust_max(rgb[0], rgb[1]), rgb[2]), 1.0f);
    const float n[3] = {rgb[0] / max, rgb[1] / max, rgb[2] / max};
    float c[3];
    get_spectrum(*t, n, c);
    *out = iris_spectrum{IRIS_SPEC_UPSAMPLED_HDR, c[0], c[1], c[2], rust_min(max, 50.0f)};
    return IRIS_OK;
}

}  // extern "C"

// ---- environment map, host side (docs/SPEC.md section 6; the reference only stubs it: scene.rs:27, :112-114) ----
namespace {

float env_sgn(float a) { return a >= 0.0f ? 1.0f : -1.0f; }
// octahedral square [-1,1]^2 -> point on the octahedron |x|+|y|+|z| = 1 (y = pole axis)
void oct_point(float u, float v, float q[3]) {
    const float y = (1.0f - std::fabs(u)) - std::fabs(v);
    if (y >= 0.0f) { q[0] = u; q[1] = y; q[2] = v; }
    else { q[0] = (1.0f - std::fabs(v)) * env_sgn(u); q[1] = y; q[2] = (1.0f - std::fabs(u)) * env_sgn(v); }
}
// cdf[0..n] of n non-negative values, accumulated in binary64; returns their mean
double build_cdf(const float *func, uint32_t n, float *cdf) {
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

}  // namespace

extern "C" {

int iris_envmap_tables(const iris_envmap_desc *m, float *func, float *cond_cdf, float *marg_cdf, float *integral) {
    if (!m || !m->texels || m->width == 0 || m->height == 0 || !func || !cond_cdf || !marg_cdf || !integral) return IRIS_E_ARG;
    const uint32_t w = m->width, h = m->height;
    std::vector<float> row_mean(h);
    for (uint32_t r = 0; r < h; r++) {
        for (uint32_t c = 0; c < w; c++) {
            const float u = (((float)c + 0.5f) / (float)w) * 2.0f - 1.0f;
            const float v = (((float)r + 0.5f) / (float)h) * 2.0f - 1.0f;
            float q[3];
            oct_point(u, v, q);
            const float l2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
            const size_t i = (size_t)r * w + c;
            float imp = m->importance ? m->importance[i] : m->texels[4 * i + 3];
            if (!(imp > 0.0f)) imp = 0.0f;  // negative or NaN importance: never sampled
            func[i] = imp / (l2 * std::sqrt(l2));  // 1 / |q|^3: the texel's solid angle up to a constant
        }
        row_mean[r] = (float)build_cdf(func + (size_t)r * w, w, cond_cdf + (size_t)r * (w + 1));
    }
    *integral = (float)build_cdf(row_mean.data(), h, marg_cdf);
    return IRIS_OK;
}

void iris_oct_to_dir(float u, float v, float dir[3]) {
    float q[3];
    oct_point(u, v, q);
    const float len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    dir[0] = q[0] / len; dir[1] = q[1] / len; dir[2] = q[2] / len;
}

void iris_dir_to_oct(const float d[3], float uv[2]) {
    const float s = (std::fabs(d[0]) + std::fabs(d[1])) + std::fabs(d[2]);
    const float qx = d[0] / s, qy = d[1] / s, qz = d[2] / s;
    if (qy >= 0.0f) { uv[0] = qx; uv[1] = qz; }
    else { uv[0] = (1.0f - std::fabs(qz)) * env_sgn(qx); uv[1] = (1.0f - std::fabs(qx)) * env_sgn(qz); }
}

int iris_envmap_from_latlong(const float *rgb, uint32_t w, uint32_t h, uint32_t ow, uint32_t oh, float *out) {
    if (!rgb || !out || w == 0 || h == 0 || ow == 0 || oh == 0) return IRIS_E_ARG;
    const double PI = 3.14159265358979323846;
    for (uint32_t r = 0; r < oh; r++) {
        for (uint32_t c = 0; c < ow; c++) {
            float d[3];
            iris_oct_to_dir((((float)c + 0.5f) / (float)ow) * 2.0f - 1.0f, (((float)r + 0.5f) / (float)oh) * 2.0f - 1.0f, d);
            const double theta = std::acos(std::max(-1.0, std::min(1.0, (double)d[1])));  // 0 at +y
            double phi = std::atan2((double)d[0], -(double)d[2]);                           // 0 at -z, towards +x
            if (phi < 0.0) phi += 2.0 * PI;
            const double fx = phi / (2.0 * PI) * w - 0.5, fy = theta / PI * h - 0.5;
            const double x0f = std::floor(fx), y0f = std::floor(fy);
            const double ax = fx - x0f, ay = fy - y0f;
            const long x0 = (long)x0f, y0 = (long)y0f;
            for (int k = 0; k < 3; k++) {
                double acc = 0.0;
                for (int dy = 0; dy < 2; dy++) {
                    for (int dx = 0; dx < 2; dx++) {
                        const long xx = ((x0 + dx) % (long)w + (long)w) % (long)w;       // longitude wraps
                        const long yy = std::max(0l, std::min((long)h - 1, y0 + dy));  // latitude clamps
                        acc += (dx ? ax : 1.0 - ax) * (dy ? ay : 1.0 - ay) * (double)rgb[3 * ((size_t)yy * w + xx) + k];
                    }
                }
                out[3 * ((size_t)r * ow + c) + k] = (float)acc;
            }
        }
    }
    return IRIS_OK;
}

int iris_envmap_texels_from_rgb(const iris_upsample_table *t, const float *rgb, size_t n, float max_hdr, float *texels, float *importance) {
    if (!t || !rgb || !texels) return IRIS_E_ARG;
    for (size_t i = 0; i < n; i++) {
        const float *p = rgb + 3 * i;
        // UpsampleTable::get_spectrum_hdr, upsample.rs:46-56, with the clamp as a parameter
        const float max = rust_max(rust_max(rust_max(p[0], p[1]), p[2]), 1.0f);
        const float nrm[3] = {p[0] / max, p[1] / max, p[2] / max};
        float c[3];
        get_spectrum(*t, nrm, c);
        texels[4 * i] = c[0]; texels[4 * i + 1] = c[1]; texels[4 * i + 2] = c[2];
        texels[4 * i + 3] = rust_min(max, max_hdr);
        if (importance) importance[i] = 0.2126f * p[0] + 0.7152f * p[1] + 0.0722f * p[2];
    }
    return IRIS_OK;
}

}  // extern "C"

// ---- Radiance RGBE pictures (.hdr): the format of the environment maps the reference's authors keep beside the ----
// ---- code (`data/*.hdr`, .gitignore:7; README.md:36 "Own PNG / HDR code" is still a TODO there)                ----
// Header lines up to an empty one ("#?RADIANCE" / "#?RGBE" first, FORMAT=32-bit_rle_rgbe), a resolution line
// "-Y h +X w" (the standard orientation: rows top to bottom, columns left to right; others are refused), then per
// scanline either the adaptive run-length form (2, 2, w_hi, w_lo, then the four channels one after the other as
// runs (count > 128: count - 128 copies of the next byte) and literals), the old run-length form (a pixel
// (1, 1, 1, n) repeats the previous pixel n << shift times) or plain pixels.  A pixel (r, g, b, e) is
// (m + 0.5) * 2^(e - 136) per channel, zero when e is 0 (Radiance's colr_color).
namespace {
struct HdrFile {
    FILE *f = nullptr;
    ~HdrFile() { if (f) std::fclose(f); }
};
bool hdr_header(FILE *f, uint32_t *w, uint32_t *h) {
    char line[512];
    if (!std::fgets(line, sizeof line, f)) return false;
    if (std::strncmp(line, "#?RADIANCE", 10) != 0 && std::strncmp(line, "#?RGBE", 6) != 0) return false;
    bool format_ok = false;
    for (;;) {
        if (!std::fgets(line, sizeof line, f)) return false;
        if (line[0] == '\n' || (line[0] == '\r' && line[1] == '\n')) break;
        if (std::strncmp(line, "FORMAT=", 7) == 0) {
            if (std::strncmp(line + 7, "32-bit_rle_rgbe", 15) != 0) return false;  // XYZE pictures are another colour space
            format_ok = true;
        }
    }
    if (!format_ok) return false;
    if (!std::fgets(line, sizeof line, f)) return false;
    unsigned long hh = 0, ww = 0;
    if (std::sscanf(line, "-Y %lu +X %lu", &hh, &ww) != 2) return false;
    if (hh == 0 || ww == 0 || hh > 65536 || ww > 65536) return false;
    *w = (uint32_t)ww;
    *h = (uint32_t)hh;
    return true;
}
bool hdr_scanline(FILE *f, uint32_t w, std::vector<unsigned char> &px /* 4 * w */) {
    unsigned char head[4];
    if (std::fread(head, 1, 4, f) != 4) return false;
    if (w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && (head[2] & 0x80) == 0) {  // adaptive run-length form
        if ((((uint32_t)head[2]) << 8 | head[3]) != w) return false;
        for (int ch = 0; ch < 4; ch++) {
            uint32_t x = 0;
            while (x < w) {
                int n = std::fgetc(f);
                if (n == EOF) return false;
                if (n > 128) {
                    n -= 128;
                    const int v = std::fgetc(f);
                    if (v == EOF || x + (uint32_t)n > w) return false;
                    for (int k = 0; k < n; k++) px[4 * (x++) + ch] = (unsigned char)v;
                } else {
                    if (n == 0 || x + (uint32_t)n > w) return false;
                    for (int k = 0; k < n; k++) {
                        const int v = std::fgetc(f);
                        if (v == EOF) return false;
                        px[4 * (x++) + ch] = (unsigned char)v;
                    }
                }
            }
        }
        return true;
    }
    // plain pixels, possibly with old-style repeats
    uint32_t x = 0;
    int shift = 0;
    unsigned char p[4] = {head[0], head[1], head[2], head[3]};
    bool have = true;
    while (x < w) {
        if (!have && std::fread(p, 1, 4, f) != 4) return false;
        have = false;
        if (p[0] == 1 && p[1] == 1 && p[2] == 1) {
            if (x == 0 || shift > 24) return false;
            const uint64_t n = (uint64_t)p[3] << shift;
            if (x + n > w) return false;
            for (uint64_t k = 0; k < n; k++, x++) std::memcpy(&px[4 * x], &px[4 * (x - 1)], 4);
            shift += 8;
        } else {
            std::memcpy(&px[4 * x], p, 4);
            x++;
            shift = 0;
        }
    }
    return true;
}
}  // namespace

extern "C" {

int iris_hdr_read(const char *path, uint32_t *width, uint32_t *height, float *rgb, size_t rgb_floats) {
    if (!path || !width || !height) return IRIS_E_ARG;
    HdrFile file;
    file.f = std::fopen(path, "rb");
    if (!file.f) return IRIS_E_ARG;
    uint32_t w = 0, h = 0;
    if (!hdr_header(file.f, &w, &h)) return IRIS_E_ARG;
    *width = w;
    *height = h;
    if (!rgb) return IRIS_OK;  // size query
    if (rgb_floats < (size_t)w * h * 3) return IRIS_E_LIMIT;
    std::vector<unsigned char> px((size_t)4 * w);
    for (uint32_t y = 0; y < h; y++) {
        if (!hdr_scanline(file.f, w, px)) return IRIS_E_ARG;
        float *row = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            const unsigned char *p = &px[4 * (size_t)x];
            if (p[3] == 0) {
                row[3 * x] = row[3 * x + 1] = row[3 * x + 2] = 0.0f;
            } else {
                const float scale = std::ldexp(1.0f, (int)p[3] - 136);
                for (int k = 0; k < 3; k++) row[3 * x + k] = ((float)p[k] + 0.5f) * scale;
            }
        }
    }
    return IRIS_OK;
}

}  // extern "C"
