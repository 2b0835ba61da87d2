// iris_host.cpp -- host-side helpers of libiris_hip.so: the scalar f32 set-up code the reference runs
// once per render on the CPU (camera matrices, roughness mapping, XYZ -> RGB).  Plain C++, no HIP.
// Compiled with -ffp-contract=off: one IEEE-754 operation per source operation, as in the reference.
#include <cmath>
#include <cstring>

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

}  // extern "C"
