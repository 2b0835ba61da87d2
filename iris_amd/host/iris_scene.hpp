// iris_scene.hpp -- C++ host-side mirror of the reference's scene-building interface (src/scene.rs,
// src/shape, src/bsdf, src/spectrum) above the C ABI of include/iris_hip.h.  Same method names and
// argument meaning as the reference, so that
//     scene.add_emissive_material(Sphere{{0, 2.3f, 3}, 1}, LambertianBsdf{ConstantSpectrum{0.5f}}, ConstantSpectrum{3});
// reads like scene.rs:43-47.  `flatten()` produces the iris_scene_desc that iris_scene_upload takes.
// The reference is compiled Rust; this image has no Rust toolchain, so the tested host caller is C++.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/iris_hip.h"

namespace iris_host {

// spectrum/constant.rs, spectrum/upsample.rs:11-42
struct Spectrum {
    iris_spectrum s;
};
inline Spectrum ConstantSpectrum(float v) { return Spectrum{{IRIS_SPEC_CONSTANT, v, 0.0f, 0.0f, 0.0f}}; }
inline Spectrum UpsampledSpectrum(float c0, float c1, float c2) { return Spectrum{{IRIS_SPEC_UPSAMPLED, c0, c1, c2, 0.0f}}; }
inline Spectrum UpsampledHdrSpectrum(float c0, float c1, float c2, float hdr) { return Spectrum{{IRIS_SPEC_UPSAMPLED_HDR, c0, c1, c2, hdr}}; }

// shape/sphere.rs:7-16
struct Sphere {
    float position[3];
    float radius;
};

// bsdf/*.rs: a material before its spectra have been given indices
struct Bsdf {
    uint32_t kind;
    Spectrum a, b;
    float p0, p1;
};
inline Bsdf LambertianBsdf(Spectrum albedo) { return Bsdf{IRIS_BSDF_LAMBERTIAN, albedo, albedo, 0.0f, 0.0f}; }
// microfacet.rs:20-30: roughness is mapped to alpha at construction
inline Bsdf MicrofacetBsdf(Spectrum reflectance, float roughness_x, float roughness_y) {
    return Bsdf{IRIS_BSDF_MICROFACET, reflectance, reflectance, iris_roughness_to_alpha(roughness_x), iris_roughness_to_alpha(roughness_y)};
}
inline Bsdf SpecularBsdf(Spectrum reflected) { return Bsdf{IRIS_BSDF_SPECULAR, reflected, reflected, 0.0f, 0.0f}; }
inline Bsdf FresnelBsdf(Spectrum reflected, Spectrum transmitted, float base_ior, float dispersion) {
    return Bsdf{IRIS_BSDF_FRESNEL, reflected, transmitted, base_ior, dispersion};
}

// PCG-XSH-RR 64/32 (O'Neill), default stream: the generator of the synthetic 10k-sphere scene
struct Pcg32 {
    uint64_t state = 0, inc = (54u << 1) | 1u;
    explicit Pcg32(uint64_t seed) {
        next();
        state += seed;
        next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ull + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    double uniform() { return (double)(next() >> 8) * (1.0 / 16777216.0); }
};

// Scene._env_map: Vec<UpsampledHdrSpectrum> (scene.rs:27; never filled by the reference).  Octahedral layout,
// (c0, c1, c2, hdr_coefficient) per texel -- include/iris_hip.h, docs/SPEC.md section 6.
struct EnvMap {
    uint32_t width = 0, height = 0;
    std::vector<float> texels, importance;
    // uniform white environment of spectral radiance `value` (zero sigmoid coefficients evaluate to 0.5)
    static EnvMap constant(float value, uint32_t w = 8, uint32_t h = 8) {
        EnvMap e;
        e.width = w; e.height = h;
        e.texels.assign((size_t)w * h * 4, 0.0f);
        for (size_t i = 0; i < (size_t)w * h; i++) e.texels[4 * i + 3] = 2.0f * value;
        return e;
    }
    // octahedral RGB image -> texels through UpsampleTable::get_spectrum_hdr (upsample.rs:46-56), no clamp
    static EnvMap from_rgb(const iris_upsample_table *table, const std::vector<float> &rgb, uint32_t w, uint32_t h) {
        EnvMap e;
        e.width = w; e.height = h;
        e.texels.resize((size_t)w * h * 4);
        e.importance.resize((size_t)w * h);
        iris_envmap_texels_from_rgb(table, rgb.data(), (size_t)w * h, INFINITY, e.texels.data(), e.importance.data());
        return e;
    }
    // the procedural HDR sky of iris_amd.EnvMap.sky: gradient over a dark ground and a sun 400x brighter
    static EnvMap sky(const iris_upsample_table *table, uint32_t w = 256, uint32_t h = 256, double sun_scale = 400.0) {
        std::vector<float> rgb((size_t)w * h * 3);
        const double PI = 3.14159265358979323846;
        double sun[3] = {0.45, 0.65, -0.6};
        const double sl = std::sqrt(sun[0] * sun[0] + sun[1] * sun[1] + sun[2] * sun[2]);
        for (double &c : sun) c /= sl;
        const double c4 = std::cos(4.0 * PI / 180.0), c2 = std::cos(2.0 * PI / 180.0);
        const double zenith[3] = {0.25, 0.45, 1.0}, horizon[3] = {0.9, 0.9, 0.85}, ground[3] = {0.08, 0.07, 0.06}, tint[3] = {1.0, 0.9, 0.7};
        for (uint32_t r = 0; r < h; r++)
            for (uint32_t c = 0; c < w; c++) {
                const double u = (c + 0.5) / w * 2 - 1, v = (r + 0.5) / h * 2 - 1;
                double y = 1 - std::fabs(u) - std::fabs(v);
                double x = y >= 0 ? u : (1 - std::fabs(v)) * (u >= 0 ? 1.0 : -1.0);
                double z = y >= 0 ? v : (1 - std::fabs(u)) * (v >= 0 ? 1.0 : -1.0);
                const double n = std::sqrt(x * x + y * y + z * z);
                x /= n; y /= n; z /= n;
                const double up = std::sqrt(std::min(1.0, std::max(0.0, y)));
                const double cs = x * sun[0] + y * sun[1] + z * sun[2];
                const double disc = std::min(1.0, std::max(0.0, (cs - c4) / (c2 - c4)));
                for (int k = 0; k < 3; k++)
                    rgb[3 * ((size_t)r * w + c) + k] = (float)((y >= 0 ? horizon[k] + (zenith[k] - horizon[k]) * up : ground[k]) + disc * sun_scale * tint[k]);
            }
        return from_rgb(table, rgb, w, h);
    }
};

// scene.rs:22-28
class Scene {
  public:
    struct Prim { Sphere geom; int32_t material, light; };
    struct Light { Spectrum spectrum; uint32_t prim; };
    std::vector<Light> lights;
    std::vector<Bsdf> materials;
    std::vector<Prim> primitives;
    EnvMap env_map;  // scene.rs:27 (`_env_map`); width 0 = none, as in the reference

    void add_light(Sphere g, Spectrum light) {  // scene.rs:71-78
        lights.push_back(Light{light, (uint32_t)primitives.size()});
        primitives.push_back(Prim{g, -1, (int32_t)lights.size() - 1});
    }
    void add_material(Sphere g, Bsdf m) {  // scene.rs:80-89
        materials.push_back(m);
        primitives.push_back(Prim{g, (int32_t)materials.size() - 1, -1});
    }
    void add_emissive_material(Sphere g, Bsdf m, Spectrum light) {  // scene.rs:91-110
        materials.push_back(m);
        lights.push_back(Light{light, (uint32_t)primitives.size()});
        primitives.push_back(Prim{g, (int32_t)materials.size() - 1, (int32_t)lights.size() - 1});
    }

    // Scene::dummy, scene.rs:31-69
    static Scene dummy() {
        Scene s;
        s.add_emissive_material(Sphere{{0.0f, 2.3f, 3.0f}, 1.0f}, LambertianBsdf(ConstantSpectrum(0.5f)), ConstantSpectrum(3.0f));
        s.add_material(Sphere{{0.0f, -0.2f, 3.0f}, 1.0f}, LambertianBsdf(ConstantSpectrum(0.5f)));
        s.add_material(Sphere{{0.0f, -101.5f, 2.0f}, 100.0f}, LambertianBsdf(ConstantSpectrum(0.8f)));
        return s;
    }
    // the "light oven" of scene.rs:36-41 / README.md:23
    static Scene furnace(float albedo = 0.5f, float emission = 0.25f) {
        Scene s;
        s.add_emissive_material(Sphere{{0.0f, 0.0f, 0.0f}, 1.0f}, LambertianBsdf(ConstantSpectrum(albedo)), ConstantSpectrum(emission));
        return s;
    }
    // BASELINE config E (the dielectric's values are the comment at scene.rs:55-60)
    static Scene glossy_dielectric() {
        Scene s;
        s.add_emissive_material(Sphere{{0.0f, 2.3f, 3.0f}, 1.0f}, LambertianBsdf(ConstantSpectrum(0.5f)), ConstantSpectrum(3.0f));
        s.add_material(Sphere{{0.0f, -0.2f, 3.0f}, 1.0f}, FresnelBsdf(ConstantSpectrum(1.0f), ConstantSpectrum(1.0f), 1.5220f, 0.00459f));
        s.add_material(Sphere{{0.0f, -101.5f, 2.0f}, 100.0f}, LambertianBsdf(ConstantSpectrum(0.8f)));
        s.add_material(Sphere{{-2.2f, -0.2f, 3.5f}, 1.0f}, MicrofacetBsdf(ConstantSpectrum(0.9f), 0.2f, 0.2f));
        s.add_material(Sphere{{2.2f, -0.2f, 3.5f}, 1.0f}, SpecularBsdf(ConstantSpectrum(0.9f)));
        return s;
    }
    // BASELINE config D: byte-identical to iris_amd.Scene.many_spheres (same PCG32 stream, same f64 arithmetic)
    static Scene many_spheres(uint32_t n_total = 10000, uint64_t seed = 0x1515) {
        Pcg32 rng(seed);
        Scene s;
        s.add_emissive_material(Sphere{{0.0f, 2.3f, 3.0f}, 1.0f}, LambertianBsdf(ConstantSpectrum(0.5f)), ConstantSpectrum(3.0f));
        s.add_material(Sphere{{0.0f, -101.5f, 2.0f}, 100.0f}, LambertianBsdf(ConstantSpectrum(0.8f)));
        while (s.primitives.size() < n_total) {
            double x = -20.0 + 40.0 * rng.uniform();
            double y = -1.3 + 7.3 * rng.uniform();
            double z = 2.0 + 58.0 * rng.uniform();
            double r = 0.05 + 0.30 * rng.uniform();
            double albedo = 0.2 + 0.7 * rng.uniform();
            if (std::sqrt(x * x + (y - 2.3) * (y - 2.3) + (z - 3.0) * (z - 3.0)) < 1.0 + r + 0.05) continue;
            if (std::sqrt(x * x + y * y + z * z) < r + 0.5) continue;
            s.add_material(Sphere{{(float)x, (float)y, (float)z}, (float)r}, LambertianBsdf(ConstantSpectrum((float)albedo)));
        }
        return s;
    }

    // The flat tables the C ABI takes; `desc` points into the vectors of this object.
    struct Flat {
        std::vector<iris_sphere> spheres;
        std::vector<int32_t> prim_material, prim_light;
        std::vector<iris_bsdf> bsdfs;
        std::vector<iris_spectrum> spectra;
        std::vector<iris_light> lights;
        EnvMap env_map;
        iris_envmap_desc env_desc;  // desc() points at it
        iris_scene_desc desc() {
            env_desc = iris_envmap_desc{env_map.width, env_map.height, env_map.texels.data(),
                                        env_map.importance.empty() ? nullptr : env_map.importance.data()};
            iris_scene_desc d;
            d.n_prims = (uint32_t)spheres.size();
            d.spheres = spheres.data();
            d.prim_material = prim_material.data();
            d.prim_light = prim_light.data();
            d.n_bsdfs = (uint32_t)bsdfs.size();
            d.bsdfs = bsdfs.data();
            d.n_spectra = (uint32_t)spectra.size();
            d.spectra = spectra.data();
            d.n_lights = (uint32_t)lights.size();
            d.lights = lights.data();
            d.env_map = env_map.width ? &env_desc : nullptr;  // Scene._env_map (scene.rs:27): the reference never fills it
            return d;
        }
    };
    Flat flatten() const {  // same table order as iris_amd.Scene.flatten: material spectra first, then light spectra
        Flat f;
        auto add_spectrum = [&](const Spectrum &s) { f.spectra.push_back(s.s); return (uint32_t)f.spectra.size() - 1; };
        for (const Bsdf &m : materials) {
            iris_bsdf b = {m.kind, 0u, 0u, m.p0, m.p1};
            if (m.kind != IRIS_BSDF_NULL) b.spec_a = add_spectrum(m.a);
            if (m.kind == IRIS_BSDF_FRESNEL) b.spec_b = add_spectrum(m.b);
            f.bsdfs.push_back(b);
        }
        for (const Light &l : lights) f.lights.push_back(iris_light{add_spectrum(l.spectrum), l.prim});
        for (const Prim &p : primitives) {
            f.spheres.push_back(iris_sphere{p.geom.position[0], p.geom.position[1], p.geom.position[2], p.geom.radius});
            f.prim_material.push_back(p.material);
            f.prim_light.push_back(p.light);
        }
        f.env_map = env_map;
        return f;
    }
};

}  // namespace iris_host
