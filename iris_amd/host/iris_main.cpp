// iris_main.cpp -- the host driver of the reference (src/main.rs) above the C ABI: builds the scene and
// the camera on the host, hands the whole film to the GPU path (one iris_ctx per GPU, sample indices
// split across them), converts XYZ to linear RGB and writes the image.
//
//   iris_render [--width W] [--height H] [--spp N] [--scene dummy|furnace|spheres10k|glossy]
//               [--integrator hwss_naive|hwss_slow|swss_naive|swss_slow] [--max-depth D] [--min-depth D]
//               [--gpus G] [--device I] [--share-device] [--out out.exr|out.pfm] [--dump-scene]
//               [--env none|white|sky|FILE.hdr] [--env-size N] [--table srgb.coeff]
//
// What it mirrors: WIDTH/HEIGHT/TOTAL_SPP and CurrentIntegrator (main.rs:27-31) become flags; the
// thread pool popping tiles (main.rs:87-107) becomes one host thread per GPU, each rendering the
// sample range [g*spp/G, (g+1)*spp/G) of every pixel; the banner and the "m ray/s" line are those of
// main.rs:80-83 and :109-114; the film is written as an RGB float EXR like main.rs:116-123.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "iris_scene.hpp"

namespace {

void put_str(std::vector<unsigned char> &b, const char *s) { b.insert(b.end(), s, s + std::strlen(s) + 1); }
template <typename T>
void put(std::vector<unsigned char> &b, T v) {
    unsigned char raw[sizeof(T)];
    std::memcpy(raw, &v, sizeof(T));  // little-endian host
    b.insert(b.end(), raw, raw + sizeof(T));
}
void put_attr(std::vector<unsigned char> &b, const char *name, const char *type, const std::vector<unsigned char> &value) {
    put_str(b, name);
    put_str(b, type);
    put<int32_t>(b, (int32_t)value.size());
    b.insert(b.end(), value.begin(), value.end());
}

// Minimal OpenEXR 2 writer: single-part scanline image, no compression, three FLOAT channels.
bool write_exr(const char *path, const float *rgb, uint32_t w, uint32_t h) {
    std::vector<unsigned char> hd;
    put<uint32_t>(hd, 20000630u);  // magic
    put<uint32_t>(hd, 2u);         // version 2, no flags
    {
        std::vector<unsigned char> v;
        for (const char *ch : {"B", "G", "R"}) {  // channels are stored in alphabetical order
            put_str(v, ch);
            put<int32_t>(v, 2);  // FLOAT
            put<uint32_t>(v, 0u);  // pLinear + 3 reserved bytes
            put<int32_t>(v, 1);
            put<int32_t>(v, 1);
        }
        v.push_back(0);
        put_attr(hd, "channels", "chlist", v);
    }
    put_attr(hd, "compression", "compression", {0});
    {
        std::vector<unsigned char> v;
        put<int32_t>(v, 0); put<int32_t>(v, 0); put<int32_t>(v, (int32_t)w - 1); put<int32_t>(v, (int32_t)h - 1);
        put_attr(hd, "dataWindow", "box2i", v);
        put_attr(hd, "displayWindow", "box2i", v);
    }
    put_attr(hd, "lineOrder", "lineOrder", {0});
    { std::vector<unsigned char> v; put<float>(v, 1.0f); put_attr(hd, "pixelAspectRatio", "float", v); }
    { std::vector<unsigned char> v; put<float>(v, 0.0f); put<float>(v, 0.0f); put_attr(hd, "screenWindowCenter", "v2f", v); }
    { std::vector<unsigned char> v; put<float>(v, 1.0f); put_attr(hd, "screenWindowWidth", "float", v); }
    hd.push_back(0);
    const uint64_t row_bytes = 8ull + 12ull * w;
    uint64_t off = hd.size() + 8ull * h;
    for (uint32_t y = 0; y < h; y++, off += row_bytes) put<uint64_t>(hd, off);
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    std::vector<float> row(3 * (size_t)w);
    for (uint32_t y = 0; y < h && ok; y++) {
        int32_t head[2] = {(int32_t)y, (int32_t)(12u * w)};
        for (uint32_t x = 0; x < w; x++) {
            const float *p = rgb + 3 * ((size_t)y * w + x);
            row[x] = p[2];
            row[w + x] = p[1];
            row[2 * (size_t)w + x] = p[0];
        }
        ok = std::fwrite(head, 4, 2, f) == 2 && std::fwrite(row.data(), 4, row.size(), f) == row.size();
    }
    return std::fclose(f) == 0 && ok;
}

// Portable float map: "PF", bottom row first, negative scale = little-endian.
bool write_pfm(const char *path, const float *rgb, uint32_t w, uint32_t h) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    bool ok = true;
    for (uint32_t y = h; y-- > 0 && ok;) ok = std::fwrite(rgb + 3 * (size_t)y * w, 4, 3 * (size_t)w, f) == 3 * (size_t)w;
    return std::fclose(f) == 0 && ok;
}

bool ends_with(const std::string &s, const char *suffix) {
    size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int usage(const char *argv0) {
    std::fprintf(stderr,
                 "usage: %s [--width W] [--height H] [--spp N] [--scene dummy|furnace|spheres10k|glossy]\n"
                 "          [--integrator hwss_naive|hwss_slow|swss_naive|swss_slow] [--max-depth D] [--min-depth D]\n"
                 "          [--gpus G] [--device I] [--share-device] [--out FILE.exr|FILE.pfm] [--dump-scene]\n"
                 "          [--env none|white|sky|FILE.hdr] [--env-size N] [--table FILE.coeff]\n", argv0);
    return 2;
}

}  // namespace

int main(int argc, char **argv) {
    uint32_t width = 512, height = 512, spp = 100;  // main.rs:27-29
    uint32_t max_depth = 15, min_depth = 2;         // hwss_naive.rs:16-17
    uint32_t integrator = IRIS_INT_HWSS_NAIVE;      // main.rs:31
    int gpus = 1, device = 0;
    bool dump_scene = false, share_device = false;
    std::string scene_name = "dummy", out = "out.exr", env_name = "none", table_path = "data/srgb.coeff" /* upsample.rs:109 */;
    uint32_t env_size = 256;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--width") width = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--height") height = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--spp") spp = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--max-depth") max_depth = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--min-depth") min_depth = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--gpus") gpus = std::atoi(val());
        else if (a == "--device") device = std::atoi(val());
        else if (a == "--scene") scene_name = val();
        else if (a == "--out") out = val();
        else if (a == "--env") env_name = val();
        else if (a == "--env-size") env_size = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (a == "--table") table_path = val();
        else if (a == "--dump-scene") dump_scene = true;
        else if (a == "--share-device") share_device = true;  // all contexts on --device (rehearses --gpus on a 1-GPU box)
        else if (a == "--integrator") {
            std::string v = val();
            if (v == "hwss_naive") integrator = IRIS_INT_HWSS_NAIVE;
            else if (v == "hwss_slow") integrator = IRIS_INT_HWSS_SLOW;
            else if (v == "swss_naive") integrator = IRIS_INT_SWSS_NAIVE;
            else if (v == "swss_slow") integrator = IRIS_INT_SWSS_SLOW;
            else return usage(argv[0]);
        } else return usage(argv[0]);
    }
    if (width == 0 || height == 0 || spp == 0 || gpus < 1) return usage(argv[0]);

    iris_host::Scene scene;
    if (scene_name == "dummy") scene = iris_host::Scene::dummy();
    else if (scene_name == "furnace") scene = iris_host::Scene::furnace();
    else if (scene_name == "spheres10k") scene = iris_host::Scene::many_spheres();
    else if (scene_name == "glossy") scene = iris_host::Scene::glossy_dielectric();
    else return usage(argv[0]);
    // Scene._env_map (scene.rs:27): none in the reference; here a uniform white map or the procedural HDR sky
    if (env_name == "white") scene.env_map = iris_host::EnvMap::constant(1.0f);
    else if (env_name == "sky") {
        iris_upsample_table *table = nullptr;
        if (env_size == 0 || iris_upsample_table_load(table_path.c_str(), &table) != IRIS_OK) {
            std::fprintf(stderr, "--env sky needs an RGB->spectrum table: cannot load %s (tools/rgb2spec_opt.c makes one)\n", table_path.c_str());
            return 1;
        }
        scene.env_map = iris_host::EnvMap::sky(table, env_size, env_size);
        iris_upsample_table_free(table);
    } else if (ends_with(env_name, ".hdr")) {  // an equirectangular Radiance picture (data/*.hdr, .gitignore:7)
        uint32_t w = 0, h = 0;
        if (iris_hdr_read(env_name.c_str(), &w, &h, nullptr, 0) != IRIS_OK) {
            std::fprintf(stderr, "cannot read %s as an RGBE picture\n", env_name.c_str());
            return 1;
        }
        std::vector<float> latlong((size_t)w * h * 3), oct((size_t)env_size * env_size * 3);
        iris_upsample_table *table = nullptr;
        if (env_size == 0 || iris_hdr_read(env_name.c_str(), &w, &h, latlong.data(), latlong.size()) != IRIS_OK ||
            iris_envmap_from_latlong(latlong.data(), w, h, env_size, env_size, oct.data()) != IRIS_OK) {
            std::fprintf(stderr, "%s is damaged\n", env_name.c_str());
            return 1;
        }
        if (iris_upsample_table_load(table_path.c_str(), &table) != IRIS_OK) {
            std::fprintf(stderr, "--env FILE.hdr needs an RGB->spectrum table: cannot load %s (tools/rgb2spec_opt.c makes one)\n", table_path.c_str());
            return 1;
        }
        scene.env_map = iris_host::EnvMap::from_rgb(table, oct, env_size, env_size);
        iris_upsample_table_free(table);
    } else if (env_name != "none") return usage(argv[0]);
    iris_host::Scene::Flat flat = scene.flatten();

    if (dump_scene) {  // the flat tables, one record per line (compared with the Python mirror in tests/)
        for (size_t i = 0; i < flat.spheres.size(); i++)
            std::printf("prim %zu %a %a %a %a %d %d\n", i, flat.spheres[i].cx, flat.spheres[i].cy, flat.spheres[i].cz, flat.spheres[i].r,
                        flat.prim_material[i], flat.prim_light[i]);
        for (size_t i = 0; i < flat.bsdfs.size(); i++)
            std::printf("bsdf %zu %u %u %u %a %a\n", i, flat.bsdfs[i].kind, flat.bsdfs[i].spec_a, flat.bsdfs[i].spec_b, flat.bsdfs[i].p0, flat.bsdfs[i].p1);
        for (size_t i = 0; i < flat.spectra.size(); i++)
            std::printf("spectrum %zu %u %a %a %a %a\n", i, flat.spectra[i].kind, flat.spectra[i].c0, flat.spectra[i].c1, flat.spectra[i].c2, flat.spectra[i].hdr);
        for (size_t i = 0; i < flat.lights.size(); i++) std::printf("light %zu %u %u\n", i, flat.lights[i].spectrum, flat.lights[i].prim);
        return 0;
    }

    // main.rs:51-54
    const float position[3] = {0.0f, 0.0f, 0.0f};
    iris_camera camera;
    if (iris_camera_new(position, (float)width / (float)height, &camera) != IRIS_OK) {
        std::fprintf(stderr, "iris_camera_new failed\n");
        return 1;
    }

    // one context per GPU, scene uploaded before the clock starts (the reference builds its scene before `start` too)
    std::vector<iris_ctx *> ctx((size_t)gpus, nullptr);
    const iris_scene_desc desc = flat.desc();
    for (int g = 0; g < gpus; g++) {
        int rc = iris_ctx_create(share_device ? device : device + g, &ctx[g]);
        if (rc != IRIS_OK) {
            std::fprintf(stderr, "iris_ctx_create(device %d) failed (%d): no HIP device? This driver has no CPU fallback.\n", device + g, rc);
            return 1;
        }
        rc = iris_scene_upload(ctx[g], &desc);
        if (rc != IRIS_OK) {
            std::fprintf(stderr, "iris_scene_upload failed (%d): %s\n", rc, iris_last_error(ctx[g]));
            return 1;
        }
    }

    std::printf("Starting render, %ux%u@%uspp...\n", width, height, spp);  // main.rs:80-83
    const auto start = std::chrono::steady_clock::now();
    const size_t n = (size_t)width * height * 3;
    std::vector<std::vector<float>> films(1, std::vector<float>(n));
    {
        // one call: a host thread per GPU, each rendering its share of the sample indices, then the RCCL
        // all-reduce of the device films over xGMI (host sum when the contexts share a device)
        const iris_render_params p = {width, height, spp, 0u, spp, 123456789u /* tile.rs:15 */, max_depth, min_depth, integrator};
        const int rc = iris_render_multi(ctx.data(), gpus, &camera, &p, films[0].data());
        if (rc != IRIS_OK) {
            std::fprintf(stderr, "iris_render_multi failed (%d): %s\n", rc, iris_last_error(ctx[0]));
            return 1;
        }
    }
    const float elapsed = std::chrono::duration<float>(std::chrono::steady_clock::now() - start).count();
    std::printf("Done in %gs (%gm ray/s)\n", elapsed, (float)((double)spp * width * height) / (1000000.0f * elapsed));  // main.rs:109-114
    uint64_t rays = 0, bounces = 0;
    for (int g = 0; g < gpus; g++) {
        iris_stats st;
        iris_get_stats(ctx[g], &st);
        rays += st.rays;
        bounces += st.bounces;
    }
    std::printf("%.1f M scene scans/s, %.3f bounces/sample, %d GPU(s)\n", (double)rays / 1e6 / elapsed,
                (double)bounces / ((double)spp * width * height), gpus);

    std::vector<float> rgb(n);
    iris_film_to_rgb(films[0].data(), (size_t)width * height, rgb.data());  // tile.rs:108-109
    const bool ok = ends_with(out, ".pfm") ? write_pfm(out.c_str(), rgb.data(), width, height) : write_exr(out.c_str(), rgb.data(), width, height);
    if (!ok) {
        std::fprintf(stderr, "cannot write %s\n", out.c_str());
        return 1;
    }
    for (iris_ctx *c : ctx) iris_ctx_destroy(c);
    return 0;
}
