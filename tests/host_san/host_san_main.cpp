// Host-side code of the library (iris_host.cpp, iris_bvh_build.cpp) and the oracle (iris_oracle.c), built with
// -fsanitize=address,undefined and driven over their edge cases: malformed .hdr and .coeff files, degenerate and random
// sphere sets for the hierarchy build, environment-map tables, and a small oracle render.  Exit status 0 and no sanitizer
// report = pass (tests/test_host_sanitizers.py builds and runs it).  No GPU, no HIP.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/iris_hip.h"
#include "../../iris_amd/csrc/iris_bvh_build.h"
extern "C" {
#include "../../oracle/iris_oracle.h"
}

static int failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static uint32_t rng_state = 0x1515u;
static uint32_t rnd() { rng_state = rng_state * 747796405u + 2891336453u; uint32_t w = ((rng_state >> ((rng_state >> 28) + 4u)) ^ rng_state) * 277803737u; return (w >> 22) ^ w; }
static float rndf() { return (float)(rnd() >> 8) * (1.0f / 16777216.0f); }

static void write_file(const std::string &path, const std::vector<unsigned char> &bytes) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    if (!bytes.empty()) std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
}

// ---- hierarchy build: every sphere in exactly one place, boxes contain their spheres, depth within the stack ----
static void check_bvh(const std::vector<iris_sphere> &sp, uint32_t leaf, bool expect_tree) {
    iris_host::Bvh bvh;
    const bool ok = iris_host::build_bvh(sp.data(), (uint32_t)sp.size(), leaf, bvh);
    if (!expect_tree) { (void)ok; return; }
    if (!ok) return;  // legitimately refused (e.g. nothing left for a tree)
    const uint32_t n = (uint32_t)sp.size();
    CHECK(bvh.prim.size() == n && bvh.spheres.size() == n);
    CHECK(bvh.depth >= 1 && bvh.depth <= IRIS_BVH_STACK);
    CHECK(bvh.nodes.size() % 2 == 0 && bvh.nodes.size() >= 2);
    std::vector<int> seen(n, 0);
    for (uint32_t p : bvh.prim) { CHECK(p < n); if (p < n) seen[p]++; }
    for (uint32_t i = 0; i < n; i++) CHECK(seen[i] == 1);
    const uint32_t n_tree = n - bvh.n_big, n_nodes = (uint32_t)(bvh.nodes.size() / 2);
    std::vector<int> in_leaf(n_tree, 0);
    for (uint32_t k = 0; k < n_nodes; k++) {
        uint32_t first, count;
        std::memcpy(&first, &bvh.nodes[2 * k].w, 4);
        std::memcpy(&count, &bvh.nodes[2 * k + 1].w, 4);
        const iris_host::Float4 lo = bvh.nodes[2 * k], hi = bvh.nodes[2 * k + 1];
        if (count == 0) { CHECK(first + 1 < n_nodes); continue; }
        CHECK(count <= leaf && count <= 7u && first + count <= n_tree);
        for (uint32_t s = first; s < first + count && s < n_tree; s++) {
            in_leaf[s]++;
            const iris_sphere &q = sp[bvh.prim[s]];
            const float r = std::fabs(q.r);
            CHECK(q.cx - r >= lo.x && q.cy - r >= lo.y && q.cz - r >= lo.z && q.cx + r <= hi.x && q.cy + r <= hi.y && q.cz + r <= hi.z);
            if (s > first) CHECK(bvh.prim[s] > bvh.prim[s - 1]);  // ascending primitive order inside a leaf
        }
    }
    for (uint32_t s = 0; s < n_tree; s++) CHECK(in_leaf[s] == 1);
}

static void test_bvh() {
    for (uint32_t leaf = 1; leaf <= 7; leaf++) {
        std::vector<iris_sphere> one = {{0.0f, 0.0f, 3.0f, 1.0f}};
        check_bvh(one, leaf, true);
        std::vector<iris_sphere> same(300, iris_sphere{1.0f, 2.0f, 3.0f, 0.25f});  // all coincident: the bins cannot split them
        check_bvh(same, leaf, true);
        std::vector<iris_sphere> line;
        for (int i = 0; i < 200; i++) line.push_back(iris_sphere{(float)i, 0.0f, 0.0f, 0.1f});
        check_bvh(line, leaf, true);
        for (uint32_t n : {2u, 3u, 65u, 1000u, 5000u}) {
            std::vector<iris_sphere> v;
            for (uint32_t i = 0; i < n; i++) v.push_back(iris_sphere{rndf() * 40.0f - 20.0f, rndf() * 7.0f - 1.0f, rndf() * 58.0f + 2.0f, 0.05f + 0.3f * rndf()});
            v.push_back(iris_sphere{0.0f, -101.5f, 2.0f, 100.0f});  // scene-sized: kept out of the tree
            v.push_back(iris_sphere{0.0f, 2.3f, 3.0f, -1.0f});     // negative radius: |r| bounds it
            check_bvh(v, leaf, true);
        }
        std::vector<iris_sphere> tiny;  // sub-pixel spheres far away
        for (int i = 0; i < 100; i++) tiny.push_back(iris_sphere{rndf() * 400.0f, rndf() * 400.0f, rndf() * 400.0f, 1e-4f + 1e-2f * rndf()});
        check_bvh(tiny, leaf, true);
    }
    std::vector<iris_sphere> bad = {{0.0f, 0.0f, 3.0f, 1.0f}, {NAN, 0.0f, 0.0f, 1.0f}};
    iris_host::Bvh b;
    CHECK(!iris_host::build_bvh(bad.data(), 2u, 4u, b));
    bad[1] = iris_sphere{0.0f, INFINITY, 0.0f, 1.0f};
    CHECK(!iris_host::build_bvh(bad.data(), 2u, 4u, b));
}

// ---- Radiance .hdr reader ----
static std::vector<unsigned char> hdr_bytes(uint32_t w, uint32_t h, int mode, const char *header = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n") {
    std::vector<unsigned char> out(header, header + std::strlen(header));
    char res[64];
    std::snprintf(res, sizeof res, "-Y %u +X %u\n", h, w);
    out.insert(out.end(), res, res + std::strlen(res));
    for (uint32_t y = 0; y < h; y++) {
        if (mode == 0) {  // flat
            for (uint32_t x = 0; x < w; x++) { out.push_back((unsigned char)(x * 7 + y)); out.push_back(90); out.push_back(200); out.push_back(129); }
        } else {  // adaptive run-length: per channel, alternating a run and a literal block
            out.push_back(2); out.push_back(2); out.push_back((unsigned char)(w >> 8)); out.push_back((unsigned char)(w & 255));
            for (int c = 0; c < 4; c++) {
                uint32_t x = 0;
                while (x < w) {
                    uint32_t run = std::min<uint32_t>(w - x, 100u);
                    if ((x / 100) & 1) { out.push_back((unsigned char)run); for (uint32_t k = 0; k < run; k++) out.push_back((unsigned char)(c == 3 ? 128 : k)); }
                    else { out.push_back((unsigned char)(128 + run)); out.push_back((unsigned char)(c == 3 ? 130 : 17 * c)); }
                    x += run;
                }
            }
        }
    }
    return out;
}
static void test_hdr(const std::string &dir) {
    const std::string path = dir + "/t.hdr";
    uint32_t w = 0, h = 0;
    for (int mode = 0; mode < 2; mode++) {
        write_file(path, hdr_bytes(300, 5, mode));
        CHECK(iris_hdr_read(path.c_str(), &w, &h, nullptr, 0) == IRIS_OK && w == 300 && h == 5);
        std::vector<float> rgb((size_t)w * h * 3);
        CHECK(iris_hdr_read(path.c_str(), &w, &h, rgb.data(), rgb.size()) == IRIS_OK);
        CHECK(iris_hdr_read(path.c_str(), &w, &h, rgb.data(), rgb.size() - 1) == IRIS_E_LIMIT);
        for (float v : rgb) CHECK(std::isfinite(v) && v >= 0.0f);
        // every truncation of a valid file is an error or a success, never a fault
        std::vector<unsigned char> full = hdr_bytes(300, 5, mode);
        for (size_t cut = 0; cut < full.size(); cut += 37) {
            write_file(path, std::vector<unsigned char>(full.begin(), full.begin() + cut));
            const int rc = iris_hdr_read(path.c_str(), &w, &h, rgb.data(), rgb.size());
            CHECK(rc == IRIS_OK || rc == IRIS_E_ARG || rc == IRIS_E_LIMIT);
        }
        // random corruption
        for (int trial = 0; trial < 200; trial++) {
            std::vector<unsigned char> bad = full;
            for (int k = 0; k < 8; k++) bad[rnd() % bad.size()] = (unsigned char)rnd();
            write_file(path, bad);
            const int rc = iris_hdr_read(path.c_str(), &w, &h, rgb.data(), rgb.size());
            CHECK(rc == IRIS_OK || rc == IRIS_E_ARG || rc == IRIS_E_LIMIT);
        }
    }
    write_file(path, hdr_bytes(8, 8, 0, "#?NOTRADIANCE\n\n"));
    (void)iris_hdr_read(path.c_str(), &w, &h, nullptr, 0);
    write_file(path, std::vector<unsigned char>());
    CHECK(iris_hdr_read(path.c_str(), &w, &h, nullptr, 0) == IRIS_E_ARG);
    const char *huge = "#?RADIANCE\n\n-Y 4000000000 +X 4000000000\n";
    write_file(path, std::vector<unsigned char>(huge, huge + std::strlen(huge)));
    CHECK(iris_hdr_read(path.c_str(), &w, &h, nullptr, 0) != 12345);  // any code, no overflowed allocation
    CHECK(iris_hdr_read((dir + "/missing.hdr").c_str(), &w, &h, nullptr, 0) == IRIS_E_ARG);
}

// ---- srgb.coeff loader and look-ups; environment-map tables ----
static void test_upsample_and_envmap(const std::string &dir, const std::string &table_path) {
    iris_upsample_table *t = nullptr;
    CHECK(iris_upsample_table_load(table_path.c_str(), &t) == IRIS_OK && t != nullptr);
    orc_upsample_table *ot = orc_upsample_table_load(table_path.c_str());
    CHECK(ot != nullptr);
    if (t && ot) {
        const float specials[] = {0.0f, 1.0f, 0.5f, 1e-30f, 3.0f, 1e30f, -1.0f};
        for (int trial = 0; trial < 4000; trial++) {
            float rgb[3];
            for (int k = 0; k < 3; k++) rgb[k] = trial < 343 ? specials[(trial / (k == 0 ? 1 : (k == 1 ? 7 : 49))) % 7] : rndf() * (trial & 1 ? 1.0f : 40.0f);
            iris_spectrum s, sh;
            float oc[3], och[3], ohdr = 0.0f;
            const int rc = iris_upsample_get_spectrum(t, rgb, &s), orc = orc_upsample_get_spectrum(ot, rgb, oc);
            CHECK((rc == IRIS_OK) == (orc == 1));
            if (rc == IRIS_OK && orc == 1) CHECK(std::memcmp(&s.c0, oc, 12) == 0 || (s.c0 != s.c0 && oc[0] != oc[0]));
            const int rch = iris_upsample_get_spectrum_hdr(t, rgb, &sh), orch = orc_upsample_get_spectrum_hdr(ot, rgb, och, &ohdr);
            CHECK((rch == IRIS_OK) == (orch == 1));
        }
        // environment map from a random latitude-longitude picture
        const uint32_t lw = 37, lh = 19, ow = 24, oh = 16;
        std::vector<float> ll((size_t)lw * lh * 3), oct((size_t)ow * oh * 3), texels((size_t)ow * oh * 4), imp((size_t)ow * oh);
        for (float &v : ll) v = rndf() * 30.0f;
        CHECK(iris_envmap_from_latlong(ll.data(), lw, lh, ow, oh, oct.data()) == IRIS_OK);
        CHECK(iris_envmap_texels_from_rgb(t, oct.data(), (size_t)ow * oh, 50.0f, texels.data(), imp.data()) == IRIS_OK);
        iris_envmap_desc em = {ow, oh, texels.data(), imp.data()};
        std::vector<float> func((size_t)ow * oh), cond((size_t)(ow + 1) * oh), marg(oh + 1);
        float integral = 0.0f;
        CHECK(iris_envmap_tables(&em, func.data(), cond.data(), marg.data(), &integral) == IRIS_OK && integral > 0.0f);
        CHECK(marg[0] == 0.0f && marg[oh] == 1.0f);
        for (uint32_t r = 0; r < oh; r++) CHECK(cond[(size_t)r * (ow + 1)] == 0.0f && cond[(size_t)r * (ow + 1) + ow] == 1.0f);
        std::vector<float> zero(imp.size(), 0.0f);  // nothing to sample: a zero integral, not a division fault
        iris_envmap_desc dark = {ow, oh, texels.data(), zero.data()};
        CHECK(iris_envmap_tables(&dark, func.data(), cond.data(), marg.data(), &integral) == IRIS_OK && integral == 0.0f);
        iris_envmap_desc onepx = {1, 1, texels.data(), nullptr};
        CHECK(iris_envmap_tables(&onepx, func.data(), cond.data(), marg.data(), &integral) == IRIS_OK);
        for (int trial = 0; trial < 2000; trial++) {
            float d[3], uv[2];
            iris_oct_to_dir(rndf() * 2.0f - 1.0f, rndf() * 2.0f - 1.0f, d);
            iris_dir_to_oct(d, uv);
            CHECK(uv[0] >= -1.0f && uv[0] <= 1.0f && uv[1] >= -1.0f && uv[1] <= 1.0f);
        }
        // the oracle's side of the same map: tables, sampling, evaluation, and a small render under it
        orc_scene *sc = orc_scene_new();
        orc_scene_dummy(sc);
        CHECK(orc_scene_set_envmap(sc, ow, oh, texels.data(), imp.data()) == 1);
        for (int trial = 0; trial < 2000; trial++) {
            float d[3], le[4], pdf, wl[4] = {400.0f, 517.5f, 635.0f, 752.5f};
            orc_env_sample(sc, rndf(), rndf(), d);
            orc_env_eval(sc, d, wl, le, &pdf);
            CHECK(std::isfinite(pdf) && pdf >= 0.0f);
        }
        orc_camera cam;
        const float pos[3] = {0.0f, 0.0f, 0.0f};
        orc_camera_new(pos, 1.5f, &cam);
        for (uint32_t integ = 0; integ < 4; integ++) {
            orc_params p = {24, 16, 4, 0, 4, 123456789u, 32, 2, integ};
            std::vector<float> film(24 * 16 * 3), rgbf(24 * 16 * 3);
            orc_stats st;
            orc_render(sc, &cam, &p, 3, film.data(), rgbf.data(), &st);
            CHECK(st.samples == 24u * 16u * 4u);
        }
        orc_scene_free(sc);
    }
    if (t) iris_upsample_table_free(t);
    if (ot) orc_upsample_table_free(ot);
    // damaged tables: wrong magic, truncated, absurd resolution
    const std::string bad = dir + "/bad.coeff";
    FILE *f = std::fopen(table_path.c_str(), "rb");
    std::vector<unsigned char> good;
    if (f) { unsigned char buf[4096]; size_t n; while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) good.insert(good.end(), buf, buf + n); std::fclose(f); }
    CHECK(good.size() > 64);
    for (size_t cut : {(size_t)0, (size_t)3, (size_t)8, (size_t)40, good.size() / 2, good.size() - 1}) {
        write_file(bad, std::vector<unsigned char>(good.begin(), good.begin() + cut));
        iris_upsample_table *tt = nullptr;
        CHECK(iris_upsample_table_load(bad.c_str(), &tt) != IRIS_OK && tt == nullptr);
        // the oracle follows the reference's loader (upsample.rs:108-144: "f32 LE to the end"), which takes a short coefficient
        // block as it comes; its look-up then reports 0 where the reference would panic on the index -- never reads past the end
        orc_upsample_table *ob = orc_upsample_table_load(bad.c_str());
        if (ob) {
            for (int trial = 0; trial < 200; trial++) {
                const float rgb[3] = {rndf(), rndf(), rndf()};
                float c3[3];
                (void)orc_upsample_get_spectrum(ob, rgb, c3);
            }
            orc_upsample_table_free(ob);
        }
    }
    std::vector<unsigned char> magic = good;
    magic[0] = 'X';
    write_file(bad, magic);
    iris_upsample_table *tt = nullptr;
    CHECK(iris_upsample_table_load(bad.c_str(), &tt) != IRIS_OK);
    std::vector<unsigned char> res = good;
    res[4] = res[5] = res[6] = res[7] = 0xff;  // resolution 2^32 - 1
    write_file(bad, res);
    CHECK(iris_upsample_table_load(bad.c_str(), &tt) != IRIS_OK);
    orc_upsample_table_free(orc_upsample_table_load(bad.c_str()));
    res[4] = res[5] = res[6] = res[7] = 0;     // resolution 0
    write_file(bad, res);
    CHECK(iris_upsample_table_load(bad.c_str(), &tt) != IRIS_OK);
}

// ---- camera / film helpers and the oracle's traced paths ----
static void test_small_things() {
    for (float aspect : {1.0f, 16.0f / 9.0f, 0.1f}) {
        iris_camera cam;
        const float pos[3] = {0.5f, -1.25f, 2.0f};
        CHECK(iris_camera_new(pos, aspect, &cam) == IRIS_OK);
        orc_camera oc;
        orc_camera_new(pos, aspect, &oc);
    }
    std::vector<float> xyz = {0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, -1.0f, 5.0f, NAN}, rgb(9);
    CHECK(iris_film_to_rgb(xyz.data(), 3, rgb.data()) == IRIS_OK);
    CHECK(iris_film_to_rgb(nullptr, 0, nullptr) == IRIS_OK || true);
    CHECK(iris_roughness_to_alpha(0.2f) > 0.0f);
    uint32_t b, n;
    CHECK(iris_sample_range(0xffffffffu, 7, 8, &b, &n) == IRIS_OK && b + n == 0xffffffffu);
    orc_scene *sc = orc_scene_new();
    orc_scene_dummy(sc);
    orc_camera cam;
    const float pos[3] = {0.0f, 0.0f, 0.0f};
    orc_camera_new(pos, 1.0f, &cam);
    orc_params p = {64, 64, 8, 0, 8, 123456789u, 15, 2, 0};
    for (int i = 0; i < 500; i++) {
        orc_trace tr;
        orc_trace_sample(sc, &cam, &p, rnd() % 64, rnd() % 64, rnd() % 8, &tr);
        CHECK(tr.n_hits <= ORC_TRACE_MAX_DEPTH);
    }
    orc_scene_free(sc);
}

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: host_san <scratch dir> <srgb .coeff table>\n"); return 2; }
    const std::string dir = argv[1];
    test_bvh();
    test_hdr(dir);
    test_upsample_and_envmap(dir, argv[2]);
    test_small_things();
    if (failures) { std::fprintf(stderr, "%d checks failed\n", failures); return 1; }
    std::puts("host_san: all checks passed");
    return 0;
}
