// iris_bvh_build.cpp -- see iris_bvh_build.h
#include "iris_bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace iris_host {
namespace {

// ---- host-side hierarchy build: balanced median splits of the sphere centres along the widest axis ----
Float4 with_bits(float x, float y, float z, uint32_t w) {
    float f;
    std::memcpy(&f, &w, 4);
    return Float4{x, y, z, f};
}
struct BvhBounds {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void add(const float c[3], float r) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], c[k] - r); hi[k] = std::max(hi[k], c[k] + r); }
    }
    float half_area() const {
        const float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return d[0] < 0.0f ? 0.0f : d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};
float centre_of(const iris_sphere &s, int axis) { return axis == 0 ? s.cx : (axis == 1 ? s.cy : s.cz); }

// One node over order[begin, end): binned surface-area-heuristic split (median split when `sah` is off or the
// bins cannot separate the centres).
void build_bvh_node(const iris_sphere *sp, std::vector<uint32_t> &order, uint32_t begin, uint32_t end, uint32_t node, uint32_t depth, bool sah, Bvh &out) {
    BvhBounds box, cen;
    for (uint32_t i = begin; i < end; i++) {
        const iris_sphere &s = sp[order[i]];
        const float c[3] = {s.cx, s.cy, s.cz};
        box.add(c, std::fabs(s.r));
        cen.add(c, 0.0f);
    }
    // Inflate by 16 ulps of the largest coordinate: covers the rounding of c +- r here and of the slab arithmetic on the device.
    // What the f32 sphere test needs beyond the exact box is distance-dependent and added per ray and box (box_entry, iris_device.h).
    float mag = 0.0f;
    for (int k = 0; k < 3; k++) mag = std::max(mag, std::max(std::fabs(box.lo[k]), std::fabs(box.hi[k])));
    const float pad = 1e-6f * mag + 1e-30f;
    const float lo[3] = {box.lo[0] - pad, box.lo[1] - pad, box.lo[2] - pad}, hi[3] = {box.hi[0] + pad, box.hi[1] + pad, box.hi[2] + pad};
    out.depth = std::max(out.depth, depth);
    if (end - begin <= out.leaf) {
        std::sort(order.begin() + begin, order.begin() + end);  // leaf: spheres in ascending primitive order
        out.nodes[2 * node] = with_bits(lo[0], lo[1], lo[2], begin);
        out.nodes[2 * node + 1] = with_bits(hi[0], hi[1], hi[2], end - begin);
        return;
    }
    int axis = 0;
    for (int k = 1; k < 3; k++)
        if (cen.hi[k] - cen.lo[k] > cen.hi[axis] - cen.lo[axis]) axis = k;
    uint32_t mid = begin + (end - begin) / 2;
    bool split_done = false;
    if (sah) {
        constexpr int BINS = 16;
        float best_cost = INFINITY, best_pos = 0.0f;
        int best_axis = -1;
        for (int ax = 0; ax < 3; ax++) {
            const float c0 = cen.lo[ax], ext = cen.hi[ax] - cen.lo[ax];
            if (!(ext > 0.0f)) continue;
            BvhBounds bin_box[BINS];
            uint32_t bin_n[BINS] = {};
            for (uint32_t i = begin; i < end; i++) {
                const iris_sphere &s = sp[order[i]];
                const int bi = std::min(BINS - 1, (int)((centre_of(s, ax) - c0) / ext * BINS));
                const float c[3] = {s.cx, s.cy, s.cz};
                bin_box[bi].add(c, std::fabs(s.r));
                bin_n[bi]++;
            }
            float right_area[BINS];
            BvhBounds acc;
            for (int k = BINS - 1; k > 0; k--) {
                for (int d = 0; d < 3; d++) { acc.lo[d] = std::min(acc.lo[d], bin_box[k].lo[d]); acc.hi[d] = std::max(acc.hi[d], bin_box[k].hi[d]); }
                right_area[k] = acc.half_area();
            }
            BvhBounds left;
            uint32_t n_left = 0;
            for (int k = 0; k < BINS - 1; k++) {
                for (int d = 0; d < 3; d++) { left.lo[d] = std::min(left.lo[d], bin_box[k].lo[d]); left.hi[d] = std::max(left.hi[d], bin_box[k].hi[d]); }
                n_left += bin_n[k];
                const uint32_t n_right = (end - begin) - n_left;
                if (n_left == 0 || n_right == 0) continue;
                const float cost = left.half_area() * (float)n_left + right_area[k + 1] * (float)n_right;
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_pos = c0 + ext * (float)(k + 1) / BINS; }
            }
        }
        if (best_axis >= 0) {
            auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t a) {
                const float c0 = cen.lo[best_axis], ext = cen.hi[best_axis] - cen.lo[best_axis];
                return std::min(15, (int)((centre_of(sp[a], best_axis) - c0) / ext * 16)) < (int)std::lround((best_pos - c0) / ext * 16);
            });
            mid = (uint32_t)(it - order.begin());
            split_done = mid > begin && mid < end;
        }
    }
    if (!split_done) {
        mid = begin + (end - begin) / 2;
        std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end, [&](uint32_t a, uint32_t b) {
            const float ca = centre_of(sp[a], axis), cb = centre_of(sp[b], axis);
            return ca < cb || (ca == cb && a < b);
        });
    }
    const uint32_t left = (uint32_t)(out.nodes.size() / 2);
    out.nodes.resize(out.nodes.size() + 4);
    out.nodes[2 * node] = with_bits(lo[0], lo[1], lo[2], left);
    out.nodes[2 * node + 1] = with_bits(hi[0], hi[1], hi[2], 0u);
    build_bvh_node(sp, order, begin, mid, left, depth + 1, sah, out);
    build_bvh_node(sp, order, mid, end, left + 1, depth + 1, sah, out);
}
// Spheres that span a large part of the scene (a ground sphere, a big light) would put a scene-sized box at every
// level of their branch; they are kept out of the tree and tested first, one by one (at most BVH_MAX_BIG).
constexpr uint32_t BVH_MAX_BIG = 16;
}  // namespace

bool build_bvh(const iris_sphere *sp, uint32_t n, uint32_t leaf, Bvh &out) {
    out.leaf = leaf;
    BvhBounds cen;
    for (uint32_t i = 0; i < n; i++) {
        if (!std::isfinite(sp[i].cx) || !std::isfinite(sp[i].cy) || !std::isfinite(sp[i].cz) || !std::isfinite(sp[i].r)) return false;
        const float c[3] = {sp[i].cx, sp[i].cy, sp[i].cz};
        cen.add(c, 0.0f);
    }
    // "big": radius above a quarter of the median-ish scale of the centres' bounding box
    const float extent = std::max(cen.hi[0] - cen.lo[0], std::max(cen.hi[1] - cen.lo[1], cen.hi[2] - cen.lo[2]));
    std::vector<uint32_t> by_radius(n);
    for (uint32_t i = 0; i < n; i++) by_radius[i] = i;
    const uint32_t n_cand = std::min(n, BVH_MAX_BIG);
    std::partial_sort(by_radius.begin(), by_radius.begin() + n_cand, by_radius.end(),
                      [&](uint32_t a, uint32_t b) { return std::fabs(sp[a].r) > std::fabs(sp[b].r) || (std::fabs(sp[a].r) == std::fabs(sp[b].r) && a < b); });
    std::vector<uint32_t> big;
    std::vector<char> is_big(n, 0);
    for (uint32_t k = 0; k < n_cand; k++)
        if (std::fabs(sp[by_radius[k]].r) > 0.125f * extent) { big.push_back(by_radius[k]); is_big[by_radius[k]] = 1; }
    std::sort(big.begin(), big.end());
    std::vector<uint32_t> order;
    for (uint32_t i = 0; i < n; i++)
        if (!is_big[i]) order.push_back(i);
    if (order.empty() || n >= (1u << 29)) return false;
    for (int attempt = 0; attempt < 2; attempt++) {  // surface-area heuristic first; balanced medians if that came out too deep
        out.nodes.assign(2, Float4{0.0f, 0.0f, 0.0f, 0.0f});
        out.depth = 0;
        std::vector<uint32_t> work = order;
        build_bvh_node(sp, work, 0, (uint32_t)work.size(), 0, 1, attempt == 0, out);
        if (out.depth <= IRIS_BVH_STACK) { order = work; break; }
        if (attempt == 1) return false;
    }
    // Nodes in breadth-first order: the top of the tree, which every ray walks, becomes a prefix of the arrays (k_wfq_trace keeps
    // that prefix in LDS).  Children stay adjacent pairs (k, k + 1) with k odd.
    {
        const size_t n_nodes = out.nodes.size() / 2;
        std::vector<Float4> bfs(out.nodes.size());
        std::vector<uint32_t> queue_old(1, 0u);  // old index of the node that gets new index i
        queue_old.reserve(n_nodes);
        for (size_t i = 0; i < queue_old.size(); i++) {
            Float4 lo = out.nodes[2 * queue_old[i]], hi = out.nodes[2 * queue_old[i] + 1];
            uint32_t first, count;
            std::memcpy(&first, &lo.w, 4);
            std::memcpy(&count, &hi.w, 4);
            if (count == 0u) {  // inner node: its children take the next two indices
                const uint32_t left_new = (uint32_t)queue_old.size();
                queue_old.push_back(first);
                queue_old.push_back(first + 1u);
                std::memcpy(&lo.w, &left_new, 4);
            }
            bfs[2 * i] = lo;
            bfs[2 * i + 1] = hi;
        }
        if (queue_old.size() != n_nodes) return false;
        out.nodes.swap(bfs);
    }
    // the 16-bit grid records (iris_bvh_build.h): grid over the root box, four units of margin below, the scale a little over
    // extent / 65528 so that every upper bound stays below 65535 after its outward rounding and the extra unit
    {
        const Float4 rlo = out.nodes[0], rhi = out.nodes[1];
        const double ext = std::max({(double)rhi.x - rlo.x, (double)rhi.y - rlo.y, (double)rhi.z - rlo.z, 1e-30});
        const float s = (float)(ext * (1.0 + 1e-6) / 65528.0);
        if (!(s > 0.0f) || !std::isfinite(s)) return false;
        out.grid_scale = s;
        out.grid_inv_scale = (float)(1.0 / (double)s);
        const float root_lo[3] = {rlo.x, rlo.y, rlo.z};
        for (int k = 0; k < 3; k++) out.grid_origin[k] = std::nextafter((float)((double)root_lo[k] - 4.0 * (double)s), -INFINITY);
        const size_t n_nodes = out.nodes.size() / 2;
        out.qnodes.assign(4 * (n_nodes - 1), 0u);
        for (size_t k = 1; k < n_nodes; k++) {
            const Float4 lo = out.nodes[2 * k], hi = out.nodes[2 * k + 1];
            const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
            uint32_t ql[3], qh[3];
            for (int a = 0; a < 3; a++) {
                const double g = (double)out.grid_origin[a];
                const double fl = std::floor(((double)l[a] - g) / (double)s) - 1.0, ch = std::ceil(((double)h[a] - g) / (double)s) + 1.0;
                if (!(fl >= 0.0) || !(ch <= 65535.0)) return false;  // cannot happen by the choice of origin and scale
                ql[a] = (uint32_t)fl;
                qh[a] = (uint32_t)ch;
            }
            uint32_t first, count;
            std::memcpy(&first, &lo.w, 4);
            std::memcpy(&count, &hi.w, 4);
            uint32_t *q = &out.qnodes[4 * (k - 1)];
            q[0] = ql[0] | (ql[1] << 16);
            q[1] = ql[2] | (qh[0] << 16);
            q[2] = qh[1] | (qh[2] << 16);
            q[3] = first | (count << 29);
        }
    }
    // leaf-ordered arrays: the tree's spheres first (node `first` fields index them), the big ones behind them
    out.n_big = (uint32_t)big.size();
    out.prim = order;
    out.prim.insert(out.prim.end(), big.begin(), big.end());
    out.spheres.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const iris_sphere &s = sp[out.prim[i]];
        out.spheres[i] = Float4{s.cx, s.cy, s.cz, s.r * s.r};
    }
    return true;
}


}  // namespace iris_host
