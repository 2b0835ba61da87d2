// iris_bvh_build.h -- host-side build of the bounding-volume hierarchy over the spheres (the reference
// lists it as a TODO, README.md:26).  Plain C++, no HIP: iris_kernels.hip uploads the arrays it returns, and
// tests/host_san/ compiles it with sanitizers.
#pragma once

#include <stdint.h>

#include <vector>

#include "../../include/iris_hip.h"

#define IRIS_BVH_STACK 24u   // traversal-stack entries per lane; no hierarchy deeper than this is built

namespace iris_host {

struct Float4 { float x, y, z, w; };  // 16 bytes, the layout of HIP's float4

struct Bvh {
    std::vector<Float4> nodes;    // two per node: (lo, first), (hi, count); count > 0: leaf over leaf-ordered spheres [first, first + count)
    std::vector<Float4> spheres;  // (cx, cy, cz, r * r) in leaf order
    std::vector<uint32_t> prim;   // primitive index of each leaf-ordered sphere
    // The nodes as the traversal reads them: one 16-byte record per node except the root, record k - 1 for node k, so that the two
    // children of a node (k, k + 1 with k odd) are one 32-byte-aligned pair: {lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16,
    // first | count << 29}, box coordinates on a 16-bit grid over the root box (grid_origin + q * grid_scale, the same scale on the
    // three axes), rounded outwards and grown by one more unit, so that the quantised box contains the node's box whatever the
    // rounding of the device's grid arithmetic (box_entry_q, iris_device.h).
    std::vector<uint32_t> qnodes;
    float grid_origin[3] = {0.0f, 0.0f, 0.0f}, grid_scale = 1.0f, grid_inv_scale = 1.0f;
    uint32_t depth = 0;
    uint32_t n_big = 0;  // the last n_big leaf-ordered spheres are outside the tree
    uint32_t leaf = 4;   // spheres per leaf, <= 7: a node reference keeps the count in three bits, the first sphere in 29 (option bvh_leaf)
};

// false: no hierarchy (non-finite geometry, nothing left for a tree, or deeper than IRIS_BVH_STACK) -- the list scan is used instead
bool build_bvh(const iris_sphere *spheres, uint32_t n, uint32_t leaf, Bvh &out);

}  // namespace iris_host
