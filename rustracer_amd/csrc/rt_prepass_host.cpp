// rt_prepass_host.cpp — host mirror of prepass_kernel (rt_kernels.hip), built from the same geometry (rt_prepass.h).
// It is the checker of the primary-visibility lists (tests/test_prepass.py: every leaf a primary ray's traversal would open
// is in its pixel's mask; the device's lists equal these) and the tool that sized RT_PRE_LEAVES (tools/prepass_stats.py).
// Not on the render path: nothing in rthip_api.cpp calls it.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/rthip.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_prepass.h"

namespace {

void camera_params(const rthip_camera &c, uint32_t width, uint32_t height, KernelParams &k) {
    std::memset(&k, 0, sizeof k);
    k.ox = float(c.origin[0]), k.oy = float(c.origin[1]), k.oz = float(c.origin[2]);
    k.d00x = float(c.d00[0]), k.d00y = float(c.d00[1]), k.d00z = float(c.d00[2]);
    k.hx = float(c.horizontal[0]), k.hy = float(c.horizontal[1]), k.hz = float(c.horizontal[2]);
    k.vx = float(c.vertical[0]), k.vy = float(c.vertical[1]), k.vz = float(c.vertical[2]);
    k.width = width, k.height = height;
    k.inv_wm1 = float(1.0 / (double(width) - 1.0));
    k.inv_hm1 = float(1.0 / (double(height) - 1.0));
    k.hwx = k.hx * k.inv_wm1, k.hwy = k.hy * k.inv_wm1, k.hwz = k.hz * k.inv_wm1;
    k.vhx = k.vx * k.inv_hm1, k.vhy = k.vy * k.inv_hm1, k.vhz = k.vz * k.inv_hm1;
}

struct LeafBox {
    uint32_t ref;
    float near;
    float lo[3], hi[3];
};

}  // namespace

// The tree the device would get for `scene` (same builder, same node order, same quantisation), then for every 8x8 wave tile
// (tx, ty) of the window [tx0, tx0 + ntx) x [ty0, ty0 + nty) of a width x height image (whole-image launch, no bands):
//   out[t]        the tile's record as prepass_kernel writes it (RtTilePre: the RT_PRE_LEAVES nearest leaves in order of
//                 (near, reference), per-pixel masks and cut distances; ent[0].ref = RT_PRE_OVERFLOW when the tile touches
//                 more than RT_PRE_COLLECT leaves)
//   leaf_count[t] leaves the tile's pyramid touches (not capped)
//   pixel_count[t*64 + p] leaves pixel p's pyramid touches (not capped)
// (The device walks breadth-first with a bounded frontier: a tile whose walk overflows THAT is marked RT_PRE_OVERFLOW there
// and not here.) Returns 0, or -1 on bad arguments / a scene without a tree.
extern "C" int rt_prepass_host(const rthip_scene_desc *scene, const rthip_camera *camera, uint32_t width, uint32_t height, uint32_t tx0, uint32_t ty0,
                               uint32_t ntx, uint32_t nty, RtTilePre *out, uint32_t *leaf_count, uint32_t *pixel_count) {
    if (!scene || !camera || width < 2 || height < 2) return -1;
    try {
        std::vector<DevBvhNode> nodes;
        std::vector<uint32_t> prims;
        if (rt_build_bvh(scene, nodes, prims) == 0) return -1;
        rt_bvh_top_first(nodes, RT_BVH_TOP_MAX);
        std::vector<DevBvhNodeQ> q;
        if (!rt_bvh_quantize(nodes, q)) return -1;
        KernelParams P;
        camera_params(*camera, width, height, P);
        const float m = RT_PRE_MARGIN;
        std::vector<LeafBox> leaves;
        std::vector<uint32_t> stack;
        for (uint32_t jy = 0; jy < nty; jy++)
            for (uint32_t jx = 0; jx < ntx; jx++) {
                const size_t t = size_t(jy) * ntx + jx;
                const uint32_t x0 = (tx0 + jx) * 8u, y0 = (ty0 + jy) * 8u;
                const uint32_t x1 = std::min(x0 + 8u, width), y1 = std::min(y0 + 8u, height);
                leaves.clear();
                if (x0 < width && y0 < height) {
                    RtPyramid tile;
                    rt_pre_pyramid(P, float(x0) - m, float(x1) + m, float(y0) - m, float(y1) + m, tile);
                    stack.assign(1, 0u);
                    while (!stack.empty()) {
                        const uint32_t ref = stack.back();
                        stack.pop_back();
                        float lo[2][3], hi[2][3];
                        bool empty[2];
                        rt_pre_decode(q[ref], lo, hi, empty);
                        const uint32_t child[2] = {q[ref].ref0, q[ref].ref1};
                        for (int c = 1; c >= 0; c--) {
                            if (empty[c] || rt_pre_box_outside(tile, lo[c], hi[c])) continue;
                            if (child[c] & RT_BVH_LEAF_BIT) {
                                LeafBox lb;
                                lb.ref = child[c];
                                for (int k = 0; k < 3; k++) lb.lo[k] = lo[c][k], lb.hi[k] = hi[c][k];
                                lb.near = rt_pre_box_near(tile.o, lb.lo, lb.hi);
                                leaves.push_back(lb);
                            } else {
                                stack.push_back(child[c]);
                            }
                        }
                    }
                }
                if (leaf_count) leaf_count[t] = uint32_t(leaves.size());
                const bool overflow = leaves.size() > RT_PRE_COLLECT;
                std::sort(leaves.begin(), leaves.end(), [](const LeafBox &a, const LeafBox &b) { return a.near < b.near || (a.near == b.near && a.ref < b.ref); });
                if (out) {
                    for (uint32_t i = 0; i < RT_PRE_LEAVES; i++) {
                        const bool used = !overflow && i < leaves.size();
                        out[t].ent[i].ref = used ? leaves[i].ref : RT_BVH_DONE;
                        out[t].ent[i].near = used ? leaves[i].near : INFINITY;
                    }
                    if (overflow) out[t].ent[0].ref = RT_PRE_OVERFLOW;
                }
                for (uint32_t p = 0; p < 64; p++) {
                    const uint32_t x = x0 + (p & 7u), y = y0 + (p >> 3);
                    unsigned long long mask = 0ull;
                    float cut = INFINITY;
                    uint32_t n = 0;
                    if (x < width && y < height) {
                        RtPyramid px;
                        rt_pre_pyramid(P, float(x) - m, float(x) + 1.0f + m, float(y) - m, float(y) + 1.0f + m, px);
                        for (size_t i = 0; i < leaves.size(); i++)
                            if (!rt_pre_box_outside(px, leaves[i].lo, leaves[i].hi)) {
                                n++;
                                if (i < RT_PRE_LEAVES) mask |= 1ull << i;
                                else cut = std::fmin(cut, leaves[i].near);
                            }
                    }
                    if (out) out[t].mask[p] = overflow ? 0ull : mask, out[t].cut[p] = overflow ? INFINITY : cut;
                    if (pixel_count) pixel_count[t * 64 + p] = n;
                }
            }
        return 0;
    } catch (...) {
        return -1;
    }
}

// The BVH-ordered primitive array of the tree above: slot j of a leaf range [first, first + count) holds scene primitive
// prims[j] (a sphere index, or n_spheres + triangle index). Returns the number of primitives (written: min(that, cap)), -1 on error.
extern "C" long long rt_prepass_host_prims(const rthip_scene_desc *scene, uint32_t *prims_out, unsigned long long cap) {
    if (!scene) return -1;
    try {
        std::vector<DevBvhNode> nodes;
        std::vector<uint32_t> prims;
        if (rt_build_bvh(scene, nodes, prims) == 0) return -1;
        for (size_t i = 0; i < prims.size() && i < cap; i++) prims_out[i] = prims[i];
        return (long long)prims.size();
    } catch (...) {
        return -1;
    }
}
