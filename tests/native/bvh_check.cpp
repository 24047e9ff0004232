// bvh_check.cpp — host-side check of the BVH builder (rustracer_amd/csrc/rt_bvh.cpp), compiled by g++ with
// -fsanitize=address,undefined (tests/test_bvh_builder.py). For random scenes of spheres + triangles it verifies the
// invariants the device traversal relies on:
//   * node 0 is the root; every inner reference is a valid node index, every node is referenced exactly once;
//   * every leaf reference is (first << 2 | count - 1) with 1 <= count <= RT_BVH_LEAF inside the ordered primitive array,
//     and the leaves partition that array: every primitive lies in exactly one leaf;
//   * a child's box contains the float geometry of every primitive below it (spheres: c +- |r|, triangles: vertices);
//   * the reported number of stack levels is the depth of the tree and does not exceed RT_BVH_MAX_DEPTH.
// usage: bvh_check <n_spheres> <n_triangles> <seed> <layout: 0 uniform, 1 all coincident, 2 on a line, 3 clustered, 4 geometric progression>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../rustracer_amd/csrc/rt_bvh.h"

static int fail(const char *what, long a = 0, long b = 0) {
    std::printf("FAIL %s (%ld, %ld)\n", what, a, b);
    return 1;
}

struct Checker {
    const rthip_scene_desc &s;
    const std::vector<DevBvhNode> &nodes;
    const std::vector<uint32_t> &prims;
    std::vector<int> node_refs, prim_seen;
    uint32_t max_levels = 0;
    int errors = 0;

    bool inside(const float *lo, const float *hi, const float *plo, const float *phi) const {
        for (int k = 0; k < 3; k++)
            if (!(lo[k] <= plo[k] && phi[k] <= hi[k])) return false;
        return true;
    }
    void prim_box(uint32_t id, float *lo, float *hi) const {
        if (id < s.n_spheres) {
            const rthip_sphere &sp = s.spheres[id];
            const float r = std::fabs(float(sp.radius));
            for (int k = 0; k < 3; k++) lo[k] = float(sp.center[k]) - r, hi[k] = float(sp.center[k]) + r;
        } else {
            const rthip_triangle &t = s.triangles[id - s.n_spheres];
            for (int k = 0; k < 3; k++) {
                const float a = float(t.v0[k]), b = float(t.v1[k]), c = float(t.v2[k]);
                lo[k] = std::fmin(a, std::fmin(b, c)), hi[k] = std::fmax(a, std::fmax(b, c));
            }
        }
    }
    // checks the subtree behind `ref` against the box the parent stores for it
    void visit(uint32_t ref, const float *lo, const float *hi, uint32_t levels) {
        if (ref & RT_BVH_LEAF_BIT) {
            const uint32_t first = (ref & ~RT_BVH_LEAF_BIT) >> 2, count = (ref & 3u) + 1u;
            if (count > RT_BVH_LEAF || size_t(first) + count > prims.size()) { errors += fail("leaf range", first, count); return; }
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t id = prims[first + k];
                if (id >= prim_seen.size()) { errors += fail("primitive id", id); return; }
                prim_seen[first + k]++;
                float plo[3], phi[3];
                prim_box(id, plo, phi);
                if (!inside(lo, hi, plo, phi)) errors += fail("primitive outside its leaf box", id, first);
            }
            return;
        }
        if (ref >= nodes.size()) { errors += fail("inner reference out of range", ref); return; }
        if (++node_refs[ref] > 1) { errors += fail("node referenced twice", ref); return; }
        if (levels + 1 > max_levels) max_levels = levels + 1;
        const DevBvhNode &n = nodes[ref];
        const bool empty1 = n.lo1[0] > n.hi1[0];  // the single-leaf root has an inverted second box
        if (ref != 0 || lo) {
            if (lo && !(inside(lo, hi, n.lo0, n.hi0) && (empty1 || inside(lo, hi, n.lo1, n.hi1)))) errors += fail("child boxes outside the parent's box", ref);
        }
        visit(n.ref0, n.lo0, n.hi0, levels + 1);
        if (!empty1) visit(n.ref1, n.lo1, n.hi1, levels + 1);
    }
};

int main(int argc, char **argv) {
    if (argc < 5) return fail("usage");
    const size_t ns = std::strtoull(argv[1], nullptr, 10), nt = std::strtoull(argv[2], nullptr, 10);
    const unsigned seed = unsigned(std::atoi(argv[3]));
    const int layout = std::atoi(argv[4]);
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    auto point = [&](double out[3]) {
        switch (layout) {
            case 1: out[0] = 0.25, out[1] = -3.0, out[2] = 7.5; break;                               // all centroids coincide
            case 2: out[0] = U(rng) * 1e3, out[1] = 1.0, out[2] = -2.0; break;                      // degenerate extent on two axes
            case 4: { static int i = 0; out[0] = std::pow(1.5, double(i++ % 190)), out[1] = 0.0, out[2] = 0.0; break; }  // SAH peels one primitive per level: the depth guard must cut in
            case 3: { double c = std::floor(U(rng) * 4.0) * 50.0; out[0] = c + U(rng) * 1e-3, out[1] = U(rng) * 1e-3, out[2] = c + U(rng) * 1e-3; break; }
            default: out[0] = U(rng) * 100.0, out[1] = U(rng) * 100.0, out[2] = U(rng) * 100.0;
        }
    };
    std::vector<rthip_sphere> spheres(ns);
    std::vector<rthip_triangle> tris(nt);
    for (auto &sp : spheres) {
        point(sp.center);
        sp.radius = (layout == 1 ? 0.5 : layout == 4 ? 1e-3 : 0.01 + std::fabs(U(rng))) * (rng() % 17 == 0 ? -1.0 : 1.0);  // negative radii are legal (sphere.rs:63-65)
        sp.material = 0, sp._pad = 0;
    }
    for (auto &t : tris) {
        point(t.v0);
        for (int k = 0; k < 3; k++) t.v1[k] = t.v0[k] + U(rng) * (layout == 1 ? 0.0 : 0.3), t.v2[k] = t.v0[k] + U(rng) * (layout == 1 ? 0.0 : 0.3);
        t.material = 0;
    }
    rthip_scene_desc s{};
    s.spheres = spheres.data(), s.n_spheres = ns, s.triangles = tris.data(), s.n_triangles = nt;
    std::vector<DevBvhNode> nodes;
    std::vector<uint32_t> prims;
    const uint32_t levels = rt_build_bvh(&s, nodes, prims);
    if (ns + nt == 0) {
        if (levels != 0 || !nodes.empty() || !prims.empty()) return fail("empty scene");
        std::printf("OK nodes 0 levels 0\n");
        return 0;
    }
    if (prims.size() != ns + nt) return fail("primitive array size", long(prims.size()));
    if (nodes.empty()) return fail("no nodes");
    Checker c{s, nodes, prims, std::vector<int>(nodes.size(), 0), std::vector<int>(prims.size(), 0)};
    c.visit(0, nullptr, nullptr, 0);
    if (c.errors) return 1;
    for (size_t i = 0; i < nodes.size(); i++)
        if (c.node_refs[i] != 1) return fail("node not reachable exactly once", long(i), c.node_refs[i]);
    for (size_t i = 0; i < prims.size(); i++)
        if (c.prim_seen[i] != 1) return fail("primitive slot not in exactly one leaf", long(i), c.prim_seen[i]);
    std::vector<char> ids(prims.size(), 0);
    for (uint32_t id : prims) {
        if (ids[id]) return fail("primitive listed twice", id);
        ids[id] = 1;
    }
    if (c.max_levels != levels) return fail("reported stack levels != tree depth", levels, c.max_levels);
    if (levels > RT_BVH_MAX_DEPTH) return fail("deeper than the traversal stack", levels);
    // the 32-byte quantised nodes the kernels read: same references, every decoded box contains the float box, empty stays empty
    std::vector<DevBvhNodeQ> qn;
    if (!rt_bvh_quantize(nodes, qn) || qn.size() != nodes.size()) return fail("quantisation failed");
    for (size_t i = 0; i < nodes.size(); i++) {
        const DevBvhNode &n = nodes[i];
        if (qn[i].ref0 != n.ref0 || qn[i].ref1 != n.ref1) return fail("quantised references differ", long(i));
        const bool empty1 = n.lo1[0] > n.hi1[0];
        for (int k = 0; k < 3; k++) {
            const uint32_t f = qn[i].frame[k], q = qn[i].q[k];
            if (int(int8_t(f & 0xffu)) < -126) return fail("step exponent not a normal float", long(i), k);
            if (!(rt_bvh_dequantize(f, q & 0xff) <= double(n.lo0[k]) && rt_bvh_dequantize(f, (q >> 8) & 0xff) >= double(n.hi0[k])))
                return fail("quantised box 0 does not contain the float box", long(i), k);
            if (empty1 ? !(((q >> 16) & 0xff) == 255 && (q >> 24) == 0)
                       : !(rt_bvh_dequantize(f, (q >> 16) & 0xff) <= double(n.lo1[k]) && rt_bvh_dequantize(f, q >> 24) >= double(n.hi1[k])))
                return fail("quantised box 1 wrong", long(i), k);
            // not looser than two steps beyond the float box
            const double step = rt_bvh_dequantize(f, 1) - rt_bvh_dequantize(f, 0);
            if (double(n.lo0[k]) - rt_bvh_dequantize(f, q & 0xff) > step || rt_bvh_dequantize(f, (q >> 8) & 0xff) - double(n.hi0[k]) > step)
                return fail("quantised box 0 looser than one step", long(i), k);
        }
    }
    // the device order: top of the tree first, breadth-first (the kernel stages a prefix of the array in LDS): still the same
    // tree — every node reachable once, every leaf intact — a parent before its children inside the prefix, nothing behind the
    // prefix pointing back into it
    {
        std::vector<DevBvhNode> re = nodes;
        const uint32_t n_top = rt_bvh_top_first(re, RT_BVH_TOP_MAX);
        if (n_top != std::min<size_t>(RT_BVH_TOP_MAX, re.size())) return fail("top-of-tree prefix size", long(n_top), long(re.size()));
        if (re.size() != nodes.size()) return fail("reordered node count");
        Checker c2{s, re, prims, std::vector<int>(re.size(), 0), std::vector<int>(prims.size(), 0)};
        c2.visit(0, nullptr, nullptr, 0);
        if (c2.errors) return fail("reordered tree fails the invariants");
        for (size_t i = 0; i < re.size(); i++)
            if (c2.node_refs[i] != 1) return fail("reordered: node not reachable exactly once", long(i), c2.node_refs[i]);
        if (c2.max_levels != levels) return fail("reordered: depth changed", levels, c2.max_levels);
        for (size_t pos = 0; pos < re.size(); pos++)
            for (uint32_t ref : {re[pos].ref0, re[pos].ref1}) {
                if (ref & RT_BVH_LEAF_BIT) continue;
                if (pos < n_top && ref <= pos) return fail("reordered: child before its parent in the prefix", long(pos), long(ref));
                if (pos >= n_top && ref < n_top) return fail("reordered: a node behind the prefix points into it", long(pos), long(ref));
            }
    }
    // FNV-1a over the node and primitive arrays: the threaded build must give the tree of the serial one
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t bytes) {
        const unsigned char *c = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < bytes; i++) hsh = (hsh ^ c[i]) * 1099511628211ull;
    };
    mix(nodes.data(), nodes.size() * sizeof(DevBvhNode));
    mix(prims.data(), prims.size() * sizeof(uint32_t));
    std::printf("OK nodes %zu levels %u hash %016llx\n", nodes.size(), levels, (unsigned long long)hsh);
    return 0;
}
