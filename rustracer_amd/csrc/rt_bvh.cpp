// rt_bvh.cpp — host-side builder of the linear BVH used for big scenes (extension: the reference
// scans every shape linearly, raytracer_core/src/shapes/ray.rs:47-67; the BVH must return exactly
// what that scan returns, including the "later index wins on equal t" rule, see rt_kernels.hip).
//
// Binned-SAH top-down build over the FLOAT primitives the device will test (so the boxes bound what
// the kernel intersects), leaves of <= RT_BVH_LEAF primitives. Emitted as 64-byte nodes that hold
// the boxes of BOTH children plus a reference to each (DevBvhNode: one fetch decides both
// descents), leaves being references into the BVH-ordered primitive array `prims`.
// Depth is capped at RT_BVH_MAX_DEPTH (the traversal stack lives in LDS) by switching to median
// splits when a subtree could otherwise exceed it.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <system_error>
#include <numeric>

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) lo[k] = INFINITY, hi[k] = -INFINITY;
    }
    void grow(const Box &b) {
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], b.lo[k]), hi[k] = std::max(hi[k], b.hi[k]);
    }
    void grow_pt(const float *p) {
        for (int k = 0; k < 3; k++) lo[k] = std::min(lo[k], p[k]), hi[k] = std::max(hi[k], p[k]);
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    std::vector<Box> pbox;       // per primitive
    std::vector<float> centroid; // 3 per primitive
    std::vector<uint32_t> &prims;
    explicit Builder(std::vector<uint32_t> &p) : prims(p) {}

    static constexpr int BINS = 16;
    // subtrees of at least PAR_MIN primitives in the first PAR_DEPTH levels are built by their own thread (they work on
    // disjoint ranges of `prims`); the node order — parent, left subtree, right subtree — and hence the tree is the same
    // as that of the serial build
    static constexpr uint32_t PAR_MIN = 1u << 15;
#ifdef RT_BVH_SERIAL
    static constexpr int PAR_DEPTH = 0;
#else
    static constexpr int PAR_DEPTH = 5;
#endif

    static void append_subtree(std::vector<DevBvhNode> &nodes, const std::vector<DevBvhNode> &sub, uint32_t &ref) {
        const uint32_t offset = uint32_t(nodes.size());
        for (DevBvhNode nd : sub) {
            if (!(nd.ref0 & RT_BVH_LEAF_BIT)) nd.ref0 += offset;
            if (!(nd.ref1 & RT_BVH_LEAF_BIT)) nd.ref1 += offset;
            nodes.push_back(nd);
        }
        if (!(ref & RT_BVH_LEAF_BIT)) ref += offset;
    }

    // Appends the subtree over prims[begin, end) to `nodes` (node indices are positions in `nodes`) and returns its reference
    // and box. `levels`: inner nodes on the deepest root-to-leaf path = stack levels the traversal needs.
    uint32_t build(uint32_t begin, uint32_t end, int depth, Box &b, std::vector<DevBvhNode> &nodes, uint32_t &levels) {
        Box cb;
        b.reset(), cb.reset();
        for (uint32_t i = begin; i < end; i++) {
            b.grow(pbox[prims[i]]);
            cb.grow_pt(&centroid[3 * size_t(prims[i])]);
        }
        const uint32_t n = end - begin;
        if (n <= RT_BVH_LEAF) return RT_BVH_LEAF_BIT | (begin << 2) | (n - 1u);
        const uint32_t me = uint32_t(nodes.size());
        nodes.emplace_back();
        levels = std::max(levels, uint32_t(depth) + 1u);
        // pick the split
        uint32_t mid = begin + n / 2;
        bool use_median = false;
        // depth guard: a subtree of n prims needs ceil(log2(n / leaf)) more levels if split evenly
        const int levels_left = RT_BVH_MAX_DEPTH - depth;
        if (double(n) > std::ldexp(double(RT_BVH_LEAF), std::max(0, levels_left - 1)) * 0.5) use_median = true;
        int axis = 0;
        float ext[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        if (!(ext[axis] > 0.0f)) use_median = true;  // all centroids coincide
        if (!use_median) {
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                if (!(ext[ax] > 0.0f)) continue;
                Box bb[BINS];
                uint32_t bc[BINS];
                for (int k = 0; k < BINS; k++) bb[k].reset(), bc[k] = 0;
                const float scale = float(BINS) / ext[ax];
                for (uint32_t i = begin; i < end; i++) {
                    int k = std::min(BINS - 1, std::max(0, int((centroid[3 * size_t(prims[i]) + ax] - cb.lo[ax]) * scale)));
                    bb[k].grow(pbox[prims[i]]);
                    bc[k]++;
                }
                float right_area[BINS];
                uint32_t right_cnt[BINS];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int k = BINS - 1; k > 0; k--) {
                    acc.grow(bb[k]);
                    cnt += bc[k];
                    right_area[k] = cnt ? acc.half_area() : 0.0f;
                    right_cnt[k] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int k = 0; k < BINS - 1; k++) {
                    acc.grow(bb[k]);
                    cnt += bc[k];
                    if (cnt == 0 || right_cnt[k + 1] == 0) continue;
                    float cost = acc.half_area() * float(cnt) + right_area[k + 1] * float(right_cnt[k + 1]);
                    if (cost < best_cost) best_cost = cost, best_axis = ax, best_bin = k;
                }
            }
            if (best_axis < 0) {
                use_median = true;
            } else {
                const float scale = float(BINS) / ext[best_axis];
                const float lo = cb.lo[best_axis];
                auto it = std::partition(prims.begin() + begin, prims.begin() + end, [&](uint32_t p) {
                    int k = std::min(BINS - 1, std::max(0, int((centroid[3 * size_t(p) + best_axis] - lo) * scale)));
                    return k <= best_bin;
                });
                mid = uint32_t(it - prims.begin());
                if (mid == begin || mid == end) use_median = true;
                else axis = best_axis;
            }
        }
        if (use_median) {
            mid = begin + n / 2;
            std::nth_element(prims.begin() + begin, prims.begin() + mid, prims.begin() + end, [&](uint32_t a, uint32_t c) {
                float ca = centroid[3 * size_t(a) + axis], cc = centroid[3 * size_t(c) + axis];
                return ca < cc || (ca == cc && a < c);
            });
        }
        Box lb, rb;
        uint32_t lref, rref;
        if (depth < PAR_DEPTH && n >= 2 * PAR_MIN) {
            std::vector<DevBvhNode> lnodes, rnodes;
            uint32_t llev = 0, rlev = 0;
            std::future<uint32_t> left;
            bool spawned = true;
            try {
                left = std::async(std::launch::async, [&] { return build(begin, mid, depth + 1, lb, lnodes, llev); });
            } catch (const std::system_error &) {
                spawned = false;  // no thread to be had: build it here
            }
            rref = build(mid, end, depth + 1, rb, rnodes, rlev);
            lref = spawned ? left.get() : build(begin, mid, depth + 1, lb, lnodes, llev);
            append_subtree(nodes, lnodes, lref);
            append_subtree(nodes, rnodes, rref);
            levels = std::max(levels, std::max(llev, rlev));
        } else {
            lref = build(begin, mid, depth + 1, lb, nodes, levels);
            rref = build(mid, end, depth + 1, rb, nodes, levels);
        }
        DevBvhNode &nd = nodes[me];
        for (int k = 0; k < 3; k++) nd.lo0[k] = lb.lo[k], nd.hi0[k] = lb.hi[k], nd.lo1[k] = rb.lo[k], nd.hi1[k] = rb.hi[k];
        nd.ref0 = lref, nd.ref1 = rref, nd._pad0 = nd._pad1 = 0;
        return me;
    }
};

inline float next_down(float x) { return std::nextafter(x, -INFINITY); }
inline float next_up(float x) { return std::nextafter(x, INFINITY); }

}  // namespace

uint32_t rt_build_bvh(const rthip_scene_desc *s, std::vector<DevBvhNode> &nodes, std::vector<uint32_t> &prims) {
    nodes.clear();
    prims.clear();
    const size_t ns = s->n_spheres, nt = s->n_triangles, n = ns + nt;
    if (n == 0) return 0;
    Builder B(prims);
    B.pbox.resize(n);
    B.centroid.resize(3 * n);
    for (size_t i = 0; i < ns; i++) {
        const rthip_sphere &sp = s->spheres[i];
        const float r = std::fabs(float(sp.radius));
        Box &b = B.pbox[i];
        for (int k = 0; k < 3; k++) {
            const float c = float(sp.center[k]);
            // pad by a few ulps: the float sphere test may accept points marginally outside the exact sphere
            const float pad = 4e-6f * (std::fabs(c) + r) + 1e-30f;
            b.lo[k] = next_down(c - r - pad), b.hi[k] = next_up(c + r + pad);
            B.centroid[3 * i + k] = c;
        }
    }
    for (size_t i = 0; i < nt; i++) {
        const rthip_triangle &t = s->triangles[i];
        Box &b = B.pbox[ns + i];
        b.reset();
        float v[3][3];
        for (int k = 0; k < 3; k++) v[0][k] = float(t.v0[k]), v[1][k] = float(t.v1[k]), v[2][k] = float(t.v2[k]);
        for (int j = 0; j < 3; j++) b.grow_pt(v[j]);
        for (int k = 0; k < 3; k++) {
            const float m = std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
            const float pad = 4e-6f * (m + (b.hi[k] - b.lo[k])) + 1e-30f;
            B.centroid[3 * (ns + i) + k] = 0.5f * (b.lo[k] + b.hi[k]);
            b.lo[k] = next_down(b.lo[k] - pad), b.hi[k] = next_up(b.hi[k] + pad);
        }
    }
    prims.resize(n);
    std::iota(prims.begin(), prims.end(), 0u);
    nodes.reserve(n / 2 + 16);
    Box root;
    uint32_t levels = 0;
    const uint32_t ref = B.build(0, uint32_t(n), 0, root, nodes, levels);
    if (ref & RT_BVH_LEAF_BIT) {
        // a single leaf: give it a root node whose second child is an empty (inverted, never hit) box
        DevBvhNode nd;
        for (int k = 0; k < 3; k++) nd.lo0[k] = root.lo[k], nd.hi0[k] = root.hi[k], nd.lo1[k] = INFINITY, nd.hi1[k] = -INFINITY;
        nd.ref0 = ref, nd.ref1 = ref, nd._pad0 = nd._pad1 = 0;
        nodes.push_back(nd);
        return 1;
    }
    return levels;
}


// Renumbers the nodes so that the top of the tree comes first, in breadth-first order (a parent before its children), up to
// `max_top` inner nodes; the others keep their depth-first order behind them. The traversal kernel stages a PREFIX of the array
// in LDS (any prefix of a breadth-first order is closed under "parent of"), so a ray's first visits — the ones every ray of a
// workgroup makes — are LDS reads instead of divergent global loads. Returns the number of nodes in breadth-first order.
uint32_t rt_bvh_top_first(std::vector<DevBvhNode> &nodes, uint32_t max_top) {
    const uint32_t n = uint32_t(nodes.size());
    if (n == 0 || max_top == 0) return 0;
    std::vector<uint32_t> order;  // new position -> old index
    order.reserve(n);
    std::vector<uint8_t> taken(n, 0);
    order.push_back(0);
    taken[0] = 1;
    for (size_t head = 0; head < order.size() && order.size() < max_top; head++) {
        const DevBvhNode &nd = nodes[order[head]];
        for (uint32_t ref : {nd.ref0, nd.ref1}) {
            if ((ref & RT_BVH_LEAF_BIT) || ref >= n || taken[ref] || order.size() >= max_top) continue;
            taken[ref] = 1;
            order.push_back(ref);
        }
    }
    const uint32_t n_top = uint32_t(order.size());
    for (uint32_t i = 0; i < n; i++)
        if (!taken[i]) order.push_back(i);
    std::vector<uint32_t> new_index(n);
    for (uint32_t pos = 0; pos < n; pos++) new_index[order[pos]] = pos;
    std::vector<DevBvhNode> out(n);
    for (uint32_t pos = 0; pos < n; pos++) {
        DevBvhNode nd = nodes[order[pos]];
        if (!(nd.ref0 & RT_BVH_LEAF_BIT)) nd.ref0 = new_index[nd.ref0];
        if (!(nd.ref1 & RT_BVH_LEAF_BIT)) nd.ref1 = new_index[nd.ref1];
        out[pos] = nd;
    }
    nodes.swap(out);
    return n_top;
}

namespace {
inline uint32_t f_bits(float x) { uint32_t u; std::memcpy(&u, &x, 4); return u; }
inline float bits_f(uint32_t u) { float x; std::memcpy(&x, &u, 4); return x; }
// Frame origin of an axis: a float <= x whatever the low 8 bits of its pattern are set to afterwards (they carry the step
// exponent and the kernel does NOT mask them off: the origin is the float with those bits in it).
inline uint32_t frame_base(float x) {
    uint32_t u = f_bits(x);
    if (!(u & 0x80000000u)) {
        u &= 0xffffff00u;
        return u >= 0x100u ? u - 0x100u : 0x80000100u;  // one frame step down; below +0: a tiny negative base
    }
    return (u | 0xffu) + 1u;  // negative: away from zero, low bits zero again (adding the exponent byte moves it further down)
}
}  // namespace

double rt_bvh_dequantize(uint32_t frame, uint32_t q8) {
    const double origin = double(bits_f(frame));
    const double step = std::ldexp(1.0, int(int8_t(frame & 0xffu)));
    return origin + double(q8) * step;
}

bool rt_bvh_quantize(const std::vector<DevBvhNode> &nodes, std::vector<DevBvhNodeQ> &out) {
    out.resize(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
        const DevBvhNode &n = nodes[i];
        DevBvhNodeQ &q = out[i];
        q.ref0 = n.ref0, q.ref1 = n.ref1;
        const bool empty0 = n.lo0[0] > n.hi0[0], empty1 = n.lo1[0] > n.hi1[0];
        for (int k = 0; k < 3; k++) {
            float lo = INFINITY, hi = -INFINITY;
            if (!empty0) lo = std::min(lo, n.lo0[k]), hi = std::max(hi, n.hi0[k]);
            if (!empty1) lo = std::min(lo, n.lo1[k]), hi = std::max(hi, n.hi1[k]);
            if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) return false;
            const uint32_t base = frame_base(lo);
            // the lowest value the origin can take once the exponent byte is in: positive patterns grow with it, negative ones fall
            const float lowest = bits_f((base & 0x80000000u) ? (base | 0xffu) : base);
            if (!std::isfinite(lowest) || lowest > lo) return false;
            const double extent = double(hi) - double(lowest);
            int e = -126;  // step = 2^e, a normal float; e travels as a signed byte
            if (extent > 0.0) {
                int ex;
                std::frexp(extent / 255.0, &ex);  // extent / 255 = m * 2^ex, m in [0.5, 1)  =>  2^ex >= extent / 255
                e = std::max(-126, ex);
            }
            while (e < 127 && 255.0 * std::ldexp(1.0, e) < extent) e++;
            if (255.0 * std::ldexp(1.0, e) < extent) return false;
            const uint32_t frame = base | (uint32_t(e) & 0xffu);
            const double origin = double(bits_f(frame)), step = std::ldexp(1.0, e);
            if (!(origin <= double(lo))) return false;
            auto qlo = [&](float v) { double t = std::floor((double(v) - origin) / step); return uint32_t(std::min(255.0, std::max(0.0, t))); };
            auto qhi = [&](float v) { double t = std::ceil((double(v) - origin) / step); return uint32_t(std::min(255.0, std::max(0.0, t))); };
            uint32_t l0 = 255, h0 = 0, l1 = 255, h1 = 0;
            if (!empty0) l0 = qlo(n.lo0[k]), h0 = qhi(n.hi0[k]);
            if (!empty1) l1 = qlo(n.lo1[k]), h1 = qhi(n.hi1[k]);
            // the decoded box must contain the float box
            if (!empty0 && !(rt_bvh_dequantize(frame, l0) <= double(n.lo0[k]) && rt_bvh_dequantize(frame, h0) >= double(n.hi0[k]))) return false;
            if (!empty1 && !(rt_bvh_dequantize(frame, l1) <= double(n.lo1[k]) && rt_bvh_dequantize(frame, h1) >= double(n.hi1[k]))) return false;
            q.frame[k] = frame;
            q.q[k] = l0 | (h0 << 8) | (l1 << 16) | (h1 << 24);
        }
    }
    return true;
}
