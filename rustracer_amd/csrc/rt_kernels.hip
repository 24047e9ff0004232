// rt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the rustracer render path.
//
// render_kernel<STATS, MODE, LIGHTS, SPH_SCENE>: a 256-thread workgroup = a 32x8 pixel tile, a wave = an 8x8 tile (coherent primary
// rays => coherent branches). The recursion of Ray::_project_ray (raytracer_core/src/shapes/ray.rs:69-87) is an iterative
// bounce loop with the path state (ray, throughput, bounce count, RNG counter) in VGPRs, and the `samples` loop of
// generate_pixel (lib.rs:281-288) is folded into the same loop by path regeneration.
//   MODE_LDS    (<= 256 spheres): lane = pixel. Sphere geometry and a merged per-sphere shading record are staged in LDS once
//               per workgroup and read back with wave-uniform (broadcast) ds_read_b128, so the closest-hit scan of
//               ray.rs:47-67 runs on VGPR operands; a sphere that no lane of the wave can hit is dropped after its
//               discriminant (wave-level early-out).
//   MODE_BVH    (big scenes, extension): 64-byte two-box nodes, leaf-ordered primitive records, a resumable while-while
//               traversal that yields when enough lanes have finished, and a per-wave pool of (pixel, sample) work items
//               so that finished lanes take new work instead of idling (bvh_run, render_kernel POOL).
//   MODE_GLOBAL (RTHIP_F_NO_BVH or > 256 spheres without a tree): the same linear scan from global memory, scalar loads.
// resolve_kernel: second half of a sample-sliced launch (small launches are split into groups of 64 samples per workgroup so
// that they fill the GPU to their end; the sums are grouped the same way when a launch is not sliced, so the image does not
// depend on it). compact_kernel: work-list of the pixels that are not yet Final (RTHIP_F_ADAPTIVE, lib.rs:276-313).
// pick_kernel: Raytracer::get_shape (lib.rs:262-267).
//
// The scan kernels are VALU-issue bound (profiles/): what matters is the instruction count per segment and the instruction
// class (measured on MI355X, tools/ubench/valu_rates.hip: VGPR-only VOP2 ~2.6-2.9 cycles per wave instruction, anything with
// an SGPR operand / VOP3-only / compare ~4.4, transcendental ~8.5). Hence:
//   * the ray direction is kept NORMALISED with its length on the side (the reference's t is in units of the un-normalised
//     direction: the window (T_MIN, T_MAX) is scaled by |d| instead); this removes a = d.d, the 1/a multiplies and every
//     later normalize();
//   * Lambertian and Metal share one random unit vector, and the scattered ray of a continuing lane shares its RNG hash and
//     its normalisation with the primary ray of a lane that starts a new sample (shade / next_ray);
//   * radiance is handed over where a path ends (emit), not through a merged return value.
// No MFMA: there is no dense contraction anywhere on this path.
#include <type_traits>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

#include "../../include/rthip.h"
#include "rt_device.h"
#include "rt_rng.h"
#include "rt_prepass.h"

#define RT_NONE 0xffffffffu
#ifndef RT_STABLE_SPHERE
#define RT_STABLE_SPHERE 1
#endif
#ifndef RT_SCAN_UNROLL
#define RT_SCAN_UNROLL 4u  /* MODE_LDS: spheres per group of the closest-hit scan; divides RT_LDS_SPHERES */
#endif

enum { MODE_LDS = 0, MODE_BVH = 1, MODE_GLOBAL = 2 };

typedef float lds_f4 __attribute__((ext_vector_type(4)));  // 16-byte aligned: LDS reads through it are ds_read_b128
typedef uint32_t lds_u4 __attribute__((ext_vector_type(4)));
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ f3 fma3(f3 a, float s, f3 b) { return f3{fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)}; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ f3 normalize(f3 a) { return a * fast_rsq(dot(a, a)); }
// number of set bits of `m` below this lane's
__device__ __forceinline__ uint32_t wave_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
}

// A ray with a UNIT direction; the reference's ray has an un-normalised direction of length `len`
// and measures t in units of it, so t_reference = t / len and the reference's fixed window
// T_MIN < t_ref < T_MAX (ray.rs:5-6) is tmin = T_MIN*len < t < tmax = T_MAX*len here.
struct Ray {
    f3 o, d;
    float tmin, tmax;
};

struct HitRec {
    float t;         // along the UNIT direction
    uint32_t shape;  // global shape index (spheres, then planes, then triangles); RT_NONE = miss
    float b1, b2;    // triangle barycentrics
};

// ---- primitive tests -------------------------------------------------------------------
// Sphere::collide — sphere.rs:29-61 with a = 1. g = {cx, cy, cz, r*r}.
// `on_surface`: the ray starts ON this sphere (it is the shape the path just bounced off). In the
// reference's f64, c = |oc|^2 - r^2 is then ~1e-13, i.e. the roots are 0 and -2*half_b. In FP32
// the same c carries ~1e-3 of cancellation error on the radius-100 ground sphere (|oc|^2 ~ 1e4),
// which turns ~0.1 % of grazing bounces into spurious self-hits beyond T_MIN; c is therefore set
// to its exact value 0 for that one sphere ("FP32 robustness").
// Returns the NEGATED roots n0 = -t0 = half_b + root, n1 = -t1 = half_b - root (plain add/sub, no
// source modifiers: those force the slower VOP3 encoding).
__device__ __forceinline__ void sphere_roots(rt_f4 g, const Ray &r, bool on_surface, float &n0, float &n1) {
    const f3 oc = mk3(r.o.x - g.x, r.o.y - g.y, r.o.z - g.z);
    const float half_b = dot(oc, r.d);
#if RT_STABLE_SPHERE
    // discriminant = r^2 - |oc - (oc.d) d|^2: the squared distance from the centre to the ray's line, taken from the vector itself.
    // half_b^2 - (|oc|^2 - r^2) is the same number, but in FP32 its error is ~1e-7 |oc|^2 — for a sphere of radius 0.1 seen from
    // 60 units away a tenth of a pixel of silhouette at 8K — where this form's is ~1e-7 r^2 ("hard parts").
    const f3 l = fma3(r.d, -half_b, oc);
    const float disc = on_surface ? half_b * half_b : g.w - dot(l, l);
#else
    float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -g.w)));
    c = on_surface ? 0.0f : c;
    const float disc = fmaf(half_b, half_b, -c);
#endif
    const float root = fast_sqrt(disc);  // NaN when disc < 0: every comparison on n0/n1 is then false
    n0 = half_b + root;
    n1 = half_b - root;
}
// The first root if it is inside the window, else the second (sphere.rs:38-58). t1 >= t0, hence
// "the chosen root is > tmin" <=> t1 > tmin, and when t0 fails only because t0 >= tmax, t1 fails too.
// A tangent ray (discriminant == 0 exactly, a miss in sphere.rs:36) is not told apart from a
// grazing hit: in FP32 the sign of a zero discriminant is rounding noise either way.
__device__ __forceinline__ bool hit_sphere(rt_f4 g, const Ray &r, bool on_surface, float &t_out) {
    float n0, n1;
    sphere_roots(g, r, on_surface, n0, n1);
    const float t = (n0 < -r.tmin) ? -n0 : -n1;
    t_out = t;
    return (n1 < -r.tmin) & (t < r.tmax);
}
// extension: plane. g0 = {nx, ny, nz, n.p0}
__device__ __forceinline__ bool hit_plane(rt_f4 g0, const Ray &r, float &t_out) {
    const f3 n = mk3(g0.x, g0.y, g0.z);
    const float denom = dot(r.d, n);
    const float t = (g0.w - dot(r.o, n)) * fast_rcp(denom);
    t_out = t;
    return denom != 0.0f && t < r.tmax && t > r.tmin;
}
// extension: Moller-Trumbore, two-sided. v0, e1 = v1-v0, e2 = v2-v0
__device__ __forceinline__ bool hit_tri(rt_f4 v0, rt_f4 e1_, rt_f4 e2_, const Ray &r, float &t_out, float &b1_out, float &b2_out) {
    const f3 e1 = mk3(e1_.x, e1_.y, e1_.z), e2 = mk3(e2_.x, e2_.y, e2_.z);
    const f3 pvec = cross(r.d, e2);
    const float det = dot(e1, pvec);
    const float inv = fast_rcp(det);
    const f3 tvec = mk3(r.o.x - v0.x, r.o.y - v0.y, r.o.z - v0.z);
    const float b1 = dot(tvec, pvec) * inv;
    const f3 qvec = cross(tvec, e1);
    const float b2 = dot(r.d, qvec) * inv;
    const float t = dot(e2, qvec) * inv;
    t_out = t, b1_out = b1, b2_out = b2;
    return det != 0.0f && b1 >= 0.0f && b1 <= 1.0f && b2 >= 0.0f && (b1 + b2) <= 1.0f && t < r.tmax && t > r.tmin;
}

// Ray::find_collision — ray.rs:47-67: every shape against the FIXED window, keep the old hit only
// if old.t < new.t (equal t: the later index wins).
#define RT_TAKE(ok, tnew, idx, ...)                  \
    if ((ok) && !(h.t < (tnew))) {                   \
        h.t = (tnew);                                \
        h.shape = (idx);                             \
        __VA_ARGS__;                                 \
    }
// the same rule when shapes are NOT visited in index order (BVH): smaller t, or equal t and larger index
#define RT_TAKE_ANY_ORDER(ok, tnew, idx, ...)                                                    \
    if ((ok) && ((tnew) < h.t || ((tnew) == h.t && (h.shape == RT_NONE || (idx) > h.shape)))) { \
        h.t = (tnew);                                                                            \
        h.shape = (idx);                                                                         \
        __VA_ARGS__;                                                                             \
    }

// ---- BVH traversal (extension; big scenes) ----------------------------------------------------
// "while-while" traversal (Aila & Laine 2009): every lane first walks inner nodes until it holds a
// leaf (or is done), then the wave intersects leaves together, so that box tests and primitive
// tests do not serialise against each other inside a wave. One 32-byte node fetch (2 independent
// 16-byte loads of 8-bit quantised boxes, DevBvhNodeQ) tests both children; the nearer child is entered first, the
// farther one is pushed on the per-lane stack: its first P.bvh_levels levels in LDS ([level][thread]: conflict-free),
// deeper ones in private memory.
// A child is entered iff the ray's slab interval overlaps (tmin, min(best t, tmax)] — INCLUSIVE of
// the current best t, so that an equal-t shape with a larger index is still found (ray.rs:55-61).
// Slab planes are evaluated as fma(q, step/d, fma(frame, 1/d, -o/d)) (q = the 8-bit plane index): against
// (plane - o)/d that moves a plane by a few ulps of its coordinate, which the builder's box padding (rt_bvh.cpp)
// covers many times over and the outward rounding of the quantiser only widens; a zero direction component gives
// NaN on that axis, which v_min/v_max drop (conservative).
#define RT_STACK_AT(level) s_stack[(level) * WG_THREADS + threadIdx.x]  /* WG_THREADS: threads of the calling workgroup (template parameter) */

// Traversal state of one lane; it survives across calls of bvh_run so that a lane whose ray is
// still inside the tree can be parked while the finished lanes of its wave shade and fetch new rays.
struct BvhState {
    f3 inv_d, noi;  // 1/d and -o/d
    uint32_t selx, sely, selz;  // v_perm selectors that put the near plane of each child first (by the sign of d)
    uint32_t ref;   // node/leaf reference in flight, RT_BVH_DONE when the ray has left the tree
    int sp;
};
// stack levels >= P.bvh_levels of one lane (private memory; rarely reached, see rt_launch_render)
struct BvhDeep { uint32_t at[RT_BVH_MAX_DEPTH]; };
__device__ __forceinline__ void bvh_begin(const Ray &r, BvhState &s) {
    s.inv_d = mk3(fast_rcp(r.d.x), fast_rcp(r.d.y), fast_rcp(r.d.z));
    s.noi = mk3(-r.o.x * s.inv_d.x, -r.o.y * s.inv_d.y, -r.o.z * s.inv_d.z);
    s.selx = s.inv_d.x < 0.0f ? 0x02030001u : 0x03020100u;
    s.sely = s.inv_d.y < 0.0f ? 0x02030001u : 0x03020100u;
    s.selz = s.inv_d.z < 0.0f ? 0x02030001u : 0x03020100u;
    s.ref = 0;  // the root is always an inner node
    s.sp = 0;
}
#ifndef RT_BVH_LEAF_DEN
#define RT_BVH_LEAF_DEN 3 /* the wave turns to its leaves once NUM / DEN of the lanes that are still inside the tree wait at one (0: at RT_BVH_LEAF_LANES
                             lanes, the rule until the end of round 3). A fixed count is too many when few rays of the wave are left and too few when all
                             64 are: 12 lanes -> a third: C4 44.8 -> 42.5 ms, C5 350.5 -> 348.4; a half: C5 362; 2/5: 352.9; 2/7, 1/4, 1/5: in between */
#endif
#ifndef RT_BVH_LEAF_NUM
#define RT_BVH_LEAF_NUM 1
#endif
#ifndef RT_BVH_VISITS_PER_CHECK
#define RT_BVH_VISITS_PER_CHECK 2 /* node visits per look at how many lanes wait at a leaf (two ballots and a branch per look): 1 -> 2: C4 46.4 -> 45.5 ms; 3: C5 +2 % */
#endif
#ifndef RT_BVH_LDS_LEVELS
#define RT_BVH_LDS_LEVELS 7u
#endif
#ifndef RT_BVH_TOP_NODES
#define RT_BVH_TOP_NODES 0  /* > 0: stage that many nodes of the top of the tree in LDS — measured slower at every size */
#endif
#ifndef RT_BVH_LEAF_LANES
#define RT_BVH_LEAF_LANES 12 /* (16 until the end of round 3; with two visits per look 12 is 1 % faster on C5, 8 and 20 slower) */
#endif
#ifndef RT_BVH_YIELD_LANES
#define RT_BVH_YIELD_LANES 40
#endif
#ifndef RT_POOL_REFILL_LANES
#define RT_POOL_REFILL_LANES 16u
#endif
// Runs the traversal of every calling lane. YIELD: returns early (s.ref != RT_BVH_DONE on the
// unfinished lanes) once RT_BVH_YIELD_LANES lanes of the wave have finished, so that the caller can
// give them new rays instead of idling them until the slowest ray of the wave is done (ray
// lengths in the tree are heavy-tailed: without this a wave runs at ~20 % lane utilisation).
// ANY: any-hit query, returns true on the first hit inside the window.
// byte k of a dword as float (the compiler selects v_cvt_f32_ubyte0..3)
__device__ __forceinline__ float ubyte0(uint32_t x) { return float(x & 0xffu); }
__device__ __forceinline__ float ubyte1(uint32_t x) { return float((x >> 8) & 0xffu); }
__device__ __forceinline__ float ubyte2(uint32_t x) { return float((x >> 16) & 0xffu); }
__device__ __forceinline__ float ubyte3(uint32_t x) { return float(x >> 24); }

// slab test from the six plane distances of a box (NaN = a zero direction component: dropped by v_min/v_max, conservative)
__device__ __forceinline__ bool slab_from_t(float ax, float bx, float ay, float by, float az, float bz, float tmin, float limit, float &tn_out) {
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000005f;
    tn_out = tn;
    return tn <= tf && tn <= limit && tf >= tmin;
}

// The primitives of one leaf against the ray: ray.rs:51-65 for shapes that are not visited in index order (smaller t, or equal t
// and larger index). ANY as in bvh_run; returns true if an any-hit query found a hit.
template <bool STATS, int ANY>
__device__ __forceinline__ bool leaf_test(const KernelParams &P, uint32_t ref, const Ray &r, uint32_t self, HitRec &h, bool any, uint32_t *cnt) {
    bool found = false;
    const uint32_t first = (ref & ~RT_BVH_LEAF_BIT) >> 2, count = (ref & 3u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
#ifdef RT_BVH_DEBUG_UTIL
        if (STATS) {
            const unsigned long long ex = __ballot(true);
            if (__builtin_amdgcn_mbcnt_hi(uint32_t(ex >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ex), 0u)) == 0) cnt[RT_STAT_HIT0] += 64;
            cnt[RT_STAT_HIT0 + 1]++;
        }
#endif
        const rt_f4 *rec = P.bvh_recs + 3 * size_t(first + k);
        const rt_f4 a = rec[0], b = rec[1], c = rec[2];
        const uint32_t id = __float_as_uint(a.w);
        if (__float_as_uint(b.w) == 0u) {
            float t;
            bool ok = hit_sphere(rt_f4{a.x, a.y, a.z, b.x}, r, id == self, t);
            if (STATS) cnt[RT_STAT_SPHERE]++;
            if (ANY == 1) {
                found |= ok;
            } else if (ANY == 2) {
                found |= ok && any;
                ok = ok && !any;
                RT_TAKE_ANY_ORDER(ok, t, id, (void)0)
            } else {
                RT_TAKE_ANY_ORDER(ok, t, id, (void)0)
            }
        } else {
            float t, b1, b2;
            bool ok = hit_tri(a, b, c, r, t, b1, b2);
            if (STATS) cnt[RT_STAT_TRI]++;
            if (ANY == 1) {
                found |= ok;
            } else if (ANY == 2) {
                found |= ok && any;
                ok = ok && !any;
                RT_TAKE_ANY_ORDER(ok, t, id, (h.b1 = b1, h.b2 = b2))
            } else {
                RT_TAKE_ANY_ORDER(ok, t, id, (h.b1 = b1, h.b2 = b2))
            }
        }
    }
    return found;
}

// ANY: 0 = closest-hit query, 1 = any-hit query, 2 = per lane (any_lane): closest-hit and shadow rays of a wave walk the tree together
template <bool STATS, int ANY, bool YIELD, int WG_THREADS = 256>
__device__ __forceinline__ bool bvh_run(const KernelParams &P, const Ray &r, uint32_t self, HitRec &h, BvhState &s, BvhDeep &deep, uint32_t *cnt, uint32_t *s_stack,
                                        bool any_lane = false, const lds_u4 *s_top = nullptr, uint32_t n_top = 0) {
    const bool any = ANY == 2 ? any_lane : (ANY == 1);
    const uint4 *nodes = (const uint4 *)P.bvh_nodes;  // DevBvhNodeQ: 2 x 16 bytes
    uint32_t ref = s.ref;
    int sp = s.sp;
    bool found = false;
    const int lds_levels = int(P.bvh_levels);
    // (the asm statements keep the compiler from merging the LDS and the private access into one flat_ access)
    auto push = [&](uint32_t x) {
        if (sp < lds_levels) {
            RT_STACK_AT(sp) = x;
            asm volatile("" ::: "memory");
        } else {
            deep.at[sp - lds_levels] = x;
            asm volatile("" ::: "memory");
        }
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        sp--;
        uint32_t x;
        if (sp < lds_levels) {
            x = RT_STACK_AT(sp);
            asm volatile("" : "+v"(x));
        } else {
            x = deep.at[sp - lds_levels];
            asm volatile("" : "+v"(x));
        }
        return x;
    };
    for (;;) {
        // inner nodes: every lane walks down until RT_BVH_LEAF_LANES lanes of the wave wait at a leaf (waiting for
        // the LAST lane to reach one leaves three quarters of the wave idle: the walk lengths are heavy-tailed)
        while (__ballot(ref < RT_BVH_DONE) != 0ull) {
#ifdef RT_BVH_DEBUG_UTIL
          if (STATS) {
              const unsigned long long ex = __ballot(true);
              if (__builtin_amdgcn_mbcnt_hi(uint32_t(ex >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ex), 0u)) == 0) cnt[RT_STAT_SKY] += 64;
              if (ref < RT_BVH_DONE) cnt[RT_STAT_SHADOW]++;
          }
#endif
          auto visit = [&]() {
          if (ref < RT_BVH_DONE) {
            // {frame x, y, z, ref0} {q x, y, z, ref1}. RT_BVH_TOP_NODES > 0: the first n_top nodes (the top of the tree, breadth-first)
            // are staged in LDS
            uint4 na, nb;
#if RT_BVH_TOP_NODES > 0
            // (measured and rejected, DESIGN.md §5: the lanes of a wave sit at all depths of the tree, so a visit issues BOTH the
            // LDS and the global path; the code is kept behind the macro for the record)
            if (n_top != 0u && ref < n_top) {
                const lds_u4 a = s_top[2 * ref], b = s_top[2 * ref + 1];
                na = make_uint4(a.x, a.y, a.z, a.w), nb = make_uint4(b.x, b.y, b.z, b.w);
                asm volatile("" : "+v"(na.x), "+v"(na.y), "+v"(na.z), "+v"(na.w), "+v"(nb.x), "+v"(nb.y), "+v"(nb.z), "+v"(nb.w));  // keep the two paths apart (no flat_ loads)
            } else
#endif
            {
                na = nodes[2 * ref], nb = nodes[2 * ref + 1];
            }
            if (STATS) cnt[RT_STAT_NODE] += 2;
            const float limit = any ? r.tmax : fminf(h.t, r.tmax);
            // per axis: plane distance t = (o + q*step)/d - origin/d = q * (step/d) + (o/d - origin/d)
            const float Ax = ldexpf(s.inv_d.x, __builtin_amdgcn_sbfe(int(na.x), 0, 8)), Bx = fmaf(__uint_as_float(na.x), s.inv_d.x, s.noi.x);
            const float Ay = ldexpf(s.inv_d.y, __builtin_amdgcn_sbfe(int(na.y), 0, 8)), By = fmaf(__uint_as_float(na.y), s.inv_d.y, s.noi.y);
            const float Az = ldexpf(s.inv_d.z, __builtin_amdgcn_sbfe(int(na.z), 0, 8)), Bz = fmaf(__uint_as_float(na.z), s.inv_d.z, s.noi.z);
            // bytes {near0, far0, near1, far1} per axis: with the planes picked by the sign of d, a child's entry distance is
            // the max of three, its exit the min of three (NaN = zero direction component on that axis: dropped)
            const uint32_t px = __builtin_amdgcn_perm(nb.x, nb.x, s.selx), py = __builtin_amdgcn_perm(nb.y, nb.y, s.sely), pz = __builtin_amdgcn_perm(nb.z, nb.z, s.selz);
            const float tn0 = fmaxf(fmaxf(fmaf(ubyte0(px), Ax, Bx), fmaf(ubyte0(py), Ay, By)), fmaf(ubyte0(pz), Az, Bz));
            const float tf0 = fminf(fminf(fmaf(ubyte1(px), Ax, Bx), fmaf(ubyte1(py), Ay, By)), fmaf(ubyte1(pz), Az, Bz)) * 1.0000005f;
            const float tn1 = fmaxf(fmaxf(fmaf(ubyte2(px), Ax, Bx), fmaf(ubyte2(py), Ay, By)), fmaf(ubyte2(pz), Az, Bz));
            const float tf1 = fminf(fminf(fmaf(ubyte3(px), Ax, Bx), fmaf(ubyte3(py), Ay, By)), fmaf(ubyte3(pz), Az, Bz)) * 1.0000005f;
            const bool h0 = tn0 <= tf0 && tn0 <= limit && tf0 >= r.tmin;
            const bool h1 = tn1 <= tf1 && tn1 <= limit && tf1 >= r.tmin;
            const uint32_t r0 = na.w, r1 = nb.w;
            if (h0 && h1) {
                const bool second_first = tn1 < tn0;
                push(second_first ? r0 : r1);
                ref = second_first ? r1 : r0;
            } else if (h0 || h1) {
                ref = h0 ? r0 : r1;
            } else if (sp > 0) {
                ref = pop();
            } else {
                ref = RT_BVH_DONE;
            }
          }
          };
          visit();
#if RT_BVH_VISITS_PER_CHECK > 1
          visit();  // (a second node before the wave looks at how many lanes wait at a leaf)
#endif
#if RT_BVH_VISITS_PER_CHECK > 2
          visit();
#endif
#if RT_BVH_LEAF_DEN
          if (uint32_t(__popcll(__ballot(ref > RT_BVH_DONE))) * RT_BVH_LEAF_DEN >= uint32_t(__popcll(__ballot(ref != RT_BVH_DONE))) * RT_BVH_LEAF_NUM) break;
#else
          if (__popcll(__ballot(ref > RT_BVH_DONE)) >= RT_BVH_LEAF_LANES) break;
#endif
        }
        if (ref > RT_BVH_DONE) {  // leaf
            found |= leaf_test<STATS, ANY>(P, ref, r, self, h, any, cnt);
            if (sp > 0 && !(any && found)) {
                ref = pop();
            } else {
                ref = RT_BVH_DONE;
            }
        }
        const unsigned long long finished = __ballot(ref == RT_BVH_DONE), here = __ballot(true);
        if (finished == here) break;
        if (YIELD && __popcll(finished) >= RT_BVH_YIELD_LANES) break;
    }
    s.ref = ref;
    s.sp = sp;
    return found;
}

// MODE_BVH, first half of the closest-hit query: the shapes outside the tree (planes are unbounded)
template <bool STATS>
__device__ __forceinline__ HitRec closest_hit_unbounded(const KernelParams &P, const Ray &r, uint32_t *cnt) {
    HitRec h;
    h.t = __uint_as_float(__float_as_uint(r.tmax) - 1u);  // largest float below tmax (tmax > 0)
    h.shape = RT_NONE;
    h.b1 = h.b2 = 0.0f;
    for (uint32_t i = 0; i < P.n_planes; i++) {
        float t;
        bool ok = hit_plane(P.pln_geom[4 * i], r, t);
        RT_TAKE(ok, t, P.n_spheres + i, (void)0)
    }
    if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
    return h;
}

// The triangles of a linear-scan scene against the closest hit so far, in index order (ray.rs:47-67 continues over them):
// hit_tri, with a wave-level early-out after the first barycentric (same arithmetic: the skipped half would reject every lane anyway)
__device__ __forceinline__ void tris_closest(const KernelParams &P, const Ray &r, HitRec &h) {
    for (uint32_t i = 0; i < P.n_tris; i++) {
        const rt_f4 v0 = P.tri_geom[3 * i], e1_ = P.tri_geom[3 * i + 1], e2_ = P.tri_geom[3 * i + 2];
        const f3 e1 = mk3(e1_.x, e1_.y, e1_.z), e2 = mk3(e2_.x, e2_.y, e2_.z);
        const f3 pvec = cross(r.d, e2);
        const float det = dot(e1, pvec);
        const float inv = fast_rcp(det);
        const f3 tvec = mk3(r.o.x - v0.x, r.o.y - v0.y, r.o.z - v0.z);
        const float b1 = dot(tvec, pvec) * inv;
        const bool first = det != 0.0f && b1 >= 0.0f && b1 <= 1.0f;
        if (__ballot(first) == 0ull) continue;
        const f3 qvec = cross(tvec, e1);
        const float b2 = dot(r.d, qvec) * inv;
        const float t = dot(e2, qvec) * inv;
        const bool ok = first && b2 >= 0.0f && (b1 + b2) <= 1.0f && t < r.tmax && t > r.tmin;
        RT_TAKE(ok, t, P.n_spheres + P.n_planes + i, (h.b1 = b1, h.b2 = b2))
    }
}

// SPH: the caller knows that the scene holds spheres only (no plane, no triangle): their (empty) loops are left out
template <bool STATS, int MODE, bool SPH = false>
__device__ __forceinline__ HitRec closest_hit(const KernelParams &P, const Ray &r, uint32_t self, uint32_t *cnt, uint32_t *s_stack,
                                              const rt_f4 *s_geom) {
    HitRec h;
    h.t = __uint_as_float(__float_as_uint(r.tmax) - 1u);  // largest float below tmax (tmax > 0)
    h.shape = RT_NONE;
    h.b1 = h.b2 = 0.0f;
    if (MODE == MODE_LDS) {
        // tracked negated: nbest = -h.t, starting just above -tmax, so "not (n < nbest)" <=> t <= best < tmax
        float nbest = -h.t;
        float ntmin = -r.tmin;
        // The staged array is padded to a multiple of RT_SCAN_UNROLL with spheres that cannot be hit (NaN centre: the discriminant
        // is NaN), so the loop runs in groups of RT_SCAN_UNROLL with constant LDS offsets and constant indices (one address
        // register per group instead of an address and an index move per sphere; the group's broadcast reads issue together).
        for (uint32_t base = 0; base < P.n_spheres; base += RT_SCAN_UNROLL) {  // wave-uniform LDS address: broadcast read, VGPR operands
            // the group's LDS address in ONE vector register (opaque to the compiler, or it re-creates the address from the
            // scalar loop counter before every read)
            uint32_t grp_addr = uint32_t(size_t((const __attribute__((address_space(3))) rt_f4 *)s_geom)) + base * 16u;
            asm volatile("" : "+v"(grp_addr));
            const __attribute__((address_space(3))) lds_f4 *grp_geom = (const __attribute__((address_space(3))) lds_f4 *)uintptr_t(grp_addr);
#pragma unroll
            for (uint32_t k = 0; k < RT_SCAN_UNROLL; k++) {
                const uint32_t i = base + k;
                const lds_f4 gv = grp_geom[k];  // one ds_read_b128 (16-byte aligned vector type)
                const rt_f4 g = rt_f4{gv.x, gv.y, gv.z, gv.w};
                const f3 oc = mk3(r.o.x - g.x, r.o.y - g.y, r.o.z - g.z);
                const float half_b = dot(oc, r.d);
                float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -g.w)));
                c = (i == self) ? 0.0f : c;  // sphere_roots: exact c for the sphere the ray starts on
                const float disc = fmaf(half_b, half_b, -c);
                // most spheres are missed by every ray of the wave (coherent primary rays, small spheres): then the square root,
                // both roots and the window/closest selects (half of the test, ~40 issue cycles) are skipped. A lane's own result
                // never depends on its neighbours: with disc < 0 (or NaN) the skipped code selects nothing either.
                if (__ballot(disc >= 0.0f) == 0ull) continue;
                const float root = fast_sqrt(disc);
                const float n0 = half_b + root, n1 = half_b - root;
                const float n = (n0 < ntmin) ? n0 : n1;
                const bool take = (n1 < ntmin) & !(n < nbest);
                nbest = take ? n : nbest;
                h.shape = take ? i : h.shape;
            }
        }
        h.t = -nbest;
    } else if (MODE == MODE_GLOBAL) {
        for (uint32_t i = 0; i < P.n_spheres; i++) {  // wave-uniform index: scalar loads
            float t;
            bool ok = hit_sphere(P.sph_geom[i], r, i == self, t);
            RT_TAKE(ok, t, i, (void)0)
        }
    }
    if (SPH) {  // (a linear-scan mode: MODE_LDS)
        if (STATS) cnt[RT_STAT_SPHERE] += P.n_spheres;
        return h;
    }
    for (uint32_t i = 0; i < P.n_planes; i++) {
        float t;
        bool ok = hit_plane(P.pln_geom[4 * i], r, t);
        RT_TAKE(ok, t, P.n_spheres + i, (void)0)
    }
    if (MODE != MODE_BVH) {
        tris_closest(P, r, h);
        if (STATS) {
            cnt[RT_STAT_SPHERE] += P.n_spheres;
            cnt[RT_STAT_TRI] += P.n_tris;
        }
    } else {
        BvhState bs;
        BvhDeep deep;
        bvh_begin(r, bs);
        bvh_run<STATS, 0, false>(P, r, self, h, bs, deep, cnt, s_stack);
    }
    if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
    return h;
}

// extension: shadow-ray any-hit for the point-light mode: the reference-style ray is (p, light - p),
// occluded iff some shape is hit with T_MIN < t_ref < 1, i.e. T_MIN*|L-p| < t < |L-p| on the unit ray.
template <bool STATS, int MODE>
__device__ __forceinline__ bool occluded(const KernelParams &P, const Ray &r, uint32_t self, uint32_t *cnt, uint32_t *s_stack, const rt_f4 *s_geom) {
    bool occ = false;
    float t, b1, b2;
    if (MODE != MODE_BVH) {
        if (MODE == MODE_LDS) {
            // as in closest_hit: a sphere whose discriminant is negative for every lane of the wave (nearly all of them for
            // shadow rays, which run from a surface to the light) is dropped before the square root and the window tests
            for (uint32_t i = 0; i < P.n_spheres; i++) {
                const rt_f4 g = s_geom[i];
                const f3 oc = mk3(r.o.x - g.x, r.o.y - g.y, r.o.z - g.z);
                const float half_b = dot(oc, r.d);
                float c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -g.w)));
                c = (i == self) ? 0.0f : c;
                const float disc = fmaf(half_b, half_b, -c);
                if (__ballot(disc >= 0.0f) == 0ull) continue;
                const float root = fast_sqrt(disc);
                const float n0 = half_b + root, n1 = half_b - root;
                const float tt = (n0 < -r.tmin) ? -n0 : -n1;  // hit_sphere
                occ |= (n1 < -r.tmin) & (tt < r.tmax);
            }
        } else {
            for (uint32_t i = 0; i < P.n_spheres; i++) occ |= hit_sphere(P.sph_geom[i], r, i == self, t);
        }
        for (uint32_t i = 0; i < P.n_tris; i++) occ |= hit_tri(P.tri_geom[3 * i], P.tri_geom[3 * i + 1], P.tri_geom[3 * i + 2], r, t, b1, b2);
        if (STATS) {
            cnt[RT_STAT_SPHERE] += P.n_spheres;
            cnt[RT_STAT_TRI] += P.n_tris;
        }
    }
    for (uint32_t i = 0; i < P.n_planes; i++) occ |= hit_plane(P.pln_geom[4 * i], r, t);
    if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
    if (MODE == MODE_BVH && !occ) {
        HitRec dummy;
        dummy.t = r.tmax;
        dummy.shape = RT_NONE;
        BvhState bs;
        BvhDeep deep;
        bvh_begin(r, bs);
        occ = bvh_run<STATS, 1, false>(P, r, self, dummy, bs, deep, cnt, s_stack);
    }
    return occ;
}

// Lambertian/Metal::random_unit_vector — lambertian_diffuse.rs:39-45, metal.rs:31-37 (a first, then z).
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so a = TAU*xi needs no multiply.
__device__ __forceinline__ f3 random_unit_vector(uint32_t h) {
    const float xa = rt_unit(h);
    const float z = fmaf(2.0f, rt_unit(rt_second(h)), -1.0f);
    const float r = fast_sqrt(fmaf(-z, z, 1.0f));
    return mk3(r * __builtin_amdgcn_cosf(xa), r * __builtin_amdgcn_sinf(xa), z);
}

// Texture::wrap — texture.rs:32-41 (`as i32`: truncate, saturate, NaN -> 0)
// The cast is the hardware's v_cvt_i32_f32 (saturating, NaN -> 0: exactly Rust's `as i32`); the remainder uses the texture's
// precomputed magic = floor(2^32 / bound) (DevTexInfo): q = mulhi(|ic|, magic) is the quotient or one below it, so one
// correction makes it exact for every |ic| <= 2^31 — 12 instructions instead of the 40 of a generic signed 32-bit `%`.
__device__ __forceinline__ int32_t texture_cast(float val, float scale, float bound) {  // bound: the texture's width or height as a float
    const float fc = val * scale * bound;
    int32_t ic;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(ic) : "v"(fc));
    return ic;
}
__device__ __forceinline__ uint32_t texture_rem(uint32_t a, uint32_t bound, uint32_t magic) {  // a % bound, a <= 2^31
    const uint32_t rem = a - __umulhi(a, magic) * bound;
    return rem >= bound ? rem - bound : rem;
}
__device__ __forceinline__ uint32_t texture_wrap_signed(int32_t ic, uint32_t bound, uint32_t magic) {
    const uint32_t a = ic < 0 ? 0u - uint32_t(ic) : uint32_t(ic);  // |ic| (2^31 for INT_MIN)
    const uint32_t rem = texture_rem(a, bound, magic);
    // Rust: w = ic % bound keeps the sign of ic; if w < 0 { w + bound }
    return (ic < 0 && rem != 0u) ? bound - rem : rem;
}
// atan2(y, x) for finite arguments (sphere.rs:70; the arguments are components of p - c): the odd minimax polynomial of the
// ROCm device library's atan2f on min/max, then the octant fix-ups — without that routine's handling of infinities, NaN classes
// and denormal scaling (~20 instructions less). atan2(0, 0) = 0 like libm.
__device__ __forceinline__ float atan2_finite(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * fast_rcp(fmaxf(mx, 1e-37f));
    const float s2 = t * t;
    float q = fmaf(s2, 0x1.5a54bp-9f, -0x1.f4b218p-7f);
    q = fmaf(s2, q, 0x1.53f67ep-5f);
    q = fmaf(s2, q, -0x1.2fa9aep-4f);
    q = fmaf(s2, q, 0x1.b26364p-4f);
    q = fmaf(s2, q, -0x1.22c1ccp-3f);
    q = fmaf(s2, q, 0x1.99717ep-3f);
    q = fmaf(s2, q, -0x1.5554c4p-2f);
    float a = fmaf(t, s2 * q, t);
    a = ay > ax ? 1.57079637f - a : a;
    a = x < 0.0f ? 3.14159274f - a : a;
    return __builtin_copysignf(a, y);
}
// acos(x) / pi (sphere.rs:71) = sqrt(1 - |x|) * P(|x|), mirrored for x < 0: Abramowitz & Stegun 4.4.46 (|error| <= 2e-8 in exact
// arithmetic) with the coefficients divided by pi. In FP32 its largest error over [-1, 1] is 8.9e-8, the device library's acosf
// followed by the multiply 1.05e-7 (2 000 001 points against double) — at 13 instructions instead of 22. |x| > 1 (a rounded
// (p - c).y / r) gives NaN like acosf, which the texture wrap's cast turns into texel row 0 like Rust's `as i32`.
__device__ __forceinline__ float acos_over_pi(float x) {
    float t = fabsf(x);
    asm volatile("" : "+v"(t));  // |x| in a register of its own: as a source modifier it forces the eight FMAs into the long (VOP3) encoding
    float q = fmaf(t, -0x1.a56262p-12f, 0x1.164948p-9f);
    q = fmaf(t, q, -0x1.6478a2p-8f);
    q = fmaf(t, q, 0x1.4236cap-7f);
    q = fmaf(t, q, -0x1.05ab22p-6f);
    q = fmaf(t, q, 0x1.d00ad2p-6f);
    q = fmaf(t, q, -0x1.17cb18p-4f);
    q = fmaf(t, q, 0.5f);
    const float v = fast_sqrt(1.0f - t) * q;
    return x < 0.0f ? 1.0f - v : v;
}

// what shading needs to know about the surface that was hit
struct Surface {
    f3 n;              // Shape::normal_at_position: (p - c)/r for spheres (sphere.rs:63-65), never flipped
    f3 tm;             // scatter factor albedo/pi*3
    float param;       // fuzziness | ior | texture scale
    float r1;          // dielectric: ((1-ior)/(1+ior))^2
    float inv_ior;     // dielectric: 1/ior
    uint32_t kind;     // RTHIP_MAT_*
    uint32_t mat;      // material index (texture / light terms are fetched from DevMaterial when needed)
    float tu, tv, tw;  // sphere: (p-c).z, (p-c).x, (p-c).y/r for the texture coordinates (sphere.rs:67-73)
    f3 la;             // point-light extension, scan modes: albedo/pi (the direct-light term) ...
    float shin;        // ... and the Phong exponent
};

// The end of generate_pixel for one pixel — lib.rs:289-313: add the prior running sum, store it, tonemap, and (adaptive)
// the convergence rule. Returns the bits to add to the PIX_TRACED / PIX_FINAL counters.
__device__ __forceinline__ void finish_pixel(const KernelParams &P, size_t oi, f3 acc, uint32_t &traced, uint32_t &fin_out) {
    if (P.accumulate) {
        acc.x += P.accum[oi * 3 + 0];
        acc.y += P.accum[oi * 3 + 1];
        acc.z += P.accum[oi * 3 + 2];
    }
    P.accum[oi * 3 + 0] = acc.x;
    P.accum[oi * 3 + 1] = acc.y;
    P.accum[oi * 3 + 2] = acc.z;
    if (P.accum_out) {
        P.accum_out[oi * 3 + 0] = acc.x;
        P.accum_out[oi * 3 + 1] = acc.y;
        P.accum_out[oi * 3 + 2] = acc.z;
    }
    if (P.rgba8 || P.pix_state) {
        // mean, clamp(0,1), sqrt, *255, `as u8` (truncate, saturate, NaN -> 0): lib.rs:294-299,30-38
        float cr = __fsqrt_rn(fminf(fmaxf(acc.x * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
        float cg = __fsqrt_rn(fminf(fmaxf(acc.y * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
        float cb = __fsqrt_rn(fminf(fmaxf(acc.z * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
        const uint32_t rgb = uint32_t(cr) | (uint32_t(cg) << 8) | (uint32_t(cb) << 16);
#ifdef RT_TIMELINE
        if (P.rgba8 && P.n_slices <= 1u) P.rgba8[oi] = rgb | 0xff000000u;  // (debug: the render kernel's time stamps stay, see there)
#else
        if (P.rgba8) P.rgba8[oi] = rgb | 0xff000000u;
#endif
        if (P.pix_state) {
            // lib.rs:300-313: compare with last_color, count equal passes, Final after more than 3
            const uint32_t st = P.pix_state[oi];
            uint32_t count = (st >> RT_PIX_COUNT_SHIFT) & RT_PIX_COUNT_MASK;
            if (st & RT_PIX_HAS_LAST) count = ((st & RT_PIX_RGB_MASK) == rgb) ? count + 1u : 0u;
            const bool fin = count > 3u;  // MAX_SIMILAR_SAMPLE_FOR_A_PIXEL, lib.rs:244,309
            P.pix_state[oi] = rgb | (count << RT_PIX_COUNT_SHIFT) | (fin ? RT_PIX_FINAL : 0u) | RT_PIX_HAS_LAST;
            traced++;
            fin_out += fin ? 1u : 0u;
        }
    }
}

// One full pass over the frame: generate_pixel (lib.rs:269-316) with project_ray (ray.rs:42-87) inlined.
//
// Work distribution inside a wave (a wave owns an 8x8 pixel tile).
//  * Scan modes (MODE_LDS, MODE_GLOBAL): lane = pixel; a lane whose path ended starts its pixel's next
//    sample at once (path regeneration), so lanes only idle at the very end of a tile.
//  * MODE_BVH (POOL): per-ray cost in a tree is heavy-tailed and differs a lot between the pixels of a
//    tile (sky vs. mesh silhouette): with lane = pixel 80 % of the lanes idle. So a lane does NOT own
//    a pixel while tracing: the tile's (pixel, sample) pairs are work items, item i = (pixel i & 63,
//    sample i >> 6), and a lane whose path has ended takes the next item (ballot + prefix count, no
//    atomics). Each finished item stores its radiance in its own LDS slot, in a ring of `P.ring_rows`
//    sample rows; when the last path of the oldest row has ended lane p adds that row's slot of pixel p
//    to its sum, so the samples are added in sample order exactly like the sequential loop of
//    lib.rs:281-288,
//    independent of which lane traced what. (Measured: the pool costs the scan modes 9 % — their
//    idle lanes come from divergent shading branches, not from unequal pixels — hence not used there.)
//
// Dynamic LDS layout (bytes): [MODE_LDS: padded n_spheres x 16 geometry | 4 x n_spheres x 16 shading | 64 x 16 camera terms]
// [MODE_BVH, per wave (a workgroup is one wave there): bvh_levels x 64 x 4 traversal stack | 8 x 4 image rows of the tile's rows
// (work-list passes: 64 x 4 x | y << 16) | 3 x 64*ring_rows x 4 radiance slots | 16 x 4 finished paths
// per ring row | 16 x 8 "ended at a texture" masks per ring row]; rt_launch_render computes the size.
// MODE_BVH needs resident waves to cover its node fetches (every visit is a dependent, divergent 32-byte fetch): its speed follows
// the number of waves (measured on C4/C5: 4 -> 5 -> 6 -> 7 waves per SIMD = 1.00 / 1.11 / 1.18 / 1.21), so the kernel is held to
// 72 VGPRs (spills outside the traversal loop) and its LDS to 4 allocation blocks of 1280 B per wave (see rt_launch_render).
#ifdef RT_ARM_COUNTS  // debug build (with RT_CLASS_CYCLES): how often a divergent arm runs (x 64) and for how many lanes
#define RT_ARM_COUNT(slot_execs, slot_lanes)                                                                                    \
    {                                                                                                                           \
        const unsigned long long ex_ = __ballot(true);                                                                          \
        if (__builtin_amdgcn_mbcnt_hi(uint32_t(ex_ >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ex_), 0u)) == 0) cnt[slot_execs] += 64; \
        cnt[slot_lanes]++;                                                                                                      \
    }
#else
#define RT_ARM_COUNT(slot_execs, slot_lanes)
#endif
#ifndef RT_PRIMARY_INLINE
#define RT_PRIMARY_INLINE 1
#endif
#ifndef RT_FAST_MIN_SPP
#define RT_FAST_MIN_SPP 4u /* the per-wave cone tests cost about one sample: passes of fewer samples (the 1-spp progressive pass) skip them, 4K 1 spp 0.176 -> 0.160 ms */
#endif
#define RT_PRIMARY_SPHERES 64u
#ifndef RT_TEX_QUEUE
#define RT_TEX_QUEUE 0 /* 1: the variant with the deferred texture queue (librthip_texqueue.so). Bit-identical images; measured on C2: 5 % fewer
                          VALU instructions, lane utilisation 0.66 -> 0.74, and no faster (12.27 vs 12.17 ms): 50 % more LDS instructions, 12
                          spilled registers, +30 % wait cycles (profiles/r03_texqueue_summary.md). Off by default. */
#endif
// deferred texture queue, per wave: 64 entries x 9 dwords {T, a, b, c, scale, texture, meta} | 64 x u64 owner masks |
// 2 x 3 x 64 floats: per lane, the deferred sums of the two groups in flight | 3 x 64 floats: a closed group waiting for its entries
#define RT_QUEUE_WAVE_DWORDS (576u + 128u + 384u + 192u)
#define RT_PRIMARY_TRIS 4u /* ... and takes scenes with at most this many planes and this many triangles beside the spheres (it tests them all per primary ray) */ /* the primary-inline loop keeps one reachability bit per sphere */
#ifndef RT_SCAN_LIGHT_WAVES
#define RT_SCAN_LIGHT_WAVES 5
#endif
#ifndef RT_BVH_LIGHT_WAVES
#define RT_BVH_LIGHT_WAVES 5
#endif
#ifndef RT_SCAN_WAVES
#define RT_SCAN_WAVES 7
#endif
#ifndef RT_SCAN_SPH_WAVES
#define RT_SCAN_SPH_WAVES 7 /* 8 waves (64 VGPRs) measured 0.8 % faster on C2, but its 8 spilled registers add 170 MB of scratch traffic per launch to a kernel whose outputs are 133 MB */
#endif
#ifndef RT_BVH_WAVES
#define RT_BVH_WAVES 7
#endif
// SPH_SCENE (MODE_LDS only): the scene holds spheres only — no plane, no triangle — so their code is left out of every loop.
#ifndef RT_POOL_WAVE_WG
#define RT_POOL_WAVE_WG 1 /* MODE_BVH: a workgroup is ONE wave (its 8x8 tile), not the four of a 32x8 pixel group: a wave's slot, registers and LDS are
                             free the moment IT ends — the four tiles of a group differ a lot in cost (sky / mesh silhouette), and with 256-thread
                             workgroups a third of the resident-wave time was waves waiting for their slowest sibling */
#endif
#ifndef RT_SCAN_WAVE_WG
#define RT_SCAN_WAVE_WG 1 /* MODE_LDS, scenes of few spheres: one-wave workgroups too (the instantiation VAR = true; rt_launch_render picks it when a
                             wave's own copy of the staged scene fits 4 LDS allocation blocks, i.e. 28 waves still fit a CU). A wave's slot is free
                             when IT ends: C2 12.18 -> 12.03 ms, its 1/8-frame launches 1.644 -> 1.609 ms. (Round 3 first measured this 1 % SLOWER:
                             the staged geometry then took RT_LDS_SPHERES entries whatever the scene, 5 blocks per wave, 25 waves per CU.) */
#endif
// VAR: the fifth template parameter of render_kernel — MODE_BVH: the primary-visibility list code (LISTS); MODE_LDS: one-wave workgroups
#define RT_WG_THREADS(MODE, VAR) (((MODE) == MODE_BVH ? RT_POOL_WAVE_WG : ((MODE) == MODE_LDS && (VAR))) ? 64 : 256)
#ifndef RT_PRIMARY_INLINE_DEN
#define RT_PRIMARY_INLINE_DEN 2u /* the primary-inline loop serves the waves whose primary rays can reach at most 1/DEN of the spheres */
#endif
#ifndef RT_ROW_ROTATE
#define RT_ROW_ROTATE 3u
#endif
template <bool STATS, int MODE, bool LIGHTS, bool SPH_SCENE, bool VAR = false>
__global__ __launch_bounds__(RT_WG_THREADS(MODE, VAR), STATS ? 4 : (LIGHTS ? (MODE == MODE_BVH ? RT_BVH_LIGHT_WAVES : RT_SCAN_LIGHT_WAVES) : (MODE == MODE_BVH ? RT_BVH_WAVES : (SPH_SCENE ? RT_SCAN_SPH_WAVES : RT_SCAN_WAVES)))) void render_kernel(const KernelParams P) {
    constexpr bool POOL = MODE == MODE_BVH;
    constexpr bool LISTS = MODE == MODE_BVH && VAR;
    constexpr uint32_t WGT = RT_WG_THREADS(MODE, VAR);  // threads of this workgroup; a pixel group (32x8 pixels, or 256 work-list entries) is 256 / WGT workgroups
    constexpr uint32_t WGW = WGT / 64u;           // ... its waves
    // Position of this thread / workgroup in the launch's sequence of 256-pixel groups. Workgroups go to the 8 XCDs round-robin
    // by their index, and a row of a 4K / 8K frame is a multiple of 8 groups: with the plain row-major order every XCD would
    // own fixed COLUMNS of the picture (15 stripes at 4K) and the launch would end with whichever XCD got the expensive
    // ones. So each row of tiles starts RT_ROW_ROTATE columns further to the right than the one below it (wrapping around):
    // an XCD's tiles lie on diagonals. Which workgroup traces a tile has no effect on the image.
    uint32_t tid, bid;
    uint32_t ns = P.n_slices;  // slices of THIS workgroup's pixel group (a mixed launch traces some groups whole and some in slices)
    {
        constexpr uint32_t WPG = 256u / WGT;                                             // workgroups per pixel group (1, or 4 one-wave workgroups)
        const uint32_t units_x = ((P.width + RT_TILE_W - 1u) / RT_TILE_W) * WPG;        // workgroups per row of tiles
        const uint32_t per = P.n_slices * units_x;                                      // ... times the slices (the fast index)
        uint32_t b = blockIdx.x;
        if (P.tile_order != nullptr && P.mixed) {
            // cost-ordered mixed launch: [head_groups sliced] [whole groups] [the rest sliced], expensive first (rthip_api.cpp render_pass)
            const uint32_t head_units = P.head_groups * WPG * P.n_slices, whole_units = (P.whole_end - P.head_groups) * WPG;
            uint32_t g, w, sl = 0u;
            if (b >= head_units && b - head_units < whole_units) {
                const uint32_t u = b - head_units;
                g = u / WPG, w = u - g * WPG;
                if (WPG == 4u && u < (whole_units & ~31u)) g = ((u >> 5) << 3) | (u & 7u), w = (u >> 3) & 3u;  // (a group's tiles on one XCD, as below)
                g += P.head_groups;
                ns = 1u;
            } else {
                const uint32_t v = b < head_units ? b : b - head_units - whole_units;
                const uint32_t u = v / P.n_slices;
                sl = v - u * P.n_slices;
                g = u / WPG, w = u - g * WPG;
                g += b < head_units ? 0u : P.whole_end;
            }
            b = (P.tile_order[g] * WPG + w) * ns + sl;
        } else if (P.tile_order != nullptr) {
            // cost-ordered launch (tile_cost_kernel / tile_order_kernel below): expensive pixel groups first, sky last
            const uint32_t u = b / P.n_slices;
            uint32_t g = u / WPG, w = u - g * WPG;
            // (one-wave workgroups, as below: the four tiles of a pixel group 8 positions apart = on one XCD, where the launch allows it)
            if (WPG == 4u && P.n_slices == 1u && u < (gridDim.x & ~31u)) g = ((u >> 5) << 3) | (u & 7u), w = (u >> 3) & 3u;
            b = (P.tile_order[g] * WPG + w) * P.n_slices + (b - u * P.n_slices);
        } else if (!P.worklist) {
            const uint32_t row = b / per, rest = b - row * per;
            uint32_t u = rest / P.n_slices;
            const uint32_t sl = rest - u * P.n_slices;
            // One-wave workgroups: inside every run of 32 consecutive (unsliced) workgroups the four tiles of a pixel group take
            // indices 8 apart, i.e. the same XCD: a row of a pixel group is a whole 128-byte line of RGBA8 / 384 bytes of
            // running sum, and lines written in part by several XCDs' L2s are read back from HBM to be merged.
            if (WPG == 4u && P.n_slices == 1u && (units_x & 31u) == 0u) u = (u & ~31u) | ((u & 7u) << 2) | ((u >> 3) & 3u);
            if (RT_ROW_ROTATE != 0u) u = (u + RT_ROW_ROTATE * WPG * row) % units_x;  // (whole pixel groups)
            b = row * per + u * P.n_slices + sl;
        }
        // b = (index of the wave tile / pixel group in row-major order) * ns + slice
        const uint32_t wt = b / ns, sl = b - wt * ns;
        tid = WPG == 1u ? threadIdx.x : ((wt & 3u) << 6) | threadIdx.x;
        bid = WPG == 1u ? b : (wt >> 2) * ns + sl;
    }
    extern __shared__ __align__(16) unsigned char s_dyn[];
    // DEFCLASS: the radiance of paths that end at a texture after at least one bounce is summed apart from the other samples of
    // its group (g_k = grp_k + def_k, both in sample order) — in EVERY kernel variant, so that they all give the same bits;
    // QUEUE: the variant that actually defers those texture evaluations to a per-wave queue (below, "deferred texture queue").
    constexpr bool DEFCLASS = !LIGHTS;
    constexpr bool QUEUE = RT_TEX_QUEUE != 0 && !STATS && MODE == MODE_LDS && !LIGHTS;
    const uint32_t n_padded = (P.n_spheres + RT_SCAN_UNROLL - 1u) & ~(RT_SCAN_UNROLL - 1u);
    // [QUEUE: per wave, the queue, RT_QUEUE_WAVE_DWORDS] [geometry, padded sphere count] [shading records] [primary table]
    rt_f4 *s_geom_mem = (rt_f4 *)(s_dyn + (QUEUE ? WGW * RT_QUEUE_WAVE_DWORDS * 4u : 0u));
    rt_f4 *s_shade_mem = s_geom_mem + (MODE == MODE_LDS ? n_padded : 0u);
    uint32_t *s_stack = (uint32_t *)(s_shade_mem + (MODE == MODE_LDS ? RT_SHADE_RECS * P.n_spheres : 0u));
    // MODE_BVH, per wave: the RNG key of each of its pixels, then the image row of each of its tile's 8 rows (72 dwords) — or, for the
    // scattered pixels of a work-list pass, x | y << 16 of each pixel (128 dwords)
    const uint32_t pix_dwords = P.worklist ? 64u : 8u;
    uint32_t *s_pix_all = s_stack + (MODE == MODE_BVH ? P.bvh_levels * WGT : 0u);
    float *s_L_all = (float *)(s_pix_all + (MODE == MODE_BVH ? pix_dwords * WGW : 0u));
    // MODE_BVH: the staged top of the tree sits behind the ring and its row counters (rt_launch_render sizes it)
    lds_u4 *s_top = (lds_u4 *)((uint32_t *)(s_L_all + WGW * 3u * (P.ring_rows << 6)) + 16u * WGW);
    const uint32_t n_top = (RT_BVH_TOP_NODES > 0 && MODE == MODE_BVH) ? P.bvh_top_nodes : 0u;
    if (RT_BVH_TOP_NODES > 0 && MODE == MODE_BVH && n_top) {
        const lds_u4 *src = (const lds_u4 *)P.bvh_nodes;
        for (uint32_t i = threadIdx.x; i < 2u * n_top; i += WGT) s_top[i] = src[i];
        __syncthreads();
    }
    const rt_f4 *s_geom = s_geom_mem;
    if (MODE == MODE_LDS) {
        for (uint32_t i = threadIdx.x; i < n_padded; i += WGT)
            // padding: never hit. The centre is NaN, not just r*r = -inf: a padding index can equal `self` when the ray starts on a
            // plane or triangle (their shape ids follow the spheres'), which replaces c by 0 — a NaN half_b keeps the discriminant NaN
            s_geom_mem[i] = i < P.n_spheres ? P.sph_geom[i] : rt_f4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), -__builtin_inff()};
        for (uint32_t i = threadIdx.x; i < RT_SHADE_RECS * P.n_spheres; i += WGT) s_shade_mem[i] = P.sph_shade[i];
        if (RT_PRIMARY_INLINE && threadIdx.x < RT_PRIMARY_SPHERES && threadIdx.x < P.n_spheres) {
            // what a sphere test needs of a ray's origin, for the camera (every primary ray starts there): o - c and |o - c|^2 - r^2,
            // with the very expressions of closest_hit() — the primary-inline loop reads them instead of recomputing them per ray
            const rt_f4 g = P.sph_geom[threadIdx.x];
            const f3 oc = mk3(P.ox - g.x, P.oy - g.y, P.oz - g.z);
            s_shade_mem[RT_SHADE_RECS * P.n_spheres + threadIdx.x] = rt_f4{oc.x, oc.y, oc.z, fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -g.w)))};
        }
        __syncthreads();
    }
    const uint32_t wave = tid >> 6, lane = threadIdx.x & 63u;  // wave: which of the pixel group's four 8x8 tiles
    const uint32_t lwave = threadIdx.x >> 6;                   // ... and which wave of this workgroup (its share of the LDS)
    // Sample slicing (strong scaling): a workgroup traces samples [slice*slice_spp, ...) of its pixels, so that a workgroup
    // lives for at most slice_spp samples however large spp is — with one workgroup per tile for all 256 spp, the last
    // workgroups of a 1/8-frame launch ran for a third of the kernel on a mostly idle GPU (73 % kernel efficiency at 8 GPUs).
    // Whether a launch is sliced is a scheduling decision (rt_launch_render: only when the grid is too small to fill the
    // GPU otherwise); the IMAGE does not depend on it, because the unsliced kernel adds the samples in the same groups:
    // sum = (...((g0 + g1) + g2)...), g_k = the sequential sum of samples [k*slice_spp, (k+1)*slice_spp).
    const bool sliced = ns > 1;
    // the slice is the FAST index of the grid: the slices of a tile are scheduled together, so that the launch ends with the
    // last (cheap, sky) tiles of the image rather than with the expensive tiles of its last slice
    const uint32_t slice = sliced ? bid % ns : 0u;
    const uint32_t group = sliced ? bid / ns : bid;  // tile (or block of the work-list)
    const uint32_t s_first = slice * P.slice_spp;
    const uint32_t n_spp = sliced ? min(P.slice_spp, P.spp - s_first) : P.spp;
    uint32_t x, lrow;
    bool listed = true;
    if (P.worklist) {
        // RTHIP_F_ADAPTIVE, pass >= 2: thread i traces the i-th pixel that is not yet Final (compact_kernel)
        const uint32_t gi = group * 256u + tid;
        listed = gi < *P.worklist_count;
        const uint32_t oi = listed ? P.worklist[gi] : 0u;
        lrow = oi / P.width;
        x = oi - lrow * P.width;
    } else {
        // a workgroup = RT_TILE_W x RT_TILE_H pixels = its 4 waves' 8x8 tiles side by side (32 x 8): a row of the workgroup is
        // 128 B of RGBA8 and 384 B of the running sum = whole cache lines that no other workgroup (hence no other XCD's L2)
        // writes, so the lines leave L2 fully written — with 16 x 16 workgroups a third of the running-sum lines were shared
        // between two XCDs and every launch read 64 MB back from HBM to complete them
        const uint32_t tiles_x = (P.width + RT_TILE_W - 1u) / RT_TILE_W;
        const uint32_t tile_x = group % tiles_x, tile_y = group / tiles_x;
        x = tile_x * RT_TILE_W + ((wave % (RT_TILE_W / 8u)) << 3) + (lane & 7u);
        lrow = tile_y * RT_TILE_H + ((wave / (RT_TILE_W / 8u)) << 3) + (lane >> 3);
    }
    // owned local row -> image row y (row bands interleaved over contexts)
    uint32_t y = lrow;
    if (P.band_rows) {
        uint32_t k = lrow / P.band_rows;
        y = (P.band_first + k * P.band_stride) * P.band_rows + (lrow - k * P.band_rows);
    }
    const bool active = listed && x < P.width && lrow < P.owned_rows && y < P.height;
    const unsigned long long amask = __ballot(active);  // pixels of this wave's tile that exist

    uint32_t cnt[RT_STAT_N];
#pragma unroll
    for (int i = 0; i < RT_STAT_N; i++) cnt[i] = 0;
#ifdef RT_CLASS_CYCLES
    uint32_t dbg_shard = 255u;  // debug: counters filed per reachability mask of the wave (255 = sky-only / other loops)
#endif

#ifdef RT_TIMELINE  // debug build (tools/timeline.py): when and where each wave ran, written over the first pixels of its tile
    const unsigned long long tl_start = wall_clock64();
#endif
    const unsigned long long t_begin = P.tile_cost_out != nullptr ? wall_clock64() : 0ull;
    if (amask != 0ull) {
        uint32_t *s_pix = s_pix_all + lwave * pix_dwords;
        // MODE_BVH: this wave tile's primary-visibility list (rt_prepass.h), if the launch made them and the tile's fits
        const RtTilePre *pre_tile = nullptr;
        const RtTileEnt *pre_list = nullptr;
        if (POOL && LISTS && P.tile_pre != nullptr && !P.worklist) {
            pre_tile = P.tile_pre + (size_t(group) * 4u + wave);
            if (uint32_t(__builtin_amdgcn_readfirstlane(int(pre_tile->ent[0].ref))) != RT_PRE_OVERFLOW) pre_list = pre_tile->ent;
        }
        const uint32_t plane = P.ring_rows << 6;  // floats per colour plane of the wave's radiance slots
        float *s_L = s_L_all + lwave * 3u * plane;
        uint32_t *s_rows = (uint32_t *)(s_L_all + WGW * 3u * plane) + (lwave << 4);  // MODE_BVH: per ring slot, the paths that have ended
        // ... and which of the slot's samples ended at a texture after a bounce (DEFCLASS): behind the row counters (and the staged tree top)
        unsigned long long *s_rowdef = (unsigned long long *)((uint32_t *)(s_L_all + WGW * 3u * plane) + 16u * WGW + 8u * n_top) + (lwave << 4);
        rt_f4 pp_own;
        {
            // per pixel: direction through its lower-left corner d00 + H*x/(W-1) + V*y/(H-1) (camera.rs:46-48,
            // lib.rs:282-285) and its RNG key
            const float pu = float(x) * P.inv_wm1, pv = float(y) * P.inv_hm1;
            rt_f4 pp;
            pp.x = fmaf(P.vx, pv, fmaf(P.hx, pu, P.d00x));
            pp.y = fmaf(P.vy, pv, fmaf(P.hy, pu, P.d00y));
            pp.z = fmaf(P.vz, pv, fmaf(P.hz, pu, P.d00z));
            pp.w = __uint_as_float(rt_pixel_key(rt_seed_key(P.seed), y * P.width + x));
            if (POOL) {
                if (P.worklist) s_pix[lane] = x | (y << 16);
                else if ((lane & 7u) == 0u) s_pix[lane >> 3] = y;
            }
            pp_own = pp;
        }
        const uint32_t x_tile0 = uint32_t(__builtin_amdgcn_readfirstlane(int(x - (lane & 7u))));  // first column of the wave's tile
        const f3 cam_o = mk3(P.ox, P.oy, P.oz);
        f3 cam_v = cam_o;  // ... in vector registers: operand of the moves in next_ray
        asm volatile("" : "+v"(cam_v.x), "+v"(cam_v.y), "+v"(cam_v.z));
        // the jitter vectors in vector registers too: an FMA with a scalar-register operand issues at 4.1 cycles, with vector
        // operands only at 2.4 (tools/ubench/valu_rates.hip), and every fresh sample does six of them
        // (not in the BVH kernel: it is short of registers, and a path there is hundreds of instructions per jitter)
        f3 hw_v = mk3(P.hwx, P.hwy, P.hwz), vh_v = mk3(P.vhx, P.vhy, P.vhz);
        if (!POOL) asm volatile("" : "+v"(hw_v.x), "+v"(hw_v.y), "+v"(hw_v.z), "+v"(vh_v.x), "+v"(vh_v.y), "+v"(vh_v.z));
        f3 acc = mk3(0.0f, 0.0f, 0.0f);
        // ---- state of the path this lane is tracing ----
        uint32_t item = 0;   // radiance slot of the work item: ring row << 6 | pixel
        uint32_t key = 0;    // RNG counter of the slot in use (advances by GOLDEN per segment)
        uint32_t bounce = 0;
        uint32_t self = RT_NONE;  // shape the current ray starts on
        Ray r;
        r.o = cam_o, r.d = mk3(0.0f, 0.0f, -1.0f), r.tmin = r.tmax = 0.0f;
        f3 T = mk3(1.0f, 1.0f, 1.0f);
        f3 Ld = mk3(0.0f, 0.0f, 0.0f);  // extension: direct light gathered along the path (point-light mode)
        // Ray::new with an un-normalised direction: keep the unit vector, fold |d| into the window
        auto set_dir = [&](f3 nd) {
            const float l2 = dot(nd, nd);
            const float rl = fast_rsq(l2);
            r.d = nd * rl;
            const float len = l2 * rl;
            r.tmin = RT_T_MIN * len;
            r.tmax = RT_T_MAX * len;
        };
        HitRec h;
        h.t = 0.0f, h.shape = RT_NONE, h.b1 = h.b2 = 0.0f;
        BvhState bs;           // MODE_BVH: the closest-hit query is resumable
        BvhDeep deep;
        bs.ref = RT_BVH_DONE, bs.sp = 0, bs.inv_d = bs.noi = mk3(0.0f, 0.0f, 0.0f);
        bool tracing = false;  // ... and this lane's ray is still inside the tree
        // what shade() leaves behind for next_ray(): the hit point (IN r.o: the ray's origin is dead once its hit is known, and the
        // next ray starts there — no copy at the loop's join), the surface and the mirror direction
        f3 &p = r.o;
        f3 refl = mk3(0.0f, 0.0f, 0.0f);
        float cosi = 0.0f;
        Surface sf;
        sf.la = mk3(0.0f, 0.0f, 0.0f), sf.shin = 0.0f;
        sf.n = sf.tm = mk3(0.0f, 0.0f, 0.0f), sf.param = sf.r1 = sf.inv_ior = sf.tu = sf.tv = sf.tw = 0.0f, sf.kind = sf.mat = 0;
        bool dead = false;  // the last scatter absorbed the path (metal.rs:62-64): it ends, with L = T, at the next shade()
        // First half of a segment: closest hit and shading. Returns 1 = the path ended with radiance L,
        // 0 = it goes on (next_ray(false) scatters it), 2 (MODE_BVH) = the ray is parked inside the tree.
        f3 grp = mk3(0.0f, 0.0f, 0.0f);  // sum of the current group of slice_spp samples
        // a finished path hands its radiance over where it ends (not through a merged return value: that costs copies)
        auto emit = [&](f3 L) {
            if (LIGHTS) L = L + Ld;
            if constexpr (POOL) {
                s_L[item] = L.x;  // lib.rs:287, summed in sample order when the row is retired
                s_L[plane + item] = L.y;
                s_L[2u * plane + item] = L.z;
            } else {
                // no fused multiply-add of the caller's T*colour into the sum: the work-pool variant stores the rounded
                // product first, and "BVH == linear scan bit for bit" (tests) compares the two variants
                asm volatile("" : "+v"(L.x), "+v"(L.y), "+v"(L.z));
                grp = grp + L;  // lib.rs:287
            }
        };
        // ... and the same for a path that ended at a texture after at least one bounce (DEFCLASS): those samples are summed apart
        // from the other samples of their group, g = grp + dgrp, so that the variant of this kernel that evaluates such hits later,
        // 64 at a time (QUEUE), adds the very same numbers in the very same order as the variants that evaluate them at once
        f3 dgrp = mk3(0.0f, 0.0f, 0.0f);
        auto emit_def = [&](f3 L) {
            if constexpr (POOL) {
                s_L[item] = L.x;
                s_L[plane + item] = L.y;
                s_L[2u * plane + item] = L.z;
                atomicOr(&s_rowdef[item >> 6], 1ull << (item & 63u));  // before the row's counter moves: LDS operations of a wave stay in order
            } else {
                asm volatile("" : "+v"(L.x), "+v"(L.y), "+v"(L.z));
                dgrp = dgrp + L;
            }
        };
        // close a group of slice_spp samples (lane = pixel loops without a queue)
        auto fold = [&]() {
            if (DEFCLASS) {
                acc = acc + (grp + dgrp);
                dgrp = mk3(0.0f, 0.0f, 0.0f);
            } else {
                acc = acc + grp;
            }
            grp = mk3(0.0f, 0.0f, 0.0f);
        };
        // Texture::scatter — texture.rs:45-57 — for a sphere hit given (a, b, c) = (p-c).z, (p-c).x, (p-c).y/r (sphere.rs:67-73), else (U, V)
        auto texel_color = [&](float a, float b, float c, bool uv_given, float scale, uint32_t tex) -> f3 {
            float U = a, V = b;
            if (!uv_given) {
                U = (1.0f + atan2_finite(a, b) * 0.31830988618379067f) * 0.5f;  // sphere.rs:70
                V = acos_over_pi(c);                                             // sphere.rs:71
            }
            const DevTexInfo ti = P.tex_info[tex];
            const int32_t icx = texture_cast(U, scale, ti.width_f), icy = texture_cast(V, scale, ti.height_f);
            uint32_t txl, tyl;
            if (__ballot((icx | icy) < 0) == 0ull) {  // the usual case (sphere coordinates in [0, 1], positive scale): no sign to carry
                txl = texture_rem(uint32_t(icx), ti.width, ti.magic_w);
                tyl = texture_rem(uint32_t(icy), ti.height, ti.magic_h);
            } else {
                txl = texture_wrap_signed(icx, ti.width, ti.magic_w);
                tyl = texture_wrap_signed(icy, ti.height, ti.magic_h);
            }
            const uint32_t texel = P.texels[ti.first_texel + (((tyl >> 3) * ti.tiles_per_row + (txl >> 3)) << 6) + ((tyl & 7u) << 3) + (txl & 7u)];
            return mk3(float(texel & 255u), float((texel >> 8) & 255u), float((texel >> 16) & 255u)) * (1.0f / 255.0f);
        };
        // MODE_BVH + point lights: the direct term of the surface shade() found, one light at a time. Finds the next light that
        // can contribute, starts its shadow ray (the lane is then "parked": 2) or, after the last light, closes the term (0).
        uint32_t lit = 0;  // index + 1 of the light whose shadow ray this lane is tracing, 0 = none
        f3 lsum = mk3(0.0f, 0.0f, 0.0f);
        float lsc = 0.0f;
        auto next_light = [&]() -> int {
            for (; lit <= P.n_lights; lit++) {
                const DevMaterial m = P.materials[sf.mat];
                const DevLight lt = P.lights[lit - 1u];
                const f3 to_light = mk3(lt.px - p.x, lt.py - p.y, lt.pz - p.z);
                const float dist2 = dot(to_light, to_light);
                const float rdist = fast_rsq(dist2);
                const f3 ldir = to_light * rdist;
                float w;
                if (sf.kind == RTHIP_MAT_LAMBERTIAN)
                    w = fmaxf(0.0f, dot(sf.n, ldir));
                else
                    w = (m.shininess > 0.0f) ? __powf(fmaxf(0.0f, dot(refl, ldir)), m.shininess) : 0.0f;
                if (!(w > 0.0f)) continue;
                cnt[RT_STAT_RAYS]++;
                if (STATS) cnt[RT_STAT_SHADOW]++;
                // the shadow ray replaces the lane's ray: next_ray() builds the scattered ray from p, the surface and refl alone
                // (lit surfaces are not dielectric)
                r.d = ldir;  // (origin: p is r.o already)
                const float dist = dist2 * rdist;
                r.tmin = RT_T_MIN * dist, r.tmax = dist;
                lsc = w * fast_rcp(dist2);
                bool occ = false;
                float t;
                for (uint32_t i = 0; i < P.n_planes; i++) occ |= hit_plane(P.pln_geom[4 * i], r, t);
                if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
                if (occ) continue;
                bvh_begin(r, bs);
                tracing = true;
                return 2;
            }
            lit = 0u;
            Ld = Ld + T * lsum;
            T = T * sf.tm;  // scatter (as at the end of shade())
            return 0;
        };
        // Second part of shade(): the surface of the hit in `h` (or the sky), once the scene query has returned.
        // (sph: a compile-time flag — the scene holds spheres only)
        // prim: the ray is a primary ray, so T = 1 and the origin is the camera: x * 1.0f = x exactly, so T is left out)
        // bounced: the caller KNOWS that the ray is not a primary ray (bounce != 0)
        // Returns 1 = the path ended (radiance handed over), 0 = it goes on, 3 (QUEUE) = it ended at a texture after a bounce and
        // the caller has to queue the evaluation (sf.tu/tv/tw, sf.param, sf.r1 and T describe it).
        auto surface = [&](auto sph, auto prim, auto bounced) -> int {
            constexpr bool SPH = decltype(sph)::value;
            constexpr bool PRIM = decltype(prim)::value;
            constexpr bool BOUNCED = decltype(bounced)::value;
            bool done;
            if (h.shape == RT_NONE) {  // ray.rs:85, background_color ray.rs:89-93 (the direction is already unit)
                // t = 0.5(d.y + 1); (1-t)(1,1,1) + t(0.5,0.7,1) = (1 - 0.5t, 1 - 0.3t, 1) = (0.75 - 0.25 d.y, 0.85 - 0.15 d.y, 1)
                if (PRIM) emit(mk3(fmaf(-0.25f, r.d.y, 0.75f), fmaf(-0.15f, r.d.y, 0.85f), 1.0f));
                else emit(mk3(T.x * fmaf(-0.25f, r.d.y, 0.75f), T.y * fmaf(-0.15f, r.d.y, 0.85f), T.z));
                done = true;
                if (STATS) cnt[RT_STAT_SKY]++;
            } else {
                r.o = fma3(r.d, h.t, PRIM ? cam_v : r.o);  // ray.rs:30-32: the hit point p (an alias of r.o, see above)
                // ---- the surface: normal + material (collision.rs:29-43) ----
                if (SPH || h.shape < P.n_spheres) {
                    rt_f4 r0, r1, r2, rec_light = rt_f4{0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (MODE == MODE_LDS) {  // kept apart so that the LDS reads stay ds_read_b128 (no generic pointer)
                        const rt_f4 *rec = (const rt_f4 *)((const unsigned char *)s_shade_mem + (h.shape << 6));  // 32-bit address math, RT_SHADE_RECS * 16 B
                        r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        if (LIGHTS && !POOL) rec_light = rec[3];
                        // pin the loaded values: otherwise the loads are sunk to their uses and merged with the
                        // material loads of the non-sphere branch into per-dword flat loads
                        asm volatile("" : "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w), "+v"(r2.x), "+v"(r2.y), "+v"(r2.z), "+v"(r2.w));
                    } else {
                        const rt_f4 *rec = P.sph_shade + RT_SHADE_RECS * h.shape;
                        r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        if (LIGHTS && !POOL) rec_light = rec[3];
                    }
                    const f3 pc = mk3(p.x - r0.x, p.y - r0.y, p.z - r0.z);
                    sf.n = pc * r0.w;  // (p - c)/r
                    sf.tm = mk3(r1.x, r1.y, r1.z);
                    sf.param = r1.w;
                    sf.kind = __float_as_uint(r2.x);
                    sf.r1 = r2.y;
                    sf.inv_ior = r2.z;
                    sf.mat = __float_as_uint(r2.w);
                    sf.tu = pc.z, sf.tv = pc.x, sf.tw = pc.y * r0.w;
                    if (LIGHTS && !POOL) {  // the fourth record: read where the others are (LDS), not from the material table
                        const rt_f4 r3 = rec_light;
                        sf.la = mk3(r3.x, r3.y, r3.z), sf.shin = r3.w;
                    }
                } else {
                    if (h.shape < P.n_spheres + P.n_planes) {
                        const uint32_t i = h.shape - P.n_spheres;
                        const rt_f4 g0 = P.pln_geom[4 * i];
                        sf.n = mk3(g0.x, g0.y, g0.z);
                        sf.mat = __float_as_uint(P.pln_geom[4 * i + 1].w);
                    } else {
                        const uint32_t i = h.shape - P.n_spheres - P.n_planes;
                        const rt_f4 e1 = P.tri_geom[3 * i + 1], e2 = P.tri_geom[3 * i + 2];
                        sf.n = normalize(cross(mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z)));
                        sf.mat = __float_as_uint(P.tri_geom[3 * i].w);
                    }
                    const DevMaterial m = P.materials[sf.mat];
                    sf.tm = mk3(m.tr, m.tg, m.tb);
                    sf.param = m.param, sf.kind = m.kind, sf.r1 = m.r1, sf.inv_ior = m.inv_ior;
                    if (LIGHTS && !POOL) sf.la = mk3(m.ar, m.ag, m.ab), sf.shin = m.shininess;
                    // (sf.tu/tv/tw keep whatever an earlier sphere hit left: only sphere textures read them)
                }
                if (STATS) {
#pragma unroll
                    for (uint32_t kk = 0; kk < 4; kk++) cnt[RT_STAT_HIT0 + kk] += (sf.kind == kk) ? 1u : 0u;
                }
                if (sf.kind == RTHIP_MAT_TEXTURE) {
                    // Texture::scatter — texture.rs:45-57; bounce -> None (texture.rs:59-61) => path ends, ray.rs:80
                    RT_ARM_COUNT(2, 3)
                    const bool defcls = DEFCLASS && !PRIM && (BOUNCED || bounce != 0u);
                    if (!(SPH || h.shape < P.n_spheres)) {
                        // (U, V) of a plane / triangle hit go where a sphere hit keeps (p-c).z, (p-c).x
                        if (h.shape < P.n_spheres + P.n_planes) {
                            const uint32_t i = h.shape - P.n_spheres;
                            const rt_f4 g1 = P.pln_geom[4 * i + 1], g2 = P.pln_geom[4 * i + 2], g3 = P.pln_geom[4 * i + 3];
                            const f3 q = mk3(p.x - g1.x, p.y - g1.y, p.z - g1.z);
                            sf.tu = dot(q, mk3(g2.x, g2.y, g2.z));
                            sf.tv = dot(q, mk3(g3.x, g3.y, g3.z));
                        } else {
                            sf.tu = h.b1, sf.tv = h.b2;
                        }
                    }
                    if constexpr (QUEUE && !PRIM) {
                        if (defcls) return 3;  // evaluated later, 64 at a time (the deferred texture queue)
                    }
                    const f3 c = texel_color(sf.tu, sf.tv, sf.tw, !(SPH || h.shape < P.n_spheres), sf.param, __float_as_uint(sf.r1));  // a texture material's r1 slot holds its texture index
                    if (defcls) emit_def(T * c);
                    else emit(PRIM ? c : T * c);
                    done = true;
                } else {
                    if (LIGHTS) {  // the mirror direction feeds the Phong term; without lights next_ray() computes it, and only for Metal / Dielectric
                        cosi = dot(r.d, sf.n);
                        refl = fma3(sf.n, -2.0f * cosi, r.d);  // metal.rs:27-29, dielectric.rs:28-30 (unit incoming direction)
                    }
                    if (LIGHTS && POOL && sf.kind != RTHIP_MAT_DIELECTRIC) {
                        // the work-pool variant traces its shadow rays like any other ray of the pool (resumable, together with the
                        // closest-hit rays of the other lanes); same terms in the same order as the loop below
                        lsum = mk3(0.0f, 0.0f, 0.0f);
                        lit = 1u;
                        return next_light();
                    }
                    if (LIGHTS && !POOL && sf.kind != RTHIP_MAT_DIELECTRIC) {
                        // extension (RTHIP_F_POINT_LIGHTS): Lambert N.L on Lambertian, Phong (R.L)^s on Metal, one shadow ray per light
                        f3 sum = mk3(0.0f, 0.0f, 0.0f);
                        for (uint32_t li = 0; li < P.n_lights; li++) {
                            const DevLight lt = P.lights[li];
                            const f3 to_light = mk3(lt.px - p.x, lt.py - p.y, lt.pz - p.z);
                            const float dist2 = dot(to_light, to_light);
                            const float rdist = fast_rsq(dist2);
                            const f3 ldir = to_light * rdist;
                            float w;
                            if (sf.kind == RTHIP_MAT_LAMBERTIAN)
                                w = fmaxf(0.0f, dot(sf.n, ldir));
                            else
                                w = (sf.shin > 0.0f) ? __powf(fmaxf(0.0f, dot(refl, ldir)), sf.shin) : 0.0f;
                            if (w > 0.0f) {
                                cnt[RT_STAT_RAYS]++;
                                if (STATS) cnt[RT_STAT_SHADOW]++;
                                Ray sr;
                                sr.o = p, sr.d = ldir;
                                const float dist = dist2 * rdist;
                                sr.tmin = RT_T_MIN * dist, sr.tmax = dist;
                                if (!occluded<STATS, MODE>(P, sr, h.shape, cnt, s_stack, s_geom)) {
                                    const float sc = w * fast_rcp(dist2);
                                    sum = sum + mk3(sf.la.x * lt.ir, sf.la.y * lt.ig, sf.la.z * lt.ib) * sc;
                                }
                            }
                        }
                        Ld = Ld + T * sum;
                    }
                    T = PRIM ? sf.tm : T * sf.tm;  // scatter: albedo/pi * 3 (lambertian_diffuse.rs:53-66, metal.rs:45-56, dielectric.rs:56-66)
                    done = false;
                }
            }
            return done ? 1 : 0;
        };
        auto shade = [&](auto sph, auto bounced) -> int {
            constexpr bool SPH = decltype(sph)::value;
            if (dead) {  // None -> new_color alone, ray.rs:80
                dead = false;
                emit(T);
                return 1;
            }
            if (MODE == MODE_BVH) {
                if (!tracing) {
                    h.shape = RT_NONE;
                    if (bounce != P.max_depth) {  // ray.rs:70-72: depth exhausted -> sky without a scene query
                        h = closest_hit_unbounded<STATS>(P, r, cnt);
                        bvh_begin(r, bs);
                        tracing = true;
                        cnt[RT_STAT_RAYS]++;
                    }
                }
                if (tracing) {
                    bool blocked;
                    if (LIGHTS) blocked = bvh_run<STATS, 2, true, int(WGT)>(P, r, lit ? h.shape : self, h, bs, deep, cnt, s_stack, lit != 0u, s_top, n_top);
                    else blocked = bvh_run<STATS, 0, true, int(WGT)>(P, r, self, h, bs, deep, cnt, s_stack, false, s_top, n_top);
                    if (bs.ref != RT_BVH_DONE) return 2;  // parked: the finished lanes of the wave shade and fetch new work
                    tracing = false;
                    if (LIGHTS && lit) {  // the shadow ray of light lit - 1 has left the tree
                        if (!blocked) {
                            const DevMaterial m = P.materials[sf.mat];
                            const DevLight lt = P.lights[lit - 1u];
                            lsum = lsum + mk3(m.ar * lt.ir, m.ag * lt.ig, m.ab * lt.ib) * lsc;
                        }
                        lit++;
                        return next_light();
                    }
                }
            } else {
                h.shape = RT_NONE;
                if (bounce != P.max_depth) {  // ray.rs:70-72: depth exhausted -> sky without a scene query
                    h = closest_hit<STATS, MODE, SPH>(P, r, self, cnt, s_stack, s_geom);
                    cnt[RT_STAT_RAYS]++;
                }
            }
            return surface(sph, std::integral_constant<bool, false>{}, bounced);
        };
        // Second half: the lane's next ray — the scattered ray of the surface shade() left behind, or
        // (fresh) the primary ray of a new sample. Both need one RNG pair and one normalisation; sharing
        // them halves that cost in a wave where some lanes continue and the others start a new path.
        uint32_t key_own = rt_counter(__float_as_uint(pp_own.w), P.sample0 + s_first, 0);  // !POOL: slot 0 of this pixel's next sample
        constexpr std::integral_constant<bool, SPH_SCENE> scene_sph{};
        auto next_ray = [&](bool fresh, uint32_t pixel, uint32_t sample) {
            rt_f4 pp = pp_own;
            uint32_t k = key + RT_RNG_GOLDEN;  // RNG slot 1 + bounce
            if (POOL) {
                if (fresh) {
                    // the pixel's coordinates (its row's y from the tile's 8-entry table, x from the tile's first column; a work-list
                    // pass keeps x | y << 16 per pixel); its RNG key and base direction are recomputed: 20 instructions per path are
                    // nothing next to a traversal, and LDS is what limits the resident waves (rt_launch_render)
                    uint32_t fx = x_tile0 + (pixel & 7u), fy;
                    if (P.worklist) {
                        const uint32_t xy = s_pix[pixel];
                        fx = xy & 0xffffu, fy = xy >> 16;
                    } else {
                        fy = s_pix[pixel >> 3];
                    }
                    const uint2 px = make_uint2(rt_pixel_key(rt_seed_key(P.seed), fy * P.width + fx), 0u);
                    const float pu = float(fx) * P.inv_wm1, pv = float(fy) * P.inv_hm1;
                    pp.x = fmaf(P.vx, pv, fmaf(P.hx, pu, P.d00x));
                    pp.y = fmaf(P.vy, pv, fmaf(P.hy, pu, P.d00y));
                    pp.z = fmaf(P.vz, pv, fmaf(P.hz, pu, P.d00z));
                    k = rt_counter(px.x, sample, 0);
                }
            } else {
                k = fresh ? key_own : k;
                key_own += fresh ? RT_RNG_SLOTS * RT_RNG_GOLDEN : 0u;
            }
            key = k;
            const uint32_t hb = rt_mix32(k);
            const float ua = rt_unit(hb), ub = rt_unit(rt_second(hb));
            f3 nd;
            if (fresh) {
                // lib.rs:282-285 (x jitter first): u = (x + xi)/(W-1) = x/(W-1) + xi/(W-1), the per-pixel part is precomputed
                // (H/(W-1) and V/(H-1) are precomputed: P.hw*, P.vh*)
                nd = mk3(fmaf(vh_v.x, ub, fmaf(hw_v.x, ua, pp.x)), fmaf(vh_v.y, ub, fmaf(hw_v.y, ua, pp.y)), fmaf(vh_v.z, ub, fmaf(hw_v.z, ua, pp.z)));
                if (LIGHTS) Ld = mk3(0.0f, 0.0f, 0.0f);
                // the rest of the path state, reset IN PLACE ("+v": same register in and out, so the join needs no copies): plain
                // moves under the branch's lane mask cost half of what selects on the mask do (tools/ubench: 2.3 vs 4.2 cycles)
                asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.x) : "v"(cam_v.x));  // camera.rs:43-50
                asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.y) : "v"(cam_v.y));
                asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.z) : "v"(cam_v.z));
                asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.x));
                asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.y));
                asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.z));
                asm volatile("v_mov_b32 %0, 0" : "+v"(bounce));
                asm volatile("v_mov_b32 %0, -1" : "+v"(self));  // RT_NONE
            } else {
                // random_unit_vector — lambertian_diffuse.rs:39-45, metal.rs:31-37 (a first, then z); v_sin/v_cos take revolutions
                const float z = fmaf(2.0f, ub, -1.0f);
                const float rr = fast_sqrt(fmaf(-z, z, 1.0f));
                const f3 w = mk3(rr * __builtin_amdgcn_cosf(ua), rr * __builtin_amdgcn_sinf(ua), z);
                // One shared unit vector, then a BRANCH on the material: most waves (the ground) hold Lambertian hits only and skip the
                // mirror direction, the absorption test and the dielectric block altogether.
                if (sf.kind == RTHIP_MAT_LAMBERTIAN) {
                    // Lambertian::bounce always returns a ray (lambertian_diffuse.rs:68-73), also when rounding puts n + w a hair
                    // below the surface
                    nd = sf.n + w;
                } else {
                    RT_ARM_COUNT(7, 8)
                    if (!LIGHTS) {
                        cosi = dot(r.d, sf.n);
                        refl = fma3(sf.n, -2.0f * cosi, r.d);  // metal.rs:27-29, dielectric.rs:28-30 (unit incoming direction)
                    }
                    nd = fma3(w, sf.param, refl);  // Metal: reflect + fuzz*w (metal.rs:59-60)
                    // metal.rs:62-64: absorbed if the new direction points into the surface
                    dead = dot(nd, sf.n) < 0.0f;
                    if (sf.kind == RTHIP_MAT_DIELECTRIC) {  // dielectric.rs:68-98
                        RT_ARM_COUNT(4, 6)
                        // dot_product = direction . normal with the UN-normalised direction; len() is the element count 3: dielectric.rs:74-83
                        const float len = r.tmin * (1.0f / RT_T_MIN);  // |direction|
                        const float dp = cosi * len;
                        const bool inside = dp > 0.0f;
                        const float eta = inside ? sf.param : sf.inv_ior;
                        const float cosine = (inside ? sf.param * dp : -dp) * (1.0f / 3.0f);
                        // outward_normal = inside ? -n : n, so unit . outward_normal = -|cosi|
                        const float dt = -fabsf(cosi);
                        const float kk = fmaf(-eta * eta, fmaf(-dt, dt, 1.0f), 1.0f);  // dielectric.rs:39
                        const float om = 1.0f - cosine;
                        const float om2 = om * om;
                        const float prob = (kk < 0.0f) ? 1.0f : fmaf(1.0f - sf.r1, om2 * om2 * om, sf.r1);  // dielectric.rs:47-52,87-92
                        // refracted = (unit - on*dt)*eta - on*sqrt(k)   (dielectric.rs:44); the coin is ua (dielectric.rs:94)
                        const float sgn = inside ? -1.0f : 1.0f;
                        const float cn = sgn * (-dt * eta - fast_sqrt(kk));
                        const f3 refr = fma3(sf.n, cn, r.d * eta);
                        nd = (prob > ua) ? refl : refr;
                        dead = false;
                    }
                }
                asm volatile("v_add_u32 %0, 1, %0" : "+v"(bounce));
                asm volatile("v_mov_b32 %0, %1" : "+v"(self) : "v"(h.shape));  // the ray starts on the shape it hit (the hit point is in r.o already)
            }
            // (The path state is updated by the asm moves inside the two branches above. Written as C++ assignments there — or as
            // C++ selects on the branch condition, which the compiler turns back into that — the updates become phi copies on both
            // sides of the join: ~25 moves per iteration.)
            set_dir(nd);
        };
        if constexpr (!POOL) {
            // ---- empty tiles: no ray of this wave's pixels can reach any shape => every path is "primary ray -> sky" -------
            // Conservative test per lane and sphere, on the bundle of rays through the pixel's jitter square: every such ray d
            // lies within the angle theta of the ray a through the square's centre, so if it hits sphere (c, r) at distance
            // s <= |q| + r (q = c - o), the centre is within r + s*theta of the half-line o + s*a. theta is bounded from the
            // square's corners with a factor pi/2 to spare, the right-hand side carries a relative margin 1e-4 (FP32: ~1e-6).
            // The fast loop below evaluates exactly the expressions of next_ray(fresh) + shade() for a miss: same bits.
            bool tile_empty = false;
            // ... and per sphere: bit i of pmask = some primary ray of this wave can reach sphere i (wave-uniform; spheres 0..63)
            unsigned long long pmask = ~0ull;
            bool primary_inline = false;
            // (point lights change neither: a primary ray that reaches no shape gathers no light, and the cone test is about primary rays)
            // A few planes and triangles do not stand in the way of the primary-inline loop (it tests them all on every primary ray,
            // after the reachable spheres); the sky-only loop needs a scene of spheres.
            if ((!STATS || P.count_executed) && MODE == MODE_LDS && P.max_depth != 0 && n_spp >= RT_FAST_MIN_SPP && (SPH_SCENE || (P.n_planes <= RT_PRIMARY_TRIS && P.n_tris <= RT_PRIMARY_TRIS))) {
                const f3 hw = mk3(P.hwx, P.hwy, P.hwz), vh = mk3(P.vhx, P.vhy, P.vhz);
                const f3 e00 = mk3(pp_own.x, pp_own.y, pp_own.z), e10 = e00 + hw, e01 = e00 + vh, e11 = e10 + vh;
                const f3 ec = e00 + (hw + vh) * 0.5f;
                const float lc = fast_sqrt(dot(ec, ec));
                const f3 a = ec * fast_rcp(lc);
                const f3 d0 = e00 - ec, d1 = e10 - ec, d2 = e01 - ec, d3 = e11 - ec;
                const float dev = fast_sqrt(fmaxf(fmaxf(dot(d0, d0), dot(d1, d1)), fmaxf(dot(d2, d2), dot(d3, d3))));
                const float lmin = fast_sqrt(fminf(fminf(dot(e00, e00), dot(e10, e10)), fminf(dot(e01, e01), dot(e11, e11))));
                const float theta = 1.6f * dev * fast_rcp(lmin);  // >= the angle between a and any ray of the bundle (sin x >= 2x/pi)
                const bool degenerate = !(theta < 0.5f);          // a degenerate camera (NaN, huge pixels): no fast path
                bool can_hit = degenerate;
                unsigned long long pm = 0ull;
                for (uint32_t i = 0; i < P.n_spheres; i++) {
                    const rt_f4 g = s_geom[i];
                    const f3 q = mk3(g.x - cam_o.x, g.y - cam_o.y, g.z - cam_o.z);
                    const float rad = fast_sqrt(g.w), lq = fast_sqrt(dot(q, q));
                    const f3 qa = cross(q, a);
                    const float dist = (dot(q, a) >= 0.0f) ? fast_sqrt(dot(qa, qa)) : lq;
                    const float reach = fmaf(lq + rad, theta, rad) * 1.0001f + 1e-30f;
                    const bool ch = degenerate | !(dist > reach);  // NaN => can hit
                    can_hit |= ch;
                    if (RT_PRIMARY_INLINE && i < 64u && __ballot(active && ch) != 0ull) pm |= 1ull << i;
                }
                tile_empty = SPH_SCENE && __ballot(active && can_hit) == 0ull;
                pmask = pm;
                // worth it where paths are short: the wave's primary rays see few of the spheres (the ground, mostly)
                primary_inline = RT_PRIMARY_INLINE && !tile_empty && P.n_spheres <= 64u && RT_PRIMARY_INLINE_DEN * uint32_t(__popcll(pm)) <= P.n_spheres;
            }
#ifdef RT_CLASS_CYCLES  // debug build: wall-clock ticks (100 MHz) and waves per tile class, in the hit / test counters
            const unsigned long long t_class0 = wall_clock64();
            const uint32_t tile_class = tile_empty ? 0u : (primary_inline ? 1u : 2u);
            dbg_shard = tile_empty ? 255u : (uint32_t(pmask) & 15u) | (primary_inline ? 0u : 16u);
#endif
            if (tile_empty) {
                if (active) {
                    const uint32_t g_mask = P.slice_spp - 1u;
                    uint32_t k = rt_counter(__float_as_uint(pp_own.w), P.sample0 + s_first, 0);
                    for (uint32_t s = 1; s <= n_spp; s++) {
                        const uint32_t hb = rt_mix32(k);
                        k += RT_RNG_SLOTS * RT_RNG_GOLDEN;
                        const float ua = rt_unit(hb), ub = rt_unit(rt_second(hb));  // lib.rs:282-285
                        const f3 nd = mk3(fmaf(P.vhx, ub, fmaf(P.hwx, ua, pp_own.x)), fmaf(P.vhy, ub, fmaf(P.hwy, ua, pp_own.y)), fmaf(P.vhz, ub, fmaf(P.hwz, ua, pp_own.z)));
                        const float dy = nd.y * fast_rsq(dot(nd, nd));
                        f3 L = mk3(fmaf(-0.25f, dy, 0.75f), fmaf(-0.15f, dy, 0.85f), 1.0f);  // background_color, ray.rs:89-93, as in shade()
                        asm volatile("" : "+v"(L.x), "+v"(L.y), "+v"(L.z));
                        grp = grp + L;
                        if ((s & g_mask) == 0u && s != n_spp) {
                            acc = acc + grp;
                            grp = mk3(0.0f, 0.0f, 0.0f);
                        }
                    }
                    acc = acc + grp;
                    cnt[RT_STAT_PATHS] += n_spp;
                    cnt[RT_STAT_RAYS] += n_spp;
                    if (STATS) cnt[RT_STAT_SKY] += n_spp;  // (RTHIP_F_COUNT_EXECUTED: no shape test is executed here)
                }
            } else {
            // ---- deferred texture queue (QUEUE) -------------------------------------------------------------------------------
            // A bounce ray that ends at a texture (terminal: texture.rs:59-61) needs the 105-instruction Texture::scatter arm, and
            // on a ground tile a handful of lanes want it in every other iteration: the arm ran at 8 % lane utilisation and was a
            // quarter of the tile's instructions. Here such a lane only QUEUES the evaluation — 9 dwords in LDS: T, the sphere
            // coordinates (or U, V), scale, texture, the parity of its sample group — and starts its next sample; when the
            // queue would overflow the wave runs the arm ONCE for its (up to) 64 entries, every lane on another entry, and
            // each lane then takes its own results, in the order it queued them (= the order of its samples), into the deferred
            // sum of the group they belong to. g_k = grp_k + def_k as everywhere (DEFCLASS): same numbers, same order, same bits
            // as the variants that evaluate at once. Two groups can be in flight per lane (entries of the group it has just left
            // may still wait): def[2]; a group that is left while entries of it wait is parked in `closed` and folded at the
            // next flush, and a flush is forced before entries get 48 iterations old, i.e. long before that lane can leave its
            // next group (a group is >= 64 samples, a lane ends at most one sample per iteration).
            uint32_t q_count = 0, q_age = 0;  // wave-uniform: entries waiting; iterations since the queue was last empty
            uint32_t qstate = 0;              // per lane: entries waiting of group parity 0 | parity 1 << 8 | a closed group waits << 16 | its parity << 17
            uint32_t *q_ent = (uint32_t *)s_dyn + lwave * RT_QUEUE_WAVE_DWORDS;  // [9][64]
            unsigned long long *q_mask = (unsigned long long *)(q_ent + 576);      // [64]: the entries of lane l
            float *q_def = (float *)(q_ent + 704);                                  // [2][3][64]
            float *q_closed = q_def + 384;                                          // [3][64]
            const uint32_t g_mask = P.slice_spp - 1u;  // slice_spp is a power of two
            const uint32_t g_shift = 31u - uint32_t(__builtin_clz(P.slice_spp));
            if constexpr (QUEUE) {
                q_mask[lane] = 0ull;
#pragma unroll
                for (uint32_t i = 0; i < 6u; i++) q_def[i * 64u + lane] = 0.0f;
            }
            // runs the arm for every queued entry and hands the results to their lanes; called by all lanes of the loop together
            // (the stages are kept apart — "memory" — so that an entry's fields are read when they are needed, not all up front:
            // the whole path state of the lanes is live here, and the arm's temporaries are what would spill)
            auto q_flush = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const unsigned long long ex = __ballot(true);
                const uint32_t step = uint32_t(__popcll(ex));
                for (uint32_t e = wave_prefix(ex); e < q_count; e += step) {
                    f3 c = texel_color(__uint_as_float(q_ent[192u + e]), __uint_as_float(q_ent[256u + e]), __uint_as_float(q_ent[320u + e]),
                                       !SPH_SCENE && (q_ent[512u + e] & 2u) != 0u, __uint_as_float(q_ent[384u + e]), q_ent[448u + e]);
                    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z) : : "memory");
                    const f3 L = mk3(__uint_as_float(q_ent[e]), __uint_as_float(q_ent[64u + e]), __uint_as_float(q_ent[128u + e])) * c;
                    q_ent[e] = __float_as_uint(L.x), q_ent[64u + e] = __float_as_uint(L.y), q_ent[128u + e] = __float_as_uint(L.z);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // every lane adds its own results to the deferred sum of their group, in queue order (LDS operations of a lane
                // execute in program order; nobody else touches the lane's sums)
                for (unsigned long long m = q_mask[lane]; m != 0ull; m &= m - 1ull) {
                    const uint32_t e = uint32_t(__builtin_ctzll(m));
                    float *d = q_def + (q_ent[512u + e] & 1u) * 192u + lane;
                    atomicAdd(d, __uint_as_float(q_ent[e]));
                    atomicAdd(d + 64, __uint_as_float(q_ent[64u + e]));
                    atomicAdd(d + 128, __uint_as_float(q_ent[128u + e]));
                }
                q_mask[lane] = 0ull;
                if (qstate & 0x10000u) {  // the group this lane left while entries of it waited: now it is complete
                    const uint32_t o = ((qstate >> 17) & 1u) * 192u + lane;
                    const f3 cg = mk3(q_closed[lane], q_closed[64u + lane], q_closed[128u + lane]);
                    const f3 d = mk3(q_def[o], q_def[64u + o], q_def[128u + o]);
                    acc = acc + (cg + d);
                    q_def[o] = 0.0f, q_def[64u + o] = 0.0f, q_def[128u + o] = 0.0f;
                }
                qstate = 0u;
                q_count = 0u, q_age = 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            // Once per loop iteration, all lanes together. push: this lane's path has just ended at a texture (surface() returned 3);
            // par: the parity of that sample's group; drain: every lane has traced its last sample — empty the queue.
            // Returns true when the loop may end (drain, and nothing waits).
            auto q_step = [&](bool push, uint32_t par, bool drain) -> bool {
                const unsigned long long pm = __ballot(push);
                const uint32_t n = uint32_t(__popcll(pm));
                if (q_count != 0u) q_age++;
                // ONE flush site (the arm is a hundred instructions: one copy per loop): before the push when the new entries would not fit
                const bool full = q_count + n > 64u;
                if (q_count != 0u && (full || q_age >= 48u || (drain && n == 0u))) q_flush();
                if (pm != 0ull) {
                    if (push) {
                        const uint32_t e = q_count + wave_prefix(pm);
                        const bool uv = !SPH_SCENE && !(h.shape < P.n_spheres);
                        q_ent[e] = __float_as_uint(T.x), q_ent[64u + e] = __float_as_uint(T.y), q_ent[128u + e] = __float_as_uint(T.z);
                        q_ent[192u + e] = __float_as_uint(sf.tu), q_ent[256u + e] = __float_as_uint(sf.tv), q_ent[320u + e] = __float_as_uint(sf.tw);
                        q_ent[384u + e] = __float_as_uint(sf.param), q_ent[448u + e] = __float_as_uint(sf.r1);
                        q_ent[512u + e] = par | (uv ? 2u : 0u);
                        q_mask[lane] |= 1ull << e;
                        qstate += par ? 0x100u : 1u;
                    }
                    q_count += n;
                }
                return drain && q_count == 0u;
            };
            // a lane has finished sample number s_done (1-based) of its n_spp: group boundary?
            auto q_account = [&](uint32_t s_done) {
                if ((s_done & g_mask) == 0u && s_done != n_spp) {
                    const uint32_t par = ((s_done - 1u) >> g_shift) & 1u;
                    const uint32_t o = par * 192u + lane;
                    if (((qstate >> (par * 8u)) & 0xffu) == 0u) {
                        const f3 d = mk3(q_def[o], q_def[64u + o], q_def[128u + o]);
                        acc = acc + (grp + d);
                        q_def[o] = 0.0f, q_def[64u + o] = 0.0f, q_def[128u + o] = 0.0f;
                    } else {
                        q_closed[lane] = grp.x, q_closed[64u + lane] = grp.y, q_closed[128u + lane] = grp.z;
                        qstate |= 0x10000u | (par << 17);
                    }
                    grp = mk3(0.0f, 0.0f, 0.0f);
                }
            };
            // after the loop: whatever still waits, then the lane's last group
            auto q_finish = [&]() {
                const uint32_t o = (((n_spp - 1u) >> g_shift) & 1u) * 192u + lane;
                const f3 d = mk3(q_def[o], q_def[64u + o], q_def[128u + o]);
                acc = acc + (grp + d);
            };
            // ---- the primary ray of a fresh sample, inline (see below); returns true if the path ended there
            const __attribute__((address_space(3))) lds_f4 *prim_lds = (const __attribute__((address_space(3))) lds_f4 *)(s_shade_mem + RT_SHADE_RECS * P.n_spheres);
            auto primary_fresh = [&]() -> bool {
                // next_ray(true) without the state a primary ray does not need (origin = camera, T = 1, no shape to start on)
                key = key_own;
                key_own += RT_RNG_SLOTS * RT_RNG_GOLDEN;
                const uint32_t hb = rt_mix32(key);
                const float ua = rt_unit(hb), ub = rt_unit(rt_second(hb));
                set_dir(mk3(fmaf(vh_v.x, ub, fmaf(hw_v.x, ua, pp_own.x)), fmaf(vh_v.y, ub, fmaf(hw_v.y, ua, pp_own.y)),
                            fmaf(vh_v.z, ub, fmaf(hw_v.z, ua, pp_own.z))));
                asm volatile("v_mov_b32 %0, 0" : "+v"(bounce));
                if (LIGHTS) {  // the direct-light code of surface() reads T and the origin: no shortcut for them
                    Ld = mk3(0.0f, 0.0f, 0.0f);
                    asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.x) : "v"(cam_v.x));
                    asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.y) : "v"(cam_v.y));
                    asm volatile("v_mov_b32 %0, %1" : "+v"(r.o.z) : "v"(cam_v.z));
                    asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.x));
                    asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.y));
                    asm volatile("v_mov_b32 %0, 1.0" : "+v"(T.z));
                }
                float nbest = -__uint_as_float(__float_as_uint(r.tmax) - 1u);
                const float ntmin = -r.tmin;
                h.shape = RT_NONE;
                for (unsigned long long m = pmask; m != 0ull; m &= m - 1ull) {
                    const uint32_t i = uint32_t(__builtin_ctzll(m));  // wave-uniform: broadcast read
                    const lds_f4 pv = prim_lds[i];                    // {camera - centre, |camera - centre|^2 - r^2}
                    const float half_b = dot(mk3(pv.x, pv.y, pv.z), r.d);
                    const float disc = fmaf(half_b, half_b, -pv.w);
                    const float root = fast_sqrt(disc);
                    const float n0 = half_b + root, n1 = half_b - root;
                    const float n = (n0 < ntmin) ? n0 : n1;
                    const bool take = (n1 < ntmin) & !(n < nbest);
                    nbest = take ? n : nbest;
                    h.shape = take ? i : h.shape;
                }
                h.t = -nbest;
                if (!SPH_SCENE) {  // ... then the planes and triangles, as closest_hit() goes on over them (the ray starts at the camera)
                    Ray pr = r;
                    pr.o = cam_v;
                    for (uint32_t i = 0; i < P.n_planes; i++) {
                        float t;
                        bool ok = hit_plane(P.pln_geom[4 * i], pr, t);
                        RT_TAKE(ok, t, P.n_spheres + i, (void)0)
                    }
                    tris_closest(P, pr, h);
                }
                cnt[RT_STAT_RAYS]++;
                if (STATS) {  // RTHIP_F_COUNT_EXECUTED: the tests this loop actually makes (the reachable spheres only)
                    cnt[RT_STAT_SPHERE] += uint32_t(__popcll(pmask));
                    if (!SPH_SCENE) cnt[RT_STAT_PLANE] += P.n_planes, cnt[RT_STAT_TRI] += P.n_tris;
                }
                return surface(scene_sph, std::integral_constant<bool, !LIGHTS>{}, std::integral_constant<bool, false>{}) != 0;
            };
            constexpr std::integral_constant<bool, true> is_bounced{};
            constexpr std::integral_constant<bool, false> maybe_primary{};
            if (RT_PRIMARY_INLINE && primary_inline) {
                // ---- lane = pixel, primary ray inline: a fresh sample traces its primary ray at once, against the spheres of pmask
                // only, and joins the other lanes' scattered rays in the same iteration. Where most paths are "ground, then sky"
                // a path then costs one full scan instead of two. The skipped spheres cannot be hit (the cone test above), the
                // others are tested in index order with the expressions of closest_hit(): same hit, same bits.
                if (active) {
                    uint32_t s_done = 0;
                    bool fresh = true;
                    if constexpr (QUEUE) {
                        // (a lane that has traced its last sample stays in the loop, idle, until every lane has: the queue is flushed by
                        // all lanes together, and the last entries are evaluated after the loop)
                        bool finished = false;
                        for (;;) {
                            int st = 0;
                            if (!finished) {
                                bool ended = false;
                                if (fresh) ended = primary_fresh();
                                st = 1;
                                if (!ended) {
                                    next_ray(false, 0, 0);
                                    st = shade(scene_sph, is_bounced);
                                }
                            }
                            const uint32_t par = (s_done >> g_shift) & 1u;
                            if (!finished) {
                                fresh = st != 0;
                                if (fresh) {
                                    s_done++;
                                    finished = s_done == n_spp;
                                }
                            }
                            if (q_step(st == 3, par, __ballot(!finished) == 0ull)) break;
                            if (st != 0) q_account(s_done);
                        }
                        q_finish();
                    } else {
                        for (;;) {
                            bool ended = false;
#ifdef RT_CLASS_CYCLES
                            {
                                const unsigned long long ex = __ballot(true);
                                if (__builtin_amdgcn_mbcnt_hi(uint32_t(ex >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ex), 0u)) == 0) cnt[RT_STAT_NODE] += 64;
                                cnt[RT_STAT_SKY]++;
                                cnt[RT_STAT_SHADOW] += fresh ? 1u : 0u;
                            }
#endif
                            if (fresh) {
                                ended = primary_fresh();
#ifdef RT_CLASS_CYCLES
                                cnt[RT_STAT_HIT0 + 3] += ended ? 1u : 0u;
#endif
                            }
                            if (!ended) {
                                next_ray(false, 0, 0);
                                fresh = shade(scene_sph, is_bounced) != 0;
                            } else {
                                fresh = true;
                            }
                            if (fresh) {
                                s_done++;
                                if (s_done == n_spp) break;
                                if ((s_done & g_mask) == 0u) fold();
                            }
                        }
                        fold();
                    }
                    cnt[RT_STAT_PATHS] += s_done;
                }
            } else
            // ---- lane = pixel: path regeneration over the pixel's samples ----
            if (active) {
                uint32_t s_done = 0;
                next_ray(true, lane, 0);
                if constexpr (QUEUE) {
                    bool finished = false;
                    for (;;) {
                        int st = 0;
                        if (!finished) st = shade(scene_sph, maybe_primary);
                        const uint32_t par = (s_done >> g_shift) & 1u;
                        const bool was_finished = finished;
                        if (st != 0) {
                            s_done++;
                            finished = s_done == n_spp;
                        }
                        if (q_step(st == 3, par, __ballot(!finished) == 0ull)) break;
                        if (st != 0) q_account(s_done);
                        if (!finished) next_ray(st != 0, lane, 0);
                        (void)was_finished;
                    }
                    q_finish();
                } else {
                    for (;;) {
#ifdef RT_BVH_DEBUG_UTIL
                        if (STATS) {  // wave iterations x 64 (vs `rays` = lane iterations): the lanes a wave idles at its end
                            const unsigned long long ex = __ballot(true);
                            if (__builtin_amdgcn_mbcnt_hi(uint32_t(ex >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(ex), 0u)) == 0) cnt[RT_STAT_SHADOW] += 64;
                        }
#endif
                        const bool fresh = shade(scene_sph, maybe_primary) != 0;
                        if (fresh) {
                            s_done++;
                            if (s_done == n_spp) break;
                            if ((s_done & g_mask) == 0u) fold();
                        }
                        next_ray(fresh, lane, 0);
                    }
                    fold();
                }
                cnt[RT_STAT_PATHS] += s_done;
            }
            }
#ifdef RT_CLASS_CYCLES
#ifndef RT_ARM_COUNTS
            if (lane == 0u) {
                cnt[RT_STAT_HIT0 + tile_class] += uint32_t(wall_clock64() - t_class0);
                cnt[RT_STAT_SPHERE + tile_class] += 1u;
            }
#endif
#endif
        } else {
            // Work items in hand-out order: item i = (pixel i & 63, sample i >> 6). The radiance of sample row r waits in ring
            // slot r & (rows - 1); the oldest row is retired (added to the pixels' sums, in sample order) as soon as its last
            // path has ended, and items are handed out up to `rows` rows ahead of it — the wave idles lanes for the slowest
            // paths only at the very end of its samples, not once per group of rows.
            const uint32_t rows = P.ring_rows, total = n_spp << 6, expect = uint32_t(__popcll(amask));
            if (lane < 16u) s_rows[lane] = 0u, s_rowdef[lane] = 0ull;  // paths that have ended, per ring slot
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t next = 0;      // first item not yet handed out (wave-uniform)
            uint32_t oldest = 0;    // first sample row not yet retired (wave-uniform)
            bool have = false;      // this lane is tracing a path
#ifdef RT_PRE_DEBUG
            const unsigned long long t_pool0 = wall_clock64();
#endif
            // Primary ray through the tile's primary-visibility list (rt_prepass.h), for the lanes that have just taken an item:
            // they test the leaves of their pixel's mask in order of distance — together, coherently, with no node visit and
            // no stack — and shade the hit, so that what joins the traversing lanes below is their BOUNCE ray. Once the next
            // leaf's box lies beyond the best hit the rest of the (sorted) list cannot matter; a hit that is not nearer than
            // the pixel's cut distance (leaves that did not fit the tile's list) is not final: that ray goes on as a normal
            // traversal, keeping its best hit (3). Otherwise returns what shade() returns.
            auto primary_listed = [&]() -> int {
                h = closest_hit_unbounded<STATS>(P, r, cnt);
                cnt[RT_STAT_RAYS]++;
                const uint32_t pix = item & 63u;
                const unsigned long long m0 = pre_tile->mask[pix];
                const float cut = pre_tile->cut[pix];
                uint32_t mlo = uint32_t(m0), mhi = uint32_t(m0 >> 32);
                for (;;) {
                    uint32_t leaf = RT_BVH_DONE;
                    if ((mlo | mhi) != 0u) {
                        const uint32_t bit = mlo ? uint32_t(__builtin_ctz(mlo)) : 32u + uint32_t(__builtin_ctz(mhi));
                        if (mlo) mlo &= mlo - 1u;
                        else mhi &= mhi - 1u;
                        const RtTileEnt e = pre_list[bit];
                        if (e.near > h.t) mlo = mhi = 0u;
                        else leaf = e.ref;
                    }
                    if (__ballot(leaf != RT_BVH_DONE) == 0ull) break;
                    if (leaf != RT_BVH_DONE) (void)leaf_test<STATS, 0>(P, leaf, r, RT_NONE, h, false, cnt);
                }
                if (!(h.t < cut)) {
                    bvh_begin(r, bs);
                    tracing = true;
                    return 3;
                }
                return surface(scene_sph, std::integral_constant<bool, false>{}, std::integral_constant<bool, false>{});
            };
            for (;;) {
                // ---- retire the rows whose paths have all ended ----
                while (oldest < n_spp && uint32_t(__builtin_amdgcn_readfirstlane(int(s_rows[oldest & (rows - 1u)]))) == expect) {
                    const uint32_t k = ((oldest & (rows - 1u)) << 6) + lane;
                    const unsigned long long defrow = DEFCLASS ? s_rowdef[oldest & (rows - 1u)] : 0ull;
                    if (active) {
                        const f3 L = mk3(s_L[k], s_L[plane + k], s_L[2u * plane + k]);
                        if (DEFCLASS && ((defrow >> lane) & 1ull)) dgrp = dgrp + L;
                        else grp = grp + L;
                    }
                    if (lane == 0u) s_rows[oldest & (rows - 1u)] = 0u, s_rowdef[oldest & (rows - 1u)] = 0ull;
                    oldest++;
                    if (oldest % P.slice_spp == 0u || oldest == n_spp) fold();
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (oldest == n_spp) break;
                // ---- idle lanes take the next items, in lane order ----
                const uint32_t limit = min(total, (oldest + rows) << 6);
                const unsigned long long need = __ballot(!have);
                bool took = false;
                if (!have) {
                    const uint32_t idx = next + __builtin_amdgcn_mbcnt_hi(uint32_t(need >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(need), 0u));
                    if (idx < limit && ((amask >> (idx & 63u)) & 1ull)) {
                        item = (((idx >> 6) & (rows - 1u)) << 6) | (idx & 63u);
                        have = true;
                        took = true;
                        next_ray(true, idx & 63u, P.sample0 + s_first + (idx >> 6));
                    }
                }
                next = min(next + uint32_t(__popcll(need)), limit);
                if (LISTS && pre_list != nullptr && P.max_depth != 0u) {
#ifdef RT_PRE_DEBUG
                    const unsigned long long t_p0 = wall_clock64();
#endif
                    if (took) {
                        const int st = primary_listed();
                        if (st == 1) {
                            cnt[RT_STAT_PATHS]++;
                            have = false;
                            atomicAdd(&s_rows[item >> 6], 1u);
                        } else if (st == 0) {
                            next_ray(false, 0, 0);
                        }
                    }
#ifdef RT_PRE_DEBUG
                    if (lane == 0u) cnt[RT_STAT_PIX_TRACED] += uint32_t(wall_clock64() - t_p0);
#endif
                    // keep the traversal below full: while many lanes are without a path (their primary ray reached the sky) and
                    // items are left to hand out, hand out again first
                    if (uint32_t(__popcll(__ballot(!have))) >= RT_POOL_REFILL_LANES && next < limit) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        continue;
                    }
                }
                if (have) {
                    const int st = shade(scene_sph, std::integral_constant<bool, false>{});
                    if (st == 1) {
                        cnt[RT_STAT_PATHS]++;
                        have = false;
                        atomicAdd(&s_rows[item >> 6], 1u);  // after the radiance stores of emit(): LDS operations of a wave stay in order
                    } else if (st == 0) {
                        next_ray(false, 0, 0);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
#ifdef RT_PRE_DEBUG
            if (lane == 0u) cnt[RT_STAT_PIX_FINAL] += uint32_t(wall_clock64() - t_pool0);
#endif
        }
        if (active) {
            const size_t oi = size_t(lrow) * P.width + x;
            if (sliced) {
                // sample slicing: this workgroup traced one slice of the pixel's samples; resolve_kernel adds the slices in order
                float *part = P.partial + (size_t(slice) * P.owned_rows * P.width + oi) * 3;
                part[0] = acc.x, part[1] = acc.y, part[2] = acc.z;
            } else {
                finish_pixel(P, oi, acc, cnt[RT_STAT_PIX_TRACED], cnt[RT_STAT_PIX_FINAL]);
            }
#ifdef RT_TIMELINE
            // (a sliced launch: slice k writes over row k of the tile, and finish_pixel() leaves the frame alone in this build)
            // (pixels 6 and 7 of the row: a marker | the slices of this workgroup's group, and | its slice — the rows of a mixed launch's whole groups keep older data)
            if (P.rgba8 && lane - 8u * (sliced ? slice & 7u : 0u) < 8u) {
                const unsigned long long tl_end = wall_clock64();
                const uint32_t v[8] = {uint32_t(tl_start), uint32_t(tl_start >> 32), uint32_t(tl_end), uint32_t(tl_end >> 32),
                                       uint32_t(__builtin_amdgcn_s_getreg(63492)), uint32_t(__builtin_amdgcn_s_getreg(63508)), 0xC0FFEE00u | ns, 0xC0FFEE00u | slice};
                P.rgba8[oi] = v[lane & 7u];
            }
#endif
        }
    }

    // feedback for the next launch's order (tile_order_kernel): how long this wave ran; a pixel group's cost is its LONGEST wave (a
    // workgroup holds its four wave slots until the slowest wave ends; with the sum as the key, a sky group with one tile on a sphere's
    // silhouette sorted among the cheap groups and its 260 us wave ended the launch alone, tools/timeline_band.py)
    if (P.tile_cost_out != nullptr && amask != 0ull && lane == 0u) atomicMax(&P.tile_cost_out[group], uint32_t(wall_clock64() - t_begin));

    // per-wave reduction of the counters, one atomic per wave and counter into a shard (skipped when nobody reads them)
    if (!P.stats) return;
#pragma unroll
    for (int i = 0; i < RT_STAT_N; i++) {
#ifndef RT_CLASS_CYCLES
        if (!STATS && i > RT_STAT_RAYS && i < RT_STAT_PIX_TRACED) continue;
#endif
        uint32_t v = cnt[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
#ifdef RT_CLASS_CYCLES
        if (lane == 0 && v) atomicAdd(&P.stats[dbg_shard * RT_STAT_N + i], (unsigned long long)v);
#else
        if (lane == 0 && v) atomicAdd(&P.stats[((bid * 4u + wave) & (RT_STAT_SHARDS - 1)) * RT_STAT_N + i], (unsigned long long)v);
#endif
    }
}

// Sample slicing, second half: pixel sums = slice 0 + slice 1 + ... in slice order (a fixed order: the image does not depend
// on how many GPUs share the frame), then the end of generate_pixel. Thread i handles owned pixel i (or the i-th listed pixel).
__global__ __launch_bounds__(256) void resolve_kernel(const KernelParams P) {
    const uint32_t gi = blockIdx.x * 256u + threadIdx.x;
    const size_t npx = size_t(P.owned_rows) * P.width;
    bool in = gi < npx;
    size_t oi = gi;
    if (P.mixed) {
        // a mixed launch: workgroup i finishes the i-th SLICED pixel group of the launch order (32 x 8 pixels)
        const uint32_t g = P.tile_order[blockIdx.x < P.head_groups ? blockIdx.x : P.whole_end + (blockIdx.x - P.head_groups)];
        const uint32_t tiles_x = (P.width + RT_TILE_W - 1u) / RT_TILE_W;
        const uint32_t x = (g % tiles_x) * RT_TILE_W + (threadIdx.x & (RT_TILE_W - 1u)), lrow = (g / tiles_x) * RT_TILE_H + threadIdx.x / RT_TILE_W;
        in = x < P.width && lrow < P.owned_rows;
        oi = size_t(lrow) * P.width + x;
    }
    if (P.worklist) {
        in = gi < *P.worklist_count;
        oi = in ? P.worklist[gi] : 0u;
    }
    uint32_t traced = 0, fin = 0;
    if (in) {
        f3 acc = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < P.n_slices; k++) {
            const float *part = P.partial + (size_t(k) * npx + oi) * 3;
            acc.x += part[0], acc.y += part[1], acc.z += part[2];
        }
        finish_pixel(P, oi, acc, traced, fin);
    }
    if (P.pix_state && P.stats) {
        const unsigned long long tm = __ballot(traced != 0), fm = __ballot(fin != 0);
        if ((threadIdx.x & 63u) == 0) {
            unsigned long long *sh = &P.stats[(blockIdx.x & (RT_STAT_SHARDS - 1)) * RT_STAT_N];
            if (tm) atomicAdd(&sh[RT_STAT_PIX_TRACED], (unsigned long long)__popcll(tm));
            if (fm) atomicAdd(&sh[RT_STAT_PIX_FINAL], (unsigned long long)__popcll(fm));
        }
    }
}

// RTHIP_F_ADAPTIVE, before every pass but the first: the work-list of the pixels that are not Final
// (generate_pixel returns at once for a Final pixel, lib.rs:277-279). Thread i looks at the i-th
// pixel in 8x8-tile order, so that 64 consecutive list entries stay a compact patch of the image
// (coherent primary rays); a workgroup of 4096 pixels appends its survivors with one atomic. The order of the list
// varies from run to run, the image does not: a pixel's samples depend on its index alone.
// Final pixels get their stored colour written to the image outputs, which therefore need not
// be the buffers of the previous pass.
#define RT_COMPACT_THREADS 1024
#define RT_COMPACT_ROUNDS 4
__global__ __launch_bounds__(RT_COMPACT_THREADS) void compact_kernel(const KernelParams P, uint32_t *list, uint32_t *count) {
    // One atomic per WORKGROUP of 4096 pixels (all waves of a frame bumping one counter serialise in L2: one atomic per wave
    // made this kernel 0.5-1.5 ms at 4K, ten times the render pass it prepares).
    __shared__ uint32_t s_wave_total[RT_COMPACT_THREADS / 64];
    __shared__ uint32_t s_block_base;
    const uint32_t tiles_x = (P.width + 7u) >> 3;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t oi[RT_COMPACT_ROUNDS], before[RT_COMPACT_ROUNDS];  // my pixel of each round; live pixels of my wave before mine
    bool live[RT_COMPACT_ROUNDS];
    uint32_t wave_total = 0, n_final = 0;
#pragma unroll
    for (int j = 0; j < RT_COMPACT_ROUNDS; j++) {
        const uint32_t i = (blockIdx.x * RT_COMPACT_ROUNDS + j) * RT_COMPACT_THREADS + threadIdx.x;
        const uint32_t tile = i >> 6, in_tile = i & 63u;
        const uint32_t x = (tile % tiles_x) * 8u + (in_tile & 7u), lrow = (tile / tiles_x) * 8u + (in_tile >> 3);
        const bool inside = x < P.width && lrow < P.owned_rows;
        oi[j] = lrow * P.width + x;
        const uint32_t st = inside ? P.pix_state[oi[j]] : RT_PIX_FINAL;
        live[j] = inside && !(st & RT_PIX_FINAL);
        const unsigned long long mask = __ballot(live[j]);
        before[j] = wave_total + __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
        wave_total += uint32_t(__popcll(mask));
        if (inside && !live[j]) {
            if (P.rgba8) P.rgba8[oi[j]] = (st & RT_PIX_RGB_MASK) | 0xff000000u;
            if (P.accum_out) {
                P.accum_out[size_t(oi[j]) * 3 + 0] = P.accum[size_t(oi[j]) * 3 + 0];
                P.accum_out[size_t(oi[j]) * 3 + 1] = P.accum[size_t(oi[j]) * 3 + 1];
                P.accum_out[size_t(oi[j]) * 3 + 2] = P.accum[size_t(oi[j]) * 3 + 2];
            }
        }
        n_final += uint32_t(__popcll(__ballot(inside && !live[j])));
    }
    if ((threadIdx.x & 63u) == 0) s_wave_total[wave] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < RT_COMPACT_THREADS / 64; k++) total += s_wave_total[k];
        s_block_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_block_base;
    for (uint32_t k = 0; k < wave; k++) base += s_wave_total[k];
#pragma unroll
    for (int j = 0; j < RT_COMPACT_ROUNDS; j++)
        if (live[j]) list[base + before[j]] = oi[j];
    if ((threadIdx.x & 63u) == 0 && n_final && P.stats)
        atomicAdd(&P.stats[((blockIdx.x * (RT_COMPACT_THREADS / 64) + wave) & (RT_STAT_SHARDS - 1)) * RT_STAT_N + RT_STAT_PIX_FINAL], (unsigned long long)n_final);
}

extern "C" hipError_t rt_launch_compact(const KernelParams *P, uint32_t *list, uint32_t *count, hipStream_t stream) {
    (void)hipGetLastError();  // hipGetLastError() below must report THIS launch, not an older failure of the thread
    const uint32_t tiles = ((P->width + 7u) >> 3) * ((P->owned_rows + 7u) >> 3);
    const uint32_t per_block = RT_COMPACT_THREADS * RT_COMPACT_ROUNDS / 64;  // tiles of 64 pixels per workgroup
    const uint32_t blocks = (tiles + per_block - 1) / per_block;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_kernel, dim3(blocks), dim3(RT_COMPACT_THREADS), 0, stream, *P, list, count);
    return hipGetLastError();
}

// ---- primary-visibility lists (MODE_BVH; rt_prepass.h) ---------------------------------------------------------------
// One wave per 8x8 wave tile of the render launch (same workgroup -> tile map as render_kernel, so record group*4 + wave
// belongs to that wave of that workgroup). The wave walks the tree breadth-first with the tile's pyramid — lane i tests the
// two children of the i-th node of the current level, survivors are appended with ballot + prefix count — collects the leaves
// it touches (at most RT_PRE_COLLECT), ranks them by (near, reference), keeps the RT_PRE_LEAVES nearest, and every lane
// (pixel) marks the kept leaves its own pyramid touches and the nearest dropped one. Host mirror: rt_prepass_host.cpp.
__global__ __launch_bounds__(256) void prepass_kernel(const KernelParams P, RtTilePre *__restrict__ out) {
    __shared__ uint32_t s_front[4][2][RT_PRE_FRONT];
    __shared__ uint32_t s_ref[4][RT_PRE_COLLECT];
    __shared__ float s_near[4][RT_PRE_COLLECT];
    __shared__ uint32_t s_rank[4][RT_PRE_COLLECT];
    __shared__ float s_box[4][RT_PRE_COLLECT][6];
    __shared__ RtTileEnt s_ent[4][RT_PRE_LEAVES];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t group = blockIdx.x;
    // the pixel of this lane, as in render_kernel
    const uint32_t tiles_x = (P.width + RT_TILE_W - 1u) / RT_TILE_W;
    const uint32_t tile_x = group % tiles_x, tile_y = group / tiles_x;
    const uint32_t x = tile_x * RT_TILE_W + ((wave % (RT_TILE_W / 8u)) << 3) + (lane & 7u);
    const uint32_t lrow = tile_y * RT_TILE_H + ((wave / (RT_TILE_W / 8u)) << 3) + (lane >> 3);
    uint32_t y = lrow;
    if (P.band_rows) {
        const uint32_t k = lrow / P.band_rows;
        y = (P.band_first + k * P.band_stride) * P.band_rows + (lrow - k * P.band_rows);
    }
    const bool active = x < P.width && lrow < P.owned_rows && y < P.height;
    RtTilePre *rec = out + (size_t(group) * 4u + wave);
    uint32_t *front[2] = {s_front[wave][0], s_front[wave][1]};
    uint32_t nleaf = 0;
    bool over = false;
    const unsigned long long amask = __ballot(active);
    if (amask != 0ull) {
        // the tile's pixel rectangle (rows need not be contiguous when a band is not a multiple of 8 rows: their hull)
        uint32_t xmin = active ? x : 0xffffffffu, xmax = active ? x : 0u, ymin = active ? y : 0xffffffffu, ymax = active ? y : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            xmin = min(xmin, uint32_t(__shfl_xor(int(xmin), off, 64))), xmax = max(xmax, uint32_t(__shfl_xor(int(xmax), off, 64)));
            ymin = min(ymin, uint32_t(__shfl_xor(int(ymin), off, 64))), ymax = max(ymax, uint32_t(__shfl_xor(int(ymax), off, 64)));
        }
        RtPyramid tile;
        rt_pre_pyramid(P, float(xmin) - RT_PRE_MARGIN, float(xmax + 1u) + RT_PRE_MARGIN, float(ymin) - RT_PRE_MARGIN, float(ymax + 1u) + RT_PRE_MARGIN, tile);
        uint32_t nfront = 1, cur = 0;
        if (lane == 0) front[0][0] = 0u;  // the root is an inner node
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (nfront != 0u && !over) {
            uint32_t nnext = 0;
            for (uint32_t base = 0; base < nfront && !over; base += 64u) {
                const bool have = base + lane < nfront;
                float lo[2][3], hi[2][3];
                uint32_t child[2] = {RT_BVH_DONE, RT_BVH_DONE};
                bool keep[2] = {false, false};
                if (have) {
                    const DevBvhNodeQ nd = P.bvh_nodes[front[cur][base + lane]];
                    bool empty[2];
                    rt_pre_decode(nd, lo, hi, empty);
                    child[0] = nd.ref0, child[1] = nd.ref1;
                    keep[0] = !empty[0] && !rt_pre_box_outside(tile, lo[0], hi[0]);
                    keep[1] = !empty[1] && !rt_pre_box_outside(tile, lo[1], hi[1]);
                }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const bool inner = keep[c] && !(child[c] & RT_BVH_LEAF_BIT), leaf = keep[c] && (child[c] & RT_BVH_LEAF_BIT);
                    const unsigned long long bi = __ballot(inner), bl = __ballot(leaf);
                    const uint32_t pi = nnext + wave_prefix(bi), pl = nleaf + wave_prefix(bl);
                    nnext += uint32_t(__popcll(bi)), nleaf += uint32_t(__popcll(bl));
                    if (nnext > RT_PRE_FRONT || nleaf > RT_PRE_COLLECT) {
                        over = true;  // (wave-uniform)
                        break;
                    }
                    if (inner) front[cur ^ 1u][pi] = child[c];
                    if (leaf) {
                        s_ref[wave][pl] = child[c];
                        s_near[wave][pl] = rt_pre_box_near(tile.o, lo[c], hi[c]);
#pragma unroll
                        for (int k = 0; k < 3; k++) s_box[wave][pl][k] = lo[c][k], s_box[wave][pl][3 + k] = hi[c][k];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cur ^= 1u;
            nfront = nnext;
        }
    }
    unsigned long long mask = 0ull;
    float cut = __builtin_inff();
    if (!over && nleaf != 0u) {
        // rank of every collected leaf by (near, reference): the position it would have in the sorted list
        for (uint32_t j = lane; j < nleaf; j += 64u) {
            const float nj = s_near[wave][j];
            const uint32_t rj = s_ref[wave][j];
            uint32_t rank = 0;
            for (uint32_t k = 0; k < nleaf; k++) {
                const float nk = s_near[wave][k];
                rank += (nk < nj || (nk == nj && s_ref[wave][k] < rj)) ? 1u : 0u;
            }
            s_rank[wave][j] = rank;
            if (rank < RT_PRE_LEAVES) s_ent[wave][rank] = RtTileEnt{rj, nj};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (active) {
            RtPyramid px;
            rt_pre_pyramid(P, float(x) - RT_PRE_MARGIN, float(x) + 1.0f + RT_PRE_MARGIN, float(y) - RT_PRE_MARGIN, float(y) + 1.0f + RT_PRE_MARGIN, px);
            for (uint32_t j = 0; j < nleaf; j++) {  // (wave-uniform addresses: broadcast reads)
                float lo[3], hi[3];
#pragma unroll
                for (int k = 0; k < 3; k++) lo[k] = s_box[wave][j][k], hi[k] = s_box[wave][j][3 + k];
                if (rt_pre_box_outside(px, lo, hi)) continue;
                const uint32_t rank = s_rank[wave][j];
                if (rank < RT_PRE_LEAVES) mask |= 1ull << rank;
                else cut = fminf(cut, s_near[wave][j]);
            }
        }
    }
    RtTileEnt e = RtTileEnt{RT_BVH_DONE, __builtin_inff()};
    if (!over && lane < nleaf) e = s_ent[wave][lane];
    if (over && lane == 0u) e.ref = RT_PRE_OVERFLOW;
    rec->ent[lane] = e;
    rec->mask[lane] = over ? 0ull : mask;
    rec->cut[lane] = over ? __builtin_inff() : cut;
}

extern "C" hipError_t rt_launch_prepass(const KernelParams *P, void *tile_pre, hipStream_t stream) {
    (void)hipGetLastError();
    const uint32_t tiles_x = (P->width + RT_TILE_W - 1u) / RT_TILE_W, tiles_y = (P->owned_rows + RT_TILE_H - 1u) / RT_TILE_H;
    const uint32_t blocks = tiles_x * tiles_y;
    if (blocks == 0 || P->n_bvh_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(prepass_kernel, dim3(blocks), dim3(256), 0, stream, *P, (RtTilePre *)tile_pre);
    return hipGetLastError();
}

// ---- cost-ordered launches of the scan kernels ---------------------------------------------------------
// The workgroups of a launch start in index order and the launch ends when the last one does; a pixel group's cost spans 37x
// (sky only: 44 us per 256 samples; ground + the glass sphere: 1.6 ms), and a launch that holds only a few workgroups per slot
// (1/8 of a frame per GPU) ended with a tail of whatever expensive groups happened to start late. So: estimate every group's
// cost from what its primary rays can reach — one cone through the group's 32x8 pixels against every sphere, weighted by what
// a hit on that material goes on to cost — sort the groups by it (counting sort over 16 classes, one workgroup), and hand the
// render kernel the order: longest first, so that the launch ends with the cheap ones. A heuristic for the SCHEDULE only: the
// image does not depend on which workgroup traces which group.
__global__ __launch_bounds__(256) void tile_cost_kernel(const KernelParams P, uint32_t n_groups, uint32_t *__restrict__ cost) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_groups) return;
    const uint32_t tiles_x = (P.width + RT_TILE_W - 1u) / RT_TILE_W;
    const uint32_t tx = g % tiles_x, ty = g / tiles_x;
    const uint32_t lrow = ty * RT_TILE_H;
    uint32_t y = lrow;
    if (P.band_rows) {
        const uint32_t k = lrow / P.band_rows;
        y = (P.band_first + k * P.band_stride) * P.band_rows + (lrow - k * P.band_rows);
    }
    // the cone around the group's centre ray that holds its corner rays (and a pixel of slack)
    const float x0 = float(tx * RT_TILE_W) - 1.0f, x1 = float(tx * RT_TILE_W + RT_TILE_W) + 1.0f, y0 = float(y) - 1.0f, y1 = float(y + RT_TILE_H) + 1.0f;
    auto dir = [&](float px, float py) { return normalize(mk3(fmaf(P.vhx, py, fmaf(P.hwx, px, P.d00x)), fmaf(P.vhy, py, fmaf(P.hwy, px, P.d00y)), fmaf(P.vhz, py, fmaf(P.hwz, px, P.d00z)))); };
    const f3 a = dir(0.5f * (x0 + x1), 0.5f * (y0 + y1));
    const float cos_half = fminf(fminf(dot(a, dir(x0, y0)), dot(a, dir(x1, y0))), fminf(dot(a, dir(x0, y1)), dot(a, dir(x1, y1))));
    const float sin_half = fast_sqrt(fmaxf(0.0f, 1.0f - cos_half * cos_half));
    float w = 0.0f;
    for (uint32_t i = 0; i < P.n_spheres; i++) {
        const rt_f4 sg = P.sph_geom[i];
        const f3 q = mk3(sg.x - P.ox, sg.y - P.oy, sg.z - P.oz);
        const float lq = fast_sqrt(dot(q, q)), rad = fast_sqrt(sg.w);
        bool reach = lq <= rad;  // the camera is inside
        if (!reach) {
            // angle(a, q) <= half angle of the cone + angle under which the sphere is seen
            const float cq = dot(a, q) * fast_rcp(lq), sq = fast_sqrt(fmaxf(0.0f, 1.0f - cq * cq));
            const float ss = rad * fast_rcp(lq), cs = fast_sqrt(fmaxf(0.0f, 1.0f - ss * ss));
            const float cos_sum = cos_half * cs - sin_half * ss, sin_sum = sin_half * cs + cos_half * ss;  // cos / sin of (half + seen)
            reach = sin_sum < 0.0f || cos_sum <= 0.0f || cq >= cos_sum || (cq > 0.0f && sq <= sin_sum);
        }
        if (reach) {
            const uint32_t kind = __float_as_uint(P.sph_shade[RT_SHADE_RECS * i + 2].x);
            w += kind == RTHIP_MAT_DIELECTRIC ? 6.0f : (kind == RTHIP_MAT_TEXTURE ? 1.0f : 3.0f);
        }
    }
    if (P.n_planes + P.n_tris != 0u) w += 3.0f;  // (not tested: they are few in a scan scene)
    cost[g] = w == 0.0f ? 0u : min(15u, 1u + uint32_t(w * 0.75f));
}
#ifndef RT_HEAD_FRACTION
#define RT_HEAD_FRACTION 0.25f
#endif
#ifndef RT_TAIL_FACTOR
#define RT_TAIL_FACTOR 1.0f
#endif
// order[] = the groups by falling cost (counting sort over 1024 buckets scaled to the largest cost; one workgroup; the order inside
// a bucket is left to the atomics). The costs are either the classes of tile_cost_kernel or MEASURED: the wave times a previous
// launch of the same frame layout took per group (its longest wave, KernelParams::tile_cost_out).
// split (optional, two dwords: head_groups, whole_end) — the MIXED launch of measured costs: order = [head: traced slice by slice]
// [traced in one piece] [tail: slice by slice]. With c the time of a wave that traces a group's tile in one piece and
// T = sum(work of a group) / slots the launch time the costs predict,
//   * head = the groups with c > T / 4: one of their waves in one piece would be a quarter of the launch or more;
//   * tail: a launch ends with a drain — the last pieces to start run for their c while the slots around them empty — that costs about
//     half of that c on every slot. So the cheapest groups that are worth it (c above 2 % of T: not the sky) are MOVED to the very end of
//     the order and traced in slices, as many of them as keep every slot busy for the time of one whole piece of their level
//     (work >= RT_TAIL_FACTOR * slots * c): the launch ends with pieces of c / slices.
//   * the groups that cost next to nothing (below 2 % of T: sky) are SPREAD evenly over the part traced in one piece instead of closing
//     the launch: their waves live for 10-40 us, the GPU starts ~30 waves per us, and 40 000 of them in a row were the last 0.7 ms of the
//     12-ms frame with a third of the slots filled (tools/timeline_mixed.py). tmp: n_groups dwords of scratch for that.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t *__restrict__ cost, uint32_t n_groups, uint32_t *__restrict__ order,
                                                          uint32_t *__restrict__ split, uint32_t *__restrict__ tmp, float slots, float work_mult, float whole_mult) {
    __shared__ uint32_t s_count[1024], s_scan[1024], s_max, s_start[1024], s_cnt[1024], s_take[1024], s_moved_before[1024], s_moved, s_head, s_trivial;
    const uint32_t t = threadIdx.x;
    s_count[t] = 0u;
    s_take[t] = 0u, s_moved_before[t] = 0u;
    if (t == 0u) s_max = 0u, s_moved = 0u, s_head = 0u, s_trivial = 0u;
    __syncthreads();
    uint32_t m = 0u;
    for (uint32_t g = t; g < n_groups; g += 1024u) m = max(m, cost[g]);
    atomicMax(&s_max, m);
    __syncthreads();
    const float scale = 1023.0f / float(max(s_max, 1u));
    auto bucket = [&](uint32_t c) { return 1023u - min(1023u, uint32_t(float(c) * scale)); };  // bucket 0 = the most expensive
    for (uint32_t g = t; g < n_groups; g += 1024u) atomicAdd(&s_count[bucket(cost[g])], 1u);
    __syncthreads();
    // exclusive prefix sum of the bucket counts (Hillis-Steele, 10 steps)
    uint32_t v = s_count[t];
    s_scan[t] = v;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
        const uint32_t add = t >= off ? s_scan[t - off] : 0u;
        __syncthreads();
        s_scan[t] += add;
        __syncthreads();
    }
    s_start[t] = s_scan[t] - v;  // where the bucket starts in the sorted order
    s_cnt[t] = v;
    s_count[t] = 0u;             // ... and from here on: the groups of the bucket placed so far
    __syncthreads();
    if (split != nullptr && t == 0u) {
        // (bucket b holds costs around (1023 - b + 0.5) x the bucket width: everything below is in that unit.
        // work_mult: waves per group x slices of the measuring pass — a group's work is work_mult x its cost; whole_mult: those slices —
        // a wave that traces the group's tile in one piece runs for whole_mult x the cost)
        float total = 0.0f;
        for (int b = 0; b < 1024; b++) total += float(s_cnt[b]) * (float(1023 - b) + 0.5f);
        const float t_launch = total * work_mult / slots;
        uint32_t head = 0;
        int head_buckets = 0;
        for (; head_buckets < 1024 && (float(1023 - head_buckets) + 0.5f) * whole_mult > RT_HEAD_FRACTION * t_launch; head_buckets++) head += s_cnt[head_buckets];
        float fine = 0.0f;  // the work of the groups moved so far
        uint32_t trivial = 0;
        for (int b = 1023; b >= head_buckets; b--) {
            const uint32_t have = s_cnt[b];
            const float c = float(1023 - b) + 0.5f, whole = c * whole_mult, need = RT_TAIL_FACTOR * slots * whole, per = work_mult * c;
            if (whole <= 0.02f * t_launch) trivial += have;
            if (have == 0u || whole <= 0.02f * t_launch) continue;
            if (fine >= need) break;
            const uint32_t take = min(have, uint32_t((need - fine) / per) + 1u);
            s_take[b] = take;
            fine += float(take) * per;
            if (take < have) break;
        }
        uint32_t moved = 0;
        for (int b = 0; b < 1024; b++) s_moved_before[b] = moved, moved += s_take[b];
        s_moved = moved;
        s_head = head, s_trivial = trivial;
        split[0] = head;
        split[1] = n_groups - moved;
    }
    __syncthreads();
    const uint32_t tail_at = n_groups - s_moved;
    const bool spread = split != nullptr && tmp != nullptr && s_trivial != 0u;
    uint32_t *sorted = spread ? tmp : order;  // [head] [whole, worth something] [whole, trivial] [moved]
    for (uint32_t g = t; g < n_groups; g += 1024u) {
        const uint32_t b = bucket(cost[g]);
        const uint32_t r = atomicAdd(&s_count[b], 1u);  // the group's rank inside its bucket
        sorted[r < s_take[b] ? tail_at + s_moved_before[b] + r : s_start[b] - s_moved_before[b] + (r - s_take[b])] = g;
    }
    if (!spread) return;
    __threadfence_block();
    __syncthreads();
    // position j of the whole part holds a trivial group iff floor((j + 1) * trivial / len) > floor(j * trivial / len)
    const uint32_t head = s_head, len = tail_at - head, trivial = s_trivial, worth = len - trivial;
    for (uint32_t k = t; k < n_groups; k += 1024u) {
        uint32_t src = k;
        if (k >= head && k < tail_at) {
            const uint32_t j = k - head;
            const uint32_t before = uint32_t((uint64_t(j) * trivial) / len), upto = uint32_t((uint64_t(j + 1u) * trivial) / len);
            src = head + (upto > before ? worth + before : j - before);
        }
        order[k] = sorted[src];
    }
}
// order_and_cost: 2 x n_groups dwords; fills the first half with the launch order of P's pixel groups
// ... from the cost model (measured == null), or from the costs a previous launch of the same layout measured
// split: device address of two dwords (head_groups, whole_end of a mixed launch), or null; slots: wave slots of the device
// measured_slices: the slices of the launch that measured
extern "C" hipError_t rt_launch_tile_order(const KernelParams *P, uint32_t *order_and_cost, const uint32_t *measured, uint32_t *split, uint32_t slots,
                                           uint32_t measured_slices, hipStream_t stream) {
    (void)hipGetLastError();
    const uint32_t tiles_x = (P->width + RT_TILE_W - 1u) / RT_TILE_W, tiles_y = (P->owned_rows + RT_TILE_H - 1u) / RT_TILE_H;
    const uint32_t n = tiles_x * tiles_y;
    if (n == 0) return hipSuccess;
    if (!measured) hipLaunchKernelGGL(tile_cost_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, *P, n, order_and_cost + n);
    // (the second half of order_and_cost holds the model's costs, or — sorting measurements — serves as scratch)
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, measured ? measured : order_and_cost + n, n, order_and_cost, split,
                       measured ? order_and_cost + n : (uint32_t *)nullptr, float(slots), 4.0f * float(measured_slices), float(measured_slices));
    return hipGetLastError();
}

// Raytracer::get_shape — lib.rs:262-267: one un-jittered primary ray, index of the closest shape.
__global__ void pick_kernel(const KernelParams P, float u, float v, long long *out) {
    __shared__ uint32_t s_stack[RT_BVH_MAX_DEPTH * 256];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t cnt[RT_STAT_N];
    Ray r;
    r.o = mk3(P.ox, P.oy, P.oz);
    const f3 nd = mk3(fmaf(P.vx, v, fmaf(P.hx, u, P.d00x)), fmaf(P.vy, v, fmaf(P.hy, u, P.d00y)), fmaf(P.vz, v, fmaf(P.hz, u, P.d00z)));
    const float l2 = dot(nd, nd), rl = fast_rsq(l2);
    r.d = nd * rl;
    r.tmin = RT_T_MIN * (l2 * rl), r.tmax = RT_T_MAX * (l2 * rl);
    HitRec h = P.n_bvh_nodes ? closest_hit<false, MODE_BVH>(P, r, RT_NONE, cnt, s_stack, nullptr)
                             : closest_hit<false, MODE_GLOBAL>(P, r, RT_NONE, cnt, s_stack, nullptr);
    *out = (h.shape == RT_NONE) ? -1ll : (long long)h.shape;
}

// ---- launch wrappers (called from rthip_api.cpp) -----------------------------------------
template <bool STATS, int MODE, bool SPH_SCENE>
static void launch_variant(const KernelParams &P, bool lights, uint32_t blocks, uint32_t lds, bool var, hipStream_t stream) {
    // var — MODE_BVH: RTHIP_F_TILE_LISTS, the instantiation with the primary-visibility list code; MODE_LDS: one-wave workgroups
    // (MODE_BVH always has them: 256 / wgt workgroups per pixel group)
    constexpr bool can_var = MODE == MODE_BVH || (MODE == MODE_LDS && !STATS && RT_SCAN_WAVE_WG);
    // pixel groups x slices (a mixed launch: its sliced groups x slices + its whole groups)
    const uint32_t units = P.mixed ? (P.n_groups - (P.whole_end - P.head_groups)) * P.n_slices + (P.whole_end - P.head_groups) : blocks * P.n_slices;
    if (can_var && var) {
        constexpr uint32_t wgt = RT_WG_THREADS(MODE, true);
        const dim3 grid(units * (256u / wgt)), block(wgt);
        if (lights) hipLaunchKernelGGL((render_kernel<STATS, MODE, true, SPH_SCENE, can_var>), grid, block, lds, stream, P);
        else hipLaunchKernelGGL((render_kernel<STATS, MODE, false, SPH_SCENE, can_var>), grid, block, lds, stream, P);
    } else {
        constexpr uint32_t wgt = RT_WG_THREADS(MODE, false);
        const dim3 grid(units * (256u / wgt)), block(wgt);
        if (lights) hipLaunchKernelGGL((render_kernel<STATS, MODE, true, SPH_SCENE>), grid, block, lds, stream, P);
        else hipLaunchKernelGGL((render_kernel<STATS, MODE, false, SPH_SCENE>), grid, block, lds, stream, P);
    }
}

template <bool STATS>
static void launch_mode(const KernelParams &P, int mode, bool lights, uint32_t blocks, uint32_t lds, bool var, hipStream_t stream) {
    if (mode == MODE_BVH) launch_variant<STATS, MODE_BVH, false>(P, lights, blocks, lds, var, stream);
    else if (mode == MODE_LDS && P.n_planes == 0 && P.n_tris == 0) launch_variant<STATS, MODE_LDS, true>(P, lights, blocks, lds, var, stream);
    else if (mode == MODE_LDS) launch_variant<STATS, MODE_LDS, false>(P, lights, blocks, lds, var, stream);
    else launch_variant<STATS, MODE_GLOBAL, false>(P, lights, blocks, lds, false, stream);
}

// MODE_BVH: sample rows in the per-wave radiance ring (768 B of LDS per row and wave; measured on C4: 2 rows 77 ms, 4 rows 61 ms,
// 8 rows — one workgroup fewer per CU — 72 ms)
static uint32_t default_ring_rows() {
    static const int env = [] {
        const char *e = getenv("RTHIP_RING_ROWS");  // tuning knob; results do not depend on it
        return e ? atoi(e) : 0;
    }();
    if (env > 0) {
        uint32_t c = 1;
        while (c * 2 <= uint32_t(env) && c < 16) c *= 2;  // a power of two, so that it divides the sample-group size
        return c;
    }
    return RT_RING_ROWS;
}

extern "C" hipError_t rt_launch_render(const KernelParams *Pin, int count_stats, hipStream_t stream) {
    (void)hipGetLastError();  // hipGetLastError() below must report THIS launch, not an older failure of the thread
    KernelParams P = *Pin;
    const uint32_t tiles_x = (P.width + RT_TILE_W - 1u) / RT_TILE_W, tiles_y = (P.owned_rows + RT_TILE_H - 1u) / RT_TILE_H;
    uint32_t blocks = tiles_x * tiles_y;
    if (P.worklist) blocks = uint32_t((uint64_t(P.owned_rows) * P.width + 255u) / 256u);  // worst case: nothing is Final yet
    if (blocks == 0 || P.spp == 0) return hipSuccess;
    const bool lights = (P.flags & RTHIP_F_POINT_LIGHTS) && P.n_lights > 0;
    const int mode = P.n_bvh_nodes > 0 ? MODE_BVH : ((P.n_spheres <= RT_LDS_SPHERES && !P.tree_scene) ? MODE_LDS : MODE_GLOBAL);
    uint32_t lds = 0;
    bool var = mode == MODE_BVH && P.tile_pre != nullptr;  // the fifth template parameter of render_kernel (launch_variant)
    if (mode == MODE_BVH) {
        if (P.bvh_levels == 0 || P.bvh_levels > RT_BVH_MAX_DEPTH) return hipErrorInvalidValue;
        P.ring_rows = default_ring_rows();  // rows of the radiance ring: a power of two <= 16
        // only the first levels of the traversal stack live in LDS, deeper ones in private memory (a lane is that deep on
        // a few per cent of its pushes): 8 KB + 2 KB + 12 KB lets 7 workgroups share a CU's 160 KB
        uint32_t lds_levels = RT_BVH_LDS_LEVELS;
        if (const char *e = getenv("RTHIP_STACK_LDS")) lds_levels = uint32_t(atoi(e));  // tuning knob; results do not depend on it
        if (lds_levels < P.bvh_levels) P.bvh_levels = lds_levels;
        // RT_BVH_TOP_NODES > 0 (off by default: measured slower): that many nodes of the top of the tree in LDS, 32 B each
        // (breadth-first prefix; knob RTHIP_BVH_TOP; results do not depend on it)
        uint32_t top = RT_BVH_TOP_NODES;
        if (const char *e = getenv("RTHIP_BVH_TOP")) top = RT_BVH_TOP_NODES > 0 ? uint32_t(atoi(e)) : 0u;
        if (top > P.bvh_top_nodes) top = P.bvh_top_nodes;
        P.bvh_top_nodes = top;
        // per wave: stack levels x 64 lanes, the pixel table (8 dwords; 64 in a work-list pass), 3 colour planes x ring rows x 64 radiance slots,
        // 16 row counters, 16 row masks
        const uint32_t waves = RT_WG_THREADS(MODE_BVH, false) / 64u;
        // (gfx950 hands out LDS in blocks of 1280 B, 128 per CU: 7 levels + the row table + a ring of 4 rows = 5088 B = 4 blocks per one-wave
        // workgroup, so 28 waves fit a CU = the 7 per SIMD that 72 VGPRs allow; at 5 blocks (8 levels and a wider pixel table, round 2) 25 fit.
        // Measured with RTHIP_STACK_LDS, tools/stack_lds_sweep.py: C4 56.1 / 56.6 / 57.4 ms at 8 / 7 / 6 levels of 5 blocks, 54.5 at 4 blocks.)
        lds = waves * (P.bvh_levels * 256u + (P.worklist ? 64u : 8u) * 4u + 3u * 64u * 4u * P.ring_rows + 16u * 4u + 16u * 8u) + top * 32u;
    } else {
        P.ring_rows = 1;  // lane = pixel: no radiance slots
        P.bvh_levels = 0;
        P.bvh_top_nodes = 0;
        // (the variant with the deferred texture queue: the queues first, then geometry sized by the scene — see render_kernel)
        const bool queue = RT_TEX_QUEUE != 0 && !count_stats && !lights;
        const uint32_t n_padded = (P.n_spheres + RT_SCAN_UNROLL - 1u) & ~(RT_SCAN_UNROLL - 1u);
        if (mode != MODE_LDS) lds = 16u;
        else if (queue) lds = 4u * RT_QUEUE_WAVE_DWORDS * 4u + (n_padded + RT_SHADE_RECS * P.n_spheres + RT_PRIMARY_SPHERES) * 16u;
        else lds = (n_padded + RT_SHADE_RECS * P.n_spheres + RT_PRIMARY_SPHERES) * 16u;
        // one-wave workgroups (every wave stages the scene for itself) while 28 of them fit a CU's 128 LDS allocation blocks of 1280 B:
        // scenes of up to 51 spheres (RT_SCAN_WAVE_WG)
        var = mode == MODE_LDS && !queue && !count_stats && (lds + 1279u) / 1280u <= 4u;
    }
    if (lds > 65536u) return hipErrorInvalidValue;
    if (P.n_slices == 0 || P.slice_spp == 0 || (P.n_slices > 1 && (uint64_t(P.n_slices) * P.slice_spp < P.spp || !P.partial))) return hipErrorInvalidValue;
    // (whether the launch is sliced — n_slices > 1 — is the caller's decision, render_pass in rthip_api.cpp; same image either way)
    if (count_stats) launch_mode<true>(P, mode, lights, blocks, lds, var, stream);
    else launch_mode<false>(P, mode, lights, blocks, lds, var, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || P.n_slices == 1) return e;
    uint32_t rblocks = uint32_t((uint64_t(P.owned_rows) * P.width + 255u) / 256u);
    if (P.mixed) rblocks = P.n_groups - (P.whole_end - P.head_groups);  // one workgroup per sliced pixel group
    if (rblocks == 0) return e;
    hipLaunchKernelGGL(resolve_kernel, dim3(rblocks), dim3(256), 0, stream, P);
    return hipGetLastError();
}

// ---- multi-GPU frame assembly (rthip_group_render) ----------------------------------------------------
// Row y of the image belongs to band b = y / R, owned by rank b % n as its local row (b / n) * R + (y - b * R)
// (rthip_band_row_owner states the same map for the host). Rank 0's rows are read from its own band buffer, the other
// ranks' from the receive buffer of the gather: [rank][pad_rows][units] 4-byte units per row.
__global__ __launch_bounds__(256) void deinterleave_kernel(uint32_t *__restrict__ full, const uint32_t *__restrict__ recv, const uint32_t *__restrict__ own,
                                                           uint32_t units, uint32_t height, uint32_t band_rows, uint32_t ndev, uint32_t pad_rows) {
    const uint32_t y = blockIdx.y;
    const uint32_t b = y / band_rows, rank = b % ndev, lrow = (b / ndev) * band_rows + (y - b * band_rows);
    const uint32_t *src = rank == 0 ? own + size_t(lrow) * units : recv + (size_t(rank) * pad_rows + lrow) * units;
    uint32_t *dst = full + size_t(y) * units;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < units; i += gridDim.x * 256u) dst[i] = src[i];
}
extern "C" hipError_t rt_launch_deinterleave(void *full, const void *recv, const void *own, uint32_t units, uint32_t height, uint32_t band_rows,
                                             uint32_t ndev, uint32_t pad_rows, hipStream_t stream) {
    if (units == 0 || height == 0) return hipSuccess;
    if (band_rows == 0 || ndev == 0 || height > 65535u) return hipErrorInvalidValue;  // gridDim.y
    (void)hipGetLastError();  // hipGetLastError() below must report THIS launch, not an older failure of the thread
    const uint32_t bx = (units + 255u) / 256u;
    hipLaunchKernelGGL(deinterleave_kernel, dim3(bx > 64u ? 64u : bx, height), dim3(256), 0, stream, (uint32_t *)full, (const uint32_t *)recv,
                       (const uint32_t *)own, units, height, band_rows, ndev, pad_rows);
    return hipGetLastError();
}

// ---- rthip_render_device_noalloc: the scratch of callers that cannot allocate ---------------------------------------
// A file-scope device array, one copy per GPU, held until the process ends. In host code its NAME is a host-side stand-in,
// not a device address: the address comes from hipGetSymbolAddress, once per device with that device current
// (rt_scratch_init, called by rthip_create), and host code only ever uses that pointer (kernel arguments, hipMemsetAsync).
static_assert(RT_NOALLOC_STATS_BYTES >= RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long) && RT_NOALLOC_STATS_BYTES % 256 == 0, "scratch layout");
static __device__ __align__(256) unsigned char scratch[RTHIP_NOALLOC_SCRATCH_BYTES];
#define RT_MAX_DEVICES 64
static void *g_scratch_ptr[RT_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// the calling thread's current device must be `device`
extern "C" hipError_t rt_scratch_init(int device) {
    if (device < 0 || device >= RT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}
extern "C" void *rt_scratch_ptr(int device) {
    if (device < 0 || device >= RT_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

extern "C" hipError_t rt_launch_pick(const KernelParams *P, float u, float v, long long *d_out, hipStream_t stream) {
    (void)hipGetLastError();  // hipGetLastError() below must report THIS launch, not an older failure of the thread
    hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(64), 0, stream, *P, u, v, d_out);
    return hipGetLastError();
}
