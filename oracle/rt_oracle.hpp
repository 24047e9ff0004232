// rt_oracle.hpp — CPU restatement of rustracer's raytracer_core render path.
//
// TEST INFRASTRUCTURE ONLY. Nothing in the product path (rustracer_amd/, include/) may include,
// link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// Every function cites the reference file:line (relative to /root/reference) it restates.
// `Real = double` is the reference-faithful mode (the reference is f64 throughout);
// `Real = float` evaluates the same formulas in binary32 (what the device arithmetic is).
//
// PARITY PINNING: the reference has no tests, golden vectors or fixtures and cannot
// be built here (Rust nightly, no toolchain), and its renders are entropy-seeded
// (SURVEY F4). What IS pinned: the closed-form known-answer vectors of SURVEY §8c (camera
// basis, closest hits of the standalone scene, UV, sky, throughput constants), checked in
// tests/test_oracle_kat.py. The random-number stream (rand 0.7 SmallRng, not vendored, entropy
// seeded) is replaced by the counter RNG below => image-level parity vs the Rust binary is
// "parity unpinned" and can only be statistical. Extensions (plane, triangle, BVH, point
// lights, max_depth) have no reference behaviour at all: this file DEFINES them.
#pragma once

#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/rthip.h"

namespace oracle {

// ----------------------------------------------------------------------------------------
// Counter-based RNG (replaces rand 0.7 SmallRng + gen_range).
// Stateless: value = f(seed, pixel_index, sample, slot). The device kernel implements the same
// integer function independently (rustracer_amd/csrc/rt_rng.h); spec in DESIGN.md §RNG.
//   pixel_key = mix32(mix32(seed ^ 0x52545243) + pixel_index)
//   pair of (sample s, slot j): h = mix32(pixel_key + (s*256 + j)*0x9E3779B9)
//       u_a = (h >> 8) / 2^24;  g = h*0x2C1B3C6D; g ^= g >> 15;  u_b = (g >> 8) / 2^24
//   slot 0: pixel jitter (u_a = x drawn first, u_b = y; lib.rs:283,285);
//   slot 1+bounce: material draws of segment `bounce` (u_a drawn first).
// ----------------------------------------------------------------------------------------
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
static inline uint32_t pixel_key(uint32_t seed, uint32_t pixel_index) { return mix32(mix32(seed ^ 0x52545243U) + pixel_index); }
// "key" of a path = the counter of its slot 0
static inline uint32_t path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample) {
    return pixel_key(seed, pixel_index) + (sample * 256U) * 0x9E3779B9U;
}
static inline uint32_t rng_first(uint32_t key, uint32_t slot) { return mix32(key + slot * 0x9E3779B9U); }
static inline uint32_t rng_second(uint32_t h) {
    uint32_t g = h * 0x2C1B3C6DU;
    return g ^ (g >> 15);
}
// U[0,1) on a 24-bit grid: exactly representable in binary32 and binary64, so both modes see the same xi.
template <class Real>
static inline Real unit_from_bits(uint32_t bits) { return Real(bits >> 8) * Real(1.0 / 16777216.0); }
// the pair (u_a, u_b) of a slot
template <class Real>
static inline void rng_pair(uint32_t key, uint32_t slot, Real *ua, Real *ub) {
    const uint32_t h = rng_first(key, slot);
    *ua = unit_from_bits<Real>(h);
    *ub = unit_from_bits<Real>(rng_second(h));
}

// ----------------------------------------------------------------------------------------
template <class Real>
struct V3 {
    Real x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Real a, Real b, Real c) : x(a), y(b), z(c) {}
    template <class U>
    explicit V3(const U *p) : x(Real(p[0])), y(Real(p[1])), z(Real(p[2])) {}
    V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
    V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 operator*(Real s) const { return V3(x * s, y * s, z * s); }
    V3 operator/(Real s) const { return V3(x / s, y / s, z / s); }
    Real dot(const V3 &o) const { return x * o.x + y * o.y + z * o.z; }
    V3 cross(const V3 &o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
    Real magnitude_squared() const { return dot(*this); }
    V3 normalize() const { return *this / std::sqrt(magnitude_squared()); }  // nalgebra normalize
};

template <class Real>
struct Consts {
    static constexpr Real PI = Real(3.14159265358979323846264338327950288);
    static constexpr Real TAU = Real(6.28318530717958647692528676655900577);
    static constexpr Real T_MIN = Real(0.001);     // ray.rs:5
    static constexpr Real T_MAX = Real(100000.0);  // ray.rs:6
};

// Camera::new_lookat — camera.rs:19-41. Always evaluated in double (host-side set-up).
static inline void camera_lookat(double x, double y, double z, const double lookat_[3], rthip_camera *out) {
    using D3 = V3<double>;
    D3 origin(x, y, z), lookat(lookat_), vup(0.0, 1.0, 0.0);
    const double vertical_field_of_view = 20.0;
    const double viewport_height = 2.0 * (vertical_field_of_view * (Consts<double>::PI / 180.0));  // camera.rs:24 (to_radians)
    const double viewport_width = (16.0 / 9.0) * viewport_height;                                  // camera.rs:5,25
    D3 w = (origin - lookat).normalize();  // camera.rs:27
    D3 u = vup.cross(w).normalize();       // camera.rs:28
    D3 v = w.cross(u);                     // camera.rs:29
    D3 horizontal = u * viewport_width;    // camera.rs:31
    D3 vertical = v * viewport_height;     // camera.rs:32
    D3 llc = origin - horizontal / 2.0 - vertical / 2.0 - w;  // camera.rs:39
    D3 d00 = llc - origin;                                    // camera.rs:46
    out->origin[0] = origin.x, out->origin[1] = origin.y, out->origin[2] = origin.z;
    out->d00[0] = d00.x, out->d00[1] = d00.y, out->d00[2] = d00.z;
    out->horizontal[0] = horizontal.x, out->horizontal[1] = horizontal.y, out->horizontal[2] = horizontal.z;
    out->vertical[0] = vertical.x, out->vertical[1] = vertical.y, out->vertical[2] = vertical.z;
}

template <class Real>
struct Ray {
    V3<Real> origin, direction;
    V3<Real> at(Real t) const { return direction * t + origin; }  // ray.rs:30-32
};

// Camera::emit_ray_at — camera.rs:43-50: llc - origin + horizontal*ox + vertical*oy, left to right.
template <class Real>
static inline Ray<Real> emit_ray_at(const rthip_camera &c, Real ox, Real oy) {
    V3<Real> o(c.origin), d00(c.d00), H(c.horizontal), V(c.vertical);
    return Ray<Real>{o, d00 + H * ox + V * oy};
}

// Ray::background_color — ray.rs:89-93
template <class Real>
static inline V3<Real> background_color(const Ray<Real> &r) {
    Real t = Real(0.5) * (r.direction.normalize().y + Real(1.0));
    return V3<Real>(1, 1, 1) * (Real(1.0) - t) + V3<Real>(Real(0.5), Real(0.7), Real(1.0)) * t;
}

// `f64 as i32` (Rust): truncate toward zero, saturate, NaN -> 0.   texture.rs:35
static inline int32_t cast_i32(double v) {
    if (std::isnan(v)) return 0;
    if (v >= 2147483647.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return int32_t(v);
}
// `f64 as u8` (Rust): truncate, saturate, NaN -> 0.   lib.rs:30-38
static inline uint8_t cast_u8(double v) {
    if (std::isnan(v)) return 0;
    if (v >= 255.0) return 255;
    if (v <= 0.0) return 0;
    return uint8_t(v);
}

// Texture::wrap — texture.rs:32-41
template <class Real>
static inline uint32_t texture_wrap(Real val, Real scale, uint32_t bound) {
    int32_t signed_bound = int32_t(bound);
    Real float_coord = val * scale * Real(bound);
    int32_t wrapped = cast_i32(double(float_coord)) % signed_bound;
    return wrapped < 0 ? uint32_t(wrapped + signed_bound) : uint32_t(wrapped);
}

// generate_pixel's tonemap — lib.rs:294-299 + 30-38: mean, clamp, sqrt, *255, `as u8`.
template <class Real>
static inline uint8_t tonemap_channel(Real sum, uint64_t nb_samples) {
    Real scale = Real(1.0) / Real(nb_samples);
    Real c = sum * scale;
    // f64::clamp(0,1): NaN stays NaN
    if (c < Real(0)) c = Real(0);
    if (c > Real(1)) c = Real(1);
    c = std::sqrt(c) * Real(255.0);
    return cast_u8(double(c));
}

// Dielectric::shlick — dielectric.rs:47-52 (powf(5.0): integer power, sign preserved)
template <class Real>
static inline Real schlick(Real cosine, Real refraction_index) {
    Real r0 = (Real(1.0) - refraction_index) / (Real(1.0) + refraction_index);
    Real r1 = r0 * r0;
    return r1 + (Real(1.0) - r1) * std::pow(Real(1.0) - cosine, Real(5.0));
}

// ----------------------------------------------------------------------------------------
// Scene in Real precision
// ----------------------------------------------------------------------------------------
enum ShapeKind : uint32_t { SHAPE_SPHERE = 0, SHAPE_PLANE = 1, SHAPE_TRIANGLE = 2 };

template <class Real>
struct Hit {
    Real t;
    V3<Real> position;
    int64_t shape;  // global shape index, -1 = none
    Real b1, b2;    // triangle barycentrics (extension)
};

template <class Real>
struct Scene {
    using Vec = V3<Real>;
    struct Sphere { Vec center; Real radius; uint32_t material; };
    struct Plane { Vec point, normal, tu, tv; uint32_t material; };
    struct Tri { Vec v0, v1, v2; uint32_t material; };
    struct Material { uint32_t kind; uint32_t texture; Vec albedo_over_pi; Real param; Real shininess; };
    struct Texture { uint32_t width, height; const uint8_t *rgb8; };
    struct Light { Vec position, intensity; };

    std::vector<Sphere> spheres;
    std::vector<Plane> planes;
    std::vector<Tri> tris;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<Light> lights;

    explicit Scene(const rthip_scene_desc &d) {
        for (uint64_t i = 0; i < d.n_spheres; i++)
            spheres.push_back({Vec(d.spheres[i].center), Real(d.spheres[i].radius), d.spheres[i].material});
        for (uint64_t i = 0; i < d.n_planes; i++) {
            // extension: normal normalised in double on load; tangent frame for texture coordinates
            V3<double> n = V3<double>(d.planes[i].normal).normalize();
            V3<double> a = std::fabs(n.x) > 0.9 ? V3<double>(0, 1, 0) : V3<double>(1, 0, 0);
            V3<double> tu = a.cross(n).normalize();
            V3<double> tv = n.cross(tu);
            planes.push_back({Vec(d.planes[i].point), Vec(Real(n.x), Real(n.y), Real(n.z)), Vec(Real(tu.x), Real(tu.y), Real(tu.z)),
                              Vec(Real(tv.x), Real(tv.y), Real(tv.z)), d.planes[i].material});
        }
        for (uint64_t i = 0; i < d.n_triangles; i++)
            tris.push_back({Vec(d.triangles[i].v0), Vec(d.triangles[i].v1), Vec(d.triangles[i].v2), d.triangles[i].material});
        for (uint64_t i = 0; i < d.n_materials; i++) {
            const rthip_material &m = d.materials[i];
            // constructors store albedo / PI: lambertian_diffuse.rs:21,32; metal.rs:21; dielectric.rs:22
            Vec a = Vec(m.albedo) / Consts<Real>::PI;
            materials.push_back({m.kind, m.texture, a, Real(m.param), Real(m.shininess)});
        }
        for (uint64_t i = 0; i < d.n_textures; i++)
            textures.push_back({d.textures[i].width, d.textures[i].height, d.textures[i].rgb8});
        for (uint64_t i = 0; i < d.n_lights; i++) lights.push_back({Vec(d.lights[i].position), Vec(d.lights[i].intensity)});
    }

    uint64_t n_shapes() const { return spheres.size() + planes.size() + tris.size(); }

    // ---- optional accelerator (test infrastructure of the test infrastructure) ------------------------------------------
    // The reference scans every shape (ray.rs:47-67) and so does find_collision() below; on the million-triangle configuration
    // that is 1e6 tests per ray. `build_accel()` puts the spheres and triangles in a median-split bounding-volume tree (boxes
    // padded far beyond rounding) and find_collision()/occluded() then test only the shapes whose box the ray enters. The
    // answer is the scan's answer, bit for bit: the same collide_* functions with the same FIXED window, the candidate set is a
    // superset of the shapes that can be hit, and the scan's "later index wins on equal t" is applied explicitly. Checked
    // against the scan itself in tests/test_oracle_accel.py.
    struct AccelNode { Real lo[3], hi[3]; uint32_t left, right, first, count; };  // count > 0: leaf over accel_prims[first, first + count)
    std::vector<AccelNode> accel_nodes;
    std::vector<uint32_t> accel_prims;  // 0 .. n_spheres-1: spheres; n_spheres + i: triangle i
    bool accel = false;

    void prim_box(uint32_t prim, double lo[3], double hi[3]) const {
        if (prim < spheres.size()) {
            const Sphere &s = spheres[prim];
            const double c[3] = {double(s.center.x), double(s.center.y), double(s.center.z)}, r = std::fabs(double(s.radius));
            for (int k = 0; k < 3; k++) lo[k] = c[k] - r, hi[k] = c[k] + r;
        } else {
            const Tri &t = tris[prim - spheres.size()];
            const double v[3][3] = {{double(t.v0.x), double(t.v0.y), double(t.v0.z)}, {double(t.v1.x), double(t.v1.y), double(t.v1.z)},
                                    {double(t.v2.x), double(t.v2.y), double(t.v2.z)}};
            for (int k = 0; k < 3; k++) lo[k] = std::min(v[0][k], std::min(v[1][k], v[2][k])), hi[k] = std::max(v[0][k], std::max(v[1][k], v[2][k]));
        }
        // padding: 1e-4 (float) / 1e-9 (double) of the box's size and distance from the origin, i.e. thousands of ulps
        const double rel = sizeof(Real) == 4 ? 1e-4 : 1e-9;
        for (int k = 0; k < 3; k++) {
            const double pad = rel * ((hi[k] - lo[k]) + std::max(std::fabs(lo[k]), std::fabs(hi[k]))) + 1e-30;
            lo[k] -= pad, hi[k] += pad;
        }
    }
    uint32_t accel_build_range(uint32_t begin, uint32_t end, const std::vector<double> &boxes) {
        const uint32_t me = uint32_t(accel_nodes.size());
        accel_nodes.emplace_back();
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = begin; i < end; i++) {
            const double *b = &boxes[6 * size_t(accel_prims[i])];
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(lo[k], b[k]), hi[k] = std::max(hi[k], b[3 + k]);
                const double c = 0.5 * (b[k] + b[3 + k]);
                clo[k] = std::min(clo[k], c), chi[k] = std::max(chi[k], c);
            }
        }
        {
            AccelNode &n = accel_nodes[me];
            for (int k = 0; k < 3; k++) {
                // outward rounding into Real (float boxes must still contain the double boxes)
                n.lo[k] = Real(lo[k]), n.hi[k] = Real(hi[k]);
                if (double(n.lo[k]) > lo[k]) n.lo[k] = std::nextafter(n.lo[k], Real(-INFINITY));
                if (double(n.hi[k]) < hi[k]) n.hi[k] = std::nextafter(n.hi[k], Real(INFINITY));
            }
            n.left = n.right = 0, n.first = begin, n.count = end - begin;
        }
        if (end - begin <= 4) return me;
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const uint32_t mid = begin + (end - begin) / 2;
        std::nth_element(accel_prims.begin() + begin, accel_prims.begin() + mid, accel_prims.begin() + end, [&](uint32_t a, uint32_t b) {
            const double ca = boxes[6 * size_t(a) + axis] + boxes[6 * size_t(a) + 3 + axis], cb = boxes[6 * size_t(b) + axis] + boxes[6 * size_t(b) + 3 + axis];
            return ca < cb || (ca == cb && a < b);
        });
        const uint32_t l = accel_build_range(begin, mid, boxes), r = accel_build_range(mid, end, boxes);
        accel_nodes[me].left = l, accel_nodes[me].right = r, accel_nodes[me].count = 0;
        return me;
    }
    void build_accel() {
        const size_t n = spheres.size() + tris.size();
        accel_nodes.clear(), accel_prims.resize(n);
        accel = n > 0;
        if (!accel) return;
        std::vector<double> boxes(6 * n);
        for (size_t i = 0; i < n; i++) {
            accel_prims[i] = uint32_t(i);
            prim_box(uint32_t(i), &boxes[6 * i], &boxes[6 * i + 3]);
        }
        accel_nodes.reserve(n / 2 + 8);
        accel_build_range(0, uint32_t(n), boxes);
    }
    // does the ray enter the node's box somewhere in [T_MIN, limit]? (conservative: boxes are padded, comparisons inclusive)
    static bool accel_enters(const AccelNode &n, const Ray<Real> &ray, const Real inv[3], Real limit) {
        const Real o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
        Real tn = Consts<Real>::T_MIN, tf = limit;
        for (int k = 0; k < 3; k++) {
            if (d[k] == Real(0.0)) {
                if (o[k] < n.lo[k] || o[k] > n.hi[k]) return false;
                continue;
            }
            Real t0 = (n.lo[k] - o[k]) * inv[k], t1 = (n.hi[k] - o[k]) * inv[k];
            if (t0 > t1) std::swap(t0, t1);
            // one more safety margin for the rounding of the two products
            t0 -= std::fabs(t0) * Real(1e-6), t1 += std::fabs(t1) * Real(1e-6);
            if (t0 > tn) tn = t0;
            if (t1 < tf) tf = t1;
        }
        return tn <= tf;
    }
    // one candidate of the closest-hit query: the scan's rule (keep the old hit only if old.t < new.t: equal t, later index wins)
    static void accel_take(bool ok, Real t, int64_t index, Real b1, Real b2, bool *have, Real *best_t, int64_t *best, Real *bb1, Real *bb2) {
        if (!ok) return;
        if (!*have || t < *best_t || (t == *best_t && index > *best)) *have = true, *best_t = t, *best = index, *bb1 = b1, *bb2 = b2;
    }

    // Sphere::collide — sphere.rs:29-61
    static bool collide_sphere(const Sphere &s, const Ray<Real> &ray, Real t_min, Real t_max, Real *t_out) {
        Vec oc = ray.origin - s.center;
        Real a = ray.direction.magnitude_squared();
        Real half_b = oc.dot(ray.direction);
        Real c = oc.magnitude_squared() - s.radius * s.radius;
        Real discriminant = half_b * half_b - a * c;
        if (discriminant > Real(0.0)) {
            Real root = std::sqrt(discriminant);
            Real t = (-half_b - root) / a;
            if (t < t_max && t > t_min) { *t_out = t; return true; }
            t = (-half_b + root) / a;
            if (t < t_max && t > t_min) { *t_out = t; return true; }
        }
        return false;
    }
    // extension: infinite plane (no reference counterpart)
    static bool collide_plane(const Plane &p, const Ray<Real> &ray, Real t_min, Real t_max, Real *t_out) {
        Real denom = ray.direction.dot(p.normal);
        if (denom == Real(0.0)) return false;
        Real t = (p.point - ray.origin).dot(p.normal) / denom;
        if (t < t_max && t > t_min) { *t_out = t; return true; }
        return false;
    }
    // extension: Moller-Trumbore, two-sided (no reference counterpart)
    static bool collide_tri(const Tri &tr, const Ray<Real> &ray, Real t_min, Real t_max, Real *t_out, Real *b1_out, Real *b2_out) {
        Vec e1 = tr.v1 - tr.v0, e2 = tr.v2 - tr.v0;
        Vec pvec = ray.direction.cross(e2);
        Real det = e1.dot(pvec);
        if (det == Real(0.0)) return false;
        Real inv = Real(1.0) / det;
        Vec tvec = ray.origin - tr.v0;
        Real b1 = tvec.dot(pvec) * inv;
        if (!(b1 >= Real(0.0) && b1 <= Real(1.0))) return false;
        Vec qvec = tvec.cross(e1);
        Real b2 = ray.direction.dot(qvec) * inv;
        if (!(b2 >= Real(0.0) && b1 + b2 <= Real(1.0))) return false;
        Real t = e2.dot(qvec) * inv;
        if (t < t_max && t > t_min) { *t_out = t; *b1_out = b1; *b2_out = b2; return true; }
        return false;
    }

    // Ray::find_collision — ray.rs:47-67. Every shape is tested against the FIXED window
    // (T_MIN, T_MAX); the old hit is kept only if old.t < new.t, so on equal t the LATER index wins.
    bool find_collision(const Ray<Real> &ray, Hit<Real> *hit, rthip_stats *st) const {
        bool have = false;
        Real best_t = 0, bb1 = 0, bb2 = 0;
        int64_t best = -1;
        int64_t index = 0;
        const Real t_min = Consts<Real>::T_MIN, t_max = Consts<Real>::T_MAX;
        if (accel) {
            uint64_t n_sph = 0, n_tri = 0;
            const Real inv[3] = {Real(1.0) / ray.direction.x, Real(1.0) / ray.direction.y, Real(1.0) / ray.direction.z};
            uint32_t stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const AccelNode &n = accel_nodes[stack[--sp]];
                if (!accel_enters(n, ray, inv, have ? best_t : t_max)) continue;
                if (n.count == 0) {
                    stack[sp++] = n.left, stack[sp++] = n.right;
                    continue;
                }
                for (uint32_t i = n.first; i < n.first + n.count; i++) {
                    const uint32_t prim = accel_prims[i];
                    Real t = 0, b1 = 0, b2 = 0;
                    if (prim < spheres.size()) {
                        n_sph++;
                        const bool ok = collide_sphere(spheres[prim], ray, t_min, t_max, &t);  // (first: argument evaluation order is unspecified)
                        accel_take(ok, t, int64_t(prim), 0, 0, &have, &best_t, &best, &bb1, &bb2);
                    } else {
                        n_tri++;
                        const uint32_t ti = prim - uint32_t(spheres.size());
                        const bool ok = collide_tri(tris[ti], ray, t_min, t_max, &t, &b1, &b2);
                        accel_take(ok, t, int64_t(spheres.size() + planes.size() + ti), b1, b2, &have, &best_t, &best, &bb1, &bb2);
                    }
                }
            }
            for (size_t i = 0; i < planes.size(); i++) {
                Real t = 0;
                const bool ok = collide_plane(planes[i], ray, t_min, t_max, &t);
                accel_take(ok, t, int64_t(spheres.size() + i), 0, 0, &have, &best_t, &best, &bb1, &bb2);
            }
            if (st) st->rays++, st->sphere_tests += n_sph, st->plane_tests += planes.size(), st->tri_tests += n_tri;
            if (!have) return false;
            hit->t = best_t, hit->position = ray.at(best_t), hit->shape = best, hit->b1 = bb1, hit->b2 = bb2;
            return true;
        }
        for (const Sphere &s : spheres) {
            Real t;
            if (collide_sphere(s, ray, t_min, t_max, &t)) {
                if (!(have && best_t < t)) { have = true; best_t = t; best = index; }
            }
            index++;
        }
        for (const Plane &p : planes) {
            Real t;
            if (collide_plane(p, ray, t_min, t_max, &t)) {
                if (!(have && best_t < t)) { have = true; best_t = t; best = index; }
            }
            index++;
        }
        for (const Tri &tr : tris) {
            Real t, b1, b2;
            if (collide_tri(tr, ray, t_min, t_max, &t, &b1, &b2)) {
                if (!(have && best_t < t)) { have = true; best_t = t; best = index; bb1 = b1; bb2 = b2; }
            }
            index++;
        }
        if (st) {
            st->rays++;
            st->sphere_tests += spheres.size();
            st->plane_tests += planes.size();
            st->tri_tests += tris.size();
        }
        if (!have) return false;
        hit->t = best_t;
        hit->position = ray.at(best_t);  // sphere.rs:41,51 (ray.at)
        hit->shape = best;
        hit->b1 = bb1;
        hit->b2 = bb2;
        return true;
    }

    // extension: shadow-ray any-hit for the point-light mode; direction = light - p, occluded iff a
    // shape is hit with T_MIN < t < 1.
    bool occluded(const Ray<Real> &ray, rthip_stats *st) const {
        const Real t_min = Consts<Real>::T_MIN, t_max = Real(1.0);
        if (accel) {
            if (st) st->rays++, st->shadow_rays++, st->plane_tests += planes.size();
            Real t, b1, b2;
            for (const Plane &p : planes)
                if (collide_plane(p, ray, t_min, t_max, &t)) return true;
            const Real inv[3] = {Real(1.0) / ray.direction.x, Real(1.0) / ray.direction.y, Real(1.0) / ray.direction.z};
            uint32_t stack[128];
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const AccelNode &n = accel_nodes[stack[--sp]];
                if (!accel_enters(n, ray, inv, t_max)) continue;
                if (n.count == 0) {
                    stack[sp++] = n.left, stack[sp++] = n.right;
                    continue;
                }
                for (uint32_t i = n.first; i < n.first + n.count; i++) {
                    const uint32_t prim = accel_prims[i];
                    if (prim < spheres.size()) {
                        if (st) st->sphere_tests++;
                        if (collide_sphere(spheres[prim], ray, t_min, t_max, &t)) return true;
                    } else {
                        if (st) st->tri_tests++;
                        if (collide_tri(tris[prim - spheres.size()], ray, t_min, t_max, &t, &b1, &b2)) return true;
                    }
                }
            }
            return false;
        }
        if (st) {
            st->rays++;
            st->shadow_rays++;
            st->sphere_tests += spheres.size();
            st->plane_tests += planes.size();
            st->tri_tests += tris.size();
        }
        Real t, b1, b2;
        for (const Sphere &s : spheres)
            if (collide_sphere(s, ray, t_min, t_max, &t)) return true;
        for (const Plane &p : planes)
            if (collide_plane(p, ray, t_min, t_max, &t)) return true;
        for (const Tri &tr : tris)
            if (collide_tri(tr, ray, t_min, t_max, &t, &b1, &b2)) return true;
        return false;
    }

    ShapeKind kind_of(int64_t shape, uint64_t *local) const {
        uint64_t s = uint64_t(shape);
        if (s < spheres.size()) { *local = s; return SHAPE_SPHERE; }
        s -= spheres.size();
        if (s < planes.size()) { *local = s; return SHAPE_PLANE; }
        *local = s - planes.size();
        return SHAPE_TRIANGLE;
    }
    uint32_t material_of(int64_t shape) const {
        uint64_t l;
        switch (kind_of(shape, &l)) {
            case SHAPE_SPHERE: return spheres[l].material;
            case SHAPE_PLANE: return planes[l].material;
            default: return tris[l].material;
        }
    }
    // Shape::normal_at_position — sphere.rs:63-65: (p - c)/r, always outward, never flipped.
    Vec normal_at(const Hit<Real> &h) const {
        uint64_t l;
        switch (kind_of(h.shape, &l)) {
            case SHAPE_SPHERE: return (h.position - spheres[l].center) / spheres[l].radius;
            case SHAPE_PLANE: return planes[l].normal;
            default: return (tris[l].v1 - tris[l].v0).cross(tris[l].v2 - tris[l].v0).normalize();
        }
    }
    // Shape::texture_coords_at_position — sphere.rs:67-73
    void texture_coords(const Hit<Real> &h, Real *u, Real *v) const {
        uint64_t l;
        switch (kind_of(h.shape, &l)) {
            case SHAPE_SPHERE: {
                Vec n = h.position - spheres[l].center;
                *u = (Real(1.0) + std::atan2(n.z, n.x) / Consts<Real>::PI) * Real(0.5);
                *v = std::acos(n.y / spheres[l].radius) / Consts<Real>::PI;
                return;
            }
            case SHAPE_PLANE: {
                Vec r = h.position - planes[l].point;
                *u = r.dot(planes[l].tu);
                *v = r.dot(planes[l].tv);
                return;
            }
            default: *u = h.b1; *v = h.b2; return;
        }
    }
};

// Lambertian/Metal::random_unit_vector — lambertian_diffuse.rs:39-45, metal.rs:31-37:
// a = U[0,TAU) is drawn BEFORE z = U[-1,1).
template <class Real>
static inline V3<Real> random_unit_vector(uint32_t key, uint32_t slot) {
    Real ua, ub;
    rng_pair<Real>(key, slot, &ua, &ub);
    Real a = Consts<Real>::TAU * ua;
    Real z = Real(-1.0) + Real(2.0) * ub;
    Real r = std::sqrt(Real(1.0) - z * z);
    return V3<Real>(r * std::cos(a), r * std::sin(a), z);
}

// Metal::reflect / Dielectric::reflect — metal.rs:27-29, dielectric.rs:28-30
template <class Real>
static inline V3<Real> reflect(const V3<Real> &ray, const V3<Real> &normal) {
    return ray - normal * (Real(2.0) * ray.dot(normal));
}

// Dielectric::refract — dielectric.rs:32-45
template <class Real>
static inline bool refract(const V3<Real> &ray, const V3<Real> &normal, Real refraction_index, V3<Real> *out) {
    V3<Real> unit_vector = ray.normalize();
    Real dt = unit_vector.dot(normal);
    Real discriminant = Real(1.0) - refraction_index * refraction_index * (Real(1.0) - dt * dt);
    if (discriminant < Real(0.0)) return false;
    *out = (unit_vector - normal * dt) * refraction_index - normal * std::sqrt(discriminant);
    return true;
}

// ----------------------------------------------------------------------------------------
// Ray::project_ray / _project_ray — ray.rs:42-45,69-87, as the equivalent iterative loop
// (SURVEY §8a row a3). Returns the radiance of one path.
//   T = (1,1,1); for each of max_depth segments: miss -> T*sky; hit -> T *= scatter;
//   bounce None -> T (the surface colour is returned as is, ray.rs:80); else continue.
//   After max_depth segments: T*sky(next ray) (ray.rs:70-72: depth exhaustion returns sky).
// ----------------------------------------------------------------------------------------
// optional per-path trace (debug aid for tests/tools): hit shape and t of each segment
struct PathTrace {
    int n = 0;
    int64_t shape[64];
    double t[64];
    double dlen[64];
};
static thread_local PathTrace *g_trace = nullptr;

template <class Real>
static inline V3<Real> mulv(const V3<Real> &a, const V3<Real> &b) { return V3<Real>(a.x * b.x, a.y * b.y, a.z * b.z); }  // blend, ray.rs:14-18

template <class Real>
static V3<Real> direct_light(const Scene<Real> &scene, const Hit<Real> &hit, const V3<Real> &normal, const Ray<Real> &ray,
                             const typename Scene<Real>::Material &m, rthip_stats *st);

template <class Real>
static V3<Real> project_ray(const Scene<Real> &scene, Ray<Real> ray, uint32_t max_depth, uint32_t key, bool point_lights,
                            rthip_stats *st) {
    using Vec = V3<Real>;
    Vec T(1, 1, 1);
    Vec Ldirect(0, 0, 0);  // extension: accumulated direct light (point-light mode)
    for (uint32_t bounce = 0;; bounce++) {
        if (bounce == max_depth) {  // ray.rs:70-72
            if (st) st->sky_evals++;
            return Ldirect + mulv(T, background_color(ray));
        }
        Hit<Real> hit;
        const bool found = scene.find_collision(ray, &hit, st);
        if (g_trace && g_trace->n < 64) {
            g_trace->shape[g_trace->n] = found ? hit.shape : -1;
            g_trace->t[g_trace->n] = found ? double(hit.t) : 0.0;
            g_trace->dlen[g_trace->n] = std::sqrt(double(ray.direction.magnitude_squared()));
            g_trace->n++;
        }
        if (!found) {  // ray.rs:85
            if (st) st->sky_evals++;
            return Ldirect + mulv(T, background_color(ray));
        }
        const auto &m = scene.materials[scene.material_of(hit.shape)];
        if (st) st->hits[m.kind]++;
        const uint32_t ctr = 1 + bounce;  // RNG slot of this segment
        const Vec normal = scene.normal_at(hit);  // collision.rs:29-31
        const Real light_intensity = Real(3.0);   // "global lightning, to be changed"
        switch (m.kind) {
            case RTHIP_MAT_LAMBERTIAN: {
                // scatter — lambertian_diffuse.rs:53-66
                Real dot_product = std::fmax(Real(0.0), normal.dot(normal));
                Vec scatter = Vec(m.albedo_over_pi.x * Real(1.0), m.albedo_over_pi.y * Real(1.0), m.albedo_over_pi.z * Real(1.0)) *
                              light_intensity * dot_product;
                if (point_lights) Ldirect = Ldirect + mulv(T, direct_light(scene, hit, normal, ray, m, st));
                T = mulv(T, scatter);
                // bounce — lambertian_diffuse.rs:68-73
                ray = Ray<Real>{hit.position, normal + random_unit_vector<Real>(key, ctr)};
                break;
            }
            case RTHIP_MAT_METAL: {
                // scatter — metal.rs:45-56
                Vec scatter = Vec(m.albedo_over_pi.x * Real(1.0), m.albedo_over_pi.y * Real(1.0), m.albedo_over_pi.z * Real(1.0)) * light_intensity;
                if (point_lights) Ldirect = Ldirect + mulv(T, direct_light(scene, hit, normal, ray, m, st));
                T = mulv(T, scatter);
                // bounce — metal.rs:58-66 (the unit vector is drawn even when fuzziness == 0)
                Vec reflected = reflect(ray.direction.normalize(), normal);
                Vec direction = reflected + random_unit_vector<Real>(key, ctr) * m.param;
                if (direction.dot(normal) < Real(0.0)) return Ldirect + T;  // None -> new_color, ray.rs:80
                ray = Ray<Real>{hit.position, direction};
                break;
            }
            case RTHIP_MAT_DIELECTRIC: {
                // scatter — dielectric.rs:56-66
                Vec scatter = Vec(m.albedo_over_pi.x * Real(1.0), m.albedo_over_pi.y * Real(1.0), m.albedo_over_pi.z * Real(1.0)) * light_intensity;
                T = mulv(T, scatter);
                // bounce — dielectric.rs:68-98
                Vec reflected = reflect(ray.direction.normalize(), normal);
                Vec outward_normal;
                Real refraction_index, cosine;
                Real dot_product = ray.direction.dot(normal);
                // `ray.direction().len()` is nalgebra's ELEMENT COUNT (3), not the magnitude: dielectric.rs:78,82
                const Real len = Real(3.0);
                if (dot_product > Real(0.0)) {
                    outward_normal = -normal;
                    refraction_index = m.param;
                    cosine = m.param * dot_product / len;
                } else {
                    outward_normal = normal;
                    refraction_index = Real(1.0) / m.param;
                    cosine = -dot_product / len;
                }
                Vec refracted;
                bool has_refracted = refract(ray.direction, outward_normal, refraction_index, &refracted);
                Real reflect_prob = has_refracted ? schlick(cosine, m.param) : Real(1.0);  // always the material's ior: dielectric.rs:89
                Real xi = unit_from_bits<Real>(rng_first(key, ctr));                        // drawn always: dielectric.rs:94
                ray = Ray<Real>{hit.position, reflect_prob > xi ? reflected : refracted};
                break;
            }
            default: {  // RTHIP_MAT_TEXTURE: scatter texture.rs:45-57, bounce None texture.rs:59-61
                const auto &tex = scene.textures[m.texture];
                Real u, v;
                scene.texture_coords(hit, &u, &v);  // collision.rs:37-39
                uint32_t tx = texture_wrap(u, m.param, tex.width);
                uint32_t ty = texture_wrap(v, m.param, tex.height);
                const uint8_t *px = tex.rgb8 + (size_t(ty) * tex.width + tx) * 3;
                Vec scatter(Real(px[0]) / Real(255.0), Real(px[1]) / Real(255.0), Real(px[2]) / Real(255.0));
                return Ldirect + mulv(T, scatter);  // ray.rs:80
            }
        }
    }
}

// extension (RTHIP_F_POINT_LIGHTS; no reference counterpart): direct illumination of a
// Lambertian (N.L) or Metal (Phong (R.L)^s, only when shininess > 0) hit by every point light,
// with a shadow ray per light. Returned radiance is multiplied by the path throughput BEFORE
// this hit's scatter factor.
template <class Real>
static V3<Real> direct_light(const Scene<Real> &scene, const Hit<Real> &hit, const V3<Real> &normal, const Ray<Real> &ray,
                             const typename Scene<Real>::Material &m, rthip_stats *st) {
    using Vec = V3<Real>;
    Vec sum(0, 0, 0);
    for (const auto &l : scene.lights) {
        Vec to_light = l.position - hit.position;
        Real dist2 = to_light.magnitude_squared();
        Vec ldir = to_light / std::sqrt(dist2);
        Real w;
        if (m.kind == RTHIP_MAT_LAMBERTIAN) {
            w = std::fmax(Real(0.0), normal.dot(ldir));
        } else {
            if (!(m.shininess > Real(0.0))) continue;
            Vec r = reflect(ray.direction.normalize(), normal);
            w = std::pow(std::fmax(Real(0.0), r.dot(ldir)), m.shininess);
        }
        if (!(w > Real(0.0))) continue;
        if (scene.occluded(Ray<Real>{hit.position, to_light}, st)) continue;
        sum = sum + mulv(m.albedo_over_pi, l.intensity) * (w / dist2);
    }
    return sum;
}

// ----------------------------------------------------------------------------------------
// Raytracer::generate_pixel — lib.rs:269-316, for every owned pixel (the batch driver of
// wasm/src/lib.rs:112-123). `sum` is PixelCache.incremental_raw_light (in/out when
// RTHIP_F_ACCUMULATE), prior_samples is PixelCache.nb_samples before this pass.
// crop = {x0, y0, w, h} restricts the pass to a window (CPU-baseline sampling); output
// buffers then hold w*h pixels.
// ----------------------------------------------------------------------------------------
// PixelCache (lib.rs:66-73) of every pixel for the progressive rule of lib.rs:276-313 (RTHIP_F_ADAPTIVE); arrays of
// crop_w*crop_h entries owned by the caller, all zero before the first pass.
struct PixelCaches {
    uint8_t *status;            // GenerationStatus: 0 NotStarted, 2 Unstable, 3 Final
    uint8_t *same_color_count;  // lib.rs:71
    uint8_t *last_color;        // 3 per pixel, valid once status != NotStarted (last_color: Option<PixelColor>)
    uint64_t *nb_samples;       // lib.rs:72 — per pixel: a Final pixel stops counting
};

template <class Real>
static void render(const rthip_scene_desc &desc, const rthip_camera &cam, const rthip_render_params &p, const uint32_t *crop,
                   uint64_t prior_samples, int threads, uint8_t *rgba8, double *sum_io, rthip_stats *stats_out,
                   const PixelCaches *caches = nullptr, int use_accel = 0, double *loop_seconds = nullptr) {
    Scene<Real> scene(desc);
    if (use_accel) scene.build_accel();
    const auto loop_t0 = std::chrono::steady_clock::now();
    const uint32_t x0 = crop ? crop[0] : 0, y0 = crop ? crop[1] : 0;
    const uint32_t cw = crop ? crop[2] : p.width, ch = crop ? crop[3] : p.height;
    const bool accumulate = (p.flags & RTHIP_F_ACCUMULATE) != 0;
    const bool lights = (p.flags & RTHIP_F_POINT_LIGHTS) != 0 && !scene.lights.empty();
    const uint64_t nb_samples_all = (accumulate ? prior_samples : 0) + p.spp;
    const uint32_t sample0_all = accumulate ? uint32_t(prior_samples) : 0;
    uint64_t n_traced = 0, n_final = 0;
    rthip_stats total;
    std::memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        rthip_stats st;
        std::memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 1) reduction(+ : n_traced, n_final)
        for (int64_t ly = 0; ly < int64_t(ch); ly++) {
            const uint32_t y = y0 + uint32_t(ly);
            for (uint32_t lx = 0; lx < cw; lx++) {
                const uint32_t x = x0 + lx;
                const uint32_t pixel_index = y * p.width + x;
                const size_t o = size_t(ly) * cw + lx;
                uint64_t nb_samples = nb_samples_all;
                uint32_t sample0 = sample0_all;
                if (caches) {
                    if (caches->status[o] == 3) {  // lib.rs:277-279: a Final pixel is left alone (its last colour stays on screen)
                        if (rgba8) {
                            rgba8[o * 4 + 0] = caches->last_color[o * 3 + 0], rgba8[o * 4 + 1] = caches->last_color[o * 3 + 1];
                            rgba8[o * 4 + 2] = caches->last_color[o * 3 + 2], rgba8[o * 4 + 3] = 255;
                        }
                        n_final++;
                        continue;
                    }
                    sample0 = accumulate ? uint32_t(caches->nb_samples[o]) : 0;
                    nb_samples = (accumulate ? caches->nb_samples[o] : 0) + p.spp;  // lib.rs:293
                    caches->nb_samples[o] = nb_samples;
                }
                V3<Real> samples_color(0, 0, 0);
                for (uint32_t s = 0; s < p.spp; s++) {
                    const uint32_t key = path_key(p.seed, pixel_index, sample0 + s);
                    // lib.rs:282-285: x jitter first, divisors are (width - 1), (height - 1)
                    Real jx, jy;
                    rng_pair<Real>(key, 0, &jx, &jy);
                    Real ox = (Real(x) + jx) / (Real(p.width) - Real(1.0));
                    Real oy = (Real(y) + jy) / (Real(p.height) - Real(1.0));
                    Ray<Real> r = emit_ray_at<Real>(cam, ox, oy);
                    samples_color = samples_color + project_ray(scene, r, p.max_depth, key, lights, &st);  // lib.rs:287
                    st.paths++;
                }
                if (accumulate && sum_io) {  // lib.rs:289-291
                    samples_color = samples_color + V3<Real>(Real(sum_io[o * 3 + 0]), Real(sum_io[o * 3 + 1]), Real(sum_io[o * 3 + 2]));
                }
                if (sum_io) { sum_io[o * 3 + 0] = double(samples_color.x); sum_io[o * 3 + 1] = double(samples_color.y); sum_io[o * 3 + 2] = double(samples_color.z); }
                const uint8_t col[3] = {tonemap_channel(samples_color.x, nb_samples), tonemap_channel(samples_color.y, nb_samples),
                                        tonemap_channel(samples_color.z, nb_samples)};
                if (rgba8) {
                    rgba8[o * 4 + 0] = col[0], rgba8[o * 4 + 1] = col[1], rgba8[o * 4 + 2] = col[2];
                    rgba8[o * 4 + 3] = 255;
                }
                if (caches) {  // lib.rs:300-313
                    n_traced++;
                    if (caches->status[o] != 0) {  // last_color is Some
                        const uint8_t *last = &caches->last_color[o * 3];
                        if (last[0] == col[0] && last[1] == col[1] && last[2] == col[2]) caches->same_color_count[o] += 1;
                        else caches->same_color_count[o] = 0;
                    } else {
                        caches->status[o] = 2;  // Unstable
                    }
                    if (caches->same_color_count[o] > 3) {  // MAX_SIMILAR_SAMPLE_FOR_A_PIXEL
                        caches->status[o] = 3;
                        n_final++;
                    }
                    caches->last_color[o * 3 + 0] = col[0], caches->last_color[o * 3 + 1] = col[1], caches->last_color[o * 3 + 2] = col[2];
                }
            }
        }
#pragma omp critical
        {
            total.paths += st.paths; total.rays += st.rays; total.sphere_tests += st.sphere_tests;
            total.plane_tests += st.plane_tests; total.tri_tests += st.tri_tests;
            for (int k = 0; k < 4; k++) total.hits[k] += st.hits[k];
            total.sky_evals += st.sky_evals; total.shadow_rays += st.shadow_rays;
        }
    }
    if (loop_seconds) *loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_t0).count();
    total.rows = ch;
    total.nb_samples = nb_samples_all;
    total.pixels_traced = caches ? n_traced : uint64_t(cw) * ch;
    total.pixels_final = n_final;
    if (stats_out) *stats_out = total;
}

}  // namespace oracle
