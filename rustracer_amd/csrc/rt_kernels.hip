// rt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the rustracer render path.
//
// One lane owns one pixel; a wave owns an 8x8 pixel tile (coherent primary rays => coherent
// branches), a 256-thread workgroup a 16x16 tile. The recursion of Ray::_project_ray
// (raytracer_core/src/shapes/ray.rs:69-87) is an iterative bounce loop with the path state
// (ray, throughput, bounce count) in VGPRs, and the `samples` loop of generate_pixel
// (lib.rs:281-288) is folded into the same loop by path regeneration: a lane whose path ended
// starts its pixel's next sample at once, so a wave only idles lanes at the very end of a tile.
// Scene data indexed by a wave-uniform loop counter (the brute-force closest-hit scan of
// ray.rs:47-67) is fetched through the scalar cache into SGPRs, i.e. it costs no VALU/LDS work.
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>

#include "../../include/rthip.h"
#include "rt_device.h"
#include "rt_rng.h"

#define RT_NONE 0xffffffffu

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

struct HitRec {
    float t;
    uint32_t shape;  // global shape index (spheres, then planes, then triangles); RT_NONE = miss
    float b1, b2;    // triangle barycentrics
};

// ---- primitive tests -------------------------------------------------------------------
// Sphere::collide — sphere.rs:29-61. g = {cx, cy, cz, r*r}; a = d.d, inv_a = 1/a are per-segment.
// `on_surface`: the ray starts ON this sphere (it is the shape the path just bounced off). In the
// reference's f64, c = |oc|^2 - r^2 is then ~1e-13, i.e. the roots are 0 and -2*half_b/a. In FP32
// the same c carries ~1e-3 of cancellation error on the radius-100 ground sphere (|oc|^2 ~ 1e4),
// which turns ~0.1 % of grazing bounces into spurious self-hits beyond T_MIN; c is therefore set
// to its exact value 0 for that one sphere ("FP32 robustness").
__device__ __forceinline__ bool hit_sphere(rt_f4 g, f3 o, f3 d, float a, float inv_a, float t_min, float t_max, bool on_surface, float &t_out) {
    f3 oc = mk3(o.x - g.x, o.y - g.y, o.z - g.z);
    float half_b = dot(oc, d);
    float c = on_surface ? 0.0f : dot(oc, oc) - g.w;
    float disc = fmaf(half_b, half_b, -(a * c));
    float root = fast_sqrt(disc);  // NaN when disc < 0: every comparison below is then false
    float t0 = (-half_b - root) * inv_a;
    float t1 = (root - half_b) * inv_a;
    bool ok0 = t0 < t_max && t0 > t_min;
    bool ok1 = t1 < t_max && t1 > t_min;
    t_out = ok0 ? t0 : t1;
    return disc > 0.0f && (ok0 || ok1);
}
// extension: plane. g0 = {nx, ny, nz, n.p0}
__device__ __forceinline__ bool hit_plane(rt_f4 g0, f3 o, f3 d, float t_min, float t_max, float &t_out) {
    f3 n = mk3(g0.x, g0.y, g0.z);
    float denom = dot(d, n);
    float t = (g0.w - dot(o, n)) * fast_rcp(denom);
    t_out = t;
    return denom != 0.0f && t < t_max && t > t_min;
}
// extension: Moller-Trumbore, two-sided. v0, e1 = v1-v0, e2 = v2-v0
__device__ __forceinline__ bool hit_tri(rt_f4 v0, rt_f4 e1_, rt_f4 e2_, f3 o, f3 d, float t_min, float t_max, float &t_out, float &b1_out,
                                        float &b2_out) {
    f3 e1 = mk3(e1_.x, e1_.y, e1_.z), e2 = mk3(e2_.x, e2_.y, e2_.z);
    f3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    float inv = fast_rcp(det);
    f3 tvec = mk3(o.x - v0.x, o.y - v0.y, o.z - v0.z);
    float b1 = dot(tvec, pvec) * inv;
    f3 qvec = cross(tvec, e1);
    float b2 = dot(d, qvec) * inv;
    float t = dot(e2, qvec) * inv;
    t_out = t, b1_out = b1, b2_out = b2;
    return det != 0.0f && b1 >= 0.0f && b1 <= 1.0f && b2 >= 0.0f && (b1 + b2) <= 1.0f && t < t_max && t > t_min;
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

// ---- linear BVH traversal (extension; big scenes) --------------------------------------------
// Per-lane stack in LDS ([level][thread]: conflict-free), near child first by the split axis.
// A node is entered iff the ray's slab interval overlaps (T_MIN, min(best t, t_max)] — INCLUSIVE of
// the current best t, so that an equal-t shape with a larger index is still found (ray.rs:55-61).
#define RT_STACK_AT(level) s_stack[(level) * 256 + threadIdx.x]

template <bool STATS, bool ANY>
__device__ __forceinline__ bool bvh_traverse(const KernelParams &P, f3 o, f3 d, float a, float inv_a, float t_max, uint32_t self, HitRec &h,
                                             uint32_t *cnt, uint32_t *s_stack) {
    const f3 inv_d = mk3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
    const uint32_t tri_base = P.n_spheres + P.n_planes;
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        const rt_f4 n0 = ((const rt_f4 *)P.bvh_nodes)[2 * node], n1 = ((const rt_f4 *)P.bvh_nodes)[2 * node + 1];
        if (STATS) cnt[RT_STAT_NODE]++;
        const float ax = (n0.x - o.x) * inv_d.x, bx = (n1.x - o.x) * inv_d.x;
        const float ay = (n0.y - o.y) * inv_d.y, by = (n1.y - o.y) * inv_d.y;
        const float az = (n0.z - o.z) * inv_d.z, bz = (n1.z - o.z) * inv_d.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)) * 1.0000005f;
        const float limit = ANY ? t_max : fminf(h.t, t_max);
        if (tn <= tf && tn <= limit && tf >= RT_T_MIN) {
            const uint32_t first = __float_as_uint(n0.w), cw = __float_as_uint(n1.w);
            const uint32_t count = cw & RT_BVH_COUNT_MASK;
            if (count) {
                for (uint32_t k = 0; k < count; k++) {
                    const uint32_t id = P.bvh_prims[first + k];
                    if (id < P.n_bvh_spheres) {
                        float t;
                        bool ok = hit_sphere(P.sph_geom[id], o, d, a, inv_a, RT_T_MIN, t_max, id == self, t);
                        if (STATS) cnt[RT_STAT_SPHERE]++;
                        if (ANY) {
                            if (ok) return true;
                        } else {
                            RT_TAKE_ANY_ORDER(ok, t, id, (void)0)
                        }
                    } else {
                        const uint32_t ti = id - P.n_bvh_spheres;
                        float t, b1, b2;
                        bool ok = hit_tri(P.tri_geom[3 * ti], P.tri_geom[3 * ti + 1], P.tri_geom[3 * ti + 2], o, d, RT_T_MIN, t_max, t, b1, b2);
                        if (STATS) cnt[RT_STAT_TRI]++;
                        if (ANY) {
                            if (ok) return true;
                        } else {
                            RT_TAKE_ANY_ORDER(ok, t, tri_base + ti, (h.b1 = b1, h.b2 = b2))
                        }
                    }
                }
            } else {
                const uint32_t axis = cw >> RT_BVH_AXIS_SHIFT;
                const float da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
                const uint32_t left = node + 1, right = first;
                const bool right_first = da < 0.0f;
                RT_STACK_AT(sp) = right_first ? left : right;
                sp++;
                node = right_first ? right : left;
                continue;
            }
        }
        if (sp == 0) break;
        sp--;
        node = RT_STACK_AT(sp);
    }
    return false;
}

template <bool STATS, bool BVH>
__device__ __forceinline__ HitRec closest_hit(const KernelParams &P, f3 o, f3 d, uint32_t self, uint32_t *cnt, uint32_t *s_stack) {
    HitRec h;
    h.t = __builtin_inff();
    h.shape = RT_NONE;
    h.b1 = h.b2 = 0.0f;
    const float a = dot(d, d);
    const float inv_a = fast_rcp(a);
    if (!BVH) {
        for (uint32_t i = 0; i < P.n_spheres; i++) {  // wave-uniform index: scalar loads
            float t;
            bool ok = hit_sphere(P.sph_geom[i], o, d, a, inv_a, RT_T_MIN, RT_T_MAX, i == self, t);
            RT_TAKE(ok, t, i, (void)0)
        }
    }
    for (uint32_t i = 0; i < P.n_planes; i++) {
        float t;
        bool ok = hit_plane(P.pln_geom[4 * i], o, d, RT_T_MIN, RT_T_MAX, t);
        RT_TAKE(ok, t, P.n_spheres + i, (void)0)
    }
    if (!BVH) {
        for (uint32_t i = 0; i < P.n_tris; i++) {
            float t, b1, b2;
            bool ok = hit_tri(P.tri_geom[3 * i], P.tri_geom[3 * i + 1], P.tri_geom[3 * i + 2], o, d, RT_T_MIN, RT_T_MAX, t, b1, b2);
            RT_TAKE(ok, t, P.n_spheres + P.n_planes + i, (h.b1 = b1, h.b2 = b2))
        }
        if (STATS) {
            cnt[RT_STAT_SPHERE] += P.n_spheres;
            cnt[RT_STAT_TRI] += P.n_tris;
        }
    } else {
        bvh_traverse<STATS, false>(P, o, d, a, inv_a, RT_T_MAX, self, h, cnt, s_stack);
    }
    if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
    return h;
}

// extension: shadow-ray any-hit for the point-light mode: direction = light - p, occluded iff some
// shape is hit with T_MIN < t < 1.
template <bool STATS, bool BVH>
__device__ __forceinline__ bool occluded(const KernelParams &P, f3 o, f3 d, uint32_t self, uint32_t *cnt, uint32_t *s_stack) {
    const float a = dot(d, d);
    const float inv_a = fast_rcp(a);
    bool occ = false;
    float t, b1, b2;
    if (!BVH) {
        for (uint32_t i = 0; i < P.n_spheres; i++) occ |= hit_sphere(P.sph_geom[i], o, d, a, inv_a, RT_T_MIN, 1.0f, i == self, t);
        for (uint32_t i = 0; i < P.n_tris; i++)
            occ |= hit_tri(P.tri_geom[3 * i], P.tri_geom[3 * i + 1], P.tri_geom[3 * i + 2], o, d, RT_T_MIN, 1.0f, t, b1, b2);
        if (STATS) {
            cnt[RT_STAT_SPHERE] += P.n_spheres;
            cnt[RT_STAT_TRI] += P.n_tris;
        }
    }
    for (uint32_t i = 0; i < P.n_planes; i++) occ |= hit_plane(P.pln_geom[4 * i], o, d, RT_T_MIN, 1.0f, t);
    if (STATS) cnt[RT_STAT_PLANE] += P.n_planes;
    if (BVH && !occ) {
        HitRec dummy;
        dummy.t = 1.0f;
        dummy.shape = RT_NONE;
        occ = bvh_traverse<STATS, true>(P, o, d, a, inv_a, 1.0f, self, dummy, cnt, s_stack);
    }
    return occ;
}

// Ray::background_color — ray.rs:89-93
__device__ __forceinline__ f3 sky_color(f3 d) {
    float t = 0.5f * (d.y * fast_rsq(dot(d, d)) + 1.0f);
    float w = 1.0f - t;
    return mk3(fmaf(0.5f, t, w), fmaf(0.7f, t, w), fmaf(1.0f, t, w));
}

// Lambertian/Metal::random_unit_vector — lambertian_diffuse.rs:39-45, metal.rs:31-37 (a first, then z).
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so a = TAU*xi needs no multiply.
__device__ __forceinline__ f3 random_unit_vector(uint32_t h) {
    float xa = rt_unit(h);
    float z = fmaf(2.0f, rt_unit(rt_second(h)), -1.0f);
    float r = fast_sqrt(fmaf(-z, z, 1.0f));
    return mk3(r * __builtin_amdgcn_cosf(xa), r * __builtin_amdgcn_sinf(xa), z);
}

// Texture::wrap — texture.rs:32-41 (`as i32`: truncate, saturate, NaN -> 0)
__device__ __forceinline__ uint32_t texture_wrap(float val, float scale, uint32_t bound) {
    float fc = val * scale * float(bound);
    int32_t ic = (fc == fc) ? int32_t(fminf(fmaxf(fc, -2147483648.0f), 2147483520.0f)) : 0;
    if (fc >= 2147483648.0f) ic = 2147483647;
    int32_t w = ic % int32_t(bound);
    return uint32_t(w < 0 ? w + int32_t(bound) : w);
}

__device__ __forceinline__ float acos_f32(float x) { return acosf(x); }

// One full pass over a pixel: generate_pixel (lib.rs:269-316) with project_ray (ray.rs:42-87) inlined.
template <bool STATS, bool BVH, bool LIGHTS>
__global__ __launch_bounds__(256) void render_kernel(const KernelParams P) {
    __shared__ uint32_t s_stack_mem[BVH ? RT_BVH_MAX_DEPTH * 256 : 1];
    uint32_t *s_stack = s_stack_mem;
    const uint32_t tiles_x = (P.width + 15u) >> 4;
    const uint32_t tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = (tile_x << 4) + ((wave & 1u) << 3) + (lane & 7u);
    const uint32_t lrow = (tile_y << 4) + ((wave >> 1) << 3) + (lane >> 3);
    // owned local row -> image row y (row bands interleaved over contexts)
    uint32_t y = lrow;
    if (P.band_rows) {
        uint32_t k = lrow / P.band_rows;
        y = (P.band_first + k * P.band_stride) * P.band_rows + (lrow - k * P.band_rows);
    }
    const bool active = x < P.width && lrow < P.owned_rows && y < P.height;

    uint32_t cnt[RT_STAT_N];
#pragma unroll
    for (int i = 0; i < RT_STAT_N; i++) cnt[i] = 0;

    if (active) {
        const uint32_t pixel_index = y * P.width + x;
        const uint32_t pkey = rt_pixel_key(rt_seed_key(P.seed), pixel_index);
        const f3 cam_o = mk3(P.ox, P.oy, P.oz);
        f3 acc = mk3(0.0f, 0.0f, 0.0f);
        uint32_t s = 0;
        uint32_t key, bounce;
        uint32_t self = RT_NONE;  // shape the current ray starts on
        f3 o, d, T;
        f3 Ld = mk3(0.0f, 0.0f, 0.0f);  // extension: direct light gathered along the path (point-light mode)
        auto new_path = [&]() {
            key = rt_counter(pkey, P.sample0 + s, 0);  // RNG counter of this path's slot 0
            // lib.rs:282-285: x jitter first
            const uint32_t hj = rt_mix32(key);
            float u = (float(x) + rt_unit(hj)) * P.inv_wm1;
            float v = (float(y) + rt_unit(rt_second(hj))) * P.inv_hm1;
            o = cam_o;  // camera.rs:43-50
            d = mk3(fmaf(P.vx, v, fmaf(P.hx, u, P.d00x)), fmaf(P.vy, v, fmaf(P.hy, u, P.d00y)), fmaf(P.vz, v, fmaf(P.hz, u, P.d00z)));
            T = mk3(1.0f, 1.0f, 1.0f);
            bounce = 0;
            self = RT_NONE;
            if (LIGHTS) Ld = mk3(0.0f, 0.0f, 0.0f);
        };
        new_path();
        for (;;) {
            f3 L;
            bool done;
            HitRec h;
            h.shape = RT_NONE;
            if (bounce != P.max_depth) {  // ray.rs:70-72: depth exhausted -> sky without a scene query
                h = closest_hit<STATS, BVH>(P, o, d, self, cnt, s_stack);
                cnt[RT_STAT_RAYS]++;
            }
            if (h.shape == RT_NONE) {  // ray.rs:85
                L = T * sky_color(d);
                done = true;
                if (STATS) cnt[RT_STAT_SKY]++;
            } else {
                const f3 p = fma3(d, h.t, o);  // ray.rs:30-32
                // ---- shape-specific: normal, material, texture coordinates (sphere.rs:63-73) ----
                f3 n;
                uint32_t mat;
                float tu = 0.0f, tv = 0.0f;
                if (h.shape < P.n_spheres) {
                    rt_f4 g = P.sph_geom[h.shape];
                    rt_f4 sh = P.sph_shade[h.shape];
                    f3 pc = mk3(p.x - g.x, p.y - g.y, p.z - g.z);
                    n = pc * sh.x;  // (p - c)/r
                    mat = __float_as_uint(sh.y);
                    tu = pc.z, tv = pc.x;  // stash for atan2; resolved below only for textures
                    h.b1 = pc.y * sh.x;    // n.y/r... = (p-c).y / r
                } else if (h.shape < P.n_spheres + P.n_planes) {
                    const uint32_t i = h.shape - P.n_spheres;
                    rt_f4 g0 = P.pln_geom[4 * i], g1 = P.pln_geom[4 * i + 1];
                    n = mk3(g0.x, g0.y, g0.z);
                    mat = __float_as_uint(g1.w);
                } else {
                    const uint32_t i = h.shape - P.n_spheres - P.n_planes;
                    rt_f4 e1 = P.tri_geom[3 * i + 1], e2 = P.tri_geom[3 * i + 2];
                    n = normalize(cross(mk3(e1.x, e1.y, e1.z), mk3(e2.x, e2.y, e2.z)));
                    mat = __float_as_uint(P.tri_geom[3 * i].w);
                }
                const DevMaterial m = P.materials[mat];
                const uint32_t hb = rt_mix32(key + (1u + bounce) * RT_RNG_GOLDEN);  // RNG slot 1 + bounce
                if (STATS) {
#pragma unroll
                    for (uint32_t kk = 0; kk < 4; kk++) cnt[RT_STAT_HIT0 + kk] += (m.kind == kk) ? 1u : 0u;
                }
                if (m.kind == RTHIP_MAT_TEXTURE) {
                    // Texture::scatter — texture.rs:45-57; bounce -> None (texture.rs:59-61) => path ends, ray.rs:80
                    float U, V;
                    if (h.shape < P.n_spheres) {
                        U = (1.0f + atan2f(tu, tv) * 0.31830988618379067f) * 0.5f;  // sphere.rs:70
                        V = acos_f32(h.b1) * 0.31830988618379067f;                  // sphere.rs:71
                    } else if (h.shape < P.n_spheres + P.n_planes) {
                        const uint32_t i = h.shape - P.n_spheres;
                        rt_f4 g1 = P.pln_geom[4 * i + 1], g2 = P.pln_geom[4 * i + 2], g3 = P.pln_geom[4 * i + 3];
                        f3 r = mk3(p.x - g1.x, p.y - g1.y, p.z - g1.z);
                        U = dot(r, mk3(g2.x, g2.y, g2.z));
                        V = dot(r, mk3(g3.x, g3.y, g3.z));
                    } else {
                        U = h.b1, V = h.b2;
                    }
                    const DevTexInfo ti = P.tex_info[m.texture];
                    uint32_t txl = texture_wrap(U, m.param, ti.width);
                    uint32_t tyl = texture_wrap(V, m.param, ti.height);
                    uint32_t texel = P.texels[ti.first_texel + (((tyl >> 3) * ti.tiles_per_row + (txl >> 3)) << 6) + ((tyl & 7u) << 3) + (txl & 7u)];
                    f3 c = mk3(float(texel & 255u), float((texel >> 8) & 255u), float((texel >> 16) & 255u)) * (1.0f / 255.0f);
                    L = T * c;
                    done = true;
                } else {
                    if (LIGHTS && m.kind != RTHIP_MAT_DIELECTRIC) {
                        // extension (RTHIP_F_POINT_LIGHTS): Lambert N.L on Lambertian, Phong (R.L)^s on Metal, one shadow ray per light
                        f3 sum = mk3(0.0f, 0.0f, 0.0f);
                        const f3 dhl = normalize(d);
                        const f3 rl = fma3(n, -2.0f * dot(dhl, n), dhl);
                        for (uint32_t li = 0; li < P.n_lights; li++) {
                            const DevLight lt = P.lights[li];
                            const f3 to_light = mk3(lt.px - p.x, lt.py - p.y, lt.pz - p.z);
                            const float dist2 = dot(to_light, to_light);
                            const f3 ldir = to_light * fast_rsq(dist2);
                            float w;
                            if (m.kind == RTHIP_MAT_LAMBERTIAN)
                                w = fmaxf(0.0f, dot(n, ldir));
                            else
                                w = (m.shininess > 0.0f) ? __powf(fmaxf(0.0f, dot(rl, ldir)), m.shininess) : 0.0f;
                            if (w > 0.0f) {
                                cnt[RT_STAT_RAYS]++;
                                if (STATS) cnt[RT_STAT_SHADOW]++;
                                if (!occluded<STATS, BVH>(P, p, to_light, h.shape, cnt, s_stack)) {
                                    const float s = w * fast_rcp(dist2);
                                    sum = sum + mk3(m.ar * lt.ir, m.ag * lt.ig, m.ab * lt.ib) * s;
                                }
                            }
                        }
                        Ld = Ld + T * sum;
                    }
                    T = T * mk3(m.tr, m.tg, m.tb);  // scatter: albedo/pi * 3 (lambertian_diffuse.rs:53-66, metal.rs:45-56, dielectric.rs:56-66)
                    f3 nd;
                    done = false;
                    if (m.kind == RTHIP_MAT_LAMBERTIAN) {
                        nd = n + random_unit_vector(hb);  // lambertian_diffuse.rs:68-73
                    } else {
                        const f3 dh = normalize(d);
                        const f3 refl = fma3(n, -2.0f * dot(dh, n), dh);  // metal.rs:27-29, dielectric.rs:28-30
                        if (m.kind == RTHIP_MAT_METAL) {
                            nd = fma3(random_unit_vector(hb), m.param, refl);  // metal.rs:59-60
                            if (dot(nd, n) < 0.0f) {                                   // metal.rs:62-64 -> None -> ray.rs:80
                                L = T;
                                done = true;
                            }
                        } else {  // dielectric.rs:68-98
                            const float dp = dot(d, n);
                            const bool inside = dp > 0.0f;
                            const f3 on = inside ? -n : n;
                            const float eta = inside ? m.param : m.inv_ior;
                            // `direction().len()` is the element count 3: dielectric.rs:78,82
                            const float cosine = (inside ? m.param * dp : -dp) * (1.0f / 3.0f);
                            const float dt = dot(dh, on);
                            const float k = fmaf(-eta * eta, fmaf(-dt, dt, 1.0f), 1.0f);  // dielectric.rs:39
                            const float om = 1.0f - cosine;
                            const float om2 = om * om;
                            const float prob = (k < 0.0f) ? 1.0f : fmaf(1.0f - m.r1, om2 * om2 * om, m.r1);  // dielectric.rs:47-52,87-92
                            const float xi = rt_unit(hb);                                             // dielectric.rs:94
                            if (prob > xi) {
                                nd = refl;
                            } else {
                                const f3 perp = fma3(on, -dt, dh);  // dielectric.rs:44
                                nd = fma3(on, -fast_sqrt(k), perp * eta);
                            }
                        }
                    }
                    o = p;
                    d = nd;
                    self = h.shape;
                    bounce++;
                }
            }
            if (done) {
                if (LIGHTS) L = L + Ld;
                acc = acc + L;  // lib.rs:287
                cnt[RT_STAT_PATHS]++;
                s++;
                if (s == P.spp) break;
                new_path();
            }
        }
        // lib.rs:289-299: add the prior running sum, store it, tonemap
        const size_t oi = size_t(lrow) * P.width + x;
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
        if (P.rgba8) {
            // mean, clamp(0,1), sqrt, *255, `as u8` (truncate, saturate, NaN -> 0): lib.rs:294-299,30-38
            float r = __fsqrt_rn(fminf(fmaxf(acc.x * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
            float g = __fsqrt_rn(fminf(fmaxf(acc.y * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
            float b = __fsqrt_rn(fminf(fmaxf(acc.z * P.inv_nb, 0.0f), 1.0f)) * 255.0f;
            P.rgba8[oi] = uint32_t(r) | (uint32_t(g) << 8) | (uint32_t(b) << 16) | 0xff000000u;
        }
    }

    // per-wave reduction of the counters, one atomic per wave and counter into a shard
#pragma unroll
    for (int i = 0; i < RT_STAT_N; i++) {
        if (!STATS && i > RT_STAT_RAYS) break;
        uint32_t v = cnt[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(&P.stats[(blockIdx.x & (RT_STAT_SHARDS - 1)) * RT_STAT_N + i], (unsigned long long)v);
    }
}

// Raytracer::get_shape — lib.rs:262-267: one un-jittered primary ray, index of the closest shape.
__global__ void pick_kernel(const KernelParams P, float u, float v, long long *out) {
    __shared__ uint32_t s_stack[RT_BVH_MAX_DEPTH * 256];
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t cnt[RT_STAT_N];
    f3 o = mk3(P.ox, P.oy, P.oz);
    f3 d = mk3(fmaf(P.vx, v, fmaf(P.hx, u, P.d00x)), fmaf(P.vy, v, fmaf(P.hy, u, P.d00y)), fmaf(P.vz, v, fmaf(P.hz, u, P.d00z)));
    HitRec h = P.n_bvh_nodes ? closest_hit<false, true>(P, o, d, RT_NONE, cnt, s_stack) : closest_hit<false, false>(P, o, d, RT_NONE, cnt, s_stack);
    *out = (h.shape == RT_NONE) ? -1ll : (long long)h.shape;
}

// ---- launch wrappers (called from rthip_api.cpp) -----------------------------------------
template <bool STATS, bool BVH>
static void launch_variant(const KernelParams &P, bool lights, uint32_t blocks, hipStream_t stream) {
    if (lights)
        hipLaunchKernelGGL((render_kernel<STATS, BVH, true>), dim3(blocks), dim3(256), 0, stream, P);
    else
        hipLaunchKernelGGL((render_kernel<STATS, BVH, false>), dim3(blocks), dim3(256), 0, stream, P);
}

extern "C" hipError_t rt_launch_render(const KernelParams *P, int count_stats, hipStream_t stream) {
    const uint32_t tiles_x = (P->width + 15u) >> 4, tiles_y = (P->owned_rows + 15u) >> 4;
    const uint32_t blocks = tiles_x * tiles_y;
    if (blocks == 0) return hipSuccess;
    const bool lights = (P->flags & RTHIP_F_POINT_LIGHTS) && P->n_lights > 0;
    const bool bvh = P->n_bvh_nodes > 0;
    if (count_stats) {
        if (bvh) launch_variant<true, true>(*P, lights, blocks, stream);
        else launch_variant<true, false>(*P, lights, blocks, stream);
    } else {
        if (bvh) launch_variant<false, true>(*P, lights, blocks, stream);
        else launch_variant<false, false>(*P, lights, blocks, stream);
    }
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_pick(const KernelParams *P, float u, float v, long long *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(64), 0, stream, *P, u, v, d_out);
    return hipGetLastError();
}
