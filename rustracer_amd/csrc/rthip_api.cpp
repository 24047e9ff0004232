// rthip_api.cpp — host side of the C ABI declared in include/rthip.h: context, scene flattening
// (Scene = Vec<Box<dyn Shape>> -> SoA device buffers), camera upload, render-pass launch,
// running-sum bookkeeping (PixelCache.incremental_raw_light / nb_samples, lib.rs:67-73,289-294).
// No exception leaves an entry point.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rthip.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_prepass.h"

extern "C" hipError_t rt_launch_tile_order(const KernelParams *P, uint32_t *order_and_cost, const uint32_t *measured, uint32_t *split, uint32_t slots,
                                           uint32_t measured_slices, hipStream_t stream);
extern "C" hipError_t rt_launch_prepass(const KernelParams *P, void *tile_pre, hipStream_t stream);
extern "C" hipError_t rt_launch_render(const KernelParams *P, int count_stats, hipStream_t stream);
extern "C" hipError_t rt_launch_compact(const KernelParams *P, uint32_t *list, uint32_t *count, hipStream_t stream);
extern "C" hipError_t rt_launch_pick(const KernelParams *P, float u, float v, long long *d_out, hipStream_t stream);
extern "C" hipError_t rt_scratch_init(int device);
extern "C" void *rt_scratch_ptr(int device);

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct rthip_ctx {
    int device = 0;
    std::string err;
    DevBuf sph_geom, sph_shade, pln_geom, tri_geom, materials, texels, tex_info, lights, bvh_nodes, bvh_recs;
    DevBuf stats, accum, staging_rgba, staging_accum, pick_out, pix_state, worklist, partial, tile_pre, tile_order, tile_cost;
    KernelParams base;
    bool have_scene = false, have_camera = false;
    rthip_camera camera;
    // running-sum state (lib.rs:67-73)
    uint64_t nb_samples = 0;
    bool adaptive = false;  // the running sum was built by RTHIP_F_ADAPTIVE passes (pix_state is live)
    uint32_t acc_width = 0, acc_height = 0, acc_rows = 0, acc_band_rows = 0, acc_band_first = 0, acc_band_stride = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t stream = nullptr;  // used by the synchronous entry points
    uint64_t n_shapes = 0;
    hipEvent_t ev_last = nullptr;   // recorded after every enqueued pass (passes on different streams are ordered through it)
    hipStream_t last_stream = nullptr;
    bool have_last_pass = false;
    bool pending_stats = false, pending_adaptive = false;  // a counting pass has been enqueued and not yet collected
    uint32_t pending_rows = 0, pending_width = 0;
    unsigned long long *stats_host = nullptr;  // page-locked landing buffer of the counter shards (no staging allocation per pass)
    void *scratch = nullptr;                   // this device's copy of the static scratch array (rthip_render_device_noalloc)
    const void *acc_buffer = nullptr;          // where the running sum lives: ctx->accum.p, or the caller's d_accum (no-alloc passes)
    // rthip_render_async / rthip_wait: two frames in flight (the copy of one overlaps the pass of the next)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_rendered[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    DevBuf staging_async[2];
    bool async_busy[2] = {false, false};
    // Cost-ordered launches (render_pass): what the order in tile_order was made for, and from what. 0 = nothing; 1 = from the cost
    // model, and the pass that used it MEASURED the groups into tile_cost; 2 = from those measurements (later passes just reuse it).
    struct OrderKey {
        uint32_t width, height, rows, band_rows, band_first, band_stride, spp, max_depth, flags;
        uint64_t frame;
        bool operator==(const OrderKey &o) const {
            return width == o.width && height == o.height && rows == o.rows && band_rows == o.band_rows && band_first == o.band_first &&
                   band_stride == o.band_stride && spp == o.spp && max_depth == o.max_depth && flags == o.flags && frame == o.frame;
        }
    } order_key = {};
    int order_state = 0;
    uint32_t order_meas_slices = 1;            // slices of the pass that measured
    uint32_t order_head = 0, order_whole_end = 0;  // the mixed launch of the measured order (KernelParams::head_groups / whole_end)
    uint32_t wave_slots = 7168;                // resident waves of the device at 7 per SIMD
    uint64_t frame_gen = 1;                    // advances with every scene upload and camera change
    const void *pre_lists = nullptr;           // the primary-visibility lists (rt_prepass.h) made for the running sum's frame, if any:
                                               // RTHIP_F_ACCUMULATE passes of the same frame reuse them (same camera, scene, layout)
};

namespace {

thread_local std::string g_err;  // errors with no context (create failures)

int fail(rthip_ctx *ctx, int status, const std::string &msg) {
    if (ctx) ctx->err = msg;
    g_err = msg;
    return status;
}

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e__ = (call);                                                                               \
        if (e__ != hipSuccess)                                                                                 \
            return fail(ctx, e__ == hipErrorOutOfMemory ? RTHIP_E_OOM : RTHIP_E_HIP,                           \
                        std::string(#call) + ": " + hipGetErrorString(e__));                                  \
    } while (0)

int upload(rthip_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
    if (bytes > b.bytes || (bytes == 0 && b.p)) {
        b.release();
        if (bytes) {
            HIPCHK(ctx, hipMalloc(&b.p, bytes));
            b.bytes = bytes;
        }
    }
    if (bytes) HIPCHK(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RTHIP_OK;
}

int ensure(rthip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes > b.bytes) {
        b.release();
        HIPCHK(ctx, hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    return RTHIP_OK;
}

inline rt_f4 f4(double x, double y, double z, double w) { return rt_f4{float(x), float(y), float(z), float(w)}; }
inline float bits_f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

struct D3 {
    double x, y, z;
};
inline D3 d3(const double *p) { return D3{p[0], p[1], p[2]}; }
inline D3 sub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 mul(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline D3 normalize(D3 a) { return mul(a, 1.0 / std::sqrt(dot(a, a))); }

// coordinates must survive the conversion to FP32 with their squares finite (|oc|^2, r^2): |x| <= 1e18
constexpr double RT_COORD_MAX = 1e18;
bool finite1(double v) { return std::isfinite(v) && std::fabs(v) <= RT_COORD_MAX; }
bool finite3(const double *p) { return finite1(p[0]) && finite1(p[1]) && finite1(p[2]); }

int owned_rows(const rthip_render_params *p, uint32_t *rows) {
    if (!p || !rows) return RTHIP_E_BADARG;
    if (p->band_rows == 0) {
        *rows = p->height;
        return RTHIP_OK;
    }
    if (p->band_stride == 0) return RTHIP_E_BADARG;
    const uint64_t nbands = (uint64_t(p->height) + p->band_rows - 1) / p->band_rows;
    uint64_t r = 0;
    for (uint64_t b = p->band_first; b < nbands; b += p->band_stride) {
        uint64_t lo = b * p->band_rows, hi = lo + p->band_rows;
        if (hi > p->height) hi = p->height;
        r += hi - lo;
    }
    *rows = uint32_t(r);
    return RTHIP_OK;
}

int validate_params(rthip_ctx *ctx, const rthip_render_params *p) {
    if (!p) return fail(ctx, RTHIP_E_BADARG, "params is NULL");
    if (p->width < 2 || p->height < 2) return fail(ctx, RTHIP_E_BADARG, "width and height must be >= 2 (the jitter divides by width-1, height-1: lib.rs:283,285)");
    if (p->width > 65536 || p->height > 65536) return fail(ctx, RTHIP_E_BADARG, "width/height above 65536 are not supported");
    if (uint64_t(p->width) * p->height > (1ull << 31)) return fail(ctx, RTHIP_E_BADARG, "more than 2^31 pixels are not supported (32-bit pixel indices)");
    if (p->spp == 0) return fail(ctx, RTHIP_E_BADARG, "spp must be >= 1");
    // the RNG counter of a sample has 256 slots: slot 0 = pixel jitter, slot 1 + bounce = the material draws of that segment (rt_rng.h)
    if (p->max_depth > 255) return fail(ctx, RTHIP_E_BADARG, "max_depth above 255 is not supported (a deeper path would reuse the random numbers of the pixel's next sample)");
    if (p->band_rows != 0 && p->band_stride == 0) return fail(ctx, RTHIP_E_BADARG, "band_stride must be >= 1 when band_rows != 0");
    if (p->flags & ~uint32_t(RTHIP_F_ACCUMULATE | RTHIP_F_POINT_LIGHTS | RTHIP_F_COUNT_STATS | RTHIP_F_NO_BVH | RTHIP_F_ADAPTIVE | RTHIP_F_TILE_LISTS | RTHIP_F_COUNT_EXECUTED))
        return fail(ctx, RTHIP_E_BADARG, "unknown flag bits");
    for (int i = 0; i < 7; i++)
        if (p->reserved[i] != 0) return fail(ctx, RTHIP_E_BADARG, "reserved fields must be 0");
    return RTHIP_OK;
}

void fill_camera(KernelParams &k, const rthip_camera &c) {
    k.ox = float(c.origin[0]), k.oy = float(c.origin[1]), k.oz = float(c.origin[2]);
    k.d00x = float(c.d00[0]), k.d00y = float(c.d00[1]), k.d00z = float(c.d00[2]);
    k.hx = float(c.horizontal[0]), k.hy = float(c.horizontal[1]), k.hz = float(c.horizontal[2]);
    k.vx = float(c.vertical[0]), k.vy = float(c.vertical[1]), k.vz = float(c.vertical[2]);
}

}  // namespace

extern "C" {

int rthip_abi_version(void) { return RTHIP_ABI_VERSION; }

const char *rthip_status_string(int status) {
    switch (status) {
        case RTHIP_OK: return "ok";
        case RTHIP_E_BADARG: return "bad argument";
        case RTHIP_E_HIP: return "HIP runtime error";
        case RTHIP_E_OOM: return "out of device memory";
        case RTHIP_E_NODEVICE: return "no such device";
        case RTHIP_E_STATE: return "call order / state error";
        case RTHIP_E_INTERNAL: return "internal error";
        case RTHIP_E_RCCL: return "RCCL error";
        default: return "unknown status";
    }
}

#ifdef RT_CLASS_CYCLES
// debug build (tools/class_cycles.py): the raw counter shards of the last pass — the kernel files its counters under shard = pmask
extern "C" const unsigned long long *rthip_debug_shards(rthip_ctx *ctx) { return ctx->stats_host; }
#endif
double rthip_algorithmic_flops(const rthip_stats *s) {
    if (!s) return 0.0;
    const double hits = double(s->hits[0]) + double(s->hits[1]) + double(s->hits[2]) + double(s->hits[3]);
    return RTHIP_FLOPS_RAY_SETUP * double(s->paths) + RTHIP_FLOPS_SEGMENT * double(s->rays) + RTHIP_FLOPS_SPHERE_TEST * double(s->sphere_tests) +
           RTHIP_FLOPS_PLANE_TEST * double(s->plane_tests) + RTHIP_FLOPS_TRI_TEST * double(s->tri_tests) +
           RTHIP_FLOPS_NODE_TEST * double(s->node_tests) + RTHIP_FLOPS_HIT * hits + RTHIP_FLOPS_SKY * double(s->sky_evals) +
           RTHIP_FLOPS_LAMBERTIAN * double(s->hits[RTHIP_MAT_LAMBERTIAN]) + RTHIP_FLOPS_METAL * double(s->hits[RTHIP_MAT_METAL]) +
           RTHIP_FLOPS_DIELECTRIC * double(s->hits[RTHIP_MAT_DIELECTRIC]) + RTHIP_FLOPS_TEXTURE * double(s->hits[RTHIP_MAT_TEXTURE]) +
           RTHIP_FLOPS_TONEMAP * double(s->paths) / double(s->nb_samples ? s->nb_samples : 1);
}

const char *rthip_last_error(const rthip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int rthip_device_count(int *out_count) {
    if (!out_count) return RTHIP_E_BADARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fail(nullptr, e == hipErrorNoDevice ? RTHIP_E_NODEVICE : RTHIP_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *out_count = n;
    return RTHIP_OK;
}

int rthip_camera_lookat(double x, double y, double z, const double lookat[3], rthip_camera *out) {
    // Camera::new_lookat — camera.rs:19-41 (fov 20 "degrees" used as 2*radians(20), 16:9 viewport regardless of image size)
    if (!lookat || !out) return RTHIP_E_BADARG;
    const double PI = 3.14159265358979323846264338327950288;
    D3 origin{x, y, z}, la = d3(lookat), vup{0.0, 1.0, 0.0};
    const double viewport_height = 2.0 * (20.0 * (PI / 180.0));
    const double viewport_width = (16.0 / 9.0) * viewport_height;
    D3 w = normalize(sub(origin, la));
    D3 u = normalize(cross(vup, w));
    D3 v = cross(w, u);
    D3 H = mul(u, viewport_width), V = mul(v, viewport_height);
    // lower_left_corner = origin - H/2 - V/2 - w ; d00 = lower_left_corner - origin
    D3 llc = sub(sub(sub(origin, mul(H, 0.5)), mul(V, 0.5)), w);
    D3 d00 = sub(llc, origin);
    out->origin[0] = origin.x, out->origin[1] = origin.y, out->origin[2] = origin.z;
    out->d00[0] = d00.x, out->d00[1] = d00.y, out->d00[2] = d00.z;
    out->horizontal[0] = H.x, out->horizontal[1] = H.y, out->horizontal[2] = H.z;
    out->vertical[0] = V.x, out->vertical[1] = V.y, out->vertical[2] = V.z;
    return RTHIP_OK;
}

int rthip_owned_rows(const rthip_render_params *params, uint32_t *out_rows) { return owned_rows(params, out_rows); }

int rthip_create(int device, rthip_ctx **out_ctx) {
    if (!out_ctx) return RTHIP_E_BADARG;
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(nullptr, RTHIP_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(nullptr, RTHIP_E_NODEVICE, "device index out of range");
    rthip_ctx *ctx = new (std::nothrow) rthip_ctx();
    if (!ctx) return RTHIP_E_OOM;
    ctx->device = device;
    std::memset(&ctx->base, 0, sizeof ctx->base);
    int st = [&]() -> int {
        HIPCHK(ctx, hipSetDevice(device));
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreate(&ctx->ev0));
        HIPCHK(ctx, hipEventCreate(&ctx->ev1));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_last, hipEventDisableTiming));
        HIPCHK(ctx, hipMalloc(&ctx->stats.p, RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long)));
        ctx->stats.bytes = RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long);
        HIPCHK(ctx, hipMalloc(&ctx->pick_out.p, sizeof(long long)));
        ctx->pick_out.bytes = sizeof(long long);
        HIPCHK(ctx, hipHostMalloc((void **)&ctx->stats_host, RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long), hipHostMallocDefault));
        // the device address of the static scratch array, once per device, with that device current (each GPU holds its own copy)
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->wave_slots = uint32_t(prop.multiProcessorCount) * 28u;
        }
        HIPCHK(ctx, rt_scratch_init(device));
        ctx->scratch = rt_scratch_ptr(device);
        if (!ctx->scratch) return fail(ctx, RTHIP_E_INTERNAL, "the scratch array has no device address");
        return RTHIP_OK;
    }();
    if (st != RTHIP_OK) {
        g_err = ctx->err;
        rthip_destroy(ctx);
        return st;
    }
    *out_ctx = ctx;
    return RTHIP_OK;
}

void rthip_destroy(rthip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : {&ctx->sph_geom, &ctx->sph_shade, &ctx->pln_geom, &ctx->tri_geom, &ctx->materials, &ctx->texels, &ctx->tex_info, &ctx->lights,
                      &ctx->bvh_nodes, &ctx->bvh_recs, &ctx->stats, &ctx->accum, &ctx->staging_rgba, &ctx->staging_accum, &ctx->pick_out, &ctx->pix_state, &ctx->worklist, &ctx->partial, &ctx->tile_pre, &ctx->tile_order, &ctx->tile_cost})
        b->release();
    if (ctx->stats_host) (void)hipHostFree(ctx->stats_host);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_last) (void)hipEventDestroy(ctx->ev_last);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (int i = 0; i < 2; i++) {
        ctx->staging_async[i].release();
        if (ctx->ev_rendered[i]) (void)hipEventDestroy(ctx->ev_rendered[i]);
        if (ctx->ev_copied[i]) (void)hipEventDestroy(ctx->ev_copied[i]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rthip_upload_scene(rthip_ctx *ctx, const rthip_scene_desc *s) {
    if (!ctx) return RTHIP_E_BADARG;
    if (!s) return fail(ctx, RTHIP_E_BADARG, "scene is NULL");
    try {
        if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_triangles && !s->triangles) ||
            (s->n_materials && !s->materials) || (s->n_textures && !s->textures) || (s->n_lights && !s->lights))
            return fail(ctx, RTHIP_E_BADARG, "a non-empty array has a NULL pointer");
        const uint64_t n_shapes = s->n_spheres + s->n_planes + s->n_triangles;
        if (n_shapes >= 0x7fffffffull) return fail(ctx, RTHIP_E_BADARG, "too many shapes");
        if (s->n_spheres + s->n_triangles >= RT_BVH_MAX_PRIMS) return fail(ctx, RTHIP_E_BADARG, "too many spheres + triangles for the BVH (2^29)");
        if (s->n_materials >= 0x7fffffffull || s->n_textures >= 0x7fffffffull || s->n_lights > 1024)
            return fail(ctx, RTHIP_E_BADARG, "too many materials/textures/lights");
        // ---- validate ----
        for (uint64_t i = 0; i < s->n_materials; i++) {
            const rthip_material &m = s->materials[i];
            if (m.kind > RTHIP_MAT_TEXTURE) return fail(ctx, RTHIP_E_BADARG, "material " + std::to_string(i) + ": unknown kind");
            if (m.kind == RTHIP_MAT_TEXTURE && m.texture >= s->n_textures)
                return fail(ctx, RTHIP_E_BADARG, "material " + std::to_string(i) + ": texture index out of range");
        }
        for (uint64_t i = 0; i < s->n_textures; i++) {
            const rthip_texture &t = s->textures[i];
            if (!t.rgb8 || t.width == 0 || t.height == 0 || t.width > 32768 || t.height > 32768)
                return fail(ctx, RTHIP_E_BADARG, "texture " + std::to_string(i) + ": bad size or NULL texels");
        }
        for (uint64_t i = 0; i < s->n_spheres; i++) {
            if (s->spheres[i].material >= s->n_materials) return fail(ctx, RTHIP_E_BADARG, "sphere " + std::to_string(i) + ": material index out of range");
            if (!finite3(s->spheres[i].center) || !finite1(s->spheres[i].radius) || s->spheres[i].radius == 0.0)
                return fail(ctx, RTHIP_E_BADARG, "sphere " + std::to_string(i) + ": centre or radius not finite (or beyond 1e18), or zero radius");
        }
        for (uint64_t i = 0; i < s->n_planes; i++) {
            if (s->planes[i].material >= s->n_materials) return fail(ctx, RTHIP_E_BADARG, "plane " + std::to_string(i) + ": material index out of range");
            D3 n = d3(s->planes[i].normal);
            if (!finite3(s->planes[i].point) || !finite3(s->planes[i].normal) || dot(n, n) == 0.0)
                return fail(ctx, RTHIP_E_BADARG, "plane " + std::to_string(i) + ": bad point/normal");
        }
        for (uint64_t i = 0; i < s->n_triangles; i++) {
            if (s->triangles[i].material >= s->n_materials) return fail(ctx, RTHIP_E_BADARG, "triangle " + std::to_string(i) + ": material index out of range");
            if (!finite3(s->triangles[i].v0) || !finite3(s->triangles[i].v1) || !finite3(s->triangles[i].v2))
                return fail(ctx, RTHIP_E_BADARG, "triangle " + std::to_string(i) + ": vertex not finite (or beyond 1e18)");
        }
        HIPCHK(ctx, hipSetDevice(ctx->device));
        // passes enqueued by rthip_render_device on the CALLER's streams may still be reading the scene buffers
        HIPCHK(ctx, hipDeviceSynchronize());
        ctx->have_scene = false;
        const double PI = 3.14159265358979323846264338327950288;

        int st;
        // ---- materials ----
        std::vector<DevMaterial> mats(s->n_materials);
        for (uint64_t i = 0; i < s->n_materials; i++) {
            const rthip_material &m = s->materials[i];
            DevMaterial &d = mats[i];
            std::memset(&d, 0, sizeof d);
            // constructors store albedo/PI (lambertian_diffuse.rs:21, metal.rs:21, dielectric.rs:22); scatter multiplies by 3.0
            d.ar = float(m.albedo[0] / PI), d.ag = float(m.albedo[1] / PI), d.ab = float(m.albedo[2] / PI);
            d.tr = float(m.albedo[0] / PI * 3.0), d.tg = float(m.albedo[1] / PI * 3.0), d.tb = float(m.albedo[2] / PI * 3.0);
            d.param = float(m.param);
            d.kind = m.kind;
            d.texture = m.texture;
            d.shininess = float(m.shininess);
            if (m.kind == RTHIP_MAT_DIELECTRIC) {
                double r0 = (1.0 - m.param) / (1.0 + m.param);  // dielectric.rs:48
                d.r1 = float(r0 * r0);
                d.inv_ior = float(1.0 / m.param);  // dielectric.rs:81
            }
            // the Schlick term is a dielectric's alone: for a texture material the slot carries the texture index (as bits), so that
            // the texture arm of the kernel finds it in the surface record it already holds instead of loading the material again
            if (m.kind == RTHIP_MAT_TEXTURE) d.r1 = bits_f(m.texture);
        }
        if ((st = upload(ctx, ctx->materials, mats.data(), mats.size() * sizeof(DevMaterial)))) return st;
        // ---- spheres: geom {c, r^2}; shading record 4 x float4 {c, 1/r} {albedo/pi*3, param} {kind, r1 | texture, 1/ior, material}
        //      {albedo/pi, shininess} (the last one is read by the point-light code only) ----
        std::vector<rt_f4> geom(s->n_spheres), shade(RT_SHADE_RECS * s->n_spheres);
        for (uint64_t i = 0; i < s->n_spheres; i++) {
            const rthip_sphere &sp = s->spheres[i];
            const DevMaterial &m = mats[sp.material];
            geom[i] = f4(sp.center[0], sp.center[1], sp.center[2], sp.radius * sp.radius);
            shade[RT_SHADE_RECS * i + 0] = f4(sp.center[0], sp.center[1], sp.center[2], 1.0 / sp.radius);
            shade[RT_SHADE_RECS * i + 1] = rt_f4{m.tr, m.tg, m.tb, m.param};
            shade[RT_SHADE_RECS * i + 2] = rt_f4{bits_f(m.kind), m.r1, m.inv_ior, bits_f(sp.material)};
            shade[RT_SHADE_RECS * i + 3] = rt_f4{m.ar, m.ag, m.ab, m.shininess};
        }
        if ((st = upload(ctx, ctx->sph_geom, geom.data(), geom.size() * sizeof(rt_f4)))) return st;
        if ((st = upload(ctx, ctx->sph_shade, shade.data(), shade.size() * sizeof(rt_f4)))) return st;
        // ---- planes: {n, n.p0} {p0, material} {tu} {tv} ----
        std::vector<rt_f4> pl(4 * s->n_planes);
        for (uint64_t i = 0; i < s->n_planes; i++) {
            const rthip_plane &p = s->planes[i];
            D3 n = normalize(d3(p.normal)), p0 = d3(p.point);
            D3 a = std::fabs(n.x) > 0.9 ? D3{0, 1, 0} : D3{1, 0, 0};
            D3 tu = normalize(cross(a, n)), tv = cross(n, tu);
            pl[4 * i + 0] = f4(n.x, n.y, n.z, dot(n, p0));
            pl[4 * i + 1] = rt_f4{float(p0.x), float(p0.y), float(p0.z), bits_f(p.material)};
            pl[4 * i + 2] = f4(tu.x, tu.y, tu.z, 0);
            pl[4 * i + 3] = f4(tv.x, tv.y, tv.z, 0);
        }
        if ((st = upload(ctx, ctx->pln_geom, pl.data(), pl.size() * sizeof(rt_f4)))) return st;
        // ---- triangles: {v0, material} {e1} {e2} ----
        std::vector<rt_f4> tr(3 * s->n_triangles);
        for (uint64_t i = 0; i < s->n_triangles; i++) {
            const rthip_triangle &t = s->triangles[i];
            // edges are differences of the FLOAT vertices so that the device sees exactly the float triangle
            float v0[3] = {float(t.v0[0]), float(t.v0[1]), float(t.v0[2])};
            float v1[3] = {float(t.v1[0]), float(t.v1[1]), float(t.v1[2])};
            float v2[3] = {float(t.v2[0]), float(t.v2[1]), float(t.v2[2])};
            tr[3 * i + 0] = rt_f4{v0[0], v0[1], v0[2], bits_f(t.material)};
            tr[3 * i + 1] = rt_f4{v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.0f};
            tr[3 * i + 2] = rt_f4{v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.0f};
        }
        if ((st = upload(ctx, ctx->tri_geom, tr.data(), tr.size() * sizeof(rt_f4)))) return st;
        // ---- textures: RGB8 -> RGBA8 in 8x8-texel tiles (a wave's nearby UVs touch few 256-B tiles) ----
        std::vector<DevTexInfo> ti(s->n_textures);
        size_t total_texels = 0;
        for (uint64_t i = 0; i < s->n_textures; i++) {
            const rthip_texture &t = s->textures[i];
            ti[i].width = t.width, ti[i].height = t.height;
            ti[i].tiles_per_row = (t.width + 7) / 8;
            ti[i].magic_w = t.width == 1 ? 0xffffffffu : uint32_t((1ull << 32) / t.width);
            ti[i].width_f = float(t.width), ti[i].height_f = float(t.height);
            ti[i].magic_h = t.height == 1 ? 0xffffffffu : uint32_t((1ull << 32) / t.height);
            if (total_texels > 0xffffffffull) return fail(ctx, RTHIP_E_BADARG, "textures exceed 2^32 texels");
            ti[i].first_texel = uint32_t(total_texels);
            total_texels += size_t(ti[i].tiles_per_row) * ((t.height + 7) / 8) * 64;
        }
        if (total_texels > 0xffffffffull) return fail(ctx, RTHIP_E_BADARG, "textures exceed 2^32 texels");
        std::vector<uint32_t> texels(total_texels, 0u);
        for (uint64_t i = 0; i < s->n_textures; i++) {
            const rthip_texture &t = s->textures[i];
            for (uint32_t y = 0; y < t.height; y++)
                for (uint32_t x = 0; x < t.width; x++) {
                    const uint8_t *px = t.rgb8 + (size_t(y) * t.width + x) * 3;
                    size_t idx = size_t(ti[i].first_texel) + ((size_t(y >> 3) * ti[i].tiles_per_row + (x >> 3)) << 6) + ((y & 7u) << 3) + (x & 7u);
                    texels[idx] = uint32_t(px[0]) | (uint32_t(px[1]) << 8) | (uint32_t(px[2]) << 16) | 0xff000000u;
                }
        }
        if ((st = upload(ctx, ctx->texels, texels.data(), texels.size() * sizeof(uint32_t)))) return st;
        if ((st = upload(ctx, ctx->tex_info, ti.data(), ti.size() * sizeof(DevTexInfo)))) return st;
        // ---- lights ----
        std::vector<DevLight> lights(s->n_lights);
        for (uint64_t i = 0; i < s->n_lights; i++) {
            const rthip_light &l = s->lights[i];
            lights[i] = DevLight{float(l.position[0]), float(l.position[1]), float(l.position[2]), 0.0f,
                                 float(l.intensity[0]), float(l.intensity[1]), float(l.intensity[2]), 0.0f};
        }
        if ((st = upload(ctx, ctx->lights, lights.data(), lights.size() * sizeof(DevLight)))) return st;
        // ---- linear BVH over spheres + triangles for big scenes ----
        std::vector<DevBvhNode> nodes;
        std::vector<uint32_t> prims;
        uint32_t bvh_levels = 0;
        if (rt_wants_bvh(s->n_spheres, s->n_triangles)) bvh_levels = rt_build_bvh(s, nodes, prims);
        if (bvh_levels > RT_BVH_MAX_DEPTH) return fail(ctx, RTHIP_E_INTERNAL, "BVH deeper than the traversal stack");
        const uint32_t bvh_top = rt_bvh_top_first(nodes, RT_BVH_TOP_MAX);
        std::vector<DevBvhNodeQ> qnodes;
        if (!rt_bvh_quantize(nodes, qnodes)) return fail(ctx, RTHIP_E_INTERNAL, "BVH node quantisation failed");
        if ((st = upload(ctx, ctx->bvh_nodes, qnodes.data(), qnodes.size() * sizeof(DevBvhNodeQ)))) return st;
        // primitives in leaf order, self-contained records (geometry + global shape index + kind)
        std::vector<rt_f4> recs(3 * prims.size());
        for (size_t j = 0; j < prims.size(); j++) {
            const uint32_t id = prims[j];
            if (id < s->n_spheres) {
                recs[3 * j + 0] = rt_f4{geom[id].x, geom[id].y, geom[id].z, bits_f(id)};
                recs[3 * j + 1] = rt_f4{geom[id].w, 0.0f, 0.0f, bits_f(0u)};
                recs[3 * j + 2] = rt_f4{0.0f, 0.0f, 0.0f, 0.0f};
            } else {
                const size_t t = id - s->n_spheres;
                recs[3 * j + 0] = rt_f4{tr[3 * t].x, tr[3 * t].y, tr[3 * t].z, bits_f(uint32_t(s->n_spheres + s->n_planes + t))};
                recs[3 * j + 1] = rt_f4{tr[3 * t + 1].x, tr[3 * t + 1].y, tr[3 * t + 1].z, bits_f(1u)};
                recs[3 * j + 2] = tr[3 * t + 2];
            }
        }
        if ((st = upload(ctx, ctx->bvh_recs, recs.data(), recs.size() * sizeof(rt_f4)))) return st;

        KernelParams &k = ctx->base;
        k.n_spheres = uint32_t(s->n_spheres), k.n_planes = uint32_t(s->n_planes), k.n_tris = uint32_t(s->n_triangles);
        k.n_lights = uint32_t(s->n_lights);
        k.n_bvh_nodes = uint32_t(nodes.size());
        k.bvh_levels = bvh_levels;
        k.bvh_top_nodes = bvh_top;
        k.sph_geom = (const rt_f4 *)ctx->sph_geom.p, k.sph_shade = (const rt_f4 *)ctx->sph_shade.p;
        k.pln_geom = (const rt_f4 *)ctx->pln_geom.p, k.tri_geom = (const rt_f4 *)ctx->tri_geom.p;
        k.materials = (const DevMaterial *)ctx->materials.p;
        k.texels = (const uint32_t *)ctx->texels.p, k.tex_info = (const DevTexInfo *)ctx->tex_info.p;
        k.lights = (const DevLight *)ctx->lights.p;
        k.bvh_nodes = (const DevBvhNodeQ *)ctx->bvh_nodes.p, k.bvh_recs = (const rt_f4 *)ctx->bvh_recs.p;
        ctx->n_shapes = n_shapes;
        ctx->have_scene = true;
        ctx->nb_samples = 0;  // a new scene invalidates the running sum
        ctx->frame_gen++;
        ctx->pre_lists = nullptr;
        return RTHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(ctx, RTHIP_E_OOM, "host allocation failed while flattening the scene");
    } catch (...) {
        return fail(ctx, RTHIP_E_INTERNAL, "unexpected exception in rthip_upload_scene");
    }
}

int rthip_set_camera(rthip_ctx *ctx, const rthip_camera *camera) {
    if (!ctx) return RTHIP_E_BADARG;
    if (!camera) return fail(ctx, RTHIP_E_BADARG, "camera is NULL");
    if (!finite3(camera->origin) || !finite3(camera->d00) || !finite3(camera->horizontal) || !finite3(camera->vertical))
        return fail(ctx, RTHIP_E_BADARG, "camera has non-finite components");
    ctx->camera = *camera;
    fill_camera(ctx->base, *camera);
    ctx->have_camera = true;
    ctx->nb_samples = 0;  // standalone/src/main.rs:103-120: a camera command invalidates the pixels
    ctx->frame_gen++;
    ctx->pre_lists = nullptr;
    return RTHIP_OK;
}

int rthip_reset_accum(rthip_ctx *ctx) {
    if (!ctx) return RTHIP_E_BADARG;
    ctx->nb_samples = 0;
    ctx->pre_lists = nullptr;
    return RTHIP_OK;
}

int rthip_accum_layout(rthip_ctx *ctx, uint32_t *out_rows, uint32_t *out_width, uint64_t *out_nb_samples) {
    if (!ctx) return RTHIP_E_BADARG;
    const bool live = ctx->nb_samples > 0;
    if (out_rows) *out_rows = live ? ctx->acc_rows : 0u;
    if (out_width) *out_width = live ? ctx->acc_width : 0u;
    if (out_nb_samples) *out_nb_samples = ctx->nb_samples;
    return RTHIP_OK;
}

int rthip_pixel_status(rthip_ctx *ctx, uint8_t *status_out, uint8_t *same_color_count_out, uint64_t capacity_pixels) {
    if (!ctx) return RTHIP_E_BADARG;
    if (!status_out) return fail(ctx, RTHIP_E_BADARG, "status_out is NULL");
    if (!ctx->adaptive || ctx->nb_samples == 0) return fail(ctx, RTHIP_E_STATE, "no RTHIP_F_ADAPTIVE pass has run since the last reset");
    try {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const size_t npx = size_t(ctx->acc_rows) * ctx->acc_width;
        if (capacity_pixels < npx)
            return fail(ctx, RTHIP_E_BADARG, "rthip_pixel_status: the buffers hold " + std::to_string(capacity_pixels) + " pixels, the running sum has " +
                                                 std::to_string(npx) + " (rthip_accum_layout)");
        std::vector<uint32_t> st(npx);
        HIPCHK(ctx, hipDeviceSynchronize());
        HIPCHK(ctx, hipMemcpy(st.data(), ctx->pix_state.p, npx * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < npx; i++) {
            status_out[i] = (st[i] & RT_PIX_FINAL) ? RTHIP_STATUS_FINAL : ((st[i] & RT_PIX_HAS_LAST) ? RTHIP_STATUS_UNSTABLE : RTHIP_STATUS_NOT_STARTED);
            if (same_color_count_out) same_color_count_out[i] = uint8_t((st[i] >> RT_PIX_COUNT_SHIFT) & RT_PIX_COUNT_MASK);
        }
        return RTHIP_OK;
    } catch (const std::bad_alloc &) {
        return fail(ctx, RTHIP_E_OOM, "host allocation failed in rthip_pixel_status");
    } catch (...) {
        return fail(ctx, RTHIP_E_INTERNAL, "unexpected exception in rthip_pixel_status");
    }
}

// One pass, enqueued on `hip_stream`. noalloc = the rthip_render_device_noalloc contract: no hipMalloc/hipFree (nor any
// allocating routine) on this path — the running sum lives in the caller's d_accum, the counter shards and the partial sums
// of a sliced launch in this device's copy of the static scratch array (ctx->scratch), and the sizes are checked, not grown.
// Second half of a pass that counts: waits for the pass, sums the counter shards (they landed in the context's page-locked
// buffer) and reads the kernel time. Split from the enqueue so that a multi-device caller can start every device first.
static int collect_stats(rthip_ctx *ctx, rthip_stats *stats) {
    if (!ctx->pending_stats) return fail(ctx, RTHIP_E_STATE, "no pass with counters is pending");
    ctx->pending_stats = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_last));
    std::memset(stats, 0, sizeof *stats);
    const unsigned long long *h = ctx->stats_host;
    float ms = 0.0f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    for (int sh = 0; sh < RT_STAT_SHARDS; sh++) {
        stats->paths += h[sh * RT_STAT_N + RT_STAT_PATHS];
        stats->rays += h[sh * RT_STAT_N + RT_STAT_RAYS];
        stats->sphere_tests += h[sh * RT_STAT_N + RT_STAT_SPHERE];
        stats->plane_tests += h[sh * RT_STAT_N + RT_STAT_PLANE];
        stats->tri_tests += h[sh * RT_STAT_N + RT_STAT_TRI];
        stats->node_tests += h[sh * RT_STAT_N + RT_STAT_NODE];
        for (int k = 0; k < 4; k++) stats->hits[k] += h[sh * RT_STAT_N + RT_STAT_HIT0 + k];
        stats->sky_evals += h[sh * RT_STAT_N + RT_STAT_SKY];
        stats->shadow_rays += h[sh * RT_STAT_N + RT_STAT_SHADOW];
        stats->pixels_traced += h[sh * RT_STAT_N + RT_STAT_PIX_TRACED];
        stats->pixels_final += h[sh * RT_STAT_N + RT_STAT_PIX_FINAL];
    }
#ifndef RT_PRE_DEBUG
    if (!ctx->pending_adaptive) stats->pixels_traced = uint64_t(ctx->pending_rows) * ctx->pending_width;
#endif
    stats->rows = ctx->pending_rows;
    stats->nb_samples = ctx->nb_samples;
    stats->kernel_ms = ms;
    return RTHIP_OK;
}

static int render_pass(rthip_ctx *ctx, const rthip_render_params *p, void *d_rgba8, void *d_accum, void *hip_stream, rthip_stats *stats, bool noalloc,
                       bool defer_stats = false) {
    if (!ctx) return RTHIP_E_BADARG;
    try {
        int st = validate_params(ctx, p);
        if (st) return st;
        if (!ctx->have_scene) return fail(ctx, RTHIP_E_STATE, "rthip_upload_scene has not been called");
        if (!ctx->have_camera) return fail(ctx, RTHIP_E_STATE, "rthip_set_camera has not been called");
        uint32_t rows = 0;
        if ((st = owned_rows(p, &rows))) return fail(ctx, st, "bad band parameters");
        if ((p->flags & RTHIP_F_POINT_LIGHTS) && ctx->base.n_lights == 0) return fail(ctx, RTHIP_E_STATE, "RTHIP_F_POINT_LIGHTS without lights in the scene");
        const bool adaptive = (p->flags & RTHIP_F_ADAPTIVE) != 0;
        if (noalloc && adaptive)
            return fail(ctx, RTHIP_E_BADARG, "rthip_render_device_noalloc: RTHIP_F_ADAPTIVE is not supported (the per-pixel convergence state is context-owned memory that grows with the image)");
        if (noalloc && !d_accum && rows)
            return fail(ctx, RTHIP_E_BADARG, "rthip_render_device_noalloc: d_accum is required (it holds the running sum; the context allocates none on this path)");
        hipStream_t stream = (hipStream_t)hip_stream;
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const bool same_layout = ctx->acc_width == p->width && ctx->acc_height == p->height && ctx->acc_rows == rows &&
                                 ctx->acc_band_rows == p->band_rows && ctx->acc_band_first == p->band_first && ctx->acc_band_stride == p->band_stride;
        const bool accumulate = (p->flags & RTHIP_F_ACCUMULATE) && ctx->nb_samples > 0;
        if (accumulate && !same_layout) return fail(ctx, RTHIP_E_STATE, "RTHIP_F_ACCUMULATE with a different image/band layout than the running sum");
        if (accumulate && adaptive != ctx->adaptive)
            return fail(ctx, RTHIP_E_STATE, "RTHIP_F_ADAPTIVE must be set on every pass of a running sum or on none (Final pixels stop accumulating)");
        const uint64_t prior = accumulate ? ctx->nb_samples : 0;
        // the RNG counter of (sample, slot) is sample*256 + slot in 32 bits (rt_rng.h): the sample streams would repeat beyond 2^24
        if (prior + p->spp > (1ull << 24)) return fail(ctx, RTHIP_E_BADARG, "more than 2^24 samples per pixel");
        if (rows == 0) {
            if (stats) std::memset(stats, 0, sizeof *stats);
            return RTHIP_OK;
        }
        if (!noalloc && (st = ensure(ctx, ctx->accum, size_t(rows) * p->width * 3 * sizeof(float)))) return st;
        // where the running sum lives: the context's buffer, or (allocation-free passes) the caller's d_accum
        const void *acc_buffer = noalloc ? (const void *)d_accum : (const void *)ctx->accum.p;
        if (accumulate && acc_buffer != ctx->acc_buffer)
            return fail(ctx, RTHIP_E_STATE, "RTHIP_F_ACCUMULATE: the running sum of this context lives in another buffer (allocation-free passes keep it in the caller's d_accum)");

        KernelParams k = ctx->base;
        k.width = p->width, k.height = p->height;
        k.inv_wm1 = float(1.0 / (double(p->width) - 1.0));
        k.inv_hm1 = float(1.0 / (double(p->height) - 1.0));
        k.hwx = k.hx * k.inv_wm1, k.hwy = k.hy * k.inv_wm1, k.hwz = k.hz * k.inv_wm1;
        k.vhx = k.vx * k.inv_hm1, k.vhy = k.vy * k.inv_hm1, k.vhz = k.vz * k.inv_hm1;
        k.spp = p->spp, k.max_depth = p->max_depth, k.seed = p->seed;
        k.sample0 = uint32_t(prior);
        k.flags = p->flags;
        k.band_rows = p->band_rows, k.band_first = p->band_first, k.band_stride = p->band_stride, k.owned_rows = rows;
        k.accumulate = accumulate ? 1u : 0u;
        k.count_executed = (p->flags & RTHIP_F_COUNT_EXECUTED) ? 1u : 0u;
        k.inv_nb = float(1.0 / double(prior + p->spp));
        k.accum = noalloc ? (float *)d_accum : (float *)ctx->accum.p;
        k.accum_out = noalloc ? nullptr : (float *)d_accum;
        k.rgba8 = (uint32_t *)d_rgba8;
        unsigned char *scratch = (unsigned char *)ctx->scratch;  // the device address from hipGetSymbolAddress, never the array's name
        unsigned long long *d_stats = noalloc ? (unsigned long long *)scratch : (unsigned long long *)ctx->stats.p;
        k.stats = stats ? d_stats : nullptr;  // no reader, no counting
        k.tree_scene = k.n_bvh_nodes > 0 ? 1u : 0u;
        if (p->flags & RTHIP_F_NO_BVH) k.n_bvh_nodes = 0;
        // Sample slicing. The GROUPING of the samples (slice_spp) is a function of spp and of the scene alone, so that the image
        // is the same for every band layout / GPU count; whether a launch IS sliced is a scheduling decision (same image).
        //  * scan scenes: groups of 64 samples, sliced only when the launch is too small to fill the GPU to its end;
        //  * scenes with a tree: a wave tile's cost spans three orders of magnitude (sky: microseconds; a tile looking into the
        //    gaps between mirror meshes at depth 50: 36 ms of a 63 ms frame, measured per wave, tools/timeline.py), and the
        //    C4 frame ended with a fifth of its time spent on a few such waves. So the samples form up to RT_TREE_SLICES groups,
        //    traced by different waves whenever the launch has fewer than RT_TREE_SLICE_BELOW_WAVES wave tiles (C4 at 4K:
        //    129 600 -> sliced, 56.8 -> 52.2 ms; C5 at 8K: 518 400 -> not: its longest wave is 7 % of the frame, and every
        //    slice costs a wave's ramp-down, 459 -> 473 ms with four).
        const bool tree_scene = ctx->base.n_bvh_nodes > 0;  // (not k.n_bvh_nodes: RTHIP_F_NO_BVH must not change the sums)
        uint32_t max_slices = RT_MAX_SLICES;
        k.slice_spp = RT_SLICE_SPP;
        if (tree_scene) {
            static const int env_slices = [] { const char *e = getenv("RTHIP_TREE_SLICES"); return e ? atoi(e) : 0; }();  // tuning knob (changes the rounding of the sums)
            max_slices = env_slices > 0 ? uint32_t(env_slices) : RT_TREE_SLICES;
            k.slice_spp = RT_TREE_SLICE_SPP;
        }
        while (uint64_t(k.slice_spp) * max_slices < p->spp) k.slice_spp *= 2;  // a power of two (the kernels mask with slice_spp - 1)
        k.n_slices = (p->spp + k.slice_spp - 1) / k.slice_spp;
        const uint32_t full_slices = k.n_slices;  // the sample groups of a pixel's sum
        const uint64_t groups = (p->flags & RTHIP_F_ADAPTIVE) && accumulate ? (uint64_t(rows) * p->width + 255u) / 256u
                                                                             : uint64_t((p->width + RT_TILE_W - 1u) / RT_TILE_W) * ((rows + RT_TILE_H - 1u) / RT_TILE_H);
        if (tree_scene ? groups * 4u >= RT_TREE_SLICE_BELOW_WAVES : groups >= RT_SLICE_BELOW_BLOCKS) k.n_slices = 1;
        // Launch order by measured cost (further down): which launches take part, and what identifies a frame layout.
        //  * scan scenes: the launches that are sliced for the same reason — too few workgroups per slot to hide their tail;
        //  * scenes with a tree: the launches that would be sliced; once their groups have been measured (the first pass of the
        //    layout: row order, sliced) they start longest-first and need NO slices — the heavy tiles start at t = 0 and the launch
        //    ends with sky tiles, without the ramp-down every slice costs: C4 54.4 (4 slices) -> 49.9 ms. Launches that are not
        //    sliced anyway gain nothing from the order (C5: 397.9 vs 397.3 ms) and keep the row order (neighbouring tiles side by side).
        //  * those scan launches, once measured, are MIXED: of the groups in order of cost, those whose waves would be long against
        //    the whole launch are traced slice by slice, the bulk in one piece (no ramp-down, no partial sums), and the cheapest that
        //    are worth it again in slices, so that the launch ends with short pieces (tile_order_kernel computes the two boundaries
        //    from the measurements). A 1/8-frame launch comes out sliced but for its sky, a 1/2-frame launch 75 % whole: 6.39 -> 6.15 ms.
        //    The whole 4K frame (RTHIP_MIXED_ALL=1) does NOT gain: a mixed launch keeps 6 900 of its 7 168 wave slots busy until 0.1 ms
        //    before its end (tools/timeline_mixed.py) and takes 12.2 ms against the 12.05 of the plain row-order launch, whose last
        //    millisecond runs half empty — the kernel is bound by instruction issue, the waves of a draining GPU run faster, and the
        //    slices cost their ramp-down. Whole frames keep the row order and are not measured.
        static const bool order_none = getenv("RTHIP_NO_TILE_ORDER") != nullptr, order_model_only = getenv("RTHIP_ORDER_MODEL_ONLY") != nullptr,
                          order_tree = getenv("RTHIP_NO_TREE_ORDER") == nullptr, no_mixed = getenv("RTHIP_NO_MIXED") != nullptr,
                          mixed_all = getenv("RTHIP_MIXED_ALL") != nullptr;
        const bool scan_ordered = !tree_scene && k.n_spheres <= RT_LDS_SPHERES && p->spp >= RT_ORDER_MIN_SPP && (groups < RT_SLICE_BELOW_BLOCKS || mixed_all);
        const bool tree_ordered = tree_scene && k.n_bvh_nodes > 0 && order_tree && p->spp >= RT_TREE_SLICE_SPP && groups * 4u < RT_TREE_SLICE_BELOW_WAVES;
        const bool ordered = !noalloc && !adaptive && groups >= 64 && (scan_ordered || tree_ordered) && !order_none;
        const rthip_ctx::OrderKey order_key = {p->width, p->height, rows, p->band_rows, p->band_first, p->band_stride, p->spp, p->max_depth,
                                               p->flags & (RTHIP_F_POINT_LIGHTS | RTHIP_F_COUNT_STATS | RTHIP_F_COUNT_EXECUTED), ctx->frame_gen};
        const bool order_measured = ordered && ctx->order_state != 0 && ctx->order_key == order_key && !order_model_only;  // this pass starts its groups by measured cost
        if (tree_ordered && order_measured) k.n_slices = 1;
        const bool mixed = scan_ordered && order_measured && full_slices > 1 && !no_mixed;
        if (mixed) k.n_slices = full_slices;  // (the partial sums are addressed as in a sliced launch; only the sliced groups' are touched)
        if (k.n_slices > 1) {
            const size_t need = size_t(k.n_slices) * rows * p->width * 3 * sizeof(float);
            if (noalloc) {
                if (need > RTHIP_NOALLOC_SCRATCH_BYTES - RT_NOALLOC_STATS_BYTES)
                    return fail(ctx, RTHIP_E_BADARG, "rthip_render_device_noalloc: this launch is sample-sliced and needs " + std::to_string(need) +
                                                         " bytes of partial sums; the static scratch holds " +
                                                         std::to_string(size_t(RTHIP_NOALLOC_SCRATCH_BYTES - RT_NOALLOC_STATS_BYTES)) +
                                                         " (use fewer rows per call, fewer samples per pass, or rthip_render_device)");
                k.partial = (float *)(scratch + RT_NOALLOC_STATS_BYTES);  // fully written by the render kernel before resolve_kernel reads it
            } else {
                if ((st = ensure(ctx, ctx->partial, need))) return st;
                k.partial = (float *)ctx->partial.p;
            }
        }

        // MODE_BVH: primary-visibility lists (rt_prepass.h) — the tree is walked once per 8x8 tile and the tile's primary rays
        // test the leaves of their pixel's list. Worth it from a few samples per pixel on (the walk costs about one primary
        // ray per pixel); the passes that continue a running sum reuse the lists of its frame (camera, scene and layout are
        // the same by construction), a pass that starts a frame makes its own. Not for work-list passes (RTHIP_F_ADAPTIVE).
        k.tile_pre = nullptr;
        bool make_lists = false;
        if (k.n_bvh_nodes > 0 && (p->flags & RTHIP_F_TILE_LISTS) && !adaptive) {
            const size_t need = size_t(groups) * 4u * sizeof(RtTilePre);
            if (accumulate && ctx->pre_lists) {
                k.tile_pre = (const RtTilePre *)ctx->pre_lists;
            } else if (p->spp >= RT_PRE_MIN_SPP || accumulate) {
                if (noalloc) {
                    // after the counter shards and the partial sums of a sliced launch, if the static scratch has the room
                    size_t at = RT_NOALLOC_STATS_BYTES;
                    if (k.n_slices > 1) at += size_t(k.n_slices) * rows * p->width * 3 * sizeof(float);
                    at = (at + 255u) & ~size_t(255);
                    if (at + need <= RTHIP_NOALLOC_SCRATCH_BYTES) k.tile_pre = (const RtTilePre *)(scratch + at);
                } else {
                    if ((st = ensure(ctx, ctx->tile_pre, need))) return st;
                    k.tile_pre = (const RtTilePre *)ctx->tile_pre.p;
                }
                make_lists = k.tile_pre != nullptr;
            }
        }
        if (!accumulate || !k.tile_pre) ctx->pre_lists = nullptr;

        // passes share the context's buffers (running sum, counters, partial sums, work-list): a pass enqueued on another
        // stream than the previous one waits for it
        if (ctx->have_last_pass && ctx->last_stream != stream) HIPCHK(ctx, hipStreamWaitEvent(stream, ctx->ev_last, 0));
        if (stats) HIPCHK(ctx, hipMemsetAsync(d_stats, 0, RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long), stream));  // the shards start at zero
        if (stats) HIPCHK(ctx, hipEventRecord(ctx->ev0, stream));
        if (adaptive) {
            const size_t npx = size_t(rows) * p->width;
            if ((st = ensure(ctx, ctx->pix_state, npx * sizeof(uint32_t)))) return st;
            if ((st = ensure(ctx, ctx->worklist, (npx + 1) * sizeof(uint32_t)))) return st;
            k.pix_state = (uint32_t *)ctx->pix_state.p;
            if (!accumulate) {
                // first pass: every PixelCache starts NotStarted (lib.rs:120-138)
                HIPCHK(ctx, hipMemsetAsync(ctx->pix_state.p, 0, npx * sizeof(uint32_t), stream));
            } else {
                uint32_t *count = (uint32_t *)ctx->worklist.p, *list = count + 1;
                HIPCHK(ctx, hipMemsetAsync(count, 0, sizeof(uint32_t), stream));
                HIPCHK(ctx, rt_launch_compact(&k, list, count, stream));
                k.worklist = list, k.worklist_count = count;
            }
        }
        // Cost-ordered launch (rt_kernels.hip, tile_cost_kernel / tile_order_kernel). The FIRST pass of a frame layout (image, bands,
        // scene, camera, samples) takes its order from a cost model (scan scenes: two tiny kernels ahead of the pass) or runs in row
        // order (tree scenes), and MEASURES: every wave reports its run time to its pixel group. The SECOND pass sorts those
        // measurements (one tiny kernel) and every later pass of the layout just reuses that order: the groups with the longest
        // waves first, by the clock. (The model alone: 86.5 -> 89.6 % of ideal on the 1/8-frame launches of C2 — it cannot know
        // which ground tiles' bounce rays reach the glass sphere, and 150 such workgroups starting late were the last 10 % of the
        // launch, tools/timeline_band.py; measured: 92 %.) A schedule only: which workgroup traces which group has no effect on the
        // image. A whole 4K frame of a scan scene gains nothing from either order and pays 1 % for the model's (the sort, and
        // neighbouring tiles no longer run side by side): RTHIP_ORDER_ALL=1 orders it too. Not for the 1-spp passes of the
        // progressive loop, not for work-list passes.
        k.tile_order = nullptr;
        k.tile_cost_out = nullptr;
        k.mixed = 0, k.head_groups = 0, k.whole_end = 0, k.n_groups = uint32_t(groups);
        if (ordered) {
            if ((st = ensure(ctx, ctx->tile_order, (size_t(groups) * 2u + 2u) * sizeof(uint32_t)))) return st;
            if ((st = ensure(ctx, ctx->tile_cost, size_t(groups) * sizeof(uint32_t)))) return st;
            uint32_t *order = (uint32_t *)ctx->tile_order.p, *split = order + 2u * groups;
            if (order_measured && ctx->order_state == 2) {
                // the measured order is in place
            } else if (order_measured) {
                HIPCHK(ctx, rt_launch_tile_order(&k, order, (const uint32_t *)ctx->tile_cost.p, mixed ? split : nullptr, ctx->wave_slots, ctx->order_meas_slices, stream));
                if (mixed) {  // the grid of a mixed launch depends on its two boundaries: read them, once per frame layout
                    uint32_t hs[2] = {0, 0};
                    HIPCHK(ctx, hipMemcpyAsync(hs, split, sizeof hs, hipMemcpyDeviceToHost, stream));
                    HIPCHK(ctx, hipStreamSynchronize(stream));
                    ctx->order_head = hs[0] <= hs[1] ? hs[0] : hs[1];
                    ctx->order_whole_end = hs[1] <= groups ? hs[1] : uint32_t(groups);
                    static const bool dbg = getenv("RTHIP_ORDER_DEBUG") != nullptr;
                    if (dbg) fprintf(stderr, "[rthip] mixed launch: %u groups = %u sliced + %u whole + %u sliced, %u slices\n", unsigned(groups), ctx->order_head,
                                     ctx->order_whole_end - ctx->order_head, unsigned(groups) - ctx->order_whole_end, k.n_slices);
                }
                ctx->order_state = 2;
            } else {
                HIPCHK(ctx, hipMemsetAsync(ctx->tile_cost.p, 0, size_t(groups) * sizeof(uint32_t), stream));
                // (a cost model for the launches that are sliced from the start; bigger ones and tree scenes start in row order)
                const bool model = !tree_scene && k.n_slices > 1;
                if (model) HIPCHK(ctx, rt_launch_tile_order(&k, order, nullptr, nullptr, ctx->wave_slots, 1u, stream));
                k.tile_cost_out = (uint32_t *)ctx->tile_cost.p;
                ctx->order_key = order_key;
                ctx->order_state = 1;
                ctx->order_meas_slices = k.n_slices;
                if (model) k.tile_order = order;
            }
            if (ctx->order_state == 2) {
                k.tile_order = order;
                if (mixed) k.mixed = 1u, k.head_groups = ctx->order_head, k.whole_end = ctx->order_whole_end;
            }
        }
        if (make_lists) {
            HIPCHK(ctx, rt_launch_prepass(&k, (void *)k.tile_pre, stream));
            ctx->pre_lists = noalloc ? nullptr : (const void *)k.tile_pre;  // (the static scratch is shared by every context of the GPU: not kept)
        }
        HIPCHK(ctx, rt_launch_render(&k, (p->flags & RTHIP_F_COUNT_STATS) ? 1 : 0, stream));
        if (stats) HIPCHK(ctx, hipEventRecord(ctx->ev1, stream));

        ctx->acc_width = p->width, ctx->acc_height = p->height, ctx->acc_rows = rows;
        ctx->acc_band_rows = p->band_rows, ctx->acc_band_first = p->band_first, ctx->acc_band_stride = p->band_stride;
        ctx->nb_samples = prior + p->spp;
        ctx->adaptive = adaptive;
        ctx->acc_buffer = acc_buffer;

        if (stats) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->stats_host, d_stats, RT_STAT_SHARDS * RT_STAT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            ctx->pending_stats = true, ctx->pending_adaptive = adaptive, ctx->pending_rows = rows, ctx->pending_width = p->width;
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev_last, stream));
        ctx->have_last_pass = true, ctx->last_stream = stream;
        if (stats && !defer_stats && (st = collect_stats(ctx, stats))) return st;
        return RTHIP_OK;
    } catch (...) {
        return fail(ctx, RTHIP_E_INTERNAL, "unexpected exception in the render pass");
    }
}

int rthip_render_device(rthip_ctx *ctx, const rthip_render_params *p, void *d_rgba8, void *d_accum, void *hip_stream, rthip_stats *stats) {
    return render_pass(ctx, p, d_rgba8, d_accum, hip_stream, stats, false);
}

// internal (rthip_group.cpp): enqueue a pass on the context's own stream without waiting for its counters / wait and read them
int rt_ctx_render_deferred(rthip_ctx *ctx, const rthip_render_params *p, void *d_rgba8, void *d_accum, rthip_stats *stats_wanted) {
    return render_pass(ctx, p, d_rgba8, d_accum, ctx ? ctx->stream : nullptr, stats_wanted, false, true);
}
int rt_ctx_collect_stats(rthip_ctx *ctx, rthip_stats *stats) {
    if (!ctx || !stats) return RTHIP_E_BADARG;
    return collect_stats(ctx, stats);
}
// internal (tests/test_prepass_gpu.py): the primary-visibility lists of the context's current frame, copied to `out`
// (at most `bytes`); returns the size of the lists in bytes, 0 if the last pass made or used none, -1 on a HIP error
long long rt_ctx_tile_lists(rthip_ctx *ctx, void *out, unsigned long long bytes) {
    if (!ctx || !ctx->pre_lists) return 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    const size_t tiles_x = (ctx->acc_width + RT_TILE_W - 1u) / RT_TILE_W, tiles_y = (ctx->acc_rows + RT_TILE_H - 1u) / RT_TILE_H;
    const size_t have = tiles_x * tiles_y * 4u * sizeof(RtTilePre);
    const size_t n = bytes < have ? size_t(bytes) : have;
    if (out && n && hipMemcpy(out, ctx->pre_lists, n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)have;
}
void *rt_ctx_stream(rthip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int rt_ctx_device(rthip_ctx *ctx) { return ctx ? ctx->device : -1; }
uint64_t rt_ctx_nb_samples(rthip_ctx *ctx) { return ctx ? ctx->nb_samples : 0; }

int rthip_render_device_noalloc(rthip_ctx *ctx, const rthip_render_params *p, void *d_rgba8, void *d_accum, void *hip_stream, rthip_stats *stats) {
    return render_pass(ctx, p, d_rgba8, d_accum, hip_stream, stats, true);
}

int rthip_render(rthip_ctx *ctx, const rthip_render_params *p, uint8_t *rgba8_out, float *accum_out, rthip_stats *stats) {
    if (!ctx) return RTHIP_E_BADARG;
    try {
        int st = validate_params(ctx, p);
        if (st) return st;
        uint32_t rows = 0;
        if ((st = owned_rows(p, &rows))) return fail(ctx, st, "bad band parameters");
        HIPCHK(ctx, hipSetDevice(ctx->device));
        const size_t npx = size_t(rows) * p->width;
        if (rgba8_out && npx && (st = ensure(ctx, ctx->staging_rgba, npx * 4))) return st;
        rthip_stats local;
        st = rthip_render_device(ctx, p, (rgba8_out && npx) ? ctx->staging_rgba.p : nullptr, nullptr, ctx->stream, &local);
        if (st) return st;
        hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
        HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
        if (rgba8_out && npx) HIPCHK(ctx, hipMemcpyAsync(rgba8_out, ctx->staging_rgba.p, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (accum_out && npx) HIPCHK(ctx, hipMemcpyAsync(accum_out, ctx->accum.p, npx * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
        local.d2h_ms = ms;
        if (stats) *stats = local;
        return RTHIP_OK;
    } catch (...) {
        return fail(ctx, RTHIP_E_INTERNAL, "unexpected exception in rthip_render");
    }
}

// The progressive window loop (standalone/src/main.rs:99-102) wants a frame per pass, and at 4K the copy of the frame (33 MB:
// ~0.6 ms into page-locked memory) is four times the pass itself. So: two frame slots. rthip_render_async enqueues the pass of
// a slot on the context's stream and the copy of its RGBA8 rows on a copy stream behind it, and returns; rthip_wait blocks
// until that slot's frame is in the caller's buffer. With two slots alternating, the copy of pass k runs beside the kernel of
// pass k+1 and a pass costs max(kernel, copy) instead of their sum. Passes run in call order (one stream); they may carry
// RTHIP_F_ACCUMULATE / RTHIP_F_ADAPTIVE like any other pass. No counters (the read-out would serialise the loop).
int rthip_render_async(rthip_ctx *ctx, const rthip_render_params *p, uint8_t *rgba8_out, uint32_t slot) {
    if (!ctx) return RTHIP_E_BADARG;
    try {
        if (slot > 1) return fail(ctx, RTHIP_E_BADARG, "rthip_render_async: slot is 0 or 1");
        if (!rgba8_out) return fail(ctx, RTHIP_E_BADARG, "rthip_render_async: rgba8_out is NULL");
        if (ctx->async_busy[slot]) return fail(ctx, RTHIP_E_STATE, "rthip_render_async: the slot still holds a frame (rthip_wait first)");
        int st = validate_params(ctx, p);
        if (st) return st;
        uint32_t rows = 0;
        if ((st = owned_rows(p, &rows))) return fail(ctx, st, "bad band parameters");
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (!ctx->copy_stream) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; i++) {
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_rendered[i], hipEventDisableTiming));
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copied[i], hipEventDisableTiming));
            }
        }
        const size_t bytes = size_t(rows) * p->width * 4;
        if (bytes && (st = ensure(ctx, ctx->staging_async[slot], bytes))) return st;
        st = rthip_render_device(ctx, p, bytes ? ctx->staging_async[slot].p : nullptr, nullptr, ctx->stream, nullptr);
        if (st) return st;
        HIPCHK(ctx, hipEventRecord(ctx->ev_rendered[slot], ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rendered[slot], 0));
        if (bytes) HIPCHK(ctx, hipMemcpyAsync(rgba8_out, ctx->staging_async[slot].p, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
        HIPCHK(ctx, hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
        ctx->async_busy[slot] = true;
        return RTHIP_OK;
    } catch (...) {
        return fail(ctx, RTHIP_E_INTERNAL, "unexpected exception in rthip_render_async");
    }
}

int rthip_wait(rthip_ctx *ctx, uint32_t slot) {
    if (!ctx) return RTHIP_E_BADARG;
    if (slot > 1) return fail(ctx, RTHIP_E_BADARG, "rthip_wait: slot is 0 or 1");
    if (!ctx->async_busy[slot]) return fail(ctx, RTHIP_E_STATE, "rthip_wait: no frame was started in this slot");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->async_busy[slot] = false;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_copied[slot]));
    return RTHIP_OK;
}

int rthip_host_alloc(size_t bytes, void **out_ptr) {
    if (!out_ptr || bytes == 0) return fail(nullptr, RTHIP_E_BADARG, "rthip_host_alloc: bad arguments");
    *out_ptr = nullptr;
    hipError_t e = hipHostMalloc(out_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? RTHIP_E_OOM : RTHIP_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return RTHIP_OK;
}

int rthip_host_free(void *ptr) {
    if (!ptr) return RTHIP_OK;
    hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(nullptr, RTHIP_E_HIP, std::string("hipHostFree: ") + hipGetErrorString(e));
    return RTHIP_OK;
}

int rthip_pick(rthip_ctx *ctx, uint32_t width, uint32_t height, double x, double y, int64_t *out_shape) {
    if (!ctx) return RTHIP_E_BADARG;
    if (!out_shape || width < 2 || height < 2) return fail(ctx, RTHIP_E_BADARG, "bad pick arguments");
    if (!ctx->have_scene || !ctx->have_camera) return fail(ctx, RTHIP_E_STATE, "scene/camera not uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // lib.rs:263-265: emit_ray_at(x/(width-1), y/(height-1))
    const float u = float(x / (double(width) - 1.0)), v = float(y / (double(height) - 1.0));
    KernelParams k = ctx->base;
    HIPCHK(ctx, rt_launch_pick(&k, u, v, (long long *)ctx->pick_out.p, ctx->stream));
    long long r = -1;
    HIPCHK(ctx, hipMemcpyAsync(&r, ctx->pick_out.p, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out_shape = r;
    return RTHIP_OK;
}

}  // extern "C"
