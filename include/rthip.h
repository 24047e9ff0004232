/*
 * rthip.h — C ABI of the MI355X (gfx950) render path for rustracer's raytracer_core.
 *
 * This is the drop-in boundary: the entry points are what a
 * `raytracer_hip` Rust crate binds with `extern "C"` (see INTEGRATION.md and
 * raytracer_hip/src/ffi.rs) to replace the per-pixel render loop
 *     raytracer_core/src/lib.rs:269-316   Raytracer::generate_pixel
 *     raytracer_core/src/camera.rs:43-50  Camera::emit_ray_at
 *     raytracer_core/src/shapes/ray.rs:42-93  project_ray / find_collision / background_color
 *     raytracer_core/src/shapes/sphere.rs:29-73, materials/{...}.rs  collide / scatter / bounce
 * All file:line citations are relative to /root/reference.
 *
 * Conventions
 *  - Every struct is POD with fixed-width members and explicit padding; no bool.
 *  - Host pointers inside a struct are BORROWED for the duration of the call and copied.
 *  - Return value: 0 = OK, negative = rthip_status. The message of the last failure on a
 *    context is available from rthip_last_error(). No C++ exception crosses the boundary.
 *  - A context is bound to ONE device and is NOT thread-safe (the reference has a single
 *    render thread, standalone/src/main.rs:65). Multi-GPU = one context per device/process;
 *    the row-band parameters of rthip_render_params select the rows a device owns.
 *  - Image rows are y-up as in the reference: row 0 of every output buffer is pos.y = 0,
 *    the BOTTOM row of the picture (raytracer_core/src/lib.rs:282-285; the display flips
 *    it in standalone/src/renderers/pixels.rs:243).
 *  - Scene geometry is given in double (the reference is f64 throughout); the device
 *    computes in FP32.
 */
#ifndef RTHIP_H
#define RTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTHIP_ABI_VERSION 4

typedef enum rthip_status {
    RTHIP_OK = 0,
    RTHIP_E_BADARG = -1,
    RTHIP_E_HIP = -2,
    RTHIP_E_OOM = -3,
    RTHIP_E_NODEVICE = -4,
    RTHIP_E_STATE = -5, /* call order: render before scene/camera upload */
    RTHIP_E_INTERNAL = -6,
    RTHIP_E_RCCL = -7 /* rthip_group_*: an RCCL call failed */
} rthip_status;

/* ---- materials: raytracer_core/src/materials/{lambertian_diffuse,metal,dielectric,texture}.rs ---- */
enum {
    RTHIP_MAT_LAMBERTIAN = 0, /* Lambertian::new(albedo)            lambertian_diffuse.rs:19 */
    RTHIP_MAT_METAL = 1,      /* Metal::new(albedo, fuzziness)      metal.rs:19              */
    RTHIP_MAT_DIELECTRIC = 2, /* Dielectric::new(albedo, ior)       dielectric.rs:20         */
    RTHIP_MAT_TEXTURE = 3     /* Texture::load_from_file(path,scale) texture.rs:21           */
};

typedef struct rthip_material {
    uint32_t kind;    /* RTHIP_MAT_* */
    uint32_t texture; /* TEXTURE: index into rthip_scene_desc.textures */
    double albedo[3]; /* the albedo AS PASSED to ::new (the constructors divide by PI themselves) */
    double param;     /* METAL: fuzziness; DIELECTRIC: refraction_idx; TEXTURE: scale */
    double shininess; /* extension (point-light mode only): Phong exponent, 0 = no specular lobe */
} rthip_material;

/* Decoded image of a Texture material (texture.rs:14-30): tightly packed RGB8, row 0 = top. */
typedef struct rthip_texture {
    uint32_t width;
    uint32_t height;
    const uint8_t *rgb8; /* width*height*3 bytes, borrowed */
} rthip_texture;

/* Sphere::new(center, radius, material)   raytracer_core/src/shapes/sphere.rs:12-25 */
typedef struct rthip_sphere {
    double center[3];
    double radius;
    uint32_t material;
    uint32_t _pad;
} rthip_sphere;

/* Extension (no reference counterpart; plugs in where `impl Shape` would, shapes/shape.rs:7-15). */
typedef struct rthip_plane {
    double point[3];
    double normal[3]; /* need not be unit; normalised on upload */
    uint32_t material;
    uint32_t _pad;
} rthip_plane;

/* Extension: triangle, counter-clockwise front face; texture UV = barycentric (b1, b2). */
typedef struct rthip_triangle {
    double v0[3];
    double v1[3];
    double v2[3];
    uint32_t material;
    uint32_t _pad;
} rthip_triangle;

/* Extension: point light for the optional direct-lighting mode (RTHIP_F_POINT_LIGHTS). */
typedef struct rthip_light {
    double position[3];
    double intensity[3];
} rthip_light;

/*
 * The scene: what the reference holds as `Scene = Vec<Box<dyn Shape>>` (lib.rs:75).
 * Shape index (tie-break of find_collision, ray.rs:51-65, and rthip_pick): spheres first in
 * array order, then planes, then triangles.
 */
typedef struct rthip_scene_desc {
    const rthip_sphere *spheres;
    uint64_t n_spheres;
    const rthip_plane *planes;
    uint64_t n_planes;
    const rthip_triangle *triangles;
    uint64_t n_triangles;
    const rthip_material *materials;
    uint64_t n_materials;
    const rthip_texture *textures;
    uint64_t n_textures;
    const rthip_light *lights;
    uint64_t n_lights;
} rthip_scene_desc;

/* Camera constants as recoverable from the reference's public API:
 * origin = camera.origin; d00 = emit_ray_at(0,0).direction();
 * horizontal = emit_ray_at(1,0).direction() - d00; vertical = emit_ray_at(0,1).direction() - d00. */
typedef struct rthip_camera {
    double origin[3];
    double d00[3]; /* lower_left_corner - origin   camera.rs:46 */
    double horizontal[3];
    double vertical[3];
} rthip_camera;

enum {
    RTHIP_F_ACCUMULATE = 1u << 0,   /* add to the context's running sum (lib.rs:289-294) instead of starting over */
    RTHIP_F_POINT_LIGHTS = 1u << 1, /* extension: direct light + shadow rays at every scattering hit */
    RTHIP_F_COUNT_STATS = 1u << 2,  /* run the instrumented kernel and fill rthip_stats counters */
    RTHIP_F_NO_BVH = 1u << 3,       /* force the brute-force linear scan (ray.rs:47-67) even for big scenes */
    RTHIP_F_ADAPTIVE = 1u << 4,     /* progressive passes with the convergence rule of lib.rs:276-313: a pixel whose 8-bit
                                       colour was the same in 5 consecutive passes (same_color_count > 3, lib.rs:244,309)
                                       becomes GenerationStatus::Final and is skipped by every later pass (lib.rs:277-279):
                                       no rays, running sum and colour unchanged. Passes after the first must also set
                                       RTHIP_F_ACCUMULATE; they trace a compacted work-list of the pixels not yet Final. */
    RTHIP_F_COUNT_EXECUTED = 1u << 6, /* with RTHIP_F_COUNT_STATS: the instrumented kernel runs the production kernel's loops, so the counters
                                       hold the shape tests actually EXECUTED (a primary ray is tested against the spheres its 8x8 tile can
                                       reach, and not at all in tiles that see only sky) instead of the reference's scan over every shape
                                       (ray.rs:47-67). Same image. For the "executed" roofline fraction beside the algorithmic one. */
    RTHIP_F_TILE_LISTS = 1u << 5    /* big scenes (BVH), experimental: primary rays test their pixel's primary-visibility list (one tree walk per
                                       8x8 tile, rt_prepass.h) instead of walking the tree themselves. Same image bit for bit; measured SLOWER
                                       than the plain traversal, kept for A/B runs and tests. Ignored without a tree. */
};

/* GenerationStatus (lib.rs:59-64) as reported by rthip_pixel_status */
enum { RTHIP_STATUS_NOT_STARTED = 0, RTHIP_STATUS_COPY_NEAR_PIXEL = 1 /* never produced: generator-side */, RTHIP_STATUS_UNSTABLE = 2, RTHIP_STATUS_FINAL = 3 };

typedef struct rthip_render_params {
    uint32_t width;       /* full image width  (u = (x+xi)/(width-1),  lib.rs:282-283) */
    uint32_t height;      /* full image height (v = (y+xi)/(height-1), lib.rs:284-285) */
    uint32_t spp;         /* samples per pixel of THIS pass (generate_pixel's `samples`); a running sum holds at most 2^24 */
    uint32_t max_depth;   /* 50 = reference (ray.rs:44); at most 255 (the counter RNG has 256 slots per sample) */
    uint32_t seed;        /* counter-RNG seed (replaces the entropy-seeded SmallRng) */
    uint32_t flags;       /* RTHIP_F_* */
    uint32_t band_rows;   /* rows per band R; 0 = the whole image is one band owned by this context */
    uint32_t band_first;  /* first band owned by this context */
    uint32_t band_stride; /* this context owns bands band_first + k*band_stride (k = 0,1,...) */
    uint32_t reserved[7]; /* must be 0 */
} rthip_render_params;

typedef struct rthip_stats {
    uint64_t paths;        /* primary paths traced this pass */
    uint64_t rays;         /* ray segments = find_collision-equivalent queries (primary+bounce+shadow) */
    uint64_t sphere_tests; /* RTHIP_F_COUNT_STATS only */
    uint64_t plane_tests;
    uint64_t tri_tests;
    uint64_t node_tests;
    uint64_t hits[4];      /* RTHIP_F_COUNT_STATS only: closest hits by material kind (RTHIP_MAT_*) */
    uint64_t sky_evals;    /* RTHIP_F_COUNT_STATS only: background_color evaluations (ray.rs:89-93) */
    uint64_t shadow_rays;  /* RTHIP_F_COUNT_STATS only: any-hit shadow segments (included in `rays`) */
    uint64_t rows;         /* rows rendered by this context this pass */
    uint64_t nb_samples;   /* samples accumulated per pixel so far (PixelCache.nb_samples, lib.rs:72) */
    double kernel_ms;      /* device time of the render kernel(s), HIP events */
    double d2h_ms;         /* device->host copy of the outputs (rthip_render only) */
    uint64_t pixels_traced; /* RTHIP_F_ADAPTIVE: pixels that received samples this pass (all owned pixels otherwise) */
    uint64_t pixels_final;  /* RTHIP_F_ADAPTIVE: owned pixels that are Final after this pass */
    double gather_ms;       /* rthip_group_render only: from the end of device 0's render to the assembled frame on device 0
                               (waits for the slowest device, RCCL gather, de-interleave) */
} rthip_stats;

typedef struct rthip_ctx rthip_ctx;

/* Number of visible HIP devices. */
int rthip_device_count(int *out_count);

/* Create a context on `device`. Replaces Raytracer::new (lib.rs:250). */
int rthip_create(int device, rthip_ctx **out_ctx);

/* Flatten the scene to device SoA buffers (and a BVH once spheres + 4 x triangles exceed 96: below that the linear scan is faster).
 * Replaces the borrowed `&[Box<dyn Shape>]` argument of generate_pixel (lib.rs:272). Waits for every pass still running
 * on the device (also those enqueued on caller streams by rthip_render_device) before touching the scene buffers.
 * RTHIP_E_BADARG for coordinates or radii that are not finite or exceed 1e18 in magnitude (their squares must stay finite
 * in the device's FP32), zero radii, zero plane normals, material / texture indices out of range. */
int rthip_upload_scene(rthip_ctx *ctx, const rthip_scene_desc *scene);

/* Replaces the `pub camera: Camera` field (lib.rs:240) read by emit_ray_at (camera.rs:43-50).
 * Changing the camera invalidates the running sum, like Command::{Move,Rotate} does in
 * standalone/src/main.rs:103-120. */
int rthip_set_camera(rthip_ctx *ctx, const rthip_camera *camera);

/* Camera::new_lookat (camera.rs:19-41) evaluated in double on the host. */
int rthip_camera_lookat(double x, double y, double z, const double lookat[3], rthip_camera *out);

/* Number of rows the context owns for these parameters (output buffers hold rows*width pixels). */
int rthip_owned_rows(const rthip_render_params *params, uint32_t *out_rows);

/*
 * Render one pass of `spp` samples for every owned pixel. Replaces the loop
 *   loop { generate_pixel(gen, scene, spp, renderer); if gen.next().is_none() break }
 * (wasm/src/lib.rs:112-123). Synchronous; outputs are HOST buffers holding the owned rows in
 * increasing y, `width` pixels each:
 *   rgba8_out: owned_rows*width*4 bytes, R,G,B = the u8 of lib.rs:294-299/30-38, A = 255; may be NULL
 *   accum_out: owned_rows*width*3 floats = the running radiance sum (PixelCache.incremental_raw_light); may be NULL
 *   stats:     may be NULL
 */
int rthip_render(rthip_ctx *ctx, const rthip_render_params *params, uint8_t *rgba8_out, float *accum_out,
                 rthip_stats *stats);

/*
 * Same pass, outputs left in DEVICE memory, enqueued on `hip_stream` (a hipStream_t, NULL = the
 * null stream) without synchronising. d_rgba8 / d_accum are device pointers of the sizes above
 * (either may be NULL). Used by callers that own device buffers (the multi-GPU gather, benches).
 * kernel_ms in `stats` is filled only if `stats` is non-NULL, which makes the call synchronous.
 * Passes of one context share its running sum and work buffers: a pass enqueued on a different stream
 * than the previous pass of the context first waits (on the device) for that pass.
 */
int rthip_render_device(rthip_ctx *ctx, const rthip_render_params *params, void *d_rgba8, void *d_accum,
                        void *hip_stream, rthip_stats *stats);

/*
 * The same pass for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc; under HIP graph
 * capture only for passes WITHOUT RTHIP_F_ACCUMULATE: the sample offset and 1/nb_samples are baked into the launch when it is
 * captured, so a replayed graph re-traces the same samples — the same frame again, never a continued running sum). No counterpart in the reference. Calls no hipMalloc / hipFree / allocating routine:
 *   - the running sum lives in the CALLER's d_accum (required; owned_rows*width*3 floats): RTHIP_F_ACCUMULATE adds to what
 *     that buffer holds from the previous pass of this context (same buffer, same layout, else RTHIP_E_STATE);
 *   - the counter shards behind `stats` and the partial sums of a sample-sliced launch (spp > 64 on a launch of fewer than
 *     20 000 workgroup tiles of 256 pixels) live in a static device array of RTHIP_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends.
 *     A sliced launch needs ceil(spp/slice)*owned_rows*width*12 bytes of it; more than it holds -> RTHIP_E_BADARG.
 *   - RTHIP_F_ADAPTIVE -> RTHIP_E_BADARG (its per-pixel state is context-owned memory that grows with the image).
 * Every call on one GPU, from any context, shares that GPU's array: calls that overlap in time on one GPU race (serialise
 * them on one stream, or wait for the previous call). Image bytes and sums are identical to rthip_render_device's.
 */
#define RTHIP_NOALLOC_SCRATCH_BYTES (64u << 20)
int rthip_render_device_noalloc(rthip_ctx *ctx, const rthip_render_params *params, void *d_rgba8, void *d_accum,
                                void *hip_stream, rthip_stats *stats);

/*
 * The same pass with the frame copy taken off the critical path, for the progressive window loop
 * (standalone/src/main.rs:99-102: one pass of 1 spp per iteration, forever). Two frame slots: rthip_render_async enqueues the
 * pass and the copy of its RGBA8 rows into rgba8_out (owned_rows*width*4 bytes; page-locked memory from rthip_host_alloc makes
 * the copy asynchronous and 5x faster) and returns at once; rthip_wait(slot) returns when that frame is in rgba8_out.
 * Alternating the slots, the copy of one pass runs beside the kernel of the next: a 4K 1-spp pass costs the copy (0.6 ms)
 * instead of copy + kernel + two synchronisations. Passes execute in call order and may carry RTHIP_F_ACCUMULATE /
 * RTHIP_F_ADAPTIVE; no counters. A slot holds one frame: RTHIP_E_STATE if it is started again before rthip_wait.
 * rgba8_out must stay valid until rthip_wait; rthip_upload_scene / rthip_destroy wait for everything in flight.
 */
int rthip_render_async(rthip_ctx *ctx, const rthip_render_params *params, uint8_t *rgba8_out, uint32_t slot);
int rthip_wait(rthip_ctx *ctx, uint32_t slot);

/* Forget the running sum: PixelRenderer::invalidate_pixels / RandomGenerator::invalidate_pixels
 * (lib.rs:79,120-138). */
int rthip_reset_accum(rthip_ctx *ctx);

/* RTHIP_F_ADAPTIVE: PixelColor.status (lib.rs:17-22,312) of every owned pixel after the last pass, one
 * byte per pixel (RTHIP_STATUS_*), owned_rows*width bytes in the layout of the image outputs. This is what
 * the reference hands to PixelRenderer::set_pixel beside r,g,b (standalone/src/renderers/pixels.rs:254 draws it).
 * same_color_count_out (nullable): PixelCache.same_color_count (lib.rs:71). RTHIP_E_STATE if no adaptive pass ran. */
int rthip_pixel_status(rthip_ctx *ctx, uint8_t *status_out, uint8_t *same_color_count_out, uint64_t capacity_pixels);

/* Layout of the context's running sum (PixelCache.incremental_raw_light / nb_samples, lib.rs:67-73) after the last pass:
 * owned rows, width, samples per pixel so far (all 0 after a reset / camera / scene change). Sizes the buffers of
 * rthip_pixel_status, whose capacity_pixels (the pixels each buffer can hold) must be >= rows*width (RTHIP_E_BADARG). */
int rthip_accum_layout(rthip_ctx *ctx, uint32_t *out_rows, uint32_t *out_width, uint64_t *out_nb_samples);

/* Optional: page-locked host memory for the image outputs of rthip_render. The reference's framebuffer is a plain Vec
 * (standalone/src/renderers/pixels.rs:190-209); a pageable destination makes the device->host copy of a 4K RGBA8 frame
 * ~3.3 ms, a page-locked one ~0.7 ms. Any pointer works with rthip_render; these only make it faster. */
int rthip_host_alloc(size_t bytes, void **out_ptr);
int rthip_host_free(void *ptr);

/* Raytracer::get_shape (lib.rs:262-267): un-jittered primary ray through pixel (x, y) of a
 * width x height image; *out_shape = index of the closest shape, or -1 for a miss. */
int rthip_pick(rthip_ctx *ctx, uint32_t width, uint32_t height, double x, double y, int64_t *out_shape);

/*
 * Algorithmic FP32 flops of a pass, from the counters of an RTHIP_F_COUNT_STATS run and the
 * per-operation constants RTHIP_FLOPS_* below (+,-,*,/,sqrt,min/max,compare-select = 1,
 * FMA = 2, RNG integer work = 0). Used for the "% of FP32 VALU roofline" figure.
 */
double rthip_algorithmic_flops(const rthip_stats *stats);

#define RTHIP_FLOPS_RAY_SETUP 16.0    /* per path: u,v (4) + direction (12)            lib.rs:282-286, camera.rs:43-50 */
#define RTHIP_FLOPS_SEGMENT 5.0       /* per segment: a = d.d                          sphere.rs:31 */
#define RTHIP_FLOPS_SPHERE_TEST 18.0  /* oc 3, half_b 5, c 7, discriminant 3           sphere.rs:30-34 */
#define RTHIP_FLOPS_PLANE_TEST 11.0   /* extension */
#define RTHIP_FLOPS_TRI_TEST 48.0     /* extension: Moller-Trumbore */
#define RTHIP_FLOPS_NODE_TEST 24.0    /* extension: AABB slab test */
#define RTHIP_FLOPS_HIT 19.0          /* sqrt + 2 roots 7, ray.at 6, normal 6          sphere.rs:37-65 */
#define RTHIP_FLOPS_SKY 20.0          /* normalize + lerp                              ray.rs:89-93 */
#define RTHIP_FLOPS_LAMBERTIAN 18.0   /* throughput 3, n + unit vector 12, sin/cos/sqrt 3   lambertian_diffuse.rs:39-73 */
#define RTHIP_FLOPS_METAL 36.0        /* throughput 3, normalize+reflect+fuzz+test 30, sin/cos/sqrt 3   metal.rs:27-66 */
#define RTHIP_FLOPS_DIELECTRIC 73.0   /* throughput 3, reflect/refract/schlick 70      dielectric.rs:28-98 */
#define RTHIP_FLOPS_TEXTURE 15.0      /* throughput 3, uv + wrap 12 (atan2/acos = 1 each)   sphere.rs:67-73, texture.rs:32-57 */
#define RTHIP_FLOPS_TONEMAP 12.0      /* per pixel: 4 per channel                      lib.rs:294-298 */

/*
 * ---- single-process multi-GPU ----------------------------------------------------------------
 * The reference's caller is ONE render thread that wants the whole frame from one call (standalone/src/main.rs:65,99-102;
 * wasm/src/lib.rs:112-123). A group owns one context + stream per listed device; the image is split into interleaved row
 * bands (band b = rows [b*R, (b+1)*R), device b mod ndev), the scene is replicated, and the bands are gathered to the first
 * device over RCCL (ncclCommInitAll once in rthip_group_create; grouped ncclSend/ncclRecv per frame, one transfer per xGMI
 * link), de-interleaved there and copied out. The RNG is keyed by the absolute pixel index, so the frame is byte-identical
 * to rthip_render's for any device count. No counterpart in the reference (it has no device or process boundary).
 */
typedef struct rthip_group rthip_group;
enum {
    RTHIP_GROUP_F_PEER_COPY = 1u << 0, /* gather with hipMemcpyPeerAsync instead of RCCL (also allows listing a device more than once:
                                          rehearsals of an N-way split on fewer GPUs) */
    RTHIP_GROUP_F_FORCE_RCCL = 1u << 1 /* use RCCL also for a single device (the gather is then a self-transfer) */
};
int rthip_group_create(const int *devices, int ndev, uint32_t flags, rthip_group **out_group);
int rthip_group_upload_scene(rthip_group *group, const rthip_scene_desc *scene); /* rthip_upload_scene on every device */
int rthip_group_set_camera(rthip_group *group, const rthip_camera *camera);
/* One pass over the WHOLE frame. params: band_rows = rows per band (0 = 8), band_first / band_stride must be 0.
 * rgba8_out: height*width*4 bytes, accum_out: height*width*3 floats (either may be NULL), both host buffers, row 0 = bottom.
 * stats (nullable): counters summed over devices, kernel_ms = the slowest device, gather_ms. RTHIP_F_ACCUMULATE works as in
 * rthip_render (every device keeps the running sum of its rows), and so does RTHIP_F_ADAPTIVE — the progressive loop of
 * standalone/src/main.rs:99-120 with the convergence rule of lib.rs:276-313 on N GPUs: every device keeps the PixelCache state
 * and the work-list of its own rows.
 * A frame that fails after its passes were enqueued (an RCCL or HIP error in the gather) leaves the group usable: the RCCL group
 * call is always closed, every device is synchronised and every running sum is reset (the next frame starts over). */
int rthip_group_render(rthip_group *group, const rthip_render_params *params, uint8_t *rgba8_out, float *accum_out, rthip_stats *stats);
/* rthip_pixel_status for the whole frame of the last rthip_group_render (RTHIP_F_ADAPTIVE): height*width bytes each, rows in image order. */
int rthip_group_pixel_status(rthip_group *group, uint8_t *status_out, uint8_t *same_color_count_out, uint64_t capacity_pixels);
int rthip_group_reset_accum(rthip_group *group);
int rthip_group_size(const rthip_group *group);
rthip_ctx *rthip_group_context(rthip_group *group, int index); /* e.g. rthip_pick on device 0; owned by the group */
const char *rthip_group_last_error(const rthip_group *group);
void rthip_group_destroy(rthip_group *group);
/* The row map of the band split: image row y is local row *out_local_row of device *out_rank. */
int rthip_band_row_owner(uint32_t band_rows, uint32_t ndev, uint32_t y, uint32_t *out_rank, uint32_t *out_local_row);

const char *rthip_last_error(const rthip_ctx *ctx);
const char *rthip_status_string(int status);
int rthip_abi_version(void);

void rthip_destroy(rthip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTHIP_H */
