// oracle_capi.cpp — C entry points over rt_oracle.hpp so tests/bench can drive the CPU
// restatement through ctypes. TEST INFRASTRUCTURE ONLY (see the header of rt_oracle.hpp).
#include <omp.h>

#include <chrono>

#include "rt_oracle.hpp"

using namespace oracle;

static bool valid_render_args(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_render_params *p) {
    if (!scene || !cam || !p || p->width < 2 || p->height < 2 || p->spp == 0) return false;
    for (uint64_t i = 0; i < scene->n_spheres; i++)
        if (scene->spheres[i].material >= scene->n_materials) return false;
    for (uint64_t i = 0; i < scene->n_planes; i++)
        if (scene->planes[i].material >= scene->n_materials) return false;
    for (uint64_t i = 0; i < scene->n_triangles; i++)
        if (scene->triangles[i].material >= scene->n_materials) return false;
    for (uint64_t i = 0; i < scene->n_materials; i++)
        if (scene->materials[i].kind == RTHIP_MAT_TEXTURE && scene->materials[i].texture >= scene->n_textures) return false;
    return true;
}

extern "C" {

int oracle_hw_threads(void) { return omp_get_max_threads(); }

// precision: 64 = reference-faithful double, 32 = the same formulas in float.
// sum_io: crop_w*crop_h*3 doubles (running radiance sum; input too when RTHIP_F_ACCUMULATE).
// Returns 0, or -1 on a bad argument. *seconds = wall time of the render loop (scene set-up excluded).
int oracle_render(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_render_params *p, const uint32_t *crop,
                  uint64_t prior_samples, int precision, int threads, uint8_t *rgba8, double *sum_io, rthip_stats *stats,
                  double *seconds) {
    if (!valid_render_args(scene, cam, p)) return -1;
    auto t0 = std::chrono::steady_clock::now();
    if (precision == 64)
        render<double>(*scene, *cam, *p, crop, prior_samples, threads, rgba8, sum_io, stats);
    else if (precision == 32)
        render<float>(*scene, *cam, *p, crop, prior_samples, threads, rgba8, sum_io, stats);
    else
        return -1;
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return 0;
}

// oracle_render with the optional accelerator (rt_oracle.hpp, build_accel): the scan's answers from a bounding-volume tree
// over the spheres and triangles — for scenes the scan cannot finish (1e6 triangles). *seconds = the render loop alone
// (tree build excluded, like the device's BVH build is not part of its kernel time).
int oracle_render_accel(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_render_params *p, const uint32_t *crop,
                        uint64_t prior_samples, int precision, int threads, int accel, uint8_t *rgba8, double *sum_io, rthip_stats *stats,
                        double *seconds) {
    if (!accel) return oracle_render(scene, cam, p, crop, prior_samples, precision, threads, rgba8, sum_io, stats, seconds);
    if (!valid_render_args(scene, cam, p)) return -1;
    if (precision == 64)
        render<double>(*scene, *cam, *p, crop, prior_samples, threads, rgba8, sum_io, stats, nullptr, 1, seconds);
    else if (precision == 32)
        render<float>(*scene, *cam, *p, crop, prior_samples, threads, rgba8, sum_io, stats, nullptr, 1, seconds);
    else
        return -1;
    return 0;
}

// The same with the PixelCache arrays of the progressive rule (lib.rs:276-313): one call = one pass over every pixel.
// status/same_color_count/last_color(3 per pixel)/nb_samples: crop_w*crop_h entries, zeroed by the caller before pass 1.
int oracle_render_adaptive(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_render_params *p, const uint32_t *crop,
                           int precision, int threads, uint8_t *rgba8, double *sum_io, uint8_t *status, uint8_t *same_color_count,
                           uint8_t *last_color, uint64_t *nb_samples, rthip_stats *stats) {
    if (!scene || !cam || !p || p->width < 2 || p->height < 2 || p->spp == 0 || !status || !same_color_count || !last_color || !nb_samples) return -1;
    PixelCaches c{status, same_color_count, last_color, nb_samples};
    if (precision == 64)
        render<double>(*scene, *cam, *p, crop, 0, threads, rgba8, sum_io, stats, &c);
    else if (precision == 32)
        render<float>(*scene, *cam, *p, crop, 0, threads, rgba8, sum_io, stats, &c);
    else
        return -1;
    return 0;
}

// ---- known-answer accessors ----
void oracle_camera_lookat(double x, double y, double z, const double lookat[3], rthip_camera *out) { camera_lookat(x, y, z, lookat, out); }

void oracle_emit_ray_at(const rthip_camera *cam, double ox, double oy, double origin[3], double dir[3]) {
    Ray<double> r = emit_ray_at<double>(*cam, ox, oy);
    origin[0] = r.origin.x, origin[1] = r.origin.y, origin[2] = r.origin.z;
    dir[0] = r.direction.x, dir[1] = r.direction.y, dir[2] = r.direction.z;
}

// find_collision of one ray; returns shape index or -1. precision 32/64.
int64_t oracle_find_collision(const rthip_scene_desc *scene, const double origin[3], const double dir[3], int precision, double *t,
                              double pos[3]) {
    if (precision == 64) {
        Scene<double> s(*scene);
        Hit<double> h;
        Ray<double> r{V3<double>(origin), V3<double>(dir)};
        if (!s.find_collision(r, &h, nullptr)) return -1;
        *t = h.t;
        pos[0] = h.position.x, pos[1] = h.position.y, pos[2] = h.position.z;
        return h.shape;
    }
    Scene<float> s(*scene);
    Hit<float> h;
    Ray<float> r{V3<float>(origin), V3<float>(dir)};
    if (!s.find_collision(r, &h, nullptr)) return -1;
    *t = h.t;
    pos[0] = h.position.x, pos[1] = h.position.y, pos[2] = h.position.z;
    return h.shape;
}

void oracle_sphere_uv(const double center[3], double radius, const double pos[3], double *u, double *v) {
    rthip_sphere sp;
    std::memcpy(sp.center, center, sizeof sp.center);
    sp.radius = radius;
    sp.material = 0;
    rthip_material m{};
    rthip_scene_desc d{};
    d.spheres = &sp, d.n_spheres = 1, d.materials = &m, d.n_materials = 1;
    Scene<double> s(d);
    Hit<double> h{0.0, V3<double>(pos), 0, 0.0, 0.0};
    s.texture_coords(h, u, v);
}

uint32_t oracle_texture_wrap(double val, double scale, uint32_t bound) { return texture_wrap<double>(val, scale, bound); }

void oracle_background_color(const double dir[3], double rgb[3]) {
    Ray<double> r{V3<double>(), V3<double>(dir)};
    V3<double> c = background_color(r);
    rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
}

uint8_t oracle_tonemap(double sum, uint64_t nb_samples) { return tonemap_channel<double>(sum, nb_samples); }
uint8_t oracle_tonemap_f32(float sum, uint64_t nb_samples) { return tonemap_channel<float>(sum, nb_samples); }
uint8_t oracle_cast_u8(double v) { return cast_u8(v); }
int32_t oracle_cast_i32(double v) { return cast_i32(v); }

double oracle_schlick(double cosine, double ior) { return schlick<double>(cosine, ior); }

int oracle_refract(const double ray[3], const double normal[3], double eta, double out[3]) {
    V3<double> o;
    if (!refract(V3<double>(ray), V3<double>(normal), eta, &o)) return 0;
    out[0] = o.x, out[1] = o.y, out[2] = o.z;
    return 1;
}
void oracle_reflect(const double ray[3], const double normal[3], double out[3]) {
    V3<double> o = reflect(V3<double>(ray), V3<double>(normal));
    out[0] = o.x, out[1] = o.y, out[2] = o.z;
}

uint32_t oracle_path_key(uint32_t seed, uint32_t pixel_index, uint32_t sample) { return path_key(seed, pixel_index, sample); }
uint32_t oracle_rng_first(uint32_t key, uint32_t slot) { return rng_first(key, slot); }
uint32_t oracle_rng_second(uint32_t h) { return rng_second(h); }
void oracle_rng_pair(uint32_t key, uint32_t slot, double *ua, double *ub) { rng_pair<double>(key, slot, ua, ub); }
void oracle_random_unit_vector(uint32_t key, uint32_t counter, double out[3]) {
    V3<double> v = random_unit_vector<double>(key, counter);
    out[0] = v.x, out[1] = v.y, out[2] = v.z;
}

// radiance of ONE path from an explicit ray (lets tests probe project_ray in isolation)
void oracle_project_ray(const rthip_scene_desc *scene, const double origin[3], const double dir[3], uint32_t max_depth, uint32_t key,
                        int point_lights, double rgb[3], uint64_t *segments) {
    Scene<double> s(*scene);
    rthip_stats st{};
    V3<double> c = project_ray<double>(s, Ray<double>{V3<double>(origin), V3<double>(dir)}, max_depth, key, point_lights != 0, &st);
    rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
    if (segments) *segments = st.rays;
}

// debug: trace the hit sequence of path (x, y, sample) in either precision; returns the number of segments
int oracle_trace_path(const rthip_scene_desc *scene, const rthip_camera *cam, const rthip_render_params *p, uint32_t x, uint32_t y,
                      uint32_t sample, int precision, int64_t *shapes, double *ts, double *dlens, double rgb[3]) {
    PathTrace tr;
    g_trace = &tr;
    const uint32_t key = path_key(p->seed, y * p->width + x, sample);
    if (precision == 64) {
        Scene<double> s(*scene);
        double jx, jy;
        rng_pair<double>(key, 0, &jx, &jy);
        double ox = (double(x) + jx) / (double(p->width) - 1.0);
        double oy = (double(y) + jy) / (double(p->height) - 1.0);
        V3<double> c = project_ray<double>(s, emit_ray_at<double>(*cam, ox, oy), p->max_depth, key, false, nullptr);
        rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
    } else {
        Scene<float> s(*scene);
        float jx, jy;
        rng_pair<float>(key, 0, &jx, &jy);
        float ox = (float(x) + jx) / (float(p->width) - 1.0f);
        float oy = (float(y) + jy) / (float(p->height) - 1.0f);
        V3<float> c = project_ray<float>(s, emit_ray_at<float>(*cam, ox, oy), p->max_depth, key, false, nullptr);
        rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
    }
    g_trace = nullptr;
    for (int i = 0; i < tr.n; i++) shapes[i] = tr.shape[i], ts[i] = tr.t[i], dlens[i] = tr.dlen[i];
    return tr.n;
}

}  // extern "C"
