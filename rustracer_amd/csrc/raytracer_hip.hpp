// raytracer_hip.hpp — header-only C++ host layer above the C ABI (include/rthip.h), mirroring the
// reference's scene / render API name for name, so a C++ front-end reads like the Rust ones:
//
//   Sphere::new(center, radius, Box<dyn Material>)      raytracer_core/src/shapes/sphere.rs:19
//   Lambertian::new / new_from_hex                      raytracer_core/src/materials/lambertian_diffuse.rs:19,26
//   Metal::new(albedo, fuzziness)                       raytracer_core/src/materials/metal.rs:19
//   Dielectric::new(albedo, refraction_idx)             raytracer_core/src/materials/dielectric.rs:20
//   Texture::load_from_file(path, scale)                raytracer_core/src/materials/texture.rs:21
//   Scene = Vec<Box<dyn Shape>>                         raytracer_core/src/lib.rs:75
//   Camera::{new, new_lookat, emit_ray_at}              raytracer_core/src/camera.rs:16-50
//   trait PixelRenderer { set_pixel, invalidate_pixels } raytracer_core/src/lib.rs:77-80
//   Raytracer::{new, camera, generate_pixel, get_shape} raytracer_core/src/lib.rs:236-316
//
// The reference is Rust and its toolchain is absent from the build image, so this C++ layer (and
// the Python one in rustracer_amd/) is the tested host side; raytracer_hip/ holds the Rust crate
// a maintainer would compile against the same ABI.
#pragma once

#include <zlib.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rthip.h"

namespace raytracer_hip {

using Vector3 = std::array<double, 3>;

struct HipError : std::runtime_error {
    int status;
    HipError(int st, const std::string &msg) : std::runtime_error("rthip status " + std::to_string(st) + ": " + msg), status(st) {}
};

// ---- materials --------------------------------------------------------------------------
struct Material {
    virtual ~Material() = default;
    virtual rthip_material record(uint32_t texture_index) const = 0;
};
struct Lambertian : Material {
    Vector3 albedo;
    explicit Lambertian(Vector3 a) : albedo(a) {}
    static std::shared_ptr<Lambertian> create(Vector3 a) { return std::make_shared<Lambertian>(a); }
    static std::shared_ptr<Lambertian> new_from_hex(int64_t color) {  // lambertian_diffuse.rs:26-35
        return create({double((color & 0xFF0000) >> 16) / 255.0, double((color & 0x00FF00) >> 8) / 255.0, double(color & 0x0000FF) / 255.0});
    }
    rthip_material record(uint32_t) const override { return rthip_material{RTHIP_MAT_LAMBERTIAN, 0, {albedo[0], albedo[1], albedo[2]}, 0.0, 0.0}; }
};
struct Metal : Material {
    Vector3 albedo;
    double fuzziness, shininess;
    Metal(Vector3 a, double f, double s = 0.0) : albedo(a), fuzziness(f), shininess(s) {}
    static std::shared_ptr<Metal> create(Vector3 a, double f) { return std::make_shared<Metal>(a, f); }
    rthip_material record(uint32_t) const override { return rthip_material{RTHIP_MAT_METAL, 0, {albedo[0], albedo[1], albedo[2]}, fuzziness, shininess}; }
};
struct Dielectric : Material {
    Vector3 albedo;
    double refraction_idx;
    Dielectric(Vector3 a, double ior) : albedo(a), refraction_idx(ior) {}
    static std::shared_ptr<Dielectric> create(Vector3 a, double ior) { return std::make_shared<Dielectric>(a, ior); }
    rthip_material record(uint32_t) const override { return rthip_material{RTHIP_MAT_DIELECTRIC, 0, {albedo[0], albedo[1], albedo[2]}, refraction_idx, 0.0}; }
};
struct Texture : Material {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb8;  // row 0 = top
    double scale = 1.0;
    // Texture::load_from_file (texture.rs:21-30). Decodes the build's texel fixture format (*.rgb8.z: u32 width, u32 height,
    // zlib(RGB8)) and binary PPM (P6); throws like ImageResult::Err on anything else.
    static std::shared_ptr<Texture> load_from_file(const std::string &path, double scale) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open texture " + path);
        std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto t = std::make_shared<Texture>();
        t->scale = scale;
        if (path.size() > 7 && path.compare(path.size() - 7, 7, ".rgb8.z") == 0) {
            if (blob.size() < 8) throw std::runtime_error("truncated texture " + path);
            std::memcpy(&t->width, blob.data(), 4);
            std::memcpy(&t->height, blob.data() + 4, 4);
            uLongf n = uLongf(t->width) * t->height * 3;
            t->rgb8.resize(n);
            if (uncompress(t->rgb8.data(), &n, blob.data() + 8, uLong(blob.size() - 8)) != Z_OK || n != t->rgb8.size())
                throw std::runtime_error("corrupt texture " + path);
            return t;
        }
        unsigned w = 0, h = 0, maxv = 0;
        int off = 0;
        if (std::sscanf(reinterpret_cast<const char *>(blob.data()), "P6 %u %u %u%n", &w, &h, &maxv, &off) == 3 && maxv == 255) {
            t->width = w, t->height = h;
            t->rgb8.assign(blob.begin() + off + 1, blob.end());
            if (t->rgb8.size() != size_t(w) * h * 3) throw std::runtime_error("corrupt PPM " + path);
            return t;
        }
        throw std::runtime_error("unsupported texture format " + path);
    }
    rthip_material record(uint32_t ti) const override { return rthip_material{RTHIP_MAT_TEXTURE, ti, {0, 0, 0}, scale, 0.0}; }
};

// ---- shapes -----------------------------------------------------------------------------
struct Shape {
    std::shared_ptr<Material> material;
    virtual ~Shape() = default;
};
struct Sphere : Shape {
    Vector3 center;
    double radius;
    static std::shared_ptr<Sphere> create(Vector3 c, double r, std::shared_ptr<Material> m) {
        auto s = std::make_shared<Sphere>();
        s->center = c, s->radius = r, s->material = std::move(m);
        return s;
    }
};
struct Plane : Shape {  // extension
    Vector3 point, normal;
    static std::shared_ptr<Plane> create(Vector3 p, Vector3 n, std::shared_ptr<Material> m) {
        auto s = std::make_shared<Plane>();
        s->point = p, s->normal = n, s->material = std::move(m);
        return s;
    }
};
struct Triangle : Shape {  // extension
    Vector3 v0, v1, v2;
    static std::shared_ptr<Triangle> create(Vector3 a, Vector3 b, Vector3 c, std::shared_ptr<Material> m) {
        auto s = std::make_shared<Triangle>();
        s->v0 = a, s->v1 = b, s->v2 = c, s->material = std::move(m);
        return s;
    }
};
using Scene = std::vector<std::shared_ptr<Shape>>;

// The POD description of include/rthip.h with the storage it points into.
struct SceneDesc {
    std::vector<rthip_sphere> spheres;
    std::vector<rthip_plane> planes;
    std::vector<rthip_triangle> triangles;
    std::vector<rthip_material> materials;
    std::vector<rthip_texture> textures;
    std::vector<rthip_light> lights;
    std::vector<std::shared_ptr<Texture>> texture_owners;
    rthip_scene_desc desc{};

    explicit SceneDesc(const Scene &scene, std::vector<rthip_light> lights_ = {}) : lights(std::move(lights_)) {
        std::unordered_map<const Material *, uint32_t> mat_index;
        std::unordered_map<const Texture *, uint32_t> tex_index;
        auto mat_id = [&](const std::shared_ptr<Material> &m) -> uint32_t {
            if (!m) throw std::invalid_argument("shape without material");
            auto it = mat_index.find(m.get());
            if (it != mat_index.end()) return it->second;
            uint32_t ti = 0;
            if (auto t = std::dynamic_pointer_cast<Texture>(m)) {
                auto jt = tex_index.find(t.get());
                if (jt == tex_index.end()) {
                    ti = uint32_t(textures.size());
                    tex_index[t.get()] = ti;
                    textures.push_back(rthip_texture{t->width, t->height, t->rgb8.data()});
                    texture_owners.push_back(t);
                } else {
                    ti = jt->second;
                }
            }
            uint32_t id = uint32_t(materials.size());
            materials.push_back(m->record(ti));
            mat_index[m.get()] = id;
            return id;
        };
        for (const auto &sh : scene) {
            if (auto s = std::dynamic_pointer_cast<Sphere>(sh))
                spheres.push_back(rthip_sphere{{s->center[0], s->center[1], s->center[2]}, s->radius, mat_id(s->material), 0});
            else if (auto p = std::dynamic_pointer_cast<Plane>(sh))
                planes.push_back(rthip_plane{{p->point[0], p->point[1], p->point[2]}, {p->normal[0], p->normal[1], p->normal[2]}, mat_id(p->material), 0});
            else if (auto t = std::dynamic_pointer_cast<Triangle>(sh))
                triangles.push_back(rthip_triangle{{t->v0[0], t->v0[1], t->v0[2]}, {t->v1[0], t->v1[1], t->v1[2]}, {t->v2[0], t->v2[1], t->v2[2]}, mat_id(t->material), 0});
            else
                throw std::invalid_argument("unknown shape type");
        }
        desc.spheres = spheres.data(), desc.n_spheres = spheres.size();
        desc.planes = planes.data(), desc.n_planes = planes.size();
        desc.triangles = triangles.data(), desc.n_triangles = triangles.size();
        desc.materials = materials.data(), desc.n_materials = materials.size();
        desc.textures = textures.data(), desc.n_textures = textures.size();
        desc.lights = lights.data(), desc.n_lights = lights.size();
    }
};

// ---- camera -----------------------------------------------------------------------------
struct Ray {
    Vector3 origin, direction;
};
class Camera {
public:
    Vector3 origin;  // pub origin — camera.rs:8
    static Camera create(double x, double y, double z) { return new_lookat(x, y, z, {0.0, 0.0, -1.0}); }  // camera.rs:16-18
    static Camera new_lookat(double x, double y, double z, Vector3 lookat) {
        Camera c;
        if (int st = rthip_camera_lookat(x, y, z, lookat.data(), &c.pod_)) throw HipError(st, "rthip_camera_lookat");
        c.origin = {x, y, z};
        return c;
    }
    Ray emit_ray_at(double ox, double oy) const {  // camera.rs:43-50
        Ray r{origin, {}};
        for (int k = 0; k < 3; k++) r.direction[k] = pod_.d00[k] + pod_.horizontal[k] * ox + pod_.vertical[k] * oy;
        return r;
    }
    const rthip_camera &pod() const { return pod_; }

private:
    rthip_camera pod_{};
};

// ---- driver -----------------------------------------------------------------------------
enum class GenerationStatus { NotStarted, CopyNearPixel, Unstable, Final };  // lib.rs:58-64
struct PixelColor {                                                           // lib.rs:17-22
    uint8_t r, g, b;
    GenerationStatus status;
};
struct PixelPosition {  // lib.rs:41-45
    size_t x, y;
};
struct PixelRenderer {  // lib.rs:77-80
    virtual ~PixelRenderer() = default;
    virtual void set_pixel(PixelPosition pos, PixelColor color) = 0;
    virtual void invalidate_pixels() = 0;
};

class HipRaytracer {
public:
    Camera camera = Camera::create(-1.8, 1.0, 2.0);  // lib.rs:252
    HipRaytracer(double width, double height, uint32_t seed, int device = 0) : width_(width), height_(height), seed_(seed) {
        if (int st = rthip_create(device, &ctx_)) throw HipError(st, rthip_last_error(nullptr));
    }
    HipRaytracer(const HipRaytracer &) = delete;
    HipRaytracer &operator=(const HipRaytracer &) = delete;
    ~HipRaytracer() { rthip_destroy(ctx_); }

    void upload(const SceneDesc &scene) { check(rthip_upload_scene(ctx_, &scene.desc)); }

    // The batch form of generate_pixel (lib.rs:269-316; driver loop of wasm/src/lib.rs:112-123): every pixel once with
    // `samples`; set_pixel is called for every pixel, y-up, status Final. accumulate = keep adding to the running sum.
    rthip_stats render(uint64_t samples, uint32_t max_depth, PixelRenderer *renderer, bool accumulate = false, uint32_t flags = 0,
                       std::vector<uint8_t> *rgba8_out = nullptr) {
        check(rthip_set_camera_if_changed());
        rthip_render_params p{};
        p.width = uint32_t(width_), p.height = uint32_t(height_), p.spp = uint32_t(samples), p.max_depth = max_depth, p.seed = seed_;
        p.flags = flags | (accumulate ? RTHIP_F_ACCUMULATE : 0);
        std::vector<uint8_t> local;
        std::vector<uint8_t> &img = rgba8_out ? *rgba8_out : local;
        img.resize(size_t(p.width) * p.height * 4);
        rthip_stats st{};
        check(rthip_render(ctx_, &p, img.data(), nullptr, &st));
        if (renderer)
            for (size_t y = 0; y < p.height; y++)
                for (size_t x = 0; x < p.width; x++) {
                    const uint8_t *px = &img[(y * p.width + x) * 4];
                    renderer->set_pixel({x, y}, {px[0], px[1], px[2], GenerationStatus::Final});
                }
        return st;
    }
    // One progressive pass with the reference's convergence rule (lib.rs:276-313, RTHIP_F_ADAPTIVE): `first` starts a new
    // running sum. As in generate_pixel, set_pixel is called only for the pixels that were not Final before the pass, with
    // their status after it (Unstable, or Final on the pass that froze them).
    rthip_stats render_adaptive_pass(uint64_t samples, uint32_t max_depth, PixelRenderer *renderer, bool first,
                                     std::vector<uint8_t> *rgba8_out = nullptr) {
        const size_t npx = size_t(width_) * size_t(height_);
        if (first || status_.size() != npx) status_.assign(npx, uint8_t(RTHIP_STATUS_NOT_STARTED));
        const std::vector<uint8_t> before = status_;
        std::vector<uint8_t> local;
        std::vector<uint8_t> &img = rgba8_out ? *rgba8_out : local;
        rthip_stats st = render(samples, max_depth, nullptr, !first, RTHIP_F_ADAPTIVE, &img);
        check(rthip_pixel_status(ctx_, status_.data(), nullptr, status_.size()));
        if (renderer)
            for (size_t y = 0; y < size_t(height_); y++)
                for (size_t x = 0; x < size_t(width_); x++) {
                    const size_t i = y * size_t(width_) + x;
                    if (before[i] == RTHIP_STATUS_FINAL) continue;  // lib.rs:277-279
                    const uint8_t *px = &img[i * 4];
                    renderer->set_pixel({x, y}, {px[0], px[1], px[2], status_[i] == RTHIP_STATUS_FINAL ? GenerationStatus::Final : GenerationStatus::Unstable});
                }
        return st;
    }
    const std::vector<uint8_t> &pixel_status() const { return status_; }
    void invalidate_pixels() {
        check(rthip_reset_accum(ctx_));
        status_.clear();
    }

    // Raytracer::get_shape — lib.rs:262-267; -1 = None
    int64_t get_shape(double x, double y) {
        check(rthip_set_camera_if_changed());
        int64_t s = -1;
        check(rthip_pick(ctx_, uint32_t(width_), uint32_t(height_), x, y, &s));
        return s;
    }

private:
    int rthip_set_camera_if_changed() {
        if (have_cam_ && std::memcmp(&uploaded_, &camera.pod(), sizeof uploaded_) == 0) return 0;
        uploaded_ = camera.pod();
        have_cam_ = true;
        return rthip_set_camera(ctx_, &uploaded_);
    }
    void check(int st) {
        if (st) throw HipError(st, rthip_last_error(ctx_));
    }
    rthip_ctx *ctx_ = nullptr;
    double width_, height_;
    uint32_t seed_;
    std::vector<uint8_t> status_;  // RTHIP_STATUS_* per pixel after the last adaptive pass
    rthip_camera uploaded_{};
    bool have_cam_ = false;
};

// One caller, several GPUs (rthip_group_*): the whole frame from one call, rendered as interleaved row bands on every listed
// device and gathered to the first over RCCL. The same batch `render` as HipRaytracer's.
class HipGroupRaytracer {
public:
    Camera camera = Camera::create(-1.8, 1.0, 2.0);  // lib.rs:252
    HipGroupRaytracer(double width, double height, uint32_t seed, const std::vector<int> &devices, uint32_t group_flags = 0)
        : width_(width), height_(height), seed_(seed) {
        if (int st = rthip_group_create(devices.data(), int(devices.size()), group_flags, &grp_)) throw HipError(st, rthip_group_last_error(nullptr));
    }
    HipGroupRaytracer(const HipGroupRaytracer &) = delete;
    HipGroupRaytracer &operator=(const HipGroupRaytracer &) = delete;
    ~HipGroupRaytracer() { rthip_group_destroy(grp_); }
    void upload(const SceneDesc &scene) { check(rthip_group_upload_scene(grp_, &scene.desc)); }
    rthip_stats render(uint64_t samples, uint32_t max_depth, PixelRenderer *renderer, bool accumulate = false, uint32_t flags = 0,
                       std::vector<uint8_t> *rgba8_out = nullptr, uint32_t band_rows = 8) {
        if (!have_cam_ || std::memcmp(&uploaded_, &camera.pod(), sizeof uploaded_) != 0) {
            uploaded_ = camera.pod();
            have_cam_ = true;
            check(rthip_group_set_camera(grp_, &uploaded_));
        }
        rthip_render_params p{};
        p.width = uint32_t(width_), p.height = uint32_t(height_), p.spp = uint32_t(samples), p.max_depth = max_depth, p.seed = seed_;
        p.flags = flags | (accumulate ? RTHIP_F_ACCUMULATE : 0);
        p.band_rows = band_rows;
        std::vector<uint8_t> local;
        std::vector<uint8_t> &img = rgba8_out ? *rgba8_out : local;
        img.resize(size_t(p.width) * p.height * 4);
        rthip_stats st{};
        check(rthip_group_render(grp_, &p, img.data(), nullptr, &st));
        if (renderer)
            for (size_t y = 0; y < p.height; y++)
                for (size_t x = 0; x < p.width; x++) {
                    const uint8_t *px = &img[(y * p.width + x) * 4];
                    renderer->set_pixel({x, y}, {px[0], px[1], px[2], GenerationStatus::Final});
                }
        return st;
    }
    void invalidate_pixels() { check(rthip_group_reset_accum(grp_)); }

private:
    void check(int st) {
        if (st) throw HipError(st, rthip_group_last_error(grp_));
    }
    rthip_group *grp_ = nullptr;
    double width_, height_;
    uint32_t seed_;
    rthip_camera uploaded_{};
    bool have_cam_ = false;
};

// ---- image writers ----------------------------------------------------------
// PPM P3 in the layout of the reference's (dead) standalone/src/renderers/ppm.rs:23-35: rows from the TOP of the
// picture down, i.e. y from height-1 to 0 of the y-up buffer.
inline void write_ppm_p3(const std::string &path, const std::vector<uint8_t> &rgba8, uint32_t width, uint32_t height) {
    std::ofstream f(path);
    f << "P3\n" << width << " " << height << " \n255\n";
    for (uint32_t y = height; y-- > 0;)
        for (uint32_t x = 0; x < width; x++) {
            const uint8_t *p = &rgba8[(size_t(y) * width + x) * 4];
            f << int(p[0]) << " " << int(p[1]) << " " << int(p[2]) << "\n";
        }
}
// top-down RGBA8 frame as the pixels/wgpu front-end wants it (standalone/src/renderers/pixels.rs:240-260)
inline std::vector<uint8_t> flip_to_top_down(const std::vector<uint8_t> &rgba8, uint32_t width, uint32_t height) {
    std::vector<uint8_t> out(rgba8.size());
    for (uint32_t y = 0; y < height; y++) std::memcpy(&out[size_t(height - 1 - y) * width * 4], &rgba8[size_t(y) * width * 4], size_t(width) * 4);
    return out;
}

}  // namespace raytracer_hip
