// rthip_cli.cpp — C++ front-end on the drop-in surface (raytracer_hip.hpp over include/rthip.h): builds the
// reference's standalone scene (standalone/src/main.rs:66-89) with the mirrored constructors, renders it
// on the GPU through a PixelRenderer (the `World` framebuffer of standalone/src/renderers/pixels.rs:205-262,
// reduced to a byte buffer), writes a PPM and prints one JSON line of statistics.
//   rthip_cli [--scene standalone|wasm|grid] [--grid N] [--width W] [--height H] [--spp N] [--depth D] [--seed S] [--texture path.rgb8.z]
//             [--out image.ppm] [--rgba-out frame.rgba]   frame.rgba: the raw top-down RGBA8 frame as the pixels/wgpu front-end
//                            takes it (standalone/src/renderers/pixels.rs:240-260: screen row 0 = pos.y = height-1)
//             [--devices 0,1,..]  render through the single-process multi-GPU group (rthip_group_*: interleaved row bands, RCCL
//                            gather); [--peer-copy 1] gathers with peer copies instead (allows a device to be listed twice)
//             --scene wasm: the three metal spheres of wasm/src/lib.rs:90-105; grid: N x N small spheres + ground + a quad (extension
//             shapes; big enough N goes through the BVH) — tests/test_cli_gpu.py builds the same scenes through the Python layer
//             [--passes P]   P > 0: the standalone window's progressive loop (standalone/src/main.rs:99-102) — P adaptive
//                            passes of N samples with the convergence rule of lib.rs:276-313 instead of one batch
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "raytracer_hip.hpp"

using namespace raytracer_hip;

static Scene standalone_scene(const std::string &texture) {  // standalone/src/main.rs:66-89
    return {
        Sphere::create({-1.01, 0.0, -1.0}, 0.5, Dielectric::create({1.0, 0.8, 0.80}, 1.05)),
        Sphere::create({0.0, -100.5, -1.0}, 100.0, Lambertian::new_from_hex(0x007070)),
        Sphere::create({1.0, 0.0, -1.0}, 0.5, Metal::create({0.8, 0.8, 0.8}, 0.1)),
        Sphere::create({-0.0, 0.0, -1.0}, 0.5, Texture::load_from_file(texture, 1.0)),
    };
}

static Scene wasm_scene() {  // wasm/src/lib.rs:90-105
    return {
        Sphere::create({0.0, 0.0, -1.0}, 0.5, Metal::create({0.8, 0.8, 0.8}, 0.0)),
        Sphere::create({0.0, -100.5, -1.0}, 100.0, Metal::create({0.8, 0.8, 0.8}, 0.0)),
        Sphere::create({0.5, -0.4, -0.85}, 0.1, Metal::create({0.8, 0.8, 0.8}, 0.0)),
    };
}

// n x n small spheres of the three scattering materials over the ground sphere, and a quad of two triangles behind them
static Scene grid_scene(int n) {
    Scene s;
    s.push_back(Sphere::create({0.0, -100.5, -1.0}, 100.0, Lambertian::new_from_hex(0x007070)));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const Vector3 c = {(i - n * 0.5) * 0.6, -0.3 + 0.05 * ((i * 7 + j * 3) % 5), -1.0 - j * 0.6};
            const double radius = 0.1 + 0.02 * ((i + 2 * j) % 4);
            std::shared_ptr<Material> m;
            switch ((i + j) % 3) {
                case 0: m = Lambertian::create({0.2 + 0.1 * (i % 5), 0.5, 0.3 + 0.1 * (j % 4)}); break;
                case 1: m = Metal::create({0.8, 0.8, 0.8}, 0.1 * ((i + j) % 4)); break;
                default: m = Dielectric::create({1.0, 1.0, 1.0}, 1.5);
            }
            s.push_back(Sphere::create(c, radius, m));
        }
    auto wall = Lambertian::create({0.7, 0.6, 0.2});
    const double z = -1.5 - n * 0.6;
    s.push_back(Triangle::create({-4.0, -0.5, z}, {4.0, -0.5, z}, {4.0, 2.0, z}, wall));
    s.push_back(Triangle::create({-4.0, -0.5, z}, {4.0, 2.0, z}, {-4.0, 2.0, z}, wall));
    return s;
}

struct World : PixelRenderer {  // standalone/src/renderers/pixels.rs:205-262
    uint32_t width, height;
    std::vector<uint8_t> pixels;  // y-up RGBA8
    uint64_t writes = 0, final_writes = 0;
    World(uint32_t w, uint32_t h) : width(w), height(h), pixels(size_t(w) * h * 4, 0) {}
    void set_pixel(PixelPosition pos, PixelColor c) override {
        uint8_t *p = &pixels[(pos.y * width + pos.x) * 4];
        p[0] = c.r, p[1] = c.g, p[2] = c.b, p[3] = 255;
        writes++;
        if (c.status == GenerationStatus::Final) final_writes++;
    }
    void invalidate_pixels() override { std::fill(pixels.begin(), pixels.end(), 0); }
};

int main(int argc, char **argv) {
    uint32_t width = 960, height = 540, spp = 16, depth = 50, seed = 1, passes = 0;  // standalone/src/main.rs:51-52, ray.rs:44
    std::string out, rgba_out, texture = "tests/golden/textures/bergsjostolen.rgb8.z", scene_name = "standalone";
    std::vector<int> devices;
    int grid = 12, peer_copy = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--width") width = std::atoi(argv[i + 1]);
        else if (k == "--height") height = std::atoi(argv[i + 1]);
        else if (k == "--spp") spp = std::atoi(argv[i + 1]);
        else if (k == "--depth") depth = std::atoi(argv[i + 1]);
        else if (k == "--seed") seed = std::atoi(argv[i + 1]);
        else if (k == "--passes") passes = std::atoi(argv[i + 1]);
        else if (k == "--texture") texture = argv[i + 1];
        else if (k == "--scene") scene_name = argv[i + 1];
        else if (k == "--grid") grid = std::atoi(argv[i + 1]);
        else if (k == "--out") out = argv[i + 1];
        else if (k == "--rgba-out") rgba_out = argv[i + 1];
        else if (k == "--peer-copy") peer_copy = std::atoi(argv[i + 1]);
        else if (k == "--devices") {
            for (const char *p = argv[i + 1]; *p;) {
                devices.push_back(std::atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    try {
        Scene scene;
        if (scene_name == "standalone") scene = standalone_scene(texture);
        else if (scene_name == "wasm") scene = wasm_scene();
        else if (scene_name == "grid" && grid > 0 && grid <= 1000) scene = grid_scene(grid);
        else { std::fprintf(stderr, "unknown scene %s\n", scene_name.c_str()); return 2; }
        SceneDesc desc(scene);
        HipRaytracer rt(width, height, seed, devices.empty() ? 0 : devices[0]);
        rt.upload(desc);
        World world(width, height);
        rthip_stats st{};
        uint64_t paths = 0, rays = 0, final_pixels = 0;
        double kernel_ms = 0.0;
        if (!devices.empty()) {
            if (passes) { std::fprintf(stderr, "--devices renders one batch (no --passes)\n"); return 2; }
            HipGroupRaytracer grp(width, height, seed, devices, peer_copy ? RTHIP_GROUP_F_PEER_COPY : 0);
            grp.upload(desc);
            st = grp.render(spp, depth, &world);
            paths = st.paths, rays = st.rays, kernel_ms = st.kernel_ms;
        } else if (passes == 0) {
            st = rt.render(spp, depth, &world);
            paths = st.paths, rays = st.rays, kernel_ms = st.kernel_ms;
        } else {
            for (uint32_t k = 0; k < passes; k++) {
                st = rt.render_adaptive_pass(spp, depth, &world, k == 0);
                paths += st.paths, rays += st.rays, kernel_ms += st.kernel_ms;
            }
            final_pixels = st.pixels_final;
        }
        int64_t centre = rt.get_shape((width - 1) * 0.5, (height - 1) * 0.5);
        if (!out.empty()) write_ppm_p3(out, world.pixels, width, height);
        if (!rgba_out.empty()) {
            const std::vector<uint8_t> frame = flip_to_top_down(world.pixels, width, height);
            std::ofstream f(rgba_out, std::ios::binary);
            f.write(reinterpret_cast<const char *>(frame.data()), std::streamsize(frame.size()));
        }
        std::printf("{\"width\": %u, \"height\": %u, \"spp\": %u, \"max_depth\": %u, \"paths\": %llu, \"rays\": %llu, \"kernel_ms\": %.3f, "
                    "\"d2h_ms\": %.3f, \"mrays_per_s\": %.1f, \"set_pixel_calls\": %llu, \"shape_at_centre\": %lld, \"passes\": %u, "
                    "\"final_pixels\": %llu, \"final_set_pixel_calls\": %llu, \"devices\": %d, \"gather_ms\": %.3f}\n",
                    width, height, spp, depth, (unsigned long long)paths, (unsigned long long)rays, kernel_ms, st.d2h_ms,
                    rays / kernel_ms / 1e3, (unsigned long long)world.writes, (long long)centre, passes, (unsigned long long)final_pixels,
                    (unsigned long long)world.final_writes, int(devices.size()), st.gather_ms);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rthip_cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
