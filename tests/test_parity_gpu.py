"""GPU parity tests proper: the HIP path, called through the C ABI (librthip.so), against
 (a) the committed golden images (f64 oracle renders, tests/golden/images/*.npz),
 (b) the live CPU oracle on the same seeded inputs at sizes it finishes in seconds,
 (c) size-independent properties at BASELINE.json's full sizes (4K): GPU-count invariance of the
     image bytes, determinism, the accumulation contract, path/ray count identities.
Bar (BASELINE.json north_star): per-channel |delta| <= 1/255 on >= 99.9 % of pixels; the mismatch
histogram is printed for every comparison. Device arithmetic is FP32, the gate oracle is f64
(reference-faithful); the f32 oracle mode is reported alongside."""
import ctypes as C
import glob
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
from parity import assert_parity, describe, mismatch_histogram  # noqa: E402

from rustracer_amd import Lambertian, Raytracer, Sphere, abi, flatten, make_params, presets  # noqa: E402
from rustracer_amd.bands import owned_row_indices  # noqa: E402

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images")


def _golden_cases():
    from make_golden_images import CASES
    return CASES


@pytest.mark.parametrize("name", sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))))
def test_golden_images(name):
    scene_f, lights_f, w, h, spp, depth, seed, flags, pose = _golden_cases()[name]
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    rt = Raytracer.new(w, h, seed=seed)
    rt.camera = type(rt.camera).new(*pose)
    res = rt.render(scene_f(), spp, max_depth=depth, flags=flags, lights=lights_f() if lights_f else ())
    hist, ok = assert_parity(res.rgba8, gold["rgba8"], name)
    print(f"\n{name}: {describe(hist, ok)}; rays device={res.stats['rays']} golden={int(gold['rays'])}")
    assert res.stats["paths"] == int(gold["paths"])
    assert abs(res.stats["rays"] - int(gold["rays"])) <= 2e-3 * int(gold["rays"])
    # radiance sums agree to FP32 accumulation accuracy on the bulk of the pixels
    d = np.abs(res.accum - gold["accum"])
    assert np.median(d) <= 1e-4 * spp
    rt.close()


@pytest.mark.parametrize("preset,w,h,spp,depth,seed", [
    ("standalone_scene", 480, 270, 1, 50, 11),
    ("standalone_scene", 480, 270, 16, 50, 12),
    ("standalone_scene", 480, 270, 16, 5, 13),
    ("standalone_scene", 400, 300, 8, 50, 14),   # C1's 4:3 aspect: the viewport stays 16:9 (camera.rs:5,25)
    ("wasm_scene", 384, 216, 64, 50, 0),         # the wasm batch render (wasm/src/lib.rs:15-18)
    ("ggez_scene", 400, 300, 4, 50, 15),
    ("textured_scene", 480, 270, 4, 50, 16),
    ("plane_scene", 320, 180, 8, 8, 17),
])
def test_parity_vs_live_oracle(oracle_mod, default_camera, preset, w, h, spp, depth, seed):
    flat = flatten(getattr(presets, preset)())
    rt = Raytracer.new(w, h, seed=seed)
    res = rt.render(flat, spp, max_depth=depth, flags=abi.F_COUNT_STATS)
    p = make_params(w, h, spp, depth, seed)
    ref64, _, st64, _ = oracle_mod.render(flat, default_camera, p, precision=64)
    ref32, _, _, _ = oracle_mod.render(flat, default_camera, p, precision=32)
    hist, ok = assert_parity(res.rgba8, ref64, f"{preset} vs oracle f64")
    h32, ok32 = mismatch_histogram(res.rgba8, ref32)
    print(f"\n{preset} {w}x{h}@{spp} d{depth}: vs f64 {describe(hist, ok)} | vs f32 within 1/255: {ok32 * 100:.3f} %")
    # the instrumented counters agree with the oracle's (same paths; rays within 0.2 %: rare FP32 path divergences)
    assert res.stats["paths"] == st64["paths"] == w * h * spp
    assert abs(res.stats["rays"] - st64["rays"]) <= 2e-3 * st64["rays"]
    assert res.stats["sphere_tests"] == res.stats["rays"] * len(flat.spheres)
    assert res.stats["tri_tests"] == res.stats["rays"] * len(flat.triangles)
    assert sum(res.stats["hits"]) + res.stats["sky_evals"] >= res.stats["rays"]
    for k in range(4):
        assert abs(res.stats["hits"][k] - st64["hits"][k]) <= 3e-3 * max(1000, st64["hits"][k])
    rt.close()


def test_edge_sizes_and_depths(oracle_mod, default_camera, standalone_flat):
    # ragged sizes (not multiples of the 16x16 tile), the minimum image, depth 0 / 1
    for (w, h, spp, depth) in [(2, 2, 3, 50), (17, 5, 2, 50), (33, 31, 1, 1), (64, 36, 2, 0), (1000, 3, 1, 50)]:
        rt = Raytracer.new(w, h, seed=5)
        res = rt.render(standalone_flat, spp, max_depth=depth)
        ref, _, st, _ = oracle_mod.render(standalone_flat, default_camera, make_params(w, h, spp, depth, 5), precision=64)
        hist, ok = mismatch_histogram(res.rgba8, ref)
        assert (hist[2:].sum() <= max(1, 0.002 * w * h)), (w, h, spp, depth, describe(hist, ok))
        assert res.stats["paths"] == w * h * spp
        if depth == 0:
            assert res.stats["rays"] == 0  # ray.rs:70-72: sky without a scene query
            assert hist[2:].sum() == 0
        if depth == 1:
            assert res.stats["rays"] == res.stats["paths"]
        rt.close()


def test_empty_scene_is_all_sky(oracle_mod, default_camera):
    flat = flatten([])
    rt = Raytracer.new(64, 36, seed=1)
    res = rt.render(flat, 2, max_depth=50)
    ref, _, _, _ = oracle_mod.render(flat, default_camera, make_params(64, 36, 2, 50, 1), precision=64)
    hist, ok = mismatch_histogram(res.rgba8, ref)
    assert hist[2:].sum() == 0
    assert res.stats["rays"] == res.stats["paths"]  # one (missing) query per path
    assert rt.get_shape(flat, 10, 10) is None
    rt.close()


def test_tie_break_later_index_wins():
    # two identical spheres: the LATER index wins (ray.rs:55-61), through pick and through shading
    red, green = Lambertian.new((1.0, 0.0, 0.0)), Lambertian.new((0.0, 1.0, 0.0))
    scene = [Sphere.new((0, 0, -1), 0.5, red), Sphere.new((0, 0, -1), 0.5, green)]
    rt = Raytracer.new(64, 36, seed=1)
    assert rt.get_shape(scene, 31.5, 17.5) == 1
    res = rt.render(scene, 4, max_depth=2)
    centre = res.accum[18, 32]
    assert centre[0] == 0.0 and centre[1] > 0.0
    rt.close()


def test_pick_matches_oracle(oracle_mod, default_camera, standalone_flat):
    w, h = 960, 540
    rt = Raytracer.new(w, h, seed=0)
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, y = rng.uniform(0, w - 1), rng.uniform(0, h - 1)
        o, d = oracle_mod.emit_ray_at(default_camera, x / (w - 1.0), y / (h - 1.0))  # lib.rs:263-265
        want = oracle_mod.find_collision(standalone_flat, o, d)[0]
        assert rt.get_shape(standalone_flat, x, y) == want, (x, y)
    rt.close()


def test_determinism_and_seed_sensitivity(standalone_flat):
    rt = Raytracer.new(320, 180, seed=21)
    a = rt.render(standalone_flat, 4)
    b = rt.render(standalone_flat, 4)
    assert np.array_equal(a.rgba8, b.rgba8) and np.array_equal(a.accum, b.accum)
    rt2 = Raytracer.new(320, 180, seed=22)
    c = rt2.render(standalone_flat, 4)
    assert not np.array_equal(a.rgba8, c.rgba8)
    rt.close(), rt2.close()


def test_accumulation_contract(oracle_mod, default_camera, standalone_flat):
    # lib.rs:289-294: passes add to the running sum; two passes of 4 == the sample stream of one pass of 8
    w, h = 320, 180
    rt = Raytracer.new(w, h, seed=31)
    one = rt.render(standalone_flat, 8)
    rt.invalidate_pixels()
    rt.render(standalone_flat, 4)
    two = rt.render(standalone_flat, 4, accumulate=True)
    assert two.stats["nb_samples"] == 8
    assert np.allclose(two.accum, one.accum, rtol=2e-6, atol=2e-6)
    assert np.abs(two.rgba8.astype(int) - one.rgba8.astype(int)).max() <= 1
    # and it agrees with the oracle driven the same way
    p4 = make_params(w, h, 4, 50, 31)
    _, s1, _, _ = oracle_mod.render(standalone_flat, default_camera, p4)
    p4b = make_params(w, h, 4, 50, 31, flags=abi.F_ACCUMULATE)
    ref, _, st, _ = oracle_mod.render(standalone_flat, default_camera, p4b, prior_samples=4, sum_in=s1)
    assert_parity(two.rgba8, ref, "accumulated pass vs oracle")
    # a camera change invalidates the running sum (standalone/src/main.rs:103-120)
    rt.camera = rt.camera.move_camera((0.0, 0.0, 0.5))
    moved = rt.render(standalone_flat, 4, accumulate=True)
    assert moved.stats["nb_samples"] == 4
    rt.close()


@pytest.mark.parametrize("h,band_rows,world", [(180, 8, 2), (180, 8, 8), (181, 16, 3), (37, 8, 4), (100, 7, 8)])
def test_band_split_is_byte_identical(standalone_flat, h, band_rows, world):
    # RNG keyed by the absolute pixel index => the assembled image is identical for any GPU count
    w = 320
    rt = Raytracer.new(w, h, seed=41)
    whole = rt.render(standalone_flat, 3)
    full_rgba = np.zeros_like(whole.rgba8)
    full_acc = np.zeros_like(whole.accum)
    rays = 0
    for rank in range(world):
        part = rt.render(standalone_flat, 3, band_rows=band_rows, band_first=rank, band_stride=world)
        idx = owned_row_indices(h, band_rows, rank, world)
        assert part.rgba8.shape[0] == len(idx)
        full_rgba[idx] = part.rgba8
        full_acc[idx] = part.accum
        rays += part.stats["rays"]
    assert np.array_equal(full_rgba, whole.rgba8)
    assert np.array_equal(full_acc, whole.accum)
    assert rays == whole.stats["rays"]
    rt.close()


def test_point_light_mode_adds_light(oracle_mod, default_camera):
    # extension: with the flag off the image is the reference image; with it on, radiance only increases
    flat = flatten(presets.standalone_scene(), presets.standalone_lights())
    rt = Raytracer.new(320, 180, seed=51)
    off = rt.render(flat, 4, max_depth=5)
    on = rt.render(flat, 4, max_depth=5, flags=abi.F_POINT_LIGHTS | abi.F_COUNT_STATS)
    assert (on.accum >= off.accum - 1e-5).all() and (on.accum > off.accum + 1e-3).any()
    assert on.stats["shadow_rays"] > 0 and on.stats["rays"] == off.stats["rays"] + on.stats["shadow_rays"]
    p = make_params(320, 180, 4, 5, 51, abi.F_POINT_LIGHTS)
    ref, _, st, _ = oracle_mod.render(flat, default_camera, p, precision=64)
    assert_parity(on.rgba8, ref, "point-light mode vs oracle")
    assert abs(on.stats["shadow_rays"] - st["shadow_rays"]) <= 3e-3 * st["shadow_rays"]
    rt.close()


def test_bvh_equals_brute_force(oracle_mod, default_camera):
    # extension: the linear BVH must return exactly what the linear scan of ray.rs:47-67 returns
    flat = flatten(presets.big_scene(n_side=20, seed=3, mesh_subdivisions=2))  # 401 spheres + 320 triangles
    w, h = 320, 180
    rt = Raytracer.new(w, h, seed=61)
    bvh = rt.render(flat, 2, max_depth=8, flags=abi.F_COUNT_STATS)
    brute = rt.render(flat, 2, max_depth=8, flags=abi.F_NO_BVH | abi.F_COUNT_STATS)
    assert brute.stats["node_tests"] == 0 and bvh.stats["node_tests"] > 0
    assert bvh.stats["sphere_tests"] < brute.stats["sphere_tests"] / 4
    assert np.array_equal(bvh.rgba8, brute.rgba8) and np.array_equal(bvh.accum, brute.accum)
    assert bvh.stats["rays"] == brute.stats["rays"]
    # vs the oracle (brute force, f64), at two and at eight bounces: the strict bar at both. (Rounds 1-2 measured 99.78 % at
    # depth 8 and called it chaotic transport. It was the sphere discriminant: half_b^2 - (|oc|^2 - r^2) cancels in FP32 for a
    # small sphere seen from afar, silhouettes moved by a tenth of a pixel; sphere_roots now takes r^2 - |oc - (oc.d) d|^2.
    # The oracle's f32 mode keeps the reference's formula in float and still shows the loss.)
    shallow = rt.render(flat, 2, max_depth=2)
    ref2, _, _, _ = oracle_mod.render(flat, default_camera, make_params(w, h, 2, 2, 61), precision=64)
    hist2, ok2 = assert_parity(shallow.rgba8, ref2, "BVH scene, depth 2, vs oracle f64")
    ref, _, st, _ = oracle_mod.render(flat, default_camera, make_params(w, h, 2, 8, 61), precision=64)
    ref32, _, _, _ = oracle_mod.render(flat, default_camera, make_params(w, h, 2, 8, 61), precision=32)
    hist, ok = mismatch_histogram(bvh.rgba8, ref)
    _, ok_f32cpu = mismatch_histogram(ref32, ref)
    print(f"\nbvh scene depth 2: {describe(hist2, ok2)}\nbvh scene depth 8: {describe(hist, ok)} (f32-on-CPU vs f64: {ok_f32cpu * 100:.3f} %)")
    assert ok >= 0.999, describe(hist, ok)  # measured on MI355X, round 3: 99.96 % (the reference's formula in float on the CPU: 99.64 %)
    assert abs(bvh.stats["rays"] - st["rays"]) <= 5e-3 * st["rays"]
    # picking goes through the same traversal
    rng = np.random.default_rng(1)
    for _ in range(100):
        x, y = rng.uniform(0, w - 1), rng.uniform(0, h - 1)
        o, d = oracle_mod.emit_ray_at(default_camera, x / (w - 1.0), y / (h - 1.0))
        assert rt.get_shape(flat, x, y) == oracle_mod.find_collision(flat, o, d)[0]
    rt.close()


def test_bvh_stack_split_does_not_change_the_image(monkeypatch):
    # the traversal stack keeps its first levels in LDS and deeper ones in private memory; where the split lies is a
    # tuning knob (RTHIP_STACK_LDS) and must not show: 1 level in LDS (nearly every push goes to private memory),
    # the default (8) and all levels in LDS give the same bits, with and without point lights, and all equal the linear scan
    from rustracer_amd import PointLight
    flat = flatten(presets.big_scene(n_side=24, seed=9, mesh_subdivisions=3),  # 577 spheres + 1280 triangles: tree deeper than 8
                   [PointLight((4.0, 5.0, 2.0), (30.0, 30.0, 30.0))])
    rt = Raytracer.new(256, 144, seed=5)
    for flags in (0, abi.F_POINT_LIGHTS):
        brute = rt.render(flat, 3, max_depth=6, flags=flags | abi.F_NO_BVH)
        images = []
        for levels in ("1", "3", None, "32"):
            if levels is None:
                monkeypatch.delenv("RTHIP_STACK_LDS", raising=False)
            else:
                monkeypatch.setenv("RTHIP_STACK_LDS", levels)
            r = rt.render(flat, 3, max_depth=6, flags=flags | abi.F_COUNT_STATS)
            images.append(r)
            assert np.array_equal(r.accum, brute.accum) and np.array_equal(r.rgba8, brute.rgba8), levels
        assert len({r.stats["node_tests"] for r in images}) == 1
    rt.close()


@pytest.mark.parametrize("spp", [1, 3, 5, 17, 67])
def test_bvh_sample_ring_any_sample_count(spp):
    # the BVH kernel hands (pixel, sample) items to lanes and retires sample rows through a ring of 4 rows: row counts
    # below, at and off multiples of the ring (and of the 64-sample groups), on an image whose tiles are partly outside
    flat = flatten(presets.mesh_scene(4, 2))
    rt = Raytracer.new(100, 52, seed=13)  # 100 x 52: the right and top tiles are cut
    a = rt.render(flat, spp, max_depth=6, flags=abi.F_COUNT_STATS)
    b = rt.render(flat, spp, max_depth=6, flags=abi.F_COUNT_STATS | abi.F_NO_BVH)
    assert a.stats["node_tests"] > 0 and b.stats["node_tests"] == 0
    assert a.stats["paths"] == b.stats["paths"] == 100 * 52 * spp and a.stats["rays"] == b.stats["rays"]
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgba8, b.rgba8)
    rt.close()


def test_bvh_tie_break_duplicates():
    # 200 coincident sphere pairs: equal t everywhere; the larger index must win inside the BVH too
    rng = np.random.default_rng(2)
    centers = rng.uniform(-3, 3, (200, 3)) + np.array([0, 0, -6.0])
    red, green = Lambertian.new((1.0, 0.0, 0.0)), Lambertian.new((0.0, 1.0, 0.0))
    scene = [Sphere.new(c, 0.3, red) for c in centers] + [Sphere.new(c, 0.3, green) for c in centers]
    rt = Raytracer.new(160, 90, seed=1)
    a = rt.render(scene, 2, max_depth=1)
    b = rt.render(scene, 2, max_depth=1, flags=abi.F_NO_BVH)
    assert np.array_equal(a.accum, b.accum)
    for _ in range(50):
        x, y = rng.uniform(0, 159), rng.uniform(0, 89)
        s = rt.get_shape(scene, x, y)
        assert s is None or s >= 200
    rt.close()


def test_error_behaviour():
    lib = abi.lib()
    ctx = C.c_void_p()
    assert lib.rthip_create(0, C.byref(ctx)) == 0
    p = make_params(64, 36, 1)
    # render before scene/camera
    assert lib.rthip_render(ctx, C.byref(p), None, None, None) == abi.E_STATE
    assert b"upload_scene" in lib.rthip_last_error(ctx)
    flat = flatten(presets.standalone_scene())
    assert lib.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
    assert lib.rthip_render(ctx, C.byref(p), None, None, None) == abi.E_STATE  # camera still missing
    # bad parameters
    for bad in (make_params(1, 36, 1), make_params(64, 1, 1), make_params(64, 36, 0), make_params(64, 36, 1, band_rows=8, band_stride=0)):
        assert lib.rthip_render(ctx, C.byref(bad), None, None, None) == abi.E_BADARG
    badflags = make_params(64, 36, 1, flags=1 << 20)
    assert lib.rthip_render(ctx, C.byref(badflags), None, None, None) == abi.E_BADARG
    # bad scene: material index out of range, texture index out of range, zero radius
    broken = flatten(presets.standalone_scene())
    broken.spheres["material"][0] = 99
    assert lib.rthip_upload_scene(ctx, C.byref(broken.desc)) == abi.E_BADARG
    broken = flatten(presets.standalone_scene())
    broken.materials["texture"][3] = 5
    assert lib.rthip_upload_scene(ctx, C.byref(broken.desc)) == abi.E_BADARG
    broken = flatten(presets.standalone_scene())
    broken.spheres["radius"][1] = 0.0
    assert lib.rthip_upload_scene(ctx, C.byref(broken.desc)) == abi.E_BADARG
    # coordinates that FP32 cannot square: rejected at the boundary instead of rendering infinities
    for bad_value in (1e19, float("inf"), float("nan")):
        broken = flatten(presets.standalone_scene())
        broken.spheres["center"][2][1] = bad_value
        assert lib.rthip_upload_scene(ctx, C.byref(broken.desc)) == abi.E_BADARG and b"sphere 2" in lib.rthip_last_error(ctx)
    assert lib.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
    # lights flag without lights
    cam = abi.Camera()
    lib.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double * 3)(0, 0, -1), C.byref(cam))
    assert lib.rthip_set_camera(ctx, C.byref(cam)) == 0
    pl = make_params(64, 36, 1, flags=abi.F_POINT_LIGHTS)
    assert lib.rthip_render(ctx, C.byref(pl), None, None, None) == abi.E_STATE
    # no device
    other = C.c_void_p()
    assert lib.rthip_create(4096, C.byref(other)) == abi.E_NODEVICE
    # a good call still works after the failures
    out = np.zeros((36, 64, 4), np.uint8)
    assert lib.rthip_render(ctx, C.byref(p), out.ctypes.data, None, None) == 0
    assert out[..., 3].min() == 255
    # a running sum holds at most 2^24 samples per pixel (the RNG counter of a sample would wrap)
    huge = make_params(64, 36, (1 << 24), flags=abi.F_ACCUMULATE)
    assert lib.rthip_render(ctx, C.byref(huge), None, None, None) == abi.E_BADARG and b"2^24" in lib.rthip_last_error(ctx)
    # ... and a path at most 255 segments (the counter RNG has 256 slots per sample: slot 0 = jitter, 1 + bounce = materials)
    deep = make_params(64, 36, 1, max_depth=256)
    assert lib.rthip_render(ctx, C.byref(deep), None, None, None) == abi.E_BADARG and b"255" in lib.rthip_last_error(ctx)
    assert lib.rthip_render(ctx, C.byref(make_params(64, 36, 1, max_depth=255)), None, None, None) == 0
    # rthip_pixel_status: needs an adaptive running sum, and buffers that hold it
    status = np.zeros(64 * 36, np.uint8)
    assert lib.rthip_pixel_status(ctx, status.ctypes.data, None, status.size) == abi.E_STATE
    ad = make_params(64, 36, 1, flags=abi.F_ADAPTIVE)
    assert lib.rthip_render(ctx, C.byref(ad), None, None, None) == 0
    rows, width, nb = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    assert lib.rthip_accum_layout(ctx, C.byref(rows), C.byref(width), C.byref(nb)) == 0 and (rows.value, width.value, nb.value) == (36, 64, 1)
    assert lib.rthip_pixel_status(ctx, status.ctypes.data, None, status.size - 1) == abi.E_BADARG
    assert lib.rthip_pixel_status(ctx, status.ctypes.data, None, status.size) == 0 and (status == abi.STATUS_UNSTABLE).all()
    lib.rthip_destroy(ctx)


def test_full_size_4k_properties(standalone_flat, oracle_mod, default_camera):
    """BASELINE.json's full size: 3840x2160. Size-independent properties, and every pixel of the frame against the f64 oracle:
    at 2 spp, and at the bench's own workload (config C2: 256 spp, depth 5, seed 1 — 4.07e9 rays, ~10 s of the host's cores)."""
    import zlib
    w, h, spp, depth = 3840, 2160, 2, 5
    rt = Raytracer.new(w, h, seed=71)
    whole = rt.render(standalone_flat, spp, max_depth=depth, flags=abi.F_COUNT_STATS)
    s = whole.stats
    assert s["paths"] == w * h * spp
    assert s["paths"] <= s["rays"] <= s["paths"] * depth
    assert sum(s["hits"]) + s["sky_evals"] >= s["rays"] and s["sky_evals"] <= s["paths"]
    assert whole.rgba8[..., 3].min() == 255
    # 8-way interleaved band split reassembles to the same bytes (checksum of checksums)
    full = np.empty_like(whole.rgba8)
    for rank in range(8):
        part = rt.render(standalone_flat, spp, max_depth=depth, band_rows=8, band_first=rank, band_stride=8, want_accum=False)
        full[owned_row_indices(h, 8, rank, 8)] = part.rgba8
    assert zlib.crc32(full.tobytes()) == zlib.crc32(whole.rgba8.tobytes())
    # sample slicing is a scheduling decision only: the whole frame (32 400 workgroups, unsliced) and 1/8-frame launches
    # (sliced into groups of 64 samples + resolve) give the same bytes and the same running sums, also when accumulating
    big = [rt.render(standalone_flat, 130, max_depth=depth, accumulate=k > 0) for k in range(2)][-1]
    parts = Raytracer.new(w, h, seed=71)
    acc8, img8 = np.empty_like(big.accum), np.empty_like(big.rgba8)
    for rank in range(8):
        rows = owned_row_indices(h, 8, rank, 8)
        for k in range(2):
            part = parts.render(standalone_flat, 130, max_depth=depth, accumulate=k > 0, band_rows=8, band_first=rank, band_stride=8)
        parts.invalidate_pixels()
        acc8[rows], img8[rows] = part.accum, part.rgba8
    assert np.array_equal(acc8, big.accum) and np.array_equal(img8, big.rgba8)
    assert big.stats["nb_samples"] == 260
    parts.close()
    # top rows are sky (blue channel saturated), bottom rows are the teal ground (red == 0)
    assert (whole.rgba8[-1, :, 2] == 255).all() and (whole.rgba8[0, :, 0] == 0).all()
    # tonemap identity on the device's own sums: u8 == trunc(255*sqrt(clamp(sum/n)))
    mean = np.clip(whole.accum / np.float32(spp), 0.0, 1.0)
    want = (np.sqrt(mean.astype(np.float32)) * np.float32(255.0)).astype(np.uint8)
    assert np.abs(want.astype(int) - whole.rgba8[..., :3].astype(int)).max() <= 1
    ref, _, st, _ = oracle_mod.render(standalone_flat, default_camera, make_params(w, h, spp, depth, 71), precision=64)
    hist, ok = assert_parity(whole.rgba8, ref, "C2 whole 4K frame, 2 spp, vs oracle f64")
    assert abs(s["rays"] - st["rays"]) <= 2e-3 * st["rays"]
    bench = Raytracer.new(w, h, seed=1)
    frame = bench.render(standalone_flat, 256, max_depth=5, want_accum=False)
    ref256, _, st256, secs = oracle_mod.render(standalone_flat, default_camera, make_params(w, h, 256, 5, 1), precision=64)
    hist256, ok256 = assert_parity(frame.rgba8, ref256, "C2 whole 4K frame, 256 spp (the bench workload), vs oracle f64")
    assert abs(frame.stats["rays"] - st256["rays"]) <= 1e-3 * st256["rays"]
    print(f"\nC2 4K vs oracle f64: 2 spp {ok * 100:.4f} % within 1/255 (max |delta| {int(np.nonzero(hist)[0].max())}); "
          f"256 spp {ok256 * 100:.4f} % (max |delta| {int(np.nonzero(hist256)[0].max())}; oracle {secs:.1f} s, rays device {frame.stats['rays']} oracle {st256['rays']})")
    bench.close()
    rt.close()


@pytest.mark.parametrize("name", ["C3", "C2+light"])
def test_full_size_4k_other_configs_vs_oracle(oracle_mod, default_camera, name):
    """The other 4K configurations of SURVEY §8d at their full size, every pixel against the f64 oracle: C3 (textured spheres +
    textured quad + ground, 64 spp, depth 50) and BASELINE's literal config 2 (the standalone scene with a point light:
    reflections at depth 5 + shadow rays; 32 spp here)."""
    w, h = 3840, 2160
    if name == "C3":
        flat, spp, depth, flags = flatten(presets.textured_scene()), 64, 50, 0
    else:
        flat, spp, depth, flags = flatten(presets.standalone_scene(), presets.standalone_lights()), 32, 5, abi.F_POINT_LIGHTS
    rt = Raytracer.new(w, h, seed=1)
    frame = rt.render(flat, spp, max_depth=depth, flags=flags, want_accum=False)
    ref, _, st, secs = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 1, flags), precision=64)
    hist, ok = assert_parity(frame.rgba8, ref, f"{name} whole 4K frame vs oracle f64")
    assert abs(frame.stats["rays"] - st["rays"]) <= 1e-3 * st["rays"]
    print(f"\n{name} 4K {spp} spp vs oracle f64 ({secs:.1f} s): {ok * 100:.4f} % within 1/255, max |delta| {int(np.nonzero(hist)[0].max())}; "
          f"rays device {frame.stats['rays']} oracle {st['rays']}")
    rt.close()


def test_full_size_8k_bvh_properties(oracle_mod, default_camera):
    """C5 at its full size: 7680x4320, 10 001 spheres + a 100 000-triangle mesh through the BVH, work pool and
    leaf-ordered records. Size-independent properties: counter identities, 8-way band split == whole frame (checksum of
    checksums), determinism, and BVH == brute force on a band the linear scan can afford. And every pixel of the 8K frame
    against the f64 oracle (answering from its own tree, oracle accel == the scan bit for bit), at depth 2 and at the config's
    depth 8, both at the strict bar — measured on MI355X in round 3: 99.9991 % and 99.965 % of the 33 M pixels within 1/255 at
    1 spp (round 2: 99.69 % at depth 8, put down to chaotic transport; it was FP32 cancellation in the sphere discriminant, see
    sphere_roots in rt_kernels.hip; the reference's own formula evaluated in float on the CPU is still at 99.6 %)."""
    import zlib
    w, h, spp, depth = 7680, 4320, 1, 8
    flat = flatten(presets.big_scene(100, 7, mesh_subdivisions="uv100k"))
    rt = Raytracer.new(w, h, seed=5)
    whole = rt.render(flat, spp, max_depth=depth, flags=abi.F_COUNT_STATS, want_accum=False)
    s = whole.stats
    assert s["paths"] == w * h * spp and s["paths"] <= s["rays"] <= s["paths"] * depth
    assert s["node_tests"] > 0 and s["node_tests"] % 2 == 0 and s["tri_tests"] > 0
    assert sum(s["hits"]) + s["sky_evals"] >= s["rays"]
    again = rt.render(flat, spp, max_depth=depth, want_accum=False)
    assert zlib.crc32(again.rgba8.tobytes()) == zlib.crc32(whole.rgba8.tobytes())
    full = np.empty_like(whole.rgba8)
    for rank in range(8):
        part = rt.render(flat, spp, max_depth=depth, band_rows=8, band_first=rank, band_stride=8, want_accum=False)
        full[owned_row_indices(h, 8, rank, 8)] = part.rgba8
    assert zlib.crc32(full.tobytes()) == zlib.crc32(whole.rgba8.tobytes())
    # one band of 8 rows through the middle of the image (61 440 pixels x 10 001 + 81 920 tests) by the linear scan of ray.rs:47-67
    mid = (h // 2) // 8
    nbands = h // 8
    brute = rt.render(flat, spp, max_depth=depth, flags=abi.F_NO_BVH, band_rows=8, band_first=mid, band_stride=nbands, want_accum=False)
    assert np.array_equal(brute.rgba8, whole.rgba8[mid * 8:(mid + 1) * 8])
    ref8, _, st8, secs8 = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 5), precision=64, accel=True)
    ref8_f32, _, _, _ = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 5), precision=32, accel=True)
    hist8, ok8 = mismatch_histogram(whole.rgba8, ref8)
    _, ok8_f32 = mismatch_histogram(ref8_f32, ref8)
    shallow = rt.render(flat, spp, max_depth=2, want_accum=False)
    ref2, _, st2, _ = oracle_mod.render(flat, default_camera, make_params(w, h, spp, 2, 5), precision=64, accel=True)
    hist2, ok2 = assert_parity(shallow.rgba8, ref2, "C5 whole 8K frame, depth 2, vs oracle f64 (accelerated)")
    print(f"\nC5 whole 8K frame vs oracle f64 with its tree ({secs8:.1f} s): depth 2 {describe(hist2, ok2)}\n"
          f"depth 8 {describe(hist8, ok8)} (f32-on-CPU vs f64: {ok8_f32 * 100:.3f} %)")
    assert ok8 >= 0.999, describe(hist8, ok8)   # the strict bar (docstring)
    assert abs(s["rays"] - st8["rays"]) <= 5e-3 * st8["rays"] and abs(shallow.stats["rays"] - st2["rays"]) <= 5e-3 * st2["rays"]
    rt.close()


def test_full_size_c4_million_triangles(oracle_mod, default_camera):
    """C4 at its full size: 49 icospheres x 20 480 = 1 003 520 triangles + the ground sphere, 3840x2160, depth 50:
    tree depth, quantised frames at that extent, the 48 MB record buffer and the leaf references of a 2^20-primitive tree.
    (i) counter identities, determinism, 8-way band split == whole frame (checksum of checksums); (ii) BVH == the linear scan
    of ray.rs:47-67 bit for bit on one 8-row band through the meshes; (iii) device vs the f64 oracle (which scans all
    1 003 521 shapes per ray) on a 48x48 window over a mesh silhouette, at the bar; (iv) the WHOLE 4K frame vs the f64 oracle
    answering from its own bounding-volume tree (oracle accel: the scan's answers bit for bit, tests/test_oracle_accel.py —
    checked here once more on the window of (iii)), at the bar, with the ray counts within 0.5 %."""
    import zlib
    w, h, spp, depth = 3840, 2160, 1, 50
    flat = flatten(presets.mesh_scene())
    assert len(flat.triangles) == 1003520 and len(flat.spheres) == 1
    rt = Raytracer.new(w, h, seed=9)
    whole = rt.render(flat, spp, max_depth=depth, flags=abi.F_COUNT_STATS)
    s = whole.stats
    assert s["paths"] == w * h * spp and s["paths"] <= s["rays"] <= s["paths"] * depth
    assert s["node_tests"] > 0 and s["node_tests"] % 2 == 0 and s["tri_tests"] > 0 and s["sphere_tests"] > 0
    assert sum(s["hits"]) + s["sky_evals"] >= s["rays"]
    assert s["hits"][abi.MAT_LAMBERTIAN] > 0 and s["hits"][abi.MAT_METAL] > 0 and s["hits"][abi.MAT_DIELECTRIC] == 0
    again = rt.render(flat, spp, max_depth=depth, want_accum=False)
    assert zlib.crc32(again.rgba8.tobytes()) == zlib.crc32(whole.rgba8.tobytes())
    full = np.empty_like(whole.rgba8)
    for rank in range(8):
        part = rt.render(flat, spp, max_depth=depth, band_rows=8, band_first=rank, band_stride=8, want_accum=False)
        full[owned_row_indices(h, 8, rank, 8)] = part.rgba8
    assert zlib.crc32(full.tobytes()) == zlib.crc32(whole.rgba8.tobytes())
    # (ii) a pixel on a mesh near the image centre (picking walks the same tree); the band of 8 rows through it by brute force
    target = None
    for dy in range(0, 800, 16):
        for dx in range(0, 1600, 16):
            for sx, sy in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
                x, y = w // 2 + sx * dx, h // 2 + sy * dy
                shape = rt.get_shape(flat, x, y)
                if shape is not None and shape >= 1:
                    target = (x, y)
                    break
            if target:
                break
        if target:
            break
    assert target, "no mesh in view"
    band = target[1] // 8
    brute = rt.render(flat, spp, max_depth=depth, flags=abi.F_NO_BVH | abi.F_COUNT_STATS, band_rows=8, band_first=band, band_stride=h // 8)
    assert brute.stats["node_tests"] == 0 and brute.stats["tri_tests"] == brute.stats["rays"] * 1003520
    assert np.array_equal(brute.rgba8, whole.rgba8[band * 8:(band + 1) * 8])
    assert np.array_equal(brute.accum.view(np.uint32), whole.accum[band * 8:(band + 1) * 8].view(np.uint32))
    # (iii) the oracle on a 48x48 window around that pixel (the first mesh pixel met walking outwards from the centre: a silhouette)
    x0, y0 = min(max(target[0] - 24, 0), w - 48), min(max(target[1] - 24, 0), h - 48)
    ref, _, st, secs = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 9), precision=64, crop=(x0, y0, 48, 48))
    hist, ok = assert_parity(whole.rgba8[y0:y0 + 48, x0:x0 + 48], ref, "C4 window vs oracle f64")
    print(f"\nC4 full size: {s['rays']} rays, {s['node_tests'] / s['rays']:.1f} node tests and {s['tri_tests'] / s['rays']:.2f} triangle tests per ray; "
          f"window ({x0},{y0}) 48x48 vs oracle f64 ({secs:.1f} s, {st['tri_tests'] / max(1, st['rays']):.0f} tests per ray): {describe(hist, ok)}")
    # (iv) every pixel of the frame
    ref_all, _, st_all, secs_all = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 9), precision=64, accel=True)
    assert np.array_equal(ref_all[y0:y0 + 48, x0:x0 + 48], ref), "the oracle's tree and its scan disagree"
    hist_all, ok_all = assert_parity(whole.rgba8, ref_all, "C4 whole 4K frame vs oracle f64 (accelerated)")
    assert abs(s["rays"] - st_all["rays"]) <= 5e-3 * st_all["rays"]
    print(f"C4 whole frame vs oracle f64 with its tree ({secs_all:.1f} s, {(st_all['tri_tests'] + st_all['sphere_tests']) / st_all['rays']:.1f} tests per ray): "
          f"{describe(hist_all, ok_all)}")
    rt.close()


def test_two_temporary_scenes_on_one_raytracer():
    """the scene is passed on every call like generate_pixel's `scene` argument (lib.rs:272): a second temporary list (whose
    id() CPython may well recycle from the first) and an in-place edit of a list must both reach the device"""
    rt = Raytracer.new(160, 90, seed=3)
    a = rt.render(presets.standalone_scene(), 2, max_depth=5)
    b = rt.render(presets.wasm_scene(), 2, max_depth=5)
    fresh = Raytracer.new(160, 90, seed=3)
    want_b = fresh.render(presets.wasm_scene(), 2, max_depth=5)
    assert np.array_equal(b.rgba8, want_b.rgba8) and not np.array_equal(a.rgba8, b.rgba8)
    scene = presets.wasm_scene()
    rt.render(scene, 2, max_depth=5)
    scene.append(Sphere.new((-0.9, 0.0, -1.0), 0.4, Lambertian.new((0.9, 0.1, 0.1))))
    c = rt.render(scene, 2, max_depth=5)
    want_c = fresh.render(list(scene), 2, max_depth=5)
    assert np.array_equal(c.rgba8, want_c.rgba8) and not np.array_equal(c.rgba8, want_b.rgba8)
    scene[-1] = Sphere.new((-0.9, 0.0, -1.0), 0.4, Lambertian.new((0.1, 0.9, 0.1)))   # same length, another shape
    d = rt.render(scene, 2, max_depth=5)
    assert not np.array_equal(d.rgba8, c.rgba8)
    rt.close(), fresh.close()


def test_bvh_point_lights_and_planes(oracle_mod, default_camera):
    """the extensions together: shadow-ray any-hit through the BVH (the shading lane's own, empty, stack column), an
    unbounded plane outside the tree, triangles with a textured material — BVH == linear scan bit for bit, and vs the oracle"""
    from rustracer_amd import Plane, PointLight
    shapes = presets.mesh_scene(5, 2) + [Plane.new((0.0, -0.45, 0.0), (0.0, 1.0, 0.0), Lambertian.new((0.6, 0.6, 0.2)))]
    flat = flatten(shapes, [PointLight((4.0, 5.0, 2.0), (30.0, 30.0, 30.0)), PointLight((-3.0, 2.0, 1.0), (6.0, 4.0, 2.0))])
    w, h, spp, depth = 256, 144, 3, 4
    rt = Raytracer.new(w, h, seed=23)
    a = rt.render(flat, spp, max_depth=depth, flags=abi.F_POINT_LIGHTS | abi.F_COUNT_STATS)
    b = rt.render(flat, spp, max_depth=depth, flags=abi.F_POINT_LIGHTS | abi.F_COUNT_STATS | abi.F_NO_BVH)
    assert a.stats["node_tests"] > 0 and b.stats["node_tests"] == 0 and a.stats["shadow_rays"] > 0
    assert np.array_equal(a.accum, b.accum) and a.stats["rays"] == b.stats["rays"] and a.stats["shadow_rays"] == b.stats["shadow_rays"]
    ref, _, st, _ = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 23, abi.F_POINT_LIGHTS), precision=64)
    hist, ok = assert_parity(a.rgba8, ref, "BVH + planes + point lights vs oracle f64")
    print(f"\nbvh+lights+plane: {describe(hist, ok)}; shadow rays device {a.stats['shadow_rays']} oracle {st['shadow_rays']}")
    assert abs(a.stats["shadow_rays"] - st["shadow_rays"]) <= 5e-3 * st["shadow_rays"]
    rt.close()


def test_page_locked_outputs(standalone_flat):
    """rthip_host_alloc: same bytes in a page-locked destination, and a faster device->host copy at 4K"""
    from rustracer_amd import host_array
    w, h = 3840, 2160
    rt = Raytracer.new(w, h, seed=2)
    pageable = rt.render(standalone_flat, 1, max_depth=5, want_accum=False)
    pinned = host_array((h, w, 4), np.uint8)
    pinned[:] = 0
    res = [rt.render(standalone_flat, 1, max_depth=5, want_accum=False, rgba_out=pinned) for _ in range(3)][-1]
    assert res.rgba8 is pinned and np.array_equal(pinned, pageable.rgba8)
    print(f"\n4K RGBA8 device->host: pageable {pageable.stats['d2h_ms']:.2f} ms, page-locked {res.stats['d2h_ms']:.2f} ms")
    assert res.stats["d2h_ms"] < pageable.stats["d2h_ms"]
    rt.close()
    lib = abi.lib()
    assert lib.rthip_host_alloc(0, C.byref(C.c_void_p())) == abi.E_BADARG and lib.rthip_host_free(None) == 0


def test_many_samples_sliced_vs_oracle(oracle_mod, default_camera, standalone_flat):
    """2050 spp on a small image: slices grow to 256 samples (at most 16 slices), the last one holds 2 samples; vs the oracle"""
    w, h, spp, depth, seed = 64, 36, 2050, 50, 8
    rt = Raytracer.new(w, h, seed=seed)
    res = rt.render(standalone_flat, spp, max_depth=depth)
    ref, acc, st, _ = oracle_mod.render(standalone_flat, default_camera, make_params(w, h, spp, depth, seed), precision=64)
    hist, ok = assert_parity(res.rgba8, ref, "2050 spp vs oracle f64")
    assert res.stats["paths"] == w * h * spp and abs(res.stats["rays"] - st["rays"]) <= 2e-3 * st["rays"]
    # converged radiance: the means agree closely (FP32 sums of 2050 samples in groups of 256)
    assert np.allclose(res.accum / spp, acc / spp, atol=4e-3)
    print(f"\n2050 spp: {describe(hist, ok)}; max |mean radiance delta| {np.abs(res.accum / spp - acc / spp).max():.2e}")
    rt.close()


def test_lambert_sky_integral_closed_form():
    """the device against a closed-form expectation (not the oracle): E[L] = albedo/pi*3 * E[sky] for a Lambertian plateau"""
    from test_oracle_kat import _lambert_plateau_scene, lambert_plateau_expected
    w, h, spp = 128, 72, 1024
    rt = Raytracer.new(w, h, seed=77)
    res = rt.render(_lambert_plateau_scene(), spp, max_depth=50, flags=abi.F_COUNT_STATS)
    got = (res.accum[4:24, 32:96] / spp).reshape(-1, 3).mean(axis=0)
    want = lambert_plateau_expected()
    print(f"\nLambertian plateau: device mean radiance {got}, closed form {want}")
    assert np.allclose(got, want, rtol=3e-3)
    # every ground path is exactly two segments (convex surface): camera -> ground -> sky
    assert res.stats["hits"][abi.MAT_LAMBERTIAN] + res.stats["paths"] == res.stats["rays"]
    rt.close()


def test_empty_tile_fast_path_is_exact(standalone_flat):
    """waves whose pixels cannot reach any sphere skip the scene query (conservative cone test per pixel). The instrumented
    kernel has no such path: both must give the same bits — on the standalone frame and on random scenes of small spheres
    whose silhouettes cross many tiles, including spheres behind the camera and one around it."""
    from rustracer_amd import Metal
    rt = Raytracer.new(3840, 2160, seed=4)
    a = rt.render(standalone_flat, 4, max_depth=5)   # (the fast loops need >= 4 samples per workgroup, RT_FAST_MIN_SPP)
    b = rt.render(standalone_flat, 4, max_depth=5, flags=abi.F_COUNT_STATS)
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgba8, b.rgba8)
    assert a.stats["rays"] == b.stats["rays"] and a.stats["paths"] == b.stats["paths"]
    rt.close()
    rng = np.random.default_rng(5)
    w, h = 640, 360
    rt = Raytracer.new(w, h, seed=6)
    n_sky_only = 0
    for trial in range(24):
        n = int(rng.integers(1, 9))
        scene = []
        for _ in range(n):
            c = rng.uniform(-4, 4, 3) + np.array([0.0, 0.0, -3.0])
            scene.append(Sphere.new(c, float(rng.choice([0.01, 0.05, 0.3, 1.0])), Metal.new((0.8, 0.8, 0.8), 0.0)))
        if trial % 6 == 5:
            scene.append(Sphere.new((-1.8, 1.0, 2.0), 0.5, Lambertian.new((0.5, 0.5, 0.5))))   # the camera sits inside this one
        flat = flatten(scene)
        a = rt.render(flat, 4, max_depth=4)
        b = rt.render(flat, 4, max_depth=4, flags=abi.F_COUNT_STATS)
        assert np.array_equal(a.accum, b.accum), f"trial {trial}: {(a.accum != b.accum).any(axis=2).sum()} pixels differ"
        assert a.stats["rays"] == b.stats["rays"]
        n_sky_only += int(b.stats["rays"] == b.stats["paths"])
    assert n_sky_only < 24   # the scenes do put spheres in view
    rt.close()


@pytest.mark.parametrize("lights", [False, True])
def test_primary_inline_loop_is_exact(standalone_flat, lights):
    """Waves whose primary rays can reach at most half of the spheres trace the primary ray inline, against those spheres
    only (per-sphere cone test), and the sky-only waves skip the scene altogether — with and without point lights. The
    instrumented kernel has neither loop: same bits, same ray counts. Scenes: the standalone frame, and random scenes of
    up to 70 spheres of all four materials (more than 64 spheres: the loop must step aside) over a ground sphere, every
    third of them with 1-4 triangles (the loop takes scenes with a few triangles: C3's quad), and the C3 scene itself."""
    from rustracer_amd import Metal, Dielectric, PointLight
    fl = abi.F_POINT_LIGHTS if lights else 0
    flat = flatten(presets.standalone_scene(), presets.standalone_lights()) if lights else standalone_flat
    rt = Raytracer.new(1920, 1080, seed=14)
    a = rt.render(flat, 5, max_depth=5, flags=fl)
    b = rt.render(flat, 5, max_depth=5, flags=fl | abi.F_COUNT_STATS)
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgba8, b.rgba8)
    assert a.stats["rays"] == b.stats["rays"] and a.stats["paths"] == b.stats["paths"]
    rt.close()
    flat3 = flatten(presets.textured_scene(), presets.standalone_lights() if lights else ())  # C3: 10 spheres + a quad of two triangles
    rt3 = Raytracer.new(960, 540, seed=17)
    a = rt3.render(flat3, 4, max_depth=50, flags=fl)
    b = rt3.render(flat3, 4, max_depth=50, flags=fl | abi.F_COUNT_STATS)
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgba8, b.rgba8) and a.stats["rays"] == b.stats["rays"]
    rt3.close()
    flatp = flatten(presets.plane_scene(), presets.standalone_lights() if lights else ())
    rtp = Raytracer.new(960, 540, seed=18)
    a = rtp.render(flatp, 4, max_depth=50, flags=fl)
    b = rtp.render(flatp, 4, max_depth=50, flags=fl | abi.F_COUNT_STATS)
    assert np.array_equal(a.accum, b.accum) and np.array_equal(a.rgba8, b.rgba8) and a.stats["rays"] == b.stats["rays"]
    rtp.close()
    rng = np.random.default_rng(15)
    rt = Raytracer.new(640, 360, seed=16)
    tex = presets.standalone_scene()[3].material
    for trial in range(12):
        n = int(rng.choice([2, 3, 7, 20, 63, 70]))
        scene = [Sphere.new((0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070))]
        for _ in range(n):
            c = rng.uniform(-3, 3, 3) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 0.0, -2.0])
            kind = int(rng.integers(0, 4))
            mat = (Lambertian.new(rng.uniform(0.1, 0.9, 3)), Metal.new(rng.uniform(0.3, 0.9, 3), float(rng.uniform(0, 0.5))),
                   Dielectric.new((1.0, 0.9, 0.9), 1.3), tex)[kind]
            scene.append(Sphere.new(c, float(rng.choice([0.05, 0.2, 0.5])), mat))
        if trial % 3 == 1:  # a ground plane (and sometimes a second, tilted one) instead of / beside the ground sphere
            from rustracer_amd import Plane
            scene.append(Plane.new((0.0, -0.45, 0.0), (0.0, 1.0, 0.0), (tex, scene[0].material)[trial % 2]))
            if trial % 2:
                scene.append(Plane.new((0.0, 0.0, -8.0), (0.2, 0.1, 1.0), scene[1].material))
        if trial % 3 == 2:  # a few triangles beside the spheres: the loop tests them all on every primary ray
            from rustracer_amd import Triangle
            for _ in range(int(rng.integers(1, 5))):
                v = rng.uniform(-2.5, 2.5, 3) * np.array([1.0, 0.3, 1.0]) + np.array([0.0, 0.4, -2.5])
                scene.append(Triangle.new(v, v + rng.uniform(-1.5, 1.5, 3), v + rng.uniform(-1.5, 1.5, 3), (tex, scene[1].material)[int(rng.integers(0, 2))]))
        flat = flatten(scene, [PointLight((3.0, 4.0, 2.0), (20.0, 20.0, 20.0))] if lights else ())
        a = rt.render(flat, 4, max_depth=6, flags=fl)
        b = rt.render(flat, 4, max_depth=6, flags=fl | abi.F_COUNT_STATS)
        assert np.array_equal(a.accum, b.accum), f"trial {trial} ({n} spheres): {(a.accum != b.accum).any(axis=2).sum()} pixels differ"
        assert a.stats["rays"] == b.stats["rays"] and a.stats["paths"] == b.stats["paths"]
    rt.close()


@pytest.mark.parametrize("name", ["C5", "C4"])
def test_bvh_configs_at_their_own_settings(oracle_mod, default_camera, name):
    """the BVH configs against the f64 oracle at THEIR OWN depth and sample count (C5: depth 8, 64 spp at 8K;
    C4: depth 50, 16 spp at 4K), on a 1920x1080 window of the frame (what the oracle's host threads trace in ~10 s), at the strict
    bar of north_star: per-channel |delta| <= 1/255 on >= 99.9 % of the pixels. (The 1-spp whole-frame comparisons above are the
    worst case for chaotic transport — one path per pixel; at the configs' sample counts a diverging path is one of 16 / 64.)"""
    if name == "C5":
        scene, (w, h, spp, depth) = presets.big_scene(100, 7, mesh_subdivisions="uv100k"), (7680, 4320, 64, 8)
        window = (2880, 1080, 1920, 1080)  # x0, y0, width, height: rows 1080..2159 = band 1 of bands of 1080 rows
    else:
        scene, (w, h, spp, depth) = presets.mesh_scene(), (3840, 2160, 16, 50)
        window = (960, 0, 1920, 1080)      # the lower half of the frame: meshes everywhere
    flat = flatten(scene)
    rt = Raytracer.new(w, h, seed=1)
    band = rt.render(flat, spp, max_depth=depth, band_rows=1080, band_first=window[1] // 1080, band_stride=1000, want_accum=False)
    got = band.rgba8[:, window[0]:window[0] + window[2]]
    ref, _, st, secs = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 1), precision=64, crop=window, accel=True)
    hist, ok = mismatch_histogram(got, ref)
    ref32, _, _, _ = oracle_mod.render(flat, default_camera, make_params(w, h, spp, depth, 1), precision=32, crop=window, accel=True)
    hist32, ok32 = mismatch_histogram(ref32, ref)
    _, ok_dev32 = mismatch_histogram(got, ref32)
    print(f"\n{name} at its own settings ({w}x{h}, depth {depth}, {spp} spp), window {window} vs oracle f64 ({secs:.1f} s, {st['rays']} rays): {describe(hist, ok)}\n"
          f"   the oracle's own formulas in float on the CPU vs f64: {describe(hist32, ok32)}; device vs that float render: {ok_dev32 * 100:.3f} %")
    for d in (2, 4):
        shallow = rt.render(flat, spp, max_depth=d, band_rows=1080, band_first=window[1] // 1080, band_stride=1000, want_accum=False).rgba8[:, window[0]:window[0] + window[2]]
        refd, _, _, _ = oracle_mod.render(flat, default_camera, make_params(w, h, spp, d, 1), precision=64, crop=window, accel=True)
        print(f"   depth {d}: {describe(*mismatch_histogram(shallow, refd))}")
    assert ok >= 0.999
    rt.close()
