"""Random-scene sweep (tools/fuzz_parity.py draws the scenes): all four materials, noise textures with odd scales, negative
radii, overlapping spheres, planes / triangles / point lights, the camera inside a sphere, 1..300 spheres through the LDS scan,
the BVH and the forced linear scan, odd image sizes, depths 1..50. For every scene:
  * the production kernel and the instrumented kernel (no empty-tile path, other register allocation) agree bit for bit;
  * device vs oracle f64 meets the bar (|delta| <= 1/255 on >= 99.9 % of pixels), strictly. A scene that legitimately cannot
    (tools/fuzz_parity.py found ~1 in 10 among 810 random scenes: noise textures on extension planes seen at grazing angles,
    hundreds of glass spheres at depth 50 with few samples) must be NAMED in KNOWN_BELOW_BAR with its measured fractions and
    the reason; for such a seed the test then demands the measured value (minus one pixel) and "no worse than the same
    formulas in float on the CPU" (the oracle's f32 mode). None of the 24 seeds run here needs an entry."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from parity import mismatch_histogram  # noqa: E402

from rustracer_amd import Raytracer, abi, flatten, make_params  # noqa: E402
from rustracer_amd.scene import Camera  # noqa: E402

pytestmark = pytest.mark.gpu

# seed -> (device fraction within 1/255 vs oracle f64, f32-on-CPU fraction vs f64, why) as measured on MI355X.
# Round 2 (GPUTEST log gpurun_out/r2a): every seed of 100..123 is at or above 99.9 %, so the table is empty and the bar is strict.
KNOWN_BELOW_BAR = {}


@pytest.mark.parametrize("seed", list(range(100, 124)))
def test_random_scene(oracle_mod, seed):
    from fuzz_parity import random_scene
    rng = np.random.default_rng(seed)
    shapes, lights, pose, flavour = random_scene(rng)
    w, h = int(rng.choice([33, 64, 97, 160])), int(rng.choice([17, 36, 55, 90]))
    spp, depth = int(rng.choice([1, 3, 8, 70])), int(rng.choice([1, 2, 5, 50]))
    flags = (abi.F_POINT_LIGHTS if lights else 0) | (abi.F_NO_BVH if rng.random() < 0.25 else 0)
    flat = flatten(shapes, lights)
    rt = Raytracer.new(w, h, seed=seed)
    rt.camera = Camera.new(*pose)
    plain = rt.render(flat, spp, max_depth=depth, flags=flags)
    counted = rt.render(flat, spp, max_depth=depth, flags=flags | abi.F_COUNT_STATS)
    assert np.array_equal(plain.accum, counted.accum) and plain.stats["rays"] == counted.stats["rays"]
    cam = oracle_mod.camera_lookat(*pose)
    p = make_params(w, h, spp, depth, seed, flags)
    ref64, _, st, _ = oracle_mod.render(flat, cam, p, precision=64)
    _, ok = mismatch_histogram(plain.rgba8, ref64)
    if seed in KNOWN_BELOW_BAR:
        measured, _, why = KNOWN_BELOW_BAR[seed]
        ref32, _, _, _ = oracle_mod.render(flat, cam, p, precision=32)
        _, ok32 = mismatch_histogram(ref32, ref64)
        print(f"\nseed {seed} ({why}): device {ok * 100:.3f} % vs f32-on-CPU {ok32 * 100:.3f} % ({len(flat.spheres)} spheres, flavour {flavour}, {w}x{h}@{spp} d{depth})")
        assert ok >= measured - 1.0 / (w * h) and ok >= ok32 - 1.0 / (w * h), f"seed {seed}: device {ok}, recorded {measured}, float oracle {ok32}"
    else:
        assert ok >= 0.999, f"seed {seed}: device within 1/255 on {ok * 100:.3f} % of pixels ({len(flat.spheres)} spheres, flavour {flavour}, {w}x{h}@{spp} d{depth})"
    assert counted.stats["paths"] == st["paths"] and abs(counted.stats["rays"] - st["rays"]) <= 5e-3 * st["rays"] + 4
    rt.close()


@pytest.mark.parametrize("seed", list(range(200, 212)))
def test_random_band_layouts_are_byte_identical(seed):
    """any interleaved band layout (band heights that are not multiples of the 8-row wave tile, ragged last bands, more ranks
    than bands), any sample count (sliced or not), with and without accumulation: the union of the ranks' rows == the whole frame"""
    from fuzz_parity import random_scene
    from rustracer_amd.bands import owned_row_indices
    rng = np.random.default_rng(seed)
    shapes, lights, pose, _ = random_scene(rng)
    w, h = int(rng.choice([40, 96, 130])), int(rng.choice([9, 50, 77]))
    band_rows, world = int(rng.choice([1, 3, 7, 8, 16])), int(rng.choice([2, 3, 5, 8]))
    spp, depth, passes = int(rng.choice([1, 5, 70, 130])), int(rng.choice([2, 6])), int(rng.choice([1, 2]))
    flags = abi.F_POINT_LIGHTS if lights else 0
    flat = flatten(shapes, lights)
    whole = Raytracer.new(w, h, seed=seed)
    whole.camera = Camera.new(*pose)
    for k in range(passes):
        full = whole.render(flat, spp, max_depth=depth, flags=flags, accumulate=k > 0)
    img, acc = np.zeros_like(full.rgba8), np.zeros_like(full.accum)
    for rank in range(world):
        rt = Raytracer.new(w, h, seed=seed)
        rt.camera = Camera.new(*pose)
        for k in range(passes):
            part = rt.render(flat, spp, max_depth=depth, flags=flags, accumulate=k > 0, band_rows=band_rows, band_first=rank, band_stride=world)
        rows = owned_row_indices(h, band_rows, rank, world)
        assert part.rgba8.shape[0] == len(rows)
        img[rows], acc[rows] = part.rgba8, part.accum
        rt.close()
    assert np.array_equal(img, full.rgba8) and np.array_equal(acc, full.accum), f"{w}x{h} bands of {band_rows} over {world}, {spp} spp x {passes}"
    whole.close()


@pytest.mark.parametrize("seed", [3, 11, 42, 77, 1003, 1200])
def test_random_geometry_bvh_equals_linear_scan(seed):
    """tools/fuzz_bvh.py (1100 scenes run on the MI355X in round 1): clouds of 97-6000 spheres / triangles at coordinate scales
    1e-3..1e5, far from the origin, flat, with needles and duplicates — the traversal of the quantised tree returns exactly
    what the linear scan returns (ray.rs:47-67), bit for bit"""
    import fuzz_bvh
    rng = np.random.default_rng(seed)
    shapes, lights, eye, lookat, _ = fuzz_bvh.scene(rng)
    flat = flatten(shapes, lights)
    rt = Raytracer.new(97, 55, seed=seed)
    rt.camera = Camera.new_lookat(eye[0], eye[1], eye[2], tuple(lookat))
    fl = abi.F_COUNT_STATS | (abi.F_POINT_LIGHTS if lights else 0)
    a = rt.render(flat, 3, max_depth=6, flags=fl)
    b = rt.render(flat, 3, max_depth=6, flags=fl | abi.F_NO_BVH)
    assert a.stats["node_tests"] > 0 and b.stats["node_tests"] == 0 and a.stats["rays"] == b.stats["rays"]
    assert np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32)) and np.array_equal(a.rgba8, b.rgba8)
    rt.close()
