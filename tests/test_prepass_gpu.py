"""Primary-visibility lists on the device (rt_prepass.h): prepass_kernel's records equal the host mirror's, and the image
is the one the plain traversal (the default) and the reference's linear scan (RTHIP_F_NO_BVH, ray.rs:47-67)
give — bit for bit, running sum included — for every launch shape that takes lists."""
import ctypes as C

import numpy as np
import pytest

from rustracer_amd import Raytracer, abi, flatten, make_params, presets
from rustracer_amd.scene import Camera
from test_prepass import OVERFLOW, RAW_DT, host_lists, scene_dense, scene_small

pytestmark = pytest.mark.gpu


def device_lists(rt, ntiles_groups):
    fn = abi.lib().rt_ctx_tile_lists
    fn.restype = C.c_longlong
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong]
    have = fn(rt._ctx, None, 0)
    assert have == ntiles_groups * 4 * 1280, have
    raw = np.zeros(ntiles_groups * 4, RAW_DT)
    assert fn(rt._ctx, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == have
    return raw


@pytest.mark.parametrize("which", ["small", "dense"])
def test_device_lists_equal_the_host_mirror(which):
    w, h = (256, 144) if which == "small" else (64, 40)  # (big tiles on the dense cloud: lists that overflow, cut distances)
    flat = scene_small() if which == "small" else scene_dense()
    rt = Raytracer.new(w, h, seed=3)
    if which == "dense":
        rt.camera = Camera.new_lookat(0.0, 0.0, 1.0, (0.0, 0.0, -1.0))
    rt.render(flat, 4, max_depth=3, want_accum=False, flags=abi.F_TILE_LISTS)
    gx, gy = (w + 31) // 32, (h + 7) // 8
    raw = device_lists(rt, gx * gy)
    ref, near, mask, cut, lc, pc, prims = host_lists(flat, rt.camera.as_abi(), w, h)
    ntx = (w + 7) // 8
    ent = raw["ent"].reshape(-1, 64, 2)
    n_over = n_cut = 0
    for g in range(gx * gy):
        for wv in range(4):
            tx, ty = (g % gx) * 4 + wv, g // gx
            if tx >= ntx:
                continue
            d, t = g * 4 + wv, ty * ntx + tx
            if ref[t, 0] == OVERFLOW or ent[d, 0, 0] == OVERFLOW:
                # the device's walk is breadth-first with a bounded frontier: it may give up where the host (depth-first,
                # unbounded) does not, never the other way round
                assert ent[d, 0, 0] == OVERFLOW, (g, wv)
                n_over += 1
                continue
            assert (ent[d, :, 0] == ref[t]).all(), (g, wv)
            assert (ent[d, :, 1].view(np.float32) == near[t]).all(), (g, wv)
            assert (raw["mask"][d] == mask[t]).all(), (g, wv)
            assert (raw["cut"][d] == cut[t]).all(), (g, wv)
            n_cut += int(np.isfinite(cut[t]).any())
    print(f"{which}: {gx * gy * 4} tile records compared, {n_over} without a list, {n_cut} with cut distances")
    if which == "dense":
        assert n_over > 0 and n_cut > 0
    rt.close()


def _three_ways(rt, flat, spp, depth, lights=(), **kw):
    fl = abi.F_POINT_LIGHTS if lights else 0
    a = rt.render(flat, spp, max_depth=depth, flags=fl | abi.F_TILE_LISTS, lights=lights, **kw)
    b = rt.render(flat, spp, max_depth=depth, flags=fl, lights=lights, **kw)
    c = rt.render(flat, spp, max_depth=depth, flags=fl | abi.F_NO_BVH, lights=lights, **kw)
    return a, b, c


@pytest.mark.parametrize("which,lights", [("small", False), ("small", True), ("dense", False)])
def test_lists_do_not_change_the_image(which, lights):
    w, h = (320, 176) if which == "small" else (72, 40)
    sc = presets.big_scene(12, 7, 2) if which == "small" else None
    lt = presets.standalone_lights() if lights else ()
    flat = flatten(sc, lt) if sc is not None else scene_dense()
    rt = Raytracer.new(w, h, seed=5)
    if which == "dense":
        rt.camera = Camera.new_lookat(0.0, 0.0, 1.0, (0.0, 0.0, -1.0))
    a, b, c = _three_ways(rt, flat, 8, 6, lt)
    assert (a.rgba8 == b.rgba8).all() and (a.accum == b.accum).all(), "lists vs plain traversal"
    assert (a.rgba8 == c.rgba8).all() and (a.accum == c.accum).all(), "lists vs the linear scan"
    assert a.stats["rays"] == b.stats["rays"] == c.stats["rays"]
    # the instrumented kernel takes the lists too, and counts what it tests
    s1 = rt.render(flat, 8, max_depth=6, flags=abi.F_COUNT_STATS | abi.F_TILE_LISTS | (abi.F_POINT_LIGHTS if lights else 0), lights=lt)
    s0 = rt.render(flat, 8, max_depth=6, flags=abi.F_COUNT_STATS | (abi.F_POINT_LIGHTS if lights else 0), lights=lt)
    assert (s1.rgba8 == a.rgba8).all() and (s0.rgba8 == a.rgba8).all()
    assert s1.stats["node_tests"] < s0.stats["node_tests"]
    print(f"{which} lights={lights}: node tests per ray {s0.stats['node_tests'] / s0.stats['rays']:.1f} -> {s1.stats['node_tests'] / s1.stats['rays']:.1f}, "
          f"primitive tests per ray {(s0.stats['tri_tests'] + s0.stats['sphere_tests']) / s0.stats['rays']:.2f} -> "
          f"{(s1.stats['tri_tests'] + s1.stats['sphere_tests']) / s1.stats['rays']:.2f}")
    rt.close()


def test_lists_with_bands_slices_and_running_sums():
    w, h = 256, 136  # 17 tile rows: the last band is short
    flat = scene_small()
    rt = Raytracer.new(w, h, seed=9)
    whole = rt.render(flat, 160, max_depth=4, flags=abi.F_TILE_LISTS)  # 160 spp: sample-sliced
    plain = rt.render(flat, 160, max_depth=4)
    assert (whole.rgba8 == plain.rgba8).all() and (whole.accum == plain.accum).all()
    # 3-way band split, bands of 8 and of 4 rows (a wave tile then spans two bands: the tile's pyramid is their hull)
    for band_rows in (8, 4):
        img = np.zeros_like(whole.rgba8)
        for r in range(3):
            part = rt.render(flat, 160, max_depth=4, band_rows=band_rows, band_first=r, band_stride=3, flags=abi.F_TILE_LISTS)
            rows = [y for y in range(h) if (y // band_rows) % 3 == r]
            img[rows] = part.rgba8
        assert (img == whole.rgba8).all(), band_rows
    # progressive passes: the second makes the lists (a 2-sample pass that starts a frame does not), the third reuses them;
    # the same passes without lists give the same running sum, bit for bit
    sums = {}
    for fl in (abi.F_TILE_LISTS, 0):
        rt.invalidate_pixels()
        rt.render(flat, 2, max_depth=4, flags=fl)
        rt.render(flat, 3, max_depth=4, accumulate=True, flags=fl)
        sums[fl] = rt.render(flat, 3, max_depth=4, accumulate=True, flags=fl)
    assert (sums[0].rgba8 == sums[abi.F_TILE_LISTS].rgba8).all() and (sums[0].accum == sums[abi.F_TILE_LISTS].accum).all()
    one = rt.render(flat, 8, max_depth=4, flags=abi.F_TILE_LISTS)  # ... and it is the sum of the same 8 samples (added in another order)
    np.testing.assert_allclose(sums[0].accum, one.accum, rtol=1e-5, atol=1e-6)
    rt.close()
