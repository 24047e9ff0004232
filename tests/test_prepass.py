"""Primary-visibility lists (rustracer_amd/csrc/rt_prepass.h), checked on the CPU through the host mirror of
prepass_kernel: the list of a pixel must hold the leaf of whatever the linear scan of ray.rs:47-67 (the oracle's
find_collision) returns for ANY primary ray of that pixel — or the pixel's cut distance must say "go on traversing"."""
import ctypes as C

import numpy as np
import pytest

import oracle
from rustracer_amd import abi, flatten, presets
from rustracer_amd.scene import Camera

LEAF_BIT, DONE, OVERFLOW = 0x80000000, 0x7FFFFFFF, 0xFFFFFFFF
TILE_DT = np.dtype([("ref", np.uint32, 64), ("near", np.float32, 64), ("mask", np.uint64, 64), ("cut", np.float32, 64)])
# RtTilePre is {ent[64] {ref, near}, mask[64], cut[64]}: read the interleaved entries through a raw view
RAW_DT = np.dtype([("ent", np.uint32, 128), ("mask", np.uint64, 64), ("cut", np.float32, 64)])


def host_lists(flat, cam, w, h):
    lib = abi.lib()
    ntx, nty = (w + 7) // 8, (h + 7) // 8
    raw = np.zeros(ntx * nty, RAW_DT)
    assert raw.itemsize == 1280
    lc = np.zeros(ntx * nty, np.uint32)
    pc = np.zeros(ntx * nty * 64, np.uint32)
    fn = lib.rt_prepass_host
    fn.restype = C.c_int
    assert fn(C.byref(flat.desc), C.byref(cam), w, h, 0, 0, ntx, nty, raw.ctypes.data_as(C.c_void_p), lc.ctypes.data_as(C.c_void_p),
              pc.ctypes.data_as(C.c_void_p)) == 0
    n = len(flat.spheres) + len(flat.triangles)
    prims = np.zeros(n, np.uint32)
    fp = lib.rt_prepass_host_prims
    fp.restype = C.c_longlong
    assert fp(C.byref(flat.desc), prims.ctypes.data_as(C.c_void_p), C.c_ulonglong(n)) == n
    ent = raw["ent"].reshape(-1, 64, 2)
    return ent[:, :, 0].copy(), ent[:, :, 1].copy().view(np.float32), raw["mask"], raw["cut"], lc, pc.reshape(-1, 64), prims


def scene_small():
    return flatten(presets.big_scene(12, 7, 2))  # 1 + 144 spheres + 320 triangles


def scene_dense():
    """hundreds of small spheres behind each other along the view direction: tiles with more leaves than a list holds"""
    rng = np.random.Generator(np.random.PCG64(5))
    from rustracer_amd.scene import Lambertian, SphereCloud, Vector3
    n = 4000
    depth = rng.random(n) * 40.0 + 2.0
    # image-plane positions from a normal distribution: the tiles in the middle of the picture touch more than RT_PRE_COLLECT
    # leaves (no list at all), those around them more than RT_PRE_LEAVES (a list of the nearest + cut distances)
    c = np.stack([rng.normal(0.0, 0.18, n) * depth, rng.normal(0.0, 0.1, n) * depth, -depth], 1) + np.array([0.0, 0.0, -1.0])
    return flatten([SphereCloud(c, 0.015 * depth, [Lambertian.new(Vector3(0.5, 0.5, 0.5))], np.zeros(n, np.int64))])


@pytest.mark.parametrize("which", ["small", "dense"])
def test_lists_hold_every_primary_hit(which):
    w, h = 64, 40
    flat = scene_small() if which == "small" else scene_dense()
    cam_obj = Camera.new(-1.8, 1.0, 2.0) if which == "small" else Camera.new_lookat(0.0, 0.0, 1.0, (0.0, 0.0, -1.0))
    cam = cam_obj.as_abi()
    ref, near, mask, cut, lc, pc, prims = host_lists(flat, cam, w, h)
    ntx = (w + 7) // 8
    ns = len(flat.spheres)
    rng = np.random.Generator(np.random.PCG64(11))
    checked = in_list = beyond_cut = overflow = 0
    for y in range(0, h, 3):
        for x in range(0, w, 3):
            t = (y // 8) * ntx + (x // 8)
            p = (x & 7) | ((y & 7) << 3)
            if ref[t, 0] == OVERFLOW:
                overflow += 1
                continue
            cand = set()
            for i in range(64):
                if (int(mask[t, p]) >> i) & 1:
                    r = int(ref[t, i])
                    assert r & LEAF_BIT and r != OVERFLOW
                    first, count = (r & ~LEAF_BIT) >> 2, (r & 3) + 1
                    for j in range(first, first + count):
                        pid = int(prims[j])
                        cand.add(pid if pid < ns else pid + len(flat.planes))  # global shape index (spheres, planes, triangles)
            for jit in [(0.0, 0.0), (0.999999, 0.999999), (0.0, 0.999999), (0.999999, 0.0), tuple(rng.random(2)), tuple(rng.random(2))]:
                o, d = oracle.emit_ray_at(cam, (x + jit[0]) / (w - 1), (y + jit[1]) / (h - 1))
                for prec in (32, 64):
                    idx, tt, _ = oracle.find_collision(flat, o, d, prec)
                    if idx is None:
                        continue
                    checked += 1
                    if idx in cand:
                        in_list += 1
                    else:
                        # the leaf did not fit the tile's list: then the pixel's cut distance must not exceed the hit distance
                        t_unit = tt * float(np.linalg.norm(d))
                        assert cut[t, p] <= t_unit * (1 + 1e-5), (x, y, jit, idx, t_unit, cut[t, p])
                        beyond_cut += 1
    assert checked > 500 and in_list > 0
    print(f"{which}: {checked} primary hits checked, {in_list} in the list, {beyond_cut} behind the cut distance, {overflow} pixels in overflowing tiles")
    if which == "dense":
        assert beyond_cut > 0 or overflow > 0  # the scene is there to exercise them


def test_list_structure():
    w, h = 96, 56
    flat = scene_small()
    cam = Camera.new(-1.8, 1.0, 2.0).as_abi()
    ref, near, mask, cut, lc, pc, prims = host_lists(flat, cam, w, h)
    for t in range(len(ref)):
        if ref[t, 0] == OVERFLOW:
            assert (mask[t] == 0).all() and np.isinf(cut[t]).all()
            continue
        n = min(int(lc[t]), 64)
        assert (ref[t, :n] & LEAF_BIT).all() and (ref[t, n:] == DONE).all() and np.isinf(near[t, n:]).all()
        key = list(zip(near[t, :n].tolist(), ref[t, :n].tolist()))
        assert key == sorted(key), "entries are in order of (near, reference)"
        assert len(set(ref[t, :n].tolist())) == n
        assert (near[t, :n] >= 0).all()
        used = np.bitwise_or.reduce(mask[t])
        assert int(used) >> n == 0, "mask bits beyond the list"
        if lc[t] <= 64:
            assert np.isinf(cut[t]).all()
    # a pixel cannot touch a leaf its tile does not
    assert (pc.max(axis=1) <= lc).all()
