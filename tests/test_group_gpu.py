"""rthip_group_* (include/rthip.h): ONE call returns the whole frame, rendered as interleaved row bands by
every device of the group and gathered to the first. The one-GPU box checks: a single-device group == rthip_render byte for
byte; RCCL forced on that single device (ncclCommInitAll + grouped send/recv to itself); and 2-, 3- and 8-way splits
rehearsed on one GPU by listing device 0 several times (RTHIP_GROUP_F_PEER_COPY: the same band logic, de-interleave kernel
and row map, with peer copies in place of the RCCL transfers)."""
import ctypes as C

import numpy as np
import pytest

from rustracer_amd import GroupRaytracer, Raytracer, abi, flatten, presets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def whole(standalone_flat):
    rt = Raytracer.new(320, 181, seed=17)
    res = rt.render(standalone_flat, 5, max_depth=6, flags=abi.F_COUNT_STATS)
    rt.close()
    return res


@pytest.mark.parametrize("devices,band_rows,kw", [
    ((0,), 8, {}),
    ((0,), 8, {"force_rccl": True}),
    ((0, 0), 8, {"peer_copy": True}),
    ((0, 0, 0), 16, {"peer_copy": True}),
    ((0,) * 8, 7, {"peer_copy": True}),
])
def test_group_frame_is_byte_identical(standalone_flat, whole, devices, band_rows, kw):
    g = GroupRaytracer(320, 181, seed=17, devices=devices, **kw)
    assert g.size == len(devices)
    g.upload(standalone_flat)
    res = g.render(5, max_depth=6, flags=abi.F_COUNT_STATS, band_rows=band_rows)
    assert np.array_equal(res.rgba8, whole.rgba8) and np.array_equal(res.accum, whole.accum)
    for k in ("paths", "rays", "sphere_tests", "hits", "sky_evals"):
        assert res.stats[k] == whole.stats[k], k
    assert res.stats["rows"] == 181 and res.stats["kernel_ms"] > 0 and res.stats["gather_ms"] >= 0
    # the running sum continues across group passes like rthip_render's
    more = g.render(3, max_depth=6, accumulate=True, band_rows=band_rows)
    assert more.stats["nb_samples"] == 8
    g.close()


def test_group_accumulates_like_one_device(standalone_flat):
    rt = Raytracer.new(200, 120, seed=2)
    for k in range(2):
        want = rt.render(standalone_flat, 70, max_depth=5, accumulate=k > 0)   # 70 spp: sample-sliced launches on every device
    rt.close()
    g = GroupRaytracer(200, 120, seed=2, devices=(0, 0, 0), peer_copy=True)
    g.upload(standalone_flat)
    for k in range(2):
        got = g.render(70, max_depth=5, accumulate=k > 0)
    assert np.array_equal(got.rgba8, want.rgba8) and np.array_equal(got.accum, want.accum)
    g.invalidate_pixels()
    fresh = g.render(70, max_depth=5, accumulate=True)   # nothing to add to after a reset
    assert fresh.stats["nb_samples"] == 70
    g.close()


def test_group_errors():
    lib = abi.lib()
    grp = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.rthip_group_create(devs, 2, 0, C.byref(grp)) == abi.E_BADARG and b"twice" in lib.rthip_group_last_error(None)
    assert lib.rthip_group_create(devs, 0, 0, C.byref(grp)) == abi.E_BADARG
    assert lib.rthip_group_create(None, 1, 0, C.byref(grp)) == abi.E_BADARG
    bad = (C.c_int * 1)(4096)
    assert lib.rthip_group_create(bad, 1, 0, C.byref(grp)) == abi.E_NODEVICE
    one = (C.c_int * 1)(0)
    assert lib.rthip_group_create(one, 1, 0, C.byref(grp)) == 0
    from rustracer_amd import make_params
    p = make_params(64, 36, 1)
    out = np.zeros((36, 64, 4), np.uint8)
    assert lib.rthip_group_render(grp, C.byref(p), out.ctypes.data, None, None) == abi.E_STATE   # no scene yet
    flat = flatten(presets.standalone_scene())
    assert lib.rthip_group_upload_scene(grp, C.byref(flat.desc)) == 0
    cam = abi.Camera()
    lib.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double * 3)(0, 0, -1), C.byref(cam))
    assert lib.rthip_group_set_camera(grp, C.byref(cam)) == 0
    p2 = make_params(64, 36, 1, band_rows=8, band_first=1, band_stride=2)
    assert lib.rthip_group_render(grp, C.byref(p2), out.ctypes.data, None, None) == abi.E_BADARG
    p3 = make_params(64, 36, 1, flags=abi.F_ADAPTIVE)
    assert lib.rthip_group_render(grp, C.byref(p3), out.ctypes.data, None, None) == abi.E_BADARG
    assert lib.rthip_group_render(grp, C.byref(p), out.ctypes.data, None, None) == 0, lib.rthip_group_last_error(grp)
    assert out[..., 3].min() == 255
    assert lib.rthip_group_context(grp, 0) and not lib.rthip_group_context(grp, 1)
    lib.rthip_group_destroy(grp)
    lib.rthip_group_destroy(None)
