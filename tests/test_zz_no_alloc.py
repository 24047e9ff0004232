"""rthip_render_device_noalloc (include/rthip.h): the pass for callers that cannot allocate — the counter shards and the
partial sums of a sample-sliced launch live in a static per-GPU device array, the running sum in the caller's d_accum.
Called through the C ABI and compared with rthip_render_device bit for bit, on two calls in a row, the second with a smaller
input (the array keeps the first call's contents). Collected last (test_zz_*): a failure here must not hide other tests."""
import ctypes as C

import numpy as np
import pytest
import torch

from rustracer_amd import Raytracer, abi, flatten, make_params, presets

pytestmark = pytest.mark.gpu


def _device_pass(lib, ctx, entry, p, want_stats=True):
    """one pass through `entry` (rthip_render_device or ..._noalloc) into fresh torch buffers; returns (rgba, accum, stats)"""
    rows = C.c_uint32(0)
    assert lib.rthip_owned_rows(C.byref(p), C.byref(rows)) == 0
    rgba = torch.full((rows.value, p.width, 4), 7, dtype=torch.uint8, device="cuda")
    acc = torch.full((rows.value, p.width, 3), -1.0, dtype=torch.float32, device="cuda")
    st = abi.Stats()
    stream = torch.cuda.current_stream().cuda_stream
    rc = entry(ctx, C.byref(p), rgba.data_ptr(), acc.data_ptr(), stream, C.byref(st) if want_stats else None)
    assert rc == 0, lib.rthip_last_error(ctx)
    torch.cuda.synchronize()
    return rgba.cpu().numpy(), acc.cpu().numpy(), st.as_dict()


def test_noalloc_matches_render_device_twice():
    lib = abi.lib()
    flat = flatten(presets.standalone_scene())
    ref_rt, na_rt = Raytracer.new(640, 360, seed=3), Raytracer.new(640, 360, seed=3)
    for rt in (ref_rt, na_rt):
        rt.upload(flat)
        rt._sync_scene_camera(None)
    # first call: 640x360 at 130 spp = a sample-sliced launch (3 slices: 64 + 64 + 2 -> partial sums in the scratch) with
    # counters (shards in the scratch); second call: a SMALLER image, band-split, fewer samples, over the first call's leftovers
    cases = [make_params(640, 360, 130, 5, 3, abi.F_COUNT_STATS),
             make_params(320, 96, 70, 50, 4, abi.F_COUNT_STATS, band_rows=8, band_first=1, band_stride=2)]
    for p in cases:
        want = _device_pass(lib, ref_rt._ctx, lib.rthip_render_device, p)
        got = _device_pass(lib, na_rt._ctx, lib.rthip_render_device_noalloc, p)
        assert np.array_equal(got[0], want[0]), "RGBA8 differs"
        assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), "running sum differs"
        for k in ("paths", "rays", "sphere_tests", "hits", "sky_evals", "rows", "nb_samples", "pixels_traced"):
            assert got[2][k] == want[2][k], k
        assert got[2]["paths"] == p.spp * p.width * got[2]["rows"]
    ref_rt.close(), na_rt.close()


def test_noalloc_accumulates_in_the_callers_buffer_and_rejects_what_it_cannot_hold():
    lib = abi.lib()
    flat = flatten(presets.standalone_scene())
    w, h = 256, 144
    ref_rt, na_rt = Raytracer.new(w, h, seed=9), Raytracer.new(w, h, seed=9)
    for rt in (ref_rt, na_rt):
        rt.upload(flat)
        rt._sync_scene_camera(None)
    stream = torch.cuda.current_stream().cuda_stream
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for k in range(3):  # three passes of 4 spp into ONE caller buffer (no stats: nothing synchronises, nothing is counted)
        p = make_params(w, h, 4, 50, 9, abi.F_ACCUMULATE if k else 0)
        assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(p), rgba.data_ptr(), acc.data_ptr(), stream, None) == 0
    torch.cuda.synchronize()
    for k in range(3):
        want = ref_rt.render(flat, 4, max_depth=50, accumulate=k > 0)
    assert np.array_equal(rgba.cpu().numpy(), want.rgba8) and np.array_equal(acc.cpu().numpy(), want.accum)
    # another buffer cannot continue that running sum; adaptive passes and missing d_accum are refused; so is a sliced launch
    # whose partial sums exceed the static array (4 slices x 1920 x 1080 x 12 B = 99.5 MB > 64 MiB)
    other = torch.zeros_like(acc)
    p = make_params(w, h, 4, 50, 9, abi.F_ACCUMULATE)
    assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(p), rgba.data_ptr(), other.data_ptr(), stream, None) == abi.E_STATE
    p = make_params(w, h, 4, 50, 9, abi.F_ADAPTIVE)
    assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(p), rgba.data_ptr(), acc.data_ptr(), stream, None) == abi.E_BADARG
    p = make_params(w, h, 4, 50, 9)
    assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(p), rgba.data_ptr(), None, stream, None) == abi.E_BADARG
    big = make_params(1920, 1080, 256, 5, 1)
    big_acc = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(big), None, big_acc.data_ptr(), stream, None) == abi.E_BADARG
    assert b"scratch" in lib.rthip_last_error(na_rt._ctx)
    # ... while 64 spp per pass (unsliced) of the same frame is fine
    ok = make_params(1920, 1080, 64, 5, 1)
    assert lib.rthip_render_device_noalloc(na_rt._ctx, C.byref(ok), None, big_acc.data_ptr(), stream, None) == 0
    torch.cuda.synchronize()
    ref_rt.close(), na_rt.close()
