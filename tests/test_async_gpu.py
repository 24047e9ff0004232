"""rthip_render_async / rthip_wait (include/rthip.h): the progressive loop of standalone/src/main.rs:99-102 with two frames in
flight. Every frame must equal the one rthip_render returns for the same pass."""
import ctypes as C

import numpy as np
import pytest

from rustracer_amd import Raytracer, abi, host_array, make_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("adaptive", [False, True])
def test_async_frames_equal_synchronous_ones(standalone_flat, adaptive):
    w, h, n = 480, 270, 12
    rt = Raytracer.new(w, h, seed=8)
    want = [rt.render(standalone_flat, 1, max_depth=50, accumulate=k > 0, adaptive=adaptive, want_accum=False).rgba8.copy() for k in range(n)]
    rt.close()
    rt = Raytracer.new(w, h, seed=8)
    frames = [host_array((h, w, 4), np.uint8), host_array((h, w, 4), np.uint8)]
    rt.render_async(standalone_flat, 1, frames[0], 0, adaptive=adaptive)
    for k in range(1, n):
        rt.render_async(standalone_flat, 1, frames[k & 1], k & 1, accumulate=True, adaptive=adaptive)
        rt.wait((k - 1) & 1)
        assert np.array_equal(frames[(k - 1) & 1], want[k - 1]), k - 1
    rt.wait((n - 1) & 1)
    assert np.array_equal(frames[(n - 1) & 1], want[n - 1])
    assert rt.accum_layout()[2] == n
    rt.close()


def test_async_call_order_errors(standalone_flat):
    lib = abi.lib()
    rt = Raytracer.new(64, 40, seed=1)
    rt.upload(standalone_flat)
    rt._sync_scene_camera(None)
    p = make_params(64, 40, 1)
    buf = np.zeros((40, 64, 4), np.uint8)  # pageable memory works too (the copy is then staged by the runtime)
    assert lib.rthip_wait(rt._ctx, 0) == abi.E_STATE
    assert lib.rthip_render_async(rt._ctx, C.byref(p), buf.ctypes.data, 2) == abi.E_BADARG
    assert lib.rthip_render_async(rt._ctx, C.byref(p), None, 0) == abi.E_BADARG
    assert lib.rthip_render_async(rt._ctx, C.byref(p), buf.ctypes.data, 0) == 0
    assert lib.rthip_render_async(rt._ctx, C.byref(p), buf.ctypes.data, 0) == abi.E_STATE
    assert lib.rthip_wait(rt._ctx, 0) == 0 and buf[..., 3].min() == 255
    assert lib.rthip_wait(rt._ctx, 0) == abi.E_STATE
    rt.close()
