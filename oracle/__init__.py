"""ctypes binding of the CPU oracle (oracle/liboracle.so, built from rt_oracle.hpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package (rustracer_amd/). It consumes the same POD scene description
(include/rthip.h) as the device library.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rustracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
_P = C.POINTER
_d3 = C.c_double * 3


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB_PATH)
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [_P(abi.SceneDesc), _P(abi.Camera), _P(abi.RenderParams), _P(C.c_uint32), C.c_uint64, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, _P(abi.Stats), _P(C.c_double)]
        l.oracle_hw_threads.restype = C.c_int
        l.oracle_camera_lookat.argtypes = [C.c_double, C.c_double, C.c_double, _P(C.c_double), _P(abi.Camera)]
        l.oracle_camera_lookat.restype = None
        l.oracle_emit_ray_at.argtypes = [_P(abi.Camera), C.c_double, C.c_double, _P(C.c_double), _P(C.c_double)]
        l.oracle_emit_ray_at.restype = None
        l.oracle_find_collision.argtypes = [_P(abi.SceneDesc), _P(C.c_double), _P(C.c_double), C.c_int, _P(C.c_double), _P(C.c_double)]
        l.oracle_find_collision.restype = C.c_int64
        l.oracle_sphere_uv.argtypes = [_P(C.c_double), C.c_double, _P(C.c_double), _P(C.c_double), _P(C.c_double)]
        l.oracle_sphere_uv.restype = None
        l.oracle_texture_wrap.argtypes = [C.c_double, C.c_double, C.c_uint32]
        l.oracle_texture_wrap.restype = C.c_uint32
        l.oracle_background_color.argtypes = [_P(C.c_double), _P(C.c_double)]
        l.oracle_background_color.restype = None
        l.oracle_tonemap.argtypes = [C.c_double, C.c_uint64]
        l.oracle_tonemap.restype = C.c_uint8
        l.oracle_tonemap_f32.argtypes = [C.c_float, C.c_uint64]
        l.oracle_tonemap_f32.restype = C.c_uint8
        l.oracle_cast_u8.argtypes = [C.c_double]
        l.oracle_cast_u8.restype = C.c_uint8
        l.oracle_cast_i32.argtypes = [C.c_double]
        l.oracle_cast_i32.restype = C.c_int32
        l.oracle_schlick.argtypes = [C.c_double, C.c_double]
        l.oracle_schlick.restype = C.c_double
        l.oracle_refract.argtypes = [_P(C.c_double), _P(C.c_double), C.c_double, _P(C.c_double)]
        l.oracle_refract.restype = C.c_int
        l.oracle_reflect.argtypes = [_P(C.c_double), _P(C.c_double), _P(C.c_double)]
        l.oracle_reflect.restype = None
        l.oracle_path_key.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        l.oracle_path_key.restype = C.c_uint32
        l.oracle_rng_first.argtypes = [C.c_uint32, C.c_uint32]
        l.oracle_rng_first.restype = C.c_uint32
        l.oracle_rng_second.argtypes = [C.c_uint32]
        l.oracle_rng_second.restype = C.c_uint32
        l.oracle_rng_pair.argtypes = [C.c_uint32, C.c_uint32, _P(C.c_double), _P(C.c_double)]
        l.oracle_rng_pair.restype = None
        l.oracle_random_unit_vector.argtypes = [C.c_uint32, C.c_uint32, _P(C.c_double)]
        l.oracle_random_unit_vector.restype = None
        l.oracle_project_ray.argtypes = [_P(abi.SceneDesc), _P(C.c_double), _P(C.c_double), C.c_uint32, C.c_uint32, C.c_int, _P(C.c_double),
                                         _P(C.c_uint64)]
        l.oracle_project_ray.restype = None
        _lib = l
    return _lib


def hw_threads():
    return lib().oracle_hw_threads()


def camera_lookat(x, y, z, lookat=(0.0, 0.0, -1.0)) -> abi.Camera:
    out = abi.Camera()
    lib().oracle_camera_lookat(x, y, z, _d3(*lookat), C.byref(out))
    return out


def emit_ray_at(cam: abi.Camera, ox, oy):
    o, d = _d3(), _d3()
    lib().oracle_emit_ray_at(C.byref(cam), ox, oy, o, d)
    return np.array(o), np.array(d)


def find_collision(flat, origin, direction, precision=64):
    t, pos = C.c_double(0), _d3()
    idx = lib().oracle_find_collision(C.byref(flat.desc), _d3(*origin), _d3(*direction), precision, C.byref(t), pos)
    return (None, None, None) if idx < 0 else (idx, t.value, np.array(pos))


def sphere_uv(center, radius, pos):
    u, v = C.c_double(0), C.c_double(0)
    lib().oracle_sphere_uv(_d3(*center), radius, _d3(*pos), C.byref(u), C.byref(v))
    return u.value, v.value


def background_color(direction):
    out = _d3()
    lib().oracle_background_color(_d3(*direction), out)
    return np.array(out)


def project_ray(flat, origin, direction, max_depth=50, key=0, point_lights=False):
    rgb, seg = _d3(), C.c_uint64(0)
    lib().oracle_project_ray(C.byref(flat.desc), _d3(*origin), _d3(*direction), max_depth, key, int(point_lights), rgb, C.byref(seg))
    return np.array(rgb), seg.value


class AdaptiveState:
    """PixelCache arrays (lib.rs:66-73) carried between the passes of render_adaptive."""

    def __init__(self, w, h):
        self.rgba = np.zeros((h, w, 4), np.uint8)
        self.sum = np.zeros((h, w, 3), np.float64)
        self.status = np.zeros((h, w), np.uint8)
        self.same_color_count = np.zeros((h, w), np.uint8)
        self.last_color = np.zeros((h, w, 3), np.uint8)
        self.nb_samples = np.zeros((h, w), np.uint64)


def render_adaptive(flat, cam: abi.Camera, params: abi.RenderParams, state: AdaptiveState, precision=64, threads=0):
    """One progressive pass of generate_pixel over every pixel with the convergence rule (lib.rs:276-313).
    params.flags must carry F_ACCUMULATE on every pass but the first. Returns the stats dict."""
    if threads <= 0:
        threads = hw_threads()
    st = abi.Stats()
    fn = lib().oracle_render_adaptive
    fn.restype = C.c_int
    rc = fn(C.byref(flat.desc), C.byref(cam), C.byref(params), None, C.c_int(precision), C.c_int(threads),
            C.c_void_p(state.rgba.ctypes.data), C.c_void_p(state.sum.ctypes.data), C.c_void_p(state.status.ctypes.data),
            C.c_void_p(state.same_color_count.ctypes.data), C.c_void_p(state.last_color.ctypes.data),
            C.c_void_p(state.nb_samples.ctypes.data), C.byref(st))
    if rc != 0:
        raise ValueError("oracle_render_adaptive: bad argument")
    return st.as_dict()


def render(flat, cam: abi.Camera, params: abi.RenderParams, precision=64, threads=0, crop=None, prior_samples=0, sum_in=None, accel=False):
    """Returns (rgba8 (h,w,4) u8, sum (h,w,3) f64, stats dict, seconds). Rows are y-up (row 0 = bottom).
    accel: answer the closest-hit / any-hit queries from a bounding-volume tree over the spheres and triangles instead of the
    reference's linear scan — the same answers bit for bit (tests/test_oracle_accel.py), for scenes the scan cannot finish;
    `seconds` then excludes the tree build and the test counters count the tests actually made."""
    if threads <= 0:
        threads = hw_threads()
    if crop is None:
        w, h, cropp = params.width, params.height, None
    else:
        w, h = crop[2], crop[3]
        cropp = (C.c_uint32 * 4)(*crop)
    rgba = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3), np.float64) if sum_in is None else np.ascontiguousarray(sum_in, dtype=np.float64).copy()
    st, secs = abi.Stats(), C.c_double(0)
    fn = lib().oracle_render_accel
    fn.restype = C.c_int
    rc = fn(C.byref(flat.desc), C.byref(cam), C.byref(params), cropp, C.c_uint64(prior_samples), C.c_int(precision), C.c_int(threads),
            C.c_int(1 if accel else 0), C.c_void_p(rgba.ctypes.data), C.c_void_p(acc.ctypes.data), C.byref(st), C.byref(secs))
    if rc != 0:
        raise ValueError("oracle_render: bad argument")
    return rgba, acc, st.as_dict(), secs.value
