"""ctypes mirror of include/rthip.h — the C ABI of the gfx950 render path.

The structures here are byte-for-byte the PODs of ``include/rthip.h``; the same structures are
handed to the CPU oracle by the tests (the oracle consumes the identical scene description).

The product library ``librthip.so`` is loaded lazily by :func:`lib`; if it is missing the call
FAILS LOUDLY — there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 4

# rthip_status
OK, E_BADARG, E_HIP, E_OOM, E_NODEVICE, E_STATE, E_INTERNAL, E_RCCL = 0, -1, -2, -3, -4, -5, -6, -7

MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_TEXTURE = 0, 1, 2, 3

F_ACCUMULATE = 1 << 0
F_POINT_LIGHTS = 1 << 1
F_COUNT_STATS = 1 << 2
F_NO_BVH = 1 << 3
F_ADAPTIVE = 1 << 4
F_TILE_LISTS = 1 << 5
F_COUNT_EXECUTED = 1 << 6
GROUP_F_PEER_COPY = 1 << 0
GROUP_F_FORCE_RCCL = 1 << 1
NOALLOC_SCRATCH_BYTES = 64 << 20
STATUS_NOT_STARTED, STATUS_COPY_NEAR_PIXEL, STATUS_UNSTABLE, STATUS_FINAL = 0, 1, 2, 3  # GenerationStatus, lib.rs:59-64

_d3 = C.c_double * 3


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", _d3), ("param", C.c_double),
                ("shininess", C.c_double)]


class Texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb8", C.c_void_p)]


class Sphere(C.Structure):
    _fields_ = [("center", _d3), ("radius", C.c_double), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class Plane(C.Structure):
    _fields_ = [("point", _d3), ("normal", _d3), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class Triangle(C.Structure):
    _fields_ = [("v0", _d3), ("v1", _d3), ("v2", _d3), ("material", C.c_uint32), ("_pad", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("position", _d3), ("intensity", _d3)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("n_spheres", C.c_uint64),
                ("planes", C.c_void_p), ("n_planes", C.c_uint64),
                ("triangles", C.c_void_p), ("n_triangles", C.c_uint64),
                ("materials", C.c_void_p), ("n_materials", C.c_uint64),
                ("textures", C.c_void_p), ("n_textures", C.c_uint64),
                ("lights", C.c_void_p), ("n_lights", C.c_uint64)]


class Camera(C.Structure):
    _fields_ = [("origin", _d3), ("d00", _d3), ("horizontal", _d3), ("vertical", _d3)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint32), ("flags", C.c_uint32), ("band_rows", C.c_uint32), ("band_first", C.c_uint32),
                ("band_stride", C.c_uint32), ("reserved", C.c_uint32 * 7)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("plane_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("node_tests", C.c_uint64),
                ("hits", C.c_uint64 * 4), ("sky_evals", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("rows", C.c_uint64), ("nb_samples", C.c_uint64), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double),
                ("pixels_traced", C.c_uint64), ("pixels_final", C.c_uint64), ("gather_ms", C.c_double)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "hits" else getattr(self, n)) for n, _ in self._fields_}


# every symbol include/rthip.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "rthip_device_count": (C.c_int, [_P(C.c_int)]),
    "rthip_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "rthip_upload_scene": (C.c_int, [C.c_void_p, _P(SceneDesc)]),
    "rthip_set_camera": (C.c_int, [C.c_void_p, _P(Camera)]),
    "rthip_camera_lookat": (C.c_int, [C.c_double, C.c_double, C.c_double, _P(C.c_double), _P(Camera)]),
    "rthip_owned_rows": (C.c_int, [_P(RenderParams), _P(C.c_uint32)]),
    "rthip_render": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_void_p, C.c_void_p, _P(Stats)]),
    "rthip_render_device": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, _P(Stats)]),
    "rthip_render_async": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_void_p, C.c_uint32]),
    "rthip_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rthip_reset_accum": (C.c_int, [C.c_void_p]),
    "rthip_render_device_noalloc": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, _P(Stats)]),
    "rthip_pixel_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rthip_accum_layout": (C.c_int, [C.c_void_p, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint64)]),
    "rthip_group_create": (C.c_int, [_P(C.c_int), C.c_int, C.c_uint32, _P(C.c_void_p)]),
    "rthip_group_upload_scene": (C.c_int, [C.c_void_p, _P(SceneDesc)]),
    "rthip_group_set_camera": (C.c_int, [C.c_void_p, _P(Camera)]),
    "rthip_group_render": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_void_p, C.c_void_p, _P(Stats)]),
    "rthip_group_reset_accum": (C.c_int, [C.c_void_p]),
    "rthip_group_pixel_status": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "rthip_group_size": (C.c_int, [C.c_void_p]),
    "rthip_group_context": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rthip_group_last_error": (C.c_char_p, [C.c_void_p]),
    "rthip_group_destroy": (None, [C.c_void_p]),
    "rthip_band_row_owner": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32), _P(C.c_uint32)]),
    "rthip_host_alloc": (C.c_int, [C.c_size_t, _P(C.c_void_p)]),
    "rthip_host_free": (C.c_int, [C.c_void_p]),
    "rthip_pick": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, _P(C.c_int64)]),
    "rthip_algorithmic_flops": (C.c_double, [_P(Stats)]),
    "rthip_last_error": (C.c_char_p, [C.c_void_p]),
    "rthip_status_string": (C.c_char_p, [C.c_int]),
    "rthip_abi_version": (C.c_int, []),
    "rthip_destroy": (None, [C.c_void_p]),
}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTHIP_LIB") or os.path.join(_HERE, "csrc", "librthip.so")  # RTHIP_LIB: A/B builds of the same ABI

_lib = None


class RthipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rthip status {status}: {message}")
        self.status = status


def lib():
    """Load librthip.so (the HIP extension). Raises if it has not been built — no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.rthip_abi_version() != ABI_VERSION:
            raise RuntimeError("librthip.so ABI version mismatch")
        _lib = l
    return _lib
