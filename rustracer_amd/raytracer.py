"""Host-side mirror of raytracer_core's driver API over the C ABI (include/rthip.h).

    Raytracer::new(width, height, random)         raytracer_core/src/lib.rs:250
    pub camera: Camera                            raytracer_core/src/lib.rs:240
    Raytracer::generate_pixel(gen, scene, samples, renderer)   lib.rs:269-316
    Raytracer::get_shape(scene, x, y)             lib.rs:262-267
    trait PixelRenderer { set_pixel, invalidate_pixels }        lib.rs:77-80
    PixelColor / PixelPosition / GenerationStatus lib.rs:17-64

The device path has batch semantics (the loop of wasm/src/lib.rs:112-123): one `render` call is
`generate_pixel` for EVERY pixel with the same `samples`; the progressive pixel scheduler
(RandomGenerator, lib.rs:82-226) is replaced by whole-image passes that add to a device-resident
running sum (accumulate=True == calling generate_pixel again on every pixel, lib.rs:289-294).

Every compute call goes through librthip.so; if the library is missing, import of this module
still works but the first call raises — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from enum import Enum
from typing import Optional

import numpy as np

from . import abi
from .scene import Camera, FlatScene, flatten


class GenerationStatus(Enum):  # lib.rs:58-64
    NotStarted = 0
    CopyNearPixel = 1
    Unstable = 2
    Final = 3


@dataclass
class PixelColor:  # lib.rs:17-22
    r: int
    g: int
    b: int
    status: GenerationStatus = GenerationStatus.Unstable

    def __eq__(self, other):  # lib.rs:24-28: status is not compared
        return (self.r, self.g, self.b) == (other.r, other.g, other.b)


@dataclass
class PixelPosition:  # lib.rs:41-45
    x: int
    y: int


class PixelRenderer:  # lib.rs:77-80
    def set_pixel(self, pos: PixelPosition, color: PixelColor) -> None:
        raise NotImplementedError

    def invalidate_pixels(self) -> None:
        raise NotImplementedError

    def set_image(self, rgba8: np.ndarray) -> None:
        """Bulk form used by the device path: (H, W, 4) uint8, row 0 = y = 0 (bottom). The default
        forwards to set_pixel pixel by pixel with status Final, y-up like the reference."""
        h, w, _ = rgba8.shape
        for y in range(h):
            row = rgba8[y]
            for x in range(w):
                self.set_pixel(PixelPosition(x, y), PixelColor(int(row[x, 0]), int(row[x, 1]), int(row[x, 2]), GenerationStatus.Final))


@dataclass
class RenderResult:
    rgba8: Optional[np.ndarray]  # (rows, W, 4) uint8, owned rows in increasing y, row 0 = bottom
    accum: Optional[np.ndarray]  # (rows, W, 3) float32 running radiance sum
    stats: dict


def _check(status, ctx=None):
    if status != 0:
        msg = abi.lib().rthip_last_error(ctx)
        raise abi.RthipError(status, (msg or b"").decode() or abi.lib().rthip_status_string(status).decode())


def make_params(width, height, spp, max_depth=50, seed=0, flags=0, band_rows=0, band_first=0, band_stride=1) -> abi.RenderParams:
    p = abi.RenderParams()
    p.width, p.height, p.spp, p.max_depth, p.seed, p.flags = int(width), int(height), int(spp), int(max_depth), int(seed) & 0xFFFFFFFF, int(flags)
    p.band_rows, p.band_first, p.band_stride = int(band_rows), int(band_first), int(band_stride)
    return p


def owned_rows(params: abi.RenderParams) -> int:
    rows = C.c_uint32(0)
    _check(abi.lib().rthip_owned_rows(C.byref(params), C.byref(rows)))
    return rows.value


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        _check(abi.lib().rthip_host_alloc(nbytes, C.byref(self.ptr)))

    def __del__(self):
        try:
            abi.lib().rthip_host_free(self.ptr)
        except Exception:
            pass


def host_array(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (rthip_host_alloc), for the image outputs of Raytracer.render"""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    block = _PinnedBlock(max(1, n * dtype.itemsize))
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(block.ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    buf._rthip_block = block
    return arr


class Raytracer:
    """Raytracer<R> with the GPU behind it. `seed` replaces the `random: R` argument (the
    reference's generators are entropy-seeded)."""

    def __init__(self, width, height, seed=0, device=0):
        self._lib = abi.lib()
        self.width, self.height = float(width), float(height)  # lib.rs:232-233 keeps them as f64
        self.seed = int(seed)
        self._ctx = C.c_void_p()
        _check(self._lib.rthip_create(int(device), C.byref(self._ctx)))
        self.camera = Camera.new(-1.8, 1.0, 2.0)  # lib.rs:252
        self._uploaded_camera = None
        self._scene_key = None
        self._flat = None

    @classmethod
    def new(cls, width, height, seed=0, device=0):
        return cls(width, height, seed, device)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.rthip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene / camera upload ----
    def upload(self, scene, lights=()):
        flat = scene if isinstance(scene, FlatScene) else flatten(scene, lights)
        _check(self._lib.rthip_upload_scene(self._ctx, C.byref(flat.desc)), self._ctx)
        self._flat = flat
        self._scene_key = id(scene)
        return flat

    def _sync_scene_camera(self, scene, lights=()):
        if scene is not None and self._scene_key != id(scene):
            self.upload(scene, lights)
        cam = self.camera.as_abi()
        key = bytes(cam)
        if key != self._uploaded_camera:
            _check(self._lib.rthip_set_camera(self._ctx, C.byref(cam)), self._ctx)
            self._uploaded_camera = key

    # ---- the batch form of generate_pixel ----
    def render(self, scene, samples, renderer: Optional[PixelRenderer] = None, max_depth=50, accumulate=False, flags=0,
               want_accum=True, band_rows=0, band_first=0, band_stride=1, lights=(), adaptive=False, rgba_out=None) -> RenderResult:
        """adaptive: the convergence rule of lib.rs:276-313 — Final pixels are skipped by later (accumulate=True) passes."""
        self._sync_scene_camera(scene, lights)
        fl = flags | (abi.F_ACCUMULATE if accumulate else 0) | (abi.F_ADAPTIVE if adaptive else 0)
        p = make_params(int(self.width), int(self.height), samples, max_depth, self.seed, fl, band_rows, band_first, band_stride)
        rows = owned_rows(p)
        self._last_rows = rows
        w = int(self.width)
        if rgba_out is not None:  # e.g. a page-locked array from host_array(): a much faster device->host copy
            assert rgba_out.shape == (rows, w, 4) and rgba_out.dtype == np.uint8 and rgba_out.flags.c_contiguous
        rgba = rgba_out if rgba_out is not None else np.empty((rows, w, 4), np.uint8)
        acc = np.empty((rows, w, 3), np.float32) if want_accum else None
        st = abi.Stats()
        _check(self._lib.rthip_render(self._ctx, C.byref(p), rgba.ctypes.data, acc.ctypes.data if acc is not None else None, C.byref(st)), self._ctx)
        if renderer is not None:
            renderer.set_image(rgba)
        return RenderResult(rgba, acc, st.as_dict())

    def render_device(self, params: abi.RenderParams, d_rgba8=0, d_accum=0, stream=0, want_stats=False):
        """rthip_render_device with raw device pointers (ints), e.g. torch tensors' data_ptr()."""
        self._sync_scene_camera(None)
        st = abi.Stats()
        _check(self._lib.rthip_render_device(self._ctx, C.byref(params), d_rgba8 or None, d_accum or None, stream or None,
                                             C.byref(st) if want_stats else None), self._ctx)
        return st.as_dict() if want_stats else None

    def pixel_status(self):
        """PixelColor.status / PixelCache.same_color_count of every owned pixel after the last adaptive pass (lib.rs:59-73)."""
        # the layout of the last pass is the context's; the full image unless bands were used
        rows = self._last_rows if getattr(self, "_last_rows", None) else int(self.height)
        status = np.empty((rows, int(self.width)), np.uint8)
        count = np.empty((rows, int(self.width)), np.uint8)
        _check(self._lib.rthip_pixel_status(self._ctx, status.ctypes.data, count.ctypes.data), self._ctx)
        return status, count

    def invalidate_pixels(self):
        _check(self._lib.rthip_reset_accum(self._ctx), self._ctx)

    def get_shape(self, scene, x, y) -> Optional[int]:
        """lib.rs:262-267"""
        self._sync_scene_camera(scene)
        out = C.c_int64(-1)
        _check(self._lib.rthip_pick(self._ctx, int(self.width), int(self.height), float(x), float(y), C.byref(out)), self._ctx)
        return None if out.value < 0 else out.value
