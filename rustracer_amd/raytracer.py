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
        self._scene_obj = None   # strong reference to the object that was uploaded: its id() cannot be recycled while we hold it
        self._scene_sig = None
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
        self._scene_obj = scene
        self._scene_sig = self._signature(scene, lights)
        return flat

    @staticmethod
    def _signature(scene, lights):
        """What makes an uploaded scene stale: a list that grew, shrank or had a shape / material / light replaced. (Mutating
        the fields of a shape in place is not seen: call upload() again, or pass a fresh FlatScene.)"""
        if isinstance(scene, FlatScene):
            return None
        if len(scene) > 4096:
            # a list of a million shape objects: the full signature would cost more host time per frame than the frame's kernel.
            # Cheap form: the length and the ends; pass big scenes as a FlatScene (or Mesh / SphereCloud objects), which skips this
            return (len(scene), id(scene[0]), id(scene[-1]), tuple(id(l) for l in lights))
        return (tuple((id(sh), id(getattr(sh, "material", None))) for sh in scene), tuple(id(l) for l in lights))

    def _sync_scene_camera(self, scene, lights=()):
        # the reference passes the scene on every generate_pixel call (lib.rs:272); here it is re-uploaded when it is another
        # object than the one on the device, or the same list with different members
        if scene is not None and (self._scene_obj is not scene or self._scene_sig != self._signature(scene, lights)):
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

    def render_async(self, scene, samples, rgba_out, slot, max_depth=50, accumulate=False, adaptive=False, flags=0):
        """rthip_render_async: start a pass whose frame lands in `rgba_out` (use host_array()); wait(slot) completes it."""
        self._sync_scene_camera(scene)
        fl = flags | (abi.F_ACCUMULATE if accumulate else 0) | (abi.F_ADAPTIVE if adaptive else 0)
        p = make_params(int(self.width), int(self.height), samples, max_depth, self.seed, fl)
        assert rgba_out.dtype == np.uint8 and rgba_out.flags.c_contiguous and rgba_out.nbytes == int(self.width) * int(self.height) * 4
        _check(self._lib.rthip_render_async(self._ctx, C.byref(p), rgba_out.ctypes.data, int(slot)), self._ctx)

    def wait(self, slot):
        _check(self._lib.rthip_wait(self._ctx, int(slot)), self._ctx)

    def render_device(self, params: abi.RenderParams, d_rgba8=0, d_accum=0, stream=0, want_stats=False):
        """rthip_render_device with raw device pointers (ints), e.g. torch tensors' data_ptr()."""
        self._sync_scene_camera(None)
        st = abi.Stats()
        _check(self._lib.rthip_render_device(self._ctx, C.byref(params), d_rgba8 or None, d_accum or None, stream or None,
                                             C.byref(st) if want_stats else None), self._ctx)
        return st.as_dict() if want_stats else None

    def pixel_status(self):
        """PixelColor.status / PixelCache.same_color_count of every owned pixel after the last adaptive pass (lib.rs:59-73)."""
        rows, width, _ = self.accum_layout()  # the layout of the running sum is the context's, whoever enqueued the last pass
        status = np.empty((rows, width), np.uint8)
        count = np.empty((rows, width), np.uint8)
        _check(self._lib.rthip_pixel_status(self._ctx, status.ctypes.data, count.ctypes.data, status.size), self._ctx)
        return status, count

    def accum_layout(self):
        """(owned rows, width, samples per pixel) of the device-resident running sum (lib.rs:67-73); zeros after a reset."""
        rows, width, nb = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
        _check(self._lib.rthip_accum_layout(self._ctx, C.byref(rows), C.byref(width), C.byref(nb)), self._ctx)
        return rows.value, width.value, nb.value

    def render_device_noalloc(self, params: abi.RenderParams, d_rgba8, d_accum, stream=0, want_stats=False):
        """rthip_render_device_noalloc: the pass for callers that cannot allocate; the running sum lives in d_accum."""
        self._sync_scene_camera(None)
        st = abi.Stats()
        _check(self._lib.rthip_render_device_noalloc(self._ctx, C.byref(params), d_rgba8 or None, d_accum or None, stream or None,
                                                     C.byref(st) if want_stats else None), self._ctx)
        return st.as_dict() if want_stats else None

    def invalidate_pixels(self):
        _check(self._lib.rthip_reset_accum(self._ctx), self._ctx)

    def get_shape(self, scene, x, y) -> Optional[int]:
        """lib.rs:262-267"""
        self._sync_scene_camera(scene)
        out = C.c_int64(-1)
        _check(self._lib.rthip_pick(self._ctx, int(self.width), int(self.height), float(x), float(y), C.byref(out)), self._ctx)
        return None if out.value < 0 else out.value


class GroupRaytracer:
    """One caller, several GPUs of one process (rthip_group_*, include/rthip.h): `render` returns the whole frame, rendered as
    interleaved row bands on every listed device and gathered to the first over RCCL. Mirrors Raytracer's batch `render`."""

    def __init__(self, width, height, seed=0, devices=(0,), peer_copy=False, force_rccl=False):
        self._lib = abi.lib()
        self.width, self.height, self.seed = float(width), float(height), int(seed)
        self._grp = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        flags = (abi.GROUP_F_PEER_COPY if peer_copy else 0) | (abi.GROUP_F_FORCE_RCCL if force_rccl else 0)
        st = self._lib.rthip_group_create(devs, len(devices), flags, C.byref(self._grp))
        if st != 0:
            raise abi.RthipError(st, (self._lib.rthip_group_last_error(None) or b"").decode())
        self.camera = Camera.new(-1.8, 1.0, 2.0)  # lib.rs:252
        self._uploaded_camera = None
        self._flat = None

    def _check(self, st):
        if st != 0:
            raise abi.RthipError(st, (self._lib.rthip_group_last_error(self._grp) or b"").decode())

    def close(self):
        if getattr(self, "_grp", None) and self._grp.value:
            self._lib.rthip_group_destroy(self._grp)
            self._grp = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self._lib.rthip_group_size(self._grp)

    def upload(self, scene, lights=()):
        flat = scene if isinstance(scene, FlatScene) else flatten(scene, lights)
        self._check(self._lib.rthip_group_upload_scene(self._grp, C.byref(flat.desc)))
        self._flat = flat
        return flat

    def invalidate_pixels(self):
        self._check(self._lib.rthip_group_reset_accum(self._grp))

    def pixel_status(self):
        """PixelColor.status / same_color_count of the whole last frame (adaptive passes), rows in image order."""
        w, h = int(self.width), int(self.height)
        status, count = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
        self._check(self._lib.rthip_group_pixel_status(self._grp, status.ctypes.data, count.ctypes.data, status.size))
        return status, count

    def render(self, samples, max_depth=50, accumulate=False, flags=0, band_rows=8, want_accum=True, rgba_out=None, adaptive=False) -> RenderResult:
        flags |= abi.F_ADAPTIVE if adaptive else 0
        cam = self.camera.as_abi()
        key = bytes(cam)
        if key != self._uploaded_camera:
            self._check(self._lib.rthip_group_set_camera(self._grp, C.byref(cam)))
            self._uploaded_camera = key
        w, h = int(self.width), int(self.height)
        p = make_params(w, h, samples, max_depth, self.seed, flags | (abi.F_ACCUMULATE if accumulate else 0), band_rows, 0, 0)
        rgba = rgba_out if rgba_out is not None else np.empty((h, w, 4), np.uint8)
        acc = np.empty((h, w, 3), np.float32) if want_accum else None
        st = abi.Stats()
        self._check(self._lib.rthip_group_render(self._grp, C.byref(p), rgba.ctypes.data, acc.ctypes.data if acc is not None else None, C.byref(st)))
        return RenderResult(rgba, acc, st.as_dict())
