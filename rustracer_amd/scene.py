"""Host-side mirror of raytracer_core's scene API (shapes, materials, camera).

Same names, argument meaning and error behaviour as the reference constructors, so the parity
tests read like code written against the reference:

    Sphere::new(center, radius, material)            raytracer_core/src/shapes/sphere.rs:19
    Lambertian::new / new_from_hex                   raytracer_core/src/materials/lambertian_diffuse.rs:19,26
    Metal::new(albedo, fuzziness)                    raytracer_core/src/materials/metal.rs:19
    Dielectric::new(albedo, refraction_idx)          raytracer_core/src/materials/dielectric.rs:20
    Texture::load_from_file(path, scale)             raytracer_core/src/materials/texture.rs:21
    Camera::{new,new_lookat,emit_ray_at,rotate,move_camera}   raytracer_core/src/camera.rs:16-78
    Scene = Vec<Box<dyn Shape>>                      raytracer_core/src/lib.rs:75

`flatten(scene)` turns the list of shapes into the POD description of include/rthip.h
(struct-of-arrays per shape kind, materials de-duplicated by identity) — the reference keeps all
geometry fields private, so this description is what both the device path and the CPU
oracle consume. Plane / Triangle / PointLight are extensions without a reference counterpart.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import zlib

import numpy as np

from . import abi

SPHERE_DT = np.dtype([("center", "f8", 3), ("radius", "f8"), ("material", "u4"), ("_pad", "u4")])
PLANE_DT = np.dtype([("point", "f8", 3), ("normal", "f8", 3), ("material", "u4"), ("_pad", "u4")])
TRIANGLE_DT = np.dtype([("v0", "f8", 3), ("v1", "f8", 3), ("v2", "f8", 3), ("material", "u4"), ("_pad", "u4")])
MATERIAL_DT = np.dtype([("kind", "u4"), ("texture", "u4"), ("albedo", "f8", 3), ("param", "f8"), ("shininess", "f8")])
LIGHT_DT = np.dtype([("position", "f8", 3), ("intensity", "f8", 3)])
assert SPHERE_DT.itemsize == C.sizeof(abi.Sphere) and PLANE_DT.itemsize == C.sizeof(abi.Plane)
assert TRIANGLE_DT.itemsize == C.sizeof(abi.Triangle) and MATERIAL_DT.itemsize == C.sizeof(abi.Material)
assert LIGHT_DT.itemsize == C.sizeof(abi.Light)


def Vector3(x, y, z):
    """nalgebra::Vector3<f64>::new"""
    return (float(x), float(y), float(z))


# ------------------------------------------------------------------------------- materials
class Material:
    kind = -1

    def _record(self, texture_index):
        raise NotImplementedError


class Lambertian(Material):
    kind = abi.MAT_LAMBERTIAN

    def __init__(self, albedo):
        self.albedo = Vector3(*albedo)

    @classmethod
    def new(cls, albedo):
        return cls(albedo)

    @classmethod
    def new_from_hex(cls, color: int):
        # lambertian_diffuse.rs:26-35
        return cls((((color & 0xFF0000) >> 16) / 255.0, ((color & 0x00FF00) >> 8) / 255.0, (color & 0x0000FF) / 255.0))

    def _record(self, texture_index):
        return (self.kind, 0, self.albedo, 0.0, 0.0)


class Metal(Material):
    kind = abi.MAT_METAL

    def __init__(self, albedo, fuzziness, shininess=0.0):
        self.albedo = Vector3(*albedo)
        self.fuzziness = float(fuzziness)
        self.shininess = float(shininess)  # extension: Phong exponent of the point-light mode

    @classmethod
    def new(cls, albedo, fuzziness):
        return cls(albedo, fuzziness)

    def _record(self, texture_index):
        return (self.kind, 0, self.albedo, self.fuzziness, self.shininess)


class Dielectric(Material):
    kind = abi.MAT_DIELECTRIC

    def __init__(self, albedo, refraction_idx):
        self.albedo = Vector3(*albedo)
        self.refraction_idx = float(refraction_idx)

    @classmethod
    def new(cls, albedo, refraction_idx):
        return cls(albedo, refraction_idx)

    def _record(self, texture_index):
        return (self.kind, 0, self.albedo, self.refraction_idx, 0.0)


class Texture(Material):
    """Texture material: nearest-texel lookup with wrap, terminal (texture.rs:32-61)."""
    kind = abi.MAT_TEXTURE

    def __init__(self, rgb8: np.ndarray, scale: float):
        rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
        if rgb8.ndim != 3 or rgb8.shape[2] != 3:
            raise ValueError("texture must be an (H, W, 3) uint8 array")
        self.rgb8 = rgb8
        self.scale = float(scale)

    @property
    def width(self):
        return self.rgb8.shape[1]

    @property
    def height(self):
        return self.rgb8.shape[0]

    @classmethod
    def load_from_file(cls, path, scale):
        """Texture::load_from_file (texture.rs:21-30). Accepts the build's decoded-texel fixture
        format (``*.rgb8.z``: 8-byte little-endian (width, height) + zlib(RGB8)) and, when PIL is
        importable, anything PIL opens; raises OSError like ImageResult::Err on failure."""
        path = os.fspath(path)
        if path.endswith(".rgb8.z"):
            with open(path, "rb") as f:
                blob = f.read()
            w, h = np.frombuffer(blob[:8], dtype="<u4")
            rgb = np.frombuffer(zlib.decompress(blob[8:]), dtype=np.uint8).reshape(int(h), int(w), 3)
            return cls(rgb, scale)
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise OSError(f"cannot decode {path}: PIL is not available") from e
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))  # get_pixel().to_rgb(): alpha dropped (texture.rs:50)
        return cls(rgb, scale)

    def _record(self, texture_index):
        return (self.kind, texture_index, (0.0, 0.0, 0.0), self.scale, 0.0)


# ---------------------------------------------------------------------------------- shapes
class Shape:
    material: Material


class Sphere(Shape):
    def __init__(self, center, radius, material: Material):
        self.center = Vector3(*center)
        self.radius = float(radius)
        self.material = material

    @classmethod
    def new(cls, center, radius, material):
        return cls(center, radius, material)


class Plane(Shape):
    """extension (no reference counterpart)"""

    def __init__(self, point, normal, material: Material):
        self.point = Vector3(*point)
        self.normal = Vector3(*normal)
        self.material = material

    @classmethod
    def new(cls, point, normal, material):
        return cls(point, normal, material)


class Triangle(Shape):
    """extension (no reference counterpart)"""

    def __init__(self, v0, v1, v2, material: Material):
        self.v0, self.v1, self.v2 = Vector3(*v0), Vector3(*v1), Vector3(*v2)
        self.material = material

    @classmethod
    def new(cls, v0, v1, v2, material):
        return cls(v0, v1, v2, material)


class PointLight:
    """extension (no reference counterpart): used only with F_POINT_LIGHTS"""

    def __init__(self, position, intensity):
        self.position = Vector3(*position)
        self.intensity = Vector3(*intensity)


class Mesh(Shape):
    """extension: a bulk triangle soup sharing one material, kept as an (n, 3, 3) float64 array so
    million-triangle scenes do not need a Python object per triangle."""

    def __init__(self, vertices: np.ndarray, material: Material):
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        if v.ndim != 3 or v.shape[1:] != (3, 3):
            raise ValueError("mesh must be an (n, 3, 3) array of triangle vertices")
        self.vertices = v
        self.material = material


class SphereCloud(Shape):
    """extension: bulk spheres, `centers` (n,3), `radii` (n,), `materials` list + per-sphere index."""

    def __init__(self, centers, radii, materials, material_index):
        self.centers = np.ascontiguousarray(centers, dtype=np.float64).reshape(-1, 3)
        self.radii = np.ascontiguousarray(radii, dtype=np.float64).reshape(-1)
        self.materials = list(materials)
        self.material_index = np.ascontiguousarray(material_index, dtype=np.int64).reshape(-1)
        if not (len(self.centers) == len(self.radii) == len(self.material_index)):
            raise ValueError("SphereCloud arrays disagree in length")
        if len(self.material_index) and (self.material_index.min() < 0 or self.material_index.max() >= len(self.materials)):
            raise ValueError("SphereCloud material index out of range")


class FlatScene:
    """The POD scene description of include/rthip.h plus the numpy arrays that own its memory.

    ``order[i]`` gives, for the i-th shape of the input list, its global shape index in the
    flattened scene (spheres first, then planes, then triangles — the tie-break order)."""

    def __init__(self, spheres, planes, triangles, materials, textures, lights, order):
        self.spheres, self.planes, self.triangles = spheres, planes, triangles
        self.materials, self.textures, self.lights = materials, textures, lights
        self.order = order
        self._tex_structs = (abi.Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            self._tex_structs[i].width = t.width
            self._tex_structs[i].height = t.height
            self._tex_structs[i].rgb8 = t.rgb8.ctypes.data
        d = abi.SceneDesc()
        d.spheres, d.n_spheres = (spheres.ctypes.data if len(spheres) else None), len(spheres)
        d.planes, d.n_planes = (planes.ctypes.data if len(planes) else None), len(planes)
        d.triangles, d.n_triangles = (triangles.ctypes.data if len(triangles) else None), len(triangles)
        d.materials, d.n_materials = (materials.ctypes.data if len(materials) else None), len(materials)
        d.textures, d.n_textures = (C.addressof(self._tex_structs) if textures else None), len(textures)
        d.lights, d.n_lights = (lights.ctypes.data if len(lights) else None), len(lights)
        self.desc = d

    @property
    def n_shapes(self):
        return len(self.spheres) + len(self.planes) + len(self.triangles)


def flatten(scene, lights=()) -> FlatScene:
    """Scene (list of shapes) -> FlatScene. Shape order inside each kind is preserved."""
    mat_index, mat_records, textures, tex_index = {}, [], [], {}

    def mat_id(m: Material) -> int:
        if not isinstance(m, Material):
            raise TypeError("shape material must be a Material")
        k = id(m)
        if k not in mat_index:
            ti = 0
            if isinstance(m, Texture):
                if id(m) not in tex_index:
                    tex_index[id(m)] = len(textures)
                    textures.append(m)
                ti = tex_index[id(m)]
            mat_index[k] = len(mat_records)
            mat_records.append(m._record(ti))
        return mat_index[k]

    sph_parts, pln, tri_parts = [], [], []
    counts = []  # (kind, count) per input shape, to compute `order`
    for sh in scene:
        if isinstance(sh, Sphere):
            a = np.zeros(1, SPHERE_DT)
            a["center"], a["radius"], a["material"] = sh.center, sh.radius, mat_id(sh.material)
            sph_parts.append(a)
            counts.append((0, 1))
        elif isinstance(sh, SphereCloud):
            ids = np.array([mat_id(m) for m in sh.materials], dtype=np.uint32)
            a = np.zeros(len(sh.radii), SPHERE_DT)
            a["center"], a["radius"] = sh.centers, sh.radii
            a["material"] = ids[sh.material_index] if len(sh.radii) else 0
            sph_parts.append(a)
            counts.append((0, len(a)))
        elif isinstance(sh, Plane):
            a = np.zeros(1, PLANE_DT)
            a["point"], a["normal"], a["material"] = sh.point, sh.normal, mat_id(sh.material)
            pln.append(a)
            counts.append((1, 1))
        elif isinstance(sh, Triangle):
            a = np.zeros(1, TRIANGLE_DT)
            a["v0"], a["v1"], a["v2"], a["material"] = sh.v0, sh.v1, sh.v2, mat_id(sh.material)
            tri_parts.append(a)
            counts.append((2, 1))
        elif isinstance(sh, Mesh):
            a = np.zeros(len(sh.vertices), TRIANGLE_DT)
            a["v0"], a["v1"], a["v2"] = sh.vertices[:, 0], sh.vertices[:, 1], sh.vertices[:, 2]
            a["material"] = mat_id(sh.material)
            tri_parts.append(a)
            counts.append((2, len(a)))
        else:
            raise TypeError(f"not a shape: {sh!r}")
    spheres = np.concatenate(sph_parts) if sph_parts else np.zeros(0, SPHERE_DT)
    planes = np.concatenate(pln) if pln else np.zeros(0, PLANE_DT)
    triangles = np.concatenate(tri_parts) if tri_parts else np.zeros(0, TRIANGLE_DT)
    materials = np.zeros(len(mat_records), MATERIAL_DT)
    for i, (kind, tex, albedo, param, shin) in enumerate(mat_records):
        materials[i] = (kind, tex, albedo, param, shin)
    lt = np.zeros(len(lights), LIGHT_DT)
    for i, l in enumerate(lights):
        lt[i] = (l.position, l.intensity)
    base = [0, len(spheres), len(spheres) + len(planes)]
    nxt = list(base)
    order = []
    for kind, n in counts:
        order.append(nxt[kind])
        nxt[kind] += n
    return FlatScene(spheres, planes, triangles, materials, textures, lt, order)


# ---------------------------------------------------------------------------------- camera
def _v(a):
    return np.asarray(a, dtype=np.float64)


class Camera:
    """raytracer_core/src/camera.rs. `origin` is public there (camera.rs:8); the other fields are
    private, so the device constants are recovered through emit_ray_at exactly as SURVEY §8b says."""

    def __init__(self, origin, lookat, lower_left_corner, horizontal, vertical):
        self.origin = _v(origin)
        self._lookat = _v(lookat)
        self._lower_left_corner = _v(lower_left_corner)
        self._horizontal = _v(horizontal)
        self._vertical = _v(vertical)

    @classmethod
    def new(cls, x, y, z):
        return cls.new_lookat(x, y, z, Vector3(0.0, 0.0, -1.0))  # camera.rs:16-18

    @classmethod
    def new_lookat(cls, x, y, z, lookat):
        la = (C.c_double * 3)(*[float(c) for c in lookat])
        out = abi.Camera()
        st = abi.lib().rthip_camera_lookat(float(x), float(y), float(z), la, C.byref(out))  # camera.rs:19-41 on the host
        if st != 0:
            raise abi.RthipError(st, "rthip_camera_lookat")
        o, d00 = _v(list(out.origin)), _v(list(out.d00))
        return cls(o, lookat, o + d00, list(out.horizontal), list(out.vertical))

    def emit_ray_at(self, offset_x, offset_y):
        """camera.rs:43-50 -> (origin, direction)"""
        d = self._lower_left_corner - self.origin + self._horizontal * offset_x + self._vertical * offset_y
        return self.origin.copy(), d

    def as_abi(self) -> abi.Camera:
        d00 = self.emit_ray_at(0.0, 0.0)[1]
        h = self.emit_ray_at(1.0, 0.0)[1] - d00
        v = self.emit_ray_at(0.0, 1.0)[1] - d00
        c = abi.Camera()
        c.origin[:], c.d00[:], c.horizontal[:], c.vertical[:] = list(self.origin), list(d00), list(h), list(v)
        return c

    # --- UI helpers (host-side only, off the hot path): camera.rs:52-78 ---
    def rotate(self, rotation):
        off = self._lookat - self.origin
        r = _euler_rotation(*rotation)  # Rotation3::from_euler_angles(roll, pitch, yaw)
        off = r @ off
        return Camera.new_lookat(*self.origin, tuple(self.origin + off))

    def move_camera(self, direction):
        off = self._lookat - self.origin
        off[1] = 0.0
        r = _face_towards(off, _v((0.0, 1.0, 0.0)))
        real = r @ _v(direction)
        return Camera.new_lookat(*(self.origin + real), tuple(self._lookat + real))


def _euler_rotation(roll, pitch, yaw):
    # nalgebra Rotation3::from_euler_angles: R = Rz(yaw) * Ry(pitch) * Rx(roll)
    sr, cr, sp, cp, sy, cy = math.sin(roll), math.cos(roll), math.sin(pitch), math.cos(pitch), math.sin(yaw), math.cos(yaw)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _face_towards(direction, up):
    # nalgebra Rotation3::face_towards: columns (x, y, z) with z = normalize(dir), x = normalize(up x z), y = z x x
    z = direction / np.linalg.norm(direction)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)
