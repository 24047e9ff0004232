"""Scene presets: the reference's own scenes plus the synthetic benchmark configurations of
SURVEY.md §8d (C1-C5). Every random parameter comes from numpy's PCG64 with a stated seed, so the
device path and the oracle see identical scenes.
"""
from __future__ import annotations

import math
import os

import numpy as np

from .scene import (Dielectric, Lambertian, Mesh, Metal, Plane, PointLight, Sphere, SphereCloud, Texture, Triangle, Vector3)

_HERE = os.path.dirname(os.path.abspath(__file__))
TEXTURE_DIR = os.path.join(_HERE, "..", "tests", "golden", "textures")


def texture_path(name):
    return os.path.join(TEXTURE_DIR, name + ".rgb8.z")


def standalone_scene():
    """standalone/src/main.rs:66-89 — shape indices: 0 dielectric, 1 ground, 2 metal, 3 texture."""
    return [
        Sphere.new(Vector3(-1.01, 0.0, -1.0), 0.5, Dielectric.new(Vector3(1.0, 0.8, 0.80), 1.05)),
        Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070)),
        Sphere.new(Vector3(1.0, 0.0, -1.0), 0.5, Metal.new(Vector3(0.8, 0.8, 0.8), 0.1)),
        Sphere.new(Vector3(-0.0, 0.0, -1.0), 0.5, Texture.load_from_file(texture_path("bergsjostolen"), 1.0)),
    ]


def standalone_lights():
    """extension: the single point light of config C2's optional direct-light line."""
    return [PointLight((5.0, 5.0, 5.0), (1.0, 1.0, 1.0))]


def wasm_scene():
    """wasm/src/lib.rs:90-105 — three fuzz-0 metal spheres."""
    m = lambda: Metal.new(Vector3(0.8, 0.8, 0.8), 0.0)  # noqa: E731
    return [
        Sphere.new(Vector3(0.0, 0.0, -1.0), 0.5, m()),
        Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, m()),
        Sphere.new(Vector3(0.5, -0.4, -0.85), 0.1, m()),
    ]


def ggez_scene(seed=0, target_shape_index=7):
    """game_ggez/src/main.rs:113-153 + poisson.rs:10-30: ground + up to 39 dielectric spheres placed by
    the reference's Poisson-disc recipe (the `random.next_u64()/u64::MAX` seed per attempt comes
    from numpy here — the reference's is entropy-seeded)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = [Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070))]
    positions = [(0.0, 5.0), (0.0, 0.0)]
    nb_new_shapes, radius, nb_attempts = 40, 3.0, 10
    odd = Dielectric.new(Vector3(1.0, 0.6, 0.60), 1.05)
    common = Dielectric.new(Vector3(0.0, 0.6, 1.0), 1.5)
    index = 0
    while index < len(positions) and len(shapes) < nb_new_shapes:
        ref = positions[index]
        s = float(rng.random())
        rpe = radius + 0.01
        found = None
        for attempt in range(nb_attempts):
            theta = 2.0 * math.pi * (s + attempt / nb_attempts)
            tp = (ref[0] + rpe * math.cos(theta), ref[1] + rpe * math.sin(theta))
            if all(math.sqrt((e[0] - tp[0]) ** 2 + (e[1] - tp[1]) ** 2) >= rpe for e in positions):
                found = tp
                break
        if found is not None:
            mat = odd if len(shapes) == target_shape_index else common
            shapes.append(Sphere.new(Vector3(found[0], 0.0, found[1]), 0.5, mat))
            positions.append(found)
        else:
            index += 1
    return shapes


def textured_scene():
    """C3: ground sphere + 3x3 grid of r=.5 spheres cycling the three reference textures,
    plus a large textured quad (two triangles, extension) at z = -4."""
    earth = Texture.load_from_file(texture_path("earth"), 1.0)
    chess = Texture.load_from_file(texture_path("chessboard"), 4.0)
    berg = Texture.load_from_file(texture_path("bergsjostolen"), 1.0)
    mats = [earth, chess, berg]
    shapes = [Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070))]
    k = 0
    for gz in (-1.2, 0.0, 1.2):
        for gx in (-1.2, 0.0, 1.2):
            shapes.append(Sphere.new(Vector3(gx, 0.0, gz - 1.0), 0.5, mats[k % 3]))
            k += 1
    a, b, c, d = (-4.0, -0.5, -4.0), (4.0, -0.5, -4.0), (4.0, 3.5, -4.0), (-4.0, 3.5, -4.0)
    shapes.append(Triangle.new(a, b, d, berg))
    shapes.append(Triangle.new(c, d, b, berg))
    return shapes


# ------------------------------------------------------------------------------------------
def icosphere(subdivisions):
    """unit icosphere: 20 * 4^subdivisions triangles as an (n, 3, 3) array"""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                  (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)])
    tri = v[f]  # (20, 3, 3)
    for _ in range(subdivisions):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return tri


def uv_sphere(n_lon=250, n_lat=201):
    """unit sphere as a latitude / longitude mesh, closed: 2 * n_lon * (n_lat - 1) triangles (the default: exactly 100 000 — the
    "100 k-tri mesh" of SURVEY §8d's C5; an icosphere only offers 20 * 4^k), as an (n, 3, 3) array. A quad of the grid gives two
    triangles, except in the two polar bands, where one of them has collapsed into the pole and is left out."""
    lat = np.linspace(0.0, math.pi, n_lat + 1)
    lon = np.linspace(0.0, 2.0 * math.pi, n_lon + 1)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    pts = np.stack([np.sin(la) * np.cos(lo), np.cos(la), np.sin(la) * np.sin(lo)], -1)  # (n_lat + 1, n_lon + 1, 3)
    a, b, c, d = pts[:-1, :-1], pts[1:, :-1], pts[1:, 1:], pts[:-1, 1:]
    rows = slice(0, n_lat - 1)  # (a, b, c) of every band but the southern polar one, (a, c, d) of every band but the northern
    tri = np.concatenate([np.stack([a[rows], b[rows], c[rows]], -2).reshape(-1, 3, 3), np.stack([a[1:], c[1:], d[1:]], -2).reshape(-1, 3, 3)])
    return tri


def _poisson_positions(n, radius, seed):
    """the placement recipe of game_ggez/src/poisson.rs:10-30 with a fixed seed"""
    rng = np.random.Generator(np.random.PCG64(seed))
    positions = [(0.0, 5.0), (0.0, 0.0)]
    out, index, rpe = [], 0, radius + 0.01
    while index < len(positions) and len(out) < n:
        ref = positions[index]
        s = float(rng.random())
        found = None
        for attempt in range(10):
            theta = 2.0 * math.pi * (s + attempt / 10)
            tp = (ref[0] + rpe * math.cos(theta), ref[1] + rpe * math.sin(theta))
            pts = np.array(positions)
            if np.all(np.hypot(pts[:, 0] - tp[0], pts[:, 1] - tp[1]) >= rpe):
                found = tp
                break
        if found is not None:
            out.append(found)
            positions.append(found)
        else:
            index += 1
    return out


def mesh_scene(n_objects=49, subdivisions=5, seed=4):
    """C4: n_objects icospheres (20*4^subdivisions triangles each; 49 x 20480 = 1 003 520) at
    Poisson-disc positions over the ground sphere, Lambertian / Metal alternating."""
    ico = icosphere(subdivisions) * 0.5
    shapes = [Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070))]
    lam = Lambertian.new(Vector3(0.8, 0.3, 0.3))
    met = Metal.new(Vector3(0.8, 0.8, 0.8), 0.05)
    for i, (px, pz) in enumerate(_poisson_positions(n_objects, 1.2, seed)):
        shapes.append(Mesh(ico + np.array([px * 0.6, 0.0, pz * 0.6 - 3.0]), lam if i % 2 == 0 else met))
    return shapes


def big_scene(n_side=100, seed=7, mesh_subdivisions=None):
    """C5: n_side^2 spheres r in U[.05,.25] on a jittered grid, 50 % Lambertian / 25 % Metal
    (fuzz U[0,.3]) / 25 % Dielectric (ior 1.5), over the ground sphere, plus an optional icosphere mesh."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_side * n_side
    gx, gz = np.meshgrid(np.arange(n_side), np.arange(n_side))
    spacing = 0.6
    cx = (gx.reshape(-1) - n_side / 2 + rng.random(n) * 0.6) * spacing
    cz = (gz.reshape(-1) - n_side / 2 + rng.random(n) * 0.6) * spacing - 1.0
    r = 0.05 + 0.2 * rng.random(n)
    # sit the spheres on the ground sphere (centre (0,-100.5,-1), radius 100)
    cy = np.sqrt(np.maximum((100.0 + r) ** 2 - cx ** 2 - (cz + 1.0) ** 2, 0.0)) - 100.5
    kind = rng.random(n)
    mats, idx = [], np.zeros(n, np.int64)
    palette = rng.random((64, 3)) * 0.8 + 0.1
    fuzz = rng.random(64) * 0.3
    for j in range(64):
        mats.append(Lambertian.new(Vector3(*palette[j])))
    for j in range(64):
        mats.append(Metal.new(Vector3(*palette[j]), fuzz[j]))
    mats.append(Dielectric.new(Vector3(1.0, 1.0, 1.0), 1.5))
    pick = rng.integers(0, 64, n)
    idx[:] = np.where(kind < 0.5, pick, np.where(kind < 0.75, 64 + pick, 128))
    shapes = [Sphere.new(Vector3(0.0, -100.5, -1.0), 100.0, Lambertian.new_from_hex(0x007070)),
              SphereCloud(np.stack([cx, cy, cz], 1), r, mats, idx)]
    if mesh_subdivisions == "uv100k":  # SURVEY §8d as written: one 100 000-triangle mesh
        shapes.append(Mesh(uv_sphere() * 1.5 + np.array([0.0, 1.5, -6.0]), Metal.new(Vector3(0.9, 0.9, 0.9), 0.0)))
    elif mesh_subdivisions is not None:
        ico = icosphere(mesh_subdivisions) * 1.5 + np.array([0.0, 1.5, -6.0])
        shapes.append(Mesh(ico, Metal.new(Vector3(0.9, 0.9, 0.9), 0.0)))
    return shapes


def plane_scene():
    """small extension test scene: a real ground plane + the three standalone spheres + a triangle"""
    sc = standalone_scene()
    sc[1] = Plane.new((0.0, -0.5, 0.0), (0.0, 1.0, 0.0), Lambertian.new_from_hex(0x007070))
    sc.append(Triangle.new((-2.0, -0.5, -2.5), (2.0, -0.5, -2.5), (0.0, 1.5, -2.5), Metal.new(Vector3(0.9, 0.7, 0.3), 0.0)))
    return sc
