"""The oracle's optional accelerator (oracle/rt_oracle.hpp, build_accel) must return exactly what the oracle's linear scan —
the restatement of ray.rs:47-67 — returns: same image bits, same sums, same ray counts, on scenes with spheres, triangles,
planes, duplicated shapes (the "later index wins on equal t" rule), point lights, in f64 and f32. CPU only."""
import numpy as np
import pytest

import oracle
from rustracer_amd import Dielectric, Lambertian, Metal, Plane, PointLight, Sphere, Triangle, abi, flatten, make_params, presets


def _random_scene(rng, n_spheres, n_tris, planes, duplicates):
    mats = [Lambertian.new(rng.uniform(0.2, 0.9, 3)), Metal.new(rng.uniform(0.3, 0.9, 3), 0.2), Dielectric.new((1.0, 0.9, 0.9), 1.4)]
    shapes = [Sphere.new((0.0, -100.5, -1.0), 100.0, mats[0])]
    for _ in range(n_spheres):
        shapes.append(Sphere.new(rng.uniform(-3, 3, 3) * (1, 0.4, 1) + (0, 0.2, -2), float(rng.choice([0.05, 0.2, 0.6])), mats[int(rng.integers(0, 3))]))
    for _ in range(n_tris):
        a = rng.uniform(-3, 3, 3) * (1, 0.4, 1) + (0, 0.3, -2)
        shapes.append(Triangle.new(a, a + rng.uniform(-0.8, 0.8, 3), a + rng.uniform(-0.8, 0.8, 3), mats[int(rng.integers(0, 2))]))
    if planes:
        shapes.append(Plane.new((0.0, -0.45, 0.0), (0.0, 1.0, 0.0), mats[1]))
    if duplicates:  # exact copies: equal t on every ray that hits them, the later index must win
        shapes += [shapes[3], shapes[len(shapes) // 2], shapes[3]]
    return shapes


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("seed,n_spheres,n_tris,planes,duplicates,lights", [
    (1, 6, 0, False, False, False), (2, 40, 60, False, True, False), (3, 150, 200, True, True, True), (4, 0, 300, True, False, True)])
def test_accelerator_equals_scan(precision, seed, n_spheres, n_tris, planes, duplicates, lights):
    rng = np.random.default_rng(seed)
    flat = flatten(_random_scene(rng, n_spheres, n_tris, planes, duplicates), [PointLight((3.0, 4.0, 2.0), (20.0, 20.0, 20.0))] if lights else ())
    cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
    p = make_params(96, 54, 3, 6, seed, abi.F_POINT_LIGHTS if lights else 0)
    a_img, a_sum, a_st, _ = oracle.render(flat, cam, p, precision=precision)
    b_img, b_sum, b_st, _ = oracle.render(flat, cam, p, precision=precision, accel=True)
    assert np.array_equal(a_img, b_img)
    assert np.array_equal(a_sum.view(np.uint64), b_sum.view(np.uint64))
    for k in ("paths", "rays", "shadow_rays", "hits", "sky_evals"):
        assert a_st[k] == b_st[k], k
    if n_spheres + n_tris > 50:  # ... and it is an accelerator: far fewer tests than the scan's N per ray
        assert b_st["sphere_tests"] + b_st["tri_tests"] < (a_st["sphere_tests"] + a_st["tri_tests"]) / 4


def test_accelerator_on_the_reference_scenes_and_crops():
    cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
    for scene in (presets.standalone_scene(), presets.wasm_scene(), presets.ggez_scene(), presets.textured_scene()):
        flat = flatten(scene)
        p = make_params(160, 90, 2, 50, 9)
        for crop in (None, (40, 20, 64, 48)):
            a = oracle.render(flat, cam, p, crop=crop)
            b = oracle.render(flat, cam, p, crop=crop, accel=True)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)) and a[2]["rays"] == b[2]["rays"]
