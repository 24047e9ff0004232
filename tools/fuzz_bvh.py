"""Random-geometry sweep of the BVH path against the linear scan on the device (bit for bit): scenes of 97-6000 spheres and
triangles at coordinate scales from 1e-3 to 1e5, far from the origin (the quantised node frames then work at the edge of
FP32 resolution), flat / needle / duplicated geometry, cameras inside the cloud, with and without a point light.
usage: python tools/fuzz_bvh.py [n_scenes] [first_seed]; prints one line per scene; exit status 1 on the first difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Dielectric, Lambertian, Metal, PointLight, Raytracer, Sphere, Triangle, abi, flatten
from rustracer_amd.scene import Camera


def scene(rng):
    n = int(rng.choice([97, 98, 130, 257, 1000, 3000, 6000]))  # 97 spheres: the smallest scene that gets a BVH
    scale = float(10.0 ** rng.uniform(-3, 5))
    offset = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.0, 1.0, 30.0, 1000.0]))
    squash = np.array([1.0, float(rng.choice([1.0, 1e-3, 0.0])), 1.0])  # flat clouds: degenerate boxes on one axis
    mats = [Lambertian.new((0.8, 0.3, 0.3)), Metal.new((0.8, 0.8, 0.8), 0.2), Dielectric.new((1.0, 1.0, 1.0), 1.5), Lambertian.new((0.2, 0.7, 0.4))]
    shapes = []
    tri_share = float(rng.choice([0.0, 0.5, 1.0]))
    dup = rng.random() < 0.2
    for i in range(n):
        c = rng.uniform(-1, 1, 3) * squash * scale + offset
        m = mats[int(rng.integers(0, 4))]
        if rng.random() < tri_share:
            e = float(rng.choice([0.02, 0.2])) * scale
            a, b = rng.uniform(-1, 1, 3) * e, rng.uniform(-1, 1, 3) * e
            if rng.random() < 0.1:
                b = a * 2.0  # needle (zero area)
            shapes.append(Triangle.new(c, c + a, c + b, m))
        else:
            shapes.append(Sphere.new(c, float(rng.choice([0.01, 0.05, 0.3])) * scale * (1.0 if rng.random() > 0.05 else -1.0), m))
        if dup and i % 7 == 0:
            shapes.append(shapes[-1])  # the same object twice: equal t, the later index must win
    eye = offset + rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.3, 2.5]))
    lights = [PointLight(tuple(offset + np.array([0.0, 3.0 * scale, 0.0])), (scale * scale,) * 3)] if rng.random() < 0.3 else []
    return shapes, lights, eye, offset, scale


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for seed in range(first, first + n_scenes):
        rng = np.random.default_rng(seed)
        shapes, lights, eye, lookat, scale = scene(rng)
        flat = flatten(shapes, lights)
        w, h = int(rng.choice([64, 97])), int(rng.choice([36, 55]))
        spp, depth = int(rng.choice([1, 3, 6])), int(rng.choice([1, 4, 12]))
        rt = Raytracer.new(w, h, seed=seed)
        rt.camera = Camera.new_lookat(eye[0], eye[1], eye[2], tuple(lookat))
        fl = abi.F_COUNT_STATS | (abi.F_POINT_LIGHTS if lights else 0)
        a = rt.render(flat, spp, max_depth=depth, flags=fl)
        b = rt.render(flat, spp, max_depth=depth, flags=fl | abi.F_NO_BVH)
        same = np.array_equal(a.accum.view(np.uint32), b.accum.view(np.uint32)) and np.array_equal(a.rgba8, b.rgba8) and a.stats["rays"] == b.stats["rays"]
        hit = sum(a.stats["hits"]) / max(1, a.stats["rays"])
        print(f"seed {seed}: {len(flat.spheres)} spheres {len(flat.triangles)} triangles scale {scale:.1e} {w}x{h}@{spp} d{depth} lights {len(lights)}: "
              f"BVH == scan {same}; node tests/ray {a.stats['node_tests'] / max(1, a.stats['rays']):.1f}; rays that hit a shape {hit:.2f}", flush=True)
        rt.close()
        if not same or a.stats["node_tests"] == 0:
            diff = np.argwhere((a.accum.view(np.uint32) != b.accum.view(np.uint32)).any(axis=2))
            print("   first differing pixels:", diff[:5].tolist())
            sys.exit(1)
    print("all equal")


if __name__ == "__main__":
    main()
