import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, oracle
from fuzz_parity import random_scene
from rustracer_amd import Raytracer, abi, flatten, make_params
from rustracer_amd.scene import Camera
for sd in [int(a) for a in sys.argv[1:]] or (23, 45, 29):
    rng = np.random.default_rng(sd)
    shapes, lights, pose, flavour = random_scene(rng)
    w, h = int(rng.choice([33, 64, 97, 160])), int(rng.choice([17, 36, 55, 90]))
    spp, depth = int(rng.choice([1, 3, 8, 70])), int(rng.choice([1, 2, 5, 50]))
    flags = (abi.F_POINT_LIGHTS if lights else 0) | (abi.F_NO_BVH if rng.random() < 0.25 else 0)
    flat = flatten(shapes, lights)
    rt = Raytracer.new(w, h, seed=sd); rt.camera = Camera.new(*pose)
    res = rt.render(flat, spp, max_depth=depth, flags=flags)
    cam = oracle.camera_lookat(*pose)
    p = make_params(w, h, spp, depth, sd, flags)
    ref64, a64, _, _ = oracle.render(flat, cam, p, precision=64)
    ref32, a32, _, _ = oracle.render(flat, cam, p, precision=32)
    bad = np.argwhere(np.abs(res.rgba8[..., :3].astype(int) - ref64[..., :3].astype(int)).max(axis=2) > 1)
    bad32 = np.abs(ref32[..., :3].astype(int) - ref64[..., :3].astype(int)).max(axis=2) > 1
    print(f"seed {sd}: {len(bad)} bad pixels of {w*h}; oracle f32 vs f64 bad: {int(bad32.sum())}; kinds {[m for m in flat.materials['kind']][:6]} n_planes {len(flat.planes)}")
    for (y, x) in bad[:6]:
        shape = rt.get_shape(flat, x + 0.5, y + 0.5)
        mk = None
        print(f"   pixel ({x},{y}) primary hit shape {shape} dev {res.rgba8[y, x, :3]} f64 {ref64[y, x, :3]} f32 {ref32[y, x, :3]}")
    rt.close()
