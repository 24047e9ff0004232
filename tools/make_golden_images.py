#!/usr/bin/env python3
"""Generate the golden images under tests/golden/images/ from the CPU oracle in f64 (reference-faithful)
mode: small renders of every scene preset, so the GPU box can parity-check against committed vectors.
These pin the BUILD'S spec (oracle + counter RNG); the reference itself holds no golden images
 and its renders are entropy-seeded (F4)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from rustracer_amd import abi, flatten, make_params, presets  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "images")

# name -> (scene factory, lights factory, width, height, spp, max_depth, seed, flags, camera pose)
CASES = {
    "standalone_160x90_s4_d50": (presets.standalone_scene, None, 160, 90, 4, 50, 1, 0, (-1.8, 1.0, 2.0)),
    "standalone_160x90_s4_d5": (presets.standalone_scene, None, 160, 90, 4, 5, 1, 0, (-1.8, 1.0, 2.0)),
    "standalone_200x150_s2_d50": (presets.standalone_scene, None, 200, 150, 2, 50, 7, 0, (-1.8, 1.0, 2.0)),  # 4:3, C1's aspect
    "wasm_192x108_s8_d50": (presets.wasm_scene, None, 192, 108, 8, 50, 0, 0, (-1.8, 1.0, 2.0)),
    "ggez_160x90_s4_d50": (presets.ggez_scene, None, 160, 90, 4, 50, 2, 0, (-1.8, 1.0, 2.0)),
    "textured_160x90_s4_d50": (presets.textured_scene, None, 160, 90, 4, 50, 3, 0, (-1.8, 1.0, 2.0)),
    "plane_160x90_s4_d8": (presets.plane_scene, None, 160, 90, 4, 8, 4, 0, (-1.8, 1.0, 2.0)),
    "lights_160x90_s4_d5": (presets.standalone_scene, presets.standalone_lights, 160, 90, 4, 5, 5, abi.F_POINT_LIGHTS, (-1.8, 1.0, 2.0)),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (scene_f, lights_f, w, h, spp, depth, seed, flags, pose) in CASES.items():
        flat = flatten(scene_f(), lights_f() if lights_f else ())
        cam = oracle.camera_lookat(*pose)
        p = make_params(w, h, spp, depth, seed, flags)
        rgba, acc, st, _ = oracle.render(flat, cam, p, precision=64)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), rgba8=rgba, accum=acc.astype(np.float32),
                            rays=np.uint64(st["rays"]), paths=np.uint64(st["paths"]))
        print(name, rgba.shape, st["rays"])


if __name__ == "__main__":
    main()
