"""A/B of library variants on the linear-scan paths: ggez scene (40 spheres, LDS scan), C2 + point light (shadow scan),
401 spheres forced to the global linear scan. usage: RTHIP_LIB=... python tools/ab_nobvh.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, abi, flatten, presets
cases = [("ggez 40 spheres 4K@16", flatten(presets.ggez_scene()), 3840, 2160, 16, 50, 0),
         ("C2 + point light 4K@64", flatten(presets.standalone_scene(), presets.standalone_lights()), 3840, 2160, 64, 5, abi.F_POINT_LIGHTS),
         ("401 spheres + 320 tris NO_BVH 1080p@4", flatten(presets.big_scene(20, 3, 2)), 1920, 1080, 4, 8, abi.F_NO_BVH)]
for name, flat, w, h, spp, depth, fl in cases:
    rt = Raytracer.new(w, h, seed=1)
    rt.render(flat, spp, max_depth=depth, want_accum=False, flags=fl)
    best = min(rt.render(flat, spp, max_depth=depth, want_accum=False, flags=fl).stats for _ in range(3))["kernel_ms"] if False else min(rt.render(flat, spp, max_depth=depth, want_accum=False, flags=fl).stats["kernel_ms"] for _ in range(3))
    rays = rt.render(flat, spp, max_depth=depth, want_accum=False, flags=fl).stats["rays"]
    print(f"{name}: {best:.3f} ms, {rays / best / 1e3:.0f} Mrays/s")
    rt.close()
