"""Where does the BVH kernel overtake the LDS sphere scan? Times scenes of n spheres with both (RTHIP_LIB selects a library built
with -DRT_BVH_THRESHOLD=8 for the BVH side, =1000 for the scan side). usage: RTHIP_LIB=... python tools/bvh_crossover.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, abi, flatten, presets
for name, scene in (("ggez 41", presets.ggez_scene()), ("big 4x4+1", presets.big_scene(4, 7)), ("big 6x6+1", presets.big_scene(6, 7)), ("big 8x8+1", presets.big_scene(8, 7)),
                    ("big 9x9+1", presets.big_scene(9, 7)), ("big 10x10+1", presets.big_scene(10, 7)), ("big 11x11+1", presets.big_scene(11, 7)), ("big 13x13+1", presets.big_scene(13, 7)), ("big 16x16+1", presets.big_scene(16, 7)),
                    ("mesh 20 tris + 17 spheres", presets.big_scene(4, 7, 0)), ("mesh 80 tris + 17 spheres", presets.big_scene(4, 7, 1))):
    flat = flatten(scene)
    rt = Raytracer.new(1920, 1080, seed=1)
    st = rt.render(flat, 32, max_depth=8, want_accum=False, flags=abi.F_COUNT_STATS).stats
    best = min(rt.render(flat, 32, max_depth=8, want_accum=False).stats["kernel_ms"] for _ in range(3))
    print(f"{name}: {len(flat.spheres)} spheres, node tests/ray {st['node_tests'] / st['rays']:.1f}, sphere tests/ray {st['sphere_tests'] / st['rays']:.1f}: {best:.2f} ms, {st['rays'] / best / 1e3:.0f} Mrays/s", flush=True)
    rt.close()
