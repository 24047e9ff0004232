import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, abi, flatten, presets
for name, sc, w, h, spp, d in (("C4", presets.mesh_scene, 3840, 2160, 16, 50), ("C5", lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 64, 8)):
    flat = flatten(sc()); rt = Raytracer.new(w, h, seed=1); rt.upload(flat)
    for label, fl in (("lists", abi.F_TILE_LISTS), ("plain", 0)):
        s = rt.render(flat, spp, max_depth=d, want_accum=False, flags=fl).stats
        print(name, label, "kernel_ms", round(s["kernel_ms"], 2), "primary-phase ticks", s["pixels_traced"], "pool ticks", s["pixels_final"],
              "share", round(s["pixels_traced"] / max(1, s["pixels_final"]), 3), flush=True)
    rt.close()
