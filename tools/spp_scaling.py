"""Kernel time per sample against the samples per pixel of one pass (the share of a wave's ramp-down — lanes running out of items while the
tile's last long paths finish — falls with the items per tile): C4 / C5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
cfgs = {"C4": (presets.mesh_scene, 3840, 2160, [16, 32, 64, 128], 50), "C5": (lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, [16, 64, 128], 8)}
for name in sys.argv[1:] or ["C4", "C5"]:
    mk, w, h, spps, depth = cfgs[name]
    flat = flatten(mk())
    rt = Raytracer.new(w, h, seed=1)
    for spp in spps:
        ts = [rt.render(flat, spp, max_depth=depth, want_accum=False).stats["kernel_ms"] for _ in range(4)]
        print(f"{name} {spp} spp: kernel_ms {min(ts[1:]):.2f} = {min(ts[1:]) / spp:.3f} ms per sample", flush=True)
    rt.close()
