"""C4 / C5: kernel time of consecutive passes on one context (pass 1 runs in row order and measures the pixel groups, pass 2 sorts the
measurements, pass 3.. reuse the order). RTHIP_NO_TREE_ORDER=1 / RTHIP_TREE_SLICES=n to compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
which = sys.argv[1:] or ["C4", "C5"]
cfgs = {"C4": (presets.mesh_scene, 3840, 2160, 16, 50), "C5": (lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 64, 8)}
for name in which:
    mk, w, h, spp, depth = cfgs[name]
    flat = flatten(mk())
    rt = Raytracer.new(w, h, seed=1)
    ts = [rt.render(flat, spp, max_depth=depth, want_accum=False).stats["kernel_ms"] for _ in range(5 if name == "C4" else 4)]
    print(f"{os.environ.get('TAG', '')} {name}: kernel_ms " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    rt.close()
