"""C4 / C5 kernel time against the number of traversal-stack levels kept in LDS (RTHIP_STACK_LDS: the rest is private memory).
The LDS of a one-wave workgroup is levels*256 + 3552 B (round 2's pixel table: + 3776); gfx950 allocates it in blocks of 1280 B, 128 per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
which = sys.argv[1:] or ["C4", "C5"]
cfgs = {"C4": (presets.mesh_scene, 3840, 2160, 16, 50), "C5": (lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 64, 8)}
for name in which:
    mk, w, h, spp, depth = cfgs[name]
    flat = flatten(mk())
    rt = Raytracer.new(w, h, seed=1)
    rt.render(flat, 1, max_depth=depth, want_accum=False)
    for levels in [int(x) for x in os.environ.get("LEVELS", "8,7,6,5,4,8,5").split(",")]:
        os.environ["RTHIP_STACK_LDS"] = str(levels)
        ts = [rt.render(flat, spp, max_depth=depth, want_accum=False).stats["kernel_ms"] for _ in range(2 if name == "C5" else 3)]
        print(f"{name} levels {levels}: LDS {levels * 256 + 3552} B  kernel_ms " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    rt.close()
