"""C4 / C5 with and without the primary-visibility lists (RTHIP_F_TILE_LISTS), one library, one box.
usage: python tools/ab_lists.py [C4 C5] ; env SPP_SCALE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Raytracer, abi, flatten, presets

scale = float(os.environ.get("SPP_SCALE", "1"))
CONFIGS = {
    "C4": dict(scene=presets.mesh_scene, w=3840, h=2160, spp=16, depth=50),
    "C4L": dict(scene=presets.mesh_scene, lights=presets.standalone_lights, flags=abi.F_POINT_LIGHTS, w=3840, h=2160, spp=16, depth=50),
    "C5": dict(scene=lambda: presets.big_scene(100, 7, "uv100k"), w=7680, h=4320, spp=64, depth=8),
}
for name in sys.argv[1:] or ["C4", "C5"]:
    cfg = CONFIGS[name]
    flat = flatten(cfg["scene"](), cfg["lights"]() if "lights" in cfg else ())
    rt = Raytracer.new(cfg["w"], cfg["h"], seed=1)
    rt.upload(flat)
    spp = max(1, int(cfg["spp"] * scale))
    fl = cfg.get("flags", 0)
    crc = {}
    for label, extra in (("lists", abi.F_TILE_LISTS), ("plain", 0), ("lists", abi.F_TILE_LISTS), ("plain", 0)):
        r = rt.render(flat, spp, max_depth=cfg["depth"], want_accum=False, flags=fl | extra)
        c = int(r.rgba8.view(np.uint32).sum(dtype=np.uint64))
        crc.setdefault(label, c)
        print(f"{name} {label:6s} kernel_ms {r.stats['kernel_ms']:9.3f} Mrays/s {r.stats['rays'] / r.stats['kernel_ms'] / 1e3:9.1f} checksum {c}", flush=True)
    assert crc["lists"] == crc["plain"], "the image changed"
    for label, extra in (("lists", abi.F_TILE_LISTS), ("plain", 0)):
        s = rt.render(flat, spp, max_depth=cfg["depth"], want_accum=False, flags=fl | extra | abi.F_COUNT_STATS).stats
        print(f"{name} {label:6s} counted: rays/path {s['rays'] / s['paths']:.3f} node tests/ray {s['node_tests'] / s['rays']:.2f} "
              f"tri tests/ray {s['tri_tests'] / s['rays']:.2f} sphere tests/ray {s['sphere_tests'] / s['rays']:.2f}", flush=True)
    rt.close()
