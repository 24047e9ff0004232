"""Tree scenes: kernel time against the number of sample slices (RTHIP_TREE_SLICES is read once per process: one process per value).
usage: python tools/ab_slices.py C4|C5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, abi, flatten, presets
name = sys.argv[1]
cfg = {"C4": (presets.mesh_scene, 3840, 2160, 16, 50), "C5": (lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 64, 8)}[name]
flat = flatten(cfg[0]())
rt = Raytracer.new(cfg[1], cfg[2], seed=1)
rt.upload(flat)
best = 1e30
for _ in range(3):
    s = rt.render(flat, cfg[3], max_depth=cfg[4], want_accum=False).stats
    best = min(best, s["kernel_ms"])
print(f"{name} slices {os.environ.get('RTHIP_TREE_SLICES', 'default')}: kernel_ms {best:.2f} Mrays/s {s['rays'] / best / 1e3:.0f}", flush=True)
