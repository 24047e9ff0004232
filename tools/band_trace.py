"""One rank's bands of the C2 frame at world = 8 (env WORLD), a few passes: the command to put under `rocprofv3 --kernel-trace --stats`
to see what a 1/8-frame pass is made of (tile cost / order kernels, render, resolve)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, int(os.environ.get("SPP", "256")), 5
world, rank = int(os.environ.get("WORLD", "8")), int(os.environ.get("RANK", "0"))
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
ts = []
for _ in range(int(os.environ.get("PASSES", "8"))):
    ts.append(rt.render(flat, SPP, max_depth=DEPTH, want_accum=False, band_rows=8, band_first=rank, band_stride=world).stats["kernel_ms"])
print("kernel_ms per pass:", " ".join(f"{t:.3f}" for t in ts))
