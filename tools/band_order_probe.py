"""Is the per-rank spread of tools/band_scaling.py a property of the rank's bands or of when it is measured? world 8, ranks measured
in the order given by ORDER (default 7..0), 5 passes each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, 256, 5
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
for _ in range(3):
    rt.render(flat, SPP, max_depth=DEPTH, want_accum=False)
order = [int(x) for x in os.environ.get("ORDER", "7,6,5,4,3,2,1,0").split(",")]
for rank in order:
    ts = [rt.render(flat, SPP, max_depth=DEPTH, want_accum=False, band_rows=8, band_first=rank, band_stride=8).stats["kernel_ms"] for _ in range(5)]
    print(f"rank {rank}: " + " ".join(f"{t:.3f}" for t in ts))
