"""Whole C2 frame, a few passes in a row on one context: kernel_ms per pass (the launch-order states: pass 1 = cost model +
measuring, pass 2 = sorts the measurements, pass 3.. = reuse). Run with RTHIP_ORDER_ALL=1 / RTHIP_NO_TILE_ORDER=1 to compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, int(os.environ.get("SPP", "256")), int(os.environ.get("DEPTH", "5"))
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
ts = [rt.render(flat, SPP, max_depth=DEPTH, want_accum=False).stats["kernel_ms"] for _ in range(int(os.environ.get("PASSES", "10")))]
print(os.environ.get("TAG", ""), "kernel_ms:", " ".join(f"{t:.3f}" for t in ts), " min of 3..:", f"{min(ts[2:]):.3f}")
