"""Kernel-level strong-scaling check on ONE GPU: render rank r's interleaved bands of the C2 frame for world = 1, 2, 4, 8
and compare the kernel time with (whole-frame time / world): what the render kernel can scale to before any exchange."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, int(os.environ.get("SPP", "256")), 5
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
rt.render(flat, SPP, max_depth=DEPTH, want_accum=False)
base = min(rt.render(flat, SPP, max_depth=DEPTH, want_accum=False).stats["kernel_ms"] for _ in range(3))
print(f"world 1: {base:.3f} ms")
for world in (2, 4, 8):
    times = []
    for rank in range(world):
        times.append(min(rt.render(flat, SPP, max_depth=DEPTH, want_accum=False, band_rows=8, band_first=rank, band_stride=world).stats["kernel_ms"] for _ in range(3)))
    worst = max(times)
    print(f"world {world}: slowest rank {worst:.3f} ms, mean {sum(times) / world:.3f} ms, ideal {base / world:.3f} ms, kernel scaling efficiency {base / world / worst * 100:.1f} %"
          f"  per rank: {' '.join(f'{t:.2f}' for t in times)}")
