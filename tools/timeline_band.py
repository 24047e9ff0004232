"""Debug (RTHIP_LIB=build_ab/timeline.so, built with -DRT_TIMELINE): when every wave of a SLICED band launch of the C2 frame ran
(rank RANK of WORLD interleaved 8-row bands). Resident waves over time, wave durations per slice, start order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, int(os.environ.get("SPP", "256")), 5
world, rank = int(os.environ.get("WORLD", "8")), int(os.environ.get("RANK", "0"))
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
kw = dict(max_depth=DEPTH, want_accum=False, band_rows=8, band_first=rank, band_stride=world)
rt.render(flat, SPP, **kw)
r = rt.render(flat, SPP, **kw)
img = r.rgba8.view(np.uint32).reshape(-1, W)
rows = img.shape[0]
n_slices = (SPP + 63) // 64
S, E, K = [], [], []
for k in range(n_slices):
    t = img[k::8]  # row k of every tile: slice k's stamps
    S.append(t[:, 0::8].astype(np.uint64) | (t[:, 1::8].astype(np.uint64) << 32))
    E.append(t[:, 2::8].astype(np.uint64) | (t[:, 3::8].astype(np.uint64) << 32))
start, end = np.stack(S), np.stack(E)
t0 = start.min()
s, e = (start - t0).astype(np.float64) / 100.0, (end - t0).astype(np.float64) / 100.0
dur = e - s
total = e.max()
print(f"world {world} rank {rank}: kernel_ms {r.stats['kernel_ms']:.3f}; waves {s.size}; span {total:.0f} us; wave time sum {dur.sum() / 1e3:.1f} ms "
      f"(/ 7168 slots = {dur.sum() / 7168:.0f} us); mean wave {dur.mean():.0f} us, median {np.median(dur):.0f}, p99 {np.percentile(dur, 99):.0f}, max {dur.max():.0f}")
edges = np.linspace(0, total, 41)
for i in range(40):
    mid = 0.5 * (edges[i] + edges[i + 1])
    live = (s <= mid) & (e > mid)
    n = int(live.sum())
    print(f"  t={mid:7.0f} us resident waves {n:5d}  mean duration of those {dur[live].mean() if n else 0:6.0f} us " + "#" * (n // 120))
# how well the launch order follows the cost: duration against start time
o = np.argsort(s.ravel())
d = dur.ravel()[o]
for i in range(0, d.size, max(1, d.size // 16)):
    seg = d[i:i + d.size // 16]
    print(f"  started {i:6d}..: mean wave {seg.mean():6.0f} us max {seg.max():6.0f}")
