"""Debug (RTHIP_LIB=build_ab/timeline.so, built with -DRT_TIMELINE): when and where every wave of a BVH launch ran.
Prints the number of resident waves over time, the per-XCD end times and the longest waves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Raytracer, abi, flatten, presets

which = sys.argv[1] if len(sys.argv) > 1 else "C4"
cfg = {"C4": (presets.mesh_scene, 3840, 2160, 16, 50), "C5": (lambda: presets.big_scene(100, 7, "uv100k"), 7680, 4320, 64, 8),
       "C2": (presets.standalone_scene, 3840, 2160, 256, 5)}[which]
flat = flatten(cfg[0]())
w, h = cfg[1], cfg[2]
rt = Raytracer.new(w, h, seed=1)
fl = 0 if os.environ.get("PLAIN", "1") == "1" else abi.F_TILE_LISTS
for _ in range(int(os.environ.get("WARM", "2"))):  # (pass 1 measures the groups, pass 2 sorts them: the third runs in the final order)
    rt.render(flat, cfg[3], max_depth=cfg[4], want_accum=False, flags=fl)
r = rt.render(flat, cfg[3], max_depth=cfg[4], want_accum=False, flags=fl)
img = r.rgba8.view(np.uint32).reshape(h, w)
t = img[0::8]  # first row of every tile
start = t[:, 0::8].astype(np.uint64) | (t[:, 1::8].astype(np.uint64) << 32)
end = t[:, 2::8].astype(np.uint64) | (t[:, 3::8].astype(np.uint64) << 32)
hw, xcc = t[:, 4::8], t[:, 5::8] & 0xF
t0 = start.min()
s, e = (start - t0).astype(np.float64) / 100.0, (end - t0).astype(np.float64) / 100.0  # microseconds (100 MHz)
dur = e - s
total = e.max()
print(f"{which}: kernel_ms {r.stats['kernel_ms']:.2f}; waves {s.size}; span {total / 1e3:.2f} ms; wave time sum {dur.sum() / 1e6:.1f} s; "
      f"mean wave {dur.mean():.0f} us, median {np.median(dur):.0f}, p99 {np.percentile(dur, 99):.0f}, max {dur.max():.0f}")
edges = np.linspace(0, total, 41)
for i in range(40):
    mid = 0.5 * (edges[i] + edges[i + 1])
    n = int(((s <= mid) & (e > mid)).sum())
    print(f"  t={mid / 1e3:6.2f} ms resident waves {n:5d} " + "#" * (n // 120))
for k in range(8):
    m = xcc == k
    if not m.any():
        continue
    print(f"  XCD {k}: waves {int(m.sum())} last end {e[m].max() / 1e3:.2f} ms, wave time {dur[m].sum() / 1e6:.2f} s")
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 50 + cu
u, inv = np.unique(key, return_inverse=True)
per = np.bincount(inv.ravel(), weights=dur.ravel())
print(f"  {len(u)} distinct (xcd, se, sh, cu); wave time per CU: min {per.min() / 1e3:.1f} ms max {per.max() / 1e3:.1f} ms mean {per.mean() / 1e3:.1f} ms")
order = np.argsort(dur.ravel())[::-1][:8]
for o in order:
    ty, tx = divmod(int(o), dur.shape[1])
    print(f"  long wave: tile ({tx},{ty}) start {s.ravel()[o] / 1e3:.2f} ms dur {dur.ravel()[o] / 1e3:.2f} ms")
# what an interleaved-band partition of this frame would give each rank (sum of its bands' wave times), straight and zigzag
band = dur.sum(axis=1)  # one entry per 8-row band
for world in (2, 4, 8):
    idx = np.arange(band.size)
    k, j = idx // world, idx % world
    for name, owner in (("straight", j), ("zigzag", np.where(k % 2 == 0, j, world - 1 - j))):
        per = np.array([band[owner == r].sum() for r in range(world)])
        print(f"  world {world} {name:8s}: max/mean over ranks {per.max() / per.mean():.4f}  per rank (ms of wave time): {' '.join(f'{p / 1e3:.0f}' for p in per)}")
