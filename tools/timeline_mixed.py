"""Debug (RTHIP_LIB=build_ab/timeline.so, built with -DRT_TIMELINE): per-wave time stamps of the whole C2 frame when the launch is MIXED
(whole groups + sliced groups, rthip_api.cpp): every row of a tile that carries the marker and a stamp of the last pass counts as a wave."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Raytracer, flatten, presets
W, H, SPP, DEPTH = 3840, 2160, int(os.environ.get("SPP", "256")), 5
flat = flatten(presets.standalone_scene())
rt = Raytracer.new(W, H, seed=1)
for _ in range(int(os.environ.get("WARM", "3"))):
    rt.render(flat, SPP, max_depth=DEPTH, want_accum=False)
r = rt.render(flat, SPP, max_depth=DEPTH, want_accum=False)
img = r.rgba8.view(np.uint32).reshape(H, W)
S, E, NS = [], [], []
for k in range(8):
    t = img[k::8]
    ok = ((t[:, 6::8] & 0xFFFFFF00) == 0xC0FFEE00) & ((t[:, 7::8] & 0xFF) == k) & ((t[:, 7::8] & 0xFFFFFF00) == 0xC0FFEE00)
    st = t[:, 0::8].astype(np.uint64) | (t[:, 1::8].astype(np.uint64) << 32)
    en = t[:, 2::8].astype(np.uint64) | (t[:, 3::8].astype(np.uint64) << 32)
    S.append(st[ok]); E.append(en[ok]); NS.append((t[:, 6::8] & 0xFF)[ok])
s_all, e_all, ns_all = np.concatenate(S), np.concatenate(E), np.concatenate(NS)
last = e_all.max()
keep = s_all.astype(np.float64) > float(last) - r.stats["kernel_ms"] * 1.08e5  # stamps of the last pass only (100 MHz ticks)
s_all, e_all, ns_all = s_all[keep], e_all[keep], ns_all[keep]
t0 = s_all.min()
s, e = (s_all - t0).astype(np.float64) / 100.0, (e_all - t0).astype(np.float64) / 100.0
dur = e - s
total = e.max()
print(f"kernel_ms {r.stats['kernel_ms']:.3f}; waves {s.size} (whole {int((ns_all == 1).sum())}, slices {int((ns_all > 1).sum())}); span {total:.0f} us; "
      f"wave time sum {dur.sum() / 1e6:.2f} s (/ 7168 = {dur.sum() / 7168:.0f} us)")
edges = np.linspace(0, total, 49)
for i in range(48):
    mid = 0.5 * (edges[i] + edges[i + 1])
    live = (s <= mid) & (e > mid)
    n = int(live.sum())
    print(f"  t={mid:7.0f} us resident {n:5d} (slices {int((live & (ns_all > 1)).sum()):5d})  mean duration {dur[live].mean() if n else 0:6.0f} us " + "#" * (n // 150))
