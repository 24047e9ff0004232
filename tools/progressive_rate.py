"""Wall-clock rate of the progressive window loop (standalone/src/main.rs:99-102: passes of 1 spp, accumulate, adaptive):
rthip_render into a page-locked frame, per pass, at the standalone window size and at 4K."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import Raytracer, flatten, host_array, presets
flat = flatten(presets.standalone_scene())
for w, h in ((960, 540), (3840, 2160)):
    for adaptive in (False, True):
        rt = Raytracer.new(w, h, seed=1)
        frame = host_array((h, w, 4), np.uint8)
        rt.render(flat, 1, max_depth=50, want_accum=False, rgba_out=frame, adaptive=adaptive)
        n = 60
        t0 = time.perf_counter()
        for k in range(n):
            res = rt.render(flat, 1, max_depth=50, accumulate=True, want_accum=False, rgba_out=frame, adaptive=adaptive)
        dt = (time.perf_counter() - t0) / n
        extra = f", {res.stats['pixels_final']} of {w * h} pixels Final after {n + 1} passes" if adaptive else ""
        print(f"{w}x{h} 1 spp/pass adaptive={adaptive}: {dt * 1e3:.3f} ms per pass wall ({1 / dt:.0f} passes/s), kernel {res.stats['kernel_ms']:.3f} ms, d2h {res.stats['d2h_ms']:.3f} ms{extra}")
        rt.close()

# the same loop through rthip_render_async / rthip_wait: two frames in flight, the copy of pass k beside the kernel of pass k+1
for w, h in ((960, 540), (3840, 2160)):
    for adaptive in (False, True):
        rt = Raytracer.new(w, h, seed=1)
        frames = [host_array((h, w, 4), np.uint8), host_array((h, w, 4), np.uint8)]
        rt.render_async(flat, 1, frames[0], 0, adaptive=adaptive)
        n = 200
        t0 = time.perf_counter()
        for k in range(1, n + 1):
            rt.render_async(flat, 1, frames[k & 1], k & 1, accumulate=True, adaptive=adaptive)
            rt.wait((k - 1) & 1)  # frame k-1 is ready for the window while pass k runs
        rt.wait(n & 1)
        dt = (time.perf_counter() - t0) / n
        print(f"{w}x{h} 1 spp/pass adaptive={adaptive} ASYNC: {dt * 1e3:.3f} ms per pass wall ({1 / dt:.0f} passes/s)")
        rt.close()
