"""ad-hoc first GPU check: device render vs oracle on the standalone scene"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from parity import mismatch_histogram, describe
from rustracer_amd import Raytracer, presets, flatten, make_params

W, H = 480, 270
scene = presets.standalone_scene()
flat = flatten(scene)
cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
rt = Raytracer.new(W, H, seed=1)
for spp, depth in ((1, 50), (16, 50), (16, 5)):
    res = rt.render(flat, spp, max_depth=depth)
    p = make_params(W, H, spp, depth, 1)
    for prec in (64, 32):
        rgba, acc, st, secs = oracle.render(flat, cam, p, precision=prec)
        hist, ok = mismatch_histogram(res.rgba8, rgba)
        print(f"spp={spp} depth={depth} vs oracle f{prec}: {describe(hist, ok)} | gpu rays={res.stats['rays']} oracle rays={st['rays']} kernel_ms={res.stats['kernel_ms']:.3f} cpu_s={secs:.2f}")
print("pick centre:", rt.get_shape(flat, (W - 1) * 0.5, (H - 1) * 0.5))
# 4K timing
rt4 = Raytracer.new(3840, 2160, seed=1)
for spp, depth in ((1, 50), (16, 5), (64, 5), (64, 50)):
    res = rt4.render(flat, spp, max_depth=depth, want_accum=False)
    s = res.stats
    print(f"4K spp={spp} depth={depth}: kernel_ms={s['kernel_ms']:.2f} rays={s['rays']} -> {s['rays'] / s['kernel_ms'] / 1e3:.1f} Mrays/s d2h_ms={s['d2h_ms']:.2f}")
