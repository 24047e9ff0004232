"""Timing of the SURVEY §8d configurations C2-C5 on one GPU (instrumented + plain runs).
usage: python tools/config_bench.py [C2 C2d50 C2L C3 C4 C4L C5] ; env SPP_SCALE=0.25 scales the sample counts;
CPU_CROP=64 also times the CPU oracle (f64, all host threads) on a CPU_CROP^2 window in the middle of the frame at 1 spp
(the CPU baseline of C4/C5 is a cropped window, reported in Mrays/s) — the linear scan of ray.rs:47-67, or with
CPU_ACCEL=1 the oracle's own bounding-volume tree (same answers, tests/test_oracle_accel.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from rustracer_amd import Raytracer, abi, flatten, presets

scale = float(os.environ.get("SPP_SCALE", "1"))
CONFIGS = {
    "C2": dict(scene=presets.standalone_scene, w=3840, h=2160, spp=256, depth=5),
    "C2d50": dict(scene=presets.standalone_scene, w=3840, h=2160, spp=256, depth=50),
    "C2L": dict(scene=presets.standalone_scene, lights=presets.standalone_lights, flags=abi.F_POINT_LIGHTS, w=3840, h=2160, spp=256, depth=5),
    "C3": dict(scene=presets.textured_scene, w=3840, h=2160, spp=64, depth=50),
    "C4": dict(scene=presets.mesh_scene, w=3840, h=2160, spp=16, depth=50),
    "C4L": dict(scene=presets.mesh_scene, lights=presets.standalone_lights, flags=abi.F_POINT_LIGHTS, w=3840, h=2160, spp=16, depth=50),
    "C5": dict(scene=lambda: presets.big_scene(100, 7, "uv100k"), w=7680, h=4320, spp=64, depth=8),
}
names = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
lib = abi.lib()
for name in names:
    cfg = CONFIGS[name]
    t0 = time.time(); flat = flatten(cfg["scene"](), cfg["lights"]() if "lights" in cfg else ()); t_flat = time.time() - t0
    fl = cfg.get("flags", 0)
    rt = Raytracer.new(cfg["w"], cfg["h"], seed=1)
    t0 = time.time(); rt.upload(flat); t_up = time.time() - t0
    spp = max(1, int(cfg["spp"] * scale))
    out = {"config": name, "spheres": len(flat.spheres), "triangles": len(flat.triangles), "flatten_s": round(t_flat, 2), "upload_bvh_s": round(t_up, 2),
           "w": cfg["w"], "h": cfg["h"], "spp": spp, "depth": cfg["depth"]}
    res = rt.render(flat, spp, max_depth=cfg["depth"], want_accum=False, flags=abi.F_COUNT_STATS | fl)
    st = abi.Stats()
    for k, v in res.stats.items():
        if k == "hits":
            for i in range(4): st.hits[i] = v[i]
        else:
            setattr(st, k, v)
    flops = lib.rthip_algorithmic_flops(C.byref(st))
    best = 1e30
    for _ in range(int(os.environ.get("FRAMES", "4"))):  # (pass 1 of a frame layout measures its pixel groups, pass 2 sorts them: rthip_api.cpp)
        r2 = rt.render(flat, spp, max_depth=cfg["depth"], want_accum=False, flags=fl)
        best = min(best, r2.stats["kernel_ms"])
    s = res.stats
    out.update(rays=s["rays"], rays_per_path=round(s["rays"] / s["paths"], 3), node_tests_per_ray=round(s["node_tests"] / max(1, s["rays"]), 2),
               sphere_tests_per_ray=round(s["sphere_tests"] / max(1, s["rays"]), 2), tri_tests_per_ray=round(s["tri_tests"] / max(1, s["rays"]), 2),
               kernel_ms=round(best, 3), mrays_s=round(s["rays"] / best / 1e3, 1), tflops=round(flops / best / 1e9, 2),
               roofline_frac=round(flops / best / 1e9 / 157.3, 4), flops_per_ray=round(flops / s["rays"], 1))
    crop = int(os.environ.get("CPU_CROP", "0"))
    if crop:
        import oracle
        from rustracer_amd import make_params
        cam = oracle.camera_lookat(-1.8, 1.0, 2.0)
        window = ((cfg["w"] - crop) // 2, (cfg["h"] - crop) // 2, crop, crop)
        _, _, ost, secs = oracle.render(flat, cam, make_params(cfg["w"], cfg["h"], 1, cfg["depth"], 1, fl), precision=64, crop=window,
                                            accel=bool(os.environ.get("CPU_ACCEL")))
        out.update(cpu_oracle_mrays_s=round(ost["rays"] / secs / 1e6, 3), cpu_threads=oracle.hw_threads(), cpu_sample=f"{crop}x{crop} window @ 1 spp, {secs:.1f} s",
                   cpu_tests_per_ray=round((ost["sphere_tests"] + ost["tri_tests"]) / max(1, ost["rays"]), 1))
    print(json.dumps(out), flush=True)
    rt.close()
