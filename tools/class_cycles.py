"""Debug: per tile class (0 sky-only, 1 primary-inline, 2 general) waves and wall-clock ticks of a -DRT_CLASS_CYCLES build.
python tools/class_cycles.py build_ab/classcyc.so"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import abi, flatten, presets, make_params

l = C.CDLL(sys.argv[1])
for name, (res, args) in abi.SYMBOLS.items():
    fn = getattr(l, name); fn.restype = res; fn.argtypes = args
flat = flatten(presets.standalone_scene())
ctx = C.c_void_p(); assert l.rthip_create(0, C.byref(ctx)) == 0
assert l.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
cam = abi.Camera(); l.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double * 3)(0, 0, -1), C.byref(cam)); l.rthip_set_camera(ctx, C.byref(cam))
p = make_params(3840, 2160, int(os.environ.get("SPP", "256")), 5, 0)
st = abi.Stats()
for _ in range(3):
    assert l.rthip_render(ctx, C.byref(p), None, None, C.byref(st)) == 0
waves = [st.sphere_tests, st.plane_tests, st.tri_tests]
ticks = [st.hits[0], st.hits[1], st.hits[2]]
tot = sum(ticks)
print("kernel_ms", st.kernel_ms, "rays", st.rays)
for c, name in enumerate(("sky-only", "primary-inline", "general")):
    print(f"{name:15s} waves {waves[c]:7d} ticks {ticks[c]:12d} share {ticks[c] / tot:.3f} us/wave {ticks[c] / max(waves[c], 1) / 100:.1f}")
print(f"primary-inline loop: wave iterations x 64 = {st.node_tests}, lane iterations = {st.sky_evals} ({st.sky_evals / max(st.node_tests, 1):.3f}), "
      f"fresh = {st.shadow_rays} ({st.shadow_rays / max(st.node_tests, 1):.3f}), ended at the primary ray = {st.hits[3]} ({st.hits[3] / max(st.node_tests, 1):.3f})")
l.rthip_debug_shards.restype = C.POINTER(C.c_ulonglong)
l.rthip_debug_shards.argtypes = [C.c_void_p]
sh = l.rthip_debug_shards(ctx)
N = 14  # RT_STAT_N: 0 paths, 1 rays, 2-4 waves per class, 5 wave iterations x 64, 6-8 ticks per class, 9 ended at primary, 10 lane iterations, 11 fresh
if os.environ.get("ARMS"):  # -DRT_ARM_COUNTS build: slots 2-4, 6-8 count the divergent arms instead
    print("pmask: wave iterations; per wave iteration: texture arm runs / lanes in it, dielectric arm, non-Lambertian scatter arm")
    for m in list(range(32)):
        r = [sh[m * N + i] for i in range(N)]
        if not r[5]:
            continue
        wi = r[5] / 64
        print(f"  {m & 15:04b}: wave iterations {wi:12.0f}  texture {r[2] / 64 / wi:.3f} runs, {r[3] / max(r[2] / 64, 1):5.1f} lanes/run | dielectric {r[4] / 64 / wi:.3f} runs, "
              f"{r[6] / max(r[4] / 64, 1):5.1f} lanes/run | non-Lambertian {r[7] / 64 / wi:.3f} runs, {r[8] / max(r[7] / 64, 1):5.1f} lanes/run")
    sys.exit(0)
print("pmask(+16 = general loop): waves, share of ticks, us/wave, paths, rays/path, wave iterations/wave, lane occupancy, fresh share")
for m in list(range(32)) + [255]:
    r = [sh[m * N + i] for i in range(N)]
    wv = r[2] + r[3] + r[4]
    if not wv:
        continue
    tk = r[6] + r[7] + r[8]
    print(f"  {m:3d} {m & 15:04b}: waves {wv:6d} share {tk / tot:.3f} us/wave {tk / wv / 100:7.1f} paths {r[0]:11d} rays/path {r[1] / max(r[0], 1):.2f} "
          f"iters/wave {r[5] / 64 / wv:7.1f} occupancy {r[10] / max(r[5], 1):.3f} fresh {r[11] / max(r[5], 1):.3f}")
