"""A/B timing of librthip variants on the bench workload: python tools/ab_bench.py libA.so libB.so ..."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rustracer_amd import abi, flatten, presets, make_params
from rustracer_amd.scene import Camera

def load(path):
    l = C.CDLL(path)
    for name, (res, args) in abi.SYMBOLS.items():
        fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    return l

scene = presets.standalone_scene()
if os.environ.get("NOTEX"):  # the texture sphere becomes Lambertian: what of the HBM fetch traffic is the texture?
    from rustracer_amd import Lambertian
    scene[3] = type(scene[3]).new(scene[3].center, scene[3].radius, Lambertian.new_from_hex(0x707000))
flat = flatten(scene)
libs = [(p, load(p)) for p in sys.argv[1:]]
W, H = 3840, 2160
spp = int(os.environ.get("SPP", "64")); depth = int(os.environ.get("DEPTH", "5"))
TARGET = tuple(float(v) for v in os.environ["TARGET"].split(",")) if os.environ.get("TARGET") else (0.0, 0.0, -1.0)  # camera target
ctxs = []
for name, l in libs:
    ctx = C.c_void_p(); assert l.rthip_create(0, C.byref(ctx)) == 0
    assert l.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
    cam = abi.Camera(); l.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double*3)(*TARGET), C.byref(cam)); l.rthip_set_camera(ctx, C.byref(cam))
    ctxs.append(ctx)
BAND = tuple(int(v) for v in os.environ["BAND"].split(",")) if os.environ.get("BAND") else (0, 0, 1)
out = np.zeros((H, W, 4), np.uint8)
ref = None
REP = int(os.environ.get("REP", "4"))  # back-to-back launches per library and round: the clocks stay up, the minimum is reported
for rnd in range(3):
    for (name, l), ctx in zip(libs, ctxs):
        p = make_params(W, H, spp, depth, rnd, 0, *BAND)  # BAND="rows,first,stride": only that band of rows (e.g. ground-only tiles)
        st = abi.Stats()
        times = []
        for _ in range(REP):
            rc = l.rthip_render(ctx, C.byref(p), None, None, C.byref(st))
            assert rc == 0, l.rthip_last_error(ctx)
            times.append(st.kernel_ms)
        rc = l.rthip_render(ctx, C.byref(p), out.ctypes.data, None, C.byref(st))
        crc = int(out.view(np.uint32).sum(dtype=np.uint64))
        print(f"round {rnd} {os.path.basename(name):28s} kernel_ms min={min(times):8.3f} med={sorted(times)[len(times)//2]:8.3f} rays={st.rays} -> {st.rays/min(times)/1e3:9.1f} Mrays/s checksum={crc}", flush=True)
