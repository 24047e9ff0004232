"""A/B of library variants on a scene preset at 4K: python tools/ab_scene.py plane_scene libA.so libB.so   (env SPP, DEPTH)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import abi, flatten, presets, make_params

flat = flatten(getattr(presets, sys.argv[1])())
spp, depth = int(os.environ.get("SPP", "64")), int(os.environ.get("DEPTH", "50"))
for path in sys.argv[2:]:
    l = C.CDLL(path)
    for name, (res, args) in abi.SYMBOLS.items():
        fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    ctx = C.c_void_p(); assert l.rthip_create(0, C.byref(ctx)) == 0
    assert l.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
    cam = abi.Camera(); l.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double * 3)(0, 0, -1), C.byref(cam)); l.rthip_set_camera(ctx, C.byref(cam))
    st, ts = abi.Stats(), []
    for k in range(6):
        p = make_params(3840, 2160, spp, depth, 1)
        assert l.rthip_render(ctx, C.byref(p), None, None, C.byref(st)) == 0
        ts.append(st.kernel_ms)
    print(f"{sys.argv[1]} {os.path.basename(path):16s} kernel_ms min {min(ts):.3f}  rays {st.rays}  {st.rays / min(ts) / 1e3:.0f} Mrays/s", flush=True)
    l.rthip_destroy(ctx)
