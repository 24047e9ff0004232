"""A/B of library variants on the progressive pass (1 and 2 spp per pass, accumulate, 4K and 960x540): kernel ms per pass.
python tools/ab_progressive.py libA.so libB.so ..."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import abi, flatten, presets, make_params

flat = flatten(presets.standalone_scene())
for path in sys.argv[1:]:
    l = C.CDLL(path)
    for name, (res, args) in abi.SYMBOLS.items():
        fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    ctx = C.c_void_p(); assert l.rthip_create(0, C.byref(ctx)) == 0
    assert l.rthip_upload_scene(ctx, C.byref(flat.desc)) == 0
    cam = abi.Camera(); l.rthip_camera_lookat(-1.8, 1.0, 2.0, (C.c_double * 3)(0, 0, -1), C.byref(cam)); l.rthip_set_camera(ctx, C.byref(cam))
    out = []
    for w, h in ((960, 540), (3840, 2160)):
        for spp in (1, 2, 8):
            st = abi.Stats(); ts = []
            for k in range(40):
                p = make_params(w, h, spp, 50, 1, abi.F_ACCUMULATE if k else 0)
                assert l.rthip_render(ctx, C.byref(p), None, None, C.byref(st)) == 0
                if k >= 10: ts.append(st.kernel_ms)
            out.append(f"{w}x{h}@{spp}: {sorted(ts)[len(ts)//2]:.4f} ms")
    print(os.path.basename(path), " | ".join(out), flush=True)
    l.rthip_destroy(ctx)
