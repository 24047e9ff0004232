"""Do HIP stream priorities order the DISPATCH of two concurrent kernels on this GPU? Context A renders the whole C2 frame on a
high-priority stream, context B a 1/8-frame band launch on a normal-priority stream, both enqueued at once. If priorities work, B's
workgroups only get slots when A has none left to dispatch: B ends right after A, and A takes no longer than alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustracer_amd import Raytracer, flatten, presets, make_params
W, H, SPP, DEPTH = 3840, 2160, 256, 5
flat = flatten(presets.standalone_scene())
A, B = Raytracer.new(W, H, seed=1), Raytracer.new(W, H, seed=1)
A.upload(flat); B.upload(flat)
pa = make_params(W, H, SPP, DEPTH, 1)
pb = make_params(W, H, SPP, DEPTH, 1, 0, 8, 0, 8)
bufa = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
bufb = torch.empty((H // 8 + 8, W, 4), dtype=torch.uint8, device="cuda")
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (lowest, highest):", lo_p, hi_p)
sa, sb = torch.cuda.Stream(priority=hi_p), torch.cuda.Stream(priority=lo_p)
def run(a, b):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    torch.cuda.synchronize()
    ev[0].record(sa); sb.wait_event(ev[0])
    if a:
        A.render_device(pa, bufa.data_ptr(), 0, sa.cuda_stream, False)
    ev[1].record(sa)
    if b:
        B.render_device(pb, bufb.data_ptr(), 0, sb.cuda_stream, False)
    ev[2].record(sb)
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[2])
for _ in range(3):
    run(True, True)
for k in range(3):
    print("A alone: %.3f ms" % run(True, False)[0], " B alone: %.3f ms" % run(False, True)[1], " together: A ends %.3f ms, B ends %.3f ms" % run(True, True))
