"""How long does rank 0's de-interleave of a gathered 4K RGBA8 frame take (RowGatherer's index_copy), as bytes and through
int32 views? python tools/gather_copy_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustracer_amd.bands import RowGatherer

H, W = 2160, 3840
for world in (2, 8):
    g = RowGatherer(H, 8, 0, world, (W, 4), torch.uint8, torch.device("cuda"))
    src = g.recv.view((world * g.pad_rows, W, 4))
    src.random_(0, 255)
    def run(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_u8 = run(lambda: g.full.index_copy_(0, g.index, src))
    full32, src32 = g.full.view(torch.int32), src.view(torch.int32)
    ref = g.full.clone()
    t_i32 = run(lambda: full32.index_copy_(0, g.index, src32))
    assert torch.equal(ref, g.full)
    print(f"world {world}: index_copy of the gathered frame: uint8 {t_u8 * 1e3:.1f} us, int32 view {t_i32 * 1e3:.1f} us")
