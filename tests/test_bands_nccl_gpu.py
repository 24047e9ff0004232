"""The RCCL leg of the multi-GPU path on the one GPU a test box has: a world-size-1 `nccl` process group drives the same
BandRenderer / RowGatherer calls bench.py makes per frame (render into a device tensor on torch's stream, `dist.gather` into
views of the preallocated receive buffer, one index_copy). Multi-rank layouts are covered on CPU by tests/test_bands_gloo.py."""
import os
import socket

import numpy as np
import pytest

from rustracer_amd import Raytracer, flatten, presets
from rustracer_amd.bands import BandRenderer, RowGatherer

pytestmark = pytest.mark.gpu


def test_band_renderer_through_rccl_world1():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        w, h, spp, depth = 256, 150, 2, 5   # 150 rows: a ragged last band
        flat = flatten(presets.standalone_scene())
        rt = Raytracer.new(w, h, seed=3)
        rt.upload(flat)
        br = BandRenderer(rt, 0, 1, 8)
        want = rt.render(flat, spp, max_depth=depth, want_accum=False).rgba8
        # what a rank > 0 world would do, forced through the collective: band layout + gatherer with world = 1
        g = RowGatherer(h, 8, 0, 1, (w, 4), torch.uint8, torch.device("cuda", 0))
        for frame in range(3):
            br.render(spp, depth, 0, 3)
            full = g(br.rgba)
            dist.barrier()
            torch.cuda.synchronize()
            assert full.shape == (h, w, 4)
            assert np.array_equal(full.cpu().numpy(), want), f"frame {frame}"
        # the pipelined form bench.py uses: double-buffered frames, gather on a side stream overlapping the next render
        seeds = [11, 12, 13, 14, 15]
        wants = []
        for sd in seeds:
            rt.seed = sd
            wants.append(rt.render(flat, spp, max_depth=depth, want_accum=False).rgba8)
        br2 = BandRenderer(rt, 0, 1, 8)
        frames = []
        for sd in seeds:
            full = br2.step(spp, depth, 0, sd, force_collective=True)
            frames.append(full.clone())   # enqueued on the current stream: must not see the next frame's bytes...
            br2.wait()                    # ...which is what wait() orders
            frames[-1] = full.clone()
        torch.cuda.synchronize()
        for sd, got, want in zip(seeds, frames, wants):
            assert np.array_equal(got.cpu().numpy(), want), f"pipelined frame seed {sd}"
        # without intermediate waits the LAST frame must still be complete after wait()
        for sd in seeds:
            full = br2.step(spp, depth, 0, sd, force_collective=True)
        br2.wait()
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), wants[-1])
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.tolist() == [1.5, 2.5]
        rt.close()
    finally:
        dist.destroy_process_group()
