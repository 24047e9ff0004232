"""The multi-GPU path on CPU: world_size-2 (and 3) `gloo` runs of the row-band scheduler. The render itself
needs a GPU, so each rank fabricates the rows it owns (every pixel carries its absolute (y, x) — what a renderer
keyed by the absolute pixel index produces); the test checks the partition, the padded gather and the
de-interleave on rank 0 for even and ragged band layouts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd.bands import RowGatherer, gather_rows, max_owned_rows, owned_row_indices  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cases, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for (h, w, band_rows) in cases:
        idx = owned_row_indices(h, band_rows, rank, world)
        local = torch.zeros((len(idx), w, 4), dtype=torch.uint8)
        ys = torch.from_numpy(idx)
        local[:, :, 0] = (ys % 251).to(torch.uint8)[:, None]
        local[:, :, 1] = (ys // 251).to(torch.uint8)[:, None]
        local[:, :, 2] = (torch.arange(w) % 256).to(torch.uint8)[None, :]
        local[:, :, 3] = 255
        full = gather_rows(local, h, band_rows, rank, world)
        if rank == 0:
            y = torch.arange(h)
            ok &= full.shape == (h, w, 4)
            ok &= bool((full[:, :, 0] == (y % 251).to(torch.uint8)[:, None]).all())
            ok &= bool((full[:, :, 1] == (y // 251).to(torch.uint8)[:, None]).all())
            ok &= bool((full[:, :, 2] == (torch.arange(w) % 256).to(torch.uint8)[None, :]).all())
            ok &= bool((full[:, :, 3] == 255).all())
        else:
            ok &= full is None
        # the per-frame form bench.py uses: buffers and the de-interleave index built once, reused for every frame
        g = RowGatherer(h, band_rows, rank, world, (w, 4), torch.uint8, torch.device("cpu"))
        for frame in range(3):
            again = g(local + frame)   # uint8 wrap-around is fine: the expected value wraps the same way
            if rank == 0:
                ok &= again.shape == (h, w, 4) and bool((again == full + frame).all())
            else:
                ok &= again is None
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t[0]) == float(world)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(out_path, "w") as f:
            f.write(str(int(flag[0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_gather_gloo(world, tmp_path):
    cases = [(2160 // 8, 64, 8), (37, 17, 8), (100, 5, 7), (5, 3, 8), (64, 8, 16)]
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), cases, str(out)), nprocs=world, join=True)
    assert out.read_text() == "1"


def test_partition_properties():
    for h, r, world in [(2160, 8, 8), (4320, 8, 8), (181, 16, 3), (7, 8, 4)]:
        rows = [owned_row_indices(h, r, k, world) for k in range(world)]
        assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(h))
        assert max(len(x) for x in rows) == max_owned_rows(h, r, world)
        # interleaving balances the load: no rank owns more than one band more than another
        assert max(len(x) for x in rows) - min(len(x) for x in rows) <= r
