"""Multi-GPU row-band scheduler: one process per GPU, `torch.distributed` over
RCCL/xGMI. The image shards naturally — pixels are independent given a read-only scene replicated
on every GPU — so the only exchange is the final framebuffer gather to rank 0.

Partition: band b = rows [b*R, (b+1)*R), owned by rank b mod world (INTERLEAVED, so the ranks that
would own only sky rows — 1 segment per path — share the rows that cross glass/metal spheres). The
RNG is keyed by the absolute pixel index, hence the gathered image is byte-identical for any world
size. Collective: one `gather` of each rank's owned rows (padded to the longest rank) to rank 0,
then one index_copy on rank 0 to de-interleave. Per-link payload at 4K RGBA8 / 8 ranks is 4.1 MB
(~27 us at 153 GB/s), i.e. invisible next to the render.
"""
from __future__ import annotations

import numpy as np

DEFAULT_BAND_ROWS = 8


def owned_row_indices(height: int, band_rows: int, rank: int, world: int) -> np.ndarray:
    """Image rows (y, increasing) owned by `rank`; matches rthip_owned_rows + the kernel's row map."""
    if band_rows <= 0:
        return np.arange(height, dtype=np.int64) if rank == 0 else np.zeros(0, np.int64)
    nbands = (height + band_rows - 1) // band_rows
    rows = []
    for b in range(rank, nbands, world):
        rows.append(np.arange(b * band_rows, min(height, (b + 1) * band_rows), dtype=np.int64))
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def max_owned_rows(height: int, band_rows: int, world: int) -> int:
    return max(len(owned_row_indices(height, band_rows, r, world)) for r in range(world))


class RowGatherer:
    """The framebuffer exchange of one (height, band_rows, world) layout with everything that does not change
    between frames built once: the receive buffer of `dst` and ONE destination-row index that de-interleaves all
    ranks' rows with a single index_copy (padding rows of the shorter ranks land in scratch rows past the image),
    so a frame costs one collective + one copy kernel and no host->device transfer."""

    def __init__(self, height: int, band_rows: int, rank: int, world: int, trailing_shape, dtype, device, dst: int = 0):
        import torch

        self.height, self.band_rows, self.rank, self.world, self.dst = height, band_rows, rank, world, dst
        self.torch = torch
        self.pad_rows = max_owned_rows(height, band_rows, world)
        self.trailing, self.dtype, self.device = tuple(trailing_shape), dtype, device
        self.own_rows = len(owned_row_indices(height, band_rows, rank, world))
        self.padded = None if self.own_rows == self.pad_rows else torch.zeros((self.pad_rows,) + self.trailing, dtype=dtype, device=device)
        if rank == dst:
            self.recv = torch.empty((world, self.pad_rows) + self.trailing, dtype=dtype, device=device)
            self.parts = [self.recv[r] for r in range(world)]
            index = np.empty(world * self.pad_rows, np.int64)
            scratch = height
            for r in range(world):
                idx = owned_row_indices(height, band_rows, r, world)
                index[r * self.pad_rows: r * self.pad_rows + len(idx)] = idx
                n_pad = self.pad_rows - len(idx)
                index[r * self.pad_rows + len(idx): (r + 1) * self.pad_rows] = np.arange(scratch, scratch + n_pad)
                scratch += n_pad
            self.n_scratch = scratch - height
            self.index = torch.from_numpy(index).to(device)
            self.full = torch.empty((height + self.n_scratch,) + self.trailing, dtype=dtype, device=device)

    def __call__(self, local, group=None):
        import torch.distributed as dist

        if local.device != self.device:
            local = local.to(self.device)  # host-side exchange (gloo has no GPU gather): synchronous copy on the current stream
        if self.padded is not None:
            self.padded[: self.own_rows].copy_(local)
            send = self.padded
        else:
            send = local.contiguous()
        if self.rank == self.dst:
            dist.gather(send, gather_list=self.parts, dst=self.dst, group=group)
            src = self.recv.view((self.world * self.pad_rows,) + self.trailing)
            if self.dtype == self.torch.uint8 and self.trailing and self.trailing[-1] == 4:
                # RGBA8 rows as int32 pixels: the same copy with a quarter of the elements (4K frame on MI355X: 18 us instead of 57,
                # tools/gather_copy_bench.py) — it runs beside the next frame's render on rank 0
                self.full.view(self.torch.int32).index_copy_(0, self.index, src.view(self.torch.int32))
            else:
                self.full.index_copy_(0, self.index, src)
            return self.full[: self.height]
        dist.gather(send, gather_list=None, dst=self.dst, group=group)
        return None


def gather_rows(local, height: int, band_rows: int, rank: int, world: int, group=None, dst: int = 0):
    """Gather every rank's owned rows (tensor (rows_r, ...), same trailing shape and dtype on all ranks)
    to `dst` and return the de-interleaved (height, ...) tensor there (None elsewhere). One-shot form of RowGatherer."""
    if world == 1:
        return local
    return RowGatherer(height, band_rows, rank, world, local.shape[1:], local.dtype, local.device, dst)(local, group)


class BandRenderer:
    """Per-rank driver: renders this rank's bands on its GPU into device tensors and gathers the
    RGBA8 image to rank 0."""

    def __init__(self, raytracer, rank: int, world: int, band_rows: int = DEFAULT_BAND_ROWS, gather_on_host: bool = False):
        """gather_on_host: exchange the rows through host memory (for process groups whose backend cannot gather GPU tensors,
        i.e. gloo); the frame then arrives on rank 0 as a CPU tensor."""
        import torch

        from .raytracer import make_params, owned_rows
        self.rt, self.rank, self.world, self.band_rows = raytracer, rank, world, band_rows
        self._make_params, self._owned_rows = make_params, owned_rows
        self.width, self.height = int(raytracer.width), int(raytracer.height)
        self.torch = torch
        p = self.params(1, 1, 0)
        self.rows = owned_rows(p)
        assert self.rows == len(owned_row_indices(self.height, band_rows if world > 1 else 0, rank, world))
        dev = torch.device("cuda", torch.cuda.current_device())
        # two frame buffers: the gather of frame k (on a side stream) overlaps the render of frame k + 1
        self._bufs = [torch.empty((self.rows, self.width, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.rgba = self._bufs[0]
        self._gather_dev = torch.device("cpu") if gather_on_host else dev
        self._gatherer = None if world == 1 else RowGatherer(self.height, band_rows, rank, world, (self.width, 4), torch.uint8, self._gather_dev)
        self._gather_stream = None if world == 1 else torch.cuda.Stream(device=dev)
        self._gathered = [None, None]  # per buffer: event "the gather that read it has finished"
        self._frame = 0

    def params(self, spp, max_depth, flags, seed=None):
        if self.world == 1:
            return self._make_params(self.width, self.height, spp, max_depth, self.rt.seed if seed is None else seed, flags)
        return self._make_params(self.width, self.height, spp, max_depth, self.rt.seed if seed is None else seed, flags,
                                 self.band_rows, self.rank, self.world)

    def render(self, spp, max_depth=50, flags=0, seed=None, want_stats=False):
        """enqueue one pass on torch's current stream; returns stats dict if requested"""
        stream = self.torch.cuda.current_stream().cuda_stream
        return self.rt.render_device(self.params(spp, max_depth, flags, seed), self.rgba.data_ptr(), 0, stream, want_stats)

    def step(self, spp, max_depth=50, flags=0, seed=None, group=None, force_collective=False, events=None):
        """One frame, pipelined: render this rank's bands on torch's current stream into one of two buffers and start the
        gather to rank 0 on a side stream, so that it overlaps the next frame's render. Returns the frame on rank 0 (None
        elsewhere); call wait() before reading it. force_collective: run the gather also when world == 1 (tests).
        events: (start, end) torch events recorded on the render stream right around the render launch — after the wait for
        the gather that last read the frame buffer — so that their distance is the kernel time alone."""
        torch = self.torch
        k = self._frame & 1
        self._frame += 1
        buf = self._bufs[k]
        main = torch.cuda.current_stream()
        if self._gathered[k] is not None:
            main.wait_event(self._gathered[k])  # the gather of two frames ago has finished reading this buffer
        if events is not None:
            events[0].record(main)
        self.rt.render_device(self.params(spp, max_depth, flags, seed), buf.data_ptr(), 0, main.cuda_stream, False)
        if events is not None:
            events[1].record(main)
        self.rgba = buf
        if self._gatherer is None and not force_collective:
            return buf
        if self._gatherer is None:
            self._gatherer = RowGatherer(self.height, self.band_rows, self.rank, self.world, (self.width, 4), torch.uint8, self._gather_dev)
            self._gather_stream = torch.cuda.Stream(device=buf.device)
        rendered = torch.cuda.Event()
        rendered.record(main)
        with torch.cuda.stream(self._gather_stream):
            self._gather_stream.wait_event(rendered)
            full = self._gatherer(buf, group)
            done = torch.cuda.Event()
            done.record(self._gather_stream)
        self._gathered[k] = done
        return full

    def wait(self):
        """make torch's current stream wait for the gathers started by step()"""
        if self._gather_stream is not None:
            self.torch.cuda.current_stream().wait_stream(self._gather_stream)

    def gather(self, group=None):
        """(unpipelined) the frame on rank 0 (a view of a buffer that the next gather overwrites), None elsewhere"""
        return self.rgba if self._gatherer is None else self._gatherer(self.rgba, group)
