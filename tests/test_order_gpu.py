"""Cost-ordered launches (rthip_api.cpp render_pass, rt_kernels.hip tile_cost_kernel / tile_order_kernel): a launch of a scan scene
that is small enough to be sliced takes its pixel groups in order of cost — pass 1 of a frame layout from a cost model (and
measures the groups), pass 2 from those measurements, pass 3.. reuse that order. A schedule only: every pass must give the
bytes of the launch in row order (rthip_render_device_noalloc never orders). Reference: the image is a pure function of
(seed, scene, camera, size, spp) — lib.rs:269-316 per pixel."""
import ctypes as C

import numpy as np
import pytest
import torch

from rustracer_amd import Raytracer, abi, flatten, make_params, presets

pytestmark = pytest.mark.gpu


def _pass(lib, ctx, entry, p):
    rows = C.c_uint32(0)
    assert lib.rthip_owned_rows(C.byref(p), C.byref(rows)) == 0
    rgba = torch.zeros((rows.value, p.width, 4), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((rows.value, p.width, 3), dtype=torch.float32, device="cuda")
    st = abi.Stats()
    rc = entry(ctx, C.byref(p), rgba.data_ptr(), acc.data_ptr(), torch.cuda.current_stream().cuda_stream, C.byref(st))
    assert rc == 0, lib.rthip_last_error(ctx)
    torch.cuda.synchronize()
    return rgba.cpu().numpy(), acc.cpu().numpy().view(np.uint32), st.as_dict()


@pytest.mark.parametrize("case", ["sliced", "unsliced", "band", "tree", "tree-band"])
def test_every_state_of_the_launch_order_gives_the_row_order_bytes(case):
    lib = abi.lib()
    tree = case.startswith("tree")
    # tree scenes (401 spheres + 320 triangles): pass 1 runs sliced in row order and measures, pass 2.. run UNSLICED, longest waves first
    flat = flatten(presets.big_scene(n_side=20, seed=3, mesh_subdivisions=2)) if tree else flatten(presets.standalone_scene())
    w, h = 960, 544  # 30 x 68 = 2040 pixel groups: ordered (>= 64, < 20 000)
    p = {"sliced": make_params(w, h, 130, 5, 77, abi.F_COUNT_STATS),                                   # 3 slices
         "unsliced": make_params(w, h, 24, 50, 78, 0),                                                 # one slice, no counters
         "band": make_params(w, h, 70, 5, 79, 0, band_rows=8, band_first=1, band_stride=2),            # a rank's bands
         "tree": make_params(w, h, 16, 8, 80, 0),
         "tree-band": make_params(w, h, 13, 50, 81, abi.F_COUNT_STATS, band_rows=16, band_first=2, band_stride=3)}[case]
    ordered, plain = Raytracer.new(w, h, seed=p.seed), Raytracer.new(w, h, seed=p.seed)
    for rt in (ordered, plain):
        rt.upload(flat)
        rt._sync_scene_camera(None)
    want = _pass(lib, plain._ctx, lib.rthip_render_device_noalloc, p)
    for k in range(4):  # model order + measuring, measured order, reuse, reuse
        got = _pass(lib, ordered._ctx, lib.rthip_render_device, p)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), f"pass {k + 1} differs from the row-order launch"
        assert got[2]["rays"] == want[2]["rays"]
    ordered.close(), plain.close()
