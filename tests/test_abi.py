"""The C-ABI library loads and exports every symbol include/rthip.h declares; host-only entry points
behave (no compute calls here — those are in the gpu-marked tests)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rustracer_amd import abi, make_params, owned_rows
from rustracer_amd.bands import owned_row_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "rthip.h")) as f:
        header = f.read()
    declared = set(re.findall(r"\b(rthip_[a-z_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(abi.SYMBOLS), declared ^ set(abi.SYMBOLS)
    lib = abi.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rthip_abi_version() == abi.ABI_VERSION


def test_struct_sizes_match_header_layout():
    assert C.sizeof(abi.Material) == 48 and C.sizeof(abi.Sphere) == 40 and C.sizeof(abi.Plane) == 56
    assert C.sizeof(abi.Triangle) == 80 and C.sizeof(abi.Light) == 48 and C.sizeof(abi.Texture) == 16
    assert C.sizeof(abi.SceneDesc) == 96 and C.sizeof(abi.Camera) == 96 and C.sizeof(abi.RenderParams) == 64
    assert C.sizeof(abi.Stats) == 8 * 14 + 16 + 16 + 8  # ABI v2: + pixels_traced, pixels_final; v3: + gather_ms


def test_camera_lookat_matches_oracle(oracle_mod):
    lib = abi.lib()
    for pose, la in (((-1.8, 1.0, 2.0), (0.0, 0.0, -1.0)), ((3.0, 0.5, -2.0), (0.1, 0.2, 0.3)), ((0.0, 5.0, 0.001), (0.0, 0.0, 0.0))):
        out = abi.Camera()
        assert lib.rthip_camera_lookat(*pose, (C.c_double * 3)(*la), C.byref(out)) == 0
        ref = oracle_mod.camera_lookat(*pose, la)
        for f in ("origin", "d00", "horizontal", "vertical"):
            assert np.allclose(list(getattr(out, f)), list(getattr(ref, f)), rtol=1e-13, atol=1e-15), f
    assert lib.rthip_camera_lookat(0, 0, 0, None, None) == abi.E_BADARG


@pytest.mark.parametrize("h,r,world", [(2160, 8, 8), (2160, 16, 3), (1, 8, 2), (17, 8, 4), (600, 7, 5), (30, 64, 2)])
def test_owned_rows_partition(h, r, world):
    total = []
    for rank in range(world):
        p = make_params(64, max(h, 2), 1, band_rows=r, band_first=rank, band_stride=world)
        p.height = h
        n = owned_rows(p)
        idx = owned_row_indices(h, r, rank, world)
        assert n == len(idx)
        total.append(idx)
    allrows = np.sort(np.concatenate(total))
    assert np.array_equal(allrows, np.arange(h))  # every row owned exactly once


@pytest.mark.parametrize("h,r,world", [(2160, 8, 2), (2160, 8, 3), (2160, 8, 8), (181, 16, 3), (37, 7, 8), (5, 8, 8)])
def test_band_row_owner_is_the_inverse_of_the_band_split(h, r, world):
    """the row map of the multi-GPU de-interleave (rthip_group_render) against the band scheduler's own index lists"""
    lib = abi.lib()
    owner = {}
    for rank in range(world):
        for lrow, y in enumerate(owned_row_indices(h, r, rank, world)):
            owner[int(y)] = (rank, lrow)
    for y in range(h):
        rank, lrow = C.c_uint32(99), C.c_uint32(99)
        assert lib.rthip_band_row_owner(r, world, y, C.byref(rank), C.byref(lrow)) == 0
        assert (rank.value, lrow.value) == owner[y], y
    assert lib.rthip_band_row_owner(0, world, 0, None, None) == abi.E_BADARG


def test_status_strings_and_null_handling():
    lib = abi.lib()
    assert lib.rthip_status_string(0) == b"ok"
    assert b"argument" in lib.rthip_status_string(abi.E_BADARG)
    assert lib.rthip_device_count(None) == abi.E_BADARG
    assert lib.rthip_create(0, None) == abi.E_BADARG
    assert lib.rthip_render(None, None, None, None, None) == abi.E_BADARG
    assert lib.rthip_upload_scene(None, None) == abi.E_BADARG
    assert lib.rthip_render_device_noalloc(None, None, None, None, None, None) == abi.E_BADARG
    assert lib.rthip_group_render(None, None, None, None, None) == abi.E_BADARG and lib.rthip_group_size(None) == 0
    assert b"RCCL" in lib.rthip_status_string(abi.E_RCCL)
    lib.rthip_destroy(None)  # no-op
    n = C.c_int(-1)
    st = lib.rthip_device_count(C.byref(n))
    if st != 0:  # CPU-only container: a clean error, not a crash
        assert st in (abi.E_NODEVICE, abi.E_HIP) and n.value == 0
        ctx = C.c_void_p()
        assert lib.rthip_create(0, C.byref(ctx)) == abi.E_NODEVICE and not ctx.value
        assert lib.rthip_last_error(None)


def test_flops_model():
    lib = abi.lib()
    st = abi.Stats()
    st.paths, st.rays, st.sphere_tests, st.nb_samples = 10, 30, 120, 1
    st.hits[0], st.hits[1], st.hits[2], st.hits[3], st.sky_evals = 5, 4, 3, 2, 8
    want = 16 * 10 + 5 * 30 + 18 * 120 + 19 * 14 + 20 * 8 + 18 * 5 + 36 * 4 + 73 * 3 + 15 * 2 + 12 * 10
    assert lib.rthip_algorithmic_flops(C.byref(st)) == want
