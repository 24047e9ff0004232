"""RTHIP_F_ADAPTIVE: the progressive convergence rule of generate_pixel (lib.rs:276-313) on the device —
per-pixel same_color_count / Final status, Final pixels skipped through a compacted work-list."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from adaptive_rule import FINAL, NOT_STARTED, UNSTABLE, RuleEmulator  # noqa: E402
from parity import describe, mismatch_histogram  # noqa: E402

from rustracer_amd import Raytracer, abi, flatten, make_params, presets  # noqa: E402
from rustracer_amd.bands import owned_row_indices  # noqa: E402

pytestmark = pytest.mark.gpu


def _run_adaptive(flat, w, h, passes, spp, depth, seed, **kw):
    """adaptive passes on one context + plain accumulate passes on another; yields per-pass results"""
    rt = Raytracer.new(w, h, seed=seed)
    plain = Raytracer.new(w, h, seed=seed)
    out = []
    for k in range(passes):
        a = rt.render(flat, spp, max_depth=depth, accumulate=k > 0, adaptive=True, flags=abi.F_COUNT_STATS, **kw)
        status, count = rt.pixel_status()
        b = plain.render(flat, spp, max_depth=depth, accumulate=k > 0, **kw)
        out.append((a, status.copy(), count.copy(), b))
    rt.close()
    plain.close()
    return out


@pytest.mark.parametrize("preset,w,h,passes,spp,depth", [
    ("standalone_scene", 320, 180, 14, 1, 50),   # the standalone window: progressive, spp = 1 (standalone/src/main.rs:99-102)
    ("standalone_scene", 200, 120, 10, 4, 5),
    ("mesh_small", 192, 108, 9, 2, 8),           # BVH + work pool + work-list
    ("standalone_scene", 96, 54, 7, 130, 5),     # sample slicing (3 groups: 64 + 64 + 2) + resolve kernel + work-list
    ("mesh_small", 96, 54, 6, 70, 4),            # the same through the BVH kernel
])
def test_adaptive_rule_matches_reference_state_machine(preset, w, h, passes, spp, depth):
    flat = flatten(presets.mesh_scene(6, 2) if preset == "mesh_small" else getattr(presets, preset)())
    emu = RuleEmulator(h, w)
    frozen_accum = np.zeros((h, w, 3), np.float32)
    n_final_seen = 0
    for k, (a, status, count, b) in enumerate(_run_adaptive(flat, w, h, passes, spp, depth, seed=5)):
        live_before = emu.live()
        emu.step(a.rgba8[..., :3])
        # status and same_color_count are exactly what lib.rs:300-311 gives for the colours the device produced
        assert np.array_equal(status, emu.status), f"pass {k}: status differs on {(status != emu.status).sum()} pixels"
        assert np.array_equal(count, emu.count), f"pass {k}: same_color_count differs"
        assert a.rgba8[..., 3].min() == 255
        # a live pixel gets exactly the samples of the plain progressive render (same running sum, same colour)
        assert np.array_equal(a.accum[live_before], b.accum[live_before])
        assert np.array_equal(a.rgba8[live_before], b.rgba8[live_before])
        # a Final pixel is frozen: colour and running sum are those of the pass that froze it (lib.rs:277-279)
        dead = ~live_before
        assert np.array_equal(a.accum[dead], frozen_accum[dead])
        assert np.array_equal(a.rgba8[..., :3][dead], emu.last[dead])
        frozen_accum[live_before] = a.accum[live_before]
        # counters: only live pixels are traced
        assert a.stats["pixels_traced"] == int(live_before.sum())
        assert a.stats["paths"] == int(live_before.sum()) * spp
        assert a.stats["pixels_final"] == int((emu.status == FINAL).sum())
        assert a.stats["nb_samples"] == (k + 1) * spp
        n_final_seen = a.stats["pixels_final"]
        if k < 4:
            assert n_final_seen == 0  # Final needs 5 equal passes in a row (same_color_count > 3)
    print(f"\n{preset} {w}x{h} {passes} passes x {spp} spp: {n_final_seen} of {w * h} pixels Final")
    assert n_final_seen > 0


def test_adaptive_vs_oracle(oracle_mod, default_camera, standalone_flat):
    """the same passes through the CPU oracle's PixelCache: image parity on the usual bar; the Final masks agree except
    where an 8-bit colour sits on a rounding boundary between FP32 and f64 (reported)"""
    w, h, passes, spp, depth, seed = 240, 135, 12, 1, 50, 9
    rt = Raytracer.new(w, h, seed=seed)
    state = oracle_mod.AdaptiveState(w, h)
    for k in range(passes):
        res = rt.render(standalone_flat, spp, max_depth=depth, accumulate=k > 0, adaptive=True)
        p = make_params(w, h, spp, depth, seed, abi.F_ADAPTIVE | (abi.F_ACCUMULATE if k else 0))
        st = oracle_mod.render_adaptive(standalone_flat, default_camera, p, state)
    status, _ = rt.pixel_status()
    hist, ok = mismatch_histogram(res.rgba8, state.rgba)
    agree = float((status == state.status).mean())
    print(f"\nadaptive {passes}x{spp}spp vs oracle f64: {describe(hist, ok)}; status agreement {agree * 100:.2f} %, "
          f"Final device {int((status == FINAL).sum())} oracle {int((state.status == FINAL).sum())} (oracle stats {st['pixels_final']})")
    assert ok >= 0.999
    assert agree >= 0.99
    assert st["pixels_final"] == int((state.status == FINAL).sum())
    rt.close()


def test_adaptive_band_split_is_identical(standalone_flat):
    w, h, passes, spp, depth, seed, band_rows = 160, 100, 8, 1, 6, 3, 8
    full = Raytracer.new(w, h, seed=seed)
    parts = [Raytracer.new(w, h, seed=seed) for _ in range(3)]
    for k in range(passes):
        f = full.render(standalone_flat, spp, max_depth=depth, accumulate=k > 0, adaptive=True)
        img = np.zeros_like(f.rgba8)
        for r, rt in enumerate(parts):
            pr = rt.render(standalone_flat, spp, max_depth=depth, accumulate=k > 0, adaptive=True, band_rows=band_rows, band_first=r, band_stride=3)
            img[owned_row_indices(h, band_rows, r, 3)] = pr.rgba8
        assert np.array_equal(img, f.rgba8), f"pass {k}"
    fs, _ = full.pixel_status()
    st = np.zeros_like(fs)
    for r, rt in enumerate(parts):
        st[owned_row_indices(h, band_rows, r, 3)] = rt.pixel_status()[0]
    assert np.array_equal(st, fs)
    for rt in parts + [full]:
        rt.close()


def test_adaptive_state_errors_and_reset(standalone_flat):
    rt = Raytracer.new(64, 36, seed=1)
    with pytest.raises(abi.RthipError) as e:
        rt.render(standalone_flat, 1, max_depth=3)
        rt.pixel_status()
    assert e.value.status == abi.E_STATE
    rt.render(standalone_flat, 1, max_depth=3, adaptive=True)
    with pytest.raises(abi.RthipError) as e:   # a running sum is adaptive on every pass or on none
        rt.render(standalone_flat, 1, max_depth=3, accumulate=True)
    assert e.value.status == abi.E_STATE
    rt.render(standalone_flat, 1, max_depth=3, accumulate=True, adaptive=True)
    status, count = rt.pixel_status()
    assert set(np.unique(status)) <= {UNSTABLE} and count.max() <= 1
    rt.invalidate_pixels()   # camera command: every pixel starts over (standalone/src/main.rs:103-120)
    rt.render(standalone_flat, 1, max_depth=3, accumulate=True, adaptive=True)
    status, count = rt.pixel_status()
    assert np.all(status == UNSTABLE) and np.all(count == 0)
    assert NOT_STARTED == abi.STATUS_NOT_STARTED and FINAL == abi.STATUS_FINAL
    rt.close()
