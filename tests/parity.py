"""Parity metric of BASELINE.json: per-channel |delta| <= 1/255 on >= 99.9 % of pixels, with the
mismatch histogram (count of pixels by max-over-channels |delta| in 8-bit steps)."""
import numpy as np

TOL_LSB = 1          # |delta| <= 1/255
MIN_FRACTION = 0.999  # on >= 99.9 % of pixels


def mismatch_histogram(a_rgba8, b_rgba8):
    a = np.asarray(a_rgba8)[..., :3].astype(np.int32)
    b = np.asarray(b_rgba8)[..., :3].astype(np.int32)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a - b).max(axis=-1).reshape(-1)
    hist = np.bincount(d, minlength=256)
    frac_ok = float((d <= TOL_LSB).mean()) if d.size else 1.0
    return hist, frac_ok


def describe(hist, frac_ok):
    nz = {int(k): int(v) for k, v in enumerate(hist) if v}
    return f"within 1/255: {frac_ok * 100:.4f} %  histogram(|delta| -> pixels): {nz}"


def assert_parity(a_rgba8, b_rgba8, what=""):
    hist, frac_ok = mismatch_histogram(a_rgba8, b_rgba8)
    assert frac_ok >= MIN_FRACTION, f"{what}: {describe(hist, frac_ok)}"
    return hist, frac_ok
