"""PSNR / SSIM of the central patch of the central frame (testfile.m:71-75)."""
import numpy as np

from superresolution_amd import metrics


def test_patch_and_frame_selection():
    x = np.arange(60 * 50 * 5, dtype=np.float64).reshape(60, 50, 5)
    fr = metrics.central_frame(x)
    assert np.array_equal(fr, x[:, :, 2])                       # ceil(5/2) = 3rd frame
    assert np.array_equal(metrics.central_frame(x[:, :, :4]), x[:, :, 1])
    p = metrics.central_patch(fr)
    assert p.shape == (60 - 20 - 19, 50 - 20 - 19) and p[0, 0] == fr[19, 19] and p[-1, -1] == fr[39, 29]


def test_psnr_known_values():
    ref = np.zeros((8, 8))
    assert abs(metrics.psnr(ref + 0.1, ref) - 20.0) < 1e-12     # mse 0.01, peak 1
    assert metrics.psnr(ref, ref) == float("inf")


def test_ssim_properties():
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:64, 0:64]
    ref = 0.5 + 0.3 * np.sin(x / 5.0) * np.cos(y / 7.0)
    assert abs(metrics.ssim(ref, ref) - 1.0) < 1e-12
    s1 = metrics.ssim(ref + 0.02 * rng.normal(size=ref.shape), ref)
    s2 = metrics.ssim(ref + 0.10 * rng.normal(size=ref.shape), ref)
    assert 0 < s2 < s1 < 1
    assert abs(metrics.ssim(ref, ref + 0.05) - metrics.ssim(ref + 0.05, ref)) < 1e-12    # symmetric
    # constant images: the index reduces to the luminance term
    a, b = np.full((32, 32), 0.4), np.full((32, 32), 0.6)
    lum = (2 * 0.4 * 0.6 + 1e-4) / (0.16 + 0.36 + 1e-4)
    assert abs(metrics.ssim(a, b) - lum) < 1e-9
    rgb = np.stack([ref, ref * 0.9, ref * 0.8], axis=2)
    assert abs(metrics.ssim(rgb, rgb) - 1.0) < 1e-12


def test_error_margin_rounding():
    rng = np.random.default_rng(1)
    truth = rng.random((80, 70, 3))
    res = np.clip(truth + 0.01 * rng.normal(size=truth.shape), 0, 1)
    p, s = metrics.error_margin(res, truth)
    assert p == round(p, 2) and s == round(s, 3) and 35 < p < 45 and 0.9 < s <= 1.0
