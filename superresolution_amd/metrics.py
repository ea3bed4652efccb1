"""Error margin of a result as the reference's demo scripts report it
(testfile.m:71-75, testfile_full_set.m): PSNR and SSIM of the central frame on
the patch 20:end-20 (MATLAB 1-based, inclusive), for doubles in [0, 1].

psnr follows MATLAB's psnr for floating-point images (peak value 1).  ssim
follows the published definition that MATLAB's ssim implements: Gaussian
weights of standard deviation 1.5 on a window of radius ceil(3*1.5) = 5 with
replicated borders, C1 = (0.01 L)^2, C2 = (0.03 L)^2, L = 1, mean of the
local index over all samples; an array with a colour axis is filtered as a
volume, as MATLAB does for N-D input.  No reference outputs exist for either
number (no golden values in the reference), so they are checked by properties.
"""
import numpy as np
from scipy import ndimage


def central_patch(x, margin=20):
    """x(20:end-20, 20:end-20, ...) of testfile.m:71."""
    x = np.asarray(x)
    return x[margin - 1:x.shape[0] - margin, margin - 1:x.shape[1] - margin]


def central_frame(seq):
    """Frame ceil(numFrames/2) (1-based) of a (rows, cols[, 3], frames) stack."""
    seq = np.asarray(seq)
    return seq[..., int(np.ceil(seq.shape[-1] / 2.0)) - 1]


def psnr(a, ref, peak=1.0):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    mse = np.mean((a - ref) ** 2)
    return float("inf") if mse == 0 else float(10.0 * np.log10(peak * peak / mse))


def ssim(a, ref, dynamic_range=1.0):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    sigma, radius = 1.5, 5
    filt = lambda x: ndimage.gaussian_filter(x, sigma, mode="nearest", truncate=radius / sigma)
    c1, c2 = (0.01 * dynamic_range) ** 2, (0.03 * dynamic_range) ** 2
    mua, mur = filt(a), filt(ref)
    va = filt(a * a) - mua * mua
    vr = filt(ref * ref) - mur * mur
    cov = filt(a * ref) - mua * mur
    idx = ((2 * mua * mur + c1) * (2 * cov + c2)) / ((mua * mua + mur * mur + c1) * (va + vr + c2))
    return float(idx.mean())


def error_margin(result, truth, margin=20):
    """(psnr rounded to 2 decimals, ssim rounded to 3) of testfile.m:71-75."""
    out = central_patch(central_frame(result), margin)
    ref = central_patch(central_frame(truth), margin)
    return round(psnr(out, ref), 2), round(ssim(out, ref), 3)
