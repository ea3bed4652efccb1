"""Host-side separable resampling with MATLAB imresize semantics.

The reference computes its start point with `imresize(LR, factor)` (bicubic,
MultiframeMotionCoupling.m:415-419) and its chroma planes the same way
(:1027-1028).  This module provides that resampling for the MATLAB-free host:
output sample x (1-based) sits at x/scale + (1 - 1/scale)/2, a Keys cubic
(a = -1/2) of support 4 (stretched by 1/scale when shrinking with
anti-aliasing) weights the neighbours, the weights are normalised and indices
beyond the edge replicate the edge pixel.
"""
import numpy as np


def _keys(t):
    t = np.abs(t)
    t2 = t * t
    t3 = t2 * t
    near = 1.5 * t3 - 2.5 * t2 + 1.0
    far = -0.5 * t3 + 2.5 * t2 - 4.0 * t + 2.0
    return np.where(t <= 1.0, near, np.where(t <= 2.0, far, 0.0))


def resize_matrix(n_in, n_out, scale, antialias=True):
    """(n_out, n_in) matrix of bicubic resampling along one axis."""
    support = 4.0
    stretch = 1.0
    if scale < 1.0 and antialias:
        stretch = scale
        support = support / scale
    centres = np.arange(1, n_out + 1, dtype=np.float64) / scale + 0.5 * (1.0 - 1.0 / scale)
    first = np.floor(centres - support / 2.0)
    ntaps = int(np.ceil(support)) + 2
    mat = np.zeros((n_out, n_in), dtype=np.float64)
    rows = np.arange(n_out)
    wsum = np.zeros(n_out)
    taps = []
    for j in range(ntaps):
        pos = first + j
        wj = stretch * _keys(stretch * (centres - pos))
        wsum += wj
        taps.append((np.clip(pos, 1, n_in).astype(np.int64) - 1, wj))
    for idx, wj in taps:
        np.add.at(mat, (rows, idx), wj / wsum)
    return mat


def imresize_bicubic(img, scale, antialias=True):
    """Resample the first two axes of `img` by `scale`; output size ceil(scale*size)."""
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape[0], img.shape[1]
    my = resize_matrix(ny, int(np.ceil(ny * scale)), scale, antialias)
    mx = resize_matrix(nx, int(np.ceil(nx * scale)), scale, antialias)
    out = np.einsum("ai,ij...->aj...", my, img)
    out = np.einsum("bj,aj...->ab...", mx, out)
    return out


def upsample_sequence(frames, factor):
    """Bicubic estimate u_up of a (ny, nx, nc) stack, MultiframeMotionCoupling.m:415-419."""
    frames = np.asarray(frames, dtype=np.float64)
    ny, nx = frames.shape[:2]
    my = resize_matrix(ny, ny * factor, float(factor))
    mx = resize_matrix(nx, nx * factor, float(factor))
    return np.einsum("ai,ijt,bj->abt", my, frames, mx, optimize=True)
