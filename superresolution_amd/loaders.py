"""Image-sequence loaders: host-side mirrors of helpers/LoadImSequence.m,
helpers/LoadImSequenceForward.m and helpers/LoadImSequenceReal.m.

They produce what the MultiframeMotionCoupling constructor takes
(`imageSequenceSmall`, rows x cols x colour x frames, double in [0,1]) and the
ground truth the test scripts compare with (testfile.m:31, :71-75).  Frames are
read with Pillow; arrays keep MATLAB's axis order, frame numbers are 1-based as
in the scripts that call these functions.
"""
import io
import os

import numpy as np

from . import sampling
from .mmc import rgb2ycbcr, ycbcr2rgb

# imresize's method names and the kernel behind each
_IMRESIZE = {"nearest": "nearest", "box": "nearest", "bilinear": "bilinear", "triangle": "bilinear",
             "bicubic": "bicubic", "cubic": "bicubic", "lanczos2": "lanczos2", "lanczos3": "lanczos3"}


def _listing(folder, order):
    """[dir('*.jpg'); dir('*.png'); dir('*.tif')]: one extension after the other, names sorted inside each."""
    names = sorted(os.listdir(folder))
    out = []
    for ext in order:
        out += [n for n in names if n.lower().endswith(ext)]
    if not out:
        raise ValueError("invalid location path, data not found")            # LoadImSequenceReal.m:7-9
    return out


def _im2double(path):
    """im2double(imread(path)): rows x cols (grey) or rows x cols x 3."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L"):
            return np.asarray(im, dtype=np.float64) / 65535.0
        if im.mode in ("1",):
            return np.asarray(im.convert("L"), dtype=np.float64) / 255.0
        if im.mode not in ("L", "RGB"):
            im = im.convert("RGB")
        return np.asarray(im, dtype=np.float64) / 255.0


def imresize(img, scale, method="bicubic", antialiasing=True):
    """imresize(img, scale, method, 'Antialiasing', antialiasing) along the first two axes; output
    size ceil(scale * size)."""
    if method not in _IMRESIZE:
        raise ValueError("unknown imresize method '%s'" % method)
    fn, width = sampling._KERNELS[_IMRESIZE[method]]
    img = np.asarray(img, dtype=np.float64)
    ny, nx = img.shape[:2]
    my = sampling._dense_rows_imresize(ny, int(np.ceil(ny * scale)), fn, float(width), antialiasing)
    mx = sampling._dense_rows_imresize(nx, int(np.ceil(nx * scale)), fn, float(width), antialiasing)
    out = np.einsum("ai,ij...->aj...", my, img)
    return np.einsum("bj,aj...->ab...", mx, out)


def _check_divisible(rows, cols, factor):
    if (rows / factor) % 1 or (cols / factor) % 1:                           # LoadImSequence.m:26-28
        raise ValueError("input image is not easily dividable by target factor, do something...")


def _load_large(folder, names, startFrame, numFrames):
    """Ground-truth frames as rows x cols x 3 x frames; grey frames are repeated over the three
    channels (what LoadImSequence.m:36-39 is there for)."""
    first = _im2double(os.path.join(folder, names[startFrame - 1]))
    rows, cols = first.shape[:2]
    large = np.zeros((rows, cols, 3, numFrames))
    for jj in range(startFrame, startFrame + numFrames):
        im = _im2double(os.path.join(folder, names[jj - 1]))
        if im.shape[:2] != (rows, cols):
            im = imresize(im, rows / float(im.shape[0]))                     # LoadImSequence.m:33
            if im.shape[:2] != (rows, cols):
                raise ValueError("frame %s has another aspect ratio than the first frame" % names[jj - 1])
        large[:, :, :, jj - startFrame] = im[:, :, None] if im.ndim == 2 else im
    return large


def LoadImSequence(datastr, startFrame, numFrames, factor=4, downSamplingMethod="bicubic", doJPEG=0):
    """helpers/LoadImSequence.m: frames startFrame .. startFrame+numFrames-1 of the folder as ground
    truth, and their imresize(., 1/factor, method, 'Antialiasing', true) as low-resolution input,
    clipped to [0,1] (:42-62).  doJPEG passes every low-resolution frame through an 8-bit JPEG of
    quality 100 (:49-52; in memory here, and the decoded values are those of this machine's JPEG
    library).  Returns (imageSequenceSmall, imageSequenceLarge)."""
    names = _listing(datastr, (".jpg", ".png", ".tif"))                      # :18
    large = _load_large(datastr, names, startFrame, numFrames)
    rows, cols = large.shape[:2]
    _check_divisible(rows, cols, factor)
    small = np.zeros((int(rows // factor), int(cols // factor), 3, numFrames))
    for jj in range(numFrames):
        tmp = imresize(large[:, :, :, jj], 1.0 / factor, downSamplingMethod, True)
        if doJPEG:
            from PIL import Image
            buf = io.BytesIO()
            u8 = np.clip(np.floor(tmp * 255.0 + 0.5), 0, 255).astype(np.uint8)
            Image.fromarray(u8, "RGB").save(buf, format="JPEG", quality=100, subsampling=0)
            buf.seek(0)
            tmp = np.asarray(Image.open(buf).convert("RGB"), dtype=np.float64) / 255.0
        small[:, :, :, jj] = tmp
    return np.clip(small, 0.0, 1.0), large                                   # :61-62


def LoadImSequenceForward(datastr, startFrame, numFrames, factor, interpMethod, kernel, Quantize=False):
    """helpers/LoadImSequenceForward.m (the 'inverse crime' data): the luma of every frame goes
    through the model's own operator D*B (samplingOperator without anti-aliasing times RepConvMtx,
    :34-35, :43), the chroma planes through imresize(., 1/factor) (:41-42), and the result back to
    RGB (:45).  Quantize rounds the low-resolution frames to 8 bits (:51-55)."""
    from scipy.ndimage import convolve
    names = _listing(datastr, (".jpg", ".png", ".tif"))                      # :8
    large = _load_large(datastr, names, startFrame, numFrames)
    rows, cols = large.shape[:2]
    _check_divisible(rows, cols, factor)
    ny, nx = int(rows // factor), int(cols // factor)
    Sy = sampling.dim1_matrix(rows, ny, interpMethod, None, False)
    Sx = sampling.dim1_matrix(cols, nx, interpMethod, None, False)
    k = np.asarray(kernel, dtype=np.float64)
    if k.ndim != 2 or k.shape[0] % 2 == 0 or k.shape[1] % 2 == 0:
        raise ValueError("only odd kernel sizes are supported")              # RepConvMtx.m:6-8
    small = np.zeros((ny, nx, 3, numFrames))
    for jj in range(numFrames):
        ycc = rgb2ycbcr(large[:, :, :, jj])                                  # :38
        blurred = convolve(ycc[:, :, 0], k, mode="nearest")                  # true convolution, replicate border
        y_s = Sy @ blurred @ Sx.T
        cb_s = imresize(ycc[:, :, 1], 1.0 / factor)
        cr_s = imresize(ycc[:, :, 2], 1.0 / factor)
        small[:, :, :, jj] = ycbcr2rgb(np.stack([y_s, cb_s, cr_s], axis=2))
    if Quantize:
        q = np.clip(np.floor(small * 255.0 + 0.5), 0, 255) / 255.0           # double(uint8(x*255))/255
        print("Quantization error: %g" % (np.abs(small - q).sum() / small.size))
        small = q
    return small, large


def LoadImSequenceReal(datastr, startFrame=1, numFrames=None):
    """helpers/LoadImSequenceReal.m: the frames as they are, rows x cols x 3 x frames for colour,
    rows x cols x frames for grey data (:24-31); no ground truth, nothing is resampled."""
    names = _listing(datastr, (".png", ".jpg", ".tif"))                      # :5
    if numFrames is None:
        numFrames = len(names)                                               # :11-13
    first = _im2double(os.path.join(datastr, names[startFrame - 1]))
    colour = first.ndim == 3
    out = np.zeros(first.shape + (numFrames,))
    for jj in range(startFrame, startFrame + numFrames):
        im = _im2double(os.path.join(datastr, names[jj - 1]))
        if im.shape != first.shape:
            raise ValueError("frame %s is %s, the first frame %s" % (names[jj - 1], im.shape, first.shape))
        if colour:
            out[:, :, :, jj - startFrame] = im
        else:
            out[:, :, jj - startFrame] = im
    return out
