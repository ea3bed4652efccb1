"""Flow fields and occlusion maps from files, in the layout the solve takes.

The reference's scripts that bring their own motion read Middlebury `.flo` files with
`readFlowFile` (variants/testfile_gt_sintel.m:33-39, variants/testfile_epicFlow.m:59-66) and
occlusion maps as images (testfile_gt_sintel.m:42-46); the arrays go into the constructor as
'flowField' (Ny x Nx x (nf-1) x 2, component 1 = displacement along columns, component 2 = along
rows) and, here, 'occlusion' (the library takes the maps themselves where the reference folds them
into a prebuilt sparse 'warpOp', testfile_gt_sintel.m:51-58, :65).

A `.flo` file: the float 202021.25 (bytes 'PIEH'), int32 width, int32 height, then height rows of
width (u, v) float32 pairs, little-endian; u is the horizontal, v the vertical displacement.
"""
import os

import numpy as np

from .loaders import imresize

FLO_TAG = 202021.25


def readFlowFile(filename):
    """rows x cols x 2 double: [:, :, 0] horizontal (u), [:, :, 1] vertical (v) displacement."""
    if not str(filename).lower().endswith(".flo"):
        raise ValueError("readFlowFile: filename %s should have extension '.flo'" % filename)
    with open(filename, "rb") as fh:
        head = fh.read(12)
        if len(head) < 12:
            raise ValueError("readFlowFile: %s is too short for a header" % filename)
        tag = np.frombuffer(head, dtype="<f4", count=1)[0]
        width, height = (int(x) for x in np.frombuffer(head, dtype="<i4", count=2, offset=4))
        if tag != np.float32(FLO_TAG):
            raise ValueError("readFlowFile(%s): wrong tag (possibly due to big-endian machine?)" % filename)
        if width < 1 or width > 99999:
            raise ValueError("readFlowFile(%s): illegal width %d" % (filename, width))
        if height < 1 or height > 99999:
            raise ValueError("readFlowFile(%s): illegal height %d" % (filename, height))
        data = np.frombuffer(fh.read(), dtype="<f4")
    if data.size != 2 * width * height:
        raise ValueError("readFlowFile(%s): expected %d values, file holds %d" % (filename, 2 * width * height, data.size))
    return data.reshape(height, width, 2).astype(np.float64)


def writeFlowFile(img, filename):
    """The inverse of readFlowFile for a rows x cols x 2 array."""
    img = np.asarray(img)
    if not str(filename).lower().endswith(".flo"):
        raise ValueError("writeFlowFile: filename %s should have extension '.flo'" % filename)
    if img.ndim != 3 or img.shape[2] != 2:
        raise ValueError("writeFlowFile: image must have two bands")
    height, width = img.shape[:2]
    with open(filename, "wb") as fh:
        fh.write(np.array([FLO_TAG], dtype="<f4").tobytes())
        fh.write(np.array([width, height], dtype="<i4").tobytes())
        fh.write(np.ascontiguousarray(img, dtype="<f4").tobytes())


def load_flow_sequence(folder, numPairs, pattern="frame_%04d.flo", start=1):
    """v(:,:,i,1) = u, v(:,:,i,2) = v of the files pattern % (start + i), i = 0 .. numPairs-1
    (testfile_gt_sintel.m:33-39): Ny x Nx x numPairs x 2."""
    v = None
    for i in range(numPairs):
        f = readFlowFile(os.path.join(folder, pattern % (start + i)))
        if v is None:
            v = np.zeros(f.shape[:2] + (numPairs, 2))
        elif f.shape[:2] != v.shape[:2]:
            raise ValueError("flow file %d is %s, the first one %s" % (start + i, f.shape[:2], v.shape[:2]))
        v[:, :, i, :] = f
    return v


def load_occlusion_sequence(folder, numPairs, pattern="frame_%04d.png", start=1):
    """logical(im2double(imread(.))) of one map per pair (testfile_gt_sintel.m:42-46): every non-zero
    pixel is occluded.  Ny x Nx x numPairs bool."""
    from PIL import Image
    occ = None
    for i in range(numPairs):
        with Image.open(os.path.join(folder, pattern % (start + i))) as im:
            m = np.asarray(im)
        if m.ndim == 3:
            raise ValueError("occlusion map %d has %d channels; a single-channel image is expected" % (start + i, m.shape[2]))
        if occ is None:
            occ = np.zeros(m.shape + (numPairs,), dtype=bool)
        elif m.shape != occ.shape[:2]:
            raise ValueError("occlusion map %d is %s, the first one %s" % (start + i, m.shape, occ.shape[:2]))
        occ[:, :, i] = m != 0
    return occ


def upscale_flow(v_small, factor):
    """A flow estimated on the low-resolution frames as a flow between high-resolution frames:
    imresize(v, factor) * factor per component (testfile_epicFlow.m:60-61, MultiframeMotionCoupling.m:958-959).
    Accepts rows x cols x 2 or rows x cols x pairs x 2."""
    v_small = np.asarray(v_small, dtype=np.float64)
    return imresize(v_small, float(factor)) * float(factor)
