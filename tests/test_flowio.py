"""Flow (.flo) and occlusion-map readers (superresolution_amd/flowio.py): the file layout, the array
layout of variants/testfile_gt_sintel.m:33-46, the errors of the Middlebury reader, the up-scaling of
variants/testfile_epicFlow.m:60-61."""
import os
import struct

import numpy as np
import pytest

import mmc_oracle as orc
from superresolution_amd import flowio

PIL = pytest.importorskip("PIL.Image")


def test_flo_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    f = rng.standard_normal((5, 7, 2)).astype(np.float32)
    path = str(tmp_path / "a.flo")
    flowio.writeFlowFile(f, path)
    raw = open(path, "rb").read()
    assert raw[:4] == b"PIEH" and struct.unpack("<ii", raw[4:12]) == (7, 5)           # tag, width, height
    assert len(raw) == 12 + 5 * 7 * 2 * 4
    # row-major, (u, v) interleaved: second pixel of the first row
    assert struct.unpack("<ff", raw[20:28]) == (float(f[0, 1, 0]), float(f[0, 1, 1]))
    back = flowio.readFlowFile(path)
    assert back.dtype == np.float64 and np.array_equal(back, f.astype(np.float64))


def test_flo_errors(tmp_path):
    good = str(tmp_path / "g.flo")
    flowio.writeFlowFile(np.zeros((2, 3, 2)), good)
    raw = open(good, "rb").read()
    with pytest.raises(ValueError, match="extension"):
        flowio.readFlowFile(str(tmp_path / "g.bin"))
    bad = str(tmp_path / "tag.flo")
    open(bad, "wb").write(struct.pack(">f", flowio.FLO_TAG) + raw[4:])                 # big-endian tag
    with pytest.raises(ValueError, match="wrong tag"):
        flowio.readFlowFile(bad)
    open(bad, "wb").write(raw[:4] + struct.pack("<ii", 0, 2) + raw[12:])
    with pytest.raises(ValueError, match="illegal width"):
        flowio.readFlowFile(bad)
    open(bad, "wb").write(raw[:-4])
    with pytest.raises(ValueError, match="expected 12 values"):
        flowio.readFlowFile(bad)
    open(bad, "wb").write(raw[:6])
    with pytest.raises(ValueError, match="too short"):
        flowio.readFlowFile(bad)
    with pytest.raises(ValueError, match="two bands"):
        flowio.writeFlowFile(np.zeros((2, 3)), good)


def test_sequences_as_the_sintel_script_reads_them(tmp_path):
    rng = np.random.default_rng(4)
    flows = rng.standard_normal((4, 8, 12, 2)).astype(np.float32)
    occs = rng.random((4, 8, 12)) < 0.3
    for i in range(4):
        flowio.writeFlowFile(flows[i], str(tmp_path / ("frame_%04d.flo" % (i + 1))))
        PIL.fromarray((occs[i] * 255).astype(np.uint8)).save(str(tmp_path / ("frame_%04d.png" % (i + 1))))
    v = flowio.load_flow_sequence(str(tmp_path), 3)
    assert v.shape == (8, 12, 3, 2)
    for i in range(3):
        assert np.array_equal(v[:, :, i, 0], flows[i][:, :, 0])                        # :37, horizontal component
        assert np.array_equal(v[:, :, i, 1], flows[i][:, :, 1])                        # :38
    occ = flowio.load_occlusion_sequence(str(tmp_path), 3)
    assert occ.shape == (8, 12, 3) and occ.dtype == bool
    assert np.array_equal(occ, np.moveaxis(occs[:3], 0, 2))
    later = flowio.load_flow_sequence(str(tmp_path), 2, start=3)
    assert np.array_equal(later[:, :, 1, :], flows[3])
    flowio.writeFlowFile(np.zeros((3, 3, 2)), str(tmp_path / "frame_0005.flo"))
    with pytest.raises(ValueError, match="the first one"):
        flowio.load_flow_sequence(str(tmp_path), 2, start=4)


def test_upscale_flow():
    rng = np.random.default_rng(5)
    v = rng.standard_normal((6, 9, 2, 2))
    up = flowio.upscale_flow(v, 4)
    assert up.shape == (24, 36, 2, 2)
    for i in range(2):
        for c in range(2):
            ref = orc.imresize(v[:, :, i, c], 4.0) * 4.0                              # testfile_epicFlow.m:60-61
            assert np.abs(up[:, :, i, c] - ref).max() < 1e-12
    one = flowio.upscale_flow(v[:, :, 0, :], 2)
    assert one.shape == (12, 18, 2)
