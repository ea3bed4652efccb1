"""Random small configurations of both solves against the scipy oracles (tools/fuzz_shapes.py):
frames from 1 x 1 low-resolution pixels to several tiles, zoom factors 1..5, 1..4 frames, every model
and flow direction, masks, blur kernels 1x1..7x7 or none; 25 iterations, relative L2 <= 1e-4."""
import os
import sys

import pytest

import spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_small_configurations(seed):
    res = spawn.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_shapes.py"), "150", str(seed)], timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "misses 0" in res.stdout


def test_random_medium_configurations():
    """tools/fuzz_medium.py: tens to hundreds of tiles per frame, 2 to 6 frames, outlier vectors, against the
    C restatement at 12 iterations, relative L2 <= 2e-5."""
    res = spawn.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_medium.py"), "40", "21"], timeout=600)
    if "C oracle not built" in res.stdout + res.stderr:
        pytest.skip("C oracle not built")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "misses 0" in res.stdout


def test_random_cuts_into_frame_blocks():
    """tools/fuzz_group.py: random sequences cut into random numbers of frame blocks on this GPU through the
    single-process group driver equal the single-block solve bit for bit."""
    # The tool runs on the HIP runtime that PyTorch-ROCm brings along, like every Python user of the library.  On
    # that runtime (7.0.51831) hipStreamDestroy frees a stream's queue object while completion handlers of its
    # commands are still pending on the runtime's signal thread; they then decrement a counter and clear a flag
    # inside the freed object (tools/watch/: a hardware watchpoint names the store).  Some hundred create / solve /
    # destroy cycles of multi-block groups used to die of that -- the freed chunk has the size of a six-block
    # table.  The library no longer destroys streams or events: they go back to a per-device pool.
    res = spawn.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_group.py"), "150", "31"], timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "misses 0" in res.stdout


def test_random_sampler_configurations():
    """tools/fuzz_samplers.py: every sampler of helpers/samplingOperator.m handed over as rows, zoom 2 to 4, with and
    without anti-aliasing and blur (composite kernels applied in one pass and, where separable and large, in two):
    operators, step sizes and 15 iterations against the explicit sparse operators of the Python oracle."""
    res = spawn.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_samplers.py"), "40", "51"], timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "misses 0" in res.stdout
