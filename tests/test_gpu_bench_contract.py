"""bench.py as a command: the one JSON line it prints, its keys and their consistency, on a small
workload; and the same command with --gpus 2, which starts its own ranks (here both on the one
visible GPU, frames staged through the host)."""
import json
import math
import os
import sys

import pytest

import spawn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--lr", "64", "96", "--frames-per-gpu", "3", "--steps", "4", "--warmup", "2", "--extra-flows", "", "--cpu-seconds", "2"]


def run_bench(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = spawn.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, cwd=ROOT, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run_bench([])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["unit"] == "HR Mpixel*PDHG-iters/s" and "workload" in d["config"] and "model" not in d["config"]
    M = 4 * 64 * 4 * 96 * 3
    assert d["config"]["hr_pixels_total"] == M
    assert math.isclose(d["value"], M / d["ms_per_step"] / 1e3, rel_tol=1e-9)          # whole-job pixels per second
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["kernel"] in ("k_primal4", "k_dual4") and rf["launches_timed"] == 4
    assert math.isclose(rf["frac"], rf["achieved"] / rf["peak"], rel_tol=1e-12)
    assert math.isclose(rf["achieved"], rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9, rel_tol=1e-9)
    assert rf["traffic"] is None                                     # counter summaries are for the 1080p workload only
    assert sum(rf["all_kernels_ms"].values()) <= d["ms_per_step"] * 1.02
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == d["unit"] and cb["sample"]
    assert d["sustained"]["steps"] >= 200 and d["sustained"]["value"] > 0
    assert "group_driver" not in d
    g = run_bench(["--no-cpu-baseline", "--group-blocks", "3"])
    assert g["group_driver"]["devices"] == [0, 0, 0] and g["group_driver"]["value"] > 0


def test_two_ranks_started_by_the_command_itself():
    d = run_bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["frames_total"] == 6 and d["scaling"] == "weak"
    assert d["value"] > 0 and "cpu_baseline" not in d
    # extra leg: the same six frames through the one-process driver, one block per rank's device
    gd = d["group_driver"]
    assert "error" not in gd, gd
    assert gd["devices"] == [0, 0] and gd["frames_total"] == 6 and gd["value"] > 0
    assert math.isclose(gd["value"], 4 * 64 * 4 * 96 * 6 / gd["ms_per_step"] / 1e3, rel_tol=1e-9)
    s = run_bench(["--gpus", "2", "--share-gpu", "--backend", "gloo", "--no-cpu-baseline", "--total-frames", "5"])
    assert s["scaling"] == "strong" and s["config"]["frames_total"] == 5
