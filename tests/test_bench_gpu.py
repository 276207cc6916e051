"""GPU (-m gpu): bench.py's contract on a small workload -- one JSON line with the fields the driver reads,
parity reported as zero mismatches, and the multi-rank path (one process per rank, barrier + MAX all-reduce of
the elapsed time) rehearsed with two ranks sharing the one GPU of the test box over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--streams", "8", "--triangles", "8000", "--cpu-seconds", "0.5", "--check-frames", "2"]
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["parity"]["mask_mismatch_pixels"] == 0 and d["parity"]["depth_mismatch_pixels"] == 0 and d["parity"]["frames_checked"] >= 2
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "workload" in d["config"]


def test_bench_two_ranks_rehearsal_over_gloo():
    env = dict(os.environ, RTUF_BENCH_BACKEND="gloo", RTUF_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["streams_per_gpu"] == 8
    assert abs(d["value"] - 2 * 8 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_fuzz_parity_sample():
    """A sample of scripts/fuzz_parity.py (random resolutions, intrinsics, soups from sub-pixel dust to
    screen-filling triangles, near-plane crossings, both modes, forced bin regrowth) against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "80", "777"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "streams with mismatches 0" in r.stdout


def test_fuzz_features_sample():
    """A sample of scripts/fuzz_features.py: streams 1-9, several models with per-stream selection, primitives
    and multi-chunk meshes, 16UC1, optional mask, several launch groups per batch, both modes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_features.py"), "60", "4242"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "streams with mismatches 0" in r.stdout
