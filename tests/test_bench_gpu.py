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
SMALL = ["--steps", "3", "--warmup", "1", "--streams", "8", "--triangles", "8000", "--cpu-seconds", "0.5", "--check-frames", "2",
         "--min-seconds", "0.5", "--isolated-seconds", "0.3", "--host-copy-seconds", "0.3"]
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
    assert d["timed_steps"] % 3 == 0 and d["timed_steps"] * d["ms_per_step"] * 1e-3 >= 0.3       # --min-seconds 0.5 (estimate-sized)
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # both big kernels are reported, the dominant one (longest launch, no exclusions) on top
    names = {rf["kernel"]} | {e["kernel"] for e in rf["all_kernels"]}
    assert names == {"tile_kernel<fused>", "setup_kernel", "clip_kernel"}
    assert all(rf["avg_launch_ms"] >= e["avg_launch_ms"] > 0 for e in rf["all_kernels"])
    # the roofline comes from the one-lane leg of the same run (every kernel alone on the GPU); the headline's own per-launch
    # figures sit beside it
    assert "one-lane context" in rf["measured_on"] and rf["in_headline_run"]["avg_launch_ms"] > 0 and rf["one_lane_leg"]["frames_per_s"] > 0
    assert d["config"]["raster_lanes"] == 3 and d["config"]["launch_groups_per_batch"] == 1      # 8 streams: not split, the lanes in turn
    hc = d["with_host_copies"]["modes"]
    assert len(hc) == 2 and all(m["frames_per_s"] > 0 and m["mismatching_values"] == 0 and 0 < m["fraction_of_link"]["host_to_device"] < 1.2 for m in hc.values())
    assert d["parity"]["mask_mismatch_pixels"] == 0 and d["parity"]["depth_mismatch_pixels"] == 0 and d["parity"]["frames_checked"] >= 2
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "workload" in d["config"]
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] > 0


def test_bench_line_carries_the_other_baseline_configs():
    """`other_configs` (the driver's default command runs them at full size; here the same legs at the tests' sizes): BASELINE
    config 2 (one camera: frames/s and the latency of one frame in flight), config 3 with the forearm in front of every lens,
    rank 0's shares of configs 4 (64 x 720p, robot + walls) and 5 (8 URDFs x 128 cameras) -- each with frames/s, memory, the
    tile kernel's launch time on a one-lane context and 8 streams (or all) against the oracle."""
    args = ["--steps", "3", "--warmup", "1", "--streams", "8", "--triangles", "8000", "--cpu-seconds", "0", "--check-frames", "2", "--min-seconds", "0.2",
            "--isolated-seconds", "0.2", "--host-copy-seconds", "0", "--other-configs", "on", "--other-configs-seconds", "0.3"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    oc = d["other_configs"]
    assert set(oc) == {"c2_batch1", "c3_near_arm", "c4_share", "c5_share", "note"}, oc.keys()
    for key, streams, size in (("c2_batch1", 1, [640, 480]), ("c3_near_arm", 8, [640, 480]), ("c4_share", 64, [1280, 720]), ("c5_share", 1024, [640, 480])):
        leg = oc[key]
        assert "error" not in leg, leg
        assert leg["streams"] == streams and leg["size"] == size and leg["frames_per_s"] > 0 and leg["device_memory_bytes"] > 0
        assert leg["mismatching_values"] == 0 and leg["frames_checked"] == min(8, streams)
        tk = leg["tile_kernel"]
        assert tk["avg_launch_ms"] > 0 and 0 < tk["frac"] < 1 and tk["streams_per_launch"] == streams
    assert oc["c2_batch1"]["latency_us_per_frame"] > 0
    assert 0 < d["hbm_frac_end_to_end"]["value"] < 1 and d["hbm_frac_end_to_end"]["n_gpus"] == 1


def test_bench_line_with_split_batches_and_every_stream_checked():
    """64 streams: the headline context splits every batch into three launch groups, one per raster lane; by default every
    stream of the last timed step is checked against the oracle."""
    args = ["--steps", "3", "--warmup", "1", "--streams", "64", "--triangles", "8000", "--width", "320", "--height", "192", "--cpu-seconds", "0",
            "--min-seconds", "0.3", "--isolated-seconds", "0.2", "--host-copy-seconds", "0"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["config"]["raster_lanes"] == 3 and d["config"]["launch_groups_per_batch"] == 3 and d["config"]["streams_per_launch_group"] == 22
    assert d["parity"]["frames_checked"] == 64 and d["parity"]["mismatching_values"] == 0
    rf = d["roofline"]
    assert rf["streams_per_launch"] == 64 and abs(rf["in_headline_run"]["streams_per_launch"] - 64 / 3) < 1e-9 and rf["launches_per_step"] == 1
    assert "with_host_copies" not in d and "cpu_baseline" not in d


def test_bench_nccl_world1():
    """torch.distributed.run with ONE rank and the real backend: bench.py's RCCL lines -- init_process_group("nccl"), the
    barriers, the MAX all-reduces and the all-gathers of the per-rank counts -- execute on hardware (the test box has one
    GPU; the two-rank rehearsals below share it over gloo)."""
    from conftest import page_in_rccl
    page_in_rccl()
    env = {k: v for k, v in os.environ.items() if k not in ("RTUF_BENCH_BACKEND", "RTUF_BENCH_DEVICE")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["collectives"]["world"] == 1 and "rccl" in d["collectives"]["backend"]
    pr = d["parity"]["per_rank"]
    assert len(pr) == 1 and pr[0]["frames"] == 8 * d["timed_steps"] and pr[0]["mismatching_values"] == 0
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def _two_ranks(extra, port):
    env = dict(os.environ, RTUF_BENCH_BACKEND="gloo", RTUF_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return last_json(r.stdout)


def test_bench_two_ranks_rehearsal_over_gloo():
    """c3 (weak scaling): both ranks run the same per-GPU batch size on their own streams."""
    d = _two_ranks(SMALL, 29533)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["streams_per_gpu"] == 8 and d["config"]["streams_total"] == 16
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    pr = d["parity"]["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["frames_checked"] == 2 and x["mismatching_values"] == 0 for x in pr)
    assert all(x["frames"] == 8 * d["timed_steps"] for x in pr)
    # what north_star asks of a multi-rank line: the per-kernel roofline still comes from the one-lane leg (rank 0 measures it
    # after the timed region while the others wait), the CPU baseline is there, the whole path is priced against N x 8 TB/s,
    # and every rank's own frames/s shows a straggler
    rf = d["roofline"]
    assert "one-lane context" in rf["measured_on"] and rf["frac"] > 0 and rf["one_lane_leg"]["frames_per_s"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    e2e = d["hbm_frac_end_to_end"]
    assert e2e["n_gpus"] == 2 and abs(e2e["value"] - d["value"] * e2e["bytes_per_frame"] / (2 * 8000e9)) < 1e-12
    assert all(x["frames_per_s"] > 0 for x in pr) and sum(x["frames_per_s"] for x in pr) >= d["value"] * 0.999


def test_bench_two_ranks_rehearsal_config4_and_config5():
    """The two 8-GPU configs of BASELINE.json run multi-rank through bench.py: c4 block-partitions a fixed total of
    streams (sharding.shard_range), c5 puts URDF m on rank m % N (sharding.models_for_rank); rank 0 sums the frames,
    MAX-reduces the time and gathers every rank's parity counts."""
    base = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--check-frames", "2", "--min-seconds", "0"]
    d = _two_ranks(base + ["--workload", "c4", "--streams", "7", "--triangles", "8000", "--width", "320", "--height", "192"], 29534)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["streams_per_gpu"] == [4, 3] and d["config"]["streams_total"] == 7
    assert d["config"]["workload"].startswith("C4") and "wall" in d["config"]["workload"]
    assert d["timed_steps"] == 3 and abs(d["value"] - 7 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert all(x["mismatching_values"] == 0 and x["frames_checked"] == 2 for x in d["parity"]["per_rank"])
    d = _two_ranks(base + ["--workload", "c5", "--urdfs", "5", "--streams", "3", "--triangles", "6000", "--width", "320", "--height", "240"], 29535)
    assert d["scaling"] == "strong" and d["config"]["streams_per_gpu"] == [9, 6] and d["config"]["streams_total"] == 15      # URDFs 0,2,4 | 1,3
    assert d["config"]["workload"].startswith("C5")
    assert all(x["mismatching_values"] == 0 and x["frames_checked"] == 2 for x in d["parity"]["per_rank"])


def test_bench_min_seconds_floor_and_per_gpu_share_flag():
    """--min-seconds repeats the --steps steps (whole multiples); --shard-of runs rank 0's share of a larger job."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--check-frames", "1",
                        "--min-seconds", "0.2", "--isolated-seconds", "0", "--host-copy-seconds", "0", "--workload", "c4", "--shard-of", "8", "--streams", "16", "--triangles", "8000", "--width", "320", "--height", "192"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["steps"] == 2 and d["timed_steps"] % 2 == 0 and d["timed_steps"] >= 2
    assert d["timed_steps"] * d["ms_per_step"] * 1e-3 >= 0.1           # the floor held (within the error of the step-time estimate)
    assert d["config"]["streams_total"] == 2 and "8-GPU job" in d["config"]["parallelism"]
    assert d["parity"]["mismatching_values"] == 0 and "cpu_baseline" not in d


# Launches of fewer than 2,048 tile workgroups -- every small scene of the tests -- take the tile kernels of 1,024 threads, larger ones
# those of 256 (csrc/rtuf_kernels.hip, launch_tile).  RTUF_SMALL_LAUNCH=0 in the environment sends everything through the
# 256-thread kernels: the fuzz samples run both ways, so both instantiations see the small, nasty scenes.
SMALL_LAUNCH = [pytest.param(None, id="tile_threads_by_launch_size"), pytest.param("0", id="tile_threads_256")]


def _env_small_launch(value):
    return dict(os.environ) if value is None else dict(os.environ, RTUF_SMALL_LAUNCH=value)


@pytest.mark.parametrize("small_launch", SMALL_LAUNCH)
def test_fuzz_parity_sample(small_launch):
    """A sample of scripts/fuzz_parity.py (random resolutions, intrinsics, soups from sub-pixel dust to
    screen-filling triangles, near-plane crossings, both modes, forced bin regrowth) against the oracle."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "80", "777"], capture_output=True, text=True, cwd=ROOT, timeout=900,
                       env=_env_small_launch(small_launch))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "streams with mismatches 0" in r.stdout


@pytest.mark.parametrize("small_launch", SMALL_LAUNCH)
def test_fuzz_features_sample(small_launch):
    """A sample of scripts/fuzz_features.py: streams 1-9, several models with per-stream selection, primitives
    and multi-chunk meshes, 16UC1, optional mask, several launch groups per batch, both modes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_features.py"), "60", "4242"], capture_output=True, text=True, cwd=ROOT, timeout=900,
                       env=_env_small_launch(small_launch))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "streams with mismatches 0" in r.stdout
