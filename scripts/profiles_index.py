#!/usr/bin/env python3
"""profiles/INDEX.json: which committed file backs which figure quoted in README.md / DESIGN.md / profiles/README.md.

    python scripts/profiles_index.py [tag]        (default tag: the one profiles/pmc_counters.json was made from)

Every entry names the claim, the document sections that quote it, the file, how to read the figure out of the file and the
value read out NOW -- the values are extracted from the files by this script, not typed, so the index cannot drift from what
is committed (tests/test_profiles_cpu.py runs it and compares)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def bench_line(name):
    with open(os.path.join(PROFILES, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def kernel_avg_us(name, pattern):
    with open(os.path.join(PROFILES, name)) as f:
        for r in csv.DictReader(f):
            if re.search(pattern, r["Name"]):
                return round(float(r["AverageNs"]) / 1e3, 1), int(r["Calls"])
    return None, None


def build(tag):
    entries = []

    def add(claim, docs, file, how, value):
        if os.path.exists(os.path.join(PROFILES, file)):
            entries.append({"claim": claim, "quoted_in": docs, "file": "profiles/" + file, "read_as": how, "value": value})

    def bench(claim, docs, file, extra=None):
        path = os.path.join(PROFILES, file)
        if not os.path.exists(path):
            return
        d = bench_line(file)
        v = {"frames_per_s": round(d["value"]), "ms_per_step": round(d["ms_per_step"], 4), "mismatching_values": d["parity"]["mismatching_values"],
             "device_memory_GB": round(d["device_memory_bytes"] / 1e9, 2)}
        r = d.get("roofline") or {}
        if r:
            v["roofline"] = {"kernel": r.get("kernel"), "avg_launch_us": round(r["avg_launch_ms"] * 1e3, 1), "frac": round(r["frac"], 3)}
        if extra:
            v.update(extra(d))
        add(claim, docs, file, "last line = the bench's JSON: value, ms_per_step, parity.mismatching_values, device_memory_bytes, roofline", v)

    def others(d):
        out = {}
        for k, o in (d.get("other_configs") or {}).items():
            if isinstance(o, dict) and "frames_per_s" in o:
                out[k] = {"frames_per_s": round(o["frames_per_s"]), "mismatching_values": o.get("mismatching_values")}
                if "latency_us_per_frame" in o:
                    out[k]["latency_us_per_frame"] = round(o["latency_us_per_frame"], 1)
        cb = d.get("cpu_baseline") or {}
        return {"other_configs": out, "step_ms": {k: round(v, 3) for k, v in (d.get("step_ms") or {}).items() if isinstance(v, float)},
                "cpu_baseline_frames_per_s": cb.get("value"), "distinct_joint_states_per_stream": d["config"].get("distinct_joint_states_per_stream")}

    T = tag
    bench("headline: python bench.py, all defaults (C3, 256 VGA streams, three raster lanes)", ["README.md", "DESIGN.md 4-5", "profiles/README.md"], T + "_bench.json", others)
    bench("one raster lane (every kernel alone on the GPU)", ["DESIGN.md 3", "profiles/README.md"], T + "_bench_one_lane.json")
    bench("two raster lanes", ["README.md", "profiles/README.md"], T + "_bench_two_lanes.json")
    bench("launch groups of 43 (half the working set)", ["README.md", "DESIGN.md 3"], T + "_bench_groups_of_43.json")
    bench("two lanes x launch groups of 64", ["README.md"], T + "_bench_two_lanes_groups_of_64.json")
    bench("launch groups of 22", ["README.md"], T + "_bench_groups_of_22.json")
    bench("1,024 streams per GPU", ["profiles/README.md"], T + "_bench_1024.json")
    bench("16UC1 planes in and out (fused conversions)", ["profiles/README.md"], T + "_bench_u16.json")
    bench("poses as host matrices instead of joint positions", ["README.md"], T + "_bench_host_poses.json")
    bench("two-kernel mode (z surface + compare kernel)", ["DESIGN.md 4"], T + "_bench_two_kernel.json")
    bench("two-kernel mode, 1,024 streams (compare kernel beyond the MALL)", ["DESIGN.md 4"], T + "_bench_two_kernel_1024.json")
    bench("BASELINE config 2: one camera, lanes taken in turn", ["README.md", "DESIGN.md 4"], T + "_bench_batch1.json")
    bench("one camera, one lane", ["profiles/README.md"], T + "_bench_batch1_one_lane.json")
    bench("one camera, three pipelines with graph replay", ["README.md", "DESIGN.md 4"], T + "_bench_batch1_pipelines3.json")
    bench("arm in front of the lens (bench.py --near-arm)", ["README.md", "DESIGN.md 8", "docs/experiments.md R6.1"], T + "_bench_near_arm.json")
    bench("BASELINE config 4: rank 0's share of the 8-GPU job (64 x 720p, robot + walls)", ["README.md", "DESIGN.md 4, 8"], T + "_bench_c4_share.json")
    bench("BASELINE config 5: rank 0's share of the 8-GPU job (8 URDFs x 128 cameras)", ["README.md"], T + "_bench_c5_share.json")
    bench("one rank through torch.distributed.run with the nccl (RCCL) backend", ["README.md", "DESIGN.md 5"], T + "_bench_rccl_world1.json")
    bench("the driver's exact command, first thing on a fresh box: python bench.py --gpus 1 --steps 20 --warmup 5", ["profiles/README.md"], T + "_bench_driver_command.json", others)

    for suffix, what, pats in (("", "one-lane launch shape, C3: the roofline's kernel times", (("tile_kernel", r"tile_kernel<false, false, false"), ("setup_kernel", r"setup_kernel<false>"), ("clip_kernel", r"clip_kernel"))),
                               ("_near_arm", "arm in front of the lens", (("tile_kernel", r"tile_kernel<false, false, false"), ("setup_kernel", r"setup_kernel<false>"))),
                               ("_c4_share", "config 4 share", (("tile_kernel<cover>", r"tile_kernel<false, false, true"), ("setup_kernel", r"setup_kernel<false>"), ("clip_kernel", r"clip_kernel"))),
                               ("_two_kernel", "two-kernel mode", (("compare_kernel", r"compare_kernel"), ("tile_kernel<two_kernel>", r"tile_kernel<true"))),
                               ("_two_kernel_1024", "two-kernel mode, 1,024 streams", (("compare_kernel", r"compare_kernel"),))):
        f = "%s_kernel_stats%s.csv" % (T, suffix)
        if os.path.exists(os.path.join(PROFILES, f)):
            v = {}
            for label, pat in pats:
                us, calls = kernel_avg_us(f, pat)
                if us is not None:
                    v[label] = {"avg_us": us, "launches": calls}
            add("rocprofv3 --kernel-trace --stats: " + what, ["DESIGN.md 4", "profiles/README.md"], f, "AverageNs / Calls of the kernel's row", v)

    pc = os.path.join(PROFILES, "pmc_counters.json")
    if os.path.exists(pc):
        c = json.load(open(pc))
        v = {}
        for k in ("tile_kernel<fused>", "setup_kernel"):
            e = c["kernels"].get(k)
            if e:
                v[k] = {"hbm_bytes_per_launch": e.get("hbm_bytes_per_launch"), "valu_instructions_per_launch": (e.get("counters_per_launch") or {}).get("SQ_INSTS_VALU"),
                        "live_lane_fraction": round(e["live_lane_fraction"], 3) if e.get("live_lane_fraction") else None}
        add("counter traffic and instruction counts bench.py quotes as OFFLINE (from %s_pmc*.txt by scripts/pmc_to_json.py)" % T, ["DESIGN.md 4", "bench.py roofline.traffic / valu_issue"],
            "pmc_counters.json", "kernels[...]: 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950 correction), SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)", v)

    def text(claim, docs, file, how):
        p = os.path.join(PROFILES, file)
        if os.path.exists(p):
            add(claim, docs, file, how, open(p).read().strip().splitlines()[-1][:200])

    text("GPU test suite on the round's tree", ["DESIGN.md 8"], T + "_gpu_tests.txt", "pytest's last line")
    text("fuzz campaigns (fuzz_parity 6,000 + fuzz_features 3,000 + FUZZ_BIG 150 scenes)", ["DESIGN.md 2"], T + "_fuzz.txt", "'streams with mismatches' of each of the three runs")
    text("wall time of the default bench command", ["README.md"], T + "_bench_wall_time.txt", "the line")
    for f, claim, docs in ((T + "_experiment_block_bounds.txt", "block depth bounds (not merged)", ["DESIGN.md 8", "docs/experiments.md R6.1"]),
                           (T + "_experiment_more_lanes.txt", "4-6 raster lanes", ["docs/experiments.md R6.6"]),
                           (T + "_experiment_compiler_flags.txt", "compiler flag sweep", ["docs/experiments.md R6.6"]),
                           (T + "_experiment_fast_class.txt", "2-cycle-class instructions in the hot walks (kept)", ["DESIGN.md 4, 8", "docs/experiments.md R6.7"]),
                           (T + "_experiment_fast_class_batches_2_3.txt", "no exact-z look without near geometry, z by one add, the division's core; fdiv_check", ["DESIGN.md 4, 8", "docs/experiments.md R6.7"]),
                           (T + "_fdiv_check.txt", "the division core equals __fdiv_rn inside the admitted domain (all float z in [-1, 1 + 2^-11])", ["DESIGN.md 4", "docs/experiments.md R6.7"]),
                           (T + "_final_campaign.txt", "final campaign on the final tree: soak, 27,300 fuzz scenes, clip stress, the driver's command", ["DESIGN.md 2", "profiles/README.md"]),
                           (T + "_same_box_final_round5_vs_round6.txt", "round 5's tree and the final tree alternating on one box", ["DESIGN.md 8", "profiles/README.md"]),
                           (T + "_pcie_probe.txt", "the host link: 56-57 GB/s one way, 40 + 50 both", ["DESIGN.md 5"]),
                           (T + "_pcie_probe_streams.txt", "the host link on 1 / 2 / 4 streams", ["DESIGN.md 5"]),
                           (T + "_same_box_c3_round5_vs_round6.txt", "round 5's tree and round 6's on one box, C3", ["DESIGN.md 4", "docs/experiments.md R6.2"]),
                           (T + "_same_box_c4_share_round5_vs_round6.txt", "... C4 share", ["DESIGN.md 4", "docs/experiments.md R6.4"]),
                           (T + "_host_planes.json", "planes in pinned host memory (PCIe-inclusive rates)", ["README.md", "DESIGN.md 5"]),
                           (T + "_lanes.json", "lanes live per trip of every walk / emission loop", ["docs/experiments.md A.5"]),
                           ("overdraw.json", "depth tests per drawn pixel", ["bench.py rasteriser.tile.overdraw"]),
                           ("valu_peak.json", "measured VALU issue rates per instruction (two classes)", ["DESIGN.md 4", "docs/experiments.md A.1"]),
                           ("llvmpipe_baseline.json", "the reference's GLSL on llvmpipe at BASELINE sizes (development container)", ["DESIGN.md 2", "BASELINE.md"])):
        if os.path.exists(os.path.join(PROFILES, f)):
            entries.append({"claim": claim, "quoted_in": docs, "file": "profiles/" + f, "read_as": "the file's own header says how it was taken", "value": None})
    return {"tag": T, "note": "made by scripts/profiles_index.py from the files named; history/ holds rounds 1-4, r05_* round 5", "entries": entries}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    tag = args[0] if args else None
    if tag is None:
        src = json.load(open(os.path.join(PROFILES, "pmc_counters.json")))["source"]
        tag = re.search(r"profiles/(r\d\d)_pmc", src).group(1)
    out = build(tag)
    if "--check" in sys.argv:
        committed = json.load(open(os.path.join(PROFILES, "INDEX.json")))
        sys.exit(0 if committed == out else 1)
    with open(os.path.join(PROFILES, "INDEX.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("profiles/INDEX.json: %d entries for %s" % (len(out["entries"]), tag))
