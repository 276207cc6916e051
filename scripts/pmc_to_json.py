#!/usr/bin/env python3
"""Turns the counter text files scripts/refresh_profiles.sh wrote into the two json files bench.py reads:

  pmc_counters.json  per kernel of the default bench command: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE with the
                     gfx950 corrections of MI355X_MICROARCH.md), wave64 VALU instructions per launch, and the mean
                     issue cycles of its dynamic instruction mix (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU, calibrated on
                     the micro-benchmark's single-instruction kernels)
  valu_peak.json     the micro-benchmark's measured issue rates + the peak rate of each kernel's own mix

usage: python scripts/pmc_to_json.py <dir with <tag>_pmc*.txt, valu_peak_raw.json, <tag>_pmc_valu_peak.txt> <tag>
"""
import json
import os
import re
import sys


def parse_pmc(path):
    """{kernel name prefix: {counter: avg}} over all '##' groups of one file."""
    out = {}
    cur = None
    for line in open(path):
        if line.startswith("#"):
            continue
        m = re.match(r"^\s+(\w+)\s+avg ([0-9.eE+\-]+)", line)
        if m and cur is not None:
            out.setdefault(cur, {})[m.group(1)] = float(m.group(2))
        elif line.strip():
            cur = line.strip()
    return out


def find(d, *needles):
    """First kernel whose (truncated) name holds every needle.  The tile kernels carry their workgroup size as a last template
    argument since round 5 (<..., 256> for launches that fill the GPU, <..., 1024> for small ones): a needle that ends in '>' also
    matches the same arguments followed by ', 256>' -- the profiled launches are the large ones -- or by a truncated name."""
    def hit(n, k):
        if n in k:
            return True
        if n.endswith(">"):
            stem = n[:-1]
            i = k.find(stem)
            return i >= 0 and (k[i + len(stem):].startswith(", 256>") or ">" not in k[i + len(stem):] and not k[i + len(stem):].startswith(", 1024"))
        return False
    for k, v in d.items():
        if all(hit(n, k) for n in needles):
            return v
    return None


def hbm_bytes(c):
    # gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2; both counters are in KiB
    if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        return None
    return int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    raw = json.load(open(os.path.join(d, "valu_peak_raw.json")))
    ops = raw["ops"]
    slow = ops["v_mul_i32_i24"]["G_wave64_instr_per_s"]
    fast = ops["v_add_u32"]["G_wave64_instr_per_s"]
    # calibration of SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU on kernels whose issue rate is known
    cal = {}
    p = os.path.join(d, "%s_pmc_valu_peak.txt" % tag)
    if os.path.exists(p):
        for line in open(p):
            m = re.match(r"^k_(\w+)\s+SQ_INSTS_VALU ([0-9.eE+\-]+)\s+SQ_ACTIVE_INST_VALU ([0-9.eE+\-]+)\s+ratio ([0-9.]+)", line)
            if m:
                cal[m.group(1)] = float(m.group(4))
    r_slow, r_fast = cal.get("v_mul_i32_i24"), cal.get("v_add_u32")
    counters_track_issue = bool(r_slow and r_fast and r_fast < 0.8 * r_slow)

    main_c = parse_pmc(os.path.join(d, "%s_pmc.txt" % tag))
    two_c = parse_pmc(os.path.join(d, "%s_pmc_two_kernel.txt" % tag)) if os.path.exists(os.path.join(d, "%s_pmc_two_kernel.txt" % tag)) else {}
    two1024_c = parse_pmc(os.path.join(d, "%s_pmc_two_kernel_streams_1024.txt" % tag)) if os.path.exists(os.path.join(d, "%s_pmc_two_kernel_streams_1024.txt" % tag)) else {}

    def entry(c):
        if c is None:
            return None
        e = {"counters_per_launch": c, "hbm_bytes_per_launch": hbm_bytes(c)}
        if "SQ_INSTS_VALU" in c:
            e["wave64_valu_instructions_per_launch"] = c["SQ_INSTS_VALU"]
        if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            e["active_over_insts"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
        if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
            # rocprofv3's derived VALUUtilization: lanes live per issued VALU instruction (a kernel with every lane on reads 1.000)
            e["live_lane_fraction"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        return e

    def add(a, b):
        if a is None or b is None:
            return a or b
        return {k: a.get(k, 0) + b.get(k, 0) for k in set(a) | set(b)}

    # template arguments: <two_kernel, 16UC1, cover pass>; the cover pass switches itself off on workloads without whole-tile
    # triangles (every 64th batch probes with it on), so the headline's kernel is <false, false, false>
    tile = find(main_c, "tile_kernel<false, false, false>") or find(main_c, "tile_kernel<false, false>")
    setup = find(main_c, "setup_kernel<false>")
    clip = add(find(main_c, "clip_kernel"), add(find(main_c, "bigrec_kernel<1>"), find(main_c, "bigrec_kernel<0>")) or find(main_c, "bigrec_kernel"))      # bench.py times them together
    kernels = {"tile_kernel<fused>": entry(tile), "tile_kernel<fused, cover pass on>": entry(find(main_c, "tile_kernel<false, false, true>")),
               "setup_kernel": entry(setup), "clip_kernel": entry(clip),
               "setup_kernel+clip_kernel": entry(add(setup, clip)),
               "tile_kernel<two_kernel>": entry(find(two_c, "tile_kernel<true, false")), "compare_kernel": entry(find(two_c, "compare_kernel"))}
    if two1024_c:
        kernels["compare_kernel@1024_streams"] = entry(find(two1024_c, "compare_kernel"))
        kernels["tile_kernel<two_kernel>@1024_streams"] = entry(find(two1024_c, "tile_kernel<true, false"))
    kernels = {k: v for k, v in kernels.items() if v}
    mix_peak = {}
    for name, e in kernels.items():
        a = e.get("active_over_insts")
        if a and counters_track_issue:
            # issue cycles of this kernel's dynamic mix relative to the all-slow (4-cycle class) calibration kernel
            rel = a / r_slow
            e["mean_issue_cycles_relative_to_slow_class"] = rel
            mix_peak[name] = max(min(slow / rel, fast), slow * 0.5)
    # config 4's per-GPU share (cover pass on: tile_kernel<false, false, true>)
    c4 = None
    p4 = os.path.join(d, "%s_pmc_workload_c4_shard_of_8.txt" % tag)
    if os.path.exists(p4):
        c4_c = parse_pmc(p4)
        k4 = {"tile_kernel<fused>": entry(find(c4_c, "tile_kernel<false, false, true>")), "setup_kernel": entry(find(c4_c, "setup_kernel<false>")),
              "clip_kernel": entry(add(find(c4_c, "clip_kernel"), add(find(c4_c, "bigrec_kernel<1>"), find(c4_c, "bigrec_kernel<0>"))))}
        c4 = {"source": "profiles/%s_pmc_workload_c4_shard_of_8.txt (rocprofv3 --kernel-trace --pmc, one pass per counter group, python bench.py --steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0 --workload c4 --shard-of 8: one raster lane, the whole share per launch)" % tag,
              "workload": {"streams": 64, "width": 1280, "height": 720, "triangles": 250388, "mode": "fused, cover pass on"},
              "kernels": {k: v for k, v in k4.items() if v}}
    json.dump({"source": "profiles/%s_pmc*.txt (rocprofv3 --kernel-trace --pmc, one pass per counter group, python bench.py --steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0: one raster lane, all 256 streams per launch)" % tag, "streams_per_launch": 256,
               "workload": {"streams": 256, "width": 640, "height": 480, "triangles": 250388, "mode": "fused (two-kernel entries from the --two-kernel passes)"},
               "corrections": "gfx950: FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): fetch bytes = 2 x FETCH_SIZE x 1024; write bytes = WRITE_SIZE x 1024",
               "kernels": kernels, "c4_share": c4}, open(os.path.join(d, "pmc_counters.json"), "w"), indent=1)
    json.dump({"source": "scripts/valu_peak.hip on %s (%s), %d CUs, %d waves/SIMD: measured wave64 VALU instructions per second over the whole GPU" % (raw["device"], raw["arch"], raw["compute_units"], raw["waves_per_simd"]),
               "peak_G_per_s": slow,
               "peak_note": "issue rate of the 4-cycle class (v_mul_i32_i24 and most integer / conversion / min-max instructions); the 2-cycle class (v_add_u32, v_sub_u32, v_and/or_b32, v_ashrrev_i32, v_mov_b32, v_fma/mul/add_f32) measures %.0f G/s" % fast,
               "slow_class_G_per_s": slow, "fast_class_G_per_s": fast,
               "counter_calibration": {"SQ_ACTIVE_INST_VALU_over_SQ_INSTS_VALU": cal, "counters_track_issue_cycles": counters_track_issue},
               "mix_peak_G_per_s": mix_peak,
               "mix_peak_note": "per kernel: slow-class rate / (its SQ_ACTIVE_INST_VALU per instruction relative to the all-slow calibration kernel), clamped to [0.5 x slow, fast]; empty if the counters do not separate the two classes",
               "ops": ops}, open(os.path.join(d, "valu_peak.json"), "w"), indent=1)
    print("kernels:", {k: (v.get("hbm_bytes_per_launch"), v.get("wave64_valu_instructions_per_launch"), v.get("active_over_insts")) for k, v in kernels.items()})
    print("calibration ratios: slow %s fast %s -> counters_track_issue=%s; mix peaks %s" % (r_slow, r_fast, counters_track_issue, mix_peak))


if __name__ == "__main__":
    main()
