#!/usr/bin/env python3
"""Static VALU instruction mix of the big kernels + the issue-rate peak of that mix (no GPU needed).

Disassembles rtuf_kernels.hip for gfx950 (hipcc -S), counts the VALU opcodes of tile_kernel<fused>, setup_kernel and
clip_kernel, and combines them with the per-instruction issue rates measured by scripts/valu_peak.hip
(profiles/valu_peak.json) into the harmonic-mean rate of each kernel's mix.  STATIC counts, i.e. every instruction
weighted once: an estimate of the dynamic mix (hot loops execute more often), which no counter exposes on gfx950
(SQ_ACTIVE_INST_VALU counts one unit per instruction for both the 2-cycle and the 4-cycle class).

    python scripts/valu_mix.py            # rewrites profiles/valu_peak.json with static_mix_peak_G_per_s
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realtime_urdf_filter_amd", "csrc")
KERNELS = {"tile_kernel<fused>": "tile_kernelILb0ELb0ELb0ELi256E", "setup_kernel": "setup_kernelILb0E", "clip_kernel": "clip_kernelE",
           "tile_kernel<two_kernel>": "tile_kernelILb1ELb0ELb0ELi256E", "compare_kernel": "compare_kernelILb0E"}
TRANSCENDENTAL = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def main():
    vp_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "valu_peak.json")
    vp = json.load(open(vp_path))
    ops = vp["ops"]
    slow, fast = vp["slow_class_G_per_s"], vp["fast_class_G_per_s"]
    rate = {k: v["G_wave64_instr_per_s"] for k, v in ops.items() if v.get("instr_per_group", 1) == 1 and k.startswith("v_")}
    rate.pop("v_cndmask_b32", None)            # (the VCC-mask form measured alone is an artefact; the SGPR-mask form below is the real rate)
    rate["v_cndmask_b32"] = ops["v_cndmask_b32_e64_sgpr_mask"]["G_wave64_instr_per_s"]
    for alias, src in (("v_cmp", "v_cmp_lt_i32_to_vcc"),):
        rate[alias] = ops[src]["G_wave64_instr_per_s"]
    rcp = ops["v_rcp_f32"]["G_wave64_instr_per_s"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, os.path.join(CSRC, "rtuf_kernels.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    per = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_ZN4rtuf\w+):", line)
        if m:
            cur = m.group(1)
            per[cur] = collections.Counter()
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if cur:
            m = re.match(r"^\s+(v_\w+)", line)
            if m:
                per[cur][m.group(1)] += 1

    def op_rate(op):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base in rate:
            return rate[base], "measured"
        if base.startswith("v_cmp") or base.startswith("v_cmpx"):
            return rate["v_cmp"], "class: compare"
        if base.startswith(TRANSCENDENTAL):
            return rcp, "class: transcendental"
        return slow, "class: 4-cycle (default for unmeasured opcodes)"

    result, detail = {}, {}
    for name, mangled in KERNELS.items():
        key = next((k for k in per if mangled in k), None)
        if key is None:
            continue
        c = per[key]
        total = sum(c.values())
        time_units = sum(n / op_rate(op)[0] for op, n in c.items())
        fast_n = sum(n for op, n in c.items() if op_rate(op)[0] > 0.75 * fast)
        result[name] = total / time_units
        detail[name] = {"static_valu_instructions": total, "share_in_the_2_cycle_class": fast_n / total,
                        "top_opcodes": dict(c.most_common(12))}
    vp["static_mix_peak_G_per_s"] = result
    vp["static_mix_detail"] = detail
    vp["static_mix_note"] = ("harmonic mean of the measured per-instruction rates over each kernel's STATIC VALU opcode histogram "
                             "(scripts/valu_mix.py, disassembly of the committed sources); an estimate of the dynamic mix")
    json.dump(vp, open(vp_path, "w"), indent=1)
    for k, v in result.items():
        print("%-26s static mix peak %.0f G/s  (%d VALU instructions, %.0f %% in the 2-cycle class)" % (k, v, detail[k]["static_valu_instructions"], 100 * detail[k]["share_in_the_2_cycle_class"]))


if __name__ == "__main__":
    main()
