#!/usr/bin/env python3
"""Markdown table of the bench lines under a directory (profiles/ or gpurun_out/profiles/):
   python scripts/results_table.py profiles r04
One row per <tag>_bench*.json: frames/s, ms per step, device memory, lanes x groups, the roofline kernels' per-launch times
(one-lane leg / in the headline run), parity."""
import glob
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
print("| file | frames/s | ms/step | device memory | lanes × groups | tile / set-up / clip µs per launch: alone (in the headline run) | one-lane leg frames/s | parity: frames checked, mismatches |")
print("|---|---|---|---|---|---|---|---|")
for f in sorted(glob.glob(os.path.join(d, tag + "_bench*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:       # noqa: BLE001
        print("| %s | unreadable: %s |" % (os.path.basename(f), e))
        continue
    r = j["roofline"]
    ks = {e["kernel"].split("_")[0].split("<")[0]: e for e in [r] + r["all_kernels"]}

    def us(k):
        e = ks.get(k)
        if not e:
            return "–"
        h = (e.get("in_headline_run") or {}).get("avg_launch_ms")
        return "%.0f" % (e["avg_launch_ms"] * 1e3) + (" (%.0f)" % (h * 1e3) if h else "")
    c = j["config"]
    print("| `%s` | **%s** | %.3f | %.2f GB | %s × %s of %s | %s / %s / %s | %s | %d, %d |" % (
        os.path.basename(f), "{:,.0f}".format(j["value"]), j["ms_per_step"], j["device_memory_bytes"] / 1e9,
        c.get("raster_lanes", "?"), c.get("launch_groups_per_batch", "?"), c.get("streams_per_launch_group", "?"),
        us("tile"), us("setup"), us("clip"),
        "{:,.0f}".format((r.get("one_lane_leg") or {}).get("frames_per_s", 0)) if r.get("one_lane_leg") else "–",
        j["parity"]["frames_checked"], j["parity"]["mismatching_values"]))
