#!/bin/bash
# round 5, last call: the GPU suite, smoke() and the driver's default bench command on the final tree
out=gpurun_out/r5z; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) | tee $out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $out/smoke.txt
t0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; t1=$(date +%s.%N)
python - $out/bench.json $t0 $t1 <<'PY' | tee $out/bench_summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]; vi = r.get("valu_issue", {})
print("wall %.1f s  value %.0f frames/s  tile %.1f us  frac %.3f  traffic %s  parity %s" % (float(sys.argv[3]) - float(sys.argv[2]), d["value"], r["avg_launch_ms"] * 1e3, r["frac"], r.get("traffic"), d["parity"]["mismatching_values"]))
print("valu_issue: frac %.3f  frac_of_static_mix_peak %s  live_lane_fraction %s" % (vi.get("frac", 0), vi.get("frac_of_static_mix_peak"), vi.get("live_lane_fraction")))
print("source:", r.get("traffic_source", "")[:80])
for k, v in d["other_configs"].items():
    if isinstance(v, dict): print("  %-12s %9.0f frames/s  tile %.1f us frac %.3f  mismatches %s  %s" % (k, v["frames_per_s"], v["tile_kernel"]["avg_launch_ms"] * 1e3, v["tile_kernel"]["frac"], v["mismatching_values"], ("latency %.0f us" % v["latency_us_per_frame"]) if "latency_us_per_frame" in v else ""))
print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], "hbm_frac_end_to_end", round(d["hbm_frac_end_to_end"]["value"], 4))
PY
