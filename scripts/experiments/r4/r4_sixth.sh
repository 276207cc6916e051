#!/bin/bash
out=gpurun_out/r4_sixth; mkdir -p $out
line() { python - "$1" <<'PY' | tee -a gpurun_out/r4_sixth/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}
    print(round(d["value"]), "frames/s", round(d["device_memory_bytes"]/1e9,2), "GB", d["config"]["raster_lanes"], "lanes", d["config"]["launch_groups_per_batch"], "groups", "parity", d["parity"]["frames_checked"], d["parity"]["mismatching_values"],
          {k.split("_")[0]:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)), "exact tiles", d["rasteriser"]["tile"]["exact_z_tiles"])
except Exception as e:
    print("no line", e)
PY
}
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q > $out/parity_tests.txt 2>&1; echo "parity tests rc=$?" | tee -a $out/summary.txt; tail -3 $out/parity_tests.txt
for v in "" "--near-arm" "--workload c4 --shard-of 8" "--workload c5 --shard-of 8"; do
  tag=$(echo "default$v" | tr -d ' -')
  timeout 600 python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1 $v > $out/bench_$tag.json 2> $out/bench_$tag.err; echo "bench [$v] rc=$?" | tee -a $out/summary.txt
  line $out/bench_$tag.json
done
python scripts/fuzz_features.py 800 616161 > $out/fuzz_features.txt 2>&1; echo "fuzz_features rc=$?" | tee -a $out/summary.txt; tail -2 $out/fuzz_features.txt
