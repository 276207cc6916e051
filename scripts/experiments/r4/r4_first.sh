#!/bin/bash
# round 4, first GPU call: the raster-lane tests, the GPU suite, and the headline workload at several launch-group sizes
out=gpurun_out/r4_first; mkdir -p $out
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "raster_lanes or small_batches or memory_limit" > $out/lanes_tests.txt 2>&1; echo "lanes tests rc=$?" | tee -a $out/summary.txt
tail -5 $out/lanes_tests.txt
for v in "" "--launch-group 128" "--launch-group 32" "--lanes 1"; do
  tag=$(echo "default$v" | tr -d ' -')
  timeout 600 python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1 $v > $out/bench_$tag.json 2> $out/bench_$tag.err; echo "bench [$v] rc=$?" | tee -a $out/summary.txt
  python - $out/bench_$tag.json <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}
    print(round(d["value"]), "frames/s", round(d["device_memory_bytes"]/1e9,2), "GB", d["config"]["raster_lanes"], "lanes", d["config"]["launch_groups_per_batch"], "groups", "parity", d["parity"]["frames_checked"], d["parity"]["mismatching_values"],
          {k:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)))
except Exception as e:
    print("no line", e)
PY
done
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1; echo "gpu suite rc=$?" | tee -a $out/summary.txt
tail -5 $out/gpu_tests.txt
