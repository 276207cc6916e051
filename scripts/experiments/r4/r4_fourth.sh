#!/bin/bash
out=gpurun_out/r4_fourth; mkdir -p $out
line() { python - "$1" <<'PY' | tee -a gpurun_out/r4_fourth/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}
    print(round(d["value"]), "frames/s", round(d["device_memory_bytes"]/1e9,2), "GB", d["config"]["raster_lanes"], "lanes", d["config"]["launch_groups_per_batch"], "groups", "parity", d["parity"]["frames_checked"], d["parity"]["mismatching_values"],
          {k.split("_")[0]:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)), "exact tiles", d["rasteriser"]["tile"]["exact_z_tiles"], "hiz", d["rasteriser"]["tile"].get("hiz_culled_entries"), "of", d["rasteriser"]["bin_entries"])
except Exception as e:
    print("no line", e)
PY
}
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q > $out/parity_tests.txt 2>&1; echo "parity tests rc=$?" | tee -a $out/summary.txt; tail -3 $out/parity_tests.txt
for lib in "" realtime_urdf_filter_amd/lib/variants/librtuf_facing1.so; do
for v in "" "--near-arm" "--workload c4 --shard-of 8" "--workload c5 --shard-of 8"; do
  tag=$(echo "$(basename ${lib:-default} .so)$v" | tr -d ' -')
  RTUF_LIB=$lib timeout 600 python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1 $v > $out/bench_$tag.json 2> $out/bench_$tag.err; echo "bench [$lib $v] rc=$?" | tee -a $out/summary.txt
  line $out/bench_$tag.json
done; done
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1; echo "gpu suite rc=$?" | tee -a $out/summary.txt
tail -5 $out/gpu_tests.txt
python scripts/fuzz_parity.py 3000 424242 > $out/fuzz_parity.txt 2>&1; echo "fuzz_parity rc=$?" | tee -a $out/summary.txt; tail -2 $out/fuzz_parity.txt
python scripts/fuzz_features.py 1500 515151 > $out/fuzz_features.txt 2>&1; echo "fuzz_features rc=$?" | tee -a $out/summary.txt; tail -2 $out/fuzz_features.txt
python scripts/clip_stress.py > $out/clip_stress.txt 2>&1; echo "clip_stress rc=$?" | tee -a $out/summary.txt; tail -8 $out/clip_stress.txt
