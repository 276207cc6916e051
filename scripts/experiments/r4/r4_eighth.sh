#!/bin/bash
# round 4: bins = direct part + overflow areas.  Parity first, then speed and memory against the direct capacity.
out=gpurun_out/r4_eighth; mkdir -p $out
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/parity.txt 2>&1; tail -4 $out/parity.txt
line() { python - $1 "$2" <<'PY' | tee -a gpurun_out/r4_eighth/summary.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; ks={e["kernel"]:e for e in [r]+r["all_kernels"]}
    print("%-40s" % sys.argv[2], round(d["value"]), "frames/s", "parity", d["parity"]["mismatching_values"], "GB", round(d["config"].get("device_memory_bytes",0)/1e9,2),
          {k.split("_")[0]:(round(v["avg_launch_ms"]*1e3,1), round((v.get("in_headline_run") or {}).get("avg_launch_ms",0)*1e3,1)) for k,v in ks.items()}, "one-lane", round((r.get("one_lane_leg") or {}).get("frames_per_s",0)))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
}
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 1.5"
for v in "" "--bin-capacity 512" "--bin-capacity 2048" "--bin-capacity 256" "--bin-capacity 6144" "--near-arm --steps 40" "--near-arm --steps 40 --bin-capacity 512" "--workload c4 --shard-of 8 --steps 50" "--workload c4 --shard-of 8 --steps 50 --bin-capacity 512" "--workload c5 --shard-of 8 --steps 30" "--workload c5 --shard-of 8 --steps 30 --bin-capacity 512" "--u16" "--two-kernel" "--streams 1 --steps 500"; do
  tag=$(echo "$v" | tr -d ' -'); timeout 600 python bench.py $Q $v > $out/bench_$tag.json 2> $out/bench_$tag.err; line $out/bench_$tag.json "[$v]"
done
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_parity_gpu.py > $out/gpu_tests.txt 2>&1; tail -3 $out/gpu_tests.txt
timeout 600 python scripts/fuzz_parity.py 3000 $((777 + RANDOM)) 2>&1 | tail -1; timeout 600 python scripts/fuzz_features.py 1500 $((888 + RANDOM)) 2>&1 | tail -1
