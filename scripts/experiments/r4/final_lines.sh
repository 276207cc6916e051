set -x
out=gpurun_out/profiles; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $out/r03_gpu_tests.txt; cat $out/r03_gpu_tests.txt
timeout 300 python bench.py 2>>$out/bench.err | grep '^{' | tail -1 > $out/r03_bench.json
timeout 300 python bench.py --workload c4 --shard-of 8 --cpu-seconds 0 2>>$out/bench.err | grep '^{' | tail -1 > $out/r03_bench_c4_share.json
timeout 300 python bench.py --near-arm --cpu-seconds 0 2>>$out/bench.err | grep '^{' | tail -1 > $out/r03_bench_near_arm.json
timeout 300 python bench.py --workload c5 --shard-of 8 --cpu-seconds 0 2>>$out/bench.err | grep '^{' | tail -1 > $out/r03_bench_c5_share.json
for f in r03_bench r03_bench_c4_share r03_bench_near_arm r03_bench_c5_share; do python - $out/$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1], round(d["value"]), d["ms_per_step"], r["kernel"], r["avg_launch_ms"], round(r["frac"],3), r.get("traffic"), d["rasteriser"]["tile"].get("overdraw"), d["parity"]["frames_checked"], d["parity"]["mismatching_values"])
PY
done
