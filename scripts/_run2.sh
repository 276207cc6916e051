set -x
mkdir -p gpurun_out/r2b
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r2b/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/gpu_tests.txt
tail -22 gpurun_out/r2b/gpu_tests.txt
python bench.py > gpurun_out/r2b/bench_warm.json 2> gpurun_out/r2b/bench.err
python bench.py > gpurun_out/r2b/bench.json 2>> gpurun_out/r2b/bench.err; tail -c 2500 gpurun_out/r2b/bench.json; tail -3 gpurun_out/r2b/bench.err
for P in 1 2 4; do python bench.py --streams 1 --steps 500 --cpu-seconds 0 --pipelines $P --overlap-pipelines 0 > gpurun_out/r2b/bench_batch1_p$P.json 2>> gpurun_out/r2b/bench.err; done
python scripts/host_planes_rate.py > gpurun_out/r2b/host_planes.json 2> gpurun_out/r2b/host_planes.err
python scripts/host_planes_rate.py --pipelines 2 > gpurun_out/r2b/host_planes_p2.json 2>> gpurun_out/r2b/host_planes.err
tail -3 gpurun_out/r2b/host_planes.err
export TMPDIR=/tmp; root=$PWD; cd /tmp
rm -rf /tmp/rp; rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o t -- python $root/bench.py --streams 1 --steps 200 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --overlap-pipelines 0 > /dev/null 2>&1
cp $(find /tmp/rp -name '*kernel_trace.csv' | head -1) $root/gpurun_out/r2b/batch1_kernel_trace.csv
cd $root
python - <<'PY'
import json
for f in ("bench.json","bench_batch1_p1.json","bench_batch1_p2.json","bench_batch1_p4.json"):
    d=json.loads(open("gpurun_out/r2b/"+f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["parity"]["mismatching_values"], d.get("overlapped",{}).get("value"))
for f in ("host_planes.json","host_planes_p2.json"):
    d=json.load(open("gpurun_out/r2b/"+f))
    for k,v in d["modes"].items(): print(f, k, round(v["frames_per_s"]), v.get("mismatching_values", v.get("mismatching_values_after_expansion")), v.get("host_expansion_frames_per_s_per_core"))
PY
