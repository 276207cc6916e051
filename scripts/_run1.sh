set -x
mkdir -p gpurun_out/r2a
scripts/bin/valu_peak 8 > gpurun_out/r2a/valu_peak_w8.json 2> gpurun_out/r2a/valu_peak.err
scripts/bin/valu_peak 4 > gpurun_out/r2a/valu_peak_w4.json 2>> gpurun_out/r2a/valu_peak.err
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/gpu_tests.txt
tail -25 gpurun_out/r2a/gpu_tests.txt
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; tail -c 1500 gpurun_out/r2a/bench.json; tail -5 gpurun_out/r2a/bench.err
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|NUMA node\(s\)"; ls /usr/lib/x86_64-linux-gnu/dri/ 2>&1 | head
