# kernel trace of the default bench (run on the GPU box); summary -> gpurun_out/kernel_summary.txt
export TMPDIR=/tmp
root=$PWD
mkdir -p $root/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof -o trace -- python $root/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --check-frames 0 "$@" > $root/gpurun_out/prof/bench.log 2>&1
cd $root
f=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
python scripts/prof_summary.py $f > gpurun_out/kernel_summary.txt
cat gpurun_out/kernel_summary.txt
