# usage: [BENCH_ARGS='--workload c4 --shard-of 8'] [FLAGS='0 0x800000 0x100000 0x200000'] scripts/ablate_clip.sh
# needs the timing-experiment build: run scripts/build_ablate.sh first (the product library rejects these flag bits)
export RTUF_LIB=${RTUF_LIB:-$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so}
export TMPDIR=/tmp
root=$PWD
for f in ${FLAGS:-0 0x100000 0x200000}; do
cd /tmp; rm -rf /tmp/rp; rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -o t -- python $root/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --check-frames 0 --min-seconds 0 --overlap-pipelines 0 --debug-flags $f $BENCH_ARGS > /dev/null 2>&1
cd $root; echo -n "flags $f: "; python scripts/prof_summary.py $(find /tmp/rp -name '*kernel_trace.csv' | head -1) | grep clip_kernel
done
