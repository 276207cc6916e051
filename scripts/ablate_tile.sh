# usage: [BENCH_ARGS='--workload c4 --shard-of 8'] scripts/ablate_tile.sh
# needs the timing-experiment build: run scripts/build_ablate.sh first (the product library rejects these flag bits)
export RTUF_LIB=${RTUF_LIB:-realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so}
for f in 0 0x100 0xc00 0x800 0x400 0x200 0x1000 0x2000 0x3000; do
 echo -n "flags=$f "; python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --check-frames 0 --debug-flags $f $BENCH_ARGS 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_per_step'], d['rasteriser']['fragments_binned'], d['rasteriser']['max_fragment_bin_fill'])"
done
