# needs the timing-experiment build: run scripts/build_ablate.sh first (the product library rejects these flag bits)
# tile / set-up / clip kernel breakdown on BASELINE config 4's per-GPU share (64 x 720p, robot + two wall URDFs)
export RTUF_LIB=${RTUF_LIB:-realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so}
for f in 0 0x100 0xc00 0x800 0x400 0x200 0x1000 0x2000 0x3000 0x10000 0x60000 0x100000 0x200000 0x800000; do
 echo -n "flags=$f "; python bench.py --workload c4 --shard-of 8 --steps 30 --warmup 3 --cpu-seconds 0 --check-frames 0 --overlap-pipelines 0 --debug-flags $f 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; ks={e['kernel']:e['avg_launch_ms'] for e in [r]+r['all_kernels']}; print(round(d['value']), {k:round(v*1e3,1) for k,v in ks.items()})"
done
