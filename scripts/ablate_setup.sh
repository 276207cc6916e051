# needs the timing-experiment build: run scripts/build_ablate.sh first (the product library rejects these flag bits)
export RTUF_LIB=${RTUF_LIB:-realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so}
# timing experiments of the set-up kernel (results are wrong with these flags): see SetupArgs.flags
for f in 0 0x10000 0x20000 0x40000 0x60000; do
 echo -n "flags=$f "; python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --check-frames 0 --debug-flags $f 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step']['ms_setup'], d['kernel_ms_per_step']['ms_raster'])"
done
