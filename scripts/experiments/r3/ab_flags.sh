# needs the timing-experiment build: run scripts/build_ablate.sh first (the product library rejects these flag bits)
export RTUF_LIB=${RTUF_LIB:-realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so}
for f in ${FLAGS:-0 0x8000 0 0x8000}; do echo -n "flags=$f "; python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --check-frames 2 --debug-flags $f 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_ms_per_step']['ms_setup'], d['kernel_ms_per_step']['ms_raster'], d['parity'])"; done
