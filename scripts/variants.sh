# usage: scripts/variants.sh name1 name2 ...   (libraries under realtime_urdf_filter_amd/lib/variants)
echo -n "base "; python bench.py --steps ${STEPS:-60} --warmup 3 --cpu-seconds 0 --check-frames 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'tile_ms(timed)', round(d['roofline']['avg_launch_ms'],4), 'setup_stage_ms', round(d['kernel_ms_per_step']['ms_setup'],4), d['parity']['mask_mismatch_pixels'] + d['parity']['depth_mismatch_pixels'])"
for v in "$@"; do
 echo -n "$v "; RTUF_LIB=realtime_urdf_filter_amd/lib/variants/librtuf_$v.so python bench.py --steps ${STEPS:-60} --warmup 3 --cpu-seconds 0 --check-frames 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'tile_ms(timed)', round(d['roofline']['avg_launch_ms'],4), 'setup_stage_ms', round(d['kernel_ms_per_step']['ms_setup'],4), d['parity']['mask_mismatch_pixels'] + d['parity']['depth_mismatch_pixels'])"
done
