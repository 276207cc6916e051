# usage: scripts/coop_variants.sh name1 name2 ...  (libraries under realtime_urdf_filter_amd/lib/variants)
# frames/s of each variant on the headline workload, two low-polygon robots and config 4 (walls)
for v in "$@"; do
  L=realtime_urdf_filter_amd/lib/variants/librtuf_$v.so
  echo -n "$v: "
  for t in 250000 1000 5000 20000; do
    RTUF_LIB=$L python bench.py --triangles $t --steps ${STEPS:-50} --warmup 3 --cpu-seconds 0 --check-frames 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('t$t', round(d['value']), round(d['kernel_ms_per_step']['ms_raster'],4), d['parity']['mask_mismatch_pixels'] + d['parity']['depth_mismatch_pixels'], end=' | ')"
  done
  RTUF_LIB=$L python scripts/baseline_configs.py --only c4 --check 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read())['C4']; print('C4', round(d['frames_per_s']), d['stage_ms_isolated']['ms_raster'], d['mismatching_values'])"
done
