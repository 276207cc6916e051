#!/bin/bash
# A/B of library variants on one GPU box: frames/s + kernel times of the headline workload (c3), config 4's share, the
# arm-in-front-of-the-lens pose of c3, and (optionally) scripts/clip_stress.py.
#   usage: scripts/ab_round3.sh [variant ...]      variant = name under realtime_urdf_filter_amd/lib/variants, or "default"
#   env:   WORKLOADS="c3 c4 near c5"  STEPS=60  STRESS=1
line() {
python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k={e['kernel']:e for e in [r]+r['all_kernels']}
t=[k[n]['avg_launch_ms'] for n in k if n.startswith('tile')][0]
print('%8.0f frames/s  step %.4f ms  tile %.4f  setup %.4f  clip+big %.4f  frac %.3f  parity %d/%d  entries %d frags %d fill %d/%d' % (d['value'], d['ms_per_step'], t, k['setup_kernel']['avg_launch_ms'], k['clip_kernel']['avg_launch_ms'], r['frac'], d['parity']['mismatching_values'], d['parity']['frames_checked'], d['rasteriser']['bin_entries'], d['rasteriser']['fragments_binned'], d['rasteriser']['max_bin_fill'], d['rasteriser']['max_fragment_bin_fill']), 'mem %.2f GB' % (d.get('device_memory_bytes', 0) / 1e9))
"
}
for v in "${@:-default}"; do
  if [ "$v" = default ]; then unset RTUF_LIB; else export RTUF_LIB=$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_$v.so; fi
  for w in ${WORKLOADS:-c3 c4 near}; do
    case $w in
      c3) args="";;
      c4) args="--workload c4 --shard-of 8";;
      c5) args="--workload c5 --shard-of 8";;
      near) args="--near-arm";;
      near4) args="--workload c4 --shard-of 8 --near-arm";;
    esac
    echo -n "$v $w: "
    python bench.py --steps ${STEPS:-60} --warmup 3 --cpu-seconds 0 --check-frames 2 --overlap-pipelines 0 $args 2>gpurun_out/ab_err_${v}_$w.txt | line || tail -3 gpurun_out/ab_err_${v}_$w.txt
  done
  if [ -n "$STRESS" ]; then echo "$v clip_stress:"; python scripts/clip_stress.py 2>&1 | tail -12; fi
done
