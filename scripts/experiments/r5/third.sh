#!/bin/bash
# round 5: fragment-list slots skipped per wave -- before (lib/variants/librtuf_before.so, built in the container) against after
out=gpurun_out/r5h; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --check-frames 8 --isolated-seconds 1.5 --other-configs off"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%9.0f frames/s  tile %.4f ms  mismatches %s' % (d['value'], r['avg_launch_ms'], d['parity']['mismatching_values']))"; }
for rep in 1 2; do
for v in before after; do
  lib=$PWD/realtime_urdf_filter_amd/lib/librtuf.so; [ $v = before ] && lib=$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_before.so
  for w in c3 near c4 c5; do
    case $w in c3) a="" ;; near) a="--near-arm --steps 40" ;; c4) a="--workload c4 --shard-of 8 --steps 50" ;; c5) a="--workload c5 --shard-of 8 --steps 30" ;; esac
    printf "%-8s %-5s " $v $w; RTUF_LIB=$lib python bench.py $Q $a 2>/dev/null | line
  done
done
done 2>&1 | tee $out/ab.txt
python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $out/gpu_suite.txt
