#!/bin/bash
# round 5: four size classes per record bin (branch size-classes) against the two of main (lib/variants/librtuf_main.so, built in the container)
out=gpurun_out/r5g; mkdir -p $out
here=$PWD; src=$here/realtime_urdf_filter_amd/csrc
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --check-frames 8 --isolated-seconds 1.5 --other-configs off"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%9.0f frames/s  tile %.4f ms  setup-stage %.4f ms  mem %.2f GB  cap %d  mismatches %s' % (d['value'], r['avg_launch_ms'], d['kernel_ms_per_step']['ms_setup'], d['device_memory_bytes']/1e9, d['rasteriser']['bin_capacity'], d['parity']['mismatching_values']))"; }
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) | tee $out/gputests.txt
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result $2 $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $here/realtime_urdf_filter_amd/lib/variants/librtuf_$1.so; }
build q3_8_16 "-DRTUF_CLASS_Q0=3 -DRTUF_CLASS_Q1=8 -DRTUF_CLASS_Q2=16"
build q2_4_8 "-DRTUF_CLASS_Q0=2 -DRTUF_CLASS_Q1=4 -DRTUF_CLASS_Q2=8"
for rep in 1 2; do
for v in main classes q3_8_16 q2_4_8; do
  lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_$v.so; [ $v = classes ] && lib=$here/realtime_urdf_filter_amd/lib/librtuf.so
  [ $rep = 2 ] && [ $v != main ] && [ $v != classes ] && continue
  for w in c3 near c4 c5; do
    case $w in c3) a="" ;; near) a="--near-arm --steps 40" ;; c4) a="--workload c4 --shard-of 8 --steps 50" ;; c5) a="--workload c5 --shard-of 8 --steps 30" ;; esac
    printf "%-8s %-5s " $v $w; RTUF_LIB=$lib python bench.py $Q $a 2>/dev/null | line
  done
done
done 2>&1 | tee $out/ab.txt
