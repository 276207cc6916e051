#!/bin/bash
# round 5: threads per tile workgroup (256 in the product) for ONE camera stream (BASELINE config 2) -- is the batch-1 tile kernel's
# time its 150 workgroups' serial window loops?
out=gpurun_out/r5q; mkdir -p $out
here=$PWD; src=$here/realtime_urdf_filter_amd/csrc; mkdir -p $here/realtime_urdf_filter_amd/lib/variants
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result $2 $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $here/realtime_urdf_filter_amd/lib/variants/librtuf_$1.so; }
build never -DRTUF_SMALL_LAUNCH=0; build upto4096 -DRTUF_SMALL_LAUNCH=4096
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=d['kernel_ms_per_step']
print('%8.0f frames/s  %.1f us/step  tile %.1f us  stages pose %.1f setup %.1f raster %.1f total %.1f us  mismatches %s' % (d['value'], d['ms_per_step']*1e3, r['avg_launch_ms']*1e3, k['ms_pose']*1e3, k['ms_setup']*1e3, k['ms_raster']*1e3, k['ms_total']*1e3, d['parity']['mismatching_values']))"; }
for rep in 1 2; do
for v in base never upto4096; do
  lib=$here/realtime_urdf_filter_amd/lib/librtuf.so; [ $v != base ] && lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_$v.so
  for a in "--streams 1 --steps 500" "--streams 1 --steps 500 --lanes 1" "--streams 8 --steps 200" "--streams 16 --steps 200" ""; do
    printf "%-6s %-36s " $v "$a"; RTUF_LIB=$lib python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 1.5 --check-frames 1 --isolated-seconds 1 --other-configs off $a 2>/dev/null | line
  done
done
done 2>&1 | tee $out/ab.txt
