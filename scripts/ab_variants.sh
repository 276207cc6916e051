#!/bin/bash
# A/B of library variants on the GPU box: builds realtime_urdf_filter_amd/lib/variants/librtuf_<name>.so with the given -D flags
# and prints, per variant and workload, frames/s, the tile kernel's and the set-up stage's time per launch and the parity count.
#   usage: scripts/ab_variants.sh "name[:-DFLAG=1 -DOTHER=2]" ...        (name "base" = the product library as built)
#   env:   WORKLOADS="c3 near c4 c5" (default c3 near c4)   LANES=1 (one-lane launch shape; default: the library's lanes)
here="$(cd "$(dirname "$0")/.." && pwd)"
src=$here/realtime_urdf_filter_amd/csrc
mkdir -p $here/realtime_urdf_filter_amd/lib/variants
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds ${SECONDS_EACH:-2} --check-frames ${CHECK:-8} --isolated-seconds 1.5"
[ -n "$LANES" ] && Q="$Q --lanes $LANES"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
ra=d['rasteriser']
print('%9.0f frames/s  tile %.4f ms  setup-stage %.4f ms  mismatches %s  bin entries %d' % (d['value'], r['avg_launch_ms'], d['kernel_ms_per_step']['ms_setup'], d['parity']['mask_mismatch_pixels'] + d['parity']['depth_mismatch_pixels'], ra['bin_entries']))"; }
for spec in "$@"; do
  name=${spec%%:*}; flags=""; [ "$spec" != "$name" ] && flags=${spec#*:}
  lib=$here/realtime_urdf_filter_amd/lib/librtuf.so
  if [ "$name" != base ]; then
    lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_$name.so
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result $flags \
      $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $lib || { echo "$name: build failed"; continue; }
  fi
  for w in ${WORKLOADS:-c3 near c4}; do
    case $w in
      c3) a="" ;; near) a="--near-arm --steps 40" ;; c4) a="--workload c4 --shard-of 8 --steps 50" ;; c5) a="--workload c5 --shard-of 8 --steps 30" ;;
    esac
    printf "%-14s %-5s " "$name" "$w"; RTUF_LIB=$lib python $here/bench.py $Q $a 2>/dev/null | line
  done
done
