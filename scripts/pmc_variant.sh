#!/bin/bash
# Instruction counters of the kernels for library VARIANTS built on the GPU box from -D switches: one rocprofv3 --pmc pass per
# counter group (<= 4 counters), per variant, of bench.py with BENCH_ARGS on the one-lane launch shape.
#   usage: BENCH_ARGS="--near-arm" scripts/pmc_variant.sh "name[:-DFLAG=..]" ...   (name "base" = the product library as built)
#   env:   GROUPS_="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES|SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
here="$(cd "$(dirname "$0")/.." && pwd)"
src=$here/realtime_urdf_filter_amd/csrc
mkdir -p $here/realtime_urdf_filter_amd/lib/variants
export TMPDIR=/tmp
GROUPS_=${GROUPS_:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES|SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"}
for spec in "$@"; do
  name=${spec%%:*}; flags=""; [ "$spec" != "$name" ] && flags=${spec#*:}
  lib=$here/realtime_urdf_filter_amd/lib/librtuf.so
  if [ "$name" != base ]; then
    lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_$name.so
    [ -f $lib ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result $flags \
      $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $lib || { echo "$name: build failed"; continue; }
  fi
  echo "# variant $name ($flags)  bench.py --lanes 1 $BENCH_ARGS"
  IFS='|' read -ra GRP <<< "$GROUPS_"
  for grp in "${GRP[@]}"; do
    echo "## $grp"
    (cd /tmp && rm -rf /tmp/rp && RTUF_LIB=$lib rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp -o t -- python $here/bench.py --steps 10 --warmup 2 --min-seconds 0 --cpu-seconds 0 \
       --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0 --other-configs off $BENCH_ARGS > /dev/null 2>&1
     python $here/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1) | grep -A6 "tile_kernel\|setup_kernel<false")
  done
done
