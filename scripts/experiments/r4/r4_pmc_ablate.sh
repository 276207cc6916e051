#!/bin/bash
# where the tile kernel's instructions are: SQ_INSTS_VALU under the ablation flags, round-3 tree vs current tree, same box
export TMPDIR=/tmp; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r4_pmc_ablate; mkdir -p $out
cd /tmp
for f in 0 0x100 0xc00 0x800 0x400 0x200 0x1000; do
for tree in $root/.r03tree $root; do
  args="--steps 12 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --debug-flags $f"
  if [ "$tree" = "$root" ]; then args="$args --lanes 1 --isolated-seconds 0 --host-copy-seconds 0"; else args="$args --overlap-pipelines 0"; fi
  rm -rf /tmp/rp; (cd $tree && RTUF_LIB=$tree/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/rp -o t -- python $tree/bench.py $args > /dev/null 2>&1)
  v=$(python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1) | grep -A1 "tile_kernel<false, false, false>" | tail -1)
  echo "flags=$f $(basename $tree) $v" | tee -a $out/summary.txt
done; done
