#!/bin/bash
# SQ_INSTS_VALU of the round-3 tree's kernels on this box (for a like-for-like comparison with the current build)
export TMPDIR=/tmp; root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r4_pmc_r03; mkdir -p $out
cd /tmp
for tree in $root/.r03tree $root; do
  args="--steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0"
  if [ "$tree" = "$root" ]; then args="$args --lanes 1 --isolated-seconds 0 --host-copy-seconds 0"; else args="$args --overlap-pipelines 0"; fi
  rm -rf /tmp/rp; (cd $tree && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/rp -o t -- python $tree/bench.py $args > /dev/null 2>&1)
  echo "== $tree" | tee -a $out/summary.txt
  python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1) | grep -A3 "tile_kernel\|setup_kernel<false" | tee -a $out/summary.txt
done
