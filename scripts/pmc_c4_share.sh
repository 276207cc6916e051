#!/bin/bash
# Counters of config 4's per-GPU share alone (the same passes scripts/refresh_profiles.sh makes for it):
#   usage (GPU box): scripts/pmc_c4_share.sh [tag]  ->  gpurun_out/profiles/<tag>_pmc_workload_c4_shard_of_8.txt
tag=${1:-r03}
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/profiles; mkdir -p $out
B="python $root/bench.py"
PROF_ARGS="--steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --overlap-pipelines 0"
mode="--workload c4 --shard-of 8"
f=$out/${tag}_pmc_workload_c4_shard_of_8.txt
cd /tmp
echo "# rocprofv3 --kernel-trace --pmc <group> (one pass per '##' group), command: python bench.py $PROF_ARGS $mode   (values per launch, averaged over the launches of the run; FETCH_SIZE / WRITE_SIZE in KiB)" > $f
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"; do
  rm -rf /tmp/rp; timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp -o t -- $B $PROF_ARGS $mode > /dev/null 2>&1
  { echo "## $grp"; python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1); } >> $f
done
cat $f
