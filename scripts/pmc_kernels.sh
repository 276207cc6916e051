#!/bin/bash
# usage: bash scripts/pmc_kernels.sh COUNTER [COUNTER...]   (one rocprofv3 --pmc pass per argument group of <=4)
export TMPDIR=/tmp
root=$PWD
cd /tmp
while [ $# -gt 0 ]; do
  grp="$1 $2 $3 $4"; shift; shift 2>/dev/null; shift 2>/dev/null; shift 2>/dev/null
  rm -rf /tmp/rp; rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp -o t -- python $root/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --check-frames 0 $BENCH_ARGS > /dev/null 2>&1
  python $root/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1)
done
