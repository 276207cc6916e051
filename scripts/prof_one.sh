#!/bin/bash
# rocprofv3 kernel trace of one bench configuration -> gpurun_out/<name>_kernel_stats.csv + a short summary on stdout
#   usage: scripts/prof_one.sh <name> [bench args...]
name=$1; shift
export TMPDIR=/tmp
root=$PWD
cd /tmp; rm -rf /tmp/rp_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o t -- python $root/bench.py --steps 40 --warmup 3 --cpu-seconds 0 --check-frames 0 --overlap-pipelines 0 "$@" > /dev/null 2>&1
f=$(find /tmp/rp_$name -name '*kernel_stats.csv' | head -1)
cp $f $root/gpurun_out/${name}_kernel_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-60s calls %6s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
