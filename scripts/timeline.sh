#!/bin/bash
# Kernel timeline of a few steady-state steps of one bench configuration (rocprofv3 --kernel-trace): start / end of every
# dispatch relative to the first one shown, its queue, and the idle gap of the whole GPU before it.
#   usage: scripts/timeline.sh <name> [bench args...]          env: SKIP=400 dispatches, SHOW=40
name=$1; shift
export TMPDIR=/tmp
root=$PWD
cd /tmp; rm -rf /tmp/tl_$name
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -o t -- python $root/bench.py --steps 40 --warmup 3 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --overlap-pipelines 0 "$@" > /dev/null 2>&1
f=$(find /tmp/tl_$name -name '*kernel_trace.csv' | head -1)
python - $f ${SKIP:-400} ${SHOW:-40} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip, show = int(sys.argv[2]), int(sys.argv[3])
rows = rows[skip:skip + show]
t0 = int(rows[0]["Start_Timestamp"]); busy_until = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_until); busy_until = max(busy_until, e)
    print("%9.1f %9.1f  dur %8.1f  idle-before %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
PY
