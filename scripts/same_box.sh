#!/bin/bash
# Two trees on ONE box: bench lines of a checkout beside the working tree (e.g. `git worktree add .r05tree <round-5 commit>`,
# built there), alternating, so that box-to-box differences (this pool: up to 20 % on the tile kernel) cancel.
#   usage: scripts/same_box.sh .r05tree [bench args...]
here="$(cd "$(dirname "$0")/.." && pwd)"
other=$here/$1; shift
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 3 --check-frames 8 --isolated-seconds 1.5 --other-configs off"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%9.0f frames/s  tile %.4f ms  mem %.2f GB  mismatches %s  %s' % (d['value'], r['avg_launch_ms'], d['device_memory_bytes']/1e9, d['parity']['mismatching_values'], d.get('step_ms')))"; }
for rep in 1 2; do
  printf "%-10s " other; (cd $other && python bench.py $Q "$@" 2>/dev/null | line)
  printf "%-10s " this;  (cd $here && python bench.py $Q "$@" 2>/dev/null | line)
  printf "%-10s " this-v2; (cd $here && python bench.py $Q --variants 2 "$@" 2>/dev/null | line)
done
