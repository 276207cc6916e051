#!/bin/bash
# Registers / LDS / occupancy of every kernel as the compiler reports them (no GPU needed).
# usage: scripts/kernel_resources.sh [extra hipcc flags, e.g. -DRTUF_TILE_H=16]
here="$(cd "$(dirname "$0")/../realtime_urdf_filter_amd/csrc" && pwd)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I"$here/../../include" -I"$here" "$@" \
  -c "$here/rtuf_kernels.hip" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: +(\S[^:]*): (.*?)\s*\[-Rpass", line) or re.search(r"remark: +(\S[^:]*): (.*?)\s*\[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name" or k == "Name":
        if cur: print(cur)
        cur = {"kernel": v.replace("_ZN4rtuf", "")}
    elif k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        cur[k.split(" [")[0]] = int(v)
if cur: print(cur)
'
