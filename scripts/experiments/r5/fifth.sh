#!/bin/bash
# round 5: what the strip class costs -- timing experiments (RTUF_ABLATE build, images wrong by design) on the near-arm pose, C4 and C5 shares, one lane
out=gpurun_out/r5e; mkdir -p $out
bash scripts/build_ablate.sh
export RTUF_LIB=$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so
for w in near c4 c5 c3; do
  case $w in c3) a="" ;; near) a="--near-arm --steps 40" ;; c4) a="--workload c4 --shard-of 8 --steps 50" ;; c5) a="--workload c5 --shard-of 8 --steps 30" ;; esac
  for f in 0 0x4000 0x2000000 0x2000 0x1000 0x200 0x800; do
    printf "%-5s flags=%-10s " $w $f
    python bench.py --lanes 1 --warmup 3 --cpu-seconds 0 --host-copy-seconds 0 --check-frames 0 --min-seconds 1 --isolated-seconds 0 --other-configs off --debug-flags $f $a 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; ks={e['kernel']:e['avg_launch_ms'] for e in [r]+r['all_kernels']}; print(round(d['value']), {k:round(v*1e3,1) for k,v in ks.items()})"
  done
done 2>&1 | tee $out/ablate.txt
