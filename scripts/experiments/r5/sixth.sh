#!/bin/bash
# round 5: exact-z pass share with the arm in front of the lens (timing experiment), then the long campaign on the final library:
# soak, 30,000-scene parity fuzz, 10,000-scene feature fuzz, 300 scenes at the largest frame sizes
out=gpurun_out/r5f; mkdir -p $out
bash scripts/build_ablate.sh
for f in 0 0x4000000; do
  printf "near-arm flags=%-10s " $f
  RTUF_LIB=$PWD/realtime_urdf_filter_amd/lib/variants/librtuf_ablate.so python bench.py --near-arm --steps 40 --lanes 1 --warmup 3 --cpu-seconds 0 --host-copy-seconds 0 --check-frames 0 --min-seconds 1 --isolated-seconds 0 --other-configs off --debug-flags $f 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(r['avg_launch_ms']*1e3,1), d['rasteriser'].get('exact_tiles'))"
done 2>&1 | tee $out/exact_pass.txt
bash scripts/soak.sh 2>&1 | tee $out/soak.txt
{ python scripts/fuzz_parity.py 30000 $((20260929 + RANDOM)) 2>&1 | tail -2; python scripts/fuzz_features.py 10000 $((20270100 + RANDOM)) 2>&1 | tail -2; FUZZ_BIG=1 python scripts/fuzz_parity.py 300 $((777 + RANDOM)) 2>&1 | tail -2; } 2>&1 | tee $out/fuzz_long.txt
