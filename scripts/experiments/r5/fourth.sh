#!/bin/bash
# round 5: the row-stepped strip walk against the linear pair runs (RTUF_STRIP_WALK=0) -- counters and per-loop lane counts, near-arm pose and C4 share
out=gpurun_out/r5d; mkdir -p $out
here=$PWD; src=$here/realtime_urdf_filter_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result -DRTUF_STRIP_WALK=0 \
  $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $here/realtime_urdf_filter_amd/lib/variants/librtuf_old.so
for v in new old; do
  lib=$here/realtime_urdf_filter_amd/lib/librtuf.so; [ $v = old ] && lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_old.so
  for w in near c4; do
    case $w in near) a="--near-arm --lanes 1" ;; c4) a="--workload c4 --shard-of 8 --lanes 1" ;; esac
    echo "== $v $w"
    RTUF_LIB=$lib BENCH_ARGS="$a --isolated-seconds 0 --host-copy-seconds 0 --min-seconds 0" bash scripts/pmc_kernels.sh SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU 2>&1 | grep -A5 "tile_kernel"
  done
done 2>&1 | tee $out/pmc.txt
LANE_CASES="c3_near_arm c4_share" bash scripts/lane_util.sh > $out/lanes_new.json 2> $out/lanes.err
LANE_FLAGS="-DRTUF_STRIP_WALK=0" LANE_CASES="c3_near_arm" bash scripts/lane_util.sh > $out/lanes_old.json 2>> $out/lanes.err
