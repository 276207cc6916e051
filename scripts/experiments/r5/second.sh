#!/bin/bash
# round 5: what the sorted walk gives and what it costs
out=gpurun_out/r5d; mkdir -p $out
bash scripts/lane_util.sh > $out/lane_util_sorted.json 2> $out/lane_util.err
LANES=1 WORKLOADS="c3 near" scripts/ab_variants.sh base "nosort:-DRTUF_SORT_WALK=0" "sortfake:-DRTUF_SORT_FAKE=1" 2>&1 | tee $out/ab_lane1.txt
export TMPDIR=/tmp
for v in base nosort; do
  lib=realtime_urdf_filter_amd/lib/librtuf.so; [ $v != base ] && lib=realtime_urdf_filter_amd/lib/variants/librtuf_$v.so
  echo "== $v" >> $out/pmc.txt
  RTUF_LIB=$PWD/$lib BENCH_ARGS="--lanes 1 --min-seconds 0 --isolated-seconds 0 --host-copy-seconds 0" bash scripts/pmc_kernels.sh SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY 2>&1 | grep -A4 "tile_kernel<false, false, false>" >> $out/pmc.txt
  RTUF_LIB=$PWD/$lib BENCH_ARGS="--lanes 1 --min-seconds 0 --isolated-seconds 0 --host-copy-seconds 0" bash scripts/pmc_kernels.sh SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES 2>&1 | grep -A4 "tile_kernel<false, false, false>" >> $out/pmc.txt
done
python -m pytest tests/test_batch_status_gpu.py -q -m gpu 2>&1 | tail -5 | tee $out/status_tests.txt
