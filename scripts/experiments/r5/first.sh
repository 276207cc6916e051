#!/bin/bash
# round 5, first GPU call: lane utilisation (instrumented build), hardware lane counters, baseline bench of the box
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/r5a; mkdir -p $out
bash scripts/lane_util.sh > $out/lane_util.json 2> $out/lane_util.err
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "THREAD_CYCLES|VALUUtil|VALUBusy|ACTIVE_INST_VALU|INSTS_VALU\b" | head -40) > $out/counters_avail.txt 2>&1
BENCH_ARGS="--lanes 1 --min-seconds 0 --isolated-seconds 0 --host-copy-seconds 0" bash scripts/pmc_kernels.sh SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU > $out/pmc_lanes_c3.txt 2>&1
BENCH_ARGS="--lanes 1 --min-seconds 0 --isolated-seconds 0 --host-copy-seconds 0 --near-arm" bash scripts/pmc_kernels.sh SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU > $out/pmc_lanes_near.txt 2>&1
BENCH_ARGS="--lanes 1 --min-seconds 0 --isolated-seconds 0 --host-copy-seconds 0 --workload c4 --shard-of 8" bash scripts/pmc_kernels.sh SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU > $out/pmc_lanes_c4.txt 2>&1
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 3 --near-arm --steps 40 > $out/bench_near.json 2>> $out/bench_default.err
python bench.py --cpu-seconds 0 --host-copy-seconds 0 --min-seconds 3 --workload c4 --shard-of 8 --steps 50 > $out/bench_c4.json 2>> $out/bench_default.err
