#!/bin/bash
out=gpurun_out/r4_torchrun3; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 0 --check-frames 4"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), d['ms_per_step'], d['config']['lanes_side_by_side'])" $1 "$2" | tee -a $out/summary.txt; }
python bench.py $Q > $out/plain.json 2>/dev/null; show $out/plain.json "plain lanes2"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 $Q 2>/dev/null | grep '^{' | tail -1 > $out/tr2.json; show $out/tr2.json "torchrun nccl lanes2"
GPU_MAX_HW_QUEUES=1 python bench.py $Q > $out/q1.json 2>/dev/null; show $out/q1.json "plain lanes2 GPU_MAX_HW_QUEUES=1 (no pair can overlap)"
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "raster_lanes or small_batches or memory_limit or two_batches or pipelines" 2>&1 | tail -2 | tee -a $out/summary.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/rp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/rp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --min-seconds 0 --cpu-seconds 0 --check-frames 0 --lanes 1 --isolated-seconds 0 --host-copy-seconds 0 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find /tmp/rp -name '*counter_collection.csv' | head -1) | tee $GRAFT_REPO_ROOT/$out/pmc_valu.txt | head -40
