#!/bin/bash
out=gpurun_out/r4_torchrun2; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 0 --check-frames 4"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), d['ms_per_step'])" $1 "$2" | tee -a $out/summary.txt; }
for q in 2 8 16; do
GPU_MAX_HW_QUEUES=$q python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2956$q bench.py --gpus 1 $Q 2>/dev/null | grep '^{' | tail -1 > $out/tr_q$q.json; show $out/tr_q$q.json "torchrun nccl lanes2 GPU_MAX_HW_QUEUES=$q"
done
GPU_MAX_HW_QUEUES=8 python bench.py $Q > $out/plain_q8.json 2>/dev/null; show $out/plain_q8.json "plain lanes2 GPU_MAX_HW_QUEUES=8"
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 $Q > $out/nccl_info.txt 2>&1; grep -i "stream\|queue\|thread\|priority" $out/nccl_info.txt | head -20 | tee -a $out/summary.txt
