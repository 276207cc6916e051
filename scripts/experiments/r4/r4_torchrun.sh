#!/bin/bash
out=gpurun_out/r4_torchrun; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 2 --isolated-seconds 0 --check-frames 4"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], round(d['value']), d['ms_per_step'])" $1 "$2" | tee -a $out/summary.txt; }
python bench.py $Q > $out/plain.json 2>/dev/null; show $out/plain.json "plain lanes2"
python bench.py $Q --lanes 1 > $out/plain1.json 2>/dev/null; show $out/plain1.json "plain lanes1"
OMP_NUM_THREADS=1 python bench.py $Q > $out/omp1.json 2>/dev/null; show $out/omp1.json "OMP=1 lanes2"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 $Q 2>/dev/null | grep '^{' | tail -1 > $out/tr2.json; show $out/tr2.json "torchrun nccl lanes2"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 1 $Q --lanes 1 2>/dev/null | grep '^{' | tail -1 > $out/tr1.json; show $out/tr1.json "torchrun nccl lanes1"
RTUF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 1 $Q 2>/dev/null | grep '^{' | tail -1 > $out/trg.json; show $out/trg.json "torchrun gloo lanes2"
