#!/bin/bash
# the pose stage's stream probed against the lanes' as well: plain process against a process with an RCCL communicator
out=gpurun_out/r4_torchrun4; mkdir -p $out
Q="--cpu-seconds 0 --host-copy-seconds 2 --min-seconds 3 --isolated-seconds 1 --check-frames 8"
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']; ks={e['kernel']:e for e in [r]+r['all_kernels']}
print(sys.argv[2], round(d['value']), d['ms_per_step'], d['config']['lanes_side_by_side'], {k.split('_')[0]:(round(v['avg_launch_ms']*1e3,1), round((v.get('in_headline_run') or {}).get('avg_launch_ms',0)*1e3,1)) for k,v in ks.items()}, 'one-lane', round((r.get('one_lane_leg') or {}).get('frames_per_s',0)), 'host copies', {k: round(v.get('frames_per_s',0)) for k,v in (d.get('with_host_copies') or {}).items() if isinstance(v, dict)})" $1 "$2" | tee -a $out/summary.txt; }
for rep in 1 2; do
python bench.py $Q > $out/plain.json 2>/dev/null; show $out/plain.json "plain"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2956$rep bench.py --gpus 1 $Q 2>/dev/null | grep '^{' | tail -1 > $out/tr.json; show $out/tr.json "torchrun nccl"
done
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "lanes or hardware_queue or asynchronous or pipelines or small_batches" 2>&1 | tail -2 | tee -a $out/summary.txt
