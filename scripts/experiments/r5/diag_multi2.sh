#!/bin/bash
out=gpurun_out/r5v; mkdir -p $out
python - > $out/scene.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from bench_support import configs as CF
import scene_file
share = CF.build("c4", 1, 0, streams=6, triangles=20000, width=640, height=360)
scene_file.write_scene("/tmp/scene.bin", share, k=0)
PY
exe=examples/bin/multi_gpu_filter
[ -x $exe ] || bash realtime_urdf_filter_amd/csrc/build_facade.sh > $out/build.log 2>&1
snap() {   # pid, file
  for t in /proc/$1/task/*; do echo "$(cat $t/comm) $(grep State $t/status | tr -s '\t ' ' ') wchan=$(cat $t/wchan 2>/dev/null)"; done > $2 2>&1
}
NCCL_DEBUG=INFO $exe /tmp/scene.bin --mode block --steps 3 --masks rccl --dump 4 /tmp/s > $out/run_info.txt 2>&1 &
pid=$!; sleep 40
if kill -0 $pid 2>/dev/null; then snap $pid $out/threads_default.txt; kill -9 $pid; echo "HUNG after 40 s" >> $out/run_info.txt; else echo "finished" >> $out/run_info.txt; fi
RTUF_QUEUE_PROBE=0 $exe /tmp/scene.bin --mode block --steps 3 --masks rccl --dump 4 /tmp/s > $out/run_noprobe.txt 2>&1 &
pid=$!; sleep 40
if kill -0 $pid 2>/dev/null; then snap $pid $out/threads_noprobe.txt; kill -9 $pid; echo "HUNG after 40 s" >> $out/run_noprobe.txt; else echo "finished" >> $out/run_noprobe.txt; fi
(timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 3 --warmup 1 --streams 8 --triangles 8000 --cpu-seconds 0 --check-frames 2 --min-seconds 0.2 --isolated-seconds 0 --host-copy-seconds 0 --other-configs off 2>&1 | tail -3 | cut -c1-300) > $out/torch_nccl.txt; echo "rc=$?" >> $out/torch_nccl.txt
tail -25 $out/run_info.txt | cut -c1-220; echo ---; sort $out/threads_default.txt | uniq -c | head -20; echo --- noprobe; tail -3 $out/run_noprobe.txt | cut -c1-200; sort $out/threads_noprobe.txt 2>/dev/null | uniq -c | head; echo --- torch; cat $out/torch_nccl.txt
