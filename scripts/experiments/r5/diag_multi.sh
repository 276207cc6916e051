#!/bin/bash
# round 5: why does examples/multi_gpu_filter --masks direct hang on some boxes?  (diagnosis run)
out=gpurun_out/r5w; mkdir -p $out
python - > $out/scene.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from bench_support import configs as CF
import scene_file
share = CF.build("c4", 1, 0, streams=6, triangles=20000, width=640, height=360)
scene_file.write_scene("/tmp/scene.bin", share, k=0)
print("scene written", share.n)
PY
exe=examples/bin/multi_gpu_filter
[ -x $exe ] || bash realtime_urdf_filter_amd/csrc/build_facade.sh > $out/build.log 2>&1
nproc > $out/host.txt; cat /sys/class/drm/card*/device/local_cpulist >> $out/host.txt 2>&1; ls /sys/class/drm >> $out/host.txt; which gdb >> $out/host.txt
for m in direct rccl direct; do
  echo "== masks $m" >> $out/run.txt
  NCCL_DEBUG=WARN timeout -s INT 60 $exe /tmp/scene.bin --mode block --steps 3 --masks $m --dump 4 /tmp/s >> $out/run.txt 2>&1
  echo "rc=$?" >> $out/run.txt
done
if which gdb > /dev/null 2>&1; then
  $exe /tmp/scene.bin --mode block --steps 3 --masks direct --dump 4 /tmp/s > /dev/null 2>&1 &
  pid=$!; sleep 25
  if kill -0 $pid 2>/dev/null; then gdb -p $pid -batch -ex "thread apply all bt 12" > $out/gdb.txt 2>&1; kill -9 $pid; else echo "finished normally" > $out/gdb.txt; fi
fi
cat $out/scene.log $out/host.txt; tail -40 $out/run.txt; head -80 $out/gdb.txt 2>/dev/null
