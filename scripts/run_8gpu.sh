#!/bin/bash
# The two multi-GPU hosts on every GPU of this node.  No data-path collective in either: streams
# are sharded, RCCL carries barriers, the MAX of the elapsed time and a few counters per rank (plus, in the C++ host, the
# optional all-gather of the bit-packed masks).  usage: scripts/run_8gpu.sh [N]   (default: every visible device)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root"
N=${1:-$(python -c 'import torch; print(torch.cuda.device_count())')}
export HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. Python host: one process per GPU over torch.distributed (backend "nccl" = RCCL), the three BASELINE configs
for wl in c3 c4 c5; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus "$N" --workload $wl --cpu-seconds 0
done
# 2. C++ host: one process, one thread + one rtuf_context per GPU (include/realtime_urdf_filter_amd/multi_gpu.hpp), both
#    partitions, masks gathered peer to peer over xGMI (falls back to RCCL, and says so, where a pair has no peer access)
[ -x examples/bin/multi_gpu_filter ] || realtime_urdf_filter_amd/csrc/build_facade.sh
python - <<'PY'
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import scene_file
from bench_support import configs as CF
scene_file.write_scene("/tmp/rtuf_c4_scene.bin", CF.build("c4", 1, 0, streams=64, triangles=250000), k=0)
scene_file.write_scene("/tmp/rtuf_c5_scene.bin", CF.build("c5", 1, 0, streams=16, urdfs=8, triangles=250000), k=0)
PY
examples/bin/multi_gpu_filter /tmp/rtuf_c4_scene.bin --all-devices --mode block --steps 20 --masks direct
examples/bin/multi_gpu_filter /tmp/rtuf_c5_scene.bin --all-devices --mode model --steps 20 --masks rccl
