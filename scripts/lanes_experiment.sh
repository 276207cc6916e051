here="$(cd "$(dirname "$0")/.." && pwd)"
src=$here/realtime_urdf_filter_amd/csrc
mkdir -p $here/realtime_urdf_filter_amd/lib/variants
lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_lanes6.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$here/include -I$src -Wno-unused-value -Wno-unused-result -DRTUF_MAX_LANES=6 $src/rtuf_kernels.hip $src/rtuf_api.cpp -o $lib || exit 1
Q="--cpu-seconds 0 --host-copy-seconds 0 --min-seconds 3 --check-frames 8 --isolated-seconds 0 --other-configs off"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%9.0f frames/s  lanes %s groups %s side_by_side %s mem %.2f GB mismatches %s' % (d['value'], d['config']['raster_lanes'], d['config']['launch_groups_per_batch'], d['config']['lanes_side_by_side'], d['device_memory_bytes']/1e9, d['parity']['mismatching_values']))"; }
for q in 4 8; do
 for l in 3 4 5 6; do
  printf "GPU_MAX_HW_QUEUES=%d lanes=%d  " $q $l; GPU_MAX_HW_QUEUES=$q RTUF_LIB=$lib python $here/bench.py $Q --lanes $l 2>/dev/null | line
 done
done
printf "default lib, default env       "; python $here/bench.py $Q 2>/dev/null | line
