#!/bin/bash
# How full are the tile bins?  Instrumented libraries (-DRTUF_HIST=<direct capacity>; never the product) count, per launch, the
# (bin, size class) pairs with more than C records, the records beyond C, the fragment bins with more than 4C fragments and
# the fragments beyond 4C.   usage (GPU box): scripts/bin_hist.sh > gpurun_out/bin_hist.txt
here="$(cd "$(dirname "$0")/.." && pwd)"
src=$here/realtime_urdf_filter_amd/csrc
mkdir -p $here/realtime_urdf_filter_amd/lib/variants
for C in 128 256 512 1024; do for back in 0 1; do
  lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_hist_${C}_$back.so
  if [ ! -f $lib ] || [ $src/rtuf_kernels.hip -nt $lib ]; then
    (cd $src && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I../../include -I. -Wno-unused-value -Wno-unused-result \
       -DRTUF_HIST=$C -DRTUF_HIST_BACK=$back rtuf_kernels.hip rtuf_api.cpp -o $lib) || exit 1
  fi
  [ -n "$BUILD_ONLY" ] && continue
  RTUF_LIB=$lib python - $C $back <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import realtime_urdf_filter_amd as R
from bench_support import configs as CF
C, back = int(sys.argv[1]), int(sys.argv[2])
for key, kw in (("c3", dict(workload="c3")), ("near_arm", dict(workload="c3", near_arm=True)), ("c4", dict(workload="c4", world=8)), ("c5", dict(workload="c5", world=8))):
    share = CF.build(kw["workload"], kw.get("world", 1), 0, near_arm=kw.get("near_arm", False))
    n, W, H = share.n, share.width, share.height
    p = R.default_params(); p.filter_replace_value = share.wl0.replace_value; p.depth_distance_threshold = share.wl0.max_diff
    p.raster_lanes = 1
    ctx = R.Context(W, H, n, 0, p)
    share.load(ctx)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(share.depth_host(0)).to(dev)
    m = torch.empty((n, H, W), dtype=torch.float32, device=dev); k = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for step in (0, 1, 2):
        share.stage(ctx, step)
        ctx.filter_batch_device(n, d.data_ptr(), m.data_ptr(), k.data_ptr()); ctx.sync()
    st = ctx.stats()
    r, f = st["raster_atomics"], st["drawn_pixels"]
    tiles = n * ((W + 63) // 64) * ((H + 31) // 32)
    print("%-8s C=%4d %s: %6d of %6d bins over (%5.2f %%), %9d records beyond of %9d bin entries (%5.2f %%) | frags 4C: %6d bins over, %9d beyond of %9d (%5.2f %%) | max fills %d / %d" % (
        key, C, "back " if back else "front", r >> 40, tiles, 100.0 * (r >> 40) / tiles, r & ((1 << 40) - 1), st["bin_entries"], 100.0 * (r & ((1 << 40) - 1)) / max(st["bin_entries"], 1),
        f >> 40, f & ((1 << 40) - 1), st["fragments_binned"], 100.0 * (f & ((1 << 40) - 1)) / max(st["fragments_binned"], 1), st["max_bin_fill"], st["max_fbin_fill"]))
    ctx.close(); del d, m, k
PY
done; done
