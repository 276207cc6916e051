#!/bin/bash
# Overdraw of the tile kernel: depth tests issued (LDS atomics of the walks, the fragment lists and the whole-tile passes)
# per pixel that ends up drawn -- counted by an instrumented library (-DRTUF_COUNT; never the product).
#   usage (GPU box): scripts/overdraw.sh > profiles/overdraw.json
here="$(cd "$(dirname "$0")/.." && pwd)"
lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_count.so
src=$here/realtime_urdf_filter_amd/csrc
if [ ! -f $lib ] || [ $src/rtuf_kernels.hip -nt $lib ] || [ $src/rtuf_api.cpp -nt $lib ] || [ $src/rtuf_device.h -nt $lib ] || [ $here/include/rtuf.h -nt $lib ]; then
  (cd $here/realtime_urdf_filter_amd/csrc && mkdir -p ../lib/variants && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I../../include -I. \
     -Wno-unused-value -Wno-unused-result -DRTUF_COUNT rtuf_kernels.hip rtuf_api.cpp -o $lib) || exit 1
fi
RTUF_LIB=$lib python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import realtime_urdf_filter_amd as R
from bench_support import configs as CF
out = {}
for key, kw in (("c3", dict(workload="c3")), ("near_arm", dict(workload="c3", near_arm=True)), ("c4", dict(workload="c4", world=8)), ("c5", dict(workload="c5", world=8))):
    share = CF.build(kw["workload"], kw.get("world", 1), 0, near_arm=kw.get("near_arm", False))
    n, W, H = share.n, share.width, share.height
    p = R.default_params(); p.filter_replace_value = share.wl0.replace_value; p.depth_distance_threshold = share.wl0.max_diff
    ctx = R.Context(W, H, n, 0, p)
    share.load(ctx)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(share.depth_host(0)).to(dev)
    m = torch.empty((n, H, W), dtype=torch.float32, device=dev); k = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for step in (0, 1):
        share.stage(ctx, step)
        ctx.filter_batch_device(n, d.data_ptr(), m.data_ptr(), k.data_ptr()); ctx.sync()
    st = ctx.stats()
    out[key] = {"depth_tests_per_launch": st["raster_atomics"], "drawn_pixels_per_launch": st["drawn_pixels"],
                "depth_tests_per_drawn_pixel": st["raster_atomics"] / max(st["drawn_pixels"], 1),
                "drawn_fraction_of_frame": st["drawn_pixels"] / float(n * W * H), "streams": n, "size": [W, H]}
    ctx.close(); del d, m, k
print(json.dumps(out, indent=1))
PY
