#!/bin/bash
# Live lanes per issued trip of the walk / emission loops of tile_kernel and setup_kernel -- counted by an instrumented
# library (-DRTUF_LANECOUNT -DRTUF_COUNT; never the product): every instrumented loop adds 64 lane slots per wave and trip and
# the lanes that were live in it (enum kLane* in csrc/rtuf_kernels.hip).
#   usage (GPU box): scripts/lane_util.sh > profiles/r05_pmc_lanes.txt
here="$(cd "$(dirname "$0")/.." && pwd)"
lib=$here/realtime_urdf_filter_amd/lib/variants/librtuf_lanes.so
src=$here/realtime_urdf_filter_amd/csrc
# (always rebuilt: 20 s, and a library that travelled to the box keeps no usable time stamp; LANE_FLAGS adds -D switches)
(cd $here/realtime_urdf_filter_amd/csrc && mkdir -p ../lib/variants && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I../../include -I. \
   -Wno-unused-value -Wno-unused-result -DRTUF_LANECOUNT -DRTUF_COUNT $LANE_FLAGS rtuf_kernels.hip rtuf_api.cpp -o $lib) || exit 1
RTUF_LIB=$lib python - "$@" <<'PY'
import ctypes, json, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import realtime_urdf_filter_amd as R
from realtime_urdf_filter_amd import _capi
from bench_support import configs as CF
NAMES = ["tile: record load+unpack", "tile: lane-walk quad trips", "tile: lane-walk depth-test bodies", "tile: quarter-wave pair trips",
         "tile: quarter-wave depth-test bodies", "tile: whole-wave pair trips", "tile: whole-wave depth-test bodies",
         "tile: parked (workgroup) trips", "tile: parked depth-test bodies", "tile: fragment list", "tile: resolve passes",
         "setup: phase 1 vertices", "setup: phase 2 triangles", "setup: small boxes (coverage)", "setup: small boxes that cover a pixel",
         "setup: fragment store loop", "setup: fragment group-finding trips", "setup: records (set-up)", "setup: records that survive",
         "setup: record tile trips", "setup: record group-finding trips"]
lib = _capi.load_library()
lib.rtuf_debug_lane_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.rtuf_debug_lane_counts.restype = ctypes.c_int
out = {}
cases = (("c3", dict(workload="c3")), ("c3_near_arm", dict(workload="c3", near_arm=True)), ("c4_share", dict(workload="c4", world=8)), ("c5_share", dict(workload="c5", world=8)))
if os.environ.get("LANE_CASES"):
    cases = tuple(c for c in cases if c[0] in os.environ["LANE_CASES"].split())
for key, kw in cases:
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
    slots = (ctypes.c_uint64 * 48)(); live = (ctypes.c_uint64 * 48)()
    lib.rtuf_debug_lane_counts(ctx._h, slots, live, 48)
    rows = {}
    for i, name in enumerate(NAMES):
        if slots[i]:
            rows[name] = {"wave_trips": int(slots[i]) // 64, "live_lanes": int(live[i]), "live_fraction": round(live[i] / slots[i], 4)}
    hist_names = ["1", "2", "3", "4", "5", "6", "7-8", "9-12", "13-16", "17-24", "25+", "quarter-wave class", "larger", "nothing in this tile",
                  "whole box <= 8x8 in one tile, not near", "whole box <= 8x4 / 4x8 in one tile, not near"]
    hist = {hist_names[i]: int(live[24 + i]) for i in range(len(hist_names))}
    out.setdefault("_hist", {})[key] = hist
    out[key] = {"streams": n, "size": [W, H], "bin_entries": st["bin_entries"], "fragments_binned": st["fragments_binned"],
                "depth_tests": st["raster_atomics"], "drawn_pixels": st["drawn_pixels"], "loops": rows}
    ctx.close(); del d, m, k
hist = out.pop("_hist", {})
for k in hist:
    out[k]["records_by_quad_trips_of_their_walk"] = hist[k]
print(json.dumps(out, indent=1))
PY
