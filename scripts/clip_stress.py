"""Stress of the frustum-crossing paths (run on a GPU box): triangle clouds far away, around the camera, and as a
shell at the near plane (every triangle clipped or closer than 2*near, so the exact-z pass runs), 64 VGA streams
each; prints stage times and counters and checks two streams per case against the oracle."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import scenes as S
import realtime_urdf_filter_amd as R
from oracle import bindings as O
W, H, n = 640, 480, 64
P = S.projection(525.0, 525.0, 319.5, 239.5, W, H)
for name, radius, tri_scale, nt in (("far cloud", 3.0, 0.02, 200000), ("around camera", 0.6, 0.02, 200000), ("around camera, larger", 0.6, 0.1, 50000), ("shell at near plane", 0.12, 0.01, 200000)):
    rng = np.random.default_rng(5)
    if name.startswith("far"):
        centre = rng.normal(size=(nt, 1, 3)) * 0.5 + np.array([0, 0, radius])
    else:
        d = rng.normal(size=(nt, 1, 3)); d /= np.linalg.norm(d, axis=2, keepdims=True)
        centre = d * radius * rng.uniform(0.7, 1.3, size=(nt, 1, 1))
    v = (centre + rng.normal(scale=tri_scale, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
    t = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
    ident = np.eye(4).T.reshape(16)
    depth = np.stack([S.sensor_depth(W, H, 0.1 * s) for s in range(n)])
    dev = torch.device("cuda:0")
    dd = torch.from_numpy(depth).to(dev); dm = torch.empty((n, H, W), dtype=torch.float32, device=dev); dk = torch.empty((n, H, W), dtype=torch.uint8, device=dev)

    def run(lanes):
        """One context with `lanes` raster lanes over the case: (stats of a timed batch, ms per batch with two in flight, cameras)."""
        p = R.default_params(); p.filter_replace_value = 5.0; p.raster_lanes = lanes
        ctx = R.Context(W, H, n, 0, p)
        m = ctx.add_model(); l = ctx.add_link(m); ctx.add_draw(m, l, v, t, 0, [0.0, 0.0, 0.0]); ctx.finalize_models()
        crng = np.random.default_rng(6)
        cams = []
        for s in range(n):
            offinv, camtf = S.random_camera(crng, small=True)
            ctx.set_camera(s, P, offinv, camtf); ctx.set_link_poses(s, m, ident[None]); cams.append((offinv, camtf))
        for _ in range(3):
            ctx.filter_batch_device(n, dd.data_ptr(), dm.data_ptr(), dk.data_ptr()); ctx.sync()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ctx.filter_batch_device(n, dd.data_ptr(), dm.data_ptr(), dk.data_ptr())
        ctx.sync(); per_batch = (time.perf_counter() - t0) / 10 * 1e3
        ctx.enable_timing(1)
        ctx.filter_batch_device(n, dd.data_ptr(), dm.data_ptr(), dk.data_ptr()); ctx.sync()
        st = ctx.stats()
        ctx.close()
        return st, per_batch, cams

    # stage times are per-kernel figures: ONE raster lane (with two, kernels of the two launch groups overlap and the sums
    # would count the overlap twice); the default two-lane context's time per batch is printed beside them
    _, two_lane_ms, _ = run(0)
    st, one_lane_ms, cams = run(1)
    bad = 0
    for s in (0, n - 1):
        om, ok = O.filter_frame(depth[s], P, [(ident, 0, [0.0, 0.0, 0.0], v, t)], cams[s][0], cams[s][1], replace_value=5.0)
        bad += int((ok != dk[s].cpu().numpy()).sum()) + int((om.view(np.uint32) != dm[s].cpu().numpy().view(np.uint32)).sum())
    print("%-24s pose %.3f setup+clip %.3f raster %.3f total %.3f ms (one lane, one batch alone) | per batch, two in flight: one lane %.3f, default lanes %.3f ms | clipped %d binned %d entries %d frags %d regrow %d | mismatches %d" % (
        name, st["ms_pose"], st["ms_setup"], st["ms_raster"], st["ms_total"], one_lane_ms, two_lane_ms, st["triangles_clipped"], st["triangles_binned"], st["bin_entries"], st["fragments_binned"], st["regrowths"], bad))
