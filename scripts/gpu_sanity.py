"""First end-to-end GPU check: random scenes through librtuf.so vs the CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import realtime_urdf_filter_amd as R
from oracle import bindings as O
import scenes as S

W, H = 640, 480
P = S.projection(525., 525., 319.5, 239.5, W, H)
for two in (0, 1):
    for seed in range(4):
        rng = np.random.default_rng(seed)
        geo = S.soup_geometry(rng, n_links=8, tris_per_link=60)
        n = 3
        prm = R.default_params(); prm.filter_replace_value = 5.0
        if two: prm.flags |= R.FLAG_TWO_KERNEL
        ctx = R.Context(W, H, max_streams=n, params=prm)
        m = ctx.add_model()
        for pre, op, v, t in geo:
            l = ctx.add_link(m); ctx.add_draw(m, l, v, t, pre, op)
        ctx.finalize_models()
        depth = np.stack([S.sensor_depth(W, H, phase=0.3 * s + seed) for s in range(n)])
        poses = []; cams = []
        for s in range(n):
            tfs = S.random_link_poses(rng, len(geo), near=(s == 1), far=(s == 2))
            offinv, camtf = S.random_camera(rng)
            poses.append(tfs); cams.append((offinv, camtf))
            ctx.set_camera(s, P, offinv, camtf); ctx.set_link_poses(s, m, np.stack(tfs))
        t0 = time.time(); masked, mask = ctx.filter_batch(depth); t1 = time.time()
        bad_m = bad_d = 0
        for s in range(n):
            draws = [(poses[s][i], geo[i][0], geo[i][1], geo[i][2], geo[i][3]) for i in range(len(geo))]
            om, ok = O.filter_frame(depth[s], P, draws, cams[s][0], cams[s][1], replace_value=5.0)
            bad_m += int((ok != mask[s]).sum()); bad_d += int((om.view(np.uint32) != masked[s].view(np.uint32)).sum())
        print("two_kernel=%d seed=%d mask_mismatch=%d depth_mismatch=%d masked_px=%d  %.1f ms  %s" % (two, seed, bad_m, bad_d, int((mask > 0).sum()), (t1 - t0) * 1e3, ctx.stats()))
        ctx.close()
