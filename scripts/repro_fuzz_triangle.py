import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, scenes as S
import realtime_urdf_filter_amd as R
from oracle import bindings as O
seed = 100008; sc = 8
rng = np.random.default_rng(seed)
W = int(rng.choice([64, 150, 160, 320, 333, 640, 1000])); H = int(rng.choice([48, 100, 120, 240, 251, 480, 700]))
f = float(rng.uniform(0.6, 1.6)) * 525.0 * W / 640
P = S.projection(f, f * float(rng.uniform(0.9, 1.1)), (W - 1) / 2 + float(rng.uniform(-20, 20)), (H - 1) / 2 + float(rng.uniform(-20, 20)), W, H)
n_links = int(rng.integers(1, 6)); geo = []
for _ in range(n_links):
    cls = rng.integers(0, 4)
    scale = [0.002, 0.02, 0.15, 1.5][cls] * float(rng.uniform(0.5, 2.0))
    nt = int([1500, 600, 200, 20][cls] * rng.uniform(0.3, 1.0)) + 1
    centre = rng.normal(scale=0.4, size=(nt, 1, 3))
    v = (centre + rng.normal(scale=scale, size=(nt, 3, 3))).reshape(-1, 3).astype(np.float32)
    t = np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3)
    pre = int(rng.integers(0, 3))
    op = [float(np.float32(rng.uniform(0.5, 1.5))) for _ in range(3)] if pre == 1 else ([float(np.float32(rng.uniform(-0.2, 0.2))) for _ in range(3)] if pre == 2 else [0.0, 0.0, 0.0])
    geo.append((pre, op, v, t))
n_streams = int(rng.integers(1, 4)); two = bool(rng.integers(0, 2))
md = float(rng.choice([0.05, 0.2, 0.0])); rng.integers(0, 4)
depth = S.sensor_depth(W, H, 0.37 * 0 + sc)
tfs = S.random_link_poses(rng, n_links, near=bool(rng.integers(0, 2)), far=bool(rng.integers(0, 3) == 0))
offinv, camtf = S.random_camera(rng, small=bool(rng.integers(0, 2)))
pre, op, v, t = geo[1]
tri = 14
for sub in ([tri], list(range(len(t)))):
    vv = v; tt = t[sub]
    p = R.default_params(); p.filter_replace_value = 5.0; p.depth_distance_threshold = md
    ctx = R.Context(W, H, 1, 0, p); m = ctx.add_model(); ctx.add_draw(m, ctx.add_link(m), vv, tt, pre, op); ctx.finalize_models()
    ctx.set_camera(0, P, offinv, camtf); ctx.set_link_poses(0, m, np.stack([tfs[1]]))
    masked, mask = ctx.filter_batch(depth[None]); st = ctx.stats()
    om, ok, zwin, prim, _ = O.filter_frame(depth, P, [(tfs[1], pre, op, vv, tt)], offinv, camtf, max_diff=md, replace_value=5.0, want_debug=True)
    print("tris", len(tt), "mask diff", int((ok != mask[0]).sum()), "oracle covered px", int((prim > 0).sum()), {k: st[k] for k in ("triangles_binned", "bin_entries", "triangles_clipped", "max_bin_fill")})
    ctx.close()
p = R.default_params(); p.filter_replace_value = 5.0; p.depth_distance_threshold = md
ctx = R.Context(W, H, 1, 0, p); m = ctx.add_model(); ctx.add_draw(m, ctx.add_link(m), v, t, pre, op); ctx.finalize_models()
ctx.set_camera(0, P, offinv, camtf); ctx.set_link_poses(0, m, np.stack([tfs[1]]))
masked, mask = ctx.filter_batch(depth[None])
om, ok, zwin, prim, _ = O.filter_frame(depth, P, [(tfs[1], pre, op, v, t)], offinv, camtf, max_diff=md, replace_value=5.0, want_debug=True)
bad = ok != mask[0]
print("prims in diff region:", np.unique(prim[bad], return_counts=True), "zwin range there", zwin[bad].min(), zwin[bad].max())
ctx.close()
for i in np.unique(prim[bad]):
    tt = t[[int(i)]]
    ctx = R.Context(W, H, 1, 0, p); m = ctx.add_model(); ctx.add_draw(m, ctx.add_link(m), v, tt, pre, op); ctx.finalize_models()
    ctx.set_camera(0, P, offinv, camtf); ctx.set_link_poses(0, m, np.stack([tfs[1]]))
    masked1, mask1 = ctx.filter_batch(depth[None]); st = ctx.stats()
    om1, ok1, zw1, pr1, _ = O.filter_frame(depth, P, [(tfs[1], pre, op, v, tt)], offinv, camtf, max_diff=md, replace_value=5.0, want_debug=True)
    print("single prim", i, "mask diff", int((ok1 != mask1[0]).sum()), "covered", int((pr1 >= 0).sum()), {k: st[k] for k in ("triangles_binned", "bin_entries", "triangles_clipped")})
    print("  verts", v[tt[0]].tolist())
    ctx.close()
