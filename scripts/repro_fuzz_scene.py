import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scenes as S
import realtime_urdf_filter_amd as R
from oracle import bindings as O
"""Re-runs one scene of scripts/fuzz_parity.py with details:  python scripts/repro_fuzz_scene.py SEED FIRST_SEED [bin_capacity]
(SEED as printed by the fuzz run, FIRST_SEED = the fuzz run's second argument)."""
seed = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0; force_cap = int(sys.argv[3]) if len(sys.argv) > 3 else -1
sc = seed - seed0
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
    geo.append((pre, op, v, t)); print("link cls", cls, "nt", nt, "scale", scale)
n_streams = int(rng.integers(1, 4)); two = bool(rng.integers(0, 2))
p = R.default_params(); p.filter_replace_value = 5.0; p.depth_distance_threshold = float(rng.choice([0.05, 0.2, 0.0]))
if two: p.flags |= R.FLAG_TWO_KERNEL
capflag = rng.integers(0, 4) == 0
if capflag: p.bin_capacity = 16
if force_cap >= 0: p.bin_capacity = force_cap
print("W,H", W, H, "streams", n_streams, "two", two, "cap16", capflag, "cap used", p.bin_capacity)
ctx = R.Context(W, H, n_streams, 0, p)
m = ctx.add_model()
for pre, op, v, t in geo: ctx.add_draw(m, ctx.add_link(m), v, t, pre, op)
ctx.finalize_models()
depth = np.stack([S.sensor_depth(W, H, 0.37 * s + sc) for s in range(n_streams)])
per = []
for s in range(n_streams):
    tfs = S.random_link_poses(rng, n_links, near=bool(rng.integers(0, 2)), far=bool(rng.integers(0, 3) == 0))
    offinv, camtf = S.random_camera(rng, small=bool(rng.integers(0, 2)))
    ctx.set_camera(s, P, offinv, camtf); ctx.set_link_poses(s, m, np.stack(tfs)); per.append((tfs, offinv, camtf))
for rep in range(3):
    masked, mask = ctx.filter_batch(depth)
    st = ctx.stats()
    print("rep", rep, {k: st[k] for k in ("triangles_binned", "bin_entries", "triangles_clipped", "max_bin_fill", "bin_capacity", "regrowths", "max_fbin_fill", "fragments_binned")})
    for s, (tfs, offinv, camtf) in enumerate(per):
        om, ok, zwin, prim, _ = O.filter_frame(depth[s], P, [(tfs[i],) + geo[i] for i in range(n_links)], offinv, camtf, max_diff=p.depth_distance_threshold, replace_value=5.0, want_debug=True)
        bad = ok != mask[s]
        print("  stream", s, "mask diff", int(bad.sum()), "depth diff", int((om.view(np.uint32) != masked[s].view(np.uint32)).sum()))
        if bad.sum():
            ys, xs = np.nonzero(bad); print("   bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max(), "prims there", np.unique(prim[bad])[:10], "gpu mask vals", np.unique(mask[s][bad]), "oracle", np.unique(ok[bad]))
ctx.close()
