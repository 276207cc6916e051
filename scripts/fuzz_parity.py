"""Randomised differential test of the HIP path against the CPU oracle (run on a GPU box):
   python scripts/fuzz_parity.py [n_scenes] [first_seed]
Every scene: random resolution, intrinsics, camera, triangle soups of random size classes (sub-pixel dust to
screen-filling), random link poses incl. near-plane crossings, both modes; reports any pixel that differs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scenes as S
import realtime_urdf_filter_amd as R
from oracle import bindings as O

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad_total = 0
t0 = time.time()
for sc in range(n_scenes):
    rng = np.random.default_rng(seed0 + sc)
    if os.environ.get("FUZZ_BIG"):        # the largest frames the ABI accepts: integer ranges at their limits
        W = int(rng.choice([1280, 1920, 2047, 2048])); H = int(rng.choice([1080, 1536, 2048]))
    else:
        W = int(rng.choice([64, 150, 160, 320, 333, 640, 1000]))
        H = int(rng.choice([48, 100, 120, 240, 251, 480, 700]))
    f = float(rng.uniform(0.6, 1.6)) * 525.0 * W / 640
    P = S.projection(f, f * float(rng.uniform(0.9, 1.1)), (W - 1) / 2 + float(rng.uniform(-20, 20)), (H - 1) / 2 + float(rng.uniform(-20, 20)), W, H)
    n_links = int(rng.integers(1, 6))
    geo = []
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
    n_streams = int(rng.integers(1, 4))
    two = bool(rng.integers(0, 2))
    p = R.default_params(); p.filter_replace_value = 5.0; p.depth_distance_threshold = float(rng.choice([0.05, 0.2, 0.0]))
    if two: p.flags |= R.FLAG_TWO_KERNEL
    if rng.integers(0, 4) == 0: p.bin_capacity = 16                     # forces regrowth
    ctx = R.Context(W, H, n_streams, 0, p)
    m = ctx.add_model()
    for pre, op, v, t in geo:
        ctx.add_draw(m, ctx.add_link(m), v, t, pre, op)
    ctx.finalize_models()
    depth = np.stack([S.sensor_depth(W, H, 0.37 * s + sc) for s in range(n_streams)])
    per = []
    for s in range(n_streams):
        tfs = S.random_link_poses(rng, n_links, near=bool(rng.integers(0, 2)), far=bool(rng.integers(0, 3) == 0))
        offinv, camtf = S.random_camera(rng, small=bool(rng.integers(0, 2)))
        ctx.set_camera(s, P, offinv, camtf)
        ctx.set_link_poses(s, m, np.stack(tfs))
        per.append((tfs, offinv, camtf))
    masked, mask = ctx.filter_batch(depth)
    for s, (tfs, offinv, camtf) in enumerate(per):
        om, ok = O.filter_frame(depth[s], P, [(tfs[i],) + geo[i] for i in range(n_links)], offinv, camtf,
                                max_diff=p.depth_distance_threshold, replace_value=5.0)
        bm = int((ok != mask[s]).sum()); bd = int((om.view(np.uint32) != masked[s].view(np.uint32)).sum())
        if bm or bd:
            bad_total += 1
            print("MISMATCH scene %d (seed %d) stream %d %dx%d two=%s: mask %d depth %d" % (sc, seed0 + sc, s, W, H, two, bm, bd), flush=True)
    ctx.close()
print("scenes %d, streams with mismatches %d, %.1f s" % (n_scenes, bad_total, time.time() - t0))
sys.exit(1 if bad_total else 0)
