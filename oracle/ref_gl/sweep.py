"""Development container only: the oracle against the reference's own shaders on Mesa llvmpipe over many
random scenes of one resolution (one GL context size per process):
    python -m oracle.ref_gl.sweep WIDTH HEIGHT N_SCENES [FIRST_SEED]
Scenes: off-centre intrinsics, 1-5 triangle soups of size classes from sub-pixel dust to screen-filling,
random poses incl. near-plane crossings, random camera, thresholds 0 / 0.05 / 0.2.  Prints every differing frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes as S
from oracle import bindings as O
from oracle.ref_gl import harness as HN

W, H, n_scenes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
assert HN.available(), "needs oracle/_ref and /root/reference"
hn = HN.Harness(W, H)
bad = 0
t0 = time.time()
for sc in range(n_scenes):
    rng = np.random.default_rng(seed0 + sc)
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
    max_diff = float(rng.choice([0.05, 0.2, 0.0]))
    tfs = S.random_link_poses(rng, n_links, near=bool(rng.integers(0, 2)), far=bool(rng.integers(0, 3) == 0))
    offinv, camtf = S.random_camera(rng, small=bool(rng.integers(0, 2)))
    depth = S.sensor_depth(W, H, 0.37 + sc)
    rend = [(tfs[i], [("mesh", geo[i][0], geo[i][1], geo[i][2], geo[i][3])]) for i in range(n_links)]
    g_masked, g_mask = hn.frame(depth, P, rend, offinv, camtf, max_diff=max_diff, replace_value=5.0)
    o_masked, o_mask = O.filter_frame(depth, P, [(tfs[i],) + geo[i] for i in range(n_links)], offinv, camtf, max_diff=max_diff, replace_value=5.0)
    bm = int((g_mask != o_mask).sum()); bd = int((g_masked.view(np.uint32) != o_masked.view(np.uint32)).sum())
    if bm or bd:
        bad += 1
        print("MISMATCH seed %d: mask %d depth %d" % (seed0 + sc, bm, bd), flush=True)
print("%dx%d: %d scenes, %d with mismatches, %.1f s" % (W, H, n_scenes, bad, time.time() - t0))
sys.exit(1 if bad else 0)
