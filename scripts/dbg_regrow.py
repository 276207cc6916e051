"""Repeats the three cases of tests/test_parity_gpu.py::test_fragment_bin_and_clip_list_overflow_regrow and prints the
statistics of every batch whose regrowth count differs from the batch before it (debug helper, GPU box)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import scenes as S
import realtime_urdf_filter_amd as R
W, H = 256, 128
P = S.projection(210.0, 210.0, (W - 1) / 2, (H - 1) / 2, W, H)
I = S.gl(np.eye(4))
depth = S.sensor_depth(W, H, 0.9)
rng = np.random.default_rng(99)
n = 60000
centre = np.stack([rng.uniform(-0.02, 0.03, n), rng.uniform(-0.02, 0.01, n), rng.uniform(0.9, 1.1, n)], axis=1)
va = (centre[:, None, :] + rng.normal(scale=0.006, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
n2 = 120000
a = np.stack([rng.uniform(-0.3, 0.3, n2), rng.uniform(-0.2, 0.2, n2), rng.uniform(0.3, 2.0, n2)], axis=1)
b = a + np.stack([rng.normal(scale=0.01, size=n2), rng.normal(scale=0.01, size=n2), -rng.uniform(2.5, 4.0, n2)], axis=1)
c = a + rng.normal(scale=0.004, size=(n2, 3))
vb = np.stack([a, b, c], axis=1).reshape(-1, 3).astype(np.float32)
n3 = 90000
a3 = np.stack([rng.uniform(-0.5, 0.5, n3), rng.uniform(-0.25, 0.25, n3), rng.uniform(0.8, 2.0, n3)], axis=1)
ang = rng.uniform(0, 2 * np.pi, n3)
b3 = a3 + np.stack([0.9 * np.cos(ang), 0.9 * np.sin(ang), rng.normal(scale=0.05, size=n3)], axis=1)
c3 = a3 + rng.normal(scale=0.004, size=(n3, 3))
vc = np.stack([a3, b3, c3], axis=1).reshape(-1, 3).astype(np.float32)
keys = ("regrowths", "max_bin_fill", "max_fbin_fill", "bin_capacity", "bin_entries", "triangles_clipped", "triangles_binned", "fragments_binned")
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for name, verts in (("a", va), ("b", vb), ("c", vc)):
        tris = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
        p = R.default_params(); p.filter_replace_value = 5.0
        ctx = R.Context(W, H, 1, 0, p)
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
        ctx.finalize_models()
        ctx.set_camera(0, P, I, I)
        ctx.set_link_poses(0, m, np.stack([I]))
        prev = None
        for k in range(3):
            masked, mask = ctx.filter_batch(depth[None])
            st = ctx.stats()
            cur = dict(st)
            if prev is not None and cur["regrowths"] != prev["regrowths"]:
                print("iteration", it, "case", name, "batch", k, "\n   before", prev, "\n   after ", cur, flush=True)
            prev = cur
        ctx.close()
print("done")
