"""Randomised differential test of the feature paths around the rasteriser (run on a GPU box):
   python scripts/fuzz_features.py [n_scenes] [first_seed]
Every scene: 1-9 streams (work items with partial stream triples; every sixth scene 32-40, which a context of several raster lanes
splits into launch groups), one to three raster lanes, 1-3 models with per-stream model selection,
links made of primitives (boxes incl. the reference's second box, spheres, cylinders: fans, strips, quads with
scale / translate ops) and of procedural meshes of up to several chunks, 32FC1 or 16UC1 frames, with or
without the mask, fused or two-kernel, one pipeline or several inside the context (small batches then replay captured
hipGraphs; every pipeline is exercised), full planes or the bit-packed mask-only output expanded on the host -- every
result compared with the CPU oracle pixel by pixel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes as S
import realtime_urdf_filter_amd as R
from realtime_urdf_filter_amd import geometry as G
from bench_support import synthetic
from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32
from oracle import bindings as O

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad_total = 0
t0 = time.time()
for sc in range(n_scenes):
    rng = np.random.default_rng(seed0 + sc)
    W = int(rng.choice([160, 320, 332, 640])); H = int(rng.choice([120, 240, 250, 480]))
    f = float(rng.uniform(0.7, 1.4)) * 525.0 * W / 640
    P = S.projection(f, f, (W - 1) / 2 + float(rng.uniform(-8, 8)), (H - 1) / 2 + float(rng.uniform(-8, 8)), W, H)
    n_models = int(rng.integers(1, 4))
    models = []                      # per model: list of links; per link: list of DrawCall
    for mi in range(n_models):
        links = []
        for li in range(int(rng.integers(1, 5))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                draws = G.box_draws(float(rng.uniform(0.05, 0.8)), float(rng.uniform(0.05, 0.8)), float(rng.uniform(0.05, 0.8)))
            elif kind == 1:
                draws = G.sphere_draws(float(rng.uniform(0.03, 0.4)))
            elif kind == 2:
                draws = G.cylinder_draws(float(rng.uniform(0.03, 0.3)), float(rng.uniform(0.05, 0.8)))
            else:
                v, t = synthetic.lumpy_ellipsoid(int(rng.integers(100, 3000)), (float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.05, 0.4))), int(rng.integers(0, 1 << 30)))
                s3 = [float(np.float32(rng.uniform(0.5, 1.5))) for _ in range(3)]
                draws = [G.DrawCall(v, t, G.OP_SCALE, s3)]
            links.append(draws)
        models.append(links)
    n_streams = int(rng.integers(1, 10))
    if rng.integers(0, 6) == 0: n_streams = int(rng.integers(32, 41))      # large enough for a context of several raster lanes to split the batch
    two = bool(rng.integers(0, 2)); u16 = bool(rng.integers(0, 3) == 0) and (W % 4 == 0); want_mask = bool(rng.integers(0, 4) != 0)
    p = R.default_params(); p.filter_replace_value = float(rng.choice([5.0, 0.0, 7.25])); p.depth_distance_threshold = float(rng.choice([0.05, 0.1, 0.0]))
    if two: p.flags |= R.FLAG_TWO_KERNEL
    if rng.integers(0, 5) == 0: p.bin_capacity = 16
    if rng.integers(0, 4) == 0: p.max_inflight_streams = int(rng.integers(1, 4))      # several launch groups per batch
    pipes = int(rng.choice([0, 0, 2, 3]))
    p.pipelines = pipes
    p.raster_lanes = int(rng.choice([0, 0, 1, 2, 3]))    # (0 = the default: three lanes; batches below 32 streams take the lanes in turn)
    bits = (not two) and (W % 4 == 0) and bool(rng.integers(0, 3) == 0)
    ctx = R.Context(W, H, n_streams, 0, p)
    ids = []
    for links in models:
        m = ctx.add_model()
        for draws in links:
            l = ctx.add_link(m)
            for d in draws:
                ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
        ids.append(m)
    ctx.finalize_models()
    depth = np.stack([S.sensor_depth(W, H, 0.41 * s + sc) for s in range(n_streams)])
    per = []
    for s in range(n_streams):
        offinv, camtf = S.random_camera(rng, small=bool(rng.integers(0, 2)))
        ctx.set_camera(s, P, offinv, camtf)
        chosen = [mi for mi in range(n_models) if rng.integers(0, 4) != 0] or [0]
        ctx.set_stream_models(s, [ids[mi] for mi in chosen])
        tfs_all = []
        for mi, links in enumerate(models):
            tfs = S.random_link_poses(rng, len(links), near=bool(rng.integers(0, 3) == 0))
            ctx.set_link_poses(s, ids[mi], np.stack(tfs))
            tfs_all.append(tfs)
        per.append((offinv, camtf, chosen, tfs_all))
    if u16:
        mm = np.clip(np.nan_to_num(depth, nan=0.0, posinf=0.0) * 1000.0 + rng.integers(-3, 4, depth.shape), 0, 65535).astype(np.uint16)
    for rep_ in range((pipes + 1 + sc % pipes) if pipes else 1):   # with pipelines: every one gets a batch, some twice (graph replay); the checked batch lands on a varying one
        if bits:
            src = np.ascontiguousarray(mm if u16 else depth)
            packed = np.zeros((n_streams, ctx.mask_bits_words()), np.uint32)
            ctx.filter_batch_bits_async(src, packed)
            ctx.sync()
            pairs = [R.expand_mask_bits(src[s], packed[s], p.filter_replace_value) for s in range(n_streams)]
            masked, mask = np.stack([a for a, _ in pairs]), np.stack([b for _, b in pairs])
            want_mask = True
        elif u16:
            masked, mask = ctx.filter_batch_u16(mm, want_mask=want_mask)
        else:
            masked, mask = ctx.filter_batch(depth, want_mask=want_mask)
    for s, (offinv, camtf, chosen, tfs_all) in enumerate(per):
        draws = []
        for mi in chosen:
            for li, dl in enumerate(models[mi]):
                for d in dl:
                    draws.append((tfs_all[mi][li], d.pre_op, d.op, d.verts, d.tris))
        din = depth_u16_to_f32(mm[s]) if u16 else depth[s]
        om, ok = O.filter_frame(din, P, draws, offinv, camtf, max_diff=p.depth_distance_threshold, replace_value=p.filter_replace_value)
        bm = int((ok != mask[s]).sum()) if want_mask else 0
        bd = int((depth_f32_to_u16(om) != masked[s]).sum()) if u16 else int((om.view(np.uint32) != masked[s].view(np.uint32)).sum())
        if bm or bd:
            bad_total += 1
            print("MISMATCH scene %d (seed %d) stream %d/%d %dx%d two=%s u16=%s mask=%s pipelines=%d bits=%s models=%s: mask %d depth %d" % (sc, seed0 + sc, s, n_streams, W, H, two, u16, want_mask, pipes, bits, chosen, bm, bd), flush=True)
    ctx.close()
print("scenes %d, streams with mismatches %d, %.1f s" % (n_scenes, bad_total, time.time() - t0))
sys.exit(1 if bad_total else 0)
