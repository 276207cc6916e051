#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz by running the REFERENCE's GLSL shaders
(read at run time from /root/reference/include/shaders, never copied) on Mesa llvmpipe through
oracle/ref_gl (development container only).  A fixture is data: the inputs of one frame
(sensor depth, matrices, indexed triangles) and the reference's outputs for it (mask bits +
sha256 of the masked depth, which is fully determined by mask, sensor depth and replace value --
the generator asserts that identity on the llvmpipe output).

Run:  python tests/golden/generate_golden.py        (rewrites every fixture)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes as S  # noqa: E402
from oracle.ref_gl import harness as HN  # noqa: E402
from oracle import bindings as O  # noqa: E402
from realtime_urdf_filter_amd import geometry as G  # noqa: E402
from bench_support import synthetic, workloads  # noqa: E402

NAN_CODE, INF_CODE = 65535, 65534


def quantise_depth(d):
    """float depth -> uint16 code (multiples of 1/1024 m, exactly representable in float32)."""
    q = np.clip(np.rint(np.nan_to_num(d, nan=0.0, posinf=0.0) * 1024.0), 0, 65000).astype(np.uint16)
    q[np.isnan(d)] = NAN_CODE
    q[np.isposinf(d)] = INF_CODE
    return q


def dequantise_depth(q):
    d = (q.astype(np.float32) / np.float32(1024.0)).astype(np.float32)
    d[q == NAN_CODE] = np.nan
    d[q == INF_CODE] = np.inf
    return d


class Case:
    def __init__(self, name, W, H, depth, projection, renderables, offset_inv=None, cam_tf=None, max_diff=0.05, replace=5.0,
                 depth_exact=None):
        self.name, self.W, self.H = name, W, H
        self.depth_q = quantise_depth(depth) if depth_exact is None else None
        self.depth = dequantise_depth(self.depth_q) if depth_exact is None else np.ascontiguousarray(depth_exact, np.float32)
        self.projection = np.asarray(projection, np.float64)
        self.renderables = renderables      # harness format, see Harness.frame
        I = np.eye(4).T.reshape(16)
        self.offset_inv = I.copy() if offset_inv is None else np.asarray(offset_inv, np.float64)
        self.cam_tf = I.copy() if cam_tf is None else np.asarray(cam_tf, np.float64)
        self.max_diff, self.replace = max_diff, replace

    def flat_draws(self):
        """Triangle draw list for the oracle / HIP path: matrix ops inside one renderable accumulate
        only in the Box case (draw 2 = scale after draw 1 without op), which flattens trivially."""
        out = []
        for link_tf, draws in self.renderables:
            seen_op = False
            for dr in draws:
                kind, pre, op = dr[0], dr[1], dr[2]
                assert not (seen_op and pre), "stacked matrix ops inside one renderable are not used by the reference"
                seen_op = seen_op or bool(pre)
                if kind == "mesh":
                    v, t = np.asarray(dr[3], np.float32), np.asarray(dr[4], np.uint32)
                elif kind == "quads":
                    v = np.asarray(dr[3], np.float32)
                    t = np.asarray(G.quads_to_tris(len(v) // 4), np.uint32)
                else:
                    d = G.prims_to_draw(dr[3])
                    v, t = d.verts, d.tris
                out.append((np.asarray(link_tf, np.float64), pre, [float(np.float32(x)) for x in op], v, t))
        return out


def box_renderable(link_tf, dx, dy, dz):
    """RenderableBox::render as GL calls: the VBO as GL_QUADS, then glScalef + glutSolidCube."""
    d = G.box_draws(dx, dy, dz)
    return (link_tf, [("quads", 0, (0, 0, 0), d[0].verts), ("prims", 1, d[1].op, G.cube_prims(np.float32(dx)))])


def cases():
    out = []
    W, H = 160, 120
    P = S.projection(131.25, 131.25, 79.5, 59.5, W, H)
    I = np.eye(4).T.reshape(16)

    # 1. analytic triangles on pixel centres / half-pixel edges, shared edges, slivers
    def win_to_obj(xw, yw, z):
        return [(xw - 79.5) / 131.25 * z, (yw - (H - 59.5)) / 131.25 * z, z]
    tri = []
    for (a, b, c, z) in [((10.5, 10.5), (30.5, 10.5), (10.5, 30.5), 1.0), ((30.5, 10.5), (30.5, 30.5), (10.5, 30.5), 1.5),
                         ((40.0, 10.0), (60.0, 10.0), (50.0, 30.0), 2.0), ((60.0, 10.0), (80.0, 10.0), (70.0, 30.0), 2.0),
                         ((50.0, 30.0), (60.0, 10.0), (70.0, 30.0), 2.0), ((90.25, 20.75), (140.5, 21.0), (90.5, 21.25), 1.2),
                         ((20.5, 50.5), (20.5, 90.5), (21.5, 70.5), 0.8), ((100.5, 60.5), (120.5, 60.5), (110.5, 60.5), 1.0),
                         ((100.5, 70.5), (130.5, 100.5), (100.5, 100.5), 3.0), ((130.5, 70.5), (130.5, 100.5), (100.5, 70.5), 3.0)]:
        tri += [win_to_obj(a[0], a[1], z), win_to_obj(b[0], b[1], z), win_to_obj(c[0], c[1], z)]
    v = np.asarray(tri, np.float32)
    t = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
    depth = np.full((H, W), 4.0, np.float32)
    out.append(Case("analytic_edges_160x120", W, H, depth, P, [(I, [("mesh", 0, (0, 0, 0), v, t)])]))

    # 2. random soups: screen-edge, near-plane and far-plane crossers, scale / translate ops
    for seed, near, far in ((11, False, False), (12, True, False), (13, True, True)):
        rng = np.random.default_rng(seed)
        geo = S.soup_geometry(rng, n_links=6, tris_per_link=30)
        tfs = S.random_link_poses(rng, len(geo), near=near, far=far)
        offinv, camtf = S.random_camera(rng)
        rend = [(tfs[i], [("mesh", geo[i][0], geo[i][1], geo[i][2], geo[i][3])]) for i in range(len(geo))]
        out.append(Case("soup_seed%d_160x120" % seed, W, H, S.sensor_depth(W, H, 0.1 * seed), P, rend, offinv, camtf))

    # 3. example.urdf.xml (quirk Q1 boxes) at two sizes
    for (w, h) in ((160, 120), (640, 480)):
        wl = workloads.example_workload(w, h)
        rend = []
        links = wl.models[0]
        for li, draws in enumerate(links):
            rend.append(box_renderable(wl.link_tf[0][0, li], 4.0, 0.5, 2.0))
        out.append(Case("example_urdf_%dx%d" % (w, h), w, h, synthetic.sensor_depth(w, h, 0), wl.projection[0], rend,
                        wl.offset_inv[0], wl.cam_tf[0]))

    # 4. primitives through their native GL primitive types (fans, quad strips, quads)
    T1 = np.eye(4); T1[:3, :3] = S.rand_rot(np.random.default_rng(5)); T1[:3, 3] = [-0.5, 0.1, 1.6]
    T2 = np.eye(4); T2[:3, :3] = S.rand_rot(np.random.default_rng(6)); T2[:3, 3] = [0.5, -0.1, 2.0]
    T3 = np.eye(4); T3[:3, :3] = S.rand_rot(np.random.default_rng(7)); T3[:3, 3] = [0.0, 0.3, 1.2]
    rend = [(S.gl(T1), [("prims", 0, (0, 0, 0), G.sphere_prims(0.35))]),
            (S.gl(T2), [("prims", 2, G.cylinder_translate(0.9), G.cylinder_prims(0.2, 0.9))]),
            box_renderable(S.gl(T3), 0.3, 0.2, 0.25)]
    out.append(Case("primitives_160x120", W, H, S.sensor_depth(W, H, 0.7), P, rend))

    # 5. procedural mesh links (mesh scale op) under random poses
    for seed in (21, 22):
        rng = np.random.default_rng(seed)
        rend = []
        for k in range(5):
            vv, tt = synthetic.lumpy_ellipsoid(400, (0.12 + 0.05 * k, 0.08, 0.2), seed * 10 + k)
            Tk = np.eye(4); Tk[:3, :3] = S.rand_rot(rng); Tk[:3, 3] = [rng.uniform(-0.8, 0.8), rng.uniform(-0.5, 0.5), rng.uniform(0.3, 3.0)]
            sc = (np.float32(rng.uniform(0.7, 1.4)), np.float32(rng.uniform(0.7, 1.4)), np.float32(rng.uniform(0.7, 1.4)))
            rend.append((S.gl(Tk), [("mesh", 1, sc, vv, tt)]))
        offinv, camtf = S.random_camera(rng, small=False)
        out.append(Case("mesh_links_seed%d_160x120" % seed, W, H, S.sensor_depth(W, H, 0.05 * seed), P, rend, offinv, camtf, max_diff=0.02, replace=0.0))

    # 6. geometry within 0.1-0.2 m of the camera: window z < 0.5, where float z is finer than the 24-bit
    #    depth buffer; near-coplanar layers in both draw orders exercise the GL_LESS tie semantics
    rng = np.random.default_rng(31)
    base = rng.normal(scale=0.05, size=(36, 3)).astype(np.float32)
    tt = np.arange(36, dtype=np.uint32).reshape(-1, 3)
    Ta = np.eye(4); Ta[:3, :3] = S.rand_rot(rng); Ta[:3, 3] = [0.0, 0.0, 0.15]
    Tb = Ta.copy(); Tb[2, 3] += 2e-9
    Tc = Ta.copy(); Tc[2, 3] -= 3e-9
    rend = [(S.gl(Ta), [("mesh", 0, (0, 0, 0), base, tt)]), (S.gl(Tb), [("mesh", 0, (0, 0, 0), base, tt)]),
            (S.gl(Tc), [("mesh", 0, (0, 0, 0), base, tt)]), (S.gl(Ta), [("mesh", 1, (np.float32(1.0000001), np.float32(1), np.float32(1)), base, tt)])]
    d6 = np.full((H, W), 0.2, np.float32)
    out.append(Case("near_range_ties_160x120", W, H, d6, P, rend, max_diff=0.001))

    # 7. nothing but the background quad
    out.append(Case("background_only_160x120", W, H, S.sensor_depth(W, H, 1.3), P, []))

    # 8. (round 5) large triangles closer than twice the near plane, at a frame size whose last tile column and row are partial:
    #    slivers one to three pixels wide and a tile tall, flats a tile wide and two pixels high, boxes of 10-60 pixels a side,
    #    triangles larger than a 64x32 tile, in pairs -- the shapes the HIP tile kernel walks as row-stepped strips
    W8, H8 = 517, 389
    f8 = 525.0 * W8 / 640
    P8 = S.projection(f8, f8, (W8 - 1) / 2, (H8 - 1) / 2, W8, H8)
    rng = np.random.default_rng(2517)
    v8 = []
    def px(cx, cy, z, pts):
        return [[(cx + dx - (W8 - 1) / 2) * (z + dz) / f8, (cy + dy - (H8 - 1) / 2) * (z + dz) / f8, z + dz] for dx, dy, dz in pts]
    for _ in range(130):
        cx, cy = rng.uniform(0, W8), rng.uniform(0, H8)
        for _ in range(2):
            z = rng.uniform(0.103, 0.19)
            kind = rng.integers(0, 5)
            ox, oy = rng.uniform(-20, 20, 2)
            if kind == 0:
                w, h = rng.uniform(0.8, 3.0), rng.uniform(20, 40)
                pts = [(ox, oy, 0.0), (ox + w, oy + rng.uniform(0, 3), rng.uniform(-0.01, 0.01)), (ox + rng.uniform(0, w), oy + h, rng.uniform(-0.02, 0.02))]
            elif kind == 1:
                w, h = rng.uniform(40, 90), rng.uniform(1.2, 3.0)
                pts = [(ox, oy, 0.0), (ox + w, oy + rng.uniform(0, h), rng.uniform(-0.02, 0.02)), (ox + rng.uniform(0, w), oy + h, rng.uniform(-0.01, 0.01))]
            elif kind == 4:
                pts = [(ox - rng.uniform(40, 90), oy - rng.uniform(20, 50), rng.uniform(-0.03, 0.03)), (ox + rng.uniform(40, 90), oy - rng.uniform(-10, 30), rng.uniform(-0.03, 0.03)),
                       (ox + rng.uniform(-30, 30), oy + rng.uniform(30, 70), rng.uniform(-0.03, 0.03))]
            else:
                a, b = rng.uniform(10, 60, 2)
                pts = [(ox, oy, 0.0), (ox + a, oy + rng.uniform(-5, 5), rng.uniform(-0.02, 0.02)), (ox + rng.uniform(-5, 5), oy + b, rng.uniform(-0.02, 0.02))]
            v8 += px(cx, cy, z, pts)
    v8 = np.asarray(v8, np.float32)
    t8 = np.arange(len(v8), dtype=np.uint32).reshape(-1, 3)
    d8 = S.sensor_depth(W8, H8, 0.4)
    d8[::3] = np.float32(0.15)
    out.append(Case("near_large_shapes_517x389", W8, H8, d8, P8, [(S.gl(np.eye(4)), [("mesh", 0, (0, 0, 0), v8, t8)])], max_diff=0.01))
    return out


def threshold_case(hn):
    """Sensor values exactly at / one ulp around `virtual - max_diff` (computed from the llvmpipe run
    itself through the oracle's debug z, which is bit-identical to it)."""
    W, H = 160, 120
    P = S.projection(131.25, 131.25, 79.5, 59.5, W, H)
    rng = np.random.default_rng(41)
    geo = S.soup_geometry(rng, n_links=4, tris_per_link=20)
    tfs = S.random_link_poses(rng, len(geo))
    rend = [(tfs[i], [("mesh", geo[i][0], geo[i][1], geo[i][2], geo[i][3])]) for i in range(len(geo))]
    c0 = Case("tmp", W, H, np.full((H, W), 3.0, np.float32), P, rend)
    _, _, zwin, prim, _ = O.filter_frame(c0.depth, P, c0.flat_draws(), want_debug=True, replace_value=5.0)
    n, f = np.float32(0.1), np.float32(8.0)
    num = np.float32(np.float32(n * f) / np.float32(n - f))
    off = np.float32(f / np.float32(f - n))
    virt = (num / (zwin - off).astype(np.float32)).astype(np.float32)
    thr = (virt - np.float32(0.05)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    sel = (xx + yy) % 3
    d = thr.copy()
    d[sel == 1] = np.nextafter(thr[sel == 1], np.float32(np.inf))
    d[sel == 2] = np.nextafter(thr[sel == 2], np.float32(-np.inf))
    d[0, :8] = [np.nan, 0.0, np.inf, 7.9, 7.8, -1.0, 7.87, 7.870001]
    return Case("threshold_ulps_160x120", W, H, None, P, rend, depth_exact=d)


def main():
    harn = {}
    allc = cases()
    for c in allc + [None]:
        if c is None:
            c = threshold_case(harn.get((160, 120)))
        key = (c.W, c.H)
        # one GL context per process size: the harness keeps a single FBO -> run each size in order
        if key not in harn:
            harn[key] = None
    by_size = {}
    for c in allc + [threshold_case(None)]:
        by_size.setdefault((c.W, c.H), []).append(c)
    import subprocess
    if len(sys.argv) > 1 and sys.argv[1] == "--size":
        W, H = int(sys.argv[2]), int(sys.argv[3])
        hn = HN.Harness(W, H)
        print("renderer:", hn.renderer())
        for c in by_size[(W, H)]:
            masked, mask = hn.frame(c.depth, c.projection, c.renderables, c.offset_inv, c.cam_tf, max_diff=c.max_diff, replace_value=c.replace)
            # the reference's output is a pure function of mask / sensor / replace value
            recon = np.where(mask > 0, np.float32(c.replace), c.depth).astype(np.float32)
            assert np.array_equal(recon.view(np.uint32), masked.view(np.uint32)), c.name
            assert set(np.unique(mask)) <= {0, 255}
            # the CPU oracle must agree with llvmpipe before a fixture is written
            om, ok = O.filter_frame(c.depth, c.projection, c.flat_draws(), c.offset_inv, c.cam_tf, max_diff=c.max_diff, replace_value=c.replace)
            assert np.array_equal(ok, mask), "%s: oracle mask differs from llvmpipe in %d px" % (c.name, int((ok != mask).sum()))
            assert np.array_equal(om.view(np.uint32), masked.view(np.uint32)), c.name
            draws = c.flat_draws()
            fx = {
                "width": c.W, "height": c.H, "max_diff": np.float32(c.max_diff), "replace_value": np.float32(c.replace),
                "projection": c.projection, "offset_inv": c.offset_inv, "cam_tf": c.cam_tf,
                "link_tf": np.stack([d[0] for d in draws]) if draws else np.zeros((0, 16)),
                "pre_op": np.asarray([d[1] for d in draws], np.int32),
                "op": np.asarray([d[2] for d in draws], np.float32).reshape(-1, 3),
                "vert_count": np.asarray([len(d[3]) for d in draws], np.int32),
                "tri_count": np.asarray([len(d[4]) for d in draws], np.int32),
                "verts": np.concatenate([d[3] for d in draws]).astype(np.float32) if draws else np.zeros((0, 3), np.float32),
                "tris": np.concatenate([d[4] for d in draws]).astype(np.uint32) if draws else np.zeros((0, 3), np.uint32),
                "mask_bits": np.packbits(mask > 0),
                "masked_sha256": np.frombuffer(hashlib.sha256(masked.tobytes()).digest(), np.uint8),
                "renderer": np.frombuffer(hn.renderer().encode(), np.uint8),
            }
            if c.depth_q is not None:
                fx["depth_q"] = c.depth_q
            else:
                fx["depth_f32"] = c.depth
            path = os.path.join(HERE, c.name + ".npz")
            np.savez_compressed(path, **fx)
            print("%-34s masked_px=%6d  %6.1f KiB" % (c.name, int((mask > 0).sum()), os.path.getsize(path) / 1024))
        return
    for (W, H) in by_size:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--size", str(W), str(H)])


if __name__ == "__main__":
    main()
