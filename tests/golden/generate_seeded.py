#!/usr/bin/env python3
"""Golden fixtures at BASELINE.json's full sizes, addressed by SEED instead of stored geometry.

tests/golden/generate_golden.py stores the inputs of a frame; a 250 k-triangle robot does not fit a small
fixture.  The bench workloads are deterministic functions of their builder arguments
(bench_support/configs.py: seeds derive from global stream and URDF numbers), so a fixture here stores

    recipe      the arguments of bench_support.configs.build + the stream and step looked at   (JSON)
    inputs_sha256   sha256 over the sensor plane, the matrices and every draw's vertices / indices the
                    recipe produced when the fixture was written (a generator that drifts is reported as
                    such, not as a rasteriser mismatch)
    mask_bits + masked_sha256   the REFERENCE's outputs for those inputs: its own GLSL (read at run time from
                    /root/reference/include/shaders) on Mesa llvmpipe through oracle/ref_gl, the geometry
                    drawn from static vertex / index buffers as src/renderable.cpp:424-452 does

and tests/golden_io.py rebuilds the inputs from the recipe.  One stream each of: BASELINE config 2 / 3 (the
250,388-triangle PR2-like robot at 640x480), config 3 with the forearm in front of the lens, config 4 (1280x720,
robot + the two wall URDFs), config 5 (one of its 64 distinct robots).  Development container only; the
fixtures travel to the GPU box, /root/reference does not.

Run:  python tests/golden/generate_seeded.py
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_io  # noqa: E402

#: name -> recipe.  `build` = keyword arguments of bench_support.configs.build; the share is built with ONE stream per
#: rank / robot, which is global stream `stream` of the full job because seeds derive from global numbers.
RECIPES = {
    "seeded_c3_stream0_640x480": {"build": {"workload": "c3", "world": 1, "rank": 0, "streams": 1, "variants": 1}, "stream": 0, "step": 0, "depth_seed": 0},
    "seeded_c3_stream5_step1_640x480": {"build": {"workload": "c3", "world": 1, "rank": 0, "streams": 6, "variants": 2}, "stream": 5, "step": 1, "depth_seed": 12},
    "seeded_c3_near_arm_640x480": {"build": {"workload": "c3", "world": 1, "rank": 0, "streams": 1, "variants": 1, "near_arm": True}, "stream": 0, "step": 0, "depth_seed": 3},
    "seeded_c3_near_arm_stream2_640x480": {"build": {"workload": "c3", "world": 1, "rank": 0, "streams": 3, "variants": 1, "near_arm": True}, "stream": 2, "step": 0, "depth_seed": 5},
    "seeded_c4_stream0_1280x720": {"build": {"workload": "c4", "world": 8, "rank": 0, "streams": 8, "variants": 1}, "stream": 0, "step": 0, "depth_seed": 1},
    "seeded_c4_near_arm_1280x720": {"build": {"workload": "c4", "world": 8, "rank": 3, "streams": 8, "variants": 1, "near_arm": True}, "stream": 0, "step": 0, "depth_seed": 2},
    "seeded_c5_urdf1_640x480": {"build": {"workload": "c5", "world": 64, "rank": 1, "streams": 1, "urdfs": 64, "variants": 1}, "stream": 0, "step": 0, "depth_seed": 4},
    "seeded_c5_urdf4_640x480": {"build": {"workload": "c5", "world": 64, "rank": 4, "streams": 1, "urdfs": 64, "variants": 1}, "stream": 0, "step": 0, "depth_seed": 9},
}


def render_reference(fi):
    """The frame on llvmpipe with the reference's shaders: static buffers per draw, one push / pop per link matrix."""
    import ctypes
    from oracle.ref_gl import harness as HN
    h = HN.Harness(fi.width, fi.height)
    L = h.L
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L.rgo_mesh_clear()
    keep, ids = [], []
    for tf, pre, op, v, t in fi.draws:
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(t, np.uint32).reshape(-1)
        keep += [v, t]
        ids.append(L.rgo_mesh_create(p(v), len(v), 3, p(t), t.size))
    out = np.zeros((fi.height, fi.width), np.float32)
    mask = np.zeros((fi.height, fi.width), np.uint8)
    depth = np.ascontiguousarray(fi.depth, np.float32)
    P, oi, ct = (np.ascontiguousarray(x, np.float64) for x in (fi.projection, fi.offset_inv, fi.cam_tf))
    for _ in range(2):       # (the reference's first frame shades the background quad with unset uniforms: DESIGN.md section 2; take the second)
        L.rgo_begin_frame(p(depth), p(P), p(oi), p(ct), ctypes.c_float(fi.near), ctypes.c_float(fi.far), ctypes.c_float(fi.max_diff), ctypes.c_float(fi.replace_value))
        for mid, (tf, pre, op, v, t) in zip(ids, fi.draws):
            tfc = np.ascontiguousarray(tf, np.float64)
            L.rgo_push_link(p(tfc))
            if pre == 1:
                L.rgo_scale(*[float(x) for x in op])
            elif pre == 2:
                L.rgo_translate(*[float(x) for x in op])
            L.rgo_mesh_draw(mid, HN.GL_TRIANGLES)
            L.rgo_pop_link()
        L.rgo_end_frame(p(out), p(mask))
    return out, mask, h.renderer()


def one(name):
    from oracle import bindings as O
    recipe = RECIPES[name]
    fi = golden_io.seeded_inputs(recipe)
    masked, mask, renderer = render_reference(fi)
    recon = np.where(mask > 0, np.float32(fi.replace_value), fi.depth).astype(np.float32)
    assert np.array_equal(recon.view(np.uint32), masked.view(np.uint32)), name
    assert set(np.unique(mask)) <= {0, 255}
    om, ok = O.filter_frame(fi.depth, fi.projection, fi.draws, fi.offset_inv, fi.cam_tf, max_diff=fi.max_diff, replace_value=fi.replace_value)
    assert np.array_equal(ok, mask), "%s: oracle mask differs from llvmpipe in %d px" % (name, int((ok != mask).sum()))
    assert np.array_equal(om.view(np.uint32), masked.view(np.uint32)), name
    fx = {"recipe": np.frombuffer(json.dumps(recipe, sort_keys=True).encode(), np.uint8),
          "width": fi.width, "height": fi.height, "triangles": fi.triangles,
          "inputs_sha256": np.frombuffer(fi.inputs_sha256, np.uint8),
          "mask_bits": np.packbits(mask > 0),
          "masked_sha256": np.frombuffer(hashlib.sha256(masked.tobytes()).digest(), np.uint8),
          "renderer": np.frombuffer(renderer.encode(), np.uint8)}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print("%-40s %4dx%-4d %7d triangles  masked_px=%7d (%.1f %%)  %6.1f KiB" % (name, fi.width, fi.height, fi.triangles, int((mask > 0).sum()),
                                                                               100.0 * (mask > 0).mean(), os.path.getsize(path) / 1024))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
        return
    # the harness keeps one framebuffer size per process: a process per fixture
    for name in RECIPES:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", name])


if __name__ == "__main__":
    main()
