"""Development container only (skipped elsewhere): the oracle against the REAL reference shaders
running live on Mesa llvmpipe, on fresh seeded scenes that are not among the committed fixtures."""
import numpy as np
import pytest

import scenes as S
from oracle import bindings as O
from oracle.ref_gl import harness as HN

pytestmark = pytest.mark.skipif(not HN.available(), reason="needs oracle/_ref + /root/reference (dev container)")

_hn = {}


def harness(w, h):
    if (w, h) not in _hn:
        if _hn:
            pytest.skip("one GL context size per process")
        _hn[(w, h)] = HN.Harness(w, h)
    return _hn[(w, h)]


@pytest.mark.parametrize("seed", range(100, 106))
def test_random_scene_matches_llvmpipe(seed):
    W, H = 320, 240
    hn = harness(W, H)
    rng = np.random.default_rng(seed)
    P = S.projection(262.5, 262.5, 159.5, 119.5, W, H)
    geo = S.soup_geometry(rng, n_links=7, tris_per_link=50)
    tfs = S.random_link_poses(rng, len(geo), near=bool(seed & 1), far=(seed % 3 == 0))
    offinv, camtf = S.random_camera(rng, small=bool(seed & 2))
    depth = S.sensor_depth(W, H, 0.21 * seed)
    rend = [(tfs[i], [("mesh", geo[i][0], geo[i][1], geo[i][2], geo[i][3])]) for i in range(len(geo))]
    g_masked, g_mask = hn.frame(depth, P, rend, offinv, camtf, replace_value=5.0)
    draws = [(tfs[i],) + geo[i] for i in range(len(geo))]
    o_masked, o_mask = O.filter_frame(depth, P, draws, offinv, camtf, replace_value=5.0)
    assert (g_mask != o_mask).sum() == 0
    assert np.array_equal(g_masked.view(np.uint32), o_masked.view(np.uint32))


def test_standin_shaders_render_exactly_what_the_reference_shaders_render():
    """oracle/ref_gl/standin_shaders/ (repo-authored; what bench.py's llvmpipe timing leg runs on the GPU box, where the
    reference's files do not exist) against the reference's own shader files, in the same GL context on the same
    scenes: all four colour attachments bit-identical -- so timing the stand-in is timing the reference's program."""
    W, H = 320, 240
    hn = harness(W, H)
    frames = {}
    for shaders in ("reference", "standin", "reference"):
        hn.use_shaders(shaders)
        for seed in (100, 103, 105):
            rng = np.random.default_rng(seed)
            P = S.projection(262.5, 262.5, 159.5, 119.5, W, H)
            geo = S.soup_geometry(rng, n_links=7, tris_per_link=50)
            tfs = S.random_link_poses(rng, len(geo), near=bool(seed & 1), far=(seed % 3 == 0))
            offinv, camtf = S.random_camera(rng, small=bool(seed & 2))
            depth = S.sensor_depth(W, H, 0.21 * seed)
            rend = [(tfs[i], [("mesh", geo[i][0], geo[i][1], geo[i][2], geo[i][3])]) for i in range(len(geo))]
            masked, mask = hn.frame(depth, P, rend, offinv, camtf, replace_value=5.0)
            att = [hn.read_attachment(i) for i in range(4)]
            key = seed
            if shaders == "reference":
                frames[key] = (masked.copy(), mask.copy(), att)
            else:
                r_masked, r_mask, r_att = frames[key]
                assert np.array_equal(mask, r_mask) and np.array_equal(masked.view(np.uint32), r_masked.view(np.uint32))
                for i in range(4):
                    assert np.array_equal(att[i].view(np.uint32), r_att[i].view(np.uint32)), "attachment %d, seed %d" % (i, seed)
    assert hn.shaders == "reference"
