"""GPU (-m gpu): the HIP path, called through the C ABI, against (1) the committed golden fixtures
(outputs of the reference's GLSL on llvmpipe) and (2) the CPU oracle on seeded inputs.
Bar: bit-exact masks and bit-exact float depth (north_star tolerance for depth is 1e-4; the
implementation is exact, so the tests demand equality)."""
import ctypes

import numpy as np
import pytest

import golden_io
import scenes as S
import realtime_urdf_filter_amd as R
from bench_support import workloads as WL
from realtime_urdf_filter_amd import _capi, urdf
from realtime_urdf_filter_amd.filter import CameraInfo, FilterParameters, RealtimeURDFFilter, depth_f32_to_u16, depth_u16_to_f32
from oracle import bindings as O

pytestmark = pytest.mark.gpu


def params(replace=5.0, max_diff=0.05, two_kernel=False, **kw):
    p = R.default_params()
    p.filter_replace_value = replace
    p.depth_distance_threshold = max_diff
    if two_kernel:
        p.flags |= R.FLAG_TWO_KERNEL
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("two_kernel", [False, True])
@pytest.mark.parametrize("name", golden_io.fixture_names())
def test_golden_fixture(name, two_kernel):
    fx = golden_io.Fixture(name)
    ctx = R.Context(fx.width, fx.height, 1, 0, params(fx.replace_value, fx.max_diff, two_kernel))
    m, tfs = fx.load_into(ctx)
    ctx.set_camera(0, fx.projection, fx.offset_inv, fx.cam_tf)
    if len(tfs):
        ctx.set_link_poses(0, m, tfs)
    masked, mask = ctx.filter_batch(fx.depth[None])
    fx.check(masked[0], mask[0])
    ctx.close()


def run_soups(W, H, n_streams, seed, two_kernel=False, tris_per_link=50, **pkw):
    rng = np.random.default_rng(seed)
    P = S.projection(525.0 * W / 640, 525.0 * W / 640, (W - 1) / 2, (H - 1) / 2, W, H)
    geo = S.soup_geometry(rng, n_links=7, tris_per_link=tris_per_link)
    ctx = R.Context(W, H, n_streams, 0, params(5.0, 0.05, two_kernel, **pkw))
    m = ctx.add_model()
    for pre, op, v, t in geo:
        l = ctx.add_link(m)
        ctx.add_draw(m, l, v, t, pre, op)
    ctx.finalize_models()
    depth = np.stack([S.sensor_depth(W, H, 0.3 * s + seed) for s in range(n_streams)])
    per = []
    for s in range(n_streams):
        tfs = S.random_link_poses(rng, len(geo), near=(s % 3 == 1), far=(s % 3 == 2))
        offinv, camtf = S.random_camera(rng, small=bool(s & 1))
        ctx.set_camera(s, P, offinv, camtf)
        ctx.set_link_poses(s, m, np.stack(tfs))
        per.append((tfs, offinv, camtf))
    return ctx, P, geo, depth, per


def check_vs_oracle(masked, mask, P, geo, depth, per, want_mask=True):
    for s, (tfs, offinv, camtf) in enumerate(per):
        draws = [(tfs[i],) + geo[i] for i in range(len(geo))]
        om, ok = O.filter_frame(depth[s], P, draws, offinv, camtf, replace_value=5.0)
        if want_mask:
            assert (ok != mask[s]).sum() == 0, "stream %d mask differs" % s
        assert bits_equal(om, masked[s]), "stream %d depth differs" % s


@pytest.mark.parametrize("two_kernel", [False, True])
@pytest.mark.parametrize("size", [(640, 480), (150, 100), (1280, 720)])
def test_random_scenes_batch(size, two_kernel):
    ctx, P, geo, depth, per = run_soups(size[0], size[1], 4, seed=size[0] + int(two_kernel), two_kernel=two_kernel)
    masked, mask = ctx.filter_batch(depth)
    check_vs_oracle(masked, mask, P, geo, depth, per)
    st = ctx.stats()
    assert st["triangles_submitted"] == 4 * 350 and st["triangles_clipped"] > 0
    ctx.close()


def test_need_mask_false():
    ctx, P, geo, depth, per = run_soups(320, 240, 2, seed=5)
    masked, mask = ctx.filter_batch(depth, want_mask=False)
    assert mask is None
    check_vs_oracle(masked, None, P, geo, depth, per, want_mask=False)
    ctx.close()


def test_bin_regrowth_and_inflight_groups():
    """Bins of one record + 2 streams per in-flight group: the batch overflows, is re-run with bins sized from what it
    asked for and still comes out exact."""
    ctx, P, geo, depth, per = run_soups(320, 240, 5, seed=9, bin_capacity=1, max_inflight_streams=2)
    masked, mask = ctx.filter_batch(depth)
    check_vs_oracle(masked, mask, P, geo, depth, per)
    st = ctx.stats()
    assert st["regrowths"] >= 1 and st["bin_capacity"] > 1 and st["bin_capacity"] % 256 == 0
    masked2, mask2 = ctx.filter_batch(depth)        # second batch: no further regrowth, same answer
    assert bits_equal(masked, masked2) and np.array_equal(mask, mask2)
    ctx.close()


def test_many_streams_in_one_launch_group():
    """600 small streams: more than the 256 a launch group used to hold, fewer than the 1024 it holds now;
    too-small bins force a regrowth of the large group as well."""
    n = 600
    ctx, P, geo, depth, per = run_soups(96, 64, n, seed=21, bin_capacity=1)
    masked, mask = ctx.filter_batch(depth)
    pick = list(range(0, n, 53)) + [255, 256, 257, n - 1]
    sub = [per[s] for s in pick]
    check_vs_oracle(masked[pick], mask[pick], P, geo, depth[pick], sub)
    assert ctx.stats()["regrowths"] >= 1
    masked2, mask2 = ctx.filter_batch(depth)
    assert bits_equal(masked, masked2) and np.array_equal(mask, mask2)
    ctx.close()


def test_pr2_like_workload_streams():
    wl = WL.pr2_workload(6, 640, 480, total_triangles=20000)
    ctx = R.Context(640, 480, 6, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    depth = wl.depth_batch()
    masked, mask = ctx.filter_batch(depth)
    for s in range(6):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    st = ctx.stats()
    assert st["fragments_binned"] > 0 and st["bin_entries"] > 0
    ctx.close()


def test_stream_model_selection():
    """Two models in one context; streams render different subsets (BASELINE config 5 layout)."""
    W, H = 320, 240
    rng = np.random.default_rng(77)
    P = S.projection(262.5, 262.5, 159.5, 119.5, W, H)
    geos = [S.soup_geometry(rng, 3, 40), S.soup_geometry(rng, 4, 30)]
    ctx = R.Context(W, H, 3, 0, params())
    mids = []
    for geo in geos:
        m = ctx.add_model()
        for pre, op, v, t in geo:
            l = ctx.add_link(m)
            ctx.add_draw(m, l, v, t, pre, op)
        mids.append(m)
    ctx.finalize_models()
    subsets = [[0], [1], [0, 1]]
    depth = np.stack([S.sensor_depth(W, H, s) for s in range(3)])
    tfs = [[S.random_link_poses(rng, len(g)) for g in geos] for _ in range(3)]
    for s in range(3):
        ctx.set_camera(s, P, None, None)
        ctx.set_stream_models(s, [mids[k] for k in subsets[s]])
        for k in range(2):
            ctx.set_link_poses(s, mids[k], np.stack(tfs[s][k]))
    masked, mask = ctx.filter_batch(depth)
    for s in range(3):
        draws = []
        for k in subsets[s]:
            draws += [(tfs[s][k][i],) + geos[k][i] for i in range(len(geos[k]))]
        om, ok = O.filter_frame(depth[s], P, draws, replace_value=5.0)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    ctx.close()


def test_general_projection_draws_background_as_geometry():
    """A projection whose background quad is not a constant full-screen plane (here: sheared z)
    takes the geometry path for the quad; results still match the oracle."""
    W, H = 320, 240
    P = S.projection(262.5, 262.5, 159.5, 119.5, W, H)
    P[2] = 0.02           # clip z picks up a little camera x: the quad's window z now varies
    rng = np.random.default_rng(3)
    geo = S.soup_geometry(rng, 3, 30)
    ctx = R.Context(W, H, 1, 0, params())
    m = ctx.add_model()
    for pre, op, v, t in geo:
        l = ctx.add_link(m)
        ctx.add_draw(m, l, v, t, pre, op)
    ctx.finalize_models()
    tfs = S.random_link_poses(rng, 3)
    ctx.set_camera(0, P, None, None)
    ctx.set_link_poses(0, m, np.stack(tfs))
    depth = S.sensor_depth(W, H, 0.5)[None]
    masked, mask = ctx.filter_batch(depth)
    om, ok = O.filter_frame(depth[0], P, [(tfs[i],) + geo[i] for i in range(3)], replace_value=5.0)
    assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
    ctx.close()


def test_single_stream_filter_api_and_errors():
    fx = golden_io.Fixture("soup_seed11_160x120")
    ctx = R.Context(fx.width, fx.height, 1, 0, params(fx.replace_value, fx.max_diff))
    with pytest.raises(R.RtufError) as e:
        ctx.set_camera(0, fx.projection, None, None)
    assert e.value.code == -6                      # RTUF_ERR_STATE: models not finalised
    m, tfs = fx.load_into(ctx)
    ctx.set_camera(0, None, fx.offset_inv, fx.cam_tf)
    ctx.set_link_poses(0, m, tfs)
    masked, mask = ctx.filter(fx.depth, fx.projection)     # filter(buffer, glTf, w, h) + getMaskedDepth()
    fx.check(masked, mask)
    with pytest.raises(R.RtufError):
        ctx.set_link_poses(0, m, tfs[:-1])
    with pytest.raises(R.RtufError):
        ctx.add_model()                            # already finalised
    with pytest.raises(R.RtufError):
        ctx.set_camera(5, fx.projection, None, None)
    lib = R.load_library()
    assert lib.rtuf_filter(ctx._h, fx.depth.ctypes.data, None, fx.width + 1, fx.height) == -1
    ctx.close()


def test_set_params_between_frames():
    """Threshold, replace value and near plane are plain uniforms that may change between frames
    (include/realtime_urdf_filter/urdf_filter.h:112-135); the far plane is part of the finalized
    background quad and is refused afterwards."""
    fx = golden_io.Fixture("soup_seed11_160x120")
    ctx = R.Context(fx.width, fx.height, 1, 0, params(fx.replace_value, fx.max_diff))
    m, tfs = fx.load_into(ctx)
    ctx.set_camera(0, None, fx.offset_inv, fx.cam_tf)
    ctx.set_link_poses(0, m, tfs)
    masked, mask = ctx.filter(fx.depth, fx.projection)
    fx.check(masked, mask)
    seen = set()
    for thr, rep, near in ((0.3, -1.0, 0.1), (0.001, 7.5, 0.25), (fx.max_diff, fx.replace_value, 0.1)):
        p = params(rep, thr, near_plane=near)
        ctx.set_params(p)
        masked, mask = ctx.filter(fx.depth, fx.projection)
        om, ok = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, z_near=near,
                                max_diff=thr, replace_value=rep)
        assert np.array_equal(mask, ok)
        assert np.array_equal(masked.view(np.uint32), om.view(np.uint32))
        seen.add(int(ok.astype(bool).sum()))
    assert len(seen) == 3                          # the three settings really classify differently
    fx.check(masked, mask)                         # back at the fixture's settings
    with pytest.raises(R.RtufError) as e:
        ctx.set_params(params(fx.replace_value, fx.max_diff, far_plane=6.0))
    assert e.value.code == -6
    masked, mask = ctx.filter(fx.depth, fx.projection)      # the refused call left the context usable
    fx.check(masked, mask)
    ctx.close()


@pytest.mark.parametrize("two_kernel", [False, True])
def test_threshold_division_with_and_without_its_core(two_kernel):
    """The compare threshold divides num = z_near z_far / (z_near - z_far) by z - z_far / (z_far - z_near) per drawn pixel
    (include/shaders/urdf_filter.frag:14-17).  The library runs the division's eight-instruction core where the two constants
    admit it (rtuf_api.cpp: off >= 1 + 2^-10, scripts/fdiv_check.hip) and the full IEEE expansion elsewhere: both sides of the
    rule, and the rule's edge, against the oracle -- which always divides."""
    fx = golden_io.Fixture("soup_seed11_160x120")
    seen = set()
    for near, far in ((0.1, 8.0), (0.3, 3.0), (0.0079, 8.0), (0.0078, 8.0), (0.005, 8.0), (0.02, 500.0), (0.001, 100.0)):
        zn, zf = np.float32(near), np.float32(far)
        off = zf / (zf - zn)
        seen.add(bool(off >= np.float32(1.0) + np.float32(2.0 ** -10)))
        ctx = R.Context(fx.width, fx.height, 1, 0, params(fx.replace_value, fx.max_diff, two_kernel, near_plane=near, far_plane=far))
        m, tfs = fx.load_into(ctx)
        ctx.set_camera(0, None, fx.offset_inv, fx.cam_tf)
        ctx.set_link_poses(0, m, tfs)
        masked, mask = ctx.filter(fx.depth, fx.projection)
        om, ok = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, z_near=near, z_far=far,
                                max_diff=fx.max_diff, replace_value=fx.replace_value)
        assert np.array_equal(mask, ok), (near, far, int((mask != ok).sum()))
        assert np.array_equal(masked.view(np.uint32), om.view(np.uint32)), (near, far)
        ctx.close()
    assert seen == {True, False}


def test_host_mirror_end_to_end_example_urdf():
    """URDF text -> URDFRenderer -> RealtimeURDFFilter.filter() on the GPU == the reference's output for
    urdf/example.urdf.xml (golden fixture of BASELINE config C1)."""
    fx = golden_io.Fixture("example_urdf_640x480")
    tf = urdf.StaticTransformProvider()
    model = urdf.Model.from_string(WL.EXAMPLE_URDF)
    tf.set_frames(urdf.forward_kinematics(model), "/EXAMPLE/")
    tf.frames["/world"] = urdf.Transform()
    tf.frames["/camera_rgb_optical_frame"] = urdf.Transform(np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]), (0, 0, 0))
    prm = FilterParameters("/world", "/camera_rgb_optical_frame",
                           [{"model": "robot_description", "tf_prefix": "/EXAMPLE", "geometry_type": "visual", "scale": 1.0, "ignore": []}],
                           depth_distance_threshold=0.05, filter_replace_value=5.0)
    f = RealtimeURDFFilter(prm, tf, {"robot_description": WL.EXAMPLE_URDF})
    info = CameraInfo(640, 480, [525.0, 0, 319.5, 0, 0, 525.0, 239.5, 0, 0, 0, 1, 0])
    out, mask = f.filter_callback(fx.depth, "32FC1", info)
    fx.check(out, mask)
    assert f.getMaskedDepth() is out
    # 16UC1 round trip (src/urdf_filter.cpp:287-288, :309-312)
    u16 = depth_f32_to_u16(np.nan_to_num(fx.depth, nan=0.0, posinf=0.0))
    out16, mask16 = f.filter_callback(u16, "16UC1", info)
    d32 = depth_u16_to_f32(u16)
    om, ok = O.filter_frame(d32, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, max_diff=0.05, replace_value=5.0)
    assert np.array_equal(mask16, ok) and np.array_equal(out16, depth_f32_to_u16(om))
    assert out16[ok > 0].min() == 5000 and out16[ok > 0].max() == 5000


def test_camera_lookup_failure_keeps_previous_output():
    """Quirk Q6: when the camera transform is unavailable the previous frame's output stays."""
    fx = golden_io.Fixture("example_urdf_160x120")
    tf = urdf.StaticTransformProvider()
    tf.set_frames(urdf.forward_kinematics(urdf.Model.from_string(WL.EXAMPLE_URDF)), "/EXAMPLE/")
    tf.frames["/world"] = urdf.Transform()
    tf.frames["/cam"] = urdf.Transform(np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]), (0, 0, 0))
    prm = FilterParameters("/world", "/cam", [{"model": "d", "tf_prefix": "/EXAMPLE", "geometry_type": "visual"}], 0.05, filter_replace_value=5.0)
    f = RealtimeURDFFilter(prm, tf, {"d": WL.EXAMPLE_URDF})
    f.filter(fx.depth, fx.projection, fx.width, fx.height)
    fx.check(f.masked_depth_, f.mask_)
    prev = f.masked_depth_.copy()
    del tf.frames["/cam"]
    f.filter(np.full_like(fx.depth, 1.0), fx.projection, fx.width, fx.height)
    assert bits_equal(prev, f.masked_depth_)


def test_full_size_batch_properties():
    """BASELINE config C3 scale (256 VGA streams): size-independent properties + oracle spot checks."""
    n = 256
    wl = WL.pr2_workload(n, 640, 480, total_triangles=60000)
    ctx = R.Context(640, 480, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    depth = wl.depth_batch()
    masked, mask = ctx.filter_batch(depth)
    # (a) output is a pure select between the sensor value and the replace value
    recon = np.where(mask > 0, np.float32(wl.replace_value), depth)
    assert bits_equal(recon, masked)
    assert set(np.unique(mask)) <= {0, 255}
    # (b) NaN and zero sensor pixels are never filtered (quirk Q9); +inf always is (quirk Q2)
    assert (mask[np.isnan(depth)] == 0).all() and (mask[depth == 0] == 0).all() and (mask[np.isposinf(depth)] == 255).all()
    # (c) deterministic: a second run is identical
    masked2, mask2 = ctx.filter_batch(depth)
    assert bits_equal(masked, masked2) and np.array_equal(mask, mask2)
    # (d) permutation equivariance: reversing the stream order reverses the outputs
    ctx.set_cameras(0, wl.projection[::-1], wl.offset_inv[::-1], wl.cam_tf[::-1])
    ctx.set_link_poses_batch(0, ids[0], wl.link_tf[0][::-1])
    masked3, mask3 = ctx.filter_batch(depth[::-1])
    assert np.array_equal(mask3[::-1], mask) and bits_equal(masked3[::-1], masked)
    # (e) oracle spot checks
    for s in (0, 101, 255):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    ctx.close()


def test_two_kernel_zsurface_matches_oracle_z():
    fx = golden_io.Fixture("mesh_links_seed21_160x120")
    ctx = R.Context(fx.width, fx.height, 1, 0, params(fx.replace_value, fx.max_diff, two_kernel=True))
    m, tfs = fx.load_into(ctx)
    ctx.set_camera(0, fx.projection, fx.offset_inv, fx.cam_tf)
    ctx.set_link_poses(0, m, tfs)
    ctx.filter_batch(fx.depth[None])
    z = ctx.read_zsurface(1)[0]
    _, _, zwin, prim, _ = O.filter_frame(fx.depth, fx.projection, fx.draws, fx.offset_inv, fx.cam_tf,
                                         max_diff=fx.max_diff, replace_value=fx.replace_value, want_debug=True)
    assert bits_equal(z, zwin)
    ctx.close()


def test_small_triangles_closer_than_twice_the_near_plane():
    """Window z <= 0.5 (eye distance <= 2nf/(n+f) ~ 0.2 m) is where the float z the shader sees is finer
    than the 24-bit depth value.  Small triangles are normally resolved to 8-byte fragments that carry
    no z plane; the ones that may reach that range must keep their plane (records + exact-z pass).
    Thousands of 1-4 pixel triangles between 0.105 m and 0.35 m: z-surface and outputs are bit-exact."""
    W, H, n_tris = 160, 120, 5000
    rng = np.random.default_rng(77)
    P = S.projection(525.0 * W / 640, 525.0 * W / 640, (W - 1) / 2, (H - 1) / 2, W, H)
    zc = rng.uniform(0.105, 0.35, n_tris)
    centre = np.stack([rng.uniform(-0.55, 0.55, n_tris) * zc, rng.uniform(-0.42, 0.42, n_tris) * zc, zc], axis=1)
    size = rng.uniform(0.004, 0.02, n_tris) * zc
    verts = (centre[:, None, :] + rng.normal(size=(n_tris, 3, 3)) * size[:, None, None]).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    depth = S.sensor_depth(W, H, 0.5)
    depth[::2] = np.float32(0.2)                 # sensor values around the rendered depths: both mask outcomes occur
    I = S.gl(np.eye(4))
    draws = [(I, 0, [0.0, 0.0, 0.0], verts, tris)]
    om, ok, zwin, prim, _ = O.filter_frame(depth, P, draws, I, I, replace_value=5.0, want_debug=True)
    near = (zwin <= 0.5) & (prim > 0)
    assert near.sum() > 500 and ((zwin > 0.5) & (prim > 0)).sum() > 500       # the scene straddles z = 0.5
    for two_kernel in (True, False):
        ctx = R.Context(W, H, 1, 0, params(5.0, 0.05, two_kernel))
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
        ctx.finalize_models()
        ctx.set_camera(0, P, I, I)
        ctx.set_link_poses(0, m, np.stack([I]))
        masked, mask = ctx.filter_batch(depth[None])
        assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
        if two_kernel:
            assert bits_equal(ctx.read_zsurface(1)[0], zwin)
        assert ctx.stats()["fragments_binned"] > 1000
        ctx.close()

@pytest.mark.parametrize("wall_at", [None, 0.2060, 0.2030])
def test_geometry_around_the_near_flag_threshold(wall_at):
    """Round 6's tile kernel takes the 24-bit depth of a fragment from the bit pattern of clamp(z) * 16777215 and skips the
    exact-z look wherever a tile holds no NEAR geometry -- a record, fragment or cover whose plane may reach window z 0.51
    somewhere in its box (kNearBit; eye distance 0.2047 m at the default planes).  Triangles of every size class between 0.196
    and 0.214 m (window z 0.49 .. 0.54), tiles with and without near ones side by side, with a wall over the whole frame just
    behind / just in front of the threshold (a cover plane that is / is not near): bit-exact in both modes."""
    W, H = 320, 240
    rng = np.random.default_rng(606 + (0 if wall_at is None else int(wall_at * 1e4)))
    P = S.projection(525.0 * W / 640, 525.0 * W / 640, (W - 1) / 2, (H - 1) / 2, W, H)
    verts, n = [], 0
    for count, lo, hi in ((4000, 0.003, 0.012), (600, 0.02, 0.06), (60, 0.1, 0.3)):      # 1-4 px, up to a tile, several tiles
        zc = rng.uniform(0.196, 0.214, count)
        centre = np.stack([rng.uniform(-0.55, 0.55, count) * zc, rng.uniform(-0.42, 0.42, count) * zc, zc], axis=1)
        size = rng.uniform(lo, hi, count) * zc
        jitter = rng.normal(size=(count, 3, 3)) * size[:, None, None]
        jitter[:, :, 2] *= 0.02                                                             # nearly fronto-parallel: z stays around the threshold
        verts.append((centre[:, None, :] + jitter).reshape(-1, 3))
        n += count
    if wall_at is not None:
        verts.append(np.array([[-3.0, -3.0, wall_at], [3.0, -3.0, wall_at + 0.0004], [0.0, 4.0, wall_at + 0.0002]]))
        n += 1
    verts = np.concatenate(verts).astype(np.float32)
    tris = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    depth = S.sensor_depth(W, H, 0.7)
    depth[::2] = np.float32(0.205)               # sensor values around the rendered depths: both mask outcomes occur
    I = S.gl(np.eye(4))
    draws = [(I, 0, [0.0, 0.0, 0.0], verts, tris)]
    om, ok, zwin, prim, _ = O.filter_frame(depth, P, draws, I, I, replace_value=5.0, want_debug=True)
    drawn = prim > 0
    assert (drawn & (zwin < 0.51)).sum() > 2000 and (drawn & (zwin > 0.51)).sum() > 2000      # the scene straddles the threshold
    for two_kernel in (True, False):
        ctx = R.Context(W, H, 1, 0, params(5.0, 0.05, two_kernel))
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
        ctx.finalize_models()
        ctx.set_camera(0, P, I, I)
        ctx.set_link_poses(0, m, np.stack([I]))
        for _ in range(2):                       # (second batch: the cover pass has switched itself on where the wall covers tiles)
            masked, mask = ctx.filter_batch(depth[None])
            assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
        if two_kernel:
            assert bits_equal(ctx.read_zsurface(1)[0], zwin)
        ctx.close()


@pytest.mark.parametrize("size,per_cluster", [((517, 389), 1), ((517, 389), 2), ((640, 480), 6), ((322, 242), 12)])
def test_large_near_triangles_of_every_shape_walked_as_strips(size, per_cluster):
    """Round 5's strip walk (tile kernel: records of more than 96 pixels in tiles with near geometry are cut into strips of 16
    columns whose lanes step down the rows) against the oracle on the shapes that decide its geometry: slivers one to three
    pixels wide and a tile tall, flats a tile wide and two pixels high, boxes of 10-60 pixels a side, triangles larger than a
    tile -- all closer than twice the near plane (window z <= 0.5: the tiles' keys carry the float's low bits, the path the
    strips are compiled into), one, two, six or twelve to a neighbourhood (a wave's round deals one strip to all 64 lanes, two to
    32 each, four to 16 each), at frame sizes whose last tile column and row are partial."""
    W, H = size
    rng = np.random.default_rng(1000 * per_cluster + W)
    f = 525.0 * W / 640
    P = S.projection(f, f, (W - 1) / 2, (H - 1) / 2, W, H)
    verts, n_tri = [], 0
    def tri_px(cx, cy, z, pts):               # a triangle from pixel offsets around (cx, cy) at eye depth z (camera looks along +z)
        out = []
        for (dx, dy, dz) in pts:
            zz = z + dz
            out.append([(cx + dx - (W - 1) / 2) * zz / f, (cy + dy - (H - 1) / 2) * zz / f, zz])
        return out
    shapes = []
    for _ in range(260 // per_cluster):
        cx, cy = rng.uniform(0, W), rng.uniform(0, H)
        for _ in range(per_cluster):
            z = rng.uniform(0.103, 0.19)
            kind = rng.integers(0, 5)
            ox, oy = rng.uniform(-20, 20, 2)
            if kind == 0:      # vertical sliver
                w, h = rng.uniform(0.8, 3.0), rng.uniform(20, 40)
                pts = [(ox, oy, 0.0), (ox + w, oy + rng.uniform(0, 3), rng.uniform(-0.01, 0.01)), (ox + rng.uniform(0, w), oy + h, rng.uniform(-0.02, 0.02))]
            elif kind == 1:    # horizontal flat
                w, h = rng.uniform(40, 90), rng.uniform(1.2, 3.0)
                pts = [(ox, oy, 0.0), (ox + w, oy + rng.uniform(0, h), rng.uniform(-0.02, 0.02)), (ox + rng.uniform(0, w), oy + h, rng.uniform(-0.01, 0.01))]
            elif kind == 4:    # larger than a tile
                pts = [(ox - rng.uniform(40, 90), oy - rng.uniform(20, 50), rng.uniform(-0.03, 0.03)), (ox + rng.uniform(40, 90), oy - rng.uniform(-10, 30), rng.uniform(-0.03, 0.03)), (ox + rng.uniform(-30, 30), oy + rng.uniform(30, 70), rng.uniform(-0.03, 0.03))]
            else:              # boxes of 10-60 pixels a side
                a, b = rng.uniform(10, 60, 2)
                pts = [(ox, oy, 0.0), (ox + a, oy + rng.uniform(-5, 5), rng.uniform(-0.02, 0.02)), (ox + rng.uniform(-5, 5), oy + b, rng.uniform(-0.02, 0.02))]
            verts += tri_px(cx, cy, z, pts)
            n_tri += 1
    verts = np.asarray(verts, np.float32)
    tris = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    depth = S.sensor_depth(W, H, 0.4)
    depth[::3] = np.float32(0.15)             # sensor values among the rendered depths: both mask outcomes occur
    I = S.gl(np.eye(4))
    om, ok, zwin, prim, _ = O.filter_frame(depth, P, [(I, 0, [0.0, 0.0, 0.0], verts, tris)], I, I, replace_value=5.0, want_debug=True)
    assert ((zwin <= 0.5) & (prim > 0)).mean() > 0.3                       # the frame is mostly near geometry
    for two_kernel in (False, True):
        ctx = R.Context(W, H, 2, 0, params(5.0, 0.05, two_kernel))
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
        ctx.finalize_models()
        for s in range(2):
            ctx.set_camera(s, P, I, I)
            ctx.set_link_poses(s, m, np.stack([I]))
        masked, mask = ctx.filter_batch(np.stack([depth, depth]))
        for s in range(2):
            assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s]), (size, per_cluster, two_kernel, s, int((ok != mask[s]).sum()))
        if two_kernel:
            assert bits_equal(ctx.read_zsurface(2)[1], zwin)
        ctx.close()



def test_cpp_facade_example_matches_reference(tmp_path):
    """examples/example_filter.cpp: the reference's single-camera C++ usage on the facade classes
    (RealtimeURDFFilter::getProjectionMatrix / filter / getMaskedDepth / mask_) reproduces the
    reference's output for urdf/example.urdf.xml."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    if not os.path.exists(exe):
        subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    fx = golden_io.Fixture("example_urdf_640x480")
    (tmp_path / "m.urdf").write_text(WL.EXAMPLE_URDF)
    fx.depth.tofile(tmp_path / "d.f32")
    subprocess.check_call([exe, str(tmp_path / "m.urdf"), str(tmp_path / "d.f32"), "640", "480", "525", "525", "319.5", "239.5", "5.0",
                           str(tmp_path / "o.f32"), str(tmp_path / "o.u8")])
    masked = np.fromfile(tmp_path / "o.f32", np.float32).reshape(480, 640)
    mask = np.fromfile(tmp_path / "o.u8", np.uint8).reshape(480, 640)
    fx.check(masked, mask)


def test_cpp_facade_filter_into_16uc1_and_mask_only(tmp_path):
    """RealtimeURDFFilter::filter_into (what the ROS adapter's callback calls): 16UC1 in, masked 16UC1 + byte mask written into
    the caller's planes, and mask-only through the bit-packed path -- against the oracle on the frame the millimetre values
    stand for, with the reference's convertTo roundings on the way in and out (src/urdf_filter.cpp:287-288, :309-312)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    if not os.path.exists(exe):
        subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    fx = golden_io.Fixture("example_urdf_640x480")
    mm = depth_f32_to_u16(np.nan_to_num(fx.depth, nan=0.0, posinf=0.0))
    (tmp_path / "m.urdf").write_text(WL.EXAMPLE_URDF)
    mm.tofile(tmp_path / "d.u16")
    r = subprocess.run([exe, str(tmp_path / "m.urdf"), str(tmp_path / "d.u16"), "640", "480", "525", "525", "319.5", "239.5", "5.0",
                        str(tmp_path / "o.u16"), str(tmp_path / "o.u8"), "into16"], capture_output=True, text=True)
    assert r.returncode == 0 and "equals" in r.stdout, (r.stdout, r.stderr)
    masked16 = np.fromfile(tmp_path / "o.u16", np.uint16).reshape(480, 640)
    mask = np.fromfile(tmp_path / "o.u8", np.uint8).reshape(480, 640)
    om, ok = O.filter_frame(depth_u16_to_f32(mm), fx.projection, fx.draws, fx.offset_inv, fx.cam_tf, max_diff=0.05, replace_value=5.0)
    assert np.array_equal(mask, ok) and np.array_equal(masked16, depth_f32_to_u16(om))
    assert mask.any() and masked16[ok > 0].min() == 5000 and masked16[ok > 0].max() == 5000


def test_on_device_forward_kinematics():
    """Joint positions in, link matrices + head-camera transform computed on the GPU: the matrices agree
    with the host-side forward kinematics to 1e-12 and the filter output is what the oracle computes
    from exactly those matrices."""
    n = 5
    wl = WL.pr2_workload(n, 320, 240, total_triangles=8000)
    ctx = R.Context(320, 240, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.stage(ctx, ids)
    host_masked, host_mask = ctx.filter_batch(depth)
    wl.load_kinematics(ctx, ids)
    wl.stage_joint_positions(ctx, ids)
    dev_masked, dev_mask = ctx.filter_batch(depth)
    L = wl.link_tf[0].shape[1]
    tf, cam = ctx.read_poses(n, L)
    assert np.abs(tf - wl.link_tf[0]).max() < 1e-12 and np.abs(cam - wl.cam_tf).max() < 1e-12
    for s in range(n):
        draws = []
        for li, dl in enumerate(wl.models[0]):
            for d in dl:
                draws.append((tf[s, li], d.pre_op, d.op, d.verts, d.tris))
        om, ok = O.filter_frame(depth[s], wl.projection[s], draws, wl.offset_inv[s], cam[s], max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != dev_mask[s]).sum() == 0 and bits_equal(om, dev_masked[s])
    # the two pose sources differ by at most rounding noise in the 16th digit: pixel flips are (almost) impossible
    assert (host_mask != dev_mask).sum() <= 2
    # handing explicit matrices again switches the model back to host poses
    wl.stage(ctx, ids)
    again_masked, again_mask = ctx.filter_batch(depth)
    assert np.array_equal(again_mask, host_mask) and bits_equal(again_masked, host_masked)
    ctx.close()


def test_staging_next_joint_positions_while_a_batch_is_in_flight():
    """rtuf_set_joint_positions is staged per batch: the next frame's joint states may be staged between
    rtuf_filter_batch_device and rtuf_sync without disturbing the batch in flight, including partial
    (per-stream) updates; setters of state that is not staged per batch wait for the batch instead."""
    import torch
    n, W, H = 4, 320, 240
    A = WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=1000)
    B = WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=2000)
    ctx = R.Context(W, H, n, 0, params(A.replace_value, A.max_diff))
    ids = A.load_into(ctx)
    A.load_kinematics(ctx, ids)
    depth = A.depth_batch()
    A.stage_joint_positions(ctx, ids)
    ref_a = ctx.filter_batch(depth)
    B.stage_joint_positions(ctx, ids, first_call=False)
    ref_b = ctx.filter_batch(depth)
    assert (ref_a[1] != ref_b[1]).sum() > 0            # the two joint states really differ on screen

    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def run_and(stage_next):
        ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
        stage_next()                                    # host-side staging while the GPU works
        ctx.sync()
        return d_masked.cpu().numpy(), d_mask.cpu().numpy()

    A.stage_joint_positions(ctx, ids, first_call=False)
    got = run_and(lambda: B.stage_joint_positions(ctx, ids, first_call=False))
    assert np.array_equal(got[1], ref_a[1]) and bits_equal(got[0], ref_a[0])
    # B is now staged; while it runs, streams 0..1 go back to A's state (partial update)
    got = run_and(lambda: ctx.set_joint_positions(0, ids[0], A.joint_q[:2], None, A.camera_frame_index))
    assert np.array_equal(got[1], ref_b[1]) and bits_equal(got[0], ref_b[0])
    got = run_and(lambda: None)
    want_mask = np.concatenate([ref_a[1][:2], ref_b[1][2:]])
    want_masked = np.concatenate([ref_a[0][:2], ref_b[0][2:]])
    assert np.array_equal(got[1], want_mask) and bits_equal(got[0], want_masked)
    # a setter whose state is not staged per batch (here: the parameters) waits for the batch in flight: the batch keeps its inputs
    ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
    ctx.set_params(params(A.replace_value, A.max_diff))
    assert np.array_equal(d_mask.cpu().numpy(), want_mask)
    ctx.sync()
    ctx.close()


def test_two_batches_in_flight_and_regrowth_reruns_both():
    """Two device batches may be enqueued before the first is retired; a third call retires the oldest.
    With bins that are too small for the scene, the overflow is only seen when the first batch is
    retired: both batches in flight are run again after the regrowth and still produce the reference
    result from their own joint positions and buffers."""
    import torch
    n, W, H = 4, 320, 240
    A = WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=1000)
    B = WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=2000)
    depth = A.depth_batch()
    refs = []
    ctx = R.Context(W, H, n, 0, params(A.replace_value, A.max_diff))
    ids = A.load_into(ctx)
    A.load_kinematics(ctx, ids)
    for wl in (A, B):
        wl.stage_joint_positions(ctx, ids, first_call=wl is A)
        refs.append(ctx.filter_batch(depth))
    ctx.close()

    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depth).to(dev)
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(3)]
    torch.cuda.synchronize()
    for cap in (0, 1):                       # default capacity, then bins that must grow
        ctx = R.Context(W, H, n, 0, params(A.replace_value, A.max_diff, bin_capacity=cap))
        ids = A.load_into(ctx)
        A.load_kinematics(ctx, ids)
        order = [A, B, A]
        for i, wl in enumerate(order):       # three enqueues, no sync in between: the third retires the first
            wl.stage_joint_positions(ctx, ids, first_call=(i == 0))
            ctx.filter_batch_device(n, d_depth.data_ptr(), outs[i][0].data_ptr(), outs[i][1].data_ptr())
        ctx.sync()
        for i, wl in enumerate(order):
            want = refs[0] if wl is A else refs[1]
            assert np.array_equal(outs[i][1].cpu().numpy(), want[1]) and bits_equal(outs[i][0].cpu().numpy(), want[0]), (cap, i)
        st = ctx.stats()
        assert (st["regrowths"] > 0) == (cap == 1)
        ctx.close()


@pytest.mark.parametrize("cap", [0, 1])
def test_asynchronous_host_planes(cap):
    """rtuf_filter_batch_async / _u16_async: host planes go up and come back on copy streams while another
    batch computes.  Three batches with different joint states and sensor images are enqueued without a wait
    in between (pinned memory, then pageable memory with planes that are not adjacent); every output must
    be the oracle's, also when too-small bins force both batches in flight to run again (cap=1)."""
    n, W, H = 4, 320, 240
    wls = [WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=seed) for seed in (1000, 2000)]
    A = wls[0]
    depths = [A.depth_batch(first=0), A.depth_batch(first=11), A.depth_batch(first=23)]
    order = [wls[0], wls[1], wls[0]]

    def oracle(wl, depth, s):
        return O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                              max_diff=wl.max_diff, replace_value=wl.replace_value)

    ctx = R.Context(W, H, n, 0, params(A.replace_value, A.max_diff, bin_capacity=cap))
    ids = A.load_into(ctx)
    A.load_kinematics(ctx, ids)
    pin_in = [ctx.host_alloc((n, H, W), np.float32) for _ in range(3)]
    pin_out = [ctx.host_alloc((n, H, W), np.float32) for _ in range(3)]
    pin_mask = [ctx.host_alloc((n, H, W), np.uint8) for _ in range(3)]
    for i in range(3):
        pin_in[i][...] = depths[i]
        pin_out[i][...] = -1.0
        wl = order[i]
        wl.stage_joint_positions(ctx, ids, first_call=(i == 0))
        ctx.filter_batch_async(pin_in[i], pin_out[i], pin_mask[i])     # the third call retires the first
    ctx.wait_oldest()
    ctx.wait_oldest()
    ctx.wait_oldest()                                                   # nothing pending: no-op
    for i in range(3):
        for s in range(n):
            om, ok = oracle(order[i], depths[i], s)
            assert np.array_equal(ok, pin_mask[i][s]) and bits_equal(om, pin_out[i][s]), (cap, i, s)
    assert (ctx.stats()["regrowths"] > 0) == (cap == 1)

    # pageable planes scattered in memory, no mask for stream 1, through the raw C ABI
    lib = R.load_library()
    planes_in = [depths[1][s].copy() for s in range(n)]
    planes_out = [np.full((H, W), -1.0, np.float32) for _ in range(n)]
    planes_mask = [np.full((H, W), 7, np.uint8) for _ in range(n)]
    PP = ctypes.c_void_p * n
    wls[1].stage_joint_positions(ctx, ids, first_call=False)
    rc = lib.rtuf_filter_batch_async(ctx._h, n, PP(*[a.ctypes.data for a in planes_in]), PP(*[a.ctypes.data for a in planes_out]),
                                     PP(*[(None if s == 1 else planes_mask[s].ctypes.data) for s in range(n)]))
    assert rc == 0
    ctx.sync()
    for s in range(n):
        om, ok = oracle(wls[1], depths[1], s)
        assert bits_equal(om, planes_out[s])
        assert np.array_equal(planes_mask[s], ok) if s != 1 else (planes_mask[s] == 7).all()

    # a device-plane batch and a host-plane batch in flight together, in both orders
    import torch
    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depths[2]).to(dev)
    d_out = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for host_first in (True, False):
        pin_out[0][...] = -1.0
        d_out.fill_(-1.0)
        torch.cuda.synchronize()
        for kind in (("host", "device") if host_first else ("device", "host")):
            wl = wls[0] if kind == "host" else wls[1]
            wl.stage_joint_positions(ctx, ids, first_call=False)
            if kind == "host":
                ctx.filter_batch_async(pin_in[0], pin_out[0], pin_mask[0])
            else:
                ctx.filter_batch_device(n, d_depth.data_ptr(), d_out.data_ptr(), d_mask.data_ptr())
        ctx.sync()
        for s in range(n):
            om, ok = oracle(wls[0], depths[0], s)
            assert np.array_equal(ok, pin_mask[0][s]) and bits_equal(om, pin_out[0][s]), (host_first, s)
            om, ok = oracle(wls[1], depths[2], s)
            assert np.array_equal(ok, d_mask[s].cpu().numpy()) and bits_equal(om, d_out[s].cpu().numpy()), (host_first, s)

    # 16UC1 planes
    mm = depth_f32_to_u16(np.nan_to_num(depths[2], nan=0.0, posinf=0.0))
    u_in, u_out = ctx.host_alloc((n, H, W), np.uint16), ctx.host_alloc((n, H, W), np.uint16)
    u_in[...] = mm
    wls[0].stage_joint_positions(ctx, ids, first_call=False)
    ctx.filter_batch_async(u_in, u_out, pin_mask[0])
    ctx.sync()
    for s in range(n):
        om, ok = oracle(wls[0], depth_u16_to_f32(mm), s)
        assert np.array_equal(ok, pin_mask[0][s]) and np.array_equal(u_out[s], depth_f32_to_u16(om))
    with pytest.raises(R.RtufError):
        ctx._check(lib.rtuf_host_free(ctx._h, ctypes.c_void_p(planes_in[0].ctypes.data)))      # not a pinned block
    for a in pin_in + pin_out + pin_mask[1:] + [u_in]:
        ctx.host_free(a)
    ctx.close()                                                          # releases the blocks still held


def test_two_contexts_on_one_gpu_interleaved():
    """Two contexts (own HIP stream and bins each) on the same GPU, batches alternating between them with
    nothing retired in between (bench.py --pipelines 2): kernels of the two overlap on the device and
    both produce the reference result."""
    import torch
    n, W, H = 4, 320, 240
    wls = [WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=seed) for seed in (1000, 2000)]
    depth = wls[0].depth_batch()
    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depth).to(dev)
    ctxs, idss, refs = [], [], []
    for wl in wls:
        c = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff))
        ids = wl.load_into(c)
        wl.load_kinematics(c, ids)
        wl.stage_joint_positions(c, ids)
        refs.append(c.filter_batch(depth))
        ctxs.append(c); idss.append(ids)
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(4)]
    torch.cuda.synchronize()
    for k in range(4):                        # A B A B, all in flight together
        ctxs[k % 2].filter_batch_device(n, d_depth.data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    for c in ctxs:
        c.sync()
    for k in range(4):
        assert np.array_equal(outs[k][1].cpu().numpy(), refs[k % 2][1]) and bits_equal(outs[k][0].cpu().numpy(), refs[k % 2][0]), k
    for c in ctxs:
        c.close()


def test_work_list_longer_than_the_estimated_setup_grid():
    """The set-up grid of a batch is sized from the previous batch's work-list length.  Robot out of view
    (short list), then in view (a list many times longer than the estimate): the batch is detected as
    under-covered when its counters are read back, run again, and matches the oracle."""
    n, W, H = 9, 320, 240
    wl = WL.pr2_workload(n, W, H, total_triangles=20000)
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.stage(ctx, ids)
    away = wl.cam_tf.copy().reshape(n, 4, 4)
    away[1:, 3, :3] += np.array([0.0, 0.0, 50.0])         # GL column-major: translation in the last row; 50 m off
    ctx.set_cameras(0, wl.projection, wl.offset_inv, away.reshape(n, 16))
    ctx.filter_batch(depth)                                # only stream 0 sees the robot: one work item per visible chunk
    few = ctx.stats()["triangles_binned"]
    ctx.set_cameras(0, wl.projection, wl.offset_inv, wl.cam_tf)
    before = ctx.stats()["regrowths"]
    masked, mask = ctx.filter_batch(depth)                 # all nine do: three items per chunk
    st = ctx.stats()
    assert st["regrowths"] == before + 1 and st["triangles_binned"] > 4 * max(few, 1)
    for s in range(n):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    ctx.close()


def test_largest_frame_and_model_without_triangles():
    """Edge sizes: the largest frame the ABI accepts (2048 x 2048, 20-bit snapped coordinates at their limit)
    with a triangle soup, and a context whose model has no geometry at all (every pixel sees only the
    background quad: everything with a finite positive sensor value closer than 0.99 * far - threshold
    survives, the rest follows the quirks Q2 / Q9)."""
    W = H = 2048
    rng = np.random.default_rng(5)
    P = S.projection(525.0 * W / 640, 525.0 * W / 640, (W - 1) / 2, (H - 1) / 2, W, H)
    geo = S.soup_geometry(rng, n_links=3, tris_per_link=40)
    tfs = S.random_link_poses(rng, len(geo), near=True)
    offinv, camtf = S.random_camera(rng, small=True)
    depth = S.sensor_depth(W, H, 0.7)
    ctx = R.Context(W, H, 1, 0, params(5.0, 0.05))
    m = ctx.add_model()
    for pre, op, v, t in geo:
        ctx.add_draw(m, ctx.add_link(m), v, t, pre, op)
    ctx.finalize_models()
    ctx.set_camera(0, P, offinv, camtf)
    ctx.set_link_poses(0, m, np.stack(tfs))
    masked, mask = ctx.filter_batch(depth[None])
    om, ok = O.filter_frame(depth, P, [(tfs[i],) + geo[i] for i in range(len(geo))], offinv, camtf, replace_value=5.0)
    assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
    ctx.close()
    with pytest.raises(Exception):
        R.Context(2049, 16, 1, 0, params())                      # beyond the coordinate range: refused, not clamped

    W, H = 64, 48
    depth = S.sensor_depth(W, H, 0.1)
    P = S.projection(52.5, 52.5, 31.5, 23.5, W, H)
    ctx = R.Context(W, H, 2, 0, params(5.0, 0.05))
    m = ctx.add_model()
    ctx.add_link(m)
    ctx.finalize_models()
    I = S.gl(np.eye(4))
    for s in range(2):
        ctx.set_camera(s, P, I, I)
    masked, mask = ctx.filter_batch(np.stack([depth, depth]))
    om, ok = O.filter_frame(depth, P, [], I, I, replace_value=5.0)
    for s in range(2):
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    assert ctx.stats()["triangles_binned"] == 0
    ctx.close()


def test_clipper_vertex_just_outside_the_frustum():
    """Found by scripts/fuzz_parity.py (scene seed 100008): a screen-filling triangle that crosses four frustum
    planes.  The clipper's interpolation is rounded, so one of its vertices lands at window x = -0.0022, which
    snaps to -129/256 px -- one unit below what an in-frustum vertex can reach.  The packed bin records used to
    assume >= -128 and lost the two sub-triangles that use the vertex (83,882 wrong pixels)."""
    W, H = 1000, 700
    P = [-1.8051159391028662, 0.0, 0.0, 0.0, 0.0, 2.507749909440404, 0.0, 0.0, 0.01473850902399576, 0.0036142699082903906, -1.0253164556962024, -1.0, 0.0, 0.0, -0.20253164556962025, 0.0]
    tf = [-0.612386267094772, 0.7155033584275173, -0.3362112489978207, 0.0, -0.32698724997434814, 0.1579548107057661, 0.9317347348516738, 0.0, 0.7197655161425449, 0.6805183479185917, 0.13723111862185855, 0.0, -1.3332428308938, 0.1446924454905245, 1.8375672740901197, 1.0]
    offinv = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.04031177960767986, 0.048556481330739414, -0.02706193609240065, 1.0]
    camtf = [0.9993483173213057, 0.0, -0.03609626943431724, 0.0, 0.0, 1.0, 0.0, 0.0, 0.03609626943431724, 0.0, 0.9993483173213057, 0.0, -0.12102787878760611, -0.10111147367741508, 0.06291488458101735, 1.0]
    v = np.array([[-1.829869031906128, -6.280371189117432, -2.3235292434692383], [-4.480249404907227, 1.0303295850753784, 7.6488237380981445],
                  [4.771854877471924, 1.2539631128311157, -1.0451332330703735]], np.float32)
    t = np.array([[0, 1, 2]], np.uint32)
    op = [0.09092016518115997, -0.024322213605046272, 0.14942046999931335]
    depth = S.sensor_depth(W, H, 8.0)
    om, ok, zwin, prim, dbg = O.filter_frame(depth, P, [(tf, 2, op, v, t)], offinv, camtf, max_diff=0.05, replace_value=5.0, want_debug=True)
    assert (prim >= 0).sum() > 300000                      # the triangle covers half the frame
    for two_kernel in (False, True):
        ctx = R.Context(W, H, 1, 0, params(5.0, 0.05, two_kernel))
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), v, t, 2, op)
        ctx.finalize_models()
        ctx.set_camera(0, P, offinv, camtf)
        ctx.set_link_poses(0, m, np.stack([np.array(tf)]))
        masked, mask = ctx.filter_batch(depth[None])
        assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
        if two_kernel:
            assert bits_equal(ctx.read_zsurface(1)[0], zwin)
        ctx.close()


def test_fragment_bin_and_clip_list_overflow_regrow():
    """Capacity stress: (a) 60,000 two-pixel triangles piled onto one 64x32 tile overflow that tile's fragment bin (and
    its record bin), (b) 120,000 triangles that all cross the near plane overflow the clip list, (c) 90,000 slivers that
    each touch more than four tiles overflow the many-tile lists, (d) 30,000 small triangles of one size class on one
    tile overflow its record bin from the front; all are detected from the batch's counters, the buffers grow to what the
    batch asked for, the batch runs again and matches the oracle."""
    W, H = 256, 128
    P = S.projection(210.0, 210.0, (W - 1) / 2, (H - 1) / 2, W, H)
    I = S.gl(np.eye(4))
    depth = S.sensor_depth(W, H, 0.9)
    rng = np.random.default_rng(99)
    # (a) dust in a 0.05 x 0.03 m window at 1 m: about 10 x 6 pixels wide, thousands of layers
    n = 60000
    centre = np.stack([rng.uniform(-0.02, 0.03, n), rng.uniform(-0.02, 0.01, n), rng.uniform(0.9, 1.1, n)], axis=1)
    va = (centre[:, None, :] + rng.normal(scale=0.006, size=(n, 3, 3))).reshape(-1, 3).astype(np.float32)
    # (b) long needles from in front of the camera to behind it
    n2 = 120000
    a = np.stack([rng.uniform(-0.3, 0.3, n2), rng.uniform(-0.2, 0.2, n2), rng.uniform(0.3, 2.0, n2)], axis=1)
    b = a + np.stack([rng.normal(scale=0.01, size=n2), rng.normal(scale=0.01, size=n2), -rng.uniform(2.5, 4.0, n2)], axis=1)
    c = a + rng.normal(scale=0.004, size=(n2, 3))
    vb = np.stack([a, b, c], axis=1).reshape(-1, 3).astype(np.float32)
    # (c) slivers across the image, all in front of the camera: every one touches more than four tiles, so all of
    # them go through the many-tile lists (bigrec_kernel), which overflow
    n3 = 90000
    a3 = np.stack([rng.uniform(-0.5, 0.5, n3), rng.uniform(-0.25, 0.25, n3), rng.uniform(0.8, 2.0, n3)], axis=1)
    ang = rng.uniform(0, 2 * np.pi, n3)
    b3 = a3 + np.stack([0.9 * np.cos(ang), 0.9 * np.sin(ang), rng.normal(scale=0.05, size=n3)], axis=1)
    c3 = a3 + rng.normal(scale=0.004, size=(n3, 3))
    vc = np.stack([a3, b3, c3], axis=1).reshape(-1, 3).astype(np.float32)
    # (d) 6 - 9 pixel wide triangles (records of the small-box class, never fragments) all inside one tile
    n4 = 30000
    c4 = np.stack([rng.uniform(-0.07, -0.04, n4), rng.uniform(-0.10, -0.05, n4), rng.uniform(0.95, 1.05, n4)], axis=1)
    vd = (c4[:, None, :] + rng.uniform(-0.022, 0.022, size=(n4, 3, 3)) * np.array([1.0, 0.3, 1.0])).reshape(-1, 3).astype(np.float32)
    for case, verts in (("a", va), ("b", vb), ("c", vc), ("d", vd)):
        tris = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
        om, ok = O.filter_frame(depth, P, [(I, 0, [0.0, 0.0, 0.0], verts, tris)], I, I, replace_value=5.0)
        ctx = R.Context(W, H, 1, 0, params(5.0, 0.05))
        m = ctx.add_model()
        ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
        ctx.finalize_models()
        ctx.set_camera(0, P, I, I)
        ctx.set_link_poses(0, m, np.stack([I]))
        masked, mask = ctx.filter_batch(depth[None])
        st = ctx.stats()
        assert st["regrowths"] >= 1, (case, st)
        assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0]), case
        masked, mask = ctx.filter_batch(depth[None])                 # steady state: no further growth, same result
        assert ctx.stats()["regrowths"] == st["regrowths"], (case, st, ctx.stats())
        assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0]), case
        ctx.close()


@pytest.mark.parametrize("size", [(640, 480), (333, 251), (96, 64)])
def test_screen_filling_layers(size):
    """Stacked screen-filling triangles (walls): coincident duplicates in both diagonal splits, a layer behind,
    a tilted layer that cuts through them, a nearer partial layer, a layer a few ulps behind -- the tile
    kernel's cooperative pass classifies tiles against the edges and skips triangles that a complete nearer
    layer hides; every pixel must still be the oracle's."""
    W, H = size
    fx = 525.0 * W / 640
    P = S.projection(fx, fx, (W - 1) / 2, (H - 1) / 2, W, H)
    rng = np.random.default_rng(W)

    def quad(z, u0=-0.2, u1=1.2, v0=-0.2, v1=1.2, flip=False):
        """z: depth at the four corners (tl, tr, bl, br); u, v: fractions of the image (beyond 0..1 = off-screen)."""
        pts = []
        for (u, v), zz in zip(((u0, v0), (u1, v0), (u0, v1), (u1, v1)), z):
            pts.append(((u * W - (W - 1) / 2) / fx * zz, (v * H - (H - 1) / 2) / fx * zz, zz))
        v = np.array(pts, np.float32)
        t = np.array([[0, 1, 2], [2, 1, 3]] if not flip else [[0, 1, 3], [0, 3, 2]], np.uint32)
        return v, t

    layers = [quad((2.0,) * 4, flip=True), quad((2.0,) * 4), quad((2.0,) * 4), quad((3.0,) * 4),
              quad((1.6, 2.4, 1.6, 2.4)), quad((1.2,) * 4, u1=0.4), quad((2.0000002,) * 4),
              quad((2.5, 2.5, 1.7, 1.7), u0=0.3)]
    if W == 96:
        # more whole-tile triangles per tile than the workgroup's list holds (255): the rest stay with their wave
        for _ in range(200):
            z = rng.uniform(1.0, 4.0) + rng.uniform(-0.3, 0.3, 4) * (rng.random() < 0.5)
            layers.append(quad(tuple(float(x) for x in np.broadcast_to(z, 4)), flip=bool(rng.integers(0, 2))))
    soup = S.soup_geometry(rng, n_links=2, tris_per_link=150, scale_lo=0.02, scale_hi=0.3)
    geo = [(0, [0.0, 0.0, 0.0], v, t) for v, t in layers] + soup
    n = 3
    ctx = R.Context(W, H, n, 0, params())
    m = ctx.add_model()
    for pre, op, v, t in geo:
        l = ctx.add_link(m)
        ctx.add_draw(m, l, v, t, pre, op)
    ctx.finalize_models()
    depth = np.stack([S.sensor_depth(W, H, 0.7 * s) for s in range(n)])
    per = []
    ident = np.eye(4).T.reshape(16)
    for s in range(n):
        tfs = [ident.copy() for _ in layers] + S.random_link_poses(rng, len(soup))
        offinv, camtf = (None, None) if s == 0 else S.random_camera(rng, small=True)
        ctx.set_camera(s, P, offinv, camtf)
        ctx.set_link_poses(s, m, np.stack(tfs))
        per.append((tfs, offinv, camtf))
    masked, mask = ctx.filter_batch(depth)
    check_vs_oracle(masked, mask, P, geo, depth, per)
    assert mask[0].astype(bool).mean() > 0.5          # the layers really fill the view
    ctx.close()


@pytest.mark.parametrize("mode", ["fused", "two_kernel", "bits"])
@pytest.mark.parametrize("size", [(640, 480), (1280, 720), (200, 150)])
def test_whole_tile_cover_planes_near_and_far(size, mode):
    """A triangle that covers a whole tile becomes that tile's initial depth keys and hides what lies behind it
    (bigrec_kernel<0> / <1>, tile kernel): many-tile records behind it are never appended, bin records and fragments
    behind it are dropped when loaded.  Layers in the exact-z range (window z <= 0.5: closer than ~0.2 m, where the
    cover's float z has to be written by the exact-z pass although it is in no bin), a tilted layer that crosses
    z = 0.5 inside the image, coincident layers (draw order decides), layers a 24-bit step apart, partial layers,
    and dust of small triangles (fragments and lane-walk records) in front of, inside and behind all of them.
    Every pixel must be the oracle's."""
    W, H = size
    fx = 525.0 * W / 640
    P = S.projection(fx, fx, (W - 1) / 2, (H - 1) / 2, W, H)
    rng = np.random.default_rng(W + len(mode))

    def quad(z, u0=-0.2, u1=1.2, v0=-0.2, v1=1.2, flip=False):
        pts = []
        for (u, v), zz in zip(((u0, v0), (u1, v0), (u0, v1), (u1, v1)), z):
            pts.append(((u * W - (W - 1) / 2) / fx * zz, (v * H - (H - 1) / 2) / fx * zz, zz))
        v = np.array(pts, np.float32)
        t = np.array([[0, 1, 2], [2, 1, 3]] if not flip else [[0, 1, 3], [0, 3, 2]], np.uint32)
        return v, t

    def dust(n, z_lo, z_hi, size_px):
        """n small triangles (about size_px pixels across) at depths z_lo..z_hi all over the image"""
        z = rng.uniform(z_lo, z_hi, n)
        cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
        d = rng.uniform(-size_px, size_px, (n, 3, 2))
        v = np.zeros((n, 3, 3), np.float32)
        for k in range(3):
            v[:, k, 0] = (cx + d[:, k, 0] - (W - 1) / 2) / fx * z
            v[:, k, 1] = (cy + d[:, k, 1] - (H - 1) / 2) / fx * z
            v[:, k, 2] = z * (1.0 + 0.01 * rng.uniform(-1, 1, n))
        return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)

    cases = {
        "far walls, dust everywhere": [quad((2.0,) * 4), quad((3.0,) * 4, flip=True), quad((2.0,) * 4, flip=True), quad((1.5, 2.5, 1.5, 2.5)),
                                       dust(3000, 0.5, 4.0, 2.0), dust(600, 0.5, 4.0, 9.0)],
        "cover inside the exact-z range": [quad((0.15,) * 4), quad((0.17,) * 4, flip=True), quad((0.15,) * 4, flip=True),
                                           dust(2000, 0.11, 0.3, 3.0), dust(500, 0.11, 0.3, 12.0), quad((0.12,) * 4, u0=0.3, u1=0.6)],
        "tilted cover across window z = 0.5": [quad((0.17, 0.24, 0.17, 0.24)), quad((0.24, 0.17, 0.30, 0.2), flip=True), quad((0.1975,) * 4),
                                               dust(2500, 0.12, 0.5, 2.5), dust(400, 0.12, 0.5, 10.0)],
        "one 24-bit step apart": [quad((2.0,) * 4), quad((2.0000002,) * 4, flip=True), quad((1.9999998,) * 4), dust(1500, 1.99, 2.01, 4.0)],
    }
    n = len(cases)
    geo, owner = [], []
    for ci, layers in enumerate(cases.values()):
        for v, t in layers:
            geo.append((0, [0.0, 0.0, 0.0], v, t))
            owner.append(ci)
    kw = dict(two_kernel=True) if mode == "two_kernel" else {}
    ctx = R.Context(W, H, n, 0, params(**kw))
    # one model per case, every stream renders its own case
    models = [ctx.add_model() for _ in range(n)]
    for (pre, op, v, t), ci in zip(geo, owner):
        l = ctx.add_link(models[ci])
        ctx.add_draw(models[ci], l, v, t, pre, op)
    ctx.finalize_models()
    ident = np.eye(4).T.reshape(16)
    depth = np.stack([S.sensor_depth(W, H, 0.4 * s) for s in range(n)])
    depth[1] *= 0.08          # sensor values around the near layers, so that both outcomes of the compare occur
    depth[2] *= 0.1
    for s in range(n):
        ctx.set_stream_models(s, [models[s]])
        ctx.set_camera(s, P, None, None)
        for ci in range(n):
            nl = owner.count(ci)
            ctx.set_link_poses(s, models[ci], np.stack([ident] * nl))
    if mode == "bits":
        pin_in = ctx.host_alloc((n, H, W), np.float32)
        pin_bits = ctx.host_alloc((n, ctx.mask_bits_words()), np.uint32)
        pin_in[...] = depth
        ctx.filter_batch_bits_async(pin_in, pin_bits)
        ctx.sync()
        both = [R.expand_mask_bits(depth[s], pin_bits[s], 5.0) for s in range(n)]
        masked, mask = np.stack([b[0] for b in both]), np.stack([b[1] for b in both])
    else:
        masked, mask = ctx.filter_batch(depth)
    for s in range(n):
        draws = [(ident,) + g for g, ci in zip(geo, owner) if ci == s]
        om, ok = O.filter_frame(depth[s], P, draws, None, None, replace_value=5.0)
        assert (ok != mask[s]).sum() == 0, "case %d (%s): %d mask pixels differ" % (s, list(cases)[s], int((ok != mask[s]).sum()))
        assert bits_equal(om, masked[s]), "case %d (%s): masked depth differs" % (s, list(cases)[s])
    st = ctx.stats()
    if "cover_tiles" in st:
        assert st["cover_tiles"] > 0 and st["occluded_entries"] > 0
    assert 0.05 < mask.astype(bool).mean() < 0.999
    ctx.close()


def test_cover_pass_switches_itself_off_and_on():
    """The cover pass (bigrec_kernel<0> + the cover-aware tile kernel) is an optimisation the context switches by what the
    batches show: a robot without any triangle that covers a whole tile runs three batches with it, then without (probing
    every 64th batch); when walls come into the picture the next probe finds covers and it stays on.  Every batch -- with the
    pass, without it, at the switches -- must be the oracle's."""
    n, W, H = 2, 640, 360
    wl = WL.pr2_workload(n, W, H, total_triangles=12000, walls=True)
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    depth = wl.depth_batch()
    far = wl.link_tf[1].copy()
    far[:, :, 14] += 100.0                      # the walls 100 m away along z: outside the frustum

    def run(walls_tf):
        ctx.set_link_poses_batch(0, ids[1], walls_tf)
        masked, mask = ctx.filter_batch(depth)
        st = ctx.stats()
        for s in range(n):
            draws = []
            for mi, links in enumerate(wl.models):
                tfm = wl.link_tf[0] if mi == 0 else walls_tf
                for li, dl in enumerate(links):
                    for d in dl:
                        draws.append((tfm[s, li], d.pre_op, d.op, d.verts, d.tris))
            om, ok = O.filter_frame(depth[s], wl.projection[s], draws, wl.offset_inv[s], wl.cam_tf[s], max_diff=wl.max_diff, replace_value=wl.replace_value)
            assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s]), (s, st["cover_pass"])
        return st

    passes = [run(far)["cover_pass"] for _ in range(6)]
    assert passes == [1, 1, 1, 0, 0, 0], passes                     # three idle batches, then off
    seen = []
    for k in range(70):                                             # walls in view: used from the next probe on
        st = run(wl.link_tf[1])
        seen.append((st["cover_pass"], st["cover_tiles"] > 0))
    first_on = [i for i, (p, _) in enumerate(seen) if p][0]
    assert 50 <= first_on <= 64 and all(p and c for p, c in seen[first_on:]), (first_on, seen[first_on:first_on + 4])
    ctx.close()


def test_config_c4_720p_pr2_plus_walls():
    """BASELINE config 4 shape: 1280x720, PR2-like robot + two static wall URDFs (full-screen boxes incl.
    quirk Q1: exercises the large-triangle path), several streams."""
    n = 3
    wl = WL.pr2_workload(n, 1280, 720, total_triangles=20000, walls=True)
    ctx = R.Context(1280, 720, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    depth = wl.depth_batch()
    masked, mask = ctx.filter_batch(depth)
    for s in range(n):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
    ctx.close()


def test_config_c5_distinct_urdfs_share_a_context():
    """BASELINE config 5 shape: several distinct articulated URDFs, a few streams each, one context per
    GPU; every stream renders only its own robot (rtuf_set_stream_models)."""
    robots = [WL.pr2_workload(3, 320, 240, total_triangles=t, seed=sd, first_state_seed=fs)
              for t, sd, fs in ((3000, 7, 3000), (6000, 8, 3100), (4500, 9, 3200), (9000, 10, 3300))]
    n = 3 * len(robots)
    ctx = R.Context(320, 240, n, 0, params())
    mids = []
    for wl in robots:
        m = ctx.add_model()
        for draws in wl.models[0]:
            l = ctx.add_link(m)
            for d in draws:
                ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
        mids.append(m)
    ctx.finalize_models()
    depth = np.stack([S.sensor_depth(320, 240, 0.11 * s) for s in range(n)])
    for r, wl in enumerate(robots):
        for k in range(3):
            s = 3 * r + k
            ctx.set_camera(s, wl.projection[k], wl.offset_inv[k], wl.cam_tf[k])
            ctx.set_stream_models(s, [mids[r]])
            ctx.set_link_poses(s, mids[r], wl.link_tf[0][k])
    masked, mask = ctx.filter_batch(depth)
    for r, wl in enumerate(robots):
        for k in range(3):
            s = 3 * r + k
            om, ok = O.filter_frame(depth[s], wl.projection[k], wl.oracle_draws(k), wl.offset_inv[k], wl.cam_tf[k], replace_value=5.0)
            assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s]), "robot %d stream %d" % (r, k)
    ctx.close()


@pytest.mark.parametrize("two_kernel", [False, True])
def test_fused_16uc1_io(two_kernel):
    """uint16 millimetres in/out with the reference's convertTo arithmetic fused into the kernels."""
    wl = WL.pr2_workload(3, 320, 240, total_triangles=6000)
    ctx = R.Context(320, 240, 3, 0, params(wl.replace_value, wl.max_diff, two_kernel))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    rng = np.random.default_rng(4)
    mm = np.clip(np.nan_to_num(wl.depth_batch(), nan=0.0, posinf=0.0) * 1000.0 + rng.integers(-3, 4, (3, 240, 320)), 0, 65535).astype(np.uint16)
    mm[:, 0, :4] = [0, 1, 65535, 7900]
    out16, mask = ctx.filter_batch_u16(mm)
    for s in range(3):
        d32 = depth_u16_to_f32(mm[s])
        om, ok = O.filter_frame(d32, wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0
        assert np.array_equal(out16[s], depth_f32_to_u16(om))
        assert np.array_equal(out16[s][ok == 0], mm[s][ok == 0])          # unfiltered pixels round-trip exactly
        assert (out16[s][ok > 0] == 5000).all()
    ctx.close()


def unpack_bits(bits, W, H):
    """[words] uint32 -> [H, W] bool (pixel x = bit x % 32 of word y * ceil(W/32) + x / 32)."""
    rw = (W + 31) // 32
    b = np.unpackbits(np.ascontiguousarray(bits, np.uint32).reshape(H, rw).view(np.uint8), axis=1, bitorder="little")
    return b[:, :W].astype(bool)


@pytest.mark.parametrize("size", [(640, 480), (324, 100), (1280, 720)])
def test_mask_bits_output_equals_full_planes(size):
    """rtuf_filter_batch_bits*: 1 bit per pixel instead of masked depth + byte mask.  The bits must be the byte mask
    of the full-plane call, rtuf_expand_mask_bits must rebuild both full planes bit for bit (32FC1 and 16UC1), for
    host planes (asynchronous, two batches in flight) and device planes; widths that are not a multiple of 32 / 64
    (partial words, partial tiles) included."""
    import torch
    W, H = size
    n = 5
    wl = WL.pr2_workload(n, W, H, total_triangles=12000, first_state_seed=77, walls=(W == 1280))
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    wl.stage(ctx, ids)
    depth = wl.depth_batch()
    masked, mask = ctx.filter_batch(depth)
    words = ctx.mask_bits_words()
    assert words == H * ((W + 31) // 32)
    # host planes, two batches in flight
    pin_in = ctx.host_alloc((n, H, W), np.float32)
    pin_bits = [ctx.host_alloc((n, words), np.uint32) for _ in range(2)]
    pin_in[...] = depth
    for b in pin_bits:
        b[...] = 0xdeadbeef
        ctx.filter_batch_bits_async(pin_in, b)
    ctx.sync()
    assert np.array_equal(pin_bits[0], pin_bits[1])
    for s in range(n):
        assert np.array_equal(unpack_bits(pin_bits[0][s], W, H), mask[s] > 0), s
        m2, k2 = R.expand_mask_bits(depth[s], pin_bits[0][s], wl.replace_value)
        assert bits_equal(m2, masked[s]) and np.array_equal(k2, mask[s])
    # device planes
    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depth).to(dev)
    d_bits = torch.zeros((n, words), dtype=torch.int32, device=dev)
    ctx.filter_batch_device_bits(n, d_depth.data_ptr(), d_bits.data_ptr())
    ctx.sync()
    assert np.array_equal(d_bits.cpu().numpy().view(np.uint32), pin_bits[0])
    # 16UC1 in
    mm = depth_f32_to_u16(np.nan_to_num(depth, nan=0.0, posinf=0.0))
    masked16, mask16 = ctx.filter_batch_u16(mm)
    u_in = ctx.host_alloc((n, H, W), np.uint16)
    u_in[...] = mm
    ctx.filter_batch_bits_async(u_in, pin_bits[1])
    ctx.sync()
    for s in range(n):
        assert np.array_equal(unpack_bits(pin_bits[1][s], W, H), mask16[s] > 0), s
        m2, k2 = R.expand_mask_bits(mm[s], pin_bits[1][s], wl.replace_value)
        assert np.array_equal(m2, masked16[s]) and np.array_equal(k2, mask16[s])
    # errors: two-kernel mode has no bits output
    p2 = params(wl.replace_value, wl.max_diff, two_kernel=True)
    ctx.set_params(p2)
    with pytest.raises(R.RtufError):
        ctx.filter_batch_bits_async(pin_in, pin_bits[0])
    ctx.close()


def test_mask_bits_refused_when_the_background_quad_does_not_cover_the_image():
    """With a projection whose background quad leaves pixels uncovered the reference's masked depth there is the GL
    clear colour, not the sensor value: the mask bits would not expand to the reference's result, so retiring such
    a batch fails loudly (the full-plane call handles the case)."""
    W, H = 320, 240
    P = S.projection(262.5, 262.5, 159.5, 119.5, W, H)
    P[12] = 160.0          # x translation in clip space: the +-100 m quad no longer covers the left part of the image
    ctx = R.Context(W, H, 1, 0, params())
    m = ctx.add_model()
    l = ctx.add_link(m)
    ctx.add_draw(m, l, np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
    ctx.finalize_models()
    T = np.eye(4)
    T[2, 3] = 2.0
    ctx.set_camera(0, P, None, None)
    ctx.set_link_poses(0, m, S.gl(T)[None])
    depth = S.sensor_depth(W, H, 0.5)[None]
    masked, mask = ctx.filter_batch(depth)
    om, ok = O.filter_frame(depth[0], P, [(S.gl(T), 0, (0, 0, 0), np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))], replace_value=5.0)
    assert (ok != mask[0]).sum() == 0 and bits_equal(om, masked[0])
    uncovered = (masked[0] == 0) & (depth[0] != 0) & (mask[0] == 0)
    assert uncovered.any()                    # the scene really has clear-colour pixels
    bits = np.zeros((1, ctx.mask_bits_words()), np.uint32)
    ctx.filter_batch_bits_async(np.ascontiguousarray(depth), bits)
    with pytest.raises(R.RtufError) as e:
        ctx.sync()
    assert "background quad" in str(e.value)
    masked2, mask2 = ctx.filter_batch(depth)          # the context stays usable
    assert bits_equal(masked2, masked[0][None])
    ctx.close()


@pytest.mark.parametrize("pipes", [2, 3])
def test_pipelines_inside_one_context(pipes):
    """rtuf_params.pipelines: one context, several internal raster pipelines; batches alternate between them and up to
    2 x pipelines are in flight.  Every batch (new joint state + sensor planes each) must be the oracle's result,
    through device planes and through asynchronous host planes; setters reach all pipelines; wait_oldest retires in
    submission order; the single-stream rtuf_filter() and the stats work on the front context."""
    import torch
    n, W, H = 4, 320, 240
    wls = [WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=seed) for seed in (1000, 2000, 3000)]
    A = wls[0]
    p = params(A.replace_value, A.max_diff, pipelines=pipes)
    ctx = R.Context(W, H, n, 0, p)
    ids = A.load_into(ctx)
    A.load_kinematics(ctx, ids)
    assert ctx.num_triangles() == A.n_triangles()
    steps = 2 * pipes + 3
    depths = [A.depth_batch(first=5 * k) for k in range(steps)]
    dev = torch.device("cuda:0")
    d_in = [torch.from_numpy(d).to(dev) for d in depths]
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(steps)]
    torch.cuda.synchronize()
    ctx.enable_timing(2)
    for k in range(steps):                     # nothing waits in between: up to 2 x pipes batches in flight
        wls[k % 3].stage_joint_positions(ctx, ids, first_call=(k == 0))
        ctx.filter_batch_device(n, d_in[k].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    ctx.sync()

    def oracle(wl, depth, s):
        return O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                              max_diff=wl.max_diff, replace_value=wl.replace_value)
    for k in range(steps):
        for s in (0, n - 1):
            om, ok = oracle(wls[k % 3], depths[k], s)
            assert np.array_equal(ok, outs[k][1][s].cpu().numpy()) and bits_equal(om, outs[k][0][s].cpu().numpy()), (k, s)
    st = ctx.stats()
    assert st["timed_batches"] == steps and st["sum_ms_raster"] > 0 and st["triangles_submitted"] == A.n_triangles() * n
    # asynchronous host planes, retired one by one in submission order
    pin_in = [ctx.host_alloc((n, H, W), np.float32) for _ in range(pipes + 1)]
    pin_out = [ctx.host_alloc((n, H, W), np.float32) for _ in range(pipes + 1)]
    pin_mask = [ctx.host_alloc((n, H, W), np.uint8) for _ in range(pipes + 1)]
    for i in range(pipes + 1):
        pin_in[i][...] = depths[i]
        pin_out[i][...] = -1.0
        wls[(i + 1) % 3].stage_joint_positions(ctx, ids, first_call=False)
        ctx.filter_batch_async(pin_in[i], pin_out[i], pin_mask[i])
    for i in range(pipes + 1):
        ctx.wait_oldest()
        om, ok = oracle(wls[(i + 1) % 3], depths[i], 1)
        assert np.array_equal(ok, pin_mask[i][1]) and bits_equal(om, pin_out[i][1]), i      # complete as soon as it is retired
    ctx.wait_oldest()                          # nothing pending: no-op
    # a setter reaches every pipeline: new threshold, then one batch per pipeline
    p2 = params(A.replace_value, 0.2, pipelines=pipes)
    ctx.set_params(p2)
    wls[0].stage_joint_positions(ctx, ids, first_call=False)
    for k in range(pipes):
        ctx.filter_batch_device(n, d_in[0].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    ctx.sync()
    om, ok = O.filter_frame(depths[0][2], A.projection[2], A.oracle_draws(2), A.offset_inv[2], A.cam_tf[2], max_diff=0.2, replace_value=A.replace_value)
    for k in range(pipes):
        assert np.array_equal(ok, outs[k][1][2].cpu().numpy()) and bits_equal(om, outs[k][0][2].cpu().numpy()), k
    # single-stream shape on the front context (host poses: forward kinematics switched off for stream 0)
    ctx.set_camera(0, None, None, A.cam_tf[0])
    ctx.set_link_poses(0, ids[0], A.link_tf[0][0])
    md, mk = ctx.filter(depths[1][0], A.projection[0])
    om, ok = O.filter_frame(depths[1][0], A.projection[0], A.oracle_draws(0), A.offset_inv[0], A.cam_tf[0], max_diff=0.2, replace_value=A.replace_value)
    assert np.array_equal(ok, mk) and bits_equal(om, md)
    ctx.close()
    with pytest.raises(R.RtufError):
        R.Context(W, H, n, 0, params(pipelines=9))


def test_fk_camera_shift_and_reverting_to_the_host_camera():
    """A camera posed by on-device forward kinematics (camera_frame >= 0) gets the camera_tx_/camera_ty_ origin shift of
    src/urdf_filter.cpp:607-611 through rtuf_set_camera_shift; going back to camera_frame = -1, or handing explicit link
    matrices for a stream, brings back the camera transform the host set (both batch slots)."""
    n = 4
    wl = WL.pr2_workload(n, 320, 240, total_triangles=8000, first_state_seed=31)
    ctx = R.Context(320, 240, n, 0, params(wl.replace_value, wl.max_diff))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.load_kinematics(ctx, ids)
    L = wl.link_tf[0].shape[1]
    host_cam = np.stack([S.gl(np.array([[1, 0, 0, 0.01 * s], [0, 1, 0, 0.02], [0, 0, 1, 0.03], [0, 0, 0, 1.0]])) for s in range(n)])
    ctx.set_cameras(0, wl.projection, wl.offset_inv, host_cam)

    def check(cam_expected, tol):
        for _ in range(2):                               # both batch slots
            masked, mask = ctx.filter_batch(depth)
            tf, cam = ctx.read_poses(n, L)
            assert np.abs(cam - cam_expected).max() <= tol
            for s in (0, n - 1):
                draws = [(tf[s, li], d.pre_op, d.op, d.verts, d.tris) for li, dl in enumerate(wl.models[0]) for d in dl]
                om, ok = O.filter_frame(depth[s], wl.projection[s], draws, wl.offset_inv[s], cam[s], max_diff=wl.max_diff, replace_value=wl.replace_value)
                assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])

    # 1. robot-mounted camera, no shift: inverse(fixed <- camera frame)
    ctx.set_joint_positions(0, ids[0], wl.joint_q, None, wl.camera_frame_index)
    check(wl.cam_tf, 1e-12)
    # 2. with the stereo-baseline shift: origin + right * tx + down * ty, right / down = columns 0 / 1 of the rotation
    tx, ty = np.linspace(0.01, 0.07, n), np.linspace(-0.02, 0.02, n)
    ctx.set_camera_shift(0, tx, ty)
    shifted = wl.cam_tf.copy().reshape(n, 4, 4)          # GL column-major: [col][row]
    for s in range(n):
        shifted[s, 3, :3] += shifted[s, 0, :3] * tx[s] + shifted[s, 1, :3] * ty[s]
    check(shifted.reshape(n, 16), 1e-12)
    assert np.abs(shifted.reshape(n, 16) - wl.cam_tf).max() > 1e-3
    # 3. camera_frame = -1: the host-set camera transforms again (the FK kernel had overwritten the device copies)
    ctx.set_joint_positions(0, ids[0], wl.joint_q, None, -1)
    check(host_cam, 0.0)
    # 4. robot camera again, then explicit link matrices for every stream: forward kinematics off, host camera back
    ctx.set_joint_positions(0, ids[0], wl.joint_q, None, wl.camera_frame_index)
    check(shifted.reshape(n, 16), 1e-12)
    ctx.set_link_poses_batch(0, ids[0], wl.link_tf[0])
    check(host_cam, 0.0)
    ctx.close()


def test_general_forward_kinematics_kernel_for_trees_of_more_than_256_frames():
    """Kinematic trees that do not fit the tree-sweep kernel's LDS (more than 256 frames) take the general per-frame
    kernel: 300 frames (50 six-joint limbs on a base), matrices against the host forward kinematics, image against the
    oracle fed the device matrices."""
    rng = np.random.default_rng(12)
    limbs, per = 50, 6
    xml = ['<robot name="hydra"><link name="base"/>']
    for a in range(limbs):
        parent = "base"
        for k in range(per):
            name = "l%d_%d" % (a, k)
            geo = '<visual><geometry><box size="0.05 0.03 0.12"/></geometry></visual>' if k == per - 1 else ""
            xml.append('<link name="%s">%s</link>' % (name, geo))
            jt = ("revolute", "prismatic", "fixed")[(a + k) % 3]
            xml.append('<joint name="j%d_%d" type="%s"><origin xyz="%.3f %.3f %.3f" rpy="%.3f %.3f %.3f"/><parent link="%s"/><child link="%s"/><axis xyz="%.3f %.3f %.3f"/><limit lower="-1" upper="1"/></joint>'
                       % (a, k, jt, *rng.uniform(-0.08, 0.08, 3), *rng.uniform(-0.5, 0.5, 3), parent, name, *rng.normal(size=3)))
            parent = name
    xml.append("</robot>")
    xml = "".join(xml)
    model = urdf.Model.from_string(xml)
    assert len(model.links) == 1 + limbs * per > 256
    n, W, H = 3, 160, 120
    from realtime_urdf_filter_amd.filter import URDFRenderer
    tf0 = urdf.StaticTransformProvider()
    rd = URDFRenderer(xml, "", "base", "base", tf0, "visual", 1.0, [])
    strip = lambda nm: nm[1:] if nm.startswith("/") else nm
    kin = urdf.kinematic_arrays(model, [strip(r.name) for r in rd.renderables_], [r.link_offset for r in rd.renderables_])
    assert len(kin["parent"]) == 301
    ctx = R.Context(W, H, n, 0, params())
    m = ctx.add_model()
    for r in rd.renderables_:
        l = ctx.add_link(m)
        for d in r.draws:
            ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
    ctx.finalize_models()
    ctx.set_kinematics(m, kin["parent"], kin["joint_type"], kin["joint_origin"], kin["joint_axis"], kin["link_frame"], kin["link_offset"])
    P = S.projection(131.25, 131.25, 79.5, 59.5, W, H)
    root = np.eye(4)
    root[:3, 3] = (0.0, 0.0, 1.2)                       # the hydra one metre in front of the camera
    qs, expect = [], []
    for s in range(n):
        q = {j: float(rng.uniform(-0.6, 0.6)) for j in model.joints}
        qs.append(urdf.joint_vector(kin, q))
        fk = urdf.forward_kinematics(model, q, urdf.Transform(root[:3, :3], root[:3, 3]))
        expect.append(np.stack([(fk[strip(r.name)] * r.link_offset).opengl_matrix() for r in rd.renderables_]))
    ctx.set_cameras(0, np.tile(P, (n, 1)), np.tile(np.eye(4).reshape(16), (n, 1)), np.tile(np.eye(4).reshape(16), (n, 1)))
    ctx.set_joint_positions(0, m, np.stack(qs), np.tile(S.gl(root), (n, 1)), -1)
    depth = np.stack([S.sensor_depth(W, H, 0.3 * s) for s in range(n)])
    masked, mask = ctx.filter_batch(depth)
    tf, cam = ctx.read_poses(n, len(rd.renderables_))
    assert np.abs(tf - np.stack(expect)).max() < 1e-12
    for s in range(n):
        draws = [(tf[s, li], d.pre_op, d.op, d.verts, d.tris) for li, r in enumerate(rd.renderables_) for d in r.draws]
        om, ok = O.filter_frame(depth[s], P, draws, None, None, replace_value=5.0)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s])
        assert (mask[s] > 0).sum() > 50                 # the limbs are in view
    ctx.close()


@pytest.mark.parametrize("cap", [0, 1])
def test_staging_next_link_matrices_while_batches_are_in_flight(cap):
    """The TF-driven shape of the reference at many streams: cameras and link matrices from the host every frame.  Their
    staging is a ring of sets (one per batch in flight + one being written), so frame k+1's matrices are staged while
    frames k and k-1 are still on the GPU, nothing waits in between -- and a bin regrowth (cap=1), which re-runs the
    batches in flight, must re-read each batch's own set, not what was staged since."""
    import torch
    n, W, H = 5, 320, 240
    wls = [WL.pr2_workload(n, W, H, total_triangles=8000, first_state_seed=seed) for seed in (100, 200, 300, 400)]
    A = wls[0]
    ctx = R.Context(W, H, n, 0, params(A.replace_value, A.max_diff, bin_capacity=cap))
    ids = A.load_into(ctx)
    steps = 7
    depths = [A.depth_batch(first=3 * k) for k in range(steps)]
    dev = torch.device("cuda:0")
    d_in = [torch.from_numpy(d).to(dev) for d in depths]
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(steps)]
    torch.cuda.synchronize()

    def stage(k):
        wl = wls[k % 4]
        if k % 3 == 0:
            ctx.set_cameras(0, wl.projection, wl.offset_inv, wl.cam_tf)            # everything
            ctx.set_link_poses_batch(0, ids[0], wl.link_tf[0])
        elif k % 3 == 1:
            ctx.set_cameras(0, None, None, wl.cam_tf)                              # camera transform only (the rest carries over)
            for s in range(n):                                                     # stream by stream
                ctx.set_link_poses(s, ids[0], wl.link_tf[0][s])
        else:
            for s in range(n):
                ctx.set_camera(s, None, None, wl.cam_tf[s])
            ctx.set_link_poses_batch(1, ids[0], wl.link_tf[0][1:])                 # partial ranges
            ctx.set_link_poses_batch(0, ids[0], wl.link_tf[0][:1])

    stage(0)
    for k in range(steps):
        ctx.filter_batch_device(n, d_in[k].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
        if k + 1 < steps:
            stage(k + 1)                       # while up to two batches are in flight
    ctx.sync()
    for k in range(steps):
        wl = wls[k % 4]
        masked, mask = outs[k][0].cpu().numpy(), outs[k][1].cpu().numpy()
        for s in range(n):
            om, ok = O.filter_frame(depths[k][s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                    max_diff=wl.max_diff, replace_value=wl.replace_value)
            assert np.array_equal(ok, mask[s]) and bits_equal(om, masked[s]), (cap, k, s)
    assert (ctx.stats()["regrowths"] > 0) == (cap == 1)
    ctx.close()


def test_stl_facets_are_welded_at_load_and_the_image_does_not_change():
    """BASELINE config 2 names STL meshes: every facet brings three vertices of its own (so does everything else Assimp
    imports for the reference, src/renderable.cpp:352-415).  rtuf_finalize_models merges bit-identical positions, so the
    de-indexed model costs the set-up kernel no more vertex work than the indexed one -- and renders the same image."""
    from realtime_urdf_filter_amd import geometry as G
    wl = WL.pr2_workload(3, 640, 480, total_triangles=20000)
    outs, verts = [], []
    for stl in (False, True):
        ctx = R.Context(640, 480, 3, 0, params(wl.replace_value, wl.max_diff))
        ids = []
        for links in wl.models:
            m = ctx.add_model()
            for draws in links:
                l = ctx.add_link(m)
                for d in draws:
                    v, t = (G.load_stl(G.write_binary_stl(d.verts, d.tris)) if stl and len(d.tris) else (d.verts, d.tris))
                    if stl and len(d.tris):
                        assert len(v) == 3 * len(t)
                    ctx.add_draw(m, l, v, t, d.pre_op, d.op)
            ids.append(m)
        ctx.finalize_models()
        wl.stage(ctx, ids)
        depth = wl.depth_batch()
        outs.append(ctx.filter_batch(depth))
        verts.append(ctx.num_vertices())
        assert ctx.num_triangles() == wl.n_triangles()
        ctx.close()
    assert verts[0] == verts[1] and verts[0] < 0.8 * wl.n_triangles()            # ~0.5-0.6 vertices per triangle, not 3
    assert bits_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    om, ok = O.filter_frame(depth[1], wl.projection[1], wl.oracle_draws(1), wl.offset_inv[1], wl.cam_tf[1], max_diff=wl.max_diff, replace_value=wl.replace_value)
    assert (ok != outs[1][1][1]).sum() == 0 and bits_equal(om, outs[1][0][1])


# ---- raster lanes (ABI 5) -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("lanes,group,want_groups", [(0, 0, 3), (3, 8, 8), (3, 6, 11), (2, 0, 2), (2, 24, 4), (2, 8, 8), (1, 0, 1), (1, 24, 3)])
def test_raster_lanes_split_a_batch_into_launch_groups(lanes, group, want_groups):
    """64 streams.  With several raster lanes (own HIP stream + own tile bins each; 0 = the default, three) the batch is
    split into a multiple of the lanes of launch groups that alternate between the lanes; with one lane into as many groups
    as the bins ask for, one after the other.  Too-small bins force a regrowth (every lane's bins grow, both batches run again).  Every stream of every batch
    equals the oracle, whatever the split; rtuf_stats reports it."""
    import torch
    n = 64
    ctx, P, geo, depth, per = run_soups(128, 96, n, seed=77, raster_lanes=lanes, max_inflight_streams=group, bin_capacity=1)
    masked, mask = ctx.filter_batch(depth)
    check_vs_oracle(masked, mask, P, geo, depth, per)
    st = ctx.stats()
    assert st["raster_lanes"] == (lanes or 3) and st["groups_last_batch"] == want_groups and st["regrowths"] >= 1 and st["lanes_side_by_side"] == 1, st
    assert st["work_items"] > 0 and st["triangles_submitted"] == n * 350
    assert (ctx.stream_handle() is None) == (lanes != 1)
    # three device batches back to back (two in flight): the lanes run ahead of each other across batch boundaries
    dev = torch.device("cuda:0")
    d_in = [torch.from_numpy(np.roll(depth, k, axis=0).copy()).to(dev) for k in range(3)]
    outs = [(torch.empty((n, 96, 128), dtype=torch.float32, device=dev), torch.empty((n, 96, 128), dtype=torch.uint8, device=dev)) for _ in range(3)]
    torch.cuda.synchronize()
    for k in range(3):
        ctx.filter_batch_device(n, d_in[k].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    # a caller's own stream ordered behind everything enqueued so far: its copies see finished planes
    s_user = torch.cuda.Stream()
    ctx.order_stream_after_batches(s_user.cuda_stream)
    with torch.cuda.stream(s_user):
        early = [(o[0].clone(), o[1].clone()) for o in outs]
    s_user.synchronize()
    ctx.sync()
    for k in range(3):
        dk = np.roll(depth, k, axis=0)
        check_vs_oracle(outs[k][0].cpu().numpy(), outs[k][1].cpu().numpy(), P, geo, dk, per)
        assert torch.equal(early[k][1], outs[k][1]) and torch.equal(early[k][0].view(torch.int32), outs[k][0].view(torch.int32))
    assert ctx.stats()["regrowths"] == st["regrowths"]
    ctx.close()


def test_small_batches_take_the_lanes_in_turn():
    """Batches below the split size are not split: each takes one lane, the next batch the next lane (so two small batches in
    flight overlap on the GPU).  Six batches of 3 streams with different sensor planes, nothing waits in between."""
    import torch
    n, W, H = 3, 160, 120
    ctx, P, geo, depth, per = run_soups(W, H, n, seed=31)
    dev = torch.device("cuda:0")
    d_in = [torch.from_numpy((depth + 0.01 * k).astype(np.float32)).to(dev) for k in range(6)]
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(6)]
    torch.cuda.synchronize()
    for k in range(6):
        ctx.filter_batch_device(n, d_in[k].data_ptr(), outs[k][0].data_ptr(), outs[k][1].data_ptr())
    ctx.sync()
    for k in range(6):
        check_vs_oracle(outs[k][0].cpu().numpy(), outs[k][1].cpu().numpy(), P, geo, (depth + 0.01 * k).astype(np.float32), per)
    st = ctx.stats()
    assert st["raster_lanes"] == 3 and st["groups_last_batch"] == 1
    ctx.close()


def test_memory_limit_shrinks_the_launch_group_instead_of_failing():
    """rtuf_params.memory_limit_mb bounds the tile bins of all lanes.  Bins of one record must grow on the first batch; the
    grown bins no longer fit the limit for the launch group the context started with, so the group shrinks (the same bytes
    hold deeper bins for fewer streams, the batch runs in more launches) -- the context stays usable and exact."""
    n = 48
    ctx, P, geo, depth, per = run_soups(128, 96, n, seed=78, bin_capacity=1, memory_limit_mb=1, raster_lanes=2)
    before = ctx.stats()
    assert before["launch_group"] == 6, before          # 2 lanes x 6 streams x 6 tiles x (32 + 1024 x 8) B = 0.56 MiB (12 streams: 1.13)
    masked, mask = ctx.filter_batch(depth)
    check_vs_oracle(masked, mask, P, geo, depth, per)
    st = ctx.stats()
    assert st["regrowths"] >= 1 and st["launch_group"] < before["launch_group"] and st["groups_last_batch"] >= n // st["launch_group"], st
    lanes_bins = 2 * st["launch_group"] * 6 * (st["bin_capacity"] * 32)
    assert lanes_bins <= 1 << 20
    masked2, mask2 = ctx.filter_batch(depth)          # steady state: no further regrowth, same answer
    assert bits_equal(masked, masked2) and np.array_equal(mask, mask2) and ctx.stats()["regrowths"] == st["regrowths"]
    ctx.close()
    with pytest.raises(R.RtufError):
        R.Context(128, 96, n, 0, params(raster_lanes=4))


def test_lanes_that_share_a_hardware_queue_still_filter_exactly(tmp_path):
    """The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues; rtuf_create measures whether the second lane's
    stream runs beside the first's and looks for another if not.  With ONE hardware queue no stream can: the context must
    say so (rtuf_stats.lanes_side_by_side == 0) and still produce the oracle's frames, split batches included."""
    import subprocess
    import sys
    prog = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_parity_gpu as T
ctx, P, geo, depth, per = T.run_soups(128, 96, 40, seed=91)
masked, mask = ctx.filter_batch(depth)
T.check_vs_oracle(masked, mask, P, geo, depth, per)
st = ctx.stats()
print("side_by_side", st["lanes_side_by_side"], "groups", st["groups_last_batch"])
ctx.close()
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, cwd=root, env=dict(os.environ, GPU_MAX_HW_QUEUES="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "side_by_side 0 groups 3" in r.stdout, r.stdout
