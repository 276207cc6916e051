"""GPU (-m gpu): the BASELINE.json configs at their REAL sizes -- the 250,388-triangle PR2-like model at batch = 1
(config 2) and batch = 256 (config 3), the per-GPU shares of the two 8-GPU configs (config 4: 64 x 720p + walls,
config 5: 8 distinct URDFs x 128 cameras = 1024 streams) -- through the C ABI, against the CPU oracle on EVERY stream of the headline
workload, of the arm-in-front-of-the-lens pose and of config 4's share (8 per robot for config 5's 1024 streams; the oracle
runs on all host cores) and through size-independent properties on every stream."""
import numpy as np
import pytest

import realtime_urdf_filter_amd as R
from bench_support import configs as CF
from bench_support import workloads as WL
from oracle import bindings as O

pytestmark = pytest.mark.gpu


def params(wl, two_kernel=False):
    p = R.default_params()
    p.filter_replace_value = wl.replace_value
    p.depth_distance_threshold = wl.max_diff
    if two_kernel:
        p.flags |= R.FLAG_TWO_KERNEL
    return p


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def check_properties(depth, masked, mask, replace_value):
    """Size-independent properties of a filtered batch (every stream, every pixel)."""
    # the output is a pure select between the sensor value and the replace value (the background quad covers the frame)
    assert bits_equal(np.where(mask > 0, np.float32(replace_value), depth), masked)
    assert set(np.unique(mask)) <= {0, 255}
    # NaN and zero sensor pixels are never filtered (quirk Q9); +inf always is (quirk Q2)
    assert (mask[np.isnan(depth)] == 0).all() and (mask[depth == 0] == 0).all() and (mask[np.isposinf(depth)] == 255).all()


def check_against_oracle(share, k, streams, depth, masked, mask, link_dev=None, cam_dev=None):
    """The oracle's frames of `streams`, computed on all usable host cores (the C call releases the GIL), against the
    device's planes -- bit for bit."""
    wl0 = share.wl0
    streams = list(streams)
    frames = []
    for s in streams:
        proj, draws, off, cam = share.oracle_frame(k, s, link_dev, cam_dev)
        frames.append(O.PreparedFrame(depth[s], proj, draws, off, cam, max_diff=wl0.max_diff, replace_value=wl0.replace_value))
    O.run_prepared(frames, O.usable_threads())
    for s, f in zip(streams, frames):
        assert (f.mask != mask[s]).sum() == 0, "stream %d: %d mask pixels differ" % (s, int((f.mask != mask[s]).sum()))
        assert bits_equal(f.masked, masked[s]), "stream %d: masked depth differs" % s


@pytest.mark.parametrize("two_kernel", [False, True])
def test_config2_pr2_250k_triangles_batch_1(two_kernel):
    """BASELINE config 2: 640x480 stream, full PR2-like URDF (250 k triangles), batch = 1 -- every pixel of several
    consecutive frames (new joint state and sensor frame each) against the oracle, through the single-stream
    rtuf_filter() shape and through a batch of one."""
    frames = 4
    wl = WL.pr2_workload(frames, 640, 480, total_triangles=250000, first_state_seed=4242)
    assert wl.n_triangles() > 240000
    ctx = R.Context(640, 480, 1, 0, params(wl, two_kernel))
    ids = wl.load_into(ctx)
    assert ctx.num_triangles() == wl.n_triangles()
    for f in range(frames):
        wl.stage(ctx, ids, first=f, n=1)
        depth = wl.depth(100 + f)
        if f % 2 == 0:
            masked, mask = ctx.filter_batch(depth[None])
            masked, mask = masked[0], mask[0]
        else:
            masked, mask = ctx.filter(depth, wl.projection[f])
        om, ok = O.filter_frame(depth, wl.projection[f], wl.oracle_draws(f), wl.offset_inv[f], wl.cam_tf[f],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask).sum() == 0 and bits_equal(om, masked), "frame %d" % f
        assert 0.02 < (mask > 0).mean() < 0.98          # the robot really is in view
    st = ctx.stats()
    assert st["triangles_submitted"] == wl.n_triangles() and st["triangles_binned"] > 1000
    ctx.close()


def test_config3_pr2_250k_triangles_256_streams():
    """BASELINE config 3 (the headline workload, exactly what bench.py runs): 256 VGA streams of the 250 k-triangle
    robot, joint positions through on-device forward kinematics, device-resident planes, the default context (two
    raster lanes, two launch groups per batch).  Properties on all 256 streams, ALL 256 streams against the oracle (fed the
    matrices the GPU's forward kinematics produced), determinism and stream-permutation equivariance with host-staged poses."""
    import torch
    share = CF.build("c3", 1, 0)
    n, W, H = share.n, share.width, share.height
    assert n == 256 and share.wl0.n_triangles() > 240000
    ctx = R.Context(W, H, n, 0, params(share.wl0))
    share.load(ctx)
    dev = torch.device("cuda:0")
    depth = share.depth_host(1)
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for k in (0, 1):          # two steps: the second one re-stages joint positions only, like the bench loop
        share.stage(ctx, k)
        ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
        ctx.sync()
    masked, mask = d_masked.cpu().numpy(), d_mask.cpu().numpy()
    check_properties(depth, masked, mask, share.wl0.replace_value)
    link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
    assert share.host_fk_error(1, link_dev, cam_dev) < 1e-12
    check_against_oracle(share, 1, range(n), depth, masked, mask, link_dev, cam_dev)
    st = ctx.stats()
    assert st["raster_lanes"] == 3 and st["groups_last_batch"] == 3 and st["launch_group"] == 86 and st["device_bytes"] < 10e9, st
    # a second run is identical
    ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
    ctx.sync()
    assert bits_equal(d_masked.cpu().numpy(), masked) and np.array_equal(d_mask.cpu().numpy(), mask)
    st = ctx.stats()
    assert st["triangles_submitted"] == share.wl0.n_triangles() * n
    ctx.close()
    # reversing the stream order reverses the outputs (host-staged matrices: a context without forward kinematics)
    wl = share.groups[0].variants[1]
    ctx = R.Context(W, H, n, 0, params(wl))
    ids = wl.load_into(ctx)
    ctx.set_cameras(0, wl.projection[::-1], wl.offset_inv[::-1], wl.cam_tf[::-1])
    ctx.set_link_poses_batch(0, ids[0], wl.link_tf[0][::-1])
    d_rev = torch.from_numpy(np.ascontiguousarray(depth[::-1])).to(dev)
    ctx.filter_batch_device(n, d_rev.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
    ctx.sync()
    m3, k3 = d_masked.cpu().numpy()[::-1], d_mask.cpu().numpy()[::-1]
    # (host FK and device FK agree to ~1e-15, not bit for bit: compare the reversed run with the oracle instead)
    check_properties(depth, m3, k3, wl.replace_value)
    for s in (0, 77, 255):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != k3[s]).sum() == 0 and bits_equal(om, m3[s])
    ctx.close()


def test_config4_per_gpu_share_64_streams_720p_with_walls():
    """BASELINE config 4 at its per-GPU share: rank 0 of 8 = 64 of the 512 streams, 1280x720, 250 k-triangle robot +
    the two wall URDFs (screen-filling boxes incl. quirk Q1)."""
    import torch
    share = CF.build("c4", 8, 0)
    n, W, H = share.n, share.width, share.height
    assert (n, W, H) == (64, 1280, 720) and share.total_streams == 512 and share.wl0.n_triangles() > 240000
    ctx = R.Context(W, H, n, 0, params(share.wl0))
    share.load(ctx)
    dev = torch.device("cuda:0")
    depth = share.depth_host(0)
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for k in (1, 2):
        share.stage(ctx, k)
        ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
        ctx.sync()
    masked, mask = d_masked.cpu().numpy(), d_mask.cpu().numpy()
    check_properties(depth, masked, mask, share.wl0.replace_value)
    link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
    check_against_oracle(share, 2, range(n), depth, masked, mask, link_dev, cam_dev)      # every stream of the share
    assert (mask > 0).mean() > 0.1           # the walls fill a good part of the view
    ctx.close()


def test_config5_per_gpu_share_8_urdfs_x_128_streams():
    """BASELINE config 5 at its per-GPU share: rank 0 of 8 holds URDFs 0, 8, ..., 56 -> 8 distinct robots (30 k to 250 k
    triangles) x 128 cameras = 1024 streams in one context (two launch groups of 512 on two raster lanes); every stream
    renders only its own robot; forward kinematics of all 8 trees on the GPU.  Eight streams per robot against the oracle."""
    import torch
    share = CF.build("c5", 8, 0)
    n, W, H = share.n, share.width, share.height
    assert n == 1024 and len(share.groups) == 8 and [g.robot_index for g in share.groups] == list(range(0, 64, 8))
    ctx = R.Context(W, H, n, 0, params(share.wl0))
    share.load(ctx)
    dev = torch.device("cuda:0")
    depth = share.depth_host(0)
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    share.stage(ctx, 0)
    ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
    ctx.sync()
    masked, mask = d_masked.cpu().numpy(), d_mask.cpu().numpy()
    check_properties(depth, masked, mask, share.wl0.replace_value)
    link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
    streams = [g.first + j for g in share.groups for j in (0, 3, 19, 42, 64, 90, 111, 127)]
    check_against_oracle(share, 0, streams, depth, masked, mask, link_dev, cam_dev)
    ctx.close()


def run_share(share, k, oracle_streams, variant=0):
    """One step of a share through the device-plane call; returns what check_* need."""
    import torch
    n, W, H = share.n, share.width, share.height
    ctx = R.Context(W, H, n, 0, params(share.wl0))
    share.load(ctx)
    dev = torch.device("cuda:0")
    depth = share.depth_host(variant)
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for kk in (k - 1, k):          # two steps: the second one sizes its set-up grid from the first
        share.stage(ctx, kk)
        ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
        ctx.sync()
    masked, mask = d_masked.cpu().numpy(), d_mask.cpu().numpy()
    check_properties(depth, masked, mask, share.wl0.replace_value)
    link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
    check_against_oracle(share, k, oracle_streams, depth, masked, mask, link_dev, cam_dev)
    st = ctx.stats()
    ctx.close()
    return mask, st


def test_config3_forearm_in_front_of_the_lens_256_streams():
    """The self-filter's own normal case: the robot's arm right in front of the sensor.  All 256
    streams of the headline workload pose the right forearm 0.1 - 0.35 m in front of the head camera (window z on both
    sides of 0.5: the exact-z pass runs, the gripper crosses the near plane, the arm covers whole tiles and hides the
    robot behind it).  Properties on every stream, all 256 streams against the oracle.  The winners' float z -- finer than
    the 24-bit depth below window z 0.5 -- comes out of the depth keys' low bits: the exact-z pass, which round 3 ran in
    two thirds of these tiles, is left with those that hold geometry within micrometres of the near plane (where the
    gripper is cut by it)."""
    share = CF.build("c3", 1, 0, near_arm=True)
    assert share.n == 256 and share.wl0.n_triangles() > 240000
    mask, st = run_share(share, 1, range(256), variant=1)
    assert (mask > 0).mean() > 0.3           # the arm fills a good part of every view
    assert st["exact_tiles"] < 12000, st     # (round 3: 24,048 of the 38,400 tiles; what is left are the tiles where the gripper crosses the near plane)


@pytest.mark.parametrize("workload,rank", [("c4", 5), ("c5", 3)])
def test_other_ranks_shares_of_the_8_gpu_configs(workload, rank):
    """Shares other than rank 0's of BASELINE configs 4 and 5 (other global stream numbers -> other joint states,
    for c5 other URDFs: 3, 11, ..., 59) at their full per-GPU size, two streams each against the oracle."""
    share = CF.build(workload, 8, rank)
    if workload == "c4":
        assert (share.n, share.width, share.height) == (64, 1280, 720) and share.groups[0].global_first == 5 * 64
        streams = range(0, 64, 4)
    else:
        assert share.n == 1024 and [g.robot_index for g in share.groups] == list(range(3, 64, 8))
        streams = [g.first + j for g in share.groups for j in (5, 100)]
    run_share(share, 1, streams)


def _multi_gpu_example():
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "multi_gpu_filter")
    if not os.path.exists(exe):
        subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    from conftest import page_in_rccl
    page_in_rccl()              # (the example links librccl: see conftest.page_in_rccl)
    return exe


@pytest.mark.parametrize("mode,masks", [("block", "direct"), ("block", "rccl"), ("model", "direct")])
def test_cpp_multi_device_host_with_rccl_gather(tmp_path, mode, masks):
    _run_cpp_multi_device_host(tmp_path, mode, masks, 0, None)


@pytest.mark.parametrize("mode,masks,logical", [("block", "direct", 2), ("block", "rccl", 2), ("model", "rccl", 2), ("model", "direct", 3)])
def test_cpp_multi_device_host_logical_devices(tmp_path, mode, masks, logical):
    """The same host with N > 1 on a box with ONE GPU: --logical-devices N maps N shares, N host threads and N contexts onto
    the visible devices in turn.  Unequal shares (6 streams over 2, 3 robots over 2 devices: 6 + 3 streams), the padded
    all-gather with its compaction, models_for_rank and the report gather all run with N members; RCCL itself is left out
    (it refuses two ranks on one GPU): the gathers travel as device-to-device copies through the same slot arithmetic.  A
    stream dumped from EACH logical device equals the oracle."""
    for pick in ((1, 4) if mode == "block" else (7, 4)):
        _run_cpp_multi_device_host(tmp_path, mode, masks, logical, pick)


def _run_cpp_multi_device_host(tmp_path, mode, masks, logical, pick_override):
    """examples/multi_gpu_filter.cpp over include/realtime_urdf_filter_amd/multi_gpu.hpp: the C++ host of SURVEY.md
    section 8e -- one thread and one rtuf_context per device, block shares (configs 3 / 4) or URDF m on device m % N
    (config 5), forward kinematics on the device, ncclCommInitAll + one ncclAllGather of {frames, seconds, mismatches}
    and the all-gather of the bit-packed masks (peer-to-peer copies or ncclAllGather).  Runs on every device the box
    has (one here); a dumped stream is checked against the oracle fed the matrices the device rendered with."""
    import json
    import subprocess
    import scene_file
    if mode == "block":
        share = CF.build("c4", 1, 0, streams=6, triangles=20000, width=640, height=360)      # robot + walls, 6 streams
        pick = 4
    else:
        share = CF.build("c5", 1, 0, streams=3, urdfs=3, triangles=15000)                     # 3 robots x 3 cameras
        pick = 7
    if pick_override is not None:
        pick = pick_override
    scene = tmp_path / "scene.bin"
    depth = scene_file.write_scene(str(scene), share, k=0)
    cmd = [_multi_gpu_example(), str(scene), "--mode", mode, "--steps", "3", "--masks", masks, "--dump", str(pick), str(tmp_path / "s")]
    if logical:
        cmd += ["--logical-devices", str(logical)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-500:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["streams"] == share.n and rep["frames"] == share.n * 3 and rep["bits_vs_bytes_mismatches"] == 0
    assert rep["gathered_masks_equal_sources"] == 1 and sum(d["streams"] for d in rep["per_device"]) == share.n
    if logical:
        assert rep["devices"] == logical and rep["logical_devices"] == 1 and len(rep["per_device"]) == logical
        per = [d["streams"] for d in rep["per_device"]]
        assert min(per) >= 1 and (mode == "block" or logical != 2 or sorted(per) == [3, 6]), per      # every member has a share; 3 robots over 2 members: 2 + 1 robots = 6 + 3 streams
        assert rep["mask_all_gather_path"] == ("peer-to-peer copies" if masks == "direct" else "padded all-gather by copies (logical devices: no communicator)")
    else:
        assert rep["mask_all_gather_path"] == ("peer-to-peer copies" if masks == "direct" else "rccl") and rep["peer_access_everywhere"] == 1
    assert all(d["host_thread_pinned_to_cpus"] >= 0 for d in rep["per_device"])
    W, H = share.width, share.height
    masked = np.fromfile(tmp_path / "s.masked.f32", np.float32).reshape(H, W)
    mask = np.fromfile(tmp_path / "s.mask.u8", np.uint8).reshape(H, W)
    link_tf = np.fromfile(tmp_path / "s.link_tf.f64", np.float64).reshape(-1, 16)
    cam_tf = np.fromfile(tmp_path / "s.cam_tf.f64", np.float64)
    order = [int(x) for x in (tmp_path / "s.models.txt").read_text().split()]
    # the oracle's draw list of that stream: the models its device loaded, in that order, with the device's matrices
    g = share.group_of(pick)
    wl = g.variants[0]
    job_models = [(gg, mi) for gg in share.groups for mi in range(len(gg.variants[0].models))]
    draws, row = [], 0
    for gm in order:
        gg, mi = job_models[gm]
        links = gg.variants[0].models[mi]
        if gg is g:
            for li, dl in enumerate(links):
                for d in dl:
                    draws.append((link_tf[row + li], d.pre_op, d.op, d.verts, d.tris))
        row += len(links)
    j = pick - g.first
    om, ok = O.filter_frame(depth[pick % len(depth)], wl.projection[j], draws, wl.offset_inv[j], cam_tf,
                            max_diff=wl.max_diff, replace_value=wl.replace_value)
    assert (ok != mask).sum() == 0 and bits_equal(om, masked)
    assert 0.01 < (mask > 0).mean() < 0.99


def test_launch_group_of_1024_streams_that_all_see_the_whole_model():
    """More than 256 streams per launch group with (nearly) every chunk visible in every stream: cull_kernel emits
    ceil(visible / 3) work items per block of 256 stream slots, i.e. more than ceil(group / 3) per chunk -- the items
    array and the set-up kernel's worst-case grid must be sized for that (out-of-bounds writes otherwise)."""
    import torch
    n, W, H = 600, 160, 120
    rng = np.random.default_rng(5)
    # a multi-chunk mesh ball in front of the camera: every chunk inside every stream's frustum
    wl = WL.pr2_workload(2, W, H, total_triangles=3000)
    nv, nt = 4000, 6000
    v = (rng.normal(size=(nv, 3)) * 0.15).astype(np.float32)
    t = rng.integers(0, nv, size=(nt, 3)).astype(np.uint32)
    p = params(wl)
    p.raster_lanes = 1                # one lane: the whole batch is ONE launch group (several lanes would split it)
    ctx = R.Context(W, H, n, 0, p)
    m = ctx.add_model()
    l = ctx.add_link(m)
    ctx.add_draw(m, l, v, t)
    ctx.finalize_models()
    tfs = np.zeros((n, 1, 16))
    for s in range(n):
        T = np.eye(4)
        T[:3, 3] = (0.02 * np.sin(s), 0.02 * np.cos(s), 2.0 + 0.001 * s)
        tfs[s, 0] = T.T.reshape(16)
    ctx.set_cameras(0, np.tile(wl.projection[0], (n, 1)), np.tile(wl.offset_inv[0], (n, 1)), np.tile(np.eye(4).reshape(16), (n, 1)))
    ctx.set_link_poses_batch(0, m, tfs)
    depth = np.stack([WL.synthetic.sensor_depth(W, H, s) for s in range(8)])
    depth = np.ascontiguousarray(depth[np.arange(n) % 8])
    dev = torch.device("cuda:0")
    d_depth = torch.from_numpy(depth).to(dev)
    d_masked = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    d_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    for _ in range(2):           # the second batch sizes its set-up grid from the first one's list length
        ctx.filter_batch_device(n, d_depth.data_ptr(), d_masked.data_ptr(), d_mask.data_ptr())
        ctx.sync()
    masked, mask = d_masked.cpu().numpy(), d_mask.cpu().numpy()
    for s in (0, 255, 256, 257, 511, 512, 599):
        om, ok = O.filter_frame(depth[s], wl.projection[0], [(tfs[s, 0], 0, (0, 0, 0), v, t)], wl.offset_inv[0], np.eye(4).reshape(16),
                                max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert (ok != mask[s]).sum() == 0 and bits_equal(om, masked[s]), "stream %d" % s
    assert ctx.stats()["groups_last_batch"] == 1 and ctx.stats()["launch_group"] == n
    ctx.close()
