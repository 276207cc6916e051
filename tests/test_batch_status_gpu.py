"""GPU (-m gpu): the device-resident status word of a batch (ABI 6, rtuf_batch_status_device).

The reference's filter() returns with final pixels (src/urdf_filter.cpp:237, :729-735).  Here the rasteriser's working
buffers are sized from what earlier batches needed, and a batch that outgrows one is run again by the HOST when it retires
the batch -- so a consumer that reads the device planes on its own stream, ordered behind the batch's kernels with
rtuf_order_stream_after_batches but ahead of the host, can meet provisional pixels.  The status word says so on the device:
these tests force the overflow, read planes and word from such a stream, and check that (a) the word is non-zero exactly
when the planes are not the oracle's, (b) after the host has retired the batch the word is 0 and the planes are."""
import numpy as np
import pytest

import scenes as S
import realtime_urdf_filter_amd as R
from bench_support import workloads as WL
from oracle import bindings as O

pytestmark = pytest.mark.gpu


def params(replace=5.0, max_diff=0.05, **kw):
    p = R.default_params()
    p.filter_replace_value = replace
    p.depth_distance_threshold = max_diff
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


class _DeviceWord:
    """One uint32 of device memory as a __cuda_array_interface__ object (torch.as_tensor wraps it without a copy)."""
    def __init__(self, ptr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (1,), "typestr": "<i4", "version": 2}


class Consumer:
    """A caller's own HIP stream behind the filter: ordered after the batches on the device, never waits for the host."""
    def __init__(self, ctx):
        import torch
        self.torch, self.ctx, self.stream = torch, ctx, torch.cuda.Stream()

    def read(self, masked, mask):
        """Copies (status word, planes) of the batch enqueued last on the consumer's stream; returns host values."""
        torch = self.torch
        word = torch.as_tensor(_DeviceWord(self.ctx.batch_status_device()), device=masked.device)
        self.ctx.order_stream_after_batches(self.stream.cuda_stream)
        with torch.cuda.stream(self.stream):
            w, m, k = word.clone(), masked.clone(), mask.clone()
        self.stream.synchronize()                       # the consumer's stream only: nothing is retired
        return int(w.cpu().numpy()[0]) & 0xffffffff, m.cpu().numpy(), k.cpu().numpy(), word


def test_status_word_marks_overflowed_batches_until_the_host_has_rerun_them():
    """3,000 triangles of 6-9 pixels on one tile.  Batch 1 (the pile 1 m further back, bins of one record): overflow.
    Batch 2 (the pile moved to 1 m: a fill many times what batch 1 sized the bins for -- the changing scene): overflow
    again.  Batch 3 (same scene): final at once."""
    import torch
    W, H, n = 256, 128, 4
    P = S.projection(210.0, 210.0, (W - 1) / 2, (H - 1) / 2, W, H)
    I = S.gl(np.eye(4))
    rng = np.random.default_rng(5)
    nt = 3000
    c = np.stack([rng.uniform(-0.07, -0.04, nt), rng.uniform(-0.10, -0.05, nt), rng.uniform(0.95, 1.05, nt)], axis=1)
    verts = (c[:, None, :] + rng.uniform(-0.022, 0.022, size=(nt, 3, 3)) * np.array([1.0, 0.3, 1.0])).reshape(-1, 3).astype(np.float32)
    tris = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    back = np.eye(4); back[2, 3] = 1.0
    depth = np.stack([S.sensor_depth(W, H, 0.9 + 0.2 * s) for s in range(n)])
    ctx = R.Context(W, H, n, 0, params(bin_capacity=1))
    m = ctx.add_model()
    ctx.add_draw(m, ctx.add_link(m), verts, tris, 0, [0.0, 0.0, 0.0])
    ctx.finalize_models()
    dev = torch.device("cuda:0")
    d = torch.from_numpy(depth).to(dev)
    masked = torch.zeros((n, H, W), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    user = Consumer(ctx)
    seen = []
    for step, pose in enumerate((S.gl(back), I, I)):
        for s in range(n):
            ctx.set_camera(s, P, I, I)
            ctx.set_link_poses(s, m, np.stack([pose]))
        want = [O.filter_frame(depth[s], P, [(pose, 0, [0.0, 0.0, 0.0], verts, tris)], I, I, replace_value=5.0) for s in range(n)]
        masked.zero_(); mask.zero_(); torch.cuda.synchronize()
        ctx.filter_batch_device(n, d.data_ptr(), masked.data_ptr(), mask.data_ptr())
        word, early_m, early_k, word_t = user.read(masked, mask)
        early_ok = all(np.array_equal(want[s][1], early_k[s]) and bits_equal(want[s][0], early_m[s]) for s in range(n))
        assert (word & R.STATUS_PENDING_MASK) == 0, hex(word)            # ordered behind the batch: every launch group had finished
        # the word is 0 exactly when the planes the consumer saw were final
        assert (word == 0) == early_ok, (step, hex(word), early_ok)
        seen.append(word)
        ctx.wait_oldest()                                                # the host retires the batch: overflowed ones are run again
        st = ctx.stats()
        assert st["batch_status"] == word and (st["batch_reruns"] >= 1) == (word != 0), (step, hex(word), st)
        torch.cuda.synchronize()
        assert int(word_t.cpu().numpy()[0]) == 0                         # ... and the word says the planes are final now
        got_m, got_k = masked.cpu().numpy(), mask.cpu().numpy()
        for s in range(n):
            assert np.array_equal(want[s][1], got_k[s]) and bits_equal(want[s][0], got_m[s]), (step, s)
    assert seen[0] & R.STATUS_BIN_OVERFLOW and seen[1] & R.STATUS_BIN_OVERFLOW and seen[2] == 0, [hex(w) for w in seen]
    ctx.close()


@pytest.mark.parametrize("strict", [False, True])
def test_status_word_reports_a_setup_grid_that_was_too_short(strict):
    """Robot out of view of eight of nine cameras (short work lists), then in view of all: the set-up grid sized from
    the previous batch does not cover the list -- RTUF_STATUS_GRID_SHORT, provisional planes, a re-run at retirement.
    With RTUF_FLAG_STRICT_GRID every launch takes the worst-case grid: the same batch is final at once."""
    import torch
    n, W, H = 9, 320, 240
    wl = WL.pr2_workload(n, W, H, total_triangles=20000)
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff, flags=R.FLAG_STRICT_GRID if strict else 0))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.stage(ctx, ids)
    away = wl.cam_tf.copy().reshape(n, 4, 4)
    away[1:, 3, :3] += np.array([0.0, 0.0, 50.0])
    ctx.set_cameras(0, wl.projection, wl.offset_inv, away.reshape(n, 16))
    ctx.filter_batch(depth)
    ctx.filter_batch(depth)                    # (bins sized, hints taken from a batch that ran without regrowth)
    ctx.set_cameras(0, wl.projection, wl.offset_inv, wl.cam_tf)
    before = ctx.stats()["regrowths"]
    dev = torch.device("cuda:0")
    d = torch.from_numpy(depth).to(dev)
    masked = torch.zeros((n, H, W), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    user = Consumer(ctx)
    ctx.filter_batch_device(n, d.data_ptr(), masked.data_ptr(), mask.data_ptr())
    word, early_m, early_k, word_t = user.read(masked, mask)
    want = [O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                           max_diff=wl.max_diff, replace_value=wl.replace_value) for s in range(n)]
    early_ok = all(np.array_equal(want[s][1], early_k[s]) and bits_equal(want[s][0], early_m[s]) for s in range(n))
    if strict:
        assert word == 0 and early_ok, hex(word)
    else:
        assert word & R.STATUS_GRID_SHORT and not (word & R.STATUS_PENDING_MASK), hex(word)
        assert not early_ok                    # (chunks beyond the grid were never set up: the robot is missing triangles)
    ctx.sync()
    st = ctx.stats()
    assert st["regrowths"] == before + (0 if strict else 1) and st["batch_status"] == word, st
    torch.cuda.synchronize()
    assert int(word_t.cpu().numpy()[0]) == 0
    got_m, got_k = masked.cpu().numpy(), mask.cpu().numpy()
    for s in range(n):
        assert np.array_equal(want[s][1], got_k[s]) and bits_equal(want[s][0], got_m[s]), s
    ctx.close()


@pytest.mark.parametrize("n,group", [(5, 4), (13, 4), (7, 3)])
def test_uneven_last_launch_group_keeps_its_own_grid_estimate(n, group):
    """Work lists are not linear in the streams (every chunk rounds its visible streams up to whole items): with launch groups
    of 3 + 2 streams and more than 400 visible chunks the smaller group needs more than its share of the larger group's list.
    Every group takes its own list length of the previous batch as the estimate, so steady-state batches run once -- and a
    batch whose grid does turn out too short is run again with the worst-case grid instead of the same estimate."""
    W, H = 160, 120
    wl = WL.pr2_workload(n, W, H, total_triangles=400000)
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff, max_inflight_streams=group, raster_lanes=1))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.stage(ctx, ids)
    want = [O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                           max_diff=wl.max_diff, replace_value=wl.replace_value) for s in range(n)]
    ctx.filter_batch(depth)
    ctx.filter_batch(depth)
    st0 = ctx.stats()
    assert st0["groups_last_batch"] == -(-n // group) and st0["work_items"] > 384 * st0["groups_last_batch"], st0
    for _ in range(3):
        masked, mask = ctx.filter_batch(depth)
        st = ctx.stats()
        assert st["regrowths"] == st0["regrowths"] and st["batch_status"] == 0, st
        for s in range(n):
            assert np.array_equal(want[s][1], mask[s]) and bits_equal(want[s][0], masked[s]), s
    ctx.close()


@pytest.mark.parametrize("n,group,lanes", [(100, 7, 2), (64, 5, 2), (40, 6, 3), (33, 4, 2)])
def test_status_word_reaches_zero_when_the_groups_do_not_divide_the_batch(n, group, lanes):
    """Several lanes round the number of launch groups up to a multiple of the lanes, and ceil(n / that) streams per group can
    then come to FEWER groups than that multiple (100 streams: 16 asked for, 15 groups of 7 made).  The status word starts at
    the number of groups and every finished group takes one off: it must start at the number MADE, or a device-side consumer
    never sees 0 (round 5's advisor finding).  Read from a consumer's stream and again after the host has retired the batch."""
    import torch
    W, H = 128, 96
    wl = WL.pr2_workload(n, W, H, total_triangles=3000)
    ctx = R.Context(W, H, n, 0, params(wl.replace_value, wl.max_diff, max_inflight_streams=group, raster_lanes=lanes))
    ids = wl.load_into(ctx)
    depth = wl.depth_batch()
    wl.stage(ctx, ids)
    ctx.filter_batch(depth)
    ctx.filter_batch(depth)                    # (bins and grid estimates settled)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(depth).to(dev)
    masked = torch.zeros((n, H, W), dtype=torch.float32, device=dev)
    mask = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    user = Consumer(ctx)
    ctx.filter_batch_device(n, d.data_ptr(), masked.data_ptr(), mask.data_ptr())
    word, early_m, early_k, word_t = user.read(masked, mask)
    st_groups = -(-n // -(-n // (-(-(-(-n // group)) // lanes) * lanes)))      # ceil(n / ceil(n / (ceil(n / group) rounded up to the lanes)))
    assert word == 0, "status word %#x behind the batch's kernels (%d launch groups expected)" % (word, st_groups)
    ctx.sync()
    st = ctx.stats()
    assert st["groups_last_batch"] == st_groups and st["batch_status"] == 0 and st["batch_reruns"] == 0, st
    torch.cuda.synchronize()
    assert int(word_t.cpu().numpy()[0]) == 0
    for s in (0, n // 2, n - 1):
        om, ok = O.filter_frame(depth[s], wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s], max_diff=wl.max_diff, replace_value=wl.replace_value)
        assert np.array_equal(ok, early_k[s]) and bits_equal(om, early_m[s]), s
    ctx.close()
