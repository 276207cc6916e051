#!/usr/bin/env python3
"""bench.py -- filtered depth frames/s of the hot path (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path over one batch of synthetic input.  Default workload (`--workload c3`):
`--streams` concurrent 640x480 camera streams of the synthetic PR2-like robot per GPU (config C3 of SURVEY.md
section 8d: "640x480, PR2 URDF, batch=256 concurrent camera streams on 1 MI355X", the configuration the BASELINE
target ">=30 frames/s per stream at >=256 streams" is quoted on).  Every step stages a fresh joint state + camera
pose for every stream, then runs forward kinematics -> pose -> cull -> set-up/binning -> clip -> tile raster +
per-pixel compare on depth frames that are already resident in HBM.  Inputs rotate through `--variants`
distinct pre-generated batches so no step can reuse the previous result.

`--workload c4` / `c5` are the two 8-GPU configs of BASELINE.json (bench_support/configs.py): c4 = 512
720p streams of the robot + two wall URDFs block-partitioned over the ranks, c5 = 64 distinct URDFs x 128 cameras
with URDF m on rank m % N.  Their totals are fixed, so they scale strongly; `--shard-of W` runs rank 0's share of
a W-GPU job on however many GPUs are present (to measure the per-GPU share on one GPU).

Multi-GPU (`--gpus N`, launched by torch.distributed.run): streams are independent, so every rank filters its own
streams with no data-path collective; RCCL carries the barriers around the timed region, the MAX all-reduce of the
elapsed time and the tiny end-of-run gather of per-rank frame and parity counts.

Timed region: exactly `--steps` steps, repeated back to back until at least `--min-seconds` have passed (repetitions
are whole multiples of `--steps`; `timed_steps` in the output says how many steps were timed in total).  The default, 7 s
(the repetition count comes from a short probe, so the region ends up a few per cent either side of it), is long enough
for an outside sampler (amd-smi every 5 s) to land inside it.

`value` comes from ONE default-configured context: three raster lanes, a batch split into three launch groups whose kernels
overlap (rtuf_params.raster_lanes).  Overlapping kernels share the GPU, so their per-launch times describe no single
kernel; the `roofline` object is therefore measured live in the same run on a second context with ONE lane (every kernel
alone on the GPU, the whole batch per launch, HIP events over its own timed region), and `roofline.in_headline_run`
carries the per-launch figures of the two-lane run beside it.  `with_host_copies` (never `value`) is the same workload
with the planes in pinned host memory (upload + kernels + read-back, what the reference's own timer wraps).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def pin_to_numa_node_of_gpu(local_rank):
    """One rank per GPU: keep the rank's host threads on the NUMA node its GPU hangs off (staging buffers are
    first-touched there).  Best effort: silently a no-op where sysfs does not say."""
    try:
        import torch
        bus = torch.cuda.get_device_properties(local_rank).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(local_rank), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(local_rank), "pci_device_id", 0)
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist" % (dom, bus, dev)
        cpus = set()
        for part in open(path).read().strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return len(cpus)
    except Exception:
        pass
    return None


def side_leg(label, workload, job_world, near_arm, streams, seconds, local_rank, triangles, reuse_planes=None):
    """One of the OTHER BASELINE.json configurations, timed for a few seconds after the headline's timed region (never
    `value`): frames/s through a default context (pipelined, inputs resident in HBM, a new joint state every step, forward
    kinematics on the GPU), the tile kernel's per-launch time from a one-lane context (every kernel alone on the GPU), and
    8 streams of the last step against the oracle.  The reference's own timer wraps the whole call for whatever scene is
    loaded (src/urdf_filter.cpp:211-244)."""
    import torch
    import realtime_urdf_filter_amd as R
    from bench_support import configs as CF
    from oracle import bindings as O
    t_leg = time.perf_counter()
    dev = torch.device("cuda", local_rank)
    sh = CF.build(workload, job_world, 0, streams=streams, triangles=triangles, variants=2, near_arm=near_arm, host_fk=False)
    n, W, H = sh.n, sh.width, sh.height
    wl0 = sh.wl0
    # sensor planes: the headline's resident ones where the frame size is the same, else a pool of 8 synthetic planes;
    # stream s sees plane s % pool (the joint states, not the planes, are what changes from step to step)
    if reuse_planes is not None and tuple(reuse_planes.shape[1:]) == (H, W) and reuse_planes.dtype == torch.float32:
        pool = reuse_planes
    else:
        pool = torch.from_numpy(sh.depth_host(0, 8, tiled=False)).to(dev)
    idx = torch.arange(n, device=dev) % pool.shape[0]
    d_in = pool if (n == pool.shape[0]) else pool.index_select(0, idx)
    sets = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(2)]
    out = {"workload": sh.describe(), "streams": n, "size": [W, H]}

    def run(lanes, secs, latency=False):
        p = R.default_params()
        p.filter_replace_value, p.depth_distance_threshold = wl0.replace_value, wl0.max_diff
        p.raster_lanes = lanes
        if lanes == 1:
            p.max_inflight_streams = min(n, 1024)
        ctx = R.Context(W, H, n, local_rank, p)
        sh.load(ctx)

        def submit(k):
            ctx.filter_batch_device(n, d_in.data_ptr(), sets[k % 2][0].data_ptr(), sets[k % 2][1].data_ptr())

        for k in range(3):                       # bin sizing, cover-pass decision
            sh.stage(ctx, k)
            submit(k)
            ctx.sync()
        k0 = 3
        sh.stage(ctx, k0)
        tq = time.perf_counter()
        for k in range(k0, k0 + 4):
            submit(k)
            sh.stage(ctx, k + 1)
        ctx.sync()
        est = (time.perf_counter() - tq) / 4
        steps = max(8, int(np.ceil(secs / max(est, 1e-9))))
        k0 += 4
        ctx.enable_timing(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if latency:
            for k in range(k0, k0 + steps):      # one frame in flight: stage, filter, wait
                sh.stage(ctx, k)
                submit(k)
                ctx.sync()
        else:
            for k in range(k0, k0 + steps):
                submit(k)
                sh.stage(ctx, k + 1)
            ctx.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = ctx.stats()
        g = max(1, st["groups_last_batch"])
        res = {"frames_per_s": n * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps, "device_memory_bytes": st["device_bytes"],
               "launch_groups_per_batch": g, "raster_lanes": st["raster_lanes"], "regrowths": st["regrowths"],
               "tile_ms": st["sum_ms_raster"] / max(st["timed_batches"], 1) / g, "setup_ms": st["sum_ms_setup"] / max(st["timed_batches"], 1) / g}
        return ctx, res, k0 + steps - 1

    ctx, head, k_last = run(0, seconds)
    out.update({"frames_per_s": head["frames_per_s"], "ms_per_step": head["ms_per_step"], "steps": head["steps"],
                "device_memory_bytes": head["device_memory_bytes"], "raster_lanes": head["raster_lanes"],
                "launch_groups_per_batch": head["launch_groups_per_batch"]})
    # parity: 8 streams of the last step, the oracle fed the matrices the device's forward kinematics produced
    link_dev, cam_dev = ctx.read_poses(n, sh.n_links_total)
    hm, hk = sets[k_last % 2][0].cpu().numpy(), sets[k_last % 2][1].cpu().numpy()
    check = sorted(set(int(x) for x in np.linspace(0, n - 1, num=min(8, n))))
    frames = [O.PreparedFrame(d_in[s].cpu().numpy(), *sh.oracle_frame(k_last, s, link_dev, cam_dev), max_diff=wl0.max_diff, replace_value=wl0.replace_value) for s in check]
    O.run_prepared(frames, max(1, min(len(os.sched_getaffinity(0)), len(frames))))
    bad = sum(int((pf.mask != hk[s]).sum()) + int((pf.masked.view(np.uint32) != hm[s].view(np.uint32)).sum()) for s, pf in zip(check, frames))
    out.update({"frames_checked": len(check), "mismatching_values": bad})
    ctx.close()
    if n == 1:
        # BASELINE config 2: one camera, one frame in flight -- the latency of a filter() call with resident planes
        ctx, lat, _ = run(0, max(0.5, seconds / 2), latency=True)
        out["latency_us_per_frame"] = lat["ms_per_step"] * 1e3
        out["latency_note"] = "stage joint state -> rtuf_filter_batch_device -> rtuf_sync, one frame in flight, planes resident in HBM"
        ctx.close()
    # the tile kernel alone on the GPU: one-lane context, all streams per launch
    ctx, one, _ = run(1, max(0.5, seconds / 2))
    alg = 9 * W * H * (n / one["launch_groups_per_batch"])
    out["tile_kernel"] = {"avg_launch_ms": one["tile_ms"], "streams_per_launch": n / one["launch_groups_per_batch"], "algorithmic_bytes_per_launch": int(alg),
                          "frac": (alg / (one["tile_ms"] * 1e-3) / 1e9 / 8000.0) if one["tile_ms"] > 0 else None, "peak": 8000.0, "unit": "GB/s",
                          "measured_on": "one-lane context of this leg (HIP events around the kernel, every eighth batch), %d steps" % one["steps"],
                          "one_lane_frames_per_s": one["frames_per_s"], "setup_kernel_avg_launch_ms": one["setup_ms"]}
    ctx.close()
    del sets, d_in
    torch.cuda.empty_cache()
    out["leg_seconds"] = time.perf_counter() - t_leg
    return label, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=7.0, help="the timed region repeats the --steps steps until it is at least this long")
    ap.add_argument("--workload", choices=["c3", "c4", "c5"], default="c3", help="BASELINE.json config: c3 (headline, weak scaling), c4, c5 (fixed totals, sharded)")
    ap.add_argument("--streams", type=int, default=None, help="c3: concurrent camera streams per GPU (256); c4: streams in total (512); c5: cameras per URDF (128)")
    ap.add_argument("--urdfs", type=int, default=64, help="c5: distinct URDFs in total")
    ap.add_argument("--shard-of", type=int, default=0, help="take the shares of a job of this many GPUs (ranks 0..N-1 of it) instead of a job of --gpus GPUs")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--triangles", type=int, default=250000, help="triangle budget of the PR2-like model")
    ap.add_argument("--variants", type=int, default=16, help="distinct joint states (and head-camera poses) per stream that the steps cycle through: every step poses every stream anew")
    ap.add_argument("--depth-variants", type=int, default=2, help="distinct sets of sensor planes resident in HBM that the steps alternate between (a set is one plane per stream: 315 MB for 256 VGA streams)")
    ap.add_argument("--pipelines", type=int, default=1, help="contexts (HIP stream + bins each) per GPU that the batches alternate between: with 2 or 3, one batch's small and low-occupancy kernels overlap another's heavy ones, but kernels then share the GPU and per-launch times (roofline) no longer describe one kernel; default 1")
    ap.add_argument("--launch-group", type=int, default=0, help="rtuf_params.max_inflight_streams: streams rasterised per internal launch group (0 = automatic: the streams divided by the raster lanes, the whole batch up to 1024 with one lane); smaller groups shrink the tile bins and cost a kernel sequence per group")
    ap.add_argument("--overlap-pipelines", type=int, default=0, help="(obsolete, ignored: the overlap is the library default now, rtuf_params.raster_lanes; kept so that older command lines still run)")
    ap.add_argument("--lanes", type=int, default=0, help="rtuf_params.raster_lanes of the headline context (0 = the library's default, 3: launch groups alternate between three HIP streams with bins of their own; 1: one lane)")
    ap.add_argument("--isolated-seconds", type=float, default=2.0, help="length of the one-lane leg the roofline is measured on (N=1 only; 0 disables: the roofline then carries the headline run's per-launch times)")
    ap.add_argument("--host-copy-seconds", type=float, default=3.0, help="length of each with_host_copies leg (planes in pinned host memory; N=1 only; 0 disables)")
    ap.add_argument("--two-kernel", action="store_true", help="rasteriser + separate compare kernel")
    ap.add_argument("--host-poses", action="store_true", help="stage explicit link matrices from the host instead of joint positions + on-device forward kinematics")
    ap.add_argument("--u16", action="store_true", help="16UC1 depth in/out (uint16 millimetres) with the conversions fused into the kernels")
    ap.add_argument("--no-mask", action="store_true", help="need_mask_ == false: no mask output")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget per leg (single thread, all cores); 0 disables")
    ap.add_argument("--other-configs-seconds", type=float, default=2.0, help="length of each `other_configs` leg (BASELINE configs 2, 4, 5 and the arm-in-front-of-the-lens pose, timed after the headline's region; 0 disables)")
    ap.add_argument("--other-configs", choices=["auto", "on", "off"], default="auto", help="auto: the legs run with the default c3 command at N=1 (what the driver runs); on: with any c3 command (the legs take --streams / --triangles, the tests' small sizes); off: never")
    ap.add_argument("--bin-capacity", type=int, default=0, help="rtuf_params.bin_capacity (records per tile bin; 0 = the library's default, grown on overflow)")
    ap.add_argument("--debug-flags", type=lambda x: int(x, 0), default=0, help="timing experiments only (needs the RTUF_ABLATE build; results are wrong)")
    ap.add_argument("--near-arm", action="store_true", help="c3 / c4: every stream poses the robot's right forearm 0.1-0.35 m in front of the lens (exact-z pass, near-plane clipping, whole-tile occluders)")
    ap.add_argument("--check-frames", type=int, default=-1, help="frames of the last step verified against the oracle, per rank (-1 = every stream; the oracle runs on all host cores)")
    args = ap.parse_args()

    import torch
    import realtime_urdf_filter_amd as R
    from realtime_urdf_filter_amd import sharding
    from bench_support import configs as CF

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RTUF_BENCH_BACKEND=gloo RTUF_BENCH_DEVICE=0: rehearsal of the multi-rank path on a box with one GPU
    # (all ranks share device 0, collectives over gloo); the real run is one rank per GPU over RCCL
    backend = os.environ.get("RTUF_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("RTUF_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    pinned_cpus = pin_to_numa_node_of_gpu(local_rank) if world > 1 and "RTUF_BENCH_DEVICE" not in os.environ else None
    dist = None
    # launched by torch.distributed.run (also with one rank: the driver's N = 1 run goes straight to python, a torchrun with
    # --nproc-per-node 1 exercises the RCCL init, barrier, all-reduce and all-gather lines on one GPU)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # torch's librccl.so is a third of a gigabyte of code objects; on a fresh box its first use faults it in page by
            # page (communicator set-up has been seen to take ten minutes that way).  One sequential read first: seconds.
            try:
                with open(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "rb", buffering=0) as f:
                    while f.read(16 << 20):
                        pass
            except OSError:
                pass
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else "cpu"       # where the (few-byte) collectives' tensors live

    # --- workload: this rank's share ------------------------------------------------------------
    job_world = args.shard_of or world
    if rank >= job_world:
        raise SystemExit("--shard-of %d: rank %d has no share" % (job_world, rank))
    # (host-side forward kinematics of a variant -- 3 ms per stream in Python -- only where it is needed: --host-poses, and the
    # one variant of the last timed step that the parity check compares the device's kinematics with)
    share = CF.build(args.workload, job_world, rank, streams=args.streams, triangles=args.triangles, variants=max(1, args.variants),
                     width=args.width, height=args.height, urdfs=args.urdfs, near_arm=args.near_arm, host_fk=False)
    n, W, H = share.n, share.width, share.height
    wl0 = share.wl0
    p = R.default_params()
    p.filter_replace_value = wl0.replace_value
    p.depth_distance_threshold = wl0.max_diff
    if args.two_kernel:
        p.flags |= R.FLAG_TWO_KERNEL
    p.flags |= args.debug_flags
    p.bin_capacity = args.bin_capacity
    p.max_inflight_streams = args.launch_group
    p.raster_lanes = args.lanes
    P = max(1, args.pipelines)
    p.pipelines = P if P > 1 else 0          # rtuf_params.pipelines: the library alternates the batches between P internal pipelines
    ctx = R.Context(W, H, n, local_rank, p)
    share.load(ctx, on_device_fk=not args.host_poses)
    if args.host_poses:
        # every variant's host-side kinematics BEFORE the warm-up: computed when first staged (3 ms per stream in Python, 0.8 s per
        # variant of 256 streams) they fell into the timed region -- round 6's first --host-poses line read 33 k frames/s for it
        for g in share.groups:
            for wl in g.variants:
                wl.ensure_host_fk()
    ctx.enable_timing(2)         # HIP events around the big kernels only (each event costs stream time)
    ctxs = [ctx]
    V = share.n_variants()
    VD = max(1, min(V, args.depth_variants))

    d_depth = []
    for v in range(VD):
        host = share.depth_host(v)
        if args.u16:
            host = np.clip(np.rint(np.nan_to_num(host, nan=0.0, posinf=0.0) * 1000.0), 0, 65535).astype(np.uint16).view(np.int16)
        d_depth.append(torch.from_numpy(host).to(dev))
        del host
    # two output sets per pipeline: with two batches in flight per context, a batch must not write where an
    # earlier batch's results are still unread
    n_sets = 2 * P
    d_masked_set = [torch.empty((n, H, W), dtype=torch.int16 if args.u16 else torch.float32, device=dev) for _ in range(n_sets)]
    d_mask_set = [None if args.no_mask else torch.empty((n, H, W), dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    torch.cuda.synchronize()
    ptrs = [(d_masked_set[i].data_ptr(), d_mask_set[i].data_ptr() if d_mask_set[i] is not None else 0) for i in range(n_sets)]
    dptr = [d.data_ptr() for d in d_depth]

    def stage_into(k):
        share.stage(ctx, k)       # joint angles in (forward kinematics on the GPU), or host matrices with --host-poses

    def submit(k):
        (ctx.filter_batch_device_u16 if args.u16 else ctx.filter_batch_device)(n, dptr[k % VD], ptrs[k % n_sets][0], ptrs[k % n_sets][1])

    def enqueue(k):
        # one step = one batch through the hot path: enqueue it, then stage the NEXT batch's joint positions (host
        # memory only) while the GPU works.  The context keeps up to two batches per pipeline in flight and retires
        # the oldest when another arrives; nothing in the loop waits for the GPU otherwise.
        submit(k)
        stage_into(k + 1)

    def isolated_step(k):
        stage_into(k)
        submit(k)
        ctx.sync()

    def barrier():
        if dist is not None:
            dist.barrier()

    k0 = args.warmup * P
    t_w = time.perf_counter()
    for k in range(k0):            # every pipeline warms up (first batch: bin sizing) with one batch in flight
        stage_into(k)
        submit(k)
        ctx.sync()
    torch.cuda.synchronize()
    # how often the --steps steps are repeated so that the timed region is at least --min-seconds long: from the
    # time of a few pipelined steps after the warm-up (agreed between the ranks: the slowest decides)
    stage_into(k0)
    t_p = time.perf_counter()
    probe = 0
    while probe < 8 or (time.perf_counter() - t_p < 0.05 and probe < 4096):
        enqueue(k0 + probe)
        probe += 1
    for c in ctxs:
        c.sync()
    est_step = (time.perf_counter() - t_p) / probe
    k0 += probe
    reps = max(1, int(np.ceil(args.min_seconds / max(est_step * max(args.steps, 1), 1e-9)))) if args.min_seconds > 0 else 1
    if dist is not None:
        t = torch.tensor([reps], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps = int(t.item())
    timed_steps = args.steps * reps
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    barrier()
    for c in ctxs:
        c.enable_timing(3)       # (re)starts the library's event sums: set-up, tile (and compare) kernels of every fourth batch (an event costs ~5 us of stream time)
    stage_into(k0)
    step_clock = np.empty(timed_steps + 1)
    t0 = time.perf_counter()
    step_clock[0] = t0
    for k in range(k0, k0 + timed_steps):
        enqueue(k)
        step_clock[k - k0 + 1] = time.perf_counter()      # (the call returns when the context has room: with two batches in flight that is the pace of the GPU)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_rank = elapsed                          # this rank's own clock (the line's value uses the MAX over ranks)
    sts = [c.stats() for c in ctxs]
    timed = sum(st["timed_batches"] for st in sts)
    assert timed >= 1 and timed >= timed_steps // 8, ([st["timed_batches"] for st in sts], timed_steps)
    st = ctx.stats()
    lanes, groups_per_batch = st["raster_lanes"], max(1, st["groups_last_batch"])
    # per-launch averages of the headline run: the library sums a batch's launch groups, so divide by their number
    head = {k: sum(x["sum_ms_" + k] for x in sts) / timed / groups_per_batch for k in ("setup", "clip", "raster", "compare")}
    overlapping = lanes > 1 or P > 1              # kernels of different groups / batches share the GPU
    k_last = k0 + timed_steps - 1
    frames_rank = n * timed_steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity on every rank: by default EVERY stream of the last timed step (oracle = checker only) ----------------
    from oracle import bindings as O
    from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32
    host_threads = len(os.sched_getaffinity(0))
    quota = None
    try:        # a container's CPU quota (cgroup v2): the cores the threads below can really use at once
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        quota = None
    threads = max(1, min(host_threads, int(quota))) if quota else host_threads
    v_last = k_last % VD
    d_masked, d_mask = d_masked_set[k_last % n_sets], d_mask_set[k_last % n_sets]
    link_dev = cam_dev = None
    fk_err = None
    if not args.host_poses:
        # the oracle is fed the very matrices the GPU's forward kinematics produced
        link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
        fk_err = share.host_fk_error(k_last, link_dev, cam_dev)
    n_check = n if args.check_frames < 0 else min(args.check_frames, n)
    check = sorted(set(int(x) for x in np.linspace(0, n - 1, num=n_check))) if n_check > 0 else []
    h_masked_all = d_masked.cpu().numpy()
    h_mask_all = d_mask.cpu().numpy() if d_mask is not None else None
    h_depth_all = d_depth[v_last].cpu().numpy()

    def fetch(s):
        hm, hd = h_masked_all[s], h_depth_all[s]
        hk = h_mask_all[s] if h_mask_all is not None else None
        if args.u16:
            hm = hm.view(np.uint16)
            hd = depth_u16_to_f32(hd.view(np.uint16))
        return hd, hm, hk

    t_par = time.perf_counter()
    prepared_check = []
    for s in check:
        hd, _, _ = fetch(s)
        proj, draws, off, cam = share.oracle_frame(k_last, s, link_dev, cam_dev)
        prepared_check.append(O.PreparedFrame(hd, proj, draws, off, cam, max_diff=wl0.max_diff, replace_value=wl0.replace_value))
    O.run_prepared(prepared_check, threads)
    bad_mask = bad_depth = 0
    for s, pf in zip(check, prepared_check):
        _, hm, hk = fetch(s)
        if hk is not None:
            bad_mask += int((pf.mask != hk).sum())
        bad_depth += int((depth_f32_to_u16(pf.masked) != hm).sum()) if args.u16 else int((pf.masked.view(np.uint32) != hm.view(np.uint32)).sum())
    t_par = time.perf_counter() - t_par
    frames_total, bad_total, checked_total = frames_rank, bad_mask + bad_depth, len(check)
    per_rank = [{"rank": rank, "streams": n, "frames": frames_rank, "frames_checked": len(check), "mismatching_values": bad_mask + bad_depth,
                 "frames_per_s": frames_rank / elapsed_rank}]
    if dist is not None:
        # the trivial end-of-run gather (a few numbers per rank): frames, parity counts
        frames_total, elapsed = sharding.gather_frame_counts(dist, frames_rank, elapsed, device=cdev)
        t = torch.tensor([n, frames_rank, len(check), bad_mask + bad_depth], dtype=torch.int64, device=cdev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        tf = torch.tensor([elapsed_rank], dtype=torch.float64, device=cdev)
        outf = [torch.zeros_like(tf) for _ in range(world)]
        dist.all_gather(outf, tf)
        # (every rank's own frames / own seconds between the two barriers: a straggler shows here, the line's value is frames / MAX)
        per_rank = [{"rank": r, "streams": int(o[0]), "frames": int(o[1]), "frames_checked": int(o[2]), "mismatching_values": int(o[3]),
                     "frames_per_s": int(o[1]) / float(f[0])} for r, (o, f) in enumerate(zip(outl, outf))]
        bad_total = sum(x["mismatching_values"] for x in per_rank)
        checked_total = sum(x["frames_checked"] for x in per_rank)

    if rank == 0:
        value = frames_total / elapsed
        # pace of the timed region, step by step, on rank 0's host clock: the time between two returns of the enqueue call (which
        # returns when the context has room again: steady state = one GPU step).  The first steps fill the two in-flight slots.
        dsteps = np.diff(step_clock)[min(4, max(timed_steps - 1, 0)):] * 1e3
        step_ms = ({"min": float(dsteps.min()), "median": float(np.median(dsteps)), "p99": float(np.percentile(dsteps, 99)), "max": float(dsteps.max()), "steps": int(dsteps.size),
                    "note": "host-clock interval between consecutive enqueue returns of the timed region (rank 0), first four steps left out"}
                   if dsteps.size else None)
        px = W * H
        two = args.two_kernel
        mask_b = 0 if args.no_mask else 1

        def fresh_share():
            return CF.build(args.workload, job_world, rank, streams=args.streams, triangles=args.triangles, variants=max(1, args.variants),
                            width=args.width, height=args.height, urdfs=args.urdfs, near_arm=args.near_arm, host_fk=False)

        # ---- one-lane leg: every kernel alone on the GPU, the whole batch per launch -- what the roofline describes -----
        iso_leg = None
        breakdown = None
        iso = None
        # (with several ranks: rank 0 measures it on its GPU while the others wait at the closing barrier -- after the timed
        # region, so nothing of it is in `value`)
        if args.isolated_seconds > 0 and not (lanes == 1 and P == 1 and groups_per_batch == 1):
            p1 = R.default_params()
            p1.filter_replace_value, p1.depth_distance_threshold, p1.flags = p.filter_replace_value, p.depth_distance_threshold, p.flags
            p1.raster_lanes, p1.max_inflight_streams, p1.bin_capacity = 1, min(n, 1024), args.bin_capacity
            ctx.sync()
            ctx1 = R.Context(W, H, n, local_rank, p1)
            share1 = fresh_share()
            share1.load(ctx1, on_device_fk=not args.host_poses)
            sets1 = [(torch.empty_like(d_masked_set[0]), None if args.no_mask else torch.empty_like(d_mask_set[0])) for _ in range(2)]

            def submit1(k):
                m, kk = sets1[k % 2]
                (ctx1.filter_batch_device_u16 if args.u16 else ctx1.filter_batch_device)(n, dptr[k % VD], m.data_ptr(), kk.data_ptr() if kk is not None else 0)

            for k in range(max(args.warmup, 2)):
                share1.stage(ctx1, k)
                submit1(k)
                ctx1.sync()
            kb = max(args.warmup, 2)
            share1.stage(ctx1, kb)
            tq = time.perf_counter()
            for k in range(kb, kb + 8):
                submit1(k)
                share1.stage(ctx1, k + 1)
            ctx1.sync()
            est1 = (time.perf_counter() - tq) / 8
            steps1 = max(16, int(np.ceil(args.isolated_seconds / max(est1, 1e-9))))
            kb += 8
            ctx1.enable_timing(3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(kb, kb + steps1):
                submit1(k)
                share1.stage(ctx1, k + 1)
            ctx1.sync()
            torch.cuda.synchronize()
            el1 = time.perf_counter() - t1
            s1 = ctx1.stats()
            g1 = max(1, s1["groups_last_batch"])
            iso_leg = {k: s1["sum_ms_" + k] / max(s1["timed_batches"], 1) / g1 for k in ("setup", "clip", "raster", "compare")}
            iso_leg.update({"timed_launches": s1["timed_batches"] * g1, "launches_per_step": g1, "steps": steps1, "seconds": el1,
                            "frames_per_s": n * steps1 / el1, "device_memory_bytes": s1["device_bytes"]})
            bctx, bshare, bk = ctx1, share1, kb + steps1
        else:
            bctx, bshare, bk = ctx, share, k_last + 1
            sets1 = [(d_masked_set[i], d_mask_set[i]) for i in range(2)]

            def submit1(k):
                m, kk = sets1[k % 2]
                (bctx.filter_batch_device_u16 if args.u16 else bctx.filter_batch_device)(n, dptr[k % VD], m.data_ptr(), kk.data_ptr() if kk is not None else 0)

        # stage-by-stage breakdown and the raster stage's kernels one by one: a few extra steps, one batch in flight, host
        # waits after every step (on the one-lane context when there is one: nothing else on the GPU)
        extra = 3 * V
        acc = {"ms_pose": 0.0, "ms_setup": 0.0, "ms_raster": 0.0, "ms_compare": 0.0, "ms_total": 0.0}
        bctx.enable_timing(1)
        for j in range(extra):
            bshare.stage(bctx, bk + j)
            submit1(bk + j)
            bctx.sync()
            sx = bctx.stats()
            for key in acc:
                acc[key] += sx[key]
        breakdown = {key: v / extra for key, v in acc.items()}
        bctx.enable_timing(2)
        gb = max(1, bctx.stats()["groups_last_batch"])
        iso = {"ms_setup": 0.0, "ms_clip": 0.0, "ms_raster": 0.0, "ms_compare": 0.0}
        for j in range(extra):
            bshare.stage(bctx, bk + extra + j)
            submit1(bk + extra + j)
            bctx.sync()
            sx = bctx.stats()
            for key in iso:
                iso[key] += sx[key] / extra / gb
        if bctx is not ctx:
            bctx.close()
            del sets1
            torch.cuda.empty_cache()

        # algorithmic bytes per launch:
        #   fused tile kernel : 9 B/pixel = 4 sensor read + 4 masked write + 1 mask write (8 without mask; 5 / 4 for 16UC1)
        #   two-kernel mode   : tile kernel writes the 4 B/pixel z-surface; compare moves 13 B/pixel
        #   set-up + clip     : every vertex (12 B) and triangle (12 B indices + 4 B order) of the model once per launch --
        #                       the geometry is shared by all streams; the records it writes are implementation traffic
        # a launch of the one-lane leg covers all n streams, a launch of the headline run n / groups_per_batch of them
        src_ms = iso_leg if iso_leg is not None else head
        launch_streams = n / (iso_leg["launches_per_step"] if iso_leg is not None else groups_per_batch)
        geo_bytes = sum(12 * g.variants[0].n_vertices() + 16 * g.variants[0].n_triangles() for g in share.groups)
        kernels = {}
        if two:
            kernels["tile_kernel<two_kernel>"] = ("raster", 4 * px, "valu-issue / LDS atomics (rasteriser: neither HBM nor MFMA, SURVEY.md 8d); HBM figure for context")
            kernels["compare_kernel"] = ("compare", ((6 if args.u16 else 12) + mask_b) * px, "hbm")
        else:
            kernels["tile_kernel<fused>"] = ("raster", ((4 if args.u16 else 8) + mask_b) * px, "hbm")
        kernels["setup_kernel"] = ("setup", None, "valu-issue (triangle set-up: neither HBM nor MFMA, SURVEY.md 8d); HBM figure for context")
        kernels["clip_kernel"] = ("clip", 0, "latency / divergent scalar code at LDS-limited occupancy; no algorithmic HBM traffic of its own (the time covers clip_kernel and bigrec_kernel, which appends the many-tile records both set-up and clip kernel listed)")
        peak = 8000.0
        # Off-line counter data of this same command (rocprofv3 --pmc passes cannot run inside the timed region):
        # used only when the committed measurement is of this exact workload AND launch shape, and labelled as such.
        default_cmd = (args.workload == "c3" and n == 256 and (W, H) == (640, 480) and not args.no_mask and not args.u16 and args.triangles == 250000 and not args.near_arm and not args.host_poses)
        # (config 4's per-GPU share -- the other workload whose counters are committed)
        c4_share_cmd = (args.workload == "c4" and args.shard_of == 8 and n == 64 and (W, H) == (1280, 720) and not args.no_mask and not args.u16
                        and args.triangles == 250000 and not args.near_arm and not args.host_poses and not args.two_kernel)
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
        except Exception:
            pmc = None
        valu_peak = None
        try:
            valu_peak = json.load(open(os.path.join(ROOT, "profiles", "valu_peak.json")))
        except Exception:
            valu_peak = None
        pmc_ok = pmc is not None and int(round(launch_streams)) == int((pmc or {}).get("streams_per_launch", 256))

        def alg_bytes_of(name, streams):
            _, per_stream, _ = kernels[name]
            return geo_bytes if per_stream is None else int(per_stream * streams)

        def kernel_entry(name):
            key, _, bound = kernels[name]
            dur_ms = src_ms[key]
            alg_bytes = alg_bytes_of(name, launch_streams)
            achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
            e = {"kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                 "traffic": None, "avg_launch_ms": dur_ms, "algorithmic_bytes_per_launch": alg_bytes, "streams_per_launch": launch_streams,
                 "measured_on": ("one-lane context in this run (rtuf_params.raster_lanes = 1, one launch group per batch): HIP events around the kernel on its stream over that leg's %d timed steps, batches pipelined, no other raster kernel on the GPU" % iso_leg["steps"]) if iso_leg is not None
                                else "the headline run's own context (HIP events around the kernel on its stream inside the timed region)"}
            if iso is not None:
                iso_ms = iso["ms_" + key]
                ib = alg_bytes_of(name, n / gb)
                e["isolated"] = {"avg_launch_ms": iso_ms, "frac": (ib / (iso_ms * 1e-3) / 1e9 / peak) if iso_ms > 0 else None,
                                 "note": "same kernel with the host waiting after every batch (nothing else on the GPU at all; %d extra steps)" % extra}
            if iso_leg is not None or overlapping:
                hb = alg_bytes_of(name, n / groups_per_batch)
                hm = head[key]
                e["in_headline_run"] = {"avg_launch_ms": hm, "streams_per_launch": n / groups_per_batch, "algorithmic_bytes_per_launch": hb,
                                        "frac": (hb / (hm * 1e-3) / 1e9 / peak) if hm > 0 else None,
                                        "note": "per-launch time inside the headline's timed region: %d raster lane(s), %d launch groups per batch -- kernels of different groups overlap, so a launch shares the GPU with another kernel and its duration is not that of one kernel alone" % (lanes, groups_per_batch)}
            if name == "compare_kernel" and 4 * px * n < 2 * (256 << 20):
                e["note"] = "HBM + MALL figure: the %d MB z-surface this kernel reads was written by the kernel before it and partly sits in the 256 MiB Infinity Cache; with --streams 1024 (z-surface 1.26 GB) the same kernel measures pure HBM" % (4 * px * n // 1000000)
            rec = (pmc or {}).get("kernels", {}).get(name) if (pmc_ok and default_cmd) else None
            if rec is None and pmc and c4_share_cmd and int(round(launch_streams)) == 64:
                rec = (pmc.get("c4_share") or {}).get("kernels", {}).get(name)
            if rec:
                e["traffic"] = rec.get("hbm_bytes_per_launch")
                e["traffic_source"] = "OFFLINE: " + (pmc.get("c4_share", {}).get("source") if (c4_share_cmd and not default_cmd) else pmc.get("source", "profiles/pmc_counters.json")) + " (rocprofv3 --pmc passes of the one-lane launch shape of this same workload; not measured in this run)"
                cnt = rec.get("wave64_valu_instructions_per_launch")
                if cnt and dur_ms > 0:
                    g = cnt / (dur_ms * 1e-3) / 1e9
                    vi = {"wave64_instructions_per_launch": cnt, "instructions_source": e["traffic_source"], "achieved_G_per_s": g}
                    if valu_peak:
                        # two measured reference rates: the 4-cycle instruction class (most integer / compare / convert
                        # instructions: a lower bound of the mix's peak) and the kernel's own STATIC instruction mix
                        # (scripts/valu_mix.py: harmonic mean over the disassembly's VALU opcodes; an estimate, the
                        # dynamic mix is not observable: SQ_ACTIVE_INST_VALU does not separate the classes)
                        pk = valu_peak.get("peak_G_per_s")
                        if pk:
                            vi.update({"peak_G_per_s": pk, "frac": g / pk, "peak_source": "OFFLINE: profiles/valu_peak.json (scripts/valu_peak.hip on this GPU model: measured issue rate of the 4-cycle instruction class; the 2-cycle class runs at %.0f G/s)" % valu_peak.get("fast_class_G_per_s", 0)})
                        mp = (valu_peak.get("static_mix_peak_G_per_s") or {}).get(name)
                        if mp:
                            vi.update({"static_mix_peak_G_per_s": mp, "frac_of_static_mix_peak": g / mp})
                    if rec.get("live_lane_fraction"):
                        # SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64): of the 64 lanes an issued VALU instruction could
                        # serve, how many were live (divergent walks; profiles/r05_pmc_lanes.txt has it per loop)
                        vi["live_lane_fraction"] = rec["live_lane_fraction"]
                    e["valu_issue"] = vi
            return e

        entries = [kernel_entry(k) for k in kernels]
        # depth tests issued per drawn pixel: needs the instrumented build (-DRTUF_COUNT: a counter in the walks), so it is an
        # OFFLINE figure of this same command (scripts/overdraw.sh -> profiles/overdraw.json), never measured in the timed run
        overdraw = None
        try:
            od = json.load(open(os.path.join(ROOT, "profiles", "overdraw.json")))
            key = "near_arm" if args.near_arm else args.workload
            if default_cmd or key in od:
                rec = od.get(key)
                if rec:
                    overdraw = dict(rec, source="OFFLINE: profiles/overdraw.json (librtuf built with -DRTUF_COUNT, same workload)")
        except Exception:
            overdraw = None
        dom = max(entries, key=lambda e: e["avg_launch_ms"])       # the dominant kernel: longest average launch, no exclusions
        # (round 6: tile and set-up kernel now take the same time to within a microsecond or two and the longer one changes from
        # run to run.  Within 3 % the line names the kernel SURVEY.md 8d's bytes per pixel flow through -- the fused tile kernel --
        # and says so; the other one's entry, with its own time, is the first of all_kernels as ever.)
        tie = None
        if dom["bound"] != "hbm":
            hbm_like = [e for e in entries if e["bound"] == "hbm" and e["avg_launch_ms"] >= 0.97 * dom["avg_launch_ms"]]
            if hbm_like:
                other = dom
                dom = max(hbm_like, key=lambda e: e["avg_launch_ms"])
                tie = "%s %.1f us and %s %.1f us per launch: a tie within the run-to-run spread; the line names the kernel the path's 9 B/pixel flow through" % (
                    dom["kernel"], dom["avg_launch_ms"] * 1e3, other["kernel"], other["avg_launch_ms"] * 1e3)
        if dom["bound"] != "hbm":
            dom = dict(dom, bound_note=dom["bound"], bound="hbm")   # (the contract's vocabulary; the note says what really limits it)
        roof = dict(dom)
        if tie:
            roof["dominant_tie"] = tie
        roof.update({"launches_per_step": iso_leg["launches_per_step"] if iso_leg is not None else groups_per_batch,
                     "timed_launches": iso_leg["timed_launches"] if iso_leg is not None else timed * groups_per_batch,
                     "all_kernels": [e for e in entries if e["kernel"] != dom["kernel"]]})
        if iso_leg is not None:
            roof["one_lane_leg"] = {"frames_per_s": iso_leg["frames_per_s"], "steps": iso_leg["steps"], "seconds": iso_leg["seconds"], "device_memory_bytes": iso_leg["device_memory_bytes"],
                                    "note": "the same workload through a context of ONE raster lane (kernels one after the other, bins for the whole batch): what the headline figure would be without the lanes' overlap"}
        out = {
            "metric": "filtered depth frames/sec (640x480, PR2 URDF)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_steps": timed_steps, "min_seconds": args.min_seconds, "timed_seconds": elapsed,
            "ms_per_step": elapsed / max(timed_steps, 1) * 1e3, "higher_is_better": True, "scaling": share.scaling, "vs_baseline": None,
            "step_ms": step_ms,
            "dtype": "f32", "data": "synthetic", "depth_format": "16UC1" if args.u16 else "32FC1",
            "config": {"workload": share.describe(), "distinct_joint_states_per_stream": V, "distinct_sensor_plane_sets": VD,
                       "streams_per_gpu": n if share.scaling == "weak" else [x["streams"] for x in per_rank], "streams_total": sum(x["streams"] for x in per_rank),
                       "poses": "host matrices" if args.host_poses else "joint positions, forward kinematics on the GPU", "mode": "two-kernel" if two else "fused", "mask_output": (not args.no_mask),
                       "parallelism": ("stream-sharded x%d" % world) + (" (shares of a %d-GPU job)" % job_world if args.shard_of else ""), "pipelines_per_gpu": P,
                       "raster_lanes": lanes, "lanes_side_by_side": bool(st["lanes_side_by_side"]), "launch_groups_per_batch": groups_per_batch, "streams_per_launch_group": st["launch_group"],
                       "host_threads_pinned_to_gpu_numa_node": pinned_cpus},
            "per_stream_fps": value / max(sum(x["streams"] for x in per_rank), 1),
            "collectives": ({"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "world": world,
                             "used_for": "barriers around the timed region, MAX all-reduce of the repetition count and the elapsed time, all-gather of per-rank frame and parity counts; no data-path collective"}
                            if dist is not None else None),
            "kernel_ms_per_step": dict(breakdown, note="stage breakdown from %d extra steps after the timed region on the %s: one batch in flight, every stage bracketed by HIP events (ms_setup there = cull + set-up + clip + many-tile kernels)" % (extra, "one-lane context" if iso_leg is not None else "headline context")),
            "rasteriser": {"triangles_per_s": float(share.triangles_per_stream().sum()) / (src_ms["setup"] * 1e-3) * (launch_streams / n) if src_ms["setup"] > 0 else None,
                           "binned_triangles_per_s": st["triangles_binned"] / (src_ms["raster"] * 1e-3) * (launch_streams / n) if src_ms["raster"] > 0 else None,
                           "triangles_submitted": int(share.triangles_per_stream().sum()), "triangles_binned": st["triangles_binned"],
                           "triangles_clipped": st["triangles_clipped"], "bin_entries": st["bin_entries"],
                           "fragments_binned": st["fragments_binned"], "max_bin_fill": st["max_bin_fill"], "max_fragment_bin_fill": st["max_fbin_fill"], "bin_capacity": st["bin_capacity"], "regrowths": st["regrowths"],
                           "setup": {"work_items": st["work_items"], "zero_survivor_items": st["zero_survivor_items"],
                                     "note": "work item = one set-up workgroup: a chunk of <= 256 triangles x up to 3 streams whose frustum its box touches; zero-survivor = none of its triangles reached a bin or the clip list's survivors (sub-pixel or outside)"},
                           "tile": {"cover_tiles": st["cover_tiles"], "occluded_entries": st["occluded_entries"], "exact_z_tiles": st["exact_tiles"],
                                    "tiles_per_launch": int(round(n / groups_per_batch)) * ((W + 63) // 64) * ((H + 31) // 32), "overdraw": overdraw}},
            "device_memory_bytes": st["device_bytes"],
            "roofline": roof,
            "parity": {"frames_checked": checked_total, "mismatching_values": bad_total, "mask_mismatch_pixels": bad_mask, "depth_mismatch_pixels": bad_depth,
                       "per_rank": per_rank, "oracle_threads": threads, "seconds": t_par,
                       "note": "every rank checks --check-frames streams (default: all) of its LAST TIMED step against the oracle, which runs on the host cores; mask/depth split is rank 0's"},
        }
        if fk_err is not None:
            out["fk"] = {"on_device": True, "max_abs_diff_vs_host_fk": fk_err}
        # the whole path against the HBM roofline: algorithmic bytes of a frame (fused: sensor read + masked write + mask write;
        # two-kernel: + the z-surface written and read) x frames/s over all GPUs / (GPUs x 8 TB/s)
        bytes_per_frame = (((4 if args.u16 else 8) + mask_b) + (8 if two else 0)) * px
        out["hbm_frac_end_to_end"] = {"value": value * bytes_per_frame / (world * peak * 1e9), "bytes_per_frame": bytes_per_frame, "peak_GB_per_s_per_gpu": peak, "n_gpus": world,
                                      "note": "value x algorithmic bytes per frame / (n_gpus x 8 TB/s): pose, cull, set-up, clip and tile kernels of a step all inside it"}
        # ---- the whole path with the planes in host memory (the reference's own timer wraps upload + render + read-back,
        # src/urdf_filter.cpp:211-244, :332-353, :729-735).  Never `value`. ----------------------------------------------
        if world == 1 and args.host_copy_seconds > 0 and P == 1 and not two and not args.no_mask:
            link_gbs = 63.0          # PCIe 5.0 x16, per direction
            hc = {"link": "PCIe 5.0 x16: %.0f GB/s per direction" % link_gbs, "modes": {}}
            words = ctx.mask_bits_words()

            def host_leg(fmt, bits):
                dt = np.uint16 if fmt == "16UC1" else np.float32
                h_in = [ctx.host_alloc((n, H, W), dt) for _ in range(2)]
                h_out = [] if bits else [ctx.host_alloc((n, H, W), dt) for _ in range(2)]
                h_mask = [] if bits else [ctx.host_alloc((n, H, W), np.uint8) for _ in range(2)]
                h_bits = [ctx.host_alloc((n, words), np.uint32) for _ in range(2)] if bits else []
                for v in range(2):
                    d = d_depth[v % VD].cpu().numpy()
                    if args.u16:
                        d = depth_u16_to_f32(d.view(np.uint16))
                    h_in[v][...] = depth_f32_to_u16(np.nan_to_num(d, nan=0.0, posinf=0.0)) if fmt == "16UC1" else d

                def go(k):
                    if bits:
                        ctx.filter_batch_bits_async(h_in[k % 2], h_bits[k % 2])
                    else:
                        ctx.filter_batch_async(h_in[k % 2], h_out[k % 2], h_mask[k % 2])

                kq = k_last + 1 + ((-(k_last + 1)) % (2 * V))      # an even step number that starts the joint states' cycle: plane set k % 2, joint states k % V
                for k in range(kq, kq + 2):
                    share.stage(ctx, k)
                    go(k)
                    ctx.sync()
                share.stage(ctx, kq + 2)
                go(kq + 2)
                ctx.sync()
                share.stage(ctx, kq + 3)
                tq = time.perf_counter()                          # (the steps before this one paid for the staging buffers)
                go(kq + 3)
                ctx.sync()
                est = time.perf_counter() - tq
                steps_h = max(4, int(np.ceil(args.host_copy_seconds / max(est, 1e-9))))
                steps_h += steps_h % 2
                ks = kq + 4
                share.stage(ctx, ks)
                th = time.perf_counter()
                for k in range(ks, ks + steps_h):
                    go(k)                                       # two batches in flight: the third call retires the first
                    share.stage(ctx, k + 1)
                ctx.sync()
                el = time.perf_counter() - th
                # parity of the last batch, first and last stream (same checker: the oracle, fed the poses the device produced)
                bad = None
                if True:
                    kl = ks + steps_h - 1
                    lk, ck = (None, None) if args.host_poses else ctx.read_poses(n, share.n_links_total)
                    bad = 0
                    for s_ in (0, n - 1):
                        d32 = depth_u16_to_f32(h_in[kl % 2][s_]) if fmt == "16UC1" else h_in[kl % 2][s_]
                        proj, draws, off, cam = share.oracle_frame(kl, s_, lk, ck)
                        om, ok = O.filter_frame(d32, proj, draws, off, cam, max_diff=wl0.max_diff, replace_value=wl0.replace_value)
                        if bits:
                            m2, k2 = R.expand_mask_bits(h_in[kl % 2][s_], h_bits[kl % 2][s_], wl0.replace_value)
                        else:
                            m2, k2 = h_out[kl % 2][s_], h_mask[kl % 2][s_]
                        want = depth_f32_to_u16(om) if fmt == "16UC1" else om
                        bad += int((ok != k2).sum()) + int((want.view(np.uint16 if fmt == "16UC1" else np.uint32) != m2.view(np.uint16 if fmt == "16UC1" else np.uint32)).sum())
                b_in = h_in[0].nbytes
                b_out = h_bits[0].nbytes if bits else h_out[0].nbytes + h_mask[0].nbytes
                up, down = b_in * steps_h / el / 1e9, b_out * steps_h / el / 1e9
                for a_ in h_in + h_out + h_mask + h_bits:
                    ctx.host_free(a_)
                return {"frames_per_s": n * steps_h / el, "ms_per_step": el / steps_h * 1e3, "steps": steps_h, "seconds": el,
                        "host_to_device_GB_per_s": up, "device_to_host_GB_per_s": down,
                        "fraction_of_link": {"host_to_device": up / link_gbs, "device_to_host": down / link_gbs},
                        "frames_checked": 2 if bad is not None else 0, "mismatching_values": bad}

            try:
                hc["modes"]["32FC1 planes in and out, two batches in flight"] = host_leg("32FC1", False)
                hc["modes"]["16UC1 planes in, mask bits out (1 bit per pixel), two batches in flight"] = host_leg("16UC1", True)
            except Exception as e:      # noqa: BLE001 - pinned memory may be short on a shared box; the headline must survive
                hc["error"] = repr(e)
            hc["note"] = ("same workload and context as `value`, but the sensor planes start in pinned HOST memory and the results end there "
                          "(rtuf_filter_batch_async / rtuf_filter_batch_bits_u16_async: upload on one copy stream, kernels, read-back on another); "
                          "PCIe-bound. Never `value`: the contract's value has its inputs resident in HBM")
            out["with_host_copies"] = hc
        # ---- the other BASELINE.json configurations, a few seconds each (never `value`) --------------------------------------
        legs_on = args.other_configs == "on" or (args.other_configs == "auto" and default_cmd and not two and P == 1 and args.lanes == 0 and args.launch_group == 0 and args.debug_flags == 0)
        if world == 1 and args.other_configs_seconds > 0 and legs_on and args.workload == "c3":
            oc = {}
            legs = (("c2_batch1", "c3", 1, False, 1), ("c3_near_arm", "c3", 1, True, n), ("c4_share", "c4", 8, False, None), ("c5_share", "c5", 8, False, None))
            for label, wname, jw, near, streams_ in legs:
                try:
                    key, rec = side_leg(label, wname, jw, near, streams_, args.other_configs_seconds, local_rank, args.triangles, reuse_planes=d_depth[0])
                    oc[key] = rec
                except Exception as e:      # noqa: BLE001 - a side leg must not cost the bench line
                    oc[label] = {"error": repr(e)}
            oc["note"] = ("BASELINE.json configs 2, 4 (rank 0's share of the 8-GPU job: 64 x 720p streams, robot + two walls), 5 (rank 0's share: 8 distinct URDFs x 128 cameras) "
                          "and config 3 with the right forearm 0.1-0.35 m in front of every lens; each timed for --other-configs-seconds after the headline's timed region "
                          "through a default context (frames/s) and a one-lane context (tile kernel alone), 8 streams of the last step against the oracle")
            out["other_configs"] = oc
        # ---- CPU baseline: the oracle port on this box's host cores, on a bounded sample of the same batch ----
        if args.cpu_seconds > 0 and n > 0:             # (rank 0 only: we are inside its branch)
            cores = host_threads
            # inputs: the first streams of the last batch (exactly what the GPU just filtered), prepared once
            n_in = min(n, 64)
            prepared = []
            for s in range(n_in):
                hd, _, _ = fetch(s)
                prepared.append(O.PreparedFrame(hd, *share.oracle_frame(k_last, s, link_dev, cam_dev), max_diff=wl0.max_diff, replace_value=wl0.replace_value))
            c0 = time.perf_counter()
            n1 = 0
            while time.perf_counter() - c0 < args.cpu_seconds:
                prepared[n1 % n_in].run()
                n1 += 1
            t1 = time.perf_counter() - c0
            # all cores: POSIX threads inside the oracle library for --cpu-seconds (a shared counter hands out frames; no
            # Python in the loop; per-thread scratch memory)
            # as many threads as cores this process may really use at once: min(visible hardware threads, cgroup CPU quota) --
            # 256 threads under a 16-core quota only measure the scheduler
            nN, tN = O.filter_throughput(prepared, args.cpu_seconds, threads)
            cb = {"value": n1 / t1, "unit": "frames/s", "cores": 1, "kind": "port",
                  "sample": "%d frames of the last batch (first %d streams, cycled) through oracle/rtuf_oracle.c, single thread, %.1f s" % (n1, n_in, t1),
                  "all_cores": {"value": nN / tN, "unit": "frames/s", "cores": threads, "hardware_threads_visible": cores, "cgroup_cpu_quota_cores": quota,
                                "sample": "%d frames of the same set on %d POSIX threads (= min(hardware threads, cgroup quota)) inside the oracle library, %.1f s" % (nN, threads, tN)}}
            try:
                lp = json.load(open(os.path.join(ROOT, "profiles", "llvmpipe_baseline.json")))
                cb["reference_llvmpipe"] = dict(lp.get("bench_workload", {}), source="OFFLINE: profiles/llvmpipe_baseline.json -- the reference's own GLSL on Mesa llvmpipe, timed in the development container (scripts/llvmpipe_baseline.py): the harness reads the reference's shaders from /root/reference at run time, which does not exist on the GPU box")
            except Exception:
                pass
            # the reference's CPU path itself, live on this box: its GL call sequence on Mesa llvmpipe through the test
            # harness (oracle/_ref), with the repo-authored stand-in shaders (the reference's files do not exist here;
            # the stand-ins are checked bit-for-bit against them in the development container)
            try:
                from oracle.ref_gl import harness as HN
                if args.cpu_seconds >= 2 and HN.available("standin"):
                    import subprocess
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "llvmpipe_baseline.py"), "--bench-leg", "--shaders", "standin",
                                        "--seconds", str(min(args.cpu_seconds, 6.0)), "--frames", "8"], capture_output=True, text=True, cwd=ROOT, timeout=240)
                    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                    if r.returncode == 0 and line:
                        cb["reference_llvmpipe_live"] = dict(json.loads(line[-1]), unit="frames/s",
                                                             note="the reference's per-frame GL sequence (upload, render from static VBOs, two read-backs) on Mesa llvmpipe on THIS box, one stream per frame of the bench model; stand-in shaders = bit-identical re-statement of include/shaders/urdf_filter.{vert,frag} (tests/test_oracle_vs_llvmpipe.py)")
                    else:
                        cb["reference_llvmpipe_live"] = {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:      # noqa: BLE001 - a missing Mesa must not cost the bench line
                cb["reference_llvmpipe_live"] = {"error": repr(e)}
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist is not None:
        barrier()               # (ranks > 0 wait here while rank 0 runs its roofline leg and the CPU baseline)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
