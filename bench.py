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
are whole multiples of `--steps`; `timed_steps` in the output says how many steps were timed in total).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--min-seconds", type=float, default=0.5, help="the timed region repeats the --steps steps until it is at least this long")
    ap.add_argument("--workload", choices=["c3", "c4", "c5"], default="c3", help="BASELINE.json config: c3 (headline, weak scaling), c4, c5 (fixed totals, sharded)")
    ap.add_argument("--streams", type=int, default=None, help="c3: concurrent camera streams per GPU (256); c4: streams in total (512); c5: cameras per URDF (128)")
    ap.add_argument("--urdfs", type=int, default=64, help="c5: distinct URDFs in total")
    ap.add_argument("--shard-of", type=int, default=0, help="take the shares of a job of this many GPUs (ranks 0..N-1 of it) instead of a job of --gpus GPUs")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--triangles", type=int, default=250000, help="triangle budget of the PR2-like model")
    ap.add_argument("--variants", type=int, default=2, help="distinct input batches rotated through the steps")
    ap.add_argument("--pipelines", type=int, default=1, help="contexts (HIP stream + bins each) per GPU that the batches alternate between: with 2 or 3, one batch's small and low-occupancy kernels overlap another's heavy ones, but kernels then share the GPU and per-launch times (roofline) no longer describe one kernel; default 1")
    ap.add_argument("--launch-group", type=int, default=0, help="rtuf_params.max_inflight_streams: streams rasterised per internal launch group (0 = automatic: the whole batch up to 1024); smaller groups shrink the tile bins and cost a kernel sequence per group")
    ap.add_argument("--overlap-pipelines", type=int, default=2, help="after the main measurement (one pipeline, clean per-kernel roofline) time the same steps once more on a second context with rtuf_params.pipelines = this, reported as `overlapped` (N=1 only; 0 disables)")
    ap.add_argument("--two-kernel", action="store_true", help="rasteriser + separate compare kernel")
    ap.add_argument("--host-poses", action="store_true", help="stage explicit link matrices from the host instead of joint positions + on-device forward kinematics")
    ap.add_argument("--u16", action="store_true", help="16UC1 depth in/out (uint16 millimetres) with the conversions fused into the kernels")
    ap.add_argument("--no-mask", action="store_true", help="need_mask_ == false: no mask output")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget per leg (single thread, all cores); 0 disables")
    ap.add_argument("--bin-capacity", type=int, default=0, help="rtuf_params.bin_capacity (records per tile bin; 0 = the library's default, grown on overflow)")
    ap.add_argument("--debug-flags", type=lambda x: int(x, 0), default=0, help="timing experiments only (needs the RTUF_ABLATE build; results are wrong)")
    ap.add_argument("--near-arm", action="store_true", help="c3 / c4: every stream poses the robot's right forearm 0.1-0.35 m in front of the lens (exact-z pass, near-plane clipping, whole-tile occluders)")
    ap.add_argument("--check-frames", type=int, default=16, help="frames of the last step verified against the oracle (per rank)")
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
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else "cpu"       # where the (few-byte) collectives' tensors live

    # --- workload: this rank's share ------------------------------------------------------------
    job_world = args.shard_of or world
    if rank >= job_world:
        raise SystemExit("--shard-of %d: rank %d has no share" % (job_world, rank))
    share = CF.build(args.workload, job_world, rank, streams=args.streams, triangles=args.triangles, variants=args.variants,
                     width=args.width, height=args.height, urdfs=args.urdfs, near_arm=args.near_arm)
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
    P = max(1, args.pipelines)
    p.pipelines = P if P > 1 else 0          # rtuf_params.pipelines: the library alternates the batches between P internal pipelines
    ctx = R.Context(W, H, n, local_rank, p)
    share.load(ctx, on_device_fk=not args.host_poses)
    ctx.enable_timing(2)         # HIP events around the big kernels only (each event costs stream time)
    ctxs = [ctx]
    V = share.n_variants()

    d_depth = []
    for v in range(V):
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
        (ctx.filter_batch_device_u16 if args.u16 else ctx.filter_batch_device)(n, dptr[k % V], ptrs[k % n_sets][0], ptrs[k % n_sets][1])

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
    t0 = time.perf_counter()
    for k in range(k0, k0 + timed_steps):
        enqueue(k)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    sts = [c.stats() for c in ctxs]
    timed = sum(st["timed_batches"] for st in sts)
    assert timed >= 1 and timed >= timed_steps // 8, ([st["timed_batches"] for st in sts], timed_steps)
    setup_ms = sum(st["sum_ms_setup"] for st in sts) / timed          # set-up kernel alone
    clip_ms = sum(st["sum_ms_clip"] for st in sts) / timed
    raster_ms = sum(st["sum_ms_raster"] for st in sts) / timed
    compare_ms = sum(st["sum_ms_compare"] for st in sts) / timed
    # stage-by-stage breakdown: a few extra steps on one pipeline, one batch in flight, every stage bracketed
    # by events, outside the timed region (kernel times without another batch sharing the GPU)
    ctx.enable_timing(1)
    extra = 3 * V       # a multiple of the variant cycle
    acc = {"ms_pose": 0.0, "ms_setup": 0.0, "ms_raster": 0.0, "ms_compare": 0.0, "ms_total": 0.0}
    k_after = k0 + timed_steps
    k_after += (-k_after) % V          # so that the last extra step is variant V-1 and k_last below is well defined
    for j in range(extra):
        isolated_step(k_after + j)
        st = ctx.stats()
        for key in acc:
            acc[key] += st[key]
    breakdown = {key: v / extra for key, v in acc.items()}
    # ... and the raster stage's kernels one by one under the same conditions (timing mode 2)
    ctx.enable_timing(2)
    iso = {"ms_setup": 0.0, "ms_clip": 0.0, "ms_raster": 0.0, "ms_compare": 0.0}
    for j in range(extra):
        isolated_step(k_after + extra + j)
        st = ctx.stats()
        for key in iso:
            iso[key] += st[key] / extra
    k_last = k_after + 2 * extra - 1
    groups_per_batch = 1
    frames_rank = n * timed_steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()

    # ---- parity spot check on every rank (oracle = checker only) -----------------------------------
    from oracle import bindings as O
    from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32
    v_last = k_last % V
    d_masked, d_mask = d_masked_set[k_last % n_sets], d_mask_set[k_last % n_sets]
    link_dev = cam_dev = None
    fk_err = None
    if not args.host_poses:
        # the oracle is fed the very matrices the GPU's forward kinematics produced
        link_dev, cam_dev = ctx.read_poses(n, share.n_links_total)
        fk_err = share.host_fk_error(k_last, link_dev, cam_dev)
    check = sorted(set(int(x) for x in np.linspace(0, n - 1, num=min(max(args.check_frames, 0), n)))) if args.check_frames > 0 else []

    def fetch(s):
        hm = d_masked[s].cpu().numpy()
        hk = d_mask[s].cpu().numpy() if d_mask is not None else None
        hd = d_depth[v_last][s].cpu().numpy()
        if args.u16:
            hm = hm.view(np.uint16)
            hd = depth_u16_to_f32(hd.view(np.uint16))
        return hd, hm, hk

    def oracle(s, hd):
        proj, draws, off, cam = share.oracle_frame(k_last, s, link_dev, cam_dev)
        return O.filter_frame(hd, proj, draws, off, cam, max_diff=wl0.max_diff, replace_value=wl0.replace_value)

    bad_mask = bad_depth = 0
    for s in check:
        hd, hm, hk = fetch(s)
        om, ok = oracle(s, hd)
        if hk is not None:
            bad_mask += int((ok != hk).sum())
        bad_depth += int((depth_f32_to_u16(om) != hm).sum()) if args.u16 else int((om.view(np.uint32) != hm.view(np.uint32)).sum())
    frames_total, bad_total, checked_total = frames_rank, bad_mask + bad_depth, len(check)
    per_rank = [{"rank": rank, "streams": n, "frames": frames_rank, "frames_checked": len(check), "mismatching_values": bad_mask + bad_depth}]
    if dist is not None:
        # the trivial end-of-run gather (a few numbers per rank): frames, parity counts
        frames_total, elapsed = sharding.gather_frame_counts(dist, frames_rank, elapsed, device=cdev)
        t = torch.tensor([n, frames_rank, len(check), bad_mask + bad_depth], dtype=torch.int64, device=cdev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        per_rank = [{"rank": r, "streams": int(o[0]), "frames": int(o[1]), "frames_checked": int(o[2]), "mismatching_values": int(o[3])} for r, o in enumerate(outl)]
        bad_total = sum(x["mismatching_values"] for x in per_rank)
        checked_total = sum(x["frames_checked"] for x in per_rank)

    if rank == 0:
        value = frames_total / elapsed
        per = dict(breakdown)
        px = W * H
        two = args.two_kernel
        # algorithmic bytes per launch:
        #   fused tile kernel : 9 B/pixel = 4 sensor read + 4 masked write + 1 mask write (8 without mask; 5 / 4 for 16UC1)
        #   two-kernel mode   : tile kernel writes the 4 B/pixel z-surface; compare moves 13 B/pixel
        #   set-up + clip     : every vertex (12 B) and triangle (12 B indices + 4 B order) of the model once per launch --
        #                       the geometry is shared by all streams; the records it writes are implementation traffic
        mask_b = 0 if args.no_mask else 1
        kernels = {}
        if two:
            kernels["tile_kernel<two_kernel>"] = (raster_ms, iso["ms_raster"], 4 * px * n, "valu-issue / LDS atomics (rasteriser: neither HBM nor MFMA, SURVEY.md 8d); HBM figure for context")
            kernels["compare_kernel"] = (compare_ms, iso["ms_compare"], ((6 if args.u16 else 12) + mask_b) * px * n, "hbm")
        else:
            kernels["tile_kernel<fused>"] = (raster_ms, iso["ms_raster"], ((4 if args.u16 else 8) + mask_b) * px * n, "hbm")
        geo_bytes = sum(12 * g.variants[0].n_vertices() + 16 * g.variants[0].n_triangles() for g in share.groups)
        kernels["setup_kernel"] = (setup_ms, iso["ms_setup"], geo_bytes, "valu-issue (triangle set-up: neither HBM nor MFMA, SURVEY.md 8d); HBM figure for context")
        kernels["clip_kernel"] = (clip_ms, iso["ms_clip"], 0, "latency / divergent scalar code at LDS-limited occupancy; no algorithmic HBM traffic of its own (the time covers clip_kernel and bigrec_kernel, which appends the many-tile records both set-up and clip kernel listed)")
        peak = 8000.0
        # Off-line counter data of this same command (rocprofv3 --pmc passes cannot run inside the timed region):
        # used only when the committed measurement is of this exact workload, and labelled as such.
        default_cmd = (args.workload == "c3" and n == 256 and (W, H) == (640, 480) and not args.no_mask and not args.u16 and args.triangles == 250000 and not args.near_arm and not args.host_poses)
        # (config 4's per-GPU share -- the other workload whose counters are committed, profiles/r03_pmc_workload_c4_shard_of_8.txt)
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

        def kernel_entry(name):
            dur_ms, iso_ms, alg_bytes, bound = kernels[name]
            achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
            e = {"kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                 "traffic": None, "avg_launch_ms": dur_ms, "algorithmic_bytes_per_launch": alg_bytes,
                 "isolated": {"avg_launch_ms": iso_ms, "frac": (alg_bytes / (iso_ms * 1e-3) / 1e9 / peak) if iso_ms > 0 else None,
                              "note": "same kernel(s) with nothing else on the GPU (extra steps after the timed region)"}}
            if name == "compare_kernel" and 4 * px * n < 2 * (256 << 20):
                e["note"] = "HBM + MALL figure: the %d MB z-surface this kernel reads was written by the kernel before it and partly sits in the 256 MiB Infinity Cache; with --streams 1024 (z-surface 1.26 GB) the same kernel measures pure HBM" % (4 * px * n // 1000000)
            rec = (pmc or {}).get("kernels", {}).get(name) if (pmc and default_cmd and P == 1) else None
            if rec is None and pmc and c4_share_cmd and P == 1:
                rec = (pmc.get("c4_share") or {}).get("kernels", {}).get(name)
            if rec:
                e["traffic"] = rec.get("hbm_bytes_per_launch")
                e["traffic_source"] = "OFFLINE: " + (pmc.get("c4_share", {}).get("source") if (c4_share_cmd and not default_cmd) else pmc.get("source", "profiles/pmc_counters.json")) + " (rocprofv3 --pmc passes of this same command; not measured in this run)"
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
        if dom["bound"] != "hbm":
            dom = dict(dom, bound_note=dom["bound"], bound="hbm")   # (the contract's vocabulary; the note says what really limits it)
        roof = dict(dom)
        roof.update({"launches_per_step": groups_per_batch, "timed_launches": timed,
                     "all_kernels": [e for e in entries if e["kernel"] != dom["kernel"]]})
        out = {
            "metric": "filtered depth frames/sec (640x480, PR2 URDF)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_steps": timed_steps, "min_seconds": args.min_seconds,
            "ms_per_step": elapsed / max(timed_steps, 1) * 1e3, "higher_is_better": True, "scaling": share.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "depth_format": "16UC1" if args.u16 else "32FC1",
            "config": {"workload": share.describe(),
                       "streams_per_gpu": n if share.scaling == "weak" else [x["streams"] for x in per_rank], "streams_total": sum(x["streams"] for x in per_rank),
                       "poses": "host matrices" if args.host_poses else "joint positions, forward kinematics on the GPU", "mode": "two-kernel" if two else "fused", "mask_output": (not args.no_mask),
                       "parallelism": ("stream-sharded x%d" % world) + (" (shares of a %d-GPU job)" % job_world if args.shard_of else ""), "pipelines_per_gpu": P,
                       "host_threads_pinned_to_gpu_numa_node": pinned_cpus},
            "per_stream_fps": value / max(sum(x["streams"] for x in per_rank), 1),
            "collectives": ({"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "world": world,
                             "used_for": "barriers around the timed region, MAX all-reduce of the repetition count and the elapsed time, all-gather of per-rank frame and parity counts; no data-path collective"}
                            if dist is not None else None),
            "kernel_ms_per_step": dict(per, note="stage breakdown from %d extra steps after the timed region: one batch in flight, every stage bracketed by HIP events (ms_setup there = cull + set-up + clip + waiting for the pose stage); roofline.avg_launch_ms is measured inside the timed region" % extra),
            "rasteriser": {"triangles_per_s": float(share.triangles_per_stream().sum()) / (setup_ms * 1e-3) if setup_ms > 0 else None,
                           "binned_triangles_per_s": st["triangles_binned"] / (raster_ms * 1e-3) if raster_ms > 0 else None,
                           "triangles_submitted": int(share.triangles_per_stream().sum()), "triangles_binned": st["triangles_binned"],
                           "triangles_clipped": st["triangles_clipped"], "bin_entries": st["bin_entries"],
                           "fragments_binned": st["fragments_binned"], "max_bin_fill": st["max_bin_fill"], "max_fragment_bin_fill": st["max_fbin_fill"], "bin_capacity": st["bin_capacity"], "regrowths": st["regrowths"],
                           "setup": {"work_items": st["work_items"], "zero_survivor_items": st["zero_survivor_items"],
                                     "note": "work item = one set-up workgroup: a chunk of <= 256 triangles x up to 3 streams whose frustum its box touches; zero-survivor = none of its triangles reached a bin or the clip list's survivors (sub-pixel or outside)"},
                           "tile": {"cover_tiles": st["cover_tiles"], "occluded_entries": st["occluded_entries"], "exact_z_tiles": st["exact_tiles"],
                                    "tiles_per_launch": n * ((W + 63) // 64) * ((H + 31) // 32), "overdraw": overdraw}},
            "device_memory_bytes": st["device_bytes"],
            "roofline": roof,
            "parity": {"frames_checked": checked_total, "mismatching_values": bad_total, "mask_mismatch_pixels": bad_mask, "depth_mismatch_pixels": bad_depth,
                       "per_rank": per_rank, "note": "every rank checks --check-frames frames of its last step against the oracle; mask/depth split is rank 0's"},
        }
        if fk_err is not None:
            out["fk"] = {"on_device": True, "max_abs_diff_vs_host_fk": fk_err}
        # ---- the same steps with the library's internal pipelines overlapping (rtuf_params.pipelines) ----------
        if world == 1 and P == 1 and args.overlap_pipelines > 1 and not args.debug_flags:
            PO = args.overlap_pipelines
            p2 = R.default_params()
            p2.filter_replace_value, p2.depth_distance_threshold, p2.flags, p2.pipelines = p.filter_replace_value, p.depth_distance_threshold, p.flags, PO
            ctx.sync()
            ctx2 = R.Context(W, H, n, local_rank, p2)
            share2 = CF.build(args.workload, job_world, rank, streams=args.streams, triangles=args.triangles, variants=args.variants,
                              width=args.width, height=args.height, urdfs=args.urdfs, near_arm=args.near_arm)
            share2.load(ctx2, on_device_fk=not args.host_poses)
            sets2 = [(torch.empty_like(d_masked_set[0]), None if args.no_mask else torch.empty_like(d_mask_set[0])) for _ in range(2 * PO)]

            def submit2(k):
                m, kk = sets2[k % len(sets2)]
                (ctx2.filter_batch_device_u16 if args.u16 else ctx2.filter_batch_device)(n, dptr[k % V], m.data_ptr(), kk.data_ptr() if kk is not None else 0)

            for k in range(max(args.warmup, 2) * PO):
                share2.stage(ctx2, k)
                submit2(k)
                ctx2.sync()
            kb = args.warmup * PO + ((-args.warmup * PO) % V)
            share2.stage(ctx2, kb)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for k in range(kb, kb + timed_steps):
                submit2(k)
                share2.stage(ctx2, k + 1)
            ctx2.sync()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t2
            # parity of its last batch (same checker)
            k2_last = kb + timed_steps - 1
            m2, kk2 = sets2[k2_last % len(sets2)]
            l2, c2 = (None, None) if args.host_poses else ctx2.read_poses(n, share2.n_links_total)
            bad2 = 0
            for s_ in check[:2]:
                hd2 = d_depth[k2_last % V][s_].cpu().numpy()
                hm2 = m2[s_].cpu().numpy()
                if args.u16:
                    hm2 = hm2.view(np.uint16)
                    hd2 = depth_u16_to_f32(hd2.view(np.uint16))
                proj, draws, off, cam = share2.oracle_frame(k2_last, s_, l2, c2)
                om, ok = O.filter_frame(hd2, proj, draws, off, cam, max_diff=wl0.max_diff, replace_value=wl0.replace_value)
                if kk2 is not None:
                    bad2 += int((ok != kk2[s_].cpu().numpy()).sum())
                bad2 += int((depth_f32_to_u16(om) != hm2).sum()) if args.u16 else int((om.view(np.uint32) != hm2.view(np.uint32)).sum())
            out["overlapped"] = {"pipelines": PO, "value": n * timed_steps / el2, "unit": "frames/s", "ms_per_step": el2 / timed_steps * 1e3,
                                 "timed_steps": timed_steps, "frames_checked": len(check[:2]), "mismatching_values": bad2,
                                 "note": "same workload and step count on a context with rtuf_params.pipelines = %d: batches alternate between %d internal pipelines, so kernels of different batches share the GPU (higher throughput; per-launch kernel times, and with them a per-kernel roofline, no longer describe one kernel -- hence not the headline)" % (PO, PO)}
            ctx2.close()
        # ---- CPU baseline: the oracle port on this box's host cores, on a bounded sample of the same batch ----
        if world == 1 and args.cpu_seconds > 0 and n > 0:
            cores = len(os.sched_getaffinity(0))
            quota = None
            try:        # a container's CPU quota (cgroup v2): the cores the threads below can really use at once
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            except Exception:
                quota = None
            # inputs: the first streams of the last batch (exactly what the GPU just filtered), prepared once
            n_in = min(n, 64)
            inputs = []
            for s in range(n_in):
                hd, _, _ = fetch(s)
                inputs.append((hd,) + tuple(share.oracle_frame(k_last, s, link_dev, cam_dev)))
            prepared = [O.PreparedFrame(*inputs[i % n_in], max_diff=wl0.max_diff, replace_value=wl0.replace_value) for i in range(n_in)]
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
            threads = max(1, min(cores, int(quota))) if quota else cores
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
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
