#!/usr/bin/env python3
"""bench.py -- filtered depth frames/s of the hot path (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: `--streams` concurrent
640x480 camera streams of the synthetic PR2-like robot (config C3 of SURVEY.md section 8d:
"640x480, PR2 URDF, batch=256 concurrent camera streams on 1 MI355X", the configuration the
BASELINE target ">=30 frames/s per stream at >=256 streams" is quoted on).  Every step stages a
fresh joint state + camera pose for every stream, then runs pose -> set-up/binning -> clip ->
tile raster + per-pixel compare on depth frames that are already resident in HBM.  Inputs rotate
through `--variants` distinct pre-generated batches so no step can reuse the previous result.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): streams are independent, so every rank
runs the same per-GPU batch on its own streams (weak scaling, no data-path collective); RCCL is
used only for the barrier and the max-over-ranks time reduction.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=256, help="concurrent camera streams per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--triangles", type=int, default=250000, help="triangle budget of the PR2-like model")
    ap.add_argument("--variants", type=int, default=2, help="distinct input batches rotated through the steps")
    ap.add_argument("--pipelines", type=int, default=1, help="contexts (HIP stream + bins each) per GPU that the batches alternate between: with 2 or 3, one batch's small and low-occupancy kernels overlap another's heavy ones (+8..12 %% frames/s), but kernels then share the GPU and per-launch times (roofline) no longer describe one kernel; default 1")
    ap.add_argument("--two-kernel", action="store_true", help="rasteriser + separate compare kernel")
    ap.add_argument("--host-poses", action="store_true", help="stage explicit link matrices from the host instead of joint positions + on-device forward kinematics")
    ap.add_argument("--u16", action="store_true", help="16UC1 depth in/out (uint16 millimetres) with the conversions fused into the kernels")
    ap.add_argument("--no-mask", action="store_true", help="need_mask_ == false: no mask output")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--debug-flags", type=lambda x: int(x, 0), default=0, help="timing experiments only (results are wrong)")
    ap.add_argument("--check-frames", type=int, default=4, help="frames of the last step verified against the oracle")
    args = ap.parse_args()

    import torch
    import realtime_urdf_filter_amd as R
    from realtime_urdf_filter_amd import workloads as WL

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
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)

    n, W, H = args.streams, args.width, args.height
    # --- workload: every rank owns `n` different streams (seeds offset by rank) --------------
    variants = []
    for v in range(args.variants):
        wl = WL.pr2_workload(n, W, H, args.triangles, first_state_seed=1000 + 100000 * v + 1000003 * rank)
        variants.append(wl)
    wl0 = variants[0]
    p = R.default_params()
    p.filter_replace_value = wl0.replace_value
    p.depth_distance_threshold = wl0.max_diff
    if args.two_kernel:
        p.flags |= R.FLAG_TWO_KERNEL
    p.flags |= args.debug_flags
    P = max(1, args.pipelines)
    ctxs, idss = [], []
    for _ in range(P):
        c = R.Context(W, H, n, local_rank, p)
        i = wl0.load_into(c)
        if not args.host_poses:
            wl0.load_kinematics(c, i)
        c.enable_timing(2)       # HIP events around the dominant kernel only (each event costs stream time)
        ctxs.append(c)
        idss.append(i)
    ctx, ids = ctxs[0], idss[0]

    d_depth = []
    for v, wl in enumerate(variants):
        host = np.stack([wl.depth(s + 7 * v + 1000 * rank) for s in range(n)])
        if args.u16:
            host = np.clip(np.rint(np.nan_to_num(host, nan=0.0, posinf=0.0) * 1000.0), 0, 65535).astype(np.uint16).view(np.int16)
        d_depth.append(torch.from_numpy(host).to(dev))
    # two output sets per pipeline: with two batches in flight per context, a batch must not write where an
    # earlier batch's results are still unread
    n_sets = 2 * P
    d_masked_set = [torch.empty((n, H, W), dtype=torch.int16 if args.u16 else torch.float32, device=dev) for _ in range(n_sets)]
    d_mask_set = [None if args.no_mask else torch.empty((n, H, W), dtype=torch.uint8, device=dev) for _ in range(n_sets)]
    torch.cuda.synchronize()
    V = len(variants)
    staged_once = [False] * P
    ptrs = [(d_masked_set[i].data_ptr(), d_mask_set[i].data_ptr() if d_mask_set[i] is not None else 0) for i in range(n_sets)]
    dptr = [d.data_ptr() for d in d_depth]

    def stage_into(ci, k):
        if args.host_poses:
            variants[k % V].stage(ctxs[ci], idss[ci])
        else:   # joint angles in, forward kinematics on the GPU
            variants[k % V].stage_joint_positions(ctxs[ci], idss[ci], first_call=not staged_once[ci])
        staged_once[ci] = True

    def submit(ci, k):
        c = ctxs[ci]
        (c.filter_batch_device_u16 if args.u16 else c.filter_batch_device)(n, dptr[k % V], ptrs[k % n_sets][0], ptrs[k % n_sets][1])

    def enqueue(k):
        # one step = one batch through the hot path: enqueue it on pipeline k mod P, then stage the NEXT batch's
        # joint positions (host memory only) while the GPU works.  Each context keeps up to two batches in
        # flight and retires its oldest when a third arrives; nothing in the loop waits for the GPU otherwise.
        submit(k % P, k)
        stage_into((k + 1) % P, k + 1)

    def isolated_step(k):
        stage_into(0, k)
        submit(0, k)
        ctx.sync()

    def barrier():
        if dist is not None:
            dist.barrier()

    k0 = args.warmup * P
    for k in range(k0):            # every pipeline warms up (first batch: bin sizing) with one batch in flight
        stage_into(k % P, k)
        submit(k % P, k)
        ctxs[k % P].sync()
    torch.cuda.synchronize()
    barrier()
    for c in ctxs:
        c.enable_timing(3)       # (re)starts the library's event sums: tile (and compare) kernel, every fourth batch (an event costs ~5 us of stream time)
    stage_into(k0 % P, k0)
    t0 = time.perf_counter()
    for k in range(k0, k0 + args.steps):
        enqueue(k)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    sts = [c.stats() for c in ctxs]
    timed = sum(st["timed_batches"] for st in sts)
    assert timed >= 1 and timed >= args.steps // 4, ([st["timed_batches"] for st in sts], args.steps)
    raster_ms = sum(st["sum_ms_raster"] for st in sts) / timed
    compare_ms = sum(st["sum_ms_compare"] for st in sts) / timed
    # stage-by-stage breakdown: a few extra steps on one pipeline, one batch in flight, every stage bracketed
    # by events, outside the timed region (kernel times without another batch sharing the GPU)
    ctx.enable_timing(1)
    extra = 3 * V       # a multiple of the variant cycle: the last step run is the last timed step's variant (parity below)
    acc = {"ms_pose": 0.0, "ms_setup": 0.0, "ms_raster": 0.0, "ms_compare": 0.0, "ms_total": 0.0}
    for j in range(extra):
        isolated_step(k0 + args.steps + j)
        st = ctx.stats()
        for key in acc:
            acc[key] += st[key]
    breakdown = {key: v / extra for key, v in acc.items()}
    k_last = k0 + args.steps + extra - 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()

    if rank == 0:
        K = max(args.steps, 1)
        frames = n * world * args.steps
        value = frames / elapsed
        per = dict(breakdown)
        per_timed = {"ms_raster": raster_ms, "ms_compare": compare_ms}
        px = W * H
        two = args.two_kernel
        # dominant kernel and its algorithmic bytes per launch:
        #   fused tile kernel : 9 B/pixel = 4 sensor read + 4 masked write + 1 mask write (8 without mask)
        #   two-kernel mode   : tile kernel writes the 4 B/pixel z-surface; compare moves 13 B/pixel
        groups = 1
        if two:
            cands = {"tile_kernel<two_kernel>": (per_timed["ms_raster"], 4 * px * n), "compare_kernel": (per_timed["ms_compare"], (13 if (not args.no_mask) else 12) * px * n)}
        else:
            bpp = (4 if args.u16 else 8) + (1 if (not args.no_mask) else 0)
            cands = {"tile_kernel<fused>": (per_timed["ms_raster"], bpp * px * n)}
        cands["setup_kernel+clip_kernel"] = (per["ms_setup"], 12 * wl0.n_vertices() + 16 * wl0.n_triangles())
        dom = max((k for k in cands if not k.startswith("setup")), key=lambda k: cands[k][0])
        dur_ms, alg_bytes = cands[dom]
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        peak = 8000.0
        # HBM bytes per launch from PMC counters are collected off-line (rocprofv3 cannot run inside the
        # timed region): profiles/hbm_traffic.json holds the committed measurement of this same command
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            if (tj["kernel"] == dom and tj["streams"] == n and tj["width"] == W and tj["height"] == H
                    and tj["triangles"] == wl0.meta["triangles"] and (not args.no_mask) and not args.u16):
                traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        # the kernels are VALU-issue-bound: wave64 VALU instructions per launch from the
        # committed SQ counters of this same command, against the machine's issue peak
        valu = None
        try:
            vj = json.load(open(os.path.join(ROOT, "profiles", "valu_counts.json")))
            cnt = vj["wave64_valu_instructions_per_launch"].get(dom)
            if (cnt and vj["streams"] == n and vj["width"] == W and vj["height"] == H and vj["triangles"] == wl0.meta["triangles"]
                    and (not args.no_mask) and not args.u16 and dur_ms > 0):
                g = cnt / (dur_ms * 1e-3) / 1e9
                valu = {"wave64_instructions_per_launch": cnt, "achieved_G_per_s": g, "peak_G_per_s": vj["peak_G_per_s"], "frac": g / vj["peak_G_per_s"],
                        "note": "instruction count from profiles/valu_counts.json (rocprofv3 SQ_INSTS_VALU), duration live; peak = " + vj["peak_note"]}
        except Exception:
            valu = None
        out = {
            "metric": "filtered depth frames/sec (640x480, PR2 URDF)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "depth_format": "16UC1" if args.u16 else "32FC1",
            "config": {"workload": "C3: %dx%d depth, synthetic PR2-like URDF (%d links with meshes, %d triangles), batch=%d concurrent streams per GPU, new joint state + camera pose every step"
                                   % (W, H, wl0.meta["links_with_geometry"], wl0.meta["triangles"], n),
                       "streams_per_gpu": n, "poses": "host matrices" if args.host_poses else "joint positions, forward kinematics on the GPU", "mode": "two-kernel" if two else "fused", "mask_output": (not args.no_mask),
                       "parallelism": "stream-sharded x%d" % world, "pipelines_per_gpu": P},
            "per_stream_fps": value / (n * world),
            "kernel_ms_per_step": dict(per, note="stage breakdown from %d extra steps after the timed region: one pipeline, one batch in flight, every stage bracketed by HIP events (kernel times without another batch sharing the GPU); roofline.avg_launch_ms is measured inside the timed region, where %d pipelines overlap" % (extra, P)),
            "rasteriser": {"triangles_per_s": wl0.n_triangles() * n / (per["ms_setup"] * 1e-3) if per["ms_setup"] > 0 else None,
                           "binned_triangles_per_s": st["triangles_binned"] / (per["ms_raster"] * 1e-3) if per["ms_raster"] > 0 else None,
                           "triangles_submitted": st["triangles_submitted"], "triangles_binned": st["triangles_binned"],
                           "triangles_clipped": st["triangles_clipped"], "bin_entries": st["bin_entries"],
                           "fragments_binned": st["fragments_binned"], "max_bin_fill": st["max_bin_fill"], "max_fragment_bin_fill": st["max_fbin_fill"], "bin_capacity": st["bin_capacity"], "regrowths": st["regrowths"]},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                         "frac": achieved / peak, "traffic": traffic, "launches_per_step": groups,
                         "avg_launch_ms": dur_ms, "timed_launches": timed, "algorithmic_bytes_per_launch": alg_bytes,
                         "valu_issue": valu,
                         "isolated": {"avg_launch_ms": per["ms_compare"] if dom == "compare_kernel" else per["ms_raster"],
                                      "frac": (alg_bytes / ((per["ms_compare"] if dom == "compare_kernel" else per["ms_raster"]) * 1e-3) / 1e9 / peak) if per["ms_raster"] > 0 else None,
                                      "note": "same kernel with nothing else on the GPU (extra steps after the timed region)"}},
        }
        # ---- parity spot check + CPU baseline (oracle = checker / reported baseline only) ------
        if world == 1:
            from oracle import bindings as O
            from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32
            v_last = k_last % V
            wl = variants[v_last]
            d_masked, d_mask = d_masked_set[k_last % n_sets], d_mask_set[k_last % n_sets]
            hm = d_masked.cpu().numpy()
            hk = d_mask.cpu().numpy() if (not args.no_mask) else None
            hd = d_depth[v_last].cpu().numpy()
            if args.u16:
                hm = hm.view(np.uint16)
                hd = depth_u16_to_f32(hd.view(np.uint16))
            if args.host_poses:
                link_tf_all, cam_all = wl.link_tf[0], wl.cam_tf
            else:
                # the oracle is fed the very matrices the GPU's forward kinematics produced
                link_tf_all, cam_all = ctx.read_poses(n, wl.link_tf[0].shape[1])
                out["fk"] = {"on_device": True, "max_abs_diff_vs_host_fk": float(max(np.abs(link_tf_all - wl.link_tf[0]).max(), np.abs(cam_all - wl.cam_tf).max()))}
            bad_mask = bad_depth = 0
            t_cpu = 0.0
            n_cpu = 0
            budget = args.cpu_seconds
            s = 0
            while s < n and (s < args.check_frames or (budget > 0 and t_cpu < budget)):
                c0 = time.perf_counter()
                draws = [(link_tf_all[s, li], d.pre_op, d.op, d.verts, d.tris) for li, dl in enumerate(wl.models[0]) for d in dl]
                om, ok = O.filter_frame(hd[s], wl.projection[s], draws, wl.offset_inv[s], cam_all[s],
                                        max_diff=wl.max_diff, replace_value=wl.replace_value)
                t_cpu += time.perf_counter() - c0
                n_cpu += 1
                if hk is not None:
                    bad_mask += int((ok != hk[s]).sum())
                bad_depth += int((depth_f32_to_u16(om) != hm[s]).sum()) if args.u16 else int((om.view(np.uint32) != hm[s].view(np.uint32)).sum())
                s += 1
            out["parity"] = {"frames_checked": n_cpu, "mask_mismatch_pixels": bad_mask, "depth_mismatch_pixels": bad_depth}
            if n_cpu:
                out["cpu_baseline"] = {"value": n_cpu / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                                       "sample": "%d frames of the same batch through oracle/rtuf_oracle.c (single thread, %.1f s)" % (n_cpu, t_cpu)}
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
