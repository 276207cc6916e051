#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: host planes in, host planes out (the reference's filter() shape,
src/urdf_filter.cpp:233-234, :729-735), through rtuf_filter_batch_async with pinned memory and two
batches in flight, beside the synchronous rtuf_filter_batch.  This is NOT bench.py's `value` (that one has
its inputs resident in HBM); DESIGN.md section 5 quotes these numbers.

    python scripts/host_planes_rate.py [--streams 256] [--steps 30] > gpurun_out/host_planes.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import realtime_urdf_filter_amd as R                      # noqa: E402
from bench_support import workloads as WL      # noqa: E402
from realtime_urdf_filter_amd.filter import depth_f32_to_u16, depth_u16_to_f32   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--triangles", type=int, default=250000)
    ap.add_argument("--check-frames", type=int, default=3)
    ap.add_argument("--pipelines", type=int, default=1, help="rtuf_params.pipelines")
    args = ap.parse_args()
    n, W, H = args.streams, 640, 480
    variants = [WL.pr2_workload(n, W, H, total_triangles=args.triangles, first_state_seed=1000 + 5000 * v) for v in range(2)]
    wl0 = variants[0]
    p = R.default_params()
    p.filter_replace_value, p.depth_distance_threshold = wl0.replace_value, wl0.max_diff
    p.pipelines = args.pipelines if args.pipelines > 1 else 0
    ctx = R.Context(W, H, n, 0, p)
    ids = wl0.load_into(ctx)
    wl0.load_kinematics(ctx, ids)
    out = {"workload": "C3 geometry and poses (%d triangles, %d streams, 640x480), planes in host memory" % (wl0.meta["triangles"], n),
           "steps": args.steps, "pipelines": args.pipelines, "modes": {}}
    from oracle import bindings as O
    for fmt in ("32FC1", "16UC1"):
        dt = np.uint16 if fmt == "16UC1" else np.float32
        h_in = [ctx.host_alloc((n, H, W), dt) for _ in range(2)]
        h_out = [ctx.host_alloc((n, H, W), dt) for _ in range(2)]
        h_mask = [ctx.host_alloc((n, H, W), np.uint8) for _ in range(2)]
        for v, wl in enumerate(variants):
            d = np.stack([wl.depth(s + 7 * v) for s in range(n)])
            h_in[v][...] = depth_f32_to_u16(np.nan_to_num(d, nan=0.0, posinf=0.0)) if fmt == "16UC1" else d
        bytes_in = h_in[0].nbytes
        bytes_out = h_out[0].nbytes + h_mask[0].nbytes
        first = [True]

        def stage(k):
            variants[k % 2].stage_joint_positions(ctx, ids, first_call=first[0])
            first[0] = False

        def run(pipelined):
            for k in range(3):                                   # warm-up
                stage(k); ctx.filter_batch_async(h_in[k % 2], h_out[k % 2], h_mask[k % 2]); ctx.sync()
            stage(0)
            t0 = time.perf_counter()
            for k in range(args.steps):
                ctx.filter_batch_async(h_in[k % 2], h_out[k % 2], h_mask[k % 2])
                if not pipelined:
                    ctx.sync()                                   # = rtuf_filter_batch: upload, kernels, download in series
                stage(k + 1)
            ctx.sync()
            return time.perf_counter() - t0

        for name, pipelined in (("synchronous", False), ("two batches in flight", True)):
            el = run(pipelined)
            k_last = args.steps - 1
            wl, v = variants[k_last % 2], k_last % 2
            bad = 0
            for s in range(min(args.check_frames, n)):
                d32 = depth_u16_to_f32(h_in[v][s]) if fmt == "16UC1" else h_in[v][s]
                om, ok = O.filter_frame(d32, wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                        max_diff=wl.max_diff, replace_value=wl.replace_value)
                want = depth_f32_to_u16(om) if fmt == "16UC1" else om
                bad += int((ok != h_mask[v][s]).sum()) + int((want.view(np.uint16 if fmt == "16UC1" else np.uint32) != h_out[v][s].view(np.uint16 if fmt == "16UC1" else np.uint32)).sum())
            fps = n * args.steps / el
            out["modes"]["%s, %s" % (fmt, name)] = {
                "frames_per_s": fps, "ms_per_step": el / args.steps * 1e3,
                "host_to_device_GB_per_s": bytes_in * args.steps / el / 1e9, "device_to_host_GB_per_s": bytes_out * args.steps / el / 1e9,
                "frames_checked": min(args.check_frames, n), "mismatching_values": bad}
        # mask-only output, one bit per pixel (rtuf_filter_batch_bits*_async): the download shrinks from 5 (3) bytes
        # per pixel to 1/8, the path becomes upload-bound; the consumer expands where it needs full planes
        words = ctx.mask_bits_words()
        h_bits = [ctx.host_alloc((n, words), np.uint32) for _ in range(2)]
        for k in range(3):
            stage(k); ctx.filter_batch_bits_async(h_in[k % 2], h_bits[k % 2]); ctx.sync()
        stage(0)
        t0 = time.perf_counter()
        for k in range(args.steps):
            ctx.filter_batch_bits_async(h_in[k % 2], h_bits[k % 2])
            stage(k + 1)
        ctx.sync()
        el = time.perf_counter() - t0
        k_last = args.steps - 1
        wl, v = variants[k_last % 2], k_last % 2
        bad = 0
        for s in range(min(args.check_frames, n)):
            d32 = depth_u16_to_f32(h_in[v][s]) if fmt == "16UC1" else h_in[v][s]
            om, ok = O.filter_frame(d32, wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s],
                                    max_diff=wl.max_diff, replace_value=wl.replace_value)
            m2, k2 = R.expand_mask_bits(h_in[v][s], h_bits[v][s], wl.replace_value)
            want = depth_f32_to_u16(om) if fmt == "16UC1" else om
            bad += int((ok != k2).sum()) + int((want.view(np.uint16 if fmt == "16UC1" else np.uint32) != m2.view(np.uint16 if fmt == "16UC1" else np.uint32)).sum())
        # host-side expansion rate of one core (masked depth + byte mask of a frame from its sensor plane and bits)
        c0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - c0 < 1.0:
            R.expand_mask_bits(h_in[0][reps % n], h_bits[0][reps % n], wl0.replace_value)
            reps += 1
        exp_rate = reps / (time.perf_counter() - c0)
        out["modes"]["%s in, mask bits out, two batches in flight" % fmt] = {
            "frames_per_s": n * args.steps / el, "ms_per_step": el / args.steps * 1e3,
            "host_to_device_GB_per_s": bytes_in * args.steps / el / 1e9, "device_to_host_GB_per_s": h_bits[0].nbytes * args.steps / el / 1e9,
            "frames_checked": min(args.check_frames, n), "mismatching_values_after_expansion": bad,
            "host_expansion_frames_per_s_per_core": exp_rate}
        for a in h_in + h_out + h_mask + h_bits:
            ctx.host_free(a)
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
