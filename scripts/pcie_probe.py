#!/usr/bin/env python3
"""PCIe link rates of the box as the host-plane path sees them: pinned host memory <-> HBM with hipMemcpyAsync, one
direction alone and both at once, each direction on 1 / 2 / 4 HIP streams (the transfers of a direction split evenly), chunked
or whole.  What rtuf_filter_batch_async can hope for: `bench.py`'s with_host_copies leg moves 315 MB up and 393 MB down per
256-stream batch.   usage (GPU box): python scripts/pcie_probe.py > gpurun_out/pcie_probe.json"""
import json
import os
import sys
import time

import torch


def run(up_mb, down_mb, up_streams, down_streams, chunks, reps=6):
    dev = torch.device("cuda:0")
    h_up = torch.empty(int(up_mb * 1e6), dtype=torch.uint8).pin_memory() if up_mb else None
    h_dn = torch.empty(int(down_mb * 1e6), dtype=torch.uint8).pin_memory() if down_mb else None
    d_up = torch.empty(int(up_mb * 1e6), dtype=torch.uint8, device=dev) if up_mb else None
    d_dn = torch.empty(int(down_mb * 1e6), dtype=torch.uint8, device=dev) if down_mb else None
    su = [torch.cuda.Stream() for _ in range(up_streams)]
    sd = [torch.cuda.Stream() for _ in range(down_streams)]

    def once():
        if up_mb:
            parts = up_streams * chunks
            step = -(-h_up.numel() // parts)
            for i in range(parts):
                with torch.cuda.stream(su[i % up_streams]):
                    d_up[i * step:(i + 1) * step].copy_(h_up[i * step:(i + 1) * step], non_blocking=True)
        if down_mb:
            parts = down_streams * chunks
            step = -(-h_dn.numel() // parts)
            for i in range(parts):
                with torch.cuda.stream(sd[i % down_streams]):
                    h_dn[i * step:(i + 1) * step].copy_(d_dn[i * step:(i + 1) * step], non_blocking=True)

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    return {"up_GB_s": up_mb / 1e3 / el if up_mb else None, "down_GB_s": down_mb / 1e3 / el if down_mb else None, "ms": el * 1e3}


def main():
    out = {"env": {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "HIP_FORCE_DEV_KERNARG", "GPU_MAX_HW_QUEUES")}, "cases": []}
    for (u, d) in ((315, 0), (0, 393), (315, 393)):
        for streams in (1, 2, 4):
            for chunks in (1, 8):
                r = run(u, d, streams, streams, chunks)
                r.update({"up_MB": u, "down_MB": d, "streams_per_direction": streams, "chunks_per_stream": chunks})
                out["cases"].append(r)
                print("up %3d MB down %3d MB  %d stream(s) x %d chunk(s):  up %s  down %s GB/s  %.2f ms" % (
                    u, d, streams, chunks, "%.1f" % r["up_GB_s"] if r["up_GB_s"] else "-", "%.1f" % r["down_GB_s"] if r["down_GB_s"] else "-", r["ms"]), file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
