"""CPU, world_size 2 over gloo: the multi-GPU path shards independent streams with no data-path
collective; the only exchange is the tiny end-of-batch gather.  Each rank filters its shard with the
CPU oracle standing in for its GPU and the gathered result must equal the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, out_q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from realtime_urdf_filter_amd import sharding
    from bench_support import workloads as WL
    from oracle import bindings as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = WL.pr2_workload(n_streams, 96, 72, total_triangles=1500)
    first, cnt = sharding.shard_range(n_streams, world, rank)
    sums = torch.zeros(n_streams, dtype=torch.int64)
    for s in range(first, first + cnt):
        _, mask = O.filter_frame(wl.depth(s), wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s], replace_value=5.0)
        sums[s] = int((mask > 0).sum())
    dist.barrier()
    total, tmax = sharding.gather_frame_counts(dist, cnt, 0.5 + rank)
    dist.all_reduce(sums)          # the optional result gather (a few bytes per stream)
    if rank == 0:
        out_q.put((total, tmax, sums.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process():
    import torch.multiprocessing as mp
    from bench_support import workloads as WL
    from oracle import bindings as O
    n = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, tmax, sums = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == n and tmax == 1.5
    wl = WL.pr2_workload(n, 96, 72, total_triangles=1500)
    ref = []
    for s in range(n):
        _, mask = O.filter_frame(wl.depth(s), wl.projection[s], wl.oracle_draws(s), wl.offset_inv[s], wl.cam_tf[s], replace_value=5.0)
        ref.append(int((mask > 0).sum()))
    assert sums == ref and sum(ref) > 0
