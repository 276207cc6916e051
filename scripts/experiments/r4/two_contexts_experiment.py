"""Experiment: alternate batches between two contexts (two HIP streams, two sets of bins) so that one
batch's small / low-occupancy kernels overlap the other's heavy ones.  Prints frames/s for 1 and 2 contexts."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import realtime_urdf_filter_amd as R
from bench_support import workloads as WL

n, W, H, steps = 256, 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 40
variants = [WL.pr2_workload(n, W, H, 250000, first_state_seed=1000 + 100000 * v) for v in range(2)]
wl0 = variants[0]
dev = torch.device("cuda:0")
d_depth = [torch.from_numpy(np.stack([wl.depth(s + 7 * v) for s in range(n)])).to(dev) for v, wl in enumerate(variants)]
for nctx in (1, 2, 1, 2):
    p = R.default_params(); p.filter_replace_value = wl0.replace_value; p.depth_distance_threshold = wl0.max_diff
    ctxs, idss = [], []
    for i in range(nctx):
        c = R.Context(W, H, n, 0, p); ids = wl0.load_into(c); wl0.load_kinematics(c, ids); ctxs.append(c); idss.append(ids)
    outs = [(torch.empty((n, H, W), dtype=torch.float32, device=dev), torch.empty((n, H, W), dtype=torch.uint8, device=dev)) for _ in range(4)]
    first = [True] * nctx
    def run(k):
        c = ctxs[k % nctx]; i = k % nctx
        variants[k % 2].stage_joint_positions(c, idss[i], first_call=first[i]); first[i] = False
        c.filter_batch_device(n, d_depth[k % 2].data_ptr(), outs[k % 4][0].data_ptr(), outs[k % 4][1].data_ptr())
    for k in range(6): run(k)
    for c in ctxs: c.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): run(6 + k)
    for c in ctxs: c.sync()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("contexts=%d  %.0f frames/s  %.4f ms/step" % (nctx, n * steps / dt, dt / steps * 1e3), flush=True)
    for c in ctxs: c.close()
