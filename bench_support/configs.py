"""Per-GPU shares of the BASELINE.json configs 3, 4 and 5, as bench.py, the GPU tests
and the multi-rank rehearsal run them.

Streams are independent, so a job is partitioned with no data-path collective (sharding.py):
  c3  640x480, PR2-like robot, `streams` camera streams PER GPU (weak scaling: the headline workload)
  c4  1280x720, PR2-like robot + the two static wall URDFs, `streams` (512) in TOTAL, block-partitioned over the
      ranks with sharding.shard_range (64 per GPU on 8 GPUs); the geometry is replicated
  c5  640x480, `urdfs` (64) distinct articulated URDFs x `per_urdf` (128) cameras each; URDF m lives on rank
      m % world (sharding.models_for_rank), so a GPU holds only its own robots (8 robots = 1024 streams on 8 GPUs);
      every stream renders only its own robot (rtuf_set_stream_models)

A RankShare is plain data plus the calls that feed it through the C ABI: load() (geometry, kinematic trees,
intrinsics, per-stream model selection -- once), stage(k) (step k's joint states: the only per-frame host input),
depth planes, and -- for the checker only -- the oracle's view of a stream.
"""
import numpy as np

from realtime_urdf_filter_amd import sharding
from . import workloads as WL

#: triangle budgets of the distinct robots of config 5 (cycled), chosen to span light to heavy URDFs
C5_BUDGETS = (40000, 90000, 150000, 60000, 250000, 120000, 30000, 200000)
DEPTH_POOL = 256        # distinct synthetic sensor frames per variant; larger shares reuse them cyclically


class Group:
    """Consecutive streams [first, first + count) of a rank that render the same robot (+ static extras)."""

    def __init__(self, first, count, global_first, variants, robot_index):
        self.first, self.count, self.global_first = first, count, global_first
        self.variants = variants              # [Workload]: one random joint state per stream and variant
        self.robot_index = robot_index        # global URDF number (c5) or 0
        self.model_ids = []                   # context model ids, in the order of variants[0].models


class RankShare:
    def __init__(self, name, workload, width, height, world, rank, scaling):
        self.name, self.workload, self.width, self.height = name, workload, width, height
        self.world, self.rank, self.scaling = world, rank, scaling
        self.groups = []
        self.n = 0
        self.total_streams = 0                # of the whole job (all ranks)
        self.link_base = {}                   # context model id -> first row in the context's link-matrix table
        self.n_links_total = 0
        self._static_staged = False
        self._cams_staged = False
        self.near_arm = False

    # ---- description --------------------------------------------------------------------------
    @property
    def wl0(self):
        return self.groups[0].variants[0]

    def n_variants(self):
        return len(self.groups[0].variants)

    def triangles_per_stream(self):
        """[n] triangles each stream renders (its own robot + static extras)."""
        out = np.zeros(self.n, np.int64)
        for g in self.groups:
            out[g.first:g.first + g.count] = g.variants[0].n_triangles()
        return out

    def describe(self):
        g0 = self.groups[0]
        if self.workload == "c5":
            tris = [g.variants[0].n_triangles() for g in self.groups]
            return ("C5: %dx%d depth, %d distinct synthetic articulated URDFs x %d cameras each (%d streams in total), "
                    "this rank: %d URDFs = %d streams, triangles per robot %s, every step poses every stream anew (%d distinct joint states per stream, cycled)"
                    % (self.width, self.height, self.total_streams // g0.count, g0.count, self.total_streams, len(self.groups), self.n, tris, self.n_variants()))
        extra = " + two static wall URDFs (urdf/example.urdf.xml boxes)" if self.workload == "c4" else ""
        extra += ", right forearm 0.1-0.35 m in front of the lens in every stream" if self.near_arm else ""
        return ("%s: %dx%d depth, synthetic PR2-like URDF (%d links with meshes, %d triangles)%s, batch=%d concurrent streams %s, "
                "every step poses every stream anew: joint state + head-camera pose (%d distinct states per stream, cycled)"
                % (self.workload.upper(), self.width, self.height, g0.variants[0].meta["links_with_geometry"], g0.variants[0].n_triangles(), extra,
                   self.n if self.scaling == "weak" else self.total_streams, "per GPU" if self.scaling == "weak" else "in total, %d on this rank" % self.n, self.n_variants()))

    # ---- feeding a context --------------------------------------------------------------------
    def load(self, ctx, on_device_fk=True):
        """Geometry (once), kinematic trees, intrinsics and model selection.  (Loading the share into another context
        starts its staging over: cameras and static extras go to the new context with the first stage() call.)"""
        base = 0
        self._static_staged = self._cams_staged = False
        for g in self.groups:
            wl = g.variants[0]
            g.model_ids = []
            for links in wl.models:
                m = ctx.add_model()
                for draws in links:
                    l = ctx.add_link(m)
                    for d in draws:
                        ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
                g.model_ids.append(m)
                self.link_base[m] = base
                base += len(links)
        self.n_links_total = base
        ctx.finalize_models()
        self.on_device_fk = on_device_fk
        for g in self.groups:
            wl = g.variants[0]
            if on_device_fk:
                k = wl.kinematics
                ctx.set_kinematics(g.model_ids[0], k["parent"], k["joint_type"], k["joint_origin"], k["joint_axis"], k["link_frame"], k["link_offset"])
            if len(self.groups) > 1:
                for s in range(g.first, g.first + g.count):
                    ctx.set_stream_models(s, g.model_ids)

    def stage(self, ctx, k):
        """Step k's poses.  With on-device forward kinematics only joint positions cross the bus (and that call never
        waits for the batches in flight); intrinsics and the static extras (walls) are staged once."""
        v = k % self.n_variants()
        for g in self.groups:
            wl = g.variants[v]
            if not self.on_device_fk:
                wl.ensure_host_fk()
            if not self._cams_staged:
                ctx.set_cameras(g.first, wl.projection, wl.offset_inv, None if self.on_device_fk else wl.cam_tf)
            if not self._static_staged:
                for m, tf in list(zip(g.model_ids, wl.link_tf))[1:]:
                    if tf.shape[1]:
                        ctx.set_link_poses_batch(g.first, m, tf)
            if self.on_device_fk:
                ctx.set_joint_positions(g.first, g.model_ids[0], wl.joint_q, None, wl.camera_frame_index)
            else:
                ctx.set_cameras(g.first, None, None, wl.cam_tf)
                ctx.set_link_poses_batch(g.first, g.model_ids[0], wl.link_tf[0])
        self._cams_staged = True
        self._static_staged = True

    def depth_host(self, variant, pool_size=DEPTH_POOL, tiled=True):
        """[n,H,W] float32 sensor planes of one variant (synthetic; shares larger than the pool reuse frames cyclically).
        tiled=False returns only the pool's [min(n, pool_size),H,W] distinct planes (stream s sees plane s % len)."""
        wl = self.wl0
        gfirst = self.groups[0].global_first
        pool = [wl.depth((gfirst + s) % 100003 + 7 * variant) for s in range(min(self.n, pool_size))]
        return np.stack([pool[s % len(pool)] for s in range(self.n)]) if tiled else np.stack(pool)

    # ---- the checker's view ---------------------------------------------------------------------
    def group_of(self, s):
        for g in self.groups:
            if g.first <= s < g.first + g.count:
                return g
        raise IndexError(s)

    def oracle_frame(self, k, s, link_tf_device=None, cam_tf_device=None):
        """(projection, draws, offset_inv, cam_tf) of rank-local stream s at step k for oracle.bindings.filter_frame.
        With the matrices the GPU's forward kinematics produced (Context.read_poses) the oracle is fed exactly
        what the rasteriser saw; otherwise the host-side FK of the workload."""
        g = self.group_of(s)
        wl = g.variants[k % self.n_variants()]
        j = s - g.first
        if link_tf_device is None or cam_tf_device is None:
            wl.ensure_host_fk()               # (a share built with host_fk=False computes a variant's host-side kinematics when first asked)
        draws = []
        for mi, (m, links) in enumerate(zip(g.model_ids, wl.models)):
            for li, dl in enumerate(links):
                tf = link_tf_device[s, self.link_base[m] + li] if link_tf_device is not None else wl.link_tf[mi][j, li]
                for d in dl:
                    draws.append((tf, d.pre_op, d.op, d.verts, d.tris))
        cam = cam_tf_device[s] if cam_tf_device is not None else wl.cam_tf[j]
        return wl.projection[j], draws, wl.offset_inv[j], cam

    def host_fk_error(self, k, link_tf_device, cam_tf_device):
        """max |device FK - host FK| over the share (diagnostic)."""
        worst = 0.0
        for g in self.groups:
            wl = g.variants[k % self.n_variants()]
            wl.ensure_host_fk()
            sl = slice(g.first, g.first + g.count)
            for mi, m in enumerate(g.model_ids):
                nl = wl.link_tf[mi].shape[1]
                if nl:
                    worst = max(worst, float(np.abs(link_tf_device[sl, self.link_base[m]:self.link_base[m] + nl] - wl.link_tf[mi]).max()))
            worst = max(worst, float(np.abs(cam_tf_device[sl] - wl.cam_tf).max()))
        return worst


def build(workload="c3", world=1, rank=0, streams=None, triangles=250000, variants=2, width=None, height=None,
          urdfs=64, per_urdf=128, near_arm=False, host_fk=True):
    """The share of `rank` in a `world`-GPU job of BASELINE config `workload`.  Seeds derive from GLOBAL stream and
    URDF numbers, so ranks never repeat each other's joint states and a share does not depend on how many ranks
    there are beyond which streams it holds."""
    workload = workload.lower()
    if workload == "c3":
        W, H = width or 640, height or 480
        n = streams or 256
        sh = RankShare("C3", "c3", W, H, world, rank, "weak")
        gfirst = rank * n
        vs = [WL.pr2_workload(n, W, H, triangles, first_state_seed=1000 + 100000 * v + 1000003 * rank, near_arm=near_arm, host_fk=host_fk) for v in range(variants)]
        sh.groups = [Group(0, n, gfirst, vs, 0)]
        sh.n, sh.total_streams = n, n * world
        sh.near_arm = near_arm
    elif workload == "c4":
        W, H = width or 1280, height or 720
        total = streams or 512
        first, n = sharding.shard_range(total, world, rank)
        if n <= 0:
            raise ValueError("c4: rank %d of %d has no streams (total %d)" % (rank, world, total))
        sh = RankShare("C4", "c4", W, H, world, rank, "strong")
        vs = [WL.pr2_workload(n, W, H, triangles, first_state_seed=2000 + 100000 * v + first, walls=True, near_arm=near_arm, host_fk=host_fk) for v in range(variants)]
        sh.groups = [Group(0, n, first, vs, 0)]
        sh.near_arm = near_arm
        sh.n, sh.total_streams = n, total
    elif workload == "c5":
        W, H = width or 640, height or 480
        per = streams or per_urdf
        mine = sharding.models_for_rank(urdfs, world, rank)
        if not mine:
            raise ValueError("c5: rank %d of %d has no URDFs (total %d)" % (rank, world, urdfs))
        sh = RankShare("C5", "c5", W, H, world, rank, "strong")
        for i, m in enumerate(mine):
            budget = C5_BUDGETS[m % len(C5_BUDGETS)] if triangles == 250000 else max(triangles // (1 + m % 4), 500)
            vs = [WL.pr2_workload(per, W, H, total_triangles=budget, seed=21 + m, first_state_seed=3000 + 1000 * m + 500 * v, host_fk=host_fk) for v in range(variants)]
            sh.groups.append(Group(i * per, per, m * per, vs, m))
        sh.n, sh.total_streams = per * len(mine), per * urdfs
    else:
        raise ValueError("unknown workload %r (c3, c4, c5)" % workload)
    return sh
