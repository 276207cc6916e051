"""Writes a job of bench_support.configs (all ranks' streams, i.e. a world-1 share) as the binary scene file that
examples/multi_gpu_filter.cpp reads: the C++ multi-device host then runs the very robots, joint states and sensor frames
the Python tests know, and its dumped stream can be checked against the oracle.  Test infrastructure."""
import struct

import numpy as np


def write_scene(path, share, k=0, n_depth=8):
    """share: configs.RankShare of world size 1 (the whole job); k: the step whose joint states / poses are written."""
    wl0 = share.wl0
    v = k % share.n_variants()
    out = [b"RTUFSCN1", struct.pack("<iiii", share.width, share.height, share.n, sum(len(g.variants[0].models) for g in share.groups)),
           struct.pack("<ffff", wl0.near, wl0.far, wl0.max_diff, wl0.replace_value)]
    model_of_stream = np.full(share.n, -1, np.int32)
    tail = []
    gm = 0
    for g in share.groups:
        wl = g.variants[v]
        for mi, links in enumerate(wl.models):
            out.append(struct.pack("<i", len(links)))
            for draws in links:
                out.append(struct.pack("<i", len(draws)))
                for d in draws:
                    verts = np.ascontiguousarray(d.verts, np.float32)
                    tris = np.ascontiguousarray(d.tris, np.uint32)
                    out.append(struct.pack("<ifffii", int(d.pre_op), float(d.op[0]), float(d.op[1]), float(d.op[2]), len(verts), len(tris)))
                    out.append(verts.tobytes())
                    out.append(tris.tobytes())
            has_kin = mi == 0 and wl.kinematics is not None
            out.append(struct.pack("<i", 1 if has_kin else 0))
            # per-stream rows of this model for ALL streams of the job (streams of other groups: zeros, never read)
            if has_kin:
                kin = wl.kinematics
                nf = len(kin["parent"])
                out.append(struct.pack("<i", nf))
                out.append(np.ascontiguousarray(kin["parent"], np.int32).tobytes())
                out.append(np.ascontiguousarray(kin["joint_type"], np.int32).tobytes())
                out.append(np.ascontiguousarray(kin["joint_origin"], np.float64).tobytes())
                out.append(np.ascontiguousarray(kin["joint_axis"], np.float64).tobytes())
                out.append(np.ascontiguousarray(kin["link_frame"], np.int32).tobytes())
                out.append(np.ascontiguousarray(kin["link_offset"], np.float64).tobytes())
                out.append(struct.pack("<i", int(wl.camera_frame_index)))
                q = np.zeros((share.n, nf), np.float64)
                q[g.first:g.first + g.count] = wl.joint_q
                tail.append(q.tobytes())
            else:
                tf = np.zeros((share.n, len(links), 16), np.float64)
                tf[:, :] = np.eye(4).reshape(16)
                tf[g.first:g.first + g.count] = wl.link_tf[mi]
                tail.append(tf.tobytes())
            if len(share.groups) > 1:
                model_of_stream[g.first:g.first + g.count] = gm       # (one robot per group: config 5)
            gm += 1
    proj = np.zeros((share.n, 16)); off = np.zeros((share.n, 16)); cam = np.zeros((share.n, 16))
    for g in share.groups:
        wl = g.variants[v]
        sl = slice(g.first, g.first + g.count)
        proj[sl], off[sl], cam[sl] = wl.projection, wl.offset_inv, wl.cam_tf
    out.append(model_of_stream.tobytes())
    out += [proj.astype(np.float64).tobytes(), off.astype(np.float64).tobytes(), cam.astype(np.float64).tobytes()]
    out += tail
    depth = np.stack([wl0.depth(s) for s in range(n_depth)]).astype(np.float32)
    out.append(struct.pack("<i", n_depth))
    out.append(depth.tobytes())
    with open(path, "wb") as f:
        for b in out:
            f.write(b)
    return depth
