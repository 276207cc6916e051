"""Synthetic workloads of BASELINE.json (configs C1-C5 of SURVEY.md section 8d), built through the
same host-side classes a user would use (URDFRenderer, forward kinematics, transform provider).

A workload is plain data: static geometry (draw calls per link, in draw order) plus, per stream,
the matrices the C ABI takes.  bench.py, __graft_entry__.smoke() and the tests feed it to the
HIP path and -- as the checker only -- to the CPU oracle.
"""
import os

import numpy as np

from realtime_urdf_filter_amd import urdf
from . import synthetic
from realtime_urdf_filter_amd.filter import URDFRenderer, CameraInfo
from realtime_urdf_filter_amd._capi import projection_from_intrinsics

EXAMPLE_URDF = """<robot name="example">
    <link name="world"/>
    <link name="wall1"><visual><geometry><box size="4 0.5 2" /></geometry></visual>
                       <collision><geometry><box size="4 0.5 2" /></geometry></collision></link>
    <link name="wall2"><visual><geometry><box size="4 0.5 2" /></geometry></visual>
                       <collision><geometry><box size="4 0.5 2" /></geometry></collision></link>
    <joint name="wall1_joint" type="fixed"><origin xyz="0 5 0" rpy="0 0 0.785398163"/>
        <parent link="world"/><child link="wall1"/></joint>
    <joint name="wall2_joint" type="fixed"><origin xyz="0 5 0" rpy="0 0 -0.785398163"/>
        <parent link="world"/><child link="wall2"/></joint>
</robot>"""
"""Same content as the reference's urdf/example.urdf.xml (two 4 x 0.5 x 2 boxes on fixed joints at
xyz 0 5 0, yaw +-0.785398163), re-typed as data for config C1 (SURVEY.md section 2)."""


class Workload:
    def __init__(self, name, width, height, n_streams):
        self.name, self.width, self.height, self.n_streams = name, width, height, n_streams
        self.models = []            # per model: list of links; per link: list of DrawCall
        self.link_tf = []           # per model: [N, L, 16] float64
        self.projection = None      # [N,16]
        self.offset_inv = None      # [N,16]
        self.cam_tf = None          # [N,16]
        self.near, self.far = 0.1, 8.0
        self.max_diff, self.replace_value = 0.05, 5.0      # launch/filter_parameters.yaml:14-16
        self.meta = {}
        self.kinematics = None      # arrays for rtuf_set_kinematics (model 0), see urdf.kinematic_arrays
        self.camera_frame_index = -1
        self.joint_q = None         # [N, frames]
        self.ensure_host_fk = lambda: None      # (pr2_workload(host_fk=False) replaces it: fills link_tf / cam_tf when first needed)

    def n_triangles(self):
        return int(sum(len(d.tris) for m in self.models for l in m for d in l))

    def n_vertices(self):
        return int(sum(len(d.verts) for m in self.models for l in m for d in l))

    def depth(self, stream):
        return synthetic.sensor_depth(self.width, self.height, stream)

    def depth_batch(self, first=0, n=None):
        n = self.n_streams if n is None else n
        return np.stack([self.depth(first + s) for s in range(n)])

    def load_into(self, ctx):
        """Adds every model to an rtuf Context and finalises; returns the model ids."""
        ids = []
        for links in self.models:
            m = ctx.add_model()
            for draws in links:
                l = ctx.add_link(m)
                for d in draws:
                    ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
            ids.append(m)
        ctx.finalize_models()
        return ids

    def stage(self, ctx, model_ids, first=0, n=None):
        n = self.n_streams if n is None else n
        sl = slice(first, first + n)
        ctx.set_cameras(0, self.projection[sl], self.offset_inv[sl], self.cam_tf[sl])
        for m, tf in zip(model_ids, self.link_tf):
            if tf.shape[1]:
                ctx.set_link_poses_batch(0, m, tf[sl])

    def stage_joint_positions(self, ctx, model_ids, first=0, n=None, first_call=True):
        """Same poses through on-device forward kinematics: only joint positions cross the bus."""
        n = self.n_streams if n is None else n
        sl = slice(first, first + n)
        if first_call:
            ctx.set_cameras(0, self.projection[sl], self.offset_inv[sl], None)     # intrinsics do not change per frame
        ctx.set_joint_positions(0, model_ids[0], self.joint_q[sl], None, self.camera_frame_index)
        for m, tf in list(zip(model_ids, self.link_tf))[1:]:
            if tf.shape[1]:
                ctx.set_link_poses_batch(0, m, tf[sl])

    def load_kinematics(self, ctx, model_ids):
        k = self.kinematics
        ctx.set_kinematics(model_ids[0], k["parent"], k["joint_type"], k["joint_origin"], k["joint_axis"], k["link_frame"], k["link_offset"])

    def oracle_draws(self, stream):
        """Draw list of one stream in the oracle's format."""
        out = []
        for links, tf in zip(self.models, self.link_tf):
            for li, draws in enumerate(links):
                for d in draws:
                    out.append((tf[stream, li], d.pre_op, d.op, d.verts, d.tris))
        return out


def _intrinsics(width, height):
    if (width, height) == (1280, 720):
        return 920.0, 920.0, 639.5, 359.5
    s = width / 640.0
    return 525.0 * s, 525.0 * s, (width - 1) / 2.0, (height - 1) / 2.0


def _projection_block(width, height, n):
    fx, fy, cx, cy = _intrinsics(width, height)
    P, _, _ = projection_from_intrinsics(fx, fy, cx, cy, width, height)
    return np.tile(P, (n, 1))


def example_workload(width=640, height=480):
    """C1: urdf/example.urdf.xml; camera at the world origin looking along +y
    (cam_x = world_x, cam_y = -world_z, cam_z = world_y)."""
    w = Workload("C1 example.urdf.xml", width, height, 1)
    model = urdf.Model.from_string(EXAMPLE_URDF)
    fk = urdf.forward_kinematics(model)
    tf = urdf.StaticTransformProvider()
    tf.set_frames(fk, "/EXAMPLE/")
    cam = urdf.Transform(np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]]), (0, 0, 0))   # world <- camera
    tf.frames["/world"] = urdf.Transform()
    tf.frames["/camera_rgb_optical_frame"] = cam
    # the model's links are published under the tf_prefix; the fixed frame is /world
    tf.frames["/EXAMPLE/world"] = urdf.Transform()
    rd = URDFRenderer(EXAMPLE_URDF, "/EXAMPLE", "/camera_rgb_optical_frame", "/world", tf, "visual", 1.0, [])
    rd.update_link_transforms()
    w.models = [[r.draws for r in rd.renderables_]]
    w.link_tf = [rd.link_matrices()[None]]
    w.projection = _projection_block(width, height, 1)
    w.offset_inv = np.tile(urdf.Transform().opengl_matrix(), (1, 1))
    w.cam_tf = tf.lookup_transform("/camera_rgb_optical_frame", "/world").opengl_matrix()[None]
    return w


_robot_cache = {}


def pr2_workload(n_streams, width=640, height=480, total_triangles=250000, seed=7, first_state_seed=1000,
                 walls=False, near_arm=False, host_fk=True):
    """C2/C3 (and C4 with walls=True): synthetic PR2-like robot, one random joint state per stream,
    camera = the head-mounted RGB optical frame, fixed frame = base_footprint.
    host_fk=False leaves out the host-side forward kinematics of every stream (3 ms each in Python): link_tf[0] and
    cam_tf then stay zero, which is fine for callers that pose the robot with on-device forward kinematics and feed the
    checker the matrices the device produced (bench.py's side legs)."""
    key = (total_triangles, seed)
    if key not in _robot_cache:
        _robot_cache[key] = synthetic.SyntheticRobot(total_triangles, seed)
    robot = _robot_cache[key]
    joint_state = robot.near_arm_joint_state if near_arm else robot.random_joint_state
    xml = robot.to_urdf_xml()
    model = urdf.Model.from_string(xml)

    def loader(uri):
        name = os.path.splitext(uri.split("://", 1)[1])[0]
        return robot.meshes[name]

    tf0 = urdf.StaticTransformProvider()
    rd = URDFRenderer(xml, "", robot.camera_frame, robot.fixed_frame, tf0, "visual", 1.0, [], loader)
    w = Workload("PR2-like %dk triangles x %d streams%s" % (robot.n_triangles() // 1000, n_streams, ", forearm in front of the lens" if near_arm else ""), width, height, n_streams)
    w.models = [[r.draws for r in rd.renderables_]]
    L = len(rd.renderables_)
    link_tf = np.zeros((n_streams, L, 16))
    cam_tf = np.zeros((n_streams, 16))
    wall_rd = None
    wall_tf = None
    if walls:
        wall_rd = URDFRenderer(EXAMPLE_URDF, "/walls", robot.camera_frame, robot.fixed_frame, tf0, "visual", 1.0, [])
        w.models.append([r.draws for r in wall_rd.renderables_])
        wall_tf = np.zeros((n_streams, len(wall_rd.renderables_), 16))
    def fill(s):
        """Host-side forward kinematics of stream s: link matrices, camera transform (and the static walls)."""
        q = joint_state(first_state_seed + s)
        fk = urdf.forward_kinematics(model, q)
        tf = urdf.StaticTransformProvider()
        tf.set_frames(fk, "/")
        rd.fixed_frame_ = "/" + robot.fixed_frame
        rd.update_link_transforms(None, tf)
        link_tf[s] = rd.link_matrices()
        cam_tf[s] = tf.lookup_transform("/" + robot.camera_frame, "/" + robot.fixed_frame).opengl_matrix()
        if walls:
            # two static walls, posed like launch/tracker.launch:5-6 relative to the fixed frame
            import math
            for k, (x, y, z, yaw) in enumerate(((1.5, -1.5, 1.0, 0.785398163), (-1.5, -1.2, 1.0, -0.785398163))):
                t = urdf.Transform.from_quaternion((0, 0, math.sin(yaw / 2), math.cos(yaw / 2)), (x + 2.0, y + 1.5, z))
                wall_tf[s, k] = (t * wall_rd.renderables_[k].link_offset).opengl_matrix()

    if host_fk:
        for s in range(n_streams):
            fill(s)
    elif walls:
        fill(0)
        wall_tf[1:] = wall_tf[0]                # (the walls do not move: posed relative to the fixed frame)
        link_tf[0] = 0.0
        cam_tf[0] = 0.0

    def ensure_host_fk():
        """Fills in the host-side forward kinematics a host_fk=False build left out (idempotent)."""
        if not w.meta["host_fk"]:
            for s in range(n_streams):
                fill(s)
            w.meta["host_fk"] = True
    w.link_tf = [link_tf] + ([wall_tf] if walls else [])
    w.projection = _projection_block(width, height, n_streams)
    w.offset_inv = np.tile(urdf.Transform().opengl_matrix(), (n_streams, 1))
    w.cam_tf = cam_tf
    w.meta = {"robot": "synthetic PR2-like", "links": len(robot.links), "links_with_geometry": L,
              "triangles": w.n_triangles(), "vertices": w.n_vertices(), "host_fk": bool(host_fk)}
    w.ensure_host_fk = ensure_host_fk
    # on-device forward kinematics inputs: the tree once, joint positions per stream
    strip = lambda n: n[1:] if n.startswith("/") else n
    w.kinematics = urdf.kinematic_arrays(model, [strip(r.name) for r in rd.renderables_], [r.link_offset for r in rd.renderables_])
    w.camera_frame_index = w.kinematics["frame_index"][robot.camera_frame]
    w.joint_q = np.stack([urdf.joint_vector(w.kinematics, joint_state(first_state_seed + s)) for s in range(n_streams)])
    return w
