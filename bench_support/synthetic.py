"""Deterministic synthetic inputs (no network, no pr2_description in the image).

* SplitMix64 generator (identical on every machine / numpy version).
* sensor depth frames of SURVEY.md section 8d (C1 formula + NaN / zero / +inf classes).
* a procedural "PR2-like" robot: the PR2's kinematic layout (base, 4 casters x 2 wheels, torso
  lift, head pan/tilt + sensor frames, two 7-dof arms with 4-finger-link grippers, ~88 links)
  expressed as URDF XML plus one procedural mesh per geometry-bearing link.  Link counts, joint
  types and limits follow the public PR2 description from memory; the meshes are lumpy
  super-ellipsoids whose triangle budget is a parameter ("visual"-like ~250k triangles,
  "collision"-like ~20k).  Every report that uses it says "synthetic PR2-like".
"""
import math

import numpy as np

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))

    def array(self, n):
        """n doubles in [0,1) (vectorised, same sequence as n calls of uniform())."""
        idx = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(self.s)
        self.s = int(idx[-1]) if n else self.s
        z = idx
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _hash2(xx, yy, salt):
    h = (xx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (yy.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ np.uint64(salt & MASK64)
    h = (h ^ (h >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    h = (h ^ (h >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return h ^ (h >> np.uint64(31))


def sensor_depth(width, height, stream=0, specials=True):
    """d(u,v) = clamp(2.5 + 1.5 sin(0.013 u + phase) cos(0.017 v), 0.4, 7.5); pixel-hash classes
    1/64 NaN, 1/64 zero, 1/1024 +inf."""
    yy, xx = np.mgrid[0:height, 0:width]
    sx, sy = width / 640.0, height / 480.0
    phase = 0.37 * stream
    d = np.clip(2.5 + 1.5 * np.sin(0.013 * xx / sx + phase) * np.cos(0.017 * yy / sy), 0.4, 7.5).astype(np.float32)
    if specials:
        h = _hash2(xx, yy, 0x1234 + stream)
        d[(h & np.uint64(63)) == np.uint64(1)] = np.nan
        d[(h & np.uint64(63)) == np.uint64(2)] = 0.0
        d[(h & np.uint64(1023)) == np.uint64(3)] = np.inf
    return d


# --------------------------------------------------------------------------------------
# procedural meshes
# --------------------------------------------------------------------------------------
def lumpy_ellipsoid(n_tris, radii, seed, lump=0.12, power=2.6):
    """Closed super-ellipsoid mesh with ~n_tris triangles (exactly 2*nu*(nv-1)).
    Returns (verts [V,3] f32, tris [T,3] u32)."""
    nv = max(3, int(round(math.sqrt(n_tris / 4.0))) + 1)
    nu = max(3, int(round(n_tris / (2.0 * (nv - 1)))))
    rng = SplitMix64(seed)
    ph = [rng.uniform(0, 2 * math.pi) for _ in range(6)]
    fr = [1 + int(rng.uniform(0, 4)) for _ in range(6)]
    verts = [(0.0, 0.0, radii[2])]
    e = 2.0 / power
    for i in range(1, nv):
        th = math.pi * i / nv
        for j in range(nu):
            p = 2 * math.pi * j / nu
            ct, st, cp, sp = math.cos(th), math.sin(th), math.cos(p), math.sin(p)
            sg = lambda v: math.copysign(abs(v) ** e, v)
            r = 1.0 + lump * (math.sin(fr[0] * p + ph[0]) * math.sin(fr[1] * th + ph[1]) +
                              0.5 * math.sin(fr[2] * 2 * p + ph[2]) * math.cos(fr[3] * th + ph[3]))
            verts.append((radii[0] * r * sg(st) * sg(cp), radii[1] * r * sg(st) * sg(sp), radii[2] * r * sg(ct)))
    verts.append((0.0, 0.0, -radii[2]))
    tris = []
    for j in range(nu):
        tris.append((0, 1 + j, 1 + (j + 1) % nu))
    for i in range(nv - 2):
        a, b = 1 + i * nu, 1 + (i + 1) * nu
        for j in range(nu):
            j2 = (j + 1) % nu
            tris.append((a + j, b + j, b + j2))
            tris.append((a + j, b + j2, a + j2))
    last = len(verts) - 1
    a = 1 + (nv - 2) * nu
    for j in range(nu):
        tris.append((a + j, last, a + (j + 1) % nu))
    return np.asarray(verts, np.float32), np.asarray(tris, np.uint32)


# --------------------------------------------------------------------------------------
# PR2-like robot description
# --------------------------------------------------------------------------------------
def _link(name, mesh=None, origin=(0, 0, 0, 0, 0, 0), weight=1.0, radii=(0.05, 0.05, 0.05)):
    return {"name": name, "mesh": mesh, "origin": origin, "weight": weight, "radii": radii}


def _joint(name, jtype, parent, child, xyz=(0, 0, 0), rpy=(0, 0, 0), axis=(0, 0, 1), lo=0.0, hi=0.0):
    return {"name": name, "type": jtype, "parent": parent, "child": child, "xyz": xyz, "rpy": rpy,
            "axis": axis, "lower": lo, "upper": hi}


def pr2_like_description():
    """Returns (links, joints): plain dict lists describing the PR2-like kinematic tree."""
    L, J = [], []
    L.append(_link("base_footprint"))
    L.append(_link("base_link", "base", (0, 0, 0.17, 0, 0, 0), 10.0, (0.33, 0.33, 0.14)))
    J.append(_joint("base_footprint_joint", "fixed", "base_footprint", "base_link", (0, 0, 0.051)))
    L.append(_link("base_laser_link", "laser", (0, 0, 0, 0, 0, 0), 0.5, (0.03, 0.03, 0.04)))
    J.append(_joint("base_laser_joint", "fixed", "base_link", "base_laser_link", (0.275, 0, 0.252)))
    for cx, cy, cname in ((0.2246, 0.2246, "fl"), (0.2246, -0.2246, "fr"), (-0.2246, 0.2246, "bl"), (-0.2246, -0.2246, "br")):
        c = "%s_caster_rotation_link" % cname
        L.append(_link(c, "caster", (0, 0, 0.03, 0, 0, 0), 1.5, (0.07, 0.05, 0.05)))
        J.append(_joint("%s_caster_rotation_joint" % cname, "continuous", "base_link", c, (cx, cy, 0.0282), axis=(0, 0, 1), lo=-math.pi, hi=math.pi))
        for wy, wname in ((0.049, "l"), (-0.049, "r")):
            w = "%s_caster_%s_wheel_link" % (cname, wname)
            L.append(_link(w, "wheel", (0, 0, 0, math.pi / 2, 0, 0), 1.0, (0.074, 0.074, 0.017)))
            J.append(_joint("%s_caster_%s_wheel_joint" % (cname, wname), "continuous", c, w, (0, wy, 0), axis=(0, 1, 0), lo=-math.pi, hi=math.pi))
    L.append(_link("torso_lift_link", "torso", (-0.05, 0, 0.35, 0, 0, 0), 8.0, (0.17, 0.2, 0.42)))
    J.append(_joint("torso_lift_joint", "prismatic", "base_link", "torso_lift_link", (-0.05, 0, 0.739675), axis=(0, 0, 1), lo=0.0, hi=0.31))
    L.append(_link("imu_link"))
    J.append(_joint("imu_joint", "fixed", "torso_lift_link", "imu_link", (-0.02977, -0.1497, 0.164)))
    L.append(_link("head_pan_link", "head_pan", (0.0, 0, 0.03, 0, 0, 0), 3.0, (0.1, 0.12, 0.06)))
    J.append(_joint("head_pan_joint", "revolute", "torso_lift_link", "head_pan_link", (-0.01707, 0, 0.38145), axis=(0, 0, 1), lo=-2.857, hi=2.857))
    L.append(_link("head_tilt_link", "head_tilt", (0.03, 0, 0.05, 0, 0, 0), 4.0, (0.1, 0.15, 0.08)))
    J.append(_joint("head_tilt_joint", "revolute", "head_pan_link", "head_tilt_link", (0.068, 0, 0), axis=(0, 1, 0), lo=-0.3712, hi=1.29626))
    L.append(_link("head_plate_frame"))
    J.append(_joint("head_plate_frame_joint", "fixed", "head_tilt_link", "head_plate_frame", (0.0232, 0, 0.0645)))
    L.append(_link("sensor_mount_link", "sensor_mount", (0, 0, 0, 0, 0, 0), 1.0, (0.04, 0.12, 0.03)))
    J.append(_joint("sensor_mount_frame_joint", "fixed", "head_plate_frame", "sensor_mount_link", (0, 0, 0)))
    for nm, y in (("high_def", -0.11), ("wide_stereo", 0.03), ("narrow_stereo", 0.06)):
        L.append(_link("%s_link" % nm))
        J.append(_joint("%s_frame_joint" % nm, "fixed", "sensor_mount_link", "%s_link" % nm, (0.045, y, 0.05)))
        L.append(_link("%s_optical_frame" % nm))
        J.append(_joint("%s_optical_frame_joint" % nm, "fixed", "%s_link" % nm, "%s_optical_frame" % nm, (0, 0, 0), (-math.pi / 2, 0, -math.pi / 2)))
    L.append(_link("head_mount_link", "head_mount", (0, 0, 0, 0, 0, 0), 1.0, (0.05, 0.14, 0.02)))
    J.append(_joint("head_mount_joint", "fixed", "head_plate_frame", "head_mount_link", (-0.138, 0, 0.091)))
    L.append(_link("head_mount_kinect_ir_link", "kinect", (0, 0, 0, 0, 0, 0), 1.5, (0.035, 0.14, 0.025)))
    J.append(_joint("head_mount_kinect_ir_joint", "fixed", "head_mount_link", "head_mount_kinect_ir_link", (-0.032267, 0.0125, 0.136453)))
    L.append(_link("head_mount_kinect_ir_optical_frame"))
    J.append(_joint("head_mount_kinect_ir_optical_frame_joint", "fixed", "head_mount_kinect_ir_link", "head_mount_kinect_ir_optical_frame", (0, 0, 0), (-math.pi / 2, 0, -math.pi / 2)))
    L.append(_link("head_mount_kinect_rgb_link"))
    J.append(_joint("head_mount_kinect_rgb_joint", "fixed", "head_mount_kinect_ir_link", "head_mount_kinect_rgb_link", (0, -0.03, 0)))
    L.append(_link("head_mount_kinect_rgb_optical_frame"))
    J.append(_joint("head_mount_kinect_rgb_optical_frame_joint", "fixed", "head_mount_kinect_rgb_link", "head_mount_kinect_rgb_optical_frame", (0, 0, 0), (-math.pi / 2, 0, -math.pi / 2)))
    L.append(_link("laser_tilt_mount_link", "tilt_laser", (0, 0, 0, 0, 0, 0), 1.0, (0.04, 0.04, 0.05)))
    J.append(_joint("laser_tilt_mount_joint", "revolute", "torso_lift_link", "laser_tilt_mount_link", (0.09893, 0, 0.227), axis=(0, 1, 0), lo=-0.7354, hi=1.43353))
    for side, sgn in (("r", -1.0), ("l", 1.0)):
        p = side + "_"
        L.append(_link(p + "shoulder_pan_link", "shoulder_pan", (0, 0, -0.1, 0, 0, 0), 6.0, (0.13, 0.13, 0.2)))
        lo, hi = (-2.2854, 0.7146) if side == "r" else (-0.7146, 2.2854)
        J.append(_joint(p + "shoulder_pan_joint", "revolute", "torso_lift_link", p + "shoulder_pan_link", (0, sgn * 0.188, 0), axis=(0, 0, 1), lo=lo, hi=hi))
        L.append(_link(p + "shoulder_lift_link", "shoulder_lift", (0, 0, 0, 0, 0, 0), 4.0, (0.09, 0.1, 0.1)))
        J.append(_joint(p + "shoulder_lift_joint", "revolute", p + "shoulder_pan_link", p + "shoulder_lift_link", (0.1, 0, 0), axis=(0, 1, 0), lo=-0.5236, hi=1.3963))
        L.append(_link(p + "upper_arm_roll_link", "upper_arm_roll", (0.08, 0, 0, 0, 0, 0), 1.0, (0.06, 0.06, 0.06)))
        J.append(_joint(p + "upper_arm_roll_joint", "revolute", p + "shoulder_lift_link", p + "upper_arm_roll_link", (0, 0, 0), axis=(1, 0, 0), lo=(-3.9 if side == "r" else -0.8), hi=(0.8 if side == "r" else 3.9)))
        L.append(_link(p + "upper_arm_link", "upper_arm", (0.21, 0, 0, 0, math.pi / 2, 0), 6.0, (0.08, 0.08, 0.2)))
        J.append(_joint(p + "upper_arm_joint", "fixed", p + "upper_arm_roll_link", p + "upper_arm_link"))
        L.append(_link(p + "elbow_flex_link", "elbow_flex", (0, 0, 0, 0, 0, 0), 2.0, (0.07, 0.08, 0.07)))
        J.append(_joint(p + "elbow_flex_joint", "revolute", p + "upper_arm_link", p + "elbow_flex_link", (0.4, 0, 0), axis=(0, 1, 0), lo=-2.3213, hi=0.0))
        L.append(_link(p + "forearm_roll_link", "forearm_roll", (0.06, 0, 0, 0, 0, 0), 0.7, (0.05, 0.05, 0.05)))
        J.append(_joint(p + "forearm_roll_joint", "continuous", p + "elbow_flex_link", p + "forearm_roll_link", axis=(1, 0, 0), lo=-math.pi, hi=math.pi))
        L.append(_link(p + "forearm_link", "forearm", (0.18, 0, 0, 0, math.pi / 2, 0), 5.0, (0.06, 0.07, 0.16)))
        J.append(_joint(p + "forearm_joint", "fixed", p + "forearm_roll_link", p + "forearm_link"))
        L.append(_link(p + "forearm_cam_frame"))
        J.append(_joint(p + "forearm_cam_frame_joint", "fixed", p + "forearm_roll_link", p + "forearm_cam_frame", (0.135, 0, 0.044), (-math.pi / 2, -0.56, 0)))
        L.append(_link(p + "wrist_flex_link", "wrist_flex", (0, 0, 0, 0, 0, 0), 1.0, (0.04, 0.05, 0.04)))
        J.append(_joint(p + "wrist_flex_joint", "revolute", p + "forearm_link", p + "wrist_flex_link", (0.321, 0, 0), axis=(0, 1, 0), lo=-2.18, hi=0.0))
        L.append(_link(p + "wrist_roll_link", "wrist_roll", (0.02, 0, 0, 0, 0, 0), 0.6, (0.035, 0.035, 0.035)))
        J.append(_joint(p + "wrist_roll_joint", "continuous", p + "wrist_flex_link", p + "wrist_roll_link", axis=(1, 0, 0), lo=-math.pi, hi=math.pi))
        L.append(_link(p + "gripper_palm_link", "gripper_palm", (0.06, 0, 0, 0, 0, 0), 3.0, (0.06, 0.065, 0.03)))
        J.append(_joint(p + "gripper_palm_joint", "fixed", p + "wrist_roll_link", p + "gripper_palm_link"))
        for f, fs in (("l", 1.0), ("r", -1.0)):
            fl = p + "gripper_%s_finger_link" % f
            L.append(_link(fl, "finger", (0.045, fs * 0.012, 0, 0, 0, 0), 1.0, (0.05, 0.015, 0.012)))
            J.append(_joint(p + "gripper_%s_finger_joint" % f, "revolute", p + "gripper_palm_link", fl, (0.07691, fs * 0.01, 0), axis=(0, 0, fs), lo=0.0, hi=0.548))
            ft = p + "gripper_%s_finger_tip_link" % f
            L.append(_link(ft, "finger_tip", (0.02, fs * 0.004, 0, 0, 0, 0), 0.7, (0.025, 0.008, 0.011)))
            J.append(_joint(p + "gripper_%s_finger_tip_joint" % f, "revolute", fl, ft, (0.09137, fs * 0.00495, 0), axis=(0, 0, -fs), lo=0.0, hi=0.548))
        L.append(_link(p + "gripper_tool_frame"))
        J.append(_joint(p + "gripper_tool_joint", "fixed", p + "gripper_palm_link", p + "gripper_tool_frame", (0.18, 0, 0)))
        L.append(_link(p + "gripper_motor_accelerometer_link", "accel", (0, 0, 0, 0, 0, 0), 0.1, (0.005, 0.005, 0.005)))
        J.append(_joint(p + "gripper_motor_accelerometer_joint", "fixed", p + "gripper_palm_link", p + "gripper_motor_accelerometer_link"))
    return L, J


class SyntheticRobot:
    """links / joints + per-link procedural mesh (object-space triangles, URDF mesh scale 1)."""

    def __init__(self, total_triangles=250000, seed=7, name="pr2_like"):
        self.name = name
        self.links, self.joints = pr2_like_description()
        wsum = sum(l["weight"] for l in self.links if l["mesh"])
        self.meshes = {}
        for i, l in enumerate(self.links):
            if not l["mesh"]:
                continue
            nt = max(24, int(total_triangles * l["weight"] / wsum))
            self.meshes[l["name"]] = lumpy_ellipsoid(nt, l["radii"], seed * 1000 + i)
        self.camera_frame = "head_mount_kinect_rgb_optical_frame"
        self.fixed_frame = "base_footprint"

    def n_triangles(self):
        return int(sum(len(t) for _, t in self.meshes.values()))

    def to_urdf_xml(self, mesh_uri_prefix="synthetic://"):
        out = ['<robot name="%s">' % self.name]
        for l in self.links:
            out.append('  <link name="%s">' % l["name"])
            if l["mesh"]:
                o = l["origin"]
                for tag in ("visual", "collision"):
                    out.append('    <%s><origin xyz="%r %r %r" rpy="%r %r %r"/><geometry><mesh filename="%s%s.stl"/></geometry></%s>' %
                               (tag, o[0], o[1], o[2], o[3], o[4], o[5], mesh_uri_prefix, l["name"], tag))
            out.append('  </link>')
        for j in self.joints:
            out.append('  <joint name="%s" type="%s">' % (j["name"], j["type"]))
            out.append('    <origin xyz="%r %r %r" rpy="%r %r %r"/>' % (tuple(j["xyz"]) + tuple(j["rpy"])))
            out.append('    <parent link="%s"/><child link="%s"/>' % (j["parent"], j["child"]))
            if j["type"] != "fixed":
                out.append('    <axis xyz="%r %r %r"/>' % tuple(j["axis"]))
                out.append('    <limit lower="%r" upper="%r" effort="100" velocity="1"/>' % (j["lower"], j["upper"]))
            out.append('  </joint>')
        out.append('</robot>')
        return "\n".join(out)

    #: the right forearm pointing at the head camera, its middle 0.25 m in front of the lens (found by hill climbing on the
    #: forward kinematics of this description): the self-filter's own normal case of an arm right in front of the sensor
    NEAR_ARM_BASE = {"r_shoulder_pan_joint": 0.7099, "r_shoulder_lift_joint": -0.5236, "r_upper_arm_roll_joint": -0.1123,
                     "r_elbow_flex_joint": -1.0739, "r_forearm_roll_joint": -2.5904, "r_wrist_flex_joint": -0.778,
                     "head_tilt_joint": 1.2172, "head_pan_joint": 0.3705}

    def near_arm_joint_state(self, seed, jitter=0.08):
        """random_joint_state(seed) with the right arm and the head moved to NEAR_ARM_BASE +- jitter rad: the forearm
        surface lies about 0.1 - 0.35 m in front of the lens (window z below and above 0.5: exact-z territory), fills
        the middle of the image and the gripper passes the near plane."""
        q = self.random_joint_state(seed)
        rng = SplitMix64(seed * 7919 + 17)
        lims = {j["name"]: (j["lower"], j["upper"]) for j in self.joints}
        for n, v in self.NEAR_ARM_BASE.items():
            lo, hi = lims[n]
            q[n] = min(hi, max(lo, v + (rng.uniform() - 0.5) * 2.0 * jitter))
        return q

    def random_joint_state(self, seed, arms_in_view=True):
        """Uniform in limits; with arms_in_view the shoulder/elbow ranges are narrowed
        so that the arms reach into the head camera's field of view, as during manipulation."""
        rng = SplitMix64(seed)
        q = {}
        for j in self.joints:
            if j["type"] == "fixed":
                continue
            lo, hi = j["lower"], j["upper"]
            u = rng.uniform()
            if arms_in_view:
                n = j["name"]
                if n.endswith("shoulder_pan_joint"):
                    c = -0.25 if n.startswith("r_") else 0.25
                    lo, hi = c - 0.35, c + 0.35
                elif n.endswith("shoulder_lift_joint"):
                    lo, hi = -0.35, 0.35
                elif n.endswith("elbow_flex_joint"):
                    lo, hi = -1.6, -0.4
                elif n.endswith("upper_arm_roll_joint"):
                    lo, hi = -0.6, 0.6
                elif n == "head_tilt_joint":
                    lo, hi = 0.55, 1.0
                elif n == "head_pan_joint":
                    lo, hi = -0.3, 0.3
            q[j["name"]] = lo + (hi - lo) * u
        return q
