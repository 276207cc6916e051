"""URDF reader, tf-style transform algebra (double precision) and forward kinematics.

Replaces, for hosts without ROS, what the reference obtains from urdfdom
(`urdf::Model::initString`, src/urdf_renderer.cpp:67-79), from tf/Bullet LinearMath
(`tf::Transform`, `getOpenGLMatrix`, src/renderable.cpp:59-68, src/urdf_filter.cpp:602-614) and
from the TF tree (`lookupTransform`, src/urdf_renderer.cpp:173-190).  None of those libraries is
in the image; their arithmetic is restated from the published sources [recall].
"""
import math
import xml.etree.ElementTree as ET

import numpy as np


# ---- tf::Transform algebra (row-major 3x3 basis + origin, doubles) ---------------------
class Transform:
    __slots__ = ("basis", "origin")

    def __init__(self, basis=None, origin=None):
        self.basis = np.eye(3) if basis is None else np.asarray(basis, np.float64).reshape(3, 3).copy()
        self.origin = np.zeros(3) if origin is None else np.asarray(origin, np.float64).reshape(3).copy()

    @staticmethod
    def from_quaternion(q, origin=(0, 0, 0)):
        """tf::Transform(q, v): Matrix3x3::setRotation."""
        x, y, z, w = [float(v) for v in q]
        d = x * x + y * y + z * z + w * w
        s = 2.0 / d
        xs, ys, zs = x * s, y * s, z * s
        wx, wy, wz = w * xs, w * ys, w * zs
        xx, xy, xz = x * xs, x * ys, x * zs
        yy, yz, zz = y * ys, y * zs, z * zs
        b = np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                      [xy + wz, 1.0 - (xx + zz), yz - wx],
                      [xz - wy, yz + wx, 1.0 - (xx + yy)]])
        return Transform(b, origin)

    def __mul__(self, other):
        if isinstance(other, Transform):
            return Transform(self.basis @ other.basis, self.basis @ other.origin + self.origin)
        return self.basis @ np.asarray(other, np.float64) + self.origin

    def inverse(self):
        inv = self.basis.T
        return Transform(inv, inv @ (-self.origin))

    def get_rotation(self):
        """Matrix3x3::getRotation -> quaternion (x, y, z, w)."""
        m = self.basis
        trace = m[0, 0] + m[1, 1] + m[2, 2]
        t = [0.0] * 4
        if trace > 0.0:
            s = math.sqrt(trace + 1.0)
            t[3] = s * 0.5
            s = 0.5 / s
            t[0] = (m[2, 1] - m[1, 2]) * s
            t[1] = (m[0, 2] - m[2, 0]) * s
            t[2] = (m[1, 0] - m[0, 1]) * s
        else:
            i = (2 if m[1, 1] < m[2, 2] else 1) if m[0, 0] < m[1, 1] else (2 if m[0, 0] < m[2, 2] else 0)
            j, k = (i + 1) % 3, (i + 2) % 3
            s = math.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
            t[i] = s * 0.5
            s = 0.5 / s
            t[3] = (m[k, j] - m[j, k]) * s
            t[j] = (m[j, i] + m[i, j]) * s
            t[k] = (m[k, i] + m[i, k]) * s
        return tuple(t)

    def opengl_matrix(self):
        """getOpenGLMatrix: column-major [R|t; 0 0 0 1]."""
        m = np.zeros(16)
        b = self.basis
        m[0], m[1], m[2] = b[0, 0], b[1, 0], b[2, 0]
        m[4], m[5], m[6] = b[0, 1], b[1, 1], b[2, 1]
        m[8], m[9], m[10] = b[0, 2], b[1, 2], b[2, 2]
        m[12], m[13], m[14] = self.origin
        m[15] = 1.0
        return m


def quaternion_from_rpy(roll, pitch, yaw):
    """urdf::Rotation::setFromRPY (urdfdom_headers pose.h)."""
    phi, the, psi = roll / 2.0, pitch / 2.0, yaw / 2.0
    x = math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi)
    y = math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi)
    z = math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi)
    w = math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi)
    n = math.sqrt(x * x + y * y + z * z + w * w)
    if n > 0:
        x, y, z, w = x / n, y / n, z / n, w / n
    return (x, y, z, w)


def normalize_quaternion(q):
    n = math.sqrt(sum(v * v for v in q))
    return tuple(v / n for v in q)


def pose_to_transform(xyz, rpy):
    """urdf::Pose -> the reference's link_offset (src/urdf_renderer.cpp:160-164)."""
    return Transform.from_quaternion(normalize_quaternion(quaternion_from_rpy(*rpy)), xyz)


# ---- URDF model ----------------------------------------------------------------------------
class Geometry:
    def __init__(self, kind, **kw):
        self.kind = kind            # "box" | "cylinder" | "sphere" | "mesh"
        self.__dict__.update(kw)


class Visual:
    def __init__(self, xyz, rpy, geometry):
        self.xyz, self.rpy, self.geometry = xyz, rpy, geometry


class Link:
    def __init__(self, name):
        self.name = name
        self.visual_array = []
        self.collision_array = []


class Joint:
    def __init__(self, name, jtype, parent, child, xyz, rpy, axis, lower, upper, mimic=None):
        self.name, self.type, self.parent, self.child = name, jtype, parent, child
        self.xyz, self.rpy, self.axis, self.lower, self.upper, self.mimic = xyz, rpy, axis, lower, upper, mimic


def _floats(s, n, default):
    if s is None:
        return tuple(default)
    v = tuple(float(x) for x in s.split())
    if len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, s))
    return v


def _parse_geometry(el):
    g = el.find("geometry")
    if g is None or len(g) == 0:
        return None
    c = g[0]
    if c.tag == "box":
        return Geometry("box", size=_floats(c.get("size"), 3, (0, 0, 0)))
    if c.tag == "cylinder":
        return Geometry("cylinder", radius=float(c.get("radius")), length=float(c.get("length")))
    if c.tag == "sphere":
        return Geometry("sphere", radius=float(c.get("radius")))
    if c.tag == "mesh":
        return Geometry("mesh", filename=c.get("filename"), scale=_floats(c.get("scale"), 3, (1, 1, 1)))
    return None


class Model:
    """Minimal urdf::Model."""

    def __init__(self):
        self.name = ""
        self.links = {}
        self.joints = {}

    @staticmethod
    def from_string(xml):
        root = ET.fromstring(xml)
        if root.tag != "robot":
            raise ValueError("URDF root element must be <robot>")
        m = Model()
        m.name = root.get("name", "")
        for le in root.findall("link"):
            link = Link(le.get("name"))
            for tag, arr in (("visual", link.visual_array), ("collision", link.collision_array)):
                for ve in le.findall(tag):
                    geo = _parse_geometry(ve)
                    if geo is None:
                        continue
                    o = ve.find("origin")
                    xyz = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
                    rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
                    arr.append(Visual(xyz, rpy, geo))
            m.links[link.name] = link
        for je in root.findall("joint"):
            o = je.find("origin")
            xyz = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
            rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
            ax = je.find("axis")
            axis = _floats(ax.get("xyz") if ax is not None else None, 3, (1, 0, 0))
            lim = je.find("limit")
            lower = float(lim.get("lower", 0.0)) if lim is not None else 0.0
            upper = float(lim.get("upper", 0.0)) if lim is not None else 0.0
            mim = je.find("mimic")
            mimic = None
            if mim is not None and mim.get("joint"):
                # <mimic joint="other" multiplier="m" offset="o">: position = m * position(other) + o  (urdfdom defaults 1, 0)
                mimic = (mim.get("joint"), float(mim.get("multiplier", 1.0)), float(mim.get("offset", 0.0)))
            j = Joint(je.get("name"), je.get("type"), je.find("parent").get("link"), je.find("child").get("link"),
                      xyz, rpy, axis, lower, upper, mimic)
            m.joints[j.name] = j
        return m

    def get_links(self):
        """urdf::Model::getLinks iterates a std::map: links ordered by name."""
        return [self.links[k] for k in sorted(self.links)]

    def root_link(self):
        children = {j.child for j in self.joints.values()}
        roots = [n for n in self.links if n not in children]
        if len(roots) != 1:
            raise ValueError("URDF must have exactly one root link, found %r" % roots)
        return roots[0]


def resolve_mimic(model, joint_positions):
    """{joint: position} with every <mimic> joint filled in from the joint it follows (multiplier * position + offset;
    chains are followed, a cycle raises).  This is what joint_state_publisher does before robot_state_publisher turns
    joint states into the TF frames the reference looks up (src/urdf_renderer.cpp:173-190): the PR2's gripper fingers
    are mimic joints.  Positions given explicitly for a mimic joint are kept."""
    q = dict(joint_positions or {})

    def value(name, seen):
        if name in q:
            return float(q[name])
        j = model.joints.get(name)
        if j is None or j.mimic is None:
            return 0.0
        if name in seen:
            raise ValueError("mimic cycle through joint %r" % name)
        src, mult, off = j.mimic
        q[name] = mult * value(src, seen | {name}) + off
        return q[name]

    for name, j in model.joints.items():
        if j.mimic is not None:
            value(name, frozenset())
    return q


def joint_motion(joint, q):
    """Transform contributed by the joint variable (KDL / robot_state_publisher semantics)."""
    if joint.type in ("revolute", "continuous"):
        ax = np.asarray(joint.axis, np.float64)
        n = np.linalg.norm(ax)
        ax = ax / n if n > 0 else np.array([1.0, 0, 0])
        h = 0.5 * q
        s = math.sin(h)
        return Transform.from_quaternion((ax[0] * s, ax[1] * s, ax[2] * s, math.cos(h)))
    if joint.type == "prismatic":
        return Transform(None, np.asarray(joint.axis, np.float64) * q)
    return Transform()


def forward_kinematics(model, joint_positions=None, root_transform=None):
    """{link name: Transform root<-link}.  The 'fixed frame' of the reference is whatever TF frame
    the caller names; with the root link as fixed frame this replaces the per-link lookupTransform
    calls of URDFRenderer::update_link_transforms."""
    q = resolve_mimic(model, joint_positions)
    by_parent = {}
    for j in model.joints.values():
        by_parent.setdefault(j.parent, []).append(j)
    out = {}
    root = model.root_link()
    out[root] = root_transform if root_transform is not None else Transform()
    stack = [root]
    while stack:
        p = stack.pop()
        for j in sorted(by_parent.get(p, []), key=lambda jj: jj.name):
            t = out[p] * pose_to_transform(j.xyz, j.rpy) * joint_motion(j, float(q.get(j.name, 0.0)))
            out[j.child] = t
            stack.append(j.child)
    return out


class StaticTransformProvider:
    """Stand-in for tf::TransformListener: frames with known poses in one common root frame.

    lookup_transform(target, source) returns the transform that takes points from `source` into
    `target` (tf semantics), or raises KeyError (tf::TransformException in the reference)."""

    def __init__(self, frames=None):
        self.frames = dict(frames or {})      # name -> Transform root<-frame

    def set_frames(self, frames, prefix=""):
        for k, v in frames.items():
            self.frames[prefix + k] = v

    def lookup_transform(self, target, source, stamp=None):
        t, s = self.frames[target], self.frames[source]
        return t.inverse() * s


def kinematic_arrays(model, link_names, link_offsets):
    """Arrays for rtuf_set_kinematics: frames = the model's links in topological order.

    link_names: for every renderable (in upload order) the URDF link it is attached to;
    link_offsets: their Renderable.link_offset transforms.
    Returns dict(parent, joint_type, joint_origin [F,16], joint_axis [F,3], link_frame, link_offset [L,16],
    frame_index {link name: index}, joint_of_frame [joint name or None per frame], mimic {joint: (source joint,
    multiplier, offset)} for joint_vector)."""
    by_parent = {}
    for j in model.joints.values():
        by_parent.setdefault(j.parent, []).append(j)
    order, parent, jtype, origin, axis, jname = [], [], [], [], [], []
    index = {}
    stack = [(model.root_link(), None)]
    while stack:
        name, joint = stack.pop()
        index[name] = len(order)
        order.append(name)
        if joint is None:
            parent.append(-1); jtype.append(0); origin.append(Transform().opengl_matrix()); axis.append((1.0, 0.0, 0.0)); jname.append(None)
        else:
            parent.append(index[joint.parent])
            jtype.append(1 if joint.type in ("revolute", "continuous") else 2 if joint.type == "prismatic" else 0)
            origin.append(pose_to_transform(joint.xyz, joint.rpy).opengl_matrix())
            axis.append(tuple(float(a) for a in joint.axis))
            jname.append(joint.name)
        for j in sorted(by_parent.get(name, []), key=lambda jj: jj.name, reverse=True):
            stack.append((j.child, j))
    return {"parent": np.asarray(parent, np.int32), "joint_type": np.asarray(jtype, np.int32),
            "joint_origin": np.stack(origin), "joint_axis": np.asarray(axis, np.float64),
            "link_frame": np.asarray([index[n] for n in link_names], np.int32),
            "link_offset": np.stack([t.opengl_matrix() for t in link_offsets]) if link_offsets else np.zeros((0, 16)),
            "frame_index": index, "joint_of_frame": jname,
            "mimic": {j.name: j.mimic for j in model.joints.values() if j.mimic is not None}}


def joint_vector(kin, joint_positions):
    """q vector for rtuf_set_joint_positions (one entry per frame) from a {joint name: position} dict.  <mimic> joints
    are resolved here, on the host (multiplier * source + offset, chains followed): the device's forward kinematics
    takes one position per frame and needs no notion of mimicry."""
    q = dict(joint_positions)
    mimic = kin.get("mimic", {})

    def value(name, depth=0):
        if name in q:
            return float(q[name])
        if name not in mimic:
            return 0.0
        if depth > len(mimic):
            raise ValueError("mimic cycle through joint %r" % name)
        src, mult, off = mimic[name]
        q[name] = mult * value(src, depth + 1) + off
        return q[name]

    return np.asarray([value(j) if j else 0.0 for j in kin["joint_of_frame"]], np.float64)
