"""CPU: host-side logic above the C ABI -- URDF reader, tf-style algebra, forward kinematics,
primitive tessellation, STL reader, 16UC1 conversions, synthetic workloads, sharding."""
import math

import numpy as np
import pytest

from realtime_urdf_filter_amd import geometry as G
from realtime_urdf_filter_amd import sharding, urdf
from bench_support import synthetic, workloads
from realtime_urdf_filter_amd.filter import (URDFRenderer, depth_f32_to_u16, depth_u16_to_f32, RenderableBox)


def test_example_urdf_links_and_q1_boxes():
    m = urdf.Model.from_string(workloads.EXAMPLE_URDF)
    assert [l.name for l in m.get_links()] == ["wall1", "wall2", "world"]      # std::map order
    assert m.root_link() == "world"
    fk = urdf.forward_kinematics(m)
    assert np.allclose(fk["wall1"].origin, [0, 5, 0])
    c, s = math.cos(0.785398163), math.sin(0.785398163)
    assert np.allclose(fk["wall1"].basis, [[c, -s, 0], [s, c, 0], [0, 0, 1]], atol=1e-12)
    rd = URDFRenderer(workloads.EXAMPLE_URDF, "/EXAMPLE", "cam", "/world", urdf.StaticTransformProvider(), "visual", 1.0, [])
    assert len(rd.renderables_) == 2 and all(isinstance(r, RenderableBox) for r in rd.renderables_)
    assert rd.renderables_[0].name == "/EXAMPLE/wall1"
    d = rd.renderables_[0].draws
    assert len(d) == 2 and d[0].pre_op == 0 and d[1].pre_op == 1 and d[1].op == (4.0, 0.5, 2.0)
    # second box (quirk Q1): glutSolidCube(dimx) scaled by the dims -> extents dimx^2, dimx*dimy, dimx*dimz
    ext = np.abs(d[1].verts).max(0) * 2 * np.array(d[1].op)
    assert np.allclose(ext, [16.0, 2.0, 8.0])
    assert np.allclose(np.abs(d[0].verts).max(0) * 2, [4.0, 0.5, 2.0])


def test_geometry_type_scale_and_ignore():
    xml = workloads.EXAMPLE_URDF
    tf = urdf.StaticTransformProvider()
    assert len(URDFRenderer(xml, "", "c", "w", tf, "collision", 1.0, []).renderables_) == 2
    assert len(URDFRenderer(xml, "", "c", "w", tf, "visual", 1.0, ["wall1"]).renderables_) == 1
    r = URDFRenderer(xml, "", "c", "w", tf, "visual", 0.5, []).renderables_[0]
    assert (r.dimx, r.dimy, r.dimz) == (2.0, 0.25, 1.0)
    assert len(URDFRenderer(xml, "", "c", "w", tf, "bogus", 1.0, []).renderables_) == 0


def test_primitive_triangle_counts():
    assert [len(d.tris) for d in G.box_draws(1, 2, 3)] == [12, 12]
    assert len(G.sphere_draws(0.5)[0].tris) == 180
    assert len(G.cylinder_draws(0.5, 1.0)[0].tris) == 220
    cyl = G.cylinder_draws(0.5, 1.0)[0]
    assert cyl.pre_op == 2 and cyl.op == (0.0, 0.0, -0.5)
    v = G.sphere_draws(0.5)[0].verts
    assert np.allclose(np.linalg.norm(v, axis=1), 0.5, atol=1e-6)


def test_quad_and_strip_decomposition_order():
    assert G.quads_to_tris(1) == [(0, 1, 3), (1, 2, 3)]
    assert G.fan_to_tris(5) == [(0, 1, 2), (0, 2, 3), (0, 3, 4)]
    assert G.quad_strip_to_tris(6) == [(0, 1, 3), (2, 0, 3), (2, 3, 5), (4, 2, 5)]


def test_transform_algebra_matches_matrix_form():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4)
        a = urdf.Transform.from_quaternion(q, rng.normal(size=3))
        b = urdf.Transform.from_quaternion(rng.normal(size=4), rng.normal(size=3))
        assert np.allclose(a.basis @ a.basis.T, np.eye(3), atol=1e-12)
        ab = a * b
        p = rng.normal(size=3)
        assert np.allclose(ab * p, a * (b * p))
        ident = a * a.inverse()
        assert np.allclose(ident.basis, np.eye(3), atol=1e-12) and np.allclose(ident.origin, 0, atol=1e-12)
        q2 = a.get_rotation()
        a2 = urdf.Transform.from_quaternion(q2)
        assert np.allclose(a2.basis, a.basis, atol=1e-12)
        m = a.opengl_matrix().reshape(4, 4).T
        assert np.allclose(m[:3, :3], a.basis) and np.allclose(m[:3, 3], a.origin) and np.allclose(m[3], [0, 0, 0, 1])


def test_rpy_convention():
    t = urdf.pose_to_transform((1, 2, 3), (0, 0, math.pi / 2))
    assert np.allclose(t * np.array([1.0, 0, 0]), [1, 3, 3])
    t = urdf.pose_to_transform((0, 0, 0), (math.pi / 2, 0, 0))
    assert np.allclose(t * np.array([0.0, 1, 0]), [0, 0, 1])


def test_forward_kinematics_joint_types():
    xml = """<robot name="r"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j1" type="revolute"><origin xyz="1 0 0"/><parent link="a"/><child link="b"/><axis xyz="0 0 1"/><limit lower="-1" upper="1"/></joint>
      <joint name="j2" type="prismatic"><origin xyz="0 1 0"/><parent link="b"/><child link="c"/><axis xyz="1 0 0"/><limit lower="0" upper="1"/></joint>
      <joint name="j3" type="fixed"><origin xyz="0 0 1" rpy="0 0 1.5707963267948966"/><parent link="c"/><child link="d"/></joint></robot>"""
    m = urdf.Model.from_string(xml)
    fk = urdf.forward_kinematics(m, {"j1": math.pi / 2, "j2": 0.5})
    assert np.allclose(fk["b"].origin, [1, 0, 0])
    assert np.allclose(fk["c"].origin, [0, 0.5, 0])
    assert np.allclose(fk["d"].origin, [0.0, 0.5, 1.0])
    tf = urdf.StaticTransformProvider(fk)
    assert np.allclose(tf.lookup_transform("a", "d").origin, fk["d"].origin)
    with pytest.raises(KeyError):
        tf.lookup_transform("a", "nope")


def test_update_link_transforms_reuses_stale_transform_on_failure():
    """Quirk Q7 (src/urdf_renderer.cpp:173-190)."""
    tf = urdf.StaticTransformProvider({"/world": urdf.Transform(), "/P/wall1": urdf.Transform(None, (1, 2, 3))})
    rd = URDFRenderer(workloads.EXAMPLE_URDF, "/P", "cam", "/world", tf, "visual", 1.0, [])
    rd.update_link_transforms()
    assert np.allclose(rd.renderables_[0].link_to_fixed.origin, [1, 2, 3])
    assert np.allclose(rd.renderables_[1].link_to_fixed.origin, [1, 2, 3])      # wall2 lookup failed


def test_depth_conversions_16uc1():
    u = np.array([[0, 1, 999, 1000, 4500, 65535]], np.uint16)
    f = depth_u16_to_f32(u)
    assert f.dtype == np.float32 and f[0, 3] == np.float32(1000) * np.float32(0.001)
    back = depth_f32_to_u16(np.array([[0.0005, 0.0015, 0.0025, 5.0, np.nan, np.inf, -1.0, 70.0]], np.float32))
    assert list(back[0]) == [0, 2, 2, 5000, 0, 0, 0, 65535]             # half-to-even, saturate, NaN / inf -> 0


def test_stl_roundtrip_and_solid_header_quirk():
    v, t = synthetic.lumpy_ellipsoid(100, (0.1, 0.2, 0.3), 3)
    data = G.write_binary_stl(v, t, header=b"solid looks like ascii but is binary")
    v2, t2 = G.load_stl(data)
    assert len(t2) == len(t) and np.array_equal(v2[t2.reshape(-1)], v[t.reshape(-1)])
    ascii_stl = b"solid x\nfacet normal 0 0 1\nouter loop\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\nendloop\nendfacet\nendsolid x\n"
    v3, t3 = G.load_stl(ascii_stl)
    assert v3.shape == (3, 3) and t3.tolist() == [[0, 1, 2]]
    with pytest.raises(ValueError):
        G.load_stl(b"garbage")


def test_synthetic_is_deterministic():
    a = synthetic.SplitMix64(42)
    b = synthetic.SplitMix64(42)
    assert [a.next_u64() for _ in range(4)] == [b.next_u64() for _ in range(4)]
    assert synthetic.SplitMix64(0).next_u64() == 0xE220A8397B1DCDAF            # published splitmix64 vector
    d0, d1 = synthetic.sensor_depth(64, 48, 3), synthetic.sensor_depth(64, 48, 3)
    assert np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
    assert np.isnan(synthetic.sensor_depth(640, 480, 0)).sum() > 1000
    w1 = workloads.pr2_workload(2, 160, 120, total_triangles=3000)
    w2 = workloads.pr2_workload(2, 160, 120, total_triangles=3000)
    assert np.array_equal(w1.link_tf[0], w2.link_tf[0]) and np.array_equal(w1.cam_tf, w2.cam_tf)
    assert w1.meta["links_with_geometry"] == 51 and 2500 < w1.n_triangles() < 3600


def test_sharding_partitions_streams():
    for n, w in ((256, 8), (513, 8), (7, 3), (2, 4)):
        seen = []
        for r in range(w):
            first, cnt = sharding.shard_range(n, w, r)
            seen += list(range(first, first + cnt))
        assert seen == list(range(n))
    assert sharding.models_for_rank(64, 8, 3) == [3, 11, 19, 27, 35, 43, 51, 59]
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_cpp_facade_parses_reference_style_urdf(tmp_path):
    """The C++ host layer (include/realtime_urdf_filter_amd/host.hpp): tolerant XML reader (the
    reference's example URDF has stray '>' characters after </visual>), primitive tessellation."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    if not os.path.exists(exe):
        subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    xml = """<?xml version="1.0"?><!-- c --><robot name="x">
      <link name="world"/>
      <link name="a"><visual><origin xyz="0 0 1" rpy="0 0 0"/><geometry><box size="4 0.5 2" /></geometry></visual>>
        <collision><geometry><box size="1 1 1"/></geometry></collision>></link>
      <link name="b"><visual><geometry><sphere radius="0.5"/></geometry></visual>
                     <visual><geometry><cylinder radius="0.2" length="1"/></geometry></visual></link>
      <joint name="j" type="fixed"><origin xyz="0 5 0"/><parent link="world"/><child link="a"/></joint>
      <joint name="k" type="revolute"><parent link="a"/><child link="b"/><axis xyz="0 0 1"/><limit lower="-1" upper="1" effort="1" velocity="1"/></joint>
    </robot>"""
    f = tmp_path / "m.urdf"
    f.write_text(xml)
    out = subprocess.check_output([exe, "--parse", str(f)]).decode()
    assert out.strip() == "renderables=3 draws=4 triangles=%d first=/P/a" % (24 + 180 + 220)


def test_kinematic_arrays_reproduce_forward_kinematics():
    """The FK tree handed to the GPU (urdf.kinematic_arrays) evaluates to the same transforms as
    urdf.forward_kinematics when multiplied top-down in numpy."""
    robot = synthetic.SyntheticRobot(2000, 3)
    model = urdf.Model.from_string(robot.to_urdf_xml())
    q = robot.random_joint_state(5)
    fk = urdf.forward_kinematics(model, q)
    names = sorted(model.links)
    kin = urdf.kinematic_arrays(model, names, [urdf.Transform() for _ in names])
    qv = urdf.joint_vector(kin, q)
    assert kin["parent"][0] == -1 and all(kin["parent"][i] < i for i in range(1, len(kin["parent"])))
    T = {}
    for i in range(len(kin["parent"])):
        p = kin["parent"][i]
        if p < 0:
            T[i] = np.eye(4)
            continue
        O4 = kin["joint_origin"][i].reshape(4, 4).T
        M = np.eye(4)
        if kin["joint_type"][i] == 1:
            ax = kin["joint_axis"][i] / np.linalg.norm(kin["joint_axis"][i])
            h = 0.5 * qv[i]
            M[:3, :3] = urdf.Transform.from_quaternion((ax[0] * math.sin(h), ax[1] * math.sin(h), ax[2] * math.sin(h), math.cos(h))).basis
        elif kin["joint_type"][i] == 2:
            M[:3, 3] = kin["joint_axis"][i] * qv[i]
        T[i] = T[p] @ O4 @ M
    for name, idx in kin["frame_index"].items():
        assert np.allclose(T[idx][:3, :3], fk[name].basis, atol=1e-12) and np.allclose(T[idx][:3, 3], fk[name].origin, atol=1e-12)


def test_mimic_joints_follow_their_source_on_host_and_in_the_device_vector():
    """<mimic joint multiplier offset> (the PR2's gripper fingers): resolved on the host, for the host FK and for the
    joint vector handed to the GPU's forward kinematics; chains are followed, explicit positions win, cycles raise."""
    xml = """<robot name="g"><link name="palm"/><link name="f1"/><link name="f2"/><link name="tip"/>
      <joint name="drive" type="revolute"><origin xyz="0 0.1 0"/><parent link="palm"/><child link="f1"/><axis xyz="0 0 1"/></joint>
      <joint name="follow" type="revolute"><origin xyz="0 -0.1 0"/><parent link="palm"/><child link="f2"/><axis xyz="0 0 1"/>
        <mimic joint="drive" multiplier="-1" offset="0.25"/></joint>
      <joint name="tipj" type="prismatic"><origin xyz="0.1 0 0"/><parent link="f2"/><child link="tip"/><axis xyz="1 0 0"/>
        <mimic joint="follow" multiplier="2"/></joint></robot>"""
    m = urdf.Model.from_string(xml)
    assert m.joints["follow"].mimic == ("drive", -1.0, 0.25) and m.joints["tipj"].mimic == ("follow", 2.0, 0.0)
    q = urdf.resolve_mimic(m, {"drive": 0.4})
    assert abs(q["follow"] - (-0.15)) < 1e-15 and abs(q["tipj"] - (-0.3)) < 1e-15
    fk = urdf.forward_kinematics(m, {"drive": 0.4})
    fk_explicit = urdf.forward_kinematics(m, {"drive": 0.4, "follow": q["follow"], "tipj": q["tipj"]})
    for name in fk:
        assert np.array_equal(fk[name].basis, fk_explicit[name].basis) and np.array_equal(fk[name].origin, fk_explicit[name].origin)
    assert not np.allclose(fk["f2"].basis, np.eye(3))                       # the follower really moved
    kin = urdf.kinematic_arrays(m, sorted(m.links), [urdf.Transform() for _ in m.links])
    qv = urdf.joint_vector(kin, {"drive": 0.4})
    by_joint = dict(zip(kin["joint_of_frame"], qv))
    assert by_joint["drive"] == 0.4 and by_joint["follow"] == q["follow"] and by_joint["tipj"] == q["tipj"]
    assert dict(zip(kin["joint_of_frame"], urdf.joint_vector(kin, {"drive": 0.4, "follow": 1.0})))["tipj"] == 2.0     # explicit wins
    cyc = urdf.Model.from_string(xml.replace('<axis xyz="0 0 1"/></joint>\n      <joint name="follow"', '<axis xyz="0 0 1"/><mimic joint="tipj"/></joint>\n      <joint name="follow"'))
    with pytest.raises(ValueError):
        urdf.resolve_mimic(cyc, {})


def test_package_resolver_finds_stl_meshes(tmp_path):
    from realtime_urdf_filter_amd import geometry as G
    pkg = tmp_path / "ws" / "my_robot_description"
    (pkg / "meshes").mkdir(parents=True)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    (pkg / "meshes" / "part.stl").write_bytes(G.write_binary_stl(v, t, header=b"solid looks like ascii"))
    (pkg / "meshes" / "part.dae").write_text("<COLLADA/>")
    for roots in ([str(tmp_path / "ws")], [str(pkg)]):               # root contains the package / root is the package
        r = G.PackageResolver(roots)
        vv, tt = r("package://my_robot_description/meshes/part.stl")
        assert vv.shape == (6, 3) and tt.shape == (2, 3)
    vv, tt = G.PackageResolver([])("file://" + str(pkg / "meshes" / "part.stl"))
    assert tt.shape == (2, 3)
    with pytest.raises(IOError):
        G.PackageResolver([str(tmp_path / "ws")])("package://other_pkg/meshes/part.stl")
    with pytest.raises(ValueError):
        G.PackageResolver([str(tmp_path / "ws")])("package://my_robot_description/meshes/part.dae")     # a Collada file without geometry


def test_cpp_host_forward_kinematics_with_mimic_joints_matches_python(tmp_path):
    """The C++ twin (include/realtime_urdf_filter_amd/host.hpp) resolves <mimic> joints like the Python mirror."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    xml = """<robot name="g"><link name="palm"/><link name="f1"/><link name="f2"/><link name="tip"/>
      <joint name="drive" type="revolute"><origin xyz="0 0.1 0" rpy="0.1 0.2 0.3"/><parent link="palm"/><child link="f1"/><axis xyz="0 0 1"/></joint>
      <joint name="follow" type="revolute"><origin xyz="0 -0.1 0"/><parent link="palm"/><child link="f2"/><axis xyz="0 1 1"/>
        <mimic joint="drive" multiplier="-1" offset="0.25"/></joint>
      <joint name="tipj" type="prismatic"><origin xyz="0.1 0 0"/><parent link="f2"/><child link="tip"/><axis xyz="1 0 0"/>
        <mimic joint="follow" multiplier="2"/></joint></robot>"""
    f = tmp_path / "g.urdf"
    f.write_text(xml)
    out = subprocess.check_output([exe, "--fk", str(f), "drive=0.4"]).decode().strip().splitlines()
    fk = urdf.forward_kinematics(urdf.Model.from_string(xml), {"drive": 0.4})
    assert len(out) == 4
    for line in out:
        parts = line.split()
        t = fk[parts[0]]
        v = np.array([float(x) for x in parts[1:]])
        assert np.allclose(v[:9].reshape(3, 3), t.basis, atol=1e-15) and np.allclose(v[9:], t.origin, atol=1e-15), parts[0]
    assert not np.allclose(fk["tip"].origin, urdf.forward_kinematics(urdf.Model.from_string(xml), {})["tip"].origin)


# ---------------------------------------------------------------------------------------------------------------------
# Collada / OBJ import (meshes.py and its C++ twin in host.hpp): the files below are written for this test
# ---------------------------------------------------------------------------------------------------------------------
_DAE = """<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="inch" meter="0.0254"/><up_axis>%(up)s</up_axis></asset>
  <library_geometries>
    <geometry id="plate"><mesh>
      <source id="plate-pos"><float_array id="plate-pos-array" count="12">0 0 0  2 0 0  2 1 0  0 1 0.5</float_array>
        <technique_common><accessor source="#plate-pos-array" count="4" stride="3"><param name="X" type="float"/><param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common></source>
      <source id="plate-nrm"><float_array id="plate-nrm-array" count="3">0 0 1</float_array>
        <technique_common><accessor source="#plate-nrm-array" count="1" stride="3"/></technique_common></source>
      <vertices id="plate-vtx"><input semantic="POSITION" source="#plate-pos"/></vertices>
      <polylist count="1" material="m"><input semantic="VERTEX" source="#plate-vtx" offset="0"/><input semantic="NORMAL" source="#plate-nrm" offset="1"/>
        <vcount>4</vcount><p>0 0 1 0 2 0 3 0</p></polylist>
    </mesh></geometry>
    <geometry id="tri"><mesh>
      <source id="tri-pos"><float_array id="tri-pos-array" count="9">0.25 0 0 0 0.5 0 0 0 0.75</float_array>
        <technique_common><accessor source="#tri-pos-array" count="3" stride="3"/></technique_common></source>
      <vertices id="tri-vtx"><input semantic="POSITION" source="#tri-pos"/></vertices>
      <triangles count="1"><input semantic="VERTEX" source="#tri-vtx" offset="0"/><p>0 1 2</p></triangles>
    </mesh></geometry>
    <geometry id="misc"><mesh>
      <source id="misc-pos"><float_array id="misc-pos-array" count="18">0 0 0  1 0 0  1 1 0  0 1 0  2 0 1  2 1 1</float_array>
        <technique_common><accessor source="#misc-pos-array" count="6" stride="3"/></technique_common></source>
      <vertices id="misc-vtx"><input semantic="POSITION" source="#misc-pos"/></vertices>
      <tristrips count="1"><input semantic="VERTEX" source="#misc-vtx" offset="0"/><p>0 1 3 2 5</p></tristrips>
      <trifans count="1"><input semantic="VERTEX" source="#misc-vtx" offset="0"/><p>1 4 5 2</p></trifans>
      <polygons count="2"><input semantic="VERTEX" source="#misc-vtx" offset="0"/><p>0 1 2 3</p><p>1 4 5</p></polygons>
    </mesh></geometry>
  </library_geometries>
  <library_nodes><node id="shared"><translate>0 0 10</translate><instance_geometry url="#tri"/></node></library_nodes>
  <library_visual_scenes><visual_scene id="Scene">
    <node id="a"><matrix>1 0 0 1  0 1 0 2  0 0 1 3  0 0 0 1</matrix><instance_geometry url="#plate"/>
      <node id="b"><rotate>0 0 1 90</rotate><scale>2 2 2</scale><instance_geometry url="#tri"/></node></node>
    <node id="c"><instance_node url="#shared"/></node>
  </visual_scene></library_visual_scenes>
  <scene><instance_visual_scene url="#Scene"/></scene>
</COLLADA>
"""

_OBJ = """# a quad and a triangle
mtllib none.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vn 0 0 1
vt 0 0
f 1/1/1 2/1/1 3/1/1 4/1/1
v 0 0 2
f -1 -4//1 -3
l 1 2
"""


def _expected_dae(up_axis_to_y, apply_unit, up="Z_UP"):
    plate = np.array([[0, 0, 0], [2, 0, 0], [2, 1, 0], [0, 1, 0.5]], np.float64)
    tri = np.array([[0.25, 0, 0], [0, 0.5, 0], [0, 0, 0.75]], np.float64)
    a = np.eye(4); a[:3, 3] = [1, 2, 3]
    rot = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    b = a @ rot @ np.diag([2.0, 2, 2, 1])
    c = np.eye(4); c[2, 3] = 10
    rootm = np.eye(4)
    if up_axis_to_y and up == "Z_UP":
        rootm = rootm @ np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float64)
    if apply_unit:
        rootm = rootm @ np.diag([0.0254, 0.0254, 0.0254, 1])
    def tf(m, p):
        return (rootm @ m @ np.c_[p, np.ones(len(p))].T).T[:, :3]
    return np.concatenate([tf(a, plate[[0, 1, 2, 0, 2, 3]]), tf(b, tri), tf(c, tri)])


def test_collada_import_node_transforms_up_axis_unit_and_polylist():
    from realtime_urdf_filter_amd import meshes
    for up_to_y, unit in ((True, False), (False, False), (True, True)):
        v, t = meshes.load_collada((_DAE % {"up": "Z_UP"}).encode(), up_axis_to_y=up_to_y, apply_unit=unit)
        assert v.dtype == np.float32 and v.shape == (12, 3) and np.array_equal(t, np.arange(12).reshape(4, 3))
        np.testing.assert_allclose(v, _expected_dae(up_to_y, unit), rtol=0, atol=2e-6)
    # the reference's behaviour for a Z_UP file (Assimp rotates the root to Y_UP): (x, y, z) -> (x, z, -y)
    vz, _ = meshes.load_collada((_DAE % {"up": "Z_UP"}).encode(), up_axis_to_y=False)
    vy, _ = meshes.load_collada((_DAE % {"up": "Z_UP"}).encode(), up_axis_to_y=True)
    np.testing.assert_array_equal(vy, np.stack([vz[:, 0], vz[:, 2], -vz[:, 1]], axis=1))
    vyy, _ = meshes.load_collada((_DAE % {"up": "Y_UP"}).encode(), up_axis_to_y=True)
    np.testing.assert_array_equal(vyy, vz)
    with pytest.raises(ValueError):
        meshes.load_collada(b"<robot/>")
    with pytest.raises(ValueError):
        meshes.load_collada(b"<COLLADA><library_geometries/></COLLADA>")


def test_obj_import_polygons_relative_indices_and_dispatch(tmp_path):
    from realtime_urdf_filter_amd import meshes
    v, t = meshes.load_obj(_OBJ.encode())
    exp = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 2], [1, 0, 0], [1, 1, 0]], np.float32)
    np.testing.assert_array_equal(v, exp)
    assert np.array_equal(t, np.arange(9).reshape(3, 3))
    with pytest.raises(ValueError):
        meshes.load_obj(b"v 0 0 0\nf 1 2 3\n")
    assert meshes.mesh_format("a/b/link.DAE", b"") == "collada" and meshes.mesh_format("x.obj", b"") == "obj" and meshes.mesh_format("x.STL", b"") == "stl"
    assert meshes.mesh_format("noext", (_DAE % {"up": "Y_UP"}).encode()) == "collada" and meshes.mesh_format("noext", _OBJ.encode()) == "obj"
    # through the resolver, all three formats
    pkg = tmp_path / "ws" / "robot_description"
    (pkg / "meshes").mkdir(parents=True)
    (pkg / "meshes" / "l.dae").write_text(_DAE % {"up": "Z_UP"})
    (pkg / "meshes" / "l.obj").write_text(_OBJ)
    (pkg / "meshes" / "l.stl").write_bytes(G.write_binary_stl(exp, np.arange(9).reshape(3, 3)))
    r = G.PackageResolver([str(tmp_path / "ws")])
    assert r("package://robot_description/meshes/l.dae")[0].shape == (12, 3)
    np.testing.assert_array_equal(r("package://robot_description/meshes/l.obj")[0], exp)
    np.testing.assert_array_equal(r("package://robot_description/meshes/l.stl")[0], exp)
    keep = G.PackageResolver([str(tmp_path / "ws")], up_axis_to_y=False)
    assert not np.array_equal(keep("package://robot_description/meshes/l.dae")[0], r("package://robot_description/meshes/l.dae")[0])


def test_cpp_mesh_import_matches_python_bit_for_bit(tmp_path):
    """host.hpp load_mesh (what the C++ RenderableMesh calls) against meshes.py on the same files."""
    import os
    import subprocess
    from realtime_urdf_filter_amd import meshes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    cases = []
    for up in ("Z_UP", "Y_UP", "X_UP"):
        f = tmp_path / ("m_%s.dae" % up)
        f.write_text(_DAE % {"up": up})
        cases += [(f, [], dict(up_axis_to_y=True, apply_unit=False)), (f, ["no-up"], dict(up_axis_to_y=False, apply_unit=False)),
                  (f, ["unit"], dict(up_axis_to_y=True, apply_unit=True))]
    fo = tmp_path / "m.obj"
    fo.write_text(_OBJ)
    cases.append((fo, [], {}))
    fs = tmp_path / "m.stl"
    fs.write_bytes(G.write_binary_stl(np.random.default_rng(3).standard_normal((9, 3)).astype(np.float32), np.arange(9).reshape(3, 3), header=b"solid quirk"))
    cases.append((fs, [], {}))
    for path, flags, kw in cases:
        out = subprocess.check_output([exe, "--mesh", str(path)] + flags).decode().split()
        nv, nt = int(out[0]), int(out[1])
        got = np.array([int(h, 16) for h in out[2:]], np.uint32).view(np.float32).reshape(-1, 3)
        v, t = meshes.load_mesh(str(path), path.read_bytes(), **kw)
        assert (nv, nt) == (len(v), len(t))
        np.testing.assert_array_equal(got.view(np.uint32), v.view(np.uint32), err_msg="%s %s" % (path, flags))
    bad = tmp_path / "bad.dae"
    bad.write_text("<COLLADA><library_geometries/></COLLADA>")
    assert subprocess.run([exe, "--mesh", str(bad)], stdout=subprocess.PIPE).returncode == 1


def test_collada_strips_fans_polygons_python_and_cpp(tmp_path):
    import os
    import subprocess
    from realtime_urdf_filter_amd import meshes
    doc = (_DAE % {"up": "Y_UP"}).replace('<node id="c">', '<node id="d"><translate>0 0 -5</translate><instance_geometry url="#misc"/></node><node id="c">')
    v, t = meshes.load_collada(doc.encode())
    assert len(t) == 4 + 8
    P = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 1], [2, 1, 1]], np.float32) + np.float32([0, 0, -5])
    want = [(0, 1, 3), (3, 1, 2), (3, 2, 5), (1, 4, 5), (1, 5, 2), (0, 1, 2), (0, 2, 3), (1, 4, 5)]
    # node "d" comes after "a" (6 + 3 corners) and before "c" (3 corners) in document order
    np.testing.assert_array_equal(v[9:9 + 24], P[np.array(want).reshape(-1)])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    f = tmp_path / "misc.dae"
    f.write_text(doc)
    out = subprocess.check_output([exe, "--mesh", str(f)]).decode().split()
    got = np.array([int(h, 16) for h in out[2:]], np.uint32).view(np.float32).reshape(-1, 3)
    np.testing.assert_array_equal(got.view(np.uint32), v.view(np.uint32))


def test_renderable_mesh_defaults_to_the_package_path_resolver(tmp_path, monkeypatch):
    """Without an explicit mesh_loader the Python mirror resolves package:// against ROS_PACKAGE_PATH, as
    resource_retriever does for the reference (src/renderable.cpp:270-300); an unresolvable mesh draws nothing (Q15)."""
    from realtime_urdf_filter_amd.filter import RenderableMesh
    pkg = tmp_path / "ws" / "arm_description"
    (pkg / "meshes").mkdir(parents=True)
    (pkg / "meshes" / "link.obj").write_text(_OBJ)
    monkeypatch.setenv("ROS_PACKAGE_PATH", str(tmp_path / "ws"))
    r = RenderableMesh("package://arm_description/meshes/link.obj", 1.0, 2.0, 3.0)
    assert len(r.draws) == 1 and len(r.draws[0].tris) == 3 and tuple(r.draws[0].op) == (1.0, 2.0, 3.0)
    assert RenderableMesh("package://missing_pkg/meshes/link.obj", 1.0, 1.0, 1.0).draws == []


def test_cpp_facade_resolves_package_uris_by_default(tmp_path):
    """RenderableMesh of the C++ façade without a MeshResolver: package:// against ROS_PACKAGE_PATH (--parse prints what
    would be uploaded)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "bin", "example_filter")
    subprocess.check_call([os.path.join(root, "realtime_urdf_filter_amd", "csrc", "build_facade.sh")])
    pkg = tmp_path / "ws" / "arm_description"
    (pkg / "meshes").mkdir(parents=True)
    (pkg / "meshes" / "link.obj").write_text(_OBJ)
    (pkg / "meshes" / "link.dae").write_text(_DAE % {"up": "Z_UP"})
    f = tmp_path / "r.urdf"
    f.write_text("""<robot name="r"><link name="a"><visual><geometry><mesh filename="package://arm_description/meshes/link.obj"/></geometry></visual></link>
      <link name="b"><visual><geometry><mesh filename="package://arm_description/meshes/link.dae" scale="2 2 2"/></geometry></visual></link>
      <link name="c"><visual><geometry><mesh filename="package://nowhere/meshes/link.stl"/></geometry></visual></link>
      <joint name="j" type="fixed"><parent link="a"/><child link="b"/></joint><joint name="k" type="fixed"><parent link="a"/><child link="c"/></joint></robot>""")
    env = dict(os.environ, ROS_PACKAGE_PATH="/nonexistent:" + str(tmp_path / "ws"))
    out = subprocess.check_output([exe, "--parse", str(f)], env=env, stderr=subprocess.DEVNULL).decode().strip()
    assert out.startswith("renderables=3 draws=2 triangles=%d " % (3 + 4)), out


def test_use_own_calibration_switch():
    """The reference's compile-time USE_OWN_CALIBRATION (src/urdf_filter.cpp:38, :462-472) as a run-time parameter of the host
    mirror: the CameraInfo's P is ignored (width / height still count), the hard-coded intrinsics -- floats in the reference --
    are used, and camera_tx_ / camera_ty_ keep their values (the #ifdef skips their assignment)."""
    from realtime_urdf_filter_amd.filter import CameraInfo, FilterParameters, RealtimeURDFFilter
    import realtime_urdf_filter_amd as R
    info = CameraInfo(640, 480, [525.0, 0, 319.5, -39.4, 0, 526.0, 239.5, 2.0, 0, 0, 1, 0])
    f = RealtimeURDFFilter(FilterParameters("/world", "/cam", [], 0.05), None)
    P = np.asarray(f.getProjectionMatrix(info))
    assert f.camera_tx_ == 39.4 / 525.0 and f.camera_ty_ == -2.0 / 526.0
    own = RealtimeURDFFilter(FilterParameters("/world", "/cam", [], 0.05, use_own_calibration=True), None)
    own.camera_tx_ = 0.25
    Q = np.asarray(own.getProjectionMatrix(info))
    fx, fy, cx, cy = (float(np.float32(v)) for v in (585.260, 585.028, 317.387, 239.264))
    want, _, _ = R.projection_from_intrinsics(fx, fy, cx, cy, 640, 480, 0.1, 8.0)
    assert np.array_equal(Q, np.asarray(want)) and not np.array_equal(Q, P) and own.camera_tx_ == 0.25 and own.camera_ty_ == 0.0
    assert Q[0] == -2.0 * fx / 640 and Q[5] == 2.0 * fy / 480
    d = FilterParameters.from_dict({"fixed_frame": "/w", "camera_frame": "/c", "depth_distance_threshold": 0.1, "use_own_calibration": True,
                                    "own_calibration": [500.0, 501.0, 320.0, 240.0]})
    assert d.use_own_calibration and d.own_calibration == (500.0, 501.0, 320.0, 240.0)
