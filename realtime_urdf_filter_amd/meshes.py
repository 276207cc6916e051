"""Mesh import for URDF <mesh filename=...> geometry: STL, Collada (.dae) and Wavefront OBJ.

Replaces, for the hot path's one-time geometry load, the Assimp import of the reference:
src/renderable.cpp:306-322 reads the file with
    aiProcess_PreTransformVertices | SortByPType | GenNormals | Triangulate | GenUVCoords | FlipUVs
and src/renderable.cpp:352-415 (`fromAssimpScene` / `initMesh`) takes, from every aiMesh of the scene, the vertex
positions as they are and three indices per face.  Node transformations are NOT applied by the reference itself (the
code that would is commented out, :389-393), so what reaches the GPU is exactly what PreTransformVertices baked in.

Assimp is a third-party dependency that is absent from /root/reference and from this image, and the reference does not
pin its version (package.xml has no versions; README.md:29 names Ubuntu 14.04, i.e. Assimp 3.0).  What is restated here
is its published behaviour for these formats [recall, parity unpinned -- no Assimp to run against]:

  * every face corner becomes its own vertex (no JoinIdenticalVertices in the flag set): verts = [3T, 3], tris = arange;
  * polygons are triangulated as a fan from their first corner (what Assimp's TriangulateProcess produces for convex
    polygons; concave polygons go through its ear clipper and may be split differently -- they cover the same pixels
    unless the polygon is non-planar);
  * PreTransformVertices multiplies every vertex by the absolute transformation of the node that instantiates its
    mesh, root node included, in float32;
  * the Collada importer converts the file's <up_axis> to Y_UP by a rotation of the ROOT node (Z_UP: (x, y, z) ->
    (x, z, -y); X_UP: (x, y, z) -> (-y, x, z)), which PreTransformVertices then bakes into the vertices.  This is why
    Z_UP Collada meshes come out rotated in the reference (its FIXME at src/renderable.cpp:386-388); `up_axis_to_y=True`
    reproduces that, `False` leaves the file's axes alone (what a robot model author would expect);
  * <unit meter="..."> is read but not applied by Assimp 3.x (`apply_unit=False`, the default); Assimp >= 4.1 scales the
    root node by it (`apply_unit=True`).

Draw order inside one mesh file (Assimp regroups meshes by material) is not reproduced: all triangles of a file are one
draw call in file order.  Order only decides which of two fragments with EQUAL 24-bit depth wins, and both then give the
same virtual depth except for window z <= 0.5, where float z is finer than 24 bits.
"""
import math
import re
import xml.etree.ElementTree as ET

import numpy as np

from .geometry import load_stl

f32 = np.float32


def _deindex(points, corners):
    """points [P,3], corners: flat list of point indices, three per triangle -> (verts [3T,3] f32, tris [T,3] u32)."""
    idx = np.asarray(corners, np.int64)
    v = np.asarray(points, f32).reshape(-1, 3)[idx] if len(idx) else np.zeros((0, 3), f32)
    return np.ascontiguousarray(v, f32), np.arange(len(idx), dtype=np.uint32).reshape(-1, 3)


def _fan(poly):
    out = []
    for k in range(1, len(poly) - 1):
        out += [poly[0], poly[k], poly[k + 1]]
    return out


# ------------------------------------------------------------------------------------------------------------------
# Wavefront OBJ
# ------------------------------------------------------------------------------------------------------------------
def load_obj(data):
    """bytes/str -> (verts, tris).  `v x y z [w]`, `f a b c ...` with `a`, `a/t`, `a/t/n`, `a//n`, negative (relative)
    indices and polygons (fan); lines, points, materials, groups, texture coordinates and normals are ignored
    (SortByPType puts points and lines in meshes of their own; `initMesh` asserts three indices per face, so a file
    with such primitives aborts a debug build of the reference -- here they are skipped)."""
    text = data.decode("utf-8", errors="replace") if isinstance(data, (bytes, bytearray)) else data
    text = text.replace("\\\n", " ")
    pts, corners = [], []
    for line in text.splitlines():
        p = line.split("#", 1)[0].split()
        if not p:
            continue
        if p[0] == "v" and len(p) >= 4:
            pts.append((float(p[1]), float(p[2]), float(p[3])))
        elif p[0] == "f" and len(p) >= 4:
            poly = []
            for tok in p[1:]:
                i = int(tok.split("/")[0])
                i = i - 1 if i > 0 else len(pts) + i
                if not 0 <= i < len(pts):
                    raise ValueError("OBJ: face index %s out of range" % tok)
                poly.append(i)
            corners += _fan(poly)
    if not pts:
        raise ValueError("OBJ: no vertices")
    return _deindex(pts, corners)


# ------------------------------------------------------------------------------------------------------------------
# Collada 1.4 / 1.5
# ------------------------------------------------------------------------------------------------------------------
def _strip_ns(root):
    for e in root.iter():
        if isinstance(e.tag, str) and "}" in e.tag:
            e.tag = e.tag.split("}", 1)[1]
    return root


def _floats(text):
    return [float(x) for x in (text or "").split()]


def _mat_mul(a, b):
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def _node_matrix(node):
    """Product of the node's transformation elements in document order (Collada 1.4 spec, section 5 'node')."""
    m = np.eye(4, dtype=f32)
    for e in node:
        t = e.tag
        if t == "matrix":
            v = _floats(e.text)
            if len(v) == 16:
                m = _mat_mul(m, np.asarray(v, f32).reshape(4, 4))          # row-major in the file
        elif t == "translate":
            v = _floats(e.text)
            k = np.eye(4, dtype=f32)
            k[:3, 3] = v[:3]
            m = _mat_mul(m, k)
        elif t == "scale":
            v = _floats(e.text)
            m = _mat_mul(m, np.diag(np.asarray(v[:3] + [1.0], f32)))
        elif t == "rotate":
            v = _floats(e.text)
            ax = np.asarray(v[:3], np.float64)
            n = np.linalg.norm(ax)
            if n > 0:
                x, y, z = ax / n
                a = math.radians(v[3])
                c, s, o = math.cos(a), math.sin(a), 1.0 - math.cos(a)
                k = np.eye(4, dtype=f32)
                k[:3, :3] = np.asarray([[c + x * x * o, x * y * o - z * s, x * z * o + y * s],
                                        [y * x * o + z * s, c + y * y * o, y * z * o - x * s],
                                        [z * x * o - y * s, z * y * o + x * s, c + z * z * o]], f32)
                m = _mat_mul(m, k)
        # lookat / skew: not used by robot description meshes; ignored
    return m


def _read_geometry(geom):
    """<geometry><mesh> -> (points [P,3] f32, corner indices of the triangulated primitives)."""
    mesh = geom.find("mesh")
    if mesh is None:
        return None                                      # convex_mesh / spline: nothing Assimp would turn into triangles here
    sources = {}
    for src in mesh.findall("source"):
        fa = src.find("float_array")
        if fa is None:
            continue
        vals = np.asarray(_floats(fa.text), f32)
        acc = src.find("technique_common/accessor")
        stride = int(acc.get("stride", "3")) if acc is not None else 3
        offset = int(acc.get("offset", "0")) if acc is not None else 0
        count = int(acc.get("count", str((len(vals) - offset) // stride))) if acc is not None else (len(vals) - offset) // stride
        arr = vals[offset:offset + count * stride].reshape(count, stride)
        sources[src.get("id")] = arr
    positions = None
    vertices_id = None
    vtx = mesh.find("vertices")
    if vtx is not None:
        vertices_id = vtx.get("id")
        for inp in vtx.findall("input"):
            if inp.get("semantic") == "POSITION":
                positions = sources.get(inp.get("source", "").lstrip("#"))
    if positions is None:
        return None
    pts = np.zeros((len(positions), 3), f32)
    pts[:, :min(3, positions.shape[1])] = positions[:, :3]
    corners = []
    for prim in mesh:
        if prim.tag not in ("triangles", "polylist", "polygons", "trifans", "tristrips"):
            continue
        inputs = prim.findall("input")
        stride = max([int(i.get("offset", "0")) for i in inputs] + [0]) + 1
        voff = None
        for i in inputs:
            if i.get("semantic") == "VERTEX" and i.get("source", "").lstrip("#") == vertices_id:
                voff = int(i.get("offset", "0"))
        if voff is None:
            continue
        def vertex_indices(text):
            p = [int(x) for x in (text or "").split()]
            return p[voff::stride]
        if prim.tag == "triangles":
            p = prim.find("p")
            idx = vertex_indices(p.text if p is not None else "")
            corners += idx[:len(idx) // 3 * 3]
        elif prim.tag == "polylist":
            p, vc = prim.find("p"), prim.find("vcount")
            idx = vertex_indices(p.text if p is not None else "")
            at = 0
            for n in [int(x) for x in ((vc.text if vc is not None else "") or "").split()]:
                if n >= 3:
                    corners += _fan(idx[at:at + n])
                at += n
        elif prim.tag == "polygons":
            for p in prim.findall("p"):
                idx = vertex_indices(p.text)
                if len(idx) >= 3:
                    corners += _fan(idx)
        elif prim.tag == "trifans":
            for p in prim.findall("p"):
                idx = vertex_indices(p.text)
                corners += _fan(idx)
        elif prim.tag == "tristrips":
            for p in prim.findall("p"):
                idx = vertex_indices(p.text)
                for k in range(len(idx) - 2):
                    corners += [idx[k], idx[k + 1], idx[k + 2]] if k % 2 == 0 else [idx[k + 1], idx[k], idx[k + 2]]
    if corners and (min(corners) < 0 or max(corners) >= len(pts)):
        raise ValueError("Collada: vertex index out of range in geometry %r" % geom.get("id"))
    return pts, corners


def load_collada(data, up_axis_to_y=True, apply_unit=False):
    """bytes/str -> (verts [3T,3] f32, tris [T,3] u32): every <instance_geometry> reachable from the instantiated
    visual scene, transformed by its node chain (and by the root adjustments described in the module docstring)."""
    text = data.decode("utf-8", errors="replace") if isinstance(data, (bytes, bytearray)) else data
    root = _strip_ns(ET.fromstring(text))
    if root.tag != "COLLADA":
        raise ValueError("not a Collada document")
    up = (root.findtext("asset/up_axis") or "Y_UP").strip().upper()
    unit = root.find("asset/unit")
    meter = float(unit.get("meter", "1")) if unit is not None else 1.0
    geoms = {g.get("id"): g for g in root.findall("library_geometries/geometry")}
    lib_nodes = {}
    for n in root.findall("library_nodes//node"):
        if n.get("id"):
            lib_nodes[n.get("id")] = n
    scenes = {s.get("id"): s for s in root.findall("library_visual_scenes/visual_scene")}
    inst = root.find("scene/instance_visual_scene")
    scene = scenes.get(inst.get("url", "").lstrip("#")) if inst is not None else None
    if scene is None and scenes:
        scene = next(iter(scenes.values()))
    rootm = np.eye(4, dtype=f32)
    if up_axis_to_y and up == "Z_UP":
        rootm = _mat_mul(rootm, np.asarray([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], f32))
    elif up_axis_to_y and up == "X_UP":
        rootm = _mat_mul(rootm, np.asarray([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], f32))
    if apply_unit:
        rootm = _mat_mul(rootm, np.diag(np.asarray([meter, meter, meter, 1.0], f32)))
    cache = {}
    out = []

    def emit(gid, m):
        if gid not in cache:
            g = geoms.get(gid)
            cache[gid] = _read_geometry(g) if g is not None else None
        got = cache[gid]
        if not got or not got[1]:
            return
        pts, corners = got
        # aiMatrix4x4 * aiVector3D in float32: x' = a1 x + a2 y + a3 z + a4, left to right
        x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
        tp = np.stack([(m[r, 0] * x + m[r, 1] * y + m[r, 2] * z + m[r, 3]).astype(f32) for r in range(3)], axis=1)
        out.append(tp[np.asarray(corners, np.int64)])

    def walk(node, m, depth=0):
        if depth > 64:
            raise ValueError("Collada: node hierarchy too deep (cyclic instance_node?)")
        m = _mat_mul(m, _node_matrix(node))
        for e in node:
            if e.tag == "instance_geometry":
                emit(e.get("url", "").lstrip("#"), m)
            elif e.tag == "instance_node":
                target = lib_nodes.get(e.get("url", "").lstrip("#"))
                if target is not None:
                    walk(target, m, depth + 1)
            elif e.tag == "node":
                walk(e, m, depth + 1)

    if scene is not None:
        for n in scene.findall("node"):
            walk(n, rootm)
    else:                                                # no scene at all: Assimp fails with "no root node"; be lenient
        for gid in geoms:
            emit(gid, rootm)
    if not out:
        raise ValueError("Collada: no triangle geometry instantiated")
    v = np.ascontiguousarray(np.concatenate(out).reshape(-1, 3), f32)
    return v, np.arange(len(v), dtype=np.uint32).reshape(-1, 3)


# ------------------------------------------------------------------------------------------------------------------
# dispatch
# ------------------------------------------------------------------------------------------------------------------
def mesh_format(name, data):
    ext = name.lower().rsplit(".", 1)[-1] if "." in name else ""
    if ext in ("stl", "stlb", "stla"):
        return "stl"
    if ext == "dae":
        return "collada"
    if ext == "obj":
        return "obj"
    head = bytes(data[:512]).lstrip()
    if head.startswith(b"<?xml") or b"<COLLADA" in head:
        return "collada"
    if re.match(rb"(#|v |vn |o |g |mtllib )", head):
        return "obj"
    return "stl"


def load_mesh(name, data, up_axis_to_y=True, apply_unit=False):
    """(file name or URI, bytes) -> (verts, tris) by extension, else by content."""
    fmt = mesh_format(name, data)
    if fmt == "collada":
        return load_collada(data, up_axis_to_y=up_axis_to_y, apply_unit=apply_unit)
    if fmt == "obj":
        return load_obj(data)
    return load_stl(data)
