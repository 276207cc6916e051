"""Host-side geometry providers (run once at load time): what the reference's Renderable*
constructors and render() methods hand to OpenGL, expressed as indexed triangles in the
order the GL driver assembles them.

Reference sites (relative to the reference checkout):
  RenderableBox      src/renderable.cpp:100-170   (24-vertex VBO as GL_QUADS, then
                                                   glScalef(dimx,dimy,dimz) + glutSolidCube(dimx): quirk Q1)
  RenderableSphere   src/renderable.cpp:76-85     glutSolidSphere(radius, 10, 10)
  RenderableCylinder src/renderable.cpp:88-98     glTranslatef(0,0,-l/2) + glutSolidCylinder(r, l, 10, 10)
  RenderableMesh     src/renderable.cpp:306-452   Assimp import, glScalef(scale) + GL_TRIANGLES

Primitive assembly order of the GL driver (Mesa llvmpipe, provoking vertex last; verified
against the llvmpipe oracle harness in tests/test_oracle_vs_llvmpipe.py):
  GL_QUADS          (v0,v1,v2,v3)      -> (v0,v1,v3), (v1,v2,v3)
  GL_TRIANGLE_FAN   (c,v1,v2,...)      -> (c,v1,v2), (c,v2,v3), ...
  GL_QUAD_STRIP     (v0,v1,v2,v3,...)  -> (v0,v1,v3), (v2,v0,v3), then (v2,v3,v5), (v4,v2,v5), ...
The freeglut tessellations are restated from freeglut 2.8's published algorithm (the
library is not in the image and is not version-pinned by the reference): [recall].
"""
import math
import struct

import numpy as np

from ._capi import OP_NONE, OP_SCALE, OP_TRANSLATE


class DrawCall:
    """One GL draw call inside a renderable's push/pop bracket."""

    def __init__(self, verts, tris, pre_op=OP_NONE, op=(0.0, 0.0, 0.0)):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        self.pre_op = int(pre_op)
        self.op = tuple(float(np.float32(x)) for x in op)


def quads_to_tris(n_quads, base=0):
    t = []
    for q in range(n_quads):
        i = base + 4 * q
        t += [(i, i + 1, i + 3), (i + 1, i + 2, i + 3)]
    return t


def fan_to_tris(n_verts, base=0):
    return [(base, base + i - 1, base + i) for i in range(2, n_verts)]


def quad_strip_to_tris(n_verts, base=0):
    t = []
    for i in range(3, n_verts, 2):
        t += [(base + i - 3, base + i - 2, base + i), (base + i - 1, base + i - 3, base + i)]
    return t


def box_draws(dimx, dimy, dimz):
    """RenderableBox::render: the VBO box, then the scaled glutSolidCube (quirk Q1)."""
    dx, dy, dz = np.float32(dimx), np.float32(dimy), np.float32(dimz)
    h = np.float32(0.5)
    X, Y, Z = h * dx, h * dy, h * dz          # float32 products, like `0.5f * dimx`
    v = np.array([
        [X, Y, -Z], [-X, Y, -Z], [-X, Y, Z], [X, Y, Z],          # top
        [X, -Y, Z], [-X, -Y, Z], [-X, -Y, -Z], [X, -Y, -Z],      # bottom
        [X, Y, Z], [-X, Y, Z], [-X, -Y, Z], [X, -Y, Z],          # front
        [X, -Y, -Z], [-X, -Y, -Z], [-X, Y, -Z], [X, Y, -Z],      # back
        [-X, Y, Z], [-X, Y, -Z], [-X, -Y, -Z], [-X, -Y, Z],      # left
        [X, Y, -Z], [X, Y, Z], [X, -Y, Z], [X, -Y, -Z]], np.float32)
    first = DrawCall(v, quads_to_tris(6))
    second = prims_to_draw(cube_prims(dx), OP_SCALE, (dx, dy, dz))
    return [first, second]


def box_vbo_vertices(dimx, dimy, dimz):
    """The 24 float32 vertices of RenderableBox::createBoxVBO (drawn as GL_QUADS)."""
    return box_draws(dimx, dimy, dimz)[0].verts


def cube_prims(size):
    """glutSolidCube(size), freeglut 2.8 [recall]: size = dSize * 0.5 in double, one GL_QUADS block."""
    s = float(np.float32(size)) * 0.5
    P, N = s, -s
    c = np.array([
        [P, N, P], [P, N, N], [P, P, N], [P, P, P],
        [P, P, P], [P, P, N], [N, P, N], [N, P, P],
        [P, P, P], [N, P, P], [N, N, P], [P, N, P],
        [N, N, P], [N, P, P], [N, P, N], [N, N, N],
        [N, N, P], [N, N, N], [P, N, N], [P, N, P],
        [N, N, N], [N, P, N], [P, P, N], [P, N, N]], np.float64)
    return [(GL_QUADS, c)]


def _circle_table(n):
    """fghCircleTable: size+1 entries, negative n reverses direction."""
    size = abs(n)
    angle = 2 * math.pi / (n if n != 0 else 1)
    sint = [0.0] * (size + 1)
    cost = [0.0] * (size + 1)
    sint[0], cost[0] = 0.0, 1.0
    for i in range(1, size):
        sint[i] = math.sin(angle * i)
        cost[i] = math.cos(angle * i)
    sint[size], cost[size] = sint[0], cost[0]
    return sint, cost


GL_TRIANGLE_FAN, GL_QUADS, GL_QUAD_STRIP = 6, 7, 8


def prims_to_draw(prims, pre_op=OP_NONE, op=(0.0, 0.0, 0.0)):
    """[(gl_mode, verts f64 [N,3]), ...] (glBegin/glVertex3d/glEnd blocks) -> one DrawCall with the
    triangles in the order the GL driver assembles them; glVertex3d rounds to float32."""
    verts, tris = [], []
    for mode, v in prims:
        base = len(verts)
        verts += [tuple(x) for x in np.asarray(v, np.float64)]
        n = len(v)
        if mode == GL_TRIANGLE_FAN:
            tris += fan_to_tris(n, base)
        elif mode == GL_QUAD_STRIP:
            tris += quad_strip_to_tris(n, base)
        elif mode == GL_QUADS:
            tris += quads_to_tris(n // 4, base)
        else:
            raise ValueError("unsupported GL mode %r" % mode)
    return DrawCall(np.asarray(verts, np.float64).astype(np.float32), tris, pre_op, op)


def sphere_prims(radius, slices=10, stacks=10):
    """glutSolidSphere(radius, slices, stacks), freeglut 2.8 [recall]: the glBegin/glEnd blocks."""
    radius = float(np.float32(radius))
    sint1, cost1 = _circle_table(-slices)
    sint2, cost2 = _circle_table(stacks * 2)
    prims = []
    z1, r1 = cost2[1 if stacks > 0 else 0], sint2[1 if stacks > 0 else 0]
    v = [(0.0, 0.0, radius)]
    for j in range(slices, -1, -1):
        v.append((cost1[j] * r1 * radius, sint1[j] * r1 * radius, z1 * radius))
    prims.append((GL_TRIANGLE_FAN, np.asarray(v, np.float64)))
    z0, r0 = z1, r1
    for i in range(1, stacks - 1):
        z0, z1 = z1, cost2[i + 1]
        r0, r1 = r1, sint2[i + 1]
        v = []
        for j in range(slices + 1):
            v.append((cost1[j] * r1 * radius, sint1[j] * r1 * radius, z1 * radius))
            v.append((cost1[j] * r0 * radius, sint1[j] * r0 * radius, z0 * radius))
        prims.append((GL_QUAD_STRIP, np.asarray(v, np.float64)))
    z0, r0 = z1, r1
    v = [(0.0, 0.0, -radius)]
    for j in range(slices + 1):
        v.append((cost1[j] * r0 * radius, sint1[j] * r0 * radius, z0 * radius))
    prims.append((GL_TRIANGLE_FAN, np.asarray(v, np.float64)))
    return prims


def sphere_draws(radius, slices=10, stacks=10):
    return [prims_to_draw(sphere_prims(radius, slices, stacks))]


def cylinder_prims(radius, length, slices=10, stacks=10):
    """glutSolidCylinder(radius, length, slices, stacks), freeglut 2.8 [recall]."""
    radius = float(np.float32(radius))
    height = float(np.float32(length))
    sint, cost = _circle_table(-slices)
    zstep = height / (stacks if stacks > 0 else 1)
    prims = []
    v = [(0.0, 0.0, 0.0)]
    for j in range(slices + 1):
        v.append((cost[j] * radius, sint[j] * radius, 0.0))
    prims.append((GL_TRIANGLE_FAN, np.asarray(v, np.float64)))
    v = [(0.0, 0.0, height)]
    for j in range(slices, -1, -1):
        v.append((cost[j] * radius, sint[j] * radius, height))
    prims.append((GL_TRIANGLE_FAN, np.asarray(v, np.float64)))
    z0, z1 = 0.0, zstep
    for i in range(1, stacks + 1):
        if i == stacks:
            z1 = height
        v = []
        for j in range(slices + 1):
            v.append((cost[j] * radius, sint[j] * radius, z0))
            v.append((cost[j] * radius, sint[j] * radius, z1))
        prims.append((GL_QUAD_STRIP, np.asarray(v, np.float64)))
        z0 = z1
        z1 += zstep
    return prims


def cylinder_translate(length):
    """The glTranslatef(0, 0, -length/2) of RenderableCylinder::render, evaluated in float."""
    return (0.0, 0.0, float(-np.float32(length) / np.float32(2)))


def cylinder_draws(radius, length, slices=10, stacks=10):
    return [prims_to_draw(cylinder_prims(radius, length, slices, stacks), OP_TRANSLATE, cylinder_translate(length))]


def mesh_draws(verts, tris, sx=1.0, sy=1.0, sz=1.0):
    """RenderableMesh::render: glScalef(scale) then indexed GL_TRIANGLES (one draw per sub-mesh)."""
    return [DrawCall(verts, tris, OP_SCALE, (np.float32(sx), np.float32(sy), np.float32(sz)))]


# --------------------------------------------------------------------------------------
# STL reader (replaces the Assimp import for the common URDF mesh format).  Handles binary
# files whose 80-byte header starts with "solid" (README.md:121-143 of the reference).
# --------------------------------------------------------------------------------------
def load_stl(data):
    """bytes -> (verts [3T,3] f32, tris [T,3] u32).  No vertex welding: like Assimp's STL
    importer every facet brings its own three vertices."""
    if len(data) >= 84:
        (n,) = struct.unpack_from("<I", data, 80)
        if 84 + 50 * n == len(data):
            rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
            v = rec["v"].reshape(-1, 3).astype(np.float32)
            return v, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    text = data.decode("ascii", errors="replace")
    if not text.lstrip().lower().startswith("solid"):
        raise ValueError("not an STL file")
    vs = []
    for line in text.splitlines():
        p = line.split()
        if len(p) == 4 and p[0].lower() == "vertex":
            vs.append((float(p[1]), float(p[2]), float(p[3])))
    if len(vs) % 3:
        raise ValueError("ASCII STL vertex count is not a multiple of 3")
    v = np.asarray(vs, np.float32).reshape(-1, 3)
    return v, np.arange(len(v), dtype=np.uint32).reshape(-1, 3)


def write_binary_stl(verts, tris, header=b"binary stl"):
    v = np.asarray(verts, np.float32)[np.asarray(tris, np.int64).reshape(-1)].reshape(-1, 3, 3)
    n = len(v)
    rec = np.zeros(n, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["v"] = v
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    nn = np.cross(e1, e2)
    ln = np.linalg.norm(nn, axis=1, keepdims=True)
    rec["n"] = np.where(ln > 0, nn / np.maximum(ln, 1e-30), 0)
    return header.ljust(80, b"\0")[:80] + struct.pack("<I", n) + rec.tobytes()


class PackageResolver:
    """mesh_loader for URDFRenderer / RealtimeURDFFilter: resolves the mesh URIs a URDF carries --
    `package://<pkg>/<path>` against a list of package search roots (ROS_PACKAGE_PATH semantics: a root either IS the
    package directory `<pkg>` or contains it), `file://<path>` and plain paths -- and loads the file.  Replaces the
    resource_retriever + Assimp pair of the reference (src/renderable.cpp:306-322).  Formats: STL (binary, incl. the
    "solid" header quirk, and ASCII), Collada (.dae) and Wavefront OBJ (`meshes.py`, which also says what of Assimp's
    behaviour is restated and what is not pinned).  `up_axis_to_y` / `apply_unit`: see `meshes.load_collada`."""

    def __init__(self, search_paths=None, up_axis_to_y=True, apply_unit=False):
        import os
        self.up_axis_to_y, self.apply_unit = up_axis_to_y, apply_unit
        paths = list(search_paths) if search_paths is not None else [p for p in os.environ.get("ROS_PACKAGE_PATH", "").split(":") if p]
        self.search_paths = paths

    def path_of(self, uri):
        import os
        if uri.startswith("package://"):
            pkg, _, rel = uri[len("package://"):].partition("/")
            for root in self.search_paths:
                for cand in (os.path.join(root, pkg, rel), os.path.join(root, rel) if os.path.basename(os.path.normpath(root)) == pkg else None):
                    if cand and os.path.isfile(cand):
                        return cand
            raise IOError("package %r of %r not found under %r" % (pkg, uri, self.search_paths))
        if uri.startswith("file://"):
            return uri[len("file://"):]
        return uri

    def __call__(self, uri):
        from . import meshes
        path = self.path_of(uri)
        with open(path, "rb") as f:
            return meshes.load_mesh(path, f.read(), up_axis_to_y=self.up_axis_to_y, apply_unit=self.apply_unit)
