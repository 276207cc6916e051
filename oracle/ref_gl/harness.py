"""ctypes wrapper of oracle/_ref/libllvmpipe_oracle.so.

Runs GLSL on Mesa llvmpipe with the reference's GL call sequence.  Two shader sets:
  "reference"  the reference's own files, read at run time from /root/reference/include/shaders (development
               container only): what tests/golden/generate_golden.py and tests/test_oracle_vs_llvmpipe.py use --
               the definition of "the reference's result" in this project;
  "standin"    oracle/ref_gl/standin_shaders/: a repo-authored re-statement of the same two tiny programs, checked
               bit-for-bit against the reference set in the development container; it exists only so that the
               timing leg of bench.py can run llvmpipe on the GPU box, where /root/reference does not exist.
TEST INFRASTRUCTURE, never imported by the product package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "..", "_ref", "libllvmpipe_oracle.so")
SHADER_DIR = "/root/reference/include/shaders"
STANDIN_DIR = os.path.join(_HERE, "standin_shaders")
SWRAST = os.environ.get("RGO_SWRAST_DRI", "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")

GL_TRIANGLES, GL_TRIANGLE_STRIP, GL_TRIANGLE_FAN, GL_QUADS, GL_QUAD_STRIP = 4, 5, 6, 7, 8


def available(shaders="reference"):
    if not (os.path.exists(_SO) and os.path.exists(SWRAST)):
        return False
    if shaders == "standin":
        return os.path.exists(os.path.join(STANDIN_DIR, "standin.frag"))
    return os.path.exists(os.path.join(SHADER_DIR, "urdf_filter.frag"))


def shader_paths(shaders="reference"):
    if shaders == "standin":
        return os.path.join(STANDIN_DIR, "standin.vert"), os.path.join(STANDIN_DIR, "standin.frag")
    return os.path.join(SHADER_DIR, "urdf_filter.vert"), os.path.join(SHADER_DIR, "urdf_filter.frag")


class Harness:
    def __init__(self, width, height, shaders="reference"):
        if not available(shaders):
            raise RuntimeError("llvmpipe harness unavailable (needs oracle/_ref, Mesa's swrast_dri.so and -- for the reference shader set -- /root/reference)")
        self.shaders = shaders
        L = ctypes.CDLL(_SO)
        vp = ctypes.c_void_p
        L.rgo_last_error.restype = ctypes.c_char_p
        L.rgo_renderer_string.restype = ctypes.c_char_p
        L.rgo_begin_frame.argtypes = [vp] * 4 + [ctypes.c_float] * 4
        L.rgo_mesh_create.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]
        L.rgo_push_link.argtypes = [vp]
        L.rgo_end_frame.argtypes = [vp] * 2
        L.rgo_scale.argtypes = [ctypes.c_float] * 3
        L.rgo_translate.argtypes = [ctypes.c_float] * 3
        L.rgo_draw_immediate_d.argtypes = [ctypes.c_int, vp, ctypes.c_int]
        L.rgo_now.restype = ctypes.c_double
        vs, fs = shader_paths(shaders)
        rc = L.rgo_create(width, height, vs.encode(), fs.encode())
        if rc != 0:
            raise RuntimeError("rgo_create failed: %s" % L.rgo_last_error().decode())
        self.L, self.width, self.height = L, width, height

    def use_shaders(self, shaders):
        """Re-links the program of the live context from the other shader set (the GL context is per process)."""
        vs, fs = shader_paths(shaders)
        self.L.rgo_set_program_source.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        rc = self.L.rgo_set_program_source(open(vs, "rb").read(), open(fs, "rb").read())
        if rc != 0:
            raise RuntimeError("linking the %s shaders failed: %s" % (shaders, self.L.rgo_last_error().decode()))
        self.shaders = shaders

    def read_attachment(self, i):
        """RGBA32F contents of colour attachment i of the last frame (0 sensor, 1 masked depth, 2 normals, 3 mask)."""
        out = np.zeros((self.height, self.width, 4), np.float32)
        self.L.rgo_read_attachment.argtypes = [ctypes.c_int, ctypes.c_void_p]
        self.L.rgo_read_attachment(i, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def renderer(self):
        return self.L.rgo_renderer_string().decode()

    def frame(self, depth, projection, renderables, camera_offset_inv=None, camera_tf=None, z_near=0.1, z_far=8.0,
              max_diff=0.05, replace_value=0.0, want_mask=True):
        """renderables: list of (link_tf[16], [draw, ...]); draw = ("mesh", pre_op, op, verts f32 [N,3], tris u32 [M,3])
        or ("prims", pre_op, op, [(gl_mode, verts f64 [N,3]), ...]).  Draws of one renderable share one
        glPushMatrix / glMultMatrixd(link_tf) bracket and matrix operations accumulate inside it, exactly
        like the reference's render() methods."""
        L = self.L
        I = np.eye(4).T.reshape(16).copy()
        d = np.ascontiguousarray(depth, np.float32)
        P = np.ascontiguousarray(projection, np.float64)
        oi = np.ascontiguousarray(I if camera_offset_inv is None else camera_offset_inv, np.float64)
        ct = np.ascontiguousarray(I if camera_tf is None else camera_tf, np.float64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L.rgo_mesh_clear()
        L.rgo_begin_frame(p(d), p(P), p(oi), p(ct), z_near, z_far, max_diff, replace_value)
        for link_tf, draws in renderables:
            tf = np.ascontiguousarray(link_tf, np.float64)
            L.rgo_push_link(p(tf))
            for dr in draws:
                kind, pre, op = dr[0], dr[1], dr[2]
                if pre == 1:
                    L.rgo_scale(float(op[0]), float(op[1]), float(op[2]))
                elif pre == 2:
                    L.rgo_translate(float(op[0]), float(op[1]), float(op[2]))
                if kind == "mesh":
                    v = np.ascontiguousarray(dr[3], np.float32)
                    t = np.ascontiguousarray(dr[4], np.uint32)
                    m = L.rgo_mesh_create(p(v), len(v), 3, p(t), t.size)
                    L.rgo_mesh_draw(m, GL_TRIANGLES)
                elif kind == "quads":
                    v = np.ascontiguousarray(dr[3], np.float32)
                    m = L.rgo_mesh_create(p(v), len(v), 3, None, 0)
                    L.rgo_mesh_draw(m, GL_QUADS)
                else:
                    for mode, verts in dr[3]:
                        vv = np.ascontiguousarray(verts, np.float64)
                        L.rgo_draw_immediate_d(mode, p(vv), len(vv))
            L.rgo_pop_link()
        out = np.zeros((self.height, self.width), np.float32)
        mask = np.zeros((self.height, self.width), np.uint8) if want_mask else None
        L.rgo_end_frame(p(out), p(mask) if want_mask else None)
        if L.rgo_gl_error() != 0:
            raise RuntimeError("GL error")
        return out, mask
