"""ctypes binding of the CPU oracle (oracle/rtuf_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package realtime_urdf_filter_amd."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librtuf_oracle.so")

OP_NONE, OP_SCALE, OP_TRANSLATE = 0, 1, 2


class Draw(ctypes.Structure):
    _fields_ = [("link_tf", ctypes.c_double * 16), ("pre_op", ctypes.c_int32), ("op", ctypes.c_float * 3),
                ("verts", ctypes.c_void_p), ("nverts", ctypes.c_int32), ("tris", ctypes.c_void_p),
                ("ntris", ctypes.c_int32)]


class Frame(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("depth", ctypes.c_void_p),
                ("z_near", ctypes.c_float), ("z_far", ctypes.c_float), ("max_diff", ctypes.c_float),
                ("replace_value", ctypes.c_float), ("projection", ctypes.c_double * 16),
                ("camera_offset_inv", ctypes.c_double * 16), ("camera_tf", ctypes.c_double * 16),
                ("draws", ctypes.c_void_p), ("ndraws", ctypes.c_int32)]


class Debug(ctypes.Structure):
    _fields_ = [("zwin", ctypes.c_void_p), ("prim", ctypes.c_void_p), ("n_tris_in", ctypes.c_long),
                ("n_tris_setup", ctypes.c_long), ("n_frags", ctypes.c_long)]


class Variants(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("vs_fma", "vp_fma", "clip_vp_fma", "interp_fma", "frag_div_rcp",
                                              "cw_swap_12", "edge_rule_flip", "clip_old_t")]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/librtuf_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.rtuf_oracle_filter.argtypes = [ctypes.POINTER(Frame), ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Debug)]
        _lib.rtuf_oracle_filter_throughput.argtypes = [ctypes.POINTER(Frame), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        _lib.rtuf_oracle_filter_throughput.restype = ctypes.c_long
        _lib.rtuf_oracle_compose_mvp.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def set_variants(**kw):
    v = Variants()
    lib().rtuf_oracle_get_variants(ctypes.byref(v))
    for k, x in kw.items():
        setattr(v, k, x)
    lib().rtuf_oracle_set_variants(ctypes.byref(v))


IDENTITY = np.eye(4).T.reshape(16).copy()


class PreparedFrame:
    """The ctypes structures of one oracle call built once; run() then spends its time in the C function only (ctypes
    releases the GIL for the call), so several threads can run prepared frames on several cores (bench.py's all-cores
    leg).  One PreparedFrame must not be run by two threads at once (it owns its output planes)."""

    def __init__(self, depth, projection, draws, camera_offset_inv=None, camera_tf=None, z_near=0.1, z_far=8.0,
                 max_diff=0.05, replace_value=0.0, want_debug=False):
        depth = np.ascontiguousarray(depth, np.float32)
        H, W = depth.shape
        self._keep = [depth]
        arr = (Draw * max(len(draws), 1))()
        for i, (tf, pre, op, v, t) in enumerate(draws):
            v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
            t = np.ascontiguousarray(t, np.uint32).reshape(-1, 3)
            self._keep += [v, t]
            arr[i].link_tf[:] = list(np.asarray(tf, np.float64).reshape(16))
            arr[i].pre_op = int(pre)
            arr[i].op[:] = [float(x) for x in op]
            arr[i].verts = v.ctypes.data
            arr[i].nverts = len(v)
            arr[i].tris = t.ctypes.data
            arr[i].ntris = len(t)
        fr = Frame()
        fr.width, fr.height, fr.depth = W, H, depth.ctypes.data
        fr.z_near, fr.z_far, fr.max_diff, fr.replace_value = z_near, z_far, max_diff, replace_value
        fr.projection[:] = list(np.asarray(projection, np.float64).reshape(16))
        fr.camera_offset_inv[:] = list(IDENTITY if camera_offset_inv is None else np.asarray(camera_offset_inv, np.float64).reshape(16))
        fr.camera_tf[:] = list(IDENTITY if camera_tf is None else np.asarray(camera_tf, np.float64).reshape(16))
        fr.draws = ctypes.addressof(arr)
        fr.ndraws = len(draws)
        self._arr, self.fr = arr, fr
        self.masked = np.zeros((H, W), np.float32)
        self.mask = np.zeros((H, W), np.uint8)
        self.dbg = Debug()
        self.zwin = self.prim = None
        if want_debug:
            self.zwin = np.zeros((H, W), np.float32)
            self.prim = np.zeros((H, W), np.int32)
            self.dbg.zwin, self.dbg.prim = self.zwin.ctypes.data, self.prim.ctypes.data
        self._fn = lib().rtuf_oracle_filter

    def run(self):
        rc = self._fn(ctypes.byref(self.fr), self.masked.ctypes.data, self.mask.ctypes.data, ctypes.byref(self.dbg))
        if rc != 0:
            raise RuntimeError("oracle failed: %d" % rc)
        return self.masked, self.mask


def run_prepared(prepared, n_threads):
    """Runs every prepared frame once, on up to n_threads host threads (the C call releases the GIL); the results are in
    each frame's .masked / .mask.  The all-stream parity checks of bench.py and the full-size GPU tests."""
    if n_threads <= 1 or len(prepared) <= 1:
        for f in prepared:
            f.run()
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(int(n_threads), len(prepared))) as ex:
        list(ex.map(lambda f: f.run(), prepared))


def usable_threads():
    """Host threads this process can really run at once: min(visible hardware threads, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def filter_throughput(prepared, seconds, n_threads):
    """Filters the prepared frames cyclically for `seconds` on n_threads POSIX threads inside the C library (no Python
    in the loop); returns (frames filtered, elapsed seconds).  bench.py's cpu_baseline.all_cores leg."""
    arr = (Frame * len(prepared))(*[p.fr for p in prepared])
    el = ctypes.c_double(0.0)
    n = lib().rtuf_oracle_filter_throughput(arr, len(prepared), float(seconds), int(n_threads), ctypes.byref(el))
    if n < 0:
        raise RuntimeError("oracle throughput run failed: %d" % n)
    return int(n), float(el.value)


def filter_frame(depth, projection, draws, camera_offset_inv=None, camera_tf=None, z_near=0.1, z_far=8.0,
                 max_diff=0.05, replace_value=0.0, want_debug=False):
    """One frame through the oracle.

    draws: list of (link_tf[16] f64 column-major, pre_op, op[3], verts [N,3] f32, tris [M,3] u32).
    Returns (masked f32 [H,W], mask u8 [H,W]) or, with want_debug, additionally
    (zwin f32 [H,W], prim i32 [H,W], (tris_in, tris_setup, frags)).
    """
    f = PreparedFrame(depth, projection, draws, camera_offset_inv, camera_tf, z_near, z_far, max_diff, replace_value, want_debug)
    masked, mask = f.run()
    if want_debug:
        return masked, mask, f.zwin, f.prim, (f.dbg.n_tris_in, f.dbg.n_tris_setup, f.dbg.n_frags)
    return masked, mask
