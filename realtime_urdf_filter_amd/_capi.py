"""ctypes binding of the C ABI in include/rtuf.h (librtuf.so, built by csrc/build.sh).

This is the same binding a maintainer of the reference would write for a Python host
(INTEGRATION.md); the C++ facade binds the identical symbols.  The library is the product
path: if it is missing or no GPU is visible, construction fails loudly -- there is no CPU
fallback anywhere in this package.
"""
import ctypes
import os

import numpy as np

_LIB_PATH = os.environ.get("RTUF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librtuf.so")

RTUF_OK = 0
RTUF_ERR_NO_DEVICE = -2
OP_NONE, OP_SCALE, OP_TRANSLATE = 0, 1, 2
FLAG_TWO_KERNEL = 2
FLAG_STRICT_GRID = 4
ABI_VERSION = 6
# rtuf_batch_status_device: bits of a batch slot's device status word (0 = the planes are final)
STATUS_PENDING_MASK = 0xffff
STATUS_BIN_OVERFLOW, STATUS_CLIP_OVERFLOW, STATUS_LIST_OVERFLOW, STATUS_GRID_SHORT, STATUS_UNCOVERED = 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20

#: every symbol include/rtuf.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "rtuf_default_params", "rtuf_abi_version", "rtuf_create", "rtuf_destroy", "rtuf_last_error",
    "rtuf_set_params", "rtuf_add_model", "rtuf_add_link", "rtuf_add_draw", "rtuf_finalize_models",
    "rtuf_num_links", "rtuf_num_triangles", "rtuf_num_vertices", "rtuf_set_stream_models", "rtuf_set_camera",
    "rtuf_projection_from_intrinsics", "rtuf_set_link_poses", "rtuf_set_cameras", "rtuf_set_camera_shift", "rtuf_set_link_poses_batch", "rtuf_set_kinematics", "rtuf_set_joint_positions", "rtuf_debug_read_poses", "rtuf_filter_batch",
    "rtuf_filter_batch_device", "rtuf_filter_batch_u16", "rtuf_filter_batch_device_u16", "rtuf_filter", "rtuf_get_masked_depth", "rtuf_get_mask", "rtuf_sync",
    "rtuf_stream", "rtuf_get_stats", "rtuf_enable_timing", "rtuf_debug_read_zsurface",
    "rtuf_filter_batch_async", "rtuf_filter_batch_u16_async", "rtuf_wait_oldest", "rtuf_host_alloc", "rtuf_host_free",
    "rtuf_mask_bits_words", "rtuf_filter_batch_bits_async", "rtuf_filter_batch_bits_u16_async", "rtuf_filter_batch_device_bits",
    "rtuf_filter_batch_device_bits_u16", "rtuf_expand_mask_bits", "rtuf_order_stream_after_batches", "rtuf_batch_status_device",
]


class Params(ctypes.Structure):
    _fields_ = [("near_plane", ctypes.c_float), ("far_plane", ctypes.c_float),
                ("depth_distance_threshold", ctypes.c_float), ("filter_replace_value", ctypes.c_float),
                ("flags", ctypes.c_uint32), ("bin_capacity", ctypes.c_uint32),
                ("max_inflight_streams", ctypes.c_uint32), ("pipelines", ctypes.c_uint32),
                ("raster_lanes", ctypes.c_uint32), ("memory_limit_mb", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 2)]


class Stats(ctypes.Structure):
    _fields_ = [("triangles_submitted", ctypes.c_uint64), ("triangles_binned", ctypes.c_uint64),
                ("bin_entries", ctypes.c_uint64), ("triangles_clipped", ctypes.c_uint64),
                ("max_bin_fill", ctypes.c_uint32), ("bin_capacity", ctypes.c_uint32),
                ("regrowths", ctypes.c_uint32), ("max_fbin_fill", ctypes.c_uint32), ("fragments_binned", ctypes.c_uint64),
                ("ms_pose", ctypes.c_float), ("ms_setup", ctypes.c_float), ("ms_raster", ctypes.c_float),
                ("ms_compare", ctypes.c_float), ("ms_total", ctypes.c_float), ("ms_clip", ctypes.c_float),
                ("timed_batches", ctypes.c_uint64), ("sum_ms_pose", ctypes.c_double), ("sum_ms_setup", ctypes.c_double),
                ("sum_ms_raster", ctypes.c_double), ("sum_ms_compare", ctypes.c_double), ("sum_ms_total", ctypes.c_double),
                ("sum_ms_clip", ctypes.c_double),
                ("device_bytes", ctypes.c_uint64), ("occluded_entries", ctypes.c_uint64),
                ("cover_tiles", ctypes.c_uint32), ("exact_tiles", ctypes.c_uint32),
                ("work_items", ctypes.c_uint32), ("zero_survivor_items", ctypes.c_uint32),
                ("cover_pass", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
                ("raster_atomics", ctypes.c_uint64), ("drawn_pixels", ctypes.c_uint64),
                ("raster_lanes", ctypes.c_uint32), ("launch_group", ctypes.c_uint32), ("groups_last_batch", ctypes.c_uint32),
                ("graphs_enabled", ctypes.c_uint32), ("graph_hits", ctypes.c_uint64), ("graph_misses", ctypes.c_uint64),
                ("lanes_side_by_side", ctypes.c_uint32),
                ("batch_status", ctypes.c_uint32), ("batch_reruns", ctypes.c_uint32), ("over_memory_limit", ctypes.c_uint32),
                ("copy_streams_side_by_side", ctypes.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


class RtufError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rtuf error %d: %s" % (code, message))
        self.code = code


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  A ROCm PyTorch wheel carries its own libamdhip64 / libhsa-runtime64 and
    loads them by path on `import torch`; librtuf.so is linked against the same soname (libamdhip64.so.7), so
    whichever of the two is loaded first decides which runtime librtuf binds to -- and if that is the system
    copy, torch later brings a second runtime into the process, which finds no GPU.  When such a wheel is
    installed and torch is not loaded yet, its runtime is mapped first (without importing torch), so both
    orders of use end up on the same runtime.  Without PyTorch the system runtime (/opt/rocm) is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)


def load_library(path=None):
    """Loads librtuf.so (no GPU needed for loading; rtuf_create needs one)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise RtufError(-100, "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % p)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(p)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.rtuf_default_params.argtypes = [ctypes.POINTER(Params)]
    lib.rtuf_default_params.restype = None
    lib.rtuf_abi_version.restype = ci
    lib.rtuf_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, ctypes.POINTER(Params)]
    lib.rtuf_destroy.argtypes = [vp]
    lib.rtuf_destroy.restype = None
    lib.rtuf_last_error.argtypes = [vp]
    lib.rtuf_last_error.restype = ctypes.c_char_p
    lib.rtuf_set_params.argtypes = [vp, ctypes.POINTER(Params)]
    lib.rtuf_add_model.argtypes = [vp]
    lib.rtuf_add_link.argtypes = [vp, ci]
    lib.rtuf_add_draw.argtypes = [vp, ci, ci, ci, vp, vp, ci, vp, ci]
    lib.rtuf_finalize_models.argtypes = [vp]
    lib.rtuf_num_links.argtypes = [vp, ci]
    lib.rtuf_num_triangles.argtypes = [vp]
    lib.rtuf_num_triangles.restype = ctypes.c_int64
    lib.rtuf_num_vertices.argtypes = [vp]
    lib.rtuf_num_vertices.restype = ctypes.c_int64
    lib.rtuf_set_stream_models.argtypes = [vp, ci, vp, ci]
    lib.rtuf_set_camera.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_projection_from_intrinsics.argtypes = [cd, cd, cd, cd, cd, cd, ci, ci, cd, cd, vp, vp, vp]
    lib.rtuf_projection_from_intrinsics.restype = None
    lib.rtuf_set_link_poses.argtypes = [vp, ci, ci, vp, ci]
    lib.rtuf_set_cameras.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.rtuf_set_link_poses_batch.argtypes = [vp, ci, ci, ci, vp, ci]
    lib.rtuf_set_camera_shift.argtypes = [vp, ci, ci, vp, vp]
    lib.rtuf_set_kinematics.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, ci]
    lib.rtuf_set_joint_positions.argtypes = [vp, ci, ci, ci, vp, vp, ci]
    lib.rtuf_debug_read_poses.argtypes = [vp, ci, vp, vp]
    lib.rtuf_filter_batch.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_filter_batch_device.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_filter_batch_u16.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_filter_batch_device_u16.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_filter.argtypes = [vp, vp, vp, ci, ci]
    lib.rtuf_get_masked_depth.argtypes = [vp]
    lib.rtuf_get_masked_depth.restype = ctypes.POINTER(ctypes.c_float)
    lib.rtuf_get_mask.argtypes = [vp]
    lib.rtuf_get_mask.restype = ctypes.POINTER(ctypes.c_uint8)
    lib.rtuf_sync.argtypes = [vp]
    lib.rtuf_stream.argtypes = [vp]
    lib.rtuf_stream.restype = vp
    lib.rtuf_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.rtuf_enable_timing.argtypes = [vp, ci]
    lib.rtuf_debug_read_zsurface.argtypes = [vp, ci, vp]
    lib.rtuf_filter_batch_async.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_filter_batch_u16_async.argtypes = [vp, ci, vp, vp, vp]
    lib.rtuf_wait_oldest.argtypes = [vp]
    lib.rtuf_host_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.rtuf_host_free.argtypes = [vp, vp]
    lib.rtuf_mask_bits_words.argtypes = [ci, ci]
    lib.rtuf_mask_bits_words.restype = ctypes.c_size_t
    lib.rtuf_filter_batch_bits_async.argtypes = [vp, ci, vp, vp]
    lib.rtuf_filter_batch_bits_u16_async.argtypes = [vp, ci, vp, vp]
    lib.rtuf_filter_batch_device_bits.argtypes = [vp, ci, vp, vp]
    lib.rtuf_filter_batch_device_bits_u16.argtypes = [vp, ci, vp, vp]
    lib.rtuf_expand_mask_bits.argtypes = [vp, ci, vp, ci, ci, ctypes.c_float, vp, vp]
    lib.rtuf_order_stream_after_batches.argtypes = [vp, vp]
    lib.rtuf_batch_status_device.argtypes = [vp, ctypes.POINTER(vp)]
    if path is None:
        _lib = lib
    return lib


def default_params():
    p = Params()
    load_library().rtuf_default_params(ctypes.byref(p))
    return p


def projection_from_intrinsics(fx, fy, cx, cy, width, height, near=0.1, far=8.0, Tx=0.0, Ty=0.0):
    """getProjectionMatrix (reference src/urdf_filter.cpp:459-501). Returns (P[16], camera_tx, camera_ty)."""
    P = np.zeros(16, np.float64)
    tx, ty = ctypes.c_double(), ctypes.c_double()
    load_library().rtuf_projection_from_intrinsics(fx, fy, cx, cy, Tx, Ty, width, height, near, far,
                                                   P.ctypes.data, ctypes.byref(tx), ctypes.byref(ty))
    return P, tx.value, ty.value


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class Context:
    """Thin object wrapper over an rtuf_context handle."""

    def __init__(self, width, height, max_streams=1, device=0, params=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        self.width, self.height, self.max_streams = width, height, max_streams
        p = params if params is not None else default_params()
        rc = self._lib.rtuf_create(ctypes.byref(self._h), device, width, height, max_streams, ctypes.byref(p))
        if rc != RTUF_OK:
            raise RtufError(rc, self._lib.rtuf_last_error(None).decode())
        self.params = p
        self._pinned = {}           # host_alloc() blocks: array address -> pointer

    def _check(self, rc):
        if rc < 0:
            raise RtufError(rc, self._lib.rtuf_last_error(self._h).decode())
        return rc

    def close(self):
        if self._h:
            self._lib.rtuf_destroy(self._h)          # also releases host_alloc() blocks still held
            self._h = ctypes.c_void_p()
            self._pinned = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # geometry
    def add_model(self):
        return self._check(self._lib.rtuf_add_model(self._h))

    def add_link(self, model):
        return self._check(self._lib.rtuf_add_link(self._h, model))

    def add_draw(self, model, link, verts, tris, pre_op=OP_NONE, op=(0.0, 0.0, 0.0)):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        o = np.asarray(op, np.float32)
        return self._check(self._lib.rtuf_add_draw(self._h, model, link, pre_op, _ptr(o), _ptr(v), len(v), _ptr(t), len(t)))

    def finalize_models(self):
        self._check(self._lib.rtuf_finalize_models(self._h))

    def num_triangles(self):
        return int(self._lib.rtuf_num_triangles(self._h))

    def num_vertices(self):
        return int(self._lib.rtuf_num_vertices(self._h))

    def set_stream_models(self, stream, model_ids):
        m = np.ascontiguousarray(model_ids, np.int32)
        self._check(self._lib.rtuf_set_stream_models(self._h, stream, _ptr(m), len(m)))

    def set_params(self, params):
        self._check(self._lib.rtuf_set_params(self._h, ctypes.byref(params)))
        self.params = params

    # poses
    def set_camera(self, stream, projection=None, camera_offset_inv=None, camera_tf=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64).reshape(16)
                for a in (projection, camera_offset_inv, camera_tf)]
        self._check(self._lib.rtuf_set_camera(self._h, stream, *[_ptr(a) for a in arrs]))

    def set_link_poses(self, stream, model, link_tf):
        a = np.ascontiguousarray(link_tf, np.float64).reshape(-1, 16)
        self._check(self._lib.rtuf_set_link_poses(self._h, stream, model, _ptr(a), len(a)))

    def set_cameras(self, first_stream, projections=None, camera_offset_inv=None, camera_tf=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64).reshape(-1, 16)
                for a in (projections, camera_offset_inv, camera_tf)]
        n = max(len(a) for a in arrs if a is not None)
        self._check(self._lib.rtuf_set_cameras(self._h, first_stream, n, *[_ptr(a) for a in arrs]))

    def set_camera_shift(self, first_stream, camera_tx, camera_ty):
        """camera_tx_/camera_ty_ (src/urdf_filter.cpp:607-611) for cameras posed by on-device forward kinematics."""
        tx = np.ascontiguousarray(camera_tx, np.float64).reshape(-1)
        ty = np.ascontiguousarray(camera_ty, np.float64).reshape(-1)
        assert len(tx) == len(ty)
        self._check(self._lib.rtuf_set_camera_shift(self._h, first_stream, len(tx), _ptr(tx), _ptr(ty)))

    def set_link_poses_batch(self, first_stream, model, link_tf):
        a = np.ascontiguousarray(link_tf, np.float64)
        n, nl = a.shape[0], a.shape[1]
        self._check(self._lib.rtuf_set_link_poses_batch(self._h, first_stream, n, model, _ptr(a), nl))

    def set_kinematics(self, model, parent, joint_type, joint_origin, joint_axis, link_frame, link_offset):
        """On-device FK tree of a model (see rtuf_set_kinematics)."""
        pa = np.ascontiguousarray(parent, np.int32)
        jt = np.ascontiguousarray(joint_type, np.int32)
        jo = np.ascontiguousarray(joint_origin, np.float64).reshape(-1, 16)
        ja = np.ascontiguousarray(joint_axis, np.float64).reshape(-1, 3)
        lf = np.ascontiguousarray(link_frame, np.int32)
        lo = np.ascontiguousarray(link_offset, np.float64).reshape(-1, 16)
        self._check(self._lib.rtuf_set_kinematics(self._h, model, len(pa), _ptr(pa), _ptr(jt), _ptr(jo), _ptr(ja), _ptr(lf), _ptr(lo), len(lf)))

    def set_joint_positions(self, first_stream, model, q, root_tf=None, camera_frame=-1):
        qa = np.ascontiguousarray(q, np.float64)
        rt = None if root_tf is None else np.ascontiguousarray(root_tf, np.float64).reshape(-1, 16)
        self._check(self._lib.rtuf_set_joint_positions(self._h, first_stream, qa.shape[0], model, _ptr(qa), _ptr(rt), camera_frame))

    def read_poses(self, n, n_links_total):
        tf = np.empty((n, max(n_links_total, 1), 16), np.float64)
        cam = np.empty((n, 16), np.float64)
        self._check(self._lib.rtuf_debug_read_poses(self._h, n, _ptr(tf), _ptr(cam)))
        return tf, cam

    # hot path
    def filter_batch(self, depth, want_mask=True):
        """depth: [n,H,W] float32 host array -> (masked [n,H,W] f32, mask [n,H,W] u8 or None)."""
        d = np.ascontiguousarray(depth, np.float32).reshape(-1, self.height, self.width)
        n = d.shape[0]
        masked = np.empty_like(d)
        mask = np.empty(d.shape, np.uint8) if want_mask else None
        PP = ctypes.c_void_p * n
        din = PP(*[d[i].ctypes.data for i in range(n)])
        mout = PP(*[masked[i].ctypes.data for i in range(n)])
        kout = PP(*[mask[i].ctypes.data for i in range(n)]) if want_mask else None
        self._check(self._lib.rtuf_filter_batch(self._h, n, din, mout, kout))
        return masked, mask

    def filter_batch_u16(self, depth_mm, want_mask=True):
        """16UC1: depth_mm [n,H,W] uint16 millimetres -> (masked uint16 [n,H,W], mask)."""
        d = np.ascontiguousarray(depth_mm, np.uint16).reshape(-1, self.height, self.width)
        n = d.shape[0]
        masked = np.empty_like(d)
        mask = np.empty(d.shape, np.uint8) if want_mask else None
        PP = ctypes.c_void_p * n
        din = PP(*[d[i].ctypes.data for i in range(n)])
        mout = PP(*[masked[i].ctypes.data for i in range(n)])
        kout = PP(*[mask[i].ctypes.data for i in range(n)]) if want_mask else None
        self._check(self._lib.rtuf_filter_batch_u16(self._h, n, din, mout, kout))
        return masked, mask

    # asynchronous host planes
    def host_alloc(self, shape, dtype):
        """Pinned host array (rtuf_host_alloc) for filter_batch_async; release with host_free()."""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = ctypes.c_void_p()
        self._check(self._lib.rtuf_host_alloc(self._h, nbytes, ctypes.byref(p)))
        buf = (ctypes.c_char * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data)
        self._check(self._lib.rtuf_host_free(self._h, ctypes.c_void_p(p)))

    def filter_batch_async(self, depth, masked, mask=None):
        """Enqueue only: depth / masked [n,H,W] float32 (or uint16 for 16UC1) host arrays, mask [n,H,W] u8 or
        None; the arrays must stay alive and untouched until wait_oldest() / sync() retires the batch."""
        n = depth.shape[0]
        u16 = depth.dtype == np.uint16
        assert depth.flags.c_contiguous and masked.flags.c_contiguous and masked.dtype == depth.dtype
        assert depth.dtype in (np.float32, np.uint16) and depth.shape == (n, self.height, self.width) == masked.shape
        assert mask is None or (mask.flags.c_contiguous and mask.dtype == np.uint8 and mask.shape == depth.shape)
        PP = ctypes.c_void_p * n
        din = PP(*[depth[i].ctypes.data for i in range(n)])
        mout = PP(*[masked[i].ctypes.data for i in range(n)])
        kout = PP(*[mask[i].ctypes.data for i in range(n)]) if mask is not None else None
        fn = self._lib.rtuf_filter_batch_u16_async if u16 else self._lib.rtuf_filter_batch_async
        self._check(fn(self._h, n, din, mout, kout))

    # mask-only output, one bit per pixel
    def mask_bits_words(self):
        return int(self._lib.rtuf_mask_bits_words(self.width, self.height))

    def filter_batch_bits_async(self, depth, bits):
        """Enqueue only: depth [n,H,W] float32 or uint16 host array, bits [n, mask_bits_words()] uint32 host array
        (pinned for overlap); retire with wait_oldest() / sync()."""
        n = depth.shape[0]
        u16 = depth.dtype == np.uint16
        assert depth.flags.c_contiguous and depth.dtype in (np.float32, np.uint16) and depth.shape == (n, self.height, self.width)
        assert bits.flags.c_contiguous and bits.dtype == np.uint32 and bits.shape == (n, self.mask_bits_words())
        PP = ctypes.c_void_p * n
        din = PP(*[depth[i].ctypes.data for i in range(n)])
        bout = PP(*[bits[i].ctypes.data for i in range(n)])
        fn = self._lib.rtuf_filter_batch_bits_u16_async if u16 else self._lib.rtuf_filter_batch_bits_async
        self._check(fn(self._h, n, din, bout))

    def filter_batch_device_bits(self, n, d_depth, d_bits, u16=False):
        fn = self._lib.rtuf_filter_batch_device_bits_u16 if u16 else self._lib.rtuf_filter_batch_device_bits
        self._check(fn(self._h, n, ctypes.c_void_p(d_depth), ctypes.c_void_p(d_bits)))

    def wait_oldest(self):
        self._check(self._lib.rtuf_wait_oldest(self._h))

    def filter_batch_device_u16(self, n, d_depth, d_masked, d_mask=None):
        self._check(self._lib.rtuf_filter_batch_device_u16(self._h, n, ctypes.c_void_p(d_depth), ctypes.c_void_p(d_masked),
                                                           ctypes.c_void_p(d_mask) if d_mask else None))

    def filter_batch_device(self, n, d_depth, d_masked, d_mask=None):
        """Device pointers (ints): enqueue only; call sync()."""
        self._check(self._lib.rtuf_filter_batch_device(self._h, n, ctypes.c_void_p(d_depth), ctypes.c_void_p(d_masked),
                                                       ctypes.c_void_p(d_mask) if d_mask else None))

    def filter(self, buffer, projection):
        """RealtimeURDFFilter::filter(buffer, glTf, w, h) + getMaskedDepth()/mask_."""
        b = np.ascontiguousarray(buffer, np.float32).reshape(self.height, self.width)
        P = np.ascontiguousarray(projection, np.float64).reshape(16)
        self._check(self._lib.rtuf_filter(self._h, _ptr(b), _ptr(P), self.width, self.height))
        n = self.width * self.height
        md = np.ctypeslib.as_array(self._lib.rtuf_get_masked_depth(self._h), shape=(n,)).reshape(self.height, self.width).copy()
        mk = np.ctypeslib.as_array(self._lib.rtuf_get_mask(self._h), shape=(n,)).reshape(self.height, self.width).copy()
        return md, mk

    def sync(self):
        self._check(self._lib.rtuf_sync(self._h))

    def stream_handle(self):
        """hipStream_t of a context with ONE raster lane and one pipeline; None for every other context (never hand that
        None to HIP as a stream: use order_stream_after_batches)."""
        return self._lib.rtuf_stream(self._h)

    def order_stream_after_batches(self, hip_stream=None):
        """Makes `hip_stream` (an int / c_void_p hipStream_t; None = the legacy default stream) wait on the device for every
        batch enqueued so far, whatever the number of lanes and pipelines."""
        self._check(self._lib.rtuf_order_stream_after_batches(self._h, ctypes.c_void_p(hip_stream) if hip_stream else None))

    def batch_status_device(self):
        """Device address (int) of the status word of the batch the last filter call enqueued: 0 there, read on a stream
        ordered behind the batch (order_stream_after_batches), means its planes are final; STATUS_* bits say why not."""
        p = ctypes.c_void_p()
        self._check(self._lib.rtuf_batch_status_device(self._h, ctypes.byref(p)))
        return p.value

    def enable_timing(self, on=True):
        # True/1: every stage; 2: only around the tile (and compare) kernel; False/0: off
        self._check(self._lib.rtuf_enable_timing(self._h, int(on)))

    def stats(self):
        s = Stats()
        self._check(self._lib.rtuf_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def read_zsurface(self, n):
        out = np.empty((n, self.height, self.width), np.float32)
        self._check(self._lib.rtuf_debug_read_zsurface(self._h, n, _ptr(out)))
        return out


def expand_mask_bits(depth, bits, replace_value, want_masked=True, want_mask=True):
    """Host helper rtuf_expand_mask_bits for one frame: depth [H,W] float32 or uint16, bits [mask_bits_words] uint32
    -> (masked like depth or None, mask u8 [H,W] or None)."""
    d = np.ascontiguousarray(depth)
    assert d.dtype in (np.float32, np.uint16) and d.ndim == 2
    H, W = d.shape
    b = np.ascontiguousarray(bits, np.uint32)
    masked = np.empty_like(d) if want_masked else None
    mask = np.empty((H, W), np.uint8) if want_mask else None
    rc = load_library().rtuf_expand_mask_bits(_ptr(d), 1 if d.dtype == np.uint16 else 0, _ptr(b), W, H, float(replace_value), _ptr(masked), _ptr(mask))
    if rc != RTUF_OK:
        raise RtufError(rc, "rtuf_expand_mask_bits failed")
    return masked, mask
