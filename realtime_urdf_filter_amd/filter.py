"""Host-side mirror of the reference's operator interface for the hot path, above the C ABI.

Same names, argument meaning and error behaviour as
  realtime_urdf_filter::RealtimeURDFFilter  include/realtime_urdf_filter/urdf_filter.h:51-143
  realtime_urdf_filter::URDFRenderer        include/realtime_urdf_filter/urdf_renderer.h:45-73
  realtime_urdf_filter::Renderable*         include/realtime_urdf_filter/renderable.h:54-143
with the ROS types replaced by plain Python ones: the private NodeHandle's parameters are a
`FilterParameters` object (+ a dict acting as the parameter server for robot descriptions),
tf::TransformListener is any object with `lookup_transform(target, source, stamp)`, and
sensor_msgs/CameraInfo is a `CameraInfo` tuple.  All per-pixel work happens in librtuf.so on
the GPU; this module only prepares matrices in double precision exactly like the reference's
host code does (tf::Transform algebra) and hands them to the C ABI.

The C++ facade (include/realtime_urdf_filter_amd/urdf_filter.hpp) exposes the same surface to
C++ hosts; both call the identical C entry points.
"""
import logging
from collections import namedtuple

import numpy as np

from . import _capi, geometry, urdf
from .urdf import Transform

log = logging.getLogger("realtime_urdf_filter")

CameraInfo = namedtuple("CameraInfo", "width height P")     # P: 12 doubles, row-major 3x4


class Renderable:
    """renderable.h:54-73."""

    def __init__(self):
        self.name = ""
        self.link_offset = Transform()
        self.link_to_fixed = Transform()
        self.draws = []

    def setLinkName(self, n):
        self.name = n

    def gl_matrix(self):
        """applyTransform (src/renderable.cpp:59-68): (link_to_fixed * link_offset).getOpenGLMatrix()."""
        return (self.link_to_fixed * self.link_offset).opengl_matrix()


class RenderableBox(Renderable):
    def __init__(self, dimx, dimy, dimz):
        super().__init__()
        self.dimx, self.dimy, self.dimz = np.float32(dimx), np.float32(dimy), np.float32(dimz)
        self.draws = geometry.box_draws(self.dimx, self.dimy, self.dimz)


class RenderableSphere(Renderable):
    def __init__(self, radius):
        super().__init__()
        self.radius = np.float32(radius)
        self.draws = geometry.sphere_draws(self.radius)


class RenderableCylinder(Renderable):
    def __init__(self, radius, length):
        super().__init__()
        self.radius, self.length = np.float32(radius), np.float32(length)
        self.draws = geometry.cylinder_draws(self.radius, self.length)


class RenderableMesh(Renderable):
    """src/renderable.cpp:306-322: a mesh that fails to load becomes a renderable with zero
    sub-meshes (draws nothing) and an error is logged (quirk Q15)."""

    def __init__(self, meshname, sx, sy, sz, mesh_loader=None):
        super().__init__()
        self.meshname = meshname
        try:
            if mesh_loader is None:
                # the reference resolves every URI through resource_retriever: package:// against ROS_PACKAGE_PATH
                mesh_loader = geometry.PackageResolver()
            v, t = mesh_loader(meshname)
            self.draws = geometry.mesh_draws(v, t, sx, sy, sz)
        except Exception as e:                      # noqa: BLE001 - mirror ROS_ERROR + continue
            log.error("Could not load resource [%s]: %s", meshname, e)
            self.draws = []


class URDFRenderer:
    """urdf_renderer.h:45-73 / src/urdf_renderer.cpp."""

    def __init__(self, model_description, tf_prefix, cam_frame, fixed_frame, tf, geometry_type="visual",
                 scale=1.0, ignore=(), mesh_loader=None):
        self.model_description_ = model_description
        self.tf_prefix_ = tf_prefix
        self.geometry_type = geometry_type
        self.scale = float(scale)
        self.ignore = set(ignore)
        self.camera_frame_ = cam_frame
        self.fixed_frame_ = fixed_frame
        self.tf_ = tf
        self.mesh_loader = mesh_loader
        self.renderables_ = []
        self.initURDFModel()

    def initURDFModel(self):
        try:
            model = urdf.Model.from_string(self.model_description_)
        except Exception as e:                      # noqa: BLE001 - ROS_FATAL + return
            log.critical("URDF failed Model parse: %s", e)
            return
        self.loadURDFModel(model)

    def loadURDFModel(self, model):
        for link in model.get_links():
            self.process_link(link)

    def process_link(self, link):
        """src/urdf_renderer.cpp:100-169."""
        if link.name in self.ignore:
            return
        if self.geometry_type in ("", "visual"):
            items = link.visual_array
        elif self.geometry_type == "collision":
            items = link.collision_array
        else:
            log.critical("invalid geometry type: %s", self.geometry_type)
            items = []
        s = self.scale
        for it in items:
            g = it.geometry
            if g.kind == "box":
                r = RenderableBox(s * g.size[0], s * g.size[1], s * g.size[2])
            elif g.kind == "cylinder":
                r = RenderableCylinder(s * g.radius, s * g.length)
            elif g.kind == "sphere":
                r = RenderableSphere(s * g.radius)
            elif g.kind == "mesh":
                r = RenderableMesh(g.filename, s * g.scale[0], s * g.scale[1], s * g.scale[2], self.mesh_loader)
            else:
                raise ValueError("unknown geometry type %r (the reference dereferences a null pointer here)" % g.kind)
            r.setLinkName(self.tf_prefix_ + "/" + link.name)
            r.link_offset = urdf.pose_to_transform(it.xyz, it.rpy)
            self.renderables_.append(r)

    def update_link_transforms(self, timestamp=None, tf=None):
        """src/urdf_renderer.cpp:173-190, including quirk Q7: a failed lookup is swallowed and the
        link re-uses the transform left by the previous link."""
        tf = tf if tf is not None else self.tf_
        t = Transform()
        for r in self.renderables_:
            try:
                t = tf.lookup_transform(self.fixed_frame_, r.name, timestamp)
            except Exception as e:                  # noqa: BLE001 - ROS_DEBUG
                log.debug("%s", e)
            # tf::Transform(t.getRotation(), t.getOrigin()) (src/urdf_renderer.cpp:187): the rotation goes through a
            # quaternion and back, in double -- kept, because the last bits of the matrix decide float32 roundings
            r.link_to_fixed = Transform.from_quaternion(t.get_rotation(), t.origin)

    def link_matrices(self):
        return np.stack([r.gl_matrix() for r in self.renderables_]) if self.renderables_ else np.zeros((0, 16))


class FilterParameters:
    """The rosparams of the private node handle (src/urdf_filter.cpp:58-111, launch/filter_parameters.yaml)."""

    def __init__(self, fixed_frame, camera_frame, models, depth_distance_threshold,
                 camera_offset_translation=(0.0, 0.0, 0.0), camera_offset_rotation=(0.0, 0.0, 0.0, 1.0),
                 show_gui=False, filter_replace_value=0.0, use_own_calibration=False,
                 own_calibration=(585.260, 585.028, 317.387, 239.264)):
        self.fixed_frame = fixed_frame
        self.camera_frame = camera_frame
        self.models = models                 # list of dicts: model, tf_prefix, geometry_type, [scale], [ignore]
        self.depth_distance_threshold = float(depth_distance_threshold)
        self.camera_offset_translation = tuple(camera_offset_translation)
        self.camera_offset_rotation = tuple(camera_offset_rotation)     # x y z w
        self.show_gui = bool(show_gui)       # accepted and ignored: there is no window system here
        self.filter_replace_value = float(filter_replace_value)
        # the reference's compile-time USE_OWN_CALIBRATION (src/urdf_filter.cpp:38, :462-472) as a parameter: fx fy cx cy used
        # instead of the CameraInfo's P (the reference's hard-coded values by default; they pass through float there)
        self.use_own_calibration = bool(use_own_calibration)
        self.own_calibration = tuple(float(np.float32(v)) for v in own_calibration)

    @staticmethod
    def from_dict(d):
        off = d.get("camera_offset", {})
        return FilterParameters(d["fixed_frame"], d["camera_frame"], d.get("models", []), d["depth_distance_threshold"],
                                off.get("translation", (0.0, 0.0, 0.0)), off.get("rotation", (0.0, 0.0, 0.0, 1.0)),
                                d.get("show_gui", False), d.get("filter_replace_value", 0.0),
                                d.get("use_own_calibration", False), d.get("own_calibration", (585.260, 585.028, 317.387, 239.264)))


class RealtimeURDFFilter:
    """urdf_filter.h:51-143.  One instance serves `max_streams` concurrent cameras (stream 0 is the
    reference-shaped single-camera interface); the public attribute names follow the reference."""

    def __init__(self, params, tf, param_server=None, max_streams=1, mesh_loader=None, device=0, two_kernel=False):
        self.params = params
        self.tf_ = tf
        self.param_server = param_server or {}
        self.mesh_loader = mesh_loader
        self.device = device
        self.max_streams = max_streams
        self.two_kernel = two_kernel
        self.fixed_frame_ = params.fixed_frame
        self.cam_frame_ = params.camera_frame
        self.camera_offset_t_ = np.asarray(params.camera_offset_translation, np.float64)
        self.camera_offset_q_ = tuple(params.camera_offset_rotation)
        self.depth_distance_threshold_ = params.depth_distance_threshold
        self.filter_replace_value_ = params.filter_replace_value
        self.show_gui_ = params.show_gui
        self.far_plane_, self.near_plane_ = 8.0, 0.1         # src/urdf_filter.cpp:53-54
        self.width_ = self.height_ = 0
        self.camera_tx_ = self.camera_ty_ = 0.0
        self.need_mask_ = True
        self.renderers_ = []
        self.masked_depth_ = None
        self.mask_ = None
        self._ctx = None
        self._model_ids = []
        self._batch_masked = None
        self._batch_mask = None

    # ---- loading -------------------------------------------------------------------------
    def loadModels(self):
        """src/urdf_filter.cpp:127-197 (appends, like the reference: quirk Q12)."""
        models = self.params.models
        if not isinstance(models, (list, tuple)):
            log.error("models parameter must be an array!")
            return
        for elem in models:
            description_param = elem["model"]
            content = self.param_server.get(description_param)
            if content is None:
                log.error("Parameter [%s] does not exist, and was not found by searchParam()", description_param)
                continue
            if not content:
                log.error("URDF is empty")
                continue
            ignore = elem.get("ignore", [])
            if isinstance(ignore, str):
                ignore = [ignore]
            self.renderers_.append(URDFRenderer(content, elem.get("tf_prefix", ""), self.cam_frame_, self.fixed_frame_,
                                                self.tf_, elem.get("geometry_type", ""), elem.get("scale", 1.0),
                                                ignore, self.mesh_loader))

    def initGL(self):
        """src/urdf_filter.cpp:386-436 without GL: create the device context for width_ x height_,
        load the models into device buffers, allocate the outputs."""
        p = _capi.default_params()
        p.near_plane, p.far_plane = self.near_plane_, self.far_plane_
        p.depth_distance_threshold = self.depth_distance_threshold_
        p.filter_replace_value = self.filter_replace_value_
        if self.two_kernel:
            p.flags |= _capi.FLAG_TWO_KERNEL
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = _capi.Context(self.width_, self.height_, self.max_streams, self.device, p)
        self.loadModels()
        if not self.renderers_:
            raise RuntimeError("Could not load any models for filtering!")
        self._model_ids = []
        for rd in self.renderers_:
            m = self._ctx.add_model()
            for r in rd.renderables_:
                l = self._ctx.add_link(m)
                for d in r.draws:
                    self._ctx.add_draw(m, l, d.verts, d.tris, d.pre_op, d.op)
            self._model_ids.append(m)
        self._ctx.finalize_models()
        self.masked_depth_ = np.zeros((self.height_, self.width_), np.float32)
        self.mask_ = np.zeros((self.height_, self.width_), np.uint8)

    # ---- camera --------------------------------------------------------------------------
    def getProjectionMatrix(self, info):
        """src/urdf_filter.cpp:459-501; sets camera_tx_/camera_ty_ as a side effect."""
        if self.params.use_own_calibration:      # the #ifdef branch: own intrinsics, camera_tx_ / camera_ty_ untouched
            fx, fy, cx, cy = self.params.own_calibration
            P, _, _ = _capi.projection_from_intrinsics(fx, fy, cx, cy, info.width, info.height, self.near_plane_, self.far_plane_, 0.0, 0.0)
            return P
        P, tx, ty = _capi.projection_from_intrinsics(info.P[0], info.P[5], info.P[2], info.P[6], info.width, info.height,
                                                     self.near_plane_, self.far_plane_, info.P[3], info.P[7])
        self.camera_tx_, self.camera_ty_ = tx, ty
        return P

    def _camera_matrices(self, tf, timestamp):
        """src/urdf_filter.cpp:520-534, :602-614.  Raises on TF failure (caller implements Q6)."""
        cam = tf.lookup_transform(self.cam_frame_, self.fixed_frame_, timestamp)
        offset = Transform.from_quaternion(self.camera_offset_q_, self.camera_offset_t_)
        offset_inv = offset.inverse().opengl_matrix()
        rot = Transform.from_quaternion(cam.get_rotation())
        right = rot * np.array([1.0, 0.0, 0.0])
        origin = cam.origin + right * self.camera_tx_
        down = rot * np.array([0.0, 1.0, 0.0])
        origin = origin + down * self.camera_ty_
        return offset_inv, Transform(cam.basis, origin).opengl_matrix()

    # ---- the hot path --------------------------------------------------------------------
    def _ensure_size(self, width, height):
        if self.width_ != width or self.height_ != height:
            if self.width_ != 0 or self.height_ != 0:
                log.error("image size has changed (%ix%i) -> (%ix%i)", self.width_, self.height_, width, height)
            self.width_, self.height_ = width, height
            self.initGL()

    def _stage_stream(self, stream, projection, tf, timestamp):
        offset_inv, cam_tf = self._camera_matrices(tf, timestamp)
        self._ctx.set_camera(stream, projection, offset_inv, cam_tf)
        for rd, m in zip(self.renderers_, self._model_ids):
            rd.update_link_transforms(timestamp, tf)
            if rd.renderables_:
                self._ctx.set_link_poses(stream, m, rd.link_matrices())

    def filter(self, buffer, projection_matrix, width, height, timestamp=None):
        """filter(unsigned char* buffer, double* glTf, int width, int height, ros::Time)."""
        self._ensure_size(width, height)
        if not self.renderers_:
            return
        try:
            self._stage_stream(0, projection_matrix, self.tf_, timestamp)
        except Exception as e:                      # noqa: BLE001 - ROS_ERROR + return (quirk Q6)
            log.error("%s", e)
            return
        depth = np.frombuffer(buffer, np.float32, width * height) if not isinstance(buffer, np.ndarray) else buffer
        masked, mask = self._ctx.filter_batch(np.asarray(depth, np.float32).reshape(1, height, width), want_mask=self.need_mask_)
        self.masked_depth_ = masked[0]
        if self.need_mask_:
            self.mask_ = mask[0]

    def getMaskedDepth(self):
        return self.masked_depth_

    def filter_callback(self, image, encoding, camera_info, stamp=None):
        """src/urdf_filter.cpp:270-330.  image: [H,W] float32 metres ("32FC1") or uint16 millimetres
        ("16UC1").  Returns (output_depth in the input encoding, mask or None)."""
        if encoding == "32FC1":
            depth_image = np.ascontiguousarray(image, np.float32)
        elif encoding == "16UC1":
            depth_image = depth_u16_to_f32(image)
        else:
            log.error("cv_bridge Exception: unsupported encoding %s", encoding)
            return None, None
        projection_matrix = self.getProjectionMatrix(camera_info)
        self.filter(depth_image, projection_matrix, depth_image.shape[1], depth_image.shape[0], stamp)
        out = self.masked_depth_
        if encoding == "16UC1":
            out = depth_f32_to_u16(out)
        return out, (self.mask_ if self.need_mask_ else None)

    # ---- batched extension ---------------------------------------------------------------
    def filter_batch(self, depths, projections, tfs, timestamp=None, want_mask=True):
        """N streams at once: depths [N,H,W] f32, projections [N,16] (or one [16]), tfs: one
        transform provider per stream.  A stream whose camera lookup fails keeps its previous
        output (quirk Q6)."""
        depths = np.ascontiguousarray(depths, np.float32)
        n, height, width = depths.shape
        self._ensure_size(width, height)
        projections = np.asarray(projections, np.float64).reshape(-1, 16)
        failed = []
        for s in range(n):
            P = projections[s if len(projections) > 1 else 0]
            try:
                self._stage_stream(s, P, tfs[s], timestamp)
            except Exception as e:                  # noqa: BLE001
                log.error("stream %d: %s", s, e)
                failed.append(s)
        masked, mask = self._ctx.filter_batch(depths, want_mask=want_mask)
        if self._batch_masked is None or self._batch_masked.shape != masked.shape:
            self._batch_masked = np.zeros_like(masked)
            self._batch_mask = np.zeros(masked.shape, np.uint8)
        for s in range(n):
            if s in failed:
                continue
            self._batch_masked[s] = masked[s]
            if want_mask:
                self._batch_mask[s] = mask[s]
        return self._batch_masked[:n], (self._batch_mask[:n] if want_mask else None)

    def stats(self):
        return self._ctx.stats() if self._ctx else {}


def depth_u16_to_f32(img_u16):
    """cv::Mat::convertTo(CV_32F, 0.001) (src/urdf_filter.cpp:288): float32 product, one rounding."""
    return (np.asarray(img_u16, np.uint16).astype(np.float32) * np.float32(0.001)).astype(np.float32)


def depth_f32_to_u16(img_f32):
    """cv::Mat::convertTo(CV_16U, 1000.0) (src/urdf_filter.cpp:311): float32 product, round half to even,
    saturate; NaN / +-inf -> 0."""
    with np.errstate(invalid="ignore", over="ignore"):
        v = (np.asarray(img_f32, np.float32) * np.float32(1000.0)).astype(np.float32)
        # cvRound on NaN / values outside the int32 range yields the "integer indefinite" value, which
        # saturate_cast<ushort> maps to 0
        bad = ~((v >= np.float32(-2147483648.0)) & (v < np.float32(2147483648.0)))
        r = np.rint(np.where(bad, 0.0, v))
        return np.clip(r, 0, 65535).astype(np.uint16)
