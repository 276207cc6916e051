// urdf_filter.hpp -- C++ facade over the C ABI (include/rtuf.h) that keeps the reference's class
// surface for the hot path: same class names, method names, argument meaning and error behaviour as
//   realtime_urdf_filter::RealtimeURDFFilter   include/realtime_urdf_filter/urdf_filter.h:51-143
//   realtime_urdf_filter::URDFRenderer         include/realtime_urdf_filter/urdf_renderer.h:45-73
//   realtime_urdf_filter::Renderable*          include/realtime_urdf_filter/renderable.h:54-143
// with the ROS types replaced by plain ones (no ROS exists where this is built):
//   ros::NodeHandle rosparams        -> FilterParameters (+ a string map as parameter server)
//   tf::TransformListener            -> rtuf_host::TransformProvider
//   sensor_msgs::CameraInfo          -> CameraInfo {width, height, P[12]}
//   ros::Time                        -> double seconds (unused by static providers)
// A ROS adapter only has to fill these from its node handle / tf listener / messages
// (INTEGRATION.md).  Header-only; link with librtuf.so.
#pragma once

#include <cmath>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "../rtuf.h"
#include "host.hpp"

namespace realtime_urdf_filter {

using rtuf_host::Transform;
using rtuf_host::TransformProvider;

struct CameraInfo {
  int width = 0, height = 0;
  double P[12] = {0};      // row-major 3x4 projection matrix of sensor_msgs/CameraInfo
};

// ---- renderable.h ---------------------------------------------------------------------------
struct Renderable {
  virtual ~Renderable() = default;
  void setLinkName(std::string n) { name = std::move(n); }
  std::string name;
  Transform link_offset;
  Transform link_to_fixed;
  std::vector<rtuf_host::DrawCall> draws;       // what render() hands to the GPU
  // applyTransform (src/renderable.cpp:59-68): (link_to_fixed * link_offset).getOpenGLMatrix()
  void gl_matrix(double g[16]) const { (link_to_fixed * link_offset).opengl_matrix(g); }
};
struct RenderableBox : Renderable {
  RenderableBox(float dimx_, float dimy_, float dimz_) : dimx(dimx_), dimy(dimy_), dimz(dimz_) { draws = rtuf_host::box_draws(dimx, dimy, dimz); }
  float dimx, dimy, dimz;
};
struct RenderableSphere : Renderable {
  explicit RenderableSphere(float radius_) : radius(radius_) { draws = rtuf_host::sphere_draws(radius); }
  float radius;
};
struct RenderableCylinder : Renderable {
  RenderableCylinder(float radius_, float length_) : radius(radius_), length(length_) { draws = rtuf_host::cylinder_draws(radius, length); }
  float radius, length;
};
// Resolves a mesh URI (package://, file://, ...) to file contents; returns false when unavailable
// (resource_retriever in the reference).
using MeshResolver = bool (*)(const std::string& uri, std::string& data, void* user);
struct RenderableMesh : Renderable {
  // src/renderable.cpp:306-322: a mesh that cannot be loaded becomes a renderable that draws nothing
  RenderableMesh(const std::string& meshname, float sx, float sy, float sz, MeshResolver resolve, void* user)
  {
    std::string data;
    std::vector<float> v;
    std::vector<uint32_t> t;
    if (!resolve) resolve = &rtuf_host::default_mesh_resolver;        // package:// against ROS_PACKAGE_PATH, file://, plain paths
    if (resolve(meshname, data, user) && rtuf_host::load_mesh(meshname, data, v, t)) draws = rtuf_host::mesh_draws(v, t, sx, sy, sz);
    else std::fprintf(stderr, "[realtime_urdf_filter] Could not load resource [%s]\n", meshname.c_str());
  }
};

// ---- urdf_renderer.h ------------------------------------------------------------------------
class URDFRenderer {
 public:
  URDFRenderer(std::string model_description, std::string tf_prefix, std::string cam_frame, std::string fixed_frame,
               const TransformProvider& tf, const std::string& geometry_type_, double scale_,
               const std::unordered_set<std::string>& ignore_, MeshResolver resolve = nullptr, void* resolve_user = nullptr)
      : model_description_(std::move(model_description)), tf_prefix_(std::move(tf_prefix)), geometry_type(geometry_type_), scale(scale_),
        ignore(ignore_), camera_frame_(std::move(cam_frame)), fixed_frame_(std::move(fixed_frame)), tf_(tf), resolve_(resolve), resolve_user_(resolve_user)
  {
    initURDFModel();
  }

  // src/urdf_renderer.cpp:173-190 incl. the stale-transform behaviour on lookup failure (quirk Q7)
  void update_link_transforms(double /*timestamp*/ = 0.0)
  {
    Transform t;
    for (auto& r : renderables_) {
      Transform looked;
      if (tf_.lookup(fixed_frame_, r->name, looked)) t = looked;
      // tf::Transform(t.getRotation(), t.getOrigin()) (src/urdf_renderer.cpp:187): matrix -> quaternion -> matrix, in double
      r->link_to_fixed = Transform::from_quaternion(t.rotation(), t.o);
    }
  }
  std::vector<std::shared_ptr<Renderable>> renderables_;

 protected:
  void initURDFModel()
  {
    try {
      loadURDFModel(rtuf_host::UrdfModel::from_string(model_description_));
    } catch (const std::exception& e) {
      std::fprintf(stderr, "[realtime_urdf_filter] URDF failed Model parse: %s\n", e.what());
    }
  }
  void loadURDFModel(const rtuf_host::UrdfModel& model)
  {
    for (const auto& l : model.links) process_link(l.second);
  }
  // src/urdf_renderer.cpp:100-169
  void process_link(const rtuf_host::UrdfLink& link)
  {
    if (ignore.count(link.name)) return;
    const std::vector<rtuf_host::UrdfVisual>* items = nullptr;
    if (geometry_type.empty() || geometry_type == "visual") items = &link.visual_array;
    else if (geometry_type == "collision") items = &link.collision_array;
    else { std::fprintf(stderr, "[realtime_urdf_filter] invalid geometry type: %s\n", geometry_type.c_str()); return; }
    for (const auto& it : *items) {
      const rtuf_host::UrdfGeometry& g = it.geometry;
      std::shared_ptr<Renderable> r;
      switch (g.kind) {
        case rtuf_host::UrdfGeometry::BOX: r = std::make_shared<RenderableBox>((float)(scale * g.size.x), (float)(scale * g.size.y), (float)(scale * g.size.z)); break;
        case rtuf_host::UrdfGeometry::CYLINDER: r = std::make_shared<RenderableCylinder>((float)(scale * g.radius), (float)(scale * g.length)); break;
        case rtuf_host::UrdfGeometry::SPHERE: r = std::make_shared<RenderableSphere>((float)(scale * g.radius)); break;
        case rtuf_host::UrdfGeometry::MESH:
          r = std::make_shared<RenderableMesh>(g.filename, (float)(scale * g.scale.x), (float)(scale * g.scale.y), (float)(scale * g.scale.z), resolve_, resolve_user_);
          break;
      }
      r->setLinkName(tf_prefix_ + "/" + link.name);
      r->link_offset = rtuf_host::pose_to_transform(it.xyz, it.rpy);
      renderables_.push_back(r);
    }
  }

  std::string model_description_, tf_prefix_;
  const std::string geometry_type;
  const double scale;
  const std::unordered_set<std::string> ignore;
  std::string camera_frame_, fixed_frame_;
  const TransformProvider& tf_;
  MeshResolver resolve_;
  void* resolve_user_;
};

// ---- the rosparams of the private node handle (src/urdf_filter.cpp:58-111) ------------------
struct ModelParameter {
  std::string model;            // name of the parameter that holds the URDF XML
  std::string tf_prefix;
  std::string geometry_type;    // "visual" | "collision"
  double scale = 1.0;
  std::unordered_set<std::string> ignore;
};
struct FilterParameters {
  std::string fixed_frame, camera_frame;
  double camera_offset_translation[3] = {0, 0, 0};
  double camera_offset_rotation[4] = {0, 0, 0, 1};   // x y z w
  double depth_distance_threshold = 0.05;
  bool show_gui = false;                             // accepted, ignored (no window system)
  double filter_replace_value = 0.0;
  std::vector<ModelParameter> models;
  // The reference's compile-time switch USE_OWN_CALIBRATION (src/urdf_filter.cpp:38, :462-472) as a run-time parameter: with
  // it, getProjectionMatrix ignores the CameraInfo's P (only width and height are used) and takes these intrinsics instead
  // -- the reference's hard-coded Kinect values by default -- and leaves camera_tx_ / camera_ty_ alone, as the #ifdef does.
  bool use_own_calibration = false;
  double own_calibration[4] = {585.260, 585.028, 317.387, 239.264};      // fx fy cx cy
};

// ---- urdf_filter.h --------------------------------------------------------------------------
class RealtimeURDFFilter {
 public:
  RealtimeURDFFilter(const FilterParameters& params, const TransformProvider& tf,
                     std::map<std::string, std::string> param_server, int device = 0, MeshResolver resolve = nullptr, void* resolve_user = nullptr)
      : tf_(tf), fixed_frame_(params.fixed_frame), cam_frame_(params.camera_frame), show_gui_(params.show_gui),
        depth_distance_threshold_(params.depth_distance_threshold), filter_replace_value_(params.filter_replace_value),
        params_(params), param_server_(std::move(param_server)), device_(device), resolve_(resolve), resolve_user_(resolve_user)
  {
  }
  ~RealtimeURDFFilter() { if (ctx_) rtuf_destroy(ctx_); }
  RealtimeURDFFilter(const RealtimeURDFFilter&) = delete;
  RealtimeURDFFilter& operator=(const RealtimeURDFFilter&) = delete;

  // loads URDF models (src/urdf_filter.cpp:127-197)
  void loadModels()
  {
    for (const ModelParameter& elem : params_.models) {
      auto it = param_server_.find(elem.model);
      if (it == param_server_.end()) { std::fprintf(stderr, "[realtime_urdf_filter] Parameter [%s] does not exist\n", elem.model.c_str()); continue; }
      if (it->second.empty()) { std::fprintf(stderr, "[realtime_urdf_filter] URDF is empty\n"); continue; }
      renderers_.push_back(new URDFRenderer(it->second, elem.tf_prefix, cam_frame_, fixed_frame_, tf_, elem.geometry_type, elem.scale, elem.ignore,
                                            resolve_, resolve_user_));
    }
  }

  // does virtual rendering and filtering based on depth buffer and opengl proj. matrix
  // (src/urdf_filter.cpp:207-267)
  void filter(unsigned char* buffer, double* glTf, int width, int height, double timestamp = 0.0)
  {
    if (width_ != width || height_ != height) {
      if (width_ != 0 || height_ != 0) std::fprintf(stderr, "[realtime_urdf_filter] image size has changed (%ix%i) -> (%ix%i)\n", width_, height_, width, height);
      width_ = width;
      height_ = height;
      this->initGL();
    }
    if (renderers_.empty()) return;
    textureBufferFromDepthBuffer(buffer, width_ * height_ * (int)sizeof(float));
    this->render(glTf, timestamp);
  }

  // set up the device context (the reference's OpenGL set-up, src/urdf_filter.cpp:386-436)
  void initGL()
  {
    rtuf_params p;
    rtuf_default_params(&p);
    p.near_plane = (float)near_plane_;
    p.far_plane = (float)far_plane_;
    p.depth_distance_threshold = (float)depth_distance_threshold_;
    p.filter_replace_value = (float)filter_replace_value_;
    if (ctx_) { rtuf_destroy(ctx_); ctx_ = nullptr; }
    if (rtuf_create(&ctx_, device_, width_, height_, 1, &p) != RTUF_OK) throw std::runtime_error(std::string("ERROR: could not initialize the GPU context: ") + rtuf_last_error(nullptr));
    this->loadModels();
    if (renderers_.empty()) throw std::runtime_error("Could not load any models for filtering!");
    model_ids_.clear();
    for (URDFRenderer* rd : renderers_) {
      const int m = rtuf_add_model(ctx_);
      for (const auto& r : rd->renderables_) {
        const int l = rtuf_add_link(ctx_, m);
        for (const rtuf_host::DrawCall& d : r->draws)
          check(rtuf_add_draw(ctx_, m, l, d.pre_op, d.op, d.verts.data(), (int)(d.verts.size() / 3), d.tris.data(), (int)(d.tris.size() / 3)));
      }
      model_ids_.push_back(m);
    }
    check(rtuf_finalize_models(ctx_));
    masked_depth_ = nullptr;
    mask_ = nullptr;
  }

  // compute Projection matrix from CameraInfo message (src/urdf_filter.cpp:459-501)
  void getProjectionMatrix(const CameraInfo& info, double* glTf)
  {
    if (params_.use_own_calibration) {          // (P[3] = P[7] = 0 into a scratch pair: the members keep their values)
      const double* k = params_.own_calibration;
      double tx = 0, ty = 0;
      rtuf_projection_from_intrinsics((float)k[0], (float)k[1], (float)k[2], (float)k[3], 0.0, 0.0, info.width, info.height, near_plane_, far_plane_, glTf, &tx, &ty);
      return;
    }
    rtuf_projection_from_intrinsics(info.P[0], info.P[5], info.P[2], info.P[6], info.P[3], info.P[7], info.width, info.height, near_plane_, far_plane_, glTf,
                                    &camera_tx_, &camera_ty_);
  }

  // src/urdf_filter.cpp:503-744 without GL
  void render(const double* camera_projection_matrix, double timestamp = 0.0)
  {
    if (!ctx_ || !stage_frame(camera_projection_matrix, timestamp)) return;
    check(rtuf_filter(ctx_, pending_buffer_, nullptr, width_, height_));
    masked_depth_ = rtuf_get_masked_depth(ctx_);
    if (need_mask_) mask_ = rtuf_get_mask(ctx_);
  }

  // ---- beyond the reference's surface: what a ROS adapter needs to leave the CPU out of the pixel path ----------------
  // The reference converts 16UC1 <-> 32FC1 with cv::Mat::convertTo around filter() (src/urdf_filter.cpp:280-289, :308-312);
  // here the frame goes to the GPU in the encoding it arrived in and the outputs land in the caller's buffers (e.g. the
  // data vectors of the messages to publish).  Either output may be null.  Returns false when the camera transform is not
  // available yet (nothing was written; the reference's quirk Q6 keeps the previous frame instead).
  bool filter_into(const void* depth, bool is_16uc1, double* glTf, int width, int height, double timestamp, void* masked_out, uint8_t* mask_out)
  {
    prepare(width, height);
    if (renderers_.empty() || !stage_frame(glTf, timestamp)) return false;
    if ((width_ & 3) != 0 && (is_16uc1 || !masked_out)) {
      // The fused 16UC1 kernels and the bit-packed mask need a width that is a multiple of 4.  Any other camera goes through
      // the full 32FC1 planes, with the reference's own two conversions (src/urdf_filter.cpp:287-288, :309-312: convertTo
      // 0.001 / 1000.0, round half to even, saturated) done here on the host -- slower, and right for every width.
      const size_t px = (size_t)width_ * height_;
      scratch_in_.resize(is_16uc1 ? px : 0);
      scratch_out_.resize(px);
      const float* in = static_cast<const float*>(depth);
      if (is_16uc1) {
        const uint16_t* u = static_cast<const uint16_t*>(depth);
        for (size_t i = 0; i < px; i++) scratch_in_[i] = (float)u[i] * 0.001f;
        in = scratch_in_.data();
      }
      float* out = (masked_out && !is_16uc1) ? static_cast<float*>(masked_out) : scratch_out_.data();
      check(rtuf_filter_batch(ctx_, 1, &in, &out, mask_out ? &mask_out : nullptr));
      if (masked_out && is_16uc1) {
        uint16_t* o = static_cast<uint16_t*>(masked_out);
        for (size_t i = 0; i < px; i++) {
          const float v = out[i] * 1000.0f;
          long q = 0;
          if (v >= -2147483648.0f && v < 2147483648.0f) q = std::lrintf(v);
          o[i] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
        }
      }
      return true;
    }
    if (!masked_out) {
      // mask only: one bit per pixel comes back over the bus (rtuf_filter_batch_bits*), expanded here
      bits_.resize(rtuf_mask_bits_words(width_, height_));
      uint32_t* bits = bits_.data();
      const void* in = depth;
      check(is_16uc1 ? rtuf_filter_batch_bits_u16_async(ctx_, 1, reinterpret_cast<const uint16_t* const*>(&in), &bits)
                     : rtuf_filter_batch_bits_async(ctx_, 1, reinterpret_cast<const float* const*>(&in), &bits));
      check(rtuf_sync(ctx_));
      if (mask_out) check(rtuf_expand_mask_bits(depth, is_16uc1 ? 1 : 0, bits, width_, height_, (float)filter_replace_value_, nullptr, mask_out));
      return true;
    }
    const void* in = depth;
    check(is_16uc1 ? rtuf_filter_batch_u16(ctx_, 1, reinterpret_cast<const uint16_t* const*>(&in), reinterpret_cast<uint16_t* const*>(&masked_out), mask_out ? &mask_out : nullptr)
                   : rtuf_filter_batch(ctx_, 1, reinterpret_cast<const float* const*>(&in), reinterpret_cast<float* const*>(&masked_out), mask_out ? &mask_out : nullptr));
    return true;
  }

 private:
  void prepare(int width, int height)
  {
    if (width_ == width && height_ == height) return;
    width_ = width;
    height_ = height;
    this->initGL();
  }
  // camera + link poses + uniforms of one frame (the part of render() before the draw calls, src/urdf_filter.cpp:576-632)
  bool stage_frame(const double* camera_projection_matrix, double timestamp)
  {
    Transform camera_transform;
    if (!tf_.lookup(cam_frame_, fixed_frame_, camera_transform)) {
      std::fprintf(stderr, "[realtime_urdf_filter] no transform %s <- %s\n", cam_frame_.c_str(), fixed_frame_.c_str());
      return false;                               // outputs keep the previous frame (quirk Q6)
    }
    double off[16], cam[16];
    const Transform offset = Transform::from_quaternion({params_.camera_offset_rotation[0], params_.camera_offset_rotation[1], params_.camera_offset_rotation[2], params_.camera_offset_rotation[3]},
                                                        {params_.camera_offset_translation[0], params_.camera_offset_translation[1], params_.camera_offset_translation[2]});
    offset.inverse().opengl_matrix(off);
    const Transform rot = Transform::from_quaternion(camera_transform.rotation());
    const rtuf_host::Vec3 right = rot.apply({1, 0, 0}), down = rot.apply({0, 1, 0});
    camera_transform.o.x += right.x * camera_tx_; camera_transform.o.y += right.y * camera_tx_; camera_transform.o.z += right.z * camera_tx_;
    camera_transform.o.x += down.x * camera_ty_; camera_transform.o.y += down.y * camera_ty_; camera_transform.o.z += down.z * camera_ty_;
    camera_transform.opengl_matrix(cam);
    check(rtuf_set_camera(ctx_, 0, camera_projection_matrix, off, cam));
    for (size_t i = 0; i < renderers_.size(); i++) {
      URDFRenderer* rd = renderers_[i];
      rd->update_link_transforms(timestamp);
      std::vector<double> tf(16 * rd->renderables_.size());
      for (size_t k = 0; k < rd->renderables_.size(); k++) rd->renderables_[k]->gl_matrix(&tf[16 * k]);
      if (!tf.empty()) check(rtuf_set_link_poses(ctx_, 0, model_ids_[i], tf.data(), (int)rd->renderables_.size()));
    }
    {
      // the thresholds are public members the reference re-reads every frame (uniform upload, :630-631)
      rtuf_params p;
      rtuf_default_params(&p);
      p.near_plane = (float)near_plane_;
      p.far_plane = (float)far_plane_;
      p.depth_distance_threshold = (float)depth_distance_threshold_;
      p.filter_replace_value = (float)filter_replace_value_;
      check(rtuf_set_params(ctx_, &p));
    }
    return true;
  }

 public:
  // copy char buffer to the device (src/urdf_filter.cpp:332-353): deferred to render(), which uploads
  void textureBufferFromDepthBuffer(unsigned char* buffer, int /*size_in_bytes*/) { pending_buffer_ = buffer; }

  const float* getMaskedDepth() { return masked_depth_; }

 public:
  const TransformProvider& tf_;
  std::vector<URDFRenderer*> renderers_;
  // parameters from launch file
  std::string fixed_frame_, cam_frame_;
  bool show_gui_;
  bool need_mask_ = true;
  int width_ = 0, height_ = 0;
  double camera_tx_ = 0, camera_ty_ = 0;
  double far_plane_ = 8, near_plane_ = 0.1;        // src/urdf_filter.cpp:53-54
  double depth_distance_threshold_, filter_replace_value_;
  // output from rendering (library-owned, valid until the next filter())
  const float* masked_depth_ = nullptr;
  const uint8_t* mask_ = nullptr;

 private:
  void check(int rc)
  {
    if (rc < 0) throw std::runtime_error(std::string("rtuf: ") + rtuf_last_error(ctx_));
  }
  FilterParameters params_;
  std::map<std::string, std::string> param_server_;
  int device_;
  MeshResolver resolve_;
  void* resolve_user_;
  rtuf_context* ctx_ = nullptr;
  std::vector<int> model_ids_;
  unsigned char* pending_buffer_ = nullptr;
  std::vector<uint32_t> bits_;
  std::vector<float> scratch_in_, scratch_out_;      // filter_into for widths that are not a multiple of 4
};

}  // namespace realtime_urdf_filter
