// ros_filter.hpp -- ROS 1 adapter: rosparams, TF and image_transport in front of the MI355X façade.
// Never built against ROS (none in either container); compiled against tests/ros_mock and the callback tested there, see ros/README.md.
//
// The contract it keeps, so that the reference's launch files and consumers keep working, is the interface only:
//   private parameters  fixed_frame, camera_frame, camera_offset, depth_distance_threshold, show_gui, filter_replace_value,
//                       models: [{model, tf_prefix, geometry_type, scale, ignore}]      (src/urdf_filter.cpp:43-197)
//   topics              input_depth (+ camera_info) in; output_depth in the input's encoding and output_mask (MONO8) out
//                                                                                         (src/urdf_filter.cpp:114-117, :270-330)
// How a frame travels is this library's own: no cv_bridge and no CPU-side conversion -- 16UC1 frames go through the fused
// 16UC1 kernels, results are written into the outgoing messages, a mask-only consumer costs one bit per pixel of
// device-to-host traffic.
#pragma once

#include <image_transport/image_transport.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>
#include <tf/transform_listener.h>

#include <boost/make_shared.hpp>
#include <cstddef>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "realtime_urdf_filter_amd/urdf_filter.hpp"

namespace realtime_urdf_filter {

// tf::TransformListener behind the façade's TransformProvider.  The stamp of the frame being filtered is set by
// the callback before filter() runs (the reference passes it down to every lookupTransform).
class TfProvider : public rtuf_host::TransformProvider {
 public:
  void set_stamp(const ros::Time& t) { stamp_ = t; }
  bool lookup(const std::string& target, const std::string& source, rtuf_host::Transform& out) const override
  {
    tf::StampedTransform t;
    try {
      listener_.lookupTransform(target, source, stamp_, t);
    } catch (const tf::TransformException& ex) {
      ROS_DEBUG("%s", ex.what());
      return false;                       // the façade keeps the previous transform / output (quirks Q6, Q7)
    }
    const tf::Matrix3x3& b = t.getBasis();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out.m[r][c] = b[r][c];
    out.o = {t.getOrigin().x(), t.getOrigin().y(), t.getOrigin().z()};
    return true;
  }

 private:
  mutable tf::TransformListener listener_;
  ros::Time stamp_;
};

// What the private node handle must provide (names and meaning as in the reference's launch files; `device` is new).
// One table instead of a sequence of getParam calls: the adapter's constructor walks it.
struct RosFilterConfig {
  std::string fixed_frame, camera_frame;
  double depth_distance_threshold = 0.05, filter_replace_value = 0.0;
  bool show_gui = false;               // accepted, ignored: there is no window system behind this back end
  int device = 0;                      // which GPU
};
// (pointers to members, one per parameter type: RosFilterConfig holds std::strings, so it is not standard-layout and
// offsetof on it would only be conditionally supported)
struct RosParamSpec {
  const char* name;
  bool required;
  std::string RosFilterConfig::*as_string;
  double RosFilterConfig::*as_double;
  bool RosFilterConfig::*as_bool;
  int RosFilterConfig::*as_int;
};
inline const std::vector<RosParamSpec>& ros_param_table()
{
  static const std::vector<RosParamSpec> table = {
      {"fixed_frame", true, &RosFilterConfig::fixed_frame, nullptr, nullptr, nullptr},
      {"camera_frame", true, &RosFilterConfig::camera_frame, nullptr, nullptr, nullptr},
      {"depth_distance_threshold", true, nullptr, &RosFilterConfig::depth_distance_threshold, nullptr, nullptr},
      {"filter_replace_value", false, nullptr, &RosFilterConfig::filter_replace_value, nullptr, nullptr},
      {"show_gui", false, nullptr, nullptr, &RosFilterConfig::show_gui, nullptr},
      {"device", false, nullptr, nullptr, nullptr, &RosFilterConfig::device},
  };
  return table;
}

class RosFilter {
 public:
  RosFilter(ros::NodeHandle& nh, int /*argc*/, char** /*argv*/) : nh_(nh), image_transport_(nh)
  {
    RosFilterConfig cfg;
    for (const RosParamSpec& spec : ros_param_table()) {
      bool found = false;
      if (spec.as_string) found = nh_.getParam(spec.name, cfg.*spec.as_string);
      else if (spec.as_double) found = nh_.getParam(spec.name, cfg.*spec.as_double);
      else if (spec.as_bool) found = nh_.getParam(spec.name, cfg.*spec.as_bool);
      else if (spec.as_int) found = nh_.getParam(spec.name, cfg.*spec.as_int);
      if (!found && spec.required) ROS_FATAL("private parameter ~%s is required", spec.name);
    }
    FilterParameters prm;
    prm.fixed_frame = cfg.fixed_frame; prm.camera_frame = cfg.camera_frame;
    prm.depth_distance_threshold = cfg.depth_distance_threshold; prm.filter_replace_value = cfg.filter_replace_value;
    prm.show_gui = cfg.show_gui;
    read_camera_offset(prm);
    std::map<std::string, std::string> urdf_by_param;
    read_models(prm, urdf_by_param);
    filter_.reset(new RealtimeURDFFilter(prm, tf_, urdf_by_param, cfg.device, &RosFilter::resolve_mesh, nullptr));

    depth_sub_ = image_transport_.subscribeCamera("input_depth", 10, &RosFilter::on_frame, this);
    depth_pub_ = image_transport_.advertiseCamera("output_depth", 10);
    mask_pub_ = image_transport_.advertiseCamera("output_mask", 10);
  }

  // One depth frame in, what the subscribers want out.  The frame reaches the GPU in the encoding it arrived in
  // (16UC1 millimetres or 32FC1 metres: the conversions are fused into the kernels), the results are written straight
  // into the messages to publish, and only what somebody listens to is produced: masked depth + mask, masked depth
  // alone, or -- when only output_mask has subscribers -- one bit per pixel over the bus, expanded on the host.
  // (The reference converts with cv::Mat::convertTo on the CPU before and after filter(), src/urdf_filter.cpp:280-312.)
  void on_frame(const sensor_msgs::ImageConstPtr& frame, const sensor_msgs::CameraInfo::ConstPtr& camera_info)
  {
    namespace enc = sensor_msgs::image_encodings;
    const bool u16 = frame->encoding == enc::TYPE_16UC1 || frame->encoding == enc::MONO16;
    if (!u16 && frame->encoding != enc::TYPE_32FC1) {
      ROS_ERROR_THROTTLE(5.0, "input_depth: encoding %s is neither 32FC1 nor 16UC1", frame->encoding.c_str());
      return;
    }
    const bool wants_depth = depth_pub_.getNumSubscribers() > 0, wants_mask = mask_pub_.getNumSubscribers() > 0;
    if (!wants_depth && !wants_mask) return;
    const size_t bpp = u16 ? 2 : 4, row = (size_t)frame->width * bpp;
    if (frame->is_bigendian) { ROS_ERROR_THROTTLE(5.0, "input_depth: big-endian images are not supported"); return; }
    // a message is untrusted input: its geometry must hold together before any row of it is read (cv_bridge did this for
    // the reference) -- a short data vector would otherwise be read past its end by the compaction or the upload
    if (frame->width == 0 || frame->height == 0 || (size_t)frame->step < row || frame->data.size() < (size_t)frame->step * frame->height) {
      ROS_ERROR_THROTTLE(5.0, "input_depth: malformed image (%ux%u, step %u, %zu bytes)", frame->width, frame->height, frame->step, frame->data.size());
      return;
    }
    // camera_info's own size fields only count when a driver fills them in (some leave them 0; the reference reads nothing but K / P
    // from the message, src/urdf_filter.cpp:459-501): a size that is stated and differs from the image is a mismatched pair
    if (camera_info->width != 0 && camera_info->height != 0 && (camera_info->width != frame->width || camera_info->height != frame->height)) {
      ROS_ERROR_THROTTLE(5.0, "input_depth: camera_info is for %ux%u, the image is %ux%u", camera_info->width, camera_info->height, frame->width, frame->height);
      return;
    }
    // rows must be dense for the device upload; a padded image is compacted once
    const uint8_t* pixels = frame->data.data();
    if (frame->step != row) {
      dense_.resize(row * frame->height);
      for (uint32_t y = 0; y < frame->height; y++) std::memcpy(&dense_[y * row], &frame->data[(size_t)y * frame->step], row);
      pixels = dense_.data();
    }
    CameraInfo info;
    info.width = (int)frame->width;
    info.height = (int)frame->height;
    for (int i = 0; i < 12; i++) info.P[i] = camera_info->P[i];
    double projection[16];
    filter_->getProjectionMatrix(info, projection);
    tf_.set_stamp(frame->header.stamp);

    sensor_msgs::ImagePtr depth_msg, mask_msg;
    if (wants_depth) {
      depth_msg = boost::make_shared<sensor_msgs::Image>();
      depth_msg->header = frame->header; depth_msg->encoding = frame->encoding;
      depth_msg->width = frame->width; depth_msg->height = frame->height; depth_msg->step = (uint32_t)row; depth_msg->is_bigendian = 0;
      depth_msg->data.resize(row * frame->height);
    }
    if (wants_mask) {
      mask_msg = boost::make_shared<sensor_msgs::Image>();
      mask_msg->header = frame->header; mask_msg->encoding = enc::MONO8;
      mask_msg->width = frame->width; mask_msg->height = frame->height; mask_msg->step = frame->width; mask_msg->is_bigendian = 0;
      mask_msg->data.resize((size_t)frame->width * frame->height);
    }
    bool done = false;
    try {
      done = filter_->filter_into(pixels, u16, projection, (int)frame->width, (int)frame->height, frame->header.stamp.toSec(),
                                  wants_depth ? depth_msg->data.data() : nullptr, wants_mask ? mask_msg->data.data() : nullptr);
    } catch (const std::runtime_error& e) {
      ROS_ERROR_STREAM(e.what());
      return;
    }
    if (!done) return;                      // camera transform not available yet
    if (wants_depth) depth_pub_.publish(depth_msg, boost::make_shared<sensor_msgs::CameraInfo>(*camera_info));
    if (wants_mask) mask_pub_.publish(mask_msg, boost::make_shared<sensor_msgs::CameraInfo>(*camera_info));
  }

 private:
  // ~camera_offset: {translation: [x, y, z], rotation: [x, y, z, w]}
  void read_camera_offset(FilterParameters& prm)
  {
    XmlRpc::XmlRpcValue v;
    if (!nh_.getParam("camera_offset", v)) return;
    auto numbers = [&](const char* key, double* dst, int n) {
      if (v.getType() != XmlRpc::XmlRpcValue::TypeStruct || !v.hasMember(key) || v[key].getType() != XmlRpc::XmlRpcValue::TypeArray || v[key].size() != n) {
        ROS_FATAL("~camera_offset/%s must be a list of %d numbers", key, n);
        return;
      }
      for (int i = 0; i < n; i++) dst[i] = v[key][i].getType() == XmlRpc::XmlRpcValue::TypeInt ? (double)(int)v[key][i] : (double)v[key][i];
    };
    numbers("translation", prm.camera_offset_translation, 3);
    numbers("rotation", prm.camera_offset_rotation, 4);
  }

  // ~models: [{model: <name of the parameter holding the URDF>, tf_prefix, geometry_type, scale, ignore}]
  void read_models(FilterParameters& prm, std::map<std::string, std::string>& urdf_by_param)
  {
    XmlRpc::XmlRpcValue models;
    if (!nh_.getParam("models", models) || models.getType() != XmlRpc::XmlRpcValue::TypeArray) {
      ROS_ERROR("~models must be a list of {model, tf_prefix, geometry_type[, scale][, ignore]} entries");
      return;
    }
    for (int i = 0; i < models.size(); ++i) {
      XmlRpc::XmlRpcValue& e = models[i];
      if (e.getType() != XmlRpc::XmlRpcValue::TypeStruct || !e.hasMember("model")) { ROS_ERROR("~models[%d] is not a struct with a model entry", i); continue; }
      ModelParameter mp;
      mp.model = static_cast<std::string>(e["model"]);
      if (e.hasMember("tf_prefix")) mp.tf_prefix = static_cast<std::string>(e["tf_prefix"]);
      if (e.hasMember("geometry_type")) mp.geometry_type = static_cast<std::string>(e["geometry_type"]);
      if (e.hasMember("scale")) mp.scale = e["scale"].getType() == XmlRpc::XmlRpcValue::TypeInt ? (double)(int)e["scale"] : (double)e["scale"];
      if (e.hasMember("ignore")) {
        XmlRpc::XmlRpcValue& ig = e["ignore"];
        if (ig.getType() == XmlRpc::XmlRpcValue::TypeString) mp.ignore.insert(static_cast<std::string>(ig));
        else if (ig.getType() == XmlRpc::XmlRpcValue::TypeArray) for (int k = 0; k < ig.size(); k++) mp.ignore.insert(static_cast<std::string>(ig[k]));
        else ROS_ERROR("~models[%d]/ignore: a link name or a list of link names", i);
      }
      // the URDF itself: the named parameter in this namespace, else wherever searchParam finds it (robot_description usually lives above)
      std::string xml, where;
      if (!nh_.getParam(mp.model, xml) && !(nh_.searchParam(mp.model, where) && nh_.getParam(where, xml))) {
        ROS_ERROR("~models[%d]: no parameter %s in reach of this node", i, mp.model.c_str());
        continue;
      }
      urdf_by_param[mp.model] = xml;
      prm.models.push_back(mp);
    }
  }

  // package:// and file:// mesh URIs -> file contents (resource_retriever in the reference, src/renderable.cpp:306-322)
  static bool resolve_mesh(const std::string& uri, std::string& data, void* /*user*/);

  ros::NodeHandle nh_;
  TfProvider tf_;
  image_transport::ImageTransport image_transport_;
  image_transport::CameraSubscriber depth_sub_;
  image_transport::CameraPublisher depth_pub_;
  image_transport::CameraPublisher mask_pub_;
  std::unique_ptr<RealtimeURDFFilter> filter_;
  std::vector<uint8_t> dense_;          // compacted copy of an input image whose rows are padded
};

}  // namespace realtime_urdf_filter
