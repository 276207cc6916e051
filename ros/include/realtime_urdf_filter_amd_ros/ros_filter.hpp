// ros_filter.hpp -- ROS 1 adapter: the reference's node-handle-facing RealtimeURDFFilter (rosparams, TF,
// image_transport, cv_bridge) over the MI355X façade.  UNBUILT HERE (no ROS in the image); see ros/README.md.
//
// Shape follows the reference so that launch files keep working:
//   constructor      src/urdf_filter.cpp:43-118   (fixed_frame, camera_frame, camera_offset, depth_distance_threshold,
//                                                  show_gui, filter_replace_value; subscribeCamera / advertiseCamera)
//   loadModels       src/urdf_filter.cpp:127-197  (models: [{model, tf_prefix, geometry_type, scale, ignore}])
//   filter_callback  src/urdf_filter.cpp:270-330  (32FC1 or 16UC1 in; output_depth in the input encoding, output_mask MONO8)
#pragma once

#include <cv_bridge/cv_bridge.h>
#include <image_transport/image_transport.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>
#include <tf/transform_listener.h>

#include <map>
#include <memory>
#include <string>

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

class RosFilter {
 public:
  RosFilter(ros::NodeHandle& nh, int /*argc*/, char** /*argv*/) : nh_(nh), image_transport_(nh)
  {
    FilterParameters prm;
    if (!nh_.getParam("fixed_frame", prm.fixed_frame)) ROS_FATAL("fixed_frame paramter!");
    ROS_INFO("using fixed frame %s", prm.fixed_frame.c_str());
    if (!nh_.getParam("camera_frame", prm.camera_frame)) ROS_FATAL("need a camera_frame paramter!");
    ROS_INFO("using camera frame %s", prm.camera_frame.c_str());
    XmlRpc::XmlRpcValue v;
    if (nh_.getParam("camera_offset", v)) {
      ROS_ASSERT(v.getType() == XmlRpc::XmlRpcValue::TypeStruct && v.hasMember("translation") && v.hasMember("rotation"));
      XmlRpc::XmlRpcValue vec = v["translation"];
      ROS_ASSERT(vec.getType() == XmlRpc::XmlRpcValue::TypeArray && vec.size() == 3);
      for (int i = 0; i < 3; i++) prm.camera_offset_translation[i] = (double)vec[i];
      vec = v["rotation"];
      ROS_ASSERT(vec.getType() == XmlRpc::XmlRpcValue::TypeArray && vec.size() == 4);
      for (int i = 0; i < 4; i++) prm.camera_offset_rotation[i] = (double)vec[i];          // x y z w
    }
    if (!nh_.getParam("depth_distance_threshold", prm.depth_distance_threshold)) ROS_FATAL("need a depth_distance_threshold paramter!");
    nh_.param<bool>("show_gui", prm.show_gui, false);                 // accepted, ignored: no window system
    nh_.param<double>("filter_replace_value", prm.filter_replace_value, 0.0);
    int device = 0;
    nh_.param<int>("device", device, 0);                              // which GPU (new; the reference has one GL context)

    // models: the URDF XML is resolved here (getParam, then searchParam, like the reference) and handed to the
    // façade through its string map
    std::map<std::string, std::string> param_server;
    XmlRpc::XmlRpcValue models;
    nh_.getParam("models", models);
    if (models.getType() == XmlRpc::XmlRpcValue::TypeArray) {
      for (int i = 0; i < models.size(); ++i) {
        XmlRpc::XmlRpcValue elem = models[i];
        ROS_ASSERT(elem.getType() == XmlRpc::XmlRpcValue::TypeStruct);
        ModelParameter mp;
        mp.model = static_cast<std::string>(elem["model"]);
        mp.tf_prefix = static_cast<std::string>(elem["tf_prefix"]);
        mp.geometry_type = static_cast<std::string>(elem["geometry_type"]);
        mp.scale = elem.hasMember("scale") ? (double)elem["scale"] : 1.0;
        if (elem.hasMember("ignore")) {
          if (elem["ignore"].getType() == XmlRpc::XmlRpcValue::TypeArray)
            for (int k = 0; k < elem["ignore"].size(); k++) mp.ignore.insert(static_cast<std::string>(elem["ignore"][k]));
          else if (elem["ignore"].getType() == XmlRpc::XmlRpcValue::TypeString)
            mp.ignore.insert(static_cast<std::string>(elem["ignore"]));
          else
            ROS_FATAL_STREAM("invalid ignore list format: use either single string or list of strings");
        }
        std::string content, loc;
        if (!nh_.getParam(mp.model, content)) {
          if (nh_.searchParam(mp.model, loc)) nh_.getParam(loc, content);
          else { ROS_ERROR("Parameter [%s] does not exist, and was not found by searchParam()", mp.model.c_str()); continue; }
        }
        param_server[mp.model] = content;
        prm.models.push_back(mp);
      }
    } else {
      ROS_ERROR("models parameter must be an array!");
    }
    filter_.reset(new RealtimeURDFFilter(prm, tf_, param_server, device, &RosFilter::resolve_mesh, nullptr));

    depth_sub_ = image_transport_.subscribeCamera("input_depth", 10, &RosFilter::filter_callback, this);
    depth_pub_ = image_transport_.advertiseCamera("output_depth", 10);
    mask_pub_ = image_transport_.advertiseCamera("output_mask", 10);
  }

  // callback function that gets ROS images and does everything (src/urdf_filter.cpp:270-330)
  void filter_callback(const sensor_msgs::ImageConstPtr& ros_depth_image, const sensor_msgs::CameraInfo::ConstPtr& camera_info)
  {
    cv_bridge::CvImageConstPtr orig_depth_img;
    cv::Mat depth_image;
    const bool is_u16 = ros_depth_image->encoding != sensor_msgs::image_encodings::TYPE_32FC1;
    try {
      if (!is_u16) {
        orig_depth_img = cv_bridge::toCvShare(ros_depth_image, sensor_msgs::image_encodings::TYPE_32FC1);
        depth_image = orig_depth_img->image;
      } else {
        orig_depth_img = cv_bridge::toCvShare(ros_depth_image, sensor_msgs::image_encodings::TYPE_16UC1);
        orig_depth_img->image.convertTo(depth_image, CV_32F, 0.001);
      }
    } catch (const cv_bridge::Exception& e) {
      ROS_ERROR("cv_bridge Exception: %s", e.what());
      return;
    }
    if (!depth_image.isContinuous()) depth_image = depth_image.clone();
    CameraInfo info;
    info.width = (int)camera_info->width;
    info.height = (int)camera_info->height;
    for (int i = 0; i < 12; i++) info.P[i] = camera_info->P[i];
    double projection_matrix[16];
    filter_->getProjectionMatrix(info, projection_matrix);
    filter_->need_mask_ = mask_pub_.getNumSubscribers() > 0;          // src/urdf_filter.cpp:226-230
    tf_.set_stamp(ros_depth_image->header.stamp);
    try {
      filter_->filter(depth_image.data, projection_matrix, depth_image.cols, depth_image.rows, ros_depth_image->header.stamp.toSec());
    } catch (const std::runtime_error& e) {
      ROS_ERROR_STREAM(e.what());
      return;
    }
    const float* masked = filter_->getMaskedDepth();
    if (!masked) return;                                              // no transform yet: nothing rendered
    if (depth_pub_.getNumSubscribers() > 0) {
      cv::Mat masked_depth_image(camera_info->height, camera_info->width, CV_32FC1, const_cast<float*>(masked));
      cv_bridge::CvImage out_masked_depth;
      out_masked_depth.header = ros_depth_image->header;
      out_masked_depth.encoding = ros_depth_image->encoding;
      if (is_u16) masked_depth_image.convertTo(out_masked_depth.image, CV_16U, 1000.0);
      else out_masked_depth.image = masked_depth_image;
      depth_pub_.publish(out_masked_depth.toImageMsg(), camera_info);
    }
    if (mask_pub_.getNumSubscribers() > 0 && filter_->mask_) {
      cv::Mat mask_image(camera_info->height, camera_info->width, CV_8UC1, const_cast<uint8_t*>(filter_->mask_));
      cv_bridge::CvImage out_mask;
      out_mask.header = ros_depth_image->header;
      out_mask.encoding = sensor_msgs::image_encodings::MONO8;
      out_mask.image = mask_image;
      mask_pub_.publish(out_mask.toImageMsg(), camera_info);
    }
  }

 private:
  // package:// and file:// mesh URIs -> file contents (resource_retriever in the reference, src/renderable.cpp:306-322)
  static bool resolve_mesh(const std::string& uri, std::string& data, void* /*user*/);

  ros::NodeHandle nh_;
  TfProvider tf_;
  image_transport::ImageTransport image_transport_;
  image_transport::CameraSubscriber depth_sub_;
  image_transport::CameraPublisher depth_pub_;
  image_transport::CameraPublisher mask_pub_;
  std::unique_ptr<RealtimeURDFFilter> filter_;
};

}  // namespace realtime_urdf_filter
