// Nodelet (shape of the reference's src/realtime_urdf_filter_nodelet.cpp:35-75; same plugin class name, so
// `nodelet load realtime_urdf_filter/RealtimeURDFFilterNodelet <manager>` keeps working).  No argv juggling: there
// is no GLUT to initialise.  Never built against ROS (compiles against tests/ros_mock), see ros/README.md.
#include <nodelet/nodelet.h>
#include <pluginlib/class_list_macros.h>

#include <memory>

#include "realtime_urdf_filter_amd_ros/ros_filter.hpp"

namespace realtime_urdf_filter {

class RealtimeURDFFilterNodelet : public nodelet::Nodelet {
 public:
  void onInit() override
  {
    NODELET_DEBUG("Initializing nodelet...");
    ros::NodeHandle nh = this->getPrivateNodeHandle();
    filter_.reset(new RosFilter(nh, 0, nullptr));
  }

 private:
  std::unique_ptr<RosFilter> filter_;
};

}  // namespace realtime_urdf_filter

PLUGINLIB_EXPORT_CLASS(realtime_urdf_filter::RealtimeURDFFilterNodelet, nodelet::Nodelet);
