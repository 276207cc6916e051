// Stand-alone node (shape of the reference's src/realtime_urdf_filter.cpp:36-53).  Never built against ROS (compiles against tests/ros_mock), see ros/README.md.
#include <ros/ros.h>

#include "realtime_urdf_filter_amd_ros/ros_filter.hpp"

int main(int argc, char** argv)
{
  ros::init(argc, argv, "realtime_urdf_filter");
  ros::NodeHandle nh("~");
  realtime_urdf_filter::RosFilter f(nh, argc, argv);
  try {
    ros::spin();
  } catch (const std::runtime_error& e) {
    ROS_FATAL_STREAM(std::string(e.what()));
  }
  return 0;
}
