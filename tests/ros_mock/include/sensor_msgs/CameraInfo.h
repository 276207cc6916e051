// MOCK (tests/ros_mock), not ROS: the fields of sensor_msgs/CameraInfo the adapter reads.
#pragma once
#include <boost/make_shared.hpp>
#include <std_msgs/Header.h>
#include <array>
namespace sensor_msgs {
struct CameraInfo {
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::array<double, 9> K{};
  std::array<double, 12> P{};
  typedef boost::shared_ptr<CameraInfo> Ptr;
  typedef boost::shared_ptr<const CameraInfo> ConstPtr;
};
typedef boost::shared_ptr<CameraInfo> CameraInfoPtr;
typedef boost::shared_ptr<const CameraInfo> CameraInfoConstPtr;
}  // namespace sensor_msgs
