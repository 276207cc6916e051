// MOCK (tests/ros_mock), not ROS: the fields of sensor_msgs/Image.
#pragma once
#include <boost/make_shared.hpp>
#include <std_msgs/Header.h>
#include <cstdint>
#include <string>
#include <vector>
namespace sensor_msgs {
struct Image {
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::string encoding;
  uint8_t is_bigendian = 0;
  uint32_t step = 0;
  std::vector<uint8_t> data;
  typedef boost::shared_ptr<Image> Ptr;
  typedef boost::shared_ptr<const Image> ConstPtr;
};
typedef boost::shared_ptr<Image> ImagePtr;
typedef boost::shared_ptr<const Image> ImageConstPtr;
}  // namespace sensor_msgs
