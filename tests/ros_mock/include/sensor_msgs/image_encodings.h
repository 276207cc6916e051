// MOCK (tests/ros_mock), not ROS: the encoding names the adapter compares against (values as in sensor_msgs).
#pragma once
#include <string>
namespace sensor_msgs { namespace image_encodings {
const std::string TYPE_16UC1 = "16UC1";
const std::string TYPE_32FC1 = "32FC1";
const std::string MONO8 = "mono8";
const std::string MONO16 = "mono16";
} }
