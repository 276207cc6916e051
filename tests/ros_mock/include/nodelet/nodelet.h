// MOCK (tests/ros_mock), not ROS.
#pragma once
#include <ros/ros.h>
#define NODELET_DEBUG(...) ROS_DEBUG(__VA_ARGS__)
namespace nodelet {
class Nodelet {
 public:
  virtual ~Nodelet() {}
  virtual void onInit() = 0;
 protected:
  ros::NodeHandle& getPrivateNodeHandle() { return private_nh_; }
 private:
  ros::NodeHandle private_nh_{"~"};
};
}  // namespace nodelet
