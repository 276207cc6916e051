// MOCK (tests/ros_mock), not ROS: camera subscriber / publishers whose traffic the test drives and inspects.
#pragma once
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <functional>
#include <map>
#include <string>
namespace image_transport {
struct MockTopic {
  uint32_t subscribers = 0;                                   // set by the test
  std::vector<std::pair<sensor_msgs::ImageConstPtr, sensor_msgs::CameraInfoConstPtr>> published;
  std::function<void(const sensor_msgs::ImageConstPtr&, const sensor_msgs::CameraInfoConstPtr&)> callback;      // subscriber side
};
inline std::map<std::string, MockTopic>& mock_topics() { static std::map<std::string, MockTopic> t; return t; }
class CameraPublisher {
 public:
  CameraPublisher() = default;
  explicit CameraPublisher(const std::string& topic) : topic_(topic) {}
  uint32_t getNumSubscribers() const { return mock_topics()[topic_].subscribers; }
  void publish(const sensor_msgs::ImageConstPtr& image, const sensor_msgs::CameraInfoConstPtr& info) const { mock_topics()[topic_].published.emplace_back(image, info); }
 private:
  std::string topic_;
};
class CameraSubscriber {};
class ImageTransport {
 public:
  explicit ImageTransport(const ros::NodeHandle&) {}
  template <class T>
  CameraSubscriber subscribeCamera(const std::string& topic, uint32_t /*queue*/,
                                   void (T::*fp)(const sensor_msgs::ImageConstPtr&, const sensor_msgs::CameraInfoConstPtr&), T* obj)
  {
    mock_topics()[topic].callback = [obj, fp](const sensor_msgs::ImageConstPtr& i, const sensor_msgs::CameraInfoConstPtr& c) { (obj->*fp)(i, c); };
    return CameraSubscriber();
  }
  CameraPublisher advertiseCamera(const std::string& topic, uint32_t /*queue*/) { mock_topics()[topic]; return CameraPublisher(topic); }
};
}  // namespace image_transport
