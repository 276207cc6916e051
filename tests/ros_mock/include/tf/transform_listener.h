// MOCK (tests/ros_mock), not ROS: lookupTransform answers from a table the test fills (target <- source as 3x3 basis + origin).
#pragma once
#include <ros/ros.h>
#include <array>
#include <map>
#include <stdexcept>
#include <string>
namespace tf {
struct Vector3 { double v[3] = {0, 0, 0}; double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; } };
struct Matrix3x3 { double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; const double* operator[](int r) const { return m[r]; } };
struct StampedTransform {
  Matrix3x3 basis; Vector3 origin; ros::Time stamp_;
  const Matrix3x3& getBasis() const { return basis; }
  const Vector3& getOrigin() const { return origin; }
};
struct TransformException : public std::runtime_error { using std::runtime_error::runtime_error; };
inline std::map<std::pair<std::string, std::string>, StampedTransform>& mock_transforms() { static std::map<std::pair<std::string, std::string>, StampedTransform> t; return t; }
class TransformListener {
 public:
  void lookupTransform(const std::string& target, const std::string& source, const ros::Time& time, StampedTransform& out) const
  {
    auto it = mock_transforms().find({target, source});
    if (it == mock_transforms().end()) throw TransformException("mock tf: no transform from " + source + " to " + target);
    out = it->second;
    out.stamp_ = time;
  }
};
}  // namespace tf
