// MOCK of the part of roscpp that ros/include/realtime_urdf_filter_amd_ros/ros_filter.hpp and ros/src/*.cpp touch
// (tests/ros_mock: lets the adapter's sources meet a compiler and its callback run in a test; it is NOT ROS).
#pragma once
#include <cstdio>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace XmlRpc {
class XmlRpcValue {
 public:
  enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString, TypeArray, TypeStruct };
  XmlRpcValue() = default;
  XmlRpcValue(bool v) : type_(TypeBoolean), b_(v) {}
  XmlRpcValue(int v) : type_(TypeInt), i_(v) {}
  XmlRpcValue(double v) : type_(TypeDouble), d_(v) {}
  XmlRpcValue(const char* v) : type_(TypeString), s_(v) {}
  XmlRpcValue(const std::string& v) : type_(TypeString), s_(v) {}
  Type getType() const { return type_; }
  int size() const { return type_ == TypeArray ? (int)a_.size() : (type_ == TypeStruct ? (int)m_.size() : 0); }
  bool hasMember(const std::string& k) const { return type_ == TypeStruct && m_.count(k) != 0; }
  XmlRpcValue& operator[](int i) { need(TypeArray); if ((int)a_.size() <= i) a_.resize((size_t)i + 1); return a_[(size_t)i]; }
  XmlRpcValue& operator[](const char* k) { need(TypeStruct); return m_[k]; }
  XmlRpcValue& operator[](const std::string& k) { need(TypeStruct); return m_[k]; }
  operator bool&() { expect(TypeBoolean); return b_; }
  operator int&() { expect(TypeInt); return i_; }
  operator double&() { expect(TypeDouble); return d_; }
  operator std::string&() { expect(TypeString); return s_; }
 private:
  void need(Type t) { if (type_ == TypeInvalid) type_ = t; expect(t); }
  void expect(Type t) const { if (type_ != t) throw std::runtime_error("XmlRpcValue: type error"); }      // (XmlRpcException in the real one)
  Type type_ = TypeInvalid;
  bool b_ = false; int i_ = 0; double d_ = 0; std::string s_;
  std::vector<XmlRpcValue> a_; std::map<std::string, XmlRpcValue> m_;
};
}  // namespace XmlRpc

namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() = default;
  explicit Time(double t) : sec((uint32_t)t), nsec((uint32_t)((t - (uint32_t)t) * 1e9)) {}
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};
// the parameter server: one map for the whole process, filled by the test ("~name" = private names of the node handle "~")
inline std::map<std::string, XmlRpc::XmlRpcValue>& mock_parameter_server() { static std::map<std::string, XmlRpc::XmlRpcValue> p; return p; }
inline std::vector<std::string>& mock_log() { static std::vector<std::string> l; return l; }
class NodeHandle {
 public:
  NodeHandle() : ns_("/") {}
  explicit NodeHandle(const std::string& ns) : ns_(ns == "~" ? "~" : ns) {}
  template <class T> bool getParam(const std::string& name, T& out) const
  {
    auto& p = mock_parameter_server();
    auto it = p.find(resolve(name));
    if (it == p.end()) return false;
    try { out = static_cast<T&>(it->second); } catch (const std::runtime_error&) { return false; }
    return true;
  }
  bool getParam(const std::string& name, XmlRpc::XmlRpcValue& out) const
  {
    auto& p = mock_parameter_server();
    auto it = p.find(resolve(name));
    if (it == p.end()) return false;
    out = it->second;
    return true;
  }
  bool searchParam(const std::string& name, std::string& where) const      // up the namespace: here "~name", then "/name"
  {
    auto& p = mock_parameter_server();
    for (const std::string& k : {resolve(name), std::string("/") + name})
      if (p.count(k)) { where = k; return true; }
    return false;
  }
 private:
  std::string resolve(const std::string& name) const { return (!name.empty() && (name[0] == '/' || name[0] == '~')) ? name : (ns_ == "~" ? "~" + name : ns_ + name); }
  std::string ns_;
};
inline void init(int&, char**, const std::string&) {}
inline void spin() {}
}  // namespace ros

#define RTUF_MOCK_LOG(level, ...) do { char b_[1024]; std::snprintf(b_, sizeof b_, __VA_ARGS__); ros::mock_log().push_back(std::string(level) + ": " + b_); } while (0)
#define ROS_DEBUG(...) RTUF_MOCK_LOG("DEBUG", __VA_ARGS__)
#define ROS_ERROR(...) RTUF_MOCK_LOG("ERROR", __VA_ARGS__)
#define ROS_FATAL(...) RTUF_MOCK_LOG("FATAL", __VA_ARGS__)
#define ROS_ERROR_THROTTLE(period, ...) RTUF_MOCK_LOG("ERROR", __VA_ARGS__)
#define ROS_ERROR_STREAM(x) do { std::ostringstream s_; s_ << x; ros::mock_log().push_back("ERROR: " + s_.str()); } while (0)
#define ROS_FATAL_STREAM(x) do { std::ostringstream s_; s_ << x; ros::mock_log().push_back("FATAL: " + s_.str()); } while (0)
