// MOCK (tests/ros_mock), not ROS: file:// URIs only.
#pragma once
#include <boost/make_shared.hpp>
#include <cstdint>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
namespace resource_retriever {
struct Exception : public std::runtime_error { using std::runtime_error::runtime_error; };
struct MemoryResource { boost::shared_array<uint8_t> data; uint32_t size = 0; };
class Retriever {
 public:
  MemoryResource get(const std::string& url)
  {
    if (url.compare(0, 7, "file://") != 0) throw Exception("mock retriever: only file:// (" + url + ")");
    std::ifstream f(url.substr(7), std::ios::binary);
    if (!f) throw Exception("mock retriever: cannot open " + url);
    const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    MemoryResource r;
    r.size = (uint32_t)bytes.size();
    uint8_t* p = new uint8_t[bytes.size() + 1];
    std::copy(bytes.begin(), bytes.end(), p);
    r.data = boost::shared_array<uint8_t>(p);
    return r;
  }
};
}  // namespace resource_retriever
