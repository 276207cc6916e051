// Mesh URI resolution for the ROS adapter (never built against ROS; compiled against tests/ros_mock, see ros/README.md): resource_retriever does what the
// reference's Assimp IOSystem wrapper does (src/renderable.cpp:173-304).
#include "realtime_urdf_filter_amd_ros/ros_filter.hpp"

#include <resource_retriever/retriever.h>

namespace realtime_urdf_filter {

bool RosFilter::resolve_mesh(const std::string& uri, std::string& data, void*)
{
  resource_retriever::Retriever retriever;
  try {
    const resource_retriever::MemoryResource res = retriever.get(uri);
    data.assign(reinterpret_cast<const char*>(res.data.get()), res.size);
    return res.size > 0;
  } catch (const resource_retriever::Exception& e) {
    ROS_ERROR("%s", e.what());
    return false;
  }
}

}  // namespace realtime_urdf_filter
