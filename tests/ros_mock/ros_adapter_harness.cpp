// Test harness for ros/: the adapter's sources compiled against tests/ros_mock (a mock of the few ROS 1 types they touch --
// NOT ROS) and its camera callback driven with one frame, the way image_transport would.  What it shows: the sources
// compile, the parameter table / ~models / ~camera_offset parsing works on XmlRpc-shaped values, and on_frame() hands the
// frame to the GPU in its own encoding and fills the outgoing messages.  What it cannot show: anything about real ROS.
//   usage: ros_adapter_harness urdf depth_file W H fx fy cx cy replace encoding out_depth out_mask mode [row_padding_bytes]
//          encoding = 16UC1 | 32FC1;  mode = both | depth_only | mask_only | nobody (no subscriber: the callback must return before it touches the GPU)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>

#include "realtime_urdf_filter_amd_ros/ros_filter.hpp"
#include "../../ros/src/ros_filter.cpp"      // RosFilter::resolve_mesh (one translation unit: no library to link)

using namespace realtime_urdf_filter;

static std::string slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
  if (argc < 14) { std::fprintf(stderr, "usage: see the head of %s\n", __FILE__); return 2; }
  const std::string xml = slurp(argv[1]), depth = slurp(argv[2]);
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
  const std::string encoding = argv[10], mode = argv[13];
  const size_t bpp = encoding == "16UC1" ? 2 : 4, pad = argc > 14 ? (size_t)std::atoi(argv[14]) : 0;
  if (depth.size() != (size_t)W * H * bpp) { std::fprintf(stderr, "depth file has the wrong size\n"); return 2; }

  // parameter server: what launch/filter_parameters.yaml + a robot_description above the node's namespace provide
  auto& P = ros::mock_parameter_server();
  P["~fixed_frame"] = "/world";
  P["~camera_frame"] = "/camera_rgb_optical_frame";
  P["~depth_distance_threshold"] = 0.05;
  P["~filter_replace_value"] = std::atof(argv[9]);
  P["~show_gui"] = false;
  XmlRpc::XmlRpcValue model;
  model["model"] = "robot_description"; model["tf_prefix"] = "/EXAMPLE"; model["geometry_type"] = "visual"; model["scale"] = 1;      // (an int, as YAML gives it)
  XmlRpc::XmlRpcValue models; models[0] = model;
  P["~models"] = models;
  XmlRpc::XmlRpcValue off;
  off["translation"][0] = 0; off["translation"][1] = 0.0; off["translation"][2] = 0.0;
  off["rotation"][0] = 0.0; off["rotation"][1] = 0.0; off["rotation"][2] = 0.0; off["rotation"][3] = 1;
  P["~camera_offset"] = off;
  P["/robot_description"] = xml;                       // found through searchParam

  // tf: the link frames of the example under the prefix, the camera looking along world +y (as examples/example_filter.cpp)
  rtuf_host::StaticTransformProvider frames;
  for (const auto& kv : rtuf_host::forward_kinematics(rtuf_host::UrdfModel::from_string(xml))) frames.frames["/EXAMPLE/" + kv.first] = kv.second;
  frames.frames["/world"] = rtuf_host::Transform();
  rtuf_host::Transform cam;
  cam.m[0][0] = 1; cam.m[0][1] = 0; cam.m[0][2] = 0;
  cam.m[1][0] = 0; cam.m[1][1] = 0; cam.m[1][2] = 1;
  cam.m[2][0] = 0; cam.m[2][1] = -1; cam.m[2][2] = 0;
  frames.frames["/camera_rgb_optical_frame"] = cam;
  for (const auto& t : frames.frames)
    for (const auto& s : frames.frames) {
      rtuf_host::Transform x;
      frames.lookup(t.first, s.first, x);
      tf::StampedTransform st;
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) st.basis.m[r][c] = x.m[r][c];
      st.origin.v[0] = x.o.x; st.origin.v[1] = x.o.y; st.origin.v[2] = x.o.z;
      tf::mock_transforms()[{t.first, s.first}] = st;
    }

  if (const char* drop = std::getenv("RTUF_MOCK_DROP_PARAM")) P.erase(drop);       // (test of the required-parameter check)
  ros::NodeHandle nh("~");
  RosFilter filter(nh, argc, argv);
  auto& topics = image_transport::mock_topics();
  if (!topics.count("input_depth") || !topics["input_depth"].callback) { std::fprintf(stderr, "the adapter did not subscribe to input_depth\n"); return 1; }
  topics["output_depth"].subscribers = (mode == "mask_only" || mode == "nobody") ? 0 : 1;
  topics["output_mask"].subscribers = (mode == "depth_only" || mode == "nobody") ? 0 : 1;

  auto image = boost::make_shared<sensor_msgs::Image>();
  image->header.stamp = ros::Time(12.5); image->header.frame_id = "/camera_rgb_optical_frame";
  image->width = (uint32_t)W; image->height = (uint32_t)H; image->encoding = encoding; image->is_bigendian = 0;
  image->step = (uint32_t)((size_t)W * bpp + pad);
  image->data.assign((size_t)image->step * H, 0xcd);
  for (int y = 0; y < H; y++) std::copy(depth.begin() + (size_t)y * W * bpp, depth.begin() + (size_t)(y + 1) * W * bpp, image->data.begin() + (size_t)y * image->step);
  auto info = boost::make_shared<sensor_msgs::CameraInfo>();
  info->width = (uint32_t)W; info->height = (uint32_t)H;
  info->P[0] = std::atof(argv[5]); info->P[5] = std::atof(argv[6]); info->P[2] = std::atof(argv[7]); info->P[6] = std::atof(argv[8]); info->P[10] = 1;

  // RTUF_MOCK_SHORT_IMAGE=<bytes>: a message whose data vector is that much shorter than step * height says it is;
  // RTUF_MOCK_INFO_WIDTH=<w>: a camera_info that belongs to another image size.  The adapter must refuse both, not read on.
  if (const char* cut = std::getenv("RTUF_MOCK_SHORT_IMAGE")) image->data.resize(image->data.size() - (size_t)std::atoi(cut));
  if (const char* iw = std::getenv("RTUF_MOCK_INFO_WIDTH")) info->width = (uint32_t)std::atoi(iw);
  // RTUF_MOCK_INFO_UNSIZED=1: a driver that leaves camera_info's width / height at 0 (only K / P filled in): the frame must be filtered
  if (std::getenv("RTUF_MOCK_INFO_UNSIZED")) info->width = info->height = 0;
  topics["input_depth"].callback(image, info);

  for (const std::string& l : ros::mock_log()) std::printf("log %s\n", l.c_str());
  const auto& dp = topics["output_depth"].published; const auto& mp = topics["output_mask"].published;
  std::printf("published depth %zu mask %zu\n", dp.size(), mp.size());
  if (dp.size() != ((mode == "mask_only" || mode == "nobody") ? 0u : 1u) || mp.size() != ((mode == "depth_only" || mode == "nobody") ? 0u : 1u)) return 1;
  if (!dp.empty()) {
    const sensor_msgs::Image& m = *dp[0].first;
    if (m.encoding != encoding || m.width != (uint32_t)W || m.height != (uint32_t)H || m.step != (uint32_t)(W * bpp) || m.header.stamp.toSec() != 12.5 || dp[0].second->P[0] != info->P[0]) return 1;
    std::ofstream(argv[11], std::ios::binary).write(reinterpret_cast<const char*>(m.data.data()), (std::streamsize)m.data.size());
  }
  if (!mp.empty()) {
    const sensor_msgs::Image& m = *mp[0].first;
    if (m.encoding != "mono8" || m.step != (uint32_t)W || m.data.size() != (size_t)W * H) return 1;
    std::ofstream(argv[12], std::ios::binary).write(reinterpret_cast<const char*>(m.data.data()), (std::streamsize)m.data.size());
  }
  return 0;
}
