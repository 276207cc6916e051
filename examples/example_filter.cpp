// example_filter.cpp -- the reference's single-camera usage, on the C++ facade:
//   RealtimeURDFFilter f(params, tf, {{"robot_description", urdf_xml}});
//   f.getProjectionMatrix(info, P);  f.filter(buffer, P, w, h);  f.getMaskedDepth();  f.mask_
// (the shape of src/urdf_filtered_tracker.cpp:161-167, :201-249).  Scene = BASELINE config C1:
// camera at the world origin looking along +y, links from the URDF's fixed joints.
//
// usage: example_filter <urdf.xml> <depth.f32> <width> <height> <fx> <fy> <cx> <cy> <replace> <out_masked.f32> <out_mask.u8>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "realtime_urdf_filter_amd/urdf_filter.hpp"

using namespace realtime_urdf_filter;

static std::string slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
  return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
  if (argc == 3 && std::string(argv[1]) == "--parse") {
    // CPU-only: parse + tessellate, print what would be uploaded
    rtuf_host::StaticTransformProvider none;
    URDFRenderer rd(slurp(argv[2]), "/P", "cam", "/world", none, "visual", 1.0, {});
    size_t tris = 0, draws = 0;
    for (const auto& r : rd.renderables_) for (const auto& d : r->draws) { tris += d.tris.size() / 3; draws++; }
    std::printf("renderables=%zu draws=%zu triangles=%zu first=%s\n", rd.renderables_.size(), draws, tris,
                rd.renderables_.empty() ? "-" : rd.renderables_[0]->name.c_str());
    return 0;
  }
  if (argc >= 3 && std::string(argv[1]) == "--fk") {
    // CPU-only: forward kinematics of a URDF for joint positions given as name=value (mimic joints resolved)
    const rtuf_host::UrdfModel model = rtuf_host::UrdfModel::from_string(slurp(argv[2]));
    std::map<std::string, double> q;
    for (int i = 3; i < argc; i++) {
      const std::string a = argv[i];
      const size_t eq = a.find('=');
      if (eq != std::string::npos) q[a.substr(0, eq)] = std::atof(a.c_str() + eq + 1);
    }
    for (const auto& kv : rtuf_host::forward_kinematics(model, q)) {
      std::printf("%s", kv.first.c_str());
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) std::printf(" %.17g", kv.second.m[r][c]);
      std::printf(" %.17g %.17g %.17g\n", kv.second.o.x, kv.second.o.y, kv.second.o.z);
    }
    return 0;
  }
  if (argc >= 3 && std::string(argv[1]) == "--mesh") {
    // CPU-only: load a mesh file (STL / Collada / OBJ) the way RenderableMesh does; prints the float bits of every vertex.
    // Optional flags after the file: "no-up" (keep the file's up axis), "unit" (apply <unit meter>)
    rtuf_host::MeshOptions opt;
    for (int i = 3; i < argc; i++) {
      if (std::string(argv[i]) == "no-up") opt.up_axis_to_y = false;
      if (std::string(argv[i]) == "unit") opt.apply_unit = true;
    }
    std::vector<float> v;
    std::vector<uint32_t> t;
    if (!rtuf_host::load_mesh(argv[2], slurp(argv[2]), v, t, opt)) { std::printf("failed\n"); return 1; }
    std::printf("%zu %zu\n", v.size() / 3, t.size() / 3);
    for (float f : v) { uint32_t u; std::memcpy(&u, &f, 4); std::printf("%08x\n", u); }
    return 0;
  }
  // (a 12th argument "into16": the same frame once more through filter_into -- what the ROS adapter's callback uses -- as
  //  16UC1 with the caller's output planes, and mask-only through the bit-packed path; depth file then holds uint16 millimetres)
  const bool into16 = argc == 13 && std::string(argv[12]) == "into16";
  if (argc != 12 && !into16) { std::fprintf(stderr, "usage: %s urdf depth.f32 W H fx fy cx cy replace out_masked out_mask [into16]\n", argv[0]); return 2; }
  const std::string xml = slurp(argv[1]);
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
  std::string depth_bytes = slurp(argv[2]);
  if (depth_bytes.size() != (size_t)W * H * (into16 ? 2 : 4)) { std::fprintf(stderr, "depth file has the wrong size\n"); return 2; }

  // TF: fixed frame /world, links under the tf_prefix, camera optical frame at the world origin
  rtuf_host::StaticTransformProvider tf;
  const rtuf_host::UrdfModel model = rtuf_host::UrdfModel::from_string(xml);
  for (const auto& kv : rtuf_host::forward_kinematics(model)) tf.frames["/EXAMPLE/" + kv.first] = kv.second;
  tf.frames["/world"] = Transform();
  Transform cam;   // world <- camera: cam_x = world_x, cam_y = -world_z, cam_z = world_y
  cam.m[0][0] = 1; cam.m[0][1] = 0; cam.m[0][2] = 0;
  cam.m[1][0] = 0; cam.m[1][1] = 0; cam.m[1][2] = 1;
  cam.m[2][0] = 0; cam.m[2][1] = -1; cam.m[2][2] = 0;
  tf.frames["/camera_rgb_optical_frame"] = cam;

  FilterParameters prm;                                    // launch/filter_parameters.yaml
  prm.fixed_frame = "/world";
  prm.camera_frame = "/camera_rgb_optical_frame";
  prm.depth_distance_threshold = 0.05;
  prm.filter_replace_value = std::atof(argv[9]);
  ModelParameter mp;
  mp.model = "robot_description"; mp.tf_prefix = "/EXAMPLE"; mp.geometry_type = "visual"; mp.scale = 1.0;
  prm.models.push_back(mp);

  try {
    RealtimeURDFFilter filter(prm, tf, {{"robot_description", xml}});
    CameraInfo info;
    info.width = W; info.height = H;
    info.P[0] = std::atof(argv[5]); info.P[5] = std::atof(argv[6]); info.P[2] = std::atof(argv[7]); info.P[6] = std::atof(argv[8]); info.P[10] = 1;
    double P[16];
    filter.getProjectionMatrix(info, P);
    if (into16) {
      std::vector<uint16_t> masked16((size_t)W * H);
      std::vector<uint8_t> mask((size_t)W * H), mask_only((size_t)W * H);
      const bool a = filter.filter_into(depth_bytes.data(), true, P, W, H, 0.0, masked16.data(), mask.data());
      const bool b = filter.filter_into(depth_bytes.data(), true, P, W, H, 0.0, nullptr, mask_only.data());      // mask bits over the bus
      if (!a || !b) { std::fprintf(stderr, "no output\n"); return 1; }
      std::ofstream(argv[10], std::ios::binary).write(reinterpret_cast<const char*>(masked16.data()), (std::streamsize)W * H * 2);
      std::ofstream(argv[11], std::ios::binary).write(reinterpret_cast<const char*>(mask.data()), (std::streamsize)W * H);
      std::printf("filter_into: mask-only path %s the full path's mask\n", mask == mask_only ? "equals" : "DIFFERS FROM");
      return mask == mask_only ? 0 : 1;
    }
    filter.filter(reinterpret_cast<unsigned char*>(&depth_bytes[0]), P, W, H);
    const float* masked = filter.getMaskedDepth();
    if (!masked || !filter.mask_) { std::fprintf(stderr, "no output\n"); return 1; }
    std::ofstream(argv[10], std::ios::binary).write(reinterpret_cast<const char*>(masked), (std::streamsize)W * H * 4);
    std::ofstream(argv[11], std::ios::binary).write(reinterpret_cast<const char*>(filter.mask_), (std::streamsize)W * H);
    size_t n = 0;
    for (int i = 0; i < W * H; i++) n += filter.mask_[i] != 0;
    std::printf("filtered %zu of %d pixels (%zu renderables)\n", n, W * H, filter.renderers_[0]->renderables_.size());
  } catch (const std::runtime_error& e) {      // what the reference's main() catches (src/realtime_urdf_filter.cpp:46-50)
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
