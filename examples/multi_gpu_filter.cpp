// multi_gpu_filter.cpp -- a C++ host that shards a job of independent camera streams over the GPUs of one node:
// one host thread and one rtuf_context per device (realtime_urdf_filter_amd/multi_gpu.hpp), forward kinematics on each
// device, no data-path collective; RCCL carries the end-of-run gather of {frames, seconds} and the optional all-gather of
// the bit-packed masks.  What bench.py does over torch.distributed, for hosts that stay C++ (north_star: "the host stays
// C++/ROS ... independent stream batches shard across 8 GPUs with RCCL only for the trivial gather").
//
//   multi_gpu_filter <scene.bin> [--devices N] [--logical-devices N] [--mode block|model] [--steps K] [--dump S PREFIX] [--masks direct|rccl|none]
//
// scene.bin (little endian; written by tests/scene_file.py from a bench workload) holds the job: image size, filter
// parameters, models (links -> draws -> vertices / triangles), kinematic trees, per-stream model selection, cameras, joint
// positions or explicit link matrices, and the sensor frames.
//   --mode block   streams are block-partitioned over the devices, every device loads every model   (BASELINE config 3 / 4)
//   --mode model   model m and the streams that render it live on device m % N                      (BASELINE config 5)
//   --logical-devices N   N shares, host threads and contexts mapped onto the visible devices in turn (N = 2 on a box with one GPU:
//                  both on device 0) -- the N > 1 logic without N GPUs; no RCCL communicator then, the gathers travel as copies
//                  through the same padding / compaction (multi_gpu.hpp)
//   --dump S P     writes the job's stream S after the last step: P.masked.f32, P.mask.u8 and the link matrices / camera
//                  transform the device rendered it with (P.link_tf.f64, P.cam_tf.f64), for a checker
// Prints one JSON line: frames/s of the job, per-device shares, what RCCL gathered.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "realtime_urdf_filter_amd/multi_gpu.hpp"

namespace mg = realtime_urdf_filter::multi_gpu;

namespace {

struct Reader {
  std::vector<char> buf;
  size_t pos = 0;
  explicit Reader(const char* path)
  {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    buf.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  template <typename T> T one() { T v; take(&v, 1); return v; }
  template <typename T> void take(T* dst, size_t n)
  {
    if (pos + n * sizeof(T) > buf.size()) { std::fprintf(stderr, "scene file truncated\n"); std::exit(2); }
    std::memcpy(dst, buf.data() + pos, n * sizeof(T));
    pos += n * sizeof(T);
  }
  template <typename T> std::vector<T> vec(size_t n) { std::vector<T> v(n); if (n) take(v.data(), n); return v; }
};

struct Draw { int pre_op; float op[3]; std::vector<float> verts; std::vector<uint32_t> tris; };
struct Model {
  std::vector<std::vector<Draw>> links;
  bool has_kin = false;
  int n_frames = 0, camera_frame = -1;
  std::vector<int32_t> parent, type, link_frame;
  std::vector<double> origin, axis, link_offset;
  std::vector<double> q;          // [streams][n_frames]          (has_kin)
  std::vector<double> link_tf;    // [streams][links][16]         (!has_kin)
};
struct Scene {
  int width = 0, height = 0, n_streams = 0;
  float near_plane = 0.1f, far_plane = 8.0f, max_diff = 0.05f, replace_value = 0.0f;
  std::vector<Model> models;
  std::vector<int32_t> model_of_stream;      // -1: the stream renders every model
  std::vector<double> projection, offset_inv, cam_tf;      // [streams][16]
  int n_depth = 0;
  std::vector<float> depth;                  // [n_depth][H][W]; stream s sees frame s % n_depth
};

Scene load_scene(const char* path)
{
  Reader r(path);
  char magic[8];
  r.take(magic, 8);
  if (std::memcmp(magic, "RTUFSCN1", 8) != 0) { std::fprintf(stderr, "%s is not a scene file\n", path); std::exit(2); }
  Scene s;
  s.width = r.one<int32_t>(); s.height = r.one<int32_t>(); s.n_streams = r.one<int32_t>();
  const int n_models = r.one<int32_t>();
  s.near_plane = r.one<float>(); s.far_plane = r.one<float>(); s.max_diff = r.one<float>(); s.replace_value = r.one<float>();
  s.models.resize(n_models);
  for (Model& m : s.models) {
    m.links.resize(r.one<int32_t>());
    for (auto& link : m.links) {
      link.resize(r.one<int32_t>());
      for (Draw& d : link) {
        d.pre_op = r.one<int32_t>();
        r.take(d.op, 3);
        const int nv = r.one<int32_t>(), nt = r.one<int32_t>();
        d.verts = r.vec<float>(3 * (size_t)nv);
        d.tris = r.vec<uint32_t>(3 * (size_t)nt);
      }
    }
    m.has_kin = r.one<int32_t>() != 0;
    if (m.has_kin) {
      m.n_frames = r.one<int32_t>();
      m.parent = r.vec<int32_t>(m.n_frames); m.type = r.vec<int32_t>(m.n_frames);
      m.origin = r.vec<double>(16 * (size_t)m.n_frames); m.axis = r.vec<double>(3 * (size_t)m.n_frames);
      m.link_frame = r.vec<int32_t>(m.links.size()); m.link_offset = r.vec<double>(16 * m.links.size());
      m.camera_frame = r.one<int32_t>();
    }
  }
  s.model_of_stream = r.vec<int32_t>(s.n_streams);
  s.projection = r.vec<double>(16 * (size_t)s.n_streams);
  s.offset_inv = r.vec<double>(16 * (size_t)s.n_streams);
  s.cam_tf = r.vec<double>(16 * (size_t)s.n_streams);
  for (Model& m : s.models) {
    if (m.has_kin) m.q = r.vec<double>((size_t)s.n_streams * m.n_frames);
    else m.link_tf = r.vec<double>((size_t)s.n_streams * m.links.size() * 16);
  }
  s.n_depth = r.one<int32_t>();
  s.depth = r.vec<float>((size_t)s.n_depth * s.width * s.height);
  return s;
}

#define RTUF_CHECK(ctx, expr)                                                                                  \
  do { const int rc_ = (expr); if (rc_ < 0) throw std::runtime_error(std::string(#expr) + ": " + rtuf_last_error(ctx)); } while (0)

// What one device holds: which of the job's streams and models, and its buffers.
struct Share {
  std::vector<int> streams;         // global stream numbers, in local order
  std::vector<int> models;          // global model numbers this device loads, in local model-id order
  float* d_depth = nullptr; float* d_masked = nullptr; uint8_t* d_mask = nullptr; uint32_t* d_bits = nullptr;
};

}  // namespace

int main(int argc, char** argv)
{
  if (argc < 2) { std::fprintf(stderr, "usage: %s scene.bin [--devices N | --all-devices] [--mode block|model] [--steps K] [--dump S PREFIX] [--masks direct|rccl|none]\n", argv[0]); return 2; }
  int want_devices = 0, logical_devices = 0, steps = 3, dump_stream = -1;
  std::string mode = "block", dump_prefix, masks = "direct";
  for (int i = 2; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--devices" && i + 1 < argc) want_devices = std::atoi(argv[++i]);
    else if (a == "--all-devices") want_devices = 0;                       // every visible device (also the default)
    else if (a == "--logical-devices" && i + 1 < argc) logical_devices = std::atoi(argv[++i]);
    else if (a == "--mode" && i + 1 < argc) mode = argv[++i];
    else if (a == "--steps" && i + 1 < argc) steps = std::atoi(argv[++i]);
    else if (a == "--masks" && i + 1 < argc) masks = argv[++i];
    else if (a == "--dump" && i + 2 < argc) { dump_stream = std::atoi(argv[++i]); dump_prefix = argv[++i]; }
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  try {
    const Scene sc = load_scene(argv[1]);
    int visible = 0;
    mg::check_hip(hipGetDeviceCount(&visible), "hipGetDeviceCount");
    const int N = logical_devices > 0 ? logical_devices : (want_devices > 0 ? want_devices : visible);
    if (N <= 0 || visible <= 0 || (logical_devices <= 0 && N > visible)) { std::fprintf(stderr, "%d devices requested, %d visible\n", N, visible); return 2; }
    std::vector<int> devices(N);
    for (int d = 0; d < N; d++) devices[d] = d % visible;                  // (logical devices wrap around the visible ones)

    // ---- the partition (no collective needed: streams are independent) ----------------------------------------
    std::vector<Share> share(N);
    const int M = (int)sc.models.size();
    for (int d = 0; d < N; d++) {
      if (mode == "model") {
        share[d].models = mg::models_for_rank(M, N, d);
        for (int s = 0; s < sc.n_streams; s++)
          for (int m : share[d].models) if (sc.model_of_stream[s] == m) share[d].streams.push_back(s);
      } else {
        const auto fc = mg::shard_range(sc.n_streams, N, d);
        for (int s = fc.first; s < fc.first + fc.second; s++) share[d].streams.push_back(s);
        for (int m = 0; m < M; m++) share[d].models.push_back(m);
      }
    }
    std::vector<int> per_device(N);
    for (int d = 0; d < N; d++) per_device[d] = (int)share[d].streams.size();

    rtuf_params p;
    rtuf_default_params(&p);
    p.near_plane = sc.near_plane; p.far_plane = sc.far_plane;
    p.depth_distance_threshold = sc.max_diff; p.filter_replace_value = sc.replace_value;
    mg::DeviceGroup group(devices, sc.width, sc.height, per_device, p);
    const size_t plane = (size_t)sc.width * sc.height, words = rtuf_mask_bits_words(sc.width, sc.height);

    // ---- load: geometry, kinematic trees, cameras, sensor frames (every device on its own thread) -------------
    group.for_each_device([&](int d) {
      Share& sh = share[d];
      rtuf_context* ctx = group.context(d);
      const int n = (int)sh.streams.size();
      if (!ctx || n == 0) return;
      for (int gm : sh.models) {
        const Model& m = sc.models[gm];
        const int id = rtuf_add_model(ctx);
        RTUF_CHECK(ctx, id);
        for (const auto& link : m.links) {
          const int l = rtuf_add_link(ctx, id);
          RTUF_CHECK(ctx, l);
          for (const Draw& dr : link)
            RTUF_CHECK(ctx, rtuf_add_draw(ctx, id, l, dr.pre_op, dr.op, dr.verts.data(), (int)(dr.verts.size() / 3), dr.tris.data(), (int)(dr.tris.size() / 3)));
        }
      }
      RTUF_CHECK(ctx, rtuf_finalize_models(ctx));
      for (size_t lm = 0; lm < sh.models.size(); lm++) {
        const Model& m = sc.models[sh.models[lm]];
        if (m.has_kin)
          RTUF_CHECK(ctx, rtuf_set_kinematics(ctx, (int)lm, m.n_frames, m.parent.data(), m.type.data(), m.origin.data(), m.axis.data(),
                                              m.link_frame.data(), m.link_offset.data(), (int)m.links.size()));
      }
      for (int ls = 0; ls < n; ls++) {
        const int gs = sh.streams[ls];
        const bool fk_camera = [&] { for (int gm : sh.models) if (sc.models[gm].has_kin && sc.models[gm].camera_frame >= 0) return true; return false; }();
        RTUF_CHECK(ctx, rtuf_set_camera(ctx, ls, &sc.projection[16 * (size_t)gs], &sc.offset_inv[16 * (size_t)gs], fk_camera ? nullptr : &sc.cam_tf[16 * (size_t)gs]));
        if (sc.model_of_stream[gs] >= 0) {
          int local = -1;
          for (size_t lm = 0; lm < sh.models.size(); lm++) if (sh.models[lm] == sc.model_of_stream[gs]) local = (int)lm;
          if (local < 0) throw std::runtime_error("a stream's model is not on its device");
          RTUF_CHECK(ctx, rtuf_set_stream_models(ctx, ls, &local, 1));
        }
      }
      mg::check_hip(hipMalloc(&sh.d_depth, (size_t)n * plane * sizeof(float)), "hipMalloc(depth)");
      mg::check_hip(hipMalloc(&sh.d_masked, (size_t)n * plane * sizeof(float)), "hipMalloc(masked)");
      mg::check_hip(hipMalloc(&sh.d_mask, (size_t)n * plane), "hipMalloc(mask)");
      mg::check_hip(hipMalloc(&sh.d_bits, (size_t)n * words * sizeof(uint32_t)), "hipMalloc(bits)");
      for (int ls = 0; ls < n; ls++)
        mg::check_hip(hipMemcpy(sh.d_depth + (size_t)ls * plane, &sc.depth[(size_t)(sh.streams[ls] % sc.n_depth) * plane], plane * sizeof(float), hipMemcpyHostToDevice), "depth upload");
    });

    // ---- the per-frame loop: stage poses, enqueue, (the context keeps two batches in flight) --------------------
    std::vector<mg::DeviceReport> mine(N);
    group.for_each_device([&](int d) {
      Share& sh = share[d];
      rtuf_context* ctx = group.context(d);
      const int n = (int)sh.streams.size();
      if (!ctx || n == 0) return;
      auto stage = [&]() {
        for (size_t lm = 0; lm < sh.models.size(); lm++) {
          const Model& m = sc.models[sh.models[lm]];
          // gather this device's streams' rows (block shares are contiguous in the job, model shares are not)
          const size_t row = m.has_kin ? (size_t)m.n_frames : m.links.size() * 16;
          const std::vector<double>& src = m.has_kin ? m.q : m.link_tf;
          std::vector<double> rows((size_t)n * row);
          for (int ls = 0; ls < n; ls++) std::memcpy(&rows[(size_t)ls * row], &src[(size_t)sh.streams[ls] * row], row * sizeof(double));
          if (m.has_kin) RTUF_CHECK(ctx, rtuf_set_joint_positions(ctx, 0, n, (int)lm, rows.data(), nullptr, m.camera_frame));
          else if (!m.links.empty()) RTUF_CHECK(ctx, rtuf_set_link_poses_batch(ctx, 0, n, (int)lm, rows.data(), (int)m.links.size()));
        }
      };
      stage();
      RTUF_CHECK(ctx, rtuf_filter_batch_device(ctx, n, sh.d_depth, sh.d_masked, sh.d_mask));      // warm-up: sizes the working set
      RTUF_CHECK(ctx, rtuf_sync(ctx));
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < steps; k++) {
        RTUF_CHECK(ctx, rtuf_filter_batch_device(ctx, n, sh.d_depth, sh.d_masked, sh.d_mask));
        stage();                                                                                    // the next frame's poses, while the GPU works
      }
      RTUF_CHECK(ctx, rtuf_sync(ctx));
      mine[d].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      mine[d].frames = (double)n * steps;
      if (masks != "none") {
        RTUF_CHECK(ctx, rtuf_filter_batch_device_bits(ctx, n, sh.d_depth, sh.d_bits));
        RTUF_CHECK(ctx, rtuf_sync(ctx));
        // the bits must be the byte mask of the full-plane call
        std::vector<uint32_t> hb((size_t)n * words);
        std::vector<uint8_t> hm((size_t)n * plane);
        mg::check_hip(hipMemcpy(hb.data(), sh.d_bits, hb.size() * 4, hipMemcpyDeviceToHost), "bits download");
        mg::check_hip(hipMemcpy(hm.data(), sh.d_mask, hm.size(), hipMemcpyDeviceToHost), "mask download");
        const int row_words = (sc.width + 31) / 32;
        for (int ls = 0; ls < n; ls++)
          for (int y = 0; y < sc.height; y++)
            for (int x = 0; x < sc.width; x++) {
              const bool bit = (hb[(size_t)ls * words + (size_t)y * row_words + (x >> 5)] >> (x & 31)) & 1u;
              if (bit != (hm[(size_t)ls * plane + (size_t)y * sc.width + x] != 0)) mine[d].mismatches += 1;
            }
      }
    });

    // ---- RCCL: the trivial gather, and the masks on every device --------------------------------------------------
    const std::vector<mg::DeviceReport> all = group.gather_reports(mine);
    double frames = 0, slowest = 0, mismatches = 0;
    for (const auto& r : all) { frames += r.frames; slowest = r.seconds > slowest ? r.seconds : slowest; mismatches += r.mismatches; }
    long long gathered_ok = -1;
    mg::GatherPath gather_path;
    if (masks != "none") {
      const bool direct = masks == "direct";
      std::vector<uint32_t*> d_all(N, nullptr);
      std::vector<const uint32_t*> d_in(N, nullptr);
      for (int d = 0; d < N; d++) {
        mg::check_hip(hipSetDevice(devices[d]), "hipSetDevice");
        mg::check_hip(hipMalloc(&d_all[d], (size_t)sc.n_streams * words * sizeof(uint32_t)), "hipMalloc(all masks)");
        d_in[d] = share[d].d_bits;                     // slices as they are: the RCCL path pads and compacts internally
      }
      gather_path = group.all_gather_mask_bits(d_in, per_device, words, d_all, direct);
      // every device now holds every stream's mask, densely, device 0's streams first: compare each table with the sources
      gathered_ok = 1;
      std::vector<uint32_t> ref((size_t)sc.n_streams * words), got;
      size_t first = 0;
      for (int d = 0; d < N; d++) {
        mg::check_hip(hipSetDevice(devices[d]), "hipSetDevice");
        if (per_device[d]) mg::check_hip(hipMemcpy(&ref[first * words], share[d].d_bits, (size_t)per_device[d] * words * 4, hipMemcpyDeviceToHost), "ref download");
        first += (size_t)per_device[d];
      }
      for (int d = 0; d < N; d++) {
        mg::check_hip(hipSetDevice(devices[d]), "hipSetDevice");
        got.assign((size_t)sc.n_streams * words, 0u);
        mg::check_hip(hipMemcpy(got.data(), d_all[d], got.size() * 4, hipMemcpyDeviceToHost), "gathered download");
        if (std::memcmp(got.data(), ref.data(), got.size() * 4) != 0) gathered_ok = 0;
        mg::check_hip(hipFree(d_all[d]), "hipFree");
      }
    }

    // ---- optional dump of one stream for a checker -----------------------------------------------------------------
    if (dump_stream >= 0) {
      for (int d = 0; d < N; d++)
        for (size_t ls = 0; ls < share[d].streams.size(); ls++) {
          if (share[d].streams[ls] != dump_stream) continue;
          mg::check_hip(hipSetDevice(devices[d]), "hipSetDevice");
          rtuf_context* ctx = group.context(d);
          const int n = (int)share[d].streams.size();
          int total_links = 0;
          for (int gm : share[d].models) total_links += (int)sc.models[gm].links.size();
          std::vector<float> masked(plane);
          std::vector<uint8_t> mask(plane);
          std::vector<double> link_tf((size_t)n * (total_links ? total_links : 1) * 16), cam((size_t)n * 16);
          mg::check_hip(hipMemcpy(masked.data(), share[d].d_masked + ls * plane, plane * 4, hipMemcpyDeviceToHost), "dump masked");
          mg::check_hip(hipMemcpy(mask.data(), share[d].d_mask + ls * plane, plane, hipMemcpyDeviceToHost), "dump mask");
          RTUF_CHECK(ctx, rtuf_debug_read_poses(ctx, n, link_tf.data(), cam.data()));
          auto put = [&](const std::string& name, const void* pdata, size_t bytes) {
            std::ofstream f(dump_prefix + name, std::ios::binary);
            f.write(static_cast<const char*>(pdata), (std::streamsize)bytes);
          };
          put(".masked.f32", masked.data(), plane * 4);
          put(".mask.u8", mask.data(), plane);
          put(".link_tf.f64", &link_tf[ls * (size_t)(total_links ? total_links : 1) * 16], (size_t)(total_links ? total_links : 1) * 16 * 8);
          put(".cam_tf.f64", &cam[ls * 16], 16 * 8);
          // which of the job's models the matrices belong to, in order
          std::string order;
          for (int gm : share[d].models) order += std::to_string(gm) + " ";
          put(".models.txt", order.data(), order.size());
        }
    }
    for (int d = 0; d < N; d++) {
      mg::check_hip(hipSetDevice(devices[d]), "hipSetDevice");
      if (share[d].d_depth) { (void)hipFree(share[d].d_depth); (void)hipFree(share[d].d_masked); (void)hipFree(share[d].d_mask); (void)hipFree(share[d].d_bits); }
    }
    std::printf("{\"devices\": %d, \"mode\": \"%s\", \"streams\": %d, \"steps\": %d, \"frames\": %.0f, \"seconds_slowest_device\": %.6f, \"frames_per_s\": %.1f, "
                "\"bits_vs_bytes_mismatches\": %.0f, \"mask_all_gather\": \"%s\", \"mask_all_gather_path\": \"%s\", \"peer_access_everywhere\": %d, "
                "\"gathered_masks_equal_sources\": %lld, \"logical_devices\": %d, \"per_device\": [",
                N, mode.c_str(), sc.n_streams, steps, frames, slowest, slowest > 0 ? frames / slowest : 0.0, mismatches, masks.c_str(),
                masks == "none" ? "none" : (gather_path.direct ? "peer-to-peer copies" : (gather_path.fell_back ? "rccl (fell back: a device pair has no peer access)" : (group.logical_devices() ? "padded all-gather by copies (logical devices: no communicator)" : "rccl"))),
                group.peer_access_everywhere() ? 1 : 0, gathered_ok, group.logical_devices() ? 1 : 0);
    for (int d = 0; d < N; d++)
      std::printf("%s{\"device\": %d, \"streams\": %d, \"frames\": %.0f, \"seconds\": %.6f, \"host_thread_pinned_to_cpus\": %d}", d ? ", " : "", devices[d], per_device[d], all[d].frames, all[d].seconds, group.pinned_cpus(d));
    std::printf("]}\n");
    return (mismatches == 0 && gathered_ok != 0) ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "multi_gpu_filter: %s\n", e.what());
    return 1;
  }
}
