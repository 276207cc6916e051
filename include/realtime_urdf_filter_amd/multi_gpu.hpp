// multi_gpu.hpp -- one host process, one thread and one rtuf_context per GPU.
//
// Camera streams are independent, so a job is partitioned with NO data-path collective:
//   * block shares   (BASELINE config 3 / 4): streams [first, first + count) of the job on device d, geometry replicated
//   * model shares   (BASELINE config 5):     URDF m lives on device m % N, a device holds only its own robots' streams
// RCCL carries only the trivial end-of-run gather ({frames, seconds, mismatches} per device: one ncclAllGather of three
// doubles) and, optionally, the all-gather of the bit-packed masks (38 KB per VGA frame) for a consumer that wants every
// stream's mask on every GPU.  The masks can also go peer to peer: xGMI is a point-to-point fabric (7 links per GPU), so
// every device writes its slice straight into each peer's buffer with hipMemcpyPeerAsync -- one hop, all links busy --
// instead of passing through a ring.  That is only true where the runtime grants peer access: the constructor asks
// hipDeviceCanAccessPeer for every ordered pair and enables it (hipDeviceEnablePeerAccess); if any pair is refused, the
// "direct" gather falls back to RCCL and says so (GatherPath), rather than silently staging through host memory.
// Every device's host thread is pinned to the CPUs of the NUMA node its GPU hangs off (sysfs local_cpulist, the same rule
// as bench.py's pin_to_numa_node_of_gpu): its pinned staging buffers are first touched there.
//
// The reference has no multi-GPU notion at all (one GL context, one camera: src/urdf_filter.cpp:207-267); this header is
// the C++ twin of realtime_urdf_filter_amd/sharding.py + the collective lines of bench.py for hosts that stay C++ / ROS.
// LOGICAL devices (round 6): the device list may name a physical device more than once ({0, 0}: two shares, two host threads,
// two contexts, two collective streams on GPU 0).  RCCL refuses two ranks on one GPU, so such a group makes no communicator:
// its gathers travel as device-to-device copies on the members' own streams (hipMemcpyPeerAsync between members that sit
// on the same GPU is a plain copy), through the SAME padding, slot arithmetic and compaction as the RCCL path.  That is
// how the N > 1 logic of this header is exercised on a box with one GPU (tests/test_baseline_configs_gpu.py, "logical2").
// Header-only; needs the HIP runtime and RCCL headers (hipcc, -lrtuf -lrccl).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <pthread.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rtuf.h"

namespace realtime_urdf_filter {
namespace multi_gpu {

// Contiguous block partition of range(n_items): (first, count) of `rank`'s share; the first n_items % world ranks get one
// extra item.  Same rule as sharding.shard_range.
inline std::pair<int, int> shard_range(int n_items, int world, int rank)
{
  if (world <= 0 || rank < 0 || rank >= world) throw std::invalid_argument("bad world / rank");
  const int base = n_items / world, extra = n_items % world;
  return {rank * base + (rank < extra ? rank : extra), base + (rank < extra ? 1 : 0)};
}

// BASELINE config 5: URDF m lives on device m % world.  Same rule as sharding.models_for_rank.
inline std::vector<int> models_for_rank(int n_models, int world, int rank)
{
  std::vector<int> out;
  for (int m = rank; m < n_models; m += world) out.push_back(m);
  return out;
}

struct DeviceReport {
  double frames = 0, seconds = 0, mismatches = 0;      // doubles: they travel as one ncclDouble triple
};

inline void check_hip(hipError_t e, const char* what)
{
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void check_nccl(ncclResult_t r, const char* what)
{
  if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

// How the last mask all-gather travelled.
struct GatherPath {
  bool direct = false;            // peer-to-peer copies (one hop over xGMI per slice); false: ncclAllGather
  bool fell_back = false;         // direct was asked for, but some pair of devices has no peer access
  std::string note;
};

// CPUs of the NUMA node a GPU hangs off: /sys/bus/pci/devices/<domain:bus:device.function>/local_cpulist ("0-31,128-159").
// Empty when sysfs does not say (containers, unknown topology): callers then leave the thread where it is.
inline std::vector<int> local_cpus_of_device(int device)
{
  std::vector<int> cpus;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return cpus;
  std::string id(bus);
  for (char& ch : id) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');
  std::ifstream f("/sys/bus/pci/devices/" + id + "/local_cpulist");
  std::string text;
  if (!f || !std::getline(f, text)) return cpus;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t end = text.find(',', pos);
    if (end == std::string::npos) end = text.size();
    const std::string part = text.substr(pos, end - pos);
    const size_t dash = part.find('-');
    if (!part.empty()) {
      const int a = std::atoi(part.substr(0, dash).c_str());
      const int b = dash == std::string::npos ? a : std::atoi(part.substr(dash + 1).c_str());
      for (int c = a; c <= b && c < CPU_SETSIZE; c++) cpus.push_back(c);
    }
    pos = end + 1;
  }
  return cpus;
}

// Pins the calling thread to the CPUs of `device`'s NUMA node (intersected with what the process may use).  Returns the
// number of CPUs the thread now runs on, 0 when nothing was changed.
inline int pin_thread_to_numa_node_of_device(int device)
{
  const std::vector<int> cpus = local_cpus_of_device(device);
  if (cpus.empty()) return 0;
  cpu_set_t allowed, want;
  CPU_ZERO(&allowed);
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
  int n = 0;
  for (int c : cpus) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
  if (n == 0) return 0;
  return pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0 ? n : 0;
}

// The devices of one node: a context, a HIP stream for the collectives and an RCCL communicator each.
class DeviceGroup {
 public:
  // devices: HIP device ids (e.g. {0, 1, ..., 7}); streams_per_device[i]: max_streams of device i's context
  DeviceGroup(const std::vector<int>& devices, int width, int height, const std::vector<int>& streams_per_device, const rtuf_params& params)
      : devices_(devices), ctx_(devices.size(), nullptr), comm_(devices.size(), nullptr), stream_(devices.size(), nullptr),
        d_report_(devices.size(), nullptr), d_pad_(devices.size(), nullptr), d_gather_(devices.size(), nullptr),
        pad_bytes_(devices.size(), 0), gather_bytes_(devices.size(), 0), pinned_cpus_(devices.size(), 0)
  {
    if (devices.empty() || streams_per_device.size() != devices.size()) throw std::invalid_argument("device / stream lists differ in length");
    try {
      build(width, height, streams_per_device, params);
    } catch (...) {
      destroy();                                         // (a constructor that throws runs no destructor: nothing may stay behind)
      throw;
    }
  }

 private:
  void build(int width, int height, const std::vector<int>& streams_per_device, const rtuf_params& params)
  {
    for (size_t i = 0; i < devices_.size(); i++) {
      if (streams_per_device[i] <= 0) continue;          // (a device without a share keeps no context but still takes part in the collectives)
      const int rc = rtuf_create(&ctx_[i], devices_[i], width, height, streams_per_device[i], &params);
      if (rc != RTUF_OK) throw std::runtime_error("rtuf_create on device " + std::to_string(devices_[i]) + ": " + rtuf_last_error(nullptr));
    }
    for (size_t i = 0; i < devices_.size(); i++)
      for (size_t e = 0; e < i; e++) if (devices_[e] == devices_[i]) logical_ = true;
    // single-process communicator set: one rank per device, in the order of `devices` (none for logical devices: see the top)
    if (!logical_) check_nccl(ncclCommInitAll(comm_.data(), (int)devices_.size(), devices_.data()), "ncclCommInitAll");
    for (size_t i = 0; i < devices_.size(); i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      check_hip(hipStreamCreateWithFlags(&stream_[i], hipStreamNonBlocking), "hipStreamCreate");
      check_hip(hipMalloc(&d_report_[i], sizeof(double) * 3 * (1 + devices_.size())), "hipMalloc(report)");
    }
    // peer access for every ordered pair: without it hipMemcpyPeerAsync still "works", staged through host memory
    peers_ok_ = true;
    for (size_t i = 0; i < devices_.size(); i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      for (size_t e = 0; e < devices_.size(); e++) {
        if (e == i || devices_[e] == devices_[i]) continue;      // (members on one GPU reach each other's memory as it is)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices_[i], devices_[e]) != hipSuccess || !can) { (void)hipGetLastError(); peers_ok_ = false; no_peer_ += " " + std::to_string(devices_[i]) + "->" + std::to_string(devices_[e]); continue; }
        const hipError_t pe = hipDeviceEnablePeerAccess(devices_[e], 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { peers_ok_ = false; no_peer_ += " " + std::to_string(devices_[i]) + "->" + std::to_string(devices_[e]); }
        (void)hipGetLastError();
      }
    }
  }

 public:
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  ~DeviceGroup() { destroy(); }

  int size() const { return (int)devices_.size(); }
  int device(int i) const { return devices_[i]; }
  rtuf_context* context(int i) const { return ctx_[i]; }
  bool peer_access_everywhere() const { return peers_ok_; }          // every ordered pair of devices granted peer access
  bool logical_devices() const { return logical_; }                  // some physical device appears more than once: no RCCL communicator, gathers by copies
  const std::string& pairs_without_peer_access() const { return no_peer_; }
  int pinned_cpus(int i) const { return pinned_cpus_[i]; }          // CPUs device i's host thread was pinned to (0: not pinned)

  // Runs f(i) for every device on a host thread of its own (the device is current on that thread) and joins them:
  // the per-frame loop of every device -- stage poses, enqueue, retire -- runs concurrently with the others'.
  void for_each_device(const std::function<void(int)>& f)
  {
    std::vector<std::thread> th;
    std::vector<std::string> err(devices_.size());
    for (size_t i = 0; i < devices_.size(); i++)
      th.emplace_back([&, i] {
        try {
          check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
          pinned_cpus_[i] = pin_thread_to_numa_node_of_device(devices_[i]);      // (a fresh thread every call: pinned every call)
          f((int)i);
        } catch (const std::exception& e) { err[i] = e.what(); }
      });
    for (auto& t : th) t.join();
    for (size_t i = 0; i < err.size(); i++)
      if (!err[i].empty()) throw std::runtime_error("device " + std::to_string(devices_[i]) + ": " + err[i]);
  }

  // The trivial gather: every device contributes one report, all devices (and the host) get all of them.
  // One ncclAllGather of three doubles per device inside a group call (single-process multi-device use of RCCL).
  std::vector<DeviceReport> gather_reports(const std::vector<DeviceReport>& mine)
  {
    const size_t n = devices_.size();
    if (mine.size() != n) throw std::invalid_argument("one report per device");
    std::vector<double> up(3 * n);                       // (outlives the asynchronous uploads: synchronised below)
    for (size_t i = 0; i < n; i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      up[3 * i] = mine[i].frames; up[3 * i + 1] = mine[i].seconds; up[3 * i + 2] = mine[i].mismatches;
      check_hip(hipMemcpyAsync(d_report_[i], &up[3 * i], sizeof(double) * 3, hipMemcpyHostToDevice, stream_[i]), "report upload");
    }
    all_gather(reinterpret_cast<const char* const*>(d_report_.data()), [&](size_t i) { return reinterpret_cast<char*>(d_report_[i] + 3); }, 3 * sizeof(double), "reports");
    std::vector<DeviceReport> all(n);
    std::vector<double> host(3 * n);
    for (size_t i = 0; i < n; i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      check_hip(hipMemcpyAsync(host.data(), d_report_[i] + 3, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_[i]), "report download");
      check_hip(hipStreamSynchronize(stream_[i]), "hipStreamSynchronize");
      // every device must hold the same gathered table
      for (size_t k = 0; k < n; k++) {
        const DeviceReport r{host[3 * k], host[3 * k + 1], host[3 * k + 2]};
        if (i == 0) all[k] = r;
        else if (r.frames != all[k].frames || r.seconds != all[k].seconds || r.mismatches != all[k].mismatches)
          throw std::runtime_error("all-gather: devices disagree on the gathered reports");
      }
    }
    return all;
  }

  // All-gather of the bit-packed masks (rtuf_filter_batch_device_bits output): device i holds `streams[i]` frames of
  // `words` 32-bit words at d_bits[i]; afterwards d_all[i] on EVERY device holds all sum(streams) frames densely, device
  // 0's first -- the same layout whichever way the slices travelled.
  // direct = true : every device writes its slice into each peer's buffer (hipMemcpyPeerAsync: point to point over xGMI),
  //                 provided every pair has peer access; otherwise the call falls back to RCCL and says so
  // direct = false: ncclAllGather.  RCCL needs equal contributions: the slices are padded to max(streams) frames in
  //                 buffers this object owns and the gathered table is compacted into d_all on the device (callers hand
  //                 in their slices as they are).
  GatherPath all_gather_mask_bits(const std::vector<const uint32_t*>& d_bits, const std::vector<int>& streams, size_t words,
                                  const std::vector<uint32_t*>& d_all, bool direct)
  {
    const size_t n = devices_.size();
    if (d_bits.size() != n || streams.size() != n || d_all.size() != n) throw std::invalid_argument("one buffer per device");
    GatherPath path;
    path.direct = direct && peers_ok_;
    path.fell_back = direct && !peers_ok_;
    if (path.fell_back) path.note = "no peer access for" + no_peer_ + ": masks gathered over RCCL instead of peer-to-peer copies";
    int most = 0;
    for (int s : streams) most = s > most ? s : most;
    const size_t frame = words * sizeof(uint32_t);
    if (path.direct) {
      size_t first = 0;
      for (size_t i = 0; i < n; i++) {
        check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
        const size_t bytes = (size_t)streams[i] * frame;
        for (size_t e = 0; e < n && bytes; e++)
          check_hip(hipMemcpyPeerAsync(d_all[e] + first * words, devices_[e], d_bits[i], devices_[i], bytes, stream_[i]), "hipMemcpyPeerAsync");
        first += (size_t)streams[i];
      }
    } else if (most > 0) {
      for (size_t i = 0; i < n; i++) {
        check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
        grow(d_pad_[i], pad_bytes_[i], (size_t)most * frame);
        grow(d_gather_[i], gather_bytes_[i], n * (size_t)most * frame);
        if (streams[i] < most) check_hip(hipMemsetAsync(d_pad_[i] + (size_t)streams[i] * words, 0, (size_t)(most - streams[i]) * frame, stream_[i]), "pad");
        if (streams[i]) check_hip(hipMemcpyAsync(d_pad_[i], d_bits[i], (size_t)streams[i] * frame, hipMemcpyDeviceToDevice, stream_[i]), "pad copy");
      }
      all_gather(reinterpret_cast<const char* const*>(d_pad_.data()), [&](size_t i) { return reinterpret_cast<char*>(d_gather_[i]); }, (size_t)most * frame, "masks");
      for (size_t i = 0; i < n; i++) {                   // compact: slice e of the padded table -> its dense place
        check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
        size_t first = 0;
        for (size_t e = 0; e < n; e++) {
          if (streams[e]) check_hip(hipMemcpyAsync(d_all[i] + first * words, d_gather_[i] + e * (size_t)most * words, (size_t)streams[e] * frame, hipMemcpyDeviceToDevice, stream_[i]), "compact");
          first += (size_t)streams[e];
        }
      }
    }
    for (size_t i = 0; i < n; i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      check_hip(hipStreamSynchronize(stream_[i]), "hipStreamSynchronize");
    }
    return path;
  }

 private:
  // Equal contributions of `bytes` from every member (send[i] on member i) into every member's table recv(i), member 0's
  // first: one ncclAllGather per member inside a group call, or -- logical devices -- every member copying its contribution
  // into its slot of every table on its own stream (tables are complete once all streams are synchronised, as the callers do).
  template <typename Recv>
  void all_gather(const char* const* send, Recv recv, size_t bytes, const char* what)
  {
    const size_t n = devices_.size();
    if (!logical_) {
      check_nccl(ncclGroupStart(), "ncclGroupStart");
      for (size_t i = 0; i < n; i++)
        check_nccl(ncclAllGather(send[i], recv(i), bytes, ncclChar, comm_[i], stream_[i]), what);
      check_nccl(ncclGroupEnd(), "ncclGroupEnd");
      return;
    }
    for (size_t i = 0; i < n; i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      for (size_t e = 0; e < n; e++)
        check_hip(hipMemcpyPeerAsync(recv(e) + i * bytes, devices_[e], send[i], devices_[i], bytes, stream_[i]), what);
    }
    // (a member's later reads of its own table happen on ITS stream: wait for the others' copies into it)
    for (size_t i = 0; i < n; i++) {
      check_hip(hipSetDevice(devices_[i]), "hipSetDevice");
      check_hip(hipStreamSynchronize(stream_[i]), "hipStreamSynchronize");
    }
  }

  static void grow(uint32_t*& buf, size_t& have, size_t want)
  {
    if (have >= want) return;
    if (buf) { (void)hipFree(buf); buf = nullptr; have = 0; }
    check_hip(hipMalloc(&buf, want), "hipMalloc(mask gather staging)");
    have = want;
  }

  void destroy()
  {
    for (size_t i = 0; i < devices_.size(); i++) {
      if (ctx_[i]) { rtuf_destroy(ctx_[i]); ctx_[i] = nullptr; }
      (void)hipSetDevice(devices_[i]);
      if (d_report_[i]) { (void)hipFree(d_report_[i]); d_report_[i] = nullptr; }
      if (d_pad_[i]) { (void)hipFree(d_pad_[i]); d_pad_[i] = nullptr; pad_bytes_[i] = 0; }
      if (d_gather_[i]) { (void)hipFree(d_gather_[i]); d_gather_[i] = nullptr; gather_bytes_[i] = 0; }
      if (stream_[i]) { (void)hipStreamDestroy(stream_[i]); stream_[i] = nullptr; }
      if (comm_[i]) { (void)ncclCommDestroy(comm_[i]); comm_[i] = nullptr; }
    }
  }
  std::vector<int> devices_;
  std::vector<rtuf_context*> ctx_;
  std::vector<ncclComm_t> comm_;
  std::vector<hipStream_t> stream_;
  std::vector<double*> d_report_;
  std::vector<uint32_t*> d_pad_, d_gather_;      // RCCL mask gather: padded slice / gathered padded table, per device
  std::vector<size_t> pad_bytes_, gather_bytes_;
  std::vector<int> pinned_cpus_;
  bool peers_ok_ = false;
  bool logical_ = false;
  std::string no_peer_;
};

}  // namespace multi_gpu
}  // namespace realtime_urdf_filter
