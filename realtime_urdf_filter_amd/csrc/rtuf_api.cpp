// rtuf_api.cpp -- host side of the C ABI declared in include/rtuf.h.
//
// Owns device memory, uploads geometry once, stages per-frame poses, and enqueues the
// kernels of rtuf_kernels.hip on one HIP stream per context.  There is no CPU compute
// path in this library: without a HIP device rtuf_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtuf.h"
#include "rtuf_device.h"


using namespace rtuf;

namespace {

struct HostDraw {
  int pre_op;
  float op[3];
  std::vector<float> verts;       // xyz
  std::vector<uint32_t> tris;     // 3 per triangle, local vertex ids
};
struct HostLink { std::vector<HostDraw> draws; };
static constexpr int kMaxInflight = 2;      // device batches that may be in flight at once
#ifndef RTUF_MAX_LANES
#define RTUF_MAX_LANES 3            // (an experiment's build may raise it: more lanes than the runtime has hardware queues -- GPU_MAX_HW_QUEUES, 4 by default -- share queues)
#endif
static constexpr int kMaxLanes = RTUF_MAX_LANES;         // raster lanes a context can have (rtuf_params.raster_lanes; the default is kDefaultLanes)
static constexpr int kDefaultLanes = 3;     // (three lanes + the pose stage's stream = the HIP runtime's four hardware queues)
static constexpr int kSplitMin = 32;        // batches of at least this many streams are split over the lanes; smaller ones
                                            // take one lane each, in turn (their cost is launches, not kernel time)

struct Kinematics {               // on-device forward kinematics of one model
  int n_frames = 0, camera_frame = -1, max_depth = 0;
  int32_t* d_depth = nullptr;
  bool has_root = false, any_enabled = false;
  bool dirty_q = true, dirty_aux = true;     // joint positions / root poses + enable flags changed on the host
  int32_t* d_parent = nullptr; int32_t* d_type = nullptr; double* d_origin = nullptr; double* d_axis = nullptr;
  int32_t* d_link_frame = nullptr; double* d_link_offset = nullptr;
  // joint positions are staged in a ring of pinned buffers that the forward-kinematics kernel reads in
  // place (zero-copy): every batch in flight owns the buffer it was enqueued with, rtuf_set_joint_positions
  // writes h_q[q_write], so the next frame's joint states can be staged while the GPU filters
  double* h_q[kMaxInflight + 1] = {};                      // [max_streams][n_frames]
  int q_live = 0, q_write = 0;
  int aux_uploaded_streams = 0;
  bool q_carried = true;                                    // h_q[q_write] holds everything h_q[q_live] does
  double* h_root = nullptr; double* d_root = nullptr;       // [max_streams][12]
  uint8_t* h_enabled = nullptr; uint8_t* d_enabled = nullptr;
};
struct HostModel { std::vector<HostLink> links; int link_base = 0; Kinematics kin; };

char g_create_error[512] = "";

constexpr uint32_t kKnownFlags = RTUF_FLAG_TWO_KERNEL | RTUF_FLAG_STRICT_GRID;
inline bool flags_valid(uint32_t flags)
{
#ifdef RTUF_ABLATE
  (void)flags; return true;          // timing-experiment build: bits 8.. select what the kernels skip
#else
  return (flags & ~kKnownFlags) == 0u;
#endif
}

}  // namespace

struct rtuf_context {
  int device = 0;
  int width = 0, height = 0, tiles_x = 0, tiles_y = 0, max_streams = 0;
  rtuf_params params{};
  std::string error;

  // models (host)
  std::vector<HostModel> models;
  bool finalized = false;
  bool broken = false;                 // a bin regrowth failed half-way: the bins are gone, every later filter call fails
  int n_links = 0, n_draws = 0, n_chunks = 0;
  int key_shift = 3;                   // depth keys: draw order << key_shift (32 - the bits the context's draw orders need)
  int64_t n_tris = 0, n_cverts = 0;
  uint32_t bg_chunk = 0;

  // static geometry (device)
  float4* d_cverts = nullptr; uint32_t* d_ctris = nullptr; uint32_t* d_corder = nullptr; Chunk* d_chunks = nullptr; Draw* d_draws = nullptr;

  // per-frame pose staging
  // Cameras and link matrices are staged in a ring of kMaxInflight + 1 pinned sets, like the joint positions: every
  // batch in flight owns the set it was enqueued with (a bin regrowth re-reads it), the setters write into a set no
  // batch reads, so staging the next frame's TF-derived matrices never waits for the GPU.  h_cams / h_link_tf alias the
  // set being written.
  Camera* ring_cams[kMaxInflight + 1] = {};        // pinned [max_streams]
  double* ring_link_tf[kMaxInflight + 1] = {};     // pinned [max_streams][n_links][16]
  Camera* h_cams = nullptr;
  double* h_link_tf = nullptr;
  struct StageRing {                               // one per array: cameras and link matrices advance independently
    int live = 0, write = 0;                       // set of the newest batch / set the setters write
    bool written = false;                          // the write set has changes no batch has taken yet
    bool carried = true;                           // the write set holds everything the live set does
  } cam_ring, link_ring;
  uint64_t* h_model_mask = nullptr;    // pinned [max_streams]
  uint64_t* d_model_mask = nullptr;

  // Rasteriser working set.  A RASTER LANE is a HIP stream plus the arrays one launch group is rasterised through (tile
  // bins, clip list, many-tile list, work list); the launch groups of a batch alternate between the lanes, so group B's
  // set-up kernel runs under group A's tile kernel and every kernel's ramp, tail and launch gap is filled by the other
  // lane -- what two contexts of half the streams gave a caller by hand (509 k instead of 455 k frames/s on the 256-stream
  // VGA workload in round 3), now inside one context.  Three lanes by default: with the pose stage's stream that is one HIP
  // stream per hardware queue of the runtime's default four.  Kernels of one lane
  // are ordered by its stream, which is all the hand-over its arrays need; lanes share nothing that is written per group.
  struct Lane {
    hipStream_t stream = nullptr;
    PackedTri* d_bins = nullptr; BinHeader* d_bin_hdr = nullptr;
    Frag* d_fbins = nullptr; uint32_t* d_fbin_count = nullptr;
    ClipItem* d_clip_list = nullptr;
    float4* d_clip_spill = nullptr;
    BigRec* d_big_list = nullptr;                                // many-tile records (per counter shard), see bigrec_kernel
    WorkItem* d_items[kMaxInflight] = {};                        // the launch group's set-up work list (cull_kernel -> setup_kernel), one per
                                                                 // batch slot: the cull of a lane's first group runs in the batch's pose
                                                                 // stage, under the set-up kernel of the batch before it
    float* d_zsurface = nullptr;
  };
  Lane lane[kMaxLanes];
  int n_lanes = 1;
  // (The lanes run free of each other.  Chaining the set-up kernels across the lanes, so that one group's set-up always
  // runs under the other group's tile kernel, was measured and is WORSE -- 426 k instead of 457 k frames/s at four groups,
  // 466 k instead of 478 k at two: both kernels are bound by VALU issue, and side by side the set-up kernel loses the
  // occupancy its latency hiding needs.  What the lanes buy is each kernel's ramp, tail and launch gap filled by the other
  // lane)
  bool lanes_share_queue = false;      // no stream was found that runs beside lane 0's: the lanes work, one after the other
  int next_lane = 0;                   // lane of the next batch that is not split
  int last_lane = 0;                   // lane of the newest batch's last group (debug read-back of the z-surface)
  int group = 0;                       // streams per launch group at most (= streams a lane's bins are sized for)
  int max_groups = 1;                  // launch groups a batch of max_streams streams is split into (sizes the counter blocks)
  uint32_t capacity = 0, fcapacity = 0, clip_capacity = 0;
  uint32_t big_capacity = 0;           // many-tile list, per counter shard
  uint32_t items_hint = 0; int items_hint_streams = 0;      // longest work list of the last batch's groups, and their size
  std::vector<uint32_t> group_items;   // work-list length of every launch group of the last retired batch ...
  std::vector<int> group_streams;      // ... and its streams: a group of the same index and size takes its own list length as the hint
  bool force_worst_grid = false;       // re-runs after a set-up grid that was too short: the worst-case grid, which cannot be
  size_t memory_budget = 0;            // upper bound of the lanes' tile bins (rtuf_params.memory_limit_mb or a third of the free memory)
  // every device allocation of the context goes through dev_alloc / dev_free: rtuf_stats.device_bytes is their sum
  std::unordered_map<void*, size_t> dev_blocks;
  size_t device_bytes = 0;

  // staging for the host-pointer API
  hipStream_t h2d = nullptr, d2h = nullptr;  // copy streams of the host-plane calls (rtuf_filter_batch*)
  std::vector<void*> pinned;                 // rtuf_host_alloc blocks not yet freed

  // single-stream outputs (masked_depth_ / mask_ of the reference)
  std::vector<float> single_masked; std::vector<uint8_t> single_mask;

  // Batches in flight.  Up to kMaxInflight device batches may be enqueued before the oldest is retired
  // (rtuf_sync, or the next rtuf_filter_batch_device* call when the ring is full), so the host round
  // trip of one batch overlaps the GPU work of the next.  A batch keeps what a re-run after a bin
  // regrowth needs: its buffers and which joint-position staging buffer it read.
  struct Batch {
    bool active = false;
    int n = 0; const float* depth = nullptr; float* masked = nullptr; uint8_t* mask = nullptr; bool u16 = false;
    uint32_t* bits = nullptr;                // mask-only output (1 bit per pixel) instead of masked / mask
    Counters* h_counters = nullptr;          // pinned [max_groups]: one block per launch group, filled by the copies that end the batch
    hipEvent_t done[kMaxLanes] = {};         // recorded on each lane after its copy
    uint32_t lanes_used = 0;                 // bit l: the batch has launch groups on lane l
    int n_groups = 0;                        // launch groups of the batch
    std::vector<hipEvent_t> events;          // stage timing
    std::vector<int> q_idx;                  // per model: joint-position staging buffer
    int cam_idx = 0, link_idx = 0;           // camera / link-matrix staging sets
    // The batch's pose stage (uploads, forward kinematics, matrix stacks, cull) runs on a side stream and
    // writes only buffers of its own slot, so it overlaps the raster kernels of the batch before it.
    Camera* d_cams = nullptr; double* d_link_tf = nullptr;
    float* d_mvp = nullptr; BgInfo* d_bg = nullptr;
    Counters* d_counters = nullptr;          // [max_groups]
    uint32_t* d_status = nullptr;            // the slot's status word (kStatus* in rtuf_device.h; rtuf_batch_status_device)
    bool dirty_cams = true, dirty_link_tf = true;
    int uploaded_streams = 0;
    hipEvent_t posed = nullptr;              // recorded on the side stream after the pose stage
    // small batches: captured launch sequences of this slot, keyed by the hash of the argument blocks they were
    // captured with (a slot meets a few recurring sets: the ring of joint-position buffers, the caller's output sets)
    static constexpr int kGraphs = 6;
    struct { uint64_t hash = 0; hipGraphExec_t exec = nullptr; } graphs[kGraphs];
    int graph_next = 0;
    std::vector<uint32_t> setup_grid;        // per launch group: work items its set-up launch covered (0xffffffff = the worst case)
    bool cover_pass = true;                  // this batch runs the cover pass (decided when it is first enqueued, kept for re-runs)
    int timing = 0;                          // event timing of this batch: 0 none, 1 every stage, 2 tile/compare kernel only
    // Host-plane batches (rtuf_filter_batch*): device staging of this slot, the caller's planes, and the
    // events that order upload -> kernels -> download across the copy streams.
    bool host_io = false;
    float* st_depth = nullptr; float* st_masked = nullptr; uint8_t* st_mask = nullptr; size_t st_streams = 0;
    uint32_t* st_bits = nullptr; size_t st_bits_streams = 0;
    std::vector<void*> h_masked, h_mask, h_bits;
    hipEvent_t uploaded = nullptr, downloaded = nullptr;
    bool wait_upload = false;                // the lanes wait for `uploaded` before the first kernel that reads the planes
  };
  Batch batch[kMaxInflight];
  hipStream_t side = nullptr;                // pose stages (see Batch)
  hipEvent_t fork_ev = nullptr;              // graph capture: forks the side stream off the main stream
  bool side_used_plain = false;              // the side stream carries work that was enqueued outside a graph
  // Small batches of a context that is one of several pipelines replay a captured hipGraph (one launch call instead of
  // ~12 API calls: there the host's launch cost is the limit, and the pipelines provide the overlap between batches).
  // A single-pipeline context keeps plain launches: its pose stage runs on the side stream underneath the previous
  // batch's raster kernels, which a graph on the main stream would serialise (batch = 1: 75 us plain, 94 us as a graph).
  // Cleared when the runtime refuses stream capture / instantiation.
  bool graphs_ok = false;
  uint32_t graph_evictions = 0;              // captures of the current 64-batch window that replaced a live cache entry
  uint64_t graph_hits = 0, graph_misses = 0; // replays / captures: a caller that never repeats an argument set (fresh output buffers
                                             // every frame) would pay a capture + instantiate per batch -- then graphs are switched off
  // The cover pass (bigrec_kernel<0> + the cover-aware tile kernel) pays where triangles cover whole tiles -- walls, anything
  // close to the lens -- and costs 3 % where none do (a finely tessellated robot at arm's length).  It is an optimisation
  // only (the image is the same with and without), so the context switches it by what the batches show: three batches in a
  // row without a single cover tile turn it off, every 64th batch runs it again as a probe.
  bool cover_on = true;
  int cover_idle = 0, cover_sleep = 0;
  int oldest = 0;                            // ring index of the oldest batch in flight
  int pending = 0;                           // batches in flight

  // host staging areas changed since the last upload?
  bool dirty_mask = true;
  int mask_uploaded_streams = 0;
  int last_slot = 0;                         // slot of the most recently enqueued batch (debug read-back)

  // rtuf_params.pipelines > 1: this context is only a front; `kids` are complete contexts (own streams, bins, staging,
  // geometry copy) that the batches alternate between, so one batch's small / low-occupancy kernels (pose stage, cull,
  // clip, kernel tails) overlap another batch's set-up and tile kernels.  Every setter goes to all kids (state is
  // replicated), every filter call to the next kid in turn; `order` lists the kids of the batches in flight, oldest first.
  std::vector<rtuf_context*> kids;
  int next_kid = 0, last_kid = 0;
  std::deque<int> order;

  rtuf_stats stats{};
#ifdef RTUF_LANECOUNT
  unsigned long long lane_slots[kLaneLoops] = {}, lane_live[kLaneLoops] = {};      // of the last retired batch (instrumented build)
#endif
  int timing = 0;            // 0 off, 1 every stage, 2 only around the tile (and compare) kernel, 3 = 2 on every fourth batch
  uint32_t timing_seq = 0;
  double acc_ms[6] = {0, 0, 0, 0, 0, 0};  // sums of ms_pose .. ms_total, ms_clip over the timed batches
  uint64_t acc_batches = 0;

  int fail(int code, const char* fmt, ...)
  {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    error = buf;
    return code;
  }
};

#define HIP_TRY(ctx, expr)                                                                 \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return (ctx)->fail(e_ == hipErrorOutOfMemory ? RTUF_ERR_OOM : RTUF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
static hipError_t dev_alloc(rtuf_context* c, T** p, size_t bytes)
{
  void* q = nullptr;
  const hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
  *p = e == hipSuccess ? static_cast<T*>(q) : nullptr;
  if (e == hipSuccess) { c->dev_blocks[q] = bytes; c->device_bytes += bytes; }
  return e;
}
template <typename T>
static void dev_free(rtuf_context* c, T*& p)
{
  if (!p) return;
  auto it = c->dev_blocks.find((void*)p);
  if (it != c->dev_blocks.end()) { c->device_bytes -= it->second; c->dev_blocks.erase(it); }
  (void)hipFree((void*)p);
  p = nullptr;
}

// Launch groups a batch of n streams is split into: as many as the lanes' bins need (c->group streams each at most), and,
// with several lanes, a multiple of the lanes for batches worth splitting, so that every lane gets the same amount of work.
static int groups_for(const rtuf_context* c, int n)
{
  int k = (n + c->group - 1) / std::max(c->group, 1);
  if (c->n_lanes > 1 && n >= kSplitMin) k = ((std::max(k, 1) + c->n_lanes - 1) / c->n_lanes) * c->n_lanes;      // a multiple of the lanes
  k = std::max(k, 1);
  // ... of ceil(n / k) streams each, which can come to FEWER than k groups (100 streams in 16 groups of 7 are 15): the number
  // returned is the number enqueue_batch makes -- the batch's status word starts at it and every group takes one off
  // (round 5 returned k: the word of such a batch never reached 0)
  const int per_group = (std::max(n, 1) + k - 1) / k;
  return (std::max(n, 1) + per_group - 1) / per_group;
}

static void sync_lanes(rtuf_context* c)
{
  for (int l = 0; l < c->n_lanes; l++) if (c->lane[l].stream) (void)hipStreamSynchronize(c->lane[l].stream);
}

// A working buffer of the rasteriser is too small for the batch in flight (its contents are not needed: the batch is run
// again).  The new one is allocated BEFORE the old one is freed whenever both fit; when they do not, the old one goes first.
// On failure the buffer is gone (nullptr) and the caller decides: smaller launch groups, or give up.
template <typename T>
static hipError_t realloc_dev(rtuf_context* c, T*& buf, size_t new_bytes)
{
  T* fresh = nullptr;
  hipError_t e = dev_alloc(c, &fresh, new_bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_free(c, buf);
    e = dev_alloc(c, &fresh, new_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return e; }
  } else {
    dev_free(c, buf);
  }
  buf = fresh;
  return hipSuccess;
}
template <typename T>
static int regrow(rtuf_context* c, T*& buf, size_t new_bytes, const char* what)
{
  const hipError_t e = realloc_dev(c, buf, new_bytes);
  if (e != hipSuccess) {
    c->broken = true;
    return c->fail(RTUF_ERR_OOM, "growing the %s to %zu bytes failed: %s", what, new_bytes, hipGetErrorString(e));
  }
  c->stats.regrowths++;
  return RTUF_OK;
}

static size_t lane_bins_bytes(const rtuf_context* c, int G, uint32_t cap, uint32_t fcap)
{
  return (size_t)G * (size_t)c->tiles_x * c->tiles_y * ((size_t)cap * sizeof(PackedTri) + (size_t)fcap * sizeof(Frag));
}

// The per-batch counter blocks (one per launch group) follow the number of groups a full batch is split into.
static int alloc_counter_blocks(rtuf_context* c)
{
  c->max_groups = groups_for(c, c->max_streams);
  for (auto& b : c->batch) {
    dev_free(c, b.d_counters);
    if (b.h_counters) { (void)hipHostFree(b.h_counters); b.h_counters = nullptr; }
    HIP_TRY(c, hipHostMalloc(&b.h_counters, sizeof(Counters) * (size_t)c->max_groups));
    HIP_TRY(c, dev_alloc(c, &b.d_counters, sizeof(Counters) * (size_t)c->max_groups));
    HIP_TRY(c, hipMemset(b.d_counters, 0, sizeof(Counters) * (size_t)c->max_groups));
  }
  return RTUF_OK;
}

// Bins sized from what the batch asked for: a quarter above the fullest record / fragment bin, in steps of 256 / 1024 entries
// (never smaller than they are).  One allocation holds what used to be sized for the worst tile any scene could have.
// When the larger bins do not fit -- the memory limit of the context, or what the device has left (a shared GPU, several
// pipelines, 720p with many streams) -- the LAUNCH GROUP shrinks instead: the same bytes then hold deeper bins for fewer
// streams, a batch is rasterised in more, smaller launches, and nothing fails.  Only when not even one stream's bins can
// be had the context is lost.
static int grow_bins(rtuf_context* c, uint32_t needed, uint32_t fneeded)
{
  auto round_up = [](uint64_t v, uint64_t q) { return (uint32_t)std::min<uint64_t>(((v + q - 1) / q) * q, 0x7fffffffu); };
  const uint32_t cap = std::max(c->capacity, needed > c->capacity ? round_up((uint64_t)needed + needed / 4, 256) : c->capacity);
  const uint32_t fcap = std::max(c->fcapacity, fneeded > c->fcapacity ? round_up((uint64_t)fneeded + fneeded / 4, 1024) : c->fcapacity);
  if (cap == c->capacity && fcap == c->fcapacity) return RTUF_OK;
  // what may be spent: the context's limit, and no more than the device can give once the present bins are returned
  size_t free_b = 0, total_b = 0;
  HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
  const size_t held = (size_t)c->n_lanes * lane_bins_bytes(c, c->group, c->capacity, c->fcapacity);
  // (the context's limit -- rtuf_params.memory_limit_mb, or the third of the free memory taken at rtuf_finalize_models -- and
  // never more than the device can give)
  const size_t budget = std::min((size_t)((double)(free_b + held) * 0.9), c->memory_budget);
  int G = c->group;
  while (G > 1 && (size_t)c->n_lanes * lane_bins_bytes(c, G, cap, fcap) > budget) G = (G + 1) / 2;
  c->stats.over_memory_limit = (size_t)c->n_lanes * lane_bins_bytes(c, G, cap, fcap) > c->memory_budget ? 1u : 0u;      // (set: G == 1 and one stream's bins alone exceed it; cleared otherwise)
  for (;;) {
    hipError_t e = hipSuccess;
    for (int l = 0; l < c->n_lanes && e == hipSuccess; l++) {
      rtuf_context::Lane& ln = c->lane[l];
      // (a smaller group with deeper bins may fit the allocation that is there)
      const size_t want = (size_t)G * c->tiles_x * c->tiles_y * (size_t)cap * sizeof(PackedTri);
      const size_t fwant = (size_t)G * c->tiles_x * c->tiles_y * (size_t)fcap * sizeof(Frag);
      auto have = [&](void* q) { auto it = c->dev_blocks.find(q); return (q && it != c->dev_blocks.end()) ? it->second : (size_t)0; };
      if (have(ln.d_bins) < want) e = realloc_dev(c, ln.d_bins, want);
      if (e == hipSuccess && have(ln.d_fbins) < fwant) e = realloc_dev(c, ln.d_fbins, fwant);
    }
    if (e == hipSuccess) break;
    if (G == 1) {
      c->broken = true;
      return c->fail(RTUF_ERR_OOM, "growing the tile bins to %u + %u entries failed even for launch groups of one stream: %s", cap, fcap, hipGetErrorString(e));
    }
    G = (G + 1) / 2;            // the device has less than it said: halve the launch group and try again
  }
  c->stats.regrowths++;
  c->capacity = cap;
  c->fcapacity = fcap;
  if (G != c->group) {
    c->group = G;
    c->items_hint = 0;
    const int rc = alloc_counter_blocks(c);
    if (rc != RTUF_OK) { c->broken = true; return rc; }
  }
  return RTUF_OK;
}

// Setters of single-buffered state (model selection, parameters, forward-kinematics root poses / enable flags) wait for
// the batches in flight, which may still read it (or would re-read it on a bin regrowth).  The per-frame pose setters --
// joint positions, cameras, link matrices -- are staged in rings and never wait.
#define WAIT_IF_PENDING(c) do { if ((c)->pending) { const int rc_ = rtuf_sync(c); if (rc_ != RTUF_OK) return rc_; } } while (0)

// ---- pipelines: forwarding from the front context to its kids -------------------------------------------------
#define KIDS_ALL(c, expr)                                                                  \
  if ((c) && !(c)->kids.empty()) {                                                         \
    int rc_ = RTUF_OK;                                                                     \
    for (rtuf_context* k : (c)->kids) { rc_ = (expr); if (rc_ < 0) { (c)->error = k->error; return rc_; } } \
    return rc_;                                                                            \
  }
#define KIDS_ONE(c, which, expr)                                                           \
  if ((c) && !(c)->kids.empty()) {                                                         \
    rtuf_context* k = (c)->kids[which];                                                    \
    const int rc_ = (expr);                                                                \
    if (rc_ < 0) (c)->error = k->error;                                                    \
    return rc_;                                                                            \
  }
// The kid that takes the next batch is chosen without side effects; the rotation and `order` (the kids of the batches in
// flight, oldest first) change only after the call, from what the kid really did: a successful call adds one entry, and
// whatever the kid retired on the way (its oldest when it was full, everything after a failed regrowth) is trimmed from
// the front -- so a call that fails before or after retiring leaves `order` describing exactly the batches in flight.
static void settle_order(rtuf_context* c, int j, bool accepted)
{
  if (accepted) {
    c->order.push_back(j);
    c->last_kid = j;
    c->next_kid = (j + 1) % (int)c->kids.size();
  }
  int have = (int)std::count(c->order.begin(), c->order.end(), j);
  for (const int want = c->kids[j]->pending; have > want; have--)
    c->order.erase(std::find(c->order.begin(), c->order.end(), j));
}
#define KIDS_NEXT(c, expr)                                                                 \
  if ((c) && !(c)->kids.empty()) {                                                         \
    const int j_ = (c)->next_kid;                                                          \
    rtuf_context* k = (c)->kids[j_];                                                       \
    const int rc_ = (expr);                                                                \
    settle_order((c), j_, rc_ >= 0);                                                       \
    if (rc_ < 0) (c)->error = k->error;                                                    \
    return rc_;                                                                            \
  }

// Do kernels of streams a and b run side by side?  The HIP runtime maps streams onto a handful of hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default) in the order they are created, across everything in the process; two streams on one queue
// execute strictly one after the other.  Whether a new stream shares the queue of another cannot be asked, only measured: an
// idle kernel of 300 us on each, timed against one alone.  (Found the hard way: with an RCCL communicator in the process the
// two raster lanes of a context landed on one queue -- 381 k instead of 484 k frames/s, per-launch times those of kernels
// running alone.)
// RTUF_QUEUE_PROBE=0 in the environment skips the measurement (every stream is taken as the runtime hands it out): for hosts
// that share the GPU with other processes, where idle kernels cannot be timed, or that cannot spare the ~10 ms at start-up.
static bool queue_probe_enabled()
{
  static const bool on = [] { const char* e = getenv("RTUF_QUEUE_PROBE"); return !(e && e[0] == '0' && e[1] == 0); }();
  return on;
}
static bool streams_run_side_by_side(hipStream_t a, hipStream_t b)
{
  if (!queue_probe_enabled()) return true;
  using clk = std::chrono::steady_clock;
  const unsigned long long ticks = 30000;      // 300 us at 100 MHz
  launch_spin(100, a); launch_spin(100, b);    // (code object load, queue creation)
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) { (void)hipGetLastError(); return true; }
  auto timed = [&](bool both) {
    const auto t0 = clk::now();
    launch_spin(ticks, a);
    if (both) launch_spin(ticks, b);
    (void)hipStreamSynchronize(a);
    if (both) (void)hipStreamSynchronize(b);
    return std::chrono::duration<double>(clk::now() - t0).count();
  };
  // (the smallest of three: whatever else the GPU or the host is doing can only make a repetition longer)
  const double one = std::min(timed(false), std::min(timed(false), timed(false))), two = std::min(timed(true), std::min(timed(true), timed(true)));
  return two < 1.5 * one;
}

// A new non-blocking stream that runs beside every stream of `refs` (none: any stream).  Streams that turn out to share a
// hardware queue with one of them are set aside -- still alive, so that the runtime places the next one elsewhere -- and
// released afterwards; after eight of them the last one is taken as it is (*beside = false: it will work, behind whatever
// is queued in front of it).
static hipError_t create_stream_beside(const std::vector<hipStream_t>& refs, hipStream_t* out, bool* beside)
{
  *beside = true;
  hipError_t e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  if (e != hipSuccess || refs.empty()) return e;
  std::vector<hipStream_t> rejected;
  auto beside_all = [&](hipStream_t s) { for (hipStream_t r : refs) if (r && !streams_run_side_by_side(r, s)) return false; return true; };
  while (e == hipSuccess && !beside_all(*out)) {
    if (rejected.size() >= 8) { *beside = false; break; }
    rejected.push_back(*out);
    e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  return e;
}
static hipError_t create_stream_beside(hipStream_t ref, hipStream_t* out, bool* beside)
{
  return create_stream_beside(ref ? std::vector<hipStream_t>{ref} : std::vector<hipStream_t>{}, out, beside);
}
static std::vector<hipStream_t> lane_streams(const rtuf_context* c)
{
  std::vector<hipStream_t> v;
  for (int l = 0; l < c->n_lanes; l++) v.push_back(c->lane[l].stream);
  return v;
}

extern "C" {

int rtuf_abi_version(void) { return RTUF_ABI_VERSION; }

void rtuf_default_params(rtuf_params* p)
{
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->near_plane = 0.1f;                  // src/urdf_filter.cpp:54
  p->far_plane = 8.0f;                   // src/urdf_filter.cpp:53
  p->depth_distance_threshold = 0.05f;   // launch/filter_parameters.yaml:14
  p->filter_replace_value = 0.0f;        // src/urdf_filter.cpp:106 default
  p->flags = RTUF_FLAG_DEFAULT;
}

const char* rtuf_last_error(const rtuf_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error; }

int rtuf_create(rtuf_context** out, int device_id, int width, int height, int max_streams,
                const rtuf_params* params)
{
  if (!out) return RTUF_ERR_INVALID;
  *out = nullptr;
  if (width <= 0 || height <= 0 || width > 2048 || height > 2048 || max_streams <= 0) {
    snprintf(g_create_error, sizeof g_create_error, "invalid size %dx%d or stream count %d", width, height, max_streams);
    return RTUF_ERR_INVALID;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    snprintf(g_create_error, sizeof g_create_error,
             "no usable HIP device (count=%d, requested %d): %s -- this library has no CPU fallback",
             ndev, device_id, e == hipSuccess ? "ok" : hipGetErrorString(e));
    return RTUF_ERR_NO_DEVICE;
  }
  if (params && !flags_valid(params->flags)) {
    snprintf(g_create_error, sizeof g_create_error, "unknown rtuf_params.flags bits 0x%x", params->flags & ~kKnownFlags);
    return RTUF_ERR_INVALID;
  }
  if (params && params->raster_lanes > (uint32_t)kMaxLanes) {
    snprintf(g_create_error, sizeof g_create_error, "raster_lanes = %u: at most %d", params->raster_lanes, kMaxLanes);
    return RTUF_ERR_INVALID;
  }
  {
    // the kernels exist for gfx950 (MI350X / MI355X) only: on any other device the first launch would fail with an opaque
    // "invalid device function"
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      snprintf(g_create_error, sizeof g_create_error,
               "device %d is %s, not gfx950 (MI350X / MI355X): this library carries gfx950 code only and has no CPU fallback",
               device_id, e == hipSuccess ? prop.gcnArchName : hipGetErrorString(e));
      return RTUF_ERR_NO_DEVICE;
    }
  }
  rtuf_context* c = new (std::nothrow) rtuf_context();
  if (!c) return RTUF_ERR_OOM;
  c->device = device_id;
  c->width = width; c->height = height; c->max_streams = max_streams;
  c->tiles_x = (width + kTileW - 1) / kTileW;
  c->tiles_y = (height + kTileH - 1) / kTileH;
  if (params) c->params = *params; else rtuf_default_params(&c->params);
  e = hipSetDevice(device_id);
  // (a front context of several pipelines owns no streams of its own: HIP multiplexes streams onto a few hardware queues,
  // and two pipelines whose main streams share a queue do not overlap at all)
  const bool front = c->params.pipelines > 1;
  c->n_lanes = c->params.raster_lanes ? (int)c->params.raster_lanes : kDefaultLanes;
  for (int l = 0; l < c->n_lanes && e == hipSuccess && !front; l++) {
    // every further lane must be able to run beside the ones before it
    bool beside = true;
    std::vector<hipStream_t> before;
    for (int k = 0; k < l; k++) before.push_back(c->lane[k].stream);
    e = create_stream_beside(before, &c->lane[l].stream, &beside);
    if (!beside) c->lanes_share_queue = true;
  }
  // The pose stage's stream must not share a hardware queue with a lane either: its few microseconds of kernels (forward
  // kinematics, matrix stacks, the first cull) would queue behind a quarter of a millisecond of set-up or tile kernel, and
  // every lane waits for them.  (Seen with an RCCL communicator in the process: 463 k instead of 504 k frames/s.)
  if (e == hipSuccess && !front) {
    bool beside = true;
    e = create_stream_beside(lane_streams(c), &c->side, &beside);
    if (!beside) c->lanes_share_queue = true;
  }
  if (e != hipSuccess) {
    snprintf(g_create_error, sizeof g_create_error, "hip init failed: %s", hipGetErrorString(e));
    delete c;
    return RTUF_ERR_HIP;
  }
  if (c->params.pipelines > 1) {
    if (c->params.pipelines > 4) { snprintf(g_create_error, sizeof g_create_error, "pipelines = %u: at most 4", c->params.pipelines); rtuf_destroy(c); return RTUF_ERR_INVALID; }
    // (the pipelines are the overlap here: every child is a context of one raster lane, whose small batches can replay graphs)
    rtuf_params kp = c->params;
    kp.pipelines = 0;
    kp.raster_lanes = 1;
    for (uint32_t i = 0; i < c->params.pipelines; i++) {
      rtuf_context* k = nullptr;
      const int rc = rtuf_create(&k, device_id, width, height, max_streams, &kp);
      if (rc != RTUF_OK) { rtuf_destroy(c); return rc; }
      k->graphs_ok = true;
      if (!c->kids.empty()) {
        // the pipelines exist to overlap: a child whose raster stream shares the first child's hardware queue gets another
        hipStream_t fresh = nullptr;
        bool beside = true;
        if (!streams_run_side_by_side(c->kids[0]->lane[0].stream, k->lane[0].stream) &&
            create_stream_beside(c->kids[0]->lane[0].stream, &fresh, &beside) == hipSuccess && fresh) {
          (void)hipStreamDestroy(k->lane[0].stream);
          k->lane[0].stream = fresh;
          if (!beside) c->lanes_share_queue = true;
        }
      }
      c->kids.push_back(k);
    }
  }
  *out = c;
  return RTUF_OK;
}

static void free_frame_buffers(rtuf_context* c)
{
  hipSetDevice(c->device);
  auto hfree = [](auto*& p) { if (p) { hipHostFree(p); p = nullptr; } };
  dev_free(c, c->d_model_mask);
  for (auto& b : c->batch) { dev_free(c, b.d_cams); dev_free(c, b.d_link_tf); dev_free(c, b.d_mvp); dev_free(c, b.d_bg); dev_free(c, b.d_counters); dev_free(c, b.d_status); }
  for (auto& ln : c->lane) {
    dev_free(c, ln.d_bins); dev_free(c, ln.d_bin_hdr); dev_free(c, ln.d_fbins); dev_free(c, ln.d_fbin_count); dev_free(c, ln.d_clip_list);
    dev_free(c, ln.d_clip_spill); dev_free(c, ln.d_big_list); dev_free(c, ln.d_zsurface);
    for (auto*& it : ln.d_items) dev_free(c, it);
  }
  for (auto& b : c->batch) { dev_free(c, b.st_depth); dev_free(c, b.st_masked); dev_free(c, b.st_mask); b.st_streams = 0; dev_free(c, b.st_bits); b.st_bits_streams = 0; }
  for (auto*& p : c->ring_cams) hfree(p);
  for (auto*& p : c->ring_link_tf) hfree(p);
  c->h_cams = nullptr; c->h_link_tf = nullptr;
  hfree(c->h_model_mask);
  for (auto& b : c->batch) hfree(b.h_counters);
}

void rtuf_destroy(rtuf_context* c)
{
  if (!c) return;
  for (rtuf_context* k : c->kids) rtuf_destroy(k);
  c->kids.clear();
  hipSetDevice(c->device);
  if (c->side) hipStreamSynchronize(c->side);
  sync_lanes(c);
  if (c->h2d) hipStreamSynchronize(c->h2d);
  if (c->d2h) hipStreamSynchronize(c->d2h);
  for (HostModel& m : c->models) {
    Kinematics& k = m.kin;
    dev_free(c, k.d_depth); dev_free(c, k.d_parent); dev_free(c, k.d_type); dev_free(c, k.d_origin); dev_free(c, k.d_axis); dev_free(c, k.d_link_frame); dev_free(c, k.d_link_offset);
    dev_free(c, k.d_root); dev_free(c, k.d_enabled);
    for (double* q : k.h_q) if (q) hipHostFree(q);
    if (k.h_root) hipHostFree(k.h_root);
    if (k.h_enabled) hipHostFree(k.h_enabled);
  }
  free_frame_buffers(c);
  dev_free(c, c->d_cverts); dev_free(c, c->d_ctris); dev_free(c, c->d_corder); dev_free(c, c->d_chunks); dev_free(c, c->d_draws);
  for (auto& b : c->batch) {
    for (hipEvent_t ev : b.events) hipEventDestroy(ev);
    for (hipEvent_t ev : b.done) if (ev) hipEventDestroy(ev);
    if (b.posed) hipEventDestroy(b.posed);
    for (auto& g : b.graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    if (b.uploaded) hipEventDestroy(b.uploaded);
    if (b.downloaded) hipEventDestroy(b.downloaded);
  }
  if (c->fork_ev) hipEventDestroy(c->fork_ev);
  for (void* p : c->pinned) hipHostFree(p);
  if (c->h2d) hipStreamDestroy(c->h2d);
  if (c->d2h) hipStreamDestroy(c->d2h);
  if (c->side) hipStreamDestroy(c->side);
  for (auto& ln : c->lane) if (ln.stream) hipStreamDestroy(ln.stream);
  delete c;
}

int rtuf_set_params(rtuf_context* c, const rtuf_params* p)
{
  if (!c || !p) return RTUF_ERR_INVALID;
  if (!flags_valid(p->flags)) return c->fail(RTUF_ERR_INVALID, "unknown rtuf_params.flags bits 0x%x", p->flags & ~kKnownFlags);
  if (!c->kids.empty()) {
    rtuf_params kp = *p;
    kp.pipelines = 0;
    for (rtuf_context* k : c->kids) { const int rc = rtuf_set_params(k, &kp); if (rc < 0) { c->error = k->error; return rc; } }
    const uint32_t keep = c->params.pipelines;
    c->params = c->kids[0]->params;
    c->params.pipelines = keep;
    return RTUF_OK;
  }
  WAIT_IF_PENDING(c);
  // the background quad's geometry (0.99 * far, src/urdf_filter.cpp:591-596) is built by rtuf_finalize_models
  if (c->finalized && p->far_plane != c->params.far_plane)
    return c->fail(RTUF_ERR_STATE, "far_plane is fixed once the models are finalized (was %g)", (double)c->params.far_plane);
  const uint32_t keep_cap = c->params.bin_capacity, keep_inf = c->params.max_inflight_streams, keep_pipes = c->params.pipelines;
  const uint32_t keep_lanes = c->params.raster_lanes, keep_limit = c->params.memory_limit_mb;
  c->params = *p;
  c->params.bin_capacity = keep_cap;
  c->params.max_inflight_streams = keep_inf;
  c->params.pipelines = keep_pipes;
  c->params.raster_lanes = keep_lanes;
  c->params.memory_limit_mb = keep_limit;
  return RTUF_OK;      // (two-kernel mode's z-surfaces are allocated by the first batch that needs them)
}

// ---- geometry -------------------------------------------------------------------------
int rtuf_add_model(rtuf_context* c)
{
  KIDS_ALL(c, rtuf_add_model(k));
  if (!c) return RTUF_ERR_INVALID;
  if (c->finalized) return c->fail(RTUF_ERR_STATE, "models already finalised");
  if (c->models.size() >= 64) return c->fail(RTUF_ERR_INVALID, "at most 64 models per context");
  c->models.emplace_back();
  return (int)c->models.size() - 1;
}

int rtuf_add_link(rtuf_context* c, int model)
{
  KIDS_ALL(c, rtuf_add_link(k, model));
  if (!c) return RTUF_ERR_INVALID;
  if (c->finalized) return c->fail(RTUF_ERR_STATE, "models already finalised");
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  c->models[model].links.emplace_back();
  return (int)c->models[model].links.size() - 1;
}

int rtuf_add_draw(rtuf_context* c, int model, int link, int pre_op, const float op_xyz[3],
                  const float* vertices_xyz, int n_vertices, const uint32_t* triangles, int n_triangles)
{
  KIDS_ALL(c, rtuf_add_draw(k, model, link, pre_op, op_xyz, vertices_xyz, n_vertices, triangles, n_triangles));
  if (!c) return RTUF_ERR_INVALID;
  if (c->finalized) return c->fail(RTUF_ERR_STATE, "models already finalised");
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  HostModel& m = c->models[model];
  if (link < 0 || link >= (int)m.links.size()) return c->fail(RTUF_ERR_INVALID, "bad link id %d", link);
  if (pre_op < RTUF_OP_NONE || pre_op > RTUF_OP_TRANSLATE) return c->fail(RTUF_ERR_INVALID, "bad pre_op %d", pre_op);
  if (n_vertices < 0 || n_triangles < 0 || (n_vertices && !vertices_xyz) || (n_triangles && !triangles))
    return c->fail(RTUF_ERR_INVALID, "bad geometry arrays");
  for (int i = 0; i < 3 * n_triangles; i++)
    if (triangles[i] >= (uint32_t)n_vertices) return c->fail(RTUF_ERR_INVALID, "triangle index %u out of range", triangles[i]);
  HostDraw d;
  d.pre_op = pre_op;
  for (int k = 0; k < 3; k++) d.op[k] = (pre_op != RTUF_OP_NONE && op_xyz) ? op_xyz[k] : 0.0f;
  d.verts.assign(vertices_xyz, vertices_xyz + 3 * (size_t)n_vertices);
  d.tris.assign(triangles, triangles + 3 * (size_t)n_triangles);
  m.links[link].draws.push_back(std::move(d));
  return (int)m.links[link].draws.size() - 1;
}

int rtuf_num_links(const rtuf_context* c, int model)
{
  if (c && !c->kids.empty()) return rtuf_num_links(c->kids[0], model);
  if (!c || model < 0 || model >= (int)c->models.size()) return RTUF_ERR_INVALID;
  return (int)c->models[model].links.size();
}

int64_t rtuf_num_triangles(const rtuf_context* c) { return c ? (c->kids.empty() ? c->n_tris : c->kids[0]->n_tris) : 0; }
int64_t rtuf_num_vertices(const rtuf_context* c) { return c ? (c->kids.empty() ? c->n_cverts : c->kids[0]->n_cverts) : 0; }

static int alloc_frame_buffers(rtuf_context* c)
{
  const int N = c->max_streams;
  const size_t tiles = (size_t)c->tiles_x * c->tiles_y;
  const size_t L = (size_t)std::max(c->n_links, 1);
  for (int r = 0; r <= kMaxInflight; r++) {
    HIP_TRY(c, hipHostMalloc(&c->ring_cams[r], sizeof(Camera) * N));
    HIP_TRY(c, hipHostMalloc(&c->ring_link_tf[r], sizeof(double) * 16 * L * N));
  }
  c->cam_ring = rtuf_context::StageRing(); c->link_ring = rtuf_context::StageRing();
  c->h_cams = c->ring_cams[0]; c->h_link_tf = c->ring_link_tf[0];
  HIP_TRY(c, hipHostMalloc(&c->h_model_mask, sizeof(uint64_t) * N));
  for (auto& b : c->batch)
    for (int l = 0; l < c->n_lanes; l++)
      if (!b.done[l]) HIP_TRY(c, hipEventCreateWithFlags(&b.done[l], hipEventDisableTiming));
  HIP_TRY(c, dev_alloc(c, &c->d_model_mask, sizeof(uint64_t) * N));
  for (auto& b : c->batch) {
    HIP_TRY(c, dev_alloc(c, &b.d_cams, sizeof(Camera) * N));
    HIP_TRY(c, dev_alloc(c, &b.d_link_tf, sizeof(double) * 16 * L * N));
    HIP_TRY(c, dev_alloc(c, &b.d_mvp, sizeof(float) * 16 * (size_t)(c->n_draws + 1) * N));
    HIP_TRY(c, dev_alloc(c, &b.d_bg, sizeof(BgInfo) * N));
    HIP_TRY(c, dev_alloc(c, &b.d_status, 64));
    HIP_TRY(c, hipMemset(b.d_status, 0, 64));
    b.dirty_cams = b.dirty_link_tf = true; b.uploaded_streams = 0;
    if (!b.posed) HIP_TRY(c, hipEventCreateWithFlags(&b.posed, hipEventDisableTiming));
  }
  c->dirty_mask = true; c->mask_uploaded_streams = 0;
  // identity defaults
  static const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int s = 0; s < N; s++) {
    for (int r = 0; r <= kMaxInflight; r++) {
      Camera& cam = c->ring_cams[r][s];
      memcpy(cam.projection, I, sizeof I);
      memcpy(cam.offset_inv, I, sizeof I);
      memcpy(cam.cam_tf, I, sizeof I);
      cam.shift[0] = cam.shift[1] = 0.0;
      for (size_t l = 0; l < L; l++) memcpy(c->ring_link_tf[r] + ((size_t)s * L + l) * 16, I, sizeof I);
    }
    c->h_model_mask[s] = ~0ull;
  }
  // rasteriser working set
  // Launch group = the streams one lane's bins are sized for.  One lane: the whole batch up to 1024 streams (kernels of 4x
  // the work lose 4x less to their ramp and tail: 1024 streams in one group 522 k frames/s, in four groups of 256 one
  // after the other 460 k).  Several lanes: the streams divided by the lanes -- a full batch is one group per lane, the
  // lanes' kernels fill each other's ramps, tails and launch gaps, and the bins are what one group for all streams would
  // take.  (Measured on the 256-stream VGA workload: three lanes x 86 streams 514-525 k frames/s in 8.9 GB, two lanes x 128
  // 497-508 k in 8.8 GB, two x four groups of 64 490 k in 4.45 GB, one lane 469 k in 8.7 GB; 64 x 720p with two walls:
  // three lanes 180 k, two 168 k, one 143 k.  rtuf_params.memory_limit_mb / max_inflight_streams pick smaller working sets.)
  int G = c->params.max_inflight_streams ? (int)c->params.max_inflight_streams
                                         : (c->n_lanes > 1 && N >= kSplitMin ? std::min((N + c->n_lanes - 1) / c->n_lanes, 1024) : 1024);
  G = std::min(G, N);
  // Bins: fixed capacity per (stream, tile), direct addressing (one atomicAdd gives the slot: anything cleverer -- paged
  // bins were built and measured, DESIGN.md appendix A.2 -- costs the two big kernels 8 to 24 %).  What is NOT fixed any more is
  // the size: 1024 records + 4096 fragments per bin to start with (64 KiB per bin), grown on the first
  // batch to a quarter above the fullest bin that batch produced (the 250 k-triangle robot: 3840 + 13568; round 2
  // reserved 8192 + 32768 whatever the scene).  rtuf_params.bin_capacity fixes the starting point.
  uint32_t cap = c->params.bin_capacity ? c->params.bin_capacity : 1024u;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
  c->memory_budget = std::max(free_b / 3, (size_t)1 << 30);
  if (c->params.memory_limit_mb) c->memory_budget = (size_t)c->params.memory_limit_mb << 20;
  c->capacity = cap;
  c->fcapacity = std::max<uint32_t>(4 * cap, 1024);   // 8-byte fragments of all boxes up to 4x4 pixel centres
  while (G > 1 && (size_t)c->n_lanes * lane_bins_bytes(c, G, c->capacity, c->fcapacity) > c->memory_budget) G = (G + 1) / 2;
  c->group = G;
  c->stats.over_memory_limit = (size_t)c->n_lanes * lane_bins_bytes(c, G, c->capacity, c->fcapacity) > c->memory_budget ? 1u : 0u;      // (the first allocation can exceed it already)
  c->clip_capacity = (uint32_t)std::min<size_t>(std::max<size_t>((size_t)G * 8192 / kCounterShards, 1024), (size_t)1 << 22);   // per shard
  c->big_capacity = (uint32_t)std::min<size_t>(std::max<size_t>((size_t)G * 64, 1024), (size_t)1 << 20);                 // per shard
  for (int l = 0; l < c->n_lanes; l++) {
    rtuf_context::Lane& ln = c->lane[l];
    HIP_TRY(c, dev_alloc(c, &ln.d_bins, (size_t)G * tiles * cap * sizeof(PackedTri)));
    HIP_TRY(c, dev_alloc(c, &ln.d_bin_hdr, (size_t)G * tiles * sizeof(BinHeader)));
    HIP_TRY(c, dev_alloc(c, &ln.d_fbins, (size_t)G * tiles * c->fcapacity * sizeof(Frag)));
    HIP_TRY(c, dev_alloc(c, &ln.d_fbin_count, (size_t)G * tiles * sizeof(uint32_t)));
    HIP_TRY(c, hipMemsetAsync(ln.d_fbin_count, 0, (size_t)G * tiles * sizeof(uint32_t), ln.stream));
    launch_init_headers(ln.d_bin_hdr, (size_t)G * tiles, ln.stream);
    HIP_TRY(c, hipStreamSynchronize(ln.stream));
    HIP_TRY(c, dev_alloc(c, &ln.d_clip_list, (size_t)c->clip_capacity * kCounterShards * sizeof(ClipItem)));
    HIP_TRY(c, dev_alloc(c, &ln.d_clip_spill, clip_spill_bytes(c->clip_capacity)));
    HIP_TRY(c, dev_alloc(c, &ln.d_big_list, (size_t)c->big_capacity * kCounterShards * sizeof(BigRec)));
    for (auto*& it : ln.d_items) HIP_TRY(c, dev_alloc(c, &it, (size_t)c->n_chunks * (size_t)max_items_per_chunk(G) * sizeof(WorkItem)));
  }
  return alloc_counter_blocks(c);
}

int rtuf_finalize_models(rtuf_context* c)
{
  KIDS_ALL(c, rtuf_finalize_models(k));
  if (!c) return RTUF_ERR_INVALID;
  if (c->finalized) return c->fail(RTUF_ERR_STATE, "models already finalised");
  hipSetDevice(c->device);
  std::vector<float4> cverts;
  std::vector<uint32_t> ctris, corder;
  std::vector<Chunk> chunks;
  std::vector<Draw> draws;
  int64_t tri_seq = 0;
  // Splits one draw's triangle list into chunks of <= kBlock triangles / <= kMaxChunkVerts vertices.
  // Triangles are taken in Morton order of their centroids, so a chunk is a compact patch of the surface:
  // fewer distinct vertices per chunk (each is transformed once per chunk and stream) and a tighter
  // bounding sphere (more whole-chunk frustum culls) than the file order of a mesh gives.  The depth
  // test's tie-break is the GL draw order, so every triangle keeps its original sequence number (corder).
  auto add_chunks = [&](const std::vector<float>& v, const std::vector<uint32_t>& t, uint32_t draw_id, uint32_t model, bool background) {
    const uint32_t nt = (uint32_t)(t.size() / 3);
    // Vertices with bit-identical coordinates are one vertex here: STL (and Assimp's importers in general) bring three
    // vertices of their own per facet, and the vertex shader gives identical inputs identical results, so the image
    // cannot change -- but a chunk then transforms ~0.5 instead of 3 vertices per triangle and holds 256 triangles.
    std::vector<uint32_t> canon(v.size() / 3);
    {
      struct Key { uint32_t x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
      struct KeyHash { size_t operator()(const Key& k) const { uint64_t h = k.x * 0x9E3779B97F4A7C15ull; h ^= (h >> 29) + k.y * 0xC2B2AE3D27D4EB4Full; h ^= (h >> 31) + k.z * 0x165667B19E3779F9ull; return (size_t)(h ^ (h >> 32)); } };
      std::unordered_map<Key, uint32_t, KeyHash> first;
      first.reserve(canon.size());
      for (uint32_t i = 0; i < (uint32_t)canon.size(); i++) {
        Key k;
        memcpy(&k, &v[3 * (size_t)i], sizeof k);
        canon[i] = first.emplace(k, i).first->second;
      }
    }
    std::vector<int32_t> local(v.size() / 3, -1);
    std::vector<uint32_t> touched;
    std::vector<uint32_t> perm(nt);
    for (uint32_t i = 0; i < nt; i++) perm[i] = i;
    if (nt > (uint32_t)kBlock && !RTUF_ABL(c->params.flags, 0x8000u)) {        // (0x8000 in RTUF_ABLATE builds: timing experiment, file order)
      float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
      for (size_t i = 0; i < v.size(); i++) { lo[i % 3] = std::min(lo[i % 3], v[i]); hi[i % 3] = std::max(hi[i % 3], v[i]); }
      auto spread = [](uint32_t x) {          // 10 bits -> every third bit
        x &= 1023u; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu;
        x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
        return x;
      };
      std::vector<uint32_t> code(nt);
      for (uint32_t i = 0; i < nt; i++) {
        uint32_t m = 0;
        for (int k = 0; k < 3; k++) {
          const float cen = (v[3 * (size_t)t[3 * (size_t)i] + k] + v[3 * (size_t)t[3 * (size_t)i + 1] + k] + v[3 * (size_t)t[3 * (size_t)i + 2] + k]) * (1.0f / 3.0f);
          const float ext = hi[k] - lo[k];
          const float u = ext > 0 ? (cen - lo[k]) / ext : 0.0f;
          m |= spread((uint32_t)std::min(1023.0f, std::max(0.0f, u * 1023.0f))) << k;
        }
        code[i] = m;
      }
      std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return code[x] < code[y]; });
    }
    uint32_t done = 0;
    while (done < nt) {
      Chunk ch{};
      ch.tri_begin = (uint32_t)ctris.size();
      ch.vert_begin = (uint32_t)cverts.size();
      ch.draw = draw_id; ch.model = model;
      touched.clear();
      uint32_t nv = 0, n = 0;
      while (done + n < nt && n < (uint32_t)kBlock) {
        const uint32_t ix[3] = {canon[t[3 * (size_t)perm[done + n]]], canon[t[3 * (size_t)perm[done + n] + 1]], canon[t[3 * (size_t)perm[done + n] + 2]]};
        uint32_t fresh = 0;
        for (int k = 0; k < 3; k++) {
          bool seen = local[ix[k]] >= 0;
          for (int q = 0; q < k && !seen; q++) seen = ix[q] == ix[k];
          if (!seen) fresh++;
        }
        if (nv + fresh > (uint32_t)kMaxChunkVerts) break;
        uint32_t li[3];
        for (int k = 0; k < 3; k++) {
          if (local[ix[k]] < 0) {
            local[ix[k]] = (int32_t)nv++;
            touched.push_back(ix[k]);
            cverts.push_back(make_float4(v[3 * (size_t)ix[k]], v[3 * (size_t)ix[k] + 1], v[3 * (size_t)ix[k] + 2], 1.0f));
          }
          li[k] = (uint32_t)local[ix[k]];
        }
        ctris.push_back(li[0] | (li[1] << 10) | (li[2] << 20));
        corder.push_back(background ? 0u : (uint32_t)(tri_seq + perm[done + n] + 1));
        n++;
      }
      ch.tri_count = n; ch.vert_count = nv;
      {
        // bounding box (centre + half extents, padded): used for whole-chunk frustum culling
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t q = 0; q < nv; q++) {
          const float4& p = cverts[ch.vert_begin + q];
          const float pp[3] = {p.x, p.y, p.z};
          for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], pp[k]); hi[k] = std::max(hi[k], pp[k]); }
        }
        for (int k = 0; k < 3; k++) {
          ch.center[k] = 0.5f * (lo[k] + hi[k]);
          // half extent about the rounded centre, rounded up (the cull test must stay conservative)
          ch.half[k] = std::nextafter(std::max(hi[k] - ch.center[k], ch.center[k] - lo[k]), INFINITY) * 1.0001f + 1e-7f;
        }
      }
      chunks.push_back(ch);
      for (uint32_t id : touched) local[id] = -1;
      done += n;
    }
    tri_seq += nt;
  };
  int link_base = 0;
  for (size_t mi = 0; mi < c->models.size(); mi++) {
    HostModel& m = c->models[mi];
    m.link_base = link_base;
    for (size_t li = 0; li < m.links.size(); li++) {
      for (const HostDraw& hd : m.links[li].draws) {
        Draw d{};
        d.link = (uint32_t)(link_base + li);
        d.pre_op = (uint32_t)hd.pre_op;
        d.op[0] = hd.op[0]; d.op[1] = hd.op[1]; d.op[2] = hd.op[2];
        d.model = (uint32_t)mi;
        const uint32_t draw_id = (uint32_t)draws.size();
        draws.push_back(d);
        add_chunks(hd.verts, hd.tris, draw_id, (uint32_t)mi, false);
      }
    }
    link_base += (int)m.links.size();
  }
  c->n_links = link_base;
  c->n_draws = (int)draws.size();
  c->n_tris = tri_seq;
  if (tri_seq >= (int64_t)kMaxOrder) return c->fail(RTUF_ERR_CAPACITY, "%lld triangles: draw-order keys are limited to %u", (long long)tri_seq, kMaxOrder);
  {
    // draw orders are 1 .. tri_seq (0 = background quad): the key's low word keeps the bits above them for the float z's
    // low bits (at most 16: by then the exact-z pass is needed only within nanometres of the near plane)
    int order_bits = 1;
    while (((int64_t)1 << order_bits) <= tri_seq) order_bits++;
    c->key_shift = std::min(32 - order_bits, 16);
  }
  // background quad as the hidden last draw (used only when a stream's projection does not make
  // it a constant full-screen plane): GL_QUADS -> (0,1,3), (1,2,3); both get order 0
  {
    const float zq = (float)((double)c->params.far_plane * 0.99);
    const std::vector<float> bv = {-100.0f, -100.0f, zq, 100.0f, -100.0f, zq, 100.0f, 100.0f, zq, -100.0f, 100.0f, zq};
    const std::vector<uint32_t> bt = {0, 1, 3, 1, 2, 3};
    c->bg_chunk = (uint32_t)chunks.size();
    add_chunks(bv, bt, (uint32_t)c->n_draws, 0, true);
  }
  c->n_chunks = (int)chunks.size();
  c->n_cverts = (int64_t)cverts.size();
  if (draws.empty()) draws.push_back(Draw{});
  HIP_TRY(c, dev_alloc(c, &c->d_cverts, cverts.size() * sizeof(float4)));
  HIP_TRY(c, dev_alloc(c, &c->d_ctris, ctris.size() * sizeof(uint32_t)));
  HIP_TRY(c, dev_alloc(c, &c->d_corder, corder.size() * sizeof(uint32_t)));
  HIP_TRY(c, dev_alloc(c, &c->d_chunks, chunks.size() * sizeof(Chunk)));
  HIP_TRY(c, dev_alloc(c, &c->d_draws, draws.size() * sizeof(Draw)));
  HIP_TRY(c, hipMemcpy(c->d_cverts, cverts.data(), cverts.size() * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_ctris, ctris.data(), ctris.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_corder, corder.data(), corder.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_draws, draws.data(), draws.size() * sizeof(Draw), hipMemcpyHostToDevice));
  const int rc = alloc_frame_buffers(c);
  if (rc != RTUF_OK) return rc;
  // host copies of the geometry are no longer needed
  for (HostModel& m : c->models)
    for (HostLink& l : m.links)
      for (HostDraw& d : l.draws) { d.verts.clear(); d.verts.shrink_to_fit(); d.tris.clear(); d.tris.shrink_to_fit(); }
  c->finalized = true;
  return RTUF_OK;
}

}  // extern "C"

// The setters of cameras / link matrices write into the staging set no batch in flight reads.  The first write after a
// batch took the previous set carries over what this call does not overwrite (`whole` = it overwrites everything).
static void begin_camera_write(rtuf_context* c, bool whole)
{
  auto& r = c->cam_ring;
  if (!r.carried) {
    if (!whole) memcpy(c->ring_cams[r.write], c->ring_cams[r.live], sizeof(Camera) * (size_t)c->max_streams);
    r.carried = true;
  }
  r.written = true;
}
static void begin_link_write(rtuf_context* c, bool whole)
{
  auto& r = c->link_ring;
  if (!r.carried) {
    if (!whole) memcpy(c->ring_link_tf[r.write], c->ring_link_tf[r.live], sizeof(double) * 16 * (size_t)std::max(c->n_links, 1) * (size_t)c->max_streams);
    r.carried = true;
  }
  r.written = true;
}
// A new batch takes the staged set of one array (if anything was staged since the last batch) and moves the setters on
// to a set no batch in flight reads (kMaxInflight + 1 sets: one is always free).  Returns the set the batch reads.
template <typename IdxOf>
static int take_staged(rtuf_context* c, rtuf_context::StageRing& r, const rtuf_context::Batch& b, IdxOf idx_of)
{
  if (r.written) {
    r.live = r.write;
    bool used[kMaxInflight + 1] = {};
    used[r.live] = true;
    for (const auto& o : c->batch) if (o.active && &o != &b) used[idx_of(o)] = true;
    for (int i = 0; i <= kMaxInflight; i++) if (!used[i]) { r.write = i; break; }
    r.written = false;
    r.carried = false;
  } else if (r.write == r.live) {
    // nothing was staged since the last batch (or ever): the batch reads the live set, so the setters must move on to a
    // set no batch reads -- a setter called while this batch is in flight would otherwise write into the pinned set its
    // upload is still reading
    bool used[kMaxInflight + 1] = {};
    used[r.live] = true;
    for (const auto& o : c->batch) if (o.active && &o != &b) used[idx_of(o)] = true;
    for (int i = 0; i <= kMaxInflight; i++) if (!used[i]) { r.write = i; break; }
    r.carried = false;
  }
  return r.live;
}

extern "C" {

int rtuf_set_stream_models(rtuf_context* c, int stream, const int* model_ids, int n_models)
{
  KIDS_ALL(c, rtuf_set_stream_models(k, stream, model_ids, n_models));
  if (!c) return RTUF_ERR_INVALID;
  WAIT_IF_PENDING(c);
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (stream < 0 || stream >= c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream %d", stream);
  uint64_t mask = 0;
  for (int i = 0; i < n_models; i++) {
    if (model_ids[i] < 0 || model_ids[i] >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model_ids[i]);
    mask |= 1ull << model_ids[i];
  }
  c->h_model_mask[stream] = mask;
  c->dirty_mask = true;
  return RTUF_OK;
}

// ---- poses ----------------------------------------------------------------------------
int rtuf_set_camera(rtuf_context* c, int stream, const double projection[16], const double camera_offset_inv[16],
                    const double camera_tf[16])
{
  KIDS_ALL(c, rtuf_set_camera(k, stream, projection, camera_offset_inv, camera_tf));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (stream < 0 || stream >= c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream %d", stream);
  begin_camera_write(c, false);
  Camera& cam = c->h_cams[stream];
  if (projection) memcpy(cam.projection, projection, sizeof cam.projection);
  if (camera_offset_inv) memcpy(cam.offset_inv, camera_offset_inv, sizeof cam.offset_inv);
  if (camera_tf) memcpy(cam.cam_tf, camera_tf, sizeof cam.cam_tf);
  for (auto& b : c->batch) b.dirty_cams = true;
  return RTUF_OK;
}

// getProjectionMatrix, src/urdf_filter.cpp:459-501
void rtuf_projection_from_intrinsics(double fx, double fy, double cx, double cy, double Tx, double Ty, int width,
                                     int height, double near_plane, double far_plane, double P[16],
                                     double* camera_tx, double* camera_ty)
{
  if (camera_tx) *camera_tx = -1 * (Tx / fx);
  if (camera_ty) *camera_ty = -1 * (Ty / fy);
  for (int i = 0; i < 16; i++) P[i] = 0.0;
  P[0] = -2.0 * fx / width;
  P[5] = 2.0 * fy / height;
  P[8] = 2.0 * (0.5 - cx / width);
  P[9] = 2.0 * (cy / height - 0.5);
  P[10] = -(far_plane + near_plane) / (far_plane - near_plane);
  P[14] = -2.0 * far_plane * near_plane / (far_plane - near_plane);
  P[11] = -1;
}

int rtuf_set_link_poses(rtuf_context* c, int stream, int model, const double* link_tf, int n_links)
{
  KIDS_ALL(c, rtuf_set_link_poses(k, stream, model, link_tf, n_links));
  if (!c || !link_tf) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (stream < 0 || stream >= c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream %d", stream);
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  const HostModel& m = c->models[model];
  if (n_links != (int)m.links.size()) return c->fail(RTUF_ERR_INVALID, "model %d has %d links, got %d", model, (int)m.links.size(), n_links);
  // (switching a stream off forward kinematics changes single-buffered state: that alone waits for the batches in flight)
  if (m.kin.h_enabled && m.kin.h_enabled[stream]) WAIT_IF_PENDING(c);
  begin_link_write(c, false);
  memcpy(c->h_link_tf + ((size_t)stream * c->n_links + m.link_base) * 16, link_tf, sizeof(double) * 16 * (size_t)n_links);
  // (a stream that leaves forward kinematics also gets its host-set camera back: the FK kernel may have overwritten cam_tf)
  if (m.kin.h_enabled && m.kin.h_enabled[stream]) { c->models[model].kin.h_enabled[stream] = 0; c->models[model].kin.dirty_aux = true; for (auto& b : c->batch) b.dirty_cams = true; }
  for (auto& b : c->batch) b.dirty_link_tf = true;
  return RTUF_OK;
}

int rtuf_set_cameras(rtuf_context* c, int first, int n, const double* projection, const double* offset_inv,
                     const double* cam_tf)
{
  KIDS_ALL(c, rtuf_set_cameras(k, first, n, projection, offset_inv, cam_tf));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (first < 0 || n < 0 || first + n > c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream range %d+%d", first, n);
  begin_camera_write(c, false);
  for (int s = 0; s < n; s++) {
    Camera& cam = c->h_cams[first + s];
    if (projection) memcpy(cam.projection, projection + 16 * (size_t)s, sizeof cam.projection);
    if (offset_inv) memcpy(cam.offset_inv, offset_inv + 16 * (size_t)s, sizeof cam.offset_inv);
    if (cam_tf) memcpy(cam.cam_tf, cam_tf + 16 * (size_t)s, sizeof cam.cam_tf);
  }
  for (auto& b : c->batch) b.dirty_cams = true;
  return RTUF_OK;
}

int rtuf_set_camera_shift(rtuf_context* c, int first, int n, const double* camera_tx, const double* camera_ty)
{
  KIDS_ALL(c, rtuf_set_camera_shift(k, first, n, camera_tx, camera_ty));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (first < 0 || n < 0 || first + n > c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream range %d+%d", first, n);
  begin_camera_write(c, false);
  for (int s = 0; s < n; s++) {
    c->h_cams[first + s].shift[0] = camera_tx ? camera_tx[s] : 0.0;
    c->h_cams[first + s].shift[1] = camera_ty ? camera_ty[s] : 0.0;
  }
  for (auto& b : c->batch) b.dirty_cams = true;
  return RTUF_OK;
}

int rtuf_set_link_poses_batch(rtuf_context* c, int first, int n, int model, const double* link_tf, int n_links)
{
  KIDS_ALL(c, rtuf_set_link_poses_batch(k, first, n, model, link_tf, n_links));
  if (!c || !link_tf) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (first < 0 || n < 0 || first + n > c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream range %d+%d", first, n);
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  const HostModel& m = c->models[model];
  if (n_links != (int)m.links.size()) return c->fail(RTUF_ERR_INVALID, "model %d has %d links, got %d", model, (int)m.links.size(), n_links);
  // (switching streams off forward kinematics changes single-buffered state: that alone waits for the batches in flight)
  if (m.kin.h_enabled) for (int s = 0; s < n; s++) if (m.kin.h_enabled[first + s]) { WAIT_IF_PENDING(c); break; }
  // one model that owns every link, all streams: this call overwrites the whole set, nothing to carry over
  begin_link_write(c, first == 0 && n == c->max_streams && n_links == c->n_links);
  for (int s = 0; s < n; s++) {
    memcpy(c->h_link_tf + ((size_t)(first + s) * c->n_links + m.link_base) * 16, link_tf + (size_t)s * n_links * 16,
           sizeof(double) * 16 * (size_t)n_links);
    if (m.kin.h_enabled && m.kin.h_enabled[first + s]) { c->models[model].kin.h_enabled[first + s] = 0; c->models[model].kin.dirty_aux = true; for (auto& b : c->batch) b.dirty_cams = true; }
  }
  for (auto& b : c->batch) b.dirty_link_tf = true;
  return RTUF_OK;
}


// column-major GL matrix -> row-major 3x3 basis + origin
static void gl_to_tf12(const double* g, double* t)
{
  t[0] = g[0]; t[1] = g[4]; t[2] = g[8];
  t[3] = g[1]; t[4] = g[5]; t[5] = g[9];
  t[6] = g[2]; t[7] = g[6]; t[8] = g[10];
  t[9] = g[12]; t[10] = g[13]; t[11] = g[14];
}

int rtuf_set_kinematics(rtuf_context* c, int model, int n_frames, const int32_t* parent, const int32_t* joint_type,
                        const double* joint_origin, const double* joint_axis, const int32_t* link_frame,
                        const double* link_offset, int n_links)
{
  KIDS_ALL(c, rtuf_set_kinematics(k, model, n_frames, parent, joint_type, joint_origin, joint_axis, link_frame, link_offset, n_links));
  if (!c || !parent || !joint_type || !joint_origin || !joint_axis || !link_frame || !link_offset) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  HostModel& m = c->models[model];
  if (n_links != (int)m.links.size()) return c->fail(RTUF_ERR_INVALID, "model %d has %d links, got %d", model, (int)m.links.size(), n_links);
  if (n_frames <= 0 || n_frames > 4096) return c->fail(RTUF_ERR_INVALID, "bad frame count %d", n_frames);
  for (int i = 0; i < n_frames; i++) {
    if (parent[i] >= i || parent[i] < -1) return c->fail(RTUF_ERR_INVALID, "frame %d: parent %d must precede it", i, parent[i]);
    if (joint_type[i] < 0 || joint_type[i] > 2) return c->fail(RTUF_ERR_INVALID, "frame %d: bad joint type %d", i, joint_type[i]);
    int depth = 0;
    for (int f = i; f >= 0; f = parent[f]) if (++depth > 64) return c->fail(RTUF_ERR_INVALID, "kinematic chain deeper than 64");
  }
  for (int l = 0; l < n_links; l++)
    if (link_frame[l] < 0 || link_frame[l] >= n_frames) return c->fail(RTUF_ERR_INVALID, "link %d: bad frame %d", l, link_frame[l]);
  hipSetDevice(c->device);
  Kinematics& k = m.kin;
  if (k.n_frames) return c->fail(RTUF_ERR_STATE, "kinematics of model %d already set", model);
  std::vector<double> org(12 * (size_t)n_frames), off(12 * (size_t)std::max(n_links, 1));
  for (int i = 0; i < n_frames; i++) gl_to_tf12(joint_origin + 16 * (size_t)i, &org[12 * (size_t)i]);
  for (int l = 0; l < n_links; l++) gl_to_tf12(link_offset + 16 * (size_t)l, &off[12 * (size_t)l]);
  const size_t N = (size_t)c->max_streams;
  std::vector<int32_t> depth(n_frames, 0);
  int max_depth = 0;
  for (int i = 0; i < n_frames; i++) { depth[i] = parent[i] < 0 ? 0 : depth[parent[i]] + 1; max_depth = std::max(max_depth, depth[i]); }
  // all or nothing: a failure part-way releases what this call allocated, so the call can simply be made again
  auto release = [&]() {
    dev_free(c, k.d_depth); dev_free(c, k.d_parent); dev_free(c, k.d_type); dev_free(c, k.d_origin); dev_free(c, k.d_axis);
    dev_free(c, k.d_link_frame); dev_free(c, k.d_link_offset); dev_free(c, k.d_root); dev_free(c, k.d_enabled);
    for (double*& q : k.h_q) if (q) { hipHostFree(q); q = nullptr; }
    if (k.h_root) { hipHostFree(k.h_root); k.h_root = nullptr; }
    if (k.h_enabled) { hipHostFree(k.h_enabled); k.h_enabled = nullptr; }
  };
  auto build = [&]() -> hipError_t {
    hipError_t e;
#define KIN_TRY(expr) do { e = (expr); if (e != hipSuccess) return e; } while (0)
    KIN_TRY(dev_alloc(c, &k.d_depth, sizeof(int32_t) * n_frames));
    KIN_TRY(dev_alloc(c, &k.d_parent, sizeof(int32_t) * n_frames));
    KIN_TRY(dev_alloc(c, &k.d_type, sizeof(int32_t) * n_frames));
    KIN_TRY(dev_alloc(c, &k.d_origin, sizeof(double) * 12 * n_frames));
    KIN_TRY(dev_alloc(c, &k.d_axis, sizeof(double) * 3 * n_frames));
    KIN_TRY(dev_alloc(c, &k.d_link_frame, sizeof(int32_t) * std::max(n_links, 1)));
    KIN_TRY(dev_alloc(c, &k.d_link_offset, sizeof(double) * off.size()));
    KIN_TRY(dev_alloc(c, &k.d_root, sizeof(double) * N * 12));
    KIN_TRY(dev_alloc(c, &k.d_enabled, N));
    for (double*& q : k.h_q) KIN_TRY(hipHostMalloc(&q, sizeof(double) * N * n_frames));
    KIN_TRY(hipHostMalloc(&k.h_root, sizeof(double) * N * 12));
    KIN_TRY(hipHostMalloc(&k.h_enabled, N));
    memset(k.h_enabled, 0, N);
    for (double* q : k.h_q) memset(q, 0, sizeof(double) * N * n_frames);
    KIN_TRY(hipMemcpy(k.d_depth, depth.data(), sizeof(int32_t) * n_frames, hipMemcpyHostToDevice));
    KIN_TRY(hipMemcpy(k.d_parent, parent, sizeof(int32_t) * n_frames, hipMemcpyHostToDevice));
    KIN_TRY(hipMemcpy(k.d_type, joint_type, sizeof(int32_t) * n_frames, hipMemcpyHostToDevice));
    KIN_TRY(hipMemcpy(k.d_origin, org.data(), sizeof(double) * org.size(), hipMemcpyHostToDevice));
    KIN_TRY(hipMemcpy(k.d_axis, joint_axis, sizeof(double) * 3 * n_frames, hipMemcpyHostToDevice));
    if (n_links) KIN_TRY(hipMemcpy(k.d_link_frame, link_frame, sizeof(int32_t) * n_links, hipMemcpyHostToDevice));
    KIN_TRY(hipMemcpy(k.d_link_offset, off.data(), sizeof(double) * off.size(), hipMemcpyHostToDevice));
#undef KIN_TRY
    return hipSuccess;
  };
  const hipError_t e = build();
  if (e != hipSuccess) {
    release();
    return c->fail(e == hipErrorOutOfMemory ? RTUF_ERR_OOM : RTUF_ERR_HIP, "rtuf_set_kinematics: %s", hipGetErrorString(e));
  }
  k.max_depth = max_depth;
  k.n_frames = n_frames;
  return RTUF_OK;
}

int rtuf_set_joint_positions(rtuf_context* c, int first, int n, int model, const double* q, const double* root_tf, int camera_frame)
{
  KIDS_ALL(c, rtuf_set_joint_positions(k, first, n, model, q, root_tf, camera_frame));
  if (!c || !q) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (model < 0 || model >= (int)c->models.size()) return c->fail(RTUF_ERR_INVALID, "bad model id %d", model);
  if (first < 0 || n < 0 || first + n > c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad stream range %d+%d", first, n);
  Kinematics& k = c->models[model].kin;
  if (!k.n_frames) return c->fail(RTUF_ERR_STATE, "call rtuf_set_kinematics for model %d first", model);
  if (camera_frame < -1 || camera_frame >= k.n_frames) return c->fail(RTUF_ERR_INVALID, "bad camera frame %d", camera_frame);
  static const double I12[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
  // root transforms, enable flags and the camera frame are staged in single buffers: changing them
  // while a batch is in flight waits for that batch first (joint positions alone never wait)
  bool aux_changes = camera_frame != k.camera_frame;
  std::vector<double> roots(12 * (size_t)n);
  for (int s = 0; s < n; s++) {
    double* r12 = &roots[12 * (size_t)s];
    if (root_tf) gl_to_tf12(root_tf + 16 * (size_t)s, r12);
    else memcpy(r12, I12, sizeof I12);
    if (memcmp(k.h_root + 12 * (size_t)(first + s), r12, sizeof I12) != 0 || !k.h_enabled[first + s]) aux_changes = true;
  }
  if (aux_changes && c->pending) { const int rc = rtuf_sync(c); if (rc != RTUF_OK) return rc; }
  if (!k.q_carried) {
    // first write after a batch took the other buffer: carry the streams this call does not set
    if (first != 0 || n != c->max_streams) memcpy(k.h_q[k.q_write], k.h_q[k.q_live], sizeof(double) * (size_t)c->max_streams * k.n_frames);
    k.q_carried = true;
  }
  memcpy(k.h_q[k.q_write] + (size_t)first * k.n_frames, q, sizeof(double) * (size_t)n * k.n_frames);
  k.dirty_q = true;
  if (aux_changes) {
    for (int s = 0; s < n; s++) {
      memcpy(k.h_root + 12 * (size_t)(first + s), &roots[12 * (size_t)s], sizeof I12);
      k.h_enabled[first + s] = 1;
    }
    k.dirty_aux = true;
  }
  // leaving the robot-mounted camera: both slots take the host-set camera transforms again
  if (camera_frame != k.camera_frame) for (auto& b : c->batch) b.dirty_cams = true;
  k.camera_frame = camera_frame;
  k.any_enabled = true;
  return RTUF_OK;
}

int rtuf_debug_read_poses(rtuf_context* c, int n, double* link_tf_out, double* cam_tf_out)
{
  KIDS_ONE(c, c->last_kid, rtuf_debug_read_poses(k, n, link_tf_out, cam_tf_out));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized || n <= 0 || n > c->max_streams) return c->fail(RTUF_ERR_INVALID, "bad arguments");
  hipSetDevice(c->device);
  HIP_TRY(c, hipStreamSynchronize(c->side));
  sync_lanes(c);
  const size_t L = (size_t)std::max(c->n_links, 1);
  if (link_tf_out) HIP_TRY(c, hipMemcpy(link_tf_out, c->batch[c->last_slot].d_link_tf, sizeof(double) * 16 * L * n, hipMemcpyDeviceToHost));
  if (cam_tf_out) {
    std::vector<Camera> cams(n);
    HIP_TRY(c, hipMemcpy(cams.data(), c->batch[c->last_slot].d_cams, sizeof(Camera) * n, hipMemcpyDeviceToHost));
    for (int s = 0; s < n; s++) memcpy(cam_tf_out + 16 * (size_t)s, cams[s].cam_tf, sizeof(double) * 16);
  }
  return RTUF_OK;
}

// ---- the hot path ---------------------------------------------------------------------
static hipEvent_t get_event(rtuf_context::Batch& b, size_t i)
{
  while (b.events.size() <= i) {
    hipEvent_t ev; hipEventCreate(&ev); b.events.push_back(ev);
  }
  return b.events[i];
}

// Everything one batch launches, as plain argument blocks: built first, then either enqueued kernel by kernel or --
// for small batches, whose cost is the host's launch overhead, not GPU time -- captured once into a hipGraph per
// batch slot and replayed with one call for as long as the blocks do not change (same buffers, sizes, parameters).
struct BatchPlan {
  std::vector<FkArgs> fks;
  PoseArgs pa{};
  struct Group { SetupArgs sa{}; TileArgs ta{}; CompareArgs ca{}; bool compare = false; int lane = 0; };
  std::vector<Group> groups;
  bool cover_pass = true;
  uint64_t hash() const
  {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t n) { const unsigned char* q = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 1099511628211ull; } };
    for (const FkArgs& f : fks) mix(&f, sizeof f);
    mix(&pa, sizeof pa);
    for (const Group& g : groups) { mix(&g.sa, sizeof g.sa); mix(&g.ta, sizeof g.ta); if (g.compare) mix(&g.ca, sizeof g.ca); mix(&g.lane, sizeof g.lane); }
    return h ^ (uint64_t)fks.size() << 56 ^ (uint64_t)groups.size() << 48 ^ (uint64_t)cover_pass << 47;
  }
};

#ifndef RTUF_CULL_ON_LANE
#define RTUF_CULL_ON_LANE 0        // (1: A/B switch -- only the first group of a one-lane batch culls in the pose stage, as before)
#endif
static constexpr int kGraphMaxStreams = 32;     // batches up to this size replay a captured hipGraph (timing off)

// Timing events of a batch: start and end of the pose stage, the end of every lane's part, and five per launch group --
// E0 before its first kernel (mode 1: the cull; mode 2: the set-up kernel), E1 after the set-up kernel (mode 2), E2 after
// clip + many-tile kernels, E3 after the tile kernel, E4 after the compare kernel (two-kernel mode).
enum { kEvStart = 0, kEvPoseEnd = 1, kEvLaneEnd = 2, kEvGroup0 = 2 + kMaxLanes, kEvPerGroup = 5 };

// Enqueues the kernels of a plan: pose stage (forward kinematics, matrix stacks) on `sp`, then every launch group on the
// stream of its lane, which waits for the pose stage through the batch's `posed` event.  The cull of every lane's first
// group still belongs to the pose stage (it then runs under the previous batch's raster kernels).
static int issue_plan(rtuf_context* c, rtuf_context::Batch& b, const BatchPlan& plan, hipStream_t sp, bool worst_case_grid)
{
  const bool two = (c->params.flags & RTUF_FLAG_TWO_KERNEL) != 0;
  worst_case_grid = worst_case_grid || c->force_worst_grid || (c->params.flags & RTUF_FLAG_STRICT_GRID) != 0;
  auto mark = [&](size_t i, hipStream_t s) { hipEventRecord(get_event(b, i), s); };
  if (b.timing) (void)get_event(b, kEvGroup0 + kEvPerGroup * plan.groups.size() - 1);      // (all of the batch's events exist)
  if (b.timing == 1) mark(kEvStart, sp);
  for (const FkArgs& fa : plan.fks) launch_fk(fa, sp);
  launch_pose(plan.pa, sp);
  if (b.timing == 1) mark(kEvPoseEnd, sp);
  // The cull of every lane's FIRST group belongs to the pose stage: it then runs under the raster kernels of the batch before
  // (its work list is the lane's own array of this batch slot, which nothing in flight reads), and the lane starts with the
  // set-up kernel instead of a 6 us kernel and the gap behind it (256 VGA streams +0.8 %, 64 x 720p +2.2 %, four groups of 64
  // +1.4 %).  Later groups of a lane reuse that array and cull in turn.  (Moving the other small kernel of a lane's chain --
  // the copy of the counters -- to a stream of its own behind an event was measured as well: -0.5 %, one camera -12 %.)
  auto cull_in_pose = [&](size_t g) {
    if (RTUF_CULL_ON_LANE) return c->n_lanes == 1 && g == 0;
    return g < (size_t)c->n_lanes && (g == 0 || plan.groups[g].lane != plan.groups[g - 1].lane);
  };
  for (size_t g = 0; g < plan.groups.size() && g < (size_t)c->n_lanes; g++)
    if (cull_in_pose(g)) launch_cull(plan.groups[g].sa, sp);
  HIP_TRY(c, hipEventRecord(b.posed, sp));
  for (int l = 0; l < c->n_lanes; l++)
    if (b.lanes_used >> l & 1u) HIP_TRY(c, hipStreamWaitEvent(c->lane[l].stream, b.posed, 0));
  for (size_t g = 0; g < plan.groups.size(); g++) {
    const BatchPlan::Group& gr = plan.groups[g];
    hipStream_t st = c->lane[gr.lane].stream;
    const size_t e0 = kEvGroup0 + kEvPerGroup * g;
    if (b.timing == 1) mark(e0, st);
    if (!cull_in_pose(g)) launch_cull(gr.sa, st);
    if (b.timing >= 2) mark(e0, st);                 // (after the wait for the pose stage: set-up time only)
    // The set-up grid is sized from the previous batch's work lists (scaled to this group's streams); the group's list
    // length comes back with its counters, and a batch whose list outgrew the grid is run again (retire_oldest).
    // A group of the same index and size as in the last retired batch takes ITS OWN list length (work lists are not linear in
    // the streams: every chunk rounds its visible streams up to whole items, and visibility differs per stream -- a smaller
    // last group can need more than its share of the largest group's list); any other group the longest list scaled to its
    // streams plus one item per chunk for that rounding.
    uint32_t hint = 0;
    if (!worst_case_grid && c->items_hint && c->items_hint_streams > 0) {
      const uint32_t scaled = (uint32_t)(((uint64_t)c->items_hint * (uint64_t)gr.sa.group_size + (uint64_t)c->items_hint_streams - 1) / (uint64_t)c->items_hint_streams);
      // (its own length, but never below half of what the longest list makes per stream: a group whose cameras saw nearly
      // nothing last time would otherwise take a grid of a few dozen workgroups, and the robot moving into ITS view would cost
      // a re-run of everything in flight where the old max-based estimate covered it; idle workgroups of the floor end at once)
      if (g < c->group_items.size() && c->group_streams[g] == gr.sa.group_size) hint = std::max(std::max(c->group_items[g], scaled / 2u), 1u);
      else hint = scaled + (uint32_t)c->n_chunks;
    }
    const uint32_t grid = launch_setup(gr.sa, hint, false, st);
    b.setup_grid[g] = worst_case_grid ? 0xffffffffu : grid;
    if (b.timing >= 2) mark(e0 + 1, st);
    launch_clip(gr.sa, st);
    launch_bigrec(gr.sa, plan.cover_pass, st);      // appends the many-tile records the two kernels above listed (after the cover pass, if it is on)
    if (b.timing) mark(e0 + 2, st);
    launch_tile(gr.ta, two, plan.cover_pass, st);
    if (b.timing) mark(e0 + 3, st);
    if (gr.compare) { launch_compare(gr.ca, st); if (b.timing) mark(e0 + 4, st); }
  }
  // every lane publishes the counter blocks of its own groups
  const int ng = (int)plan.groups.size();
  const PublishLimits lim = {c->capacity, c->fcapacity, c->big_capacity};
  for (int l = 0; l < c->n_lanes; l++) {
    if (!(b.lanes_used >> l & 1u)) continue;
    if (ng == 1) launch_publish_counters(b.d_counters, b.h_counters, 0, 1, 1, b.d_status, lim, c->lane[l].stream);
    else launch_publish_counters(b.d_counters, b.h_counters, l, c->n_lanes, (ng - l + c->n_lanes - 1) / c->n_lanes, b.d_status, lim, c->lane[l].stream);
    if (b.timing == 1) mark(kEvLaneEnd + l, c->lane[l].stream);
  }
  return RTUF_OK;
}

static int enqueue_batch(rtuf_context* c, rtuf_context::Batch& b, bool rerun)
{
  const int n = b.n;
  const float* d_depth = b.depth; float* d_masked = b.masked; uint8_t* d_mask = b.mask;
  const bool io_u16 = b.u16;
  const size_t esz = io_u16 ? sizeof(uint16_t) : sizeof(float);
  const bool two = (c->params.flags & RTUF_FLAG_TWO_KERNEL) != 0;
  // Launch groups: as many as the lanes' bins ask for, alternating between the lanes; a batch that is not split takes one
  // lane, the next such batch the other.
  const int n_groups = groups_for(c, n);
  const int per_group = (n + n_groups - 1) / n_groups;
  if (!rerun) {
    b.lanes_used = 0;
    if (n_groups == 1) { b.lanes_used = 1u << c->next_lane; c->next_lane = (c->next_lane + 1) % c->n_lanes; }
    else for (int l = 0; l < std::min(c->n_lanes, n_groups); l++) b.lanes_used |= 1u << l;
  } else if (n_groups > 1) {
    // (a re-run after the launch group shrank may need every lane where the first run needed one)
    for (int l = 0; l < std::min(c->n_lanes, n_groups); l++) b.lanes_used |= 1u << l;
  }
  const int lane0 = __builtin_ctz(b.lanes_used);
  hipStream_t st = c->lane[lane0].stream;        // (graph replay: single-lane contexts only)
  const size_t L = (size_t)std::max(c->n_links, 1);
  const size_t plane = (size_t)c->width * c->height;
  if (!rerun) {
    // mode 3 = mode 2 on every eighth batch only (each event costs ~5 us of stream time)
    b.timing = c->timing == 3 ? ((c->timing_seq++ & 7u) == 0 ? 2 : 0) : c->timing;
  }
  const bool use_graph = c->graphs_ok && c->n_lanes == 1 && !b.timing && n <= kGraphMaxStreams && n_groups == 1;
  // the pose stage of this batch runs on the side stream, concurrent with the raster kernels of the batch before it
  // (graph replay: everything hangs off the main stream, the side stream is forked inside the graph)
  hipStream_t sp = use_graph ? st : c->side;
  c->last_slot = (int)(&b - &c->batch[0]);
  // only what the host changed since the last batch crosses the bus (with on-device forward
  // kinematics that is just the joint positions below)
  const bool more = n > b.uploaded_streams;
  if (!rerun) {
    b.cam_idx = take_staged(c, c->cam_ring, b, [](const rtuf_context::Batch& o) { return o.cam_idx; });
    b.link_idx = take_staged(c, c->link_ring, b, [](const rtuf_context::Batch& o) { return o.link_idx; });
    c->h_cams = c->ring_cams[c->cam_ring.write];
    c->h_link_tf = c->ring_link_tf[c->link_ring.write];
  }
  // (a re-run after a bin regrowth finds this slot's device copies as its first run left them; its dirty flags may
  // already belong to setter calls made since, for the batch that comes next)
  if (!rerun) {
    if (b.dirty_cams || more) HIP_TRY(c, hipMemcpyAsync(b.d_cams, c->ring_cams[b.cam_idx], sizeof(Camera) * n, hipMemcpyHostToDevice, sp));
    if (b.dirty_link_tf || more) HIP_TRY(c, hipMemcpyAsync(b.d_link_tf, c->ring_link_tf[b.link_idx], sizeof(double) * 16 * L * n, hipMemcpyHostToDevice, sp));
    b.dirty_cams = b.dirty_link_tf = false;
  }
  if (c->dirty_mask || n > c->mask_uploaded_streams) HIP_TRY(c, hipMemcpyAsync(c->d_model_mask, c->h_model_mask, sizeof(uint64_t) * n, hipMemcpyHostToDevice, sp));
  c->dirty_mask = false;
  b.uploaded_streams = std::max(b.uploaded_streams, n);
  c->mask_uploaded_streams = std::max(c->mask_uploaded_streams, n);
  BatchPlan plan;
  if (!rerun) b.cover_pass = c->cover_on;
  plan.cover_pass = b.cover_pass;
  // on-device forward kinematics overwrites the link matrices (and camera) of the streams that use it
  for (HostModel& m : c->models) {
    Kinematics& k = m.kin;
    if (!k.n_frames || !k.any_enabled) continue;
    const size_t mi = (size_t)(&m - &c->models[0]);
    if (b.q_idx.size() < c->models.size()) b.q_idx.resize(c->models.size(), 0);
    if (!rerun) {
      if (k.dirty_q) {                       // this batch takes the staged joint positions ...
        k.q_live = k.q_write; k.q_carried = false; k.dirty_q = false;
        // ... and later writes go to a buffer no batch in flight reads (kMaxInflight + 1 buffers: one is always free)
        bool used[kMaxInflight + 1] = {};
        used[k.q_live] = true;
        for (const auto& o : c->batch) if (o.active && &o != &b && mi < o.q_idx.size()) used[o.q_idx[mi]] = true;
        for (int i = 0; i <= kMaxInflight; i++) if (!used[i]) { k.q_write = i; break; }
      }
      b.q_idx[mi] = k.q_live;
    }
    if (k.dirty_aux || n > k.aux_uploaded_streams) {
      HIP_TRY(c, hipMemcpyAsync(k.d_root, k.h_root, sizeof(double) * 12 * (size_t)n, hipMemcpyHostToDevice, sp));
      HIP_TRY(c, hipMemcpyAsync(k.d_enabled, k.h_enabled, (size_t)n, hipMemcpyHostToDevice, sp));
    }
    k.dirty_aux = false;
    k.aux_uploaded_streams = std::max(k.aux_uploaded_streams, n);
    FkArgs fa;
    memset(&fa, 0, sizeof fa);               // (padding too: the plan is hashed byte-wise)
    fa.parent = k.d_parent; fa.depth = k.d_depth; fa.max_depth = k.max_depth; fa.joint_type = k.d_type; fa.joint_origin = k.d_origin; fa.joint_axis = k.d_axis;
    fa.link_frame = k.d_link_frame; fa.link_offset = k.d_link_offset; fa.q = k.h_q[b.q_idx[mi]]; fa.root_tf = k.d_root;
    fa.enabled = k.d_enabled; fa.link_tf = b.d_link_tf; fa.cams = b.d_cams;
    fa.n_streams = n; fa.n_frames = k.n_frames; fa.n_links_model = (int)m.links.size(); fa.link_base = m.link_base;
    fa.n_links_total = (int)L; fa.camera_frame = k.camera_frame;
    plan.fks.push_back(fa);
  }
  PoseArgs& pa = plan.pa;
  memset(&pa, 0, sizeof pa);
  pa.cams = b.d_cams; pa.link_tf = b.d_link_tf; pa.draws = c->d_draws; pa.mvp = b.d_mvp;
  // to_linear_depth's constants exactly as the shader evaluates them (include/shaders/urdf_filter.frag:14-17), in float
  const float zn = c->params.near_plane, zf = c->params.far_plane;
  const float sc_num = (zn * zf) / (zn - zf), sc_off = zf / (zf - zn);
  // The per-pixel division num / (z - off) needs neither the operand scaling nor the special-case fix-up of the IEEE expansion
  // when no operand or intermediate can leave the normal range: |num| within 2^+-40, off in [1 + 2^-10, 2^20] (every z the
  // kernels hand to it lies in [-1, 1 + 2^-11]: |z - off| >= 2^-11).  The kernels then run its eight-instruction core
  // (shade_threshold; scripts/fdiv_check.hip compares the two forms over every float z of that range); anything else -- a far
  // plane more than a thousand times the near plane, non-finite parameters -- keeps the full expansion.
  const float sc_abs = std::fabs(sc_num);
  const int fast_div = std::isfinite(sc_num) && std::isfinite(sc_off) && sc_abs >= 0x1p-40f && sc_abs <= 0x1p40f && sc_off >= 1.0f + 0x1p-10f && sc_off <= 0x1p20f;
  pa.bg = b.d_bg; pa.counters = b.d_counters; pa.n_counters = n_groups; pa.status = b.d_status;
  pa.sc_num = sc_num; pa.sc_off = sc_off; pa.max_diff = c->params.depth_distance_threshold;
  pa.n_streams = n; pa.n_draws = c->n_draws; pa.n_links = (int)L; pa.z_far = c->params.far_plane;
  pa.width = c->width; pa.height = c->height;
  for (int g = 0, base = 0; base < n; g++, base += per_group) {
    const int gs = std::min(per_group, n - base);
    plan.groups.emplace_back();
    BatchPlan::Group& gr = plan.groups.back();
    memset(&gr.sa, 0, sizeof gr.sa); memset(&gr.ta, 0, sizeof gr.ta); memset(&gr.ca, 0, sizeof gr.ca);
    gr.lane = n_groups == 1 ? lane0 : g % c->n_lanes;
    const rtuf_context::Lane& ln = c->lane[gr.lane];
    Counters* const d_counters = b.d_counters + g;
    SetupArgs& sa = gr.sa;
    sa.cverts = c->d_cverts; sa.ctris = c->d_ctris; sa.corder = c->d_corder; sa.chunks = c->d_chunks; sa.mvp = b.d_mvp;
    sa.model_mask = c->d_model_mask; sa.bg = b.d_bg;
    sa.bins = ln.d_bins; sa.bin_hdr = ln.d_bin_hdr; sa.fbins = ln.d_fbins; sa.fbin_count = ln.d_fbin_count; sa.fcapacity = c->fcapacity; sa.capacity = c->capacity;
    sa.clip_list = ln.d_clip_list; sa.clip_spill = ln.d_clip_spill; sa.big_list = ln.d_big_list; sa.big_capacity = c->big_capacity; sa.counters = d_counters; sa.group_base = base; sa.group_size = gs;
    sa.n_draws = c->n_draws; sa.width = c->width; sa.height = c->height; sa.tiles_x = c->tiles_x; sa.tiles_y = c->tiles_y;
    sa.clip_capacity = c->clip_capacity; sa.bg_chunk = c->bg_chunk;
    sa.items = ln.d_items[c->last_slot]; sa.n_chunks = c->n_chunks; sa.flags = c->params.flags;
    TileArgs& ta = gr.ta;
    ta.bins = ln.d_bins; ta.bin_hdr = ln.d_bin_hdr; ta.fbins = ln.d_fbins; ta.fbin_count = ln.d_fbin_count; ta.fcapacity = c->fcapacity; ta.capacity = c->capacity;
    ta.depth = d_depth; ta.masked = d_masked; ta.mask = d_mask;
    ta.zsurface = ln.d_zsurface; ta.bg = b.d_bg; ta.counters = d_counters;
    ta.big_list = ln.d_big_list;
    ta.group_base = base; ta.group_size = gs; ta.width = c->width; ta.height = c->height;
    ta.tiles_x = c->tiles_x; ta.tiles_y = c->tiles_y; ta.flags = c->params.flags;
    ta.z_near = c->params.near_plane; ta.z_far = c->params.far_plane;
    ta.max_diff = c->params.depth_distance_threshold; ta.replace_value = c->params.filter_replace_value;
    ta.sc_num = sc_num; ta.sc_off = sc_off;
    ta.io_u16 = io_u16 ? 1 : 0;
    ta.key_shift = c->key_shift;
    ta.fast_div = fast_div;
    ta.bits = b.bits;
    gr.compare = two && !b.bits;
    if (gr.compare) {
      CompareArgs& ca = gr.ca;
      ca.depth = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_depth) + (size_t)base * plane * esz); ca.zsurface = ln.d_zsurface;
      ca.masked = reinterpret_cast<float*>(reinterpret_cast<char*>(d_masked) + (size_t)base * plane * esz);
      ca.io_u16 = io_u16 ? 1 : 0; ca.mask = d_mask ? d_mask + (size_t)base * plane : nullptr;
      ca.n_pixels = (size_t)gs * plane;
      ca.z_near = ta.z_near; ca.z_far = ta.z_far; ca.max_diff = ta.max_diff; ca.replace_value = ta.replace_value;
      ca.sc_num = sc_num; ca.sc_off = sc_off; ca.fast_div = fast_div;
    }
  }
  b.n_groups = (int)plan.groups.size();
  if (b.n_groups != n_groups) return c->fail(RTUF_ERR_STATE, "internal: %d launch groups planned, %d made (the status word counts the former down)", n_groups, b.n_groups);
  b.setup_grid.assign(plan.groups.size(), 0xffffffffu);
  c->last_lane = plan.groups.back().lane;
  // host-plane batches: the lanes' first kernels wait for the upload of the planes
  if (b.wait_upload)
    for (int l = 0; l < c->n_lanes; l++)
      if (b.lanes_used >> l & 1u) HIP_TRY(c, hipStreamWaitEvent(c->lane[l].stream, b.uploaded, 0));
  bool launched = false;
  if (use_graph) {
    // Small batch: its ~8 launches, two event operations and the cross-stream wait cost the host more than the GPU
    // spends on the kernels.  Capture them once per batch slot (pose stage forked onto the side stream inside the
    // graph; the set-up grid is the worst case, so no list-length estimate is baked in) and replay with one call.
    const uint64_t h = plan.hash();
    hipGraphExec_t exec = nullptr;
    for (auto& g : b.graphs) if (g.exec && g.hash == h) exec = g.exec;
    // A capture that has to evict a LIVE entry of the slot's cache is what thrashing looks like (a caller that never repeats
    // an argument set: fresh output buffers every frame); filling free entries -- the ring of staging buffers, the cover
    // pass going on and off, a regrowth -- is not.  Judged over windows of 64 batches, so a burst long ago does not count.
    if (exec) c->graph_hits++;
    else {
      c->graph_misses++;
      if (b.graphs[b.graph_next].exec) c->graph_evictions++;
    }
    if (((c->graph_hits + c->graph_misses) & 63u) == 0u) {
      if (c->graph_evictions > 32) c->graphs_ok = false;       // more than half of the window's batches captured over a live entry
      c->graph_evictions = 0;
    }
    if (!exec && c->graphs_ok) {
      hipGraph_t graph = nullptr;
      // (the side stream joins the capture below: it must not still carry plain launches of an earlier batch)
      if (c->side_used_plain) { hipStreamSynchronize(c->side); c->side_used_plain = false; }
      hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
      if (e == hipSuccess) {
        if (!c->fork_ev) hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming);
        hipEventRecord(c->fork_ev, st);
        hipStreamWaitEvent(c->side, c->fork_ev, 0);
        const int rc = issue_plan(c, b, plan, c->side, true);
        e = hipStreamEndCapture(st, &graph);
        if (rc != RTUF_OK) e = hipErrorUnknown;
      }
      if (e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph) hipGraphDestroy(graph);
      if (e != hipSuccess || !exec) {
        // no graphs on this runtime: fall back to plain launches for the life of the context
        (void)hipGetLastError();
        exec = nullptr;
        c->graphs_ok = false;
      } else {
        auto& slot = b.graphs[b.graph_next];
        b.graph_next = (b.graph_next + 1) % rtuf_context::Batch::kGraphs;
        if (slot.exec) hipGraphExecDestroy(slot.exec);       // (not in flight: a slot's batches are retired before it is reused)
        slot.exec = exec;
        slot.hash = h;
      }
    }
    if (exec) {
      HIP_TRY(c, hipGraphLaunch(exec, st));
      b.setup_grid.assign(plan.groups.size(), 0xffffffffu);          // worst-case grid: the work list always fits
      launched = true;
    }
  }
  if (!launched) {
    if (!use_graph) c->side_used_plain = true;
    const int rc = issue_plan(c, b, plan, use_graph ? st : sp, false);
    if (rc != RTUF_OK) return rc;
  }
  for (int l = 0; l < c->n_lanes; l++)
    if (b.lanes_used >> l & 1u) HIP_TRY(c, hipEventRecord(b.done[l], c->lane[l].stream));
  HIP_TRY(c, hipGetLastError());
  return RTUF_OK;
}

// Copies the results of a host-plane batch to the caller's planes on the download stream, after the
// batch's kernels; consecutive planes go out as one transfer.
static int enqueue_download(rtuf_context* c, rtuf_context::Batch& b)
{
  const size_t plane = (size_t)c->width * c->height;
  const size_t esz = b.u16 ? sizeof(uint16_t) : sizeof(float);
  for (int l = 0; l < c->n_lanes; l++)
    if (b.lanes_used >> l & 1u) HIP_TRY(c, hipStreamWaitEvent(c->d2h, b.done[l], 0));
  if (b.bits) {
    const size_t words = (size_t)c->height * (size_t)((c->width + 31) / 32);
    for (int s = 0; s < b.n;) {
      int e = s + 1;
      while (e < b.n && (char*)b.h_bits[e] == (char*)b.h_bits[e - 1] + words * 4) e++;
      HIP_TRY(c, hipMemcpyAsync(b.h_bits[s], b.st_bits + (size_t)s * words, (size_t)(e - s) * words * 4, hipMemcpyDeviceToHost, c->d2h));
      s = e;
    }
    HIP_TRY(c, hipEventRecord(b.downloaded, c->d2h));
    return RTUF_OK;
  }
  for (int s = 0; s < b.n;) {
    int e = s + 1;
    while (e < b.n && (char*)b.h_masked[e] == (char*)b.h_masked[e - 1] + plane * esz) e++;
    HIP_TRY(c, hipMemcpyAsync(b.h_masked[s], (char*)b.st_masked + (size_t)s * plane * esz, (size_t)(e - s) * plane * esz, hipMemcpyDeviceToHost, c->d2h));
    s = e;
  }
  for (int s = 0; s < b.n;) {
    if (!b.h_mask[s]) { s++; continue; }
    int e = s + 1;
    while (e < b.n && b.h_mask[e] && (char*)b.h_mask[e] == (char*)b.h_mask[e - 1] + plane) e++;
    HIP_TRY(c, hipMemcpyAsync(b.h_mask[s], b.st_mask + (size_t)s * plane, (size_t)(e - s) * plane, hipMemcpyDeviceToHost, c->d2h));
    s = e;
  }
  HIP_TRY(c, hipEventRecord(b.downloaded, c->d2h));
  return RTUF_OK;
}

// Retires the oldest batch in flight: waits for it, reads its counters, and if a bin or the clip list
// overflowed, enlarges them and runs that batch and every later one again (their inputs are intact).
static int retire_oldest(rtuf_context* c)
{
  for (int attempt = 0; attempt < 8; attempt++) {
    rtuf_context::Batch& b = c->batch[c->oldest];
    if (!c->pending || !b.active) { c->pending = 0; return RTUF_OK; }
    for (int l = 0; l < c->n_lanes; l++)
      if (b.lanes_used >> l & 1u) HIP_TRY(c, hipEventSynchronize(b.done[l]));
    struct { unsigned long long tris_binned = 0, bin_entries = 0, clip_count = 0, frags = 0, occluded = 0, raster_atomics = 0, drawn_pixels = 0, work_items = 0;
             unsigned max_bin_fill = 0, max_fbin_fill = 0, clip_overflow = 0, uncovered = 0, max_big_fill = 0, cover_tiles = 0, exact_tiles = 0, zero_items = 0,
                      max_items = 0; } k;
    bool list_over = false;                    // some group's set-up grid was sized too small for its work list
    int group_streams = 0;
#ifdef RTUF_LANECOUNT
    for (int i = 0; i < kLaneLoops; i++) c->lane_slots[i] = c->lane_live[i] = 0;
    for (int g = 0; g < b.n_groups; g++)
      for (int sh = 0; sh < kCounterShards; sh++)
        for (int i = 0; i < kLaneLoops; i++) { c->lane_slots[i] += b.h_counters[g].shard[sh].lane_slots[i]; c->lane_live[i] += b.h_counters[g].shard[sh].lane_live[i]; }
#endif
    for (int g = 0; g < b.n_groups; g++) {
      const Counters& cn = b.h_counters[g];
      k.work_items += cn.work.n_items;
      k.max_items = std::max(k.max_items, cn.work.n_items);
      list_over = list_over || cn.work.n_items > b.setup_grid[g];
      for (int i = 0; i < kCounterShards; i++) {
        const CounterShard& sh = cn.shard[i];
        k.tris_binned += sh.tris_binned; k.bin_entries += sh.bin_entries; k.clip_count += sh.clip_count;
        k.max_bin_fill = std::max(k.max_bin_fill, sh.max_bin_fill); k.clip_overflow |= sh.clip_overflow;
        k.max_fbin_fill = std::max(k.max_fbin_fill, sh.max_fbin_fill); k.frags += sh.frags; k.uncovered |= sh.uncovered;
        k.max_big_fill = std::max(k.max_big_fill, sh.max_big_fill);
        k.occluded += sh.occluded; k.cover_tiles += sh.cover_tiles; k.exact_tiles += sh.exact_tiles; k.zero_items += sh.zero_items;
        k.raster_atomics += sh.raster_atomics; k.drawn_pixels += sh.drawn_pixels;
      }
    }
    group_streams = (b.n + b.n_groups - 1) / std::max(b.n_groups, 1);
    c->items_hint = k.max_items;               // sizes the next batches' set-up grids (the longest list of this batch's groups ...
    c->items_hint_streams = group_streams;     // ... of so many streams each)
    c->group_items.resize((size_t)b.n_groups); c->group_streams.resize((size_t)b.n_groups);
    for (int g = 0; g < b.n_groups; g++) {
      c->group_items[(size_t)g] = b.h_counters[g].work.n_items;
      c->group_streams[(size_t)g] = std::min(group_streams, b.n - g * group_streams);
    }
    c->stats.triangles_submitted = (uint64_t)c->n_tris * (uint64_t)b.n;
    c->stats.triangles_binned = k.tris_binned;
    c->stats.bin_entries = k.bin_entries;
    c->stats.triangles_clipped = k.clip_count;
    c->stats.max_bin_fill = k.max_bin_fill;
    c->stats.bin_capacity = c->capacity;
    c->stats.fragments_binned = k.frags;
    c->stats.max_fbin_fill = k.max_fbin_fill;
    c->stats.occluded_entries = k.occluded; c->stats.cover_tiles = k.cover_tiles; c->stats.exact_tiles = k.exact_tiles;
    c->stats.work_items = (uint32_t)std::min<unsigned long long>(k.work_items, 0xffffffffu); c->stats.zero_survivor_items = k.zero_items;
    c->stats.raster_atomics = k.raster_atomics; c->stats.drawn_pixels = k.drawn_pixels;
    c->stats.cover_pass = b.cover_pass ? 1u : 0u;
    c->stats.groups_last_batch = (uint32_t)b.n_groups;
    const bool bin_over = k.max_bin_fill > c->capacity || k.max_fbin_fill > c->fcapacity;
    const bool clip_over = k.clip_overflow != 0;
    const bool big_over = k.max_big_fill > c->big_capacity;
    if (list_over) c->stats.regrowths++;
    // host mirror of the device status word (rtuf_batch_status_device): what the FIRST run of the batch being retired left there
    if (attempt == 0) {
      c->stats.batch_status = (bin_over ? kStatusBinOverflow : 0u) | (clip_over ? kStatusClipOverflow : 0u) | (big_over ? kStatusBigOverflow : 0u) |
                              (list_over ? kStatusGridShort : 0u) | (k.uncovered ? kStatusUncovered : 0u);
      c->stats.batch_reruns = 0;
    }
    if (!bin_over && !clip_over && !list_over && !big_over) {
      if (b.host_io) HIP_TRY(c, hipEventSynchronize(b.downloaded));
      if (b.timing && b.events.size() >= (size_t)(kEvGroup0 + kEvPerGroup * b.n_groups)) {
        // per launch group E0 .. E4 (see issue_plan).  With several lanes the kernels of different groups overlap: the sums
        // below add up per-launch durations, they are not wall time.
        const bool two = (c->params.flags & RTUF_FLAG_TWO_KERNEL) != 0 && !b.bits;
        auto el = [&](size_t i, size_t j) { float ms = 0; hipEventElapsedTime(&ms, b.events[i], b.events[j]); return ms; };
        c->stats.ms_pose = c->stats.ms_setup = c->stats.ms_clip = c->stats.ms_raster = c->stats.ms_compare = c->stats.ms_total = 0;
        for (int g = 0; g < b.n_groups; g++) {
          const size_t e0 = kEvGroup0 + kEvPerGroup * (size_t)g;
          if (b.timing == 1) {
            c->stats.ms_setup += el(e0, e0 + 2);
          } else {
            c->stats.ms_setup += el(e0, e0 + 1);
            c->stats.ms_clip += el(e0 + 1, e0 + 2);
          }
          c->stats.ms_raster += el(e0 + 2, e0 + 3);
          if (two) c->stats.ms_compare += el(e0 + 3, e0 + 4);
        }
        if (b.timing == 1) {
          c->stats.ms_pose = el(kEvStart, kEvPoseEnd);
          for (int l = 0; l < c->n_lanes; l++)
            if (b.lanes_used >> l & 1u) c->stats.ms_total = std::max(c->stats.ms_total, el(kEvStart, kEvLaneEnd + l));
        }
        c->acc_ms[0] += c->stats.ms_pose; c->acc_ms[1] += c->stats.ms_setup; c->acc_ms[2] += c->stats.ms_raster;
        c->acc_ms[3] += c->stats.ms_compare; c->acc_ms[4] += c->stats.ms_total; c->acc_ms[5] += c->stats.ms_clip;
        c->acc_batches++;
        c->stats.timed_batches = c->acc_batches;
        c->stats.sum_ms_pose = c->acc_ms[0]; c->stats.sum_ms_setup = c->acc_ms[1]; c->stats.sum_ms_raster = c->acc_ms[2];
        c->stats.sum_ms_compare = c->acc_ms[3]; c->stats.sum_ms_total = c->acc_ms[4]; c->stats.sum_ms_clip = c->acc_ms[5];
      }
      if (b.cover_pass) {
        if (k.cover_tiles) c->cover_idle = 0;
        else if (++c->cover_idle >= 3) { c->cover_on = false; c->cover_sleep = 64; c->cover_idle = 2; }      // (idle = 2: one empty probe sends it back to sleep)
      } else if (--c->cover_sleep <= 0) {
        c->cover_on = true;
      }
      b.active = false;
      c->oldest = (c->oldest + 1) % kMaxInflight;
      c->pending--;
      if (b.bits && k.uncovered)
        return c->fail(RTUF_ERR_STATE, "mask bits: a stream's background quad does not cover its whole image (non-standard projection), so "
                                       "masked depth != select(bit, replace, sensor) there; use the full-plane calls for this camera");
      return RTUF_OK;
    }
    // overflow: wait for the later batches too, enlarge, and run everything in flight again in order
    sync_lanes(c);
    if (c->d2h) HIP_TRY(c, hipStreamSynchronize(c->d2h));
    auto give_up = [&](int rc) { for (auto& o : c->batch) o.active = false; c->pending = 0; return rc; };
    if (bin_over) { const int rc = grow_bins(c, k.max_bin_fill, k.max_fbin_fill); if (rc != RTUF_OK) return give_up(rc); }
    if (clip_over) {
      const uint32_t larger = c->clip_capacity * 4;
      for (int l = 0; l < c->n_lanes; l++) {
        int rc = regrow(c, c->lane[l].d_clip_list, (size_t)larger * kCounterShards * sizeof(ClipItem), "clip list");
        if (rc == RTUF_OK && clip_spill_bytes(larger) != clip_spill_bytes(c->clip_capacity)) { rc = regrow(c, c->lane[l].d_clip_spill, clip_spill_bytes(larger), "clip spill area"); c->stats.regrowths--; }
        if (rc != RTUF_OK) return give_up(rc);
        if (l) c->stats.regrowths--;           // (one regrowth, however many lanes)
      }
      c->clip_capacity = larger;
    }
    if (big_over) {
      // (half as much again as this run asked for: poses move, and a list that just fits overflows on the next batch)
      uint32_t larger = c->big_capacity;
      while (larger < k.max_big_fill + k.max_big_fill / 2) larger *= 2;
      for (int l = 0; l < c->n_lanes; l++) {
        const int rc = regrow(c, c->lane[l].d_big_list, (size_t)larger * kCounterShards * sizeof(BigRec), "many-tile list");
        if (rc != RTUF_OK) return give_up(rc);
        if (l) c->stats.regrowths--;
      }
      c->big_capacity = larger;
    }
    const size_t tiles = (size_t)c->tiles_x * c->tiles_y;
    for (int l = 0; l < c->n_lanes; l++) {
      // (the arrays were sized for the launch group the context started with: at least the present one)
      launch_init_headers(c->lane[l].d_bin_hdr, (size_t)c->group * tiles, c->lane[l].stream);
      HIP_TRY(c, hipMemsetAsync(c->lane[l].d_fbin_count, 0, (size_t)c->group * tiles * sizeof(uint32_t), c->lane[l].stream));
    }
    // (a grid that was too short: the re-run takes the worst-case grid, which no list can outgrow -- the estimate that failed
    // once would size the same grid again)
    c->force_worst_grid = list_over;
    c->stats.batch_reruns++;
    for (int i = 0; i < c->pending; i++) {
      rtuf_context::Batch& r = c->batch[(c->oldest + i) % kMaxInflight];
      int rc = enqueue_batch(c, r, true);
      if (rc == RTUF_OK && r.host_io) rc = enqueue_download(c, r);
      if (rc != RTUF_OK) { c->force_worst_grid = false; for (auto& o : c->batch) o.active = false; c->pending = 0; return rc; }
    }
    c->force_worst_grid = false;
  }
  for (auto& o : c->batch) o.active = false;
  c->pending = 0;
  return c->fail(RTUF_ERR_CAPACITY, "tile bins still overflow after regrowth");
}

static int submit_batch(rtuf_context* c, int n, const float* d_depth, float* d_masked, uint8_t* d_mask, bool u16, uint32_t* d_bits = nullptr,
                        bool wait_upload = false)
{
  if (c->broken) return c->fail(RTUF_ERR_STATE, "context unusable: a bin regrowth failed (%s)", c->error.c_str());
  hipSetDevice(c->device);
  if (c->params.flags & RTUF_FLAG_TWO_KERNEL)
    for (int l = 0; l < c->n_lanes; l++)
      if (!c->lane[l].d_zsurface) HIP_TRY(c, dev_alloc(c, &c->lane[l].d_zsurface, (size_t)c->group * c->width * c->height * sizeof(float)));
  // two-kernel mode keeps one z-surface per lane: its batches do not overlap
  const int limit = (c->params.flags & RTUF_FLAG_TWO_KERNEL) ? 1 : kMaxInflight;
  while (c->pending >= limit) { const int rc = retire_oldest(c); if (rc != RTUF_OK) return rc; }
  rtuf_context::Batch& b = c->batch[(c->oldest + c->pending) % kMaxInflight];
  b.n = n; b.depth = d_depth; b.masked = d_masked; b.mask = d_mask; b.u16 = u16; b.host_io = false; b.bits = d_bits;
  b.wait_upload = wait_upload;
  const int rc = enqueue_batch(c, b, false);
  if (rc == RTUF_OK) { b.active = true; c->pending++; }
  return rc;
}

int rtuf_filter_batch_device(rtuf_context* c, int n, const float* d_depth, float* d_masked, uint8_t* d_mask)
{
  KIDS_NEXT(c, rtuf_filter_batch_device(k, n, d_depth, d_masked, d_mask));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (n <= 0 || n > c->max_streams || !d_depth || !d_masked) return c->fail(RTUF_ERR_INVALID, "bad batch arguments (n=%d)", n);
  return submit_batch(c, n, d_depth, d_masked, d_mask, false);
}

int rtuf_filter_batch_device_u16(rtuf_context* c, int n, const uint16_t* d_depth, uint16_t* d_masked, uint8_t* d_mask)
{
  KIDS_NEXT(c, rtuf_filter_batch_device_u16(k, n, d_depth, d_masked, d_mask));
  if (!c) return RTUF_ERR_INVALID;
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (n <= 0 || n > c->max_streams || !d_depth || !d_masked) return c->fail(RTUF_ERR_INVALID, "bad batch arguments (n=%d)", n);
  if (c->width & 3) return c->fail(RTUF_ERR_INVALID, "16UC1 path needs a width that is a multiple of 4");
  return submit_batch(c, n, reinterpret_cast<const float*>(d_depth), reinterpret_cast<float*>(d_masked), d_mask, true);
}

static int check_bits_call(rtuf_context* c, int n, const void* in, const void* out)
{
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (n <= 0 || n > c->max_streams || !in || !out) return c->fail(RTUF_ERR_INVALID, "bad batch arguments (n=%d)", n);
  if (c->width & 3) return c->fail(RTUF_ERR_INVALID, "mask-bits output needs a width that is a multiple of 4");
  if (c->params.flags & RTUF_FLAG_TWO_KERNEL) return c->fail(RTUF_ERR_INVALID, "mask-bits output exists in fused mode only (RTUF_FLAG_TWO_KERNEL is set)");
  return RTUF_OK;
}

size_t rtuf_mask_bits_words(int width, int height)
{
  return (width > 0 && height > 0) ? (size_t)height * (size_t)((width + 31) / 32) : 0;
}

int rtuf_filter_batch_device_bits(rtuf_context* c, int n, const float* d_depth, uint32_t* d_bits)
{
  KIDS_NEXT(c, rtuf_filter_batch_device_bits(k, n, d_depth, d_bits));
  if (!c) return RTUF_ERR_INVALID;
  const int rc = check_bits_call(c, n, d_depth, d_bits);
  return rc != RTUF_OK ? rc : submit_batch(c, n, d_depth, nullptr, nullptr, false, d_bits);
}

int rtuf_filter_batch_device_bits_u16(rtuf_context* c, int n, const uint16_t* d_depth, uint32_t* d_bits)
{
  KIDS_NEXT(c, rtuf_filter_batch_device_bits_u16(k, n, d_depth, d_bits));
  if (!c) return RTUF_ERR_INVALID;
  const int rc = check_bits_call(c, n, d_depth, d_bits);
  return rc != RTUF_OK ? rc : submit_batch(c, n, reinterpret_cast<const float*>(d_depth), nullptr, nullptr, true, d_bits);
}

// Host side of the mask-bits calls: masked depth / byte mask of one frame from its sensor plane and its mask bits,
// with the arithmetic of the kernels (a pure select; for 16UC1 the reference's two convertTo roundings, see
// metres_to_u16 in rtuf_kernels.hip).  No GPU involved.
int rtuf_expand_mask_bits(const void* depth_in, int is_u16, const uint32_t* bits, int width, int height, float replace_value,
                          void* masked_out, uint8_t* mask_out)
{
  if (!depth_in || !bits || width <= 0 || height <= 0 || (!masked_out && !mask_out)) return RTUF_ERR_INVALID;
  const int row_words = (width + 31) / 32;
  uint16_t rep16 = 0;
  if (is_u16) {
    const float v = replace_value * 1000.0f;
    if (v >= -2147483648.0f && v < 2147483648.0f) { const long r = lrintf(v); rep16 = (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r)); }
  }
  for (int y = 0; y < height; y++) {
    const uint32_t* wrow = bits + (size_t)y * row_words;
    const size_t o = (size_t)y * width;
    for (int x = 0; x < width; x++) {
      const bool f = (wrow[x >> 5] >> (x & 31)) & 1u;
      if (mask_out) mask_out[o + x] = f ? 255 : 0;
      if (!masked_out) continue;
      if (is_u16) {
        const uint16_t u = static_cast<const uint16_t*>(depth_in)[o + x];
        uint16_t r = rep16;
        if (!f) {                              // float(u) * 0.001f * 1000.0f, rounded half to even and saturated, like the kernels
          const float m = (float)u * 0.001f;
          const float v = m * 1000.0f;
          const long q = lrintf(v);
          r = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
        }
        static_cast<uint16_t*>(masked_out)[o + x] = r;
      } else {
        static_cast<float*>(masked_out)[o + x] = f ? replace_value : static_cast<const float*>(depth_in)[o + x];
      }
    }
  }
  return RTUF_OK;
}

int rtuf_sync(rtuf_context* c)
{
  if (!c) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) {
    // every kid is synchronised even if one fails (the first failure is reported)
    int first_rc = RTUF_OK;
    c->order.clear();
    for (rtuf_context* k : c->kids) { const int rc = rtuf_sync(k); if (rc < 0 && first_rc == RTUF_OK) { first_rc = rc; c->error = k->error; } }
    return first_rc;
  }
  hipSetDevice(c->device);
  while (c->pending) { const int rc = retire_oldest(c); if (rc != RTUF_OK) return rc; }
  for (int l = 0; l < c->n_lanes; l++) HIP_TRY(c, hipStreamSynchronize(c->lane[l].stream));
  return RTUF_OK;
}

void* rtuf_stream(rtuf_context* c) { return (c && c->kids.empty() && c->n_lanes == 1) ? (void*)c->lane[0].stream : nullptr; }

int rtuf_batch_status_device(rtuf_context* c, const uint32_t** d_status)
{
  if (!c || !d_status) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) c = c->kids[c->last_kid];
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  *d_status = c->batch[c->last_slot].d_status;
  return RTUF_OK;
}

int rtuf_order_stream_after_batches(rtuf_context* c, void* hip_stream)
{
  KIDS_ALL(c, rtuf_order_stream_after_batches(k, hip_stream));
  if (!c) return RTUF_ERR_INVALID;
  hipSetDevice(c->device);
  hipStream_t user = static_cast<hipStream_t>(hip_stream);
  for (const auto& b : c->batch) {
    if (!b.active) continue;
    for (int l = 0; l < c->n_lanes; l++)
      if (b.lanes_used >> l & 1u) HIP_TRY(c, hipStreamWaitEvent(user, b.done[l], 0));
    if (b.host_io && b.downloaded) HIP_TRY(c, hipStreamWaitEvent(user, b.downloaded, 0));
  }
  return RTUF_OK;
}

// ---- host planes -----------------------------------------------------------------------------------
// The reference's filter() takes a host buffer and leaves host results (src/urdf_filter.cpp:233-234,
// :729-735).  Here the planes of a batch go up on one copy stream and come back on another, so with two
// batches in flight the transfers of one overlap the kernels of the other; every slot has its own
// device staging.
static int submit_host_batch(rtuf_context* c, int n, const void* const* depth_in, void* const* masked_out,
                             void* const* mask_out, bool u16, uint32_t* const* bits_out = nullptr)
{
  if (!c->finalized) return c->fail(RTUF_ERR_STATE, "call rtuf_finalize_models first");
  if (n <= 0 || n > c->max_streams || !depth_in || (!masked_out && !bits_out)) return c->fail(RTUF_ERR_INVALID, "bad batch arguments (n=%d)", n);
  if (u16 && (c->width & 3)) return c->fail(RTUF_ERR_INVALID, "16UC1 path needs a width that is a multiple of 4");
  if (bits_out) { const int rc = check_bits_call(c, n, depth_in, bits_out); if (rc != RTUF_OK) return rc; }
  for (int s = 0; s < n; s++)
    if (!depth_in[s] || (bits_out ? !bits_out[s] : !masked_out[s])) return c->fail(RTUF_ERR_INVALID, "null plane for stream %d", s);
  hipSetDevice(c->device);
  // (copy streams beside the lanes as well: an upload queued behind a lane's kernels would hold up the next batch)
  if (!c->h2d || !c->d2h) {
    bool beside = true;
    sync_lanes(c);                      // (the probe times idle kernels: nothing else may be running; first host-plane call only)
    bool beside_up = true;
    if (!c->h2d) HIP_TRY(c, create_stream_beside(lane_streams(c), &c->h2d, &beside_up));
    if (!c->d2h) HIP_TRY(c, create_stream_beside(lane_streams(c), &c->d2h, &beside));
    c->stats.copy_streams_side_by_side = (beside && beside_up) ? 1u : 0u;
  }
  const int limit = (c->params.flags & RTUF_FLAG_TWO_KERNEL) ? 1 : kMaxInflight;
  while (c->pending >= limit) { const int rc = retire_oldest(c); if (rc != RTUF_OK) return rc; }
  rtuf_context::Batch& b = c->batch[(c->oldest + c->pending) % kMaxInflight];     // the slot submit_batch takes next
  const size_t plane = (size_t)c->width * c->height;
  if (b.st_streams < (size_t)n) {
    dev_free(c, b.st_depth); dev_free(c, b.st_masked); dev_free(c, b.st_mask);
    b.st_streams = 0;
    HIP_TRY(c, dev_alloc(c, &b.st_depth, (size_t)n * plane * sizeof(float)));     // float-sized: large enough for uint16 planes
    HIP_TRY(c, dev_alloc(c, &b.st_masked, (size_t)n * plane * sizeof(float)));
    HIP_TRY(c, dev_alloc(c, &b.st_mask, (size_t)n * plane));
    b.st_streams = (size_t)n;
  }
  if (bits_out && b.st_bits_streams < (size_t)n) {
    dev_free(c, b.st_bits);
    b.st_bits_streams = 0;
    HIP_TRY(c, dev_alloc(c, &b.st_bits, (size_t)n * rtuf_mask_bits_words(c->width, c->height) * sizeof(uint32_t)));
    b.st_bits_streams = (size_t)n;
  }
  if (!b.uploaded) HIP_TRY(c, hipEventCreateWithFlags(&b.uploaded, hipEventDisableTiming));
  if (!b.downloaded) HIP_TRY(c, hipEventCreateWithFlags(&b.downloaded, hipEventDisableTiming));
  const size_t esz = u16 ? sizeof(uint16_t) : sizeof(float);
  for (int s = 0; s < n;) {
    int e = s + 1;
    while (e < n && (const char*)depth_in[e] == (const char*)depth_in[e - 1] + plane * esz) e++;
    HIP_TRY(c, hipMemcpyAsync((char*)b.st_depth + (size_t)s * plane * esz, depth_in[s], (size_t)(e - s) * plane * esz, hipMemcpyHostToDevice, c->h2d));
    s = e;
  }
  HIP_TRY(c, hipEventRecord(b.uploaded, c->h2d));          // (the lanes' first kernels wait for it: enqueue_batch)
  bool any_mask = false;
  b.h_bits.clear();
  if (bits_out) {
    b.h_bits.assign(reinterpret_cast<void* const*>(bits_out), reinterpret_cast<void* const*>(bits_out) + n);
  } else {
    b.h_masked.assign(masked_out, masked_out + n);
    b.h_mask.assign((size_t)n, nullptr);
    if (mask_out) for (int s = 0; s < n; s++) { b.h_mask[s] = mask_out[s]; any_mask |= mask_out[s] != nullptr; }
  }
  int rc = submit_batch(c, n, b.st_depth, b.st_masked, any_mask ? b.st_mask : nullptr, u16, bits_out ? b.st_bits : nullptr, true);
  if (rc != RTUF_OK) return rc;
  b.host_io = true;
  return enqueue_download(c, b);
}

int rtuf_filter_batch_async(rtuf_context* c, int n, const float* const* depth_in, float* const* masked_out,
                            uint8_t* const* mask_out)
{
  KIDS_NEXT(c, rtuf_filter_batch_async(k, n, depth_in, masked_out, mask_out));
  if (!c) return RTUF_ERR_INVALID;
  return submit_host_batch(c, n, reinterpret_cast<const void* const*>(depth_in), reinterpret_cast<void* const*>(masked_out),
                           reinterpret_cast<void* const*>(mask_out), false);
}

int rtuf_filter_batch_u16_async(rtuf_context* c, int n, const uint16_t* const* depth_in, uint16_t* const* masked_out,
                                uint8_t* const* mask_out)
{
  KIDS_NEXT(c, rtuf_filter_batch_u16_async(k, n, depth_in, masked_out, mask_out));
  if (!c) return RTUF_ERR_INVALID;
  return submit_host_batch(c, n, reinterpret_cast<const void* const*>(depth_in), reinterpret_cast<void* const*>(masked_out),
                           reinterpret_cast<void* const*>(mask_out), true);
}

int rtuf_filter_batch_bits_async(rtuf_context* c, int n, const float* const* depth_in, uint32_t* const* bits_out)
{
  KIDS_NEXT(c, rtuf_filter_batch_bits_async(k, n, depth_in, bits_out));
  if (!c) return RTUF_ERR_INVALID;
  return submit_host_batch(c, n, reinterpret_cast<const void* const*>(depth_in), nullptr, nullptr, false, bits_out);
}

int rtuf_filter_batch_bits_u16_async(rtuf_context* c, int n, const uint16_t* const* depth_in, uint32_t* const* bits_out)
{
  KIDS_NEXT(c, rtuf_filter_batch_bits_u16_async(k, n, depth_in, bits_out));
  if (!c) return RTUF_ERR_INVALID;
  return submit_host_batch(c, n, reinterpret_cast<const void* const*>(depth_in), nullptr, nullptr, true, bits_out);
}

int rtuf_filter_batch(rtuf_context* c, int n, const float* const* depth_in, float* const* masked_out,
                      uint8_t* const* mask_out)
{
  const int rc = rtuf_filter_batch_async(c, n, depth_in, masked_out, mask_out);
  return rc != RTUF_OK ? rc : rtuf_sync(c);
}

int rtuf_filter_batch_u16(rtuf_context* c, int n, const uint16_t* const* depth_in, uint16_t* const* masked_out,
                          uint8_t* const* mask_out)
{
  const int rc = rtuf_filter_batch_u16_async(c, n, depth_in, masked_out, mask_out);
  return rc != RTUF_OK ? rc : rtuf_sync(c);
}

int rtuf_wait_oldest(rtuf_context* c)
{
  if (!c) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) {
    if (c->order.empty()) return RTUF_OK;
    rtuf_context* k = c->kids[c->order.front()];
    c->order.pop_front();
    const int rc = rtuf_wait_oldest(k);
    if (rc < 0) c->error = k->error;
    return rc;
  }
  hipSetDevice(c->device);
  return c->pending ? retire_oldest(c) : RTUF_OK;
}

int rtuf_host_alloc(rtuf_context* c, size_t bytes, void** out)
{
  KIDS_ONE(c, 0, rtuf_host_alloc(k, bytes, out));
  if (!c || !out || !bytes) return RTUF_ERR_INVALID;
  hipSetDevice(c->device);
  void* p = nullptr;
  HIP_TRY(c, hipHostMalloc(&p, bytes));
  c->pinned.push_back(p);
  *out = p;
  return RTUF_OK;
}

int rtuf_host_free(rtuf_context* c, void* p)
{
  if (!c) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) {
    if (!p) return RTUF_OK;
    const int rc = rtuf_sync(c);           // no transfer of any pipeline may still target the block
    if (rc < 0) return rc;
    KIDS_ONE(c, 0, rtuf_host_free(k, p));
  }
  if (!p) return RTUF_OK;
  hipSetDevice(c->device);
  auto it = std::find(c->pinned.begin(), c->pinned.end(), p);
  if (it == c->pinned.end()) return c->fail(RTUF_ERR_INVALID, "not a block of rtuf_host_alloc");
  const int rc = rtuf_sync(c);          // no transfer may still target the block
  if (rc != RTUF_OK) return rc;
  c->pinned.erase(it);
  HIP_TRY(c, hipHostFree(p));
  return RTUF_OK;
}

int rtuf_filter(rtuf_context* c, const unsigned char* buffer, const double* projection, int width, int height)
{
  if (c && !c->kids.empty()) {
    // the single-stream call runs on the first pipeline; its projection is a setter like any other (all pipelines)
    if (projection) { const int rc_ = rtuf_set_camera(c, 0, projection, nullptr, nullptr); if (rc_ < 0) return rc_; }
    KIDS_ONE(c, 0, rtuf_filter(k, buffer, nullptr, width, height));
  }
  if (!c || !buffer) return RTUF_ERR_INVALID;
  if (width != c->width || height != c->height)
    return c->fail(RTUF_ERR_INVALID, "image size %dx%d differs from the context's %dx%d (the reference re-runs initGL here; create a new context instead)",
                   width, height, c->width, c->height);
  if (projection) { const int rc = rtuf_set_camera(c, 0, projection, nullptr, nullptr); if (rc != RTUF_OK) return rc; }
  const size_t plane = (size_t)width * height;
  c->single_masked.resize(plane);
  c->single_mask.resize(plane);
  const float* in = reinterpret_cast<const float*>(buffer);
  float* mo = c->single_masked.data();
  uint8_t* mk = c->single_mask.data();
  return rtuf_filter_batch(c, 1, &in, &mo, &mk);
}

const float* rtuf_get_masked_depth(const rtuf_context* c) { if (c && !c->kids.empty()) c = c->kids[0]; return (c && !c->single_masked.empty()) ? c->single_masked.data() : nullptr; }
const uint8_t* rtuf_get_mask(const rtuf_context* c) { if (c && !c->kids.empty()) c = c->kids[0]; return (c && !c->single_mask.empty()) ? c->single_mask.data() : nullptr; }

int rtuf_get_stats(rtuf_context* c, rtuf_stats* out)
{
  if (!c || !out) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) {
    // counters and last-batch times of the pipeline that ran last (cover_pass, cover_tiles, work_items ... are that
    // pipeline's: each decides about its cover pass on its own); event sums, regrowths, memory and graph counts over all pipelines
    *out = c->kids[c->last_kid]->stats;
    out->bin_capacity = c->kids[c->last_kid]->capacity;
    out->raster_lanes = (uint32_t)c->kids[c->last_kid]->n_lanes; out->launch_group = (uint32_t)c->kids[c->last_kid]->group;
    out->regrowths = 0; out->timed_batches = 0; out->device_bytes = 0;
    out->graphs_enabled = 1u; out->graph_hits = out->graph_misses = 0; out->lanes_side_by_side = c->lanes_share_queue ? 0u : 1u;
    for (const rtuf_context* k : c->kids) { out->device_bytes += k->device_bytes; out->graphs_enabled &= k->graphs_ok ? 1u : 0u; out->graph_hits += k->graph_hits; out->graph_misses += k->graph_misses; }
    out->sum_ms_pose = out->sum_ms_setup = out->sum_ms_raster = out->sum_ms_compare = out->sum_ms_total = out->sum_ms_clip = 0;
    for (const rtuf_context* k : c->kids) {
      out->regrowths += k->stats.regrowths; out->timed_batches += k->stats.timed_batches;
      out->sum_ms_pose += k->stats.sum_ms_pose; out->sum_ms_setup += k->stats.sum_ms_setup; out->sum_ms_raster += k->stats.sum_ms_raster;
      out->sum_ms_compare += k->stats.sum_ms_compare; out->sum_ms_total += k->stats.sum_ms_total; out->sum_ms_clip += k->stats.sum_ms_clip;
    }
    return RTUF_OK;
  }
  *out = c->stats;
  out->bin_capacity = c->capacity;
  out->device_bytes = c->device_bytes;
  out->raster_lanes = (uint32_t)c->n_lanes; out->launch_group = (uint32_t)c->group;
  out->graphs_enabled = c->graphs_ok ? 1u : 0u; out->graph_hits = c->graph_hits; out->graph_misses = c->graph_misses;
  out->lanes_side_by_side = c->lanes_share_queue ? 0u : 1u;
  return RTUF_OK;
}

#ifdef RTUF_LANECOUNT
// instrumented builds only (scripts/lane_util.sh): lane slots issued / lanes live per instrumented loop, last retired batch
int rtuf_debug_lane_counts(rtuf_context* c, unsigned long long* slots, unsigned long long* live, int n)
{
  if (!c || !slots || !live) return RTUF_ERR_INVALID;
  if (!c->kids.empty()) c = c->kids[c->last_kid];
  for (int i = 0; i < n && i < kLaneLoops; i++) { slots[i] = c->lane_slots[i]; live[i] = c->lane_live[i]; }
  return kLaneLoops;
}
#endif

int rtuf_enable_timing(rtuf_context* c, int on)
{
  KIDS_ALL(c, rtuf_enable_timing(k, on));
  if (!c) return RTUF_ERR_INVALID;
  c->timing = on < 0 ? 0 : (on > 3 ? 1 : on);
  c->timing_seq = 0;
  for (double& v : c->acc_ms) v = 0;
  c->acc_batches = 0;
  return RTUF_OK;
}

int rtuf_debug_read_zsurface(rtuf_context* c, int n, float* host_out)
{
  KIDS_ONE(c, c->last_kid, rtuf_debug_read_zsurface(k, n, host_out));
  if (!c || !host_out) return RTUF_ERR_INVALID;
  const float* zs = c->lane[c->last_lane].d_zsurface;
  if (!(c->params.flags & RTUF_FLAG_TWO_KERNEL) || !zs) return c->fail(RTUF_ERR_STATE, "z-surface exists only in two-kernel mode");
  if (n <= 0 || n > c->group) return c->fail(RTUF_ERR_INVALID, "z-surface holds the last launch group (at most %d streams)", c->group);
  hipSetDevice(c->device);
  sync_lanes(c);
  HIP_TRY(c, hipMemcpy(host_out, zs, (size_t)n * c->width * c->height * sizeof(float), hipMemcpyDeviceToHost));
  return RTUF_OK;
}

}  // extern "C"
