// rtuf_device.h -- device-side data layout shared by the HIP kernels and the host API.
//
// HBM layout (all arrays owned by one rtuf_context, one GPU):
//   static geometry, uploaded once by rtuf_finalize_models()
//     chunks Chunk[C]    <= 256 triangles of ONE draw call (a compact patch: Morton order of the
//                        centroids) + the <= 256 distinct vertices they use + bounding box   64 B / chunk
//     cverts float4[Vc]  object-space positions, grouped per chunk        16 B / vertex
//     ctris  u32[T]      3 x 10-bit chunk-local vertex ids                 4 B / triangle
//                        (draw-order sequence numbers in corder, parallel to ctris; 0 = background)
//     draws  Draw[D]     link id + the glScalef/glTranslatef of the draw
//   per frame, per stream s (slot within the batch)
//     cams   Camera[N]   projection / camera_offset_inv / camera_tf as f64
//     link_tf f64[N][L][16]                                         (two slots: one per batch in flight)
//     mvp    f32[N][D][16]  written by pose_kernel;  bg BgInfo[N];  items WorkItem[] written by cull_kernel (one list per raster lane)
//     depth  f32[N][H][W] in, masked f32[N][H][W] + mask u8[N][H][W] out
//   rasteriser working set of ONE RASTER LANE (a context has one to three: a lane is a HIP stream plus the arrays below; the
//   launch groups of a batch alternate between the lanes, so one group's set-up runs under the other's tile kernel),
//   per stream g of the launch group and screen tile
//     bin_hdr   BinHeader[G][tiles]   records binned from the front (small boxes) and from the back of the bin, and the tile's
//                                     cover (nearest triangle that covers the WHOLE tile): 16 bytes, one scalar load in the tile kernel
//     bins      PackedTri[G][tiles][capacity]  (32 B records: small boxes from the front, larger from the back)
//     fbin_count u32[G][tiles], fbins Frag[G][tiles][fcapacity]    (pixels of small triangles, 8 B; top bit of the count: the
//                                                                    record bin holds a near record, see KeyFmt)
//     clip_list ClipItem[shards][clip_capacity]   triangles that cross a frustum plane (set-up kernel -> clip kernel)
//     big_list  BigRec[shards][big_capacity]      records over more than 4 tiles (set-up / clip kernel -> bigrec_kernel)
//     zsurface  f32[G][H][W]                       two-kernel mode only
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtuf {

// Timing experiments (scripts/ablate_*.sh) skip work and produce WRONG images.  They exist only in a library
// built with -DRTUF_ABLATE (lib/variants/librtuf_ablate.so); the product library contains none of that code and
// rtuf_create / rtuf_set_params reject the flag bits.
#ifdef RTUF_ABLATE
#define RTUF_ABL(flags, bits) (((flags) & (bits)) != 0u)
#else
#define RTUF_ABL(flags, bits) false
#endif

#ifndef RTUF_TILE_W
#define RTUF_TILE_W 64
#endif
#ifndef RTUF_TILE_H
#define RTUF_TILE_H 32
#endif
constexpr int kTileW = RTUF_TILE_W;      // screen tile of one raster workgroup (LDS: 8 B per pixel; 64x32 = 16 KB)
constexpr int kTileH = RTUF_TILE_H;
constexpr int kBlock = 256;
#ifndef RTUF_TILE_THREADS
#define RTUF_TILE_THREADS 256
#endif
constexpr int kTileThreads = RTUF_TILE_THREADS;   // threads of a tile workgroup
#ifndef RTUF_MAX_CHUNK_VERTS
#define RTUF_MAX_CHUNK_VERTS 256
#endif
constexpr int kMaxChunkVerts = RTUF_MAX_CHUNK_VERTS;     // unique vertices per set-up chunk (one per lane; LDS: 24 B each per stream)
#ifndef RTUF_STREAMS_PER_BLOCK
#define RTUF_STREAMS_PER_BLOCK 3
#endif
constexpr int kStreamsPerBlock = RTUF_STREAMS_PER_BLOCK;     // streams a set-up workgroup loops over per chunk

// Bias of the 20-bit snapped coordinates in a PackedTri.  In-frustum vertices snap to [-128, W*256 + 128]; the
// clipper's interpolation is rounded, so its vertices may land a few units outside (seen: -129): 1024 leaves room
// on both sides (2048*256 + 128 + 2*1024 < 2^20).
constexpr int kCoordBias = 1024;

struct alignas(16) TriRec {     // 64 B: one rasterisable triangle inside one tile bin
  int32_t A[3];                 // edge i is inside  <=>  A[i]*px + B[i]*py + C[i] > 0
  int32_t B[3];
  int32_t C[3];
  float a0, dzdx, dzdy;         // z(px,py) = fma(dzdy, py, fma(dzdx, px, a0))
  uint32_t bbx;                 // x0 | x1 << 16   (inclusive pixel bounds)
  uint32_t bby;                 // y0 | y1 << 16
  uint32_t order;               // draw-order sequence number of the source triangle
  uint32_t pad;
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

struct alignas(16) PackedTri {  // 32 B: what a triangle bin stores; the tile kernel rebuilds TriRec from it
  unsigned long long v01;       // (x0+bias) | (y0+bias) << 20 | (x1+bias) << 40   snapped 1/256-px coordinates,
  unsigned long long v12;       // (y1+bias) | (x2+bias) << 20 | (y2+bias) << 40   already oriented (area > 0)
  float a0, dzdx, dzdy;         // z plane
  uint32_t order;               // draw-order key (29 bits) | kNearBit
};
static_assert(sizeof(PackedTri) == 32, "PackedTri must be 32 bytes");
// Set in PackedTri.order when the record may produce a window z <= 0.5 somewhere in its box (geometry within about twice
// the near distance of the camera), where the float z the shader sees is finer than the 24-bit depth: only such records
// are looked at by the tile kernel's exact-z pass.
constexpr uint32_t kNearBit = 1u << 31;
constexpr uint32_t kOrderMask = (1u << 29) - 1u;

// A record whose bounding box touches more than kCoopTiles tiles is not appended to its bins by the lane that made it
// (a wave with a run of wall triangles would append hundreds of (record, tile) pairs one record after the other while
// the rest of the GPU idles): set-up and clip kernel put it on a list, bigrec_kernel gives every list entry a wave.
struct alignas(16) BigRec {
  PackedTri pk;
  uint32_t slot;                // stream slot within the in-flight group
  uint32_t pad[3];
};
static_assert(sizeof(BigRec) == 48, "BigRec is three 16-byte stores");

// 8 B: one covered pixel of a small (<= 4x4 pixel centres, single tile) triangle, ready for the depth
// test:  z24 << 40 | order << kFragPosBits | position in the tile (y * kTileW + x).
// Fragments carry no z plane, so triangles that may win with window z <= 0.5 (where the float z the
// shader sees is finer than 24 bits) are never resolved to fragments; they stay records.
constexpr int kFragPosBits = 11;
constexpr uint32_t kMaxOrder = (1u << (40 - kFragPosBits)) - 1u;    // draw-order keys must fit 29 bits
static_assert(kTileW * kTileH <= (1 << kFragPosBits), "tile positions must fit the fragment's position field");
static_assert(kMaxOrder == kOrderMask, "records and fragments carry the same 29-bit draw-order keys");
struct alignas(8) Frag { unsigned long long v; };
static_assert(sizeof(Frag) == 8, "Frag must be 8 bytes");

// Per-bin header: the two fill counters of the record bin and the tile's cover -- the nearest triangle that covers the whole
// tile (bigrec_kernel<0>): largest 24-bit depth it has there << 32 | its index in big_list; kNoCover = none.
constexpr unsigned long long kNoCover = ~0ull;
struct alignas(16) BinHeader {
  uint32_t count[2];             // records binned from the front (boxes of at most kFrontArea pixel centres) / from the back
  unsigned long long cover;
};
static_assert(sizeof(BinHeader) == 16, "one scalar load, one store");

struct Chunk {                  // <= 256 consecutive triangles of one draw + their vertex list
  uint32_t tri_begin;           // into ctris
  uint32_t tri_count;
  uint32_t vert_begin;          // into cverts
  uint32_t vert_count;          // <= kMaxChunkVerts
  uint32_t draw;
  uint32_t model;
  uint32_t reserved;
  uint32_t pad;
  float center[3];              // object-space bounding box of the chunk's vertices: centre ...
  uint32_t pad1;
  float half[3];                // ... and half extents
  uint32_t pad2;
};

struct Draw {
  uint32_t link;                // global link index (row of link_tf)
  uint32_t pre_op;              // RTUF_OP_*
  float op[3];
  uint32_t model;
  uint32_t pad[2];
};

struct Camera {
  double projection[16];
  double offset_inv[16];
  double cam_tf[16];
  double shift[2];              // camera_tx_, camera_ty_ (src/urdf_filter.cpp:607-611): applied by the forward-kinematics
                                // kernels when they derive cam_tf from a robot frame; a host-supplied cam_tf has them applied already
};

struct alignas(16) ClipItem {   // one triangle that crosses a frustum plane; self-contained, so the clip
  uint32_t slot;                // kernel's loads depend on nothing but the item (stream slot within the group)
  uint32_t draw;
  uint32_t vert_begin;          // of the triangle's chunk, into cverts
  uint32_t packed;              // 3 x 10-bit chunk-local vertex ids
  uint32_t order;               // draw-order key (0 = background quad)
  uint32_t pad[3];
};

// Device-side statistics / overflow detection.  One hot word would serialise every workgroup at
// a single L2 atomic unit (~12 ns per atomic), so everything is sharded over kCounterShards
// cache-line-sized slots that the host sums after the batch.
#ifndef RTUF_COUNTER_SHARDS
#define RTUF_COUNTER_SHARDS 64
#endif
constexpr int kCounterShards = RTUF_COUNTER_SHARDS;
constexpr int kLaneLoops = 48;           // instrumented loops (and histogram buckets) of a -DRTUF_LANECOUNT build
struct alignas(128) CounterShard {
  unsigned long long tris_binned;
  unsigned long long bin_entries;
  unsigned int clip_count;      // entries in this shard's segment of clip_list
  unsigned int max_bin_fill;
  unsigned int clip_overflow;
  unsigned int max_fbin_fill;
  unsigned long long frags;
  unsigned int uncovered;       // mask-bits output only: some pixel was reached by no fragment (see rtuf_filter_batch_bits*)
  unsigned int big_count;       // entries in this shard's segment of big_list (reset per in-flight group)
  unsigned int max_big_fill;    // largest big_count of the batch's groups (overflow detection)
  unsigned int cover_tiles;     // tiles whose initial depth keys came from a whole-cover triangle (statistics)
  unsigned int exact_tiles;     // tiles that ran the exact-z pass (statistics)
  unsigned int zero_items;      // set-up work items (chunk x <= 3 streams) none of whose triangles survived (statistics)
  unsigned long long occluded;  // (record, tile) pairs not appended because they lie behind a whole-cover triangle (statistics)
  unsigned long long raster_atomics;   // RTUF_COUNT builds only: depth tests issued by the tile kernel (LDS atomics)
  unsigned long long drawn_pixels;     // RTUF_COUNT builds only: pixels whose final key is not the background's
  unsigned int pad[10];
#ifdef RTUF_LANECOUNT
  // Lane-utilisation builds (scripts/lane_util.sh; never the product): per instrumented loop the lane slots its trips
  // issued (64 per wave and trip) and the lanes that were live in them, see kLane* in rtuf_kernels.hip.
  unsigned long long lane_slots[kLaneLoops];
  unsigned long long lane_live[kLaneLoops];
#endif
};
#ifndef RTUF_LANECOUNT
static_assert(sizeof(CounterShard) == 128, "CounterShard must be one 128-byte line");
#endif
struct alignas(128) WorkCount {
  unsigned int n_items;         // length of the launch group's work list (cull_kernel)
  unsigned int grid;            // work items the group's set-up launch covered (written by the set-up kernel itself: the batch's
                                // last kernel compares the two on the device, see BatchStatus)
  unsigned int pad[30];
};
struct Counters { CounterShard shard[kCounterShards]; WorkCount work; };

// One set-up workgroup's job: a chunk and up to kStreamsPerBlock stream slots (of the in-flight
// group) whose frustum the chunk's bounding sphere touches; 0xffff = unused.  Written by cull_kernel.
struct alignas(32) WorkItem {
  uint32_t chunk;
  uint32_t tri_begin;           // copies of the chunk's fields: the set-up workgroup needs no Chunk load
  uint32_t vert_begin;
  uint32_t reserved;
  uint32_t draw;
  uint16_t tri_count, vert_count;
  uint16_t slot[4];
};
static_assert(sizeof(WorkItem) == 32, "WorkItem is two 16-byte loads");
// Upper bound of the work items one chunk can produce for a launch group of `group` streams: cull_kernel
// compacts the visible streams per block of kBlock stream slots and rounds EACH block's count up to whole
// items, so a group of more than kBlock streams can need more than ceil(group / kStreamsPerBlock).  Sizes the
// items array and the set-up kernel's worst-case grid.
inline int max_items_per_chunk(int group)
{
  int n = 0;
  for (int first = 0; first < group; first += kBlock) {
    const int in_block = group - first < kBlock ? group - first : kBlock;
    n += (in_block + kStreamsPerBlock - 1) / kStreamsPerBlock;
  }
  return n;
}

// Per stream: what the background quad (src/urdf_filter.cpp:589-597) amounts to, computed once per batch
// by pose_kernel so that the tile kernel's workgroups do no per-stream arithmetic of their own.
struct alignas(16) BgInfo {
  float z;                      // window z of the quad when it is a constant full-screen plane
  uint32_t mode;                // 1 = constant full-screen plane (analytic), 0 = draw it as geometry
  float thr;                    // compare threshold of that depth: sensor > thr <=> filtered
  uint32_t z24;                 // its 24-bit depth-buffer value
};

// Device-resident status word of a batch slot (rtuf_batch_status_device): the low 16 bits count the launch groups of the batch
// that have not finished yet (set by pose_kernel, the first kernel of a batch; every group's publish_counters workgroup, the
// last kernel of its lane, takes one off), the bits above say why the batch's planes are NOT final -- a working buffer was too
// small, and the host will run the batch again when it retires it.  0 <=> every group finished and nothing overflowed.
constexpr uint32_t kStatusPendingMask = 0xffffu;
constexpr uint32_t kStatusBinOverflow = 1u << 16;     // a record or fragment bin was fuller than its capacity
constexpr uint32_t kStatusClipOverflow = 1u << 17;    // the clip list was too short
constexpr uint32_t kStatusBigOverflow = 1u << 18;     // the many-tile list was too short
constexpr uint32_t kStatusGridShort = 1u << 19;       // the set-up grid did not cover the work list
constexpr uint32_t kStatusUncovered = 1u << 20;       // mask-bits output: a pixel no fragment reached (retiring the batch fails)

// kernel argument blocks (passed by value) and host-callable launchers (rtuf_kernels.hip)
struct PoseArgs {
  const Camera* cams;        // [n_streams]
  const double* link_tf;     // [n_streams][n_links][16]
  const Draw* draws;         // [n_draws]
  float* mvp;                // [n_streams][n_draws + 1][16]
  BgInfo* bg;                // [n_streams]
  float sc_num, sc_off, max_diff;   // to_linear_depth constants (host-computed, see shade_consts) and the threshold
  Counters* counters;        // [n_counters]: one block per launch group of the batch, zeroed by this kernel
  int n_counters;
  uint32_t* status;          // the batch slot's status word: set to n_counters (launch groups still to finish) by this kernel
  int n_streams, n_draws, n_links;
  float z_far;
  int width, height;
};

struct SetupArgs {
  const float4* cverts;          // chunk-local vertex lists (object space)
  const uint32_t* ctris;         // 3 x 10-bit chunk-local vertex ids per triangle
  const uint32_t* corder;        // draw-order sequence number per triangle (>= 1; 0 = background quad), parallel to ctris
  const Chunk* chunks;
  const float* mvp;              // [n_streams][n_draws + 1][16]
  const uint64_t* model_mask;    // [n_streams] bit m set: stream renders model m
  const BgInfo* bg;              // [n_streams]
  PackedTri* bins;               // [G][tiles][capacity]   triangles that are not resolved to fragments
  BinHeader* bin_hdr;            // [G][tiles]
  Frag* fbins;                   // [G][tiles][fcapacity]  covered pixels of the small (<= 4x4, single-tile) boxes
  uint32_t* fbin_count;          // [G][tiles]
  uint32_t fcapacity;
  ClipItem* clip_list;
  float4* clip_spill;            // clip kernel: polygon vertices beyond the eight per thread that live in LDS
  BigRec* big_list;              // [kCounterShards][big_capacity]  many-tile records, appended to their bins by bigrec_kernel
  uint32_t big_capacity;         // per shard segment
  WorkItem* items;               // [n_chunks * ceil(group / kStreamsPerBlock)] visible (chunk, streams) jobs of this group
  Counters* counters;
  int n_chunks;
  int group_base;                // first stream slot of this in-flight group
  int group_size;
  int n_draws;
  int width, height, tiles_x, tiles_y;
  uint32_t capacity;
  uint32_t clip_capacity;        // per shard segment
  uint32_t bg_chunk;             // index of the background-quad chunk
  uint32_t flags;                // rtuf_params.flags
};

struct TileArgs {
  const PackedTri* bins;
  BinHeader* bin_hdr;            // [G][tiles], reset to "nothing binned, no cover" by this kernel after use
  const Frag* fbins;
  uint32_t* fbin_count;          // reset to 0 by this kernel after use
  uint32_t fcapacity;
  const float* depth;            // [n][H][W]   (f32 metres) or, with io_u16, uint16 millimetres
  float* masked;                 // [n][H][W]   same element type as depth
  uint8_t* mask;                 // [n][H][W] or nullptr
  uint32_t* bits;                // mask-only output: [n][H][ceil(W/32)] words, pixel x = bit x % 32 of word x / 32; masked / mask unused
  float* zsurface;               // [G][H][W]  (two-kernel mode)
  const BgInfo* bg;              // [n]
  Counters* counters;
  const BigRec* big_list;        // the record a bin's cover entry points to
  int group_base, group_size;
  int width, height, tiles_x, tiles_y;
  uint32_t capacity;
  uint32_t flags;
  float z_near, z_far, max_diff, replace_value;
  float sc_num, sc_off;          // z_near*z_far/(z_near-z_far) and z_far/(z_far-z_near) in float, as the shader computes them
  int io_u16;                    // 16UC1 in/out fused into the kernel (src/urdf_filter.cpp:287-288, :309-312)
  int key_shift;                 // depth keys: draw order << key_shift in the low word, the float z's low bits below it (KeyFmt)
  int fast_div;                  // the threshold's division may run without its scaling / fix-up instructions (host: shade_consts_admit_core)
};

struct CompareArgs {
  const float* depth;     // group base
  const float* zsurface;  // group base
  float* masked;
  uint8_t* mask;          // may be nullptr
  size_t n_pixels;        // multiple of 4 handled vectorised, tail scalar
  float z_near, z_far, max_diff, replace_value;
  float sc_num, sc_off;
  int io_u16;
  int fast_div;
};


struct FkArgs {
  const int32_t* parent;        // [F]
  const int32_t* joint_type;    // [F]
  const int32_t* depth;         // [F] distance from the root (0 for the root)
  const double* joint_origin;   // [F][12]  row-major 3x3 basis + origin
  const double* joint_axis;     // [F][3]
  const int32_t* link_frame;    // [L_model]
  const double* link_offset;    // [L_model][12]
  const double* q;              // [N][F]
  const double* root_tf;        // [N][12] or nullptr
  const uint8_t* enabled;       // [N] stream uses FK for this model
  double* link_tf;              // [N][L_total][16]
  Camera* cams;                 // [N]
  int n_streams, n_frames, n_links_model, link_base, n_links_total, camera_frame, max_depth;
};
void launch_fk(const FkArgs& a, hipStream_t st);
void launch_pose(const PoseArgs& a, hipStream_t st);
void launch_cull(const SetupArgs& a, hipStream_t st);
uint32_t launch_setup(const SetupArgs& a, uint32_t items_hint, bool sweep, hipStream_t st);    // returns the main grid size
void launch_clip(const SetupArgs& a, hipStream_t st);
size_t clip_spill_bytes(uint32_t clip_capacity);
void launch_bigrec(const SetupArgs& a, bool cover_pass, hipStream_t st);      // cover_pass: bigrec_kernel<0> runs first (and launch_tile gets the same flag)
void launch_init_headers(BinHeader* hdr, size_t n_bins, hipStream_t st);
// copies the counter blocks first, first + stride, ... (count of them) of a batch to the same places of the pinned host array
// ... and folds what they say about overflows into the batch's status word (PublishLimits: the capacities they are held against)
struct PublishLimits { uint32_t capacity, fcapacity, big_capacity; };
void launch_publish_counters(const Counters* src, Counters* host_dst, int first, int stride, int count, uint32_t* status, PublishLimits lim, hipStream_t st);
void launch_tile(const TileArgs& a, bool two_kernel, bool cover_pass, hipStream_t st);   // a.io_u16 selects the 16UC1 variant
void launch_compare(const CompareArgs& a, hipStream_t st);
void launch_spin(unsigned long long ticks, hipStream_t st);      // a one-wave kernel that idles for `ticks` of the 100 MHz clock

}  // namespace rtuf
