// rtuf_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the depth self-filter.
//
// Pipeline per batch of camera streams (pose stage on a side stream, raster stage on the main stream):
//   fk_tree_kernel  joint positions -> link matrices + camera transform   replaces urdf_renderer.cpp:173-190 (TF lookups)
//   pose_kernel     float32 OpenGL matrix stack per (stream, draw);       replaces urdf_filter.cpp:576-614,
//                   per-stream background plane; zeroes the counters      renderable.cpp:59-68, :95, :128, :427
//   cull_kernel     chunk bounding boxes vs every stream's frustum ->     (none: GL draws everything)
//                   compact work list of (chunk, 3 streams) items
//   setup_kernel    per item: vertex transform once per chunk vertex,     replaces urdf_filter.vert + the GL
//                   clip test, viewport, 1/256-px snap, sub-pixel cull,   driver's primitive assembly / set-up
//                   z plane; small boxes -> 8-B fragments, others ->
//                   32-B records, binned per 64x32 tile
//   clip_kernel     the few triangles that cross a frustum plane:         (GL driver clipper)
//                   Sutherland-Hodgman in clip space, fan, same set-up
//   bigrec_kernel   one wave per record that touches more than 4 tiles:   (binning: no counterpart in the reference)
//                   appends it to the bins of the tiles it really touches
//   tile_kernel     one workgroup per (stream, 64x32 tile): the tile's    replaces GL rasterisation + 24-bit
//                   depth keys live in LDS, fragments resolve with        GL_LESS depth test + urdf_filter.frag
//                   64-bit LDS atomicMin, then the per-pixel compare      + glGetTexImage conversions
//                   is done in place (fused) or the z-surface is written
//   compare_kernel  two-kernel mode only: z-surface + sensor -> outputs   replaces urdf_filter.frag:19-36
//
// Exactness: results must equal the reference's GLSL running on Mesa llvmpipe bit for bit
// (see oracle/rtuf_oracle.c for what that pins).  Every float operation whose rounding matters
// is an explicit __f*_rn intrinsic, and the file is built with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rtuf_device.h"

namespace rtuf {

// Lane-utilisation builds (-DRTUF_LANECOUNT, scripts/lane_util.sh; never the product): every instrumented loop counts, per
// trip of a wave, 64 lane slots and the lanes for which `pred` holds (of those the hardware has active at that point) in two
// LDS words per workgroup, summed into the batch's counters when the workgroup ends.  In the product the macros are empty.
enum {
  kLaneTileLoad = 0,      // tile kernel: a wave's load + unpack of up to 64 bin records (live: lanes that hold one)
  kLaneWalkTrip,          // lane-per-triangle walk: quad trips (live: lanes whose box still has quads)
  kLaneWalkFrag,          // ... depth-test bodies the walk executed (live: lanes whose candidate is covered)
  kLaneQuarterTrip,       // quarter-wave walk: pair trips
  kLaneQuarterFrag,
  kLaneWaveTrip,          // whole-wave walk of one record (huge boxes that stay with their wave)
  kLaneWaveFrag,
  kLaneParkTrip,          // workgroup-cooperative walks of the parked records
  kLaneParkFrag,
  kLaneFragList,          // fragment list: one depth test per 8-byte fragment
  kLaneResolve,           // resolve passes (4 pixels per lane)
  kLaneSetupVert,         // set-up kernel: phase 1, a vertex per lane
  kLaneSetupTri,          // phase 2, a triangle per lane
  kLaneSetupSmall,        // phase 3a/b trips: coverage of the <= 4x4 boxes (live: lanes with a triangle)
  kLaneSetupSmallHit,     // ... of which cover a pixel centre (z plane + emission)
  kLaneSetupFragStore,    // fragment emission: per-lane store loop (trips until the lane with most covered pixels is done)
  kLaneSetupFragGroup,    // fragment emission: group-finding trips (one per distinct bin of the wave)
  kLaneSetupRec,          // phase 3c trips: set-up of the records (live: lanes with a triangle)
  kLaneSetupRecHit,       // ... of which survive orientation / bounding
  kLaneSetupRecTile,      // record emission: trips over the tiles a record touches
  kLaneSetupRecGroup,     // record emission: group-finding trips
  kLaneCount,
  // histogram of the tile kernel's bin records (live = records): by quad trips of the lane-per-triangle walk (1, 2, 3, 4, 5, 6,
  // 7-8, 9-12, 13-16, 17-24, 25+), quarter-wave class, larger, nothing in this tile; and how many records have a WHOLE box of
  // at most 8 x 8 / 8 x 4 (or 4 x 8) pixel centres inside one tile (what a larger fragment class of the set-up kernel would take)
  kLaneHist = 24, kLaneHistQuarter = kLaneHist + 11, kLaneHistLarge, kLaneHistNone, kLaneHist8x8, kLaneHist8x4, kLaneHistEnd
};
static_assert(kLaneHistEnd <= kLaneLoops, "CounterShard holds kLaneLoops pairs");
static_assert(kLaneCount <= kLaneLoops, "CounterShard holds kLaneLoops pairs");
#ifdef RTUF_LANECOUNT
__device__ __forceinline__ uint32_t* lane_words() { __shared__ uint32_t s_lw[2 * kLaneLoops]; return s_lw; }
#define RTUF_LANES(id, pred)                                                                                          \
  do {                                                                                                                \
    const unsigned long long act_ = __ballot(true), m_ = __ballot(pred);                                              \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)act_) - 1) {                                                    \
      atomicAdd(&lane_words()[2 * (id)], 64u);                                                                        \
      atomicAdd(&lane_words()[2 * (id) + 1], (uint32_t)__popcll(m_));                                                 \
    }                                                                                                                 \
  } while (0)
#define RTUF_LANES_INIT()                                                                                             \
  do { for (int i_ = threadIdx.x; i_ < 2 * kLaneLoops; i_ += blockDim.x) lane_words()[i_] = 0u; __syncthreads(); } while (0)
#define RTUF_LANES_FLUSH(shard)                                                                                       \
  do {                                                                                                                \
    __syncthreads();                                                                                                  \
    if ((int)threadIdx.x < kLaneLoops && (lane_words()[2 * threadIdx.x] | lane_words()[2 * threadIdx.x + 1])) {        \
      atomicAdd(&(shard).lane_slots[threadIdx.x], (unsigned long long)lane_words()[2 * threadIdx.x]);                 \
      atomicAdd(&(shard).lane_live[threadIdx.x], (unsigned long long)lane_words()[2 * threadIdx.x + 1]);              \
    }                                                                                                                 \
  } while (0)
#else
#define RTUF_LANES(id, pred) ((void)0)
#define RTUF_LANES_INIT() ((void)0)
#define RTUF_LANES_FLUSH(shard) ((void)0)
#endif

// ---------------------------------------------------------------------------------------
// float32 matrix stack (GL semantics: every glMultMatrixd rounds its argument to float and
// multiplies in float32, products accumulated left to right, no fused multiply-add)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void matmul4(float* __restrict__ out, const float* a, const float* b)
{
  float p[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float ai0 = a[i], ai1 = a[4 + i], ai2 = a[8 + i], ai3 = a[12 + i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float s = __fmul_rn(ai0, b[4 * j]);
      s = __fadd_rn(s, __fmul_rn(ai1, b[4 * j + 1]));
      s = __fadd_rn(s, __fmul_rn(ai2, b[4 * j + 2]));
      s = __fadd_rn(s, __fmul_rn(ai3, b[4 * j + 3]));
      p[4 * j + i] = s;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) out[k] = p[k];
}

__device__ __forceinline__ void mult_d(float* top, const double* m)
{
  float f[16];
#pragma unroll
  for (int k = 0; k < 16; k++) f[k] = (float)m[k];
  matmul4(top, top, f);
}

__device__ __forceinline__ void vs_position(const float* __restrict__ M, float x, float y, float z, float* c)
{
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float s = __fmul_rn(M[r], x);
    s = __fadd_rn(s, __fmul_rn(M[4 + r], y));
    s = __fadd_rn(s, __fmul_rn(M[8 + r], z));
    s = __fadd_rn(s, M[12 + r]);
    c[r] = s;
  }
}

__device__ __forceinline__ unsigned clipmask_of(const float* c)
{
  unsigned m = 0;
  if (c[0] > c[3]) m |= 1u;
  if (0.0f > __fadd_rn(c[0], c[3])) m |= 2u;
  if (c[1] > c[3]) m |= 4u;
  if (0.0f > __fadd_rn(c[1], c[3])) m |= 8u;
  if (0.0f > __fadd_rn(c[2], c[3])) m |= 16u;
  if (c[2] > c[3]) m |= 32u;
  return m;
}

struct Win { float x, y, z; };

// viewport transform of a shaded vertex (fused multiply-add form)
__device__ __forceinline__ Win viewport_vs(const float* c, float sx, float sy)
{
  const float rhw = __fdiv_rn(1.0f, c[3]);
  Win w;
  w.x = __fmaf_rn(__fmul_rn(c[0], rhw), sx, sx);
  w.y = __fmaf_rn(__fmul_rn(c[1], rhw), sy, sy);
  w.z = __fmaf_rn(__fmul_rn(c[2], rhw), 0.5f, 0.5f);
  return w;
}

// viewport transform of a vertex made by the clipper (separate multiply and add)
__device__ __forceinline__ Win viewport_clip(const float* c, float sx, float sy)
{
  const float oow = __fdiv_rn(1.0f, c[3]);
  Win w;
  w.x = __fadd_rn(__fmul_rn(__fmul_rn(c[0], oow), sx), sx);
  w.y = __fadd_rn(__fmul_rn(__fmul_rn(c[1], oow), sy), sy);
  w.z = __fadd_rn(__fmul_rn(__fmul_rn(c[2], oow), 0.5f), 0.5f);
  return w;
}

// ---------------------------------------------------------------------------------------
// pose_kernel: one thread per (stream, draw).  Draw index == n_draws is the background quad.
// ---------------------------------------------------------------------------------------

__global__ void pose_kernel(PoseArgs a)
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = a.n_draws + 1;
  // first kernel of a batch that every configuration runs: it also zeroes the batch's counters
  // (statistics, clip lists, work-list length), which saves a separate fill launch per batch
  // (one block of counters per launch group of the batch)
  for (int cb = blockIdx.x; cb < a.n_counters; cb += gridDim.x) {
    uint32_t* w = reinterpret_cast<uint32_t*>(a.counters + cb);
    for (int i = threadIdx.x; i < (int)(sizeof(Counters) / 4); i += blockDim.x) w[i] = 0u;
  }
  // ... and says so in the batch slot's status word: this many launch groups have not finished (see kStatus* in rtuf_device.h)
  if (gid == 0) *a.status = (uint32_t)a.n_counters;
  if (gid >= a.n_streams * per) return;
  const int s = gid / per, d = gid - s * per;
  const Camera& cam = a.cams[s];

  float proj[16], mv[16];
#pragma unroll
  for (int k = 0; k < 16; k++) proj[k] = mv[k] = (k % 5 == 0) ? 1.0f : 0.0f;
  mult_d(proj, cam.projection);
  {
    // gluLookAt(0,0,0, 0,0,1, 0,1,0) == diag(-1,1,-1,1), then glTranslated(-0,-0,-0)
    const float la[16] = {-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
    matmul4(mv, mv, la);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float t = __fmul_rn(mv[r], -0.0f);
      t = __fadd_rn(t, __fmul_rn(mv[4 + r], -0.0f));
      t = __fadd_rn(t, __fmul_rn(mv[8 + r], -0.0f));
      mv[12 + r] = __fadd_rn(t, mv[12 + r]);
    }
  }
  float* out = a.mvp + ((size_t)s * per + d) * 16;
  if (d == a.n_draws) {
    // background quad (urdf_filter.cpp:591-596): drawn before the camera transforms
    float m[16];
    matmul4(m, proj, mv);
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = m[k];
    const float zq = (float)((double)a.z_far * 0.99);
    const float qx[4] = {-100.0f, 100.0f, 100.0f, -100.0f};
    const float qy[4] = {-100.0f, -100.0f, 100.0f, 100.0f};
    float c0[4];
    bool constant = true, covers = true;
    unsigned andm = ~0u;
    for (int i = 0; i < 4; i++) {
      float c[4];
      vs_position(m, qx[i], qy[i], zq, c);
      if (i == 0) { c0[0] = c[0]; c0[1] = c[1]; c0[2] = c[2]; c0[3] = c[3]; }
      if (c[2] != c0[2] || c[3] != c0[3]) constant = false;
      const unsigned cm = clipmask_of(c);
      if (cm & 48u) constant = false;               // must lie strictly inside near/far
      if (!(c[3] > 0.0f)) constant = false;
      // corner i must be outside the frustum sideways by a wide margin in both x and y
      if (!(fabsf(c[0]) > 2.0f * c[3] && fabsf(c[1]) > 2.0f * c[3])) covers = false;
      andm &= cm;
    }
    // the four corners must lie in four different xy quadrants (quad surrounds the frustum axis)
    float cA[4], cB[4], cC[4];
    vs_position(m, qx[1], qy[1], zq, cA);
    vs_position(m, qx[2], qy[2], zq, cB);
    vs_position(m, qx[3], qy[3], zq, cC);
    const bool quadrants = (c0[0] < 0) != (cA[0] < 0) && (cA[1] < 0) != (cB[1] < 0) &&
                           (cB[0] < 0) != (cC[0] < 0) && (cC[1] < 0) != (c0[1] < 0);
    const bool ok = constant && covers && quadrants && andm == 0;
    // constant plane: every clipped vertex keeps z_clip and w, so its window z is (z/w)*0.5+0.5
    const float zw = __fadd_rn(__fmul_rn(__fmul_rn(c0[2], __fdiv_rn(1.0f, c0[3])), 0.5f), 0.5f);
    BgInfo bi;
    bi.z = zw;
    bi.mode = ok ? 1u : 0u;
    bi.thr = __fsub_rn(__fdiv_rn(a.sc_num, __fsub_rn(zw, a.sc_off)), a.max_diff);                  // shade_threshold(zw)
    bi.z24 = (uint32_t)__float2int_rn(__fmul_rn(fminf(fmaxf(zw, 0.0f), 1.0f), 16777215.0f));       // z24_of(zw)
    a.bg[s] = bi;
    return;
  }
  const Draw dr = a.draws[d];
  mult_d(mv, cam.offset_inv);
  mult_d(mv, cam.cam_tf);
  mult_d(mv, a.link_tf + ((size_t)s * a.n_links + dr.link) * 16);
  if (dr.pre_op == 1) {            // glScalef
#pragma unroll
    for (int r = 0; r < 4; r++) {
      mv[r] = __fmul_rn(mv[r], dr.op[0]);
      mv[4 + r] = __fmul_rn(mv[4 + r], dr.op[1]);
      mv[8 + r] = __fmul_rn(mv[8 + r], dr.op[2]);
    }
  } else if (dr.pre_op == 2) {     // glTranslatef
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float t = __fmul_rn(mv[r], dr.op[0]);
      t = __fadd_rn(t, __fmul_rn(mv[4 + r], dr.op[1]));
      t = __fadd_rn(t, __fmul_rn(mv[8 + r], dr.op[2]));
      mv[12 + r] = __fadd_rn(t, mv[12 + r]);
    }
  }
  float m[16];
  matmul4(m, proj, mv);
#pragma unroll
  for (int k = 0; k < 16; k++) out[k] = m[k];
}

// ---------------------------------------------------------------------------------------
// fk_kernel: forward kinematics in double precision, one thread per (stream, frame).  The chain
// root -> frame is multiplied top-down, T_child = (T_parent * origin) * motion(q), the same
// association order as the host-side forward kinematics.
// ---------------------------------------------------------------------------------------
struct Tf12 { double m[9]; double o[3]; };

__device__ __forceinline__ Tf12 tf_mul(const Tf12& a, const Tf12& b)
{
  Tf12 r;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++)
      r.m[3 * i + j] = __dadd_rn(__dadd_rn(__dmul_rn(a.m[3 * i], b.m[j]), __dmul_rn(a.m[3 * i + 1], b.m[3 + j])), __dmul_rn(a.m[3 * i + 2], b.m[6 + j]));
    r.o[i] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(a.m[3 * i], b.o[0]), __dmul_rn(a.m[3 * i + 1], b.o[1])), __dmul_rn(a.m[3 * i + 2], b.o[2])), a.o[i]);
  }
  return r;
}

__device__ __forceinline__ Tf12 tf_load(const double* p)
{
  Tf12 t;
#pragma unroll
  for (int k = 0; k < 9; k++) t.m[k] = p[k];
  t.o[0] = p[9]; t.o[1] = p[10]; t.o[2] = p[11];
  return t;
}

__device__ __forceinline__ Tf12 tf_from_quat(double x, double y, double z, double w)
{
  // Matrix3x3::setRotation
  Tf12 t;
  const double d = x * x + y * y + z * z + w * w;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs;
  const double xx = x * xs, xy = x * ys, xz = x * zs;
  const double yy = y * ys, yz = y * zs, zz = z * zs;
  t.m[0] = 1.0 - (yy + zz); t.m[1] = xy - wz; t.m[2] = xz + wy;
  t.m[3] = xy + wz; t.m[4] = 1.0 - (xx + zz); t.m[5] = yz - wx;
  t.m[6] = xz - wy; t.m[7] = yz + wx; t.m[8] = 1.0 - (xx + yy);
  t.o[0] = t.o[1] = t.o[2] = 0.0;
  return t;
}

__device__ __forceinline__ void tf_store_gl(const Tf12& t, double* g)
{
  g[0] = t.m[0]; g[1] = t.m[3]; g[2] = t.m[6]; g[3] = 0.0;
  g[4] = t.m[1]; g[5] = t.m[4]; g[6] = t.m[7]; g[7] = 0.0;
  g[8] = t.m[2]; g[9] = t.m[5]; g[10] = t.m[8]; g[11] = 0.0;
  g[12] = t.o[0]; g[13] = t.o[1]; g[14] = t.o[2]; g[15] = 1.0;
}

// src/urdf_filter.cpp:607-611: the camera origin moves by camera_tx_ along the camera transform's x axis
// ("right" = rotation * (1,0,0)), then by camera_ty_ along its y axis ("down")
__device__ __forceinline__ void apply_camera_shift(Tf12& t, const double* shift)
{
  const double tx = shift[0], ty = shift[1];
#pragma unroll
  for (int i = 0; i < 3; i++) t.o[i] = __dadd_rn(t.o[i], __dmul_rn(t.m[3 * i], tx));
#pragma unroll
  for (int i = 0; i < 3; i++) t.o[i] = __dadd_rn(t.o[i], __dmul_rn(t.m[3 * i + 1], ty));
}

__device__ Tf12 fk_frame(const FkArgs& a, int s, int frame)
{
  int chain[64];
  int n = 0;
  for (int f = frame; f >= 0 && n < 64; f = a.parent[f]) chain[n++] = f;
  Tf12 t;
  if (a.root_tf) t = tf_load(a.root_tf + (size_t)s * 12);
  else { for (int k = 0; k < 9; k++) t.m[k] = (k % 4 == 0) ? 1.0 : 0.0; t.o[0] = t.o[1] = t.o[2] = 0.0; }
  for (int k = n - 1; k >= 0; k--) {
    const int f = chain[k];
    if (a.parent[f] < 0) continue;                  // the root frame carries no joint
    t = tf_mul(t, tf_load(a.joint_origin + (size_t)f * 12));
    const int jt = a.joint_type[f];
    if (jt == 1) {
      const double q = a.q[(size_t)s * a.n_frames + f];
      const double ax = a.joint_axis[3 * f], ay = a.joint_axis[3 * f + 1], az = a.joint_axis[3 * f + 2];
      const double nn = sqrt(ax * ax + ay * ay + az * az);
      const double ux = nn > 0 ? ax / nn : 1.0, uy = nn > 0 ? ay / nn : 0.0, uz = nn > 0 ? az / nn : 0.0;
      const double sh = sin(0.5 * q), ch = cos(0.5 * q);
      t = tf_mul(t, tf_from_quat(ux * sh, uy * sh, uz * sh, ch));
    } else if (jt == 2) {
      const double q = a.q[(size_t)s * a.n_frames + f];
      Tf12 m;
      for (int k2 = 0; k2 < 9; k2++) m.m[k2] = (k2 % 4 == 0) ? 1.0 : 0.0;
      m.o[0] = a.joint_axis[3 * f] * q; m.o[1] = a.joint_axis[3 * f + 1] * q; m.o[2] = a.joint_axis[3 * f + 2] * q;
      t = tf_mul(t, m);
    }
  }
  return t;
}

__global__ void fk_kernel(FkArgs a)
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = a.n_links_model + 1;
  if (gid >= a.n_streams * per) return;
  const int s = gid / per, l = gid - s * per;
  if (!a.enabled[s]) return;
  if (l < a.n_links_model) {
    // link matrix = (fixed<-frame) * link_offset   (Renderable::applyTransform)
    const Tf12 t = tf_mul(fk_frame(a, s, a.link_frame[l]), tf_load(a.link_offset + (size_t)l * 12));
    tf_store_gl(t, a.link_tf + ((size_t)s * a.n_links_total + a.link_base + l) * 16);
  } else if (a.camera_frame >= 0) {
    // camera_transform = lookupTransform(cam_frame, fixed_frame) = inverse(fixed<-camera)
    const Tf12 t = fk_frame(a, s, a.camera_frame);
    Tf12 inv;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv.m[3 * i + j] = t.m[3 * j + i];
    for (int i = 0; i < 3; i++) inv.o[i] = inv.m[3 * i] * (-t.o[0]) + inv.m[3 * i + 1] * (-t.o[1]) + inv.m[3 * i + 2] * (-t.o[2]);
    apply_camera_shift(inv, a.cams[s].shift);
    tf_store_gl(inv, a.cams[s].cam_tf);
  }
}

// Same arithmetic, organised as a tree sweep: one workgroup per stream keeps every frame's
// fixed<-frame transform in LDS and fills it level by level (all frames of one depth in parallel,
// T_child = (T_parent * origin) * motion(q)), then the links and the camera read it.  Used when the
// tree fits in LDS (kFkMaxFrames frames); fk_kernel above is the general fallback.
constexpr int kFkMaxFrames = 256;
__global__ __launch_bounds__(kFkMaxFrames) void fk_tree_kernel(FkArgs a)
{
  __shared__ double s_t[kFkMaxFrames][12];
  const int s = blockIdx.x;
  if (!a.enabled[s]) return;
  const int f = threadIdx.x;                 // one frame per lane; everything it needs is fetched up front
  const bool have = f < a.n_frames;
  int parent = -1, depth = -1;
  Tf12 origin, motion;
  bool has_motion = false;
  if (have) {
    parent = a.parent[f];
    depth = a.depth[f];
    if (depth > 0) {
      origin = tf_load(a.joint_origin + (size_t)f * 12);
      const int jt = a.joint_type[f];
      if (jt == 1) {
        const double q = a.q[(size_t)s * a.n_frames + f];
        const double ax = a.joint_axis[3 * f], ay = a.joint_axis[3 * f + 1], az = a.joint_axis[3 * f + 2];
        const double nn = sqrt(ax * ax + ay * ay + az * az);
        const double ux = nn > 0 ? ax / nn : 1.0, uy = nn > 0 ? ay / nn : 0.0, uz = nn > 0 ? az / nn : 0.0;
        const double sh = sin(0.5 * q), ch = cos(0.5 * q);
        motion = tf_from_quat(ux * sh, uy * sh, uz * sh, ch);
        has_motion = true;
      } else if (jt == 2) {
        const double q = a.q[(size_t)s * a.n_frames + f];
        for (int k2 = 0; k2 < 9; k2++) motion.m[k2] = (k2 % 4 == 0) ? 1.0 : 0.0;
        motion.o[0] = a.joint_axis[3 * f] * q; motion.o[1] = a.joint_axis[3 * f + 1] * q; motion.o[2] = a.joint_axis[3 * f + 2] * q;
        has_motion = true;
      }
    }
  }
  for (int d = 0; d <= a.max_depth; d++) {
    if (have && depth == d) {
      Tf12 t;
      if (d == 0) {
        if (a.root_tf) t = tf_load(a.root_tf + (size_t)s * 12);
        else { for (int k = 0; k < 9; k++) t.m[k] = (k % 4 == 0) ? 1.0 : 0.0; t.o[0] = t.o[1] = t.o[2] = 0.0; }
      } else {
        t = tf_mul(tf_load(s_t[parent]), origin);
        if (has_motion) t = tf_mul(t, motion);
      }
#pragma unroll
      for (int k = 0; k < 9; k++) s_t[f][k] = t.m[k];
      s_t[f][9] = t.o[0]; s_t[f][10] = t.o[1]; s_t[f][11] = t.o[2];
    }
    __syncthreads();
  }
  for (int l = threadIdx.x; l < a.n_links_model; l += blockDim.x) {
    const Tf12 t = tf_mul(tf_load(s_t[a.link_frame[l]]), tf_load(a.link_offset + (size_t)l * 12));
    tf_store_gl(t, a.link_tf + ((size_t)s * a.n_links_total + a.link_base + l) * 16);
  }
  if (threadIdx.x == 0 && a.camera_frame >= 0) {
    const Tf12 t = tf_load(s_t[a.camera_frame]);
    Tf12 inv;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) inv.m[3 * i + j] = t.m[3 * j + i];
    for (int i = 0; i < 3; i++) inv.o[i] = inv.m[3 * i] * (-t.o[0]) + inv.m[3 * i + 1] * (-t.o[1]) + inv.m[3 * i + 2] * (-t.o[2]);
    apply_camera_shift(inv, a.cams[s].shift);
    tf_store_gl(inv, a.cams[s].cam_tf);
  }
}

// ---------------------------------------------------------------------------------------
// triangle set-up
// ---------------------------------------------------------------------------------------
// (issue classes: see depth_test below)
#ifndef RTUF_FAST_CLASS
#define RTUF_FAST_CLASS 1
#endif
#ifndef RTUF_FAST_COVER
#define RTUF_FAST_COVER RTUF_FAST_CLASS        // (A/B switch: cover-only tiles compare 24-bit depths instead of composing keys)
#endif
#ifndef RTUF_FAST_RESOLVE
#define RTUF_FAST_RESOLVE RTUF_FAST_CLASS      // (A/B switch of the second batch: no exact-z look in tiles without near geometry, z of a winner by one add)
#endif
#ifndef RTUF_SMALL_FRAGS
#define RTUF_SMALL_FRAGS 1      // resolve <= 4x4 single-tile boxes to fragments in the set-up kernel
#endif

__device__ __forceinline__ int snap(float v)
{
  return __float2int_rn(__fmul_rn(__fsub_rn(v, 0.5f), 256.0f));
}

// Integer edge functions + bounding box of an ORIENTED (area > 0) snapped triangle.  Used by the
// set-up kernel (fragment path) and by the tile kernel when it unpacks a 32-byte bin record.
__device__ __forceinline__ void edges_from_snapped(int x0, int y0, int x1, int y1, int x2, int y2, int width, int height,
                                                   TriRec& r)
{
  const int minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
  const int miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  const int bx0 = max((minx + 255) >> 8, 0), bx1 = min((maxx - 1) >> 8, width - 1);
  const int by0 = max((miny + 255) >> 8, 0), by1 = min((maxy - 1) >> 8, height - 1);
  const int xs[3] = {x0, x1, x2}, ys[3] = {y0, y1, y2};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3;
    const int dcdx = ys[i] - ys[j];
    const int dcdy = xs[i] - xs[j];
    r.A[i] = -dcdx;
    r.B[i] = dcdy;
#if RTUF_FAST_CLASS
    // The same C without 64-bit arithmetic (two 24-bit multiplies with their high halves, four carry operations, a 64-bit
    // shift and a branch for the bias, all in the 4-cycle class): with x = 256 X + xf, y = 256 Y + yf
    //   c = 256 (dcdx X - dcdy Y) + t,   t = dcdx xf - dcdy yf + bias   (|t| < 2^29: exact in 32 bits)
    //   ceil(c / 256) = (dcdx X - dcdy Y) + ceil(t / 256)
    // and only C's low 32 bits were ever used (the edge value at a pixel of the tile is small; A px + B py + C is evaluated
    // modulo 2^32).  bias = dcdx < 0 || (dcdx == 0 && dcdy > 0)  <=>  2 dcdx - (dcdy > 0) < 0, as shifts and subtractions.
    // (400 M random vertex pairs, incl. coincident and nearly coincident ones, against the 64-bit form on the CPU: identical.)
    {
      const int X = xs[i] >> 8, Y = ys[i] >> 8, xf = xs[i] & 255, yf = ys[i] & 255;
      const uint32_t bias = ((uint32_t)(dcdx + dcdx) - ((uint32_t)(0 - dcdy) >> 31)) >> 31;
      const int t = __mul24(dcdx, xf) - __mul24(dcdy, yf) + (int)bias;
      r.C[i] = (int)((uint32_t)__mul24(dcdx, X) - (uint32_t)__mul24(dcdy, Y) + (uint32_t)(-((-t) >> 8)));
    }
#else
    long long c = (long long)dcdx * xs[i] - (long long)dcdy * ys[i];
    if (dcdx < 0 || (dcdx == 0 && dcdy > 0)) c += 1;   // inclusive on low-x / low-row edges
    // inside <=> c - dcdx*256*px + dcdy*256*py > 0  <=>  ceil(c/256) - dcdx*px + dcdy*py > 0
    r.C[i] = (int)(-((-c) >> 8));
#endif
  }
  r.bbx = (uint32_t)bx0 | ((uint32_t)bx1 << 16);
  r.bby = (uint32_t)by0 | ((uint32_t)by1 << 16);
}

__device__ __forceinline__ int sext21(int v) { return (v << 11) >> 11; }

// Snap, sub-pixel cull, orientation.  Returns false when the triangle covers no pixel centre or is
// degenerate; on success the snapped coordinates are oriented (area > 0, v0/v1 swapped if needed)
// and bx/by hold the inclusive pixel bounding box.
__device__ __forceinline__ bool orient_and_bound(Win& v0, Win& v1, const Win& v2, int width, int height,
                                                 int& x0, int& y0, int& x1, int& y1, int& x2, int& y2,
                                                 int& bx0, int& bx1, int& by0, int& by1)
{
  // vertices of unclipped triangles lie inside the frustum: snapped values are in [-128, 2048*256+128]
  // (clipper-made vertices, a few units more, never come here).
  // The (value-preserving) 21-bit sign extension tells the compiler so, which turns the 64-bit
  // products below into full-rate 24-bit multiplies.
  // x0..y2 come in as the snapped coordinates of phase 1 (s_snap): snap(v.x), snap(v.y)
  x0 = sext21(x0); y0 = sext21(y0);
  x1 = sext21(x1); y1 = sext21(y1);
  x2 = sext21(x2); y2 = sext21(y2);
  const int minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
  const int miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  bx0 = max((minx + 255) >> 8, 0); bx1 = min((maxx - 1) >> 8, width - 1);
  by0 = max((miny + 255) >> 8, 0); by1 = min((maxy - 1) >> 8, height - 1);
  if (bx1 < bx0 || by1 < by0) return false;
  const long long area = (long long)(x0 - x1) * (y2 - y0) - (long long)(x2 - x0) * (y0 - y1);
  if (area == 0) return false;
  if (area < 0) {   // orient: swap vertices 0 and 1 (fixed and float)
    int t = x0; x0 = x1; x1 = t;
    t = y0; y0 = y1; y1 = t;
    Win tw = v0; v0 = v1; v1 = tw;
  }
  return true;
}

// z plane from the unsnapped float vertices (lp_state_setup.c order)
__device__ __forceinline__ void z_plane(const Win& v0, const Win& v1, const Win& v2, float& a0, float& dzdx, float& dzdy)
{
  const float x0c = __fsub_rn(v0.x, 0.5f), y0c = __fsub_rn(v0.y, 0.5f);
  const float dx01 = __fsub_rn(v0.x, v1.x), dy01 = __fsub_rn(v0.y, v1.y);
  const float dx20 = __fsub_rn(v2.x, v0.x), dy20 = __fsub_rn(v2.y, v0.y);
  const float ooa = __fdiv_rn(1.0f, __fsub_rn(__fmul_rn(dx01, dy20), __fmul_rn(dy01, dx20)));
  const float dy20o = __fmul_rn(dy20, ooa), dy01o = __fmul_rn(dy01, ooa);
  const float dx20o = __fmul_rn(dx20, ooa), dx01o = __fmul_rn(dx01, ooa);
  const float da01 = __fsub_rn(v0.z, v1.z), da20 = __fsub_rn(v2.z, v0.z);
  dzdx = __fsub_rn(__fmul_rn(da01, dy20o), __fmul_rn(da20, dy01o));
  dzdy = __fsub_rn(__fmul_rn(da20, dx01o), __fmul_rn(da01, dx20o));
  a0 = __fsub_rn(v0.z, __fadd_rn(__fmul_rn(dzdx, x0c), __fmul_rn(dzdy, y0c)));
}

// Smallest / largest value the z plane takes over an inclusive rectangle of pixel coordinates, evaluated exactly like
// fragment() does: fma(dzdy, y, fma(dzdx, x, a0)) is monotonic in x and in y (rounding is monotonic), so the extreme is at
// the corner the two slopes point away from / towards.  NaN when the plane is not finite there: callers treat NaN as
// "unknown" (z24_of(NaN) = 0 never drops anything; a NaN maximum publishes no bound).
__device__ __forceinline__ float plane_min(float a0, float dzdx, float dzdy, int x0, int x1, int y0, int y1)
{
  const float xm = (float)(dzdx < 0.0f ? x1 : x0), ym = (float)(dzdy < 0.0f ? y1 : y0);
  return __fmaf_rn(dzdy, ym, __fmaf_rn(dzdx, xm, a0));
}
__device__ __forceinline__ float plane_max(float a0, float dzdx, float dzdy, int x0, int x1, int y0, int y1)
{
  const float xm = (float)(dzdx < 0.0f ? x0 : x1), ym = (float)(dzdy < 0.0f ? y0 : y1);
  return __fmaf_rn(dzdy, ym, __fmaf_rn(dzdx, xm, a0));
}
// A record is "near" when its plane may reach window z <= 0.5 somewhere in its bounding box (the same margin as the
// fragment path's hand-over: 0.51; a NaN counts as near).  Only near records take part in the tile kernel's exact-z pass.
__device__ __forceinline__ uint32_t near_bit(float a0, float dzdx, float dzdy, int bx0, int bx1, int by0, int by1)
{
  return !(plane_min(a0, dzdx, dzdy, bx0, bx1, by0, by1) >= 0.51f) ? kNearBit : 0u;
}

__device__ __forceinline__ PackedTri pack_record(int x0, int y0, int x1, int y1, int x2, int y2, float a0, float dzdx,
                                                 float dzdy, uint32_t order)
{
  // snapped coordinates of in-frustum vertices lie in [-128, 2048*256 + 128]; vertices made by the clipper can be a
  // few 1/256 px outside that (their interpolation is rounded): 20 bits after the bias of kCoordBias
  PackedTri pk;
  pk.v01 = (unsigned long long)(uint32_t)(x0 + kCoordBias) | ((unsigned long long)(uint32_t)(y0 + kCoordBias) << 20) | ((unsigned long long)(uint32_t)(x1 + kCoordBias) << 40);
  pk.v12 = (unsigned long long)(uint32_t)(y1 + kCoordBias) | ((unsigned long long)(uint32_t)(x2 + kCoordBias) << 20) | ((unsigned long long)(uint32_t)(y2 + kCoordBias) << 40);
  pk.a0 = a0; pk.dzdx = dzdx; pk.dzdy = dzdy; pk.order = order;
  return pk;
}

// Builds the raster record of one window-space triangle.  Returns false when it covers no
// pixel centre of the width x height frame.  `pk` receives the 32-byte bin form.
__device__ __forceinline__ bool make_record(Win v0, Win v1, Win v2, uint32_t order, int width, int height,
                                            TriRec& r, PackedTri& pk)
{
  int x0 = snap(v0.x), y0 = snap(v0.y);
  int x1 = snap(v1.x), y1 = snap(v1.y);
  const int x2 = snap(v2.x), y2 = snap(v2.y);
  // bounding box of covered pixel centres first: most mesh triangles are sub-pixel and end here
  const int minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
  const int miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  int bx0 = (minx + 255) >> 8, bx1 = (maxx - 1) >> 8;
  int by0 = (miny + 255) >> 8, by1 = (maxy - 1) >> 8;
  bx0 = max(bx0, 0); by0 = max(by0, 0);
  bx1 = min(bx1, width - 1); by1 = min(by1, height - 1);
  if (bx1 < bx0 || by1 < by0) return false;
  const long long area = (long long)(x0 - x1) * (y2 - y0) - (long long)(x2 - x0) * (y0 - y1);
  if (area == 0) return false;
  if (area < 0) {   // orient: swap vertices 0 and 1 (fixed and float)
    int t = x0; x0 = x1; x1 = t;
    t = y0; y0 = y1; y1 = t;
    Win tw = v0; v0 = v1; v1 = tw;
  }
  edges_from_snapped(x0, y0, x1, y1, x2, y2, width, height, r);
  // z plane from the unsnapped float vertices
  const float x0c = __fsub_rn(v0.x, 0.5f), y0c = __fsub_rn(v0.y, 0.5f);
  const float dx01 = __fsub_rn(v0.x, v1.x), dy01 = __fsub_rn(v0.y, v1.y);
  const float dx20 = __fsub_rn(v2.x, v0.x), dy20 = __fsub_rn(v2.y, v0.y);
  const float ooa = __fdiv_rn(1.0f, __fsub_rn(__fmul_rn(dx01, dy20), __fmul_rn(dy01, dx20)));
  const float dy20o = __fmul_rn(dy20, ooa), dy01o = __fmul_rn(dy01, ooa);
  const float dx20o = __fmul_rn(dx20, ooa), dx01o = __fmul_rn(dx01, ooa);
  const float da01 = __fsub_rn(v0.z, v1.z), da20 = __fsub_rn(v2.z, v0.z);
  r.dzdx = __fsub_rn(__fmul_rn(da01, dy20o), __fmul_rn(da20, dy01o));
  r.dzdy = __fsub_rn(__fmul_rn(da20, dx01o), __fmul_rn(da01, dx20o));
  r.a0 = __fsub_rn(v0.z, __fadd_rn(__fmul_rn(r.dzdx, x0c), __fmul_rn(r.dzdy, y0c)));
  r.order = order;
  r.pad = 0;
  // snapped coordinates of in-frustum vertices lie in [-128, 2048*256 + 128]; vertices made by the clipper can be a
  // few 1/256 px outside that (their interpolation is rounded): 20 bits after the bias of kCoordBias
  pk.v01 = (unsigned long long)(uint32_t)(x0 + kCoordBias) | ((unsigned long long)(uint32_t)(y0 + kCoordBias) << 20) | ((unsigned long long)(uint32_t)(x1 + kCoordBias) << 40);
  pk.v12 = (unsigned long long)(uint32_t)(y1 + kCoordBias) | ((unsigned long long)(uint32_t)(x2 + kCoordBias) << 20) | ((unsigned long long)(uint32_t)(y2 + kCoordBias) << 40);
  pk.a0 = r.a0; pk.dzdx = r.dzdx; pk.dzdy = r.dzdy;
  pk.order = order | near_bit(r.a0, r.dzdx, r.dzdy, (int)(r.bbx & 0xffff), (int)(r.bbx >> 16), (int)(r.bby & 0xffff), (int)(r.bby >> 16));
  return true;
}

__device__ __forceinline__ TriRec unpack_record(const PackedTri& pk, int width, int height)
{
  TriRec r;
  const int x0 = (int)(pk.v01 & 0xfffffu) - kCoordBias, y0 = (int)((pk.v01 >> 20) & 0xfffffu) - kCoordBias, x1 = (int)((pk.v01 >> 40) & 0xfffffu) - kCoordBias;
  const int y1 = (int)(pk.v12 & 0xfffffu) - kCoordBias, x2 = (int)((pk.v12 >> 20) & 0xfffffu) - kCoordBias, y2 = (int)((pk.v12 >> 40) & 0xfffffu) - kCoordBias;
  edges_from_snapped(x0, y0, x1, y1, x2, y2, width, height, r);
  r.a0 = pk.a0; r.dzdx = pk.dzdx; r.dzdy = pk.dzdy; r.order = pk.order & kOrderMask; r.pad = pk.order >> 31;
  return r;
}

__device__ __forceinline__ void store_record(PackedTri* dst, const PackedTri& r)
{
  const uint4* s = reinterpret_cast<const uint4*>(&r);
  uint4* d = reinterpret_cast<uint4*>(dst);
  d[0] = s[0]; d[1] = s[1];
}

// Records whose bounding box touches more than kCoopTiles tiles (the robot's own arm in front of the camera, clipped
// near-plane triangles, walls) are not appended to their bins here: they go on the shard's big-record list (one
// reservation per wave), and bigrec_kernel appends every list entry with a wave of its own.  Appending them where they
// are made costs a wave one atomic round trip per record, and such records come in runs (a wall's fan: hundreds of
// (record, tile) pairs in a handful of waves while the rest of the GPU has finished).
constexpr int kCoopTiles = 4;            // (8: the same; 2: +20 us of bigrec_kernel; 1: every straddling record on the lists, 0.5 ms)
#ifndef RTUF_FRONT_AREA
#define RTUF_FRONT_AREA 24
#endif
constexpr int kFrontArea = RTUF_FRONT_AREA;      // boxes up to this many pixel centres are binned from the front of a bin
__device__ __forceinline__ void list_big_records_wave(const SetupArgs& a, int shard_id, int slot, bool big, const PackedTri& pk)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long bm = __ballot(big);
  const int leader = __ffsll((long long)bm) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(&a.counters->shard[shard_id].big_count, (uint32_t)__popcll(bm));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  if (big) {
    const uint32_t idx = base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
    if (idx < a.big_capacity) {          // (an over-full list is detected from the counter and the batch run again)
      uint4* dst = reinterpret_cast<uint4*>(a.big_list + (size_t)shard_id * a.big_capacity + idx);
      const uint4* src = reinterpret_cast<const uint4*>(&pk);
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = make_uint4((uint32_t)slot, 0u, 0u, 0u);
    }
  }
}

// Wave-cooperative form: all 64 lanes call it; lanes with `have` own a record.  Lanes that
// target the same bin are grouped with ballots (ALU only), then every group leader issues its
// atomicAdd in the SAME instruction, so a wave pays one atomic round trip per tile index
// instead of one per distinct bin; group members get consecutive slots, which makes the
// 32-byte record stores of neighbouring mesh triangles contiguous.
__device__ __forceinline__ uint32_t emit_record_wave(const SetupArgs& a, int shard_id, int slot, bool have, uint32_t bbx, uint32_t bby, const PackedTri& pk)
{
  const int lane = threadIdx.x & 63;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (have) {
    tx0 = (int)(bbx & 0xffff) / kTileW; tx1 = (int)(bbx >> 16) / kTileW;
    ty0 = (int)(bby & 0xffff) / kTileH; ty1 = (int)(bby >> 16) / kTileH;
  }
  const int tiles = a.tiles_x * a.tiles_y;
  uint32_t n = 0;
  // Two size classes per bin: records with a box of at most kFrontArea pixel centres fill the bin from
  // the front, larger ones from the back, so the tile kernel's waves get boxes of similar size (its
  // lane-per-triangle walk runs as long as the largest box in the wave).
  const int cls = ((int)(bbx >> 16) - (int)(bbx & 0xffff) + 1) * ((int)(bby >> 16) - (int)(bby & 0xffff) + 1) > kFrontArea ? 1 : 0;
  const bool big = have && (tx1 - tx0 + 1) * (ty1 - ty0 + 1) > kCoopTiles;
  if (__ballot(big)) list_big_records_wave(a, shard_id, slot, big, pk);          // (their bin entries are counted by bigrec_kernel)
  have = have && !big;
  int tx = tx0, ty = ty0;                  // walks the touched tiles row by row
  for (;;) {
    const bool act = have && ty <= ty1;
    unsigned long long pending = __ballot(act);
    if (!pending) break;
    const int bin = act ? 2 * (__mul24(slot, tiles) + __mul24(ty, a.tiles_x) + tx) + cls : -1;      // (bin, class) = one counter
    RTUF_LANES(kLaneSetupRecTile, act);
    if (++tx > tx1) { tx = tx0; ty++; }
    unsigned long long mymask = 0;
    int myleader = lane;
    while (pending) {
      const int leader = __ffsll((long long)pending) - 1;
      const int lbin = __builtin_amdgcn_readlane(bin, leader);      // leader is wave-uniform: no LDS crossbar round trip
      const unsigned long long m = __ballot(act && bin == lbin);
      RTUF_LANES(kLaneSetupRecGroup, act && bin == lbin);
      if (act && bin == lbin) { mymask = m; myleader = leader; }
      pending &= ~m;
    }
    uint32_t base = 0;
    if (act && lane == myleader) base = atomicAdd(&a.bin_hdr[bin >> 1].count[cls], (uint32_t)__popcll(mymask));
    // a near record (it may produce window z <= 0.5) marks its bin: top bit of the fragment counter, read by the tile kernel
    const unsigned long long nearm = __ballot(act && (pk.order & kNearBit) != 0u);
    if (nearm && act && lane == myleader && (mymask & nearm)) atomicOr(&a.fbin_count[bin >> 1], 0x80000000u);
    base = __shfl(base, myleader);
    if (act) {
      const uint32_t pos = base + (uint32_t)__popcll(mymask & ((1ull << lane) - 1ull));
      if (pos < a.capacity) store_record(a.bins + (size_t)(bin >> 1) * a.capacity + (cls ? a.capacity - 1u - pos : pos), pk);
      n++;
    }
  }
  return n;
}

// 24-bit depth-buffer value of a window z: round(clamp(z, 0, 1) * (2^24 - 1)), half to even (llvmpipe Z24)
__device__ __forceinline__ uint32_t z24_of(float z)
{
  const float zc = fminf(fmaxf(z, 0.0f), 1.0f);
  return (uint32_t)__float2int_rn(__fmul_rn(zc, 16777215.0f));
}

// 24-bit depth of a window z that is KNOWN to be above 0.5 (tiles without near geometry: every record and fragment there has
// z >= 0.51 over its whole box, that is what kNearBit / the bin's near flag say): p = clamp(z) * 16777215 then lies in
// [2^23, 2^24), where a float IS an integer (ulp 1: the product's rounding is the rounding to integer, half to even, that
// v_rndne_f32 would repeat), and its bit pattern is 0x4B000000 + (p - 2^23): one integer add instead of v_rndne + v_cvt.
__device__ __forceinline__ uint32_t z24_of_upper_half(float z)
{
  const float zc = fminf(fmaxf(z, 0.0f), 1.0f);
  return __float_as_uint(__fmul_rn(zc, 16777215.0f)) - 0x4A800000u;
}

// Set-up + coverage of a triangle (snapped coordinates x0..y2 from phase 1) whose pixel-centre bounding box
// is at most N x N (N <= 4): returns
// the covered box positions (bit dy*4+dx, origin bx0,by0) and orients v0/v1 like orient_and_bound.
// All integer work is done relative to the box origin, where vertex coordinates are within
// [-256, (N+1)*256] and every product fits 32 bits (full-rate 24-bit multiplies); the values are
// the same integers the absolute 64-bit form yields.  Straight-line: all N*N candidates per lane.
template <int N>
__device__ __forceinline__ uint32_t small_box_coverage(Win& v0, Win& v1, const Win& v2, int x0, int y0, int x1, int y1, int x2, int y2,
                                                       int width, int height, int& bx0, int& by0)
{
  const int minx = min(x0, min(x1, x2)), maxx = max(x0, max(x1, x2));
  const int miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
  bx0 = max((minx + 255) >> 8, 0);
  by0 = max((miny + 255) >> 8, 0);
  const int bx1 = min((maxx - 1) >> 8, width - 1), by1 = min((maxy - 1) >> 8, height - 1);
  if (bx1 < bx0 || by1 < by0) return 0;
  const int ox = bx0 << 8, oy = by0 << 8;
  x0 -= ox; x1 -= ox; x2 -= ox;
  y0 -= oy; y1 -= oy; y2 -= oy;
  const int area = __mul24(x0 - x1, y2 - y0) - __mul24(x2 - x0, y0 - y1);
  if (area == 0) return 0;
  if (area < 0) {
    int t = x0; x0 = x1; x1 = t;
    t = y0; y0 = y1; y1 = t;
    Win tw = v0; v0 = v1; v1 = tw;
  }
  const int xs[3] = {x0, x1, x2}, ys[3] = {y0, y1, y2};
  int A[3], B[3], E[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3;
    const int dcdx = ys[i] - ys[j], dcdy = xs[i] - xs[j];
    int c = __mul24(dcdx, xs[i]) - __mul24(dcdy, ys[i]);
    if (dcdx < 0 || (dcdx == 0 && dcdy > 0)) c += 1;      // (as shifts and subtractions, like edges_from_snapped: 65 instead of 59 registers, 7 waves/SIMD)
    A[i] = -dcdx; B[i] = dcdy;
    E[i] = -((-c) >> 8) - 1;          // inside <=> E + A*dx + B*dy >= 0 (see edges_from_snapped)
  }
  uint32_t out = 0;
#pragma unroll
  for (int dy = 0; dy < N; dy++) {
#pragma unroll
    for (int dx = 0; dx < N; dx++) {
      const int v = (E[0] + dx * A[0] + dy * B[0]) | (E[1] + dx * A[1] + dy * B[1]) | (E[2] + dx * A[2] + dy * B[2]);
      out |= (uint32_t)(~v >> 31 & 1) << (dy * 4 + dx);
    }
  }
  const uint32_t cols = ((2u << (bx1 - bx0)) - 1u) * 0x1111u;
  const uint32_t rows = (uint32_t)((1ull << (4 * (by1 - by0 + 1))) - 1ull);
  return out & cols & rows;
}

// Wave-cooperative emission of the covered pixels of small triangles as 8-byte fragments.  `mask`
// holds the coverage of the box positions (bit dy*4+dx, box origin bx0,by0).  Lanes are grouped by bin
// with ballots, the group leaders reserve popcount(mask) slots each in ONE atomic round trip and
// every lane writes its fragments to consecutive slots.  With MAY_STRADDLE a box whose covered
// positions lie in more than one tile falls back to one plain atomic per fragment.
template <bool MAY_STRADDLE>
__device__ __forceinline__ uint32_t emit_fragments_wave(const SetupArgs& a, int slot, uint32_t mask, int bx0, int by0,
                                                        float a0, float dzdx, float dzdy, uint32_t order)
{
  const int lane = threadIdx.x & 63;
  const int tiles = a.tiles_x * a.tiles_y;
  bool straddle = false;
  if (MAY_STRADDLE) {
    // position dx (dy) lies in the next tile when bx0 % kTileW + dx >= kTileW
    const int rx = kTileW - (bx0 % kTileW), ry = kTileH - (by0 % kTileH);     // positions left in this tile
    const uint32_t colmask = rx >= 4 ? 0xffffu : (((1u << rx) - 1u) * 0x1111u);
    const uint32_t rowmask = ry >= 4 ? 0xffffu : ((1u << (4 * ry)) - 1u);
    straddle = (mask & ~(colmask & rowmask)) != 0;
  }
  const bool act = mask && !straddle;
  const uint32_t cnt = (uint32_t)__popc(mask);
  unsigned long long pending = __ballot(act);
  if (pending) {
    const int bin = act ? __mul24(slot, tiles) + __mul24(by0 / kTileH, a.tiles_x) + (bx0 / kTileW) : -1;
    const unsigned long long c0 = __ballot(act && (cnt & 1u)), c1 = __ballot(act && (cnt & 2u)), c2 = __ballot(act && (cnt & 4u)),
                             c3 = __ballot(act && (cnt & 8u)), c4 = __ballot(act && (cnt & 16u));
    const unsigned long long below = (1ull << lane) - 1ull;
    // The loop only finds every lane's group (the lanes that target the same bin) -- a few instructions per distinct bin.
    // The lane's offset inside its group's reservation and the group's total are then computed ONCE, from the group mask
    // and the five bit planes of the counts (they used to be recomputed inside the loop for every distinct bin: with the
    // three to six bins a wave of small triangles touches, that was most of this function).
    unsigned long long mymask = 0ull;
    int myleader = lane;
    while (pending) {
      const int leader = __ffsll((long long)pending) - 1;
      const int lbin = __builtin_amdgcn_readlane(bin, leader);      // leader is wave-uniform: no LDS crossbar round trip
      const bool mine = act && bin == lbin;
      const unsigned long long m = __ballot(mine);
      RTUF_LANES(kLaneSetupFragGroup, mine);
      if (mine) { mymask = m; myleader = leader; }
      pending &= ~m;
    }
    const unsigned long long lo = mymask & below;
    const uint32_t base_in_group = (uint32_t)(__popcll(c0 & lo) + 2 * __popcll(c1 & lo) + 4 * __popcll(c2 & lo) + 8 * __popcll(c3 & lo) + 16 * __popcll(c4 & lo));
    // the group's total = offset + count of the group's highest lane
    const int top = 63 - __clzll((long long)(mymask | 1ull));
    const uint32_t group_total = (uint32_t)__shfl((int)(base_in_group + cnt), top);
    uint32_t base = 0;
    if (act && lane == myleader) base = atomicAdd(&a.fbin_count[bin], group_total) & 0x7fffffffu;      // (top bit: the bin's near flag)
    base = __shfl(base, myleader);
    if (act) {
      uint32_t pos = base + base_in_group;
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.fbins) + (size_t)bin * a.fcapacity;
      const int lbase = (by0 % kTileH) * kTileW + (bx0 % kTileW);
      uint32_t m = mask;
      while (m) {
        RTUF_LANES(kLaneSetupFragStore, true);
        const int k = __ffs((int)m) - 1;
        m &= m - 1;
        const int px = bx0 + (k & 3), py = by0 + (k >> 2);
        if (pos < a.fcapacity) {
          const float z = __fmaf_rn(dzdy, (float)py, __fmaf_rn(dzdx, (float)px, a0));
          // (a triangle that may reach z <= 0.51 anywhere in its box never gets here: it was handed to the record pass)
          const unsigned long long f = ((unsigned long long)(RTUF_FAST_CLASS ? z24_of_upper_half(z) : z24_of(z)) << 40) | ((unsigned long long)order << kFragPosBits) |
                                       (unsigned long long)(lbase + (k >> 2) * kTileW + (k & 3));
          dst[pos] = f;
        }
        pos++;
      }
    }
  }
  if (MAY_STRADDLE && straddle) {
    uint32_t m = mask;
    while (m) {
      const int k = __ffs((int)m) - 1;
      m &= m - 1;
      const int px = bx0 + (k & 3), py = by0 + (k >> 2);
      const int bin = __mul24(slot, tiles) + __mul24(py / kTileH, a.tiles_x) + (px / kTileW);
      const uint32_t pos = atomicAdd(&a.fbin_count[bin], 1u) & 0x7fffffffu;
      if (pos < a.fcapacity) {
        const float z = __fmaf_rn(dzdy, (float)py, __fmaf_rn(dzdx, (float)px, a0));
        const unsigned long long f = ((unsigned long long)(RTUF_FAST_CLASS ? z24_of_upper_half(z) : z24_of(z)) << 40) | ((unsigned long long)order << kFragPosBits) |
                                     (unsigned long long)((py % kTileH) * kTileW + (px % kTileW));
        reinterpret_cast<unsigned long long*>(a.fbins)[(size_t)bin * a.fcapacity + pos] = f;
      }
    }
  }
  return cnt;
}

// True when the chunk's bounding box lies completely outside frustum plane `plane` (0..5 =
// +x,-x,+y,-y,near,far): then all its vertices carry that plane's clip bit and every triangle
// would be rejected by the per-triangle test anyway (same result, decided once per chunk).
// Margins keep the test conservative.
__device__ __forceinline__ bool chunk_outside_plane(const float* M, const Chunk& ch, int plane)
{
  const int k = plane >> 1;
  const float sg = (plane & 1) ? 1.0f : -1.0f;          // plane: w + sg * coord >= 0
  const float cx = ch.center[0], cy = ch.center[1], cz = ch.center[2];
  const float cw = M[3] * cx + M[7] * cy + M[11] * cz + M[15];
  const float ck = M[k] * cx + M[4 + k] * cy + M[8 + k] * cz + M[12 + k];
  const float nx = M[3] + sg * M[k], ny = M[7] + sg * M[4 + k], nz = M[11] + sg * M[8 + k];
  const float d = cw + sg * ck;
  // the corner of the box farthest along the plane normal is sum |n_i| * half_i above the centre
  const float reach = fabsf(nx) * ch.half[0] + fabsf(ny) * ch.half[1] + fabsf(nz) * ch.half[2];
  const float slack = reach * 1.001f + 1e-5f * (fabsf(cw) + fabsf(ck)) + 1e-30f;
  return d < -slack;
}

// setup_kernel: one workgroup per (chunk, group of kStreamsPerBlock streams).
//   once:        the group's matrices are fetched into LDS; one lane per (stream, frustum plane)
//                decides whether the whole chunk is outside for that stream
//   per stream:  phase 1  every chunk vertex once: vertex shader, clip test, viewport, 1/256-px
//                         snap -> LDS
//                phase 2  one lane per triangle on the snapped integers: trivial reject,
//                         frustum-crossers to the clip list, sub-pixel cull; survivors
//                         (typically < 20 %) are appended to an LDS work list
//   once:        phase 3  the work list of all streams is processed by DENSE waves: full
//                         set-up (edge functions, z plane) and binning
// cull_kernel: one workgroup per chunk, one thread per stream slot of the in-flight group.  Tests
// the chunk's bounding box against the six frustum planes of every stream (plus model selection /
// background mode) and appends the visible streams, kStreamsPerBlock at a time, to the work list
// the set-up kernel runs from.  The test is conservative, so it never changes the image; it only
// keeps ~70 % of the (chunk, stream) pairs of a robot that is partly in view from ever starting a
// set-up workgroup.
static_assert(kStreamsPerBlock <= 4, "WorkItem holds at most 4 stream slots");
__global__ __launch_bounds__(kBlock) void cull_kernel(SetupArgs a)
{
  __shared__ uint16_t s_vis[kBlock];
  __shared__ uint32_t s_wave_n[kBlock / 64];
  __shared__ uint32_t s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk_id = blockIdx.x;
  const Chunk ch = a.chunks[chunk_id];
  const bool is_bg = (uint32_t)chunk_id == a.bg_chunk;
  for (int first = 0; first < a.group_size; first += kBlock) {
    const int slot = first + tid;
    bool vis = false;
    if (slot < a.group_size) {
      const int stream = a.group_base + slot;
      vis = is_bg ? (a.bg[stream].mode == 0u) : (((a.model_mask[stream] >> ch.model) & 1ull) != 0ull);
      if (vis) {
        float M[16];
        const float4* src = reinterpret_cast<const float4*>(a.mvp + ((size_t)stream * (a.n_draws + 1) + ch.draw) * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float4 col = src[q];
          M[4 * q] = col.x; M[4 * q + 1] = col.y; M[4 * q + 2] = col.z; M[4 * q + 3] = col.w;
        }
#pragma unroll
        for (int pl = 0; pl < 6; pl++) vis = vis && !chunk_outside_plane(M, ch, pl);
      }
    }
    // ordered compaction (wave ballots + a 4-entry prefix): the list order is deterministic
    const unsigned long long vm = __ballot(vis);
    if (lane == 0) s_wave_n[wave] = (uint32_t)__popcll(vm);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { const uint32_t c = s_wave_n[w]; if (w < wave) before += c; total += c; }
    if (vis) s_vis[before + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = (uint16_t)slot;
    const uint32_t n_items = (total + kStreamsPerBlock - 1) / kStreamsPerBlock;
    if (tid == 0 && n_items) s_base = atomicAdd(&a.counters->work.n_items, n_items);
    __syncthreads();
    if ((uint32_t)tid < n_items) {
      WorkItem it;
      it.chunk = (uint32_t)chunk_id;
      it.tri_begin = ch.tri_begin; it.vert_begin = ch.vert_begin; it.reserved = 0; it.draw = ch.draw;
      it.tri_count = (uint16_t)ch.tri_count; it.vert_count = (uint16_t)ch.vert_count;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t j = (uint32_t)tid * kStreamsPerBlock + k;
        it.slot[k] = (k < kStreamsPerBlock && j < total) ? s_vis[j] : (uint16_t)0xffffu;
      }
      uint4* dst = reinterpret_cast<uint4*>(&a.items[s_base + tid]);
      dst[0] = reinterpret_cast<const uint4*>(&it)[0];
      dst[1] = reinterpret_cast<const uint4*>(&it)[1];
    }
    __syncthreads();
  }
}

// setup_kernel: one workgroup per work item = (chunk, up to kStreamsPerBlock streams that see it).  A chunk is
// <= 256 triangles of one draw call with its own <= kMaxChunkVerts vertex list, so each vertex is transformed
// once per stream (into LDS) instead of once per incident triangle, and the chunk's geometry is loaded
// once for all streams of the item.
template <bool STRIDED>
__global__ __launch_bounds__(kBlock) void setup_kernel(SetupArgs a, uint32_t item_base)
{
  __shared__ float s_win[kStreamsPerBlock][3][kMaxChunkVerts];  // window x, y, z (SoA: 12 B per vertex)
  __shared__ int2 s_snap[kStreamsPerBlock][kMaxChunkVerts];    // snapped x; snapped y << 8 | clip mask
  __shared__ uint32_t s_packed[kBlock];                         // the chunk's triangles
  __shared__ uint16_t s_list[kStreamsPerBlock * kBlock];        // survivors binned as records: stream k << 8 | triangle
  __shared__ uint16_t s_list2[kStreamsPerBlock * kBlock];       // survivors resolved to fragments: 4x4 boxes from the bottom, 2x2 from the top
  __shared__ uint32_t s_nlist, s_ntiny, s_nsmall;
  __shared__ uint32_t s_stat[3];
  __shared__ float s_mvp[kStreamsPerBlock][16];
  __shared__ uint32_t s_on[kStreamsPerBlock];
  __shared__ int s_slot[kStreamsPerBlock];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  if (tid < 3) s_stat[tid] = 0;
  const int shard_id = (int)(blockIdx.x % kCounterShards);
  CounterShard& shard = a.counters->shard[shard_id];
  const float sx = 0.5f * (float)a.width, sy = 0.5f * (float)a.height;
  uint32_t binned = 0, entries = 0, nfrag = 0;
  RTUF_LANES_INIT();
  // launch_setup sizes the main grid from the previous batch's work-list length: one item per
  // workgroup (STRIDED = false, no loop: keeps the register count at 8 waves/SIMD).  Items beyond that
  // grid, if the list grew, are swept by a small strided launch of the same code.
  const uint32_t n_items = a.counters->work.n_items;
  // (what this launch covers, for the batch's status word: the main grid takes items [0, gridDim.x); a strided sweep behind
  // it takes everything)
  if (blockIdx.x == 0 && tid == 0) a.counters->work.grid = STRIDED ? 0xffffffffu : gridDim.x;
  for (uint32_t item_id = item_base + blockIdx.x; item_id < n_items; item_id += gridDim.x) {
  if (tid == 3) s_nlist = 0;
  if (tid == 4) s_ntiny = 0;
  if (tid == 5) s_nsmall = 0;
  // the item carries the chunk's ranges, so the geometry loads depend on it alone
  struct { uint32_t tri_begin, vert_begin, draw, tri_count, vert_count; } ch;
  uint32_t slots01, slots23;
  // The shard whose clip list and many-tile list take this item's triangles comes from the item's CONTENT, not from the
  // workgroup that happens to run it: the cull kernel lists the items in the order its atomics land, so a shard chosen by
  // workgroup index filled differently on every run of the same batch (and a list sized to one run overflowed on the next).
  int list_shard;
  {
    const uint4* src = reinterpret_cast<const uint4*>(&a.items[item_id]);
    uint4 w0 = src[0], w1 = src[1];
    // same address in every lane: move the words to scalar registers so that everything derived
    // from them (ranges, loop bounds, matrix addresses) is wave-uniform for the compiler as well
    w0.x = __builtin_amdgcn_readfirstlane(w0.x); w0.y = __builtin_amdgcn_readfirstlane(w0.y);
    w0.z = __builtin_amdgcn_readfirstlane(w0.z); w0.w = __builtin_amdgcn_readfirstlane(w0.w);
    w1.x = __builtin_amdgcn_readfirstlane(w1.x); w1.y = __builtin_amdgcn_readfirstlane(w1.y);
    w1.z = __builtin_amdgcn_readfirstlane(w1.z); w1.w = __builtin_amdgcn_readfirstlane(w1.w);
    ch.tri_begin = w0.y; ch.vert_begin = w0.z;
    ch.draw = w1.x; ch.tri_count = w1.y & 0xffffu; ch.vert_count = w1.y >> 16;
    slots01 = w1.z; slots23 = w1.w;
    list_shard = (int)((w0.x + (w1.z & 0xffffu)) % (uint32_t)kCounterShards);      // chunk id + first stream slot
  }
  CounterShard& lshard = a.counters->shard[list_shard];
  auto item_slot = [&](int k) { return (int)(((k < 2 ? slots01 : slots23) >> (16 * (k & 1))) & 0xffffu); };

  float4 pv = make_float4(0, 0, 0, 1);
  const bool have_vert = tid < (int)ch.vert_count;
  if (have_vert) pv = a.cverts[ch.vert_begin + tid];
  const bool have_tri = tid < (int)ch.tri_count;
  const uint32_t packed = have_tri ? a.ctris[ch.tri_begin + tid] : 0u;
  s_packed[tid] = packed;
  const uint32_t i0 = packed & 1023u, i1 = (packed >> 10) & 1023u, i2 = (packed >> 20) & 1023u;
  // all matrices of the item's streams are fetched up front (one global round trip per workgroup
  // instead of one per stream) and kept in LDS
  if (tid < kStreamsPerBlock * 16) {
    const int k = tid >> 4, slot = item_slot(k);
    float v = 0.0f;
    if (slot != 0xffff)
      v = a.mvp[((size_t)(a.group_base + slot) * (a.n_draws + 1) + ch.draw) * 16 + (tid & 15)];
    s_mvp[k][tid & 15] = v;
  }
  if (tid >= 64 && tid < 64 + kStreamsPerBlock) {
    const int k = tid - 64;
    s_on[k] = item_slot(k) != 0xffff ? 1u : 0u;
    s_slot[k] = item_slot(k);
  }
  __syncthreads();

  for (int k = 0; k < kStreamsPerBlock; k++) {
    const int slot = s_slot[k];
    if (!s_on[k]) continue;                              // uniform per workgroup
    // (phase 1 of all three streams before ONE barrier, then phase 2 of all three: measured, no difference)
    // phase 1
    RTUF_LANES(kLaneSetupVert, have_vert);
    if (have_vert) {
      float M[16];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float4 col = reinterpret_cast<const float4*>(s_mvp[k])[q];
        M[4 * q] = col.x; M[4 * q + 1] = col.y; M[4 * q + 2] = col.z; M[4 * q + 3] = col.w;
      }
      float c[4];
      vs_position(M, pv.x, pv.y, pv.z, c);
      const Win w = viewport_vs(c, sx, sy);
      const unsigned cm = clipmask_of(c);
      s_win[k][0][tid] = w.x; s_win[k][1][tid] = w.y; s_win[k][2][tid] = w.z;
      s_snap[k][tid] = make_int2(snap(w.x), (int)(((unsigned)snap(w.y) << 8) | cm));
    }
    __syncthreads();
    // phase 2
    bool survive = false, needs_clip = false, tiny = false, small = false;
    RTUF_LANES(kLaneSetupTri, have_tri);
    if (have_tri) {
      const int2 p0 = s_snap[k][i0], p1 = s_snap[k][i1], p2 = s_snap[k][i2];
      const unsigned m0 = (unsigned)p0.y & 63u, m1 = (unsigned)p1.y & 63u, m2 = (unsigned)p2.y & 63u;
      if ((m0 & m1 & m2) == 0) {
        if ((m0 | m1 | m2) != 0) {
          needs_clip = true;
        } else {
          const int y0 = p0.y >> 8, y1 = p1.y >> 8, y2 = p2.y >> 8;
          const int minx = min(p0.x, min(p1.x, p2.x)), maxx = max(p0.x, max(p1.x, p2.x));
          const int miny = min(y0, min(y1, y2)), maxy = max(y0, max(y1, y2));
          const int bx0 = max((minx + 255) >> 8, 0), bx1 = min((maxx - 1) >> 8, a.width - 1);
          const int by0 = max((miny + 255) >> 8, 0), by1 = min((maxy - 1) >> 8, a.height - 1);
          survive = bx1 >= bx0 && by1 >= by0;
          tiny = survive && (bx1 - bx0) <= 1 && (by1 - by0) <= 1;
          // up to 4x4 candidates inside ONE tile: also resolved to fragments here (boxes that straddle
          // a tile boundary stay records)
          small = survive && !tiny && (bx1 - bx0) <= 3 && (by1 - by0) <= 3 && RTUF_SMALL_FRAGS &&
                  (bx0 / kTileW) == (bx1 / kTileW) && (by0 / kTileH) == (by1 / kTileH);
        }
      }
    }
    // triangles that cross a frustum plane go to clip_kernel: one list append per wave
    const unsigned long long cmk = __ballot(needs_clip);
    if (cmk) {
      const int leader = __ffsll((long long)cmk) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&lshard.clip_count, (uint32_t)__popcll(cmk));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
      if (needs_clip) {
        const uint32_t kk = base + (uint32_t)__popcll(cmk & ((1ull << lane) - 1ull));
        if (kk < a.clip_capacity) {
          uint4* dst = reinterpret_cast<uint4*>(&a.clip_list[(size_t)list_shard * a.clip_capacity + kk]);
          dst[0] = make_uint4((uint32_t)slot, ch.draw, ch.vert_begin, packed);
          dst[1] = make_uint4(a.corder[ch.tri_begin + tid], 0u, 0u, 0u);
        } else {
          lshard.clip_overflow = 1u;
        }
      }
    }
    // survivors go to LDS work lists: records into s_list; boxes resolved to fragments into s_list2
    // (<= 4x4 single-tile boxes from the bottom, <= 2x2 boxes from the top)
    const bool rec = survive && !tiny && !small;
    const unsigned long long sm = __ballot(rec), tm = __ballot(tiny), qm = __ballot(small);
    if (sm) {
      const int leader = __ffsll((long long)sm) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_nlist, (uint32_t)__popcll(sm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
      if (rec) s_list[base + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))] = (uint16_t)((k << 8) | tid);
    }
    if (tm) {
      const int leader = __ffsll((long long)tm) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_ntiny, (uint32_t)__popcll(tm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
      if (tiny) s_list2[kStreamsPerBlock * kBlock - 1 - (base + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull)))] = (uint16_t)((k << 8) | tid);
    }
    if (qm) {
      const int leader = __ffsll((long long)qm) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_nsmall, (uint32_t)__popcll(qm));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
      if (small) s_list2[base + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull))] = (uint16_t)((k << 8) | tid);
    }
  }
  __syncthreads();
  if (RTUF_ABL(a.flags, 0x10000u)) { __syncthreads(); if (!STRIDED) break; continue; }     // timing experiment
  // phase 3a/3b: tiny (2x2) and small (4x4, single tile) survivors -> coverage of their box positions
  // -> 8-byte fragments; the z plane (one division) is only evaluated for triangles that actually
  // cover a pixel centre
  const uint32_t ntiny = s_ntiny, nsmall = s_nsmall;
  if (tid == 0 && (ntiny | nsmall | s_nlist) == 0u) atomicAdd(&shard.zero_items, 1u);      // (statistics: an item the cull pass could have spared)
#pragma unroll
  for (int cls = 0; cls < 2; cls++) {
    const uint32_t ncls = cls == 0 ? ntiny : nsmall;
    for (uint32_t base = 0; base < ncls; base += kBlock) {
      const uint32_t j = base + tid;
      uint32_t mask = 0;
      int slot = 0, bx0 = 0, by0 = 0;
      float a0 = 0, dzdx = 0, dzdy = 0;
      uint32_t order = 0;
      bool near = false;
      uint32_t ent = 0;
      RTUF_LANES(kLaneSetupSmall, j < ncls);
      if (j < ncls) {
        Win v0, v1, v2;
        const uint32_t e = s_list2[cls == 0 ? kStreamsPerBlock * kBlock - 1 - j : j];
        ent = e;
        const int k = (int)(e >> 8), t = (int)(e & 255u);
        slot = s_slot[k];
        // (the draw-order number is asked for right away, not where it is needed: its global-memory latency then passes
        // under the coverage arithmetic instead of after it)
        order = a.corder[ch.tri_begin + t];
        const uint32_t p = s_packed[t];
        const uint32_t j0 = p & 1023u, j1 = (p >> 10) & 1023u, j2 = (p >> 20) & 1023u;
        v0.x = s_win[k][0][j0]; v0.y = s_win[k][1][j0]; v0.z = s_win[k][2][j0];
        v1.x = s_win[k][0][j1]; v1.y = s_win[k][1][j1]; v1.z = s_win[k][2][j1];
        v2.x = s_win[k][0][j2]; v2.y = s_win[k][1][j2]; v2.z = s_win[k][2][j2];
        const int2 q0 = s_snap[k][j0], q1 = s_snap[k][j1], q2 = s_snap[k][j2];      // snapped in phase 1 (y << 8 | clip mask)
        mask = cls == 0 ? small_box_coverage<2>(v0, v1, v2, q0.x, q0.y >> 8, q1.x, q1.y >> 8, q2.x, q2.y >> 8, a.width, a.height, bx0, by0)
                        : small_box_coverage<4>(v0, v1, v2, q0.x, q0.y >> 8, q1.x, q1.y >> 8, q2.x, q2.y >> 8, a.width, a.height, bx0, by0);
        if (mask) {
          z_plane(v0, v1, v2, a0, dzdx, dzdy);
          // z is monotone along x and along y (also as evaluated in float), so its minimum over the box
          // is at a corner.  Anything that may reach window z <= 0.5 needs its plane in the tile
          // kernel (exact float z): it goes out as a record instead (geometry within ~2 x near of the camera).
          // (monotone along each axis: the minimum is at the corner the two slopes point away from)
          const float xm = (float)(dzdx < 0.0f ? bx0 + 3 : bx0), ym = (float)(dzdy < 0.0f ? by0 + 3 : by0);
          const float zmin = __fmaf_rn(dzdy, ym, __fmaf_rn(dzdx, xm, a0));
          near = !(zmin >= 0.51f);      // (a NaN anywhere in the plane makes zmin NaN: counts as near)
        }
      }
      if (near) {      // rare: hand the triangle to the record pass below (one LDS append)
        s_list[atomicAdd(&s_nlist, 1u)] = (uint16_t)ent;
        mask = 0;
      }
      RTUF_LANES(kLaneSetupSmallHit, mask != 0);
      if (__ballot(mask != 0) && !RTUF_ABL(a.flags, 0x40000u)) {
        nfrag += cls == 0 ? emit_fragments_wave<true>(a, slot, mask, bx0, by0, a0, dzdx, dzdy, order)
                          : emit_fragments_wave<false>(a, slot, mask, bx0, by0, a0, dzdx, dzdy, order);
        binned += mask ? 1u : 0u;
      }
    }
  }
  __syncthreads();        // appends of the fragment pass to the record list
  // phase 3c: dense set-up + binning of the larger survivors as 32-byte records (the tile kernel
  // rebuilds the edge functions, so only orientation, bounding box and z plane are needed here)
  const uint32_t nlist = s_nlist;
  for (uint32_t base = 0; base < nlist; base += kBlock) {
    const uint32_t j = base + tid;
    bool have = false;
    PackedTri pk;
    uint32_t bbx = 0, bby = 0;
    int slot = 0;
    RTUF_LANES(kLaneSetupRec, j < nlist);
    if (j < nlist) {
      const uint32_t e = s_list[j];
      const int k = (int)(e >> 8), t = (int)(e & 255u);
      slot = s_slot[k];
      const uint32_t order = a.corder[ch.tri_begin + t];      // (asked for early, see above)
      const uint32_t p = s_packed[t];
      const uint32_t j0 = p & 1023u, j1 = (p >> 10) & 1023u, j2 = (p >> 20) & 1023u;
      Win v0, v1, v2;
      v0.x = s_win[k][0][j0]; v0.y = s_win[k][1][j0]; v0.z = s_win[k][2][j0];
      v1.x = s_win[k][0][j1]; v1.y = s_win[k][1][j1]; v1.z = s_win[k][2][j1];
      v2.x = s_win[k][0][j2]; v2.y = s_win[k][1][j2]; v2.z = s_win[k][2][j2];
      const int2 q0 = s_snap[k][j0], q1 = s_snap[k][j1], q2 = s_snap[k][j2];        // snapped in phase 1 (y << 8 | clip mask)
      int x0 = q0.x, y0 = q0.y >> 8, x1 = q1.x, y1 = q1.y >> 8, x2 = q2.x, y2 = q2.y >> 8, bx0, bx1, by0, by1;
      have = orient_and_bound(v0, v1, v2, a.width, a.height, x0, y0, x1, y1, x2, y2, bx0, bx1, by0, by1);
      if (have) {
        float a0, dzdx, dzdy;
        z_plane(v0, v1, v2, a0, dzdx, dzdy);
        pk = pack_record(x0, y0, x1, y1, x2, y2, a0, dzdx, dzdy, order | near_bit(a0, dzdx, dzdy, bx0, bx1, by0, by1));
        bbx = (uint32_t)bx0 | ((uint32_t)bx1 << 16);
        bby = (uint32_t)by0 | ((uint32_t)by1 << 16);
      }
    }
    RTUF_LANES(kLaneSetupRecHit, have);
    if (__ballot(have) && !RTUF_ABL(a.flags, 0x20000u)) {
      entries += emit_record_wave(a, list_shard, slot, have, bbx, bby, pk);
      binned += have ? 1u : 0u;
    }
  }
  if (!STRIDED) break;
  __syncthreads();        // LDS is reused by the next item
  }
  // statistics: one (sharded) atomic triple per workgroup
  uint32_t b = binned, e = entries, f = nfrag;
  for (int off = 32; off > 0; off >>= 1) { b += __shfl_down(b, off); e += __shfl_down(e, off); f += __shfl_down(f, off); }
  if (lane == 0 && b) {
    atomicAdd(&s_stat[0], b);
    atomicAdd(&s_stat[1], e);
    atomicAdd(&s_stat[2], f);
  }
  __syncthreads();
  if (tid == 0 && s_stat[0]) {
    atomicAdd(&shard.tris_binned, (unsigned long long)s_stat[0]);
    atomicAdd(&shard.bin_entries, (unsigned long long)s_stat[1]);
    atomicAdd(&shard.frags, (unsigned long long)s_stat[2]);
  }
  RTUF_LANES_FLUSH(shard);
}

// ---------------------------------------------------------------------------------------
// clip_kernel: one thread per triangle that crosses a frustum plane.  Sutherland-Hodgman in
// clip space, planes in the order +x, -x, +y, -y, near, far; each new vertex is interpolated
// from the end point closer to the plane; the polygon is emitted as the fan
// (v[i-1], v[i], v[0]).
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ float clipdist(const float* c, int plane)
{
  // dot4(clip, plane) with planes (-1,0,0,1) (1,0,0,1) (0,-1,0,1) (0,1,0,1) (0,0,1,1) (0,0,-1,1)
  const float px = plane == 0 ? -1.0f : (plane == 1 ? 1.0f : 0.0f);
  const float py = plane == 2 ? -1.0f : (plane == 3 ? 1.0f : 0.0f);
  const float pz = plane == 4 ? 1.0f : (plane == 5 ? -1.0f : 0.0f);
  float s = __fmul_rn(c[0], px);
  s = __fadd_rn(s, __fmul_rn(c[1], py));
  s = __fadd_rn(s, __fmul_rn(c[2], pz));
  s = __fadd_rn(s, c[3]);
  return s;
}

constexpr int kClipBlock = 128;          // threads per clip workgroup (16 KB of LDS polygon storage)
constexpr int kClipMaxV = 16;            // clip-space vertices per triangle: 3 + at most 2 new ones per frustum plane (+1 spare)
constexpr int kClipLdsV = 8;             // ... of which this many live in LDS (a triangle that crosses one or two planes needs 5..7);
                                         // the rest, for the rare triangle that crosses three and more, in a per-thread global spill area
constexpr int kClipGridWgs = 64;         // workgroups per counter shard at most (the spill area is sized for the grid)
constexpr int kClipMaxP = 12;            // polygon vertices (5-bit pool indices packed in one 64-bit register)

// Per-thread polygon clipper.  The vertex pool lives in LDS (thread-interleaved float4s: conflict
// free, no scratch memory) and the two polygon index lists are 5-bit fields of 64-bit registers.
__device__ __forceinline__ int list_get(unsigned long long l, int i) { return (int)((l >> (5 * i)) & 31ull); }

__device__ __forceinline__ void clip_one(const SetupArgs& a, const ClipItem it, int shard_id, float4 (*lds_pool)[kClipBlock], bool valid)
{
  const int t = threadIdx.x;
  float4* const spill = a.clip_spill + (size_t)blockIdx.x * kClipBlock + t;        // [kClipMaxV - kClipLdsV][grid threads]
  const size_t spill_stride = (size_t)gridDim.x * kClipBlock;
  auto pget = [&](int i) -> float4 { return i < kClipLdsV ? lds_pool[i][t] : spill[(size_t)(i - kClipLdsV) * spill_stride]; };
  auto pset = [&](int i, const float4& v) { if (i < kClipLdsV) lds_pool[i][t] = v; else spill[(size_t)(i - kClipLdsV) * spill_stride] = v; };
  const int slot = (int)it.slot, stream = a.group_base + slot;
  const float* __restrict__ M = a.mvp + ((size_t)stream * (a.n_draws + 1) + it.draw) * 16;
  const uint32_t packed = it.packed;
  const uint32_t order = it.order;
  const float sx = 0.5f * (float)a.width, sy = 0.5f * (float)a.height;

  int npool = 3;
  unsigned ormask = 0;
  {
    const uint32_t vi[3] = {packed & 1023u, (packed >> 10) & 1023u, (packed >> 20) & 1023u};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float4 p = a.cverts[it.vert_begin + vi[i]];
      float c[4];
      vs_position(M, p.x, p.y, p.z, c);
      lds_pool[i][t] = make_float4(c[0], c[1], c[2], c[3]);
      ormask |= clipmask_of(c);
    }
  }
  unsigned long long inl = 0ull | (1ull << 5) | (2ull << 10), outl = 0;
  int nv = valid ? 3 : 0;
  if (RTUF_ABL(a.flags, 0x200000u)) nv = 0;      // timing experiment: loads and vertex transform only
  unsigned clipmask = ormask;
  bool bad = false;
  while (clipmask && nv >= 3 && !bad) {
    const int plane = __ffs((int)clipmask) - 1;
    clipmask &= ~(1u << plane);
    if (nv >= kClipMaxP) { bad = true; break; }
    int prev = list_get(inl, 0);
    float4 cprev = pget(prev);
    float dp_prev = clipdist(&cprev.x, plane);
    int outc = 0;
    outl = 0;
    for (int i = 1; i <= nv; i++) {
      const int cur = list_get(inl, i == nv ? 0 : i);
      const float4 ccur = pget(cur);
      const float dp = clipdist(&ccur.x, plane);
      bool diff;
      if (dp_prev >= 0.0f) {
        if (outc >= kClipMaxP) { bad = true; break; }
        outl |= (unsigned long long)prev << (5 * outc++);
        diff = dp < 0.0f;
      } else {
        diff = !(dp < 0.0f);
      }
      if (diff) {
        if (npool >= kClipMaxV || outc >= kClipMaxP) { bad = true; break; }
        const float denom = __fsub_rn(dp, dp_prev);
        bool from_cur;
        if (dp < 0.0f) from_cur = dp_prev > -dp;        // going out
        else from_cur = !(dp > -dp_prev);               // coming in
        const float tt = from_cur ? __fdiv_rn(dp, denom) : __fdiv_rn(-dp_prev, denom);
        const float4 o = from_cur ? ccur : cprev;
        const float4 in = from_cur ? cprev : ccur;
        float4 nc;
        nc.x = __fadd_rn(__fmul_rn(__fsub_rn(in.x, o.x), tt), o.x);
        nc.y = __fadd_rn(__fmul_rn(__fsub_rn(in.y, o.y), tt), o.y);
        nc.z = __fadd_rn(__fmul_rn(__fsub_rn(in.z, o.z), tt), o.z);
        nc.w = __fadd_rn(__fmul_rn(__fsub_rn(in.w, o.w), tt), o.w);
        pset(npool, nc);
        outl |= (unsigned long long)npool << (5 * outc++);
        npool++;
      }
      prev = cur;
      cprev = ccur;
      dp_prev = dp;
    }
    inl = outl;
    nv = outc;
  }
  if (bad || nv < 3) nv = 0;           // (all lanes stay for the cooperative emission below)
  if (RTUF_ABL(a.flags, 0x100000u)) nv = 0;      // timing experiment: clip only, emit nothing
  // window coordinates: shaded (original) vertices and clipper-made ones go through different
  // viewport arithmetic (viewport_vs / viewport_clip)
  auto window_of = [&](int idx) {
    const float4 c4 = pget(idx);
    const float c[4] = {c4.x, c4.y, c4.z, c4.w};
    return idx < 3 ? viewport_vs(c, sx, sy) : viewport_clip(c, sx, sy);
  };
  Win w0, wprev;
  if (nv) { w0 = window_of(list_get(inl, 0)); wprev = window_of(list_get(inl, 1)); }
  uint32_t binned = 0, entries = 0;
  // the polygon as a fan (v[i-1], v[i], v[0]); the wave appends its records together (emit_record_wave):
  // the clipped triangles of a stream pile up in the few tiles along the frustum planes, and one
  // reservation per wave and bin instead of one per lane keeps the atomics off each other's cache lines
  for (int i = 2; __ballot(i < nv); i++) {
    bool have = false;
    TriRec r;
    PackedTri pk;
    r.bbx = r.bby = 0;
    if (i < nv) {
      const Win wi = window_of(list_get(inl, i));
      have = make_record(wprev, wi, w0, order, a.width, a.height, r, pk);
      wprev = wi;
    }
    binned += have ? 1u : 0u;
    if (__ballot(have) && !RTUF_ABL(a.flags, 0x800000u)) entries += emit_record_wave(a, shard_id, slot, have, r.bbx, r.bby, pk);      // (0x800000: timing experiment)
  }
  // statistics: one atomic pair per wave (per-lane atomics on a shard's counters serialise at one L2
  // atomic unit -- that alone used to be three quarters of this kernel's time)
  for (int off = 32; off > 0; off >>= 1) { binned += __shfl_down(binned, off); entries += __shfl_down(entries, off); }
  if ((threadIdx.x & 63) == 0 && binned) {
    atomicAdd(&a.counters->shard[shard_id].tris_binned, (unsigned long long)binned);
    atomicAdd(&a.counters->shard[shard_id].bin_entries, (unsigned long long)entries);
  }
}

__global__ __launch_bounds__(kClipBlock) void clip_kernel(SetupArgs a)
{
  __shared__ float4 s_pool[kClipLdsV][kClipBlock];
  // workgroups b, b + kCounterShards, ... serve shard b % kCounterShards
  const int shard_id = blockIdx.x % kCounterShards;
  const uint32_t n = min(a.counters->shard[shard_id].clip_count, a.clip_capacity);
  const uint32_t per = gridDim.x / kCounterShards;
  const ClipItem* list = a.clip_list + (size_t)shard_id * a.clip_capacity;
  // Whole blocks of consecutive list entries per workgroup: neighbours come from the same chunk and stream, so the item,
  // vertex and matrix loads of a wave share cache lines.  (While many-tile records were appended right here, runs of
  // wall triangles had to be dealt out 16 at a time over the workgroups; with bigrec_kernel doing those appends whole
  // blocks are the faster order again.)  Only as many workgroups as the list needs take part: the grid is fixed, the
  // list length lives on the device.
  const uint32_t wg = blockIdx.x / kCounterShards;
  const uint32_t used = min(per, (n + blockDim.x - 1) / blockDim.x);
  if (wg >= used) return;
  for (uint32_t base = 0; base < n; base += used * blockDim.x) {
    const uint32_t i = base + wg * blockDim.x + threadIdx.x;          // whole waves stay in the loop (cooperative emission)
    ClipItem it; it.slot = 0; it.draw = 0; it.vert_begin = 0; it.packed = 0; it.order = 0;
    if (i < n) {
      const uint4* src = reinterpret_cast<const uint4*>(&list[i]);
      const uint4 w0 = src[0];
      it.slot = w0.x; it.draw = w0.y; it.vert_begin = w0.z; it.packed = w0.w; it.order = src[1].x;
    }
    clip_one(a, it, shard_id, s_pool, i < n);
  }
}

// ---------------------------------------------------------------------------------------
// bigrec_kernel: one wave per many-tile record of the shards' lists, in two launches.
//   PHASE 0 (cover)   the wave rebuilds the record's edge functions (uniform work, as in the tile kernel), spreads the
//                     tiles of its bounding box over the lanes and, for every tile the triangle covers COMPLETELY (all
//                     three edge functions positive at their smallest corner of the tile), publishes the largest 24-bit
//                     depth its plane has there: tile_cover[bin] = min over such records of {zmax24, list index}.  No key
//                     of that tile can end up with a larger depth, and the winner of the min becomes the tile's initial
//                     depth keys in the tile kernel (evaluated per pixel there; never walked, never appended).
//   PHASE 1 (append)  drops the tiles the triangle does not touch at all -- an edge function is largest at one corner of
//                     the tile's part of the box; not positive there means no pixel centre of that tile is covered (half
//                     of the tiles of a clipped wall's box) -- and the tiles where the record lies entirely behind the
//                     published cover (its smallest depth over the tile's part of its box is larger: it can win no pixel;
//                     the back of a wall, the wall behind it, the robot's far side behind its own arm), and appends the
//                     record to the back of the others' bins.  Up to four slot reservations per lane are in flight before
//                     the first store needs its answer.
// Exactness: a dropped (record, tile) pair could not have won any depth test of that tile (strictly larger 24-bit depth
// than a fragment that is certainly there), so the keys -- and with them the image -- are the same.
// ---------------------------------------------------------------------------------------
constexpr int kBigWavesPerShard = 64;     // (32 / 128 waves per shard, 2 / 8 reservations in flight per lane: no difference)
template <int PHASE>
__global__ __launch_bounds__(256) void bigrec_kernel(SetupArgs a)
{
  const int lane = threadIdx.x & 63;
  const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  const int shard_id = wave % kCounterShards, first = wave / kCounterShards;
  CounterShard& shard = a.counters->shard[shard_id];
  const uint32_t count = shard.big_count;
  if (PHASE == 1 && first == 0 && lane == 0 && count > shard.max_big_fill) shard.max_big_fill = count;      // (this wave alone writes the field)
  const uint32_t n = min(count, a.big_capacity);
  const int tiles = a.tiles_x * a.tiles_y;
  uint32_t mine = 0, dropped = 0;
  for (uint32_t j = (uint32_t)first; j < n; j += kBigWavesPerShard) {
    const uint32_t my_id = (uint32_t)shard_id * a.big_capacity + j;
    const BigRec* rec = a.big_list + my_id;
    PackedTri q;
    {
      const uint4* src = reinterpret_cast<const uint4*>(rec);
      uint4* dst = reinterpret_cast<uint4*>(&q);
      dst[0] = src[0]; dst[1] = src[1];
    }
    const int qslot = (int)rec->slot;
    const TriRec r = unpack_record(q, a.width, a.height);
    const int bx0 = (int)(r.bbx & 0xffff), bx1 = (int)(r.bbx >> 16), by0 = (int)(r.bby & 0xffff), by1 = (int)(r.bby >> 16);
    const int tx0 = bx0 / kTileW, tx1 = bx1 / kTileW;
    const int ty0 = by0 / kTileH, ty1 = by1 / kTileH;
    const int tw = tx1 - tx0 + 1, ntile = tw * (ty1 - ty0 + 1);
    if (PHASE == 0) {
      for (int k = lane; k < ntile; k += 64) {
        const int row = k / tw, tx = tx0 + k - row * tw, ty = ty0 + row;
        // the tile's pixels inside the frame
        const int X0 = tx * kTileW, X1 = min(X0 + kTileW, a.width) - 1, Y0 = ty * kTileH, Y1 = min(Y0 + kTileH, a.height) - 1;
        if (bx0 > X0 || bx1 < X1 || by0 > Y0 || by1 < Y1) continue;
        bool inside = true;
#pragma unroll
        for (int e = 0; e < 3; e++) {
          const int xi = r.A[e] > 0 ? X0 : X1, yi = r.B[e] > 0 ? Y0 : Y1;
          inside = inside && (__mul24(r.A[e], xi) + __mul24(r.B[e], yi) + r.C[e]) > 0;
        }
        if (!inside) continue;
        const float zmax = plane_max(r.a0, r.dzdx, r.dzdy, X0, X1, Y0, Y1);
        if (!(zmax == zmax)) continue;
        const int bin = __mul24(qslot, tiles) + __mul24(ty, a.tiles_x) + tx;
        atomicMin(&a.bin_hdr[bin].cover, ((unsigned long long)z24_of(zmax) << 32) | my_id);
        if (q.order & kNearBit) atomicOr(&a.fbin_count[bin], 0x80000000u);      // (a near cover: the tile's keys carry the float's low bits)
      }
      continue;
    }
    constexpr int kPerRound = 4;
    for (int k0 = lane; k0 < ntile; k0 += 64 * kPerRound) {
      int bin[kPerRound];
      uint32_t pos[kPerRound];
#pragma unroll
      for (int jj = 0; jj < kPerRound; jj++) {
        const int k = k0 + 64 * jj;
        bin[jj] = -1;
        pos[jj] = 0;
        if (k < ntile) {
          const int row = k / tw, tx = tx0 + k - row * tw, ty = ty0 + row;
          // the tile's part of the box, in pixel-centre coordinates
          const int x0 = max(bx0, tx * kTileW), x1 = min(bx1, tx * kTileW + kTileW - 1);
          const int y0 = max(by0, ty * kTileH), y1 = min(by1, ty * kTileH + kTileH - 1);
          bool touches = true;
#pragma unroll
          for (int e = 0; e < 3; e++) {
            const int xa = r.A[e] > 0 ? x1 : x0, ya = r.B[e] > 0 ? y1 : y0;
            touches = touches && (__mul24(r.A[e], xa) + __mul24(r.B[e], ya) + r.C[e]) > 0;
          }
          if (touches) {
            const int b = __mul24(qslot, tiles) + __mul24(ty, a.tiles_x) + tx;
            const unsigned long long cv = a.bin_hdr[b].cover;
            if (cv != kNoCover) {
              // this record IS the tile's cover (it becomes the initial keys), or it lies behind the cover everywhere
              const bool behind = z24_of(plane_min(r.a0, r.dzdx, r.dzdy, x0, x1, y0, y1)) > (uint32_t)(cv >> 32);
              if ((uint32_t)cv == my_id || behind) { touches = false; dropped++; }
            }
            if (touches) {
              bin[jj] = b;
              pos[jj] = atomicAdd(&a.bin_hdr[b].count[1], 1u);          // many-tile records are large: back of the bin
              if (q.order & kNearBit) atomicOr(&a.fbin_count[b], 0x80000000u);
              mine++;
            }
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < kPerRound; jj++)
        if (bin[jj] >= 0 && pos[jj] < a.capacity) store_record(a.bins + (size_t)bin[jj] * a.capacity + (a.capacity - 1u - pos[jj]), q);
    }
  }
  if (PHASE == 0) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mine += (uint32_t)__shfl_down((int)mine, off); dropped += (uint32_t)__shfl_down((int)dropped, off); }
  if (lane == 0 && mine) atomicAdd(&shard.bin_entries, (unsigned long long)mine);
  if (lane == 0 && dropped) atomicAdd(&shard.occluded, (unsigned long long)dropped);
}

// ---------------------------------------------------------------------------------------
// tile_kernel
// ---------------------------------------------------------------------------------------

constexpr unsigned long long kNoFragment = 0x00ffffff00000000ull;   // cleared depth 1.0, order 0
constexpr unsigned long long kResolvedBit = 1ull << 63;


// The key tile's rows may be padded in LDS (RTUF_KEY_PAD keys per row: an A/B switch, 0 in the product -- 2 moves vertically
// adjacent pixels four banks apart; measured).
#ifndef RTUF_KEY_PAD
#define RTUF_KEY_PAD 0
#endif
constexpr int kKeyStride = kTileW + RTUF_KEY_PAD, kKeyCount = kKeyStride * kTileH;
// MODE 0: depth test (atomicMin of {z24, order});  MODE 1: write the exact float z of the
// fragment that won (needed only for window z <= 0.5, where float z is finer than 24 bits).
// Instrumented builds (-DRTUF_COUNT, scripts/overdraw.sh; never the product): every depth test the tile kernel issues and
// every pixel that ends up drawn is counted through two LDS words per workgroup, summed into the batch's counters.
#ifdef RTUF_COUNT
__device__ __forceinline__ uint32_t* count_words() { __shared__ uint32_t s_cnt[2]; return s_cnt; }
#define RTUF_COUNT_TEST() atomicAdd(&count_words()[0], 1u)
#else
#define RTUF_COUNT_TEST() ((void)0)
#endif

// Depth keys: {z24, draw order << shift | low bits of the fragment's float z}.  The draw order takes as many of the low
// word's bits as the context's triangle count needs (shift = 32 - those), the rest carries the LOW BITS OF THE FLOAT Z in
// tiles that hold near geometry: the shader sees the unquantised gl_FragCoord.z, which below window z 0.5 is finer than
// the 24-bit depth that decides the test -- but z24 pins z to an interval of 2^-24, and the float's low `shift` bits single
// out one float in it as long as the interval holds fewer than 2^(shift-1) floats, i.e. for z24 >= 2^(26-shift)
// (near_z_from_key; checked exhaustively on the CPU, tests/test_near_keys_cpu.py).  The low bits sit below the draw
// order, and a triangle meets a pixel once, so they never decide a comparison: atomicMin still is "nearest 24-bit depth,
// first drawn wins ties".  Round 3 re-walked every near record of such a tile to fetch the winners' float z (24,048 of
// 38,400 tiles with the arm in front of the lens); now only winners closer than z24 < zexact still need that pass
// (15 um in front of the near plane for the 250 k-triangle model).
struct KeyFmt {
  int shift;            // draw order << shift in the key's low word (TileArgs::key_shift)
  uint32_t lowmask;     // (1 << shift) - 1 in tiles that hold near records (or a near cover), else 0
  uint32_t zexact;      // winners with z24 < zexact need the exact-z pass: 2^(26-shift) in near tiles, else 2^23 + 1
  uint32_t abl;         // RTUF_ABLATE builds: the launch's timing-experiment bits (0x4000: depth tests without their LDS atomic, 0x2000000: strips fetched but not walked)
};

// float z of a fragment from its 24-bit depth (zexact <= z24 <= 2^23) and the low `shift` bits of its float
__device__ __forceinline__ float near_z_from_key(uint32_t z24, uint32_t low, int shift)
{
  const uint32_t cb = __float_as_uint(__fmul_rn((float)z24, 5.9604648328104515e-08f));      // about the middle of z24's interval
  const uint32_t span = 1u << shift;
  uint32_t cand = (cb & ~(span - 1u)) | low;
  const int d = (int)(cand - cb);
  const int half = (int)(span >> 1);
  cand = d > half ? cand - span : (d < -half ? cand + span : cand);
  return __uint_as_float(cand);
}

// One depth test of a fragment whose window z is already evaluated (`order`: the draw order shifted into place).
// Issue classes (profiles/valu_peak.json: gfx950 issues v_fma/mul/add_f32, v_add/sub_u32, v_and/or/xor_b32 and the right shifts
// in 2 cycles per wave64, everything else -- conversions, v_rndne, min/max, compares, selects, 24-bit multiplies, left shifts --
// in 4): RTUF_FAST_CLASS = 1 takes the 4-cycle instructions out of the hot walks where a 2-cycle one computes the same number
// (0: A/B switch, the code of rounds 1-5).
template <int MODE, bool LOW>
__device__ __forceinline__ void depth_test(unsigned long long* keys, uint32_t order, float z, int lidx, const KeyFmt& kf, int lc)
{
  if (MODE == 0) RTUF_COUNT_TEST();
  if (MODE == 0) RTUF_LANES(lc, true);
  const uint32_t lo = LOW ? (order | (__float_as_uint(z) & kf.lowmask)) : order;
  const unsigned long long key = ((unsigned long long)((RTUF_FAST_CLASS && !LOW && MODE == 0) ? z24_of_upper_half(z) : z24_of(z)) << 32) | lo;
  if (RTUF_ABL(kf.abl, 0x4000u)) { if (key == 0x0123456789abcdefull) keys[lidx] = key; return; }      // timing experiment: everything but the atomic
  if (MODE == 0) {
    atomicMin(&keys[lidx], key);
  } else {
    if (keys[lidx] == key) keys[lidx] = kResolvedBit | (unsigned long long)__float_as_uint(z);
  }
}

template <int MODE, bool LOW>
__device__ __forceinline__ void fragment(unsigned long long* keys, const TriRec& r, int px, int py, int lidx, const KeyFmt& kf, int lc = kLaneWalkFrag)
{
  // (r.order holds the draw order already shifted into place)
  depth_test<MODE, LOW>(keys, r.order, __fmaf_rn(r.dzdy, (float)py, __fmaf_rn(r.dzdx, (float)px, r.a0)), lidx, kf, lc);
}


#ifndef RTUF_SMALL_AREA
#define RTUF_SMALL_AREA 96
#endif
#ifndef RTUF_QUARTER_AREA
#define RTUF_QUARTER_AREA 256
#endif
constexpr int kSmallArea = RTUF_SMALL_AREA;     // bounding boxes up to this many pixels are walked by their own lane
#ifndef RTUF_PARK_BELOW
#define RTUF_PARK_BELOW 128
#endif
#ifndef RTUF_WALL_AREA
#define RTUF_WALL_AREA 1536
#endif
constexpr int kWallArea = RTUF_WALL_AREA;         // ... larger bins only for boxes covering more of the tile than this
constexpr int kParkBelow = RTUF_PARK_BELOW;       // bins of at most this many records use the cooperative whole-tile pass
constexpr int kHugeMax = 127;
// Exact-z pass: which draw-order keys won a pixel that needs its exact float z -- a 4096-bit filter in LDS (one multiplicative
// hash); records whose key is not in it cannot be a winner and are not walked a second time.
constexpr int kWinnerWords = 128;
__device__ __forceinline__ uint32_t winner_slot(uint32_t order) { return (order * 0x9E3779B1u) >> 20; }      // 12 bits
constexpr int kQuarterArea = RTUF_QUARTER_AREA;   // up to this many by a quarter wave (4 triangles at a time), larger by the whole wave

// The tile's part of a record's box (local lx0..ly1) against the three edges: an edge function is largest /
// smallest at a corner.  Not positive at its largest for some edge: no pixel centre of this tile is covered
// (half of the tiles of a screen-filling triangle's box) -> returns 0.  Positive at its smallest for all
// three: every candidate is covered -> returns 2 (no edge tests needed).  Otherwise 1.
__device__ __forceinline__ int classify_box(const TriRec& r, int x_base, int y_base, int lx0, int lx1, int ly0, int ly1)
{
  bool reject = false, inside = true;
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const int xa = x_base + (r.A[e] > 0 ? lx1 : lx0), xi = x_base + (r.A[e] > 0 ? lx0 : lx1);
    const int ya = y_base + (r.B[e] > 0 ? ly1 : ly0), yi = y_base + (r.B[e] > 0 ? ly0 : ly1);
    reject = reject || (__mul24(r.A[e], xa) + __mul24(r.B[e], ya) + r.C[e]) <= 0;
    inside = inside && (__mul24(r.A[e], xi) + __mul24(r.B[e], yi) + r.C[e]) > 0;
  }
  return reject ? 0 : (inside ? 2 : 1);
}

// Broadcast of one lane's record to the whole wave through scalar registers (v_readlane): the
// cooperative path then runs with the triangle's 16 words as SGPR operands.
__device__ __forceinline__ TriRec broadcast_record(const TriRec& r, int src_lane)
{
  TriRec q;
  const int* s = reinterpret_cast<const int*>(&r);
  int* d = reinterpret_cast<int*>(&q);
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = __builtin_amdgcn_readlane(s[k], src_lane);
  return q;
}

// One lane's step of the cooperative paths: pixel (lx, ly) of the tile and the one below it (limited to the
// box, qy1 inclusive).  Neighbouring lanes take neighbouring columns, so each of the two LDS atomics of a
// step is conflict-free across the wave; the lower pixel's edge values are the upper one's plus B.
template <int MODE, bool LOW>
__device__ __forceinline__ void raster_pair(unsigned long long* keys, const TriRec& q, int x_base, int y_base, int lx, int ly, int qy1, const KeyFmt& kf, int lc = kLaneQuarterFrag)
{
  const int px = x_base + lx, py = y_base + ly;
  const int e0 = __mul24(q.A[0], px) + __mul24(q.B[0], py) + q.C[0];
  const int e1 = __mul24(q.A[1], px) + __mul24(q.B[1], py) + q.C[1];
  const int e2 = __mul24(q.A[2], px) + __mul24(q.B[2], py) + q.C[2];
  const int lidx = ly * kKeyStride + lx;
  if (min(e0, min(e1, e2)) > 0) fragment<MODE, LOW>(keys, q, px, py, lidx, kf, lc);
  if (min(e0 + q.B[0], min(e1 + q.B[1], e2 + q.B[2])) > 0 && ly < qy1) fragment<MODE, LOW>(keys, q, px, py + 1, lidx + kKeyStride, kf, lc);
}

// Row-stepped walk of a strip -- columns qx0 .. qx0 + wc - 1 (wc <= 2^team_log) of the record's part of the tile, rows qy0 ..
// qy0 + h - 1 -- by a TEAM of 2^team_log lanes (`sub` = the lane's number in its team).  The team forms 2^plog columns (the
// power of two that holds wc) x 2^lr rows; every lane keeps its column and steps DOWN the rows, 2^lr at a time, two steps per
// trip.  Nothing is evaluated per candidate but what changes: the three edge values move by B << lr per step (integer adds
// modulo 2^32 of the same integers A*px + B*py + C is made of -- the value at a pixel is the same number whichever way it is
// reached), the plane's column term fma(dzdx, px, a0) is the lane's constant and the row's float coordinate moves by 2^lr (small
// integers: exact), so fma(dzdy, py, .) sees the operands fragment() gives it.  A trip costs 16 VALU instructions for 2 x 2^team_log
// candidates where the linear run of pairs it replaces (raster_pair: index -> column / row by a reciprocal multiply, six 24-bit
// multiplies per pair) took 42.  EDGES = false: the caller knows that every candidate is covered (classify_box == 2).
#ifndef RTUF_STRIP_WALK
#define RTUF_STRIP_WALK 2          // (A/B switch -- 0: quarter-wave and whole-wave walks as linear runs of candidate pairs, as up to round 4; 2: strips only in tiles with near geometry)
#endif
template <int MODE, bool LOW, bool EDGES>
__device__ __forceinline__ void strip_walk(unsigned long long* keys, const TriRec& q, int x_base, int y_base, int qx0, int wc, int qy0, int h,
                                           int team_log, int sub, bool on, const KeyFmt& kf, int lc_trip, int lc_frag)
{
  const int plog = wc > 1 ? 32 - __clz(wc - 1) : 0, lr = team_log - plog;
  const int col = sub & ((1 << plog) - 1), row0 = sub >> plog;
  // rows row0, row0 + 2^lr, ... below h (row0 < 2^lr: the numerator is never negative)
  int nrow = (on && col < wc) ? (h - 1 - row0 + (1 << lr)) >> lr : 0;
  const int px = x_base + qx0 + col, py = y_base + qy0 + row0;
  int e0 = 1, e1 = 1, e2 = 1, d0 = 0, d1 = 0, d2 = 0;
  if (EDGES) {
    e0 = __mul24(q.A[0], px) + __mul24(q.B[0], py) + q.C[0];
    e1 = __mul24(q.A[1], px) + __mul24(q.B[1], py) + q.C[1];
    e2 = __mul24(q.A[2], px) + __mul24(q.B[2], py) + q.C[2];
    d0 = (int)((uint32_t)q.B[0] << lr); d1 = (int)((uint32_t)q.B[1] << lr); d2 = (int)((uint32_t)q.B[2] << lr);
  }
  const float zc = __fmaf_rn(q.dzdx, (float)px, q.a0);
  float fy = (float)py;
  const float dfy = (float)(1 << lr);
  int lidx = (qy0 + row0) * kKeyStride + qx0 + col;
  const int dl = kKeyStride << lr;
  while (__ballot(nrow > 0)) {
    if (MODE == 0) RTUF_LANES(lc_trip, nrow > 0);
    const int f0 = e0 + d0, f1 = e1 + d1, f2 = e2 + d2;
    if (nrow > 0 && (!EDGES || min(e0, min(e1, e2)) > 0)) depth_test<MODE, LOW>(keys, q.order, __fmaf_rn(q.dzdy, fy, zc), lidx, kf, lc_frag);
    if (nrow > 1 && (!EDGES || min(f0, min(f1, f2)) > 0)) depth_test<MODE, LOW>(keys, q.order, __fmaf_rn(q.dzdy, __fadd_rn(fy, dfy), zc), lidx + dl, kf, lc_frag);
    e0 = f0 + d0; e1 = f1 + d1; e2 = f2 + d2;
    fy = __fadd_rn(fy, __fadd_rn(dfy, dfy));
    lidx += 2 * dl;
    nrow -= 2;
  }
}

// Rasterises the bin's records into the LDS key tile.  Every wave works on the records it loaded:
// lane-per-triangle for small bounding boxes, a quarter wave each for the middle class; triangles
// that cover a large part of the tile are parked and, after one workgroup barrier at the end, walked
// by all waves together (all threads of the workgroup must call this function).
// `zcover` is the tile's occlusion bound (24-bit depth no key of the tile can exceed; 0xffffffff = none, wave-uniform): records
// that lie behind it over their whole part of the tile are dropped when they are loaded.  MODE 1 looks only at records the
// set-up marked as near (kNearBit) and, of those, only at the ones that can reach the lower half of the depth range here.

template <int MODE, bool LOW, int NT>
__device__ __forceinline__ void raster_bin(unsigned long long* keys, const PackedTri* recs, uint32_t n,
                                           int x_base, int y_base, int tid, bool dbg_load_only, int width, int height, uint32_t n_front, uint32_t capacity,
                                           uint32_t* s_huge, TriRec* s_prec, uint4* s_pmeta, uint32_t zcover, const uint32_t* s_winners, const KeyFmt& kf,
                                           uint4 first0, uint4 first1, bool have_first, int dbg_skip = 0)
{
  const int lane = tid & 63;
  const uint32_t zdrop = MODE == 1 ? min(zcover, kf.zexact - 1u) : zcover;      // MODE 1: only what can reach a depth that needs the pass
  // (round 5 measured two ways of balancing the walks here and took both out again: windows sorted by box size, branch
  // sorted-walk-experiment -- fewer trips, slower for its barriers; records behind the cover compacted through a per-wave queue
  // before the unpack, branch cover-compaction-experiment -- fuller waves, slower for its second load.  DESIGN.md A.5)
  for (uint32_t base = 0; base < n; base += NT) {
    const uint32_t i = base + tid;
    bool have = i < n;
    const uint32_t ri = i < n_front ? i : capacity - 1u - (i - n_front);      // small boxes from the front, larger ones from the back
    const bool first_in_regs = have_first && base == 0u;
    TriRec r;
    int lx0 = 0, lx1 = -1, ly0 = 0, ly1 = -1;
    PackedTri pk;
    if (MODE == 0) RTUF_LANES(kLaneTileLoad, have);
    if (have) {
      uint4* dst = reinterpret_cast<uint4*>(&pk);
      if (first_in_regs) {
        dst[0] = first0; dst[1] = first1;      // (the caller asked for this lane's first record before it initialised the key tile)
      } else {
        const uint4* src = reinterpret_cast<const uint4*>(recs + ri);
        dst[0] = src[0]; dst[1] = src[1];
      }
      if (MODE == 1) {          // only near records whose draw-order key won a pixel that needs resolving
        const uint32_t h = winner_slot(pk.order & kOrderMask);
        if (!(pk.order & kNearBit) || !((s_winners[h >> 5] >> (h & 31u)) & 1u)) have = false;
      }
    }
    if (have) {
      r = unpack_record(pk, width, height);
      r.order <<= kf.shift;
      lx0 = max((int)(r.bbx & 0xffff) - x_base, 0);
      lx1 = min((int)(r.bbx >> 16) - x_base, kTileW - 1);
      ly0 = max((int)(r.bby & 0xffff) - y_base, 0);
      ly1 = min((int)(r.bby >> 16) - y_base, kTileH - 1);
    } else {
#pragma unroll
      for (int q = 0; q < 16; q++) reinterpret_cast<int*>(&r)[q] = 0;
    }
    const int w = lx1 - lx0 + 1, h = ly1 - ly0 + 1;
    int area = (have && w > 0 && h > 0) ? w * h : 0;
    if (zdrop != 0xffffffffu) {               // (uniform) behind the tile's cover / out of the exact-z range: cannot matter here
      if (area > 0 && z24_of(plane_min(r.a0, r.dzdx, r.dzdy, x_base + lx0, x_base + lx1, y_base + ly0, y_base + ly1)) > zdrop) area = 0;
    }
    if (dbg_load_only) { if (r.order == 0xdeadbeefu) keys[0] = 0; area = 0; }
    if (dbg_skip == 1 && area <= kSmallArea) area = 0;      // timing experiment: no lane-per-triangle walk
    if (dbg_skip == 2 && area > kSmallArea) area = 0;       // timing experiment: no quarter-wave walk
    if (dbg_skip == 3 && area > kQuarterArea) area = 0;     // timing experiment: no whole-tile walks
    const bool small = area > 0 && area <= kSmallArea;
#ifdef RTUF_LANECOUNT
    if (MODE == 0 && have) {
      const int td = ((lx1 - lx0 + 2) >> 1) * ((ly1 - ly0 + 2) >> 1);
      const int b = area <= 0 ? kLaneHistNone : (area > kQuarterArea ? kLaneHistLarge : (area > kSmallArea ? kLaneHistQuarter :
                    kLaneHist + (td <= 6 ? td - 1 : (td <= 8 ? 6 : (td <= 12 ? 7 : (td <= 16 ? 8 : (td <= 24 ? 9 : 10)))))));
      atomicAdd(&lane_words()[2 * b + 1], 1u);
      const int fw = (int)(r.bbx >> 16) - (int)(r.bbx & 0xffff) + 1, fh = (int)(r.bby >> 16) - (int)(r.bby & 0xffff) + 1;
      const bool one_tile = (int)(r.bbx & 0xffff) / kTileW == (int)(r.bbx >> 16) / kTileW && (int)(r.bby & 0xffff) / kTileH == (int)(r.bby >> 16) / kTileH;
      if (one_tile && fw <= 8 && fh <= 8 && !(pk.order & kNearBit)) atomicAdd(&lane_words()[2 * kLaneHist8x8 + 1], 1u);
      if (one_tile && ((fw <= 8 && fh <= 4) || (fw <= 4 && fh <= 8)) && !(pk.order & kNearBit)) atomicAdd(&lane_words()[2 * kLaneHist8x4 + 1], 1u);
    }
#endif
    // lane-per-triangle: walk the bounding box as one run of 2x2 candidate QUADS, quad row by quad row.
    // The three edge values of a quad's upper left pixel are stepped incrementally (one add each, a
    // different step at the end of a quad row); the other three pixels' are those plus A, B, A + B -- the
    // same integer arithmetic as evaluating A*px + B*py + C at every candidate, at about half the
    // instructions per candidate and a quarter of the loop trips.
    if constexpr (RTUF_FAST_CLASS != 0) {
      // The same walk with the pixel position kept as two floats (small integers: every add is exact) -- the plane wants them
      // as floats, and four conversions per trip were 4-cycle instructions; the key's LDS address is stepped in bytes (no
      // left shift per trip); the right-hand pixels of a quad lie outside the box only in the last quad column of a box of odd
      // width, which the wrap test already knows (one compare less).  Same integers, same floats, same order of depth tests.
      const int qcols = (lx1 - lx0 + 2) >> 1;                              // quads per quad row
      const int back = 2 * (qcols - 1);                                    // x distance from the last quad of a row to the first
      const float fpx0 = (float)(x_base + lx0), fpy_last = (float)(y_base + ly1), fpx_lastq = (float)(x_base + lx0 + back);
      float fpx = fpx0, fpy = (float)(y_base + ly0);
      uint32_t kofs = (uint32_t)(ly0 * kKeyStride + lx0) * 8u;             // byte offset of the quad's upper left key
      int e0 = __mul24(r.A[0], x_base + lx0) + __mul24(r.B[0], y_base + ly0) + r.C[0];
      int e1 = __mul24(r.A[1], x_base + lx0) + __mul24(r.B[1], y_base + ly0) + r.C[1];
      int e2 = __mul24(r.A[2], x_base + lx0) + __mul24(r.B[2], y_base + ly0) + r.C[2];
      // a step to the right always; the lanes at the end of a quad row then add what takes them to the start of the next one
      // (five 2-cycle adds under the wrap's lane mask instead of six selects)
      const int a0x2 = 2 * r.A[0], a1x2 = 2 * r.A[1], a2x2 = 2 * r.A[2];
      const int d0 = 2 * r.B[0] - __mul24(back + 2, r.A[0]), d1 = 2 * r.B[1] - __mul24(back + 2, r.A[1]), d2 = 2 * r.B[2] - __mul24(back + 2, r.A[2]);
      const uint32_t dk = (uint32_t)(2 * kKeyStride - back - 2) * 8u;
      const bool odd_w = ((lx1 - lx0) & 1) == 0;                           // the last quad column holds one pixel column
      int todo = small ? __mul24(qcols, (ly1 - ly0 + 2) >> 1) : 0;
      auto test = [&](float x, float y, uint32_t ofs) {
        depth_test<MODE, LOW>(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(keys) + ofs), r.order,
                              __fmaf_rn(r.dzdy, y, __fmaf_rn(r.dzdx, x, r.a0)), 0, kf, kLaneWalkFrag);
      };
      while (__ballot(todo > 0)) {
        if (MODE == 0) RTUF_LANES(kLaneWalkTrip, todo > 0);
        if (todo > 0) {
          const bool wrap = fpx == fpx_lastq;
          const bool no_right = wrap && odd_w, below = fpy < fpy_last;
          const int f0 = e0 + r.B[0], f1 = e1 + r.B[1], f2 = e2 + r.B[2];
          const float fx1 = __fadd_rn(fpx, 1.0f), fy1 = __fadd_rn(fpy, 1.0f);
          if (min(e0, min(e1, e2)) > 0) test(fpx, fpy, kofs);
          if (min(e0 + r.A[0], min(e1 + r.A[1], e2 + r.A[2])) > 0 && !no_right) test(fx1, fpy, kofs + 8u);
          if (min(f0, min(f1, f2)) > 0 && below) test(fpx, fy1, kofs + 8u * kKeyStride);
          if (min(f0 + r.A[0], min(f1 + r.A[1], f2 + r.A[2])) > 0 && !no_right && below) test(fx1, fy1, kofs + 8u * kKeyStride + 8u);
          e0 += a0x2; e1 += a1x2; e2 += a2x2;
          kofs += 16u;
          fpx = __fadd_rn(fpx, 2.0f);
          todo--;
          if (wrap) {
            asm volatile("" ::: "memory");      // (keeps this a masked block: as selects it is six 4-cycle instructions)
            e0 += d0; e1 += d1; e2 += d2;
            kofs += dk;
            fpy = __fadd_rn(fpy, 2.0f);
            fpx = fpx0;
          }
        }
      }
    } else
    {
      const int px0 = x_base + lx0, px1 = x_base + lx1, py_last = y_base + ly1;
      int px = px0, py = y_base + ly0, lidx = ly0 * kKeyStride + lx0;
      int e0 = __mul24(r.A[0], px) + __mul24(r.B[0], py) + r.C[0];
      int e1 = __mul24(r.A[1], px) + __mul24(r.B[1], py) + r.C[1];
      int e2 = __mul24(r.A[2], px) + __mul24(r.B[2], py) + r.C[2];
      const int qcols = (lx1 - lx0 + 2) >> 1;                              // quads per quad row
      const int back = 2 * (qcols - 1);                                    // x distance from the last quad of a row to the first
      const int s0 = 2 * r.B[0] - __mul24(back, r.A[0]), s1 = 2 * r.B[1] - __mul24(back, r.A[1]), s2 = 2 * r.B[2] - __mul24(back, r.A[2]);
      const int a0x2 = 2 * r.A[0], a1x2 = 2 * r.A[1], a2x2 = 2 * r.A[2];
      const int row_step = 2 * kKeyStride - back;
      const int px_lastq = px0 + back;
      int todo = small ? __mul24(qcols, (ly1 - ly0 + 2) >> 1) : 0;
      while (__ballot(todo > 0)) {
        if (MODE == 0) RTUF_LANES(kLaneWalkTrip, todo > 0);
        if (todo > 0) {
          const bool right = px < px1, below = py < py_last;
          const int f0 = e0 + r.B[0], f1 = e1 + r.B[1], f2 = e2 + r.B[2];
          if (min(e0, min(e1, e2)) > 0) fragment<MODE, LOW>(keys, r, px, py, lidx, kf);
          if (min(e0 + r.A[0], min(e1 + r.A[1], e2 + r.A[2])) > 0 && right) fragment<MODE, LOW>(keys, r, px + 1, py, lidx + 1, kf);
          if (min(f0, min(f1, f2)) > 0 && below) fragment<MODE, LOW>(keys, r, px, py + 1, lidx + kKeyStride, kf);
          if (min(f0 + r.A[0], min(f1 + r.A[1], f2 + r.A[2])) > 0 && right && below) fragment<MODE, LOW>(keys, r, px + 1, py + 1, lidx + kKeyStride + 1, kf);
          const bool wrap = px == px_lastq;
          e0 += wrap ? s0 : a0x2;
          e1 += wrap ? s1 : a1x2;
          e2 += wrap ? s2 : a2x2;
          lidx += wrap ? row_step : 2;
          py += wrap ? 2 : 0;
          px = wrap ? px0 : px + 2;
          todo--;
        }
      }
    }
    // quarter-wave cooperative: four triangles at a time, 16 lanes each, one vertical candidate pair per
    // lane and step (the bounding box is walked as a linear run of pairs, so slivers waste little)
    // whole-wave cooperative: triangles that cover a large part of the tile, one at a time with the
    // record in scalar registers (v_readlane), 64 candidate pairs per step
    unsigned long long huge = __ballot(area > kQuarterArea);
    if (huge) {
      // those whose triangle misses this tile altogether drop out here
      huge &= ~__ballot(area > kQuarterArea && classify_box(r, x_base, y_base, lx0, lx1, ly0, ly1) == 0);
    }
    const unsigned long long park = huge ? (huge & __ballot(area > (int)s_huge[1 + kHugeMax])) : 0ull;
    if (park) {
      // In a bin that one or two waves hold alone (typical: a few walls) all of them, in fuller bins only those
      // that cover nearly the whole tile (the kernel puts the area threshold behind the list) are parked in
      // the workgroup's list (s_huge[0] = count, then bin indices): after this loop all four waves rasterise
      // each of them together.  Otherwise every wave has its own, and walking them alone avoids paying each
      // triangle's fixed cost four times.  What does not fit the list stays with this wave, too.
      const int leader = __ffsll((long long)park) - 1;
      uint32_t hb = 0;
      if (lane == leader) hb = atomicAdd(&s_huge[0], (uint32_t)__popcll(park));
      hb = (uint32_t)__builtin_amdgcn_readlane((int)hb, leader);
      const uint32_t hs = hb + (uint32_t)__popcll(park & ((1ull << lane) - 1ull));
      const bool parked = ((park >> lane) & 1ull) != 0 && hs < (uint32_t)kHugeMax;
      if (parked) s_huge[1 + hs] = ri;
      huge &= ~__ballot(parked);
    }
    if constexpr (RTUF_STRIP_WALK == 1 || (RTUF_STRIP_WALK == 2 && (LOW || MODE == 1))) {
    // Everything beyond the lane-per-triangle class that stays with this wave: STRIPS, one per team of lanes.  A strip is up to
    // 16 columns of a record's part of the tile over all its rows (a wider record is dealt out as two to four strips, to
    // neighbouring teams or successive rounds); its lanes step down the rows (strip_walk).  With three and more records waiting
    // a round deals four strips to the four quarters of the wave; the last two share the wave half and half, a single one gets
    // all 64 lanes (16 columns x 4 rows per step).  The record travels from the lane that loaded it with 15 shuffles.
    // Taken in tiles with near geometry only (RTUF_STRIP_WALK = 2: the LOW instance and the exact-z pass), where such records
    // come by the dozen per wave-load and the rounds are full: tile kernel 1.117 -> 1.07 ms with the arm in front of the lens.
    // Elsewhere a wave-load holds one such record in four, a round is one strip, and the pair runs below measure the same or
    // better (C3 +0.5 %, C4 / C5 shares +-0.5 %: profiles/r05_experiment_strips.txt).
    {
      unsigned long long big = __ballot(area > kSmallArea && (area <= kQuarterArea || ((huge >> lane) & 1ull) != 0ull));
      const int nstrip = (lx1 - lx0 + 16) >> 4;
      int head_strip = 0;                      // (scalar) the next strip of the record at the head of `big`
      while (big) {
        const int waiting = __popcll(big);
        const int team_log = waiting >= 3 ? 4 : (waiting == 2 ? 5 : 6);      // (scalar)
        const int grp = lane >> team_log, sub = lane & ((1 << team_log) - 1);
        int src = -1, strip = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (g < (64 >> team_log)) {
            const int sl = big ? __ffsll((long long)big) - 1 : -1;
            const int si = head_strip;
            if (big) {
              if (++head_strip >= __builtin_amdgcn_readlane(nstrip, sl)) { big &= big - 1; head_strip = 0; }
            }
            if (g == grp) { src = sl; strip = si; }
          }
        }
        const int srcl = src < 0 ? lane : src;
        TriRec q;
        {
          const int* sp = reinterpret_cast<const int*>(&r);
          int* dp = reinterpret_cast<int*>(&q);
#pragma unroll
          for (int k = 0; k < 15; k++) dp[k] = __shfl(sp[k], srcl);
        }
        const int qx0 = max((int)(q.bbx & 0xffff) - x_base, 0) + 16 * strip, qx1 = min((int)(q.bbx >> 16) - x_base, kTileW - 1);
        const int qy0 = max((int)(q.bby & 0xffff) - y_base, 0), qy1 = min((int)(q.bby >> 16) - y_base, kTileH - 1);
        if (RTUF_ABL(kf.abl, 0x2000000u)) { if (q.order == 0xdeadbeefu) keys[0] = 0; continue; }      // timing experiment: strips dealt out and fetched, not walked
        strip_walk<MODE, LOW, true>(keys, q, x_base, y_base, qx0, min(qx1 - qx0 + 1, 16), qy0, qy1 - qy0 + 1, team_log, sub, src >= 0, kf, kLaneQuarterTrip, kLaneQuarterFrag);
      }
    }
    } else {
    const int grp = lane >> 4, sub = lane & 15;
    while (huge) {
      const int src = __ffsll((long long)huge) - 1;
      huge &= huge - 1;
      const TriRec q = broadcast_record(r, src);
      const int qx0 = max((int)(q.bbx & 0xffff) - x_base, 0), qx1 = min((int)(q.bbx >> 16) - x_base, kTileW - 1);
      const int qy0 = max((int)(q.bby & 0xffff) - y_base, 0), qy1 = min((int)(q.bby >> 16) - y_base, kTileH - 1);
      // the box as a linear run of vertical pixel pairs, one pair per lane and step
      const int qw = qx1 - qx0 + 1, npair = qw * ((qy1 - qy0 + 2) >> 1);
      const uint32_t inv = (uint32_t)ceilf(1048576.0f * __builtin_amdgcn_rcpf((float)qw));
      for (int idx = lane; idx < npair; idx += 64) {
        if (MODE == 0) RTUF_LANES(kLaneWaveTrip, true);
        const int yy = (int)((uint32_t)__mul24(idx, (int)inv) >> 20);
        raster_pair<MODE, LOW>(keys, q, x_base, y_base, qx0 + idx - __mul24(yy, qw), qy0 + 2 * yy, qy1, kf, kLaneWaveFrag);
      }
    }
    unsigned long long big = __ballot(area > kSmallArea && area <= kQuarterArea);
    while (big) {
      int src = -1;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int sl = big ? __ffsll((long long)big) - 1 : -1;
        if (big) big &= big - 1;
        if (g == grp) src = sl;
      }
      const int srcl = src < 0 ? lane : src;
      TriRec q;
      {
        const int* sp = reinterpret_cast<const int*>(&r);
        int* dp = reinterpret_cast<int*>(&q);
#pragma unroll
        for (int k = 0; k < 15; k++) dp[k] = __shfl(sp[k], srcl);
      }
      const int qx0 = max((int)(q.bbx & 0xffff) - x_base, 0), qx1 = min((int)(q.bbx >> 16) - x_base, kTileW - 1);
      const int qy0 = max((int)(q.bby & 0xffff) - y_base, 0), qy1 = min((int)(q.bby >> 16) - y_base, kTileH - 1);
      const int qw = qx1 - qx0 + 1;
      const int npair = src < 0 ? 0 : qw * ((qy1 - qy0 + 2) >> 1);
      // idx / qw for idx < 4096, qw <= 128 via a reciprocal multiply that is exact in that range:
      // inv = ceil(2^20 / qw) (+1 at most if v_rcp_f32 is an ulp high), and idx * (inv - 2^20/qw) * qw < 2^20
      // (checked exhaustively on the CPU, including +-1 ulp of the reciprocal)
      const uint32_t inv = (uint32_t)ceilf(1048576.0f * __builtin_amdgcn_rcpf((float)max(qw, 1)));
      for (int idx = sub; __ballot(idx < npair); idx += 16) {
        if (MODE == 0) RTUF_LANES(kLaneQuarterTrip, idx < npair);
        if (idx < npair) {
          const int yy = (int)((uint32_t)__mul24(idx, (int)inv) >> 20);
          raster_pair<MODE, LOW>(keys, q, x_base, y_base, qx0 + idx - __mul24(yy, qw), qy0 + 2 * yy, qy1, kf);
        }
      }
    }
    }
  }
  // workgroup-cooperative: the parked triangles (those that cover a large part of the tile: walls, close
  // links), one at a time, the box's run of candidate pairs spread over all four waves.  ONE wave loads, unpacks
  // and classifies a batch of them (a lane each) and leaves the records and what the others need to know in LDS; the
  // other three waves wait at the barrier instead of repeating the same two hundred instructions (with ten wall
  // records in every tile of a 720p frame that repetition was 30 % of the walls' cost).
  __syncthreads();
  const uint32_t nh = min(s_huge[0], (uint32_t)kHugeMax);
  uint32_t zfull = zcover;                       // largest depth any pixel of the tile can still have (24-bit)
  for (uint32_t hb = 0; hb < nh; hb += 64) {
    const bool have = hb + (uint32_t)lane < nh;
    if (tid < 64) {
      TriRec r;
      if (have) {
        PackedTri pk;
        const uint4* src = reinterpret_cast<const uint4*>(recs + s_huge[1 + hb + lane]);
        uint4* dst = reinterpret_cast<uint4*>(&pk);
        dst[0] = src[0]; dst[1] = src[1];
        r = unpack_record(pk, width, height);
        r.order <<= kf.shift;
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) reinterpret_cast<int*>(&r)[k] = 0;
      }
      // per lane: the record's part of the tile, whether it is covered entirely, and the range of its depth
      // there.  The plane is evaluated exactly like fragment() does, which is monotonic in px and in py, so
      // the extremes over the box are at its corners.
      const int lx0 = max((int)(r.bbx & 0xffff) - x_base, 0), lx1 = min((int)(r.bbx >> 16) - x_base, kTileW - 1);
      const int ly0 = max((int)(r.bby & 0xffff) - y_base, 0), ly1 = min((int)(r.bby >> 16) - y_base, kTileH - 1);
      const int cls = have ? classify_box(r, x_base, y_base, lx0, lx1, ly0, ly1) : 0;
      const float cx0 = __fmaf_rn(r.dzdx, (float)(x_base + lx0), r.a0), cx1 = __fmaf_rn(r.dzdx, (float)(x_base + lx1), r.a0);
      const float fy0 = (float)(y_base + ly0), fy1 = (float)(y_base + ly1);
      const float z00 = __fmaf_rn(r.dzdy, fy0, cx0), z10 = __fmaf_rn(r.dzdy, fy0, cx1);
      const float z01 = __fmaf_rn(r.dzdy, fy1, cx0), z11 = __fmaf_rn(r.dzdy, fy1, cx1);
      const uint32_t zmin24 = z24_of(fminf(fminf(z00, z10), fminf(z01, z11))), zmax24 = z24_of(fmaxf(fmaxf(z00, z10), fmaxf(z01, z11)));
      const bool whole = cls == 2 && lx0 == 0 && ly0 == 0 && lx1 == min(kTileW, width - x_base) - 1 && ly1 == min(kTileH, height - y_base) - 1;
      if (have) {
        const uint4* sp = reinterpret_cast<const uint4*>(&r);
        uint4* dp = reinterpret_cast<uint4*>(&s_prec[lane]);
        dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; dp[3] = sp[3];
        s_pmeta[lane] = make_uint4((uint32_t)cls, zmin24, whole ? zmax24 : 0xffffffffu, 0u);
      }
    }
    __syncthreads();
    uint4 meta = make_uint4(0u, 0xffffffffu, 0xffffffffu, 0u);
    if (have) meta = s_pmeta[lane];
    const uint32_t zmin24 = meta.y;
    // Occlusion among them: once a triangle covers every pixel of the tile, no pixel's key can have a
    // larger depth than that triangle's largest; a triangle whose smallest depth here is larger still
    // (the back of a wall, the wall behind it) cannot win anywhere in this tile and is skipped.
    uint32_t zf = meta.z;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zf = min(zf, (uint32_t)__shfl_xor((int)zf, o));
    zfull = min(zfull, zf);
    unsigned long long m = __ballot(have && zmin24 <= zfull);
    const unsigned long long full = __ballot(have && meta.x == 2u);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const bool inside = ((full >> src) & 1ull) != 0;
      TriRec q;                                        // the record from LDS into scalar registers (same address in every lane)
      {
        const uint4* sp = reinterpret_cast<const uint4*>(&s_prec[src]);
        uint4 w[4] = {sp[0], sp[1], sp[2], sp[3]};
        const int* wi = reinterpret_cast<const int*>(w);
        int* d = reinterpret_cast<int*>(&q);
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = __builtin_amdgcn_readfirstlane(wi[k]);
      }
      const int qx0 = max((int)(q.bbx & 0xffff) - x_base, 0), qx1 = min((int)(q.bbx >> 16) - x_base, kTileW - 1);
      const int qy0 = max((int)(q.bby & 0xffff) - y_base, 0), qy1 = min((int)(q.bby >> 16) - y_base, kTileH - 1);
      const int qw = qx1 - qx0 + 1, npair = qw * ((qy1 - qy0 + 2) >> 1);
      const uint32_t inv = (uint32_t)ceilf(1048576.0f * __builtin_amdgcn_rcpf((float)qw));
      if (inside && qw == kTileW) {
        // the triangle covers the tile's full width on these rows (the interior of a wall): one column per
        // lane, the waves interleave the rows; per pixel one fma on top of the column's part of the plane
        // (the same two roundings as fragment()), the 24-bit conversion and the LDS atomic
        const int px = x_base + lane;
        const float zc = __fmaf_rn(q.dzdx, (float)px, q.a0);
        for (int ly = qy0 + (tid >> 6); ly <= qy1; ly += NT / 64) {
          if (MODE == 0) { RTUF_LANES(kLaneParkTrip, true); RTUF_LANES(kLaneParkFrag, true); }
          const float z = __fmaf_rn(q.dzdy, (float)(y_base + ly), zc);
          const unsigned long long key = ((unsigned long long)((RTUF_FAST_CLASS && !LOW && MODE == 0) ? z24_of_upper_half(z) : z24_of(z)) << 32) | (LOW ? (q.order | (__float_as_uint(z) & kf.lowmask)) : q.order);
          const int lidx = ly * kKeyStride + lane;
          if (MODE == 0) {
            RTUF_COUNT_TEST();
            atomicMin(&keys[lidx], key);
          } else {
            if (keys[lidx] == key) keys[lidx] = kResolvedBit | (unsigned long long)__float_as_uint(z);
          }
        }
      } else if (inside) {
        for (int idx = tid; idx < npair; idx += NT) {
          if (MODE == 0) RTUF_LANES(kLaneParkTrip, true);
          const int yy = (int)((uint32_t)__mul24(idx, (int)inv) >> 20);
          const int lx = qx0 + idx - __mul24(yy, qw), ly = qy0 + 2 * yy;
          const int lidx = ly * kKeyStride + lx;
          fragment<MODE, LOW>(keys, q, x_base + lx, y_base + ly, lidx, kf, kLaneParkFrag);
          if (ly < qy1) fragment<MODE, LOW>(keys, q, x_base + lx, y_base + ly + 1, lidx + kKeyStride, kf, kLaneParkFrag);
        }
      } else {
        for (int idx = tid; idx < npair; idx += NT) {
          if (MODE == 0) RTUF_LANES(kLaneParkTrip, true);
          const int yy = (int)((uint32_t)__mul24(idx, (int)inv) >> 20);
          raster_pair<MODE, LOW>(keys, q, x_base, y_base, qx0 + idx - __mul24(yy, qw), qy0 + 2 * yy, qy1, kf, kLaneParkFrag);
        }
      }
    }
    if (hb + 64 < nh) __syncthreads();               // (more than 64 parked: the next batch overwrites the LDS records)
  }
}

#ifndef RTUF_COVER_ONLY
#define RTUF_COVER_ONLY 1          // (A/B switch -- 0: tiles that hold nothing but their cover take the key tile like every other raster tile, as up to round 5)
#endif
#ifndef RTUF_FRAG_UNROLL
#define RTUF_FRAG_UNROLL 4
#endif
constexpr int kFragUnroll = RTUF_FRAG_UNROLL;
// Fragments of the small triangles: 8 bytes each, perfectly coalesced, one LDS atomic each.  They never need the exact-float-z
// pass (the set-up kernel keeps anything with window z near 0.5 or below as a record).  kFragUnroll fragments per lane and
// trip, ALL their loads issued before the first is used: the loop used to be load -> wait -> one LDS atomic, one exposed
// global-memory latency per 256 fragments of a tile (a tile of the headline workload holds ~1,000): tile kernel 297 -> 285 us.
// The first trip's loads are issued by the caller before the key tile is initialised (load_frags / apply_frags).
template <int NT>
__device__ __forceinline__ void load_frags(unsigned long long (&f)[kFragUnroll], const unsigned long long* frags, uint32_t nf, uint32_t base, int tid)
{
#pragma unroll
  for (int u = 0; u < kFragUnroll; u++) {
    const uint32_t i = base + (uint32_t)u * NT + (uint32_t)tid;
    f[u] = i < nf ? frags[i] : ~0ull;
  }
}
template <int NT>
__device__ __forceinline__ void apply_frags(unsigned long long* keys, const unsigned long long (&f)[kFragUnroll], uint32_t nf, uint32_t base, int tid, uint32_t zcover, int shift)
{
#pragma unroll
  for (int u = 0; u < kFragUnroll; u++) {
    const uint32_t i = base + (uint32_t)u * NT + (uint32_t)tid;
    const int fpos = (int)((uint32_t)f[u] & ((1u << kFragPosBits) - 1u));      // row * kTileW + column, as the set-up kernel wrote it
    const int lidx = RTUF_KEY_PAD ? fpos + (fpos / kTileW) * RTUF_KEY_PAD : fpos;
    const unsigned long long key = ((f[u] >> 40) << 32) | (((uint32_t)(f[u] >> kFragPosBits) & kMaxOrder) << shift);
    // (zcover == 0xffffffff: no cover, every fragment passes; behind the tile's cover: cannot win)
    RTUF_LANES(kLaneFragList, i < nf && (uint32_t)(f[u] >> 40) <= zcover);
    if (i < nf && (uint32_t)(f[u] >> 40) <= zcover) { RTUF_COUNT_TEST(); atomicMin(&keys[lidx], key); }
  }
}

// The output planes (and, in the compare kernel, the sensor plane) are touched exactly once: non-temporal
// accesses keep them from displacing the bins and the geometry in L2 / MALL (tile kernel -3 %, compare
// kernel -6 %).
__device__ __forceinline__ void store_stream4(float* dst, float a, float b, float c, float d)
{
  __builtin_nontemporal_store(a, dst); __builtin_nontemporal_store(b, dst + 1);
  __builtin_nontemporal_store(c, dst + 2); __builtin_nontemporal_store(d, dst + 3);
}
__device__ __forceinline__ void store_stream4(uint16_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  uint32_t* w = reinterpret_cast<uint32_t*>(dst);
  __builtin_nontemporal_store(a | (b << 16), w); __builtin_nontemporal_store(c | (d << 16), w + 1);
}
__device__ __forceinline__ float4 load_stream4(const float* src)
{
  return make_float4(__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 1),
                     __builtin_nontemporal_load(src + 2), __builtin_nontemporal_load(src + 3));
}

// 16UC1 <-> float exactly like the reference's cv::Mat::convertTo calls:
//   in : convertTo(CV_32F, 0.001)  -> float(u16) * 0.001f               (src/urdf_filter.cpp:287-288)
//   out: convertTo(CV_16U, 1000.0) -> saturate_cast<ushort>(cvRound(v * 1000.0f)): round half to even,
//        NaN / out-of-int-range -> "integer indefinite" -> 0, otherwise clamped to [0, 65535]   (:309-312)
__device__ __forceinline__ float u16_to_metres(uint32_t u) { return __fmul_rn((float)u, 0.001f); }
__device__ __forceinline__ uint32_t metres_to_u16(float m)
{
  const float v = __fmul_rn(m, 1000.0f);
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return 0u;
  const int i = __float2int_rn(v);
  return (uint32_t)min(max(i, 0), 65535);
}

// urdf_filter.frag:14-35.  num = z_near*z_far/(z_near-z_far) and off = z_far/(z_far-z_near)
// depend on uniforms only and are evaluated once per thread (same float operations).
// num = z_near*z_far/(z_near-z_far), off = z_far/(z_far-z_near): evaluated once per batch on the host, in float,
// exactly as the shader's to_linear_depth does (rtuf_api.cpp, enqueue_batch)
struct ShadeConsts { float num, off, max_diff, replace_value; bool core; };

// The IEEE division without the instructions that only matter for operands near the ends of the exponent range: the compiler
// expands a correctly rounded a / b into v_div_scale_f32 twice (operand pre-scaling), v_rcp_f32, one Newton step, the quotient
// with two residual corrections (the last as v_div_fmas_f32, which undoes the scaling) and v_div_fixup_f32 (zero / infinite /
// NaN / denormal operands).  With both operands and the quotient far inside the normal range the scalings are identities and
// the fix-up returns its input, and what is left is this: one v_rcp_f32 and seven 2-cycle instructions instead of eleven, four
// of them in the 4-cycle class.  The host admits it per batch (TileArgs::fast_div, rtuf_api.cpp) from the two constants;
// scripts/fdiv_check.hip compares it with __fdiv_rn for every float z in [-1, 1 + 2^-11] on the GPU: 0 of 5.75e10 quotients differ
// inside the admitted domain, and the two pairs outside it (z_far 10,000 x z_near) show what the rule is for -- there z - off
// passes through zero and the fix-up's infinity is not what the core returns (profiles/r06_experiment_fast_class_batches_2_3.txt).
#ifndef RTUF_FAST_DIV
#define RTUF_FAST_DIV RTUF_FAST_CLASS
#endif
__device__ __forceinline__ float div_core(float n, float d)
{
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __fmaf_rn(-d, r, 1.0f);
  r = __fmaf_rn(e, r, r);
  float q = __fmul_rn(n, r);
  float res = __fmaf_rn(-d, q, n);
  q = __fmaf_rn(res, r, q);
  res = __fmaf_rn(-d, q, n);
  return __fmaf_rn(res, r, q);
}

// sensor > shade_threshold(z)  <=>  should_filter of include/shaders/urdf_filter.frag:22-23
__device__ __forceinline__ float shade_threshold(float z, const ShadeConsts& k)
{
  const float d = __fsub_rn(z, k.off);
  const float virt = (RTUF_FAST_DIV && k.core) ? div_core(k.num, d) : __fdiv_rn(k.num, d);
  return __fsub_rn(virt, k.max_diff);
}

__device__ __forceinline__ float shade(float sensor, float z, const ShadeConsts& k, bool& filt)
{
  filt = sensor > shade_threshold(z, k);
  return filt ? k.replace_value : sensor;
}

// COVER: the batch ran the cover pass (bigrec_kernel<0>), so a bin's header may name a cover.  Without it (the host skips the
// pass while no scene has triangles that cover whole tiles) the cover code is compiled out: it costs the headline workload 3 %.
template <bool TWO_KERNEL, bool U16, bool BITS, bool COVER, int NT>
__device__ __forceinline__ void tile_body(const TileArgs& a)
{
  __shared__ unsigned long long keys[kKeyCount];
  __shared__ uint32_t s_huge[2 + kHugeMax];        // raster_bin's list of whole-tile triangles (+ count in front, area threshold behind)
  __shared__ uint32_t s_winners[kWinnerWords];     // exact-z pass: filter of the draw-order keys that won a pixel in need
  __shared__ TriRec s_prec[64];                    // ... a batch of them unpacked by the first wave for all four,
  __shared__ uint4 s_pmeta[64];                    //     with {class, smallest depth, largest depth if it covers the whole tile}

  const int tid = threadIdx.x;
  RTUF_LANES_INIT();
  const int tiles = a.tiles_x * a.tiles_y;
  // grid = (tiles_x, tiles_y, streams of the group): no integer divisions to find the tile
  const int txi = blockIdx.x, tyi = blockIdx.y, slot = blockIdx.z;
  const int bin = slot * tiles + tyi * a.tiles_x + txi;
  const int stream = a.group_base + slot;
  const int x_base = txi * kTileW, y_base = tyi * kTileH;

  // the record bin's two fill counters and the tile's cover (the nearest triangle that covers this whole tile, if any:
  // bound of every key's depth, and its plane) in ONE 16-byte scalar load, the fragment bin's counter in a second one
  uint32_t count_front, count_back;
  unsigned long long cover;
  {
    const uint4 h = *reinterpret_cast<const uint4*>(a.bin_hdr + bin);
    count_front = h.x; count_back = h.y;
    cover = COVER ? ((unsigned long long)h.w << 32) | h.z : kNoCover;
  }
  // fragment count, and in the top bit: the bin holds a near record (or its cover is one): the depth keys of this tile
  // carry the low bits of the float z (KeyFmt)
  const uint32_t fraw = a.fbin_count[bin];
  const uint32_t fcount = fraw & 0x7fffffffu;
  const bool near_tile = (fraw >> 31) != 0u;              // (wave-uniform: a scalar load of a workgroup-uniform address)
  KeyFmt kf;
  kf.shift = a.key_shift;
  kf.abl = a.flags;
  kf.lowmask = near_tile ? (1u << a.key_shift) - 1u : 0u;
  kf.zexact = near_tile ? 1u << (26 - a.key_shift) : 8388609u;
  const uint32_t count = count_front + count_back;
  // (the stream's background entry after the bin's header in program order: the compiler then issues the three scalar loads
  // together -- with the background first it waited for it before it even computed the header's address)
  const BgInfo bi = a.bg[stream];
  const bool analytic_bg = bi.mode != 0;
  const float bgz = bi.z, thr_bg = bi.thr;
  const unsigned long long bgkey = analytic_bg ? ((unsigned long long)bi.z24 << 32) : kNoFragment;
  ShadeConsts sc;
  sc.num = a.sc_num; sc.off = a.sc_off; sc.max_diff = a.max_diff; sc.replace_value = a.replace_value; sc.core = a.fast_div != 0;
  const uint32_t zcover = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cover >> 32));      // (<= 0xffffff, or all ones: none)
  const uint32_t cover_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cover);
  const bool has_cover = zcover != 0xffffffffu;
  // The sensor pixels this lane will resolve.  A tile without geometry requests them as soon as the bin header says so (that
  // latency is all there is to such a tile); a tile with geometry requests them after the rasterisation: eight registers
  // held across the walks do not fit the kernel's 80-register budget (they spill, and the spill waits for the load), and
  // requesting them early AND late costs more in traffic than the early request hides (both measured, profiles/README.md).
  constexpr int kLanesPerRow = kTileW / 4, kRowsPerPass = NT / kLanesPerRow;
  constexpr int kPasses = (kTileH + kRowsPerPass - 1) / kRowsPerPass;      // resolve passes per tile (2 for 64x32)
  const bool vec = (a.width & 3) == 0;
  const int r_ly0 = tid / kLanesPerRow, r_lx = (tid % kLanesPerRow) * 4;
  const int r_px = x_base + r_lx;
  float4 sens_p[kPasses];
  auto request_sensor = [&]() {
#pragma unroll
    for (int ps = 0; ps < kPasses; ps++) {
      const int r_ly = r_ly0 + ps * kRowsPerPass, r_py = y_base + r_ly;
      const bool r_valid = r_ly < kTileH && r_py < a.height && r_px < a.width;
      // 64-bit part uniform (scalar), per-lane part 24-bit (W, H <= 2048)
      const size_t gofs = (size_t)stream * ((size_t)a.height * a.width) + (uint32_t)(__mul24(r_py, a.width) + r_px);
      sens_p[ps] = make_float4(0, 0, 0, 0);
      if (!TWO_KERNEL && r_valid && vec) {
        if (U16) {
          const ushort4 q = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(a.depth) + gofs);
          sens_p[ps] = make_float4(u16_to_metres(q.x), u16_to_metres(q.y), u16_to_metres(q.z), u16_to_metres(q.w));
        } else {
          sens_p[ps] = *reinterpret_cast<const float4*>(a.depth + gofs);
        }
      }
    }
  };
  // (an over-full bin is detected from the counters and the batch run again; until then stay inside the array)
  const uint32_t n_front = min(count_front, a.capacity), n = n_front + min(count_back, a.capacity - n_front), nf = min(fcount, a.fcapacity);
  const PackedTri* recs = a.bins + (size_t)bin * a.capacity;
  const unsigned long long* frags = reinterpret_cast<const unsigned long long*>(a.fbins) + (size_t)bin * a.fcapacity;
  // (RTUF_ABLATE builds only: flags bits 8.. are timing experiments, e.g. 0x100 skip rasterisation, 0x200 skip pixel loops)
  const bool empty = (n == 0 && nf == 0 && !has_cover) || RTUF_ABL(a.flags, 0x100u);   // no geometry in this tile: pure streaming compare
  if (empty && RTUF_ABL(a.flags, 0x1000000u)) return;               // timing experiment: raster tiles only
  // A tile that a triangle covers entirely and that holds NOTHING else (the wall behind the robot, wherever the robot is not:
  // most tiles of BASELINE config 4) needs no key tile either: every pixel is the cover's fragment or the background's,
  // decided and shaded from registers -- the same two fused multiply-adds, 24-bit conversion and key comparison the key tile
  // would see, and the fragment's float z at hand instead of rebuilt from the key.  One barrier (before the bin header is reset
  // for the next batch: every wave must have read it) instead of three, no LDS traffic.  Only in the kernels that run with the
  // cover pass: the others' code is what it was (this kernel sits on its register budget: a variable more in scope of the
  // common path cost the headline workload 20 % when this was first written for all variants).
  bool cover_only = false;
  float cov_a0 = 0.0f, cov_dzdx = 0.0f, cov_dzdy = 0.0f;
  uint32_t cov_order = 0u;
  if constexpr (COVER && RTUF_COVER_ONLY != 0) {
    cover_only = has_cover && n == 0 && nf == 0 && !empty;
    if (cover_only) {
      request_sensor();
      const uint4 pl = reinterpret_cast<const uint4*>(a.big_list + cover_idx)[1];       // {a0, dzdx, dzdy, order}: same address in every lane
      cov_a0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.x));
      cov_dzdx = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.y));
      cov_dzdy = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.z));
      cov_order = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl.w) & kOrderMask;
#ifdef RTUF_COUNT
      if (tid < 2) count_words()[tid] = 0u;
#endif
      __syncthreads();
      if (tid == 0) {
        *reinterpret_cast<uint4*>(a.bin_hdr + bin) = make_uint4(0u, 0u, 0xffffffffu, 0xffffffffu);      // ready for the next batch: nothing binned, no cover
        a.fbin_count[bin] = 0;                                                                          // (a near cover had set the near flag)
        atomicAdd(&a.counters->shard[bin % kCounterShards].cover_tiles, 1u);
      }
    }
  }
  if (empty) request_sensor();
  if (!empty && !cover_only) {
    // Initial depth keys: the background plane and, where a triangle covers the whole tile, that triangle's fragments --
    // evaluated per pixel exactly as fragment() would (same two fused multiply-adds, same 24-bit conversion, its draw
    // order), but written instead of min'ed: it is the first thing the tile sees, and bigrec_kernel<1> left it out of the bin.
    // (the plane is fetched again by the rare exact-z pass below rather than kept in registers across the rasterisation)
    struct CoverPlane { float a0, dzdx, dzdy; uint32_t order; };
    auto cover_plane = [&]() {
      const uint4 pl = reinterpret_cast<const uint4*>(a.big_list + cover_idx)[1];       // {a0, dzdx, dzdy, order}: same address in every lane
      CoverPlane c;
      c.a0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.x));
      c.dzdx = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.y));
      c.dzdy = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)pl.z));
      c.order = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl.w) & kOrderMask;
      return c;
    };
    // Ask for the first things this lane will need from the bins BEFORE the key tile is initialised: its first record and
    // its first fragments (their latency then passes under the initialisation and the barrier instead of after it).
#ifndef RTUF_PRELOAD
#define RTUF_PRELOAD 1           // (0: A/B switch -- first record and first fragments are requested after the barrier, as before)
#endif
    uint4 first_rec0 = make_uint4(0u, 0u, 0u, 0u), first_rec1 = make_uint4(0u, 0u, 0u, 0u);
    if (RTUF_PRELOAD && (uint32_t)tid < n) {
      const uint4* src = reinterpret_cast<const uint4*>(recs + ((uint32_t)tid < n_front ? (uint32_t)tid : a.capacity - 1u - ((uint32_t)tid - n_front)));
      first_rec0 = src[0]; first_rec1 = src[1];
    }
    unsigned long long first_frags[kFragUnroll];
    if (RTUF_PRELOAD) load_frags<NT>(first_frags, frags, nf, 0u, tid);
    if (has_cover) {
      const CoverPlane c = cover_plane();
      for (int i = tid; i < kKeyCount; i += NT) {        // (padding columns, if any, hold the background key: the scans below skip them like any pixel nothing was drawn to)
        const float z = __fmaf_rn(c.dzdy, (float)(y_base + i / kKeyStride), __fmaf_rn(c.dzdx, (float)(x_base + i % kKeyStride), c.a0));
        const unsigned long long key = ((unsigned long long)z24_of(z) << 32) | (c.order << kf.shift) | (__float_as_uint(z) & kf.lowmask);
        keys[i] = (RTUF_KEY_PAD == 0 || i % kKeyStride < kTileW) ? min(key, bgkey) : bgkey;
      }
    } else {
      for (int i = tid; i < kKeyCount; i += NT) keys[i] = bgkey;
    }
    if (tid == 0) { s_huge[0] = 0; s_huge[1 + kHugeMax] = n <= (uint32_t)kParkBelow ? (uint32_t)kQuarterArea : (uint32_t)kWallArea; }
#ifdef RTUF_COUNT
    if (tid < 2) count_words()[tid] = 0u;
#endif
    __syncthreads();
    if (tid == 0) {
      *reinterpret_cast<uint4*>(a.bin_hdr + bin) = make_uint4(0u, 0u, 0xffffffffu, 0xffffffffu);      // ready for the next batch: nothing binned, no cover
      a.fbin_count[bin] = 0;
      CounterShard& sh = a.counters->shard[bin % kCounterShards];
      if (count) atomicMax(&sh.max_bin_fill, count);
      if (fcount) atomicMax(&sh.max_fbin_fill, fcount);
      if (has_cover) atomicAdd(&sh.cover_tiles, 1u);
#ifdef RTUF_HIST            // (scripts/bin_hist.sh, never the product: how much of the bins lies beyond a direct capacity of RTUF_HIST)
      {
        const uint32_t cls_count = RTUF_HIST_BACK ? count_back : count_front;
        if (cls_count > (uint32_t)RTUF_HIST) atomicAdd(&sh.raster_atomics, (1ull << 40) + (cls_count - (uint32_t)RTUF_HIST));
        if (fcount > 4u * RTUF_HIST) atomicAdd(&sh.drawn_pixels, (1ull << 40) + (fcount - 4u * RTUF_HIST));
      }
#endif
    }
    // fragments first (one LDS atomic each; the first trip's are in registers already), then the records
    // (two instances of the rasterisation: tiles without near geometry -- every tile of a robot at arm's length -- do not
    // pay the instruction that puts the float's low bits into the key)
    {
      bool do_frags = true, do_records = true, load_only = false;
      int skip = 0;
#ifdef RTUF_ABLATE
      do_frags = !(a.flags & 0x400u); do_records = !(a.flags & 0x800u); load_only = (a.flags & 0x200u) != 0; skip = (int)((a.flags >> 12) & 3u);
#endif
      if (do_frags) {
        if (!RTUF_PRELOAD) load_frags<NT>(first_frags, frags, nf, 0u, tid);
        apply_frags<NT>(keys, first_frags, nf, 0u, tid, zcover, kf.shift);
        for (uint32_t base = kFragUnroll * NT; base < nf; base += kFragUnroll * NT) {
          unsigned long long f[kFragUnroll];
          load_frags<NT>(f, frags, nf, base, tid);
          apply_frags<NT>(keys, f, nf, base, tid, zcover, kf.shift);
        }
      }
      if (do_records) {
        if (near_tile) raster_bin<0, true, NT>(keys, recs, n, x_base, y_base, tid, load_only, a.width, a.height, n_front, a.capacity, s_huge, s_prec, s_pmeta, zcover, s_winners, kf, first_rec0, first_rec1, RTUF_PRELOAD != 0, skip);
        else raster_bin<0, false, NT>(keys, recs, n, x_base, y_base, tid, load_only, a.width, a.height, n_front, a.capacity, s_huge, s_prec, s_pmeta, zcover, s_winners, kf, first_rec0, first_rec1, RTUF_PRELOAD != 0, skip);
      }
    }
    __syncthreads();
    if (tid == 0) s_huge[0] = 0;             // the exact-z pass below builds its list again

    // Does any pixel need a second look for the exact float z of its winner?  In the upper half of the depth range
    // (z24 > 2^23) float z == (z24 + 1) * 2^-24 exactly; below, a tile with near geometry has the float's low bits in its
    // keys, which settle everything but the last micrometres in front of the near plane (z24 < zexact); a tile without
    // them has no record that could get there at all (the set-up marks those near) -- the test stays, it costs nothing.
    // (round 6: a tile without near geometry does not look -- no record, fragment or cover of it can produce z < 0.51, which is
    // what its depth tests already rely on (z24_of_upper_half) -- and is spared eight key reads per lane and a barrier)
    bool need = false;
    if (near_tile || !RTUF_FAST_RESOLVE) {
      for (int i = tid; i < kKeyCount; i += NT) {
        const unsigned long long k = keys[i];
        if (k != bgkey && (uint32_t)(k >> 32) < kf.zexact) need = true;
      }
      need = __syncthreads_or(need) != 0;
    }
    if (need && !RTUF_ABL(a.flags, 0x4000000u)) {      // (0x4000000: timing experiment, no exact-z pass)
      // which draw-order keys won such a pixel: only their records are walked again
      if (tid < kWinnerWords) s_winners[tid] = 0u;
      if (tid == 0) atomicAdd(&a.counters->shard[bin % kCounterShards].exact_tiles, 1u);
      __syncthreads();
      for (int i = tid; i < kKeyCount; i += NT) {
        const unsigned long long k = keys[i];
        if (k != bgkey && (uint32_t)(k >> 32) < kf.zexact) {
          const uint32_t h = winner_slot((uint32_t)k >> kf.shift);
          atomicOr(&s_winners[h >> 5], 1u << (h & 31u));
        }
      }
      __syncthreads();
      raster_bin<1, true, NT>(keys, recs, n, x_base, y_base, tid, false, a.width, a.height, n_front, a.capacity, s_huge, s_prec, s_pmeta, zcover, s_winners, kf, first_rec0, first_rec1, false);
      if (has_cover) {                       // ... and the cover triangle, which is in no bin
        const CoverPlane c = cover_plane();
        for (int i = tid; i < kKeyCount; i += NT) {
          const float z = __fmaf_rn(c.dzdy, (float)(y_base + i / kKeyStride), __fmaf_rn(c.dzdx, (float)(x_base + i % kKeyStride), c.a0));
          const unsigned long long key = ((unsigned long long)z24_of(z) << 32) | (c.order << kf.shift) | (__float_as_uint(z) & kf.lowmask);
          if (keys[i] == key) keys[i] = kResolvedBit | (unsigned long long)__float_as_uint(z);
        }
      }
      __syncthreads();
    }
    // (requested here, not before the scan above: with the eight registers live from there the kernel sits at its 80-register
    // limit and the allocation of everything before it suffers -- measured once more in round 4: tile kernel 296 instead of 280 us)
    request_sensor();
  }

  // resolve: kLanesPerRow lanes x 4 pixels per tile row, kRowsPerPass rows per pass.  `finish` turns four
  // window depths (or "nothing drawn") and their compare thresholds into the outputs.
  // With BITS the only output is one mask bit per pixel: `finish` returns the lane's four mask flags (bit j = pixel j)
  // and stores nothing; the caller packs the flags of 8 neighbouring lanes into one 32-bit word.
  bool uncovered = false;                              // BITS: a pixel no fragment reached (its masked depth would be the clear colour, not the sensor value)
  auto finish = [&](int ps, const float (&z)[4], const float (&thr)[4], const bool (&frag)[4]) -> uint32_t {
    const int r_ly = r_ly0 + ps * kRowsPerPass, py = y_base + r_ly, px = r_px;
    const size_t gofs = (size_t)stream * ((size_t)a.height * a.width) + (uint32_t)(__mul24(py, a.width) + px);
    const int nvalid = min(4, a.width - px);
    if (TWO_KERNEL) {
      const size_t zofs = (size_t)slot * ((size_t)a.height * a.width) + (uint32_t)(__mul24(py, a.width) + px);
      // "no fragment" (only without background quad) is encoded as NaN
      float zz[4];
#pragma unroll
      for (int j = 0; j < 4; j++) zz[j] = frag[j] ? z[j] : __uint_as_float(0x7fc00000u);
      if (vec) *reinterpret_cast<float4*>(a.zsurface + zofs) = make_float4(zz[0], zz[1], zz[2], zz[3]);
      else for (int j = 0; j < nvalid; j++) a.zsurface[zofs + j] = zz[j];
      return 0u;
    }
    float s[4];
    if (vec) {
      s[0] = sens_p[ps].x; s[1] = sens_p[ps].y; s[2] = sens_p[ps].z; s[3] = sens_p[ps].w;
    } else {
      for (int j = 0; j < 4; j++)
        s[j] = j < nvalid ? (U16 ? u16_to_metres(reinterpret_cast<const uint16_t*>(a.depth)[gofs + j]) : a.depth[gofs + j]) : 0.0f;
    }
    float o[4];
    uint32_t mbits = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      bool f = s[j] > thr[j];
      o[j] = f ? sc.replace_value : s[j];
      if (!frag[j]) { o[j] = 0.0f; f = false; if (BITS) uncovered = true; }     // GL clear colour
      if (f) mbits |= (BITS ? 1u : 0xffu) << ((BITS ? 1 : 8) * j);
    }
    if (BITS) return mbits;
    if (vec) {
      if (U16) {
        store_stream4(reinterpret_cast<uint16_t*>(a.masked) + gofs, metres_to_u16(o[0]), metres_to_u16(o[1]), metres_to_u16(o[2]), metres_to_u16(o[3]));
      } else {
        store_stream4(a.masked + gofs, o[0], o[1], o[2], o[3]);
      }
      if (a.mask) __builtin_nontemporal_store(mbits, reinterpret_cast<uint32_t*>(a.mask + gofs));
    } else {
      for (int j = 0; j < nvalid; j++) {
        if (U16) reinterpret_cast<uint16_t*>(a.masked)[gofs + j] = (uint16_t)metres_to_u16(o[j]);
        else a.masked[gofs + j] = o[j];
        if (a.mask) a.mask[gofs + j] = (uint8_t)(mbits >> (8 * j));
      }
    }
    return mbits;
  };
  // Most pixels of a frame see only the background plane, whose depth is one value per stream: its compare
  // threshold (one IEEE division) is computed once per lane.  Same operations on the same values as per pixel.
  const float bg_z4[4] = {bgz, bgz, bgz, bgz}, bg_thr4[4] = {thr_bg, thr_bg, thr_bg, thr_bg};
  const bool bg_frag4[4] = {analytic_bg, analytic_bg, analytic_bg, analytic_bg};
#pragma unroll
  for (int ps = 0; ps < kPasses; ps++) {
    const int r_ly = r_ly0 + ps * kRowsPerPass;
    const bool valid = r_ly < kTileH && y_base + r_ly < a.height && r_px < a.width;
    uint32_t flags4 = 0;
    RTUF_LANES(kLaneResolve, valid);
    if (valid) {
      if (empty) {                                   // tile without geometry: a streaming compare against the plane
        if (RTUF_FAST_RESOLVE && analytic_bg) {          // (uniform: with the flags known to be set the four selects on them fall away)
          const bool all4[4] = {true, true, true, true};
          flags4 = finish(ps, bg_z4, bg_thr4, all4);
        } else {
          flags4 = finish(ps, bg_z4, bg_thr4, bg_frag4);
        }
      } else if (COVER && RTUF_COVER_ONLY != 0 && cover_only) {      // nothing but a triangle over the whole tile: its fragment or the background's, from registers
        float z[4], thr[4];
        bool frag[4];
        const float zrow = (float)(y_base + r_ly);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float zf = __fmaf_rn(cov_dzdy, zrow, __fmaf_rn(cov_dzdx, (float)(r_px + j), cov_a0));
#if RTUF_FAST_COVER
          // what atomicMin on a key tile initialised with the background would keep: the cover's key {z24, order << shift | low
          // bits} against the background's (or "no fragment"'s) {z24, 0} -- its draw order is at least 1, so it is below exactly
          // when its 24-bit depth is: one 32-bit compare, no key to put together; without near geometry (a near cover marks its
          // tile) the depth comes from the product's bit pattern like everywhere else
          const uint32_t cz24 = near_tile ? z24_of(zf) : z24_of_upper_half(zf);
          const bool drawn = cz24 < (uint32_t)(bgkey >> 32);
#else
          const unsigned long long key = ((unsigned long long)z24_of(zf) << 32) | (cov_order << kf.shift) | (__float_as_uint(zf) & kf.lowmask);
          const bool drawn = key < bgkey;            // (what atomicMin on a key tile initialised with the background would keep)
#endif
          z[j] = drawn ? zf : bgz;
          frag[j] = drawn ? true : analytic_bg;
          thr[j] = thr_bg;
          if (drawn && !TWO_KERNEL) thr[j] = shade_threshold(zf, sc);
#ifdef RTUF_COUNT
          if (drawn && r_px + j < a.width) atomicAdd(&count_words()[1], 1u);
#endif
        }
        flags4 = finish(ps, z, thr, frag);
      } else {
        float z[4], thr[4];
        bool frag[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned long long k = keys[r_ly * kKeyStride + r_lx + j];
          frag[j] = true;
          thr[j] = thr_bg;
          if (k == bgkey) { z[j] = bgz; frag[j] = analytic_bg; }
          else {                                        // the per-pixel division only runs where something was drawn
#ifdef RTUF_COUNT
            if (r_px + j < a.width) atomicAdd(&count_words()[1], 1u);
#endif
            // float z of the winner: written by the exact-z pass (rare), or (z24 + 1) * 2^-24 in the upper half of the depth
            // range, or -- only in tiles with near geometry (uniform test: the headline workload never gets there) -- from
            // z24 and the low bits of the float the key carries
            const uint32_t khi = (uint32_t)(k >> 32);
            if (RTUF_FAST_RESOLVE && !near_tile) {          // (uniform) no exact-z pass ran here, no depth below 0.51 exists
              z[j] = __uint_as_float(khi + 0x3E800001u);
            } else if (k & kResolvedBit) {
              z[j] = __uint_as_float((uint32_t)k);
            } else {
              // (z24 + 1) * 2^-24.  For z24 >= 2^23 - 1 that float's bit pattern is z24 + 0x3E800001 (the integer is its own
              // mantissa, the power of two an exponent offset, and 2^24 carries into the exponent: checked for all 2^23 values
              // on the CPU) -- one 2-cycle add instead of a conversion and a multiply; smaller depths exist only in tiles
              // with near geometry and take near_z_from_key below.
              z[j] = RTUF_FAST_RESOLVE ? __uint_as_float(khi + 0x3E800001u) : __fmul_rn((float)(khi + 1u), 5.9604644775390625e-08f);
              if (near_tile) {
                if (khi <= 8388608u) z[j] = near_z_from_key(khi, (uint32_t)k & kf.lowmask, kf.shift);
              }
            }
            if (!TWO_KERNEL) thr[j] = shade_threshold(z[j], sc);
          }
        }
        flags4 = finish(ps, z, thr, frag);
      }
    }
    if (BITS) {
      // 8 neighbouring lanes hold the 32 pixels of one output word (pixel x -> bit x % 32 of word x / 32 of its row):
      // OR their nibbles together (all lanes take part, invalid ones with 0) and let the group's first lane store
      uint32_t word = flags4 << (4 * (tid & 7));
      word |= (uint32_t)__shfl_xor((int)word, 1);
      word |= (uint32_t)__shfl_xor((int)word, 2);
      word |= (uint32_t)__shfl_xor((int)word, 4);
      if (valid && (tid & 7) == 0) {
        const int row_words = (a.width + 31) >> 5;
        const size_t wofs = ((size_t)stream * a.height + (size_t)(y_base + r_ly)) * (size_t)row_words + (size_t)(r_px >> 5);
        __builtin_nontemporal_store(word, a.bits + wofs);
      }
    }
  }
  if (BITS && __syncthreads_or(uncovered) && tid == 0) a.counters->shard[bin % kCounterShards].uncovered = 1u;
  RTUF_LANES_FLUSH(a.counters->shard[bin % kCounterShards]);
#ifdef RTUF_COUNT
  __syncthreads();
  if (!empty && tid == 0) {
    atomicAdd(&a.counters->shard[bin % kCounterShards].raster_atomics, (unsigned long long)count_words()[0]);
    atomicAdd(&a.counters->shard[bin % kCounterShards].drawn_pixels, (unsigned long long)count_words()[1]);
  }
#endif
}

// Registers: 7 waves/SIMD (72 VGPRs; seven workgroups' key tiles are 159.6 of the CU's 160 KB of LDS) for the variants
// without the cover pass -- the compiler spills two or three registers there and the kernel is still 5 % faster than at
// 6 waves/SIMD (80 VGPRs, where it needs 74: tile kernel 275 -> 261 us on the 256-stream VGA workload), the seventh
// workgroup per CU hides what the lanes wait for; the cover variants (walls: C4) were left at 6 up to round 5, where they
// measured the same alone and 2 % better beside the other lane's set-up kernel -- with the cover-only tiles resolved from
// registers (round 6) 7 is ahead there as well.  Left alone the compiler takes 84 registers = 5 waves/SIMD.
#ifndef RTUF_TILE_WAVES
#define RTUF_TILE_WAVES 7
#endif
#ifndef RTUF_TILE_WAVES_COVER
#define RTUF_TILE_WAVES_COVER 7      // (round 6, with the cover-only tiles off the key tile: 7 is 2.7 % ahead of 6 on C4's tile kernel, +0.7 % frames/s; up to round 5: 6)
#endif
// NT = threads of the workgroup: 256 (four waves per tile) for launches that fill the GPU; 1,024 (sixteen) for the small launches of
// one or a few camera streams, where the kernel's time is the serial window loop of its fullest tiles -- 150 workgroups for one
// VGA stream, each walking its bin 256 records at a time -- and more waves per tile shorten exactly that (one stream: raster
// stage 42 -> 23 us, 14.2 k -> 17.6 k frames/s; eight streams 98 k -> 109 k; profiles/r05_experiment_tile_threads.txt).
constexpr int kTileThreadsSmall = 1024;
template <bool TWO_KERNEL, bool U16, bool COVER, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(COVER ? RTUF_TILE_WAVES_COVER : RTUF_TILE_WAVES))) void tile_kernel(TileArgs a) { tile_body<TWO_KERNEL, U16, false, COVER, NT>(a); }
// mask-only output, one bit per pixel (rtuf_filter_batch_bits*): 4 (2) B/pixel in, 1/8 B/pixel out
template <bool U16, bool COVER, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(COVER ? RTUF_TILE_WAVES_COVER : RTUF_TILE_WAVES))) void tile_bits_kernel(TileArgs a) { tile_body<false, U16, true, COVER, NT>(a); }

// ---------------------------------------------------------------------------------------
// compare_kernel (two-kernel mode): streaming, 13 B/pixel (4 sensor + 4 z + 4 masked + 1 mask)
// ---------------------------------------------------------------------------------------

#ifndef RTUF_CMP_UNROLL
#define RTUF_CMP_UNROLL 1
#endif
constexpr int kCmpUnroll = RTUF_CMP_UNROLL;
template <bool U16>
__global__ __launch_bounds__(kBlock) void compare_kernel(CompareArgs a)
{
  ShadeConsts sc;
  sc.num = a.sc_num; sc.off = a.sc_off; sc.max_diff = a.max_diff; sc.replace_value = a.replace_value; sc.core = a.fast_div != 0;
  const size_t n4 = a.n_pixels >> 2;
  const uint16_t* in16 = reinterpret_cast<const uint16_t*>(a.depth);
  uint16_t* out16 = reinterpret_cast<uint16_t*>(a.masked);
  // No grid-stride loop: every workgroup takes kCmpUnroll * kBlock consecutive quads of pixels, so the workgroups in
  // flight form ONE compact front through the four arrays (a strided loop over 32 k workgroups had every lane jump
  // 134 MB per trip at 1,024 streams: 829 us instead of 640 for the same 4.09 GB).  All loads of a lane are issued
  // before the first is used.
  {
    const size_t i0 = (size_t)blockIdx.x * blockDim.x * kCmpUnroll + threadIdx.x;
    const size_t stride = blockDim.x;
    float sv[kCmpUnroll][4], zv[kCmpUnroll][4];
#pragma unroll
    for (int u = 0; u < kCmpUnroll; u++) {
      const size_t i = i0 + (size_t)u * stride;
      if (i >= n4) break;
      if (U16) {
        const ushort4 q = reinterpret_cast<const ushort4*>(in16)[i];
        sv[u][0] = u16_to_metres(q.x); sv[u][1] = u16_to_metres(q.y); sv[u][2] = u16_to_metres(q.z); sv[u][3] = u16_to_metres(q.w);
      } else {
        const float4 s = load_stream4(a.depth + 4 * i);
        sv[u][0] = s.x; sv[u][1] = s.y; sv[u][2] = s.z; sv[u][3] = s.w;
      }
      const float4 z = load_stream4(a.zsurface + 4 * i);      // (the tile kernel's z stores stay temporal: this read may still find them in MALL)
      zv[u][0] = z.x; zv[u][1] = z.y; zv[u][2] = z.z; zv[u][3] = z.w;
    }
#pragma unroll
    for (int u = 0; u < kCmpUnroll; u++) {
      const size_t i = i0 + (size_t)u * stride;
      if (i >= n4) break;
      float o[4];
      uint32_t mbits = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        bool f;
        o[j] = shade(sv[u][j], zv[u][j], sc, f);
        if (zv[u][j] != zv[u][j]) { o[j] = 0.0f; f = false; }
        if (f) mbits |= 0xffu << (8 * j);
      }
      if (U16) {
        store_stream4(out16 + 4 * i, metres_to_u16(o[0]), metres_to_u16(o[1]), metres_to_u16(o[2]), metres_to_u16(o[3]));
      } else {
        store_stream4(a.masked + 4 * i, o[0], o[1], o[2], o[3]);
      }
      if (a.mask) __builtin_nontemporal_store(mbits, reinterpret_cast<uint32_t*>(a.mask) + i);
    }
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (a.n_pixels & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    bool f;
    const float sen = U16 ? u16_to_metres(in16[i]) : a.depth[i];
    float o = shade(sen, a.zsurface[i], sc, f);
    if (a.zsurface[i] != a.zsurface[i]) { o = 0.0f; f = false; }
    if (U16) out16[i] = (uint16_t)metres_to_u16(o);
    else a.masked[i] = o;
    if (a.mask) a.mask[i] = f ? 255 : 0;
  }
}

// A batch's counters (one block per launch group) go to pinned host memory with plain stores from a tiny kernel, one
// workgroup per block: a hipMemcpyAsync of 8 KB costs a blit launch plus ~10 us of copy-engine set-up on the stream.
// Every raster lane publishes the blocks of its own groups (first, first + stride, ...) at the end of its part of the batch.
// It is the last kernel a lane runs for a launch group, and it sees every counter the host will judge the batch by when it
// retires it: it makes the same judgement on the device and leaves it in the batch slot's status word, for consumers that read
// the output planes on a stream of their own (rtuf_order_stream_after_batches) before the host has retired the batch.
__global__ void publish_counters_kernel(const Counters* __restrict__ src, Counters* __restrict__ host_dst, int first, int stride,
                                        uint32_t* status, PublishLimits lim)
{
  const int b = first + (int)blockIdx.x * stride;
  const uint4* s = reinterpret_cast<const uint4*>(src + b);
  uint4* d = reinterpret_cast<uint4*>(host_dst + b);
  for (int i = threadIdx.x; i < (int)(sizeof(Counters) / 16); i += blockDim.x) d[i] = s[i];
  if (threadIdx.x < 64) {            // the first wave: a shard per lane (kCounterShards <= 64)
    uint32_t f = 0;
    for (int i = threadIdx.x; i < kCounterShards; i += 64) {
      const CounterShard& sh = src[b].shard[i];
      if (sh.max_bin_fill > lim.capacity || sh.max_fbin_fill > lim.fcapacity) f |= kStatusBinOverflow;
      if (sh.clip_overflow) f |= kStatusClipOverflow;
      if (sh.max_big_fill > lim.big_capacity) f |= kStatusBigOverflow;
      if (sh.uncovered) f |= kStatusUncovered;
    }
    if (threadIdx.x == 0 && src[b].work.n_items > src[b].work.grid) f |= kStatusGridShort;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) f |= (uint32_t)__shfl_xor((int)f, o);
    if (threadIdx.x == 0) {
      if (f) atomicOr(status, f);
      __threadfence();               // the flags are visible before this group counts as finished
      atomicSub(status, 1u);
    }
  }
}

// bin headers of a fresh (or regrown) working set: nothing binned, no cover
__global__ void init_headers_kernel(BinHeader* hdr, size_t n_bins)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_bins) *reinterpret_cast<uint4*>(hdr + i) = make_uint4(0u, 0u, 0xffffffffu, 0xffffffffu);
}

// One wave that does nothing for `ticks` of the constant-rate clock: rtuf_create uses two of them to find out whether two HIP
// streams can run side by side (the runtime multiplexes streams onto a few hardware queues, and two streams that share a
// queue run their kernels one after the other).
__global__ void spin_kernel(unsigned long long ticks)
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
void launch_spin(unsigned long long ticks, hipStream_t st) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ticks); }

// host-callable launchers ---------------------------------------------------------------
void launch_publish_counters(const Counters* src, Counters* host_dst, int first, int stride, int count, uint32_t* status, PublishLimits lim, hipStream_t st)
{
  if (count > 0) hipLaunchKernelGGL(publish_counters_kernel, dim3((unsigned)count), dim3(256), 0, st, src, host_dst, first, stride, status, lim);
}
void launch_init_headers(BinHeader* hdr, size_t n_bins, hipStream_t st)
{
  hipLaunchKernelGGL(init_headers_kernel, dim3((unsigned)((n_bins + 255) / 256)), dim3(256), 0, st, hdr, n_bins);
}
void launch_fk(const FkArgs& a, hipStream_t st)
{
  if (a.n_frames <= kFkMaxFrames) {
    const int threads = a.n_frames <= 64 ? 64 : (a.n_frames <= 128 ? 128 : kFkMaxFrames);
    hipLaunchKernelGGL(fk_tree_kernel, dim3(a.n_streams), dim3(threads), 0, st, a);
    return;
  }
  const int total = a.n_streams * (a.n_links_model + 1);
  hipLaunchKernelGGL(fk_kernel, dim3((total + 63) / 64), dim3(64), 0, st, a);
}
void launch_pose(const PoseArgs& a, hipStream_t st)
{
  const int total = a.n_streams * (a.n_draws + 1);
  hipLaunchKernelGGL(pose_kernel, dim3((total + 127) / 128), dim3(128), 0, st, a);
}
void launch_cull(const SetupArgs& a, hipStream_t st)
{
  hipLaunchKernelGGL(cull_kernel, dim3(a.n_chunks), dim3(kBlock), 0, st, a);
}
uint32_t launch_setup(const SetupArgs& a, uint32_t items_hint, bool sweep, hipStream_t st)
{
  // The work-list length is only known on the device.  The grid is sized from the previous batch's
  // length (+25 %; robot and camera move little between frames).  Without a hint: the worst case (every
  // chunk visible in every stream).  Neighbouring workgroups take neighbouring items, which share a
  // chunk's geometry in L2.  If the list may be longer than the grid, either a small strided launch of
  // the same code sweeps the remainder (sweep = true), or the caller compares the returned grid size
  // with the list length it reads back and runs the batch again (a wrong guess costs time, never pixels).
  const long long worst = (long long)a.n_chunks * max_items_per_chunk(a.group_size);
  long long grid = worst;
  if (items_hint) grid = std::min<long long>(worst, (long long)items_hint + items_hint / 4 + 64);
  hipLaunchKernelGGL(setup_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, st, a, 0u);
  if (grid < worst && sweep) hipLaunchKernelGGL(setup_kernel<true>, dim3(256), dim3(kBlock), 0, st, a, (uint32_t)grid);
  return (uint32_t)grid;
}
// workgroups per counter shard: enough to take the shard's whole list in one pass, at most kClipGridWgs (a context of one
// camera stream launches 8 per shard and holds an 8-MiB spill area instead of 64 MiB)
static int clip_wgs_per_shard(uint32_t clip_capacity) { return (int)std::min<uint32_t>((clip_capacity + kClipBlock - 1) / kClipBlock, (uint32_t)kClipGridWgs); }
void launch_clip(const SetupArgs& a, hipStream_t st)
{
  // the item count lives on the device: fixed grid, grid-stride loop
  hipLaunchKernelGGL(clip_kernel, dim3(kCounterShards * clip_wgs_per_shard(a.clip_capacity)), dim3(kClipBlock), 0, st, a);
}
size_t clip_spill_bytes(uint32_t clip_capacity) { return (size_t)(kClipMaxV - kClipLdsV) * kCounterShards * clip_wgs_per_shard(clip_capacity) * kClipBlock * sizeof(float4); }
void launch_bigrec(const SetupArgs& a, bool cover_pass, hipStream_t st)
{
  // the list lengths live on the device: fixed grid, kBigWavesPerShard waves per shard, each strides over its shard's list
  if (cover_pass) hipLaunchKernelGGL(bigrec_kernel<0>, dim3(kCounterShards * kBigWavesPerShard / 4), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bigrec_kernel<1>, dim3(kCounterShards * kBigWavesPerShard / 4), dim3(256), 0, st, a);
}
#ifndef RTUF_SMALL_LAUNCH
#define RTUF_SMALL_LAUNCH 2048     // launches of fewer tile workgroups than this take the 1,024-thread kernels (0: never)
#endif
// (RTUF_SMALL_LAUNCH in the environment overrides the built-in threshold: the tests run their small scenes through both kernels)
static long long small_launch_threshold()
{
  static const long long t = [] { const char* e = getenv("RTUF_SMALL_LAUNCH"); return e && *e ? atoll(e) : (long long)RTUF_SMALL_LAUNCH; }();
  return t;
}
template <bool COVER, int NT>
static void launch_tile_variant(const TileArgs& a, bool two_kernel, hipStream_t st)
{
  const dim3 grid(a.tiles_x, a.tiles_y, a.group_size);
  if (a.bits) {
    if (a.io_u16) hipLaunchKernelGGL((tile_bits_kernel<true, COVER, NT>), grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((tile_bits_kernel<false, COVER, NT>), grid, dim3(NT), 0, st, a);
  } else if (two_kernel) hipLaunchKernelGGL((tile_kernel<true, false, COVER, NT>), grid, dim3(NT), 0, st, a);
  else if (a.io_u16) hipLaunchKernelGGL((tile_kernel<false, true, COVER, NT>), grid, dim3(NT), 0, st, a);
  else hipLaunchKernelGGL((tile_kernel<false, false, COVER, NT>), grid, dim3(NT), 0, st, a);
}
void launch_tile(const TileArgs& a, bool two_kernel, bool cover_pass, hipStream_t st)
{
  const bool small = (long long)a.tiles_x * a.tiles_y * a.group_size < small_launch_threshold();
  if (small) {
    if (cover_pass) launch_tile_variant<true, kTileThreadsSmall>(a, two_kernel, st);
    else launch_tile_variant<false, kTileThreadsSmall>(a, two_kernel, st);
  } else {
    if (cover_pass) launch_tile_variant<true, kTileThreads>(a, two_kernel, st);
    else launch_tile_variant<false, kTileThreads>(a, two_kernel, st);
  }
}
void launch_compare(const CompareArgs& a, hipStream_t st)
{
  size_t n4 = a.n_pixels >> 2;
  size_t blocks = (n4 + (size_t)kBlock * kCmpUnroll - 1) / ((size_t)kBlock * kCmpUnroll);
  if (blocks == 0) blocks = 1;
  if (a.io_u16) hipLaunchKernelGGL(compare_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL(compare_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

}  // namespace rtuf
