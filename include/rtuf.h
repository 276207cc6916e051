/*
 * rtuf.h -- C ABI of the MI355X-native realtime URDF depth self-filter.
 *
 * This is the drop-in boundary for the hot path of blodow/realtime_urdf_filter:
 *   RealtimeURDFFilter::filter() -> textureBufferFromDepthBuffer() -> render()
 *   -> URDFRenderer::render() -> Renderable*::render() -> urdf_filter.{vert,frag}
 *   -> glGetTexImage read-back.
 * Every entry point cites the reference interface it replaces (paths relative to
 * the reference checkout).  The reference has no C ABI of its own -- its boundary
 * is the C++ class surface of include/realtime_urdf_filter/urdf_filter.h:51-143 and
 * urdf_renderer.h:45-73 -- so these are the calls a C++ facade (include/
 * realtime_urdf_filter_amd/urdf_filter.hpp), a ROS adapter or a ctypes/cgo/JNI stub
 * binds (INTEGRATION.md shows the bindings).
 *
 * Conventions: plain pointers and sizes only; every function returns an int status
 * (RTUF_OK = 0, negative = error) and never throws; rtuf_last_error() returns the
 * message of the last failure on that context (or of rtuf_create when ctx == NULL).
 * The caller owns every host buffer; the library owns device memory.  One context is
 * bound to one GPU and used from one host thread at a time; several contexts (one per
 * GPU) may run concurrently.  All matrices are OpenGL style: 16 values, column-major.
 *
 * There is NO CPU fallback: without a visible gfx950 device (hipDeviceProp_t::gcnArchName) rtuf_create()
 * fails with RTUF_ERR_NO_DEVICE.
 */
#ifndef RTUF_H_
#define RTUF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTUF_ABI_VERSION 6

typedef struct rtuf_context rtuf_context;

enum {
  RTUF_OK = 0,
  RTUF_ERR_INVALID = -1,      /* bad argument / call order                      */
  RTUF_ERR_NO_DEVICE = -2,    /* no usable HIP device (no CPU fallback exists)  */
  RTUF_ERR_HIP = -3,          /* HIP runtime error, see rtuf_last_error         */
  RTUF_ERR_OOM = -4,          /* host or device allocation failed               */
  RTUF_ERR_CAPACITY = -5,     /* a per-tile bin overflowed even after regrowth  */
  RTUF_ERR_STATE = -6         /* models not finalised / frame not ready         */
};

/* Matrix operation a renderable applies between glMultMatrixd(link) and its draw call
 * (src/renderable.cpp:95 glTranslatef for cylinders, :128 / :427 glScalef for the second
 * box and for meshes). */
enum { RTUF_OP_NONE = 0, RTUF_OP_SCALE = 1, RTUF_OP_TRANSLATE = 2 };

/* rtuf_params.flags */
enum {
  /* (bit 0 was RTUF_FLAG_BACKGROUND_QUAD up to ABI 2.  The reference always draws the background quad at 0.99*far,
   * src/urdf_filter.cpp:591-596, and so does this library: there is nothing to switch.) */
  RTUF_FLAG_TWO_KERNEL      = 1u << 1,  /* rasteriser writes the z-surface to HBM and a separate compare kernel consumes it
                                           (default: compare fused into the tile kernel, the z-surface never leaves LDS)        */
  RTUF_FLAG_STRICT_GRID     = 1u << 2,  /* ABI 6: every set-up launch takes the worst-case grid (every chunk visible in every
                                           stream) instead of one sized from the previous batch's work lists, so a batch is never
                                           run again because its work list outgrew the estimate (RTUF_STATUS_GRID_SHORT cannot
                                           happen).  Costs empty workgroups: for consumers that read device planes before the
                                           host has retired the batch (rtuf_batch_status_device) and want one reason less for a
                                           provisional batch */
  RTUF_FLAG_DEFAULT = 0
  /* Any other bit makes rtuf_create / rtuf_set_params fail with RTUF_ERR_INVALID.  (The kernels' timing experiments
   * live only in a separate library built with -DRTUF_ABLATE for scripts/ablate_*.sh; the product has no such code.) */
};

/* Replaces the constructor's rosparam parsing (src/urdf_filter.cpp:43-118) and the
 * hard-coded clip planes (:53-54). */
typedef struct {
  float near_plane;                 /* 0.1  */
  float far_plane;                  /* 8.0  */
  float depth_distance_threshold;   /* rosparam depth_distance_threshold -> shader uniform max_diff (:630) */
  float filter_replace_value;       /* rosparam filter_replace_value     -> shader uniform replace_value (:631) */
  uint32_t flags;                   /* RTUF_FLAG_* */
  uint32_t bin_capacity;            /* triangles per (stream, screen tile) bin to start with; 0 = automatic (1024); grown from what
                                       the first batches ask for */
  uint32_t max_inflight_streams;    /* streams rasterised per internal launch group (= streams a raster lane's bins are sized for);
                                       0 = automatic: max_streams divided by the raster lanes (256 streams, three lanes: three
                                       groups of 86, one per lane), the whole batch up to 1024 with one lane; fewer if the bins
                                       would exceed a third of the free device memory or memory_limit_mb.  Smaller groups trade
                                       frames/s for memory: 256 VGA streams of a 250 k-triangle robot run at 514-525 k frames/s
                                       in 8.9 GB with three groups of 86, at 480 k in 4.5 GB with six groups of 43 (490 k in
                                       4.45 GB with two lanes and four groups of 64) */
  uint32_t pipelines;               /* 0 / 1: one raster pipeline.  2..4: that many complete pipelines (HIP streams, bins,
                                       staging, geometry copy) inside the context; batches alternate between them, so the
                                       small and low-occupancy kernels of one batch (pose stage, cull, clip, kernel tails)
                                       run underneath another batch's set-up / tile kernels (+10 % frames/s with 2 on the
                                       256-stream VGA workload).  Setters apply to all pipelines; up to 2 x pipelines batches
                                       may be in flight; rtuf_filter() and the debug read-backs use the first / the last-used
                                       pipeline.  Fixed at rtuf_create. */
  /* ABI 5 */
  uint32_t raster_lanes;            /* 0 = automatic (3; at most 3).  A raster lane is a HIP stream plus a set of tile bins sized for
                                       ONE launch group.  With several, a batch of >= 32 streams is split into launch groups (a
                                       multiple of the lanes) that alternate between the lanes, which run free of each other: every
                                       kernel's ramp, tail and launch gap is filled by the other lanes' kernels.  Smaller batches
                                       take one lane each, in turn.  Three lanes and the pose stage's stream are one HIP stream per
                                       hardware queue of the runtime's default four (GPU_MAX_HW_QUEUES).  1 = one lane, every kernel
                                       alone on the GPU (what per-kernel timings and rooflines should be measured with).  Fixed at
                                       rtuf_create. */
  uint32_t memory_limit_mb;         /* upper bound of the rasteriser's working set (tile bins of all lanes), MiB; 0 = a third of
                                       the free device memory at rtuf_finalize_models.  When the bins a scene needs would exceed it
                                       the launch groups shrink (more, smaller launches) instead of the call failing; when even
                                       one stream's bins are larger the bins are grown all the same (up to what the device has)
                                       and rtuf_stats.over_memory_limit says so. */
  uint32_t reserved[2];
} rtuf_params;

void rtuf_default_params(rtuf_params *p);
int rtuf_abi_version(void);

/* ctor + initGL() + initFrameBufferObject() (src/urdf_filter.cpp:43-118, :386-456):
 * binds a GPU, allocates the per-stream frame resources for `max_streams` concurrent
 * width x height depth streams.  No GL / X11 context exists anywhere. */
int rtuf_create(rtuf_context **out, int device_id, int width, int height, int max_streams,
                const rtuf_params *params);
void rtuf_destroy(rtuf_context *ctx);
const char *rtuf_last_error(const rtuf_context *ctx);

/* Uniforms can change between frames like the public data members of the reference
 * (include/realtime_urdf_filter/urdf_filter.h:112-135).  far_plane is the exception: the background
 * quad is part of the finalized geometry, so changing it after rtuf_finalize_models returns
 * RTUF_ERR_STATE. */
int rtuf_set_params(rtuf_context *ctx, const rtuf_params *params);

/* ---- geometry: loaded once into device buffers ------------------------------------
 * loadModels() -> URDFRenderer ctor -> process_link -> Renderable* ctor
 * (src/urdf_filter.cpp:127-197, src/urdf_renderer.cpp:44-169, src/renderable.cpp:100-170,
 * :306-415).  A model is one `models[i]` entry (one URDFRenderer); a link is one
 * Renderable (one glPushMatrix/glPopMatrix bracket, posed per frame by its
 * link_to_fixed * link_offset matrix); a draw is one GL draw call inside that bracket.
 * Geometry is already tessellated into indexed triangles, in draw order. */
int rtuf_add_model(rtuf_context *ctx);                       /* returns model id >= 0 */
int rtuf_add_link(rtuf_context *ctx, int model);             /* returns link id >= 0 (index within the model) */
int rtuf_add_draw(rtuf_context *ctx, int model, int link, int pre_op, const float op_xyz[3],
                  const float *vertices_xyz, int n_vertices,
                  const uint32_t *triangles, int n_triangles);
/* Uploads everything added so far (VBO/IBO creation, src/renderable.cpp:167-169, :343-349). */
int rtuf_finalize_models(rtuf_context *ctx);
int rtuf_num_links(const rtuf_context *ctx, int model);
int64_t rtuf_num_triangles(const rtuf_context *ctx);
/* Vertices resident on the device after rtuf_finalize_models: per set-up chunk (<= 256 triangles) every distinct position
 * once (bit-identical positions of one draw are merged: STL facets bring three vertices of their own each, which the
 * reference uploads as they are, src/renderable.cpp:339-350; the image cannot tell), plus the background quad's four. */
int64_t rtuf_num_vertices(const rtuf_context *ctx);

/* Which models a stream renders (default: all).  Lets independent robots share a context
 * (BASELINE config 5); the reference renders every loaded model for its single stream. */
int rtuf_set_stream_models(rtuf_context *ctx, int stream, const int *model_ids, int n_models);

/* ---- per-frame pose inputs -------------------------------------------------------- */
/* Camera of one stream: projection = getProjectionMatrix() result (src/urdf_filter.cpp:459-501);
 * camera_offset_inv = inverse(camera_offset).getOpenGLMatrix() (:602-604);
 * camera_tf = camera_transform.getOpenGLMatrix() after the tx/ty origin shift (:607-614).
 * Composed on the device in float32 in OpenGL matrix-stack order. */
int rtuf_set_camera(rtuf_context *ctx, int stream, const double projection[16],
                    const double camera_offset_inv[16], const double camera_tf[16]);
/* K (fx,fy,cx,cy,Tx,Ty) -> projection[16] and camera_tx/ty, exactly getProjectionMatrix. */
void rtuf_projection_from_intrinsics(double fx, double fy, double cx, double cy, double Tx, double Ty,
                                     int width, int height, double near_plane, double far_plane,
                                     double projection_out[16], double *camera_tx, double *camera_ty);
/* Link poses of one model for one stream: n_links matrices
 * (link_to_fixed * link_offset).getOpenGLMatrix(), i.e. what Renderable::applyTransform
 * passes to glMultMatrixd (src/renderable.cpp:59-68) after URDFRenderer::update_link_transforms
 * (src/urdf_renderer.cpp:173-190). */
int rtuf_set_link_poses(rtuf_context *ctx, int stream, int model, const double *link_tf, int n_links);

/* Batched forms of the two setters above (one FFI crossing per frame for N streams):
 * matrices are [n_streams][16] resp. [n_streams][n_links][16], for streams
 * first_stream .. first_stream + n_streams - 1.  Any of the three camera arrays may be NULL
 * (that matrix keeps its previous value for all n streams). */
int rtuf_set_cameras(rtuf_context *ctx, int first_stream, int n_streams, const double *projection,
                     const double *camera_offset_inv, const double *camera_tf);
int rtuf_set_link_poses_batch(rtuf_context *ctx, int first_stream, int n_streams, int model,
                              const double *link_tf, int n_links);
/* camera_tx_ / camera_ty_ of getProjectionMatrix (src/urdf_filter.cpp:481-482) for streams whose camera transform
 * is derived on the device (rtuf_set_joint_positions with camera_frame >= 0): the forward-kinematics kernel moves
 * the camera origin by tx along the transform's x axis and ty along its y axis, src/urdf_filter.cpp:607-611.
 * (A camera_tf handed to rtuf_set_camera(s) has the shift applied by the caller already and is not touched.)
 * NULL = 0 for all n streams. */
int rtuf_set_camera_shift(rtuf_context *ctx, int first_stream, int n_streams, const double *camera_tx,
                          const double *camera_ty);

/* ---- on-device forward kinematics --------------------------------------------------
 * Replaces the per-renderable TF lookups of URDFRenderer::update_link_transforms
 * (src/urdf_renderer.cpp:173-190) for robots whose link poses follow from joint positions: the
 * kinematic tree of a model is uploaded once, per frame only the joint positions cross the bus and
 * a kernel computes every  fixed<-link  transform and the matrices of rtuf_set_link_poses on the GPU
 * (double precision, same multiplication order as the host-side forward kinematics).
 *   n_frames frames, parent[i] < i or -1 for the root; joint_type 0 fixed, 1 revolute/continuous,
 *   2 prismatic; joint_origin[i] = parent<-child transform at q = 0 (column-major); joint_axis[i] in
 *   the child frame; link_frame[l] = frame the l-th link (renderable) of the model is attached to;
 *   link_offset[l] = the URDF <origin> of its visual/collision (Renderable::link_offset). */
enum { RTUF_JOINT_FIXED = 0, RTUF_JOINT_REVOLUTE = 1, RTUF_JOINT_PRISMATIC = 2 };
int rtuf_set_kinematics(rtuf_context *ctx, int model, int n_frames, const int32_t *parent,
                        const int32_t *joint_type, const double *joint_origin, const double *joint_axis,
                        const int32_t *link_frame, const double *link_offset, int n_links);
/* Joint positions q[n_streams][n_frames] (entries of fixed joints are ignored) and, optionally, the
 * pose of the root frame in the fixed frame root_tf[n_streams][16] (NULL = identity).  With
 * camera_frame >= 0 the camera transform of each stream becomes inverse(fixed<-camera_frame)
 * (a camera mounted on the robot, shifted by rtuf_set_camera_shift); with -1 it is what rtuf_set_camera(s) set
 * (also again after an earlier call with camera_frame >= 0).  Overrides
 * rtuf_set_link_poses for this model on these streams until rtuf_set_link_poses is called again. */
int rtuf_set_joint_positions(rtuf_context *ctx, int first_stream, int n_streams, int model, const double *q,
                             const double *root_tf, int camera_frame);
/* Debug / test access: the link matrices ([n_streams][total links][16]) and camera transforms
 * ([n_streams][16]) the last batch was rendered with. */
int rtuf_debug_read_poses(rtuf_context *ctx, int n_streams, double *link_tf_out, double *cam_tf_out);

/* ---- the hot path ------------------------------------------------------------------ */
/* filter() for n streams at once (stream ids 0..n-1), host buffers:
 * depth_in[s]: width*height float32 metres, row 0 first (src/urdf_filter.cpp:233-234);
 * masked_out[s]: width*height float32 (getMaskedDepth()); mask_out may be NULL, or
 * mask_out[s] NULL (need_mask_ == false, :226-230), else width*height bytes 0/255. */
int rtuf_filter_batch(rtuf_context *ctx, int n_streams, const float *const *depth_in,
                      float *const *masked_out, uint8_t *const *mask_out);
/* Same with device-resident planes: depth/masked are [n][H][W] float32, mask [n][H][W] u8 or
 * NULL.  Asynchronous on the context's stream.  Up to two batches may be in flight: a call made
 * while two are pending first retires the oldest (as rtuf_sync does for all of them), so a caller
 * that keeps enqueueing never leaves the GPU idle during the host round trip.  The buffers of a
 * batch must stay valid and unmodified until it is retired (a bin regrowth runs it again);
 * the per-frame pose setters -- rtuf_set_joint_positions, rtuf_set_camera(s), rtuf_set_camera_shift,
 * rtuf_set_link_poses[_batch] -- may be called at any time: their staging is a ring of sets, one per batch in
 * flight plus one being written, so the next frame's poses are staged while the GPU filters this one (only a call
 * that switches a stream between forward kinematics and explicit matrices waits).  Every other setter (parameters,
 * model selection, kinematic trees) waits for the batches in flight.  rtuf_sync() retires everything in flight.
 * ORDER OF BATCHES IN FLIGHT: none.  With more than one raster lane (the default) consecutive batches run on different
 * HIP streams -- a batch of one launch group takes the lanes in turn -- and nothing orders the kernels of one batch behind
 * those of the batch before it.  Two batches in flight must therefore not share an output plane, and a batch must not read
 * as its sensor plane what the batch before it writes: retire the first (rtuf_wait_oldest / rtuf_sync) or run the context with
 * rtuf_params.raster_lanes = 1, where batches execute in the order they were enqueued. */
int rtuf_filter_batch_device(rtuf_context *ctx, int n_streams, const float *d_depth,
                             float *d_masked, uint8_t *d_mask);
/* 16UC1 variants: depth in / out as uint16 millimetres with the reference's conversions fused into the
 * kernels (filter_callback, src/urdf_filter.cpp:280-289: convertTo(CV_32F, 0.001); :308-312:
 * convertTo(CV_16U, 1000.0), i.e. filtered pixels become filter_replace_value * 1000).  Halves the
 * HBM and PCIe bytes per pixel (5 instead of 9). */
int rtuf_filter_batch_u16(rtuf_context *ctx, int n_streams, const uint16_t *const *depth_mm_in,
                          uint16_t *const *masked_mm_out, uint8_t *const *mask_out);
/* Asynchronous forms of the two host-plane calls: the planes go up on one copy stream and come back on
 * another, so with two batches in flight the PCIe transfers of one overlap the kernels of the other
 * (the reference uploads, renders and reads back serially: src/urdf_filter.cpp:332-353, :729-735).
 * All planes of a batch must stay valid until it is retired -- by rtuf_wait_oldest, rtuf_sync, or
 * the second-next asynchronous call -- and should be pinned memory (rtuf_host_alloc); with pageable
 * memory the calls are correct but block for the transfers.  Planes that follow each other in memory
 * are moved as one transfer. */
int rtuf_filter_batch_async(rtuf_context *ctx, int n_streams, const float *const *depth_in,
                            float *const *masked_out, uint8_t *const *mask_out);
int rtuf_filter_batch_u16_async(rtuf_context *ctx, int n_streams, const uint16_t *const *depth_mm_in,
                                uint16_t *const *masked_mm_out, uint8_t *const *mask_out);
/* ---- mask-only output, one bit per pixel -----------------------------------------------------------------
 * The reference publishes the mask on its own topic (src/urdf_filter.cpp:321-329) and the masked depth is a pure
 * function of sensor plane and mask: masked = mask ? filter_replace_value : sensor (tests check this bit for bit on
 * every full-plane result).  These calls return ONLY the mask, packed: pixel (x, y) of a stream is bit x % 32 of
 * 32-bit word y * ceil(width / 32) + x / 32 (rtuf_mask_bits_words() words per stream).  Device-to-host traffic per
 * VGA frame drops from 1.54 MB to 38 KB, which turns the PCIe-bound host-plane path from download-bound into
 * upload-bound; rtuf_expand_mask_bits() rebuilds masked depth and/or the byte mask of a frame on the host where and
 * when a consumer needs them.  Fused mode only; width must be a multiple of 4.  Exactness: the identity above holds
 * wherever the background quad covers the image -- every getProjectionMatrix() camera.  For a projection where it
 * does not (pixels the reference leaves at the GL clear colour 0), retiring the batch fails with RTUF_ERR_STATE
 * instead of returning bits that would expand to something else. */
size_t rtuf_mask_bits_words(int width, int height);
int rtuf_filter_batch_bits_async(rtuf_context *ctx, int n_streams, const float *const *depth_in, uint32_t *const *bits_out);
int rtuf_filter_batch_bits_u16_async(rtuf_context *ctx, int n_streams, const uint16_t *const *depth_mm_in,
                                     uint32_t *const *bits_out);
/* device-resident planes: depth [n][H][W], bits [n][rtuf_mask_bits_words()] */
int rtuf_filter_batch_device_bits(rtuf_context *ctx, int n_streams, const float *d_depth, uint32_t *d_bits);
int rtuf_filter_batch_device_bits_u16(rtuf_context *ctx, int n_streams, const uint16_t *d_depth_mm, uint32_t *d_bits);
/* Host helper (no GPU, no context): one frame's masked depth (float32, or uint16 millimetres with the reference's
 * convertTo arithmetic when is_u16) and/or 0/255 byte mask from its sensor plane and mask bits.  Either output may
 * be NULL. */
int rtuf_expand_mask_bits(const void *depth_in, int is_u16, const uint32_t *bits, int width, int height,
                          float replace_value, void *masked_out, uint8_t *mask_out);

/* Waits for the oldest batch in flight (device or host planes); its outputs are complete afterwards.
 * RTUF_OK when nothing is pending. */
int rtuf_wait_oldest(rtuf_context *ctx);
/* Pinned host memory for the asynchronous calls.  rtuf_host_free waits for the batches in flight. */
int rtuf_host_alloc(rtuf_context *ctx, size_t bytes, void **out);
int rtuf_host_free(rtuf_context *ctx, void *ptr);
int rtuf_filter_batch_device_u16(rtuf_context *ctx, int n_streams, const uint16_t *d_depth_mm,
                                 uint16_t *d_masked_mm, uint8_t *d_mask);
/* Exact single-stream shape of RealtimeURDFFilter::filter(buffer, glTf, w, h)
 * (include/realtime_urdf_filter/urdf_filter.h:70-72): stream 0, projection given per call,
 * results kept in library-owned host buffers like masked_depth_/mask_. */
int rtuf_filter(rtuf_context *ctx, const unsigned char *buffer, const double *projection,
                int width, int height);
const float *rtuf_get_masked_depth(const rtuf_context *ctx);      /* getMaskedDepth() */
const uint8_t *rtuf_get_mask(const rtuf_context *ctx);            /* mask_ */

int rtuf_sync(rtuf_context *ctx);
/* The HIP stream the raster stage of this context is enqueued on -- defined only for a context of ONE raster lane and ONE
 * pipeline (rtuf_params.raster_lanes = 1, pipelines <= 1).  Every other context spreads a batch over several internal
 * streams and returns NULL here: do NOT pass that NULL to hipStreamWaitEvent / hipEventRecord (NULL is the legacy default
 * stream there: nothing fails and the ordering is simply wrong).  Callers that order their own device work behind the
 * batches use rtuf_order_stream_after_batches() instead, which is right for every configuration. */
void *rtuf_stream(rtuf_context *ctx);
/* Makes `hip_stream` (a hipStream_t of the context's device; NULL = the legacy default stream) wait, on the device, for
 * every batch enqueued on this context so far: all raster lanes, all pipelines, and for host-plane batches their
 * downloads.  Does not block the host and does not retire anything (the batches' buffers stay owned by the library until
 * rtuf_wait_oldest / rtuf_sync).  ABI 5.
 *
 * WHAT THIS DOES NOT GUARANTEE: that the planes are final.  The rasteriser's working buffers (tile bins, clip list, many-tile
 * list, the set-up grid) are sized from what earlier batches needed; whether a batch outgrew one of them is known only once
 * its kernels have run, and the HOST acts on it when it retires the batch (rtuf_wait_oldest / rtuf_sync / the second-next
 * filter call): it enlarges the buffer and runs the batch again into the same planes.  Until then the planes of such a batch
 * hold pixels of a rasterisation that dropped triangles.  The reference's filter() returns with final pixels
 * (src/urdf_filter.cpp:237, :729-735); so do all calls here that retire through the host.  A consumer that reads device
 * planes behind this call, before the host has retired the batch, must check the batch's STATUS WORD on the device: */
int rtuf_order_stream_after_batches(rtuf_context *ctx, void *hip_stream);
/* ABI 6.  Device address of the status word (one uint32) of the batch the most recent rtuf_filter_batch* call enqueued
 * (with pipelines: of the pipeline that took it).  The planes of that batch are final IF AND ONLY IF the word reads 0
 * after the batch's kernels -- i.e. on a stream ordered behind them with rtuf_order_stream_after_batches.  Otherwise:
 *   bits 0..15   launch groups of the batch that have not finished yet (non-zero only for a reader that is NOT ordered behind
 *                the batch, or while the library runs the batch again)
 *   RTUF_STATUS_*_OVERFLOW / _GRID_SHORT   a working buffer was too small: the planes are provisional, and the library will
 *                run the batch again when the host retires it; the word reads 0 once that run has finished
 *   RTUF_STATUS_UNCOVERED   mask-bits output only: the bits do not expand to the reference's planes (retiring the batch fails)
 * The address belongs to one of the context's two batch slots: it is reused by the second-next batch, whose first kernel
 * overwrites the word -- read it in the same stream-ordered work that reads the planes.  rtuf_stats.batch_status is the
 * host's mirror for the batch retired last (what its first run left in the word). */
enum {
  RTUF_STATUS_PENDING_MASK  = 0xffffu,
  RTUF_STATUS_BIN_OVERFLOW  = 1u << 16,   /* a (stream, tile) bin held more records or fragments than its capacity      */
  RTUF_STATUS_CLIP_OVERFLOW = 1u << 17,   /* more triangles crossed a frustum plane than the clip list holds             */
  RTUF_STATUS_LIST_OVERFLOW = 1u << 18,   /* more many-tile records than their list holds                               */
  RTUF_STATUS_GRID_SHORT    = 1u << 19,   /* a launch group's work list outgrew the set-up grid sized from the last batch */
  RTUF_STATUS_UNCOVERED     = 1u << 20    /* rtuf_filter_batch*_bits*: a pixel no fragment reached                       */
};
int rtuf_batch_status_device(rtuf_context *ctx, const uint32_t **d_status);

/* Counters of the last batch and kernel timings measured with HIP events on the context's
 * stream (replaces the wall-clock statistics of src/urdf_filter.cpp:239-266). */
typedef struct {
  uint64_t triangles_submitted;     /* triangles x streams handed to set-up            */
  uint64_t triangles_binned;        /* (sub-)triangles that reached at least one tile  */
  uint64_t bin_entries;             /* records written to tile bins                    */
  uint64_t triangles_clipped;       /* triangles that went through the clipper         */
  uint32_t max_bin_fill;            /* largest bin of the last batch                   */
  uint32_t bin_capacity;
  uint32_t regrowths;               /* times the bins were enlarged and a batch re-run */
  uint32_t max_fbin_fill;           /* largest fragment bin of the last batch          */
  uint64_t fragments_binned;        /* covered pixels of small (<= 4x4 px box) triangles binned as fragments */
  float ms_pose, ms_setup, ms_raster, ms_compare, ms_total;   /* last timed batch       */
  float ms_clip;                    /* clip kernel alone (timing modes 2 / 3; in mode 1 it is part of ms_setup) */
  uint64_t timed_batches;           /* batches retired since rtuf_enable_timing, and the sums  */
  double sum_ms_pose, sum_ms_setup, sum_ms_raster, sum_ms_compare, sum_ms_total;   /* of their times */
  double sum_ms_clip;
  /* ABI 4 */
  uint64_t device_bytes;            /* device memory the context holds right now (all pipelines)                      */
  uint64_t occluded_entries;        /* (record, tile) pairs never appended: they lie behind a triangle that covers the tile */
  uint32_t cover_tiles;             /* tiles whose initial depth keys came from a triangle covering the whole tile     */
  uint32_t exact_tiles;             /* tiles that ran the exact-z pass (a winner with window z <= 0.5)                 */
  uint32_t work_items;              /* set-up work items (chunk x up to 3 streams that see it) of the last batch       */
  uint32_t zero_survivor_items;     /* ... of which no triangle survived the clip / sub-pixel culls                    */
  uint32_t cover_pass;              /* 1: the last batch ran the cover pass (on while scenes have whole-tile triangles,
                                       switched off after three batches without one, probed again every 64th batch)     */
  uint32_t reserved0;
  uint64_t raster_atomics;          /* instrumented builds (-DRTUF_COUNT) only: depth tests the tile kernel issued      */
  uint64_t drawn_pixels;            /* instrumented builds only: pixels whose final depth key is not the background's   */
  /* ABI 5 */
  uint32_t raster_lanes;            /* raster lanes of the context (of each pipeline)                                   */
  uint32_t launch_group;            /* streams a lane's bins are sized for right now (shrinks when memory is short)      */
  uint32_t groups_last_batch;       /* launch groups the last batch was split into                                       */
  uint32_t graphs_enabled;          /* 1 while small batches replay captured hipGraphs (pipeline children only); the library
                                       switches them off for good when captures keep evicting live entries               */
  uint64_t graph_hits, graph_misses;/* replays / captures so far                                                          */
  uint32_t lanes_side_by_side;      /* 1: the lanes' HIP streams and the pose stage's stream were measured to execute side by
                                       side at rtuf_create (with one lane: the lane's and the pose stage's).  0: every stream the
                                       runtime handed out shared a hardware queue with one of them -- everything then works, one
                                       kernel after the other; raise GPU_MAX_HW_QUEUES (HIP runtime, default 4)  */
  /* ABI 6 */
  uint32_t batch_status;            /* RTUF_STATUS_* bits the FIRST run of the batch retired last left in its device status word
                                       (0: its planes were final as soon as its kernels had run).  A batch that was run again only
                                       because an OLDER batch in flight overflowed is retired on its re-run's counters: its mirror
                                       then reads what that run left (0, batch_reruns 0) although a device-side consumer saw the
                                       first run's word -- the device word, not this mirror, is what such a consumer goes by       */
  uint32_t batch_reruns;            /* times that batch (and everything in flight behind it) was run again before it was retired */
  uint32_t over_memory_limit;       /* 1: the tile bins exceed rtuf_params.memory_limit_mb (or the automatic third of the free
                                       memory): one stream's bins alone are larger, and launch groups cannot shrink below one   */
  uint32_t copy_streams_side_by_side;  /* host-plane calls: 1 once the upload and download streams were measured to run beside the
                                       raster lanes' (first such call), 0 before that or when they share a hardware queue with a lane
                                       (copies then wait behind that lane's kernels).  RTUF_QUEUE_PROBE=0 in the environment skips
                                       all of these measurements (rtuf_create is ~10 ms faster, the flags then read 1 unmeasured)  */
} rtuf_stats;
int rtuf_get_stats(rtuf_context *ctx, rtuf_stats *out);
/* Per-kernel HIP-event timing (off by default: every event costs a few microseconds of stream
 * time).  on = 1: every stage (ms_pose, ms_setup = cull .. clip, ms_raster, ms_compare, ms_total);  on = 2: the
 * raster stage's kernels one by one: ms_setup (set-up kernel), ms_clip (clip kernel), ms_raster (tile kernel),
 * ms_compare (two-kernel mode);  on = 3: as 2, but only every eighth batch is timed (timed_batches and the sums
 * count those). */
int rtuf_enable_timing(rtuf_context *ctx, int on);

/* Debug / test access: copy the z-surface of the last batch (float window z of the winning
 * fragment per pixel, [n][H][W]) to the host.  Only valid with RTUF_FLAG_TWO_KERNEL. */
int rtuf_debug_read_zsurface(rtuf_context *ctx, int n_streams, float *host_out);

#ifdef __cplusplus
}
#endif
#endif /* RTUF_H_ */
