/*
 * rtuf_oracle.h -- interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * See rtuf_oracle.c for what is restated and from where.
 */
#ifndef RTUF_ORACLE_H_
#define RTUF_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RTUF_ORACLE_OP_NONE = 0, RTUF_ORACLE_OP_SCALE = 1, RTUF_ORACLE_OP_TRANSLATE = 2 };

/* One GL draw call of a renderable (/root/reference/src/renderable.cpp:80-131, :424-452):
 *   glPushMatrix; glMultMatrixd(link_tf); [glScalef | glTranslatef](op); draw triangles; glPopMatrix */
typedef struct {
  double link_tf[16];      /* (link_to_fixed * link_offset).getOpenGLMatrix(), column-major */
  int32_t pre_op;          /* RTUF_ORACLE_OP_* applied after link_tf */
  float op[3];
  const float *verts;      /* nverts x (x,y,z) */
  int32_t nverts;
  const uint32_t *tris;    /* ntris x 3 indices, in draw order */
  int32_t ntris;
} rtuf_oracle_draw;

typedef struct {
  int32_t width, height;
  const float *depth;              /* width*height float32 metres, row 0 first */
  float z_near, z_far;             /* 0.1, 8 (src/urdf_filter.cpp:53-54) */
  float max_diff;                  /* depth_distance_threshold */
  float replace_value;             /* filter_replace_value */
  double projection[16];           /* getProjectionMatrix result, column-major */
  double camera_offset_inv[16];    /* inverse(camera_offset).getOpenGLMatrix() */
  double camera_tf[16];            /* camera_transform (cam <- fixed, incl. tx/ty shift) */
  const rtuf_oracle_draw *draws;
  int32_t ndraws;
} rtuf_oracle_frame;

typedef struct {
  float *zwin;        /* optional width*height: float window z of the winning fragment */
  int32_t *prim;      /* optional width*height: winning source-triangle id (-2 background, -1 none) */
  long n_tris_in, n_tris_setup, n_frags;
} rtuf_oracle_debug;

typedef struct {
  int32_t vs_fma;         /* vertex shader mat*vec uses fused multiply-add           */
  int32_t vp_fma;         /* viewport transform of shaded vertices is fused          */
  int32_t clip_vp_fma;    /* viewport transform of clipper-made vertices is fused    */
  int32_t interp_fma;     /* z interpolation a0 + dzdx*x + dzdy*y is fused           */
  int32_t frag_div_rcp;   /* a/b in the fragment shader evaluated as a * (1/b)       */
  int32_t cw_swap_12;     /* clockwise triangles: swap v1,v2 instead of v0,v1        */
  int32_t edge_rule_flip; /* horizontal edges inclusive on the high-row side         */
  int32_t clip_old_t;     /* clipper always interpolates from the outside vertex (Mesa < 21) */
} rtuf_oracle_variants;

#define RTUF_ORACLE_VARIANTS_DEFAULT { 0, 1, 0, 1, 0, 0, 0, 0 }

void rtuf_oracle_set_variants(const rtuf_oracle_variants *v);
void rtuf_oracle_get_variants(rtuf_oracle_variants *v);

/* One frame of RealtimeURDFFilter::filter().  mask may be NULL (need_mask_ == false).
 * Returns 0 on success. */
int rtuf_oracle_filter(const rtuf_oracle_frame *in, float *masked_depth, uint8_t *mask,
                       rtuf_oracle_debug *dbg);

/* Throughput leg of bench.py's cpu_baseline ("all cores"): n_threads POSIX threads filter the frames [0, n_frames)
 * cyclically for `seconds` (a shared atomic counter hands out frames; every thread writes into private scratch
 * planes, the results are discarded).  Returns the number of frames filtered (< 0 on failure); *elapsed_out gets the
 * wall time including the last frames in flight at the deadline. */
long rtuf_oracle_filter_throughput(const rtuf_oracle_frame *frames, int n_frames, double seconds, int n_threads,
                                   double *elapsed_out);

void rtuf_oracle_compose_mvp(const double *projection, const double *camera_offset_inv,
                             const double *camera_tf, const double *link_tf,
                             int pre_op, const float *op, float *out_mvp);

#ifdef __cplusplus
}
#endif
#endif
