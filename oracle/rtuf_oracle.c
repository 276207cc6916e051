/*
 * rtuf_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's depth self-filter frame
 *   RealtimeURDFFilter::filter() -> render()        /root/reference/src/urdf_filter.cpp:207-267, :503-744
 *   Renderable::applyTransform / *::render()        /root/reference/src/renderable.cpp:59-131, :424-452
 *   urdf_filter.vert / urdf_filter.frag             /root/reference/include/shaders/urdf_filter.vert:4-9,
 *                                                   /root/reference/include/shaders/urdf_filter.frag:14-36
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file; the product path (realtime_urdf_filter_amd/csrc) never links it.
 *
 * The reference does its arithmetic inside an OpenGL driver.  The driver that
 * executes it in the development container -- and therefore defines
 * "bit-identical" for this project -- is Mesa 23.2.1 llvmpipe (third-party,
 * binary only; NOT under /root/reference).  The pieces of the GL pipeline the
 * reference relies on are restated here from Mesa's published algorithm and
 * pinned empirically against the real thing (oracle/ref_gl/llvmpipe_oracle.c
 * runs the reference's GLSL verbatim on llvmpipe; tests/golden/ holds its
 * outputs):
 *
 *   1. float32 matrix stacks: glMultMatrixd rounds its argument to float and
 *      multiplies in float32 (m_matrix.c matmul4 ordering); glScalef /
 *      glTranslatef update the top matrix in place; MVP = P x MV in float32.
 *   2. vertex shader  clip = MVP * (x,y,z,1): column-major mul + 3 x mul-add.
 *   3. frustum clip test -w<=x,y,z<=w; Sutherland-Hodgman in clip space in the
 *      plane order +x,-x,+y,-y,near,far (draw_pipe_clip.c), fan re-triangulation.
 *   4. perspective divide + viewport (scale W/2,H/2,1/2).
 *   5. triangle set-up: vertices snapped to 1/256 px (round-half-even) after
 *      subtracting the 0.5 pixel centre, integer edge functions, top-left
 *      style fill rule (inclusive on low-x / low-row edges), z plane
 *      (a0, dz/dx, dz/dy) from the *unsnapped* float vertices.
 *   6. per fragment: z = a0 + dzdx*px + dzdy*py; depth test GL_LESS on
 *      z24 = rint(clamp(z,0,1) * 16777215); the shader sees the float z.
 *   7. fragment shader arithmetic of urdf_filter.frag:14-36.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (Makefile).
 * Every float operation below is written out so that the compiler cannot
 * change rounding; fused multiply-adds are explicit fmaf() calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rtuf_oracle.h"

/* Numerical-variant switches.  The defaults are the combination that
 * reproduces llvmpipe bit-for-bit on this project's fixtures; the others exist
 * so that tests can show they do NOT (and to document what was probed). */
static rtuf_oracle_variants g_var = RTUF_ORACLE_VARIANTS_DEFAULT;

void rtuf_oracle_set_variants(const rtuf_oracle_variants *v) { g_var = *v; }
void rtuf_oracle_get_variants(rtuf_oracle_variants *v) { *v = g_var; }

/* ------------------------------------------------------------------ */
/* 1. float32 matrix stack (Mesa src/mesa/math/m_matrix.c semantics)  */
/* ------------------------------------------------------------------ */
#define A(row, col) a[((col) << 2) + (row)]
#define B(row, col) b[((col) << 2) + (row)]
#define Pm(row, col) p[((col) << 2) + (row)]

static void matmul4(float *out, const float *a, const float *b)
{
  float p[16];
  for (int i = 0; i < 4; i++) {
    const float ai0 = A(i, 0), ai1 = A(i, 1), ai2 = A(i, 2), ai3 = A(i, 3);
    for (int j = 0; j < 4; j++) {
      float s = ai0 * B(0, j);
      s = s + ai1 * B(1, j);
      s = s + ai2 * B(2, j);
      s = s + ai3 * B(3, j);
      Pm(i, j) = s;
    }
  }
  memcpy(out, p, sizeof p);
}

static void mat_identity(float *m)
{
  memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}

static void mat_mult_d(float *top, const double *m)   /* glMultMatrixd */
{
  float f[16];
  for (int i = 0; i < 16; i++) f[i] = (float)m[i];
  matmul4(top, top, f);
}

static void mat_mult_f(float *top, const float *m) { matmul4(top, top, m); }

static void mat_scale(float *m, float x, float y, float z)   /* glScalef */
{
  m[0] *= x; m[4] *= y; m[8] *= z;
  m[1] *= x; m[5] *= y; m[9] *= z;
  m[2] *= x; m[6] *= y; m[10] *= z;
  m[3] *= x; m[7] *= y; m[11] *= z;
}

static void mat_translate(float *m, float x, float y, float z)   /* glTranslatef */
{
  m[12] = m[0] * x + m[4] * y + m[8] * z + m[12];
  m[13] = m[1] * x + m[5] * y + m[9] * z + m[13];
  m[14] = m[2] * x + m[6] * y + m[10] * z + m[14];
  m[15] = m[3] * x + m[7] * y + m[11] * z + m[15];
}

/* ------------------------------------------------------------------ */
/* 2. vertex shader (urdf_filter.vert:5)                              */
/* ------------------------------------------------------------------ */
static void vs_position(const float *mvp, const float *v, float *clip)
{
  for (int r = 0; r < 4; r++) {
    float s = mvp[r] * v[0];
    if (g_var.vs_fma) {
      s = fmaf(mvp[4 + r], v[1], s);
      s = fmaf(mvp[8 + r], v[2], s);
      s = fmaf(mvp[12 + r], 1.0f, s);
    } else {
      s = s + mvp[4 + r] * v[1];
      s = s + mvp[8 + r] * v[2];
      s = s + mvp[12 + r] * 1.0f;
    }
    clip[r] = s;
  }
}

/* ------------------------------------------------------------------ */
/* 3./4. clip test, viewport, clipper                                 */
/* ------------------------------------------------------------------ */
typedef struct {
  float clip[4];   /* clip-space position (shader output)  */
  float win[4];    /* window x, y, z and 1/w               */
  unsigned mask;   /* frustum clip mask                    */
} vtx;

static unsigned clipmask_of(const float *c)
{
  unsigned m = 0;
  if (c[0] > c[3]) m |= 1u;
  if (0.0f > c[0] + c[3]) m |= 2u;
  if (c[1] > c[3]) m |= 4u;
  if (0.0f > c[1] + c[3]) m |= 8u;
  if (0.0f > c[2] + c[3]) m |= 16u;
  if (c[2] > c[3]) m |= 32u;
  return m;
}

/* viewport for vertices produced by the (JIT-compiled) vertex stage */
static void viewport_vs(const float *scale, const float *trans, vtx *v)
{
  const float rhw = 1.0f / v->clip[3];
  for (int i = 0; i < 3; i++) {
    const float t = v->clip[i] * rhw;
    v->win[i] = g_var.vp_fma ? fmaf(t, scale[i], trans[i]) : t * scale[i] + trans[i];
  }
  v->win[3] = rhw;
}

/* viewport for vertices created by the clipper (C code in the driver) */
static void viewport_clip(const float *scale, const float *trans, vtx *v)
{
  const float oow = 1.0f / v->clip[3];
  for (int i = 0; i < 3; i++) {
    const float t = v->clip[i] * oow;
    v->win[i] = g_var.clip_vp_fma ? fmaf(t, scale[i], trans[i]) : t * scale[i] + trans[i];
  }
  v->win[3] = oow;
}

static const float k_planes[6][4] = {
  { -1, 0, 0, 1 }, { 1, 0, 0, 1 }, { 0, -1, 0, 1 }, { 0, 1, 0, 1 }, { 0, 0, 1, 1 }, { 0, 0, -1, 1 },
};

static float clipdist(const vtx *v, int plane)
{
  const float *b = k_planes[plane];
  const float *a = v->clip;
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

/* dst = out + t * (in - out), then project */
static void clip_interp(vtx *dst, float t, const vtx *out, const vtx *in, const float *scale, const float *trans)
{
  for (int i = 0; i < 4; i++) dst->clip[i] = (in->clip[i] - out->clip[i]) * t + out->clip[i];
  dst->mask = 0;
  viewport_clip(scale, trans, dst);
}

#define MAX_CLIPPED 16

/* ------------------------------------------------------------------ */
/* 5./6./7. set-up, rasterise, shade                                  */
/* ------------------------------------------------------------------ */
typedef struct {
  int w, h;
  const float *sensor;
  float z_near, z_far, max_diff, replace_value;
  uint32_t *z24;         /* depth buffer                              */
  float *masked;         /* colour attachment 1, red                  */
  uint8_t *mask;         /* colour attachment 3, red, as UNORM8       */
  float *dbg_zwin;       /* optional: winning fragment's float z      */
  int32_t *dbg_prim;     /* optional: winning primitive id            */
  int prim_id;           /* id of the source triangle being drawn     */
  /* statistics */
  long n_tris_in, n_tris_setup, n_frags;
} frame;

/* urdf_filter.frag:14-17 */
static float to_linear_depth(const frame *f, float d)
{
  const float n = f->z_near, fa = f->z_far;
  float num, off;
  if (g_var.frag_div_rcp) {
    num = (n * fa) * (1.0f / (n - fa));
    off = fa * (1.0f / (fa - n));
    return num * (1.0f / (d - off));
  }
  num = (n * fa) / (n - fa);
  off = fa / (fa - n);
  return num / (d - off);
}

static void shade(frame *f, int px, int py, float z)
{
  const size_t p = (size_t)py * f->w + px;
  /* depth test: GL_LESS on a 24-bit unorm buffer */
  float zc = z < 0.0f ? 0.0f : (z > 1.0f ? 1.0f : z);
  const uint32_t zi = (uint32_t)lrintf(zc * 16777215.0f);
  f->n_frags++;
  if (!(zi < f->z24[p])) return;
  f->z24[p] = zi;
  /* urdf_filter.frag:21-35 */
  const float sensor = f->sensor[p];
  const float virt = to_linear_depth(f, z);
  const int filt = sensor > (virt - f->max_diff);
  f->masked[p] = filt ? f->replace_value : sensor;
  if (f->mask) f->mask[p] = filt ? 255 : 0;
  if (f->dbg_zwin) f->dbg_zwin[p] = z;
  if (f->dbg_prim) f->dbg_prim[p] = f->prim_id;
}

static int iround_even(float x) { return (int)lrintf(x); }   /* cvtps2dq under default MXCSR */

static void setup_tri(frame *f, const vtx *va, const vtx *vb, const vtx *vc)
{
  const vtx *v0 = va, *v1 = vb, *v2 = vc;
  int x[3], y[3];
  const vtx *vv[3] = { v0, v1, v2 };
  for (int i = 0; i < 3; i++) {
    x[i] = iround_even((vv[i]->win[0] - 0.5f) * 256.0f);
    y[i] = iround_even((vv[i]->win[1] - 0.5f) * 256.0f);
  }
  int64_t dx01 = x[0] - x[1], dy01 = y[0] - y[1], dx20 = x[2] - x[0], dy20 = y[2] - y[0];
  int64_t area = dx01 * dy20 - dx20 * dy01;
  if (area == 0) return;
  if (area < 0) {
    /* make it counter-clockwise (in llvmpipe's sense) */
    if (g_var.cw_swap_12) {
      const vtx *t = v1; v1 = v2; v2 = t;
      int ti = x[1]; x[1] = x[2]; x[2] = ti; ti = y[1]; y[1] = y[2]; y[2] = ti;
    } else {
      const vtx *t = v0; v0 = v1; v1 = t;
      int ti = x[0]; x[0] = x[1]; x[1] = ti; ti = y[0]; y[0] = y[1]; y[1] = ti;
    }
  }
  f->n_tris_setup++;

  /* bounding box of pixel centres, inclusive */
  int minx = x[0] < x[1] ? x[0] : x[1]; if (x[2] < minx) minx = x[2];
  int maxx = x[0] > x[1] ? x[0] : x[1]; if (x[2] > maxx) maxx = x[2];
  int miny = y[0] < y[1] ? y[0] : y[1]; if (y[2] < miny) miny = y[2];
  int maxy = y[0] > y[1] ? y[0] : y[1]; if (y[2] > maxy) maxy = y[2];
  int bx0 = (minx + 255) >> 8, bx1 = (maxx - 1) >> 8;   /* ceil(min), and max exclusive when on a centre */
  int by0 = (miny + 255) >> 8, by1 = (maxy - 1) >> 8;
  if (bx0 < 0) bx0 = 0;
  if (by0 < 0) by0 = 0;
  if (bx1 > f->w - 1) bx1 = f->w - 1;
  if (by1 > f->h - 1) by1 = f->h - 1;
  if (bx1 < bx0 || by1 < by0) return;

  /* edge functions on the snapped vertices: inside <=> E_i > 0 after bias */
  int64_t ea[3], eb[3], ec[3];
  for (int i = 0; i < 3; i++) {
    const int j = (i + 1) % 3;
    const int64_t dcdx = y[i] - y[j];
    const int64_t dcdy = x[i] - x[j];
    int64_t c = dcdx * x[i] - dcdy * y[i];
    if (dcdx < 0) c++;                     /* left edge: inclusive */
    else if (dcdx == 0) {
      /* horizontal edge: inclusive when it is the low-row edge */
      if (g_var.edge_rule_flip ? (dcdy < 0) : (dcdy > 0)) c++;
    }
    ea[i] = dcdx; eb[i] = dcdy; ec[i] = c;
  }

  /* z plane from the float vertices, lp_state_setup.c ordering */
  const float x0c = v0->win[0] - 0.5f, y0c = v0->win[1] - 0.5f;
  const float fdx01 = v0->win[0] - v1->win[0], fdy01 = v0->win[1] - v1->win[1];
  const float fdx20 = v2->win[0] - v0->win[0], fdy20 = v2->win[1] - v0->win[1];
  const float e = fdx01 * fdy20, g = fdy01 * fdx20;
  const float ooa = 1.0f / (e - g);
  const float dy20_ooa = fdy20 * ooa, dy01_ooa = fdy01 * ooa;
  const float dx20_ooa = fdx20 * ooa, dx01_ooa = fdx01 * ooa;
  const float da01 = v0->win[2] - v1->win[2], da20 = v2->win[2] - v0->win[2];
  const float dzdx = da01 * dy20_ooa - da20 * dy01_ooa;
  const float dzdy = da20 * dx01_ooa - da01 * dx20_ooa;
  const float a0 = v0->win[2] - (dzdx * x0c + dzdy * y0c);

  for (int py = by0; py <= by1; py++) {
    for (int px = bx0; px <= bx1; px++) {
      const int64_t X = (int64_t)px << 8, Y = (int64_t)py << 8;
      int inside = 1;
      for (int i = 0; i < 3; i++) {
        /* c - dcdx*X + dcdy*Y > 0 */
        const int64_t v = ec[i] - ea[i] * X + eb[i] * Y;
        if (v <= 0) { inside = 0; break; }
      }
      if (!inside) continue;
      float z;
      if (g_var.interp_fma) {
        z = fmaf(dzdx, (float)px, a0);
        z = fmaf(dzdy, (float)py, z);
      } else {
        z = a0 + dzdx * (float)px;
        z = z + dzdy * (float)py;
      }
      shade(f, px, py, z);
    }
  }
}

static void clip_and_setup(frame *f, const vtx *t0, const vtx *t1, const vtx *t2,
                           const float *scale, const float *trans)
{
  f->n_tris_in++;
  const unsigned ormask = t0->mask | t1->mask | t2->mask;
  if (ormask == 0) { setup_tri(f, t0, t1, t2); return; }
  if (t0->mask & t1->mask & t2->mask) return;

  vtx tmp[MAX_CLIPPED * 2];
  int ntmp = 0;
  const vtx *a[MAX_CLIPPED + 1], *b[MAX_CLIPPED + 1];
  const vtx **inlist = a, **outlist = b;
  int n = 3;
  unsigned clipmask = ormask;
  inlist[0] = t0; inlist[1] = t1; inlist[2] = t2;

  while (clipmask && n >= 3) {
    int plane = 0;
    while (!(clipmask & (1u << plane))) plane++;
    clipmask &= ~(1u << plane);
    const vtx *vert_prev = inlist[0];
    float dp_prev = clipdist(vert_prev, plane);
    int outcount = 0;
    if (n >= MAX_CLIPPED) return;
    inlist[n] = inlist[0];
    for (int i = 1; i <= n; i++) {
      const vtx *vert = inlist[i];
      const float dp = clipdist(vert, plane);
      int different_sign;
      if (dp_prev >= 0.0f) {
        if (outcount >= MAX_CLIPPED) return;
        outlist[outcount++] = vert_prev;
        different_sign = dp < 0.0f;
      } else {
        different_sign = !(dp < 0.0f);
      }
      if (different_sign) {
        if (ntmp >= MAX_CLIPPED * 2 || outcount >= MAX_CLIPPED) return;
        vtx *nv = &tmp[ntmp++];
        outlist[outcount++] = nv;
        /* Mesa >= 21 interpolates from whichever end point is closer to the
         * plane (draw_pipe_clip.c do_clip_tri; confirmed on the 23.2.1 binary). */
        const float denom = dp - dp_prev;
        int from_vert;
        if (dp < 0.0f) from_vert = g_var.clip_old_t ? 1 : (dp_prev > -dp);       /* going out  */
        else from_vert = g_var.clip_old_t ? 0 : !(dp > -dp_prev);               /* coming in  */
        if (from_vert) clip_interp(nv, dp / denom, vert, vert_prev, scale, trans);
        else clip_interp(nv, -dp_prev / denom, vert_prev, vert, scale, trans);
      }
      vert_prev = vert;
      dp_prev = dp;
    }
    const vtx **sw = inlist; inlist = outlist; outlist = sw;
    n = outcount;
  }
  if (n < 3) return;
  for (int i = 2; i < n; i++) setup_tri(f, inlist[i - 1], inlist[i], inlist[0]);
}

/* Scratch memory of one frame (depth buffer, transformed vertices).  Normally malloc/free per call; the worker
 * threads of rtuf_oracle_filter_throughput switch their thread to two grow-only buffers instead: hundreds of
 * threads each mapping and unmapping megabytes per frame serialise on the process's address-space lock. */
static __thread int tl_arena_on;
static __thread void *tl_buf[2];
static __thread size_t tl_cap[2];
static void *scratch_get(int slot, size_t bytes)
{
  if (!tl_arena_on) return malloc(bytes);
  if (tl_cap[slot] < bytes) {
    free(tl_buf[slot]);
    tl_buf[slot] = malloc(bytes);
    tl_cap[slot] = tl_buf[slot] ? bytes : 0;
  }
  return tl_buf[slot];
}
static void scratch_put(void *p) { if (!tl_arena_on) free(p); }

/* ------------------------------------------------------------------ */
/* Public entry                                                       */
/* ------------------------------------------------------------------ */
int rtuf_oracle_filter(const rtuf_oracle_frame *in, float *masked_depth, uint8_t *mask,
                       rtuf_oracle_debug *dbg)
{
  const int w = in->width, h = in->height;
  if (w <= 0 || h <= 0 || !in->depth || !masked_depth) return -1;
  frame f;
  memset(&f, 0, sizeof f);
  f.w = w; f.h = h; f.sensor = in->depth;
  f.z_near = in->z_near; f.z_far = in->z_far;
  f.max_diff = in->max_diff; f.replace_value = in->replace_value;
  f.masked = masked_depth; f.mask = mask;
  f.z24 = (uint32_t *)scratch_get(0, sizeof(uint32_t) * (size_t)w * h);
  if (!f.z24) return -2;
  for (size_t i = 0; i < (size_t)w * h; i++) f.z24[i] = 0xffffffu;   /* glClear depth = 1.0 */
  /* glClear colour: (0,0,0,1); every pixel is overwritten by the background
   * quad, kept for exactness if far-plane clipping removed it. */
  for (size_t i = 0; i < (size_t)w * h; i++) masked_depth[i] = 0.0f;
  if (mask) memset(mask, 0, (size_t)w * h);
  if (dbg) {
    f.dbg_zwin = dbg->zwin; f.dbg_prim = dbg->prim;
    if (f.dbg_prim) for (size_t i = 0; i < (size_t)w * h; i++) f.dbg_prim[i] = -1;
  }

  const float scale[3] = { 0.5f * (float)w, 0.5f * (float)h, 0.5f };
  const float trans[3] = { 0.5f * (float)w, 0.5f * (float)h, 0.5f };

  float proj[16], mv[16], mvp[16];
  mat_identity(proj);
  mat_mult_d(proj, in->projection);                     /* urdf_filter.cpp:576-580 */
  mat_identity(mv);
  {
    static const float la[16] = { -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1 };
    mat_mult_f(mv, la);                                 /* gluLookAt, :587 */
    mat_translate(mv, -0.0f, -0.0f, -0.0f);
  }

  /* background quad, :591-596 (GL_QUADS -> (0,1,3),(1,2,3)) */
  {
    const float zq = (float)(in->z_far * 0.99);
    const float q[4][3] = { { -100.0f, -100.0f, zq }, { 100.0f, -100.0f, zq },
                            { 100.0f, 100.0f, zq }, { -100.0f, 100.0f, zq } };
    vtx v[4];
    matmul4(mvp, proj, mv);
    for (int i = 0; i < 4; i++) {
      vs_position(mvp, q[i], v[i].clip);
      v[i].mask = clipmask_of(v[i].clip);
      viewport_vs(scale, trans, &v[i]);
    }
    f.prim_id = -2;
    clip_and_setup(&f, &v[0], &v[1], &v[3], scale, trans);
    clip_and_setup(&f, &v[1], &v[2], &v[3], scale, trans);
  }

  mat_mult_d(mv, in->camera_offset_inv);                /* :602-604 */
  mat_mult_d(mv, in->camera_tf);                        /* :613-614 */

  int prim_base = 0;
  for (int d = 0; d < in->ndraws; d++) {
    const rtuf_oracle_draw *dr = &in->draws[d];
    float m[16];
    memcpy(m, mv, sizeof m);                            /* glPushMatrix */
    mat_mult_d(m, dr->link_tf);                         /* renderable.cpp:59-68 */
    if (dr->pre_op == RTUF_ORACLE_OP_SCALE) mat_scale(m, dr->op[0], dr->op[1], dr->op[2]);
    else if (dr->pre_op == RTUF_ORACLE_OP_TRANSLATE) mat_translate(m, dr->op[0], dr->op[1], dr->op[2]);
    matmul4(mvp, proj, m);
    vtx *tv = (vtx *)scratch_get(1, sizeof(vtx) * (size_t)(dr->nverts > 0 ? dr->nverts : 1));
    if (!tv) { scratch_put(f.z24); return -2; }
    for (int i = 0; i < dr->nverts; i++) {
      vs_position(mvp, dr->verts + 3 * (size_t)i, tv[i].clip);
      tv[i].mask = clipmask_of(tv[i].clip);
      viewport_vs(scale, trans, &tv[i]);
    }
    for (int t = 0; t < dr->ntris; t++) {
      const uint32_t *ix = dr->tris + 3 * (size_t)t;
      f.prim_id = prim_base + t;
      clip_and_setup(&f, &tv[ix[0]], &tv[ix[1]], &tv[ix[2]], scale, trans);
    }
    prim_base += dr->ntris;
    scratch_put(tv);
  }
  if (dbg) { dbg->n_tris_in = f.n_tris_in; dbg->n_tris_setup = f.n_tris_setup; dbg->n_frags = f.n_frags; }
  scratch_put(f.z24);
  return 0;
}

/* Expose the float32 matrix-stack result for host-side tests:
 * out_mvp = P x (LA x offset_inv x cam_tf x link_tf [x op]) in GL float order. */
void rtuf_oracle_compose_mvp(const double *projection, const double *camera_offset_inv,
                             const double *camera_tf, const double *link_tf,
                             int pre_op, const float *op, float *out_mvp)
{
  float proj[16], mv[16];
  static const float la[16] = { -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1 };
  mat_identity(proj); mat_mult_d(proj, projection);
  mat_identity(mv); mat_mult_f(mv, la); mat_translate(mv, -0.0f, -0.0f, -0.0f);
  mat_mult_d(mv, camera_offset_inv);
  mat_mult_d(mv, camera_tf);
  if (link_tf) mat_mult_d(mv, link_tf);
  if (pre_op == RTUF_ORACLE_OP_SCALE) mat_scale(mv, op[0], op[1], op[2]);
  else if (pre_op == RTUF_ORACLE_OP_TRANSLATE) mat_translate(mv, op[0], op[1], op[2]);
  matmul4(out_mvp, proj, mv);
}

/* ------------------------------------------------------------------ */
/* all-cores throughput (bench.py cpu_baseline.all_cores)             */
/* ------------------------------------------------------------------ */
#include <pthread.h>
#include <stdatomic.h>
#include <time.h>

static double tp_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
  const rtuf_oracle_frame *frames;
  int n_frames;
  double deadline;
  atomic_long *next;
  atomic_long *done;
  atomic_int *failed;
} tp_job;

static void *tp_worker(void *arg)
{
  tp_job *j = (tp_job *)arg;
  float *masked = NULL;
  uint8_t *mask = NULL;
  size_t cap = 0;
  tl_arena_on = 1;
  while (tp_now() < j->deadline) {
    const long i = atomic_fetch_add(j->next, 1);
    const rtuf_oracle_frame *fr = &j->frames[i % j->n_frames];
    const size_t px = (size_t)fr->width * (size_t)fr->height;
    if (px > cap) {
      free(masked); free(mask);
      masked = (float *)malloc(px * sizeof(float));
      mask = (uint8_t *)malloc(px);
      cap = px;
      if (!masked || !mask) { atomic_store(j->failed, 1); break; }
    }
    if (rtuf_oracle_filter(fr, masked, mask, NULL) != 0) { atomic_store(j->failed, 1); break; }
    atomic_fetch_add(j->done, 1);
  }
  free(masked); free(mask);
  free(tl_buf[0]); free(tl_buf[1]);
  tl_buf[0] = tl_buf[1] = NULL; tl_cap[0] = tl_cap[1] = 0;
  tl_arena_on = 0;
  return NULL;
}

long rtuf_oracle_filter_throughput(const rtuf_oracle_frame *frames, int n_frames, double seconds, int n_threads, double *elapsed_out)
{
  if (!frames || n_frames <= 0 || !(seconds > 0) || n_threads <= 0) return -1;
  atomic_long next = 0, done = 0;
  atomic_int failed = 0;
  const double t0 = tp_now();
  tp_job job = { frames, n_frames, t0 + seconds, &next, &done, &failed };
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  if (!th) return -2;
  int started = 0;
  for (int t = 0; t < n_threads; t++) {
    if (pthread_create(&th[t], NULL, tp_worker, &job) != 0) break;
    started++;
  }
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  free(th);
  if (elapsed_out) *elapsed_out = tp_now() - t0;
  if (started == 0 || atomic_load(&failed)) return -3;
  return atomic_load(&done);
}
