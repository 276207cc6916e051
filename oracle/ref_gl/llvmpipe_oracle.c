/*
 * llvmpipe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Headless replay of the reference's OpenGL frame on Mesa llvmpipe, used to
 * pin the CPU restatement (oracle/rtuf_oracle.c) and to generate the golden
 * fixtures under tests/golden/.  It runs only in the development container:
 * the two GLSL shader files are read AT RUN TIME from the reference checkout
 * (default /root/reference/include/shaders/urdf_filter.{vert,frag}); nothing
 * of the reference is compiled in or copied.
 *
 * What is restated here (reference file:line):
 *   - depth upload as GL_TEXTURE_BUFFER / GL_R32F      src/urdf_filter.cpp:332-353
 *   - FBO "rgba=4x32t depth=24t stencil=8t": 4 x RGBA32F rectangle colour
 *     textures + DEPTH_COMPONENT24 rectangle texture    src/urdf_filter.cpp:442-443,
 *                                                        src/FrameBufferObject.cpp:246-729
 *   - viewport (0,0,w,h) on capture                     src/FrameBufferObject.cpp:772-785
 *   - per-frame GL state / matrix / draw order          src/urdf_filter.cpp:542-644
 *   - per-renderable push / glMultMatrixd / pop         src/renderable.cpp:59-73
 *   - read-back of attachment 1 (GL_RED, GL_FLOAT) and
 *     attachment 3 (GL_RED, GL_UNSIGNED_BYTE)           src/urdf_filter.cpp:729-735
 *
 * The GL context comes from the DRI software-rasteriser loader interface of
 * swrast_dri.so (no X11 / GLX / EGL needed); GL entry points are resolved via
 * _glapi_get_proc_address from libglapi.
 *
 * gluLookAt(0,0,0, 0,0,1, 0,1,0) (src/urdf_filter.cpp:587) is replaced by the
 * matrix it evaluates to, diag(-1,1,-1,1), passed through glMultMatrixf like
 * GLU does (GLU is not installed here).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define GL_GLEXT_PROTOTYPES 0
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

/* ------------------------------------------------------------------ */
/* GL entry points                                                    */
/* ------------------------------------------------------------------ */
#define GLFUNCS(X) \
  X(PFNGLGENBUFFERSPROC, glGenBuffers) \
  X(PFNGLBINDBUFFERPROC, glBindBuffer) \
  X(PFNGLBUFFERDATAPROC, glBufferData) \
  X(PFNGLDELETEBUFFERSPROC, glDeleteBuffers) \
  X(PFNGLTEXBUFFERPROC, glTexBuffer) \
  X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers) \
  X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer) \
  X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D) \
  X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus) \
  X(PFNGLDRAWBUFFERSPROC, glDrawBuffers) \
  X(PFNGLCREATESHADERPROC, glCreateShader) \
  X(PFNGLSHADERSOURCEPROC, glShaderSource) \
  X(PFNGLCOMPILESHADERPROC, glCompileShader) \
  X(PFNGLGETSHADERIVPROC, glGetShaderiv) \
  X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) \
  X(PFNGLCREATEPROGRAMPROC, glCreateProgram) \
  X(PFNGLATTACHSHADERPROC, glAttachShader) \
  X(PFNGLLINKPROGRAMPROC, glLinkProgram) \
  X(PFNGLGETPROGRAMIVPROC, glGetProgramiv) \
  X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) \
  X(PFNGLUSEPROGRAMPROC, glUseProgram) \
  X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation) \
  X(PFNGLUNIFORM1IPROC, glUniform1i) \
  X(PFNGLUNIFORM1FPROC, glUniform1f) \
  X(PFNGLACTIVETEXTUREPROC, glActiveTexture)

#define DECL(T, n) static T p_##n;
GLFUNCS(DECL)
#undef DECL

/* GL 1.x functions are resolved dynamically too (glvnd's libGL does not
 * dispatch without GLX), under p_ names. */
#define GL1FUNCS(X) \
  X(const GLubyte *, glGetString, (GLenum)) \
  X(GLenum, glGetError, (void)) \
  X(void, glGenTextures, (GLsizei, GLuint *)) \
  X(void, glBindTexture, (GLenum, GLuint)) \
  X(void, glTexImage2D, (GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *)) \
  X(void, glTexParameteri, (GLenum, GLenum, GLint)) \
  X(void, glGetTexImage, (GLenum, GLint, GLenum, GLenum, void *)) \
  X(void, glViewport, (GLint, GLint, GLsizei, GLsizei)) \
  X(void, glClearColor, (GLclampf, GLclampf, GLclampf, GLclampf)) \
  X(void, glClearStencil, (GLint)) \
  X(void, glClear, (GLbitfield)) \
  X(void, glEnable, (GLenum)) \
  X(void, glDisable, (GLenum)) \
  X(void, glMatrixMode, (GLenum)) \
  X(void, glLoadIdentity, (void)) \
  X(void, glMultMatrixd, (const GLdouble *)) \
  X(void, glMultMatrixf, (const GLfloat *)) \
  X(void, glTranslated, (GLdouble, GLdouble, GLdouble)) \
  X(void, glTranslatef, (GLfloat, GLfloat, GLfloat)) \
  X(void, glScalef, (GLfloat, GLfloat, GLfloat)) \
  X(void, glPushMatrix, (void)) \
  X(void, glPopMatrix, (void)) \
  X(void, glPushAttrib, (GLbitfield)) \
  X(void, glPopAttrib, (void)) \
  X(void, glBegin, (GLenum)) \
  X(void, glEnd, (void)) \
  X(void, glVertex3f, (GLfloat, GLfloat, GLfloat)) \
  X(void, glVertex3d, (GLdouble, GLdouble, GLdouble)) \
  X(void, glColor3f, (GLfloat, GLfloat, GLfloat)) \
  X(void, glStencilFunc, (GLenum, GLint, GLuint)) \
  X(void, glStencilOp, (GLenum, GLenum, GLenum)) \
  X(void, glEnableClientState, (GLenum)) \
  X(void, glDisableClientState, (GLenum)) \
  X(void, glVertexPointer, (GLint, GLenum, GLsizei, const void *)) \
  X(void, glNormalPointer, (GLenum, GLsizei, const void *)) \
  X(void, glDrawArrays, (GLenum, GLint, GLsizei)) \
  X(void, glDrawElements, (GLenum, GLsizei, GLenum, const void *)) \
  X(void, glFinish, (void)) \
  X(void, glDepthRange, (GLclampd, GLclampd)) \
  X(void, glGetFloatv, (GLenum, GLfloat *))

#define DECL1(R, n, A) static R (*p_##n) A;
GL1FUNCS(DECL1)
#undef DECL1

/* ------------------------------------------------------------------ */
/* DRI swrast bring-up                                                */
/* ------------------------------------------------------------------ */
static int g_w, g_h;

static void ld_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = g_w > 0 ? g_w : 16; *h = g_h > 0 ? g_h : 16; }
static void ld_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension g_loader = {
  .base = { __DRI_SWRAST_LOADER, 3 },
  .getDrawableInfo = ld_getDrawableInfo,
  .putImage = ld_putImage,
  .getImage = ld_getImage,
  .putImage2 = ld_putImage2,
  .getImage2 = ld_getImage2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_getproc)(const char *);

static GLuint g_fbo, g_color[4], g_depth, g_prog;
static GLuint g_depth_pbo = 0, g_depth_tbo = 0;
static char g_err[1024];

const char *rgo_last_error(void) { return g_err; }

static int fail(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }

static char *read_file(const char *path)
{
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  char *s = (char *)malloc((size_t)n + 1);
  if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); return NULL; }
  s[n] = 0; fclose(f); return s;
}

static GLuint compile(GLenum type, const char *src)
{
  GLuint s = p_glCreateShader(type);
  p_glShaderSource(s, 1, &src, NULL);
  p_glCompileShader(s);
  GLint ok = 0; p_glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
  if (!ok) {
    char log[800]; p_glGetShaderInfoLog(s, sizeof log, NULL, log);
    snprintf(g_err, sizeof g_err, "shader compile failed: %s", log);
    return 0;
  }
  return s;
}

static int link_program(const char *vs_src, const char *fs_src, GLuint *out)
{
  GLuint vs = compile(GL_VERTEX_SHADER, vs_src); if (!vs) return -1;
  GLuint fs = compile(GL_FRAGMENT_SHADER, fs_src); if (!fs) return -1;
  GLuint p = p_glCreateProgram();
  p_glAttachShader(p, vs); p_glAttachShader(p, fs);
  p_glLinkProgram(p);
  GLint ok = 0; p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
  if (!ok) {
    char log[800]; p_glGetProgramInfoLog(p, sizeof log, NULL, log);
    snprintf(g_err, sizeof g_err, "program link failed: %s", log);
    return -1;
  }
  *out = p;
  return 0;
}

/* Create context + FBO + shader program.  vert_path / frag_path: files to
 * load the GLSL from (the reference's, read at run time). */
int rgo_create(int w, int h, const char *vert_path, const char *frag_path)
{
  g_w = w; g_h = h;
  if (!g_ctx) {
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return fail("dlopen libglapi.so.0 failed");
    g_getproc = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!g_getproc) return fail("no _glapi_get_proc_address");
    const char *drv = getenv("RGO_SWRAST_DRI");
    if (!drv) drv = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *dri = dlopen(drv, RTLD_NOW | RTLD_GLOBAL);
    if (!dri) return fail("dlopen swrast_dri.so failed");
    const __DRIextension **(*getexts)(void) =
        (const __DRIextension **(*)(void))dlsym(dri, "__driDriverGetExtensions_swrast");
    if (!getexts) return fail("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = getexts();
    for (int i = 0; exts[i]; i++) {
      if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
      if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast) return fail("DRI core/swrast extension missing");
    const __DRIconfig **configs = NULL;
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_screen) return fail("createNewScreen2 failed");
    const __DRIconfig *cfg = NULL;
    for (int i = 0; configs[i]; i++) {
      unsigned r = 0, d = 0, s = 0, db = 1, a = 0;
      g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
      g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
      g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &d);
      g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_STENCIL_SIZE, &s);
      g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
      if (r == 8 && a == 8 && d == 24 && s == 8 && !db) { cfg = configs[i]; break; }
    }
    if (!cfg) cfg = configs[0];
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 1 };
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL, cfg, NULL, 2, attribs, &err, NULL);
    if (!g_ctx) return fail("createContextAttribs failed");
    g_draw = g_swrast->createNewDrawable(g_screen, cfg, NULL);
    if (!g_draw) return fail("createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return fail("bindContext failed");

#define LOAD(T, n) p_##n = (T)g_getproc(#n); if (!p_##n) return fail("missing GL function " #n);
    GLFUNCS(LOAD)
#undef LOAD
#define LOAD1(R, n, A) p_##n = (R (*) A)g_getproc(#n); if (!p_##n) return fail("missing GL function " #n);
    GL1FUNCS(LOAD1)
#undef LOAD1
  }

  /* FBO: 4 x RGBA32F rectangle textures + depth24 rectangle texture
   * (src/urdf_filter.cpp:442-443 mode string). */
  p_glGenFramebuffers(1, &g_fbo);
  p_glBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
  p_glGenTextures(4, g_color);
  for (int i = 0; i < 4; i++) {
    p_glBindTexture(GL_TEXTURE_RECTANGLE, g_color[i]);
    p_glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    p_glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    p_glTexImage2D(GL_TEXTURE_RECTANGLE, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, NULL);
    p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_RECTANGLE, g_color[i], 0);
  }
  p_glGenTextures(1, &g_depth);
  p_glBindTexture(GL_TEXTURE_RECTANGLE, g_depth);
  p_glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  p_glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  p_glTexImage2D(GL_TEXTURE_RECTANGLE, 0, GL_DEPTH_COMPONENT24, w, h, 0, GL_DEPTH_COMPONENT, GL_FLOAT, NULL);
  p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_TEXTURE_RECTANGLE, g_depth, 0);
  if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("FBO incomplete");
  p_glBindFramebuffer(GL_FRAMEBUFFER, 0);

  char *vs = read_file(vert_path), *fs = read_file(frag_path);
  if (!vs || !fs) return fail("cannot read shader files");
  int rc = link_program(vs, fs, &g_prog);
  free(vs); free(fs);
  if (rc) return rc;
  if (p_glGetError() != GL_NO_ERROR) return fail("GL error during init");
  return 0;
}

/* Replace the fragment (and optionally vertex) program by source strings: a
 * development probe (e.g. dump gl_FragCoord.z) -- never used for fixtures. */
int rgo_set_program_source(const char *vs_src, const char *fs_src)
{
  return link_program(vs_src, fs_src, &g_prog);
}

const char *rgo_renderer_string(void) { return (const char *)p_glGetString(GL_RENDERER); }
const char *rgo_version_string(void) { return (const char *)p_glGetString(GL_VERSION); }

/* ------------------------------------------------------------------ */
/* Static geometry (VBO / IBO), like renderable.cpp:167-169, :343-349 */
/* ------------------------------------------------------------------ */
typedef struct { GLuint vbo, ibo; int nverts, nidx, stride; } rgo_mesh;
static rgo_mesh *g_meshes; static int g_nmeshes, g_capmeshes;

int rgo_mesh_create(const float *verts, int nverts, int stride_floats, const unsigned *idx, int nidx)
{
  if (g_nmeshes == g_capmeshes) {
    g_capmeshes = g_capmeshes ? 2 * g_capmeshes : 64;
    g_meshes = (rgo_mesh *)realloc(g_meshes, sizeof(rgo_mesh) * (size_t)g_capmeshes);
  }
  rgo_mesh *m = &g_meshes[g_nmeshes];
  memset(m, 0, sizeof *m);
  m->nverts = nverts; m->nidx = nidx; m->stride = stride_floats;
  p_glGenBuffers(1, &m->vbo);
  p_glBindBuffer(GL_ARRAY_BUFFER, m->vbo);
  p_glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)sizeof(float) * stride_floats * nverts, verts, GL_STATIC_DRAW);
  p_glBindBuffer(GL_ARRAY_BUFFER, 0);
  if (idx && nidx) {
    p_glGenBuffers(1, &m->ibo);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, m->ibo);
    p_glBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)sizeof(unsigned) * nidx, idx, GL_STATIC_DRAW);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, 0);
  }
  return g_nmeshes++;
}

void rgo_mesh_clear(void)
{
  for (int i = 0; i < g_nmeshes; i++) {
    p_glDeleteBuffers(1, &g_meshes[i].vbo);
    if (g_meshes[i].ibo) p_glDeleteBuffers(1, &g_meshes[i].ibo);
  }
  g_nmeshes = 0;
}

static void set_uniforms(float z_near, float z_far, float max_diff, float replace_value)
{
  p_glActiveTexture(GL_TEXTURE0);
  p_glUniform1i(p_glGetUniformLocation(g_prog, "depth_texture"), 0);
  p_glUniform1i(p_glGetUniformLocation(g_prog, "width"), g_w);
  p_glUniform1i(p_glGetUniformLocation(g_prog, "height"), g_h);
  p_glUniform1f(p_glGetUniformLocation(g_prog, "z_far"), z_far);
  p_glUniform1f(p_glGetUniformLocation(g_prog, "z_near"), z_near);
  p_glUniform1f(p_glGetUniformLocation(g_prog, "max_diff"), max_diff);
  p_glUniform1f(p_glGetUniformLocation(g_prog, "replace_value"), replace_value);
}

/* ------------------------------------------------------------------ */
/* Frame replay (src/urdf_filter.cpp:207-267, :503-744)                */
/* ------------------------------------------------------------------ */
void rgo_begin_frame(const float *depth, const double *P, const double *cam_offset_inv,
                     const double *cam_tf, float z_near, float z_far, float max_diff, float replace_value)
{
  static const GLenum bufs[4] = { GL_COLOR_ATTACHMENT0, GL_COLOR_ATTACHMENT1,
                                  GL_COLOR_ATTACHMENT2, GL_COLOR_ATTACHMENT3 };
  /* textureBufferFromDepthBuffer, src/urdf_filter.cpp:332-353 */
  if (!g_depth_pbo) p_glGenBuffers(1, &g_depth_pbo);
  p_glBindBuffer(GL_ARRAY_BUFFER, g_depth_pbo);
  p_glBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)g_w * g_h * 4, depth, GL_DYNAMIC_DRAW);
  p_glBindBuffer(GL_ARRAY_BUFFER, 0);
  if (!g_depth_tbo) p_glGenTextures(1, &g_depth_tbo);
  p_glBindTexture(GL_TEXTURE_BUFFER, g_depth_tbo);
  p_glTexBuffer(GL_TEXTURE_BUFFER, GL_R32F, g_depth_pbo);

  /* render(), src/urdf_filter.cpp:542-632 */
  p_glPushAttrib(GL_ALL_ATTRIB_BITS);
  p_glEnable(GL_NORMALIZE);
  p_glBindFramebuffer(GL_FRAMEBUFFER, g_fbo);          /* beginCapture */
  p_glViewport(0, 0, g_w, g_h);
  p_glUseProgram(g_prog);
  p_glDrawBuffers(4, bufs);
  p_glClearColor(0.0f, 0.0f, 0.0f, 1.0f);
  p_glClearStencil(0);
  p_glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT | GL_STENCIL_BUFFER_BIT);
  p_glEnable(GL_DEPTH_TEST);
  p_glDisable(GL_TEXTURE_2D);

  /* The reference's shader object is a function-local static (src/urdf_filter.cpp:549):
   * uniform values persist from the previous frame, so from frame 2 on the
   * background quad below is shaded with the (unchanged) uniforms that
   * :623-632 set at the end of frame N-1.  Frame 1 of the reference shades the
   * background with all-zero uniforms (a start-up artefact); this harness
   * replays the steady state, i.e. the uniforms are already in place here. */
  set_uniforms(z_near, z_far, max_diff, replace_value);

  p_glMatrixMode(GL_PROJECTION);
  p_glLoadIdentity();
  p_glMultMatrixd(P);
  p_glMatrixMode(GL_MODELVIEW);
  p_glLoadIdentity();
  {
    /* gluLookAt(0,0,0, 0,0,1, 0,1,0): GLU builds this float matrix, calls
     * glMultMatrixf, then glTranslated(-eye). */
    static const GLfloat la[16] = { -1, 0, 0, 0,  0, 1, 0, 0,  0, 0, -1, 0,  0, 0, 0, 1 };
    p_glMultMatrixf(la);
    p_glTranslated(-0.0, -0.0, -0.0);
  }
  p_glBegin(GL_QUADS);                                  /* background quad, :591-596 */
  p_glVertex3f(-100.0f, -100.0f, (float)(z_far * 0.99));
  p_glVertex3f(100.0f, -100.0f, (float)(z_far * 0.99));
  p_glVertex3f(100.0f, 100.0f, (float)(z_far * 0.99));
  p_glVertex3f(-100.0f, 100.0f, (float)(z_far * 0.99));
  p_glEnd();

  p_glMultMatrixd(cam_offset_inv);                      /* :602-604 */
  p_glMultMatrixd(cam_tf);                              /* :613-614 */

  p_glEnable(GL_STENCIL_TEST);                          /* :618-620 */
  p_glStencilFunc(GL_ALWAYS, 0x1, 0x1);
  p_glStencilOp(GL_KEEP, GL_KEEP, GL_REPLACE);

  set_uniforms(z_near, z_far, max_diff, replace_value);  /* :623-632 */
  p_glBindTexture(GL_TEXTURE_BUFFER, g_depth_tbo);
}

/* Renderable::applyTransform, src/renderable.cpp:59-68 */
void rgo_push_link(const double *link_tf) { p_glPushMatrix(); p_glMultMatrixd(link_tf); }
void rgo_pop_link(void) { p_glPopMatrix(); }
void rgo_scale(float x, float y, float z) { p_glScalef(x, y, z); }
void rgo_translate(float x, float y, float z) { p_glTranslatef(x, y, z); }

/* mode: GL primitive enum (GL_TRIANGLES=4, GL_TRIANGLE_STRIP=5, GL_TRIANGLE_FAN=6,
 * GL_QUADS=7, GL_QUAD_STRIP=8).  Indexed when the mesh has an IBO. */
void rgo_mesh_draw(int id, int mode)
{
  rgo_mesh *m = &g_meshes[id];
  p_glEnableClientState(GL_VERTEX_ARRAY);
  p_glBindBuffer(GL_ARRAY_BUFFER, m->vbo);
  p_glVertexPointer(3, GL_FLOAT, (GLsizei)(m->stride * sizeof(float)), (const void *)0);
  if (m->stride >= 6) {
    p_glEnableClientState(GL_NORMAL_ARRAY);
    p_glNormalPointer(GL_FLOAT, (GLsizei)(m->stride * sizeof(float)), (const void *)(3 * sizeof(float)));
  }
  if (m->ibo) {
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, m->ibo);
    p_glDrawElements((GLenum)mode, m->nidx, GL_UNSIGNED_INT, (const void *)0);
    p_glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, 0);
  } else {
    p_glDrawArrays((GLenum)mode, 0, m->nverts);
  }
  p_glDisableClientState(GL_VERTEX_ARRAY);
  if (m->stride >= 6) p_glDisableClientState(GL_NORMAL_ARRAY);
  p_glBindBuffer(GL_ARRAY_BUFFER, 0);
}

/* Immediate-mode draw with double vertices (freeglut 2.8 glutSolid* style). */
void rgo_draw_immediate_d(int mode, const double *xyz, int nverts)
{
  p_glBegin((GLenum)mode);
  for (int i = 0; i < nverts; i++) p_glVertex3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  p_glEnd();
}

void rgo_get_matrix(int which, float *m16)
{
  p_glGetFloatv(which == 0 ? GL_MODELVIEW_MATRIX : GL_PROJECTION_MATRIX, m16);
}

/* src/urdf_filter.cpp:641-644, :729-735 */
void rgo_end_frame(float *masked_depth, unsigned char *mask)
{
  p_glUseProgram(0);
  p_glBindFramebuffer(GL_FRAMEBUFFER, 0);              /* endCapture */
  p_glPopAttrib();
  p_glBindTexture(GL_TEXTURE_RECTANGLE, g_color[1]);
  p_glGetTexImage(GL_TEXTURE_RECTANGLE, 0, GL_RED, GL_FLOAT, masked_depth);
  if (mask) {
    /* The reference reads the mask with the default GL_PACK_ALIGNMENT of 4 into a width*height buffer
     * (src/urdf_filter.cpp:733-735): for widths that are not a multiple of 4 GL pads every row and the
     * reference overruns mask_.  The harness reads into a row-padded scratch buffer and hands back the
     * dense width*height image the texture holds. */
    const int stride = (g_w + 3) & ~3;
    p_glBindTexture(GL_TEXTURE_RECTANGLE, g_color[3]);
    if (stride == g_w) {
      p_glGetTexImage(GL_TEXTURE_RECTANGLE, 0, GL_RED, GL_UNSIGNED_BYTE, mask);
    } else {
      unsigned char *tmp = (unsigned char *)malloc((size_t)stride * g_h);
      p_glGetTexImage(GL_TEXTURE_RECTANGLE, 0, GL_RED, GL_UNSIGNED_BYTE, tmp);
      for (int y = 0; y < g_h; y++) memcpy(mask + (size_t)y * g_w, tmp + (size_t)y * stride, (size_t)g_w);
      free(tmp);
    }
  }
}

/* Debug: full RGBA32F read-back of one colour attachment (probe shaders). */
void rgo_read_attachment(int i, float *rgba)
{
  p_glBindTexture(GL_TEXTURE_RECTANGLE, g_color[i]);
  p_glGetTexImage(GL_TEXTURE_RECTANGLE, 0, GL_RGBA, GL_FLOAT, rgba);
}

/* Debug: read the 24-bit depth attachment as float / uint. */
void rgo_read_depth_u32(unsigned *z)
{
  p_glBindTexture(GL_TEXTURE_RECTANGLE, g_depth);
  p_glGetTexImage(GL_TEXTURE_RECTANGLE, 0, GL_DEPTH_COMPONENT, GL_UNSIGNED_INT, z);
}

/* Debug probe only: non-default depth range (the reference never changes it). */
void rgo_debug_depth_range(double n, double f) { p_glDepthRange(n, f); }

int rgo_gl_error(void) { return (int)p_glGetError(); }

double rgo_now(void)
{
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
