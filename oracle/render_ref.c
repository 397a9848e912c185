/*
 * render_ref.c -- ORACLE (test infrastructure, not product code).
 *
 * Scalar CPU restatement of the reference's G-buffer render path:
 *   RenderPass::render            /root/reference/src/render_pass.cpp:303-796
 *   vertex shader                 src/shaders/render_shader.vert:57-95
 *   geometry shader               src/shaders/render_shader.geom:13-35
 *   fragment shader               src/shaders/render_shader.frag:225-412
 *   uniform plumbing              src/shaders/render_shader.cpp:233-461
 *   shadow pass                   src/render_pass.cpp:408-460, shadow_shader.vert
 *   SSAO                          src/shaders/ssao_shader.frag:20-56, ssao_shader.cpp:72-112
 *   SSAO blur/apply               src/shaders/ssao_apply_shader.frag:29-75
 *   tone map                      src/shaders/tone_map_shader.frag:102-131
 * plus the fixed-function rasteriser those shaders run under (OpenGL 4.5 core, state set at
 * render_pass.cpp:325-332: depth test LESS, no culling, FrontFace = CW, D24 depth buffer
 * render_pass.cpp:373).
 *
 * PARITY STATUS: "parity unpinned" for pixel values.  The reference cannot be built or run in
 * the build container (Magnum/Corrade submodules are empty, no EGL) and its tests pin only
 * structural properties (tests/basic.cpp:108-261, :375-453); those known-answer properties
 * are checked against this file in tests/test_oracle_render.py.  The rasterisation rules GL
 * leaves implementation-defined (sub-pixel snapping, interpolation order, tie-breaks) are
 * FIXED here and are the contract the HIP path must reproduce bit-for-bit:
 *
 *   R1  every 4x4 * vec4 (and 4x4 * 4x4) product is the k-ordered chain
 *          fma(m3,v3, fma(m2,v2, fma(m1,v1, fma(m0,v0, 0))))
 *       (bitwise what v_mfma_f32_16x16x4_f32 computes), compiled with -ffp-contract=off.
 *       Clip positions use the composite matrix P (W2C (O2W M2O)) (one product per vertex);
 *       the varyings object/world/camera coordinates follow the shader's sequential chain.
 *   R2  x_win = fma(x_ndc, W/2, W/2), y likewise, z_win = fma(z_ndc, .5, .5); ndc = clip/w
 *       with a correctly rounded division.
 *   R3  window x,y snapped to 1/256 px:  X = (int)floorf(fmaf(x_win, 256, 0.5)).
 *   R4  coverage by exact 64-bit integer edge functions sampled at pixel centres
 *       (256*i+128), ownership of shared edges: edge (dx,dy) of the CCW-normalised triangle
 *       is owned iff dy < 0 || (dy == 0 && dx < 0).
 *   R5  lambda_i = (float)E_i / (float)area2;  z = fma(l2, z2-z0, fma(l1, z1-z0, z0));
 *       perspective: p_i = lambda_i * (1/w_i), s = (p0+p1)+p2, beta_i = p_i / s;
 *       varying a = fma(b2, a2-a0, fma(b1, a1-a0, a0)).
 *   R6  depth: d24 = min((uint)floorf(fmaf(z, 16777215, 0.5)), 0xFFFFFF), test LESS, draw
 *       order = draw index then triangle index (earlier wins ties); fragments with z outside
 *       [0,1] are clipped.
 *   R7  triangles crossing the near plane (z_clip < -w_clip) are clipped in clip space
 *       (Sutherland-Hodgman, t = d_in / (d_in - d_out)); sub-triangles carry the barycentric
 *       coordinates of the original triangle.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/slhip.h"
#include "cubemap_ref.h"

#define REF_PI 3.141592653589793f
#define INVALID_VALUE 3000.0f /* render_pass.cpp:316 */

/* ------------------------------------------------------------------------------------------ */
/* fixed-order arithmetic                                                                      */
/* ------------------------------------------------------------------------------------------ */
static inline float dot3(const float* a, const float* b)
{
    float d = a[0] * b[0];
    d = fmaf(a[1], b[1], d);
    d = fmaf(a[2], b[2], d);
    return d;
}

static inline void mv4(const float* M, const float* v, float* o)
{
    for (int r = 0; r < 4; ++r) {
        float a = fmaf(M[4 * r + 0], v[0], 0.0f);
        a = fmaf(M[4 * r + 1], v[1], a);
        a = fmaf(M[4 * r + 2], v[2], a);
        a = fmaf(M[4 * r + 3], v[3], a);
        o[r] = a;
    }
}

/* 3x3 (rows padded to 4) times vec3 */
static inline void mv3p(const float* M, const float* v, float* o)
{
    for (int r = 0; r < 3; ++r) {
        float a = fmaf(M[4 * r + 0], v[0], 0.0f);
        a = fmaf(M[4 * r + 1], v[1], a);
        a = fmaf(M[4 * r + 2], v[2], a);
        o[r] = a;
    }
}

/* C = A*B, same chain per element */
static inline void mm4(const float* A, const float* B, float* C)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float a = fmaf(A[4 * r + 0], B[0 + c], 0.0f);
            a = fmaf(A[4 * r + 1], B[4 + c], a);
            a = fmaf(A[4 * r + 2], B[8 + c], a);
            a = fmaf(A[4 * r + 3], B[12 + c], a);
            C[4 * r + c] = a;
        }
}

static inline void normalize3(float* v)
{
    /* one reciprocal and three products: GLSL normalize() is x * inversesqrt(dot(x, x)) */
    const float s = 1.0f / sqrtf(dot3(v, v));
    v[0] = v[0] * s;
    v[1] = v[1] * s;
    v[2] = v[2] * s;
}

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ------------------------------------------------------------------------------------------ */
/* vertex stage (render_shader.vert:57-95)                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float objc[4];   /* objectCoordinates xyz, w = camera z (vert:72)  */
    float world[3];
    float cam[3];
    float nrm[3];    /* normalInWorld                                  */
    float uv[2];
    float clip[4];
} vs_out;

static void vertex_stage(const slhip_mesh_pool* pool, const slhip_scene* sc, const slhip_draw* dr,
                         uint32_t v, vs_out* o)
{
    const float* p = pool->d_pos + 4 * (size_t)v;
    float pos[4] = {p[0], p[1], p[2], 1.0f};
    float obj4[4], world4[4], cam4[4];
    mv4(dr->mesh_to_object, pos, obj4);
    o->objc[0] = obj4[0] / obj4[3];
    o->objc[1] = obj4[1] / obj4[3];
    o->objc[2] = obj4[2] / obj4[3];
    mv4(dr->object_to_world, obj4, world4);
    for (int i = 0; i < 3; ++i) o->world[i] = world4[i] / world4[3];
    mv4(sc->world_to_cam, world4, cam4);
    for (int i = 0; i < 3; ++i) o->cam[i] = cam4[i] / cam4[3];
    o->objc[3] = o->cam[2];
    const float* n = pool->d_nrm + 4 * (size_t)v;
    mv3p(dr->normal_to_world, n, o->nrm);
    normalize3(o->nrm);
    /* clip position: ONE product with the composite matrix MVP = P (W2C (O2W M2O)), each
       matrix product being the same k-ordered chain -- the form the MFMA vertex kernel of the
       product evaluates (v_mfma_f32_16x16x4_f32: 16 vertices x {camera, 3 light} clip rows) */
    {
        float T1[16], T2[16], MVP[16];
        mm4(dr->object_to_world, dr->mesh_to_object, T1);
        mm4(sc->world_to_cam, T1, T2);
        mm4(sc->proj, T2, MVP);
        mv4(MVP, pos, o->clip);
    }
    o->uv[0] = pool->d_uv[2 * (size_t)v + 0];
    o->uv[1] = pool->d_uv[2 * (size_t)v + 1];
}

/* ------------------------------------------------------------------------------------------ */
/* triangle setup                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float clip[4];
    float bary[3]; /* barycentric coordinates w.r.t. the ORIGINAL triangle */
} clip_vert;

typedef struct {
    int64_t X[3], Y[3];
    float z[3], invw[3];
    float bary[3][3];
    int64_t area2; /* > 0 after normalisation */
    int flipped;   /* 1 if the orientation was negated (i.e. original area2 < 0) */
    int64_t bias[3];
    int xmin, xmax, ymin, ymax; /* pixel bbox clipped to the viewport, inclusive */
} tri_setup;

static inline int64_t snap(float w)
{
    float s = floorf(fmaf(w, 256.0f, 0.5f));
    /* keep coordinates inside int32 and edge products inside int64: |coord| <= 2^26 */
    if (!(s == s)) s = 0.0f;
    if (s > 67108864.0f) s = 67108864.0f;
    if (s < -67108864.0f) s = -67108864.0f;
    return (int64_t)s;
}

/* returns 0 if the (sub-)triangle is degenerate or misses the viewport */
static int setup_triangle(const clip_vert* v0, const clip_vert* v1, const clip_vert* v2, int W,
                          int H, tri_setup* t)
{
    const clip_vert* vs[3] = {v0, v1, v2};
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    for (int i = 0; i < 3; ++i) {
        const float* c = vs[i]->clip;
        float xn = c[0] / c[3], yn = c[1] / c[3], zn = c[2] / c[3];
        float xw = fmaf(xn, hw, hw), yw = fmaf(yn, hh, hh);
        t->X[i] = snap(xw);
        t->Y[i] = snap(yw);
        t->z[i] = fmaf(zn, 0.5f, 0.5f);
        t->invw[i] = 1.0f / c[3];
        memcpy(t->bary[i], vs[i]->bary, sizeof(float) * 3);
    }
    int64_t area2 = (t->X[1] - t->X[0]) * (t->Y[2] - t->Y[0]) - (t->Y[1] - t->Y[0]) * (t->X[2] - t->X[0]);
    if (area2 == 0) return 0;
    t->flipped = area2 < 0;
    t->area2 = area2 < 0 ? -area2 : area2;
    /* top-left style ownership per edge; edge i is opposite vertex i: (i+1) -> (i+2) */
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, b = (i + 2) % 3;
        int64_t dx = t->X[b] - t->X[a], dy = t->Y[b] - t->Y[a];
        if (t->flipped) { dx = -dx; dy = -dy; }
        int owned = (dy < 0) || (dy == 0 && dx < 0);
        t->bias[i] = owned ? 0 : -1;
    }
    int64_t xmn = t->X[0], xmx = t->X[0], ymn = t->Y[0], ymx = t->Y[0];
    for (int i = 1; i < 3; ++i) {
        if (t->X[i] < xmn) xmn = t->X[i];
        if (t->X[i] > xmx) xmx = t->X[i];
        if (t->Y[i] < ymn) ymn = t->Y[i];
        if (t->Y[i] > ymx) ymx = t->Y[i];
    }
    /* pixel i has its centre at 256 i + 128: first centre >= xmn, last centre <= xmx */
    int64_t x0 = (xmn - 128 + 255) >> 8, x1 = (xmx - 128) >> 8;
    int64_t y0 = (ymn - 128 + 255) >> 8, y1 = (ymx - 128) >> 8;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W - 1) x1 = W - 1;
    if (y1 > H - 1) y1 = H - 1;
    if (x0 > x1 || y0 > y1) return 0;
    t->xmin = (int)x0; t->xmax = (int)x1; t->ymin = (int)y0; t->ymax = (int)y1;
    return 1;
}

/* evaluates coverage at pixel (px,py); on success fills lambda (screen-space barycentrics) */
static inline int coverage(const tri_setup* t, int px, int py, float lambda[3])
{
    int64_t cx = 256 * (int64_t)px + 128, cy = 256 * (int64_t)py + 128;
    int64_t E[3];
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, b = (i + 2) % 3;
        int64_t e = (t->X[b] - t->X[a]) * (cy - t->Y[a]) - (t->Y[b] - t->Y[a]) * (cx - t->X[a]);
        if (t->flipped) e = -e;
        if (e + t->bias[i] < 0) return 0;
        E[i] = e;
    }
    float fa = (float)t->area2;
    lambda[0] = (float)E[0] / fa;
    lambda[1] = (float)E[1] / fa;
    lambda[2] = (float)E[2] / fa;
    return 1;
}

/* perspective-correct barycentrics (w.r.t. the original triangle) of pixel (px,py) on the plane of the
   set-up sub-triangle, also outside its edges: the texture footprint */
static inline void bary_at(const tri_setup* t, int px, int py, float b[3])
{
    int64_t cx = 256 * (int64_t)px + 128, cy = 256 * (int64_t)py + 128;
    float l[3];
    float fa = (float)t->area2;
    for (int i = 0; i < 3; ++i) {
        int a = (i + 1) % 3, c = (i + 2) % 3;
        int64_t e = (t->X[c] - t->X[a]) * (cy - t->Y[a]) - (t->Y[c] - t->Y[a]) * (cx - t->X[a]);
        if (t->flipped) e = -e;
        l[i] = (float)e / fa;
    }
    float pw0 = l[0] * t->invw[0], pw1 = l[1] * t->invw[1], pw2 = l[2] * t->invw[2];
    float sw = (pw0 + pw1) + pw2;
    float bs[3] = {pw0 * (1.0f / sw), pw1 * (1.0f / sw), pw2 * (1.0f / sw)};
    for (int k = 0; k < 3; ++k) b[k] = fmaf(bs[2], t->bary[2][k], fmaf(bs[1], t->bary[1][k], bs[0] * t->bary[0][k]));
}

static inline float interp(const float b[3], float a0, float a1, float a2)
{
    return fmaf(b[2], a2 - a0, fmaf(b[1], a1 - a0, a0));
}

/* near-plane clipping (R7).  in: 3 verts, out: up to 4 verts; returns the count (0, 3 or 4) */
static int clip_near(const clip_vert in[3], clip_vert out[4])
{
    float d[3];
    int inside[3], n_in = 0;
    for (int i = 0; i < 3; ++i) {
        d[i] = in[i].clip[2] + in[i].clip[3];
        inside[i] = in[i].clip[2] >= -in[i].clip[3];
        n_in += inside[i];
    }
    if (n_in == 0) return 0;
    if (n_in == 3) {
        memcpy(out, in, sizeof(clip_vert) * 3);
        return 3;
    }
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        if (inside[i]) out[n++] = in[i];
        if (inside[i] != inside[j]) {
            /* always interpolate from the inside vertex towards the outside one so that the
               two triangles sharing an edge produce the same point */
            const clip_vert* a = inside[i] ? &in[i] : &in[j];
            const clip_vert* b = inside[i] ? &in[j] : &in[i];
            float da = inside[i] ? d[i] : d[j], db = inside[i] ? d[j] : d[i];
            float tt = da / (da - db);
            clip_vert* o = &out[n++];
            for (int k = 0; k < 4; ++k) o->clip[k] = fmaf(tt, b->clip[k] - a->clip[k], a->clip[k]);
            for (int k = 0; k < 3; ++k) o->bary[k] = fmaf(tt, b->bary[k] - a->bary[k], a->bary[k]);
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* texture fetch: bilinear, mip 0, repeat wrap; texels are RGBA8 UNORM                          */
/* ------------------------------------------------------------------------------------------ */
static inline int wrapi(int i, int n)
{
    int m = i % n;
    return m < 0 ? m + n : m;
}

/* ---- 2D textures with mip chains and the asset's sampler state (include/slhip.h, SLHIP_SAMPLER_*) ---- */
static inline int wrap_coord(int i, int n, unsigned mode)
{
    if (mode == 1u) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    if (mode == 2u) {
        const int m = wrapi(i, 2 * n);
        return m < n ? m : 2 * n - 1 - m;
    }
    return wrapi(i, n);
}

static inline void texel_rgba(const uint8_t* lvl, int w, int x, int y, float out[4])
{
    const uint8_t* t = lvl + 4 * ((size_t)y * w + x);
    for (int c = 0; c < 4; ++c) out[c] = (float)t[c] / 255.0f;
}

static void tex_level(const uint8_t* lvl, int w, int h, unsigned sampler, int linear, float u, float v, float out[4])
{
    const unsigned ws = SLHIP_SAMPLER_WRAP_S(sampler), wt = SLHIP_SAMPLER_WRAP_T(sampler);
    if (!linear) {
        texel_rgba(lvl, w, wrap_coord((int)floorf(u * (float)w), w, ws), wrap_coord((int)floorf(v * (float)h), h, wt), out);
        return;
    }
    float x = fmaf(u, (float)w, -0.5f), y = fmaf(v, (float)h, -0.5f);
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = wrap_coord((int)fx, w, ws), y0 = wrap_coord((int)fy, h, wt);
    int x1 = wrap_coord((int)fx + 1, w, ws), y1 = wrap_coord((int)fy + 1, h, wt);
    float c00[4], c10[4], c01[4], c11[4];
    texel_rgba(lvl, w, x0, y0, c00); texel_rgba(lvl, w, x1, y0, c10);
    texel_rgba(lvl, w, x0, y1, c01); texel_rgba(lvl, w, x1, y1, c11);
    for (int c = 0; c < 4; ++c) {
        float top = fmaf(ax, c10[c] - c00[c], c00[c]), bot = fmaf(ax, c11[c] - c01[c], c01[c]);
        out[c] = fmaf(ay, bot - top, top);
    }
}

static const uint8_t* tex_level_ptr(const uint8_t* tex, int w0, int h0, int l, int* w, int* h)
{
    size_t off = 0;
    *w = w0; *h = h0;
    for (int k = 0; k < l; ++k) {
        off += 4 * (size_t)(*w) * (*h);
        *w = *w >> 1 > 1 ? *w >> 1 : 1;
        *h = *h >> 1 > 1 ? *h >> 1 : 1;
    }
    return tex + off;
}

/* log2 of a positive normal float: exponent + degree-6 polynomial in fmaf form (max error 1.4e-6), the same
   operations as the HIP path (bit-identical level of detail; libm's log2f differs from the device's) */
static inline float det_log2(float x)
{
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const int e = (int)(bits >> 23) - 127;
    uint32_t mb = (bits & 0x7fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mb, 4);
    const float t = m - 1.0f;
    float q = 0.02049034833908081f;
    q = fmaf(q, t, -0.09606625884771347f);
    q = fmaf(q, t, 0.2155885398387909f);
    q = fmaf(q, t, -0.33924779295921326f);
    q = fmaf(q, t, 0.4777059257030487f);
    q = fmaf(q, t, -0.721162736415863f);
    q = fmaf(q, t, 1.4426932334899902f);
    return fmaf(q, t, (float)e);
}

/* texture2D(): level of detail from the forward differences to the +x / +y pixel neighbours (GL 4.5, 8.14) */
static void tex_sample(const uint8_t* tex, int w, int h, unsigned sampler, float u, float v, float dudx, float dvdx,
                       float dudy, float dvdy, float out[4])
{
    const unsigned mip = SLHIP_SAMPLER_MIP(sampler);
    const float ax = dudx * (float)w, bx = dvdx * (float)h, ay = dudy * (float)w, by = dvdy * (float)h;
    const float rx = sqrtf(fmaf(bx, bx, ax * ax)), ry = sqrtf(fmaf(by, by, ay * ay));
    const float rho = fmaxf(rx, ry);
    const int magnify = !(rho > 1.0f);
    if (magnify || mip == 0u) {
        tex_level(tex, w, h, sampler, (sampler & (magnify ? SLHIP_SAMPLER_MAG_LINEAR : SLHIP_SAMPLER_MIN_LINEAR)) != 0u, u, v, out);
        return;
    }
    int top = 0;
    for (int m = w > h ? w : h; m > 1; m >>= 1) ++top;
    const float lambda = fminf(det_log2(rho), (float)top);
    const int lin = (sampler & SLHIP_SAMPLER_MIN_LINEAR) != 0u;
    int lw, lh;
    if (mip == 1u) {
        int l = (int)ceilf(lambda + 0.5f) - 1;
        if (l > top) l = top;
        if (l < 0) l = 0;
        const uint8_t* p = tex_level_ptr(tex, w, h, l, &lw, &lh);
        tex_level(p, lw, lh, sampler, lin, u, v, out);
        return;
    }
    int l0 = (int)floorf(lambda);
    if (l0 > top) l0 = top;
    const int l1 = l0 + 1 < top ? l0 + 1 : top;
    const float f = lambda - (float)l0;
    float a[4], b[4];
    const uint8_t* p0 = tex_level_ptr(tex, w, h, l0, &lw, &lh);
    tex_level(p0, lw, lh, sampler, lin, u, v, a);
    if (l1 == l0 || f == 0.0f) {
        for (int c = 0; c < 4; ++c) out[c] = a[c];
        return;
    }
    const uint8_t* p1 = tex_level_ptr(tex, w, h, l1, &lw, &lh);
    tex_level(p1, lw, lh, sampler, lin, u, v, b);
    for (int c = 0; c < 4; ++c) out[c] = fmaf(f, b[c] - a[c], a[c]);
}

/* rectangle texture (the sticker, frag:254): unnormalised texel coordinates, LINEAR, clamp to edge; the
   image is stored top row first while GL texel row 0 is the bottom row of an imported image */
static void tex_rect_bilinear(const uint8_t* tex, int w, int h, float xt, float yt, float out[4])
{
    const float x = xt - 0.5f, y = yt - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, x1 = (int)fx + 1, y0 = (int)fy, y1 = (int)fy + 1;
    if (x0 < 0) x0 = 0; if (x0 > w - 1) x0 = w - 1; if (x1 < 0) x1 = 0; if (x1 > w - 1) x1 = w - 1;
    if (y0 < 0) y0 = 0; if (y0 > h - 1) y0 = h - 1; if (y1 < 0) y1 = 0; if (y1 > h - 1) y1 = h - 1;
    y0 = (h - 1) - y0; y1 = (h - 1) - y1;
    for (int c = 0; c < 4; ++c) {
        const float a = (float)tex[4 * ((size_t)y0 * w + x0) + c] / 255.0f, b = (float)tex[4 * ((size_t)y0 * w + x1) + c] / 255.0f;
        const float cc = (float)tex[4 * ((size_t)y1 * w + x0) + c] / 255.0f, d = (float)tex[4 * ((size_t)y1 * w + x1) + c] / 255.0f;
        const float top = fmaf(ax, b - a, a), bot = fmaf(ax, d - cc, cc);
        out[c] = fmaf(ay, bot - top, top);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* PBR shading (render_shader.frag:181-221, 248-399)                                           */
/* ------------------------------------------------------------------------------------------ */
static float distribution_ggx(const float* N, const float* Hv, float roughness)
{
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH = fmaxf(dot3(N, Hv), 0.0f);
    float NdotH2 = NdotH * NdotH;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    denom = REF_PI * denom * denom;
    return a2 / denom;
}

static float geometry_schlick_ggx(float NdotV, float roughness)
{
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}

static float geometry_smith(const float* N, const float* V, const float* L, float roughness)
{
    float NdotV = fmaxf(dot3(N, V), 0.0f);
    float NdotL = fmaxf(dot3(N, L), 0.0f);
    return geometry_schlick_ggx(NdotL, roughness) * geometry_schlick_ggx(NdotV, roughness);
}

/* 16-tap PCF over a LINEAR-filtered LessOrEqual compare sampler (frag:321-337,
   render_pass.cpp:273-279).  shadow: f32 [S,S] window depth; background = 1.0 */
static float shadow_tap(const float* sm, int S, float u, float v, float ref)
{
    float x = fmaf(u, (float)S, -0.5f), y = fmaf(v, (float)S, -0.5f);
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 = 0; if (x0 > S - 1) x0 = S - 1;
    if (x1 < 0) x1 = 0; if (x1 > S - 1) x1 = S - 1;
    if (y0 < 0) y0 = 0; if (y0 > S - 1) y0 = S - 1;
    if (y1 < 0) y1 = 0; if (y1 > S - 1) y1 = S - 1;
    float r = clampf(ref, 0.0f, 1.0f);
    float c00 = r <= sm[(size_t)y0 * S + x0] ? 1.0f : 0.0f;
    float c10 = r <= sm[(size_t)y0 * S + x1] ? 1.0f : 0.0f;
    float c01 = r <= sm[(size_t)y1 * S + x0] ? 1.0f : 0.0f;
    float c11 = r <= sm[(size_t)y1 * S + x1] ? 1.0f : 0.0f;
    float top = fmaf(ax, c10 - c00, c00), bot = fmaf(ax, c11 - c01, c01);
    return fmaf(ay, bot - top, top);
}

typedef struct {
    const slhip_mesh_pool* pool;
    const slhip_scene* sc;
    int W, H;
    uint32_t flags;
    const float* shadow; /* [NUM_LIGHTS,S,S] or NULL */
    int S;
} shade_ctx;

static void shade_fragment(const shade_ctx* cx, const slhip_draw* dr, const float base_in[4],
                           const float world[3], const float nrm_in[3], int front_facing,
                           float roughness_in, float metallic_in, float occlusion, const float emissive[3],
                           float color[4], float normal_out[4])
{
    const slhip_scene* sc = cx->sc;
    float base[4] = {base_in[0], base_in[1], base_in[2], base_in[3]};
    float normal[3] = {nrm_in[0], nrm_in[1], nrm_in[2]};
    if (!front_facing) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }

    float V[3] = {sc->cam_position[0] - world[0], sc->cam_position[1] - world[1],
                  sc->cam_position[2] - world[2]};
    normalize3(V);
    float NoV = clampf(dot3(normal, V), 1e-5f, 1.0f);

    float roughness = fmaxf(roughness_in, 0.045f);
    float metallic = metallic_in;

    color[0] = color[1] = color[2] = 0.0f;
    color[3] = base[3];

    float F0[3], kS[3];
    float p5 = powf(1.0f - NoV, 5.0f);
    for (int c = 0; c < 3; ++c) {
        F0[c] = 0.04f * (1.0f - metallic) + base[c] * metallic;
        float Fr = fmaxf(1.0f - roughness, F0[c]) - F0[c];
        kS[c] = F0[c] + Fr * p5;
    }

    /* Lambert term base / pi once per pixel; the divisions below are one reciprocal each */
    const float base_pi[3] = {base[0] / REF_PI, base[1] / REF_PI, base[2] / REF_PI};
    for (int i = 0; i < SLHIP_NUM_LIGHTS; ++i) {
        const float* lc = sc->light_color[i];
        const float* ld = sc->light_dir[i];
        if ((lc[0] == 0.0f && lc[1] == 0.0f && lc[2] == 0.0f) ||
            (ld[0] == 0.0f && ld[1] == 0.0f && ld[2] == 0.0f))
            continue;

        float inverse_shadow = 1.0f;
        if ((cx->flags & SLHIP_RENDER_SHADOWS) && cx->shadow) {
            float w4[4] = {world[0], world[1], world[2], 1.0f}, pc[4];
            mv4(sc->shadow_mat[i], w4, pc);
            const float rpw = 1.0f / pc[3];   /* shared by the three perspective divisions */
            float px = fmaf(pc[0] * rpw, 0.5f, 0.5f);
            float py = fmaf(pc[1] * rpw, 0.5f, 0.5f);
            float pz = fmaf(pc[2] * rpw, 0.5f, 0.5f);
            const float* sm = cx->shadow + (size_t)i * cx->S * cx->S;
            float scale = 1.0f / (float)cx->S;
            float acc = 0.0f;
            for (int yy = 0; yy < 4; ++yy)
                for (int xx = 0; xx < 4; ++xx) {
                    float ox = (-1.5f + (float)xx) * scale, oy = (-1.5f + (float)yy) * scale;
                    acc += shadow_tap(sm, cx->S, px + ox, py + oy, pz - 0.00003f);
                }
            inverse_shadow = acc / 16.0f;
        }

        float L[3] = {-ld[0], -ld[1], -ld[2]};
        normalize3(L);
        float Hv[3] = {V[0] + L[0], V[1] + L[1], V[2] + L[2]};
        normalize3(Hv);
        float NDF = distribution_ggx(normal, Hv, roughness);
        float G = geometry_smith(normal, V, L, roughness);
        float NdotL = fmaxf(dot3(normal, L), 0.0f);
        float denominator = fmaxf(4.0f * NoV * NdotL, 0.001f);
        const float rden = 1.0f / denominator;
        for (int c = 0; c < 3; ++c) {
            float specular = (NDF * G * kS[c]) * rden;
            float kD = (1.0f - kS[c]) * (1.0f - metallic);
            color[c] += inverse_shadow * (kD * base_pi[c] + specular) * lc[c] * NdotL;
        }
    }
    for (int c = 0; c < 3; ++c) color[c] += sc->ambient[c] * base[c];
    if (sc->light_map != 0u && cx->pool->d_light_maps) {
        /* image-based lighting (render_shader.frag:375-394); sampling rules: include/slhip.h, slhip_light_map */
        const slhip_light_map* lm = cx->pool->d_light_maps + (sc->light_map - 1u);
        const float d2 = 2.0f * dot3(normal, V);
        cm3 refl = {d2 * normal[0] - V[0], d2 * normal[1] - V[1], d2 * normal[2] - V[2]};
        float fab[2];
        {
            const int n = (int)lm->lut_size;
            const float x = NoV * (float)n - 0.5f, y = roughness * (float)n - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const float a = x - fx, b = y - fy;
            int x0 = (int)fx, x1 = (int)fx + 1, y0 = (int)fy, y1 = (int)fy + 1;
            if (x0 < 0) x0 = 0; if (x0 > n - 1) x0 = n - 1; if (x1 < 0) x1 = 0; if (x1 > n - 1) x1 = n - 1;
            if (y0 < 0) y0 = 0; if (y0 > n - 1) y0 = n - 1; if (y1 < 0) y1 = 0; if (y1 > n - 1) y1 = n - 1;
            const float* t = lm->d_brdf_lut;
            for (int k = 0; k < 2; ++k)
                fab[k] = cm_bil(a, b, t[2 * (y0 * n + x0) + k], t[2 * (y0 * n + x1) + k], t[2 * (y1 * n + x0) + k], t[2 * (y1 * n + x1) + k]);
        }
        const cm4 rad4 = cm_sample_lod(lm->d_prefilter, lm->pre_size, lm->pre_levels, refl, roughness * 4.0f);
        cm3 nd = {normal[0], normal[1], normal[2]};
        const cm4 irr4 = cm_sample_lod(lm->d_irradiance, lm->irr_size, 1u, nd, 0.0f);
        const float rad[3] = {rad4.x, rad4.y, rad4.z}, irr[3] = {irr4.x, irr4.y, irr4.z};
        const float Ems = 1.0f - (fab[0] + fab[1]);
        for (int c = 0; c < 3; ++c) {
            const float c_diff = base[c] * (1.0f - 0.04f) * (1.0f - metallic);
            const float FssEss = kS[c] * fab[0] + fab[1];
            const float F_avg = F0[c] + (1.0f - F0[c]) / 21.0f;
            const float FmsEms = Ems * FssEss * F_avg / (1.0f - F_avg * Ems);
            const float k_D = c_diff * (1.0f - FssEss - FmsEms);
            color[c] += (FssEss * rad[c] + (FmsEms + k_D) * irr[c]) * occlusion;
        }
    }
    for (int c = 0; c < 3; ++c) color[c] += emissive[c];

    /* normalOut (frag:404-405): camera-frame normal and n.v */
    float nc[3];
    for (int r = 0; r < 3; ++r) {
        float a = fmaf(sc->world_to_cam[4 * r + 0], normal[0], 0.0f);
        a = fmaf(sc->world_to_cam[4 * r + 1], normal[1], a);
        a = fmaf(sc->world_to_cam[4 * r + 2], normal[2], a);
        nc[r] = a;
    }
    normalize3(nc);
    normal_out[0] = nc[0]; normal_out[1] = nc[1]; normal_out[2] = nc[2];
    normal_out[3] = dot3(normal, V);
}

/* ------------------------------------------------------------------------------------------ */
/* shadow pass (render_pass.cpp:408-460): depth-only, FRONT faces culled, D32F, LESS           */
/* ------------------------------------------------------------------------------------------ */
static void shadow_pass(const slhip_mesh_pool* pool, const slhip_scene* sc, const slhip_draw* draws,
                        int light, float* sm, int S)
{
    for (size_t i = 0; i < (size_t)S * S; ++i) sm[i] = 1.0f;
    for (uint32_t di = sc->draw_begin; di < sc->draw_end; ++di) {
        const slhip_draw* dr = &draws[di];
        if (!(dr->flags & SLHIP_DRAW_CASTS_SHADOW)) continue;
        /* transformation = shadowMatrix * absoluteTransformation (render_pass.cpp:441-444) */
        float OM[16], T[16];
        mm4(dr->object_to_world, dr->mesh_to_object, OM);
        mm4(sc->shadow_mat[light], OM, T);
        for (uint32_t t = 0; t < dr->n_tris; ++t) {
            clip_vert cv[3];
            for (int k = 0; k < 3; ++k) {
                uint32_t v = dr->vtx_base + pool->d_idx[dr->idx_base + 3 * (size_t)t + k];
                const float* p = pool->d_pos + 4 * (size_t)v;
                float pos[4] = {p[0], p[1], p[2], 1.0f};
                mv4(T, pos, cv[k].clip);
                cv[k].bary[0] = cv[k].bary[1] = cv[k].bary[2] = 0.0f;
            }
            tri_setup ts;
            if (!setup_triangle(&cv[0], &cv[1], &cv[2], S, S, &ts)) continue;
            if (ts.flipped) continue; /* front face (CW => area2 < 0) culled */
            for (int py = ts.ymin; py <= ts.ymax; ++py)
                for (int px = ts.xmin; px <= ts.xmax; ++px) {
                    float l[3];
                    if (!coverage(&ts, px, py, l)) continue;
                    float z = interp(l, ts.z[0], ts.z[1], ts.z[2]);
                    if (!(z >= 0.0f && z <= 1.0f)) continue;
                    float* d = &sm[(size_t)py * S + px];
                    if (z < *d) *d = z;
                }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SSAO kernel + noise: std::mt19937{0xdeadbeef} + uniform_real_distribution<float>(0,1)       */
/* (ssao_shader.cpp:72-112); libstdc++'s generate_canonical<float,24> = float(u32) / 2^32,      */
/* clamped below 1.                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937_t* g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static float mt_uniform01(mt19937_t* g)
{
    float r = (float)mt_next(g) / 4294967296.0f;
    if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
    return r;
}

/* out_noise: float[16*3], out_kernel: float[64*3] */
void slref_ssao_tables(float* out_noise, float* out_kernel)
{
    mt19937_t g;
    mt_seed(&g, 0xdeadbeefu);
    for (int i = 0; i < 16; ++i) {
        out_noise[3 * i + 0] = 2.0f * mt_uniform01(&g) - 1.0f;
        out_noise[3 * i + 1] = 2.0f * mt_uniform01(&g) - 1.0f;
        out_noise[3 * i + 2] = 0.0f;
    }
    for (int i = 0; i < 64; ++i) {
        float s[3];
        s[0] = 2.0f * mt_uniform01(&g) - 1.0f;
        s[1] = 2.0f * mt_uniform01(&g) - 1.0f;
        s[2] = mt_uniform01(&g);
        float len = sqrtf(dot3(s, s));
        float r = mt_uniform01(&g);
        float scale = (float)i / 64.0f;
        /* Math::lerp(0.1, 1.0, t) = (1-t)*a + t*b with t = scale^2 */
        float lerp = (1.0f - scale * scale) * 0.1f + (scale * scale) * 1.0f;
        for (int c = 0; c < 3; ++c) out_kernel[3 * i + c] = (r * (s[c] / len)) * lerp;
    }
}

/* bilinear fetch of channel `ch` from an RGBA32F rectangle texture, clamp-to-edge, at texel
   coordinates (x,y) in pixels (sampler2DRect, LINEAR; Appendix B of SURVEY.md) */
/* rect sampler, LINEAR: `xs`, `ys` are the coordinates already shifted to texel centres (x - 0.5, y - 0.5) */
static float rect_bilinear_s(const float* img, int W, int H, float xs, float ys, int ch)
{
    float fx = floorf(xs), fy = floorf(ys);
    float ax = xs - fx, ay = ys - fy;
    /* clamp-to-edge BEFORE the float -> int conversion: the same texels for every finite coordinate, and a
       defined result (edge texel; 0 for NaN) when a sample projects to infinity -- (int)inf is undefined
       in C and saturates on the GPU */
    const float wm = (float)(W - 1), hm = (float)(H - 1);
    int x0 = (int)clampf(fx, 0.0f, wm), x1 = (int)clampf(fx + 1.0f, 0.0f, wm);
    int y0 = (int)clampf(fy, 0.0f, hm), y1 = (int)clampf(fy + 1.0f, 0.0f, hm);
    float a = img[4 * ((size_t)y0 * W + x0) + ch], b = img[4 * ((size_t)y0 * W + x1) + ch];
    float c = img[4 * ((size_t)y1 * W + x0) + ch], d = img[4 * ((size_t)y1 * W + x1) + ch];
    float top = fmaf(ax, b - a, a), bot = fmaf(ax, d - c, c);
    return fmaf(ay, bot - top, top);
}

static float rect_bilinear(const float* img, int W, int H, float x, float y, int ch)
{
    return rect_bilinear_s(img, W, H, x - 0.5f, y - 0.5f, ch);
}

/* Reciprocal of the SSAO taps' perspective division: |x| through three Newton steps r <- r + r (1 - |x| r) from the
   integer-subtraction seed (relative error ~1e-7, i.e. 1e-4 px at 640 px), the sign put back -- an integer subtraction, six
   multiply-adds and a bit select instead of an IEEE division (11 instructions + a quarter-rate reciprocal on the GPU) in the
   innermost loop of the most expensive image pass.  GLSL leaves the precision of a division to the driver (2.5 ulp, GLSL 4.5
   section 4.7.1); zero, infinities and NaN give NaN or garbage coordinates, which the sampler clamps like any other. */
static inline float ssao_rcp(float x)
{
    uint32_t xi, ri;
    float a, r;
    memcpy(&xi, &x, 4);
    const uint32_t ai = xi & 0x7fffffffu;
    ri = 0x7EF311C7u - ai;
    memcpy(&a, &ai, 4);
    memcpy(&r, &ri, 4);
    for (int k = 0; k < 3; ++k) {
        const float e = fmaf(-a, r, 1.0f);
        r = fmaf(r, e, r);
    }
    memcpy(&ri, &r, 4);
    ri = (ri & 0x7fffffffu) | (xi & 0x80000000u);
    memcpy(&r, &ri, 4);
    return r;
}

static float smoothstep01(float x)
{
    float t = clampf(x, 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

/* ssao_shader.frag:20-56.  cam: [H,W,4] camCoordinates, nrm: [H,W,4] normals, ao: [H,W] */
static void ssao_pass(const float* proj, const float* cam, const float* nrm, int W, int H, float* ao)
{
    float noise[48], kern[192];
    slref_ssao_tables(noise, kern);
    const float radius = 0.1f, bias = 0.0025f;
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)j * W + i;
            float frag[3] = {cam[4 * p], cam[4 * p + 1], cam[4 * p + 2]};
            float n[3] = {nrm[4 * p], nrm[4 * p + 1], nrm[4 * p + 2]};
            /* DEVIATION (documented in DESIGN.md): pixels without geometry have a zero normal;
               GLSL normalize(0) is undefined (NaN on most drivers, which then bleeds into the
               2-px silhouette band through the blur).  We define AO = 1 there (the clear
               value of the AO target, render_pass.cpp:671). */
            if (n[0] == 0.0f && n[1] == 0.0f && n[2] == 0.0f) { ao[p] = 1.0f; continue; }
            normalize3(n);
            const float* rv0 = &noise[3 * ((j & 3) * 4 + (i & 3))];
            float rv[3] = {rv0[0], rv0[1], rv0[2]};
            normalize3(rv);
            float d = dot3(rv, n);
            float tg[3] = {rv[0] - n[0] * d, rv[1] - n[1] * d, rv[2] - n[2] * d};
            normalize3(tg);
            float bt[3] = {n[1] * tg[2] - n[2] * tg[1], n[2] * tg[0] - n[0] * tg[2],
                           n[0] * tg[1] - n[1] * tg[0]};
            /* The sample position frag + radius * TBN * kernel[k] is linear in the kernel vector, and so is
               its projection: both are evaluated as fma chains over per-pixel bases (rows 0, 1, 3 of the
               projection; row 2 is not needed).  The perspective division is one reciprocal shared by x
               and y, and ((q * 0.5 + 0.5) * W) is a single fma -- GLSL leaves all of this to the compiler. */
            float tgR[3], btR[3], nR[3];
            for (int c = 0; c < 3; ++c) { tgR[c] = tg[c] * radius; btR[c] = bt[c] * radius; nR[c] = n[c] * radius; }
            const float f4[4] = {frag[0], frag[1], frag[2], 1.0f};
            float A[4], B[3], C[3], D[3];
            mv4(proj, f4, A);
            {
                static const int rows[3] = {0, 1, 3};
                for (int q = 0; q < 3; ++q) {
                    const float* m = proj + 4 * rows[q];
                    B[q] = dot3(m, tgR); C[q] = dot3(m, btR); D[q] = dot3(m, nR);
                }
            }
            const float A3[3] = {A[0], A[1], A[3]};
            /* window coordinates of a tap, already shifted to texel centres: (q * 0.5 + 0.5) * W - 0.5 as ONE fma */
            const float hw = 0.5f * (float)W, hh = 0.5f * (float)H, hwm = hw - 0.5f, hhm = hh - 0.5f;
            float occlusion = 0.0f;
            for (int k = 0; k < 64; ++k) {
                const float* s = &kern[3 * k];
                const float spz = fmaf(nR[2], s[2], fmaf(btR[2], s[1], fmaf(tgR[2], s[0], frag[2])));
                float o[3];
                for (int q = 0; q < 3; ++q) o[q] = fmaf(D[q], s[2], fmaf(C[q], s[1], fmaf(B[q], s[0], A3[q])));
                const float rw = ssao_rcp(o[2]);
                const float xs = fmaf(o[0] * rw, hw, hwm), ys = fmaf(o[1] * rw, hh, hhm);
                float sd = rect_bilinear_s(cam, W, H, xs, ys, 2);
                float rc = smoothstep01(radius / fabsf(frag[2] - sd));
                occlusion += (sd <= spz - bias ? 1.0f : 0.0f) * rc;
            }
            ao[p] = 1.0f - occlusion / 64.0f;
        }
}

/* The unblurred occlusion of one picture (tests: the analytic inner-corner known answer looks at the pass itself, not at the
   colour it ends up in).  proj: row-major 4 x 4, cam / nrm: [H, W, 4] as slref_render writes them, ao: [H, W]. */
void slref_ssao_pass(const float* proj, const float* cam, const float* nrm, int W, int H, float* ao) { ssao_pass(proj, cam, nrm, W, H, ao); }

/* ssao_apply_shader.frag:29-75: 4x4 bilateral blur (offsets -2..1), multiplies rgb.
   texelFetch outside the image is undefined in GL; we clamp to the edge. */
static void ssao_apply(const float* hdr_in, const float* ao, const float* cam, int W, int H,
                       float* hdr_out)
{
    const float sigma = 3.0f * 0.5f;
    const float falloff = 1.0f / (2.0f * sigma * sigma);
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            size_t p = (size_t)j * W + i;
            /* center_d = texture(coordinateSampler, ivec2) -- integer coordinate on a LINEAR
               rect sampler, i.e. the corner between 4 texels (ssao_apply_shader.frag:50) */
            float cd = rect_bilinear(cam, W, H, (float)i, (float)j, 2);
            float result = 0.0f, wt = 0.0f;
            for (int x = -2; x < 2; ++x)
                for (int y = -2; y < 2; ++y) {
                    int xi = i + x, yj = j + y;
                    if (xi < 0) xi = 0; if (xi > W - 1) xi = W - 1;
                    if (yj < 0) yj = 0; if (yj > H - 1) yj = H - 1;
                    size_t q = (size_t)yj * W + xi;
                    float c = ao[q];
                    /* texture(coordinateSampler, ivec) on a rect sampler at integer coords:
                       bilinear between the 4 texels around the corner */
                    float d = rect_bilinear(cam, W, H, (float)(i + x), (float)(j + y), 2);
                    float r = sqrtf((float)(x * x + y * y));
                    float dd = (d - cd) * 300.0f;
                    float w = exp2f(-r * r * falloff - dd * dd);
                    wt += w;
                    result += c * w;
                }
            float a = result / wt;
            hdr_out[4 * p + 0] = hdr_in[4 * p + 0] * a;
            hdr_out[4 * p + 1] = hdr_in[4 * p + 1] * a;
            hdr_out[4 * p + 2] = hdr_in[4 * p + 2] * a;
            hdr_out[4 * p + 3] = hdr_in[4 * p + 3];
        }
}

/* tone_map_shader.frag:102-131.  lum_src: the HDR image the average is taken from */
static void tone_map(const float* hdr, const float* lum_src, int W, int H, float manual_exposure,
                     uint8_t* rgb)
{
    float lum = 0.0f;
    if (!(manual_exposure >= 0.0f)) {
        /* 1x1 mip level == arithmetic mean, accumulated in double */
        double s[4] = {0, 0, 0, 0};
        for (size_t p = 0; p < (size_t)W * H; ++p)
            for (int c = 0; c < 4; ++c) s[c] += lum_src[4 * p + c];
        float avg[4];
        for (int c = 0; c < 4; ++c) avg[c] = (float)(s[c] / (double)((size_t)W * H));
        lum = 0.1f * (0.2125f * (avg[0] / avg[3]) + 0.7154f * (avg[1] / avg[3]) + 0.0721f * (avg[2] / avg[3]));
    }
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        const float* c = &hdr[4 * p];
        float X = 0.4124564f * c[0] + 0.3575761f * c[1] + 0.1804375f * c[2];
        float Y = 0.2126729f * c[0] + 0.7151522f * c[1] + 0.0721750f * c[2];
        float Z = 0.0193339f * c[0] + 0.1191920f * c[1] + 0.9503041f * c[2];
        float inv = 1.0f / (X + Y + Z);
        float Yxy[3] = {Y, X * inv, Y * inv};
        if (manual_exposure >= 0.0f) Yxy[0] *= manual_exposure;
        else Yxy[0] /= (9.6f * lum + 0.0001f);
        float x2 = Yxy[0] * Yxy[1] / Yxy[2];
        float y2 = Yxy[0];
        float z2 = Yxy[0] * (1.0f - Yxy[1] - Yxy[2]) / Yxy[2];
        float o[3];
        o[0] = 3.2404542f * x2 + -1.5371385f * y2 + -0.4985314f * z2;
        o[1] = -0.9692660f * x2 + 1.8760108f * y2 + 0.0415560f * z2;
        o[2] = 0.0556434f * x2 + -0.2040259f * y2 + 1.0572252f * z2;
        for (int k = 0; k < 3; ++k) {
            float x = o[k];
            float v = (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f);
            v = fminf(fmaxf(v, 0.0f), 1.0f); /* NaN -> 0 (fmaxf returns the non-NaN operand) */
            rgb[4 * p + k] = (uint8_t)floorf(v * 255.0f + 0.5f);
        }
        float a = fminf(fmaxf(c[3], 0.0f), 1.0f);
        rgb[4 * p + 3] = (uint8_t)floorf(a * 255.0f + 0.5f);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* main entry: renders n_scenes scenes.  Same signature family as slhip_render, but all
 * pointers are HOST pointers and scratch is allocated internally.  hdr_out (optional,
 * f32 [B,H,W,4]) receives the pre-tone-map HDR colour for debugging.                           */
/* ------------------------------------------------------------------------------------------ */
int slref_render(const slhip_mesh_pool* pool, const slhip_scene* scenes, const slhip_draw* draws,
                 uint32_t n_scenes, uint32_t width, uint32_t height, uint32_t flags,
                 const float* depth_peel, const slhip_render_out* out, uint32_t shadow_res,
                 float* hdr_out, float* shadow_out)
{
    const int W = (int)width, H = (int)height;
    const size_t P = (size_t)W * H;
    uint32_t* depth = (uint32_t*)malloc(P * sizeof(uint32_t));
    float* hdr = (float*)malloc(P * 4 * sizeof(float));
    float* hdr2 = (float*)malloc(P * 4 * sizeof(float));
    float* hdr_lum = (float*)malloc(P * 4 * sizeof(float)); /* the image the auto-exposure average is taken from */
    float* aobuf = (float*)malloc(P * sizeof(float));
    float* camc = (float*)malloc(P * 4 * sizeof(float));
    float* nrmb = (float*)malloc(P * 4 * sizeof(float));
    float* shadow = NULL;
    const int S = (int)shadow_res;
    if (flags & SLHIP_RENDER_SHADOWS) shadow = (float*)malloc((size_t)SLHIP_NUM_LIGHTS * S * S * sizeof(float));
    if (!depth || !hdr || !hdr2 || !aobuf || !camc || !nrmb) return -1;

    for (uint32_t s = 0; s < n_scenes; ++s) {
        const slhip_scene* sc = &scenes[s];
        const size_t base = (size_t)s * P;
        /* clears (render_pass.cpp:523-532) */
        for (size_t p = 0; p < P; ++p) {
            depth[p] = 0xFFFFFFu; /* cleared to 1.0; LESS => a fragment at exactly 1.0 loses */
            for (int c = 0; c < 4; ++c) {
                hdr[4 * p + c] = 0.0f;
                camc[4 * p + c] = INVALID_VALUE;
                nrmb[4 * p + c] = 0.0f;
                if (out->d_coord) out->d_coord[4 * (base + p) + c] = INVALID_VALUE;
                if (out->d_vertex_idx) out->d_vertex_idx[4 * (base + p) + c] = 0;
                if (out->d_bary) out->d_bary[4 * (base + p) + c] = 0.0f;
            }
            if (out->d_class) out->d_class[base + p] = 0;
            if (out->d_instance) out->d_instance[base + p] = 0;
        }

        /* shadow pass */
        if (shadow) {
            for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l) {
                const float* lc = sc->light_color[l];
                const float* ld = sc->light_dir[l];
                float* sm = shadow + (size_t)l * S * S;
                if ((lc[0] == 0.0f && lc[1] == 0.0f && lc[2] == 0.0f) ||
                    (ld[0] == 0.0f && ld[1] == 0.0f && ld[2] == 0.0f)) {
                    for (size_t i = 0; i < (size_t)S * S; ++i) sm[i] = 1.0f;
                    continue;
                }
                shadow_pass(pool, sc, draws, l, sm, S);
            }
            if (shadow_out)
                memcpy(shadow_out + (size_t)s * SLHIP_NUM_LIGHTS * S * S, shadow,
                       (size_t)SLHIP_NUM_LIGHTS * S * S * sizeof(float));
        }

        shade_ctx cx = {pool, sc, W, H, flags, shadow, S};

        /* main pass, strict draw order */
        for (uint32_t di = sc->draw_begin; di < sc->draw_end; ++di) {
            const slhip_draw* dr = &draws[di];
            const uint8_t* tex = (dr->flags & SLHIP_DRAW_HAS_BASE_TEX) ? pool->d_tex + dr->tex_offset : NULL;
            for (uint32_t t = 0; t < dr->n_tris; ++t) {
                uint32_t vi[3];
                vs_out vo[3];
                clip_vert cv[3];
                for (int k = 0; k < 3; ++k) {
                    vi[k] = pool->d_idx[dr->idx_base + 3 * (size_t)t + k];
                    vertex_stage(pool, sc, dr, dr->vtx_base + vi[k], &vo[k]);
                    memcpy(cv[k].clip, vo[k].clip, sizeof(float) * 4);
                    cv[k].bary[0] = k == 0; cv[k].bary[1] = k == 1; cv[k].bary[2] = k == 2;
                }
                clip_vert poly[4];
                int n = clip_near(cv, poly);
                if (n == 0) continue;
                for (int sub = 0; sub < n - 2; ++sub) {
                    tri_setup ts;
                    if (!setup_triangle(&poly[0], &poly[sub + 1], &poly[sub + 2], W, H, &ts)) continue;
                    const int front_facing = ts.flipped; /* FrontFace = CW => area2 < 0 is front */
                    for (int py = ts.ymin; py <= ts.ymax; ++py)
                        for (int px = ts.xmin; px <= ts.xmax; ++px) {
                            float l[3];
                            if (!coverage(&ts, px, py, l)) continue;
                            float z = interp(l, ts.z[0], ts.z[1], ts.z[2]);
                            if (!(z >= 0.0f && z <= 1.0f)) continue;
                            float fd = floorf(fmaf(z, 16777215.0f, 0.5f));
                            uint32_t d24 = (uint32_t)fd;
                            if (d24 > 0xFFFFFFu) d24 = 0xFFFFFFu;
                            size_t p = (size_t)py * W + px;
                            if (!(d24 < depth[p])) continue; /* early-z is equivalent: discards below do not write depth */

                            /* perspective-correct barycentrics of the sub-triangle ... */
                            float pw0 = l[0] * ts.invw[0], pw1 = l[1] * ts.invw[1], pw2 = l[2] * ts.invw[2];
                            float sw = (pw0 + pw1) + pw2;
                            float bs[3] = {pw0 * (1.0f / sw), pw1 * (1.0f / sw), pw2 * (1.0f / sw)};
                            /* ... mapped to the original triangle */
                            float b[3];
                            for (int k = 0; k < 3; ++k)
                                b[k] = fmaf(bs[2], ts.bary[2][k], fmaf(bs[1], ts.bary[1][k], bs[0] * ts.bary[0][k]));

                            float camz = interp(b, vo[0].objc[3], vo[1].objc[3], vo[2].objc[3]);
                            /* depth peeling (frag:229-233) */
                            if (depth_peel && (camz - 0.00001f <= depth_peel[4 * (base + p) + 3])) continue;

                            float basec[4] = {dr->base_color[0], dr->base_color[1], dr->base_color[2], dr->base_color[3]};
                            const float u = interp(b, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]);
                            const float v = interp(b, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]);
                            float bxn[3], byn[3];
                            bary_at(&ts, px + 1, py, bxn);
                            bary_at(&ts, px, py + 1, byn);
                            const float dudx = interp(bxn, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]) - u, dvdx = interp(bxn, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]) - v;
                            const float dudy = interp(byn, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]) - u, dvdy = interp(byn, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]) - v;
                            if (tex) {
                                float tc[4];
                                tex_sample(tex, (int)dr->tex_w, (int)dr->tex_h, dr->tex_sampler[0], u, v, dudx, dvdx, dudy, dvdy, tc);
                                basec[0] *= powf(tc[0], 2.2f);
                                basec[1] *= powf(tc[1], 2.2f);
                                basec[2] *= powf(tc[2], 2.2f);
                                basec[3] *= tc[3];
                            }
                            if (basec[3] < dr->alpha_cutoff) continue; /* frag:242-246 */

                            depth[p] = d24;

                            float world[3], nrm[3], objc[3], cam[3];
                            for (int k = 0; k < 3; ++k) {
                                world[k] = interp(b, vo[0].world[k], vo[1].world[k], vo[2].world[k]);
                                nrm[k] = interp(b, vo[0].nrm[k], vo[1].nrm[k], vo[2].nrm[k]);
                                objc[k] = interp(b, vo[0].objc[k], vo[1].objc[k], vo[2].objc[k]);
                                cam[k] = interp(b, vo[0].cam[k], vo[1].cam[k], vo[2].cam[k]);
                            }
                            if (dr->flags & SLHIP_DRAW_HAS_STICKER) { /* vert:89-94, frag:248-256 */
                                float sx[3], sy[3];
                                for (int k = 0; k < 3; ++k) {
                                    const float* pp = pool->d_pos + 4 * (size_t)(dr->vtx_base + vi[k]);
                                    const float pos[4] = {pp[0], pp[1], pp[2], 1.0f};
                                    float obj4[4], sp[4];
                                    mv4(dr->mesh_to_object, pos, obj4);
                                    mv4(dr->sticker_projection, obj4, sp);
                                    sx[k] = (sp[0] / sp[3] - dr->sticker_range[0]) / dr->sticker_range[2];
                                    sy[k] = (sp[1] / sp[3] - dr->sticker_range[1]) / dr->sticker_range[3];
                                }
                                const float cxs = interp(b, sx[0], sx[1], sx[2]), cys = interp(b, sy[0], sy[1], sy[2]);
                                if (cxs >= 0.0f && cys >= 0.0f && cxs < 1.0f && cys < 1.0f) {
                                    float st[4];
                                    tex_rect_bilinear(pool->d_tex + dr->sticker_tex_offset, (int)dr->sticker_tex_w, (int)dr->sticker_tex_h,
                                                      cxs * (float)dr->sticker_tex_w, cys * (float)dr->sticker_tex_h, st);
                                    const float sc4[4] = {powf(st[0], 2.2f), powf(st[1], 2.2f), powf(st[2], 2.2f), st[3]};
                                    for (int c = 0; c < 4; ++c) basec[c] = basec[c] * (1.0f - st[3]) + sc4[c] * st[3];
                                }
                            }
                            if (dr->flags & SLHIP_DRAW_HAS_NORMAL_TEX) { /* vert:77-80, frag:262-266 */
                                float tw[3][3], bw[3][3];
                                for (int k = 0; k < 3; ++k) {
                                    const float* t4 = pool->d_tan + 4 * (size_t)(dr->vtx_base + vi[k]);
                                    const float tv[3] = {t4[0], t4[1], t4[2]};
                                    mv3p(dr->normal_to_world, tv, tw[k]);
                                    normalize3(tw[k]);
                                    bw[k][0] = vo[k].nrm[1] * tw[k][2] - vo[k].nrm[2] * tw[k][1];
                                    bw[k][1] = vo[k].nrm[2] * tw[k][0] - vo[k].nrm[0] * tw[k][2];
                                    bw[k][2] = vo[k].nrm[0] * tw[k][1] - vo[k].nrm[1] * tw[k][0];
                                    normalize3(bw[k]);
                                    for (int c = 0; c < 3; ++c) bw[k][c] *= t4[3];
                                }
                                float tc[4];
                                tex_sample(pool->d_tex + dr->normal_tex_offset, (int)dr->normal_tex_w, (int)dr->normal_tex_h, dr->tex_sampler[1], u, v, dudx, dvdx, dudy, dvdy, tc);
                                const float nx = tc[0] * 2.0f - 1.0f, ny = tc[1] * 2.0f - 1.0f, nz = tc[2] * 2.0f - 1.0f;
                                float nn[3];
                                for (int c = 0; c < 3; ++c)
                                    nn[c] = nx * interp(b, tw[0][c], tw[1][c], tw[2][c]) + ny * interp(b, bw[0][c], bw[1][c], bw[2][c]) + nz * nrm[c];
                                normalize3(nn);
                                nrm[0] = nn[0]; nrm[1] = nn[1]; nrm[2] = nn[2];
                            }
                            float roughness = dr->roughness, metallic = dr->metallic, occlusion = 1.0f;
                            float emissive[3] = {dr->emissive[0], dr->emissive[1], dr->emissive[2]};
                            if (dr->flags & SLHIP_DRAW_HAS_MR_TEX) { /* frag:284-288 */
                                float tc[4];
                                tex_sample(pool->d_tex + dr->mr_tex_offset, (int)dr->mr_tex_w, (int)dr->mr_tex_h, dr->tex_sampler[2], u, v, dudx, dvdx, dudy, dvdy, tc);
                                roughness *= tc[1];
                                metallic *= tc[2];
                            }
                            if (dr->flags & SLHIP_DRAW_HAS_OCCLUSION_TEX) { /* frag:292-294 */
                                float tc[4];
                                tex_sample(pool->d_tex + dr->occlusion_tex_offset, (int)dr->occlusion_tex_w, (int)dr->occlusion_tex_h, dr->tex_sampler[3], u, v, dudx, dvdx, dudy, dvdy, tc);
                                occlusion = tc[0];
                            }
                            if (dr->flags & SLHIP_DRAW_HAS_EMISSIVE_TEX) { /* frag:296-298 */
                                float tc[4];
                                tex_sample(pool->d_tex + dr->emissive_tex_offset, (int)dr->emissive_tex_w, (int)dr->emissive_tex_h, dr->tex_sampler[4], u, v, dudx, dvdx, dudy, dvdy, tc);
                                for (int c = 0; c < 3; ++c) emissive[c] *= powf(tc[c], 2.2f);
                            }
                            float color[4], nout[4];
                            shade_fragment(&cx, dr, basec, world, nrm, front_facing, roughness, metallic, occlusion, emissive, color, nout);
                            for (int c = 0; c < 4; ++c) hdr[4 * p + c] = color[c];
                            camc[4 * p + 0] = cam[0]; camc[4 * p + 1] = cam[1];
                            camc[4 * p + 2] = cam[2]; camc[4 * p + 3] = 1.0f;
                            for (int c = 0; c < 4; ++c) nrmb[4 * p + c] = nout[c];
                            if (out->d_coord) {
                                float* o = out->d_coord + 4 * (base + p);
                                o[0] = objc[0]; o[1] = objc[1]; o[2] = objc[2]; o[3] = camz;
                            }
                            if (out->d_class) out->d_class[base + p] = (uint16_t)dr->class_index;
                            if (out->d_instance) out->d_instance[base + p] = (uint16_t)dr->instance_index;
                            if (out->d_vertex_idx) {
                                uint32_t* o = out->d_vertex_idx + 4 * (base + p);
                                if (dr->flags & SLHIP_DRAW_NO_VERTEX_ID) { o[0] = o[1] = o[2] = o[3] = 0; }
                                else { o[0] = vi[0] + 1; o[1] = vi[1] + 1; o[2] = vi[2] + 1; o[3] = 0; }
                            }
                            if (out->d_bary) {
                                float* o = out->d_bary + 4 * (base + p);
                                o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = 0.0f;
                            }
                        }
                }
            }
        }

        /* the exposure average (1x1 mip level) is generated BEFORE any background is drawn
           (render_pass.cpp:632-635 precede :637-661): it sees the objects over the cleared target only */
        memcpy(hdr_lum, hdr, P * 4 * sizeof(float));

        /* background image (render_pass.cpp:637-646, background_shader.vert/frag): the rectangle texture
           stretched over the viewport, integer texel coordinates on a LINEAR rectangle sampler, alpha 0.
           DEVIATION: the reference draws the quad at NDC z = 0 with the depth test on, which also paints over
           every object farther than 0.198 m; here the image fills the pixels where nothing was rasterised. */
        if (sc->bg_tex[1] != 0u && sc->bg_tex[2] != 0u) {
            const int tw = (int)sc->bg_tex[1], th = (int)sc->bg_tex[2];
            for (int py = 0; py < H; ++py)
                for (int px = 0; px < W; ++px) {
                    const size_t p = (size_t)py * W + px;
                    if (depth[p] != 0xFFFFFFu) continue;
                    const float tcx = ((float)px + 0.5f) / (float)W, tcy = 1.0f - ((float)py + 0.5f) / (float)H;
                    const int tx = (int)(tcx * (float)tw), ty = (int)(tcy * (float)th);
                    float c[4];
                    tex_rect_bilinear(pool->d_tex + sc->bg_tex[0], tw, th, (float)tx, (float)ty, c);
                    hdr[4 * p + 0] = c[0]; hdr[4 * p + 1] = c[1]; hdr[4 * p + 2] = c[2]; hdr[4 * p + 3] = 0.0f;
                }
        } else
        /* sky background (render_pass.cpp:647-661): drawn at depth 1 with LEQUAL, i.e. wherever nothing was
           rasterised; colour attachment 0 only, alpha 0 */
        if (sc->light_map != 0u && pool->d_light_maps) {
            const slhip_light_map* lm = pool->d_light_maps + (sc->light_map - 1u);
            const float* m = sc->world_to_cam;
            for (int py = 0; py < H; ++py)
                for (int px = 0; px < W; ++px) {
                    const size_t p = (size_t)py * W + px;
                    if (depth[p] != 0xFFFFFFu) continue;   /* something was rasterised here */
                    const float xn = (2.0f * ((float)px + 0.5f)) / (float)W - 1.0f, yn = (2.0f * ((float)py + 0.5f)) / (float)H - 1.0f;
                    const float dcx = (xn - sc->proj[2]) / sc->proj[0], dcy = (yn - sc->proj[6]) / sc->proj[5];
                    cm3 dw = {fmaf(m[8], 1.0f, fmaf(m[4], dcy, m[0] * dcx)), fmaf(m[9], 1.0f, fmaf(m[5], dcy, m[1] * dcx)),
                              fmaf(m[10], 1.0f, fmaf(m[6], dcy, m[2] * dcx))};
                    const cm4 e = cm_sample_lod(lm->d_env, lm->env_size, lm->env_levels, dw, 0.0f);
                    hdr[4 * p + 0] = e.x; hdr[4 * p + 1] = e.y; hdr[4 * p + 2] = e.z; hdr[4 * p + 3] = 0.0f;
                }
        }

        if (out->d_normals) memcpy(out->d_normals + 4 * base, nrmb, P * 4 * sizeof(float));
        if (out->d_cam_coord) memcpy(out->d_cam_coord + 4 * base, camc, P * 4 * sizeof(float));

        /* post: SSAO -> tone map (render_pass.cpp:662-710) */
        const float* tm_in = hdr;
        if (flags & SLHIP_RENDER_SSAO) {
            ssao_pass(sc->proj, camc, nrmb, W, H, aobuf);
            ssao_apply(hdr, aobuf, camc, W, H, hdr2);
            tm_in = hdr2;
        }
        if (hdr_out) memcpy(hdr_out + 4 * base, tm_in, P * 4 * sizeof(float));
        if (out->d_rgb) tone_map(tm_in, hdr_lum, W, H, sc->manual_exposure, out->d_rgb + 4 * base);
    }
    free(depth); free(hdr); free(hdr2); free(aobuf); free(camc); free(nrmb); free(shadow);
    return 0;
}
