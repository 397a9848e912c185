// slhip_render.hip -- G-buffer render path for gfx950 (MI355X), replacing the Magnum/OpenGL
// RenderPass of the reference (src/render_pass.cpp:303-796 + src/shaders/render_shader.*).
//
// Design (see DESIGN.md "Render half"):
//   k_raster   one thread per triangle of a chunk (scene/draw are workgroup-uniform).  Vertex
//              transform chain, near clipping, 1/256-px snapping, exact integer edge functions.
//              Small triangles are resolved by the owning thread with a 64-bit atomicMin on the
//              visibility buffer (key = depth24 << 32 | primitive id, so the depth test AND the
//              GL draw-order tie-break are one atomic); larger ones are split into 8x8-pixel
//              tile items, compacted into a work queue with one wave-aggregated atomic per wave.
//   k_large    one wave per (triangle, 8x8 tile) item: lane == pixel.
//   k_shade    deferred: one thread per pixel decodes the winning primitive, re-runs the
//              vertex stage for its three vertices, reproduces the interpolation of the
//              fixed-function pipeline and evaluates the fragment shader; writes the selected
//              render targets with fully coalesced 16 B/lane stores.
//   k_shadow*  the same rasteriser, depth only, front faces culled, into the shadow maps.
//   k_ssao / k_ssao_apply / k_tonemap  image-space passes.
//
// All arithmetic that feeds integer outputs or depth follows rules R1..R7 of
// oracle/render_ref.c; this file is compiled with -ffp-contract=off so that only the explicit
// fmaf() calls fuse.
#include <vector>
#include <type_traits>
#include "slhip_common.h"
#include "slhip_cubemap.h"

// Occupancy ceilings of the render kernels (waves per SIMD; 0 = whatever the registers allow; build-time experiment knobs).  In
// the pipeline the settle stream is the critical path -- 2400 dependent launches whose single-wave blocks have to find a free
// wave slot and 100-170 free VGPRs on a SIMD -- and how fast it runs beside the render depends on what the render kernels leave
// free.  With two scenes per solver wave, k_ssao held to five waves per SIMD bought 2 % (9 507 -> 9 703 scenes/s; six: 9 212, four:
// 9 440); with ONE scene per solver wave (the default since) no ceiling wins: k_ssao at 8 / 7 / 6 / 5 waves 9 789 / 9 433 / 9 555 /
// 9 434, the other light kernels at 7 / 6: 9 795 / 9 631, k_shade at two waves: a loss.  All off: DESIGN.md section 4.
#ifndef SLHIP_RENDER_WAVES
#define SLHIP_RENDER_WAVES 0
#endif
#ifndef SLHIP_SHADE_WAVES
#define SLHIP_SHADE_WAVES 0
#endif
#ifndef SLHIP_SSAO_WAVES
#define SLHIP_SSAO_WAVES 0
#endif
#if SLHIP_RENDER_WAVES > 0
#define SLHIP_LIGHT_KERNEL __attribute__((amdgpu_waves_per_eu(1, SLHIP_RENDER_WAVES)))
#else
#define SLHIP_LIGHT_KERNEL
#endif
#if SLHIP_SSAO_WAVES > 0
#define SLHIP_SSAO_KERNEL __attribute__((amdgpu_waves_per_eu(1, SLHIP_SSAO_WAVES)))
#else
#define SLHIP_SSAO_KERNEL
#endif
#if SLHIP_SHADE_WAVES > 0
#define SLHIP_SHADE_KERNEL __attribute__((amdgpu_waves_per_eu(1, SLHIP_SHADE_WAVES)))
#else
#define SLHIP_SHADE_KERNEL
#endif
namespace {

static_assert(sizeof(slhip_draw) == 432, "slhip_draw layout");
static_assert(sizeof(slhip_scene) == 480, "slhip_scene layout");
static_assert(sizeof(slhip_chunk) == 16, "slhip_chunk layout");

constexpr float kInvalid = 3000.0f;  // render_pass.cpp:316
constexpr float kPi = 3.141592653589793f;
constexpr unsigned long long kVisEmpty = ~0ull;
constexpr int kSmallArea = 256;  // bbox pixels a single thread rasterises itself (larger boxes: tile queue); 128 / 256 / 1024 measured: shadow pass 17.2 / 16.7 / 16.4 ms

// ---------------------------------------------------------------------------------------------
// fixed-order arithmetic (R1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot3(const float* a, const float* b)
{
    float d = a[0] * b[0];
    d = fmaf(a[1], b[1], d);
    d = fmaf(a[2], b[2], d);
    return d;
}

__device__ __forceinline__ void mv4(const float* __restrict__ M, const float* v, float* o)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = fmaf(M[4 * r + 0], v[0], 0.0f);
        a = fmaf(M[4 * r + 1], v[1], a);
        a = fmaf(M[4 * r + 2], v[2], a);
        a = fmaf(M[4 * r + 3], v[3], a);
        o[r] = a;
    }
}

__device__ __forceinline__ void mv3p(const float* __restrict__ M, const float* v, float* o)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a = fmaf(M[4 * r + 0], v[0], 0.0f);
        a = fmaf(M[4 * r + 1], v[1], a);
        a = fmaf(M[4 * r + 2], v[2], a);
        o[r] = a;
    }
}

__device__ __forceinline__ void mm4(const float* __restrict__ A, const float* B, float* C)
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = fmaf(A[4 * r + 0], B[0 + c], 0.0f);
            a = fmaf(A[4 * r + 1], B[4 + c], a);
            a = fmaf(A[4 * r + 2], B[8 + c], a);
            a = fmaf(A[4 * r + 3], B[12 + c], a);
            C[4 * r + c] = a;
        }
}

__device__ __forceinline__ void normalize3(float* v)
{
    const float s = 1.0f / sqrtf(dot3(v, v));   // one reciprocal, three products (oracle normalize3)
    v[0] = v[0] * s;
    v[1] = v[1] * s;
    v[2] = v[2] * s;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ float interp(const float* b, float a0, float a1, float a2)
{
    return fmaf(b[2], a2 - a0, fmaf(b[1], a1 - a0, a0));
}

// ---------------------------------------------------------------------------------------------
// triangle setup (R2-R4)
// ---------------------------------------------------------------------------------------------
struct Setup {
    int X[3], Y[3];
    float z[3], invw[3];
    long long area2;  // > 0
    int flipped;      // original area2 < 0  (== front facing under FrontFace = CW)
    int bias[3];
    int xmin, xmax, ymin, ymax;
};

__device__ __forceinline__ int snap(float w)
{
    float s = floorf(fmaf(w, 256.0f, 0.5f));
    if (!(s == s)) s = 0.0f;
    s = fminf(fmaxf(s, -67108864.0f), 67108864.0f);
    return (int)s;
}

// window coordinates of one clip-space position: 1/256-px snapped x / y, depth in [0, 1], 1 / w
__device__ __forceinline__ void screen_vertex(const float* c, float hw, float hh, int& X, int& Y, float& z, float& invw)
{
    const float xn = c[0] / c[3], yn = c[1] / c[3], zn = c[2] / c[3];
    X = snap(fmaf(xn, hw, hw));
    Y = snap(fmaf(yn, hh, hh));
    z = fmaf(zn, 0.5f, 0.5f);
    invw = 1.0f / c[3];
}

// X / Y / z / invw are filled in: signed area, ownership of the edges, pixel box.  Returns false if degenerate or outside
// the W x H target.
__device__ __forceinline__ bool setup_finish(int W, int H, Setup& t)
{
    long long area2 = (long long)(t.X[1] - t.X[0]) * (long long)(t.Y[2] - t.Y[0]) -
                      (long long)(t.Y[1] - t.Y[0]) * (long long)(t.X[2] - t.X[0]);
    if (area2 == 0) return false;
    t.flipped = area2 < 0;
    t.area2 = area2 < 0 ? -area2 : area2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int a = (i + 1) % 3, b = (i + 2) % 3;
        int dx = t.X[b] - t.X[a], dy = t.Y[b] - t.Y[a];
        if (t.flipped) { dx = -dx; dy = -dy; }
        const bool owned = (dy < 0) || (dy == 0 && dx < 0);
        t.bias[i] = owned ? 0 : -1;
    }
    int xmn = min(t.X[0], min(t.X[1], t.X[2])), xmx = max(t.X[0], max(t.X[1], t.X[2]));
    int ymn = min(t.Y[0], min(t.Y[1], t.Y[2])), ymx = max(t.Y[0], max(t.Y[1], t.Y[2]));
    int x0 = (xmn - 128 + 255) >> 8, x1 = (xmx - 128) >> 8;
    int y0 = (ymn - 128 + 255) >> 8, y1 = (ymx - 128) >> 8;
    x0 = max(x0, 0); y0 = max(y0, 0);
    x1 = min(x1, W - 1); y1 = min(y1, H - 1);
    if (x0 > x1 || y0 > y1) return false;
    t.xmin = x0; t.xmax = x1; t.ymin = y0; t.ymax = y1;
    return true;
}

// c0..c2: clip-space positions.  Returns false if degenerate or outside the W x H target.
__device__ __forceinline__ bool setup_tri(const float* c0, const float* c1, const float* c2, int W,
                                          int H, Setup& t)
{
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float* cs[3] = {c0, c1, c2};
#pragma unroll
    for (int i = 0; i < 3; ++i) screen_vertex(cs[i], hw, hh, t.X[i], t.Y[i], t.z[i], t.invw[i]);
    return setup_finish(W, H, t);
}

// The vertex pass (k_vertex_xform) stores screen_vertex() of every vertex in front of the near plane as one float4 (X and Y
// as integer bit patterns); a vertex behind the plane gets kScreenClipped in X -- snap() never returns it -- and its triangles
// take the clip-space path.  A triangle whose three corners are stored is set up without a single division.
constexpr int kScreenClipped = (int)0x80000000;
__device__ __forceinline__ bool setup_from_screen(const uint4& s0, const uint4& s1, const uint4& s2, int W, int H, Setup& t)
{
    t.X[0] = (int)s0.x; t.Y[0] = (int)s0.y; t.z[0] = __uint_as_float(s0.z); t.invw[0] = __uint_as_float(s0.w);
    t.X[1] = (int)s1.x; t.Y[1] = (int)s1.y; t.z[1] = __uint_as_float(s1.z); t.invw[1] = __uint_as_float(s1.w);
    t.X[2] = (int)s2.x; t.Y[2] = (int)s2.y; t.z[2] = __uint_as_float(s2.z); t.invw[2] = __uint_as_float(s2.w);
    return setup_finish(W, H, t);
}
__device__ __forceinline__ bool screen_all_inside(const uint4& s0, const uint4& s1, const uint4& s2)
{
    return (int)s0.x != kScreenClipped && (int)s1.x != kScreenClipped && (int)s2.x != kScreenClipped;
}

// coverage + screen-space barycentrics at pixel (px,py) (R4, R5)
__device__ __forceinline__ bool coverage(const Setup& t, int px, int py, float* lambda)
{
    // every factor fits 32 bits (snap() keeps |X|, |Y| <= 2^26, a pixel centre is below 2^23): each product is ONE
    // 32 x 32 -> 64-bit multiply-add instead of a 64 x 64 product
    const int cx = 256 * px + 128, cy = 256 * py + 128;
    long long E[3];
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int a = (i + 1) % 3, b = (i + 2) % 3;
        long long e = (long long)(t.X[b] - t.X[a]) * (long long)(cy - t.Y[a]) -
                      (long long)(t.Y[b] - t.Y[a]) * (long long)(cx - t.X[a]);
        if (t.flipped) e = -e;
        inside = inside && (e + t.bias[i] >= 0);
        E[i] = e;
    }
    if (!inside) return false;
    const float fa = (float)t.area2;
    lambda[0] = (float)E[0] / fa;
    lambda[1] = (float)E[1] / fa;
    lambda[2] = (float)E[2] / fa;
    return true;
}

__device__ __forceinline__ unsigned depth24(float z)
{
    float fd = floorf(fmaf(z, 16777215.0f, 0.5f));
    unsigned d = (unsigned)fd;
    return d > 0xFFFFFFu ? 0xFFFFFFu : d;
}

// ---------------------------------------------------------------------------------------------
// near-plane clipping (R7)
// ---------------------------------------------------------------------------------------------
struct ClipVert {
    float clip[4];
    float bary[3];
};

// Sutherland-Hodgman against the near plane.  The output polygon is written to FIXED slots (a switch
// over the six partially-inside cases) -- `out[n++] = ...` with a run-time n would push the polygon into
// scratch memory, and that scratch traffic reaches HBM.  Vertex order and arithmetic are those of the
// sequential form: for i = 0..2 { emit in[i] if inside; emit the edge (i, i+1) crossing if its ends
// differ }, every crossing interpolated from the inside vertex towards the outside one.
__device__ __forceinline__ ClipVert clip_cross(const ClipVert& a, const ClipVert& b, float da, float db)
{
    const float tt = da / (da - db);
    ClipVert o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.clip[k] = fmaf(tt, b.clip[k] - a.clip[k], a.clip[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) o.bary[k] = fmaf(tt, b.bary[k] - a.bary[k], a.bary[k]);
    return o;
}

__device__ __forceinline__ int clip_near(const ClipVert* in, ClipVert* out)
{
    const float d0 = in[0].clip[2] + in[0].clip[3], d1 = in[1].clip[2] + in[1].clip[3], d2 = in[2].clip[2] + in[2].clip[3];
    const bool i0 = in[0].clip[2] >= -in[0].clip[3], i1 = in[1].clip[2] >= -in[1].clip[3], i2 = in[2].clip[2] >= -in[2].clip[3];
    const int m = (i0 ? 1 : 0) | (i1 ? 2 : 0) | (i2 ? 4 : 0);
    if (m == 0) return 0;
    if (m == 7) {
        out[0] = in[0]; out[1] = in[1]; out[2] = in[2];
        return 3;
    }
    switch (m) {
    case 1:   // only v0 inside: v0, x01, x20
        out[0] = in[0]; out[1] = clip_cross(in[0], in[1], d0, d1); out[2] = clip_cross(in[0], in[2], d0, d2);
        return 3;
    case 2:   // only v1: x01, v1, x12
        out[0] = clip_cross(in[1], in[0], d1, d0); out[1] = in[1]; out[2] = clip_cross(in[1], in[2], d1, d2);
        return 3;
    case 4:   // only v2: x12, v2, x20
        out[0] = clip_cross(in[2], in[1], d2, d1); out[1] = in[2]; out[2] = clip_cross(in[2], in[0], d2, d0);
        return 3;
    case 3:   // v0, v1 inside: v0, v1, x12, x20
        out[0] = in[0]; out[1] = in[1]; out[2] = clip_cross(in[1], in[2], d1, d2); out[3] = clip_cross(in[0], in[2], d0, d2);
        return 4;
    case 5:   // v0, v2 inside: v0, x01, x12, v2
        out[0] = in[0]; out[1] = clip_cross(in[0], in[1], d0, d1); out[2] = clip_cross(in[2], in[1], d2, d1); out[3] = in[2];
        return 4;
    default:  // 6: v1, v2 inside: x01, v1, v2, x20
        out[0] = clip_cross(in[1], in[0], d1, d0); out[1] = in[1]; out[2] = in[2]; out[3] = clip_cross(in[2], in[0], d2, d0);
        return 4;
    }
}

// ---------------------------------------------------------------------------------------------
// vertex stage (render_shader.vert:57-95)
// ---------------------------------------------------------------------------------------------
struct VsOut {
    float objc[4];
    float world[3];
    float cam[3];
    float nrm[3];
    float uv[2];
    float clip[4];
};

// position-only part: clip position and camera z
__device__ __forceinline__ void vertex_clip(const slhip_mesh_pool& pool, const slhip_scene* __restrict__ sc,
                                            const slhip_draw* __restrict__ dr, unsigned v, float* clip,
                                            float& camz)
{
    const float4 p = reinterpret_cast<const float4*>(pool.d_pos)[v];
    const float pos[4] = {p.x, p.y, p.z, 1.0f};
    float obj4[4], world4[4], cam4[4];
    mv4(dr->mesh_to_object, pos, obj4);
    mv4(dr->object_to_world, obj4, world4);
    mv4(sc->world_to_cam, world4, cam4);
    camz = cam4[2] / cam4[3];
    mv4(sc->proj, cam4, clip);
}

__device__ __forceinline__ void vertex_full(const slhip_mesh_pool& pool, const slhip_scene* __restrict__ sc,
                                            const slhip_draw* __restrict__ dr, unsigned v, VsOut& o)
{
    const float4 p = reinterpret_cast<const float4*>(pool.d_pos)[v];
    const float pos[4] = {p.x, p.y, p.z, 1.0f};
    float obj4[4], world4[4], cam4[4];
    // x / 1.0f is x, bit for bit: with affine matrices (the rule) the homogeneous coordinate is exactly 1 and the nine IEEE
    // divisions (eleven instructions each) are jumped over by the whole wave
    mv4(dr->mesh_to_object, pos, obj4);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.objc[i] = obj4[i];
    if (obj4[3] != 1.0f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) o.objc[i] = obj4[i] / obj4[3];
    }
    mv4(dr->object_to_world, obj4, world4);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.world[i] = world4[i];
    if (world4[3] != 1.0f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) o.world[i] = world4[i] / world4[3];
    }
    mv4(sc->world_to_cam, world4, cam4);
#pragma unroll
    for (int i = 0; i < 3; ++i) o.cam[i] = cam4[i];
    if (cam4[3] != 1.0f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) o.cam[i] = cam4[i] / cam4[3];
    }
    o.objc[3] = o.cam[2];
    const float4 n4 = reinterpret_cast<const float4*>(pool.d_nrm)[v];
    const float n[3] = {n4.x, n4.y, n4.z};
    mv3p(dr->normal_to_world, n, o.nrm);
    normalize3(o.nrm);
    const float2 uv = reinterpret_cast<const float2*>(pool.d_uv)[v];
    o.uv[0] = uv.x;
    o.uv[1] = uv.y;
}

// ---------------------------------------------------------------------------------------------
// k_vertex_xform: batched 4x4 vertex transforms on the matrix cores.
//   D[16x16] = A[16x4] * B[4x16] with v_mfma_f32_16x16x4_f32 (exact fp32, bitwise a k-ordered
//   fmaf chain):  A rows 0-3 = MVP = P (W2C (O2W M2O)), rows 4-7 / 8-11 / 12-15 = the light
//   clip matrices S_l (O2W M2O) of the three lights; B column j = (x,y,z,1) of vertex j.
//   One instruction transforms 16 vertices into camera clip space AND the three shadow clip
//   spaces.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; it receives D[(l>>4)*4+r][l&15],
//   i.e. lanes 0-15 hold the camera clip position of their vertex, lanes 16-31 light 0, ...
//   and every lane parks its float4 in LDS for the per-vertex half of the kernel (below).
// ---------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));

// a light with zero colour or zero direction casts no shadow and adds no radiance
__device__ __forceinline__ bool light_active(const slhip_scene* sc, int l)
{
    const float* lc = sc->light_color[l];
    const float* ld = sc->light_dir[l];
    return !((lc[0] == 0.0f && lc[1] == 0.0f && lc[2] == 0.0f) ||
             (ld[0] == 0.0f && ld[1] == 0.0f && ld[2] == 0.0f));
}

// ... and the REST of the vertex stage in the same kernel, once per vertex (thread = vertex) -- a post-transform vertex cache.
// The wave's four MFMA results go through LDS (lane l of batch b holds plane l >> 4 of vertex 16 b + (l & 15); thread j wants
// all planes of vertex j), so the clip positions never make the round trip through HBM that a separate attribute kernel paid
// (64 B per vertex and pass saved of ~160: the pass is bound by its bytes).  Per vertex:
//   * clip plane 0: the camera clip position (the near-plane clipper of the raster / shading passes reads it for the rare
//     triangles that cross the plane);
//   * clip planes 1..3: the WINDOW coordinates screen_vertex() of the light clip positions in the S x S map (the shadow pass
//     clips nothing, so its set-up never needs the clip position itself);
//   * one 64-byte record for the shading pass: (object xyz, camera z), (world xyz, camera x), (world normal, camera y) =
//     vertex_full(), and the window coordinates of the camera clip position (kScreenClipped behind the near plane);
//   * the camera window coordinates once more as a dense plane behind the records, for the raster pass.
// The shading pass then fetches one cache line per corner instead of re-running three matrix products, nine divisions and a
// normalisation per corner of every PIXEL and twelve more divisions per triangle set-up (160 k vertices per C2 scene against
// 3 x 307 k pixel corners).  Same functions, same inputs, same bits as the per-pixel evaluation they replace.
__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_vertex_xform(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                                      const slhip_draw* __restrict__ draws, float4* __restrict__ clip,
                                                      unsigned n_clip_verts, float4* __restrict__ vattr, int W, int H,
                                                      int with_lights, int S)
{
    __shared__ float4 s_clip[4][4][64];   // [wave][MFMA batch][lane]
    const slhip_draw* dr = draws + blockIdx.x;
    const unsigned nv = dr->n_verts;
    if (blockIdx.y * 256 >= nv) return;
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const slhip_scene* sc = scenes + dr->scene;
    // composite matrices (wave-uniform inputs; every lane evaluates the same chains)
    float T1[16], T2[16], M[16];
    mm4(dr->object_to_world, dr->mesh_to_object, T1);
    const unsigned row = lane & 15, k = lane >> 4;
    const unsigned grp = row >> 2;        // 0: camera, 1..3: light grp-1
    float a = 0.0f;
    if (grp == 0) {
        mm4(sc->world_to_cam, T1, T2);
        mm4(sc->proj, T2, M);
        a = M[4 * (row & 3) + k];
    } else if (with_lights) {
        mm4(sc->shadow_mat[grp - 1], T1, M);
        a = M[4 * (row & 3) + k];
    }
    const float* pos = pool.d_pos + 4 * (size_t)dr->vtx_base;
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H, hs = 0.5f * (float)S;
    // 256 vertices per block and pass (64 per wave = 4 MFMA batches); grid-stride over the draw's vertices, block-uniform trip count
    for (unsigned base = blockIdx.y * 256; base < nv; base += gridDim.y * 256) {
        const unsigned first = base + wave * 64;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned v0 = first + 16 * b;
            const unsigned vj = min(v0 + (lane & 15), nv - 1);
            // B[k][j]: component k of vertex j (w = 1): the wave reads 16 x 16 contiguous bytes
            const float bval = k < 3 ? pos[4 * (size_t)vj + k] : 1.0f;
            floatx4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bval, acc, 0, 0, 0);
            // lane holds rows (lane>>4)*4 + 0..3 of column lane&15 = the float4 of plane (lane>>4)
            s_clip[wave][b][lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        __syncthreads();
        const unsigned v = first + lane;
        if (v < nv) {
            const float4* mine = &s_clip[wave][lane >> 4][lane & 15];   // + 16 * plane
            if (with_lights)
                for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l) {
                    if (!light_active(sc, l)) continue;
                    const float4 l4 = mine[16 * (1 + l)];
                    const float lc[4] = {l4.x, l4.y, l4.z, l4.w};
                    int X, Y;
                    float z, invw;
                    screen_vertex(lc, hs, hs, X, Y, z, invw);
                    reinterpret_cast<uint4*>(clip)[(size_t)(1 + l) * n_clip_verts + dr->clip_base + v] =
                        make_uint4((unsigned)X, (unsigned)Y, __float_as_uint(z), __float_as_uint(invw));
                }
            VsOut o;
            vertex_full(pool, sc, dr, dr->vtx_base + v, o);
            const float4 c4 = mine[0];
            clip[dr->clip_base + v] = c4;
            const float c[4] = {c4.x, c4.y, c4.z, c4.w};
            int X = kScreenClipped, Y = 0;
            float z = 0.0f, invw = 0.0f;
            if (c[2] >= -c[3]) screen_vertex(c, hw, hh, X, Y, z, invw);   // the inside test of clip_near()
            // one 64-byte record per vertex for the shading pass (one cache line per pixel corner) ...
            float4* dst = vattr + 4 * (size_t)(dr->clip_base + v);
            const uint4 scr = make_uint4((unsigned)X, (unsigned)Y, __float_as_uint(z), __float_as_uint(invw));
            dst[0] = make_float4(o.objc[0], o.objc[1], o.objc[2], o.cam[2]);
            dst[1] = make_float4(o.world[0], o.world[1], o.world[2], o.cam[0]);
            dst[2] = make_float4(o.nrm[0], o.nrm[1], o.nrm[2], o.cam[1]);
            reinterpret_cast<uint4*>(dst)[3] = scr;
            // ... and the window coordinates once more as a dense plane for the raster pass (16 B per vertex, like the clip plane)
            reinterpret_cast<uint4*>(vattr)[4 * (size_t)n_clip_verts + dr->clip_base + v] = scr;
        }
        __syncthreads();   // the LDS batch is free for the next pass
    }
}

// ---------------------------------------------------------------------------------------------
// texture fetch: bilinear, mip 0, repeat
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrapi(int i, int n)
{
    if ((n & (n - 1)) == 0) return i & (n - 1);   // power of two: the low bits of a two's-complement i ARE the non-negative remainder
    int m = i % n;                                // (an integer division is ~40 instructions on this machine)
    return m < 0 ? m + n : m;
}

// ---- 2D textures with mip chains and the asset's sampler state (include/slhip.h, SLHIP_SAMPLER_*) ----
__device__ __forceinline__ int wrap_coord(int i, int n, unsigned mode)
{
    if (mode == 1u) return min(max(i, 0), n - 1);             // clamp to edge
    if (mode == 2u) {                                         // mirrored repeat
        const int m = wrapi(i, 2 * n);
        return m < n ? m : 2 * n - 1 - m;
    }
    return wrapi(i, n);                                       // repeat
}

// Textures are addressed as (texel pool base, 32-bit byte offset): the pool pointer is a kernel argument, so a fetch needs ONE
// address register per texel (scalar base + vector offset) instead of a 64-bit pointer pair, and one integer add instead of
// an add-with-carry chain -- eight texels of a trilinear fetch are in flight at once.  (The texel pool is < 4 GB: n_tex_bytes is
// a uint32_t.)
__device__ __forceinline__ void texel_rgba(const uint8_t* __restrict__ pool_tex, unsigned lvl, int w, int x, int y, float* out)
{
    const uchar4 t = *reinterpret_cast<const uchar4*>(pool_tex + (lvl + 4u * ((unsigned)y * (unsigned)w + (unsigned)x)));
    out[0] = slhip::unorm8(t.x); out[1] = slhip::unorm8(t.y); out[2] = slhip::unorm8(t.z); out[3] = slhip::unorm8(t.w);
}

// one level, nearest or bilinear
__device__ __forceinline__ void tex_level(const uint8_t* __restrict__ pool_tex, unsigned lvl, int w, int h, unsigned sampler,
                                          bool linear, float u, float v, float* out)
{
    const unsigned ws = SLHIP_SAMPLER_WRAP_S(sampler), wt = SLHIP_SAMPLER_WRAP_T(sampler);
    if (!linear) {
        texel_rgba(pool_tex, lvl, w, wrap_coord((int)floorf(u * (float)w), w, ws), wrap_coord((int)floorf(v * (float)h), h, wt), out);
        return;
    }
    const float x = fmaf(u, (float)w, -0.5f), y = fmaf(v, (float)h, -0.5f);
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    const int x0 = wrap_coord((int)fx, w, ws), y0 = wrap_coord((int)fy, h, wt);
    const int x1 = wrap_coord((int)fx + 1, w, ws), y1 = wrap_coord((int)fy + 1, h, wt);
    float c00[4], c10[4], c01[4], c11[4];
    texel_rgba(pool_tex, lvl, w, x0, y0, c00); texel_rgba(pool_tex, lvl, w, x1, y0, c10);
    texel_rgba(pool_tex, lvl, w, x0, y1, c01); texel_rgba(pool_tex, lvl, w, x1, y1, c11);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float top = fmaf(ax, c10[c] - c00[c], c00[c]), bot = fmaf(ax, c11[c] - c01[c], c01[c]);
        out[c] = fmaf(ay, bot - top, top);
    }
}

// level l of a texture whose level 0 is w0 x h0 at byte offset `tex` of the pool: offset and size
__device__ __forceinline__ unsigned tex_level_off(unsigned tex, int w0, int h0, int l, int& w, int& h)
{
    unsigned off = tex;
    w = w0; h = h0;
    for (int k = 0; k < l; ++k) {
        off += 4u * (unsigned)w * (unsigned)h;
        w = max(1, w >> 1); h = max(1, h >> 1);
    }
    return off;
}

// log2 of a positive normal float from exponent + a degree-6 polynomial in fmaf form (max error 1.4e-6): the
// SAME operations run in the oracle, so the level of detail -- and everything sampled with it, normal maps
// included -- is bit-identical on both sides (libm's and the device's log2f differ in the last bits)
__device__ __forceinline__ float det_log2(float x)
{
    const unsigned bits = __float_as_uint(x);
    const int e = (int)(bits >> 23) - 127;
    const float t = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u) - 1.0f;
    float q = 0.02049034833908081f;
    q = fmaf(q, t, -0.09606625884771347f);
    q = fmaf(q, t, 0.2155885398387909f);
    q = fmaf(q, t, -0.33924779295921326f);
    q = fmaf(q, t, 0.4777059257030487f);
    q = fmaf(q, t, -0.721162736415863f);
    q = fmaf(q, t, 1.4426932334899902f);
    return fmaf(q, t, (float)e);
}

// texture2D() of the fragment shader: (du, dv) to the +x and +y pixel neighbours select the level
__device__ __forceinline__ void tex_sample(const uint8_t* __restrict__ pool_tex, unsigned tex, int w, int h, unsigned sampler, float u,
                                           float v, float dudx, float dvdx, float dudy, float dvdy, float* out)
{
    const unsigned mip = SLHIP_SAMPLER_MIP(sampler);
    const float ax = dudx * (float)w, bx = dvdx * (float)h, ay = dudy * (float)w, by = dvdy * (float)h;
    const float rx = sqrtf(fmaf(bx, bx, ax * ax)), ry = sqrtf(fmaf(by, by, ay * ay));
    const float rho = fmaxf(rx, ry);
    const bool magnify = !(rho > 1.0f);            // lambda <= 0 (or a degenerate footprint)
    if (magnify || mip == 0u) {
        tex_level(pool_tex, tex, w, h, sampler, (sampler & (magnify ? SLHIP_SAMPLER_MAG_LINEAR : SLHIP_SAMPLER_MIN_LINEAR)) != 0u, u, v, out);
        return;
    }
    const int top = 31 - __clz(max(max(w, h), 1));  // last level = floor(log2(max(w, h)))
    const float lambda = fminf(det_log2(rho), (float)top);
    const bool lin = (sampler & SLHIP_SAMPLER_MIN_LINEAR) != 0u;
    int lw, lh;
    if (mip == 1u) {                                // nearest level: round half up (section 8.14.3)
        const int l = min((int)ceilf(lambda + 0.5f) - 1, top);
        const unsigned p = tex_level_off(tex, w, h, max(l, 0), lw, lh);
        tex_level(pool_tex, p, lw, lh, sampler, lin, u, v, out);
        return;
    }
    const int l0 = min((int)floorf(lambda), top), l1 = min(l0 + 1, top);
    const float f = lambda - (float)l0;
    float a[4], b[4];
    const unsigned p0 = tex_level_off(tex, w, h, l0, lw, lh);
    tex_level(pool_tex, p0, lw, lh, sampler, lin, u, v, a);
    if (l1 == l0 || f == 0.0f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = a[c];
        return;
    }
    const unsigned p1 = p0 + 4u * (unsigned)lw * (unsigned)lh;   // l1 == l0 + 1: the next level follows level l0
    lw = max(1, lw >> 1); lh = max(1, lh >> 1);
    tex_level(pool_tex, p1, lw, lh, sampler, lin, u, v, b);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = fmaf(f, b[c] - a[c], a[c]);
}

// perspective-correct barycentrics (w.r.t. the ORIGINAL triangle) of pixel (px, py) on the plane of the
// set-up sub-triangle -- also outside its edges (used for the texture footprint)
__device__ __forceinline__ void bary_at(const Setup& t, const float* b0, const float* b1, const float* b2, int px, int py, float* b)
{
    const int cx = 256 * px + 128, cy = 256 * py + 128;   // 32-bit factors, see coverage()
    float l[3];
    const float fa = (float)t.area2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int a = (i + 1) % 3, c = (i + 2) % 3;
        long long e = (long long)(t.X[c] - t.X[a]) * (long long)(cy - t.Y[a]) - (long long)(t.Y[c] - t.Y[a]) * (long long)(cx - t.X[a]);
        if (t.flipped) e = -e;
        l[i] = (float)e / fa;
    }
    const float pw0 = l[0] * t.invw[0], pw1 = l[1] * t.invw[1], pw2 = l[2] * t.invw[2];
    const float sw = (pw0 + pw1) + pw2;
    const float bs[3] = {pw0 * (1.0f / sw), pw1 * (1.0f / sw), pw2 * (1.0f / sw)};
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = fmaf(bs[2], b2[k], fmaf(bs[1], b1[k], bs[0] * b0[k]));
}

// rectangle texture (the sticker, render_shader.frag:254): unnormalised texel coordinates, LINEAR, clamp to
// edge.  The image is stored top row first while GL texel row 0 is the BOTTOM row of an imported image.
__device__ __forceinline__ void tex_rect_bilinear(const uint8_t* __restrict__ tex, int w, int h, float xt, float yt, float* out)
{
    const float x = xt - 0.5f, y = yt - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    const int x0 = min(max((int)fx, 0), w - 1), x1 = min(max((int)fx + 1, 0), w - 1);
    const int y0 = (h - 1) - min(max((int)fy, 0), h - 1), y1 = (h - 1) - min(max((int)fy + 1, 0), h - 1);
    const uchar4 t00 = reinterpret_cast<const uchar4*>(tex)[(size_t)y0 * w + x0];
    const uchar4 t10 = reinterpret_cast<const uchar4*>(tex)[(size_t)y0 * w + x1];
    const uchar4 t01 = reinterpret_cast<const uchar4*>(tex)[(size_t)y1 * w + x0];
    const uchar4 t11 = reinterpret_cast<const uchar4*>(tex)[(size_t)y1 * w + x1];
    const unsigned char c00[4] = {t00.x, t00.y, t00.z, t00.w}, c10[4] = {t10.x, t10.y, t10.z, t10.w};
    const unsigned char c01[4] = {t01.x, t01.y, t01.z, t01.w}, c11[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = slhip::unorm8(c00[c]), b = slhip::unorm8(c10[c]);
        const float cc = slhip::unorm8(c01[c]), d = slhip::unorm8(c11[c]);
        const float top = fmaf(ax, b - a, a), bot = fmaf(ax, d - cc, cc);
        out[c] = fmaf(ay, bot - top, top);
    }
}

// ---------------------------------------------------------------------------------------------
// fragment emission
// ---------------------------------------------------------------------------------------------
struct MainTarget {
    unsigned long long* vis;  // scene's [H,W] keys
    const float* peel;        // scene's [H,W,4] previous objectCoordinates or nullptr
    int W;
    unsigned prim;
    // for the (rare) discard tests that run before the depth write
    bool need_attr;
    const float* camz;   // [3] camera z of the ORIGINAL vertices
    const float* bary;   // [3][3] barycentrics of the sub-triangle vertices w.r.t. the original
    const float* uv;     // [3][2]
    const uint8_t* pool_tex;   // texel pool, nullptr = no alpha test
    unsigned tex;              // byte offset of the base-colour texture in the pool
    int tex_w, tex_h;
    unsigned tex_sampler;
    float base_alpha, alpha_cutoff;

    __device__ __forceinline__ void emit(const Setup& t, int px, int py, const float* l) const
    {
        const float z = interp(l, t.z[0], t.z[1], t.z[2]);
        if (!(z >= 0.0f && z <= 1.0f)) return;
        const unsigned d24 = depth24(z);
        if (d24 >= 0xFFFFFFu) return;   // GL_LESS against the cleared depth 1.0: a fragment AT the far plane loses (R6)
        const unsigned long long key = ((unsigned long long)d24 << 32) | prim;
        unsigned long long* slot = vis + (size_t)py * W + px;
        // monotone pre-test only where the (rare, expensive) discard tests follow: keys only ever
        // decrease, so a stale read is conservative.  The plain path fires the atomic without
        // waiting on a read (a memory round trip per fragment costs more than a lost atomic).
        if (need_attr && __builtin_nontemporal_load(slot) <= key) return;
        if (need_attr) {
            const float pw0 = l[0] * t.invw[0], pw1 = l[1] * t.invw[1], pw2 = l[2] * t.invw[2];
            const float sw = (pw0 + pw1) + pw2;
            const float bs[3] = {pw0 * (1.0f / sw), pw1 * (1.0f / sw), pw2 * (1.0f / sw)};
            float b[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                b[k] = fmaf(bs[2], bary[6 + k], fmaf(bs[1], bary[3 + k], bs[0] * bary[k]));
            if (peel) {
                const float cz = interp(b, camz[0], camz[1], camz[2]);
                if (cz - 0.00001f <= peel[4 * ((size_t)py * W + px) + 3]) return;
            }
            if (pool_tex) {
                const float u = interp(b, uv[0], uv[2], uv[4]);
                const float v = interp(b, uv[1], uv[3], uv[5]);
                float bx[3], by[3];
                bary_at(t, bary, bary + 3, bary + 6, px + 1, py, bx);
                bary_at(t, bary, bary + 3, bary + 6, px, py + 1, by);
                float tc[4];
                tex_sample(pool_tex, tex, tex_w, tex_h, tex_sampler, u, v, interp(bx, uv[0], uv[2], uv[4]) - u, interp(bx, uv[1], uv[3], uv[5]) - v,
                           interp(by, uv[0], uv[2], uv[4]) - u, interp(by, uv[1], uv[3], uv[5]) - v, tc);
                if (base_alpha * tc[3] < alpha_cutoff) return;
            }
        }
        atomicMin(slot, key);
    }
};

#ifndef SLHIP_SHADOW_WINDOW
#define SLHIP_SHADOW_WINDOW 64     // texels per side of the LDS window of k_shadow_raster (0: every fragment is a global atomic)
#endif
#ifndef SLHIP_SHADOW_WINDOW_PITCH
#define SLHIP_SHADOW_WINDOW_PITCH SLHIP_SHADOW_WINDOW         // words per window row.  Round 6 measured a pitch of 65 (the texels of a column in different LDS banks; the counters' conflict share of this kernel is 0.31): 11.0 -> 11.8 ms per 1024 scenes -- the conflicts are lanes meeting in the SAME texel (neighbouring triangles), which no pitch separates, and the flush pays the odd pitch's index arithmetic
#endif
struct ShadowTarget {
    unsigned* sm;  // [S,S] float bits
    int W;
    // per-chunk depth resolve in LDS (the north star's "LDS per-tile bins"): fragments inside the block's window take an LDS
    // atomic; the window is written to the map once, row by row (k_shadow_raster).  win == nullptr: no window (k_shadow_large)
    unsigned* win;
    int wx0, wy0;
    __device__ __forceinline__ void emit(const Setup& t, int px, int py, const float* l) const
    {
        const float z = interp(l, t.z[0], t.z[1], t.z[2]);
        if (!(z >= 0.0f && z <= 1.0f)) return;
        const unsigned bits = __float_as_uint(z);  // z >= 0: uint order == float order
#if SLHIP_SHADOW_WINDOW
        const unsigned ux = (unsigned)(px - wx0), uy = (unsigned)(py - wy0);
        if (win != nullptr && ux < (unsigned)SLHIP_SHADOW_WINDOW && uy < (unsigned)SLHIP_SHADOW_WINDOW) {
            atomicMin(win + uy * SLHIP_SHADOW_WINDOW_PITCH + ux, bits);
            return;
        }
#endif
        unsigned* slot = sm + (size_t)py * W + px;
        // fire-and-forget: front faces are culled, so almost every fragment wins anyway; a
        // pre-read would only serialise the loop on a memory round trip
        atomicMin(slot, bits);
    }
};

// work item of the large-triangle queue
struct QItem {
    unsigned draw;      // global draw index
    unsigned tri_sub;   // triangle | sub-triangle << 31
    unsigned scene_aux; // scene | light << 24
    unsigned tile;      // tx | ty << 16   (8x8 pixel tiles)
};

// All pixels of the triangle's bounding box by one thread.  The edge functions are exact integers,
// so walking them incrementally (E(px+1) = E(px) - 256 dY, E(py+1) = E(py) + 256 dX) gives the very
// values coverage() computes from scratch -- at three 64-bit adds per pixel instead of six 64-bit
// multiplies.  The top-left bias is folded in: a pixel is covered iff all three biased values are
// >= 0, i.e. iff the sign bit of their OR is clear.
template <class Target>
__device__ __forceinline__ void raster_bbox(const Setup& t, const Target& tgt)
{
    const int cx = 256 * t.xmin + 128, cy = 256 * t.ymin + 128;   // 32-bit factors, see coverage()
    long long row[3], sx[3], sy[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int a = (i + 1) % 3, b = (i + 2) % 3;
        const int Ai = t.X[b] - t.X[a], Bi = t.Y[b] - t.Y[a];
        long long e = (long long)Ai * (long long)(cy - t.Y[a]) - (long long)Bi * (long long)(cx - t.X[a]);
        long long dx = -256ll * (long long)Bi, dy = 256ll * (long long)Ai;
        if (t.flipped) { e = -e; dx = -dx; dy = -dy; }
        row[i] = e + t.bias[i]; sx[i] = dx; sy[i] = dy;
    }
    const float fa = (float)t.area2;
    // ONE flat loop over the box: the lanes of a wave hold boxes of different shapes, and a nested
    // loop would run (max height) x (max width) iterations instead of (max area)
    const int bw = t.xmax - t.xmin + 1, n = bw * (t.ymax - t.ymin + 1);
    long long e0 = row[0], e1 = row[1], e2 = row[2];
    int px = t.xmin, py = t.ymin;
    for (int k = 0; k < n; ++k) {
        if ((e0 | e1 | e2) >= 0) {
            float l[3];
            l[0] = (float)(e0 - t.bias[0]) / fa;
            l[1] = (float)(e1 - t.bias[1]) / fa;
            l[2] = (float)(e2 - t.bias[2]) / fa;
            tgt.emit(t, px, py, l);
        }
        if (px == t.xmax) {
            px = t.xmin; ++py;
            row[0] += sy[0]; row[1] += sy[1]; row[2] += sy[2];
            e0 = row[0]; e1 = row[1]; e2 = row[2];
        } else {
            ++px;
            e0 += sx[0]; e1 += sx[1]; e2 += sx[2];
        }
    }
}

// Rasterise one sub-triangle: small ones in place, larger ones to the queue (or in place if
// the queue is full).
template <class Target>
__device__ __forceinline__ void raster_or_enqueue(const Setup& t, const Target& tgt, unsigned* queue,
                                                  unsigned capacity, unsigned draw, unsigned tri_sub,
                                                  unsigned scene_aux, int small_area)
{
    const int bw = t.xmax - t.xmin + 1, bh = t.ymax - t.ymin + 1;
    bool in_place = bw * bh <= small_area;
    if (!in_place) {
        const int tx0 = t.xmin >> 3, tx1 = t.xmax >> 3, ty0 = t.ymin >> 3, ty1 = t.ymax >> 3;
        const unsigned n = (unsigned)((tx1 - tx0 + 1) * (ty1 - ty0 + 1));
        const unsigned base = atomicAdd(queue, n);
        if (base + n <= capacity) {
            QItem* items = reinterpret_cast<QItem*>(queue + 4);
            unsigned k = base;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {
                    QItem it;
                    it.draw = draw; it.tri_sub = tri_sub; it.scene_aux = scene_aux;
                    it.tile = (unsigned)tx | ((unsigned)ty << 16);
                    items[k++] = it;
                }
        } else {
            // queue full: fall back to the (slow) in-place loop; mark the reservation as void
            // by writing empty items where it overlaps the queue
            QItem* items = reinterpret_cast<QItem*>(queue + 4);
            for (unsigned k = base; k < min(base + n, capacity); ++k) {
                QItem it;
                it.draw = 0xFFFFFFFFu; it.tri_sub = 0; it.scene_aux = 0; it.tile = 0;
                items[k] = it;
            }
            in_place = true;
        }
    }
    if (in_place) raster_bbox(t, tgt);
}

// ---------------------------------------------------------------------------------------------
// k_raster: main pass, one thread per triangle
// ---------------------------------------------------------------------------------------------
// Two instantiations over the same chunk list; a chunk (block) belongs to exactly one of them, the other returns at once:
//   kAttr = false: plain triangles -- depth and primitive id only.  No texture sampling, no per-vertex attributes: a fraction
//                  of the registers of the general form (the kernel's time follows its waves per SIMD), and triangles
//                  entirely in front of the near plane are set up from the window coordinates k_vertex_xform stored per
//                  vertex (`screen`), without a division;
//   kAttr = true:  chunks whose fragments may be discarded before the depth write (alpha test against the base texture,
//                  depth peeling): barycentrics, texture coordinates and camera z per fragment.
template <bool kAttr>
__device__ __forceinline__ void raster_chunk(unsigned chunk, const slhip_mesh_pool& pool, const slhip_scene* __restrict__ scenes,
                                             const slhip_draw* __restrict__ draws,
                                             const slhip_chunk* __restrict__ chunks, int W, int H,
                                             const float* __restrict__ depth_peel,
                                             unsigned long long* __restrict__ vis, unsigned* queue,
                                             unsigned capacity, const float4* __restrict__ clipbuf,
                                             const uint4* __restrict__ screen, int small_area)
{
    const slhip_chunk ch = chunks[chunk];
    if (threadIdx.x >= ch.count) return;
    const slhip_scene* sc = scenes + ch.scene;
    const slhip_draw* dr = draws + ch.draw;
    const bool alpha_test = (dr->flags & SLHIP_DRAW_ALPHA_TEST) && (dr->flags & SLHIP_DRAW_HAS_BASE_TEX);
    if ((alpha_test || depth_peel != nullptr) != kAttr) return;      // block-uniform: the chunk is the other instantiation's
    const unsigned tri = ch.first_tri + threadIdx.x;
    const unsigned* ip = pool.d_idx + dr->idx_base + 3 * (size_t)tri;
    const unsigned vi[3] = {ip[0], ip[1], ip[2]};
    const size_t P = (size_t)W * H;
    MainTarget tgt;
    tgt.vis = vis + (size_t)ch.scene * P;
    tgt.W = W;
    tgt.prim = dr->prim_base + tri;
    tgt.need_attr = kAttr;
    tgt.peel = nullptr;
    tgt.pool_tex = nullptr;

    if constexpr (!kAttr) {
        tgt.camz = nullptr; tgt.uv = nullptr; tgt.bary = nullptr;
        const uint4 s0 = screen[dr->clip_base + vi[0]], s1 = screen[dr->clip_base + vi[1]], s2 = screen[dr->clip_base + vi[2]];
        if (screen_all_inside(s0, s1, s2)) {     // clip_near() would hand the triangle through
            Setup t;
            if (setup_from_screen(s0, s1, s2, W, H, t)) raster_or_enqueue(t, tgt, queue, capacity, ch.draw, tri, ch.scene, small_area);
            return;
        }
        ClipVert cv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 c4 = clipbuf[dr->clip_base + vi[k]];   // written by k_vertex_xform (MFMA)
            cv[k].clip[0] = c4.x; cv[k].clip[1] = c4.y; cv[k].clip[2] = c4.z; cv[k].clip[3] = c4.w;
            cv[k].bary[0] = cv[k].bary[1] = cv[k].bary[2] = 0.0f;   // (not needed here)
        }
        ClipVert poly[4];
        const int n = clip_near(cv, poly);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {   // n <= 4: at most two sub-triangles; unrolled so that poly[] keeps static indices
            if (sub >= n - 2) break;
            Setup t;
            if (!setup_tri(poly[0].clip, poly[sub + 1].clip, poly[sub + 2].clip, W, H, t)) continue;
            raster_or_enqueue(t, tgt, queue, capacity, ch.draw, tri | ((unsigned)sub << 31), ch.scene, small_area);
        }
    } else {
        ClipVert cv[3];
        float camz[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 c4 = clipbuf[dr->clip_base + vi[k]];   // written by k_vertex_xform (MFMA)
            cv[k].clip[0] = c4.x; cv[k].clip[1] = c4.y; cv[k].clip[2] = c4.z; cv[k].clip[3] = c4.w;
            if (depth_peel) {  // camera z of the vertex (shader chain) for the depth-peel test
                float unused[4];
                vertex_clip(pool, sc, dr, dr->vtx_base + vi[k], unused, camz[k]);
            }
            cv[k].bary[0] = k == 0 ? 1.0f : 0.0f;
            cv[k].bary[1] = k == 1 ? 1.0f : 0.0f;
            cv[k].bary[2] = k == 2 ? 1.0f : 0.0f;
        }
        ClipVert poly[4];
        const int n = clip_near(cv, poly);
        if (n == 0) return;
        tgt.peel = depth_peel ? depth_peel + 4 * (size_t)ch.scene * P : nullptr;
        float uvs[6] = {0, 0, 0, 0, 0, 0};
        if (alpha_test) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float2 uv = reinterpret_cast<const float2*>(pool.d_uv)[dr->vtx_base + vi[k]];
                uvs[2 * k] = uv.x; uvs[2 * k + 1] = uv.y;
            }
            tgt.pool_tex = pool.d_tex; tgt.tex = dr->tex_offset;
            tgt.tex_w = (int)dr->tex_w; tgt.tex_h = (int)dr->tex_h;
            tgt.tex_sampler = dr->tex_sampler[0];
        }
        tgt.camz = camz;
        tgt.uv = uvs;
        tgt.base_alpha = dr->base_color[3];
        tgt.alpha_cutoff = dr->alpha_cutoff;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub >= n - 2) break;
            Setup t;
            if (!setup_tri(poly[0].clip, poly[sub + 1].clip, poly[sub + 2].clip, W, H, t)) continue;
            float bary[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                bary[k] = poly[0].bary[k];
                bary[3 + k] = poly[sub + 1].bary[k];
                bary[6 + k] = poly[sub + 2].bary[k];
            }
            tgt.bary = bary;
            // the discard tests need per-vertex data that only this thread holds: such triangles stay in place
            for (int py = t.ymin; py <= t.ymax; ++py)
                for (int px = t.xmin; px <= t.xmax; ++px) {
                    float l[3];
                    if (coverage(t, px, py, l)) tgt.emit(t, px, py, l);
                }
        }
    }
}

// The grid strides over the chunk list: the plain instantiation is launched with one block per chunk, the discard-testing one
// with a few thousand blocks that skim the list for the (usually few or no) chunks that are theirs -- 340 k blocks that look at
// one chunk header each and leave cost 1.1 ms per 1024 C2 scenes.
template <bool kAttr>
__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_raster(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                                const slhip_draw* __restrict__ draws,
                                                const slhip_chunk* __restrict__ chunks, unsigned n_chunks, int W, int H,
                                                const float* __restrict__ depth_peel,
                                                unsigned long long* __restrict__ vis, unsigned* queue,
                                                unsigned capacity, const float4* __restrict__ clipbuf,
                                                const uint4* __restrict__ screen, int small_area)
{
    for (unsigned c = blockIdx.x; c < n_chunks; c += gridDim.x)
        raster_chunk<kAttr>(c, pool, scenes, draws, chunks, W, H, depth_peel, vis, queue, capacity, clipbuf, screen, small_area);
}

// k_large: one wave per (triangle, 8x8 tile); lane == pixel.  Every wave takes a CONTIGUOUS run of
// queue items: the tiles of one triangle sit next to each other in the queue, so the fetch, the
// near-plane clip and the setup are done once per run of equal triangles, not once per tile.
__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_large(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                               const slhip_draw* __restrict__ draws, int W, int H,
                                               unsigned long long* __restrict__ vis,
                                               const unsigned* __restrict__ queue, unsigned capacity,
                                               const float4* __restrict__ clipbuf, const uint4* __restrict__ screen)
{
    const unsigned count = min(queue[0], capacity);
    const QItem* items = reinterpret_cast<const QItem*>(queue + 4);
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned n_waves = (gridDim.x * blockDim.x) >> 6;
    const unsigned per = (count + n_waves - 1) / n_waves;
    const unsigned i0 = wave * per, i1 = min(i0 + per, count);
    unsigned p_draw = 0xFFFFFFFFu, p_tri = 0u, p_scene = 0u, prim = 0u;
    bool ok = false;
    Setup t;
    for (unsigned i = i0; i < i1; ++i) {
        const QItem it = items[i];
        if (it.draw == 0xFFFFFFFFu) continue;
        if (it.draw != p_draw || it.tri_sub != p_tri || it.scene_aux != p_scene) {
            p_draw = it.draw; p_tri = it.tri_sub; p_scene = it.scene_aux;
            ok = false;
            const slhip_draw* dr = draws + it.draw;
            const unsigned tri = it.tri_sub & 0x7FFFFFFFu;
            const int sub = (int)(it.tri_sub >> 31);
            const unsigned* ip = pool.d_idx + dr->idx_base + 3 * (size_t)tri;
            ClipVert cv[3];
            const uint4 s0 = screen[dr->clip_base + ip[0]], s1 = screen[dr->clip_base + ip[1]], s2 = screen[dr->clip_base + ip[2]];
            const bool cached = screen_all_inside(s0, s1, s2);
            if (cached && (sub != 0 || !setup_from_screen(s0, s1, s2, W, H, t))) continue;
            if (!cached) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 c4 = clipbuf[dr->clip_base + ip[k]];
                    cv[k].clip[0] = c4.x; cv[k].clip[1] = c4.y; cv[k].clip[2] = c4.z; cv[k].clip[3] = c4.w;
                    cv[k].bary[0] = cv[k].bary[1] = cv[k].bary[2] = 0.0f;
                }
                ClipVert poly[4];
                const int n = clip_near(cv, poly);
                if (sub > n - 3) continue;
                const ClipVert& pb = sub == 0 ? poly[1] : poly[2];
                const ClipVert& pc = sub == 0 ? poly[2] : poly[3];
                if (!setup_tri(poly[0].clip, pb.clip, pc.clip, W, H, t)) continue;
            }
            prim = dr->prim_base + tri;
            ok = true;
        }
        if (!ok) continue;
        const int px = (int)((it.tile & 0xFFFFu) << 3) + (int)(lane & 7);
        const int py = (int)((it.tile >> 16) << 3) + (int)(lane >> 3);
        if (px < t.xmin || px > t.xmax || py < t.ymin || py > t.ymax) continue;
        float l[3];
        if (!coverage(t, px, py, l)) continue;
        MainTarget tgt;
        tgt.vis = vis + (size_t)it.scene_aux * W * H;
        tgt.peel = nullptr;
        tgt.W = W;
        tgt.prim = prim;
        tgt.need_attr = false;
        tgt.pool_tex = nullptr;
        tgt.emit(t, px, py, l);
    }
}

// ---------------------------------------------------------------------------------------------
// shadow pass (render_pass.cpp:408-460): depth only, front faces culled
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void shadow_clip(const slhip_mesh_pool& pool, const float* T, unsigned v, float* clip)
{
    const float4 p = reinterpret_cast<const float4*>(pool.d_pos)[v];
    const float pos[4] = {p.x, p.y, p.z, 1.0f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = fmaf(T[4 * r + 0], pos[0], 0.0f);
        a = fmaf(T[4 * r + 1], pos[1], a);
        a = fmaf(T[4 * r + 2], pos[2], a);
        a = fmaf(T[4 * r + 3], pos[3], a);
        clip[r] = a;
    }
}

// Shadow maps are kept CLEAN BETWEEN CALLS (all texels 1.0): a render marks the 64x64-texel tiles its casters touch
// (one bit per tile, per scene and light: every triangle marks the tiles of its pixel box -- which also covers the 8x8 tiles
// k_shadow_large resolves for the large ones), and after the shading pass k_shadow_restore resets exactly those tiles.  The
// per-call clear of the whole 2048^2 map (16.8 MB per scene and light, more than the whole G-buffer of the scene) is gone;
// what is written back is proportional to what was drawn.
constexpr int kShadowTile = 64;
constexpr int kShadowMaxWords = 32;   // LDS bitmap of k_shadow_raster: 1024 tiles = a 2048^2 map
__host__ __device__ inline int shadow_tiles_x(int S) { return (S + kShadowTile - 1) / kShadowTile; }
__host__ __device__ inline int shadow_tile_words(int S) { return (shadow_tiles_x(S) * shadow_tiles_x(S) + 31) / 32; }

__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_shadow_raster(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                                       const slhip_draw* __restrict__ draws,
                                                       const slhip_chunk* __restrict__ chunks, int S,
                                                       unsigned* __restrict__ shadow, unsigned* queue,
                                                       unsigned capacity, const float4* __restrict__ clipbuf,
                                                       unsigned n_clip_verts, unsigned* __restrict__ tile_bits, int nl,
                                                       int small_area)
{
    __shared__ unsigned bm[SLHIP_NUM_LIGHTS][kShadowMaxWords];   // the chunk's touched tiles, ORed into the scene's bits at the end
    const slhip_chunk ch = chunks[blockIdx.x];
    if (ch.count == 0) return;
    const slhip_scene* sc = scenes + ch.scene;
    const slhip_draw* dr = draws + ch.draw;
    if (!(dr->flags & SLHIP_DRAW_CASTS_SHADOW)) return;      // block-uniform
    const int tx_n = shadow_tiles_x(S), words = shadow_tile_words(S);
    for (int k = (int)threadIdx.x; k < SLHIP_NUM_LIGHTS * kShadowMaxWords; k += 256) bm[k / kShadowMaxWords][k % kShadowMaxWords] = 0u;
    __syncthreads();
    const bool have_tri = threadIdx.x < ch.count;
    const unsigned tri = ch.first_tri + (have_tri ? threadIdx.x : 0u);
    const unsigned* ip = pool.d_idx + dr->idx_base + 3 * (size_t)tri;
    const unsigned i0 = ip[0], i1 = ip[1], i2 = ip[2];
#if SLHIP_SHADOW_WINDOW
    // The chunk's triangles are neighbours on the mesh: most of their fragments fall into one small patch of the map.  A
    // SLHIP_SHADOW_WINDOW^2-texel window of LDS is placed at the corner of the chunk's pixel boxes; fragments inside it are
    // resolved there (LDS atomics), and the touched texels go to the map once, 64 consecutive texels per atomic instruction --
    // the same minimum per texel, whatever the order.
    constexpr int kWin = SLHIP_SHADOW_WINDOW;
    constexpr int kPitch = SLHIP_SHADOW_WINDOW_PITCH;
    __shared__ unsigned win[kWin * kPitch];
    __shared__ int worg[2];
#endif
    // one block per chunk, the (few) active lights in a loop: the index fetch is shared and no
    // workgroups are launched for lights that are off
    for (int light = 0; light < nl; ++light) {    // (lights beyond the nl maps the caller provided cast no shadow)
        if (!light_active(sc, light)) continue;   // block-uniform
        const float4* plane = clipbuf + (size_t)(1 + light) * n_clip_verts + dr->clip_base;
        Setup t;
        const uint4* sp = reinterpret_cast<const uint4*>(plane);   // k_vertex_xform stores the light's clip positions as window coordinates
        bool draw = have_tri && setup_from_screen(sp[i0], sp[i1], sp[i2], S, S, t);
        if (draw && t.flipped) draw = false;  // front face culled
#if SLHIP_SHADOW_WINDOW
        if (threadIdx.x < 2) worg[threadIdx.x] = 0x7fffffff;
        for (int k = (int)threadIdx.x; k < kWin * kPitch; k += 256) win[k] = 0x3F800000u;      // 1.0: the cleared map
        __syncthreads();
        if (draw) { atomicMin(&worg[0], t.xmin); atomicMin(&worg[1], t.ymin); }
        __syncthreads();
        const int wx0 = worg[0], wy0 = worg[1];
#endif
        if (draw) {
            // tiles of the triangle's pixel box (a 16k-triangle object: almost always one tile, at most a handful)
            if (words <= kShadowMaxWords) {
                const int tx0 = t.xmin / kShadowTile, tx1 = t.xmax / kShadowTile, ty0 = t.ymin / kShadowTile, ty1 = t.ymax / kShadowTile;
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx) {
                        const int tile = ty * tx_n + tx;
                        atomicOr(&bm[light][tile >> 5], 1u << (tile & 31));
                    }
            }
            ShadowTarget tgt;
            tgt.sm = shadow + ((size_t)ch.scene * nl + light) * S * S;
            tgt.W = S;
#if SLHIP_SHADOW_WINDOW
            tgt.win = win; tgt.wx0 = wx0; tgt.wy0 = wy0;
#else
            tgt.win = nullptr; tgt.wx0 = 0; tgt.wy0 = 0;
#endif
            raster_or_enqueue(t, tgt, queue, capacity, ch.draw, tri, ch.scene | ((unsigned)light << 24), small_area);
        }
#if SLHIP_SHADOW_WINDOW
        __syncthreads();
        if (wx0 != 0x7fffffff) {
            unsigned* sm = shadow + ((size_t)ch.scene * nl + light) * S * S;
            for (int k = (int)threadIdx.x; k < kWin * kWin; k += 256) {
                const unsigned v = win[(k / kWin) * kPitch + (k % kWin)];
                const int x = wx0 + (k % kWin), y = wy0 + (k / kWin);
                if (v != 0x3F800000u && x < S && y < S) atomicMin(sm + (size_t)y * S + x, v);
            }
        }
        __syncthreads();
#endif
    }
    __syncthreads();
    for (int k = (int)threadIdx.x; k < SLHIP_NUM_LIGHTS * words; k += 256) {
        const int light = k / words, w = k % words;
        if (light >= nl || !light_active(sc, light)) continue;
        // maps with more tiles than the LDS bitmap holds (shadow_res > 2048) are marked wholesale
        const unsigned m = words <= kShadowMaxWords ? bm[light][w] : 0xFFFFFFFFu;
        if (m) atomicOr(tile_bits + ((size_t)ch.scene * SLHIP_NUM_LIGHTS + light) * words + w, m);
    }
}

// after the shading pass: every marked tile back to 1.0, its bit cleared (one block per 32-tile word)
__global__ __launch_bounds__(256) void k_shadow_restore(unsigned* __restrict__ shadow, unsigned* __restrict__ tile_bits, int S,
                                                        unsigned n_words_total, int nl)
{
    const unsigned wi = blockIdx.x;
    if (wi >= n_words_total) return;
    unsigned m = tile_bits[wi];
    if (m == 0u) return;
    const int words = shadow_tile_words(S), tx_n = shadow_tiles_x(S);
    const unsigned map = wi / (unsigned)words, w = wi % (unsigned)words;       // tile bits: [scene][SLHIP_NUM_LIGHTS][words]
    const unsigned m_scene = map / SLHIP_NUM_LIGHTS, m_light = map % SLHIP_NUM_LIGHTS;
    if ((int)m_light >= nl) return;                                            // maps: [scene][nl]
    unsigned* base = shadow + ((size_t)m_scene * nl + m_light) * S * S;
    const uint4 one = make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u);
    while (m) {
        const int b = __ffs((int)m) - 1;
        m &= m - 1;
        const int tile = (int)w * 32 + b;
        const int x0 = (tile % tx_n) * kShadowTile, y0 = (tile / tx_n) * kShadowTile;
        // 64 x 64 texels = 16 uint4 per row; S is a multiple of 4 (checked by slhip_render), tiles at the right / bottom edge are cut
        for (int k = (int)threadIdx.x; k < kShadowTile * (kShadowTile / 4); k += 256) {
            const int y = y0 + k / (kShadowTile / 4), x = x0 + 4 * (k % (kShadowTile / 4));
            if (y < S && x < S) *reinterpret_cast<uint4*>(base + (size_t)y * S + x) = one;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_bits[wi] = 0u;
}

__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_shadow_large(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                                      const slhip_draw* __restrict__ draws, int S,
                                                      unsigned* __restrict__ shadow,
                                                      const unsigned* __restrict__ queue, unsigned capacity,
                                                      const float4* __restrict__ clipbuf, unsigned n_clip_verts, int nl)
{
    const unsigned count = min(queue[0], capacity);
    const QItem* items = reinterpret_cast<const QItem*>(queue + 4);
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned n_waves = (gridDim.x * blockDim.x) >> 6;
    // contiguous runs per wave, setup shared by the tiles of one triangle (see k_large)
    const unsigned per = (count + n_waves - 1) / n_waves;
    const unsigned i0 = wave * per, i1 = min(i0 + per, count);
    unsigned p_draw = 0xFFFFFFFFu, p_tri = 0u, p_scene = 0u;
    bool ok = false;
    Setup t;
    for (unsigned i = i0; i < i1; ++i) {
        const QItem it = items[i];
        if (it.draw == 0xFFFFFFFFu) continue;
        const unsigned scene = it.scene_aux & 0xFFFFFFu;
        const int light = (int)(it.scene_aux >> 24);
        if (it.draw != p_draw || it.tri_sub != p_tri || it.scene_aux != p_scene) {
            p_draw = it.draw; p_tri = it.tri_sub; p_scene = it.scene_aux;
            ok = false;
            const slhip_draw* dr = draws + it.draw;
            const unsigned* ip = pool.d_idx + dr->idx_base + 3 * (size_t)it.tri_sub;
            const float4* plane = clipbuf + (size_t)(1 + light) * n_clip_verts + dr->clip_base;
            const uint4* sp = reinterpret_cast<const uint4*>(plane);
            if (!setup_from_screen(sp[ip[0]], sp[ip[1]], sp[ip[2]], S, S, t)) continue;
            ok = true;
        }
        if (!ok) continue;
        const int px = (int)((it.tile & 0xFFFFu) << 3) + (int)(lane & 7);
        const int py = (int)((it.tile >> 16) << 3) + (int)(lane >> 3);
        if (px < t.xmin || px > t.xmax || py < t.ymin || py > t.ymax) continue;
        float l[3];
        if (!coverage(t, px, py, l)) continue;
        ShadowTarget tgt;
        tgt.win = nullptr; tgt.wx0 = 0; tgt.wy0 = 0;
        tgt.sm = shadow + ((size_t)scene * nl + light) * S * S;
        tgt.W = S;
        tgt.emit(t, px, py, l);
    }
}

// ---------------------------------------------------------------------------------------------
// PBR shading (render_shader.frag:181-221, 248-399)
// ---------------------------------------------------------------------------------------------
// The BRDF terms feed the colour only (compared with the oracle to the 8-bit tolerance, not bit for bit): their divisions and
// the normalisation of L and H go through the hardware reciprocal / reciprocal square root (1 ulp) -- an IEEE division is 11
// instructions and a square root about as many in a kernel bound by VALU issue.  V stays exact: N . V is an output channel.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void normalize3_fast(float* v)
{
    const float r = __builtin_amdgcn_rsqf(fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
    v[0] *= r; v[1] *= r; v[2] *= r;
}

__device__ __forceinline__ float distribution_ggx(const float* N, const float* Hv, float roughness)
{
    const float a = roughness * roughness;
    const float a2 = a * a;
    const float NdotH = fmaxf(dot3(N, Hv), 0.0f);
    const float NdotH2 = NdotH * NdotH;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    denom = kPi * denom * denom;
    return a2 * frcp(denom);
}

__device__ __forceinline__ float geometry_schlick_ggx(float NdotV, float roughness)
{
    const float r = roughness + 1.0f;
    const float k = (r * r) * 0.125f;
    return NdotV * frcp(NdotV * (1.0f - k) + k);
}

__device__ __forceinline__ float shadow_tap(const float* __restrict__ sm, int S, float u, float v, float ref)
{
    const float x = fmaf(u, (float)S, -0.5f), y = fmaf(v, (float)S, -0.5f);
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), S - 1); x1 = min(max(x1, 0), S - 1);
    y0 = min(max(y0, 0), S - 1); y1 = min(max(y1, 0), S - 1);
    const float r = clampf(ref, 0.0f, 1.0f);
    const float c00 = r <= sm[(size_t)y0 * S + x0] ? 1.0f : 0.0f;
    const float c10 = r <= sm[(size_t)y0 * S + x1] ? 1.0f : 0.0f;
    const float c01 = r <= sm[(size_t)y1 * S + x0] ? 1.0f : 0.0f;
    const float c11 = r <= sm[(size_t)y1 * S + x1] ? 1.0f : 0.0f;
    const float top = fmaf(ax, c10 - c00, c00), bot = fmaf(ax, c11 - c01, c01);
    return fmaf(ay, bot - top, top);
}

// 4x4 PCF of bilinear depth compares (render_shader.frag:181-221).  The sixteen taps of
// shadow_tap() touch a 5x5 texel window; when the per-axis tap coordinates are consecutive (always,
// except where rounding makes two taps share a texel) the window is fetched once and every tap is
// evaluated from registers with exactly the arithmetic and summation order of the tap-by-tap form.
__device__ __forceinline__ float shadow_pcf16(const float* __restrict__ sm, int S, float px, float py, float ref,
                                              const unsigned* __restrict__ tile_bits)
{
    const float scale = 1.0f / (float)S;
    int ix[4], iy[4];
    float ax[4], ay[4];
    bool regular = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float o = (-1.5f + (float)k) * scale;
        const float x = fmaf(px + o, (float)S, -0.5f), y = fmaf(py + o, (float)S, -0.5f);
        const float fx = floorf(x), fy = floorf(y);
        ax[k] = x - fx; ay[k] = y - fy;
        ix[k] = (int)fx; iy[k] = (int)fy;
        if (k > 0 && (ix[k] != ix[0] + k || iy[k] != iy[0] + k)) regular = false;
    }
    float acc = 0.0f;
    if (regular) {
        const float r = clampf(ref, 0.0f, 1.0f);
        unsigned off_y[5];
        int cx[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            cx[k] = min(max(ix[0] + k, 0), S - 1);
            off_y[k] = (unsigned)min(max(iy[0] + k, 0), S - 1) * (unsigned)S;
        }
        if (tile_bits) {
            // The shadow raster files which 64 x 64-texel tiles of the map its casters touched (the bits k_shadow_restore clears the
            // map by): a window that lies in untouched tiles holds the cleared depth 1.0 in all 25 texels, every compare passes
            // (r <= 1), every tap is exactly 1 -- the same 1.0 as below, without fetching the window.  Most of the background plane.
            const int tn = shadow_tiles_x(S);
            const int ta = cx[0] / kShadowTile, tb = cx[4] / kShadowTile;
            const int ra = (int)(off_y[0] / (unsigned)S) / kShadowTile, rb = (int)(off_y[4] / (unsigned)S) / kShadowTile;
            const int t00 = ra * tn + ta, t01 = ra * tn + tb, t10 = rb * tn + ta, t11 = rb * tn + tb;
            const unsigned touched = ((tile_bits[t00 >> 5] >> (t00 & 31)) | (tile_bits[t01 >> 5] >> (t01 & 31)) |
                                      (tile_bits[t10 >> 5] >> (t10 & 31)) | (tile_bits[t11 >> 5] >> (t11 & 31))) & 1u;
            if (__ballot(touched == 0u) == __ballot(true)) return 1.0f;
        }
        bool lit[5][5];
        unsigned long long all_lit = ~0ull, any_lit = 0ull;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                lit[j][i] = r <= sm[off_y[j] + (unsigned)cx[i]];
                const unsigned long long m = __ballot(lit[j][i]);
                all_lit &= m; any_lit |= m;
            }
        // Away from shadow edges all 25 compares of a pixel agree, and then the sixteen bilinear taps are exactly 1 (or 0) each
        // and their sum exactly 16 (or 0): when that holds for every lane of the wave that is here, the lerps are skipped.
        const unsigned long long here = __ballot(true);
        if (all_lit == here) return 1.0f;
        if (any_lit == 0ull) return 0.0f;
        float c[5][5];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i) c[j][i] = lit[j][i] ? 1.0f : 0.0f;
#pragma unroll
        for (int yy = 0; yy < 4; ++yy)
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                const float top = fmaf(ax[xx], c[yy][xx + 1] - c[yy][xx], c[yy][xx]);
                const float bot = fmaf(ax[xx], c[yy + 1][xx + 1] - c[yy + 1][xx], c[yy + 1][xx]);
                acc += fmaf(ay[yy], bot - top, top);
            }
    } else {
        for (int yy = 0; yy < 4; ++yy)
            for (int xx = 0; xx < 4; ++xx) {
                const float ox = (-1.5f + (float)xx) * scale, oy = (-1.5f + (float)yy) * scale;
                acc += shadow_tap(sm, S, px + ox, py + oy, ref);
            }
    }
    return acc / 16.0f;
}

// x^2.2 of a filtered texel (sRGB -> linear, render_shader.frag:243) through the hardware log2 / exp2 (1 ulp each): colour
// values are compared with the oracle to the 8-bit tolerance, not bit for bit, and libm-grade powf costs ~50 instructions
// per channel in the kernel that is bound by VALU issue.  0 -> exp2(-inf) = 0.
__device__ __forceinline__ float pow22(float x) { return __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(x)); }

__device__ __forceinline__ void shade_fragment(const slhip_scene* __restrict__ sc, const slhip_draw* __restrict__ dr,
                                               const float* base, const float* world, const float* nrm_in,
                                               bool front_facing, const float* __restrict__ shadow, int S, int shadow_lights,
                                               const unsigned* __restrict__ shadow_tiles, const slhip_light_map* __restrict__ lm, float roughness_in, float metallic_in,
                                               float occlusion, const float* emissive, float* color, float* normal_out)
{
    float normal[3] = {nrm_in[0], nrm_in[1], nrm_in[2]};
    if (!front_facing) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
    float V[3] = {sc->cam_position[0] - world[0], sc->cam_position[1] - world[1],
                  sc->cam_position[2] - world[2]};
    normalize3(V);
    const float NoV = clampf(dot3(normal, V), 1e-5f, 1.0f);
    const float roughness = fmaxf(roughness_in, 0.045f);
    const float metallic = metallic_in;

    color[0] = color[1] = color[2] = 0.0f;
    color[3] = base[3];

    float F0[3], kS[3];
    const float om = 1.0f - NoV, om2 = om * om;
    const float p5 = om2 * om2 * om;                  // (1 - NoV)^5
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        F0[c] = 0.04f * (1.0f - metallic) + base[c] * metallic;
        const float Fr = fmaxf(1.0f - roughness, F0[c]) - F0[c];
        kS[c] = F0[c] + Fr * p5;
    }
    const float base_pi[3] = {base[0] * (1.0f / kPi), base[1] * (1.0f / kPi), base[2] * (1.0f / kPi)};   // Lambert term, once per pixel
    for (int i = 0; i < SLHIP_NUM_LIGHTS; ++i) {
        if (!light_active(sc, i)) continue;
        const float* lc = sc->light_color[i];
        const float* ld = sc->light_dir[i];
        float inverse_shadow = 1.0f;
        if (shadow && i < shadow_lights) {
            const float w4[4] = {world[0], world[1], world[2], 1.0f};
            float pc[4];
            mv4(sc->shadow_mat[i], w4, pc);
            const float rpw = 1.0f / pc[3];   /* shared by the three perspective divisions */
            const float px = fmaf(pc[0] * rpw, 0.5f, 0.5f);
            const float py = fmaf(pc[1] * rpw, 0.5f, 0.5f);
            const float pz = fmaf(pc[2] * rpw, 0.5f, 0.5f);
            const float* sm = shadow + (size_t)i * S * S;
            inverse_shadow = shadow_pcf16(sm, S, px, py, (pz - 0.00003f), shadow_tiles ? shadow_tiles + (size_t)i * shadow_tile_words(S) : nullptr);
        }
        float L[3] = {-ld[0], -ld[1], -ld[2]};
        normalize3_fast(L);
        float Hv[3] = {V[0] + L[0], V[1] + L[1], V[2] + L[2]};
        normalize3_fast(Hv);
        const float NDF = distribution_ggx(normal, Hv, roughness);
        const float NdotVg = fmaxf(dot3(normal, V), 0.0f);
        const float NdotL = fmaxf(dot3(normal, L), 0.0f);
        const float G = geometry_schlick_ggx(NdotL, roughness) * geometry_schlick_ggx(NdotVg, roughness);
        const float denominator = fmaxf(4.0f * NoV * NdotL, 0.001f);
        const float rden = frcp(denominator);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float specular = (NDF * G * kS[c]) * rden;
            const float kD = (1.0f - kS[c]) * (1.0f - metallic);
            color[c] += inverse_shadow * (kD * base_pi[c] + specular) * lc[c] * NdotL;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) color[c] += sc->ambient[c] * base[c];
    if (lm) {
        // image-based lighting (render_shader.frag:375-394): split-sum specular + diffuse irradiance with the
        // multiple-scattering correction of Fdez-Aguera; occlusion = 1 (no occlusion texture on this path)
        const float d2 = 2.0f * dot3(normal, V);
        const slcube::f3 refl = slcube::F3(d2 * normal[0] - V[0], d2 * normal[1] - V[1], d2 * normal[2] - V[2]);   // reflect(-V, N)
        float fab[2];
        {
            const int n = (int)lm->lut_size;
            const float x = NoV * (float)n - 0.5f, y = roughness * (float)n - 0.5f;
            const float fx = floorf(x), fy = floorf(y);
            const float a = x - fx, b = y - fy;
            const int x0 = min(max((int)fx, 0), n - 1), x1 = min(max((int)fx + 1, 0), n - 1);
            const int y0 = min(max((int)fy, 0), n - 1), y1 = min(max((int)fy + 1, 0), n - 1);
            const float2* t = reinterpret_cast<const float2*>(lm->d_brdf_lut);
            const float2 c00 = t[y0 * n + x0], c10 = t[y0 * n + x1], c01 = t[y1 * n + x0], c11 = t[y1 * n + x1];
            const float tx = fmaf(a, c10.x - c00.x, c00.x), bx = fmaf(a, c11.x - c01.x, c01.x);
            const float ty = fmaf(a, c10.y - c00.y, c00.y), by = fmaf(a, c11.y - c01.y, c01.y);
            fab[0] = fmaf(b, bx - tx, tx); fab[1] = fmaf(b, by - ty, ty);
        }
        const float4 rad4 = slcube::sample_lod(lm->d_prefilter, lm->pre_size, lm->pre_levels, refl, roughness * 4.0f);
        const float4 irr4 = slcube::sample_lod(lm->d_irradiance, lm->irr_size, 1u, slcube::F3(normal[0], normal[1], normal[2]), 0.0f);
        const float rad[3] = {rad4.x, rad4.y, rad4.z}, irr[3] = {irr4.x, irr4.y, irr4.z};
        const float Ems = 1.0f - (fab[0] + fab[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float c_diff = base[c] * (1.0f - 0.04f) * (1.0f - metallic);
            const float FssEss = kS[c] * fab[0] + fab[1];
            const float F_avg = F0[c] + (1.0f - F0[c]) / 21.0f;
            const float FmsEms = Ems * FssEss * F_avg / (1.0f - F_avg * Ems);
            const float k_D = c_diff * (1.0f - FssEss - FmsEms);
            color[c] += (FssEss * rad[c] + (FmsEms + k_D) * irr[c]) * occlusion;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) color[c] += emissive[c];
    float nc[3];
#pragma unroll
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

// tone map of one HDR texel (tone_map_shader.frag:102-131); exposure_div < 0 => multiply by
// manual exposure.  Colour only (8-bit tolerance): five of the six divisions go through the hardware reciprocal (1 ulp) -- an IEEE
// division is eleven instructions in a pipeline bound by instruction issue.
__device__ __forceinline__ uchar4 tone_map_px(const float* c, float manual_exposure, float lum)
{
    const float X = 0.4124564f * c[0] + 0.3575761f * c[1] + 0.1804375f * c[2];
    float Y = 0.2126729f * c[0] + 0.7151522f * c[1] + 0.0721750f * c[2];
    const float Z = 0.0193339f * c[0] + 0.1191920f * c[1] + 0.9503041f * c[2];
    const float inv = frcp(X + Y + Z);
    const float xx = X * inv, yy = Y * inv;
    if (manual_exposure >= 0.0f) Y *= manual_exposure;
    else Y /= (9.6f * lum + 0.0001f);
    const float ryy = frcp(yy);
    const float x2 = Y * xx * ryy;
    const float y2 = Y;
    const float z2 = Y * (1.0f - xx - yy) * ryy;
    float o[3];
    o[0] = 3.2404542f * x2 + -1.5371385f * y2 + -0.4985314f * z2;
    o[1] = -0.9692660f * x2 + 1.8760108f * y2 + 0.0415560f * z2;
    o[2] = 0.0556434f * x2 + -0.2040259f * y2 + 1.0572252f * z2;
    unsigned char r[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = o[k];
        float v = (x * (2.51f * x + 0.03f)) * frcp(x * (2.43f * x + 0.59f) + 0.14f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        if (!(v == v)) v = 0.0f;
        r[k] = (unsigned char)floorf(v * 255.0f + 0.5f);
    }
    const float a = fminf(fmaxf(c[3], 0.0f), 1.0f);
    r[3] = (unsigned char)floorf(a * 255.0f + 0.5f);
    return make_uchar4(r[0], r[1], r[2], r[3]);
}

// ---------------------------------------------------------------------------------------------
// k_shade: deferred resolve, one thread per pixel
// ---------------------------------------------------------------------------------------------
// Per-pixel kernels: block -> (scene, 256-pixel block).  Consecutive blocks are observed to land on
// consecutive XCDs (block b on XCD b % 8, MI355X_MICROARCH.md "Workgroup dispatch"), and every XCD has
// its own L2: all blocks of a scene get the same b % 8, so the scene's vertices, shadow map, z plane
// and AO plane are fetched into ONE L2 instead of eight.  Placement only affects speed.
// Grid = 8 * ceil(n_scenes / 8) * blocks_per_scene.
__device__ __forceinline__ bool scene_block(unsigned blocks_per_scene, unsigned n_scenes, unsigned& scene, unsigned& blk)
{
    const unsigned x = blockIdx.x & 7u, j = blockIdx.x >> 3;
    scene = x + 8u * (j / blocks_per_scene);
    blk = j % blocks_per_scene;
    return scene < n_scenes;
}

// The shading pass's geometry resolve of one pixel against one set-up (sub-)triangle whose corners carry the barycentrics
// q0..q2 of the ORIGINAL triangle: perspective-correct barycentrics b, facing, and -- where a texture will be sampled -- the
// barycentrics one pixel to the right / below (the texture footprint).  Returns false if the pixel is not covered.
struct PixelBary {
    float b[3], bx[3], by[3];
    bool front;
};
__device__ __forceinline__ bool resolve_pixel(const Setup& t, const float (&q0)[3], const float (&q1)[3], const float (&q2)[3],
                                              int px, int py, bool textured, PixelBary& r)
{
    float l[3];
    if (px < t.xmin || px > t.xmax || py < t.ymin || py > t.ymax) return false;
    if (!coverage(t, px, py, l)) return false;
    const float pw0 = l[0] * t.invw[0], pw1 = l[1] * t.invw[1], pw2 = l[2] * t.invw[2];
    const float sw = (pw0 + pw1) + pw2;
    const float bs[3] = {pw0 * (1.0f / sw), pw1 * (1.0f / sw), pw2 * (1.0f / sw)};
#pragma unroll
    for (int k = 0; k < 3; ++k) r.b[k] = fmaf(bs[2], q2[k], fmaf(bs[1], q1[k], bs[0] * q0[k]));
    r.front = t.flipped;
    if (textured) {
        bary_at(t, q0, q1, q2, px + 1, py, r.bx);
        bary_at(t, q0, q1, q2, px, py + 1, r.by);
    }
    return true;
}

struct ShadeParams {
    int W, H;
    unsigned n_scenes;
    unsigned flags;
    int S;
    int shadow_lights;   // light maps per scene in `shadow`
    int inline_tonemap;  // 1: no SSAO and manual exposure -> write rgb directly
    int want_lum;        // 1: write per-block HDR sums for auto exposure
    int tiled;           // 1: a block shades a 32 x 8 pixel region, a wave an 8 x 8 tile (needs W % 32 == 0 and H % 8 == 0)
};

// The corners' vertex-stage outputs and window coordinates come from the records k_vertex_xform wrote (`vattr`).
__global__ __launch_bounds__(256) SLHIP_SHADE_KERNEL void k_shade(slhip_mesh_pool pool, const slhip_scene* __restrict__ scenes,
                                               const slhip_draw* __restrict__ draws, ShadeParams prm,
                                               const unsigned long long* __restrict__ vis,
                                               slhip_render_out out, float* __restrict__ hdr,
                                               const float* __restrict__ shadow, float* __restrict__ lum_part,
                                               const float4* __restrict__ clipbuf, float* __restrict__ zplane,
                                               const float4* __restrict__ vattr, float2* __restrict__ ssao_tiles,
                                               const unsigned* __restrict__ shadow_tiles)
{
    const int W = prm.W, H = prm.H;
    const size_t P = (size_t)W * H;
    const unsigned blocks_per_scene = (unsigned)((P + 255) / 256);
    unsigned scene, blk;
    if (!scene_block(blocks_per_scene, prm.n_scenes, scene, blk)) return;
    // Pixel of the thread.  Tiled (whenever the viewport divides into 32 x 8 regions): a wave shades an 8 x 8 tile, so the
    // silhouette of an object cuts through fewer waves than with 64 x 1 strips -- a wave that holds one object pixel pays the
    // whole fragment path (textures, 25 shadow texels) for all its lanes -- and the gathers of a wave land in a compact
    // window of the texture and of the shadow map.  Every lane still stores 16 contiguous bytes per target, a tile row is
    // one 128-byte line.  Otherwise: 256 consecutive pixels per block.  Placement only: the values do not depend on it.
    unsigned pix;
    bool active;
    if (prm.tiled) {
        const unsigned per_row = (unsigned)W >> 5, l = threadIdx.x & 63u;
        const unsigned tx = (blk % per_row) * 32u + (threadIdx.x >> 6) * 8u + (l & 7u), ty = (blk / per_row) * 8u + (l >> 3);
        pix = ty * (unsigned)W + tx;
        active = true;
    } else {
        pix = blk * 256 + threadIdx.x;
        active = pix < P;
    }
    const slhip_scene* sc = scenes + scene;
    const size_t gp = (size_t)scene * P + pix;
    // the block's pixels belong to one scene: its draws' first primitive ids go to LDS once
    __shared__ unsigned s_prim_base[64];
    const unsigned n_scene_draws = sc->draw_end - sc->draw_begin;
    if (threadIdx.x < 64) s_prim_base[threadIdx.x] = threadIdx.x < n_scene_draws ? draws[sc->draw_begin + threadIdx.x].prim_base : 0xFFFFFFFFu;
    __syncthreads();

    const slhip_light_map* lm = (sc->light_map != 0u && pool.d_light_maps) ? pool.d_light_maps + (sc->light_map - 1u) : nullptr;
    float color[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float lum_color[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // what the exposure average sees: objects over the cleared target
    if (active) {
        const unsigned long long key = vis[gp];
        float coord[4] = {kInvalid, kInvalid, kInvalid, kInvalid};
        float camc[4] = {kInvalid, kInvalid, kInvalid, kInvalid};
        float nout[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float bary[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        unsigned vidx[4] = {0u, 0u, 0u, 0u};
        unsigned cls = 0u, inst = 0u;
        bool plane_px = false;     // a pixel of the background plane (flat, geometric normal): see k_ssao_mask
        if (key != kVisEmpty) {
            const unsigned prim = (unsigned)(key & 0xFFFFFFFFull);
            unsigned d = sc->draw_begin;
            if (n_scene_draws <= 64) {
                // the last draw whose first primitive id is <= prim: six halving steps over the table (padded with ~0)
                unsigned lo = 0u;
#pragma unroll
                for (unsigned step = 32u; step > 0u; step >>= 1)
                    if (prim >= s_prim_base[lo + step]) lo += step;
                d = sc->draw_begin + lo;
            } else {
                for (unsigned i = sc->draw_begin + 1; i < sc->draw_end; ++i)
                    if (prim >= draws[i].prim_base) d = i;
            }
            const slhip_draw* dr = draws + d;
            const unsigned tri = prim - dr->prim_base;
            const unsigned* ip = pool.d_idx + dr->idx_base + 3 * (size_t)tri;
            const unsigned vi[3] = {ip[0], ip[1], ip[2]};
            VsOut vo[3];
            const int px = (int)(pix % (unsigned)W), py = (int)(pix / (unsigned)W);
            float b[3] = {0.0f, 0.0f, 0.0f}, bx[3] = {0.0f, 0.0f, 0.0f}, by[3] = {0.0f, 0.0f, 0.0f};
            bool found = false, front = false;
            uint4 sv[3] = {};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4* va = vattr + 4 * (size_t)(dr->clip_base + vi[k]);   // the corner's 64-byte record
                const float4 a0 = va[0], a1 = va[1], a2 = va[2];
                sv[k] = reinterpret_cast<const uint4*>(va)[3];
                const float2 uv = reinterpret_cast<const float2*>(pool.d_uv)[dr->vtx_base + vi[k]];
                vo[k].objc[0] = a0.x; vo[k].objc[1] = a0.y; vo[k].objc[2] = a0.z; vo[k].objc[3] = a0.w;
                vo[k].world[0] = a1.x; vo[k].world[1] = a1.y; vo[k].world[2] = a1.z;
                vo[k].cam[0] = a1.w; vo[k].cam[1] = a2.w; vo[k].cam[2] = a0.w;
                vo[k].nrm[0] = a2.x; vo[k].nrm[1] = a2.y; vo[k].nrm[2] = a2.z;
                vo[k].uv[0] = uv.x; vo[k].uv[1] = uv.y;
            }
            // the texture footprint (barycentrics one pixel to the right / below) only where a texture will be sampled
            const bool textured = (dr->flags & (SLHIP_DRAW_HAS_BASE_TEX | SLHIP_DRAW_HAS_NORMAL_TEX | SLHIP_DRAW_HAS_MR_TEX |
                                                SLHIP_DRAW_HAS_OCCLUSION_TEX | SLHIP_DRAW_HAS_EMISSIVE_TEX)) != 0u;
            PixelBary pb;
#pragma unroll
            for (int k = 0; k < 3; ++k) pb.b[k] = pb.bx[k] = pb.by[k] = 0.0f;
            pb.front = false;
            if (screen_all_inside(sv[0], sv[1], sv[2])) {
                // all three corners in front of the near plane (clip_near() would hand the triangle through): set-up from the
                // per-vertex window coordinates, no clip positions needed, the corners' barycentrics are the unit vectors
                const float e0[3] = {1.0f, 0.0f, 0.0f}, e1[3] = {0.0f, 1.0f, 0.0f}, e2[3] = {0.0f, 0.0f, 1.0f};
                Setup t;
                if (setup_from_screen(sv[0], sv[1], sv[2], W, H, t)) found = resolve_pixel(t, e0, e1, e2, px, py, textured, pb);
            } else {
                ClipVert cv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float4 c4 = clipbuf[dr->clip_base + vi[k]];   // identical to what the raster used
                    cv[k].clip[0] = c4.x; cv[k].clip[1] = c4.y; cv[k].clip[2] = c4.z; cv[k].clip[3] = c4.w;
                    cv[k].bary[0] = k == 0 ? 1.0f : 0.0f;
                    cv[k].bary[1] = k == 1 ? 1.0f : 0.0f;
                    cv[k].bary[2] = k == 2 ? 1.0f : 0.0f;
                }
                ClipVert poly[4];
                const int n = clip_near(cv, poly);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {   // unrolled: static indices into poly[]
                    if (sub >= n - 2 || found) break;
                    Setup t;
                    if (!setup_tri(poly[0].clip, poly[sub + 1].clip, poly[sub + 2].clip, W, H, t)) continue;
                    found = resolve_pixel(t, poly[0].bary, poly[sub + 1].bary, poly[sub + 2].bary, px, py, textured, pb);
                }
            }
            front = pb.front;
#pragma unroll
            for (int k = 0; k < 3; ++k) { b[k] = pb.b[k]; bx[k] = pb.bx[k]; by[k] = pb.by[k]; }
            if (found) {
                const float camz = interp(b, vo[0].objc[3], vo[1].objc[3], vo[2].objc[3]);
                float base[4] = {dr->base_color[0], dr->base_color[1], dr->base_color[2], dr->base_color[3]};
                const float u = interp(b, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]);
                const float v = interp(b, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]);
                const float dudx = interp(bx, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]) - u, dvdx = interp(bx, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]) - v;
                const float dudy = interp(by, vo[0].uv[0], vo[1].uv[0], vo[2].uv[0]) - u, dvdy = interp(by, vo[0].uv[1], vo[1].uv[1], vo[2].uv[1]) - v;
                if (dr->flags & SLHIP_DRAW_HAS_BASE_TEX) {
                    float tc[4];
                    tex_sample(pool.d_tex, dr->tex_offset, (int)dr->tex_w, (int)dr->tex_h, dr->tex_sampler[0], u, v, dudx, dvdx, dudy, dvdy, tc);
                    base[0] *= pow22(tc[0]);
                    base[1] *= pow22(tc[1]);
                    base[2] *= pow22(tc[2]);
                    base[3] *= tc[3];
                }
                float world[3], nrm[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    world[k] = interp(b, vo[0].world[k], vo[1].world[k], vo[2].world[k]);
                    nrm[k] = interp(b, vo[0].nrm[k], vo[1].nrm[k], vo[2].nrm[k]);
                    coord[k] = interp(b, vo[0].objc[k], vo[1].objc[k], vo[2].objc[k]);
                    camc[k] = interp(b, vo[0].cam[k], vo[1].cam[k], vo[2].cam[k]);
                }
                coord[3] = camz;
                camc[3] = 1.0f;
                if (dr->flags & SLHIP_DRAW_HAS_STICKER) {
                    // projected decal (render_shader.vert:89-94, frag:248-256): per-vertex projection of the object
                    // coordinates into the sticker frame, interpolated like every other varying
                    float sx[3], sy[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 p4 = reinterpret_cast<const float4*>(pool.d_pos)[dr->vtx_base + vi[k]];
                        const float pos[4] = {p4.x, p4.y, p4.z, 1.0f};
                        float obj4[4], sp[4];
                        mv4(dr->mesh_to_object, pos, obj4);
                        mv4(dr->sticker_projection, obj4, sp);
                        sx[k] = (sp[0] / sp[3] - dr->sticker_range[0]) / dr->sticker_range[2];
                        sy[k] = (sp[1] / sp[3] - dr->sticker_range[1]) / dr->sticker_range[3];
                    }
                    const float cx = interp(b, sx[0], sx[1], sx[2]), cy = interp(b, sy[0], sy[1], sy[2]);
                    if (cx >= 0.0f && cy >= 0.0f && cx < 1.0f && cy < 1.0f) {
                        float st[4];
                        tex_rect_bilinear(pool.d_tex + dr->sticker_tex_offset, (int)dr->sticker_tex_w, (int)dr->sticker_tex_h,
                                          cx * (float)dr->sticker_tex_w, cy * (float)dr->sticker_tex_h, st);
                        const float sc4[4] = {pow22(st[0]), pow22(st[1]), pow22(st[2]), st[3]};
#pragma unroll
                        for (int c = 0; c < 4; ++c) base[c] = base[c] * (1.0f - st[3]) + sc4[c] * st[3];   // mix(base, sticker, sticker.a)
                    }
                }
                if (dr->flags & SLHIP_DRAW_HAS_NORMAL_TEX) {
                    // tangent-space normal map (vert:77-80, frag:262-266)
                    float tw[3][3], bw[3][3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 t4 = reinterpret_cast<const float4*>(pool.d_tan)[dr->vtx_base + vi[k]];
                        const float tv[3] = {t4.x, t4.y, t4.z};
                        mv3p(dr->normal_to_world, tv, tw[k]);
                        normalize3(tw[k]);
                        bw[k][0] = vo[k].nrm[1] * tw[k][2] - vo[k].nrm[2] * tw[k][1];
                        bw[k][1] = vo[k].nrm[2] * tw[k][0] - vo[k].nrm[0] * tw[k][2];
                        bw[k][2] = vo[k].nrm[0] * tw[k][1] - vo[k].nrm[1] * tw[k][0];
                        normalize3(bw[k]);
#pragma unroll
                        for (int c = 0; c < 3; ++c) bw[k][c] *= t4.w;
                    }
                    float tc[4];
                    tex_sample(pool.d_tex, dr->normal_tex_offset, (int)dr->normal_tex_w, (int)dr->normal_tex_h, dr->tex_sampler[1], u, v, dudx, dvdx, dudy, dvdy, tc);
                    const float nx = tc[0] * 2.0f - 1.0f, ny = tc[1] * 2.0f - 1.0f, nz = tc[2] * 2.0f - 1.0f;
                    float nn[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        nn[c] = nx * interp(b, tw[0][c], tw[1][c], tw[2][c]) + ny * interp(b, bw[0][c], bw[1][c], bw[2][c]) + nz * nrm[c];
                    normalize3(nn);
                    nrm[0] = nn[0]; nrm[1] = nn[1]; nrm[2] = nn[2];
                }
                float roughness = dr->roughness, metallic = dr->metallic, occlusion = 1.0f;
                float emissive[3] = {dr->emissive[0], dr->emissive[1], dr->emissive[2]};
                if (dr->flags & SLHIP_DRAW_HAS_MR_TEX) {
                    float tc[4];
                    tex_sample(pool.d_tex, dr->mr_tex_offset, (int)dr->mr_tex_w, (int)dr->mr_tex_h, dr->tex_sampler[2], u, v, dudx, dvdx, dudy, dvdy, tc);
                    roughness *= tc[1];
                    metallic *= tc[2];
                }
                if (dr->flags & SLHIP_DRAW_HAS_OCCLUSION_TEX) {
                    float tc[4];
                    tex_sample(pool.d_tex, dr->occlusion_tex_offset, (int)dr->occlusion_tex_w, (int)dr->occlusion_tex_h, dr->tex_sampler[3], u, v, dudx, dvdx, dudy, dvdy, tc);
                    occlusion = tc[0];
                }
                if (dr->flags & SLHIP_DRAW_HAS_EMISSIVE_TEX) {
                    float tc[4];
                    tex_sample(pool.d_tex, dr->emissive_tex_offset, (int)dr->emissive_tex_w, (int)dr->emissive_tex_h, dr->tex_sampler[4], u, v, dudx, dvdx, dudy, dvdy, tc);
#pragma unroll
                    for (int c = 0; c < 3; ++c) emissive[c] *= pow22(tc[c]);
                }
                const float* sm = (prm.flags & SLHIP_RENDER_SHADOWS) && shadow
                                      ? shadow + (size_t)scene * prm.shadow_lights * prm.S * prm.S
                                      : nullptr;
                shade_fragment(sc, dr, base, world, nrm, front, sm, prm.S, prm.shadow_lights,
                               sm && shadow_tiles ? shadow_tiles + (size_t)scene * SLHIP_NUM_LIGHTS * shadow_tile_words(prm.S) : nullptr, lm, roughness, metallic, occlusion, emissive, color, nout);
                cls = dr->class_index & 0xFFFFu;
                inst = dr->instance_index & 0xFFFFu;
                plane_px = (dr->flags & (SLHIP_DRAW_NO_VERTEX_ID | SLHIP_DRAW_HAS_NORMAL_TEX)) == SLHIP_DRAW_NO_VERTEX_ID && inst == 0u;
                if (!(dr->flags & SLHIP_DRAW_NO_VERTEX_ID)) { vidx[0] = vi[0] + 1; vidx[1] = vi[1] + 1; vidx[2] = vi[2] + 1; }
                bary[0] = b[0]; bary[1] = b[1]; bary[2] = b[2];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) lum_color[c] = color[c];   // the 1x1 mip is generated before any background is drawn
        if (key == kVisEmpty && sc->bg_tex[1] != 0u && sc->bg_tex[2] != 0u) {
            // background image (render_pass.cpp:637-646): stretched over the viewport, integer texel coordinates on a
            // LINEAR rectangle sampler, alpha 0; fills the pixels where nothing was rasterised (see oracle/render_ref.c)
            const int px = (int)(pix % (unsigned)W), py = (int)(pix / (unsigned)W);
            const int tw = (int)sc->bg_tex[1], th = (int)sc->bg_tex[2];
            const float tcx = ((float)px + 0.5f) / (float)W, tcy = 1.0f - ((float)py + 0.5f) / (float)H;
            float c[4];
            tex_rect_bilinear(pool.d_tex + sc->bg_tex[0], tw, th, (float)(int)(tcx * (float)tw), (float)(int)(tcy * (float)th), c);
            color[0] = c[0]; color[1] = c[1]; color[2] = c[2]; color[3] = 0.0f;
        } else if (key == kVisEmpty && lm) {
            // sky background (render_pass.cpp:647-661, background_cube_shader.*): the environment along the
            // pixel's view ray, alpha 0; the other targets keep their clear values
            const int px = (int)(pix % (unsigned)W), py = (int)(pix / (unsigned)W);
            const float xn = (2.0f * ((float)px + 0.5f)) / (float)W - 1.0f, yn = (2.0f * ((float)py + 0.5f)) / (float)H - 1.0f;
            const float dcx = (xn - sc->proj[2]) / sc->proj[0], dcy = (yn - sc->proj[6]) / sc->proj[5];
            // camera -> world rotation = transpose of the rotation block of world_to_cam
            const float* m = sc->world_to_cam;
            const slcube::f3 dw = slcube::F3(fmaf(m[8], 1.0f, fmaf(m[4], dcy, m[0] * dcx)), fmaf(m[9], 1.0f, fmaf(m[5], dcy, m[1] * dcx)),
                                            fmaf(m[10], 1.0f, fmaf(m[6], dcy, m[2] * dcx)));
            const float4 e = slcube::sample_lod(lm->d_env, lm->env_size, lm->env_levels, dw, 0.0f);
            color[0] = e.x; color[1] = e.y; color[2] = e.z; color[3] = 0.0f;
        }
        if (out.d_coord) reinterpret_cast<float4*>(out.d_coord)[gp] = make_float4(coord[0], coord[1], coord[2], coord[3]);
        if (out.d_class) out.d_class[gp] = (uint16_t)cls;
        if (out.d_instance) out.d_instance[gp] = (uint16_t)inst;
        if (out.d_normals) reinterpret_cast<float4*>(out.d_normals)[gp] = make_float4(nout[0], nout[1], nout[2], nout[3]);
        if (out.d_vertex_idx) reinterpret_cast<uint4*>(out.d_vertex_idx)[gp] = make_uint4(vidx[0], vidx[1], vidx[2], vidx[3]);
        if (out.d_bary) reinterpret_cast<float4*>(out.d_bary)[gp] = make_float4(bary[0], bary[1], bary[2], bary[3]);
        if (out.d_cam_coord) reinterpret_cast<float4*>(out.d_cam_coord)[gp] = make_float4(camc[0], camc[1], camc[2], camc[3]);
        if (zplane) {
            // compact camera-z plane for the SSAO passes, with a one-texel border that repeats the edge (what the clamped
            // texel fetches of a rect sampler return there): the 64-tap gather then needs no per-tap clamps of x + 1 / y + 1
            const int Wp = W + 2;
            float* zp = zplane + (size_t)scene * ((size_t)Wp * (H + 2));
            const int zi = (int)(pix % (unsigned)W), zj = (int)(pix / (unsigned)W);
            const int o = (zj + 1) * Wp + zi + 1;
            const float z = camc[2];
            const bool l = zi == 0, r = zi == W - 1, t = zj == 0, b = zj == H - 1;
            zp[o] = z;
            if (l) zp[o - 1] = z;
            if (r) zp[o + 1] = z;
            if (t) zp[o - Wp] = z;
            if (b) zp[o + Wp] = z;
            if (l && t) zp[o - Wp - 1] = z;
            if (r && t) zp[o - Wp + 1] = z;
            if (l && b) zp[o + Wp - 1] = z;
            if (r && b) zp[o + Wp + 1] = z;
        }
        if (ssao_tiles) {
            // what k_ssao_mask needs of the wave's 8 x 8 tile: the nearest camera z of its geometry, and whether any of it is
            // something else than the background plane (tiled placement only: the wave IS the tile, every lane is active)
            const bool geo = key != kVisEmpty;
            const unsigned long long other = __ballot(geo && !plane_px), any = __ballot(geo);
            float zm = geo ? camc[2] : 3.0e38f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) zm = fminf(zm, __shfl_xor(zm, o, 64));
            if ((threadIdx.x & 63u) == 0u) {
                const unsigned tw = (unsigned)W >> 3;
                const unsigned tx = (pix % (unsigned)W) >> 3, ty = (pix / (unsigned)W) >> 3;
                ssao_tiles[(size_t)scene * (tw * ((unsigned)H >> 3)) + ty * tw + tx] =
                    make_float2(zm, __uint_as_float((other ? 1u : 0u) | (any ? 2u : 0u)));
            }
        }
        if (prm.inline_tonemap) {
            if (out.d_rgb) reinterpret_cast<uchar4*>(out.d_rgb)[gp] = tone_map_px(color, sc->manual_exposure, 0.0f);
        } else if (hdr) {
            reinterpret_cast<float4*>(hdr)[gp] = make_float4(color[0], color[1], color[2], color[3]);
        }
    }
    if (prm.want_lum) {
        // deterministic block sum (fixed tree) -> one partial per block
        __shared__ float red[4][256];
#pragma unroll
        for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = lum_color[c];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
#pragma unroll
                for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x < 4) lum_part[((size_t)scene * blocks_per_scene + blk) * 4 + threadIdx.x] = red[threadIdx.x][0];
    }
}

// ---------------------------------------------------------------------------------------------
// SSAO (ssao_shader.frag:20-56), blur/apply (ssao_apply_shader.frag:29-75), tone map
// ---------------------------------------------------------------------------------------------
__constant__ float c_ssao_noise[48];
__constant__ float c_ssao_kernel[192];

struct __attribute__((packed, aligned(4))) ZPair { float a, b; };   // two neighbouring texels: one 8-byte load, 4-byte aligned

// `img`: the scene's padded camera-z plane (k_shade), pitch W + 2, texel (x, y) at [y + 1][x + 1]; `xs`, `ys`: window
// coordinates already shifted to texel centres (oracle rect_bilinear_s)
__device__ __forceinline__ float rect_bilinear_z(const float* __restrict__ img, int W, int H, float xs, float ys)
{
    const float fx = floorf(xs), fy = floorf(ys);
    const float ax = xs - fx, ay = ys - fy;
    // the rect sampler's texels clamp(int(f), 0, n-1) and clamp(int(f) + 1, 0, n-1) are the padded plane's columns
    // c and c + 1 with c = clamp(int(f), -1, n-1) + 1; clamp(int(f), lo, hi) == int(med3(f, lo, hi)) for every non-NaN f
    // (integers below 2^24 are exact in fp32, and so is row * pitch + column: one v_med3_f32 per coordinate, one fma, one
    // conversion).  The plane (4 B/px instead of the 16 B/px of the camCoordinates target) keeps the 64-tap gather L2 resident.
    const float xc = __builtin_amdgcn_fmed3f(fx, -1.0f, (float)(W - 1));
    const float yc = __builtin_amdgcn_fmed3f(fy, -1.0f, (float)(H - 1));
    const int pitch = W + 2;
    const int idx = (int)fmaf(yc, (float)pitch, xc);                  // (row - 1) * pitch + (column - 1), may be negative
    // byte offsets in 32 bits from wave-uniform bases: ONE VGPR offset (a shift-and-add of the index) serves both rows, the
    // lower row's base is the upper one's plus a pitch, in scalar registers
    unsigned c0 = (unsigned)(pitch + 1) * 4u;
    asm("" : "+s"(c0));   // opaque scalar: (idx << 2) + c0 is ONE v_lshl_add_u32 (the compiler otherwise splits the constant: add, shift-add)
    const unsigned off = ((unsigned)idx << 2) + c0;
    const char* base = reinterpret_cast<const char*>(img);
    const char* base_bot = base + (size_t)pitch * 4u;
    const ZPair top2 = *reinterpret_cast<const ZPair*>(base + off);
    const ZPair bot2 = *reinterpret_cast<const ZPair*>(base_bot + off);
    const float top = fmaf(ax, top2.b - top2.a, top2.a), bot = fmaf(ax, bot2.b - bot2.a, bot2.a);
    return fmaf(ay, bot - top, top);
}

// oracle ssao_rcp: three Newton steps on |x| from the integer-subtraction seed, sign restored
__device__ __forceinline__ float ssao_rcp(float x)
{
    const float a = fabsf(x);
    float r = __uint_as_float(0x7EF311C7u - __float_as_uint(a));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = fmaf(-a, r, 1.0f);
        r = fmaf(r, e, r);
    }
    return __uint_as_float((__float_as_uint(r) & 0x7fffffffu) | (__float_as_uint(x) & 0x80000000u));
}

// occlusion of one pixel with geometry (normal n4, camera position f4): ssao_shader.frag:20-56 as the oracle's ssao_pass spells it
__device__ __forceinline__ float ssao_pixel(const float* __restrict__ proj, const float* __restrict__ camS, int W, int H, int i, int j,
                                            float4 n4, float4 f4, const float* __restrict__ kern)
{
    float n[3] = {n4.x, n4.y, n4.z};
    normalize3(n);
    const float frag[3] = {f4.x, f4.y, f4.z};
    const float* rv0 = &c_ssao_noise[3 * ((j & 3) * 4 + (i & 3))];
    float rv[3] = {rv0[0], rv0[1], rv0[2]};
    normalize3(rv);
    const float d = dot3(rv, n);
    float tg[3] = {rv[0] - n[0] * d, rv[1] - n[1] * d, rv[2] - n[2] * d};
    normalize3(tg);
    const float bt[3] = {n[1] * tg[2] - n[2] * tg[1], n[2] * tg[0] - n[0] * tg[2], n[0] * tg[1] - n[1] * tg[0]};
    const float radius = 0.1f, bias = 0.0025f;
    // The sample position and its projection are linear in the kernel vector: per-pixel bases, then
    // four 3-fma chains per tap (rows 0, 1, 3 of the projection + the sample's z); one reciprocal
    // serves both perspective divisions -- the oracle's ssao_pass spells out the same arithmetic.
    float tgR[3], btR[3], nR[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { tgR[c] = tg[c] * radius; btR[c] = bt[c] * radius; nR[c] = n[c] * radius; }
    const float fr4[4] = {frag[0], frag[1], frag[2], 1.0f};
    float A[4];
    mv4(proj, fr4, A);
    const float B0 = dot3(proj, tgR), C0 = dot3(proj, btR), D0 = dot3(proj, nR);
    const float B1 = dot3(proj + 4, tgR), C1 = dot3(proj + 4, btR), D1 = dot3(proj + 4, nR);
    const float B3 = dot3(proj + 12, tgR), C3 = dot3(proj + 12, btR), D3 = dot3(proj + 12, nR);
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H, hwm = hw - 0.5f, hhm = hh - 0.5f;
    // a perspective projection's last row is (0, 0, 1, 0): the clip-space w of a sample IS its camera z, bit for bit
    // (0 * a + 0 * b + 1 * z + 0 and the products with exact zeros add nothing), and one of the four chains goes
    const bool w_is_z = proj[12] == 0.0f && proj[13] == 0.0f && proj[14] == 1.0f && proj[15] == 0.0f;
    float occlusion = 0.0f;
    auto taps = [&](auto w_is_z_c) {
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
        const float s0 = kern[3 * k], s1 = kern[3 * k + 1], s2 = kern[3 * k + 2];
        const float spz = fmaf(nR[2], s2, fmaf(btR[2], s1, fmaf(tgR[2], s0, frag[2])));
        const float o0 = fmaf(D0, s2, fmaf(C0, s1, fmaf(B0, s0, A[0])));
        const float o1 = fmaf(D1, s2, fmaf(C1, s1, fmaf(B1, s0, A[1])));
        const float o3 = decltype(w_is_z_c)::value ? spz : fmaf(D3, s2, fmaf(C3, s1, fmaf(B3, s0, A[3])));
        const float rw = ssao_rcp(o3);
        const float sd = rect_bilinear_z(camS, W, H, fmaf(o0 * rw, hw, hwm), fmaf(o1 * rw, hh, hhm));
        // range check smoothstep(clamp(radius / |dz|)): exactly 1 whenever |dz| <= radius, and
        // irrelevant for taps that do not occlude -- the division runs only where it matters
        // (on open surfaces whole waves skip it)
        const bool occludes = sd <= spz - bias;
        const float adz = fabsf(frag[2] - sd);
        float add = occludes ? 1.0f : 0.0f;
        if (occludes && adz > radius) {
            const float tt = clampf(radius / adz, 0.0f, 1.0f);
            add = tt * tt * (3.0f - 2.0f * tt);
        }
        occlusion += add;
    }
    };
    if (w_is_z) taps(std::true_type{});   // scene-uniform: the whole wave takes one side
    else taps(std::false_type{});
    return 1.0f - occlusion / 64.0f;
}

__global__ __launch_bounds__(256) SLHIP_SSAO_KERNEL void k_ssao(const slhip_scene* __restrict__ scenes, unsigned n_scenes, int W, int H,
                                              const float* __restrict__ cam, const float* __restrict__ nrm,
                                              const float* __restrict__ zplane, float* __restrict__ ao,
                                              const float* __restrict__ kern)
{
    const size_t P = (size_t)W * H;
    const unsigned blocks_per_scene = (unsigned)((P + 255) / 256);
    unsigned scene, blk;
    if (!scene_block(blocks_per_scene, n_scenes, scene, blk)) return;
    // A wave takes the pixels of ONE column class (x & 3) of the block's 256-pixel strip: the 4x4 noise tile rotates the
    // sample kernel per class, so the 64 taps of a sample then move together and land in one strip of the z plane per row
    // (8-9 cache lines per load instead of the ~12 of four interleaved, differently shifted classes).  Placement only.
    const unsigned pix = blk * 256 + (threadIdx.x >> 6) + 4u * (threadIdx.x & 63u);
    if (pix >= P) return;
    const float* proj = scenes[scene].proj;
    const float* camS = zplane + (size_t)scene * ((size_t)(W + 2) * (H + 2));
    const size_t gp = (size_t)scene * P + pix;
    const int i = (int)(pix % (unsigned)W), j = (int)(pix / (unsigned)W);
    const float4 n4 = reinterpret_cast<const float4*>(nrm)[gp];
    if (n4.x == 0.0f && n4.y == 0.0f && n4.z == 0.0f) { ao[gp] = 1.0f; return; }
    ao[gp] = ssao_pixel(proj, camS, W, H, i, j, n4, reinterpret_cast<const float4*>(cam)[gp], kern);
}

// ---- SSAO only where something can occlude ----------------------------------------------------------------------------------
// On the open background plane no tap occludes: the 64 samples lie in the hemisphere over the plane (kernel z >= 0, |sample| <=
// radius), so along the ray through a sample the plane is never nearer than the sample, and `sd <= spz - bias` fails by the bias
// (2.5 mm against the micrometres of the bilinear fetch) -- AS LONG AS every texel a tap can reach belongs to that plane or to
// the cleared background (z = 3000) and no tap leaves the image (a clamped fetch returns the plane's depth somewhere else).  Then
// occlusion = 0 and ao = 1 - 0 / 64 = 1 exactly, which is what the full loop computes.  k_shade files, per 8 x 8 tile, the nearest
// camera z and whether the tile holds anything but that plane; k_ssao_mask marks the tiles whose reach -- a sample is at most
// `radius` from its pixel's position P = (X, Y, Z), so it projects within fx radius sqrt(1 + (X/Z)^2) / (Z - radius) pixels of it
// (Cauchy-Schwarz on (dx Z - X dz) / (Z (Z + dz))), + 2 for the bilinear footprint and rounding -- stays inside the image and
// touches plane / background tiles only (a summed-area table of the "other" flags in LDS); k_ssao_tiled writes 1 there and runs
// the taps for the rest, compacted: a wave walks the pixels of ONE noise class (x & 3, y & 3) of a 16-row band in raster order
// and queues those that need the loop, 64 at a time (the taps of a wave still move together).
constexpr int kSsaoTile = 8;
__global__ __launch_bounds__(256) void k_ssao_mask(const slhip_scene* __restrict__ scenes, int W, int H,
                                                  const float2* __restrict__ tiles, unsigned char* __restrict__ skip, int dbg)
{
    extern __shared__ int s_sat[];              // (TH + 1) x (TW + 1)
    const int TW = W / kSsaoTile, TH = H / kSsaoTile, SW = TW + 1;
    const unsigned scene = blockIdx.x;
    const float2* T = tiles + (size_t)scene * (TW * TH);
    unsigned char* S = skip + (size_t)scene * (TW * TH);
    const float* proj = scenes[scene].proj;
    // the rule is derived for a plain perspective projection (no skew, w = camera z): anything else runs every pixel
    const bool persp = proj[1] == 0.0f && proj[3] == 0.0f && proj[4] == 0.0f && proj[7] == 0.0f &&
                       proj[12] == 0.0f && proj[13] == 0.0f && proj[14] == 1.0f && proj[15] == 0.0f && proj[0] > 0.0f && proj[5] > 0.0f;
    for (int k = threadIdx.x; k < (TH + 1) * SW; k += 256) {
        const int r = k / SW, c = k % SW;
        s_sat[k] = (r > 0 && c > 0) ? (int)(__float_as_uint(T[(r - 1) * TW + (c - 1)].y) & 1u) : 0;
    }
    __syncthreads();
    for (int r = threadIdx.x + 1; r <= TH; r += 256) { int acc = 0; for (int c = 1; c <= TW; ++c) { acc += s_sat[r * SW + c]; s_sat[r * SW + c] = acc; } }
    __syncthreads();
    for (int c = threadIdx.x + 1; c <= TW; c += 256) { int acc = 0; for (int r = 1; r <= TH; ++r) { acc += s_sat[r * SW + c]; s_sat[r * SW + c] = acc; } }
    __syncthreads();
    const float fx = proj[0] * 0.5f * (float)W, fy = proj[5] * 0.5f * (float)H;
    const float tx = (1.0f + fabsf(proj[2])) / proj[0], ty = (1.0f + fabsf(proj[6])) / proj[5];   // largest |X / Z|, |Y / Z| in the image
    const float tmax = fmaxf(tx, ty);
    const float reach = fmaxf(fx, fy) * 0.1f * sqrtf(1.0f + tmax * tmax) * 1.001f;
    for (int k = threadIdx.x; k < TW * TH; k += 256) {
        const float2 t = T[k];
        const unsigned fl = __float_as_uint(t.y);
        unsigned char sk = 0;
        if (!(fl & 2u)) sk = 1;                                   // no geometry at all: every pixel is 1 anyway
        else if (persp && !(fl & 1u) && t.x > 0.2f) {
            const int R = (int)ceilf(reach / (t.x - 0.1f)) + 2;
            const int x0 = (k % TW) * kSsaoTile - R, x1 = (k % TW) * kSsaoTile + kSsaoTile - 1 + R;
            const int y0 = (k / TW) * kSsaoTile - R, y1 = (k / TW) * kSsaoTile + kSsaoTile - 1 + R;
            if (x0 >= 0 && y0 >= 0 && x1 < W && y1 < H) {
                const int ca = x0 / kSsaoTile, cb = x1 / kSsaoTile + 1, ra = y0 / kSsaoTile, rb = y1 / kSsaoTile + 1;
                const int others = s_sat[rb * SW + cb] - s_sat[ra * SW + cb] - s_sat[rb * SW + ca] + s_sat[ra * SW + ca];
                sk = others == 0 ? 1 : 0;
            }
        }
        S[k] = dbg == 1 ? 1 : dbg == 2 ? 0 : sk;
    }
}

// grid: scenes x (H / 16) bands x 4 row classes; block = 4 waves = the 4 column classes of the same rows (their normal / position
// fetches share cache lines)
__global__ __launch_bounds__(256) SLHIP_SSAO_KERNEL void k_ssao_tiled(const slhip_scene* __restrict__ scenes, unsigned n_scenes, int W, int H,
                                                    const float* __restrict__ cam, const float* __restrict__ nrm,
                                                    const float* __restrict__ zplane, float* __restrict__ ao,
                                                    const float* __restrict__ kern, const unsigned char* __restrict__ skip)
{
    __shared__ unsigned s_q[4][128];
    __shared__ float4 s_n[4][128];              // the queued pixels' normals (read once, in the scan)
    const unsigned bands = (unsigned)H >> 4;
    unsigned scene, rest;
    if (!scene_block(bands * 4u, n_scenes, scene, rest)) return;       // (all blocks of a scene on one XCD: its z plane in ONE L2)
    const unsigned band = rest >> 2, yc = rest & 3u, xc = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t P = (size_t)W * H;
    const float* proj = scenes[scene].proj;
    const float* camS = zplane + (size_t)scene * ((size_t)(W + 2) * (H + 2));
    const int TW = W / kSsaoTile;
    const unsigned char* S = skip + (size_t)scene * ((size_t)TW * (H / kSsaoTile));
    const float4* N4 = reinterpret_cast<const float4*>(nrm) + (size_t)scene * P;
    const float4* C4 = reinterpret_cast<const float4*>(cam) + (size_t)scene * P;
    float* A = ao + (size_t)scene * P;
    unsigned* q = s_q[xc];
    float4* qn = s_n[xc];
    const unsigned per_row = (unsigned)W >> 2, total = 4u * per_row;       // the class's pixels in the band: 4 rows x W / 4
    unsigned count = 0;
    auto run = [&](unsigned n_run) {
        // the first n_run (<= 64) queued pixels
        if (lane < n_run) {
            const unsigned pix = q[lane];
            const int i = (int)(pix % (unsigned)W), j = (int)(pix / (unsigned)W);
            A[pix] = ssao_pixel(proj, camS, W, H, i, j, qn[lane], C4[pix], kern);
        }
    };
    // one chunk of candidates ahead: the loads of the next scan are in flight while the taps of the queue run
    auto pixel_of = [&](unsigned ci) -> unsigned { return (band * 16u + yc + 4u * (ci / per_row)) * (unsigned)W + xc + 4u * (ci % per_row); };
    auto fetch = [&](unsigned ci, unsigned& pix, bool& sk, float4& n4) {
        pix = 0; sk = true; n4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ci < total) {
            pix = pixel_of(ci);
            const unsigned y = pix / (unsigned)W, x = pix % (unsigned)W;
            sk = S[(y >> 3) * (unsigned)TW + (x >> 3)] != 0;
            if (!sk) n4 = N4[pix];
        }
    };
    unsigned pix_n; bool sk_n; float4 n4_n;
    fetch(lane, pix_n, sk_n, n4_n);
    for (unsigned c0 = 0; c0 < total; c0 += 64u) {
        const unsigned pix = pix_n;
        const bool sk = sk_n;
        const float4 n4 = n4_n;
        const bool valid = c0 + lane < total;
        fetch(c0 + 64u + lane, pix_n, sk_n, n4_n);
        const bool need = valid && !sk && !(n4.x == 0.0f && n4.y == 0.0f && n4.z == 0.0f);
        if (valid && !need) A[pix] = 1.0f;
        const unsigned long long m = __ballot(need);
        if (need) {
            const unsigned at = count + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            q[at] = pix; qn[at] = n4;
        }
        count += (unsigned)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (count >= 64u) {
            run(64u);
            __builtin_amdgcn_wave_barrier();
            const unsigned left = count - 64u;
            unsigned mv = 0;
            float4 mn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (lane < left) { mv = q[64u + lane]; mn = qn[64u + lane]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < left) { q[lane] = mv; qn[lane] = mn; }
            count = left;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (count > 0u) run(count);
}

// ... and the tone map of the result in the same pass (tone_map_shader.frag after ssao_apply_shader.frag): the blurred-AO colour
// goes straight into tone_map_px, the float image is stored only on request (SLHIP_RENDER_KEEP_HDR).  `lum_scene`: the scene's
// exposure luminance from k_lum_reduce (auto exposure), one float per scene at a stride of 4 * blocks_per_scene.
__global__ __launch_bounds__(256) SLHIP_LIGHT_KERNEL void k_ssao_apply(const slhip_scene* __restrict__ scenes, unsigned n_scenes, int W, int H,
                                                    const float* __restrict__ hdr_in, const float* __restrict__ ao,
                                                    const float* __restrict__ zplane, float* __restrict__ hdr_out,
                                                    const float* __restrict__ lum_scene, uint8_t* __restrict__ rgb,
                                                    const unsigned char* __restrict__ skip)
{
    const size_t P = (size_t)W * H;
    const unsigned blocks_per_scene = (unsigned)((P + 255) / 256);
    unsigned scene, blk;
    if (!scene_block(blocks_per_scene, n_scenes, scene, blk)) return;
    const unsigned pix = blk * 256 + threadIdx.x;
    if (pix >= P) return;
    const float* camS = zplane + (size_t)scene * ((size_t)(W + 2) * (H + 2));   // padded: texel (x, y) at [y + 1][x + 1]
    const float* aoS = ao + (size_t)scene * P;
    const size_t gp = (size_t)scene * P + pix;
    const int i = (int)(pix % (unsigned)W), j = (int)(pix / (unsigned)W);
    if (skip) {
        // every occlusion value the blur of this pixel reads (i-2..i+1, j-2..j+1, clamped) is 1 when the tiles of the footprint's
        // corners were skipped by the SSAO pass (k_ssao_mask): the weighted mean of ones is 1 -- the colour goes through as it is
        const int TW = W / kSsaoTile;
        const unsigned char* S = skip + (size_t)scene * ((size_t)TW * (H / kSsaoTile));
        const int xa = max(i - 2, 0) >> 3, xb = min(i + 1, W - 1) >> 3, ya = max(j - 2, 0) >> 3, yb = min(j + 1, H - 1) >> 3;
        const bool ones = (S[ya * TW + xa] & S[ya * TW + xb] & S[yb * TW + xa] & S[yb * TW + xb]) != 0;
        if (__all(ones)) {
            const float4 h = reinterpret_cast<const float4*>(hdr_in)[gp];
            const float c4[4] = {h.x, h.y, h.z, h.w};
            if (hdr_out) reinterpret_cast<float4*>(hdr_out)[gp] = h;
            if (rgb) {
                const float manual = scenes[scene].manual_exposure;
                const float lum = (manual >= 0.0f) ? 0.0f : lum_scene[(size_t)scene * blocks_per_scene * 4];
                reinterpret_cast<uchar4*>(rgb)[gp] = tone_map_px(c4, manual, lum);
            }
            return;
        }
    }
    const float sigma = 3.0f * 0.5f;
    const float falloff = 1.0f / (2.0f * sigma * sigma);
    // The 16 taps (and the centre) sample the z plane at integer coordinates of a LINEAR rect
    // sampler, i.e. at the corner between 4 texels with weights exactly 1/2: together they touch
    // the 5x5 texels around (i-3..i+1, j-3..j+1) (clamped).  Fetch them once; every tap is then
    // rect_bilinear_z()'s arithmetic on registers.
    unsigned rowo[5];
    int colx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        colx[k] = min(max(i - 3 + k, 0), W - 1) + 1;
        rowo[k] = (unsigned)(min(max(j - 3 + k, 0), H - 1) + 1) * (unsigned)(W + 2);
    }
    float z[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) z[r][c] = camS[rowo[r] + (unsigned)colx[c]];
    // corner value at (i + x, j + y), x, y in -2..1: texels (i+x-1, i+x) x (j+y-1, j+y) = window [y+2..y+3][x+2..x+3]
    auto corner = [&](int x, int y) {
        const float a = z[y + 2][x + 2], b = z[y + 2][x + 3], c = z[y + 3][x + 2], d = z[y + 3][x + 3];
        const float top = fmaf(0.5f, b - a, a), bot = fmaf(0.5f, d - c, c);
        return fmaf(0.5f, bot - top, top);
    };
    const float cd = corner(0, 0);
    float result = 0.0f, wt = 0.0f;
#pragma unroll
    for (int x = -2; x < 2; ++x)
#pragma unroll
        for (int y = -2; y < 2; ++y) {
            const int xi = min(max(i + x, 0), W - 1), yj = min(max(j + y, 0), H - 1);
            const float c = aoS[(unsigned)yj * (unsigned)W + (unsigned)xi];
            const float dz = corner(x, y);
            const float r = sqrtf((float)(x * x + y * y));
            const float dd = (dz - cd) * 300.0f;
            const float w = __builtin_amdgcn_exp2f(-r * r * falloff - dd * dd);   // (a weight that underflows adds nothing to sums that hold the centre's 1)
            wt += w;
            result += c * w;
        }
    const float a = result * frcp(wt);
    const float4 h = reinterpret_cast<const float4*>(hdr_in)[gp];
    const float c4[4] = {h.x * a, h.y * a, h.z * a, h.w};
    if (hdr_out) reinterpret_cast<float4*>(hdr_out)[gp] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    if (rgb) {
        const float manual = scenes[scene].manual_exposure;
        const float lum = (manual >= 0.0f) ? 0.0f : lum_scene[(size_t)scene * blocks_per_scene * 4];
        reinterpret_cast<uchar4*>(rgb)[gp] = tone_map_px(c4, manual, lum);
    }
}

// Auto exposure (tone_map_shader.frag: the 1x1 mip of the HDR target): ONE block per scene adds the per-block sums k_shade left, in
// a fixed order, and leaves the exposure luminance in the first float of the scene's partials (they are spent by then).  Every
// block of the tone map used to repeat this reduction for itself: 1200 blocks x 19 KB of L2 reads per scene.
__global__ __launch_bounds__(256) void k_lum_reduce(const slhip_scene* __restrict__ scenes, unsigned blocks_per_scene, size_t P,
                                                    float* __restrict__ lum_part)
{
    const unsigned scene = blockIdx.x;
    if (scenes[scene].manual_exposure >= 0.0f) return;
    __shared__ float red[4][256];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float* part = lum_part + (size_t)scene * blocks_per_scene * 4;
    for (unsigned b = threadIdx.x; b < blocks_per_scene; b += 256)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += part[4 * b + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
#pragma unroll
            for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float inv = 1.0f / (float)P;
        const float a0 = red[0][0] * inv, a1 = red[1][0] * inv, a2 = red[2][0] * inv, a3 = red[3][0] * inv;
        part[0] = 0.1f * (0.2125f * (a0 / a3) + 0.7154f * (a1 / a3) + 0.0721f * (a2 / a3));
    }
}

// tone map of the shaded image when no SSAO pass precedes it (with SSAO: k_ssao_apply)
__global__ __launch_bounds__(256) void k_tonemap(const slhip_scene* __restrict__ scenes, unsigned n_scenes, int W, int H,
                                                 const float* __restrict__ hdr, const float* __restrict__ lum_scene,
                                                 uint8_t* __restrict__ rgb)
{
    const size_t P = (size_t)W * H;
    const unsigned blocks_per_scene = (unsigned)((P + 255) / 256);
    unsigned scene, blk;
    if (!scene_block(blocks_per_scene, n_scenes, scene, blk)) return;
    const unsigned pix = blk * 256 + threadIdx.x;
    if (pix >= P) return;
    const float manual = scenes[scene].manual_exposure;
    const float lum = (manual >= 0.0f) ? 0.0f : lum_scene[(size_t)scene * blocks_per_scene * 4];
    const size_t gp = (size_t)scene * P + pix;
    const float4 h = reinterpret_cast<const float4*>(hdr)[gp];
    const float c[4] = {h.x, h.y, h.z, h.w};
    reinterpret_cast<uchar4*>(rgb)[gp] = tone_map_px(c, manual, lum);
}

// clears the shadow maps of the ACTIVE lights only (blockIdx.y = scene * NUM_LIGHTS + light)
__global__ __launch_bounds__(256) void k_clear_shadow(unsigned* __restrict__ shadow, size_t n4)
{
    uint4* p = reinterpret_cast<uint4*>(shadow);
    const uint4 v = make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = v;
}

bool g_ssao_tables_uploaded[16] = {};

// optional per-phase HIP-event timing (bench.py's roofline leg).  Every timed slhip_render call records into its OWN set of
// events, and nothing waits for them until slhip_render_timings() asks: the host must be free to enqueue the next calls (and the
// next step's settle) while this one runs -- a readback of the previous call's events at the start of every call kept the host
// in lockstep with the render stream, so that settle and render of consecutive batches never overlapped.
constexpr int kNumPhases = 8;
bool g_timing = false;
bool g_ev_created = false;      // timing has been enabled at least once
struct TimedCall { hipEvent_t ev[kNumPhases + 1]; bool recorded[kNumPhases + 1]; };
std::vector<TimedCall> g_timed;
TimedCall* g_cur = nullptr;     // the call being enqueued
double g_acc_ms[kNumPhases] = {};

inline void mark(int i, hipStream_t stream)
{
    if (!g_timing || !g_cur) return;
    if (hipEventCreate(&g_cur->ev[i]) != hipSuccess) return;
    (void)hipEventRecord(g_cur->ev[i], stream);
    g_cur->recorded[i] = true;
}

}  // namespace

// phases: 0 shadow clear+raster, 1 shadow large, 2 vis clear + raster, 3 large, 4 shade, 5 ssao,
//         6 ssao apply, 7 tone map
extern "C" int slhip_timing_enable(int on)
{
    if (on) g_ev_created = true;
    g_timing = on != 0;
    return 0;
}

// waits for the recorded calls, adds their phase durations to the accumulators, frees their events
static int flush_timings()
{
    int status = 0;
    for (TimedCall& c : g_timed) {
        int prev = -1;
        for (int i = 0; i <= kNumPhases; ++i) {
            if (!c.recorded[i]) continue;
            if (status == 0 && hipEventSynchronize(c.ev[i]) != hipSuccess) status = -1;
            if (status == 0 && prev >= 0) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, c.ev[prev], c.ev[i]) == hipSuccess) g_acc_ms[prev] += ms;
            }
            prev = i;
        }
        for (int i = 0; i <= kNumPhases; ++i)
            if (c.recorded[i]) (void)hipEventDestroy(c.ev[i]);
    }
    g_timed.clear();
    g_cur = nullptr;
    if (status != 0) slhip::set_error("slhip_render_timings: event readback failed");
    return status;
}

// returns the per-phase totals accumulated over all slhip_render calls since the last query
extern "C" int slhip_render_timings(float* ms_out)
{
    if (!g_ev_created) { slhip::set_error("timing not enabled"); return -1; }
    const int st = flush_timings();
    if (st != 0) return st;
    for (int i = 0; i < kNumPhases; ++i) { ms_out[i] = (float)g_acc_ms[i]; g_acc_ms[i] = 0.0; }
    return 0;
}

// SSAO tables (generated at build time by tools/gen_ssao_tables.py from the recipe of
// ssao_shader.cpp:72-112)
#include "ssao_tables.inc"

// layout of d_ao: [B][H][W] occlusion, [B][H + 2][W + 2] camera z, then (8-byte aligned) the SSAO tile records and skip bytes
static inline uint64_t ssao_tiles_per_scene(uint32_t w, uint32_t h) { return (uint64_t)((w + 7) / 8) * ((h + 7) / 8); }
static inline uint64_t ssao_tiles_offset(uint64_t B, uint32_t w, uint32_t h)
{
    const uint64_t b = B * (uint64_t)w * h * 4 + B * (uint64_t)(w + 2) * (h + 2) * 4;
    return (b + 7u) & ~(uint64_t)7u;
}

extern "C" int slhip_render_scratch_bytes(uint32_t n_scenes, uint32_t width, uint32_t height,
                                          uint32_t shadow_res, uint32_t queue_capacity, uint64_t bytes_out[7])
{
    const uint64_t P = (uint64_t)width * height, B = n_scenes;
    const uint64_t blocks = (P + 255) / 256;
    bytes_out[0] = B * P * 8;                                              // d_vis
    bytes_out[1] = 2 * B * P * 16;                                         // d_hdr (two planes)
    // d_ao + padded camera-z plane + (k_ssao_mask) per 8 x 8 tile: float2 record, skip byte
    bytes_out[2] = ssao_tiles_offset(B, width, height) + B * ssao_tiles_per_scene(width, height) * 9 + 16;
    bytes_out[3] = B * SLHIP_NUM_LIGHTS * (uint64_t)shadow_res * shadow_res * 4;  // d_shadow
    bytes_out[4] = 16 + (uint64_t)queue_capacity * 16;                     // d_queue
    bytes_out[5] = B * blocks * 16;                                        // d_lum
    bytes_out[6] = B * SLHIP_NUM_LIGHTS * (uint64_t)shadow_tile_words((int)shadow_res) * 4;   // d_shadow_tiles
    return 0;
}

// How much of the last SSAO pass on this scratch was skipped (k_ssao_mask): 8 x 8 tiles in all, tiles whose occlusion is 1 without
// running the taps.  Synchronises `stream`.  counts = {0, 0} when the pass did not run tiled (viewport not a multiple of 32 x 16).
extern "C" int slhip_render_ssao_skipped(const slhip_render_scratch* scratch, uint32_t n_scenes, uint32_t width, uint32_t height,
                                         uint64_t counts[2], void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!scratch || !scratch->d_ao || !counts) {
        slhip::set_error("slhip_render_ssao_skipped: null argument");
        return -1;
    }
    counts[0] = counts[1] = 0;
    if (width % 32 != 0 || height % 16 != 0 || n_scenes == 0) return 0;
    const uint64_t nt = (uint64_t)n_scenes * ssao_tiles_per_scene(width, height);
    std::vector<unsigned char> h(nt);
    const char* base = reinterpret_cast<const char*>(scratch->d_ao) + ssao_tiles_offset(n_scenes, width, height) + nt * 8;
    SLHIP_CHECK(hipMemcpyAsync(h.data(), base, nt, hipMemcpyDeviceToHost, stream));
    SLHIP_CHECK(hipStreamSynchronize(stream));
    counts[0] = nt;
    for (unsigned char b : h) counts[1] += b ? 1 : 0;
    return 0;
}

extern "C" int slhip_render(const slhip_mesh_pool* pool, const slhip_scene* d_scenes, const slhip_draw* d_draws,
                            const slhip_chunk* d_chunks, uint32_t n_scenes, uint32_t n_draws, uint32_t n_chunks,
                            uint32_t width,
                            uint32_t height, uint32_t flags, const float* d_depth_peel,
                            const slhip_render_out* out, const slhip_render_scratch* scratch, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pool || !d_scenes || !d_draws || !out || !scratch) {
        slhip::set_error("slhip_render: null argument");
        return -1;
    }
    if (n_scenes == 0 || width == 0 || height == 0) return 0;
    if (!scratch->d_vis || !scratch->d_queue) {
        slhip::set_error("slhip_render: d_vis and d_queue scratch are required");
        return -1;
    }
    const bool want_rgb = out->d_rgb != nullptr;
    const bool ssao = (flags & SLHIP_RENDER_SSAO) && want_rgb;
    const bool shadows = (flags & SLHIP_RENDER_SHADOWS) && want_rgb && scratch->d_shadow != nullptr;
    if (shadows && (!scratch->d_shadow_tiles || (scratch->shadow_res & 3u))) {
        slhip::set_error("slhip_render: shadows need the d_shadow_tiles scratch and a shadow_res that is a multiple of 4");
        return -1;
    }
    if (ssao && (!out->d_cam_coord || !out->d_normals || !scratch->d_ao)) {
        slhip::set_error("slhip_render: SSAO needs the cam_coord and normals outputs and d_ao scratch");
        return -1;
    }
    if (want_rgb && (!scratch->d_hdr || !scratch->d_lum)) {
        slhip::set_error("slhip_render: rgb output needs d_hdr and d_lum scratch");
        return -1;
    }
    const int W = (int)width, H = (int)height;
    const size_t P = (size_t)W * H;
    const unsigned blocks_per_scene = (unsigned)((P + 255) / 256);
    const unsigned pix_blocks = blocks_per_scene * 8u * ((n_scenes + 7u) / 8u);   // see scene_block()
    const int S = (int)scratch->shadow_res;
    // light maps per scene in d_shadow: 0 = all SLHIP_NUM_LIGHTS; a caller whose scenes only use the first lights saves
    // 4 S^2 bytes per scene and unused light (lights beyond the count cast no shadow)
    // bounding boxes up to this many pixels are walked by the triangle's own thread, larger ones go to the tile queue
    // (developer knobs; on the C2 scenes 48 / 16 instead of 128 cost the shadow pass +16 % / +100 %)
    static const int raster_small = getenv("SLHIP_RASTER_SMALL") ? atoi(getenv("SLHIP_RASTER_SMALL")) : kSmallArea;
    static const int shadow_small = getenv("SLHIP_SHADOW_SMALL") ? atoi(getenv("SLHIP_SHADOW_SMALL")) : kSmallArea;
    const int NL = scratch->shadow_lights == 0u ? SLHIP_NUM_LIGHTS : (int)min(scratch->shadow_lights, (uint32_t)SLHIP_NUM_LIGHTS);
    if (n_chunks > 0 && (!scratch->d_clip || scratch->n_clip_verts == 0)) {
        slhip::set_error("slhip_render: d_clip scratch (n_clip_verts x 16 B x planes) is required");
        return -1;
    }
    if (n_chunks > 0 && !scratch->d_vattr) {
        slhip::set_error("slhip_render: d_vattr scratch (n_clip_verts x 80 B) is required");
        return -1;
    }
    const float4* clipbuf = reinterpret_cast<const float4*>(scratch->d_clip);
    // post-transform vertex cache: 64-byte records, then the dense plane of window coordinates
    float4* vattr = reinterpret_cast<float4*>(scratch->d_vattr);
    const uint4* screen = reinterpret_cast<const uint4*>(vattr) + 4 * (size_t)scratch->n_clip_verts;

    if (ssao) {
        int dev = 0;
        SLHIP_CHECK(hipGetDevice(&dev));
        if (dev < 16 && !g_ssao_tables_uploaded[dev]) {
            SLHIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ssao_noise), k_ssao_noise_host, sizeof(float) * 48, 0,
                                               hipMemcpyHostToDevice, stream));
            SLHIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ssao_kernel), k_ssao_kernel_host, sizeof(float) * 192, 0,
                                               hipMemcpyHostToDevice, stream));
            g_ssao_tables_uploaded[dev] = true;
        }
    }

    if (g_timing) {
        // a caller that enables timing and never reads it must not grow the list for ever: beyond kMaxTimedCalls the oldest
        // records are dropped (their events destroyed, their durations lost)
        constexpr size_t kMaxTimedCalls = 4096;
        if (g_timed.size() >= kMaxTimedCalls) {
            const size_t drop = g_timed.size() / 2;
            for (size_t k = 0; k < drop; ++k)
                for (int i = 0; i <= kNumPhases; ++i)
                    if (g_timed[k].recorded[i]) (void)hipEventDestroy(g_timed[k].ev[i]);
            g_timed.erase(g_timed.begin(), g_timed.begin() + (long)drop);
        }
        g_timed.emplace_back();
        g_cur = &g_timed.back();
        for (int i = 0; i <= kNumPhases; ++i) g_cur->recorded[i] = false;
    }
    // vertex transform on the matrix cores: camera clip + (if shadows) the three light clips
    if (n_chunks > 0) {
        k_vertex_xform<<<dim3(n_draws, 32), 256, 0, stream>>>(*pool, d_scenes, d_draws, reinterpret_cast<float4*>(scratch->d_clip),
                                                                      scratch->n_clip_verts, vattr, W, H, shadows ? 1 : 0, S);
        SLHIP_LAUNCH_CHECK();
    }
    // shadow pass (maps clean on entry, see k_shadow_raster)
    const unsigned n_tile_words = shadows ? n_scenes * SLHIP_NUM_LIGHTS * (unsigned)shadow_tile_words(S) : 0u;
    if (shadows && (flags & SLHIP_RENDER_SHADOW_RESET)) {
        k_clear_shadow<<<4096, 256, 0, stream>>>(reinterpret_cast<unsigned*>(scratch->d_shadow),
                                                 (size_t)n_scenes * NL * S * S / 4);
        SLHIP_CHECK(hipMemsetAsync(scratch->d_shadow_tiles, 0, (size_t)n_tile_words * 4, stream));
    }
    if (shadows && n_chunks > 0) {
        mark(0, stream);
        SLHIP_CHECK(hipMemsetAsync(scratch->d_queue, 0, 16, stream));
        k_shadow_raster<<<n_chunks, 256, 0, stream>>>(
            *pool, d_scenes, d_draws, d_chunks, S, reinterpret_cast<unsigned*>(scratch->d_shadow),
            scratch->d_queue, scratch->queue_capacity, clipbuf, scratch->n_clip_verts, scratch->d_shadow_tiles, NL, shadow_small);
        mark(1, stream);
        k_shadow_large<<<2048, 256, 0, stream>>>(*pool, d_scenes, d_draws, S,
                                                 reinterpret_cast<unsigned*>(scratch->d_shadow), scratch->d_queue,
                                                 scratch->queue_capacity, clipbuf, scratch->n_clip_verts, NL);
        SLHIP_LAUNCH_CHECK();
    }

    // main pass: visibility
    mark(2, stream);
    SLHIP_CHECK(hipMemsetAsync(scratch->d_vis, 0xFF, (size_t)n_scenes * P * 8, stream));
    SLHIP_CHECK(hipMemsetAsync(scratch->d_queue, 0, 16, stream));
    if (n_chunks > 0) {
        // with depth peeling every chunk takes the discard-testing form; without it only the alpha-tested draws do
        if (!d_depth_peel)
            k_raster<false><<<n_chunks, 256, 0, stream>>>(*pool, d_scenes, d_draws, d_chunks, n_chunks, W, H, d_depth_peel,
                                                          reinterpret_cast<unsigned long long*>(scratch->d_vis),
                                                          scratch->d_queue, scratch->queue_capacity, clipbuf, screen, raster_small);
        k_raster<true><<<d_depth_peel ? n_chunks : min(n_chunks, 4096u), 256, 0, stream>>>(
            *pool, d_scenes, d_draws, d_chunks, n_chunks, W, H, d_depth_peel, reinterpret_cast<unsigned long long*>(scratch->d_vis),
            scratch->d_queue, scratch->queue_capacity, clipbuf, screen, raster_small);
        mark(3, stream);
        k_large<<<2048, 256, 0, stream>>>(*pool, d_scenes, d_draws, W, H,
                                          reinterpret_cast<unsigned long long*>(scratch->d_vis), scratch->d_queue,
                                          scratch->queue_capacity, clipbuf, screen);
        SLHIP_LAUNCH_CHECK();
    }

    // deferred shade.  Auto exposure is decided per scene on the device; the host only
    // knows whether ANY post pass is needed, so inline tone mapping is used only when the
    // caller guarantees manual exposure through the flag below.
    ShadeParams prm;
    prm.W = W; prm.H = H; prm.n_scenes = n_scenes; prm.flags = flags; prm.S = S; prm.shadow_lights = NL;
    prm.inline_tonemap = 0;
    prm.want_lum = want_rgb ? 1 : 0;
    static const int shade_tiled = getenv("SLHIP_SHADE_TILED") ? atoi(getenv("SLHIP_SHADE_TILED")) : 1;   // developer knob
    prm.tiled = (shade_tiled && W % 32 == 0 && H % 8 == 0) ? 1 : 0;
    float* hdr0 = want_rgb ? scratch->d_hdr : nullptr;
    float* hdr1 = want_rgb ? scratch->d_hdr + 4 * (size_t)n_scenes * P : nullptr;
    mark(4, stream);
    // SSAO only where something can occlude (k_ssao_mask): needs k_shade's tiles to be the 8 x 8 tiles of the mask and 16-row bands
    static const int ssao_tiled_on = getenv("SLHIP_SSAO_TILED") ? atoi(getenv("SLHIP_SSAO_TILED")) : 1;   // developer knob
    const bool ssao_tiled = ssao && want_rgb && ssao_tiled_on && prm.tiled && H % 16 == 0 && W % 32 == 0 && (W / 8 + 1) * (H / 8 + 1) * 4 <= 60000;
    float2* ssao_tiles = ssao_tiled ? reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch->d_ao) + ssao_tiles_offset(n_scenes, W, H)) : nullptr;
    unsigned char* ssao_skip = ssao_tiled ? reinterpret_cast<unsigned char*>(ssao_tiles + (size_t)n_scenes * ssao_tiles_per_scene(W, H)) : nullptr;
    k_shade<<<pix_blocks, 256, 0, stream>>>(*pool, d_scenes, d_draws, prm,
                                            reinterpret_cast<const unsigned long long*>(scratch->d_vis), *out, hdr0,
                                            shadows ? scratch->d_shadow : nullptr, scratch->d_lum, clipbuf,
                                            ssao ? scratch->d_ao + (size_t)n_scenes * P : nullptr, vattr, ssao_tiles,
                                            shadows && !getenv("SLHIP_NO_PCF_TILES") ? scratch->d_shadow_tiles : nullptr);
    SLHIP_LAUNCH_CHECK();
    if (shadows && n_chunks > 0) {   // the maps were read for the last time: marked tiles back to 1.0
        k_shadow_restore<<<n_tile_words, 256, 0, stream>>>(reinterpret_cast<unsigned*>(scratch->d_shadow), scratch->d_shadow_tiles,
                                                          S, n_tile_words, NL);
        SLHIP_LAUNCH_CHECK();
    }

    if (want_rgb) {
        k_lum_reduce<<<n_scenes, 256, 0, stream>>>(d_scenes, (unsigned)((P + 255) / 256), P, scratch->d_lum);
        if (ssao) {
            mark(5, stream);
            const float* zpl = scratch->d_ao + (size_t)n_scenes * P;   // second half of d_ao
            const float* d_kernel_table = nullptr;   // the table's device address as a plain kernel argument
            SLHIP_CHECK(hipGetSymbolAddress((void**)&d_kernel_table, HIP_SYMBOL(c_ssao_kernel)));
            if (ssao_tiled) {
                k_ssao_mask<<<n_scenes, 256, (size_t)(W / 8 + 1) * (H / 8 + 1) * 4, stream>>>(d_scenes, W, H, ssao_tiles, ssao_skip, getenv("SLHIP_SSAO_DEBUG") ? atoi(getenv("SLHIP_SSAO_DEBUG")) : 0);
                k_ssao_tiled<<<8u * ((n_scenes + 7u) / 8u) * (unsigned)(H / 16) * 4u, 256, 0, stream>>>(d_scenes, n_scenes, W, H, out->d_cam_coord, out->d_normals,
                                                                                   zpl, scratch->d_ao, d_kernel_table, ssao_skip);
            } else
                k_ssao<<<pix_blocks, 256, 0, stream>>>(d_scenes, n_scenes, W, H, out->d_cam_coord, out->d_normals, zpl, scratch->d_ao,
                                                       d_kernel_table);
            mark(6, stream);
            k_ssao_apply<<<pix_blocks, 256, 0, stream>>>(d_scenes, n_scenes, W, H, hdr0, scratch->d_ao, zpl,
                                                         (flags & SLHIP_RENDER_KEEP_HDR) ? hdr1 : nullptr, scratch->d_lum, out->d_rgb, ssao_skip);
            mark(7, stream);
        } else {
            mark(7, stream);
            k_tonemap<<<pix_blocks, 256, 0, stream>>>(d_scenes, n_scenes, W, H, hdr0, scratch->d_lum, out->d_rgb);
        }
        SLHIP_LAUNCH_CHECK();
    }
    mark(kNumPhases, stream);
    return 0;
}
