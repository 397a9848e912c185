// slhip_synth.hip -- scene synthesis on the device (gfx950): the host work the reference does either
// side of the settle, as two kernels that write the records slhip_settle and slhip_render consume.
//
//   k_synth_stage   Scene::simulateTableTopScene, set-up part   reference src/scene.cpp:612-678
//                   randomQuaternion                            include/stillleben/pose.h:25-35
//   k_synth_place   Scene::chooseRandomCameraPose               src/scene.cpp:472-610
//                   Scene::chooseRandomLightDirection           src/scene.cpp:453-470
//                   computeFrustumCorners/computeShadowMapMatrix src/render_pass.cpp:69-211
//                   per-drawable uniforms                       src/render_pass.cpp:534-621,
//                                                               src/shaders/render_shader.cpp:233-265,355-377
//
// One 64-lane workgroup per scene; lane o owns object o (bbox-corner extrema and light-space bounds are
// wave min/max reductions -- exact, order-independent), lanes share the writes of the draw and chunk
// records.  A scene's inputs are a few hundred bytes and its outputs ~30 KB of records: the kernels are
// latency-bound and tiny next to the settle (microseconds per batch); what they buy is that nothing
// crosses PCIe and no host thread sits between the settle and the render.
//
// Arithmetic contract with oracle/synth_ref.c (bit-exact): float32, k-ordered fmaf chains for every
// 4x4 product, single rounded + - * / sqrt otherwise (-ffp-contract=off), polynomial log / sin / cos,
// float64 cofactors for the normal matrices.
#include "slhip_common.h"

namespace {

constexpr float kTwoPi = 6.28318530717958647692f, kPi = 3.14159265358979323846f;
constexpr float kElevSpan = 0.52359877559829887308f;   // pi/6: elevation in [30, 60] degrees (scene.cpp:483-487)

__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                       uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { STREAM_SCENE = 0, STREAM_ASSETS = 1, STREAM_QUAT = 2, STREAM_PBR = 3 };

__device__ __forceinline__ void draw4(const slhip_synth_params& p, uint32_t scene, uint32_t stream, uint32_t idx,
                                      uint32_t out[4])
{
    philox(p.scene_id_base + scene, stream, idx, 0x51DE5EEDu, p.seed_lo, p.seed_hi, out);
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// natural log of a positive normal float (Cephes logf scheme in fmaf form)
__device__ float det_logf(float x)
{
    uint32_t b = __float_as_uint(x);
    int e = (int)(b >> 23) - 126;
    b = (b & 0x007FFFFFu) | 0x3F000000u;
    float m = __uint_as_float(b);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else m = m - 1.0f;
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

// sin, cos for |x| <= 2 pi (quadrant reduction + Cephes polynomials)
__device__ void det_sincosf(float x, float& so, float& co)
{
    const float q = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-q, 1.5707962512969970703125f, x);
    r = fmaf(-q, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float sp = -1.9515295891e-4f;
    sp = fmaf(sp, z, 8.3321608736e-3f);
    sp = fmaf(sp, z, -1.6666654611e-1f);
    const float s = fmaf(sp * z, r, r);
    float cp = 2.443315711809948e-5f;
    cp = fmaf(cp, z, -1.388731625493765e-3f);
    cp = fmaf(cp, z, 4.166664568298827e-2f);
    const float c = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    const int k = (int)q & 3;
    if (k == 0) { so = s; co = c; }
    else if (k == 1) { so = c; co = -s; }
    else if (k == 2) { so = -s; co = -c; }
    else { so = -c; co = s; }
}

__device__ void normal2(uint32_t xa, uint32_t xb, float& n0, float& n1)
{
    const float u1 = u01(xa), u2 = u01(xb);
    const float r = sqrtf(-2.0f * det_logf(u1));
    float s, c;
    det_sincosf(fmaf(u2, kTwoPi, -kPi), s, c);
    n0 = r * c;
    n1 = r * s;
}

__device__ void mm4(const float* A, const float* B, float* C)
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
__device__ void mv4(const float* M, const float* v, float* o)
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
__device__ __forceinline__ float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void normalize3(float* v)
{
    const float l = sqrtf(dot3(v, v));
    v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}
__device__ __forceinline__ void identity4(float* m)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}
__device__ void inv_rigid(const float* m, float* o)
{
    identity4(o);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[4 * r + c] = m[4 * c + r];
        float a = fmaf(m[0 + r], m[3], 0.0f);
        a = fmaf(m[4 + r], m[7], a);
        a = fmaf(m[8 + r], m[11], a);
        o[4 * r + 3] = -a;
    }
}
__device__ void xform_point(const float* m, const float* p, float* o)
{
    const float v[4] = {p[0], p[1], p[2], 1.0f};
    float q[4];
    mv4(m, v, q);
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
}
__device__ void normal_matrix(const float* m, float* o)
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const double c10 = c * h - b * i, c11 = a * i - c * g, c12 = b * g - a * h;
    const double c20 = b * f - c * e, c21 = c * d - a * f, c22 = a * e - b * d;
    const double det = a * c00 + b * c01 + c * c02;
    o[0] = (float)(c00 / det); o[1] = (float)(c01 / det); o[2] = (float)(c02 / det); o[3] = 0.0f;
    o[4] = (float)(c10 / det); o[5] = (float)(c11 / det); o[6] = (float)(c12 / det); o[7] = 0.0f;
    o[8] = (float)(c20 / det); o[9] = (float)(c21 / det); o[10] = (float)(c22 / det); o[11] = 0.0f;
}
__device__ void rot_z(float a, float* m)
{
    float s, c;
    det_sincosf(a, s, c);
    identity4(m);
    m[0] = c; m[1] = -s; m[4] = s; m[5] = c;
}
__device__ void rot_y(float a, float* m)
{
    float s, c;
    det_sincosf(a, s, c);
    identity4(m);
    m[0] = c; m[2] = s; m[8] = -s; m[10] = c;
}
__device__ void translation4(float x, float y, float z, float* m)
{
    identity4(m);
    m[3] = x; m[7] = y; m[11] = z;
}
__device__ __forceinline__ void bbox_center(const slhip_asset& a, float* c)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = (a.bbox_min[k] + a.bbox_max[k]) / 2.0f;
}
__device__ __forceinline__ float bbox_diagonal(const slhip_asset& a)
{
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = a.bbox_max[k] - a.bbox_min[k];
    return sqrtf(dot3(d, d));
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ bool finite16(const float* m)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && (fabsf(m[i]) <= 3.4028234663852886e38f);   // false for inf and NaN
    return ok;
}

// 16-byte copies of a record by the lanes of the wave
__device__ __forceinline__ void copy16(void* dst, const void* src, unsigned bytes, unsigned lane, unsigned lanes)
{
    uint4* d = reinterpret_cast<uint4*>(dst);
    const uint4* s = reinterpret_cast<const uint4*>(src);
    for (unsigned i = lane; i < bytes / 16; i += lanes) d[i] = s[i];
}

// ------------------------------------------------------------------------------------------ stage
__global__ __launch_bounds__(64) void k_synth_stage(slhip_synth_params p, const slhip_asset* __restrict__ assets,
                                                   const uint16_t* __restrict__ asset_ids, slhip_body* __restrict__ bodies,
                                                   slhip_settle_scene* __restrict__ sscenes,
                                                   slhip_synth_object* __restrict__ objects,
                                                   slhip_synth_scene* __restrict__ scenes)
{
    __shared__ uint16_t perm[SLHIP_SYNTH_MAX_ASSETS];
    __shared__ uint16_t ids[SLHIP_SYNTH_MAX_OBJECTS];
    __shared__ float diam[SLHIP_SYNTH_MAX_OBJECTS];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    uint32_t x[4];
    if (asset_ids) {
        if (lane < p.n_objects) ids[lane] = asset_ids[(size_t)s * p.n_objects + lane];
    } else {
        for (uint32_t i = lane; i < p.n_assets; i += 64) perm[i] = (uint16_t)i;
        __syncthreads();
        if (lane == 0) {   // partial Fisher-Yates, serial: position i takes a uniformly chosen element of [i, n)
            for (uint32_t i = 0; i < p.n_objects; ++i) {
                if ((i & 3u) == 0) draw4(p, s, STREAM_ASSETS, i >> 2, x);
                const uint32_t span = p.n_assets - i;
                const uint32_t j = i + (uint32_t)(((uint64_t)x[i & 3u] * span) >> 32);
                const uint16_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
                ids[i] = perm[i];
            }
        }
    }
    __syncthreads();
    if (lane < p.n_objects) diam[lane] = bbox_diagonal(assets[ids[lane]]);
    __syncthreads();
    if (lane == 0) {
        // scene.cpp:650-657: plane pose = rotationZ(U(-pi,pi)) * translation(0, 0, half z)
        draw4(p, s, STREAM_SCENE, 0, x);
        const float yaw = fmaf(u01(x[0]), kTwoPi, -kPi);
        float rz[16], tz[16], pp[16];
        rot_z(yaw, rz);
        translation4(0.0f, 0.0f, p.plane_z, tz);
        mm4(rz, tz, pp);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            scenes[s].plane_pose[i] = pp[i];
            scenes[s].camera_pose[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        }
        slhip_settle_scene ss;
        ss.body_begin = s * p.n_objects;
        ss.body_end = (s + 1) * p.n_objects;
        ss.has_plane = 1;
        ss.plane_z = p.plane_z;
        sscenes[s] = ss;
    }
    if (lane >= p.n_objects) return;
    const uint32_t o = lane;
    const slhip_asset& a = assets[ids[o]];
    // scene.cpp:667-678: z advances by half a diameter before and after every object, in scene order
    float z = p.plane_z;
    for (uint32_t k = 0; k < o; ++k) {
        z = z + diam[k] / 2.0f;
        z = z + diam[k] / 2.0f;
    }
    const float diameter = diam[o];
    z = z + diameter / 2.0f;
    float q[4];
    draw4(p, s, STREAM_QUAT, o, x);
    normal2(x[0], x[1], q[0], q[1]);
    normal2(x[2], x[3], q[2], q[3]);
    const float ql = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
    const float qx = q[0] / ql, qy = q[1] / ql, qz = q[2] / ql, qw = q[3] / ql;
    float A[16], T[16], c[3], pose[16];
    identity4(A);
    A[0] = 1.0f - 2.0f * (qy * qy + qz * qz); A[1] = 2.0f * (qx * qy - qz * qw); A[2] = 2.0f * (qx * qz + qy * qw);
    A[4] = 2.0f * (qx * qy + qz * qw); A[5] = 1.0f - 2.0f * (qx * qx + qz * qz); A[6] = 2.0f * (qy * qz - qx * qw);
    A[8] = 2.0f * (qx * qz - qy * qw); A[9] = 2.0f * (qy * qz + qx * qw); A[10] = 1.0f - 2.0f * (qx * qx + qy * qy);
    A[11] = z;
    bbox_center(a, c);
    translation4(-c[0], -c[1], -c[2], T);
    mm4(A, T, pose);
    slhip_body b;
    memset(&b, 0, sizeof(b));
#pragma unroll
    for (int i = 0; i < 16; ++i) b.pose[i] = pose[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b.com[i] = a.com[i]; b.bsphere[i] = a.bsphere[i]; }
#pragma unroll
    for (int i = 0; i < 12; ++i) b.inv_inertia[i] = a.inv_inertia[i];
    b.inv_mass = 1.0f / a.mass;
    b.mu_s = a.mu_s; b.mu_d = a.mu_d; b.restitution = a.restitution;
    b.bbox_center[0] = c[0]; b.bbox_center[1] = c[1]; b.bbox_center[2] = c[2];
    b.bbox_center[3] = diameter / 2.0f;
    b.separation = __uint_as_float(0x7F800000u);
    b.wake_counter = 0.4f;
    b.hull_begin = a.hull_begin; b.hull_end = a.hull_end;
    bodies[(size_t)s * p.n_objects + o] = b;
    slhip_synth_object so;
    so.asset = ids[o];
    so.instance_index = o + 1;
    so.metallic = -1.0f; so.roughness = -1.0f;
    if (p.flags & SLHIP_SYNTH_RANDOM_PBR) {
        draw4(p, s, STREAM_PBR, o, x);
        so.metallic = u01(x[0]);
        so.roughness = u01(x[1]);
    }
    objects[(size_t)s * p.n_objects + o] = so;
}

// ------------------------------------------------------------------------------------------ place
__global__ __launch_bounds__(64) void k_synth_place(slhip_synth_params p, const slhip_asset* __restrict__ assets,
                                                   const slhip_draw* __restrict__ templates,
                                                   const slhip_body* __restrict__ bodies_all,
                                                   const slhip_synth_object* __restrict__ objects_all,
                                                   slhip_synth_scene* __restrict__ scenes, slhip_scene* __restrict__ out_scenes,
                                                   slhip_draw* __restrict__ out_draws, slhip_chunk* __restrict__ out_chunks)
{
    __shared__ slhip_scene sc_s;
    __shared__ uint32_t obj_nd[SLHIP_SYNTH_MAX_OBJECTS + 1], obj_nk[SLHIP_SYNTH_MAX_OBJECTS + 1];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint32_t rc = p.render_chunk ? p.render_chunk : p.n_scenes;
    const uint32_t local = s % rc;
    const bool has_obj = lane < p.n_objects;
    const slhip_body* bodies = bodies_all + (size_t)s * p.n_objects;
    const slhip_synth_object* objs = objects_all + (size_t)s * p.n_objects;
    // own object (lanes beyond n_objects carry neutral values into the reductions)
    float pose[16];
    slhip_synth_object me;
    me.asset = 0; me.instance_index = 0; me.metallic = -1.0f; me.roughness = -1.0f;
    identity4(pose);
    if (has_obj) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pose[i] = bodies[lane].pose[i];
        me = objs[lane];
    }
    const slhip_asset& a = assets[me.asset];
    const float INF = __uint_as_float(0x7F800000u);

    uint32_t x[4];
    draw4(p, s, STREAM_SCENE, 0, x);
    const float azimuth = fmaf(u01(x[1]), kTwoPi, -kPi);
    const float elevation = fmaf(u01(x[2]), kElevSpan, kElevSpan);

    // ---- camera pose (scene.cpp:472-610) ----
    float cam[16];
    {
        float rz[16], ry[16], t0[16], cam_rot[16], to_work[16];
        const float C[16] = {0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 1};
        rot_z(azimuth, rz);
        rot_y(elevation, ry);
        mm4(rz, ry, t0);
        mm4(t0, C, cam_rot);
        inv_rigid(cam_rot, to_work);
        float fr[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fr[0][c] = p.proj[12 + c] + p.proj[0 + c];
            fr[1][c] = p.proj[12 + c] - p.proj[0 + c];
            fr[2][c] = p.proj[12 + c] + p.proj[4 + c];
            fr[3][c] = p.proj[12 + c] - p.proj[4 + c];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float l = sqrtf(dot3(fr[k], fr[k]));
#pragma unroll
            for (int c = 0; c < 4; ++c) fr[k][c] = fr[k][c] / l;
        }
        float mn[4] = {INF, INF, INF, INF};
        if (has_obj) {
            float trans[16];
            mm4(to_work, pose, trans);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float c[3], pt[3];
                c[0] = (k & 1) ? a.bbox_max[0] : a.bbox_min[0];
                c[1] = (k & 2) ? a.bbox_max[1] : a.bbox_min[1];
                c[2] = (k & 4) ? a.bbox_max[2] : a.bbox_min[2];
                xform_point(trans, c, pt);
#pragma unroll
                for (int f = 0; f < 4; ++f) mn[f] = fminf(mn[f], dot3(fr[f], pt));
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) fr[f][3] = -wave_min(mn[f]);
        float la[3] = {fr[0][0], fr[0][2], fr[0][3]}, lb[3] = {fr[1][0], fr[1][2], fr[1][3]}, xi[3];
        cross3(la, lb, xi);
        if (fabsf(xi[2]) < 1e-3f) { xi[0] = 0.0f; xi[1] = 0.0f; xi[2] = 1.0f; }
        const float lr_x = xi[0] / xi[2], lr_z = xi[1] / xi[2];
        float ta[3] = {fr[2][1], fr[2][2], fr[2][3]}, tb[3] = {fr[3][1], fr[3][2], fr[3][3]};
        cross3(ta, tb, xi);
        if (fabsf(xi[2]) < 1e-3f) { xi[0] = 0.0f; xi[1] = 0.0f; xi[2] = 1.0f; }
        const float tb_y = xi[0] / xi[2], tb_z = xi[1] / xi[2];
        float tr[16];
        translation4(lr_x, tb_y, fminf(lr_z, tb_z), tr);
        mm4(cam_rot, tr, cam);
    }
    float w2c[16], c2w[16];
    inv_rigid(cam, w2c);
    inv_rigid(w2c, c2w);

    // ---- light direction (scene.cpp:453-470) ----
    float ld[3];
    {
        draw4(p, s, STREAM_SCENE, 1, x);
        float n0, n1, n2, unused;
        normal2(x[0], x[1], n0, n1);
        normal2(x[2], x[3], n2, unused);
        float d[3] = {n0, -fabsf(n1), -fabsf(n2)};
        normalize3(d);
        normalize3(d);
        const float lc[3] = {-d[0], -d[1], -d[2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float t = fmaf(cam[4 * r + 0], lc[0], 0.0f);
            t = fmaf(cam[4 * r + 1], lc[1], t);
            t = fmaf(cam[4 * r + 2], lc[2], t);
            ld[r] = t;
        }
    }
    const bool light_on = (p.light_color[0] != 0.0f || p.light_color[1] != 0.0f || p.light_color[2] != 0.0f) &&
                          (ld[0] != 0.0f || ld[1] != 0.0f || ld[2] != 0.0f);

    // ---- shadow matrix of light 0 (render_pass.cpp:69-211) ----
    float sm[16];
    identity4(sm);
    if ((p.flags & SLHIP_SYNTH_SHADOWS) && light_on) {
        float bc[3];
        bbox_center(a, bc);
        const float radius = bbox_diagonal(a) / 2.0f;
        float near_obj = INF, far_obj = -INF;
        if (has_obj) {
            float M[16], cc[3];
            mm4(w2c, pose, M);
            xform_point(M, bc, cc);
            const float np[4] = {cc[0], cc[1], cc[2] - radius, 1.0f}, fp[4] = {cc[0], cc[1], cc[2] + radius, 1.0f};
            float qn[4], qf[4];
            mv4(p.proj, np, qn);
            mv4(p.proj, fp, qf);
            near_obj = qn[2] / qn[3];
            far_obj = qf[2] / qf[3];
        }
        near_obj = wave_min(near_obj);
        far_obj = wave_max(far_obj);
        const float near = fmaxf(fmaxf(-1.0f, near_obj), -1.0f);
        const float far = fminf(far_obj, 1.0f);
        float z[3] = {ld[0], ld[1], ld[2]}, xa[3], ya[3];
        normalize3(z);
        const float up[3] = {0.0f, 0.0f, 1.0f};
        cross3(z, up, xa);
        normalize3(xa);
        cross3(z, xa, ya);
        normalize3(ya);
        float l2w[16], w2l[16];
        identity4(l2w);
#pragma unroll
        for (int r = 0; r < 3; ++r) { l2w[4 * r + 0] = xa[r]; l2w[4 * r + 1] = ya[r]; l2w[4 * r + 2] = z[r]; }
        inv_rigid(l2w, w2l);
        float mnv[3] = {INF, INF, INF}, mxv[3] = {-INF, -INF, -INF};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sx = (i == 1 || i == 2 || i == 5 || i == 6) ? 1.0f : -1.0f;
            const float sy = ((i & 3) < 2) ? 1.0f : -1.0f;
            const float h[4] = {sx, sy, i < 4 ? near : far, 1.0f};
            float t0[4], t1[4], corner[3], qv[3];
            mv4(p.proj_inv, h, t0);
            mv4(c2w, t0, t1);
            corner[0] = t1[0] / t1[3]; corner[1] = t1[1] / t1[3]; corner[2] = t1[2] / t1[3];
            xform_point(w2l, corner, qv);
#pragma unroll
            for (int k = 0; k < 3; ++k) { mnv[k] = fminf(mnv[k], qv[k]); mxv[k] = fmaxf(mxv[k], qv[k]); }
        }
        float near_l = mnv[2], far_l = mxv[2];
        const float mean_z = (near_l + far_l) / 2.0f;
        const float spread = far_l - mean_z;
        far_l = mean_z + 5.0f * spread;
        near_l = mean_z - 5.0f * spread;
        float L = mnv[0], R = mxv[0], T = mnv[1], B = mxv[1];
        float lo[2] = {INF, INF}, hi[2] = {-INF, -INF};
        if (has_obj) {
            float M[16], cc[3];
            mm4(w2l, pose, M);
            xform_point(M, bc, cc);
            lo[0] = cc[0] - radius; lo[1] = cc[1] - radius;
            hi[0] = cc[0] + radius; hi[1] = cc[1] + radius;
        }
        L = fmaxf(L, wave_min(lo[0])); R = fminf(R, wave_max(hi[0]));
        T = fmaxf(T, wave_min(lo[1])); B = fminf(B, wave_max(hi[1]));
        float Pm[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) Pm[i] = 0.0f;
        Pm[0] = 2.0f / (R - L); Pm[3] = -(R + L) / (R - L);
        Pm[5] = 2.0f / (B - T); Pm[7] = -(B + T) / (B - T);
        Pm[10] = 2.0f / (far_l - near_l); Pm[11] = -(far_l + near_l) / (far_l - near_l);
        Pm[15] = 1.0f;
        mm4(Pm, w2l, sm);
        if (!finite16(sm)) identity4(sm);
    }

    // ---- per-object record counts, prefix over the scene's objects ----
    const bool has_plane = p.plane_size[0] * p.plane_size[0] + p.plane_size[1] * p.plane_size[1] > 0.0f;
    if (has_obj) {
        obj_nd[lane] = a.draw_count;
        obj_nk[lane] = a.n_chunks;
    }
    __syncthreads();
    uint32_t nd0 = has_plane ? 1u : 0u, nk0 = nd0, prim0 = has_plane ? 2u : 0u, clip0 = has_plane ? 4u : 0u;
    uint32_t nd_total = nd0, nk_total = nk0, prim_total = prim0;
    for (uint32_t k = 0; k < p.n_objects; ++k) {          // serial prefix, wave-uniform (<= 64 objects)
        const slhip_asset& ak = assets[objs[k].asset];
        uint32_t tris = 0;
        for (uint32_t t = 0; t < ak.draw_count; ++t) tris += templates[ak.draw_begin + t].n_tris;
        if (k < lane) {
            nd0 += ak.draw_count; nk0 += ak.n_chunks; prim0 += tris; clip0 += ak.n_verts * ak.draw_count;
        }
        nd_total += ak.draw_count; nk_total += ak.n_chunks; prim_total += tris;
    }
    const uint32_t d0 = local * p.max_draws_per_scene;
    // strides too small for this scene (a host error: the strides come from the asset table's maxima): write the
    // scene EMPTY rather than out of bounds -- an all-background image is impossible to miss
    const bool fits = nd_total <= p.max_draws_per_scene && nk_total <= p.max_chunks_per_scene;
    if (!fits) { nd_total = 0; nk_total = 0; prim_total = 0; }
    slhip_draw* draws = out_draws + (size_t)s * p.max_draws_per_scene;
    slhip_chunk* chunks = out_chunks + (size_t)s * p.max_chunks_per_scene;
    const uint32_t clip_base_scene = local * p.max_clip_verts_per_scene;

    // ---- scene record (lane 0 assembles it in LDS, the wave stores it) ----
    if (lane == 0) {
        slhip_scene sc;
        memset(&sc, 0, sizeof(sc));
#pragma unroll
        for (int i = 0; i < 16; ++i) { sc.proj[i] = p.proj[i]; sc.world_to_cam[i] = w2c[i]; }
        sc.cam_position[0] = c2w[3]; sc.cam_position[1] = c2w[7]; sc.cam_position[2] = c2w[11]; sc.cam_position[3] = 1.0f;
        sc.light_dir[0][0] = ld[0]; sc.light_dir[0][1] = ld[1]; sc.light_dir[0][2] = ld[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { sc.light_color[0][i] = p.light_color[i]; sc.ambient[i] = p.ambient[i]; }
#pragma unroll
        for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l)
#pragma unroll
            for (int i = 0; i < 16; ++i) sc.shadow_mat[l][i] = l == 0 ? sm[i] : ((i % 5 == 0) ? 1.0f : 0.0f);
        sc.manual_exposure = p.manual_exposure;
        sc.draw_begin = d0;
        sc.draw_end = d0 + nd_total;
        sc.n_prims = prim_total;
        sc_s = sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) scenes[s].camera_pose[i] = cam[i];
    }
    __syncthreads();
    copy16(out_scenes + s, &sc_s, sizeof(slhip_scene), lane, 64);

    // ---- background plane (render_pass.cpp:545-582): lane 63 (never an object lane's job when n_objects < 64) ----
    if (fits && has_plane && lane == 63) {
        slhip_draw dr;
        memset(&dr, 0, sizeof(dr));
        float scal[16], o2w[16], nm[12];
        identity4(scal);
        scal[0] = p.plane_size[0] / 2.0f; scal[5] = p.plane_size[1] / 2.0f;
        mm4(scenes[s].plane_pose, scal, o2w);
        normal_matrix(o2w, nm);
#pragma unroll
        for (int i = 0; i < 16; ++i) { dr.mesh_to_object[i] = (i % 5 == 0) ? 1.0f : 0.0f; dr.object_to_world[i] = o2w[i]; }
#pragma unroll
        for (int i = 0; i < 12; ++i) dr.normal_to_world[i] = nm[i];
        dr.base_color[0] = 0.0f; dr.base_color[1] = 0.8f; dr.base_color[2] = 0.0f; dr.base_color[3] = 1.0f;
        dr.alpha_cutoff = 0.5f; dr.metallic = 0.04f; dr.roughness = 0.5f;
        dr.scene = local; dr.flags = SLHIP_DRAW_NO_VERTEX_ID;
        dr.n_verts = 4; dr.n_tris = 2; dr.prim_base = 0; dr.clip_base = clip_base_scene;
        draws[0] = dr;
        slhip_chunk ck;
        ck.scene = local; ck.draw = d0; ck.first_tri = 0; ck.count = 2;
        chunks[0] = ck;
    }

    // ---- object draws + chunks ----
    if (fits && has_obj) {
        float m2o[16], m2w[16], nm[12];
#pragma unroll
        for (int i = 0; i < 16; ++i) m2o[i] = a.mesh_to_object[i];
        mm4(pose, m2o, m2w);
        normal_matrix(m2w, nm);
        uint32_t nd = nd0, nk = nk0, prim = prim0, clip = clip_base_scene + clip0;
        for (uint32_t t = 0; t < a.draw_count; ++t) {
            slhip_draw dr = templates[a.draw_begin + t];
#pragma unroll
            for (int i = 0; i < 16; ++i) { dr.mesh_to_object[i] = m2o[i]; dr.object_to_world[i] = pose[i]; }
#pragma unroll
            for (int i = 0; i < 12; ++i) dr.normal_to_world[i] = nm[i];
            if (me.metallic >= 0.0f) dr.metallic = me.metallic;
            if (me.roughness >= 0.0f) dr.roughness = me.roughness;
            dr.instance_index = me.instance_index;
            dr.scene = local;
            dr.n_verts = a.n_verts;
            dr.prim_base = prim;
            dr.clip_base = clip;
            draws[nd] = dr;
            for (uint32_t first = 0; first < dr.n_tris; first += SLHIP_CHUNK_TRIS) {
                const uint32_t left = dr.n_tris - first;
                slhip_chunk ck;
                ck.scene = local; ck.draw = d0 + nd; ck.first_tri = first;
                ck.count = left < SLHIP_CHUNK_TRIS ? left : SLHIP_CHUNK_TRIS;
                chunks[nk++] = ck;
            }
            prim += dr.n_tris;
            clip += a.n_verts;
            ++nd;
        }
    }
    // ---- unused slots: empty draws / chunks ----
    for (uint32_t i = nd_total + lane; i < p.max_draws_per_scene; i += 64) {
        slhip_draw dr;
        memset(&dr, 0, sizeof(dr));
        dr.scene = local;
        draws[i] = dr;
    }
    for (uint32_t i = nk_total + lane; i < p.max_chunks_per_scene; i += 64) {
        slhip_chunk ck;
        ck.scene = local; ck.draw = d0; ck.first_tri = 0; ck.count = 0;
        chunks[i] = ck;
    }
}

int check_params(const slhip_synth_params* p, const char* what)
{
    if (!p) {
        slhip::set_error("%s: null params", what);
        return -1;
    }
    if (p->n_objects == 0 || p->n_objects > SLHIP_SYNTH_MAX_OBJECTS) {
        slhip::set_error("%s: n_objects must be in [1, %u]", what, SLHIP_SYNTH_MAX_OBJECTS);
        return -1;
    }
    if (p->n_assets == 0) {
        slhip::set_error("%s: empty asset table", what);
        return -1;
    }
    return 0;
}

}  // namespace

static_assert(sizeof(slhip_asset) == 224, "slhip_asset layout");
static_assert(sizeof(slhip_synth_params) == 224, "slhip_synth_params layout");
static_assert(sizeof(slhip_synth_object) == 16, "slhip_synth_object layout");
static_assert(sizeof(slhip_synth_scene) == 128, "slhip_synth_scene layout");
static_assert(sizeof(slhip_draw) % 16 == 0 && sizeof(slhip_scene) % 16 == 0, "records are copied in 16-byte units");

extern "C" int slhip_synth_stage(const slhip_synth_params* params, const slhip_asset* d_assets, const uint16_t* d_asset_ids,
                                 slhip_body* d_bodies, slhip_settle_scene* d_settle_scenes, slhip_synth_object* d_objects,
                                 slhip_synth_scene* d_scenes, void* stream)
{
    if (int st = check_params(params, "slhip_synth_stage")) return st;
    if (!d_assets || !d_bodies || !d_settle_scenes || !d_objects || !d_scenes) {
        slhip::set_error("slhip_synth_stage: null argument");
        return -1;
    }
    if (!d_asset_ids) {
        if (!(params->flags & SLHIP_SYNTH_SAMPLE_DISTINCT)) {
            slhip::set_error("slhip_synth_stage: d_asset_ids is NULL and SLHIP_SYNTH_SAMPLE_DISTINCT is not set");
            return -1;
        }
        if (params->n_assets < params->n_objects || params->n_assets > SLHIP_SYNTH_MAX_ASSETS) {
            slhip::set_error("slhip_synth_stage: sampling %u distinct classes needs %u <= n_assets (%u) <= %u",
                             params->n_objects, params->n_objects, params->n_assets, SLHIP_SYNTH_MAX_ASSETS);
            return -1;
        }
    }
    if (params->n_scenes == 0) return 0;
    k_synth_stage<<<params->n_scenes, 64, 0, (hipStream_t)stream>>>(*params, d_assets, d_asset_ids, d_bodies, d_settle_scenes,
                                                                    d_objects, d_scenes);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_synth_place(const slhip_synth_params* params, const slhip_asset* d_assets, const slhip_draw* d_templates,
                                 const slhip_body* d_bodies, const slhip_synth_object* d_objects, slhip_synth_scene* d_scenes,
                                 slhip_scene* d_out_scenes, slhip_draw* d_out_draws, slhip_chunk* d_out_chunks, void* stream)
{
    if (int st = check_params(params, "slhip_synth_place")) return st;
    if (!d_assets || !d_templates || !d_bodies || !d_objects || !d_scenes || !d_out_scenes || !d_out_draws || !d_out_chunks) {
        slhip::set_error("slhip_synth_place: null argument");
        return -1;
    }
    if (params->max_draws_per_scene == 0 || params->max_chunks_per_scene == 0) {
        slhip::set_error("slhip_synth_place: record strides (max_draws_per_scene, max_chunks_per_scene) must be set");
        return -1;
    }
    if (params->n_scenes == 0) return 0;
    k_synth_place<<<params->n_scenes, 64, 0, (hipStream_t)stream>>>(*params, d_assets, d_templates, d_bodies, d_objects,
                                                                    d_scenes, d_out_scenes, d_out_draws, d_out_chunks);
    SLHIP_LAUNCH_CHECK();
    return 0;
}
