/*
 * ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * CPU restatement of the host work the reference does either side of the settle, in the form the
 * device kernels of slhip_synth_stage / slhip_synth_place (include/slhip.h) compute it:
 *
 *   stage   Scene::simulateTableTopScene, set-up part          /root/reference/src/scene.cpp:612-678
 *           randomQuaternion                                   include/stillleben/pose.h:25-35
 *   place   Scene::chooseRandomCameraPose                      src/scene.cpp:472-610
 *           Scene::chooseRandomLightDirection                  src/scene.cpp:453-470
 *           computeFrustumCorners / computeShadowMapMatrix     src/render_pass.cpp:69-211
 *           per-drawable uniforms                              src/render_pass.cpp:534-621,
 *                                                              src/shaders/render_shader.cpp:233-265,355-377
 *
 * PARITY STATUS: the reference seeds std::mt19937 from std::random_device (scene.cpp:147-148) and
 * draws through libstdc++ distributions, so no two reference runs agree and the library cannot be
 * built here: the DISTRIBUTIONS and the geometry are the contract.  Pinned by
 * tests/test_oracle_synth.py against (i) the per-scene Python mirror of the same reference functions
 * (stillleben_amd/physics.py, camera_placement.py, _shadow.py, _batch.py) fed with this file's random
 * draws, (ii) analytic properties (all objects inside the frustum, elevation in [30,60] deg, unit
 * quaternions, light from above and from the camera side).  "parity unpinned" against reference OUTPUT.
 *
 * Arithmetic rules (shared with the HIP path, which must match bit for bit): float32; every 4x4
 * product is the k-ordered fmaf chain of render_ref.c:mm4; dot3 = fmaf(a2,b2,fmaf(a1,b1,a0*b0));
 * everything else is single rounded + - * / sqrt; -ffp-contract=off.  log / sin / cos are the
 * polynomials below (libm and the device's differ in the last bits).  Normal matrices are the
 * cofactor inverse-transpose evaluated in float64.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/slhip.h"

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { STREAM_SCENE = 0, STREAM_ASSETS = 1, STREAM_QUAT = 2, STREAM_PBR = 3 };

static void draw4(const slhip_synth_params* p, uint32_t scene, uint32_t stream, uint32_t idx, uint32_t out[4])
{
    philox(p->scene_id_base + scene, stream, idx, 0x51DE5EEDu, p->seed_lo, p->seed_hi, out);
}

static float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); } /* (0,1) */

/* ------------------------------------------------------------------ deterministic log / sincos */
/* natural log of a positive normal float (Cephes logf scheme, fmaf form); |err| < 2 ulp */
static float det_logf(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    int e = (int)(b >> 23) - 126;                       /* x = m * 2^e, m in [0.5, 1) */
    b = (b & 0x007FFFFFu) | 0x3F000000u;
    float m;
    memcpy(&m, &b, 4);
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

/* sin and cos for |x| <= 2 pi: quadrant by round-to-nearest of x * 2/pi, two-term Cody-Waite
   reduction, Cephes sinf / cosf polynomials on [-pi/4, pi/4]; |err| < 1e-7 absolute */
static void det_sincosf(float x, float* s_out, float* c_out)
{
    const float q = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-q, 1.5707962512969970703125f, x);     /* pi/2 high part (exact product for |q| <= 4) */
    r = fmaf(-q, 7.54978995489188216e-8f, r);             /* pi/2 low part */
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
    float so, co;
    if (k == 0) { so = s; co = c; }
    else if (k == 1) { so = c; co = -s; }
    else if (k == 2) { so = -s; co = -c; }
    else { so = -c; co = s; }
    *s_out = so;
    *c_out = co;
}

/* Box-Muller: two uniforms -> two N(0,1) */
static void normal2(uint32_t xa, uint32_t xb, float* n0, float* n1)
{
    const float u1 = u01(xa), u2 = u01(xb);
    const float r = sqrtf(-2.0f * det_logf(u1));
    float s, c;
    det_sincosf(fmaf(u2, 6.28318530717958647692f, -3.14159265358979323846f), &s, &c);
    *n0 = r * c;
    *n1 = r * s;
}

/* ------------------------------------------------------------------ small linear algebra */
static void mm4(const float* A, const float* B, float* C)
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
static void mv4(const float* M, const float* v, float* o)
{
    for (int r = 0; r < 4; ++r) {
        float a = fmaf(M[4 * r + 0], v[0], 0.0f);
        a = fmaf(M[4 * r + 1], v[1], a);
        a = fmaf(M[4 * r + 2], v[2], a);
        a = fmaf(M[4 * r + 3], v[3], a);
        o[r] = a;
    }
}
static float dot3(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
static void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void normalize3(float* v)
{
    const float l = sqrtf(dot3(v, v));
    v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}
static void identity4(float* m)
{
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
/* Matrix4::invertedRigid: [R^T | -R^T t] */
static void inv_rigid(const float* m, float* o)
{
    identity4(o);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o[4 * r + c] = m[4 * c + r];
        float a = fmaf(m[0 + r], m[3], 0.0f);
        a = fmaf(m[4 + r], m[7], a);
        a = fmaf(m[8 + r], m[11], a);
        o[4 * r + 3] = -a;
    }
}
/* affine transformPoint (w == 1 exactly) */
static void xform_point(const float* m, const float* p, float* o)
{
    const float v[4] = {p[0], p[1], p[2], 1.0f};
    float q[4];
    mv4(m, v, q);
    o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
}
/* Matrix4::normalMatrix(): inverse-transpose of the upper 3x3, float64 cofactors (rows padded to 4) */
static void normal_matrix(const float* m, float* o)
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
static void rot_z(float a, float* m)
{
    float s, c;
    det_sincosf(a, &s, &c);
    identity4(m);
    m[0] = c; m[1] = -s; m[4] = s; m[5] = c;
}
static void rot_y(float a, float* m)
{
    float s, c;
    det_sincosf(a, &s, &c);
    identity4(m);
    m[0] = c; m[2] = s; m[8] = -s; m[10] = c;
}
static void translation4(float x, float y, float z, float* m)
{
    identity4(m);
    m[3] = x; m[7] = y; m[11] = z;
}

/* Range3D helpers: centre = (min + max) / 2, radius = |max - min| / 2 ("the diameter") */
static void bbox_center(const slhip_asset* a, float* c)
{
    for (int k = 0; k < 3; ++k) c[k] = (a->bbox_min[k] + a->bbox_max[k]) / 2.0f;
}
static float bbox_diagonal(const slhip_asset* a)
{
    float d[3];
    for (int k = 0; k < 3; ++k) d[k] = a->bbox_max[k] - a->bbox_min[k];
    return sqrtf(dot3(d, d));
}

/* ------------------------------------------------------------------ stage */
/* raw random draws of one scene, for cross-checks against the per-scene Python mirror:
   out = yaw, azimuth, elevation, light normals[3], then per object quaternion[4], metallic, roughness */
int slref_synth_draws(const slhip_synth_params* p, uint32_t scene, float* out)
{
    uint32_t x[4];
    draw4(p, scene, STREAM_SCENE, 0, x);
    out[0] = fmaf(u01(x[0]), 6.28318530717958647692f, -3.14159265358979323846f);
    out[1] = fmaf(u01(x[1]), 6.28318530717958647692f, -3.14159265358979323846f);
    out[2] = fmaf(u01(x[2]), 0.52359877559829887308f, 0.52359877559829887308f);
    draw4(p, scene, STREAM_SCENE, 1, x);
    float dummy;
    normal2(x[0], x[1], &out[3], &out[4]);
    normal2(x[2], x[3], &out[5], &dummy);
    for (uint32_t o = 0; o < p->n_objects; ++o) {
        float* q = out + 6 + 6 * o;
        draw4(p, scene, STREAM_QUAT, o, x);
        normal2(x[0], x[1], &q[0], &q[1]);
        normal2(x[2], x[3], &q[2], &q[3]);
        draw4(p, scene, STREAM_PBR, o, x);
        q[4] = u01(x[0]);
        q[5] = u01(x[1]);
    }
    return 0;
}

static void sample_distinct(const slhip_synth_params* p, uint32_t scene, uint16_t* ids)
{
    /* partial Fisher-Yates over the class list: position i takes a uniformly chosen element of [i, n) */
    uint16_t perm[SLHIP_SYNTH_MAX_ASSETS];
    for (uint32_t i = 0; i < p->n_assets; ++i) perm[i] = (uint16_t)i;
    uint32_t x[4];
    for (uint32_t i = 0; i < p->n_objects; ++i) {
        if ((i & 3u) == 0) draw4(p, scene, STREAM_ASSETS, i >> 2, x);
        const uint32_t span = p->n_assets - i;
        const uint32_t j = i + (uint32_t)(((uint64_t)x[i & 3u] * span) >> 32);
        const uint16_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        ids[i] = perm[i];
    }
}

int slref_synth_stage(const slhip_synth_params* p, const slhip_asset* assets, const uint16_t* asset_ids,
                      slhip_body* bodies, slhip_settle_scene* sscenes, slhip_synth_object* objects,
                      slhip_synth_scene* scenes)
{
    if (p->n_objects > SLHIP_SYNTH_MAX_OBJECTS) return -1;
    if (!asset_ids && (!(p->flags & SLHIP_SYNTH_SAMPLE_DISTINCT) || p->n_assets < p->n_objects ||
                       p->n_assets > SLHIP_SYNTH_MAX_ASSETS))
        return -1;
    for (uint32_t s = 0; s < p->n_scenes; ++s) {
        uint32_t x[4];
        uint16_t ids[SLHIP_SYNTH_MAX_OBJECTS];
        if (asset_ids) memcpy(ids, asset_ids + (size_t)s * p->n_objects, 2 * p->n_objects);
        else sample_distinct(p, s, ids);
        /* scene.cpp:650-657: topSidePlanePose = T * rotationZ(U(-pi,pi)) * translation(0,0,half z), T = identity */
        draw4(p, s, STREAM_SCENE, 0, x);
        const float yaw = fmaf(u01(x[0]), 6.28318530717958647692f, -3.14159265358979323846f);
        float rz[16], tz[16];
        rot_z(yaw, rz);
        translation4(0.0f, 0.0f, p->plane_z, tz);
        mm4(rz, tz, scenes[s].plane_pose);
        identity4(scenes[s].camera_pose);
        sscenes[s].body_begin = s * p->n_objects;
        sscenes[s].body_end = (s + 1) * p->n_objects;
        sscenes[s].has_plane = 1;
        sscenes[s].plane_z = p->plane_z;
        float z = p->plane_z;                                       /* scene.cpp:662 */
        for (uint32_t o = 0; o < p->n_objects; ++o) {
            const slhip_asset* a = assets + ids[o];
            slhip_body* b = bodies + (size_t)s * p->n_objects + o;
            slhip_synth_object* so = objects + (size_t)s * p->n_objects + o;
            memset(b, 0, sizeof(*b));
            /* scene.cpp:667-678: stack along z by bbox diameters, random orientation */
            const float diameter = bbox_diagonal(a);
            z = z + diameter / 2.0f;
            const float pos_z = z;
            z = z + diameter / 2.0f;
            float q[4];
            draw4(p, s, STREAM_QUAT, o, x);
            normal2(x[0], x[1], &q[0], &q[1]);
            normal2(x[2], x[3], &q[2], &q[3]);
            const float ql = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));   /* pose.h:25-35 */
            const float qx = q[0] / ql, qy = q[1] / ql, qz = q[2] / ql, qw = q[3] / ql;
            float A[16], T[16], c[3];
            identity4(A);                                          /* Quaternion::toMatrix, row-major */
            A[0] = 1.0f - 2.0f * (qy * qy + qz * qz); A[1] = 2.0f * (qx * qy - qz * qw); A[2] = 2.0f * (qx * qz + qy * qw);
            A[4] = 2.0f * (qx * qy + qz * qw); A[5] = 1.0f - 2.0f * (qx * qx + qz * qz); A[6] = 2.0f * (qy * qz - qx * qw);
            A[8] = 2.0f * (qx * qz - qy * qw); A[9] = 2.0f * (qy * qz + qx * qw); A[10] = 1.0f - 2.0f * (qx * qx + qy * qy);
            A[11] = pos_z;
            bbox_center(a, c);
            translation4(-c[0], -c[1], -c[2], T);
            mm4(A, T, b->pose);
            memcpy(b->com, a->com, 16);
            memcpy(b->inv_inertia, a->inv_inertia, 48);
            b->inv_mass = 1.0f / a->mass;
            b->mu_s = a->mu_s; b->mu_d = a->mu_d; b->restitution = a->restitution;
            memcpy(b->bsphere, a->bsphere, 16);
            b->bbox_center[0] = c[0]; b->bbox_center[1] = c[1]; b->bbox_center[2] = c[2];
            b->bbox_center[3] = diameter / 2.0f;
            b->separation = INFINITY;
            b->wake_counter = 0.4f;
            b->hull_begin = a->hull_begin; b->hull_end = a->hull_end;
            so->asset = ids[o];
            so->instance_index = o + 1;                            /* scene.cpp:285-287 */
            so->metallic = -1.0f; so->roughness = -1.0f;
            if (p->flags & SLHIP_SYNTH_RANDOM_PBR) {
                draw4(p, s, STREAM_PBR, o, x);
                so->metallic = u01(x[0]);
                so->roughness = u01(x[1]);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ place */
static void corner(const slhip_asset* a, int k, float* c)
{
    /* Range3D corner order of scene.cpp:535-542: x fastest, then y, then z */
    c[0] = (k & 1) ? a->bbox_max[0] : a->bbox_min[0];
    c[1] = (k & 2) ? a->bbox_max[1] : a->bbox_min[1];
    c[2] = (k & 4) ? a->bbox_max[2] : a->bbox_min[2];
}

static void camera_pose(const slhip_synth_params* p, const slhip_asset* assets, const slhip_body* bodies,
                        const slhip_synth_object* objs, float azimuth, float elevation, float* out)
{
    /* scene.cpp:489-499 */
    float rz[16], ry[16], t0[16], cam_rot[16], to_work[16];
    const float C[16] = {0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 1};   /* columns (-y, -z, x) */
    rot_z(azimuth, rz);
    rot_y(elevation, ry);
    mm4(rz, ry, t0);
    mm4(t0, C, cam_rot);
    inv_rigid(cam_rot, to_work);
    /* a) frustum planes (scene.cpp:546-557) */
    float fr[4][4];
    for (int c = 0; c < 4; ++c) {
        fr[0][c] = p->proj[12 + c] + p->proj[0 + c];
        fr[1][c] = p->proj[12 + c] - p->proj[0 + c];
        fr[2][c] = p->proj[12 + c] + p->proj[4 + c];
        fr[3][c] = p->proj[12 + c] - p->proj[4 + c];
    }
    for (int k = 0; k < 4; ++k) {
        const float l = sqrtf(dot3(fr[k], fr[k]));
        for (int c = 0; c < 4; ++c) fr[k][c] = fr[k][c] / l;
    }
    /* b) push each plane to the nearest bbox corner (scene.cpp:525-573) */
    float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    for (uint32_t o = 0; o < p->n_objects; ++o) {
        float trans[16];
        mm4(to_work, bodies[o].pose, trans);
        for (int k = 0; k < 8; ++k) {
            float c[3], pt[3];
            corner(assets + objs[o].asset, k, c);
            xform_point(trans, c, pt);
            for (int f = 0; f < 4; ++f) mn[f] = fminf(mn[f], dot3(fr[f], pt));
        }
    }
    for (int f = 0; f < 4; ++f) fr[f][3] = -mn[f];
    /* c) left/right and top/bottom intersection lines (scene.cpp:575-606) */
    float la[3] = {fr[0][0], fr[0][2], fr[0][3]}, lb[3] = {fr[1][0], fr[1][2], fr[1][3]}, x[3];
    cross3(la, lb, x);
    if (fabsf(x[2]) < 1e-3f) { x[0] = 0.0f; x[1] = 0.0f; x[2] = 1.0f; }
    const float lr_x = x[0] / x[2], lr_z = x[1] / x[2];
    float ta[3] = {fr[2][1], fr[2][2], fr[2][3]}, tb[3] = {fr[3][1], fr[3][2], fr[3][3]};
    cross3(ta, tb, x);
    if (fabsf(x[2]) < 1e-3f) { x[0] = 0.0f; x[1] = 0.0f; x[2] = 1.0f; }
    const float tb_y = x[0] / x[2], tb_z = x[1] / x[2];
    float tr[16];
    translation4(lr_x, tb_y, fminf(lr_z, tb_z), tr);
    mm4(cam_rot, tr, out);                                           /* scene.cpp:608-610 */
}

static int finite16(const float* m)
{
    for (int i = 0; i < 16; ++i)
        if (!isfinite(m[i])) return 0;
    return 1;
}

/* render_pass.cpp:69-211 for one light */
static void shadow_matrix(const slhip_synth_params* p, const slhip_asset* assets, const slhip_body* bodies,
                          const slhip_synth_object* objs, const float* w2c, const float* light_dir, float* out)
{
    float near_obj = INFINITY, far_obj = -INFINITY;
    for (uint32_t o = 0; o < p->n_objects; ++o) {
        const slhip_asset* a = assets + objs[o].asset;
        float M[16], c[3], cc[3];
        mm4(w2c, bodies[o].pose, M);
        bbox_center(a, c);
        xform_point(M, c, cc);
        const float radius = bbox_diagonal(a) / 2.0f;
        const float np[4] = {cc[0], cc[1], cc[2] - radius, 1.0f}, fp[4] = {cc[0], cc[1], cc[2] + radius, 1.0f};
        float qn[4], qf[4];
        mv4(p->proj, np, qn);
        mv4(p->proj, fp, qf);
        near_obj = fminf(near_obj, qn[2] / qn[3]);
        far_obj = fmaxf(far_obj, qf[2] / qf[3]);
    }
    const float near = fmaxf(fmaxf(-1.0f, near_obj), -1.0f);
    const float far = fminf(far_obj, 1.0f);
    float c2w[16];
    inv_rigid(w2c, c2w);
    static const float sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {1, 1, -1, -1, 1, 1, -1, -1};
    float corners[8][3];
    for (int i = 0; i < 8; ++i) {
        const float h[4] = {sx[i], sy[i], i < 4 ? near : far, 1.0f};
        float a[4], b[4];
        mv4(p->proj_inv, h, a);
        mv4(c2w, a, b);
        corners[i][0] = b[0] / b[3]; corners[i][1] = b[1] / b[3]; corners[i][2] = b[2] / b[3];
    }
    float z[3] = {light_dir[0], light_dir[1], light_dir[2]}, x[3], y[3];
    normalize3(z);
    const float up[3] = {0.0f, 0.0f, 1.0f};
    cross3(z, up, x);
    normalize3(x);
    cross3(z, x, y);
    normalize3(y);
    float l2w[16], w2l[16];
    identity4(l2w);
    for (int r = 0; r < 3; ++r) { l2w[4 * r + 0] = x[r]; l2w[4 * r + 1] = y[r]; l2w[4 * r + 2] = z[r]; }
    inv_rigid(l2w, w2l);
    float mnv[3] = {INFINITY, INFINITY, INFINITY}, mxv[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < 8; ++i) {
        float q[3];
        xform_point(w2l, corners[i], q);
        for (int k = 0; k < 3; ++k) { mnv[k] = fminf(mnv[k], q[k]); mxv[k] = fmaxf(mxv[k], q[k]); }
    }
    float near_l = mnv[2], far_l = mxv[2];
    const float mean_z = (near_l + far_l) / 2.0f;
    const float spread = far_l - mean_z;
    far_l = mean_z + 5.0f * spread;
    near_l = mean_z - 5.0f * spread;
    float L = mnv[0], R = mxv[0], T = mnv[1], B = mxv[1];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t o = 0; o < p->n_objects; ++o) {
        const slhip_asset* a = assets + objs[o].asset;
        float M[16], c[3], cc[3];
        mm4(w2l, bodies[o].pose, M);
        bbox_center(a, c);
        xform_point(M, c, cc);
        const float radius = bbox_diagonal(a) / 2.0f;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], cc[k] - radius); hi[k] = fmaxf(hi[k], cc[k] + radius); }
    }
    L = fmaxf(L, lo[0]); R = fminf(R, hi[0]);
    T = fmaxf(T, lo[1]); B = fminf(B, hi[1]);
    float Pm[16];
    memset(Pm, 0, 64);
    Pm[0] = 2.0f / (R - L); Pm[3] = -(R + L) / (R - L);
    Pm[5] = 2.0f / (B - T); Pm[7] = -(B + T) / (B - T);
    Pm[10] = 2.0f / (far_l - near_l); Pm[11] = -(far_l + near_l) / (far_l - near_l);
    Pm[15] = 1.0f;
    mm4(Pm, w2l, out);
    if (!finite16(out)) identity4(out);
}

int slref_synth_place(const slhip_synth_params* p, const slhip_asset* assets, const slhip_draw* templates,
                      const slhip_body* bodies_all, const slhip_synth_object* objects_all, slhip_synth_scene* scenes,
                      slhip_scene* out_scenes, slhip_draw* out_draws, slhip_chunk* out_chunks)
{
    const uint32_t rc = p->render_chunk ? p->render_chunk : p->n_scenes;
    for (uint32_t s = 0; s < p->n_scenes; ++s) {
        const slhip_body* bodies = bodies_all + (size_t)s * p->n_objects;
        const slhip_synth_object* objs = objects_all + (size_t)s * p->n_objects;
        const uint32_t local = s % rc;                              /* indices are relative to the render chunk */
        slhip_scene* sc = out_scenes + s;
        memset(sc, 0, sizeof(*sc));
        uint32_t x[4];
        draw4(p, s, STREAM_SCENE, 0, x);
        const float azimuth = fmaf(u01(x[1]), 6.28318530717958647692f, -3.14159265358979323846f);
        const float elevation = fmaf(u01(x[2]), 0.52359877559829887308f, 0.52359877559829887308f);
        float cam[16], w2c[16], c2w[16];
        camera_pose(p, assets, bodies, objs, azimuth, elevation, cam);
        memcpy(scenes[s].camera_pose, cam, 64);
        inv_rigid(cam, w2c);
        inv_rigid(w2c, c2w);
        memcpy(sc->proj, p->proj, 64);
        memcpy(sc->world_to_cam, w2c, 64);
        sc->cam_position[0] = c2w[3]; sc->cam_position[1] = c2w[7]; sc->cam_position[2] = c2w[11];   /* render_shader.cpp:246 */
        sc->cam_position[3] = 1.0f;
        /* scene.cpp:453-470 */
        draw4(p, s, STREAM_SCENE, 1, x);
        float n0, n1, n2, unused;
        normal2(x[0], x[1], &n0, &n1);
        normal2(x[2], x[3], &n2, &unused);
        float d[3] = {n0, -fabsf(n1), -fabsf(n2)};
        normalize3(d);
        normalize3(d);
        const float lc[3] = {-d[0], -d[1], -d[2]};
        float ld[3];
        for (int r = 0; r < 3; ++r) {
            float a = fmaf(cam[4 * r + 0], lc[0], 0.0f);
            a = fmaf(cam[4 * r + 1], lc[1], a);
            a = fmaf(cam[4 * r + 2], lc[2], a);
            ld[r] = a;
        }
        sc->light_dir[0][0] = ld[0]; sc->light_dir[0][1] = ld[1]; sc->light_dir[0][2] = ld[2];
        memcpy(sc->light_color[0], p->light_color, 12);
        memcpy(sc->ambient, p->ambient, 12);
        for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l) identity4(sc->shadow_mat[l]);
        const int light_on = (p->light_color[0] != 0.0f || p->light_color[1] != 0.0f || p->light_color[2] != 0.0f) &&
                             (ld[0] != 0.0f || ld[1] != 0.0f || ld[2] != 0.0f);
        if ((p->flags & SLHIP_SYNTH_SHADOWS) && light_on) shadow_matrix(p, assets, bodies, objs, w2c, ld, sc->shadow_mat[0]);
        sc->manual_exposure = p->manual_exposure;

        /* draw list: background plane first (render_pass.cpp:545-582), then the objects in scene order */
        const uint32_t d0 = local * p->max_draws_per_scene;
        slhip_draw* draws = out_draws + (size_t)s * p->max_draws_per_scene;
        slhip_chunk* chunks = out_chunks + (size_t)s * p->max_chunks_per_scene;
        uint32_t nd = 0, nk = 0, prim = 0, clip = local * p->max_clip_verts_per_scene;
        const int has_plane = p->plane_size[0] * p->plane_size[0] + p->plane_size[1] * p->plane_size[1] > 0.0f;
        {   /* the strides must hold the scene: checked before anything is written */
            uint32_t need_d = has_plane ? 1u : 0u, need_k = need_d;
            for (uint32_t o = 0; o < p->n_objects; ++o) {
                need_d += assets[objs[o].asset].draw_count;
                need_k += assets[objs[o].asset].n_chunks;
            }
            if (need_d > p->max_draws_per_scene || need_k > p->max_chunks_per_scene) return -2;
        }
        if (has_plane) {
            slhip_draw* dr = draws + nd;
            memset(dr, 0, sizeof(*dr));
            float scal[16];
            identity4(scal);
            scal[0] = p->plane_size[0] / 2.0f; scal[5] = p->plane_size[1] / 2.0f;
            identity4(dr->mesh_to_object);
            mm4(scenes[s].plane_pose, scal, dr->object_to_world);
            normal_matrix(dr->object_to_world, dr->normal_to_world);
            dr->base_color[0] = 0.0f; dr->base_color[1] = 0.8f; dr->base_color[2] = 0.0f; dr->base_color[3] = 1.0f;  /* render_pass.cpp:557-571 */
            dr->alpha_cutoff = 0.5f; dr->metallic = 0.04f; dr->roughness = 0.5f;
            dr->scene = local; dr->flags = SLHIP_DRAW_NO_VERTEX_ID;
            dr->n_verts = 4; dr->vtx_base = 0; dr->idx_base = 0; dr->n_tris = 2; dr->prim_base = prim;
            dr->clip_base = clip;
            chunks[nk].scene = local; chunks[nk].draw = d0 + nd; chunks[nk].first_tri = 0; chunks[nk].count = 2;
            ++nk; ++nd; prim += 2; clip += 4;
        }
        for (uint32_t o = 0; o < p->n_objects; ++o) {
            const slhip_asset* a = assets + objs[o].asset;
            float m2w[16], nm[12];
            mm4(bodies[o].pose, a->mesh_to_object, m2w);
            normal_matrix(m2w, nm);
            for (uint32_t t = 0; t < a->draw_count; ++t) {
                slhip_draw* dr = draws + nd;
                *dr = templates[a->draw_begin + t];
                memcpy(dr->mesh_to_object, a->mesh_to_object, 64);
                memcpy(dr->object_to_world, bodies[o].pose, 64);
                memcpy(dr->normal_to_world, nm, 48);
                if (objs[o].metallic >= 0.0f) dr->metallic = objs[o].metallic;       /* render_shader.cpp:366-377 */
                if (objs[o].roughness >= 0.0f) dr->roughness = objs[o].roughness;
                dr->instance_index = objs[o].instance_index;
                dr->scene = local;
                dr->n_verts = a->n_verts;
                dr->prim_base = prim;
                dr->clip_base = clip;
                for (uint32_t first = 0; first < dr->n_tris; first += SLHIP_CHUNK_TRIS) {
                    const uint32_t left = dr->n_tris - first;
                    chunks[nk].scene = local; chunks[nk].draw = d0 + nd; chunks[nk].first_tri = first;
                    chunks[nk].count = left < SLHIP_CHUNK_TRIS ? left : SLHIP_CHUNK_TRIS;
                    ++nk;
                }
                prim += dr->n_tris;
                clip += a->n_verts;
                ++nd;
            }
        }
        for (uint32_t i = nd; i < p->max_draws_per_scene; ++i) {      /* unused slots: empty draws / chunks */
            memset(draws + i, 0, sizeof(slhip_draw));
            draws[i].scene = local;
        }
        for (uint32_t i = nk; i < p->max_chunks_per_scene; ++i) {
            chunks[i].scene = local; chunks[i].draw = d0; chunks[i].first_tri = 0; chunks[i].count = 0;
        }
        sc->draw_begin = d0;
        sc->draw_end = d0 + nd;
        sc->n_prims = prim;
    }
    return 0;
}

/* the deterministic transcendentals, exported for the accuracy test */
float slref_det_logf(float x) { return det_logf(x); }
void slref_det_sincosf(float x, float* s, float* c) { det_sincosf(x, s, c); }
