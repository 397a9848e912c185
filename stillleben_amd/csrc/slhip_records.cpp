// slhip_records.cpp -- host side of the per-object API in C++ (no device code): what the reference does in C++ behind pybind11
// when Python calls renderer.render(scene) (python/src/py_render_pass.cpp:252-258 -> RenderPass::render, src/render_pass.cpp:
// 303-796: shadow matrices :69-211, per-drawable uniforms :534-621; RenderShader::setTransformations / setMaterial,
// src/shaders/render_shader.cpp:233-265, :326-417).  Here: the slhip_scene / slhip_draw / slhip_chunk records of a BATCH of
// scenes are assembled from flat descriptors of the scenes and their objects plus per-mesh draw templates -- one call per batch
// instead of numpy per scene and object.  float32 throughout, every product and sum in a fixed order (this translation unit is
// compiled with -ffp-contract=off like the rest): the per-scene Python path (_shadow.py, _batch.py) calls the same functions,
// so both paths hand bit-identical records to the kernels and to the oracle.
#include "slhip_common.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace {

struct M4 { float m[16]; };   // row-major

inline M4 identity()
{
    M4 r;
    for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    return r;
}
inline M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a.m[4 * i] * b.m[j];
            s = s + a.m[4 * i + 1] * b.m[4 + j];
            s = s + a.m[4 * i + 2] * b.m[8 + j];
            s = s + a.m[4 * i + 3] * b.m[12 + j];
            r.m[4 * i + j] = s;
        }
    return r;
}
// Matrix4::invertedRigid: [R^T | -R^T t]
inline M4 inverted_rigid(const M4& a)
{
    M4 r = identity();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[4 * i + j] = a.m[4 * j + i];
    for (int i = 0; i < 3; ++i) {
        float s = r.m[4 * i] * a.m[3];
        s = s + r.m[4 * i + 1] * a.m[7];
        s = s + r.m[4 * i + 2] * a.m[11];
        r.m[4 * i + 3] = -s;
    }
    return r;
}
inline void mul_vec4(const M4& a, const float* v, float* out)
{
    for (int i = 0; i < 4; ++i) {
        float s = a.m[4 * i] * v[0];
        s = s + a.m[4 * i + 1] * v[1];
        s = s + a.m[4 * i + 2] * v[2];
        s = s + a.m[4 * i + 3] * v[3];
        out[i] = s;
    }
}
// Matrix4::transformPoint: (R p + t) / (w-row . p + m33)
inline void transform_point(const M4& a, const float* p, float* out)
{
    float q[4];
    for (int i = 0; i < 4; ++i) {
        float s = a.m[4 * i] * p[0];
        s = s + a.m[4 * i + 1] * p[1];
        s = s + a.m[4 * i + 2] * p[2];
        q[i] = s + a.m[4 * i + 3];
    }
    out[0] = q[0] / q[3]; out[1] = q[1] / q[3]; out[2] = q[2] / q[3];
}
inline bool normalize3(float* v)
{
    const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
    return l > 0.0f;
}
inline void cross3(const float* a, const float* b, float* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
inline bool any3(const float* v) { return v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f; }

// Matrix4::normalMatrix(): inverse transpose of the upper 3x3, by cofactors in double (rows padded to 4)
inline void normal_matrix(const M4& a, float* out12)
{
    const double m00 = a.m[0], m01 = a.m[1], m02 = a.m[2], m10 = a.m[4], m11 = a.m[5], m12 = a.m[6], m20 = a.m[8], m21 = a.m[9],
                 m22 = a.m[10];
    const double c00 = m11 * m22 - m12 * m21, c01 = m12 * m20 - m10 * m22, c02 = m10 * m21 - m11 * m20;
    const double c10 = m02 * m21 - m01 * m22, c11 = m00 * m22 - m02 * m20, c12 = m01 * m20 - m00 * m21;
    const double c20 = m01 * m12 - m02 * m11, c21 = m02 * m10 - m00 * m12, c22 = m00 * m11 - m01 * m10;
    const double det = m00 * c00 + m01 * c01 + m02 * c02;
    const double c[9] = {c00, c01, c02, c10, c11, c12, c20, c21, c22};   // inverse transpose = cofactors / det
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) out12[4 * r + k] = (float)(c[3 * r + k] / det);
        out12[4 * r + 3] = 0.0f;
    }
}

// computeFrustumCorners (render_pass.cpp:69-129): the view frustum between the nearest and the farthest bounding sphere
void frustum_corners(const slhip_host_scene& sc, const slhip_host_object* objs, float corners[8][3])
{
    M4 P, Pinv, cam_pose;
    std::memcpy(P.m, sc.proj, 64);
    std::memcpy(Pinv.m, sc.proj_inv, 64);
    std::memcpy(cam_pose.m, sc.camera_pose, 64);
    const M4 cam_matrix = inverted_rigid(cam_pose);
    float nearz = -1.0f, farz = 1.0f;
    if (sc.obj_end > sc.obj_begin) {
        float near_obj = std::numeric_limits<float>::infinity(), far_obj = -std::numeric_limits<float>::infinity();
        for (uint32_t o = sc.obj_begin; o < sc.obj_end; ++o) {
            M4 pose;
            std::memcpy(pose.m, objs[o].pose, 64);
            const M4 t = mul(cam_matrix, pose);
            float c[3];
            transform_point(t, objs[o].bbox_center, c);
            const float radius = objs[o].bbox_center[3];
            const float pn[4] = {c[0], c[1], c[2] - radius, 1.0f}, pf[4] = {c[0], c[1], c[2] + radius, 1.0f};
            float vn[4], vf[4];
            mul_vec4(P, pn, vn);
            mul_vec4(P, pf, vf);
            const float zn = vn[2] / vn[3], zf = vf[2] / vf[3];
            near_obj = std::fmin(near_obj, zn);
            far_obj = std::fmax(far_obj, zf);
        }
        nearz = std::fmax(std::fmax(-1.0f, near_obj), nearz);
        farz = std::fmin(far_obj, farz);
    }
    static const float sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {1, 1, -1, -1, 1, 1, -1, -1};
    const M4 cam_to_world = inverted_rigid(cam_matrix);
    for (int i = 0; i < 8; ++i) {
        const float h[4] = {sx[i], sy[i], i < 4 ? nearz : farz, 1.0f};
        float a[4], p[4];
        mul_vec4(Pinv, h, a);
        mul_vec4(cam_to_world, a, p);
        corners[i][0] = p[0] / p[3]; corners[i][1] = p[1] / p[3]; corners[i][2] = p[2] / p[3];
    }
}

// computeShadowMapMatrix (render_pass.cpp:131-211): orthographic fit of the frustum in the light's frame, depth range x 5,
// x / y clamped to the objects' bounding spheres.  Returns false when the result is not finite.
bool shadow_matrix(const slhip_host_scene& sc, const slhip_host_object* objs, const float corners[8][3], const float* light_dir, float* out16)
{
    float z[3] = {light_dir[0], light_dir[1], light_dir[2]};
    normalize3(z);
    const float up[3] = {0.0f, 0.0f, 1.0f};
    float x[3], y[3];
    cross3(z, up, x);
    normalize3(x);
    cross3(z, x, y);
    normalize3(y);
    M4 l2w = identity();
    for (int r = 0; r < 3; ++r) { l2w.m[4 * r] = x[r]; l2w.m[4 * r + 1] = y[r]; l2w.m[4 * r + 2] = z[r]; }
    const M4 w2l = inverted_rigid(l2w);
    const float inf = std::numeric_limits<float>::infinity();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    for (int i = 0; i < 8; ++i) {
        float p[3];
        transform_point(w2l, corners[i], p);
        for (int k = 0; k < 3; ++k) { mn[k] = std::fmin(mn[k], p[k]); mx[k] = std::fmax(mx[k], p[k]); }
    }
    float nearz = mn[2], farz = mx[2];
    const float mean_z = (nearz + farz) / 2.0f;
    const float spread = farz - mean_z;
    farz = mean_z + 5.0f * spread;
    nearz = mean_z - 5.0f * spread;
    float L = mn[0], R = mx[0], T = mn[1], B = mx[1];
    if (sc.obj_end > sc.obj_begin) {
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        for (uint32_t o = sc.obj_begin; o < sc.obj_end; ++o) {
            M4 pose;
            std::memcpy(pose.m, objs[o].pose, 64);
            const M4 t = mul(w2l, pose);
            float c[3];
            transform_point(t, objs[o].bbox_center, c);
            const float radius = objs[o].bbox_center[3];
            for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], c[k] - radius); hi[k] = std::fmax(hi[k], c[k] + radius); }
        }
        L = std::fmax(L, lo[0]); R = std::fmin(R, hi[0]);
        T = std::fmax(T, lo[1]); B = std::fmin(B, hi[1]);
    }
    M4 P;
    std::memset(P.m, 0, 64);
    P.m[0] = 2.0f / (R - L); P.m[3] = -(R + L) / (R - L);
    P.m[5] = 2.0f / (B - T); P.m[7] = -(B + T) / (B - T);
    P.m[10] = 2.0f / (farz - nearz); P.m[11] = -(farz + nearz) / (farz - nearz);
    P.m[15] = 1.0f;
    const M4 r = mul(P, w2l);
    bool ok = true;
    for (int i = 0; i < 16; ++i) ok = ok && std::isfinite(r.m[i]);
    if (ok) std::memcpy(out16, r.m, 64);
    return ok;
}

void scene_shadow_matrices(const slhip_host_scene& sc, const slhip_host_object* objs, float out[SLHIP_NUM_LIGHTS][16])
{
    const M4 I = identity();
    bool have_corners = false;
    float corners[8][3];
    for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l) {
        std::memcpy(out[l], I.m, 64);
        if (!any3(sc.light_color[l]) || !any3(sc.light_dir[l])) continue;
        if (!have_corners) { frustum_corners(sc, objs, corners); have_corners = true; }
        shadow_matrix(sc, objs, corners, sc.light_dir[l], out[l]);
    }
}

}  // namespace

extern "C" int slhip_host_shadow_matrices(const slhip_host_scene* scene, const slhip_host_object* objects, float* out48)
{
    if (!scene || !out48 || (scene->obj_end > scene->obj_begin && !objects)) {
        slhip::set_error("slhip_host_shadow_matrices: null argument");
        return -1;
    }
    scene_shadow_matrices(*scene, objects, reinterpret_cast<float(*)[16]>(out48));
    return 0;
}

extern "C" int slhip_host_normal_matrix(const float* m16, float* out12)
{
    if (!m16 || !out12) {
        slhip::set_error("slhip_host_normal_matrix: null argument");
        return -1;
    }
    M4 a;
    std::memcpy(a.m, m16, 64);
    normal_matrix(a, out12);
    return 0;
}

extern "C" int slhip_records_count(const slhip_host_scene* scenes, uint32_t n_scenes, const slhip_host_object* objects,
                                   const slhip_draw* templates, uint32_t* n_draws, uint32_t* n_chunks)
{
    if (!scenes || !n_draws || !n_chunks) {
        slhip::set_error("slhip_records_count: null argument");
        return -1;
    }
    uint64_t nd = 0, nc = 0;
    for (uint32_t s = 0; s < n_scenes; ++s) {
        const slhip_host_scene& sc = scenes[s];
        if (sc.plane_template >= 0) { nd += 1; nc += 1; }
        for (uint32_t o = sc.obj_begin; o < sc.obj_end; ++o)
            for (uint32_t k = 0; k < objects[o].tmpl_count; ++k) {
                nd += 1;
                nc += (templates[objects[o].tmpl_begin + k].n_tris + SLHIP_CHUNK_TRIS - 1) / SLHIP_CHUNK_TRIS;
            }
    }
    if (nd > 0xffffffffull || nc > 0xffffffffull) {
        slhip::set_error("slhip_records_count: batch too large");
        return -1;
    }
    *n_draws = (uint32_t)nd; *n_chunks = (uint32_t)nc;
    return 0;
}

extern "C" int slhip_records_build_render(const slhip_host_scene* scenes, uint32_t n_scenes, const slhip_host_object* objects,
                                          const slhip_draw* templates, uint32_t with_shadows, slhip_scene* srec, slhip_draw* drec,
                                          uint32_t draw_capacity, slhip_chunk* crec, uint32_t chunk_capacity)
{
    if (!scenes || !srec || (draw_capacity && !drec) || (chunk_capacity && !crec)) {
        slhip::set_error("slhip_records_build_render: null argument");
        return -1;
    }
    uint32_t nd = 0, nc = 0;
    uint64_t clip = 0;
    for (uint32_t s = 0; s < n_scenes; ++s) {
        const slhip_host_scene& sc = scenes[s];
        slhip_scene& R = srec[s];
        std::memset(&R, 0, sizeof(R));
        std::memcpy(R.proj, sc.proj, 64);
        M4 cam_pose;
        std::memcpy(cam_pose.m, sc.camera_pose, 64);
        const M4 w2c = inverted_rigid(cam_pose);
        std::memcpy(R.world_to_cam, w2c.m, 64);
        const M4 c2w = inverted_rigid(w2c);          // camPosition = worldToCam.invertedRigid().translation() (render_shader.cpp:246)
        R.cam_position[0] = c2w.m[3]; R.cam_position[1] = c2w.m[7]; R.cam_position[2] = c2w.m[11]; R.cam_position[3] = 1.0f;
        for (int l = 0; l < SLHIP_NUM_LIGHTS; ++l)
            for (int k = 0; k < 3; ++k) { R.light_dir[l][k] = sc.light_dir[l][k]; R.light_color[l][k] = sc.light_color[l][k]; }
        for (int k = 0; k < 3; ++k) R.ambient[k] = sc.ambient[k];
        R.manual_exposure = sc.manual_exposure;
        R.light_map = sc.light_map;
        R.bg_tex[0] = sc.bg_tex[0]; R.bg_tex[1] = sc.bg_tex[1]; R.bg_tex[2] = sc.bg_tex[2];
        if (with_shadows) scene_shadow_matrices(sc, objects, R.shadow_mat);
        R.draw_begin = nd;
        uint32_t prim = 0;
        auto emit = [&](const slhip_draw& d) -> bool {
            if (nd >= draw_capacity) return false;
            drec[nd] = d;
            slhip_draw& D = drec[nd];
            D.scene = s;
            D.prim_base = prim;
            D.clip_base = (uint32_t)clip;
            clip += D.n_verts;
            for (uint32_t first = 0; first < D.n_tris; first += SLHIP_CHUNK_TRIS) {
                if (nc >= chunk_capacity) return false;
                slhip_chunk c;
                c.scene = s; c.draw = nd; c.first_tri = first;
                c.count = D.n_tris - first < (uint32_t)SLHIP_CHUNK_TRIS ? D.n_tris - first : (uint32_t)SLHIP_CHUNK_TRIS;
                crec[nc++] = c;
            }
            prim += D.n_tris;
            ++nd;
            return true;
        };
        if (sc.plane_template >= 0) {
            // background plane first (render_pass.cpp:545-582): the unit plane scaled to plane_size / 2
            slhip_draw d = templates[sc.plane_template];
            M4 pose, scal = identity();
            std::memcpy(pose.m, sc.plane_pose, 64);
            scal.m[0] = sc.plane_size[0] / 2.0f;
            scal.m[5] = sc.plane_size[1] / 2.0f;
            const M4 o2w = mul(pose, scal);
            const M4 I = identity();
            std::memcpy(d.mesh_to_object, I.m, 64);
            std::memcpy(d.object_to_world, o2w.m, 64);
            normal_matrix(o2w, d.normal_to_world);
            if (!emit(d)) { slhip::set_error("slhip_records_build_render: record capacity too small"); return -1; }
        }
        for (uint32_t o = sc.obj_begin; o < sc.obj_end; ++o) {
            const slhip_host_object& ob = objects[o];
            M4 pose;
            std::memcpy(pose.m, ob.pose, 64);
            for (uint32_t k = 0; k < ob.tmpl_count; ++k) {
                slhip_draw d = templates[ob.tmpl_begin + k];
                M4 m2o;
                std::memcpy(m2o.m, d.mesh_to_object, 64);
                std::memcpy(d.object_to_world, ob.pose, 64);
                normal_matrix(mul(pose, m2o), d.normal_to_world);
                if (ob.force_color) std::memcpy(d.base_color, ob.color, 16);
                if (ob.metallic >= 0.0f) d.metallic = ob.metallic;      // RenderShader::setMaterial overrides (render_shader.cpp:355-377)
                if (ob.roughness >= 0.0f) d.roughness = ob.roughness;
                d.instance_index = ob.instance_index;
                if (ob.casts_shadows) d.flags |= SLHIP_DRAW_CASTS_SHADOW; else d.flags &= ~(uint32_t)SLHIP_DRAW_CASTS_SHADOW;
                if (!emit(d)) { slhip::set_error("slhip_records_build_render: record capacity too small"); return -1; }
            }
        }
        R.draw_end = nd;
        R.n_prims = prim;
    }
    if (clip > 0xffffffffull) {
        slhip::set_error("slhip_records_build_render: more than 2^32 clip positions in one batch");
        return -1;
    }
    return 0;
}

static_assert(sizeof(slhip_host_object) == 128, "slhip_host_object layout");
static_assert(sizeof(slhip_host_scene) == 408, "slhip_host_scene layout");
