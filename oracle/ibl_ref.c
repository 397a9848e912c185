/* CPU restatement of the image-based-lighting precompute ('next' row f1): LightMap::load's GL passes
 * (reference src/light_map.cpp:360-606) with the shaders cubemap_shader_equirectangular.frag,
 * cubemap_shader_irradiance.frag, cubemap_shader_prefilter.frag and brdf_shader.frag.
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into oracle/.  PARITY UNPINNED with respect to the reference's OpenGL output: texture filtering, LOD
 * selection and mip generation are implementation-defined there; the rules used instead are documented
 * at slhip_light_map in include/slhip.h.  The HIP kernels follow the same operation order; libm vs
 * device transcendentals (sin, cos, atan2, asin, log2, pow) differ in the last bits, hence the
 * tolerances in tests/test_gpu_ibl.py. */
#include <math.h>
#include <stdint.h>

#include "../include/slhip.h"
#include "cubemap_ref.h"

#define IBL_PI 3.14159265359f

static cm3 nrm3(cm3 v)
{
    const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    cm3 r = {v.x / l, v.y / l, v.z / l};
    return r;
}
static cm3 crs3(cm3 a, cm3 b) { cm3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
static float dt3(cm3 a, cm3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

static cm3 texel_dir(int face, int i, int j, int n)
{
    const float sc = (2.0f * ((float)i + 0.5f)) / (float)n - 1.0f, tc = (2.0f * ((float)j + 0.5f)) / (float)n - 1.0f;
    return cm_face_to_dir(face, sc, tc);
}

static uint64_t cube_floats(uint32_t size, uint32_t levels)
{
    uint64_t o = 0;
    for (uint32_t l = 0; l < levels; ++l) { uint64_t m = size >> l; o += 24 * m * m; }
    return o;
}

static float radical_inverse(uint32_t bits)
{
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

static cm3 importance_sample_ggx(float xi_x, float xi_y, cm3 N, float roughness)
{
    const float a = roughness * roughness;
    const float phi = 2.0f * IBL_PI * xi_x;
    const float cos_t = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    const float sin_t = sqrtf(1.0f - cos_t * cos_t);
    const float hx = cosf(phi) * sin_t, hy = sinf(phi) * sin_t, hz = cos_t;
    cm3 up = {0.0f, 0.0f, 1.0f};
    if (!(fabsf(N.z) < 0.999f)) { up.x = 1.0f; up.z = 0.0f; }
    const cm3 tangent = nrm3(crs3(up, N));
    const cm3 bitangent = crs3(N, tangent);
    cm3 s = {tangent.x * hx + bitangent.x * hy + N.x * hz, tangent.y * hx + bitangent.y * hy + N.y * hz,
             tangent.z * hx + bitangent.z * hy + N.z * hz};
    return nrm3(s);
}

static float g_schlick_ibl(float ndv, float roughness)
{
    const float k = (roughness * roughness) / 2.0f;
    return ndv / (ndv * (1.0f - k) + k);
}

/* lm holds HOST pointers here */
int slref_light_map_build(const float* eq, int H, int W, const slhip_light_map* lm)
{
    const int n = (int)lm->env_size;
    /* equirectangular -> cube */
    for (int face = 0; face < 6; ++face)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                const cm3 v = nrm3(texel_dir(face, i, j, n));
                const float u = atan2f(v.y, v.x) * 0.1591f + 0.5f, w = asinf(v.z) * 0.3183f + 0.5f;
                const float x = u * (float)W - 0.5f, y = (1.0f - w) * (float)H - 0.5f;
                const float fx = floorf(x), fy = floorf(y);
                const float a = x - fx, b = y - fy;
                int x0 = (int)fx, x1 = (int)fx + 1, y0 = (int)fy, y1 = (int)fy + 1;
                if (x0 < 0) x0 = 0; if (x0 > W - 1) x0 = W - 1; if (x1 < 0) x1 = 0; if (x1 > W - 1) x1 = W - 1;
                if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1; if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
                float* dst = lm->d_env + 4 * (((size_t)face * n + j) * n + i);
                for (int k = 0; k < 3; ++k)
                    dst[k] = cm_bil(a, b, eq[((size_t)y0 * W + x0) * 3 + k], eq[((size_t)y0 * W + x1) * 3 + k],
                                    eq[((size_t)y1 * W + x0) * 3 + k], eq[((size_t)y1 * W + x1) * 3 + k]);
                dst[3] = 1.0f;
            }
    /* mip chain */
    for (uint32_t l = 1; l < lm->env_levels; ++l) {
        const int m = n >> l;
        const float* src = lm->d_env + cube_floats(lm->env_size, l - 1);
        float* dst = lm->d_env + cube_floats(lm->env_size, l);
        for (int face = 0; face < 6; ++face)
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < m; ++i)
                    for (int k = 0; k < 4; ++k) {
                        const float* s = src + 4 * ((size_t)face * (2 * m) * (2 * m));
                        const float a = s[4 * ((size_t)(2 * j) * (2 * m) + 2 * i) + k], b = s[4 * ((size_t)(2 * j) * (2 * m) + 2 * i + 1) + k];
                        const float c = s[4 * ((size_t)(2 * j + 1) * (2 * m) + 2 * i) + k], d = s[4 * ((size_t)(2 * j + 1) * (2 * m) + 2 * i + 1) + k];
                        dst[4 * (((size_t)face * m + j) * m + i) + k] = ((a + b) + (c + d)) * 0.25f;
                    }
    }
    /* diffuse irradiance */
    const int ni = (int)lm->irr_size;
    for (int face = 0; face < 6; ++face)
        for (int j = 0; j < ni; ++j)
            for (int i = 0; i < ni; ++i) {
                const cm3 N = nrm3(texel_dir(face, i, j, ni));
                cm3 up = {0.0f, 1.0f, 0.0f};
                const cm3 right = crs3(up, N);
                up = crs3(N, right);
                float acc[3] = {0.0f, 0.0f, 0.0f}, nr = 0.0f;
                const float delta = 0.020f;
                for (float phi = 0.0f; phi < 2.0f * IBL_PI; phi += delta) {
                    const float sp = sinf(phi), cp = cosf(phi);
                    for (float theta = 0.0f; theta < 0.5f * IBL_PI; theta += delta) {
                        const float st = sinf(theta), ct = cosf(theta);
                        const float tx = st * cp, ty = st * sp, tz = ct;
                        cm3 sv = {tx * right.x + ty * up.x + tz * N.x, tx * right.y + ty * up.y + tz * N.y, tx * right.z + ty * up.z + tz * N.z};
                        const cm4 c = cm_sample_level(lm->d_env, n, sv);
                        acc[0] += c.x * ct * st; acc[1] += c.y * ct * st; acc[2] += c.z * ct * st;
                        nr += 1.0f;
                    }
                }
                const float k = 1.0f / nr;
                float* dst = lm->d_irradiance + 4 * (((size_t)face * ni + j) * ni + i);
                dst[0] = IBL_PI * acc[0] * k; dst[1] = IBL_PI * acc[1] * k; dst[2] = IBL_PI * acc[2] * k; dst[3] = 1.0f;
            }
    /* GGX prefilter */
    const unsigned kSamples = 1024u;
    for (uint32_t l = 0; l < lm->pre_levels; ++l) {
        const int m = (int)(lm->pre_size >> l);
        const float roughness = lm->pre_levels > 1 ? (float)l / (float)(lm->pre_levels - 1) : 0.0f;
        float* dstl = lm->d_prefilter + cube_floats(lm->pre_size, l);
        for (int face = 0; face < 6; ++face)
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < m; ++i) {
                    const cm3 N = nrm3(texel_dir(face, i, j, m));
                    const cm3 V = N;
                    float acc[3] = {0.0f, 0.0f, 0.0f}, total = 0.0f;
                    for (unsigned s = 0; s < kSamples; ++s) {
                        const cm3 Hh = importance_sample_ggx((float)s / (float)kSamples, radical_inverse(s), N, roughness);
                        const float vh2 = 2.0f * dt3(V, Hh);
                        cm3 Lr = {vh2 * Hh.x - V.x, vh2 * Hh.y - V.y, vh2 * Hh.z - V.z};
                        const cm3 L = nrm3(Lr);
                        const float NdotL = fmaxf(dt3(N, L), 0.0f);
                        if (NdotL > 0.0f) {
                            const float a = roughness * roughness, a2 = a * a;
                            const float NdotH = fmaxf(dt3(N, Hh), 0.0f), HdotV = fmaxf(dt3(Hh, V), 0.0f);
                            float denom = NdotH * NdotH * (a2 - 1.0f) + 1.0f;
                            denom = IBL_PI * denom * denom;
                            const float D = a2 / denom;
                            const float pdf = D * NdotH / (4.0f * HdotV) + 0.0001f;
                            const float resolution = 512.0f;
                            const float sa_texel = 4.0f * IBL_PI / (6.0f * resolution * resolution);
                            const float sa_sample = 1.0f / ((float)kSamples * pdf + 0.0001f);
                            const float mip = roughness == 0.0f ? 0.0f : 0.5f * log2f(sa_sample / sa_texel);
                            const cm4 c = cm_sample_lod(lm->d_env, lm->env_size, lm->env_levels, L, mip);
                            acc[0] += c.x * NdotL; acc[1] += c.y * NdotL; acc[2] += c.z * NdotL;
                            total += NdotL;
                        }
                    }
                    float* dst = dstl + 4 * (((size_t)face * m + j) * m + i);
                    dst[0] = acc[0] / total; dst[1] = acc[1] / total; dst[2] = acc[2] / total; dst[3] = 1.0f;
                }
    }
    /* split-sum BRDF table */
    const int nl = (int)lm->lut_size;
    for (int j = 0; j < nl; ++j)
        for (int i = 0; i < nl; ++i) {
            const float NdotV = ((float)i + 0.5f) / (float)nl, roughness = ((float)j + 0.5f) / (float)nl;
            const cm3 V = {sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV};
            const cm3 N = {0.0f, 0.0f, 1.0f};
            float A = 0.0f, B = 0.0f;
            for (unsigned s = 0; s < kSamples; ++s) {
                const cm3 Hh = importance_sample_ggx((float)s / (float)kSamples, radical_inverse(s), N, roughness);
                const float vh2 = 2.0f * dt3(V, Hh);
                cm3 Lr = {vh2 * Hh.x - V.x, vh2 * Hh.y - V.y, vh2 * Hh.z - V.z};
                const cm3 L = nrm3(Lr);
                const float NdotL = fmaxf(L.z, 0.0f), NdotH = fmaxf(Hh.z, 0.0f), VdotH = fmaxf(dt3(V, Hh), 0.0f);
                if (NdotL > 0.0f) {
                    const float G = g_schlick_ibl(NdotL, roughness) * g_schlick_ibl(fmaxf(dt3(N, V), 0.0f), roughness);
                    const float G_vis = (G * VdotH) / (NdotH * NdotV);
                    const float Fc = powf(1.0f - VdotH, 5.0f);
                    A += (1.0f - Fc) * G_vis;
                    B += Fc * G_vis;
                }
            }
            lm->d_brdf_lut[2 * ((size_t)j * nl + i)] = A / (float)kSamples;
            lm->d_brdf_lut[2 * ((size_t)j * nl + i) + 1] = B / (float)kSamples;
        }
    return 0;
}
