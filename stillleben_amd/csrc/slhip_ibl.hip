// Image-based lighting precompute ('next' row f1): what LightMap::load renders with OpenGL
// (reference src/light_map.cpp:360-606 and src/shaders/cubemap_shader_*.frag, brdf_shader.frag),
// as gfx950 kernels writing the slhip_light_map buffers.
#include <hip/hip_runtime.h>

#include "slhip.h"
#include "slhip_common.h"
#include "slhip_cubemap.h"

namespace {

using slcube::f3;
using slcube::F3;

constexpr float kPi = 3.14159265359f;   // the shaders' constant

__device__ __forceinline__ f3 normalize(f3 v)
{
    const float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return F3(v.x / l, v.y / l, v.z / l);
}
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// direction of texel (i, j) of `face` at size n (centre of the texel)
__device__ __forceinline__ f3 texel_dir(int face, int i, int j, int n)
{
    const float sc = (2.0f * ((float)i + 0.5f)) / (float)n - 1.0f, tc = (2.0f * ((float)j + 0.5f)) / (float)n - 1.0f;
    return slcube::face_to_dir(face, sc, tc);
}

// ---- equirectangular -> cube (cubemap_shader_equirectangular.frag): uv = (atan(y, x), asin(z)) * invAtan + 0.5
__global__ __launch_bounds__(256) void k_equirect_to_cube(const float* __restrict__ eq, int H, int W, float* __restrict__ env, int n)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 6 * n * n) return;
    const int face = idx / (n * n), j = (idx / n) % n, i = idx % n;
    const f3 v = normalize(texel_dir(face, i, j, n));
    const float u = atan2f(v.y, v.x) * 0.1591f + 0.5f, w = asinf(v.z) * 0.3183f + 0.5f;
    // bilinear, clamp to edge; image row 0 is the top (w = 1)
    const float x = u * (float)W - 0.5f, y = (1.0f - w) * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int x0 = min(max((int)fx, 0), W - 1), x1 = min(max((int)fx + 1, 0), W - 1);
    const int y0 = min(max((int)fy, 0), H - 1), y1 = min(max((int)fy + 1, 0), H - 1);
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c00 = eq[((size_t)y0 * W + x0) * 3 + k], c10 = eq[((size_t)y0 * W + x1) * 3 + k];
        const float c01 = eq[((size_t)y1 * W + x0) * 3 + k], c11 = eq[((size_t)y1 * W + x1) * 3 + k];
        const float top = fmaf(a, c10 - c00, c00), bot = fmaf(a, c11 - c01, c01);
        c[k] = fmaf(b, bot - top, top);
    }
    reinterpret_cast<float4*>(env)[idx] = make_float4(c[0], c[1], c[2], 1.0f);
}

// ---- mip chain (glGenerateMipmap: 2x2 box per face) ----
__global__ __launch_bounds__(256) void k_cube_mip(const float* __restrict__ src, float* __restrict__ dst, int n)   // n = destination size
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 6 * n * n) return;
    const int face = idx / (n * n), j = (idx / n) % n, i = idx % n;
    const float4* s = reinterpret_cast<const float4*>(src) + (size_t)face * (2 * n) * (2 * n);
    const float4 a = s[(size_t)(2 * j) * (2 * n) + 2 * i], b = s[(size_t)(2 * j) * (2 * n) + 2 * i + 1];
    const float4 c = s[(size_t)(2 * j + 1) * (2 * n) + 2 * i], d = s[(size_t)(2 * j + 1) * (2 * n) + 2 * i + 1];
    reinterpret_cast<float4*>(dst)[idx] = make_float4(((a.x + b.x) + (c.x + d.x)) * 0.25f, ((a.y + b.y) + (c.y + d.y)) * 0.25f,
                                                     ((a.z + b.z) + (c.z + d.z)) * 0.25f, ((a.w + b.w) + (c.w + d.w)) * 0.25f);
}

// ---- diffuse irradiance (cubemap_shader_irradiance.frag) ----
__global__ __launch_bounds__(64) void k_irradiance(const float* __restrict__ env, int env_n, float* __restrict__ irr, int n)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= 6 * n * n) return;
    const int face = idx / (n * n), j = (idx / n) % n, i = idx % n;
    const f3 N = normalize(texel_dir(face, i, j, n));
    f3 up = F3(0.0f, 1.0f, 0.0f);
    const f3 right = cross(up, N);
    up = cross(N, right);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    float nr = 0.0f;
    const float delta = 0.020f;
    for (float phi = 0.0f; phi < 2.0f * kPi; phi += delta) {
        const float sp = sinf(phi), cp = cosf(phi);
        for (float theta = 0.0f; theta < 0.5f * kPi; theta += delta) {
            const float st = sinf(theta), ct = cosf(theta);
            const float tx = st * cp, ty = st * sp, tz = ct;
            const f3 sv = F3(tx * right.x + ty * up.x + tz * N.x, tx * right.y + ty * up.y + tz * N.y, tx * right.z + ty * up.z + tz * N.z);
            const float4 c = slcube::sample_level(env, env_n, sv);
            acc[0] += c.x * ct * st; acc[1] += c.y * ct * st; acc[2] += c.z * ct * st;
            nr += 1.0f;
        }
    }
    const float k = 1.0f / nr;
    reinterpret_cast<float4*>(irr)[idx] = make_float4(kPi * acc[0] * k, kPi * acc[1] * k, kPi * acc[2] * k, 1.0f);
}

// ---- shared by prefilter and BRDF LUT ----
__device__ __forceinline__ float radical_inverse(unsigned bits)
{
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

__device__ __forceinline__ f3 importance_sample_ggx(float xi_x, float xi_y, f3 N, float roughness)
{
    const float a = roughness * roughness;
    const float phi = 2.0f * kPi * xi_x;
    const float cos_t = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    const float sin_t = sqrtf(1.0f - cos_t * cos_t);
    const float hx = cosf(phi) * sin_t, hy = sinf(phi) * sin_t, hz = cos_t;
    const f3 up = fabsf(N.z) < 0.999f ? F3(0.0f, 0.0f, 1.0f) : F3(1.0f, 0.0f, 0.0f);
    const f3 tangent = normalize(cross(up, N));
    const f3 bitangent = cross(N, tangent);
    return normalize(F3(tangent.x * hx + bitangent.x * hy + N.x * hz, tangent.y * hx + bitangent.y * hy + N.y * hz,
                        tangent.z * hx + bitangent.z * hy + N.z * hz));
}

// ---- GGX prefilter (cubemap_shader_prefilter.frag); one launch per level, roughness = level / (levels - 1) ----
__global__ __launch_bounds__(64) void k_prefilter(const float* __restrict__ env, unsigned env_n, unsigned env_levels,
                                                  float* __restrict__ dst, int n, float roughness)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= 6 * n * n) return;
    const int face = idx / (n * n), j = (idx / n) % n, i = idx % n;
    const f3 N = normalize(texel_dir(face, i, j, n));
    const f3 V = N;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    float total = 0.0f;
    const unsigned kSamples = 1024u;
    for (unsigned s = 0; s < kSamples; ++s) {
        const f3 Hh = importance_sample_ggx((float)s / (float)kSamples, radical_inverse(s), N, roughness);
        const float vh2 = 2.0f * dot(V, Hh);
        const f3 L = normalize(F3(vh2 * Hh.x - V.x, vh2 * Hh.y - V.y, vh2 * Hh.z - V.z));
        const float NdotL = fmaxf(dot(N, L), 0.0f);
        if (NdotL > 0.0f) {
            const float a = roughness * roughness, a2 = a * a;
            const float NdotH = fmaxf(dot(N, Hh), 0.0f), HdotV = fmaxf(dot(Hh, V), 0.0f);
            float denom = NdotH * NdotH * (a2 - 1.0f) + 1.0f;
            denom = kPi * denom * denom;
            const float D = a2 / denom;
            const float pdf = D * NdotH / (4.0f * HdotV) + 0.0001f;
            const float resolution = 512.0f;   // the shader's constant (resolution of the source cube map)
            const float sa_texel = 4.0f * kPi / (6.0f * resolution * resolution);
            const float sa_sample = 1.0f / ((float)kSamples * pdf + 0.0001f);
            const float mip = roughness == 0.0f ? 0.0f : 0.5f * log2f(sa_sample / sa_texel);
            const float4 c = slcube::sample_lod(env, env_n, env_levels, L, mip);
            acc[0] += c.x * NdotL; acc[1] += c.y * NdotL; acc[2] += c.z * NdotL;
            total += NdotL;
        }
    }
    reinterpret_cast<float4*>(dst)[idx] = make_float4(acc[0] / total, acc[1] / total, acc[2] / total, 1.0f);
}

// ---- split-sum BRDF table (brdf_shader.frag) ----
__device__ __forceinline__ float g_schlick_ibl(float ndv, float roughness)
{
    const float k = (roughness * roughness) / 2.0f;
    return ndv / (ndv * (1.0f - k) + k);
}

__global__ __launch_bounds__(256) void k_brdf_lut(float* __restrict__ lut, int n)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int j = idx / n, i = idx % n;
    const float NdotV = ((float)i + 0.5f) / (float)n, roughness = ((float)j + 0.5f) / (float)n;
    const f3 V = F3(sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV);
    const f3 N = F3(0.0f, 0.0f, 1.0f);
    float A = 0.0f, B = 0.0f;
    const unsigned kSamples = 1024u;
    for (unsigned s = 0; s < kSamples; ++s) {
        const f3 Hh = importance_sample_ggx((float)s / (float)kSamples, radical_inverse(s), N, roughness);
        const float vh2 = 2.0f * dot(V, Hh);
        const f3 L = normalize(F3(vh2 * Hh.x - V.x, vh2 * Hh.y - V.y, vh2 * Hh.z - V.z));
        const float NdotL = fmaxf(L.z, 0.0f), NdotH = fmaxf(Hh.z, 0.0f), VdotH = fmaxf(dot(V, Hh), 0.0f);
        if (NdotL > 0.0f) {
            const float G = g_schlick_ibl(NdotL, roughness) * g_schlick_ibl(fmaxf(dot(N, V), 0.0f), roughness);
            const float G_vis = (G * VdotH) / (NdotH * NdotV);
            const float Fc = powf(1.0f - VdotH, 5.0f);
            A += (1.0f - Fc) * G_vis;
            B += Fc * G_vis;
        }
    }
    lut[2 * idx] = A / (float)kSamples;
    lut[2 * idx + 1] = B / (float)kSamples;
}

uint64_t cube_floats(uint32_t size, uint32_t levels)
{
    uint64_t o = 0;
    for (uint32_t l = 0; l < levels; ++l) { const uint64_t m = size >> l; o += 24 * m * m; }
    return o;
}

}  // namespace

extern "C" int slhip_light_map_floats(uint32_t env_size, uint32_t env_levels, uint32_t irr_size, uint32_t pre_size,
                                      uint32_t pre_levels, uint32_t lut_size, uint64_t out[4])
{
    if (!out || env_levels == 0 || pre_levels == 0 || (env_size >> (env_levels - 1)) == 0 || (pre_size >> (pre_levels - 1)) == 0) {
        slhip::set_error("slhip_light_map_floats: bad sizes");
        return -1;
    }
    out[0] = cube_floats(env_size, env_levels);
    out[1] = cube_floats(irr_size, 1);
    out[2] = cube_floats(pre_size, pre_levels);
    out[3] = 2ull * lut_size * lut_size;
    return 0;
}

extern "C" int slhip_light_map_build(const float* d_equirect, int H, int W, const slhip_light_map* lm, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_equirect || !lm || !lm->d_env || !lm->d_irradiance || !lm->d_prefilter || !lm->d_brdf_lut) {
        slhip::set_error("slhip_light_map_build: null argument");
        return -1;
    }
    uint64_t need[4];
    if (H <= 0 || W <= 0 ||
        slhip_light_map_floats(lm->env_size, lm->env_levels, lm->irr_size, lm->pre_size, lm->pre_levels, lm->lut_size, need) != 0 ||
        lm->irr_size == 0 || lm->lut_size == 0) {
        slhip::set_error("slhip_light_map_build: bad sizes");
        return -1;
    }
    const int n = (int)lm->env_size;
    k_equirect_to_cube<<<(6 * n * n + 255) / 256, 256, 0, stream>>>(d_equirect, H, W, lm->d_env, n);
    for (uint32_t l = 1; l < lm->env_levels; ++l) {
        const int m = n >> l;
        k_cube_mip<<<(6 * m * m + 255) / 256, 256, 0, stream>>>(lm->d_env + cube_floats(lm->env_size, l - 1), lm->d_env + cube_floats(lm->env_size, l), m);
    }
    const int ni = (int)lm->irr_size;
    k_irradiance<<<(6 * ni * ni + 63) / 64, 64, 0, stream>>>(lm->d_env, n, lm->d_irradiance, ni);
    for (uint32_t l = 0; l < lm->pre_levels; ++l) {
        const int m = (int)(lm->pre_size >> l);
        const float roughness = lm->pre_levels > 1 ? (float)l / (float)(lm->pre_levels - 1) : 0.0f;
        k_prefilter<<<(6 * m * m + 63) / 64, 64, 0, stream>>>(lm->d_env, lm->env_size, lm->env_levels,
                                                              lm->d_prefilter + cube_floats(lm->pre_size, l), m, roughness);
    }
    const int nl = (int)lm->lut_size;
    k_brdf_lut<<<(nl * nl + 255) / 256, 256, 0, stream>>>(lm->d_brdf_lut, nl);
    SLHIP_LAUNCH_CHECK();
    return 0;
}
