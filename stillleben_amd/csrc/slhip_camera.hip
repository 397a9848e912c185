// Camera noise model ('next' row f4): python/stillleben/camera_model.py:222-263 of the reference as two
// fused gfx950 kernels.  Elementwise stages follow the reference's float32 operation order exactly;
// the two 5x5 convolutions and the bilinear resampling accumulate in a fixed order of their own
// (torch's CPU kernels do not document theirs), so parity with the reference is to ~1e-6.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "slhip.h"
#include "slhip_common.h"

namespace {

static_assert(sizeof(slhip_camera_params) == 268, "slhip_camera_params layout");

// ---- chromatic aberration: affine_grid (align_corners=False) + grid_sample(bilinear, reflection) ----
// torch: base coordinate of pixel i = linspace(-1, 1, n)[i] * (n - 1) / n; grid = s * base + t;
// unnormalise ((g + 1) * n - 1) / 2; reflect about [-0.5, n - 0.5]; clip to [0, n - 1]
__device__ __forceinline__ float base_coord(int i, int n)
{
    if (n <= 1) return 0.0f;
    const float step = 2.0f / (float)(n - 1);
    const float lin = i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
    return lin * (float)(n - 1) / (float)n;
}

__device__ __forceinline__ float reflect_clip(float x, int n)
{
    const float lo = -0.5f, span = (float)n;
    float in = fabsf(x - lo);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    float r = (flips & 1) == 0 ? extra + lo : span - extra + lo;
    return fminf((float)(n - 1), fmaxf(r, 0.0f));
}

__device__ __forceinline__ float chroma_sample(const float* __restrict__ plane, int H, int W, int x, int y, float s,
                                               float tx, float ty)
{
    const float gx = fmaf(s, base_coord(x, W), tx), gy = fmaf(s, base_coord(y, H), ty);
    const float ix = reflect_clip(((gx + 1.0f) * (float)W - 1.0f) / 2.0f, W);
    const float iy = reflect_clip(((gy + 1.0f) * (float)H - 1.0f) / 2.0f, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const float we = ix - fx, ww = 1.0f - we, ws = iy - fy, wn = 1.0f - ws;
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const bool bx1 = x1 < W, by1 = y1 < H;   // x0, y0 are in range after the clip
    const float nw = plane[y0 * W + x0];
    const float ne = bx1 ? plane[y0 * W + x1] : 0.0f;
    const float sw = by1 ? plane[y1 * W + x0] : 0.0f;
    const float se = (bx1 && by1) ? plane[y1 * W + x1] : 0.0f;
    return ((nw * (wn * ww) + ne * (wn * we)) + sw * (ws * ww)) + se * (ws * we);
}

// ---- counter-based RNG (Philox4x32-10) for the noise stage ----
struct Philox {
    uint32_t c[4], k[2];
    __device__ void round()
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    __device__ void block()
    {
#pragma unroll
        for (int i = 0; i < 10; ++i) round();
    }
};

struct Rng {
    uint32_t key0, key1, ctr0, ctr1, sub;
    uint32_t buf[4];
    int left;
    __device__ Rng(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) : key0(k0), key1(k1), ctr0(c0), ctr1(c1), sub(0), left(0) {}
    __device__ uint32_t next()
    {
        if (left == 0) {
            Philox p;
            p.c[0] = ctr0; p.c[1] = ctr1; p.c[2] = sub++; p.c[3] = 0x5114EBE2u;
            p.k[0] = key0; p.k[1] = key1;
            p.block();
            buf[0] = p.c[0]; buf[1] = p.c[1]; buf[2] = p.c[2]; buf[3] = p.c[3];
            left = 4;
        }
        return buf[--left];
    }
    __device__ float uniform() { return ((float)(next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)
    __device__ float normal()
    {
        const float u1 = uniform(), u2 = uniform();
        return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    // Poisson(lambda): multiplication method below 10, Hoermann's transformed rejection (PTRS) above
    __device__ float poisson(float lambda)
    {
        if (!(lambda > 0.0f)) return 0.0f;
        if (lambda < 10.0f) {
            const float limit = expf(-lambda);
            float prod = uniform();
            int k = 0;
            while (prod > limit && k < 200) { prod *= uniform(); ++k; }
            return (float)k;
        }
        const float slam = sqrtf(lambda), loglam = logf(lambda);
        const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
        const float inv_alpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.0f);
        for (int it = 0; it < 64; ++it) {
            const float U = uniform() - 0.5f, V = uniform();
            const float us = 0.5f - fabsf(U);
            const float k = floorf((2.0f * a / us + b) * U + lambda + 0.43f);
            if (us >= 0.07f && V <= vr) return k;
            if (k < 0.0f || (us < 0.013f && V > us)) continue;
            if (logf(V) + logf(inv_alpha) - logf(a / (us * us) + b) <= -lambda + k * loglam - lgammaf(k + 1.0f)) return k;
        }
        return floorf(lambda + 0.5f);
    }
};

// ---- hue jitter (camera_model.py:165-220), one pixel ----
__device__ __forceinline__ void hue_jitter(float R, float G, float B, float hue_shift, float* out)
{
    // torch.max / torch.min over dim 0: first index among equal values
    float M = R; int Mi = 0;
    if (G > M) { M = G; Mi = 1; }
    if (B > M) { M = B; Mi = 2; }
    float m = R;
    if (G < m) m = G;
    if (B < m) m = B;
    const float C = M - m;
    float Hh;
    if (C == 0.0f) Hh = 0.0f;
    else if (Mi == 0) Hh = (G - B) / C + 0.0f;
    else if (Mi == 1) Hh = (B - R) / C + 2.0f;
    else Hh = (R - G) / C + 4.0f;
    float h = 60.0f * Hh;
    if (h < 0.0f) h += 360.0f;
    h = h + hue_shift * 360.0f;
    if (h < 0.0f) h += 360.0f;
    if (h > 360.0f) h -= 360.0f;
    h /= 60.0f;
    const float X = C * (1.0f - fabsf(fmodf(h, 2.0f) - 1.0f));
    int oc = (int)h;          // .long() truncates; h >= 0 here
    oc = oc < 0 ? 0 : (oc > 5 ? 5 : oc);
    const float cx0[3] = {C, X, 0.0f};
    // order[case] = which of (C, X, 0) goes to R, G, B
    const int o0 = oc == 0 || oc == 5 ? 0 : (oc == 1 || oc == 4 ? 1 : 2);
    const int o1 = oc == 0 || oc == 3 ? 1 : (oc == 1 || oc == 2 ? 0 : 2);
    const int o2 = oc == 0 || oc == 1 ? 2 : (oc == 2 || oc == 5 ? 1 : 0);
    out[0] = cx0[o0] + m; out[1] = cx0[o1] + m; out[2] = cx0[o2] + m;
}

// stage 1: chromatic aberration -> blur -> exposure -> noise -> clamp -> hue jitter.
// Block = 32 x 8 output pixels.  With the blur on, the resampled image is needed on the 36 x 12 halo
// tile around the block: it is evaluated once per position into LDS (432 samples per channel for 256
// outputs instead of 25 per output) and the 25 taps read LDS in the reference order.
__global__ __launch_bounds__(256) void k_camera_stage1(const float* __restrict__ in, float* __restrict__ tmp, int H, int W,
                                                       const slhip_camera_params* __restrict__ params)
{
    __shared__ float tile[3][12][36];
    const int img = blockIdx.z;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 8;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x = bx + lx, y = by + ly;
    const slhip_camera_params& p = params[img];
    const size_t P = (size_t)H * W;
    const float* src = in + (size_t)img * 3 * P;
    if (p.blur_enabled) {
        for (int i = threadIdx.x; i < 3 * 12 * 36; i += 256) {
            const int c = i / (12 * 36), r = (i / 36) % 12, q = i % 36;
            const int xx = bx + q - 2, yy = by + r - 2;
            float v = 0.0f;
            if (xx >= 0 && xx < W && yy >= 0 && yy < H)
                v = chroma_sample(src + c * P, H, W, xx, yy, p.scaling[c], p.translation[2 * c], p.translation[2 * c + 1]);
            tile[c][r][q] = v;
        }
        __syncthreads();
    }
    if (x >= W || y >= H) return;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a;
        if (p.blur_enabled) {
            a = 0.0f;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx, yy = y + dy;
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;   // conv2d zero padding
                    a = fmaf(p.blur_kernel[(dy + 2) * 5 + (dx + 2)], tile[c][ly + dy + 2][lx + dx + 2], a);
                }
        } else {
            a = chroma_sample(src + c * P, H, W, x, y, p.scaling[c], p.translation[2 * c], p.translation[2 * c + 1]);
        }
        // exposure: 1 / (1 + e^dS * (1 / (rgb + 0.0001) - 1)), every step rounded to f32 like torch
        const float t1 = a + 0.0001f;
        const float t2 = 1.0f / t1;
        const float t3 = t2 - 1.0f;
        const float t4 = p.exposure_gain * t3;
        const float t5 = 1.0f + t4;
        v[c] = 1.0f / t5;
    }
    if (p.noise_enabled) {
        Rng rng(p.seed_lo, p.seed_hi, (uint32_t)(y * W + x), (uint32_t)img);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float pois = v[c];
            if (p.noise_a > 0.0f) {
                const float chi = 1.0f / p.noise_a;
                pois = rng.poisson(chi * v[c]) / chi;
            }
            const float g = p.noise_b > 0.0f ? rng.normal() * p.noise_b : 0.0f;
            v[c] = fminf(fmaxf(pois + g, 0.0f), 1.0f);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(v[c], 0.0f), 1.0f);
    float o[3];
    hue_jitter(v[0], v[1], v[2], p.hue_shift, o);
    float* dst = tmp + (size_t)img * 3 * P + (size_t)y * W + x;
    dst[0] = o[0]; dst[P] = o[1]; dst[2 * P] = o[2];
}

// stage 2: 5x5 post blur (zero padding) -> clamp
__global__ __launch_bounds__(256) void k_camera_stage2(const float* __restrict__ tmp, float* __restrict__ out, int H, int W,
                                                       const slhip_camera_params* __restrict__ params)
{
    const int img = blockIdx.z;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const slhip_camera_params& p = params[img];
    const size_t P = (size_t)H * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* plane = tmp + ((size_t)img * 3 + c) * P;
        float a = 0.0f;
        for (int dy = -2; dy <= 2; ++dy)
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx, yy = y + dy;
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                a = fmaf(p.post_kernel[(dy + 2) * 5 + (dx + 2)], plane[yy * W + xx], a);
            }
        out[((size_t)img * 3 + c) * P + (size_t)y * W + x] = fminf(fmaxf(a, 0.0f), 1.0f);
    }
}

}  // namespace

extern "C" int slhip_camera_model(const float* d_in, float* d_out, float* d_tmp, uint32_t n_images, int H, int W,
                                  const slhip_camera_params* d_params, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_in || !d_out || !d_tmp || !d_params) {
        slhip::set_error("slhip_camera_model: null argument");
        return -1;
    }
    if (d_tmp == d_in || d_tmp == d_out) {
        slhip::set_error("slhip_camera_model: d_tmp must not alias the input or the output");
        return -1;
    }
    if (H <= 0 || W <= 0 || (size_t)H * W > 0x7fffffffu) {
        slhip::set_error("slhip_camera_model: bad image size %d x %d", W, H);
        return -1;
    }
    if (n_images == 0) return 0;
    const dim3 grid((W + 31) / 32, (H + 7) / 8, n_images);
    k_camera_stage1<<<grid, 256, 0, stream>>>(d_in, d_tmp, H, W, d_params);
    k_camera_stage2<<<grid, 256, 0, stream>>>(d_tmp, d_out, H, W, d_params);
    SLHIP_LAUNCH_CHECK();
    return 0;
}
