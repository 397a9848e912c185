// slhip_diff.hip -- sl.diff hot path for gfx950: the two 3x3 stencils of the reference's diff.cu /
// bridge_diff.cpp (CPU-loop semantics, python/src/bridge_diff.cpp:13-157), the image-space
// gradients (python/stillleben/diff.py:73-127) and a FUSED pose backward
// (diff.py:355-523): one pass over the G-buffer computes, per pixel, the central-difference
// image gradient, the dilated object membership and the 1x3 . 3x2 . 2x3 . 3x6 chain, and
// accumulates [n_obj, 6] through LDS atomics (one global atomic per object and block).
// All of it is HBM-streaming work: ~35 B/pixel read once (rgb 4, coord+depth 16, instance 2,
// grad 12, valid 1), nothing written but the [n_obj,6] result.
#include "slhip_common.h"

namespace {

__global__ __launch_bounds__(256) void k_sobel_valid(const int16_t* __restrict__ inst, const float* __restrict__ depth,
                                                     int dstride, int H, int W, uint8_t* __restrict__ valid)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    {   // blockIdx.y = image of a batch (pose hypotheses): [K, H, W] planes one after the other
        const size_t img = (size_t)blockIdx.y * H * W;
        inst += img; depth += img * dstride; valid += img;
    }
    const int h = p / W, w = p % W;
    uint8_t v = 1;
    if (h >= 1 && h < H - 1 && w >= 1 && w < W - 1) {
        const int16_t cur = inst[p];
        if (cur != 0) {
            const float cd = depth[(size_t)p * dstride];
#pragma unroll
            for (int x = -1; x <= 1; ++x)
#pragma unroll
                for (int y = -1; y <= 1; ++y) {
                    const int q = (h + x) * W + (w + y);
                    const int16_t o = inst[q];
                    if (o != cur && o != 0 && depth[(size_t)q * dstride] < cd) v = 0;
                }
        }
    }
    valid[p] = v;
}

__global__ __launch_bounds__(256) void k_dilate(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ valid,
                                                const float* __restrict__ coords, int cstride, int H, int W,
                                                uint8_t* __restrict__ out_mask, float* __restrict__ out_coords)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int h = p / W, w = p % W;
    uint8_t om = 0;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    if (h >= 1 && h < H - 1 && w >= 1 && w < W - 1) {
        om = mask[p];
        cx = coords[(size_t)p * cstride]; cy = coords[(size_t)p * cstride + 1]; cz = coords[(size_t)p * cstride + 2];
        if (om == 0) {
            bool all_valid = true, all_background = true;
            float nx = 0.0f, ny = 0.0f, nz = 0.0f;
            // the reference's `break` only leaves the inner loop; it never changes the outcome
            // because all_valid is already false then (bridge_diff.cpp:125-142)
#pragma unroll
            for (int x = -1; x <= 1; ++x)
#pragma unroll
                for (int y = -1; y <= 1; ++y) {
                    const int q = (h + x) * W + (w + y);
                    if (mask[q] != 0) {
                        all_background = false;
                        nx = coords[(size_t)q * cstride]; ny = coords[(size_t)q * cstride + 1]; nz = coords[(size_t)q * cstride + 2];
                    }
                    if (valid[q] == 0) all_valid = false;
                }
            if (!all_background && all_valid) { om = 1; cx = nx; cy = ny; cz = nz; }
        }
    }
    out_mask[p] = om;
    out_coords[3 * (size_t)p] = cx; out_coords[3 * (size_t)p + 1] = cy; out_coords[3 * (size_t)p + 2] = cz;
}

__device__ __forceinline__ void pixel_gradients(const uint8_t* __restrict__ rgb, int h, int w, int H, int W, float* gx, float* gy)
{
    const float sx = (float)W / 4.0f, sy = (float)H / 4.0f;
    const size_t p = (size_t)h * W + w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float l = w > 0 ? slhip::unorm8(rgb[4 * (p - 1) + c]) : 0.0f;
        const float r = w < W - 1 ? slhip::unorm8(rgb[4 * (p + 1) + c]) : 0.0f;
        const float u = h > 0 ? slhip::unorm8(rgb[4 * (p - W) + c]) : 0.0f;
        const float d = h < H - 1 ? slhip::unorm8(rgb[4 * (p + W) + c]) : 0.0f;
        gx[c] = -((r - l) * sx);
        gy[c] = -((d - u) * sy);
    }
}

__global__ __launch_bounds__(256) void k_image_gradients(const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ valid,
                                                         int H, int W, float* __restrict__ grad_x, float* __restrict__ grad_y)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    float gx[3], gy[3];
    pixel_gradients(rgb, p / W, p % W, H, W, gx, gy);
    const bool v = valid[p] != 0;
    const size_t N = (size_t)H * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        grad_x[c * N + p] = v ? gx[c] : 0.0f;
        grad_y[c * N + p] = v ? gy[c] : 0.0f;
    }
}

struct Mat4 { float m[16]; };

constexpr int kMaxDiffObjects = 256;

__global__ __launch_bounds__(256) void k_pose_backward(const uint8_t* __restrict__ rgb, const float* __restrict__ coord,
                                                       const int16_t* __restrict__ inst, const uint8_t* __restrict__ valid,
                                                       const float* __restrict__ grad_img, Mat4 P,
                                                       const float* __restrict__ poses, const int* __restrict__ obj_inst,
                                                       int n_obj, int H, int W, double* __restrict__ acc, size_t grad_stride)
{
    {   // blockIdx.y = pose hypothesis of a batch: its G-buffer, its poses, its accumulators; the gradient image is shared
        // (grad_stride 0) or per hypothesis
        const size_t k = blockIdx.y, img = k * (size_t)H * W;
        rgb += 4 * img; coord += 4 * img; inst += img; valid += img;
        grad_img += k * grad_stride; poses += k * 16 * (size_t)n_obj; acc += k * 6 * (size_t)n_obj;
    }
    __shared__ double s_acc[kMaxDiffObjects * 6];
    __shared__ int s_touched[kMaxDiffObjects];
    for (int i = threadIdx.x; i < n_obj * 6; i += 256) s_acc[i] = 0.0;
    for (int i = threadIdx.x; i < n_obj; i += 256) s_touched[i] = 0;
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int h = p / W, w = p % W;
    if (p < H * W && h >= 1 && h < H - 1 && w >= 1 && w < W - 1) {
        int16_t nid[9];
        bool all_valid = true;
#pragma unroll
        for (int x = -1; x <= 1; ++x)
#pragma unroll
            for (int y = -1; y <= 1; ++y) {
                const int q = (h + x) * W + (w + y);
                nid[(x + 1) * 3 + (y + 1)] = inst[q];
                if (valid[q] == 0) all_valid = false;
            }
        const int16_t own = nid[4];
        bool any = own != 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) any = any || (nid[k] != 0 && all_valid);
        if (any) {
            const size_t N = (size_t)H * W;
            float gx[3], gy[3];
            pixel_gradients(rgb, h, w, H, W, gx, gy);
            double s0 = 0.0, s1 = 0.0;
            if (valid[p]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double gi = (double)grad_img[c * N + p];
                    s0 += gi * (double)gx[c];
                    s1 += gi * (double)gy[c];
                }
            }
            for (int o = 0; o < n_obj; ++o) {
                const int16_t id = (int16_t)obj_inst[o];
                int src = -1;
                if (own == id) src = 4;
                else if (all_valid) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) if (nid[k] == id) src = k;  // last neighbour in scan order
                }
                if (src < 0) continue;
                const int q = (h + src / 3 - 1) * W + (w + src % 3 - 1);
                const double X[4] = {(double)coord[4 * (size_t)q], (double)coord[4 * (size_t)q + 1], (double)coord[4 * (size_t)q + 2], 1.0};
                const float* T = poses + 16 * o;
                double y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    y[r] = (double)T[4 * r] * X[0] + (double)T[4 * r + 1] * X[1] + (double)T[4 * r + 2] * X[2] + (double)T[4 * r + 3] * X[3];
                double Py[3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    Py[r] = (double)P.m[4 * r] * y[0] + (double)P.m[4 * r + 1] * y[1] + (double)P.m[4 * r + 2] * y[2] + (double)P.m[4 * r + 3] * y[3];
                double r3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double g0 = (double)P.m[i] * (1.0 / Py[2]) + ((double)P.m[8 + i] * (-1.0 / (Py[2] * Py[2]))) * Py[0];
                    const double g1 = (double)P.m[4 + i] * (1.0 / Py[2]) + ((double)P.m[8 + i] * (-1.0 / (Py[2] * Py[2]))) * Py[1];
                    r3[i] = s0 * g0 + s1 * g1;
                }
                const double GX[6][4] = {{0, -X[2], X[1], 0}, {X[2], 0, -X[0], 0}, {-X[1], X[0], 0, 0},
                                         {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    double g = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double wv = (double)T[4 * i] * GX[k][0] + (double)T[4 * i + 1] * GX[k][1] + (double)T[4 * i + 2] * GX[k][2] + (double)T[4 * i + 3] * GX[k][3];
                        g += r3[i] * wv;
                    }
                    atomicAdd(&s_acc[o * 6 + k], g);
                }
                s_touched[o] = 1;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_obj * 6; i += 256)
        if (s_touched[i / 6]) atomicAdd(&acc[i], s_acc[i]);
}

__global__ void k_acc_to_float(const double* __restrict__ acc, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)acc[i];
}


// D6 (diff.py:215-352): per pixel of an object, -bary_k * dL/dX and -bary_k * dL/dI.  float32 like the
// reference's torch code: x = (P X)_0 / (P X)_2 with P = proj * pose, so dx/dX_j = (P2 P[0,j] - P0 P[2,j]) / P2^2.
__global__ __launch_bounds__(256) void k_vertex_backward(const uint8_t* __restrict__ rgb, const float* __restrict__ coord,
                                                         const int16_t* __restrict__ inst, const uint8_t* __restrict__ valid,
                                                         const float* __restrict__ bary, const float* __restrict__ grad_img, Mat4 P,
                                                         const float* __restrict__ poses, const int* __restrict__ obj_inst, int n_obj,
                                                         int H, int W, float* __restrict__ gv, float* __restrict__ gc)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const size_t N = (size_t)H * W;
    float outv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, outc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int16_t id = inst[p];
    int o = -1;
    for (int k = 0; k < n_obj; ++k)
        if ((int16_t)obj_inst[k] == id) { o = k; break; }
    if (o >= 0) {
        const int h = p / W, w = p % W;
        float gx[3] = {0, 0, 0}, gy[3] = {0, 0, 0};
        if (valid[p]) pixel_gradients(rgb, h, w, H, W, gx, gy);   // zero padding at the image border, zero where !valid (D3)
        // M = proj * pose, rows 0..2 (the reference's [3x4] <= [4x4] @ [4x4], fp32 dot products in k order)
        const float* T = poses + 16 * o;
        float M[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                M[r][c] = ((P.m[4 * r] * T[c] + P.m[4 * r + 1] * T[4 + c]) + P.m[4 * r + 2] * T[8 + c]) + P.m[4 * r + 3] * T[12 + c];
        const float X[4] = {coord[4 * (size_t)p], coord[4 * (size_t)p + 1], coord[4 * (size_t)p + 2], 1.0f};
        float PX[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) PX[r] = ((M[r][0] * X[0] + M[r][1] * X[1]) + M[r][2] * X[2]) + M[r][3] * X[3];
        const float den = PX[2] * PX[2];
        float gl[3], g3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gl[c] = grad_img[c * N + p];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float gxw = (PX[2] * M[0][j] - PX[0] * M[2][j]) / den;
            const float gyw = (PX[2] * M[1][j] - PX[1] * M[2][j]) / den;
            // grad_img_wrt_3D[c][j] = gx[c] * gxw + gy[c] * gyw;  grad_loss_wrt_3D[j] = sum_c gl[c] * that
            float a = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) a += gl[c] * (gx[c] * gxw + gy[c] * gyw);
            g3[j] = a;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float bk = bary[4 * (size_t)p + k];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                outv[3 * k + j] = -(bk * g3[j]);
                outc[3 * k + j] = -(bk * gl[j]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        gv[9 * (size_t)p + k] = outv[k];
        gc[9 * (size_t)p + k] = outc[k];
    }
}

}  // namespace

extern "C" int slhip_diff_sobel_valid(const int16_t* d_inst, const float* d_depth, int depth_stride, int H, int W,
                                      uint8_t* d_valid, void* stream)
{
    if (!d_inst || !d_depth || !d_valid || H <= 0 || W <= 0) { slhip::set_error("slhip_diff_sobel_valid: bad argument"); return -1; }
    k_sobel_valid<<<(H * W + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_inst, d_depth, depth_stride, H, W, d_valid);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_diff_dilate(const uint8_t* d_mask, const uint8_t* d_valid, const float* d_coords, int coord_stride,
                                 int H, int W, uint8_t* d_out_mask, float* d_out_coords, void* stream)
{
    if (!d_mask || !d_valid || !d_coords || !d_out_mask || !d_out_coords) { slhip::set_error("slhip_diff_dilate: null argument"); return -1; }
    k_dilate<<<(H * W + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_mask, d_valid, d_coords, coord_stride, H, W, d_out_mask, d_out_coords);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_diff_image_gradients(const uint8_t* d_rgb, const uint8_t* d_valid, int H, int W, float* d_grad_x,
                                          float* d_grad_y, void* stream)
{
    if (!d_rgb || !d_valid || !d_grad_x || !d_grad_y) { slhip::set_error("slhip_diff_image_gradients: null argument"); return -1; }
    k_image_gradients<<<(H * W + 255) / 256, 256, 0, (hipStream_t)stream>>>(d_rgb, d_valid, H, W, d_grad_x, d_grad_y);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_diff_pose_backward(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                                        const float* d_grad_img, const float* h_proj, const float* d_poses,
                                        const int32_t* d_obj_inst, int n_obj, int H, int W, uint8_t* d_valid,
                                        double* d_acc, float* d_out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_rgb || !d_coord || !d_inst || !d_grad_img || !h_proj || !d_poses || !d_obj_inst || !d_valid || !d_acc || !d_out) {
        slhip::set_error("slhip_diff_pose_backward: null argument");
        return -1;
    }
    if (n_obj <= 0) return 0;
    if (n_obj > kMaxDiffObjects) { slhip::set_error("slhip_diff_pose_backward: at most %d objects", kMaxDiffObjects); return -1; }
    Mat4 P;
    for (int i = 0; i < 16; ++i) P.m[i] = h_proj[i];
    const int blocks = (H * W + 255) / 256;
    k_sobel_valid<<<blocks, 256, 0, stream>>>(d_inst, d_coord + 3, 4, H, W, d_valid);
    SLHIP_CHECK(hipMemsetAsync(d_acc, 0, sizeof(double) * 6 * n_obj, stream));
    k_pose_backward<<<blocks, 256, 0, stream>>>(d_rgb, d_coord, d_inst, d_valid, d_grad_img, P, d_poses, d_obj_inst, n_obj, H, W, d_acc, 0);
    k_acc_to_float<<<(6 * n_obj + 63) / 64, 64, 0, stream>>>(d_acc, d_out, 6 * n_obj);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

// K pose hypotheses of one scene in ONE launch sequence (BASELINE config C5: 64 objects x 32 hypotheses): the G-buffers
// [K,H,W,..] of the hypotheses' renders, their poses [K,n_obj,16], one gradient image for all (grad_stride_floats = 0) or one
// per hypothesis (3 * H * W); three launches whatever K is -- the hypothesis is a grid dimension.  Row k of d_out [K,n_obj,6] is
// what slhip_diff_pose_backward returns for hypothesis k (the same kernels, the same arithmetic).
extern "C" int slhip_diff_pose_backward_batch(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                                              const float* d_grad_img, uint64_t grad_stride_floats, const float* h_proj,
                                              const float* d_poses, const int32_t* d_obj_inst, int n_obj, int n_hyp, int H, int W,
                                              uint8_t* d_valid, double* d_acc, float* d_out, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_rgb || !d_coord || !d_inst || !d_grad_img || !h_proj || !d_poses || !d_obj_inst || !d_valid || !d_acc || !d_out) {
        slhip::set_error("slhip_diff_pose_backward_batch: null argument");
        return -1;
    }
    if (n_obj <= 0 || n_hyp <= 0) return 0;
    if (n_obj > kMaxDiffObjects) { slhip::set_error("slhip_diff_pose_backward_batch: at most %d objects", kMaxDiffObjects); return -1; }
    if (n_hyp > 65535) { slhip::set_error("slhip_diff_pose_backward_batch: at most 65535 hypotheses per call"); return -1; }
    Mat4 P;
    for (int i = 0; i < 16; ++i) P.m[i] = h_proj[i];
    const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)n_hyp);
    k_sobel_valid<<<grid, 256, 0, stream>>>(d_inst, d_coord + 3, 4, H, W, d_valid);
    SLHIP_CHECK(hipMemsetAsync(d_acc, 0, sizeof(double) * 6 * (size_t)n_obj * n_hyp, stream));
    k_pose_backward<<<grid, 256, 0, stream>>>(d_rgb, d_coord, d_inst, d_valid, d_grad_img, P, d_poses, d_obj_inst, n_obj, H, W, d_acc,
                                              (size_t)grad_stride_floats);
    const int total = 6 * n_obj * n_hyp;
    k_acc_to_float<<<(total + 255) / 256, 256, 0, stream>>>(d_acc, d_out, total);
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_diff_vertex_backward(const uint8_t* d_rgb, const float* d_coord, const int16_t* d_inst,
                                          const float* d_bary, const float* d_grad_img, const float* h_proj,
                                          const float* d_poses, const int32_t* d_obj_inst, int n_obj, int H, int W,
                                          uint8_t* d_valid, float* d_grad_vertices, float* d_grad_colors, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_rgb || !d_coord || !d_inst || !d_bary || !d_grad_img || !h_proj || !d_poses || !d_obj_inst || !d_valid ||
        !d_grad_vertices || !d_grad_colors) {
        slhip::set_error("slhip_diff_vertex_backward: null argument");
        return -1;
    }
    if (H <= 0 || W <= 0 || n_obj < 0) {
        slhip::set_error("slhip_diff_vertex_backward: bad sizes");
        return -1;
    }
    Mat4 P;
    for (int i = 0; i < 16; ++i) P.m[i] = h_proj[i];
    const int blocks = (H * W + 255) / 256;
    k_sobel_valid<<<blocks, 256, 0, stream>>>(d_inst, d_coord + 3, 4, H, W, d_valid);
    k_vertex_backward<<<blocks, 256, 0, stream>>>(d_rgb, d_coord, d_inst, d_valid, d_bary, d_grad_img, P, d_poses, d_obj_inst,
                                                  n_obj, H, W, d_grad_vertices, d_grad_colors);
    SLHIP_LAUNCH_CHECK();
    return 0;
}
