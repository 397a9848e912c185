/*
 * diff_ref.c -- ORACLE (test infrastructure, not product code).
 *
 * CPU restatement of the sl.diff hot path of the reference:
 *   D1 generate_sobel_valid_mask   /root/reference/python/src/bridge_diff.cpp:13-69  (CPU loops)
 *   D2 dilate_object_mask          python/src/bridge_diff.cpp:71-157                 (CPU loops)
 *   D3 compute_image_space_gradients   python/stillleben/diff.py:73-127
 *   D4 backpropagate_gradient_to_poses python/stillleben/diff.py:355-523
 *
 * PARITY STATUS: pinned.  D1/D2 are checked bit-for-bit against the reference's own CPU loops
 * compiled from bridge_diff.cpp where it lies (oracle/ref_build -> oracle/_ref/
 * libstillleben_diff_python.so), D3/D4 against the reference's diff.py imported in the build
 * container; both through the committed vectors tests/golden/diff_golden.npz
 * (generator: oracle/ref_build/gen_diff_golden.py).
 *
 * The CUDA kernels of the reference (python/src/diff.cu) differ from these CPU loops at the
 * 1-pixel image border (clamp vs skip); the CPU semantics are the contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* D1: valid[h][w] = 0 if a 3x3 neighbour belongs to a different non-zero instance and is closer */
void slref_sobel_valid(const int16_t* inst, const float* depth, int H, int W, uint8_t* valid)
{
    for (int i = 0; i < H * W; ++i) valid[i] = 1;
    for (int h = 1; h < H - 1; ++h)
        for (int w = 1; w < W - 1; ++w) {
            int16_t cur = inst[h * W + w];
            if (cur == 0) continue;
            float cd = depth[h * W + w];
            for (int x = -1; x <= 1; ++x)
                for (int y = -1; y <= 1; ++y) {
                    int16_t o = inst[(h + x) * W + (w + y)];
                    if (o != cur && o != 0 && depth[(h + x) * W + (w + y)] < cd) valid[h * W + w] = 0;
                }
        }
}

/* D2: 1-pixel dilation of an object mask where all 3x3 neighbours scanned so far are valid.
   coords: [H,W,3]; outputs zero-initialised; the 1-pixel image border stays zero. */
void slref_dilate(const uint8_t* mask, const uint8_t* valid, const float* coords, int H, int W,
                  uint8_t* out_mask, float* out_coords)
{
    memset(out_mask, 0, (size_t)H * W);
    memset(out_coords, 0, sizeof(float) * 3 * (size_t)H * W);
    for (int h = 1; h < H - 1; ++h)
        for (int w = 1; w < W - 1; ++w) {
            size_t p = (size_t)h * W + w;
            out_mask[p] = mask[p];
            out_coords[3 * p + 0] = coords[3 * p + 0];
            out_coords[3 * p + 1] = coords[3 * p + 1];
            out_coords[3 * p + 2] = coords[3 * p + 2];
            if (mask[p] != 0) continue;
            int all_valid = 1, all_background = 1;
            float cx = 0, cy = 0, cz = 0;
            for (int x = -1; x <= 1; ++x) {
                for (int y = -1; y <= 1; ++y) {
                    size_t q = (size_t)(h + x) * W + (w + y);
                    if (mask[q] != 0) {
                        all_background = 0;
                        cx = coords[3 * q]; cy = coords[3 * q + 1]; cz = coords[3 * q + 2];
                    }
                    if (valid[q] == 0) { all_valid = 0; break; } /* leaves only the y loop (bridge_diff.cpp:136-140) */
                }
            }
            if (all_background || !all_valid) continue;
            out_mask[p] = 1;
            out_coords[3 * p] = cx; out_coords[3 * p + 1] = cy; out_coords[3 * p + 2] = cz;
        }
}

/* D3: central differences scaled to NDC units, negated, zeroed where !valid.
   rgb: u8 [H,W,4]; grad_x, grad_y: f32 [3,H,W] */
void slref_image_gradients(const uint8_t* rgb, const uint8_t* valid, int H, int W, float* grad_x, float* grad_y)
{
    const float sx = (float)W / 4.0f, sy = (float)H / 4.0f; /* [-1,0,1] / (2/W*2) */
    for (int c = 0; c < 3; ++c)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) {
                size_t p = (size_t)h * W + w;
                float l = w > 0 ? (float)rgb[4 * (p - 1) + c] / 255.0f : 0.0f;
                float r = w < W - 1 ? (float)rgb[4 * (p + 1) + c] / 255.0f : 0.0f;
                float u = h > 0 ? (float)rgb[4 * (p - W) + c] / 255.0f : 0.0f;
                float d = h < H - 1 ? (float)rgb[4 * (p + W) + c] / 255.0f : 0.0f;
                float gx = -((r - l) * sx), gy = -((d - u) * sy);
                if (!valid[p]) { gx = 0.0f; gy = 0.0f; }
                grad_x[(size_t)c * H * W + p] = gx;
                grad_y[(size_t)c * H * W + p] = gy;
            }
}

/* D4: gradient of the objective w.r.t. the 6 pose parameters (alpha,beta,gamma,a,b,c) of every
   object.  rgb u8[H,W,4]; coord f32[H,W,4] (object xyz, depth); inst i16[H,W]; grad_img f32[3,H,W];
   P row-major 4x4; poses [n_obj,16]; obj_inst [n_obj]; out f64->f32 [n_obj,6] */
void slref_pose_backward(const uint8_t* rgb, const float* coord, const int16_t* inst, const float* grad_img,
                         const float* P, const float* poses, const int32_t* obj_inst, int n_obj, int H, int W,
                         float* out)
{
    size_t N = (size_t)H * W;
    uint8_t* valid = (uint8_t*)malloc(N);
    uint8_t* mask = (uint8_t*)malloc(N);
    uint8_t* dmask = (uint8_t*)malloc(N);
    float* c3 = (float*)malloc(sizeof(float) * 3 * N);
    float* dc = (float*)malloc(sizeof(float) * 3 * N);
    float* gx = (float*)malloc(sizeof(float) * 3 * N);
    float* gy = (float*)malloc(sizeof(float) * 3 * N);
    float* depth = (float*)malloc(sizeof(float) * N);
    for (size_t p = 0; p < N; ++p) {
        depth[p] = coord[4 * p + 3];
        c3[3 * p] = coord[4 * p]; c3[3 * p + 1] = coord[4 * p + 1]; c3[3 * p + 2] = coord[4 * p + 2];
    }
    slref_sobel_valid(inst, depth, H, W, valid);
    slref_image_gradients(rgb, valid, H, W, gx, gy);
    for (int o = 0; o < n_obj; ++o) {
        const float* T = poses + 16 * o;
        for (size_t p = 0; p < N; ++p) mask[p] = inst[p] == (int16_t)obj_inst[o];
        slref_dilate(mask, valid, c3, H, W, dmask, dc);
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (size_t p = 0; p < N; ++p) {
            if (!dmask[p]) continue;
            double X[4] = {dc[3 * p], dc[3 * p + 1], dc[3 * p + 2], 1.0};
            double y[4];
            for (int r = 0; r < 4; ++r) y[r] = T[4 * r] * X[0] + T[4 * r + 1] * X[1] + T[4 * r + 2] * X[2] + T[4 * r + 3] * X[3];
            double Py[3];
            for (int r = 0; r < 3; ++r) Py[r] = P[4 * r] * y[0] + P[4 * r + 1] * y[1] + P[4 * r + 2] * y[2] + P[4 * r + 3] * y[3];
            /* d(proj_j)/d(world_i) (diff.py:435-443) */
            double gc[2][3];
            for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 3; ++i)
                    gc[j][i] = P[4 * j + i] * (1.0 / Py[2]) + (P[4 * 2 + i] * (-1.0 / (Py[2] * Py[2]))) * Py[j];
            /* s_j = sum_c grad_in[c] * g_xy[c][j] */
            double s[2] = {0, 0};
            for (int c = 0; c < 3; ++c) {
                double gi = grad_img[(size_t)c * N + p];
                s[0] += gi * gx[(size_t)c * N + p];
                s[1] += gi * gy[(size_t)c * N + p];
            }
            double r3[3];
            for (int i = 0; i < 3; ++i) r3[i] = s[0] * gc[0][i] + s[1] * gc[1][i];
            /* d(world)/d(param) = (T0 G_k X)[:3] (diff.py:445-483) */
            double GX[6][4] = {
                {0, -X[2], X[1], 0},   /* alpha: G[1,2]=-1, G[2,1]=1 */
                {X[2], 0, -X[0], 0},   /* beta:  G[0,2]=1,  G[2,0]=-1 */
                {-X[1], X[0], 0, 0},   /* gamma: G[0,1]=-1, G[1,0]=1 */
                {X[3], 0, 0, 0}, {0, X[3], 0, 0}, {0, 0, X[3], 0}};
            for (int k = 0; k < 6; ++k) {
                double g = 0;
                for (int i = 0; i < 3; ++i) {
                    double w = T[4 * i] * GX[k][0] + T[4 * i + 1] * GX[k][1] + T[4 * i + 2] * GX[k][2] + T[4 * i + 3] * GX[k][3];
                    g += r3[i] * w;
                }
                acc[k] += g;
            }
        }
        for (int k = 0; k < 6; ++k) out[6 * o + k] = (float)acc[k];
    }
    free(valid); free(mask); free(dmask); free(c3); free(dc); free(gx); free(gy); free(depth);
}


/* D6: bp_to_vertices_and_colors (diff.py:215-352), dense form (see slhip_diff_vertex_backward in include/slhip.h):
   per pixel of an object -bary_k * dL/dX and -bary_k * dL/dI, float32 like the reference's torch code.
   gv, gc: f32[H,W,3,3]. */
void slref_vertex_backward(const uint8_t* rgb, const float* coord, const int16_t* inst, const float* bary,
                           const float* grad_img, const float* P, const float* poses, const int32_t* obj_inst,
                           int n_obj, int H, int W, float* gv, float* gc)
{
    const size_t N = (size_t)H * W;
    uint8_t* valid = (uint8_t*)malloc(N);
    float* depth = (float*)calloc(N, sizeof(float));
    float* gx = (float*)malloc(3 * N * sizeof(float));
    float* gy = (float*)malloc(3 * N * sizeof(float));
    for (size_t p = 0; p < N; ++p) depth[p] = coord[4 * p + 3];
    slref_sobel_valid(inst, depth, H, W, valid);
    slref_image_gradients(rgb, valid, H, W, gx, gy);
    for (size_t p = 0; p < N; ++p) {
        float* ov = gv + 9 * p;
        float* oc = gc + 9 * p;
        for (int k = 0; k < 9; ++k) { ov[k] = 0.0f; oc[k] = 0.0f; }
        int o = -1;
        for (int k = 0; k < n_obj; ++k)
            if ((int16_t)obj_inst[k] == inst[p]) { o = k; break; }
        if (o < 0) continue;
        const float* T = poses + 16 * o;
        float M[3][4];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                M[r][c] = ((P[4 * r] * T[c] + P[4 * r + 1] * T[4 + c]) + P[4 * r + 2] * T[8 + c]) + P[4 * r + 3] * T[12 + c];
        const float X[4] = {coord[4 * p], coord[4 * p + 1], coord[4 * p + 2], 1.0f};
        float PX[3];
        for (int r = 0; r < 3; ++r) PX[r] = ((M[r][0] * X[0] + M[r][1] * X[1]) + M[r][2] * X[2]) + M[r][3] * X[3];
        const float den = PX[2] * PX[2];
        float gl[3], g3[3];
        for (int c = 0; c < 3; ++c) gl[c] = grad_img[(size_t)c * N + p];
        for (int j = 0; j < 3; ++j) {
            const float gxw = (PX[2] * M[0][j] - PX[0] * M[2][j]) / den;
            const float gyw = (PX[2] * M[1][j] - PX[1] * M[2][j]) / den;
            float a = 0.0f;
            for (int c = 0; c < 3; ++c) a += gl[c] * (gx[(size_t)c * N + p] * gxw + gy[(size_t)c * N + p] * gyw);
            g3[j] = a;
        }
        for (int k = 0; k < 3; ++k) {
            const float bk = bary[4 * p + k];
            for (int j = 0; j < 3; ++j) {
                ov[3 * k + j] = -(bk * g3[j]);
                oc[3 * k + j] = -(bk * gl[j]);
            }
        }
    }
    free(valid); free(depth); free(gx); free(gy);
}
