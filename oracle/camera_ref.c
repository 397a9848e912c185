/* CPU restatement of the reference's camera noise model, deterministic stages only
 * (python/stillleben/camera_model.py:47-130,165-263).  TEST INFRASTRUCTURE: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into oracle/.
 *
 * Pinned against the reference itself: tests/golden/camera_model_golden.npz holds the outputs of the
 * reference's own camera_model.py (imported in the build container by
 * oracle/ref_build/gen_camera_golden.py) for every deterministic stage; tests/test_oracle_camera.py
 * checks this file against them (to ~1e-6: torch's CPU conv2d / grid_sample do not document their
 * summation order).  The HIP path uses the same operation order as this file and is compared bit
 * for bit.  The noise stage is random (torch.poisson / normal_) and is pinned by moments only. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/slhip.h"

static float base_coord(int i, int n)
{
    /* affine_grid, align_corners=False: linspace(-1, 1, n)[i] * (n - 1) / n; torch's linspace fills
       the upper half from the end (camera_model.py:66) */
    if (n <= 1) return 0.0f;
    const float step = 2.0f / (float)(n - 1);
    const float lin = i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
    return lin * (float)(n - 1) / (float)n;
}

static float reflect_clip(float x, int n)
{
    /* grid_sample padding_mode='reflection', align_corners=False: reflect about [-0.5, n-0.5], clip */
    const float lo = -0.5f, span = (float)n;
    float in = fabsf(x - lo);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    float r = (flips & 1) == 0 ? extra + lo : span - extra + lo;
    return fminf((float)(n - 1), fmaxf(r, 0.0f));
}

static float chroma_sample(const float* plane, int H, int W, int x, int y, float s, float tx, float ty)
{
    const float gx = fmaf(s, base_coord(x, W), tx), gy = fmaf(s, base_coord(y, H), ty);
    const float ix = reflect_clip(((gx + 1.0f) * (float)W - 1.0f) / 2.0f, W);
    const float iy = reflect_clip(((gy + 1.0f) * (float)H - 1.0f) / 2.0f, H);
    const float fx = floorf(ix), fy = floorf(iy);
    const float we = ix - fx, ww = 1.0f - we, ws = iy - fy, wn = 1.0f - ws;
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const int bx1 = x1 < W, by1 = y1 < H;
    const float nw = plane[y0 * W + x0];
    const float ne = bx1 ? plane[y0 * W + x1] : 0.0f;
    const float sw = by1 ? plane[y1 * W + x0] : 0.0f;
    const float se = (bx1 && by1) ? plane[y1 * W + x1] : 0.0f;
    return ((nw * (wn * ww) + ne * (wn * we)) + sw * (ws * ww)) + se * (ws * we);
}

static void hue_jitter(float R, float G, float B, float hue_shift, float* out)
{
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
    int oc = (int)h;
    oc = oc < 0 ? 0 : (oc > 5 ? 5 : oc);
    static const int order[6][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}, {1, 2, 0}, {0, 2, 1}};
    const float cx0[3] = {C, X, 0.0f};
    for (int c = 0; c < 3; ++c) out[c] = cx0[order[oc][c]] + m;
}

/* stage selects how far the pipeline runs (for the stage-wise golden vectors):
   0 chromatic aberration, 1 + blur, 2 + exposure, 3 + clamp + hue jitter, 4 full (+ post blur + clamp) */
int slref_camera_model(const float* in, float* out, float* tmp, uint32_t n_images, int H, int W,
                       const slhip_camera_params* params, int stage)
{
    const size_t P = (size_t)H * W;
    for (uint32_t img = 0; img < n_images; ++img) {
        const slhip_camera_params* p = params + img;
        if (p->noise_enabled) return -1; /* random stage: not restated */
        const float* src = in + (size_t)img * 3 * P;
        float* mid = (stage >= 4 ? tmp : out) + (size_t)img * 3 * P;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float v[3];
                for (int c = 0; c < 3; ++c) {
                    const float* plane = src + c * P;
                    const float s = p->scaling[c], tx = p->translation[2 * c], ty = p->translation[2 * c + 1];
                    float a;
                    if (p->blur_enabled && stage >= 1) {
                        a = 0.0f;
                        for (int dy = -2; dy <= 2; ++dy)
                            for (int dx = -2; dx <= 2; ++dx) {
                                const int xx = x + dx, yy = y + dy;
                                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                                a = fmaf(p->blur_kernel[(dy + 2) * 5 + (dx + 2)], chroma_sample(plane, H, W, xx, yy, s, tx, ty), a);
                            }
                    } else {
                        a = chroma_sample(plane, H, W, x, y, s, tx, ty);
                    }
                    if (stage >= 2) {
                        const float t1 = a + 0.0001f;
                        const float t2 = 1.0f / t1;
                        const float t3 = t2 - 1.0f;
                        const float t4 = p->exposure_gain * t3;
                        const float t5 = 1.0f + t4;
                        a = 1.0f / t5;
                    }
                    v[c] = a;
                }
                if (stage >= 3) {
                    for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(v[c], 0.0f), 1.0f);
                    float o[3];
                    hue_jitter(v[0], v[1], v[2], p->hue_shift, o);
                    v[0] = o[0]; v[1] = o[1]; v[2] = o[2];
                }
                for (int c = 0; c < 3; ++c) mid[c * P + (size_t)y * W + x] = v[c];
            }
        if (stage < 4) continue;
        for (int c = 0; c < 3; ++c) {
            const float* plane = mid + c * P;
            float* dst = out + ((size_t)img * 3 + c) * P;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float a = 0.0f;
                    for (int dy = -2; dy <= 2; ++dy)
                        for (int dx = -2; dx <= 2; ++dx) {
                            const int xx = x + dx, yy = y + dy;
                            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                            a = fmaf(p->post_kernel[(dy + 2) * 5 + (dx + 2)], plane[yy * W + xx], a);
                        }
                    dst[(size_t)y * W + x] = fminf(fmaxf(a, 0.0f), 1.0f);
                }
        }
    }
    return 0;
}
