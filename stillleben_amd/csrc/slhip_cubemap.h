// slhip_cubemap.h -- device-side cube-map addressing and filtering shared by the IBL precompute kernels
// (slhip_ibl.hip) and the fragment stage (slhip_render.hip).  Rules: include/slhip.h, slhip_light_map.
#pragma once

#include <hip/hip_runtime.h>

#include "slhip.h"

namespace slcube {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }

// offset (in floats) of level l of an RGBA cube of base size n
__device__ __forceinline__ size_t level_offset(unsigned n, unsigned l)
{
    size_t o = 0;
    for (unsigned k = 0; k < l; ++k) { const size_t m = n >> k; o += 24 * m * m; }
    return o;
}

// OpenGL 4.5 table 8.19: direction -> face, (sc, tc) / |ma| in [-1, 1]
__device__ __forceinline__ int dir_to_face(f3 d, float& sc, float& tc)
{
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face;
    float ma, s, t;
    if (ax >= ay && ax >= az) {
        ma = ax;
        if (d.x >= 0.0f) { face = 0; s = -d.z; t = -d.y; } else { face = 1; s = d.z; t = -d.y; }
    } else if (ay >= az) {
        ma = ay;
        if (d.y >= 0.0f) { face = 2; s = d.x; t = d.z; } else { face = 3; s = d.x; t = -d.z; }
    } else {
        ma = az;
        if (d.z >= 0.0f) { face = 4; s = d.x; t = -d.y; } else { face = 5; s = -d.x; t = -d.y; }
    }
    sc = s / ma; tc = t / ma;
    return face;
}

// face + (sc, tc) in face coordinates (may lie outside [-1, 1]: the extended face plane) -> direction
__device__ __forceinline__ f3 face_to_dir(int face, float sc, float tc)
{
    switch (face) {
    case 0: return F3(1.0f, -tc, -sc);
    case 1: return F3(-1.0f, -tc, sc);
    case 2: return F3(sc, 1.0f, tc);
    case 3: return F3(sc, -1.0f, -tc);
    case 4: return F3(sc, -tc, 1.0f);
    default: return F3(-sc, -tc, -1.0f);
    }
}

// texel (i, j) of `face` at a level of size n; indices one step outside the face are resolved through
// the direction of that texel centre (seamless edges)
__device__ __forceinline__ float4 texel(const float* __restrict__ level, int n, int face, int i, int j)
{
    if (i < 0 || i >= n || j < 0 || j >= n) {
        const float sc = (2.0f * ((float)i + 0.5f)) / (float)n - 1.0f, tc = (2.0f * ((float)j + 0.5f)) / (float)n - 1.0f;
        float s2, t2;
        face = dir_to_face(face_to_dir(face, sc, tc), s2, t2);
        i = min(max((int)floorf((s2 + 1.0f) * 0.5f * (float)n), 0), n - 1);
        j = min(max((int)floorf((t2 + 1.0f) * 0.5f * (float)n), 0), n - 1);
    }
    return reinterpret_cast<const float4*>(level)[((size_t)face * n + j) * n + i];
}

// bilinear fetch of one level
__device__ __forceinline__ float4 sample_level(const float* __restrict__ level, int n, f3 d)
{
    float sc, tc;
    const int face = dir_to_face(d, sc, tc);
    const float u = (sc + 1.0f) * 0.5f * (float)n - 0.5f, v = (tc + 1.0f) * 0.5f * (float)n - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = (int)fu, j0 = (int)fv;
    const float4 c00 = texel(level, n, face, i0, j0), c10 = texel(level, n, face, i0 + 1, j0);
    const float4 c01 = texel(level, n, face, i0, j0 + 1), c11 = texel(level, n, face, i0 + 1, j0 + 1);
    float4 r;
    r.x = fmaf(b, fmaf(a, c11.x - c01.x, c01.x) - fmaf(a, c10.x - c00.x, c00.x), fmaf(a, c10.x - c00.x, c00.x));
    r.y = fmaf(b, fmaf(a, c11.y - c01.y, c01.y) - fmaf(a, c10.y - c00.y, c00.y), fmaf(a, c10.y - c00.y, c00.y));
    r.z = fmaf(b, fmaf(a, c11.z - c01.z, c01.z) - fmaf(a, c10.z - c00.z, c00.z), fmaf(a, c10.z - c00.z, c00.z));
    r.w = fmaf(b, fmaf(a, c11.w - c01.w, c01.w) - fmaf(a, c10.w - c00.w, c00.w), fmaf(a, c10.w - c00.w, c00.w));
    return r;
}

// textureLod: linear blend of the two nearest levels
__device__ __forceinline__ float4 sample_lod(const float* __restrict__ cube, unsigned size, unsigned levels, f3 d, float lod)
{
    lod = fminf(fmaxf(lod, 0.0f), (float)(levels - 1));
    const unsigned l0 = (unsigned)floorf(lod);
    const unsigned l1 = min(l0 + 1u, levels - 1u);
    const float f = lod - (float)l0;
    const float4 a = sample_level(cube + level_offset(size, l0), (int)(size >> l0), d);
    if (f == 0.0f || l1 == l0) return a;
    const float4 b = sample_level(cube + level_offset(size, l1), (int)(size >> l1), d);
    float4 r;
    r.x = fmaf(f, b.x - a.x, a.x); r.y = fmaf(f, b.y - a.y, a.y); r.z = fmaf(f, b.z - a.z, a.z); r.w = fmaf(f, b.w - a.w, a.w);
    return r;
}

}  // namespace slcube
