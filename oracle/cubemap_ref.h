/* Cube-map addressing and filtering of the oracle (TEST INFRASTRUCTURE; see render_ref.c).  Restates the
 * sampling rules documented at slhip_light_map in include/slhip.h: OpenGL 4.5 table 8.19 face selection,
 * bilinear within a level, texels beyond a face edge resolved through their direction (seamless),
 * explicit-LOD fetches blend the two nearest levels. */
#ifndef SLREF_CUBEMAP_H
#define SLREF_CUBEMAP_H
#include <math.h>
#include <stddef.h>

typedef struct { float x, y, z; } cm3;
typedef struct { float x, y, z, w; } cm4;

static inline size_t cm_level_offset(unsigned n, unsigned l)
{
    size_t o = 0;
    for (unsigned k = 0; k < l; ++k) { size_t m = n >> k; o += 24 * m * m; }
    return o;
}

static inline int cm_dir_to_face(cm3 d, float* sc, float* tc)
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
    *sc = s / ma; *tc = t / ma;
    return face;
}

static inline cm3 cm_face_to_dir(int face, float sc, float tc)
{
    cm3 r;
    switch (face) {
    case 0: r.x = 1.0f; r.y = -tc; r.z = -sc; break;
    case 1: r.x = -1.0f; r.y = -tc; r.z = sc; break;
    case 2: r.x = sc; r.y = 1.0f; r.z = tc; break;
    case 3: r.x = sc; r.y = -1.0f; r.z = -tc; break;
    case 4: r.x = sc; r.y = -tc; r.z = 1.0f; break;
    default: r.x = -sc; r.y = -tc; r.z = -1.0f; break;
    }
    return r;
}

static inline cm4 cm_texel(const float* level, int n, int face, int i, int j)
{
    if (i < 0 || i >= n || j < 0 || j >= n) {
        const float sc = (2.0f * ((float)i + 0.5f)) / (float)n - 1.0f, tc = (2.0f * ((float)j + 0.5f)) / (float)n - 1.0f;
        float s2, t2;
        face = cm_dir_to_face(cm_face_to_dir(face, sc, tc), &s2, &t2);
        i = (int)floorf((s2 + 1.0f) * 0.5f * (float)n);
        j = (int)floorf((t2 + 1.0f) * 0.5f * (float)n);
        if (i < 0) i = 0; if (i > n - 1) i = n - 1;
        if (j < 0) j = 0; if (j > n - 1) j = n - 1;
    }
    const float* p = level + 4 * (((size_t)face * n + j) * n + i);
    cm4 r = {p[0], p[1], p[2], p[3]};
    return r;
}

static inline float cm_bil(float a, float b, float c00, float c10, float c01, float c11)
{
    const float top = fmaf(a, c10 - c00, c00), bot = fmaf(a, c11 - c01, c01);
    return fmaf(b, bot - top, top);
}

static inline cm4 cm_sample_level(const float* level, int n, cm3 d)
{
    float sc, tc;
    const int face = cm_dir_to_face(d, &sc, &tc);
    const float u = (sc + 1.0f) * 0.5f * (float)n - 0.5f, v = (tc + 1.0f) * 0.5f * (float)n - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = (int)fu, j0 = (int)fv;
    const cm4 c00 = cm_texel(level, n, face, i0, j0), c10 = cm_texel(level, n, face, i0 + 1, j0);
    const cm4 c01 = cm_texel(level, n, face, i0, j0 + 1), c11 = cm_texel(level, n, face, i0 + 1, j0 + 1);
    cm4 r;
    r.x = cm_bil(a, b, c00.x, c10.x, c01.x, c11.x); r.y = cm_bil(a, b, c00.y, c10.y, c01.y, c11.y);
    r.z = cm_bil(a, b, c00.z, c10.z, c01.z, c11.z); r.w = cm_bil(a, b, c00.w, c10.w, c01.w, c11.w);
    return r;
}

static inline cm4 cm_sample_lod(const float* cube, unsigned size, unsigned levels, cm3 d, float lod)
{
    lod = fminf(fmaxf(lod, 0.0f), (float)(levels - 1));
    const unsigned l0 = (unsigned)floorf(lod);
    const unsigned l1 = l0 + 1u < levels - 1u ? l0 + 1u : levels - 1u;
    const float f = lod - (float)l0;
    const cm4 a = cm_sample_level(cube + cm_level_offset(size, l0), (int)(size >> l0), d);
    if (f == 0.0f || l1 == l0) return a;
    const cm4 b = cm_sample_level(cube + cm_level_offset(size, l1), (int)(size >> l1), d);
    cm4 r;
    r.x = fmaf(f, b.x - a.x, a.x); r.y = fmaf(f, b.y - a.y, a.y); r.z = fmaf(f, b.z - a.z, a.z); r.w = fmaf(f, b.w - a.w, a.w);
    return r;
}
#endif
