// slhip_settle.hip -- batched rigid-body settling for gfx950 (MI355X), replacing the PhysX calls
// of Scene::simulateTableTopScene / Scene::simulate (reference src/scene.cpp:612-759, :903-912).
//
// Execution model (DESIGN.md "Settle half"): ONE persistent 64-lane workgroup (one wavefront)
// per scene runs all frames x substeps without returning to the host -- the 400 dependent
// steps of a settle are a latency chain, so throughput comes from running thousands of scenes
// side by side (8 waves/SIMD x 4 SIMDs x 256 CUs), not from splitting one scene.  Inside a step
// the 64 lanes fan out over independent items:
//     bodies      -> force integration, pose integration, sleep bookkeeping
//     body pairs  -> bounding-sphere broadphase, survivors compacted IN ORDER with a
//                    wave ballot + popcount prefix (no atomics, deterministic)
//     hull pairs  -> GJK distance + 4 tilted GJK runs for the contact manifold (hull vertices
//                    are <= 64 float4, read through L1/L2; body state lives in LDS)
//     groups      -> Gauss-Seidel contact solve, one colour at a time (groups of one colour
//                    touch disjoint bodies, so lanes never race on a body's velocity)
// Only + - * / sqrt and explicit fmaf are used and every reduction has a fixed order, so the
// result is bit-identical to oracle/settle_ref.c (the parity contract) for any lane count.
#include "slhip_common.h"

namespace {

static_assert(sizeof(slhip_body) == 240, "slhip_body layout");
static_assert(sizeof(slhip_hull) == 32, "slhip_hull layout");
static_assert(sizeof(slhip_settle_params) == 88, "slhip_settle_params layout");

constexpr int kMaxContactsPerHP = 4;
constexpr int kPlaneSlots = 4;
constexpr int kMaxGroups = SLHIP_MAX_HULL_PAIRS + SLHIP_MAX_BODIES;
constexpr int kMaxContacts = kMaxGroups * kMaxContactsPerHP;
constexpr int kPlaneBase = SLHIP_MAX_HULL_PAIRS * kMaxContactsPerHP;
constexpr float kInf = 3.0e38f;
constexpr float kDepthWeight = 30.0f;

struct v3 { float x, y, z; };
struct quat { float x, y, z, w; };
struct m3 { float m[9]; };

__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ v3 cross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ v3 madd(v3 a, v3 b, float s)
{
    return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z));
}
__device__ __forceinline__ v3 m3_mul(const m3& M, v3 v)
{
    return V(fmaf(M.m[2], v.z, fmaf(M.m[1], v.y, M.m[0] * v.x)), fmaf(M.m[5], v.z, fmaf(M.m[4], v.y, M.m[3] * v.x)),
             fmaf(M.m[8], v.z, fmaf(M.m[7], v.y, M.m[6] * v.x)));
}
__device__ __forceinline__ v3 m3_tmul(const m3& M, v3 v)
{
    return V(fmaf(M.m[6], v.z, fmaf(M.m[3], v.y, M.m[0] * v.x)), fmaf(M.m[7], v.z, fmaf(M.m[4], v.y, M.m[1] * v.x)),
             fmaf(M.m[8], v.z, fmaf(M.m[5], v.y, M.m[2] * v.x)));
}
__device__ __forceinline__ quat quat_normalize(quat q)
{
    const float n = sqrtf(fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x))));
    quat r; r.x = q.x / n; r.y = q.y / n; r.z = q.z / n; r.w = q.w / n;
    return r;
}
__device__ __forceinline__ void quat_to_m3(quat q, m3& R)
{
    const float x = q.x, y = q.y, z = q.z, w = q.w;
    R.m[0] = 1.0f - 2.0f * (y * y + z * z); R.m[1] = 2.0f * (x * y - z * w); R.m[2] = 2.0f * (x * z + y * w);
    R.m[3] = 2.0f * (x * y + z * w); R.m[4] = 1.0f - 2.0f * (x * x + z * z); R.m[5] = 2.0f * (y * z - x * w);
    R.m[6] = 2.0f * (x * z - y * w); R.m[7] = 2.0f * (y * z + x * w); R.m[8] = 1.0f - 2.0f * (x * x + y * y);
}
__device__ __forceinline__ quat m3_to_quat(const m3& R)
{
    const float* m = R.m;
    const float t = m[0] + m[4] + m[8];
    quat q;
    if (t > 0.0f) {
        const float s = sqrtf(t + 1.0f) * 2.0f;
        q.w = 0.25f * s; q.x = (m[7] - m[5]) / s; q.y = (m[2] - m[6]) / s; q.z = (m[3] - m[1]) / s;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        const float s = sqrtf(1.0f + m[0] - m[4] - m[8]) * 2.0f;
        q.w = (m[7] - m[5]) / s; q.x = 0.25f * s; q.y = (m[1] + m[3]) / s; q.z = (m[2] + m[6]) / s;
    } else if (m[4] > m[8]) {
        const float s = sqrtf(1.0f + m[4] - m[0] - m[8]) * 2.0f;
        q.w = (m[2] - m[6]) / s; q.x = (m[1] + m[3]) / s; q.y = 0.25f * s; q.z = (m[5] + m[7]) / s;
    } else {
        const float s = sqrtf(1.0f + m[8] - m[0] - m[4]) * 2.0f;
        q.w = (m[3] - m[1]) / s; q.x = (m[2] + m[6]) / s; q.y = (m[5] + m[7]) / s; q.z = 0.25f * s;
    }
    return quat_normalize(q);
}
__device__ __forceinline__ quat quat_mul(quat a, quat b)
{
    quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

// ---------------------------------------------------------------------------------------------
// working state
// ---------------------------------------------------------------------------------------------
struct WBody {
    v3 x; quat q; m3 R; v3 t; v3 v, w; m3 Iinv_w;
    float inv_mass; int dynamic;
};

struct Contact {
    int a, b;
    v3 ra, rb, n, t1, t2;
    float sep, rest, kn, kt1, kt2, ln, lt1, lt2, vn0, mu_s, mu_d, e;
    int valid;
};

// per-scene scratch in global memory (L2 resident)
struct SceneScratch {
    Contact c[kMaxContacts];
    float hp_sep[SLHIP_MAX_HULL_PAIRS];
#ifdef SLHIP_SETTLE_PROFILE
    unsigned long long cycles[16];
    unsigned long long counts[16];
#endif
};

#ifdef SLHIP_SETTLE_PROFILE
#define PROF_T0() unsigned long long _pt = wall_clock64()
#define PROF(i) do { unsigned long long _n = wall_clock64(); if (threadIdx.x == 0) X.cycles[i] += _n - _pt; _pt = _n; } while (0)
#define PROF_COUNT(i, v) do { if (threadIdx.x == 0) X.counts[i] += (v); } while (0)
#else
#define PROF_T0()
#define PROF(i)
#define PROF_COUNT(i, v)
#endif

struct Shape {
    const float* verts;
    int count;
    m3 R;
    v3 t;
};

__device__ __forceinline__ v3 support(const Shape& s, v3 d)
{
    const v3 dl = m3_tmul(s.R, d);
    int best = 0;
    const float4* vp = reinterpret_cast<const float4*>(s.verts);
    float4 p0 = vp[0];
    float bd = dot(V(p0.x, p0.y, p0.z), dl);
    for (int i = 1; i < s.count; ++i) {
        const float4 p = vp[i];
        const float dd = dot(V(p.x, p.y, p.z), dl);
        if (dd > bd) { bd = dd; best = i; }
    }
    const float4 p = vp[best];
    return add(m3_mul(s.R, V(p.x, p.y, p.z)), s.t);
}

struct SV { v3 w, a, b; };

__device__ __forceinline__ int closest_segment(const SV* s, float* l)
{
    const v3 a = s[0].w, b = s[1].w;
    const v3 ab = sub(b, a);
    float t = dot(neg(a), ab);
    if (t <= 0.0f) { l[0] = 1.0f; l[1] = 0.0f; return 1; }
    const float den = dot(ab, ab);
    if (t >= den) { l[0] = 0.0f; l[1] = 1.0f; return 2; }
    t = t / den;
    l[0] = 1.0f - t; l[1] = t;
    return 3;
}

__device__ int closest_triangle(v3 a, v3 b, v3 c, float* l)
{
    const v3 ab = sub(b, a), ac = sub(c, a), ap = neg(a);
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) { l[0] = 1; l[1] = 0; l[2] = 0; return 1; }
    const v3 bp = neg(b);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) { l[0] = 0; l[1] = 1; l[2] = 0; return 2; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        l[0] = 1.0f - v; l[1] = v; l[2] = 0; return 3;
    }
    const v3 cp = neg(c);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) { l[0] = 0; l[1] = 0; l[2] = 1; return 4; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float w = d2 / (d2 - d6);
        l[0] = 1.0f - w; l[1] = 0; l[2] = w; return 5;
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        l[0] = 0; l[1] = 1.0f - w; l[2] = w; return 6;
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    l[0] = 1.0f - v - w; l[1] = v; l[2] = w;
    return 7;
}

__device__ __forceinline__ v3 comb3(v3 a, v3 b, v3 c, const float* l)
{
    return madd(madd(scale(a, l[0]), b, l[1]), c, l[2]);
}

__device__ __forceinline__ bool outside_plane(v3 a, v3 b, v3 c, v3 d)
{
    const v3 n = cross(sub(b, a), sub(c, a));
    const float sp = dot(neg(a), n);
    const float sd = dot(sub(d, a), n);
    return sp * sd < 0.0f || sd == 0.0f;
}

__device__ int reduce_simplex(SV* s, int n, float* lam, v3* v)
{
    if (n == 1) { lam[0] = 1.0f; *v = s[0].w; return 1; }
    if (n == 2) {
        float l[2];
        const int mask = closest_segment(s, l);
        *v = madd(scale(s[0].w, l[0]), s[1].w, l[1]);
        if (mask == 1) { lam[0] = 1.0f; return 1; }
        if (mask == 2) { s[0] = s[1]; lam[0] = 1.0f; return 1; }
        lam[0] = l[0]; lam[1] = l[1];
        return 2;
    }
    if (n == 3) {
        float l[3];
        const int mask = closest_triangle(s[0].w, s[1].w, s[2].w, l);
        *v = comb3(s[0].w, s[1].w, s[2].w, l);
        int k = 0;
        for (int i = 0; i < 3; ++i)
            if (mask & (1 << i)) { s[k] = s[i]; lam[k] = l[i]; ++k; }
        return k;
    }
    const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    float best = 3.0e38f;
    int best_mask = 0, best_face = -1;
    float best_l[3] = {0, 0, 0};
    v3 best_v = V(0, 0, 0);
    for (int f = 0; f < 4; ++f) {
        const v3 a = s[F[f][0]].w, b = s[F[f][1]].w, c = s[F[f][2]].w, d = s[F[f][3]].w;
        if (!outside_plane(a, b, c, d)) continue;
        float l[3];
        const int mask = closest_triangle(a, b, c, l);
        const v3 q = comb3(a, b, c, l);
        const float dd = dot(q, q);
        if (dd < best) { best = dd; best_mask = mask; best_face = f; best_l[0] = l[0]; best_l[1] = l[1]; best_l[2] = l[2]; best_v = q; }
    }
    if (best_face < 0) return 0;
    const SV t[3] = {s[F[best_face][0]], s[F[best_face][1]], s[F[best_face][2]]};
    int k = 0;
    for (int i = 0; i < 3; ++i)
        if (best_mask & (1 << i)) { s[k] = t[i]; lam[k] = best_l[i]; ++k; }
    *v = best_v;
    return k;
}

constexpr int kGjkMaxIter = 32;

__device__ int gjk_distance(const Shape& A, const Shape& B, v3 init_dir, v3* pa, v3* pb, float* dist)
{
    SV s[4];
    float lam[4] = {1, 0, 0, 0};
    int n = 0;
    v3 v = init_dir;
    if (dot(v, v) < 1e-12f) v = V(1, 0, 0);
    float vv = dot(v, v);
    for (int it = 0; it < kGjkMaxIter; ++it) {
        SV w;
        w.a = support(A, neg(v));
        w.b = support(B, v);
        w.w = sub(w.a, w.b);
        if (n > 0) {
            const float vw = dot(v, w.w);
            if (vv - vw <= 1e-6f * vv) break;
            bool dup = false;
            for (int i = 0; i < n; ++i)
                if (s[i].w.x == w.w.x && s[i].w.y == w.w.y && s[i].w.z == w.w.z) dup = true;
            if (dup) break;
        }
        s[n++] = w;
        v3 nv;
        const int nn = reduce_simplex(s, n, lam, &nv);
        if (nn == 0) return 0;
        const float nvv = dot(nv, nv);
        if (n > 1 && nvv >= vv && it > 0) { n = nn; v = nv; vv = nvv; break; }
        n = nn; v = nv; vv = nvv;
        if (vv < 1e-12f) return 0;
    }
    if (n == 0) return 0;
    v3 a = V(0, 0, 0), b = V(0, 0, 0);
    for (int i = 0; i < n; ++i) { a = madd(a, s[i].a, lam[i]); b = madd(b, s[i].b, lam[i]); }
    *pa = a; *pb = b;
    const float d = sqrtf(vv);
    *dist = d;
    return d > 1e-6f;
}

__device__ __forceinline__ void tangents(v3 n, v3* t1, v3* t2)
{
    v3 a;
    if (fabsf(n.x) > 0.57735f) a = V(n.y, -n.x, 0.0f);
    else a = V(0.0f, n.z, -n.y);
    const float l = sqrtf(dot(a, a));
    *t1 = scale(a, 1.0f / l);
    *t2 = cross(n, *t1);
}

__device__ void overlap_fallback(const Shape& A, const Shape& B, v3 ca, v3 cb, v3* n, float* sep, v3* pa, v3* pb)
{
    v3 axes[7];
    const v3 c = sub(ca, cb);
    const float cl = sqrtf(dot(c, c));
    axes[0] = cl > 1e-6f ? scale(c, 1.0f / cl) : V(0, 0, 1);
    axes[1] = V(1, 0, 0); axes[2] = V(-1, 0, 0); axes[3] = V(0, 1, 0);
    axes[4] = V(0, -1, 0); axes[5] = V(0, 0, 1); axes[6] = V(0, 0, -1);
    float best = -3.0e38f;
    for (int i = 0; i < 7; ++i) {
        const v3 a = support(A, neg(axes[i]));
        const v3 b = support(B, axes[i]);
        const float s = dot(sub(a, b), axes[i]);
        if (s > best) { best = s; *n = axes[i]; *pa = a; *pb = b; }
    }
    *sep = best;
}

__device__ __forceinline__ void make_shape(const WBody& wb, const slhip_hull& h, const float* hull_verts, Shape& s)
{
    s.verts = hull_verts + 4 * (size_t)h.vtx_begin;
    s.count = (int)h.vtx_count;
    s.R = wb.R;
    s.t = wb.t;
}

__device__ int reduce4(int n, const v3* p, const float* sep, v3 nrm, int* keep)
{
    if (n <= 4) { for (int i = 0; i < n; ++i) keep[i] = i; return n; }
    int i0 = 0;
    for (int i = 1; i < n; ++i) if (sep[i] < sep[i0]) i0 = i;
    int i1 = -1; float best = -3.0e38f;
    for (int i = 0; i < n; ++i) {
        if (i == i0) continue;
        const v3 d = sub(p[i], p[i0]);
        const float pen = kDepthWeight * (sep[i] - sep[i0]);
        const float score = sqrtf(dot(d, d)) - pen;
        if (score > best) { best = score; i1 = i; }
    }
    int i2 = -1, i3 = -1; float mx = 0.0f, mn = 0.0f;
    const v3 e = sub(p[i1], p[i0]);
    const float el = sqrtf(dot(e, e));
    for (int i = 0; i < n; ++i) {
        if (i == i0 || i == i1) continue;
        const float a = dot(cross(e, sub(p[i], p[i0])), nrm);
        const float pen = kDepthWeight * (sep[i] - sep[i0]) * el;
        if (a - pen > mx) { mx = a - pen; i2 = i; }
        if (a + pen < mn) { mn = a + pen; i3 = i; }
    }
    int k = 0;
    keep[k++] = i0; keep[k++] = i1;
    if (i2 >= 0) keep[k++] = i2;
    if (i3 >= 0) keep[k++] = i3;
    return k;
}

__device__ __forceinline__ void fill_contact(Contact* c, int a, int b, const WBody& wa, const WBody* wbb, v3 pa, v3 pb,
                                             v3 n, float sep, float rest, float mu_s, float mu_d, float e)
{
    Contact k;
    k.a = a; k.b = b;
    k.ra = sub(pa, wa.x);
    k.rb = wbb ? sub(pb, wbb->x) : V(0, 0, 0);
    k.n = n;
    tangents(n, &k.t1, &k.t2);
    k.sep = sep; k.rest = rest;
    k.kn = k.kt1 = k.kt2 = 0.0f;
    k.ln = k.lt1 = k.lt2 = 0.0f;
    k.vn0 = 0.0f;
    k.mu_s = mu_s; k.mu_d = mu_d; k.e = e;
    k.valid = 1;
    *c = k;
}

__device__ float hull_pair_contacts(const slhip_body* bodies, const WBody* wbs, int ia, int ib, const slhip_hull& ha,
                                    const slhip_hull& hb, const float* hull_verts, const slhip_settle_params& prm,
                                    float margin, Contact* out)
{
    for (int i = 0; i < kMaxContactsPerHP; ++i) out[i].valid = 0;
    const WBody& wa = wbs[ia];
    const WBody& wb = wbs[ib];
    Shape A, B;
    make_shape(wa, ha, hull_verts, A);
    make_shape(wb, hb, hull_verts, B);
    const v3 ca = add(m3_mul(wa.R, V(ha.sphere[0], ha.sphere[1], ha.sphere[2])), wa.t);
    const v3 cb = add(m3_mul(wb.R, V(hb.sphere[0], hb.sphere[1], hb.sphere[2])), wb.t);
    v3 pa, pb, n;
    float dist;
    const float rest = 2.0f * prm.rest_offset;
    const float mu_s = 0.5f * (bodies[ia].mu_s + bodies[ib].mu_s);
    const float mu_d = 0.5f * (bodies[ia].mu_d + bodies[ib].mu_d);
    const float e = 0.5f * (bodies[ia].restitution + bodies[ib].restitution);
    if (!gjk_distance(A, B, sub(ca, cb), &pa, &pb, &dist)) {
        float sep;
        overlap_fallback(A, B, ca, cb, &n, &sep, &pa, &pb);
        if (sep > 0.0f) sep = 0.0f;
        fill_contact(&out[0], ia, ib, wa, &wb, pa, pb, n, sep, rest, mu_s, mu_d, e);
        return sep;
    }
    if (dist > margin) return kInf;
    n = scale(sub(pa, pb), 1.0f / dist);

    v3 cp[5], cq[5];
    float cs[5];
    int nc = 0;
    cp[0] = pa; cq[0] = pb; cs[0] = dist; nc = 1;

    const bool tilt_a = ha.sphere[3] <= hb.sphere[3];
    const float radius = tilt_a ? ha.sphere[3] : hb.sphere[3];
    float ang = 2.0f * prm.contact_offset / radius;
    if (ang > 0.2f) ang = 0.2f;
    const float lift = radius * ang;
    const float sh = 0.5f * ang;
    const float ch = sqrtf(1.0f - sh * sh);
    v3 t1, t2;
    tangents(n, &t1, &t2);
    const WBody& wt = tilt_a ? wa : wb;
    for (int k = 0; k < 4; ++k) {
        const v3 ax = (k == 0) ? t1 : (k == 1) ? t2 : (k == 2) ? neg(t1) : neg(t2);
        quat dq; dq.x = ax.x * sh; dq.y = ax.y * sh; dq.z = ax.z * sh; dq.w = ch;
        const quat q2 = quat_normalize(quat_mul(dq, wt.q));
        Shape T = tilt_a ? A : B;
        quat_to_m3(q2, T.R);
        const v3 cl = tilt_a ? V(ha.sphere[0], ha.sphere[1], ha.sphere[2]) : V(hb.sphere[0], hb.sphere[1], hb.sphere[2]);
        v3 cw = tilt_a ? ca : cb;
        cw = madd(cw, n, tilt_a ? lift : -lift);
        T.t = sub(cw, m3_mul(T.R, cl));
        v3 qa, qb;
        float d2;
        const int ok = tilt_a ? gjk_distance(T, B, sub(ca, cb), &qa, &qb, &d2) : gjk_distance(A, T, sub(ca, cb), &qa, &qb, &d2);
        if (!ok) continue;
        if (tilt_a) {
            const v3 loc = m3_tmul(T.R, sub(qa, T.t));
            qa = add(m3_mul(A.R, loc), A.t);
        } else {
            const v3 loc = m3_tmul(T.R, sub(qb, T.t));
            qb = add(m3_mul(B.R, loc), B.t);
        }
        const float s = dot(sub(qa, qb), n);
        if (s > margin) continue;
        const v3 lat = sub(sub(qa, qb), scale(n, s));
        if (dot(lat, lat) > 4.0f * margin * margin) continue;
        bool dup = false;
        for (int j = 0; j < nc; ++j) {
            const v3 dd = sub(cp[j], qa);
            if (dot(dd, dd) < 2.5e-3f * radius * radius) dup = true;
        }
        if (dup) continue;
        cp[nc] = qa; cq[nc] = qb; cs[nc] = s; ++nc;
    }
    int keep[4];
    const int nk = reduce4(nc, cp, cs, n, keep);
    float mins = kInf;
    for (int i = 0; i < nk; ++i) {
        const int j = keep[i];
        fill_contact(&out[i], ia, ib, wa, &wb, cp[j], cq[j], n, cs[j], rest, mu_s, mu_d, e);
        if (cs[j] < mins) mins = cs[j];
    }
    return mins;
}

__device__ void plane_contacts(const slhip_body* bodies, const WBody* wbs, int ia, const slhip_hull* hulls,
                               const float* hull_verts, const slhip_settle_params& prm, float plane_z, float margin,
                               Contact* out)
{
    for (int i = 0; i < kPlaneSlots; ++i) out[i].valid = 0;
    const slhip_body& b = bodies[ia];
    const WBody& w = wbs[ia];
    v3 cand[64];
    float cs[64];
    int nc = 0;
    for (unsigned h = b.hull_begin; h < b.hull_end; ++h) {
        const slhip_hull hh = hulls[h];
        const v3 c = add(m3_mul(w.R, V(hh.sphere[0], hh.sphere[1], hh.sphere[2])), w.t);
        if (c.z - hh.sphere[3] - plane_z > margin) continue;
        const float4* vs = reinterpret_cast<const float4*>(hull_verts) + hh.vtx_begin;
        for (unsigned i = 0; i < hh.vtx_count; ++i) {
            const float4 q = vs[i];
            const v3 p = add(m3_mul(w.R, V(q.x, q.y, q.z)), w.t);
            const float d = p.z - plane_z;
            if (d > margin) continue;
            if (nc < 64) { cand[nc] = p; cs[nc] = d; ++nc; }
            else {
                int worst = 0;
                for (int k = 1; k < 64; ++k) if (cs[k] > cs[worst]) worst = k;
                if (d < cs[worst]) { cand[worst] = p; cs[worst] = d; }
            }
        }
    }
    if (nc == 0) return;
    int keep[4];
    const v3 n = V(0, 0, 1);
    const int nk = reduce4(nc, cand, cs, n, keep);
    const float mu_s = 0.5f * (b.mu_s + prm.plane_mu_s);
    const float mu_d = 0.5f * (b.mu_d + prm.plane_mu_d);
    const float e = 0.5f * (b.restitution + prm.plane_restitution);
    for (int i = 0; i < nk; ++i) {
        const int j = keep[i];
        const v3 pb = V(cand[j].x, cand[j].y, plane_z);
        fill_contact(&out[i], ia, -1, w, nullptr, cand[j], pb, n, cs[j], prm.rest_offset, mu_s, mu_d, e);
    }
}

// ---------------------------------------------------------------------------------------------
// solver
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ v3 vel_at(const WBody& b, v3 r) { return add(b.v, cross(b.w, r)); }

__device__ __forceinline__ float eff_mass(const WBody& a, const WBody* b, v3 ra, v3 rb, v3 d)
{
    float k = 0.0f;
    if (a.dynamic) {
        const v3 rn = cross(ra, d);
        k += a.inv_mass + dot(cross(m3_mul(a.Iinv_w, rn), ra), d);
    }
    if (b && b->dynamic) {
        const v3 rn = cross(rb, d);
        k += b->inv_mass + dot(cross(m3_mul(b->Iinv_w, rn), rb), d);
    }
    return k > 0.0f ? 1.0f / k : 0.0f;
}

__device__ void prep_contact(Contact* c, const WBody* wbs)
{
    if (!c->valid) return;
    const WBody& a = wbs[c->a];
    const WBody* b = c->b >= 0 ? &wbs[c->b] : nullptr;
    if (!a.dynamic && !(b && b->dynamic)) { c->valid = 0; return; }
    c->kn = eff_mass(a, b, c->ra, c->rb, c->n);
    c->kt1 = eff_mass(a, b, c->ra, c->rb, c->t1);
    c->kt2 = eff_mass(a, b, c->ra, c->rb, c->t2);
    v3 rel = vel_at(a, c->ra);
    if (b) rel = sub(rel, vel_at(*b, c->rb));
    c->vn0 = dot(rel, c->n);
}

__device__ __forceinline__ void apply_impulse(WBody& a, WBody* b, const Contact& c, v3 J)
{
    if (a.dynamic) {
        a.v = madd(a.v, J, a.inv_mass);
        a.w = add(a.w, m3_mul(a.Iinv_w, cross(c.ra, J)));
    }
    if (b && b->dynamic) {
        b->v = madd(b->v, J, -b->inv_mass);
        b->w = sub(b->w, m3_mul(b->Iinv_w, cross(c.rb, J)));
    }
}

__device__ void solve_contact(Contact* cp, WBody* wbs, const slhip_settle_params& prm, bool biased)
{
    if (!cp->valid) return;
    Contact c = *cp;
    WBody& a = wbs[c.a];
    WBody* b = c.b >= 0 ? &wbs[c.b] : nullptr;
    const float inv_dt = 1.0f / prm.dt;
    v3 rel = vel_at(a, c.ra);
    if (b) rel = sub(rel, vel_at(*b, c.rb));
    const float vn = dot(rel, c.n);
    const float err = c.sep - c.rest;
    float target;
    if (err > 0.0f) target = -err * inv_dt;
    else target = biased ? -0.8f * err * inv_dt : 0.0f;
    if (c.vn0 < -prm.bounce_threshold && c.e > 0.0f) {
        const float bounce = -c.e * c.vn0;
        if (bounce > target) target = bounce;
    }
    float dl = (target - vn) * c.kn;
    float ln = c.ln + dl;
    if (ln < 0.0f) ln = 0.0f;
    dl = ln - c.ln;
    c.ln = ln;
    apply_impulse(a, b, c, scale(c.n, dl));
    rel = vel_at(a, c.ra);
    if (b) rel = sub(rel, vel_at(*b, c.rb));
    float l1 = c.lt1 - dot(rel, c.t1) * c.kt1;
    float l2 = c.lt2 - dot(rel, c.t2) * c.kt2;
    const float mag2 = fmaf(l2, l2, l1 * l1);
    const float lim_s = c.mu_s * c.ln;
    if (mag2 > lim_s * lim_s) {
        const float mag = sqrtf(mag2);
        const float k = (c.mu_d * c.ln) / mag;
        l1 *= k; l2 *= k;
    }
    const float d1 = l1 - c.lt1, d2 = l2 - c.lt2;
    c.lt1 = l1; c.lt2 = l2;
    apply_impulse(a, b, c, madd(scale(c.t1, d1), c.t2, d2));
    cp->ln = c.ln; cp->lt1 = c.lt1; cp->lt2 = c.lt2;
}

__device__ void load_body(const slhip_body& b, WBody& w)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) w.R.m[3 * r + c] = b.pose[4 * r + c];
    w.t = V(b.pose[3], b.pose[7], b.pose[11]);
    w.q = m3_to_quat(w.R);
    quat_to_m3(w.q, w.R);
    w.x = add(m3_mul(w.R, V(b.com[0], b.com[1], b.com[2])), w.t);
    w.v = V(b.lin_vel[0], b.lin_vel[1], b.lin_vel[2]);
    w.w = V(b.ang_vel[0], b.ang_vel[1], b.ang_vel[2]);
    w.inv_mass = b.inv_mass;
    w.dynamic = (!(b.flags & (SLHIP_BODY_STATIC | SLHIP_BODY_ASLEEP)) && b.inv_mass > 0.0f) ? 1 : 0;
}

__device__ void update_world_inertia(const slhip_body& b, WBody& w)
{
    m3 L, T;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) L.m[3 * r + c] = b.inv_inertia[4 * r + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T.m[3 * r + c] = fmaf(w.R.m[3 * r + 2], L.m[6 + c], fmaf(w.R.m[3 * r + 1], L.m[3 + c], w.R.m[3 * r] * L.m[c]));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            w.Iinv_w.m[3 * r + c] =
                fmaf(T.m[3 * r + 2], w.R.m[3 * c + 2], fmaf(T.m[3 * r + 1], w.R.m[3 * c + 1], T.m[3 * r] * w.R.m[3 * c]));
}

__device__ void store_body(slhip_body& b, const WBody& w)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) b.pose[4 * r + c] = w.R.m[3 * r + c];
    b.pose[3] = w.t.x; b.pose[7] = w.t.y; b.pose[11] = w.t.z;
    b.pose[12] = 0.0f; b.pose[13] = 0.0f; b.pose[14] = 0.0f; b.pose[15] = 1.0f;
    b.lin_vel[0] = w.v.x; b.lin_vel[1] = w.v.y; b.lin_vel[2] = w.v.z;
    b.ang_vel[0] = w.w.x; b.ang_vel[1] = w.w.y; b.ang_vel[2] = w.w.z;
}

__device__ void redrop(slhip_body* bodies, int nb, int me, const slhip_settle_params& prm)
{
    float max_z = 0.0f;
    for (int o = 0; o < nb; ++o) {
        if (o == me || (bodies[o].flags & SLHIP_BODY_STATIC)) continue;
        const float* P = bodies[o].pose;
        const float* c = bodies[o].bbox_center;
        const float cz = fmaf(P[10], c[2], fmaf(P[9], c[1], P[8] * c[0])) + P[11];
        const float top = cz + c[3];
        if (top > max_z) max_z = top;
    }
    float* P = bodies[me].pose;
    const float* c = bodies[me].bbox_center;
    const float off_z = fmaf(P[10], c[2], fmaf(P[9], c[1], P[8] * c[0])) - c[3];
    P[3] = 0.0f; P[7] = 0.0f; P[11] = max_z - off_z;
    bodies[me].stuck_counter = 0;
    for (int k = 0; k < 4; ++k) { bodies[me].lin_vel[k] = 0.0f; bodies[me].ang_vel[k] = 0.0f; }
    for (int o = 0; o < nb; ++o) {
        bodies[o].flags &= ~SLHIP_BODY_ASLEEP;
        bodies[o].wake_counter = prm.wake_time;
    }
}

// in-order compaction of a per-lane predicate: returns this lane's slot (or -1) and advances
// *count by the number of set lanes (wave64 ballot + popcount prefix)
__device__ __forceinline__ int compact_slot(bool pred, int count)
{
    const unsigned long long m = __ballot(pred);
    const unsigned lane = threadIdx.x & 63;
    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
    return pred ? count + prefix : -1;
}

// ---------------------------------------------------------------------------------------------
// the persistent per-scene kernel
// ---------------------------------------------------------------------------------------------
struct SceneLds {
    WBody wb[SLHIP_MAX_BODIES];
    unsigned long long used[SLHIP_MAX_BODIES];
    int wake[SLHIP_MAX_BODIES];
    short hp_ba[SLHIP_MAX_HULL_PAIRS], hp_bb[SLHIP_MAX_HULL_PAIRS];
    int hp_ha[SLHIP_MAX_HULL_PAIRS], hp_hb[SLHIP_MAX_HULL_PAIRS];
    short g_a[kMaxGroups], g_b[kMaxGroups];
    short g_begin[kMaxGroups], g_end[kMaxGroups];
    signed char g_color[kMaxGroups];
    short bp_i[SLHIP_MAX_BODIES * (SLHIP_MAX_BODIES - 1) / 2], bp_j[SLHIP_MAX_BODIES * (SLHIP_MAX_BODIES - 1) / 2];
    int n_hp, n_groups, n_colors, n_bp;
};

__global__ __launch_bounds__(64) void k_settle(const slhip_settle_scene* __restrict__ scenes, slhip_body* bodies_all,
                                               const slhip_hull* __restrict__ hulls,
                                               const float* __restrict__ hull_verts, slhip_settle_params prm,
                                               SceneScratch* scratch_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SceneLds& S = *reinterpret_cast<SceneLds*>(smem);
    const slhip_settle_scene sc = scenes[blockIdx.x];
    slhip_body* bodies = bodies_all + sc.body_begin;
    const int nb = (int)(sc.body_end - sc.body_begin);
    SceneScratch& X = scratch_all[blockIdx.x];
    const int lane = threadIdx.x;
    const float dt = prm.dt;
    WBody* wb = S.wb;

    for (unsigned frame = 0; frame < prm.frames; ++frame) {
        for (unsigned sub_ = 0; sub_ < prm.substeps; ++sub_) {
            PROF_T0();
            // (a) load, integrate forces
            for (int i = lane; i < nb; i += 64) {
                load_body(bodies[i], wb[i]);
                update_world_inertia(bodies[i], wb[i]);
                bodies[i].separation = kInf;
                if (wb[i].dynamic) {
                    wb[i].v = madd(wb[i].v, V(prm.gravity[0], prm.gravity[1], prm.gravity[2]), dt);
                    float damp = 1.0f - prm.angular_damping * dt;
                    if (damp < 0.0f) damp = 0.0f;
                    wb[i].w = scale(wb[i].w, damp);
                }
                S.wake[i] = 0;
            }
            if (lane == 0) { S.n_hp = 0; S.n_groups = 0; S.n_bp = 0; }
            __syncthreads();

            PROF(0);
            // (b) body-pair broadphase in (i<j) order, ballot-compacted in order
            {
                const int n_pairs = nb * (nb - 1) / 2;
                int count = 0;
                for (int base = 0; base < n_pairs; base += 64) {
                    const int p = base + lane;
                    bool pass = false;
                    int i = 0, j = 0;
                    if (p < n_pairs) {
                        // unrank p -> (i, j), i < j, lexicographic
                        int rem = p;
                        i = 0;
                        int row = nb - 1;
                        while (rem >= row) { rem -= row; ++i; --row; }
                        j = i + 1 + rem;
                        if (wb[i].dynamic || wb[j].dynamic) {
                            const v3 ci = add(m3_mul(wb[i].R, V(bodies[i].bsphere[0], bodies[i].bsphere[1], bodies[i].bsphere[2])), wb[i].t);
                            const v3 cj = add(m3_mul(wb[j].R, V(bodies[j].bsphere[0], bodies[j].bsphere[1], bodies[j].bsphere[2])), wb[j].t);
                            const v3 dv = sub(wb[i].v, wb[j].v);
                            const float spec = sqrtf(dot(dv, dv)) * dt;
                            const float margin = 2.0f * prm.contact_offset + spec;
                            const v3 d = sub(ci, cj);
                            const float rr = bodies[i].bsphere[3] + bodies[j].bsphere[3] + margin;
                            pass = !(dot(d, d) > rr * rr);
                        }
                    }
                    const int slot = compact_slot(pass, count);
                    if (pass) { S.bp_i[slot] = (short)i; S.bp_j[slot] = (short)j; }
                    count += __popcll(__ballot(pass));
                }
                if (lane == 0) S.n_bp = count;
            }
            __syncthreads();

            PROF(1);
            // (c) hull pairs of every surviving body pair (pairs sequential, combos across lanes)
            {
                int n_hp = 0, n_groups = 0;
                const int n_bp = S.n_bp;
                for (int bp = 0; bp < n_bp; ++bp) {
                    const int i = S.bp_i[bp], j = S.bp_j[bp];
                    const unsigned hb0 = bodies[j].hull_begin;
                    const int na = (int)(bodies[i].hull_end - bodies[i].hull_begin);
                    const int nbh = (int)(bodies[j].hull_end - hb0);
                    const v3 dv = sub(wb[i].v, wb[j].v);
                    const float margin = 2.0f * prm.contact_offset + sqrtf(dot(dv, dv)) * dt;
                    const int first = n_hp;
                    const int combos = na * nbh;
                    for (int base = 0; base < combos; base += 64) {
                        const int q = base + lane;
                        bool pass = false;
                        unsigned ha = 0, hb = 0;
                        if (q < combos) {
                            ha = bodies[i].hull_begin + (unsigned)(q / nbh);
                            hb = hb0 + (unsigned)(q % nbh);
                            const slhip_hull A = hulls[ha], B = hulls[hb];
                            const v3 ca = add(m3_mul(wb[i].R, V(A.sphere[0], A.sphere[1], A.sphere[2])), wb[i].t);
                            const v3 cb = add(m3_mul(wb[j].R, V(B.sphere[0], B.sphere[1], B.sphere[2])), wb[j].t);
                            const v3 dd = sub(ca, cb);
                            const float r2 = A.sphere[3] + B.sphere[3] + margin;
                            pass = !(dot(dd, dd) > r2 * r2);
                        }
                        const int slot = compact_slot(pass, n_hp);
                        if (pass && slot < SLHIP_MAX_HULL_PAIRS) {
                            S.hp_ba[slot] = (short)i; S.hp_bb[slot] = (short)j;
                            S.hp_ha[slot] = (int)ha; S.hp_hb[slot] = (int)hb;
                        }
                        n_hp = min(n_hp + (int)__popcll(__ballot(pass)), (int)SLHIP_MAX_HULL_PAIRS);
                    }
                    if (n_hp > first) {
                        if (lane == 0) {
                            S.g_a[n_groups] = (short)i; S.g_b[n_groups] = (short)j;
                            S.g_begin[n_groups] = (short)(first * kMaxContactsPerHP);
                            S.g_end[n_groups] = (short)(n_hp * kMaxContactsPerHP);
                        }
                        ++n_groups;
                    }
                }
                if (lane == 0) { S.n_hp = n_hp; S.n_groups = n_groups; }
            }
            __syncthreads();

            PROF(2);
            // (d) narrowphase: one lane per hull pair
            {
                const int n_hp = S.n_hp;
                for (int k = lane; k < n_hp; k += 64) {
                    const int i = S.hp_ba[k], j = S.hp_bb[k];
                    const v3 dv = sub(wb[i].v, wb[j].v);
                    const float margin = 2.0f * prm.contact_offset + sqrtf(dot(dv, dv)) * dt;
                    X.hp_sep[k] = hull_pair_contacts(bodies, wb, i, j, hulls[S.hp_ha[k]], hulls[S.hp_hb[k]], hull_verts, prm,
                                                     margin, &X.c[k * kMaxContactsPerHP]);
                }
            }
            __syncthreads();
            // min separation per body over its hull pairs (order-independent)
            {
                const int n_hp = S.n_hp;
                for (int i = lane; i < nb; i += 64) {
                    float s = kInf;
                    for (int k = 0; k < n_hp; ++k)
                        if (S.hp_ba[k] == i || S.hp_bb[k] == i) s = fminf(s, X.hp_sep[k]);
                    bodies[i].separation = s;
                }
            }

            PROF(3);
            PROF_COUNT(0, S.n_hp); PROF_COUNT(1, S.n_bp);
            // (e) plane contacts, groups appended in body order
            if (sc.has_plane) {
                int n_groups = S.n_groups;
                for (int base = 0; base < nb; base += 64) {
                    const int i = base + lane;
                    bool pass = false;
                    float margin = 0.0f;
                    if (i < nb && wb[i].dynamic) {
                        const v3 ci = add(m3_mul(wb[i].R, V(bodies[i].bsphere[0], bodies[i].bsphere[1], bodies[i].bsphere[2])), wb[i].t);
                        const float vz = wb[i].v.z < 0.0f ? -wb[i].v.z * dt : 0.0f;
                        margin = prm.contact_offset + vz;
                        pass = !(ci.z - bodies[i].bsphere[3] - sc.plane_z > margin);
                    }
                    const int g = compact_slot(pass, n_groups);
                    if (pass) {
                        plane_contacts(bodies, wb, i, hulls, hull_verts, prm, sc.plane_z, margin, &X.c[kPlaneBase + i * kPlaneSlots]);
                        S.g_a[g] = (short)i; S.g_b[g] = -1;
                        S.g_begin[g] = (short)(kPlaneBase + i * kPlaneSlots);
                        S.g_end[g] = (short)(kPlaneBase + (i + 1) * kPlaneSlots);
                    }
                    n_groups += __popcll(__ballot(pass));
                }
                if (lane == 0) S.n_groups = n_groups;
            }
            __syncthreads();

            PROF(4);
            // wake sleeping bodies touched by a moving body
            {
                const int ng = S.n_groups;
                for (int g = lane; g < ng; g += 64) {
                    const int a = S.g_a[g], b = S.g_b[g];
                    if (b < 0) continue;
                    bool touching = false;
                    for (int i = S.g_begin[g]; i < S.g_end[g]; ++i)
                        if (X.c[i].valid && X.c[i].sep < 2.0f * prm.contact_offset) touching = true;
                    if (!touching) continue;
                    for (int s = 0; s < 2; ++s) {
                        const int me = s ? b : a, other = s ? a : b;
                        if ((bodies[me].flags & SLHIP_BODY_ASLEEP) && wb[other].dynamic) {
                            const float en = 0.5f * dot(wb[other].v, wb[other].v);
                            if (en > prm.sleep_threshold) S.wake[me] = 1;
                        }
                    }
                }
            }
            __syncthreads();
            for (int i = lane; i < nb; i += 64)
                if (S.wake[i]) {
                    bodies[i].flags &= ~SLHIP_BODY_ASLEEP;
                    bodies[i].wake_counter = prm.wake_time;
                }

            PROF(5);
            // (f) prep: every slot of every group
            {
                const int ng = S.n_groups;
                for (int g = 0; g < ng; ++g) {
                    const int b0 = S.g_begin[g], b1 = S.g_end[g];
                    for (int i = b0 + lane; i < b1; i += 64) prep_contact(&X.c[i], wb);
                }
            }
            PROF(6);
            // (g) greedy colouring in group order (serial by definition)
            if (lane == 0) {
                for (int i = 0; i < nb; ++i) S.used[i] = 0ull;
                int ncol = 0;
                const int ng = S.n_groups;
                for (int g = 0; g < ng; ++g) {
                    unsigned long long m = S.used[S.g_a[g]];
                    if (S.g_b[g] >= 0) m |= S.used[S.g_b[g]];
                    int c = 0;
                    while (c < 63 && ((m >> c) & 1ull)) ++c;
                    S.g_color[g] = (signed char)c;
                    S.used[S.g_a[g]] |= 1ull << c;
                    if (S.g_b[g] >= 0) S.used[S.g_b[g]] |= 1ull << c;
                    if (c + 1 > ncol) ncol = c + 1;
                }
                S.n_colors = ncol;
            }
            __syncthreads();

            PROF(7);
            PROF_COUNT(2, S.n_groups); PROF_COUNT(3, S.n_colors);
            // (h) position iterations
            const int ng = S.n_groups, ncol = S.n_colors;
            for (unsigned it = 0; it < prm.pos_iters; ++it)
                for (int col = 0; col < ncol; ++col) {
                    for (int g = lane; g < ng; g += 64) {
                        if (S.g_color[g] != col) continue;
                        for (int i = S.g_begin[g]; i < S.g_end[g]; ++i) solve_contact(&X.c[i], wb, prm, true);
                    }
                    __syncthreads();
                }

            PROF(8);
            // (i) integrate poses
            for (int i = lane; i < nb; i += 64) {
                if (!wb[i].dynamic) continue;
                const float lim = bodies[i].max_lin_vel;
                const float vv = dot(wb[i].v, wb[i].v);
                if (lim > 0.0f && vv > lim * lim) wb[i].v = scale(wb[i].v, lim / sqrtf(vv));
                const float ww = dot(wb[i].w, wb[i].w);
                const float wl = prm.max_angular_velocity;
                if (ww > wl * wl) wb[i].w = scale(wb[i].w, wl / sqrtf(ww));
                wb[i].x = madd(wb[i].x, wb[i].v, dt);
                quat wq; wq.x = wb[i].w.x; wq.y = wb[i].w.y; wq.z = wb[i].w.z; wq.w = 0.0f;
                const quat dq = quat_mul(wq, wb[i].q);
                quat q;
                q.x = fmaf(0.5f * dt, dq.x, wb[i].q.x); q.y = fmaf(0.5f * dt, dq.y, wb[i].q.y);
                q.z = fmaf(0.5f * dt, dq.z, wb[i].q.z); q.w = fmaf(0.5f * dt, dq.w, wb[i].q.w);
                wb[i].q = quat_normalize(q);
            }
            __syncthreads();

            PROF(9);
            // (j) velocity iterations
            for (unsigned it = 0; it < prm.vel_iters; ++it)
                for (int col = 0; col < ncol; ++col) {
                    for (int g = lane; g < ng; g += 64) {
                        if (S.g_color[g] != col) continue;
                        for (int i = S.g_begin[g]; i < S.g_end[g]; ++i) solve_contact(&X.c[i], wb, prm, false);
                    }
                    __syncthreads();
                }

            PROF(10);
            // (k) store + sleep bookkeeping
            for (int i = lane; i < nb; i += 64) {
                if (!wb[i].dynamic) continue;
                quat_to_m3(wb[i].q, wb[i].R);
                wb[i].t = sub(wb[i].x, m3_mul(wb[i].R, V(bodies[i].com[0], bodies[i].com[1], bodies[i].com[2])));
                const float r = bodies[i].bsphere[3];
                const float en = 0.5f * (dot(wb[i].v, wb[i].v) + r * r * dot(wb[i].w, wb[i].w));
                if (en >= prm.sleep_threshold) bodies[i].wake_counter = prm.wake_time;
                else {
                    bodies[i].wake_counter -= dt;
                    if (bodies[i].wake_counter <= 0.0f) {
                        bodies[i].flags |= SLHIP_BODY_ASLEEP;
                        wb[i].v = V(0, 0, 0);
                        wb[i].w = V(0, 0, 0);
                    }
                }
                store_body(bodies[i], wb[i]);
            }
            __threadfence_block();
            __syncthreads();
            PROF(11);
        }
        // redrop heuristic of simulateTableTopScene (scene.cpp:742-755), serial
        if (prm.tabletop) {
            if (lane == 0) {
                for (int i = 0; i < nb; ++i) {
                    if (bodies[i].flags & SLHIP_BODY_STATIC) continue;
                    if (bodies[i].pose[11] < prm.redrop_z) redrop(bodies, nb, i, prm);
                    else if (bodies[i].separation < prm.stuck_separation) {
                        if (++bodies[i].stuck_counter > prm.stuck_frames) redrop(bodies, nb, i, prm);
                    } else if (bodies[i].stuck_counter > 0) bodies[i].stuck_counter--;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

// boolean overlap (scene.cpp:355-385): one lane per body
__global__ __launch_bounds__(64) void k_overlap(const slhip_settle_scene* __restrict__ scenes,
                                                const slhip_body* __restrict__ bodies_all,
                                                const slhip_hull* __restrict__ hulls,
                                                const float* __restrict__ hull_verts, uint8_t* __restrict__ flags)
{
    __shared__ WBody wb[SLHIP_MAX_BODIES];
    const slhip_settle_scene sc = scenes[blockIdx.x];
    const slhip_body* b = bodies_all + sc.body_begin;
    const int nb = (int)(sc.body_end - sc.body_begin);
    for (int i = threadIdx.x; i < nb; i += 64) load_body(b[i], wb[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 64) {
        bool hit = false;
        for (int j = 0; j < nb && !hit; ++j) {
            if (j == i) continue;
            const v3 ci = add(m3_mul(wb[i].R, V(b[i].bsphere[0], b[i].bsphere[1], b[i].bsphere[2])), wb[i].t);
            const v3 cj = add(m3_mul(wb[j].R, V(b[j].bsphere[0], b[j].bsphere[1], b[j].bsphere[2])), wb[j].t);
            const v3 d = sub(ci, cj);
            const float rr = b[i].bsphere[3] + b[j].bsphere[3];
            if (dot(d, d) > rr * rr) continue;
            for (unsigned ha = b[i].hull_begin; ha < b[i].hull_end && !hit; ++ha)
                for (unsigned hb = b[j].hull_begin; hb < b[j].hull_end && !hit; ++hb) {
                    Shape A, B;
                    make_shape(wb[i], hulls[ha], hull_verts, A);
                    make_shape(wb[j], hulls[hb], hull_verts, B);
                    const v3 ca = add(m3_mul(wb[i].R, V(hulls[ha].sphere[0], hulls[ha].sphere[1], hulls[ha].sphere[2])), wb[i].t);
                    const v3 cb = add(m3_mul(wb[j].R, V(hulls[hb].sphere[0], hulls[hb].sphere[1], hulls[hb].sphere[2])), wb[j].t);
                    const v3 dd = sub(ca, cb);
                    const float r2 = hulls[ha].sphere[3] + hulls[hb].sphere[3];
                    if (dot(dd, dd) > r2 * r2) continue;
                    v3 pa, pb;
                    float dist;
                    if (!gjk_distance(A, B, dd, &pa, &pb, &dist)) hit = true;
                }
        }
        if (!hit && sc.has_plane) {
            for (unsigned h = b[i].hull_begin; h < b[i].hull_end && !hit; ++h) {
                const float4* vs = reinterpret_cast<const float4*>(hull_verts) + hulls[h].vtx_begin;
                for (unsigned k = 0; k < hulls[h].vtx_count; ++k) {
                    const float4 q = vs[k];
                    const v3 p = add(m3_mul(wb[i].R, V(q.x, q.y, q.z)), wb[i].t);
                    if (p.z <= sc.plane_z) { hit = true; break; }
                }
            }
        }
        flags[sc.body_begin + i] = hit ? 1 : 0;
    }
}

}  // namespace

extern "C" int slhip_settle_scratch_bytes(uint32_t n_scenes, uint64_t* bytes_out)
{
    *bytes_out = (uint64_t)n_scenes * sizeof(SceneScratch);
    return 0;
}

extern "C" int slhip_settle(const slhip_settle_scene* d_scenes, uint32_t n_scenes, slhip_body* d_bodies,
                            const slhip_hull* d_hulls, const float* d_hull_verts, const slhip_settle_params* params,
                            void* d_scratch, uint64_t scratch_bytes, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_scenes || !d_bodies || !d_hulls || !d_hull_verts || !params || !d_scratch) {
        slhip::set_error("slhip_settle: null argument");
        return -1;
    }
    if (n_scenes == 0) return 0;
    if (scratch_bytes < (uint64_t)n_scenes * sizeof(SceneScratch)) {
        slhip::set_error("slhip_settle: scratch too small (%llu < %llu)", (unsigned long long)scratch_bytes,
                         (unsigned long long)((uint64_t)n_scenes * sizeof(SceneScratch)));
        return -1;
    }
    static bool attr_set = false;
    if (!attr_set) {
        SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_settle), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)sizeof(SceneLds)));
        attr_set = true;
    }
    k_settle<<<n_scenes, 64, sizeof(SceneLds), stream>>>(d_scenes, d_bodies, d_hulls, d_hull_verts, *params,
                                                         reinterpret_cast<SceneScratch*>(d_scratch));
    SLHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int slhip_overlap_any(const slhip_settle_scene* d_scenes, uint32_t n_scenes, const slhip_body* d_bodies,
                                 const slhip_hull* d_hulls, const float* d_hull_verts, uint8_t* d_flags, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_scenes || !d_bodies || !d_hulls || !d_hull_verts || !d_flags) {
        slhip::set_error("slhip_overlap_any: null argument");
        return -1;
    }
    if (n_scenes == 0) return 0;
    k_overlap<<<n_scenes, 64, 0, stream>>>(d_scenes, d_bodies, d_hulls, d_hull_verts, d_flags);
    SLHIP_LAUNCH_CHECK();
    return 0;
}
