// slhip_settle.hip -- batched rigid-body settling for gfx950 (MI355X), replacing the PhysX calls
// of Scene::simulateTableTopScene / Scene::simulate (reference src/scene.cpp:612-759, :903-912).
//
// Execution model (DESIGN.md "Settle half"): the 400 dependent steps of a settle are a latency chain, so throughput comes from
// running thousands of scenes side by side.  Every step is a short sequence of kernels over the WHOLE batch (the lockstep
// pipeline, slhip_settle_wide.inc); this file holds the device functions of the step -- the arithmetic of oracle/settle_ref.c:
//     bodies      -> force integration, pose integration, sleep bookkeeping
//     body pairs  -> bounding-sphere broadphase, survivors compacted IN ORDER with a
//                    wave ballot + popcount prefix (no atomics, deterministic)
//     hull pairs  -> GJK distance (portal refinement when the hulls overlap); a contact pair that is new or whose manifold lost a
//                    point gets its face manifold in one step (the two support features clipped against each other), the others
//                    keep the manifold the previous step filed, its points refreshed
//     groups      -> warm-started Gauss-Seidel contact solve, one colour at a time (groups of one colour
//                    touch disjoint bodies, so lanes never race on a body's velocity)
// Only + - * / sqrt and explicit fmaf are used and every reduction has a fixed order, so the
// result is bit-identical to oracle/settle_ref.c (the parity contract) for any lane count.
#include "slhip_common.h"
#include <vector>
#include <cstdlib>

namespace {

static_assert(sizeof(slhip_body) == 304, "slhip_body layout");
static_assert(sizeof(slhip_hull) == 64, "slhip_hull layout");
static_assert(sizeof(slhip_settle_params) == 128, "slhip_settle_params layout");

constexpr int kMaxContactsPerHP = 4;
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
    float mu_s, mu_d;     // friction of the body (combined per group in the solver)
};

// The part of a working body the solver's sweeps touch -- the TAIL of WBody, word for word: what a solver wave keeps in LDS per body
// (76 bytes instead of 152: the pose part stays in the scratch, read where the step integrates and stores; an odd number of words per
// body, so lanes walking different bodies hit different banks).
struct SBody {
    v3 v, w; m3 Iinv_w;
    float inv_mass; int dynamic;
    float mu_s, mu_d;
};
static_assert(sizeof(WBody) == 152 && sizeof(SBody) == 76, "working body layouts");
constexpr int kSBodyWords = (int)sizeof(SBody) / 4, kWBodyWords = (int)sizeof(WBody) / 4;

// accumulated drive impulses of ManipulationSim bodies (global scratch: only driven bodies touch it)
struct DriveAcc { float dl[3], da[3]; };

// solver contact, compacted in LDS (plane contacts first, then hull-pair contacts in pair order).
// 72 bytes: the bodies and the friction coefficients come from the contact's group, the tangent
// basis is a pure function of n and is recomputed in the solver, the restitution target is folded
// into `bounce` (-inf = none).  Between fill_contact and prep_contact the fields kt1 / kt2 / bounce
// carry the body indices (as integer bits) and the restitution; ln is the warm-start impulse from the
// moment the contact is filled (oracle: WARM_START x what the persistent point ended the last step with).
struct Contact {
    v3 ra, rb, n;
    float err;            // sep - rest; after prep_contact: the normal row's target velocity in the biased sweeps
    float kn, kt1, kt2, ln, lt1, lt2;
    float bounce;         // restitution; after prep_contact: the normal row's target velocity in the unbiased sweeps
    float til;            // 1 / |a| of the tangent construction (prep_contact): spares the solver a sqrt and a division per row
};
static_assert(sizeof(Contact) == 72, "Contact layout");

// raw narrowphase result of one hull pair / one body-vs-plane test (registers)
struct RawContacts {
    v3 pa[4], pb[4];
    float sep[4];
    float w[4];        // impulse carried over from the previous step (persistent points), 0 for new ones
    v3 n;
    int count;
};

#ifdef SLHIP_SETTLE_PROFILE
struct ProfScratch { unsigned long long status; unsigned long long cycles[16]; unsigned long long counts[16]; };
#define PROF_T0() unsigned long long _pt = wall_clock64()
#define PROF(i) do { unsigned long long _n = wall_clock64(); if (threadIdx.x == 0) X.cycles[i] += _n - _pt; _pt = _n; } while (0)
#define PROF_COUNT(i, v) do { if (threadIdx.x == 0) X.counts[i] += (v); } while (0)
#else
struct ProfScratch { unsigned long long status; };   // per scene: 0 = stepped, else SLHIP_SETTLE_REFUSED_*
#define PROF_T0()
#define PROF(i)
#define PROF_COUNT(i, v)
#endif

// hull vertices: either in the scene's LDS copy (index into hv) or in the global pool
struct Shape {
    const float4* g;   // global vertices (valid when lds < 0)
    int lds;           // first vertex in the LDS copy, or -1
    int count;
    m3 R;
    v3 t;
};

// argmax_i dot(v_i, d) with first-maximum tie break.  Vertices are fetched eight at a time so
// that the (LDS or L1) latency of the batch overlaps; the compare chain stays in index order.
struct f3 { float x, y, z; };   // 12-byte LDS vertex

__device__ __forceinline__ v3 shape_vertex(const Shape& s, const f3* __restrict__ hv, int i)
{
    if (s.lds >= 0) {
        const f3 q = hv[s.lds + i];
        return add(m3_mul(s.R, V(q.x, q.y, q.z)), s.t);
    }
    const float4 q = s.g[i];
    return add(m3_mul(s.R, V(q.x, q.y, q.z)), s.t);
}

__device__ __forceinline__ v3 support(const Shape& s, const f3* __restrict__ hv, v3 d, int& index)
{
    const v3 dl = m3_tmul(s.R, d);
    int best = 0;
    float bd = -3.0e38f;
    const int n = s.count;
    // Only (value, index) are tracked; the winner is fetched again at the end.  Slots past the last
    // vertex repeat vertex n-1: their dot product equals one already seen, and the strict compare
    // never lets a repeat win, so no bounds test is needed in the chain.
    if (s.lds >= 0) {
        const f3* v = hv + s.lds;
        for (int base = 0; base < n; base += 8) {
            f3 p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = v[min(base + j, n - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dd = dot(V(p[j].x, p[j].y, p[j].z), dl);
                if (dd > bd) { bd = dd; best = base + j; }
            }
        }
        const f3 q = v[best];
        index = best;
        return add(m3_mul(s.R, V(q.x, q.y, q.z)), s.t);
    }
    // the winner's slot within a batch is tracked as an inline constant (one v_cndmask per vertex) and the
    // batch base once per batch; unsigned indices keep the address arithmetic in 32 bits until the final add
    const unsigned last = (unsigned)(n - 1);
    int best_base = 0, best_j = 0;
    for (int base = 0; base < n; base += 8) {
        float4 p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = s.g[min((unsigned)(base + j), last)];
        const float before = bd;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dd = dot(V(p[j].x, p[j].y, p[j].z), dl);
            const bool win = dd > bd;
            bd = win ? dd : bd;
            best_j = win ? j : best_j;
        }
        if (bd != before) best_base = base;   // a win strictly raises bd; without one best_j is untouched too
    }
    best = best_base + best_j;
    const float4 q = s.g[best];
    index = best;
    return add(m3_mul(s.R, V(q.x, q.y, q.z)), s.t);
}

__device__ __forceinline__ v3 support(const Shape& s, const f3* __restrict__ hv, v3 d)
{
    int unused;
    return support(s, hv, d, unused);
}

// simplex vertex: w = a - b; idx = vertex of A | vertex of B << 16
struct SV { v3 w, a, b; int idx; };

// vertices of a converged simplex (oracle gjk_seed): what the pair cache hands the next step.s run
struct GjkSeed { int n; int i0, i1, i2; };
static_assert(sizeof(GjkSeed) == 16, "GjkSeed is stored as one int4");

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

// The simplex lives in four named registers sets (no runtime-indexed arrays: those would be
// placed in scratch memory, cdna_hip_programming.md rule 20).  Same arithmetic as the oracle's
// array version.
struct Simplex {
    SV p0, p1, p2, p3;
    float l0, l1, l2, l3;
    int n;
};

__device__ __forceinline__ SV sel3(const SV& a, const SV& b, const SV& c, int i) { return i == 0 ? a : (i == 1 ? b : c); }
__device__ __forceinline__ float self3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

// keeps the vertices of (a,b,c) whose bit is set in mask, in order, with their weights
__device__ __forceinline__ void keep_masked(Simplex& S, const SV& a, const SV& b, const SV& c, float la, float lb, float lc,
                                            int mask)
{
    // index of the k-th set bit among bits 0..2
    const int b0 = mask & 1, b1 = (mask >> 1) & 1, b2 = (mask >> 2) & 1;
    const int first = b0 ? 0 : (b1 ? 1 : 2);
    const int second = (b0 && b1) ? 1 : 2;  // valid when popcount >= 2
    const int cnt = b0 + b1 + b2;
    S.p0 = sel3(a, b, c, first); S.l0 = self3(la, lb, lc, first);
    if (cnt >= 2) { S.p1 = sel3(a, b, c, second); S.l1 = self3(la, lb, lc, second); }
    if (cnt >= 3) { S.p2 = c; S.l2 = lc; }
    S.n = cnt;
}

// returns the new size (0 = origin inside a tetrahedron) and the closest point in *v
__device__ int reduce_simplex(Simplex& S, v3* v)
{
    if (S.n == 1) { S.l0 = 1.0f; *v = S.p0.w; return 1; }
    if (S.n == 2) {
        float l[2];
        SV two[2] = {S.p0, S.p1};
        const int mask = closest_segment(two, l);
        *v = madd(scale(S.p0.w, l[0]), S.p1.w, l[1]);
        if (mask == 1) { S.l0 = 1.0f; S.n = 1; return 1; }
        if (mask == 2) { S.p0 = S.p1; S.l0 = 1.0f; S.n = 1; return 1; }
        S.l0 = l[0]; S.l1 = l[1];
        return 2;
    }
    if (S.n == 3) {
        float l[3];
        const int mask = closest_triangle(S.p0.w, S.p1.w, S.p2.w, l);
        *v = comb3(S.p0.w, S.p1.w, S.p2.w, l);
        const SV a = S.p0, b = S.p1, c = S.p2;
        keep_masked(S, a, b, c, l[0], l[1], l[2], mask);
        return S.n;
    }
    // tetrahedron: faces (0,1,2|3) (0,2,3|1) (0,3,1|2) (1,3,2|0), first-best wins
    float best = 3.0e38f;
    int best_mask = 0, best_face = -1;
    float bl0 = 0, bl1 = 0, bl2 = 0;
    v3 best_v = V(0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const SV& A = (f == 3) ? S.p1 : S.p0;
        const SV& B = (f == 0) ? S.p1 : (f == 1 ? S.p2 : S.p3);
        const SV& C = (f == 0) ? S.p2 : (f == 1 ? S.p3 : (f == 2 ? S.p1 : S.p2));
        const SV& D = (f == 0) ? S.p3 : (f == 1 ? S.p1 : (f == 2 ? S.p2 : S.p0));
        if (!outside_plane(A.w, B.w, C.w, D.w)) continue;
        float l[3];
        const int mask = closest_triangle(A.w, B.w, C.w, l);
        const v3 q = comb3(A.w, B.w, C.w, l);
        const float dd = dot(q, q);
        if (dd < best) { best = dd; best_mask = mask; best_face = f; bl0 = l[0]; bl1 = l[1]; bl2 = l[2]; best_v = q; }
    }
    if (best_face < 0) return 0;
    const SV q0 = S.p0, q1 = S.p1, q2 = S.p2, q3 = S.p3;
    const SV A = (best_face == 3) ? q1 : q0;
    const SV B = (best_face == 0) ? q1 : (best_face == 1 ? q2 : q3);
    const SV C = (best_face == 0) ? q2 : (best_face == 1 ? q3 : (best_face == 2 ? q1 : q2));
    keep_masked(S, A, B, C, bl0, bl1, bl2, best_mask);
    *v = best_v;
    return S.n;
}

constexpr int kGjkMaxIter = 32;
constexpr unsigned kPersistentMaxScenes = 2048u;   // slhip_settle: batches up to this many scenes take k_w_persistent (measured: DESIGN.md section 4)

__device__ __forceinline__ bool same_w(const SV& a, const SV& b)
{
    return a.w.x == b.w.x && a.w.y == b.w.y && a.w.z == b.w.z;
}

// returns 1 (separated, witnesses valid), 0 (touching / overlapping), 2 (farther than margin)
__device__ __forceinline__ SV seed_vertex(const Shape& A, const Shape& B, const f3* __restrict__ hv, int idx)
{
    SV p;
    p.idx = idx;
    p.a = shape_vertex(A, hv, idx & 0xffff);
    p.b = shape_vertex(B, hv, (int)((unsigned)idx >> 16));
    p.w = sub(p.a, p.b);
    return p;
}

// `exhausted` (optional): set when the loop ran out of iterations instead of ending by one of its own criteria -- with the
// full budget that is an answer like any other, with a smaller one (first pass of the lockstep narrowphase) it means "not done"
__device__ __forceinline__ int gjk_distance(const Shape& A, const Shape& B, const f3* __restrict__ hv, v3 init_dir, float margin,
                                            v3* pa, v3* pb, float* dist, const GjkSeed seed_in, GjkSeed* seed_out, int max_iter,
                                            bool* exhausted = nullptr)
{
    bool ran_out = true;
    Simplex S;
    S.n = 0;
    S.l0 = 1.0f; S.l1 = S.l2 = S.l3 = 0.0f;
    v3 v = init_dir;
    if (dot(v, v) < 1e-12f) v = V(1, 0, 0);
    float vv = dot(v, v);
    const float m2 = margin * margin;
    if (seed_in.n > 0) {
        S.p0 = seed_vertex(A, B, hv, seed_in.i0);
        if (seed_in.n > 1) S.p1 = seed_vertex(A, B, hv, seed_in.i1);
        if (seed_in.n > 2) S.p2 = seed_vertex(A, B, hv, seed_in.i2);
        S.n = seed_in.n;
        reduce_simplex(S, &v);      // at most a triangle: never 0
        vv = dot(v, v);
        if (vv < 1e-12f) return 0;
    }
    for (int it = 0; it < max_iter; ++it) {
        SV w;
        int ia, ib;
        w.a = support(A, hv, neg(v), ia);
        w.b = support(B, hv, v, ib);
        w.idx = ia | (ib << 16);
        w.w = sub(w.a, w.b);
        const float vw = dot(v, w.w);
        if (vw > 0.0f && vw * vw > m2 * vv) {
            // separated beyond the margin: hand the simplex back too (the pair cache restarts from it)
            if (seed_out) {
                if (S.n == 0) { seed_out->n = 1; seed_out->i0 = w.idx; seed_out->i1 = 0; seed_out->i2 = 0; }
                else { seed_out->n = S.n; seed_out->i0 = S.p0.idx; seed_out->i1 = S.p1.idx; seed_out->i2 = S.p2.idx; }
            }
            return 2;
        }
        const int n = S.n;
        if (n > 0) {
            if (vv - vw <= 1e-6f * vv) { ran_out = false; break; }
            bool dup = same_w(S.p0, w);
            if (n > 1) dup = dup || same_w(S.p1, w);
            if (n > 2) dup = dup || same_w(S.p2, w);
            if (dup) { ran_out = false; break; }
        }
        if (n == 0) S.p0 = w; else if (n == 1) S.p1 = w; else if (n == 2) S.p2 = w; else S.p3 = w;
        S.n = n + 1;
        v3 nv;
        const int nn = reduce_simplex(S, &nv);
        if (nn == 0) return 0;
        const float nvv = dot(nv, nv);
        if (n + 1 > 1 && nvv >= vv && it > 0) { v = nv; vv = nvv; ran_out = false; break; }
        v = nv; vv = nvv;
        if (vv < 1e-12f) return 0;
    }
    if (exhausted) *exhausted = ran_out;
    if (S.n == 0) return 0;
    v3 a = V(0, 0, 0), b = V(0, 0, 0);
    a = madd(a, S.p0.a, S.l0); b = madd(b, S.p0.b, S.l0);
    if (S.n > 1) { a = madd(a, S.p1.a, S.l1); b = madd(b, S.p1.b, S.l1); }
    if (S.n > 2) { a = madd(a, S.p2.a, S.l2); b = madd(b, S.p2.b, S.l2); }
    *pa = a; *pb = b;
    if (seed_out) { seed_out->n = S.n; seed_out->i0 = S.p0.idx; seed_out->i1 = S.p1.idx; seed_out->i2 = S.p2.idx; }
    const float d = sqrtf(vv);
    *dist = d;
    return d > 1e-6f ? 1 : 0;
}

__device__ __forceinline__ int gjk_distance(const Shape& A, const Shape& B, const f3* __restrict__ hv, v3 init_dir, float margin,
                                            v3* pa, v3* pb, float* dist)
{
    GjkSeed none;
    none.n = 0; none.i0 = none.i1 = none.i2 = 0;
    return gjk_distance(A, B, hv, init_dir, margin, pa, pb, dist, none, nullptr, kGjkMaxIter);
}

__device__ __forceinline__ v3 tangent_axis(v3 n)
{
    if (fabsf(n.x) > 0.57735f) return V(n.y, -n.x, 0.0f);
    return V(0.0f, n.z, -n.y);
}
__device__ __forceinline__ float tangent_inv_len(v3 n)
{
    const v3 a = tangent_axis(n);
    return 1.0f / sqrtf(dot(a, a));
}
// same values as tangents() with the inverse length supplied
__device__ __forceinline__ void tangents_cached(v3 n, float inv_len, v3* t1, v3* t2)
{
    *t1 = scale(tangent_axis(n), inv_len);
    *t2 = cross(n, *t1);
}
__device__ __forceinline__ void tangents(v3 n, v3* t1, v3* t2) { tangents_cached(n, tangent_inv_len(n), t1, t2); }

__device__ void overlap_fallback(const Shape& A, const Shape& B, const f3* __restrict__ hv, v3 ca, v3 cb, v3* n,
                                 float* sep, v3* pa, v3* pb)
{
    v3 axes[7];
    const v3 c = sub(ca, cb);
    const float cl = sqrtf(dot(c, c));
    axes[0] = cl > 1e-6f ? scale(c, 1.0f / cl) : V(0, 0, 1);
    axes[1] = V(1, 0, 0); axes[2] = V(-1, 0, 0); axes[3] = V(0, 1, 0);
    axes[4] = V(0, -1, 0); axes[5] = V(0, 0, 1); axes[6] = V(0, 0, -1);
    float best = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const v3 a = support(A, hv, neg(axes[i]));
        const v3 b = support(B, hv, axes[i]);
        const float s = dot(sub(a, b), axes[i]);
        if (s > best) { best = s; *n = axes[i]; *pa = a; *pb = b; }
    }
    *sep = best;
}

// ---- penetration of two OVERLAPPING hulls: Minkowski portal refinement (oracle mpr_penetration, same arithmetic) ----
// M = A - B contains the origin; from the interior point v0 (difference of the hulls' sphere centres) the ray through the origin
// leaves M through the portal (v1 v2 v3), refined by support points along its normal.  The portal point closest to the origin,
// pt = pa - pb, is the contact: n = -pt / |pt| (from B to A), separation -|pt|.  The state is four simplex vertices in registers.
constexpr float kMprTol = 1.0e-4f;
constexpr int kMprMaxDiscover = 16, kMprMaxRefine = 24;
__device__ __forceinline__ SV mpr_support(const Shape& A, const Shape& B, const f3* __restrict__ hv, v3 d)
{
    SV w;
    int ia, ib;
    w.a = support(A, hv, d, ia);
    w.b = support(B, hv, neg(d), ib);
    w.idx = ia | (ib << 16);
    w.w = sub(w.a, w.b);
    return w;
}
__device__ __forceinline__ v3 normalized(v3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }

__device__ bool mpr_penetration(const Shape& A, const Shape& B, const f3* __restrict__ hv, v3 ca, v3 cb, v3* n, float* sep,
                                v3* pa, v3* pb)
{
    v3 v0 = sub(ca, cb);
    if (dot(v0, v0) < 1.0e-12f) v0 = V(1.0e-5f, 0.0f, 0.0f);
    v3 dir = normalized(neg(v0));
    SV v1 = mpr_support(A, B, hv, dir);
    if (!(dot(v1.w, dir) > 0.0f)) return false;
    dir = cross(v0, v1.w);
    if (dot(dir, dir) < 1.0e-20f) {
        const float l = sqrtf(dot(v1.w, v1.w));
        if (l < 1.0e-9f) return false;
        *n = scale(v1.w, -1.0f / l); *sep = -l; *pa = v1.a; *pb = v1.b;
        return true;
    }
    dir = normalized(dir);
    SV v2 = mpr_support(A, B, hv, dir);
    if (!(dot(v2.w, dir) > 0.0f)) return false;
    dir = cross(sub(v1.w, v0), sub(v2.w, v0));
    if (dot(dir, dir) < 1.0e-24f) return false;
    dir = normalized(dir);
    if (dot(dir, v0) > 0.0f) { const SV t = v1; v1 = v2; v2 = t; dir = neg(dir); }
    SV v3_ = v2;
    bool found = false;
    for (int it = 0; it < kMprMaxDiscover; ++it) {
        v3_ = mpr_support(A, B, hv, dir);
        if (!(dot(v3_.w, dir) > 0.0f)) return false;
        bool cont = false;
        if (dot(cross(v1.w, v3_.w), v0) < 0.0f) { v2 = v3_; cont = true; }
        else if (dot(cross(v3_.w, v2.w), v0) < 0.0f) { v1 = v3_; cont = true; }
        if (!cont) { found = true; break; }
        dir = cross(sub(v1.w, v0), sub(v2.w, v0));
        if (dot(dir, dir) < 1.0e-24f) return false;
        dir = normalized(dir);
    }
    if (!found) return false;
    for (int it = 0; it < kMprMaxRefine; ++it) {
        dir = cross(sub(v2.w, v1.w), sub(v3_.w, v1.w));
        if (dot(dir, dir) < 1.0e-24f) break;
        dir = normalized(dir);
        const SV v4 = mpr_support(A, B, hv, dir);
        const float d4 = dot(v4.w, dir);
        float m = d4 - dot(v1.w, dir);
        const float m2 = d4 - dot(v2.w, dir), m3 = d4 - dot(v3_.w, dir);
        if (m2 < m) m = m2;
        if (m3 < m) m = m3;
        if (m <= kMprTol) break;
        const v3 x = cross(v4.w, v0);
        if (dot(v1.w, x) > 0.0f) {
            if (dot(v2.w, x) > 0.0f) v1 = v4; else v3_ = v4;
        } else {
            if (dot(v3_.w, x) > 0.0f) v2 = v4; else v1 = v4;
        }
    }
    float l[3];
    closest_triangle(v1.w, v2.w, v3_.w, l);
    const v3 pt = comb3(v1.w, v2.w, v3_.w, l);
    const float d = sqrtf(dot(pt, pt));
    *pa = comb3(v1.a, v2.a, v3_.a, l);
    *pb = comb3(v1.b, v2.b, v3_.b, l);
    if (d > 1.0e-7f) *n = scale(pt, -1.0f / d);
    else {
        const v3 pn = cross(sub(v2.w, v1.w), sub(v3_.w, v1.w));
        if (dot(pn, pn) < 1.0e-24f) return false;
        *n = neg(normalized(pn));
    }
    *sep = -d;
    return true;
}

// hull description resolved for this scene: vertices either in LDS (copied once per settle) or
// in the global pool
struct HullRef {
    int src;        // >= 0: first vertex (float4 index) in the global pool; < 0: ~src = first vertex of the LDS copy
    int count;
    v3 sc;          // bounding sphere centre (object frame)
    float sr;
};
static_assert(sizeof(HullRef) == 24, "HullRef layout");

__device__ __forceinline__ void make_shape(const WBody& wb, const HullRef& h, const float4* __restrict__ gv, Shape& s)
{
    s.g = gv + (h.src >= 0 ? h.src : 0);
    s.lds = h.src >= 0 ? -1 : ~h.src;
    s.count = h.count;
    s.R = wb.R;
    s.t = wb.t;
}

// Selections are written component by component on VALUES: a conditional copy of a whole v3 (`cond ? a : b`, or a struct store under
// a branch) is lowered to a copy through a selected POINTER, which pins the structs in scratch memory.
__device__ __forceinline__ v3 vsel(bool c, v3 a, v3 b) { return V(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// ---- narrowphase of one hull pair in three stages (same arithmetic as the oracle's hull_pair_contacts, which runs them one
// ---- after the other): (1) GJK / portal refinement + the previous manifold's points refreshed in the new poses (lane = pair),
// ---- (2) pairs that are new or lost a point: the face manifold, sixteen lanes per pair, (3) contact list (wave = scene)
struct MainResult {       // stage 1
    int type;             // 0 no contact; 1 contact pair whose manifold is BUILT in this step (new pair, or its manifold lost a point);
                          // 3 pair whose manifold of the previous step stays as it was built (its points refreshed)
    v3 n, pa, pb;
    float dist;           // distance, or the (negative) separation of an overlap
    GjkSeed seed;         // while stage 1 runs: the converged simplex (the pair cache takes it); in the result slot: n = the previous
                          // step's pair slot, i0 = 1 when that step left a manifold, i1 = points parked in the pair's stage slots
                          // (stage 1: the refreshed ones; after stage 2: the built manifold)
};

// constants of the contact persistence (oracle/settle_ref.c WARM_START, DRIFT_OFFSETS, NORMAL_COS, PLANE_DEPTH_WEIGHT, FM_*)
constexpr float kWarmStart = 0.8f;
constexpr float kDriftOffsets = 2.0f;
constexpr float kNormalCos = 0.9848f;
constexpr float kPlaneDepthWeight = 5.0f;
constexpr float kFmBandOffsets = 1.0f;
constexpr float kFmBandExtent = 0.25f;
constexpr float kFmAreaMin = 1.0e-6f;
constexpr float kFmInsideTol = 1.0e-6f;
constexpr float kFmParallel2 = 1.0e-6f;
constexpr float kFmDeeperOffsets = 0.05f;
constexpr float kFmDeepOffsets = 2.5f;

// Returns whether the pair cache takes the new simplex.
__device__ __forceinline__ bool pair_main(const WBody& wa, const WBody& wb, const HullRef& ha, const HullRef& hb,
                                          const f3* __restrict__ hv, const float4* __restrict__ gv, float margin, const GjkSeed cached,
                                          MainResult& r, int max_iter = kGjkMaxIter, bool* unfinished = nullptr)
{
    r.type = 0;
    Shape A, B;
    make_shape(wa, ha, gv, A);
    make_shape(wb, hb, gv, B);
    const v3 ca = add(m3_mul(wa.R, ha.sc), wa.t);
    const v3 cb = add(m3_mul(wb.R, hb.sc), wb.t);
    v3 pa, pb, n;
    float dist;
    r.seed.n = 0; r.seed.i0 = r.seed.i1 = r.seed.i2 = 0;
    bool ran_out = false;
    const int code = gjk_distance(A, B, hv, sub(ca, cb), margin, &pa, &pb, &dist, cached, &r.seed, max_iter, &ran_out);
    if (unfinished) {
        *unfinished = ran_out && max_iter < kGjkMaxIter;
        if (*unfinished) return false;
    }
    if (code == 2) return true;
    if (code == 0) {
        float sep;
        if (!mpr_penetration(A, B, hv, ca, cb, &n, &sep, &pa, &pb)) {
            overlap_fallback(A, B, hv, ca, cb, &n, &sep, &pa, &pb);
            if (sep > 0.0f) sep = 0.0f;
        }
        r.type = 1; r.n = n; r.pa = pa; r.pb = pb; r.dist = sep;
        return false;   // overlap keeps the previous cache entry
    }
    if (dist > margin) return true;
    r.type = 1;
    r.n = scale(sub(pa, pb), 1.0f / dist);
    r.pa = pa; r.pb = pb; r.dist = dist;
    return true;
}

// Persistent manifold of one hull pair as the previous step left it (oracle pmanifold; the impulses live in the step's
// impulse array at c_off): contact points in the two bodies' object frames, the normal the manifold was BUILT with in B's frame.
struct PM {
    int count, c_off;
    v3 nb;
    v3 la[4], lb[4];
    float pad[3];
};
static_assert(sizeof(PM) == 128, "PM layout");

// a contact point parked between the stages (and, in k_w_finish, until the contact list is written): witnesses on A and B,
// separation, the impulse it carries (float bits)
struct StagePt { v3 qa, qb; float sp; int imp; };
static_assert(sizeof(StagePt) == 32, "StagePt layout");

// stage 1, pairs with a manifold: the old points in the new poses -- separations along the new normal; a point whose witnesses
// drifted apart laterally or that left the contact band is lost, all of them when the normal turned against B since the manifold
// was built.  The points that stay are parked in `stage` in their order; returns their number and whether the manifold is LOST
// (a point went, or the closest points are deeper than all of it: it no longer holds the pair's deepest feature).
__device__ __forceinline__ int refresh_manifold(const MainResult& m, const PM& prev, const float* __restrict__ ln_prev, const WBody& wa,
                                                const WBody& wb, float margin, float contact_offset, StagePt* __restrict__ stage,
                                                bool& lost)
{
    const v3 npw = m3_mul(wb.R, prev.nb);
    const float lim = kDriftOffsets * contact_offset;
    const bool same_normal = dot(m.n, npw) >= kNormalCos;
    int no = 0;
    float omin = kInf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < prev.count && same_normal) {
            const v3 qa = add(m3_mul(wa.R, prev.la[i]), wa.t);
            const v3 qb = add(m3_mul(wb.R, prev.lb[i]), wb.t);
            const v3 d = sub(qa, qb);
            const float sp = dot(d, m.n);
            const v3 lat = sub(d, scale(m.n, sp));
            const bool keep = !(sp > margin) && !(dot(lat, lat) > lim * lim);
            if (keep) {
                StagePt t;
                t.qa = qa; t.qb = qb; t.sp = sp; t.imp = __float_as_int(ln_prev[i]);
                stage[no] = t;
                ++no;
                if (sp < omin) omin = sp;
            }
        }
    }
    lost = no < prev.count || m.dist < omin - kFmDeeperOffsets * contact_offset;
    return no;
}

// body vs table plane: four order-independent selections over all hull vertices in the band
// (same rule as the oracle's plane_contacts, oracle/settle_ref.c)
__device__ __forceinline__ float4 hull_vertex(const HullRef& h, const f3* __restrict__ hv, const float4* __restrict__ gv, int i)
{
    if (h.src < 0) {
        const f3 q = hv[~h.src + i];
        return make_float4(q.x, q.y, q.z, 1.0f);
    }
    return gv[h.src + i];
}

// ---- cooperative variant: one 16-lane sub-group per body (four bodies per wave round) ----------
// Every arg-min/arg-max is taken over (value, band ordinal)
// with the lower ordinal winning ties, which is exactly what the serial strict compares pick.
struct BandPt { v3 p; float d; int id; };
constexpr int kBandCap = 32;   // in-band vertices cached per body; beyond that the passes re-walk the hulls

// A 16-lane sub-group is one DPP row: rotating the row by 8, 4, 2, 1 lanes and combining leaves the
// reduction of all 16 lanes in every lane (the (value, ordinal) minimum is a total order, so the
// combine is associative and commutative) -- full-rate VALU moves instead of eight ds_bpermute trips.
template <int CTRL>
__device__ __forceinline__ void sg16_argmin_step(float& val, int& idx)
{
    const float ov = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(val), CTRL, 0xF, 0xF, true));
    const int oi = __builtin_amdgcn_mov_dpp(idx, CTRL, 0xF, 0xF, true);
    if (ov < val || (ov == val && oi < idx)) { val = ov; idx = oi; }
}
__device__ __forceinline__ void sg16_argmin(float& val, int& idx)
{
    sg16_argmin_step<0x128>(val, idx);   // row_ror:8
    sg16_argmin_step<0x124>(val, idx);   // row_ror:4
    sg16_argmin_step<0x122>(val, idx);   // row_ror:2
    sg16_argmin_step<0x121>(val, idx);   // row_ror:1
}

// broadcast the winning lane's point (and its vertex number) within the 16-lane sub-group
__device__ __forceinline__ void sg16_fetch(bool mine, int sg, v3& p, float& d, int& id)
{
    const unsigned om = (unsigned)(__ballot(mine) >> (16 * sg)) & 0xffffu;
    const int owner = 16 * sg + (om ? __ffs(om) - 1 : 0);
    p.x = __shfl(p.x, owner, 64); p.y = __shfl(p.y, owner, 64); p.z = __shfl(p.z, owner, 64);
    d = __shfl(d, owner, 64);
    id = __shfl(id, owner, 64);
}

// f(p, d, ordinal, id) for every in-band vertex of the body, ordinals in hull/vertex order; id = the vertex's number within the
// body, hull << 8 | vertex (the oracle's PLANE_ID: what a persistent table contact is matched by)
template <class F>
__device__ __forceinline__ int sg16_walk_band(const WBody& w, const HullRef* lh, const f3* __restrict__ hv,
                                              const float4* __restrict__ gv, int lh_begin,
                                              int lh_end, float plane_z, float margin, int sg, int sl, F&& f)
{
    int ord = 0;
    for (int h = lh_begin; h < lh_end; ++h) {
        const HullRef H = lh[h];
        // a hull whose bounding sphere clears the contact band has no in-band vertex: skipping it leaves the
        // ordinals (they count in-band vertices only) and every selection untouched.  The slack (0.1 mm) is
        // orders of magnitude above the rounding of the sphere and vertex transforms.
        if (fmaf(w.R.m[8], H.sc.z, fmaf(w.R.m[7], H.sc.y, w.R.m[6] * H.sc.x)) + w.t.z - H.sr - plane_z > margin + 1.0e-4f) continue;
        for (int i0 = 0; i0 < H.count; i0 += 16) {
            const int i = i0 + sl;
            bool in = false;
            v3 p = V(0, 0, 0);
            float d = 0.0f;
            if (i < H.count) {
                const float4 q = hull_vertex(H, hv, gv, i);
                p = add(m3_mul(w.R, V(q.x, q.y, q.z)), w.t);
                d = p.z - plane_z;
                in = d <= margin;
            }
            const unsigned sm = (unsigned)(__ballot(in) >> (16 * sg)) & 0xffffu;
            if (in) f(p, d, ord + (int)__popc(sm & ((1u << sl) - 1u)), ((h - lh_begin) << 8) | i);
            ord += (int)__popc(sm);
        }
    }
    return ord;
}

// `ids`: the vertex numbers of the selected contacts (see sg16_walk_band)
__device__ void plane_contacts_sg16(const WBody& w, const HullRef* lh, const f3* __restrict__ hv,
                                    const float4* __restrict__ gv, int lh_begin, int lh_end,
                                    float plane_z, float margin, BandPt* band, int sg, int sl, RawContacts& out, int* ids)
{
    constexpr int kNone = 0x7fffffff;
    out.count = 0;
    out.n = V(0, 0, 1);
    ids[0] = ids[1] = ids[2] = ids[3] = 0;
    // pass 0: deepest in-band vertex; the band is cached in LDS on the way
    float bv = kInf; int bi = kNone; v3 bp = V(0, 0, 0); int bid = 0;
    const int band_n = sg16_walk_band(w, lh, hv, gv, lh_begin, lh_end, plane_z, margin, sg, sl,
                                      [&](v3 p, float d, int ord, int id) {
                                          if (d < bv || bi == kNone) { bv = d; bi = ord; bp = p; bid = id; }
                                          if (ord < kBandCap) { BandPt b; b.p = p; b.d = d; b.id = id; band[ord] = b; }
                                      });
    if (band_n == 0) return;
    const bool cached = band_n <= kBandCap;
    float s0; v3 p0; int id0 = bid;
    {
        float v = bi == kNone ? kInf : bv; int idx = bi;
        sg16_argmin(v, idx);
        s0 = bv; p0 = bp;
        sg16_fetch(bi == idx, sg, p0, s0, id0);
    }
    // pass 1: farthest from p0, deep points preferred
    float b1 = kInf; int i1 = kNone; v3 p1 = V(0, 0, 0); float s1 = 0.0f; int id1 = 0;
    auto f1 = [&](v3 p, float d, int ord, int id) {
        const v3 dd = sub(p, p0);
        const float score = sqrtf(dot(dd, dd)) - kPlaneDepthWeight * (d - s0);
        if (score > 0.0f && (i1 == kNone || -score < b1)) { b1 = -score; i1 = ord; p1 = p; s1 = d; id1 = id; }
    };
    if (cached) {
        for (int k = sl; k < band_n; k += 16) { const BandPt b = band[k]; f1(b.p, b.d, k, b.id); }
    } else {
        sg16_walk_band(w, lh, hv, gv, lh_begin, lh_end, plane_z, margin, sg, sl, f1);
    }
    bool have1;
    {
        float v = b1; int idx = i1;
        sg16_argmin(v, idx);
        have1 = idx != kNone;
        sg16_fetch(have1 && i1 == idx, sg, p1, s1, id1);
    }
    int nk = 1;
    out.pa[0] = p0; out.sep[0] = s0; ids[0] = id0;
    if (have1) {
        out.pa[1] = p1; out.sep[1] = s1; ids[1] = id1; nk = 2;
        const v3 e = sub(p1, p0);
        const float el = sqrtf(dot(e, e));
        float b2 = kInf, b3 = kInf; int i2 = kNone, i3 = kNone;
        v3 p2 = V(0, 0, 0), p3 = V(0, 0, 0); float s2 = 0.0f, s3 = 0.0f; int id2 = 0, id3 = 0;
        auto f2 = [&](v3 p, float d, int ord, int id) {
            const float a = dot(cross(e, sub(p, p0)), out.n);
            const float pen = kPlaneDepthWeight * (d - s0) * el;
            const float hi = a - pen, lo = a + pen;
            if (hi > 0.0f && (i2 == kNone || -hi < b2)) { b2 = -hi; i2 = ord; p2 = p; s2 = d; id2 = id; }
            if (lo < 0.0f && (i3 == kNone || lo < b3)) { b3 = lo; i3 = ord; p3 = p; s3 = d; id3 = id; }
        };
        if (cached) {
            for (int k = sl; k < band_n; k += 16) { const BandPt b = band[k]; f2(b.p, b.d, k, b.id); }
        } else {
            sg16_walk_band(w, lh, hv, gv, lh_begin, lh_end, plane_z, margin, sg, sl, f2);
        }
        float v2 = b2, v3_ = b3; int x2 = i2, x3 = i3;
        sg16_argmin(v2, x2);
        sg16_argmin(v3_, x3);
        const bool have2 = x2 != kNone, have3 = x3 != kNone;
        sg16_fetch(have2 && i2 == x2, sg, p2, s2, id2);
        sg16_fetch(have3 && i3 == x3, sg, p3, s3, id3);
        if (have2) { out.pa[2] = p2; out.sep[2] = s2; ids[2] = id2; nk = 3; }
        if (have3) {
            if (have2) { out.pa[3] = p3; out.sep[3] = s3; ids[3] = id3; nk = 4; }
            else { out.pa[2] = p3; out.sep[2] = s3; ids[2] = id3; nk = 3; }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out.pb[i] = V(out.pa[i].x, out.pa[i].y, plane_z);
    out.count = nk;
}

// `head`: the first contact of its friction patch (= the manifold of one hull pair / of one body against the table); carried
// in the sign of `til` (the tangent construction's 1 / |a| is positive)
__device__ __forceinline__ void fill_contact(Contact* c, int a, int b, const WBody& wa, const WBody* wbb, v3 pa, v3 pb,
                                             v3 n, float sep, float rest, float e, bool head, float warm)
{
    Contact k;
    k.ra = sub(pa, wa.x);
    k.rb = wbb ? sub(pb, wbb->x) : V(0, 0, 0);
    k.n = n;
    k.err = sep - rest;
    k.kn = 0.0f;
    k.kt1 = __int_as_float(a); k.kt2 = __int_as_float(b);   // consumed by prep_contact
    k.ln = warm;                                            // the sweeps start from it (k_w_solve applies it first)
    k.lt1 = 0.0f; k.lt2 = 0.0f;
    k.bounce = e;                                          // restitution until prep_contact
    k.til = head ? -1.0f : 1.0f;                           // sign = patch head; magnitude set by prep_contact
    *c = k;
}

// The CENTRE ROW of a patch of three or four points (oracle patch_centre): one more normal row at the mean of the points' arms with
// their mean separation, the HEAD of its patch in the list (the points follow; none of them carries the head mark); no friction
// acts at it -- marked by lt1 < 0 until prep_contact, by kt1 < 0 from then on.
__device__ __forceinline__ void fill_centre(Contact* c, int a, int b, int m, v3 sum_ra, v3 sum_rb, float sum_sep,
                                            v3 n, float rest, float e, float warm)
{
    // (the sums run over the points in their order, from +0: sum = sum + x)
    const float inv = 1.0f / (float)m;
    Contact k;
    k.ra = scale(sum_ra, inv);
    k.rb = scale(sum_rb, inv);
    k.n = n;
    k.err = sum_sep * inv - rest;
    k.kn = 0.0f;
    k.kt1 = __int_as_float(a); k.kt2 = __int_as_float(b);
    k.ln = warm;
    k.lt1 = -1.0f; k.lt2 = 0.0f;                            // the centre mark (prep_contact moves it to kt1)
    k.bounce = e;
    k.til = -1.0f;                                         // the patch's head
    *c = k;
}

// world AABB overlap of two hulls within margin (|R| * half extents around R c + t)
__device__ __forceinline__ bool aabb_overlap(const WBody& wa, const slhip_hull& ha, const WBody& wb, const slhip_hull& hb,
                                             float margin)
{
    const v3 ca = add(m3_mul(wa.R, V(ha.aabb_center[0], ha.aabb_center[1], ha.aabb_center[2])), wa.t);
    const v3 cb = add(m3_mul(wb.R, V(hb.aabb_center[0], hb.aabb_center[1], hb.aabb_center[2])), wb.t);
    float ea[3], eb[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        ea[r] = fmaf(fabsf(wa.R.m[3 * r + 2]), ha.aabb_half[2], fmaf(fabsf(wa.R.m[3 * r + 1]), ha.aabb_half[1], fabsf(wa.R.m[3 * r]) * ha.aabb_half[0]));
        eb[r] = fmaf(fabsf(wb.R.m[3 * r + 2]), hb.aabb_half[2], fmaf(fabsf(wb.R.m[3 * r + 1]), hb.aabb_half[1], fabsf(wb.R.m[3 * r]) * hb.aabb_half[0]));
    }
    if (fabsf(ca.x - cb.x) > ea[0] + eb[0] + margin) return false;
    if (fabsf(ca.y - cb.y) > ea[1] + eb[1] + margin) return false;
    if (fabsf(ca.z - cb.z) > ea[2] + eb[2] + margin) return false;
    return true;
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

__device__ void prep_contact(Contact* cp, const WBody* wbs, float bounce_threshold, float inv_dt)
{
    Contact c = *cp;
    const int ia = __float_as_int(c.kt1), ib = __float_as_int(c.kt2);
    const float e = c.bounce;
    const WBody& a = wbs[ia];
    const WBody* b = ib >= 0 ? &wbs[ib] : nullptr;
    v3 t1, t2;
    const float til = tangent_inv_len(c.n);
    tangents_cached(c.n, til, &t1, &t2);
    c.kn = eff_mass(a, b, c.ra, c.rb, c.n);
    c.kt1 = eff_mass(a, b, c.ra, c.rb, t1);
    c.kt2 = eff_mass(a, b, c.ra, c.rb, t2);
    v3 rel = vel_at(a, c.ra);
    if (b) rel = sub(rel, vel_at(*b, c.rb));
    const float vn0 = dot(rel, c.n);
    // The oracle builds the normal row's target velocity in every solve -- speculative contacts (err > 0) may close their gap,
    // penetrating ones are pushed out in the biased sweeps only, a rebound above the bounce threshold overrides both.  The same
    // expressions once per step: `err` becomes the target of the biased sweeps, `bounce` that of the unbiased ones.
    float bounce = -3.0e38f;
    if (vn0 < -bounce_threshold && e > 0.0f) bounce = -e * vn0;
    const float err = c.err;
    float tb, tu;
    if (err > 0.0f) { tb = -err * inv_dt; tu = tb; }
    else { tb = -0.8f * err * inv_dt; tu = 0.0f; }
    if (bounce > tb) tb = bounce;
    if (bounce > tu) tu = bounce;
    cp->err = tb;
    cp->kn = c.kn; cp->kt1 = c.kt1; cp->kt2 = c.kt2; cp->bounce = tu;
    if (c.lt1 < 0.0f) { cp->kt1 = -1.0f; cp->lt1 = 0.0f; }   // a patch's centre row: no friction acts at it
    cp->til = c.til < 0.0f ? -til : til;
}

// Gauss-Seidel sweep over the contacts of ONE group (all share the same two bodies) by a PAIR of
// adjacent lanes: lane `side` 0 owns body a, lane 1 owns body b (or nothing, for a plane group).
// Each lane keeps its body's velocities, inverse inertia and mass in registers for the whole group
// and writes them back once; the only exchange per row is the body's velocity at the contact point
// (one DPP quad permute per component).  Lane 1 works with the negated relative velocity and the
// negated impulse: IEEE negation commutes with every + - * fma, so its results are bit-for-bit
// those of the oracle's solve_contact (a - b, then -J on body b) applied to the contacts in order.
struct BodyRegs {
    v3 v, w;
    m3 Iinv;
    float inv_mass;
    bool dynamic;
};

template <class Body>
__device__ __forceinline__ void load_regs(const Body& b, BodyRegs& r)
{
    r.v = b.v; r.w = b.w; r.Iinv = b.Iinv_w; r.inv_mass = b.inv_mass; r.dynamic = b.dynamic != 0;
}

// value held by the other lane of the (2k, 2k+1) pair: DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float pair_swap(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ v3 pair_swap(v3 a) { return V(pair_swap(a.x), pair_swap(a.y), pair_swap(a.z)); }

__device__ __forceinline__ void apply_mine(BodyRegs& m, v3 r, v3 J)
{
    if (m.dynamic) {
        m.v = madd(m.v, J, m.inv_mass);
        m.w = add(m.w, m3_mul(m.Iinv, cross(r, J)));
    }
}

// The step's contact list: entries [0, n_lds) live in LDS (`lds`), the rest are read from (and their impulses written to)
// the scene's list in global memory (`glb`, indexed like the list): no cap but the capacity of the scratch.
struct ContactList {
    Contact* lds;
    Contact* glb;
    int n_lds;
    __device__ __forceinline__ Contact load(int i) const { return i < n_lds ? lds[i] : glb[i]; }
    __device__ __forceinline__ Contact* at(int i) const { return i < n_lds ? lds + i : glb + i; }
    __device__ __forceinline__ float ln(int i) const { return i < n_lds ? lds[i].ln : glb[i].ln; }
};

// Developer build -DSLHIP_ROW_PROFILE (tools/row_profile.py): shader-clock stamps (s_memtime) around the parts of a contact row of the
// LDS-resident sweep, summed over the lane pairs' first lanes -- [0] rows, [1] cycles from the row's start until its contact has
// arrived from LDS, [2] the normal row's arithmetic (velocity at the point, DPP swap, impulse, both bodies updated), [3] patches,
// [4] cycles of a patch's friction rows (two anchors, or one), [5] group visits, [6] cycles of a visit outside its rows (body
// registers in / out, set-up).  A stamp costs an s_memtime + s_waitcnt: the figures are upper bounds of the unstamped code.
#ifdef SLHIP_ROW_PROFILE
__device__ unsigned long long g_row_prof[8];
#define ROWP_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define ROWP_ADD(slot_, val_) do { if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_row_prof[slot_], (unsigned long long)(val_)); } while (0)
#else
#define ROWP_STAMP(v)
#define ROWP_ADD(slot_, val_)
#endif

// Round 6 measured the parts of a row of this loop with s_memtime stamps (tools/row_profile.py, profiles/r06/row_profile.txt: 228
// cycles until the row's 72-byte contact has arrived from LDS, 337 cycles of arithmetic, 1 457 cycles of friction rows per patch
// of 2.8 points) and then built the loop that hides the first and halves the last -- the next contact fetched while the current
// row runs (two register sets in turns: no copies), a patch's anchors kept in registers from their normal rows to the friction
// rows; bit-exact (61 settle tests), 177 instead of 156 registers -- : k_w_solve 1.995 -> 1.987 ms per launch over 32 768 scenes.
// The row's latency is not what bounds the kernel (the SIMD's other wave fills the wait); taken back (git history:
// "k_w_solve: the next contact is fetched while the current row runs").
// (the group lies in the LDS-resident part of the list: every row reads its contact where it needs it -- LDS latency is short, and
// carrying a prefetched contact around the loop costs eighteen register moves per row)
template <class Body>
__device__ void solve_group_lds(Contact* ac, int begin, int end, int ia, int ib, int side, Body* wbs, float inv_dt,
                            bool biased, float plane_mu_s, float plane_mu_d)
{
    if (begin >= end) return;
    ROWP_STAMP(tg0);
#ifdef SLHIP_ROW_PROFILE
    unsigned long long rows_cyc = 0ull;
#endif
    const bool has_b = ib >= 0;
    const int mine = side ? ib : ia;
    BodyRegs M;
    if (mine >= 0) load_regs(wbs[mine], M);
    else {   // the idle lane of a plane group: +0 velocities, so a - b == a exactly
        M.v = V(0, 0, 0); M.w = V(0, 0, 0); M.inv_mass = 0.0f; M.dynamic = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) M.Iinv.m[k] = 0.0f;
    }
    const bool other_dynamic = pair_swap(M.dynamic ? 1.0f : 0.0f) != 0.0f;
    if (!M.dynamic && !other_dynamic) return;  // the oracle invalidates such contacts in prep
    const float mu_s = 0.5f * (wbs[ia].mu_s + (has_b ? wbs[ib].mu_s : plane_mu_s));
    const float mu_d = 0.5f * (wbs[ia].mu_d + (has_b ? wbs[ib].mu_d : plane_mu_d));
    const float sgn = side ? -1.0f : 1.0f;
    // Friction patches (oracle solve_patch; PhysX's ePATCH model): the contacts of one manifold are contiguous, the first one
    // carries the head mark.  Normal rows in order; when the patch ends, the friction rows of its (at most two) anchors -- its
    // first two contacts -- against their share of the patch's accumulated normal impulse.
    float nsum = 0.0f;
    int p0 = begin;
    for (int ci = begin; ci < end; ++ci) {
        ROWP_STAMP(tr0);
        const Contact c = ac[ci];
#ifdef SLHIP_ROW_PROFILE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        ROWP_STAMP(tr1);
        if (c.til < 0.0f) { nsum = 0.0f; p0 = c.kt1 < 0.0f ? ci + 1 : ci; }   // (behind the patch's centre row, if it has one)
        const v3 r = side ? c.rb : c.ra;
        v3 pv = add(M.v, cross(M.w, r));
        v3 d = sub(pv, pair_swap(pv));              // side 0: a - b, side 1: b - a
        const float vn = sgn * dot(d, c.n);
        const float target = biased ? c.err : c.bounce;   // prep_contact
        float dl = (target - vn) * c.kn;
        float ln = c.ln + dl;
        if (ln < 0.0f) ln = 0.0f;
        dl = ln - c.ln;
        apply_mine(M, r, scale(c.n, sgn * dl));
        if (side == 0) ac[ci].ln = ln;
        nsum = nsum + ln;
        const bool last = ci + 1 == end || ac[ci + 1].til < 0.0f;
#ifdef SLHIP_ROW_PROFILE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        { ROWP_STAMP(tr2); ROWP_ADD(0, 1); ROWP_ADD(1, tr1 - tr0); ROWP_ADD(2, tr2 - tr1); rows_cyc += tr2 - tr0; }
#endif
        if (!last) continue;
        ROWP_STAMP(tf0);
        const int anchors = ci - p0 >= 1 ? 2 : 1;
        const float share = anchors == 2 ? 0.5f * nsum : nsum;
        for (int ai = 0; ai < anchors; ++ai) {
            const Contact q = ac[p0 + ai];
            const v3 rq = side ? q.rb : q.ra;
            pv = add(M.v, cross(M.w, rq));
            d = sub(pv, pair_swap(pv));
            v3 t1, t2;
            tangents_cached(q.n, fabsf(q.til), &t1, &t2);
            float l1 = q.lt1 - (sgn * dot(d, t1)) * q.kt1;
            float l2 = q.lt2 - (sgn * dot(d, t2)) * q.kt2;
            const float mag2 = fmaf(l2, l2, l1 * l1);
            const float lim_s = mu_s * share;
            if (mag2 > lim_s * lim_s) {
                const float mag = sqrtf(mag2);
                const float k = (mu_d * share) / mag;
                l1 *= k; l2 *= k;
            }
            const float d1 = l1 - q.lt1, d2 = l2 - q.lt2;
            apply_mine(M, rq, madd(scale(t1, sgn * d1), t2, sgn * d2));
            if (side == 0) { ac[p0 + ai].lt1 = l1; ac[p0 + ai].lt2 = l2; }
        }
#ifdef SLHIP_ROW_PROFILE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        { ROWP_STAMP(tf1); ROWP_ADD(3, 1); ROWP_ADD(4, tf1 - tf0); rows_cyc += tf1 - tf0; }
#endif
    }
    if (M.dynamic) { wbs[mine].v = M.v; wbs[mine].w = M.w; }
#ifdef SLHIP_ROW_PROFILE
    { ROWP_STAMP(tg1); ROWP_ADD(5, 1); ROWP_ADD(6, (tg1 - tg0) - rows_cyc); }
#endif
}

// what the friction rows need of a patch's anchor contact, kept from the moment it passes through its normal row (no second
// fetch: beyond the LDS-resident part that would be an exposed round trip to the L2 per patch)
struct Anchor { v3 r, n; float til, kt1, kt2, lt1, lt2; };

template <class Body>
__device__ void solve_group(const ContactList ac, int begin, int end, int ia, int ib, int side, Body* wbs, float inv_dt,
                            bool biased, float plane_mu_s, float plane_mu_d)
{
    if (begin >= end) return;
    if (end <= ac.n_lds) { solve_group_lds(ac.lds, begin, end, ia, ib, side, wbs, inv_dt, biased, plane_mu_s, plane_mu_d); return; }
    const bool has_b = ib >= 0;
    const int mine = side ? ib : ia;
    BodyRegs M;
    if (mine >= 0) load_regs(wbs[mine], M);
    else {   // the idle lane of a plane group: +0 velocities, so a - b == a exactly
        M.v = V(0, 0, 0); M.w = V(0, 0, 0); M.inv_mass = 0.0f; M.dynamic = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) M.Iinv.m[k] = 0.0f;
    }
    const bool other_dynamic = pair_swap(M.dynamic ? 1.0f : 0.0f) != 0.0f;
    if (!M.dynamic && !other_dynamic) return;  // the oracle invalidates such contacts in prep
    const float mu_s = 0.5f * (wbs[ia].mu_s + (has_b ? wbs[ib].mu_s : plane_mu_s));
    const float mu_d = 0.5f * (wbs[ia].mu_d + (has_b ? wbs[ib].mu_d : plane_mu_d));
    const float sgn = side ? -1.0f : 1.0f;
    // Friction patches (oracle solve_patch; PhysX's ePATCH model): the contacts of one manifold are contiguous, the first one
    // carries the head mark.  Normal rows in order; when the patch ends, the friction rows of its (at most two) anchors -- its
    // first two contacts -- against their share of the patch's accumulated normal impulse.  The loop walks patch by patch; the
    // next contact is fetched while the current row runs (beyond the LDS-resident part it comes from global memory).
    Contact nxt = ac.load(begin);
    int ci = begin;
    bool more = true;
    float nsum = 0.0f;
    auto normal_row = [&](Anchor* keep) {
        const Contact c = nxt;
        more = ci + 1 < end;
        if (more) nxt = ac.load(ci + 1);
        const v3 r = side ? c.rb : c.ra;
        if (keep) { keep->r = r; keep->n = c.n; keep->til = c.til; keep->kt1 = c.kt1; keep->kt2 = c.kt2; keep->lt1 = c.lt1; keep->lt2 = c.lt2; }
        const v3 pv = add(M.v, cross(M.w, r));
        const v3 d = sub(pv, pair_swap(pv));        // side 0: a - b, side 1: b - a
        const float vn = sgn * dot(d, c.n);
        const float target = biased ? c.err : c.bounce;   // prep_contact
        float dl = (target - vn) * c.kn;
        float ln = c.ln + dl;
        if (ln < 0.0f) ln = 0.0f;
        dl = ln - c.ln;
        apply_mine(M, r, scale(c.n, sgn * dl));
        if (side == 0) ac.at(ci)->ln = ln;
        nsum = nsum + ln;
        ++ci;
    };
    auto friction_rows = [&](const Anchor& q, int at, float share) {
        const v3 pv = add(M.v, cross(M.w, q.r));
        const v3 d = sub(pv, pair_swap(pv));
        v3 t1, t2;
        tangents_cached(q.n, fabsf(q.til), &t1, &t2);
        float l1 = q.lt1 - (sgn * dot(d, t1)) * q.kt1;
        float l2 = q.lt2 - (sgn * dot(d, t2)) * q.kt2;
        const float mag2 = fmaf(l2, l2, l1 * l1);
        const float lim_s = mu_s * share;
        if (mag2 > lim_s * lim_s) {
            const float mag = sqrtf(mag2);
            const float k = (mu_d * share) / mag;
            l1 *= k; l2 *= k;
        }
        const float d1 = l1 - q.lt1, d2 = l2 - q.lt2;
        apply_mine(M, q.r, madd(scale(t1, sgn * d1), t2, sgn * d2));
        if (side == 0) { Contact* w = ac.at(at); w->lt1 = l1; w->lt2 = l2; }
    };
    while (ci < end) {
        nsum = 0.0f;
        if (nxt.kt1 < 0.0f) normal_row(nullptr);          // the patch's centre row: first, never an anchor (three or four points follow)
        const int p0 = ci;
        Anchor a0, a1;
        normal_row(&a0);
        if (more && !(nxt.til < 0.0f)) {
            normal_row(&a1);
            while (more && !(nxt.til < 0.0f)) normal_row(nullptr);
            const float share = 0.5f * nsum;
            friction_rows(a0, p0, share);
            friction_rows(a1, p0 + 1, share);
        } else {
            friction_rows(a0, p0, nsum);
        }
    }
    if (M.dynamic) { wbs[mine].v = M.v; wbs[mine].w = M.w; }
}

// Warm start of ONE group by its lane pair (oracle: the loop before the first sweep): every contact's carried normal impulse is
// applied to the two bodies, contacts in order; same lane roles and sign conventions as solve_group.
template <class Body>
__device__ void warm_group(const ContactList ac, int begin, int end, int ia, int ib, int side, Body* wbs)
{
    if (begin >= end) return;
    const int mine = side ? ib : ia;
    if (mine < 0) return;
    BodyRegs M;
    load_regs(wbs[mine], M);
    if (!M.dynamic) return;
    const float sgn = side ? -1.0f : 1.0f;
    for (int ci = begin; ci < end; ++ci) {
        const Contact c = ac.load(ci);
        const v3 r = side ? c.rb : c.ra;
        apply_mine(M, r, scale(c.n, sgn * c.ln));
    }
    wbs[mine].v = M.v; wbs[mine].w = M.w;
}

// D6 joint drive of ManipulationSim (same arithmetic as the oracle's solve_drive)
__device__ void solve_drive(const slhip_body& b, WBody& w, DriveAcc& acc, const slhip_settle_params& prm, bool biased)
{
    if (!(b.drive_flags & 1u) || !w.dynamic) return;
    const float dt = prm.dt;
    const float k = b.drive_params[0], d = b.drive_params[1], flim = b.drive_params[2] * dt;
    quat qj; qj.x = b.drive_frame[0]; qj.y = b.drive_frame[1]; qj.z = b.drive_frame[2]; qj.w = b.drive_frame[3];
    m3 J;
    quat_to_m3(qj, J);
    const v3 r = sub(w.t, w.x);
    const v3 err = sub(w.t, V(b.drive_target[0], b.drive_target[1], b.drive_target[2]));
    const float den = d + dt * k;
    const float gamma = 1.0f / (dt * den);
    const float beta = dt * k / den;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const v3 ax = V(J.m[a], J.m[3 + a], J.m[6 + a]);
        const v3 rn = cross(r, ax);
        const float K = w.inv_mass + dot(cross(m3_mul(w.Iinv_w, rn), r), ax);
        const float meff = 1.0f / (K + gamma);
        const float u = dot(vel_at(w, r), ax);
        const float C = dot(err, ax);
        float dlam = -meff * (u + (beta / dt) * C + gamma * acc.dl[a]);
        float lam = acc.dl[a] + dlam;
        if (lam > flim) lam = flim;
        if (lam < -flim) lam = -flim;
        dlam = lam - acc.dl[a];
        acc.dl[a] = lam;
        const v3 Jimp = scale(ax, dlam);
        w.v = madd(w.v, Jimp, w.inv_mass);
        w.w = add(w.w, m3_mul(w.Iinv_w, cross(r, Jimp)));
    }
    quat qc; qc.x = -qj.x; qc.y = -qj.y; qc.z = -qj.z; qc.w = qj.w;
    const quat qe = quat_mul(w.q, qc);
    const float sgn = qe.w < 0.0f ? -2.0f : 2.0f;
    const v3 theta = V(qe.x * sgn, qe.y * sgn, qe.z * sgn);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(b.drive_flags & (2u << a))) continue;
        const v3 ax = V(J.m[a], J.m[3 + a], J.m[6 + a]);
        const float K = dot(m3_mul(w.Iinv_w, ax), ax);
        if (!(K > 0.0f)) continue;
        const float bias = biased ? 0.8f * dot(theta, ax) / dt : 0.0f;
        const float dlam = -(dot(w.w, ax) + bias) / K;
        acc.da[a] += dlam;
        w.w = add(w.w, m3_mul(w.Iinv_w, scale(ax, dlam)));
    }
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
    w.mu_s = b.mu_s; w.mu_d = b.mu_d;
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

// mass-normalised kinetic energy of the sleep test (oracle step_scene (k)): 0.5 (v.v + w.(I w) / m), I = inverse of inv_inertia
// in object axes by cofactors
// (L: the 3x3 of the body record's inv_inertia rows, L[3 r + c] = inv_inertia[4 r + c])
__device__ __forceinline__ float kinetic_energy(const float (&L)[9], const m3& R, v3 v, v3 w, float inv_mass)
{
    const v3 wl = m3_tmul(R, w);
    const float c00 = L[4] * L[8] - L[5] * L[7], c01 = L[5] * L[6] - L[3] * L[8], c02 = L[3] * L[7] - L[4] * L[6];
    const float c11 = L[0] * L[8] - L[2] * L[6], c12 = L[1] * L[6] - L[0] * L[7], c22 = L[0] * L[4] - L[1] * L[3];
    const float det = fmaf(L[2], c02, fmaf(L[1], c01, L[0] * c00));
    const v3 iw = V(fmaf(c02, wl.z, fmaf(c01, wl.y, c00 * wl.x)), fmaf(c12, wl.z, fmaf(c11, wl.y, c01 * wl.x)),
                    fmaf(c22, wl.z, fmaf(c12, wl.y, c02 * wl.x)));
    const float ang = det != 0.0f ? dot(wl, iw) / det * inv_mass : 0.0f;
    return 0.5f * (dot(v, v) + ang);
}

__device__ __forceinline__ void store_velocities(slhip_body& b, v3 v, v3 w)
{
    b.lin_vel[0] = v.x; b.lin_vel[1] = v.y; b.lin_vel[2] = v.z;
    b.ang_vel[0] = w.x; b.ang_vel[1] = w.y; b.ang_vel[2] = w.z;
}

__device__ __forceinline__ void store_pose(slhip_body& b, const m3& R, v3 t)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) b.pose[4 * r + c] = R.m[3 * r + c];
    b.pose[3] = t.x; b.pose[7] = t.y; b.pose[11] = t.z;
    b.pose[12] = 0.0f; b.pose[13] = 0.0f; b.pose[14] = 0.0f; b.pose[15] = 1.0f;
}

// the spring drive of a ManipulationSim body inside a solver wave: the working body put together from its pose in the scratch and
// its solver part in LDS (driven bodies are rare: one per manipulation scene)
__device__ void solve_drive(const slhip_body& b, WBody& w, DriveAcc& acc, const slhip_settle_params& prm, bool biased);
__device__ __forceinline__ void drive_body(const slhip_body& b, const WBody& pose, SBody& sb, DriveAcc& acc, const slhip_settle_params& prm, bool biased)
{
    WBody w = pose;
    w.v = sb.v; w.w = sb.w; w.Iinv_w = sb.Iinv_w; w.inv_mass = sb.inv_mass; w.dynamic = sb.dynamic;
    solve_drive(b, w, acc, prm, biased);
    sb.v = w.v; sb.w = w.w;
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
    for (int k = 0; k < 4; ++k) { bodies[me].lin_vel[k] = 0.0f; bodies[me].ang_vel[k] = 0.0f; bodies[me].stab[k] = 0.0f; }
    bodies[me].flags &= ~SLHIP_BODY_FROZEN;
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

// exclusive prefix sum of a small per-lane count over the wave, plus the total
__device__ __forceinline__ int wave_excl_scan(int v, int& total)
{
    const int lane = threadIdx.x & 63;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// candidate hull pair, 64 bits: low word = body a | body b << 16 (the pair group's key), high word = hull of a | hull of b << 16
// (hull numbers are local to their body)
constexpr int kMaxHullsPerBody = 65535;
typedef unsigned long long HullPair;
__device__ __forceinline__ HullPair hp_pack(int ba, int bb, int ha, int hb)
{
    return (HullPair)((unsigned)ba | ((unsigned)bb << 16)) | ((HullPair)((unsigned)ha | ((unsigned)hb << 16)) << 32);
}
__device__ __forceinline__ int hp_ba(HullPair e) { return (int)((unsigned)e & 0xffffu); }
__device__ __forceinline__ int hp_bb(HullPair e) { return (int)((unsigned)e >> 16); }
__device__ __forceinline__ int hp_ha(HullPair e) { return (int)((unsigned)(e >> 32) & 0xffffu); }
__device__ __forceinline__ int hp_hb(HullPair e) { return (int)((unsigned)(e >> 48)); }
__device__ __forceinline__ unsigned hp_key(HullPair e) { return (unsigned)e; }      // the body pair

// solver group = all contacts between one body pair (b = kNoBody: body a against the plane);
// [begin, end) is its range in the step's contact list
struct Group { unsigned short a, b, begin, end, color; };
constexpr int kNoBody = 0xffff;
static_assert(sizeof(Group) == 10, "Group layout");
static_assert(SLHIP_MAX_BODIES < kNoBody, "16-bit body fields");

// boolean overlap (scene.cpp:355-385): one lane per body
__global__ __launch_bounds__(64) void k_overlap(const slhip_settle_scene* __restrict__ scenes,
                                                const slhip_body* __restrict__ bodies_all,
                                                const slhip_hull* __restrict__ hulls,
                                                const float* __restrict__ hull_verts, uint8_t* __restrict__ flags)
{
    __shared__ WBody wb[SLHIP_MAX_BODIES];   // (a query, not the hot loop: the static worst case)
    const slhip_settle_scene sc = scenes[blockIdx.x];
    const slhip_body* b = bodies_all + sc.body_begin;
    const int nb = (int)(sc.body_end - sc.body_begin);
    if (nb > SLHIP_MAX_BODIES) return;
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
                    const v3 sa = V(hulls[ha].sphere[0], hulls[ha].sphere[1], hulls[ha].sphere[2]);
                    const v3 sb = V(hulls[hb].sphere[0], hulls[hb].sphere[1], hulls[hb].sphere[2]);
                    const v3 ca = add(m3_mul(wb[i].R, sa), wb[i].t);
                    const v3 cb = add(m3_mul(wb[j].R, sb), wb[j].t);
                    const v3 dd = sub(ca, cb);
                    const float r2 = hulls[ha].sphere[3] + hulls[hb].sphere[3];
                    if (dot(dd, dd) > r2 * r2) continue;
                    Shape A, B;
                    A.g = reinterpret_cast<const float4*>(hull_verts) + hulls[ha].vtx_begin; A.lds = -1;
                    A.count = (int)hulls[ha].vtx_count; A.R = wb[i].R; A.t = wb[i].t;
                    B.g = reinterpret_cast<const float4*>(hull_verts) + hulls[hb].vtx_begin; B.lds = -1;
                    B.count = (int)hulls[hb].vtx_count; B.R = wb[j].R; B.t = wb[j].t;
                    v3 pa, pb;
                    float dist;
                    if (gjk_distance(A, B, (const f3*)nullptr, dd, 0.0f, &pa, &pb, &dist) == 0) hit = true;
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

#include "slhip_settle_wide.inc"

}  // namespace

// Capacities and layout of the scratch, from the hints of slhip_settle_params (the same function sizes and carves).
struct SettleDims {
    int nb_cap, lh_cap, p_cap, c_cap, bp_cap;   // bp_cap: body pairs (0 = the default rule of wide_g_cap)
    unsigned cache_stride;   // pair cache entries per scene
    int cache_hashed;
};
static SettleDims settle_dims(const slhip_settle_params* params)
{
    SettleDims D;
    D.nb_cap = params && params->max_bodies_per_scene ? (int)params->max_bodies_per_scene : SLHIP_MAX_BODIES;
    D.lh_cap = params && params->max_hulls_per_scene ? (int)params->max_hulls_per_scene : 1024;
    D.p_cap = params && params->max_hull_pairs_per_scene ? (int)params->max_hull_pairs_per_scene : SLHIP_DEFAULT_HULL_PAIRS;
    D.c_cap = params && params->max_contacts_per_scene ? (int)params->max_contacts_per_scene : SLHIP_DEFAULT_CONTACTS;
    if (D.p_cap > 65535) D.p_cap = 65535;
    if (D.c_cap > 65535) D.c_cap = 65535;
    D.bp_cap = params && params->max_body_pairs_per_scene ? (int)(params->max_body_pairs_per_scene > 65000u ? 65000u : params->max_body_pairs_per_scene) : 0;
    // pair cache: dense [hulls]^2 when the hint says the scenes are small, else hashed (8 slots per list entry, a power of two)
    const unsigned h = params ? params->max_hulls_per_scene : 0u;
    if (h != 0u && h <= SLHIP_PAIR_CACHE_DENSE_HULLS) { D.cache_stride = h * h; D.cache_hashed = 0; }
    else {
        unsigned t = 1024u;                 // per table: four slots per list entry, a power of two; two tables (step parity)
        while (t < 4u * (unsigned)D.p_cap) t <<= 1;
        D.cache_stride = 2u * t; D.cache_hashed = 1;
    }
    return D;
}

// [n_scenes x ProfScratch][n_scenes x nb_cap x DriveAcc][n_scenes x pair cache][state of the lockstep pipeline (slhip_settle_wide.inc)]
static uint64_t settle_fixed_bytes(uint32_t n_scenes, const SettleDims& D)
{
    const uint64_t b = (uint64_t)n_scenes * (sizeof(ProfScratch) + (uint64_t)D.nb_cap * sizeof(DriveAcc));
    return (b + 255u) & ~(uint64_t)255u;
}
static uint64_t settle_cache_bytes(uint32_t n_scenes, const SettleDims& D)
{
    const uint64_t b = (uint64_t)n_scenes * D.cache_stride * sizeof(int4);
    return (b + 255u) & ~(uint64_t)255u;
}
static uint64_t settle_scratch_bytes(uint32_t n_scenes, const slhip_settle_params* params)
{
    const SettleDims D = settle_dims(params);
    return settle_fixed_bytes(n_scenes, D) + settle_cache_bytes(n_scenes, D) + wide_bytes(n_scenes, D.nb_cap, D.lh_cap, D.p_cap, D.c_cap, D.bp_cap) + 256;
}

// Optional live timing of the lockstep kernels (bench.py's roofline leg): HIP events on the launch's stream around every
// kernel of every 8th step; slhip_settle_timings() synchronises them and returns the average launch duration per kernel.
struct SettleTiming {
    bool on = false;
    struct Rec { int kernel; hipEvent_t e0, e1; uint32_t step; };
    std::vector<Rec> pending;          // kernel k of a timed step ran between e0 and e1 (neighbours share an event)
    std::vector<hipEvent_t> events;    // every event once, destroyed after the read-out
};
static SettleTiming g_settle_timing;

extern "C" int slhip_settle_timing_enable(int on)
{
    g_settle_timing.on = on != 0;
    return 0;
}

static uint32_t g_settle_timing_every = 8;

// reads the pending records out: averages per kernel and, when by_step_ms is given, every timed step's five durations
static int settle_timings_read(float avg_ms_out[5], uint32_t launches_out[5], float* by_step_ms, uint32_t* steps_out, uint32_t cap,
                               uint32_t* n_out)
{
    double acc[5] = {0, 0, 0, 0, 0};
    uint32_t n[5] = {0, 0, 0, 0, 0};
    bool failed = false;
    uint32_t rows = 0;
    bool row_open = false;             // the current step has a row of its own (false once `cap` rows are written: later steps are averaged only)
    for (auto& r : g_settle_timing.pending) {
        float ms = 0.0f;
        if (failed || hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) { failed = true; continue; }
        acc[r.kernel] += ms;
        ++n[r.kernel];
        if (by_step_ms) {
            if (r.kernel == 0) {
                row_open = rows < cap;
                if (row_open) { if (steps_out) steps_out[rows] = r.step; ++rows; }
            }
            if (row_open) by_step_ms[5 * (size_t)(rows - 1) + r.kernel] = ms;
        }
    }
    for (hipEvent_t e : g_settle_timing.events) (void)hipEventDestroy(e);   // on the error path as well
    g_settle_timing.events.clear();
    g_settle_timing.pending.clear();
    if (n_out) *n_out = rows;
    if (failed) {
        slhip::set_error("slhip_settle_timings: event readback failed");
        return -1;
    }
    for (int k = 0; k < 5; ++k) {
        if (avg_ms_out) avg_ms_out[k] = n[k] ? (float)(acc[k] / n[k]) : 0.0f;
        if (launches_out) launches_out[k] = n[k];
    }
    return 0;
}

extern "C" int slhip_settle_timings(float avg_ms_out[5], uint32_t launches_out[5])
{
    return settle_timings_read(avg_ms_out, launches_out, nullptr, nullptr, 0, nullptr);
}

#ifdef SLHIP_ROW_PROFILE
// developer build only: reads and clears the row profile (tools/row_profile.py)
extern "C" int slhip_settle_row_profile(unsigned long long out[8])
{
    SLHIP_CHECK(hipDeviceSynchronize());
    SLHIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_row_prof), 8 * sizeof(unsigned long long)));
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SLHIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_row_prof), zero, sizeof(zero)));
    return 0;
}
#endif

static int g_last_solve_lds = 0;   // LDS bytes per solver wave of the last lockstep slhip_settle call (measurement read-out)
extern "C" int slhip_settle_solver_wave_lds(void) { return g_last_solve_lds; }

extern "C" int slhip_settle_timing_every(uint32_t every)
{
    g_settle_timing_every = every ? every : 1u;
    return 0;
}

extern "C" int slhip_settle_timings_by_step(float* ms_out, uint32_t* steps_out, uint32_t capacity, uint32_t* n_out)
{
    if (!ms_out || !n_out) {
        slhip::set_error("slhip_settle_timings_by_step: null output");
        return -1;
    }
    return settle_timings_read(nullptr, nullptr, ms_out, steps_out, capacity, n_out);
}

extern "C" int slhip_settle_scratch_bytes(uint32_t n_scenes, const slhip_settle_params* params, uint64_t* bytes_out)
{
    if (!bytes_out) {
        slhip::set_error("slhip_settle_scratch_bytes: null output");
        return -1;
    }
    *bytes_out = settle_scratch_bytes(n_scenes, params);
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
    if (scratch_bytes < settle_scratch_bytes(n_scenes, params)) {
        slhip::set_error("slhip_settle: scratch too small");
        return -1;
    }
    const SettleDims D = settle_dims(params);
    const int nb_cap = D.nb_cap;
    if (nb_cap > SLHIP_MAX_BODIES) {
        slhip::set_error("slhip_settle: at most %d bodies per scene", SLHIP_MAX_BODIES);
        return -1;
    }
    {
        if (n_scenes > 65535u) {
            slhip::set_error("slhip_settle: at most 65535 scenes per launch");
            return -1;
        }
        ProfScratch* prof_w = reinterpret_cast<ProfScratch*>(d_scratch);
        DriveAcc* drive_w = reinterpret_cast<DriveAcc*>(prof_w + n_scenes);
        char* base = reinterpret_cast<char*>(d_scratch) + settle_fixed_bytes(n_scenes, D);
        PairCache pc;
        pc.base = reinterpret_cast<int4*>(base);
        pc.stride = D.cache_stride;
        pc.hashed = D.cache_hashed;
        WideBufs W = wide_carve(base + settle_cache_bytes(n_scenes, D), n_scenes, nb_cap, D.lh_cap, D.p_cap, D.c_cap, D.bp_cap);
        if (((uint64_t)n_scenes << W.pair_bits) > (1ull << 32)) {
            slhip::set_error("slhip_settle: n_scenes x max_hull_pairs_per_scene exceeds the 32-bit work list entries");
            return -1;
        }
        const BeginLds BL = begin_layout(nb_cap, D.lh_cap);
        const FinishLds FL = finish_layout(nb_cap, W.g_cap);
        // LDS of a solver wave: 20 KB (eight waves per CU), more when a scene of the batch's largest shape needs it.  Its quarters and
        // halves are the wave classes of k_w_finish (SLHIP_SOLVE_SPW = 1 / 2: at most one / two scenes per wave).  The sweeps are
        // latency chains: what counts is how many solver waves a CU holds.  Per launch over 32768 C2 scenes alone, LDS per wave
        // 48 / 40 / 32 / 24 / 20 / 16 KB (3 / 4 / 5 / 6 / 8 / 10 waves per CU): 2.41 / 1.91 / 2.10 / 1.59 / 1.38 / 1.42 ms -- the settle 1.88 s at
        // 32 KB (rounds 4 and 5 until now), 1.57 s at 20; beside the render 9 920 - 9 980 against 10 060 scenes/s.  At 20 KB 8 % of the
        // scenes sweep part of their contacts from global memory in some step (same bits), at 32 KB 0.05 %.
        int solve_lds = 20 * 1024;
        if (solve_min_lds(nb_cap, W.g_cap) > solve_lds) solve_lds = (solve_min_lds(nb_cap, W.g_cap) + 1023) & ~1023;
        if (const char* e = getenv("SLHIP_SOLVE_LDS_KB")) { const int kb = atoi(e) * 1024; if (kb >= solve_min_lds(nb_cap, W.g_cap) && kb <= 160 * 1024) solve_lds = kb; }
        int max_class = 2;
        if (const char* e = getenv("SLHIP_SOLVE_SPW")) { const int v = atoi(e); max_class = v == 1 ? 0 : v == 2 ? 1 : 2; }
        W.solve_lds = solve_lds; W.solve_max_class = max_class;
        g_last_solve_lds = solve_lds;
        if (BL.total > 160 * 1024 || FL.total > 160 * 1024 || solve_lds > 160 * 1024) {
            slhip::set_error("slhip_settle: scenes of %d bodies / %d hulls do not fit the kernels' LDS", nb_cap, D.lh_cap);
            return -1;
        }
        SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_begin), hipFuncAttributeMaxDynamicSharedMemorySize, BL.total));
        SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_finish), hipFuncAttributeMaxDynamicSharedMemorySize, FL.total));
        SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_solve), hipFuncAttributeMaxDynamicSharedMemorySize, solve_lds));
        // a resumed call finds everything the prologue would set up -- and the contact state it would clear -- in the scratch
        if (params->resume == 0u) k_w_prologue<<<n_scenes, 64, 0, stream>>>(d_scenes, d_bodies, d_hulls, W, prof_w, pc);
        // the compacted narrowphase passes walk their work lists with a grid stride: enough waves for a step's typical list
        // (a scene has ~40 candidate pairs), never more than the worst case needs
        const unsigned list_stride = n_scenes * (unsigned)D.p_cap;
        unsigned work_grid = n_scenes < 16u ? n_scenes * 8u : n_scenes;
        if (const char* e = getenv("SLHIP_WORK_GRID_DIV")) { const unsigned d = (unsigned)atoi(e); if (d > 1u && work_grid / d >= 64u) work_grid /= d; }
        // small batches: one launch, a wave per scene through every step (k_w_persistent); large ones: six launches per step over
        // the whole batch.  The results are the same bits; the choice is about time only.
        bool persistent = n_scenes <= kPersistentMaxScenes;
        if (const char* e = getenv("SLHIP_SETTLE_PERSISTENT")) persistent = atoi(e) != 0;
        if (persistent) {
            // (one wave per SIMD by its registers: a wave may as well have a quarter of the CU's LDS for its scene's contacts)
            WideBufs Wp = W;
            Wp.solve_lds = max(solve_lds, 32 * 1024);      // (+ 6 KB of static LDS: four waves per CU still fit)
            const int lds = max(max(BL.total, FL.total), Wp.solve_lds);
            SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_persistent), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            unsigned grid = n_scenes < 4096u ? n_scenes : 4096u;
            if (const char* e = getenv("SLHIP_SETTLE_PERSISTENT_GRID")) { const unsigned g = (unsigned)atoi(e); if (g > 0u && g < grid) grid = g; }
            k_w_persistent<<<grid, 64, lds, stream>>>(d_scenes, d_bodies, d_hulls, d_hull_verts, *params, Wp, BL, FL, pc, drive_w, list_stride, n_scenes);
            SLHIP_LAUNCH_CHECK();
            return 0;
        }
        // (probe: the frames from SLHIP_SETTLE_SWITCH_FRAME on in the persistent form -- both forms work on the same state)
        uint32_t lockstep_frames = params->frames;
        if (const char* e = getenv("SLHIP_SETTLE_SWITCH_FRAME")) { const int v = atoi(e); if (v >= 0 && (uint32_t)v < params->frames) lockstep_frames = (uint32_t)v; }
        uint32_t step = params->resume;
        for (uint32_t f = 0; f < lockstep_frames; ++f)
            for (uint32_t sub = 0; sub < params->substeps; ++sub, ++step) {
                // (a caller that never reads the timings must not grow the lists for ever: sampling stops at kMaxTimedEvents)
                constexpr size_t kMaxTimedEvents = 1u << 16;
                bool timed = g_settle_timing.on && step % g_settle_timing_every == 0u && g_settle_timing.events.size() < kMaxTimedEvents;
                hipEvent_t ev[6];
                if (timed)
                    for (int k = 0; k < 6; ++k)
                        if (hipEventCreate(&ev[k]) != hipSuccess) {
                            for (int q = 0; q < k; ++q) (void)hipEventDestroy(ev[q]);
                            timed = false;
                            break;
                        }
                if (timed) (void)hipEventRecord(ev[0], stream);
                k_w_begin<<<n_scenes, 64, BL.total, stream>>>(d_scenes, d_bodies, d_hulls, d_hull_verts, *params, W, BL, drive_w, step + 1u);
                if (timed) (void)hipEventRecord(ev[1], stream);
                k_w_gjk_first<<<work_grid, 64, 0, stream>>>(d_hull_verts, *params, W, pc, list_stride, step + 1u, n_scenes);
                k_w_gjk_rest<<<work_grid, 64, 0, stream>>>(d_hull_verts, *params, W, pc, list_stride, step + 1u, n_scenes);
                if (timed) (void)hipEventRecord(ev[2], stream);
                k_w_manifold<<<work_grid, 64, 0, stream>>>(d_hull_verts, *params, W, list_stride);
                if (timed) (void)hipEventRecord(ev[3], stream);
                k_w_finish<<<n_scenes, 64, FL.total, stream>>>(d_scenes, d_bodies, *params, W, FL, pc, step + 1u);
                if (timed) (void)hipEventRecord(ev[4], stream);
                k_w_solve<<<n_scenes, 64, solve_lds, stream>>>(d_scenes, d_bodies, *params, W, drive_w,
                                                               sub + 1 == params->substeps ? 1 : 0, n_scenes);
                if (timed) {
                    (void)hipEventRecord(ev[5], stream);
                    for (int k = 0; k < 5; ++k) g_settle_timing.pending.push_back({k, ev[k], ev[k + 1], step});
                    for (int k = 0; k < 6; ++k) g_settle_timing.events.push_back(ev[k]);
                }
            }
        if (lockstep_frames < params->frames) {
            slhip_settle_params rest = *params;
            rest.resume = step;
            rest.frames = params->frames - lockstep_frames;
            const int lds = max(max(BL.total, FL.total), solve_lds);
            SLHIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_persistent), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            unsigned grid = n_scenes < 4096u ? n_scenes : 4096u;
            if (const char* e = getenv("SLHIP_SETTLE_PERSISTENT_GRID")) { const unsigned g = (unsigned)atoi(e); if (g > 0u && g < grid) grid = g; }
            k_w_persistent<<<grid, 64, lds, stream>>>(d_scenes, d_bodies, d_hulls, d_hull_verts, rest, W, BL, FL, pc, drive_w, list_stride, n_scenes);
        }
        SLHIP_LAUNCH_CHECK();
        return 0;
    }
}

// What the capacities cost since the last cold start on this scratch (include/slhip.h): spills beyond the solver's LDS-resident
// contacts (nothing lost), contacts / hull pairs DROPPED beyond the capacities the caller sized (the contract is zero), the
// scenes concerned, the most a step offered.  The reference's PhysX has no caps (scene.cpp:738-739).
extern "C" int slhip_settle_caps(const void* d_scratch, uint32_t n_scenes, const slhip_settle_params* params,
                                 uint64_t counts[10], void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_scratch || !params || !counts) {
        slhip::set_error("slhip_settle_caps: null argument");
        return -1;
    }
    for (int k = 0; k < 10; ++k) counts[k] = 0;
    if (n_scenes == 0) return 0;
    const SettleDims D = settle_dims(params);
    const char* base = reinterpret_cast<const char*>(d_scratch) + settle_fixed_bytes(n_scenes, D) + settle_cache_bytes(n_scenes, D);
    const WideBufs W = wide_carve(const_cast<char*>(base), n_scenes, D.nb_cap, D.lh_cap, D.p_cap, D.c_cap, D.bp_cap);
    std::vector<unsigned> h((size_t)n_scenes * kCapWords);
    SLHIP_CHECK(hipMemcpyAsync(h.data(), W.caps, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    SLHIP_CHECK(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < n_scenes; ++i) {
        const unsigned* c = &h[(size_t)kCapWords * i];
        counts[0] += c[kCapSpillSteps]; counts[1] += c[kCapContactDropSteps]; counts[2] += c[kCapPairDropSteps];
        if (c[kCapContactDropSteps] || c[kCapPairDropSteps] || c[kCapGroupDropSteps]) ++counts[3];
        if (c[kCapSpillSteps]) ++counts[4];
        if (c[kCapMaxContacts] > counts[5]) counts[5] = c[kCapMaxContacts];
        if (c[kCapMaxPairs] > counts[6]) counts[6] = c[kCapMaxPairs];
        counts[7] += c[kCapReducedSteps];
        counts[8] += c[kCapGroupDropSteps];
        counts[9] += c[kCapContactSum];
    }
    return 0;
}

extern "C" int slhip_settle_status(const void* d_scratch, uint32_t n_scenes, uint32_t* h_status, uint32_t* h_n_refused,
                                   void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_scratch || !h_n_refused) {
        slhip::set_error("slhip_settle_status: null argument");
        return -1;
    }
    *h_n_refused = 0;
    if (n_scenes == 0) return 0;
    std::vector<unsigned long long> st(n_scenes);
    SLHIP_CHECK(hipMemcpy2DAsync(st.data(), sizeof(unsigned long long), d_scratch, sizeof(ProfScratch),
                                 sizeof(unsigned long long), n_scenes, hipMemcpyDeviceToHost, stream));
    SLHIP_CHECK(hipStreamSynchronize(stream));
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n_scenes; ++i) {
        if (h_status) h_status[i] = (uint32_t)st[i];
        if (st[i] != 0) ++bad;
    }
    *h_n_refused = bad;
    if (bad) {
        slhip::set_error("slhip_settle: %u of %u scenes exceeded the sizing hints of slhip_settle_params and were left "
                         "untouched", bad, n_scenes);
        return -2;
    }
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
