/*
 * settle_ref.c -- ORACLE (test infrastructure, not product code).
 *
 * Scalar CPU restatement of the rigid-body stepping the reference delegates to PhysX 4.1
 * (NVIDIAGameWorks/PhysX @ 4050bbfdc2699dfab7edbf0393df8ff96bbe06c5, not vendored in
 * /root/reference, see contrib/physx/CMakeLists.txt:26-29) as configured by the reference's
 * call sites:
 *   Scene::simulateTableTopScene   /root/reference/src/scene.cpp:612-759  (settle loop, redrop)
 *   Scene::Scene                   src/scene.cpp:134-173                  (gravity, flags)
 *   Object::loadPhysics            src/object.cpp:142-213                 (shapes, rest offset,
 *                                                                          4+4 iterations)
 *   Context                        src/context.cpp:236-252                (tolerances, material)
 *   SimulationCallback::onContact  src/scene.cpp:73-116                   (min separation)
 *
 * PARITY STATUS: "parity unpinned".  PhysX's arithmetic is third-party code that is absent
 * from the tree and the reference's tests pin only a free-flight step (tests/test_python.py:
 * 111-130) -- checked in tests/test_oracle_settle.py together with the KATs k1..k7 of
 * SURVEY.md 8c.  The ALGORITHM below (documented [ext] behaviour of PhysX: speculative
 * contacts inside contactOffset, rest offset, PGS with 4 biased position + 4 unbiased velocity
 * iterations, Coulomb friction with static/dynamic coefficients, restitution above the bounce
 * threshold, semi-implicit Euler, sleeping) is this repository's contract; the HIP kernel must
 * reproduce it BIT-FOR-BIT, which is possible because
 *   - only + - * / sqrt and explicit fmaf are used (no libm transcendentals),
 *   - every reduction has a fixed order (first-maximum argmax, ordered lists),
 *   - the Gauss-Seidel order is defined by a greedy colouring of the body-pair groups and is
 *     independent of the number of lanes.
 */
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/slhip.h"

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } quat;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* a + b*s */
static inline v3 madd(v3 a, v3 b, float s) { return V(fmaf(b.x, s, a.x), fmaf(b.y, s, a.y), fmaf(b.z, s, a.z)); }

typedef struct { float m[9]; } m3; /* row-major */

static inline v3 m3_mul(const m3* M, v3 v)
{
    return V(fmaf(M->m[2], v.z, fmaf(M->m[1], v.y, M->m[0] * v.x)),
             fmaf(M->m[5], v.z, fmaf(M->m[4], v.y, M->m[3] * v.x)),
             fmaf(M->m[8], v.z, fmaf(M->m[7], v.y, M->m[6] * v.x)));
}
static inline v3 m3_tmul(const m3* M, v3 v) /* M^T v */
{
    return V(fmaf(M->m[6], v.z, fmaf(M->m[3], v.y, M->m[0] * v.x)),
             fmaf(M->m[7], v.z, fmaf(M->m[4], v.y, M->m[1] * v.x)),
             fmaf(M->m[8], v.z, fmaf(M->m[5], v.y, M->m[2] * v.x)));
}

static inline quat quat_normalize(quat q)
{
    float n = sqrtf(fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x))));
    quat r = {q.x / n, q.y / n, q.z / n, q.w / n};
    return r;
}

static inline void quat_to_m3(quat q, m3* R)
{
    float x = q.x, y = q.y, z = q.z, w = q.w;
    R->m[0] = 1.0f - 2.0f * (y * y + z * z); R->m[1] = 2.0f * (x * y - z * w); R->m[2] = 2.0f * (x * z + y * w);
    R->m[3] = 2.0f * (x * y + z * w); R->m[4] = 1.0f - 2.0f * (x * x + z * z); R->m[5] = 2.0f * (y * z - x * w);
    R->m[6] = 2.0f * (x * z - y * w); R->m[7] = 2.0f * (y * z + x * w); R->m[8] = 1.0f - 2.0f * (x * x + y * y);
}

static inline quat m3_to_quat(const m3* R)
{
    const float* m = R->m;
    float t = m[0] + m[4] + m[8];
    quat q;
    if (t > 0.0f) {
        float s = sqrtf(t + 1.0f) * 2.0f;
        q.w = 0.25f * s; q.x = (m[7] - m[5]) / s; q.y = (m[2] - m[6]) / s; q.z = (m[3] - m[1]) / s;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        float s = sqrtf(1.0f + m[0] - m[4] - m[8]) * 2.0f;
        q.w = (m[7] - m[5]) / s; q.x = 0.25f * s; q.y = (m[1] + m[3]) / s; q.z = (m[2] + m[6]) / s;
    } else if (m[4] > m[8]) {
        float s = sqrtf(1.0f + m[4] - m[0] - m[8]) * 2.0f;
        q.w = (m[2] - m[6]) / s; q.x = (m[1] + m[3]) / s; q.y = 0.25f * s; q.z = (m[5] + m[7]) / s;
    } else {
        float s = sqrtf(1.0f + m[8] - m[0] - m[4]) * 2.0f;
        q.w = (m[3] - m[1]) / s; q.x = (m[2] + m[6]) / s; q.y = (m[5] + m[7]) / s; q.z = 0.25f * s;
    }
    return quat_normalize(q);
}

static inline quat quat_mul(quat a, quat b)
{
    quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* working state                                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    v3 x;          /* centre of mass, world */
    quat q;
    m3 R;
    v3 t;          /* object-frame origin in world: x - R com */
    v3 v, w;
    m3 Iinv_w;     /* world inverse inertia */
    float inv_mass;
    int dynamic;   /* moves this step (not static, not asleep) */
    float dl[3], da[3]; /* accumulated drive impulses (linear along the joint axes, angular) */
} wbody;

typedef struct {
    int a, b;      /* body indices within the scene; b = -1: the plane */
    v3 ra, rb;     /* contact arms from the COMs */
    v3 n, t1, t2;  /* n points from B to A */
    float sep;     /* signed distance, before subtracting the rest offset */
    float rest;
    float kn, kt1, kt2;   /* inverse effective masses' reciprocals (i.e. effective masses) */
    float ln, lt1, lt2;   /* accumulated impulses */
    float vn0;            /* normal velocity before the solve (restitution) */
    float mu_s, mu_d, e;
    int valid;
    int culled;           /* offered by its manifold but left out of the solver by pair_contact_budget (valid = 0) */
} contact;

#define MAX_CONTACTS_PER_HP 4   /* points of a manifold */
#define PLANE_SLOTS 4
#define PATCH_STRIDE 5          /* slots of a patch in the step's contact array: its points, then its centre row (patch_centre) */
#define PATCH_ORDER(k) ((k) == 0 ? MAX_CONTACTS_PER_HP : (k) - 1)   /* ... in the order of the solver's list: the centre row first */

/* Persistent contact manifold of one hull pair (PhysX keeps one per shape pair, PCM [ext]): the contact points in the two
   bodies' object frames, the normal in B's frame, and the impulses the solver ended the step with -- the next step refreshes
   the points in the new poses, adds the new closest-point pair and starts its sweeps from those impulses. */
typedef struct {
    int stamp;             /* step that wrote it (1-based); valid for the step after */
    int count;
    v3 nb;                 /* the contact normal the manifold was BUILT with, in B's object frame */
    v3 la[4], lb[4];       /* contact points in A's / B's object frame */
    float ln[4];           /* accumulated normal impulses at the end of the step */
    float lc;              /* ... and that of the patch's centre row */
} pmanifold;

/* ... and of one body against the table: the contact points are hull vertices, matched by their number */
typedef struct {
    int stamp, count;
    int id[4];
    float ln[4];
    float lc;
} pplane;

typedef struct {
    /* capacities of the per-step lists (slhip_settle_params.max_hull_pairs_per_scene / max_contacts_per_scene) and bodies */
    int P, C, NB;
    int G;                               /* groups: NB table groups + min(body pairs, P, max_body_pairs_per_scene or 12 NB + 64) (the kernels' wide_g_cap) */
    /* hull pair list */
    int n_hp;
    int *hp_ba, *hp_bb;                  /* [P] bodies */
    int *hp_ha, *hp_hb;                  /* [P] global hull ids */
    /* constraint groups: contiguous contact ranges sharing a body pair */
    int n_groups;
    int *g_begin, *g_end, *g_a, *g_b, *g_color;   /* [P + NB] */
    int n_colors;
    contact* c;                          /* [(P + NB) * PATCH_STRIDE]: five slots per hull pair (four points + the centre row), then five per body (table) */
    wbody* wb;                           /* [NB] */
    /* Pair cache (temporal coherence, like PhysX's cached separating axis): the last converged or
       separating simplex of every hull pair of the scene, [n_hulls][n_hulls] by scene-local hull
       ordinals, cleared when a cold settle call starts. */
    struct gjk_seed_s* cache;
    pmanifold* pm;                       /* [n_hulls][n_hulls], beside the cache */
    pmanifold** hp_pm;                   /* [P] */
    pplane* pp;                          /* [NB] */
    int step;                            /* 1-based step number since the cold start */
    int hp_overflow;
    unsigned cap_hits[7];                /* steps of this scene that dropped contacts beyond C / hull pairs beyond P; the most contacts /
                                            hull pairs a step offered; steps in which pair_contact_budget reduced a body pair; steps that
                                            dropped body pairs beyond the group list; contacts the solver took, summed over the steps */
    int n_hulls;
    int* body_lh;                        /* [NB + 1] first hull ordinal of every body */
    int *order, *size;                   /* [P + NB] colouring scratch */
    uint64_t* used;                      /* [NB] */
    v3 *mv, *mw;                         /* [NB] the velocities that moved the bodies this step (after the position iterations) */
    int *st_n, *st_touch;                /* [NB] stabilisation: touching body pairs of a body, its island rests on something static */
} scene_ws;

static void ws_free(scene_ws* ws)
{
    if (!ws) return;
    free(ws->hp_ba); free(ws->hp_bb); free(ws->hp_ha); free(ws->hp_hb);
    free(ws->g_begin); free(ws->g_end); free(ws->g_a); free(ws->g_b); free(ws->g_color);
    free(ws->c); free(ws->wb); free(ws->hp_pm); free(ws->pp); free(ws->body_lh); free(ws->order); free(ws->size); free(ws->used);
    free(ws->mv); free(ws->mw); free(ws->st_n); free(ws->st_touch);
    free(ws);
}

static scene_ws* ws_alloc(int P, int C, int NB, int BP)
{
    scene_ws* ws = (scene_ws*)calloc(1, sizeof(scene_ws));
    if (!ws) return NULL;
    ws->P = P; ws->C = C; ws->NB = NB;
    {
        long long pairs = (long long)NB * (NB - 1) / 2;
        if (pairs > P) pairs = P;
        const long long lim = BP > 0 ? (long long)BP : 12ll * NB + 64;
        if (pairs > lim) pairs = lim;
        ws->G = NB + (int)pairs;
    }
    const size_t G = (size_t)P + NB;
    ws->hp_ba = (int*)malloc(sizeof(int) * P); ws->hp_bb = (int*)malloc(sizeof(int) * P);
    ws->hp_ha = (int*)malloc(sizeof(int) * P); ws->hp_hb = (int*)malloc(sizeof(int) * P);
    ws->g_begin = (int*)malloc(sizeof(int) * G); ws->g_end = (int*)malloc(sizeof(int) * G);
    ws->g_a = (int*)malloc(sizeof(int) * G); ws->g_b = (int*)malloc(sizeof(int) * G); ws->g_color = (int*)malloc(sizeof(int) * G);
    ws->c = (contact*)malloc(sizeof(contact) * G * PATCH_STRIDE);
    ws->wb = (wbody*)malloc(sizeof(wbody) * (NB ? NB : 1));
    ws->hp_pm = (pmanifold**)malloc(sizeof(pmanifold*) * P);
    ws->pp = (pplane*)calloc(NB ? NB : 1, sizeof(pplane));
    ws->body_lh = (int*)malloc(sizeof(int) * (NB + 1));
    ws->order = (int*)malloc(sizeof(int) * G); ws->size = (int*)malloc(sizeof(int) * G);
    ws->used = (uint64_t*)malloc(sizeof(uint64_t) * (NB ? NB : 1));
    ws->mv = (v3*)malloc(sizeof(v3) * (NB ? NB : 1)); ws->mw = (v3*)malloc(sizeof(v3) * (NB ? NB : 1));
    ws->st_n = (int*)malloc(sizeof(int) * (NB ? NB : 1)); ws->st_touch = (int*)malloc(sizeof(int) * (NB ? NB : 1));
    if (!ws->hp_ba || !ws->hp_bb || !ws->hp_ha || !ws->hp_hb || !ws->g_begin || !ws->g_end || !ws->g_a || !ws->g_b || !ws->g_color ||
        !ws->c || !ws->wb || !ws->hp_pm || !ws->pp || !ws->body_lh || !ws->order || !ws->size || !ws->used ||
        !ws->mv || !ws->mw || !ws->st_n || !ws->st_touch) { ws_free(ws); return NULL; }
    return ws;
}

static inline int cap_pairs(const slhip_settle_params* prm)
{
    return prm->max_hull_pairs_per_scene ? (int)prm->max_hull_pairs_per_scene : SLHIP_DEFAULT_HULL_PAIRS;
}
static inline int cap_contacts(const slhip_settle_params* prm)
{
    return prm->max_contacts_per_scene ? (int)prm->max_contacts_per_scene : SLHIP_DEFAULT_CONTACTS;
}


/* ------------------------------------------------------------------------------------------ */
/* support mapping + GJK distance                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* verts; /* float4 each, object frame */
    int count;
    m3 R;
    v3 t;
} shape;

static inline v3 shape_vertex(const shape* s, int i)
{
    v3 p = V(s->verts[4 * i], s->verts[4 * i + 1], s->verts[4 * i + 2]);
    return add(m3_mul(&s->R, p), s->t);
}

/* support point along d; *index receives the vertex number (first maximum) */
static inline v3 support_i(const shape* s, v3 d, int* index)
{
    v3 dl = m3_tmul(&s->R, d);
    int best = 0;
    float bd = dot(V(s->verts[0], s->verts[1], s->verts[2]), dl);
    for (int i = 1; i < s->count; ++i) {
        float dd = dot(V(s->verts[4 * i], s->verts[4 * i + 1], s->verts[4 * i + 2]), dl);
        if (dd > bd) { bd = dd; best = i; }
    }
    *index = best;
    return shape_vertex(s, best);
}

static inline v3 support(const shape* s, v3 d)
{
    int unused;
    return support_i(s, d, &unused);
}

/* simplex vertex: w = a - b; idx = vertex of A | vertex of B << 16 (hulls have < 65536 vertices) */
typedef struct { v3 w, a, b; int idx; } sv;

/* The vertices of a converged simplex: what the pair cache hands the next step's run (temporal coherence). */
typedef struct gjk_seed_s { int n; int idx[3]; } gjk_seed;

/* closest point to the origin on segment / triangle; returns barycentric weights and the mask
   of vertices that stay in the simplex (Ericson, Real-Time Collision Detection 5.1.2/5.1.5) */
static int closest_segment(const sv* s, float* l)
{
    v3 a = s[0].w, b = s[1].w;
    v3 ab = sub(b, a);
    float t = dot(neg(a), ab);
    if (t <= 0.0f) { l[0] = 1.0f; l[1] = 0.0f; return 1; }
    float den = dot(ab, ab);
    if (t >= den) { l[0] = 0.0f; l[1] = 1.0f; return 2; }
    t = t / den;
    l[0] = 1.0f - t; l[1] = t;
    return 3;
}

static int closest_triangle(v3 a, v3 b, v3 c, float* l)
{
    v3 ab = sub(b, a), ac = sub(c, a), ap = neg(a);
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) { l[0] = 1; l[1] = 0; l[2] = 0; return 1; }
    v3 bp = neg(b);
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) { l[0] = 0; l[1] = 1; l[2] = 0; return 2; }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        float v = d1 / (d1 - d3);
        l[0] = 1.0f - v; l[1] = v; l[2] = 0; return 3;
    }
    v3 cp = neg(c);
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) { l[0] = 0; l[1] = 0; l[2] = 1; return 4; }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        float w = d2 / (d2 - d6);
        l[0] = 1.0f - w; l[1] = 0; l[2] = w; return 5;
    }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        l[0] = 0; l[1] = 1.0f - w; l[2] = w; return 6;
    }
    float denom = 1.0f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    l[0] = 1.0f - v - w; l[1] = v; l[2] = w;
    return 7;
}

static inline v3 comb3(v3 a, v3 b, v3 c, const float* l)
{
    return madd(madd(scale(a, l[0]), b, l[1]), c, l[2]);
}

/* origin outside the plane of (a,b,c) on the side opposite to d? */
static inline int outside_plane(v3 a, v3 b, v3 c, v3 d)
{
    v3 n = cross(sub(b, a), sub(c, a));
    float sp = dot(neg(a), n);
    float sd = dot(sub(d, a), n);
    return sp * sd < 0.0f || sd == 0.0f; /* degenerate tetra: treat as outside */
}

/* Reduces the simplex to the sub-simplex closest to the origin; returns the new size (0 = the
   origin is inside a tetrahedron, i.e. overlap) and writes the closest point to *v. */
static int reduce_simplex(sv* s, int n, float* lam, v3* v)
{
    if (n == 1) { lam[0] = 1.0f; *v = s[0].w; return 1; }
    if (n == 2) {
        float l[2];
        int mask = closest_segment(s, l);
        *v = madd(scale(s[0].w, l[0]), s[1].w, l[1]);
        if (mask == 1) { lam[0] = 1.0f; return 1; }
        if (mask == 2) { s[0] = s[1]; lam[0] = 1.0f; return 1; }
        lam[0] = l[0]; lam[1] = l[1];
        return 2;
    }
    if (n == 3) {
        float l[3];
        int mask = closest_triangle(s[0].w, s[1].w, s[2].w, l);
        *v = comb3(s[0].w, s[1].w, s[2].w, l);
        int k = 0;
        for (int i = 0; i < 3; ++i)
            if (mask & (1 << i)) { s[k] = s[i]; lam[k] = l[i]; ++k; }
        return k;
    }
    /* tetrahedron: test the four faces */
    static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    float best = 3.0e38f;
    int best_mask = 0, best_face = -1;
    float best_l[3] = {0, 0, 0};
    v3 best_v = V(0, 0, 0);
    for (int f = 0; f < 4; ++f) {
        v3 a = s[F[f][0]].w, b = s[F[f][1]].w, c = s[F[f][2]].w, d = s[F[f][3]].w;
        if (!outside_plane(a, b, c, d)) continue;
        float l[3];
        int mask = closest_triangle(a, b, c, l);
        v3 q = comb3(a, b, c, l);
        float dd = dot(q, q);
        if (dd < best) { best = dd; best_mask = mask; best_face = f; best_l[0] = l[0]; best_l[1] = l[1]; best_l[2] = l[2]; best_v = q; }
    }
    if (best_face < 0) return 0; /* origin inside */
    sv t[3] = {s[F[best_face][0]], s[F[best_face][1]], s[F[best_face][2]]};
    int k = 0;
    for (int i = 0; i < 3; ++i)
        if (best_mask & (1 << i)) { s[k] = t[i]; lam[k] = best_l[i]; ++k; }
    *v = best_v;
    return k;
}

/* GJK distance.  Returns 1 and (pa, pb, dist) if the shapes are separated by more than ~1e-6,
   0 if they touch/overlap, 2 if a separating plane farther than `margin` was found (early out:
   for any direction v, min over the Minkowski difference of v.x = v.w bounds the distance from
   below by v.w/|v|). */
#define GJK_MAX_ITER 32
static uint64_t* g_stats; /* statistics hook, see slref_settle_set_stats */
static int g_last_gjk_iters = 0; /* support evaluations of the last run (statistics only) */
static int gjk_distance_seeded(const shape* A, const shape* B, v3 init_dir, float margin, v3* pa, v3* pb, float* dist,
                               const gjk_seed* seed_in, gjk_seed* seed_out, int max_iter)
{
    sv s[4];
    float lam[4] = {1, 0, 0, 0};
    int n = 0;
    v3 v = init_dir;
    if (dot(v, v) < 1e-12f) v = V(1, 0, 0);
    float vv = dot(v, v);
    const float m2 = margin * margin;
    if (seed_out) seed_out->n = 0;
    if (seed_in && seed_in->n > 0) {
        for (int k = 0; k < seed_in->n; ++k) {
            s[k].idx = seed_in->idx[k];
            s[k].a = shape_vertex(A, seed_in->idx[k] & 0xffff);
            s[k].b = shape_vertex(B, (int)((unsigned)seed_in->idx[k] >> 16));
            s[k].w = sub(s[k].a, s[k].b);
        }
        n = reduce_simplex(s, seed_in->n, lam, &v); /* <= 3 vertices: never 0 */
        vv = dot(v, v);
        if (vv < 1e-12f) return 0;
    }
    g_last_gjk_iters = 0;
    for (int it = 0; it < max_iter; ++it) {
        sv w;
        int ia, ib;
        g_last_gjk_iters = it + 1;
        w.a = support_i(A, neg(v), &ia);
        w.b = support_i(B, v, &ib);
        w.idx = ia | (ib << 16);
        w.w = sub(w.a, w.b);
        float vw = dot(v, w.w);
        if (vw > 0.0f && vw * vw > m2 * vv) {
            /* separated beyond the margin: the simplex that produced the separating direction is handed
               back too (the pair cache restarts from it; before the first reduction it is w alone) */
            if (seed_out) {
                if (n == 0) { seed_out->n = 1; seed_out->idx[0] = w.idx; }
                else { seed_out->n = n; for (int i = 0; i < n; ++i) seed_out->idx[i] = s[i].idx; }
            }
            return 2;
        }
        if (n > 0) {
            /* no progress towards the origin: v is the closest point */
            if (vv - vw <= 1e-6f * vv) break;
            int dup = 0;
            for (int i = 0; i < n; ++i)
                if (s[i].w.x == w.w.x && s[i].w.y == w.w.y && s[i].w.z == w.w.z) dup = 1;
            if (dup) break;
        }
        s[n++] = w;
        v3 nv;
        int nn = reduce_simplex(s, n, lam, &nv);
        if (nn == 0) return 0;
        float nvv = dot(nv, nv);
        if (n > 1 && nvv >= vv && it > 0) { /* numerical stall: keep the previous result */
            n = nn; v = nv; vv = nvv; break;
        }
        n = nn; v = nv; vv = nvv;
        if (vv < 1e-12f) return 0;
    }
    if (n == 0) return 0;
    v3 a = V(0, 0, 0), b = V(0, 0, 0);
    for (int i = 0; i < n; ++i) { a = madd(a, s[i].a, lam[i]); b = madd(b, s[i].b, lam[i]); }
    *pa = a; *pb = b;
    if (seed_out) {
        seed_out->n = n;   /* 1..3 after reduce_simplex */
        for (int i = 0; i < n; ++i) seed_out->idx[i] = s[i].idx;
    }
    float d = sqrtf(vv);
    *dist = d;
    return d > 1e-6f ? 1 : 0;
}

static int gjk_distance(const shape* A, const shape* B, v3 init_dir, float margin, v3* pa, v3* pb, float* dist)
{
    return gjk_distance_seeded(A, B, init_dir, margin, pa, pb, dist, NULL, NULL, GJK_MAX_ITER);
}

/* tangent basis (deterministic) */
static inline void tangents(v3 n, v3* t1, v3* t2)
{
    v3 a;
    if (fabsf(n.x) > 0.57735f) a = V(n.y, -n.x, 0.0f);
    else a = V(0.0f, n.z, -n.y);
    float l = sqrtf(dot(a, a));
    *t1 = scale(a, 1.0f / l);
    *t2 = cross(n, *t1);
}

/* crude penetration fallback when GJK reports overlap: least overlap among 7 fixed axes */
static void overlap_fallback(const shape* A, const shape* B, v3 ca, v3 cb, v3* n, float* sep, v3* pa, v3* pb)
{
    v3 axes[7];
    v3 c = sub(ca, cb);
    float cl = sqrtf(dot(c, c));
    axes[0] = cl > 1e-6f ? scale(c, 1.0f / cl) : V(0, 0, 1);
    axes[1] = V(1, 0, 0); axes[2] = V(-1, 0, 0); axes[3] = V(0, 1, 0);
    axes[4] = V(0, -1, 0); axes[5] = V(0, 0, 1); axes[6] = V(0, 0, -1);
    float best = -3.0e38f;
    for (int i = 0; i < 7; ++i) {
        v3 a = support(A, neg(axes[i])); /* lowest point of A along the axis */
        v3 b = support(B, axes[i]);      /* highest point of B */
        float s = dot(sub(a, b), axes[i]); /* negative = overlap depth */
        if (s > best) { best = s; *n = axes[i]; *pa = a; *pb = b; }
    }
    *sep = best;
}

/* ------------------------------------------------------------------------------------------ */
/* penetration of two OVERLAPPING hulls: Minkowski portal refinement (Snethen, "XenoCollide", Game Programming Gems 7;       */
/* the published algorithm, fixed-size state like GJK's simplex).  PhysX answers the same question with GJK + EPA [ext].     */
/* M = A - B contains the origin.  From an interior point v0 (difference of the hulls' sphere centres) the ray through the    */
/* origin leaves M through a boundary triangle -- the portal (v1 v2 v3), refined by support points along its normal until it   */
/* lies within MPR_TOL of the boundary.  The point of the portal closest to the origin, pt = pa - pb, gives the contact:      */
/* normal n = -pt / |pt| (from B to A), separation -|pt|, witnesses pa on A and pb on B by the portal's barycentric weights.   */
/* Returns 1 on success, 0 when no portal is found (the hulls only touch): the caller falls back to overlap_fallback.         */
/* ------------------------------------------------------------------------------------------ */
#define MPR_TOL 1.0e-4f
#define MPR_MAX_DISCOVER 16
#define MPR_MAX_REFINE 24
static inline sv mpr_support(const shape* A, const shape* B, v3 d)
{
    sv w;
    int ia, ib;
    w.a = support_i(A, d, &ia);
    w.b = support_i(B, neg(d), &ib);
    w.idx = ia | (ib << 16);
    w.w = sub(w.a, w.b);
    return w;
}
static inline v3 normalized(v3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }

static int mpr_penetration(const shape* A, const shape* B, v3 ca, v3 cb, v3* n, float* sep, v3* pa, v3* pb)
{
    v3 v0 = sub(ca, cb);
    if (dot(v0, v0) < 1.0e-12f) v0 = V(1.0e-5f, 0.0f, 0.0f);
    v3 dir = normalized(neg(v0));
    sv v1 = mpr_support(A, B, dir);
    if (!(dot(v1.w, dir) > 0.0f)) return 0;
    dir = cross(v0, v1.w);
    if (dot(dir, dir) < 1.0e-20f) {
        /* the origin lies on the ray through v1: the boundary point IS v1 */
        float l = sqrtf(dot(v1.w, v1.w));
        if (l < 1.0e-9f) return 0;
        *n = scale(v1.w, -1.0f / l); *sep = -l; *pa = v1.a; *pb = v1.b;
        return 1;
    }
    dir = normalized(dir);
    sv v2 = mpr_support(A, B, dir);
    if (!(dot(v2.w, dir) > 0.0f)) return 0;
    dir = cross(sub(v1.w, v0), sub(v2.w, v0));
    if (dot(dir, dir) < 1.0e-24f) return 0;
    dir = normalized(dir);
    if (dot(dir, v0) > 0.0f) { sv t = v1; v1 = v2; v2 = t; dir = neg(dir); }
    sv v3_;
    int found = 0;
    for (int it = 0; it < MPR_MAX_DISCOVER; ++it) {
        v3_ = mpr_support(A, B, dir);
        if (!(dot(v3_.w, dir) > 0.0f)) return 0;
        int cont = 0;
        if (dot(cross(v1.w, v3_.w), v0) < 0.0f) { v2 = v3_; cont = 1; }
        else if (dot(cross(v3_.w, v2.w), v0) < 0.0f) { v1 = v3_; cont = 1; }
        if (!cont) { found = 1; break; }
        dir = cross(sub(v1.w, v0), sub(v2.w, v0));
        if (dot(dir, dir) < 1.0e-24f) return 0;
        dir = normalized(dir);
    }
    if (!found) return 0;
    for (int it = 0; it < MPR_MAX_REFINE; ++it) {
        dir = cross(sub(v2.w, v1.w), sub(v3_.w, v1.w));
        if (dot(dir, dir) < 1.0e-24f) break;
        dir = normalized(dir);
        sv v4 = mpr_support(A, B, dir);
        float d4 = dot(v4.w, dir);
        float m = d4 - dot(v1.w, dir);
        float m2 = d4 - dot(v2.w, dir), m3 = d4 - dot(v3_.w, dir);
        if (m2 < m) m = m2;
        if (m3 < m) m = m3;
        if (m <= MPR_TOL) break;
        v3 x = cross(v4.w, v0);
        if (dot(v1.w, x) > 0.0f) {
            if (dot(v2.w, x) > 0.0f) v1 = v4; else v3_ = v4;
        } else {
            if (dot(v3_.w, x) > 0.0f) v2 = v4; else v1 = v4;
        }
    }
    float l[3];
    closest_triangle(v1.w, v2.w, v3_.w, l);
    v3 pt = comb3(v1.w, v2.w, v3_.w, l);
    float d = sqrtf(dot(pt, pt));
    *pa = comb3(v1.a, v2.a, v3_.a, l);
    *pb = comb3(v1.b, v2.b, v3_.b, l);
    if (d > 1.0e-7f) *n = scale(pt, -1.0f / d);
    else {
        v3 pn = cross(sub(v2.w, v1.w), sub(v3_.w, v1.w));
        if (dot(pn, pn) < 1.0e-24f) return 0;
        *n = neg(normalized(pn));
    }
    *sep = -d;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* contact generation                                                                          */
/* ------------------------------------------------------------------------------------------ */
static void make_shape(const wbody* wb, const slhip_hull* h, const float* hull_verts, shape* s)
{
    s->verts = hull_verts + 4 * (size_t)h->vtx_begin;
    s->count = (int)h->vtx_count;
    s->R = wb->R;
    s->t = wb->t;
}

/* keeps at most 4 of n candidate points (p on A, separation): the deepest, then the point
   farthest from it, then the two extremes of the signed area -- where every metre of extra
   separation costs DEPTH_WEIGHT metres of lateral reach, so that the true support points win
   over far-away points that merely lie inside the contact band.  First-index tie break. */
#define DEPTH_WEIGHT 30.0f
static int reduce4(int n, const v3* p, const float* sep, v3 nrm, int* keep)
{
    if (n <= 4) { for (int i = 0; i < n; ++i) keep[i] = i; return n; }
    int i0 = 0;
    for (int i = 1; i < n; ++i) if (sep[i] < sep[i0]) i0 = i;
    int i1 = -1; float best = -3.0e38f;
    for (int i = 0; i < n; ++i) {
        if (i == i0) continue;
        v3 d = sub(p[i], p[i0]);
        float pen = DEPTH_WEIGHT * (sep[i] - sep[i0]);
        float score = sqrtf(dot(d, d)) - pen;
        if (score > best) { best = score; i1 = i; }
    }
    int i2 = -1, i3 = -1; float mx = 0.0f, mn = 0.0f;
    v3 e = sub(p[i1], p[i0]);
    float el = sqrtf(dot(e, e));
    for (int i = 0; i < n; ++i) {
        if (i == i0 || i == i1) continue;
        float a = dot(cross(e, sub(p[i], p[i0])), nrm);
        float pen = DEPTH_WEIGHT * (sep[i] - sep[i0]) * el;
        if (a - pen > mx) { mx = a - pen; i2 = i; }
        if (a + pen < mn) { mn = a + pen; i3 = i; }
    }
    int k = 0;
    keep[k++] = i0; keep[k++] = i1;
    if (i2 >= 0) keep[k++] = i2;
    if (i3 >= 0) keep[k++] = i3;
    return k;
}

static void fill_contact(contact* c, int a, int b, const wbody* wa, const wbody* wbb, v3 pa, v3 pb, v3 n,
                         float sep, float rest, float mu_s, float mu_d, float e)
{
    memset(c, 0, sizeof(*c));
    c->a = a; c->b = b;
    c->ra = sub(pa, wa->x);
    c->rb = wbb ? sub(pb, wbb->x) : V(0, 0, 0);
    c->n = n;
    tangents(n, &c->t1, &c->t2);
    c->sep = sep; c->rest = rest;
    c->mu_s = mu_s; c->mu_d = mu_d; c->e = e;
    c->valid = 1;
}

/* Contact persistence (the part of PhysX's pipeline that lets piles come to rest [ext]: persistent manifolds, impulses carried
   from step to step).  Constants of this restatement: */
#define WARM_START 0.8f          /* share of the previous step's normal impulses the sweeps start from (friction rows start at 0) */
#define DRIFT_OFFSETS 2.0f       /* a persistent point breaks when its two witnesses drift apart laterally by more than this many
                                    contact offsets ... */
#define NORMAL_COS 0.9848f       /* ... and all of a manifold's points when the normal turned by more than 10 degrees */
#define THIN_FULL 4
#define THIN_HALF 8
#define PLANE_DEPTH_WEIGHT 5.0f  /* table contacts: a metre of extra separation costs this many metres of lateral reach (the far
                                    edge of a slightly tilted box has to stay in the manifold as a speculative contact, or the
                                    push-out of the near edge rocks the box for ever) */

/* ------------------------------------------------------------------------------------------ */
/* Face manifold of a hull pair in ONE step (PhysX: PCM full contact generation -- the incident polygon clipped against the     */
/* reference polygon, <= 4 points kept [ext]; scene.cpp:156-163 runs PhysX with its default PCM narrowphase).  Hulls are vertex */
/* clouds here, so the two polygons are the hulls' SUPPORT FEATURES along the contact normal n (from B to A): the vertices      */
/* within a band of the hull's extreme vertex along n, reduced to the <= 8 of them that are extreme in eight tangent            */
/* directions 45 degrees apart -- an ordered (counter-clockwise about n) convex polygon inscribed in the feature, exact for      */
/* triangles, rectangles and every polygon whose vertices each own one of the directions; a segment or a point for edge and      */
/* vertex features.  In tangent coordinates (u along t1, w along t2, h along n) the clipped polygon's corners are               */
/*   (i)   vertices of A's polygon inside B's      (height on B from B's polygon plane),                                        */
/*   (ii)  vertices of B's polygon inside A's      (height on A from A's polygon plane),                                        */
/*   (iii) crossings of an edge of A with an edge of B (heights along the two edges).                                           */
/* Every candidate has a fixed number -- 0..3 the points the pair's manifold kept from the previous step (with their impulses),  */
/* 4 the closest-point (deepest-point) pair, 5 + (i: 0..7, ii: 8..15, iii: 16 + 8 i + j) the clipped corners -- and the four     */
/* that stay are chosen by the rule of reduce4 (the deepest, the farthest from it, the two area extremes) as arg-min / arg-max   */
/* over (value, number), the lower number winning ties: independent of the order of evaluation (the kernel spreads the           */
/* candidates over sixteen lanes).                                                                                               */
/* ------------------------------------------------------------------------------------------ */
#define FM_BAND_OFFSETS 1.0f     /* thickness of a support feature in contact offsets ... */
#define FM_BAND_EXTENT 0.25f     /* ... and at most this share of the hull's extent along n (a thin piece of a shell must not offer both sides) */
#define FM_AREA_MIN 1.0e-6f      /* twice the area [m^2] below which a polygon counts as a segment or a point (nothing is "inside") */
#define FM_INSIDE_TOL 1.0e-6f    /* a vertex this far [m^2: edge length x distance] outside an edge still counts as inside */
#define FM_PARALLEL2 1.0e-6f     /* edges whose sin^2 of the enclosed angle is below this do not cross (the vertices cover aligned shapes) */
#define FM_DEEPER_OFFSETS 0.05f  /* the closest-point pair joins only when it is deeper than every other candidate by this many contact
                                    offsets -- and a manifold whose points are all shallower than that no longer holds the pair's deepest
                                    feature: it is rebuilt */
#define FM_DEEP_OFFSETS 2.5f     /* no clipped corners for hulls that overlap by more than this many contact offsets */
#define FM_CANDIDATES 85
typedef struct { float u[8], w[8], h[8]; float nu, nw, nh, cu, cw, ch; } fm_poly;
static uint64_t* g_fm_stats = NULL; /* [16]: (5.. rebuilt because: the normal turned, a point drifted, a point left the band, the closest points are deeper) manifolds built for new pairs, rebuilt, kept; points of the built ones; closest-point pairs that joined */
void slref_settle_set_fm_stats(uint64_t* h) { g_fm_stats = h; }

static void fm_feature(const shape* S, v3 n, v3 t1, v3 t2, int lowest, float band, fm_poly* P)
{
    const v3 nl = m3_tmul(&S->R, n), l1 = m3_tmul(&S->R, t1), l2 = m3_tmul(&S->R, t2);
    const float on = dot(S->t, n), o1 = dot(S->t, t1), o2 = dot(S->t, t2);
    float lo = 0.0f, hi = 0.0f;
    for (int i = 0; i < S->count; ++i) {
        const float h = dot(V(S->verts[4 * i], S->verts[4 * i + 1], S->verts[4 * i + 2]), nl) + on;
        if (i == 0 || h < lo) lo = h;
        if (i == 0 || h > hi) hi = h;
    }
    const float ext = lowest ? lo : hi;
    if (band > FM_BAND_EXTENT * (hi - lo)) band = FM_BAND_EXTENT * (hi - lo);
    float best[8];
    int have = 0;
    for (int i = 0; i < S->count; ++i) {
        const v3 v = V(S->verts[4 * i], S->verts[4 * i + 1], S->verts[4 * i + 2]);
        const float h = dot(v, nl) + on;
        if (lowest ? h > ext + band : h < ext - band) continue;
        const float u = dot(v, l1) + o1, w = dot(v, l2) + o2;
        float s[8];
        s[0] = u; s[1] = u + w; s[2] = w; s[3] = w - u;
        s[4] = -s[0]; s[5] = -s[1]; s[6] = -s[2]; s[7] = -s[3];
        for (int k = 0; k < 8; ++k)
            if (!have || s[k] > best[k]) { best[k] = s[k]; P->u[k] = u; P->w[k] = w; P->h[k] = h; }
        have = 1;
    }
    /* plane of the polygon (Newell, relative to slot 0) through the mean of its slots */
    float nu = 0.0f, nw = 0.0f, nh = 0.0f, su = 0.0f, sw = 0.0f, sh = 0.0f;
    for (int k = 0; k < 8; ++k) {
        const int k1 = (k + 1) & 7;
        const float au = P->u[k] - P->u[0], aw = P->w[k] - P->w[0], ah = P->h[k] - P->h[0];
        const float bu = P->u[k1] - P->u[0], bw = P->w[k1] - P->w[0], bh = P->h[k1] - P->h[0];
        nu = nu + (aw * bh - ah * bw);
        nw = nw + (ah * bu - au * bh);
        nh = nh + (au * bw - aw * bu);
        su = su + au; sw = sw + aw; sh = sh + ah;
    }
    P->nu = nu; P->nw = nw; P->nh = nh;
    P->cu = su * 0.125f; P->cw = sw * 0.125f; P->ch = sh * 0.125f;
}

static inline float fm_plane_h(const fm_poly* P, float u, float w)
{
    const float du = (u - P->u[0]) - P->cu, dw = (w - P->w[0]) - P->cw;
    return (P->h[0] + P->ch) - (P->nu * du + P->nw * dw) / P->nh;
}

static inline int fm_inside(const fm_poly* Q, float u, float w)
{
    if (!(Q->nh > FM_AREA_MIN)) return 0;
    for (int k = 0; k < 8; ++k) {
        const int k1 = (k + 1) & 7;
        const float eu = Q->u[k1] - Q->u[k], ew = Q->w[k1] - Q->w[k];
        const float cr = eu * (w - Q->w[k]) - ew * (u - Q->u[k]);
        if (cr < -FM_INSIDE_TOL) return 0;
    }
    return 1;
}

/* clipped corner c (0..79) -> (u, w, hA, hB); returns 0 when it does not exist */
static int fm_corner(const fm_poly* A, const fm_poly* B, int c, float* u, float* w, float* ha, float* hb)
{
    if (c < 8) {
        *u = A->u[c]; *w = A->w[c]; *ha = A->h[c];
        if (!fm_inside(B, *u, *w)) return 0;
        *hb = fm_plane_h(B, *u, *w);
        return 1;
    }
    if (c < 16) {
        *u = B->u[c - 8]; *w = B->w[c - 8]; *hb = B->h[c - 8];
        if (!fm_inside(A, *u, *w)) return 0;
        *ha = fm_plane_h(A, *u, *w);
        return 1;
    }
    const int i = (c - 16) >> 3, j = (c - 16) & 7, i1 = (i + 1) & 7, j1 = (j + 1) & 7;
    const float eau = A->u[i1] - A->u[i], eaw = A->w[i1] - A->w[i];
    const float ebu = B->u[j1] - B->u[j], ebw = B->w[j1] - B->w[j];
    const float den = eau * ebw - eaw * ebu;
    const float la = eau * eau + eaw * eaw, lb = ebu * ebu + ebw * ebw;
    if (!(den * den > FM_PARALLEL2 * (la * lb))) return 0;
    const float du = B->u[j] - A->u[i], dw = B->w[j] - A->w[i];
    const float s = (du * ebw - dw * ebu) / den, t = (du * eaw - dw * eau) / den;
    if (!(s >= 0.0f && s <= 1.0f && t >= 0.0f && t <= 1.0f)) return 0;
    *u = fmaf(s, eau, A->u[i]); *w = fmaf(s, eaw, A->w[i]);
    *ha = fmaf(s, A->h[i1] - A->h[i], A->h[i]);
    *hb = fmaf(t, B->h[j1] - B->h[j], B->h[j]);
    return 1;
}

/* Builds the pair's manifold: <= 4 points (cp on A, cq on B, separation cs, carried impulse cw); returns their number.
   (op, oq, os, ow)[no]: the points the previous manifold kept; (pa, pb, sep_new): the closest-point / deepest-point pair --
   no pair of points of the two hulls is closer (deeper) than it: a clipped corner's separation below it is an artefact of
   the plane fit and is raised to it. */
static int face_manifold(const shape* A, const shape* B, v3 n, float margin, float dup2, const slhip_settle_params* prm,
                         int no, const v3* op, const v3* oq, const float* os, const float* ow, v3 pa, v3 pb, float sep_new,
                         v3* cp, v3* cq, float* cs, float* cw, int* gjk_joined)
{
    v3 t1, t2;
    tangents(n, &t1, &t2);
    fm_poly PA, PB;
    const float band = FM_BAND_OFFSETS * prm->contact_offset;
    fm_feature(A, n, t1, t2, 1, band, &PA);
    fm_feature(B, n, t1, t2, 0, band, &PB);
    float cu[FM_CANDIDATES], cv[FM_CANDIDATES], cha[FM_CANDIDATES], chb[FM_CANDIDATES], csep[FM_CANDIDATES], cimp[FM_CANDIDATES];
    int ok[FM_CANDIDATES];
    /* 0..3 the kept points, 4 the closest points: in tangent coordinates like the corners */
    for (int c = 0; c < 5; ++c) {
        ok[c] = c < no || c == 4;
        const v3 a = c < 4 ? (c < no ? op[c] : V(0, 0, 0)) : pa;
        cu[c] = dot(a, t1); cv[c] = dot(a, t2); cha[c] = dot(a, n);
        csep[c] = c < 4 ? (c < no ? os[c] : 0.0f) : sep_new;
        chb[c] = cha[c] - csep[c];
        cimp[c] = c < no ? ow[c] : 0.0f;
    }
    float cmin = 3.0e38f;
    for (int c = 0; c < no; ++c) if (csep[c] < cmin) cmin = csep[c];
    /* hulls that overlap deeper than their support features reach have no face manifold: the kept points and the deepest points */
    const int deep = sep_new < -FM_DEEP_OFFSETS * prm->contact_offset;
    for (int c = 5; c < FM_CANDIDATES; ++c) {
        cimp[c] = 0.0f;
        ok[c] = deep ? 0 : fm_corner(&PA, &PB, c - 5, &cu[c], &cv[c], &cha[c], &chb[c]);
        if (!ok[c]) continue;
        float sep = cha[c] - chb[c];
        if (sep < sep_new) { sep = sep_new; chb[c] = cha[c] - sep; }
        csep[c] = sep;
        if (sep > margin) { ok[c] = 0; continue; }
        if (sep < cmin) cmin = sep;
    }
    /* the closest points join when nothing else is there, or when they are deeper than everything else */
    int any = 0;
    for (int c = 0; c < FM_CANDIDATES; ++c) if (c != 4 && ok[c]) any = 1;
    ok[4] = !any || sep_new < cmin - FM_DEEPER_OFFSETS * prm->contact_offset;
    *gjk_joined = ok[4];
    /* a corner that coincides with a kept point or with the closest points is dropped (the kept point has the impulse) */
    for (int c = 5; c < FM_CANDIDATES; ++c) {
        if (!ok[c]) continue;
        for (int k = 0; k < 5; ++k) {
            if (!ok[k]) continue;
            const float du = cu[c] - cu[k], dw = cv[c] - cv[k], dh = cha[c] - cha[k];
            if (fmaf(dh, dh, fmaf(dw, dw, du * du)) < dup2) ok[c] = 0;
        }
    }
    /* the deepest */
    int i0 = -1;
    for (int c = 0; c < FM_CANDIDATES; ++c) if (ok[c] && (i0 < 0 || csep[c] < csep[i0])) i0 = c;
    /* the farthest from it (not a duplicate of it), a metre of extra separation costing DEPTH_WEIGHT metres of reach */
    int i1 = -1; float best = 0.0f;
    for (int c = 0; c < FM_CANDIDATES; ++c) {
        if (!ok[c] || c == i0) continue;
        const float du = cu[c] - cu[i0], dw = cv[c] - cv[i0], dh = cha[c] - cha[i0];
        const float d2 = fmaf(dh, dh, fmaf(dw, dw, du * du));
        if (d2 < dup2) continue;
        const float score = sqrtf(d2) - DEPTH_WEIGHT * (csep[c] - csep[i0]);
        if (i1 < 0 || score > best) { best = score; i1 = c; }
    }
    int i2 = -1, i3 = -1;
    if (i1 >= 0) {
        const float eu = cu[i1] - cu[i0], ew = cv[i1] - cv[i0];
        const float el = sqrtf(fmaf(ew, ew, eu * eu));
        float mx = 0.0f, mn = 0.0f;
        for (int c = 0; c < FM_CANDIDATES; ++c) {
            if (!ok[c] || c == i0 || c == i1) continue;
            const float a = eu * (cv[c] - cv[i0]) - ew * (cu[c] - cu[i0]);
            const float pen = DEPTH_WEIGHT * (csep[c] - csep[i0]) * el;
            if (a - pen > mx) { mx = a - pen; i2 = c; }
            if (a + pen < mn) { mn = a + pen; i3 = c; }
        }
    }
    const int sel[4] = {i0, i1, i2, i3};
    int nc = 0;
    for (int k = 0; k < 4; ++k) {
        const int c = sel[k];
        if (c < 0) continue;
        if (c < 4) { cp[nc] = op[c]; cq[nc] = oq[c]; }
        else if (c == 4) { cp[nc] = pa; cq[nc] = pb; }
        else {
            const v3 base = madd(scale(t1, cu[c]), t2, cv[c]);
            cp[nc] = madd(base, n, cha[c]); cq[nc] = madd(base, n, chb[c]);
        }
        cs[nc] = csep[c]; cw[nc] = cimp[c]; ++nc;
    }
    return nc;
}

/* hull pair -> up to 4 contacts written to out[0..3]; returns min separation (or +inf).
   `pm`: the pair's persistent manifold (NULL: none is kept -- every step builds its manifold from scratch). */
static float hull_pair_contacts(const slhip_body* bodies, const wbody* wbs, int ia, int ib,
                                const slhip_hull* ha, const slhip_hull* hb, const float* hull_verts,
                                const slhip_settle_params* prm, float margin, contact* out, gjk_seed* cached,
                                pmanifold* pm, int step)
{
    for (int i = 0; i < PATCH_STRIDE; ++i) { out[i].valid = 0; out[i].culled = 0; out[i].ln = 0.0f; }
    const wbody* wa = &wbs[ia];
    const wbody* wb = &wbs[ib];
    shape A, B;
    make_shape(wa, ha, hull_verts, &A);
    make_shape(wb, hb, hull_verts, &B);
    v3 ca = add(m3_mul(&wa->R, V(ha->sphere[0], ha->sphere[1], ha->sphere[2])), wa->t);
    v3 cb = add(m3_mul(&wb->R, V(hb->sphere[0], hb->sphere[1], hb->sphere[2])), wb->t);
    v3 pa, pb, n;
    float dist;
    float rest = 2.0f * prm->rest_offset;
    float mu_s = 0.5f * (bodies[ia].mu_s + bodies[ib].mu_s);
    float mu_d = 0.5f * (bodies[ia].mu_d + bodies[ib].mu_d);
    float e = 0.5f * (bodies[ia].restitution + bodies[ib].restitution);
    gjk_seed seed;
    pmanifold prev;
    memset(&prev, 0, sizeof(prev));
    if (pm && pm->stamp == step - 1) { prev = *pm; out[MAX_CONTACTS_PER_HP].ln = WARM_START * prev.lc; }
    /* what a pair keeps from step to step lives as long as the pair stays a broadphase candidate (PhysX destroys a pair's contact
       manager and cache when its bounds stop overlapping [ext]): a pair that was not listed in the previous step starts cold */
    if (pm && cached && pm->stamp != step - 1) { cached->n = 0; cached->idx[0] = cached->idx[1] = cached->idx[2] = 0; }
    if (pm) { pm->stamp = step; pm->count = 0; pm->lc = 0.0f; }
    int code = gjk_distance_seeded(&A, &B, sub(ca, cb), margin, &pa, &pb, &dist, cached, &seed, GJK_MAX_ITER);
    if (g_stats) g_stats[1024 + (g_last_gjk_iters > 63 ? 63 : g_last_gjk_iters)]++; /* [1024, 1088): iterations of the main runs */
    if (cached && code != 0) *cached = seed; /* overlap keeps the previous entry */
    if (code == 2) return 3.0e38f;
    if (code == 1 && dist > margin) return 3.0e38f;
    const float radius = ha->sphere[3] <= hb->sphere[3] ? ha->sphere[3] : hb->sphere[3];
    const float dup2 = 2.5e-3f * radius * radius;

    v3 cp[4], cq[4];
    float cs[4], cw[4];
    int nc = 0;
    float sep_new;
    if (code == 0) {
        /* overlap: penetration depth, normal and witnesses by portal refinement (7 fixed axes if it finds no portal) */
        if (!mpr_penetration(&A, &B, ca, cb, &n, &sep_new, &pa, &pb)) {
            overlap_fallback(&A, &B, ca, cb, &n, &sep_new, &pa, &pb);
            if (sep_new > 0.0f) sep_new = 0.0f;
        }
    } else {
        n = scale(sub(pa, pb), 1.0f / dist);
        sep_new = dist;
    }
    v3 built_nb = m3_tmul(&wb->R, n);   /* a manifold built in this step is filed with this step's normal */
    /* the previous step's points in the new poses: separations along the new normal; a point whose witnesses drifted apart
       laterally, or that left the contact band, is lost -- all of them when the normal turned by more than 10 degrees against B
       since the manifold was built */
    v3 op[4], oq[4];
    float os[4], ow[4];
    int no = 0, lost = 0, n_left = 0, n_drift = 0, n_turn = 0;
    if (prev.count > 0) {
        const float lim = DRIFT_OFFSETS * prm->contact_offset;
        float omin = 3.0e38f;
        if (dot(n, m3_mul(&wb->R, prev.nb)) >= NORMAL_COS)
            for (int i = 0; i < prev.count; ++i) {
                v3 qa = add(m3_mul(&wa->R, prev.la[i]), wa->t);
                v3 qb = add(m3_mul(&wb->R, prev.lb[i]), wb->t);
                v3 d = sub(qa, qb);
                float sp = dot(d, n);
                if (sp > margin) { ++n_left; continue; }
                v3 lat = sub(d, scale(n, sp));
                if (dot(lat, lat) > lim * lim) { ++n_drift; continue; }
                op[no] = qa; oq[no] = qb; os[no] = sp; ow[no] = prev.ln[i]; ++no;
                if (sp < omin) omin = sp;
            }
        else n_turn = 1;
        /* ... and the manifold no longer holds the pair's deepest feature when the closest points are deeper than all of it */
        lost = no < prev.count || sep_new < omin - FM_DEEPER_OFFSETS * prm->contact_offset;
        if (g_fm_stats && lost) g_fm_stats[n_turn ? 5 : n_drift ? 6 : n_left ? 7 : 8]++;
    }
    if (prev.count == 0 || lost) {
        /* a NEW contact pair, or one whose manifold lost a point: the face manifold in one step -- the points that stayed (with
           their impulses), the clipped support features, the closest points when they are deeper than all of those */
        int joined = 0;
        nc = face_manifold(&A, &B, n, margin, dup2, prm, no, op, oq, os, ow, pa, pb, sep_new, cp, cq, cs, cw, &joined);
        if (g_fm_stats) { g_fm_stats[prev.count == 0 ? 0 : 1]++; g_fm_stats[3] += (uint64_t)nc; g_fm_stats[4] += (uint64_t)joined; }
    } else {
        /* the manifold as it was built, its points where the bodies carried them */
        if (g_fm_stats) g_fm_stats[2]++;
        for (int i = 0; i < no; ++i) { cp[i] = op[i]; cq[i] = oq[i]; cs[i] = os[i]; cw[i] = ow[i]; }
        nc = no;
        built_nb = prev.nb;
    }
    int keep[4];
    int nk = reduce4(nc, cp, cs, n, keep);
    float mins = 3.0e38f;
    for (int i = 0; i < nk; ++i) {
        int j = keep[i];
        fill_contact(&out[i], ia, ib, wa, wb, cp[j], cq[j], n, cs[j], rest, mu_s, mu_d, e);
        out[i].ln = WARM_START * cw[j];
        if (cs[j] < mins) mins = cs[j];
        if (pm) {
            pm->la[i] = m3_tmul(&wa->R, sub(cp[j], wa->t));
            pm->lb[i] = m3_tmul(&wb->R, sub(cq[j], wb->t));
            pm->ln[i] = 0.0f;
        }
    }
    if (pm) {
        pm->count = nk;
        pm->nb = built_nb;
    }
    return mins;
}

/* body vs plane (top face of the table box, scene.cpp:629-663): up to 4 contacts chosen from ALL
   hull vertices inside the contact band by four order-independent reductions with first-index
   tie breaks (the same selection rule as reduce4, without a candidate buffer):
     i0 = argmin sep;  i1 = argmax |p - p0| - W (sep - sep0);
     i2 = argmax  area(p0,p1,p) - W (sep - sep0) |p1 - p0|   (only if > 0)
     i3 = argmin  area(p0,p1,p) + W (sep - sep0) |p1 - p0|   (only if < 0)                    */
typedef struct { const wbody* w; const slhip_body* b; const slhip_hull* hulls; const float* verts; float plane_z, margin; } plane_it;

static int plane_vertex(const plane_it* it, uint32_t h, uint32_t i, v3* p, float* d)
{
    const slhip_hull* hh = &it->hulls[h];
    const float* vs = it->verts + 4 * (size_t)hh->vtx_begin;
    *p = add(m3_mul(&it->w->R, V(vs[4 * i], vs[4 * i + 1], vs[4 * i + 2])), it->w->t);
    *d = p->z - it->plane_z;
    return *d <= it->margin;
}

static void plane_contacts(const slhip_body* bodies, const wbody* wbs, int ia, const slhip_hull* hulls,
                           const float* hull_verts, const slhip_settle_params* prm, float plane_z,
                           float margin, contact* out, pplane* pp, int step)
{
    for (int i = 0; i < PATCH_STRIDE; ++i) { out[i].valid = 0; out[i].culled = 0; out[i].ln = 0.0f; }
    const slhip_body* b = &bodies[ia];
    const wbody* w = &wbs[ia];
    plane_it it = {w, b, hulls, hull_verts, plane_z, margin};
    /* pass 0: deepest */
    int have0 = 0; v3 p0 = V(0, 0, 0); float s0 = 0.0f;
    int id0 = 0, id1 = 0, id2 = 0, id3 = 0;
    pplane prev;
    prev.count = 0;
    if (pp && pp->stamp == step - 1) { prev = *pp; out[PLANE_SLOTS].ln = WARM_START * prev.lc; }
    if (pp) { pp->stamp = step; pp->count = 0; pp->lc = 0.0f; }
#define PLANE_ID(h, i) ((int)(((h) - b->hull_begin) << 8 | (i)))
    for (uint32_t h = b->hull_begin; h < b->hull_end; ++h)
        for (uint32_t i = 0; i < hulls[h].vtx_count; ++i) {
            v3 p; float d;
            if (!plane_vertex(&it, h, i, &p, &d)) continue;
            if (!have0 || d < s0) { have0 = 1; p0 = p; s0 = d; id0 = PLANE_ID(h, i); }
        }
    if (!have0) return;
    /* pass 1: farthest (depth-penalised) */
    int have1 = 0; v3 p1 = V(0, 0, 0); float s1 = 0.0f, best = 0.0f;
    for (uint32_t h = b->hull_begin; h < b->hull_end; ++h)
        for (uint32_t i = 0; i < hulls[h].vtx_count; ++i) {
            v3 p; float d;
            if (!plane_vertex(&it, h, i, &p, &d)) continue;
            v3 dd = sub(p, p0);
            float score = sqrtf(dot(dd, dd)) - PLANE_DEPTH_WEIGHT * (d - s0);
            if (score > 0.0f && (!have1 || score > best)) { have1 = 1; best = score; p1 = p; s1 = d; id1 = PLANE_ID(h, i); }
        }
    v3 n = V(0, 0, 1);
    v3 sel[4]; float ss[4]; int ids[4]; int nk = 0;
    sel[nk] = p0; ids[nk] = id0; ss[nk++] = s0;
    if (have1) {
        sel[nk] = p1; ids[nk] = id1; ss[nk++] = s1;
        /* pass 2+3: area extremes */
        v3 e = sub(p1, p0);
        float el = sqrtf(dot(e, e));
        int have2 = 0, have3 = 0; v3 p2 = V(0, 0, 0), p3 = V(0, 0, 0); float s2 = 0, s3 = 0, mx = 0.0f, mn = 0.0f;
        for (uint32_t h = b->hull_begin; h < b->hull_end; ++h)
            for (uint32_t i = 0; i < hulls[h].vtx_count; ++i) {
                v3 p; float d;
                if (!plane_vertex(&it, h, i, &p, &d)) continue;
                float a = dot(cross(e, sub(p, p0)), n);
                float pen = PLANE_DEPTH_WEIGHT * (d - s0) * el;
                if (a - pen > mx) { mx = a - pen; have2 = 1; p2 = p; s2 = d; id2 = PLANE_ID(h, i); }
                if (a + pen < mn) { mn = a + pen; have3 = 1; p3 = p; s3 = d; id3 = PLANE_ID(h, i); }
            }
        if (have2) { sel[nk] = p2; ids[nk] = id2; ss[nk++] = s2; }
        if (have3) { sel[nk] = p3; ids[nk] = id3; ss[nk++] = s3; }
    }
#undef PLANE_ID
    float mu_s = 0.5f * (b->mu_s + prm->plane_mu_s);
    float mu_d = 0.5f * (b->mu_d + prm->plane_mu_d);
    float e_ = 0.5f * (b->restitution + prm->plane_restitution);
    for (int i = 0; i < nk; ++i) {
        v3 pb = V(sel[i].x, sel[i].y, plane_z);
        fill_contact(&out[i], ia, -1, w, NULL, sel[i], pb, n, ss[i], prm->rest_offset, mu_s, mu_d, e_);
        /* the vertex was a contact in the last step: its impulse is where the sweeps start */
        for (int j = 0; j < prev.count; ++j)
            if (prev.id[j] == ids[i]) out[i].ln = WARM_START * prev.ln[j];
        if (pp) { pp->id[i] = ids[i]; pp->ln[i] = 0.0f; }
    }
    if (pp) pp->count = nk;
}

/* ------------------------------------------------------------------------------------------ */
/* solver                                                                                      */
/* ------------------------------------------------------------------------------------------ */
static inline v3 vel_at(const wbody* b, v3 r) { return add(b->v, cross(b->w, r)); }

static inline float eff_mass(const wbody* a, const wbody* b, v3 ra, v3 rb, v3 d)
{
    float k = 0.0f;
    if (a->dynamic) {
        v3 rn = cross(ra, d);
        k += a->inv_mass + dot(cross(m3_mul(&a->Iinv_w, rn), ra), d);
    }
    if (b && b->dynamic) {
        v3 rn = cross(rb, d);
        k += b->inv_mass + dot(cross(m3_mul(&b->Iinv_w, rn), rb), d);
    }
    return k > 0.0f ? 1.0f / k : 0.0f;
}

static void prep_contact(contact* c, const wbody* wbs)
{
    if (!c->valid) return;
    const wbody* a = &wbs[c->a];
    const wbody* b = c->b >= 0 ? &wbs[c->b] : NULL;
    if (!a->dynamic && !(b && b->dynamic)) { c->valid = 0; return; }
    c->kn = eff_mass(a, b, c->ra, c->rb, c->n);
    c->kt1 = eff_mass(a, b, c->ra, c->rb, c->t1);
    c->kt2 = eff_mass(a, b, c->ra, c->rb, c->t2);
    v3 rel = vel_at(a, c->ra);
    if (b) rel = sub(rel, vel_at(b, c->rb));
    c->vn0 = dot(rel, c->n);
}

static inline void apply_impulse(wbody* a, wbody* b, const contact* c, v3 J)
{
    if (a->dynamic) {
        a->v = madd(a->v, J, a->inv_mass);
        a->w = add(a->w, m3_mul(&a->Iinv_w, cross(c->ra, J)));
    }
    if (b && b->dynamic) {
        b->v = madd(b->v, J, -b->inv_mass);
        b->w = sub(b->w, m3_mul(&b->Iinv_w, cross(c->rb, J)));
    }
}

/* normal row of one contact */
static void solve_normal(contact* c, wbody* wbs, const slhip_settle_params* prm, int biased)
{
    wbody* a = &wbs[c->a];
    wbody* b = c->b >= 0 ? &wbs[c->b] : NULL;
    const float inv_dt = 1.0f / prm->dt;
    v3 rel = vel_at(a, c->ra);
    if (b) rel = sub(rel, vel_at(b, c->rb));
    float vn = dot(rel, c->n);
    float err = c->sep - c->rest;
    float target; /* required vn >= target */
    if (err > 0.0f) target = -err * inv_dt;                 /* speculative: may close the gap */
    else {                                                  /* push out (position iterations only) */
        target = biased ? -0.8f * err * inv_dt : 0.0f;
    }
    if (c->vn0 < -prm->bounce_threshold && c->e > 0.0f) {
        float bounce = -c->e * c->vn0;
        if (bounce > target) target = bounce;
    }
    float dl = (target - vn) * c->kn;
    float ln = c->ln + dl;
    if (ln < 0.0f) ln = 0.0f;
    dl = ln - c->ln;
    c->ln = ln;
    apply_impulse(a, b, c, scale(c->n, dl));
}

/* friction rows of one ANCHOR of a patch: static / dynamic Coulomb on the tangent impulse vector against the anchor's
   share of the patch's normal impulse */
static void solve_friction(contact* c, wbody* wbs, float share)
{
    wbody* a = &wbs[c->a];
    wbody* b = c->b >= 0 ? &wbs[c->b] : NULL;
    v3 rel = vel_at(a, c->ra);
    if (b) rel = sub(rel, vel_at(b, c->rb));
    float l1 = c->lt1 - dot(rel, c->t1) * c->kt1;
    float l2 = c->lt2 - dot(rel, c->t2) * c->kt2;
    float mag2 = fmaf(l2, l2, l1 * l1);
    float lim_s = c->mu_s * share;
    if (mag2 > lim_s * lim_s) {
        float mag = sqrtf(mag2);
        float k = (c->mu_d * share) / mag;
        l1 *= k; l2 *= k;
    }
    float d1 = l1 - c->lt1, d2 = l2 - c->lt2;
    c->lt1 = l1; c->lt2 = l2;
    apply_impulse(a, b, c, madd(scale(c->t1, d1), c->t2, d2));
}

/* One friction PATCH = the manifold of one hull pair (or of one body against the table): <= 4 contacts with a common normal
   in a block of MAX_CONTACTS_PER_HP slots, the valid ones first.  PhysX's configured friction model (PxFrictionType::ePATCH
   [ext]): every contact has a normal row, friction acts at two ANCHORS per patch -- here the first two
   contacts of the manifold, i.e. its deepest point and the point farthest from it (reduce4) -- each limited by mu times its share
   of the patch's accumulated normal impulse (half each with two anchors: the total stays inside the Coulomb cone).  Order within
   a patch: the normal rows in slot order, then the anchors. */
static void solve_patch(contact* c, wbody* wbs, const slhip_settle_params* prm, int biased)
{
    float nsum = 0.0f;
    int m = 0;
    /* the centre row first: it takes the part of the correction the whole face shares, without turning the bodies against each
       other; the points' rows then see what is left (patch_centre) */
    if (c[MAX_CONTACTS_PER_HP].valid) {
        solve_normal(&c[MAX_CONTACTS_PER_HP], wbs, prm, biased);
        nsum = nsum + c[MAX_CONTACTS_PER_HP].ln;
    }
    for (int i = 0; i < MAX_CONTACTS_PER_HP; ++i) {
        if (!c[i].valid) continue;
        solve_normal(&c[i], wbs, prm, biased);
        nsum = nsum + c[i].ln;
        ++m;
    }
    if (m == 0) return;
    const int anchors = m >= 2 ? 2 : 1;
    const float share = anchors == 2 ? 0.5f * nsum : nsum;
    int done = 0;
    for (int i = 0; i < MAX_CONTACTS_PER_HP && done < anchors; ++i) {
        if (!c[i].valid) continue;
        solve_friction(&c[i], wbs, share);
        ++done;
    }
}

/* The CENTRE ROW of a patch of three or four points (a face resting on a face): one more normal row at the mean of the points,
   with their mean separation -- a point of the same contact face, so nothing the manifold does not already say.  Gauss-Seidel
   walks a patch's points one after the other; the first one met takes most of a correction all of them share and turns the
   bodies against each other, the following rows take part of that back, and what is left after 4 + 4 sweeps is a net angular
   velocity in the same sense step after step: a column of cubes leans over within seconds (measured in round 6: exact boxes,
   0.1 rad/s per step from a symmetric start).  The centre row carries the shared part through the patch's centre, where it
   turns nothing; the corner rows are left with the differences.  No friction acts at it (the anchors stay the first two
   points); its impulse counts towards the patch's normal impulse and is carried from step to step like the points'. */
static void patch_centre(contact* c)
{
    contact* cc = &c[MAX_CONTACTS_PER_HP];
    const float carried = cc->ln;
    cc->valid = 0;
    int m = 0;
    v3 ra = V(0, 0, 0), rb = V(0, 0, 0);
    float sep = 0.0f;
    for (int i = 0; i < MAX_CONTACTS_PER_HP; ++i) {
        if (!c[i].valid) continue;
        ra = add(ra, c[i].ra); rb = add(rb, c[i].rb); sep = sep + c[i].sep;
        ++m;
    }
    if (m < 3) return;
    const float inv = 1.0f / (float)m;
    *cc = c[0];
    cc->ra = scale(ra, inv); cc->rb = scale(rb, inv); cc->sep = sep * inv;
    cc->ln = carried; cc->lt1 = 0.0f; cc->lt2 = 0.0f;
    cc->valid = 1; cc->culled = 0;
}

/* D6 joint of ManipulationSim (manipulation_sim.cpp:46-93): world-anchored, linear X/Y/Z driven by
   an implicit spring (soft constraint: gamma = 1/(dt (d + dt k)), beta = dt k / (d + dt k), impulse
   per axis limited to forceLimit * dt), locked rotation axes as hard rows with the same 0.8/dt
   positional bias as contacts (position iterations only). */
static void solve_drive(const slhip_body* b, wbody* w, const slhip_settle_params* prm, int biased)
{
    if (!(b->drive_flags & 1u) || !w->dynamic) return;
    const float dt = prm->dt;
    const float k = b->drive_params[0], d = b->drive_params[1], flim = b->drive_params[2] * dt;
    quat qj = {b->drive_frame[0], b->drive_frame[1], b->drive_frame[2], b->drive_frame[3]};
    m3 J;
    quat_to_m3(qj, &J);
    v3 r = sub(w->t, w->x); /* object origin relative to the COM */
    v3 err = sub(w->t, V(b->drive_target[0], b->drive_target[1], b->drive_target[2]));
    const float den = d + dt * k;
    const float gamma = 1.0f / (dt * den);
    const float beta = dt * k / den;
    for (int a = 0; a < 3; ++a) {
        v3 ax = V(J.m[a], J.m[3 + a], J.m[6 + a]); /* column a of the joint rotation */
        v3 rn = cross(r, ax);
        float K = w->inv_mass + dot(cross(m3_mul(&w->Iinv_w, rn), r), ax);
        float meff = 1.0f / (K + gamma);
        float u = dot(vel_at(w, r), ax);
        float C = dot(err, ax);
        float dlam = -meff * (u + (beta / dt) * C + gamma * w->dl[a]);
        float lam = w->dl[a] + dlam;
        if (lam > flim) lam = flim;
        if (lam < -flim) lam = -flim;
        dlam = lam - w->dl[a];
        w->dl[a] = lam;
        v3 Jimp = scale(ax, dlam);
        w->v = madd(w->v, Jimp, w->inv_mass);
        w->w = add(w->w, m3_mul(&w->Iinv_w, cross(r, Jimp)));
    }
    /* rotation error of the body w.r.t. the joint frame: q_err = q * conj(qj), theta ~ 2 q_err.xyz */
    quat qc = {-qj.x, -qj.y, -qj.z, qj.w};
    quat qe = quat_mul(w->q, qc);
    float sgn = qe.w < 0.0f ? -2.0f : 2.0f;
    v3 theta = V(qe.x * sgn, qe.y * sgn, qe.z * sgn);
    for (int a = 0; a < 3; ++a) {
        if (!(b->drive_flags & (2u << a))) continue;
        v3 ax = V(J.m[a], J.m[3 + a], J.m[6 + a]);
        float K = dot(m3_mul(&w->Iinv_w, ax), ax);
        if (!(K > 0.0f)) continue;
        float bias = biased ? 0.8f * dot(theta, ax) / dt : 0.0f;
        float dlam = -(dot(w->w, ax) + bias) / K;
        w->da[a] += dlam;
        w->w = add(w->w, m3_mul(&w->Iinv_w, scale(ax, dlam)));
    }
}

/* Greedy colouring, LARGEST GROUP FIRST (size = valid contacts, ties in group order): a group gets the
   smallest colour unused by its bodies.  A colour's groups are solved side by side, so a sweep
   costs the sum over the colours of their largest group; taking the big groups first lets them share
   the low colours (measured on the C2 workload: 19.1 instead of 22.5 contacts per sweep, 17.8 being the
   bound set by the busiest body).  The order within a colour does not matter (disjoint bodies). */
static void color_groups(scene_ws* ws, int n_bodies)
{
    uint64_t* used = ws->used;
    for (int i = 0; i < n_bodies; ++i) used[i] = 0;
    int *order = ws->order, *size = ws->size;
    for (int g = 0; g < ws->n_groups; ++g) {
        int c = 0;
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) c += ws->c[i].valid ? 1 : 0;
        size[g] = c;
    }
    for (int g = 0; g < ws->n_groups; ++g) { /* rank = groups that come before g */
        int rank = 0;
        for (int h = 0; h < ws->n_groups; ++h)
            if (size[h] > size[g] || (size[h] == size[g] && h < g)) ++rank;
        order[rank] = g;
    }
    /* greedy, largest group first: the lowest of the colours 0..62 that neither body uses; a group whose bodies leave none of them
       free (a hub with more than 63 neighbours) gets a colour of its own from 63 on -- it is swept alone, after the others */
    int nc = 0, extra = 63;
    for (int q = 0; q < ws->n_groups; ++q) {
        const int g = order[q];
        uint64_t m = used[ws->g_a[g]];
        if (ws->g_b[g] >= 0) m |= used[ws->g_b[g]];
        int c = 0;
        while (c < 63 && (m >> c) & 1ull) ++c;
        if (c == 63) c = extra++;
        else {
            used[ws->g_a[g]] |= 1ull << c;
            if (ws->g_b[g] >= 0) used[ws->g_b[g]] |= 1ull << c;
        }
        ws->g_color[g] = c;
        if (c + 1 > nc) nc = c + 1;
    }
    ws->n_colors = nc;
}

/* optional statistics of the solver's work for tools/solver_stats.py and DESIGN.md (not thread safe): four histograms of 256
   bins each over the steps (+ 64 bins from 1024 on: iterations of the main GJK runs, per hull pair) -- active contacts, friction anchors, colours, and the chain length of one sweep (sum over the colours
   of the largest group's contacts + anchors: what a lane pair per group has to walk in sequence) */
void slref_settle_set_stats(uint64_t* h) { g_stats = h; }
static uint64_t* g_offered = NULL; /* histogram [2048] of the contacts a step offers before the cap (tools/physics_quality.py) */
void slref_settle_set_offered_hist(uint64_t* h) { g_offered = h; }
static uint64_t* g_pairs_hist = NULL; /* histogram [8192] of the candidate hull pairs a step's broadphase finds (incl. those beyond the list) */
void slref_settle_set_pairs_hist(uint64_t* h) { g_pairs_hist = h; }
static FILE* g_profile_fp = NULL;
void slref_settle_set_profile_dump(const char* path)
{
    if (g_profile_fp) { fclose(g_profile_fp); g_profile_fp = NULL; }
    if (path) g_profile_fp = fopen(path, "w");
}

static void step_stats(const scene_ws* ws)
{
    int active = 0, anchors = 0, chain = 0;
    int longest[64] = {0};
    for (int g = 0; g < ws->n_groups; ++g) {
        int rows = 0;
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; i += PATCH_STRIDE) {
            int m = 0;
            for (int k = 0; k < PATCH_STRIDE; ++k) m += ws->c[i + k].valid ? 1 : 0;
            active += m;
            anchors += m >= 2 ? 2 : m;
            rows += m + (m >= 2 ? 2 : m);
        }
        if (ws->g_color[g] < 64 && rows > longest[ws->g_color[g]]) longest[ws->g_color[g]] = rows;
    }
    for (int c = 0; c < ws->n_colors && c < 64; ++c) chain += longest[c];
    if (g_profile_fp && ws->n_colors <= 64) { /* developer dump (tools/solver_pairing.py): one line per scene and step, the longest group per colour */
        fprintf(g_profile_fp, "%d", ws->n_colors);
        for (int c = 0; c < ws->n_colors; ++c) fprintf(g_profile_fp, " %d", longest[c]);
        fprintf(g_profile_fp, "\n");
    }
    g_stats[active > 255 ? 255 : active]++;
    g_stats[256 + (anchors > 255 ? 255 : anchors)]++;
    g_stats[512 + (ws->n_colors > 255 ? 255 : ws->n_colors)]++;
    g_stats[768 + (chain > 255 ? 255 : chain)]++;
}

static void solve_iteration(scene_ws* ws, const slhip_body* bodies, int nb, const slhip_settle_params* prm, int biased)
{
    for (int col = 0; col < ws->n_colors; ++col)
        for (int g = 0; g < ws->n_groups; ++g) {
            if (ws->g_color[g] != col) continue;
            for (int i = ws->g_begin[g]; i < ws->g_end[g]; i += PATCH_STRIDE) solve_patch(&ws->c[i], ws->wb, prm, biased);
        }
    for (int i = 0; i < nb; ++i) solve_drive(&bodies[i], &ws->wb[i], prm, biased);
}

/* world AABB of a hull = |R| * local half extents around R c + t; boxes must overlap within margin */
static int aabb_overlap(const wbody* wa, const slhip_hull* ha, const wbody* wb_, const slhip_hull* hb, float margin)
{
    v3 ca = add(m3_mul(&wa->R, V(ha->aabb_center[0], ha->aabb_center[1], ha->aabb_center[2])), wa->t);
    v3 cb = add(m3_mul(&wb_->R, V(hb->aabb_center[0], hb->aabb_center[1], hb->aabb_center[2])), wb_->t);
    float ea[3], eb[3];
    for (int r = 0; r < 3; ++r) {
        ea[r] = fmaf(fabsf(wa->R.m[3 * r + 2]), ha->aabb_half[2], fmaf(fabsf(wa->R.m[3 * r + 1]), ha->aabb_half[1], fabsf(wa->R.m[3 * r]) * ha->aabb_half[0]));
        eb[r] = fmaf(fabsf(wb_->R.m[3 * r + 2]), hb->aabb_half[2], fmaf(fabsf(wb_->R.m[3 * r + 1]), hb->aabb_half[1], fabsf(wb_->R.m[3 * r]) * hb->aabb_half[0]));
    }
    if (fabsf(ca.x - cb.x) > ea[0] + eb[0] + margin) return 0;
    if (fabsf(ca.y - cb.y) > ea[1] + eb[1] + margin) return 0;
    if (fabsf(ca.z - cb.z) > ea[2] + eb[2] + margin) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* one step                                                                                    */
/* ------------------------------------------------------------------------------------------ */
static void load_body(const slhip_body* b, wbody* w)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) w->R.m[3 * r + c] = b->pose[4 * r + c];
    w->t = V(b->pose[3], b->pose[7], b->pose[11]);
    w->q = m3_to_quat(&w->R);
    quat_to_m3(w->q, &w->R); /* re-orthonormalised rotation is what the step uses */
    w->x = add(m3_mul(&w->R, V(b->com[0], b->com[1], b->com[2])), w->t);
    w->v = V(b->lin_vel[0], b->lin_vel[1], b->lin_vel[2]);
    w->w = V(b->ang_vel[0], b->ang_vel[1], b->ang_vel[2]);
    w->inv_mass = b->inv_mass;
    w->dynamic = !(b->flags & (SLHIP_BODY_STATIC | SLHIP_BODY_ASLEEP)) && b->inv_mass > 0.0f;
    for (int k = 0; k < 3; ++k) { w->dl[k] = 0.0f; w->da[k] = 0.0f; }
}

static void update_world_inertia(const slhip_body* b, wbody* w)
{
    /* Iinv_w = R Iinv R^T */
    m3 L, T;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) L.m[3 * r + c] = b->inv_inertia[4 * r + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            T.m[3 * r + c] = fmaf(w->R.m[3 * r + 2], L.m[6 + c], fmaf(w->R.m[3 * r + 1], L.m[3 + c], w->R.m[3 * r] * L.m[c]));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            w->Iinv_w.m[3 * r + c] = fmaf(T.m[3 * r + 2], w->R.m[3 * c + 2], fmaf(T.m[3 * r + 1], w->R.m[3 * c + 1], T.m[3 * r] * w->R.m[3 * c]));
}

/* mass-normalised kinetic energy [ext]: 0.5 (v.v + w.(I w) / m), I = inverse of inv_inertia (object axes, by cofactors) */
static float kinetic_energy(const slhip_body* b, const wbody* wbd, v3 v, v3 w)
{
    const float* L = b->inv_inertia;
    v3 wl = m3_tmul(&wbd->R, w);
    float c00 = L[5] * L[10] - L[6] * L[9], c01 = L[6] * L[8] - L[4] * L[10], c02 = L[4] * L[9] - L[5] * L[8];
    float c11 = L[0] * L[10] - L[2] * L[8], c12 = L[1] * L[8] - L[0] * L[9], c22 = L[0] * L[5] - L[1] * L[4];
    float det = fmaf(L[2], c02, fmaf(L[1], c01, L[0] * c00));
    v3 iw = V(fmaf(c02, wl.z, fmaf(c01, wl.y, c00 * wl.x)), fmaf(c12, wl.z, fmaf(c11, wl.y, c01 * wl.x)),
              fmaf(c22, wl.z, fmaf(c12, wl.y, c02 * wl.x)));
    float ang = det != 0.0f ? dot(wl, iw) / det * wbd->inv_mass : 0.0f;
    return 0.5f * (dot(v, v) + ang);
}

static void store_body(slhip_body* b, const wbody* w)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) b->pose[4 * r + c] = w->R.m[3 * r + c];
    b->pose[3] = w->t.x; b->pose[7] = w->t.y; b->pose[11] = w->t.z;
    b->pose[12] = 0.0f; b->pose[13] = 0.0f; b->pose[14] = 0.0f; b->pose[15] = 1.0f;
    b->lin_vel[0] = w->v.x; b->lin_vel[1] = w->v.y; b->lin_vel[2] = w->v.z;
    b->ang_vel[0] = w->w.x; b->ang_vel[1] = w->w.y; b->ang_vel[2] = w->w.z;
}

static void step_scene(const slhip_settle_scene* sc, slhip_body* bodies_all, const slhip_hull* hulls,
                       const float* hull_verts, const slhip_settle_params* prm, scene_ws* ws)
{
    slhip_body* bodies = bodies_all + sc->body_begin;
    const int nb = (int)(sc->body_end - sc->body_begin);
    const float dt = prm->dt;
    wbody* wb = ws->wb;

    /* (a) load, integrate forces */
    for (int i = 0; i < nb; ++i) {
        load_body(&bodies[i], &wb[i]);
        update_world_inertia(&bodies[i], &wb[i]);
        bodies[i].separation = 3.0e38f; /* +inf (scene.cpp:732) */
        if (wb[i].dynamic) {
            /* stabilisation: a settling body feels a share of gravity (PhysX: accelScale in the unconstrained-velocity integration [ext]) */
            wb[i].v = madd(wb[i].v, V(prm->gravity[0], prm->gravity[1], prm->gravity[2]), dt * (1.0f - bodies[i].stab[1]));
            float damp = 1.0f - prm->angular_damping * dt;
            if (damp < 0.0f) damp = 0.0f;
            wb[i].w = scale(wb[i].w, damp);
        }
    }

    ws->n_hp = 0;
    ws->n_groups = 0;
    ws->hp_overflow = 0;
    int pairs_found = 0, g_overflow = 0;
    /* (b) plane contacts FIRST: one group per dynamic body near the table (their contacts have
       priority under the active-contact cap); slots live after the hull-pair slots */
    const int plane_base = ws->P * PATCH_STRIDE;
    if (sc->has_plane) {
        for (int i = 0; i < nb; ++i) {
            if (!wb[i].dynamic) continue;
            v3 ci = add(m3_mul(&wb[i].R, V(bodies[i].bsphere[0], bodies[i].bsphere[1], bodies[i].bsphere[2])), wb[i].t);
            float vz = wb[i].v.z < 0.0f ? -wb[i].v.z * dt : 0.0f;
            float margin = prm->contact_offset + vz;
            if (ci.z - bodies[i].bsphere[3] - sc->plane_z > margin) continue;
            contact* out = &ws->c[plane_base + i * PATCH_STRIDE];
            plane_contacts(bodies, wb, i, hulls, hull_verts, prm, sc->plane_z, margin, out, &ws->pp[i], ws->step);
            int g = ws->n_groups++;
            ws->g_a[g] = i; ws->g_b[g] = -1;
            ws->g_begin[g] = plane_base + i * PATCH_STRIDE;
            ws->g_end[g] = plane_base + (i + 1) * PATCH_STRIDE;
        }
    }

    /* (c) broadphase: body pairs in (i<j) order, then their hull pairs */
    for (int i = 0; i < nb; ++i)
        for (int j = i + 1; j < nb; ++j) {
            if (!wb[i].dynamic && !wb[j].dynamic) continue;
            v3 ci = add(m3_mul(&wb[i].R, V(bodies[i].bsphere[0], bodies[i].bsphere[1], bodies[i].bsphere[2])), wb[i].t);
            v3 cj = add(m3_mul(&wb[j].R, V(bodies[j].bsphere[0], bodies[j].bsphere[1], bodies[j].bsphere[2])), wb[j].t);
            /* speculative margin: contact offsets + relative motion of this step */
            v3 dv = sub(wb[i].v, wb[j].v);
            float spec = sqrtf(dot(dv, dv)) * dt;
            float margin = 2.0f * prm->contact_offset + spec;
            v3 d = sub(ci, cj);
            float rr = bodies[i].bsphere[3] + bodies[j].bsphere[3] + margin;
            if (dot(d, d) > rr * rr) continue;
            int first = ws->n_hp;
            for (uint32_t ha = bodies[i].hull_begin; ha < bodies[i].hull_end; ++ha)
                for (uint32_t hb = bodies[j].hull_begin; hb < bodies[j].hull_end; ++hb) {
                    v3 ca = add(m3_mul(&wb[i].R, V(hulls[ha].sphere[0], hulls[ha].sphere[1], hulls[ha].sphere[2])), wb[i].t);
                    v3 cb = add(m3_mul(&wb[j].R, V(hulls[hb].sphere[0], hulls[hb].sphere[1], hulls[hb].sphere[2])), wb[j].t);
                    v3 dd = sub(ca, cb);
                    float r2 = hulls[ha].sphere[3] + hulls[hb].sphere[3] + margin;
                    if (dot(dd, dd) > r2 * r2) continue;
                    if (!aabb_overlap(&wb[i], &hulls[ha], &wb[j], &hulls[hb], margin)) continue;
                    ++pairs_found;
                    if (ws->n_hp >= ws->P) { ws->hp_overflow = 1; continue; } /* beyond the capacity: dropped in list order, counted */
                    int k = ws->n_hp++;
                    ws->hp_ba[k] = i; ws->hp_bb[k] = j; ws->hp_ha[k] = (int)ha; ws->hp_hb[k] = (int)hb;
                }
            if (ws->n_hp > first && ws->n_groups >= ws->G) { /* no room for another group: the body pair is dropped, counted */
                g_overflow = 1;
                ws->n_hp = first;
            }
            if (ws->n_hp > first) {
                int g = ws->n_groups++;
                ws->g_a[g] = i; ws->g_b[g] = j;
                ws->g_begin[g] = first * PATCH_STRIDE;
                ws->g_end[g] = ws->n_hp * PATCH_STRIDE;
            }
        }

    if (ws->hp_overflow) ws->cap_hits[1]++;
    if (g_overflow) ws->cap_hits[5]++;
    if (g_pairs_hist) g_pairs_hist[pairs_found > 8191 ? 8191 : pairs_found]++;
    if ((unsigned)pairs_found > ws->cap_hits[3]) ws->cap_hits[3] = (unsigned)pairs_found;
    /* (d) narrowphase per hull pair */
    for (int k = 0; k < ws->n_hp; ++k) {
        int i = ws->hp_ba[k], j = ws->hp_bb[k];
        v3 dv = sub(wb[i].v, wb[j].v);
        float margin = 2.0f * prm->contact_offset + sqrtf(dot(dv, dv)) * dt;
        gjk_seed* cached = NULL;
        ws->hp_pm[k] = NULL;
        if (ws->cache) {
            int la = ws->body_lh[i] + (ws->hp_ha[k] - (int)bodies[i].hull_begin);
            int lb = ws->body_lh[j] + (ws->hp_hb[k] - (int)bodies[j].hull_begin);
            cached = &ws->cache[(size_t)la * ws->n_hulls + lb];
            ws->hp_pm[k] = &ws->pm[(size_t)la * ws->n_hulls + lb];
        }
        float s = hull_pair_contacts(bodies, wb, i, j, &hulls[ws->hp_ha[k]], &hulls[ws->hp_hb[k]], hull_verts, prm,
                                     margin, &ws->c[k * PATCH_STRIDE], cached, ws->hp_pm[k], ws->step);
        /* min separation per object (scene.cpp:73-116; plane contacts are ignored there) */
        if (s < bodies[i].separation) bodies[i].separation = s;
        if (s < bodies[j].separation) bodies[j].separation = s;
    }

    /* manifold thinning: a body pair in contact through MANY hull pairs (decomposed shapes) keeps fewer points per hull pair --
       all four up to THIN_FULL contact pairs, two up to THIN_HALF, one beyond (reduce4's order: the deepest first) */
    for (int g = 0; g < ws->n_groups; ++g) {
        if (ws->g_b[g] < 0) continue;
        int n_cp = 0;
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; i += PATCH_STRIDE) n_cp += ws->c[i].valid ? 1 : 0;
        const int limit = n_cp <= THIN_FULL ? 4 : n_cp <= THIN_HALF ? 2 : 1;
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; i += PATCH_STRIDE)
            for (int k = limit; k < MAX_CONTACTS_PER_HP; ++k) ws->c[i + k].valid = 0;
    }

    /* compound manifold reduction (slhip_settle_params.pair_contact_budget; not in the reference): a body pair offering more
       points than the budget keeps the deepest ones -- keys ordered like the floats, ties in list order; the others stay in their
       manifolds without impulse */
    {
        int B = (int)prm->pair_contact_budget;
        if (B > 0 && B < 16) B = 16;
        int reduced = 0;
        if (B > 0) for (int g = 0; g < ws->n_groups; ++g) {
            if (ws->g_b[g] < 0) continue;
            int n = 0;
            for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) n += ws->c[i].valid ? 1 : 0;
            if (n <= B) continue;
            reduced = 1;
            while (n > B) {
                int worst = -1, wkey = 0;
                for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) {
                    if (!ws->c[i].valid) continue;
                    int key; memcpy(&key, &ws->c[i].sep, 4);
                    key = key >= 0 ? key : key ^ 0x7fffffff;
                    if (worst < 0 || key >= wkey) { wkey = key; worst = i; }
                }
                ws->c[worst].valid = 0; ws->c[worst].culled = 1; --n;
            }
        }
        if (reduced) ws->cap_hits[4]++;
    }

    /* the centre rows of the patches that kept three or four points */
    for (int g = 0; g < ws->n_groups; ++g)
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; i += PATCH_STRIDE) patch_centre(&ws->c[i]);

    /* The solver takes every contact (PhysX has no cap, scene.cpp:738-739) -- up to the capacity of the list the caller sized
       (max_contacts_per_scene): what a step offers beyond it is dropped in list order, the table's contacts first in the list,
       and counted. */
    {
        int offered = 0;
        for (int g = 0; g < ws->n_groups; ++g)
            for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) offered += ws->c[i].valid ? 1 : 0;
        if (g_offered) g_offered[offered > 2047 ? 2047 : offered]++;
        if ((unsigned)offered > ws->cap_hits[2]) ws->cap_hits[2] = (unsigned)offered;
        ws->cap_hits[6] += (unsigned)(offered > ws->C ? ws->C : offered);
        if (offered > ws->C) {
            ws->cap_hits[0]++;
            int active = 0;
            for (int g = 0; g < ws->n_groups; ++g)
                for (int p = ws->g_begin[g]; p < ws->g_end[g]; p += PATCH_STRIDE)
                    for (int k = 0; k < PATCH_STRIDE; ++k) {
                        const int i = p + PATCH_ORDER(k);      /* list order: a patch's centre row, then its points */
                        if (!ws->c[i].valid) continue;
                        if (active >= ws->C) ws->c[i].valid = 0;
                        else ++active;
                    }
        }
    }

    /* wake sleeping bodies touched by a moving body */
    for (int g = 0; g < ws->n_groups; ++g) {
        int a = ws->g_a[g], b = ws->g_b[g];
        if (b < 0) continue;
        int touching = 0;
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i)
            if (ws->c[i].valid && ws->c[i].sep < 2.0f * prm->contact_offset) touching = 1;
        if (!touching) continue;
        for (int s = 0; s < 2; ++s) {
            int me = s ? b : a, other = s ? a : b;
            if ((bodies[me].flags & SLHIP_BODY_ASLEEP) && wb[other].dynamic) {
                float en = 0.5f * dot(wb[other].v, wb[other].v);
                if (en > prm->sleep_threshold) {
                    bodies[me].flags &= ~SLHIP_BODY_ASLEEP;
                    bodies[me].wake_counter = prm->wake_time;
                    /* becomes dynamic from the next step on */
                }
            }
        }
    }

    /* (f) prep */
    for (int g = 0; g < ws->n_groups; ++g)
        for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) prep_contact(&ws->c[i], wb);

    /* (g) colouring, (h) position iterations (biased) */
    color_groups(ws, nb);
    if (g_stats) step_stats(ws);
    /* warm start: the normal impulses carried over from the previous step (fill: WARM_START of what the manifold's points ended
       with), applied to the bodies in the sweeps' order before the first sweep; the rows then accumulate on top of them */
    for (int col = 0; col < ws->n_colors; ++col)
        for (int g = 0; g < ws->n_groups; ++g) {
            if (ws->g_color[g] != col) continue;
            for (int p = ws->g_begin[g]; p < ws->g_end[g]; p += PATCH_STRIDE)
                for (int k = 0; k < PATCH_STRIDE; ++k) {
                    contact* c = &ws->c[p + PATCH_ORDER(k)];
                    if (!c->valid) continue;
                    apply_impulse(&wb[c->a], c->b >= 0 ? &wb[c->b] : NULL, c, scale(c->n, c->ln));
                }
        }
    for (uint32_t it = 0; it < prm->pos_iters; ++it) solve_iteration(ws, bodies, nb, prm, 1);

    /* (i) integrate poses with the biased velocities */
    for (int i = 0; i < nb; ++i) {
        if (!wb[i].dynamic) continue;
        float lim = bodies[i].max_lin_vel;
        float vv = dot(wb[i].v, wb[i].v);
        if (lim > 0.0f && vv > lim * lim) wb[i].v = scale(wb[i].v, lim / sqrtf(vv));
        float ww = dot(wb[i].w, wb[i].w);
        float wl = prm->max_angular_velocity;
        if (ww > wl * wl) wb[i].w = scale(wb[i].w, wl / sqrtf(ww));
        ws->mv[i] = wb[i].v; ws->mw[i] = wb[i].w;
        wb[i].x = madd(wb[i].x, wb[i].v, dt);
        quat wq = {wb[i].w.x, wb[i].w.y, wb[i].w.z, 0.0f};
        quat dq = quat_mul(wq, wb[i].q);
        quat q = {fmaf(0.5f * dt, dq.x, wb[i].q.x), fmaf(0.5f * dt, dq.y, wb[i].q.y),
                  fmaf(0.5f * dt, dq.z, wb[i].q.z), fmaf(0.5f * dt, dq.w, wb[i].q.w)};
        wb[i].q = quat_normalize(q);
    }

    /* (j) velocity iterations (unbiased): what is left in v,w is carried to the next step */
    for (uint32_t it = 0; it < prm->vel_iters; ++it) solve_iteration(ws, bodies, nb, prm, 0);

    /* the impulses the manifolds carry into the next step */
    for (int k = 0; k < ws->n_hp; ++k) {
        pmanifold* pm = ws->hp_pm[k];
        if (!pm) continue;
        const contact* c = &ws->c[k * PATCH_STRIDE];
        int kept = 0;   /* points the active-contact cap dropped (a suffix) leave the manifold */
        while (kept < pm->count && (c[kept].valid || c[kept].culled)) { pm->ln[kept] = c[kept].valid ? c[kept].ln : 0.0f; ++kept; }
        pm->count = kept;
        pm->lc = c[MAX_CONTACTS_PER_HP].valid ? c[MAX_CONTACTS_PER_HP].ln : 0.0f;
    }
    if (sc->has_plane)
        for (int i = 0; i < nb; ++i) {
            pplane* pp = &ws->pp[i];
            if (pp->stamp != ws->step) continue;
            const contact* c = &ws->c[plane_base + i * PATCH_STRIDE];
            int kept = 0;
            while (kept < pp->count && c[kept].valid) { pp->ln[kept] = c[kept].ln; ++kept; }
            pp->count = kept;
            pp->lc = c[PLANE_SLOTS].valid ? c[PLANE_SLOTS].ln : 0.0f;
        }
    ws->step++;

    /* Stabilisation, PxSceneFlag::eENABLE_STABILIZATION (scene.cpp:163); PhysX 4.1 Dy::updateWakeCounter, stabilisation branch, as
       remembered [ext].  Per body: the body pairs it touches through contacts of this step's solver (PhysX: numCountedInteractions),
       and whether its island -- the bodies connected through such contacts between dynamic bodies -- rests on something that does not
       move: the table, a static or a sleeping body (PhysX: hasStaticTouch).  The flag spreads over the groups until nothing changes. */
    const int stab = prm->stabilization_threshold > 0.0f;
    if (stab) {
        for (int i = 0; i < nb; ++i) { ws->st_n[i] = 0; ws->st_touch[i] = 0; }
        for (int g = 0; g < ws->n_groups; ++g) {
            int any = 0;
            for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) any |= ws->c[i].valid;
            if (!any) continue;
            const int a = ws->g_a[g], b = ws->g_b[g];
            ws->st_n[a]++;
            if (b >= 0) ws->st_n[b]++;
            if (b < 0 || !wb[b].dynamic) ws->st_touch[a] = 1;
            if (b >= 0 && !wb[a].dynamic) ws->st_touch[b] = 1;
        }
        for (int changed = 1; changed;) {
            changed = 0;
            for (int g = 0; g < ws->n_groups; ++g) {
                const int a = ws->g_a[g], b = ws->g_b[g];
                if (b < 0 || !wb[a].dynamic || !wb[b].dynamic || ws->st_touch[a] == ws->st_touch[b]) continue;
                int any = 0;
                for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) any |= ws->c[i].valid;
                if (!any) continue;
                ws->st_touch[a] = 1; ws->st_touch[b] = 1; changed = 1;
            }
        }
    }

    /* (k) store, sleep bookkeeping */
    for (int i = 0; i < nb; ++i) {
        if (!wb[i].dynamic) continue;
        quat_to_m3(wb[i].q, &wb[i].R);
        wb[i].t = sub(wb[i].x, m3_mul(&wb[i].R, V(bodies[i].com[0], bodies[i].com[1], bodies[i].com[2])));
        int frozen = 0;
        if (stab) {
            /* frame energy from the velocities that moved the body (mass-normalised, like the sleep energy) */
            float fen = kinetic_energy(&bodies[i], &wb[i], ws->mv[i], ws->mw[i]);
            int ni = ws->st_n[i] > SLHIP_STAB_MAX_INTERACTIONS ? SLHIP_STAB_MAX_INTERACTIONS : ws->st_n[i];
            float thresh = ws->st_touch[i] ? (float)ni * prm->stabilization_threshold : 0.0f;
            float fc = bodies[i].stab[0] + dt;   /* seconds the body has been below the threshold (PhysX counts freezeCount down from the interval) */
            float as = (1.0f - bodies[i].stab[1]) + dt;
            if (as > 1.0f) as = 1.0f;
            int settled = 1;
            if (fen >= thresh) { settled = 0; fc = 0.0f; }
            if (!ws->st_touch[i]) { as = 1.0f; settled = 0; }
            if (settled) {
                float d = 1.0f - SLHIP_STAB_DAMPING * dt;
                wb[i].v = scale(wb[i].v, d);
                wb[i].w = scale(wb[i].w, d);
                as = as * 0.75f + 0.25f * SLHIP_STAB_GRAVITY;
                if (fc >= SLHIP_STAB_FREEZE_INTERVAL && fen < prm->stabilization_threshold * SLHIP_STAB_FREEZE_TOLERANCE) frozen = 1;
            }
            bodies[i].stab[0] = fc;
            bodies[i].stab[1] = 1.0f - as;
        }
        if (frozen) bodies[i].flags |= SLHIP_BODY_FROZEN; else bodies[i].flags &= ~SLHIP_BODY_FROZEN;
        /* the sleep energy: from the velocities the body keeps (damped above) */
        float en = kinetic_energy(&bodies[i], &wb[i], wb[i].v, wb[i].w);
        if (en >= prm->sleep_threshold || (bodies[i].drive_flags & 1u)) bodies[i].wake_counter = prm->wake_time;
        else {
            bodies[i].wake_counter -= dt;
            if (bodies[i].wake_counter <= 0.0f) {
                bodies[i].flags |= SLHIP_BODY_ASLEEP;
                wb[i].v = V(0, 0, 0);
                wb[i].w = V(0, 0, 0);
            }
        }
        if (frozen) {   /* held in place: the step's pose change is taken back (PhysX: body2World = the last transform [ext]), the velocities stay */
            bodies[i].lin_vel[0] = wb[i].v.x; bodies[i].lin_vel[1] = wb[i].v.y; bodies[i].lin_vel[2] = wb[i].v.z;
            bodies[i].ang_vel[0] = wb[i].w.x; bodies[i].ang_vel[1] = wb[i].w.y; bodies[i].ang_vel[2] = wb[i].w.z;
        } else store_body(&bodies[i], &wb[i]);
    }
}

/* redrop (scene.cpp:686-711): bounding-sphere bottom onto the highest top of all others */
static void redrop(const slhip_settle_scene* sc, slhip_body* bodies, int me, const slhip_settle_params* prm)
{
    const int nb = (int)(sc->body_end - sc->body_begin);
    float max_z = 0.0f;
    for (int o = 0; o < nb; ++o) {
        if (o == me || (bodies[o].flags & SLHIP_BODY_STATIC)) continue;
        const float* P = bodies[o].pose;
        const float* c = bodies[o].bbox_center;
        float cz = fmaf(P[10], c[2], fmaf(P[9], c[1], P[8] * c[0])) + P[11];
        float top = cz + c[3];
        if (top > max_z) max_z = top;
    }
    float* P = bodies[me].pose;
    const float* c = bodies[me].bbox_center;
    float off_z = fmaf(P[10], c[2], fmaf(P[9], c[1], P[8] * c[0])) - c[3];
    P[3] = 0.0f; P[7] = 0.0f; P[11] = max_z - off_z;
    bodies[me].stuck_counter = 0;
    for (int k = 0; k < 4; ++k) { bodies[me].lin_vel[k] = 0.0f; bodies[me].ang_vel[k] = 0.0f; bodies[me].stab[k] = 0.0f; }
    bodies[me].flags &= ~SLHIP_BODY_FROZEN;
    /* a teleport invalidates every resting state */
    for (int o = 0; o < nb; ++o) {
        bodies[o].flags &= ~SLHIP_BODY_ASLEEP;
        bodies[o].wake_counter = prm->wake_time;
    }
}

/* optional per-frame trace for the tests: trace[(s * frames + f) * 4 + {0,1,2,3}] = bodies asleep, redrops so far,
   active contacts of the frame's last step, max |v| */
static unsigned* g_caps = NULL; /* per scene [7] (+ steps that dropped body pairs, contacts the solver took in all steps): {steps that dropped contacts, steps that dropped hull pairs, most contacts offered, most hull pairs found,
                                   steps reduced by pair_contact_budget} */
void slref_settle_set_caps(unsigned* c) { g_caps = c; }
static float* g_trace = NULL;
void slref_settle_set_trace(float* t) { g_trace = t; }

/* What a scene keeps between steps -- and, for a caller that steps one long-lived scene through many calls (Scene::simulate,
   ManipulationSim::step, simulateTableTopScene with a visualisation callback: one PxScene in the reference, scene.cpp:720-739,
   903-912, manipulation_sim.cpp:83-93), between calls: slhip_settle_params.resume.  */
typedef struct {
    gjk_seed* cache;
    pmanifold* pm;
    pplane* pp;            /* [bodies of the scene] */
    int step, n_hulls, nb;
    unsigned cap_hits[7];
} scene_keep;
typedef struct { uint32_t n_scenes; scene_keep* k; } settle_state;

void slref_settle_state_free(void* state_)
{
    settle_state* st = (settle_state*)state_;
    if (!st) return;
    for (uint32_t s = 0; s < st->n_scenes; ++s) { free(st->k[s].cache); free(st->k[s].pm); free(st->k[s].pp); }
    free(st->k);
    free(st);
}

/* `state` NULL: the contact state lives for this call only.  Otherwise *state is created by a call with prm->resume == 0 (an old
   one is released) and continued by calls with prm->resume == the steps run so far (checked).  */
int slref_settle_ex(const slhip_settle_scene* scenes, uint32_t n_scenes, slhip_body* bodies,
                    const slhip_hull* hulls, const float* hull_verts, const slhip_settle_params* prm, void** state)
{
    settle_state* st = state ? (settle_state*)*state : NULL;
    if (prm->resume != 0u && (!st || st->n_scenes != n_scenes)) return -3;
    if (prm->resume == 0u) {
        if (st) slref_settle_state_free(st);
        st = (settle_state*)calloc(1, sizeof(settle_state));
        if (!st) return -1;
        st->n_scenes = n_scenes;
        st->k = (scene_keep*)calloc(n_scenes ? n_scenes : 1u, sizeof(scene_keep));
        if (!st->k) { free(st); if (state) *state = NULL; return -1; }
        if (state) *state = st;
    }
    int nb_max = 0;
    for (uint32_t s = 0; s < n_scenes; ++s) {
        const int nb = (int)(scenes[s].body_end - scenes[s].body_begin);
        if (nb > nb_max) nb_max = nb;
    }
    scene_ws* ws = nb_max <= SLHIP_MAX_BODIES ? ws_alloc(cap_pairs(prm), cap_contacts(prm), nb_max, prm->max_body_pairs_per_scene > 65000u ? 65000 : (int)prm->max_body_pairs_per_scene) : NULL;
    int rc = ws ? 0 : (nb_max > SLHIP_MAX_BODIES ? -2 : -1);
    for (uint32_t s = 0; s < n_scenes && rc == 0; ++s) {
        const slhip_settle_scene* sc = &scenes[s];
        const int nb = (int)(sc->body_end - sc->body_begin);
        slhip_body* b = bodies + sc->body_begin;
        scene_keep* K = &st->k[s];
        ws->n_hulls = 0;
        for (int i = 0; i < nb; ++i) { ws->body_lh[i] = ws->n_hulls; ws->n_hulls += (int)(b[i].hull_end - b[i].hull_begin); }
        ws->body_lh[nb] = ws->n_hulls;
        if (prm->resume == 0u) {
            K->step = 1;
            K->n_hulls = ws->n_hulls;
            K->nb = nb;
            K->pp = (pplane*)calloc(nb ? nb : 1, sizeof(pplane));
            if (!K->pp) { rc = -1; break; }
            if (ws->n_hulls > 0) {
                K->cache = (gjk_seed*)calloc((size_t)ws->n_hulls * ws->n_hulls, sizeof(gjk_seed));
                K->pm = (pmanifold*)calloc((size_t)ws->n_hulls * ws->n_hulls, sizeof(pmanifold));
                if (!K->cache || !K->pm) { rc = -1; break; }
            }
        } else if (K->step != (int)prm->resume + 1 || K->n_hulls != ws->n_hulls || K->nb != nb) { rc = -3; break; }
        ws->cache = K->cache;
        ws->pm = K->pm;
        ws->step = K->step;
        memcpy(ws->pp, K->pp, sizeof(pplane) * nb);
        memcpy(ws->cap_hits, K->cap_hits, sizeof(ws->cap_hits));
        ws->n_groups = 0;
        for (uint32_t f = 0; f < prm->frames; ++f) {
            for (uint32_t ss = 0; ss < prm->substeps; ++ss) step_scene(sc, bodies, hulls, hull_verts, prm, ws);
            if (!prm->tabletop) continue;
            int moved = 0;
            for (int i = 0; i < nb; ++i) { /* scene.cpp:742-755 */
                if (b[i].flags & SLHIP_BODY_STATIC) continue;
                if (b[i].pose[11] < prm->redrop_z) { redrop(sc, b, i, prm); moved = 1; }
                else if (b[i].separation < prm->stuck_separation) {
                    if (++b[i].stuck_counter > prm->stuck_frames) {
                        redrop(sc, b, i, prm); moved = 1; }
                } else if (b[i].stuck_counter > 0) b[i].stuck_counter--;
            }
            if (g_trace) {
                float* t = g_trace + ((size_t)s * prm->frames + f) * 4;
                int asleep = 0, active = 0;
                float vmax = 0.0f;
                for (int i = 0; i < nb; ++i) {
                    if (b[i].flags & SLHIP_BODY_ASLEEP) ++asleep;
                    const float v = sqrtf(dot(V(b[i].lin_vel[0], b[i].lin_vel[1], b[i].lin_vel[2]), V(b[i].lin_vel[0], b[i].lin_vel[1], b[i].lin_vel[2])));
                    if (v > vmax) vmax = v;
                }
                for (int g = 0; g < ws->n_groups; ++g)
                    for (int i = ws->g_begin[g]; i < ws->g_end[g]; ++i) active += ws->c[i].valid ? 1 : 0;
                t[0] = (float)asleep; t[1] = (f ? t[1 - 4] : 0.0f) + (moved ? 1.0f : 0.0f); t[2] = (float)active; t[3] = vmax;
            }
        }
        if (g_caps) { for (int q = 0; q < 7; ++q) g_caps[7 * s + q] = ws->cap_hits[q]; }
        K->step = ws->step;
        memcpy(K->pp, ws->pp, sizeof(pplane) * nb);
        memcpy(K->cap_hits, ws->cap_hits, sizeof(ws->cap_hits));
    }
    ws_free(ws);
    if (!state) slref_settle_state_free(st);
    else if (rc != 0 && prm->resume == 0u) { slref_settle_state_free(st); *state = NULL; }
    return rc;
}

int slref_settle(const slhip_settle_scene* scenes, uint32_t n_scenes, slhip_body* bodies,
                 const slhip_hull* hulls, const float* hull_verts, const slhip_settle_params* prm)
{
    return slref_settle_ex(scenes, n_scenes, bodies, hulls, hull_verts, prm, NULL);
}

/* boolean overlap of each body against all others (+ the plane); scene.cpp:355-385 */
int slref_overlap_any(const slhip_settle_scene* scenes, uint32_t n_scenes, const slhip_body* bodies,
                      const slhip_hull* hulls, const float* hull_verts, uint8_t* flags)
{
    for (uint32_t s = 0; s < n_scenes; ++s) {
        const slhip_settle_scene* sc = &scenes[s];
        const int nb = (int)(sc->body_end - sc->body_begin);
        wbody wb[SLHIP_MAX_BODIES];
        if (nb > SLHIP_MAX_BODIES) return -2;
        const slhip_body* b = bodies + sc->body_begin;
        for (int i = 0; i < nb; ++i) load_body(&b[i], &wb[i]);
        for (int i = 0; i < nb; ++i) {
            int hit = 0;
            for (int j = 0; j < nb && !hit; ++j) {
                if (j == i) continue;
                v3 ci = add(m3_mul(&wb[i].R, V(b[i].bsphere[0], b[i].bsphere[1], b[i].bsphere[2])), wb[i].t);
                v3 cj = add(m3_mul(&wb[j].R, V(b[j].bsphere[0], b[j].bsphere[1], b[j].bsphere[2])), wb[j].t);
                v3 d = sub(ci, cj);
                float rr = b[i].bsphere[3] + b[j].bsphere[3];
                if (dot(d, d) > rr * rr) continue;
                for (uint32_t ha = b[i].hull_begin; ha < b[i].hull_end && !hit; ++ha)
                    for (uint32_t hb = b[j].hull_begin; hb < b[j].hull_end && !hit; ++hb) {
                        shape A, B;
                        make_shape(&wb[i], &hulls[ha], hull_verts, &A);
                        make_shape(&wb[j], &hulls[hb], hull_verts, &B);
                        v3 ca = add(m3_mul(&wb[i].R, V(hulls[ha].sphere[0], hulls[ha].sphere[1], hulls[ha].sphere[2])), wb[i].t);
                        v3 cb = add(m3_mul(&wb[j].R, V(hulls[hb].sphere[0], hulls[hb].sphere[1], hulls[hb].sphere[2])), wb[j].t);
                        v3 dd = sub(ca, cb);
                        float r2 = hulls[ha].sphere[3] + hulls[hb].sphere[3];
                        if (dot(dd, dd) > r2 * r2) continue;
                        v3 pa, pb; float dist;
                        if (gjk_distance(&A, &B, dd, 0.0f, &pa, &pb, &dist) == 0) hit = 1;
                    }
            }
            if (!hit && sc->has_plane) {
                for (uint32_t h = b[i].hull_begin; h < b[i].hull_end && !hit; ++h) {
                    const float* vs = hull_verts + 4 * (size_t)hulls[h].vtx_begin;
                    for (uint32_t k = 0; k < hulls[h].vtx_count; ++k) {
                        v3 p = add(m3_mul(&wb[i].R, V(vs[4 * k], vs[4 * k + 1], vs[4 * k + 2])), wb[i].t);
                        if (p.z <= sc->plane_z) { hit = 1; break; }
                    }
                }
            }
            flags[sc->body_begin + i] = (uint8_t)hit;
        }
    }
    return 0;
}

/* debug/test hook: contacts of the FIRST scene for the current state (no stepping).
   out: rows of 12 floats {a, b, pax, pay, paz, nx, ny, nz, sep, pbx, pby, pbz}; returns count */
int slref_debug_contacts(const slhip_settle_scene* sc, const slhip_body* bodies_all, const slhip_hull* hulls,
                         const float* hull_verts, const slhip_settle_params* prm, float* out, int max_rows)
{
    const slhip_body* bodies = bodies_all + sc->body_begin;
    const int nb = (int)(sc->body_end - sc->body_begin);
    scene_ws* ws = ws_alloc(1, 1, nb, 0);
    if (!ws) return 0;
    for (int i = 0; i < nb; ++i) load_body(&bodies[i], &ws->wb[i]);
    int rows = 0;
    for (int i = 0; i < nb; ++i)
        for (int j = i + 1; j < nb; ++j)
            for (uint32_t ha = bodies[i].hull_begin; ha < bodies[i].hull_end; ++ha)
                for (uint32_t hb = bodies[j].hull_begin; hb < bodies[j].hull_end; ++hb) {
                    contact c[PATCH_STRIDE];
                    hull_pair_contacts(bodies, ws->wb, i, j, &hulls[ha], &hulls[hb], hull_verts, prm,
                                       2.0f * prm->contact_offset, c, NULL, NULL, 0);
                    for (int k = 0; k < MAX_CONTACTS_PER_HP; ++k) {
                        if (!c[k].valid || rows >= max_rows) continue;
                        float* o = out + 12 * rows++;
                        v3 pa = add(c[k].ra, ws->wb[i].x), pb = add(c[k].rb, ws->wb[j].x);
                        o[0] = (float)i; o[1] = (float)j; o[2] = pa.x; o[3] = pa.y; o[4] = pa.z;
                        o[5] = c[k].n.x; o[6] = c[k].n.y; o[7] = c[k].n.z; o[8] = c[k].sep;
                        o[9] = pb.x; o[10] = pb.y; o[11] = pb.z;
                    }
                }
    if (sc->has_plane)
        for (int i = 0; i < nb; ++i) {
            contact c[PATCH_STRIDE];
            plane_contacts(bodies, ws->wb, i, hulls, hull_verts, prm, sc->plane_z, prm->contact_offset, c, NULL, 0);
            for (int k = 0; k < PLANE_SLOTS; ++k) {
                if (!c[k].valid || rows >= max_rows) continue;
                float* o = out + 12 * rows++;
                v3 pa = add(c[k].ra, ws->wb[i].x);
                o[0] = (float)i; o[1] = -1.0f; o[2] = pa.x; o[3] = pa.y; o[4] = pa.z;
                o[5] = c[k].n.x; o[6] = c[k].n.y; o[7] = c[k].n.z; o[8] = c[k].sep;
                o[9] = pa.x; o[10] = pa.y; o[11] = sc->plane_z;
            }
        }
    ws_free(ws);
    return rows;
}
