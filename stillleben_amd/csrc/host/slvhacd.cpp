// slvhacd.cpp -- C-ABI shim (ours) over V-HACD, the third-party convex decomposition the reference vendors
// (BSD-3, /root/reference/contrib/v-hacd) and calls from Mesh::loadPhysics (reference src/mesh.cpp:335-470).
// __graft_entry__.build() compiles this file together with the library's own sources, WHERE THEY LIE under
// /root/reference, into stillleben_amd/lib/libslvhacd.so (git-ignored; nothing of V-HACD is copied into the
// repository).  The procedure is the reference's: pass A = one hull {concavity 1.0, maxConvexHulls 1, no hull
// approximation} (mesh.cpp:351-355), pass B = {concavity 0.002} + library defaults (mesh.cpp:394-396), use B iff
// volume(B) / volume(A) < 0.75 (mesh.cpp:426-429); volume(A) < 1e-9 is reported so that the caller can fall back to
// the raw vertices as the reference does (mesh.cpp:373-378).  Both passes run synchronously (m_asyncACD = false).
#include <cstdint>
#include <cstring>
#include <vector>

#include "VHACD.h"

namespace {
struct Result {
    struct HullOut {
        std::vector<float> verts;       // double -> float exactly as mesh.cpp:447-449
        std::vector<uint32_t> tris;
        double volume;
    };
    std::vector<HullOut> hulls;
    double volume_single = 0.0, volume_decomposition = 0.0;
    int used_decomposition = 0, raw_fallback = 0;
};

double total_volume(VHACD::IVHACD* v)
{
    double vol = 0.0;
    for (uint32_t i = 0; i < v->GetNConvexHulls(); ++i) {
        VHACD::IVHACD::ConvexHull h;
        v->GetConvexHull(i, h);
        vol += h.m_volume;
    }
    return vol;
}
}  // namespace

extern "C" {

// Returns an opaque result (free with slvhacd_free) or NULL on failure.
void* slvhacd_decompose(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris, int force_single)
{
    if (!verts || !tris || n_verts == 0 || n_tris == 0) return nullptr;
    Result* res = new Result();
    VHACD::IVHACD* single = VHACD::CreateVHACD();
    VHACD::IVHACD* dec = VHACD::CreateVHACD();
    {
        VHACD::IVHACD::Parameters p;
        p.m_concavity = 1.0;
        p.m_asyncACD = false;
        p.m_convexhullApproximation = false;
        p.m_maxConvexHulls = 1;
        single->Compute(verts, n_verts, tris, n_tris, p);
    }
    res->volume_single = total_volume(single);
    VHACD::IVHACD* source = single;
    if (res->volume_single < 1e-9) {
        res->raw_fallback = 1;
    } else if (!force_single) {
        VHACD::IVHACD::Parameters p;
        p.m_concavity = 0.002;
        p.m_asyncACD = false;
        dec->Compute(verts, n_verts, tris, n_tris, p);
        res->volume_decomposition = total_volume(dec);
        if (res->volume_decomposition / res->volume_single < 0.75) {
            source = dec;
            res->used_decomposition = 1;
        }
    }
    if (!res->raw_fallback) {
        const uint32_t n = source->GetNConvexHulls();
        res->hulls.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            VHACD::IVHACD::ConvexHull h;
            source->GetConvexHull(i, h);
            Result::HullOut& o = res->hulls[i];
            o.volume = h.m_volume;
            o.verts.resize(3 * (size_t)h.m_nPoints);
            for (size_t k = 0; k < o.verts.size(); ++k) o.verts[k] = (float)h.m_points[k];
            o.tris.assign(h.m_triangles, h.m_triangles + 3 * (size_t)h.m_nTriangles);
        }
    }
    single->Clean(); single->Release();
    dec->Clean(); dec->Release();
    return res;
}

void slvhacd_free(void* r) { delete static_cast<Result*>(r); }

// info[0] = n_hulls, [1] = used_decomposition, [2] = raw_fallback; vol[0] = single-hull volume, [1] = decomposition volume
void slvhacd_info(const void* r, uint32_t info[3], double vol[2])
{
    const Result* res = static_cast<const Result*>(r);
    info[0] = (uint32_t)res->hulls.size(); info[1] = (uint32_t)res->used_decomposition; info[2] = (uint32_t)res->raw_fallback;
    vol[0] = res->volume_single; vol[1] = res->volume_decomposition;
}

void slvhacd_hull_size(const void* r, uint32_t i, uint32_t* n_verts, uint32_t* n_tris, double* volume)
{
    const Result::HullOut& h = static_cast<const Result*>(r)->hulls[i];
    *n_verts = (uint32_t)(h.verts.size() / 3);
    *n_tris = (uint32_t)(h.tris.size() / 3);
    *volume = h.volume;
}

void slvhacd_hull_copy(const void* r, uint32_t i, float* verts_out, uint32_t* tris_out)
{
    const Result::HullOut& h = static_cast<const Result*>(r)->hulls[i];
    std::memcpy(verts_out, h.verts.data(), h.verts.size() * sizeof(float));
    std::memcpy(tris_out, h.tris.data(), h.tris.size() * sizeof(uint32_t));
}

}  // extern "C"
