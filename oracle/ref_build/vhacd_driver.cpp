// vhacd_driver.cpp -- ORACLE tooling (ours): runs the reference's vendored V-HACD
// (/root/reference/contrib/v-hacd, compiled where it lies by oracle/ref_build/Makefile) with
// exactly the two parameter sets of Mesh::loadPhysics (reference src/mesh.cpp:351-355, :394-396)
// and applies its selection rule (decomposition iff volume ratio < 0.75, mesh.cpp:426-429).
//
//   vhacd_driver <in.bin> <out.bin>
// in.bin : u32 nV, u32 nT, float[3 nV], u32[3 nT]
// out.bin: u32 nHulls, then per hull: u32 nV, u32 nT, double volume, float[3 nV], u32[3 nT]
#include <cstdint>
#include <cstdio>
#include <vector>

#include "VHACD.h"

static double total_volume(VHACD::IVHACD* v)
{
    double vol = 0.0;
    for (uint32_t i = 0; i < v->GetNConvexHulls(); ++i) {
        VHACD::IVHACD::ConvexHull h;
        v->GetConvexHull(i, h);
        vol += h.m_volume;
    }
    return vol;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s in.bin out.bin [force_single]\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    uint32_t nv = 0, nt = 0;
    if (std::fread(&nv, 4, 1, f) != 1 || std::fread(&nt, 4, 1, f) != 1) return 1;
    std::vector<float> pts(3 * (size_t)nv);
    std::vector<uint32_t> tris(3 * (size_t)nt);
    if (std::fread(pts.data(), 4, pts.size(), f) != pts.size()) return 1;
    if (std::fread(tris.data(), 4, tris.size(), f) != tris.size()) return 1;
    std::fclose(f);
    const bool force_single = argc > 3;

    VHACD::IVHACD* single = VHACD::CreateVHACD();
    {
        VHACD::IVHACD::Parameters p;  // mesh.cpp:351-355
        p.m_concavity = 1.0;
        p.m_asyncACD = false;
        p.m_convexhullApproximation = false;
        p.m_maxConvexHulls = 1;
        single->Compute(pts.data(), nv, tris.data(), nt, p);
    }
    const double v_single = total_volume(single);
    VHACD::IVHACD* source = single;
    VHACD::IVHACD* dec = VHACD::CreateVHACD();
    double v_dec = 0.0;
    if (v_single >= 1e-9 && !force_single) {
        VHACD::IVHACD::Parameters p;  // mesh.cpp:394-396
        p.m_concavity = 0.002;
        p.m_asyncACD = false;
        dec->Compute(pts.data(), nv, tris.data(), nt, p);
        v_dec = total_volume(dec);
        if (v_dec / v_single < 0.75) source = dec;  // mesh.cpp:426-429
    }
    std::fprintf(stderr, "single hull volume %g, decomposition volume %g (%u hulls) -> using %s\n", v_single, v_dec,
                 dec->GetNConvexHulls(), source == dec ? "decomposition" : "single hull");
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 1;
    uint32_t n = source->GetNConvexHulls();
    std::fwrite(&n, 4, 1, o);
    for (uint32_t i = 0; i < n; ++i) {
        VHACD::IVHACD::ConvexHull h;
        source->GetConvexHull(i, h);
        std::fwrite(&h.m_nPoints, 4, 1, o);
        std::fwrite(&h.m_nTriangles, 4, 1, o);
        std::fwrite(&h.m_volume, 8, 1, o);
        std::vector<float> p(3 * (size_t)h.m_nPoints);
        for (size_t k = 0; k < p.size(); ++k) p[k] = (float)h.m_points[k];  // double -> float (mesh.cpp:447-449)
        std::fwrite(p.data(), 4, p.size(), o);
        std::fwrite(h.m_triangles, 4, 3 * (size_t)h.m_nTriangles, o);
    }
    std::fclose(o);
    single->Clean(); single->Release();
    dec->Clean(); dec->Release();
    return 0;
}
