// slhip_hull.cpp -- host geometry of the collision-shape stage (Mesh::loadPhysics, reference src/mesh.cpp:335-470):
// the 3-D convex hull of a point set (quick-hull, double precision) and the solid fill of a voxel grid.  The reference gets both
// from its vendored V-HACD / PhysX cooking; round 5 took them from SciPy (Qhull, ndimage) -- these are the in-tree replacements
// behind the C-ABI, called by stillleben_amd/hulls.py and acd.py.
//
// Quick-hull as in Barber, Dobkin, Huhdanpaa 1996: an initial tetrahedron of extreme points; every face keeps the input points
// strictly above it (farther than eps); the face with the farthest such point is expanded -- the faces that point sees are
// removed, the hole's rim (the horizon: edges between a seen and an unseen face) is joined to the point by new triangles, the
// removed faces' points move to the new faces.  Points within eps of the surface are inside: of a set of coplanar points only
// the corners of their convex polygon become hull vertices (Qhull's merged facets give the same vertex set).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "slhip_common.h"

namespace {

struct P3 { double x, y, z; };
inline P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 cross(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Face {
    int v[3];
    P3 n;            // unit outward normal
    double d;        // n . x = d on the plane
    std::vector<int> outside;
    int far_pt;      // the farthest of them, and its height (kept as points are filed: the search for the next point to add walks the
    double far_h;    // faces, not their points)
    bool alive;
};

struct Hull {
    const P3* p;
    double eps;
    std::vector<Face> faces;

    double height(const Face& f, int i) const { return dot(f.n, p[i]) - f.d; }

    bool make_face(int a, int b, int c, Face& f) const
    {
        f.v[0] = a; f.v[1] = b; f.v[2] = c;
        P3 n = cross(sub(p[b], p[a]), sub(p[c], p[a]));
        const double l = sqrt(dot(n, n));
        if (!(l > 0.0)) return false;
        f.n = {n.x / l, n.y / l, n.z / l};
        f.d = dot(f.n, p[a]);
        f.alive = true;
        f.outside.clear();
        f.far_pt = -1; f.far_h = 0.0;
        return true;
    }
};

// the hull of points[0..n): triangles as index triples into `points`, outward (counter-clockwise seen from outside).
// false: the points do not span a volume (fewer than four, coincident, collinear or coplanar within eps)
bool quick_hull(const P3* pts, int n, std::vector<int>& tris)
{
    tris.clear();
    if (n < 4) return false;
    P3 lo = pts[0], hi = pts[0];
    int ext[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 1; i < n; ++i) {
        if (pts[i].x < lo.x) { lo.x = pts[i].x; ext[0] = i; }
        if (pts[i].x > hi.x) { hi.x = pts[i].x; ext[1] = i; }
        if (pts[i].y < lo.y) { lo.y = pts[i].y; ext[2] = i; }
        if (pts[i].y > hi.y) { hi.y = pts[i].y; ext[3] = i; }
        if (pts[i].z < lo.z) { lo.z = pts[i].z; ext[4] = i; }
        if (pts[i].z > hi.z) { hi.z = pts[i].z; ext[5] = i; }
    }
    const double scale = std::max(std::max(hi.x - lo.x, hi.y - lo.y), hi.z - lo.z);
    if (!(scale > 0.0)) return false;
    Hull H;
    H.p = pts;
    H.eps = 1e-10 * scale;
    // initial tetrahedron: the two extreme points farthest apart, the point farthest from their line, the point farthest from
    // the plane of the three
    int a = ext[0], b = ext[1];
    double best = -1.0;
    for (int i = 0; i < 6; ++i)
        for (int j = i + 1; j < 6; ++j) {
            const P3 d = sub(pts[ext[i]], pts[ext[j]]);
            const double l = dot(d, d);
            if (l > best) { best = l; a = ext[i]; b = ext[j]; }
        }
    if (!(best > 0.0)) return false;
    const P3 ab = sub(pts[b], pts[a]);
    int c = -1;
    best = 0.0;
    for (int i = 0; i < n; ++i) {
        const P3 q = cross(ab, sub(pts[i], pts[a]));
        const double l = dot(q, q);
        if (l > best) { best = l; c = i; }
    }
    if (c < 0 || !(sqrt(best) > H.eps * sqrt(dot(ab, ab)))) return false;
    Face base;
    if (!H.make_face(a, b, c, base)) return false;
    int d = -1;
    best = 0.0;
    for (int i = 0; i < n; ++i) {
        const double h = fabs(H.height(base, i));
        if (h > best) { best = h; d = i; }
    }
    if (d < 0 || !(best > 1e-9 * scale)) return false;
    if (H.height(base, d) > 0.0) std::swap(b, c);   // the base must face away from d
    const int tet[4][3] = {{a, b, c}, {a, d, b}, {b, d, c}, {c, d, a}};
    for (int k = 0; k < 4; ++k) {
        Face f;
        if (!H.make_face(tet[k][0], tet[k][1], tet[k][2], f)) return false;
        H.faces.push_back(f);
    }
    auto assign = [&](int i, size_t first) {
        for (size_t k = first; k < H.faces.size(); ++k) {
            Face& f = H.faces[k];
            if (!f.alive) continue;
            const double h = H.height(f, i);
            if (h > H.eps) {
                f.outside.push_back(i);
                if (h > f.far_h) { f.far_h = h; f.far_pt = i; }      // (ties: the first point filed)
                return;
            }
        }
    };
    for (int i = 0; i < n; ++i)
        if (i != a && i != b && i != c && i != d) assign(i, 0);
    std::unordered_map<uint64_t, int> edges;   // directed edge (u, v) of a seen face -> 1
    std::vector<int> seen, orphans;
    for (;;) {
        // the face whose farthest outside point is the farthest of all (ties: the first face, the first point)
        int fbest = -1, pbest = -1;
        double hbest = 0.0;
        for (size_t k = 0; k < H.faces.size(); ++k) {
            const Face& f = H.faces[k];
            if (f.alive && f.far_pt >= 0 && f.far_h > hbest) { hbest = f.far_h; fbest = (int)k; pbest = f.far_pt; }
        }
        if (fbest < 0) break;
        seen.clear();
        edges.clear();
        orphans.clear();
        for (size_t k = 0; k < H.faces.size(); ++k) {
            Face& f = H.faces[k];
            if (!f.alive || !(H.height(f, pbest) > H.eps)) continue;
            seen.push_back((int)k);
            for (int e = 0; e < 3; ++e) edges[((uint64_t)(uint32_t)f.v[e] << 32) | (uint32_t)f.v[(e + 1) % 3]] = 1;
        }
        const size_t first_new = H.faces.size();
        for (int k : seen) {
            // (copy: push_back below may move the faces)
            const int v0 = H.faces[k].v[0], v1 = H.faces[k].v[1], v2 = H.faces[k].v[2];
            const int vv[3] = {v0, v1, v2};
            for (int e = 0; e < 3; ++e) {
                const int u = vv[e], w = vv[(e + 1) % 3];
                if (edges.count(((uint64_t)(uint32_t)w << 32) | (uint32_t)u)) continue;   // the neighbour across is seen as well
                Face f;
                if (H.make_face(u, w, pbest, f)) H.faces.push_back(f);
            }
        }
        for (int k : seen) {
            Face& f = H.faces[k];
            f.alive = false;
            for (int i : f.outside)
                if (i != pbest) orphans.push_back(i);
            std::vector<int>().swap(f.outside);
        }
        for (int i : orphans) assign(i, first_new);
    }
    for (const Face& f : H.faces)
        if (f.alive) { tris.push_back(f.v[0]); tris.push_back(f.v[1]); tris.push_back(f.v[2]); }
    return tris.size() >= 12;
}

}  // namespace

// Convex hull of n points (xyz, double).  tris_out receives up to tri_capacity index triples into `points` (a hull of n points has
// at most 2 n - 4 triangles), outward oriented; *n_tris_out the number found.  Returns 0; 1 when the points span no volume (within
// 1e-9 of their extent: *n_tris_out = 0, the caller decides what a flat cloud's hull is); -1 on a null / too small output.
extern "C" int slhip_host_convex_hull(const double* points, uint32_t n, uint32_t* tris_out, uint32_t tri_capacity, uint32_t* n_tris_out)
{
    if (!points || !tris_out || !n_tris_out) {
        slhip::set_error("slhip_host_convex_hull: null argument");
        return -1;
    }
    *n_tris_out = 0;
    std::vector<int> tris;
    if (!quick_hull(reinterpret_cast<const P3*>(points), (int)n, tris)) return 1;
    const uint32_t nt = (uint32_t)(tris.size() / 3);
    if (nt > tri_capacity) {
        slhip::set_error("slhip_host_convex_hull: %u triangles do not fit the output (%u)", nt, tri_capacity);
        return -1;
    }
    for (size_t i = 0; i < tris.size(); ++i) tris_out[i] = (uint32_t)tris[i];
    *n_tris_out = nt;
    return 0;
}

// Solid fill of a voxel grid [nx][ny][nz] (bytes, C order; non-zero = occupied): every empty cell that cannot be reached from the
// grid's border through empty face neighbours becomes 1 -- V-HACD's inside / outside classification (a surface with a hole larger
// than a cell stays a shell).  In place.
extern "C" int slhip_host_fill_holes(uint8_t* grid, uint32_t nx, uint32_t ny, uint32_t nz)
{
    if (!grid) {
        slhip::set_error("slhip_host_fill_holes: null grid");
        return -1;
    }
    const size_t N = (size_t)nx * ny * nz;
    if (N == 0) return 0;
    std::vector<uint8_t> out(N, 0);   // 1: reached from outside
    std::vector<uint32_t> stack;
    auto at = [&](uint32_t x, uint32_t y, uint32_t z) { return ((size_t)x * ny + y) * nz + z; };
    auto push = [&](uint32_t x, uint32_t y, uint32_t z) {
        const size_t i = at(x, y, z);
        if (grid[i] || out[i]) return;
        out[i] = 1;
        stack.push_back(x); stack.push_back(y); stack.push_back(z);
    };
    for (uint32_t x = 0; x < nx; ++x)
        for (uint32_t y = 0; y < ny; ++y) { push(x, y, 0); push(x, y, nz - 1); }
    for (uint32_t x = 0; x < nx; ++x)
        for (uint32_t z = 0; z < nz; ++z) { push(x, 0, z); push(x, ny - 1, z); }
    for (uint32_t y = 0; y < ny; ++y)
        for (uint32_t z = 0; z < nz; ++z) { push(0, y, z); push(nx - 1, y, z); }
    while (!stack.empty()) {
        const uint32_t z = stack.back(); stack.pop_back();
        const uint32_t y = stack.back(); stack.pop_back();
        const uint32_t x = stack.back(); stack.pop_back();
        if (x > 0) push(x - 1, y, z);
        if (x + 1 < nx) push(x + 1, y, z);
        if (y > 0) push(x, y - 1, z);
        if (y + 1 < ny) push(x, y + 1, z);
        if (z > 0) push(x, y, z - 1);
        if (z + 1 < nz) push(x, y, z + 1);
    }
    for (size_t i = 0; i < N; ++i)
        if (!grid[i] && !out[i]) grid[i] = 1;
    return 0;
}
