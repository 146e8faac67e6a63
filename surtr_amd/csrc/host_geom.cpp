// host_geom.cpp -- host-side helpers of the harness that sit either side of the
// GPU path: mesh ingestion (neighbour rings) and fracture-pattern construction
// (bounded Voronoi cells).  Both run once per model / pattern, not per event.
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <map>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/surtr_hip.h"

namespace {

// ---------------------------------------------------------------------------
// Neighbour rings of a welded, closed, consistently wound triangle mesh with
// the ring order AND rotation of Poly::ExtractNeighborFromMesh
// (Src/Poly.cpp:128-263), computed from a directed-edge table instead of the
// reference's per-vertex triangle-fan search.
//   Let t0 = lowest-numbered triangle holding v, s = position of v in t0, and
//   a, b the two vertices after v in t0.  The reference walks the fan starting
//   at t0 and picks the first entry of t0's AdjTriangleVec that holds v; that
//   is the triangle across edge (v,a) when s == 0 and across (v,b) otherwise
//   (:149-174 list adjacent triangles in edge order e0,e1,e2).  After its
//   swap/unique/reverse step (:235-248) the ring is the winding-order cycle
//   a,b,c,... ; started at a when s != 0, and at c (two places later) when s == 0.
int rings_from_triangles(uint32_t nv, uint32_t nt, const int32_t* tris, uint32_t* off, int32_t* nbr)
{
    std::unordered_map<uint64_t, int32_t> third;     // directed edge (x -> y) -> vertex after y in that triangle
    third.reserve((size_t)nt * 4);
    std::vector<int32_t> first_tri(nv, -1);
    for (uint32_t t = 0; t < nt; ++t)
    {
        const int32_t* q = tris + 3 * (size_t)t;
        for (int c = 0; c < 3; ++c)
        {
            const int32_t x = q[c], y = q[(c + 1) % 3], z = q[(c + 2) % 3];
            if (x < 0 || (uint32_t)x >= nv || x == y) return SURTR_E_TOPOLOGY;
            const uint64_t key = ((uint64_t)(uint32_t)x << 32) | (uint32_t)y;
            if (!third.emplace(key, z).second) return SURTR_E_TOPOLOGY;      // non-manifold or inconsistent winding
            if (first_tri[x] < 0) first_tri[x] = (int32_t)t;
        }
    }
    uint32_t at = 0;
    off[0] = 0;
    std::vector<int32_t> ring;
    for (uint32_t v = 0; v < nv; ++v)
    {
        ring.clear();
        if (first_tri[v] >= 0)
        {
            const int32_t* q = tris + 3 * (size_t)first_tri[v];
            int s = 0;
            for (int c = 0; c < 3; ++c) if (q[c] == (int32_t)v) { s = c; break; }
            const int32_t a = q[(s + 1) % 3];
            int32_t cur = a;
            do
            {
                ring.push_back(cur);
                auto it = third.find(((uint64_t)v << 32) | (uint32_t)cur);   // triangle (v, cur, next)
                if (it == third.end()) return SURTR_E_TOPOLOGY;               // open fan
                cur = it->second;
                if (ring.size() > (size_t)nt) return SURTR_E_TOPOLOGY;
            } while (cur != a);
            if (s == 0 && ring.size() >= 3) std::rotate(ring.begin(), ring.begin() + 2, ring.end());
        }
        for (int32_t u : ring) nbr[at++] = u;
        off[v + 1] = at;
    }
    // symmetric links (:253-260)
    for (uint32_t v = 0; v < nv; ++v)
        for (uint32_t j = off[v]; j < off[v + 1]; ++j)
        {
            const int32_t u = nbr[j];
            bool back = false;
            for (uint32_t k = off[u]; k < off[u + 1]; ++k) if (nbr[k] == (int32_t)v) { back = true; break; }
            if (!back) return SURTR_E_TOPOLOGY;
        }
    return SURTR_OK;
}

// ---------------------------------------------------------------------------
// Bounded Voronoi cell as a convex polyhedron with neighbour rings, clipped in
// double by bisector half-spaces (voro++ also works in double,
// Src/Surtr.cpp:2007-2040).  Canonical face order: DESIGN.md "Voronoi cells".
struct D3 { double x, y, z; };

struct ConvexCell
{
    std::vector<D3> p;
    std::vector<std::vector<int>> ring;   // CCW seen from outside
};

ConvexCell unit_cube()
{
    static const double P[8][3] = {{-.5, -.5, -.5}, {.5, -.5, -.5}, {.5, .5, -.5}, {-.5, .5, -.5},
                                   {-.5, -.5, .5},  {.5, -.5, .5},  {.5, .5, .5},  {-.5, .5, .5}};
    static const int NB[8][3] = {{1, 4, 3}, {5, 0, 2}, {3, 6, 1}, {7, 2, 0}, {5, 7, 0}, {1, 6, 4}, {5, 2, 7}, {4, 6, 3}};
    ConvexCell c;
    for (int i = 0; i < 8; ++i) { c.p.push_back(D3{P[i][0], P[i][1], P[i][2]}); c.ring.emplace_back(NB[i], NB[i] + 3); }
    return c;
}

inline int ring_prev(const std::vector<int>& r, int who)
{
    size_t k = 0;
    while (k < r.size() && r[k] != who) ++k;
    return k == 0 ? r.back() : r[k - 1];
}

// keep n.x <= c; returns false when nothing is left
bool cut_cell(ConvexCell& c, D3 n, double cc)
{
    const int n0 = (int)c.p.size();
    std::vector<double> s(n0);
    bool any_out = false, any_in = false;
    for (int i = 0; i < n0; ++i)
    {
        s[i] = n.x * c.p[i].x + n.y * c.p[i].y + n.z * c.p[i].z - cc;
        if (s[i] > 0) any_out = true; else any_in = true;
    }
    if (!any_out) return true;
    if (!any_in) { c.p.clear(); c.ring.clear(); return false; }
    std::vector<char> out(n0);
    for (int i = 0; i < n0; ++i) out[i] = s[i] > 0;
    // new vertices on crossing edges
    for (int i = 0; i < n0; ++i)
    {
        if (!out[i]) continue;
        for (size_t j = 0; j < c.ring[i].size(); ++j)
        {
            const int k = c.ring[i][j];
            if (k >= n0 || out[k]) continue;
            const double t = s[k] / (s[k] - s[i]);            // from the kept end towards the cut end
            const D3 a = c.p[k], b = c.p[i];
            const int fresh = (int)c.p.size();
            c.p.push_back(D3{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)});
            c.ring.push_back({i, k});
            for (int& e : c.ring[k]) if (e == i) { e = fresh; break; }
            c.ring[i][j] = fresh;
        }
    }
    const int n1 = (int)c.p.size();
    // cap: successor of every new vertex along the face through its cut end
    std::vector<int> succ(n1, -1), pred(n1, -1);
    for (int x = n0; x < n1; ++x)
    {
        int prev = x, cur = c.ring[x][0], guard = 0;
        while (cur < n0 && out[cur] && guard++ < n1)
        {
            const int nx = ring_prev(c.ring[cur], prev);
            prev = cur; cur = nx;
        }
        succ[x] = cur;
        if (cur >= n0) pred[cur] = x;
    }
    for (int x = n0; x < n1; ++x)
    {
        const int kept = c.ring[x][1];
        c.ring[x] = {pred[x], succ[x], kept};
    }
    // compaction
    std::vector<int> id(n1, -1);
    int live = 0;
    for (int i = 0; i < n1; ++i) if (i >= n0 || !out[i]) id[i] = live++;
    ConvexCell d;
    d.p.resize(live); d.ring.resize(live);
    for (int i = 0; i < n1; ++i)
    {
        if (id[i] < 0) continue;
        d.p[id[i]] = c.p[i];
        for (int e : c.ring[i]) d.ring[id[i]].push_back(e >= 0 ? id[e] : -1);
    }
    c = std::move(d);
    return c.p.size() >= 4;
}

struct CellFace { int gen; std::vector<D3> loop; };

std::vector<CellFace> build_cell(const std::vector<D3>& seeds, int self)
{
    const int C = (int)seeds.size();
    ConvexCell cell = unit_cube();
    const D3 s = seeds[self];
    struct Gen { int id; D3 n; double c; };
    std::vector<Gen> gens;
    gens.push_back({C + 0, D3{-1, 0, 0}, 0.5}); gens.push_back({C + 1, D3{1, 0, 0}, 0.5});
    gens.push_back({C + 2, D3{0, -1, 0}, 0.5}); gens.push_back({C + 3, D3{0, 1, 0}, 0.5});
    gens.push_back({C + 4, D3{0, 0, -1}, 0.5}); gens.push_back({C + 5, D3{0, 0, 1}, 0.5});
    for (int o = 0; o < C; ++o)
    {
        if (o == self) continue;
        const D3 q = seeds[o];
        const D3 n{q.x - s.x, q.y - s.y, q.z - s.z};
        const double cc = 0.5 * ((q.x * q.x + q.y * q.y + q.z * q.z) - (s.x * s.x + s.y * s.y + s.z * s.z));
        const size_t before = cell.p.size();
        bool touched = false;
        for (const D3& p : cell.p) if (n.x * p.x + n.y * p.y + n.z * p.z - cc > 0) { touched = true; break; }
        if (!touched) continue;
        cut_cell(cell, n, cc);
        (void)before;
        gens.push_back({o, n, cc});
    }
    // faces: every directed edge belongs to one loop
    std::vector<CellFace> faces;
    const int nvert = (int)cell.p.size();
    std::vector<std::vector<char>> seen(nvert);
    for (int i = 0; i < nvert; ++i) seen[i].assign(cell.ring[i].size(), 0);
    for (int i = 0; i < nvert; ++i)
        for (size_t j = 0; j < cell.ring[i].size(); ++j)
        {
            if (seen[i][j]) continue;
            std::vector<int> loop;
            int prev = i, cur = cell.ring[i][j];
            seen[i][j] = 1; loop.push_back(i);
            while (cur != i && (int)loop.size() <= nvert)
            {
                loop.push_back(cur);
                const int nx = ring_prev(cell.ring[cur], prev);
                for (size_t q = 0; q < cell.ring[cur].size(); ++q) if (cell.ring[cur][q] == nx) seen[cur][q] = 1;
                prev = cur; cur = nx;
            }
            // generator = the plane all loop vertices lie on
            int best = -1; double bestErr = 1e300;
            for (const Gen& g : gens)
            {
                const double nl = std::sqrt(g.n.x * g.n.x + g.n.y * g.n.y + g.n.z * g.n.z);
                double worst = 0;
                for (int v : loop)
                    worst = std::max(worst, std::fabs(g.n.x * cell.p[v].x + g.n.y * cell.p[v].y + g.n.z * cell.p[v].z - g.c) / nl);
                if (worst < bestErr) { bestErr = worst; best = g.id; }
            }
            CellFace f; f.gen = best;
            for (int v : loop) f.loop.push_back(cell.p[v]);
            faces.push_back(std::move(f));
        }
    for (auto& f : faces)
    {
        // outward winding: (v1-v0)x(v2-v0) must point away from the seed
        const D3 a = f.loop[0], b = f.loop[1], c = f.loop[2];
        const D3 u{b.x - a.x, b.y - a.y, b.z - a.z}, w{c.x - a.x, c.y - a.y, c.z - a.z};
        const D3 n{u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x};
        if (n.x * (a.x - s.x) + n.y * (a.y - s.y) + n.z * (a.z - s.z) < 0) std::reverse(f.loop.begin(), f.loop.end());
        size_t st = 0;
        for (size_t i = 1; i < f.loop.size(); ++i)
        {
            const D3& p = f.loop[i]; const D3& q = f.loop[st];
            if (p.x < q.x || (p.x == q.x && (p.y < q.y || (p.y == q.y && p.z < q.z)))) st = i;
        }
        std::rotate(f.loop.begin(), f.loop.begin() + st, f.loop.end());
    }
    std::stable_sort(faces.begin(), faces.end(), [](const CellFace& x, const CellFace& y) { return x.gen < y.gen; });
    return faces;
}

} // namespace

extern "C" {

int surtr_neighbors_from_mesh(uint32_t nv, uint32_t n_tris, const int32_t* tris, uint32_t* nbr_off, int32_t* nbr)
{
    if (!tris || !nbr_off || !nbr) return SURTR_E_INVALID;
    return rings_from_triangles(nv, n_tris, tris, nbr_off, nbr);
}

int surtr_voronoi_cells(uint32_t n, const double* seeds, uint32_t* n_faces, uint32_t* n_face_verts,
                        uint32_t* cell_face_off, int32_t* face_gen, uint32_t* face_vert_off, double* verts)
{
    if (!seeds || n == 0 || !n_faces || !n_face_verts) return SURTR_E_INVALID;
    std::vector<D3> S(n);
    for (uint32_t i = 0; i < n; ++i) S[i] = D3{seeds[3 * i], seeds[3 * i + 1], seeds[3 * i + 2]};
    uint32_t nf = 0, nfv = 0;
    const bool fill = cell_face_off && face_gen && face_vert_off && verts;
    if (fill) { cell_face_off[0] = 0; face_vert_off[0] = 0; }
    for (uint32_t c = 0; c < n; ++c)
    {
        std::vector<CellFace> fs = build_cell(S, (int)c);
        for (const CellFace& f : fs)
        {
            if (fill)
            {
                face_gen[nf] = f.gen;
                for (const D3& p : f.loop) { verts[3 * (size_t)nfv] = p.x; verts[3 * (size_t)nfv + 1] = p.y; verts[3 * (size_t)nfv + 2] = p.z; ++nfv; }
                face_vert_off[nf + 1] = nfv;
            }
            else nfv += (uint32_t)f.loop.size();
            ++nf;
        }
        if (fill) cell_face_off[c + 1] = nf;
    }
    *n_faces = nf; *n_face_verts = nfv;
    return SURTR_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------
// Limited greedy convex hull -> unit face normals (VMACH::ConvexHull with a point limit,
// Src/VMACH.cpp:869-1161, + Surtr::GenerateICHNormal, Src/Surtr.cpp:1961-1974), index-based: faces and
// edges live in vectors in creation order with tombstones instead of std::list nodes.  Edges are keyed like the
// reference's Key2Edge (:941-948): by the "%f" text of their end points, so two points that print alike share a key.
namespace {

struct F3 { float x, y, z; };
inline F3 f3sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline bool f3eq(F3 a, F3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct LimitedHull
{
    struct Face { F3 v[3]; bool visible = false, dead = false; };
    struct Edge { F3 e[2]; int f1 = -1, f2 = -1; bool remove = false, dead = false; std::string key; };
    std::vector<F3> pts; std::vector<char> done; std::vector<float> gain;
    std::vector<Face> faces; std::vector<Edge> edges;
    std::unordered_map<std::string, int> emap;
    std::vector<int> added, visible;

    static float vol(const Face& f, F3 p)                             // :922-939
    {
        const float ax = f.v[0].x - p.x, ay = f.v[0].y - p.y, az = f.v[0].z - p.z;
        const float bx = f.v[1].x - p.x, by = f.v[1].y - p.y, bz = f.v[1].z - p.z;
        const float cx = f.v[2].x - p.x, cy = f.v[2].y - p.y, cz = f.v[2].z - p.z;
        return ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
    }
    static std::string text(F3 p) { return std::to_string(p.x) + std::to_string(p.y) + std::to_string(p.z); }
    static std::string key(F3 a, F3 b) { std::string s = text(a), t = text(b); return s < t ? s + "|" + t : t + "|" + s; }
    void edge(F3 a, F3 b, int face)                                   // CreateEdge :971-983 + LinkFace
    {
        const std::string k = key(a, b);
        auto it = emap.find(k);
        if (it == emap.end())
        {
            Edge e; e.e[0] = a; e.e[1] = b; e.key = k;
            edges.push_back(e);
            it = emap.emplace(k, (int)edges.size() - 1).first;
        }
        Edge& e = edges[it->second];
        if (e.f1 >= 0 && e.f2 >= 0) return;
        (e.f1 < 0 ? e.f1 : e.f2) = face;
    }
    void face(F3 a, F3 b, F3 c, F3 inner)                             // CreateFace :955-969
    {
        Face f; f.v[0] = a; f.v[1] = b; f.v[2] = c;
        if (vol(f, inner) < 0.f) std::swap(f.v[0], f.v[2]);
        faces.push_back(f);
        const int id = (int)faces.size() - 1;
        added.push_back(id);
        edge(a, b, id); edge(a, c, id); edge(b, c, id);
    }
    bool first()                                                      // BuildFirstHull :1036-1085
    {
        const size_t n = pts.size();
        if (n <= 3) return false;
        size_t i1 = 0;
        for (size_t i = 1; i < n; ++i) if (pts[i1].x < pts[i].x) i1 = i;
        auto dist = [&](F3 a) {
            const double dx = (double)(a.x - pts[i1].x), dy = (double)(a.y - pts[i1].y), dz = (double)(a.z - pts[i1].z);
            return std::sqrt(dx * dx + dy * dy + dz * dz);
        };
        size_t i2 = 0;
        for (size_t i = 1; i < n; ++i) if (dist(pts[i2]) < dist(pts[i])) i2 = i;
        auto area = [&](F3 c) {
            const F3 u = f3sub(pts[i2], pts[i1]), w = f3sub(c, pts[i1]);
            const float kx = u.y * w.z - u.z * w.y, ky = u.z * w.x - u.x * w.z, kz = u.x * w.y - u.y * w.x;
            float t = kx * kx + ky * ky; t = t + kz * kz;
            return 0.5f * std::sqrt(t);
        };
        size_t i3 = 0;
        for (size_t i = 1; i < n; ++i) if (area(pts[i3]) < area(pts[i])) i3 = i;
        Face base; base.v[0] = pts[i1]; base.v[1] = pts[i2]; base.v[2] = pts[i3];
        size_t i4 = 0;
        for (size_t i = 1; i < n; ++i) if (vol(base, pts[i4]) < vol(base, pts[i])) i4 = i;
        done[i1] = done[i2] = done[i3] = done[i4] = 1;
        face(pts[i1], pts[i2], pts[i3], pts[i4]);
        face(pts[i1], pts[i2], pts[i4], pts[i3]);
        face(pts[i1], pts[i3], pts[i4], pts[i2]);
        face(pts[i2], pts[i3], pts[i4], pts[i1]);
        return true;
    }
    void add_point(F3 p)                                              // AddPointToHull :994-1034
    {
        bool any = false;
        for (size_t f = 0; f < faces.size(); ++f)
            if (!faces[f].dead && vol(faces[f], p) < 0.f) { faces[f].visible = true; visible.push_back((int)f); any = true; }
        if (!any) return;
        for (size_t i = 0; i < edges.size(); ++i)                      // edges appended below are visited too, like the list
        {
            if (edges[i].dead) continue;
            int f1 = edges[i].f1, f2 = edges[i].f2;
            if (f1 < 0 || f2 < 0) continue;
            const bool v1 = faces[f1].visible, v2 = faces[f2].visible;
            if (v1 && v2) { edges[i].remove = true; continue; }
            if (!(v1 || v2)) continue;
            if (v1) { std::swap(f1, f2); edges[i].f1 = f1; edges[i].f2 = f2; }
            F3 inner = faces[f2].v[0];
            for (int q = 0; q < 3; ++q)
                if (!f3eq(faces[f2].v[q], edges[i].e[0]) && !f3eq(faces[f2].v[q], edges[i].e[1])) { inner = faces[f2].v[q]; break; }
            if (edges[i].f1 == f2) edges[i].f1 = -1; else edges[i].f2 = -1;          // EraseFace(face2)
            const F3 a = edges[i].e[0], b = edges[i].e[1];
            face(a, b, p, inner);                                      // may grow `edges`
        }
    }
    void cleanup()                                                    // :1140-1161
    {
        visible.clear(); added.clear();
        for (auto& e : edges) if (!e.dead && e.remove) { emap.erase(e.key); e.dead = true; }
        for (auto& f : faces) if (!f.dead && f.visible) f.dead = true;
    }
    void build(uint32_t limit)                                        // CreateConvexHull :1087-1138
    {
        done.assign(pts.size(), 0); gain.assign(pts.size(), 0.f);
        if (!first()) return;
        uint32_t used = 4;
        for (size_t i = 0; i < pts.size(); ++i)
        {
            if (done[i]) continue;
            for (const Face& f : faces) gain[i] += std::max(0.0f, vol(f, pts[i]));
        }
        if (limit == 0) limit = (uint32_t)pts.size();
        while (used < limit)
        {
            size_t k = 0;
            for (size_t i = 1; i < gain.size(); ++i) if (gain[k] < gain[i]) k = i;
            add_point(pts[k]);
            done[k] = 1; gain[k] = -3.402823466e+38f; ++used;
            for (size_t i = 0; i < pts.size(); ++i)
            {
                if (done[i]) continue;
                float gone = 0.f, came = 0.f;
                for (int f : visible) gone += std::max(0.0f, vol(faces[f], pts[i]));
                for (int f : added) came += std::max(0.0f, vol(faces[f], pts[i]));
                gain[i] -= gone; gain[i] += came;
            }
            cleanup();
        }
    }
};

} // namespace

extern "C" {

int surtr_hull_normals(uint32_t n, const float* points, uint32_t limit, uint32_t capacity, float* normals, uint32_t* count)
{
    if (!points || !count) return SURTR_E_INVALID;
    LimitedHull h;
    h.pts.resize(n);
    for (uint32_t i = 0; i < n; ++i) h.pts[i] = F3{points[3 * i], points[3 * i + 1], points[3 * i + 2]};
    h.build(limit);
    uint32_t m = 0;
    for (const auto& f : h.faces)
    {
        if (f.dead) continue;
        if (normals && m < capacity)
        {
            const F3 u = f3sub(f.v[1], f.v[0]), w = f3sub(f.v[2], f.v[0]);
            float nx = u.y * w.z - u.z * w.y, ny = u.z * w.x - u.x * w.z, nz = u.x * w.y - u.y * w.x;
            float t = nx * nx + ny * ny; t = t + nz * nz;
            const float len = std::sqrt(t);
            if (len != 0.f) { nx = nx / len; ny = ny / len; nz = nz / len; } else { nx = ny = nz = 0.f; }
            normals[3 * m] = nx; normals[3 * m + 1] = ny; normals[3 * m + 2] = nz;
        }
        ++m;
    }
    *count = m;
    return (normals && m > capacity) ? SURTR_E_CAPACITY : SURTR_OK;
}

// Kdop::KdopContainer::Calc(vertices, maxAxisScale, planeGapInv) (Src/Kdop.cpp:15-51) + the plane order of
// ClipWithPolyhedron (:166-179): for every normal the Min plane then the Max plane, pushed out by the gap.
int surtr_kdop_ach_planes(uint32_t n, const float* points, uint32_t k, const float* normals, double max_axis_scale,
                          float plane_gap_inv, float* planes /* 8 floats per normal */)
{
    if (!points || !normals || !planes || n == 0) return SURTR_E_INVALID;
    auto dot = [](const float* a, const float* b) { float t = a[0] * b[0] + a[1] * b[1]; return t + a[2] * b[2]; };
    for (uint32_t j = 0; j < k; ++j)
    {
        const float* nr = normals + 3 * j;
        double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
        uint32_t vlo = 0, vhi = 0;
        for (uint32_t v = 0; v < n; ++v)
        {
            const float t = dot(points + 3 * v, nr);
            if (lo > t) { lo = t; vlo = v; }
            if (hi < t) { hi = t; vhi = v; }
        }
        const float gap = (float)(max_axis_scale / plane_gap_inv);
        for (int side = 0; side < 2; ++side)
        {
            float nn[3] = {side ? nr[0] : -nr[0], side ? nr[1] : -nr[1], side ? nr[2] : -nr[2]};
            float t = nn[0] * nn[0] + nn[1] * nn[1]; t = t + nn[2] * nn[2];
            const float len = std::sqrt(t);
            if (len != 0.f) { nn[0] = nn[0] / len; nn[1] = nn[1] / len; nn[2] = nn[2] / len; } else { nn[0] = nn[1] = nn[2] = 0.f; }
            const float* pv = points + 3 * (size_t)(side ? vhi : vlo);
            const float q[3] = {pv[0] + nn[0] * gap, pv[1] + nn[1] * gap, pv[2] + nn[2] * gap};
            float* out = planes + 8 * (size_t)j + 4 * side;
            out[0] = nn[0]; out[1] = nn[1]; out[2] = nn[2]; out[3] = -dot(q, nn);
        }
    }
    return SURTR_OK;
}

// Plane(p0, p1, p2) of SimpleMath (SimpleMath.inl:2773-2780, XMPlaneFromPoints): normal = normalize((p0-p1) x (p0-p2)),
// d = -dot(normal, p0) -- the arithmetic of k_place_cells, for PolygonFace::ConstructFacePlane (Src/VMACH.cpp:302-310) in the
// C++ host layer.
int surtr_plane_from_points(const float p0[3], const float p1[3], const float p2[3], float plane[4])
{
    if (!p0 || !p1 || !p2 || !plane) return SURTR_E_INVALID;
    const float ax = p0[0] - p1[0], ay = p0[1] - p1[1], az = p0[2] - p1[2];
    const float bx = p0[0] - p2[0], by = p0[1] - p2[1], bz = p0[2] - p2[2];
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    float t = nx * nx + ny * ny; t = t + nz * nz;
    const float len = std::sqrt(t);
    if (len != 0.f) { nx = nx / len; ny = ny / len; nz = nz / len; } else { nx = 0.f; ny = 0.f; nz = 0.f; }
    float d = nx * p0[0] + ny * p0[1]; d = d + nz * p0[2];
    plane[0] = nx; plane[1] = ny; plane[2] = nz; plane[3] = -d;
    return SURTR_OK;
}

// Kdop::KdopContainer::Calc(const Poly::Polyhedron&) (Src/Kdop.cpp:92-115): first minimum / first maximum of n.v over the
// vertices in order; MinPlane = Plane(vert, -n), MaxPlane = Plane(vert, n) (point-normal form, not normalised:
// SimpleMath.inl:2782-2788).  What k_refit computes on the device for every fragment, here for the Kdop API surface.
int surtr_kdop_planes(uint32_t n, const float* points, uint32_t k, const float* normals, float* planes /* 8 floats per normal */)
{
    if (!points || !normals || !planes || n == 0) return SURTR_E_INVALID;
    auto dot = [](const float* a, const float* b) { float t = a[0] * b[0] + a[1] * b[1]; return t + a[2] * b[2]; };
    for (uint32_t j = 0; j < k; ++j)
    {
        const float* nr = normals + 3 * j;
        double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308;
        uint32_t vlo = 0, vhi = 0;
        for (uint32_t v = 0; v < n; ++v)
        {
            const float t = dot(points + 3 * v, nr);
            if (lo > t) { lo = t; vlo = v; }
            if (hi < t) { hi = t; vhi = v; }
        }
        for (int side = 0; side < 2; ++side)
        {
            const float nn[3] = {side ? nr[0] : -nr[0], side ? nr[1] : -nr[1], side ? nr[2] : -nr[2]};
            const float* pv = points + 3 * (size_t)(side ? vhi : vlo);
            float* out = planes + 8 * (size_t)j + 4 * side;
            out[0] = nn[0]; out[1] = nn[1]; out[2] = nn[2]; out[3] = -dot(pv, nn);
        }
    }
    return SURTR_OK;
}

// ---- mesh files either side of the path ----------------------------------------------
// Surtr::LoadModelData (Src/Surtr.cpp:2683-2727) reads through assimp with Triangulate | FlipWindingOrder |
// JoinIdenticalVertices, negates x and applies scale/translate.  assimp is not in this image; this reader restates
// those conventions for Wavefront OBJ (v / f records, negative indices, "i/j/k" corners): one vertex per distinct
// position in order of first use by a face, polygons as fans, every triangle reversed, x negated.  How assimp
// triangulates faces of more than four corners is not reproduced (unpinned): feed triangle meshes for exact results.
int surtr_read_obj(const char* path, const float scale[3], const float translate[3], uint32_t cap_verts, uint32_t cap_tris,
                   float* pos, int32_t* tris, uint32_t* n_verts, uint32_t* n_tris)
{
    if (!path || !scale || !translate || !n_verts || !n_tris) return SURTR_E_INVALID;
    FILE* f = fopen(path, "r");
    if (!f) return SURTR_E_INVALID;
    std::vector<std::array<float, 3>> raw;
    std::vector<int> corner;                 // raw vertex index per triangle corner
    char line[4096];
    while (fgets(line, sizeof line, f))
    {
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t'))
        {
            float x = 0, y = 0, z = 0;
            if (sscanf(line + 1, "%f %f %f", &x, &y, &z) == 3) raw.push_back({x, y, z});
        }
        else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t'))
        {
            std::vector<int> poly;
            for (char* tok = strtok(line + 1, " \t\r\n"); tok; tok = strtok(nullptr, " \t\r\n"))
            {
                const long i = strtol(tok, nullptr, 10);
                if (i == 0) continue;
                poly.push_back(i > 0 ? (int)i - 1 : (int)raw.size() + (int)i);
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k) { corner.push_back(poly[0]); corner.push_back(poly[k]); corner.push_back(poly[k + 1]); }
        }
    }
    fclose(f);
    std::map<std::array<float, 3>, int> id;
    std::vector<int> first;                  // joined vertex -> raw index
    std::vector<int> out(corner.size());
    for (size_t c = 0; c < corner.size(); ++c)
    {
        if (corner[c] < 0 || corner[c] >= (int)raw.size()) return SURTR_E_INVALID;
        auto it = id.find(raw[corner[c]]);
        if (it == id.end()) { it = id.emplace(raw[corner[c]], (int)first.size()).first; first.push_back(corner[c]); }
        out[c] = it->second;
    }
    *n_verts = (uint32_t)first.size(); *n_tris = (uint32_t)(out.size() / 3);
    if (!pos || !tris) return SURTR_OK;
    if (cap_verts < *n_verts || cap_tris < *n_tris) return SURTR_E_CAPACITY;
    for (size_t v = 0; v < first.size(); ++v)
    {
        const auto& p = raw[first[v]];
        pos[3 * v] = -p[0] * scale[0] + translate[0]; pos[3 * v + 1] = p[1] * scale[1] + translate[1]; pos[3 * v + 2] = p[2] * scale[2] + translate[2];
    }
    for (size_t t = 0; t < out.size() / 3; ++t) { tris[3 * t] = out[3 * t + 2]; tris[3 * t + 1] = out[3 * t + 1]; tris[3 * t + 2] = out[3 * t]; }
    return SURTR_OK;
}

// Fragment writer for the headless harness: the render buffers of an event (36-byte VertexNormalColor + uint32 indices,
// what DynamicMesh::UpdateMeshData consumes, Inc/Mesh.h:163-207) as one OBJ object per fragment.
int surtr_write_obj(const char* path, uint32_t n_frag, const int32_t* frag_ids, const uint32_t* mesh_vert_off, const float* vnc,
                    const uint32_t* idx_off, const uint32_t* idx)
{
    if (!path || (n_frag && (!frag_ids || !mesh_vert_off || !vnc || !idx_off || !idx))) return SURTR_E_INVALID;
    FILE* f = fopen(path, "w");
    if (!f) return SURTR_E_INVALID;
    for (uint32_t k = 0; k < n_frag; ++k)
    {
        fprintf(f, "o cell%d_piece%d_island%d\n", frag_ids[3 * k], frag_ids[3 * k + 1], frag_ids[3 * k + 2]);
        for (uint32_t v = mesh_vert_off[k]; v < mesh_vert_off[k + 1]; ++v) fprintf(f, "v %.9g %.9g %.9g\n", vnc[9 * (size_t)v], vnc[9 * (size_t)v + 1], vnc[9 * (size_t)v + 2]);
        const uint32_t base = mesh_vert_off[k] + 1u;
        for (uint32_t i = idx_off[k]; i + 2 < idx_off[k + 1]; i += 3) fprintf(f, "f %u %u %u\n", base + idx[i], base + idx[i + 1], base + idx[i + 2]);
    }
    fclose(f);
    return SURTR_OK;
}

} // extern "C"
