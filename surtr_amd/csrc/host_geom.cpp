// host_geom.cpp -- host-side helpers of the harness that sit either side of the
// GPU path: mesh ingestion (neighbour rings) and fracture-pattern construction
// (bounded Voronoi cells).  Both run once per model / pattern, not per event.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
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
