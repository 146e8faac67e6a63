// host_regroup.cpp -- the step right after the fracture event, on the host:
// bind bookkeeping of ApplyFracture (Src/Surtr.cpp:2103-2146), ConvexOutOfSphere (:2415-2458),
// MergeOutOfImpact (:2368-2403) and HandleConvexIsland (:2203-2366).  Inputs are the Convex solids of the event
// (surtr_fragments, un-refitted: the reference regroups before Refitting, :1921-1938) in flat CSR form.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <queue>
#include <set>
#include <vector>

#include "../../include/surtr_hip.h"

namespace {

struct P3 { float x, y, z; };
inline P3 sub(P3 a, P3 b) { return P3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(P3 a, P3 b) { float t = a.x * b.x + a.y * b.y; return t + a.z * b.z; }
inline P3 cross(P3 a, P3 b) { return P3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline P3 unit(P3 a)
{
    const float l = std::sqrt(dot(a, a));
    if (!(l != 0.f)) return P3{0.f, 0.f, 0.f};
    return P3{a.x / l, a.y / l, a.z / l};
}
inline bool right_of(P3 a, P3 b, P3 c, P3 n) { return dot(cross(sub(b, a), sub(c, a)), n) > 0.f; }   // VMACH::OnYourRight

struct SolidView
{
    const float* pos; const uint32_t* off; const int32_t* nbr; uint32_t nv;    // off is local (off[0] = 0 not required: use off[v]-off[0])
    P3 p(int v) const { return P3{pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]}; }
    uint32_t deg(int v) const { return off[v + 1] - off[v]; }
    const int32_t* ring(int v) const { return nbr + off[v]; }
};

// Poly::ExtractFaces (Src/Poly.cpp:89-126) with the visited set keyed by (vertex, neighbour) like the reference.
std::vector<std::vector<int>> faces_of(const SolidView& S)
{
    std::vector<std::vector<int>> out;
    std::set<std::pair<int, int>> seen;
    auto before = [&](int v, int who) {
        const int32_t* r = S.ring(v); const uint32_t n = S.deg(v);
        uint32_t k = 0;
        while (k < n && r[k] != who) ++k;
        return k == 0 ? r[n - 1] : r[k - 1];
    };
    for (int i = 0; i < (int)S.nv; ++i)
        for (uint32_t s = 0; s < S.deg(i); ++s)
        {
            const int adj = S.ring(i)[s];
            if (seen.count({i, adj})) continue;
            std::vector<int> loop(1, i);
            int prev = i, cur = adj;
            while (cur != i && loop.size() <= (size_t)S.nv * 8 + 8)
            {
                seen.insert({prev, cur});
                loop.push_back(cur);
                const int nx = before(cur, prev);
                prev = cur; cur = nx;
            }
            seen.insert({prev, cur});
            out.push_back(loop);
        }
    return out;
}

// Surtr::ConvexOutOfSphere, Src/Surtr.cpp:2415-2458.
bool out_of_sphere(const SolidView& S, const std::vector<std::vector<int>>& faces, uint32_t ns, const float* sphere, P3 origin, float radius)
{
    for (uint32_t v = 0; v < S.nv; ++v)
    {
        const P3 d = sub(origin, S.p((int)v));
        if (std::sqrt(dot(d, d)) < radius) return false;
    }
    for (uint32_t q = 0; q < ns; ++q)
    {
        const P3 po{sphere[3 * q], sphere[3 * q + 1], sphere[3 * q + 2]};
        bool contain = true;
        for (const auto& f : faces)
        {
            const P3 n = unit(cross(sub(S.p(f[1]), S.p(f[0])), sub(S.p(f[2]), S.p(f[0]))));
            const float d = -dot(S.p(f[0]), n);
            if (dot(n, po) + d > 0.f) { contain = false; break; }
        }
        if (contain) return false;
    }
    return true;
}

struct FaceNode { int cid; double absd; P3 n; std::vector<P3> pts; };

} // namespace

extern "C" {

// Poly::Moments (Src/Poly.cpp:55-87): signed volume and centroid from the fan of every face around vertex 0;
// per-face terms in float (Vector3 arithmetic), accumulated in double / float exactly as the reference's members are.
int surtr_moments(uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr, double* volume, float centroid[3])
{
    if (!volume || !centroid || (nv && (!pos || !nbr_off || !nbr))) return SURTR_E_INVALID;
    *volume = 0.0; centroid[0] = centroid[1] = centroid[2] = 0.f;
    if (nv <= 3) return SURTR_OK;
    const SolidView S{pos, nbr_off, nbr, nv};
    const P3 origin = S.p(0);
    double vol = 0.0; P3 c{0.f, 0.f, 0.f};
    for (const auto& f : faces_of(S))
    {
        const size_t n = f.size();
        const P3 p0 = sub(S.p(f[0]), origin);
        for (size_t k = 1; k + 1 < n; ++k)
        {
            const P3 p1 = sub(S.p(f[k]), origin), p2 = sub(S.p(f[(k + 1) % n]), origin);
            const float dV = dot(p0, cross(p1, p2));
            vol += dV;
            c = P3{c.x + (p0.x + p1.x + p2.x) * dV, c.y + (p0.y + p1.y + p2.y) * dV, c.z + (p0.z + p1.z + p2.z) * dV};
        }
    }
    vol /= 6.0;
    const double x = 24.0 * vol;
    const float inv = (float)((x >= 0.0 ? 1.0 : -1.0) / std::max(1.0e-30, std::fabs(x)));     // safeInv (Src/Poly.cpp:33)
    *volume = vol;
    centroid[0] = c.x * inv + origin.x; centroid[1] = c.y * inv + origin.y; centroid[2] = c.z * inv + origin.z;
    return SURTR_OK;
}

int surtr_convex_out_of_sphere(uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr, uint32_t n_sphere,
                               const float* sphere_points, const float origin[3], float radius, int* out)
{
    if (!pos || !nbr_off || !nbr || !origin || !out || (n_sphere && !sphere_points)) return SURTR_E_INVALID;
    SolidView S{pos, nbr_off, nbr, nv};
    *out = out_of_sphere(S, faces_of(S), n_sphere, sphere_points, P3{origin[0], origin[1], origin[2]}, radius) ? 1 : 0;
    return SURTR_OK;
}

int surtr_regroup(uint32_t n_pieces, uint32_t n_outside, const int32_t* piece_cell,
                  const uint32_t* conv_vert_off, const float* conv_pos, const uint32_t* conv_nbr_off, const int32_t* conv_nbr,
                  int partial, uint32_t n_sphere, const float* sphere_points, const float origin[3], float radius,
                  uint32_t* n_compounds, uint32_t* compound_off, int32_t* compound_piece)
{
    if (!conv_vert_off || !conv_pos || !conv_nbr_off || !conv_nbr || !n_compounds || !compound_off || !compound_piece) return SURTR_E_INVALID;
    if (n_outside > n_pieces || (n_pieces > n_outside && !piece_cell)) return SURTR_E_INVALID;
    // bind sets of ApplyFracture: [0] = pieces outside the impact sphere, then one set per cell that produced pieces
    std::vector<std::set<int>> bind(1);
    for (uint32_t p = 0; p < n_outside; ++p) bind[0].insert((int)p);
    for (uint32_t p = n_outside; p < n_pieces; ++p)
    {
        if (p == n_outside || piece_cell[p] != piece_cell[p - 1]) bind.emplace_back();
        bind.back().insert((int)p);
    }
    // SetExtract: faces of every Convex
    std::vector<SolidView> sol(n_pieces);
    std::vector<std::vector<std::vector<int>>> ext(n_pieces);
    for (uint32_t p = 0; p < n_pieces; ++p)
    {
        const uint32_t a = conv_vert_off[p], b = conv_vert_off[p + 1];
        sol[p] = SolidView{conv_pos + 3 * (size_t)a, conv_nbr_off + a, conv_nbr, b - a};
        ext[p] = faces_of(sol[p]);
    }
    if (partial)       // MergeOutOfImpact
    {
        if (!origin || (n_sphere && !sphere_points)) return SURTR_E_INVALID;
        const P3 org{origin[0], origin[1], origin[2]};
        for (size_t i = 1; i < bind.size(); ++i)
        {
            std::set<int> outside;
            for (int c : bind[i])
                if (out_of_sphere(sol[c], ext[c], n_sphere, sphere_points, org, radius)) outside.insert(c);
            if (!outside.empty())
            {
                for (int c : outside) { bind[i].erase(c); bind[0].insert(c); }
            }
        }
        bind.erase(std::remove_if(bind.begin() + 1, bind.end(), [](const std::set<int>& s) { return s.empty(); }), bind.end());
    }
    // HandleConvexIsland: pieces of one compound that do not touch through a pair of opposite, overlapping faces are split
    std::vector<std::set<int>> extra;
    for (auto& local : bind)
    {
        if (local.size() <= 1) continue;
        std::vector<FaceNode> nodes;
        for (int cid : local)
            for (const auto& poly : ext[cid])
            {
                FaceNode fn; fn.cid = cid;
                for (int v : poly) fn.pts.push_back(sol[cid].p(v));
                if (fn.pts.size() < 3) continue;
                const P3 n = unit(cross(sub(fn.pts[0], fn.pts[1]), sub(fn.pts[0], fn.pts[2])));     // Plane(p0,p1,p2)
                const float d = -dot(n, fn.pts[0]);
                fn.n = unit(n);                                                                    // .Normal() re-normalised (:2252-2254)
                fn.absd = (double)std::fabs(d);
                nodes.push_back(std::move(fn));
            }
        std::sort(nodes.begin(), nodes.end(), [](const FaceNode& a, const FaceNode& b) { return a.absd < b.absd; });
        std::vector<std::set<int>> nei_keys;
        std::vector<int> ids(local.begin(), local.end());
        auto slot = [&](int cid) { return (size_t)(std::lower_bound(ids.begin(), ids.end(), cid) - ids.begin()); };
        std::vector<std::set<int>> nei(ids.size());
        for (size_t i = 0; i + 1 < nodes.size(); ++i)
        {
            bool lower = false;
            for (size_t j = i + 1; j < nodes.size(); ++j)
            {
                if (lower && nodes[i].absd > nodes[j].absd) break;
                if (std::fabs(nodes[i].absd - nodes[j].absd) > 1e-3) continue;
                lower = true;
                const P3 in = nodes[i].n, jn = nodes[j].n;
                if (!((double)std::fabs(1.f + dot(in, jn)) < 1e-4)) continue;       // normals must be opposite
                bool touch = false;
                const size_t nj = nodes[j].pts.size(), ni = nodes[i].pts.size();
                for (const P3& ip : nodes[i].pts)
                {
                    bool inside = true;
                    for (size_t v = 0; v < nj; ++v)
                        if (!right_of(nodes[j].pts[v], nodes[j].pts[(v + 1) % nj], ip, jn)) { inside = false; break; }
                    if (inside) { touch = true; break; }
                }
                if (!touch)
                    for (const P3& jp : nodes[j].pts)
                    {
                        bool inside = true;
                        for (size_t v = 0; v < ni; ++v)
                            if (!right_of(nodes[i].pts[v], nodes[i].pts[(v + 1) % ni], jp, in)) { inside = false; break; }
                        if (inside) { touch = true; break; }
                    }
                if (touch) { nei[slot(nodes[i].cid)].insert(nodes[j].cid); nei[slot(nodes[j].cid)].insert(nodes[i].cid); }
            }
        }
        std::set<int> remain(local.begin(), local.end());
        std::vector<std::set<int>> groups;
        while (!remain.empty())
        {
            std::set<int> g;
            std::queue<int> q;
            q.push(*remain.begin());
            while (!q.empty())
            {
                const int cur = q.front(); q.pop();
                if (!remain.count(cur)) continue;
                g.insert(cur); remain.erase(cur);
                for (int a : nei[slot(cur)]) q.push(a);
            }
            groups.push_back(g);
        }
        if (groups.size() >= 2)
        {
            local = groups[0];
            extra.insert(extra.end(), groups.begin() + 1, groups.end());
        }
    }
    bind.insert(bind.end(), extra.begin(), extra.end());
    uint32_t at = 0;
    compound_off[0] = 0;
    for (size_t i = 0; i < bind.size(); ++i)
    {
        for (int c : bind[i]) compound_piece[at++] = c;
        compound_off[i + 1] = at;
    }
    *n_compounds = (uint32_t)bind.size();
    return SURTR_OK;
}

} // extern "C"
