// surtr_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the per-fracture-event decomposition path of W298/Surtr,
// written from scratch on struct-of-vectors solids.  It is the parity checker
// for the HIP engine in surtr_amd/ and the timed "port" CPU baseline of
// bench.py.  Nothing under surtr_amd/ may include, link or call this file:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// UNDEFINED BEHAVIOUR OF THE REFERENCE: where ClipPolyhedron's compaction stores a link that names no vertex (an ID of -1 or
// past the new size, Src/Poly.cpp:484-493) the reference goes on to index outside its arrays.  There is nothing to restate
// there: clip() counts the link (orc_links_off_the_array), stops with an empty solid, and the tests accept the engine's flag
// for that pair or fragment instead of comparing results.
//
// PINNING STATUS: the reference ships no tests, fixtures or golden vectors
//  and cannot be built in this image (every translation
// unit includes Inc/pch.h -> Windows.h, d3d12.h, DirectXMath.h, none present;
// no stand-ins are written).  The oracle is therefore pinned only against the
// known answers that SURVEY.md section 6 / 8(c) records from the survey's run of the
// reference kernels (tests/test_oracle_kat.py) -- full-vector parity is
// "parity unpinned" beyond those.
//
// Float semantics restated here (SimpleMath over DirectXMath, SSE2 path,
// ThirdParty/Inc/SimpleMath.inl:729-1006, 2773-2788; DirectXMath itself is not
// in the reference tree): every Vector3 op is float32, dot = (x*x'+y*y')+z*z',
// Vector3/float multiplies by 1.f/S (SimpleMath.inl:870-878), Normalize
// divides by sqrt(dot), Plane(p0,p1,p2) is normalised, Plane(point,normal) is
// not.  Build with -ffp-contract=off and without fast-math.
//
// Every function cites the reference file:line (relative to /root/reference)
// whose behaviour it follows.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <random>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- math -----
struct V3 { float x = 0.f, y = 0.f, z = 0.f; };
struct Plane { float x = 0.f, y = 0.f, z = 0.f, w = 0.f; };

static inline V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 scl(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// Vector3 / float  (SimpleMath.inl:870-878): scale by the reciprocal.
static inline V3 divs(V3 a, float s) { float r = 1.f / s; return scl(a, r); }
static inline float dot(V3 a, V3 b) { float t = a.x * b.x + a.y * b.y; return t + a.z * b.z; }
static inline V3 cross(V3 a, V3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
static inline bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
// Vector3::Normalize -> XMVector3Normalize: v / sqrt(dot), zero vector stays zero.
static inline V3 normalize(V3 a)
{
    float l = length(a);
    if (!(l != 0.f)) return mk(0.f, 0.f, 0.f);
    return mk(a.x / l, a.y / l, a.z / l);
}
static inline V3 pnormal(const Plane& p) { return mk(p.x, p.y, p.z); }
// Plane(p0,p1,p2): SimpleMath.inl:2773-2780 -> XMPlaneFromPoints (normalised).
static inline Plane plane_from_points(V3 p0, V3 p1, V3 p2)
{
    V3 n = normalize(cross(sub(p0, p1), sub(p0, p2)));
    Plane r; r.x = n.x; r.y = n.y; r.z = n.z; r.w = -dot(n, p0);
    return r;
}
// Plane(point, normal): SimpleMath.inl:2782-2788 -> XMPlaneFromPointNormal (not normalised).
static inline Plane plane_from_point_normal(V3 p, V3 n)
{
    Plane r; r.x = n.x; r.y = n.y; r.z = n.z; r.w = -dot(p, n);
    return r;
}

// --------------------------------------------------------------- solids ----
// Poly::Vertex / Poly::Polyhedron (Inc/Poly.h:15-32) as struct-of-vectors.
static std::atomic<long> g_links_off_the_array{0};      // see clip_polyhedron's compaction; read through orc_links_off_the_array()

struct Solid
{
    std::vector<V3> pos;
    std::vector<std::vector<int>> nb;   // NeighborVertexVec, CCW seen from outside
    std::vector<int> comp;              // Vertex::comp  (default 1)
    std::vector<int> id;                // Vertex::ID    (default -1)
    size_t size() const { return pos.size(); }
    bool empty() const { return pos.empty(); }
    void clear() { pos.clear(); nb.clear(); comp.clear(); id.clear(); }
    void push(V3 p, int c)
    {
        pos.push_back(p); nb.emplace_back(); comp.push_back(c); id.push_back(-1);
    }
};
typedef std::vector<std::vector<int>> Faces;   // Poly::Extract (Inc/Poly.h:32)

// FaceLoop, Src/Poly.cpp:34-41: the neighbour listed just before `prev` (cyclic).
static inline int face_next(const std::vector<int>& ring, int prev)
{
    size_t k = 0;
    while (k < ring.size() && ring[k] != prev) ++k;
    // std::find returning end(): *(end-1) in the reference, i.e. the last element.
    if (k == 0) return ring.back();
    return ring[k - 1];
}

// ComparePlanePoint, Src/Poly.cpp:716-723.
static inline int side_of(const Plane& pl, V3 p)
{
    float s = pl.w + dot(pnormal(pl), p);
    if (std::fabs((double)s) < 1.0e-10) return 0;
    float m = -s;
    return m > 0.f ? 1 : (m < 0.f ? -1 : 0);
}

// ComparePlaneBB, Src/Poly.cpp:725-744 (corners narrowed to float first).
static inline int box_side(const Plane& pl, double x0, double y0, double z0, double x1, double y1, double z1)
{
    float fx0 = (float)x0, fy0 = (float)y0, fz0 = (float)z0;
    float fx1 = (float)x1, fy1 = (float)y1, fz1 = (float)z1;
    int c[8] = {
        side_of(pl, mk(fx0, fy0, fz0)), side_of(pl, mk(fx1, fy0, fz0)),
        side_of(pl, mk(fx1, fy1, fz0)), side_of(pl, mk(fx0, fy1, fz0)),
        side_of(pl, mk(fx0, fy0, fz1)), side_of(pl, mk(fx1, fy0, fz1)),
        side_of(pl, mk(fx1, fy1, fz1)), side_of(pl, mk(fx0, fy1, fz1)) };
    int lo = c[0], hi = c[0];
    for (int i = 1; i < 8; ++i) { lo = std::min(lo, c[i]); hi = std::max(hi, c[i]); }
    if (lo >= 0) return 1;
    if (hi <= 0) return -1;
    return 0;
}

// PlaneLineIntersection, Src/Poly.cpp:746-751.
static inline V3 edge_cut(V3 a, V3 b, const Plane& pl)
{
    float sa = pl.w + dot(pnormal(pl), a);
    float sb = pl.w + dot(pnormal(pl), b);
    return divs(sub(scl(a, sb), scl(b, sa)), sb - sa);
}

struct Box
{
    double lo[3], hi[3];
    void reset()
    {
        for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<double>::max(); hi[i] = std::numeric_limits<double>::lowest(); }
    }
    void grow(V3 p)
    {
        lo[0] = std::min(lo[0], (double)p.x); hi[0] = std::max(hi[0], (double)p.x);
        lo[1] = std::min(lo[1], (double)p.y); hi[1] = std::max(hi[1], (double)p.y);
        lo[2] = std::min(lo[2], (double)p.z); hi[2] = std::max(hi[2], (double)p.z);
    }
};

// Poly::ClipPolyhedron(Polyhedron&, planes), Src/Poly.cpp:265-500.
// Keeps the side n.x + d <= 0 of every plane, in plane order.
static void clip(Solid& S, const std::vector<Plane>& planes)
{
    Box bb; bb.reset();
    for (size_t v = 0; v < S.size(); ++v) bb.grow(S.pos[v]);          // :276-287

    for (size_t kp = 0; kp < planes.size() && !S.empty(); ++kp)       // :290-294
    {
        const Plane& pl = planes[kp];
        int bc = box_side(pl, bb.lo[0], bb.lo[1], bb.lo[2], bb.hi[0], bb.hi[1], bb.hi[2]);  // :297
        bool above = bc == 1, below = bc == -1;
        if (!(above || below))                                        // :303-319
        {
            above = true; below = true;
            for (size_t v = 0; v < S.size(); ++v)
            {
                S.comp[v] = side_of(pl, S.pos[v]);
                if (S.comp[v] == 1) below = false;
                else if (S.comp[v] == -1) above = false;
            }
        }
        if (below) { S.clear(); continue; }                           // :322-327
        if (above) continue;                                          // :328

        // -- new vertices on straddling edges, :332-363
        const int n0 = (int)S.size();
        for (int i = 0; i < n0; ++i)
        {
            if (S.comp[i] != -1) continue;
            const int deg = (int)S.nb[i].size();
            for (int j = 0; j < deg; ++j)
            {
                const int jn = S.nb[i][j];
                if (S.comp[jn] > 0)
                {
                    const int fresh = (int)S.size();
                    S.push(edge_cut(S.pos[i], S.pos[jn], pl), 2);
                    S.nb[fresh].push_back(i);
                    S.nb[fresh].push_back(jn);
                    std::vector<int>& rj = S.nb[jn];
                    size_t q = 0;
                    while (q < rj.size() && rj[q] != i) ++q;
                    if (q < rj.size()) rj[q] = fresh;                  // :350-353
                    S.nb[i][j] = fresh;                                // :354
                }
            }
        }
        const int n1 = (int)S.size();

        // -- patch links to clipped vertices, new vertices first, :367-425
        std::vector<std::vector<int>> snap(S.nb.begin(), S.nb.end());  // old_neighbors
        for (int t = 0; t < n1; ++t)
        {
            const int i = (t + n0) % n1;
            if (!(S.comp[i] == 0 || S.comp[i] == 2)) continue;
            const int deg = (int)S.nb[i].size();
            for (int j = 0; j < deg; ++j)
            {
                const int jn = S.nb[i][j];
                if (S.comp[jn] != -1) continue;
                int prev = i, cur = jn, steps = 0;
                while (S.comp[cur] == -1 && steps++ < n1)              // :389-394
                {
                    int hold = cur;
                    cur = face_next(S.nb[cur], prev);
                    prev = hold;
                }
                const std::vector<int>& ri = S.nb[i];
                if (ri[(size_t)(j + 1) % ri.size()] == cur || cur == i)
                {
                    S.nb[i][j] = -1;                                   // :400
                }
                else
                {
                    S.nb[i][j] = cur;                                  // :404
                    if (S.comp[cur] == 2)
                    {
                        S.nb[cur].insert(S.nb[cur].begin(), i);        // :407-409
                        snap[cur].insert(snap[cur].begin(), -1);
                    }
                    else
                    {
                        std::vector<int>& sn = snap[cur];
                        size_t at = 0;
                        while (at < sn.size() && sn[at] != prev) ++at; // :413-415
                        S.nb[cur].insert(S.nb[cur].begin() + at, i);
                        sn.insert(sn.begin() + at, i);
                    }
                }
            }
        }
        for (int i = 0; i < n1; ++i)                                   // :426-431
        {
            std::vector<int>& r = S.nb[i];
            r.erase(std::remove(r.begin(), r.end(), -1), r.end());
        }

        // -- drop two-neighbour vertices, :433-462
        bool again = true;
        while (again)
        {
            again = false;
            for (int i = 0; i < n1; ++i)
            {
                if (S.comp[i] >= 0 && S.nb[i].size() == 2)
                {
                    again = true;
                    const int a = S.nb[i][0], b = S.nb[i][1];
                    std::vector<int>& ra = S.nb[a];
                    size_t q = 0;
                    while (q < ra.size() && ra[q] != i) ++q;
                    if (q < ra.size()) ra[q] = b;
                    std::vector<int>& rb = S.nb[b];
                    q = 0;
                    while (q < rb.size() && rb[q] != i) ++q;
                    if (q < rb.size()) rb[q] = a;
                    S.comp[i] = -1;
                }
            }
        }

        // -- compaction, :464-495
        bb.reset();
        int live = 0;
        bool off_the_array = false;
        for (int i = 0; i < n1; ++i)
            if (S.comp[i] >= 0) { S.id[i] = live++; bb.grow(S.pos[i]); }
        for (int i = 0; i < n1; ++i)
            if (S.comp[i] >= 0)
                for (size_t j = 0; j < S.nb[i].size(); ++j)
                {
                    const int e_old = S.nb[i][j];
                    S.nb[i][j] = S.id[S.nb[i][j]];
                    // (checker bookkeeping, not reference code) a surviving link to a CLIPPED vertex is renumbered through that
                    // vertex's stale ID -- an index of an earlier compaction, or one that was never set and names no vertex of
                    // the compacted solid, after which the reference indexes outside its vertex array.  Either way what it
                    // returns is an accident of its memory: the engine flags such a solid instead, and
                    // the tests ask this counter whether a flag was justified
                    if (live >= 4 && (S.comp[e_old] < 0 || S.nb[i][j] < 0 || S.nb[i][j] >= live)) g_links_off_the_array.fetch_add(1);
                    if (live >= 4 && (S.nb[i][j] < 0 || S.nb[i][j] >= live)) off_the_array = true;
                }
        int w = 0;
        for (int i = 0; i < n1; ++i)
        {
            if (S.comp[i] < 0) continue;
            if (w != i)
            {
                S.pos[w] = S.pos[i]; S.nb[w].swap(S.nb[i]); S.comp[w] = S.comp[i]; S.id[w] = S.id[i];
            }
            ++w;
        }
        S.pos.resize(w); S.nb.resize(w); S.comp.resize(w); S.id.resize(w);
        if (S.size() < 4) S.clear();                                   // :497-499
        // (checker behaviour, not reference code) a stored link that names no vertex: from here on the reference indexes outside
        // its vertex array (comp[-1] at :336 of the next plane, or whoever walks the solid next) and what it returns, if it
        // returns, depends on its heap.  There is nothing to restate: the checker stops with an empty solid, and the counter
        // above tells the tests why (the engine flags such a pair or fragment)
        if (off_the_array) { S.clear(); return; }
    }
}

// Poly::ExtractFaces, Src/Poly.cpp:89-126.  `ended` (optional): the reference's `while (cur != i)` has no bound; on a
// degenerate solid a walk can fall into a cycle of (previous, current) states that does not hold its start vertex and the
// reference never returns.  A walk longer than the number of half-edges has repeated a state: the restatement stops there
// and reports it (the reference gives no result to compare with); without `ended` it is the literal, unbounded loop.
static Faces extract_faces(const Solid& S, bool* ended = nullptr)
{
    Faces out;
    size_t H = 0;
    for (const auto& r : S.nb) H += r.size();
    if (ended) *ended = true;
    std::set<std::pair<int, int>> seen;
    for (int i = 0; i < (int)S.size(); ++i)
    {
        if (S.comp[i] < 0) continue;
        for (size_t s = 0; s < S.nb[i].size(); ++s)
        {
            const int adj = S.nb[i][s];
            if (seen.count(std::make_pair(i, adj))) continue;
            std::vector<int> loop(1, i);
            int prev = i, cur = adj;
            while (cur != i)
            {
                if (ended && loop.size() > H) { *ended = false; return Faces(); }
                seen.insert(std::make_pair(prev, cur));
                loop.push_back(cur);
                int hold = cur;
                cur = face_next(S.nb[cur], prev);
                prev = hold;
            }
            seen.insert(std::make_pair(prev, cur));
            out.push_back(loop);
        }
    }
    return out;
}

// Poly::Moments, Src/Poly.cpp:55-87 (zeroth moment in double, first in float Vector3).
static void moments(const Solid& S, double& vol, V3& cen)
{
    vol = 0.0; cen = mk(0.f, 0.f, 0.f);
    if (S.size() <= 3) return;
    const V3 origin = S.pos[0];
    Faces fs = extract_faces(S);
    for (const auto& f : fs)
    {
        const size_t n = f.size();
        const V3 p0 = sub(S.pos[f[0]], origin);
        for (size_t k = 1; k + 1 < n; ++k)
        {
            const V3 p1 = sub(S.pos[f[k]], origin);
            const V3 p2 = sub(S.pos[f[(k + 1) % n]], origin);
            const float dV = dot(p0, cross(p1, p2));
            vol += dV;
            cen = add(cen, scl(add(add(p0, p1), p2), dV));
        }
    }
    vol /= 6.0;
    double x = 24.0 * vol;
    double inv = (x >= 0.0 ? 1.0 : -1.0) / std::max(1.0e-30, std::fabs(x));   // safeInv :33
    cen = scl(cen, (float)inv);
    cen = add(cen, origin);
}

// VMACH::OnYourRight, Src/VMACH.cpp:1240-1243.
static inline bool on_right(V3 a, V3 b, V3 c, V3 n) { return dot(cross(sub(b, a), sub(c, a)), n) > 0.f; }

// Poly::IsCCW, Src/Poly.cpp:753-762.
static bool loop_is_ccw(const Solid& S, const std::vector<int>& f, V3 n)
{
    const V3 P = S.pos[f[0]];
    V3 acc = mk(0.f, 0.f, 0.f);
    for (size_t v = 0; v < f.size(); ++v)
        acc = add(acc, cross(sub(S.pos[f[v]], P), sub(S.pos[f[(v + 1) % f.size()]], P)));
    return dot(acc, n) < 0.f;
}

// Poly::EarClipping, Src/Poly.cpp:764-913.  Returns face-local indices; an
// empty vector when the ear search stalls (:899-903).
static std::vector<int> ear_clip(const Solid& S, const std::vector<int>& f)
{
    std::vector<int> tri;
    const int N = (int)f.size();
    if (N <= 2) return tri;
    if (N == 3) { tri = {0, 1, 2}; return tri; }

    V3 nrm = cross(sub(S.pos[f[1]], S.pos[f[0]]), sub(S.pos[f[2]], S.pos[f[0]]));
    if (loop_is_ccw(S, f, nrm)) nrm = neg(nrm);

    std::vector<int> prv(N), nxt(N);
    std::vector<char> reflex(N);
    for (int i = 0; i < N; ++i) { prv[i] = (i + N - 1) % N; nxt[i] = (i + 1) % N; }
    auto is_reflex = [&](int i) {
        return !on_right(S.pos[f[prv[i]]], S.pos[f[i]], S.pos[f[nxt[i]]], nrm);
    };
    for (int i = 0; i < N; ++i) reflex[i] = is_reflex(i) ? 1 : 0;
    std::list<int> rlist;                                             // :823-826
    for (int i = 0; i < N; ++i) if (reflex[i]) rlist.push_back(i);

    auto is_ear = [&](int i) {
        if (reflex[i]) return false;
        const V3 a = S.pos[f[prv[i]]], b = S.pos[f[i]], c = S.pos[f[nxt[i]]];
        for (int r : rlist)
        {
            if (r == prv[i] || r == nxt[i]) continue;
            const V3 p = S.pos[f[r]];
            // third clause of :844 compares a whole Vertex with Vertex(c): never equal for a linked vertex
            if (same(p, a) || same(p, b)) continue;
            if (!on_right(a, b, p, nrm)) continue;
            if (!on_right(b, c, p, nrm)) continue;
            if (!on_right(c, a, p, nrm)) continue;
            return false;
        }
        return true;
    };

    tri.resize(3 * (size_t)(N - 2));
    int skipped = 0, at = 0, left = N, cur = 0;
    while (left > 3)
    {
        const int p = prv[cur], n = nxt[cur];
        if (is_ear(cur))
        {
            tri[at + 0] = p; tri[at + 1] = cur; tri[at + 2] = n;
            nxt[p] = n; prv[n] = p;
            const int adj[2] = {p, n};
            for (int a = 0; a < 2; ++a)
            {
                if (!reflex[adj[a]]) continue;
                reflex[adj[a]] = is_reflex(adj[a]) ? 1 : 0;
                if (!reflex[adj[a]]) rlist.remove(adj[a]);
            }
            at += 3; --left; skipped = 0;
        }
        else if (++skipped > left)
        {
            tri.clear();
            return tri;
        }
        cur = n;
    }
    tri[at + 0] = prv[cur]; tri[at + 1] = cur; tri[at + 2] = nxt[cur];
    return tri;
}

// Poly::RenderPolyhedron, Src/Poly.cpp:681-714.  vnc = 9 floats per vertex
// (VertexNormalColor, Inc/Mesh.h:4-13: position, normal (0,0,0), colour).
static void render(std::vector<float>& vnc, std::vector<uint32_t>& idx, const Solid& S, const Faces& fs,
                   bool convex, V3 colour)
{
    const uint32_t base = (uint32_t)(vnc.size() / 9);
    for (size_t v = 0; v < S.size(); ++v)
    {
        const float rec[9] = {S.pos[v].x, S.pos[v].y, S.pos[v].z, 0.f, 0.f, 0.f, colour.x, colour.y, colour.z};
        vnc.insert(vnc.end(), rec, rec + 9);
    }
    if (convex)
    {
        for (const auto& f : fs)
            for (size_t v = 1; v + 1 < f.size(); ++v)
            {
                idx.push_back(base + f[0]); idx.push_back(base + f[v]); idx.push_back(base + f[v + 1]);
            }
    }
    else
    {
        for (const auto& f : fs)
            for (int l : ear_clip(S, f)) idx.push_back(base + f[l]);
    }
}

// Surtr::CheckMeshIsland + _MeshIslandLoop, Src/Surtr.cpp:2157-2201, restated
// iteratively: groups in discovery order (lowest not-yet-grouped vertex), each
// group = the set reached from the start vertex through neighbour links.
static std::vector<std::vector<int>> islands(const Solid& S)
{
    std::vector<std::vector<int>> groups;
    const int n = (int)S.size();
    std::vector<char> taken(n, 0);
    int start = 0;
    while (true)
    {
        std::vector<char> in(n, 0);
        std::vector<int> stack;
        // the recursion inserts neighbours of `start`, not `start` itself (:2160-2168)
        for (int a : S.nb[start]) if (!in[a]) { in[a] = 1; stack.push_back(a); }
        while (!stack.empty())
        {
            int v = stack.back(); stack.pop_back();
            for (int a : S.nb[v]) if (!in[a]) { in[a] = 1; stack.push_back(a); }
        }
        std::vector<int> g;
        for (int v = 0; v < n; ++v) if (in[v]) { g.push_back(v); taken[v] = 1; }   // std::set order
        groups.push_back(g);
        bool remain = false;
        for (int v = 0; v < n; ++v) if (!taken[v]) { remain = true; start = v; break; }
        if (!remain) break;
        if (g.empty()) break;   // isolated start vertex: the reference would spin forever
    }
    return groups;
}

// Island split of m_fractureTask, Src/Surtr.cpp:1474-1500.
static std::vector<Solid> split_islands(const Solid& mesh)
{
    std::vector<Solid> out;
    auto groups = islands(mesh);
    if (groups.size() < 2) { out.push_back(mesh); return out; }
    for (const auto& g : groups)
    {
        Solid isl;
        std::unordered_map<int, int> remap;
        for (int v : g)
        {
            remap[v] = (int)isl.size();
            isl.pos.push_back(mesh.pos[v]); isl.nb.push_back(mesh.nb[v]);
            isl.comp.push_back(mesh.comp[v]); isl.id.push_back(mesh.id[v]);
        }
        for (auto& ring : isl.nb) for (int& a : ring) a = remap[a];
        out.push_back(isl);
    }
    return out;
}

// Poly::GetBB, Src/Poly.cpp:587-617.
static Solid unit_box()
{
    static const float P[8][3] = {{-.5f, -.5f, -.5f}, {.5f, -.5f, -.5f}, {.5f, .5f, -.5f}, {-.5f, .5f, -.5f},
                                  {-.5f, -.5f, .5f},  {.5f, -.5f, .5f},  {.5f, .5f, .5f},  {-.5f, .5f, .5f}};
    static const int NB[8][3] = {{1, 4, 3}, {5, 0, 2}, {3, 6, 1}, {7, 2, 0}, {5, 7, 0}, {1, 6, 4}, {5, 2, 7}, {4, 6, 3}};
    Solid S;
    for (int i = 0; i < 8; ++i)
    {
        S.push(mk(P[i][0], P[i][1], P[i][2]), 1);
        S.nb[i].assign(NB[i], NB[i] + 3);
    }
    return S;
}

// Poly::ExtractNeighborFromMesh, Src/Poly.cpp:128-263.  Returns false where
// the reference throws (:253-260).  The reference iterates an unordered_map
// of vertices; per-vertex results do not depend on that order.
static bool neighbours_from_mesh(const std::vector<V3>& P, const std::vector<int>& tris,
                                 std::vector<std::vector<int>>& ring)
{
    const int T = (int)(tris.size() / 3);
    const int V = (int)P.size();
    std::vector<std::vector<int>> vtri(V);          // vertexAdjTriIndex (ascending triangle ids)
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 3; ++c) vtri[tris[3 * t + c]].push_back(t);

    std::vector<std::vector<int>> tadj(T);          // AdjTriangleVec, :149-174
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 3; ++c)
        {
            const std::vector<int>& A = vtri[tris[3 * t + c]];
            const std::vector<int>& B = vtri[tris[3 * t + (c + 1) % 3]];
            std::vector<int> both;
            std::set_intersection(A.begin(), A.end(), B.begin(), B.end(), std::back_inserter(both));
            for (int o : both)
            {
                if (o == t) continue;
                if (std::find(tadj[t].begin(), tadj[t].end(), o) != tadj[t].end()) continue;
                tadj[t].push_back(o);
            }
        }

    ring.assign(V, std::vector<int>());
    for (int v = 0; v < V; ++v)
    {
        if (vtri[v].empty()) continue;
        std::vector<int> fan(1, vtri[v][0]);        // "tail", :184-216
        int cur = vtri[v][0];
        while (true)
        {
            int pick = -1, cand = 0;
            for (int o : tadj[cur])
            {
                if (std::find(fan.begin(), fan.end(), o) != fan.end()) continue;
                const int* tv = &tris[3 * o];
                if (tv[0] != v && tv[1] != v && tv[2] != v) continue;
                if (cand == 0) pick = o;
                ++cand;
            }
            if (cand == 0) break;
            if (cand == 1 || cand == 2) { fan.push_back(pick); cur = pick; }
            // more than two candidates: the reference loops without progress; not reachable on manifold input
            else break;
        }
        std::vector<int> col;                       // :218-233
        for (int t : fan)
        {
            int s = 0;
            for (int c = 0; c < 3; ++c) if (tris[3 * t + c] == v) { s = c; break; }
            col.push_back(tris[3 * t + (s + 1) % 3]);
            col.push_back(tris[3 * t + (s + 2) % 3]);
        }
        if (col.size() >= 3)                        // :235-248
        {
            const bool flip = col[1] != col[2];
            if (flip) for (size_t i = 0; i + 1 < col.size(); i += 2) std::swap(col[i], col[i + 1]);
            std::vector<int> uniq;
            for (int e : col) if (std::find(uniq.begin(), uniq.end(), e) == uniq.end()) uniq.push_back(e);
            col = uniq;
            if (flip) std::reverse(col.begin(), col.end());
        }
        ring[v] = col;
    }
    for (int v = 0; v < V; ++v)                      // :253-260
        for (int a : ring[v])
            if (std::find(ring[a].begin(), ring[a].end(), v) == ring[a].end()) return false;
    return true;
}

// ------------------------------------------------------- limited hull ------
// VMACH::ConvexHull, Src/VMACH.cpp:869-1161 (greedy hull with a point limit).
struct HullFace { V3 v[3]; bool visible = false; };
struct HullEdge { V3 e[2]; HullFace* f1 = nullptr; HullFace* f2 = nullptr; bool remove = false; };

struct Hull
{
    std::vector<V3> pts;
    std::vector<char> done;
    std::vector<float> gain;
    std::list<HullFace> faces;
    std::list<HullEdge> edges;
    std::map<std::pair<std::string, std::string>, HullEdge*> emap;   // Key2Edge :941-948 (xor of string hashes)
    std::vector<HullFace*> added, visible;
    uint32_t limit = 0, used = 0;

    static float vol(const HullFace& f, V3 p)                        // :922-939
    {
        float ax = f.v[0].x - p.x, ay = f.v[0].y - p.y, az = f.v[0].z - p.z;
        float bx = f.v[1].x - p.x, by = f.v[1].y - p.y, bz = f.v[1].z - p.z;
        float cx = f.v[2].x - p.x, cy = f.v[2].y - p.y, cz = f.v[2].z - p.z;
        return ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
    }
    static std::string key1(V3 p) { return std::to_string(p.x) + std::to_string(p.y) + std::to_string(p.z); }
    static std::pair<std::string, std::string> key(V3 a, V3 b)
    {
        std::string s = key1(a), t = key1(b);
        return s < t ? std::make_pair(s, t) : std::make_pair(t, s);
    }
    void link(HullEdge* e, HullFace* f)                              // ConvexHullEdge::LinkFace
    {
        if (e->f1 != nullptr && e->f2 != nullptr) return;
        (e->f1 == nullptr ? e->f1 : e->f2) = f;
    }
    void make_edge(V3 a, V3 b, HullFace& f)                          // :971-983
    {
        auto k = key(a, b);
        auto it = emap.find(k);
        if (it == emap.end())
        {
            edges.emplace_back(); edges.back().e[0] = a; edges.back().e[1] = b;
            it = emap.insert({k, &edges.back()}).first;
        }
        link(it->second, &f);
    }
    void make_face(V3 a, V3 b, V3 c, V3 inner)                       // :955-969
    {
        faces.emplace_back();
        HullFace& f = faces.back();
        f.v[0] = a; f.v[1] = b; f.v[2] = c;
        added.push_back(&f);
        if (vol(f, inner) < 0.f) std::swap(f.v[0], f.v[2]);          // ConvexHullFace::Rewind
        make_edge(a, b, f); make_edge(a, c, f); make_edge(b, c, f);
    }
    static float area(V3 a, V3 b, V3 c)                              // ConvexHullFace::CalcArea
    {
        V3 n = cross(sub(b, a), sub(c, a));
        return 0.5f * length(n);
    }
    bool first()                                                     // :1036-1085
    {
        if (pts.size() <= 3) return false;
        size_t i1 = 0;
        for (size_t i = 1; i < pts.size(); ++i) if (pts[i1].x < pts[i].x) i1 = i;
        auto dist = [&](V3 a) {
            return std::sqrt(std::pow((double)(a.x - pts[i1].x), 2) + std::pow((double)(a.y - pts[i1].y), 2) +
                             std::pow((double)(a.z - pts[i1].z), 2));
        };
        size_t i2 = 0;
        for (size_t i = 1; i < pts.size(); ++i) if (dist(pts[i2]) < dist(pts[i])) i2 = i;
        size_t i3 = 0;
        for (size_t i = 1; i < pts.size(); ++i)
            if (area(pts[i1], pts[i2], pts[i3]) < area(pts[i1], pts[i2], pts[i])) i3 = i;
        HullFace base; base.v[0] = pts[i1]; base.v[1] = pts[i2]; base.v[2] = pts[i3];
        size_t i4 = 0;
        for (size_t i = 1; i < pts.size(); ++i) if (vol(base, pts[i4]) < vol(base, pts[i])) i4 = i;
        done[i1] = done[i2] = done[i3] = done[i4] = 1;
        used = 4;
        make_face(pts[i1], pts[i2], pts[i3], pts[i4]);
        make_face(pts[i1], pts[i2], pts[i4], pts[i3]);
        make_face(pts[i1], pts[i3], pts[i4], pts[i2]);
        make_face(pts[i2], pts[i3], pts[i4], pts[i1]);
        return true;
    }
    static V3 inner_point(const HullFace* f, const HullEdge& e)      // :950-962
    {
        for (int i = 0; i < 3; ++i)
        {
            if (same(f->v[i], e.e[0])) continue;
            if (same(f->v[i], e.e[1])) continue;
            return f->v[i];
        }
        return f->v[0];
    }
    void add_point(V3 p)                                             // :994-1034
    {
        bool any = false;
        for (HullFace& f : faces)
            if (vol(f, p) < 0.f) { f.visible = true; visible.push_back(&f); any = true; }
        if (!any) return;
        for (auto it = edges.begin(); it != edges.end(); ++it)       // list grows while iterating
        {
            HullEdge& e = *it;
            if (e.f1 == nullptr || e.f2 == nullptr) continue;
            if (e.f1->visible && e.f2->visible) e.remove = true;
            else if (e.f1->visible || e.f2->visible)
            {
                if (e.f1->visible) std::swap(e.f1, e.f2);
                V3 inner = inner_point(e.f2, e);
                if (e.f1 == e.f2) e.f1 = nullptr; else e.f2 = nullptr;   // EraseFace(face2)
                make_face(e.e[0], e.e[1], p, inner);
            }
        }
    }
    void cleanup()                                                   // :1140-1161
    {
        visible.clear(); added.clear();
        for (auto it = edges.begin(); it != edges.end();)
        {
            if (it->remove) { emap.erase(key(it->e[0], it->e[1])); it = edges.erase(it); }
            else ++it;
        }
        faces.remove_if([](const HullFace& f) { return f.visible; });
    }
    void build(const std::vector<V3>& cloud, uint32_t lim)           // :1087-1138
    {
        pts = cloud; limit = lim;
        done.assign(pts.size(), 0); gain.assign(pts.size(), 0.f);
        if (!first()) return;
        for (size_t i = 0; i < pts.size(); ++i)
        {
            if (done[i]) continue;
            for (const HullFace& f : faces) gain[i] += std::max(0.0f, vol(f, pts[i]));
        }
        if (limit == 0) limit = (uint32_t)pts.size();
        // the initial tetrahedron's faces stay in `added` until the first CleanUp, as in the reference
        while (used < limit)
        {
            size_t k = 0;
            for (size_t i = 1; i < gain.size(); ++i) if (gain[k] < gain[i]) k = i;
            add_point(pts[k]);
            done[k] = 1; gain[k] = -FLT_MAX; ++used;
            for (size_t i = 0; i < pts.size(); ++i)
            {
                if (done[i]) continue;
                float gone = 0.f, came = 0.f;
                for (HullFace* f : visible) gone += std::max(0.0f, vol(*f, pts[i]));
                for (HullFace* f : added) came += std::max(0.0f, vol(*f, pts[i]));
                gain[i] -= gone; gain[i] += came;
            }
            cleanup();
        }
    }
};

// Surtr::GenerateICHNormal, Src/Surtr.cpp:1961-1974.
static std::vector<V3> hull_normals(const std::vector<V3>& pts, int limit)
{
    Hull h; h.build(pts, (uint32_t)limit);
    std::vector<V3> out;
    for (const HullFace& f : h.faces) out.push_back(normalize(cross(sub(f.v[1], f.v[0]), sub(f.v[2], f.v[0]))));
    return out;
}

// --------------------------------------------------------------- k-DOP -----
// Kdop::KdopContainer::Calc(vertices, maxAxisScale, planeGapInv), Src/Kdop.cpp:15-51
// and Calc(Polyhedron), :92-115; plane order of ClipWithPolyhedron, :166-179.
static std::vector<Plane> kdop_planes(const std::vector<V3>& verts, const std::vector<V3>& normals, bool ach,
                                      double maxAxisScale, float gapInv)
{
    const size_t K = normals.size();
    std::vector<double> lo(K, DBL_MAX), hi(K, -DBL_MAX);
    std::vector<V3> vlo(K), vhi(K);
    std::vector<Plane> plo(K), phi(K);
    for (const V3& v : verts)
        for (size_t k = 0; k < K; ++k)
        {
            float t = dot(v, normals[k]);
            if (lo[k] > t) { lo[k] = t; plo[k] = plane_from_point_normal(v, neg(normals[k])); vlo[k] = v; }
            if (hi[k] < t) { hi[k] = t; phi[k] = plane_from_point_normal(v, normals[k]); vhi[k] = v; }
        }
    if (ach)                                                          // :39-50
        for (size_t k = 0; k < K; ++k)
        {
            V3 nlo = normalize(pnormal(plo[k])), nhi = normalize(pnormal(phi[k]));
            float gap = (float)(maxAxisScale / gapInv);
            plo[k] = plane_from_point_normal(add(vlo[k], scl(nlo, gap)), nlo);
            phi[k] = plane_from_point_normal(add(vhi[k], scl(nhi, gap)), nhi);
        }
    std::vector<Plane> out;
    for (size_t k = 0; k < K; ++k) { out.push_back(plo[k]); out.push_back(phi[k]); }
    return out;
}

// m_refittingTask, Src/Surtr.cpp:1449-1455.
static void refit(Solid& convex, const Solid& mesh, int pointLimit)
{
    std::vector<V3> pts(mesh.pos.begin(), mesh.pos.end());
    std::vector<V3> nrm = hull_normals(pts, std::min((int)mesh.size(), pointLimit));
    std::vector<Plane> pls = kdop_planes(pts, nrm, false, 0.0, 1.f);
    clip(convex, pls);
}

// ------------------------------------------------------- Voronoi cells -----
// The reference obtains cells from voro++ (Src/Surtr.cpp:2003-2070), which is
// not in the tree and not in this image; its face order is unpinned.  The
// canonical cell used by this project: the unit box [-0.5,0.5]^3 intersected
// (in double) with the bisector half-spaces of all other seeds; faces listed
// by neighbour seed id ascending, then walls -x,+x,-y,+y,-z,+z; each face loop
// starts at the face's vertex that comes first in (x,y,z) lexicographic order
// and is wound so that (v1-v0)x(v2-v0) points out of the cell (the state after
// PolygonFace::Rewind, :2062).  This is an independent convex-polytope
// construction (vertex enumeration from plane triples is avoided; a convex
// polygon soup is clipped instead), used to cross-check surtr_amd's host builder.
struct D3 { double x, y, z; };
struct CellFace { int gen; std::vector<D3> loop; };

static std::vector<CellFace> voronoi_cell(const std::vector<D3>& seeds, int self)
{
    // start: 6 wall quads, outward winding
    std::vector<CellFace> fs;
    const double h = 0.5;
    auto quad = [&](int gen, D3 a, D3 b, D3 c, D3 d) { CellFace f; f.gen = gen; f.loop = {a, b, c, d}; fs.push_back(f); };
    const int C = (int)seeds.size();
    quad(C + 0, D3{-h, -h, -h}, D3{-h, -h, h}, D3{-h, h, h}, D3{-h, h, -h});   // -x
    quad(C + 1, D3{h, -h, -h}, D3{h, h, -h}, D3{h, h, h}, D3{h, -h, h});       // +x
    quad(C + 2, D3{-h, -h, -h}, D3{h, -h, -h}, D3{h, -h, h}, D3{-h, -h, h});   // -y
    quad(C + 3, D3{-h, h, -h}, D3{-h, h, h}, D3{h, h, h}, D3{h, h, -h});       // +y
    quad(C + 4, D3{-h, -h, -h}, D3{-h, h, -h}, D3{h, h, -h}, D3{h, -h, -h});   // -z
    quad(C + 5, D3{-h, -h, h}, D3{h, -h, h}, D3{h, h, h}, D3{-h, h, h});       // +z
    const D3 s = seeds[self];
    for (int o = 0; o < C; ++o)
    {
        if (o == self) continue;
        const D3 q = seeds[o];
        // half-space n.x <= c with n = q - s, c = (|q|^2-|s|^2)/2
        const D3 n{q.x - s.x, q.y - s.y, q.z - s.z};
        const double c = 0.5 * ((q.x * q.x + q.y * q.y + q.z * q.z) - (s.x * s.x + s.y * s.y + s.z * s.z));
        auto sd = [&](const D3& p) { return n.x * p.x + n.y * p.y + n.z * p.z - c; };
        bool cuts = false;
        for (const auto& f : fs) for (const auto& p : f.loop) if (sd(p) > 1e-12) cuts = true;
        if (!cuts) continue;
        std::vector<CellFace> keep;
        std::vector<std::pair<D3, D3>> capseg;
        for (const auto& f : fs)
        {
            CellFace g; g.gen = f.gen;
            const size_t m = f.loop.size();
            D3 enter{0, 0, 0}, leave{0, 0, 0}; bool he = false, hl = false;
            for (size_t i = 0; i < m; ++i)
            {
                const D3& a = f.loop[i]; const D3& b = f.loop[(i + 1) % m];
                const double da = sd(a), db = sd(b);
                if (da <= 0) g.loop.push_back(a);
                if ((da <= 0) != (db <= 0))
                {
                    const double t = da / (da - db);
                    D3 x{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)};
                    g.loop.push_back(x);
                    if (da <= 0) { leave = x; hl = true; } else { enter = x; he = true; }
                }
            }
            if (g.loop.size() >= 3) keep.push_back(g);
            if (he && hl) capseg.push_back({leave, enter});   // the cut edge, in face order leave -> enter
        }
        if (capseg.size() >= 3)
        {
            // chain the cut segments into the cap loop; reversed so the cap winds outward
            CellFace cap; cap.gen = o;
            std::vector<char> usedseg(capseg.size(), 0);
            D3 curp = capseg[0].second; cap.loop.push_back(capseg[0].first); usedseg[0] = 1;
            for (size_t it = 1; it < capseg.size(); ++it)
            {
                size_t best = capseg.size(); double bd = 1e300;
                for (size_t j = 0; j < capseg.size(); ++j)
                {
                    if (usedseg[j]) continue;
                    const D3& a = capseg[j].first;
                    double d = (a.x - curp.x) * (a.x - curp.x) + (a.y - curp.y) * (a.y - curp.y) + (a.z - curp.z) * (a.z - curp.z);
                    if (d < bd) { bd = d; best = j; }
                }
                if (best == capseg.size()) break;
                usedseg[best] = 1; cap.loop.push_back(capseg[best].first); curp = capseg[best].second;
            }
            std::reverse(cap.loop.begin(), cap.loop.end());
            keep.push_back(cap);
        }
        fs.swap(keep);
    }
    // drop degenerate duplicates inside loops, canonical start + order
    for (auto& f : fs)
    {
        std::vector<D3> u;
        for (const auto& p : f.loop)
        {
            bool dup = false;
            for (const auto& r : u)
                if (std::fabs(r.x - p.x) + std::fabs(r.y - p.y) + std::fabs(r.z - p.z) < 1e-13) dup = true;
            if (!dup) u.push_back(p);
        }
        f.loop.swap(u);
        if (f.loop.empty()) continue;
        size_t st = 0;
        for (size_t i = 1; i < f.loop.size(); ++i)
        {
            const D3& a = f.loop[i]; const D3& b = f.loop[st];
            if (a.x < b.x || (a.x == b.x && (a.y < b.y || (a.y == b.y && a.z < b.z)))) st = i;
        }
        std::rotate(f.loop.begin(), f.loop.begin() + st, f.loop.end());
    }
    fs.erase(std::remove_if(fs.begin(), fs.end(), [](const CellFace& f) { return f.loop.size() < 3; }), fs.end());
    std::stable_sort(fs.begin(), fs.end(), [](const CellFace& a, const CellFace& b) { return a.gen < b.gen; });
    return fs;
}

// Cell placement (Polygon3D::Scale/Translate + ConstructFacePlane,
// Src/VMACH.cpp:302-310, 506-534): first three face vertices, narrowed to
// float, scaled then translated in float, plane rebuilt from them.
static Plane place_face(const float v012[9], V3 scale, V3 shift)
{
    V3 p[3];
    for (int i = 0; i < 3; ++i) p[i] = add(mul(mk(v012[3 * i], v012[3 * i + 1], v012[3 * i + 2]), scale), shift);
    return plane_from_points(p[0], p[1], p[2]);
}

// ----------------------------------------------------------- the event -----
struct Piece { Solid convex, mesh; };
struct Fragment { int cell, piece, island; Solid convex, mesh; };

// m_fractureTask for one cell, Src/Surtr.cpp:1457-1504.
static void fracture_cell(int cell, const std::vector<Plane>& planes, const std::vector<Piece>& pieces,
                          const std::vector<char>& outside, std::vector<Fragment>& out)
{
    for (int c = 0; c < (int)pieces.size(); ++c)
    {
        if (!outside.empty() && outside[c]) continue;
        Solid cv = pieces[c].convex;                                  // copy, Src/Poly.cpp:562
        clip(cv, planes);
        if (cv.empty()) continue;
        Solid ms = pieces[c].mesh;
        clip(ms, planes);
        if (ms.empty()) continue;
        std::vector<Solid> parts = split_islands(ms);
        for (size_t k = 0; k < parts.size(); ++k)
        {
            Fragment f; f.cell = cell; f.piece = c; f.island = (int)k; f.convex = cv; f.mesh.pos.swap(parts[k].pos);
            f.mesh.nb.swap(parts[k].nb); f.mesh.comp.swap(parts[k].comp); f.mesh.id.swap(parts[k].id);
            out.push_back(std::move(f));
        }
    }
}


// ------------------------------------------------- compound regrouping -----
// Surtr::ConvexOutOfSphere, Src/Surtr.cpp:2415-2458.
static bool convex_out_of_sphere(const Solid& S, const Faces& ext, const std::vector<V3>& cloud, V3 origin, float radius)
{
    bool none_inside = true;
    for (size_t v = 0; v < S.size(); ++v)
        if (length(sub(origin, S.pos[v])) < radius) { none_inside = false; break; }
    if (!none_inside) return false;
    for (const V3& po : cloud)
    {
        bool contain = true;
        for (const auto& f : ext)
        {
            V3 n = normalize(cross(sub(S.pos[f[1]], S.pos[f[0]]), sub(S.pos[f[2]], S.pos[f[0]])));
            float d = -dot(S.pos[f[0]], n);
            float dist = dot(n, po) + d;
            if (dist > 0) { contain = false; break; }
        }
        if (contain) return false;
    }
    return true;
}

// Surtr::MergeOutOfImpact, Src/Surtr.cpp:2368-2403.
static void merge_out_of_impact(std::vector<std::set<int>>& bind, const std::vector<Solid>& conv, const std::vector<Faces>& ext,
                                const std::vector<V3>& cloud, V3 origin, float radius)
{
    for (size_t i = 1; i < bind.size(); ++i)
    {
        std::set<int> outside;
        for (int c : bind[i])
            if (convex_out_of_sphere(conv[c], ext[c], cloud, origin, radius)) outside.insert(c);
        if (!outside.empty())
        {
            std::set<int> rest;
            std::set_difference(bind[i].begin(), bind[i].end(), outside.begin(), outside.end(), std::inserter(rest, rest.end()));
            bind[i].swap(rest);
            bind[0].insert(outside.begin(), outside.end());
        }
    }
    bind.erase(std::remove_if(std::next(bind.begin()), bind.end(), [](const std::set<int>& s) { return s.empty(); }), bind.end());
}

// Surtr::HandleConvexIsland, Src/Surtr.cpp:2203-2366.
static void handle_convex_island(std::vector<std::set<int>>& bind, const std::vector<Solid>& conv, const std::vector<Faces>& ext)
{
    struct Node { int cid; double absd; Plane plane; std::vector<V3> pts; };
    std::vector<std::set<int>> fresh;
    for (auto& local : bind)
    {
        if (local.size() <= 1) continue;
        std::vector<Node> nodes;
        for (int cid : local)
            for (const auto& poly : ext[cid])
            {
                Node nd; nd.cid = cid;
                for (int v : poly) nd.pts.push_back(conv[cid].pos[v]);
                nd.plane = plane_from_points(nd.pts[0], nd.pts[1], nd.pts[2]);
                nd.absd = std::abs(nd.plane.w);
                nodes.push_back(nd);
            }
        std::sort(nodes.begin(), nodes.end(), [](const Node& a, const Node& b) { return a.absd < b.absd; });
        std::map<int, std::set<int>> nei;
        for (int cid : local) nei[cid] = std::set<int>();
        for (int i = 0; i < (int)nodes.size() - 1; ++i)
        {
            bool lower = false;
            for (int j = i + 1; j < (int)nodes.size(); ++j)
            {
                if (lower && nodes[i].absd > nodes[j].absd) break;
                if (std::abs(nodes[i].absd - nodes[j].absd) > 1e-3) continue;
                lower = true;
                V3 in = normalize(pnormal(nodes[i].plane)), jn = normalize(pnormal(nodes[j].plane));
                bool opposite = std::abs(1 + dot(in, jn)) < 1e-4;
                if (!opposite) continue;
                bool hit = false;
                const int nj = (int)nodes[j].pts.size();
                for (const V3& ip : nodes[i].pts)
                {
                    bool inc = true;
                    for (int v = 0; v < nj; ++v)
                        if (!on_right(nodes[j].pts[v], nodes[j].pts[(v + 1) % nj], ip, jn)) { inc = false; break; }
                    if (inc) { hit = true; break; }
                }
                if (!hit)
                {
                    const int ni = (int)nodes[i].pts.size();
                    for (const V3& jp : nodes[j].pts)
                    {
                        bool inc = true;
                        for (int v = 0; v < ni; ++v)
                            if (!on_right(nodes[i].pts[v], nodes[i].pts[(v + 1) % ni], jp, in)) { inc = false; break; }
                        if (inc) { hit = true; break; }
                    }
                }
                if (hit) { nei[nodes[i].cid].insert(nodes[j].cid); nei[nodes[j].cid].insert(nodes[i].cid); }
            }
        }
        std::set<int> remain(local.begin(), local.end());
        std::vector<std::set<int>> groups;
        while (!remain.empty())
        {
            std::set<int> g;
            std::list<int> q(1, *remain.begin());
            while (!q.empty())
            {
                int cur = q.front(); q.pop_front();
                if (remain.count(cur))
                {
                    g.insert(cur); remain.erase(cur);
                    for (int a : nei[cur]) q.push_back(a);
                }
            }
            groups.push_back(g);
        }
        if (groups.size() >= 2)
        {
            local = groups[0];
            fresh.insert(fresh.end(), std::next(groups.begin()), groups.end());
        }
    }
    bind.insert(bind.end(), fresh.begin(), fresh.end());
}

} // namespace orc

// ============================================================ C interface ===
// Results travel in a "bag" of byte arrays; the Python side knows the order.
struct orc_bag { std::vector<std::vector<char>> a; };

template <class T> static void bag_put(orc_bag* b, const std::vector<T>& v)
{
    std::vector<char> raw(v.size() * sizeof(T));
    if (!v.empty()) std::memcpy(raw.data(), v.data(), raw.size());
    b->a.push_back(std::move(raw));
}

static orc::Solid solid_in(int nv, const float* pos, const uint32_t* off, const int32_t* nbr)
{
    orc::Solid S;
    for (int v = 0; v < nv; ++v)
    {
        S.push(orc::mk(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]), 1);
        S.nb[v].assign(nbr + off[v], nbr + off[v + 1]);
    }
    return S;
}
static void solid_out(orc_bag* b, const orc::Solid& S)
{
    std::vector<float> pos; std::vector<uint32_t> off(1, 0); std::vector<int32_t> nbr;
    for (size_t v = 0; v < S.size(); ++v)
    {
        pos.push_back(S.pos[v].x); pos.push_back(S.pos[v].y); pos.push_back(S.pos[v].z);
        for (int a : S.nb[v]) nbr.push_back(a);
        off.push_back((uint32_t)nbr.size());
    }
    bag_put(b, pos); bag_put(b, off); bag_put(b, nbr);
}
static std::vector<orc::Plane> planes_in(int n, const float* p)
{
    std::vector<orc::Plane> out(n);
    for (int i = 0; i < n; ++i) { out[i].x = p[4 * i]; out[i].y = p[4 * i + 1]; out[i].z = p[4 * i + 2]; out[i].w = p[4 * i + 3]; }
    return out;
}

extern "C" {

int orc_bag_count(orc_bag* b) { return (int)b->a.size(); }
uint64_t orc_bag_bytes(orc_bag* b, int i) { return (uint64_t)b->a[i].size(); }
void orc_bag_copy(orc_bag* b, int i, void* dst) { if (!b->a[i].empty()) std::memcpy(dst, b->a[i].data(), b->a[i].size()); }
void orc_bag_free(orc_bag* b) { delete b; }

// -> pos, off, nbr
orc_bag* orc_clip(int nv, const float* pos, const uint32_t* off, const int32_t* nbr, int np, const float* planes)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    orc::clip(S, planes_in(np, planes));
    orc_bag* b = new orc_bag; solid_out(b, S); return b;
}

// -> face_off, face_idx
orc_bag* orc_extract_faces(int nv, const float* pos, const uint32_t* off, const int32_t* nbr)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    orc::Faces fs = orc::extract_faces(S);
    std::vector<uint32_t> fo(1, 0); std::vector<int32_t> fi;
    for (auto& f : fs) { for (int v : f) fi.push_back(v); fo.push_back((uint32_t)fi.size()); }
    orc_bag* b = new orc_bag; bag_put(b, fo); bag_put(b, fi); return b;
}

// -> vnc (9 floats / vertex), idx
orc_bag* orc_render(int nv, const float* pos, const uint32_t* off, const int32_t* nbr, int convex, const float* colour)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    std::vector<float> vnc; std::vector<uint32_t> idx;
    orc::render(vnc, idx, S, orc::extract_faces(S), convex != 0, orc::mk(colour[0], colour[1], colour[2]));
    orc_bag* b = new orc_bag; bag_put(b, vnc); bag_put(b, idx); return b;
}

// -> [volume (double), cx, cy, cz (as doubles)]
orc_bag* orc_moments(int nv, const float* pos, const uint32_t* off, const int32_t* nbr)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    double vol; orc::V3 c; orc::moments(S, vol, c);
    std::vector<double> r = {vol, (double)c.x, (double)c.y, (double)c.z};
    orc_bag* b = new orc_bag; bag_put(b, r); return b;
}

// -> label per vertex (island id in discovery order), n_islands
orc_bag* orc_islands(int nv, const float* pos, const uint32_t* off, const int32_t* nbr)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    auto g = orc::islands(S);
    std::vector<int32_t> lab(nv, -1);
    for (size_t k = 0; k < g.size(); ++k) for (int v : g[k]) lab[v] = (int32_t)k;
    std::vector<int32_t> n(1, (int32_t)g.size());
    orc_bag* b = new orc_bag; bag_put(b, lab); bag_put(b, n); return b;
}

// -> pos, off, nbr  (unit box, Poly::GetBB)
orc_bag* orc_unit_box()
{
    orc_bag* b = new orc_bag; solid_out(b, orc::unit_box()); return b;
}

// -> ok flag, off, nbr
orc_bag* orc_neighbours_from_mesh(int nv, const float* pos, int ntri, const int32_t* tris)
{
    std::vector<orc::V3> P(nv);
    for (int v = 0; v < nv; ++v) P[v] = orc::mk(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
    std::vector<int> T(tris, tris + 3 * (size_t)ntri);
    std::vector<std::vector<int>> ring;
    bool ok = orc::neighbours_from_mesh(P, T, ring);
    std::vector<int32_t> flag(1, ok ? 1 : 0); std::vector<uint32_t> off(1, 0); std::vector<int32_t> nbr;
    for (auto& r : ring) { for (int a : r) nbr.push_back(a); off.push_back((uint32_t)nbr.size()); }
    orc_bag* b = new orc_bag; bag_put(b, flag); bag_put(b, off); bag_put(b, nbr); return b;
}

// -> normals (3 floats each)
orc_bag* orc_hull_normals(int n, const float* pts, int limit)
{
    std::vector<orc::V3> P(n);
    for (int v = 0; v < n; ++v) P[v] = orc::mk(pts[3 * v], pts[3 * v + 1], pts[3 * v + 2]);
    auto nr = orc::hull_normals(P, limit);
    std::vector<float> out;
    for (auto& q : nr) { out.push_back(q.x); out.push_back(q.y); out.push_back(q.z); }
    orc_bag* b = new orc_bag; bag_put(b, out); return b;
}

// -> planes (4 floats each), order: for each normal Min plane then Max plane
orc_bag* orc_kdop_planes(int n, const float* pts, int k, const float* normals, int ach, double maxAxisScale, float gapInv)
{
    std::vector<orc::V3> P(n), N(k);
    for (int v = 0; v < n; ++v) P[v] = orc::mk(pts[3 * v], pts[3 * v + 1], pts[3 * v + 2]);
    for (int v = 0; v < k; ++v) N[v] = orc::mk(normals[3 * v], normals[3 * v + 1], normals[3 * v + 2]);
    auto pl = orc::kdop_planes(P, N, ach != 0, maxAxisScale, gapInv);
    std::vector<float> out;
    for (auto& q : pl) { out.push_back(q.x); out.push_back(q.y); out.push_back(q.z); out.push_back(q.w); }
    orc_bag* b = new orc_bag; bag_put(b, out); return b;
}

// -> convex pos, off, nbr after m_refittingTask
// Number of times (since the last call with reset != 0) a compaction left a link that names no vertex: the tests use it to tell
// "the reference has no defined result here" from a plain difference.
long orc_links_off_the_array(int reset)
{
    return reset ? orc::g_links_off_the_array.exchange(0) : orc::g_links_off_the_array.load();
}

orc_bag* orc_refit(int cnv, const float* cpos, const uint32_t* coff, const int32_t* cnbr,
                   int mnv, const float* mpos, const uint32_t* moff, const int32_t* mnbr, int pointLimit)
{
    orc::Solid C = solid_in(cnv, cpos, coff, cnbr), M = solid_in(mnv, mpos, moff, mnbr);
    orc::refit(C, M, pointLimit);
    orc_bag* b = new orc_bag; solid_out(b, C); return b;
}

// Canonical Voronoi cells of `n` seeds (doubles) in the unit box.
// -> cell_face_off (n+1), face_gen (per face), face_vert_off (F+1), verts (3 doubles each)
orc_bag* orc_voronoi_cells(int n, const double* seeds)
{
    std::vector<orc::D3> S(n);
    for (int i = 0; i < n; ++i) S[i] = orc::D3{seeds[3 * i], seeds[3 * i + 1], seeds[3 * i + 2]};
    std::vector<uint32_t> cfo(1, 0), fvo(1, 0); std::vector<int32_t> gen; std::vector<double> verts;
    for (int c = 0; c < n; ++c)
    {
        auto fs = orc::voronoi_cell(S, c);
        for (auto& f : fs)
        {
            gen.push_back(f.gen);
            for (auto& p : f.loop) { verts.push_back(p.x); verts.push_back(p.y); verts.push_back(p.z); }
            fvo.push_back((uint32_t)(verts.size() / 3));
        }
        cfo.push_back((uint32_t)gen.size());
    }
    orc_bag* b = new orc_bag; bag_put(b, cfo); bag_put(b, gen); bag_put(b, fvo); bag_put(b, verts); return b;
}

// Cell placement -> planes (4 floats per face) from the first three vertices of each face.
orc_bag* orc_place_cells(int nfaces, const float* v012, const float* scale, const float* shift)
{
    std::vector<float> out;
    for (int f = 0; f < nfaces; ++f)
    {
        orc::Plane p = orc::place_face(v012 + 9 * (size_t)f, orc::mk(scale[0], scale[1], scale[2]),
                                       orc::mk(shift[0], shift[1], shift[2]));
        out.push_back(p.x); out.push_back(p.y); out.push_back(p.z); out.push_back(p.w);
    }
    orc_bag* b = new orc_bag; bag_put(b, out); return b;
}

// Seeds: Surtr::GenerateVoronoi(int) (Src/Surtr.cpp:1984-2001) and
// GenerateFracturePattern (:2072-2096) with libstdc++'s distributions.
// mode 0: uniform(-0.5,0.5)^3; mode 1: exponential length x normalised direction.
orc_bag* orc_seeds(int n, unsigned seed, int mode, double mean)
{
    std::vector<double> out;
    std::mt19937 gen(seed);
    if (mode == 0)
    {
        std::uniform_real_distribution<double> u(-0.5, 0.5);
        for (int i = 0; i < n; ++i) { double x = u(gen), y = u(gen), z = u(gen); out.push_back((float)x); out.push_back((float)y); out.push_back((float)z); }
    }
    else
    {
        std::uniform_real_distribution<double> d(-1.0, 1.0);
        std::exponential_distribution<double> e(1.0 / mean);
        for (int i = 0; i < n; ++i)
        {
            double len = std::max(std::min(e(gen), 0.5), 1e-12);
            double x = d(gen), y = d(gen), z = d(gen);
            orc::V3 v = orc::normalize(orc::mk((float)x, (float)y, (float)z));
            v = orc::scl(v, (float)len);
            out.push_back(v.x); out.push_back(v.y); out.push_back(v.z);
        }
    }
    orc_bag* b = new orc_bag; bag_put(b, out); return b;
}

// The whole event (ApplyFracture Src/Surtr.cpp:2098-2149 without the bind
// bookkeeping, then Refitting :2405-2413, then ExtractFaces+RenderPolyhedron
// of every Mesh as in m_initCompoundTask :1436-1447), one task per cell on
// `threads` workers, results consumed in cell order.
//   pieces: concatenated solids, piece p = vertices [pvo[p], pvo[p+1]) with a
//   global CSR (off over all vertices, nbr piece-local).
// flags bit0: refit, bit1: render.
// -> frag_ids (3 int32 each: cell, piece, island),
//    mesh_vert_off, mesh_pos, mesh_nbr_off, mesh_nbr,
//    conv_vert_off, conv_pos, conv_nbr_off, conv_nbr,
//    vnc, idx_off, idx, seconds (double[1])
orc_bag* orc_event(int npieces,
                   const uint32_t* mvo, const float* mpos, const uint32_t* moff, const int32_t* mnbr,
                   const uint32_t* cvo, const float* cpos, const uint32_t* coff, const int32_t* cnbr,
                   int ncells, const uint32_t* plane_off, const float* planes, const uint8_t* outside,
                   int flags, int threads, int cell_begin, int cell_end)
{
    std::vector<orc::Piece> pieces(npieces);
    for (int p = 0; p < npieces; ++p)
    {
        {
            const uint32_t b0 = mvo[p], b1 = mvo[p + 1];
            for (uint32_t v = b0; v < b1; ++v)
            {
                pieces[p].mesh.push(orc::mk(mpos[3 * v], mpos[3 * v + 1], mpos[3 * v + 2]), 1);
                pieces[p].mesh.nb.back().assign(mnbr + moff[v], mnbr + moff[v + 1]);
            }
        }
        {
            const uint32_t b0 = cvo[p], b1 = cvo[p + 1];
            for (uint32_t v = b0; v < b1; ++v)
            {
                pieces[p].convex.push(orc::mk(cpos[3 * v], cpos[3 * v + 1], cpos[3 * v + 2]), 1);
                pieces[p].convex.nb.back().assign(cnbr + coff[v], cnbr + coff[v + 1]);
            }
        }
    }
    std::vector<char> out_mask;
    if (outside) out_mask.assign(outside, outside + npieces);
    if (cell_end < 0) cell_end = ncells;
    const int ncell_run = std::max(0, cell_end - cell_begin);

    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::vector<orc::Fragment>> per_cell(ncell_run);
    std::atomic<int> next(0);
    auto worker = [&]() {
        while (true)
        {
            int k = next.fetch_add(1);
            if (k >= ncell_run) break;
            const int cell = cell_begin + k;
            std::vector<orc::Plane> pl = planes_in((int)(plane_off[cell + 1] - plane_off[cell]), planes + 4 * (size_t)plane_off[cell]);
            orc::fracture_cell(cell, pl, pieces, out_mask, per_cell[k]);
        }
    };
    const int nt = std::max(1, threads);
    if (nt == 1) worker();
    else
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    std::vector<orc::Fragment*> frags;
    for (auto& v : per_cell) for (auto& f : v) frags.push_back(&f);

    if (flags & 1)
    {
        std::atomic<int> nx(0);
        auto rw = [&]() {
            while (true) { int k = nx.fetch_add(1); if (k >= (int)frags.size()) break; orc::refit(frags[k]->convex, frags[k]->mesh, 4); }
        };
        if (nt == 1) rw();
        else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(rw); for (auto& t : th) t.join(); }
    }
    std::vector<std::vector<float>> vncs(frags.size());
    std::vector<std::vector<uint32_t>> idxs(frags.size());
    if (flags & 2)
    {
        std::atomic<int> nx(0);
        auto rw = [&]() {
            while (true)
            {
                int k = nx.fetch_add(1); if (k >= (int)frags.size()) break;
                // a fragment on which ExtractFaces would never end gets its vertices and no triangles (the engine flags
                // it in frag_status; the reference itself hangs there)
                bool ended = true;
                const orc::Faces fs = orc::extract_faces(frags[k]->mesh, &ended);
                orc::render(vncs[k], idxs[k], frags[k]->mesh, fs, false, orc::mk(0.25f, 0.25f, 0.25f));
            }
        };
        if (nt == 1) rw();
        else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(rw); for (auto& t : th) t.join(); }
    }
    auto t1 = std::chrono::steady_clock::now();

    std::vector<int32_t> ids;
    std::vector<uint32_t> mvoff(1, 0), mnoff(1, 0), cvoff(1, 0), cnoff(1, 0), ioff(1, 0);
    std::vector<float> mp, cp, vnc; std::vector<int32_t> mn, cn; std::vector<uint32_t> idx;
    for (size_t k = 0; k < frags.size(); ++k)
    {
        const orc::Fragment& f = *frags[k];
        ids.push_back(f.cell); ids.push_back(f.piece); ids.push_back(f.island);
        for (size_t v = 0; v < f.mesh.size(); ++v)
        {
            mp.push_back(f.mesh.pos[v].x); mp.push_back(f.mesh.pos[v].y); mp.push_back(f.mesh.pos[v].z);
            for (int a : f.mesh.nb[v]) mn.push_back(a);
            mnoff.push_back((uint32_t)mn.size());
        }
        mvoff.push_back((uint32_t)(mp.size() / 3));
        for (size_t v = 0; v < f.convex.size(); ++v)
        {
            cp.push_back(f.convex.pos[v].x); cp.push_back(f.convex.pos[v].y); cp.push_back(f.convex.pos[v].z);
            for (int a : f.convex.nb[v]) cn.push_back(a);
            cnoff.push_back((uint32_t)cn.size());
        }
        cvoff.push_back((uint32_t)(cp.size() / 3));
        vnc.insert(vnc.end(), vncs[k].begin(), vncs[k].end());
        idx.insert(idx.end(), idxs[k].begin(), idxs[k].end());
        ioff.push_back((uint32_t)idx.size());
    }
    std::vector<double> secs(1, std::chrono::duration<double>(t1 - t0).count());
    orc_bag* b = new orc_bag;
    bag_put(b, ids);
    bag_put(b, mvoff); bag_put(b, mp); bag_put(b, mnoff); bag_put(b, mn);
    bag_put(b, cvoff); bag_put(b, cp); bag_put(b, cnoff); bag_put(b, cn);
    bag_put(b, vnc); bag_put(b, ioff); bag_put(b, idx); bag_put(b, secs);
    return b;
}


// -> [flag]
orc_bag* orc_convex_out_of_sphere(int nv, const float* pos, const uint32_t* off, const int32_t* nbr, int ns, const float* sphere,
                                  const float* origin, float radius)
{
    orc::Solid S = solid_in(nv, pos, off, nbr);
    std::vector<orc::V3> cloud(ns);
    for (int i = 0; i < ns; ++i) cloud[i] = orc::mk(sphere[3 * i], sphere[3 * i + 1], sphere[3 * i + 2]);
    bool r = orc::convex_out_of_sphere(S, orc::extract_faces(S), cloud, orc::mk(origin[0], origin[1], origin[2]), radius);
    std::vector<int32_t> f(1, r ? 1 : 0);
    orc_bag* b = new orc_bag; bag_put(b, f); return b;
}

// Bind sets of ApplyFracture + MergeOutOfImpact (when partial) + HandleConvexIsland.
// -> compound_off, compound_piece
orc_bag* orc_regroup(int npieces, int noutside, const int32_t* piece_cell, const uint32_t* cvo, const float* cpos,
                     const uint32_t* coff, const int32_t* cnbr, int partial, int ns, const float* sphere, const float* origin, float radius)
{
    std::vector<orc::Solid> conv(npieces);
    std::vector<orc::Faces> ext(npieces);
    for (int p = 0; p < npieces; ++p)
    {
        for (uint32_t v = cvo[p]; v < cvo[p + 1]; ++v)
        {
            conv[p].push(orc::mk(cpos[3 * v], cpos[3 * v + 1], cpos[3 * v + 2]), 1);
            conv[p].nb.back().assign(cnbr + coff[v], cnbr + coff[v + 1]);
        }
        ext[p] = orc::extract_faces(conv[p]);
    }
    std::vector<std::set<int>> bind(1);
    for (int p = 0; p < noutside; ++p) bind[0].insert(p);
    for (int p = noutside; p < npieces; ++p)
    {
        if (p == noutside || piece_cell[p] != piece_cell[p - 1]) bind.emplace_back();
        bind.back().insert(p);
    }
    if (partial)
    {
        std::vector<orc::V3> cloud(ns);
        for (int i = 0; i < ns; ++i) cloud[i] = orc::mk(sphere[3 * i], sphere[3 * i + 1], sphere[3 * i + 2]);
        orc::merge_out_of_impact(bind, conv, ext, cloud, orc::mk(origin[0], origin[1], origin[2]), radius);
    }
    orc::handle_convex_island(bind, conv, ext);
    std::vector<uint32_t> co(1, 0); std::vector<int32_t> cp;
    for (auto& s : bind) { for (int c : s) cp.push_back(c); co.push_back((uint32_t)cp.size()); }
    orc_bag* b = new orc_bag; bag_put(b, co); bag_put(b, cp); return b;
}

} // extern "C"
