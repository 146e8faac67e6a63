// literal_clip.h -- Poly::ClipPolyhedron (Src/Poly.cpp:265-500) statement by statement on ONE lane, for the solids the
// parallel clipper refuses.
//
// clip_core.h reproduces the reference wherever its relink is well behaved; where a face walk of the relink runs into its
// step bound on a clipped vertex (:389-394: degenerate slivers -- coincident vertices, rings that list a neighbour several
// times) the reference carries on with a link to a clipped vertex and even inserts into that vertex's ring, which no
// order-free formulation follows.  Mostly the solid is gone a moment later (fewer than four vertices left, :497-499) and the
// reference's answer is simply "empty".  This file is the last resort for such a solid when it is small: the whole clip again,
// literally (bounding-box shortcut, snapshot of the rings, insertions, two-neighbour collapse, compaction), with fixed-stride
// rings in the workgroup's global scratch.  THE DEGENERATE POLICY (frozen in round 3): where the reference itself leaves its
// domain -- an index that is no vertex (out of range, a removal mark read as a vertex), or a link to a clipped vertex that
// survives the compaction and would be renumbered through a stale or never-set ID -- this returns SURTR_E_TOPOLOGY and the
// caller flags the fragment / pair; nothing the reference does after that point is emulated.  What stays: the reference's walk
// bound (:389-394) and its "fewer than four vertices: empty" answers.  Found by scripts/fuzz_refracture_gpu.py, seed 555
// case 110 (tests/golden/degenerate_walk_bound_convex.npz).
#pragma once
#include "clip_core.h"

#define LIT_STRIDE 32u       // ring entries a vertex may hold (insertions included)

namespace surtr {

struct LitSolid
{
    float* pos; uint32_t* len; int32_t* ring; uint32_t* slen; int32_t* snap; int8_t* comp; int32_t* id;
    uint32_t capV;           // vertices (rings: capV * LIT_STRIDE entries in `ring` and in `snap`)
};

__device__ inline int lit_face_next(const int32_t* r, uint32_t n, int32_t prev)       // FaceLoop (:34-41)
{
    uint32_t k = 0;
    while (k < n && r[k] != prev) ++k;
    return (k == 0 || k == n) ? r[n - 1] : r[k - 1];       // (std::find returning end(): the reference reads *(end - 1))
}

// ComparePlaneBB (:725-744): the box corners narrowed to float first
__device__ inline int lit_box_side(const float4 pl, const double lo[3], const double hi[3])
{
    int cmin = 2, cmax = -2;
    for (int q = 0; q < 8; ++q)
    {
        const int c = side_of(plane_dist(pl, (float)((q & 1) ? hi[0] : lo[0]), (float)((q & 2) ? hi[1] : lo[1]), (float)((q & 4) ? hi[2] : lo[2])));
        cmin = c < cmin ? c : cmin; cmax = c > cmax ? c : cmax;
    }
    if (cmin >= 0) return 1;
    if (cmax <= 0) return -1;
    return 0;
}

// One lane.  Returns 0 (n_out vertices left in S, compacted: rings of vertex v = S.ring[v * LIT_STRIDE ..], S.len[v] entries),
// SURTR_E_TOPOLOGY where the reference leaves its domain, SURTR_E_CAPACITY when the solid does not fit.
// ids_set: the vertices come with ID = own index (the solid is the result of an earlier ClipPolyhedron that compacted it, as the
// Convex that m_refittingTask clips); otherwise with ID = -1 (a solid built from arrays).
__device__ inline int literal_clip(const SolidIn in, const uint32_t F, const float4* planes, LitSolid S, uint32_t* n_out, bool* stale_out = nullptr,
                                   bool ids_set = false)
{
    uint32_t n = in.nv;
    if (n > S.capV) return SURTR_E_CAPACITY;
    for (uint32_t v = 0; v < n; ++v)
    {
        S.pos[3 * v] = in.pos[3 * v]; S.pos[3 * v + 1] = in.pos[3 * v + 1]; S.pos[3 * v + 2] = in.pos[3 * v + 2];
        const uint32_t deg = in.llen[v];
        if (deg > LIT_STRIDE) return SURTR_E_CAPACITY;
        S.len[v] = deg; S.comp[v] = 1; S.id[v] = -1;
        for (uint32_t j = 0; j < deg; ++j) S.ring[v * LIT_STRIDE + j] = (in.nbr + in.loff[v])[j];
    }
    double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308}, hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
    auto grow = [&](uint32_t v) {
        for (int a = 0; a < 3; ++a) { const double x = S.pos[3 * v + a]; lo[a] = x < lo[a] ? x : lo[a]; hi[a] = x > hi[a] ? x : hi[a]; }
    };
    for (uint32_t v = 0; v < n; ++v) grow(v);                                   // :276-287
    auto ok = [&](int32_t e, uint32_t count) { return e >= 0 && (uint32_t)e < count; };
    for (uint32_t kp = 0; kp < F && n != 0u; ++kp)
    {
        const float4 pl = planes[kp];
        const int bc = lit_box_side(pl, lo, hi);                                // :297
        bool above = bc == 1, below = bc == -1;
        if (!(above || below))                                                  // :303-319
        {
            above = true; below = true;
            for (uint32_t v = 0; v < n; ++v)
            {
                const int c = side_of(plane_dist(pl, S.pos[3 * v], S.pos[3 * v + 1], S.pos[3 * v + 2]));
                S.comp[v] = (int8_t)c;
                if (c == 1) below = false; else if (c == -1) above = false;
            }
        }
        if (below) { n = 0; break; }                                            // :322-327
        if (above) continue;
        // new vertices on straddling edges (:332-363)
        const uint32_t n0 = n;
        for (uint32_t i = 0; i < n0; ++i)
        {
            if (S.comp[i] != -1) continue;
            const uint32_t deg = S.len[i];
            for (uint32_t j = 0; j < deg; ++j)
            {
                const int32_t jn = S.ring[i * LIT_STRIDE + j];
                if (!ok(jn, n)) return SURTR_E_TOPOLOGY;
                if (S.comp[jn] <= 0) continue;
                if (n >= S.capV) return SURTR_E_CAPACITY;
                const uint32_t fresh = n++;
                const float ax = S.pos[3 * i], ay = S.pos[3 * i + 1], az = S.pos[3 * i + 2];
                const float bx = S.pos[3 * jn], by = S.pos[3 * jn + 1], bz = S.pos[3 * jn + 2];
                const float sa = plane_dist(pl, ax, ay, az), sb = plane_dist(pl, bx, by, bz);
                const float inv = 1.f / (sb - sa);                               // PlaneLineIntersection (:746-751)
                S.pos[3 * fresh] = (ax * sb - bx * sa) * inv; S.pos[3 * fresh + 1] = (ay * sb - by * sa) * inv; S.pos[3 * fresh + 2] = (az * sb - bz * sa) * inv;
                S.comp[fresh] = 2; S.id[fresh] = -1; S.len[fresh] = 2;
                S.ring[fresh * LIT_STRIDE] = (int32_t)i; S.ring[fresh * LIT_STRIDE + 1] = jn;
                int32_t* rj = S.ring + (uint32_t)jn * LIT_STRIDE;
                for (uint32_t q = 0; q < S.len[jn]; ++q) if (rj[q] == (int32_t)i) { rj[q] = (int32_t)fresh; break; }      // :350-353
                S.ring[i * LIT_STRIDE + j] = (int32_t)fresh;                                                               // :354
            }
        }
        const uint32_t n1 = n;
        // patch links to clipped vertices, new vertices first (:367-425)
        for (uint32_t v = 0; v < n1; ++v) { S.slen[v] = S.len[v]; for (uint32_t j = 0; j < S.len[v]; ++j) S.snap[v * LIT_STRIDE + j] = S.ring[v * LIT_STRIDE + j]; }
        for (uint32_t t = 0; t < n1; ++t)
        {
            const uint32_t i = (t + n0) % n1;
            if (!(S.comp[i] == 0 || S.comp[i] == 2)) continue;
            const uint32_t deg = S.len[i];
            for (uint32_t j = 0; j < deg; ++j)
            {
                const int32_t jn = S.ring[i * LIT_STRIDE + j];
                // (a removal mark that an insertion moved up into the part of the ring still to be visited is no vertex either:
                // the reference reads comp[-1] there, outside its array -- undefined, flagged, not emulated)
                if (!ok(jn, n1)) return SURTR_E_TOPOLOGY;
                if (S.comp[jn] != -1) continue;
                int32_t prev = (int32_t)i, cur = jn; uint32_t steps = 0;
                while (S.comp[cur] == -1 && steps++ < n1)                        // :389-394
                {
                    const int32_t hold = cur;
                    if (S.len[cur] == 0) return SURTR_E_TOPOLOGY;
                    cur = lit_face_next(S.ring + (uint32_t)cur * LIT_STRIDE, S.len[cur], prev);
                    if (!ok(cur, n1)) return SURTR_E_TOPOLOGY;
                    prev = hold;
                }
                int32_t* ri = S.ring + i * LIT_STRIDE;
                if (ri[(j + 1u) % S.len[i]] == cur || cur == (int32_t)i) ri[j] = -1;                                       // :400
                else
                {
                    ri[j] = cur;                                                 // :404
                    const uint32_t c = (uint32_t)cur;
                    if (S.len[c] >= LIT_STRIDE || S.slen[c] >= LIT_STRIDE) return SURTR_E_CAPACITY;
                    uint32_t at = 0;
                    if (S.comp[c] != 2) { while (at < S.slen[c] && S.snap[c * LIT_STRIDE + at] != prev) ++at; }            // :413-415
                    if (at > S.len[c]) return SURTR_E_TOPOLOGY;
                    for (uint32_t q = S.len[c]; q > at; --q) S.ring[c * LIT_STRIDE + q] = S.ring[c * LIT_STRIDE + q - 1];
                    S.ring[c * LIT_STRIDE + at] = (int32_t)i; ++S.len[c];
                    for (uint32_t q = S.slen[c]; q > at; --q) S.snap[c * LIT_STRIDE + q] = S.snap[c * LIT_STRIDE + q - 1];
                    S.snap[c * LIT_STRIDE + at] = S.comp[c] == 2 ? -1 : (int32_t)i; ++S.slen[c];                          // :407-409 / :416-417
                }
            }
        }
        for (uint32_t v = 0; v < n1; ++v)                                        // :426-431
        {
            int32_t* r = S.ring + v * LIT_STRIDE; uint32_t w = 0;
            for (uint32_t q = 0; q < S.len[v]; ++q) if (r[q] != -1) r[w++] = r[q];
            S.len[v] = w;
        }
        // two-neighbour vertices (:433-462)
        bool again = true; uint32_t rounds = 0;
        while (again)
        {
            again = false;
            if (rounds++ > n1) return SURTR_E_TOPOLOGY;                          // (the reference would not return)
            for (uint32_t i = 0; i < n1; ++i)
            {
                if (S.comp[i] < 0 || S.len[i] != 2u) continue;
                again = true;
                const int32_t a = S.ring[i * LIT_STRIDE], b = S.ring[i * LIT_STRIDE + 1];
                if (!ok(a, n1) || !ok(b, n1)) return SURTR_E_TOPOLOGY;
                int32_t* ra = S.ring + (uint32_t)a * LIT_STRIDE;
                for (uint32_t q = 0; q < S.len[a]; ++q) if (ra[q] == (int32_t)i) { ra[q] = b; break; }
                int32_t* rb = S.ring + (uint32_t)b * LIT_STRIDE;
                for (uint32_t q = 0; q < S.len[b]; ++q) if (rb[q] == (int32_t)i) { rb[q] = a; break; }
                S.comp[i] = -1;
            }
        }
        // compaction (:464-495)
        for (int a = 0; a < 3; ++a) { lo[a] = 1.7976931348623157e308; hi[a] = -1.7976931348623157e308; }
        uint32_t live = 0;
        for (uint32_t i = 0; i < n1; ++i) if (S.comp[i] >= 0) { S.id[i] = (int32_t)live++; grow(i); }
        if (live < 4u) { n = 0; break; }                                         // :497-499 (whatever the links look like)
        for (uint32_t i = 0; i < n1; ++i)
        {
            if (S.comp[i] < 0) continue;
            for (uint32_t j = 0; j < S.len[i]; ++j)
            {
                const int32_t e = S.ring[i * LIT_STRIDE + j];
                if (!ok(e, n1)) return SURTR_E_TOPOLOGY;
                // A link to a clipped vertex that survived the relink would be renumbered through that vertex's stale ID
                // (:484-493: an index of an earlier compaction, or one that was never set): whatever the reference returns
                // after that is an accident of its memory.  Flagged, not emulated ("the degenerate policy").
                if (S.comp[e] < 0) return SURTR_E_TOPOLOGY;
                S.ring[i * LIT_STRIDE + j] = S.id[e];
            }
        }
        uint32_t w = 0;
        for (uint32_t i = 0; i < n1; ++i)
        {
            if (S.comp[i] < 0) continue;
            if (w != i)
            {
                S.pos[3 * w] = S.pos[3 * i]; S.pos[3 * w + 1] = S.pos[3 * i + 1]; S.pos[3 * w + 2] = S.pos[3 * i + 2];
                S.len[w] = S.len[i]; S.comp[w] = S.comp[i]; S.id[w] = S.id[i];
                for (uint32_t j = 0; j < S.len[i]; ++j) S.ring[w * LIT_STRIDE + j] = S.ring[i * LIT_STRIDE + j];
            }
            ++w;
        }
        n = w;
        if (n < 4u) { n = 0; break; }                                            // :497-499
    }
    *n_out = n;
    (void)ids_set;
    if (stale_out != nullptr) *stale_out = false;      // (kept for the callers' signature: a stale link is an error now)
    return 0;
}

} // namespace surtr
