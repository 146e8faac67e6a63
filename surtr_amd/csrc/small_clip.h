// small_clip.h -- Poly::ClipPolyhedron (Src/Poly.cpp:265-500) for a SMALL solid on ONE wave, regular planes only.
//
// The general clipper of clip_core.h pays ~25 000 cycles per plane whatever the size of the solid (its lists, scans and
// look-ahead are built for bands of thousands of vertices).  The Convex solids of the path -- the piece's hull clipped by a
// cell (k_clip_convex), a fragment's Convex clipped by its k-DOP slabs (k_refit) -- have a few dozen vertices and almost
// always clip "regularly": no vertex in a plane, no ring that lists a neighbour twice, every cap a simple loop.  For those this
// file follows the reference plane by plane, compaction included (:464-495), with the whole solid in LDS and no list at all:
//   classify (:307-318)            one lane per vertex; any in-plane vertex -> not regular
//   new vertices (:333-357)        numbered in (clipped vertex, slot) order by one prefix sum over the vertices
//   relink (:367-431)              every new vertex walks its face through clipped vertices to the next new vertex: its ring
//                                  is [predecessor, successor, kept end]; a walk that does not end on a new vertex, or two walks
//                                  ending on the same one -> not regular
//   compaction (:464-495)          kept vertices in order, then the new ones, rings renumbered, into the other LDS buffer
// "Not regular" (or too large) returns SC_FALLBACK before anything is published: the caller runs the general clipper on the
// same input, which handles in-plane vertices, doubled neighbours, two-neighbour collapses and the reference's error cases.
// Results are those of the general clipper bit for bit (same arithmetic: plane_dist / side_of / the intersection formula).
#pragma once
#include "clip_core.h"

#define SC_V 192u            // vertices of the solid at any time
#define SC_H 640u            // ring entries
#define SC_FALLBACK 101      // internal: not regular / does not fit -> general clipper

namespace surtr {

// The solid in LDS.  A ring entry carries, beside the neighbour's id, the slot of the reverse entry in the neighbour's ring (its
// "twin"), so that a face walk and the renumbering of the compaction never search a ring: entry = id | twin << 16.  The twins
// are found once, when the solid is loaded; a regular plane keeps them up to date for free -- the ring of a kept vertex keeps
// its length and order (:350-354 patch entries in place), and a new vertex [pred, succ, kept end] is entry 1 of its predecessor,
// entry 0 of its successor, and takes the place of its clipped end in the kept end's ring.
struct ScSolid
{
    float pos[3 * SC_V];
    uint32_t vw[SC_V];                    // ring offset | ring length << 16
    uint32_t re[SC_H];                    // neighbour | twin slot << 16
};

struct ScLds
{
    ScSolid buf[2];
    int8_t c[SC_V];                       // comp of the plane in progress
    uint16_t km[SC_V];                    // clipped vertex: bit j = ring slot j holds a kept neighbour
    uint16_t mp[SC_V];                    // kept vertex: its index after the compaction; clipped: 0x8000 | number of its first new vertex
    uint16_t noff[SC_V];                  // kept vertex: its ring offset after the compaction
    uint16_t succ[SC_V], pred[SC_V];      // per new vertex
    uint32_t nv[2], flag;
#ifdef SURTR_STAMP
    unsigned long long tph[8];            // lane 0's cycles: load, classify, numbering, relink, check, compaction; cutting planes
#endif
};

// Clips `in` by sh.planes[0..F).  0: done, the result is L.buf[*which] with L.nv[*which] vertices (0 = empty);
// SC_FALLBACK: use the general clipper.  One wave; every lane must call it.
// stop (optional): with SC_FALLBACK, the plane it stopped at when the solid before that plane -- L.buf[*which], L.nv[*which]
// vertices, exactly the reference's compacted solid after the planes before (:464-495) -- is there for the general clipper to go
// on from (sc_stage); 0xFFFFFFFF when there is none (the input itself is not regular).
__device__ inline int small_clip(const SolidIn in, const uint32_t F, const Shared& sh, ScLds& L, uint32_t* which, uint32_t* stop = nullptr)
{
    const uint32_t lane = lane_id();
    const uint32_t V = in.nv;
#ifdef SURTR_STAMP
    unsigned long long sc_t0 = __builtin_readcyclecounter();
    if (lane == 0) for (int q = 0; q < 8; ++q) L.tph[q] = 0;
#define SC_STAMP(i) do { if (lane == 0) { const unsigned long long t1 = __builtin_readcyclecounter(); L.tph[i] += t1 - sc_t0; sc_t0 = t1; } } while (0)
#else
#define SC_STAMP(i) do { } while (0)
#endif
    if (stop != nullptr) *stop = 0xFFFFFFFFu;
    if (V == 0 || V > SC_V) return SC_FALLBACK;
    const uint32_t hbase = in.loff[0], H = in.loff[V - 1u] + in.llen[V - 1u] - hbase;
    if (H > SC_H) return SC_FALLBACK;
    bool odd = false;
    for (uint32_t v = lane; v < V; v += SURTR_LANES)
    {
        ScSolid& S = L.buf[0];
        S.pos[3 * v] = in.pos[3 * v]; S.pos[3 * v + 1] = in.pos[3 * v + 1]; S.pos[3 * v + 2] = in.pos[3 * v + 2];
        const uint32_t lo = in.loff[v] - hbase, deg = in.llen[v];
        if (deg > 15u || deg < 3u) odd = true;
        const uint32_t dl = deg > 15u ? 15u : deg;
        S.vw[v] = lo | (dl << 16);
        const int32_t* r = in.nbr + in.loff[v];
        for (uint32_t j = 0; j < dl && lo + j < SC_H; ++j)
        {
            if ((uint32_t)r[j] >= V) odd = true;
            S.re[lo + j] = (uint32_t)r[j] & 0xFFFFu;
            // A ring that lists a vertex twice (slivers) makes the reference's first-occurrence patches and walks order
            // dependent: the general clipper reproduces that, this one does not try.  Checked once, here: a regular plane
            // cannot create such a ring in a kept vertex (its clipped neighbours become the distinct new vertices of distinct
            // edges), and a new vertex [pred, succ, kept] has one only when pred == succ, which the plane loop checks.
            for (uint32_t jj = 0; jj < j; ++jj) if (r[jj] == r[j]) odd = true;
        }
    }
    if (__ballot(odd) != 0ull) return SC_FALLBACK;
    __syncthreads();
    // the twins: slot of v in the ring of every neighbour (a neighbour that does not list v back: not regular)
    for (uint32_t v = lane; v < V; v += SURTR_LANES)
    {
        ScSolid& S = L.buf[0];
        const uint32_t w = S.vw[v], lo = w & 0xFFFFu, deg = w >> 16;
        for (uint32_t j = 0; j < deg; ++j)
        {
            const uint32_t e = S.re[lo + j] & 0xFFFFu;
            const uint32_t we = S.vw[e], elo = we & 0xFFFFu, edeg = we >> 16;
            uint32_t q = 0;
            while (q < edeg && (S.re[elo + q] & 0xFFFFu) != v) ++q;
            if (q >= edeg) odd = true;
            S.re[lo + j] = e | (q << 16);      // (only this lane writes the high half of its own entries; the others read the low half)
        }
    }
    if (__ballot(odd) != 0ull) return SC_FALLBACK;
    __syncthreads();
    SC_STAMP(0);
    // (nothing of L.buf[cur] has been touched when a plane gives up: the next solid is written to the other buffer)
#define SC_GIVE_UP do { if (stop != nullptr) { *stop = k; *which = cur; if (lane == 0) L.nv[cur] = nv; __syncthreads(); } return SC_FALLBACK; } while (0)
    uint32_t cur = 0, nv = V;
    for (uint32_t k = 0; k < F && nv != 0u; ++k)
    {
        const float4 pl = sh.planes[k];
        ScSolid& S = L.buf[cur]; ScSolid& N = L.buf[cur ^ 1u];
        // ---- classify ----
        bool zero = false, cut = false, keep = false;
        for (uint32_t v = lane; v < nv; v += SURTR_LANES)
        {
            const int c = side_of(plane_dist(pl, S.pos[3 * v], S.pos[3 * v + 1], S.pos[3 * v + 2]));
            L.c[v] = (int8_t)c;
            zero = zero || c == 0; cut = cut || c < 0; keep = keep || c > 0;
        }
        // (:303-328: "below" = no vertex strictly inside, tested first; "above" = none strictly outside.  Vertices in the plane
        //  matter only when the plane cuts: a slab plane through an extreme vertex of the solid is "above" like any other)
        // (every vertex in the plane -- a flat solid: the reference's answer then depends on its bounding-box shortcut, :297-301;
        //  the general clipper has that case)
        const bool anyCut = __ballot(cut) != 0ull;
        if (__ballot(keep) == 0ull) { if (!anyCut) SC_GIVE_UP; nv = 0; break; }      // "below": everything goes (:322-327)
        if (!anyCut) continue;                                       // "above": nothing to do (the solid has >= 4 vertices)
        if (__ballot(zero) != 0ull) SC_GIVE_UP;
        __syncthreads();
        SC_STAMP(1);
        // ---- new vertices in (clipped vertex, slot) order; kept vertices' new indices and ring offsets ----
        uint32_t carryM = 0, carryK = 0, carryH = 0;
        for (uint32_t v0 = 0; v0 < nv; v0 += SURTR_LANES)
        {
            const uint32_t v = v0 + lane;
            uint32_t mask = 0, isKept = 0, deg = 0;
            if (v < nv)
            {
                const uint32_t w = S.vw[v], lo = w & 0xFFFFu;
                deg = w >> 16;
                if (L.c[v] > 0) isKept = 1u;
                else
                    for (uint32_t j = 0; j < deg; ++j) if (L.c[S.re[lo + j] & 0xFFFFu] > 0) mask |= 1u << j;
            }
            const uint32_t cnt = (uint32_t)__builtin_popcount(mask);
            // one scan for the three running sums: new vertices | kept vertices << 16, kept ring entries
            const uint2 inc = wave_incl_scan2(make_uint2(cnt | (isKept << 16), isKept ? deg : 0u));
            if (v < nv)
            {
                L.km[v] = (uint16_t)mask;
                L.mp[v] = isKept ? (uint16_t)(carryK + (inc.x >> 16) - 1u) : (uint16_t)(0x8000u | (carryM + (inc.x & 0xFFFFu) - cnt));
                L.noff[v] = (uint16_t)(carryH + inc.y - (isKept ? deg : 0u));
            }
            const uint32_t tx = lane_bcast(inc.x, SURTR_LANES - 1u), ty = lane_bcast(inc.y, SURTR_LANES - 1u);
            carryM += tx & 0xFFFFu; carryK += tx >> 16; carryH += ty;
        }
        const uint32_t M = carryM, nKeep = carryK, HK = carryH;
        if (nKeep + M > SC_V || HK + 3u * M > SC_H) SC_GIVE_UP;
        if (nKeep + M < 4u) { nv = 0; break; }                        // (:497-499)
        for (uint32_t t = lane; t < M; t += SURTR_LANES) L.pred[t] = 0xFFFFu;
        __syncthreads();
        SC_STAMP(2);
        // ---- relink: successor of every new vertex along the face through its clipped end ----
        bool bad = false;
        for (uint32_t v = lane; v < nv; v += SURTR_LANES)
        {
            uint32_t mask = L.c[v] < 0 ? L.km[v] : 0u;
            if (!mask) continue;
            uint32_t t = L.mp[v] & 0x7FFFu;
            const uint32_t wv = S.vw[v];
            for (; mask; mask &= mask - 1u, ++t)
            {
                // new vertex X on the edge (v, slot j): FaceLoop from X through v takes the entry before slot j, and so on
                uint32_t cv = v, w = wv, slot = (uint32_t)__builtin_ctz(mask), steps = 0, end = 0xFFFFu;
                while (steps++ <= nv)
                {
                    const uint32_t len = w >> 16;
                    const uint32_t p = slot == 0u ? len - 1u : slot - 1u;
                    const uint32_t rw = S.re[(w & 0xFFFFu) + p];
                    const uint32_t e = rw & 0xFFFFu;
                    if (L.c[e] > 0) { end = (L.mp[cv] & 0x7FFFu) + (uint32_t)__builtin_popcount(L.km[cv] & ((1u << p) - 1u)); break; }
                    cv = e; slot = rw >> 16; w = S.vw[e];              // the walk arrives at clipped e from cv: the twin is the slot
                }
                if (end == 0xFFFFu || end == t) { bad = true; continue; }
                L.succ[t] = (uint16_t)end;
                L.pred[end] = (uint16_t)t;                             // two walks ending on `end`: one of them does not find itself below
            }
        }
        if (__ballot(bad) != 0ull) SC_GIVE_UP;
        __syncthreads();
        SC_STAMP(3);
        // (pred == succ: a cap of two vertices, i.e. a new ring that lists a vertex twice)
        for (uint32_t t = lane; t < M; t += SURTR_LANES) if (L.pred[t] == 0xFFFFu || L.pred[L.succ[t]] != t || L.pred[t] == L.succ[t]) bad = true;
        if (__ballot(bad) != 0ull) SC_GIVE_UP;
        SC_STAMP(4);
        // ---- the solid after this plane, compacted (:464-495): kept vertices in order, then the new ones ----
        for (uint32_t v = lane; v < nv; v += SURTR_LANES)
        {
            const uint32_t w = S.vw[v], lo0 = w & 0xFFFFu, deg = w >> 16;
            if (L.c[v] > 0)
            {
                const uint32_t id = L.mp[v], lo = L.noff[v];
                N.pos[3 * id] = S.pos[3 * v]; N.pos[3 * id + 1] = S.pos[3 * v + 1]; N.pos[3 * id + 2] = S.pos[3 * v + 2];
                N.vw[id] = lo | (deg << 16);
                for (uint32_t j = 0; j < deg; ++j)
                {
                    const uint32_t rw = S.re[lo0 + j], e = rw & 0xFFFFu, q = rw >> 16;
                    const uint32_t m = L.mp[e], kme = L.km[e];
                    // a kept neighbour under its new number, same twin; the link to a clipped neighbour now holds the new vertex on
                    // that edge (:350-354), whose third entry this vertex is
                    N.re[lo + j] = (m & 0x8000u) ? (nKeep + (m & 0x7FFFu) + (uint32_t)__builtin_popcount(kme & ((1u << q) - 1u))) | (2u << 16)
                                                 : m | (q << 16);
                }
            }
            else
            {
                const float ax = S.pos[3 * v], ay = S.pos[3 * v + 1], az = S.pos[3 * v + 2];
                const float sa = plane_dist(pl, ax, ay, az);
                uint32_t t = L.mp[v] & 0x7FFFu;
                for (uint32_t mask = L.km[v]; mask; mask &= mask - 1u, ++t)
                {
                    const uint32_t rw = S.re[lo0 + (uint32_t)__builtin_ctz(mask)], u = rw & 0xFFFFu;
                    const float bx = S.pos[3 * u], by = S.pos[3 * u + 1], bz = S.pos[3 * u + 2];
                    const float sb = plane_dist(pl, bx, by, bz);
                    // PlaneLineIntersection (:746-751): (a*sb - b*sa) * (1/(sb-sa))
                    const float inv = 1.f / (sb - sa);
                    const uint32_t id = nKeep + t, lo = HK + 3u * t;
                    N.pos[3 * id] = (ax * sb - bx * sa) * inv;
                    N.pos[3 * id + 1] = (ay * sb - by * sa) * inv;
                    N.pos[3 * id + 2] = (az * sb - bz * sa) * inv;
                    N.vw[id] = lo | (3u << 16);
                    N.re[lo] = (nKeep + L.pred[t]) | (1u << 16); N.re[lo + 1] = (nKeep + L.succ[t]) | (0u << 16);
                    N.re[lo + 2] = (uint32_t)L.mp[u] | (rw & 0xFFFF0000u);      // (the kept end lists this vertex where it listed v)
                }
            }
        }
        nv = nKeep + M;
        cur ^= 1u;
        __syncthreads();
        SC_STAMP(5);
#ifdef SURTR_STAMP
        if (lane == 0) L.tph[6] += 1;
#endif
    }
#undef SC_GIVE_UP
    SC_STAMP(1);
    *which = cur;
    if (lane == 0) L.nv[cur] = nv;
    __syncthreads();
    return 0;
}

// The solid small_clip stopped with, as an input of the general clipper: positions and rings into global staging arrays
// (`ring` is indexed by the offsets written to `loff`).  One wave; ends with a barrier.
__device__ inline SolidIn sc_stage(const ScSolid& S, uint32_t nv, float* pos, uint32_t* loff, uint32_t* llen, uint32_t* ring)
{
    for (uint32_t v = lane_id(); v < nv; v += SURTR_LANES)
    {
        pos[3 * v] = S.pos[3 * v]; pos[3 * v + 1] = S.pos[3 * v + 1]; pos[3 * v + 2] = S.pos[3 * v + 2];
        const uint32_t w = S.vw[v], lo = w & 0xFFFFu, len = w >> 16;
        loff[v] = lo; llen[v] = len;
        for (uint32_t q = 0; q < len; ++q) ring[lo + q] = S.re[lo + q] & 0xFFFFu;
    }
    __threadfence_block();
    __syncthreads();
    return SolidIn{pos, loff, llen, (const int32_t*)ring, nv, nullptr, nullptr, nullptr, nullptr, nullptr};
}

// Writes the result of small_clip to the arena as one packed solid (the layout park_topo writes).  One wave.
__device__ inline int sc_park(const ScSolid& S, uint32_t nv, Shared& sh, uint32_t* cursors, float* apos, uint32_t* aloff, uint32_t* allen, int32_t* anbr,
                              uint32_t capV, uint32_t capH, uint32_t& voff, uint32_t& n, uint32_t& hoff, uint32_t& nh)
{
    const uint32_t wl = S.vw[nv - 1u];
    const uint32_t H = (wl & 0xFFFFu) + (wl >> 16);
    __syncthreads();
    if (threadIdx.x == 0) { sh.misc[0] = atomicAdd(&cursors[0], nv); sh.misc[1] = atomicAdd(&cursors[1], H); }
    __syncthreads();
    voff = sh.misc[0]; hoff = sh.misc[1];
    __syncthreads();
    if ((uint64_t)voff + nv > capV || (uint64_t)hoff + H > capH) return SURTR_E_CAPACITY;
    for (uint32_t v = lane_id(); v < nv; v += SURTR_LANES)
    {
        const size_t id = (size_t)voff + v;
        apos[3 * id] = S.pos[3 * v]; apos[3 * id + 1] = S.pos[3 * v + 1]; apos[3 * id + 2] = S.pos[3 * v + 2];
        const uint32_t w = S.vw[v];
        const uint32_t lo = hoff + (w & 0xFFFFu), len = w >> 16;
        aloff[id] = lo; allen[id] = len;
        const uint32_t* r = S.re + (w & 0xFFFFu);
        for (uint32_t q = 0; q < len; ++q) anbr[lo + q] = (int32_t)(r[q] & 0xFFFFu);
    }
    n = nv; nh = H;
    __syncthreads();
    return 0;
}

} // namespace surtr
