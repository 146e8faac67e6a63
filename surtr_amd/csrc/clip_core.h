// clip_core.h -- device-side core of the MI355X fracture engine (gfx950, wave64).
//
// One 256-thread workgroup owns one (cell, piece) task and reproduces
// Poly::ClipPolyhedron (reference Src/Poly.cpp:265-500) with the same output
// vertex order, neighbour-ring order and float arithmetic:
//
//   pre-pass   every original vertex is classified against all planes of the
//              cell at once ("first plane that cuts it"); vertices whose whole
//              1-ring is cut by the same plane can never be touched by the
//              sequential algorithm and are dropped ("band
//              reduction" -- a result-preserving cull, not an approximation);
//   per plane  classify / create edge-cut vertices in (vertex, slot) order by
//              ballot + prefix sums / relink the cap by face-loop walks /
//              order-preserving compaction -- the steps of :307-495;
//   fallback   a plane that puts any vertex exactly in-plane (comp == 0), or any
//              irregular cap, runs the reference's relink/collapse sequence
//              literally on one lane of the same workgroup (still on the GPU).
//
// All arithmetic is float32 without contraction (compile with -ffp-contract=off);
// see oracle/surtr_oracle.cpp for the SimpleMath semantics restated.
#pragma once
#ifdef SURTR_EMUL
#include "hip_emul.h"        // tests/emul: single-lane CPU emulation, test infrastructure only
#define SURTR_LANES 1
#define SURTR_LSH 0
#define SURTR_WG 1
#define SURTR_NWAVE 1
#else
#include <hip/hip_runtime.h>
#define SURTR_LANES 64
#define SURTR_LSH 6
#ifndef SURTR_WG
#define SURTR_WG 256
#endif
#define SURTR_NWAVE (SURTR_WG / 64)
#endif
#include <stdint.h>
#ifdef SURTR_EMUL
#include <cstdio>
#define SURTR_DBG(...) fprintf(stderr, __VA_ARGS__)
#else
#define SURTR_DBG(...)
#endif

// Diagnostic build only (-DSURTR_STAMP): lane 0 accumulates s_memtime deltas per phase into a
// global table that no product code reads.
#if defined(SURTR_STAMP) && !defined(SURTR_EMUL)
__device__ unsigned long long g_stamp[32];
#define STAMP_DECL unsigned long long st_t0 = __builtin_readcyclecounter(); unsigned long long st_t1
#define STAMP(i) do { if (threadIdx.x == 0) { st_t1 = __builtin_readcyclecounter(); atomicAdd(&g_stamp[i], st_t1 - st_t0); st_t0 = st_t1; } } while (0)
#define COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_stamp[i], 1ull); } while (0)
#else
#define COUNT(i) do { } while (0)
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#endif

#define SURTR_MAXF 255
#define SURTR_SENT (-2)
#define SURTR_LDS_BLOCKS 1024   // pre-pass masks live in LDS for solids up to 64*1024 vertices

namespace surtr {

struct Buf
{
    float* pos;       // 3 floats per vertex
    uint32_t* loff;   // start of the vertex's ring in nbr
    uint32_t* llen;   // ring length
    int8_t* comp;     // ComparePlanePoint result of the current plane (2 = created by it)
    int32_t* nbr;     // ring entries; SURTR_SENT = a dropped original vertex, -1 = marked for removal
};

struct Scratch
{
    Buf b[2];
    uint32_t* aux0;   // [CV]
    uint32_t* aux1;   // [CV]
    uint32_t* aux2;   // [CV]
    unsigned long long* gmask;   // [VMAX/64] pre-pass band mask when the solid is too big for the LDS copy
    uint2* blk;       // per-64-block (count, weight) of the ordered scans
    uint32_t CV, CH;
};

struct Shared
{
    float4 planes[SURTR_MAXF + 1];
    uint32_t hist[SURTR_MAXF + 1];
    uint32_t zhist[SURTR_MAXF + 1];   // dropped vertices that lie in plane k while still alive
    uint32_t wsum[2 * SURTR_NWAVE];
    uint32_t flagCut, flagKeep, flagZero, flagBad, flagErr;
    uint32_t hend;
    uint32_t nodrop;
    uint32_t changed;
    uint32_t misc[8];
    unsigned long long bmask[SURTR_LDS_BLOCKS];   // band bit mask of the pre-pass, one word per 64 vertices
    uint2 bblk[SURTR_LDS_BLOCKS];                 // (count, ring entries) bases per 64 vertices
};

// A solid handed to the clipper: positions + (loff, llen) rings with entries local to the solid.
struct SolidIn
{
    const float* pos;
    const uint32_t* loff;
    const uint32_t* llen;
    const int32_t* nbr;     // already offset so that loff indexes it directly
    uint32_t nv;
    const uint8_t* tri;     // per vertex: 1 = every incident face is a triangle (nullptr = unknown)
};

__device__ __forceinline__ float plane_dist(const float4 pl, float x, float y, float z)
{
    float t = pl.x * x + pl.y * y;
    t = t + pl.z * z;
    return pl.w + t;
}

// ComparePlanePoint, Src/Poly.cpp:716-723.
__device__ __forceinline__ int side_of(float s)
{
    if (fabs((double)s) < 1.0e-10) return 0;
    float m = -s;
    return m > 0.f ? 1 : (m < 0.f ? -1 : 0);
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (SURTR_LANES - 1u); }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> SURTR_LSH; }

__device__ __forceinline__ uint2 wave_incl_scan2(uint2 v)
{
    const uint32_t l = lane_id();
#pragma unroll
    for (int d = 1; d < SURTR_LANES; d <<= 1)
    {
        uint32_t a = __shfl_up(v.x, d, SURTR_LANES);
        uint32_t b = __shfl_up(v.y, d, SURTR_LANES);
        if (l >= (uint32_t)d) { v.x += a; v.y += b; }
    }
    return v;
}

// Ordered two-level scan over items [0,n): phase 1+2.  fn(i) -> (count, weight).
// After the call blk[b] holds the exclusive (count, weight) base of 64-block b
// and (totC, totW) the totals.  Every thread must call it (barriers inside).
// Exclusive scan, in place, of the per-64-block (count, weight) array; returns the totals.
__device__ inline void scan_block_array(uint32_t nb, uint2* blk, Shared& sh, uint32_t& totC, uint32_t& totW)
{
    const uint32_t l = lane_id(), w = wave_id();
    uint2 carry = make_uint2(0u, 0u);
    for (uint32_t c0 = 0; c0 < nb; c0 += SURTR_WG)
    {
        const uint32_t i = c0 + threadIdx.x;
        uint2 x = make_uint2(0u, 0u);
        if (i < nb) x = blk[i];
        uint2 inc = wave_incl_scan2(x);
        if (l == SURTR_LANES - 1u) { sh.wsum[2 * w] = inc.x; sh.wsum[2 * w + 1] = inc.y; }
        __syncthreads();
        uint2 woff = make_uint2(0u, 0u), tot = make_uint2(0u, 0u);
#pragma unroll
        for (uint32_t q = 0; q < SURTR_NWAVE; ++q)
        {
            const uint32_t a = sh.wsum[2 * q], bq = sh.wsum[2 * q + 1];
            if (q < w) { woff.x += a; woff.y += bq; }
            tot.x += a; tot.y += bq;
        }
        if (i < nb) blk[i] = make_uint2(carry.x + woff.x + inc.x - x.x, carry.y + woff.y + inc.y - x.y);
        carry.x += tot.x; carry.y += tot.y;
        __syncthreads();
    }
    totC = carry.x; totW = carry.y;
}

// Ordered two-level scan over items [0,n): phase 1+2.  fn(i) -> (count, weight).
// After the call blk[b] holds the exclusive (count, weight) base of 64-block b
// and (totC, totW) the totals.  Every thread must call it (barriers inside).
template <class Fn>
__device__ void scan_blocks(uint32_t n, uint2* blk, Shared& sh, Fn fn, uint32_t& totC, uint32_t& totW)
{
    const uint32_t nb = (n + SURTR_LANES - 1u) >> SURTR_LSH;
    const uint32_t l = lane_id(), w = wave_id();
    for (uint32_t b = w; b < nb; b += SURTR_NWAVE)
    {
        const uint32_t i = (b << SURTR_LSH) + l;
        uint2 c = make_uint2(0u, 0u);
        if (i < n) c = fn(i);
        uint2 s = wave_incl_scan2(c);
        if (l == SURTR_LANES - 1u) blk[b] = s;
    }
    __syncthreads();
    scan_block_array(nb, blk, sh, totC, totW);
}

// Phase 3 helper: exclusive position of item i inside its 64-block (call with the whole wave).
__device__ __forceinline__ uint2 wave_excl2(uint2 c)
{
    uint2 s = wave_incl_scan2(c);
    return make_uint2(s.x - c.x, s.y - c.y);
}

// FaceLoop, Src/Poly.cpp:34-41.
__device__ __forceinline__ int32_t face_next(const int32_t* ring, uint32_t len, int32_t prev)
{
    uint32_t k = 0;
    while (k < len && ring[k] != prev) ++k;
    if (k == 0) return ring[len - 1];
    return ring[k - 1];
}

__device__ __forceinline__ int comp_of(const Buf& B, int32_t v) { return v < 0 ? 0 : (int)B.comp[v]; }

// ---------------------------------------------------------------------------
// Serial tail of one plane (Src/Poly.cpp:367-462) run by one lane when the plane
// has in-plane vertices or an irregular cap.  Rings of comp 0 / comp 2 vertices
// have been given room for the insertions; snap = old_neighbors of comp-0 vertices.
__device__ void relink_serial(Buf& B, uint32_t n0, uint32_t n1, const uint32_t* snapoff, const uint32_t* cap,
                              const uint32_t* zlist, uint32_t nz, Shared& sh)
{
    // visiting order of :370-373: the new vertices, then the in-plane ones (ascending)
    const uint32_t M = n1 - n0;
    for (uint32_t t = 0; t < M + nz; ++t)
    {
        const uint32_t i = t < M ? n0 + t : zlist[t - M];
        const uint32_t deg = B.llen[i];
        for (uint32_t j = 0; j < deg; ++j)
        {
            int32_t* ri = B.nbr + B.loff[i];
            const int32_t jn = ri[j];
            if (jn < 0 || B.comp[jn] != -1) continue;
            int32_t prev = (int32_t)i, cur = jn;
            uint32_t steps = 0;
            while (cur >= 0 && B.comp[cur] == -1 && steps++ < n1)
            {
                int32_t hold = cur;
                cur = face_next(B.nbr + B.loff[cur], B.llen[cur], prev);
                prev = hold;
            }
            if (cur < 0) { SURTR_DBG("serial: walk hit sentinel i=%u\n", i); sh.flagErr = 1; return; }
            if (ri[(j + 1u) % B.llen[i]] == cur || cur == (int32_t)i)
            {
                ri[j] = -1;
            }
            else
            {
                ri[j] = cur;
                int32_t* rc = B.nbr + B.loff[cur];
                const uint32_t lc = B.llen[cur];
                if (B.comp[cur] == 2)
                {
                    if (lc >= 3u) { SURTR_DBG("serial: comp2 ring full cur=%d i=%u n0=%u n1=%u\n", cur, i, n0, n1); sh.flagErr = 1; return; }     // room reserved for one insertion only
                    for (uint32_t q = lc; q > 0; --q) rc[q] = rc[q - 1];
                    rc[0] = (int32_t)i;
                    B.llen[cur] = lc + 1;
                }
                else if (B.comp[cur] == 0 && (uint32_t)cur < n0)
                {
                    if (lc >= cap[cur]) { SURTR_DBG("serial: comp0 ring full cur=%d\n", cur); sh.flagErr = 1; return; }
                    int32_t* sn = B.nbr + snapoff[cur];
                    uint32_t at = 0;
                    while (at < lc && sn[at] != prev) ++at;
                    for (uint32_t q = lc; q > at; --q) { rc[q] = rc[q - 1]; sn[q] = sn[q - 1]; }
                    rc[at] = (int32_t)i; sn[at] = (int32_t)i;
                    B.llen[cur] = lc + 1;
                }
                else { SURTR_DBG("serial: walk ended on comp %d vertex %d (i=%u j=%u n0=%u n1=%u steps=%u)\n", (int)B.comp[cur], cur, i, j, n0, n1, steps); sh.flagErr = 1; return; }   // walk ended on a kept or clipped vertex
            }
        }
    }
}

// Two-neighbour vertices, Src/Poly.cpp:433-462, literal and serial (rare).
__device__ void collapse_serial(Buf& B, uint32_t n1)
{
    bool again = true;
    while (again)
    {
        again = false;
        for (uint32_t i = 0; i < n1; ++i)
        {
            if (B.comp[i] >= 0 && B.llen[i] == 2u)
            {
                again = true;
                const int32_t a = B.nbr[B.loff[i]], b = B.nbr[B.loff[i] + 1];
                if (a >= 0)
                {
                    int32_t* ra = B.nbr + B.loff[a];
                    for (uint32_t q = 0; q < B.llen[a]; ++q) if (ra[q] == (int32_t)i) { ra[q] = b; break; }
                }
                if (b >= 0)
                {
                    int32_t* rb = B.nbr + B.loff[b];
                    for (uint32_t q = 0; q < B.llen[b]; ++q) if (rb[q] == (int32_t)i) { rb[q] = a; break; }
                }
                B.comp[i] = -1;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Clips `in` by sh.planes[0..F).  The result is left in S.b[*outBuf] with
// *outN vertices (0 = empty) and rings packed in vertex order (loff is the
// exclusive scan of llen).  Returns 0 or an error code (uniform over the group).
__device__ int clip_solid(const SolidIn in, const uint32_t F, Scratch& S, Shared& sh, uint32_t* outN, uint32_t* outBuf,
                          uint32_t* outH)
{
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id();
    const uint32_t V = in.nv;
    *outN = 0; *outBuf = 0; *outH = 0;
    STAMP_DECL;

    // ---- pre-pass: which original vertices can the sequential algorithm ever touch? ----------
    // fc(v) = first plane that cuts v (0xFF: none).  A vertex is dropped when every vertex of every
    // incident face has the same finite fc: those faces are never walked by the relink step and the
    // vertex only disappears at plane fc (DESIGN.md "band reduction").  Nothing is stored per vertex:
    // fc of a neighbour is re-derived from its position (planes are tested in order, so "same fc" costs
    // fc+1 plane evaluations), and the survivors' new indices come from a bit mask + popcounts in LDS.
    for (uint32_t k = tid; k <= SURTR_MAXF; k += SURTR_WG) { sh.hist[k] = 0; sh.zhist[k] = 0; }
    if (tid == 0) { sh.nodrop = 0; sh.flagErr = 0; }
    __syncthreads();
    const uint32_t nbV = (V + SURTR_LANES - 1u) >> SURTR_LSH;
    unsigned long long* bmask = (nbV <= SURTR_LDS_BLOCKS) ? sh.bmask : S.gmask;
    uint2* bblk = (nbV <= SURTR_LDS_BLOCKS) ? sh.bblk : S.blk;
    auto first_cut = [&](float x, float y, float z, bool& zero) -> uint32_t {
        for (uint32_t k = 0; k < F; ++k)
        {
            const int c = side_of(plane_dist(sh.planes[k], x, y, z));
            if (c == 0) zero = true;
            if (c < 0) return k;
        }
        return 0xFFu;
    };
    auto same_fc = [&](int32_t u, uint32_t f) -> bool {
        const float x = in.pos[3 * u], y = in.pos[3 * u + 1], z = in.pos[3 * u + 2];
        for (uint32_t k = 0; k < f; ++k)
            if (side_of(plane_dist(sh.planes[k], x, y, z)) < 0) return false;
        return side_of(plane_dist(sh.planes[f], x, y, z)) < 0;
    };
    {
        for (uint32_t b = w; b < nbV; b += SURTR_NWAVE)
        {
            const uint32_t v = (b << SURTR_LSH) + l;
            bool keep = false, zero = false; uint32_t deg = 0;
            if (v < V)
            {
                const uint32_t f = first_cut(in.pos[3 * v], in.pos[3 * v + 1], in.pos[3 * v + 2], zero);
                deg = in.llen[v];
                keep = f == 0xFFu;
                if (!keep)
                {
                    const int32_t* r = in.nbr + in.loff[v];
                    // neighbours in groups of 8 with all loads issued before any is used (latency, not bandwidth, rules here)
                    for (uint32_t j0 = 0; j0 < deg && !keep; j0 += 8)
                    {
                        int32_t u[8]; float ux[8], uy[8], uz[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) u[q] = (j0 + q < deg) ? r[j0 + q] : -1;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                        {
                            const int32_t uu = u[q] < 0 ? (int32_t)v : u[q];
                            ux[q] = in.pos[3 * uu]; uy[q] = in.pos[3 * uu + 1]; uz[q] = in.pos[3 * uu + 2];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                        {
                            if (u[q] < 0) continue;
                            bool same = true;
                            for (uint32_t k = 0; k < f && same; ++k)
                                if (side_of(plane_dist(sh.planes[k], ux[q], uy[q], uz[q])) < 0) same = false;
                            if (same && !(side_of(plane_dist(sh.planes[f], ux[q], uy[q], uz[q])) < 0)) same = false;
                            if (!same) keep = true;
                        }
                    }
                    if (!keep && !(in.tri != nullptr && in.tri[v]))
                    {
                        for (uint32_t j = 0; j < deg && !keep; ++j)
                        {
                            int32_t prev = (int32_t)v, cur = r[j];
                            uint32_t steps = 0;
                            while (cur != (int32_t)v && steps++ < V)
                            {
                                if (!same_fc(cur, f)) { keep = true; break; }
                                const int32_t nx = face_next(in.nbr + in.loff[cur], in.llen[cur], prev);
                                prev = cur; cur = nx;
                            }
                        }
                    }
                    if (!keep)
                    {
                        atomicAdd(&sh.hist[f], 1u);
                        if (zero)     // in-plane at an earlier plane while alive: it is no "kept" vertex there
                            for (uint32_t k = 0; k < f; ++k)
                                if (side_of(plane_dist(sh.planes[k], in.pos[3 * v], in.pos[3 * v + 1], in.pos[3 * v + 2])) == 0)
                                    atomicAdd(&sh.zhist[k], 1u);
                    }
                }
            }
            const uint2 c = keep ? make_uint2(1u, deg) : make_uint2(0u, 0u);
            const uint2 inc = wave_incl_scan2(c);
#ifdef SURTR_EMUL
            const unsigned long long m = keep ? 1ull : 0ull;
#else
            const unsigned long long m = __ballot(keep);
#endif
            if (l == SURTR_LANES - 1u) { bblk[b] = inc; bmask[b] = m; }
        }
    }
    __syncthreads();
    STAMP(0);
    uint32_t n = 0, hsum = 0;
    scan_block_array(nbV, bblk, sh, n, hsum);
    STAMP(1);
    if (n > S.CV || hsum > S.CH) return 3;
    // index of an original vertex in the reduced solid
    auto newid = [&](int32_t u) -> int32_t {
        const unsigned long long m = bmask[(uint32_t)u >> SURTR_LSH];
        const uint32_t bit = (uint32_t)u & (SURTR_LANES - 1u);
        if (!((m >> bit) & 1ull)) return SURTR_SENT;
        return (int32_t)(bblk[(uint32_t)u >> SURTR_LSH].x + (uint32_t)__builtin_popcountll(m & ((1ull << bit) - 1ull)));
    };
    {
        Buf& A = S.b[0];
        for (uint32_t b = w; b < nbV; b += SURTR_NWAVE)
        {
            const unsigned long long m = bmask[b];
            if (m == 0ull) continue;
            const uint32_t v = (b << SURTR_LSH) + l;
            const bool keep = (m >> l) & 1ull;
            const uint32_t deg = keep ? in.llen[v] : 0u;
            const uint2 e = wave_excl2(make_uint2(keep ? 1u : 0u, deg));
            if (keep)
            {
                const uint2 base = bblk[b];
                const uint32_t id = base.x + e.x;
                A.pos[3 * id] = in.pos[3 * v]; A.pos[3 * id + 1] = in.pos[3 * v + 1]; A.pos[3 * id + 2] = in.pos[3 * v + 2];
                const uint32_t lo = base.y + e.y;
                A.loff[id] = lo; A.llen[id] = deg; A.comp[id] = 1;
                const int32_t* r = in.nbr + in.loff[v];
                int32_t* d = A.nbr + lo;
                for (uint32_t j = 0; j < deg; ++j) d[j] = newid(r[j]);
            }
        }
    }
    STAMP(2);
    // dropAlive[k] = dropped vertices still alive after plane k = sum_{f>k} hist[f]
    __syncthreads();
    if (tid == 0)
    {
        uint32_t run = 0;
        for (int k = (int)F - 1; k >= 0; --k) { const uint32_t h = sh.hist[k]; sh.hist[k] = run; run += h; }
        // hist[k] now = number of dropped vertices with fc > k
    }
    __syncthreads();
    STAMP(3);
    if (n == 0) return 0;

    uint32_t cur = 0;
    uint32_t hcur = hsum;
    for (uint32_t k = 0; k < F; ++k)
    {
        Buf& B = S.b[cur];
        const float4 pl = sh.planes[k];
        __syncthreads();     // every lane has read the previous plane's flags before they are reset
        if (tid == 0) { sh.flagCut = 0; sh.flagKeep = 0; sh.flagZero = 0; sh.flagBad = 0; }
        __syncthreads();
        // ---- classify (:307-318) ----
        {
            bool anyc = false, anyk = false, anyz = false;
            for (uint32_t v = tid; v < n; v += SURTR_WG)
            {
                const int c = side_of(plane_dist(pl, B.pos[3 * v], B.pos[3 * v + 1], B.pos[3 * v + 2]));
                B.comp[v] = (int8_t)c;
                anyc |= c < 0; anyk |= c > 0; anyz |= c == 0;
            }
            if (anyc) sh.flagCut = 1;
            if (anyk) sh.flagKeep = 1;
            if (anyz) sh.flagZero = 1;
        }
        __syncthreads();
        STAMP(4);
        const bool anyCut = sh.flagCut != 0, anyKeep = sh.flagKeep != 0, anyZero = sh.flagZero != 0;
        const uint32_t dropAlive = sh.hist[k];
        const uint32_t dropKept = dropAlive - sh.zhist[k];   // dropped vertices strictly on the kept side of this plane
        if (!anyCut && !anyKeep && dropKept == 0)
        {
            // Every vertex is in-plane (e.g. a zero plane from a degenerate hull face).  The reference
            // consults the bounding box first (:296-299): all corners >= 0 skips the plane, anything
            // else ends in "below" (:322-327).
            if (tid == 0)
            {
                float lo[3] = {B.pos[0], B.pos[1], B.pos[2]}, hi[3] = {B.pos[0], B.pos[1], B.pos[2]};
                for (uint32_t v = 1; v < n; ++v)
                    for (int a = 0; a < 3; ++a)
                    {
                        const float c = B.pos[3 * v + a];
                        lo[a] = c < lo[a] ? c : lo[a]; hi[a] = c > hi[a] ? c : hi[a];
                    }
                if (dropAlive != 0)
                {
                    // dropped vertices still alive (all in-plane here) belong to the box too
                    for (uint32_t v = 0; v < V; ++v)
                    {
                        if ((bmask[v >> SURTR_LSH] >> (v & (SURTR_LANES - 1u))) & 1ull) continue;
                        bool z0 = false;
                        if (first_cut(in.pos[3 * v], in.pos[3 * v + 1], in.pos[3 * v + 2], z0) <= k) continue;
                        for (int a = 0; a < 3; ++a)
                        {
                            const float c = in.pos[3 * v + a];
                            lo[a] = c < lo[a] ? c : lo[a]; hi[a] = c > hi[a] ? c : hi[a];
                        }
                    }
                }
                int cmin = 1;
                for (int q = 0; q < 8; ++q)
                {
                    const int c = side_of(plane_dist(pl, (q & 1) ? hi[0] : lo[0], (q & 2) ? hi[1] : lo[1], (q & 4) ? hi[2] : lo[2]));
                    cmin = c < cmin ? c : cmin;
                }
                sh.misc[0] = cmin >= 0 ? 1u : 0u;
            }
            __syncthreads();
            const bool boxAbove = sh.misc[0] != 0;
            __syncthreads();
            if (boxAbove) continue;
            n = 0; break;
        }
        if (!anyKeep && dropKept == 0) { n = 0; break; }       // "below": everything goes (:322-327)
        if (!anyCut)
        {
            // "above" for the reduced solid; dropped vertices may still vanish here (size check :497-499)
            if (n + dropAlive < 4u) { n = 0; break; }
            continue;
        }

        // ---- new vertices on straddling edges, in (vertex, slot) order (:333-357) ----
        auto cutfn = [&](uint32_t v) -> uint2 {
            if (B.comp[v] >= 0) return make_uint2(0u, 0u);
            const int32_t* r = B.nbr + B.loff[v];
            const uint32_t deg = B.llen[v];
            uint32_t c = 0;
            for (uint32_t j = 0; j < deg; ++j) { const int32_t u = r[j]; if (u >= 0 && B.comp[u] > 0) ++c; }
            return make_uint2(c, 0u);
        };
        uint32_t M = 0, dummy = 0;
        scan_blocks(n, S.blk, sh, cutfn, M, dummy);
        const uint32_t n0 = n, n1 = n + M;
        if (n1 > S.CV || hcur + 3u * M > S.CH) return 3;
        {
            const uint32_t nb = (n0 + SURTR_LANES - 1u) >> SURTR_LSH;
            for (uint32_t b = w; b < nb; b += SURTR_NWAVE)
            {
                const uint32_t v = (b << SURTR_LSH) + l;
                uint2 c = make_uint2(0u, 0u);
                if (v < n0) c = cutfn(v);
                const uint2 e = wave_excl2(c);
                if (v < n0 && c.x)
                {
                    uint32_t fresh = n0 + S.blk[b].x + e.x;
                    int32_t* r = B.nbr + B.loff[v];
                    const uint32_t deg = B.llen[v];
                    const float ax = B.pos[3 * v], ay = B.pos[3 * v + 1], az = B.pos[3 * v + 2];
                    const float sa = plane_dist(pl, ax, ay, az);
                    for (uint32_t j = 0; j < deg; ++j)
                    {
                        const int32_t u = r[j];
                        if (u < 0 || B.comp[u] <= 0) continue;
                        const float bx = B.pos[3 * u], by = B.pos[3 * u + 1], bz = B.pos[3 * u + 2];
                        const float sb = plane_dist(pl, bx, by, bz);
                        // PlaneLineIntersection (:746-751): (a*sb - b*sa) * (1/(sb-sa))
                        const float inv = 1.f / (sb - sa);
                        B.pos[3 * fresh] = (ax * sb - bx * sa) * inv;
                        B.pos[3 * fresh + 1] = (ay * sb - by * sa) * inv;
                        B.pos[3 * fresh + 2] = (az * sb - bz * sa) * inv;
                        B.comp[fresh] = 2;
                        const uint32_t lo = hcur + 3u * (fresh - n0);
                        B.loff[fresh] = lo; B.llen[fresh] = 2;
                        B.nbr[lo] = (int32_t)v; B.nbr[lo + 1] = u; B.nbr[lo + 2] = -1;
                        int32_t* ru = B.nbr + B.loff[u];
                        const uint32_t du = B.llen[u];
                        for (uint32_t q = 0; q < du; ++q) if (ru[q] == (int32_t)v) { ru[q] = (int32_t)fresh; break; }
                        r[j] = (int32_t)fresh;
                        ++fresh;
                    }
                }
            }
        }
        uint32_t hend = hcur + 3u * M;
        __syncthreads();
        STAMP(5);

        // ---- relink (:367-431) ----
        bool serial = anyZero;
        if (!serial)
        {
            // regular cap: every new vertex X=[cut, kept] finds its successor by walking the face
            // loop through clipped vertices; its final ring is [pred, succ, kept].
            uint32_t* succ = S.aux0; uint32_t* pred = S.aux1; uint32_t* pcnt = S.aux2;
            for (uint32_t t = tid; t < M; t += SURTR_WG) pcnt[t] = 0;
            __syncthreads();
            bool bad = false;
            for (uint32_t t = tid; t < M; t += SURTR_WG)
            {
                const uint32_t X = n0 + t;
                int32_t prev = (int32_t)X, c = B.nbr[B.loff[X]];
                uint32_t steps = 0;
                while (c >= 0 && B.comp[c] == -1 && steps++ < n1)
                {
                    const int32_t hold = c;
                    c = face_next(B.nbr + B.loff[c], B.llen[c], prev);
                    prev = hold;
                }
                if (c < (int32_t)n0 || c == (int32_t)X || B.comp[c] != 2) { bad = true; succ[t] = X; }
                else
                {
                    succ[t] = (uint32_t)c;
                    atomicAdd(&pcnt[(uint32_t)c - n0], 1u);
                    pred[(uint32_t)c - n0] = X;
                }
            }
            if (bad) sh.flagBad = 1;
            __syncthreads();
            bad = false;
            for (uint32_t t = tid; t < M; t += SURTR_WG) if (pcnt[t] != 1u) bad = true;
            if (bad) sh.flagBad = 1;
            __syncthreads();
            serial = sh.flagBad != 0;
            if (!serial)
            {
                for (uint32_t t = tid; t < M; t += SURTR_WG)
                {
                    const uint32_t lo = B.loff[n0 + t];
                    const int32_t kept = B.nbr[lo + 1];
                    B.nbr[lo] = (int32_t)pred[t]; B.nbr[lo + 1] = (int32_t)succ[t]; B.nbr[lo + 2] = kept;
                    B.llen[n0 + t] = 3;
                }
            }
        }
        if (serial)
        {
            COUNT(19);
            // give every in-plane vertex a ring with room for insertions plus its snapshot
            uint32_t* snapoff = S.aux0; uint32_t* cap = S.aux1; uint32_t* zlist = S.aux2;
            auto zfn = [&](uint32_t v) -> uint2 {
                return (B.comp[v] == 0) ? make_uint2(1u, 4u * B.llen[v]) : make_uint2(0u, 0u);
            };
            uint32_t zc = 0, zw = 0;
            scan_blocks(n0, S.blk, sh, zfn, zc, zw);
            if (hend + zw > S.CH) return 3;
            const uint32_t nb = (n0 + SURTR_LANES - 1u) >> SURTR_LSH;
            for (uint32_t b = w; b < nb; b += SURTR_NWAVE)
            {
                const uint32_t v = (b << SURTR_LSH) + l;
                uint2 c = make_uint2(0u, 0u);
                if (v < n0) c = zfn(v);
                const uint2 e = wave_excl2(c);
                if (v < n0 && c.x)
                {
                    const uint32_t len = B.llen[v];
                    const uint32_t dst = hend + S.blk[b].y + e.y;
                    const int32_t* src = B.nbr + B.loff[v];
                    for (uint32_t q = 0; q < len; ++q) { B.nbr[dst + q] = src[q]; B.nbr[dst + 2u * len + q] = src[q]; }
                    B.loff[v] = dst; snapoff[v] = dst + 2u * len; cap[v] = 2u * len;
                    zlist[S.blk[b].x + e.x] = v;
                }
            }
            hend += zw;
            __syncthreads();
            if (tid == 0) relink_serial(B, n0, n1, snapoff, cap, zlist, zc, sh);
            __syncthreads();
            if (sh.flagErr) return 2;
            // drop the -1 marks (:426-431), one vertex per lane; then look for two-neighbour vertices
            if (tid == 0) sh.changed = 0;
            __syncthreads();
            {
                bool two = false;
                for (uint32_t v = tid; v < n1; v += SURTR_WG)
                {
                    int32_t* r = B.nbr + B.loff[v];
                    const uint32_t len = B.llen[v];
                    uint32_t wq = 0;
                    for (uint32_t q = 0; q < len; ++q) { const int32_t u = r[q]; if (u != -1) r[wq++] = u; }
                    B.llen[v] = wq;
                    if (B.comp[v] >= 0 && wq == 2u) two = true;
                }
                if (two) sh.changed = 1;
            }
            __syncthreads();
            if (sh.changed)
            {
                if (tid == 0) collapse_serial(B, n1);
                __syncthreads();
            }
        }

        __syncthreads();     // rings and lengths of this plane are final
        STAMP(6);
        // ---- compaction (:464-495) ----
        Buf& D = S.b[cur ^ 1u];
        auto livefn = [&](uint32_t v) -> uint2 {
            return (B.comp[v] >= 0) ? make_uint2(1u, B.llen[v]) : make_uint2(0u, 0u);
        };
        uint32_t nn = 0, hh = 0;
        scan_blocks(n1, S.blk, sh, livefn, nn, hh);
        uint32_t* idmap = S.aux0;
        {
            const uint32_t nb = (n1 + SURTR_LANES - 1u) >> SURTR_LSH;
            for (uint32_t b = w; b < nb; b += SURTR_NWAVE)
            {
                const uint32_t v = (b << SURTR_LSH) + l;
                uint2 c = make_uint2(0u, 0u);
                if (v < n1) c = livefn(v);
                const uint2 e = wave_excl2(c);
                if (v < n1 && c.x)
                {
                    const uint32_t id = S.blk[b].x + e.x;
                    idmap[v] = id;
                    D.pos[3 * id] = B.pos[3 * v]; D.pos[3 * id + 1] = B.pos[3 * v + 1]; D.pos[3 * id + 2] = B.pos[3 * v + 2];
                    D.loff[id] = S.blk[b].y + e.y; D.llen[id] = c.y; D.comp[id] = B.comp[v];
                }
            }
        }
        __syncthreads();
        for (uint32_t v = tid; v < n1; v += SURTR_WG)
        {
            if (B.comp[v] < 0) continue;
            const uint32_t id = idmap[v];
            const int32_t* r = B.nbr + B.loff[v];
            int32_t* d = D.nbr + D.loff[id];
            const uint32_t len = B.llen[v];
            for (uint32_t q = 0; q < len; ++q)
            {
                const int32_t u = r[q];
                if (u >= 0 && B.comp[u] < 0) { SURTR_DBG("compaction: live %u links clipped %d (k=%u)\n", v, u, k); sh.flagErr = 1; d[q] = 0; continue; }
                d[q] = u < 0 ? u : (int32_t)idmap[u];
            }
        }
        __syncthreads();
        if (sh.flagErr) return 2;
        STAMP(7);
        cur ^= 1u; n = nn; hcur = hh;
        if (n + dropAlive < 4u) { n = 0; break; }
    }
    *outN = n; *outBuf = cur; *outH = (n == 0) ? 0u : hcur;
    return 0;
}

} // namespace surtr
