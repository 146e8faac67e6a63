// wave_clip.h -- the regular planes of Poly::ClipPolyhedron (Src/Poly.cpp:265-500) for the band of a Mesh, on a layout where
// every step of a plane's dependent chain is ONE memory access ("record clipper").
//
// clip_core.h's plane loop keeps the topology of the whole band in LDS as offset / length / state arrays over slots: a walk
// step is four dependent LDS round trips, the clipped set of a plane comes from a scan over every slot, dead slots are squeezed
// out when they run out, and the band (thousands of vertices of which a plane touches a few hundred) decides how many pairs a
// CU holds.  Here:
//
//   records   a vertex is one aligned 16-byte record: seven 16-bit ring entries + a 16-bit tail (first clipping plane and ring
//             length while alive; position in the plane's clipped list while it is being clipped).  A walk step, a
//             kept-neighbour count or a back-link patch reads ONE record.  Cut points have the ring
//             [predecessor, successor, kept end] a regular relink leaves.
//   buckets   the originals of the band are sorted by the first plane that clips them (stable: ascending original index
//             inside a bucket) and stay in HBM / L2.  The originals a plane clips are a contiguous range of records, copied
//             into LDS when their plane comes ("stage"): no selection scan, and only what a plane works on is in LDS.  Whether
//             an original neighbour is kept is a comparison of its id with the bucket boundary: no lookup.  The back-link of a
//             kept original is patched where it lives (one 16-byte read, one 2-byte write, beside the position gathers).
//   nlist     the cut points that are still alive (records in LDS), in creation order.  ONE ordered scan per plane over
//             (the plane's originals, then this list) gives the clipped vertices in the reference's order, the numbers of
//             their new vertices (Src/Poly.cpp:333-357 numbers new vertices by (clipped vertex, slot)) and the next plane's
//             list (the compaction :464-495 keeps creation order); at the end the list is the output order.
//   no slots  ids are record addresses and order is carried by bucket / nlist positions, so nothing is ever renumbered,
//             squeezed or tombstoned; the records of clipped cut points go to a free list.
//
// Only REGULAR planes are handled (what clip_core.h calls the fast relink): no live vertex in the plane, no ring that lists a
// kept neighbour twice, every new vertex the successor of exactly one other, rings of at most seven entries, cells of at most
// 64 planes.  Anything else -- and any capacity limit -- returns WC_BAIL before the pair has published anything, and the
// caller hands the pair to the general clipper, which reproduces the reference's in-plane / degenerate behaviour.  Results
// are bit-identical to clip_planes(): the same order rules, the same float program (plane_dist / side_of /
// PlaneLineIntersection :746-751).  Written for a workgroup of any number of waves (blockDim.x): loops stride by the group,
// ordered scans give every wave a contiguous range.
#pragma once
#include "clip_core.h"

#ifndef SURTR_WR
#define SURTR_WR 3584u          // 16-byte units of LDS per pair: cut-point records (8 bytes each) from the bottom, stage + lists from the top (56 KiB)
#endif
#ifndef SURTR_WNL
#define SURTR_WNL 2560u         // cut points alive at any time
#endif
#ifndef SURTR_WWALK0
#define SURTR_WWALK0 4u         // walk steps before the runs of clipped vertices are collapsed by pointer jumping
#endif
#define WC_MAXF 64u             // planes per cell the record clipper takes (in-plane mask = one 64-bit word)
#define WC_MAXN 0x2000u         // band vertices (ids of originals are below, ids of cut points from here on)
#define WC_BAIL 102             // internal: not regular / does not fit -> the general clipper takes the pair
#define WC_SENT 0xFFFEu         // ring entry: a vertex the band reduction dropped (InLds::SENT)
#define WC_NONE 0xFFFFu
// why[site]: pairs handed on per rule (Arena::cursors[96 + site], surtr_queue_stats); the single-lane CPU build of the tests
// also says so on stderr
#define WC_RET(site) do { if (threadIdx.x == 0u) atomicAdd(&why[site], 1u); SURTR_DBG("record clip: pair handed on at site %d\n", site); return WC_BAIL; } while (0)
// Diagnostic build only (-DSURTR_STAMP): lane-0 cycles per phase
#ifdef SURTR_STAMP
__device__ unsigned long long g_wstamp[64];
__device__ unsigned long long g_wplane[32];      // per class of items per plane (<= 32, 64, 128, 256, 512, 1024, 2048, more): planes, lane-0 cycles
__device__ uint32_t g_wneed[4 * 8192];      // per pair: band vertices, largest bucket, LDS bytes needed at the worst plane, cycles
#define WSTAMP_DECL unsigned long long ws_t0 = __builtin_readcyclecounter(), ws_t1
// (accumulated in LDS, flushed once per pair by the kernel: a global atomic per phase would be what the stamps measure)
#define WSTAMP(i) do { if (threadIdx.x == 0u) { ws_t1 = __builtin_readcyclecounter(); W.ph[i] += ws_t1 - ws_t0; ws_t0 = ws_t1; } } while (0)
#define WCOUNT(i, v) do { if (threadIdx.x == 0u) W.ph[i] += (unsigned long long)(v); } while (0)
#else
#define WSTAMP_DECL
#define WSTAMP(i) do { } while (0)
#define WCOUNT(i, v) do { } while (0)
#endif

namespace surtr {

struct alignas(16) WcW4 { uint32_t a, b, c, d; };

// All LDS record / list accesses go through memcpy on a byte array (no type punning for the optimiser to trip over); the
// alignment hints make them single ds_read / ds_write instructions.  i16: index in 16-bit words from the start of the array.
__device__ __forceinline__ uint32_t wc_ld16(const unsigned char* B, uint32_t i16) { uint16_t v; __builtin_memcpy(&v, __builtin_assume_aligned(B + 2u * (size_t)i16, 2), 2); return v; }
__device__ __forceinline__ void wc_st16(unsigned char* B, uint32_t i16, uint32_t v) { const uint16_t x = (uint16_t)v; __builtin_memcpy(__builtin_assume_aligned(B + 2u * (size_t)i16, 2), &x, 2); }
__device__ __forceinline__ uint32_t wc_ld32(const unsigned char* B, uint32_t i16) { uint32_t v; __builtin_memcpy(&v, __builtin_assume_aligned(B + 2u * (size_t)i16, 4), 4); return v; }
__device__ __forceinline__ void wc_st32(unsigned char* B, uint32_t i16, uint32_t v) { __builtin_memcpy(__builtin_assume_aligned(B + 2u * (size_t)i16, 4), &v, 4); }
__device__ __forceinline__ uint32_t wc_ld8(const unsigned char* B, uint32_t i8) { return B[i8]; }
__device__ __forceinline__ void wc_st8(unsigned char* B, uint32_t i8, uint32_t v) { B[i8] = (unsigned char)v; }

// A record in registers: entries 0..6 in w0..w3 (two per word), tail = high half of w3.
//   alive:         tail = first clipping plane (SURTR_NEVER: none) | ring length << 8
//   being clipped: tail = 0x8000 | ring length << 12 | position in the plane's clipped list
// Unused entries hold WC_NONE.
struct WcRec
{
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ uint32_t tail() const { return w3 >> 16; }
    __device__ __forceinline__ uint32_t e(uint32_t q) const
    {
        const uint32_t w = q < 2u ? w0 : (q < 4u ? w1 : (q < 6u ? w2 : w3));
        return (q & 1u) ? (w >> 16) : (w & 0xFFFFu);
    }
    // first slot below len that holds `who` (std::find, Src/Poly.cpp:34-41); len when absent
    __device__ __forceinline__ uint32_t find(uint32_t who, uint32_t len) const
    {
        const uint32_t x = who | (who << 16);
        const uint32_t d0 = w0 ^ x, d1 = w1 ^ x, d2 = w2 ^ x, d3 = w3 ^ x;
        uint32_t m = 0;
        m |= (d0 & 0xFFFFu) ? 0u : 1u;  m |= (d0 >> 16) ? 0u : 2u;
        m |= (d1 & 0xFFFFu) ? 0u : 4u;  m |= (d1 >> 16) ? 0u : 8u;
        m |= (d2 & 0xFFFFu) ? 0u : 16u; m |= (d2 >> 16) ? 0u : 32u;
        m |= (d3 & 0xFFFFu) ? 0u : 64u;
        m &= (1u << len) - 1u;
        return m ? (uint32_t)__builtin_ctz(m) : len;
    }
};
// The record of a vertex that is in LDS, by its byte offset: an original (16 bytes, in the plane's stage) or a cut point (8 bytes
// in the pool: [pred, succ, kept end, tail]; presented with entries 3..6 = WC_NONE and the tail in its usual place).
struct alignas(8) WcW2 { uint32_t a, b; };
__device__ __forceinline__ WcRec wc_rec(const unsigned char* B, uint32_t byte_off, bool cut)
{
    WcW2 lo, hi;
    __builtin_memcpy(&lo, __builtin_assume_aligned(B + byte_off, 8), 8);
    __builtin_memcpy(&hi, __builtin_assume_aligned(B + byte_off + 8u, 8), 8);      // (a cut point: the next record's bytes, unused)
    WcRec r;
    r.w0 = lo.a;
    r.w1 = cut ? (lo.b | 0xFFFF0000u) : lo.b;
    r.w2 = cut ? 0xFFFFFFFFu : hi.a;
    r.w3 = cut ? ((lo.b & 0xFFFF0000u) | 0xFFFFu) : hi.b;
    return r;
}
// a 16-byte record at unit `unit` (park: none; loader / global copies use WcW4 directly)

template <uint32_t NR, uint32_t NL = SURTR_WNL, uint32_t NW = SURTR_NWAVE, bool GSTAGE = (NR < SURTR_WR)>
struct alignas(16) WcLdsT
{
    static constexpr uint32_t kNR = NR, kNL = NL;
    static constexpr bool kGStage = GSTAGE;      // a plane whose stage does not fit works on the records in global memory (else: WC_BAIL)      // (NW: waves of the largest group a kernel with this LDS is launched with)
    alignas(16) unsigned char U[16u * NR];    // cut-point records from the bottom; the plane's stage and lists from the top
    float4 planes[WC_MAXF];
    uint16_t nlist[2][NL];             // alive cut points (ids) in creation order; the planes alternate between the two
    uint16_t freel[NL];                // record units of cut points that are gone
    uint32_t hist[WC_MAXF + 1], zhist[WC_MAXF + 1];
    uint32_t bst[WC_MAXF + 2];                // first id of bucket k (bucket F: never clipped); bst[F + 1] = n
    uint32_t wcnt[NW][WC_MAXF + 2];           // loader: originals per (wave, bucket)
    uint32_t wsum[2][2 * NW];                 // ordered scans: totals per wave (two sets, alternating)
    uint32_t bsum[2][2u * (4096u / SURTR_LANES)];   // the item scan: totals per 64-item block (two sets, alternating)
    uint32_t fl[3][2];                        // group-wide flags, three sets in rotation (see wc_any)
    uint32_t zm[2];                           // planes some cut point lies in
    uint32_t misc[8];
#ifdef SURTR_STAMP
    unsigned long long ph[32];
#endif
};
typedef WcLdsT<SURTR_WR> WcLds;
// the same with a whole CU's LDS, for the bands that are too large for the regular one (one workgroup per CU)
#ifndef SURTR_WR_BIG
#define SURTR_WR_BIG 7424u
#define SURTR_WNL_BIG 6144u
#endif
typedef WcLdsT<SURTR_WR_BIG, SURTR_WNL_BIG> WcLdsBig;

// The reduced Mesh of a pair as k_prep_pairs left it (ImgLayout, surtr_ctx.h).
struct WcImg
{
    const uint16_t* loff; const uint8_t* llen; const uint8_t* fc; const uint16_t* ring; const float* pos;
    const uint32_t* hist; const uint32_t* zhist; const uint32_t* nzero;
    uint32_t n, hsum;
};
// Per-workgroup global scratch: the sorted records of the originals and their positions (n each), positions of the cut
// points (by record unit).
struct WcGlob { WcW4* grec; float4* gpos; float4* cpos; };
__device__ __forceinline__ WcGlob wc_glob(char* slot, size_t slot_bytes, uint32_t n, uint32_t pool, bool& fits)
{
    WcGlob g;
    const size_t nn = ((size_t)n + 15u) & ~(size_t)15u;
    g.grec = (WcW4*)slot; g.gpos = (float4*)(slot + 16u * nn); g.cpos = (float4*)(slot + 32u * nn);
    fits = 32u * nn + 16u * (size_t)pool <= slot_bytes;
    return g;
}

// What the plane loop leaves for wc_park.
struct WcOut { uint32_t nl, nLive, rtop, cur; };
// Call counters of wc_any / wc_scan of one pair (uniform over the group).
struct WcCtr { uint32_t ac, sc; };

// Group-wide "does any thread say yes": one barrier.  Calls are numbered by the counter c (uniform); set c % 3 is written
// before the barrier and read after it, set (c + 1) % 3 is cleared before the barrier -- its readers (call c - 2) are all past
// the barrier of call c - 1, its writers (call c + 1) come after this one.
template <class LT>
__device__ __forceinline__ bool wc_any(LT& W, uint32_t& c, bool pred, uint32_t slot = 0u)
{
    if (threadIdx.x == 0u) { W.fl[(c + 1u) % 3u][0] = 0u; W.fl[(c + 1u) % 3u][1] = 0u; }
    if (pred) W.fl[c % 3u][slot] = 1u;
    __syncthreads();
    const bool r = W.fl[c % 3u][slot] != 0u;
    ++c;
    return r;
}

// Ordered scan over items [0, N): every wave sweeps a contiguous range of 64-item blocks twice -- count(i) -> (a, b), then
// place(i, exclusive a, exclusive b) -- with one barrier between the sweeps.  Returns the totals.  Every thread must call it;
// sc: the pair's call counter (the per-wave totals alternate between two sets, so that no barrier is needed after it).
template <class LT, class C1, class C2, class P>
__device__ __forceinline__ uint2 wc_scan(LT& W, uint32_t& sc, uint32_t N, C1 count1, C2 count2, P place)
{
    const uint32_t lane = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t nb = (N + SURTR_LANES - 1u) >> SURTR_LSH, nbw = (nb + nw - 1u) / nw;
    const uint32_t b0 = w * nbw, b1 = b0 + nbw < nb ? b0 + nbw : nb;
    uint32_t* ws = W.wsum[sc & 1u];
    ++sc;
    uint2 mine = make_uint2(0u, 0u);
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        if (i < N) { const uint2 c = count1(i); mine.x += c.x; mine.y += c.y; }
    }
    const uint2 inc = wave_incl_scan2(mine);
    if (lane == SURTR_LANES - 1u) { ws[2u * w] = inc.x; ws[2u * w + 1u] = inc.y; }
    __syncthreads();
    uint2 run = make_uint2(0u, 0u), tot = make_uint2(0u, 0u);
    for (uint32_t q = 0; q < nw; ++q)
    {
        const uint32_t a = ws[2u * q], b = ws[2u * q + 1u];
        if (q < w) { run.x += a; run.y += b; }
        tot.x += a; tot.y += b;
    }
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        uint2 c = make_uint2(0u, 0u);
        if (i < N) c = count2(i);
        const uint2 s = wave_incl_scan2(c);
        if (i < N) place(i, run.x + s.x - c.x, run.y + s.y - c.y);
        run.x += lane_bcast(s.x, SURTR_LANES - 1u); run.y += lane_bcast(s.y, SURTR_LANES - 1u);
    }
    return make_uint2(SURTR_UNIFORM(tot.x), SURTR_UNIFORM(tot.y));
}

// The same scan in two halves, so that the caller can act on the totals (carve lists, check room) before anything is placed,
// and with the counts of the first NC blocks of every wave kept in registers between the sweeps (what count() read need not be
// read again): count(i, aux&) -> (a, b); mid() runs between the first sweep and the barrier; place(i, exclusive a, exclusive b,
// (a, b), aux).
template <int NC> struct WcScanState { uint32_t b0, b1; uint2 run; uint2 cc[NC]; uint32_t ax[NC]; };
template <int NC, class LT, class C, class M>
__device__ __forceinline__ uint2 wc_scan_count(LT& W, uint32_t& sc, uint32_t N, WcScanState<NC>& st, C count, M mid)
{
    const uint32_t lane = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t nb = (N + SURTR_LANES - 1u) >> SURTR_LSH, nbw = (nb + nw - 1u) / nw;
    st.b0 = w * nbw; st.b1 = st.b0 + nbw < nb ? st.b0 + nbw : nb;
    uint32_t* ws = W.wsum[sc & 1u];
    ++sc;
    uint2 mine = make_uint2(0u, 0u);
#pragma unroll
    for (int bi = 0; bi < NC; ++bi)
    {
        st.cc[bi] = make_uint2(0u, 0u); st.ax[bi] = 0u;
        const uint32_t i = ((st.b0 + (uint32_t)bi) << SURTR_LSH) + lane;
        if (st.b0 + (uint32_t)bi < st.b1 && i < N) { st.cc[bi] = count(i, st.ax[bi]); mine.x += st.cc[bi].x; mine.y += st.cc[bi].y; }
    }
    for (uint32_t b = st.b0 + (uint32_t)NC; b < st.b1; ++b)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        if (i < N) { uint32_t a; const uint2 c = count(i, a); mine.x += c.x; mine.y += c.y; }
    }
    const uint2 inc = wave_incl_scan2(mine);
    if (lane == SURTR_LANES - 1u) { ws[2u * w] = inc.x; ws[2u * w + 1u] = inc.y; }
    mid();
    __syncthreads();
    uint2 run = make_uint2(0u, 0u), tot = make_uint2(0u, 0u);
    for (uint32_t q = 0; q < nw; ++q)
    {
        const uint32_t a = ws[2u * q], b = ws[2u * q + 1u];
        if (q < w) { run.x += a; run.y += b; }
        tot.x += a; tot.y += b;
    }
    st.run = run;
    return make_uint2(SURTR_UNIFORM(tot.x), SURTR_UNIFORM(tot.y));
}
template <int NC, class C, class P>
__device__ __forceinline__ void wc_scan_place(uint32_t N, WcScanState<NC>& st, C count, P place)
{
    const uint32_t lane = lane_id();
    uint2 run = st.run;
#pragma unroll
    for (int bi = 0; bi < NC; ++bi)
    {
        if (st.b0 + (uint32_t)bi >= st.b1) break;
        const uint32_t i = ((st.b0 + (uint32_t)bi) << SURTR_LSH) + lane;
        const uint2 c = st.cc[bi];
        const uint2 s = wave_incl_scan2(c);
        if (i < N) place(i, run.x + s.x - c.x, run.y + s.y - c.y, c, st.ax[bi]);
        run.x += lane_bcast(s.x, SURTR_LANES - 1u); run.y += lane_bcast(s.y, SURTR_LANES - 1u);
    }
    for (uint32_t b = st.b0 + (uint32_t)NC; b < st.b1; ++b)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        uint2 c = make_uint2(0u, 0u); uint32_t a = 0u;
        if (i < N) c = count(i, a);
        const uint2 s = wave_incl_scan2(c);
        if (i < N) place(i, run.x + s.x - c.x, run.y + s.y - c.y, c, a);
        run.x += lane_bcast(s.x, SURTR_LANES - 1u); run.y += lane_bcast(s.y, SURTR_LANES - 1u);
    }
}

// The item scan of the plane loop: the same contract as wc_scan_count / wc_scan_place (counts of the first NC blocks of a wave
// stay in registers), but 64-item block b belongs to wave b % waves -- a plane's originals, whose records come from global
// memory, are the first blocks: this way every wave has at most a few of them and LDS-only blocks to work on meanwhile.
// pref(i) fetches what count(i, aux&, fetched) needs from far away; the fetches of a wave's first NC blocks are issued before
// anything is counted, and the blocks are counted last to first.  At most 4096 items (64 blocks of 64).
template <int NC> struct WcRRState { uint2 cc[NC], ss[NC], px; uint32_t ax[NC]; };
template <int NC, class LT, class PF, class C, class M>
__device__ __forceinline__ uint2 wc_rr_count(LT& W, uint32_t& sc, uint32_t N, WcRRState<NC>& st, PF pref, C count, M mid)
{
    const uint32_t lane = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t nb = (N + SURTR_LANES - 1u) >> SURTR_LSH;
    uint32_t* bs = W.bsum[sc & 1u];
    ++sc;
    WcW4 pf[NC];
#pragma unroll
    for (int bi = 0; bi < NC; ++bi)
    {
        const uint32_t b = w + (uint32_t)bi * nw, i = (b << SURTR_LSH) + lane;
        pf[bi] = WcW4{0u, 0u, 0u, 0u};
        if (b < nb && i < N) pf[bi] = pref(i);
    }
#pragma unroll
    for (int bi = NC - 1; bi >= 0; --bi)
    {
        const uint32_t b = w + (uint32_t)bi * nw, i = (b << SURTR_LSH) + lane;
        st.cc[bi] = make_uint2(0u, 0u); st.ax[bi] = 0u; st.ss[bi] = make_uint2(0u, 0u);
        if (b < nb)
        {
            if (i < N) st.cc[bi] = count(i, st.ax[bi], pf[bi]);
            st.ss[bi] = wave_incl_scan2(st.cc[bi]);
            if (lane == SURTR_LANES - 1u) { bs[2u * b] = st.ss[bi].x; bs[2u * b + 1u] = st.ss[bi].y; }
        }
    }
    for (uint32_t b = w + (uint32_t)NC * nw; b < nb; b += nw)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        uint2 c = make_uint2(0u, 0u);
        if (i < N) { uint32_t a; c = count(i, a, pref(i)); }
        const uint2 s2 = wave_incl_scan2(c);
        if (lane == SURTR_LANES - 1u) { bs[2u * b] = s2.x; bs[2u * b + 1u] = s2.y; }
    }
    mid();
    __syncthreads();
    // exclusive prefix of the block totals: lane l of every wave holds block (chunk + l)'s; a wave keeps those of its own blocks
    uint2 run = make_uint2(0u, 0u);
    st.px = make_uint2(0u, 0u);
    for (uint32_t c0 = 0; c0 < nb; c0 += SURTR_LANES)
    {
        const uint32_t b = c0 + lane;
        const uint2 v = b < nb ? make_uint2(bs[2u * b], bs[2u * b + 1u]) : make_uint2(0u, 0u);
        const uint2 inc = wave_incl_scan2(v);
        // (one chunk on the device: at most 64 blocks, kept in px; the single-lane, single-thread build of the tests has one
        //  block per chunk and turns the totals into prefixes where they are)
        if (SURTR_LANES > 1u) st.px = make_uint2(run.x + inc.x - v.x, run.y + inc.y - v.y);
        else if (b < nb) { bs[2u * b] = run.x + inc.x - v.x; bs[2u * b + 1u] = run.y + inc.y - v.y; }      // (one thread: in place)
        run.x += lane_bcast(inc.x, SURTR_LANES - 1u); run.y += lane_bcast(inc.y, SURTR_LANES - 1u);
    }
    return make_uint2(SURTR_UNIFORM(run.x), SURTR_UNIFORM(run.y));
}
// exclusive prefix of block b (uniform)
template <int NC, class LT>
__device__ __forceinline__ uint2 wc_rr_base(LT& W, const uint32_t* bs, const WcRRState<NC>& st, uint32_t b)
{
    if (SURTR_LANES > 1u) return make_uint2(lane_bcast(st.px.x, b), lane_bcast(st.px.y, b));
    return make_uint2(bs[2u * b], bs[2u * b + 1u]);
}
template <int NC, class LT, class C, class P>
__device__ __forceinline__ void wc_rr_place(LT& W, uint32_t sc_after, uint32_t N, WcRRState<NC>& st, C count, P place)
{
    const uint32_t lane = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t nb = (N + SURTR_LANES - 1u) >> SURTR_LSH;
    const uint32_t* bs = W.bsum[(sc_after - 1u) & 1u];
#pragma unroll
    for (int bi = 0; bi < NC; ++bi)
    {
        const uint32_t b = w + (uint32_t)bi * nw, i = (b << SURTR_LSH) + lane;
        if (b >= nb) break;
        const uint2 base = wc_rr_base<NC>(W, bs, st, b);
        const uint2 c = st.cc[bi], s2 = st.ss[bi];
        if (i < N) place(i, base.x + s2.x - c.x, base.y + s2.y - c.y, c, st.ax[bi]);
    }
    for (uint32_t b = w + (uint32_t)NC * nw; b < nb; b += nw)
    {
        const uint32_t i = (b << SURTR_LSH) + lane;
        const uint2 base = wc_rr_base<NC>(W, bs, st, b);
        uint2 c = make_uint2(0u, 0u); uint32_t a = 0u;
        if (i < N) c = count(i, a);
        const uint2 s2 = wave_incl_scan2(c);
        if (i < N) place(i, base.x + s2.x - c.x, base.y + s2.y - c.y, c, a);
    }
}

__device__ __forceinline__ uint32_t wc_popc64(unsigned long long m) { return (uint32_t)__builtin_popcountll(m); }

// Sorts the band: stable counting sort by first clipping plane; records and positions of the originals go to the workgroup's
// global scratch (g.grec / g.gpos, indexed by the sorted id).  The caller has put the cell's planes into W.planes.
// Returns 0 or WC_BAIL (uniform); zmask = planes some band vertex lies in.
template <class LT>
__device__ __attribute__((always_inline)) inline int wc_load(LT& W, const WcImg im, const uint32_t F, const WcGlob g,
                                                             unsigned long long& zmask, WcCtr& ctr, uint32_t* __restrict__ why)
{
    const uint32_t tid = threadIdx.x, lane = lane_id(), w = wave_id(), nw = group_waves();
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned char* B = W.U;
    const uint32_t n = im.n;
    ctr.ac = 0u; ctr.sc = 0u;
    // the id map (16 bits per band vertex) sits in LDS while the records are written
    if (F > WC_MAXF || n == 0u || n >= WC_MAXN || n > 8u * LT::kNR) WC_RET(1);
    const uint32_t tmp16 = 0u;
    for (uint32_t k = tid; k < nw * (WC_MAXF + 2u); k += group_size()) (&W.wcnt[0][0])[k] = 0u;
    for (uint32_t k = tid; k < F; k += group_size()) { W.hist[k] = im.hist[k]; W.zhist[k] = im.zhist[k]; }
    if (tid == 0u) { W.zm[0] = 0u; W.zm[1] = 0u; for (int q = 0; q < 6; ++q) (&W.fl[0][0])[q] = 0u; }
    unsigned long long zm = 0ull;
    for (uint32_t k0 = 0; k0 < F; k0 += SURTR_LANES)
    {
        const uint32_t k = k0 + lane;
        zm |= __ballot(k < F && im.nzero[k] != 0u) << k0;
    }
    zmask = zm;
    // (conservative: a plane an original lies in might never be reached; the general clipper decides)
    if (zm != 0ull) WC_RET(17);
    __syncthreads();
    WSTAMP_DECL;
    // every wave takes a contiguous range of the band (in 64-vertex blocks): vertices per bucket, then ranks
    const uint32_t nb = (n + SURTR_LANES - 1u) >> SURTR_LSH, nbw = (nb + nw - 1u) / nw;
    const uint32_t bb0 = w * nbw, bb1 = bb0 + nbw < nb ? bb0 + nbw : nb;
    bool odd = false;
    for (int pass = 0; pass < 2; ++pass)
    {
        for (uint32_t bb = bb0; bb < bb1; ++bb)
        {
            const uint32_t b = (bb << SURTR_LSH) + lane;
            const bool valid = b < n;
            const uint32_t f = valid ? (uint32_t)im.fc[b] : 0u;
            uint32_t bk = f == SURTR_NEVER ? F : f;
            if (valid && pass == 0 && (bk > F || (uint32_t)im.llen[b] > 7u)) odd = true;
            if (bk > F) bk = F;
            unsigned long long todo = __ballot(valid);
            uint32_t sid = 0;
            while (todo)
            {
                const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
                const uint32_t k0 = lane_bcast(bk, leader);
                const unsigned long long same = __ballot(valid && bk == k0);
                uint32_t base = 0;
                if (lane == leader) { base = W.wcnt[w][k0]; W.wcnt[w][k0] = base + wc_popc64(same); }      // (this wave's own row)
                base = lane_bcast(base, leader);
                if (valid && bk == k0) sid = base + wc_popc64(same & lt);
                todo &= ~same;
            }
            if (valid && pass == 1) wc_st16(B, tmp16 + b, sid);
        }
        if (pass == 1) break;
        __syncthreads();
        // bucket starts, and in every wave's row the first id it hands out per bucket
        if (w == 0u)
        {
            uint32_t carry = 0;
            for (uint32_t k0 = 0; k0 <= F; k0 += SURTR_LANES)
            {
                const uint32_t k = k0 + lane;
                uint32_t c = 0;
                if (k <= F) for (uint32_t q = 0; q < nw; ++q) c += W.wcnt[q][k];
                const uint32_t inc = wave_incl_scan2(make_uint2(c, 0u)).x;
                if (k <= F)
                {
                    uint32_t at = carry + inc - c;
                    W.bst[k] = at;
                    for (uint32_t q = 0; q < nw; ++q) { const uint32_t cq = W.wcnt[q][k]; W.wcnt[q][k] = at; at += cq; }
                }
                carry += lane_bcast(inc, SURTR_LANES - 1u);
            }
            if (lane == 0u) W.bst[F + 1u] = carry;
        }
        __syncthreads();
    }
    if (wc_any(W, ctr.ac, odd)) WC_RET(2);
    WSTAMP(0);
    for (uint32_t b = tid; b < n; b += group_size())
    {
        const uint32_t sid = wc_ld16(B, tmp16 + b);
        const uint32_t lo = im.loff[b], len = im.llen[b], f = im.fc[b];
        uint32_t e[7];
#pragma unroll
        for (uint32_t q = 0; q < 7u; ++q) e[q] = im.ring[lo + (q < len ? q : 0u)];
#pragma unroll
        for (uint32_t q = 0; q < 7u; ++q)
        {
            const uint32_t m = wc_ld16(B, tmp16 + (e[q] < n ? e[q] : 0u));
            e[q] = q >= len ? WC_NONE : (e[q] < n ? m : WC_SENT);
        }
        WcW4 wr;
        wr.a = e[0] | (e[1] << 16); wr.b = e[2] | (e[3] << 16); wr.c = e[4] | (e[5] << 16); wr.d = e[6] | ((f | (len << 8)) << 16);
        g.grec[sid] = wr;
        g.gpos[sid] = make_float4(im.pos[3u * b], im.pos[3u * b + 1u], im.pos[3u * b + 2u], 0.f);
    }
    __syncthreads();
    WSTAMP(1);
    WCOUNT(16, 1); WCOUNT(17, n);
    return 0;
}

// The plane loop.  Returns 0 (out.nLive == 0: nothing is left) or WC_BAIL.
template <class LT>
__device__ __attribute__((always_inline)) inline int wc_planes(LT& W, const uint32_t F, const uint32_t n, const uint32_t dropTotal, unsigned long long zmask,
                                                               const WcGlob g, const uint32_t capPool, WcOut& out, WcCtr& ctr, uint32_t* __restrict__ why, const uint32_t walk0 = SURTR_WWALK0)
{
    const uint32_t tid = threadIdx.x, G = group_size();
    unsigned char* B = W.U;
    unsigned char* GB = (unsigned char*)g.grec;               // (16-bit patches of original records)
    uint32_t rtop = 0, nl = 0, nLive = n, cur = 0, nfree = 0, fhead = 0;  // cut-point records in use from the bottom; free list (a ring in W.freel): length, head
    uint32_t ac = ctr.ac, sc = ctr.sc;
    out.nl = 0; out.nLive = 0; out.rtop = 0; out.cur = 0;
    WSTAMP_DECL;
    for (uint32_t k = 0; k < F; ++k)
    {
#ifdef SURTR_STAMP
        const unsigned long long pl_t0 = __builtin_readcyclecounter();
#endif
        if ((zmask >> k) & 1ull) WC_RET(3);          // a live vertex lies in this plane: the reference's general relink
        const float4 pl = W.planes[k];
        // originals this plane clips: ids [b0, b1), in order; later buckets are kept, earlier ones gone
        const uint32_t b0 = SURTR_UNIFORM(W.bst[k]), b1 = SURTR_UNIFORM(W.bst[k + 1u]);
        const uint32_t nCo = b1 - b0;
        uint32_t ltop = 8u * LT::kNR;                     // stage and lists are carved downwards from the top (16-bit word index, 16-byte steps)
        auto carve = [&](uint32_t cnt16) -> uint32_t { ltop -= (cnt16 + 7u) & ~7u; return ltop; };
        // ---- the plane's ITEMS: its originals [0, nCo) -- all clipped --, then the alive cut points in creation order (clipped
        //      when their first clipping plane is this one).  One ordered scan over the items gives the clipped list in the
        //      reference's order (a clipped vertex is known by its item index from here on: tails, kept masks, first new vertex),
        //      the number of the first new vertex of every clipped vertex (Src/Poly.cpp:333-357 numbers them by (clipped vertex,
        //      slot)), and the place of every kept cut point in the next plane's list. ----
        const uint32_t NI = nCo + nl;
        if (NI > 4095u) WC_RET(6);
        // The originals a plane clips are normally copied into LDS (the "stage": 16 bytes each; the walks step through them).  The
        // first planes of a band clip hundreds of them at once -- the stage of that one plane is what decides how much LDS a pair
        // needs (scripts/wave_need.py: 42 bytes per vertex of the largest bucket).  When stage + lists would not fit, the plane
        // works on the records where they are, in global memory (this XCD's L2: the item scan has just fetched them): its walk
        // steps through originals pay an L2 round trip instead of an LDS one, and the pair keeps its place in a kernel with a
        // smaller LDS area instead of being handed on.
        // (only the kernels without the general clipper inside do this: there the alternative is to hand the pair on)
        const bool gstage = LT::kGStage && nCo != 0u && 4u * rtop + 8u * nCo + NI + (NI + 1u) / 2u + 7u * NI + 3u * nl + 128u > ltop;
        if (4u * rtop + (gstage ? 0u : 8u * nCo) + NI + (NI + 1u) / 2u + 64u > ltop) WC_RET(4);
        WCOUNT(26, gstage ? 1 : 0);
        // stage: this plane's originals in LDS (their records are final: every patch of an earlier plane is in)
        const uint32_t stage = gstage ? 0u : carve(8u * nCo) / 8u;      // first unit of the stage
        auto stage_put = [&](uint32_t i, const WcW4& wr) { if (!gstage) __builtin_memcpy(__builtin_assume_aligned(B + 16u * (size_t)(stage + i), 16), &wr, 16); };
        // the record of a vertex the plane works on: a cut point (pool, 8-byte units from the bottom of the LDS area) or an original of
        // this plane (stage, or global memory); its tail (16 bits) is read and written on its own
        auto rec_of = [&](uint32_t e) -> WcRec {
            if (e >= WC_MAXN) return wc_rec(B, 8u * (e - WC_MAXN), true);
            if (gstage) { const WcW4 wr = g.grec[e]; return WcRec{wr.a, wr.b, wr.c, wr.d}; }
            return wc_rec(B, 16u * (stage + (e - b0)), false);
        };
        auto tail_ld = [&](uint32_t e) -> uint32_t {
            if (e >= WC_MAXN) return wc_ld16(B, 4u * (e - WC_MAXN) + 3u);
            if (gstage) { uint16_t t; __builtin_memcpy(&t, GB + 16u * (size_t)e + 14u, 2); return t; }
            return wc_ld16(B, 8u * (stage + (e - b0)) + 7u);
        };
        auto tail_st = [&](uint32_t e, uint32_t v) {
            if (e >= WC_MAXN) wc_st16(B, 4u * (e - WC_MAXN) + 3u, v);
            else if (gstage) { const uint16_t t = (uint16_t)v; __builtin_memcpy(GB + 16u * (size_t)e + 14u, &t, 2); }
            else wc_st16(B, 8u * (stage + (e - b0)) + 7u, v);
        };
        const uint16_t* nin = W.nlist[cur]; uint16_t* nout = W.nlist[cur ^ 1u];
        auto clipped_id = [&](uint32_t i) -> uint32_t { return i < nCo ? b0 + i : (uint32_t)nin[i - nCo]; };
        const uint32_t cbase = carve(NI), ckm8 = 2u * carve((NI + 1u) / 2u);      // ckm8: byte index; 0x80 | kept neighbours (bit j = ring slot j) of a clipped item, 0 for a kept one
        bool bad = false;
        WcRRState<2> st2;
        // what an item needs from global memory: the record of an original
        auto pfn = [&](uint32_t i) -> WcW4 { return i < nCo ? g.grec[b0 + i] : WcW4{0u, 0u, 0u, 0u}; };
        // count: (clipped, new vertices); aux = kept mask | 0x80 | id << 8 | ring length << 24 (a kept cut point: id << 8)
        auto kfn = [&](uint32_t i, uint32_t& aux, const WcW4 wr) -> uint2 {
            uint32_t id; WcRec r;
            if (i < nCo)
            {
                stage_put(i, wr);
                id = b0 + i; r = WcRec{wr.a, wr.b, wr.c, wr.d};
            }
            else
            {
                id = nin[i - nCo];
                r = wc_rec(B, 8u * (id - WC_MAXN), true);
                if ((r.tail() & 0xFFu) != k) { wc_st8(B, ckm8 + i, 0u); aux = id << 8; return make_uint2(0u, 0u); }
            }
            // the first clipping plane of every cut point among the neighbours: all seven loads in flight together
            uint32_t ee[7], tl[7];
#pragma unroll
            for (uint32_t q = 0; q < 7u; ++q) ee[q] = r.e(q);
#pragma unroll
            for (uint32_t q = 0; q < 7u; ++q) tl[q] = wc_ld16(B, ((ee[q] >= WC_MAXN ? 1u : 0u) & (ee[q] < WC_SENT ? 1u : 0u)) ? 4u * (ee[q] - WC_MAXN) + 3u : 3u);
            uint32_t km = 0;
#pragma unroll
            for (uint32_t q = 0; q < 7u; ++q)
            {
                // an original is kept when it sits in a later bucket, a cut point when its own first clipping plane is later;
                // a dropped vertex (WC_SENT) goes with this plane
                // (bitwise, not short-circuit: the compiler turns && / ?: on lane values into exec-mask branches, seven times over)
                const uint32_t isO = ee[q] < WC_MAXN ? 1u : 0u;
                const uint32_t kept = (isO & (ee[q] >= b1 ? 1u : 0u)) | ((isO ^ 1u) & (ee[q] < WC_SENT ? 1u : 0u) & ((tl[q] & 0xFFu) > k ? 1u : 0u));
                km |= kept << q;
            }
            // A ring that lists the same kept neighbour twice makes the back-link patch order dependent (:350-354).  The rings of
            // the originals list no vertex twice (the piece has none: Pieces::mdup, checked by the caller; a patch puts the cut
            // point of ONE edge into ONE entry), so only a cut point's ring can: [Z, Z, kept] of a cap of two vertices, or
            // [pred, succ, kept] with kept among them.
            {
                const uint32_t d01 = ((km & 3u) == 3u ? 1u : 0u) & (ee[0] == ee[1] ? 1u : 0u), d02 = ((km & 5u) == 5u ? 1u : 0u) & (ee[0] == ee[2] ? 1u : 0u),
                               d12 = ((km & 6u) == 6u ? 1u : 0u) & (ee[1] == ee[2] ? 1u : 0u);
                bad = bad | (((id >= WC_MAXN ? 1u : 0u) & (d01 | d02 | d12)) != 0u);
            }
            wc_st8(B, ckm8 + i, km | 0x80u);
            aux = km | 0x80u | (id << 8) | (((r.tail() >> 8) & 7u) << 24);
            return make_uint2(1u, (uint32_t)__builtin_popcount(km));
        };
        const uint2 nt = wc_rr_count<2>(W, sc, NI, st2, pfn, kfn, [&]() { WSTAMP(14); });
        WSTAMP(2);
        const uint32_t nC = nt.x, M = nt.y, nCn = nC - nCo, keepn = nl - nCn;
        const uint32_t dropAlive = SURTR_UNIFORM(W.hist[k]);
        const uint32_t dropKept = dropAlive - SURTR_UNIFORM(W.zhist[k]);      // dropped vertices strictly on the kept side
        if (nLive <= nC && dropKept == 0u)
        {
            if (nC == 0u) WC_RET(5);                 // every vertex in the plane: the bounding-box rule (:296-299) of the general clipper
            nLive = 0; break;                             // "below": everything goes (:322-327)
        }
        if (nC == 0u)
        {
            // (nothing is clipped: the list of alive cut points stays where it is)
            if (nLive + dropAlive < 4u) { nLive = 0; break; }      // (:497-499)
            continue;
        }
        if (M > 4095u || keepn + M > LT::kNL || nfree + nCn > LT::kNL) WC_RET(9);
        const uint32_t bmw = (M + 31u) / 32u;
        if (4u * rtop + 3u * M + 2u * bmw + 2u * nl + nCn + 72u > ltop) { if (tid == 0u && atomicAdd(&why[20], 1u) == 3u) { why[21] = n; why[22] = k; why[23] = rtop; why[24] = M; why[25] = nC; why[26] = nl; why[27] = ltop; why[28] = nfree; } WC_RET(10); }
        const uint32_t src = carve(M), srcid = carve(M), wst = carve(M), nd = carve(2u * nl), bm = carve(2u * bmw), clist = carve(nCn);      // clist: item indices of the clipped cut points
        // ---- record units of the new vertices: those of cut points that are gone first (the free list is a ring: taken at its
        //      head, the units of this plane's clipped cut points are put behind its end while the list is placed), then the end
        //      of the pool ----
        const uint32_t fromf = nfree < M ? nfree : M, fromt = M - fromf, fh0 = fhead, fpush = fhead + nfree, tbase = rtop;
        {
            const uint32_t room = ltop / 4u > rtop ? ltop / 4u - rtop : 0u;
            if (fromt > room || rtop + fromt > capPool) { if (tid == 0u && atomicAdd(&why[20], 1u) == 3u) { why[21] = n; why[22] = k; why[23] = rtop; why[24] = M; why[25] = nC; why[26] = nl; why[27] = ltop; why[28] = nfree; } WC_RET(16); }
            rtop += fromt;
            fhead += fromf; if (fhead >= LT::kNL) fhead -= LT::kNL;
            nfree = nfree - fromf + nCn;
        }
        auto fring = [&](uint32_t at) -> uint32_t { return at >= LT::kNL ? at - LT::kNL : at; };      // (at < 2 * kNL)
        auto xof = [&](uint32_t t) -> uint32_t { return WC_MAXN + (t < fromf ? (uint32_t)W.freel[fring(fh0 + t)] : tbase + (t - fromf)); };
        for (uint32_t q = tid; q < bmw; q += G) wc_st32(B, bm + 2u * q, 0u);
        WSTAMP(3);
        WCOUNT(18, 1); WCOUNT(19, nC); WCOUNT(20, M); WCOUNT(21, nl);
#ifdef SURTR_STAMP
        if (tid == 0u) { const unsigned long long need = 2ull * (4u * rtop + (8u * LT::kNR - ltop)); if (need > W.ph[31]) W.ph[31] = need; if (k >= 2u && need > W.ph[29]) W.ph[29] = need; }
#endif
        // ---- every clipped vertex gets its item index into its tail; source (clipped vertex, slot) of every new vertex, in the
        //      reference's order (:333-357); the kept cut points move to the other list, in order ----
        // (blocks beyond those held in registers: the mask is read back, not derived again -- tails are being rewritten by now)
        auto kfn2 = [&](uint32_t i, uint32_t& aux) -> uint2 {
            const uint32_t id = clipped_id(i), km = wc_ld8(B, ckm8 + i);
            if (!(km & 0x80u)) { aux = id << 8; return make_uint2(0u, 0u); }
            aux = km | (id << 8) | (((tail_ld(id) >> 8) & 7u) << 24);
            return make_uint2(1u, (uint32_t)__builtin_popcount(km & 0x7Fu));
        };
        wc_rr_place<2>(W, sc, NI, st2, kfn2, [&](uint32_t i, uint32_t xc, uint32_t xm, uint2 c, uint32_t aux) {
            const uint32_t km = aux & 0x7Fu, id = (aux >> 8) & 0xFFFFu, len = aux >> 24;
            if (!c.x) { nout[i - xc] = (uint16_t)id; return; }      // (a kept cut point: every original before it is clipped)
            wc_st16(B, cbase + i, xm);
            tail_st(id, 0x8000u | (len << 12) | i);
            if (i >= nCo) { W.freel[fring(fpush + (xc - nCo))] = (uint16_t)(id - WC_MAXN); wc_st16(B, clist + (xc - nCo), i); }
            uint32_t t = xm;
            for (uint32_t m = km; m; m &= m - 1u, ++t) { wc_st16(B, src + t, i | ((uint32_t)__builtin_ctz(m) << 12)); wc_st16(B, srcid + t, id); }
        });
        cur ^= 1u;
        WSTAMP(15);
        __syncthreads();
        WSTAMP(4);
        // ---- successor of every new vertex X on the edge (v, slot j): FaceLoop from X through v takes the entry before slot j,
        //      and so on through clipped vertices, until the entry is a kept vertex -- the new vertex on that edge (:367-425) ----
        const uint32_t nref = nLive + (k ? SURTR_UNIFORM(W.hist[k - 1u]) : dropTotal) + M;      // the reference's walk bound (:389-394)
        bool fail = false;
        // state of a walk: at clipped vertex cv (position pcv in the clipped list, record rc) about to take ring slot p.
        // Returns the number of the new vertex it ends on, WC_NONE when it pauses after `limit` steps (or fails).
        auto walk = [&](uint32_t& cv, uint32_t& pcv, uint32_t& p, WcRec& rc, uint32_t& steps, const uint32_t limit, const bool jump) -> uint32_t {
            while (true)
            {
                if (jump && p == 0u && pcv >= nCo)
                {
                    // a run of clipped vertices each entered through its slot 1 and left through its slot 0: the pointer
                    // jumping below collapsed it (clip_core.h does the same for its resumed walks)
                    const uint32_t a = wc_ld32(B, nd + 2u * (pcv - nCo)), jx = a & 0xFFFFu;
                    if (jx != pcv) { pcv = jx; cv = clipped_id(jx); steps += a >> 16; rc = rec_of(cv); }
                }
                const uint32_t kmc = wc_ld8(B, ckm8 + pcv);
                if ((kmc >> p) & 1u) return wc_ld16(B, cbase + pcv) + (uint32_t)__builtin_popcount(kmc & ((1u << p) - 1u));
                if (steps >= limit) return WC_NONE;
                const uint32_t e = rc.e(p);
                // the next vertex must be one this plane clips: an original of its bucket or a cut point (checked by its tail)
                if (((e >= WC_SENT ? 1u : 0u) | ((e < WC_MAXN ? 1u : 0u) & ((e < b0 ? 1u : 0u) | (e >= b1 ? 1u : 0u))) | (steps + 2u >= nref ? 1u : 0u)) != 0u)
                { SURTR_DBG("  wc walk: sentinel / not clipped / bound e=%u steps=%u nref=%u k=%u\n", e, steps, nref, k); fail = true; return WC_NONE; }
                ++steps;
                const WcRec re = rec_of(e);
                const uint32_t te = re.tail();
                if (!(te & 0x8000u)) { SURTR_DBG("  wc walk: vertex %u not clipped now (tail %x) k=%u\n", e, te, k); fail = true; return WC_NONE; }
                const uint32_t le = (te >> 12) & 7u, q = re.find(cv, le);
                if (q >= le) { SURTR_DBG("  wc walk: no way back k=%u\n", k); fail = true; return WC_NONE; }
                p = q ? q - 1u : le - 1u; cv = e; pcv = te & 0xFFFu; rc = re;
            }
        };
        // every new vertex must be the successor of exactly one other: all M walks end on one of the M new vertices, so it is
        // enough that no two end on the same one (a bit per new vertex).  (A cap of two vertices gives rings [Z, Z, kept], as in
        // the reference; a later plane that meets such a ring at a clipped vertex finds it in the doubled-neighbour test above.)
        // The walk of new vertex t that ends on new vertex `end` links the two: X is the predecessor of Z, Z the successor of X.
        auto arrive = [&](uint32_t t, uint32_t end) {
            wc_st16(B, wst + t, 0xFFFFu);
            uint32_t* word = (uint32_t*)(void*)(B + 2u * (size_t)(bm + 2u * (end >> 5)));
            const uint32_t before = atomicOr(word, 1u << (end & 31u));
            if (((end == t ? 1u : 0u) | ((before >> (end & 31u)) & 1u)) != 0u) { SURTR_DBG("  wc walk: ends on itself / second arrival t=%u end=%u k=%u\n", t, end, k); fail = true; }
            const uint32_t X = xof(t), Z = xof(end);
            wc_st16(B, 4u * (Z - WC_MAXN), X);
            wc_st16(B, 4u * (X - WC_MAXN) + 1u, Z);
        };
        // ---- the new vertices: first steps of the walk, position, first clipping plane, record [pred, succ, kept end], back-link
        //      of the kept end -- one pass: the positions of the two ends and the record of the kept end come from global memory /
        //      L2 and are in flight while the walk takes its steps in LDS.  (Nothing but the two links depends on where the walk
        //      ends; nothing here reads what another thread writes here: the records of new vertices and of kept vertices are
        //      not on any walk's way.) ----
        bool paused = false;
        uint32_t myz0 = 0, myz1 = 0;
        uint16_t* nnow = W.nlist[cur];
        for (uint32_t t = tid; t < M; t += G)
        {
            const uint32_t s = wc_ld16(B, src + t);
            const uint32_t v = wc_ld16(B, srcid + t), j = s >> 12;
            uint32_t pcv = s & 0xFFFu, cv = v;
            WcRec rc = rec_of(cv);
            const uint32_t u = rc.e(j);
            // (one load each through a selected address: gpos and cpos are parts of the same scratch slot)
            const float4 pa = *(v < WC_MAXN ? g.gpos + v : g.cpos + (v - WC_MAXN));
            const float4 pb = *(u < WC_MAXN ? g.gpos + u : g.cpos + (u - WC_MAXN));
            // the kept end: an original lives in global memory, a cut point in LDS
            WcW4 wru = WcW4{0u, 0u, 0u, 0u};
            if (u < WC_MAXN) wru = g.grec[u];
            const uint32_t lv = (rc.tail() >> 12) & 7u;
            uint32_t p = j ? j - 1u : lv - 1u, steps = 0;
            const uint32_t end = walk(cv, pcv, p, rc, steps, walk0, false);
            const uint32_t X = xof(t), ux = X - WC_MAXN;
            // PlaneLineIntersection (:746-751): (a*sb - b*sa) * (1/(sb-sa))
            const float sa = plane_dist(pl, pa.x, pa.y, pa.z), sb = plane_dist(pl, pb.x, pb.y, pb.z);
            const float inv = 1.f / (sb - sa);
            const float nx = (pa.x * sb - pb.x * sa) * inv, ny = (pa.y * sb - pb.y * sa) * inv, nz = (pa.z * sb - pb.z * sa) * inv;
            g.cpos[ux] = make_float4(nx, ny, nz, 0.f);
            uint32_t f = SURTR_NEVER;
            for (uint32_t q = k + 1u; q < F; ++q)
            {
                const int cq = side_of(plane_dist(W.planes[q], nx, ny, nz));
                if (cq < 0) { f = q; break; }
                if (cq == 0) { if (q < 32u) myz0 |= 1u << q; else myz1 |= 1u << (q - 32u); }
            }
            // the record but for its slots 0 and 1, which the walks that end on X and start from X write
            wc_st32(B, 4u * ux + 2u, u | ((f | (3u << 8)) << 16));
            nnow[keepn + t] = (uint16_t)X;
            // the kept end now links X instead of v (:350-354: first occurrence)
            const WcRec ru = u < WC_MAXN ? WcRec{wru.a, wru.b, wru.c, wru.d} : wc_rec(B, 8u * (u - WC_MAXN), true);
            const uint32_t tu = ru.tail(), qu = ru.find(v, (tu >> 8) & 7u);
            if ((tu & 0x8000u) || qu >= ((tu >> 8) & 7u)) { SURTR_DBG("  wc patch: kept end does not link the clipped vertex k=%u\n", k); fail = true; }
            else if (u < WC_MAXN) { const uint16_t xv = (uint16_t)X; __builtin_memcpy(GB + 16u * (size_t)u + 2u * qu, &xv, 2); }
            else wc_st16(B, 4u * (u - WC_MAXN) + qu, X);
            if (end != WC_NONE) arrive(t, end);
            else { wc_st16(B, wst + t, pcv | (p << 12)); paused = true; }
        }
        const bool anyPaused = wc_any(W, ac, paused, 0u);
        WSTAMP(5);
        if (anyPaused)
        {
            WCOUNT(24, 1);
            // nd[i] = (where the walk that stands on clipped vertex i, about to take its slot 0, gets to along its run | steps):
            // it moves on to e0 whenever slot 0 holds no kept vertex, and is there about to take slot 0 again when it arrives
            // through e0's slot 1
            // (only the cut points get an entry: the runs are runs of cap vertices; a walk that stands on an original steps on)
            for (uint32_t ci = tid; ci < nCn; ci += G)
            {
                const uint32_t i = wc_ld16(B, clist + ci);
                const uint32_t c = clipped_id(i);
                const WcRec r = rec_of(c);
                const uint32_t e0 = r.e(0u);
                uint32_t nx = i, d = 0;
                if (!(wc_ld8(B, ckm8 + i) & 1u) && e0 >= WC_MAXN && e0 < WC_SENT)
                {
                    const WcRec r0 = rec_of(e0);
                    const uint32_t t0 = r0.tail();
                    if ((t0 & 0x8000u) && r0.find(c, (t0 >> 12) & 7u) == 1u) { nx = t0 & 0xFFFu; d = 1u; }
                }
                wc_st32(B, nd + 2u * (i - nCo), nx | (d << 16));
            }
            __syncthreads();
            for (uint32_t round = 0; round < 12u; ++round)
            {
                bool ch = false;
                for (uint32_t ci = tid; ci < nCn; ci += G)
                {
                    const uint32_t i = wc_ld16(B, clist + ci);
                    // (two levels per round: whatever value a lane reads is a vertex further down the same run, with its distance)
                    const uint32_t a = wc_ld32(B, nd + 2u * (i - nCo)), j1 = a & 0xFFFFu;
                    if (j1 == i) continue;
                    const uint32_t b2 = wc_ld32(B, nd + 2u * (j1 - nCo)), j2 = b2 & 0xFFFFu;
                    if (j2 == j1) continue;
                    const uint32_t b3 = wc_ld32(B, nd + 2u * (j2 - nCo)), j3 = b3 & 0xFFFFu;
                    wc_st32(B, nd + 2u * (i - nCo), j3 | (((a >> 16) + (b2 >> 16) + (j3 != j2 ? (b3 >> 16) : 0u)) << 16));
                    ch = true;
                }
                if (!wc_any(W, ac, ch, 0u)) break;
            }
            WSTAMP(6);
            for (uint32_t t = tid; t < M; t += G)
            {
                const uint32_t ws = wc_ld16(B, wst + t);
                if (ws == 0xFFFFu) continue;
                uint32_t pcv = ws & 0xFFFu, cv = clipped_id(pcv), p = ws >> 12, steps = walk0;
                WcRec rc = rec_of(cv);
                const uint32_t end = walk(cv, pcv, p, rc, steps, 0xFFFFFFFFu, true);
                if (end != WC_NONE) arrive(t, end);
                else fail = true;
#ifdef SURTR_STAMP
                atomicAdd(&g_wstamp[22], (unsigned long long)steps); atomicMax(&g_wstamp[23], (unsigned long long)steps);      // (rare: resumed walks only)
#endif
            }
            WSTAMP(7);
        }
        if (myz0) atomicOr(&W.zm[0], myz0);
        if (myz1) atomicOr(&W.zm[1], myz1);
        if (wc_any(W, ac, fail || bad, 1u)) WC_RET(11);
        WSTAMP(8);
        zmask |= (unsigned long long)SURTR_UNIFORM(W.zm[0]) | ((unsigned long long)SURTR_UNIFORM(W.zm[1]) << 32);
        // (the units of the cut points this plane clipped went to the free list with the scan; nobody takes from it before the
        // next plane's scan is through its barrier)
        nLive = nLive - nC + M; nl = keepn + M;
        WSTAMP(9);
#ifdef SURTR_STAMP
        if (tid == 0u) { int c = 0; while (c < 7 && (32u << c) < NI) ++c; atomicAdd(&g_wplane[2 * c], 1ull); atomicAdd(&g_wplane[2 * c + 1], __builtin_readcyclecounter() - pl_t0); }
#endif
        if (nLive + dropAlive < 4u) { nLive = 0; break; }                          // (:497-499)
    }
    __syncthreads();
    out.nl = nl; out.nLive = nLive; out.rtop = rtop; out.cur = cur;
    ctr.ac = ac; ctr.sc = sc;
    return 0;
}

// Islands of what is left (CheckMeshIsland, Src/Surtr.cpp:2157-2201) + the island-major copy to the arena (:1474-1500):
// what park_mesh_islands does for a Topo.  Returns 0, SURTR_E_CAPACITY or WC_BAIL.
template <class LT, class AR, class PR>
__device__ __attribute__((always_inline)) inline int wc_park(LT& W, const uint32_t F, const uint32_t n, const WcOut o, const WcGlob g,
                                                             const AR& A, PR& rec, WcCtr& ctr, uint32_t* __restrict__ why)
{
    const uint32_t tid = threadIdx.x, G = group_size(), lane = lane_id();
    unsigned char* B = W.U;
    const uint32_t bN = SURTR_UNIFORM(W.bst[F]);
    const uint32_t nOn = n - bN, nA = nOn + o.nl;
    const uint16_t* nlist = W.nlist[o.cur];
    uint32_t ac = ctr.ac, sc = ctr.sc;
    if (nA != o.nLive || nA >= 4096u) WC_RET(12);
    uint32_t ltop = 8u * LT::kNR;
    auto carve = [&](uint32_t cnt16) -> uint32_t { ltop -= (cnt16 + 7u) & ~7u; return ltop; };
    if (4u * o.rtop + 5u * (nA + 8u) > ltop) WC_RET(13);
    const uint32_t clen = carve(nA), coff = carve(nA), lab = carve(nA), local = carve(nA), olo = carve(nA);
    // packed index of a survivor: originals by their place in the never-clipped bucket, cut points behind them in creation
    // order (their tail holds it from here on)
    auto idof = [&](uint32_t i) -> uint32_t { return i < nOn ? bN + i : (uint32_t)nlist[i - nOn]; };
    auto recof = [&](uint32_t id) -> WcRec { if (id < WC_MAXN) { const WcW4 wr = g.grec[id]; return WcRec{wr.a, wr.b, wr.c, wr.d}; } return wc_rec(B, 8u * (id - WC_MAXN), true); };
    for (uint32_t i = tid; i < o.nl; i += G) wc_st16(B, 4u * ((uint32_t)nlist[i] - WC_MAXN) + 3u, 0x4000u | (nOn + i));
    WSTAMP_DECL;
    auto lenfn = [&](uint32_t i) -> uint2 { return make_uint2(i < nOn ? (g.grec[bN + i].d >> 24) & 7u : 3u, 0u); };
    const uint32_t H = wc_scan(W, sc, nA, lenfn, lenfn, [&](uint32_t i, uint32_t xo, uint32_t) {
        wc_st16(B, clen + i, lenfn(i).x); wc_st16(B, coff + i, xo); wc_st16(B, lab + i, i); }).x;
    if (H >= 0xFFFFu || 4u * o.rtop + H + 16u > ltop) WC_RET(14);
    const uint32_t cring = carve(H);
    __syncthreads();
    WSTAMP(10);
    bool odd = false;
    for (uint32_t i = tid; i < nA; i += G)
    {
        const WcRec r = recof(idof(i));
        const uint32_t off = wc_ld16(B, coff + i), len = wc_ld16(B, clen + i);
        for (uint32_t q = 0; q < len; ++q)
        {
            const uint32_t e = r.e(q);
            uint32_t rk = 0;
            if (e < WC_MAXN) { if (e < bN || e >= n) odd = true; else rk = e - bN; }
            else if (e >= WC_SENT) odd = true;
            else { const uint32_t te = wc_ld16(B, 4u * (e - WC_MAXN) + 3u); if ((te & 0xC000u) != 0x4000u) odd = true; rk = te & 0xFFFu; }
            if (rk >= nA) { odd = true; rk = 0; }
            wc_st16(B, cring + off + q, rk);
        }
    }
    if (wc_any(W, ac, odd)) WC_RET(15);             // a survivor links a vertex that is gone: not a regular result
    WSTAMP(11);
    // min-label propagation with one pointer jump per round: labels end as the lowest vertex of the island
    while (true)
    {
        bool ch = false;
        for (uint32_t i = tid; i < nA; i += G)
        {
            const uint32_t mine = wc_ld16(B, lab + i), off = wc_ld16(B, coff + i), len = wc_ld16(B, clen + i);
            uint32_t m = mine;
            for (uint32_t q = 0; q < len; ++q) { const uint32_t l2 = wc_ld16(B, lab + wc_ld16(B, cring + off + q)); m = l2 < m ? l2 : m; }
            const uint32_t mm = wc_ld16(B, lab + m);
            m = mm < m ? mm : m;
            if (m < mine) { wc_st16(B, lab + i, m); ch = true; }
        }
        if (!wc_any(W, ac, ch)) break;
    }
    WSTAMP(12);
    auto rootfn = [&](uint32_t i) -> uint2 { return make_uint2(wc_ld16(B, lab + i) == i ? 1u : 0u, 0u); };
    const uint32_t ni = wc_scan(W, sc, nA, rootfn, rootfn, [&](uint32_t, uint32_t, uint32_t) {}).x;
    __syncthreads();
    if (tid == 0u) { W.misc[0] = atomicAdd(&A.cursors[0], nA); W.misc[1] = atomicAdd(&A.cursors[1], H); W.misc[2] = atomicAdd(&A.cursors[3], ni); }
    __syncthreads();
    const uint32_t voff = W.misc[0], hoff = W.misc[1], ioff = W.misc[2];
    ctr.ac = ac; ctr.sc = sc;
    if ((uint64_t)voff + nA > A.capV || (uint64_t)hoff + H > A.capH || (uint64_t)ioff + ni > A.capIsl) return SURTR_E_CAPACITY;
    // islands in order of their lowest vertex; inside an island the vertices keep their order (std::set, :1482-1492)
    uint32_t vbase = 0, hbase = 0, root = 0;
    for (uint32_t t = 0; t < ni; ++t)
    {
        // the next root at or after `root` (every wave looks for itself: the labels are final)
        while (root < nA)
        {
            const uint32_t i = root + lane;
            const unsigned long long m = __ballot(i < nA && wc_ld16(B, lab + i) == i);
            if (m) { root += (uint32_t)__builtin_ctzll(m); break; }
            root += SURTR_LANES;
        }
        auto infn = [&](uint32_t i) -> uint2 { const bool in = wc_ld16(B, lab + i) == root; return make_uint2(in ? 1u : 0u, in ? wc_ld16(B, clen + i) : 0u); };
        const uint2 tt = wc_scan(W, sc, nA, infn, infn, [&](uint32_t i, uint32_t lv, uint32_t lh) {
            if (wc_ld16(B, lab + i) != root) return;
            wc_st16(B, local + i, lv); wc_st16(B, olo + i, lh);
            const size_t dv = (size_t)voff + vbase + lv;
            const uint32_t id = idof(i);
            const float4 p = id < WC_MAXN ? g.gpos[id] : g.cpos[id - WC_MAXN];
            A.pos[3u * dv] = p.x; A.pos[3u * dv + 1u] = p.y; A.pos[3u * dv + 2u] = p.z;
            A.llen[dv] = wc_ld16(B, clen + i); A.loff[dv] = hoff + hbase + lh;
        });
        if (tid == 0u) A.isl[ioff + t] = tt;
        __syncthreads();
        for (uint32_t i = tid; i < nA; i += G)
        {
            if (wc_ld16(B, lab + i) != root) continue;
            const uint32_t off = wc_ld16(B, coff + i), len = wc_ld16(B, clen + i);
            int32_t* d = A.nbr + (size_t)hoff + hbase + wc_ld16(B, olo + i);
            for (uint32_t q = 0; q < len; ++q) d[q] = (int32_t)wc_ld16(B, local + wc_ld16(B, cring + off + q));
        }
        vbase += tt.x; hbase += tt.y; root += 1u;
        __syncthreads();
    }
    ctr.sc = sc;
    rec.mv_off = voff; rec.mv_n = nA; rec.mh_off = hoff; rec.mh_n = H; rec.ni = ni; rec.isl_off = ioff;
    WSTAMP(13);
    return 0;
}

} // namespace surtr
