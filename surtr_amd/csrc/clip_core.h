// clip_core.h -- device-side core of the MI355X fracture engine (gfx950, wave64).
//
// One 256-thread workgroup owns one (cell, piece) task and reproduces
// Poly::ClipPolyhedron (reference Src/Poly.cpp:265-500) with the same output
// vertex order, neighbour-ring order and float arithmetic:
//
//   pre-pass   every original vertex is classified against all planes of the
//              cell at once ("first plane that cuts it"); vertices whose
//              incident faces are all cut by one and the same plane can never be
//              touched by the sequential algorithm and are dropped (DESIGN.md
//              "band reduction" -- a result-preserving cull, not an approximation);
//   per plane  classify / create edge-cut vertices in (vertex, slot) order by
//              ballot + prefix sums / relink the cap by face-loop walks.  Clipped
//              vertices stay behind as tombstones: survivors never move, so the
//              reference's per-plane compaction (:464-495) collapses into ONE
//              ordered compaction when the solid is written out;
//   fallback   a plane that puts any vertex exactly in-plane (comp == 0), or any
//              irregular cap, runs the reference's relink/collapse sequence
//              literally on one lane of the same workgroup (still on the GPU).
//
// Topology (ring offsets/lengths/entries, comp) of the reduced solid lives in
// LDS with 16-bit indices (Topo<InLds>); a solid that does not fit is redone by
// the same code on global scratch with 32-bit indices (Topo<InGlobal>).
//
// All arithmetic is float32 without contraction (compile with -ffp-contract=off);
// see oracle/surtr_oracle.cpp for the SimpleMath semantics restated.
#pragma once
#include <hip/hip_runtime.h>      // (the CPU test tier builds these sources with g++ against tests/emul/hip/hip_runtime.h: a single-lane
                                  // emulation that sets SURTR_LANES = 1 and small capacities; test infrastructure only)
#include <stdint.h>
// wave / workgroup geometry (gfx950: wave64)
#ifndef SURTR_LANES
#define SURTR_LANES 64
#define SURTR_LSH 6
#endif
#ifndef SURTR_WG
#define SURTR_WG 256
#endif
#if SURTR_LANES == 64
#define SURTR_WG_WIDE (4u * SURTR_WG)        // k_prep_pairs_wide: events of so few pairs that every pair gets a quarter of a CU
#else
#define SURTR_WG_WIDE SURTR_WG               // (the single-lane CPU build runs one thread per workgroup)
#endif
#define SURTR_NWAVE_WIDE (SURTR_WG_WIDE / SURTR_LANES)
#define SURTR_NWAVE (SURTR_WG / SURTR_LANES)   // the LARGEST group a kernel is launched with; smaller launches use fewer
#ifndef SURTR_DBG
#define SURTR_DBG(...)
#endif

// Diagnostic build only (-DSURTR_STAMP): lane 0 accumulates s_memtime deltas per phase into a
// global table that no product code reads.
#ifdef SURTR_STAMP
__device__ unsigned long long g_stamp[96];
__device__ unsigned long long g_stamp2[64];      // per-task cost of the per-fragment kernels
#ifdef SURTR_STAMP_SMALL
#define STAMP_WHO (blockDim.x == 64 && gridDim.x > 1900)
#else
#define STAMP_WHO (blockDim.x > 64)
#endif
#define STAMP_DECL unsigned long long st_t0 = __builtin_readcyclecounter(); unsigned long long st_t1
#ifdef SURTR_STAMP_SMALL
// one-wave kernels: accumulate in LDS only (thousands of workgroups on one global counter would measure the atomics)
// (phases 76..78 of the selection go to slots 1..3, which the one-wave kernels' load_whole path leaves free)
#define STAMP(i) do { if (threadIdx.x == 0 && STAMP_WHO) { st_t1 = __builtin_readcyclecounter(); sh.ph[((i) >= 76 && (i) <= 78 ? (i) - 75 : (i)) & 15] += st_t1 - st_t0; st_t0 = st_t1; } } while (0)
#else
#define STAMP(i) do { if (threadIdx.x == 0 && STAMP_WHO) { st_t1 = __builtin_readcyclecounter(); atomicAdd(&g_stamp[i], st_t1 - st_t0); sh.ph[(i) & 15] += st_t1 - st_t0; st_t0 = st_t1; } } while (0)
#endif
#define COUNT(i) do { if (threadIdx.x == 0 && STAMP_WHO) atomicAdd(&g_stamp[i], 1ull); } while (0)
#else
#define COUNT(i) do { } while (0)
#define STAMP_DECL
#define STAMP(i) do { } while (0)
#endif

#define SURTR_MAXF 127          // planes per cell (Voronoi cells have ~15, an ACH k-DOP up to 72)
#define SURTR_DEAD (-3)          // comp of a tombstone
// a value every lane holds equally, moved to a scalar register so that branches on it are scalar branches
#define SURTR_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
// vertices per bounding sphere of the spatially sorted copy (pre-pass A0); a wave of A1 takes LANES / SB undecided groups
#ifndef SURTR_SB
#define SURTR_SB 8u
#endif
#ifndef SURTR_KEEPALL_V
#define SURTR_KEEPALL_V 1536u   // solids up to this many vertices skip the pre-pass culling (they fit the LDS topology whole)
#endif
#define SURTR_NEVER 0xFFu       // fc of a vertex no plane clips
#ifndef SURTR_WALK0
#define SURTR_WALK0 3u         // walk steps before the cap-run shortcut is built (configs[3]: 1 1.93, 2 1.90, 3 1.88, 4 1.89, 6 1.91, 10 1.95 ms)
#endif
#define SURTR_OVERFLOW 100       // internal: the solid does not fit this Topo, redo with the larger one

// LDS-resident topology: capacities per workgroup
#ifndef SURTR_LV
#define SURTR_LV 5632            // vertex slots
#endif
#ifndef SURTR_LH
#define SURTR_LH 24576           // ring entries (16-bit); LdsTopo + Shared stay under 80 KiB => 2 workgroups per CU
#endif

namespace surtr {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (SURTR_LANES - 1u); }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> SURTR_LSH; }

// Cross-lane moves without the LDS crossbar.  `__shfl*` compiles to ds_bpermute_b32 on gfx950 (an LDS-pipeline round trip per
// call: six dependent ones per wave scan); the data-parallel-primitive modifiers move data between lanes inside the VALU.
// lane i <- lane (i - N) of its row of 16 (row_shr:N), lanes without a source keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_move(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}
// value of a wave-uniform lane, through a scalar register (v_readlane_b32)
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)SURTR_UNIFORM(lane)); }
__device__ __forceinline__ float lane_bcast(float v, uint32_t lane) { return __uint_as_float(lane_bcast(__float_as_uint(v), lane)); }
__device__ __forceinline__ int lane_bcast(int v, uint32_t lane) { return (int)lane_bcast((uint32_t)v, lane); }


// hist[f] += number of active lanes holding f: one LDS atomic per distinct value per wave instead of one per lane.
__device__ __forceinline__ void wave_hist_add(uint32_t* hist, uint32_t f, bool active)
{
    unsigned long long todo = __ballot(active);
    while (todo)
    {
        const int leader = __builtin_ctzll(todo);
        const uint32_t f0 = lane_bcast(f, (uint32_t)leader);
        const unsigned long long same = __ballot(active && f == f0);
        if ((int)lane_id() == leader) atomicAdd(&hist[f0], (uint32_t)__builtin_popcountll(same));
        todo &= ~same;
    }
}

struct InLds
{
    typedef uint16_t off_t; typedef uint8_t len_t; typedef uint16_t idx_t;
    static constexpr uint32_t SENT = 0xFFFEu, REM = 0xFFFFu, MAXLEN = 255u;
};
struct InGlobal
{
    typedef uint32_t off_t; typedef uint32_t len_t; typedef uint32_t idx_t;
    static constexpr uint32_t SENT = 0xFFFFFFFEu, REM = 0xFFFFFFFFu, MAXLEN = 0x7FFFFFFFu;
};

// The reduced solid of one task.  Slots [0,nS) are vertices in creation order; comp == SURTR_DEAD marks
// a clipped one.  ring entries >= TT::SENT are not vertices (SENT: a dropped original, REM: marked for removal).
template <class TT>
struct Topo
{
    typename TT::off_t* loff; typename TT::len_t* llen; typename TT::idx_t* ring;
    // fc[v]: the first plane that clips vertex v (SURTR_NEVER: none).  It is known when the vertex enters the solid
    // (pre-pass for input vertices, creation for cut points: positions never change), so a plane needs no
    // classification pass: at plane kcur a slot is dead if fc < kcur, clipped if fc == kcur, kept otherwise.
    // The exception are planes some live vertex lies in (|s| < 1e-10, sh.nzero[k] != 0): they are classified
    // explicitly into gcomp (zmode) and take the reference's general path.
    uint8_t* fc;
    int8_t* gcomp;                                         // global scratch i8[capV], valid in zmode only
    float* pos;                                            // 3 floats per slot (global scratch)
    uint32_t* succ; uint32_t* pred; uint32_t* pcnt;        // per new vertex of the current plane, global u32[capV] (streamed)
    uint32_t* aux0; uint32_t* aux1; uint32_t* aux2; uint32_t* aux3;   // u32[capV], global scratch
    uint2* blk;                                            // scan blocks for capV slots
    uint32_t capV, capH;
    uint32_t nS, nLive, hUsed;
    uint32_t kcur, n0cur; bool zmode;                      // plane in progress, first slot of its new vertices, see above

    // comp of the reference (:307-318) for the plane in progress: -1 clipped, 0 in-plane, 1 kept, 2 new, SURTR_DEAD
    __device__ __forceinline__ int cmp(uint32_t v) const
    {
        if (zmode) return gcomp[v];
        const uint32_t f = fc[v];
        return f < kcur ? SURTR_DEAD : (f == kcur ? -1 : (v >= n0cur ? 2 : 1));
    }
    __device__ __forceinline__ bool alive(uint32_t v) const { return fc[v] >= kcur; }
    // removes v with the plane in progress (two-neighbour collapse, :433-462)
    __device__ __forceinline__ void kill(uint32_t v) { fc[v] = (uint8_t)kcur; if (zmode) gcomp[v] = -1; }
};

struct Shared
{
    float4 planes[SURTR_MAXF + 1];
    union {
        float4 pmar[SURTR_MAXF + 1];  // per plane: ball-test margin = rad*x + y + z*(|px|+|py|+|pz|)
        uint32_t pw[4 * (SURTR_MAXF + 1)];      // the same bytes as counters, once the sphere tests are done (prep_sorted.h)
    };
    uint32_t hist[SURTR_MAXF + 1];    // after the pre-pass: dropped vertices still alive after plane k
    uint32_t zhist[SURTR_MAXF + 1];   // dropped vertices that lie in plane k while still alive
    uint32_t nzero[SURTR_MAXF + 1];   // vertices of the reduced solid that lie in plane k before any plane clips them (>0: general path)
    uint32_t wsum[2 * SURTR_NWAVE_WIDE];
    uint32_t flagBad, flagErr;
    uint32_t deg7;                    // pre-pass: a kept vertex has more than seven ring entries (no 16-byte record holds it)
    uint32_t pf[3][8];                // per-plane flags, triple buffered: 0 cut, 1 keep, 2 in-plane, 3 dup, 4 pred, 5 live, 6 long
    uint32_t changed;
    uint32_t misc[8];
    uint32_t cutmask[4];              // prepass_emit: bit k = plane k is the first clipping plane of some vertex of the reduced solid
#ifdef SURTR_STAMP
    unsigned long long ph[16];        // per-pair phase cycles (diagnostic build)
#endif
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
    const float* rad;       // per vertex: radius of a ball around it holding every vertex of its incident faces (nullptr = unknown)
    // spatially sorted copy for the pre-pass (nullptr = absent): sorted index i is vertex perm[i]; bsph[b] bounds
    // the balls of the SURTR_SB vertices of sorted group b (centre xyz, radius w)
    const uint32_t* perm; const float4* posr_s; const float4* bsph;      // posr_s: (x, y, z, ball radius) of sorted vertex i: one 16-byte load
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
    // The reference compares std::abs(float) with the double literal 1.0e-10.  (float)1e-10 = 0x2EDBE6FF is
    // >= 1e-10 and its predecessor is < 1e-10, so for every float x: (double)|x| < 1e-10  <=>  |x| < 1.0e-10f.
    if (fabsf(s) < 1.0e-10f) return 0;
    float m = -s;
    return m > 0.f ? 1 : (m < 0.f ? -1 : 0);
}

// Size of the launched group and its wave count (kernels for small solids run one wave per task).
__device__ __forceinline__ uint32_t group_size() { return blockDim.x; }
__device__ __forceinline__ uint32_t group_waves() { return blockDim.x >> SURTR_LSH; }

// Inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the row totals
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Six VALU operations per value.
__device__ __forceinline__ uint2 wave_incl_scan2(uint2 v)
{
    v.x += dpp_move<0x111, 0xF>(0u, v.x); v.y += dpp_move<0x111, 0xF>(0u, v.y);
    v.x += dpp_move<0x112, 0xF>(0u, v.x); v.y += dpp_move<0x112, 0xF>(0u, v.y);
    v.x += dpp_move<0x114, 0xF>(0u, v.x); v.y += dpp_move<0x114, 0xF>(0u, v.y);
    v.x += dpp_move<0x118, 0xF>(0u, v.x); v.y += dpp_move<0x118, 0xF>(0u, v.y);
    v.x += dpp_move<0x142, 0xA>(0u, v.x); v.y += dpp_move<0x142, 0xA>(0u, v.y);
    v.x += dpp_move<0x143, 0xC>(0u, v.x); v.y += dpp_move<0x143, 0xC>(0u, v.y);
    return v;
}

__device__ __forceinline__ uint2 wave_excl2(uint2 c)
{
    uint2 s = wave_incl_scan2(c);
    return make_uint2(s.x - c.x, s.y - c.y);
}

// Exclusive scan, in place, of the per-64-block (count, weight) array; returns the totals.
__device__ inline void scan_block_array(uint32_t nb, uint2* blk, Shared& sh, uint32_t& totC, uint32_t& totW)
{
    const uint32_t l = lane_id(), w = wave_id();
    uint2 carry = make_uint2(0u, 0u);
    for (uint32_t c0 = 0; c0 < nb; c0 += group_size())
    {
        const uint32_t i = c0 + threadIdx.x;
        uint2 x = make_uint2(0u, 0u);
        if (i < nb) x = blk[i];
        uint2 inc = wave_incl_scan2(x);
        if (l == SURTR_LANES - 1u) { sh.wsum[2 * w] = inc.x; sh.wsum[2 * w + 1] = inc.y; }
        __syncthreads();
        uint2 woff = make_uint2(0u, 0u), tot = make_uint2(0u, 0u);
        for (uint32_t q = 0; q < group_waves(); ++q)
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

// Ordered two-level scan over items [0,n).  fn(i) -> (count, weight).  Afterwards blk[b] holds the exclusive
// (count, weight) base of 64-block b; the position inside the block comes from wave_excl2 in the caller's
// second sweep.  Every thread must call it (barriers inside).
template <class Fn>
__device__ void scan_blocks(uint32_t n, uint2* blk, Shared& sh, Fn fn, uint32_t& totC, uint32_t& totW)
{
    const uint32_t nb = (n + SURTR_LANES - 1u) >> SURTR_LSH;
    const uint32_t l = lane_id(), w = wave_id();
    for (uint32_t b = w; b < nb; b += group_waves())
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

// FaceLoop, Src/Poly.cpp:34-41: the ring entry listed just before `prev` (cyclic).
template <class I>
__device__ __forceinline__ uint32_t face_next(const I* ring, uint32_t len, uint32_t prev)
{
    if (len <= 4u)
    {
        // the common case (new vertices have 3 entries, most others <= 4): fetch the whole ring at once so a walk
        // step costs one LDS round trip instead of one per entry; indices are clamped, so no read leaves the ring
        const uint32_t l1 = len > 1u ? 1u : 0u, l2 = len > 2u ? 2u : 0u, l3 = len > 3u ? 3u : 0u;
        const uint32_t e0 = ring[0], e1 = ring[l1], e2 = ring[l2], e3 = ring[l3];
        const uint32_t last = len == 4u ? e3 : (len == 3u ? e2 : (len == 2u ? e1 : e0));
        if (e0 == prev) return last;
        if (len > 1u && e1 == prev) return e0;
        if (len > 2u && e2 == prev) return e1;
        if (len > 3u && e3 == prev) return e2;
        return last;     // not found: std::find returns end(), the reference then reads *(end-1)
    }
    uint32_t k = 0;
    while (k < len && (uint32_t)ring[k] != prev) ++k;
    if (k == 0) return (uint32_t)ring[len - 1];
    return (uint32_t)ring[k - 1];
}

__device__ __forceinline__ int32_t face_next_in(const int32_t* ring, uint32_t len, int32_t prev)
{
    uint32_t k = 0;
    while (k < len && ring[k] != prev) ++k;
    if (k == 0) return ring[len - 1];
    return ring[k - 1];
}

// ---------------------------------------------------------------------------
// Literal relink of one plane (Src/Poly.cpp:367-425) on one lane: new vertices in creation order, then the in-plane
// ones ascending (:370-373).  Rings of in-plane vertices were given room for the insertions; snap = old_neighbors.
//
// Only the vertices whose result depends on the processing order take part (slist, then zlist).  A new vertex Y that
// exactly one walk ends on, started by another new vertex ("easy"), always ends as [that source, its own target,
// kept]: its own slot is replaced by its target (the entry after it is the kept end, never the target, :399), and
// the one insertion goes to the front (:404-408) whenever it happens.  The caller writes those rings afterwards;
// here an insertion into an easy vertex is skipped.  wcur/wprev: end and last hop of the walk of every new vertex
// (from the dry run -- walks only cross clipped vertices, whose rings the relink never changes).
// (T by value: a reference would force the caller's Topo into every lane's private memory for the whole plane loop.)
template <class TT>
__device__ void relink_serial(const Topo<TT> T, uint32_t n0, uint32_t n1, uint32_t nref, const uint32_t* snapoff, const uint32_t* cap,
                              const uint32_t* zlist, uint32_t nz, const uint32_t* slist, uint32_t ns,
                              const uint32_t* wcur, const uint32_t* wprev, const uint32_t* arrive, const uint32_t* srcof, Shared& sh)
{
    typedef typename TT::idx_t I;
    auto easy = [&](uint32_t c) -> bool {
        return c >= n0 && c < n1 && arrive[c] == 1u && srcof[c] >= n0 && wcur[c - n0] < TT::SENT && wcur[c - n0] != c;
    };
    for (uint32_t t = 0; t < ns + nz; ++t)
    {
        const uint32_t i = t < ns ? slist[t] : zlist[t - ns];
        const uint32_t deg = T.llen[i];
        for (uint32_t j = 0; j < deg; ++j)
        {
            I* ri = T.ring + T.loff[i];
            const uint32_t jn = ri[j];
            if (jn >= TT::SENT || T.cmp(jn) != -1) continue;
            uint32_t prev = i, cur = jn;
            if (i >= n0) { cur = wcur[i - n0]; prev = wprev[i - n0]; }
            else
            {
                uint32_t steps = 0;
                while (cur < TT::SENT && T.cmp(cur) == -1 && steps++ < nref)      // (the reference's bound: its vertex count)
                {
                    const uint32_t hold = cur;
                    cur = face_next(T.ring + T.loff[cur], T.llen[cur], prev);
                    prev = hold;
                }
            }
            if (cur >= TT::SENT) { SURTR_DBG("serial: walk hit a non-vertex i=%u\n", i); sh.flagErr = 1; return; }
            if ((uint32_t)ri[(j + 1u) % T.llen[i]] == cur || cur == i)
            {
                ri[j] = (I)TT::REM;
            }
            else
            {
                ri[j] = (I)cur;
                if (easy(cur)) continue;
                I* rc = T.ring + T.loff[cur];
                const uint32_t lc = T.llen[cur];
                if (T.cmp(cur) == 2)
                {
                    if (lc >= cap[cur]) { SURTR_DBG("serial: comp2 ring full cur=%u\n", cur); sh.flagErr = 1; return; }
                    for (uint32_t q = lc; q > 0; --q) rc[q] = rc[q - 1];
                    rc[0] = (I)i;
                    T.llen[cur] = (typename TT::len_t)(lc + 1);
                }
                else if (T.cmp(cur) == 0 && cur < n0)
                {
                    if (lc >= cap[cur]) { SURTR_DBG("serial: comp0 ring full cur=%u\n", cur); sh.flagErr = 1; return; }
                    I* sn = T.ring + snapoff[cur];
                    uint32_t at = 0;
                    while (at < lc && (uint32_t)sn[at] != prev) ++at;
                    for (uint32_t q = lc; q > at; --q) { rc[q] = rc[q - 1]; sn[q] = sn[q - 1]; }
                    rc[at] = (I)i; sn[at] = (I)i;
                    T.llen[cur] = (typename TT::len_t)(lc + 1);
                }
                else { SURTR_DBG("serial: walk ended on comp %d vertex %u\n", (int)T.cmp(cur), cur); sh.flagErr = 1; return; }
            }
        }
    }
}

// Two-neighbour vertices, Src/Poly.cpp:433-462, literal and serial (rare).
template <class TT>
__device__ void collapse_serial(Topo<TT>& T, uint32_t n1)
{
    typedef typename TT::idx_t I;
    bool again = true;
    while (again)
    {
        again = false;
        for (uint32_t i = 0; i < n1; ++i)
        {
            if (T.cmp(i) >= 0 && T.llen[i] == 2u)
            {
                again = true;
                const uint32_t a = T.ring[T.loff[i]], b = T.ring[T.loff[i] + 1];
                if (a < TT::SENT)
                {
                    I* ra = T.ring + T.loff[a];
                    for (uint32_t q = 0; q < T.llen[a]; ++q) if ((uint32_t)ra[q] == i) { ra[q] = (I)b; break; }
                }
                if (b < TT::SENT)
                {
                    I* rb = T.ring + T.loff[b];
                    for (uint32_t q = 0; q < T.llen[b]; ++q) if ((uint32_t)rb[q] == i) { rb[q] = (I)a; break; }
                }
                T.kill(i);
            }
        }
    }
}

// A kept vertex: its ring entries are added to its 64-block's count; a ring too long for a narrow (8-bit) length is flagged.
__device__ __forceinline__ void prepass_keep_deg(uint2* bblk, Shared& sh, uint32_t v, uint32_t deg)
{
    atomicAdd(&bblk[v >> SURTR_LSH].y, deg);
    if (deg > InLds::MAXLEN / 2u) sh.flagBad = 1;
    if (deg > 7u) sh.deg7 = 1;
}

// A2 of the pre-pass: the exact test, densely over the work list `needy` (v | fc << 24): a vertex is dropped iff every vertex
// of every incident face has the same first clipping plane.  klist / kcount (optional, the sorted pre-pass of k_prep_pairs): the
// kept vertices are appended as sorted index | (first clipping plane | 0x80 when the vertex lies in an earlier plane) << 16 | ring
// length << 24;
// sh.misc[5] = some kept vertex does.
template <int NB>
__device__ inline void prepass_exact(const SolidIn in, const uint32_t F, Shared& sh, unsigned long long* bmask, uint2* bblk,
                                     const uint32_t* needy, const uint32_t nNeedy, uint32_t* klist = nullptr, uint32_t* kcount = nullptr, const uint32_t* iperm = nullptr)
{
    const uint32_t l = lane_id(), w = wave_id();
    const uint32_t V = in.nv;
    auto keep_deg = [&](uint32_t v, uint32_t deg) { prepass_keep_deg(bblk, sh, v, deg); };
    // ---- A2: the exact test, densely over the work list (neighbour loads batched: latency rules here) ----
#ifdef SURTR_STAMP
    if (threadIdx.x == 0 && V > 10000u) atomicAdd(&g_stamp[47], (unsigned long long)nNeedy);
#endif
    for (uint32_t i0 = w << SURTR_LSH; i0 < nNeedy; i0 += group_size())
    {
        const uint32_t i = i0 + l;
        bool keep = false, drop = false, zero = false; uint32_t f = 0, v = 0;
        if (i < nNeedy)
        {
            const uint32_t e = needy[i];
            v = e & 0xFFFFFFu; f = e >> 24;
            const uint32_t deg = in.llen[v];
            const int32_t* r = in.nbr + in.loff[v];
            auto same_fc = [&](float x, float y, float z) -> bool {
                for (uint32_t k = 0; k < f; ++k)
                    if (side_of(plane_dist(sh.planes[k], x, y, z)) < 0) return false;
                return side_of(plane_dist(sh.planes[f], x, y, z)) < 0;
            };
            for (uint32_t j0 = 0; j0 < deg && !keep; j0 += NB)
            {
                int32_t u[NB]; float ux[NB], uy[NB], uz[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) u[q] = (j0 + q < deg) ? r[j0 + q] : -1;
#pragma unroll
                for (int q = 0; q < NB; ++q)
                {
                    const int32_t uu = u[q] < 0 ? (int32_t)v : u[q];
                    ux[q] = in.pos[3 * uu]; uy[q] = in.pos[3 * uu + 1]; uz[q] = in.pos[3 * uu + 2];
                }
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (u[q] >= 0 && !same_fc(ux[q], uy[q], uz[q])) keep = true;
            }
            if (!keep && !(in.tri != nullptr && in.tri[v]))
            {
                // faces that are not triangles: walk every incident face loop
                for (uint32_t j = 0; j < deg && !keep; ++j)
                {
                    int32_t prev = (int32_t)v, cur = r[j];
                    uint32_t steps = 0;
                    while (cur != (int32_t)v && steps++ < V)
                    {
                        if (!same_fc(in.pos[3 * cur], in.pos[3 * cur + 1], in.pos[3 * cur + 2])) { keep = true; break; }
                        const int32_t nx = face_next_in(in.nbr + in.loff[cur], in.llen[cur], prev);
                        prev = cur; cur = nx;
                    }
                }
            }
            drop = !keep;
            if (keep)
            {
                // (the sorted pre-pass builds its masks from the kept list afterwards: bmask == nullptr)
                if (bmask != nullptr) { atomicOr(&bmask[v >> SURTR_LSH], 1ull << (v & (SURTR_LANES - 1u))); keep_deg(v, deg); }
                if (klist != nullptr)
                {
                    const float px = in.pos[3 * v], py = in.pos[3 * v + 1], pz = in.pos[3 * v + 2];
                    uint32_t z = 0;
                    for (uint32_t k = 0; k < f; ++k) if (side_of(plane_dist(sh.planes[k], px, py, pz)) == 0) z = 0x80u;
                    klist[atomicAdd(kcount, 1u)] = iperm[v] | ((f | z) << 16) | ((deg < 255u ? deg : 255u) << 24);
                    if (z) sh.misc[5] = 1u;
                }
            }
            else
            {
                // in-plane at an earlier plane while alive: it is no "kept" vertex there
                const float px = in.pos[3 * v], py = in.pos[3 * v + 1], pz = in.pos[3 * v + 2];
                for (uint32_t k = 0; k < f; ++k)
                    if (side_of(plane_dist(sh.planes[k], px, py, pz)) == 0) { zero = true; atomicAdd(&sh.zhist[k], 1u); }
            }
        }
        (void)zero;
        wave_hist_add(sh.hist, f, drop);
    }
}

// ---------------------------------------------------------------------------
// The pre-pass has two halves, so that a kernel of its own can run the first one without a Topo:
//   prepass_select  which vertices form the reduced solid (bit mask + per-block counts), hist/zhist; n vertices,
//                   hsum ring entries; sh.flagBad = a kept vertex has more neighbours than a narrow ring may hold
//   prepass_emit    writes the reduced solid into T (any memory)
// needy/und: u32 work lists (V and V/64 entries).
template <int G, int NB>
__device__ inline void prepass_select(const SolidIn in, const uint32_t F, Shared& sh, unsigned long long* bmask, uint2* bblk,
                                      uint32_t* needy, uint32_t* und, uint32_t& n_out, uint32_t& hsum_out)
{
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id();
    const uint32_t V = in.nv;
    STAMP_DECL;
    for (uint32_t k = tid; k <= SURTR_MAXF; k += group_size())
    {
        sh.hist[k] = 0; sh.zhist[k] = 0; sh.nzero[k] = 0;
        if (k < F)
        {
            // conservative: distance to the plane of anything within r of p is >= (|n.p+d| - r|n|)/|n|; the second and
            // third terms bound the float rounding of n.p + d for p and for the points of the ball
            const float4 pk = sh.planes[k];
            const float n1 = fabsf(pk.x) + fabsf(pk.y) + fabsf(pk.z);                         // >= |n|, bounds the rounding terms
            const float n2 = sqrtf(pk.x * pk.x + pk.y * pk.y + pk.z * pk.z) * 1.0001f;       // |n|, rounded up
            sh.pmar[k] = make_float4(n2 * 1.00101f, 1.0e-5f * fabsf(pk.w), 1.0e-5f * n1, 0.f);
        }
    }
    if (tid == 0) { sh.flagErr = 0; sh.flagBad = 0; sh.misc[3] = 0; }
    __syncthreads();
    const uint32_t nbV = (V + SURTR_LANES - 1u) >> SURTR_LSH;
    // needy: work list of vertices that need the exact neighbour test: v | fc << 24

    // A solid of a few hundred vertices (a Convex, a refit box) is kept whole: the plane loop only touches what a
    // plane clips anyway, and the tests below would cost more than they save.
    // bblk[b].y collects the ring entries of the kept vertices of block b wherever a vertex is found to be kept (its
    // degree is at hand or one more load in flight there), so that the block table needs no pass over the vertices
    const bool keepall = V <= SURTR_KEEPALL_V;
    for (uint32_t b = tid; b < nbV; b += group_size()) bblk[b] = make_uint2(0u, 0u);
    __syncthreads();
    auto keep_deg = [&](uint32_t v, uint32_t deg) { prepass_keep_deg(bblk, sh, v, deg); };
    if (keepall)
    {
        for (uint32_t b = tid; b < nbV; b += group_size())
        {
            const uint32_t left = V - (b << SURTR_LSH);
            bmask[b] = left >= SURTR_LANES ? (~0ull >> (64u - SURTR_LANES)) : ((1ull << left) - 1ull);
        }
        for (uint32_t v = tid; v < V; v += group_size()) keep_deg(v, in.llen[v]);
    }
    if (!keepall)
    {
    // ---- A1: stream all vertices: first cutting plane + conservative ball test, no neighbour is read ----
    // If the ball that holds every vertex of v's incident faces stays strictly on v's side of every plane up
    // to and including fc(v), all those vertices have the same fc and v is dropped right here.
    const bool sorted = in.bsph != nullptr && in.perm != nullptr && V < (1u << 24);
    uint32_t nWork = nbV, nUnd = 0;       // wave-loads (64 vertices) that need the per-vertex pass; undecided groups
    // und: (sorted path) the blocks the sphere test could not decide
    if (sorted)
    {
        // ---- A0: one lane per group of SURTR_SB spatially close vertices: if the sphere around their balls is entirely on
        // the cut side of plane k and entirely on the kept side of planes 0..k-1, all of them have fc = k and every one
        // is dropped (same argument as the per-vertex ball test); nothing of the group is read.
        for (uint32_t bq = tid; bq < nbV; bq += group_size()) bmask[bq] = 0ull;
        if (tid == 0) sh.misc[4] = 0;
        __syncthreads();
        const uint32_t nsb = (V + SURTR_SB - 1u) / SURTR_SB;
        for (uint32_t sb = tid; sb < nsb; sb += group_size())
        {
            const float4 sp = in.bsph[sb];
            const float mag = fabsf(sp.x) + fabsf(sp.y) + fabsf(sp.z) + sp.w;
            uint32_t f = 0xFFu;
            for (uint32_t k = 0; k < F; ++k)
            {
                const float4 mk = sh.pmar[k];
                const float sk = plane_dist(sh.planes[k], sp.x, sp.y, sp.z);
                const float margin = sp.w * mk.x + mk.y + mk.z * mag;
                if (sk > margin) { f = k; break; }
                if (!(sk < -margin)) break;
            }
            if (f != 0xFFu)
            {
                const uint32_t left = V - sb * SURTR_SB;
                atomicAdd(&sh.hist[f], left < SURTR_SB ? left : (uint32_t)SURTR_SB);
            }
            else und[atomicAdd(&sh.misc[4], 1u)] = sb;
        }
        __syncthreads();
        nUnd = sh.misc[4];
        nWork = (nUnd + SURTR_LANES / SURTR_SB - 1u) / (SURTR_LANES / SURTR_SB);     // wave-loads of undecided groups
#ifdef SURTR_STAMP
        if (tid == 0 && V > 10000u) { atomicAdd(&g_stamp[45], (unsigned long long)nUnd); atomicAdd(&g_stamp[46], (unsigned long long)nsb); }
#endif
    }
    for (uint32_t b0 = w; b0 < nWork; b0 += (uint32_t)G * group_waves())
    {
        // G (four) 64-blocks per wave iteration: their loads are in flight together, and every plane fetched
        // from LDS is applied to all four (planes outermost: one LDS fetch per plane, four independent chains)
        float px4[G], py4[G], pz4[G], rv4[G], mag4[G];
        uint32_t f4[G], id4[G]; bool done4[G], clear4[G], valid4[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            const uint32_t wb = b0 + g * group_waves();
            uint32_t i = (wb << SURTR_LSH) + l;
            bool ok = wb < nWork;
            if (sorted)
            {
                // lane l reads vertex l % SB of the (l / SB)-th undecided group of this wave-load
                const uint32_t sub = wb * (SURTR_LANES / SURTR_SB) + l / SURTR_SB;
                ok = ok && sub < nUnd;
                i = (ok ? und[sub] : 0u) * SURTR_SB + (l % SURTR_SB);
            }
            valid4[g] = ok && i < V;
            const uint32_t ii = valid4[g] ? i : 0u;
            if (sorted)
            {
                id4[g] = in.perm[ii];
                const float4 pr = in.posr_s[ii];
                px4[g] = pr.x; py4[g] = pr.y; pz4[g] = pr.z; rv4[g] = pr.w;
            }
            else
            {
                id4[g] = ii;
                px4[g] = in.pos[3 * ii]; py4[g] = in.pos[3 * ii + 1]; pz4[g] = in.pos[3 * ii + 2];
                rv4[g] = (in.rad != nullptr && V < (1u << 24)) ? in.rad[ii] : -1.f;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            mag4[g] = fabsf(px4[g]) + fabsf(py4[g]) + fabsf(pz4[g]);
            f4[g] = 0xFFu; done4[g] = !valid4[g]; clear4[g] = rv4[g] >= 0.f;
        }
        for (uint32_t k = 0; k < F; ++k)
        {
            bool all_done = true;
#pragma unroll
            for (int g = 0; g < G; ++g) all_done = all_done && done4[g];
            if (__all(all_done)) break;
            const float4 pk = sh.planes[k];
            const float4 mk = sh.pmar[k];
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                if (done4[g]) continue;
                const float sk = plane_dist(pk, px4[g], py4[g], pz4[g]);
                const int c = side_of(sk);
                if (clear4[g] && !(fabsf(sk) > rv4[g] * mk.x + mk.y + mk.z * mag4[g])) clear4[g] = false;
                if (c < 0) { f4[g] = k; done4[g] = true; }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            const uint32_t wb = b0 + g * group_waves();
            if (wb >= nWork) break;
            const uint32_t v = id4[g];
            const uint32_t f = f4[g];
            const bool keep = valid4[g] && f == 0xFFu;
            const bool drop = valid4[g] && !keep && clear4[g];     // cannot be in-plane anywhere before fc (|s| > margin there)
            const bool need = valid4[g] && !keep && !clear4[g];
            wave_hist_add(sh.hist, f, drop);
            const unsigned long long mk = __ballot(keep), mn = __ballot(need);
            if (sorted)
            {
                if (keep) atomicOr(&bmask[v >> SURTR_LSH], 1ull << (v & (SURTR_LANES - 1u)));
            }
            else if (l == 0) bmask[wb] = mk;
            if (keep) keep_deg(v, in.llen[v]);
            if (mn)
            {
                uint32_t base = 0;
                if (l == 0) base = atomicAdd(&sh.misc[3], (uint32_t)__builtin_popcountll(mn));
                base = lane_bcast(base, 0u);
                if (need) needy[base + (uint32_t)__builtin_popcountll(mn & ((1ull << l) - 1ull))] = v | (f << 24);
            }
        }
    }
    __syncthreads();
    STAMP(0);
    prepass_exact<NB>(in, F, sh, bmask, bblk, needy, sh.misc[3]);
    }   // !keepall
    __syncthreads();
    STAMP(1);
    // ---- A3: per 64-block (kept vertices, their ring entries): the counts come from the mask, the entries were summed above ----
    for (uint32_t b = tid; b < nbV; b += group_size()) bblk[b].x = (uint32_t)__builtin_popcountll(bmask[b]);
    __syncthreads();
    uint32_t n = 0, hsum = 0;
    scan_block_array(nbV, bblk, sh, n, hsum);
    __syncthreads();
    STAMP(3);
    n_out = n; hsum_out = hsum;
}

// hist[k] := dropped vertices still alive after plane k = sum_{f>k} hist[f] (once per pre-pass, after prepass_select)
__device__ inline void prepass_finish_hist(const uint32_t F, Shared& sh)
{
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t run = 0;
        for (int k = (int)F - 1; k >= 0; --k) { const uint32_t h = sh.hist[k]; sh.hist[k] = run; run += h; }
    }
    __syncthreads();
}

template <class TT>
__device__ __attribute__((always_inline)) inline void prepass_emit(const SolidIn in, const uint32_t F, Shared& sh, Topo<TT>& T, const unsigned long long* bmask,
                             const uint2* bblk, uint32_t* orig, const uint32_t n, const uint32_t hsum)
{
    typedef typename TT::idx_t I;
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id();
    const uint32_t nbV = (in.nv + SURTR_LANES - 1u) >> SURTR_LSH;
    STAMP_DECL;
    // ---- emit: slot table first (sparse sweep, no gathers), then rings densely over the kept vertices ----
    for (uint32_t b = w; b < nbV; b += group_waves())
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
            orig[id] = v;
            T.loff[id] = (typename TT::off_t)(base.y + e.y); T.llen[id] = (typename TT::len_t)deg;
        }
    }
    __syncthreads();
    // index of an original vertex in the reduced solid
    auto newid = [&](int32_t u) -> uint32_t {
        const unsigned long long m = bmask[(uint32_t)u >> SURTR_LSH];
        const uint32_t bit = (uint32_t)u & (SURTR_LANES - 1u);
        if (!((m >> bit) & 1ull)) return TT::SENT;
        return bblk[(uint32_t)u >> SURTR_LSH].x + (uint32_t)__builtin_popcountll(m & ((1ull << bit) - 1ull));
    };
    for (uint32_t id = tid; id < n; id += group_size())
    {
        const uint32_t v = orig[id];
        const float px = in.pos[3 * v], py = in.pos[3 * v + 1], pz = in.pos[3 * v + 2];
        T.pos[3 * id] = px; T.pos[3 * id + 1] = py; T.pos[3 * id + 2] = pz;
        {
            // first clipping plane; planes the vertex lies in before that (:307-318 would give comp 0 there)
            uint32_t f = SURTR_NEVER;
            for (uint32_t k = 0; k < F; ++k)
            {
                const int c = side_of(plane_dist(sh.planes[k], px, py, pz));
                if (c < 0) { f = k; break; }
                if (c == 0) atomicAdd(&sh.nzero[k], 1u);
            }
            T.fc[id] = (uint8_t)f;
            if (f < 128u) atomicOr(&sh.cutmask[f >> 5], 1u << (f & 31u));      // planes that clip something of the band (cost estimate)
        }
        const uint32_t deg = T.llen[id];
        const int32_t* r = in.nbr + in.loff[v];
        I* d = T.ring + T.loff[id];
        for (uint32_t j0 = 0; j0 < deg; j0 += 8)
        {
            int32_t u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = (j0 + q < deg) ? r[j0 + q] : -1;
#pragma unroll
            for (int q = 0; q < 8; ++q) if (u[q] >= 0) d[j0 + q] = (I)newid(u[q]);
        }
    }
    STAMP(2);
    __syncthreads();
    T.nS = n; T.nLive = n; T.hUsed = hsum;
}

// Pre-pass: builds the reduced solid of `in` for the planes sh.planes[0..F) in T.
// bmask/bblk: one word / one (count, ring entries) pair per 64 input vertices (LDS or global).
// capH_emit: ring entries available while the masks are still in use.
// Returns 0 or SURTR_OVERFLOW (does not fit T) -- uniform over the workgroup.
template <class TT>
__device__ __attribute__((always_inline)) inline int prepass(const SolidIn in, const uint32_t F, Topo<TT>& T, Shared& sh, unsigned long long* bmask, uint2* bblk,
                       uint32_t capH_emit, unsigned long long* spill_mask, uint2* spill_blk)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t nbV = (in.nv + SURTR_LANES - 1u) >> SURTR_LSH;
    uint32_t n = 0, hsum = 0;
    prepass_select<4, 8>(in, F, sh, bmask, bblk, T.aux0, T.aux2, n, hsum);
    const bool toolong = TT::MAXLEN <= InLds::MAXLEN && sh.flagBad != 0;
    __syncthreads();
    if (spill_mask != nullptr && spill_mask != bmask)
    {
        // a global copy of the bit mask always exists (clip_planes' all-in-plane corner case reads it).  The LDS
        // masks share the tail of the ring area; a band that needs all of it moves both mask arrays to global scratch.
        const bool spill = hsum > capH_emit && hsum <= T.capH;
        for (uint32_t b = tid; b < nbV; b += group_size()) { spill_mask[b] = bmask[b]; if (spill) spill_blk[b] = bblk[b]; }
        __syncthreads();
        if (spill) { bmask = spill_mask; bblk = spill_blk; capH_emit = T.capH; }
    }
    if (toolong) COUNT(32);
    if (n > T.capV) COUNT(33);
    if (hsum > capH_emit) COUNT(34);
    if (toolong || n > T.capV || hsum > capH_emit || n >= TT::SENT) return SURTR_OVERFLOW;
    prepass_emit(in, F, sh, T, bmask, bblk, T.aux1, n, hsum);
    prepass_finish_hist(F, sh);
    return 0;
}

// A per-plane work list: 16-bit (or 8-bit) entries in the free tail of the LDS ring area when there is room, 32-bit words
// in global scratch otherwise (l == nullptr).  The plane loop waits for one memory round trip per phase on these lists;
// in LDS that wait is an order of magnitude shorter.
template <class E>
struct WArr
{
    E* l; uint32_t* g;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return l ? (uint32_t)l[i] : g[i]; }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { if (l) l[i] = (E)v; else g[i] = v; }
};

// Slots [0,nS) whose byte in `arr` equals `val`, in slot order.  One lane inspects four consecutive slots with one 32-bit
// read (arrays are 4-byte aligned and padded), so a wave covers 256 slots per round trip.
__device__ __forceinline__ uint32_t select_match4(const uint8_t* arr, uint32_t val, uint32_t nS, uint32_t base)
{
    if (base >= nS) return 0u;
    const uint32_t wv = *(const uint32_t*)(arr + base);
    uint32_t m = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) if (base + q < nS && ((wv >> (8u * q)) & 0xFFu) == val) m |= 1u << q;
    return m;
}
// Every wave takes a contiguous range of 256-slot blocks, so that one barrier and a sum over the waves' totals give both
// the total and the wave's base.  select_count returns the total (sh.wsum[w] = matches of wave w); select_write stores
// the list.  Barriers inside: every thread must call them.
__device__ inline uint32_t select_count(const uint8_t* arr, uint32_t val, uint32_t nS, Shared& sh)
{
    const uint32_t l = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t per = 4u * SURTR_LANES;
    const uint32_t nb = (nS + per - 1u) / per, nbw = (nb + nw - 1u) / nw;
    const uint32_t b0 = w * nbw, b1 = b0 + nbw < nb ? b0 + nbw : nb;
    uint32_t mine = 0;
    for (uint32_t b = b0; b < b1; ++b) mine += (uint32_t)__builtin_popcount(select_match4(arr, val, nS, b * per + 4u * l));
    const uint2 inc = wave_incl_scan2(make_uint2(mine, 0u));
    if (l == SURTR_LANES - 1u) sh.wsum[w] = inc.x;
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t q = 0; q < nw; ++q) total += sh.wsum[q];
    return total;
}
template <class L>
__device__ inline void select_write(const uint8_t* arr, uint32_t val, uint32_t nS, Shared& sh, const L list)
{
    const uint32_t l = lane_id(), w = wave_id(), nw = group_waves();
    const uint32_t per = 4u * SURTR_LANES;
    const uint32_t nb = (nS + per - 1u) / per, nbw = (nb + nw - 1u) / nw;
    const uint32_t b0 = w * nbw, b1 = b0 + nbw < nb ? b0 + nbw : nb;
    uint32_t run = 0;
    for (uint32_t q = 0; q < w; ++q) run += sh.wsum[q];
    for (uint32_t b = b0; b < b1; ++b)
    {
        const uint32_t base = b * per + 4u * l;
        const uint32_t m = select_match4(arr, val, nS, base);
        const uint2 s2 = wave_incl_scan2(make_uint2((uint32_t)__builtin_popcount(m), 0u));
        uint32_t at = run + s2.x - (uint32_t)__builtin_popcount(m);
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) if (m & (1u << q)) list.set(at++, base + q);
        run += lane_bcast(s2.x, SURTR_LANES - 1u);
    }
    __syncthreads();
}

// A small solid goes into T whole, without the pre-pass machinery (what prepass() does for a kept-whole solid, minus its
// masks, scans and lists): rings are contiguous per solid, so the ring offset of v is in.loff[v] - in.loff[0].
// Returns 0 or SURTR_OVERFLOW (does not fit T) -- uniform over the workgroup.
template <class TT>
__device__ __attribute__((always_inline)) inline int load_whole(const SolidIn in, const uint32_t F, Topo<TT>& T, Shared& sh)
{
    typedef typename TT::idx_t I;
    const uint32_t tid = threadIdx.x, V = in.nv;
    for (uint32_t k = tid; k <= SURTR_MAXF; k += group_size()) { sh.hist[k] = 0; sh.zhist[k] = 0; sh.nzero[k] = 0; }
    if (tid == 0) { sh.flagErr = 0; sh.flagBad = 0; }
    __syncthreads();
    if (V == 0) { T.nS = T.nLive = T.hUsed = 0; return 0; }
    const uint32_t base = in.loff[0], H = in.loff[V - 1u] + in.llen[V - 1u] - base;
    if (V > T.capV || H > T.capH || V >= TT::SENT) return SURTR_OVERFLOW;
    bool toolong = false;
    for (uint32_t v = tid; v < V; v += group_size())
    {
        const float px = in.pos[3 * v], py = in.pos[3 * v + 1], pz = in.pos[3 * v + 2];
        T.pos[3 * v] = px; T.pos[3 * v + 1] = py; T.pos[3 * v + 2] = pz;
        const uint32_t lo = in.loff[v] - base, deg = in.llen[v];
        if (deg > TT::MAXLEN / 2u) toolong = true;
        T.loff[v] = (typename TT::off_t)lo; T.llen[v] = (typename TT::len_t)deg;
        const int32_t* r = in.nbr + in.loff[v];
        I* d = T.ring + lo;
        for (uint32_t j = 0; j < deg; ++j) d[j] = (I)r[j];
        uint32_t f = SURTR_NEVER;
        for (uint32_t k = 0; k < F; ++k)
        {
            const int c = side_of(plane_dist(sh.planes[k], px, py, pz));
            if (c < 0) { f = k; break; }
            if (c == 0) atomicAdd(&sh.nzero[k], 1u);
        }
        T.fc[v] = (uint8_t)f;
    }
    if (toolong) sh.flagBad = 1;
    __syncthreads();
    if (sh.flagBad != 0) return SURTR_OVERFLOW;
    T.nS = V; T.nLive = V; T.hUsed = H;
    return 0;
}

// Global scratch used to squeeze tombstones out of a Topo (any variant) without in-place hazards.
struct SqueezeTmp
{
    float* pos; uint32_t* loff; uint32_t* llen; int8_t* comp; uint32_t* ring;
};

// Order-preserving removal of the dead slots (fc < T.kcur): live slots move down, rings are renumbered.
// The caller restarts its plane afterwards.
template <class TT>
__device__ void squeeze(Topo<TT>& T, Shared& sh, const SqueezeTmp tmp)
{
    typedef typename TT::idx_t I;
    COUNT(31);
    auto livefn = [&](uint32_t v) -> uint2 {
        return T.alive(v) ? make_uint2(1u, (uint32_t)T.llen[v]) : make_uint2(0u, 0u);
    };
    uint32_t nn = 0, hh = 0;
    scan_blocks(T.nS, T.blk, sh, livefn, nn, hh);
    const uint32_t nb = (T.nS + SURTR_LANES - 1u) >> SURTR_LSH;
    uint32_t* idmap = T.aux0;
    for (uint32_t b = wave_id(); b < nb; b += group_waves())
    {
        const uint32_t v = (b << SURTR_LSH) + lane_id();
        uint2 c = make_uint2(0u, 0u);
        if (v < T.nS) c = livefn(v);
        const uint2 e = wave_excl2(c);
        if (v < T.nS && c.x)
        {
            const uint32_t id = T.blk[b].x + e.x;
            idmap[v] = id;
            tmp.pos[3 * id] = T.pos[3 * v]; tmp.pos[3 * id + 1] = T.pos[3 * v + 1]; tmp.pos[3 * id + 2] = T.pos[3 * v + 2];
            tmp.loff[id] = T.blk[b].y + e.y; tmp.llen[id] = c.y; tmp.comp[id] = (int8_t)T.fc[v];
        }
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < T.nS; v += group_size())
    {
        if (!T.alive(v)) continue;
        const I* r = T.ring + T.loff[v];
        uint32_t* d = tmp.ring + tmp.loff[idmap[v]];
        const uint32_t len = T.llen[v];
        for (uint32_t q = 0; q < len; ++q) { const uint32_t u = r[q]; d[q] = u >= TT::SENT ? u : idmap[u]; }
    }
    __syncthreads();
    for (uint32_t v = threadIdx.x; v < nn; v += group_size())
    {
        T.pos[3 * v] = tmp.pos[3 * v]; T.pos[3 * v + 1] = tmp.pos[3 * v + 1]; T.pos[3 * v + 2] = tmp.pos[3 * v + 2];
        T.loff[v] = (typename TT::off_t)tmp.loff[v]; T.llen[v] = (typename TT::len_t)tmp.llen[v]; T.fc[v] = (uint8_t)tmp.comp[v];
    }
    for (uint32_t e = threadIdx.x; e < hh; e += group_size()) T.ring[e] = (I)tmp.ring[e];
    T.nS = nn; T.hUsed = hh;
    __syncthreads();
}

// ---------------------------------------------------------------------------
// The plane loop on a reduced solid.  `in`/`bmask` are only consulted in the all-in-plane corner case.
// Returns 0 (T.nLive == 0: empty), an error code, or SURTR_OVERFLOW.
// (always inlined: as a call the Topo would live in private memory and every field access would be a scratch load)
// MULTIWAVE: compiled for a workgroup of several waves (a variant of the edge-cut sweep that pays off there only).
template <class TT, bool MULTIWAVE = true>
__device__ __attribute__((always_inline)) inline int clip_planes(Topo<TT>& T, const uint32_t F, Shared& sh, const SolidIn in, const unsigned long long* bmask,
                           const SqueezeTmp tmp)
{
    typedef typename TT::idx_t I;
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id();
    STAMP_DECL;
    if (T.nLive == 0) return 0;
    bool squeezed = false;
    uint32_t it = 0;
    const uint32_t dropTotal = in.nv > T.nS ? in.nv - T.nS : 0u;      // vertices of the input that the band reduction left out
    for (uint32_t q = tid; q < 24; q += group_size()) (&sh.pf[0][0])[q] = 0;
    __syncthreads();
    for (uint32_t k = 0; k < F; ++k)
    {
        const float4 pl = sh.planes[k];
        const uint32_t nS = SURTR_UNIFORM(T.nS);
        // Per-plane flags are triple buffered so that no barrier is spent on resetting them: set (it+2)%3 is cleared
        // right after this plane's first barrier -- every lane has finished reading it (plane it-1) by then, and nobody
        // writes it before the first barrier of plane it+1.  `it` counts loop passes (a squeeze retries a plane).
        uint32_t* pf = sh.pf[it % 3u];
        T.kcur = k; T.n0cur = nS; T.zmode = SURTR_UNIFORM(sh.nzero[k]) != 0u;
        // per-plane work lists (WArr): carved from the top of the LDS ring area downwards while the ring itself grows
        // upwards from hUsed; a list that would not fit goes to global scratch (wtop: first entry in use by a list)
        uint32_t wtop = T.capH;
        auto carve16 = [&](uint32_t n, uint32_t floor_, uint32_t* g) -> WArr<uint16_t> {
            if (sizeof(I) == 2 && wtop >= floor_ + n + 8u) { wtop -= n; return WArr<uint16_t>{(uint16_t*)(void*)(T.ring + wtop), g}; }
            return WArr<uint16_t>{nullptr, g};
        };
        auto carve8 = [&](uint32_t n, uint32_t floor_, uint32_t* g) -> WArr<uint8_t> {
            const uint32_t e = (n + 1u) / 2u;
            if (sizeof(I) == 2 && wtop >= floor_ + e + 8u) { wtop -= e; return WArr<uint8_t>{(uint8_t*)(void*)(T.ring + wtop), g}; }
            return WArr<uint8_t>{nullptr, g};
        };
        if (T.zmode)
        {
            // ---- classify (:307-318), only for a plane some live vertex lies in ----
            COUNT(41);
            uint32_t zlen = 0;
            for (uint32_t v = tid; v < nS; v += group_size())
            {
                int c = SURTR_DEAD;
                if (T.alive(v))
                {
                    c = side_of(plane_dist(pl, T.pos[3 * v], T.pos[3 * v + 1], T.pos[3 * v + 2]));
                    if (c == 0) zlen += 1u + (uint32_t)T.llen[v];
                }
                T.gcomp[v] = (int8_t)c;
            }
            if (zlen) atomicAdd(&pf[2], zlen);      // in-plane vertices: nonzero = any, value = their ring entries (+1 each)
            __syncthreads();
        }
        // ---- the clipped set: fc == k (or comp == -1 of the explicit classification) ----
        const uint8_t* sel_arr = T.zmode ? (const uint8_t*)T.gcomp : T.fc;
        const uint32_t sel_val = T.zmode ? 0xFFu : k;
        STAMP(76);
        const uint32_t nC = select_count(sel_arr, sel_val, nS, sh);
        STAMP(77);
        // the clipped vertices, ascending, and their kept-neighbour counts (the ring is not growing yet: floor = hUsed)
        WArr<uint16_t> clist = carve16(nC, T.hUsed, T.aux3);
        WArr<uint16_t> cutcnt = carve16(nC, T.hUsed, T.aux2);      // per clipped vertex: bit j = ring slot j holds a kept neighbour (rings of
                                                                   // 16 and more entries: 0x8000 | their number)
        if (nC) select_write(sel_arr, sel_val, nS, sh, clist); else __syncthreads();
        STAMP(78);
        if (T.zmode)
        {
            // kept vertices, counted (fast mode derives them from the live count)
            uint32_t kept = 0;
            for (uint32_t v = tid; v < nS; v += group_size()) if (T.gcomp[v] > 0) ++kept;
            if (kept) atomicAdd(&pf[1], kept);
            __syncthreads();
        }
        if (tid == 0) { uint32_t* nx = sh.pf[(it + 2u) % 3u]; for (int q = 0; q < 8; ++q) nx[q] = 0; }
        ++it;
        STAMP(4);
        const bool anyCut = nC != 0, anyZero = pf[2] != 0;
        const bool anyKeep = T.zmode ? pf[1] != 0 : T.nLive > nC;
        const uint32_t dropAlive = sh.hist[k];
        const uint32_t dropKept = dropAlive - sh.zhist[k];   // dropped vertices strictly on the kept side of this plane
        if (!anyCut && !anyKeep && dropKept == 0)
        {
            // Every vertex is in-plane (e.g. a zero plane from a degenerate hull face).  The reference
            // consults the bounding box first (:296-299): all corners >= 0 skips the plane, anything
            // else ends in "below" (:322-327).
            if (tid == 0)
            {
                float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};
                bool first = true;
                auto grow = [&](const float* p) {
                    for (int a = 0; a < 3; ++a)
                    {
                        if (first) { lo[a] = p[a]; hi[a] = p[a]; }
                        else { lo[a] = p[a] < lo[a] ? p[a] : lo[a]; hi[a] = p[a] > hi[a] ? p[a] : hi[a]; }
                    }
                    first = false;
                };
                for (uint32_t v = 0; v < nS; ++v) if (T.alive(v)) grow(T.pos + 3 * v);
                if (dropAlive != 0)
                {
                    // dropped vertices still alive (all in-plane here) belong to the box too
                    for (uint32_t v = 0; v < in.nv; ++v)
                    {
                        if ((bmask[v >> SURTR_LSH] >> (v & (SURTR_LANES - 1u))) & 1ull) continue;
                        bool cutBefore = false;
                        for (uint32_t q = 0; q <= k && !cutBefore; ++q)
                            if (side_of(plane_dist(sh.planes[q], in.pos[3 * v], in.pos[3 * v + 1], in.pos[3 * v + 2])) < 0) cutBefore = true;
                        if (!cutBefore) grow(in.pos + 3 * v);
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
            T.nLive = 0; break;
        }
        if (!anyKeep && dropKept == 0) { T.nLive = 0; break; }     // "below": everything goes (:322-327)
        if (!anyCut)
        {
            // "above" for the reduced solid; dropped vertices may still vanish here (size check :497-499)
            if (T.nLive + dropAlive < 4u) { T.nLive = 0; break; }
            continue;
        }

        // ---- new vertices on straddling edges, in (vertex, slot) order (:333-357) ----
        auto cutfn = [&](uint32_t i) -> uint2 {
            const uint32_t v = clist.get(i);
            uint32_t c = 0, mask = 0;
            const I* r = T.ring + T.loff[v];
            const uint32_t deg = T.llen[v];
            for (uint32_t j0 = 0; j0 < deg; j0 += 4)
            {
                // four entries per round trip: ring entries first, then their state (indices clamped into the ring)
                uint32_t u[4]; int cu[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] = r[j0 + q < deg ? j0 + q : deg - 1u];
#pragma unroll
                for (int q = 0; q < 4; ++q) cu[q] = T.cmp(u[q] < TT::SENT ? u[q] : 0u);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < deg && u[q] < TT::SENT && cu[q] > 0) { ++c; if (j0 + q < 15u) mask |= 1u << (j0 + q); }
            }
            cutcnt.set(i, deg < 16u ? mask : (0x8000u | c));
            return make_uint2(c, 1u);
        };
        // every wave counts a contiguous range of 64-vertex blocks of the clipped list: one barrier gives M and, in the
        // sweep below, the wave's base
        const uint32_t cnb = (nC + SURTR_LANES - 1u) >> SURTR_LSH, cnbw = (cnb + group_waves() - 1u) / group_waves();
        const uint32_t cb0 = w * cnbw, cb1 = cb0 + cnbw < cnb ? cb0 + cnbw : cnb;
        uint32_t M = 0;
        {
            uint32_t mine = 0;
            for (uint32_t b = cb0; b < cb1; ++b) { const uint32_t i = (b << SURTR_LSH) + l; if (i < nC) mine += cutfn(i).x; }
            const uint2 inc = wave_incl_scan2(make_uint2(mine, 0u));
            if (l == SURTR_LANES - 1u) sh.wsum[w] = inc.x;
            __syncthreads();
            for (uint32_t q = 0; q < group_waves(); ++q) M += sh.wsum[q];
        }
        STAMP(8);
        // An in-plane plane takes the serial relink, which re-homes the rings of in-plane vertices (room for the walks
        // that arrive + the old_neighbors snapshot): about 4 entries per ring entry of an in-plane vertex.  New vertices
        // reached by several walks (degenerate rings) need room too; the exact demand is checked once it is known
        // (SURTR_OVERFLOW below), this estimate only decides whether to squeeze first.
        const uint32_t zroom = anyZero ? 4u * pf[2] + 64u : 0u;
        if (nS + M > T.capV || T.hUsed + 3u * M + zroom > T.capH || nS + M >= TT::SENT)
        {
            // out of slots: squeeze the tombstones out (order-preserving, like :464-495) and retry this plane once
            if (squeezed)
            {
                COUNT(36);
#ifdef SURTR_STAMP
                if (tid == 0) printf("overflow after squeeze: plane %u of %u, nS %u + M %u (capV %u), hUsed %u + 3M (capH %u), nLive %u\n", k, F, nS, M, T.capV, T.hUsed, T.capH, T.nLive);
#endif
                return SURTR_OVERFLOW;
            }
            squeeze(T, sh, tmp);
            squeezed = true;
            --k;
            continue;
        }
        squeezed = false;
        const uint32_t n0 = nS, n1 = nS + M;
        // The reference bounds every relink walk by the number of vertices its (compacted) solid has at this point (:389-394):
        // live vertices, the dropped ones among them, plus this plane's new ones -- fewer than the n1 slots here, which still
        // hold the vertices clipped by earlier planes.  On a regular solid no walk comes near either bound; on a sliver whose
        // walk wanders among clipped vertices the place where it stops is the reference's result.
        const uint32_t nref = T.nLive + (k ? sh.hist[k - 1u] : dropTotal) + M;
        if ((clist.l || cutcnt.l) && T.hUsed + 3u * M + 8u > wtop)
        {
            // the new rings would run into the lists: move them out (a band that nearly fills the ring area)
            for (uint32_t i = tid; i < nC; i += group_size()) { if (clist.l) T.aux3[i] = clist.l[i]; if (cutcnt.l) T.aux2[i] = cutcnt.l[i]; }
            __syncthreads();
            clist.l = nullptr; cutcnt.l = nullptr; wtop = T.capH;
        }
        // per new vertex: source (clipped vertex, slot), later its successor (same array); predecessor; state of a paused walk
        const uint32_t wfloor = T.hUsed + 3u * M;
        const WArr<uint16_t> srcv = carve16(M, wfloor, T.succ);
        const WArr<uint8_t> srcj = carve8(M, wfloor, T.aux1);
        const WArr<uint16_t> wpred = carve16(M, wfloor, T.pred);
        const WArr<uint16_t> wsave = carve16(M, wfloor, T.aux0);
        {
            // which (clipped vertex, slot) makes new vertex n0 + t, in reference order
            bool dup = false;
            uint32_t run = 0;
            for (uint32_t q = 0; q < w; ++q) run += sh.wsum[q];
            for (uint32_t b = cb0; b < cb1; ++b)
            {
                const uint32_t i = (b << SURTR_LSH) + l;
                uint2 c = make_uint2(0u, 0u);
                uint32_t km = 0;
                if (i < nC) { km = cutcnt.get(i); c.x = (km & 0x8000u) ? (km & 0x7FFFu) : (uint32_t)__builtin_popcount(km); }
                const uint2 s2 = wave_incl_scan2(c);
                const uint32_t base = run;
                run += lane_bcast(s2.x, SURTR_LANES - 1u);
                if (i < nC && c.x)
                {
                    const uint32_t v = clist.get(i);
                    uint32_t t = base + s2.x - c.x;
                    if (MULTIWAVE && !(km & 0x8000u))
                    {
                        // the count left the kept slots as a bit mask: no ring is read unless two of them could hold the same vertex
                        if (c.x >= 2u)
                        {
                            const I* r2 = T.ring + T.loff[v];
                            for (uint32_t mm = km; mm; mm &= mm - 1u)
                            {
                                const uint32_t ja = (uint32_t)__builtin_ctz(mm), ua = r2[ja];
                                for (uint32_t lo = km & ((1u << ja) - 1u); lo; lo &= lo - 1u) if ((uint32_t)r2[__builtin_ctz(lo)] == ua) dup = true;
                            }
                        }
                        for (uint32_t mm = km; mm; mm &= mm - 1u) { srcv.set(t, v); srcj.set(t, (uint32_t)__builtin_ctz(mm)); ++t; }
                        continue;
                    }
                    const I* r = T.ring + T.loff[v];
                    const uint32_t deg = T.llen[v];
                    for (uint32_t j0 = 0; j0 < deg; j0 += 4)
                    {
                        // four entries per round trip, as in the count (indices clamped into the ring)
                        uint32_t u[4]; int cu[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = r[j0 + q < deg ? j0 + q : deg - 1u];
#pragma unroll
                        for (int q = 0; q < 4; ++q) cu[q] = T.cmp(u[q] < TT::SENT ? u[q] : 0u);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                        {
                            const uint32_t j = j0 + q;
                            if (j >= deg || u[q] >= TT::SENT || cu[q] <= 0) continue;
                            srcv.set(t, v); srcj.set(t, j); ++t;
                            // a ring that lists the same kept neighbour twice (sliver input) makes the back-link patch order dependent
                            for (uint32_t jj = 0; jj < j; ++jj) if ((uint32_t)r[jj] == u[q]) dup = true;
                        }
                    }
                }
            }
            if (dup) pf[3] = 1;
            __syncthreads();
            STAMP(9);
            const bool ordered = pf[3] != 0;
            // dense pass: one lane per new vertex (two position gathers each, all lanes busy); unless the order matters,
            // the lane also patches the links of the two ends (:350-354): "find the clipped vertex in the kept vertex's
            // ring, overwrite it" -- concurrent patches touch different entries, and an entry already patched holds a new
            // vertex, which no search looks for
            for (uint32_t t = tid; t < M; t += group_size())
            {
                const uint32_t v = srcv.get(t), j = srcj.get(t), fresh = n0 + t;
                I* r = T.ring + T.loff[v];
                const uint32_t u = r[j];
                const float ax = T.pos[3 * v], ay = T.pos[3 * v + 1], az = T.pos[3 * v + 2];
                const float bx = T.pos[3 * u], by = T.pos[3 * u + 1], bz = T.pos[3 * u + 2];
                const float sa = plane_dist(pl, ax, ay, az);
                const float sb = plane_dist(pl, bx, by, bz);
                // PlaneLineIntersection (:746-751): (a*sb - b*sa) * (1/(sb-sa))
                const float inv = 1.f / (sb - sa);
                T.pos[3 * fresh] = (ax * sb - bx * sa) * inv;
                T.pos[3 * fresh + 1] = (ay * sb - by * sa) * inv;
                T.pos[3 * fresh + 2] = (az * sb - bz * sa) * inv;
                {
                    // the first later plane that clips the new vertex, and those it lies in before that
                    const float nx = T.pos[3 * fresh], ny = T.pos[3 * fresh + 1], nz = T.pos[3 * fresh + 2];
                    uint32_t f = SURTR_NEVER;
                    for (uint32_t q = k + 1u; q < F; ++q)
                    {
                        const int cq = side_of(plane_dist(sh.planes[q], nx, ny, nz));
                        if (cq < 0) { f = q; break; }
                        if (cq == 0) atomicAdd(&sh.nzero[q], 1u);
                    }
                    T.fc[fresh] = (uint8_t)f;
                    if (T.zmode) T.gcomp[fresh] = 2;
                }
                const uint32_t lo = T.hUsed + 3u * t;
                T.loff[fresh] = (typename TT::off_t)lo; T.llen[fresh] = 2;
                T.ring[lo] = (I)v; T.ring[lo + 1] = (I)u; T.ring[lo + 2] = (I)TT::REM;
                wpred.set(t, 0u);          // no predecessor yet (0 is no new vertex: n0 >= 4)
                if (!ordered)
                {
                    I* ru = T.ring + T.loff[u];
                    const uint32_t du = T.llen[u];
                    for (uint32_t q = 0; q < du; ++q) if ((uint32_t)ru[q] == v) { ru[q] = (I)fresh; break; }
                    r[j] = (I)fresh;
                }
            }
            STAMP(10);
            if (ordered) __syncthreads();
            if (ordered && tid == 0)
            {
                COUNT(39);
                for (uint32_t t = 0; t < M; ++t)        // reference order: first remaining occurrence each time
                {
                    const uint32_t v = srcv.get(t), fresh = n0 + t, u = T.ring[T.loff[fresh] + 1u];
                    I* ru = T.ring + T.loff[u];
                    const uint32_t du = T.llen[u];
                    for (uint32_t q = 0; q < du; ++q) if ((uint32_t)ru[q] == v) { ru[q] = (I)fresh; break; }
                    T.ring[T.loff[v] + srcj.get(t)] = (I)fresh;
                }
            }
        }
        uint32_t hend = T.hUsed + 3u * M;
        if (tid == 0) sh.misc[6] = 0;      // length of the relink's chain list
        __syncthreads();
        STAMP(5);

        // ---- relink (:367-431) ----
        bool serial = anyZero;
        if (!serial)
        {
            // regular cap: every new vertex X=[cut, kept] finds its successor by walking the face
            // loop through clipped vertices; its final ring is [pred, succ, kept].
            //
            // Most walks take 2-3 steps, but a plane that trims the cap of an earlier plane walks along that cap's
            // boundary: one lane chases ~100 dependent ring reads while the workgroup waits.  So walks stop after
            // SURTR_WALK0 steps; if any is unfinished, the runs of clipped cap vertices are collapsed by pointer
            // jumping first.  A cap vertex c=[pred, succ, kept] entered from succ is left towards pred (the entry
            // before succ), so along a run jump[c] -> pred while pred is again a clipped 3-ring entered from its succ
            // slot; after the rounds jump[c] is a vertex further down the same run (the walk would have reached it,
            // arriving from its succ slot), and the resumed walk takes the shortcut.  Results are those of the
            // plain walk.
            bool bad = false, longw = false;
            const WArr<uint16_t> wsucc = srcv;      // the sources are not needed any more
            auto finish = [&](uint32_t t, uint32_t X, uint32_t c) {
                if (c >= TT::SENT || c < n0 || c == X || T.cmp(c) != 2) { bad = true; wsucc.set(t, X); }
                else
                {
                    // successor of X, X as predecessor of c: if two walks end on c one of them will not find itself there
                    wsucc.set(t, c);
                    wpred.set(c - n0, X);
                }
            };
            for (uint32_t t = tid; t < M; t += group_size())
            {
                const uint32_t X = n0 + t;
                uint32_t prev = X, c = T.ring[T.loff[X]], steps = 0;
                bool clipped;
                while (true)
                {
                    // state, ring offset and length of c in one LDS round trip (index clamped: c may be a sentinel)
                    const uint32_t cc = c < TT::SENT ? c : 0u;
                    const int cm = T.cmp(cc); const uint32_t lo = T.loff[cc], len = T.llen[cc];
                    clipped = c < TT::SENT && cm == -1;
                    if (!clipped || steps >= SURTR_WALK0) break;
                    const uint32_t hold = c;
                    c = face_next(T.ring + lo, len, prev);
                    prev = hold; ++steps;
                }
                if (clipped) { wsave.set(t, prev + 1u); wsucc.set(t, c); longw = true; continue; }      // paused: last hop + 1
                wsave.set(t, 0u);
                finish(t, X, c);
            }
            if (longw) pf[7] = 1;
            if (bad) pf[4] = 1;
            __syncthreads();
            STAMP(12);
            COUNT(43);
            if (pf[7] != 0)
            {
                COUNT(40);
                const uint32_t jfloor = T.hUsed + 3u * M;
                const WArr<uint16_t> chain = carve16(nC, jfloor, T.aux2);
                const WArr<uint16_t> jump = carve16(n0, jfloor, T.pcnt);
                for (uint32_t i = tid; i < nC; i += group_size())
                {
                    const uint32_t v = clist.get(i);
                    if (T.llen[v] != 3u) continue;
                    const uint32_t p = T.ring[T.loff[v]];
                    uint32_t j = v;
                    if (p < TT::SENT && T.cmp(p) == -1 && T.llen[p] == 3u)
                    {
                        const I* rp = T.ring + T.loff[p];
                        if ((uint32_t)rp[1] == v && (uint32_t)rp[0] != v) j = p;     // entered from its succ slot only
                    }
                    jump.set(v, j);
                    if (j != v) chain.set(atomicAdd(&sh.misc[6], 1u), v);
                }
                __syncthreads();
                const uint32_t L = sh.misc[6];
                for (uint32_t round = 0; round < 6u && (1u << round) < L; ++round)
                {
                    // in place: whatever value a lane reads is a vertex further down the same run
                    for (uint32_t i = tid; i < L; i += group_size())
                    {
                        const uint32_t v = chain.get(i), j = jump.get(v), jj = jump.get(j);
                        if (jj != j) jump.set(v, jj);
                    }
                    __syncthreads();
                }
                STAMP(11);
                for (uint32_t t = tid; t < M; t += group_size())
                {
                    const uint32_t w0 = wsave.get(t);
                    if (w0 == 0u) continue;
                    const uint32_t X = n0 + t;
                    uint32_t prev = w0 - 1u, c = wsucc.get(t), steps = SURTR_WALK0;
                    while (true)
                    {
                        const uint32_t cc = c < TT::SENT ? c : 0u;
                        const int cm = T.cmp(cc); const uint32_t lo = T.loff[cc], len = T.llen[cc];
                        if (!(c < TT::SENT && cm == -1 && steps++ < nref)) break;
                        const I* r = T.ring + lo;
                        if (len == 3u && (uint32_t)r[1] == prev && (uint32_t)r[0] != prev)
                        {
                            const uint32_t e = jump.get(c);
                            if (e != c) { c = e; prev = T.ring[T.loff[e] + 1u]; continue; }
                        }
                        const uint32_t hold = c;
                        c = face_next(r, len, prev);
                        prev = hold;
                    }
                    finish(t, X, c);
                }
                if (bad) pf[4] = 1;
                __syncthreads();
                STAMP(14);
            }
            bad = false;
            // every new vertex is the successor of exactly one other: each walk must find itself as the predecessor of its end
            for (uint32_t t = tid; t < M; t += group_size())
            {
                const uint32_t X = n0 + t, c = wsucc.get(t);
                if (c == X || c < n0 || wpred.get(c - n0) != X || wpred.get(t) == 0u) bad = true;
            }
            if (bad) pf[4] = 1;
            __syncthreads();
            STAMP(13);
            serial = pf[4] != 0;
            if (!serial)
            {
                for (uint32_t t = tid; t < M; t += group_size())
                {
                    const uint32_t lo = T.loff[n0 + t];
                    const I kept = T.ring[lo + 1];
                    T.ring[lo] = (I)wpred.get(t); T.ring[lo + 1] = (I)wsucc.get(t); T.ring[lo + 2] = kept;
                    T.llen[n0 + t] = 3;
                }
            }
        }
        if (serial)
        {
            COUNT(19);
            // Dry run (parallel): how many walks end on each surviving vertex = an upper bound of the insertions it
            // receives (:404-421).  Walks only traverse rings of clipped vertices, which the relink never modifies,
            // so their targets do not depend on the processing order.
            uint32_t* snapoff = T.aux0; uint32_t* cap = T.aux1; uint32_t* zlist = T.aux2; uint32_t* arrive = T.pcnt;
            uint32_t* wcur = T.succ; uint32_t* wprev = T.pred;      // per new vertex: end and last hop of its walk
            uint32_t* srcof = T.aux0;                               // per new vertex: who walked to it (snapoff is only used below n0)
            uint32_t* slist = T.aux3;                               // new vertices the serial relink has to visit (clist is done with)
            for (uint32_t v = tid; v < n1; v += group_size()) { arrive[v] = 0; if (v >= n0) { wcur[v - n0] = TT::SENT; wprev[v - n0] = v; srcof[v] = 0; } }
            __syncthreads();
            for (uint32_t v = tid; v < n1; v += group_size())
            {
                const int cv = T.cmp(v);
                if (!(cv == 0 || cv == 2)) continue;
                const I* r = T.ring + T.loff[v];
                const uint32_t deg = T.llen[v];
                for (uint32_t j = 0; j < deg; ++j)
                {
                    const uint32_t jn = r[j];
                    if (jn >= TT::SENT || T.cmp(jn) != -1) continue;
                    uint32_t prev = v, c = jn, steps = 0;
                    while (c < TT::SENT && T.cmp(c) == -1 && steps++ < nref)
                    {
                        const uint32_t hold = c;
                        c = face_next(T.ring + T.loff[c], T.llen[c], prev);
                        prev = hold;
                    }
                    if (c < TT::SENT && c != v) { atomicAdd(&arrive[c], 1u); if (c >= n0) srcof[c] = v; }
                    if (v >= n0) { wcur[v - n0] = c; wprev[v - n0] = prev; }     // a new vertex has one clipped neighbour
                }
            }
            __syncthreads();
            // every in-plane vertex gets a ring with room for its arrivals plus the old_neighbors snapshot (:367-369);
            // a new vertex keeps its 3 slots unless more than one walk ends on it (degenerate input rings)
            auto zfn = [&](uint32_t v) -> uint2 {
                const int cv = T.cmp(v);
                if (cv == 0) return make_uint2(v < n0 ? 1u : 0u, 2u * ((uint32_t)T.llen[v] + arrive[v]));
                if (cv == 2 && arrive[v] > 1u) return make_uint2(0u, (uint32_t)T.llen[v] + arrive[v]);
                return make_uint2(0u, 0u);
            };
            uint32_t zc = 0, zw = 0;
            scan_blocks(n1, T.blk, sh, zfn, zc, zw);
            if (hend + zw > T.capH) { COUNT(37); return SURTR_OVERFLOW; }
            const uint32_t nb = (n1 + SURTR_LANES - 1u) >> SURTR_LSH;
            bool toolong = false;
            for (uint32_t b = w; b < nb; b += group_waves())
            {
                const uint32_t v = (b << SURTR_LSH) + l;
                uint2 c = make_uint2(0u, 0u);
                if (v < n1) c = zfn(v);
                const uint2 e = wave_excl2(c);
                if (v < n1)
                {
                    const int cv = T.cmp(v);
                    const uint32_t len = T.llen[v];
                    if (c.y)
                    {
                        const uint32_t room = len + arrive[v];
                        const uint32_t dst = hend + T.blk[b].y + e.y;
                        const I* src = T.ring + T.loff[v];
                        for (uint32_t q = 0; q < len; ++q) { T.ring[dst + q] = src[q]; if (cv == 0) T.ring[dst + room + q] = src[q]; }
                        T.loff[v] = (typename TT::off_t)dst; if (cv == 0) snapoff[v] = dst + room;
                        cap[v] = room;
                        if (room > TT::MAXLEN) toolong = true;
                    }
                    else if (cv == 2) cap[v] = 3u;
                    if (c.x) zlist[T.blk[b].x + e.x] = v;
                }
            }
            if (toolong) pf[6] = 1;
            hend += zw;
            __syncthreads();
            if (pf[6]) { COUNT(38); return SURTR_OVERFLOW; }     // a ring could outgrow this Topo's length type
            // the order-dependent part: new vertices that are not "easy" (see relink_serial) or walk to one that is not
            auto easy = [&](uint32_t c) -> bool {
                return c >= n0 && c < n1 && arrive[c] == 1u && srcof[c] >= n0 && wcur[c - n0] < TT::SENT && wcur[c - n0] != c;
            };
            auto sfn = [&](uint32_t t) -> uint2 {
                const uint32_t X = n0 + t;
                return make_uint2((!easy(X) || !easy(wcur[t])) ? 1u : 0u, 0u);
            };
            uint32_t ns = 0, sdum = 0;
            scan_blocks(M, T.blk, sh, sfn, ns, sdum);
            {
                const uint32_t nbm = (M + SURTR_LANES - 1u) >> SURTR_LSH;
                for (uint32_t b = w; b < nbm; b += group_waves())
                {
                    const uint32_t t = (b << SURTR_LSH) + l;
                    uint2 c = make_uint2(0u, 0u);
                    if (t < M) c = sfn(t);
                    const uint2 e = wave_excl2(c);
                    if (t < M && c.x) slist[T.blk[b].x + e.x] = n0 + t;
                }
            }
            __syncthreads();
            if (tid == 0) relink_serial(T, n0, n1, nref, snapoff, cap, zlist, zc, slist, ns, wcur, wprev, arrive, srcof, sh);
            __syncthreads();
            if (sh.flagErr) return 2;
            // easy vertices: [source, target, kept]
            for (uint32_t t = tid; t < M; t += group_size())
            {
                const uint32_t X = n0 + t;
                if (!easy(X)) continue;
                const uint32_t lo = T.loff[X];
                const I kept = T.ring[lo + 1];
                T.ring[lo] = (I)srcof[X]; T.ring[lo + 1] = (I)wcur[t]; T.ring[lo + 2] = kept;
                T.llen[X] = 3;
            }
            __syncthreads();
            // drop the REM marks (:426-431), one vertex per lane; then look for two-neighbour vertices
            if (tid == 0) sh.changed = 0;
            __syncthreads();
            {
                bool two = false;
                for (uint32_t v = tid; v < n1; v += group_size())
                {
                    if (T.cmp(v) == SURTR_DEAD) continue;
                    I* r = T.ring + T.loff[v];
                    const uint32_t len = T.llen[v];
                    uint32_t wq = 0;
                    for (uint32_t q = 0; q < len; ++q) { const I u = r[q]; if ((uint32_t)u != TT::REM) r[wq++] = u; }
                    T.llen[v] = (typename TT::len_t)wq;
                    if (T.cmp(v) >= 0 && wq == 2u) two = true;
                }
                if (two) sh.changed = 1;
            }
            __syncthreads();
            if (sh.changed)
            {
                if (tid == 0) collapse_serial(T, n1);
                __syncthreads();
            }
        }
        __syncthreads();     // rings and lengths of this plane are final
        STAMP(6);
        // ---- no compaction (:464-495): a clipped vertex is dead from the next plane on (fc < k); live count for :497-499 ----
        uint32_t nLiveNew = T.nLive - nC + M;
        if (serial || T.zmode)
        {
            // the general relink may also have removed vertices: count, and check that no live vertex links a dead one
            uint32_t live = 0; bool dangling = false;
            for (uint32_t v = tid; v < n1; v += group_size())
            {
                const int c = T.cmp(v);
                if (c == SURTR_DEAD) continue;
                if (c < 0) { T.fc[v] = (uint8_t)k; continue; }
                ++live;
                const I* r = T.ring + T.loff[v];
                const uint32_t len = T.llen[v];
                for (uint32_t q = 0; q < len; ++q)
                {
                    const uint32_t u = r[q];
                    if (u < TT::SENT && T.cmp(u) < 0) dangling = true;   // a live vertex still links a clipped one
                }
            }
            if (dangling) { SURTR_DBG("plane %u: live vertex links a clipped one\n", k); sh.flagErr = 1; }
            if (live) atomicAdd(&pf[5], live);
            __syncthreads();
            if (sh.flagErr) return 2;
            nLiveNew = pf[5];
        }
        T.nS = n1; T.hUsed = hend; T.nLive = nLiveNew;
        STAMP(7);
        if (T.nLive + dropAlive < 4u) { T.nLive = 0; break; }
    }
    T.kcur = F; T.n0cur = T.nS; T.zmode = false;      // alive(v) from here on: no plane clipped v
    return 0;
}

// Assigns the final (compacted) index of every live slot: idmap = T.aux0, ring offset in the packed
// ring array = T.aux2.  Returns (vertices, ring entries).
template <class TT>
__device__ __attribute__((always_inline)) inline uint2 index_live(Topo<TT>& T, Shared& sh)
{
    auto livefn = [&](uint32_t v) -> uint2 {
        return T.alive(v) ? make_uint2(1u, (uint32_t)T.llen[v]) : make_uint2(0u, 0u);
    };
    uint32_t nn = 0, hh = 0;
    scan_blocks(T.nS, T.blk, sh, livefn, nn, hh);
    const uint32_t nb = (T.nS + SURTR_LANES - 1u) >> SURTR_LSH;
    for (uint32_t b = wave_id(); b < nb; b += group_waves())
    {
        const uint32_t v = (b << SURTR_LSH) + lane_id();
        uint2 c = make_uint2(0u, 0u);
        if (v < T.nS) c = livefn(v);
        const uint2 e = wave_excl2(c);
        if (v < T.nS && c.x) { T.aux0[v] = T.blk[b].x + e.x; T.aux2[v] = T.blk[b].y + e.y; }
    }
    __syncthreads();
    return make_uint2(nn, hh);
}

// Writes the live part of T as a packed solid: positions, absolute ring offsets (hoff + ...), lengths, rings.
template <class TT>
__device__ __attribute__((always_inline)) inline void write_solid(const Topo<TT>& T, float* dpos, uint32_t* dloff, uint32_t* dllen, int32_t* dnbr, uint32_t voff,
                            uint32_t hoff)
{
    for (uint32_t v = threadIdx.x; v < T.nS; v += group_size())
    {
        if (!T.alive(v)) continue;
        const uint32_t id = voff + T.aux0[v];
        dpos[3 * (size_t)id] = T.pos[3 * v]; dpos[3 * (size_t)id + 1] = T.pos[3 * v + 1]; dpos[3 * (size_t)id + 2] = T.pos[3 * v + 2];
        const uint32_t lo = hoff + T.aux2[v], len = T.llen[v];
        dloff[id] = lo; dllen[id] = len;
        const typename TT::idx_t* r = T.ring + T.loff[v];
        for (uint32_t q = 0; q < len; ++q) dnbr[lo + q] = (int32_t)T.aux0[(uint32_t)r[q]];
    }
}

} // namespace surtr
