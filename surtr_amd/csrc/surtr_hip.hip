// surtr_hip.hip -- kernels of the fracture event + the C ABI of include/surtr_hip.h.
//
// Event pipeline (on the caller's HIP stream plus two internal streams for the kernels that may overlap -- k_clip_pairs and
// k_clip_pairs_half beside k_clip_pairs_big, k_refit beside k_faces -- fenced by events; inputs resident in HBM):
//   k_place_cells   A3   Polygon3D::Scale/Translate + ConstructFacePlane      (Src/VMACH.cpp:302-310, 506-534)
//   k_clip_convex   A7   Convex of every (cell, piece) pair, one wave per task (Src/Surtr.cpp:1466-1468)
//   k_prep_pairs    A7   pre-pass of the Mesh of every pair whose Convex survived: the vertices the planes can touch,
//                        left in HBM in the layout of the LDS topology
//   k_clip_pairs(_big, _half)  A7+A8+A11  Mesh of those pairs: plane loop on the reduced solid, label islands,
//                        park the result in the arena                          (Src/Surtr.cpp:1470-1504, Src/Poly.cpp:265-500);
//                        three LDS topology sizes (regular, double for large bands, half for light pairs of small pieces)
//   k_frag_table    A11  cell-major fragment table                             (Src/Surtr.cpp:2133-2146)
//   k_refit         A12  limit-4 hull normals + k-DOP slabs + clip Convex      (Src/Surtr.cpp:1449-1455)
//   k_faces         A9+A10  ExtractFaces + EarClipping of every Mesh           (Src/Poly.cpp:89-126, 764-913)
//   k_out_scan / k_pack  coalesced write of the packed fragment blob
// There is no CPU fallback: without a HIP device surtr_create fails with SURTR_E_NOGPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "surtr_ctx.h"
#include "small_clip.h"
#ifndef SC_RESUME
#define SC_RESUME 1      // the general one-wave clipper goes on from the plane small_clip stopped at (0: from the input, as in round 3)
#endif
#include "literal_clip.h"
#include "wave_clip.h"
#include "prep_sorted.h"

// LDS-resident topology of one workgroup (Topo<InLds>) + the dispatcher that falls back to global scratch.
template <uint32_t LV, uint32_t LH>
struct alignas(16) LdsTopoT
{
    static_assert(LV % 16 == 0 && LH % 8 == 0, "clip_image copies 16-byte words into these arrays");
    static constexpr uint32_t kLV = LV, kLH = LH;
    alignas(16) uint16_t loff[LV];
    alignas(16) uint8_t llen[LV];
    alignas(16) uint8_t fc[LV];
    alignas(16) uint2 blk[LV / SURTR_LANES + 2];
    alignas(16) uint16_t ring[LH];
};
typedef LdsTopoT<SURTR_LV, SURTR_LH> LdsTopo;          // Mesh solids: two workgroups of 256 threads per CU
typedef LdsTopoT<2 * SURTR_LV, 2 * SURTR_LH> LdsTopoBig;   // the few Mesh solids with a large band: one workgroup per CU
// Half-size topology for the light pairs: four workgroups of 128 threads per CU (LdsTopoHalf + Shared <= 40 KiB).  These
// kernels are bound by dependent round trips, not by lanes: half the lanes cost a pair ~1.3x, twice the pairs in flight win.
#define SURTR_WGS (SURTR_WG / 2u > SURTR_LANES ? SURTR_WG / 2u : SURTR_LANES)
typedef LdsTopoT<SURTR_LVS, SURTR_LHS> LdsTopoHalf;
// does a solid of n vertices / h ring entries leave a topology of capV / capH room to grow by the cuts?
__host__ __device__ static inline bool fits_with_room(uint32_t n, uint32_t h, uint32_t capV, uint32_t capH)
{
    return n + capV / 5u <= capV && h + capH / 6u <= capH;
}
#ifndef SURTR_SMALL_LV
#define SURTR_SMALL_LV 256
#define SURTR_SMALL_LH 2048
#endif
typedef LdsTopoT<SURTR_SMALL_LV, SURTR_SMALL_LH> LdsTopoSmall;      // Convex solids: one wave per task, many tasks per CU

// Work arrays of a Topo in LDS.  Convex solids have a few dozen vertices: with positions and work lists next to the
// topology no phase of their plane loop waits for HBM.  Used when the input solid has at most N vertices.
template <uint32_t N>
struct LdsWorkT
{
    static constexpr uint32_t kN = N;
    float pos[3 * N];
    uint32_t aux0[N], aux1[N], aux2[N];      // the lists every solid uses; the rarely used ones stay in global scratch
    int8_t gcomp[N];
};
typedef LdsWorkT<LdsTopoSmall::kLV> LdsWorkSmall;
// LDS of a one-wave kernel: the regular small-solid clipper (small_clip.h) and the general one-wave clipper take turns on the
// same bytes (a task is tried by the first and, when it is not regular, redone from its input by the second).
struct OneWaveGeneral { LdsTopoSmall L; LdsWorkSmall W; };
union alignas(16) OneWaveLds { OneWaveGeneral g; ScLds f; };
struct NoLdsWork { static constexpr uint32_t kN = 0; };

template <class LT>
__device__ __attribute__((always_inline)) static inline Topo<InLds> lds_topo(Scratch& S, LT& L)
{
    Topo<InLds> T;
    T.loff = L.loff; T.llen = L.llen; T.fc = L.fc; T.gcomp = S.g_gcomp; T.ring = L.ring; T.pos = S.pos;
    T.succ = S.g_succ; T.pred = S.g_pred; T.pcnt = S.g_pcnt; T.aux0 = S.aux0; T.aux1 = S.aux1; T.aux2 = S.aux2; T.aux3 = S.aux3; T.blk = L.blk;
    T.capV = LT::kLV < S.CV ? LT::kLV : S.CV; T.capH = LT::kLH;
    T.nS = T.nLive = T.hUsed = 0; T.kcur = T.n0cur = 0; T.zmode = false;
    return T;
}
// the same with the work arrays in LDS (W).  Unconditional for a kernel that has them: a pointer that is LDS or global
// depending on the input would make every access through it a flat_load / flat_store (clip_any sends larger solids to
// clip_global instead).
template <class LT, class LW>
__device__ __attribute__((always_inline)) static inline Topo<InLds> lds_topo(Scratch& S, LT& L, LW* W)
{
    Topo<InLds> T = lds_topo(S, L);
    if constexpr (LW::kN != 0)
    {
        T.pos = W->pos; T.aux0 = W->aux0; T.aux1 = W->aux1; T.aux2 = W->aux2; T.gcomp = W->gcomp;
        if (T.capV > LW::kN) T.capV = LW::kN;
    }
    return T;
}

// The wide variant on global scratch: pre-pass + plane loop.  Returns 0 or an error code.
template <class Consume>
__device__ static int clip_global(const SolidIn in, uint32_t F, Scratch& S, Shared& sh, Consume consume)
{
    COUNT(20);
    Topo<InGlobal> T;
    T.loff = S.g_loff; T.llen = S.g_llen; T.fc = (uint8_t*)S.g_comp; T.gcomp = S.g_gcomp; T.ring = S.g_ring; T.pos = S.pos;
    T.succ = S.g_succ; T.pred = S.g_pred; T.pcnt = S.g_pcnt; T.aux0 = S.aux0; T.aux1 = S.aux1; T.aux2 = S.aux2; T.aux3 = S.aux3; T.blk = S.blk;
    T.capV = S.CV; T.capH = S.CH;
    T.nS = T.nLive = T.hUsed = 0; T.kcur = T.n0cur = 0; T.zmode = false;
    int rc = prepass(in, F, T, sh, S.gmask, S.gblk, S.CH, nullptr, nullptr);
    if (rc == 0) { __syncthreads(); rc = clip_planes(T, F, sh, in, S.gmask, SqueezeTmp{S.t_pos, S.t_loff, S.t_llen, S.t_comp, S.t_ring}); }
    __syncthreads();
    if (rc == SURTR_OVERFLOW) return SURTR_E_CAPACITY;
    if (rc != 0) return rc;
    return consume(T);
}

// Clips `in` by sh.planes[0..F) and hands the resulting Topo (nLive == 0: empty) to `consume`.
// Returns 0 or an error code (uniform over the workgroup).
// (always inlined into the kernel: as a function it would get its LDS objects through generic pointers and every LDS access
// would be a flat_load / flat_store)
template <bool GLOBAL_FALLBACK = true, class LT, class Consume, class LW = NoLdsWork>
__device__ __attribute__((always_inline)) static inline int clip_any(const SolidIn in, uint32_t F, Scratch& S, Shared& sh, LT& L, Consume consume, LW* W = nullptr)
{
    const uint32_t nbV = (in.nv + SURTR_LANES - 1u) >> SURTR_LSH;
    int rc = SURTR_OVERFLOW;
    if (LW::kN == 0 || in.nv <= LW::kN)
    {
        Topo<InLds> T = lds_topo(S, L, W);
        // pre-pass masks sit in the tail of the ring area while the reduced solid is being emitted
        // (the tail is free again afterwards); a copy of the bit mask goes to global scratch for the
        // all-in-plane corner case of clip_planes
        unsigned long long* bmask = S.gmask; uint2* bblk = S.gblk; uint32_t capEmit = LT::kLH;
        if ((size_t)nbV * 8u <= LT::kLH / 2u)
        {
            bblk = (uint2*)(L.ring + LT::kLH) - nbV;
            bmask = (unsigned long long*)bblk - nbV;
            capEmit = LT::kLH - nbV * 8u;
        }
#ifdef SURTR_STAMP
        const unsigned long long q0 = __builtin_readcyclecounter();
#endif
        if (LW::kN != 0 && in.nv <= LW::kN && in.nv <= SURTR_KEEPALL_V) rc = load_whole(in, F, T, sh);      // a Convex: no culling, no masks
        else rc = prepass(in, F, T, sh, bmask, bblk, capEmit, S.gmask, S.gblk);
#ifdef SURTR_STAMP
        const unsigned long long q1 = __builtin_readcyclecounter();
#endif
        if (rc == 0)
        {
            __syncthreads();
            rc = clip_planes<InLds, LW::kN == 0>(T, F, sh, in, S.gmask, SqueezeTmp{S.t_pos, S.t_loff, S.t_llen, S.t_comp, S.t_ring});
        }
        __syncthreads();
#ifdef SURTR_STAMP
        const unsigned long long q2 = __builtin_readcyclecounter();
        if (rc == 0)
        {
            const int r2 = consume(T);
            if (threadIdx.x == 0 && blockDim.x == 64)
            {
                const int o = gridDim.x > 1900 && F > 8 ? 86 : 90;      // 86..89 convex kernel, 90..93 refit
                atomicAdd(&g_stamp[o], q1 - q0); atomicAdd(&g_stamp[o + 1], q2 - q1); atomicAdd(&g_stamp[o + 2], __builtin_readcyclecounter() - q2); atomicAdd(&g_stamp[o + 3], 1ull);
            }
            return r2;
        }
#endif
        if (rc == 0) return consume(T);
    }
    if (rc != SURTR_OVERFLOW || !GLOBAL_FALLBACK) return rc;
    return clip_global(in, F, S, sh, consume);
}

// The same from an image of the reduced solid (k_prep_pairs ran the pre-pass): load it into the LDS topology,
// run the plane loop; a solid that outgrows the LDS topology is redone on global scratch from the input.
template <bool GLOBAL_FALLBACK = true, class LT, class Consume>
__device__ __attribute__((always_inline)) static inline int clip_image(char* img, uint32_t n, uint32_t hsum, uint32_t posCap, const SolidIn in, uint32_t F, Scratch& S, Shared& sh,
                                 LT& L, Consume consume)
{
    const uint32_t tid = threadIdx.x;
    STAMP_DECL;
    const uint32_t nbV = (in.nv + SURTR_LANES - 1u) >> SURTR_LSH;
    const ImgLayout lay = img_layout(F, nbV, n, hsum);
    Topo<InLds> T = lds_topo(S, L);
    int rc = SURTR_OVERFLOW;
    if (n <= T.capV && hsum <= T.capH)
    {
        const uint32_t* hs = (const uint32_t*)(img + lay.hist); const uint32_t* zs = (const uint32_t*)(img + lay.zhist);
        const uint32_t* nz = (const uint32_t*)(img + lay.nzero);
        for (uint32_t k = tid; k < F; k += group_size()) { sh.hist[k] = hs[k]; sh.zhist[k] = zs[k]; sh.nzero[k] = nz[k]; }
        if (tid == 0) { sh.flagErr = 0; sh.flagBad = 0; }
        // sections are padded to 16 bytes, the LDS arrays are 16-byte aligned multiples of 16 bytes: copy 16-byte words
        struct alignas(16) W16 { uint32_t a, b, c, d; };
        auto copy_words = [&](void* dst, const void* src, uint32_t bytes) {       // 16 bytes per lane and round trip
            W16* d = (W16*)dst; const W16* q = (const W16*)src;
            for (uint32_t i = tid; i < (bytes + 15u) / 16u; i += group_size()) d[i] = q[i];
        };
        copy_words(L.loff, img + lay.loff, 2u * n);
        copy_words(L.llen, img + lay.llen, n);
        copy_words(L.fc, img + lay.comp, n);
        copy_words(L.ring, img + lay.ring, 2u * hsum);
        // positions: in place when k_prep_pairs reserved room for this topology's cut points (the usual case), else a copy.
        // In place means the image is consumed (squeeze() compacts positions where they are): only for callers that never
        // hand the pair to another pass over the same image (k_clip_pairs_half does: its retry list; it passes posCap = 0).
        if (posCap >= T.capV) T.pos = (float*)(img + lay.pos);
        else copy_words(S.pos, img + lay.pos, 12u * n);
        T.nS = n; T.nLive = n; T.hUsed = hsum;
        __syncthreads();
        STAMP(84);
        rc = clip_planes(T, F, sh, in, (const unsigned long long*)(img + lay.mask), SqueezeTmp{S.t_pos, S.t_loff, S.t_llen, S.t_comp, S.t_ring});
        __syncthreads();
        if (rc == 0) return consume(T);
    }
    if (rc != SURTR_OVERFLOW || !GLOBAL_FALLBACK) return rc;
    return clip_global(in, F, S, sh, consume);
}

// ------------------------------------------------------------- small helpers
struct ParkOut { int err; uint32_t voff, n, hoff, nh; bool stale; };
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    float t = ax * bx + ay * by;
    return t + az * bz;
}

// Plane(p0,p1,p2): XMPlaneFromPoints, normalised (SimpleMath.inl:2773-2780).
__device__ __forceinline__ float4 plane_from_points(const float* p0, const float* p1, const float* p2)
{
    const float ax = p0[0] - p1[0], ay = p0[1] - p1[1], az = p0[2] - p1[2];
    const float bx = p0[0] - p2[0], by = p0[1] - p2[1], bz = p0[2] - p2[2];
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const float len = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
    if (len != 0.f) { nx = nx / len; ny = ny / len; nz = nz / len; }
    else { nx = 0.f; ny = 0.f; nz = 0.f; }
    return make_float4(nx, ny, nz, -dot3(nx, ny, nz, p0[0], p0[1], p0[2]));
}

// ------------------------------------------------------------ k_place_cells
__global__ void k_place_cells(uint32_t nfaces, const float* __restrict__ v012, float sx, float sy, float sz,
                              float tx, float ty, float tz, float4* __restrict__ planes)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    float p[9];
    for (int i = 0; i < 3; ++i)
    {
        p[3 * i] = v012[9 * f + 3 * i] * sx + tx;
        p[3 * i + 1] = v012[9 * f + 3 * i + 1] * sy + ty;
        p[3 * i + 2] = v012[9 * f + 3 * i + 2] * sz + tz;
    }
    planes[f] = plane_from_points(p, p + 3, p + 6);
}

// Same with one (scale, translate) per group of cells (recursive refracture: every first-level fragment places its own
// cells in its own bounding box, Src/Surtr.cpp:1799-1803 applied per piece).
__global__ void k_place_cells_groups(uint32_t nfaces, const float* __restrict__ v012, const uint32_t* __restrict__ face_group,
                                     const float* __restrict__ scale3, const float* __restrict__ shift3, float4* __restrict__ planes)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    const uint32_t g = face_group[f];
    const float sx = scale3[3 * g], sy = scale3[3 * g + 1], sz = scale3[3 * g + 2];
    const float tx = shift3[3 * g], ty = shift3[3 * g + 1], tz = shift3[3 * g + 2];
    float p[9];
    for (int i = 0; i < 3; ++i)
    {
        p[3 * i] = v012[9 * f + 3 * i] * sx + tx;
        p[3 * i + 1] = v012[9 * f + 3 * i + 1] * sy + ty;
        p[3 * i + 2] = v012[9 * f + 3 * i + 2] * sz + tz;
    }
    planes[f] = plane_from_points(p, p + 3, p + 6);
}

// A pair whose Mesh clip has no valid answer in the reference (SURTR_E_TOPOLOGY: the degenerate policy of literal_clip.h) yields
// no fragment, is counted in surtr_counts::n_failed and keeps its status for surtr_pair_status; the event goes on.  Any other
// error is the event's.
__device__ __forceinline__ void pair_failed(const Arena& A, int err)
{
    // (cursor 15: pairs; cursor 14 counts flagged FRAGMENTS and is reset by a re-triangulation -- the pairs' count is the event's)
    if (err == SURTR_E_TOPOLOGY) atomicAdd(&A.cursors[15], 1u); else atomicMax(&A.cursors[5], (uint32_t)err);
}

// ------------------------------------------------------------- arena output
__device__ __attribute__((always_inline)) static inline bool arena_take(const Arena& A, Shared& sh, uint32_t nv, uint32_t nh, uint32_t nisl,
                                  uint32_t& voff, uint32_t& hoff, uint32_t& ioff)
{
    __syncthreads();
    if (threadIdx.x == 0)
    {
        sh.misc[0] = atomicAdd(&A.cursors[0], nv);
        sh.misc[1] = atomicAdd(&A.cursors[1], nh);
        sh.misc[2] = nisl ? atomicAdd(&A.cursors[3], nisl) : 0u;
    }
    __syncthreads();
    voff = sh.misc[0]; hoff = sh.misc[1]; ioff = sh.misc[2];
    const bool ok = (uint64_t)voff + nv <= A.capV && (uint64_t)hoff + nh <= A.capH && (uint64_t)ioff + nisl <= A.capIsl;
    __syncthreads();
    return ok;
}

// Writes the clipped solid (one piece of output) to the arena.  Returns 0 / SURTR_E_CAPACITY.
template <class TT>
__device__ __attribute__((always_inline)) static inline int park_topo(Topo<TT>& T, Shared& sh, const Arena& A, uint32_t& voff, uint32_t& n, uint32_t& hoff, uint32_t& nh)
{
    const uint2 tot = index_live(T, sh);
    uint32_t ioff;
    if (!arena_take(A, sh, tot.x, tot.y, 0, voff, hoff, ioff)) return SURTR_E_CAPACITY;
    write_solid(T, A.pos, A.loff, A.llen, A.nbr, voff, hoff);
    n = tot.x; nh = tot.y;
    __syncthreads();
    return 0;
}

// Islands of the clipped Mesh (CheckMeshIsland, Src/Surtr.cpp:2157-2201) + the island-major copy to the
// arena (the re-indexing of m_fractureTask, :1474-1500).  Islands are numbered by their lowest vertex.
template <class TT>
__device__ __attribute__((always_inline)) static inline int park_mesh_islands(Topo<TT>& T, Shared& sh, const Arena& A, PairRec& rec)
{
    typedef typename TT::idx_t I;
    const uint32_t tid = threadIdx.x, nS = T.nS;
    STAMP_DECL;
    // Fast path (LDS topology, one island -- the common case): the live slots as a list, labels and packed indices as
    // 16-bit arrays in the free tail of the ring area, every pass over the live vertices only.
    if (sizeof(I) == 2)
    {
        const uint32_t nA = select_count(T.fc, SURTR_NEVER, nS, sh);        // live <=> no plane clipped it
        uint32_t wtop = T.capH;
        auto carve = [&](uint32_t n) -> uint16_t* {
            if (wtop < T.hUsed + n + 8u) return nullptr;
            wtop -= n; return (uint16_t*)(void*)(T.ring + wtop);
        };
        uint16_t* alist = carve(nA); uint16_t* lab = carve(nS); uint16_t* pidx = carve(nS); uint16_t* roff = carve(nA + 1u);
        if (alist && lab && pidx && roff && T.hUsed < 0xFFFFu)
        {
            select_write(T.fc, SURTR_NEVER, nS, sh, WArr<uint16_t>{alist, nullptr});
            for (uint32_t i = tid; i < nA; i += group_size()) { const uint32_t v = alist[i]; lab[v] = (uint16_t)v; pidx[v] = (uint16_t)i; }
            __syncthreads();
            // min-label propagation with one pointer jump per round (CheckMeshIsland, Src/Surtr.cpp:2157-2201, finds the same components)
            while (true)
            {
                if (tid == 0) sh.changed = 0;
                __syncthreads();
                bool ch = false;
                for (uint32_t i = tid; i < nA; i += group_size())
                {
                    const uint32_t v = alist[i];
                    uint32_t m = lab[v];
                    const I* r = T.ring + T.loff[v];
                    const uint32_t len = T.llen[v];
                    for (uint32_t q = 0; q < len; ++q) { const uint32_t o = lab[(uint32_t)r[q]]; m = o < m ? o : m; }
                    const uint32_t mm = lab[m];
                    m = mm < m ? mm : m;
                    if (m < lab[v]) { lab[v] = (uint16_t)m; ch = true; }
                }
                if (ch) sh.changed = 1;
                __syncthreads();
                const bool more = sh.changed != 0;
                __syncthreads();
                if (!more) break;
            }
            if (tid == 0) sh.changed = 0;
            __syncthreads();
            {
                bool other = false;
                const uint32_t root = lab[alist[0]];
                for (uint32_t i = tid; i < nA; i += group_size()) if (lab[alist[i]] != root) other = true;
                if (other) sh.changed = 1;
            }
            __syncthreads();
            const bool single = sh.changed == 0;
            __syncthreads();
            if (single)
            {
                // ring offsets of the packed solid: exclusive scan of the ring lengths in list order
                auto lenfn = [&](uint32_t i) -> uint2 { return make_uint2((uint32_t)T.llen[alist[i]], 0u); };
                uint32_t hh = 0, dum = 0;
                scan_blocks(nA, T.blk, sh, lenfn, hh, dum);
                const uint32_t nbA = (nA + SURTR_LANES - 1u) >> SURTR_LSH;
                for (uint32_t b = wave_id(); b < nbA; b += group_waves())
                {
                    const uint32_t i = (b << SURTR_LSH) + lane_id();
                    uint2 c = make_uint2(0u, 0u);
                    if (i < nA) c = lenfn(i);
                    const uint2 e = wave_excl2(c);
                    if (i < nA) roff[i] = (uint16_t)(T.blk[b].x + e.x);
                }
                uint32_t voff, hoff, ioff;
                if (!arena_take(A, sh, nA, hh, 1u, voff, hoff, ioff)) return SURTR_E_CAPACITY;
                for (uint32_t i = tid; i < nA; i += group_size())
                {
                    const uint32_t v = alist[i], id = voff + i;
                    A.pos[3 * (size_t)id] = T.pos[3 * v]; A.pos[3 * (size_t)id + 1] = T.pos[3 * v + 1]; A.pos[3 * (size_t)id + 2] = T.pos[3 * v + 2];
                    const uint32_t lo = hoff + roff[i], len = T.llen[v];
                    A.loff[id] = lo; A.llen[id] = len;
                    const I* r = T.ring + T.loff[v];
                    for (uint32_t q = 0; q < len; ++q) A.nbr[lo + q] = (int32_t)pidx[(uint32_t)r[q]];
                }
                if (tid == 0) A.isl[ioff] = make_uint2(nA, hh);
                rec.mv_off = voff; rec.mv_n = nA; rec.mh_off = hoff; rec.mh_n = hh; rec.ni = 1u; rec.isl_off = ioff;
                __syncthreads();
                STAMP(83);
                return 0;
            }
        }
    }
    const uint2 tot = index_live(T, sh);              // aux0 = packed index, aux2 = packed ring offset
    STAMP(80);
    uint32_t* lab = T.aux1;
    for (uint32_t v = tid; v < nS; v += group_size()) lab[v] = v;
    __syncthreads();
    while (true)
    {
        if (tid == 0) sh.changed = 0;
        __syncthreads();
        bool ch = false;
        for (uint32_t v = tid; v < nS; v += group_size())
        {
            if (!T.alive(v)) continue;
            uint32_t m = lab[v];
            const I* r = T.ring + T.loff[v];
            const uint32_t len = T.llen[v];
            for (uint32_t q = 0; q < len; ++q) { const uint32_t o = lab[(uint32_t)r[q]]; m = o < m ? o : m; }
            const uint32_t mm = lab[m];
            m = mm < m ? mm : m;
            if (m < lab[v]) { lab[v] = m; ch = true; }
        }
        if (ch) sh.changed = 1;
        __syncthreads();
        const bool more = sh.changed != 0;
        __syncthreads();     // read before lane 0 clears it again
        if (!more) break;
    }
    STAMP(81);
    auto rootfn = [&](uint32_t v) -> uint2 { return make_uint2((T.alive(v) && lab[v] == v) ? 1u : 0u, 0u); };
    uint32_t ni = 0, dum = 0;
    scan_blocks(nS, T.blk, sh, rootfn, ni, dum);
    uint32_t voff, hoff, ioff;
    if (!arena_take(A, sh, tot.x, tot.y, ni, voff, hoff, ioff)) return SURTR_E_CAPACITY;
    STAMP(82);
    if (ni == 1)
    {
        write_solid(T, A.pos, A.loff, A.llen, A.nbr, voff, hoff);
        if (tid == 0) A.isl[ioff] = tot;
    }
    else
    {
        // island index of a root = its rank among roots (discovery order = lowest vertex first)
        uint32_t* irank = T.aux2; uint32_t* local = T.aux0;
        const uint32_t nb = (nS + SURTR_LANES - 1u) >> SURTR_LSH;
        for (uint32_t b = wave_id(); b < nb; b += group_waves())
        {
            const uint32_t v = (b << SURTR_LSH) + lane_id();
            uint2 c = make_uint2(0u, 0u);
            if (v < nS) c = rootfn(v);
            const uint2 e = wave_excl2(c);
            if (v < nS && c.x) irank[v] = T.blk[b].x + e.x;
        }
        __syncthreads();
        uint32_t vbase = 0, hbase = 0;
        for (uint32_t t = 0; t < ni; ++t)
        {
            auto isfn = [&](uint32_t v) -> uint2 {
                return (T.alive(v) && irank[lab[v]] == t) ? make_uint2(1u, (uint32_t)T.llen[v]) : make_uint2(0u, 0u);
            };
            uint32_t tv = 0, th = 0;
            scan_blocks(nS, T.blk, sh, isfn, tv, th);
            for (uint32_t b = wave_id(); b < nb; b += group_waves())
            {
                const uint32_t v = (b << SURTR_LSH) + lane_id();
                uint2 c = make_uint2(0u, 0u);
                if (v < nS) c = isfn(v);
                const uint2 e = wave_excl2(c);
                if (v < nS && c.x)
                {
                    const uint32_t lv = T.blk[b].x + e.x;
                    const uint32_t dv = voff + vbase + lv;
                    local[v] = lv;
                    A.pos[3 * (size_t)dv] = T.pos[3 * v]; A.pos[3 * (size_t)dv + 1] = T.pos[3 * v + 1];
                    A.pos[3 * (size_t)dv + 2] = T.pos[3 * v + 2];
                    A.llen[dv] = c.y;
                    A.loff[dv] = hoff + hbase + T.blk[b].y + e.y;
                }
            }
            if (tid == 0) A.isl[ioff + t] = make_uint2(tv, th);
            __syncthreads();
            for (uint32_t v = tid; v < nS; v += group_size())
            {
                if (!T.alive(v) || irank[lab[v]] != t) continue;
                const uint32_t dv = voff + vbase + local[v];
                const I* r = T.ring + T.loff[v];
                int32_t* d = A.nbr + A.loff[dv];
                const uint32_t len = T.llen[v];
                for (uint32_t q = 0; q < len; ++q) d[q] = (int32_t)local[(uint32_t)r[q]];
            }
            vbase += tv; hbase += th;
            __syncthreads();
        }
    }
    rec.mv_off = voff; rec.mv_n = tot.x; rec.mh_off = hoff; rec.mh_n = tot.y; rec.ni = ni; rec.isl_off = ioff;
    __syncthreads();
    STAMP(83);
    return 0;
}

// ------------------------------------------------------------- k_clip_convex
// Convex of every (cell, piece) pair first (Src/Surtr.cpp:1466-1468): small solids, one wave per task.
// A solid of the one-wave kernels on global scratch (it does not fit their LDS topology; rare).  Out of line and with
// every argument by value, see pair_global.
__device__ __attribute__((noinline)) static ParkOut solid_global(SolidIn in, uint32_t F, ScratchPool pool, uint32_t wg, Arena A, Shared* shp)
{
    Shared& sh = *shp;
    Scratch S = carve(pool, wg);
    ParkOut o{0, 0u, 0u, 0u, 0u, false};
    o.err = clip_global(in, F, S, sh, [&](auto& T) -> int {
        if (T.nLive == 0) return 0;
        return park_topo(T, sh, A, o.voff, o.n, o.hoff, o.nh);
    });
    return o;
}

// Last resort of the one-wave kernels for a solid whose clip raised SURTR_E_TOPOLOGY in the parallel clipper: the literal,
// single-lane ClipPolyhedron of literal_clip.h on the workgroup's global scratch, result parked like park_topo's.  Out of line,
// everything by value (see pair_global).  o.err: 0 (o.n == 0: the reference's answer is "empty") or the error that stands.
// True (uniform over the workgroup) when some ring of the solid lists a neighbour twice, or a link has no way back (the result
// of a clip that went through a stale ID): a solid the literal clipper takes from the start (see Pieces::mdup).
__device__ static bool solid_is_sliver(const SolidIn in)
{
    bool odd = false;
    for (uint32_t v = threadIdx.x; v < in.nv; v += group_size())
    {
        const int32_t* r = in.nbr + in.loff[v];
        const uint32_t deg = in.llen[v];
        for (uint32_t j = 0; j < deg; ++j)
        {
            const int32_t u = r[j];
            for (uint32_t q = 0; q < j; ++q) if (r[q] == u) odd = true;
            if (u < 0 || (uint32_t)u >= in.nv) { odd = true; continue; }
            const int32_t* ru = in.nbr + in.loff[u];
            const uint32_t du = in.llen[u];
            bool back = false;
            for (uint32_t q = 0; q < du; ++q) if (ru[q] == (int32_t)v) back = true;
            if (!back) odd = true;
        }
    }
    return __syncthreads_or(odd ? 1 : 0) != 0;
}

struct LitRun { int err; uint32_t n, nh; LitSolid LS; const uint32_t* off; bool stale; };
__device__ __attribute__((noinline)) static LitRun literal_run(SolidIn in, uint32_t F, ScratchPool pool, uint32_t wg, Shared* shp, bool ids_set = false,
                                                                uint32_t* too_big = nullptr)
{
    Shared& sh = *shp;
    Scratch S = carve(pool, wg);
    const uint32_t capV = S.CV < S.CH / LIT_STRIDE ? S.CV : S.CH / LIT_STRIDE;
    LitRun r{SURTR_E_TOPOLOGY, 0u, 0u, LitSolid{S.pos, S.g_loff, (int32_t*)S.g_ring, S.g_llen, (int32_t*)S.t_ring, S.g_comp, (int32_t*)S.aux0, capV}, S.aux1, false};
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t n = 0; bool stale = false;
        int rc = literal_clip(in, F, sh.planes, r.LS, &n, &stale, ids_set);
        // Too large for the literal path (a ring of more than LIT_STRIDE entries, more vertices than its scratch holds).  The
        // solids that come here are those the parallel clippers cannot answer for (a walk they cannot follow, or a ring that
        // lists a neighbour twice, on which they differ from the reference), so there is nobody left to ask: the first error
        // stands -- the pair or fragment is flagged as for a case without a valid result in the reference -- and the case is
        // counted (surtr_queue_stats out[90]) so that a host can tell an engine limit from the reference's undefined behaviour.
        if (rc == SURTR_E_CAPACITY) { rc = SURTR_E_TOPOLOGY; if (too_big != nullptr) atomicAdd(too_big, 1u); }
        uint32_t h = 0;
        if (rc == 0) for (uint32_t v = 0; v < n; ++v) { S.aux1[v] = h; h += r.LS.len[v]; }
        sh.misc[2] = (uint32_t)rc; sh.misc[3] = n; sh.misc[4] = h; sh.misc[5] = stale ? 1u : 0u;
    }
    __syncthreads();
    r.err = (int)sh.misc[2]; r.n = sh.misc[3]; r.nh = sh.misc[4]; r.stale = sh.misc[5] != 0u;
    __syncthreads();
    return r;
}
__device__ static void literal_write(const LitRun& r, float* pos, uint32_t* loff, uint32_t* llen, int32_t* nbr, uint32_t voff, uint32_t hoff)
{
    for (uint32_t v = threadIdx.x; v < r.n; v += group_size())
    {
        const size_t id = (size_t)voff + v;
        pos[3 * id] = r.LS.pos[3 * v]; pos[3 * id + 1] = r.LS.pos[3 * v + 1]; pos[3 * id + 2] = r.LS.pos[3 * v + 2];
        const uint32_t lo = hoff + r.off[v], len = r.LS.len[v];
        loff[id] = lo; llen[id] = len;
        for (uint32_t q = 0; q < len; ++q) nbr[lo + q] = r.LS.ring[v * LIT_STRIDE + q];
    }
    __syncthreads();
}
__device__ __attribute__((noinline)) static ParkOut solid_literal(SolidIn in, uint32_t F, ScratchPool pool, uint32_t wg, Arena A, Shared* shp, bool ids_set = false)
{
    const LitRun r = literal_run(in, F, pool, wg, shp, ids_set, &A.cursors[90]);
    ParkOut o{r.err, 0u, 0u, 0u, 0u, r.stale};
    if (r.err != 0 || r.n == 0u) return o;
    uint32_t ioff;
    if (!arena_take(A, *shp, r.n, r.nh, 0, o.voff, o.hoff, ioff)) { o.err = SURTR_E_CAPACITY; return o; }
    literal_write(r, A.pos, A.loff, A.llen, A.nbr, o.voff, o.hoff);
    o.n = r.n; o.nh = r.nh;
    return o;
}

#ifndef SURTR_LITERAL_MESH_V
#define SURTR_LITERAL_MESH_V 2048u      // Mesh solids up to this size may take the literal clip after a topology error (one lane: slow)
#define SURTR_LITERAL_START_V 64u       // Mesh solids up to this size take it from the start when a ring lists a neighbour twice
#endif
#ifndef SURTR_SMALL_WAVES
#define SURTR_SMALL_WAVES 2
#endif
__global__ __launch_bounds__(SURTR_LANES) __attribute__((amdgpu_waves_per_eu(SURTR_SMALL_WAVES, 8))) void k_clip_convex(Pieces P, const float4* __restrict__ planes,
                                                    const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                    uint32_t n_pairs, const uint8_t* __restrict__ outside,
                                                    ScratchPool pool, Arena A, PairRec* __restrict__ pairs,
                                                    const uint2* __restrict__ pair_list, uint32_t* __restrict__ porder,
                                                    const uint32_t* __restrict__ pair_order, uint32_t front_par)
{
    __shared__ Shared sh;
    __shared__ OneWaveLds U;
    LdsTopoSmall& L = U.g.L; LdsWorkSmall& W = U.g.W;
    Scratch S = carve(pool, blockIdx.x);
    const uint32_t tid = threadIdx.x;
    while (true)
    {
        __syncthreads();
        if (tid == 0) sh.misc[7] = atomicAdd(&A.cursors[8], 1u);
        __syncthreads();
        uint32_t p = sh.misc[7];
        if (p >= n_pairs) break;
        // pairs of cells with many planes first (pair_order: the event's pairs by plane count of their cell, descending): the
        // clip of a Convex costs about one step per plane, and the last tasks of the queue set the length of the kernel
        if (pair_order != nullptr) p = pair_order[p];
        const uint32_t cell = pair_list ? pair_list[p].x : cell_begin + p / P.n;
        const uint32_t piece = pair_list ? pair_list[p].y : p % P.n;
        PairRec rec;
        memset(&rec, 0, sizeof(rec));
#ifdef SURTR_STAMP_SMALL
        if (tid == 0) for (int q = 0; q < 16; ++q) sh.ph[q] = 0;
#endif
        bool skip = outside != nullptr && outside[piece] != 0;
        const uint32_t f0 = plane_off[cell], F = plane_off[cell + 1] - f0;
        if (F > SURTR_MAXF) { rec.status = SURTR_E_INVALID; skip = true; }
        int err = 0;
#ifdef SURTR_STAMP
        const unsigned long long cx0 = __builtin_readcyclecounter();
        unsigned long long cx1 = cx0, cx2 = cx0, cx3 = cx0;
#endif
        if (!skip)
        {
            for (uint32_t k = tid; k < F; k += group_size()) sh.planes[k] = planes[f0 + k];
            __syncthreads();
            const uint32_t c0 = P.cvo[piece];
            SolidIn cin{P.cpos + 3 * (size_t)c0, P.cloff + c0, P.cllen + c0, P.cnbr, P.cvo[piece + 1] - c0, P.ctri + c0, P.crad + c0,
                        P.cperm + c0, P.cposr_s + c0, P.cbsph + P.cbo[piece]};
            uint32_t which = 0;
            const bool sliver = P.cdup[piece] != 0;      // a ring lists a neighbour twice: the literal clipper below, from the start
#ifdef SURTR_NO_SMALL_CLIP      // (diagnostic builds: the general clipper alone)
            err = SC_FALLBACK;
#else
            uint32_t stop = 0xFFFFFFFFu;
            err = sliver ? SURTR_E_TOPOLOGY : small_clip(cin, F, sh, U.f, &which, &stop);
#endif
            if (tid == 0) atomicAdd(&A.cursors[err == 0 ? 80 : 81], 1u);       // (diagnostic: tasks the regular clipper took / handed on)
#ifdef SURTR_STAMP
            cx1 = __builtin_readcyclecounter();
#endif
            if (err == 0)
            {
                const uint32_t nv = U.f.nv[which];
                if (nv != 0u) err = sc_park(U.f.buf[which], nv, sh, A.cursors, A.pos, A.loff, A.llen, A.nbr, A.capV, A.capH, rec.cv_off, rec.cv_n, rec.ch_off, rec.ch_n);
            }
            __syncthreads();
#ifdef SURTR_STAMP
            cx2 = __builtin_readcyclecounter();
#endif
            if (err == SC_FALLBACK)
            {
                // The general clipper goes on where the regular one stopped: from the solid before that plane (the reference's
                // compacted solid, :464-495), staged through this workgroup's global scratch (the arrays of the wide topology,
                // which the LDS variant leaves alone), with the remaining planes.  Anything but success there starts over from
                // the piece's Convex and all planes, as before.
                SolidIn cx = cin; uint32_t Fx = F;
                const bool resumed = SC_RESUME && stop != 0xFFFFFFFFu && stop > 0u && stop < F;
                if (resumed)
                {
                    const uint32_t nvs = U.f.nv[which];
                    cx = sc_stage(U.f.buf[which], nvs, S.pos, S.g_loff, S.g_llen, S.g_ring);
                    Fx = F - stop;
                    for (uint32_t k = tid; k < Fx; k += group_size()) sh.pmar[k] = sh.planes[k + stop];      // (pmar: free until the estimate below)
                    __syncthreads();
                    for (uint32_t k = tid; k < Fx; k += group_size()) sh.planes[k] = sh.pmar[k];
                    __syncthreads();
                    if (tid == 0) atomicAdd(&A.cursors[78], 1u);       // (diagnostic: general clips resumed from a later plane)
                }
                err = clip_any<false>(cx, Fx, S, sh, L, [&](auto& T) -> int {
                    if (T.nLive == 0) return 0;
                    return park_topo(T, sh, A, rec.cv_off, rec.cv_n, rec.ch_off, rec.ch_n);
                }, &W);
                __syncthreads();
                if (resumed)
                {
                    for (uint32_t k = tid; k < F; k += group_size()) sh.planes[k] = planes[f0 + k];      // (the cost estimate below and the fall-backs want them all)
                    __syncthreads();
                }
            }
            if (err == SURTR_OVERFLOW)
            {
                const ParkOut o = solid_global(cin, F, pool, blockIdx.x, A, &sh);
                err = o.err; rec.cv_off = o.voff; rec.cv_n = o.n; rec.ch_off = o.hoff; rec.ch_n = o.nh;
                __syncthreads();
            }
        }
        if (err == SURTR_E_TOPOLOGY)
        {
            // the parallel relink met a walk it cannot follow (a degenerate sliver): the literal single-lane clip has the
            // reference's answer wherever the reference has one (mostly "empty")
            const uint32_t c0 = P.cvo[piece];
            const SolidIn cin{P.cpos + 3 * (size_t)c0, P.cloff + c0, P.cllen + c0, P.cnbr, P.cvo[piece + 1] - c0, P.ctri + c0, P.crad + c0,
                              P.cperm + c0, P.cposr_s + c0, P.cbsph + P.cbo[piece]};
            const ParkOut o = solid_literal(cin, F, pool, blockIdx.x, A, &sh);
            if (o.err == 0) { err = 0; rec.cv_off = o.voff; rec.cv_n = o.n; rec.ch_off = o.hoff; rec.ch_n = o.nh; }
            else if (o.err != SURTR_E_TOPOLOGY) err = o.err;
            __syncthreads();
        }
        if (err == SURTR_E_TOPOLOGY) { rec.cv_bad = 1; rec.cv_off = 0; rec.cv_n = 1; rec.ch_off = 0; rec.ch_n = 0; err = 0; }
        if (err != 0) { rec.status = (uint32_t)err; rec.cv_n = 0; if (tid == 0) atomicMax(&A.cursors[5], (uint32_t)err); }
        if (tid == 0)
        {
            if (!front_par) pairs[p] = rec;
            else
            {
                // k_prep_pairs runs beside this kernel and writes the img_* words of the same record: everything but those
                PairRec& o = pairs[p];
                o.cv_off = rec.cv_off; o.cv_n = rec.cv_n; o.ch_off = rec.ch_off; o.ch_n = rec.ch_n;
                o.mv_off = 0; o.mv_n = 0; o.mh_off = 0; o.mh_n = 0; o.ni = 0; o.isl_off = 0; o.status = rec.status; o.cv_bad = rec.cv_bad;
            }
        }
        if (rec.cv_n != 0 && porder != nullptr)
        {
            // Cost class of the Mesh pre-pass of this pair, so that k_prep_pairs can start with the expensive ones: the
            // sphere test of its pass A0 on every eighth group of the piece's sorted vertices (sh.planes / sh.pmar still
            // hold this cell's planes and margins).  An estimate only: it orders work, it decides nothing.
            const uint32_t m0 = P.mvo[piece], V = P.mvo[piece + 1] - m0;
            const float4* bs = P.mbsph + P.mbo[piece];
            const uint32_t nsb = (V + SURTR_SB - 1u) / SURTR_SB;
            __syncthreads();
            for (uint32_t k = tid; k < F; k += group_size())      // the margins of prepass_select (the Convex itself was loaded whole)
            {
                const float4 pk = sh.planes[k];
                const float n1 = fabsf(pk.x) + fabsf(pk.y) + fabsf(pk.z);
                const float n2 = sqrtf(pk.x * pk.x + pk.y * pk.y + pk.z * pk.z) * 1.0001f;
                sh.pmar[k] = make_float4(n2 * 1.00101f, 1.0e-5f * fabsf(pk.w), 1.0e-5f * n1, 0.f);
            }
            __syncthreads();
            uint32_t und = 0;
            for (uint32_t sb = tid * 8u; sb < nsb; sb += group_size() * 8u)
            {
                const float4 sp = bs[sb];
                const float mag = fabsf(sp.x) + fabsf(sp.y) + fabsf(sp.z) + sp.w;
                bool decided = false;
                for (uint32_t k = 0; k < F; ++k)
                {
                    const float4 mk = sh.pmar[k];
                    const float sk = plane_dist(sh.planes[k], sp.x, sp.y, sp.z);
                    const float margin = sp.w * mk.x + mk.y + mk.z * mag;
                    if (sk > margin) { decided = true; break; }
                    if (!(sk < -margin)) break;
                }
                if (!decided) ++und;
            }
            const uint2 tot = wave_incl_scan2(make_uint2(und, 0u));
            if (tid == group_size() - 1u)
            {
                // half-octave classes of the sampled count (1 .. ~V/64)
                uint32_t l2 = 0; while ((tot.x >> (l2 + 1u)) != 0u) ++l2;
                uint32_t cls = 2u * l2 + (l2 ? ((tot.x >> (l2 - 1u)) & 1u) : 0u);
                cls = cls > 4u ? cls - 4u : 0u; if (cls > 15u) cls = 15u;
                porder[(size_t)cls * n_pairs + atomicAdd(&A.cursors[48u + cls], 1u)] = p;
            }
        }
#ifdef SURTR_STAMP_SMALL
        if (tid == 0) for (int q = 0; q < 16; ++q) if (sh.ph[q]) atomicAdd(&g_stamp[q], sh.ph[q]);
#endif
#ifdef SURTR_STAMP
        if (tid == 0)
        {
            // the slowest pair of the launch: where its time went (scripts/stamps_convex.py)
            cx3 = __builtin_readcyclecounter();
            const unsigned long long d = cx3 - cx0;
            atomicAdd(&g_stamp2[32], d); atomicAdd(&g_stamp2[33], 1ull);
            const unsigned long long old = atomicMax(&g_stamp2[34], d);
            if (d > old)
            {
                g_stamp2[35] = cx1 - cx0; g_stamp2[36] = cx2 - cx1; g_stamp2[37] = cx3 - cx2; g_stamp2[38] = F;
                for (int q = 0; q < 7; ++q) g_stamp2[40 + q] = U.f.tph[q];
            }
        }
#endif
    }
}

// -------------------------------------------------------------- k_prep_pairs
// Pre-pass of the Mesh of every pair whose Convex survived, as a kernel of its own: it needs no LDS topology, so
// several workgroups share a CU and hide each other's gather latency.  The reduced solid goes to HBM as an
// "image" in the layout of the LDS topology; k_clip_pairs loads it with plain copies.  Pairs this kernel leaves
// alone (small solids, image arena full) are pre-passed by k_clip_pairs itself.
#ifndef SURTR_PREP_MINV
#define SURTR_PREP_MINV 2048u      // smaller meshes are pre-passed by k_clip_pairs itself (measured on BASELINE configs[4])
#endif
#ifndef SURTR_PREP_NB
#define SURTR_PREP_NB 1024u         // 64-vertex blocks whose masks fit this kernel's LDS (65536 vertices)
#endif
#ifndef SURTR_PREP_WAVES
#define SURTR_PREP_WAVES 7       // workgroups per CU = waves per SIMD: the register budget is set for that (<= 72 VGPRs)
#endif
// OLD_SELECT = false: every piece of the event takes the sorted selection (the host checks: up to 65 534 vertices each) -- round 3's
// selection is then not compiled into the kernel (28 spilled registers -> 2)
template <bool OLD_SELECT = true>
__device__ __attribute__((always_inline)) static inline void prep_pairs_body(Shared& sh, unsigned long long* lmask, uint2* lblk, uint32_t rec_on,
                                                         const Pieces& P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs, const PrepPool& pool, const Arena& A, const ImgArena& IA, uint32_t capV, uint32_t capVs,
                                                         PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                                         uint32_t* __restrict__ order, const uint32_t* __restrict__ porder,
                                                         uint32_t* __restrict__ horder, uint32_t half_on, uint32_t big_quota, uint32_t big_n, uint32_t small_cap, uint32_t heavy_need)
{
    // order[c * n_pairs + i]: the pairs of cost class c (0 light .. 15 heavy); k_clip_pairs starts with the heavy
    // ones, so that a pair that takes milliseconds (one that outgrows the LDS topology) is not left for the end
    auto enqueue = [&](uint32_t p, uint32_t cls) { order[(size_t)cls * n_pairs + atomicAdd(&A.cursors[16u + cls], 1u)] = p; };
    auto enqueue_half = [&](uint32_t p, uint32_t cls) { horder[(size_t)cls * n_pairs + atomicAdd(&A.cursors[64u + cls], 1u)] = p; };
    // the small tier of the record clipper (k_clip_pairs_rec): its table follows the half kernel's
    auto enqueue_small = [&](uint32_t p, uint32_t cls) { horder[(size_t)(16u + cls) * n_pairs + atomicAdd(&A.cursors[128u + cls], 1u)] = p; };
    const uint32_t tid = threadIdx.x;
    const bool front_par = (rec_on & 4u) != 0u;
    char* sp = pool.base + (size_t)blockIdx.x * pool.per_wg;
    auto take = [&](size_t bytes) { char* r = sp; sp += (bytes + 255) & ~(size_t)255; return r; };
    uint32_t* needy = (uint32_t*)take((size_t)pool.VMAX * 4);
    uint32_t* orig = (uint32_t*)take((size_t)pool.VMAX * 4);
    uint32_t* und = (uint32_t*)take((size_t)(pool.VMAX / SURTR_SB + 2) * 4);
    unsigned long long* gmask = (unsigned long long*)take((size_t)(pool.VMAX / SURTR_LANES + 2) * 8);
    uint2* gblk = (uint2*)take((size_t)(pool.VMAX / SURTR_LANES + 2) * 8);
    uint8_t* vfc = (uint8_t*)take((size_t)pool.VMAX + 64);
    uint8_t* fcb = (uint8_t*)take((size_t)pool.VMAX + 64);
    uint32_t* klist = (uint32_t*)take((size_t)pool.VMAX * 4);
    uint32_t* walks = (uint32_t*)take((size_t)pool.VMAX * 4);
    uint16_t* sid16 = (uint16_t*)take((size_t)pool.VMAX * 2 + 64);
#ifdef SURTR_STAMP
    const unsigned long long wg_t0 = __builtin_readcyclecounter();
    unsigned long long wg_work = 0;
#endif
    while (true)
    {
        __syncthreads();
        if (tid == 0)
        {
            // next ticket -> pair, expensive cost classes first (k_clip_convex filled porder with the pairs whose Convex survived)
            uint32_t t = atomicAdd(&A.cursors[9], 1u), pp = 0xFFFFFFFFu;
            if (front_par) { if (t < n_pairs) pp = t; }      // (no queue: k_clip_convex is still running, see launch_event)
            else
                for (int cls = 15; cls >= 0; --cls)
                {
                    const uint32_t cnt = A.cursors[48 + cls];
                    if (t < cnt) { pp = porder[(size_t)cls * n_pairs + t]; break; }
                    t -= cnt;
                }
            sh.misc[7] = pp;
        }
        __syncthreads();
        const uint32_t p = sh.misc[7];
        if (p >= n_pairs) break;
        const uint32_t cell = pair_list ? pair_list[p].x : cell_begin + p / P.n;
        const uint32_t piece = pair_list ? pair_list[p].y : p % P.n;
        if (front_par)
        {
            // the Convex of this pair is being clipped beside this kernel: whether it is empty is not known yet.  The band of every pair
            // is prepared (a pair whose Convex turns out empty is skipped by the clip kernels; its Mesh is outside the cell as well,
            // as a rule found so by the sphere tests in a few steps), and the words of the record this kernel owns start from zero.
            if (tid == 0) { pairs[p].img_fmt = IMG_NONE; pairs[p].img_off = 0; pairs[p].img_n = 0; pairs[p].img_h = 0; pairs[p].img_pc = 0; }
            if (plane_off[cell + 1] - plane_off[cell] > SURTR_MAXF) continue;      // (k_clip_convex fails the pair with SURTR_E_INVALID)
        }
        else if (pairs[p].cv_n == 0 || pairs[p].status != 0) continue;
        const uint32_t m0 = P.mvo[piece], V = P.mvo[piece + 1] - m0;
        if (V < SURTR_PREP_MINV || V > pool.VMAX)
        {
            // not pre-passed here: the whole Mesh decides between the half-size kernel and the regular one
            if (tid == 0)
            {
                // (a sliver Mesh goes to the regular kernel, which has the scratch for the literal clipper: see clip_pairs_body)
                const bool sliver = P.mdup[piece] != 0 && V <= SURTR_LITERAL_START_V;
                if (half_on && !sliver && V <= pool.VMAX && fits_half(V, P.mloff[m0 + V] - P.mloff[m0], capVs)) enqueue_half(p, 1u);
                else enqueue(p, V > pool.VMAX ? 13u : 0u);
            }
            continue;
        }
#ifdef SURTR_STAMP
        const unsigned long long pair_t0 = __builtin_readcyclecounter();
#endif
        STAMP_DECL;
        const uint32_t f0 = plane_off[cell], F = plane_off[cell + 1] - f0;
        for (uint32_t k = tid; k < F; k += group_size()) sh.planes[k] = planes[f0 + k];
        __syncthreads();
        SolidIn min{P.mpos + 3 * (size_t)m0, P.mloff + m0, P.mllen + m0, P.mnbr, V, P.mtri + m0, P.mrad + m0,
                    P.mperm + m0, P.mposr_s + m0, P.mbsph + P.mbo[piece]};
        STAMP(70);
        const uint32_t nbV = (V + SURTR_LANES - 1u) >> SURTR_LSH;
        unsigned long long* bmask = nbV <= SURTR_PREP_NB ? lmask : gmask;
        uint2* bblk = nbV <= SURTR_PREP_NB ? lblk : gblk;
        uint32_t n = 0, hsum = 0;
#ifndef SURTR_PREP_G
#define SURTR_PREP_G 1           // occupancy hides the gather latency here, not unrolling
#define SURTR_PREP_NBATCH 4
#endif
        // a piece with a sorted copy whose groups fit the LDS table: selection by sphere hierarchy and fc look-ups (prep_sorted.h)
        constexpr uint32_t kUbWords = (SURTR_PREP_NB - SURTR_PS_NB) * 2u;      // 32-bit words of one bit per group (the tail of the table area)
        const bool sorted_sel = !(rec_on & 2u) && nbV <= SURTR_PREP_NB && (V + SURTR_SB - 1u) / SURTR_SB <= 32u * kUbWords && V < 0xFFFFu && P.mrow_s != nullptr;
        if (sorted_sel)
        {
            const SortedRings sr{P.mrow_s + m0, P.miperm + m0, P.mbsph2 + P.mbo2[piece], P.mbsph3 + P.mbo3[piece]};
            prepass_select_sorted<SURTR_PREP_NB, kUbWords>(min, sr, F, sh, (unsigned char*)lmask, vfc, (uint16_t*)needy, (uint16_t*)und, klist, walks, n, hsum);
        }
        else if constexpr (OLD_SELECT) prepass_select<SURTR_PREP_G, SURTR_PREP_NBATCH>(min, F, sh, bmask, bblk, needy, und, n, hsum);
        const bool toolong = sh.flagBad != 0;
        __syncthreads();
        STAMP(71);
        uint32_t fmt = IMG_NARROW, off16 = 0;
        if (!OLD_SELECT && !sorted_sel) fmt = IMG_NONE;      // (cannot happen: the host launches this variant only when every piece qualifies)
        else if (n == 0) fmt = IMG_EMPTY;
        else if (toolong || n > 2u * capV || hsum > 2u * SURTR_LH || n >= InLds::SENT) fmt = IMG_WIDE;
        // room for the cut points behind the positions, for the topology of the kernel that will take the pair
        const bool to_half = half_on && fits_half(n, hsum, capVs);
        // The band goes out as a record image when the record clipper will take the pair as it is: a regular band (no vertex in a
        // plane before its first clipping plane, rings of at most seven entries, no sliver piece), of a size the id map and the
        // record ids hold, of a cost class a record kernel serves (the whole-CU kernel only when it runs: big_quota all ones).
        const bool rec_fmt = (rec_on & 1u) != 0u && sorted_sel && fmt == IMG_NARROW && !to_half && F <= WC_MAXF && n < WC_MAXN && n <= (rec_on >> 8) &&
                             sh.misc[5] == 0u && sh.deg7 == 0u && P.mdup[piece] == 0 &&
                             (fits_with_room(n, hsum, capV, SURTR_LH) || big_quota == 0xFFFFFFFFu);
        uint32_t ncut_rec = 0, maxb_rec = 0;
        if (rec_fmt)
        {
            const RecLayout rl = rec_layout(F, n);
            const uint32_t need16 = rl.total / 16u;
            if (tid == 0) sh.misc[0] = atomicAdd(&A.cursors[10], need16);
            __syncthreads();
            off16 = sh.misc[0];
            __syncthreads();
            if ((uint64_t)off16 + need16 > IA.cap16) fmt = IMG_NONE;      // arena full: the clip kernel does this pair alone
            else
            {
                prepass_emit_records(min, F, sh, bmask, bblk, klist, orig, fcb, sid16, sh.pw, IA.base + (size_t)off16 * 16u, n, ncut_rec, maxb_rec);
                fmt = IMG_REC;
            }
        }
        // (none for the half-size kernel: it works on a copy, so that its retry list finds the image as it was)
        const uint32_t posCap = to_half ? 0u : (fits_with_room(n, hsum, capV, SURTR_LH) ? capV : 2u * capV);
        const ImgLayout lay = img_layout(F, nbV, n, hsum, posCap);
        if (fmt == IMG_NARROW && !rec_fmt)
        {
            const uint32_t need16 = lay.total / 16u;
            if (tid == 0) sh.misc[0] = atomicAdd(&A.cursors[10], need16);
            __syncthreads();
            off16 = sh.misc[0];
            __syncthreads();
            if ((uint64_t)off16 + need16 > IA.cap16) fmt = IMG_NONE;      // arena full: k_clip_pairs does this pair alone
        }
        STAMP(72);
        if (fmt == IMG_NARROW)
        {
            char* img = IA.base + (size_t)off16 * 16u;
            unsigned long long* gm = (unsigned long long*)(img + lay.mask);
            for (uint32_t b = tid; b < nbV; b += group_size()) gm[b] = bmask[b];
            STAMP(73);
            Topo<InLds> T;
            T.loff = (uint16_t*)(img + lay.loff); T.llen = (uint8_t*)(img + lay.llen); T.fc = (uint8_t*)(img + lay.comp); T.gcomp = nullptr;
            T.ring = (uint16_t*)(img + lay.ring); T.pos = (float*)(img + lay.pos);
            T.succ = T.pred = T.pcnt = T.aux0 = T.aux1 = T.aux2 = T.aux3 = nullptr; T.blk = nullptr;
            T.capV = n; T.capH = hsum; T.nS = T.nLive = T.hUsed = 0; T.kcur = T.n0cur = 0; T.zmode = false;
            if (tid < 4u) sh.cutmask[tid] = 0u;
            __syncthreads();
            // (also counts sh.nzero, collects sh.cutmask)
            if (sorted_sel) prepass_emit_klist(min, F, sh, T, bmask, bblk, klist, orig, (uint2*)und, n, hsum, maxb_rec);
            else if constexpr (OLD_SELECT) prepass_emit(min, F, sh, T, bmask, bblk, orig, n, hsum);
            STAMP(74);
            prepass_finish_hist(F, sh);
            uint32_t* hs = (uint32_t*)(img + lay.hist); uint32_t* zs = (uint32_t*)(img + lay.zhist); uint32_t* nz = (uint32_t*)(img + lay.nzero);
            for (uint32_t k = tid; k < F; k += group_size()) { hs[k] = sh.hist[k]; zs[k] = sh.zhist[k]; nz[k] = sh.nzero[k]; }
            STAMP(75);
        }
        if (tid == 0)
        {
            pairs[p].img_fmt = fmt; pairs[p].img_off = off16; pairs[p].img_n = n; pairs[p].img_h = hsum; pairs[p].img_pc = posCap;
            if (fmt == IMG_REC) atomicAdd(&A.cursors[91], 1u);      // (diagnostic: surtr_queue_stats)
            if (sorted_sel) atomicAdd(&A.cursors[92], 1u);
            // classes 14 and 15 go to k_clip_pairs_big: bands that leave the regular LDS topology little room to grow
            // (the first plane alone may add a thousand vertices), and solids beyond any LDS topology
            // solids of at most half the half-size topology have their own table (k_clip_pairs_half)
            uint32_t cls = 15u;
            if (fmt == IMG_NONE) cls = 13u;
            else if (fmt == IMG_NARROW || fmt == IMG_REC)
            {
                // cost of the clip = cutting planes x ~41 000 cycles + load / islands / copy ~25 cycles per vertex, in units of
                // 110 000 cycles; the planes that clip an original vertex of the band are a lower bound of the cutting planes
                const uint32_t ncut = fmt == IMG_REC ? ncut_rec : (uint32_t)(__builtin_popcount(sh.cutmask[0]) + __builtin_popcount(sh.cutmask[1]) +
                                                 __builtin_popcount(sh.cutmask[2]) + __builtin_popcount(sh.cutmask[3]));
                const uint32_t cost = (41u * ncut + n / 40u) / 110u;
                cls = !fits_with_room(n, hsum, capV, SURTR_LH) ? 14u : 1u + (cost < 11u ? cost : 11u);
                // split arrangement (heavy_need != 0): a band the record clipper will run out of LDS on (its need at the worst plane
                // follows the band size and the largest bucket: scripts/wave_need.py) would be handed on after a plane or two and then
                // be the slowest task of the catcher -- it goes to the double-size general clipper (k_clip_pairs_big) from the start
                if (cls < 12u && heavy_need != 0u && sorted_sel && (515u * n + 4185u * maxb_rec) / 100u + 393u > heavy_need) cls = 14u;
                // a band vertex lies in a plane: the record clipper hands the pair to the general clipper, which takes longer --
                // such pairs go first (the top regular class), not into the tail of the queue
                if (cls < 12u && fmt == IMG_NARROW) { bool inplane = false; for (uint32_t k = 0; k < F; ++k) if (sh.nzero[k] != 0u) inplane = true; if (inplane) cls = 12u; }
            }
            // k_clip_pairs_big has a few dozen workgroups: on a mesh whose bands outgrow the regular topology as a rule (some
            // 100 000 vertices) it takes the first `big_quota` such pairs and the regular kernel's workgroups do the others on
            // their global scratch (class 13), all of them at once instead of a queue behind 48
            // (big_quota = all ones: the record clipper's whole-CU kernel takes the large bands -- those it can hold, narrow images of
            // fewer than WC_MAXN vertices; the others are spread over the regular kernel's workgroups as before)
            if (big_quota == 0xFFFFFFFFu)
            {
                if (cls >= 14u && !((fmt == IMG_NARROW || fmt == IMG_REC) && n < WC_MAXN)) cls = 13u;
                // ... and a band of more than big_n vertices that the general clipper's topology would still hold: its first planes
                // clip thousands of vertices at once, more than the regular record clipper's 56 KB take (it would hand the pair on
                // after the loader and a plane or two) -- the whole-CU variant has the room
                else if (cls < 12u && (fmt == IMG_NARROW || fmt == IMG_REC) && n > big_n && n < WC_MAXN) cls = 14u;
            }
            else if (cls >= 14u && fmt != IMG_EMPTY && atomicAdd(&A.cursors[84], 1u) >= big_quota) cls = 13u;
            // The small tier takes the record images whose worst plane will fit its LDS: the need follows the band size and the largest
            // bucket (measured on configs[3], scripts/wave_need.py: need ~ 5.15 n + 41.85 maxbucket + 393 bytes, residual sigma 2.3 KB),
            // of which 16 bytes per vertex of the bucket are the stage, which a plane leaves in global memory when it must
            // (wave_clip.h); what does not fit after all comes back through class 12 of the large tier
            const uint32_t s_cap = small_cap;
            const bool to_small = s_cap != 0u && fmt == IMG_REC && cls < 12u && (515u * n + 2585u * maxb_rec) / 100u + 393u + 4096u <= s_cap;
            if (to_small) enqueue_small(p, cls);
            else if (fmt == IMG_NARROW && to_half) enqueue_half(p, cls < 6u ? cls : 6u);      // (a record image is never to_half)
            else if (fmt != IMG_EMPTY) enqueue(p, cls);
        }
#ifdef SURTR_STAMP
        if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - pair_t0; wg_work += d; int bkt = 0; while ((d >> bkt) > 1 && bkt < 30) ++bkt; bkt = bkt < 16 ? 0 : bkt - 16; if (bkt > 7) bkt = 7; atomicAdd(&g_stamp[62 + bkt], 1ull); }
#endif
    }
#ifdef SURTR_STAMP
    if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - wg_t0; atomicAdd(&g_stamp[56], d); atomicMax(&g_stamp[57], d); atomicAdd(&g_stamp[58], 1ull); atomicAdd(&g_stamp[59], wg_work); atomicMax(&g_stamp[60], wg_work); }
#endif
}

__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(SURTR_PREP_WAVES, 8))) void k_prep_pairs(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs, PrepPool pool, Arena A, ImgArena IA, uint32_t capV, uint32_t capVs,
                                                         PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                                         uint32_t* __restrict__ order, const uint32_t* __restrict__ porder,
                                                         uint32_t* __restrict__ horder, uint32_t half_on, uint32_t big_quota, uint32_t big_n, uint32_t rec_on, uint32_t small_cap, uint32_t heavy_need)
{
    __shared__ Shared sh;
    __shared__ unsigned long long lbuf[2u * SURTR_PREP_NB];      // masks (first half) + per-block pairs (second half); the sorted selection's tables
    unsigned long long* lmask = lbuf; uint2* lblk = (uint2*)(lbuf + SURTR_PREP_NB);
    prep_pairs_body(sh, lmask, lblk, rec_on, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, capV, capVs, pairs, pair_list, order, porder, horder, half_on, big_quota, big_n, small_cap, heavy_need);
}

// the same for events whose every piece takes the sorted selection
__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(SURTR_PREP_WAVES, 8))) void k_prep_pairs_sorted(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs, PrepPool pool, Arena A, ImgArena IA, uint32_t capV, uint32_t capVs,
                                                         PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                                         uint32_t* __restrict__ order, const uint32_t* __restrict__ porder,
                                                         uint32_t* __restrict__ horder, uint32_t half_on, uint32_t big_quota, uint32_t big_n, uint32_t rec_on, uint32_t small_cap, uint32_t heavy_need)
{
    __shared__ Shared sh;
    __shared__ unsigned long long lbuf[2u * SURTR_PREP_NB];      // masks (first half) + per-block pairs (second half); the sorted selection's tables
    unsigned long long* lmask = lbuf; uint2* lblk = (uint2*)(lbuf + SURTR_PREP_NB);
    prep_pairs_body<false>(sh, lmask, lblk, rec_on, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, capV, capVs, pairs, pair_list, order, porder, horder, half_on, big_quota, big_n, small_cap, heavy_need);
}

// The same with four times the threads per pair, for events of so few pairs (a rank's block of a sharded event) that the
// regular launch leaves most of the chip idle while every pair waits for its own 50 000 vertices: the passes over the
// vertices are data parallel, so a pair is done in about a third of the time.
__global__ __launch_bounds__(SURTR_WG_WIDE) void k_prep_pairs_wide(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs, PrepPool pool, Arena A, ImgArena IA, uint32_t capV, uint32_t capVs,
                                                         PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                                         uint32_t* __restrict__ order, const uint32_t* __restrict__ porder,
                                                         uint32_t* __restrict__ horder, uint32_t half_on, uint32_t big_quota, uint32_t big_n, uint32_t rec_on, uint32_t small_cap, uint32_t heavy_need)
{
    __shared__ Shared sh;
    __shared__ unsigned long long lbuf[2u * SURTR_PREP_NB];      // masks (first half) + per-block pairs (second half); the sorted selection's tables
    unsigned long long* lmask = lbuf; uint2* lblk = (uint2*)(lbuf + SURTR_PREP_NB);
    prep_pairs_body(sh, lmask, lblk, rec_on, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, capV, capVs, pairs, pair_list, order, porder, horder, half_on, big_quota, big_n, small_cap, heavy_need);
}

// -------------------------------------------------------------- k_clip_pairs
// Mesh of every pair whose Convex survived (:1470-1500): clip, islands, island-major copy to the arena.
// One workgroup: pairs of cost classes cls_hi..cls_lo, heavy first (tickets from A.cursors[qcur]).
// The Mesh of pair p on global scratch (32-bit topology): solids that do not fit, or outgrew, the LDS topology.  Rare, so
// out of line and with every argument by value: nothing of the caller's state has to live in memory for it (structures
// handed over by reference would be written to every lane's private scratch once per pair).  Stores pairs[p] itself.
// literal != 0: the Mesh goes through the literal single-lane clipper first -- a sliver piece (coincident vertices, doubled
// neighbours) on which the parallel relink met a walk it cannot follow.  The literal clip of the whole Mesh has the reference's
// answer wherever the reference has one; it is parked as one solid and takes the island split below as a clip by no planes.
__device__ __attribute__((noinline)) static int pair_global(Pieces P, uint32_t piece, uint32_t F, ScratchPool pool, uint32_t wg, Arena A,
                                                            Shared* shp, PairRec* pairs, uint32_t p, uint32_t literal)
{
    Shared& sh = *shp;
    Scratch S = carve(pool, wg);
    PairRec rec = pairs[p];
    const uint32_t m0 = P.mvo[piece];
    SolidIn min{P.mpos + 3 * (size_t)m0, P.mloff + m0, P.mllen + m0, P.mnbr, P.mvo[piece + 1] - m0, P.mtri + m0, P.mrad + m0,
                P.mperm + m0, P.mposr_s + m0, P.mbsph + P.mbo[piece]};
    auto consume = [&](auto& T) -> int {
        if (T.nLive == 0) return 0;
        return park_mesh_islands(T, sh, A, rec);
    };
    int err = 0;
    bool gone = false;
    if (literal != 0u)
    {
        const ParkOut o = solid_literal(min, F, pool, wg, A, shp);
        err = o.err;
        gone = err == 0 && o.n == 0u;
        if (err == 0 && o.n != 0u)
        {
            min = SolidIn{A.pos + 3 * (size_t)o.voff, A.loff + o.voff, A.llen + o.voff, A.nbr, o.n, nullptr, nullptr, nullptr, nullptr, nullptr};
            F = 0u;
        }
        __syncthreads();
    }
    if (err == 0 && !gone) err = clip_global(min, F, S, sh, consume);
    __syncthreads();
    if (err == 0 && rec.cv_bad != 0 && rec.ni != 0) err = SURTR_E_TOPOLOGY;
    if (err != 0) { rec.status = (uint32_t)err; rec.ni = 0; if (threadIdx.x == 0) pair_failed(A, err); }
    if (threadIdx.x == 0) pairs[p] = rec;
    return err;
}

// One pair through the general clipper: image (or pre-pass) -> plane loop -> islands -> arena, with the global-scratch and
// literal fallbacks.  rec: the pair's record as k_clip_convex / k_prep_pairs left it; stores pairs[p] unless the pair was handed
// to the half table's retry list.  Every thread of the group must call it.
template <bool HALF = false, class LT>
__device__ __attribute__((always_inline)) static inline void clip_pair_general(Shared& sh, LT& L, Scratch& S, const ScratchPool& pool, uint32_t wg, const Pieces& P,
                                       const float4* __restrict__ planes, const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                       const Arena& A, const ImgArena& IA, PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                       uint32_t* __restrict__ horder, const uint32_t p, PairRec rec)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t cell = pair_list ? pair_list[p].x : cell_begin + p / P.n;
    const uint32_t piece = pair_list ? pair_list[p].y : p % P.n;
    const uint32_t f0 = plane_off[cell], F = plane_off[cell + 1] - f0;
    for (uint32_t k = tid; k < F; k += group_size()) sh.planes[k] = planes[f0 + k];
    __syncthreads();
    const uint32_t m0 = P.mvo[piece];
    SolidIn min{P.mpos + 3 * (size_t)m0, P.mloff + m0, P.mllen + m0, P.mnbr, P.mvo[piece + 1] - m0, P.mtri + m0, P.mrad + m0,
                P.mperm + m0, P.mposr_s + m0, P.mbsph + P.mbo[piece]};
    auto consume = [&](auto& T) -> int {
        if (T.nLive == 0) return 0;
        return park_mesh_islands(T, sh, A, rec);
    };
    if (P.mdup[piece] != 0 && min.nv <= SURTR_LITERAL_START_V)
    {
        // a sliver Mesh (a few vertices, a ring lists a neighbour twice): literal clipper from the start (see pair_global) --
        // on the regular kernel's scratch (the half-size kernel's has no room for it: its retry list).  A larger Mesh with
        // such a ring keeps the parallel clipper (one lane would take milliseconds for it) and falls back only on an error.
        if (HALF) { if (tid == 0) horder[atomicAdd(&A.cursors[64], 1u)] = p; }
        else pair_global(P, piece, F, pool, wg, A, &sh, pairs, p, 1u);
        return;
    }
    int err;
    if (HALF)
    {
        // no global fallback here: a pair that outgrows the half-size topology after all goes to the retry class
        // (class 0), which a second launch of k_clip_pairs picks up
        if (rec.img_fmt == IMG_NARROW) err = clip_image<false>(IA.base + (size_t)rec.img_off * 16u, rec.img_n, rec.img_h, 0u, min, F, S, sh, L, consume);
        else err = clip_any<false>(min, F, S, sh, L, consume);
        __syncthreads();
        if (err == SURTR_OVERFLOW)
        {
            if (tid == 0) horder[atomicAdd(&A.cursors[64], 1u)] = p;
            return;
        }
    }
    else
    {
        err = SURTR_OVERFLOW;
        if (rec.img_fmt == IMG_NARROW) err = clip_image<false>(IA.base + (size_t)rec.img_off * 16u, rec.img_n, rec.img_h, rec.img_pc, min, F, S, sh, L, consume);
        else if (rec.img_fmt != IMG_WIDE) err = clip_any<false>(min, F, S, sh, L, consume);
        __syncthreads();
        if (err == SURTR_OVERFLOW)
        {
            pair_global(P, piece, F, pool, wg, A, &sh, pairs, p, 0u);
            return;
        }
    }
    if (err == SURTR_E_TOPOLOGY && min.nv <= SURTR_LITERAL_MESH_V)
    {
        __syncthreads();
        if (HALF) { if (tid == 0) horder[atomicAdd(&A.cursors[64], 1u)] = p; }       // (the regular kernel redoes the pair: see above)
        else pair_global(P, piece, F, pool, wg, A, &sh, pairs, p, 1u);
        return;
    }
    if (err == 0 && rec.cv_bad != 0 && rec.ni != 0) err = SURTR_E_TOPOLOGY;       // a fragment with an invalid Convex
    if (err != 0) { rec.status = (uint32_t)err; rec.ni = 0; if (tid == 0) pair_failed(A, err); }
    if (tid == 0) pairs[p] = rec;
}

template <bool HALF = false, class LT>
__device__ __attribute__((always_inline)) static inline void clip_pairs_body(Shared& sh, LT& L, const ScratchPool& pool, uint32_t wg, const Pieces& P, const float4* __restrict__ planes,
                                       const uint32_t* __restrict__ plane_off, uint32_t cell_begin, uint32_t n_pairs,
                                       const Arena& A, const ImgArena& IA, PairRec* __restrict__ pairs,
                                       const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                       uint32_t* __restrict__ horder, int cls_hi, int cls_lo, uint32_t qcur)
{
    const uint32_t tid = threadIdx.x;
    Scratch S = carve(pool, wg);
#ifdef SURTR_STAMP
    const unsigned long long wg_t0 = __builtin_readcyclecounter();
#endif
    // classes of the half table this launch takes: 6..1 (k_clip_pairs_half), its retry list (cls_hi < 0), or none
    const int h_hi = HALF ? 6 : (cls_hi < 0 ? 0 : -1), h_lo = cls_hi < 0 ? 0 : 1;
    while (true)
    {
        __syncthreads();
        if (tid == 0)
        {
            // next ticket -> pair (k_prep_pairs filled the tables), heavy classes first
            uint32_t t = atomicAdd(&A.cursors[qcur], 1u), p = 0xFFFFFFFFu;
            if (!HALF)
                for (int cls = cls_hi; cls >= cls_lo; --cls)
                {
                    const uint32_t cnt = A.cursors[16 + cls];
                    if (t < cnt) { p = order[(size_t)cls * n_pairs + t]; break; }
                    t -= cnt;
                }
            if (p == 0xFFFFFFFFu)
                for (int cls = h_hi; cls >= h_lo; --cls)
                {
                    const uint32_t cnt = A.cursors[64 + cls];
                    if (t < cnt) { p = horder[(size_t)cls * n_pairs + t]; break; }
                    t -= cnt;
                }
            sh.misc[7] = p;
        }
        __syncthreads();
        const uint32_t p = sh.misc[7];
        if (p >= n_pairs) break;
        PairRec rec = pairs[p];
        if (rec.cv_n == 0 || rec.status != 0) continue;       // empty Convex: the Mesh is not clipped (:1467-1468)
        if (rec.img_fmt == IMG_EMPTY) continue;               // the pre-pass kernel found nothing left of the Mesh
        if (rec.img_fmt == IMG_REC) rec.img_fmt = IMG_NONE;   // (a record image is no input of the general clipper: never queued here)
#ifdef SURTR_STAMP
        const unsigned long long pair_t0 = __builtin_readcyclecounter();
        if (tid == 0) for (int q = 0; q < 16; ++q) sh.ph[q] = 0;
#endif
        clip_pair_general<HALF>(sh, L, S, pool, wg, P, planes, plane_off, cell_begin, A, IA, pairs, pair_list, horder, p, rec);
#ifdef SURTR_STAMP
        if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - pair_t0; int bkt = 0; while ((d >> bkt) > 1 && bkt < 30) ++bkt; bkt = bkt < 16 ? 0 : bkt - 16; if (bkt > 9) bkt = 9; atomicAdd(&g_stamp[(cls_hi == 15 ? 51 : 21) + bkt], 1ull); }
#endif
    }
#ifdef SURTR_STAMP
    if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - wg_t0; const int o = cls_hi == 15 ? 48 : 16; atomicAdd(&g_stamp[o], d); atomicMax(&g_stamp[o + 1], d); atomicAdd(&g_stamp[o + 2], 1ull); }
#endif
}

// two workgroups per CU = two waves per SIMD: the register budget must stay within 256 VGPRs
__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_clip_pairs(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ horder, int cls_hi, int cls_lo, uint32_t qcur)
{
    __shared__ Shared sh;
    __shared__ LdsTopo L;
    clip_pairs_body(sh, L, pool, blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, A, IA, pairs, pair_list, order, horder, cls_hi, cls_lo, qcur);
}

// The light pairs (cost classes 1..6: reduced solids that leave the half-size topology room to grow): half the
// threads, half the LDS, four workgroups per CU; its own, smaller scratch pool.  Runs beside the other two.
__global__ __launch_bounds__(SURTR_WGS) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_clip_pairs_half(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, uint32_t* __restrict__ horder)
{
    __shared__ Shared sh;
    __shared__ LdsTopoHalf L;
    clip_pairs_body<true>(sh, L, pool, blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, A, IA, pairs, pair_list, nullptr, horder, 6, 1, 12u);
}

// The same with the double-size LDS topology (one workgroup per CU) for cost classes 14 and 15; runs beside
// k_clip_pairs on a second stream.  Its workgroups use the scratch slots after those of k_clip_pairs.
__global__ __launch_bounds__(SURTR_WG) void k_clip_pairs_big(Pieces P, const float4* __restrict__ planes,
                                                             const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                             uint32_t n_pairs,
                                                             ScratchPool pool, uint32_t wg_base, Arena A, ImgArena IA,
                                                             PairRec* __restrict__ pairs, const uint2* __restrict__ pair_list,
                                                             const uint32_t* __restrict__ order)
{
    __shared__ Shared sh;
    __shared__ LdsTopoBig L;
    clip_pairs_body(sh, L, pool, wg_base + blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, A, IA, pairs, pair_list, order, nullptr, 15, 14, 11u);
}

// -------------------------------------------------------------- k_clip_pairs_wave
// The same pairs through the record clipper (wave_clip.h): regular planes only, one memory access per step of a plane's
// dependent chain, the band streamed from HBM bucket by bucket.  It takes the tickets of cost classes cls_hi..cls_lo like
// k_clip_pairs.  A pair it cannot take (no narrow image, a sliver piece, a cell of more than 64 planes, an original that lies in
// a plane) or has to give up (a cut point in a later plane, an irregular cap, a ring of more than seven entries, out of room:
// WC_BAIL -- nothing of the pair has been published, its image is untouched) is clipped right here by the general clipper, on
// the same LDS bytes: k_prep_pairs puts the pairs it knows to be irregular into the heaviest class, so they come first.
struct GenLds { Shared sh; LdsTopo L; };
struct GenLdsBig { Shared sh; LdsTopoBig L; };
// FALLBACK = false: the small tier (k_clip_pairs_rec) -- record images only, no general clipper in the kernel (its LDS and registers
// are the record clipper's alone: three workgroups per CU); a pair it gives up on goes to class 12 of the large tier's table,
// whose kernel runs behind this one.
template <class WL, class GL, bool FALLBACK = true, bool OLD_IMAGES = FALLBACK>
__device__ __attribute__((always_inline)) static inline void clip_pairs_wave_body(unsigned char* lds_raw, uint32_t wg, const Pieces& P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         const ScratchPool& pool, const Arena& A, const ImgArena& IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ horder, int cls_hi, int cls_lo, uint32_t qcur, uint32_t walk0,
                                                         uint32_t* __restrict__ hlist = nullptr, uint32_t cbase_arg = 0u)
{
    WL& W = *reinterpret_cast<WL*>(lds_raw);
    GL& Gn = *reinterpret_cast<GL*>(lds_raw);
    const uint32_t tid = threadIdx.x;
    // this workgroup's scratch slot: used raw by the record clipper (sorted records + positions of a band's originals, positions
    // of the cut points), carved as a Scratch by the general clipper
    char* slot = pool.base + (size_t)wg * pool.per_wg;
    Scratch S{};
    if (FALLBACK) S = carve(pool, wg);
    const uint32_t cbase = cbase_arg ? cbase_arg : (FALLBACK ? 16u : 128u);       // class counts of the table this kernel pulls from
    if (!FALLBACK && hlist != nullptr && tid == 0) atomicAdd(&A.cursors[147], 1u);      // (the catcher beside this kernel: "it has started")
    while (true)
    {
        __syncthreads();
        if (tid == 0)
        {
            uint32_t t = atomicAdd(&A.cursors[qcur], 1u), p = 0xFFFFFFFFu;
            for (int cls = cls_hi; cls >= cls_lo; --cls)
            {
                const uint32_t cnt = A.cursors[cbase + cls];
                if (t < cnt) { p = order[(size_t)cls * n_pairs + t]; break; }
                t -= cnt;
            }
            W.misc[7] = p;
        }
        __syncthreads();
        const uint32_t p = W.misc[7];
        if (p >= n_pairs) break;
        PairRec rec = pairs[p];
        if (rec.cv_n == 0 || rec.status != 0) continue;       // empty Convex: the Mesh is not clipped (:1467-1468)
        if (rec.img_fmt == IMG_EMPTY) continue;
        const uint32_t cell = pair_list ? pair_list[p].x : cell_begin + p / P.n;
        const uint32_t piece = pair_list ? pair_list[p].y : p % P.n;
        const uint32_t f0 = plane_off[cell], F = plane_off[cell + 1] - f0;
        int err = WC_BAIL;
        if ((rec.img_fmt == IMG_NARROW || rec.img_fmt == IMG_REC) && P.mdup[piece] == 0 && F <= WC_MAXF)
        {
            for (uint32_t k = tid; k < F; k += group_size()) W.planes[k] = planes[f0 + k];
            const uint32_t m0 = P.mvo[piece], V = P.mvo[piece + 1] - m0;
            const uint32_t nbV = (V + SURTR_LANES - 1u) >> SURTR_LSH;
            char* img = IA.base + (size_t)rec.img_off * 16u;
            unsigned long long zmask = 0ull;
            WcOut o{0u, 0u, 0u, 0u};
            WcCtr ctr{0u, 0u};
#ifdef SURTR_STAMP
            if (tid == 0) for (int q = 0; q < 32; ++q) W.ph[q] = 0ull;
#endif
            bool fits = false;
            WcGlob g;
            if (rec.img_fmt == IMG_REC)
            {
                // the band is there as the record clipper streams it (k_prep_pairs, prep_sorted.h): used, and patched, in place; this
                // workgroup's scratch slot only holds the positions of the cut points
                const RecLayout rl = rec_layout(F, rec.img_n);
                g.grec = (WcW4*)(img + rl.grec); g.gpos = (float4*)(img + rl.gpos); g.cpos = (float4*)slot;
                fits = 16u * (size_t)(2u * WL::kNR) <= pool.per_wg;
                if (fits) err = wc_attach(W, (const uint32_t*)(img + rl.hist), (const uint32_t*)(img + rl.zhist), (const uint32_t*)(img + rl.bst), F, rec.img_n, zmask, ctr, A.cursors + 96);
            }
            else if constexpr (OLD_IMAGES)
            {
                const ImgLayout lay = img_layout(F, nbV, rec.img_n, rec.img_h);
                const WcImg im{(const uint16_t*)(img + lay.loff), (const uint8_t*)(img + lay.llen), (const uint8_t*)(img + lay.comp),
                               (const uint16_t*)(img + lay.ring), (const float*)(img + lay.pos), (const uint32_t*)(img + lay.hist),
                               (const uint32_t*)(img + lay.zhist), (const uint32_t*)(img + lay.nzero), rec.img_n, rec.img_h};
                g = wc_glob(slot, pool.per_wg, rec.img_n, 2u * WL::kNR, fits);
                if (fits) err = wc_load(W, im, F, g, zmask, ctr, A.cursors + 96);
            }
            if (err == 0) err = wc_planes(W, F, rec.img_n, V - rec.img_n, zmask, g, 2u * WL::kNR, o, ctr, A.cursors + 96, walk0);
            if (err == 0 && o.nLive != 0u) err = wc_park(W, F, rec.img_n, o, g, A, rec, ctr, A.cursors + 96);
#ifdef SURTR_STAMP
            __syncthreads();
            if (tid == 0)
            {
                for (int q = 0; q < 29; ++q) if (q != 22 && q != 23 && W.ph[q]) atomicAdd(&g_wstamp[q], W.ph[q]);
                // LDS the pair needed at its worst plane, and at its worst plane from the third on (classes of 4 KiB)
                if (p < 8192u)
                {
                    uint32_t mb = 0; for (uint32_t k = 0; k < F; ++k) { const uint32_t c = W.bst[k + 1u] - W.bst[k]; mb = c > mb ? c : mb; }
                    g_wneed[4u * p] = rec.img_n; g_wneed[4u * p + 1u] = mb; g_wneed[4u * p + 2u] = err == 0 ? (uint32_t)W.ph[31] : 0xFFFFFFFFu;
                    unsigned long long cyc = 0; for (int q = 0; q < 16; ++q) cyc += W.ph[q];
                    g_wneed[4u * p + 3u] = (uint32_t)cyc;
                }
                if (err == 0)
                {
                    unsigned long long c = W.ph[31] / 4096ull, c2 = W.ph[29] / 4096ull;
                    atomicAdd(&g_wstamp[32 + (int)(c > 15ull ? 15ull : c)], 1ull);
                    atomicAdd(&g_wstamp[48 + (int)(c2 > 15ull ? 15ull : c2)], 1ull);
                }
            }
#endif
            // a record image the clipper gave up on has been patched by the planes it did (and the general clipper has no use for
            // records anyway): it starts from the piece (its own pre-pass), as for a pair that never had an image
            if (err == WC_BAIL && rec.img_fmt == IMG_REC) { rec.img_fmt = IMG_NONE; if (tid == 0) atomicAdd(&A.cursors[93], 1u); }
        }
        if (tid == 0) atomicAdd(&A.cursors[err == WC_BAIL ? 89 : 88], 1u);       // (diagnostic: pairs the record clipper took / handed on)
        if (err == WC_BAIL)
        {
            if constexpr (FALLBACK)
            {
                __syncthreads();
                clip_pair_general(Gn.sh, Gn.L, S, pool, wg, P, planes, plane_off, cell_begin, A, IA, pairs, pair_list, horder, p, rec);
            }
            else if (tid == 0)
            {
                pairs[p] = rec;          // (a record image is spent: rec.img_fmt is IMG_NONE by now; an old image is as it was)
                if (hlist != nullptr)
                {
                    // to the catcher that runs beside this kernel (k_clip_pairs_catch): a slot of its list, filled with one atomic
                    // so that whoever polls the slot sees either nothing or the pair
                    __threadfence();
                    atomicExch(&hlist[atomicAdd(&A.cursors[146], 1u)], p);
                }
                else
                {
                    // to the large tier, first in its queue (class 12), whose kernel is launched behind this one
                    horder[(size_t)12 * n_pairs + atomicAdd(&A.cursors[16u + 12u], 1u)] = p;      // (horder: the large tier's table here)
                    atomicAdd(&A.cursors[146], 1u);
                }
            }
            continue;
        }
        if (err == 0 && rec.cv_bad != 0 && rec.ni != 0) err = SURTR_E_TOPOLOGY;       // a fragment with an invalid Convex
        if (err != 0) { rec.status = (uint32_t)err; rec.ni = 0; if (tid == 0) pair_failed(A, err); }
        if (tid == 0) pairs[p] = rec;
    }
    // the catcher stops polling when every workgroup of this kernel has said so (after its last hand-over)
    if (!FALLBACK && hlist != nullptr && tid == 0) { __threadfence(); atomicAdd(&A.cursors[148], 1u); }
}

__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_clip_pairs_wave(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ horder, int cls_hi, int cls_lo, uint32_t qcur, uint32_t walk0)
{
    constexpr size_t kBytes = sizeof(WcLds) > sizeof(GenLds) ? sizeof(WcLds) : sizeof(GenLds);
    __shared__ alignas(16) unsigned char lds_raw[kBytes];
    clip_pairs_wave_body<WcLds, GenLds>(lds_raw, blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, pairs, pair_list, order, horder, cls_hi, cls_lo, qcur, walk0);
}

// The small tier: the record images whose worst plane fits SURTR_WR_S units of LDS (k_prep_pairs predicts it from the band size and
// the largest bucket), three workgroups per CU -- 12 waves per CU instead of 8 for a kernel that is bound by instruction issue and
// dependent LDS round trips (measured in round 3 on a timing-only build: -17 % for the pairs it takes).  No general clipper
// inside: 0 bytes of private scratch, the registers of the record clipper alone.
#ifndef SURTR_WR_S
#define SURTR_WR_S 2240u
#define SURTR_WNL_S 2048u
#endif
#ifndef SURTR_S_THREADS
#define SURTR_S_THREADS SURTR_WG
#endif
#ifndef SURTR_S_WAVES_EU
#define SURTR_S_WAVES_EU 3
#endif
typedef WcLdsT<SURTR_WR_S, SURTR_WNL_S, (SURTR_S_THREADS / SURTR_LANES > SURTR_NWAVE ? SURTR_S_THREADS / SURTR_LANES : SURTR_NWAVE)> WcLdsS;
__global__ __launch_bounds__(SURTR_S_THREADS) __attribute__((amdgpu_waves_per_eu(SURTR_S_WAVES_EU, 4))) void k_clip_pairs_rec(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ sorder,
                                                         uint32_t* __restrict__ order, uint32_t walk0)
{
    __shared__ alignas(16) unsigned char lds_raw[sizeof(WcLdsS)];
    clip_pairs_wave_body<WcLdsS, WcLdsS, false>(lds_raw, blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, pairs, pair_list, sorder, order, 11, 0, 144u, walk0);
}

// ---- The split arrangement (round 4): the regular pairs on a kernel that holds the record clipper ALONE, everything else beside it.
// k_clip_pairs_wave carries the general clipper for the 2 % of its pairs that need it, and pays for it on every pair: 256 registers,
// 1.9 KB of private scratch per lane.  The same pairs on the record clipper alone (114 registers, no scratch, the same LDS and
// occupancy) take 1.12 instead of 1.40 ms on configs[3], 1.01 ms with 512 threads per pair (the planes of a heavy band hold a
// thousand items and more).  So: k_clip_pairs_main takes cost classes 11..0 (record images and old images alike) and hands what
// it cannot finish to k_clip_pairs_catch -- the general clipper on a few workgroups that run BESIDE it on another stream: they
// start with the pairs k_prep_pairs knows to be irregular (classes 13..12), then poll the hand-over list until every workgroup
// of the main kernel has signed off.  No later launch, no tail of its own.
#ifndef SURTR_MAIN_THREADS
#define SURTR_MAIN_THREADS (2u * SURTR_WG)
#endif
#ifndef SURTR_CATCH_POLL
#define SURTR_CATCH_POLL 8u       // workgroups of k_clip_pairs_catch that wait for hand-overs
#endif
#ifndef SURTR_WR_MAIN
#define SURTR_WR_MAIN 3072u       // (a plane that needs more stages its originals in global memory: the lists get the room instead)
#define SURTR_WNL_MAIN 3584u
#endif
typedef WcLdsT<SURTR_WR_MAIN, SURTR_WNL_MAIN, (SURTR_MAIN_THREADS / SURTR_LANES > SURTR_NWAVE ? SURTR_MAIN_THREADS / SURTR_LANES : SURTR_NWAVE), true> WcLdsMain;
// (two workgroups of 512 threads per CU = four waves per SIMD: at most 128 registers)
__global__ __launch_bounds__(SURTR_MAIN_THREADS) __attribute__((amdgpu_waves_per_eu(SURTR_MAIN_THREADS > SURTR_WG ? 4 : 2, 4))) void k_clip_pairs_main(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ hlist, uint32_t walk0)
{
    __shared__ alignas(16) unsigned char lds_raw[sizeof(WcLdsMain)];
    clip_pairs_wave_body<WcLdsMain, WcLdsMain, false, true>(lds_raw, blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, pairs, pair_list, order, nullptr, 11, 0, 4u, walk0, hlist, 16u);
}

// The catcher: see above.  n_main: workgroups of k_clip_pairs_main.  Its workgroups use the scratch slots from wg_base on.
// Nothing depends on the two kernels really running side by side: a polling workgroup marks the slot it takes (SURTR_H_TAKEN), leaves
// when the main kernel has not even started after a few thousand polls (a profiler that serialises kernels, a device with no room
// for both) or has not finished after a very long wait, and a SWEEP launch of this kernel (sweep != 0) behind both takes whatever
// slots still hold a pair -- as a rule none: one workgroup looks at the counters and returns.
#define SURTR_H_TAKEN 0xFFFFFFFEu
__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(2, 4))) void k_clip_pairs_catch(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, uint32_t wg_base, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         uint32_t* __restrict__ hlist, uint32_t hcap, uint32_t n_main, uint32_t sweep, uint32_t n_poll)
{
    __shared__ Shared sh;
    __shared__ LdsTopo L;
    const uint32_t tid = threadIdx.x, wg = wg_base + blockIdx.x;
    Scratch S = carve(pool, wg);
    bool own = sweep == 0u;
    while (true)
    {
        __syncthreads();
        if (tid == 0)
        {
            uint32_t p = 0xFFFFFFFFu;
            if (sweep != 0u)
            {
                // behind both kernels: the slots that were filled and never taken
                const uint32_t pushed = A.cursors[146] < hcap ? A.cursors[146] : hcap;
                for (uint32_t t = atomicAdd(&A.cursors[151], 1u); t < pushed; t = atomicAdd(&A.cursors[151], 1u))
                {
                    const uint32_t v = atomicExch(&hlist[t], SURTR_H_TAKEN);
                    if (v < n_pairs) { p = v; break; }
                }
            }
            else if (own)
            {
                uint32_t t = atomicAdd(&A.cursors[150], 1u);
                for (int cls = 13; cls >= 12; --cls)
                {
                    const uint32_t cnt = A.cursors[16 + cls];
                    if (t < cnt) { p = order[(size_t)cls * n_pairs + t]; break; }
                    t -= cnt;
                }
            }
            if (sweep == 0u && p == 0xFFFFFFFFu && n_poll != 0u && (own ? blockIdx.x : 0u) < n_poll)
            {
                // (only the first few workgroups stay for the hand-overs -- a handful per event; the others give their LDS back to the
                //  main kernel as soon as the irregular pairs are done)
                own = false;
                const uint32_t t = atomicAdd(&A.cursors[149], 1u);          // this workgroup's slot of the hand-over list
                for (uint32_t spin = 0; t < hcap; ++spin)
                {
                    const uint32_t v = atomicAdd(&hlist[t], 0u);
                    if (v < n_pairs) { atomicExch(&hlist[t], SURTR_H_TAKEN); p = v; break; }
                    if (atomicAdd(&A.cursors[148], 0u) >= n_main && atomicAdd(&A.cursors[146], 0u) <= t) break;      // nobody will fill it
                    if (spin > (1u << 12) && atomicAdd(&A.cursors[147], 0u) == 0u) break;      // the main kernel is not running beside this one: the sweep's
                    if (spin > (1u << 22)) break;                                              // ... or for very long: the sweep's as well
                    __builtin_amdgcn_s_sleep(32);
                }
            }
            sh.misc[7] = p;
            sh.misc[6] = own ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t p = sh.misc[7];
        own = sh.misc[6] != 0u;
        if (p >= n_pairs) break;
        PairRec rec = pairs[p];
        if (rec.cv_n == 0 || rec.status != 0) continue;
        if (rec.img_fmt == IMG_EMPTY) continue;
        if (rec.img_fmt == IMG_REC) rec.img_fmt = IMG_NONE;
        if (tid == 0) atomicAdd(&A.cursors[94], 1u);      // (diagnostic, surtr_queue_stats: pairs the catcher clipped)
        clip_pair_general(sh, L, S, pool, wg, P, planes, plane_off, cell_begin, A, IA, pairs, pair_list, (uint32_t*)nullptr, p, rec);
    }
}

// The same with a whole CU's LDS for the record clipper (and the double-size topology for the general one it falls back to): the
// bands that leave the regular kernels no room (cost classes 15..14, and 13 on meshes where such bands are the rule).  One
// workgroup per CU; its workgroups use the scratch slots after those of the regular kernel.
__global__ __launch_bounds__(SURTR_WG) void k_clip_pairs_wave_big(Pieces P, const float4* __restrict__ planes,
                                                         const uint32_t* __restrict__ plane_off, uint32_t cell_begin,
                                                         uint32_t n_pairs,
                                                         ScratchPool pool, uint32_t wg_base, Arena A, ImgArena IA, PairRec* __restrict__ pairs,
                                                         const uint2* __restrict__ pair_list, const uint32_t* __restrict__ order,
                                                         int cls_hi, int cls_lo, uint32_t qcur, uint32_t walk0)
{
    constexpr size_t kBytes = sizeof(WcLdsBig) > sizeof(GenLdsBig) ? sizeof(WcLdsBig) : sizeof(GenLdsBig);
    __shared__ alignas(16) unsigned char lds_raw[kBytes];
    clip_pairs_wave_body<WcLdsBig, GenLdsBig>(lds_raw, wg_base + blockIdx.x, P, planes, plane_off, cell_begin, n_pairs, pool, A, IA, pairs, pair_list, order, nullptr, cls_hi, cls_lo, qcur, walk0);
}

// -------------------------------------------------------------- k_frag_table
// One workgroup: exclusive scan of islands per pair -> fragment records in cell-major order.
__global__ __launch_bounds__(SURTR_WG_WIDE) void k_frag_table(const PairRec* __restrict__ pairs, uint32_t n_pairs,
                                                         uint32_t n_pieces, uint32_t cell_begin, Arena A,
                                                         uint2* __restrict__ blk, FragRec* __restrict__ frags,
                                                         uint32_t cap_frags, surtr_counts* __restrict__ counts,
                                                         const uint2* __restrict__ pair_list, uint32_t* __restrict__ forder)
{
    __shared__ Shared sh;
    for (uint32_t q = threadIdx.x; q < 16u; q += group_size()) sh.hist[q] = 0;
    auto fn = [&](uint32_t p) -> uint2 { return make_uint2(pairs[p].ni, 0u); };
    uint32_t nf = 0, dum = 0;
    scan_blocks(n_pairs, blk, sh, fn, nf, dum);
    const uint32_t nb = (n_pairs + SURTR_LANES - 1u) >> SURTR_LSH;
    if (nf <= cap_frags)
    {
        for (uint32_t b = wave_id(); b < nb; b += group_waves())
        {
            const uint32_t p = (b << SURTR_LSH) + lane_id();
            uint2 c = make_uint2(0u, 0u);
            if (p < n_pairs) c = fn(p);
            const uint2 e = wave_excl2(c);
            if (p < n_pairs && c.x)
            {
                const PairRec r = pairs[p];
                uint32_t f = blk[b].x + e.x;
                uint32_t vo = r.mv_off, ho = r.mh_off;
                for (uint32_t t = 0; t < r.ni; ++t, ++f)
                {
                    const uint2 is = A.isl[r.isl_off + t];
                    FragRec fr;
                    fr.cell = (int32_t)(pair_list ? pair_list[p].x : cell_begin + p / n_pieces);
                    fr.piece = (int32_t)(pair_list ? pair_list[p].y : p % n_pieces); fr.island = (int32_t)t;
                    fr.mv_off = vo; fr.mv_n = is.x; fr.mh_off = ho; fr.mh_n = is.y;
                    fr.cv_off = r.cv_off; fr.cv_n = r.cv_n; fr.ch_off = r.ch_off; fr.ch_n = r.ch_n;
                    fr.idx_off = 0; fr.idx_n = 0; fr.o_mv = fr.o_mh = fr.o_cv = fr.o_ch = fr.o_idx = 0;
                    frags[f] = fr;
                    // size class (log2 of the vertex count): k_refit / k_faces start with the large fragments
                    uint32_t cls = 0; while ((is.x >> (cls + 1u)) != 0u && cls < 15u) ++cls;
                    forder[(size_t)cls * cap_frags + atomicAdd(&sh.hist[cls], 1u)] = f;      // (one workgroup: LDS counters)
                    vo += is.x; ho += is.y;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < 16u; q += group_size()) A.cursors[32u + q] = sh.hist[q];
    if (threadIdx.x == 0)
    {
        counts->n_frag = nf <= cap_frags ? nf : 0u;
        counts->n_pairs = n_pairs;
        if (nf > cap_frags) atomicMax(&A.cursors[5], (uint32_t)SURTR_E_CAPACITY);
    }
}

// next ticket of a per-fragment kernel -> fragment, large size classes first (k_frag_table filled forder)
__device__ static uint32_t frag_of_ticket(const Arena& A, const uint32_t* __restrict__ forder, uint32_t cap_frags, uint32_t t)
{
    for (int cls = 15; cls >= 0; --cls)
    {
        const uint32_t cnt = A.cursors[32 + cls];
        if (t < cnt) return forder[(size_t)cls * cap_frags + t];
        t -= cnt;
    }
    return 0xFFFFFFFFu;
}

// ------------------------------------------------------------------ k_refit
// argmax with "first maximum wins" (std::max_element) over the workgroup.
struct ArgF { float v; uint32_t i; };
struct ArgD { double v; uint32_t i; };

template <class T, class A>
__device__ static A wg_argmax(A mine, A* slots /* shared, SURTR_NWAVE */)
{
    // wave reduce ("larger value, then smaller index" is associative and commutative): the scan pattern of wave_incl_scan2
    // with that operator leaves the result in the last lane; lanes without a source receive the invalid index
    auto step = [&](auto mv) {
        A o = mv(mine);
        if (o.i != 0xFFFFFFFFu && (mine.i == 0xFFFFFFFFu || o.v > mine.v || (o.v == mine.v && o.i < mine.i))) mine = o;
    };
    auto mover = [](auto ctrl_tag, auto mask_tag) {
        return [](A a) {
            constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
            A o;
            o.i = dpp_move<CTRL, MASK>(0xFFFFFFFFu, a.i);
            if constexpr (sizeof(T) == 4)
            {
                uint32_t b; memcpy(&b, &a.v, 4);
                b = dpp_move<CTRL, MASK>(0u, b);
                memcpy(&o.v, &b, 4);
            }
            else
            {
                uint32_t b[2]; memcpy(b, &a.v, 8);
                b[0] = dpp_move<CTRL, MASK>(0u, b[0]); b[1] = dpp_move<CTRL, MASK>(0u, b[1]);
                memcpy(&o.v, b, 8);
            }
            return o;
        };
    };
    step(mover(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{}));
    step(mover(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{}));
    step(mover(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{}));
    step(mover(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{}));
    step(mover(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}));
    step(mover(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{}));
    __syncthreads();
    if (lane_id() == SURTR_LANES - 1u) slots[wave_id()] = mine;
    __syncthreads();
    A best = slots[0];
    for (uint32_t q = 1; q < group_waves(); ++q)
    {
        const A o = slots[q];
        if (o.i != 0xFFFFFFFFu && (best.i == 0xFFFFFFFFu || o.v > best.v || (o.v == best.v && o.i < best.i))) best = o;
    }
    __syncthreads();
    return best;
}

__device__ __forceinline__ float hull_vol(const float* a, const float* b, const float* c, const float* p)
{
    // VMACH::ConvexHull::Volume, Src/VMACH.cpp:922-939
    const float ax = a[0] - p[0], ay = a[1] - p[1], az = a[2] - p[2];
    const float bx = b[0] - p[0], by = b[1] - p[1], bz = b[2] - p[2];
    const float cx = c[0] - p[0], cy = c[1] - p[1], cz = c[2] - p[2];
    return ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
}

__global__ __launch_bounds__(SURTR_LANES) void k_refit(FragRec* __restrict__ frags, const surtr_counts* __restrict__ counts,
                                                    ScratchPool pool, Arena A, const uint32_t* __restrict__ forder, uint32_t cap_frags,
                                                    uint32_t* __restrict__ frag_status, const float* __restrict__ piece_cpos,
                                                    const uint32_t* __restrict__ piece_cvo)
{
    __shared__ Shared sh;
    __shared__ ArgF slotF[SURTR_NWAVE];
    __shared__ ArgD slotD[SURTR_NWAVE];
    __shared__ float nrm[4][3];
    __shared__ OneWaveLds U;      // (the one-wave clipper of regular planes shares the bytes of the general one's arrays)
    LdsTopoSmall& L = U.g.L; LdsWorkSmall& W = U.g.W;
    Scratch S = carve(pool, blockIdx.x);
    const uint32_t tid = threadIdx.x;
    const uint32_t nf = counts->n_frag;
#ifdef SURTR_STAMP
    const unsigned long long wg_t0 = __builtin_readcyclecounter();
    unsigned long long wg_tasks = 0;
#endif
    while (true)
    {
        __syncthreads();
        if (tid == 0) sh.misc[7] = frag_of_ticket(A, forder, cap_frags, atomicAdd(&A.cursors[6], 1u));
        __syncthreads();
        const uint32_t f = sh.misc[7];
        if (f >= nf) break;
        FragRec fr = frags[f];
        const float* mpg = A.pos + 3 * (size_t)fr.mv_off;
        const uint32_t n = fr.mv_n;
#ifdef SURTR_STAMP
        const unsigned long long r0 = __builtin_readcyclecounter();
#endif
        // The fragment's vertices are read a dozen times (four hull passes, eight slab extremes): a fragment of up to
        // 2 * LdsWorkSmall::kN vertices is staged once in the LDS work arrays, which nothing uses before the clip below.
#ifdef SURTR_STAMP
        unsigned long long r1 = 0;
#endif
        static_assert(offsetof(LdsWorkSmall, aux0) == sizeof(float) * 3 * LdsWorkSmall::kN && offsetof(LdsWorkSmall, aux1) == sizeof(float) * 4 * LdsWorkSmall::kN &&
                      offsetof(LdsWorkSmall, aux2) == sizeof(float) * 5 * LdsWorkSmall::kN, "pos, aux0, aux1, aux2 are contiguous");
        auto slabs = [&](const auto* mp) {
        // ---- BuildFirstHull (Src/VMACH.cpp:1036-1085) with limit min(n,4) = 4 ----
        ArgF a; a.i = 0xFFFFFFFFu; a.v = 0.f;
        for (uint32_t v = tid; v < n; v += group_size())
        {
            const float x = mp[3 * v];
            if (a.i == 0xFFFFFFFFu || x > a.v) { a.v = x; a.i = v; }
        }
        a = wg_argmax<float, ArgF>(a, slotF);
        const uint32_t i1 = a.i;
        const float p1x = mp[3 * i1], p1y = mp[3 * i1 + 1], p1z = mp[3 * i1 + 2];
        ArgD d; d.i = 0xFFFFFFFFu; d.v = 0.0;
        for (uint32_t v = tid; v < n; v += group_size())
        {
            const double dx = (double)(mp[3 * v] - p1x), dy = (double)(mp[3 * v + 1] - p1y), dz = (double)(mp[3 * v + 2] - p1z);
            const double dist = sqrt(dx * dx + dy * dy + dz * dz);
            if (d.i == 0xFFFFFFFFu || dist > d.v) { d.v = dist; d.i = v; }
        }
        d = wg_argmax<double, ArgD>(d, slotD);
        const uint32_t i2 = d.i;
        const float p2x = mp[3 * i2], p2y = mp[3 * i2 + 1], p2z = mp[3 * i2 + 2];
        a.i = 0xFFFFFFFFu; a.v = 0.f;
        for (uint32_t v = tid; v < n; v += group_size())
        {
            // ConvexHullFace(v1, v2, p).CalcArea(): 0.5 * |(v2-v1) x (p-v1)|
            const float ux = p2x - p1x, uy = p2y - p1y, uz = p2z - p1z;
            const float wx = mp[3 * v] - p1x, wy = mp[3 * v + 1] - p1y, wz = mp[3 * v + 2] - p1z;
            const float cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
            const float area = 0.5f * sqrtf(dot3(cx, cy, cz, cx, cy, cz));
            if (a.i == 0xFFFFFFFFu || area > a.v) { a.v = area; a.i = v; }
        }
        a = wg_argmax<float, ArgF>(a, slotF);
        const uint32_t i3 = a.i;
        const float q1[3] = {p1x, p1y, p1z}, q2[3] = {p2x, p2y, p2z};
        const float q3[3] = {mp[3 * i3], mp[3 * i3 + 1], mp[3 * i3 + 2]};
        a.i = 0xFFFFFFFFu; a.v = 0.f;
        for (uint32_t v = tid; v < n; v += group_size())
        {
            const float vol = hull_vol(q1, q2, q3, mp + 3 * v);
            if (a.i == 0xFFFFFFFFu || vol > a.v) { a.v = vol; a.i = v; }
        }
        a = wg_argmax<float, ArgF>(a, slotF);
        const uint32_t i4 = a.i;
        if (tid == 0)
        {
            const float q4[3] = {mp[3 * i4], mp[3 * i4 + 1], mp[3 * i4 + 2]};
            // faces in creation order, each rewound when Volume(face, inner) < 0 (:955-969)
            const float* fv[4][3] = {{q1, q2, q3}, {q1, q2, q4}, {q1, q3, q4}, {q2, q3, q4}};
            const float* inner[4] = {q4, q3, q2, q1};
            for (int k = 0; k < 4; ++k)
            {
                const float* v0 = fv[k][0]; const float* v1 = fv[k][1]; const float* v2 = fv[k][2];
                if (hull_vol(v0, v1, v2, inner[k]) < 0.f) { const float* t = v0; v0 = v2; v2 = t; }
                // GenerateICHNormal (Src/Surtr.cpp:1961-1974): normalize((v1-v0) x (v2-v0))
                const float ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2];
                const float bx = v2[0] - v0[0], by = v2[1] - v0[1], bz = v2[2] - v0[2];
                float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
                const float len = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
                if (len != 0.f) { nx = nx / len; ny = ny / len; nz = nz / len; } else { nx = ny = nz = 0.f; }
                nrm[k][0] = nx; nrm[k][1] = ny; nrm[k][2] = nz;
            }
        }
        __syncthreads();
#ifdef SURTR_STAMP
        r1 = __builtin_readcyclecounter();
#endif
        // ---- Kdop::Calc(Polyhedron) (Src/Kdop.cpp:92-115): first minimum / first maximum of n.v ----
        for (int k = 0; k < 4; ++k)
        {
            ArgF lo, hi; lo.i = hi.i = 0xFFFFFFFFu; lo.v = hi.v = 0.f;
            const float nx = nrm[k][0], ny = nrm[k][1], nz = nrm[k][2];
            for (uint32_t v = tid; v < n; v += group_size())
            {
                const float t = dot3(mp[3 * v], mp[3 * v + 1], mp[3 * v + 2], nx, ny, nz);
                if (hi.i == 0xFFFFFFFFu || t > hi.v) { hi.v = t; hi.i = v; }
                if (lo.i == 0xFFFFFFFFu || -t > lo.v) { lo.v = -t; lo.i = v; }
            }
            hi = wg_argmax<float, ArgF>(hi, slotF);
            lo = wg_argmax<float, ArgF>(lo, slotF);
            if (tid == 0)
            {
                // MinPlane = Plane(vert, -n), MaxPlane = Plane(vert, n); order Min, Max (:166-179)
                const float* a0 = mp + 3 * lo.i; const float* a1 = mp + 3 * hi.i;
                sh.planes[2 * k] = make_float4(-nx, -ny, -nz, -dot3(a0[0], a0[1], a0[2], -nx, -ny, -nz));
                sh.planes[2 * k + 1] = make_float4(nx, ny, nz, -dot3(a1[0], a1[1], a1[2], nx, ny, nz));
            }
        }
        __syncthreads();
        };
        if (n <= 2u * LdsWorkSmall::kN)
        {
            float* lp = W.pos;
            for (uint32_t i = tid; i < 3u * n; i += group_size()) lp[i] = mpg[i];
            __syncthreads();
            slabs((const float*)lp);
            __syncthreads();
        }
        else slabs(mpg);
#ifdef SURTR_STAMP
        const unsigned long long r2 = __builtin_readcyclecounter();
        if (tid == 0) { atomicAdd(&g_stamp[94], r1 - r0); atomicAdd(&g_stamp[95], r2 - r1); }
        int dbg_path = 0;
#endif
        SolidIn cin{A.pos + 3 * (size_t)fr.cv_off, A.loff + fr.cv_off, A.llen + fr.cv_off, A.nbr, fr.cv_n, nullptr, nullptr, nullptr, nullptr, nullptr};
        // arena rings are absolute offsets into A.nbr, which is what SolidIn expects
        uint32_t nvoff = 0, ncn = 0, nhoff = 0, nchn = 0;
        // (no small_clip attempt here: the slab planes pass through extreme vertices of the fragment, which are vertices of its
        // Convex as often as not -- measured on configs[3]: 2 385 of 2 692 refits have a vertex exactly in a plane)
        int err = SURTR_E_TOPOLOGY;      // a sliver Convex: the literal clipper below, from the start
        if (!solid_is_sliver(cin))
        {
            // the regular clipper first: it takes the Convex whose slab planes either cut no vertex (the plane through an extreme
            // vertex of the fragment that is an extreme vertex of its Convex too) or cut with no vertex in the plane
            uint32_t which = 0, stop = 0xFFFFFFFFu;
            err = small_clip(cin, 8, sh, U.f, &which, &stop);
            if (tid == 0) atomicAdd(&A.cursors[err == 0 ? 82 : 83], 1u);       // (diagnostic: refits the regular clipper took / handed on)
            if (err == 0)
            {
                const uint32_t nv = U.f.nv[which];
                if (nv != 0u) err = sc_park(U.f.buf[which], nv, sh, A.cursors, A.pos, A.loff, A.llen, A.nbr, A.capV, A.capH, nvoff, ncn, nhoff, nchn);
            }
            __syncthreads();
            if (err == SC_FALLBACK)
            {
                // (as in k_clip_convex: the general clipper goes on from the plane the regular one stopped at)
                SolidIn cx = cin; uint32_t Fx = 8u;
                const bool resumed = SC_RESUME && stop != 0xFFFFFFFFu && stop > 0u && stop < 8u;
                float4* const keep = (float4*)S.g_comp;      // (a byte array of the wide topology: 128 bytes of it hold the eight planes meanwhile)
                if (resumed)
                {
                    const uint32_t nvs = U.f.nv[which];
                    cx = sc_stage(U.f.buf[which], nvs, S.pos, S.g_loff, S.g_llen, S.g_ring);
                    Fx = 8u - stop;
                    for (uint32_t k = tid; k < 8u; k += group_size()) { keep[k] = sh.planes[k]; sh.pmar[k] = sh.planes[k]; }
                    __syncthreads();
                    for (uint32_t k = tid; k < Fx; k += group_size()) sh.planes[k] = sh.pmar[k + stop];
                    __syncthreads();
                    if (tid == 0) atomicAdd(&A.cursors[79], 1u);       // (diagnostic)
                }
                err = clip_any<false>(cx, Fx, S, sh, L, [&](auto& T) -> int {
                    if (T.nLive == 0) return 0;
                    return park_topo(T, sh, A, nvoff, ncn, nhoff, nchn);
                }, &W);
                __syncthreads();
                if (resumed)
                {
                    for (uint32_t k = tid; k < 8u; k += group_size()) sh.planes[k] = keep[k];      // (the fall-backs below start over with all eight)
                    __syncthreads();
                }
            }
            __syncthreads();
#ifdef SURTR_STAMP
            if (err == SURTR_OVERFLOW) dbg_path |= 1;
            if (err == SURTR_E_TOPOLOGY) dbg_path |= 2;
#endif
            if (err == SURTR_OVERFLOW)
            {
                const ParkOut o = solid_global(cin, 8, pool, blockIdx.x, A, &sh);
                err = o.err; nvoff = o.voff; ncn = o.n; nhoff = o.hoff; nchn = o.nh;
                __syncthreads();
            }
        }
#ifdef SURTR_STAMP
        if (err == SURTR_E_TOPOLOGY) dbg_path |= 4;
#endif
        if (err == SURTR_E_TOPOLOGY)
        {
            // The Convex is the result of the pair's clip: its vertices carry the IDs of that clip's last compaction, their own
            // indices -- unless no cell plane cut the piece's Convex, which then is a copy of the piece's with the IDs it came
            // with (-1: built from arrays).  A cut changes the vertex set, so "uncut" = same vertices as the piece's Convex.
            bool ids_set = piece_cpos != nullptr;
            if (piece_cpos != nullptr)
            {
                const uint32_t c0 = piece_cvo[fr.piece], pn = piece_cvo[fr.piece + 1] - c0;
                bool differs = pn != fr.cv_n;
                if (!differs)
                    for (uint32_t i = tid; i < 3u * pn; i += group_size()) if (piece_cpos[3 * (size_t)c0 + i] != cin.pos[i]) differs = true;
                ids_set = __syncthreads_or(differs ? 1 : 0) != 0;
            }
            const ParkOut o = solid_literal(cin, 8, pool, blockIdx.x, A, &sh, ids_set);
            if (o.err == 0) { err = 0; nvoff = o.voff; ncn = o.n; nhoff = o.hoff; nchn = o.nh; }
            else if (o.err != SURTR_E_TOPOLOGY) err = o.err;
            __syncthreads();
        }
        if (err == 0 && tid == 0)
        {
            // field-wise: k_faces updates other fields of the same record at the same time
            frags[f].cv_off = nvoff; frags[f].cv_n = ncn; frags[f].ch_off = nhoff; frags[f].ch_n = nchn;
        }
        if (err == SURTR_E_TOPOLOGY && frag_status != nullptr)
        {
            // Where the reference's own clip of this Convex by the slabs is no polyhedron any more (a link to a clipped vertex
            // that survives, renumbered through a stale ID), the fragment keeps the Convex it had -- a superset of the refitted
            // one -- and is flagged; the event and the other fragments stand (as for a fragment without triangles in k_faces).
            if (tid == 0 && atomicExch(&frag_status[f], (uint32_t)SURTR_E_TOPOLOGY) == 0u) atomicAdd(&A.cursors[14], 1u);
        }
        else if (err != 0 && tid == 0) atomicMax(&A.cursors[5], (uint32_t)err);
#ifdef SURTR_STAMP
        if (tid == 0)
        {
            const unsigned long long d = __builtin_readcyclecounter() - r0; ++wg_tasks;
            int bkt = 0; while ((d >> bkt) > 1 && bkt < 40) ++bkt; bkt = bkt < 12 ? 0 : bkt - 12; if (bkt > 15) bkt = 15;
            atomicAdd(&g_stamp2[bkt], 1ull); atomicAdd(&g_stamp2[19], d);
            const unsigned long long old = atomicMax(&g_stamp2[16], d);
            if (d > old) { g_stamp2[17] = n; g_stamp2[18] = fr.cv_n; g_stamp2[60] = r1 - r0; g_stamp2[61] = r2 - r1; g_stamp2[62] = (unsigned long long)dbg_path; g_stamp2[63] = ncn; }
            if (dbg_path & 1) atomicAdd(&g_stamp2[30], 1ull);
            if (dbg_path & 6) atomicAdd(&g_stamp2[31], 1ull);
        }
#endif
    }
#ifdef SURTR_STAMP
    if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - wg_t0; atomicAdd(&g_stamp2[20], d); atomicMax(&g_stamp2[21], d); atomicAdd(&g_stamp2[22], 1ull); atomicMax(&g_stamp2[23], wg_tasks); }
#endif
}

// ------------------------------------------------------------------ k_faces

__device__ __forceinline__ bool on_right(const float* a, const float* b, const float* c, float nx, float ny, float nz)
{
    // VMACH::OnYourRight, Src/VMACH.cpp:1240-1243
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2];
    const float wx = c[0] - a[0], wy = c[1] - a[1], wz = c[2] - a[2];
    const float cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
    return dot3(cx, cy, cz, nx, ny, nz) > 0.f;
}

// Poly::EarClipping (Src/Poly.cpp:764-913) for one face, one lane.  loop = vertex ids of the face;
// tmp = 3*N ints (prev, next, reflex).  Writes vertex ids (3 per triangle) to out; returns the count.
// pos / loop may be global or LDS (k_faces stages small fragments): always inlined, so that the compiler sees which.
template <class LP>
__device__ __attribute__((always_inline)) static inline uint32_t ear_clip_face(const float* pos, const LP* loop, int N, int32_t* tmp, uint32_t* out)
{
    if (N <= 2) return 0;
    if (N == 3) { out[0] = loop[0]; out[1] = loop[1]; out[2] = loop[2]; return 3; }
    const float* A0 = pos + 3 * loop[0]; const float* B0 = pos + 3 * loop[1]; const float* C0 = pos + 3 * loop[2];
    float nx, ny, nz;
    {
        const float ux = B0[0] - A0[0], uy = B0[1] - A0[1], uz = B0[2] - A0[2];
        const float wx = C0[0] - A0[0], wy = C0[1] - A0[1], wz = C0[2] - A0[2];
        nx = uy * wz - uz * wy; ny = uz * wx - ux * wz; nz = ux * wy - uy * wx;
    }
    {   // IsCCW (:753-762)
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int v = 0; v < N; ++v)
        {
            const float* p = pos + 3 * loop[v]; const float* q = pos + 3 * loop[(v + 1) % N];
            const float ux = p[0] - A0[0], uy = p[1] - A0[1], uz = p[2] - A0[2];
            const float wx = q[0] - A0[0], wy = q[1] - A0[1], wz = q[2] - A0[2];
            sx = sx + (uy * wz - uz * wy); sy = sy + (uz * wx - ux * wz); sz = sz + (ux * wy - uy * wx);
        }
        if (dot3(sx, sy, sz, nx, ny, nz) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    }
    int32_t* prv = tmp; int32_t* nxt = tmp + N; int32_t* rfx = tmp + 2 * N;
    for (int i = 0; i < N; ++i) { prv[i] = (i + N - 1) % N; nxt[i] = (i + 1) % N; }
    for (int i = 0; i < N; ++i)
        rfx[i] = on_right(pos + 3 * loop[prv[i]], pos + 3 * loop[i], pos + 3 * loop[nxt[i]], nx, ny, nz) ? 0 : 1;
    int skipped = 0, left = N, cur = 0;
    uint32_t at = 0;
    while (left > 3)
    {
        const int p = prv[cur], n = nxt[cur];
        bool ear = rfx[cur] == 0;
        if (ear)
        {
            const float* a = pos + 3 * loop[p]; const float* b = pos + 3 * loop[cur]; const float* c = pos + 3 * loop[n];
            for (int r = 0; r < N && ear; ++r)
            {
                if (!rfx[r]) continue;                    // reflexVertices, in index order
                if (r == p || r == n) continue;
                const float* q = pos + 3 * loop[r];
                if ((q[0] == a[0] && q[1] == a[1] && q[2] == a[2]) || (q[0] == b[0] && q[1] == b[1] && q[2] == b[2])) continue;
                if (!on_right(a, b, q, nx, ny, nz)) continue;
                if (!on_right(b, c, q, nx, ny, nz)) continue;
                if (!on_right(c, a, q, nx, ny, nz)) continue;
                ear = false;
            }
        }
        if (ear)
        {
            out[at] = loop[p]; out[at + 1] = loop[cur]; out[at + 2] = loop[n];
            nxt[p] = n; prv[n] = p;
            const int adj[2] = {p, n};
            for (int k = 0; k < 2; ++k)
            {
                const int v = adj[k];
                if (!rfx[v]) continue;
                rfx[v] = on_right(pos + 3 * loop[prv[v]], pos + 3 * loop[v], pos + 3 * loop[nxt[v]], nx, ny, nz) ? 0 : 1;
            }
            at += 3; --left; skipped = 0;
        }
        else if (++skipped > left) return 0;             // stalled: the face is dropped (:899-903)
        cur = n;
    }
    out[at] = loop[prv[cur]]; out[at + 1] = loop[cur]; out[at + 2] = loop[nxt[cur]];
    return at + 3;
}

// The same for a face of at most 8 vertices (the quads, in practice: 5..64-gons go to the wave version) with prev / next /
// reflex packed into registers instead of a scratch array in global memory.  Same traversal, same triangles.
template <class LP>
__device__ __attribute__((always_inline)) static inline uint32_t ear_clip_small(const float* pos, const LP* loop, int N, uint32_t* out)
{
    if (N <= 2) return 0;
    if (N == 3) { out[0] = loop[0]; out[1] = loop[1]; out[2] = loop[2]; return 3; }
    auto P = [&](int i) -> const float* { return pos + 3 * (int)loop[i]; };
    const float* A0 = P(0); const float* B0 = P(1); const float* C0 = P(2);
    float nx, ny, nz;
    {
        const float ux = B0[0] - A0[0], uy = B0[1] - A0[1], uz = B0[2] - A0[2];
        const float wx = C0[0] - A0[0], wy = C0[1] - A0[1], wz = C0[2] - A0[2];
        nx = uy * wz - uz * wy; ny = uz * wx - ux * wz; nz = ux * wy - uy * wx;
    }
    {   // IsCCW (:753-762)
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int v = 0; v < N; ++v)
        {
            const float* p = P(v); const float* q = P((v + 1) % N);
            const float ux = p[0] - A0[0], uy = p[1] - A0[1], uz = p[2] - A0[2];
            const float wx = q[0] - A0[0], wy = q[1] - A0[1], wz = q[2] - A0[2];
            sx = sx + (uy * wz - uz * wy); sy = sy + (uz * wx - ux * wz); sz = sz + (ux * wy - uy * wx);
        }
        if (dot3(sx, sy, sz, nx, ny, nz) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    }
    uint32_t prv = 0, nxt = 0, rfx = 0;                        // 4 bits per vertex / 1 bit per vertex
    auto get = [](uint32_t w, int i) -> int { return (int)((w >> (4 * i)) & 15u); };
    auto put = [](uint32_t& w, int i, int v) { w = (w & ~(15u << (4 * i))) | ((uint32_t)v << (4 * i)); };
    for (int i = 0; i < N; ++i) { put(prv, i, (i + N - 1) % N); put(nxt, i, (i + 1) % N); }
    for (int i = 0; i < N; ++i)
        if (!on_right(P(get(prv, i)), P(i), P(get(nxt, i)), nx, ny, nz)) rfx |= 1u << i;
    int skipped = 0, left = N, cur = 0;
    uint32_t at = 0;
    while (left > 3)
    {
        const int p = get(prv, cur), n = get(nxt, cur);
        bool ear = ((rfx >> cur) & 1u) == 0u;
        if (ear)
        {
            const float* a = P(p); const float* b = P(cur); const float* c = P(n);
            for (int r = 0; r < N && ear; ++r)
            {
                if (!((rfx >> r) & 1u)) continue;              // reflexVertices, in index order
                if (r == p || r == n) continue;
                const float* q = P(r);
                if ((q[0] == a[0] && q[1] == a[1] && q[2] == a[2]) || (q[0] == b[0] && q[1] == b[1] && q[2] == b[2])) continue;
                if (!on_right(a, b, q, nx, ny, nz)) continue;
                if (!on_right(b, c, q, nx, ny, nz)) continue;
                if (!on_right(c, a, q, nx, ny, nz)) continue;
                ear = false;
            }
        }
        if (ear)
        {
            out[at] = loop[p]; out[at + 1] = loop[cur]; out[at + 2] = loop[n];
            put(nxt, p, n); put(prv, n, p);
            const int adj[2] = {p, n};
            for (int k = 0; k < 2; ++k)
            {
                const int v = adj[k];
                if (!((rfx >> v) & 1u)) continue;
                if (on_right(P(get(prv, v)), P(v), P(get(nxt, v)), nx, ny, nz)) rfx &= ~(1u << v);
            }
            at += 3; --left; skipped = 0;
        }
        else if (++skipped > left) return 0;             // stalled: the face is dropped (:899-903)
        cur = n;
    }
    out[at] = loop[get(prv, cur)]; out[at + 1] = loop[cur]; out[at + 2] = loop[get(nxt, cur)];
    return at + 3;
}

#ifndef SURTR_EAR_WAVE_MAX
#define SURTR_EAR_WAVE_MAX 256u      // faces of up to this many vertices are triangulated by a wave (64 * K vertices, K per lane)
#endif
#if SURTR_LANES == 64
// Poly::EarClipping for one face of 5..64 vertices on one wave: lane i holds vertex i (position, prev/next
// link, reflex flag) in registers; the sequential ear loop (:868-906) runs wave-uniformly, and the scan of the
// reflex list (:837-856, an "any reflex vertex inside the candidate ear") is one ballot.  Same triangles, same
// order, same stall rule as ear_clip_face.
template <class LP>
__device__ __attribute__((always_inline)) static inline uint32_t ear_clip_face_wave(const float* pos, const LP* loop, int N, uint32_t* out)
{
    const int lane = (int)lane_id();
    const bool mine = lane < N;
    const int32_t vid = (int32_t)loop[mine ? lane : 0];
    const float x = pos[3 * vid], y = pos[3 * vid + 1], z = pos[3 * vid + 2];
    // (every index below is wave-uniform: broadcasts go through scalar registers, not the LDS crossbar)
#ifdef SURTR_EAR_SHFL
#define EAR_BCAST(v, i) __shfl(v, i, 64)
#else
#define EAR_BCAST(v, i) lane_bcast(v, (uint32_t)(i))
#endif
    auto bx = [&](int i) { return EAR_BCAST(x, i); };
    auto by = [&](int i) { return EAR_BCAST(y, i); };
    auto bz = [&](int i) { return EAR_BCAST(z, i); };
    const float ax0 = bx(0), ay0 = by(0), az0 = bz(0);
    float nx, ny, nz;
    {
        const float ux = bx(1) - ax0, uy = by(1) - ay0, uz = bz(1) - az0;
        const float wx = bx(2) - ax0, wy = by(2) - ay0, wz = bz(2) - az0;
        nx = uy * wz - uz * wy; ny = uz * wx - ux * wz; nz = ux * wy - uy * wx;
    }
    {   // IsCCW (:753-762): the sum runs over v = 0..N-1 in order (float addition is not associative)
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int v = 0; v < N; ++v)
        {
            const int q = (v + 1) % N;
            const float ux = bx(v) - ax0, uy = by(v) - ay0, uz = bz(v) - az0;
            const float wx = bx(q) - ax0, wy = by(q) - ay0, wz = bz(q) - az0;
            sx = sx + (uy * wz - uz * wy); sy = sy + (uz * wx - ux * wz); sz = sz + (ux * wy - uy * wx);
        }
        if (dot3(sx, sy, sz, nx, ny, nz) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    }
    int prv = (lane + N - 1) % N, nxt = (lane + 1) % N;
    auto right_of = [&](float ax, float ay, float az, float bx_, float by_, float bz_, float cx, float cy, float cz) {
        const float ux = bx_ - ax, uy = by_ - ay, uz = bz_ - az;
        const float wx = cx - ax, wy = cy - ay, wz = cz - az;
        const float kx = uy * wz - uz * wy, ky = uz * wx - ux * wz, kz = ux * wy - uy * wx;
        return dot3(kx, ky, kz, nx, ny, nz) > 0.f;
    };
    bool rfx = false;
    {
        // (the one place where every lane reads ANOTHER lane of its own choosing: a real shuffle)
        const float px_ = __shfl(x, prv, 64), py_ = __shfl(y, prv, 64), pz_ = __shfl(z, prv, 64);
        const float qx = __shfl(x, nxt, 64), qy = __shfl(y, nxt, 64), qz = __shfl(z, nxt, 64);
        rfx = mine && !right_of(px_, py_, pz_, x, y, z, qx, qy, qz);
    }
    int skipped = 0, left = N, cur = 0;
    uint32_t at = 0;
    while (left > 3)
    {
        const int p = EAR_BCAST(prv, cur), n = EAR_BCAST(nxt, cur);
        const bool cur_reflex = EAR_BCAST((int)rfx, cur) != 0;
        bool ear = !cur_reflex;
        if (ear)
        {
            const float ax = bx(p), ay = by(p), az = bz(p);
            const float cx_ = bx(cur), cy_ = by(cur), cz_ = bz(cur);
            const float ex = bx(n), ey = by(n), ez = bz(n);
            bool inside = false;
            if (mine && rfx && lane != p && lane != n)
            {
                const bool same = (x == ax && y == ay && z == az) || (x == cx_ && y == cy_ && z == cz_);
                if (!same)
                    inside = right_of(ax, ay, az, cx_, cy_, cz_, x, y, z) && right_of(cx_, cy_, cz_, ex, ey, ez, x, y, z) &&
                             right_of(ex, ey, ez, ax, ay, az, x, y, z);
            }
            ear = __ballot(inside) == 0ull;
        }
        if (ear)
        {
            if (lane == 0) { out[at] = (uint32_t)loop[p]; out[at + 1] = (uint32_t)loop[cur]; out[at + 2] = (uint32_t)loop[n]; }
            if (lane == p) nxt = n;
            if (lane == n) prv = p;
            // reflex flags of the two neighbours, only if they were reflex (:885-893)
            // (all broadcasts are done by the whole wave; only the two lanes use them)
            const int np = EAR_BCAST(prv, p), nn = EAR_BCAST(nxt, n);
            const float npx = bx(np), npy = by(np), npz = bz(np), nnx = bx(nn), nny = by(nn), nnz = bz(nn);
            const float ppx = bx(p), ppy = by(p), ppz = bz(p), qqx = bx(n), qqy = by(n), qqz = bz(n);
            if (lane == p && rfx) rfx = !right_of(npx, npy, npz, x, y, z, qqx, qqy, qqz);
            if (lane == n && rfx) rfx = !right_of(ppx, ppy, ppz, x, y, z, nnx, nny, nnz);
            at += 3; --left; skipped = 0;
        }
        else if (++skipped > left) return 0;             // stalled: the face is dropped (:899-903)
        cur = n;
    }
    {
        const int p = EAR_BCAST(prv, cur), n = EAR_BCAST(nxt, cur);
        if (lane == 0) { out[at] = (uint32_t)loop[p]; out[at + 1] = (uint32_t)loop[cur]; out[at + 2] = (uint32_t)loop[n]; }
    }
    return at + 3;
}

// The same for a face of up to 64 * K vertices: lane l holds the vertices l, l + 64, ... (K of them) in registers.  Every index
// the sequential loop works with is wave-uniform, so "vertex i" is slot i / 64 of lane i % 64: K lane reads and a scalar select.
// The cap of a cell on a piece of 100 000 vertices and more has that many vertices; one lane walking it with its prev / next /
// reflex arrays in global memory (ear_clip_face) took milliseconds.  Same triangles, same order, same stall rule.
// (out of line, on the global copies of positions and loops: such faces are rare in fragments small enough for the LDS staging, and
// inlined twice the function cost the common path registers -- k_faces 0.36 -> 0.39 ms on configs[3])
template <int K, class LP>
__device__ __attribute__((noinline)) static uint32_t ear_clip_face_wave_k(const float* pos, const LP* loop, int N, uint32_t* out)
{
    const int lane = (int)lane_id();
    float x[K], y[K], z[K]; int prv[K], nxt[K]; bool rfx[K], mine[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
        const int v = lane + 64 * j;
        mine[j] = v < N;
        const int32_t vid = (int32_t)loop[mine[j] ? v : 0];
        x[j] = pos[3 * vid]; y[j] = pos[3 * vid + 1]; z[j] = pos[3 * vid + 2];
        prv[j] = (v + N - 1) % N; nxt[j] = (v + 1) % N; rfx[j] = false;
    }
    // value of vertex i (wave-uniform i)
    auto bf = [&](const float (&a)[K], int i) -> float {
        const int jj = i >> 6; const uint32_t l = (uint32_t)(i & 63);
        float r = lane_bcast(a[0], l);
#pragma unroll
        for (int j = 1; j < K; ++j) { const float t = lane_bcast(a[j], l); r = jj == j ? t : r; }
        return r;
    };
    auto bi = [&](const int (&a)[K], int i) -> int {
        const int jj = i >> 6; const uint32_t l = (uint32_t)(i & 63);
        int r = lane_bcast(a[0], l);
#pragma unroll
        for (int j = 1; j < K; ++j) { const int t = lane_bcast(a[j], l); r = jj == j ? t : r; }
        return r;
    };
    auto bb = [&](const bool (&a)[K], int i) -> bool {
        const int jj = i >> 6; const uint32_t l = (uint32_t)(i & 63);
        int r = lane_bcast((int)a[0], l);
#pragma unroll
        for (int j = 1; j < K; ++j) { const int t = lane_bcast((int)a[j], l); r = jj == j ? t : r; }
        return r != 0;
    };
    const float ax0 = bf(x, 0), ay0 = bf(y, 0), az0 = bf(z, 0);
    float nx, ny, nz;
    {
        const float ux = bf(x, 1) - ax0, uy = bf(y, 1) - ay0, uz = bf(z, 1) - az0;
        const float wx = bf(x, 2) - ax0, wy = bf(y, 2) - ay0, wz = bf(z, 2) - az0;
        nx = uy * wz - uz * wy; ny = uz * wx - ux * wz; nz = ux * wy - uy * wx;
    }
    {   // IsCCW (:753-762): the sum runs over v = 0..N-1 in order (float addition is not associative)
        float sx = 0.f, sy = 0.f, sz = 0.f;
        float px_ = ax0, py_ = ay0, pz_ = az0;
        for (int v = 0; v < N; ++v)
        {
            const int q = (v + 1) % N;
            const float qx = bf(x, q), qy = bf(y, q), qz = bf(z, q);
            const float ux = px_ - ax0, uy = py_ - ay0, uz = pz_ - az0;
            const float wx = qx - ax0, wy = qy - ay0, wz = qz - az0;
            sx = sx + (uy * wz - uz * wy); sy = sy + (uz * wx - ux * wz); sz = sz + (ux * wy - uy * wx);
            px_ = qx; py_ = qy; pz_ = qz;
        }
        if (dot3(sx, sy, sz, nx, ny, nz) < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    }
    auto right_of = [&](float ax, float ay, float az, float bx_, float by_, float bz_, float cx, float cy, float cz) {
        const float ux = bx_ - ax, uy = by_ - ay, uz = bz_ - az;
        const float wx = cx - ax, wy = cy - ay, wz = cz - az;
        const float kx = uy * wz - uz * wy, ky = uz * wx - ux * wz, kz = ux * wy - uy * wx;
        return dot3(kx, ky, kz, nx, ny, nz) > 0.f;
    };
    // reflex flags: every lane fetches the neighbours of its own vertices (a real shuffle per slot of the source)
#pragma unroll
    for (int j = 0; j < K; ++j)
    {
        float pxv = 0.f, pyv = 0.f, pzv = 0.f, qxv = 0.f, qyv = 0.f, qzv = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2)
        {
            const float tx = __shfl(x[s2], prv[j] & 63, 64), ty = __shfl(y[s2], prv[j] & 63, 64), tz = __shfl(z[s2], prv[j] & 63, 64);
            if ((prv[j] >> 6) == s2) { pxv = tx; pyv = ty; pzv = tz; }
            const float ux = __shfl(x[s2], nxt[j] & 63, 64), uy = __shfl(y[s2], nxt[j] & 63, 64), uz = __shfl(z[s2], nxt[j] & 63, 64);
            if ((nxt[j] >> 6) == s2) { qxv = ux; qyv = uy; qzv = uz; }
        }
        rfx[j] = mine[j] && !right_of(pxv, pyv, pzv, x[j], y[j], z[j], qxv, qyv, qzv);
    }
    auto set_i = [&](int (&a)[K], int i, int val) {
#pragma unroll
        for (int j = 0; j < K; ++j) if (lane == (i & 63) && (i >> 6) == j) a[j] = val;
    };
    int skipped = 0, left = N, cur = 0;
    uint32_t at = 0;
    while (left > 3)
    {
        const int p = bi(prv, cur), n = bi(nxt, cur);
        bool ear = !bb(rfx, cur);
        if (ear)
        {
            const float ax = bf(x, p), ay = bf(y, p), az = bf(z, p);
            const float cx_ = bf(x, cur), cy_ = bf(y, cur), cz_ = bf(z, cur);
            const float ex = bf(x, n), ey = bf(y, n), ez = bf(z, n);
            bool inside = false;
#pragma unroll
            for (int j = 0; j < K; ++j)
            {
                const int v = lane + 64 * j;
                if (mine[j] && rfx[j] && v != p && v != n)
                {
                    const bool same = (x[j] == ax && y[j] == ay && z[j] == az) || (x[j] == cx_ && y[j] == cy_ && z[j] == cz_);
                    if (!same && right_of(ax, ay, az, cx_, cy_, cz_, x[j], y[j], z[j]) && right_of(cx_, cy_, cz_, ex, ey, ez, x[j], y[j], z[j]) &&
                        right_of(ex, ey, ez, ax, ay, az, x[j], y[j], z[j])) inside = true;
                }
            }
            ear = __ballot(inside) == 0ull;
        }
        if (ear)
        {
            if (lane == 0) { out[at] = (uint32_t)loop[p]; out[at + 1] = (uint32_t)loop[cur]; out[at + 2] = (uint32_t)loop[n]; }
            set_i(nxt, p, n);
            set_i(prv, n, p);
            // reflex flags of the two neighbours, only if they were reflex (:885-893)
            const int np = bi(prv, p), nn = bi(nxt, n);
            const float npx = bf(x, np), npy = bf(y, np), npz = bf(z, np), nnx = bf(x, nn), nny = bf(y, nn), nnz = bf(z, nn);
            const float ppx = bf(x, p), ppy = bf(y, p), ppz = bf(z, p), qqx = bf(x, n), qqy = bf(y, n), qqz = bf(z, n);
#pragma unroll
            for (int j = 0; j < K; ++j)
            {
                const int v = lane + 64 * j;
                if (v == p && rfx[j]) rfx[j] = !right_of(npx, npy, npz, x[j], y[j], z[j], qqx, qqy, qqz);
                if (v == n && rfx[j]) rfx[j] = !right_of(ppx, ppy, ppz, x[j], y[j], z[j], nnx, nny, nnz);
            }
            at += 3; --left; skipped = 0;
        }
        else if (++skipped > left) return 0;             // stalled: the face is dropped (:899-903)
        cur = n;
    }
    {
        const int p = bi(prv, cur), n = bi(nxt, cur);
        if (lane == 0) { out[at] = (uint32_t)loop[p]; out[at + 1] = (uint32_t)loop[cur]; out[at + 2] = (uint32_t)loop[n]; }
    }
    return at + 3;
}
#endif

// fan != 0: RenderPolyhedron's isConvex branch (triangle fan per face, Src/Poly.cpp:696-706) instead of EarClipping.
// face_n / face_off / face_idx (nullptr in an event): the face loops of Poly::ExtractFaces for the single-solid operators
// (surtr_extract_faces: one fragment loaded); face_n = {faces, loop entries}.
__global__ __launch_bounds__(SURTR_WG) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_faces(FragRec* __restrict__ frags, const surtr_counts* __restrict__ counts,
                                                    FaceScratch FS, uint2* __restrict__ blkpool, uint32_t blk_per_wg,
                                                    Arena A, const uint32_t* __restrict__ forder, uint32_t cap_frags,
                                                    uint32_t fan, uint32_t* __restrict__ face_n, uint32_t* __restrict__ face_off,
                                                    int32_t* __restrict__ face_idx, uint32_t* __restrict__ frag_status,
                                                    const uint32_t* __restrict__ take_list, uint32_t* __restrict__ push_list)
{
    __shared__ Shared sh;
    // staging of a small fragment for the serial ExtractFaces (see "irregular" below)
#ifndef SURTR_FL_V
#define SURTR_FL_V 1024
#define SURTR_FL_H 4096
#endif
    constexpr uint32_t FL_V = SURTR_FL_V, FL_H = SURTR_FL_H, FL_J = SURTR_FL_H / 2u;
    __shared__ uint16_t f_loff[FL_V]; __shared__ uint16_t f_llen[FL_V]; __shared__ uint16_t f_nbr[FL_H];
    // the sliver path's visit marks, or (regular fragments of at most FL_J half-edges) the pointer-jumping arrays
    struct IrrLds { uint8_t vis[FL_H]; uint32_t cov[FL_H]; };
    struct JumpLds { uint16_t k0[FL_J], k1[FL_J], x0[FL_J], x1[FL_J], d1[FL_J]; };      // (the other distance array is the unused upper half of f_nbr)
    struct EarLds { float pos[3 * FL_V]; uint16_t loop[FL_J]; };       // once the faces are known: positions and face loops for the ear clippers
    union FacesLds { IrrLds irr; JumpLds jmp; EarLds ear; };
    static_assert(sizeof(EarLds) <= sizeof(IrrLds), "the ear staging reuses the bytes of the face search");
    static_assert(sizeof(JumpLds) <= sizeof(IrrLds), "the pointer-jumping arrays fit the bytes of the irregular path");
    __shared__ FacesLds FU;
    uint8_t* const f_vis = FU.irr.vis; uint32_t* const f_cov = FU.irr.cov;
    const uint32_t tid = threadIdx.x;
    const uint32_t nf = counts->n_frag;
    int32_t* base = FS.base + (size_t)blockIdx.x * FS.per_wg;
    const uint32_t HF = FS.HF;
    int32_t* keyA = base; int32_t* keyB = base + HF; int32_t* nxA = base + 2 * (size_t)HF; int32_t* nxB = base + 3 * (size_t)HF;
    int32_t* loopbuf = base + 4 * (size_t)HF; int32_t* eartmp = base + 5 * (size_t)HF;   // 3*HF
    uint32_t* tri = (uint32_t*)(base + 8 * (size_t)HF);                                   // 3*HF
    uint32_t* fcnt = (uint32_t*)(base + 11 * (size_t)HF);                                 // HF (per owner edge)
    uint2* blk = blkpool + (size_t)blockIdx.x * blk_per_wg;
#ifdef SURTR_STAMP
    const unsigned long long wg_t0f = __builtin_readcyclecounter();
#endif
    while (true)
    {
        __syncthreads();
        if (tid == 0)
        {
            if (take_list == nullptr) sh.misc[7] = frag_of_ticket(A, forder, cap_frags, atomicAdd(&A.cursors[7], 1u));
            else { const uint32_t t = atomicAdd(&A.cursors[86], 1u); sh.misc[7] = t < A.cursors[85] ? take_list[t] : 0xFFFFFFFFu; }      // second tier: the list of the first
        }
        __syncthreads();
        const uint32_t f = sh.misc[7];
        if (f >= nf) break;
        FragRec fr = frags[f];
        const uint32_t n = fr.mv_n, H = fr.mh_n;
        STAMP_DECL;
#ifdef SURTR_STAMP
        const unsigned long long frag_t0 = __builtin_readcyclecounter();
        if (tid == 0) for (int q = 0; q < 16; ++q) sh.ph[q] = 0;
#endif
        if (H > HF)
        {
            if (tid == 0)
            {
                if (push_list != nullptr) push_list[atomicAdd(&A.cursors[85], 1u)] = f;      // a fragment for the second tier's scratch
                else atomicMax(&A.cursors[5], (uint32_t)SURTR_E_CAPACITY);
            }
            continue;
        }
        const float* pos = A.pos + 3 * (size_t)fr.mv_off;
        const uint32_t* loff = A.loff + fr.mv_off; const uint32_t* llen = A.llen + fr.mv_off;
        const int32_t* nbr = A.nbr + fr.mh_off;           // ring of v starts at loff[v]-mh_off
        // 1. successor half-edge in the face loop; half-edge id = position in the packed ring array.
        //    A ring that lists a neighbour twice (sliver fragments) breaks the one-loop-per-half-edge
        //    property: such fragments take the literal, serial ExtractFaces below.
        // Fragments are small: the topology is staged in LDS (16-bit) whenever it fits, and the passes below read that copy.
        const bool topo_lds = n <= FL_V && H <= FL_H && n < 0xFFFFu;
        const bool jump_lds = topo_lds && H <= FL_J;
        if (tid == 0) sh.flagBad = 0;
        if (topo_lds)
        {
            for (uint32_t v = tid; v < n; v += group_size()) { f_loff[v] = (uint16_t)(loff[v] - fr.mh_off); f_llen[v] = (uint16_t)llen[v]; }
            for (uint32_t e = tid; e < H; e += group_size()) f_nbr[e] = (uint16_t)nbr[e];
        }
        __syncthreads();
        auto gLO = [&](uint32_t v) -> uint32_t { return loff[v] - fr.mh_off; };
        auto gLN = [&](uint32_t v) -> uint32_t { return llen[v]; };
        auto lLO = [&](uint32_t v) -> uint32_t { return f_loff[v]; };
        auto lLN = [&](uint32_t v) -> uint32_t { return f_llen[v]; };
        auto successors = [&](auto LO, auto LN, const auto* NB, auto* NX, auto* KEY, int32_t* copy) {
            typedef typename std::remove_reference<decltype(NX[0])>::type XT;
            bool dup = false;
            for (uint32_t v = tid; v < n; v += group_size())
            {
                const uint32_t lo = LO(v), len = LN(v);
                for (uint32_t s = 0; s < len; ++s)
                {
                    const uint32_t b = (uint32_t)NB[lo + s];
                    for (uint32_t s2 = 0; s2 < s; ++s2) if ((uint32_t)NB[lo + s2] == b) dup = true;
                    const uint32_t lb = LO(b), nbq = LN(b);
                    uint32_t q = 0;
                    while (q < nbq && (uint32_t)NB[lb + q] != v) ++q;
                    const uint32_t sq = (q == 0) ? nbq - 1 : q - 1;   // FaceLoop
                    NX[lo + s] = (XT)(lb + sq);
                    KEY[lo + s] = (XT)(lo + s);
                    if (copy != nullptr) copy[lo + s] = (int32_t)(lb + sq);
                }
            }
            if (dup) sh.flagBad = 1;
        };
        if (jump_lds) successors(lLO, lLN, f_nbr, FU.jmp.x0, FU.jmp.k0, nxA);      // (nxA: the successors as they are, for the checks after the jumps)
        else if (topo_lds) successors(lLO, lLN, f_nbr, nxA, keyA, (int32_t*)nullptr);
        else successors(gLO, gLN, nbr, nxA, keyA, (int32_t*)nullptr);
        __syncthreads();
        STAMP(60);
        bool irregular = sh.flagBad != 0;
        uint32_t nfaces = 0, lensum = 0;
        int32_t* faceLo = nxB; int32_t* faceLen = keyB;
        bool serial_extract = false;
        int32_t* kc = keyA;           // per half-edge: the smallest half-edge whose face walk visits it (== itself: a face starts here)
        bool staged = false, failed = false;
        // Two attempts at most: a fragment that looked regular may turn out to have a half-edge loop that passes through
        // a vertex twice (coincident vertices of a degenerate fragment; no ring lists a neighbour twice).  The reference
        // closes a face when its walk is back at the start VERTEX (:100-118), so such a loop is several faces; only the
        // literal path below reproduces that.  The loops then cover fewer than H half-edges: redo as irregular.
        for (int attempt = 0; attempt < 2; ++attempt)
        {
        nfaces = 0; lensum = 0; faceLo = nxB; faceLen = keyB; serial_extract = false; kc = keyA;
        const uint16_t* own16 = nullptr;      // LDS pointer jumping: per half-edge the smallest half-edge of its loop
        const uint16_t* dist16 = nullptr;     // ... and its distance to it along the loop
        uint16_t* free16a = nullptr; uint16_t* free16b = nullptr;      // the jump pointers' arrays, dead after the jumps
        staged = irregular && n <= FL_V && H <= FL_H;
        bool pinched = false;
        if (irregular)
        {
            // Sliver fragment (a ring lists a neighbour twice): the reference keys its visited set by the (vertex, neighbour)
            // pair = the first slot holding that neighbour, and walks by first occurrence (FaceLoop).  Such fragments are
            // small: work on an LDS copy.  Face starts are found speculatively -- every pair walks its face and leaves its
            // id (atomicMin) on the pairs it visits; starts = pairs that kept their own id -- and then checked against the
            // sequential definition (a pair starts a face iff no earlier start's walk visits it) by walking from the starts
            // only.  A failed check falls back to the literal serial loop.
            if (!staged) serial_extract = true;
            else
            {
                for (uint32_t v = tid; v < n; v += group_size()) { f_loff[v] = (uint16_t)(loff[v] - fr.mh_off); f_llen[v] = (uint16_t)llen[v]; }
                for (uint32_t e = tid; e < H; e += group_size()) { f_nbr[e] = (uint16_t)nbr[e]; f_vis[e] = 0; f_cov[e] = 0xFFFFu; }
                __syncthreads();
                auto slot_of = [&](uint32_t a, uint32_t b) -> uint32_t {
                    const uint32_t la = f_loff[a], na = f_llen[a];
                    uint32_t q = 0;
                    while (q < na && (uint32_t)f_nbr[la + q] != b) ++q;
                    return la + (q < na ? q : 0u);
                };
                auto walk_mark = [&](uint32_t i, uint32_t e) {          // visits of the face that starts with pair e = (i, nbr[e])
                    uint32_t prev = i, curv = f_nbr[e], steps = 0;
                    while (curv != i && steps++ <= H)
                    {
                        atomicMin(&f_cov[slot_of(prev, curv)], e);
                        const uint32_t nx = face_next(f_nbr + f_loff[curv], f_llen[curv], prev);
                        prev = curv; curv = nx;
                    }
                    atomicMin(&f_cov[slot_of(prev, curv)], e);
                    return steps;
                };
                bool toolong = false;
                for (uint32_t v = tid; v < n; v += group_size())
                {
                    const uint32_t lo = f_loff[v], len = f_llen[v];
                    for (uint32_t sI = 0; sI < len; ++sI)
                        if (slot_of(v, f_nbr[lo + sI]) == lo + sI && walk_mark(v, lo + sI) > H) toolong = true;
                }
                if (toolong) sh.flagBad = 2;
                __syncthreads();
                // starts by the speculative pass; then the check pass from those only
                for (uint32_t v = tid; v < n; v += group_size())
                {
                    const uint32_t lo = f_loff[v], len = f_llen[v];
                    for (uint32_t sI = 0; sI < len; ++sI) { const uint32_t e = lo + sI; f_vis[e] = (f_cov[e] == e) ? 1 : 0; }
                }
                __syncthreads();
                for (uint32_t e = tid; e < H; e += group_size()) f_cov[e] = 0xFFFFu;
                __syncthreads();
                for (uint32_t v = tid; v < n; v += group_size())
                {
                    const uint32_t lo = f_loff[v], len = f_llen[v];
                    for (uint32_t sI = 0; sI < len; ++sI) if (f_vis[lo + sI]) walk_mark(v, lo + sI);
                }
                __syncthreads();
                bool mism = false;
                for (uint32_t v = tid; v < n; v += group_size())
                {
                    const uint32_t lo = f_loff[v], len = f_llen[v];
                    for (uint32_t sI = 0; sI < len; ++sI)
                    {
                        const uint32_t e = lo + sI;
                        const bool canon = slot_of(v, f_nbr[e]) == e;
                        const uint32_t c = f_cov[e];
                        if (canon && (c > e || (c == e) != (f_vis[e] != 0))) mism = true;
                        keyA[e] = (canon && f_vis[e]) ? (int32_t)e : -1;
                    }
                }
                if (mism) sh.flagBad = 2;
                __syncthreads();
                if (sh.flagBad == 2) serial_extract = true;
                __syncthreads();
                faceLo = nxB; faceLen = keyB;
            }
        }
        else if (jump_lds)
        {
            // 2. minimum half-edge id of every loop by pointer jumping, on the LDS arrays
            // ... and, with it, the distance of every half-edge to that minimum along the loop (list ranking): a window of
            // 2^i successors per half-edge holds its smallest id and how far ahead it lies; when the window of e's 2^i-th
            // successor has a smaller one, that lies 2^i further.  Once the windows cover the loops, every half-edge knows its
            // face (the minimum = the half-edge ExtractFaces starts the face with) and its place in it -- no walk.
            uint16_t* k16 = FU.jmp.k0; uint16_t* kn = FU.jmp.k1; uint16_t* xc = FU.jmp.x0; uint16_t* xn = FU.jmp.x1;
            uint16_t* dc = f_nbr + FL_J; uint16_t* dn = FU.jmp.d1;
            for (uint32_t e = tid; e < H; e += group_size()) dc[e] = 0;
            __syncthreads();
            for (uint32_t span = 1; span < H; span <<= 1)
            {
                for (uint32_t e = tid; e < H; e += group_size())
                {
                    const uint32_t t = xc[e];
                    const uint16_t a = k16[e], b = k16[t];
                    const uint16_t da = dc[e], db = dc[t];
                    kn[e] = a <= b ? a : b;
                    dn[e] = a <= b ? da : (uint16_t)(db + span);
                    xn[e] = xc[t];
                }
                __syncthreads();
                uint16_t* t1 = k16; k16 = kn; kn = t1; t1 = xc; xc = xn; xn = t1; t1 = dc; dc = dn; dn = t1;
            }
            own16 = k16; dist16 = dc; free16a = xc; free16b = xn;
            STAMP(61);
        }
        else
        {
            // 2. minimum half-edge id of every loop by pointer jumping
            int32_t* kn = keyB; int32_t* xc = nxA; int32_t* xn = nxB;
            for (uint32_t span = 1; span < H; span <<= 1)
            {
                for (uint32_t e = tid; e < H; e += group_size())
                {
                    const int32_t t = xc[e];
                    const int32_t a = kc[e], b = kc[t];
                    kn[e] = a < b ? a : b;
                    xn[e] = xc[t];
                }
                __syncthreads();
                int32_t* t1 = kc; kc = kn; kn = t1; t1 = xc; xc = xn; xn = t1;
            }
            faceLo = xn; faceLen = kn;
            STAMP(61);
        }
        if (!serial_extract)
        {
            // 3. faces = owner half-edges in ascending order (ExtractFaces visiting order, Src/Poly.cpp:94-122)
            auto owners = [&](auto LO, auto LN, const auto* NB, auto OWN) {
                auto vertex_of = [&](uint32_t e) -> uint32_t {
                    uint32_t lo_v = 0, hi_v = n;
                    while (hi_v - lo_v > 1) { const uint32_t mid = (lo_v + hi_v) >> 1; if (LO(mid) <= e) lo_v = mid; else hi_v = mid; }
                    return lo_v;
                };
                auto ownfn = [&](uint32_t e) -> uint2 {
                    if (!OWN(e)) return make_uint2(0u, 0u);
                    const uint32_t start = vertex_of(e);
                    uint32_t prev = start, curv = (uint32_t)NB[e];
                    uint32_t len = 1;
                    while (curv != start && len <= H)
                    {
                        const uint32_t nx = (uint32_t)face_next(NB + LO(curv), LN(curv), prev);
                        prev = curv; curv = nx; ++len;
                    }
                    return make_uint2(1u, len);
                };
                scan_blocks(H, blk, sh, ownfn, nfaces, lensum);
                if (lensum > HF) return false;
                if (!irregular && lensum != H) pinched = true;
                const uint32_t nb = pinched ? 0u : (H + SURTR_LANES - 1u) >> SURTR_LSH;
                for (uint32_t b = wave_id(); b < nb; b += group_waves())
                {
                    const uint32_t e = (b << SURTR_LSH) + lane_id();
                    uint2 c = make_uint2(0u, 0u);
                    if (e < H) c = ownfn(e);
                    const uint2 ex = wave_excl2(c);
                    if (e < H && c.x)
                    {
                        const uint32_t fi = blk[b].x + ex.x, lo = blk[b].y + ex.y;
                        int32_t* loop = loopbuf + lo;
                        const uint32_t start = vertex_of(e);
                        uint32_t prev = start, curv = (uint32_t)NB[e];
                        uint32_t len = 1; loop[0] = (int32_t)start;
                        while (curv != start && len < c.y)
                        {
                            loop[len++] = (int32_t)curv;
                            const uint32_t nx = (uint32_t)face_next(NB + LO(curv), LN(curv), prev);
                            prev = curv; curv = nx;
                        }
                        faceLo[fi] = (int32_t)lo; faceLen[fi] = (int32_t)len;
                    }
                }
                return true;
            };
            // The same from the ranks of the LDS pointer jumping: a face's length is the distance of its first half-edge's successor
            // + 1, its vertices are the sources of its half-edges at (length - distance).  That holds when every loop is a simple
            // cycle that passes through no vertex twice (the reference closes a face at the start VERTEX, :100-118); the checks
            // below say so, and a fragment that fails one takes the path of the irregular ones (pinched).
            auto owners_ranked = [&]() -> bool {
                auto vertex_of = [&](uint32_t e) -> uint32_t {
                    uint32_t lo_v = 0, hi_v = n;
                    while (hi_v - lo_v > 1) { const uint32_t mid = (lo_v + hi_v) >> 1; if (f_loff[mid] <= e) lo_v = mid; else hi_v = mid; }
                    return lo_v;
                };
                bool bad = false;
                for (uint32_t e = tid; e < H; e += group_size())
                {
                    const uint32_t t = (uint32_t)nxA[e], o = own16[e];
                    if (t >= H || own16[t] != o) { bad = true; continue; }
                    const uint32_t want = o == e ? 0u : (t == o ? 1u : (uint32_t)dist16[t] + 1u);
                    if ((uint32_t)dist16[e] != want) bad = true;
                }
                for (uint32_t v = tid; v < n; v += group_size())
                {
                    const uint32_t lo = f_loff[v], len = f_llen[v];
                    for (uint32_t s1 = 1; s1 < len; ++s1)
                        for (uint32_t s2 = 0; s2 < s1; ++s2) if (own16[lo + s1] == own16[lo + s2]) bad = true;
                }
                if (bad) sh.flagBad = 3;
                __syncthreads();
                if (sh.flagBad == 3) { pinched = true; return true; }
                auto ownfn = [&](uint32_t e) -> uint2 {
                    if ((uint32_t)own16[e] != e) return make_uint2(0u, 0u);
                    return make_uint2(1u, (uint32_t)dist16[(uint32_t)nxA[e]] + 1u);
                };
                scan_blocks(H, blk, sh, ownfn, nfaces, lensum);
                if (lensum != H) { pinched = true; return true; }
                uint16_t* lo_of = free16a; uint16_t* len_of = free16b;
                const uint32_t nb = (H + SURTR_LANES - 1u) >> SURTR_LSH;
                for (uint32_t b = wave_id(); b < nb; b += group_waves())
                {
                    const uint32_t e = (b << SURTR_LSH) + lane_id();
                    uint2 c = make_uint2(0u, 0u);
                    if (e < H) c = ownfn(e);
                    const uint2 ex = wave_excl2(c);
                    if (e < H && c.x)
                    {
                        const uint32_t fi = blk[b].x + ex.x, lo = blk[b].y + ex.y;
                        faceLo[fi] = (int32_t)lo; faceLen[fi] = (int32_t)c.y;
                        lo_of[e] = (uint16_t)lo; len_of[e] = (uint16_t)c.y;
                    }
                }
                __syncthreads();
                for (uint32_t e = tid; e < H; e += group_size())
                {
                    const uint32_t o = own16[e], d = dist16[e];
                    loopbuf[(uint32_t)lo_of[o] + (d == 0u ? 0u : (uint32_t)len_of[o] - d)] = (int32_t)vertex_of(e);
                }
                return true;
            };
            bool ok;
            if (own16 != nullptr && dist16 != nullptr) ok = owners_ranked();
            else if (own16 != nullptr) ok = owners(lLO, lLN, f_nbr, [&](uint32_t e) { return (uint32_t)own16[e] == e; });
            else if (topo_lds) ok = owners(lLO, lLN, f_nbr, [&](uint32_t e) { return kc[e] == (int32_t)e; });
            else ok = owners(gLO, gLN, nbr, [&](uint32_t e) { return kc[e] == (int32_t)e; });
            if (!ok) { failed = true; break; }
        }
        else
        {
            // literal ExtractFaces (Src/Poly.cpp:89-126) on one lane; the visited set is keyed by the
            // (vertex, neighbour) pair = the first slot holding that neighbour
            if (staged) { for (uint32_t e = tid; e < H; e += group_size()) f_vis[e] = 0; }
            else for (uint32_t e = tid; e < H; e += group_size()) keyA[e] = 0;
            __syncthreads();
            auto extract = [&](auto LO, auto LN, auto* NB, auto* visited) {
                auto slot_of = [&](uint32_t a, uint32_t b) -> uint32_t {
                    const uint32_t la = LO(a), na = LN(a);
                    uint32_t q = 0;
                    while (q < na && (uint32_t)NB[la + q] != b) ++q;
                    return la + (q < na ? q : 0u);
                };
                uint32_t nfc = 0, lo = 0; bool bad = false;
                for (uint32_t i = 0; i < n && !bad; ++i)
                {
                    const uint32_t li = LO(i), ni = LN(i);
                    for (uint32_t s = 0; s < ni && !bad; ++s)
                    {
                        const uint32_t adj = NB[li + s];
                        if (visited[slot_of(i, adj)]) continue;
                        if (lo + 1 > HF || nfc >= HF) { bad = true; break; }
                        uint32_t len = 1; loopbuf[lo] = (int32_t)i;
                        uint32_t prev = i, curv = adj;
                        while (curv != i)
                        {
                            visited[slot_of(prev, curv)] = 1;
                            if (lo + len >= HF || len > H) { bad = true; break; }
                            loopbuf[lo + len++] = (int32_t)curv;
                            const uint32_t nx = face_next(NB + LO(curv), LN(curv), prev);
                            prev = curv; curv = nx;
                        }
                        if (bad) break;
                        visited[slot_of(prev, curv)] = 1;
                        faceLo[nfc] = (int32_t)lo; faceLen[nfc] = (int32_t)len; ++nfc; lo += len;
                    }
                }
                sh.misc[0] = nfc; sh.misc[1] = lo; sh.misc[2] = bad ? 1u : 0u;
            };
            if (tid == 0)
            {
                if (staged) extract([&](uint32_t v) { return (uint32_t)f_loff[v]; }, [&](uint32_t v) { return (uint32_t)f_llen[v]; }, f_nbr, f_vis);
                else extract([&](uint32_t v) { return loff[v] - fr.mh_off; }, [&](uint32_t v) { return llen[v]; }, nbr, keyA);
            }
            __syncthreads();
            nfaces = sh.misc[0]; lensum = sh.misc[1];
            const bool bad = sh.misc[2] != 0;
            __syncthreads();
            if (bad) { failed = true; break; }
        }
        __syncthreads();
        if (!pinched) break;
        irregular = true;
        if (tid == 0) sh.flagBad = 0;
        __syncthreads();
        }   // attempts
        if (failed)
        {
            // The reference's ExtractFaces never ends on this fragment (a walk that does not come back to its start vertex,
            // Src/Poly.cpp:100-118): it gets no triangles and a status of its own; the other fragments of the event stand.
            if (tid == 0)
            {
                frags[f].idx_off = 0; frags[f].idx_n = 0;
                // (k_refit, beside this kernel, may flag the same fragment: it is counted once)
                if (frag_status == nullptr || atomicExch(&frag_status[f], (uint32_t)SURTR_E_TOPOLOGY) == 0u) atomicAdd(&A.cursors[14], 1u);
            }
            continue;
        }
        if (face_n != nullptr)
        {
            for (uint32_t fi = tid; fi < nfaces; fi += group_size()) face_off[fi] = (uint32_t)faceLo[fi];
            for (uint32_t e = tid; e < lensum; e += group_size()) face_idx[e] = loopbuf[e];
            if (tid == 0) { face_off[nfaces] = lensum; face_n[0] = nfaces; face_n[1] = lensum; }
        }
        STAMP(62);
        // 4. triangulate: faces of 5..256 vertices one per wave (registers only), the others one per lane;
        //    room for 3*len indices at 3*lo.  The arrays of the face search are dead now: a small fragment's positions and
        //    loops go there, so that the ear tests read LDS instead of waiting for global memory vertex by vertex.
        const bool ear_lds = !fan && n <= FL_V && lensum <= FL_J;
        __syncthreads();
        if (ear_lds)
        {
            for (uint32_t i = tid; i < 3u * n; i += group_size()) FU.ear.pos[i] = pos[i];
            for (uint32_t e = tid; e < lensum; e += group_size()) FU.ear.loop[e] = (uint16_t)loopbuf[e];
            __syncthreads();
        }
        if (fan)
        {
            // isConvex: (f[0], f[v], f[v+1]) for v = 1 .. size-2 (Src/Poly.cpp:698-705)
            for (uint32_t fi = tid; fi < nfaces; fi += group_size())
            {
                const uint32_t lo = (uint32_t)faceLo[fi], len = (uint32_t)faceLen[fi];
                uint32_t* out = tri + 3u * (size_t)lo;
                const int32_t* loop = loopbuf + lo;
                uint32_t at = 0;
                for (uint32_t v = 1; v + 1u < len; ++v) { out[at] = (uint32_t)loop[0]; out[at + 1] = (uint32_t)loop[v]; out[at + 2] = (uint32_t)loop[v + 1]; at += 3u; }
                fcnt[fi] = at;
            }
        }
#if SURTR_LANES == 64
        // every wave reads the lengths of 64 faces at a time (one round trip) and takes every group_waves()-th face of 5..64
        // vertices among them: most faces are triangles, and a wave that looked at them one by one spent its time waiting
        for (uint32_t f0 = 0; f0 < (fan ? 0u : nfaces); f0 += SURTR_LANES)
        {
            const uint32_t fl = f0 + lane_id();
            uint32_t mylo = 0, mylen = 0;
            if (fl < nfaces) { mylo = (uint32_t)faceLo[fl]; mylen = (uint32_t)faceLen[fl]; }
            unsigned long long todo = __ballot(mylen >= 5u && mylen <= SURTR_EAR_WAVE_MAX);
            for (uint32_t ord = 0; todo != 0ull; todo &= todo - 1ull, ++ord)
            {
                if (ord % group_waves() != wave_id()) continue;
                const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                const uint32_t lo = lane_bcast(mylo, src), len = lane_bcast(mylen, src);
                uint32_t cnt;
                if (len > 64u) cnt = ear_clip_face_wave_k<SURTR_EAR_WAVE_MAX / 64>(pos, (const int32_t*)loopbuf + lo, (int)len, tri + 3u * (size_t)lo);      // (several vertices per lane)
                else if (ear_lds) cnt = ear_clip_face_wave((const float*)FU.ear.pos, (const uint16_t*)FU.ear.loop + lo, (int)len, tri + 3u * (size_t)lo);
                else cnt = ear_clip_face_wave(pos, (const int32_t*)loopbuf + lo, (int)len, tri + 3u * (size_t)lo);
                if (lane_id() == 0) fcnt[f0 + src] = cnt;
            }
        }
        __syncthreads();
        STAMP(63);
#endif
        for (uint32_t fi = tid; fi < (fan ? 0u : nfaces); fi += group_size())
        {
            const uint32_t lo = (uint32_t)faceLo[fi], len = (uint32_t)faceLen[fi];
#if SURTR_LANES == 64
            if (len >= 5u && len <= SURTR_EAR_WAVE_MAX) continue;
#endif
            uint32_t* out = tri + 3u * (size_t)lo;
            uint32_t cnt = 0u;
            if (len >= 3u && len <= 8u)
                cnt = ear_lds ? ear_clip_small((const float*)FU.ear.pos, (const uint16_t*)FU.ear.loop + lo, (int)len, out)
                              : ear_clip_small(pos, (const int32_t*)loopbuf + lo, (int)len, out);
            else if (len > 8u) cnt = ear_clip_face(pos, (const int32_t*)loopbuf + lo, (int)len, eartmp + 3 * (size_t)lo, out);
            fcnt[fi] = cnt;
#ifdef SURTR_STAMP
            if (len > 64u) { atomicAdd(&g_stamp[66], 1ull); atomicAdd(&g_stamp[67], (unsigned long long)len); }
#endif
        }
        __syncthreads();
        STAMP(64);
        // 5. compact the triangle lists of the faces, in face order, into the index arena
        auto cntfn = [&](uint32_t fi) -> uint2 { return make_uint2(fcnt[fi], 0u); };
        uint32_t nidx = 0, dum = 0;
        scan_blocks(nfaces, blk, sh, cntfn, nidx, dum);
        if (tid == 0) sh.misc[0] = atomicAdd(&A.cursors[2], nidx);
        __syncthreads();
        const uint32_t ioff = sh.misc[0];
        if ((uint64_t)ioff + nidx > A.capI) { if (tid == 0) atomicMax(&A.cursors[5], (uint32_t)SURTR_E_CAPACITY); continue; }
        {
            const uint32_t nb = (nfaces + SURTR_LANES - 1u) >> SURTR_LSH;
            for (uint32_t b = wave_id(); b < nb; b += group_waves())
            {
                const uint32_t fi = (b << SURTR_LSH) + lane_id();
                uint2 c = make_uint2(0u, 0u);
                if (fi < nfaces) c = cntfn(fi);
                const uint2 ex = wave_excl2(c);
                if (fi < nfaces && c.x)
                {
                    const uint32_t* src = tri + 3u * (size_t)(uint32_t)faceLo[fi];
                    uint32_t* dst = A.idx + ioff + blk[b].x + ex.x;
                    for (uint32_t q = 0; q < c.x; ++q) dst[q] = src[q];
                }
            }
        }
        if (tid == 0) { frags[f].idx_off = ioff; frags[f].idx_n = nidx; }     // field-wise: k_refit runs beside this kernel
        STAMP(65);
#ifdef SURTR_STAMP
        if (tid == 0)
        {
            const unsigned long long d = __builtin_readcyclecounter() - frag_t0;
            int bkt = 0; while ((d >> bkt) > 1 && bkt < 40) ++bkt; bkt = bkt < 12 ? 0 : bkt - 12; if (bkt > 15) bkt = 15;
            atomicAdd(&g_stamp2[32 + bkt], 1ull); atomicAdd(&g_stamp2[51], d);
            for (int q = 0; q < 6; ++q) atomicAdd(&g_stamp2[24 + q], sh.ph[(60 + q) & 15]);
            const unsigned long long old = atomicMax(&g_stamp2[48], d);
            if (d > old) { g_stamp2[49] = n; g_stamp2[50] = H; g_stamp2[55] = nfaces; for (int q = 0; q < 6; ++q) g_stamp2[56 + q] = sh.ph[(60 + q) & 15]; }
        }
#endif
    }
#ifdef SURTR_STAMP
    if (tid == 0) { const unsigned long long d = __builtin_readcyclecounter() - wg_t0f; atomicAdd(&g_stamp2[52], d); atomicMax(&g_stamp2[53], d); atomicAdd(&g_stamp2[54], 1ull); }
#endif
}

// --------------------------------------------------------------- k_out_scan
// (1 024 threads: one workgroup's scans over a few thousand records are chains of L2 round trips -- more threads, fewer links)
__global__ __launch_bounds__(SURTR_WG_WIDE) void k_out_scan(FragRec* __restrict__ frags, uint2* __restrict__ blk,
                                                       surtr_counts* __restrict__ counts, Arena A)
{
    __shared__ Shared sh;
    const uint32_t nf = counts->n_frag;
    const uint32_t nb = (nf + SURTR_LANES - 1u) >> SURTR_LSH;
    uint32_t t0 = 0, t1 = 0;
    auto f1 = [&](uint32_t f) -> uint2 { return make_uint2(frags[f].mv_n, frags[f].mh_n); };
    scan_blocks(nf, blk, sh, f1, t0, t1);
    for (uint32_t b = wave_id(); b < nb; b += group_waves())
    {
        const uint32_t f = (b << SURTR_LSH) + lane_id();
        uint2 c = make_uint2(0u, 0u);
        if (f < nf) c = f1(f);
        const uint2 e = wave_excl2(c);
        if (f < nf) { frags[f].o_mv = blk[b].x + e.x; frags[f].o_mh = blk[b].y + e.y; }
    }
    const uint32_t mv = t0, mh = t1;
    __syncthreads();
    auto f2 = [&](uint32_t f) -> uint2 { return make_uint2(frags[f].cv_n, frags[f].ch_n); };
    scan_blocks(nf, blk, sh, f2, t0, t1);
    for (uint32_t b = wave_id(); b < nb; b += group_waves())
    {
        const uint32_t f = (b << SURTR_LSH) + lane_id();
        uint2 c = make_uint2(0u, 0u);
        if (f < nf) c = f2(f);
        const uint2 e = wave_excl2(c);
        if (f < nf) { frags[f].o_cv = blk[b].x + e.x; frags[f].o_ch = blk[b].y + e.y; }
    }
    const uint32_t cv = t0, chh = t1;
    __syncthreads();
    auto f3 = [&](uint32_t f) -> uint2 { return make_uint2(frags[f].idx_n, 0u); };
    scan_blocks(nf, blk, sh, f3, t0, t1);
    for (uint32_t b = wave_id(); b < nb; b += group_waves())
    {
        const uint32_t f = (b << SURTR_LSH) + lane_id();
        uint2 c = make_uint2(0u, 0u);
        if (f < nf) c = f3(f);
        const uint2 e = wave_excl2(c);
        if (f < nf) frags[f].o_idx = blk[b].x + e.x;
    }
    if (threadIdx.x == 0)
    {
        counts->mesh_verts = mv; counts->mesh_nbrs = mh; counts->conv_verts = cv; counts->conv_nbrs = chh;
        counts->n_idx = t0; counts->status = A.cursors[5]; counts->n_failed = A.cursors[14] + A.cursors[15];
    }
}

// ------------------------------------------------------------------- k_pack
// Blob layout (all sections 64-byte aligned, in this order):
//   header  surtr_counts (32 B)
//   frag_ids i32[3*nf] | mesh_vert_off u32[nf+1] | mesh_pos f32[3*mv] | mesh_nbr_off u32[mv+1] | mesh_nbr i32[mh]
//   conv_vert_off u32[nf+1] | conv_pos f32[3*cv] | conv_nbr_off u32[cv+1] | conv_nbr i32[ch]
//   vnc f32[9*mv] | idx_off u32[nf+1] | idx u32[ni] | frag_status u32[nf]
struct BlobLayout
{
    size_t ids, mvo, mpos, mno, mnbr, cvo, cpos, cno, cnbr, vnc, ioff, idx, fstat, total;
};

__host__ __device__ static BlobLayout blob_layout(const surtr_counts& c)
{
    BlobLayout L;
    size_t at = 64;
    auto put = [&](size_t bytes) { size_t r = at; at += (bytes + 63) & ~(size_t)63; return r; };
    L.ids = put((size_t)c.n_frag * 12);
    L.mvo = put(((size_t)c.n_frag + 1) * 4);
    L.mpos = put((size_t)c.mesh_verts * 12);
    L.mno = put(((size_t)c.mesh_verts + 1) * 4);
    L.mnbr = put((size_t)c.mesh_nbrs * 4);
    L.cvo = put(((size_t)c.n_frag + 1) * 4);
    L.cpos = put((size_t)c.conv_verts * 12);
    L.cno = put(((size_t)c.conv_verts + 1) * 4);
    L.cnbr = put((size_t)c.conv_nbrs * 4);
    L.vnc = put((size_t)c.mesh_verts * 36);
    L.ioff = put(((size_t)c.n_frag + 1) * 4);
    L.idx = put((size_t)c.n_idx * 4);
    L.fstat = put((size_t)c.n_frag * 4);
    L.total = at;
    return L;
}

__global__ __launch_bounds__(SURTR_WG) void k_pack(const FragRec* __restrict__ frags, surtr_counts* __restrict__ counts,
                                                   Arena A, char* __restrict__ blob, size_t capacity, uint32_t with_vnc,
                                                   const uint32_t* __restrict__ frag_status, float cr, float cg, float cb)
{
    const surtr_counts c = *counts;
    const BlobLayout L = blob_layout(c);
    if (L.total > capacity)
    {
        // the caller's buffer is too small: say so in the blob header (zero counts) and in the event's status word, so
        // that neither a stale nor a zero-filled blob is taken for a result
        if (blockIdx.x == 0 && threadIdx.x == 0)
        {
            atomicMax(&A.cursors[5], (uint32_t)SURTR_E_CAPACITY);
            atomicMax(&counts->status, (uint32_t)SURTR_E_CAPACITY);
            if (capacity >= sizeof(surtr_counts))
            {
                surtr_counts z; memset(&z, 0, sizeof(z)); z.n_pairs = c.n_pairs; z.status = SURTR_E_CAPACITY;
                *(surtr_counts*)blob = z;
            }
        }
        return;
    }
    const uint32_t nf = c.n_frag, tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0)
    {
        *(surtr_counts*)blob = c;
        ((uint32_t*)(blob + L.mvo))[nf] = c.mesh_verts;
        ((uint32_t*)(blob + L.mno))[c.mesh_verts] = c.mesh_nbrs;
        ((uint32_t*)(blob + L.cvo))[nf] = c.conv_verts;
        ((uint32_t*)(blob + L.cno))[c.conv_verts] = c.conv_nbrs;
        ((uint32_t*)(blob + L.ioff))[nf] = c.n_idx;
    }
    for (uint32_t f = blockIdx.x; f < nf; f += gridDim.x)
    {
        const FragRec fr = frags[f];
        if (tid == 0)
        {
            int32_t* ids = (int32_t*)(blob + L.ids);
            ids[3 * f] = fr.cell; ids[3 * f + 1] = fr.piece; ids[3 * f + 2] = fr.island;
            ((uint32_t*)(blob + L.mvo))[f] = fr.o_mv;
            ((uint32_t*)(blob + L.cvo))[f] = fr.o_cv;
            ((uint32_t*)(blob + L.ioff))[f] = fr.o_idx;
            ((uint32_t*)(blob + L.fstat))[f] = frag_status[f];
        }
        float* mpos = (float*)(blob + L.mpos) + 3 * (size_t)fr.o_mv;
        const float* sp = A.pos + 3 * (size_t)fr.mv_off;
        for (uint32_t i = tid; i < 3 * fr.mv_n; i += group_size()) mpos[i] = sp[i];
        uint32_t* mno = (uint32_t*)(blob + L.mno) + fr.o_mv;
        for (uint32_t v = tid; v < fr.mv_n; v += group_size()) mno[v] = fr.o_mh + (A.loff[fr.mv_off + v] - fr.mh_off);
        int32_t* mnbr = (int32_t*)(blob + L.mnbr) + fr.o_mh;
        for (uint32_t e = tid; e < fr.mh_n; e += group_size()) mnbr[e] = A.nbr[fr.mh_off + e];
        float* cpos = (float*)(blob + L.cpos) + 3 * (size_t)fr.o_cv;
        const float* scp = A.pos + 3 * (size_t)fr.cv_off;
        for (uint32_t i = tid; i < 3 * fr.cv_n; i += group_size()) cpos[i] = scp[i];
        uint32_t* cno = (uint32_t*)(blob + L.cno) + fr.o_cv;
        for (uint32_t v = tid; v < fr.cv_n; v += group_size()) cno[v] = fr.o_ch + (A.loff[fr.cv_off + v] - fr.ch_off);
        int32_t* cnbr = (int32_t*)(blob + L.cnbr) + fr.o_ch;
        for (uint32_t e = tid; e < fr.ch_n; e += group_size()) cnbr[e] = A.nbr[fr.ch_off + e];
        if (with_vnc)
        {
            // VertexNormalColor{pos, (0,0,0), color} (Src/Poly.cpp:690-694; the default colour is 0.25 grey, Inc/Poly.h:68)
            float* vnc = (float*)(blob + L.vnc) + 9 * (size_t)fr.o_mv;
            for (uint32_t i = tid; i < 9 * fr.mv_n; i += group_size())
            {
                const uint32_t v = i / 9, k = i % 9;
                vnc[i] = k < 3 ? sp[3 * v + k] : (k < 6 ? 0.f : (k == 6 ? cr : (k == 7 ? cg : cb)));
            }
        }
        uint32_t* idx = (uint32_t*)(blob + L.idx) + fr.o_idx;
        for (uint32_t e = tid; e < fr.idx_n; e += group_size()) idx[e] = A.idx[fr.idx_off + e];
    }
}

// ------------------------------------------------------------ single clip op
__global__ __launch_bounds__(SURTR_WG) void k_clip_single(SolidIn in, const float4* __restrict__ planes, uint32_t F,
                                                          ScratchPool pool, float* opos, uint32_t* ooff, int32_t* onbr,
                                                          uint32_t* ollen, uint32_t cap_v, uint32_t cap_h,
                                                          uint32_t* result /* n, nh, status */)
{
    __shared__ Shared sh;
    __shared__ LdsTopo L;
    Scratch S = carve(pool, 0);
    for (uint32_t k = threadIdx.x; k < F; k += group_size()) sh.planes[k] = planes[k];
    __syncthreads();
    uint32_t n = 0, nh = 0;
    int err = SURTR_E_TOPOLOGY;      // a small sliver: the literal clipper below, from the start
    if (in.nv > SURTR_LITERAL_MESH_V || !solid_is_sliver(in))
        err = clip_any(in, F, S, sh, L, [&](auto& T) -> int {
            if (T.nLive == 0) return 0;
            const uint2 tot = index_live(T, sh);
            if (tot.x > cap_v || tot.y > cap_h) return SURTR_E_CAPACITY;
            write_solid(T, opos, ooff, ollen, onbr, 0, 0);
            n = tot.x; nh = tot.y;
            return 0;
        });
    __syncthreads();
    if (err == SURTR_E_TOPOLOGY)
    {
        const LitRun r = literal_run(in, F, pool, 0, &sh);
        if (r.err == 0 && (r.n > cap_v || r.nh > cap_h)) err = SURTR_E_CAPACITY;
        else if (r.err == 0) { literal_write(r, opos, ooff, ollen, onbr, 0, 0); n = r.n; nh = r.nh; err = 0; }
        else if (r.err != SURTR_E_TOPOLOGY) err = r.err;
    }
    if (threadIdx.x == 0) { if (err == 0) ooff[n] = nh; result[0] = n; result[1] = nh; result[2] = (uint32_t)err; }
}

// =================================================================== host ===

extern "C" {

const char* surtr_strerror(int code)
{
    switch (code)
    {
    case SURTR_OK: return "ok";
    case SURTR_E_INVALID: return "invalid argument";
    case SURTR_E_TOPOLOGY: return "topology error (asymmetric links or degree < 3)";
    case SURTR_E_CAPACITY: return "capacity exceeded";
    case SURTR_E_HIP: return "HIP runtime error";
    case SURTR_E_STATE: return "call order violated";
    case SURTR_E_NOGPU: return "no HIP device (the engine has no CPU fallback)";
    default: return "unknown";
    }
}

int surtr_create(int device, surtr_ctx** out)
{
    if (!out) return SURTR_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SURTR_E_NOGPU;
    if (device < 0 || device >= n) return SURTR_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return SURTR_E_HIP;
    surtr_ctx* ctx = new surtr_ctx;
    ctx->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        {
            uint32_t per_cu = 2u;      // k_clip_pairs / k_refit: LDS (one LdsTopo per workgroup) admits two per CU
            if (const char* e = getenv("SURTR_WG_PER_CU")) { const int v = atoi(e); if (v > 0 && v <= 32) per_cu = (uint32_t)v; }
            ctx->max_wg = (uint32_t)prop.multiProcessorCount * per_cu;
            ctx->max_wg_faces = (uint32_t)prop.multiProcessorCount * 4u;
            uint32_t half_per_cu = 4u;
            if (const char* e = getenv("SURTR_HALF_PER_CU")) { const int v = atoi(e); if (v > 0 && v <= 32) half_per_cu = (uint32_t)v; }
            ctx->max_wg_half = (uint32_t)prop.multiProcessorCount * half_per_cu;
            uint32_t small_per_cu = 8u;
            if (const char* e = getenv("SURTR_SMALL_PER_CU")) { const int v = atoi(e); if (v > 0 && v <= 32) small_per_cu = (uint32_t)v; }
            ctx->max_wg_small = (uint32_t)prop.multiProcessorCount * small_per_cu;
            uint32_t prep_per_cu = SURTR_PREP_WAVES;
            if (const char* e = getenv("SURTR_PREP_PER_CU")) { const int v = atoi(e); if (v > 0 && v <= 32) prep_per_cu = (uint32_t)v; }
            ctx->max_wg_prep = (uint32_t)prop.multiProcessorCount * prep_per_cu;
        }
        ctx->hw_wg = ctx->max_wg; ctx->hw_wg_faces = ctx->max_wg_faces; ctx->hw_wg_prep = ctx->max_wg_prep; ctx->hw_wg_big = ctx->n_wg_big;
    }
    // (surtr_destroy releases whatever was created so far: no leak on a failure half-way)
    if (hipMalloc((void**)&ctx->d_counts, sizeof(surtr_counts)) != hipSuccess ||
        hipMalloc((void**)&ctx->arena.cursors, 1024) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_half, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_cvx, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_big, hipEventDisableTiming) != hipSuccess) { surtr_destroy(ctx); return SURTR_E_HIP; }
    // (tests / fuzzers: the arrangement of several busy contexts without the call)
    if (const char* e = getenv("SURTR_EVENTS_IN_FLIGHT")) { const int v = atoi(e); if (v > 0) ctx->events_in_flight = (uint32_t)v; }
    *out = ctx;
    return SURTR_OK;
}

void surtr_destroy(surtr_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ctx->mset.release(); ctx->cset.release(); ctx->cells.release();
    free_dev(ctx->d_upload_err); free_dev(ctx->d_group_xf); free_dev(ctx->d_world); free_dev(ctx->sort_tmp); free_dev(ctx->d_from);
    free_dev(ctx->d_v012); free_dev(ctx->d_planes); free_dev(ctx->d_plane_off);
    free_dev(ctx->pool.base); free_dev(ctx->pool_small.base); free_dev(ctx->pool_half.base); free_dev(ctx->fs.base); free_dev(ctx->d_blk);
    free_dev(ctx->fs_big.base); free_dev(ctx->d_blk_big); free_dev(ctx->d_face_list);
    free_dev(ctx->d_pair_order); free_dev(ctx->d_face_group);
    free_dev(ctx->prep.base); free_dev(ctx->img.base); free_dev(ctx->d_order); free_dev(ctx->d_forder); free_dev(ctx->pool_rec.base); free_dev(ctx->d_hlist);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->ev_half) (void)hipEventDestroy(ctx->ev_half);
    if (ctx->ev_cvx) (void)hipEventDestroy(ctx->ev_cvx);
    if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
    if (ctx->ev_big) (void)hipEventDestroy(ctx->ev_big);
    for (int i = 0; i < 32; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 16; ++i) for (int q = 0; q < 2; ++q) if (ctx->hev[i][q]) (void)hipEventDestroy(ctx->hev[i][q]);
    free_dev(ctx->arena.pos); free_dev(ctx->arena.loff); free_dev(ctx->arena.llen); free_dev(ctx->arena.nbr);
    free_dev(ctx->arena.idx); free_dev(ctx->arena.isl); free_dev(ctx->arena.cursors);
    free_dev(ctx->d_pairs); free_dev(ctx->d_frags); free_dev(ctx->d_frag_status); free_dev(ctx->d_scanblk); free_dev(ctx->d_counts);
    free_dev(ctx->d_outside); free_dev(ctx->d_blob); free_dev(ctx->d_pair_list);
    delete ctx;
}

const char* surtr_last_error(surtr_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int surtr_set_stream(surtr_ctx* ctx, void* s)
{
    if (!ctx) return SURTR_E_INVALID;
    ctx->stream = (hipStream_t)s;
    return SURTR_OK;
}

int surtr_get_stream(surtr_ctx* ctx, void** s)
{
    if (!ctx || !s) return SURTR_E_INVALID;
    *s = (void*)ctx->stream;
    return SURTR_OK;
}

int surtr_set_scratch(surtr_ctx* ctx, uint32_t mv, uint32_t mh)
{
    if (!ctx) return SURTR_E_INVALID;
    ctx->user_cv = mv; ctx->user_ch = mh;
    free_dev(ctx->pool.base); ctx->pool.base = nullptr;
    return SURTR_OK;
}

int surtr_set_arena(surtr_ctx* ctx, uint64_t v, uint64_t h, uint64_t i)
{
    if (!ctx) return SURTR_E_INVALID;
    ctx->user_av = v; ctx->user_ah = h; ctx->user_ai = i;
    free_dev(ctx->arena.pos); ctx->arena.pos = nullptr;
    return SURTR_OK;
}

// Validates one solid: indices in range, degree >= 3, symmetric links (Src/Poly.cpp:253-260).
static int check_solid(uint32_t nv, const uint32_t* off, const int32_t* nbr)
{
    for (uint32_t v = 0; v < nv; ++v)
    {
        if (off[v + 1] < off[v]) return SURTR_E_INVALID;
        const uint32_t deg = off[v + 1] - off[v];
        if (deg < 3) return SURTR_E_TOPOLOGY;
        for (uint32_t j = off[v]; j < off[v + 1]; ++j)
        {
            const int32_t u = nbr[j];
            if (u < 0 || (uint32_t)u >= nv || (uint32_t)u == v) return SURTR_E_TOPOLOGY;
            bool back = false;
            for (uint32_t q = off[u]; q < off[u + 1]; ++q) if (nbr[q] == (int32_t)v) { back = true; break; }
            if (!back) return SURTR_E_TOPOLOGY;
        }
    }
    return SURTR_OK;
}

int surtr_upload_pattern(surtr_ctx* ctx, uint32_t n_cells, const uint32_t* face_off, const float* v012)
{
    if (!ctx || n_cells == 0 || !face_off || !v012) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const uint32_t nf = face_off[n_cells];
    for (uint32_t c = 0; c < n_cells; ++c)
        if (face_off[c + 1] < face_off[c] || face_off[c + 1] - face_off[c] > SURTR_MAXF) return SURTR_E_INVALID;
    free_dev(ctx->d_v012); free_dev(ctx->d_planes); free_dev(ctx->d_plane_off);
    ctx->d_v012 = nullptr; ctx->d_planes = nullptr; ctx->d_plane_off = nullptr; ctx->cap_pattern_faces = 0; ctx->cap_pattern_cells = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_v012, std::max<size_t>(16, (size_t)nf * 36)));
    HIPCHK(hipMalloc((void**)&ctx->d_planes, std::max<size_t>(16, (size_t)nf * 16)));
    HIPCHK(hipMalloc((void**)&ctx->d_plane_off, (size_t)(n_cells + 1) * 4));
    HIPCHK(hipMemcpy(ctx->d_v012, v012, (size_t)nf * 36, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_plane_off, face_off, (size_t)(n_cells + 1) * 4, hipMemcpyHostToDevice));
    ctx->h_plane_off.assign(face_off, face_off + n_cells + 1);
    ctx->n_cells = n_cells; ctx->n_faces = nf; ctx->planes_ready = false; ctx->pair_order_count = 0;
    return SURTR_OK;
}

int surtr_place_cells(surtr_ctx* ctx, const float scale[3], const float translate[3])
{
    if (!ctx || !scale || !translate) return SURTR_E_INVALID;
    if (!ctx->d_v012) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    const uint32_t nf = ctx->n_faces;
    if (nf)
        hipLaunchKernelGGL(k_place_cells, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, nf, ctx->d_v012, scale[0], scale[1],
                           scale[2], translate[0], translate[1], translate[2], ctx->d_planes);
    HIPCHK(hipGetLastError());
    ctx->planes_ready = true;
    return SURTR_OK;
}

int surtr_upload_planes(surtr_ctx* ctx, uint32_t n_cells, const uint32_t* plane_off, const float* planes)
{
    if (!ctx || n_cells == 0 || !plane_off || !planes) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const uint32_t nf = plane_off[n_cells];
    for (uint32_t c = 0; c < n_cells; ++c)
        if (plane_off[c + 1] < plane_off[c] || plane_off[c + 1] - plane_off[c] > SURTR_MAXF) return SURTR_E_INVALID;
    free_dev(ctx->d_v012); free_dev(ctx->d_planes); free_dev(ctx->d_plane_off);
    ctx->d_v012 = nullptr; ctx->d_planes = nullptr; ctx->d_plane_off = nullptr; ctx->cap_pattern_faces = 0; ctx->cap_pattern_cells = 0;
    HIPCHK(hipMalloc((void**)&ctx->d_planes, std::max<size_t>(16, (size_t)nf * 16)));
    HIPCHK(hipMalloc((void**)&ctx->d_plane_off, (size_t)(n_cells + 1) * 4));
    HIPCHK(hipMemcpy(ctx->d_planes, planes, (size_t)nf * 16, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_plane_off, plane_off, (size_t)(n_cells + 1) * 4, hipMemcpyHostToDevice));
    ctx->h_plane_off.assign(plane_off, plane_off + n_cells + 1);
    ctx->n_cells = n_cells; ctx->n_faces = nf; ctx->planes_ready = true; ctx->pair_order_count = 0;
    return SURTR_OK;
}

// Per-workgroup scratch grows with the largest piece (the wide variant of the clip holds a whole band with all its cuts, a
// k_faces workgroup the loops of a whole fragment).  With pieces of a few 100 000 vertices the full complement of workgroups
// would not fit in HBM: the persistent kernels then run with as many workgroups as a fixed share of the memory holds (a
// quarter for the clip, an eighth for k_faces, a sixteenth for the pre-pass) -- slower, not out of memory.  configs[3] needs
// 8 GB + 22 GB + 0.8 GB at full size on a 288 GB part and is not affected.
#define SURTR_FACES_TIER_DEFAULT 524288u      // half-edges of a fragment the regular k_faces workgroups have room for (25 MB each)
static void budget_workgroups(surtr_ctx* ctx)
{
    if (ctx->budget_vmax == ctx->vmax && ctx->budget_hmax == ctx->hmax) return;
    ctx->budget_vmax = ctx->vmax; ctx->budget_hmax = ctx->hmax;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) total_b = (size_t)64 << 30;
    if (const char* e = getenv("SURTR_MEM_BUDGET_MB")) { const long v = atol(e); if (v > 0) total_b = (size_t)v << 20; }
    auto fit = [](size_t budget, size_t per, uint32_t lo, uint32_t hi) {
        const size_t n = per ? budget / per : hi;
        return (uint32_t)std::min<size_t>(hi, std::max<size_t>(lo, n));
    };
    const uint32_t CV = ctx->user_cv ? ctx->user_cv : 2 * ctx->vmax + 4096, CH = ctx->user_ch ? ctx->user_ch : 3 * ctx->hmax + 16384;
    const uint32_t all = fit(total_b / 4, scratch_bytes_per_wg(std::max(CV, 64u), CH, ctx->vmax), 20u, ctx->hw_wg + std::max(ctx->hw_wg / 2u, ctx->hw_wg_big));
    // the large bands go through the record clipper with a whole CU's LDS (k_clip_pairs_wave_big, one workgroup per CU) on
    // meshes where such bands are the rule; otherwise through k_clip_pairs_big's few dozen workgroups
    // (measured at 4 096 cells: 100 000-vertex piece 9.5 -> 7.8 ms, 210 000 18.4 -> 14.5 ms; at 500 000 vertices most bands outgrow even
    // a whole CU's LDS and are better spread over all workgroups' global scratch: 39.7 ms against 61.6)
    ctx->wave_big = ctx->vmax >= 80000u && ctx->vmax < 300000u;
    if (const char* e = getenv("SURTR_WAVE_BIG")) ctx->wave_big = atoi(e) != 0;
    const uint32_t hw_big = ctx->wave_big ? std::max(ctx->hw_wg / 2u, ctx->hw_wg_big) : ctx->hw_wg_big;
    ctx->n_wg_big = all >= ctx->hw_wg + hw_big ? hw_big : std::min(hw_big, std::max(4u, all / 8u));
    ctx->max_wg = std::min(ctx->hw_wg, all - ctx->n_wg_big);
    size_t HF = (size_t)ctx->hmax + ctx->hmax / 2 + 8192;
    {
        size_t tier = SURTR_FACES_TIER_DEFAULT;
        if (const char* e = getenv("SURTR_FACES_TIER_HE")) { const long v = atol(e); if (v >= 64) tier = (size_t)v; }
        HF = std::min(HF, tier);      // larger fragments go to the second tier of k_faces (ensure_scratch)
    }
    ctx->max_wg_faces = fit(total_b / 8, HF * 48, 16u, ctx->hw_wg_faces);
    ctx->max_wg_prep = fit(total_b / 16, prep_bytes_per_wg(std::max(ctx->vmax, 64u)), 16u, ctx->hw_wg_prep);
}

static int ensure_scratch(surtr_ctx* ctx, uint32_t need_v, uint32_t need_h, uint32_t n_wg)
{
    uint32_t n_wg_faces = std::max(1u, ctx->max_wg_faces);
    if (const char* e = getenv("SURTR_FACES_WG")) { const uint32_t v = (uint32_t)atoi(e); if (v > n_wg_faces && v <= 8192u) n_wg_faces = v; }
    // Tombstones keep every vertex ever created in its slot, so the wide (global) variant is sized for
    // the band plus all cuts; the LDS variant has fixed capacities (SURTR_LV / SURTR_LH).
    uint32_t CV = ctx->user_cv ? ctx->user_cv : 2 * need_v + 4096;
    uint32_t CH = ctx->user_ch ? ctx->user_ch : 3 * need_h + 16384;
    if (CV < 64) CV = 64;
    const uint32_t VMAX = need_v;
    if (ctx->pool.base && ctx->pool.CV >= CV && ctx->pool.CH >= CH && ctx->pool.VMAX >= VMAX && ctx->n_wg >= n_wg && ctx->n_wg_faces_alloc >= n_wg_faces) return SURTR_OK;
    ctx->n_wg_faces_alloc = n_wg_faces;
    free_dev(ctx->pool.base); free_dev(ctx->fs.base); free_dev(ctx->d_blk);
    ctx->pool.base = nullptr; ctx->fs.base = nullptr; ctx->d_blk = nullptr;
    ctx->pool.CV = CV; ctx->pool.CH = CH; ctx->pool.VMAX = VMAX;
    ctx->pool.per_wg = scratch_bytes_per_wg(CV, CH, VMAX);
    ctx->n_wg = n_wg;
    HIPCHK(hipMalloc((void**)&ctx->pool.base, ctx->pool.per_wg * n_wg));
    const uint32_t HF_full = need_h + need_h / 2 + 8192;
    uint32_t tier = SURTR_FACES_TIER_DEFAULT;
    // (several contexts on the GPU, surtr_set_events_in_flight: every one of them holds this scratch -- 22 GB at configs[3] with the
    //  full tier, sized for a fragment as large as the piece; a quarter of it, and the second tier takes the fragments beyond)
    if (ctx->events_in_flight > 1u) tier /= 4u;
    if (const char* e = getenv("SURTR_FACES_TIER_HE")) { const long v = atol(e); if (v >= 64) tier = (uint32_t)v; }
    ctx->fs.HF = std::min(HF_full, tier);
    ctx->fs.per_wg = (size_t)12 * ctx->fs.HF;
    HIPCHK(hipMalloc((void**)&ctx->fs.base, ctx->fs.per_wg * 4 * n_wg_faces));
    ctx->blk_per_wg = ctx->fs.HF / SURTR_LANES + 4;
    HIPCHK(hipMalloc((void**)&ctx->d_blk, (size_t)ctx->blk_per_wg * 8 * n_wg_faces));
    free_dev(ctx->fs_big.base); free_dev(ctx->d_blk_big); ctx->fs_big.base = nullptr; ctx->d_blk_big = nullptr; ctx->n_wg_faces_big = 0;
    if (HF_full > ctx->fs.HF)
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) total_b = (size_t)64 << 30;
        if (const char* e = getenv("SURTR_MEM_BUDGET_MB")) { const long v = atol(e); if (v > 0) total_b = (size_t)v << 20; }
        ctx->fs_big.HF = HF_full; ctx->fs_big.per_wg = (size_t)12 * HF_full;
        ctx->n_wg_faces_big = (uint32_t)std::min<size_t>(64, std::max<size_t>(2, (total_b / 16) / (ctx->fs_big.per_wg * 4)));
        HIPCHK(hipMalloc((void**)&ctx->fs_big.base, ctx->fs_big.per_wg * 4 * ctx->n_wg_faces_big));
        ctx->blk_per_wg_big = HF_full / SURTR_LANES + 4;
        HIPCHK(hipMalloc((void**)&ctx->d_blk_big, (size_t)ctx->blk_per_wg_big * 8 * ctx->n_wg_faces_big));
    }
    return SURTR_OK;
}

static int ensure_scratch_small(surtr_ctx* ctx, uint32_t n_wg)
{
    // Convex solids: the input hull plus at most a few vertices per clipping plane
    const uint32_t need_v = 2 * ctx->cvmax + 4 * SURTR_MAXF + 256, need_h = 2 * ctx->chmax + 12 * SURTR_MAXF + 1024;
    const uint32_t CV = 2 * need_v + 1024, CH = 3 * need_h + 4096, VMAX = need_v;
    if (ctx->pool_small.base && ctx->pool_small.CV >= CV && ctx->pool_small.CH >= CH && ctx->n_wg_small >= n_wg) return SURTR_OK;
    free_dev(ctx->pool_small.base); ctx->pool_small.base = nullptr;
    ctx->pool_small.CV = CV; ctx->pool_small.CH = CH; ctx->pool_small.VMAX = VMAX;
    ctx->pool_small.per_wg = scratch_bytes_per_wg(CV, CH, VMAX);
    ctx->n_wg_small = n_wg;
    HIPCHK(hipMalloc((void**)&ctx->pool_small.base, ctx->pool_small.per_wg * n_wg));
    return SURTR_OK;
}

static int ensure_scratch_half(surtr_ctx* ctx, uint32_t n_wg)
{
    // k_clip_pairs_half never leaves its LDS topology: positions, work lists and squeeze staging for SURTR_LVS / SURTR_LHS
    const uint32_t CV = SURTR_LVS + 256u, CH = SURTR_LHS + 256u, VMAX = SURTR_LVS + 64u;
    if (ctx->pool_half.base && ctx->n_wg_half >= n_wg) return SURTR_OK;
    free_dev(ctx->pool_half.base); ctx->pool_half.base = nullptr;
    ctx->pool_half.CV = CV; ctx->pool_half.CH = CH; ctx->pool_half.VMAX = VMAX;
    ctx->pool_half.per_wg = scratch_bytes_per_wg(CV, CH, VMAX);
    ctx->n_wg_half = n_wg;
    HIPCHK(hipMalloc((void**)&ctx->pool_half.base, ctx->pool_half.per_wg * n_wg));
    return SURTR_OK;
}

// Scratch of k_prep_pairs and the arena its images go to.  An image is at most the LDS topology plus masks; when the
// arena runs out, k_clip_pairs pre-passes the remaining pairs itself, so its size only matters for speed.
static int ensure_prep(surtr_ctx* ctx, uint32_t n_pairs, uint32_t n_wg)
{
    const uint32_t VMAX = std::max(ctx->vmax, 64u);
    if (!(ctx->prep.base && ctx->prep.VMAX >= VMAX && ctx->n_wg_prep >= n_wg))
    {
        free_dev(ctx->prep.base); ctx->prep.base = nullptr;
        ctx->prep.VMAX = VMAX; ctx->prep.per_wg = prep_bytes_per_wg(VMAX); ctx->n_wg_prep = n_wg;
        HIPCHK(hipMalloc((void**)&ctx->prep.base, ctx->prep.per_wg * n_wg));
    }
    if (ctx->cap_order < n_pairs)
    {
        free_dev(ctx->d_order); ctx->d_order = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->d_order, (size_t)n_pairs * 64 * 4));      // four tables: clip order, pre-pass order, half clip order, small record tier
        ctx->cap_order = n_pairs;
    }
    const uint64_t full = (uint64_t)ctx->vmax * 16 + (uint64_t)ctx->hmax * 2;
    const uint64_t lds = (uint64_t)SURTR_LV * 16 + (uint64_t)SURTR_LH * 2;
    const uint64_t per_pair = std::min(full, lds) + ctx->vmax / 8 + 2048;
    uint64_t bytes = std::min<uint64_t>(std::max<uint64_t>((uint64_t)n_pairs * per_pair, 1ull << 20), 4ull << 30);
    if (const char* e = getenv("SURTR_IMG_BYTES")) { const long long v = atoll(e); if (v >= 4096) bytes = (uint64_t)v; }      // tests: a small arena
    const uint32_t cap16 = (uint32_t)(bytes / 16);
    if (!(ctx->img.base && ctx->img.cap16 >= cap16) || (getenv("SURTR_IMG_BYTES") && ctx->img.cap16 != cap16))
    {
        free_dev(ctx->img.base); ctx->img.base = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->img.base, (size_t)cap16 * 16));
        ctx->img.cap16 = cap16;
    }
    return SURTR_OK;
}

static int ensure_arena(surtr_ctx* ctx, uint32_t n_pairs, uint64_t min_v = 0, uint64_t min_h = 0, uint64_t min_i = 0)
{
    // Result sizes are data dependent.  Cells partition space, so the Mesh fragments of an event add up to the pieces plus
    // their cut points: the default reserves three times all pieces (at least eight times the largest one) plus slack per pair
    // (Convex results); surtr_set_arena overrides it, SURTR_E_CAPACITY reports a default that was too small.
    uint64_t av = ctx->user_av ? ctx->user_av : std::max<uint64_t>(std::max<uint64_t>((uint64_t)ctx->vmax * 8, ctx->tot_mv * 3) + (uint64_t)n_pairs * 256, 1u << 16);
    uint64_t ah = ctx->user_ah ? ctx->user_ah : std::max<uint64_t>(std::max<uint64_t>((uint64_t)ctx->hmax * 8, ctx->tot_mh * 3) + (uint64_t)n_pairs * 1024, 1u << 18);
    uint64_t ai = ctx->user_ai ? ctx->user_ai : ah * 2;
    av = std::max(av, min_v); ah = std::max(ah, min_h); ai = std::max(ai, min_i);
    av = std::min<uint64_t>(av, 0xFFFFFFF0ull); ah = std::min<uint64_t>(ah, 0xFFFFFFF0ull); ai = std::min<uint64_t>(ai, 0xFFFFFFF0ull);
    const uint32_t capIsl = (uint32_t)std::min<uint64_t>((uint64_t)n_pairs * 4 + 1024, 0x7FFFFFFFull);
    if (!(ctx->arena.pos && ctx->arena.capV >= av && ctx->arena.capH >= ah && ctx->arena.capI >= ai && ctx->arena.capIsl >= capIsl))
    {
        free_dev(ctx->arena.pos); free_dev(ctx->arena.loff); free_dev(ctx->arena.llen); free_dev(ctx->arena.nbr);
        free_dev(ctx->arena.idx); free_dev(ctx->arena.isl); free_dev(ctx->d_frags); free_dev(ctx->d_frag_status);
        ctx->arena.pos = nullptr; ctx->arena.loff = nullptr; ctx->arena.llen = nullptr; ctx->arena.nbr = nullptr;
        ctx->arena.idx = nullptr; ctx->arena.isl = nullptr; ctx->d_frags = nullptr; ctx->d_frag_status = nullptr;
        ctx->arena.capV = ctx->arena.capH = ctx->arena.capI = ctx->arena.capIsl = 0; ctx->cap_frags = 0;
        free_dev(ctx->d_forder); ctx->d_forder = nullptr;
        hipError_t e = hipMalloc((void**)&ctx->arena.pos, av * 12);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->arena.loff, av * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->arena.llen, av * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->arena.nbr, ah * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->arena.idx, ai * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->arena.isl, (size_t)capIsl * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_frags, (size_t)capIsl * sizeof(FragRec));
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_forder, (size_t)capIsl * 16 * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_frag_status, (size_t)capIsl * 4);
        if (e == hipSuccess) { free_dev(ctx->d_face_list); ctx->d_face_list = nullptr; e = hipMalloc((void**)&ctx->d_face_list, (size_t)capIsl * 4); }
        if (e != hipSuccess)
        {
            // out of memory half-way: leave no buffer behind, so that a later, smaller event allocates afresh
            free_dev(ctx->arena.pos); free_dev(ctx->arena.loff); free_dev(ctx->arena.llen); free_dev(ctx->arena.nbr);
            free_dev(ctx->arena.idx); free_dev(ctx->arena.isl); free_dev(ctx->d_frags); free_dev(ctx->d_forder); free_dev(ctx->d_frag_status);
            ctx->d_frag_status = nullptr;
            free_dev(ctx->d_face_list); ctx->d_face_list = nullptr;
            ctx->arena.pos = nullptr; ctx->arena.loff = nullptr; ctx->arena.llen = nullptr; ctx->arena.nbr = nullptr;
            ctx->arena.idx = nullptr; ctx->arena.isl = nullptr; ctx->d_frags = nullptr; ctx->d_forder = nullptr;
            ctx->err = std::string("arena allocation: ") + hipGetErrorString(e);
            return SURTR_E_HIP;
        }
        ctx->arena.capV = (uint32_t)av; ctx->arena.capH = (uint32_t)ah; ctx->arena.capI = (uint32_t)ai; ctx->arena.capIsl = capIsl;
        ctx->cap_frags = capIsl;
    }
    if (ctx->cap_pairs < n_pairs)
    {
        free_dev(ctx->d_pairs); ctx->d_pairs = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->d_pairs, (size_t)n_pairs * sizeof(PairRec)));
        ctx->cap_pairs = n_pairs;
    }
    const uint32_t need_blk = std::max(n_pairs, ctx->cap_frags) / SURTR_LANES + 4;
    if (ctx->cap_scanblk < need_blk)
    {
        free_dev(ctx->d_scanblk); ctx->d_scanblk = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->d_scanblk, (size_t)need_blk * 8));
        ctx->cap_scanblk = need_blk;
    }
    return SURTR_OK;
}

// Start of an event: queue cursors and counters to zero, fragment status words to zero, hand-over list to "empty".
__global__ void k_event_init(uint32_t* __restrict__ cursors, uint32_t* __restrict__ counts, uint32_t n_counts, uint32_t* __restrict__ frag_status,
                             uint32_t n_status, uint32_t* __restrict__ hlist, uint32_t n_hlist)
{
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (uint32_t i = i0; i < 256u; i += step) cursors[i] = 0u;
    for (uint32_t i = i0; i < n_counts; i += step) counts[i] = 0u;
    for (uint32_t i = i0; i < n_status; i += step) frag_status[i] = 0u;
    for (uint32_t i = i0; i < n_hlist; i += step) hlist[i] = 0xFFFFFFFFu;
}

static int upload_pair_order(surtr_ctx* ctx, const uint32_t* ord, uint32_t n_pairs)
{
    if (ctx->cap_pair_order < n_pairs)
    {
        free_dev(ctx->d_pair_order); ctx->d_pair_order = nullptr; ctx->cap_pair_order = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_pair_order, (size_t)n_pairs * 4));
        ctx->cap_pair_order = n_pairs;
    }
    // a previous event (possibly still running on a non-blocking stream) may be reading the table
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_pair_order, ord, (size_t)n_pairs * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));     // `ord` is the caller's staging buffer
    return SURTR_OK;
}

#ifndef SURTR_WAVE_BIG_N
#define SURTR_WAVE_BIG_N 2800u
#endif
#ifndef SURTR_REC_MAXN
#define SURTR_REC_MAXN 2304u
#endif
#ifndef SURTR_FRONT_PAR_MAX
#define SURTR_FRONT_PAR_MAX 1024u      // pairs of an event up to which k_clip_convex and the pre-pass kernel run side by side
#endif
static int launch_event(surtr_ctx* ctx, uint32_t cell_begin, uint32_t n_pairs, const uint2* d_pair_list, const uint8_t* outside, uint32_t flags)
{
    (void)hipSetDevice(ctx->device);
    budget_workgroups(ctx);
    const uint32_t max_wg = ctx->max_wg;
    const uint32_t n_wg = std::max(1u, std::min(std::max(n_pairs, 1u), max_wg));
    // scratch slots [0, max_wg) belong to k_clip_pairs, the n_wg_big after them to k_clip_pairs_big
    // scratch slots [0, max_wg) belong to the Mesh clip's main kernel, the n_wg_big after them to k_clip_pairs_big, then k_clip_pairs_catch's
    int rc = ensure_scratch(ctx, ctx->vmax, ctx->hmax, max_wg + ctx->n_wg_big + ctx->n_wg_catch);
    if (rc) return rc;
    const uint32_t n_wg_small = std::max(1u, std::min(std::max(n_pairs, 1u), ctx->max_wg_small));
    rc = ensure_scratch_small(ctx, std::max(ctx->max_wg_small, ctx->n_wg_small));
    if (rc) return rc;
    const uint32_t n_wg_half = std::max(1u, std::min(std::max(n_pairs, 1u), ctx->max_wg_half));
    rc = ensure_scratch_half(ctx, std::max(ctx->max_wg_half, ctx->n_wg_half));
    if (rc) return rc;
    rc = ensure_arena(ctx, std::max(n_pairs, 1u));
    if (rc) return rc;
    const uint32_t n_wg_prep = std::max(1u, std::min(std::max(n_pairs, 1u), ctx->max_wg_prep));
    rc = ensure_prep(ctx, std::max(n_pairs, 1u), std::max(n_wg_prep, ctx->n_wg_prep));
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    // (the event's zeroes and the hand-over list's "empty" words in one launch, not four fills with a few microseconds between each)
    const uint32_t hcap = n_pairs + 4096u;
    if (ctx->cap_hlist < hcap)
    {
        HIPCHK(hipStreamSynchronize(st));      // (a previous event may still read the list)
        free_dev(ctx->d_hlist); ctx->d_hlist = nullptr; ctx->cap_hlist = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_hlist, (size_t)hcap * 4));
        ctx->cap_hlist = hcap;
    }
    static_assert(sizeof(surtr_counts) % 4 == 0, "surtr_counts is cleared by words");
    hipLaunchKernelGGL(k_event_init, dim3(SURTR_LANES == 1 ? 1 : 64), dim3(SURTR_LANES == 1 ? 1 : 256), 0, st, ctx->arena.cursors, (uint32_t*)ctx->d_counts,
                       (uint32_t)(sizeof(surtr_counts) / 4), ctx->d_frag_status, ctx->cap_frags, ctx->d_hlist, hcap);
    const uint8_t* d_out = nullptr;
    ctx->last_outside.clear();
    if (outside) ctx->last_outside.assign(outside, outside + ctx->n_pieces);
    if (outside)
    {
        HIPCHK(hipMemcpyAsync(ctx->d_outside, outside, ctx->n_pieces, hipMemcpyHostToDevice, st));
        d_out = ctx->d_outside;
    }
    const PieceSet& M = ctx->mset; const PieceSet& C = ctx->cset;
    Pieces P{M.pos, M.loff, M.llen, M.nbr, M.vo, M.tri, M.rad, M.perm, M.posr_s, M.bsph, M.bo,
             C.pos, C.loff, C.llen, C.nbr, C.vo, C.tri, C.rad, C.perm, C.posr_s, C.bsph, C.bo, ctx->n_pieces, M.dup, C.dup,
             M.row_s, M.iperm, M.bsph2, M.bo2, M.bsph3, M.bo3};
    for (int i = 0; i < 16; ++i) ctx->ev_valid[i] = false;
    const uint32_t* d_pair_order = nullptr;
    if (d_pair_list && ctx->pair_order_is_list && ctx->pair_order_count == n_pairs) d_pair_order = ctx->d_pair_order;      // surtr_fracture_pairs_async made it
    else if (!d_pair_list && n_pairs && ctx->n_pieces && n_pairs % ctx->n_pieces == 0 && !getenv("SURTR_NO_CELL_ORDER"))
    {
        const uint32_t np = ctx->n_pieces, nc = n_pairs / np;
        if (ctx->pair_order_is_list || ctx->pair_order_begin != cell_begin || ctx->pair_order_count != n_pairs)
        {
            std::vector<uint32_t> cells(nc), ord((size_t)n_pairs);
            for (uint32_t i = 0; i < nc; ++i) cells[i] = i;
            const std::vector<uint32_t>& po = ctx->h_plane_off;
            std::stable_sort(cells.begin(), cells.end(), [&](uint32_t a, uint32_t b) {
                return po[cell_begin + a + 1] - po[cell_begin + a] > po[cell_begin + b + 1] - po[cell_begin + b]; });
            for (uint32_t i = 0; i < nc; ++i) for (uint32_t q = 0; q < np; ++q) ord[(size_t)i * np + q] = cells[i] * np + q;
            int rc2 = upload_pair_order(ctx, ord.data(), n_pairs);
            if (rc2) return rc2;
            ctx->pair_order_is_list = false; ctx->pair_order_begin = cell_begin; ctx->pair_order_count = n_pairs;
        }
        d_pair_order = ctx->d_pair_order;
    }
    uint32_t big_quota = ctx->wave_big ? 0xFFFFFFFFu : 2u * ctx->n_wg_big;
    if (const char* e = getenv("SURTR_BIG_QUOTA")) big_quota = (uint32_t)atoi(e);      // (tests: 0 sends every big band to the regular kernel's global scratch)
    // (pieces of 80 000 vertices and more: bands beyond this size go to the whole-CU record clipper, see k_prep_pairs)
    // measured at 4 096 cells with 2 800: 100 000-vertex piece 6.50 -> 5.96 ms, 150 000 vertices 10.5 -> 9.6 ms, 210 000 vertices 12.6 -> 12.8 ms
    // (there nearly every band is beyond it and the regular kernel runs dry): applied below 180 000 vertices
    uint32_t big_n = (ctx->wave_big && ctx->vmax < 180000u) ? SURTR_WAVE_BIG_N : 0xFFFFFFFFu;
    if (const char* e = getenv("SURTR_WAVE_BIG_N")) big_n = (uint32_t)atoi(e);
    // the regular pairs through the record clipper (wave_clip.h) once the pairs queue up; k_prep_pairs then leaves their bands as
    // record images (rec_on)
    // (round 4, split arrangement, blocks of configs[3]: 2 048 pairs 1.83 against 2.23 ms for the event, 1 024 pairs 1.53 against 1.59,
    //  512 pairs 1.40 against 1.32: from 3/2 of the workgroup count on)
    // ... and for events of any size over pieces whose bands come as record images as a rule (every piece pre-passed by k_prep_pairs
    // and small enough that a band seldom passes SURTR_REC_MAXN vertices): one pair on the record clipper alone takes 0.18 ms where
    // the general clipper takes 0.26 (configs[1]: 0.73 -> 0.72 ms per event, configs[2]: 1.06 -> 1.00)
    // (SURTR_HALF=1 -- tests that want the half-size general kernel -- keeps such events on it)
    const bool half_forced = getenv("SURTR_HALF") != nullptr && atoi(getenv("SURTR_HALF")) != 0;
    // (several contexts busy on the GPU, surtr_set_events_in_flight: the lean arrangement for events of any size whose pieces the
    //  split arrangement takes -- the record clipper's 0.27 ms per pair against the general clipper's 0.55 leave the other events
    //  the LDS; blocks of configs[3] with four contexts: 512 cells 0.65 -> 0.55 ms per step, 1 024 cells 0.84 -> 0.75, 2 048 cells
    //  1.40 -> 1.21; one at a time 1.29 -> 1.37 / 1.45 -> 1.44 / 1.73 -> 1.78)
    const bool many = ctx->events_in_flight > 1u && !half_forced;
    bool wave_on = 2u * n_pairs > 3u * max_wg || (!half_forced && ctx->vmin >= SURTR_PREP_MINV && ctx->vmax <= 4u * SURTR_REC_MAXN) ||
                   (many && ctx->vmin >= SURTR_PREP_MINV && ctx->vmax <= ctx->prep.VMAX);
    if (const char* e = getenv("SURTR_WAVE")) wave_on = atoi(e) != 0;
    uint32_t rec_on = wave_on ? 1u : 0u;
    if (const char* e = getenv("SURTR_REC")) rec_on = (wave_on && atoi(e) != 0) ? 1u : 0u;      // (tests / A-B: 0 = images + wc_load as in round 3)
    if (const char* e = getenv("SURTR_PREP_SORTED")) { if (atoi(e) == 0) rec_on = 2u; }             // (tests / A-B: 0 = round 3's selection, prepass_select)
    // bands beyond this size keep the old image: they are the ones the record clipper runs out of room on, and a pair it gives up
    // on is then finished in place from that image instead of from the piece
    uint32_t rec_maxn = SURTR_REC_MAXN;
    if (const char* e = getenv("SURTR_REC_MAXN")) rec_maxn = (uint32_t)atoi(e);
    rec_on |= (rec_maxn < 0xFFFFFFu ? rec_maxn : 0xFFFFFFu) << 8;
    // the small tier of the record clipper: three workgroups per CU for the record images whose worst plane fits its LDS
    // OFF by default.  Measured on configs[3] (MI355X, round 4): the small tier's kernel has 114 registers, no private scratch and
    // 53 680 B of LDS (three workgroups per CU) and takes 2 400 of 3 335 pairs in 0.64 ms -- but the 900 heavy pairs left for the
    // large tier then take 1.02 ms on their own (two or three 0.4 ms pairs per workgroup: no light pairs left to level the end),
    // 1.66 ms for the two kernels one behind the other against 1.45 ms for the one kernel; 2.51 against 2.39 ms per step with
    // three events in flight.  Side by side on two streams the event took 2.58 ms when the two grids happened to interleave
    // and 2.98 ms when one filled the CUs first (SURTR_SMALL_CONC, timing only).  SURTR_SMALL=1 turns the tier on.
    uint32_t small_cap = 0u;
    if (const char* e = getenv("SURTR_SMALL")) { if (atoi(e) != 0 && wave_on && (rec_on & 1u)) small_cap = 16u * SURTR_WR_S; }
    // the split arrangement (k_clip_pairs_main + k_clip_pairs_catch, see there) for events whose every pair gets an image from
    // k_prep_pairs: pieces of fewer than SURTR_PREP_MINV vertices are pre-passed by the general clipper itself, which only the
    // one-kernel arrangement has on every workgroup
    bool split_on = wave_on && !small_cap && ctx->vmin >= SURTR_PREP_MINV && ctx->vmax <= ctx->prep.VMAX;
    if (const char* e = getenv("SURTR_SPLIT")) split_on = split_on && atoi(e) != 0;
    // (the split arrangement takes the light pairs too: the half-size general kernel stays out of it)
    const bool use_half = ctx->half_on && (!split_on || half_forced);
    // (measured on configs[3]: 16 .. 32 workgroups end with the main kernel -- 35 pairs with a vertex in a plane + a hand-over or two;
    //  64 and more take LDS from it: 2.50 / 2.50 / 2.56 ms per event with 16 / 32 / 64)
    uint32_t n_catch = std::min(std::min(ctx->n_wg_catch, 32u), std::max(n_pairs, 1u));
    if (const char* e = getenv("SURTR_CATCH_WG")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= ctx->n_wg_catch) n_catch = v; }
    // catcher workgroups that wait for hand-overs (tests: 0 = none, everything handed on is the sweep's).  With other contexts busy on
    // the GPU (surtr_set_events_in_flight) two: a polling workgroup holds 78 KB of LDS for the length of the main kernel, which the
    // other events' kernels want (configs[3], four contexts, 60 steps: 2.15 -> 2.09 ms per step; one poller: 2.08, but then the two
    // hand-overs of an event wait for each other, one event 2.38 -> 2.54 ms)
    uint32_t n_poll = ctx->events_in_flight > 1u ? 2u : SURTR_CATCH_POLL;
    if (const char* e = getenv("SURTR_CATCH_POLL")) { const int v = atoi(e); if (v >= 0 && v <= 1024) n_poll = (uint32_t)v; }
    uint32_t heavy_need = 0u;      // (off: k_clip_pairs_big takes 2 x its grid of such pairs and no more -- the rest would land on the catcher)
    if (const char* e = getenv("SURTR_HEAVY_NEED")) { if (split_on) heavy_need = (uint32_t)atoi(e); }
    uint32_t n_wg_rec = 0;
    if (small_cap)
    {
        n_wg_rec = std::max(1u, std::min(std::max(n_pairs, 1u), ctx->hw_wg / 2u * 3u));
        if (const char* e = getenv("SURTR_SMALL_WG")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= 4096u) n_wg_rec = v; }
        const size_t per = ((size_t)16u * 2u * SURTR_WR_S + 255u) & ~(size_t)255u;
        if (!(ctx->pool_rec.base && ctx->pool_rec.per_wg >= per && ctx->n_wg_rec >= n_wg_rec))
        {
            free_dev(ctx->pool_rec.base); ctx->pool_rec.base = nullptr;
            ctx->pool_rec.per_wg = per; ctx->pool_rec.CV = ctx->pool_rec.CH = ctx->pool_rec.VMAX = 0; ctx->n_wg_rec = n_wg_rec;
            HIPCHK(hipMalloc((void**)&ctx->pool_rec.base, per * n_wg_rec));
        }
    }
    // measured on blocks of configs[3]: 512 pairs 0.60 -> 0.23 ms, 1 024 pairs 0.70 -> 0.37, 2 048 pairs 0.77 -> 0.68, 4 096 pairs 0.96 -> 1.32
    uint32_t wide_max = 4u * ctx->max_wg;
    if (const char* e = getenv("SURTR_PREP_WIDE_MAX")) wide_max = (uint32_t)atoi(e);
    // The clip of the Convexes and the pre-pass of the Meshes are independent but for two things: the pre-pass skips the pairs whose
    // Convex came out empty, and takes the others from a queue by estimated cost that k_clip_convex builds.  An event of so few pairs
    // that every pair finds a free workgroup at once needs neither -- there the two kernels run side by side (front_par: the
    // pre-pass takes the pairs by index and prepares the empty ones too; the clip kernels wait for both and skip those).
    // Measured per event, one at a time: configs[1] 0.64 -> 0.58 ms, configs[2] 0.92 -> 0.875; not for the wide pre-pass of a block
    // of large pieces (1 024 threads a pair: beside k_clip_convex it takes twice as long, 512-cell block of configs[3] 1.16 -> 1.19 ms)
    // nor for events whose pairs queue up (configs[3]: 2.47 -> 2.71 ms, the cost order and the 761 skipped pairs are worth more).
    const bool prep_wide = n_pairs != 0 && n_pairs <= wide_max && ctx->vmax >= 8192u && !many;      // few pairs, large meshes, nothing else on the GPU
    bool front_par = n_pairs != 0 && n_pairs <= SURTR_FRONT_PAR_MAX && !prep_wide;
    if (const char* e = getenv("SURTR_FRONT_PAR")) front_par = n_pairs != 0 && atoi(e) != 0;
    hipStream_t st_cvx = st;
    if (front_par)
    {
        rec_on |= 4u;
        st_cvx = ctx->stream3;
        HIPCHK(hipEventRecord(ctx->ev_prep, st));      // (behind the memsets above)
        HIPCHK(hipStreamWaitEvent(st_cvx, ctx->ev_prep, 0));
    }
    PROF_BEGIN_ON(6, st_cvx);
    if (n_pairs)
        hipLaunchKernelGGL(k_clip_convex, dim3(n_wg_small), dim3(SURTR_LANES), 0, st_cvx, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           d_out, ctx->pool_small, ctx->arena, ctx->d_pairs, d_pair_list, front_par ? (uint32_t*)nullptr : ctx->d_order + (size_t)16 * ctx->cap_order, d_pair_order,
                           front_par ? 1u : 0u);
    PROF_END_ON(6, st_cvx);
    if (front_par) HIPCHK(hipEventRecord(ctx->ev_cvx, st_cvx));
    PROF_BEGIN(7);
    if (n_pairs && prep_wide)
        hipLaunchKernelGGL(k_prep_pairs_wide, dim3(n_wg_prep), dim3(SURTR_WG_WIDE), 0, st, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->prep, ctx->arena, ctx->img, std::min((uint32_t)SURTR_LV, ctx->pool.CV), std::min((uint32_t)SURTR_LVS, ctx->pool_half.CV), ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)16 * ctx->cap_order,
                           ctx->d_order + (size_t)32 * ctx->cap_order, use_half ? 1u : 0u, big_quota, big_n, rec_on, small_cap, heavy_need);
    else if (n_pairs && !(rec_on & 2u) && ctx->vmax < 0xFFFFu && ctx->vmin >= SURTR_PREP_MINV && (ctx->vmax + SURTR_LANES - 1u) / SURTR_LANES <= SURTR_PREP_NB)
        hipLaunchKernelGGL(k_prep_pairs_sorted, dim3(n_wg_prep), dim3(SURTR_WG), 0, st, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->prep, ctx->arena, ctx->img, std::min((uint32_t)SURTR_LV, ctx->pool.CV), std::min((uint32_t)SURTR_LVS, ctx->pool_half.CV), ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)16 * ctx->cap_order,
                           ctx->d_order + (size_t)32 * ctx->cap_order, use_half ? 1u : 0u, big_quota, big_n, rec_on, small_cap, heavy_need);
    else if (n_pairs)
        hipLaunchKernelGGL(k_prep_pairs, dim3(n_wg_prep), dim3(SURTR_WG), 0, st, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->prep, ctx->arena, ctx->img, std::min((uint32_t)SURTR_LV, ctx->pool.CV), std::min((uint32_t)SURTR_LVS, ctx->pool_half.CV), ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)16 * ctx->cap_order,
                           ctx->d_order + (size_t)32 * ctx->cap_order, use_half ? 1u : 0u, big_quota, big_n, rec_on, small_cap, heavy_need);
    PROF_END(7);
    // k_clip_pairs_big goes first on the caller's stream, right behind k_prep_pairs, so that its few whole-CU workgroups
    // are placed before k_clip_pairs (second stream) and k_clip_pairs_half (third) fill the CUs; all three run side by side.
    // (front_par: the main kernel stays on the caller's stream and the whole-CU kernel goes to the second -- every wait of one
    //  stream for another costs some 12 us, and there the main kernel is the critical path: it then waits only for k_clip_convex)
    hipStream_t st2 = ctx->stream2, st3 = ctx->stream3;
    hipStream_t st_main = front_par ? st : st2, st_bigk = front_par ? st2 : st;
    HIPCHK(hipEventRecord(ctx->ev_prep, st));
    HIPCHK(hipStreamWaitEvent(st2, ctx->ev_prep, 0));
    HIPCHK(hipStreamWaitEvent(st3, ctx->ev_prep, 0));
    if (front_par)
    {
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_cvx, 0));      // (stream3 has k_clip_convex in order)
        HIPCHK(hipStreamWaitEvent(st2, ctx->ev_cvx, 0));
    }
    PROF_BEGIN_ON(8, st_bigk);
    uint32_t walk0 = SURTR_WWALK0;
    if (const char* e = getenv("SURTR_WWALK0")) { const int v = atoi(e); if (v >= 0 && v <= 64) walk0 = (uint32_t)v; }
    if (n_pairs && ctx->wave_big)
        hipLaunchKernelGGL(k_clip_pairs_wave_big, dim3(std::min(ctx->n_wg_big, std::max(n_pairs, 1u))), dim3(SURTR_WG), 0, st_bigk, P, ctx->d_planes,
                           ctx->d_plane_off, cell_begin, n_pairs, ctx->pool, max_wg, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, 15, 14, 11u, walk0);
    else if (n_pairs)
        hipLaunchKernelGGL(k_clip_pairs_big, dim3(std::min(ctx->n_wg_big, std::max(n_pairs, 1u))), dim3(SURTR_WG), 0, st_bigk, P, ctx->d_planes,
                           ctx->d_plane_off, cell_begin, n_pairs, ctx->pool, max_wg, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order);
    PROF_END_ON(8, st_bigk);
    // the regular pairs on one wave each (wave_clip.h); what it hands on comes back through the retry launch below
    // (measured on blocks of configs[3]: the record clipper wins once the pairs queue up -- 4 096 pairs 1.88 -> 1.65 ms, 2 048 pairs
    // 2.35 -> 2.30 ms for the event -- and loses when every pair has a workgroup to itself: 1 024 pairs 1.63 -> 1.69 ms, 512 pairs
    // 1.33 -> 1.45 ms; its loader sorts the band, which the general clipper's image copy does not have to)
    hipStream_t st_rec = st_main;
    if (getenv("SURTR_SMALL_CONC")) st_rec = st3;      // (timing experiment only: the large tier then misses late hand-overs)
    PROF_BEGIN_ON(12, st_rec);
    if (n_pairs && wave_on && small_cap)
        hipLaunchKernelGGL(k_clip_pairs_rec, dim3(n_wg_rec), dim3(SURTR_S_THREADS), 0, st_rec, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool_rec, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list,
                           (const uint32_t*)(ctx->d_order + (size_t)32 * ctx->cap_order + (size_t)16 * n_pairs), ctx->d_order, walk0);
    PROF_END_ON(12, st_rec);
    PROF_BEGIN_ON(11, st_main);
    if (n_pairs && wave_on) PROF_HIST_BEGIN(11, st_main);
    uint32_t n_wg_main = n_wg;
    if (const char* e = getenv("SURTR_MAIN_WG")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= max_wg) n_wg_main = v; }
    if (n_pairs && wave_on && split_on)
        hipLaunchKernelGGL(k_clip_pairs_main, dim3(n_wg_main), dim3(SURTR_MAIN_THREADS), 0, st_main, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_hlist, walk0);
    else if (n_pairs && wave_on)
        hipLaunchKernelGGL(k_clip_pairs_wave, dim3(n_wg), dim3(SURTR_WG), 0, st_main, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)32 * ctx->cap_order, 13, 0, 4u, walk0);
    if (n_pairs && wave_on) PROF_HIST_END(11, st_main);
    PROF_END_ON(11, st_main);
    PROF_BEGIN_ON(0, st_main);
    if (n_pairs && !wave_on) PROF_HIST_BEGIN(0, st_main);
    if (n_pairs && !wave_on)
        hipLaunchKernelGGL(k_clip_pairs, dim3(n_wg), dim3(SURTR_WG), 0, st_main, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)32 * ctx->cap_order, 13, 0, 4u);
    if (n_pairs && !wave_on) PROF_HIST_END(0, st_main);
    PROF_END_ON(0, st_main);
    PROF_BEGIN_ON(13, st3);
    if (n_pairs && wave_on && split_on)
        hipLaunchKernelGGL(k_clip_pairs_catch, dim3(n_catch), dim3(SURTR_WG), 0, st3, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, max_wg + ctx->n_wg_big, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_hlist, hcap, n_wg_main, 0u, n_poll);
    PROF_END_ON(13, st3);
    PROF_BEGIN_ON(9, st3);
    if (n_pairs && use_half)
        hipLaunchKernelGGL(k_clip_pairs_half, dim3(n_wg_half), dim3(SURTR_WGS), 0, st3, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool_half, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order + (size_t)32 * ctx->cap_order);
    PROF_END_ON(9, st3);
    HIPCHK(hipEventRecord(ctx->ev_half, st3));
    HIPCHK(hipStreamWaitEvent(st_main, ctx->ev_half, 0));
    // the pairs that outgrew the half-size topology (class 0, normally none): the regular kernel once more, behind both
    // (it reuses the scratch slots of the first launch)
    PROF_BEGIN_ON(10, st_main);
    if (n_pairs && use_half)
        hipLaunchKernelGGL(k_clip_pairs, dim3(std::min(n_wg, 64u)), dim3(SURTR_WG), 0, st_main, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_order + (size_t)32 * ctx->cap_order, -1, 0, 13u);
    // ... and the sweep of the hand-over list (see k_clip_pairs_catch): normally nothing is left and its workgroups return at once
    if (n_pairs && wave_on && split_on)
        hipLaunchKernelGGL(k_clip_pairs_catch, dim3(std::min(n_catch, SURTR_CATCH_POLL)), dim3(SURTR_WG), 0, st_main, P, ctx->d_planes, ctx->d_plane_off, cell_begin, n_pairs,
                           ctx->pool, max_wg + ctx->n_wg_big, ctx->arena, ctx->img, ctx->d_pairs, d_pair_list, ctx->d_order, ctx->d_hlist, hcap, n_wg_main, 1u, 0u);
    PROF_END_ON(10, st_main);
    HIPCHK(hipEventRecord(ctx->ev_big, st2));      // (the main kernel's stream, or the whole-CU kernel's when the main one runs on `st`)
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_big, 0));
    PROF_BEGIN(1);
    hipLaunchKernelGGL(k_frag_table, dim3(1), dim3(SURTR_WG_WIDE), 0, st, ctx->d_pairs, n_pairs, ctx->n_pieces, cell_begin, ctx->arena,
                       ctx->d_scanblk, ctx->d_frags, ctx->cap_frags, ctx->d_counts, d_pair_list, ctx->d_forder);
    PROF_END(1);
    // refit (Convex) and faces (Mesh) of the fragments are independent: side by side on the two streams
    const bool both = (flags & SURTR_EVT_REFIT) && (flags & SURTR_EVT_RENDER);
    hipStream_t st_refit = st;
    if (both)
    {
        HIPCHK(hipEventRecord(ctx->ev_prep, st));
        HIPCHK(hipStreamWaitEvent(st2, ctx->ev_prep, 0));
        st_refit = st2;
    }
    if (flags & SURTR_EVT_REFIT)
    {
        PROF_BEGIN_ON(2, st_refit);
        // side by side the two kernels share the CUs' registers and LDS: k_faces at two workgroups per CU leaves room for five
        // of k_refit's (measured on configs[3]: 3.72 -> 3.65 ms per event against both at their stand-alone sizes)
        uint32_t g_refit = ctx->n_wg_small;
        if (both) g_refit = std::min(g_refit, ctx->max_wg_faces / 4u * 5u);
        if (both) if (const char* e = getenv("SURTR_REFIT_WG_BOTH")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= ctx->n_wg_small) g_refit = v; }
        hipLaunchKernelGGL(k_refit, dim3(g_refit), dim3(SURTR_LANES), 0, st_refit, ctx->d_frags, ctx->d_counts, ctx->pool_small, ctx->arena, ctx->d_forder, ctx->cap_frags, ctx->d_frag_status,
                           (const float*)ctx->cset.pos, (const uint32_t*)ctx->cset.vo);
        PROF_END_ON(2, st_refit);
    }
    if (flags & SURTR_EVT_RENDER)
    {
        PROF_BEGIN(3);
        uint32_t g_faces = std::max(1u, ctx->max_wg_faces), t_faces = SURTR_WG;
        if (both) g_faces = std::max(1u, ctx->max_wg_faces / 2u);
        if (both) if (const char* e = getenv("SURTR_FACES_WG_BOTH")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= ctx->n_wg_faces_alloc) g_faces = v; }
        if (const char* e = getenv("SURTR_FACES_WG")) { const uint32_t v = (uint32_t)atoi(e); if (v > 0 && v <= ctx->n_wg_faces_alloc) g_faces = v; }
        if (const char* e = getenv("SURTR_FACES_THREADS")) { const uint32_t v = (uint32_t)atoi(e); if (v == 64 || v == 128 || v == 256) t_faces = v; }
        const bool tiers = ctx->n_wg_faces_big != 0;
        hipLaunchKernelGGL(k_faces, dim3(g_faces), dim3(t_faces), 0, st, ctx->d_frags, ctx->d_counts, ctx->fs, ctx->d_blk,
                           ctx->blk_per_wg, ctx->arena, ctx->d_forder, ctx->cap_frags, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr,
                           ctx->d_frag_status, (const uint32_t*)nullptr, tiers ? ctx->d_face_list : (uint32_t*)nullptr);
        if (tiers)      // the fragments too large for the scratch of those workgroups
            hipLaunchKernelGGL(k_faces, dim3(ctx->n_wg_faces_big), dim3(t_faces), 0, st, ctx->d_frags, ctx->d_counts, ctx->fs_big, ctx->d_blk_big,
                               ctx->blk_per_wg_big, ctx->arena, ctx->d_forder, ctx->cap_frags, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr,
                               ctx->d_frag_status, (const uint32_t*)ctx->d_face_list, (uint32_t*)nullptr);
        PROF_END(3);
    }
    if (both)
    {
        HIPCHK(hipEventRecord(ctx->ev_big, st2));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_big, 0));
    }
    PROF_BEGIN(4);
    hipLaunchKernelGGL(k_out_scan, dim3(1), dim3(SURTR_WG_WIDE), 0, st, ctx->d_frags, ctx->d_scanblk, ctx->d_counts, ctx->arena);
    PROF_END(4);
    HIPCHK(hipGetLastError());
    ctx->have_event = true; ctx->last_flags = flags; ctx->last_current = false; ctx->frags_of_pieces = true;
    return SURTR_OK;
}

int surtr_fracture_event_async(surtr_ctx* ctx, uint32_t cell_begin, uint32_t cell_end, const uint8_t* outside, uint32_t flags)
{
    if (!ctx) return SURTR_E_INVALID;
    if (!ctx->n_pieces || !ctx->planes_ready) return SURTR_E_STATE;
    if (cell_end > ctx->n_cells || cell_begin > cell_end) return SURTR_E_INVALID;
    return launch_event(ctx, cell_begin, (cell_end - cell_begin) * ctx->n_pieces, nullptr, outside, flags);
}

int surtr_fracture_pairs_async(surtr_ctx* ctx, uint32_t n_pairs, const uint32_t* pair_cell, const uint32_t* pair_piece, uint32_t flags)
{
    if (!ctx || (n_pairs && (!pair_cell || !pair_piece))) return SURTR_E_INVALID;
    if (!ctx->n_pieces || !ctx->planes_ready) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    std::vector<uint2> list(n_pairs);
    for (uint32_t i = 0; i < n_pairs; ++i)
    {
        if (pair_cell[i] >= ctx->n_cells || pair_piece[i] >= ctx->n_pieces) return SURTR_E_INVALID;
        list[i].x = pair_cell[i]; list[i].y = pair_piece[i];
    }
    if (ctx->cap_pair_list < std::max(n_pairs, 1u))
    {
        free_dev(ctx->d_pair_list); ctx->d_pair_list = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->d_pair_list, (size_t)std::max(n_pairs, 1u) * sizeof(uint2)));
        ctx->cap_pair_list = std::max(n_pairs, 1u);
    }
    if (n_pairs) HIPCHK(hipMemcpyAsync(ctx->d_pair_list, list.data(), (size_t)n_pairs * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));     // `list` is a stack-owned staging buffer
    ctx->pair_order_count = 0;
    if (n_pairs && !getenv("SURTR_NO_CELL_ORDER"))
    {
        // queue order of k_clip_convex: pairs by plane count of their cell, descending (counting sort, stable)
        const std::vector<uint32_t>& po = ctx->h_plane_off;
        uint32_t start[SURTR_MAXF + 2] = {0};
        for (uint32_t i = 0; i < n_pairs; ++i) ++start[SURTR_MAXF - (po[pair_cell[i] + 1] - po[pair_cell[i]]) + 1];
        for (uint32_t k = 1; k <= SURTR_MAXF + 1; ++k) start[k] += start[k - 1];
        std::vector<uint32_t> ord(n_pairs);
        for (uint32_t i = 0; i < n_pairs; ++i) ord[start[SURTR_MAXF - (po[pair_cell[i] + 1] - po[pair_cell[i]])]++] = i;
        int rc2 = upload_pair_order(ctx, ord.data(), n_pairs);
        if (rc2) return rc2;
        ctx->pair_order_is_list = true; ctx->pair_order_count = n_pairs;
    }
    return launch_event(ctx, 0, n_pairs, ctx->d_pair_list, nullptr, flags);
}

// face -> group table of the pattern (cached per pattern) and the placement kernel; scale3 / shift3 are device arrays
int surtr_place_cells_groups_dev(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off, const float* d_scale3, const float* d_shift3)
{
    if (!ctx || !n_groups || !group_cell_off || !d_scale3 || !d_shift3) return SURTR_E_INVALID;
    if (!ctx->d_v012) return SURTR_E_STATE;
    if (group_cell_off[0] != 0 || group_cell_off[n_groups] != ctx->n_cells) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const uint32_t nf = ctx->n_faces;
    std::vector<uint32_t> face_group(std::max(nf, 1u));
    for (uint32_t g = 0; g < n_groups; ++g)
    {
        if (group_cell_off[g + 1] < group_cell_off[g]) return SURTR_E_INVALID;
        for (uint32_t c = group_cell_off[g]; c < group_cell_off[g + 1]; ++c)
            for (uint32_t f = ctx->h_plane_off[c]; f < ctx->h_plane_off[c + 1]; ++f) face_group[f] = g;
    }
    if (ctx->cap_face_group < std::max(nf, 1u))
    {
        free_dev(ctx->d_face_group); ctx->d_face_group = nullptr; ctx->cap_face_group = 0;
        HIPCHK(hipMalloc((void**)&ctx->d_face_group, (size_t)(nf + nf / 4 + 64) * 4));
        ctx->cap_face_group = nf + nf / 4 + 64;
    }
    HIPCHK(hipMemcpyAsync(ctx->d_face_group, face_group.data(), (size_t)nf * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nf)
        hipLaunchKernelGGL(k_place_cells_groups, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, nf, ctx->d_v012, ctx->d_face_group, d_scale3, d_shift3, ctx->d_planes);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));      // face_group is a stack-owned staging buffer
    ctx->planes_ready = true;
    return SURTR_OK;
}

int surtr_place_cells_groups(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off, const float* scale3, const float* translate3)
{
    if (!ctx || !n_groups || !group_cell_off || !scale3 || !translate3) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    float* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)n_groups * 24) != hipSuccess) return SURTR_E_HIP;
    hipError_t e = hipMemcpy(d, scale3, (size_t)n_groups * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 3 * (size_t)n_groups, translate3, (size_t)n_groups * 12, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? surtr_place_cells_groups_dev(ctx, n_groups, group_cell_off, d, d + 3 * (size_t)n_groups) : SURTR_E_HIP;
    free_dev(d);
    return rc;
}

int surtr_event_refit(surtr_ctx* ctx)
{
    if (!ctx) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    // k_refit pulls fragments from work queue 6
    HIPCHK(hipMemsetAsync(ctx->arena.cursors + 6, 0, 4, st));
    PROF_BEGIN(2);
    hipLaunchKernelGGL(k_refit, dim3(ctx->n_wg_small), dim3(SURTR_LANES), 0, st, ctx->d_frags, ctx->d_counts, ctx->pool_small, ctx->arena, ctx->d_forder, ctx->cap_frags, ctx->d_frag_status,
                       ctx->frags_of_pieces ? (const float*)ctx->cset.pos : (const float*)nullptr, ctx->frags_of_pieces ? (const uint32_t*)ctx->cset.vo : (const uint32_t*)nullptr);
    PROF_END(2);
    hipLaunchKernelGGL(k_out_scan, dim3(1), dim3(SURTR_WG_WIDE), 0, st, ctx->d_frags, ctx->d_scanblk, ctx->d_counts, ctx->arena);
    HIPCHK(hipGetLastError());
    ctx->last_flags |= SURTR_EVT_REFIT; ctx->last_current = false;
    return SURTR_OK;
}

// k_faces on the current fragments (an event's, or those of surtr_load_fragments).
static int launch_faces(surtr_ctx* ctx, uint32_t fan, uint32_t* d_face_n, uint32_t* d_face_off, int32_t* d_face_idx)
{
    hipStream_t st = ctx->stream;
    // queue 7 = fragments for k_faces, cursor 2 = index arena, 14 = flagged fragments (flagged pairs are counted in 15: they stay)
    HIPCHK(hipMemsetAsync(ctx->arena.cursors + 7, 0, 4, st));
    HIPCHK(hipMemsetAsync(ctx->arena.cursors + 2, 0, 4, st));
    if (!(ctx->last_flags & SURTR_EVT_REFIT))      // (a refit of these fragments may have flagged some: those flags stay)
    {
        HIPCHK(hipMemsetAsync(ctx->arena.cursors + 14, 0, 4, st));
        HIPCHK(hipMemsetAsync(ctx->d_frag_status, 0, (size_t)ctx->cap_frags * 4, st));
    }
    PROF_BEGIN(3);
    const bool tiers = ctx->n_wg_faces_big != 0;
    if (tiers) HIPCHK(hipMemsetAsync(ctx->arena.cursors + 85, 0, 8, st));      // the second tier's list: fragments pushed (85), tickets taken (86)
    hipLaunchKernelGGL(k_faces, dim3(std::max(1u, ctx->max_wg_faces)), dim3(SURTR_WG), 0, st, ctx->d_frags, ctx->d_counts, ctx->fs, ctx->d_blk,
                       ctx->blk_per_wg, ctx->arena, ctx->d_forder, ctx->cap_frags, fan, d_face_n, d_face_off, d_face_idx, ctx->d_frag_status,
                       (const uint32_t*)nullptr, tiers ? ctx->d_face_list : (uint32_t*)nullptr);
    if (tiers)
        hipLaunchKernelGGL(k_faces, dim3(ctx->n_wg_faces_big), dim3(SURTR_WG), 0, st, ctx->d_frags, ctx->d_counts, ctx->fs_big, ctx->d_blk_big,
                           ctx->blk_per_wg_big, ctx->arena, ctx->d_forder, ctx->cap_frags, fan, d_face_n, d_face_off, d_face_idx, ctx->d_frag_status,
                           (const uint32_t*)ctx->d_face_list, (uint32_t*)nullptr);
    PROF_END(3);
    hipLaunchKernelGGL(k_out_scan, dim3(1), dim3(SURTR_WG_WIDE), 0, st, ctx->d_frags, ctx->d_scanblk, ctx->d_counts, ctx->arena);
    HIPCHK(hipGetLastError());
    ctx->last_flags |= SURTR_EVT_RENDER; ctx->last_current = false;
    return SURTR_OK;
}

int surtr_event_triangulate(surtr_ctx* ctx, int is_convex)
{
    if (!ctx) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    return launch_faces(ctx, is_convex ? 1u : 0u, nullptr, nullptr, nullptr);
}

int surtr_load_fragments(surtr_ctx* ctx, uint32_t n, const uint32_t* mvo, const float* mpos, const uint32_t* moff, const int32_t* mnbr,
                         const uint32_t* cvo, const float* cpos, const uint32_t* coff, const int32_t* cnbr, const int32_t* frag_ids)
{
    if (!ctx || n == 0 || !mvo || !mpos || !moff || !mnbr || !cvo || !cpos || !coff || !cnbr) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const uint32_t MV = mvo[n], CV = cvo[n];
    const uint32_t MH = moff[MV], CH = coff[CV];
    uint32_t vmax = 0, hmax = 0, cvmax = 0, chmax = 0;
    for (uint32_t k = 0; k < n; ++k)
    {
        for (int set = 0; set < 2; ++set)
        {
            const uint32_t* vo = set ? cvo : mvo; const uint32_t* off = set ? coff : moff; const int32_t* nbr = set ? cnbr : mnbr;
            const uint32_t a = vo[k], b = vo[k + 1];
            if (b < a || b - a < 4) return SURTR_E_INVALID;
            std::vector<uint32_t> loc(b - a + 1);
            for (uint32_t v = a; v <= b; ++v) loc[v - a] = off[v] - off[a];
            const int rc = check_solid(b - a, loc.data(), nbr + off[a]);
            if (rc) return rc;
            if (set) { cvmax = std::max(cvmax, b - a); chmax = std::max(chmax, off[b] - off[a]); }
            else { vmax = std::max(vmax, b - a); hmax = std::max(hmax, off[b] - off[a]); }
        }
    }
    ctx->cvmax = std::max(ctx->cvmax, cvmax); ctx->chmax = std::max(ctx->chmax, chmax);
    ctx->vmax = std::max(ctx->vmax, std::max(vmax, cvmax)); ctx->hmax = std::max(ctx->hmax, std::max(hmax, chmax));
    int rc = ensure_scratch(ctx, ctx->vmax, ctx->hmax, std::max(1u, ctx->n_wg));
    if (rc) return rc;
    rc = ensure_scratch_small(ctx, std::max(ctx->max_wg_small, ctx->n_wg_small));
    if (rc) return rc;
    // the solids, the refitted Convex solids (a clip by 8 planes adds a few vertices per plane), 3 indices per half-edge at most
    rc = ensure_arena(ctx, n, (uint64_t)MV + 3ull * CV + 64ull * n, (uint64_t)MH + 4ull * CH + 256ull * n, 3ull * MH + 64);
    if (rc) return rc;
    if (n > ctx->cap_frags) return SURTR_E_CAPACITY;
    hipStream_t st = ctx->stream;
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> loff((size_t)MV + CV), llen((size_t)MV + CV);
    for (uint32_t v = 0; v < MV; ++v) { loff[v] = moff[v]; llen[v] = moff[v + 1] - moff[v]; }
    for (uint32_t v = 0; v < CV; ++v) { loff[MV + v] = MH + coff[v]; llen[MV + v] = coff[v + 1] - coff[v]; }
    std::vector<FragRec> fr(n);
    std::vector<uint32_t> cursors(128, 0u), forder((size_t)16 * ctx->cap_frags, 0u);
    for (uint32_t k = 0; k < n; ++k)
    {
        FragRec& r = fr[k];
        memset(&r, 0, sizeof(r));
        r.cell = frag_ids ? frag_ids[3 * k] : (int32_t)k; r.piece = frag_ids ? frag_ids[3 * k + 1] : 0; r.island = frag_ids ? frag_ids[3 * k + 2] : 0;
        r.mv_off = mvo[k]; r.mv_n = mvo[k + 1] - mvo[k]; r.mh_off = moff[mvo[k]]; r.mh_n = moff[mvo[k + 1]] - moff[mvo[k]];
        r.cv_off = MV + cvo[k]; r.cv_n = cvo[k + 1] - cvo[k]; r.ch_off = MH + coff[cvo[k]]; r.ch_n = coff[cvo[k + 1]] - coff[cvo[k]];
        uint32_t cls = 0; while ((r.mv_n >> (cls + 1u)) != 0u && cls < 15u) ++cls;      // size classes of k_frag_table
        forder[(size_t)cls * ctx->cap_frags + cursors[32u + cls]++] = k;
    }
    cursors[0] = MV + CV; cursors[1] = MH + CH;
    surtr_counts c; memset(&c, 0, sizeof(c)); c.n_frag = n; c.n_pairs = n;
    HIPCHK(hipMemcpyAsync(ctx->arena.pos, mpos, (size_t)MV * 12, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.pos + 3 * (size_t)MV, cpos, (size_t)CV * 12, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.loff, loff.data(), loff.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.llen, llen.data(), llen.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.nbr, mnbr, (size_t)MH * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.nbr + MH, cnbr, (size_t)CH * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_frags, fr.data(), (size_t)n * sizeof(FragRec), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_forder, forder.data(), forder.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->arena.cursors, cursors.data(), 512, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_counts, &c, sizeof(c), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(ctx->d_frag_status, 0, (size_t)ctx->cap_frags * 4, st));
    hipLaunchKernelGGL(k_out_scan, dim3(1), dim3(SURTR_WG_WIDE), 0, st, ctx->d_frags, ctx->d_scanblk, ctx->d_counts, ctx->arena);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));       // the staging vectors go out of scope
    ctx->have_event = true; ctx->last_flags = 0; ctx->last_current = false; ctx->frags_of_pieces = false;
    return SURTR_OK;
}

// One solid as the only fragment (its own Mesh; `conv` or the solid again as the Convex).
static int load_one(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* off, const int32_t* nbr,
                    uint32_t cnv, const float* cpos, const uint32_t* coff, const int32_t* cnbr)
{
    const uint32_t mvo[2] = {0u, nv}, cvo[2] = {0u, cnv};
    return surtr_load_fragments(ctx, 1, mvo, pos, off, nbr, cvo, cpos, coff, cnbr, nullptr);
}

int surtr_refit_solid(surtr_ctx* ctx, uint32_t mnv, const float* mpos, const uint32_t* moff, const int32_t* mnbr,
                      uint32_t cnv, const float* cpos, const uint32_t* coff, const int32_t* cnbr,
                      uint32_t* out_nv, uint32_t* out_nh, float* out_pos, uint32_t* out_off, int32_t* out_nbr)
{
    if (!ctx || !mpos || !moff || !mnbr || !cpos || !coff || !cnbr || mnv < 4 || cnv < 4) return SURTR_E_INVALID;
    int rc = load_one(ctx, mnv, mpos, moff, mnbr, cnv, cpos, coff, cnbr);
    if (rc) return rc;
    rc = surtr_event_refit(ctx);
    if (rc) return rc;
    surtr_counts c;
    rc = surtr_event_counts(ctx, &c);
    if (rc) return rc;
    if (c.n_failed != 0)      // one solid was asked for: the reference's clip of it by the slabs is no polyhedron
    {
        ctx->err = "refit: the reference's result for this solid is not a polyhedron";
        return SURTR_E_TOPOLOGY;
    }
    if (out_nv) *out_nv = c.conv_verts;
    if (out_nh) *out_nh = c.conv_nbrs;
    if (!out_pos && !out_off && !out_nbr) return SURTR_OK;
    std::vector<uint32_t> cvo(2);
    surtr_fragments fr; memset(&fr, 0, sizeof(fr));
    fr.conv_vert_off = cvo.data(); fr.conv_pos = out_pos; fr.conv_nbr_off = out_off; fr.conv_nbr = out_nbr;
    return surtr_event_download(ctx, &fr);
}

int surtr_extract_faces(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* off, const int32_t* nbr,
                        uint32_t* n_faces, uint32_t* n_face_idx, uint32_t* face_off, int32_t* face_idx)
{
    if (!ctx || !pos || !off || !nbr || nv < 4) return SURTR_E_INVALID;
    int rc = load_one(ctx, nv, pos, off, nbr, nv, pos, off, nbr);
    if (rc) return rc;
    const uint32_t H = off[nv];
    uint32_t *d_n = nullptr, *d_off = nullptr; int32_t* d_idx = nullptr;
    auto cleanup = [&]() { free_dev(d_n); free_dev(d_off); free_dev(d_idx); };
    if (hipMalloc((void**)&d_n, 16) != hipSuccess || hipMalloc((void**)&d_off, ((size_t)H + 2) * 4) != hipSuccess ||
        hipMalloc((void**)&d_idx, ((size_t)std::max(ctx->fs.HF, ctx->fs_big.base ? ctx->fs_big.HF : 0u) + 2) * 4) != hipSuccess) { cleanup(); return SURTR_E_HIP; }
    (void)hipMemsetAsync(d_n, 0, 16, ctx->stream);
    rc = launch_faces(ctx, 0u, d_n, d_off, d_idx);
    surtr_counts c;
    if (rc == 0) rc = surtr_event_counts(ctx, &c);
    uint32_t cnt[2] = {0u, 0u};
    if (rc == 0 && hipMemcpy(cnt, d_n, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SURTR_E_HIP;
    if (rc == 0 && c.n_failed != 0) rc = SURTR_E_TOPOLOGY;       // the reference's walk never ends on this solid
    if (rc == 0)
    {
        if (n_faces) *n_faces = cnt[0];
        if (n_face_idx) *n_face_idx = cnt[1];
        if (face_off && hipMemcpy(face_off, d_off, ((size_t)cnt[0] + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SURTR_E_HIP;
        if (rc == 0 && face_idx && cnt[1] && hipMemcpy(face_idx, d_idx, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = SURTR_E_HIP;
    }
    cleanup();
    return rc;
}

int surtr_triangulate(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* off, const int32_t* nbr, int is_convex,
                      const float color[3], float* vnc, uint32_t* n_idx, uint32_t* idx)
{
    if (!ctx || !pos || !off || !nbr || nv < 4) return SURTR_E_INVALID;
    int rc = load_one(ctx, nv, pos, off, nbr, nv, pos, off, nbr);
    if (rc) return rc;
    rc = launch_faces(ctx, is_convex ? 1u : 0u, nullptr, nullptr, nullptr);
    if (rc) return rc;
    surtr_counts c;
    rc = surtr_event_counts(ctx, &c);
    if (rc) return rc;
    if (c.n_failed != 0) return SURTR_E_TOPOLOGY;
    if (n_idx) *n_idx = c.n_idx;
    if (!vnc && !idx) return SURTR_OK;
    for (int k = 0; k < 3; ++k) ctx->color[k] = color ? color[k] : 0.25f;
    std::vector<uint32_t> io(2);
    surtr_fragments fr; memset(&fr, 0, sizeof(fr));
    fr.vnc = vnc; fr.idx_off = io.data(); fr.idx = idx;
    rc = surtr_event_download(ctx, &fr);
    for (int k = 0; k < 3; ++k) ctx->color[k] = 0.25f;
    return rc;
}

int surtr_event_counts(surtr_ctx* ctx, surtr_counts* counts)
{
    if (!ctx || !counts) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipMemcpyAsync(&ctx->last, ctx->d_counts, sizeof(surtr_counts), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *counts = ctx->last; ctx->last_current = true;
    return ctx->last.status ? (int)ctx->last.status : SURTR_OK;
}

int surtr_fracture_event(surtr_ctx* ctx, uint32_t cell_begin, uint32_t cell_end, const uint8_t* outside, uint32_t flags,
                         surtr_counts* counts)
{
    int rc = surtr_fracture_event_async(ctx, cell_begin, cell_end, outside, flags);
    if (rc) return rc;
    surtr_counts c;
    rc = surtr_event_counts(ctx, &c);
    if (counts) *counts = c;
    return rc;
}

size_t surtr_event_blob_bytes(const surtr_counts* counts)
{
    if (!counts) return 0;
    return blob_layout(*counts).total;
}

int surtr_event_pack_dev(surtr_ctx* ctx, void* dev_blob, size_t capacity)
{
    if (!ctx || !dev_blob) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    // capacity is checked on the device against the counts it holds (k_pack reports a blob that does not fit in its header
    // and in the event's status word); when the host already holds this event's counts it refuses here
    if (ctx->last_current && blob_layout(ctx->last).total > capacity) return SURTR_E_CAPACITY;
    const uint32_t grid = std::max(1u, std::min(ctx->cap_frags, 2048u));
    hipStream_t st = ctx->stream;
    PROF_BEGIN(5);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(SURTR_WG), 0, ctx->stream, ctx->d_frags, ctx->d_counts, ctx->arena, (char*)dev_blob,
                       capacity, (ctx->last_flags & SURTR_EVT_RENDER) ? 1u : 0u, ctx->d_frag_status,
                       ctx->color[0], ctx->color[1], ctx->color[2]);
    PROF_END(5);
    HIPCHK(hipGetLastError());
    return SURTR_OK;
}

#ifdef SURTR_STAMP
int surtr_debug_stamps2(unsigned long long out[64], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp2), sizeof(unsigned long long) * 64) != hipSuccess) return SURTR_E_HIP;
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp2), z, sizeof(z)); }
    return SURTR_OK;
}
int surtr_debug_wplane(unsigned long long out[32], int reset)
{
#ifdef SURTR_STAMP
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wplane), sizeof(unsigned long long) * 32) != hipSuccess) return SURTR_E_HIP;
    if (reset) { unsigned long long z[32] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wplane), z, sizeof(z)); }
    return SURTR_OK;
#else
    (void)out; (void)reset;
    return SURTR_E_STATE;
#endif
}

int surtr_debug_wneed(uint32_t* out, uint32_t n_words)
{
#ifdef SURTR_STAMP
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wneed), sizeof(uint32_t) * std::min<uint32_t>(n_words, 4u * 8192u)) != hipSuccess) return SURTR_E_HIP;
    return SURTR_OK;
#else
    (void)out; (void)n_words;
    return SURTR_E_STATE;
#endif
}

int surtr_debug_stamps_wave(unsigned long long out[64], int reset)
{
#ifdef SURTR_STAMP
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamp), sizeof(unsigned long long) * 64) != hipSuccess) return SURTR_E_HIP;
    if (reset) { unsigned long long z[64] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp), z, sizeof(z)); }
    return SURTR_OK;
#else
    (void)out; (void)reset;
    return SURTR_E_STATE;
#endif
}

int surtr_debug_stamps(unsigned long long out[96], int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 96) != hipSuccess) return SURTR_E_HIP;
    if (reset) { unsigned long long z[96] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
    return SURTR_OK;
}
#endif

int surtr_set_events_in_flight(surtr_ctx* ctx, uint32_t n)
{
    if (!ctx) return SURTR_E_INVALID;
    ctx->events_in_flight = n ? n : 1u;
    return SURTR_OK;
}

int surtr_set_profiling(surtr_ctx* ctx, int on)
{
    if (!ctx) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    if (on && !ctx->ev[0])
        for (int i = 0; i < 32; ++i) HIPCHK(hipEventCreate(&ctx->ev[i]));
    if (on && !ctx->hev[0][0])
        for (int i = 0; i < 16; ++i) { HIPCHK(hipEventCreate(&ctx->hev[i][0])); HIPCHK(hipEventCreate(&ctx->hev[i][1])); }
    ctx->hcount = 0;
    ctx->profiling = on != 0;
    return SURTR_OK;
}

int surtr_kernel_times(surtr_ctx* ctx, float ms[16])
{
    if (!ctx || !ms) return SURTR_E_INVALID;
    for (int i = 0; i < 16; ++i) ms[i] = -1.f;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 16; ++i)
        if (ctx->ev_valid[i]) { float t = 0.f; if (hipEventElapsedTime(&t, ctx->ev[2 * i], ctx->ev[2 * i + 1]) == hipSuccess) ms[i] = t; }
    return SURTR_OK;
}

int surtr_kernel_history(surtr_ctx* ctx, float ms[16], int slot[16], uint32_t* n)
{
    if (!ctx || !ms || !slot || !n) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const uint32_t have = ctx->hcount < 16u ? ctx->hcount : 16u;
    for (uint32_t k = 0; k < have; ++k)
    {
        const uint32_t at = (ctx->hcount - have + k) % 16u;      // oldest first
        float t = -1.f;
        if (hipEventElapsedTime(&t, ctx->hev[at][0], ctx->hev[at][1]) != hipSuccess) t = -1.f;
        ms[k] = t; slot[k] = ctx->hslot[at];
    }
    *n = have;
    return SURTR_OK;
}

int surtr_pair_status(surtr_ctx* ctx, uint32_t n_pairs, uint32_t* status)
{
    if (!ctx || !status) return SURTR_E_INVALID;
    if (!ctx->have_event || n_pairs > ctx->cap_pairs) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<PairRec> recs(n_pairs);
    if (n_pairs) HIPCHK(hipMemcpy(recs.data(), ctx->d_pairs, (size_t)n_pairs * sizeof(PairRec), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n_pairs; ++i) status[i] = recs[i].status;
    return SURTR_OK;
}

int surtr_event_pair_costs(surtr_ctx* ctx, uint32_t n_pairs, uint32_t* cost)
{
    if (!ctx || !cost) return SURTR_E_INVALID;
    if (!ctx->have_event || n_pairs > ctx->cap_pairs) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<PairRec> recs(n_pairs);
    if (n_pairs) HIPCHK(hipMemcpy(recs.data(), ctx->d_pairs, (size_t)n_pairs * sizeof(PairRec), hipMemcpyDeviceToHost));
    // what the Mesh clip of the pair worked on: vertices of the band (the image k_prep_pairs left; the whole Mesh when the pair
    // had none) + the vertices that came out, as a proxy for the cut points it went through.  A pair whose Convex came out empty
    // costs its Convex clip only.
    for (uint32_t i = 0; i < n_pairs; ++i)
    {
        const PairRec& r = recs[i];
        uint32_t c = 8u;
        if (r.cv_n != 0) c += (r.img_fmt == IMG_NARROW || r.img_fmt == IMG_WIDE ? r.img_n : 256u) + 4u * r.mv_n + 64u;
        cost[i] = c;
    }
    return SURTR_OK;
}

int surtr_queue_stats(surtr_ctx* ctx, uint32_t out[128])
{
    if (!ctx || !out) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(out, ctx->arena.cursors, 512, hipMemcpyDeviceToHost));
    return SURTR_OK;
}

int surtr_blob_unpack_host(const void* blob, size_t bytes, surtr_counts* counts, surtr_fragments* out)
{
    if (!blob || bytes < 64) return SURTR_E_INVALID;
    surtr_counts c;
    memcpy(&c, blob, sizeof(c));
    const BlobLayout L = blob_layout(c);
    if (L.total > bytes) return SURTR_E_CAPACITY;
    if (counts) *counts = c;
    if (!out) return SURTR_OK;
    const char* b = (const char*)blob;
    auto cp = [&](void* dst, size_t off, size_t n) { if (dst && n) memcpy(dst, b + off, n); };
    cp(out->frag_ids, L.ids, (size_t)c.n_frag * 12);
    cp(out->mesh_vert_off, L.mvo, ((size_t)c.n_frag + 1) * 4);
    cp(out->mesh_pos, L.mpos, (size_t)c.mesh_verts * 12);
    cp(out->mesh_nbr_off, L.mno, ((size_t)c.mesh_verts + 1) * 4);
    cp(out->mesh_nbr, L.mnbr, (size_t)c.mesh_nbrs * 4);
    cp(out->conv_vert_off, L.cvo, ((size_t)c.n_frag + 1) * 4);
    cp(out->conv_pos, L.cpos, (size_t)c.conv_verts * 12);
    cp(out->conv_nbr_off, L.cno, ((size_t)c.conv_verts + 1) * 4);
    cp(out->conv_nbr, L.cnbr, (size_t)c.conv_nbrs * 4);
    cp(out->vnc, L.vnc, (size_t)c.mesh_verts * 36);
    cp(out->idx_off, L.ioff, ((size_t)c.n_frag + 1) * 4);
    cp(out->idx, L.idx, (size_t)c.n_idx * 4);
    cp(out->frag_status, L.fstat, (size_t)c.n_frag * 4);
    return SURTR_OK;
}

int surtr_event_download(surtr_ctx* ctx, surtr_fragments* out)
{
    if (!ctx || !out) return SURTR_E_INVALID;
    surtr_counts c;
    int rc = surtr_event_counts(ctx, &c);
    if (rc) return rc;
    const size_t need = blob_layout(c).total;
    if (ctx->blob_cap < need)
    {
        free_dev(ctx->d_blob); ctx->d_blob = nullptr;
        HIPCHK(hipMalloc(&ctx->d_blob, need));
        ctx->blob_cap = need;
    }
    rc = surtr_event_pack_dev(ctx, ctx->d_blob, ctx->blob_cap);
    if (rc) return rc;
    std::vector<char> host(need);
    HIPCHK(hipMemcpyAsync(host.data(), ctx->d_blob, need, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return surtr_blob_unpack_host(host.data(), need, nullptr, out);
}

int surtr_clip_polyhedron(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* off, const int32_t* nbr,
                          uint32_t n_planes, const float* planes, uint32_t* out_nv, uint32_t* out_nh, float* out_pos,
                          uint32_t* out_off, int32_t* out_nbr)
{
    if (!ctx || !pos || !off || !nbr || !planes || nv < 4 || n_planes > SURTR_MAXF) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc = check_solid(nv, off, nbr);
    if (rc) return rc;
    const uint32_t H = off[nv];
    rc = ensure_scratch(ctx, std::max(nv, ctx->vmax), std::max(H, ctx->hmax), std::max(1u, ctx->n_wg));
    if (rc) return rc;
    const uint32_t capv = ctx->pool.CV, caph = ctx->pool.CH;
    float *d_pos = nullptr, *d_opos = nullptr; uint32_t *d_loff = nullptr, *d_llen = nullptr, *d_ooff = nullptr, *d_res = nullptr, *d_ollen = nullptr;
    int32_t *d_nbr = nullptr, *d_onbr = nullptr; float4* d_pl = nullptr;
    std::vector<uint32_t> llen(nv);
    for (uint32_t v = 0; v < nv; ++v) llen[v] = off[v + 1] - off[v];
    auto cleanup = [&]() { free_dev(d_pos); free_dev(d_opos); free_dev(d_loff); free_dev(d_llen); free_dev(d_ooff); free_dev(d_res);
                           free_dev(d_nbr); free_dev(d_onbr); free_dev(d_pl); free_dev(d_ollen); };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx->err = hipGetErrorString(e_); cleanup(); return SURTR_E_HIP; } } while (0)
    CK(hipMalloc((void**)&d_pos, (size_t)nv * 12)); CK(hipMalloc((void**)&d_loff, (size_t)(nv + 1) * 4));
    CK(hipMalloc((void**)&d_llen, (size_t)nv * 4)); CK(hipMalloc((void**)&d_nbr, std::max<size_t>(16, (size_t)H * 4)));
    CK(hipMalloc((void**)&d_pl, std::max<size_t>(16, (size_t)n_planes * 16)));
    CK(hipMalloc((void**)&d_opos, (size_t)capv * 12)); CK(hipMalloc((void**)&d_ooff, (size_t)(capv + 1) * 4));
    CK(hipMalloc((void**)&d_onbr, (size_t)caph * 4)); CK(hipMalloc((void**)&d_res, 16)); CK(hipMalloc((void**)&d_ollen, (size_t)(capv + 1) * 4));
    CK(hipMemcpy(d_pos, pos, (size_t)nv * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_loff, off, (size_t)(nv + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_llen, llen.data(), (size_t)nv * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nbr, nbr, (size_t)H * 4, hipMemcpyHostToDevice));
    if (n_planes) CK(hipMemcpy(d_pl, planes, (size_t)n_planes * 16, hipMemcpyHostToDevice));
    SolidIn in{d_pos, d_loff, d_llen, d_nbr, nv, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_clip_single, dim3(1), dim3(SURTR_WG), 0, ctx->stream, in, d_pl, n_planes, ctx->pool, d_opos, d_ooff, d_onbr,
                       d_ollen, capv, caph, d_res);
    CK(hipGetLastError());
    uint32_t res[3] = {0, 0, 0};
    CK(hipMemcpyAsync(res, d_res, 12, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (res[2] == 0)
    {
        if (out_nv) *out_nv = res[0];
        if (out_nh) *out_nh = res[1];
        if (out_pos && res[0]) CK(hipMemcpy(out_pos, d_opos, (size_t)res[0] * 12, hipMemcpyDeviceToHost));
        if (out_off) CK(hipMemcpy(out_off, d_ooff, (size_t)(res[0] + 1) * 4, hipMemcpyDeviceToHost));
        if (out_nbr && res[1]) CK(hipMemcpy(out_nbr, d_onbr, (size_t)res[1] * 4, hipMemcpyDeviceToHost));
        if (out_off && res[0] == 0) out_off[0] = 0;
    }
#undef CK
    cleanup();
    return (int)res[2];
}

} // extern "C"
