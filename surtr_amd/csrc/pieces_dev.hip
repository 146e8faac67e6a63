// pieces_dev.hip -- the resident pieces (Compound::PieceVec, Inc/Surtr.h:113-134) and everything that is derived from them
// for the pre-pass, built ON THE DEVICE: neighbour-link validation (Src/Poly.cpp:253-260), the "all incident faces are
// triangles" flags, per-vertex ball radii, the Morton-sorted copy with one bounding sphere per SURTR_SB vertices.
//
//   surtr_upload_pieces      host pieces -> HBM (four copies per set) + the kernels below; no host-side preprocessing
//   surtr_transform_pieces   Poly::Transform (Src/Poly.cpp:580-585) of every resident piece by its world matrix
//                            (ExecuteFractureRoutine, Src/Surtr.cpp:1846-1851), then the derived data again
//   surtr_pieces_from_event  the fragments of the last event become the pieces of the next one without leaving HBM
//                            (recursive refracture, BASELINE configs[4])
// All piece buffers come from a grow-only pool: in steady state (same or smaller pieces) no call allocates or frees.
#include <chrono>
#include <cstring>

#include "surtr_ctx.h"
#include <hipcub/hipcub.hpp>      // (the emulation header brings its own two hipcub algorithms)

namespace {

// ------------------------------------------------------------------ kernels
__device__ __forceinline__ int32_t ring_prev(const int32_t* r, uint32_t len, int32_t who)
{
    uint32_t k = 0;
    while (k < len && r[k] != who) ++k;
    return k == 0 ? r[len - 1] : r[k - 1];
}

// piece of global vertex v: last p with vo[p] <= v
__device__ __forceinline__ uint32_t piece_of(const uint32_t* __restrict__ vo, uint32_t n, uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (vo[mid] <= v) lo = mid; else hi = mid; }
    return lo;
}

// llen + the checks of Poly::ExtractNeighborFromMesh's postcondition (Src/Poly.cpp:253-260): indices in range, no self
// link, every link has its back link; degree >= 3.  err = max SURTR_E_* seen.
__global__ void k_piece_check(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const uint32_t* __restrict__ loff,
                              const int32_t* __restrict__ nbr, uint32_t* __restrict__ llen, uint32_t check, uint32_t* __restrict__ err,
                              uint8_t* __restrict__ dup)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const uint32_t lo = loff[v], hi = loff[v + 1];
    if (hi < lo) { atomicMax(err, (uint32_t)SURTR_E_INVALID); llen[v] = 0; return; }
    const uint32_t deg = hi - lo;
    llen[v] = deg;
    const uint32_t p = piece_of(vo, n, v), a = vo[p], m = vo[p + 1] - a;
    // a ring that lists a neighbour twice marks its piece (whether or not the links are validated: fragments turned into
    // pieces on the device have such rings too)
    for (uint32_t j = lo + 1u; j < hi; ++j)
        for (uint32_t q = lo; q < j; ++q) if (nbr[q] == nbr[j]) { dup[p] = 1; j = hi; break; }
    if (!check) return;
    if (deg < 3u) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); return; }
    const int32_t lv = (int32_t)(v - a);
    for (uint32_t j = lo; j < hi; ++j)
    {
        const int32_t u = nbr[j];
        if (u < 0 || (uint32_t)u >= m || u == lv) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); return; }
        const uint32_t ulo = loff[a + (uint32_t)u], uhi = loff[a + (uint32_t)u + 1];
        bool back = false;
        for (uint32_t q = ulo; q < uhi && q >= ulo; ++q) if (nbr[q] == lv) { back = true; break; }
        if (!back) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); return; }
    }
}

// tri[v] = 1 when every face around v is a triangle (then the 1-ring holds every vertex of v's incident faces);
// rad[v] = radius of a ball around v that holds every vertex of every face incident to v (rounded up).
__global__ void k_piece_tri_rad(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const float* __restrict__ pos,
                                const uint32_t* __restrict__ loff, const int32_t* __restrict__ nbr, uint8_t* __restrict__ tri,
                                float* __restrict__ rad)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const uint32_t p = piece_of(vo, n, v), a = vo[p], m = vo[p + 1] - a;
    const int32_t lv = (int32_t)(v - a);
    auto ring = [&](int32_t x) { return nbr + loff[a + (uint32_t)x]; };
    auto rlen = [&](int32_t x) { return loff[a + (uint32_t)x + 1] - loff[a + (uint32_t)x]; };
    const uint32_t lo = loff[v], hi = loff[v + 1];
    bool t = true;
    for (uint32_t j = lo; j < hi; ++j)
    {
        const int32_t x = nbr[j];
        const int32_t y = ring_prev(ring(x), rlen(x), lv);
        if (y == lv || ring_prev(ring(y), rlen(y), x) != lv) { t = false; break; }
    }
    tri[v] = t ? 1 : 0;
    const double px = pos[3 * (size_t)v], py = pos[3 * (size_t)v + 1], pz = pos[3 * (size_t)v + 2];
    auto dist = [&](int32_t x) {
        const size_t g = a + (uint32_t)x;
        const double dx = px - (double)pos[3 * g], dy = py - (double)pos[3 * g + 1], dz = pz - (double)pos[3 * g + 2];
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    double r = 0.0;
    for (uint32_t j = lo; j < hi; ++j)
    {
        const double d = dist(nbr[j]);
        r = d > r ? d : r;
        if (!t)
        {
            int32_t prev = lv, cur = nbr[j]; uint32_t steps = 0;
            while (cur != lv && steps++ < m)
            {
                const double d2 = dist(cur);
                r = d2 > r ? d2 : r;
                const int32_t nx = ring_prev(ring(cur), rlen(cur), prev);
                prev = cur; cur = nx;
            }
        }
    }
    rad[v] = (float)(r * 1.000001) + 1e-30f;
}

// Axis-aligned box of every piece: one workgroup per piece.  box[6p..] = lo xyz, hi xyz (exact float min / max).
__global__ __launch_bounds__(SURTR_WG) void k_piece_box(uint32_t n, const uint32_t* __restrict__ vo, const float* __restrict__ pos, float* __restrict__ box)
{
    __shared__ float part[SURTR_NWAVE][6];
    const uint32_t p = blockIdx.x;
    if (p >= n) return;
    const uint32_t a = vo[p], b = vo[p + 1];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (uint32_t v = a + threadIdx.x; v < b; v += group_size())
        for (int c = 0; c < 3; ++c) { const float x = pos[3 * (size_t)v + c]; lo[c] = x < lo[c] ? x : lo[c]; hi[c] = x > hi[c] ? x : hi[c]; }
    for (int d = SURTR_LANES / 2; d >= 1; d >>= 1)
        for (int c = 0; c < 3; ++c)
        {
            const float ol = __shfl_down(lo[c], d, SURTR_LANES), oh = __shfl_down(hi[c], d, SURTR_LANES);
            lo[c] = ol < lo[c] ? ol : lo[c]; hi[c] = oh > hi[c] ? oh : hi[c];
        }
    if (lane_id() == 0) for (int c = 0; c < 3; ++c) { part[wave_id()][c] = lo[c]; part[wave_id()][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (uint32_t w = 1; w < group_waves(); ++w)
            for (int c = 0; c < 3; ++c)
            {
                lo[c] = part[w][c] < lo[c] ? part[w][c] : lo[c];
                hi[c] = part[w][3 + c] > hi[c] ? part[w][3 + c] : hi[c];
            }
        for (int c = 0; c < 3; ++c) { box[6 * p + c] = lo[c]; box[6 * p + 3 + c] = hi[c]; }
    }
}

// Sort key of vertex v: piece << 32 | 30-bit Morton code of its position in the piece's box; value = piece-local index.
__global__ void k_piece_keys(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const float* __restrict__ pos,
                             const float* __restrict__ box, unsigned long long* __restrict__ key, uint32_t* __restrict__ val)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const uint32_t p = piece_of(vo, n, v);
    uint32_t code = 0;
    for (int c = 0; c < 3; ++c)
    {
        const double lo = box[6 * p + c], ext = (double)box[6 * p + 3 + c] - lo;
        uint32_t q = 0;
        if (ext > 0)
        {
            double t = ((double)pos[3 * (size_t)v + c] - lo) / ext * 1024.0;
            t = t < 0.0 ? 0.0 : (t > 1023.0 ? 1023.0 : t);
            q = (uint32_t)t;
        }
        for (int bit = 0; bit < 10; ++bit) code |= ((q >> bit) & 1u) << (3 * bit + c);
    }
    key[v] = ((unsigned long long)p << 32) | code;
    val[v] = v - vo[p];
}

// Sorted copy: sorted slot i of piece p holds vertex perm[i] (piece-local), its position and its ball radius.
__global__ void k_piece_sorted(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const float* __restrict__ pos,
                               const float* __restrict__ rad, const uint32_t* __restrict__ perm, float4* __restrict__ posr_s)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const uint32_t p = piece_of(vo, n, i);
    const size_t g = (size_t)vo[p] + perm[i];
    posr_s[i] = make_float4(pos[3 * g], pos[3 * g + 1], pos[3 * g + 2], rad[g]);
}

// One bounding sphere per SURTR_SB consecutive sorted vertices of a piece: it holds their balls (pre-pass A0).
__global__ void k_piece_spheres(uint32_t NB, uint32_t n, const uint32_t* __restrict__ vo, const uint32_t* __restrict__ bo,
                                const float4* __restrict__ posr_s, float4* __restrict__ bsph)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= NB) return;
    const uint32_t p = piece_of(bo, n, g), a = vo[p], m = vo[p + 1] - a;
    const uint32_t i0 = (g - bo[p]) * SURTR_SB, i1 = i0 + SURTR_SB < m ? i0 + SURTR_SB : m;
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = i0; i < i1; ++i)
        for (int c = 0; c < 3; ++c)
        {
            const float4 pr = posr_s[a + i];
            const double x = c == 0 ? pr.x : (c == 1 ? pr.y : pr.z);
            blo[c] = x < blo[c] ? x : blo[c]; bhi[c] = x > bhi[c] ? x : bhi[c];
        }
    const float cx = (float)((blo[0] + bhi[0]) / 2), cy = (float)((blo[1] + bhi[1]) / 2), cz = (float)((blo[2] + bhi[2]) / 2);
    double R = 0;
    for (uint32_t i = i0; i < i1; ++i)
    {
        const float4 pr = posr_s[a + i];
        const double dx = pr.x - (double)cx, dy = pr.y - (double)cy, dz = pr.z - (double)cz;
        const double d = sqrt(dx * dx + dy * dy + dz * dz) + (double)pr.w;
        R = d > R ? d : R;
    }
    bsph[g] = make_float4(cx, cy, cz, (float)(R * 1.000001) + 1e-30f);
}

// A coarser level of spheres: sphere g of piece p holds SURTR_SPH_FAN consecutive spheres of the level below (bo / boc: first
// sphere of every piece at the lower / this level).
#ifndef SURTR_SPH_FAN
#define SURTR_SPH_FAN 8u
#endif
__global__ void k_piece_spheres_up(uint32_t NBc, uint32_t n, const uint32_t* __restrict__ bo, const uint32_t* __restrict__ boc,
                                   const float4* __restrict__ lower, float4* __restrict__ upper)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= NBc) return;
    const uint32_t p = piece_of(boc, n, g), a = bo[p], m = bo[p + 1] - a;
    const uint32_t i0 = (g - boc[p]) * SURTR_SPH_FAN, i1 = i0 + SURTR_SPH_FAN < m ? i0 + SURTR_SPH_FAN : m;
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = i0; i < i1; ++i)
    {
        const float4 s = lower[a + i];
        const double c[3] = {s.x, s.y, s.z};
        for (int q = 0; q < 3; ++q) { blo[q] = c[q] - s.w < blo[q] ? c[q] - s.w : blo[q]; bhi[q] = c[q] + s.w > bhi[q] ? c[q] + s.w : bhi[q]; }
    }
    const float cx = (float)((blo[0] + bhi[0]) / 2), cy = (float)((blo[1] + bhi[1]) / 2), cz = (float)((blo[2] + bhi[2]) / 2);
    double R = 0;
    for (uint32_t i = i0; i < i1; ++i)
    {
        const float4 s = lower[a + i];
        const double dx = s.x - (double)cx, dy = s.y - (double)cy, dz = s.z - (double)cz;
        const double d = sqrt(dx * dx + dy * dy + dz * dz) + (double)s.w;
        R = d > R ? d : R;
    }
    upper[g] = make_float4(cx, cy, cz, (float)(R * 1.000001) + 1e-30f);
}

// Rings in sorted space.  iperm: piece-local sorted index of every vertex; row_s: see Pieces.
__global__ void k_piece_iperm(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const uint32_t* __restrict__ perm, uint32_t* __restrict__ iperm)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const uint32_t p = piece_of(vo, n, i), a = vo[p];
    iperm[a + perm[i]] = i - a;
}
__global__ void k_piece_row_s(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const uint32_t* __restrict__ perm,
                              const uint32_t* __restrict__ loff, const uint32_t* __restrict__ llen, const int32_t* __restrict__ nbr,
                              const uint8_t* __restrict__ tri, const uint32_t* __restrict__ iperm, SRow* __restrict__ row_s)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const uint32_t p = piece_of(vo, n, i), a = vo[p], m = vo[p + 1] - a;
    const uint32_t g = a + perm[i];
    const uint32_t lo = loff[g], len = llen[g];
    uint32_t h[8];
    h[0] = (len <= 7u ? len : 0x40u) | (tri[g] ? 0u : 0x80u);
    for (uint32_t j = 0; j < 7u; ++j)
    {
        uint32_t e = 0xFFFFu;
        if (len <= 7u && j < len)
        {
            const int32_t u = nbr[lo + j];
            if (u >= 0 && (uint32_t)u < m) { const uint32_t x = iperm[a + (uint32_t)u]; e = x < 0xFFFFu ? x : 0xFFFFu; }      // (an invalid link: the upload refuses the piece)
        }
        h[1u + j] = e;
    }
    SRow r;
    for (int q = 0; q < 4; ++q) r.w[q] = h[2 * q] | (h[2 * q + 1] << 16);
    row_s[i] = r;
}

// Poly::Transform (Src/Poly.cpp:580-585): Position = XMVector3TransformCoord(Position, XMMatrixTranspose(matrix)).
// XMVector3TransformCoord (DirectXMath, not in the reference tree): r = z*M.r[2] + M.r[3]; r = y*M.r[1] + r;
// r = x*M.r[0] + r; result = r.xyz / r.w -- with M = the transpose, M.r[k][c] = world[4*c + k].
__global__ void k_transform(uint32_t V, uint32_t n, const uint32_t* __restrict__ vo, const float* __restrict__ world, float* __restrict__ pos)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float* W = world + 16 * (size_t)piece_of(vo, n, v);
    const float x = pos[3 * (size_t)v], y = pos[3 * (size_t)v + 1], z = pos[3 * (size_t)v + 2];
    float r[4];
    for (int c = 0; c < 4; ++c)
    {
        float t = z * W[4 * c + 2] + W[4 * c + 3];
        t = y * W[4 * c + 1] + t;
        r[c] = x * W[4 * c] + t;
    }
    pos[3 * (size_t)v] = r[0] / r[3]; pos[3 * (size_t)v + 1] = r[1] / r[3]; pos[3 * (size_t)v + 2] = r[2] / r[3];
}

// Fragments of the last event -> piece buffers.  One workgroup per kept fragment and set (blockIdx = 2*piece + set).
struct FromEvent { const uint32_t* frag; const uint32_t* vo[2]; const uint32_t* ho[2]; };
__global__ __launch_bounds__(SURTR_WG) void k_pieces_from_frags(uint32_t n, FromEvent E, const FragRec* __restrict__ frags, Arena A,
                                                                float* mpos, uint32_t* mloff, int32_t* mnbr, float* cpos, uint32_t* cloff, int32_t* cnbr)
{
    const uint32_t p = blockIdx.x >> 1, set = blockIdx.x & 1u;
    if (p >= n) return;
    const FragRec fr = frags[E.frag[p]];
    const uint32_t sv = set ? fr.cv_off : fr.mv_off, nv = set ? fr.cv_n : fr.mv_n, sh = set ? fr.ch_off : fr.mh_off, nh = set ? fr.ch_n : fr.mh_n;
    const uint32_t dv = E.vo[set][p], dh = E.ho[set][p];
    float* pos = set ? cpos : mpos; uint32_t* loff = set ? cloff : mloff; int32_t* nbr = set ? cnbr : mnbr;
    for (uint32_t i = threadIdx.x; i < 3u * nv; i += group_size()) pos[3 * (size_t)dv + i] = A.pos[3 * (size_t)sv + i];
    for (uint32_t v = threadIdx.x; v < nv; v += group_size()) loff[dv + v] = dh + (A.loff[sv + v] - sh);
    for (uint32_t e = threadIdx.x; e < nh; e += group_size()) nbr[dh + e] = A.nbr[sh + e];
    if (p + 1u == n && threadIdx.x == 0) loff[dv + nv] = dh + nh;
}

// ------------------------------------------------------------------- host
template <class T>
int pool_reserve(surtr_ctx* ctx, T** p, size_t& cap, size_t need)
{
    if (*p && cap >= need) return SURTR_OK;
    free_dev(*p); *p = nullptr; cap = 0;
    const size_t want = std::max<size_t>(need + need / 4, 64);      // a little room, so that slightly larger pieces fit too
    if (hipMalloc((void**)p, want * sizeof(T)) != hipSuccess) { ctx->err = "piece pool allocation failed"; return SURTR_E_HIP; }
    cap = want; ++ctx->upload_allocs;
    return SURTR_OK;
}

static inline uint32_t up_count(uint32_t m) { return (m + SURTR_SPH_FAN - 1u) / SURTR_SPH_FAN; }

int reserve_set(surtr_ctx* ctx, PieceSet& S, uint32_t n, uint32_t V, uint32_t H, uint32_t NB)
{
    int rc = 0;
#define R(ptr, cap, need) do { rc = pool_reserve(ctx, &S.ptr, S.cap, (size_t)(need)); if (rc) return rc; } while (0)
    R(pos, c_pos, 3 * (size_t)V + 3); R(loff, c_loff, (size_t)V + 1); R(llen, c_llen, V); R(nbr, c_nbr, (size_t)H + 1); R(vo, c_vo, n + 1);
    R(tri, c_tri, V); R(rad, c_rad, V); R(perm, c_perm, V); R(posr_s, c_posr_s, (size_t)V + 1);
    R(bsph, c_bsph, NB + 1); R(bo, c_bo, n + 1); R(box, c_box, 6 * (size_t)n); R(key, c_key, V); R(key2, c_key2, V); R(val, c_val, V);
    R(dup, c_dup, n + 1);
    // (every piece has at least one sphere per level: NB / 8 + n and NB / 64 + n bound the coarser levels)
    R(iperm, c_iperm, V); R(row_s, c_row_s, (size_t)V + 1);
    R(bsph2, c_bsph2, (size_t)NB / SURTR_SPH_FAN + n + 1); R(bo2, c_bo2, n + 1);
    R(bsph3, c_bsph3, (size_t)NB / (SURTR_SPH_FAN * SURTR_SPH_FAN) + n + 1); R(bo3, c_bo3, n + 1);
#undef R
    return SURTR_OK;
}

// Derived data of one set whose pos / loff / nbr / vo are in place.  `check`: validate the links (host uploads).
int derive_set(surtr_ctx* ctx, PieceSet& S, uint32_t n, uint32_t V, const std::vector<uint32_t>& bo_h, bool check)
{
    hipStream_t st = ctx->stream;
    const uint32_t NB = bo_h[n];
    HIPCHK(hipMemcpyAsync(S.bo, bo_h.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    const dim3 blk(256), gridV((V + 255) / 256);
    HIPCHK(hipMemsetAsync(S.dup, 0, (size_t)n + 1, st));
    hipLaunchKernelGGL(k_piece_check, gridV, blk, 0, st, V, n, S.vo, S.loff, S.nbr, S.llen, check ? 1u : 0u, ctx->d_upload_err, S.dup);
    hipLaunchKernelGGL(k_piece_tri_rad, gridV, blk, 0, st, V, n, S.vo, S.pos, S.loff, S.nbr, S.tri, S.rad);
    hipLaunchKernelGGL(k_piece_box, dim3(n), dim3(SURTR_WG), 0, st, n, S.vo, S.pos, S.box);
    hipLaunchKernelGGL(k_piece_keys, gridV, blk, 0, st, V, n, S.vo, S.pos, S.box, S.key, S.val);
    // stable sort of (piece, Morton code) -> piece-local vertex: per piece the host order std::sort gave pairs (code, vertex)
    int end_bit = 32;
    while (end_bit < 64 && (n >> (end_bit - 32)) != 0u) ++end_bit;
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, S.key, S.key2, S.val, S.perm, (int)V, 0, end_bit, st) != hipSuccess) return SURTR_E_HIP;
    int rc = pool_reserve(ctx, &ctx->sort_tmp, ctx->c_sort_tmp, tmp_bytes + 16);
    if (rc) return rc;
    if (hipcub::DeviceRadixSort::SortPairs(ctx->sort_tmp, tmp_bytes, S.key, S.key2, S.val, S.perm, (int)V, 0, end_bit, st) != hipSuccess) return SURTR_E_HIP;
    hipLaunchKernelGGL(k_piece_sorted, gridV, blk, 0, st, V, n, S.vo, S.pos, S.rad, S.perm, S.posr_s);
    if (NB) hipLaunchKernelGGL(k_piece_spheres, dim3((NB + 255) / 256), blk, 0, st, NB, n, S.vo, S.bo, S.posr_s, S.bsph);
    // two coarser sphere levels for the hierarchical cull of the pre-pass
    std::vector<uint32_t> bo2(n + 1, 0u), bo3(n + 1, 0u);
    for (uint32_t p = 0; p < n; ++p) { bo2[p + 1] = bo2[p] + up_count(bo_h[p + 1] - bo_h[p]); bo3[p + 1] = bo3[p] + up_count(bo2[p + 1] - bo2[p]); }
    HIPCHK(hipMemcpyAsync(S.bo2, bo2.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(S.bo3, bo3.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    // (pageable source: staged before the call returns, like S.bo above)
    if (bo2[n]) hipLaunchKernelGGL(k_piece_spheres_up, dim3((bo2[n] + 255) / 256), blk, 0, st, bo2[n], n, S.bo, S.bo2, S.bsph, S.bsph2);
    if (bo3[n]) hipLaunchKernelGGL(k_piece_spheres_up, dim3((bo3[n] + 255) / 256), blk, 0, st, bo3[n], n, S.bo2, S.bo3, S.bsph2, S.bsph3);
    // the rings in sorted space
    hipLaunchKernelGGL(k_piece_iperm, gridV, blk, 0, st, V, n, S.vo, S.perm, S.iperm);
    hipLaunchKernelGGL(k_piece_row_s, gridV, blk, 0, st, V, n, S.vo, S.perm, S.loff, S.llen, S.nbr, S.tri, S.iperm, S.row_s);
    HIPCHK(hipGetLastError());
    return SURTR_OK;
}

std::vector<uint32_t> sphere_offsets(uint32_t n, const uint32_t* vo)
{
    std::vector<uint32_t> bo(n + 1, 0u);
    for (uint32_t p = 0; p < n; ++p) bo[p + 1] = bo[p] + (vo[p + 1] - vo[p] + SURTR_SB - 1u) / SURTR_SB;
    return bo;
}

// What the event sizes its scratch from, and whether the half-size clip kernel is worth launching.
void set_piece_stats(surtr_ctx* ctx, uint32_t n, const uint32_t* mvo, const uint32_t* mho, const uint32_t* cvo, const uint32_t* cho)
{
    uint32_t vmax = 0, hmax = 0, cvmax = 0, chmax = 0, small = 0, vmin = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i)
    {
        vmax = std::max(vmax, mvo[i + 1] - mvo[i]); vmin = std::min(vmin, mvo[i + 1] - mvo[i]); hmax = std::max(hmax, mho[i + 1] - mho[i]);
        cvmax = std::max(cvmax, cvo[i + 1] - cvo[i]); chmax = std::max(chmax, cho[i + 1] - cho[i]);
        if (surtr_fits_half((mvo[i + 1] - mvo[i]) / 4u, (mho[i + 1] - mho[i]) / 4u)) ++small;      // what a cell keeps of it is likely light
    }
    ctx->vmin = n ? vmin : 0u;
    ctx->n_pieces = n; ctx->vmax = std::max(vmax, cvmax); ctx->hmax = std::max(hmax, chmax); ctx->cvmax = cvmax; ctx->chmax = chmax;
    ctx->tot_mv = mvo[n]; ctx->tot_mh = mho[n]; ctx->pair_order_count = 0;
    ctx->half_on = 4ull * small >= 3ull * n;
    if (const char* e = getenv("SURTR_HALF")) ctx->half_on = atoi(e) != 0;      // tests: force either way
    ctx->have_event = false; ctx->frags_of_pieces = false;
}

int finish_upload(surtr_ctx* ctx, uint32_t n, bool check)
{
    if (!ctx->d_outside || ctx->cap_outside < n)
    {
        free_dev(ctx->d_outside); ctx->d_outside = nullptr;
        HIPCHK(hipMalloc((void**)&ctx->d_outside, std::max<uint32_t>(n + n / 4, 64)));
        ctx->cap_outside = std::max<uint32_t>(n + n / 4, 64); ++ctx->upload_allocs;
    }
    uint32_t err = 0;
    HIPCHK(hipMemcpyAsync(&err, ctx->d_upload_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (check && err) { ctx->n_pieces = 0; return (int)err; }
    return SURTR_OK;
}

struct Timer
{
    surtr_ctx* ctx; std::chrono::steady_clock::time_point t0;
    explicit Timer(surtr_ctx* c) : ctx(c), t0(std::chrono::steady_clock::now()) { c->upload_allocs = 0; }
    ~Timer() { ctx->upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

} // namespace

extern "C" {

int surtr_upload_pieces(surtr_ctx* ctx, uint32_t n, const uint32_t* mvo, const float* mpos, const uint32_t* moff,
                        const int32_t* mnbr, const uint32_t* cvo, const float* cpos, const uint32_t* coff, const int32_t* cnbr)
{
    if (!ctx || n == 0 || !mvo || !mpos || !moff || !mnbr || !cvo || !cpos || !coff || !cnbr) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    Timer timer(ctx);
    ctx->n_pieces = 0;
    std::vector<uint32_t> mho(n + 1), cho(n + 1);
    for (int set = 0; set < 2; ++set)
    {
        const uint32_t* vo = set ? cvo : mvo; const uint32_t* off = set ? coff : moff;
        if (vo[0] != 0) return SURTR_E_INVALID;
        for (uint32_t p = 0; p < n; ++p)
        {
            const uint32_t a = vo[p], b = vo[p + 1];
            if (b < a || b - a < 4 || b - a >= (1u << 24)) return SURTR_E_INVALID;     // the pre-pass packs (vertex, plane) in 32 bits
            if (off[b] < off[a]) return SURTR_E_INVALID;
        }
        for (uint32_t p = 0; p <= n; ++p) (set ? cho : mho)[p] = off[vo[p]];
    }
    hipStream_t st = ctx->stream;
    HIPCHK(hipStreamSynchronize(st));       // an event may still be reading the pieces
    if (!ctx->d_upload_err) { HIPCHK(hipMalloc((void**)&ctx->d_upload_err, 16)); ++ctx->upload_allocs; }
    HIPCHK(hipMemsetAsync(ctx->d_upload_err, 0, 4, st));
    for (int set = 0; set < 2; ++set)
    {
        PieceSet& S = set ? ctx->cset : ctx->mset;
        const uint32_t* vo = set ? cvo : mvo; const float* pos = set ? cpos : mpos; const uint32_t* off = set ? coff : moff; const int32_t* nbr = set ? cnbr : mnbr;
        const uint32_t V = vo[n], H = off[V];
        const std::vector<uint32_t> bo = sphere_offsets(n, vo);
        int rc = reserve_set(ctx, S, n, V, H, bo[n]);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(S.pos, pos, (size_t)V * 12, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(S.loff, off, (size_t)(V + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(S.nbr, nbr, (size_t)H * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(S.vo, vo, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
        rc = derive_set(ctx, S, n, V, bo, true);
        if (rc) return rc;
    }
    set_piece_stats(ctx, n, mvo, mho.data(), cvo, cho.data());
    return finish_upload(ctx, n, true);
}

int surtr_transform_pieces(surtr_ctx* ctx, uint32_t n, const float* world)
{
    if (!ctx || !world) return SURTR_E_INVALID;
    if (!ctx->n_pieces || n != ctx->n_pieces) return ctx && ctx->n_pieces ? SURTR_E_INVALID : SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    Timer timer(ctx);
    hipStream_t st = ctx->stream;
    int rc = pool_reserve(ctx, &ctx->d_world, ctx->c_world, (size_t)16 * n);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpyAsync(ctx->d_world, world, (size_t)64 * n, hipMemcpyHostToDevice, st));
    std::vector<uint32_t> vo(n + 1);
    for (int set = 0; set < 2; ++set)
    {
        PieceSet& S = set ? ctx->cset : ctx->mset;
        HIPCHK(hipMemcpyAsync(vo.data(), S.vo, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const uint32_t V = vo[n];
        hipLaunchKernelGGL(k_transform, dim3((V + 255) / 256), dim3(256), 0, st, V, n, S.vo, ctx->d_world, S.pos);
        rc = derive_set(ctx, S, n, V, sphere_offsets(n, vo.data()), false);
        if (rc) return rc;
    }
    ctx->have_event = false; ctx->frags_of_pieces = false;
    HIPCHK(hipStreamSynchronize(st));
    return SURTR_OK;
}

int surtr_pieces_from_event(surtr_ctx* ctx, const uint8_t* keep, uint32_t* n_out)
{
    if (!ctx) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    Timer timer(ctx);
    hipStream_t st = ctx->stream;
    surtr_counts c;
    int rc = surtr_event_counts(ctx, &c);
    if (rc) return rc;
    // the fragment table is small (one record per fragment): sizes and offsets of the new pieces are laid out on the host,
    // the solids themselves never leave HBM
    std::vector<FragRec> fr(c.n_frag);
    if (c.n_frag) HIPCHK(hipMemcpy(fr.data(), ctx->d_frags, (size_t)c.n_frag * sizeof(FragRec), hipMemcpyDeviceToHost));
    std::vector<uint32_t> frag, vo[2], ho[2];
    vo[0].push_back(0); vo[1].push_back(0); ho[0].push_back(0); ho[1].push_back(0);
    for (uint32_t k = 0; k < c.n_frag; ++k)
    {
        if (keep && !keep[k]) continue;
        // fewer than four vertices is no solid (Src/Poly.cpp:497-499; a Convex the refit clipped away): with an explicit
        // mask that is the caller's error, without one such fragments are left out (the reference's m_fractureTask would get
        // nothing out of them either, Src/Surtr.cpp:1466-1468)
        if (fr[k].mv_n < 4 || fr[k].cv_n < 4) { if (keep) return SURTR_E_INVALID; continue; }
        frag.push_back(k);
        vo[0].push_back(vo[0].back() + fr[k].mv_n); ho[0].push_back(ho[0].back() + fr[k].mh_n);
        vo[1].push_back(vo[1].back() + fr[k].cv_n); ho[1].push_back(ho[1].back() + fr[k].ch_n);
    }
    const uint32_t n = (uint32_t)frag.size();
    if (n_out) *n_out = n;
    if (n == 0) return SURTR_E_INVALID;
    if (!ctx->d_upload_err) { HIPCHK(hipMalloc((void**)&ctx->d_upload_err, 16)); ++ctx->upload_allocs; }
    HIPCHK(hipMemsetAsync(ctx->d_upload_err, 0, 4, st));
    std::vector<uint32_t> bo[2] = {sphere_offsets(n, vo[0].data()), sphere_offsets(n, vo[1].data())};
    for (int set = 0; set < 2; ++set)
    {
        rc = reserve_set(ctx, set ? ctx->cset : ctx->mset, n, vo[set][n], ho[set][n], bo[set][n]);
        if (rc) return rc;
    }
    rc = pool_reserve(ctx, &ctx->d_from, ctx->c_from, (size_t)5 * (n + 1));
    if (rc) return rc;
    uint32_t* d = ctx->d_from;
    FromEvent E{d, {d + (n + 1), d + 2 * (size_t)(n + 1)}, {d + 3 * (size_t)(n + 1), d + 4 * (size_t)(n + 1)}};
    HIPCHK(hipMemcpyAsync(d, frag.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    for (int set = 0; set < 2; ++set)
    {
        HIPCHK(hipMemcpyAsync((void*)E.vo[set], vo[set].data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync((void*)E.ho[set], ho[set].data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync((set ? ctx->cset : ctx->mset).vo, vo[set].data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_pieces_from_frags, dim3(2 * n), dim3(SURTR_WG), 0, st, n, E, ctx->d_frags, ctx->arena,
                       ctx->mset.pos, ctx->mset.loff, ctx->mset.nbr, ctx->cset.pos, ctx->cset.loff, ctx->cset.nbr);
    for (int set = 0; set < 2; ++set)
    {
        rc = derive_set(ctx, set ? ctx->cset : ctx->mset, n, vo[set][n], bo[set], false);
        if (rc) return rc;
    }
    set_piece_stats(ctx, n, vo[0].data(), ho[0].data(), vo[1].data(), ho[1].data());
    ctx->have_event = true;       // the event's fragments are still in the arena: they can be downloaded after this call
    ctx->frags_of_pieces = false; // (but the pieces they came from are gone)
    return finish_upload(ctx, n, false);
}

// scale / shift of every group = extent / centre of the Mesh box of the piece with the same number (k_piece_box left it in S.box)
__global__ void k_group_boxes(uint32_t n, const float* __restrict__ box, float* __restrict__ scale3, float* __restrict__ shift3)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    for (int c = 0; c < 3; ++c)
    {
        const float lo = box[6 * g + c], hi = box[6 * g + 3 + c];
        scale3[3 * g + c] = hi - lo;                                           // Vector3(maxX - minX, ...) (Src/Surtr.cpp:1800)
        shift3[3 * g + c] = (float)(((double)hi + (double)lo) / 2.0);          // BBCenter: double arithmetic, narrowed (:1771)
    }
}

int surtr_place_cells_in_pieces(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off)
{
    if (!ctx || !n_groups || !group_cell_off) return SURTR_E_INVALID;
    if (!ctx->n_pieces || !ctx->d_v012) return SURTR_E_STATE;
    if (n_groups != ctx->n_pieces) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    int rc = pool_reserve(ctx, &ctx->d_group_xf, ctx->c_group_xf, (size_t)6 * n_groups);
    if (rc) return rc;
    float* sc = ctx->d_group_xf; float* sh = sc + 3 * (size_t)n_groups;
    hipLaunchKernelGGL(k_group_boxes, dim3((n_groups + 255) / 256), dim3(256), 0, ctx->stream, n_groups, ctx->mset.box, sc, sh);
    HIPCHK(hipGetLastError());
    return surtr_place_cells_groups_dev(ctx, n_groups, group_cell_off, sc, sh);
}

int surtr_download_piece(surtr_ctx* ctx, uint32_t piece, int set, uint32_t* out_nv, uint32_t* out_nh, float* out_pos, uint32_t* out_off, int32_t* out_nbr)
{
    if (!ctx || set < 0 || set > 1) return SURTR_E_INVALID;
    if (!ctx->n_pieces) return SURTR_E_STATE;
    if (piece >= ctx->n_pieces) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    const PieceSet& S = set ? ctx->cset : ctx->mset;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint32_t vo[2], ho[2];
    HIPCHK(hipMemcpy(vo, S.vo + piece, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&ho[0], S.loff + vo[0], 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&ho[1], S.loff + vo[1], 4, hipMemcpyDeviceToHost));
    const uint32_t nv = vo[1] - vo[0], nh = ho[1] - ho[0];
    if (out_nv) *out_nv = nv;
    if (out_nh) *out_nh = nh;
    if (out_pos) HIPCHK(hipMemcpy(out_pos, S.pos + 3 * (size_t)vo[0], (size_t)nv * 12, hipMemcpyDeviceToHost));
    if (out_off)
    {
        HIPCHK(hipMemcpy(out_off, S.loff + vo[0], ((size_t)nv + 1) * 4, hipMemcpyDeviceToHost));
        for (uint32_t v = 0; v <= nv; ++v) out_off[v] -= ho[0];
    }
    if (out_nbr && nh) HIPCHK(hipMemcpy(out_nbr, S.nbr + ho[0], (size_t)nh * 4, hipMemcpyDeviceToHost));
    return SURTR_OK;
}

int surtr_upload_stats(surtr_ctx* ctx, float* ms, uint32_t* n_alloc)
{
    if (!ctx) return SURTR_E_INVALID;
    if (ms) *ms = ctx->upload_ms;
    if (n_alloc) *n_alloc = ctx->upload_allocs;
    return SURTR_OK;
}

} // extern "C"
