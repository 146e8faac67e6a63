// cells_dev.hip -- Voronoi cell construction on the device (replaces the voro++ call of
// Surtr::GenerateVoronoi, Src/Surtr.cpp:2003-2070, and the host builder surtr_voronoi_cells).
//
// One wave per cell: the unit box clipped in double by the bisector half-spaces of the other seeds of its group, in
// ascending seed order -- the canonical cell of DESIGN.md section 5 and, operation for operation, the arithmetic of
// build_cell() in host_geom.cpp, so that faces, their order and every coordinate are bit-equal to the host builder's.
//   - the 64 lanes test 64 seeds at a time against the current cell (a conservative distance filter first: a seed farther
//     than twice the cell's radius cannot touch it; then the exact "some vertex is on its side" test of the host code);
//   - the lowest seed that touches the cell cuts it (cell = at most SURTR_CELL_V vertices of degree 3 in LDS), the lanes
//     above it test again against the smaller cell: exactly the sequential order;
//   - faces: loops over unseen directed edges, generator = best-fitting plane, outward winding, start at the
//     lexicographically smallest vertex, stable order by generator id (seeds ascending, then the walls -x +x -y +y -z +z).
// The cells go straight into the context as the fracture pattern (v012 + face offsets = surtr_upload_pattern).
#include <chrono>
#include <cstdio>
#include <cstring>

#include "surtr_ctx.h"

#define SURTR_CELL_V 192u      // vertices of a cell while it is being cut
#define SURTR_CELL_G 448u      // generators that cut a cell (6 walls + seeds)
#define SURTR_CELL_F 64u       // faces of a finished cell
#define SURTR_CELL_FV 384u     // face vertices of a finished cell (3 per vertex)

namespace {

struct D3 { double x, y, z; };

struct CellOut      // per cell, fixed stride (compacted by k_pack_cells)
{
    uint32_t nf, nfv, err, pad;
    int32_t gen[SURTR_CELL_F];
    uint16_t fvo[SURTR_CELL_F + 1];
    double v[3 * SURTR_CELL_FV];
};

struct CellLds
{
    D3 p[SURTR_CELL_V]; D3 q[SURTR_CELL_V];
    int16_t ring[SURTR_CELL_V][3]; int16_t ring2[SURTR_CELL_V][3];
    double s[SURTR_CELL_V];
    uint8_t out[SURTR_CELL_V];
    int16_t succ[SURTR_CELL_V], pred[SURTR_CELL_V], id[SURTR_CELL_V];
    int32_t gid[SURTR_CELL_G]; D3 gn[SURTR_CELL_G]; double gc[SURTR_CELL_G];
    uint8_t seen[SURTR_CELL_V][3];
    int16_t loop[SURTR_CELL_FV]; uint16_t flo[SURTR_CELL_F + 1]; int32_t fgen[SURTR_CELL_F]; uint8_t forder[SURTR_CELL_F];
    uint32_t nv, ng, err, nfaces, flag[2], cut_lane;
    double r2; D3 cut_n; double cut_c;
};

__device__ __forceinline__ int ring_prev3(const int16_t* r, int who)
{
    int k = 0;
    while (k < 3 && r[k] != who) ++k;
    return k == 0 ? r[2] : r[k - 1];
}

__device__ __forceinline__ double plane_side(const D3 n, double cc, const D3 p) { return n.x * p.x + n.y * p.y + n.z * p.z - cc; }

// cut_cell() of host_geom.cpp over the lanes of the wave: keep n.x <= cc.  Same results as the sequential code: new
// vertices are numbered in (cut vertex, ring slot) order by a prefix sum, every other step touches disjoint entries.
// Every lane must call it (barriers inside).
__device__ void cut_cell_wave(CellLds& L, const D3 n, const double cc)
{
    const int n0 = (int)L.nv;
    const uint32_t lane = lane_id();
    if (lane == 0) { L.flag[0] = 0; L.flag[1] = 0; }
    __syncthreads();
    bool any_out = false, any_in = false;
    for (int i = (int)lane; i < n0; i += SURTR_LANES)
    {
        const double si = plane_side(n, cc, L.p[i]);
        L.s[i] = si; L.out[i] = si > 0 ? 1 : 0;
        if (si > 0) any_out = true; else any_in = true;
    }
    if (any_out) L.flag[0] = 1;
    if (any_in) L.flag[1] = 1;
    __syncthreads();
    if (!L.flag[0]) return;
    if (!L.flag[1]) { __syncthreads(); if (lane == 0) L.nv = 0; __syncthreads(); return; }
    // crossing edges per cut vertex, numbered in (vertex, slot) order
    uint32_t carry = 0;
    for (int i0 = 0; i0 < n0; i0 += SURTR_LANES)
    {
        const int i = i0 + (int)lane;
        uint32_t cnt = 0;
        if (i < n0 && L.out[i])
            for (int j = 0; j < 3; ++j) { const int k = L.ring[i][j]; if (k < n0 && !L.out[k]) ++cnt; }
        const uint2 inc = wave_incl_scan2(make_uint2(cnt, 0u));
        if (i < n0) L.id[i] = (int16_t)(carry + inc.x - cnt);          // first new vertex of cut vertex i (relative)
        carry += lane_bcast(inc.x, SURTR_LANES - 1u);
    }
    const int n1 = n0 + (int)carry;
    if (n1 > (int)SURTR_CELL_V) { if (lane == 0) L.err = SURTR_E_CAPACITY; __syncthreads(); return; }
    __syncthreads();
    for (int i = (int)lane; i < n0; i += SURTR_LANES)
    {
        if (!L.out[i]) continue;
        int fresh = n0 + L.id[i];
        for (int j = 0; j < 3; ++j)
        {
            const int k = L.ring[i][j];
            if (k >= n0 || L.out[k]) continue;
            const double t = L.s[k] / (L.s[k] - L.s[i]);            // from the kept end towards the cut end
            const D3 a = L.p[k], b = L.p[i];
            L.p[fresh] = D3{a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z)};
            L.ring[fresh][0] = (int16_t)i; L.ring[fresh][1] = (int16_t)k; L.ring[fresh][2] = -1;
            for (int e = 0; e < 3; ++e) if (L.ring[k][e] == i) { L.ring[k][e] = (int16_t)fresh; break; }
            L.ring[i][j] = (int16_t)fresh;
            ++fresh;
        }
    }
    for (int x = (int)lane; x < n1; x += SURTR_LANES) { L.succ[x] = -1; L.pred[x] = -1; }
    __syncthreads();
    // cap: successor of every new vertex along the face through its cut end
    for (int x = n0 + (int)lane; x < n1; x += SURTR_LANES)
    {
        int prev = x, cur = L.ring[x][0], guard = 0;
        while (cur < n0 && L.out[cur] && guard++ < n1)
        {
            const int nx = ring_prev3(L.ring[cur], prev);
            prev = cur; cur = nx;
        }
        L.succ[x] = (int16_t)cur;
        if (cur >= n0) L.pred[cur] = (int16_t)x;
    }
    __syncthreads();
    for (int x = n0 + (int)lane; x < n1; x += SURTR_LANES)
    {
        const int16_t kept = L.ring[x][1];
        L.ring[x][0] = L.pred[x]; L.ring[x][1] = L.succ[x]; L.ring[x][2] = kept;
    }
    __syncthreads();
    // compaction (order preserving)
    carry = 0;
    for (int i0 = 0; i0 < n1; i0 += SURTR_LANES)
    {
        const int i = i0 + (int)lane;
        const uint32_t keep = (i < n1 && (i >= n0 || !L.out[i])) ? 1u : 0u;
        const uint2 inc = wave_incl_scan2(make_uint2(keep, 0u));
        if (i < n1) L.id[i] = keep ? (int16_t)(carry + inc.x - 1u) : (int16_t)-1;
        carry += lane_bcast(inc.x, SURTR_LANES - 1u);
    }
    const int live = (int)carry;
    __syncthreads();
    for (int i = (int)lane; i < n1; i += SURTR_LANES)
    {
        if (L.id[i] < 0) continue;
        L.q[L.id[i]] = L.p[i];
        for (int e = 0; e < 3; ++e) { const int r = L.ring[i][e]; L.ring2[L.id[i]][e] = r >= 0 ? L.id[r] : (int16_t)-1; }
    }
    __syncthreads();
    for (int i = (int)lane; i < live; i += SURTR_LANES) { L.p[i] = L.q[i]; for (int e = 0; e < 3; ++e) L.ring[i][e] = L.ring2[i][e]; }
    if (lane == 0) L.nv = (uint32_t)live;
    __syncthreads();
}

// Faces of the finished cell (the second half of build_cell()): loops on one lane, generators over the lanes, output on one lane.
__device__ void cell_loops_serial(CellLds& L)
{
    const int nvert = (int)L.nv;
    for (int i = 0; i < nvert; ++i) for (int j = 0; j < 3; ++j) L.seen[i][j] = 0;
    int nf = 0, lo = 0;
    for (int i = 0; i < nvert && L.err == 0; ++i)
        for (int j = 0; j < 3; ++j)
        {
            if (L.seen[i][j]) continue;
            if (nf >= (int)SURTR_CELL_F || lo >= (int)SURTR_CELL_FV) { L.err = SURTR_E_CAPACITY; break; }
            const int start = lo;
            int prev = i, cur = L.ring[i][j], len = 1;
            L.seen[i][j] = 1;
            L.loop[lo++] = (int16_t)i;
            while (cur != i && len <= nvert)
            {
                if (lo >= (int)SURTR_CELL_FV) { L.err = SURTR_E_CAPACITY; break; }
                L.loop[lo++] = (int16_t)cur; ++len;
                const int nx = ring_prev3(L.ring[cur], prev);
                for (int q = 0; q < 3; ++q) if (L.ring[cur][q] == nx) L.seen[cur][q] = 1;
                prev = cur; cur = nx;
            }
            L.flo[nf] = (uint16_t)start;
            ++nf;
        }
    L.flo[nf] = (uint16_t)lo;
    L.nfaces = (uint32_t)nf;
}

// generator of face f = the plane all its loop vertices lie on: the FIRST generator (in cutting order) with the smallest
// worst distance.  Lanes take generators lane, lane + 64, ...; the reduction keeps the smaller error, then the earlier one.
__device__ void cell_face_generator(CellLds& L, uint32_t f)
{
    const int a0 = L.flo[f], a1 = L.flo[f + 1];
    double bestErr = 1e300; uint32_t bestG = 0xFFFFFFFFu;
    for (uint32_t g = lane_id(); g < L.ng; g += SURTR_LANES)
    {
        const D3 gn = L.gn[g];
        const double nl = sqrt(gn.x * gn.x + gn.y * gn.y + gn.z * gn.z);
        double worst = 0;
        for (int t = a0; t < a1; ++t)
        {
            const D3 pv = L.p[L.loop[t]];
            const double e = fabs(gn.x * pv.x + gn.y * pv.y + gn.z * pv.z - L.gc[g]) / nl;
            worst = e > worst ? e : worst;
        }
        if (worst < bestErr) { bestErr = worst; bestG = g; }
    }
    for (int d = SURTR_LANES / 2; d >= 1; d >>= 1)
    {
        const double oe = __shfl_down(bestErr, d, SURTR_LANES);
        const uint32_t og = (uint32_t)__shfl_down((int)bestG, d, SURTR_LANES);
        if (og != 0xFFFFFFFFu && (bestG == 0xFFFFFFFFu || oe < bestErr || (oe == bestErr && og < bestG))) { bestErr = oe; bestG = og; }
    }
    if (lane_id() == 0) L.fgen[f] = bestG == 0xFFFFFFFFu ? -1 : L.gid[bestG];
}

__device__ void cell_output_serial(CellLds& L, const D3 s, CellOut& o)
{
    const int nf = (int)L.nfaces;
    // stable order by generator id (insertion sort of the face numbers)
    for (int f = 0; f < nf; ++f)
    {
        int at = f;
        while (at > 0 && L.fgen[L.forder[at - 1]] > L.fgen[f]) { L.forder[at] = L.forder[at - 1]; --at; }
        L.forder[at] = (uint8_t)f;
    }
    o.nf = (uint32_t)nf; o.nfv = (uint32_t)L.flo[nf];
    uint32_t w = 0;
    for (int k = 0; k < nf; ++k)
    {
        const int f = L.forder[k];
        const int a0 = L.flo[f], len = L.flo[f + 1] - a0;
        // outward winding: (v1-v0)x(v2-v0) must point away from the seed; then start at the lexicographically smallest vertex
        bool rev = false;
        if (len >= 3)
        {
            const D3 a = L.p[L.loop[a0]], b = L.p[L.loop[a0 + 1]], c = L.p[L.loop[a0 + 2]];
            const D3 u{b.x - a.x, b.y - a.y, b.z - a.z}, ww{c.x - a.x, c.y - a.y, c.z - a.z};
            const D3 n{u.y * ww.z - u.z * ww.y, u.z * ww.x - u.x * ww.z, u.x * ww.y - u.y * ww.x};
            rev = n.x * (a.x - s.x) + n.y * (a.y - s.y) + n.z * (a.z - s.z) < 0;
        }
        auto at = [&](int i) { return L.p[L.loop[a0 + (rev ? len - 1 - i : i)]]; };
        int st = 0;
        for (int i = 1; i < len; ++i)
        {
            const D3 p = at(i), q = at(st);
            if (p.x < q.x || (p.x == q.x && (p.y < q.y || (p.y == q.y && p.z < q.z)))) st = i;
        }
        o.gen[k] = L.fgen[f]; o.fvo[k] = (uint16_t)w;
        for (int i = 0; i < len; ++i)
        {
            const D3 p = at((st + i) % len);
            o.v[3 * w] = p.x; o.v[3 * w + 1] = p.y; o.v[3 * w + 2] = p.z; ++w;
        }
    }
    o.fvo[nf] = (uint16_t)w;
}

// cell c of group g: seeds [goff[g], goff[g+1]); one wave per cell
__global__ __launch_bounds__(SURTR_LANES) void k_build_cells(uint32_t n_cells, uint32_t n_groups, const uint32_t* __restrict__ goff,
                                                             const double* __restrict__ seeds, CellOut* __restrict__ outv, uint32_t* __restrict__ heads)
{
    __shared__ CellLds L;
    const uint32_t cell = blockIdx.x, lane = threadIdx.x;
    if (cell >= n_cells) return;
    uint32_t g = 0;
    { uint32_t lo = 0, hi = n_groups; while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (goff[mid] <= cell) lo = mid; else hi = mid; } g = lo; }
    const uint32_t s0 = goff[g], C = goff[g + 1] - s0, self = cell - s0;
    const D3 s{seeds[3 * (size_t)cell], seeds[3 * (size_t)cell + 1], seeds[3 * (size_t)cell + 2]};
    if (lane == 0)
    {
        const double P[8][3] = {{-.5, -.5, -.5}, {.5, -.5, -.5}, {.5, .5, -.5}, {-.5, .5, -.5}, {-.5, -.5, .5}, {.5, -.5, .5}, {.5, .5, .5}, {-.5, .5, .5}};
        const int NB[8][3] = {{1, 4, 3}, {5, 0, 2}, {3, 6, 1}, {7, 2, 0}, {5, 7, 0}, {1, 6, 4}, {5, 2, 7}, {4, 6, 3}};
        for (int i = 0; i < 8; ++i) { L.p[i] = D3{P[i][0], P[i][1], P[i][2]}; for (int e = 0; e < 3; ++e) L.ring[i][e] = (int16_t)NB[i][e]; }
        L.nv = 8; L.err = 0;
        const double WN[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        for (int w = 0; w < 6; ++w) { L.gid[w] = (int32_t)(C + (uint32_t)w); L.gn[w] = D3{WN[w][0], WN[w][1], WN[w][2]}; L.gc[w] = 0.5; }
        L.ng = 6;
    }
    __syncthreads();
    auto radius2 = [&]() {      // (all lanes) max squared distance of a cell vertex from the seed
        double r2 = 0;
        for (uint32_t i = lane; i < L.nv; i += SURTR_LANES)
        {
            const double dx = L.p[i].x - s.x, dy = L.p[i].y - s.y, dz = L.p[i].z - s.z, d = dx * dx + dy * dy + dz * dz;
            r2 = d > r2 ? d : r2;
        }
        for (int d = SURTR_LANES / 2; d >= 1; d >>= 1) { const double o2 = __shfl_down(r2, d, SURTR_LANES); r2 = o2 > r2 ? o2 : r2; }
        __syncthreads();
        if (lane == 0) L.r2 = r2;
        __syncthreads();
    };
    radius2();
    for (uint32_t base = 0; base < C; base += SURTR_LANES)
    {
        const uint32_t o = base + lane;
        bool cand = o < C && o != self;
        D3 q{0, 0, 0}, n{0, 0, 0}; double cc = 0, d2 = 0;
        if (cand)
        {
            q = D3{seeds[3 * (size_t)(s0 + o)], seeds[3 * (size_t)(s0 + o) + 1], seeds[3 * (size_t)(s0 + o) + 2]};
            n = D3{q.x - s.x, q.y - s.y, q.z - s.z};
            cc = 0.5 * ((q.x * q.x + q.y * q.y + q.z * q.z) - (s.x * s.x + s.y * s.y + s.z * s.z));
            d2 = n.x * n.x + n.y * n.y + n.z * n.z;
        }
        while (true)
        {
            bool touch = false;
            // a seed farther than twice the cell's radius (with room for rounding) is on nobody's side: skip the exact test
            if (cand && L.err == 0 && d2 <= 4.0 * L.r2 * 1.00001 + 1e-18)
                for (uint32_t i = 0; i < L.nv; ++i) if (plane_side(n, cc, L.p[i]) > 0) { touch = true; break; }
            const unsigned long long mask = __ballot(touch);
            if (mask == 0ull) break;
            const uint32_t first = (uint32_t)__builtin_ctzll(mask);
            __syncthreads();
            if (lane == first)
            {
                L.cut_n = n; L.cut_c = cc;
                if (L.ng < SURTR_CELL_G) { L.gid[L.ng] = (int32_t)o; L.gn[L.ng] = n; L.gc[L.ng] = cc; ++L.ng; }
                else L.err = SURTR_E_CAPACITY;
            }
            __syncthreads();
            cut_cell_wave(L, L.cut_n, L.cut_c);      // (all lanes: the plane of the lowest touching seed)
            radius2();
            if (lane <= first) cand = false;
        }
    }
    __syncthreads();
    if (lane == 0) { L.nfaces = 0; if (L.err == 0) cell_loops_serial(L); }
    __syncthreads();
    if (L.err == 0) for (uint32_t f = 0; f < L.nfaces; ++f) cell_face_generator(L, f);
    __syncthreads();
    if (lane == 0)
    {
        CellOut& o = outv[cell];
        o.nf = 0; o.nfv = 0;
        if (L.err == 0) cell_output_serial(L, s, o);
        o.err = L.err;
        heads[4 * (size_t)cell] = o.nf; heads[4 * (size_t)cell + 1] = o.nfv; heads[4 * (size_t)cell + 2] = o.err;      // one contiguous read-back
    }
}

// Compact arrays + the pattern (v012 = the first three vertices of every face, narrowed to float) from the per-cell slots.
__global__ __launch_bounds__(SURTR_LANES) void k_pack_cells(uint32_t n_cells, const CellOut* __restrict__ outv, const uint32_t* __restrict__ cfo,
                                                            const uint32_t* __restrict__ cvo, int32_t* __restrict__ gen, uint32_t* __restrict__ fvo,
                                                            double* __restrict__ verts, float* __restrict__ v012)
{
    const uint32_t cell = blockIdx.x;
    if (cell >= n_cells) return;
    const CellOut& o = outv[cell];
    const uint32_t f0 = cfo[cell], v0 = cvo[cell];
    for (uint32_t f = threadIdx.x; f < o.nf; f += group_size())
    {
        gen[f0 + f] = o.gen[f]; fvo[f0 + f] = v0 + o.fvo[f];
        for (int k = 0; k < 9; ++k) v012[9 * (size_t)(f0 + f) + k] = (float)o.v[3 * (size_t)o.fvo[f] + k];
    }
    for (uint32_t i = threadIdx.x; i < 3u * o.nfv; i += group_size()) verts[3 * (size_t)v0 + i] = o.v[i];
    if (cell + 1u == n_cells && threadIdx.x == 0) fvo[f0 + o.nf] = v0 + o.nfv;
}

template <class T>
int grow(surtr_ctx* ctx, T** p, size_t& cap, size_t need)
{
    if (*p && cap >= need) return SURTR_OK;
    free_dev(*p); *p = nullptr; cap = 0;
    if (hipMalloc((void**)p, std::max<size_t>(need, 16) * sizeof(T)) != hipSuccess) { ctx->err = "cell buffer allocation failed"; return SURTR_E_HIP; }
    cap = std::max<size_t>(need, 16);
    return SURTR_OK;
}

} // namespace

extern "C" {

int surtr_build_cells(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_seed_off, const double* seeds, uint32_t* n_faces, uint32_t* n_face_verts)
{
    if (!ctx || n_groups == 0 || !group_seed_off || !seeds || group_seed_off[0] != 0) return SURTR_E_INVALID;
    for (uint32_t g = 0; g < n_groups; ++g) if (group_seed_off[g + 1] <= group_seed_off[g]) return SURTR_E_INVALID;
    const uint32_t n = group_seed_off[n_groups];
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    CellBuffers& B = ctx->cells;
    int rc = grow(ctx, &B.seeds, B.c_seeds, 3 * (size_t)n);
    if (rc == 0) rc = grow(ctx, &B.goff, B.c_goff, (size_t)n_groups + 1);
    if (rc == 0) rc = grow(ctx, &B.slots, B.c_slots, (size_t)n * sizeof(CellOut));
    if (rc == 0) rc = grow(ctx, &B.cfo, B.c_cfo, (size_t)n + 1);
    if (rc == 0) rc = grow(ctx, &B.cvo, B.c_cvo, (size_t)n + 1);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));      // an event may still be reading the old pattern
    HIPCHK(hipMemcpyAsync(B.seeds, seeds, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(B.goff, group_seed_off, ((size_t)n_groups + 1) * 4, hipMemcpyHostToDevice, st));
    CellOut* slots = (CellOut*)B.slots;
    rc = grow(ctx, &B.heads, B.c_heads, 4 * (size_t)n);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_build_cells, dim3(n), dim3(SURTR_LANES), 0, st, n, n_groups, B.goff, B.seeds, slots, B.heads);
    HIPCHK(hipGetLastError());
    // sizes per cell -> offsets (a few bytes per cell cross the bus; the cells themselves stay in HBM)
    std::vector<uint32_t> head(4 * (size_t)n);
    HIPCHK(hipMemcpyAsync(head.data(), B.heads, head.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<uint32_t> cfo(n + 1, 0u), cvo(n + 1, 0u);
    for (uint32_t c = 0; c < n; ++c)
    {
        if (head[4 * (size_t)c + 2]) return (int)head[4 * (size_t)c + 2];
        if (head[4 * (size_t)c] > SURTR_MAXF) return SURTR_E_INVALID;
        cfo[c + 1] = cfo[c] + head[4 * (size_t)c]; cvo[c + 1] = cvo[c] + head[4 * (size_t)c + 1];
    }
    const uint32_t nf = cfo[n], nfv = cvo[n];
    rc = grow(ctx, &B.gen, B.c_gen, nf);
    if (rc == 0) rc = grow(ctx, &B.fvo, B.c_fvo, (size_t)nf + 1);
    if (rc == 0) rc = grow(ctx, &B.verts, B.c_verts, 3 * (size_t)nfv);
    if (rc) return rc;
    // the pattern buffers of the context (what surtr_upload_pattern fills)
    if (!(ctx->d_v012 && ctx->d_planes && ctx->d_plane_off && ctx->cap_pattern_faces >= nf && ctx->cap_pattern_cells >= n))
    {
        free_dev(ctx->d_v012); free_dev(ctx->d_planes); free_dev(ctx->d_plane_off);
        ctx->d_v012 = nullptr; ctx->d_planes = nullptr; ctx->d_plane_off = nullptr; ctx->cap_pattern_faces = 0; ctx->cap_pattern_cells = 0;
        const uint32_t capf = nf + nf / 8 + 16, capc = n + n / 8 + 16;
        HIPCHK(hipMalloc((void**)&ctx->d_v012, (size_t)capf * 36));
        HIPCHK(hipMalloc((void**)&ctx->d_planes, (size_t)capf * 16));
        HIPCHK(hipMalloc((void**)&ctx->d_plane_off, (size_t)(capc + 1) * 4));
        ctx->cap_pattern_faces = capf; ctx->cap_pattern_cells = capc;
    }
    HIPCHK(hipMemcpyAsync(B.cfo, cfo.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(B.cvo, cvo.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->d_plane_off, cfo.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pack_cells, dim3(n), dim3(SURTR_LANES), 0, st, n, slots, B.cfo, B.cvo, B.gen, B.fvo, B.verts, ctx->d_v012);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    ctx->h_plane_off.assign(cfo.begin(), cfo.end());
    ctx->n_cells = n; ctx->n_faces = nf; ctx->planes_ready = false; ctx->pair_order_count = 0;
    B.n = n; B.nf = nf; B.nfv = nfv;
    if (getenv("SURTR_TIMING"))
        fprintf(stderr, "surtr_build_cells: %u cells, kernel + size read-back %.3f ms, pack + pattern %.3f ms\n", n,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    if (n_faces) *n_faces = nf;
    if (n_face_verts) *n_face_verts = nfv;
    return SURTR_OK;
}

int surtr_download_cells(surtr_ctx* ctx, uint32_t* cell_face_off, int32_t* face_gen, uint32_t* face_vert_off, double* verts, float* v012)
{
    if (!ctx) return SURTR_E_INVALID;
    const CellBuffers& B = ctx->cells;
    if (!B.n) return SURTR_E_STATE;
    (void)hipSetDevice(ctx->device);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (cell_face_off) HIPCHK(hipMemcpy(cell_face_off, B.cfo, ((size_t)B.n + 1) * 4, hipMemcpyDeviceToHost));
    if (face_gen && B.nf) HIPCHK(hipMemcpy(face_gen, B.gen, (size_t)B.nf * 4, hipMemcpyDeviceToHost));
    if (face_vert_off) HIPCHK(hipMemcpy(face_vert_off, B.fvo, ((size_t)B.nf + 1) * 4, hipMemcpyDeviceToHost));
    if (verts && B.nfv) HIPCHK(hipMemcpy(verts, B.verts, (size_t)B.nfv * 24, hipMemcpyDeviceToHost));
    if (v012 && B.nf) HIPCHK(hipMemcpy(v012, ctx->d_v012, (size_t)B.nf * 36, hipMemcpyDeviceToHost));
    return SURTR_OK;
}

} // extern "C"
