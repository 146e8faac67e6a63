// regroup_dev.hip -- the step right after the fracture event as a device step: bind sets of
// ApplyFracture (Src/Surtr.cpp:2103-2146), MergeOutOfImpact (:2368-2403, ConvexOutOfSphere :2415-2458) and
// HandleConvexIsland (:2203-2366) on the UN-refitted Convex solids of the last event, which never leave HBM.
//
//   k_rg_count / k_rg_faces   one wave per piece: Poly::ExtractFaces of its Convex; per face the plane
//                             (normalised 3-point plane, |d|) and the polygon's points; ConvexOutOfSphere per fragment
//   (sort)                    faces by (compound, |d|): device radix sort
//   k_rg_pairs                one lane per face: the faces of its compound within 1e-3 of its |d| -> opposite normals ->
//                             point-in-polygon both ways (VMACH::OnYourRight) -> an edge between the two pieces
//   k_rg_labels               min-label propagation over the pieces until nothing changes
// Only per-piece flags and labels (a few bytes per piece) cross the bus; the host turns them into the reference's bind sets
// (first group of a compound stays, the others are appended in discovery order).  Pieces are numbered as in
// surtr_regroup: the resident pieces the event skipped (its `outside` mask), ascending, then the event's fragments.
#include <cstring>
#include <set>

#include "surtr_ctx.h"
#include <hipcub/hipcub.hpp>

#define RG_MAXH 4096u       // half-edges of one Convex the face extraction handles

namespace {

struct P3 { float x, y, z; };
__device__ __forceinline__ P3 sub3(P3 a, P3 b) { return P3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dotp(P3 a, P3 b) { float t = a.x * b.x + a.y * b.y; return t + a.z * b.z; }
__device__ __forceinline__ P3 cross3(P3 a, P3 b) { return P3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ P3 unit3(P3 a)
{
    const float l = sqrtf(dotp(a, a));
    if (!(l != 0.f)) return P3{0.f, 0.f, 0.f};
    return P3{a.x / l, a.y / l, a.z / l};
}
__device__ __forceinline__ bool right_of3(P3 a, P3 b, P3 c, P3 n) { return dotp(cross3(sub3(b, a), sub3(c, a)), n) > 0.f; }   // VMACH::OnYourRight

// A piece's Convex wherever it lives: resident (CSR loff, len = loff[v+1]-loff[v]) or in the arena (loff absolute + llen).
struct RgSolid { const float* pos; const uint32_t* loff; const uint32_t* llen; const int32_t* nbr; uint32_t nv; };
struct RgSrc { const uint32_t* kind; const uint32_t* index; };      // per piece: 0 = resident piece `index`, 1 = fragment `index`

__device__ RgSolid rg_solid(uint32_t p, RgSrc src, const FragRec* __restrict__ frags, Arena A, const float* cpos, const uint32_t* cloff,
                            const int32_t* cnbr, const uint32_t* cvo)
{
    const uint32_t i = src.index[p];
    if (src.kind[p] == 0u)
    {
        const uint32_t a = cvo[i];
        return RgSolid{cpos + 3 * (size_t)a, cloff + a, nullptr, cnbr, cvo[i + 1] - a};
    }
    const FragRec fr = frags[i];
    return RgSolid{A.pos + 3 * (size_t)fr.cv_off, A.loff + fr.cv_off, A.llen + fr.cv_off, A.nbr, fr.cv_n};
}
__device__ __forceinline__ uint32_t rg_len(const RgSolid& S, uint32_t v) { return S.llen ? S.llen[v] : S.loff[v + 1] - S.loff[v]; }
__device__ __forceinline__ P3 rg_pos(const RgSolid& S, int v) { return P3{S.pos[3 * (size_t)v], S.pos[3 * (size_t)v + 1], S.pos[3 * (size_t)v + 2]}; }
__device__ __forceinline__ int rg_before(const RgSolid& S, int v, int who)
{
    const int32_t* r = S.nbr + S.loff[v]; const uint32_t n = rg_len(S, (uint32_t)v);
    uint32_t k = 0;
    while (k < n && r[k] != who) ++k;
    return k == 0 ? r[n - 1] : r[k - 1];
}

struct FaceNode { uint32_t piece, pts_off, pts_n, pad; P3 n; float absd; };

// Poly::ExtractFaces of piece p on one lane (visited set keyed by (vertex, neighbour) = the first slot that holds the
// neighbour, as in the reference); emit(face points...) per face with >= 3 points.  Returns false when the solid is too large.
template <class Emit>
__device__ bool rg_walk_faces(const RgSolid& S, uint8_t* seen /* RG_MAXH */, uint32_t* base /* per vertex, nv <= RG_MAXH */, Emit emit)
{
    uint32_t H = 0;
    if (S.nv > RG_MAXH) return false;
    for (uint32_t v = 0; v < S.nv; ++v) { base[v] = H; H += rg_len(S, v); if (H > RG_MAXH) return false; }
    for (uint32_t e = 0; e < H; ++e) seen[e] = 0;
    auto slot = [&](int a, int b) -> uint32_t {
        const int32_t* r = S.nbr + S.loff[a]; const uint32_t n = rg_len(S, (uint32_t)a);
        uint32_t q = 0;
        while (q < n && r[q] != b) ++q;
        return base[a] + (q < n ? q : 0u);
    };
    for (int i = 0; i < (int)S.nv; ++i)
    {
        const uint32_t deg = rg_len(S, (uint32_t)i);
        for (uint32_t s = 0; s < deg; ++s)
        {
            const int adj = (S.nbr + S.loff[i])[s];
            if (seen[slot(i, adj)]) continue;
            emit(i, -1, 0);                                   // start of a face
            int prev = i, cur = adj; uint32_t len = 1;
            while (cur != i && len <= S.nv * 8u + 8u)
            {
                seen[slot(prev, cur)] = 1;
                emit(cur, -1, 1);
                const int nx = rg_before(S, cur, prev);
                prev = cur; cur = nx; ++len;
            }
            seen[slot(prev, cur)] = 1;
            emit(-1, -1, 2);                                  // end of the face
        }
    }
    return true;
}

// pass 0: counts (faces with >= 3 points, their points) per piece; pass 1: the face nodes + ConvexOutOfSphere
__global__ __launch_bounds__(SURTR_LANES) void k_rg_faces(uint32_t n_pieces, RgSrc src, const FragRec* __restrict__ frags, Arena A, const float* cpos,
                                                          const uint32_t* cloff, const int32_t* cnbr, const uint32_t* cvo, uint32_t pass,
                                                          uint32_t* __restrict__ cnt /* 2 per piece */, const uint32_t* __restrict__ face_off,
                                                          const uint32_t* __restrict__ pts_off, FaceNode* __restrict__ nodes, P3* __restrict__ pts,
                                                          uint32_t n_sphere, const float* __restrict__ sphere, float ox, float oy, float oz, float radius,
                                                          uint8_t* __restrict__ out_flag, uint32_t* __restrict__ err)
{
    __shared__ uint8_t seen[RG_MAXH];
    __shared__ uint32_t base[RG_MAXH];
    __shared__ uint32_t sh_in;
    const uint32_t p = blockIdx.x;
    if (p >= n_pieces) return;
    const RgSolid S = rg_solid(p, src, frags, A, cpos, cloff, cnbr, cvo);
    if (threadIdx.x == 0)
    {
        uint32_t nf = 0, np = 0, cur_n = 0, cur_start = 0;
        const uint32_t f0 = pass ? face_off[p] : 0u, p0 = pass ? pts_off[p] : 0u;
        bool inside_possible = true;      // ConvexOutOfSphere part 1: every vertex at least `radius` from the origin (:2420-2427)
        if (pass && out_flag)
            for (uint32_t v = 0; v < S.nv; ++v)
            {
                const P3 d = sub3(P3{ox, oy, oz}, rg_pos(S, (int)v));
                if (sqrtf(dotp(d, d)) < radius) { inside_possible = false; break; }
            }
        const bool ok = rg_walk_faces(S, seen, base, [&](int v, int, int what) {
            if (what == 0) { cur_n = 0; cur_start = np; }
            if (what <= 1) { if (pass) pts[p0 + np] = rg_pos(S, v); ++np; ++cur_n; }
            if (what == 2)
            {
                if (cur_n < 3u) { np = cur_start; return; }      // (:2243: faces of fewer than three points are skipped)
                if (pass)
                {
                    const P3 a = pts[p0 + cur_start], b = pts[p0 + cur_start + 1], c = pts[p0 + cur_start + 2];
                    const P3 n = unit3(cross3(sub3(a, b), sub3(a, c)));      // Plane(p0, p1, p2), normalised
                    const float d = -dotp(n, a);
                    FaceNode fn; fn.piece = p; fn.pts_off = p0 + cur_start; fn.pts_n = cur_n; fn.pad = 0;
                    fn.n = unit3(n); fn.absd = fabsf(d);                      // .Normal() re-normalised (:2252-2254)
                    nodes[f0 + nf] = fn;
                }
                ++nf;
            }
        });
        if (!ok) atomicMax(err, (uint32_t)SURTR_E_CAPACITY);
        if (!pass) { cnt[2 * p] = nf; cnt[2 * p + 1] = np; }
        sh_in = inside_possible ? nf : 0xFFFFFFFFu;
    }
    __syncthreads();
    if (pass && out_flag)
    {
        // ConvexOutOfSphere part 2 (:2429-2455): no point of the sphere cloud inside the Convex (all its face planes)
        const uint32_t nf = sh_in;
        bool hit = nf == 0xFFFFFFFFu;
        if (!hit)
            for (uint32_t q = threadIdx.x; q < n_sphere && !hit; q += group_size())
            {
                const P3 po{sphere[3 * q], sphere[3 * q + 1], sphere[3 * q + 2]};
                bool contain = true;
                for (uint32_t f = 0; f < nf && contain; ++f)
                {
                    const FaceNode& fn = nodes[face_off[p] + f];
                    const P3 a = pts[fn.pts_off], b = pts[fn.pts_off + 1], c = pts[fn.pts_off + 2];
                    const P3 n = unit3(cross3(sub3(b, a), sub3(c, a)));
                    const float d = -dotp(a, n);
                    if (dotp(n, po) + d > 0.f) contain = false;
                }
                if (contain) hit = true;
            }
        const bool any = __ballot(hit) != 0ull;
        if (threadIdx.x == 0) out_flag[p] = any ? 0 : 1;
    }
}

__global__ void k_rg_keys(uint32_t nf, const FaceNode* __restrict__ nodes, const uint32_t* __restrict__ set_of, unsigned long long* __restrict__ key,
                          uint32_t* __restrict__ val)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    uint32_t bits; const float ad = nodes[f].absd;
    memcpy(&bits, &ad, 4);                                                                               // |d| >= 0: the bits order like the value
    key[f] = ((unsigned long long)set_of[nodes[f].piece] << 32) | bits;
    val[f] = f;
}

__global__ void k_rg_pairs(uint32_t nf, const uint32_t* __restrict__ order, const unsigned long long* __restrict__ skey, const FaceNode* __restrict__ nodes,
                           const P3* __restrict__ pts, uint2* __restrict__ edges, uint32_t cap_edges, uint32_t* __restrict__ n_edges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf) return;
    const FaceNode A = nodes[order[i]];
    const uint32_t set = (uint32_t)(skey[i] >> 32);
    for (uint32_t j = i + 1; j < nf; ++j)
    {
        if ((uint32_t)(skey[j] >> 32) != set) break;
        const FaceNode B = nodes[order[j]];
        if (fabs((double)A.absd - (double)B.absd) > 1e-3) break;                 // sorted: every later face is farther still
        if (!((double)fabsf(1.f + dotp(A.n, B.n)) < 1e-4)) continue;             // normals must be opposite
        bool touch = false;
        for (uint32_t a = 0; a < A.pts_n && !touch; ++a)
        {
            const P3 ip = pts[A.pts_off + a];
            bool inside = true;
            for (uint32_t v = 0; v < B.pts_n; ++v)
                if (!right_of3(pts[B.pts_off + v], pts[B.pts_off + (v + 1u) % B.pts_n], ip, B.n)) { inside = false; break; }
            if (inside) touch = true;
        }
        for (uint32_t b = 0; b < B.pts_n && !touch; ++b)
        {
            const P3 jp = pts[B.pts_off + b];
            bool inside = true;
            for (uint32_t v = 0; v < A.pts_n; ++v)
                if (!right_of3(pts[A.pts_off + v], pts[A.pts_off + (v + 1u) % A.pts_n], jp, A.n)) { inside = false; break; }
            if (inside) touch = true;
        }
        if (touch && A.piece != B.piece)
        {
            const uint32_t at = atomicAdd(n_edges, 1u);
            if (at < cap_edges) edges[at] = make_uint2(A.piece, B.piece);
        }
    }
}

// one round of min-label propagation over the edges + one pointer jump; *changed != 0 when a label moved
__global__ void k_rg_labels(uint32_t ne, const uint2* __restrict__ edges, uint32_t n_pieces, uint32_t* __restrict__ lab, uint32_t* __restrict__ changed)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ne)
    {
        const uint32_t a = edges[e].x, b = edges[e].y;
        const uint32_t la = lab[a], lb = lab[b], m = la < lb ? la : lb;
        if (m < la) { atomicMin(&lab[a], m); *changed = 1u; }
        if (m < lb) { atomicMin(&lab[b], m); *changed = 1u; }
    }
    if (e < n_pieces) { const uint32_t l = lab[e], ll = lab[l]; if (ll < l) { atomicMin(&lab[e], ll); *changed = 1u; } }
}

// The label rounds to convergence in ONE launch: a single workgroup sweeps the edges and the pointer jumps round after round
// (labels only decrease, so a round that changes nothing ends it; at most n_pieces + 8 rounds, as the host loop had) -- no
// host round trip per round.  rounds_out: rounds run, or 0xFFFFFFFF when the bound was hit.
#define RG_LABEL_THREADS 1024u
__global__ __launch_bounds__(RG_LABEL_THREADS) void k_rg_labels_all(uint32_t ne, const uint2* __restrict__ edges, uint32_t n_pieces, uint32_t* __restrict__ lab,
                                                                    uint32_t* __restrict__ rounds_out)
{
    __shared__ uint32_t changed[2];
    const uint32_t tid = threadIdx.x, G = blockDim.x;
    if (tid < 2u) changed[tid] = 0u;
    __syncthreads();
    uint32_t round = 0;
    for (; round < n_pieces + 8u; ++round)
    {
        bool ch = false;
        for (uint32_t e = tid; e < ne; e += G)
        {
            const uint32_t a = edges[e].x, b = edges[e].y;
            const uint32_t la = lab[a], lb = lab[b], m = la < lb ? la : lb;
            if (m < la) { atomicMin(&lab[a], m); ch = true; }
            if (m < lb) { atomicMin(&lab[b], m); ch = true; }
        }
        __syncthreads();
        for (uint32_t v = tid; v < n_pieces; v += G) { const uint32_t l = lab[v], ll = lab[l]; if (ll < l) { atomicMin(&lab[v], ll); ch = true; } }
        if (ch) changed[round & 1u] = 1u;
        __syncthreads();
        const bool any = changed[round & 1u] != 0u;
        if (tid == 0u) changed[(round + 1u) & 1u] = 0u;
        __syncthreads();
        if (!any) break;
    }
    if (tid == 0u) *rounds_out = round < n_pieces + 8u ? round + 1u : 0xFFFFFFFFu;
}

template <class T>
struct Tmp
{
    T* p = nullptr;
    bool alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 4) * sizeof(T)) == hipSuccess; }
    ~Tmp() { if (p) (void)hipFree(p); }
};

} // namespace

extern "C" int surtr_event_regroup(surtr_ctx* ctx, int partial, uint32_t n_sphere, const float* sphere_points, const float origin[3], float radius,
                                   uint32_t* n_pieces_out, uint32_t* n_compounds, uint32_t* compound_off, int32_t* compound_piece)
{
    if (!ctx || !n_compounds) return SURTR_E_INVALID;
    if (!ctx->have_event) return SURTR_E_STATE;
    if (partial && (!origin || (n_sphere && !sphere_points))) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    surtr_counts c;
    int rc = surtr_event_counts(ctx, &c);
    if (rc) return rc;
    // pieces: the resident pieces the event skipped, then its fragments; bind sets as ApplyFracture leaves them
    std::vector<uint32_t> kind, index, set_of;
    for (uint32_t p = 0; p < ctx->n_pieces && ctx->last_outside.size() == ctx->n_pieces; ++p)
        if (ctx->last_outside[p]) { kind.push_back(0u); index.push_back(p); }
    const uint32_t n_outside = (uint32_t)kind.size();
    std::vector<FragRec> fr(c.n_frag);
    if (c.n_frag) HIPCHK(hipMemcpy(fr.data(), ctx->d_frags, (size_t)c.n_frag * sizeof(FragRec), hipMemcpyDeviceToHost));
    for (uint32_t f = 0; f < c.n_frag; ++f) { kind.push_back(1u); index.push_back(f); }
    const uint32_t n = (uint32_t)kind.size();
    if (n_pieces_out) *n_pieces_out = n;
    if (!compound_off || !compound_piece) { *n_compounds = 0; return SURTR_OK; }      // sizes only: n + 2 / n entries are enough
    std::vector<std::set<int>> bind(1);
    for (uint32_t p = 0; p < n_outside; ++p) bind[0].insert((int)p);
    for (uint32_t f = 0; f < c.n_frag; ++f)
    {
        if (f == 0 || fr[f].cell != fr[f - 1].cell) bind.emplace_back();
        bind.back().insert((int)(n_outside + f));
    }
    if (n == 0) { *n_compounds = 1; compound_off[0] = compound_off[1] = 0; return SURTR_OK; }
    Tmp<uint32_t> d_kind, d_index, d_cnt, d_foff, d_poff, d_set, d_val, d_order, d_lab, d_flag32; Tmp<uint8_t> d_out; Tmp<float> d_sph;
    if (!d_kind.alloc(n) || !d_index.alloc(n) || !d_cnt.alloc(2 * (size_t)n) || !d_foff.alloc(n + 1) || !d_poff.alloc(n + 1) || !d_set.alloc(n) ||
        !d_lab.alloc(n) || !d_flag32.alloc(4) || !d_out.alloc(n) || !d_sph.alloc(3 * (size_t)n_sphere + 3)) return SURTR_E_HIP;
    HIPCHK(hipMemcpyAsync(d_kind.p, kind.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_index.p, index.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_flag32.p, 0, 16, st));
    if (n_sphere) HIPCHK(hipMemcpyAsync(d_sph.p, sphere_points, (size_t)n_sphere * 12, hipMemcpyHostToDevice, st));
    const RgSrc src{d_kind.p, d_index.p};
    const PieceSet& C = ctx->cset;
    const float ox = origin ? origin[0] : 0.f, oy = origin ? origin[1] : 0.f, oz = origin ? origin[2] : 0.f;
    uint32_t* d_err = d_flag32.p;            // [0] error, [1] edges, [2] changed
    hipLaunchKernelGGL(k_rg_faces, dim3(n), dim3(SURTR_LANES), 0, st, n, src, ctx->d_frags, ctx->arena, C.pos, C.loff, C.nbr, C.vo, 0u, d_cnt.p,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (FaceNode*)nullptr, (P3*)nullptr, 0u, (const float*)nullptr, 0.f, 0.f, 0.f, 0.f,
                       (uint8_t*)nullptr, d_err);
    std::vector<uint32_t> cnt(2 * (size_t)n), foff(n + 1, 0u), poff(n + 1, 0u);
    HIPCHK(hipMemcpyAsync(cnt.data(), d_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint32_t p = 0; p < n; ++p) { foff[p + 1] = foff[p] + cnt[2 * p]; poff[p + 1] = poff[p] + cnt[2 * p + 1]; }
    const uint32_t nf = foff[n], npts = poff[n];
    Tmp<FaceNode> d_nodes; Tmp<P3> d_pts; Tmp<unsigned long long> d_key, d_key2; Tmp<uint2> d_edges;
    const uint32_t cap_edges = 16u * nf + 1024u;
    if (!d_nodes.alloc(nf) || !d_pts.alloc(npts) || !d_key.alloc(nf) || !d_key2.alloc(nf) || !d_val.alloc(nf) || !d_order.alloc(nf) || !d_edges.alloc(cap_edges))
        return SURTR_E_HIP;
    HIPCHK(hipMemcpyAsync(d_foff.p, foff.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_poff.p, poff.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rg_faces, dim3(n), dim3(SURTR_LANES), 0, st, n, src, ctx->d_frags, ctx->arena, C.pos, C.loff, C.nbr, C.vo, 1u, d_cnt.p,
                       d_foff.p, d_poff.p, d_nodes.p, d_pts.p, n_sphere, d_sph.p, ox, oy, oz, radius, partial ? d_out.p : (uint8_t*)nullptr, d_err);
    HIPCHK(hipGetLastError());
    if (partial)       // MergeOutOfImpact (:2368-2403): fragments out of the impact sphere move to bind 0; emptied compounds go
    {
        std::vector<uint8_t> flag(n);
        HIPCHK(hipMemcpyAsync(flag.data(), d_out.p, n, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (size_t i = 1; i < bind.size(); ++i)
        {
            std::set<int> outside;
            for (int cpi : bind[i]) if (flag[cpi]) outside.insert(cpi);
            for (int cpi : outside) { bind[i].erase(cpi); bind[0].insert(cpi); }
        }
        bind.erase(std::remove_if(bind.begin() + 1, bind.end(), [](const std::set<int>& s) { return s.empty(); }), bind.end());
    }
    set_of.assign(n, 0u);
    for (size_t i = 0; i < bind.size(); ++i) for (int p : bind[i]) set_of[p] = (uint32_t)i;
    HIPCHK(hipMemcpyAsync(d_set.p, set_of.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    std::vector<uint32_t> lab(n);
    for (uint32_t p = 0; p < n; ++p) lab[p] = p;
    HIPCHK(hipMemcpyAsync(d_lab.p, lab.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (nf)
    {
        const dim3 blk(256), grid((nf + 255) / 256);
        hipLaunchKernelGGL(k_rg_keys, grid, blk, 0, st, nf, d_nodes.p, d_set.p, d_key.p, d_val.p);
        size_t tmp_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_key.p, d_key2.p, d_val.p, d_order.p, (int)nf, 0, 64, st);
        Tmp<char> d_tmp;
        if (!d_tmp.alloc(tmp_bytes + 16)) return SURTR_E_HIP;
        if (hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, d_key.p, d_key2.p, d_val.p, d_order.p, (int)nf, 0, 64, st) != hipSuccess) return SURTR_E_HIP;
        HIPCHK(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_rg_pairs, grid, blk, 0, st, nf, d_order.p, d_key2.p, d_nodes.p, d_pts.p, d_edges.p, cap_edges, d_err + 1);
        uint32_t head[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(head, d_err, 12, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (head[0]) return (int)head[0];
        if (head[1] > cap_edges) return SURTR_E_CAPACITY;
        const uint32_t ne = head[1];
        // min-label propagation with one pointer jump per round converges in a number of rounds bounded by the number of
        // pieces (labels only decrease); it runs until a round changes nothing -- a chain numbered against the grain takes as
        // many rounds as it is long, not 64
        uint32_t rounds = 1u;
        if (ne)
        {
            hipLaunchKernelGGL(k_rg_labels_all, dim3(1), dim3(SURTR_LANES == 1u ? 1u : RG_LABEL_THREADS)      /* (the emulation runs one thread per workgroup) */, 0, st, ne, d_edges.p, n, d_lab.p, d_err + 2);
            HIPCHK(hipMemcpyAsync(&rounds, d_err + 2, 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipMemcpyAsync(lab.data(), d_lab.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));        // (the one synchronisation of the label phase)
        if (rounds == 0xFFFFFFFFu) return SURTR_E_STATE;      // (cannot happen: see the bound in the kernel; never hand out unconverged labels)
        ctx->regroup_rounds = rounds;
    }
    // HandleConvexIsland's outcome: per compound, groups = label classes in order of their lowest piece; the first stays
    std::vector<std::set<int>> extra;
    for (auto& local : bind)
    {
        if (local.size() <= 1) continue;
        std::vector<std::pair<uint32_t, std::set<int>>> groups;      // (label = lowest piece, members)
        for (int p : local)
        {
            bool found = false;
            for (auto& g : groups) if (g.first == lab[p]) { g.second.insert(p); found = true; break; }
            if (!found) groups.push_back({lab[p], std::set<int>{p}});
        }
        std::sort(groups.begin(), groups.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        if (groups.size() >= 2)
        {
            local = groups[0].second;
            for (size_t g = 1; g < groups.size(); ++g) extra.push_back(groups[g].second);
        }
    }
    bind.insert(bind.end(), extra.begin(), extra.end());
    uint32_t at = 0;
    compound_off[0] = 0;
    for (size_t i = 0; i < bind.size(); ++i)
    {
        for (int p : bind[i]) compound_piece[at++] = p;
        compound_off[i + 1] = at;
    }
    *n_compounds = (uint32_t)bind.size();
    return SURTR_OK;
}
