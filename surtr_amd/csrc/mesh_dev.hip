// mesh_dev.hip -- Poly::ExtractNeighborFromMesh (Src/Poly.cpp:128-263) on the device: welded,
// closed, consistently wound triangle soup -> neighbour rings with the reference's ring order AND rotation.
//
// Same formulation as the host helper rings_from_triangles (host_geom.cpp): a table of directed edges
// (x -> y) |-> the vertex after y in that triangle; the ring of v is the cycle a, third(v,a), third(v,third(v,a)), ...
// started from the lowest-numbered triangle that holds v, rotated by two places when v is that triangle's first corner.
//   k_edges    one lane per triangle corner: insert the directed edge into an open-addressing hash table (64-bit CAS),
//              atomicMin of the triangle number per vertex; a duplicate edge = non-manifold / inconsistent winding
//   k_degree   one lane per vertex: walk the fan once, count
//   (scan)     ring offsets
//   k_rings    one lane per vertex: walk again, write the ring with its rotation
//   k_links    symmetric-link check (:253-260)
#include <cstring>

#include "surtr_ctx.h"
#include <hipcub/hipcub.hpp>

namespace {

#define EDGE_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint32_t edge_hash(unsigned long long k, uint32_t mask)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k & mask;
}

__global__ void k_edges(uint32_t nv, uint32_t nt, const int32_t* __restrict__ tris, unsigned long long* __restrict__ keys, int32_t* __restrict__ third,
                        uint32_t mask, uint32_t* __restrict__ first_tri, uint32_t* __restrict__ err)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3u * nt) return;
    const uint32_t t = i / 3u, c = i % 3u;
    const int32_t* q = tris + 3 * (size_t)t;
    const int32_t x = q[c], y = q[(c + 1u) % 3u], z = q[(c + 2u) % 3u];
    if (x < 0 || (uint32_t)x >= nv || y < 0 || (uint32_t)y >= nv || x == y) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); return; }
    const unsigned long long key = ((unsigned long long)(uint32_t)x << 32) | (uint32_t)y;
    uint32_t h = edge_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe, h = (h + 1u) & mask)
    {
        const unsigned long long old = atomicCAS(&keys[h], EDGE_EMPTY, key);
        if (old == EDGE_EMPTY) { third[h] = z; break; }
        if (old == key) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); break; }      // non-manifold or inconsistent winding
    }
    atomicMin(&first_tri[x], t);
}

__device__ __forceinline__ int32_t edge_third(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ third, uint32_t mask, uint32_t x, uint32_t y)
{
    const unsigned long long key = ((unsigned long long)x << 32) | y;
    uint32_t h = edge_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe, h = (h + 1u) & mask)
    {
        const unsigned long long k = keys[h];
        if (k == key) return third[h];
        if (k == EDGE_EMPTY) return -1;
    }
    return -1;
}

// fill == 0: deg[v] = ring length; fill != 0: write the ring at off[v]
__global__ void k_fans(uint32_t nv, uint32_t nt, const int32_t* __restrict__ tris, const unsigned long long* __restrict__ keys,
                       const int32_t* __restrict__ third, uint32_t mask, const uint32_t* __restrict__ first_tri, uint32_t* __restrict__ deg,
                       const uint32_t* __restrict__ off, int32_t* __restrict__ nbr, uint32_t fill, uint32_t* __restrict__ err)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const uint32_t t0 = first_tri[v];
    if (t0 == 0xFFFFFFFFu) { if (!fill) deg[v] = 0; return; }
    const int32_t* q = tris + 3 * (size_t)t0;
    uint32_t s = 0;
    for (uint32_t c = 0; c < 3; ++c) if (q[c] == (int32_t)v) { s = c; break; }
    const int32_t a = q[(s + 1u) % 3u];
    int32_t cur = a;
    uint32_t n = 0;
    const uint32_t len = fill ? off[v + 1] - off[v] : 0u;
    // the rotation of the reference (see rings_from_triangles): by two places when v is the first corner of its first triangle
    const uint32_t rot = (fill && s == 0u && len >= 3u) ? 2u : 0u;
    do
    {
        if (fill) nbr[off[v] + (n + len - rot) % len] = cur;
        ++n;
        cur = edge_third(keys, third, mask, v, (uint32_t)cur);      // triangle (v, cur, next)
        if (cur < 0 || n > nt) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); break; }      // open fan
    } while (cur != a);
    if (!fill) deg[v] = n;
}

__global__ void k_links(uint32_t nv, const uint32_t* __restrict__ off, const int32_t* __restrict__ nbr, uint32_t* __restrict__ err)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    for (uint32_t j = off[v]; j < off[v + 1]; ++j)
    {
        const int32_t u = nbr[j];
        bool back = false;
        for (uint32_t k = off[u]; k < off[u + 1]; ++k) if (nbr[k] == (int32_t)v) { back = true; break; }
        if (!back) { atomicMax(err, (uint32_t)SURTR_E_TOPOLOGY); return; }
    }
}

} // namespace

extern "C" int surtr_neighbors_from_mesh_dev(surtr_ctx* ctx, uint32_t nv, uint32_t nt, const int32_t* tris, uint32_t* nbr_off, int32_t* nbr, float* kernel_ms)
{
    if (!ctx || !tris || !nbr_off || !nbr || nv == 0 || nt == 0) return SURTR_E_INVALID;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    uint32_t cap = 16; while (cap < 6u * nt) cap <<= 1;      // load factor <= 1/2
    int32_t *d_tris = nullptr, *d_third = nullptr, *d_nbr = nullptr; unsigned long long* d_keys = nullptr;
    uint32_t *d_first = nullptr, *d_deg = nullptr, *d_off = nullptr, *d_err = nullptr; void* d_tmp = nullptr;
    auto cleanup = [&]() { free_dev(d_tris); free_dev(d_third); free_dev(d_nbr); free_dev(d_keys); free_dev(d_first); free_dev(d_deg); free_dev(d_off); free_dev(d_err); free_dev(d_tmp); };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx->err = hipGetErrorString(e_); cleanup(); return SURTR_E_HIP; } } while (0)
    CK(hipMalloc((void**)&d_tris, (size_t)nt * 12)); CK(hipMalloc((void**)&d_third, (size_t)cap * 4)); CK(hipMalloc((void**)&d_keys, (size_t)cap * 8));
    CK(hipMalloc((void**)&d_first, (size_t)nv * 4)); CK(hipMalloc((void**)&d_deg, ((size_t)nv + 1) * 4)); CK(hipMalloc((void**)&d_off, ((size_t)nv + 1) * 4));
    CK(hipMalloc((void**)&d_nbr, (size_t)nt * 12 + 16)); CK(hipMalloc((void**)&d_err, 16));
    CK(hipMemcpyAsync(d_tris, tris, (size_t)nt * 12, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(d_keys, 0xFF, (size_t)cap * 8, st)); CK(hipMemsetAsync(d_first, 0xFF, (size_t)nv * 4, st)); CK(hipMemsetAsync(d_err, 0, 4, st));
    CK(hipMemsetAsync(d_deg, 0, ((size_t)nv + 1) * 4, st));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kernel_ms) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
    const dim3 blk(256);
    hipLaunchKernelGGL(k_edges, dim3((3u * nt + 255) / 256), blk, 0, st, nv, nt, d_tris, d_keys, d_third, cap - 1u, d_first, d_err);
    hipLaunchKernelGGL(k_fans, dim3((nv + 255) / 256), blk, 0, st, nv, nt, d_tris, d_keys, d_third, cap - 1u, d_first, d_deg, d_off, d_nbr, 0u, d_err);
    uint32_t err = 0;
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_deg, d_off, (int)nv + 1, st);
    CK(hipMalloc(&d_tmp, tmp_bytes + 16));
    if (hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_deg, d_off, (int)nv + 1, st) != hipSuccess) { cleanup(); return SURTR_E_HIP; }
    // sum of the ring lengths = 3 T on a closed manifold; anything else was an error (checked before the rings are written)
    uint32_t total = 0;
    CK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&total, d_off + nv, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (err == 0 && total > 3u * nt) err = SURTR_E_TOPOLOGY;
    if (err == 0)
    {
        hipLaunchKernelGGL(k_fans, dim3((nv + 255) / 256), blk, 0, st, nv, nt, d_tris, d_keys, d_third, cap - 1u, d_first, d_deg, d_off, d_nbr, 1u, d_err);
        hipLaunchKernelGGL(k_links, dim3((nv + 255) / 256), blk, 0, st, nv, d_off, d_nbr, d_err);
        if (kernel_ms) (void)hipEventRecord(e1, st);
        CK(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
        CK(hipMemcpyAsync(nbr_off, d_off, ((size_t)nv + 1) * 4, hipMemcpyDeviceToHost, st));
        if (total) CK(hipMemcpyAsync(nbr, d_nbr, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
    }
    if (kernel_ms)
    {
        *kernel_ms = -1.f;
        if (err == 0) (void)hipEventElapsedTime(kernel_ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
#undef CK
    cleanup();
    return (int)err;
}
