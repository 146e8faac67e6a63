// hip_emul.h -- TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for <hip/hip_runtime.h> that lets surtr_amd/csrc/*.hip be
// compiled by g++ as a single-lane, single-thread-per-workgroup emulation
// (SURTR_EMUL: wave = 1 lane, workgroup = 1 thread, blocks run one after the
// other).  It exists so the CPU test tier can exercise the *kernel logic* of
// the product against the oracle without a GPU; it cannot show races and is
// never loaded by surtr_amd (engine.py only opens libsurtr_hip.so).
#pragma once
// geometry and capacities of the emulation: one lane per wave, one thread per workgroup; small thresholds so that small test
// meshes still go through the culling, the separate pre-pass kernel and the global-scratch fallback of the one-wave kernels
#define SURTR_LANES 1
#define SURTR_LSH 0
#define SURTR_WG 1
#define SURTR_SB 1u
#ifndef SURTR_KEEPALL_V
#define SURTR_KEEPALL_V 12u
#endif
#define SURTR_PREP_MINV 48u
// one vertex per "64-block" and per sphere group here: the LDS tables of k_prep_pairs (static arrays in this build) are sized so
// that the test meshes take the same paths as 50 000-vertex pieces do on the device
#define SURTR_PREP_NB 65536u
#define SURTR_MAIN_THREADS 1u      // (one thread per workgroup in this build)
#define SURTR_PS_NB 57344u
#define SURTR_SMALL_LV 64
#define SURTR_SMALL_LH 512
#include <cstdio>
#define SURTR_DBG(...) fprintf(stderr, __VA_ARGS__)
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t a = 1, uint32_t b = 1, uint32_t c = 1) : x(a), y(b), z(c) {} };

struct EmulIdx { uint32_t x = 0, y = 0, z = 0; };
static EmulIdx threadIdx, blockIdx, blockDim, gridDim;

static inline void __syncthreads() {}
template <class T> static inline T __shfl_up(T v, int, int) { return v; }
template <class T> static inline T __shfl_down(T v, int, int) { return v; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// one lane per wave: the cross-lane primitives degenerate to the identity
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __syncthreads_or(int p) { return p; }
static inline bool __all(bool p) { return p; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline int __builtin_amdgcn_readlane(int v, int) { return v; }
static inline int __builtin_amdgcn_update_dpp(int old, int, int, int, int, bool) { return old; }      // lane 0 never has a source lane
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
struct hipDeviceProp_t { int multiProcessorCount; };
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 1; return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
// streams and events: everything runs in program order
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// the two hipCUB device algorithms the host code calls (stable LSD radix sort of pairs by key bits; exclusive sum)
#include <algorithm>
#include <vector>
namespace hipcub {
struct DeviceRadixSort
{
    template <class K, class V>
    static hipError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const V* vin, V* vout, int n, int begin_bit, int end_bit, hipStream_t)
    {
        if (!tmp) { tmp_bytes = 16; return hipSuccess; }
        std::vector<int> ord(n);
        for (int i = 0; i < n; ++i) ord[i] = i;
        const K mask = end_bit >= (int)(8 * sizeof(K)) ? ~(K)0 : (((K)1 << end_bit) - 1);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return ((kin[a] & mask) >> begin_bit) < ((kin[b] & mask) >> begin_bit); });
        for (int i = 0; i < n; ++i) { kout[i] = kin[ord[i]]; vout[i] = vin[ord[i]]; }
        return hipSuccess;
    }
};
struct DeviceScan
{
    template <class T>
    static hipError_t ExclusiveSum(void* tmp, size_t& tmp_bytes, const T* in, T* out, int n, hipStream_t)
    {
        if (!tmp) { tmp_bytes = 16; return hipSuccess; }
        T run = 0;
        for (int i = 0; i < n; ++i) { const T v = in[i]; out[i] = run; run += v; }
        return hipSuccess;
    }
};
} // namespace hipcub

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                                   \
        dim3 g_ = (grid), b_ = (block);                                                    \
        gridDim.x = g_.x; blockDim.x = b_.x;                                               \
        for (uint32_t bi_ = 0; bi_ < g_.x; ++bi_)                                          \
            for (uint32_t ti_ = 0; ti_ < b_.x; ++ti_) { blockIdx.x = bi_; threadIdx.x = ti_; kern(__VA_ARGS__); } \
    } while (0)
template <class T> static inline T __shfl(T v, int, int) { return v; }
