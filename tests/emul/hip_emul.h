// hip_emul.h -- TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for <hip/hip_runtime.h> that lets surtr_amd/csrc/*.hip be
// compiled by g++ as a single-lane, single-thread-per-workgroup emulation
// (SURTR_EMUL: wave = 1 lane, workgroup = 1 thread, blocks run one after the
// other).  It exists so the CPU test tier can exercise the *kernel logic* of
// the product against the oracle without a GPU; it cannot show races and is
// never loaded by surtr_amd (engine.py only opens libsurtr_hip.so).
#pragma once
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

typedef int hipError_t;
typedef void* hipStream_t;
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
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                                   \
        dim3 g_ = (grid), b_ = (block);                                                    \
        gridDim.x = g_.x; blockDim.x = b_.x;                                               \
        for (uint32_t bi_ = 0; bi_ < g_.x; ++bi_)                                          \
            for (uint32_t ti_ = 0; ti_ < b_.x; ++ti_) { blockIdx.x = bi_; threadIdx.x = ti_; kern(__VA_ARGS__); } \
    } while (0)
template <class T> static inline T __shfl(T v, int, int) { return v; }
