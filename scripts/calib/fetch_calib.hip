// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access widths of the clip kernels (diagnostic tool, not
// part of the product).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read;
// other widths are uncalibrated.  Each kernel below reads a KNOWN number of bytes from a buffer far larger than the 256 MiB
// Infinity Cache, once, with one access width:  k16 16 B/lane (dwordx4), k4 4 B/lane (dword), k12 three dwords at stride 12
// (positions as the kernels read them), k2 2 B/lane (the 16-bit topology images), g4 a 4-B gather at a pseudo-random index.
// Run under `rocprofv3 --pmc FETCH_SIZE` (scripts/calib/fetch_calib.sh): FETCH_SIZE[KB] * 1024 / bytes = the factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k16(const uint4* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void k4(const unsigned* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void k12(const float* __restrict__ p, size_t n, unsigned* out)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    if (acc == 0.12345f) out[0] = 1u;
}
__global__ void k2(const unsigned short* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void g4(const unsigned* __restrict__ p, size_t n, size_t reads, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < reads; i += (size_t)gridDim.x * blockDim.x)
    {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        acc ^= p[h % n];
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)3 << 30;           // 3 GiB: twelve times the Infinity Cache
    void* buf = nullptr; unsigned* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&out, 16) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    const dim3 grid(256 * 16), blk(256);
    hipLaunchKernelGGL(k16, grid, blk, 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k4, grid, blk, 0, 0, (const unsigned*)buf, bytes / 4, out);
    hipLaunchKernelGGL(k12, grid, blk, 0, 0, (const float*)buf, bytes / 12, out);
    hipLaunchKernelGGL(k2, grid, blk, 0, 0, (const unsigned short*)buf, bytes / 2 / 4, out);      // a quarter of the buffer: 2-byte loads are slow
    hipLaunchKernelGGL(g4, grid, blk, 0, 0, (const unsigned*)buf, bytes / 4, (size_t)1 << 26, out);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    printf("{\"k16\": %zu, \"k4\": %zu, \"k12\": %zu, \"k2\": %zu, \"g4_useful\": %zu, \"g4_lines64\": %zu}\n", bytes, bytes, bytes / 12 * 12, bytes / 4,
           ((size_t)1 << 26) * 4, ((size_t)1 << 26) * 64);
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}
