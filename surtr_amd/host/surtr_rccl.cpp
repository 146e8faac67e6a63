// surtr_rccl.cpp -- see surtr_rccl.hpp.
#include "surtr_rccl.hpp"

#include <cstring>
#include <string>

namespace surtr {

namespace {
void hip_ck(hipError_t e, const char* what) { if (e != hipSuccess) throw Error(SURTR_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
void nccl_ck(ncclResult_t r, const char* what) { if (r != ncclSuccess) throw Error(SURTR_E_HIP, std::string(what) + ": " + ncclGetErrorString(r)); }
struct DevBuf
{
    void* p = nullptr;
    explicit DevBuf(size_t n) { hip_ck(hipMalloc(&p, n ? n : 16), "hipMalloc"); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
} // namespace

GatheredFragments AllGatherFragments(FractureEngine& engine, ncclComm_t comm, int world, hipStream_t stream)
{
    surtr_counts c;
    int rc = surtr_event_counts(engine.Raw(), &c);
    if (rc) throw Error(rc, "surtr_event_counts");
    const uint64_t mine = surtr_event_blob_bytes(&c);
    // 1. sizes
    DevBuf d_sizes((size_t)world * 8), d_mine(8);
    hip_ck(hipMemcpyAsync(d_mine.p, &mine, 8, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    nccl_ck(ncclAllGather(d_mine.p, d_sizes.p, 1, ncclUint64, comm, stream), "ncclAllGather(sizes)");
    std::vector<uint64_t> sizes(world);
    hip_ck(hipMemcpyAsync(sizes.data(), d_sizes.p, (size_t)world * 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ck(hipStreamSynchronize(stream), "hipStreamSynchronize");
    uint64_t stride = 0;
    for (uint64_t s : sizes) stride = s > stride ? s : stride;
    stride = (stride + 255) & ~(uint64_t)255;
    // 2. blobs, padded to the largest
    DevBuf d_blob(stride), d_all(stride * (size_t)world);
    hip_ck(hipMemsetAsync(d_blob.p, 0, stride, stream), "hipMemsetAsync");
    rc = surtr_event_pack_dev(engine.Raw(), d_blob.p, stride);
    if (rc) throw Error(rc, "surtr_event_pack_dev");
    nccl_ck(ncclAllGather(d_blob.p, d_all.p, stride, ncclUint8, comm, stream), "ncclAllGather(blobs)");
    GatheredFragments out;
    out.stride = stride; out.host.resize(stride * (size_t)world); out.counts.resize(world);
    hip_ck(hipMemcpyAsync(out.host.data(), d_all.p, out.host.size(), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ck(hipStreamSynchronize(stream), "hipStreamSynchronize");
    for (int r = 0; r < world; ++r)
    {
        rc = surtr_blob_unpack_host(out.host.data() + (size_t)r * stride, sizes[r], &out.counts[r], nullptr);
        if (rc) throw Error(rc, "surtr_blob_unpack_host");
        if (out.counts[r].status) throw Error((int)out.counts[r].status, "device status of rank " + std::to_string(r));
    }
    return out;
}

} // namespace surtr
