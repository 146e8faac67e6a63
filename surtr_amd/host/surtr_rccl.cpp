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
    // A rank whose event failed must not leave the others waiting in a collective: its size word says so, and every rank
    // throws after the size exchange.
    surtr_counts c;
    const int rc0 = surtr_event_counts(engine.Raw(), &c);
    const uint64_t kFailed = ~(uint64_t)0;
    const uint64_t mine = rc0 ? kFailed : (uint64_t)surtr_event_blob_bytes(&c);
    // 1. sizes
    DevBuf d_sizes((size_t)world * 8), d_mine(8);
    hip_ck(hipMemcpyAsync(d_mine.p, &mine, 8, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    nccl_ck(ncclAllGather(d_mine.p, d_sizes.p, 1, ncclUint64, comm, stream), "ncclAllGather(sizes)");
    std::vector<uint64_t> sizes(world);
    hip_ck(hipMemcpyAsync(sizes.data(), d_sizes.p, (size_t)world * 8, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ck(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (rc0) throw Error(rc0, "surtr_event_counts");
    for (int r = 0; r < world; ++r) if (sizes[r] == kFailed) throw Error(SURTR_E_STATE, "the event of rank " + std::to_string(r) + " failed");
    uint64_t stride = 0;
    for (uint64_t s : sizes) stride = s > stride ? s : stride;
    stride = (stride + 255) & ~(uint64_t)255;
    // 2. blobs, padded to the largest.  k_pack runs on the ENGINE's stream (surtr_get_stream), the collective on the caller's:
    // the padding is cleared on the engine's stream before the pack, and the caller's stream waits for the pack.
    void* es = nullptr;
    int rc = surtr_get_stream(engine.Raw(), &es);
    if (rc) throw Error(rc, "surtr_get_stream");
    hipStream_t estream = (hipStream_t)es;
    DevBuf d_blob(stride), d_all(stride * (size_t)world);
    // From here to the collective nothing throws: every rank enters the all-gather whatever happened to it, and errors are raised
    // after it.  Without an event the pack is ordered before the collective by a synchronisation of the engine's stream.
    const hipError_t e0 = hipMemsetAsync(d_blob.p, 0, stride, estream);
    rc = surtr_event_pack_dev(engine.Raw(), d_blob.p, stride);
    hipEvent_t packed = nullptr;
    hipError_t e1 = hipEventCreateWithFlags(&packed, hipEventDisableTiming);
    if (e1 == hipSuccess) e1 = hipEventRecord(packed, estream);
    hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream, packed, 0) : hipStreamSynchronize(estream);
    ncclResult_t nr = ncclAllGather(d_blob.p, d_all.p, stride, ncclUint8, comm, stream);
    GatheredFragments out;
    out.stride = stride; out.host.resize(stride * (size_t)world); out.counts.resize(world);
    hipError_t e3 = hipMemcpyAsync(out.host.data(), d_all.p, out.host.size(), hipMemcpyDeviceToHost, stream);
    hipError_t e4 = hipStreamSynchronize(stream);
    if (packed) (void)hipEventDestroy(packed);
    if (rc) throw Error(rc, "surtr_event_pack_dev");
    hip_ck(e0, "hipMemsetAsync"); hip_ck(e1, "hipEventCreate / hipEventRecord"); hip_ck(e2, "hipStreamWaitEvent"); nccl_ck(nr, "ncclAllGather(blobs)"); hip_ck(e3, "hipMemcpyAsync"); hip_ck(e4, "hipStreamSynchronize");
    for (int r = 0; r < world; ++r)
    {
        rc = surtr_blob_unpack_host(out.host.data() + (size_t)r * stride, sizes[r], &out.counts[r], nullptr);
        if (rc) throw Error(rc, "surtr_blob_unpack_host");
        if (out.counts[r].status) throw Error((int)out.counts[r].status, "device status of rank " + std::to_string(r));
    }
    return out;
}

} // namespace surtr
