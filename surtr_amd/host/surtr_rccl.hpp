// surtr_rccl.hpp -- the one exchange step of a sharded fracture event in C++: every rank packs its fragments into one device
// blob (surtr_event_pack_dev) and ONE all-gather over RCCL / xGMI gives every rank all of them (SURVEY.md section 8e;
// the reference's per-cell fan-out returns in cell order, Src/Surtr.cpp:2129-2146, so rank order = cell order).
//
// One process per GPU.  Sizes first (ncclAllGather of one uint64 per rank), then the blobs padded to the largest
// (ncclAllGather of bytes); both on the caller's stream, so the kernels of the next event can overlap on another one.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <vector>

#include "surtr_host.hpp"

namespace surtr {

// Contiguous block of `n` units for rank r of `world`: [floor(r n / G), floor((r+1) n / G)).  Cells of one event, or the
// fragment-major (fragment, cell) pairs of a recursive refracture.
inline void RankBlock(uint32_t rank, uint32_t world, uint32_t n, uint32_t& begin, uint32_t& end)
{
    begin = (uint32_t)(((uint64_t)rank * n) / world); end = (uint32_t)(((uint64_t)(rank + 1) * n) / world);
}

struct GatheredFragments
{
    std::vector<surtr_counts> counts;      // per rank, rank order = cell order
    std::vector<char> host;                // all blobs, rank r at r * stride
    size_t stride = 0;
    uint64_t total_fragments() const { uint64_t t = 0; for (const auto& c : counts) t += c.n_frag; return t; }
};

// All-gather of the last event's fragments of `engine` over `comm`; every rank gets every blob.  Throws surtr::Error.
GatheredFragments AllGatherFragments(FractureEngine& engine, ncclComm_t comm, int world, hipStream_t stream);

} // namespace surtr
