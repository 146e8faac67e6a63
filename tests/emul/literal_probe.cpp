// literal_probe.cpp -- test infrastructure: literal_clip.h (the product's last-resort single-lane clipper) behind a C entry
// point, compiled for the CPU through hip_emul.h, so that tests can run it on solids the parallel clipper handles itself and
// compare with the oracle.
#include "../../include/surtr_hip.h"
#include "../../surtr_amd/csrc/literal_clip.h"
#include <vector>
#include <cstring>

extern "C" int literal_probe(uint32_t nv, const float* pos, const uint32_t* off, const int32_t* nbr, uint32_t F, const float* planes,
                             uint32_t cap_v, float* opos, uint32_t* ooff, int32_t* onbr, uint32_t* n_out)
{
    using namespace surtr;
    std::vector<uint32_t> llen(nv), loff(off, off + nv);
    for (uint32_t v = 0; v < nv; ++v) llen[v] = off[v + 1] - off[v];
    SolidIn in{};
    in.pos = pos; in.loff = loff.data(); in.llen = llen.data(); in.nbr = nbr; in.nv = nv;
    std::vector<float> p(3 * (size_t)cap_v);
    std::vector<uint32_t> len(cap_v), slen(cap_v);
    std::vector<int32_t> ring((size_t)cap_v * LIT_STRIDE), snap((size_t)cap_v * LIT_STRIDE), id(cap_v);
    std::vector<int8_t> comp(cap_v);
    LitSolid S{p.data(), len.data(), ring.data(), slen.data(), snap.data(), comp.data(), id.data(), cap_v};
    std::vector<float4> pl(F);
    for (uint32_t k = 0; k < F; ++k) pl[k] = make_float4(planes[4 * k], planes[4 * k + 1], planes[4 * k + 2], planes[4 * k + 3]);
    uint32_t n = 0;
    const int rc = literal_clip(in, F, pl.data(), S, &n);
    if (rc != 0) return rc;
    uint32_t h = 0;
    for (uint32_t v = 0; v < n; ++v)
    {
        std::memcpy(opos + 3 * v, p.data() + 3 * v, 12);
        ooff[v] = h;
        for (uint32_t q = 0; q < len[v]; ++q) onbr[h++] = ring[(size_t)v * LIT_STRIDE + q];
    }
    ooff[n] = h; *n_out = n;
    return 0;
}
