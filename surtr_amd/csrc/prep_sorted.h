// prep_sorted.h -- the pre-pass of k_prep_pairs for a piece with a spatially sorted copy (round 4), and the record image it
// leaves for the record clipper (wave_clip.h).
//
// What the pre-pass has to find (clip_core.h): the BAND of a (cell, piece) pair -- the vertices of the Mesh whose
// incident faces do not all share one first clipping plane fc -- plus, per plane, how many of the dropped vertices are still
// alive (hist) or lie in the plane (zhist).  prepass_select() decides every vertex from positions: a ball test per vertex and,
// where that fails, fc of every neighbour from the neighbour's POSITION (six neighbours x up to fc + 1 planes each).  Here:
//
//   A0  three levels of bounding spheres over the Morton-sorted vertices (512 / 64 / SURTR_SB = 8 vertices, pieces_dev.hip):
//       a sphere on the cut side of plane k and on the kept side of planes 0..k-1 decides its whole range (all dropped with
//       fc = k) with one test; only the children of an undecided sphere are looked at.  What is left is a bit per 8-vertex group:
//       "undecided".
//   P1  fc of every vertex of the undecided groups, ONE plane loop per vertex, stored as a byte by sorted index.
//   P2  the exact test by LOOK-UP: the neighbours of a vertex are listed by sorted index (Pieces::mnbr_s), so a neighbour's fc is
//       its byte -- no position is read, no plane is evaluated twice.  A neighbour u in a DECIDED group needs no look-up at all:
//       the group's sphere holds the ball of u (every vertex of u's faces), so the vertex asking lies in that sphere and has the
//       group's fc itself.
//   E   the band is written ONCE, as what the record clipper streams: 16-byte records + positions, stably sorted by first
//       clipping plane (wc_load's counting sort, moved here: this kernel has every fc in hand).  wc_load -- which read an image
//       in the LDS-topology layout, ranked it and wrote the sorted copy -- is not run for such a pair.
//
// The results are those of prepass_select / prepass_emit + wc_load: same band, same order (ascending vertex index inside a
// bucket = the reference's numbering rule, Src/Poly.cpp:333-357), same counters.  Pairs the record clipper cannot take
// (an in-plane band vertex, a ring of more than seven entries, a band beyond the id map, a cell of more than 64 planes) get the
// old image through prepass_emit, as before.
#pragma once
#include "wave_clip.h"

#define PS_NEVER 0x7Fu        // fc byte: no plane clips the vertex (bit 7 of the byte: it lies in a plane before fc)
#ifndef SURTR_PS_NB
#define SURTR_PS_NB 896u      // 64-vertex blocks of a piece this selection takes (57 344 vertices): the last 128 of the 1 024 entries of
#endif                        // the kernel's per-block table hold one bit per 8-vertex group ("undecided") instead
#ifndef SURTR_SPH_FAN
#define SURTR_SPH_FAN 8u
#endif

namespace surtr {

struct SortedRings
{
    const SRow* row_s; const uint32_t* iperm;
    const float4* bsph2; const float4* bsph3;
};

// first plane that has the whole sphere on its cut side while every earlier plane has it on its kept side; 0xFF: undecided
// (conservative margins: prepass_select, A0)
// k0: the first plane to look at (the enclosing sphere of the level above was wholly on the kept side of the planes before it);
// kstop: the plane the test stopped at -- the first one the sphere is not wholly on the kept side of (F: none).
__device__ __forceinline__ uint32_t ps_sphere_fc(const Shared& sh, const uint32_t F, const float4 sp, const uint32_t k0, uint32_t& kstop)
{
    const float mag = fabsf(sp.x) + fabsf(sp.y) + fabsf(sp.z) + sp.w;
    for (uint32_t k = k0; k < F; ++k)
    {
        const float4 mk = sh.pmar[k];
        const float sk = plane_dist(sh.planes[k], sp.x, sp.y, sp.z);
        const float margin = sp.w * mk.x + mk.y + mk.z * mag;
        kstop = k;
        if (sk > margin) return k;
        if (!(sk < -margin)) return 0xFFu;
    }
    kstop = F;
    return 0xFFu;
}

// Appends the lanes with `on` to a list in global memory (one LDS counter; order inside a wave = lane order).
__device__ __forceinline__ void ps_append(uint32_t* list, uint32_t* counter, bool on, uint32_t value)
{
    const unsigned long long m = __ballot(on);
    if (!m) return;
    const uint32_t l = lane_id();
    uint32_t base = 0;
    if (l == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = lane_bcast(base, (uint32_t)__builtin_ctzll(m));
    if (on) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull))] = value;
}

__device__ __forceinline__ void ps_append16(uint16_t* list, uint32_t* counter, bool on, uint32_t value)
{
    const unsigned long long m = __ballot(on);
    if (!m) return;
    const uint32_t l = lane_id();
    uint32_t base = 0;
    if (l == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = lane_bcast(base, (uint32_t)__builtin_ctzll(m));
    if (on) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull))] = (uint16_t)value;
}
__device__ __forceinline__ void ps_append2(uint2* list, uint32_t* counter, bool on, uint2 value)
{
    const unsigned long long m = __ballot(on);
    if (!m) return;
    const uint32_t l = lane_id();
    uint32_t base = 0;
    if (l == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
    base = lane_bcast(base, (uint32_t)__builtin_ctzll(m));
    if (on) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull))] = value;
}

// Selection.  lbuf: the kernel's LDS table area, 16 * NB bytes (NB = 64-vertex blocks it is sized for).  While the band is
// selected it holds: the first clipping planes of the undecided groups' vertices (bytes, in ascending group order; what does
// not fit goes to vfc_g), then `ubw` words "set bits before this word" and `ubw` words of one bit per group ("undecided").
// On return it holds what the emits read: bmask (one 64-bit word per 64 vertices, first half) and bblk (one pair per 64
// vertices, second half; x: kept vertices before the block, y: their ring entries), built from the kept list.
// vfc_g, needy, und, klist, walks: global scratch (V entries each at most; und: one per group; needy, und: 16-bit sorted indices /
// groups -- what moves between the passes of this kernel is what its counter traffic is made of).
// Requires in.nv < 65535, ceil(in.nv / SURTR_SB) <= 32 * ubw, in.nv <= 64 * NB, the sorted copy.
// Leaves: klist[0 .. n) = sorted index | (fc | 0x80: in a plane before fc) << 16 | ring length (255: more) << 24, in no order
// (the vertex is perm[sorted index]);
// sh.hist / zhist raw (prepass_finish_hist); sh.misc[5] = a kept vertex lies in a plane before its fc; sh.deg7, sh.flagBad.
template <uint32_t NB, uint32_t UBW>
__device__ __attribute__((always_inline)) inline void prepass_select_sorted(const SolidIn in, const SortedRings sr, const uint32_t F, Shared& sh,
                                                                            unsigned char* lbuf, uint8_t* vfc_g, uint16_t* needy, uint16_t* und, uint32_t* klist, uint32_t* walks,
                                                                            uint32_t& n_out, uint32_t& hsum_out)
{
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id(), G = group_size(), nw = group_waves();
    const uint32_t V = in.nv;
    constexpr uint32_t kCap = 16u * NB - 8u * UBW;          // bytes of the byte table that fit
    uint8_t* vfc = lbuf;
    uint32_t* pre = (uint32_t*)(lbuf + kCap);
    uint32_t* ub = pre + UBW;
    unsigned long long* bmask = (unsigned long long*)lbuf;
    uint2* bblk = (uint2*)(lbuf + 8u * NB);
    STAMP_DECL;
    for (uint32_t k = tid; k <= SURTR_MAXF; k += G)
    {
        sh.hist[k] = 0; sh.zhist[k] = 0; sh.nzero[k] = 0;
        if (k < F)
        {
            const float4 pk = sh.planes[k];
            const float n1 = fabsf(pk.x) + fabsf(pk.y) + fabsf(pk.z);
            const float n2 = sqrtf(pk.x * pk.x + pk.y * pk.y + pk.z * pk.z) * 1.0001f;
            sh.pmar[k] = make_float4(n2 * 1.00101f, 1.0e-5f * fabsf(pk.w), 1.0e-5f * n1, 0.f);
        }
    }
    // misc: 1, 2 undecided spheres of levels 3, 2; 3 needy; 4 undecided groups; 5 in-plane kept vertex; 6 kept; 0 face walks
    if (tid == 0) { sh.flagErr = 0; sh.flagBad = 0; sh.deg7 = 0; for (int q = 0; q < 7; ++q) sh.misc[q] = 0; }
    const uint32_t nbV = (V + SURTR_LANES - 1u) >> SURTR_LSH;
    for (uint32_t q = tid; q < UBW; q += G) ub[q] = 0u;
    __syncthreads();
    // ---- A0: spheres, coarse to fine.  Work lists of undecided spheres (16-bit ids) live at the start of the byte table until
    //      A0 is done.  A decided sphere only counts its vertices (all dropped with one fc); an undecided group sets its bit. ----
    const uint32_t nsb = (V + SURTR_SB - 1u) / SURTR_SB;
    const uint32_t nb2 = (nsb + SURTR_SPH_FAN - 1u) / SURTR_SPH_FAN, nb3 = (nb2 + SURTR_SPH_FAN - 1u) / SURTR_SPH_FAN;
    // (entries: sphere | plane its test stopped at << 16 -- the level below starts its own tests there)
    uint32_t* list2 = (uint32_t*)lbuf;                 // undecided level-2 spheres (nb2 of them at most)
    uint32_t* list3 = list2 + nb2 + 8u;                // undecided level-3 spheres
    auto verts_in = [&](uint32_t g0, uint32_t g1) -> uint32_t {      // vertices of groups [g0, g1)
        const uint32_t a = g0 * SURTR_SB, b = g1 * SURTR_SB;
        return (b < V ? b : V) - (a < V ? a : V);
    };
    for (uint32_t s3 = tid; s3 < nb3; s3 += G)
    {
        uint32_t ks = 0;
        const uint32_t f = ps_sphere_fc(sh, F, sr.bsph3[s3], 0u, ks);
        const uint32_t g0 = s3 * SURTR_SPH_FAN * SURTR_SPH_FAN;
        if (f != 0xFFu) atomicAdd(&sh.hist[f], verts_in(g0, g0 + SURTR_SPH_FAN * SURTR_SPH_FAN));
        else list3[atomicAdd(&sh.misc[1], 1u)] = s3 | (ks << 16);
    }
    __syncthreads();
    const uint32_t nU3 = sh.misc[1];
    for (uint32_t t = tid; t < nU3 * SURTR_SPH_FAN; t += G)
    {
        const uint32_t e3 = list3[t / SURTR_SPH_FAN];
        const uint32_t s2 = (e3 & 0xFFFFu) * SURTR_SPH_FAN + t % SURTR_SPH_FAN;
        if (s2 >= nb2) continue;
        uint32_t ks = 0;
        const uint32_t f = ps_sphere_fc(sh, F, sr.bsph2[s2], e3 >> 16, ks);
        if (f != 0xFFu) atomicAdd(&sh.hist[f], verts_in(s2 * SURTR_SPH_FAN, s2 * SURTR_SPH_FAN + SURTR_SPH_FAN));
        else list2[atomicAdd(&sh.misc[2], 1u)] = s2 | (ks << 16);
    }
    __syncthreads();
    const uint32_t nU2 = sh.misc[2];
    for (uint32_t t = tid; t < nU2 * SURTR_SPH_FAN; t += G)
    {
        const uint32_t e2 = list2[t / SURTR_SPH_FAN];
        const uint32_t sb = (e2 & 0xFFFFu) * SURTR_SPH_FAN + t % SURTR_SPH_FAN;
        if (sb >= nsb) continue;
        uint32_t ks = 0;
        const uint32_t f = ps_sphere_fc(sh, F, in.bsph[sb], e2 >> 16, ks);
        if (f != 0xFFu) atomicAdd(&sh.hist[f], verts_in(sb, sb + 1u));
        else atomicOr(&ub[sb >> 5], 1u << (sb & 31u));
    }
    __syncthreads();
    // the undecided groups in ascending (Morton) order, from the bits: und[rank] = group, pre[q] = set bits before word q.  The
    // rank of a group is also where its vertices' bytes sit in the table.
    {
        uint32_t carry = 0;
        for (uint32_t q0 = 0; q0 < UBW; q0 += G)
        {
            const uint32_t q = q0 + tid;
            const uint32_t c = q < UBW ? (uint32_t)__builtin_popcount(ub[q]) : 0u;
            const uint32_t inc = wave_incl_scan2(make_uint2(c, 0u)).x;
            if (l == SURTR_LANES - 1u) sh.wsum[w] = inc;
            __syncthreads();
            uint32_t woff = 0, tot = 0;
            for (uint32_t x = 0; x < nw; ++x) { const uint32_t a = sh.wsum[x]; if (x < w) woff += a; tot += a; }
            if (q < UBW)
            {
                uint32_t at = carry + woff + inc - c;
                pre[q] = at;
                for (uint32_t m = ub[q]; m; m &= m - 1u) und[at++] = (uint16_t)(32u * q + (uint32_t)__builtin_ctz(m));
            }
            carry += tot;
            __syncthreads();
        }
        if (tid == 0) sh.misc[4] = carry;
    }
    __syncthreads();
    const uint32_t nUnd = sh.misc[4];
#ifdef SURTR_STAMP
    if (tid == 0 && V > 10000u) { atomicAdd(&g_stamp[45], (unsigned long long)nUnd); atomicAdd(&g_stamp[46], (unsigned long long)nsb); }
#endif
    STAMP(44);
    // ---- P1: first clipping plane of every vertex of the undecided groups (ComparePlanePoint, Src/Poly.cpp:716-723: clipped iff
    //      s >= 1e-10f, in the plane iff |s| < 1e-10f -- side_of() spelled without its branches), stored as a byte at (rank of
    //      the group) * SB + place; and the ball test of prepass_select: a clipped vertex whose ball (every vertex of its incident
    //      faces) stays strictly on its side of every plane up to its first clipping plane is dropped here, the others go to P2 ----
    constexpr uint32_t GPW = SURTR_LANES / SURTR_SB;      // groups per wave-load
    const uint32_t nWork = (nUnd + GPW - 1u) / GPW;
    for (uint32_t wb = w; wb < nWork; wb += nw)
    {
        const uint32_t sub = wb * GPW + l / SURTR_SB;
        const bool okg = sub < nUnd;
        const uint32_t i = (okg ? (uint32_t)und[sub] : 0u) * SURTR_SB + (l % SURTR_SB);
        const bool valid = okg && i < V;
        const float4 pr = in.posr_s[valid ? i : 0u];
        const float mag = fabsf(pr.x) + fabsf(pr.y) + fabsf(pr.z);
        // (starting the loop at the plane the group's sphere test stopped at was measured: +-0 -- the band lies ON the early planes)
        uint32_t f = PS_NEVER, z = 0u;
        bool done = !valid, clear = true;
        for (uint32_t k = 0; k < F; ++k)
        {
            if (__all(done)) break;
            const float4 mk = sh.pmar[k];
            const float s = plane_dist(sh.planes[k], pr.x, pr.y, pr.z);
            const bool cut = s >= 1.0e-10f, zz = fabsf(s) < 1.0e-10f, live = !done;
            const bool far = fabsf(s) > __builtin_fmaf(pr.w, mk.x, __builtin_fmaf(mk.z, mag, mk.y));      // (a bound with slack: contraction is welcome)
            clear = clear & (far | !live);
            z |= (live & zz) ? 0x80u : 0u;
            f = (live & cut) ? k : f;
            done = done | cut;
        }
        const uint32_t at = sub * SURTR_SB + (l % SURTR_SB);
        if (valid) { if (at < kCap) vfc[at] = (uint8_t)(f | z); else vfc_g[at] = (uint8_t)(f | z); }
        const bool never = valid && f == PS_NEVER;
        const bool drop = valid && !never && clear;          // (|s| > margin at every plane up to fc: in no plane either)
        const bool need = valid && !never && !clear;
        wave_hist_add(sh.hist, f, drop);
        ps_append16(needy, &sh.misc[3], need, i);
        if (__ballot(never))
        {
            // never clipped: kept whatever its neighbours are
            const uint32_t v = never ? in.perm[i] : 0u;
            const uint32_t deg = never ? in.llen[v] : 0u;
            if (never && z) sh.misc[5] = 1u;
            ps_append(klist, &sh.misc[6], never, i | ((f | z) << 16) | ((deg < 255u ? deg : 255u) << 24));
        }
    }
    __syncthreads();
    STAMP(0);
    // ---- P2, densely over the vertices the ball test left: dropped iff all neighbours share the first clipping plane.  A
    //      neighbour in a decided group does (see the head of this file); one in an undecided group has its byte.  Vertices
    //      with a face that is no triangle, or more than seven neighbours, go on to the exact test with face walks (prepass_exact). ----
    const uint32_t nNeedy = sh.misc[3];
#ifdef SURTR_STAMP
    if (tid == 0 && V > 10000u) atomicAdd(&g_stamp[47], (unsigned long long)nNeedy);
#endif
    for (uint32_t t0 = w << SURTR_LSH; t0 < nNeedy; t0 += G)
    {
        const uint32_t t = t0 + l;
        const bool valid = t < nNeedy;
        const uint32_t i = valid ? (uint32_t)needy[t] : 0u;
        const SRow row = sr.row_s[i];
        // its own byte: at (rank of its group) * SB + place, like its neighbours' below
        uint32_t byte;
        {
            const uint32_t g = i / SURTR_SB, word = ub[g >> 5], bit = g & 31u;
            const uint32_t at = (pre[g >> 5] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u))) * SURTR_SB + i % SURTR_SB;
            byte = (valid && at < kCap) ? (uint32_t)vfc[valid ? at : 0u] : 0u;
            if (valid && at >= kCap) byte = vfc_g[at];
        }
        const uint32_t f = byte & 0x7Fu;
        const uint32_t hd = row.w[0] & 0xFFFFu;
        const uint32_t deg = valid ? (hd & 7u) : 0u, notri = (hd >> 7) & 1u, big = (hd >> 6) & 1u;
        // the first clipping plane of every neighbour in an undecided group: its byte sits at (rank of the group) * SB + place
        uint32_t uq[7], on[7], fq[7];
#pragma unroll
        for (int q = 0; q < 7; ++q)
        {
            const uint32_t x = row.w[(q + 1) >> 1];
            const uint32_t u = (uint32_t)q < deg ? (((q + 1) & 1) ? (x >> 16) : (x & 0xFFFFu)) : 0u;
            const uint32_t g = u / SURTR_SB, word = ub[g >> 5], bit = g & 31u;
            on[q] = ((uint32_t)q < deg ? 1u : 0u) & ((word >> bit) & 1u);
            uq[q] = (pre[g >> 5] + (uint32_t)__builtin_popcount(word & ((1u << bit) - 1u))) * SURTR_SB + u % SURTR_SB;
        }
#pragma unroll
        for (int q = 0; q < 7; ++q) fq[q] = (on[q] && uq[q] < kCap) ? ((uint32_t)vfc[uq[q]] & 0x7Fu) : f;
#pragma unroll
        for (int q = 0; q < 7; ++q) if (on[q] && uq[q] >= kCap) fq[q] = (uint32_t)vfc_g[uq[q]] & 0x7Fu;      // (a band beyond the table: rare)
        bool differ = false;
#pragma unroll
        for (int q = 0; q < 7; ++q) differ = differ | (fq[q] != f);
        // all neighbours agree but the faces are larger than the 1-ring, or the ring is not in the row: the exact test with face walks
        const bool walk = valid && (big != 0u || (notri != 0u && !differ));
        const bool keep = valid && differ && !walk, drop = valid && !differ && !walk;
        if (keep && (byte & 0x80u)) sh.misc[5] = 1u;
        ps_append(klist, &sh.misc[6], keep, i | (byte << 16) | (deg << 24));
        if (__ballot(walk))
        {
            const uint32_t v = walk ? in.perm[i] : 0u;
            ps_append(walks, &sh.misc[0], walk, v | (f << 24));
        }
        wave_hist_add(sh.hist, f, drop);
        if (drop && (byte & 0x80u))
        {
            // in a plane before its first clipping plane while alive: it is no "kept" vertex there (rare: |s| < 1e-10)
            const float4 pr = in.posr_s[i];
            for (uint32_t k = 0; k < f; ++k)
                if (side_of(plane_dist(sh.planes[k], pr.x, pr.y, pr.z)) == 0) atomicAdd(&sh.zhist[k], 1u);
        }
    }
    __syncthreads();
    if (sh.misc[0] != 0u) prepass_exact<4>(in, F, sh, (unsigned long long*)nullptr, (uint2*)nullptr, walks, sh.misc[0], klist, &sh.misc[6], sr.iperm);
    __syncthreads();
    STAMP(1);
    // ---- the masks the emits number the band with, from the kept list (the table area is free now) ----
    const uint32_t n = sh.misc[6];
    for (uint32_t b = tid; b < nbV; b += G) { bmask[b] = 0ull; bblk[b] = make_uint2(0u, 0u); }
    __syncthreads();
    for (uint32_t t = tid; t < n; t += G)
    {
        const uint32_t ke = klist[t];
        const uint32_t v = in.perm[ke & 0xFFFFu], deg = ke >> 24;
        atomicOr(&bmask[v >> SURTR_LSH], 1ull << (v & (SURTR_LANES - 1u)));
        uint32_t d = deg;
        if (deg >= 255u) d = in.llen[v];
        prepass_keep_deg(bblk, sh, v, d);
    }
    __syncthreads();
    for (uint32_t b = tid; b < nbV; b += G) bblk[b].x = (uint32_t)__builtin_popcountll(bmask[b]);
    __syncthreads();
    uint32_t n2 = 0, hsum = 0;
    scan_block_array(nbV, bblk, sh, n2, hsum);
    __syncthreads();
    STAMP(3);
    n_out = n2; hsum_out = hsum;
}

// index of vertex u in the band (ascending vertex index), or `absent`
__device__ __forceinline__ uint32_t ps_newid(const unsigned long long* bmask, const uint2* bblk, uint32_t u, uint32_t absent)
{
    const unsigned long long m = bmask[u >> SURTR_LSH];
    const uint32_t bit = u & (SURTR_LANES - 1u);
    if (!((m >> bit) & 1ull)) return absent;
    return bblk[u >> SURTR_LSH].x + (uint32_t)__builtin_popcountll(m & ((1ull << bit) - 1ull));
}

// The old image (ImgLayout: the reduced solid in the layout of the LDS topology) from the kept list: what prepass_emit writes,
// without its sweep over every 64-vertex block of the piece and without its plane loop (the first clipping planes are in the list;
// only a vertex that lies in a plane before its fc is evaluated again, for sh.nzero).  scan: global, ceil(n / 64) + 2 pairs.
__device__ __attribute__((always_inline)) inline void prepass_emit_klist(const SolidIn in, const uint32_t F, Shared& sh, Topo<InLds>& T, const unsigned long long* bmask,
                                                                         const uint2* bblk, const uint32_t* klist, uint32_t* orig, uint2* scan, const uint32_t n, const uint32_t hsum, uint32_t& maxb_out)
{
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id(), G = group_size(), nw = group_waves();
    STAMP_DECL;
    // band vertices per first clipping plane (sh.pw: the sphere margins are done with): the largest bucket is what the record
    // clipper's LDS need at its worst plane follows
    for (uint32_t k = tid; k <= F; k += G) sh.pw[k] = 0u;
    __syncthreads();
    for (uint32_t t = tid; t < n; t += G)
    {
        const uint32_t ke = klist[t], si = ke & 0xFFFFu;
        const uint32_t v = in.perm[si], byte = (ke >> 16) & 0xFFu, f = byte & 0x7Fu;
        const uint32_t id = ps_newid(bmask, bblk, v, 0u);
        orig[id] = v | (si << 16);          // (vertex, sorted index: both below 65 535 here)
        T.llen[id] = (uint8_t)((ke >> 24) < 255u ? (ke >> 24) : in.llen[v]);
        T.fc[id] = (uint8_t)(f == PS_NEVER ? SURTR_NEVER : f);
        if (f != PS_NEVER) { atomicOr(&sh.cutmask[f >> 5], 1u << (f & 31u)); if (f < F) atomicAdd(&sh.pw[f], 1u); }
        if (byte & 0x80u)
        {
            const float4 pr = in.posr_s[si];
            const float px = pr.x, py = pr.y, pz = pr.z;
            for (uint32_t k = 0; k < F; ++k)
            {
                const int c = side_of(plane_dist(sh.planes[k], px, py, pz));
                if (c < 0) break;
                if (c == 0) atomicAdd(&sh.nzero[k], 1u);
            }
        }
    }
    __syncthreads();
    // ring offsets: exclusive scan of the lengths in band order
    uint32_t tot = 0, dum = 0;
    scan_blocks(n, scan, sh, [&](uint32_t id) -> uint2 { return make_uint2((uint32_t)T.llen[id], 0u); }, tot, dum);
    const uint32_t nb = (n + SURTR_LANES - 1u) >> SURTR_LSH;
    for (uint32_t b = w; b < nb; b += nw)
    {
        const uint32_t id = (b << SURTR_LSH) + l;
        const uint32_t len = id < n ? (uint32_t)T.llen[id] : 0u;
        const uint2 ex = wave_excl2(make_uint2(len, 0u));
        if (id < n) T.loff[id] = (uint16_t)(scan[b].x + ex.x);
    }
    __syncthreads();
    for (uint32_t id = tid; id < n; id += G)
    {
        const uint32_t v = orig[id] & 0xFFFFu;
        const float4 pr = in.posr_s[orig[id] >> 16];
        T.pos[3 * id] = pr.x; T.pos[3 * id + 1] = pr.y; T.pos[3 * id + 2] = pr.z;
        const uint32_t deg = T.llen[id];
        const int32_t* r = in.nbr + in.loff[v];
        uint16_t* d = T.ring + T.loff[id];
        for (uint32_t j0 = 0; j0 < deg; j0 += 8)
        {
            int32_t u[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = (j0 + q < deg) ? r[j0 + q] : -1;
#pragma unroll
            for (int q = 0; q < 8; ++q) if (u[q] >= 0) d[j0 + q] = (uint16_t)ps_newid(bmask, bblk, (uint32_t)u[q], InLds::SENT);
        }
    }
    STAMP(2);
    __syncthreads();
    uint32_t maxb = 0;
    for (uint32_t k = 0; k < F; ++k) maxb = sh.pw[k] > maxb ? sh.pw[k] : maxb;
    maxb_out = maxb;
    T.nS = n; T.nLive = n; T.hUsed = hsum;
}

// Byte offsets of the sections of a record image (IMG_REC): hist / zhist (F words each), bucket starts (F + 2), then the sorted
// 16-byte records and the positions (x, y, z, band index) of the n band vertices.
struct RecLayout { uint32_t hist, zhist, bst, grec, gpos, total; };
__host__ __device__ static inline RecLayout rec_layout(uint32_t F, uint32_t n)
{
    auto up = [](uint32_t b) { return (b + 15u) & ~15u; };
    const uint32_t nn = (n + 15u) & ~15u;
    RecLayout L;
    L.hist = 0; L.zhist = up(4u * F); L.bst = L.zhist + up(4u * F); L.grec = L.bst + up(4u * (F + 2u)); L.gpos = L.grec + 16u * nn;
    L.total = L.gpos + 16u * nn;
    return L;
}

// Emits the band as a record image from the kept list.  bmask / bblk as prepass_select_sorted left them; orig, fcb, sid16: global
// scratch by band index (vertex | sorted index << 16, first clipping plane, sorted id); cnt: >= 5 * (WC_MAXF + 2) words of LDS.  The caller has checked that no band
// vertex has more than seven ring entries (sh.deg7), F <= WC_MAXF and n < WC_MAXN.
// ncut_out: planes that are the first clipping plane of some band vertex (the cost estimate of the clip); maxb_out: the largest
// number of band vertices one plane clips (what the record clipper's LDS need at its worst plane follows).
__device__ __attribute__((always_inline)) inline void prepass_emit_records(const SolidIn in, const uint32_t F, Shared& sh, const unsigned long long* bmask, const uint2* bblk,
                                                                           const uint32_t* klist, uint32_t* orig, uint8_t* fcb, uint16_t* sid16, uint32_t* cnt,
                                                                           char* img, const uint32_t n, uint32_t& ncut_out, uint32_t& maxb_out)
{
    const uint32_t tid = threadIdx.x, l = lane_id(), w = wave_id(), G = group_size(), nw = group_waves();
    const RecLayout lay = rec_layout(F, n);
    STAMP_DECL;
    // ---- band order = ascending vertex index: orig[id], first clipping plane of id ----
    for (uint32_t t = tid; t < n; t += G)
    {
        const uint32_t ke = klist[t], si = ke & 0xFFFFu;
        const uint32_t v = in.perm[si];
        const uint32_t id = ps_newid(bmask, bblk, v, 0u);
        orig[id] = v | (si << 16);          // (vertex, sorted index: both below 65 535 here)
        fcb[id] = (uint8_t)((ke >> 16) & 0x7Fu);
    }
    // sort waves: at most four take part in the ranking (contiguous ranges of 64-vertex blocks of the band)
    const uint32_t nsw = nw < 4u ? nw : 4u;
    uint32_t* wcnt = cnt;                               // [nsw][WC_MAXF + 2]
    uint32_t* bst = cnt + 4u * (WC_MAXF + 2u);          // [WC_MAXF + 2]
    for (uint32_t k = tid; k < 4u * (WC_MAXF + 2u); k += G) wcnt[k] = 0u;
    __syncthreads();
    // ---- stable counting sort by first clipping plane (bucket F: never clipped): wc_load's two sweeps ----
    const unsigned long long lt = (1ull << l) - 1ull;
    const uint32_t nb = (n + SURTR_LANES - 1u) >> SURTR_LSH, nbw = (nb + nsw - 1u) / nsw;
    const uint32_t bb0 = w * nbw, bb1 = w < nsw ? (bb0 + nbw < nb ? bb0 + nbw : nb) : bb0;
    for (int pass = 0; pass < 2; ++pass)
    {
        for (uint32_t bb = bb0; bb < bb1; ++bb)
        {
            const uint32_t id = (bb << SURTR_LSH) + l;
            const bool valid = id < n;
            const uint32_t f = valid ? (uint32_t)fcb[id] : 0u;
            const uint32_t bk = (f == PS_NEVER || f > F) ? F : f;
            unsigned long long todo = __ballot(valid);
            uint32_t s = 0;
            while (todo)
            {
                const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
                const uint32_t k0 = lane_bcast(bk, leader);
                const unsigned long long same = __ballot(valid && bk == k0);
                uint32_t base = 0;
                if (l == leader) { base = wcnt[w * (WC_MAXF + 2u) + k0]; wcnt[w * (WC_MAXF + 2u) + k0] = base + wc_popc64(same); }
                base = lane_bcast(base, leader);
                if (valid && bk == k0) s = base + wc_popc64(same & lt);
                todo &= ~same;
            }
            if (valid && pass == 1) sid16[id] = (uint16_t)s;
        }
        if (pass == 1) break;
        __syncthreads();
        if (w == 0u)
        {
            uint32_t carry = 0;
            for (uint32_t k0 = 0; k0 <= F; k0 += SURTR_LANES)
            {
                const uint32_t k = k0 + l;
                uint32_t c = 0;
                if (k <= F) for (uint32_t q = 0; q < nsw; ++q) c += wcnt[q * (WC_MAXF + 2u) + k];
                const uint32_t inc = wave_incl_scan2(make_uint2(c, 0u)).x;
                if (k <= F)
                {
                    uint32_t at = carry + inc - c;
                    bst[k] = at;
                    for (uint32_t q = 0; q < nsw; ++q) { const uint32_t cq = wcnt[q * (WC_MAXF + 2u) + k]; wcnt[q * (WC_MAXF + 2u) + k] = at; at += cq; }
                }
                carry += lane_bcast(inc, SURTR_LANES - 1u);
            }
            if (l == 0u) bst[F + 1u] = carry;
        }
        __syncthreads();
    }
    __syncthreads();
    STAMP(2);
    // ---- records: ring entries as sorted ids (a dropped neighbour: WC_SENT), tail = fc | length << 8; positions ----
    WcW4* grec = (WcW4*)(img + lay.grec);
    float4* gpos = (float4*)(img + lay.gpos);
    for (uint32_t id = tid; id < n; id += G)
    {
        const uint32_t vi = orig[id], v = vi & 0xFFFFu;
        const uint32_t f = fcb[id];
        const uint32_t len = in.llen[v];
        const int32_t* r = in.nbr + in.loff[v];
        const uint32_t s = sid16[id];
        const float4 pr = in.posr_s[vi >> 16];
        int32_t u[7]; uint32_t e[7];
#pragma unroll
        for (uint32_t q = 0; q < 7u; ++q) u[q] = r[q < len ? q : 0u];
#pragma unroll
        for (uint32_t q = 0; q < 7u; ++q) e[q] = ps_newid(bmask, bblk, (uint32_t)u[q], 0xFFFFFFFFu);
#pragma unroll
        for (uint32_t q = 0; q < 7u; ++q) e[q] = q >= len ? WC_NONE : (e[q] == 0xFFFFFFFFu ? WC_SENT : (uint32_t)sid16[e[q]]);
        WcW4 wr;
        wr.a = e[0] | (e[1] << 16); wr.b = e[2] | (e[3] << 16); wr.c = e[4] | (e[5] << 16);
        wr.d = e[6] | (((f == PS_NEVER ? SURTR_NEVER : f) | (len << 8)) << 16);
        grec[s] = wr;
        gpos[s] = make_float4(pr.x, pr.y, pr.z, __uint_as_float(id));
    }
    prepass_finish_hist(F, sh);
    uint32_t* hs = (uint32_t*)(img + lay.hist); uint32_t* zs = (uint32_t*)(img + lay.zhist); uint32_t* bs = (uint32_t*)(img + lay.bst);
    for (uint32_t k = tid; k < F; k += G) { hs[k] = sh.hist[k]; zs[k] = sh.zhist[k]; }
    for (uint32_t k = tid; k <= F + 1u; k += G) bs[k] = bst[k];
    uint32_t ncut = 0, maxb = 0;
    for (uint32_t k = 0; k < F; ++k) { const uint32_t c = bst[k + 1u] - bst[k]; ncut += c != 0u ? 1u : 0u; maxb = c > maxb ? c : maxb; }
    ncut_out = ncut; maxb_out = maxb;
    STAMP(74);
}

// The record clipper's state for a pair whose band k_prep_pairs left as a record image: what wc_load sets up, without the sort and
// without a copy -- the records and positions are used, and patched, where they are (a copy into the workgroup's scratch slot
// was measured: the same 1.39 ms for the kernel, 60 KB more written per pair).
template <class LT>
__device__ __attribute__((always_inline)) inline int wc_attach(LT& W, const uint32_t* hist, const uint32_t* zhist, const uint32_t* bst, const uint32_t F, const uint32_t n,
                                                               unsigned long long& zmask, WcCtr& ctr, uint32_t* __restrict__ why)
{
    const uint32_t tid = threadIdx.x;
    ctr.ac = 0u; ctr.sc = 0u;
    zmask = 0ull;
    if (F > WC_MAXF || n == 0u || n >= WC_MAXN) WC_RET(1);
    WSTAMP_DECL;
    for (uint32_t k = tid; k < F; k += group_size()) { W.hist[k] = hist[k]; W.zhist[k] = zhist[k]; }
    for (uint32_t k = tid; k <= F + 1u; k += group_size()) W.bst[k] = bst[k];
    if (tid == 0u) { W.zm[0] = 0u; W.zm[1] = 0u; for (int q = 0; q < 6; ++q) (&W.fl[0][0])[q] = 0u; }
    __syncthreads();
    WSTAMP(1);
    WCOUNT(16, 1); WCOUNT(17, n);
    return 0;
}

} // namespace surtr
