// surtr_ctx.h -- records shared by the translation units of libsurtr_hip.so: what the kernels of one event leave in HBM
// (PairRec, FragRec, Arena), the resident pieces, the scratch pools, and the host-side context behind the C ABI.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/surtr_hip.h"
#include "clip_core.h"

using namespace surtr;

// ------------------------------------------------------------------ records
struct PairRec
{
    uint32_t cv_off, cv_n, ch_off, ch_n;   // clipped Convex in the arena
    uint32_t mv_off, mv_n, mh_off, mh_n;   // clipped Mesh (all islands, island-major)
    uint32_t ni, isl_off;                  // islands and where their (nv, nh) records start
    uint32_t status;
    // reduced Mesh left in HBM by k_prep_pairs: img_fmt = IMG_*, offset in 16-byte units, vertices, ring entries,
    // position slots reserved (>= the capacity of the topology that will clip it: positions are then used in place)
    uint32_t img_fmt, img_off, img_n, img_h, img_pc;
    // 1: the clip of the Convex ended in an inconsistent solid (degenerate input, where the reference produces an invalid
    // polyhedron and carries on).  The pair goes on like the reference's: if nothing is left of the Mesh it yields no fragment
    // and the event is fine; a fragment that would carry the invalid Convex fails the event with SURTR_E_TOPOLOGY.
    uint32_t cv_bad;
};

enum { IMG_NONE = 0,      // no image: k_clip_pairs runs the pre-pass itself
       IMG_NARROW = 1,    // 16-bit image, loads straight into the LDS topology
       IMG_WIDE = 2,      // the reduced solid does not fit the LDS topology: k_clip_pairs goes to global scratch directly
       IMG_EMPTY = 3,     // nothing of the Mesh is left
       IMG_REC = 4 };     // the band as sorted 16-byte records + positions (prep_sorted.h): the record clipper streams it as it is

// Byte offsets of the sections of one image (all 16-byte aligned): hist/zhist/nzero (F words each), the keep mask
// (one word per 64 input vertices), then the reduced solid in the LDS layout, then its positions.
struct ImgLayout { uint32_t hist, zhist, nzero, mask, loff, llen, comp, ring, pos, total; };
__host__ __device__ static inline ImgLayout img_layout(uint32_t F, uint32_t nbV, uint32_t n, uint32_t hsum, uint32_t posCap = 0)
{
    auto up = [](uint32_t b) { return (b + 15u) & ~15u; };
    ImgLayout L;
    L.hist = 0; L.zhist = up(4u * F); L.nzero = L.zhist + up(4u * F); L.mask = L.nzero + up(4u * F); L.loff = L.mask + up(8u * nbV);
    L.llen = L.loff + up(2u * n); L.comp = L.llen + up(n); L.ring = L.comp + up(n); L.pos = L.ring + up(2u * hsum);
    L.total = L.pos + up(12u * (n > posCap ? n : posCap));      // positions last: room for the cut points of the clip
    return L;
}

struct ImgArena { char* base; uint32_t cap16; };      // capacity in 16-byte units; cursor = Arena::cursors[10]

// Per-workgroup scratch of k_prep_pairs: work lists of the pre-pass and (for very large solids) its masks.
struct PrepPool { char* base; size_t per_wg; uint32_t VMAX; };
static size_t prep_bytes_per_wg(uint32_t VMAX)
{
    auto r = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 2 * r((size_t)VMAX * 4) + r((size_t)(VMAX / SURTR_SB + 2) * 4) + 2 * r((size_t)(VMAX / SURTR_LANES + 2) * 8) +
           2 * r((size_t)VMAX + 64) +     // + first clipping planes of the undecided groups' vertices / by band index (prep_sorted.h)
           2 * r((size_t)VMAX * 4) +      // + kept list, face-walk list
           r((size_t)VMAX * 2 + 64);      // + sorted id by band index (record emit)
}

struct FragRec
{
    int32_t cell, piece, island;
    uint32_t mv_off, mv_n, mh_off, mh_n;
    uint32_t cv_off, cv_n, ch_off, ch_n;
    uint32_t idx_off, idx_n;
    // output bases (filled by k_out_scan)
    uint32_t o_mv, o_mh, o_cv, o_ch, o_idx;
};

struct Arena
{
    float* pos; uint32_t* loff; uint32_t* llen; int32_t* nbr; uint32_t* idx;
    uint2* isl;
    uint32_t capV, capH, capI, capIsl;
    uint32_t* cursors;   // [0]=V [1]=H [2]=I [3]=Isl [4]=clip queue [5]=status [6]=refit queue [7]=faces queue [8]=convex queue
                         // [9]=pre-pass queue [10]=image arena (16-byte units) [11]=big clip queue
                         // [12]=half clip queue [13]=retry queue [14]=flagged fragments [15]=flagged pairs (n_failed = 14 + 15)
                         // [16..31]=pairs per cost class (k_prep_pairs) [32..47]=fragments per size class [48..63]=pairs per
                         // pre-pass class [64..79]=pairs per cost class of k_clip_pairs_half (64 = its retry list)
};

struct alignas(16) SRow { uint32_t w[4]; };

struct Pieces
{
    const float* mpos; const uint32_t* mloff; const uint32_t* mllen; const int32_t* mnbr; const uint32_t* mvo; const uint8_t* mtri; const float* mrad;
    const uint32_t* mperm; const float4* mposr_s; const float4* mbsph; const uint32_t* mbo;
    const float* cpos; const uint32_t* cloff; const uint32_t* cllen; const int32_t* cnbr; const uint32_t* cvo; const uint8_t* ctri; const float* crad;
    const uint32_t* cperm; const float4* cposr_s; const float4* cbsph; const uint32_t* cbo;
    uint32_t n;
    // per piece: 1 = some ring of the solid lists a neighbour twice (a sliver with coincident vertices).  Face walks on such a
    // solid need not close, and where the reference's bounded walk stops then depends on its vertex count of the moment: these
    // solids take the literal clipper (literal_clip.h) from the start.
    const uint8_t* mdup; const uint8_t* cdup;
    // the Mesh rings once more in SORTED space (round 4, pieces_dev.hip): sorted slot i (global over the set) has the 16-byte row
    // mrow_s[i] = eight 16-bit words: header (ring length 0..7 | 0x80: some incident face is no triangle | 0x40: more than seven
    // neighbours, not listed), then the neighbours as piece-local SORTED indices (pieces of up to 65 535 vertices; 0xFFFF: none).
    // One aligned load gives the pre-pass of k_prep_pairs a vertex's whole ring; it looks the first clipping plane of a
    // neighbour up by that index.  miperm: sorted index of every vertex.
    // mbsph2 / mbsph3: one sphere per 8 / 64 of the SURTR_SB-vertex spheres (mbo2 / mbo3: first such sphere of every piece)
    const SRow* mrow_s; const uint32_t* miperm;
    const float4* mbsph2; const uint32_t* mbo2; const float4* mbsph3; const uint32_t* mbo3;
};

struct ScratchPool
{
    char* base; size_t per_wg;
    uint32_t CV, CH, VMAX;
};

// Per-workgroup global scratch: positions of the reduced solid (both variants), the wide (32-bit)
// topology used when a solid does not fit the LDS one, three u32 work arrays, scan blocks, pre-pass masks.
struct Scratch
{
    float* pos;
    uint32_t* g_loff; uint32_t* g_llen; int8_t* g_comp; uint32_t* g_ring;
    uint32_t* g_succ; uint32_t* g_pred; uint32_t* g_pcnt;
    uint32_t* aux0; uint32_t* aux1; uint32_t* aux2; uint32_t* aux3;
    int8_t* g_gcomp;       // explicit classification of a plane some live vertex lies in (both variants)
    uint2* blk;
    unsigned long long* gmask; uint2* gblk;
    float* t_pos; uint32_t* t_loff; uint32_t* t_llen; int8_t* t_comp; uint32_t* t_ring;   // squeeze() staging
    uint32_t CV, CH;
};

__device__ static Scratch carve(const ScratchPool& P, uint32_t wg)
{
    Scratch S;
    char* p = P.base + (size_t)wg * P.per_wg;
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    S.pos = (float*)take((size_t)P.CV * 12);
    S.g_loff = (uint32_t*)take((size_t)P.CV * 4);
    S.g_llen = (uint32_t*)take((size_t)P.CV * 4);
    S.g_comp = (int8_t*)take((size_t)P.CV);
    S.g_ring = (uint32_t*)take((size_t)P.CH * 4);
    S.g_succ = (uint32_t*)take((size_t)P.CV * 4);
    S.g_pred = (uint32_t*)take((size_t)P.CV * 4);
    S.g_pcnt = (uint32_t*)take((size_t)P.CV * 4);
    S.aux0 = (uint32_t*)take((size_t)P.CV * 4);
    S.aux1 = (uint32_t*)take((size_t)P.CV * 4);
    S.aux2 = (uint32_t*)take((size_t)P.CV * 4);
    S.aux3 = (uint32_t*)take((size_t)P.CV * 4);
    S.g_gcomp = (int8_t*)take((size_t)P.CV);
    S.blk = (uint2*)take((size_t)(P.CV / SURTR_LANES + 4) * 8);
    S.gmask = (unsigned long long*)take((size_t)(P.VMAX / SURTR_LANES + 2) * 8);
    S.gblk = (uint2*)take((size_t)(P.VMAX / SURTR_LANES + 2) * 8);
    S.t_pos = (float*)take((size_t)P.CV * 12); S.t_loff = (uint32_t*)take((size_t)P.CV * 4); S.t_llen = (uint32_t*)take((size_t)P.CV * 4);
    S.t_comp = (int8_t*)take((size_t)P.CV); S.t_ring = (uint32_t*)take((size_t)P.CH * 4);
    S.CV = P.CV; S.CH = P.CH;
    return S;
}

static size_t scratch_bytes_per_wg(uint32_t CV, uint32_t CH, uint32_t VMAX)
{
    auto r = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return 2 * r((size_t)CV * 12) + 11 * r((size_t)CV * 4) + 3 * r((size_t)CV) + 2 * r((size_t)CH * 4) + r((size_t)(CV / SURTR_LANES + 4) * 8) +
           2 * r((size_t)(VMAX / SURTR_LANES + 2) * 8);
}

struct FaceScratch
{
    int32_t* base; size_t per_wg; uint32_t HF;   // HF = max half-edges of one fragment
};

// One set of resident solids (all Mesh solids, or all Convex solids, of the pieces) in a grow-only pool.
struct PieceSet
{
    float* pos = nullptr; uint32_t* loff = nullptr; uint32_t* llen = nullptr; int32_t* nbr = nullptr; uint32_t* vo = nullptr;
    uint8_t* tri = nullptr; float* rad = nullptr;
    uint32_t* perm = nullptr; float4* posr_s = nullptr; float4* bsph = nullptr; uint32_t* bo = nullptr;
    float* box = nullptr; unsigned long long* key = nullptr; unsigned long long* key2 = nullptr; uint32_t* val = nullptr;    // Morton sort
    uint8_t* dup = nullptr; size_t c_dup = 0;       // per piece: a ring lists a neighbour twice
    // rings in sorted space + two coarser sphere levels (see Pieces)
    uint32_t* iperm = nullptr; SRow* row_s = nullptr;
    float4* bsph2 = nullptr; uint32_t* bo2 = nullptr; float4* bsph3 = nullptr; uint32_t* bo3 = nullptr;
    size_t c_iperm = 0, c_row_s = 0, c_bsph2 = 0, c_bo2 = 0, c_bsph3 = 0, c_bo3 = 0;
    size_t c_pos = 0, c_loff = 0, c_llen = 0, c_nbr = 0, c_vo = 0, c_tri = 0, c_rad = 0, c_perm = 0, c_posr_s = 0, c_bsph = 0,
           c_bo = 0, c_box = 0, c_key = 0, c_key2 = 0, c_val = 0;
    void release()
    {
        void* all[] = {pos, loff, llen, nbr, vo, tri, rad, perm, posr_s, bsph, bo, box, key, key2, val, dup, iperm, row_s, bsph2, bo2, bsph3, bo3};
        for (void* p : all) if (p) (void)hipFree(p);
        *this = PieceSet();
    }
};

// Device buffers of surtr_build_cells (cells_dev.hip): seeds, per-cell slots, and the compact cell arrays.
struct CellBuffers
{
    double* seeds = nullptr; uint32_t* goff = nullptr; char* slots = nullptr; uint32_t* cfo = nullptr; uint32_t* cvo = nullptr;
    int32_t* gen = nullptr; uint32_t* fvo = nullptr; double* verts = nullptr; uint32_t* heads = nullptr;
    size_t c_heads = 0, c_seeds = 0, c_goff = 0, c_slots = 0, c_cfo = 0, c_cvo = 0, c_gen = 0, c_fvo = 0, c_verts = 0;
    uint32_t n = 0, nf = 0, nfv = 0;
    void release()
    {
        void* all[] = {seeds, goff, slots, cfo, cvo, gen, fvo, verts, heads};
        for (void* p : all) if (p) (void)hipFree(p);
        *this = CellBuffers();
    }
};

// Half-size LDS topology of k_clip_pairs_half (capacities; the kernel is in surtr_hip.hip).
#define SURTR_LVS (SURTR_LV / 2u)
#define SURTR_LHS ((SURTR_LH * 11u / 24u) & ~7u)
// It takes solids of up to half its capacity: thin bands can double under the cuts (measured on BASELINE configs[3]: a
// fifth of the pairs admitted with 20 % room outgrew it), and a retry costs the pair twice.
#ifndef SURTR_HALF_ROOM
#define SURTR_HALF_ROOM 2u       // (tests build with 1 to make pairs outgrow it)
#endif
__host__ __device__ static inline bool fits_half(uint32_t n, uint32_t h, uint32_t capVs) { return SURTR_HALF_ROOM * n <= capVs && SURTR_HALF_ROOM * h <= SURTR_LHS; }
static inline bool surtr_fits_half(uint32_t n, uint32_t h) { return fits_half(n, h, SURTR_LVS); }

struct surtr_ctx
{
    int device = 0;
    uint32_t n_wg_faces_alloc = 0;
    bool frags_of_pieces = false;      // the current fragments are an event's over the resident pieces (k_refit may look at the piece a Convex came from)
    // what the CUs can hold (surtr_create); max_wg* below are those, cut down to what the scratch of the current pieces leaves room for
    uint32_t hw_wg = 512, hw_wg_faces = 1024, hw_wg_prep = 1792, hw_wg_big = 48, budget_vmax = 0xFFFFFFFFu, budget_hmax = 0xFFFFFFFFu;
    uint32_t max_wg = 512, max_wg_faces = 1024, max_wg_small = 2048, max_wg_prep = 1792, max_wg_half = 1024;
    ScratchPool pool_rec{}; uint32_t n_wg_rec = 0;          // k_clip_pairs_rec: positions of the cut points only
    uint32_t* d_hlist = nullptr; uint32_t cap_hlist = 0;    // hand-over list of the split arrangement (k_clip_pairs_main -> k_clip_pairs_catch)
    uint32_t n_wg_catch = 128;                              // workgroups of k_clip_pairs_catch (at most; scratch slots are reserved for them)
    uint32_t vmin = 0;                                      // smallest Mesh of the resident pieces
    ScratchPool pool_half{}; uint32_t n_wg_half = 0;       // k_clip_pairs_half: scratch for the half-size LDS topology only
    // Light pairs go to k_clip_pairs_half only when they are most of the event (small pieces: refracture).  Beside a
    // full k_clip_pairs a third kernel costs more than it gains (configs[3]: +0.2 ms even when its workgroups exit at
    // once), and what a large piece leaves of itself in a cell is seldom small enough.  Decided per upload from the piece sizes.
    bool half_on = false;
    PrepPool prep{nullptr, 0, 0}; uint32_t n_wg_prep = 0;
    ImgArena img{nullptr, 0};
    uint32_t* d_order = nullptr; uint32_t cap_order = 0;
    uint32_t* d_forder = nullptr;    // fragments by size class, 16 x cap_frags
    bool wave_big = false;           // the large bands through k_clip_pairs_wave_big (one workgroup per CU) instead of k_clip_pairs_big
    uint32_t n_wg_big = 48;          // workgroups of k_clip_pairs_big
    hipStream_t stream2 = nullptr;   // k_clip_pairs runs here, beside k_clip_pairs_big on the caller's stream
    hipStream_t stream3 = nullptr;   // k_clip_pairs_half (+ the retry launch) beside both
    hipEvent_t ev_prep = nullptr, ev_big = nullptr, ev_half = nullptr, ev_cvx = nullptr;
    uint32_t events_in_flight = 1;      // surtr_set_events_in_flight: contexts the host keeps busy on this GPU at once
    hipStream_t stream = nullptr;
    std::string err;
    // pieces
    uint32_t n_pieces = 0, vmax = 0, hmax = 0, cvmax = 0, chmax = 0;
    CellBuffers cells;               // surtr_build_cells
    PieceSet mset, cset;             // the resident pieces: Mesh and Convex solids + what the pre-pass derives from them (pieces_dev.hip)
    uint32_t* d_upload_err = nullptr; uint32_t cap_outside = 0;
    float* d_group_xf = nullptr; size_t c_group_xf = 0;  // surtr_place_cells_in_pieces: per-group scale / shift
    float* d_world = nullptr; size_t c_world = 0;        // surtr_transform_pieces: the world matrices
    char* sort_tmp = nullptr; size_t c_sort_tmp = 0;      // radix-sort scratch of the Morton sort
    uint32_t* d_from = nullptr; size_t c_from = 0;       // surtr_pieces_from_event: fragment list and offsets
    float upload_ms = 0.f; uint32_t upload_allocs = 0;   // surtr_upload_stats
    uint32_t regroup_rounds = 0;                         // label rounds of the last surtr_event_regroup (one launch)
    uint64_t tot_mv = 0, tot_mh = 0;
    // cells
    uint32_t n_cells = 0, n_faces = 0, cap_pattern_faces = 0, cap_pattern_cells = 0;      // (capacities: set by surtr_build_cells only)
    uint32_t* d_pair_order = nullptr; uint32_t pair_order_begin = 0, pair_order_count = 0, cap_pair_order = 0;   // k_clip_convex: pairs by plane count
    bool pair_order_is_list = false;
    float* d_v012 = nullptr; float4* d_planes = nullptr; uint32_t* d_plane_off = nullptr;
    std::vector<uint32_t> h_plane_off;
    bool planes_ready = false;
    // scratch + arena
    uint32_t user_cv = 0, user_ch = 0;
    uint64_t user_av = 0, user_ah = 0, user_ai = 0;
    ScratchPool pool{}; uint32_t n_wg = 0;
    ScratchPool pool_small{}; uint32_t n_wg_small = 0;      // one-wave kernels (Convex clip, refit)
    FaceScratch fs{}; uint2* d_blk = nullptr; uint32_t blk_per_wg = 0;
    // second tier of k_faces scratch (pieces of more than SURTR_FACES_TIER half-edges): a few workgroups with room for a fragment
    // as large as the largest piece; the first launch hands them the fragments that do not fit its own (d_face_list)
    FaceScratch fs_big{}; uint2* d_blk_big = nullptr; uint32_t blk_per_wg_big = 0, n_wg_faces_big = 0; uint32_t* d_face_list = nullptr;
    Arena arena{};
    PairRec* d_pairs = nullptr; uint32_t cap_pairs = 0;
    FragRec* d_frags = nullptr; uint32_t cap_frags = 0;
    uint32_t* d_frag_status = nullptr;      // per fragment: SURTR_OK or why it has no triangles (u32[cap_frags])
    uint2* d_scanblk = nullptr; uint32_t cap_scanblk = 0;
    surtr_counts* d_counts = nullptr;
    uint32_t* d_face_group = nullptr; uint32_t cap_face_group = 0;      // surtr_place_cells_groups: group of every pattern face
    uint8_t* d_outside = nullptr;
    std::vector<uint8_t> last_outside;       // the `outside` mask of the last event (empty: none), for surtr_event_regroup
    uint2* d_pair_list = nullptr; uint32_t cap_pair_list = 0;
    float color[3] = {0.25f, 0.25f, 0.25f};      // VertexNormalColor::Color written by k_pack (Inc/Poly.h:68 default)
    surtr_counts last{}; bool last_current = false;     // `last` holds the counts of the event in the arena
    bool have_event = false; uint32_t last_flags = 0;
    // staging for downloads
    void* d_blob = nullptr; size_t blob_cap = 0;
    // per-kernel timing with HIP events on the work stream (surtr_set_profiling)
    bool profiling = false;
    // history of the Mesh clip kernel (slot 0: k_clip_pairs, slot 11: k_clip_pairs_wave) over the last events, read without a
    // synchronisation in between (surtr_kernel_history): what a caller with several events in flight averages over
    hipEvent_t hev[16][2] = {}; uint32_t hcount = 0; int hslot[16] = {};
    hipEvent_t ev[32] = {};     // begin/end per kernel slot 0..15
    bool ev_valid[16] = {};
};

#define PROF_BEGIN_ON(i, strm) do { if (ctx->profiling) { (void)hipEventRecord(ctx->ev[2 * (i)], strm); } } while (0)
#define PROF_END_ON(i, strm) do { if (ctx->profiling) { (void)hipEventRecord(ctx->ev[2 * (i) + 1], strm); ctx->ev_valid[i] = true; } } while (0)
// the same into the history ring (only where a kernel was really launched)
#define PROF_HIST_BEGIN(i, strm) do { if (ctx->profiling && ctx->hev[0][0]) { (void)hipEventRecord(ctx->hev[ctx->hcount % 16u][0], strm); } } while (0)
#define PROF_HIST_END(i, strm) do { if (ctx->profiling && ctx->hev[0][0]) { (void)hipEventRecord(ctx->hev[ctx->hcount % 16u][1], strm); ctx->hslot[ctx->hcount % 16u] = (i); ++ctx->hcount; } } while (0)
#define PROF_BEGIN(i) PROF_BEGIN_ON(i, st)
#define PROF_END(i) PROF_END_ON(i, st)

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return SURTR_E_HIP; } \
    } while (0)

static inline void free_dev(void* p) { if (p) (void)hipFree(p); }

// placement of cell groups with per-group scale / shift already in device memory (surtr_hip.hip)
extern "C" int surtr_place_cells_groups_dev(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off, const float* d_scale3, const float* d_shift3);
