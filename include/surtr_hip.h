/* surtr_hip.h -- C ABI of the MI355X-native fracture engine (libsurtr_hip.so).
 *
 * The reference (W298/Surtr) has no FFI layer; its operator boundary for the
 * fracture-event path is the trio of std::function tasks
 *     m_fractureTask      Inc/Surtr.h:272, body Src/Surtr.cpp:1457-1504
 *     m_refittingTask     Inc/Surtr.h:271, body Src/Surtr.cpp:1449-1455
 *     m_initCompoundTask  Inc/Surtr.h:270, body Src/Surtr.cpp:1436-1447
 * fanned out by Surtr::ApplyFracture (Src/Surtr.cpp:2098-2149), Refitting
 * (2405-2413) and InitCompound (2499-2529).  The entry points below are what a
 * host shim that keeps those C++ signatures binds.
 *
 * Conventions
 *   - plain C types only; every array is a flat buffer with an explicit count;
 *   - a solid (Poly::Polyhedron, Inc/Poly.h:15-32) is CSR: pos f32[3*V],
 *     nbr_off u32[V+1], nbr i32[H] (neighbour rings, CCW seen from outside);
 *   - a set of solids shares one vertex numbering: vert_off u32[n+1] gives the
 *     vertex range of solid i, nbr_off is global over all vertices and the
 *     entries of nbr are indices local to their solid;
 *   - planes are float[4] = (nx,ny,nz,d); the kept side is n.x + d <= 0
 *     (Poly::ClipPolyhedron, Src/Poly.cpp:265-500);
 *   - errors are integer codes (the reference throws std::exception,
 *     Src/Poly.cpp:258, Src/VMACH.cpp:91); 0 is success;
 *   - one context per GPU; calls on a context are serialised by the caller;
 *     contexts are independent (thread-compatible, not thread-safe);
 *   - "dev" pointers are device (HBM) addresses, everything else is host memory.
 */
#ifndef SURTR_HIP_H
#define SURTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct surtr_ctx surtr_ctx;

enum {
    SURTR_OK = 0,
    SURTR_E_INVALID = 1,   /* bad argument */
    SURTR_E_TOPOLOGY = 2,  /* asymmetric neighbour links / vertex of degree < 3 (Src/Poly.cpp:253-260) */
    SURTR_E_CAPACITY = 3,  /* a device arena or the caller's output buffer is too small */
    SURTR_E_HIP = 4,       /* HIP runtime error, see surtr_last_error */
    SURTR_E_STATE = 5,     /* call order violated (e.g. event before upload) */
    SURTR_E_NOGPU = 6      /* no HIP device: the engine has no CPU fallback */
};

/* Event flags. */
enum {
    SURTR_EVT_REFIT = 1,   /* run m_refittingTask on every fragment's Convex */
    SURTR_EVT_RENDER = 2   /* ExtractFaces + RenderPolyhedron(isConvex=false) of every Mesh */
};

/* Sizes of one event's result (all counts, not bytes). */
typedef struct surtr_counts {
    uint32_t n_frag;       /* non-empty (cell, piece, island) outputs */
    uint32_t mesh_verts, mesh_nbrs;
    uint32_t conv_verts, conv_nbrs;
    uint32_t n_idx;        /* triangle indices (render buffers) */
    uint32_t n_pairs;      /* (cell, piece) pairs processed */
    uint32_t status;       /* device-side status word, SURTR_OK or an error code */
    uint32_t n_failed;     /* THE DEGENERATE POLICY.  Where the reference leaves its own arrays -- a ring entry that names no vertex
                            * (out of range, a removal mark read as a vertex), a link to a clipped vertex renumbered through a stale
                            * or never-set ID (Src/Poly.cpp:484-493), an ExtractFaces walk that never ends (:100-118) -- nothing it
                            * does afterwards is emulated; the unit is FLAGGED and the event stays SURTR_OK.  n_failed counts the
                            * flags: a pair whose Mesh clip has no valid answer yields no fragment (surtr_pair_status names it); a
                            * fragment whose refit has no valid answer keeps its un-refitted Convex, one whose faces cannot be
                            * extracted has no triangles (frag_status names them).  Asked for one solid (surtr_clip_polyhedron,
                            * surtr_refit_solid, ...), the call returns SURTR_E_TOPOLOGY instead. */
} surtr_counts;

/* Host-side view used by surtr_event_download: every pointer may be NULL to
 * skip that array; otherwise it must hold the size noted (from surtr_counts). */
typedef struct surtr_fragments {
    int32_t* frag_ids;          /* 3*n_frag: cell, piece, island (cell-major order, Src/Surtr.cpp:2133-2146) */
    uint32_t* mesh_vert_off;    /* n_frag+1 */
    float* mesh_pos;            /* 3*mesh_verts */
    uint32_t* mesh_nbr_off;     /* mesh_verts+1 (global over all fragments) */
    int32_t* mesh_nbr;          /* mesh_nbrs, fragment-local indices */
    uint32_t* conv_vert_off;    /* n_frag+1 */
    float* conv_pos;            /* 3*conv_verts */
    uint32_t* conv_nbr_off;     /* conv_verts+1 */
    int32_t* conv_nbr;          /* conv_nbrs */
    float* vnc;                 /* 9*mesh_verts: VertexNormalColor (Inc/Mesh.h:4-13) of every Mesh vertex */
    uint32_t* idx_off;          /* n_frag+1 */
    uint32_t* idx;              /* n_idx, fragment-local vertex indices (Src/Poly.cpp:708-713) */
    uint32_t* frag_status;      /* n_frag: SURTR_OK, or SURTR_E_TOPOLOGY for a fragment without triangles / with its Convex not refitted (see n_failed) */
} surtr_fragments;

/* ---- life cycle -------------------------------------------------------- */
int surtr_create(int device, surtr_ctx** out);
void surtr_destroy(surtr_ctx* ctx);
const char* surtr_strerror(int code);
const char* surtr_last_error(surtr_ctx* ctx);
/* Use this HIP stream (hipStream_t passed as void*) for all work; NULL = default stream. */
int surtr_set_stream(surtr_ctx* ctx, void* hip_stream);
/* The stream the context's kernels (surtr_event_pack_dev included) are launched on: a caller that consumes a device blob on
 * another stream orders the two with an event recorded here. */
int surtr_get_stream(surtr_ctx* ctx, void** hip_stream);
/* How many contexts the host keeps busy on this GPU at once (default 1).  It changes no result, only which kernels an event of a
 * few hundred pairs takes: alone on the GPU such an event is shortest with the arrangement that puts the most threads on every pair
 * (the wide pre-pass, one workgroup of the general clipper per pair); with other events beside it the lean arrangement (the regular
 * pre-pass, the record clipper + catcher) leaves them the room and the steps come faster (a 512-cell block of configs[3], four
 * contexts: 0.65 -> 0.55 ms per step; one at a time 1.29 -> 1.37 ms).  The reference has no counterpart (one event at a time,
 * Src/Surtr.cpp:178-254). */
int surtr_set_events_in_flight(surtr_ctx* ctx, uint32_t n);
/* Override the per-workgroup scratch capacities (vertices, neighbour entries); 0 = automatic. */
int surtr_set_scratch(surtr_ctx* ctx, uint32_t max_verts, uint32_t max_nbrs);
/* Override the result arena capacities (vertices, neighbour entries, indices); 0 = automatic. */
int surtr_set_arena(surtr_ctx* ctx, uint64_t verts, uint64_t nbrs, uint64_t idx);

/* Per-kernel timing with HIP events recorded on the work stream (the reference's TIMER_* phase
 * timers, Inc/pch.h:122-141, Src/Surtr.cpp:1917-1941).  ms[i] = duration of the last launch of
 * 0 clip_pairs (Mesh), 1 frag_table, 2 refit, 3 faces, 4 out_scan, 5 pack, 6 clip_convex, 7 prep_pairs,
 * 8 clip_pairs_big, 9 clip_pairs_half, 10 clip_pairs retry launch (0, 8 and 9 run side by side on the caller's and
 * two internal streams); -1 where not run. */
int surtr_set_profiling(surtr_ctx* ctx, int on);
int surtr_kernel_times(surtr_ctx* ctx, float ms[16]);
/* The durations of the Mesh clip kernel (slot[k] = 0: k_clip_pairs, 11: k_clip_pairs_wave) over the last *n <= 16 events since
 * surtr_set_profiling(ctx, 1), oldest first, without a synchronisation between the events: what a caller that keeps several
 * events in flight (several contexts on several streams) averages for the kernel's launch duration under those conditions. */
int surtr_kernel_history(surtr_ctx* ctx, float ms[16], int slot[16], uint32_t* n);
/* Diagnostic: the device-side counters of the last event (synchronises the stream).  out[0..3] arena use (vertices, ring
 * entries, indices, islands), [5] status, [16+c] pairs of cost class c handed to k_clip_pairs(_big), [32+c] fragments of size
 * class c, [48+c] pairs of pre-pass class c, [64+c] pairs of class c handed to k_clip_pairs_half, [64] pairs that outgrew
 * its half-size LDS topology and were redone by k_clip_pairs, [88] / [89] pairs the record clipper took / handed on to the
 * general clipper ([96+r]: by rule r), [90] solids that were too large for the literal last-resort clipper (more than 32 ring
 * entries at a vertex, or more vertices than its scratch): their pair / fragment is flagged like one without a valid result in
 * the reference -- this counter is how to tell the engine's limit from the reference's undefined behaviour. */
int surtr_queue_stats(surtr_ctx* ctx, uint32_t out[128]);
/* Diagnostic: the status of every pair of the last event (0, or the SURTR_E_* code that pair raised), in pair order
 * (cell-major for surtr_fracture_event, list order for surtr_fracture_pairs).  Works after an event that failed. */
int surtr_pair_status(surtr_ctx* ctx, uint32_t n_pairs, uint32_t* status);
/* A cost estimate per pair of the last event (band vertices its Mesh clip worked on + vertices it produced; small for a pair
 * whose Convex came out empty): what a sharded run balances its contiguous rank blocks with ("optional cost
 * balancing must not change output order"; the reference hands one task per cell to whichever pool thread is free,
 * Src/Surtr.cpp:2129-2131). */
int surtr_event_pair_costs(surtr_ctx* ctx, uint32_t n_pairs, uint32_t* cost);

/* ---- inputs ------------------------------------------------------------ */
/* Replaces compound.PieceVec (Inc/Surtr.h:113-134): n pieces, each a (Convex, Mesh)
 * pair of solids.  Copies host -> device once; validates neighbour symmetry
 * and degree >= 3 (SURTR_E_TOPOLOGY). */
int surtr_upload_pieces(surtr_ctx* ctx, uint32_t n_pieces,
                        const uint32_t* mesh_vert_off, const float* mesh_pos,
                        const uint32_t* mesh_nbr_off, const int32_t* mesh_nbr,
                        const uint32_t* conv_vert_off, const float* conv_pos,
                        const uint32_t* conv_nbr_off, const int32_t* conv_nbr);

/* Replaces the std::vector<VMACH::Polygon3D> fracture pattern
 * (Src/Surtr.cpp:1806-1807): n_cells cells, cell c owns faces
 * [face_off[c], face_off[c+1]); v012 holds the first three vertices of every
 * face in pattern space (9 floats), which is all ConstructFacePlane reads
 * (Src/VMACH.cpp:302-310). */
int surtr_upload_pattern(surtr_ctx* ctx, uint32_t n_cells, const uint32_t* face_off, const float* v012);

/* Surtr::GenerateVoronoi(cellPointVec) (Src/Surtr.cpp:2003-2070, the voro++ call) on the device: the bounded Voronoi cells of
 * the seeds in the unit box, one diagram per group (group g owns seeds [group_seed_off[g], group_seed_off[g+1]); one group for
 * a plain pattern, one per first-level fragment for a refracture), installed as the context's fracture pattern exactly as
 * surtr_upload_pattern would (cell = seed, in seed order).  Same cells, face order and coordinates as surtr_voronoi_cells
 * (canonical cell: DESIGN.md section 5).  seeds: 3 doubles per seed. */
int surtr_build_cells(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_seed_off, const double* seeds,
                      uint32_t* n_faces, uint32_t* n_face_verts);
/* The cells of the last surtr_build_cells in the layout of surtr_voronoi_cells (+ v012, 9 floats per face); NULL skips an array. */
int surtr_download_cells(surtr_ctx* ctx, uint32_t* cell_face_off, int32_t* face_gen, uint32_t* face_vert_off, double* verts, float* v012);

/* Polygon3D::Scale + Translate + ConstructFacePlane for every face, on the
 * device (Src/VMACH.cpp:506-534; per event at Src/Surtr.cpp:1891-1896). */
int surtr_place_cells(surtr_ctx* ctx, const float scale[3], const float translate[3]);

/* Alternative to pattern+place: give the cell planes directly. */
int surtr_upload_planes(surtr_ctx* ctx, uint32_t n_cells, const uint32_t* plane_off, const float* planes);

/* ---- the event --------------------------------------------------------- */
/* ApplyFracture over cells [cell_begin, cell_end) x all pieces on device-resident
 * inputs: clip Convex, clip Mesh, split islands, then (flags) refit and
 * triangulate.  `outside` (n_pieces bytes, may be NULL) marks pieces skipped as
 * in Src/Surtr.cpp:1463-1464.  Results stay on the device; counts are returned.
 * Synchronises the stream once to read the counts. */
int surtr_fracture_event(surtr_ctx* ctx, uint32_t cell_begin, uint32_t cell_end,
                         const uint8_t* outside, uint32_t flags, surtr_counts* counts);

/* Same, but does not synchronise or read anything back: for timing loops.
 * The counts of the last event are fetched with surtr_event_counts. */
int surtr_fracture_event_async(surtr_ctx* ctx, uint32_t cell_begin, uint32_t cell_end,
                               const uint8_t* outside, uint32_t flags);
int surtr_event_counts(surtr_ctx* ctx, surtr_counts* counts);

/* Recursive refracture (BASELINE configs[4]): an explicit list of (cell, piece) pairs, processed and output in
 * list order (e.g. fragment-major: every first-level fragment with its own cells).  Same pipeline and result
 * layout as surtr_fracture_event; frag_ids carry the listed cell / piece numbers. */
int surtr_fracture_pairs_async(surtr_ctx* ctx, uint32_t n_pairs, const uint32_t* pair_cell, const uint32_t* pair_piece,
                               uint32_t flags);
/* surtr_place_cells with one (scale, translate) per group of consecutive cells: group g owns cells
 * [group_cell_off[g], group_cell_off[g+1]); scale3 / translate3 hold 3 floats per group. */
int surtr_place_cells_groups(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off, const float* scale3,
                             const float* translate3);

/* The same with group g placed over the bounding box of resident piece g's Mesh (scale = its extent, translate = its centre,
 * Src/Surtr.cpp:1799-1803 applied per piece): the boxes are taken on the device, nothing is read back.  n_groups must equal the
 * number of resident pieces (a recursive refracture after surtr_pieces_from_event + surtr_build_cells). */
int surtr_place_cells_in_pieces(surtr_ctx* ctx, uint32_t n_groups, const uint32_t* group_cell_off);

/* Bytes of the packed device blob holding the last event's fragments. */
size_t surtr_event_blob_bytes(const surtr_counts* counts);
/* Packs the last event's fragments into one contiguous device buffer (for an
 * all-gather over RCCL or a single D2H copy).  Layout: see DESIGN.md. */
int surtr_event_pack_dev(surtr_ctx* ctx, void* dev_blob, size_t capacity_bytes);
/* Splits a host copy of a blob into the arrays of surtr_fragments. */
int surtr_blob_unpack_host(const void* host_blob, size_t bytes, surtr_counts* counts, surtr_fragments* out);
/* Convenience: pack + copy to host + unpack into caller arrays. */
int surtr_event_download(surtr_ctx* ctx, surtr_fragments* out);

/* ---- single-solid operators (the Poly / Kdop API surface) -------------- */
/* Poly::ClipPolyhedron(polyhedron, planes) for one solid, Src/Poly.cpp:556-566.
 * Count-then-fill: call with out_* NULL to get sizes. */
int surtr_clip_polyhedron(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr,
                          uint32_t n_planes, const float* planes,
                          uint32_t* out_nv, uint32_t* out_nh, float* out_pos, uint32_t* out_nbr_off, int32_t* out_nbr);

/* The three per-Piece tasks for ONE solid (what m_refittingTask and m_initCompoundTask do to a Piece, Inc/Surtr.h:270-271),
 * run by the same kernels as the event.  They use the event arena: the fragments of the last event are gone afterwards.
 *
 * m_refittingTask (Src/Surtr.cpp:1449-1455): ConvexHull(mesh points, min(n, 4)) -> Kdop::Calc(mesh) ->
 * ClipWithPolyhedron(convex).  Count-then-fill like surtr_clip_polyhedron. */
int surtr_refit_solid(surtr_ctx* ctx, uint32_t mesh_nv, const float* mesh_pos, const uint32_t* mesh_nbr_off, const int32_t* mesh_nbr,
                      uint32_t conv_nv, const float* conv_pos, const uint32_t* conv_nbr_off, const int32_t* conv_nbr,
                      uint32_t* out_nv, uint32_t* out_nh, float* out_pos, uint32_t* out_nbr_off, int32_t* out_nbr);
/* Poly::ExtractFaces (Src/Poly.cpp:89-126): face loops in visiting order, face f = face_idx[face_off[f] .. face_off[f+1]).
 * Count-then-fill: NULL arrays return n_faces / n_face_idx only.  SURTR_E_TOPOLOGY where the reference's walk never ends. */
int surtr_extract_faces(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr,
                        uint32_t* n_faces, uint32_t* n_face_idx, uint32_t* face_off, int32_t* face_idx);
/* Poly::ExtractFaces + Poly::RenderPolyhedron (Src/Poly.cpp:681-714): vnc = 9 floats per vertex (VertexNormalColor with
 * `color`, NULL = 0.25), idx = triangle indices; is_convex selects the fan (:696-706) or EarClipping (:707-713).
 * Count-then-fill: idx NULL returns n_idx. */
int surtr_triangulate(surtr_ctx* ctx, uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr, int is_convex,
                      const float color[3], float* vnc, uint32_t* n_idx, uint32_t* idx);

/* Presents n host pieces as the fragments of an event (fragment k = Mesh k + Convex k, ids = frag_ids[3k..] or
 * (k, 0, 0) when NULL), so that surtr_event_refit / surtr_event_triangulate / surtr_event_download /
 * surtr_pieces_from_event work on solids that did not come out of surtr_fracture_event. */
int surtr_load_fragments(surtr_ctx* ctx, uint32_t n,
                         const uint32_t* mesh_vert_off, const float* mesh_pos, const uint32_t* mesh_nbr_off, const int32_t* mesh_nbr,
                         const uint32_t* conv_vert_off, const float* conv_pos, const uint32_t* conv_nbr_off, const int32_t* conv_nbr,
                         const int32_t* frag_ids);
/* m_initCompoundTask's triangulation (Src/Surtr.cpp:1436-1447) of every Mesh of the current fragments; is_convex as above. */
int surtr_event_triangulate(surtr_ctx* ctx, int is_convex);

/* ---- device-resident pieces -------------------------------------------- */
/* Poly::Transform (Src/Poly.cpp:580-585) of piece i's Convex and Mesh by world[16*i .. 16*i+16) -- the row-major XMMATRIX of
 * m_structuredBufferData[i].WorldMatrix, transposed before use as the reference does -- on the resident copies
 * (ExecuteFractureRoutine's pre-transform, Src/Surtr.cpp:1846-1851).  No host round trip. */
int surtr_transform_pieces(surtr_ctx* ctx, uint32_t n_pieces, const float* world);
/* The fragments of the last event become the resident pieces (recursive refracture without leaving HBM): fragment k is kept
 * when keep[k] != 0; keep == NULL keeps every fragment that is a solid (at least four vertices in Mesh and Convex; a Convex the
 * refit clipped away yields nothing in the reference either, Src/Surtr.cpp:1466-1468).  Kept fragments become pieces 0, 1, ...
 * in fragment order; n_pieces returns their number. */
int surtr_pieces_from_event(surtr_ctx* ctx, const uint8_t* keep, uint32_t* n_pieces);
/* Reads resident piece `piece` back (set 0 = Mesh, 1 = Convex): count-then-fill like surtr_clip_polyhedron.  For tests and for
 * Poly::Transform of a single host polyhedron; the event path never needs it. */
int surtr_download_piece(surtr_ctx* ctx, uint32_t piece, int set, uint32_t* out_nv, uint32_t* out_nh, float* out_pos,
                         uint32_t* out_nbr_off, int32_t* out_nbr);
/* Host time of the last surtr_upload_pieces / surtr_pieces_from_event / surtr_transform_pieces call, in milliseconds,
 * and how many device allocations it made (0 in steady state: the piece buffers are pooled). */
int surtr_upload_stats(surtr_ctx* ctx, float* ms, uint32_t* n_alloc);

/* ---- host-side helpers of the harness (no GPU needed) ------------------ */
/* Poly::ExtractNeighborFromMesh, Src/Poly.cpp:128-263: welded triangle soup ->
 * neighbour rings.  nbr must hold 2*3*n_tris entries at most; returns
 * SURTR_E_TOPOLOGY where the reference throws. */
int surtr_neighbors_from_mesh(uint32_t nv, uint32_t n_tris, const int32_t* tris, uint32_t* nbr_off, int32_t* nbr);

/* The same on the device (directed-edge hash table + one fan walk per vertex): identical rings.  kernel_ms (may be NULL)
 * returns the time of the kernels, copies excluded. */
int surtr_neighbors_from_mesh_dev(surtr_ctx* ctx, uint32_t nv, uint32_t n_tris, const int32_t* tris, uint32_t* nbr_off, int32_t* nbr,
                                  float* kernel_ms);

/* Canonical bounded Voronoi cells of n seeds in the unit box (replaces the
 * voro++ call of Src/Surtr.cpp:2003-2070; see DESIGN.md for the face order).
 * Count-then-fill: pass NULL arrays to get n_faces / n_face_verts. */
int surtr_voronoi_cells(uint32_t n, const double* seeds, uint32_t* n_faces, uint32_t* n_face_verts,
                        uint32_t* cell_face_off, int32_t* face_gen, uint32_t* face_vert_off, double* verts);

/* VMACH::ConvexHull(points, limit) + Surtr::GenerateICHNormal (Src/VMACH.cpp:869-1161, Src/Surtr.cpp:1961-1974):
 * unit normals of the faces of the greedy limited hull, in face creation order.  Count-then-fill. */
int surtr_hull_normals(uint32_t n, const float* points, uint32_t limit, uint32_t capacity, float* normals, uint32_t* count);

/* Kdop::KdopContainer::Calc(vertices, maxAxisScale, planeGapInv) (Src/Kdop.cpp:15-51): for every normal the Min
 * plane then the Max plane (the order ClipWithPolyhedron clips in, :166-179); planes holds 8 floats per normal. */
int surtr_kdop_ach_planes(uint32_t n, const float* points, uint32_t k, const float* normals, double max_axis_scale,
                          float plane_gap_inv, float* planes);

/* Plane(p0, p1, p2) as PolygonFace::ConstructFacePlane builds it (Src/VMACH.cpp:302-310; normalised, SimpleMath.inl:2773-2780). */
int surtr_plane_from_points(const float p0[3], const float p1[3], const float p2[3], float plane[4]);

/* Kdop::KdopContainer::Calc(const Poly::Polyhedron&) (Src/Kdop.cpp:92-115): Min plane then Max plane per normal through the
 * first extreme vertices, not normalised, no gap (the refit variant; k_refit does the same per fragment on the device). */
int surtr_kdop_planes(uint32_t n, const float* points, uint32_t k, const float* normals, float* planes);

/* Poly::Moments (Src/Poly.cpp:55-87): signed volume and centroid of a closed solid (host; the test invariant of
 * fragment volumes partition the input). */
int surtr_moments(uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr, double* volume, float centroid[3]);

/* ---- mesh files either side of the path (host) --------------------------------- */
/* Wavefront OBJ in, with the conventions of Surtr::LoadModelData and its assimp flags (Src/Surtr.cpp:2683-2727):
 * one vertex per distinct position in order of first use, polygons as fans, winding flipped, x negated, then
 * scale/translate.  Count-then-fill: with pos or tris NULL only *n_verts / *n_tris are written. */
int surtr_read_obj(const char* path, const float scale[3], const float translate[3], uint32_t cap_verts, uint32_t cap_tris,
                   float* pos, int32_t* tris, uint32_t* n_verts, uint32_t* n_tris);
/* The render buffers of an event (surtr_fragments: vnc = 36-byte VertexNormalColor, Inc/Mesh.h:4-13) as one OBJ
 * object per fragment. */
int surtr_write_obj(const char* path, uint32_t n_frag, const int32_t* frag_ids, const uint32_t* mesh_vert_off, const float* vnc,
                    const uint32_t* idx_off, const uint32_t* idx);

/* ---- the step after the event: compound regrouping (host side) ---------------- */
/* Surtr::ConvexOutOfSphere (Src/Surtr.cpp:2415-2458) for one Convex; sphere_points are already placed. */
int surtr_convex_out_of_sphere(uint32_t nv, const float* pos, const uint32_t* nbr_off, const int32_t* nbr, uint32_t n_sphere,
                               const float* sphere_points, const float origin[3], float radius, int* out);

/* Bind sets of ApplyFracture (Src/Surtr.cpp:2103-2146) + MergeOutOfImpact (:2368-2403, when `partial`) +
 * HandleConvexIsland (:2203-2366) on the un-refitted Convex solids of an event.
 * Pieces [0, n_outside) are the pieces kept whole outside the impact sphere (bind 0); pieces
 * [n_outside, n_pieces) are the event's fragments in output order, piece_cell[p] = their cell (consecutive
 * fragments of one cell form one compound).  conv_nbr_off is global over all vertices, conv_nbr local per piece.
 * Output: compound c owns compound_piece[compound_off[c] .. compound_off[c+1]) (ascending piece indices);
 * compound 0 is the outside set; compound_off needs n_pieces + 2 entries, compound_piece n_pieces. */
int surtr_regroup(uint32_t n_pieces, uint32_t n_outside, const int32_t* piece_cell,
                  const uint32_t* conv_vert_off, const float* conv_pos, const uint32_t* conv_nbr_off, const int32_t* conv_nbr,
                  int partial, uint32_t n_sphere, const float* sphere_points, const float origin[3], float radius,
                  uint32_t* n_compounds, uint32_t* compound_off, int32_t* compound_piece);

/* The same regrouping as a device step on the last event (regroup_dev.hip): the Convex solids stay in HBM; faces, planes, the
 * ConvexOutOfSphere test, candidate face pairs (radix sort by |d|), the overlap tests and the label propagation over pieces
 * run in kernels; only per-piece flags and labels come back.  Pieces are numbered as for surtr_regroup: the resident pieces
 * the event skipped (its `outside` mask, ascending), then the event's fragments in output order.  Call it BEFORE
 * surtr_event_refit (the reference regroups on the un-refitted Convex solids).  compound_off needs *n_pieces + 2 entries,
 * compound_piece *n_pieces (call with both NULL to get n_pieces). */
int surtr_event_regroup(surtr_ctx* ctx, int partial, uint32_t n_sphere, const float* sphere_points, const float origin[3], float radius,
                        uint32_t* n_pieces, uint32_t* n_compounds, uint32_t* compound_off, int32_t* compound_piece);

/* Runs m_refittingTask (and the output scan) on the fragments of the last event: the reference regroups on the
 * un-refitted Convex solids and refits afterwards (Src/Surtr.cpp:1921-1939). */
int surtr_event_refit(surtr_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SURTR_HIP_H */
