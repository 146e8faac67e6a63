// surtr_host.hpp -- C++ host layer that keeps the reference's type and function names for the
// fracture-event path (Inc/Poly.h:15-76, Inc/Kdop.h:16-41, Inc/VMACH.h:17-86, Inc/Surtr.h:113-134, 270-272) and
// forwards the work to the HIP engine through the C ABI of include/surtr_hip.h.
//
// It is what a front-end (the DX12 demo or a headless harness) links instead of Src/Poly.cpp / Src/Kdop.cpp and the
// hot-path members of Src/Surtr.cpp.  No geometry is computed here: the types are thin views that are flattened to CSR
// buffers, sent through the C ABI and rebuilt from the result.  The free functions of namespace Poly / Kdop keep the
// reference's signatures; they run on a process-wide default engine (DefaultEngine()).
#pragma once
#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/surtr_hip.h"

namespace surtr {

struct Vector3 { float x = 0, y = 0, z = 0; Vector3() = default; Vector3(float a, float b, float c) : x(a), y(b), z(c) {} };
struct Plane { float x = 0, y = 0, z = 0, w = 0; Plane() = default; Plane(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} };
// DirectX::XMMATRIX as the reference stores it in m_structuredBufferData[i].WorldMatrix (row-major, 16 floats).
struct Matrix { float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; };

// The reference throws std::exception (Src/Poly.cpp:258); the C ABI returns codes; this layer rethrows.
struct Error : std::runtime_error
{
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// VertexNormalColor, Inc/Mesh.h:4-13: three XMFLOAT3, 36 bytes, what DynamicMesh::UploadVB copies into the D3D12 upload
// heap (Inc/Mesh.h:163-183).  The engine's `vnc` buffers have exactly this layout.
struct VertexNormalColor { float Position[3]; float Normal[3]; float Color[3]; };
static_assert(sizeof(VertexNormalColor) == 36, "VertexNormalColor is three packed XMFLOAT3 (Inc/Mesh.h:4-13)");
static_assert(offsetof(VertexNormalColor, Position) == 0 && offsetof(VertexNormalColor, Normal) == 12 && offsetof(VertexNormalColor, Color) == 24,
              "field offsets of Inc/Mesh.h:4-13");
static_assert(sizeof(Vector3) == 12, "Vector3 = XMFLOAT3: a Convex's positions are a stride-12 PxConvexMeshDesc::points array (Src/Surtr.cpp:2531-2553)");

class FractureEngine;
// Engine behind the free functions below (created on first use on device `SetDefaultDevice`, default 0).
FractureEngine& DefaultEngine();
void SetDefaultDevice(int device);

namespace VMACH {
// Inc/VMACH.h:17-58, the members the path reads and the ones that keep FacePlane current.
struct PolygonFace
{
    bool GuaranteeConvex = true;
    std::vector<Vector3> VertexVec;
    Plane FacePlane;
    bool FacePlaneConstructed = false;
    PolygonFace() = default;
    explicit PolygonFace(bool guaranteeConvex) : GuaranteeConvex(guaranteeConvex) {}
    PolygonFace(bool guaranteeConvex, std::vector<Vector3> vertexVec) : GuaranteeConvex(guaranteeConvex), VertexVec(std::move(vertexVec)) { ConstructFacePlane(); }
    void AddVertex(const Vector3& newVertex);          // Src/VMACH.cpp:289-300 (exact duplicates are dropped)
    void ConstructFacePlane();                         // Src/VMACH.cpp:302-310: Plane(V0, V1, V2), normalised
    Vector3 GetNormal() const;                         // Src/VMACH.cpp:88-97 (throws if the plane was never built)
};
struct Polygon3D                                       // Inc/VMACH.h:60-86
{
    bool GuaranteeConvex = true;
    std::vector<PolygonFace> FaceVec;
    Polygon3D() = default;
    explicit Polygon3D(bool guaranteeConvex) : GuaranteeConvex(guaranteeConvex) {}
    void AddFace(const PolygonFace& newFace) { FaceVec.push_back(newFace); }
    void Translate(const Vector3& vector);             // Src/VMACH.cpp:506-514 (+ ConstructFacePlane)
    void Scale(const Vector3& vector);                 // Src/VMACH.cpp:525-534
};
Plane ConstructFacePlane(const PolygonFace& f);
} // namespace VMACH

namespace Poly {
struct Vertex                                   // Inc/Poly.h:15-30
{
    Vector3 Position;
    std::vector<int> NeighborVertexVec;
    int comp = 1;
    mutable int ID = -1;
};
typedef std::vector<Vertex> Polyhedron;         // Inc/Poly.h:31
typedef std::vector<std::vector<int>> Extract;  // Inc/Poly.h:32

// Inc/Poly.h:35-76, same names and argument meaning.
void InitPolyhedron(Polyhedron& polyhedron, const std::vector<Vector3>& positionVec, const std::vector<std::vector<int>>& neighborVec);
void Moments(double& zerothMoment, Vector3& firstMoment, const Polyhedron& polyhedron);          // Src/Poly.cpp:55-87
Extract* ExtractFaces(const Polyhedron& polyhedron);                                              // Src/Poly.cpp:89-126 (caller owns)
std::vector<std::vector<int>> ExtractNeighborFromMesh(std::vector<Vector3>& vertices, std::vector<int>& indices);   // Src/Poly.cpp:128-263
void ClipPolyhedron(Polyhedron& polyhedron, const std::vector<Plane>& planes);                    // Src/Poly.cpp:265-554
Polyhedron ClipPolyhedron(const Polyhedron& polyhedron, const VMACH::Polygon3D& polygon3D);       // Src/Poly.cpp:556-566
void Translate(Polyhedron& polyhedron, const Vector3& v);
void Scale(Polyhedron& polyhedron, const Vector3& v);
void Transform(Polyhedron& polyhedron, const Matrix& matrix);                                     // Src/Poly.cpp:580-585
Polyhedron GetBB();                                                                               // Src/Poly.cpp:587-617
// Src/Poly.cpp:681-714.  `extract` must be ExtractFaces(poly) (the kernels derive the faces themselves); vertices and
// indices are appended after what the two vectors hold, indices offset by the vertices already there, as in the reference.
void RenderPolyhedron(std::vector<VertexNormalColor>& vertexData, std::vector<uint32_t>& indexData, const Polyhedron& poly,
                      const Extract* extract, bool isConvex = true, Vector3 color = Vector3(0.25f, 0.25f, 0.25f));
} // namespace Poly

namespace Kdop {
struct KdopElement                              // Inc/Kdop.h:16-27
{
    Vector3 Normal, MinVertex, MaxVertex;
    double MinDist = DBL_MAX, MaxDist = -DBL_MAX;
    Plane MinPlane, MaxPlane;
    explicit KdopElement(const Vector3& normal) : Normal(normal) {}
};
struct KdopContainer                            // Inc/Kdop.h:29-41
{
    std::vector<KdopElement> ElementVec;
    explicit KdopContainer(const std::vector<Vector3>& normalVec);                                          // Src/Kdop.cpp:10-13
    void Calc(const std::vector<Vector3>& vertices, const double& maxAxisScale, const float& planeGapInv);   // Src/Kdop.cpp:15-51
    void Calc(const Poly::Polyhedron& mesh);                                                                // Src/Kdop.cpp:92-115
    Poly::Polyhedron ClipWithPolyhedron(const Poly::Polyhedron& polyhedron);                                // Src/Kdop.cpp:166-179
};
} // namespace Kdop

// Surtr::GenerateICHNormal (Src/Surtr.cpp:1961-1982): unit face normals of VMACH::ConvexHull(vertices, limitCnt).
std::vector<Vector3> GenerateICHNormal(const std::vector<Vector3>& vertices, int limitCnt);

// Surtr::LoadModelData (Src/Surtr.cpp:2683-2727) for Wavefront OBJ: positions with x negated, scaled, translated; triangles
// with the winding flipped; identical positions joined (normals are not read: the path does not use them).
void LoadModelData(const std::string& fileName, const Vector3& scale, const Vector3& translate, std::vector<Vector3>& vertices,
                   std::vector<int>& indices);

struct Piece                                                // Inc/Surtr.h:113-119 (the reference heap-allocates them: Piece*)
{
    Poly::Polyhedron Convex, Mesh;
    Piece() = default;
    Piece(const Poly::Polyhedron& convex, const Poly::Polyhedron& mesh) : Convex(convex), Mesh(mesh) {}
};
struct Compound { std::vector<Piece> PieceVec; };           // Inc/Surtr.h:121-127 (value semantics: no leaks)
// Inc/Surtr.h:129-134: pieces + bind sets (CompoundBind[0] = the pieces outside the impact).
struct CompoundInfo { std::vector<Piece> PieceVec; std::vector<std::set<int>> CompoundBind; };

// Surtr::FractureArgs, the members the event reads (Inc/Surtr.h:90-110; Src/Surtr.cpp:1885-1912).
struct FractureArgs
{
    bool PartialFracture = false;
    Vector3 ImpactPosition;
    float ImpactRadius = 1.f;
    int Seed = 46354;
};

struct FragmentRender { std::vector<VertexNormalColor> vertexData; std::vector<uint32_t> indexData; };

// Surtr::ConvexOutOfSphere (Src/Surtr.cpp:2415-2458); `extract` = ExtractFaces(polyhedron) (derived again inside).
bool ConvexOutOfSphere(const Poly::Polyhedron& polyhedron, const Poly::Extract* extract, const std::vector<Vector3>& spherePointCloud,
                       const Vector3 origin, const float radius);
// Surtr::MergeOutOfImpact (Src/Surtr.cpp:2368-2403): pieces of the compounds 1.. that are out of the sphere move to compound 0;
// compounds left empty are removed.
void MergeOutOfImpact(CompoundInfo& compoundInfo, const std::vector<Vector3>& spherePointCloud, const Vector3 origin, const float radius);
// Surtr::HandleConvexIsland (Src/Surtr.cpp:2203-2366): every compound is split into the groups of pieces that touch through a
// pair of opposite, overlapping faces; the first group stays, the others are appended.
void HandleConvexIsland(CompoundInfo& compoundInfo);

// The three per-Piece tasks with the reference's EXACT signatures (Inc/Surtr.h:270-272: pointer semantics; what they return is
// the caller's, as Src/Surtr.cpp:1494,1499 `new Piece`), on the default engine.  m_initCompoundTask's result pair
// <PxConvexMeshGeometry, DynamicMesh*> needs PhysX and D3D12: InitCompoundResult holds what those two are built from.
std::vector<Piece*> FractureTask(const VMACH::Polygon3D& voroPoly, const std::vector<Piece*>& targetPieceVec, const std::set<int>& outside);
void RefittingTask(Piece* piece);
struct InitCompoundResult { std::vector<Vector3> ConvexPoints; FragmentRender Mesh; };
InitCompoundResult InitCompoundTask(const Piece* piece, const Poly::Extract* extract, bool renderConvex);

struct Fragment { int cell, piece, island; Piece piece_data; FragmentRender render; int status = 0; };

// One engine per GPU (wraps surtr_ctx).  Mirrors the calls of Surtr::DoFracture (Src/Surtr.cpp:1885-1959).
class FractureEngine
{
public:
    explicit FractureEngine(int device = 0);
    ~FractureEngine();
    FractureEngine(const FractureEngine&) = delete;
    FractureEngine& operator=(const FractureEngine&) = delete;

    // Fracture pattern in pattern space (GenerateFracturePattern / GenerateVoronoi, Src/Surtr.cpp:2003-2096).
    void SetPattern(const std::vector<VMACH::Polygon3D>& voroPolyVec);
    // Surtr::GenerateVoronoi(cellPointVec) (Src/Surtr.cpp:2003-2070; voro++ replaced by the canonical cell of DESIGN section 5)
    // ON THE DEVICE: surtr_build_cells, which also installs the cells as this engine's pattern, + surtr_download_cells.
    std::vector<VMACH::Polygon3D> GenerateVoronoi(const std::vector<Vector3>& cellPointVec);
    // Surtr::GenerateVoronoi(cellCnt) (Src/Surtr.cpp:1984-2001): mt19937(seed), uniform(-0.5, 0.5) in x, y, z draw order.
    std::vector<VMACH::Polygon3D> GenerateVoronoi(int cellCnt, int seed = 46354);
    // Surtr::GenerateFracturePattern (Src/Surtr.cpp:2072-2096): exponential length (clamped to [1e-12, 0.5]) x normalised
    // uniform(-1, 1)^3 direction.
    std::vector<VMACH::Polygon3D> GenerateFracturePattern(int cellCount, double mean, int seed = 46354);
    // the same cells from the host builder (surtr_voronoi_cells): the CPU-tier cross-check of the device builder
    static std::vector<VMACH::Polygon3D> GenerateVoronoiHost(const std::vector<Vector3>& cellPointVec);
    // voro.Scale(scale); voro.Translate(translate) for every cell (Src/Surtr.cpp:1799-1803, 1891-1896).
    void PlacePattern(const Vector3& scale, const Vector3& translate);
    void SetCompound(const Compound& compound);
    // ExecuteFractureRoutine's pre-transform (Src/Surtr.cpp:1846-1851): Poly::Transform of every resident piece's Convex and
    // Mesh by its world matrix, on the device.
    void TransformCompound(const std::vector<Matrix>& worldMatrices);
    // ApplyFracture + Refitting + per-piece ExtractFaces/RenderPolyhedron (Src/Surtr.cpp:2098-2149, 2405-2413, 1436-1447):
    // fragments in cell-major, piece, island order.
    std::vector<Fragment> ApplyFracture(const std::set<int>& outside = {}, bool refit = true, bool render = true,
                                        uint32_t cellBegin = 0, uint32_t cellEnd = 0xFFFFFFFFu);
    // Recursive refracture: the fragments of the last ApplyFracture (all, or those with keep[k] != 0) become the compound of
    // the next one without leaving the device.  Returns the number of pieces.
    uint32_t CompoundFromLastEvent(const std::vector<uint8_t>& keep = {});
    // Surtr::Refitting (Src/Surtr.cpp:2405-2413): m_refittingTask on every piece.
    void Refitting(std::vector<Piece>& pieceVec);
    // Steps 1-6 of Surtr::PrepareFracture (Src/Surtr.cpp:1750-1785): ICH(limit) face normals -> k-DOP slabs moved out by
    // MaxAxisScale / planeGapInverse -> the 2x bounding box clipped by all of them (the approximate convex hull).
    Poly::Polyhedron BuildACH(const std::vector<Vector3>& vertices, uint32_t ichIncludePointLimit = 20, float achPlaneGapInverse = 2000.f);
    // Surtr::PrepareFracture (Src/Surtr.cpp:1747-1827) from a raw triangle mesh: ACH, neighbour rings, `cellCnt` Voronoi cells
    // placed over the bounding box, one event with refit + extraction.  Returns the pieces of the initial compound.
    std::vector<Fragment> PrepareFracture(std::vector<Vector3>& vertices, std::vector<int>& indices, const std::vector<Vector3>& cellPointVec);
    // Poly::ClipPolyhedron(polyhedron, polygon3D), Src/Poly.cpp:556-566.
    Poly::Polyhedron ClipPolyhedron(const Poly::Polyhedron& polyhedron, const VMACH::Polygon3D& polygon3D);
    Poly::Polyhedron ClipPolyhedron(const Poly::Polyhedron& polyhedron, const std::vector<Plane>& planes);
    // The per-Piece tasks on one solid (m_refittingTask, Poly::ExtractFaces, Poly::RenderPolyhedron, Poly::Transform).
    Poly::Polyhedron RefitSolid(const Poly::Polyhedron& mesh, const Poly::Polyhedron& convex);
    Poly::Extract ExtractFaces(const Poly::Polyhedron& polyhedron);
    FragmentRender RenderPolyhedron(const Poly::Polyhedron& poly, bool isConvex, const Vector3& color);
    Poly::Polyhedron TransformSolid(const Poly::Polyhedron& polyhedron, const Matrix& matrix);
    // What DoFracture regrouped (for tests and tools): the un-refitted Convex solids in piece order (the compound's pieces the
    // event skipped, ascending, then the event's fragments), the cell of every piece (-1: skipped), the bind sets found.
    struct FractureTrace { std::vector<Poly::Polyhedron> Convex; std::vector<int> PieceCell; uint32_t nOutside = 0; std::vector<std::set<int>> CompoundBind; };
    // Surtr::DoFracture (Src/Surtr.cpp:1885-1959) on the device, in the reference's order: pattern scaled by 2 * maxAxisScale
    // and moved to the impact, sphere cloud scaled by the radius and moved to the impact, ApplyFracture (pieces out of the
    // sphere are skipped when PartialFracture), MergeOutOfImpact (when PartialFracture), HandleConvexIsland -- both on the
    // un-refitted Convex solids, surtr_event_regroup --, Refitting (surtr_event_refit), one Compound per bind set.  The
    // pattern is the engine's (SetPattern / GenerateVoronoi / GenerateFracturePattern), in pattern space.
    std::vector<Compound> DoFracture(const Compound& targetCompound, float maxAxisScale, const FractureArgs& args,
                                     const std::vector<Vector3>& spherePointCloud, FractureTrace* trace = nullptr);
    surtr_counts LastCounts() const { return counts_; }
    // The degenerate policy at this level (include/surtr_hip.h, surtr_counts::n_failed): where the reference leaves its own
    // arrays the engine flags the unit instead of emulating what the reference's memory happens to hold -- a flagged (cell,
    // piece) pair yields NO fragment, a flagged fragment keeps its un-refitted Convex / has no triangles.  A caller must not
    // lose pieces without knowing: by default ApplyFracture / PrepareFracture / DoFracture THROW Error(SURTR_E_TOPOLOGY) when
    // the event flagged anything (the reference's own signal on that path is `throw std::exception()`, Src/Poly.cpp:258);
    // after AllowFlagged(true) they return what is valid and LastFlagged() names what was left out.
    struct FlaggedUnits { uint32_t n_failed = 0; std::vector<std::pair<int, int>> pairs /* (cell, piece): no fragment */; std::vector<uint32_t> fragments /* output index */; };
    void AllowFlagged(bool allow) { allow_flagged_ = allow; }
    const FlaggedUnits& LastFlagged() const { return flagged_; }
    // How many engines the host keeps busy on this GPU at once (surtr_set_events_in_flight: no result changes, events of a few
    // hundred pairs then take the kernels that leave the other engines room).  The reference runs one event at a time.
    void SetEventsInFlight(uint32_t n) { check(surtr_set_events_in_flight(ctx_, n), "surtr_set_events_in_flight"); }
    surtr_ctx* Raw() { return ctx_; }

private:
    void check(int rc, const char* what);
    std::vector<Fragment> download_fragments(bool render);
    void report_flagged(const std::vector<Fragment>& frags, uint32_t cellBegin, uint32_t cellEnd, const char* what);
    bool allow_flagged_ = false;
    FlaggedUnits flagged_;
    surtr_ctx* ctx_ = nullptr;
    uint32_t n_cells_ = 0, n_pieces_ = 0;
    surtr_counts counts_{};
};

} // namespace surtr
