// surtr_host.hpp -- C++ host layer that keeps the reference's type and function names for the
// fracture-event path (Inc/Poly.h:15-76, Inc/VMACH.h:17-86, Inc/Surtr.h:113-134, 270-272) and
// forwards the work to the HIP engine through the C ABI of include/surtr_hip.h.
//
// It is what a front-end (the DX12 demo or a headless harness) links instead of Src/Poly.cpp's
// ClipPolyhedron / Src/Surtr.cpp's ApplyFracture.  No geometry is computed here: the types are thin
// views that are flattened to CSR buffers, sent through the C ABI and rebuilt from the result.
#pragma once
#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/surtr_hip.h"

namespace surtr {

struct Vector3 { float x = 0, y = 0, z = 0; Vector3() = default; Vector3(float a, float b, float c) : x(a), y(b), z(c) {} };
struct Plane { float x = 0, y = 0, z = 0, w = 0; };

// The reference throws std::exception (Src/Poly.cpp:258); the C ABI returns codes; this layer rethrows.
struct Error : std::runtime_error
{
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// VertexNormalColor, Inc/Mesh.h:4-13 (36 bytes).
struct VertexNormalColor { float Position[3]; float Normal[3]; float Color[3]; };

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
void InitPolyhedron(Polyhedron& polyhedron, const std::vector<Vector3>& positionVec, const std::vector<std::vector<int>>& neighborVec);
Polyhedron GetBB();                             // Src/Poly.cpp:587-617
void Translate(Polyhedron& polyhedron, const Vector3& v);
void Scale(Polyhedron& polyhedron, const Vector3& v);
std::vector<std::vector<int>> ExtractNeighborFromMesh(std::vector<Vector3>& vertices, std::vector<int>& indices);   // Src/Poly.cpp:128-263
} // namespace Poly

namespace VMACH {
struct PolygonFace                              // Inc/VMACH.h:17-58 (the members the path reads)
{
    std::vector<Vector3> VertexVec;
    Plane FacePlane;
};
struct Polygon3D                                // Inc/VMACH.h:60-86
{
    std::vector<PolygonFace> FaceVec;
};
} // namespace VMACH

// Surtr::LoadModelData (Src/Surtr.cpp:2683-2727) for Wavefront OBJ: positions with x negated, scaled, translated; triangles
// with the winding flipped; identical positions joined (normals are not read: the path does not use them).
void LoadModelData(const std::string& fileName, const Vector3& scale, const Vector3& translate, std::vector<Vector3>& vertices,
                   std::vector<int>& indices);

struct Piece { Poly::Polyhedron Convex, Mesh; };            // Inc/Surtr.h:113-119
struct Compound { std::vector<Piece> PieceVec; };           // Inc/Surtr.h:121-127 (value semantics: no leaks)

struct FragmentRender { std::vector<VertexNormalColor> vertexData; std::vector<uint32_t> indexData; };
struct Fragment { int cell, piece, island; Piece piece_data; FragmentRender render; };

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
    // Pattern from seeds through the canonical Voronoi builder (voro++ replacement).
    static std::vector<VMACH::Polygon3D> GenerateVoronoi(const std::vector<Vector3>& cellPointVec);
    // voro.Scale(scale); voro.Translate(translate) for every cell (Src/Surtr.cpp:1799-1803, 1891-1896).
    void PlacePattern(const Vector3& scale, const Vector3& translate);
    void SetCompound(const Compound& compound);
    // ApplyFracture + Refitting + per-piece ExtractFaces/RenderPolyhedron (Src/Surtr.cpp:2098-2149, 2405-2413, 1436-1447):
    // fragments in cell-major, piece, island order.
    std::vector<Fragment> ApplyFracture(const std::set<int>& outside = {}, bool refit = true, bool render = true,
                                        uint32_t cellBegin = 0, uint32_t cellEnd = 0xFFFFFFFFu);
    // Steps 1-6 of Surtr::PrepareFracture (Src/Surtr.cpp:1750-1785): ICH(limit) face normals -> k-DOP slabs moved out by
    // MaxAxisScale / planeGapInverse -> the 2x bounding box clipped by all of them (the approximate convex hull).
    Poly::Polyhedron BuildACH(const std::vector<Vector3>& vertices, uint32_t ichIncludePointLimit = 20, float achPlaneGapInverse = 2000.f);
    // Surtr::PrepareFracture (Src/Surtr.cpp:1747-1827) from a raw triangle mesh: ACH, neighbour rings, `cellCnt` Voronoi cells
    // placed over the bounding box, one event with refit + extraction.  Returns the pieces of the initial compound.
    std::vector<Fragment> PrepareFracture(std::vector<Vector3>& vertices, std::vector<int>& indices, const std::vector<Vector3>& cellPointVec);
    // Poly::ClipPolyhedron(polyhedron, polygon3D), Src/Poly.cpp:556-566.
    Poly::Polyhedron ClipPolyhedron(const Poly::Polyhedron& polyhedron, const VMACH::Polygon3D& polygon3D);
    Poly::Polyhedron ClipPolyhedron(const Poly::Polyhedron& polyhedron, const std::vector<Plane>& planes);
    surtr_counts LastCounts() const { return counts_; }
    surtr_ctx* Raw() { return ctx_; }

private:
    void check(int rc, const char* what);
    surtr_ctx* ctx_ = nullptr;
    uint32_t n_cells_ = 0, n_pieces_ = 0;
    surtr_counts counts_{};
};

} // namespace surtr
