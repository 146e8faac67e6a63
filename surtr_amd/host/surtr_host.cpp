// surtr_host.cpp -- see surtr_host.hpp.  Flatten -> C ABI -> rebuild; no geometry here.
#include "surtr_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

namespace surtr {

namespace {
struct Flat
{
    std::vector<uint32_t> vert_off{0}, nbr_off{0};
    std::vector<float> pos;
    std::vector<int32_t> nbr;
    void add(const Poly::Polyhedron& p)
    {
        for (const auto& v : p)
        {
            pos.push_back(v.Position.x); pos.push_back(v.Position.y); pos.push_back(v.Position.z);
            for (int a : v.NeighborVertexVec) nbr.push_back(a);
            nbr_off.push_back((uint32_t)nbr.size());
        }
        vert_off.push_back((uint32_t)(pos.size() / 3));
    }
};
Poly::Polyhedron rebuild(const float* pos, const uint32_t* off, const int32_t* nbr, uint32_t v0, uint32_t v1)
{
    Poly::Polyhedron p(v1 - v0);
    for (uint32_t v = v0; v < v1; ++v)
    {
        p[v - v0].Position = Vector3(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
        p[v - v0].NeighborVertexVec.assign(nbr + off[v], nbr + off[v + 1]);
    }
    return p;
}
} // namespace

namespace {
int g_default_device = 0;
std::vector<float> flat_points(const std::vector<Vector3>& v)
{
    std::vector<float> pts; pts.reserve(3 * v.size());
    for (const auto& p : v) { pts.push_back(p.x); pts.push_back(p.y); pts.push_back(p.z); }
    return pts;
}
} // namespace

void SetDefaultDevice(int device) { g_default_device = device; }
FractureEngine& DefaultEngine()
{
    static FractureEngine engine(g_default_device);
    return engine;
}

// ---- VMACH: the cell container ------------------------------------------------------------------------------------
Plane VMACH::ConstructFacePlane(const PolygonFace& f)
{
    if (f.VertexVec.size() < 3) throw Error(SURTR_E_INVALID, "ConstructFacePlane: face with fewer than 3 vertices");
    const float a[3] = {f.VertexVec[0].x, f.VertexVec[0].y, f.VertexVec[0].z}, b[3] = {f.VertexVec[1].x, f.VertexVec[1].y, f.VertexVec[1].z},
                c[3] = {f.VertexVec[2].x, f.VertexVec[2].y, f.VertexVec[2].z};
    float pl[4];
    surtr_plane_from_points(a, b, c, pl);
    return Plane(pl[0], pl[1], pl[2], pl[3]);
}
void VMACH::PolygonFace::ConstructFacePlane()
{
    if (GuaranteeConvex && VertexVec.size() >= 3) { FacePlane = VMACH::ConstructFacePlane(*this); FacePlaneConstructed = true; }
}
void VMACH::PolygonFace::AddVertex(const Vector3& v)
{
    // NearlyEqual: (v1 - v2).Length() < 1e-12 on floats, i.e. practically exact equality (Src/VMACH.cpp:1205, Inc/pch.h:18)
    for (const auto& o : VertexVec)
    {
        const float dx = o.x - v.x, dy = o.y - v.y, dz = o.z - v.z;
        if ((double)(dx * dx + dy * dy + dz * dz) < 1e-24) return;
    }
    VertexVec.push_back(v);
    if (GuaranteeConvex && VertexVec.size() == 3) ConstructFacePlane();
}
Vector3 VMACH::PolygonFace::GetNormal() const
{
    if (!FacePlaneConstructed) throw Error(SURTR_E_STATE, "PolygonFace::GetNormal: the face plane was never built (Src/VMACH.cpp:88-97)");
    return Vector3(FacePlane.x, FacePlane.y, FacePlane.z);      // already unit length
}
void VMACH::Polygon3D::Translate(const Vector3& t)
{
    for (auto& f : FaceVec) { for (auto& v : f.VertexVec) { v.x += t.x; v.y += t.y; v.z += t.z; } f.ConstructFacePlane(); }
}
void VMACH::Polygon3D::Scale(const Vector3& s)
{
    for (auto& f : FaceVec) { for (auto& v : f.VertexVec) { v.x *= s.x; v.y *= s.y; v.z *= s.z; } f.ConstructFacePlane(); }
}

// ---- Poly: free functions with the reference's signatures, on the default engine -------------------------------------
void Poly::InitPolyhedron(Polyhedron& polyhedron, const std::vector<Vector3>& positionVec, const std::vector<std::vector<int>>& neighborVec)
{
    polyhedron.resize(positionVec.size());
    for (size_t i = 0; i < positionVec.size(); ++i) { polyhedron[i].Position = positionVec[i]; polyhedron[i].NeighborVertexVec = neighborVec[i]; }
}

void Poly::Moments(double& zerothMoment, Vector3& firstMoment, const Polyhedron& polyhedron)
{
    Flat in; in.add(polyhedron);
    float cen[3] = {0, 0, 0};
    const int rc = surtr_moments((uint32_t)polyhedron.size(), in.pos.data(), in.nbr_off.data(), in.nbr.data(), &zerothMoment, cen);
    if (rc) throw Error(rc, std::string("Moments: ") + surtr_strerror(rc));
    firstMoment = Vector3(cen[0], cen[1], cen[2]);
}

Poly::Extract* Poly::ExtractFaces(const Polyhedron& polyhedron) { return new Extract(DefaultEngine().ExtractFaces(polyhedron)); }

void Poly::ClipPolyhedron(Polyhedron& polyhedron, const std::vector<Plane>& planes)
{
    if (polyhedron.empty()) return;
    polyhedron = DefaultEngine().ClipPolyhedron(polyhedron, planes);
}
Poly::Polyhedron Poly::ClipPolyhedron(const Polyhedron& polyhedron, const VMACH::Polygon3D& polygon3D)
{
    return DefaultEngine().ClipPolyhedron(polyhedron, polygon3D);
}
void Poly::Transform(Polyhedron& polyhedron, const Matrix& matrix) { polyhedron = DefaultEngine().TransformSolid(polyhedron, matrix); }

void Poly::RenderPolyhedron(std::vector<VertexNormalColor>& vertexData, std::vector<uint32_t>& indexData, const Polyhedron& poly,
                            const Extract* extract, bool isConvex, Vector3 color)
{
    (void)extract;      // == ExtractFaces(poly): the kernels derive the face loops themselves
    const FragmentRender r = DefaultEngine().RenderPolyhedron(poly, isConvex, color);
    const uint32_t vertexOffset = (uint32_t)vertexData.size();
    vertexData.insert(vertexData.end(), r.vertexData.begin(), r.vertexData.end());
    for (uint32_t i : r.indexData) indexData.push_back(vertexOffset + i);
}

// ---- Kdop ------------------------------------------------------------------------------------------------------------
Kdop::KdopContainer::KdopContainer(const std::vector<Vector3>& normalVec)
{
    for (const auto& n : normalVec) ElementVec.emplace_back(n);
}
namespace {
void fill_kdop(Kdop::KdopContainer& K, const std::vector<float>& pts, bool ach, double maxAxisScale, float planeGapInv)
{
    const uint32_t k = (uint32_t)K.ElementVec.size(), n = (uint32_t)(pts.size() / 3);
    if (k == 0 || n == 0) return;
    std::vector<float> nrm, pl(8 * (size_t)k);
    for (const auto& e : K.ElementVec) { nrm.push_back(e.Normal.x); nrm.push_back(e.Normal.y); nrm.push_back(e.Normal.z); }
    const int rc = ach ? surtr_kdop_ach_planes(n, pts.data(), k, nrm.data(), maxAxisScale, planeGapInv, pl.data())
                       : surtr_kdop_planes(n, pts.data(), k, nrm.data(), pl.data());
    if (rc) throw Error(rc, std::string("KdopContainer::Calc: ") + surtr_strerror(rc));
    for (uint32_t j = 0; j < k; ++j)
    {
        K.ElementVec[j].MinPlane = Plane(pl[8 * j], pl[8 * j + 1], pl[8 * j + 2], pl[8 * j + 3]);
        K.ElementVec[j].MaxPlane = Plane(pl[8 * j + 4], pl[8 * j + 5], pl[8 * j + 6], pl[8 * j + 7]);
    }
}
} // namespace
void Kdop::KdopContainer::Calc(const std::vector<Vector3>& vertices, const double& maxAxisScale, const float& planeGapInv)
{
    fill_kdop(*this, flat_points(vertices), true, maxAxisScale, planeGapInv);
}
void Kdop::KdopContainer::Calc(const Poly::Polyhedron& mesh)
{
    std::vector<float> pts;
    for (const auto& v : mesh) { pts.push_back(v.Position.x); pts.push_back(v.Position.y); pts.push_back(v.Position.z); }
    fill_kdop(*this, pts, false, 0.0, 1.f);
}
Poly::Polyhedron Kdop::KdopContainer::ClipWithPolyhedron(const Poly::Polyhedron& polyhedron)
{
    std::vector<Plane> planes;
    for (const auto& e : ElementVec) { planes.push_back(e.MinPlane); planes.push_back(e.MaxPlane); }
    Poly::Polyhedron res = polyhedron;
    Poly::ClipPolyhedron(res, planes);
    return res;
}

std::vector<Vector3> GenerateICHNormal(const std::vector<Vector3>& vertices, int limitCnt)
{
    const std::vector<float> pts = flat_points(vertices);
    uint32_t k = 0;
    int rc = surtr_hull_normals((uint32_t)vertices.size(), pts.data(), (uint32_t)limitCnt, 0, nullptr, &k);
    if (rc) throw Error(rc, std::string("GenerateICHNormal: ") + surtr_strerror(rc));
    std::vector<float> nrm(3 * (size_t)k + 3);
    rc = surtr_hull_normals((uint32_t)vertices.size(), pts.data(), (uint32_t)limitCnt, k, nrm.data(), &k);
    if (rc) throw Error(rc, std::string("GenerateICHNormal: ") + surtr_strerror(rc));
    std::vector<Vector3> out(k);
    for (uint32_t i = 0; i < k; ++i) out[i] = Vector3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    return out;
}

Poly::Polyhedron Poly::GetBB()
{
    static const float P[8][3] = {{-.5f, -.5f, -.5f}, {.5f, -.5f, -.5f}, {.5f, .5f, -.5f}, {-.5f, .5f, -.5f},
                                  {-.5f, -.5f, .5f},  {.5f, -.5f, .5f},  {.5f, .5f, .5f},  {-.5f, .5f, .5f}};
    static const int NB[8][3] = {{1, 4, 3}, {5, 0, 2}, {3, 6, 1}, {7, 2, 0}, {5, 7, 0}, {1, 6, 4}, {5, 2, 7}, {4, 6, 3}};
    Polyhedron p(8);
    for (int i = 0; i < 8; ++i) { p[i].Position = Vector3(P[i][0], P[i][1], P[i][2]); p[i].NeighborVertexVec.assign(NB[i], NB[i] + 3); }
    return p;
}

void Poly::Translate(Polyhedron& polyhedron, const Vector3& v)
{
    for (auto& i : polyhedron) { i.Position.x += v.x; i.Position.y += v.y; i.Position.z += v.z; }
}

void Poly::Scale(Polyhedron& polyhedron, const Vector3& v)
{
    for (auto& i : polyhedron) { i.Position.x *= v.x; i.Position.y *= v.y; i.Position.z *= v.z; }
}

std::vector<std::vector<int>> Poly::ExtractNeighborFromMesh(std::vector<Vector3>& vertices, std::vector<int>& indices)
{
    // on the device (surtr_neighbors_from_mesh_dev: directed-edge hash + one fan walk per vertex; same rings, same refusals)
    const uint32_t nv = (uint32_t)vertices.size(), nt = (uint32_t)(indices.size() / 3);
    std::vector<uint32_t> off(nv + 1);
    std::vector<int32_t> nbr(6 * (size_t)nt + 1);
    std::vector<int32_t> tris(indices.begin(), indices.end());
    const int rc = surtr_neighbors_from_mesh_dev(DefaultEngine().Raw(), nv, nt, tris.data(), off.data(), nbr.data(), nullptr);
    if (rc) throw Error(rc, std::string("ExtractNeighborFromMesh: ") + surtr_strerror(rc));
    std::vector<std::vector<int>> out(nv);
    for (uint32_t v = 0; v < nv; ++v) out[v].assign(nbr.begin() + off[v], nbr.begin() + off[v + 1]);
    return out;
}

FractureEngine::FractureEngine(int device) { check(surtr_create(device, &ctx_), "surtr_create"); }
FractureEngine::~FractureEngine() { surtr_destroy(ctx_); }

void LoadModelData(const std::string& fileName, const Vector3& scale, const Vector3& translate, std::vector<Vector3>& vertices,
                   std::vector<int>& indices)
{
    const float sc[3] = {scale.x, scale.y, scale.z}, tr[3] = {translate.x, translate.y, translate.z};
    uint32_t nv = 0, nt = 0;
    int rc = surtr_read_obj(fileName.c_str(), sc, tr, 0, 0, nullptr, nullptr, &nv, &nt);
    if (rc) throw Error(rc, "LoadModelData: cannot read " + fileName);
    std::vector<float> pos(3 * (size_t)nv + 3); std::vector<int32_t> tri(3 * (size_t)nt + 3);
    rc = surtr_read_obj(fileName.c_str(), sc, tr, nv, nt, pos.data(), tri.data(), &nv, &nt);
    if (rc) throw Error(rc, "LoadModelData: cannot read " + fileName);
    vertices.resize(nv); indices.assign(tri.begin(), tri.begin() + 3 * (size_t)nt);
    for (uint32_t v = 0; v < nv; ++v) vertices[v] = Vector3(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]);
}

Poly::Polyhedron FractureEngine::BuildACH(const std::vector<Vector3>& vertices, uint32_t ichIncludePointLimit, float achPlaneGapInverse)
{
    if (vertices.empty()) throw Error(SURTR_E_INVALID, "BuildACH: no vertices");
    std::vector<float> pts; pts.reserve(3 * vertices.size());
    Vector3 lo = vertices[0], hi = vertices[0];
    for (const auto& p : vertices)
    {
        pts.push_back(p.x); pts.push_back(p.y); pts.push_back(p.z);
        lo.x = std::min(lo.x, p.x); hi.x = std::max(hi.x, p.x); lo.y = std::min(lo.y, p.y); hi.y = std::max(hi.y, p.y);
        lo.z = std::min(lo.z, p.z); hi.z = std::max(hi.z, p.z);
    }
    // steps 1-2: ICH face normals
    std::vector<float> nrm(3 * (size_t)(2 * ichIncludePointLimit + 8));
    uint32_t k = 0;
    check(surtr_hull_normals((uint32_t)vertices.size(), pts.data(), ichIncludePointLimit, (uint32_t)(nrm.size() / 3), nrm.data(), &k), "surtr_hull_normals");
    // steps 3-4: bounding box, k-DOP min/max planes
    const double maxAxis = std::max(std::max((double)hi.x - lo.x, (double)hi.y - lo.y), (double)hi.z - lo.z);
    std::vector<float> pl(8 * (size_t)k);
    check(surtr_kdop_ach_planes((uint32_t)vertices.size(), pts.data(), k, nrm.data(), maxAxis, achPlaneGapInverse, pl.data()), "surtr_kdop_ach_planes");
    // steps 5-6: 2x box, clipped by every plane
    Poly::Polyhedron box = Poly::GetBB();
    Poly::Scale(box, Vector3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z)); Poly::Scale(box, Vector3(2, 2, 2));
    Poly::Translate(box, Vector3((float)(((double)hi.x + lo.x) / 2.0), (float)(((double)hi.y + lo.y) / 2.0), (float)(((double)hi.z + lo.z) / 2.0)));
    std::vector<Plane> planes(2 * (size_t)k);
    for (size_t i = 0; i < planes.size(); ++i) { planes[i].x = pl[4 * i]; planes[i].y = pl[4 * i + 1]; planes[i].z = pl[4 * i + 2]; planes[i].w = pl[4 * i + 3]; }
    return ClipPolyhedron(box, planes);
}

std::vector<Fragment> FractureEngine::PrepareFracture(std::vector<Vector3>& vertices, std::vector<int>& indices, const std::vector<Vector3>& cellPointVec)
{
    Piece piece;
    piece.Convex = BuildACH(vertices);
    Poly::InitPolyhedron(piece.Mesh, vertices, Poly::ExtractNeighborFromMesh(vertices, indices));      // step 7
    Vector3 lo = vertices[0], hi = vertices[0];
    for (const auto& p : vertices)
    {
        lo.x = std::min(lo.x, p.x); hi.x = std::max(hi.x, p.x); lo.y = std::min(lo.y, p.y); hi.y = std::max(hi.y, p.y);
        lo.z = std::min(lo.z, p.z); hi.z = std::max(hi.z, p.z);
    }
    SetPattern(GenerateVoronoi(cellPointVec));                                                          // step 8
    PlacePattern(Vector3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z),
                 Vector3((float)(((double)hi.x + lo.x) / 2.0), (float)(((double)hi.y + lo.y) / 2.0), (float)(((double)hi.z + lo.z) / 2.0)));
    Compound comp; comp.PieceVec.push_back(piece);
    SetCompound(comp);
    return ApplyFracture();                                                                             // step 10 (+ Refitting, SetExtract)
}

void FractureEngine::check(int rc, const char* what)
{
    if (rc) throw Error(rc, std::string(what) + ": " + surtr_strerror(rc) + " " + (ctx_ ? surtr_last_error(ctx_) : ""));
}

void FractureEngine::SetPattern(const std::vector<VMACH::Polygon3D>& voroPolyVec)
{
    std::vector<uint32_t> face_off{0};
    std::vector<float> v012;
    for (const auto& cell : voroPolyVec)
    {
        for (const auto& f : cell.FaceVec)
        {
            if (f.VertexVec.size() < 3) throw Error(SURTR_E_INVALID, "face with fewer than 3 vertices");
            for (int k = 0; k < 3; ++k) { v012.push_back(f.VertexVec[k].x); v012.push_back(f.VertexVec[k].y); v012.push_back(f.VertexVec[k].z); }
        }
        face_off.push_back((uint32_t)(v012.size() / 9));
    }
    n_cells_ = (uint32_t)voroPolyVec.size();
    check(surtr_upload_pattern(ctx_, n_cells_, face_off.data(), v012.data()), "surtr_upload_pattern");
}

std::vector<VMACH::Polygon3D> FractureEngine::GenerateVoronoiHost(const std::vector<Vector3>& cellPointVec)
{
    const uint32_t n = (uint32_t)cellPointVec.size();
    std::vector<double> seeds;
    for (const auto& s : cellPointVec) { seeds.push_back(s.x); seeds.push_back(s.y); seeds.push_back(s.z); }
    uint32_t nf = 0, nfv = 0;
    int rc = surtr_voronoi_cells(n, seeds.data(), &nf, &nfv, nullptr, nullptr, nullptr, nullptr);
    if (rc) throw Error(rc, "surtr_voronoi_cells");
    std::vector<uint32_t> cfo(n + 1), fvo(nf + 1);
    std::vector<int32_t> gen(nf);
    std::vector<double> verts(3 * (size_t)nfv);
    rc = surtr_voronoi_cells(n, seeds.data(), &nf, &nfv, cfo.data(), gen.data(), fvo.data(), verts.data());
    if (rc) throw Error(rc, "surtr_voronoi_cells");
    std::vector<VMACH::Polygon3D> out(n);
    for (uint32_t c = 0; c < n; ++c)
        for (uint32_t f = cfo[c]; f < cfo[c + 1]; ++f)
        {
            VMACH::PolygonFace face(true);
            for (uint32_t v = fvo[f]; v < fvo[f + 1]; ++v)
                face.VertexVec.emplace_back((float)verts[3 * v], (float)verts[3 * v + 1], (float)verts[3 * v + 2]);
            face.ConstructFacePlane();
            out[c].FaceVec.push_back(face);
        }
    return out;
}

std::vector<VMACH::Polygon3D> FractureEngine::GenerateVoronoi(const std::vector<Vector3>& cellPointVec)
{
    const uint32_t n = (uint32_t)cellPointVec.size();
    std::vector<double> seeds;
    for (const auto& s : cellPointVec) { seeds.push_back(s.x); seeds.push_back(s.y); seeds.push_back(s.z); }
    const uint32_t group_off[2] = {0u, n};
    uint32_t nf = 0, nfv = 0;
    check(surtr_build_cells(ctx_, 1, group_off, seeds.data(), &nf, &nfv), "surtr_build_cells");
    n_cells_ = n;                                       // the cells are this engine's pattern now
    std::vector<uint32_t> cfo(n + 1), fvo(nf + 1);
    std::vector<double> verts(3 * (size_t)nfv + 3);
    check(surtr_download_cells(ctx_, cfo.data(), nullptr, fvo.data(), verts.data(), nullptr), "surtr_download_cells");
    std::vector<VMACH::Polygon3D> out(n);
    for (uint32_t c = 0; c < n; ++c)
        for (uint32_t f = cfo[c]; f < cfo[c + 1]; ++f)
        {
            VMACH::PolygonFace face(true);
            for (uint32_t v = fvo[f]; v < fvo[f + 1]; ++v)
                face.VertexVec.emplace_back((float)verts[3 * v], (float)verts[3 * v + 1], (float)verts[3 * v + 2]);
            face.ConstructFacePlane();
            out[c].FaceVec.push_back(face);
        }
    return out;
}

std::vector<VMACH::Polygon3D> FractureEngine::GenerateVoronoi(int cellCnt, int seed)
{
    // Src/Surtr.cpp:1984-2001 (libstdc++'s distributions, like the oracle's and surtr_amd/scenes.py)
    std::mt19937 gen((unsigned)seed);
    std::uniform_real_distribution<double> uniformDist(-0.5, 0.5);
    std::vector<Vector3> cellPointVec;
    for (int i = 0; i < cellCnt; ++i)
    {
        const double x = uniformDist(gen), y = uniformDist(gen), z = uniformDist(gen);
        cellPointVec.emplace_back((float)x, (float)y, (float)z);
    }
    return GenerateVoronoi(cellPointVec);
}

std::vector<VMACH::Polygon3D> FractureEngine::GenerateFracturePattern(int cellCount, double mean, int seed)
{
    // Src/Surtr.cpp:2072-2096
    std::mt19937 gen((unsigned)seed);
    std::uniform_real_distribution<double> directionUniformDist(-1.0, 1.0);
    std::exponential_distribution<double> lengthExpDist(1.0 / mean);
    std::vector<Vector3> cellPointVec;
    for (int i = 0; i < cellCount; ++i)
    {
        const double len = std::max(std::min(lengthExpDist(gen), 0.5), 1e-12);
        const double x = directionUniformDist(gen), y = directionUniformDist(gen), z = directionUniformDist(gen);
        Vector3 v((float)x, (float)y, (float)z);
        const float t = v.x * v.x + v.y * v.y;
        const float l = std::sqrt(t + v.z * v.z);                                 // Vector3::Normalize: divide by sqrt(dot)
        v.x = v.x / l; v.y = v.y / l; v.z = v.z / l;
        const float fl = (float)len;                                              // v *= len (XMVectorScale by a float)
        v.x *= fl; v.y *= fl; v.z *= fl;
        cellPointVec.push_back(v);
    }
    return GenerateVoronoi(cellPointVec);
}

void FractureEngine::PlacePattern(const Vector3& scale, const Vector3& translate)
{
    const float s[3] = {scale.x, scale.y, scale.z}, t[3] = {translate.x, translate.y, translate.z};
    check(surtr_place_cells(ctx_, s, t), "surtr_place_cells");
}

void FractureEngine::SetCompound(const Compound& compound)
{
    Flat m, c;
    for (const auto& p : compound.PieceVec) { m.add(p.Mesh); c.add(p.Convex); }
    n_pieces_ = (uint32_t)compound.PieceVec.size();
    check(surtr_upload_pieces(ctx_, n_pieces_, m.vert_off.data(), m.pos.data(), m.nbr_off.data(), m.nbr.data(),
                              c.vert_off.data(), c.pos.data(), c.nbr_off.data(), c.nbr.data()), "surtr_upload_pieces");
}

std::vector<Fragment> FractureEngine::ApplyFracture(const std::set<int>& outside, bool refit, bool render, uint32_t cellBegin, uint32_t cellEnd)
{
    std::vector<uint8_t> mask(n_pieces_, 0);
    for (int o : outside) if (o >= 0 && (uint32_t)o < n_pieces_) mask[o] = 1;
    if (cellEnd == 0xFFFFFFFFu) cellEnd = n_cells_;
    const uint32_t flags = (refit ? SURTR_EVT_REFIT : 0u) | (render ? SURTR_EVT_RENDER : 0u);
    check(surtr_fracture_event(ctx_, cellBegin, cellEnd, outside.empty() ? nullptr : mask.data(), flags, &counts_), "surtr_fracture_event");
    std::vector<Fragment> out = download_fragments(render);
    report_flagged(out, cellBegin, cellEnd, "ApplyFracture");
    return out;
}

// What the event flagged (surtr_counts::n_failed): the pairs without a fragment from surtr_pair_status (pair = (cell - cellBegin)
// * pieces + piece, the order of the reference's double loop, Src/Surtr.cpp:1457-1504), the fragments from their status word.
void FractureEngine::report_flagged(const std::vector<Fragment>& frags, uint32_t cellBegin, uint32_t cellEnd, const char* what)
{
    flagged_ = FlaggedUnits();
    flagged_.n_failed = counts_.n_failed;
    if (counts_.n_failed == 0) return;
    for (size_t f = 0; f < frags.size(); ++f) if (frags[f].status != 0) flagged_.fragments.push_back((uint32_t)f);
    const uint32_t n_pairs = (cellEnd - cellBegin) * n_pieces_;
    std::vector<uint32_t> st(n_pairs);
    if (n_pairs != 0 && surtr_pair_status(ctx_, n_pairs, st.data()) == SURTR_OK)
        for (uint32_t p = 0; p < n_pairs; ++p)
            if (st[p] != 0) flagged_.pairs.emplace_back((int)(cellBegin + p / n_pieces_), (int)(p % n_pieces_));
    if (!allow_flagged_)
        throw Error(SURTR_E_TOPOLOGY, std::string(what) + ": " + std::to_string(flagged_.pairs.size()) + " (cell, piece) pair(s) without a valid clip and " +
                    std::to_string(flagged_.fragments.size()) + " fragment(s) flagged (input outside the reference's domain; AllowFlagged(true) returns the rest)");
}

std::vector<Fragment> FractureEngine::download_fragments(bool render)
{
    const surtr_counts& c = counts_;
    std::vector<int32_t> ids(3 * (size_t)c.n_frag), mnbr(c.mesh_nbrs), cnbr(c.conv_nbrs);
    std::vector<uint32_t> mvo(c.n_frag + 1), mno(c.mesh_verts + 1), cvo(c.n_frag + 1), cno(c.conv_verts + 1), ioff(c.n_frag + 1), idx(c.n_idx);
    std::vector<float> mpos(3 * (size_t)c.mesh_verts), cpos(3 * (size_t)c.conv_verts), vnc(9 * (size_t)c.mesh_verts);
    std::vector<uint32_t> fstat(c.n_frag);
    surtr_fragments fr{ids.data(), mvo.data(), mpos.data(), mno.data(), mnbr.data(), cvo.data(), cpos.data(), cno.data(), cnbr.data(),
                       vnc.data(), ioff.data(), idx.data(), fstat.data()};
    check(surtr_event_download(ctx_, &fr), "surtr_event_download");
    std::vector<Fragment> out(c.n_frag);
    for (uint32_t f = 0; f < c.n_frag; ++f)
    {
        Fragment& F = out[f];
        F.cell = ids[3 * f]; F.piece = ids[3 * f + 1]; F.island = ids[3 * f + 2]; F.status = (int)fstat[f];
        F.piece_data.Mesh = rebuild(mpos.data(), mno.data(), mnbr.data(), mvo[f], mvo[f + 1]);
        // rings are fragment-local already; offsets are global over the vertex array
        F.piece_data.Convex = rebuild(cpos.data(), cno.data(), cnbr.data(), cvo[f], cvo[f + 1]);
        if (render && c.n_idx != 0)
        {
            F.render.vertexData.resize(mvo[f + 1] - mvo[f]);
            std::memcpy(F.render.vertexData.data(), vnc.data() + 9 * (size_t)mvo[f], sizeof(VertexNormalColor) * F.render.vertexData.size());
            F.render.indexData.assign(idx.begin() + ioff[f], idx.begin() + ioff[f + 1]);
        }
    }
    return out;
}

Poly::Polyhedron FractureEngine::ClipPolyhedron(const Poly::Polyhedron& polyhedron, const std::vector<Plane>& planes)
{
    if (polyhedron.size() < 4) return Poly::Polyhedron();      // fewer than four vertices: the reference's answer is the empty solid (Src/Poly.cpp:497-499)
    Flat in; in.add(polyhedron);
    std::vector<float> pl;
    for (const auto& p : planes) { pl.push_back(p.x); pl.push_back(p.y); pl.push_back(p.z); pl.push_back(p.w); }
    uint32_t nv = 0, nh = 0;
    const uint32_t V = (uint32_t)polyhedron.size();
    check(surtr_clip_polyhedron(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), (uint32_t)planes.size(), pl.data(), &nv, &nh,
                                nullptr, nullptr, nullptr), "surtr_clip_polyhedron");
    std::vector<float> pos(3 * (size_t)nv + 3); std::vector<uint32_t> off(nv + 1); std::vector<int32_t> nbr(nh + 1);
    check(surtr_clip_polyhedron(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), (uint32_t)planes.size(), pl.data(), &nv, &nh,
                                pos.data(), off.data(), nbr.data()), "surtr_clip_polyhedron");
    return rebuild(pos.data(), off.data(), nbr.data(), 0, nv);
}

Poly::Polyhedron FractureEngine::ClipPolyhedron(const Poly::Polyhedron& polyhedron, const VMACH::Polygon3D& polygon3D)
{
    // The reference keeps PolygonFace::FacePlane current through ConstructFacePlane on AddVertex / Scale / Translate
    // (Src/VMACH.cpp:302-310); this layer's PolygonFace is a plain struct, so the plane is rebuilt here from the first three
    // vertices with the arithmetic of k_place_cells (Plane(p0,p1,p2), normalised: SimpleMath.inl:2773-2780).
    std::vector<Plane> planes;
    for (const auto& f : polygon3D.FaceVec)
    {
        if (f.VertexVec.size() < 3) throw Error(SURTR_E_INVALID, "ClipPolyhedron: face with fewer than 3 vertices");
        planes.push_back(VMACH::ConstructFacePlane(f));
    }
    return ClipPolyhedron(polyhedron, planes);
}

void FractureEngine::TransformCompound(const std::vector<Matrix>& worldMatrices)
{
    if (worldMatrices.size() != n_pieces_) throw Error(SURTR_E_INVALID, "TransformCompound: one matrix per piece");
    std::vector<float> w; w.reserve(16 * worldMatrices.size());
    for (const auto& m : worldMatrices) w.insert(w.end(), m.m, m.m + 16);
    check(surtr_transform_pieces(ctx_, n_pieces_, w.data()), "surtr_transform_pieces");
}

uint32_t FractureEngine::CompoundFromLastEvent(const std::vector<uint8_t>& keep)
{
    uint32_t n = 0;
    check(surtr_pieces_from_event(ctx_, keep.empty() ? nullptr : keep.data(), &n), "surtr_pieces_from_event");
    n_pieces_ = n;
    return n;
}

void FractureEngine::Refitting(std::vector<Piece>& pieceVec)
{
    if (pieceVec.empty()) return;
    Flat m, c;
    for (const auto& p : pieceVec) { m.add(p.Mesh); c.add(p.Convex); }
    const uint32_t n = (uint32_t)pieceVec.size();
    check(surtr_load_fragments(ctx_, n, m.vert_off.data(), m.pos.data(), m.nbr_off.data(), m.nbr.data(),
                               c.vert_off.data(), c.pos.data(), c.nbr_off.data(), c.nbr.data(), nullptr), "surtr_load_fragments");
    check(surtr_event_refit(ctx_), "surtr_event_refit");
    check(surtr_event_counts(ctx_, &counts_), "surtr_event_counts");
    std::vector<uint32_t> cvo(n + 1), cno(counts_.conv_verts + 1);
    std::vector<float> cpos(3 * (size_t)counts_.conv_verts + 3);
    std::vector<int32_t> cnbr(counts_.conv_nbrs + 1);
    surtr_fragments fr{};
    fr.conv_vert_off = cvo.data(); fr.conv_pos = cpos.data(); fr.conv_nbr_off = cno.data(); fr.conv_nbr = cnbr.data();
    check(surtr_event_download(ctx_, &fr), "surtr_event_download");
    for (uint32_t k = 0; k < n; ++k) pieceVec[k].Convex = rebuild(cpos.data(), cno.data(), cnbr.data(), cvo[k], cvo[k + 1]);
}

Poly::Polyhedron FractureEngine::RefitSolid(const Poly::Polyhedron& mesh, const Poly::Polyhedron& convex)
{
    Flat m, c; m.add(mesh); c.add(convex);
    uint32_t nv = 0, nh = 0;
    const uint32_t MV = (uint32_t)mesh.size(), CV = (uint32_t)convex.size();
    check(surtr_refit_solid(ctx_, MV, m.pos.data(), m.nbr_off.data(), m.nbr.data(), CV, c.pos.data(), c.nbr_off.data(), c.nbr.data(),
                            &nv, &nh, nullptr, nullptr, nullptr), "surtr_refit_solid");
    std::vector<float> pos(3 * (size_t)nv + 3); std::vector<uint32_t> off(nv + 1); std::vector<int32_t> nbr(nh + 1);
    check(surtr_refit_solid(ctx_, MV, m.pos.data(), m.nbr_off.data(), m.nbr.data(), CV, c.pos.data(), c.nbr_off.data(), c.nbr.data(),
                            &nv, &nh, pos.data(), off.data(), nbr.data()), "surtr_refit_solid");
    return rebuild(pos.data(), off.data(), nbr.data(), 0, nv);
}

Poly::Extract FractureEngine::ExtractFaces(const Poly::Polyhedron& polyhedron)
{
    Flat in; in.add(polyhedron);
    const uint32_t V = (uint32_t)polyhedron.size();
    uint32_t nf = 0, ni = 0;
    check(surtr_extract_faces(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), &nf, &ni, nullptr, nullptr), "surtr_extract_faces");
    std::vector<uint32_t> fo(nf + 1); std::vector<int32_t> fi(ni + 1);
    check(surtr_extract_faces(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), &nf, &ni, fo.data(), fi.data()), "surtr_extract_faces");
    Poly::Extract out(nf);
    for (uint32_t f = 0; f < nf; ++f) out[f].assign(fi.begin() + fo[f], fi.begin() + fo[f + 1]);
    return out;
}

FragmentRender FractureEngine::RenderPolyhedron(const Poly::Polyhedron& poly, bool isConvex, const Vector3& color)
{
    Flat in; in.add(poly);
    const uint32_t V = (uint32_t)poly.size();
    const float col[3] = {color.x, color.y, color.z};
    uint32_t ni = 0;
    check(surtr_triangulate(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), isConvex ? 1 : 0, col, nullptr, &ni, nullptr), "surtr_triangulate");
    FragmentRender r;
    r.vertexData.resize(V); r.indexData.resize(ni);
    std::vector<uint32_t> idx(ni + 1);
    check(surtr_triangulate(ctx_, V, in.pos.data(), in.nbr_off.data(), in.nbr.data(), isConvex ? 1 : 0, col, (float*)r.vertexData.data(), &ni, idx.data()),
          "surtr_triangulate");
    std::copy(idx.begin(), idx.begin() + ni, r.indexData.begin());
    return r;
}

Poly::Polyhedron FractureEngine::TransformSolid(const Poly::Polyhedron& polyhedron, const Matrix& matrix)
{
    // One lone host polyhedron: the arithmetic of k_transform (XMVector3TransformCoord(v, XMMatrixTranspose(M)): per row
    // x*m0 + (y*m1 + (z*m2 + m3)), multiply-then-add -- this file is built with -ffp-contract=off -- then the divide by w) here on
    // the host; the resident compound of this engine is not touched.  Pieces of an event are moved on the device
    // (TransformCompound).
    Poly::Polyhedron out = polyhedron;
    const float* m = matrix.m;
    for (auto& v : out)
    {
        const float x = v.Position.x, y = v.Position.y, z = v.Position.z;
        float r[4];
        for (int c = 0; c < 4; ++c)
        {
            float t = z * m[4 * c + 2] + m[4 * c + 3];
            t = y * m[4 * c + 1] + t;
            r[c] = x * m[4 * c] + t;
        }
        v.Position = Vector3(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
    }
    return out;
}

// ---- the step after the event: regrouping, and DoFracture as a whole ---------------------------------------------------
bool ConvexOutOfSphere(const Poly::Polyhedron& polyhedron, const Poly::Extract* extract, const std::vector<Vector3>& spherePointCloud,
                       const Vector3 origin, const float radius)
{
    (void)extract;       // == ExtractFaces(polyhedron): derived again behind the C ABI
    Flat in; in.add(polyhedron);
    const std::vector<float> cloud = flat_points(spherePointCloud);
    const float org[3] = {origin.x, origin.y, origin.z};
    int out = 0;
    const int rc = surtr_convex_out_of_sphere((uint32_t)polyhedron.size(), in.pos.data(), in.nbr_off.data(), in.nbr.data(),
                                              (uint32_t)spherePointCloud.size(), cloud.data(), org, radius, &out);
    if (rc) throw Error(rc, std::string("ConvexOutOfSphere: ") + surtr_strerror(rc));
    return out != 0;
}

void MergeOutOfImpact(CompoundInfo& compoundInfo, const std::vector<Vector3>& spherePointCloud, const Vector3 origin, const float radius)
{
    // the set bookkeeping of Src/Surtr.cpp:2368-2403; the geometry is ConvexOutOfSphere
    for (size_t i = 1; i < compoundInfo.CompoundBind.size(); ++i)
    {
        auto& local = compoundInfo.CompoundBind[i];
        std::set<int> outside;
        for (const int c : local)
            if (ConvexOutOfSphere(compoundInfo.PieceVec[(size_t)c].Convex, nullptr, spherePointCloud, origin, radius)) outside.insert(c);
        if (!outside.empty())
        {
            for (const int c : outside) local.erase(c);
            compoundInfo.CompoundBind[0].insert(outside.begin(), outside.end());
        }
    }
    if (!compoundInfo.CompoundBind.empty())
        compoundInfo.CompoundBind.erase(std::remove_if(std::next(compoundInfo.CompoundBind.begin()), compoundInfo.CompoundBind.end(),
                                                       [](const std::set<int>& local) { return local.empty(); }),
                                        compoundInfo.CompoundBind.end());
}

void HandleConvexIsland(CompoundInfo& compoundInfo)
{
    // surtr_regroup works on pieces numbered so that compound 0 comes first and every other compound is a run of consecutive
    // pieces: renumber, regroup without the merge step, number back.  Its result order is the reference's: the first group of
    // a split compound stays where the compound was, the others are appended (Src/Surtr.cpp:2356-2365).
    auto& bind = compoundInfo.CompoundBind;
    if (bind.empty()) return;
    std::vector<int> order;                       // new index -> old piece index
    std::vector<int32_t> cell;
    for (size_t i = 0; i < bind.size(); ++i)
        for (const int c : bind[i]) { order.push_back(c); cell.push_back(i == 0 ? -1 : (int32_t)i); }
    const uint32_t n = (uint32_t)order.size(), n0 = (uint32_t)bind[0].size();
    if (n == 0) return;
    Flat cv;
    for (const int c : order) cv.add(compoundInfo.PieceVec[(size_t)c].Convex);
    std::vector<uint32_t> off(n + 2);
    std::vector<int32_t> piece(n);
    uint32_t nc = 0;
    const float org[3] = {0.f, 0.f, 0.f};
    const int rc = surtr_regroup(n, n0, cell.data(), cv.vert_off.data(), cv.pos.data(), cv.nbr_off.data(), cv.nbr.data(), 0, 0, nullptr, org, 0.f,
                                 &nc, off.data(), piece.data());
    if (rc) throw Error(rc, std::string("HandleConvexIsland: ") + surtr_strerror(rc));
    std::vector<std::set<int>> out(nc);
    for (uint32_t c = 0; c < nc; ++c)
        for (uint32_t k = off[c]; k < off[c + 1]; ++k) out[c].insert(order[(size_t)piece[k]]);
    bind.swap(out);
}

std::vector<Compound> FractureEngine::DoFracture(const Compound& targetCompound, float maxAxisScale, const FractureArgs& args,
                                                 const std::vector<Vector3>& spherePointCloud, FractureTrace* trace)
{
    // Scale + alignment of the pattern (Src/Surtr.cpp:1890-1896): every cell scaled by MaxAxisScale * 2, moved to the impact
    const float s2 = maxAxisScale * 2.f;
    PlacePattern(Vector3(s2, s2, s2), args.ImpactPosition);
    // the sphere point cloud: v *= ImpactRadius; v += ImpactPosition (:1911-1915)
    std::vector<Vector3> cloud = spherePointCloud;
    for (auto& v : cloud)
    {
        v.x *= args.ImpactRadius; v.y *= args.ImpactRadius; v.z *= args.ImpactRadius;
        v.x += args.ImpactPosition.x; v.y += args.ImpactPosition.y; v.z += args.ImpactPosition.z;
    }
    SetCompound(targetCompound);
    // ApplyFracture (:2098-2149): with PartialFracture the pieces whose Convex is out of the sphere are kept whole (bind 0)
    std::set<int> outside;
    if (args.PartialFracture)
        for (size_t c = 0; c < targetCompound.PieceVec.size(); ++c)
            if (ConvexOutOfSphere(targetCompound.PieceVec[c].Convex, nullptr, cloud, args.ImpactPosition, args.ImpactRadius)) outside.insert((int)c);
    std::vector<uint8_t> mask(n_pieces_, 0);
    for (const int o : outside) mask[(size_t)o] = 1;
    check(surtr_fracture_event(ctx_, 0, n_cells_, outside.empty() ? nullptr : mask.data(), 0u, &counts_), "surtr_fracture_event");
    std::vector<Fragment> raw;
    if (trace) raw = download_fragments(false);            // (the un-refitted Convex solids, for whoever wants to check the regrouping)
    // MergeOutOfImpact (PartialFracture) + HandleConvexIsland, on the un-refitted Convex solids where the event left them
    const std::vector<float> fc = flat_points(cloud);
    const float org[3] = {args.ImpactPosition.x, args.ImpactPosition.y, args.ImpactPosition.z};
    uint32_t np = 0, nc = 0;
    check(surtr_event_regroup(ctx_, args.PartialFracture ? 1 : 0, (uint32_t)cloud.size(), fc.data(), org, args.ImpactRadius, &np, &nc, nullptr, nullptr),
          "surtr_event_regroup");
    std::vector<uint32_t> off(np + 2);
    std::vector<int32_t> piece(np + 1);
    check(surtr_event_regroup(ctx_, args.PartialFracture ? 1 : 0, (uint32_t)cloud.size(), fc.data(), org, args.ImpactRadius, &np, &nc, off.data(), piece.data()),
          "surtr_event_regroup");
    // Refitting (:1937-1939), then the pieces: the skipped ones first (ascending), then the fragments in output order
    check(surtr_event_refit(ctx_), "surtr_event_refit");
    check(surtr_event_counts(ctx_, &counts_), "surtr_event_counts");
    const std::vector<Fragment> frags = download_fragments(false);
    report_flagged(frags, 0, n_cells_, "DoFracture");
    const uint32_t n0 = (uint32_t)outside.size();
    if (np != n0 + frags.size()) throw Error(SURTR_E_STATE, "DoFracture: piece count of the regrouping does not match the event");
    std::vector<const Piece*> all;
    for (const int o : outside) all.push_back(&targetCompound.PieceVec[(size_t)o]);
    for (const auto& f : frags) all.push_back(&f.piece_data);
    std::vector<Compound> result(nc);
    for (uint32_t c = 0; c < nc; ++c)
        for (uint32_t k = off[c]; k < off[c + 1]; ++k) result[c].PieceVec.push_back(*all[(size_t)piece[k]]);
    if (trace)
    {
        trace->nOutside = n0; trace->Convex.clear(); trace->PieceCell.clear(); trace->CompoundBind.assign(nc, {});
        for (const int o : outside) { trace->Convex.push_back(targetCompound.PieceVec[(size_t)o].Convex); trace->PieceCell.push_back(-1); }
        for (const auto& f : raw) { trace->Convex.push_back(f.piece_data.Convex); trace->PieceCell.push_back(f.cell); }
        for (uint32_t c = 0; c < nc; ++c) for (uint32_t k = off[c]; k < off[c + 1]; ++k) trace->CompoundBind[c].insert(piece[k]);
    }
    return result;
}

// ---- the per-Piece tasks with the reference's signatures (Inc/Surtr.h:270-272) --------------------------------------------
std::vector<Piece*> FractureTask(const VMACH::Polygon3D& voroPoly, const std::vector<Piece*>& targetPieceVec, const std::set<int>& outside)
{
    // m_fractureTask (Src/Surtr.cpp:1457-1504) for one cell: clip Convex, then Mesh, split islands; no refit here
    FractureEngine& eng = DefaultEngine();
    Compound comp;
    for (const Piece* p : targetPieceVec) comp.PieceVec.push_back(*p);
    eng.SetPattern(std::vector<VMACH::Polygon3D>{voroPoly});
    eng.PlacePattern(Vector3(1.f, 1.f, 1.f), Vector3(0.f, 0.f, 0.f));      // the cell comes placed: x * 1 + 0 leaves every float as it is
    eng.SetCompound(comp);
    const std::vector<Fragment> frags = eng.ApplyFracture(outside, false, false);
    std::vector<Piece*> out;
    for (const auto& f : frags) out.push_back(new Piece(f.piece_data.Convex, f.piece_data.Mesh));
    return out;
}

void RefittingTask(Piece* piece)
{
    // m_refittingTask (Src/Surtr.cpp:1449-1455)
    piece->Convex = DefaultEngine().RefitSolid(piece->Mesh, piece->Convex);
}

InitCompoundResult InitCompoundTask(const Piece* piece, const Poly::Extract* extract, bool renderConvex)
{
    // m_initCompoundTask (Src/Surtr.cpp:1436-1447): the points PxCreateConvexMesh cooks (:2531-2553) and the buffers
    // DynamicMesh::UpdateMeshData uploads -- RenderPolyhedron of the Convex (as convex) or of the Mesh (isConvex = false)
    (void)extract;
    InitCompoundResult r;
    for (const auto& v : piece->Convex) r.ConvexPoints.push_back(v.Position);
    r.Mesh = DefaultEngine().RenderPolyhedron(renderConvex ? piece->Convex : piece->Mesh, renderConvex, Vector3(0.25f, 0.25f, 0.25f));
    return r;
}

} // namespace surtr
