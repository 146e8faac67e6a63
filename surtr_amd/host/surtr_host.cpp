// surtr_host.cpp -- see surtr_host.hpp.  Flatten -> C ABI -> rebuild; no geometry here.
#include "surtr_host.hpp"

#include <algorithm>
#include <cstring>

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

void Poly::InitPolyhedron(Polyhedron& polyhedron, const std::vector<Vector3>& positionVec, const std::vector<std::vector<int>>& neighborVec)
{
    polyhedron.resize(positionVec.size());
    for (size_t i = 0; i < positionVec.size(); ++i) { polyhedron[i].Position = positionVec[i]; polyhedron[i].NeighborVertexVec = neighborVec[i]; }
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
    const uint32_t nv = (uint32_t)vertices.size(), nt = (uint32_t)(indices.size() / 3);
    std::vector<uint32_t> off(nv + 1);
    std::vector<int32_t> nbr(6 * (size_t)nt + 1);
    std::vector<int32_t> tris(indices.begin(), indices.end());
    const int rc = surtr_neighbors_from_mesh(nv, nt, tris.data(), off.data(), nbr.data());
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

std::vector<VMACH::Polygon3D> FractureEngine::GenerateVoronoi(const std::vector<Vector3>& cellPointVec)
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
            VMACH::PolygonFace face;
            for (uint32_t v = fvo[f]; v < fvo[f + 1]; ++v)
                face.VertexVec.emplace_back((float)verts[3 * v], (float)verts[3 * v + 1], (float)verts[3 * v + 2]);
            out[c].FaceVec.push_back(face);
        }
    return out;
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
    const surtr_counts& c = counts_;
    std::vector<int32_t> ids(3 * (size_t)c.n_frag), mnbr(c.mesh_nbrs), cnbr(c.conv_nbrs);
    std::vector<uint32_t> mvo(c.n_frag + 1), mno(c.mesh_verts + 1), cvo(c.n_frag + 1), cno(c.conv_verts + 1), ioff(c.n_frag + 1), idx(c.n_idx);
    std::vector<float> mpos(3 * (size_t)c.mesh_verts), cpos(3 * (size_t)c.conv_verts), vnc(9 * (size_t)c.mesh_verts);
    surtr_fragments fr{ids.data(), mvo.data(), mpos.data(), mno.data(), mnbr.data(), cvo.data(), cpos.data(), cno.data(), cnbr.data(),
                       vnc.data(), ioff.data(), idx.data()};
    check(surtr_event_download(ctx_, &fr), "surtr_event_download");
    std::vector<Fragment> out(c.n_frag);
    for (uint32_t f = 0; f < c.n_frag; ++f)
    {
        Fragment& F = out[f];
        F.cell = ids[3 * f]; F.piece = ids[3 * f + 1]; F.island = ids[3 * f + 2];
        F.piece_data.Mesh = rebuild(mpos.data(), mno.data(), mnbr.data(), mvo[f], mvo[f + 1]);
        // rings are fragment-local already; offsets are global over the vertex array
        F.piece_data.Convex = rebuild(cpos.data(), cno.data(), cnbr.data(), cvo[f], cvo[f + 1]);
        if (render)
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
    std::vector<Plane> planes;
    for (const auto& f : polygon3D.FaceVec) planes.push_back(f.FacePlane);
    return ClipPolyhedron(polyhedron, planes);
}

} // namespace surtr
