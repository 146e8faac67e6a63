// harness.cpp -- headless replacement of the Win32/DX12 shell for the fracture-event path:
// builds a synthetic closed mesh, a Voronoi pattern, runs ONE fracture event on the GPU and prints a
// JSON summary (optionally writes the fragments as an OBJ).  Usage:
//   surtr_harness [--mesh cube|torus | --obj-in mesh.obj [--scale S]] [--cells N] [--nu A --nv B] [--ach] [--obj out.obj]
// --ach runs Surtr::PrepareFracture end to end (ACH convex instead of the plain 2x box).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>

#include "surtr_host.hpp"

using namespace surtr;

static void make_cube(std::vector<Vector3>& v, std::vector<int>& t)
{
    const float s = 3.f;   // Src/Surtr.cpp:1403
    const float P[8][3] = {{-1, -1, 1}, {-1, 1, 1}, {-1, -1, -1}, {-1, 1, -1}, {1, -1, 1}, {1, 1, 1}, {1, -1, -1}, {1, 1, -1}};
    const int Q[6][4] = {{0, 1, 3, 2}, {2, 3, 7, 6}, {6, 7, 5, 4}, {4, 5, 1, 0}, {2, 6, 4, 0}, {7, 3, 1, 5}};
    for (auto& p : P) v.emplace_back(p[0] * s, p[1] * s, p[2] * s);
    // same triangles as surtr_amd/meshgen.py (quads split (a,b,c),(a,c,d); outward winding)
    for (auto& q : Q) { t.insert(t.end(), {q[0], q[1], q[2]}); t.insert(t.end(), {q[0], q[2], q[3]}); }
}

static void make_torus(int nu, int nv, std::vector<Vector3>& v, std::vector<int>& t)
{
    const double R = 1.0, r0 = 0.35, PI = 3.14159265358979323846;
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nv; ++j)
        {
            const double U = i * (2.0 * PI / nu), W = j * (2.0 * PI / nv);
            const double r = r0 * (1.0 + 0.25 * std::sin(5.0 * U) * std::cos(3.0 * W));
            v.emplace_back((float)((R + r * std::cos(W)) * std::cos(U)), (float)((R + r * std::cos(W)) * std::sin(U)), (float)(r * std::sin(W)));
        }
    auto id = [&](int i, int j) { return ((i + nu) % nu) * nv + ((j + nv) % nv); };
    std::vector<int> second;
    for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nv; ++j)
        {
            const int a = id(i, j), b = id(i + 1, j), c = id(i + 1, j + 1), d = id(i, j + 1);
            t.insert(t.end(), {a, b, c});
            second.insert(second.end(), {a, c, d});
        }
    t.insert(t.end(), second.begin(), second.end());
}

// --api-dump FILE: runs the reference-named API (Poly::*, Kdop::*, VMACH::*) on fragments of the event and writes inputs and
// results as JSON, so that a test can hold them against the oracle (tests/test_gpu_parity.py::test_cpp_api_surface).
static void js_solid(FILE* f, const char* name, const Poly::Polyhedron& p)
{
    fprintf(f, "\"%s\": {\"pos\": [", name);
    for (size_t i = 0; i < p.size(); ++i) fprintf(f, "%s%.9g, %.9g, %.9g", i ? ", " : "", p[i].Position.x, p[i].Position.y, p[i].Position.z);
    fprintf(f, "], \"off\": [0");
    size_t h = 0;
    for (const auto& v : p) { h += v.NeighborVertexVec.size(); fprintf(f, ", %zu", h); }
    fprintf(f, "], \"nbr\": [");
    bool first = true;
    for (const auto& v : p) for (int a : v.NeighborVertexVec) { fprintf(f, "%s%d", first ? "" : ", ", a); first = false; }
    fprintf(f, "]}");
}

static void api_dump(const char* path, const std::vector<Fragment>& frags, const std::vector<VMACH::Polygon3D>& cells,
                     const Vector3& ext, const Vector3& cen)
{
    FILE* f = fopen(path, "w");
    if (!f) throw Error(SURTR_E_INVALID, std::string("cannot open ") + path);
    fprintf(f, "{\"cases\": [");
    for (size_t k = 0; k < frags.size() && k < 4; ++k)
    {
        const Piece& pc = frags[k].piece_data;
        fprintf(f, "%s{", k ? ", " : "");
        js_solid(f, "mesh", pc.Mesh); fprintf(f, ", "); js_solid(f, "convex", pc.Convex);
        // Poly::ExtractFaces
        Poly::Extract* ex = Poly::ExtractFaces(pc.Mesh);
        fprintf(f, ", \"faces\": [");
        for (size_t i = 0; i < ex->size(); ++i) { fprintf(f, "%s[", i ? ", " : ""); for (size_t j = 0; j < (*ex)[i].size(); ++j) fprintf(f, "%s%d", j ? ", " : "", (*ex)[i][j]); fprintf(f, "]"); }
        fprintf(f, "]");
        // Poly::RenderPolyhedron, appended after one dummy vertex to exercise the vertex offset (Src/Poly.cpp:688, 700-712)
        for (int convex = 0; convex < 2; ++convex)
        {
            std::vector<VertexNormalColor> vd(1); std::vector<uint32_t> id;
            const Poly::Polyhedron& solid = convex ? pc.Convex : pc.Mesh;
            Poly::Extract* e2 = Poly::ExtractFaces(solid);
            Poly::RenderPolyhedron(vd, id, solid, e2, convex != 0, Vector3(0.5f, 0.25f, 1.f));
            delete e2;
            fprintf(f, ", \"%s\": {\"nv\": %zu, \"idx\": [", convex ? "render_convex" : "render_mesh", vd.size());
            for (size_t i = 0; i < id.size(); ++i) fprintf(f, "%s%u", i ? ", " : "", id[i]);
            fprintf(f, "], \"color\": [%.9g, %.9g, %.9g]}", vd.back().Color[0], vd.back().Color[1], vd.back().Color[2]);
        }
        delete ex;
        // m_refittingTask spelled with the reference's names (Src/Surtr.cpp:1449-1455)
        {
            std::vector<Vector3> pts; for (const auto& v : pc.Mesh) pts.push_back(v.Position);
            Kdop::KdopContainer kdop(GenerateICHNormal(pts, (int)std::min<size_t>(pc.Mesh.size(), 4)));
            kdop.Calc(pc.Mesh);
            fprintf(f, ", "); js_solid(f, "refit_kdop", kdop.ClipWithPolyhedron(pc.Convex));
            fprintf(f, ", "); js_solid(f, "refit_task", DefaultEngine().RefitSolid(pc.Mesh, pc.Convex));
        }
        // Poly::Moments, Poly::Transform
        double vol = 0; Vector3 c1;
        Poly::Moments(vol, c1, pc.Mesh);
        fprintf(f, ", \"volume\": %.17g, \"centroid\": [%.9g, %.9g, %.9g]", vol, c1.x, c1.y, c1.z);
        Matrix w; const float W[16] = {0.f, -1.25f, 0.f, 0.5f, 1.25f, 0.f, 0.f, -1.f, 0.f, 0.f, 1.25f, 2.f, 0.f, 0.f, 0.f, 1.f};
        for (int i = 0; i < 16; ++i) w.m[i] = W[i];
        Poly::Polyhedron moved = pc.Mesh;
        Poly::Transform(moved, w);
        fprintf(f, ", \"world\": [");
        for (int i = 0; i < 16; ++i) fprintf(f, "%s%.9g", i ? ", " : "", W[i]);
        fprintf(f, "], "); js_solid(f, "moved", moved);
        fprintf(f, "}");
    }
    // Poly::ClipPolyhedron(polyhedron, polygon3D) on a placed GenerateVoronoi cell == the same clip with explicit planes
    fprintf(f, "], \"cell_clip\": [");
    for (size_t c = 0; c < cells.size() && c < 3; ++c)
    {
        VMACH::Polygon3D cell = cells[c];
        cell.Scale(ext); cell.Translate(cen);
        Poly::Polyhedron box = Poly::GetBB();
        Poly::Scale(box, ext); Poly::Translate(box, cen);
        const Poly::Polyhedron a = Poly::ClipPolyhedron(box, cell);
        std::vector<Plane> planes; for (const auto& face : cell.FaceVec) planes.push_back(face.FacePlane);
        Poly::Polyhedron b = box; Poly::ClipPolyhedron(b, planes);
        fprintf(f, "%s{", c ? ", " : ""); js_solid(f, "by_polygon", a); fprintf(f, ", "); js_solid(f, "by_planes", b);
        fprintf(f, ", \"planes\": [");
        for (size_t i = 0; i < planes.size(); ++i) fprintf(f, "%s%.9g, %.9g, %.9g, %.9g", i ? ", " : "", planes[i].x, planes[i].y, planes[i].z, planes[i].w);
        fprintf(f, "], "); js_solid(f, "box", box); fprintf(f, "}");
    }
    fprintf(f, "]");
    // Surtr::DoFracture (Src/Surtr.cpp:1885-1959) with PartialFracture on the fragments of the event as one compound: what it
    // regrouped (un-refitted Convex solids, cells, the impact sphere) and what came out (bind sets, refitted pieces), plus the
    // three tasks through their reference signatures (Inc/Surtr.h:270-272)
    {
        auto valid = [](const Poly::Polyhedron& p) {
            if (p.size() < 4) return false;
            for (size_t v = 0; v < p.size(); ++v)
            {
                if (p[v].NeighborVertexVec.size() < 3) return false;
                for (int u : p[v].NeighborVertexVec)
                {
                    if (u < 0 || (size_t)u >= p.size() || (size_t)u == v) return false;
                    const auto& r = p[(size_t)u].NeighborVertexVec;
                    if (std::find(r.begin(), r.end(), (int)v) == r.end()) return false;
                }
            }
            return true;
        };
        Compound comp;
        for (const auto& fr : frags) if (valid(fr.piece_data.Mesh) && valid(fr.piece_data.Convex)) comp.PieceVec.push_back(fr.piece_data);
        FractureEngine& eng = DefaultEngine();
        const std::vector<VMACH::Polygon3D> cells2 = eng.GenerateVoronoi(5, 46354 + 7);
        const float maxAxis = std::max(ext.x, std::max(ext.y, ext.z));
        FractureArgs args; args.PartialFracture = true;
        args.ImpactPosition = Vector3(cen.x + 0.2f * ext.x, cen.y + 0.1f * ext.y, cen.z); args.ImpactRadius = 0.3f * maxAxis;
        std::vector<Vector3> sphere;
        for (int i = 0; i < 96; ++i)       // a Fibonacci sphere stands for m_spherePointCloud
        {
            const double z = 1.0 - 2.0 * (i + 0.5) / 96.0, r = std::sqrt(1.0 - z * z), a = i * 2.399963229728653;
            sphere.emplace_back((float)(r * std::cos(a)), (float)(r * std::sin(a)), (float)z);
        }
        FractureEngine::FractureTrace tr;
        const std::vector<Compound> res = eng.DoFracture(comp, maxAxis, args, sphere, &tr);
        fprintf(f, ", \"do_fracture\": {\"n_pieces_in\": %zu, \"n_outside\": %u, \"origin\": [%.9g, %.9g, %.9g], \"radius\": %.9g, \"cloud\": [",
                comp.PieceVec.size(), tr.nOutside, args.ImpactPosition.x, args.ImpactPosition.y, args.ImpactPosition.z, args.ImpactRadius);
        for (size_t i = 0; i < sphere.size(); ++i)
        {
            Vector3 v = sphere[i];
            v.x *= args.ImpactRadius; v.y *= args.ImpactRadius; v.z *= args.ImpactRadius;
            v.x += args.ImpactPosition.x; v.y += args.ImpactPosition.y; v.z += args.ImpactPosition.z;
            fprintf(f, "%s%.9g, %.9g, %.9g", i ? ", " : "", v.x, v.y, v.z);
        }
        fprintf(f, "], \"piece_cell\": [");
        for (size_t i = 0; i < tr.PieceCell.size(); ++i) fprintf(f, "%s%d", i ? ", " : "", tr.PieceCell[i]);
        fprintf(f, "], \"convex\": [");
        for (size_t i = 0; i < tr.Convex.size(); ++i) { fprintf(f, "%s{", i ? ", " : ""); js_solid(f, "s", tr.Convex[i]); fprintf(f, "}"); }
        fprintf(f, "], \"binds\": [");
        for (size_t c = 0; c < tr.CompoundBind.size(); ++c)
        {
            fprintf(f, "%s[", c ? ", " : "");
            bool first = true;
            for (int pc : tr.CompoundBind[c]) { fprintf(f, "%s%d", first ? "" : ", ", pc); first = false; }
            fprintf(f, "]");
        }
        // the pieces that came out, compound by compound (refitted)
        fprintf(f, "], \"pieces_out\": [");
        bool firstp = true;
        for (const auto& cpd : res)
            for (const auto& pc : cpd.PieceVec) { fprintf(f, "%s{", firstp ? "" : ", "); js_solid(f, "mesh", pc.Mesh); fprintf(f, ", "); js_solid(f, "convex", pc.Convex); fprintf(f, "}"); firstp = false; }
        fprintf(f, "]");
        // the set functions on their own, from the bind sets of ApplyFracture (bind 0 = the skipped pieces, one bind per cell)
        {
            CompoundInfo info;
            for (const auto& cv : tr.Convex) info.PieceVec.emplace_back(cv, Poly::Polyhedron());
            info.CompoundBind.emplace_back();
            for (size_t i = 0; i < tr.PieceCell.size(); ++i)
            {
                if (tr.PieceCell[i] < 0) { info.CompoundBind[0].insert((int)i); continue; }
                if (i == 0 || tr.PieceCell[i] != tr.PieceCell[i - 1] || tr.PieceCell[i - 1] < 0) info.CompoundBind.emplace_back();
                info.CompoundBind.back().insert((int)i);
            }
            std::vector<Vector3> cloud = sphere;
            for (auto& v : cloud) { v.x *= args.ImpactRadius; v.y *= args.ImpactRadius; v.z *= args.ImpactRadius; v.x += args.ImpactPosition.x; v.y += args.ImpactPosition.y; v.z += args.ImpactPosition.z; }
            MergeOutOfImpact(info, cloud, args.ImpactPosition, args.ImpactRadius);
            HandleConvexIsland(info);
            fprintf(f, ", \"binds_by_set_functions\": [");
            for (size_t c = 0; c < info.CompoundBind.size(); ++c)
            {
                fprintf(f, "%s[", c ? ", " : "");
                bool first = true;
                for (int pc : info.CompoundBind[c]) { fprintf(f, "%s%d", first ? "" : ", ", pc); first = false; }
                fprintf(f, "]");
            }
            fprintf(f, "]");
        }
        // m_fractureTask / m_refittingTask / m_initCompoundTask through their reference signatures, on one placed cell
        {
            std::vector<Piece*> target;
            for (size_t i = 0; i < comp.PieceVec.size() && i < 12; ++i) target.push_back(new Piece(comp.PieceVec[i].Convex, comp.PieceVec[i].Mesh));
            // the cell of the second pattern (placed over the bounding box) that holds the first vertex of the first piece: it cuts
            // through that piece and its neighbours
            VMACH::Polygon3D cell;
            for (const auto& c2 : cells2)
            {
                VMACH::Polygon3D c = c2;
                c.Scale(ext); c.Translate(cen);
                const Vector3 q = target[0]->Mesh[0].Position;
                bool inside = true;
                for (const auto& face : c.FaceVec) if (face.FacePlane.x * q.x + face.FacePlane.y * q.y + face.FacePlane.z * q.z + face.FacePlane.w > 0.f) inside = false;
                if (inside || cell.FaceVec.empty()) cell = c;
                if (inside) break;
            }
            std::set<int> outside; outside.insert(1);
            std::vector<Piece*> got = FractureTask(cell, target, outside);
            fprintf(f, ", \"tasks\": {\"cell_planes\": [");
            for (size_t i = 0; i < cell.FaceVec.size(); ++i) fprintf(f, "%s%.9g, %.9g, %.9g, %.9g", i ? ", " : "", cell.FaceVec[i].FacePlane.x, cell.FaceVec[i].FacePlane.y, cell.FaceVec[i].FacePlane.z, cell.FaceVec[i].FacePlane.w);
            fprintf(f, "], \"target\": [");
            for (size_t i = 0; i < target.size(); ++i) { fprintf(f, "%s{", i ? ", " : ""); js_solid(f, "mesh", target[i]->Mesh); fprintf(f, ", "); js_solid(f, "convex", target[i]->Convex); fprintf(f, "}"); }
            fprintf(f, "], \"fractured\": [");
            for (size_t i = 0; i < got.size(); ++i) { fprintf(f, "%s{", i ? ", " : ""); js_solid(f, "mesh", got[i]->Mesh); fprintf(f, ", "); js_solid(f, "convex", got[i]->Convex); fprintf(f, "}"); }
            fprintf(f, "], \"refitted\": [");
            for (size_t i = 0; i < got.size(); ++i) { RefittingTask(got[i]); fprintf(f, "%s{", i ? ", " : ""); js_solid(f, "convex", got[i]->Convex); fprintf(f, "}"); }
            fprintf(f, "], \"init\": [");
            for (size_t i = 0; i < got.size(); ++i)
            {
                Poly::Extract* ex = Poly::ExtractFaces(got[i]->Convex);
                const InitCompoundResult r = InitCompoundTask(got[i], ex, false);
                delete ex;
                fprintf(f, "%s{\"points\": %zu, \"nv\": %zu, \"idx\": [", i ? ", " : "", r.ConvexPoints.size(), r.Mesh.vertexData.size());
                for (size_t q = 0; q < r.Mesh.indexData.size(); ++q) fprintf(f, "%s%u", q ? ", " : "", r.Mesh.indexData[q]);
                fprintf(f, "]}");
            }
            fprintf(f, "]}");
            for (Piece* p : target) delete p;
            for (Piece* p : got) delete p;
        }
        fprintf(f, "}");
    }
    fprintf(f, "}\n");
    fclose(f);
}

int main(int argc, char** argv)
{
    std::string mesh = "cube", obj, obj_in, dump;
    int cells = 8, nu = 250, nv = 200;
    bool ach = false; float in_scale = 1.f;
    for (int i = 1; i < argc; ++i)
    {
        if (!strcmp(argv[i], "--mesh") && i + 1 < argc) mesh = argv[++i];
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nu") && i + 1 < argc) nu = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nv") && i + 1 < argc) nv = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--obj") && i + 1 < argc) obj = argv[++i];
        else if (!strcmp(argv[i], "--obj-in") && i + 1 < argc) { obj_in = argv[++i]; mesh = obj_in; }
        else if (!strcmp(argv[i], "--scale") && i + 1 < argc) in_scale = (float)atof(argv[++i]);
        else if (!strcmp(argv[i], "--ach")) ach = true;
        else if (!strcmp(argv[i], "--api-dump") && i + 1 < argc) dump = argv[++i];
    }
    try
    {
        std::vector<Vector3> verts; std::vector<int> tris;
        if (!obj_in.empty()) LoadModelData(obj_in, Vector3(in_scale, in_scale, in_scale), Vector3(0, 0, 0), verts, tris);
        else if (mesh == "torus") make_torus(nu, nv, verts, tris);
        else make_cube(verts, tris);
        // PrepareFracture steps 3, 5, 7, 8 (Src/Surtr.cpp:1757-1803); with --ach the whole routine incl. the k-DOP clip of step 6
        Vector3 lo = verts[0], hi = verts[0];
        for (auto& p : verts) { lo.x = std::min(lo.x, p.x); hi.x = std::max(hi.x, p.x); lo.y = std::min(lo.y, p.y); hi.y = std::max(hi.y, p.y); lo.z = std::min(lo.z, p.z); hi.z = std::max(hi.z, p.z); }
        const Vector3 ext(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
        const Vector3 cen((float)(((double)hi.x + lo.x) / 2.0), (float)(((double)hi.y + lo.y) / 2.0), (float)(((double)hi.z + lo.z) / 2.0));
        Piece piece;
        Poly::InitPolyhedron(piece.Mesh, verts, Poly::ExtractNeighborFromMesh(verts, tris));
        piece.Convex = Poly::GetBB();
        Poly::Scale(piece.Convex, ext); Poly::Scale(piece.Convex, Vector3(2, 2, 2)); Poly::Translate(piece.Convex, cen);
        // GenerateVoronoi(int) (Src/Surtr.cpp:1984-2001), libstdc++ distributions
        std::mt19937 gen(46354);
        std::uniform_real_distribution<double> u(-0.5, 0.5);
        std::vector<Vector3> seeds;
        for (int i = 0; i < cells; ++i) { double x = u(gen), y = u(gen), z = u(gen); seeds.emplace_back((float)x, (float)y, (float)z); }

        FractureEngine eng(0);
        std::vector<Fragment> frags;
        if (ach) frags = eng.PrepareFracture(verts, tris, seeds);
        else
        {
            eng.SetPattern(eng.GenerateVoronoi(seeds));
            eng.PlacePattern(ext, cen);
            Compound comp; comp.PieceVec.push_back(piece);
            eng.SetCompound(comp);
            frags = eng.ApplyFracture();
        }
        if (!dump.empty()) api_dump(dump.c_str(), frags, eng.GenerateVoronoi(seeds), ext, cen);
        const surtr_counts c = eng.LastCounts();
        printf("{\"mesh\": \"%s\", \"verts\": %zu, \"tris\": %zu, \"cells\": %d, \"fragments\": %u, \"mesh_verts\": %u, \"mesh_nbrs\": %u, "
               "\"conv_verts\": %u, \"indices\": %u}\n", mesh.c_str(), verts.size(), tris.size() / 3, cells, c.n_frag, c.mesh_verts,
               c.mesh_nbrs, c.conv_verts, c.n_idx);
        if (!obj.empty())
        {
            FILE* f = fopen(obj.c_str(), "w");
            if (!f) throw Error(SURTR_E_INVALID, "cannot open " + obj);
            size_t base = 1;
            for (size_t k = 0; k < frags.size(); ++k)
            {
                fprintf(f, "o cell%d_piece%d_island%d\n", frags[k].cell, frags[k].piece, frags[k].island);
                for (auto& v : frags[k].render.vertexData) fprintf(f, "v %.9g %.9g %.9g\n", v.Position[0], v.Position[1], v.Position[2]);
                const auto& ix = frags[k].render.indexData;
                for (size_t i = 0; i + 2 < ix.size(); i += 3) fprintf(f, "f %zu %zu %zu\n", base + ix[i], base + ix[i + 1], base + ix[i + 2]);
                base += frags[k].render.vertexData.size();
            }
            fclose(f);
        }
    }
    catch (const Error& e)
    {
        fprintf(stderr, "surtr_harness: %s (code %d)\n", e.what(), e.code);
        return 2;
    }
    return 0;
}
