// harness.cpp -- headless replacement of the Win32/DX12 shell for the fracture-event path:
// builds a synthetic closed mesh, a Voronoi pattern, runs ONE fracture event on the GPU and prints a
// JSON summary (optionally writes the fragments as an OBJ).  Usage:
//   surtr_harness [--mesh cube|torus | --obj-in mesh.obj [--scale S]] [--cells N] [--nu A --nv B] [--ach] [--obj out.obj]
// --ach runs Surtr::PrepareFracture end to end (ACH convex instead of the plain 2x box).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

int main(int argc, char** argv)
{
    std::string mesh = "cube", obj, obj_in;
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
            eng.SetPattern(FractureEngine::GenerateVoronoi(seeds));
            eng.PlacePattern(ext, cen);
            Compound comp; comp.PieceVec.push_back(piece);
            eng.SetCompound(comp);
            frags = eng.ApplyFracture();
        }
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
