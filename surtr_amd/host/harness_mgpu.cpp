// harness_mgpu.cpp -- the sharded fracture event in C++, one process per GPU, RCCL for the single exchange step.
//   surtr_harness_mgpu --ranks N [--cells C] [--nu A --nv B] [--steps K]
// The parent forks N ranks before anything touches the GPU; rank r takes device r (mod the device count), the contiguous
// cell block RankBlock(r, N, C) of ONE C-cell event on the bumpy torus, and after every event all ranks all-gather their
// fragment blobs.  Rank 0 prints one JSON line: fragments per rank, total, ms per event (max over ranks).
// The ncclUniqueId travels through a file in /tmp (single node).
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>

#include "surtr_rccl.hpp"

using namespace surtr;

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

static int run_rank(int rank, int world, int cells, int nu, int nv, int steps, const std::string& idfile)
{
    try
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "rank %d: no HIP device\n", rank); return 3; }
        const int dev = rank % ndev;
        if (hipSetDevice(dev) != hipSuccess) return 3;
        ncclUniqueId id;
        if (rank == 0)
        {
            if (ncclGetUniqueId(&id) != ncclSuccess) return 4;
            FILE* f = fopen((idfile + ".tmp").c_str(), "wb");
            if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 4;
            fclose(f);
            rename((idfile + ".tmp").c_str(), idfile.c_str());
        }
        else
        {
            FILE* f = nullptr;
            for (int tries = 0; tries < 600 && !(f = fopen(idfile.c_str(), "rb")); ++tries) usleep(100000);
            if (!f || fread(&id, sizeof(id), 1, f) != 1) return 4;
            fclose(f);
        }
        ncclComm_t comm;
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); return 4; }
        hipStream_t stream;
        if (hipStreamCreate(&stream) != hipSuccess) return 3;

        std::vector<Vector3> verts; std::vector<int> tris;
        make_torus(nu, nv, verts, tris);
        Vector3 lo = verts[0], hi = verts[0];
        for (auto& p : verts) { lo.x = std::min(lo.x, p.x); hi.x = std::max(hi.x, p.x); lo.y = std::min(lo.y, p.y); hi.y = std::max(hi.y, p.y); lo.z = std::min(lo.z, p.z); hi.z = std::max(hi.z, p.z); }
        const Vector3 ext(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
        const Vector3 cen((float)(((double)hi.x + lo.x) / 2.0), (float)(((double)hi.y + lo.y) / 2.0), (float)(((double)hi.z + lo.z) / 2.0));
        std::mt19937 gen(46354);
        std::uniform_real_distribution<double> u(-0.5, 0.5);
        std::vector<Vector3> seeds;
        for (int i = 0; i < cells; ++i) { double x = u(gen), y = u(gen), z = u(gen); seeds.emplace_back((float)x, (float)y, (float)z); }

        FractureEngine eng(dev);
        Piece piece;
        Poly::InitPolyhedron(piece.Mesh, verts, Poly::ExtractNeighborFromMesh(verts, tris));
        piece.Convex = eng.BuildACH(verts);
        eng.GenerateVoronoi(seeds);       // built on the device and installed as the pattern; inputs are replicated on every rank
        eng.PlacePattern(ext, cen);
        Compound comp; comp.PieceVec.push_back(piece);
        eng.SetCompound(comp);
        uint32_t cb = 0, ce = 0;
        RankBlock((uint32_t)rank, (uint32_t)world, (uint32_t)cells, cb, ce);
        surtr_counts c{};
        double best_ms = 1e30;
        GatheredFragments all;
        for (int s = 0; s < steps + 1; ++s)
        {
            const auto t0 = std::chrono::steady_clock::now();
            eng.PlacePattern(ext, cen);
            int rc = surtr_fracture_event(eng.Raw(), cb, ce, nullptr, SURTR_EVT_REFIT | SURTR_EVT_RENDER, &c);
            if (rc) throw Error(rc, "surtr_fracture_event");
            all = AllGatherFragments(eng, comm, world, stream);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (s > 0 && ms < best_ms) best_ms = ms;      // the first step allocates
        }
        // max over ranks of the best step: one more tiny all-gather through the host layer's stream
        double* d_ms = nullptr; double* d_all = nullptr;
        std::vector<double> all_ms(world);
        if (hipMalloc((void**)&d_ms, 8) != hipSuccess || hipMalloc((void**)&d_all, 8 * (size_t)world) != hipSuccess) return 3;
        (void)hipMemcpyAsync(d_ms, &best_ms, 8, hipMemcpyHostToDevice, stream);
        if (ncclAllGather(d_ms, d_all, 1, ncclDouble, comm, stream) != ncclSuccess) return 4;
        (void)hipMemcpyAsync(all_ms.data(), d_all, 8 * (size_t)world, hipMemcpyDeviceToHost, stream);
        (void)hipStreamSynchronize(stream);
        (void)hipFree(d_ms); (void)hipFree(d_all);
        double worst = 0; for (double m : all_ms) worst = std::max(worst, m);
        if (rank == 0)
        {
            printf("{\"ranks\": %d, \"cells\": %d, \"fragments\": %llu, \"per_rank\": [", world, cells, (unsigned long long)all.total_fragments());
            for (int r = 0; r < world; ++r) printf("%s%u", r ? ", " : "", all.counts[r].n_frag);
            printf("], \"ms_per_event_incl_allgather_and_host_copy\": %.3f, \"fragments_per_s\": %.1f}\n", worst, all.total_fragments() / (worst * 1e-3));
            fflush(stdout);
        }
        ncclCommDestroy(comm);
        (void)hipStreamDestroy(stream);
    }
    catch (const Error& e)
    {
        fprintf(stderr, "surtr_harness_mgpu rank %d: %s (code %d)\n", rank, e.what(), e.code);
        return 2;
    }
    return 0;
}

int main(int argc, char** argv)
{
    int ranks = 1, cells = 4096, nu = 250, nv = 200, steps = 5;
    for (int i = 1; i < argc; ++i)
    {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nu") && i + 1 < argc) nu = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--nv") && i + 1 < argc) nv = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
    }
    if (ranks < 1 || ranks > 8) { fprintf(stderr, "--ranks 1..8\n"); return 1; }
    const std::string idfile = "/tmp/surtr_nccl_id_" + std::to_string((long)getpid());
    unlink(idfile.c_str());
    // fork BEFORE any HIP call: a process that has initialised the GPU must not be duplicated
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r)
    {
        const pid_t p = fork();
        if (p < 0) { perror("fork"); return 1; }
        if (p == 0) _exit(run_rank(r, ranks, cells, nu, nv, steps, idfile));
        kids.push_back(p);
    }
    int worst = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128; worst = code > worst ? code : worst; }
    unlink(idfile.c_str());
    return worst;
}
