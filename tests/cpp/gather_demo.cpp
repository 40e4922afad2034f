// A C++ caller that shards alignments over ranks WITHOUT Python or torch (Next #3c): every rank of `nranks` processes
// (one per GPU) reads the same batch file, solves its shard through the C ABI (include/eds_hip.h), and the 16-double result rows are
// all-gathered over RCCL with include/eds_hip_rccl.h.  The rendezvous is the caller's: rank 0 writes the ncclUniqueId to a file, the
// others wait for it.  On the 1-GPU test box this runs with nranks = 1 (RCCL refuses two ranks on one device).
//   gather_demo <batch.bin> <rank> <nranks> <id file> <table.bin out> [device]
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eds_hip.h"
#include "../../include/eds_hip_rccl.h"

#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "gather_demo: %s failed (line %d): %s | %s\n", #x, __LINE__, eds_last_error(), eds_gather_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: gather_demo <batch.bin> <rank> <nranks> <id file> <table out> [device]\n"); return 2; }
    const int rank = std::atoi(argv[2]), nranks = std::atoi(argv[3]), dev = argc > 6 ? std::atoi(argv[6]) : rank;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[5]; double K[4];                      // total, N, H, W, iterations
    if (std::fread(hdr, sizeof(int), 5, f) != 5 || std::fread(K, sizeof(double), 4, f) != 4) return 2;
    const int total = hdr[0], N = hdr[1], H = hdr[2], W = hdr[3], iters = hdr[4];
    const size_t per_al = (size_t)2 * N + 2 * N + N + N + (size_t)H * W + 6;          // norm_coord, grad, idp, weights, frame, v0
    int first = 0, count = 0;
    eds_gather_shard(total, nranks, rank, &first, &count);
    std::vector<double> buf(per_al * (size_t)(count > 0 ? count : 1));
    std::fseek(f, (long)(sizeof(int) * 5 + sizeof(double) * 4 + sizeof(double) * per_al * (size_t)first), SEEK_SET);
    if (count > 0 && std::fread(buf.data(), sizeof(double), per_al * (size_t)count, f) != per_al * (size_t)count) return 2;
    std::fclose(f);

    CHECK(hipSetDevice(dev) == hipSuccess);
    ncclUniqueId id;
    if (rank == 0) {
        CHECK(ncclGetUniqueId(&id) == ncclSuccess);
        FILE* g = std::fopen((std::string(argv[4]) + ".tmp").c_str(), "wb");
        CHECK(g && std::fwrite(&id, sizeof(id), 1, g) == 1);
        std::fclose(g);
        std::rename((std::string(argv[4]) + ".tmp").c_str(), argv[4]);
    } else {
        FILE* g = nullptr;
        for (int t = 0; t < 600 && !(g = std::fopen(argv[4], "rb")); ++t) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        CHECK(g && std::fread(&id, sizeof(id), 1, g) == 1);
        std::fclose(g);
    }
    ncclComm_t comm;
    CHECK(ncclCommInitRank(&comm, nranks, id, rank) == ncclSuccess);

    eds_trk_cfg cfg;
    eds_trk_cfg_default(&cfg);
    cfg.device = dev; cfg.solver = EDS_SOLVER_LM6; cfg.exec = EDS_EXEC_DEVICE;
    for (int i = 0; i < EDS_MAX_LEVELS; ++i) cfg.max_num_iterations[i] = iters;
    eds_trk* h = nullptr;
    std::vector<double> local((size_t)EDS_GATHER_ROW * (count > 0 ? count : 1)), table((size_t)EDS_GATHER_ROW * total);
    if (count > 0) {
        CHECK(eds_trk_create(&cfg, count, N, H, W, &h) == EDS_OK);
        const double p0[3] = {0, 0, 0}, q0[4] = {0, 0, 0, 1};
        for (int i = 0; i < count; ++i) {
            const double* a = buf.data() + per_al * (size_t)i;
            CHECK(eds_trk_set_keyframe(h, i, N, a, a + 2 * N, a + 4 * N, a + 5 * N, K[0], K[1], K[2], K[3]) == EDS_OK);
            CHECK(eds_trk_set_event_frame(h, i, a + 6 * N) == EDS_OK);
            CHECK(eds_trk_set_state(h, i, p0, q0, a + 6 * N + (size_t)H * W) == EDS_OK);
        }
        CHECK(eds_trk_optimize_batch(h, 0, 0, count) == EDS_OK);
        CHECK(eds_trk_sync(h) == EDS_OK);
        CHECK(eds_trk_get_results(h, 0, count, local.data()) == EDS_OK);
    }
    // the one-shot form, then the persistent context twice (start / finish: what overlaps the next step's solve)
    CHECK(eds_gather_results(comm, nullptr, local.data(), count, table.data(), total) == 0);
    std::vector<double> table2(table.size(), -1.0), table3(table.size(), -2.0);
    eds_gather* g = nullptr;
    CHECK(eds_gather_create(comm, nullptr, total, &g) == 0);
    CHECK(eds_gather_start(g, local.data(), count) == 0);
    CHECK(eds_gather_finish(g, table2.data()) == 0);
    CHECK(eds_gather_start(g, local.data(), count) == 0);
    CHECK(eds_gather_start(g, local.data(), count) != 0);                  // a second start before finish is refused
    CHECK(eds_gather_finish(g, table3.data()) == 0);
    CHECK(eds_gather_start(g, local.data(), count + 1) != 0);              // not this rank's shard size
    eds_gather_destroy(g);
    const bool same = std::memcmp(table.data(), table2.data(), table.size() * 8) == 0 && std::memcmp(table.data(), table3.data(), table.size() * 8) == 0;
    if (rank == 0) {
        FILE* o = std::fopen(argv[5], "wb");
        CHECK(o && std::fwrite(table.data(), sizeof(double), table.size(), o) == table.size());
        std::fclose(o);
    }
    if (h) eds_trk_destroy(h);
    ncclCommDestroy(comm);
    std::printf("GATHER_DEMO rank %d of %d: shard [%d, %d), forms agree %d\n", rank, nranks, first, first + count, same ? 1 : 0);
    return same ? 0 : 4;
}
