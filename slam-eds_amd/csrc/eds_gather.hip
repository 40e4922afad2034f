// libeds_hip_rccl.so — include/eds_hip_rccl.h: the all-gather of the 16-double result rows over RCCL (xGMI) for a C / C++ caller.
// The data-parallel path has exactly one collective: 8 KB for BASELINE.json's configs[4], latency-bound; what matters is
// that nothing is allocated per step and that it can run behind the next step's solve.  Pinned host staging both ways, two device
// buffers, a stream and an event — all made once per (communicator, total).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/eds_hip_rccl.h"

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }
}  // namespace

// the caller's current device is left as it was found: every entry point that touches HIP runs on the communicator's device inside
// one of these (a one-process, many-GPU caller — ncclCommInitAll — has a different device current for every context)
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct eds_gather {
    int dev = 0;
    ncclComm_t comm = nullptr;
    hipStream_t st = nullptr;
    bool own_stream = false;
    hipEvent_t ev = nullptr;
    int total = 0, nranks = 1, rank = 0, per = 0;
    double *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    bool pending = false;
};

extern "C" {

const char* eds_gather_last_error(void) { return g_err.c_str(); }

void eds_gather_shard(int total, int nranks, int rank, int* first, int* count) {
    if (nranks < 1) nranks = 1;
    const int per = (total + nranks - 1) / nranks;
    int f = rank * per; if (f > total) f = total;
    int c = total - f; if (c > per) c = per; if (c < 0) c = 0;
    if (first) *first = f;
    if (count) *count = c;
}

// Pure: rank r's padded block of `per` rows inside the gathered buffer -> its rows [first, first + count) of the table.  (Table-tested
// on the CPU for ragged totals and worlds of 2, 3, 8: tests/test_gather_capi.py through eds_gather_unpack.)
void eds_gather_unpack(const double* gathered, int total, int nranks, double* table) {
    if (nranks < 1) nranks = 1;
    const int per = (total + nranks - 1) / nranks;
    for (int r = 0; r < nranks; ++r) {
        int first = 0, cnt = 0;
        eds_gather_shard(total, nranks, r, &first, &cnt);
        if (cnt > 0) std::memcpy(table + (size_t)first * EDS_GATHER_ROW, gathered + (size_t)r * per * EDS_GATHER_ROW, sizeof(double) * EDS_GATHER_ROW * (size_t)cnt);
    }
}

void eds_gather_destroy(eds_gather* g) {
    if (!g) return;
    DeviceGuard guard(g->dev);
    if (g->st) (void)hipStreamSynchronize(g->st);        // whatever was queued (also by a start that failed half-way) is done with the buffers
    if (g->d_in) (void)hipFree(g->d_in);
    if (g->d_out) (void)hipFree(g->d_out);
    if (g->h_in) (void)hipHostFree(g->h_in);
    if (g->h_out) (void)hipHostFree(g->h_out);
    if (g->ev) (void)hipEventDestroy(g->ev);
    if (g->own_stream && g->st) (void)hipStreamDestroy(g->st);
    delete g;
}

int eds_gather_create(void* nccl_comm, void* hip_stream, int total, eds_gather** out) {
    if (!nccl_comm || !out || total < 0) return fail(-1, "eds_gather_create: null communicator / output or negative total");
    *out = nullptr;
    eds_gather* g = new (std::nothrow) eds_gather();
    if (!g) return fail(-1, "out of memory");
    g->comm = (ncclComm_t)nccl_comm;
    g->total = total;
    int dev = -1;
    if (ncclCommCount(g->comm, &g->nranks) != ncclSuccess || ncclCommUserRank(g->comm, &g->rank) != ncclSuccess || ncclCommCuDevice(g->comm, &dev) != ncclSuccess) {
        delete g; return fail(-2, "the communicator does not answer (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)");
    }
    g->dev = dev;
    DeviceGuard guard(dev);
    if (!guard.ok) { delete g; return fail(-2, "hipSetDevice(communicator's device) failed"); }
    g->per = (total + g->nranks - 1) / g->nranks;
    const size_t in_b = sizeof(double) * EDS_GATHER_ROW * (size_t)(g->per > 0 ? g->per : 1), out_b = in_b * (size_t)g->nranks;
    hipError_t e = hipSuccess;
    if (hip_stream) g->st = (hipStream_t)hip_stream;
    else { e = hipStreamCreateWithFlags(&g->st, hipStreamNonBlocking); g->own_stream = (e == hipSuccess); }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&g->h_in, in_b, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&g->h_out, out_b, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_in, in_b);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_out, out_b);
    if (e != hipSuccess) { const std::string m = std::string("eds_gather_create: ") + hipGetErrorString(e); eds_gather_destroy(g); return fail(-2, m); }
    std::memset(g->h_in, 0, in_b);
    std::memset(g->h_out, 0, out_b);
    *out = g;
    return 0;
}

int eds_gather_start(eds_gather* g, const double* local, int count) {
    if (!g || (count > 0 && !local)) return fail(-1, "eds_gather_start: null argument");
    if (g->pending) return fail(-1, "eds_gather_start: the previous gather was not finished");
    int first = 0, mine = 0;
    eds_gather_shard(g->total, g->nranks, g->rank, &first, &mine);
    if (count != mine) return fail(-1, "eds_gather_start: count is not this rank's shard size (eds_gather_shard)");
    if (g->per == 0) { g->pending = true; return 0; }
    const size_t row = sizeof(double) * EDS_GATHER_ROW;
    if (count > 0) std::memcpy(g->h_in, local, row * (size_t)count);
    if (count < g->per) std::memset(reinterpret_cast<char*>(g->h_in) + row * (size_t)count, 0, row * (size_t)(g->per - count));     // ragged last shard: padded
    DeviceGuard guard(g->dev);
    if (!guard.ok) return fail(-2, "eds_gather_start: hipSetDevice(communicator's device) failed");
    // From the first enqueue on, a failure must not leave work in flight over buffers the next start would overwrite:
    // the stream is drained before the error is reported.
    auto bail = [&](const std::string& m) { (void)hipStreamSynchronize(g->st); return fail(-2, m); };
    hipError_t e = hipMemcpyAsync(g->d_in, g->h_in, row * (size_t)g->per, hipMemcpyHostToDevice, g->st);
    if (e != hipSuccess) return bail(std::string("eds_gather_start: ") + hipGetErrorString(e));
    const ncclResult_t r = ncclAllGather(g->d_in, g->d_out, (size_t)g->per * EDS_GATHER_ROW, ncclDouble, g->comm, g->st);
    if (r != ncclSuccess) return bail(std::string("ncclAllGather: ") + ncclGetErrorString(r));
    e = hipMemcpyAsync(g->h_out, g->d_out, row * (size_t)g->per * (size_t)g->nranks, hipMemcpyDeviceToHost, g->st);
    if (e == hipSuccess) e = hipEventRecord(g->ev, g->st);
    if (e != hipSuccess) return bail(std::string("eds_gather_start: ") + hipGetErrorString(e));
    g->pending = true;
    return 0;
}

int eds_gather_finish(eds_gather* g, double* table) {
    if (!g) return fail(-1, "eds_gather_finish: null context");
    if (!g->pending) return fail(-1, "eds_gather_finish: nothing was started");
    g->pending = false;
    if (g->per == 0) return 0;
    DeviceGuard guard(g->dev);
    const hipError_t e = hipEventSynchronize(g->ev);
    if (e != hipSuccess) return fail(-2, std::string("eds_gather_finish: ") + hipGetErrorString(e));
    if (table) eds_gather_unpack(g->h_out, g->total, g->nranks, table);
    return 0;
}

int eds_gather_results(void* nccl_comm, void* hip_stream, const double* local, int count, double* table, int total) {
    if (!table && total > 0) return fail(-1, "eds_gather_results: null table");
    eds_gather* g = nullptr;
    int rc = eds_gather_create(nccl_comm, hip_stream, total, &g);
    if (rc) return rc;
    rc = eds_gather_start(g, local, count);
    if (!rc) rc = eds_gather_finish(g, table);
    const std::string keep = g_err;
    eds_gather_destroy(g);
    if (rc) g_err = keep;
    return rc;
}

}  // extern "C"
