// Persistent on-device solve (EDS_EXEC_DEVICE): one workgroup per alignment runs the whole
// Gauss-Newton / damped loop — residual+Jacobian pass, wavefront reduction, 6x6 solve, SE(3)
// update, accept test — without returning to the host between iterations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct eds_trk;

struct EdsFusedIn {            // start state of a slot
    double p[3], q[4], v[6], pad[3];
};
struct EdsFusedOut {           // compact result of a slot (trace stays in HBM until asked for)
    double p[3], q[4];
    double initial_cost, final_cost;
    int32_t iterations, ntrace, failed, naccepted;
    double pad[3];
};

struct EdsFusedBuffers {
    EdsFusedIn* d_in = nullptr;
    EdsFusedOut* d_out = nullptr;
    void* d_sv = nullptr;          // edss::Solver6 per slot (full state incl. trace)
    EdsFusedIn* h_in = nullptr;    // pinned
    EdsFusedOut* h_out = nullptr;  // pinned
    int B = 0;
    int pending_first = 0, pending_count = 0;   // range launched but not yet collected
    double launch_wall_us = 0.0;
    void* t0 = nullptr;
};

int  eds_fused_alloc(EdsFusedBuffers* fb, int B);
void eds_fused_free(EdsFusedBuffers* fb);
int  eds_fused_solve(eds_trk* h, int level, int first, int count);   // asynchronous on h->st
int  eds_fused_collect(eds_trk* h);                                  // after the stream is idle
int  eds_fused_fetch_trace(eds_trk* h, int slot);                    // D2H of one slot's trace

// defined in eds_capi.hip
int eds_internal_fail(int code, const char* msg);
int eds_internal_solve_host(eds_trk* h, int level, int first, int count);
