// Internals shared by the three translation units of the C ABI (include/eds_hip.h):
//   eds_capi.hip          handle lifecycle, configuration, knobs, states, sync / info, the rows around the path (points, keyframes)
//   eds_capi_inputs.hip   what goes INTO a slot: keyframe points, inverse depths, event frames (host buffers or events), shared frames
//   eds_capi_solve.hip    the passes and solves: eval, the host-driven loop (EDS_EXEC_HOST), optimize, residuals, loss scale, bench hooks
// Nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/eds_hip.h"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_kernels.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"

namespace edscapi {

int fail(int code, const std::string& msg);          // records the message for eds_last_error() (thread-local), returns `code`

#define EDS_HIP_TRY(expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return edscapi::fail(EDS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// eds_capi.hip
int effective_blocks(const eds_trk* h);
int level_iters(const eds_trk* h, int level);
int check_slot(const eds_trk* h, int slot);
int check_range(const eds_trk* h, int first, int count);
void fill_static(const eds_trk* h, int slot);
void fill_pose(eds_trk* h, int slot, const double* p, const double* q, const double* v);
int upload_pose(eds_trk* h, int first, int count);
int max_points(const eds_trk* h, int first, int count);
// eds_capi_inputs.hip
int refresh_gram(eds_trk* h, int slot, bool wait = true);
// eds_capi_solve.hip
int solve_host(eds_trk* h, int level, int first, int count);
int materialise_residuals(eds_trk* h, int slot);

}  // namespace edscapi
