// Speculative proposals for the damped pose-only solver (EDS_SOLVER_LM6) of the persistent kernels.
//
// edss::Solver6 (eds_solver.hpp) is a serial state machine: sums of pass n -> accept test -> damped 6x6 solve -> exp(xi) T ->
// pose of pass n + 1.  On one lane that solve costs ~6 000 cycles of dependent fp64, during which every other lane of the
// workgroup waits; spreading it over a wavefront (v_readlane pivots) was measured SLOWER (7 900 cycles: a lone wavefront issues
// one instruction per 4 cycles whatever the lane count, and the cross-lane moves add instructions).
//
// What makes the solve disappear from most passes instead: LM REJECTS more than half of its candidates on this problem, and
// after a rejection the next candidate depends only on the linearisation at the accepted pose and on lambda * 4 — both known
// BEFORE the rejected candidate was even evaluated.  So whenever a fresh linearisation is solved, lanes 0..7 of wavefront 0
// solve it for lambda, 4 lambda, 16 lambda, ... in lockstep: ONE instruction stream (the solve has no data-dependent control
// flow to speak of), eight proposals, the wall time of one.  A rejected pass then just advances to the next prepared
// candidate: no solve on the critical path.  The sequence of candidates, lambdas, accept decisions and trace records is
// exactly the one Solver6::on_eval produces (tests compare both with the CPU oracle).
#pragma once
#include <hip/hip_runtime.h>

#include "eds_math.hpp"
#include "eds_solver.hpp"

namespace edsp {

#define EDS_NSPEC 8

struct PoseRT {                 // the part of a pose block that changes from pass to pass (eds_layout.hpp EDS_PB_R / _D / _T)
    double D[9], t[3];
    float f[12];                // the same twelve numbers narrowed once by the lane that made them: every wavefront of every pass reads
};                              // these (ds_read + v_readfirstlane) instead of converting twelve doubles itself
struct Spec6 {                  // one prepared candidate
    double p[3], q[4], xi[6];
    PoseRT rt;
    int ok, pad;
};
enum { MODE_USE = 0, MODE_SOLVE = 1, MODE_DONE = 2 };
struct SpecState {              // LDS
    Spec6 spec[EDS_NSPEC];
    double cur[EDS_RED_N6];     // linearisation at the accepted pose, packed like the reduction record (21 + 6 + 1)
    int k;                      // candidate under evaluation
    int mode;
};

// offset of entry (a, b), a <= b, in the packed upper triangle
__host__ __device__ constexpr int tri_off(int a, int b) { return a * 6 - a * (a - 1) / 2 + (b - a); }

__device__ __forceinline__ double next_lambda_after_reject(double lambda) {
    lambda *= 4.0;
    return lambda < 1e-6 ? 1e-6 : lambda;          // Solver6::on_eval
}

// Solver6::propose for lambda0 advanced by `w` rejections, from the packed linearisation `cur` at the accepted pose (p, q).
// One lane; everything in registers.
__device__ __noinline__ void propose(const double* __restrict__ cur, double lambda, int w, const double* __restrict__ p,
                                        const double* __restrict__ q, Spec6& out) {
    for (int i = 0; i < w; ++i) lambda = next_lambda_after_reject(lambda);
    double L[21], x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) L[EDS_TRI(i, j)] = cur[tri_off(j, i)];
        L[EDS_TRI(i, i)] = cur[tri_off(i, i)] * (1.0 + lambda);
        x[i] = -cur[21 + i];
    }
    const bool ok = edsm::chol_solve_packed<6>(L, x);
    double tp[3], tq[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) tp[i] = p[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) tq[i] = q[i];
    if (ok) edsm::se3_left_update(x, tp, tq);
#pragma unroll
    for (int i = 0; i < 6; ++i) out.xi[i] = x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { out.p[i] = tp[i]; out.rt.t[i] = tp[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) out.q[i] = tq[i];
    edsm::quat_to_RmI(tq, out.rt.D);
#pragma unroll
    for (int i = 0; i < 9; ++i) out.rt.f[i] = (float)out.rt.D[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) out.rt.f[9 + i] = (float)tp[i];
    out.ok = ok ? 1 : 0;
}

}  // namespace edsp
