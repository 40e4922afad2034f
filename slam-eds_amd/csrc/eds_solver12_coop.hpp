// Wavefront-cooperative execution of edss::Solver12 (eds_solver.hpp) for the persistent REF12 kernel.
//
// The state machine is the one the host-driven loop runs — same decisions, same constants, same order of the
// floating-point sums inside every dot product — but the O(12^2)..O(12^3) pieces of an LM iteration (assembling the
// corrected normal equations from the per-block sums, projecting the velocity columns, Jacobi scaling, the damped
// 12x12 Cholesky solve, the model-cost quadratic form, the per-block constants of the next pose block) are spread over
// the 64 lanes of wavefront 0 instead of running on one lane with 78 + 144 doubles in registers (which spilled to
// scratch and cost ~60 us per iteration).  The factorisation keeps row i of the triangle in lane i's registers and
// moves pivots by v_readlane; the other pieces exchange through a small LDS work area.  Scalar decisions stay on
// lane 0 and reach the other lanes through LDS.
//
// All 64 lanes of ONE wavefront must call these functions together.  LDS operations of a wavefront retire in order,
// so a wavefront-scope fence + wave barrier is all the synchronisation needed.
#pragma once
#include <hip/hip_runtime.h>

#include "eds_math.hpp"
#include "eds_solver.hpp"

namespace edsc {

#define EDS_WSYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

enum { M_RETURN = 0, M_LIN_ITER0 = 1, M_LIN_ACCEPT = 2, M_ADVANCE = 3, M_LOOP = 4, M_SOLVE = 5 };

struct Work12 {                 // LDS scratch of the cooperative solver
    double y[12], t[12];
    double T[144];              // A P while the velocity columns are being projected
    double r0[EDS_DEV_MAX_BLOCKS], r1[EDS_DEV_MAX_BLOCKS];
    double cost, rel;
    int mode, ok, accepted;     // accepted: the evaluation just consumed became the accepted point
#ifdef EDS_FUSED_STAMPS
    unsigned long long st[8], st_t;
#endif
};
#ifdef EDS_FUSED_STAMPS
#define EDS_CSTAMP(k) do { if (lane == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); W.st[k] += n_ - W.st_t; W.st_t = n_; } } while (0)
#else
#define EDS_CSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int uniform_int(int x) { return __builtin_amdgcn_readfirstlane(x); }
// value of lane `src` (compile-time constant after unrolling) for every lane: two v_readlane_b32
__device__ __forceinline__ double bcast(double x, int src) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// edss::Solver12::on_eval + advance, cooperatively.  On return either sv.done is set, or (sv.cp, sv.cq, sv.cv) is
// the next point to evaluate (sv.final_pass tells whether that evaluation is the residual pass at the solution).
// `pb` is the pose block the sums were evaluated at: the persistent kernel accumulates the velocity columns WITHOUT the
// local-parameterisation factor Pv = (I - v v^T/|v|^2)/|v| (identical for all points), so linearising applies
// J^T J -> P^T (J^T J) P, J^T r -> P^T (J^T r) with P = blockdiag(I_6, Pv) here, once, on the 12 x 12 sums.
__device__ inline void coop12_on_eval(edss::Solver12& sv, const edss::Sums12Dev& S, Work12& W, const double* pb, const int lane) {
    using namespace edss;
    const int nb = S.nb;
#ifdef EDS_FUSED_STAMPS
    if (lane == 0) W.st_t = __builtin_readcyclecounter();
#endif
    if (lane < nb) {
        double r0, r1;
        loss_eval(sv.loss_type, sv.loss_a, S.s[lane], &r0, &r1);
        W.r0[lane] = r0; W.r1[lane] = r1;
    }
    EDS_WSYNC();
    if (lane == 0) {                    // what does this evaluation mean?  (Solver12::on_eval)
        double c = 0.0;
        for (int k = 0; k < nb; ++k) c += 0.5 * W.r0[k];
        W.cost = c;
        W.accepted = 0;
        int mode;
        if (sv.final_pass) {
            sv.final_cost = c; sv.done = 1; mode = M_RETURN;
        } else if (!sv.started) {       // IterationZero
            sv.started = 1;
            sv.x_norm = Solver12::norm13(sv.p, sv.q, sv.v);
            mode = M_LIN_ITER0;
        } else {
            const double cand_cost = ((c == c) && fabs(c) < 1e300) ? c : 1.7976931348623157e308;
            double sn = 0.0;
            for (int i = 0; i < 3; ++i) sn += (sv.p[i] - sv.cp[i]) * (sv.p[i] - sv.cp[i]);
            for (int i = 0; i < 4; ++i) sn += (sv.q[i] - sv.cq[i]) * (sv.q[i] - sv.cq[i]);
            for (int i = 0; i < 6; ++i) sn += (sv.v[i] - sv.cv[i]) * (sv.v[i] - sv.cv[i]);
            const double cost_change = sv.x_cost - cand_cost;
            if (sqrt(sn) <= sv.ptol * (sv.x_norm + sv.ptol)) {          // ParameterToleranceReached
                sv.finish(TERM_CONVERGENCE); mode = M_RETURN;
            } else if (fabs(cost_change) <= sv.ftol * sv.x_cost) {      // FunctionToleranceReached
                sv.finish(TERM_CONVERGENCE); mode = M_RETURN;
            } else {
                const double rel = cost_change / sv.model_cost_change;
                if (rel > 1e-3) {                                       // HandleSuccessfulStep, first half
                    for (int i = 0; i < 3; ++i) sv.p[i] = sv.cp[i];
                    for (int i = 0; i < 4; ++i) sv.q[i] = sv.cq[i];
                    for (int i = 0; i < 6; ++i) sv.v[i] = sv.cv[i];
                    sv.x_norm = Solver12::norm13(sv.p, sv.q, sv.v);
                    W.rel = rel;
                    mode = M_LIN_ACCEPT;
                } else {                                                // HandleUnsuccessfulStep
                    sv.radius /= sv.decrease_factor; sv.decrease_factor *= 2.0; sv.reuse_diagonal = 1;
                    mode = M_ADVANCE;
                }
            }
        }
        W.mode = mode;
    }
    EDS_WSYNC();
    EDS_CSTAMP(0);
    const int mode = uniform_int(W.mode);
    if (mode == M_RETURN) return;

    if (mode != M_ADVANCE) {            // Solver12::linearise at the (new) accepted point
        bool bad = !(fabs(W.cost) < 1e300);
        for (int i = lane; i < 144; i += 64) {
            double a = 0.0;
            for (int k = 0; k < nb; ++k) a += W.r1[k] * S.H[k][i];
            sv.A[i] = a;
            bad |= !(fabs(a) < 1e300);
        }
        if (lane < 12) {
            double a = 0.0;
            for (int k = 0; k < nb; ++k) a += W.r1[k] * S.g[k][lane];
            sv.g[lane] = a;
            bad |= !(fabs(a) < 1e300);
        }
        const bool anybad = __ballot(bad) != 0ull;
        EDS_WSYNC();
        if (anybad) {
            if (lane == 0) {
                if (mode == M_LIN_ITER0) { sv.termination = TERM_FAILURE; sv.done = 1; }
                else sv.finish(TERM_FAILURE);
            }
            EDS_WSYNC();
            return;
        }
        {
            const double* Pv = pb + EDS_PB_PV;
            for (int e = lane; e < 144; e += 64) {                  // W.T = A P
                const int i = e / 12, j = e - 12 * i;
                double t = sv.A[e];
                if (j >= 6) {
                    t = 0.0;
                    for (int c = 0; c < 6; ++c) t += sv.A[12 * i + 6 + c] * Pv[6 * c + (j - 6)];
                }
                W.T[e] = t;
            }
            double gp = 0.0;
            if (lane < 12) {
                gp = sv.g[lane];
                if (lane >= 6) {
                    gp = 0.0;
                    for (int c = 0; c < 6; ++c) gp += Pv[6 * c + (lane - 6)] * sv.g[6 + c];
                }
            }
            EDS_WSYNC();
            for (int e = lane; e < 144; e += 64) {                  // A = P^T W.T
                const int i = e / 12, j = e - 12 * i;
                double t = W.T[e];
                if (i >= 6) {
                    t = 0.0;
                    for (int c = 0; c < 6; ++c) t += Pv[6 * c + (i - 6)] * W.T[12 * (6 + c) + j];
                }
                sv.A[e] = t;
            }
            if (lane < 12) sv.g[lane] = gp;
            EDS_WSYNC();
        }
        if (!sv.have_scale && lane < 12) sv.scale[lane] = 1.0 / (1.0 + sqrt(sv.A[13 * lane]));
        EDS_WSYNC();
        if (lane == 0) {
            sv.have_scale = 1;
            sv.x_cost = W.cost;
            double ng[12], pp[3], pq[4], pv[6];
            for (int k = 0; k < 12; ++k) ng[k] = -sv.g[k];
            edsm::state_plus12(sv.p, sv.q, sv.v, ng, pp, pq, pv);
            double m = 0.0;
            for (int i = 0; i < 3; ++i) m = fmax(m, fabs(sv.p[i] - pp[i]));
            for (int i = 0; i < 4; ++i) m = fmax(m, fabs(sv.q[i] - pq[i]));
            for (int i = 0; i < 6; ++i) m = fmax(m, fabs(sv.v[i] - pv[i]));
            sv.grad_max_norm = m;
            if (mode == M_LIN_ITER0) {
                sv.initial_cost = sv.x_cost; sv.minimum_cost = sv.x_cost;
                sv.iteration = 0; sv.step_successful = 1;
            } else {                                                    // HandleSuccessfulStep, second half
                sv.step_successful = 1;
                const double t = 2.0 * W.rel - 1.0;
                sv.radius = sv.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
                sv.radius = fmin(1e16, sv.radius);
                sv.decrease_factor = 2.0; sv.reuse_diagonal = 0;
            }
            W.accepted = 1;
        }
        EDS_WSYNC();
        EDS_CSTAMP(1);
    }

    for (;;) {                          // Solver12::advance
        if (lane == 0) {
            int m = M_SOLVE;
            if (sv.step_successful) {
                ++sv.num_successful;
                if (sv.x_cost < sv.minimum_cost || sv.iteration == 0) {
                    sv.minimum_cost = sv.x_cost;
                    for (int i = 0; i < 3; ++i) sv.best_p[i] = sv.p[i];
                    for (int i = 0; i < 4; ++i) sv.best_q[i] = sv.q[i];
                    for (int i = 0; i < 6; ++i) sv.best_v[i] = sv.v[i];
                }
            } else {
                ++sv.num_unsuccessful;
            }
            if (sv.iteration >= sv.max_iters) { sv.finish(TERM_NO_CONVERGENCE); m = M_RETURN; }
            else if (sv.step_successful && sv.grad_max_norm <= sv.gtol) { sv.finish(TERM_CONVERGENCE); m = M_RETURN; }
            else if (sv.radius < 1e-32) { sv.finish(TERM_CONVERGENCE); m = M_RETURN; }
            else { ++sv.iteration; sv.step_successful = 0; }
            W.mode = m; W.ok = 1;
        }
        EDS_WSYNC();
        EDS_CSTAMP(2);
        if (uniform_int(W.mode) == M_RETURN) return;
        // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system.  Lane i (< 12) holds ROW i of the lower
        // triangle in registers; pivots and pivot-row entries travel by v_readlane (wave-uniform SGPR broadcasts), so the
        // whole factorisation and both substitutions run without touching LDS.  Operation order per entry is that of
        // edsm::chol_solve_packed.
        const int row = lane < 12 ? lane : 0;
        double Lr[12];
        {
            const double sr = sv.scale[row];
#pragma unroll
            for (int b = 0; b < 12; ++b) Lr[b] = sv.A[12 * row + b] * sr * sv.scale[b];     // only b <= row is used
        }
        double bi = sv.g[row] * sv.scale[row];
        {
            double dg = 0.0;
#pragma unroll
            for (int b = 0; b < 12; ++b) dg = (b == row) ? Lr[b] : dg;
            double dd = sv.reuse_diagonal ? sv.diagonal[row] : fmin(fmax(dg, 1e-6), 1e32);
            if (!sv.reuse_diagonal && lane < 12) sv.diagonal[lane] = dd;
            dg += dd / sv.radius;
#pragma unroll
            for (int b = 0; b < 12; ++b) Lr[b] = (b == row) ? dg : Lr[b];
        }
        double idr = 0.0;               // 1 / L_rr of this lane's row
        bool okl = true;
        // right-looking: once column j is scaled, every row subtracts its share from the columns to the right.  Each entry
        // still receives its subtractions in ascending k — the same sums as the serial left-looking code — but the updates of
        // one step are independent of each other, so only pivot -> rsqrt -> scale is on the critical path.
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const double d = bcast(Lr[j], j);
            okl = okl && (d > 0.0) && (d < 1e300);
            const double inv = edsm::rsqrt_(d);
            if (row == j) { idr = inv; Lr[j] = d * inv; }
            else Lr[j] = Lr[j] * inv;                                        // meaningful for row > j
#pragma unroll
            for (int c = j + 1; c < 12; ++c) Lr[c] -= Lr[j] * bcast(Lr[j], c);   // L[i][j] * L[c][j]
        }
        // L y = b  (each row subtracts in ascending k, y_k broadcast from lane k)
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if (row == k) bi = bi * idr;
            const double yk = bcast(bi, k);
            if (row > k) bi -= Lr[k] * yk;
        }
        // L^T x = y: x_i = (y_i - sum_{k > i} L[k][i] x_k) / L_ii with k ascending, exactly like the serial code
#pragma unroll
        for (int i = 11; i >= 0; --i) {
            double s = bcast(bi, i);
#pragma unroll
            for (int k = i + 1; k < 12; ++k) s -= bcast(Lr[i], k) * bcast(bi, k);          // L[k][i] lives in lane k
            const double xi = s * bcast(idr, i);
            if (row == i) bi = xi;
        }
        if (lane == 0) W.ok = okl ? 1 : 0;
        if (lane < 12) W.y[lane] = bi;
        EDS_WSYNC();
        EDS_CSTAMP(3);
        if (lane < 12) sv.step[lane] = -W.y[lane];
        EDS_WSYNC();
        if (lane < 12) {
            double t = 0.0;
            for (int b = 0; b < 12; ++b) t += sv.A[12 * lane + b] * sv.scale[b] * sv.step[b];
            W.t[lane] = t;
        }
        EDS_WSYNC();
        EDS_CSTAMP(4);
        if (lane == 0) {
            sv.reuse_diagonal = 1;
            double chk = 0.0;
            for (int a = 0; a < 12; ++a) chk += W.y[a];
            bool valid = W.ok && (chk == chk) && (fabs(chk) < 1e300);
            if (valid) {
                double sg = 0.0, sAs = 0.0;
                for (int a = 0; a < 12; ++a) {
                    sg += sv.step[a] * sv.g[a] * sv.scale[a];
                    sAs += sv.step[a] * sv.scale[a] * W.t[a];
                }
                sv.model_cost_change = -sg - 0.5 * sAs;
                valid = sv.model_cost_change > 0.0;
            }
            int m;
            if (!valid) {               // HandleInvalidStep
                if (++sv.consecutive_invalid >= 5) { sv.finish(TERM_FAILURE); m = M_RETURN; }
                else { sv.radius /= sv.decrease_factor; sv.decrease_factor *= 2.0; sv.reuse_diagonal = 1; m = M_LOOP; }
            } else {
                sv.consecutive_invalid = 0;
                double delta[12];
                for (int k = 0; k < 12; ++k) delta[k] = sv.step[k] * sv.scale[k];
                edsm::state_plus12(sv.p, sv.q, sv.v, delta, sv.cp, sv.cq, sv.cv);
                m = M_RETURN;
            }
            W.mode = m;
        }
        EDS_WSYNC();
        EDS_CSTAMP(5);
        if (uniform_int(W.mode) == M_RETURN) return;
    }
}

// edsm::fill_pose_block, cooperatively (G: the per-block Gram matrices of the slot, nb <= EDS_DEV_MAX_BLOCKS)
__device__ inline void coop_fill_pose_block(const double* p, const double* q, const double* v, const double* __restrict__ G,
                                            int nb, double* pb, const int lane) {
    if (lane == 0) {
        edsm::quat_to_R(q, pb + EDS_PB_R);
        edsm::quat_to_RmI(q, pb + EDS_PB_D);
        for (int i = 0; i < 3; ++i) pb[EDS_PB_T + i] = p[i];
        for (int i = 0; i < 4; ++i) pb[EDS_PB_Q + i] = q[i];
    }
    if (lane < 6) pb[EDS_PB_V + lane] = v[lane];
    double vv = 0.0;
    for (int i = 0; i < 6; ++i) vv += v[i] * v[i];
    const double vn = sqrt(vv);
    if (lane < 36) {
        const int i = lane / 6, j = lane - 6 * i;
        pb[EDS_PB_PV + lane] = ((i == j ? 1.0 : 0.0) - v[i] * v[j] / vv) / vn;
    }
    const int k = lane / 6, i6 = lane - 6 * k;
    double* o = pb + EDS_PB_BLK + EDS_PB_BLK_STRIDE * (k < nb ? k : 0);
    if (k < nb) {
        const double* Gk = G + 36 * k;
        double s = 0.0;
        for (int j = 0; j < 6; ++j) s += Gk[6 * i6 + j] * v[j];
        o[1 + i6] = s;
    }
    EDS_WSYNC();
    if (k < nb) {                       // the six lanes of a block all form S, n (same operation order as the serial code);
        double raw[6];                  // every read of the raw dots precedes the writes below: one wavefront, in lockstep
        for (int i = 0; i < 6; ++i) raw[i] = o[1 + i];
        double S = 1e-3;
        for (int i = 0; i < 6; ++i) S += v[i] * raw[i];
        const double n = sqrt(S);
        o[1 + i6] = raw[i6] / (n * n * n);
        if (i6 == 0) { o[0] = 1.0 / n; o[7] = S; }
    }
    EDS_WSYNC();
}

}  // namespace edsc
