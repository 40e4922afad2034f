// Iteration state machines of the tracker, driven by REDUCED SUMS only (J^T J, J^T r,
// sum r^2 per residual block) — never by per-point data.  The same code runs on the host
// (EDS_EXEC_HOST: sums come back over PCIe each iteration) and in the solver lane of the
// persistent device kernel (EDS_EXEC_DEVICE).
//
//   Solver6   pose-only SE(3) Gauss-Newton (EDS_SOLVER_GN6) and its damped, accept/reject
//             variant (EDS_SOLVER_LM6; template: reference CoarseTracker.cpp:545-664)
//   Solver12  the reference problem solved the way ceres::Solve does it with the
//             reference's options (Tracker.cpp:117-143,197-202): 12 local parameters,
//             per-block robust loss, Jacobi scaling, Levenberg-Marquardt trust region,
//             Ceres default constants, same termination tests and step accounting.
//
// Protocol: init() -> the caller evaluates at cand_* -> on_eval(sums) -> repeat until done.
#pragma once
#include "eds_math.hpp"

namespace edss {

enum { TERM_CONVERGENCE = 0, TERM_NO_CONVERGENCE = 1, TERM_FAILURE = 2 };
#define EDS_MAX_TRACE 128

// ---------------------------------------------------------------------------------------
struct Sums6 {
    double H[36], b[6], cost;   // sum hw J^T J (full symmetric), sum hw J^T r, sum hw r^2 (2 - hw)
};
// unpack an EDS_RED record (upper triangle, J^T r, cost)
EDS_HD void unpack6(const double* rec, Sums6* s) {
    int c = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { s->H[6 * a + b] = rec[c]; s->H[6 * b + a] = rec[c]; ++c; }
    for (int a = 0; a < 6; ++a) s->b[a] = rec[c++];
    s->cost = rec[c];
}

struct Solver6 {
    int damped;                 // 0: GN6, 1: LM6
    int max_iters;
    double lambda;
    double p[3], q[4];          // accepted pose
    double cp[3], cq[4];        // pose to evaluate next
    Sums6 cur;
    int have_cur, iter, done, failed, final_pass;
    int skip_final;             // the caller keeps the residuals of the accepted pose itself (persistent kernel): no extra pass
    int last_accepted;          // whether the pass just consumed became the accepted pose
    double initial_cost, final_cost;
    double xi[6];               // increment that produced the current candidate
    // trace of the solve (parity / diagnostics)
    int ntrace;
    double tr_xi[EDS_MAX_TRACE][6], tr_cost[EDS_MAX_TRACE];
    int tr_acc[EDS_MAX_TRACE];

    EDS_HD void init(int damped_, int max_iters_, double lambda0, const double* p0, const double* q0, int skip_final_ = 0) {
        damped = damped_; max_iters = max_iters_; lambda = damped_ ? lambda0 : 0.0;
        skip_final = skip_final_; last_accepted = 0;
        for (int i = 0; i < 3; ++i) p[i] = cp[i] = p0[i];
        for (int i = 0; i < 4; ++i) q[i] = cq[i] = q0[i];
        have_cur = 0; iter = 0; done = 0; failed = 0; final_pass = 0; ntrace = 0;
        initial_cost = final_cost = 0.0;
        if (max_iters <= 0) { final_pass = 1; }
    }
    EDS_HD bool finite6(const Sums6& s) const {
        double t = s.cost;
        for (int i = 0; i < 36; ++i) t += s.H[i];
        for (int i = 0; i < 6; ++i) t += s.b[i];
        return (t == t) && (fabs(t) < 1e300);
    }
    // Solves from `cur`, writes the next candidate.  Returns false if the system is not PD.
    EDS_HD bool propose() {
        double L[21], x[6];
        EDS_UNROLL
        for (int i = 0; i < 6; ++i) {
            EDS_UNROLL
            for (int j = 0; j < i; ++j) L[EDS_TRI(i, j)] = cur.H[6 * i + j];
            L[EDS_TRI(i, i)] = cur.H[7 * i] * (1.0 + lambda);
            x[i] = -cur.b[i];
        }
        if (!edsm::chol_solve_packed<6>(L, x)) return false;
        double tp[3], tq[4];
        EDS_UNROLL
        for (int i = 0; i < 6; ++i) xi[i] = x[i];
        EDS_UNROLL
        for (int i = 0; i < 3; ++i) tp[i] = p[i];
        EDS_UNROLL
        for (int i = 0; i < 4; ++i) tq[i] = q[i];
        edsm::se3_left_update(x, tp, tq);
        EDS_UNROLL
        for (int i = 0; i < 3; ++i) cp[i] = tp[i];
        EDS_UNROLL
        for (int i = 0; i < 4; ++i) cq[i] = tq[i];
        return true;
    }
    EDS_HD void record(double cost, int acc) {
        if (ntrace < EDS_MAX_TRACE) {
            for (int i = 0; i < 6; ++i) tr_xi[ntrace][i] = xi[i];
            tr_cost[ntrace] = cost; tr_acc[ntrace] = acc; ++ntrace;
        }
    }
    EDS_HD void finish() {          // one more pass at the accepted pose for the residuals (Tracker.cpp:223-230)
        if (skip_final && damped && have_cur) { final_cost = cur.cost; done = 1; return; }   // ... unless the caller kept them
        for (int i = 0; i < 3; ++i) cp[i] = p[i];
        for (int i = 0; i < 4; ++i) cq[i] = q[i];
        final_pass = 1;
    }
    // Consumes the sums evaluated at (cp, cq).
    EDS_HD void on_eval(const Sums6& s) {
        last_accepted = 1;          // a final pass, a Gauss-Newton pass and the first damped pass are all at the accepted pose
        if (final_pass) { final_cost = s.cost; done = 1; return; }
        if (!finite6(s)) {
            if (!have_cur) { failed = 1; done = 1; return; }
            if (!damped) { failed = 1; finish(); return; }
        }
        if (!damped) {              // Gauss-Newton: linearise here, step, repeat
            cur = s;
            if (!have_cur) { have_cur = 1; initial_cost = s.cost; }
            if (!propose()) { failed = (iter == 0); finish(); return; }
            record(s.cost, 1);
            for (int i = 0; i < 3; ++i) p[i] = cp[i];
            for (int i = 0; i < 4; ++i) q[i] = cq[i];
            if (++iter >= max_iters) finish();
            return;
        }
        if (!have_cur) {            // damped: first pass linearises at the start pose
            cur = s; have_cur = 1; initial_cost = s.cost;
        } else {
            const int ok = finite6(s) && (s.cost < cur.cost);
            last_accepted = ok;
            record(s.cost, ok);
            ++iter;
            if (ok) {
                for (int i = 0; i < 3; ++i) p[i] = cp[i];
                for (int i = 0; i < 4; ++i) q[i] = cq[i];
                cur = s; lambda *= 0.5;
            } else {
                lambda *= 4.0;
                if (lambda < 1e-6) lambda = 1e-6;
            }
        }
        if (iter >= max_iters) { finish(); return; }
        if (!propose()) { failed = (iter == 0); finish(); }
    }
};

// ---------------------------------------------------------------------------------------
template <int MAXB>
struct Sums12T {
    int nb;
    double s[MAXB];                     // ||r_block||^2
    double H[MAXB][144];                // J_b^T J_b  (Ceres-local columns, uncorrected)
    double g[MAXB][12];                 // J_b^T r_b
};
typedef Sums12T<EDS_MAX_BLOCKS> Sums12;          // host-driven loop: up to 16 residual blocks
#define EDS_DEV_MAX_BLOCKS 8                     // persistent kernel: the sums live in LDS next to the patch cache
typedef Sums12T<EDS_DEV_MAX_BLOCKS> Sums12Dev;
template <class S12>
EDS_HD void unpack12_add(const double* rec, S12* S, int k, bool first) {
    int c = 0;
    for (int a = 0; a < 12; ++a)
        for (int b = a; b < 12; ++b) {
            const double v = rec[c++];
            if (first) { S->H[k][12 * a + b] = v; S->H[k][12 * b + a] = v; }
            else { S->H[k][12 * a + b] += v; if (a != b) S->H[k][12 * b + a] += v; }
        }
    for (int a = 0; a < 12; ++a) { if (first) S->g[k][a] = rec[c]; else S->g[k][a] += rec[c]; ++c; }
    if (first) S->s[k] = rec[c]; else S->s[k] += rec[c];
}

// ceres::HuberLoss / CauchyLoss: rho(s), rho'(s)  (upstream loss_function.cc; Tracker.cpp:146-161)
EDS_HD void loss_eval(int type, double a, double s, double* rho0, double* rho1) {
    const double tiny = 2.2250738585072014e-308;
    if (type == 1) {
        const double b = a * a;
        if (s > b) { const double r = sqrt(s); *rho0 = 2.0 * a * r - b; *rho1 = (a / r > tiny) ? a / r : tiny; }
        else { *rho0 = s; *rho1 = 1.0; }
    } else if (type == 2) {
        const double b = a * a, sum = 1.0 + s / b, inv = 1.0 / sum;
        *rho0 = b * log(sum); *rho1 = inv > tiny ? inv : tiny;
    } else { *rho0 = s; *rho1 = 1.0; }
}

struct Solver12 {
    // options
    int max_iters, loss_type;
    double loss_a, ftol, gtol, ptol;
    // accepted point and candidate
    double p[3], q[4], v[6];
    double cp[3], cq[4], cv[6];
    double best_p[3], best_q[4], best_v[6];
    // linearisation at the accepted point (corrected, unscaled)
    double A[144], g[12];
    double scale[12], diagonal[12];
    double x_cost, x_norm, grad_max_norm, minimum_cost;
    double radius, decrease_factor;
    double step[12], model_cost_change;
    int reuse_diagonal, consecutive_invalid, have_scale;
    int iteration, step_successful, started;
    int done, termination, num_successful, num_unsuccessful, final_pass;
    int skip_final;                     // the caller keeps the residuals of the accepted point itself: no residual pass at the end
    double initial_cost, final_cost;

    EDS_HD void init(int max_iters_, int loss_type_, double loss_a_, double ftol_, double gtol_, double ptol_,
                     const double* p0, const double* q0, const double* v0) {
        max_iters = max_iters_; loss_type = loss_type_; loss_a = loss_a_; ftol = ftol_; gtol = gtol_; ptol = ptol_;
        for (int i = 0; i < 3; ++i) p[i] = cp[i] = best_p[i] = p0[i];
        for (int i = 0; i < 4; ++i) q[i] = cq[i] = best_q[i] = q0[i];
        for (int i = 0; i < 6; ++i) v[i] = cv[i] = best_v[i] = v0[i];
        radius = 1e4; decrease_factor = 2.0; reuse_diagonal = 0; consecutive_invalid = 0; have_scale = 0;
        iteration = 0; step_successful = 0; started = 0; done = 0; termination = TERM_FAILURE;
        num_successful = num_unsuccessful = 0; final_pass = 0; skip_final = 0;
        x_cost = x_norm = grad_max_norm = minimum_cost = initial_cost = final_cost = model_cost_change = 0.0;
    }
    EDS_HD static double norm13(const double* p_, const double* q_, const double* v_) {
        double s = 0;
        for (int i = 0; i < 3; ++i) s += p_[i] * p_[i];
        for (int i = 0; i < 4; ++i) s += q_[i] * q_[i];
        for (int i = 0; i < 6; ++i) s += v_[i] * v_[i];
        return sqrt(s);
    }
    // cost = 1/2 sum rho(s_b); optionally the corrected normal equations (rho'' <= 0 for both
    // losses, so the Ceres corrector reduces to scaling rows by sqrt(rho')).
    template <class S12>
    EDS_HD bool reduce(const S12& S, double* cost, double* A_, double* g_) const {
        double c = 0.0;
        if (A_) { for (int i = 0; i < 144; ++i) A_[i] = 0.0; for (int i = 0; i < 12; ++i) g_[i] = 0.0; }
        for (int k = 0; k < S.nb; ++k) {
            double r0, r1;
            loss_eval(loss_type, loss_a, S.s[k], &r0, &r1);
            c += 0.5 * r0;
            if (A_) {
                for (int i = 0; i < 144; ++i) A_[i] += r1 * S.H[k][i];
                for (int i = 0; i < 12; ++i) g_[i] += r1 * S.g[k][i];
            }
        }
        *cost = c;
        double t = c;
        if (A_) { for (int i = 0; i < 144; ++i) t += A_[i]; for (int i = 0; i < 12; ++i) t += g_[i]; }
        return (t == t) && (fabs(t) < 1e300);
    }
    // EvaluateGradientAndJacobian at the accepted point
    template <class S12>
    EDS_HD bool linearise(const S12& S) {
        if (!reduce(S, &x_cost, A, g)) return false;
        if (!have_scale) {
            for (int k = 0; k < 12; ++k) scale[k] = 1.0 / (1.0 + sqrt(A[13 * k]));
            have_scale = 1;
        }
        double ng[12], pp[3], pq[4], pv[6];
        for (int k = 0; k < 12; ++k) ng[k] = -g[k];
        edsm::state_plus12(p, q, v, ng, pp, pq, pv);
        double m = 0.0;
        for (int i = 0; i < 3; ++i) m = fmax(m, fabs(p[i] - pp[i]));
        for (int i = 0; i < 4; ++i) m = fmax(m, fabs(q[i] - pq[i]));
        for (int i = 0; i < 6; ++i) m = fmax(m, fabs(v[i] - pv[i]));
        grad_max_norm = m;
        return true;
    }
    EDS_HD void finish(int term) {
        termination = term;
        const bool usable = (term == TERM_CONVERGENCE || term == TERM_NO_CONVERGENCE);
        // residuals at the solution (Tracker.cpp:223-230); skipped when the solution is not usable
        if (usable && !skip_final) {
            for (int i = 0; i < 3; ++i) cp[i] = best_p[i];
            for (int i = 0; i < 4; ++i) cq[i] = best_q[i];
            for (int i = 0; i < 6; ++i) cv[i] = best_v[i];
            final_pass = 1;
        } else {
            done = 1;
        }
    }
    // FinalizeIterationAndCheckIfMinimizerCanContinue + ComputeTrustRegionStep.
    // Returns when a candidate is ready for evaluation or the solve has ended.
    EDS_HD void advance() {
        for (;;) {
            if (step_successful) {
                ++num_successful;
                if (x_cost < minimum_cost || iteration == 0) {
                    minimum_cost = x_cost;
                    for (int i = 0; i < 3; ++i) best_p[i] = p[i];
                    for (int i = 0; i < 4; ++i) best_q[i] = q[i];
                    for (int i = 0; i < 6; ++i) best_v[i] = v[i];
                }
            } else {
                ++num_unsuccessful;
            }
            if (iteration >= max_iters) { finish(TERM_NO_CONVERGENCE); return; }
            if (step_successful && grad_max_norm <= gtol) { finish(TERM_CONVERGENCE); return; }
            if (radius < 1e-32) { finish(TERM_CONVERGENCE); return; }
            ++iteration;
            step_successful = 0;
            // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system
            // (packed lower triangle, scaled on the fly: on the GPU the whole solve stays in registers)
            double L[78], y[12];
            EDS_UNROLL
            for (int a = 0; a < 12; ++a) {
                EDS_UNROLL
                for (int b = 0; b <= a; ++b) L[EDS_TRI(a, b)] = A[12 * a + b] * scale[a] * scale[b];
                y[a] = g[a] * scale[a];
            }
            if (!reuse_diagonal) {
                EDS_UNROLL
                for (int k = 0; k < 12; ++k) diagonal[k] = fmin(fmax(L[EDS_TRI(k, k)], 1e-6), 1e32);
            }
            EDS_UNROLL
            for (int k = 0; k < 12; ++k) L[EDS_TRI(k, k)] += diagonal[k] / radius;
            reuse_diagonal = 1;
            bool valid = edsm::chol_solve_packed<12>(L, y);
            if (valid) {
                double sg = 0.0, sAs = 0.0;
                EDS_UNROLL
                for (int a = 0; a < 12; ++a) step[a] = -y[a];
                EDS_UNROLL
                for (int a = 0; a < 12; ++a) {
                    sg += step[a] * g[a] * scale[a];
                    double t = 0.0;
                    EDS_UNROLL
                    for (int b = 0; b < 12; ++b) t += A[12 * a + b] * scale[b] * step[b];
                    sAs += step[a] * scale[a] * t;
                }
                model_cost_change = -sg - 0.5 * sAs;
                valid = model_cost_change > 0.0;
            }
            if (!valid) {               // HandleInvalidStep
                if (++consecutive_invalid >= 5) { finish(TERM_FAILURE); return; }
                radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
                continue;               // counts as an unsuccessful iteration
            }
            consecutive_invalid = 0;
            double delta[12];
            for (int k = 0; k < 12; ++k) delta[k] = step[k] * scale[k];
            edsm::state_plus12(p, q, v, delta, cp, cq, cv);
            return;                     // evaluate the candidate
        }
    }
    // Consumes the sums evaluated at (cp, cq, cv).
    template <class S12>
    EDS_HD void on_eval(const S12& S) {
        if (final_pass) { double c; reduce(S, &c, nullptr, nullptr); final_cost = c; done = 1; return; }
        if (!started) {                 // IterationZero
            started = 1;
            x_norm = norm13(p, q, v);
            if (!linearise(S)) { termination = TERM_FAILURE; done = 1; return; }
            initial_cost = x_cost; minimum_cost = x_cost;
            iteration = 0; step_successful = 1;
            advance();
            return;
        }
        double cand_cost;
        if (!reduce(S, &cand_cost, nullptr, nullptr)) cand_cost = 1.7976931348623157e308;
        // ParameterToleranceReached
        double sn = 0.0;
        for (int i = 0; i < 3; ++i) sn += (p[i] - cp[i]) * (p[i] - cp[i]);
        for (int i = 0; i < 4; ++i) sn += (q[i] - cq[i]) * (q[i] - cq[i]);
        for (int i = 0; i < 6; ++i) sn += (v[i] - cv[i]) * (v[i] - cv[i]);
        if (sqrt(sn) <= ptol * (x_norm + ptol)) { finish(TERM_CONVERGENCE); return; }
        // FunctionToleranceReached
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= ftol * x_cost) { finish(TERM_CONVERGENCE); return; }
        // IsStepSuccessful (monotonic steps, min_relative_decrease 1e-3)
        const double rel = cost_change / model_cost_change;
        if (rel > 1e-3) {               // HandleSuccessfulStep
            for (int i = 0; i < 3; ++i) p[i] = cp[i];
            for (int i = 0; i < 4; ++i) q[i] = cq[i];
            for (int i = 0; i < 6; ++i) v[i] = cv[i];
            x_norm = norm13(p, q, v);
            if (!linearise(S)) { finish(TERM_FAILURE); return; }
            step_successful = 1;
            const double t = 2.0 * rel - 1.0;
            radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            radius = fmin(1e16, radius);
            decrease_factor = 2.0; reuse_diagonal = 0;
        } else {                        // HandleUnsuccessfulStep
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
        }
        advance();
    }
};

}  // namespace edss
