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
    double r1[EDS_DEV_MAX_BLOCKS];   // rho' of every block's loss (the linearisation weighs the blocks' sums with it)
    double cost, rel;
    int ok, accepted;           // accepted: the evaluation just consumed became the accepted point
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

// ---- prepared steps (round 2) ----------------------------------------------------------------------------------------------
// Ceres rejects more than half of its steps on this problem (5 successful / 6 unsuccessful of 11 on the bench workload), and after
// an unsuccessful (or invalid) step the next one depends only on the linearisation at the accepted point and on the shrunken
// radius — radius / decrease_factor, decrease_factor * 2, LevenbergMarquardtStrategy::StepRejected — both known before the rejected
// step was even evaluated.  So whenever steps have to be computed, EDS_NCAND wavefronts compute them side by side for the radius
// the solver is at and for the next EDS_NCAND - 1 radii of that sequence (each on its own SIMD: the wall time of one), including
// the candidate point and its pose block.  An unsuccessful evaluation then only does the O(1) bookkeeping and moves on to the
// prepared step: linearisation, factorisation and pose block are off its critical path.  Decisions, counters, radii and the order of
// every floating-point operation are those of edss::Solver12 (which the host-driven loop and the CPU tests run) — except the back
// substitution, whose terms leave in the opposite order (last bits).
#define EDS_NCAND 4
enum { W_RETURN = 0, W_EVAL = 1, W_NEED = 2 };

struct Cand12 {                 // one prepared step
    double step[12], mcc;       // scaled step, model cost change
    double cp[3], cq[4], cv[6]; // the point it leads to
    int valid, pad;
};
struct Step12 {                 // LDS scratch of one proposing wavefront
    double L[144];              // the Cholesky factor, row-major, so that lane i can pick up COLUMN i for the back substitution
    int ok, pad;
#ifdef EDS_FUSED_STAMPS
    unsigned long long pst[8];  // diagnostic build: cycles of the pieces of coop12_propose (lane 0), summed over the calls
#endif
};

// Solver12::on_eval up to (and including) the linearisation at a newly accepted point.  Wavefront 0, all lanes.
// Returns M_RETURN (solve ended: sv.done set), M_ADVANCE (unsuccessful step: radius already shrunk, prepared steps still valid) or
// M_LIN_ITER0 / M_LIN_ACCEPT (fresh linearisation: prepared steps are stale).  `pb`: the pose block the sums were evaluated at.
// Round 5: in two halves, because the candidate groups of the team kernel (eds_fused12.hip, GROUPS > 1) take the DECISION on the block
// costs alone — sblk[b] = ||r_b||^2 of the candidate, all a rejection ever needs — and fetch the 157 sums per block only of the
// candidate that was accepted, for the linearisation.  coop12_decide = head + linearise is what one team per alignment runs.
__device__ inline int coop12_decide_head(edss::Solver12& sv, const int nb, const double* sblk, Work12& W, const int lane) {
    using namespace edss;
#ifdef EDS_FUSED_STAMPS
    if (lane == 0) W.st_t = __builtin_readcyclecounter();
#endif
    // Round 3: every lane takes the decision (same operands, read from LDS as broadcasts, all loads in flight together; the block
    // costs come by v_readlane) — on lane 0 alone this was a chain of dependent LDS round trips: r0 -> cost -> final_pass? ->
    // started? -> the 26 coordinates -> mode -> W.mode -> every lane.  Lane 0 stores what the decision changes.
    double r0 = 0.0, r1 = 1.0;
    if (lane < nb) {
        loss_eval(sv.loss_type, sv.loss_a, sblk[lane], &r0, &r1);
        W.r1[lane] = r1;                // (the linearisation weighs the blocks' sums with it)
    }
    const int final_pass = sv.final_pass, started = sv.started;
    double xa[13], xc[13];
#pragma unroll
    for (int i = 0; i < 3; ++i) { xa[i] = sv.p[i]; xc[i] = sv.cp[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xa[3 + i] = sv.q[i]; xc[3 + i] = sv.cq[i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) { xa[7 + i] = sv.v[i]; xc[7 + i] = sv.cv[i]; }
    const double x_cost = sv.x_cost, x_norm0 = sv.x_norm, mcc = sv.model_cost_change, ptol = sv.ptol, ftol = sv.ftol;
    const double radius = sv.radius, df = sv.decrease_factor;
    double c = 0.0;
    {
        const long long rb = __double_as_longlong(r0);
        const int nbu = uniform_int(nb);
        for (int k = 0; k < nbu; ++k) {
            const int lo = __builtin_amdgcn_readlane((int)(rb & 0xffffffffll), k), hi = __builtin_amdgcn_readlane((int)(rb >> 32), k);
            c += 0.5 * __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        }
    }
    // the three 13-term sums with a square root behind them (distance moved, norm of the candidate / of the start point) are
    // independent chains: formed side by side here, whichever the decision then needs
    double sn = 0.0;
#pragma unroll
    for (int i = 0; i < 13; ++i) sn += (xa[i] - xc[i]) * (xa[i] - xc[i]);
    const double moved = sqrt(sn), xn_c = Solver12::norm13(xc, xc + 3, xc + 7);
    const double cand_cost = ((c == c) && fabs(c) < 1e300) ? c : 1.7976931348623157e308;
    const double cost_change = x_cost - cand_cost;
    double rel = cost_change / mcc;     // (meaningful where it is used: after the start point has been linearised)
    int mode;
    if (final_pass) {
        mode = M_RETURN;
        if (lane == 0) { sv.final_cost = c; sv.done = 1; }
    } else if (!started) {              // IterationZero
        mode = M_LIN_ITER0;
        const double xn = Solver12::norm13(xa, xa + 3, xa + 7);
        if (lane == 0) { sv.started = 1; sv.x_norm = xn; }
    } else {
        if (moved <= ptol * (x_norm0 + ptol)) {                      // ParameterToleranceReached
            mode = M_RETURN;
            if (lane == 0) sv.finish(TERM_CONVERGENCE);
        } else if (fabs(cost_change) <= ftol * x_cost) {                // FunctionToleranceReached
            mode = M_RETURN;
            if (lane == 0) sv.finish(TERM_CONVERGENCE);
        } else {
            if (rel > 1e-3) {                                           // HandleSuccessfulStep, first half
                mode = M_LIN_ACCEPT;                                    // (x <- candidate: the 13 copies are done by 13 lanes below)
                if (lane == 0) sv.x_norm = xn_c;
            } else {                                                    // HandleUnsuccessfulStep
                mode = M_ADVANCE;
                if (lane == 0) { sv.radius = radius / df; sv.decrease_factor = df * 2.0; sv.reuse_diagonal = 1; }
            }
        }
    }
    mode = uniform_int(mode);
    if (lane == 0) { W.cost = c; W.rel = rel; W.accepted = 0; }
    EDS_WSYNC();
    EDS_CSTAMP(0);
    if (mode == M_RETURN || mode == M_ADVANCE) return mode;
    if (mode == M_LIN_ACCEPT) {          // x <- candidate point (HandleSuccessfulStep), one double per lane
        if (lane < 3) sv.p[lane] = sv.cp[lane];
        else if (lane < 7) sv.q[lane - 3] = sv.cq[lane - 3];
        else if (lane < 13) sv.v[lane - 7] = sv.cv[lane - 7];
        EDS_WSYNC();
    }
    return mode;
}

// Second half: Solver12::linearise at the (new) accepted point, from the per-block sums S of the evaluation that was accepted (W.r1, W.cost,
// W.rel: left by the head).  `pb`: the pose block those sums were evaluated at.  mode: M_LIN_ITER0 | M_LIN_ACCEPT; returns it, or M_RETURN.
template <class SUMS>              // edss::Sums12T<MAXB>: the sums of up to MAXB residual blocks (the full-cache shape of eds_fused12_kernel holds one)
__device__ inline int coop12_linearise(edss::Solver12& sv, const SUMS& S, Work12& W, const double* pb, const int mode, const int lane) {
    using namespace edss;
    const int nb = S.nb;
    {
        bool bad = !(fabs(W.cost) < 1e300);
        const int nbu = uniform_int(nb);        // (scalar loops over the blocks: nb comes out of LDS, i.e. in a vector register)
        for (int i = lane; i < 144; i += 64) {
            double a = 0.0;
            for (int k = 0; k < nbu; ++k) a += W.r1[k] * S.H[k][i];
            sv.A[i] = a;
            bad |= !(fabs(a) < 1e300);
        }
        if (lane < 12) {
            double a = 0.0;
            for (int k = 0; k < nbu; ++k) a += W.r1[k] * S.g[k][lane];
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
            return M_RETURN;
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
        if (!uniform_int(sv.have_scale)) {      // (first linearisation of a solve only)
            if (lane < 12) sv.scale[lane] = 1.0 / (1.0 + sqrt(sv.A[13 * lane]));
            EDS_WSYNC();
        }
        if (lane == 0) {
            sv.have_scale = 1;
            sv.x_cost = W.cost;
            // max |x - Plus(x, -g)| is only ever compared with the gradient tolerance (coop12_walk).  Its translation part is |g_t| as the
            // additive block forms it; when that alone is clearly above the tolerance — every linearisation of a solve that is not
            // about to end — the quaternion and velocity parts (a square root, two polynomials, a normalisation: ~1 500 cycles on
            // this lane, on the critical path of every accepted step) cannot change the comparison and are not formed.
            double ng[12], pp[3], pq[4], pv[6];
            for (int k = 0; k < 12; ++k) ng[k] = -sv.g[k];
            double m = 0.0;
            for (int i = 0; i < 3; ++i) { const double pi_ = sv.p[i]; m = fmax(m, fabs(pi_ - (pi_ + ng[i]))); }
            if (!(m > 2.0 * sv.gtol)) {
                edsm::state_plus12(sv.p, sv.q, sv.v, ng, pp, pq, pv);
                m = 0.0;
                for (int i = 0; i < 3; ++i) m = fmax(m, fabs(sv.p[i] - pp[i]));
                for (int i = 0; i < 4; ++i) m = fmax(m, fabs(sv.q[i] - pq[i]));
                for (int i = 0; i < 6; ++i) m = fmax(m, fabs(sv.v[i] - pv[i]));
            }
            sv.grad_max_norm = m;          // (a lower bound above the tolerance, or the value itself)
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
    return mode;
}

template <class SUMS>
__device__ inline int coop12_decide(edss::Solver12& sv, const SUMS& S, Work12& W, const double* pb, const int lane) {
    const int mode = coop12_decide_head(sv, S.nb, S.s, W, lane);
    if (mode == M_RETURN || mode == M_ADVANCE) return mode;
    return coop12_linearise(sv, S, W, pb, mode, lane);
}

// LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system for the radius that `ahead` further shrinkages lead to, the
// model cost change, and — for a valid step — the candidate point.  One wavefront, all lanes; reads the solver state, writes only
// `c`, its scratch `W` and (ahead == 0, first solve after a linearisation) sv.diagonal.  Lane i (< 12) holds ROW i of the lower
// triangle in registers; pivots and pivot-row entries travel by v_readlane, so the factorisation and both substitutions never
// touch LDS.  Operation order per entry is that of edsm::chol_solve_packed.
#ifdef EDS_FUSED_STAMPS
#define EDS_PT(k) do { pt_[k] = __builtin_readcyclecounter(); } while (0)
#else
#define EDS_PT(k) do { } while (0)
#endif
__device__ inline void coop12_propose(edss::Solver12& sv, const int ahead, Cand12& c, Step12& W, const int lane) {
#ifdef EDS_FUSED_STAMPS
    unsigned long long pt_[8];
#endif
    EDS_PT(0);
    double radius = sv.radius, df = sv.decrease_factor;
    for (int i = 0; i < ahead; ++i) { radius /= df; df *= 2.0; }
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
        const double dd = sv.reuse_diagonal ? sv.diagonal[row] : fmin(fmax(dg, 1e-6), 1e32);
        EDS_WSYNC();                                                 // every proposing wavefront has read reuse_diagonal / diagonal ...
        if (!sv.reuse_diagonal && ahead == 0 && lane < 12) sv.diagonal[lane] = dd;      // ... before the first one stores it
        dg += dd / radius;
#pragma unroll
        for (int b = 0; b < 12; ++b) Lr[b] = (b == row) ? dg : Lr[b];
    }
    EDS_PT(1);                      // loads, scaling, damping
    double idr = 0.0;               // 1 / L_rr of this lane's row
    bool okl = true;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const double d = bcast(Lr[j], j);
        okl = okl && (d > 0.0) && (d < 1e300);
        const double inv = edsm::rsqrt_(d);
        if (row == j) { idr = inv; Lr[j] = d * inv; }
        else Lr[j] = Lr[j] * inv;                                        // meaningful for row > j
#pragma unroll
        for (int cc = j + 1; cc < 12; ++cc) Lr[cc] -= Lr[j] * bcast(Lr[j], cc);   // L[i][j] * L[c][j]
    }
    EDS_PT(2);                      // factorisation
#pragma unroll
    for (int k = 0; k < 12; ++k) {                                       // L y = b
        if (row == k) bi = bi * idr;
        const double yk = bcast(bi, k);
        if (row > k) bi -= Lr[k] * yk;
    }
    // L^T x = y.  Entry (k, i) of L lives in lane k, the sum of row i of L^T needs it in lane i: written with v_readlane that was two
    // cross-lane moves per term inside the dependent chain (66 terms, ~4 000 cycles per proposal).  Round 3: the factor goes through
    // LDS once (12 stores and 11 loads per lane, all in flight together), lane i then holds column i, and the substitution is the
    // mirror image of the forward one: x_k is final on lane k, one broadcast, every lane i < k takes its term off.  (Terms leave
    // s_i for k = 11 down to i + 1 — the serial code subtracts them upwards; the sums differ in the last bits.)
    if (lane < 12) {
#pragma unroll
        for (int b = 0; b < 12; ++b) W.L[12 * lane + b] = Lr[b];
    }
    EDS_WSYNC();
    double Lc[12];                                                       // Lc[k] = L[k][row], k > row
#pragma unroll
    for (int k = 0; k < 12; ++k) Lc[k] = W.L[12 * k + row];
#pragma unroll
    for (int k = 11; k >= 0; --k) {
        if (row == k) bi = bi * idr;
        const double xk = bcast(bi, k);
        if (row < k) bi -= Lc[k] * xk;
    }
    EDS_PT(3);                      // both substitutions (the factor through LDS in between)
    // Round 3: the rest — A s, the model cost change, the candidate point — without a trip through LDS per piece and without a serial
    // lane: lane a (< 12) holds y_a; the step goes round by v_readlane, lane a forms t_a = sum_b A[a][b] scale[b] step[b] from its own
    // row of A, the t_a go round the same way, and EVERY lane forms the three 12-term sums and the candidate point (same operands,
    // same order as edss::Solver12::advance: the results are bit for bit what lane 0 computed alone); lane 0 stores.
    double Ar[12], sc[12], gg[12];
#pragma unroll
    for (int b = 0; b < 12; ++b) { Ar[b] = sv.A[12 * row + b]; sc[b] = sv.scale[b]; gg[b] = sv.g[b]; }      // one batch of LDS reads (scale, g: broadcasts)
    double pq_[13];
#pragma unroll
    for (int i = 0; i < 3; ++i) pq_[i] = sv.p[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) pq_[3 + i] = sv.q[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) pq_[7 + i] = sv.v[i];
    double st_[12];                 // the step, on every lane
#pragma unroll
    for (int b = 0; b < 12; ++b) st_[b] = -bcast(bi, b);
    double t = 0.0;
#pragma unroll
    for (int b = 0; b < 12; ++b) t += Ar[b] * sc[b] * st_[b];
    EDS_PT(4);                      // A s
    double chk = 0.0, sg = 0.0, sAs = 0.0;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
        chk += -st_[a];             // (= y_a)
        sg += st_[a] * gg[a] * sc[a];
        sAs += st_[a] * sc[a] * bcast(t, a);
    }
    bool valid = okl && (chk == chk) && (fabs(chk) < 1e300);
    double mcc = 0.0;
    if (valid) {
        mcc = -sg - 0.5 * sAs;
        valid = mcc > 0.0;
    }
    EDS_PT(5);                      // model cost change
    double cp_[3], cq_[4], cv_[6];
    if (valid) {
        double delta[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) delta[k] = st_[k] * sc[k];
        edsm::state_plus12(pq_, pq_ + 3, pq_ + 7, delta, cp_, cq_, cv_);
    }
    if (lane < 12) c.step[lane] = -bi;
    if (lane == 0) {
        W.ok = okl ? 1 : 0;
        c.mcc = mcc; c.valid = valid ? 1 : 0;
        if (valid) {
#pragma unroll
            for (int i = 0; i < 3; ++i) c.cp[i] = cp_[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) c.cq[i] = cq_[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) c.cv[i] = cv_[i];
        }
    }
    EDS_WSYNC();
#ifdef EDS_FUSED_STAMPS
    EDS_PT(6);                      // candidate point
    if (lane == 0) for (int k = 0; k < 6; ++k) W.pst[k] += pt_[k + 1] - pt_[k];
#endif
}

// Solver12::advance on prepared steps.  Wavefront 0, ALL lanes (round 3: every lane walks, on state read from LDS in one batch of
// broadcast loads, and lane 0 stores what changed — on lane 0 alone each test waited for its own LDS round trip).  `k` = index of the
// prepared step that belongs to the radius the solver is at (EDS_NCAND or more: none prepared), `head_done`: the per-iteration
// bookkeeping of advance() has already run for the step about to be taken (the walk was interrupted to have steps prepared).
// Returns {W_RETURN (solve ended) | W_EVAL (take prepared step k: coop12_take) | W_NEED (steps for the current radius are missing), k,
// head_done}, the same on every lane.
struct Walk12 { int walk, k, head; };
__device__ inline Walk12 coop12_walk(edss::Solver12& sv, const Cand12* cand, int k, int head_done, const int lane) {
    using namespace edss;
    int step_successful = sv.step_successful, num_s = sv.num_successful, num_u = sv.num_unsuccessful;
    int iteration = sv.iteration, cinv = sv.consecutive_invalid;
    const int max_iters = sv.max_iters;
    const double x_cost = sv.x_cost, gmn = sv.grad_max_norm, gtol = sv.gtol;
    double minimum_cost = sv.minimum_cost, radius = sv.radius, df = sv.decrease_factor;
    int valid[EDS_NCAND];
#pragma unroll
    for (int i = 0; i < EDS_NCAND; ++i) valid[i] = cand[i].valid;
    bool copy_best = false, touched_radius = false, reuse = false;
    int term = -1, res;
    for (;;) {
        if (!head_done) {
            if (step_successful) {
                ++num_s;
                if (x_cost < minimum_cost || iteration == 0) { minimum_cost = x_cost; copy_best = true; }
            } else {
                ++num_u;
            }
            if (iteration >= max_iters) { term = TERM_NO_CONVERGENCE; res = W_RETURN; break; }
            if (step_successful && gmn <= gtol) { term = TERM_CONVERGENCE; res = W_RETURN; break; }
            if (radius < 1e-32) { term = TERM_CONVERGENCE; res = W_RETURN; break; }
            ++iteration; step_successful = 0;
            head_done = 1;
        }
        if (k >= EDS_NCAND) { res = W_NEED; break; }
        reuse = true;
        int vk = valid[0];
#pragma unroll
        for (int i = 1; i < EDS_NCAND; ++i) vk = (k == i) ? valid[i] : vk;
        if (!vk) {                      // HandleInvalidStep
            if (++cinv >= 5) { term = TERM_FAILURE; res = W_RETURN; break; }
            radius /= df; df *= 2.0; touched_radius = true;
            ++k; head_done = 0;
            continue;                   // counts as an unsuccessful iteration
        }
        cinv = 0;
        head_done = 0;
        res = W_EVAL;                   // the step itself is taken by coop12_take (26 doubles: one per lane instead of one after the other)
        break;
    }
    copy_best = __builtin_amdgcn_readfirstlane(copy_best ? 1 : 0) != 0;
    if (copy_best) {                    // the accepted point is the best one so far: 13 doubles, one per lane
        if (lane < 3) sv.best_p[lane] = sv.p[lane];
        else if (lane < 7) sv.best_q[lane - 3] = sv.q[lane - 3];
        else if (lane < 13) sv.best_v[lane - 7] = sv.v[lane - 7];
    }
    if (lane == 0) {
        sv.step_successful = step_successful; sv.num_successful = num_s; sv.num_unsuccessful = num_u;
        sv.iteration = iteration; sv.consecutive_invalid = cinv; sv.minimum_cost = minimum_cost;
        if (reuse) sv.reuse_diagonal = 1;
        if (touched_radius) { sv.radius = radius; sv.decrease_factor = df; }
    }
    term = uniform_int(term);
    if (term >= 0) {                    // (finish() reads the best point: after the copy above)
        EDS_WSYNC();
        if (lane == 0) sv.finish(term);
    }
    EDS_WSYNC();
    return Walk12{uniform_int(res), uniform_int(k), uniform_int(head_done)};
}

// Second half of the W_EVAL case of coop12_walk: the prepared step `c` becomes the solver's step and candidate point.  Wavefront 0, all
// lanes, after lane 0's coop12_walk returned W_EVAL (and a wave-level sync).  On lane 0 these 26 LDS-to-LDS copies ran one after the
// other, each waiting for its load: ~2 000 cycles of every evaluation.
__device__ inline void coop12_take(edss::Solver12& sv, const Cand12& c, const int lane) {
    if (lane < 12) sv.step[lane] = c.step[lane];
    else if (lane == 12) sv.model_cost_change = c.mcc;
    else if (lane < 16) sv.cp[lane - 13] = c.cp[lane - 13];
    else if (lane < 20) sv.cq[lane - 16] = c.cq[lane - 16];
    else if (lane < 26) sv.cv[lane - 20] = c.cv[lane - 20];
    EDS_WSYNC();
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
    // (reciprocal square roots and multiplications instead of the serial code's sqrt + divisions: every fp64 division or square root
    // is a 20-30 instruction dependent sequence on the solver's critical path; the results agree to the last bits and feed fp32 kernels)
    const double inv_vn = edsm::rsqrt_(vv), inv_vv = inv_vn * inv_vn;
    if (lane < 36) {
        const int i = lane / 6, j = lane - 6 * i;
        pb[EDS_PB_PV + lane] = ((i == j ? 1.0 : 0.0) - v[i] * v[j] * inv_vv) * inv_vn;
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
        const double inv_n = edsm::rsqrt_(S);
        o[1 + i6] = raw[i6] * (inv_n * inv_n * inv_n);
        if (i6 == 0) { o[0] = inv_n; o[7] = S; }
    }
    EDS_WSYNC();
}

}  // namespace edsc
