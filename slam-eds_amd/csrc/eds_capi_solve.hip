// C ABI of libeds_hip.so, third translation unit: the passes and the solves.  eds_trk_eval and the host-driven loop (EDS_EXEC_HOST:
// north-star's literal structure — residual/Jacobian kernel, reduction kernel, 6x6 / 12x12 step on the host) live here; a device solve
// (EDS_EXEC_DEVICE) goes on to eds_fused_solve.  Replaces reference Tracker.cpp:104-241 (optimize) and :281-317 (getLossParams).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "eds_capi_internal.hpp"

using namespace edscapi;

namespace {

// geometry of the reduction grid for `count` slots with at most N points each
// (a segment = one workgroup's record: 256 points, or 1 024 for the 6-column pass, whose lanes fold four points each — eds_kernels.hpp)
void reduce_geometry(int N, int nb_red, int ncols, int* cpb, int* nseg, int ppl_knob = 4) {
    const int ne = N / nb_red;
    const int last = ne + (N - nb_red * ne);
    const int per_seg = EDS_TPB * eds_reduce_points_per_lane(ncols, nb_red, ppl_knob);
    *cpb = std::max(1, (last + per_seg - 1) / per_seg);
    *nseg = nb_red * (*cpb);
}

// One residual/Jacobian pass + reduction over slots [first, first+count) at the poses currently
// in h_pose; brings the partial sums back to h_part.
// The streaming residual/Jacobian kernel samples the strip copies of the frames (eds_layout.hpp) when they are worth making: batches
// (every frame is touched by every pass of a host-driven solve or of a benchmark loop), or whenever a solve has made them already.
// Returns the arrays with `strips` set only if the copies of this range are current.
static EdsArrays arrays_for_pass(eds_trk* h, int first, int count) {
    EdsArrays A = h->arrays();
    bool ok = false;
    if (h->tiled && h->cfg.sampling == EDS_SAMPLE_BICUBIC && h->H < 8000) {
        // (a stand-alone pass never has the copies MADE: they cost ~60 passes' worth of what a pass gains from them — it uses the
        // ones a solve or eds_trk_prepare_frames left behind)
        if (!h->knobs.layout_tiles) ok = eds_strips_current(h, first, count);
    }
    A.strips = ok ? h->dstrips : nullptr;
    A.strip_phases = h->strip_phases;
    return A;
}

int run_pass(eds_trk* h, int first, int count, int ncols, bool refresh_model, bool with_reduction, bool fetch) {
    const EdsArrays A = arrays_for_pass(h, first, count);
    const int N = max_points(h, first, count);
    if (N <= 0) return fail(EDS_ERR_STATE, "no keyframe set");
    const int nchunk = (N + EDS_TPB - 1) / EDS_TPB;
    int rc = upload_pose(h, first, count);
    if (rc) return rc;
    if (refresh_model && ncols == 6) eds_launch_model(A, first, count, nchunk, h->st);
    eds_launch_resjac(A, h->cfg.sampling, ncols, first, count, nchunk, h->st);
    if (ncols == 12 && h->cfg.nc) eds_launch_nc_normalise(A, first, count, effective_blocks(h), nchunk, h->st);
    if (with_reduction) {
        const int nb_red = (ncols == 12) ? effective_blocks(h) : 1;
        int cpb, nseg;
        reduce_geometry(N, nb_red, ncols, &cpb, &nseg, h->knobs.reduce_ppl);
        if (nseg > h->max_seg) return fail(EDS_ERR_INVALID, "reduction grid exceeds allocation");
        eds_launch_reduce(A, ncols, first, count, nseg, nb_red, cpb, h->st, h->knobs.reduce_ppl);
        if (fetch)
            EDS_HIP_TRY(hipMemcpyAsync(h->h_part + (size_t)first * h->max_seg * EDS_RED_K,
                                       h->dpart + (size_t)first * h->max_seg * EDS_RED_K,
                                       sizeof(double) * h->max_seg * EDS_RED_K * count, hipMemcpyDeviceToHost, h->st));
    }
    EDS_HIP_TRY(hipGetLastError());
    if (fetch) EDS_HIP_TRY(hipStreamSynchronize(h->st));
    return EDS_OK;
}

// sums of a slot after run_pass (host side, fp64)
void gather6(const eds_trk* h, int slot, edss::Sums6* S) {
    int cpb, nseg;
    reduce_geometry(h->slots[slot].N, 1, 6, &cpb, &nseg, h->knobs.reduce_ppl);
    // NB: the grid was sized for the max N of the range; segments beyond this slot's own are all-zero
    double rec[EDS_RED_N6];
    for (int i = 0; i < EDS_RED_N6; ++i) rec[i] = 0.0;
    const double* base = h->h_part + (size_t)slot * h->max_seg * EDS_RED_K;
    for (int s = 0; s < nseg; ++s)
        for (int i = 0; i < EDS_RED_N6; ++i) rec[i] += base[(size_t)s * EDS_RED_K + i];
    edss::unpack6(rec, S);
}
void gather12(const eds_trk* h, int slot, int range_max_N, edss::Sums12* S) {
    const int nb = effective_blocks(h);
    int cpb, nseg;
    reduce_geometry(range_max_N, nb, 12, &cpb, &nseg);
    S->nb = nb;
    const double* base = h->h_part + (size_t)slot * h->max_seg * EDS_RED_K;
    for (int k = 0; k < nb; ++k)
        for (int c = 0; c < cpb; ++c) edss::unpack12_add(base + (size_t)(k * cpb + c) * EDS_RED_K, S, k, c == 0);
    // The kernels emit the velocity columns WITHOUT the local-parameterisation factor Pv = (I - v v^T/|v|^2)/|v| (the same for
    // every point): apply it here, once and in fp64, J^T J -> P^T (J^T J) P, J^T r -> P^T (J^T r), P = blockdiag(I_6, Pv).
    // (Done per point in fp32 it left 1e-7-level noise along v, which the weakly determined velocity block amplified.)
    const double* Pv = h->h_pose + (size_t)slot * EDS_POSE_STRIDE + EDS_PB_PV;
    for (int k = 0; k < nb; ++k) {
        double T[144];
        double* H = S->H[k];
        for (int i = 0; i < 12; ++i)
            for (int j = 0; j < 12; ++j) {
                double t = H[12 * i + j];
                if (j >= 6) { t = 0.0; for (int c = 0; c < 6; ++c) t += H[12 * i + 6 + c] * Pv[6 * c + (j - 6)]; }
                T[12 * i + j] = t;
            }
        for (int i = 0; i < 12; ++i)
            for (int j = 0; j < 12; ++j) {
                double t = T[12 * i + j];
                if (i >= 6) { t = 0.0; for (int c = 0; c < 6; ++c) t += Pv[6 * c + (i - 6)] * T[12 * (6 + c) + j]; }
                H[12 * i + j] = t;
            }
        double g6[6];
        for (int i = 0; i < 6; ++i) { g6[i] = 0.0; for (int c = 0; c < 6; ++c) g6[i] += Pv[6 * c + i] * S->g[k][6 + c]; }
        for (int i = 0; i < 6; ++i) S->g[k][6 + i] = g6[i];
    }
}

int fetch_residuals(eds_trk* h, int first, int count) {
    EDS_HIP_TRY(hipMemcpyAsync(h->h_r + (size_t)first * h->Np, h->dr + (size_t)first * h->Np,
                               sizeof(float) * h->Np * count, hipMemcpyDeviceToHost, h->st));
    EDS_HIP_TRY(hipStreamSynchronize(h->st));
    for (int s = first; s < first + count; ++s) {
        Slot& sl = h->slots[s];
        sl.residuals.resize(sl.N);
        const float* r = h->h_r + (size_t)s * h->Np;
        for (int i = 0; i < sl.N; ++i) sl.residuals[i] = r[i];
    }
    return EDS_OK;
}

void store_trace(Slot& sl, const edss::Solver6& sv) {
    sl.ntrace = sv.ntrace;
    sl.tr_xi.assign(&sv.tr_xi[0][0], &sv.tr_xi[0][0] + 6 * sv.ntrace);
    sl.tr_cost.assign(sv.tr_cost, sv.tr_cost + sv.ntrace);
    sl.tr_acc.assign(sv.tr_acc, sv.tr_acc + sv.ntrace);
}

}  // namespace

namespace edscapi {

// Host-driven lockstep solve of slots [first, first+count).
int solve_host(eds_trk* h, int level, int first, int count) {
    const auto t0 = std::chrono::steady_clock::now();
    const int iters = level_iters(h, level);
    const bool ref12 = h->cfg.solver == EDS_SOLVER_REF12;
    for (int s = first; s < first + count; ++s)
        if (!h->slots[s].has_kf || !h->slots[s].has_frame) return fail(EDS_ERR_STATE, "keyframe or event frame not set");
    const int rangeN = max_points(h, first, count);
    std::vector<char> active(count, 1);
    int nactive = count, rc = EDS_OK;
    bool first_pass = true;
    if (!ref12) {
        std::vector<edss::Solver6> sv(count);
        for (int i = 0; i < count; ++i) {
            Slot& sl = h->slots[first + i];
            sv[i].init(h->cfg.solver == EDS_SOLVER_LM6, iters, h->cfg.lambda0, sl.p, sl.q);
            fill_pose(h, first + i, sv[i].cp, sv[i].cq, sl.v);
        }
        while (nactive > 0) {
            if ((rc = run_pass(h, first, count, 6, first_pass, true, true))) return rc;
            first_pass = false;
            for (int i = 0; i < count; ++i) {
                if (!active[i]) continue;
                edss::Sums6 S;
                gather6(h, first + i, &S);
                sv[i].on_eval(S);
                if (sv[i].done) { active[i] = 0; --nactive; }
                else fill_pose(h, first + i, sv[i].cp, sv[i].cq, h->slots[first + i].v);
            }
        }
        if ((rc = fetch_residuals(h, first, count))) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < count; ++i) {
            Slot& sl = h->slots[first + i];
            const bool ok = !sv[i].failed;
            if (ok) { std::memcpy(sl.p, sv[i].p, sizeof(sl.p)); std::memcpy(sl.q, sv[i].q, sizeof(sl.q)); }
            store_trace(sl, sv[i]);
            sl.res_on_device = false; sl.trace_on_device = false;
            eds_trk_info& in = sl.info;
            std::memset(&in, 0, sizeof(in));
            in.meas_time_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
            in.time_seconds = in.meas_time_us * 1e-6;
            in.num_points = sl.N;
            in.num_iterations = sv[i].iter;
            in.success = ok;
            in.termination = ok ? edss::TERM_NO_CONVERGENCE : edss::TERM_FAILURE;
            in.num_successful_steps = 0;
            for (int k = 0; k < sv[i].ntrace; ++k) in.num_successful_steps += sv[i].tr_acc[k];
            in.num_unsuccessful_steps = sv[i].ntrace - in.num_successful_steps;
            in.initial_cost = 0.5 * sv[i].initial_cost;
            in.final_cost = 0.5 * sv[i].final_cost;
        }
        return EDS_OK;
    }
    // reference problem: Ceres-style LM over 12 local parameters
    std::vector<edss::Solver12> sv(count);
    std::vector<edss::Sums12>* S = new (std::nothrow) std::vector<edss::Sums12>(1);
    if (!S) return fail(EDS_ERR_INVALID, "out of memory");
    for (int i = 0; i < count; ++i) {
        Slot& sl = h->slots[first + i];
        sv[i].init(iters, h->cfg.loss_type, h->cfg.loss_param, h->cfg.function_tolerance, h->cfg.gradient_tolerance,
                   h->cfg.parameter_tolerance, sl.p, sl.q, sl.v);
        fill_pose(h, first + i, sv[i].cp, sv[i].cq, sv[i].cv);
    }
    while (nactive > 0) {
        if ((rc = run_pass(h, first, count, 12, false, true, true))) { delete S; return rc; }
        for (int i = 0; i < count; ++i) {
            if (!active[i]) continue;
            gather12(h, first + i, rangeN, &(*S)[0]);
            sv[i].on_eval((*S)[0]);
            if (sv[i].done) { active[i] = 0; --nactive; }
            else fill_pose(h, first + i, sv[i].cp, sv[i].cq, sv[i].cv);
        }
    }
    delete S;
    if ((rc = fetch_residuals(h, first, count))) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < count; ++i) {
        Slot& sl = h->slots[first + i];
        const bool ok = sv[i].termination != edss::TERM_FAILURE;
        if (ok) {
            std::memcpy(sl.p, sv[i].best_p, sizeof(sl.p)); std::memcpy(sl.q, sv[i].best_q, sizeof(sl.q));
            std::memcpy(sl.v, sv[i].best_v, sizeof(sl.v));
        } else {
            sl.residuals.clear();
        }
        sl.ntrace = 0;
        sl.res_on_device = false; sl.trace_on_device = false;
        eds_trk_info& in = sl.info;
        std::memset(&in, 0, sizeof(in));
        in.meas_time_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        in.time_seconds = in.meas_time_us * 1e-6;
        in.num_points = sl.N;
        in.num_successful_steps = sv[i].num_successful;
        in.num_unsuccessful_steps = sv[i].num_unsuccessful;
        in.num_iterations = sv[i].num_successful + sv[i].num_unsuccessful;   // Tracker.cpp:211
        in.success = ok;
        in.termination = sv[i].termination;
        in.initial_cost = sv[i].initial_cost;
        in.final_cost = sv[i].minimum_cost;
    }
    return EDS_OK;
}

// residuals of a device-mode solve stay in HBM until somebody asks for them
int materialise_residuals(eds_trk* h, int slot) {
    Slot& s = h->slots[slot];
    if (!s.res_on_device) return EDS_OK;
    if (s.res_in_hostmap) {             // the kernel left a copy in pinned host memory: no HIP call at all (a solve seen complete through its
        const float* r = h->h_rmap + (size_t)slot * h->Np;      // workgroups' done words has the mirror behind those words: eds_capi.hip wait_stream)
        s.residuals.resize(s.N);
        for (int i = 0; i < s.N; ++i) s.residuals[i] = r[i];
        s.res_on_device = false; s.res_in_hostmap = false;
        return EDS_OK;
    }
    EDS_HIP_TRY(hipSetDevice(h->dev));
    int rc = fetch_residuals(h, slot, 1);
    if (rc) return rc;
    s.res_on_device = false;
    return EDS_OK;
}

}  // namespace edscapi

extern "C" {

int eds_trk_eval(eds_trk* h, int slot, const double p[3], const double q[4], const double v[6], int ncols, double* r,
                 double* J, double* JtJ, double* Jtr, double* cost) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (ncols != 6 && ncols != 12) return fail(EDS_ERR_INVALID, "ncols must be 6 or 12");
    if (ncols == 6 && h->cfg.nc) return fail(EDS_ERR_INVALID, "the NC residual (cfg.nc) has 12-column rows only");
    if (!p || !q || !v) return fail(EDS_ERR_INVALID, "null state");
    Slot& s = h->slots[slot];
    if (!s.has_kf || !s.has_frame) return fail(EDS_ERR_STATE, "keyframe or event frame not set");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    fill_pose(h, slot, p, q, v);
    if ((rc = run_pass(h, slot, 1, ncols, true, true, true))) return rc;
    const int N = s.N;
    if (r) {
        EDS_HIP_TRY(hipMemcpy(h->h_r + (size_t)slot * h->Np, h->dr + (size_t)slot * h->Np, (size_t)N * 4, hipMemcpyDeviceToHost));
        const float* src = h->h_r + (size_t)slot * h->Np;
        for (int i = 0; i < N; ++i) r[i] = src[i];
    }
    if (J) {
        const size_t plane = (size_t)h->B * h->Np;
        for (int k = 0; k < ncols; ++k) {
            EDS_HIP_TRY(hipMemcpy(h->h_f32, h->dJ + k * plane + (size_t)slot * h->Np, (size_t)N * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < N; ++i) J[(size_t)i * ncols + k] = h->h_f32[i];
        }
        if (ncols == 12) {               // velocity columns: the local-parameterisation factor, in fp64 (see gather12)
            const double* Pv = h->h_pose + (size_t)slot * EDS_POSE_STRIDE + EDS_PB_PV;
            for (int i = 0; i < N; ++i) {
                double* row = J + (size_t)i * 12 + 6;
                double out[6];
                for (int c = 0; c < 6; ++c) { out[c] = 0.0; for (int k = 0; k < 6; ++k) out[c] += row[k] * Pv[6 * k + c]; }
                for (int c = 0; c < 6; ++c) row[c] = out[c];
            }
        }
    }
    if (ncols == 6) {
        edss::Sums6 S;
        gather6(h, slot, &S);
        if (JtJ) std::memcpy(JtJ, S.H, sizeof(S.H));
        if (Jtr) std::memcpy(Jtr, S.b, sizeof(S.b));
        if (cost) *cost = 0.5 * S.cost;
    } else {
        edss::Sums12* S = new edss::Sums12();
        gather12(h, slot, N, S);
        if (JtJ) { for (int i = 0; i < 144; ++i) { JtJ[i] = 0; for (int k = 0; k < S->nb; ++k) JtJ[i] += S->H[k][i]; } }
        if (Jtr) { for (int i = 0; i < 12; ++i) { Jtr[i] = 0; for (int k = 0; k < S->nb; ++k) Jtr[i] += S->g[k][i]; } }
        if (cost) { double c = 0; for (int k = 0; k < S->nb; ++k) c += S->s[k]; *cost = 0.5 * c; }
        delete S;
    }
    return EDS_OK;
}

static int solve_range(eds_trk* h, int level, int first, int count) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (first < 0 || count < 1 || first + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    if (h->cfg.nc && h->cfg.solver != EDS_SOLVER_REF12)
        return fail(EDS_ERR_INVALID, "the NC residual (cfg.nc) is defined for EDS_SOLVER_REF12 only");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if (h->cfg.exec == EDS_EXEC_DEVICE) return eds_fused_solve(h, level, first, count);
    return solve_host(h, level, first, count);
}

int eds_trk_optimize(eds_trk* h, int slot, int level, double p[3], double q[4], double v[6], eds_trk_info* info) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    double sp[3], sq[4], svv[6];
    std::memcpy(sp, s.p, sizeof(sp)); std::memcpy(sq, s.q, sizeof(sq)); std::memcpy(svv, s.v, sizeof(svv));
    if (p) std::memcpy(s.p, p, sizeof(s.p));
    if (q) std::memcpy(s.q, q, sizeof(s.q));
    if (v) std::memcpy(s.v, v, sizeof(s.v));
    rc = solve_range(h, level, slot, 1);
    if (rc == EDS_OK && h->cfg.exec == EDS_EXEC_DEVICE) rc = eds_trk_sync(h);
    if (rc != EDS_OK) {                 // leave everything at its pre-call value
        std::memcpy(s.p, sp, sizeof(sp)); std::memcpy(s.q, sq, sizeof(sq)); std::memcpy(s.v, svv, sizeof(svv));
        return rc;
    }
    if (info) *info = s.info;
    if (!s.info.success) {              // Tracker.cpp:236-239: nothing is updated
        std::memcpy(s.p, sp, sizeof(sp)); std::memcpy(s.q, sq, sizeof(sq)); std::memcpy(s.v, svv, sizeof(svv));
        return fail(EDS_ERR_NOT_USABLE, "solution not usable");
    }
    if (p) std::memcpy(p, s.p, sizeof(s.p));
    if (q) std::memcpy(q, s.q, sizeof(s.q));
    if (v) std::memcpy(v, s.v, sizeof(s.v));
    return EDS_OK;
}

int eds_trk_optimize_batch(eds_trk* h, int level, int first, int count) { return solve_range(h, level, first, count); }
int eds_trk_optimize_batch_wait(eds_trk* h, int level, int first, int count) {
    const int rc = solve_range(h, level, first, count);
    return rc != EDS_OK ? rc : eds_trk_sync(h);
}

int eds_trk_get_residuals(eds_trk* h, int slot, double* r) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!r) return fail(EDS_ERR_INVALID, "null output");
    if ((rc = materialise_residuals(h, slot))) return rc;
    const Slot& s = h->slots[slot];
    if ((int)s.residuals.size() != s.N) return fail(EDS_ERR_STATE, "no residuals stored (no usable solve yet)");
    std::memcpy(r, s.residuals.data(), sizeof(double) * s.N);
    return EDS_OK;
}

int eds_trk_loss_param(eds_trk* h, int slot, int method, double* tau) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!tau) return fail(EDS_ERR_INVALID, "null output");
    Slot& s = h->slots[slot];
    if (method == EDS_LP_CONSTANT) return EDS_OK;
    // One slot: bring the residuals over (8 KB; the caller wants them for kf->residuals anyway, Tracker.cpp:223-230) and
    // select on the host — 40 us against 80 us for the LDS sort of a single alignment.  Batches use
    // eds_trk_loss_param_batch, which selects on the device (0.5 us per alignment).
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if ((rc = materialise_residuals(h, slot))) return rc;
    if ((int)s.residuals.size() != s.N || s.N < 1) return fail(EDS_ERR_STATE, "no residuals stored");
    std::vector<double>& r = s.residuals;
    if (method == EDS_LP_MAD) {         // Tracker.cpp:292-305 incl. the in-place partial reorder
        const size_t n = r.size() / 2;
        std::nth_element(r.begin(), r.begin() + n, r.end());
        const double median = r[n];
        std::vector<double>& am = h->scratch;           // (no allocation per call on the live path)
        am.resize(r.size());
        for (size_t i = 0; i < r.size(); ++i) am[i] = std::fabs(r[i] - median);
        const size_t m = am.size() / 2;
        std::nth_element(am.begin(), am.begin() + m, am.end());
        *tau = 1.345 * (1.4826 * am[m]);
        return EDS_OK;
    }
    if (method == EDS_LP_STD) {         // Tracker.cpp:306-314; mean_std_vector returns the variance (Utils.hpp:272-290)
        const size_t sz = r.size();
        if (sz == 1) { *tau = 0.0; return EDS_OK; }
        double mu = 0.0;
        for (double x : r) mu += x;
        mu /= (double)sz;
        double var = 0.0;
        for (double x : r) var += (x - mu) * (x - mu) / (double)(sz - 1);
        *tau = 1.345 * var;
        return EDS_OK;
    }
    return fail(EDS_ERR_INVALID, "unknown loss-param method");
}

// Tracker.cpp:223-233 in one call: kf->residuals <- the residuals at the solution, config.loss_params <- getLossParams(method) — whose
// MAD selection partially reorders kf->residuals in place (n_quantile_vector, Utils.hpp:316-319).  `r` receives the residuals as that
// sequence leaves them; one read-back instead of get_residuals -> loss_param -> get_residuals.
int eds_trk_residuals_and_loss(eds_trk* h, int slot, int method, double* r, double* tau) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!r || !tau) return fail(EDS_ERR_INVALID, "null output");
    if ((rc = materialise_residuals(h, slot))) return rc;
    const Slot& s = h->slots[slot];
    if ((int)s.residuals.size() != s.N) return fail(EDS_ERR_STATE, "no residuals stored (no usable solve yet)");
    if (method != EDS_LP_CONSTANT && (rc = eds_trk_loss_param(h, slot, method, tau))) return rc;
    std::memcpy(r, s.residuals.data(), sizeof(double) * s.N);
    return EDS_OK;
}

int eds_trk_bench_live(eds_trk* h, int slot, int level, const double* idp, const double* frame, const double p0[3], const double q0[4],
                       const double v0[6], int method, int reps, double out_us[6]) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!p0 || !q0 || !v0 || !out_us || reps < 1) return fail(EDS_ERR_INVALID, "null state / output or reps < 1");
    const int N = h->slots[slot].N;
    std::vector<double> t[6], res((size_t)(N > 0 ? N : 1));
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int r = 0; r < reps; ++r) {
        double p[3], q[4], v[6], tau = 0.0;
        std::memcpy(p, p0, sizeof(p)); std::memcpy(q, q0, sizeof(q)); std::memcpy(v, v0, sizeof(v));
        eds_trk_info info;
        const clk::time_point a = clk::now();
        if (idp && (rc = eds_trk_set_idepth(h, slot, N, idp))) return rc;
        const clk::time_point b = clk::now();
        if (frame && (rc = eds_trk_set_event_frame(h, slot, frame))) return rc;
        const clk::time_point c = clk::now();
        rc = eds_trk_optimize(h, slot, level, p, q, v, &info);
        if (rc != EDS_OK && rc != EDS_ERR_NOT_USABLE) return rc;
        const clk::time_point d = clk::now();
        if (method >= 0 && rc == EDS_OK && (rc = eds_trk_residuals_and_loss(h, slot, method, res.data(), &tau))) return rc;
        const clk::time_point e = clk::now();
        t[0].push_back(us(a, e)); t[1].push_back(us(a, b)); t[2].push_back(us(b, c)); t[3].push_back(us(c, d)); t[4].push_back(us(d, e));
        t[5].push_back(info.device_time_us);
    }
    for (int k = 0; k < 6; ++k) {
        std::nth_element(t[k].begin(), t[k].begin() + t[k].size() / 2, t[k].end());
        out_us[k] = t[k][t[k].size() / 2];
    }
    return EDS_OK;
}

int eds_trk_bench_batch(eds_trk* h, int level, int first, int count, const double* p, const double* q, const double* v, int reps, double out_us[5]) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (!p || !q || !v || !out_us || reps < 1) return fail(EDS_ERR_INVALID, "null state / output or reps < 1");
    std::vector<double> t[4];
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    double worst = 0.0;
    for (int r = 0; r < reps; ++r) {
        const clk::time_point a = clk::now();
        if ((rc = eds_trk_set_states(h, first, count, p, q, v))) return rc;
        const clk::time_point b = clk::now();
        if ((rc = eds_trk_optimize_batch_wait(h, level, first, count))) return rc;
        const clk::time_point c = clk::now();
        t[0].push_back(us(a, c)); t[1].push_back(us(a, b)); t[2].push_back(us(b, c)); t[3].push_back(h->slots[first].info.device_time_us);
        worst = std::max(worst, us(a, c));
    }
    for (int k = 0; k < 4; ++k) {
        std::nth_element(t[k].begin(), t[k].begin() + t[k].size() / 2, t[k].end());
        out_us[k] = t[k][t[k].size() / 2];
    }
    out_us[4] = worst;
    return EDS_OK;
}

int eds_trk_bench_eval(eds_trk* h, int first, int count, int ncols, int with_reduction, int reps, float* mean_ms) {
    if (!h || !mean_ms) return fail(EDS_ERR_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > h->B || reps < 1) return fail(EDS_ERR_INVALID, "bad range");
    if (ncols != 6 && ncols != 12) return fail(EDS_ERR_INVALID, "ncols must be 6 or 12");
    if (ncols == 6 && h->cfg.nc) return fail(EDS_ERR_INVALID, "the NC residual (cfg.nc) has 12-column rows only");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    for (int s = first; s < first + count; ++s) {
        const Slot& sl = h->slots[s];
        if (!sl.has_kf || !sl.has_frame) return fail(EDS_ERR_STATE, "keyframe or event frame not set");
        fill_pose(h, s, sl.p, sl.q, sl.v);
    }
    int rc = run_pass(h, first, count, ncols, true, with_reduction != 0, false);   // warm-up + model
    if (rc) return rc;
    EDS_HIP_TRY(hipStreamSynchronize(h->st));
    const EdsArrays A = arrays_for_pass(h, first, count);
    const int N = max_points(h, first, count);
    const int nchunk = (N + EDS_TPB - 1) / EDS_TPB;
    const int nb_red = (ncols == 12) ? effective_blocks(h) : 1;
    int cpb, nseg;
    reduce_geometry(N, nb_red, ncols, &cpb, &nseg, h->knobs.reduce_ppl);
    EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i) {
        eds_launch_resjac(A, h->cfg.sampling, ncols, first, count, nchunk, h->st);
        if (with_reduction) eds_launch_reduce(A, ncols, first, count, nseg, nb_red, cpb, h->st, h->knobs.reduce_ppl);
    }
    EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
    EDS_HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    EDS_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    EDS_HIP_TRY(hipGetLastError());
    *mean_ms = ms / (float)reps;
    return EDS_OK;
}

// ---- measurement helpers of bench.py (round 5) -------------------------------------------------------------------------------------
// A plain streaming kernel of the library's own: 16 bytes per lane, grid-stride, 8 192 workgroups of 256 threads — what the box's HBM
// gives a coalesced read (mode 0), a copy (1).  Also the EVICTION pass of the cold timings below: 1 GiB through the caches.
__global__ __launch_bounds__(256) void eds_probe_kernel(const float4* __restrict__ a, float4* __restrict__ b, float* __restrict__ sink, size_t n, int mode) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        if (mode == 1) b[i] = v; else acc += v.x + v.y + v.z + v.w;
    }
    if (mode == 0 && acc == 12345.678f) sink[0] = acc;
}

static int probe_buffers(eds_trk* h, size_t bytes) {
    if (h->probe_bytes >= bytes) return EDS_OK;
    if (h->d_probe) { hipFree(h->d_probe); h->d_probe = nullptr; h->probe_bytes = 0; }
    EDS_HIP_TRY(hipMalloc(&h->d_probe, 2 * bytes + 256));
    EDS_HIP_TRY(hipMemsetAsync(h->d_probe, 0, 2 * bytes + 256, h->st));
    h->probe_bytes = bytes;
    return EDS_OK;
}

int eds_trk_hbm_probe(eds_trk* h, size_t bytes, int reps, float* read_GBps, float* copy_GBps) {
    if (!h || !read_GBps || !copy_GBps) return fail(EDS_ERR_INVALID, "null argument");
    if (bytes < (1u << 20) || reps < 1) return fail(EDS_ERR_INVALID, "bad size");
    bytes &= ~(size_t)255;
    EDS_HIP_TRY(hipSetDevice(h->dev));
    int rc = probe_buffers(h, bytes);
    if (rc) return rc;
    const float4* a = reinterpret_cast<const float4*>(h->d_probe);
    float4* b = reinterpret_cast<float4*>(reinterpret_cast<char*>(h->d_probe) + bytes);
    float* sink = reinterpret_cast<float*>(reinterpret_cast<char*>(h->d_probe) + 2 * bytes);
    const size_t n = bytes / 16;
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(eds_probe_kernel, dim3(8192), dim3(256), 0, h->st, a, b, sink, n, mode);      // warm-up
        EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(eds_probe_kernel, dim3(8192), dim3(256), 0, h->st, a, b, sink, n, mode);
        EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
        EDS_HIP_TRY(hipEventSynchronize(h->ev1));
        float ms = 0.f;
        EDS_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        const float gbps = (float)((double)(mode == 1 ? 2 : 1) * (double)bytes * reps / ((double)ms * 1e-3) / 1e9);
        if (mode == 0) *read_GBps = gbps; else *copy_GBps = gbps;
    }
    EDS_HIP_TRY(hipGetLastError());
    return EDS_OK;
}

// One kernel of the streaming path timed COLD: before every repetition 1 GiB is READ through the caches (the Infinity Cache holds
// 256 MB: the planes the previous kernel wrote, or the frames an earlier repetition read, are gone — and what replaced them is clean),
// then the kernel runs between its own pair of events.  which: 0 the residual/Jacobian kernel, 1 the reduction kernel (over the planes of a residual/Jacobian pass made
// beforehand).  Reports the mean over the repetitions.
int eds_trk_bench_kernel_cold(eds_trk* h, int first, int count, int ncols, int which, int reps, float* mean_ms) {
    if (!h || !mean_ms) return fail(EDS_ERR_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > h->B || reps < 1) return fail(EDS_ERR_INVALID, "bad range");
    if (ncols != 6 && ncols != 12) return fail(EDS_ERR_INVALID, "ncols must be 6 or 12");
    if (which != 0 && which != 1) return fail(EDS_ERR_INVALID, "which: 0 residual/Jacobian, 1 reduction");
    if (ncols == 6 && h->cfg.nc) return fail(EDS_ERR_INVALID, "the NC residual (cfg.nc) has 12-column rows only");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    for (int s = first; s < first + count; ++s) {
        const Slot& sl = h->slots[s];
        if (!sl.has_kf || !sl.has_frame) return fail(EDS_ERR_STATE, "keyframe or event frame not set");
        fill_pose(h, s, sl.p, sl.q, sl.v);
    }
    int rc = run_pass(h, first, count, ncols, true, true, false);   // model + one warm pass (the planes the reduction reads)
    if (rc) return rc;
    const size_t evict_bytes = (size_t)1 << 30;
    if ((rc = probe_buffers(h, evict_bytes))) return rc;
    EDS_HIP_TRY(hipStreamSynchronize(h->st));
    const EdsArrays A = arrays_for_pass(h, first, count);
    const int N = max_points(h, first, count);
    const int nchunk = (N + EDS_TPB - 1) / EDS_TPB;
    const int nb_red = (ncols == 12) ? effective_blocks(h) : 1;
    int cpb, nseg;
    reduce_geometry(N, nb_red, ncols, &cpb, &nseg, h->knobs.reduce_ppl);
    const float4* a = reinterpret_cast<const float4*>(h->d_probe);
    float4* b = reinterpret_cast<float4*>(reinterpret_cast<char*>(h->d_probe) + evict_bytes);
    float* sink = reinterpret_cast<float*>(reinterpret_cast<char*>(h->d_probe) + 2 * evict_bytes);
    double total = 0.0;
    for (int i = 0; i < reps; ++i) {
        // (a READ-only pass: lines a copy had written would sit in the Infinity Cache dirty, and their write-back — forced by the very
        // reads under test — would be billed to the kernel: 70 instead of 45 us for the reduction's 229 MB)
        hipLaunchKernelGGL(eds_probe_kernel, dim3(8192), dim3(256), 0, h->st, a, b, sink, evict_bytes / 16, 0);
        EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
        if (which == 0) eds_launch_resjac(A, h->cfg.sampling, ncols, first, count, nchunk, h->st);
        else eds_launch_reduce(A, ncols, first, count, nseg, nb_red, cpb, h->st, h->knobs.reduce_ppl);
        EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
        EDS_HIP_TRY(hipEventSynchronize(h->ev1));
        float ms = 0.f;
        EDS_HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        total += ms;
    }
    EDS_HIP_TRY(hipGetLastError());
    *mean_ms = (float)(total / reps);
    return EDS_OK;
}

int eds_trk_loss_param_batch(eds_trk* h, int first, int count, int method, double* tau) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (!tau) return fail(EDS_ERR_INVALID, "null output");
    if (method == EDS_LP_CONSTANT) return EDS_OK;
    if (method != EDS_LP_MAD && method != EDS_LP_STD) return fail(EDS_ERR_INVALID, "unknown loss-param method");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    bool on_device = eds_points_supported(h, first, count);
    for (int s = first; s < first + count && on_device; ++s) on_device = h->slots[s].res_on_device;
    if (on_device) return eds_points_loss_param(h, first, count, method, tau);
    for (int s = first; s < first + count; ++s)          // residuals already on the host (or too many points): host selection
        if ((rc = eds_trk_loss_param(h, s, method, &tau[s - first]))) return rc;
    return EDS_OK;
}

}  // extern "C"
