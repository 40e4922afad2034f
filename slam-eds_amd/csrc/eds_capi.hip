// C ABI of libeds_hip.so (include/eds_hip.h): handle management, host<->HBM staging and the
// host-driven solve loop (EDS_EXEC_HOST).  The persistent on-device loop lives in eds_fused.hip.
//
// Replaces, for the hot path only, reference src/tracking/Tracker.cpp:40-102 (state handling),
// :104-241 (optimize) and :281-317 (getLossParams).  There is deliberately NO CPU fallback:
// without a HIP device every entry point fails with EDS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/eds_hip.h"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_kernels.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

#define EDS_HIP_TRY(expr)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(EDS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

}  // namespace

namespace {

int effective_blocks(const eds_trk* h) {
    int nb = h->cfg.num_blocks;
    if (nb < 1) nb = 1;
    if (nb > EDS_MAX_BLOCKS) nb = EDS_MAX_BLOCKS;
    return nb;
}
int level_iters(const eds_trk* h, int level) {
    if (level < 0) level = 0;
    if (level >= EDS_MAX_LEVELS) level = EDS_MAX_LEVELS - 1;
    return h->cfg.max_num_iterations[level];
}

int check_slot(const eds_trk* h, int slot) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (slot < 0 || slot >= h->B) return fail(EDS_ERR_INVALID, "slot out of range");
    return EDS_OK;
}

// constants of a slot's pose block that do not depend on (p,q,v)
void fill_static(const eds_trk* h, int slot) {
    const Slot& s = h->slots[slot];
    double* pb = h->h_pose + (size_t)slot * EDS_POSE_STRIDE;
    const int nb = effective_blocks(h);
    for (int i = 0; i < 4; ++i) pb[EDS_PB_K + i] = s.K[i];
    pb[EDS_PB_HUBER] = h->cfg.huber_tau > 0 ? h->cfg.huber_tau : 0.0;
    pb[EDS_PB_NB] = nb;
    pb[EDS_PB_NE] = s.N / nb;
    pb[EDS_PB_N] = s.N;
    pb[EDS_PB_NCMODE] = h->cfg.nc ? 1.0 : 0.0;
    pb[EDS_PB_FRAME] = (double)(s.frame_slot >= 0 ? s.frame_slot : slot);
}

void fill_pose(eds_trk* h, int slot, const double* p, const double* q, const double* v) {
    if (h->gram_pending) { hipStreamSynchronize(h->st); h->gram_pending = false; }      // h_G as an earlier refresh left it
    if (h->slots[slot].gram_host_stale) {           // set_idepth refreshed the Gram matrices in HBM only: fetch this slot's now
        const size_t off = (size_t)slot * EDS_MAX_BLOCKS * 36;
        hipMemcpyAsync(h->h_G + off, h->dG + off, (size_t)EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st);
        hipStreamSynchronize(h->st);
        h->slots[slot].gram_host_stale = false;
    }
    fill_static(h, slot);
    edsm::fill_pose_block(p, q, v, h->h_G + (size_t)slot * EDS_MAX_BLOCKS * 36, effective_blocks(h),
                          h->h_pose + (size_t)slot * EDS_POSE_STRIDE);
}

int upload_pose(eds_trk* h, int first, int count) {
    EDS_HIP_TRY(hipMemcpyAsync(h->dpose + (size_t)first * EDS_POSE_STRIDE, h->h_pose + (size_t)first * EDS_POSE_STRIDE,
                               sizeof(double) * EDS_POSE_STRIDE * count, hipMemcpyHostToDevice, h->st));
    return EDS_OK;
}

int max_points(const eds_trk* h, int first, int count) {
    int m = 0;
    for (int s = first; s < first + count; ++s) m = std::max(m, h->slots[s].N);
    return m;
}

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
    if (s.res_in_hostmap) {             // the kernel left a copy in pinned host memory: no HIP call at all
        const float* r = h->h_rmap + (size_t)slot * h->Np;
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

void free_all(eds_trk* h) {
    if (!h) return;
    hipSetDevice(h->dev);
    void* dptrs[] = {h->dkf, h->dpose, h->dG, h->dpart, h->dncstat,
                     h->dmhat, h->dframe, h->dr, h->dJ};
    for (void* p : dptrs) if (p) hipFree(p);
    eds_fused_free(&h->fused);
    eds_strips_free(h);
    eds_frame_free(&h->frame_build);
    eds_points_free(&h->point_ops);
    eds_keyframe_free(&h->kf_build);
    void* hptrs[] = {h->h_pose, h->h_part, h->h_G, h->h_f32, h->h_r, h->h_fstage, h->h_rmap, h->h_idp, h->h_fprog};
    for (void* p : hptrs) if (p) hipHostFree(p);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_stage) hipEventDestroy(h->ev_stage);
    if (h->ev_idp) hipEventDestroy(h->ev_idp);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

}  // namespace

// Row-major H x W host frame (double or float) -> the slot's frame in HBM.  The host only narrows to fp32 (a loop the compiler
// vectorises; no index arithmetic) into device-mapped pinned staging, in EDS_UPLOAD_BANDS bands of rows; behind every band a
// launch of k_store_rowmajor reads it over PCIe and writes tiles, padding and the replicated margin (= Grid2D's clamp), while
// the host narrows the next band.  Nothing is waited for: whatever uses the frame is ordered behind the launches on the handle's
// stream, and the staging buffer is private to this function (its event is waited for before the next frame overwrites it).
// (Round 1 built the tiled, margin-padded image element by element on one host thread: 300 us for 640x480, more than the solve;
// chunked hipMemcpyAsync into HBM + one tiling launch: 80 us, 30 of them after the host had finished.)
// slots that are about to receive a frame of their own stop sampling somebody else's
static int unshare_frames(eds_trk* h, int first, int count) {
    for (int s = first; s < first + count; ++s) {
        if (h->slots[s].frame_slot < 0) continue;
        h->slots[s].frame_slot = -1;
        fill_static(h, s);
        int rc = upload_pose(h, s, 1);
        if (rc) return rc;
    }
    return EDS_OK;
}
#define EDS_UPLOAD_BANDS 4
// fp64 -> fp32 narrowing of a band of the frame (set_event_frame's host work: 2.46 MB in, 1.23 MB out for 640x480).  The library is
// built without -march, so the plain loop is SSE2 — cvtpd2ps, two doubles per instruction, 36 us per VGA frame; the hosts of the pool
// (Zen 4 / 5) have AVX-512, older ones AVX2: pick at run time (function multiversioning by hand, __builtin_cpu_supports).
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx512f"))) static void narrow_avx512(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        _mm256_storeu_ps(dst + i, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i)));
        _mm256_storeu_ps(dst + i + 8, _mm512_cvtpd_ps(_mm512_loadu_pd(src + i + 8)));
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
}
__attribute__((target("avx2"))) static void narrow_avx2(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        _mm_storeu_ps(dst + i, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i)));
        _mm_storeu_ps(dst + i + 4, _mm256_cvtpd_ps(_mm256_loadu_pd(src + i + 4)));
    }
    for (; i < n; ++i) dst[i] = (float)src[i];
}
#endif
static void narrow_band(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
#if defined(__x86_64__)
    static const int level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    if (level == 2) return narrow_avx512(src, dst, n);
    if (level == 1) return narrow_avx2(src, dst, n);
#endif
    for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];       // (round-to-nearest-even in every variant: bit-identical results)
}
static void narrow_band(const float* __restrict__ src, float* __restrict__ dst, size_t n) { std::memcpy(dst, src, n * sizeof(float)); }

template <class T>
static int upload_frame(eds_trk* h, int slot, const T* frame) {
    { int rc_ = unshare_frames(h, slot, 1); if (rc_) return rc_; }     // a frame of its own again
    float* stage = h->h_fstage;
    if (h->stage_busy) { EDS_HIP_TRY(hipEventSynchronize(h->ev_stage)); h->stage_busy = false; }   // the previous frame's reads (long done)
    const bool banded = h->knobs.upload_bands != 0;     // A/B knob: one launch per band (round 2)
    if (h->d_fprog && !banded && h->H < (1 << 20)) {
        // ONE launch (round 3): k_store_follow's workgroups wait for the rows they move; the host publishes its progress after every
        // band in a pinned word (release store behind the band's plain stores: x86 keeps them in order for the device's reads).
        // 16 bands: what is left after the host's last store is 1/16 of a frame over PCIe.  (4 launches cost the host 16 of its 44 us.)
        if (h->h_fprog[1] & 0x80000000u) { h->h_fprog[1] = 0; return fail(EDS_ERR_HIP, "the previous frame upload timed out waiting for the host"); }
        // bands of 32 k rows (a band boundary is then a multiple of 128 bytes into the staging buffer whatever W is), at most 16 of them
        const int rows_per = 32 * std::max(1, (h->H + 32 * 16 - 1) / (32 * 16)), nbands = (h->H + rows_per - 1) / rows_per;
        const unsigned seq = (++h->upload_seq) & 0xfffu;
        __atomic_store_n(&h->h_fprog[0], seq << 20, __ATOMIC_RELEASE);
        const auto t_host = std::chrono::steady_clock::now();
        eds_frame_store_follow(h, slot, seq, rows_per);
        for (int k = 0; k < nbands; ++k) {
            const int rb = rows_per * k, re = std::min(h->H, rows_per * (k + 1));
            const size_t b = (size_t)rb * h->W, e = (size_t)re * h->W;
            narrow_band(frame + b, stage + b, e - b);
            __atomic_store_n(&h->h_fprog[0], (seq << 20) | (unsigned)re, __ATOMIC_RELEASE);
        }
        // A workgroup of the follower gives up after 2 s without progress (a host thread that was descheduled mid-frame).  Only then can
        // the slot be half-written, and only if this loop took that long: in that case wait, and store the frame again from the (now
        // complete) staging buffer with plain launches — a slot is never left holding a partial frame, has_frame stays truthful, and a
        // pyramid built on it (k_pyr_down) sees whole levels.
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host).count() > 1.0) {
            EDS_HIP_TRY(hipStreamSynchronize(h->st));
            if (h->h_fprog[1] & 0x80000000u) {
                h->h_fprog[1] = 0;
                for (int k = 0; k < nbands; ++k) eds_frame_store_rowmajor(h, slot, h->d_fstage, rows_per * k, std::min(h->H, rows_per * (k + 1)));
            }
        }
    } else
    for (int k = 0; k < EDS_UPLOAD_BANDS; ++k) {
        const int rb = h->H * k / EDS_UPLOAD_BANDS, re = h->H * (k + 1) / EDS_UPLOAD_BANDS;
        const size_t b = (size_t)rb * h->W, e = (size_t)re * h->W;
        narrow_band(frame + b, stage + b, e - b);
        eds_frame_store_rowmajor(h, slot, h->d_fstage, rb, re);
    }
    EDS_HIP_TRY(hipGetLastError());
    EDS_HIP_TRY(hipEventRecord(h->ev_stage, h->st));
    h->stage_busy = true;
    h->slots[slot].has_frame = true;
    ++h->slots[slot].frame_version;             // its strip copy (eds_strips.hip) is out of date
    return EDS_OK;
}

int eds_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }
int eds_internal_solve_host(eds_trk* h, int level, int first, int count) { return solve_host(h, level, first, count); }

extern "C" {

int eds_abi_version(void) { return EDS_HIP_ABI_VERSION; }

int eds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* eds_last_error(void) { return g_last_error.c_str(); }

int eds_trk_cfg_size(void) { return (int)sizeof(eds_trk_cfg); }
int eds_trk_info_size(void) { return (int)sizeof(eds_trk_info); }

void eds_trk_cfg_default(eds_trk_cfg* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->sampling = EDS_SAMPLE_BICUBIC;
    cfg->solver = EDS_SOLVER_LM6;
    cfg->exec = EDS_EXEC_DEVICE;
    cfg->num_blocks = 1;
    cfg->loss_type = EDS_LOSS_NONE;
    cfg->loss_param = 1.0;
    cfg->huber_tau = 0.0;
    cfg->lambda0 = 0.01;
    cfg->num_levels = 1;
    for (int i = 0; i < EDS_MAX_LEVELS; ++i) cfg->max_num_iterations[i] = 10;
    cfg->function_tolerance = 1e-6;
    cfg->gradient_tolerance = 1e-8;
    cfg->parameter_tolerance = 1e-6;
}

int eds_trk_create(const eds_trk_cfg* cfg, int batch, int max_points_, int H, int W, eds_trk** out) {
    if (!cfg || !out) return fail(EDS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (batch < 1 || max_points_ < 1 || H < 4 || W < 4) return fail(EDS_ERR_INVALID, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(EDS_ERR_NO_DEVICE, "no HIP device visible; libeds_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(EDS_ERR_INVALID, "device ordinal out of range");
    {   // the code object holds gfx950 kernels only (include/eds_hip.h: EDS_ERR_NO_DEVICE = "no gfx950 device visible")
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(EDS_ERR_HIP, "hipGetDeviceProperties failed");
        if (!std::strstr(prop.gcnArchName, "gfx950"))
            return fail(EDS_ERR_NO_DEVICE, std::string("device ") + std::to_string(cfg->device) + " is " + prop.gcnArchName +
                                               ", not gfx950 (MI355X); libeds_hip has no other code path and no CPU fallback");
    }
    eds_trk* h = new (std::nothrow) eds_trk();
    if (!h) return fail(EDS_ERR_INVALID, "out of memory");
    h->cfg = *cfg;
    h->B = batch; h->Nmax = max_points_; h->H = H; h->W = W; h->dev = cfg->device;
    h->Hp = eds_frame_extent(H); h->Wp = eds_frame_extent(W);
    h->tiled = 1;
    eds_knobs_from_env(&h->knobs);       // the ONLY place the library reads tuning variables from the environment (eds_launch_rule.hpp)
    h->tiled = h->knobs.frame_rowmajor ? 0 : 1;
    h->Np = ((max_points_ + EDS_POINT_ALIGN - 1) / EDS_POINT_ALIGN) * EDS_POINT_ALIGN;
    h->max_seg = h->Np / EDS_TPB + 2 * EDS_MAX_BLOCKS + 2;
    h->slots.resize(batch);
    for (Slot& s : h->slots) {                       // Tracker ctor, Tracker.cpp:43-46
        const double c = 0.001 / std::sqrt(6.0 * 0.001 * 0.001);
        for (int i = 0; i < 6; ++i) s.v[i] = c;
        std::memset(&s.info, 0, sizeof(s.info));
    }
    const size_t BN = (size_t)batch * h->Np;
#define EDS_ALLOC(ptr, bytes)                                                                  \
    do {                                                                                       \
        hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                    \
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e_)); } \
    } while (0)
#define EDS_HALLOC(ptr, bytes)                                                                 \
    do {                                                                                       \
        hipError_t e_ = hipHostMalloc((void**)&(ptr), (bytes), hipHostMallocDefault);          \
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e_)); } \
    } while (0)
    hipError_t e = hipSetDevice(h->dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_idp, hipEventDisableTiming);
    if (e != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("stream/event: ") + hipGetErrorString(e)); }
    EDS_ALLOC(h->dkf, BN * 4 * EDS_KF_PLANES);
    h->dx = h->dkf + EDS_KF_X * BN; h->dy = h->dkf + EDS_KF_Y * BN; h->drho = h->dkf + EDS_KF_RHO * BN;
    h->dgx = h->dkf + EDS_KF_GX * BN; h->dgy = h->dkf + EDS_KF_GY * BN; h->dw = h->dkf + EDS_KF_W * BN;
    h->df0x = h->dkf + EDS_KF_F0X * BN; h->df0y = h->dkf + EDS_KF_F0Y * BN;
    h->dcell0 = reinterpret_cast<int*>(h->dkf + EDS_KF_CELL0 * BN);
    EDS_ALLOC(h->dmhat, BN * 4); EDS_ALLOC(h->dr, BN * 4); EDS_ALLOC(h->dJ, BN * 4 * 12);
    EDS_ALLOC(h->dframe, (size_t)batch * h->Hp * h->Wp * 4);
    EDS_ALLOC(h->dpose, (size_t)batch * EDS_POSE_STRIDE * 8);
    EDS_ALLOC(h->dG, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    EDS_ALLOC(h->dpart, (size_t)batch * h->max_seg * EDS_RED_K * 8);
    EDS_ALLOC(h->dncstat, (size_t)batch * EDS_MAX_BLOCKS * 8 * 8);
    EDS_HALLOC(h->h_pose, (size_t)batch * EDS_POSE_STRIDE * 8);
    EDS_HALLOC(h->h_part, (size_t)batch * h->max_seg * EDS_RED_K * 8);
    EDS_HALLOC(h->h_G, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    h->h_f32_elems = std::max((size_t)h->Hp * h->Wp, (size_t)h->Np * 12);
    EDS_HALLOC(h->h_f32, h->h_f32_elems * 4);
    EDS_HALLOC(h->h_r, BN * 4);
    EDS_HALLOC(h->h_idp, (size_t)h->Np * 4);
    if (hipHostGetDevicePointer((void**)&h->d_idp, h->h_idp, 0) != hipSuccess) { (void)hipGetLastError(); h->d_idp = nullptr; }
    {   // mirror of the residual plane for the first few slots (eds_mirror_residuals)
        const size_t nr = (size_t)std::min(batch, EDS_RHOST_SLOTS) * h->Np;
        hipError_t e_ = hipHostMalloc((void**)&h->h_rmap, nr * 4, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_rmap, h->h_rmap, 0);
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc (residual mirror): ") + hipGetErrorString(e_)); }
        std::memset(h->h_rmap, 0, nr * 4);
    }
    {   // device-mapped: the tiling kernel reads the staging buffer in place
        hipError_t e_ = hipHostMalloc((void**)&h->h_fstage, (size_t)H * W * 4, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_fstage, h->h_fstage, 0);
        if (e_ != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("hipHostMalloc (frame staging): ") + hipGetErrorString(e_)); }
    }
    {   // progress / time-out words of the follower upload (eds_frame_store_follow); without them set_event_frame launches per band
        hipError_t e_ = hipHostMalloc((void**)&h->h_fprog, 64, hipHostMallocMapped);
        if (e_ == hipSuccess) e_ = hipHostGetDevicePointer((void**)&h->d_fprog, h->h_fprog, 0);
        if (e_ != hipSuccess) { (void)hipGetLastError(); if (h->h_fprog) hipHostFree(h->h_fprog); h->h_fprog = nullptr; h->d_fprog = nullptr; }
        else std::memset(h->h_fprog, 0, 64);
    }
    std::memset(h->h_pose, 0, (size_t)batch * EDS_POSE_STRIDE * 8);
    std::memset(h->h_G, 0, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8);
    hipMemsetAsync(h->dcell0, 0, BN * 4, h->st);
    float* f32s[] = {h->dx, h->dy, h->drho, h->dgx, h->dgy, h->dw, h->dmhat, h->dr, h->df0x, h->df0y};
    for (float* p : f32s) hipMemsetAsync(p, 0, BN * 4, h->st);
    hipMemsetAsync(h->dJ, 0, BN * 4 * 12, h->st);
    hipMemsetAsync(h->dframe, 0, (size_t)batch * h->Hp * h->Wp * 4, h->st);
    hipMemsetAsync(h->dpose, 0, (size_t)batch * EDS_POSE_STRIDE * 8, h->st);
    hipMemsetAsync(h->dG, 0, (size_t)batch * EDS_MAX_BLOCKS * 36 * 8, h->st);
    hipMemsetAsync(h->dpart, 0, (size_t)batch * h->max_seg * EDS_RED_K * 8, h->st);
    hipMemsetAsync(h->dncstat, 0, (size_t)batch * EDS_MAX_BLOCKS * 8 * 8, h->st);
    int rc = eds_fused_alloc(&h->fused, batch);
    if (rc != 0) { free_all(h); return fail(EDS_ERR_HIP, "fused buffers: hipMalloc failed"); }
    e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) { free_all(h); return fail(EDS_ERR_HIP, std::string("init: ") + hipGetErrorString(e)); }
    *out = h;
    return EDS_OK;
#undef EDS_ALLOC
#undef EDS_HALLOC
}

void eds_trk_destroy(eds_trk* h) { free_all(h); }

int eds_trk_set_config(eds_trk* h, const eds_trk_cfg* cfg) {
    if (!h || !cfg) return fail(EDS_ERR_INVALID, "null argument");
    if (cfg->device != h->cfg.device) return fail(EDS_ERR_INVALID, "device cannot change after create");
    const bool blocks_changed = cfg->num_blocks != h->cfg.num_blocks;
    h->cfg = *cfg;
    if (blocks_changed) {               // Gram matrices are per residual block
        for (int s = 0; s < h->B; ++s) {
            if (!h->slots[s].has_kf) continue;
            fill_static(h, s);
            int rc = upload_pose(h, s, 1);
            if (rc) return rc;
            eds_launch_gram(h->arrays(), s, effective_blocks(h), h->st);
        }
        EDS_HIP_TRY(hipMemcpyAsync(h->h_G, h->dG, (size_t)h->B * EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st));
        EDS_HIP_TRY(hipStreamSynchronize(h->st));
    }
    return EDS_OK;
}

int eds_trk_get_config(const eds_trk* h, eds_trk_cfg* cfg) {
    if (!h || !cfg) return fail(EDS_ERR_INVALID, "null argument");
    *cfg = h->cfg;
    return EDS_OK;
}

static int upload_points(eds_trk* h, int slot, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                         const double* w) {
    const size_t off = (size_t)slot * h->Np;
    const Slot& s = h->slots[slot];
    float* f = h->h_f32;
    const int Np = h->Np;
    int* cell = reinterpret_cast<int*>(f + (size_t)8 * Np);
    for (int i = 0; i < Np; ++i) {
        const bool in = i < N;
        f[0 * Np + i] = in ? (float)norm_xy[2 * i] : 0.f;
        f[1 * Np + i] = in ? (float)norm_xy[2 * i + 1] : 0.f;
        f[2 * Np + i] = in ? (float)idp[i] : 1.f;
        if (grad_xy) { f[3 * Np + i] = in ? (float)grad_xy[2 * i] : 0.f; f[4 * Np + i] = in ? (float)grad_xy[2 * i + 1] : 0.f; }
        if (w) f[5 * Np + i] = in ? (float)w[i] : 0.f;
        // the point's own keyframe pixel u0 = fx x + cx, v0 = fy y + cy in fp64, split into an integer
        // cell and an fp32 fraction: the kernels only ever add a small displacement to it (eds_device.hpp)
        double u0 = in ? s.K[0] * norm_xy[2 * i] + s.K[2] : 0.0, v0 = in ? s.K[1] * norm_xy[2 * i + 1] + s.K[3] : 0.0;
        double cu = std::floor(u0), cv = std::floor(v0);
        if (!(cu > -32000.0)) cu = -32000.0; if (cu > 32000.0) cu = 32000.0;     // far-off points keep the excess in the fraction
        if (!(cv > -32000.0)) cv = -32000.0; if (cv > 32000.0) cv = 32000.0;
        f[6 * Np + i] = (float)(u0 - cu);
        f[7 * Np + i] = (float)(v0 - cv);
        cell[i] = (int)(((unsigned)(int)cv << 16) | ((unsigned)(int)cu & 0xffffu));
    }
    float* dst[8] = {h->dx, h->dy, h->drho, h->dgx, h->dgy, h->dw, h->df0x, h->df0y};
    for (int k = 0; k < 8; ++k) {
        if ((k == 3 || k == 4) && !grad_xy) continue;
        if (k == 5 && !w) continue;
        EDS_HIP_TRY(hipMemcpyAsync(dst[k] + off, f + (size_t)k * Np, (size_t)Np * 4, hipMemcpyHostToDevice, h->st));
    }
    EDS_HIP_TRY(hipMemcpyAsync(h->dcell0 + off, cell, (size_t)Np * 4, hipMemcpyHostToDevice, h->st));
    return EDS_OK;
}

static int refresh_gram(eds_trk* h, int slot, bool wait = true) {
    fill_static(h, slot);
    int rc = upload_pose(h, slot, 1);
    if (rc) return rc;
    eds_launch_gram(h->arrays(), slot, effective_blocks(h), h->st);
    EDS_HIP_TRY(hipGetLastError());
    const size_t off = (size_t)slot * EDS_MAX_BLOCKS * 36;
    EDS_HIP_TRY(hipMemcpyAsync(h->h_G + off, h->dG + off, (size_t)EDS_MAX_BLOCKS * 36 * 8, hipMemcpyDeviceToHost, h->st));
    h->slots[slot].gram_host_stale = false;
    if (wait) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); h->gram_pending = false; }
    else h->gram_pending = true;        // the device solvers read dG on the stream; host readers of h_G wait in fill_pose
    return EDS_OK;
}

int eds_trk_set_keyframe(eds_trk* h, int slot, int N, const double* norm_xy, const double* grad_xy, const double* idp,
                         const double* w, double fx, double fy, double cx, double cy) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (N < 1 || N > h->Nmax) return fail(EDS_ERR_INVALID, "N out of range for this handle");
    if (!norm_xy || !grad_xy || !idp || !w) return fail(EDS_ERR_INVALID, "null keyframe array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    Slot& s = h->slots[slot];
    s.N = N; s.K[0] = fx; s.K[1] = fy; s.K[2] = cx; s.K[3] = cy;
    if ((rc = upload_points(h, slot, N, norm_xy, grad_xy, idp, w))) return rc;
    if ((rc = refresh_gram(h, slot))) return rc;
    s.has_kf = true;
    // residuals and trace of an earlier solve belong to the previous keyframe: drop the host copy AND the "still in HBM" marks,
    // so that get_residuals / loss_param before the next optimize report EDS_ERR_STATE instead of another keyframe's plane
    s.residuals.clear();
    s.res_on_device = false; s.trace_on_device = false; s.ntrace = 0;
    return EDS_OK;
}

int eds_trk_set_idepth(eds_trk* h, int slot, int N, const double* idp) { return eds_trk_set_idepth_strided(h, slot, N, idp, 1); }

int eds_trk_set_idepth_strided(eds_trk* h, int slot, int N, const double* idp, int stride) {
    if (stride < 1) return fail(EDS_ERR_INVALID, "stride must be at least 1");
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (!s.has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
    if (N != s.N || !idp) return fail(EDS_ERR_INVALID, "idp size mismatch");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    // only the inverse-depth plane changes (the geometry uses rho' = idp + 1e-5, the model the raw idp)
    // (Tracker.cpp:167 re-reads the depths on every optimize: this is on the live path, so nothing here waits for the GPU)
    if (h->idp_busy) { EDS_HIP_TRY(hipEventSynchronize(h->ev_idp)); h->idp_busy = false; }
    for (int i = 0; i < h->Np; ++i) h->h_idp[i] = i < N ? (float)idp[(size_t)i * stride] : 1.f;
    if (h->d_idp) {
        // ONE launch: the Gram kernel reads the new depths out of the mapped staging, stores them into the rho plane on its way and
        // leaves the Gram matrices in HBM, where the device solvers read them; the host copy is fetched only if a host-side solver or
        // eval asks for it (fill_pose).  (Round 2: copy + event + pose upload + launch + copy back = ~20 us of host time.)
        eds_launch_gram(h->arrays(), slot, effective_blocks(h), h->st, h->d_idp);
        EDS_HIP_TRY(hipGetLastError());
        EDS_HIP_TRY(hipEventRecord(h->ev_idp, h->st));
        h->idp_busy = true;
        s.gram_host_stale = true;
        return EDS_OK;
    }
    EDS_HIP_TRY(hipMemcpyAsync(h->drho + (size_t)slot * h->Np, h->h_idp, (size_t)h->Np * 4, hipMemcpyHostToDevice, h->st));
    EDS_HIP_TRY(hipEventRecord(h->ev_idp, h->st));
    h->idp_busy = true;
    return refresh_gram(h, slot, false);
}

int eds_trk_set_event_frame(eds_trk* h, int slot, const double* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frame(h, slot, frame);
}

int eds_trk_set_event_frame_f32(eds_trk* h, int slot, const float* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null frame");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return upload_frame(h, slot, frame);
}

int eds_trk_set_undistort_map(eds_trk* h, const float* mapx, const float* mapy) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if ((mapx == nullptr) != (mapy == nullptr)) return fail(EDS_ERR_INVALID, "mapx and mapy must both be given or both be NULL");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_frame_set_map(h, mapx, mapy, h->H, h->W);
}

int eds_trk_set_undistort_map_sized(eds_trk* h, const float* mapx, const float* mapy, int sensor_H, int sensor_W) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if ((mapx == nullptr) != (mapy == nullptr)) return fail(EDS_ERR_INVALID, "mapx and mapy must both be given or both be NULL");
    if (mapx && (sensor_H < 1 || sensor_W < 1)) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_frame_set_map(h, mapx, mapy, sensor_H, sensor_W);
}

int eds_trk_build_event_frame(eds_trk* h, int slot, int n_events, const uint16_t* x, const uint16_t* y, const uint8_t* polarity,
                              int level, double blur_sigma, int use_exp_weights, double* norm_out) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (n_events < 0 || level < 0 || level > 16) return fail(EDS_ERR_INVALID, "bad event count or level");
    if (n_events > 0 && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if ((rc = unshare_frames(h, slot, 1))) return rc;
    return eds_frame_build_levels(h, slot, level, 1, n_events, x, y, polarity, h->H, h->W, blur_sigma, use_exp_weights, norm_out);
}

int eds_trk_build_event_frames(eds_trk* h, int first_slot, int num_levels, int n_events, const uint16_t* x, const uint16_t* y,
                               const uint8_t* polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (num_levels < 1 || num_levels > EDS_MAX_LEVELS) return fail(EDS_ERR_INVALID, "num_levels out of range");
    if (first_slot < 0 || first_slot + num_levels > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds (one slot per level)");
    if (n_events < 0) return fail(EDS_ERR_INVALID, "bad event count");
    if (n_events > 0 && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    if (sensor_H <= 0 || sensor_W <= 0) { sensor_H = h->H; sensor_W = h->W; }
    if (sensor_H < 2 || sensor_W < 2) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, num_levels); if (rc_) return rc_; }
    return eds_frame_build_levels(h, first_slot, 0, num_levels, n_events, x, y, polarity, sensor_H, sensor_W, blur_sigma, use_exp_weights, norms);
}

int eds_trk_build_event_frames_aos(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                   int off_y, int off_polarity, int sensor_H, int sensor_W, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (num_levels < 1 || num_levels > EDS_MAX_LEVELS) return fail(EDS_ERR_INVALID, "num_levels out of range");
    if (first_slot < 0 || first_slot + num_levels > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds (one slot per level)");
    if (n_events < 0 || (n_events > 0 && !events)) return fail(EDS_ERR_INVALID, "bad event array");
    if (stride < 5 || off_x < 0 || off_y < 0 || off_polarity < 0 || off_x + 2 > stride || off_y + 2 > stride || off_polarity + 1 > stride ||
        (off_x & 1) || (off_y & 1) || (stride & 1))
        return fail(EDS_ERR_INVALID, "bad event layout (x, y: 2-byte aligned uint16 fields inside an even stride)");
    if (sensor_H <= 0 || sensor_W <= 0) { sensor_H = h->H; sensor_W = h->W; }
    if (sensor_H < 2 || sensor_W < 2) return fail(EDS_ERR_INVALID, "bad sensor size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, num_levels); if (rc_) return rc_; }
    const EdsEventAos aos = {events, stride, off_x, off_y, off_polarity};
    return eds_frame_build_levels(h, first_slot, 0, num_levels, n_events, nullptr, nullptr, nullptr, sensor_H, sensor_W, blur_sigma, use_exp_weights,
                                  norms, &aos);
}

// EventFrame::create's time bookkeeping (EventFrame.cpp:313-335), host only.  The reference object is STATEFUL: clear() (called at the head
// of create) does not touch first_time / last_time, and last_time is only assigned in the `else if ((it + 1) == events.end())` branch — so a
// slice of ONE event keeps the PREVIOUS slice's last_time, and both the order check and delta_time use that.  `out->last_time` is therefore
// in/out: on entry the previous slice's last_time (0 on a fresh object), on return this slice's — unchanged for a single event
// (last_valid = 0) and for an empty slice.
int eds_event_times_aos(int n_events, const void* events, int stride, int off_ts, eds_event_times* out) {
    if (!out) return fail(EDS_ERR_INVALID, "null output");
    const int64_t prev_last = out->last_time;
    std::memset(out, 0, sizeof(*out));
    out->last_time = prev_last;
    if (n_events < 0 || (n_events > 0 && !events)) return fail(EDS_ERR_INVALID, "bad event array");
    if (stride < 8 || off_ts < 0 || off_ts + 8 > stride) return fail(EDS_ERR_INVALID, "bad event layout (ts: int64 field inside the stride)");
    if (n_events == 0) { out->delta_time = out->last_time; return EDS_OK; }   // (the loop does not run: first_time stays as well — reported as 0 here, the caller holds the state)
    auto ts = [&](int i) { int64_t t; std::memcpy(&t, static_cast<const char*>(events) + (size_t)i * stride + off_ts, 8); return t; };
    out->first_time = ts(0);
    if (n_events > 1) { out->last_time = ts(n_events - 1); out->last_valid = 1; }     // `else if ((it + 1) == events.end())`: never for a single event
    if (out->first_time > out->last_time)
        return fail(EDS_ERR_INVALID, "[EVENT_FRAME] Event time[0] > event time [N-1] (EventFrame.cpp:325-329)");
    out->time = ts(n_events / 2);
    out->delta_time = out->last_time - out->first_time;
    return EDS_OK;
}

int eds_trk_build_event_frames_aos_timed(eds_trk* h, int first_slot, int num_levels, int n_events, const void* events, int stride, int off_x,
                                         int off_y, int off_polarity, int off_ts, int sensor_H, int sensor_W, double blur_sigma,
                                         int use_exp_weights, double* norms, eds_event_times* times) {
    eds_event_times local;
    std::memset(&local, 0, sizeof(local));           // (last_time is in/out: a caller without a struct of its own has no history)
    int rc = eds_event_times_aos(n_events, events, stride, off_ts, times ? times : &local);
    if (rc) return rc;
    return eds_trk_build_event_frames_aos(h, first_slot, num_levels, n_events, events, stride, off_x, off_y, off_polarity, sensor_H, sensor_W,
                                          blur_sigma, use_exp_weights, norms);
}

int eds_trk_build_event_frame_batch(eds_trk* h, int first_slot, int count, const int* offsets, const uint16_t* x, const uint16_t* y,
                                    const uint8_t* polarity, int level, double blur_sigma, int use_exp_weights, double* norms) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (count < 1 || first_slot < 0 || first_slot + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    if (!offsets || offsets[0] < 0) return fail(EDS_ERR_INVALID, "bad offsets");
    if (level < 0 || level > 16) return fail(EDS_ERR_INVALID, "bad level");
    if (offsets[count] > offsets[0] && (!x || !y || !polarity)) return fail(EDS_ERR_INVALID, "null event array");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    { int rc_ = unshare_frames(h, first_slot, count); if (rc_) return rc_; }
    return eds_frame_build_batch(h, first_slot, count, offsets, x, y, polarity, level, blur_sigma, use_exp_weights, norms);
}

int eds_trk_share_event_frame(eds_trk* h, int slot, int src_slot) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if ((rc = check_slot(h, src_slot))) return rc;
    Slot& s = h->slots[slot];
    const Slot& src = h->slots[src_slot];
    if (src.frame_slot >= 0 && src_slot != slot) return fail(EDS_ERR_INVALID, "the source slot itself shares another slot's frame");
    if (!src.has_frame && src_slot != slot) return fail(EDS_ERR_STATE, "the source slot has no event frame yet");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    s.frame_slot = src_slot == slot ? -1 : src_slot;
    if (src_slot != slot) s.has_frame = true;
    fill_static(h, slot);
    return upload_pose(h, slot, 1);                             // ordered before the next solve on the handle's stream
}

int eds_trk_get_event_frame(eds_trk* h, int slot, double* frame) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!frame) return fail(EDS_ERR_INVALID, "null output");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    const size_t n = (size_t)h->Hp * h->Wp;
    EDS_HIP_TRY(hipStreamSynchronize(h->st));                    // set_event_frame does not wait for its own upload
    const int fs = h->slots[slot].frame_slot >= 0 ? h->slots[slot].frame_slot : slot;      // a sharing slot: the frame it samples
    EDS_HIP_TRY(hipMemcpy(h->h_f32, h->dframe + (size_t)fs * n, n * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < h->H; ++r)
        for (int c = 0; c < h->W; ++c) frame[(size_t)r * h->W + c] = h->h_f32[eds_frame_index(r, c, h->Wp, h->tiled)];
    return EDS_OK;
}

int eds_trk_set_state(eds_trk* h, int slot, const double p[3], const double q[4], const double v[6]) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (p) std::memcpy(s.p, p, sizeof(s.p));
    if (q) std::memcpy(s.q, q, sizeof(s.q));
    if (v) std::memcpy(s.v, v, sizeof(s.v));
    return EDS_OK;
}

int eds_trk_get_state(eds_trk* h, int slot, double p[3], double q[4], double v[6]) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    const Slot& s = h->slots[slot];
    if (p) std::memcpy(p, s.p, sizeof(s.p));
    if (q) std::memcpy(q, s.q, sizeof(s.q));
    if (v) std::memcpy(v, s.v, sizeof(s.v));
    return EDS_OK;
}

static int check_range(const eds_trk* h, int first, int count) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (first < 0 || count < 1 || first + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    return EDS_OK;
}

int eds_trk_set_states(eds_trk* h, int first, int count, const double* p, const double* q, const double* v) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        Slot& s = h->slots[first + i];
        if (p) std::memcpy(s.p, p + 3 * i, sizeof(s.p));
        if (q) std::memcpy(s.q, q + 4 * i, sizeof(s.q));
        if (v) std::memcpy(s.v, v + 6 * i, sizeof(s.v));
    }
    return EDS_OK;
}

int eds_trk_get_states(eds_trk* h, int first, int count, double* p, double* q, double* v) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        const Slot& s = h->slots[first + i];
        if (p) std::memcpy(p + 3 * i, s.p, sizeof(s.p));
        if (q) std::memcpy(q + 4 * i, s.q, sizeof(s.q));
        if (v) std::memcpy(v + 6 * i, s.v, sizeof(s.v));
    }
    return EDS_OK;
}

int eds_trk_get_results(eds_trk* h, int first, int count, double* t) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (!t) return fail(EDS_ERR_INVALID, "null output");
    for (int i = 0; i < count; ++i) {
        const Slot& s = h->slots[first + i];
        double* o = t + 16 * i;
        std::memcpy(o, s.p, 24); std::memcpy(o + 3, s.q, 32); std::memcpy(o + 7, s.v, 48);
        o[13] = s.info.final_cost; o[14] = s.info.num_iterations; o[15] = s.info.success ? 1.0 : 0.0;
    }
    return EDS_OK;
}

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

// Waits for the handle's stream.  A launch of a few alignments is over in 0.1-0.3 ms, and a blocking wait adds the wake-up of the
// calling thread to every such call; the latency regime therefore polls the stream (hipStreamQuery) for up to EDS_SPIN_US before
// it blocks.  Batches block right away: nobody should burn a core for milliseconds.
#define EDS_SPIN_US 500.0
static hipError_t wait_stream(eds_trk* h) {
    const bool spin = h->cfg.exec == EDS_EXEC_DEVICE && h->fused.pending_count > 0 && h->fused.pending_count <= 64 && !h->knobs.no_spin;
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(h->st);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > EDS_SPIN_US) break;
        }
    }
    return hipStreamSynchronize(h->st);
}

int eds_trk_sync(eds_trk* h) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(wait_stream(h));
    if (h->h_fprog && (h->h_fprog[1] & 0x80000000u)) {      // a frame upload gave up waiting for the host: that slot's frame is incomplete
        h->h_fprog[1] = 0;
        if (h->cfg.exec == EDS_EXEC_DEVICE) (void)eds_fused_collect(h);
        return fail(EDS_ERR_HIP, "a frame upload timed out waiting for the host (set the event frame again)");
    }
    if (h->cfg.exec == EDS_EXEC_DEVICE) return eds_fused_collect(h);
    return EDS_OK;
}

int eds_trk_get_info(eds_trk* h, int slot, eds_trk_info* info) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!info) return fail(EDS_ERR_INVALID, "null info");
    *info = h->slots[slot].info;
    return EDS_OK;
}

int eds_trk_get_trace(eds_trk* h, int slot, int max_iters, double* increments, double* costs, int32_t* accepted) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (h->slots[slot].trace_on_device && (rc = eds_fused_fetch_trace(h, slot))) return rc;
    const Slot& s = h->slots[slot];
    const int n = std::min(max_iters, s.ntrace);
    for (int i = 0; i < n; ++i) {
        if (increments) std::memcpy(increments + 6 * i, &s.tr_xi[6 * i], 6 * sizeof(double));
        if (costs) costs[i] = 0.5 * s.tr_cost[i];
        if (accepted) accepted[i] = s.tr_acc[i];
    }
    return n;
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

int eds_trk_timer_start(eds_trk* h) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
    return EDS_OK;
}

int eds_trk_prepare_frames(eds_trk* h, int first, int count, int force, float* elapsed_ms) {
    int rc = check_range(h, first, count);
    if (rc) return rc;
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (!h->tiled) return EDS_OK;      // (both samplers gather from the strips: the bilinear one reads the middle of the same patches)
    EDS_HIP_TRY(hipSetDevice(h->dev));
    if (force)
        for (int s = first; s < first + count; ++s) h->slots[h->slots[s].frame_slot >= 0 ? h->slots[s].frame_slot : s].strips_version = 0;
    if (elapsed_ms) EDS_HIP_TRY(hipEventRecord(h->ev0, h->st));
    if (!eds_strips_prepare(h, first, count)) return fail(EDS_ERR_HIP, "no memory for the strip copies of the frames");
    if (elapsed_ms) {
        EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
        EDS_HIP_TRY(hipEventSynchronize(h->ev1));
        EDS_HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    }
    return EDS_OK;
}

int eds_trk_set_knob(eds_trk* h, const char* name, const char* value) {
    if (!h || !name) return fail(EDS_ERR_INVALID, "null argument");
    if (std::strcmp(name, "EDS_FRAME_LAYOUT") == 0) return fail(EDS_ERR_STATE, "EDS_FRAME_LAYOUT decides the allocation: environment at eds_trk_create only");
    if (eds_knobs_set(&h->knobs, name, value) != 0) return fail(EDS_ERR_INVALID, std::string("unknown knob ") + name);
    if (std::strncmp(name, "EDS_STRIPS_", 11) == 0) h->strips_unavailable = false;       // a new budget / phase count: ask again
    return EDS_OK;
}

int eds_trk_get_strips_info(eds_trk* h, int64_t* bytes, int32_t* row_phases, int32_t* unavailable) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (bytes) *bytes = (int64_t)h->strips_bytes;
    if (row_phases) *row_phases = h->dstrips ? h->strip_phases : 0;
    if (unavailable) *unavailable = h->strips_unavailable ? 1 : 0;
    return EDS_OK;
}

int eds_trk_last_launch(eds_trk* h, eds_trk_launch_info* out) {
    if (!h || !out) return fail(EDS_ERR_INVALID, "null argument");
    return eds_fused_last_launch(h, out);
}

int eds_trk_timer_stop(eds_trk* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return fail(EDS_ERR_INVALID, "null argument");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    EDS_HIP_TRY(hipEventRecord(h->ev1, h->st));
    EDS_HIP_TRY(hipEventSynchronize(h->ev1));
    EDS_HIP_TRY(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
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

int eds_trk_update_points(eds_trk* h, int slot, int delete_out_points, double* coord_xy, double* tracks_xy, int32_t* kept_index,
                          int* n_kept, double* mean_sq_flow) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    Slot& s = h->slots[slot];
    if (!s.has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    int n = 0;
    if ((rc = eds_points_update(h, slot, delete_out_points != 0, coord_xy, tracks_xy, kept_index, &n, mean_sq_flow))) return rc;
    if (n_kept) *n_kept = n;
    if (n != s.N) {                      // points were erased: every index-aligned plane was compacted on the device
        s.N = n;
        s.residuals.clear();
        s.res_on_device = false;
        if (n > 0 && (rc = refresh_gram(h, slot))) return rc;
        if (n == 0) s.has_kf = false;
    }
    return EDS_OK;
}

int eds_trk_update_points_batch(eds_trk* h, int first, int count, int delete_out_points, int stride, double* coord_xy, double* tracks_xy,
                                int32_t* kept_index, int* n_kept, double* mean_sq_flow) {
    if (!h) return fail(EDS_ERR_INVALID, "null handle");
    if (count < 1 || first < 0 || first + count > h->B) return fail(EDS_ERR_INVALID, "slot range out of bounds");
    int maxN = 0;
    for (int s = first; s < first + count; ++s) {
        if (!h->slots[s].has_kf) return fail(EDS_ERR_STATE, "keyframe not set");
        maxN = std::max(maxN, h->slots[s].N);
    }
    if ((coord_xy || tracks_xy || kept_index) && stride < maxN) return fail(EDS_ERR_INVALID, "stride smaller than the largest point count");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    std::vector<int> n(count);
    int rc = eds_points_update_batch(h, first, count, delete_out_points != 0, stride, coord_xy, tracks_xy, kept_index, n.data(), mean_sq_flow);
    if (rc) return rc;
    bool any = false;
    for (int b = 0; b < count; ++b) {
        Slot& s = h->slots[first + b];
        if (n_kept) n_kept[b] = n[b];
        if (n[b] != s.N) {                  // points were erased: every index-aligned plane was compacted on the device
            s.N = n[b];
            s.residuals.clear();
            s.res_on_device = false;
            if (n[b] > 0) { if ((rc = refresh_gram(h, first + b, false))) return rc; any = true; }
            else s.has_kf = false;
        }
    }
    if (any) { EDS_HIP_TRY(hipStreamSynchronize(h->st)); h->gram_pending = false; }
    return EDS_OK;
}

/* ---- keyframe point set-up on the device ------------------------------------------ */
void eds_kf_select_default(eds_kf_select* sel) {
    if (!sel) return;
    std::memset(sel, 0, sizeof(*sel));
    sel->method = EDS_KF_MEDIAN;            // KeyFrame::create falls back to MEDIAN without a point target (KeyFrame.cpp:410-411)
    sel->cell = 20;                          // cv::Size(20, 20)  (KeyFrame.cpp:408)
    sel->num_points = 0;
    sel->sobel_ksize = 3;                    // KeyFrame::create (KeyFrame.cpp:384-385); 7: the constructor's (KeyFrame.cpp:239-240)
    sel->min_depth = 1.0; sel->max_depth = 3.0;
    sel->weight_threshold = 0.7;             // cleanPoints(0.7)  (KeyFrame.cpp:451)
}

int eds_trk_build_keyframe(eds_trk* h, int slot, int img_type, const void* img, const eds_kf_select* sel, int n_depth,
                           const double* depth_xy, const double* depth_idp, double fx, double fy, double cx, double cy, int* n_points) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!img || !sel) return fail(EDS_ERR_INVALID, "null image or selection parameters");
    if (n_depth < 0) return fail(EDS_ERR_INVALID, "negative depth-map size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_build(h, slot, img_type, img, h->H, h->W, 1, sel, n_depth, depth_xy, depth_idp, fx, fy, cx, cy, n_points);
}

int eds_trk_build_keyframe_image(eds_trk* h, int slot, int img_type, const void* img, int img_H, int img_W, int channels,
                                 const eds_kf_select* sel, int n_depth, const double* depth_xy, const double* depth_idp, double fx, double fy,
                                 double cx, double cy, int* n_points) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    if (!img || !sel) return fail(EDS_ERR_INVALID, "null image or selection parameters");
    if (n_depth < 0) return fail(EDS_ERR_INVALID, "negative depth-map size");
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_build(h, slot, img_type, img, img_H, img_W, channels, sel, n_depth, depth_xy, depth_idp, fx, fy, cx, cy, n_points);
}

int eds_trk_get_keyframe_points(eds_trk* h, int slot, double* coord_xy, double* norm_xy, double* grad_xy, double* idp, double* weights) {
    int rc = check_slot(h, slot);
    if (rc) return rc;
    EDS_HIP_TRY(hipSetDevice(h->dev));
    return eds_keyframe_get_points(h, slot, coord_xy, norm_xy, grad_xy, idp, weights);
}

}  // extern "C"

int eds_internal_refresh_gram(eds_trk* h, int slot) { return refresh_gram(h, slot); }
