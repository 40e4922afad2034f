// Persistent on-device solve of THE REFERENCE PROBLEM (EDS_SOLVER_REF12 with EDS_EXEC_DEVICE):
// 12 local parameters (translation, quaternion-local, unit velocity), per-block model normalisation and
// robust loss, Ceres trust-region LM semantics (reference Tracker.cpp:104-241, PhotometricError.hpp:124-182).
//
// Same organisation as eds_fused.hip — one workgroup per alignment, points resident in registers, frame
// patches in the LDS cache, lane 0 runs the shared state machine (edss::Solver12) — with two differences:
//   * every point contributes a 1x12 row (closed forms of SURVEY §8a), so the running sums are the 78 + 12 + 1
//     entries of the upper triangle of J^T J, J^T r and r^2: a 128-wide reduce-scatter butterfly;
//   * the reference normalises the model and applies its loss per residual block (options.num_threads blocks,
//     Tracker.cpp:178-195), so the sums are formed block by block (up to EDS_DEV_MAX_BLOCKS = 8 on the device;
//     more blocks fall back to the host-driven loop).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"
#include "eds_solver12_coop.hpp"

using namespace edsd;

#ifndef EDS12_THREADS
#define EDS12_THREADS 512
#endif
#define EDS12_WAVES (EDS12_THREADS / 64)
#ifndef EDS12_MIN_WAVES_PER_SIMD
#define EDS12_MIN_WAVES_PER_SIMD 2          // 8 wavefronts per CU: one 512-thread workgroup, or two of 256 threads
#endif
#define EDS12_MAX_POINTS 2048
#define EDS12_MAX_PPT (EDS12_MAX_POINTS / EDS12_THREADS)
#ifndef EDS12_MFMA_F64
#define EDS12_MFMA_F64 1
#endif
#ifndef EDS12_CACHE_CAP
#define EDS12_CACHE_CAP 1536
#endif                             // 96 KB of patches: the rest of the LDS holds the MFMA staging rows and the solver

template <int SAMPLING, int PPT>
__global__ __launch_bounds__(EDS12_THREADS, EDS12_MIN_WAVES_PER_SIMD) void eds_fused12_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                                  EdsFused12Out* __restrict__ out, int first, int iters,
                                                                  int loss_type, double loss_a, double ftol, double gtol,
                                                                  double ptol, int nb) {
    const int slot = first + blockIdx.x;
    const int tid = threadIdx.x, nthr = EDS12_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NTAP = (SAMPLING == 0) ? 16 : 4;
    constexpr int HALF = PPT > 1 ? PPT / 2 : 1;          // points whose gathers are in flight together, per lane
    __shared__ edss::Solver12 sv;
    __shared__ edss::Sums12Dev sums;
    __shared__ edsc::Work12 work;
    __shared__ double s_pose[EDS_POSE_STRIDE];
    __shared__ float s_stage[EDS12_WAVES][64 * 17];   // per wavefront: 64 rows [J | r] padded to 16 columns, stride 17 (bank-conflict-free)
    __shared__ int s_state;            // 0: iterate, 2: done
    __shared__ int s_accept;           // the evaluation just consumed became the accepted point
    __shared__ float s_patch[NTAP][EDS12_CACHE_CAP];
    __shared__ int s_cell[EDS12_CACHE_CAP];
    __shared__ double s_G[EDS_DEV_MAX_BLOCKS * 36];   // per-block Gram matrices (constant per keyframe): read by every pose-block refill

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const double* __restrict__ Gg = A.G + (size_t)slot * EDS_MAX_BLOCKS * 36;
    const double* G = s_G;
    const int N = (int)gpb[EDS_PB_N];
    const int ne = N / nb;
    const size_t base = (size_t)slot * A.Np;
    const FrameView frame = make_frame_view(A.frame, slot, A.H, A.W, A.Hp, A.Wp, A.tiled);

    if (wave == 0) {
        for (int k = lane; k < nb * 36; k += 64) s_G[k] = Gg[k];
        if (lane == 0) {
            const EdsFusedIn& I = in[slot];
            for (int i = 0; i < 4; ++i) s_pose[EDS_PB_K + i] = gpb[EDS_PB_K + i];
            sv.init(iters, loss_type, loss_a, ftol, gtol, ptol, I.p, I.q, I.v);
            sv.skip_final = 1;              // the residuals of the accepted point stay in registers (racc below)
            sums.nb = nb;
            s_state = 0; s_accept = 0;
#ifdef EDS_FUSED_STAMPS
            for (int k = 0; k < 8; ++k) work.st[k] = 0;
#endif
        }
        EDS_WSYNC();
        edsc::coop_fill_pose_block(sv.cp, sv.cq, sv.cv, G, nb, s_pose, lane);
    }
    // per-point constants in registers for the whole solve
    PointKf kf[PPT];
    float kw[PPT], kgx[PPT], kgy[PPT];  // the 6-entry model row a_i is rebuilt from (x, y, rho, gx, gy) when needed: 2 registers, not 6
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * nthr;
        const bool in_range = i < N;
        const size_t o = base + (in_range ? i : 0);
        kf[j].x = A.x[o]; kf[j].y = A.y[o]; kf[j].rhop = A.rho[o] + 1e-5f;
        kf[j].f0x = A.f0x[o]; kf[j].f0y = A.f0y[o]; kf[j].cell0 = A.cell0[o];
        kw[j] = in_range ? A.w[o] : 0.0f;
        kgx[j] = A.gx[o]; kgy[j] = A.gy[o];
        if (i < EDS12_CACHE_CAP) s_cell[i] = 0x7fffffff;
    }
    for (int k = tid; k < EDS12_WAVES * 64 * 17; k += nthr) (&s_stage[0][0])[k] = 0.0f;     // columns 13..15 stay zero for good
    for (int k = tid; k < (int)(sizeof(sums) / sizeof(double)); k += nthr)
        if (k > 0) reinterpret_cast<double*>(&sums)[k] = 0.0;                               // word 0 holds nb
    float rcand[PPT], racc[PPT];        // residuals of the evaluation in flight / of the accepted point
#pragma unroll
    for (int j = 0; j < PPT; ++j) { rcand[j] = 0.0f; racc[j] = 0.0f; }
    __syncthreads();

#ifdef EDS_FUSED_STAMPS
    unsigned long long st_acc[3] = {0, 0, 0}, st_t = __builtin_readcyclecounter();
    int st_n = 0;
#define EDS12_STAMP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[k] += n_ - st_t; st_t = n_; } while (0)
#else
#define EDS12_STAMP(k) do { } while (0)
#endif
#if EDS12_MFMA_F64
    typedef double acc4 __attribute__((ext_vector_type(4)));     // v_mfma_f64_16x16x4_f64: products and sums in fp64
#else
    typedef float acc4 __attribute__((ext_vector_type(4)));      // v_mfma_f32_16x16x4_f32: exact fp32 fmaf chain
#endif
    float* const stage = &s_stage[wave][0];
    // adds this wavefront's 16x16 tile (rows/cols 0..11: J^T J, column 12: J^T r, [12][12]: sum r^2) to block b's sums
    auto flush = [&](const acc4& C, int b) {
        const int col = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#if EDS12_MFMA_F64
            const int row = (lane >> 4) + 4 * r;        // C/D layout of the f64 form
#else
            const int row = (lane >> 4) * 4 + r;
#endif
            const double v = (double)C[r];
            if (row < 12 && col < 12) unsafeAtomicAdd(&sums.H[b][12 * row + col], v);
            else if (row < 12 && col == 12) unsafeAtomicAdd(&sums.g[b][row], v);
            else if (row == 12 && col == 12) unsafeAtomicAdd(&sums.s[b], v);
        }
    };
    for (;;) {
        PoseF ps;
        load_pose(s_pose, ps);
        float vf[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) vf[k] = uniformf((float)s_pose[EDS_PB_V + k]);
        acc4 C = {0, 0, 0, 0}, C2 = {0, 0, 0, 0};
        int cb = -1;                    // residual block the tile C currently belongs to (wave-uniform)
#pragma unroll
        for (int h0 = 0; h0 < PPT; h0 += HALF) {
            // phase A: project this half's points, probe the patch cache, put every missing gather in flight
            PointGeom pg[HALF];
            float tap[HALF][NTAP];
            bool miss[HALF];
#pragma unroll
            for (int jj = 0; jj < HALF; ++jj) {
                const int j = h0 + jj;
                const int i = tid + j * nthr;
                project_point(ps, kf[j], pg[jj]);
                const bool cached = i < EDS12_CACHE_CAP;
                const int key = (pg[jj].r0 << 16) ^ (pg[jj].c0 & 0xffff);
                miss[jj] = !(cached && s_cell[i] == key);
                if (miss[jj]) {
                    if (SAMPLING == 0) load_patch16(frame, pg[jj].r0, pg[jj].c0, reinterpret_cast<float(&)[16]>(tap[jj]));
                    else load_patch4(frame, pg[jj].r0, pg[jj].c0, reinterpret_cast<float(&)[4]>(tap[jj]));
                    if (cached) s_cell[i] = key;
                } else {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) tap[jj][t] = s_patch[t][i];
                }
            }
            // phase B: refill the cache; residual and 1x12 row (closed forms of SURVEY §8a); then the wavefront's 64 rows
            // go through LDS into the operand layout of v_mfma_f32_16x16x4_f32, which forms X^T X for X = [J | r] (64 x 13)
            // with four accumulator registers per lane — instead of 91 running sums per lane and a 91-value butterfly.
            // The matrix core is used as a register-free reduction primitive here (exact fp32, same as an fmaf chain),
            // not for throughput: the kernel stays bound by the scattered frame reads.
#pragma unroll
            for (int jj = 0; jj < HALF; ++jj) {
                const int j = h0 + jj;
                const int i = tid + j * nthr;
                const bool valid = i < N;
                if (miss[jj] && i < EDS12_CACHE_CAP) {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) s_patch[t][i] = tap[jj][t];
                }
                float E, Er, Ec;
                if (SAMPLING == 0) bicubic_patch(reinterpret_cast<float(&)[16]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                else bilinear_patch(reinterpret_cast<float(&)[4]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                PointProj pp;
                finish_point(ps, pg[jj], E, Er, Ec, pp);
                // residual blocks touched by this wavefront's 64 consecutive points (usually one: constants then wave-uniform)
                const int i_first = j * nthr + wave * 64;
                const int i_last = (i_first + 63 < N) ? i_first + 63 : N - 1;
                const int b_lo = edsc::uniform_int(block_of(i_first < N ? i_first : 0, ne, nb));
                const int b_hi = edsc::uniform_int(block_of(i_last > 0 ? i_last : 0, ne, nb));
                int myb = b_lo;
                float inv_n, gv[6];
                if (b_lo == b_hi) {
                    const double* bk = s_pose + EDS_PB_BLK + EDS_PB_BLK_STRIDE * b_lo;
                    inv_n = uniformf((float)bk[0]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) gv[k] = uniformf((float)bk[1 + k]);
                } else {
                    myb = block_of(valid ? i : 0, ne, nb);
                    const double* bk = s_pose + EDS_PB_BLK + EDS_PB_BLK_STRIDE * myb;
                    inv_n = (float)bk[0];
#pragma unroll
                    for (int k = 0; k < 6; ++k) gv[k] = (float)bk[1 + k];
                }
                const float w = kw[j];              // 0 for out-of-range lanes: their rows vanish
                float ka[6];
                model_row(kf[j].x, kf[j].y, kf[j].rhop - 1e-5f, kgx[j], kgy[j], ka);
                float m = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) m += ka[k] * vf[k];
                float x[13];
                x[12] = w * (m * inv_n - pp.E);
                x[0] = -w * pp.g0; x[1] = -w * pp.g1; x[2] = -w * pp.g2;
                // quaternion local: -2 w (R X) x gradE_P with R X = P - t
                const float rx = pp.Px - ps.t[0], ry = pp.Py - ps.t[1], rz = pp.Pz - ps.t[2];
                const float w2 = -2.0f * w;
                x[3] = w2 * (ry * pp.g2 - rz * pp.g1);
                x[4] = w2 * (rz * pp.g0 - rx * pp.g2);
                x[5] = w2 * (rx * pp.g1 - ry * pp.g0);
                // velocity part WITHOUT the local-parameterisation projector (I - v v^T/|v|^2)/|v|: that factor is the same
                // for every point, so the solver applies it to the 12 x 12 sums (edsc::coop12_on_eval) instead of 36 FMAs here
#pragma unroll
                for (int k = 0; k < 6; ++k) x[6 + k] = w * (ka[k] * inv_n - m * gv[k]);
                rcand[j] = x[12];
                if (i_first < N) {
                    for (int b = b_lo; b <= b_hi; ++b) {
                        if (b != cb) {
                            if (cb >= 0) flush(C + C2, cb);
                            C = acc4{0, 0, 0, 0}; C2 = acc4{0, 0, 0, 0};
                            cb = b;
                        }
                        const bool on = valid && myb == b;
#pragma unroll
                        for (int c = 0; c < 13; ++c) stage[lane * 17 + c] = on ? x[c] : 0.0f;
                        EDS_WSYNC();
#pragma unroll
                        for (int mm = 0; mm < 16; mm += 2) {       // two independent accumulator chains
                            const float a0 = stage[(4 * mm + (lane >> 4)) * 17 + (lane & 15)];
                            const float a1 = stage[(4 * mm + 4 + (lane >> 4)) * 17 + (lane & 15)];
#if EDS12_MFMA_F64
                            C = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0, (double)a0, C, 0, 0, 0);
                            C2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a1, (double)a1, C2, 0, 0, 0);
#else
                            C = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, C, 0, 0, 0);
                            C2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, C2, 0, 0, 0);
#endif
                        }
                        EDS_WSYNC();
                    }
                }
            }
        }
        if (cb >= 0) flush(C + C2, cb);
        EDS12_STAMP(0);
        __syncthreads();
        EDS12_STAMP(1);
        if (wave == 0) {                        // the LM state machine, spread over this wavefront (eds_solver12_coop.hpp)
            edsc::coop12_on_eval(sv, sums, work, s_pose, lane);
            const int done = edsc::uniform_int(sv.done);
            for (int k = 1 + lane; k < (int)(sizeof(sums) / sizeof(double)); k += 64) reinterpret_cast<double*>(&sums)[k] = 0.0;   // consumed
            if (!done) edsc::coop_fill_pose_block(sv.cp, sv.cq, sv.cv, G, nb, s_pose, lane);
#ifdef EDS_FUSED_STAMPS
            if (lane == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); work.st[6] += n_ - work.st_t; work.st_t = n_; }
#endif
            if (lane == 0) { s_state = done ? 2 : 0; s_accept = work.accepted; }
        }
        __syncthreads();
        EDS12_STAMP(2);
#ifdef EDS_FUSED_STAMPS
        ++st_n;
#endif
        if (s_accept) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) racc[j] = rcand[j];
        }
        if (s_state == 2) break;
    }
    // residuals at the solution (what Tracker.cpp:223-230 writes to kf->residuals): those of the last accepted point
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int i = tid + j * nthr;
        if (i < N) A.r[base + i] = racc[j];
    }

#ifdef EDS_FUSED_STAMPS
    if (tid == 0 && blockIdx.x == 0)
        printf("[stamps12] lane-0 cycles per evaluation: points %llu  reduce %llu  solver %llu  (%d evaluations)\n",
               st_acc[0] / st_n, st_acc[1] / st_n, st_acc[2] / st_n, st_n);
    if (tid == 0 && blockIdx.x == 0)
        printf("[stamps12]   solver split: decide %llu linearise %llu bookkeeping %llu cholesky %llu step %llu tail %llu pose %llu\n",
               work.st[0] / st_n, work.st[1] / st_n, work.st[2] / st_n, work.st[3] / st_n, work.st[4] / st_n, work.st[5] / st_n, work.st[6] / st_n);
#endif
    if (tid == 0) {
        EdsFused12Out& O = out[slot];
        const bool ok = sv.termination != edss::TERM_FAILURE;
        for (int i = 0; i < 3; ++i) O.p[i] = ok ? sv.best_p[i] : sv.p[i];
        for (int i = 0; i < 4; ++i) O.q[i] = ok ? sv.best_q[i] : sv.q[i];
        for (int i = 0; i < 6; ++i) O.v[i] = ok ? sv.best_v[i] : sv.v[i];
        O.initial_cost = sv.initial_cost; O.final_cost = sv.minimum_cost;
        O.termination = sv.termination; O.num_successful = sv.num_successful; O.num_unsuccessful = sv.num_unsuccessful;
        O.failed = ok ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------
bool eds_fused12_supported(const eds_trk* h, int first, int count) {
    int nb = h->cfg.num_blocks < 1 ? 1 : h->cfg.num_blocks;
    return nb <= EDS_DEV_MAX_BLOCKS;            // any number of points: beyond 2 048 the streaming variant takes over
}

int eds_fused12_solve(eds_trk* h, int level, int first, int count) {
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count > 0) return eds_internal_fail(EDS_ERR_STATE, "previous batch not collected: call eds_trk_sync first");
    int lv = level < 0 ? 0 : (level >= EDS_MAX_LEVELS ? EDS_MAX_LEVELS - 1 : level);
    const int iters = h->cfg.max_num_iterations[lv];
    const int nb = h->cfg.num_blocks < 1 ? 1 : h->cfg.num_blocks;
    int maxN = 0;
    for (int s = first; s < first + count; ++s) {
        const Slot& sl = h->slots[s];
        if (!sl.has_kf || !sl.has_frame) return eds_internal_fail(EDS_ERR_STATE, "keyframe or event frame not set");
        if (sl.N > maxN) maxN = sl.N;
        EdsFusedIn& I = fb.h_in[s];
        std::memcpy(I.p, sl.p, sizeof(I.p)); std::memcpy(I.q, sl.q, sizeof(I.q)); std::memcpy(I.v, sl.v, sizeof(I.v));
    }
    hipError_t e = hipMemcpyAsync(fb.d_in + first, fb.h_in + first, sizeof(EdsFusedIn) * count, hipMemcpyHostToDevice, h->st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    const EdsArrays A = h->arrays();
    const int ppt = (maxN + EDS12_THREADS - 1) / EDS12_THREADS;
    const bool bicubic = h->cfg.sampling == EDS_SAMPLE_BICUBIC;
    // Which kernel: the register-resident one (one alignment per CU) has the lower latency (0.29 vs 0.34 ms for one
    // alignment); the streaming one (two alignments per CU, solver phases overlapped) the higher throughput — measured
    // crossover between 64 and 128 alignments, 8.3 M vs 6.2 M LM iterations/s at 1 024.
    bool stream = maxN > EDS12_MAX_POINTS || count >= 96;
    if (const char* e = getenv("EDS_REF12_KERNEL")) {                 // tuning knob: "resident" | "stream"
        if (std::strcmp(e, "stream") == 0) stream = true;
        else if (std::strcmp(e, "resident") == 0 && maxN <= EDS12_MAX_POINTS) stream = false;
    }
    hipEventRecord(h->ev0, h->st);
    if (stream)
        eds_stream12_launch(A, h->cfg.sampling, fb.d_in, fb.d_out12, first, count, iters, h->cfg.loss_type, h->cfg.loss_param,
                            h->cfg.function_tolerance, h->cfg.gradient_tolerance, h->cfg.parameter_tolerance, nb, h->st);
    else {
#define EDS_LAUNCH12(S, P)                                                                                                  \
    hipLaunchKernelGGL((eds_fused12_kernel<S, P>), dim3(count), dim3(EDS12_THREADS), 0, h->st, A, fb.d_in, fb.d_out12, first, \
                       iters, h->cfg.loss_type, h->cfg.loss_param, h->cfg.function_tolerance, h->cfg.gradient_tolerance,    \
                       h->cfg.parameter_tolerance, nb)
    if (ppt <= 1) { if (bicubic) EDS_LAUNCH12(0, 1); else EDS_LAUNCH12(1, 1); }
    else if (ppt <= 2) { if (bicubic) EDS_LAUNCH12(0, 2); else EDS_LAUNCH12(1, 2); }
#if EDS12_MAX_PPT >= 8
    else if (ppt <= 4) { if (bicubic) EDS_LAUNCH12(0, 4); else EDS_LAUNCH12(1, 4); }
    else { if (bicubic) EDS_LAUNCH12(0, 8); else EDS_LAUNCH12(1, 8); }
#else
    else { if (bicubic) EDS_LAUNCH12(0, 4); else EDS_LAUNCH12(1, 4); }
#endif
    }
#undef EDS_LAUNCH12
    hipEventRecord(h->ev1, h->st);
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    e = hipMemcpyAsync(fb.h_out12 + first, fb.d_out12 + first, sizeof(EdsFused12Out) * count, hipMemcpyDeviceToHost, h->st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.pending_first = first;
    fb.pending_count = count;
    fb.pending_kind = 12;
    fb.launch_wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return EDS_OK;
}

int eds_fused12_collect(eds_trk* h) {
    EdsFusedBuffers& fb = h->fused;
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    float dev_ms = 0.f;
    hipEventElapsedTime(&dev_ms, h->ev0, h->ev1);
    for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) {
        Slot& sl = h->slots[s];
        const EdsFused12Out& O = fb.h_out12[s];
        const bool ok = O.failed == 0;
        if (ok) { std::memcpy(sl.p, O.p, sizeof(sl.p)); std::memcpy(sl.q, O.q, sizeof(sl.q)); std::memcpy(sl.v, O.v, sizeof(sl.v)); }
        sl.res_on_device = ok;
        sl.trace_on_device = false;
        sl.residuals.clear();
        sl.ntrace = 0;
        eds_trk_info& in = sl.info;
        std::memset(&in, 0, sizeof(in));
        in.meas_time_us = now - fb.launch_wall_us;
        in.time_seconds = in.meas_time_us * 1e-6;
        in.device_time_us = dev_ms * 1e3;
        in.num_points = sl.N;
        in.num_successful_steps = O.num_successful;
        in.num_unsuccessful_steps = O.num_unsuccessful;
        in.num_iterations = O.num_successful + O.num_unsuccessful;       // Tracker.cpp:211
        in.success = ok;
        in.termination = O.termination;
        in.initial_cost = O.initial_cost;
        in.final_cost = O.final_cost;
    }
    fb.pending_count = 0;
    fb.pending_kind = 0;
    return EDS_OK;
}
