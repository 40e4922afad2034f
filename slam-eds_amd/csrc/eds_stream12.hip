// Streaming variant of the persistent REF12 kernel (eds_fused12.hip) for LARGE BATCHES and for keyframes with more
// than 2 048 points.
//
// eds_fused12_kernel gives one alignment a whole CU: 512 threads, point constants in registers, 96 KB of patch
// cache.  While wavefront 0 runs the LM state machine (≈ 28 k cycles per evaluation, a third of the total) the other
// seven wavefronts — and three of the four SIMDs — idle.  Here a workgroup is 256 threads with 79 KB of LDS, so TWO
// alignments share a CU and one's solver phase overlaps the other's point phase; per-point constants are re-read
// from HBM/L2 every evaluation (36 B per point, coalesced — small next to the ~200 B scattered frame read), which
// turns the point loop into a real loop (any N) with two points per lane in flight and keeps the register count
// where two wavefronts per SIMD fit.  Candidate residuals go to the (otherwise unused) mhat plane and are copied to
// the residual plane when the candidate is accepted.  Sums, solver and results are those of eds_fused12_kernel.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"
#include "eds_solver12_coop.hpp"

using namespace edsd;

#define EDS12S_THREADS 256
#define EDS12S_WAVES (EDS12S_THREADS / 64)
#ifndef EDS12S_CACHE_CAP
#define EDS12S_CACHE_CAP 320        // 20 KB of patches (measured: 160 .. 640 within 3 %): two workgroups fit a CU with room to spare
#endif
#ifndef EDS12S_WG_PER_CU
#define EDS12S_WG_PER_CU 2
#endif

typedef double acc4d __attribute__((ext_vector_type(4)));

template <int SAMPLING>
__global__ __launch_bounds__(EDS12S_THREADS, EDS12S_WG_PER_CU) void eds_stream12_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                                        EdsFused12Out* __restrict__ out, int first, int iters,
                                                                        int loss_type, double loss_a, double ftol, double gtol,
                                                                        double ptol, int nb) {
    const int slot = first + blockIdx.x;
    const int tid = threadIdx.x;
    constexpr int nthr = EDS12S_THREADS;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NTAP = (SAMPLING == 0) ? 16 : 4;
    __shared__ edss::Solver12 sv;
    __shared__ edss::Sums12Dev sums;
    __shared__ edsc::Work12 work;
    __shared__ double s_pose[EDS_POSE_STRIDE];
    __shared__ float s_stage[EDS12S_WAVES][64 * 17];
    __shared__ int s_state, s_accept;
    __shared__ float s_patch[NTAP][EDS12S_CACHE_CAP];
    __shared__ int s_cell[EDS12S_CACHE_CAP];
    __shared__ double s_G[EDS_DEV_MAX_BLOCKS * 36];

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const double* __restrict__ Gg = A.G + (size_t)slot * EDS_MAX_BLOCKS * 36;
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
            sv.skip_final = 1;              // accepted-point residuals are kept in the residual plane as the solve goes
            sums.nb = nb;
            s_state = 0; s_accept = 0;
#ifdef EDS_FUSED_STAMPS
            for (int k = 0; k < 8; ++k) work.st[k] = 0;
#endif
        }
        EDS_WSYNC();
        edsc::coop_fill_pose_block(sv.cp, sv.cq, sv.cv, s_G, nb, s_pose, lane);
    }
    for (int i = tid; i < EDS12S_CACHE_CAP; i += nthr) s_cell[i] = 0x7fffffff;
    for (int k = tid; k < EDS12S_WAVES * 64 * 17; k += nthr) (&s_stage[0][0])[k] = 0.0f;    // columns 13..15 stay zero for good
    for (int k = tid; k < (int)(sizeof(sums) / sizeof(double)); k += nthr)
        if (k > 0) reinterpret_cast<double*>(&sums)[k] = 0.0;                               // word 0 holds nb
    __syncthreads();

    float* const stage = &s_stage[wave][0];
    auto flush = [&](const acc4d& C, int b) {       // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 r
        const int col = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r;
            const double v = C[r];
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
        acc4d C = {0, 0, 0, 0}, C2 = {0, 0, 0, 0};
        int cb = -1;                    // residual block the tile currently belongs to (wave-uniform)
        for (int j0 = 0; j0 < N; j0 += 2 * nthr) {
            // phase A: two points per lane: constants from HBM/L2, projection, cache probe, gathers in flight
            PointKf kf[2];
            float kw[2], kgx[2], kgy[2];
            PointGeom pg[2];
            float tap[2][NTAP];
            bool miss[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = j0 + jj * nthr + tid;
                const bool valid = i < N;
                const size_t o = base + (valid ? i : 0);
                kf[jj].x = A.x[o]; kf[jj].y = A.y[o]; kf[jj].rhop = A.rho[o] + 1e-5f;
                kf[jj].f0x = A.f0x[o]; kf[jj].f0y = A.f0y[o]; kf[jj].cell0 = A.cell0[o];
                kw[jj] = valid ? A.w[o] : 0.0f;
                kgx[jj] = A.gx[o]; kgy[jj] = A.gy[o];
                project_point(ps, kf[jj], pg[jj]);
                const bool cached = i < EDS12S_CACHE_CAP;
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
            // phase B: residual + 1x12 row (closed forms of SURVEY §8a), rows through LDS into the MFMA (eds_fused12.hip)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int i = j0 + jj * nthr + tid;
                const bool valid = i < N;
                if (miss[jj] && i < EDS12S_CACHE_CAP) {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) s_patch[t][i] = tap[jj][t];
                }
                float E, Er, Ec;
                if (SAMPLING == 0) bicubic_patch(reinterpret_cast<float(&)[16]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                else bilinear_patch(reinterpret_cast<float(&)[4]>(tap[jj]), pg[jj].ay, pg[jj].ax, E, Er, Ec);
                PointProj pp;
                finish_point(ps, pg[jj], E, Er, Ec, pp);
                const int i_first = j0 + jj * nthr + wave * 64;        // this wavefront's 64 consecutive points
                const int i_last = (i_first + 63 < N) ? i_first + 63 : N - 1;
                const int b_lo = edsc::uniform_int(block_of(i_first < N ? i_first : 0, ne, nb));
                const int b_hi = edsc::uniform_int(block_of(i_last > 0 ? i_last : 0, ne, nb));
                int myb = b_lo;
                float inv_n, gv[6];
                if (b_lo == b_hi) {                                  // the usual case: block constants are wave-uniform
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
                const float w = kw[jj];                              // 0 for out-of-range lanes: their rows vanish
                float ka[6];
                model_row(kf[jj].x, kf[jj].y, kf[jj].rhop - 1e-5f, kgx[jj], kgy[jj], ka);
                float m = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) m += ka[k] * vf[k];
                float x[13];
                x[12] = w * (m * inv_n - pp.E);
                x[0] = -w * pp.g0; x[1] = -w * pp.g1; x[2] = -w * pp.g2;
                const float rx = pp.Px - ps.t[0], ry = pp.Py - ps.t[1], rz = pp.Pz - ps.t[2];      // R X = P - t
                const float w2 = -2.0f * w;
                x[3] = w2 * (ry * pp.g2 - rz * pp.g1);
                x[4] = w2 * (rz * pp.g0 - rx * pp.g2);
                x[5] = w2 * (rx * pp.g1 - ry * pp.g0);
#pragma unroll
                for (int k = 0; k < 6; ++k) x[6 + k] = w * (ka[k] * inv_n - m * gv[k]);             // projector applied by the solver
                if (valid) A.mhat[base + i] = x[12];                  // candidate residual
                if (i_first < N) {
                    for (int b = b_lo; b <= b_hi; ++b) {
                        if (b != cb) {
                            if (cb >= 0) flush(C + C2, cb);
                            C = acc4d{0, 0, 0, 0}; C2 = acc4d{0, 0, 0, 0};
                            cb = b;
                        }
                        const bool on = valid && myb == b;
#pragma unroll
                        for (int c = 0; c < 13; ++c) stage[lane * 17 + c] = on ? x[c] : 0.0f;
                        EDS_WSYNC();
#pragma unroll
                        for (int mm = 0; mm < 16; mm += 2) {
                            const float a0 = stage[(4 * mm + (lane >> 4)) * 17 + (lane & 15)];
                            const float a1 = stage[(4 * mm + 4 + (lane >> 4)) * 17 + (lane & 15)];
                            C = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a0, (double)a0, C, 0, 0, 0);
                            C2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a1, (double)a1, C2, 0, 0, 0);
                        }
                        EDS_WSYNC();
                    }
                }
            }
        }
        if (cb >= 0) flush(C + C2, cb);
        __syncthreads();
        if (wave == 0) {                        // the LM state machine, spread over this wavefront (eds_solver12_coop.hpp)
            edsc::coop12_on_eval(sv, sums, work, s_pose, lane);
            const int done = edsc::uniform_int(sv.done);
            for (int k = 1 + lane; k < (int)(sizeof(sums) / sizeof(double)); k += 64) reinterpret_cast<double*>(&sums)[k] = 0.0;
            if (!done) edsc::coop_fill_pose_block(sv.cp, sv.cq, sv.cv, s_G, nb, s_pose, lane);
            if (lane == 0) { s_state = done ? 2 : 0; s_accept = work.accepted; }
        }
        __syncthreads();
        if (s_accept) {                         // the candidate became the accepted point: its residuals are the ones to keep
            for (int i = tid; i < N; i += nthr) A.r[base + i] = A.mhat[base + i];      // each thread copies what it wrote itself
        }
        if (s_state == 2) break;
    }

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

void eds_stream12_launch(const EdsArrays& A, int sampling, const EdsFusedIn* d_in, EdsFused12Out* d_out, int first, int count, int iters,
                         int loss_type, double loss_a, double ftol, double gtol, double ptol, int nb, hipStream_t st) {
    if (sampling == 0)
        hipLaunchKernelGGL((eds_stream12_kernel<0>), dim3(count), dim3(EDS12S_THREADS), 0, st, A, d_in, d_out, first, iters, loss_type, loss_a,
                           ftol, gtol, ptol, nb);
    else
        hipLaunchKernelGGL((eds_stream12_kernel<1>), dim3(count), dim3(EDS12S_THREADS), 0, st, A, d_in, d_out, first, iters, loss_type, loss_a,
                           ftol, gtol, ptol, nb);
}
