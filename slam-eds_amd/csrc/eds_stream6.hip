// Streaming variants of the persistent pose-only kernel (eds_fused.hip): per-point constants re-read every pass, any N.
// Since round 2 optimize() uses them only above 2 048 points per alignment when teams do not apply ("wide": one 512-thread
// workgroup with the whole patch cache); the "paired" shape described next lost its place to the register-resident kernel with
// the quad-cooperative gather and prepared candidates (13.2 M vs 10.5 M iterations/s at 4 096 alignments) and is kept for A/B
// runs (EDS_LM6_KERNEL=paired) and for its tests.
//
// eds_fused6_kernel gives one alignment a whole CU (512 threads, constants in registers, 128 KB patch cache); while
// lane 0 solves the 6x6 system and the wavefronts wait at the reduction (≈ 27 % of a pass on the bench workload) the
// CU does nothing else.  Here a workgroup is 256 threads with half the patch cache, so TWO alignments share a CU
// and one's reduction / solver phase overlaps the other's point phase.  The pose-only pass is bound by the scattered
// frame reads, so this is a trade: half the cache per alignment costs ~17 % more gathers, the overlap wins back more —
// measured +0 % at 1 024 alignments per launch, +3 % at 1 536, +5 % at 4 096; with three or four smaller workgroups per
// CU (less cache still, more gathers in flight) it LOSES 10-30 %.  (Round 1 picked it from 1 536 alignments per launch.)
// Per-point constants are re-read from HBM/L2 every pass (28 B per point, coalesced), two points per lane are in flight, candidate residuals go to plane 0 of the
// (otherwise unused) Jacobian buffer and are copied to the residual plane when the pose is accepted.  Solver
// (edss::Solver6), sums, trace and results are those of eds_fused6_kernel.
#include <hip/hip_runtime.h>

#include <cstring>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"

using namespace edsd;

#define EDS6S_THREADS 256
#define EDS6S_CACHE_CAP 1024        // 64 KB of patches per workgroup: the most that lets two workgroups share a CU
#ifndef EDS6S_WG_PER_CU
#define EDS6S_WG_PER_CU 2
#endif
#ifndef EDS6S_INFLIGHT
#define EDS6S_INFLIGHT 2            // points per lane whose gathers are in flight together
#endif

template <int SAMPLING, int NTHR, int CAP>
__global__ __launch_bounds__(NTHR, EDS6S_WG_PER_CU) void eds_stream6_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                                                  EdsFusedOut* __restrict__ out, edss::Solver6* __restrict__ sv_all,
                                                                                  int first, int iters, int damped, double lambda0,
                                                                                  double huber_tau, int nb) {
    const int slot = first + blockIdx.x;
    const int tid = threadIdx.x;
    constexpr int nthr = NTHR;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NTAP = (SAMPLING == 0) ? 16 : 4;
    __shared__ edss::Solver6 sv;
    __shared__ double s_pose[EDS_POSE_STRIDE];
    __shared__ float s_red[(NTHR / 64)][EDS_RED_K6];
    __shared__ edss::Sums6 s_sums;
    __shared__ int s_state;            // 0: iterate, 1: this pass is the final one, 2: done
    __shared__ int s_accept;           // the pass just consumed is at the accepted pose
    __shared__ float s_patch[NTAP][CAP];
    __shared__ int s_cell[CAP];

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)gpb[EDS_PB_N];
    const int ne = N / nb;
    const size_t base = (size_t)slot * A.Np;
    const FrameView frame = make_frame_view(A.frame, (int)gpb[EDS_PB_FRAME], A.H, A.W, A.Hp, A.Wp, 1);     // persistent kernels: tiled frames only (eds_fused_solve)
    float* __restrict__ rcand = A.J + base;              // plane 0 of the Jacobian buffer: residuals of the pass in flight

    if (tid == 0) {
        const EdsFusedIn& I = in[slot];
        for (int i = 0; i < 4; ++i) s_pose[EDS_PB_K + i] = gpb[EDS_PB_K + i];
        edsm::fill_pose_block(I.p, I.q, I.v, A.G + (size_t)slot * EDS_MAX_BLOCKS * 36, nb, s_pose);
        sv.init(damped, iters, lambda0, I.p, I.q, damped ? 1 : 0);     // damped: accepted-pose residuals are kept as the solve goes
        s_state = sv.final_pass ? 1 : 0;
        s_accept = 0;
    }
    for (int i = tid; i < CAP; i += nthr) s_cell[i] = 0x7fffffff;
    __syncthreads();
    {   // normalised model for the fixed velocity, mhat_i = a_i.v / n_block(i), once per solve
        float vf[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) vf[k] = (float)s_pose[EDS_PB_V + k];
        for (int i = tid; i < N; i += nthr) {
            const size_t o = base + i;
            float a[6];
            model_row(A.x[o], A.y[o], A.rho[o], A.gx[o], A.gy[o], a);
            float m = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) m += a[k] * vf[k];
            A.mhat[o] = m * (float)s_pose[EDS_PB_BLK + EDS_PB_BLK_STRIDE * block_of(i, ne, nb)];
        }
    }
    const float tau = (float)huber_tau;

    for (;;) {
        PoseF ps;
        load_pose(s_pose, ps);
        float acc[EDS_RED_K6];
#pragma unroll
        for (int j = 0; j < EDS_RED_K6; ++j) acc[j] = 0.0f;
        for (int j0 = 0; j0 < N; j0 += EDS6S_INFLIGHT * nthr) {
            // phase A: EDS6S_INFLIGHT points per lane: constants, projection, cache probe, gathers in flight
            PointGeom pg[EDS6S_INFLIGHT];
            float tap[EDS6S_INFLIGHT][NTAP], kw[EDS6S_INFLIGHT], kmh[EDS6S_INFLIGHT];
            bool miss[EDS6S_INFLIGHT];
#pragma unroll
            for (int jj = 0; jj < EDS6S_INFLIGHT; ++jj) {
                const int i = j0 + jj * nthr + tid;
                const bool valid = i < N;
                const size_t o = base + (valid ? i : 0);
                PointKf kf;
                const float* __restrict__ c = A.kf + o;                     // one base pointer, nine planes (eds_layout.hpp EDS_KF_*)
                const size_t pl = A.kf_plane;
                kf.x = c[EDS_KF_X * pl]; kf.y = c[EDS_KF_Y * pl]; kf.rhop = c[EDS_KF_RHO * pl] + 1e-5f;   // rho' = idp + eps (PhotometricError.hpp:100,200)
                kf.f0x = c[EDS_KF_F0X * pl]; kf.f0y = c[EDS_KF_F0Y * pl]; kf.cell0 = __float_as_int(c[EDS_KF_CELL0 * pl]);
                kw[jj] = valid ? c[EDS_KF_W * pl] : 0.0f;                   // w = 0 silences out-of-range lanes
                kmh[jj] = A.mhat[o];
                project_point(ps, kf, pg[jj]);
                const bool cached = i < CAP;
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
            // phase B: refill the cache, residual, 1x6 row, running sums
#pragma unroll
            for (int jj = 0; jj < EDS6S_INFLIGHT; ++jj) {
                const int i = j0 + jj * nthr + tid;
                if (miss[jj] && i < CAP) {
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) s_patch[t][i] = tap[jj][t];
                }
                const float r = point_row6<SAMPLING, NTAP>(ps, pg[jj], tap[jj], kw[jj], kmh[jj], tau, acc);
                if (i < N) rcand[i] = r;
            }
        }
        wave_reduce_scatter<EDS_RED_K6>(acc, lane);
        if (lane < 32) s_red[wave][wave_red_index<EDS_RED_K6>(lane, 0)] = acc[0];
        __syncthreads();
        if (tid < EDS_RED_N6) {          // cross-wavefront sum in fp64, unpacked straight into the solver's input
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < (NTHR / 64); ++wv) s += (double)s_red[wv][tid];
            if (tid < 21) {
                int a = 0, rem = tid;           // record index -> (a, b) of the upper triangle
                while (rem >= 6 - a) { rem -= 6 - a; ++a; }
                const int b = a + rem;
                s_sums.H[6 * a + b] = s;
                s_sums.H[6 * b + a] = s;
            } else if (tid < 27) {
                s_sums.b[tid - 21] = s;
            } else {
                s_sums.cost = s;
            }
        }
        // the summing lanes and the solver lane live in wavefront 0: LDS operations of one wavefront retire in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid == 0) {
            sv.on_eval(s_sums);
            s_accept = sv.last_accepted;
            if (sv.done) {
                s_state = 2;
            } else {
                edsm::fill_pose_rt(sv.cp, sv.cq, s_pose);
                s_state = sv.final_pass ? 1 : 0;
            }
        }
        __syncthreads();
        if (s_accept) {                         // the pass just consumed is at the accepted pose: keep its residuals
            for (int i = tid; i < N; i += nthr) A.r[base + i] = rcand[i];      // each thread copies what it wrote itself
        }
        if (s_state == 2) break;
    }

    if (tid == 0) {
        EdsFusedOut& O = out[slot];
        for (int i = 0; i < 3; ++i) O.p[i] = sv.p[i];
        for (int i = 0; i < 4; ++i) O.q[i] = sv.q[i];
        O.initial_cost = sv.initial_cost; O.final_cost = sv.final_cost;
        O.iterations = sv.iter; O.ntrace = sv.ntrace; O.failed = sv.failed;
        int na = 0;
        for (int k = 0; k < sv.ntrace; ++k) na += sv.tr_acc[k];
        O.naccepted = na;
    }
    {   // full solver state (trace) to HBM, cooperatively
        const int nwords = (int)(sizeof(edss::Solver6) / sizeof(int));
        const int* src = reinterpret_cast<const int*>(&sv);
        int* dst = reinterpret_cast<int*>(sv_all + slot);
        for (int i = tid; i < nwords; i += nthr) dst[i] = src[i];
    }
}

void eds_stream6_launch(const EdsArrays& A, int sampling, int wide, const EdsFusedIn* d_in, EdsFusedOut* d_out, void* d_sv, int first,
                        int count, int iters, int damped, double lambda0, double huber_tau, int nb, hipStream_t st) {
    edss::Solver6* svp = reinterpret_cast<edss::Solver6*>(d_sv);
#define EDS_LAUNCH6S(S, T, C)                                                                                                    \
    hipLaunchKernelGGL((eds_stream6_kernel<S, T, C>), dim3(count), dim3(T), 0, st, A, d_in, d_out, svp, first, iters, damped, lambda0, \
                       huber_tau, nb)
    // paired: two 256-thread workgroups per CU; wide: one 512-thread workgroup with the whole patch cache (few alignments
    // of a large keyframe, where the register-resident kernel would need 1 024 threads at 128 registers)
    if (wide) { if (sampling == 0) EDS_LAUNCH6S(0, 512, 2048); else EDS_LAUNCH6S(1, 512, 2048); }
    else { if (sampling == 0) EDS_LAUNCH6S(0, EDS6S_THREADS, EDS6S_CACHE_CAP); else EDS_LAUNCH6S(1, EDS6S_THREADS, EDS6S_CACHE_CAP); }
#undef EDS_LAUNCH6S
}
