// Persistent per-alignment solver kernel for gfx950 (EDS_EXEC_DEVICE).
//
// At N = 2 000 points one Gauss-Newton iteration moves ~0.2 MB and takes a few microseconds —
// two orders of magnitude below a launch + PCIe round trip — so the iteration loop itself has
// to live on the GPU.  One workgroup owns one alignment:
//
//   every lane   strides over the alignment's points: project (fp64), sample the frame (bicubic
//                4x4 taps or bilinear 2x2, through L1/L2), form r and the 1x6 SE(3) row in
//                registers and fold them straight into 28 running sums (J is never written)
//   wavefront    reduce-scatter butterfly (eds_device.hpp), LDS across the wavefronts
//   lane 0       unpacks the sums (fp64), runs the SAME edss::Solver6 state machine the host
//                mode runs (eds_solver.hpp): damped 6x6 Cholesky, exp(xi) T, accept / reject
//
// and loops until the solver reports done; the last pass stores the residuals at the accepted
// pose (what reference Tracker.cpp:223-230 writes to kf->residuals).  Independent alignments
// run on different CUs, B >> 256 fills the chip.  The bound is the frame gather (HBM on first
// touch, L2 afterwards) plus VALU; there is no MFMA-shaped work here.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_math.hpp"
#include "eds_solver.hpp"

using namespace edsd;

#define EDS_FUSED_MAX_WAVES 16

template <int SAMPLING>
__global__ __launch_bounds__(1024) void eds_fused6_kernel(EdsArrays A, const EdsFusedIn* __restrict__ in,
                                                          EdsFusedOut* __restrict__ out, edss::Solver6* __restrict__ sv_all,
                                                          int first, int iters, int damped, double lambda0,
                                                          double huber_tau, int nb) {
    const int slot = first + blockIdx.x;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    __shared__ edss::Solver6 sv;
    __shared__ double s_pose[EDS_PB_BLK + EDS_PB_BLK_STRIDE * EDS_MAX_BLOCKS];
    __shared__ float s_red[EDS_FUSED_MAX_WAVES][EDS_RED_K6];
    __shared__ double s_rec[EDS_RED_K6];
    __shared__ int s_state;            // 0: iterate, 1: this pass is the final one, 2: done

    const double* __restrict__ gpb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)gpb[EDS_PB_N];
    const int ne = N / nb;
    const size_t base = (size_t)slot * A.Np;
    const float* __restrict__ frame = A.frame + (size_t)slot * A.H * A.W;

    if (tid == 0) {
        const EdsFusedIn& I = in[slot];
        for (int i = 0; i < 4; ++i) s_pose[EDS_PB_K + i] = gpb[EDS_PB_K + i];
        edsm::fill_pose_block(I.p, I.q, I.v, A.G + (size_t)slot * EDS_MAX_BLOCKS * 36, nb, s_pose);
        sv.init(damped, iters, lambda0, I.p, I.q);
        s_state = sv.final_pass ? 1 : 0;
    }
    __syncthreads();

    // normalised model for the fixed velocity: mhat_i = a_i.v / n_block(i)
    {
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
        const int state = s_state;
        PoseRT ps;
        load_pose(s_pose, ps);
        float acc[EDS_RED_K6];
#pragma unroll
        for (int j = 0; j < EDS_RED_K6; ++j) acc[j] = 0.0f;
        for (int i = tid; i < N; i += nthr) {
            const size_t o = base + i;
            PointProj pp;
            project_sample<SAMPLING>(frame, A.H, A.W, ps, A.X[o], A.Y[o], A.Z[o], pp);
            const float w = A.w[o];
            const float r = w * (A.mhat[o] - pp.E);
            float J[6];
            J[0] = -w * pp.g0;
            J[1] = -w * pp.g1;
            J[2] = -w * pp.g2;
            J[3] = -w * (pp.Py * pp.g2 - pp.Pz * pp.g1);
            J[4] = -w * (pp.Pz * pp.g0 - pp.Px * pp.g2);
            J[5] = -w * (pp.Px * pp.g1 - pp.Py * pp.g0);
            float hw = 1.0f, ct = r * r;
            if (tau > 0.0f) {
                const float ar = fabsf(r);
                if (ar > tau) hw = tau / ar;
                ct = hw * r * r * (2.0f - hw);
            }
            accumulate_normal<6>(acc, J, r, hw, ct);
            if (state == 1) A.r[o] = r;
        }
        wave_reduce_scatter<EDS_RED_K6>(acc, lane);
        if (lane < 32) s_red[wave][wave_red_index<EDS_RED_K6>(lane, 0)] = acc[0];
        __syncthreads();
        if (tid < EDS_RED_N6) {
            double s = 0.0;
            for (int wv = 0; wv < nwave; ++wv) s += (double)s_red[wv][tid];
            s_rec[tid] = s;
        }
        __syncthreads();
        if (tid == 0) {
            edss::Sums6 S;
            edss::unpack6(s_rec, &S);
            sv.on_eval(S);
            if (sv.done) {
                s_state = 2;
            } else {
                edsm::quat_to_R(sv.cq, s_pose + EDS_PB_R);
                for (int i = 0; i < 3; ++i) s_pose[EDS_PB_T + i] = sv.cp[i];
                s_state = sv.final_pass ? 1 : 0;
            }
        }
        __syncthreads();
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
    // full solver state (trace) to HBM, cooperatively
    {
        const int nwords = (int)(sizeof(edss::Solver6) / sizeof(int));
        const int* src = reinterpret_cast<const int*>(&sv);
        int* dst = reinterpret_cast<int*>(sv_all + slot);
        for (int i = tid; i < nwords; i += nthr) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------------
int eds_fused_alloc(EdsFusedBuffers* fb, int B) {
    fb->B = B;
    if (hipMalloc((void**)&fb->d_in, sizeof(EdsFusedIn) * B) != hipSuccess) return -1;
    if (hipMalloc((void**)&fb->d_out, sizeof(EdsFusedOut) * B) != hipSuccess) return -1;
    if (hipMalloc(&fb->d_sv, sizeof(edss::Solver6) * (size_t)B) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_in, sizeof(EdsFusedIn) * B, hipHostMallocDefault) != hipSuccess) return -1;
    if (hipHostMalloc((void**)&fb->h_out, sizeof(EdsFusedOut) * B, hipHostMallocDefault) != hipSuccess) return -1;
    std::memset(fb->h_in, 0, sizeof(EdsFusedIn) * B);
    std::memset(fb->h_out, 0, sizeof(EdsFusedOut) * B);
    return 0;
}

void eds_fused_free(EdsFusedBuffers* fb) {
    if (fb->d_in) hipFree(fb->d_in);
    if (fb->d_out) hipFree(fb->d_out);
    if (fb->d_sv) hipFree(fb->d_sv);
    if (fb->h_in) hipHostFree(fb->h_in);
    if (fb->h_out) hipHostFree(fb->h_out);
    *fb = EdsFusedBuffers();
}

static int pick_block_threads(int count, int N) {
    // few alignments: spend a whole CU's wave slots on each (latency); many: smaller workgroups so
    // several alignments share a CU and one's serial 6x6 solve hides behind the others' passes
    int t = (count >= 512) ? 256 : (count >= 128 ? 512 : 1024);
    while (t > 64 && t / 2 >= N) t /= 2;
    return t;
}

int eds_fused_solve(eds_trk* h, int level, int first, int count) {
    if (h->cfg.solver == EDS_SOLVER_REF12) return eds_internal_solve_host(h, level, first, count);
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count > 0) return eds_internal_fail(EDS_ERR_STATE, "previous batch not collected: call eds_trk_sync first");
    int lv = level < 0 ? 0 : (level >= EDS_MAX_LEVELS ? EDS_MAX_LEVELS - 1 : level);
    const int iters = h->cfg.max_num_iterations[lv];
    if (iters > EDS_MAX_TRACE) return eds_internal_fail(EDS_ERR_INVALID, "max_num_iterations exceeds EDS_MAX_TRACE for the device solver");
    int nb = h->cfg.num_blocks < 1 ? 1 : (h->cfg.num_blocks > EDS_MAX_BLOCKS ? EDS_MAX_BLOCKS : h->cfg.num_blocks);
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
    const int threads = pick_block_threads(count, maxN);
    const double tau = h->cfg.huber_tau > 0 ? h->cfg.huber_tau : 0.0;
    const int damped = h->cfg.solver == EDS_SOLVER_LM6;
    edss::Solver6* svp = reinterpret_cast<edss::Solver6*>(fb.d_sv);
    hipEventRecord(h->ev0, h->st);
    if (h->cfg.sampling == EDS_SAMPLE_BICUBIC)
        hipLaunchKernelGGL((eds_fused6_kernel<0>), dim3(count), dim3(threads), 0, h->st, A, fb.d_in, fb.d_out, svp, first,
                           iters, damped, h->cfg.lambda0, tau, nb);
    else
        hipLaunchKernelGGL((eds_fused6_kernel<1>), dim3(count), dim3(threads), 0, h->st, A, fb.d_in, fb.d_out, svp, first,
                           iters, damped, h->cfg.lambda0, tau, nb);
    hipEventRecord(h->ev1, h->st);
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    e = hipMemcpyAsync(fb.h_out + first, fb.d_out + first, sizeof(EdsFusedOut) * count, hipMemcpyDeviceToHost, h->st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.pending_first = first;
    fb.pending_count = count;
    fb.launch_wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return EDS_OK;
}

int eds_fused_collect(eds_trk* h) {
    EdsFusedBuffers& fb = h->fused;
    if (fb.pending_count <= 0) return EDS_OK;
    const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    float dev_ms = 0.f;
    hipEventElapsedTime(&dev_ms, h->ev0, h->ev1);
    for (int s = fb.pending_first; s < fb.pending_first + fb.pending_count; ++s) {
        Slot& sl = h->slots[s];
        const EdsFusedOut& O = fb.h_out[s];
        const bool ok = O.failed == 0;
        if (ok) { std::memcpy(sl.p, O.p, sizeof(sl.p)); std::memcpy(sl.q, O.q, sizeof(sl.q)); }
        sl.res_on_device = ok;
        sl.trace_on_device = true;
        sl.residuals.clear();
        sl.ntrace = O.ntrace;
        eds_trk_info& in = sl.info;
        std::memset(&in, 0, sizeof(in));
        in.meas_time_us = now - fb.launch_wall_us;
        in.time_seconds = in.meas_time_us * 1e-6;
        in.device_time_us = dev_ms * 1e3;
        in.num_points = sl.N;
        in.num_iterations = O.iterations;
        in.success = ok;
        in.termination = ok ? edss::TERM_NO_CONVERGENCE : edss::TERM_FAILURE;
        in.num_successful_steps = O.naccepted;
        in.num_unsuccessful_steps = O.ntrace - O.naccepted;
        in.initial_cost = 0.5 * O.initial_cost;
        in.final_cost = 0.5 * O.final_cost;
    }
    fb.pending_count = 0;
    return EDS_OK;
}

int eds_fused_fetch_trace(eds_trk* h, int slot) {
    Slot& sl = h->slots[slot];
    if (!sl.trace_on_device) return EDS_OK;
    edss::Solver6* tmp = new edss::Solver6();
    hipError_t e = hipMemcpy(tmp, reinterpret_cast<edss::Solver6*>(h->fused.d_sv) + slot, sizeof(edss::Solver6), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { delete tmp; return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e)); }
    sl.ntrace = tmp->ntrace;
    sl.tr_xi.assign(&tmp->tr_xi[0][0], &tmp->tr_xi[0][0] + 6 * tmp->ntrace);
    sl.tr_cost.assign(tmp->tr_cost, tmp->tr_cost + tmp->ntrace);
    sl.tr_acc.assign(tmp->tr_acc, tmp->tr_acc + tmp->ntrace);
    sl.trace_on_device = false;
    delete tmp;
    return EDS_OK;
}
