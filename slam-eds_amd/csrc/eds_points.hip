// Per-alignment bookkeeping around the solve, on the GPU:
//
//   k_loss_param     Tracker::getLossParams (reference src/tracking/Tracker.cpp:281-317): median / MAD selection
//                    (tau = 1.345 * 1.4826 * MAD) or the variance-based scale, on the residuals that the last solve
//                    left in HBM — one workgroup per alignment, bitonic sort in LDS, 8 bytes back to the host
//   k_update_points  Tracker::getCoord(delete_out_point) (Tracker.cpp:319-376): re-project every point under the solved
//                    pose, flag the ones that left the frame, compact ALL per-point planes in place keeping their order
//                    (what KeyFrame::erasePoint does one point at a time, KeyFrame.cpp:1060-1106), tracks = new - old
//                    pixel, mean squared flow for Tracker::needNewKeyframe (Tracker.cpp:650-654)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <tuple>

#include <cmath>
#include <cstring>

#include "eds_device.hpp"
#include "eds_fused.hpp"
#include "eds_handle.hpp"
#include "eds_math.hpp"

using namespace edsd;

#define EDS_PTS_THREADS 1024
#define EDS_PTS_CHUNK 4096          // points one sweep of k_update_points keeps in registers (4 per lane)
#define EDS_SORT_MAX 16384          // points k_loss_param can sort in LDS (128 KB of fp64 keys); beyond: the host path

namespace {

__device__ void bitonic_sort(double* s, int M, int tid, int nthr) {
    for (int k = 2; k <= M; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M; i += nthr) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const double a = s[i], b = s[ixj];
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// out[slot - first] = tau.  method: 1 MAD, 2 STD (EDS_LP_*)
__global__ __launch_bounds__(EDS_PTS_THREADS) void k_loss_param(EdsArrays A, int first, int method, double* __restrict__ out) {
    const int slot = first + blockIdx.x, tid = threadIdx.x, nthr = EDS_PTS_THREADS;
    const int N = (int)A.pose[(size_t)slot * EDS_POSE_STRIDE + EDS_PB_N];
    const float* __restrict__ r = A.r + (size_t)slot * A.Np;
    extern __shared__ double s[];       // next power of two >= N keys, sized by the launcher
    __shared__ double s_part[EDS_PTS_THREADS / 64];
    __shared__ double s_bcast;
    if (method == 2) {                  // mean_std_vector returns the VARIANCE (Utils.hpp:272-290)
        if (N == 1) { if (tid == 0) out[blockIdx.x] = 0.0; return; }
        double acc = 0.0;
        for (int i = tid; i < N; i += nthr) acc += (double)r[i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((tid & 63) == 0) s_part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int w = 0; w < EDS_PTS_THREADS / 64; ++w) t += s_part[w]; s_bcast = t / (double)N; }
        __syncthreads();
        const double mu = s_bcast;
        acc = 0.0;
        for (int i = tid; i < N; i += nthr) { const double d = (double)r[i] - mu; acc += d * d / (double)(N - 1); }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int w = 0; w < EDS_PTS_THREADS / 64; ++w) t += s_part[w]; out[blockIdx.x] = 1.345 * t; }
        return;
    }
    int M = 1;
    while (M < N) M <<= 1;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    for (int i = tid; i < M; i += nthr) s[i] = i < N ? (double)r[i] : inf;
    __syncthreads();
    bitonic_sort(s, M, tid, nthr);
    const double median = s[N / 2];     // nth_element(N/2): the value at sorted position N/2 (Utils.hpp:315-320)
    __syncthreads();
    for (int i = tid; i < M; i += nthr) s[i] = i < N ? fabs((double)r[i] - median) : inf;
    __syncthreads();
    bitonic_sort(s, M, tid, nthr);
    if (tid == 0) out[blockIdx.x] = 1.345 * (1.4826 * s[N / 2]);
}

// One workgroup per alignment; per sweep lane t owns the CONTIGUOUS points [t*cppt, (t+1)*cppt) so that an exclusive scan of the
// per-lane keep counts gives order-preserving destinations.
// (blockIdx.x: alignment of a batch — slot first + blockIdx.x, its pose, outputs and summary at their blockIdx.x-th places; coord /
// track / kept may be null: only the compaction, the count and the mean squared flow are wanted)
__global__ __launch_bounds__(EDS_PTS_THREADS) void k_update_points(EdsArrays A, int first, int ppt, int delete_out, const double* __restrict__ pose_in,
                                                                  double* __restrict__ coord, double* __restrict__ track,
                                                                  int* __restrict__ kept, double* __restrict__ summary) {
    const int tid = threadIdx.x;
    const int slot = first + (int)blockIdx.x;
    pose_in += 16 * (size_t)blockIdx.x; summary += 2 * (size_t)blockIdx.x;
    if (coord) coord += 2 * (size_t)A.Np * blockIdx.x;
    if (track) track += 2 * (size_t)A.Np * blockIdx.x;
    if (kept) kept += (size_t)A.Np * blockIdx.x;
    double* pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N];
    const size_t base = (size_t)slot * A.Np;
    __shared__ int s_cnt[EDS_PTS_THREADS];
    __shared__ double s_flow[EDS_PTS_THREADS / 64];
    __shared__ double s_pose[16];
    if (tid < 16) s_pose[tid] = pose_in[tid];      // D (9), t (3), fx, fy, cols, rows
    __syncthreads();
    PoseF ps;
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.D[i] = (float)s_pose[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = (float)s_pose[9 + i];
    ps.fx = (float)s_pose[12]; ps.fy = (float)s_pose[13];
    const double cols = s_pose[14], rows = s_pose[15];

    constexpr int MAXP = EDS_PTS_CHUNK / EDS_PTS_THREADS;
    __shared__ int s_run;
    if (tid == 0) s_run = 0;
    double flow = 0.0;
    // Sweeps of EDS_PTS_CHUNK points.  A sweep reads its points completely before it writes, and a destination never
    // lies beyond the source (points only move towards the front), so the in-place compaction stays order-preserving
    // for any N.
    for (int c0 = 0; c0 < N; c0 += EDS_PTS_CHUNK) {
        const int nc = (N - c0 < EDS_PTS_CHUNK) ? N - c0 : EDS_PTS_CHUNK;
        const int cppt = (nc + EDS_PTS_THREADS - 1) / EDS_PTS_THREADS;
        float fx_[MAXP], fy_[MAXP], frho[MAXP], fgx[MAXP], fgy[MAXP], fw[MAXP], ff0x[MAXP], ff0y[MAXP];
        int fcell[MAXP];
        double xp[MAXP], yp[MAXP];
        bool keep[MAXP];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int li = tid * cppt + k;
            const int i = c0 + li;
            keep[k] = false;
            if (k < cppt && li < nc) {
                const size_t o = base + i;
                fx_[k] = A.x[o]; fy_[k] = A.y[o]; frho[k] = A.rho[o]; fgx[k] = A.gx[o]; fgy[k] = A.gy[o]; fw[k] = A.w[o];
                ff0x[k] = A.f0x[o]; ff0y[k] = A.f0y[o]; fcell[k] = A.cell0[o];
                // p = R (x, y, 1)/mu + t with the RAW inverse depth (Tracker.cpp:343-347), projected (:350-351)
                const float rho = frho[k];
                const float d0 = ps.D[0] * fx_[k] + ps.D[1] * fy_[k] + ps.D[2] + ps.t[0] * rho;
                const float d1 = ps.D[3] * fx_[k] + ps.D[4] * fy_[k] + ps.D[5] + ps.t[1] * rho;
                const float d2 = ps.D[6] * fx_[k] + ps.D[7] * fy_[k] + ps.D[8] + ps.t[2] * rho;
                const float is = 1.0f / (1.0f + d2);
                const double du = (double)(ps.fx * (d0 - fx_[k] * d2) * is), dv = (double)(ps.fy * (d1 - fy_[k] * d2) * is);
                const double u0 = (double)(short)(fcell[k] & 0xffff) + (double)ff0x[k], v0 = (double)(fcell[k] >> 16) + (double)ff0y[k];
                xp[k] = u0 + du; yp[k] = v0 + dv;
                const bool outlier = (xp[k] < 0.0 || xp[k] > cols) || (yp[k] < 0.0 || yp[k] > rows);      // Tracker.cpp:354
                keep[k] = !(delete_out && outlier);
                if (keep[k]) { ++mine; flow += du * du + dv * dv; }                                     // track = new - old pixel (:364-366)
            }
        }
        s_cnt[tid] = mine;
        __syncthreads();
        for (int off = 1; off < EDS_PTS_THREADS; off <<= 1) {       // inclusive Hillis-Steele scan of the keep counts
            const int v = tid >= off ? s_cnt[tid - off] : 0;
            __syncthreads();
            s_cnt[tid] += v;
            __syncthreads();
        }
        const int run = s_run;
        int dst = run + s_cnt[tid] - mine;
        const int ctotal = s_cnt[EDS_PTS_THREADS - 1];
        __syncthreads();        // every lane has read its points and the running offset: the planes can be overwritten in place
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if (!keep[k]) continue;
            const size_t o = base + dst;
            const_cast<float*>(A.x)[o] = fx_[k]; const_cast<float*>(A.y)[o] = fy_[k]; const_cast<float*>(A.rho)[o] = frho[k]; const_cast<float*>(A.gx)[o] = fgx[k]; const_cast<float*>(A.gy)[o] = fgy[k]; const_cast<float*>(A.w)[o] = fw[k];
            const_cast<float*>(A.f0x)[o] = ff0x[k]; const_cast<float*>(A.f0y)[o] = ff0y[k]; const_cast<int*>(A.cell0)[o] = fcell[k];
            if (coord) { coord[2 * dst] = xp[k]; coord[2 * dst + 1] = yp[k]; }
            const double u0 = (double)(short)(fcell[k] & 0xffff) + (double)ff0x[k], v0 = (double)(fcell[k] >> 16) + (double)ff0y[k];
            if (track) { track[2 * dst] = xp[k] - u0; track[2 * dst + 1] = yp[k] - v0; }
            if (kept) kept[dst] = c0 + tid * cppt + k;
            ++dst;
        }
        if (tid == 0) s_run = run + ctotal;
        __syncthreads();
    }
    const int total = s_run;
    for (int off = 32; off > 0; off >>= 1) flow += __shfl_down(flow, off, 64);
    if ((tid & 63) == 0) s_flow[tid >> 6] = flow;
    __syncthreads();
    if (tid == 0) {
        double f = 0.0;
        for (int w = 0; w < EDS_PTS_THREADS / 64; ++w) f += s_flow[w];
        summary[0] = (double)total;
        summary[1] = total > 0 ? f / (double)total : 0.0;        // squared_norm_flow /= idx (:372)
        const int nb = (int)pb[EDS_PB_NB];
        pb[EDS_PB_N] = (double)total;
        pb[EDS_PB_NE] = (double)(total / nb);
    }
}

}  // namespace

void eds_points_free(EdsPointBuffers* pbuf) {
    if (pbuf->h_block) hipHostFree(pbuf->h_block);
    if (pbuf->d_tau) hipFree(pbuf->d_tau);
    *pbuf = EdsPointBuffers();
}

#define EDS_PTS_BATCH 64                // alignments per launch of the batched getCoord (4.7 MB of pinned outputs at 2 048 points)
static int ensure(eds_trk* h, int cap = 1) {
    EdsPointBuffers& pb = h->point_ops;
    if (pb.h_block && pb.cap >= cap) return EDS_OK;
    if (pb.h_block) { hipHostFree(pb.h_block); pb.h_block = nullptr; }
    const size_t Np = (size_t)h->Np;
    const size_t bytes = (size_t)cap * (16 + 128 + Np * 16 + Np * 16 + Np * 4);        // summary | pose | coord | track | kept, `cap` of each
    char* dblock = nullptr;
    if (hipHostMalloc((void**)&pb.h_block, bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&dblock, pb.h_block, 0) != hipSuccess ||
        (!pb.d_tau && hipMalloc((void**)&pb.d_tau, (size_t)h->B * 8) != hipSuccess)) {
        eds_points_free(&pb);
        return eds_internal_fail(EDS_ERR_HIP, "allocation of the point buffers failed");
    }
    pb.cap = cap;
    auto carve = [&](char* base) {
        double* sum = reinterpret_cast<double*>(base);
        double* pose = sum + 2 * (size_t)cap;
        double* coord = pose + 16 * (size_t)cap;
        double* track = coord + 2 * Np * cap;
        int* kept = reinterpret_cast<int*>(track + 2 * Np * cap);
        return std::make_tuple(sum, pose, coord, track, kept);
    };
    std::tie(pb.h_summary, pb.h_pose, pb.h_coord, pb.h_track, pb.h_kept) = carve(pb.h_block);
    std::tie(pb.d_summary, pb.d_pose, pb.d_coord, pb.d_track, pb.d_kept) = carve(dblock);
    return EDS_OK;
}

// the device loss scale sorts in LDS: up to EDS_SORT_MAX points per alignment (else the host nth_element path is used)
bool eds_points_supported(const eds_trk* h, int first, int count) {
    for (int s = first; s < first + count; ++s)
        if (h->slots[s].N > EDS_SORT_MAX || h->slots[s].N < 1) return false;
    return true;
}

// tau_out[count]; the residuals must be resident in HBM (res_on_device) for every slot of the range
int eds_points_loss_param(eds_trk* h, int first, int count, int method, double* tau_out) {
    int rc = ensure(h);
    if (rc) return rc;
    EdsPointBuffers& pb = h->point_ops;
    int maxN = 1;
    for (int s = first; s < first + count; ++s) maxN = h->slots[s].N > maxN ? h->slots[s].N : maxN;
    size_t M = 1;
    while ((int)M < maxN) M <<= 1;
    if (M * 8 > 64 * 1024)              // more LDS than the default per-kernel limit: ask for it explicitly
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_loss_param), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(M * 8));
    hipLaunchKernelGGL(k_loss_param, dim3(count), dim3(EDS_PTS_THREADS), M * 8, h->st, h->arrays(), first, method, pb.d_tau);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(tau_out, pb.d_tau, (size_t)count * 8, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    return EDS_OK;
}

// getCoord (+ culling) of slots [first, first + count) — one workgroup per alignment, EDS_PTS_BATCH alignments per launch.  Outputs of
// alignment b start at index b * stride of the caller's arrays (points) resp. b (n_kept, mean_sq_flow).
int eds_points_update_batch(eds_trk* h, int first, int count, int delete_out, int stride, double* coord_xy, double* tracks_xy, int32_t* kept_index,
                            int* n_kept, double* mean_sq_flow) {
    int rc = ensure(h, std::min(count, EDS_PTS_BATCH));
    if (rc) return rc;
    EdsPointBuffers& pb = h->point_ops;
    const size_t Np = (size_t)h->Np;
    for (int c0 = 0; c0 < count; c0 += pb.cap) {
        const int cn = std::min(pb.cap, count - c0);
        int maxN = 0;
        for (int b = 0; b < cn; ++b) {
            const Slot& sl = h->slots[first + c0 + b];
            double* hp = pb.h_pose + 16 * (size_t)b;                // the kernel reads the poses where the host writes them
            edsm::quat_to_RmI(sl.q, hp);
            for (int i = 0; i < 3; ++i) hp[9 + i] = sl.p[i];
            hp[12] = sl.K[0]; hp[13] = sl.K[1]; hp[14] = (double)h->W; hp[15] = (double)h->H;     // kf->img.cols / rows
            maxN = std::max(maxN, sl.N);
        }
        const int ppt = (maxN + EDS_PTS_THREADS - 1) / EDS_PTS_THREADS;
        hipLaunchKernelGGL(k_update_points, dim3(cn), dim3(EDS_PTS_THREADS), 0, h->st, h->arrays(), first + c0, ppt, delete_out, pb.d_pose,
                           coord_xy ? pb.d_coord : nullptr, tracks_xy ? pb.d_track : nullptr, kept_index ? pb.d_kept : nullptr, pb.d_summary);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->st);
        if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
        for (int b = 0; b < cn; ++b) {
            const int n = (int)pb.h_summary[2 * b];
            const size_t o = (size_t)(c0 + b) * stride;
            if (coord_xy && n > 0) std::memcpy(coord_xy + 2 * o, pb.h_coord + 2 * Np * b, (size_t)n * 16);
            if (tracks_xy && n > 0) std::memcpy(tracks_xy + 2 * o, pb.h_track + 2 * Np * b, (size_t)n * 16);
            if (kept_index && n > 0) std::memcpy(kept_index + o, pb.h_kept + Np * b, (size_t)n * 4);
            if (n_kept) n_kept[c0 + b] = n;
            if (mean_sq_flow) mean_sq_flow[c0 + b] = pb.h_summary[2 * b + 1];
        }
    }
    return EDS_OK;
}

int eds_points_update(eds_trk* h, int slot, int delete_out, double* coord_xy, double* tracks_xy, int32_t* kept_index, int* n_kept,
                      double* mean_sq_flow) {
    // (single alignment: all three outputs are produced — the caller of the single-slot entry point nearly always wants them, and the
    // kernel is the same)
    return eds_points_update_batch(h, slot, 1, delete_out, h->Np, coord_xy, tracks_xy, kept_index, n_kept, mean_sq_flow);
}
