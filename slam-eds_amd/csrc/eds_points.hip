// Per-alignment bookkeeping around the solve, on the GPU:
//
//   k_loss_param     Tracker::getLossParams (reference src/tracking/Tracker.cpp:281-317): median / MAD selection
//                    (tau = 1.345 * 1.4826 * MAD) or the variance-based scale, on the residuals that the last solve
//                    left in HBM — one workgroup per alignment, radix select from the residual plane, 8 bytes back to the host
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

namespace {

// order-preserving map double -> uint64 (and back): a radix select on these keys returns exactly the order statistic a sort would
__device__ __forceinline__ unsigned long long key_of(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

#define EDS_LP_THREADS 256
// The k-th smallest (0-based) of the N keys `key(i)`: most-significant-digit radix select, 8 bits per pass — histogram of the
// candidates' digit in LDS, a wavefront scan picks the bin holding rank k, the candidates narrow to that bin.  Stops as soon as
// one candidate is left (2 000 keys: after 2-3 passes) and fetches it.  O(N) per pass against the O(N log^2 N) compare-exchanges
// and 66 workgroup barriers of the bitonic sort it replaces (60 us per alignment; this: ~5 us), no key buffer in LDS, any N.
template <class KeyFn>
__device__ unsigned long long radix_select(KeyFn key, int N, int k, int tid, int* hist, unsigned long long* s_sel, int* s_cnt) {
    unsigned long long prefix = 0, mask = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += EDS_LP_THREADS) hist[i] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < N; i0 += EDS_LP_THREADS) {           // (uniform trip count: the wavefront votes below need every lane)
            const int i = i0 + tid;
            int digit = -1;                                         // -1: not a candidate
            if (i < N) {
                const unsigned long long q = key(i);
                if ((q & mask) == prefix) digit = (int)((q >> shift) & 255ull);
            }
            // residuals of one alignment share sign / exponent digits: a plain LDS atomic per lane would serialise on one or two
            // bins.  Up to four rounds of "the first pending lane's digit, counted by a ballot, added once"; what is still pending
            // after that is spread over many bins and goes in lane by lane.
#pragma unroll 1
            for (int round = 0; round < 4; ++round) {
                const unsigned long long pending = __ballot(digit >= 0);
                if (pending == 0ull) break;
                const int d = __shfl(digit, __ffsll((long long)pending) - 1, 64);
                const unsigned long long same = __ballot(digit == d);
                if ((tid & 63) == __ffsll((long long)same) - 1) atomicAdd(&hist[d], __popcll(same));
                if (digit == d) digit = -1;
            }
            if (digit >= 0) atomicAdd(&hist[digit], 1);
        }
        __syncthreads();
        if (tid < 64) {                 // bins 4 tid .. 4 tid + 3
            const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const int mine = c0 + c1 + c2 + c3;
            int incl = mine;
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (tid >= off) incl += v; }
            const int excl = incl - mine;
            if (excl <= k && k < incl) {        // exactly one lane
                int r = k - excl, bin = 4 * tid, cnt = c0;
                if (r >= c0) { r -= c0; bin = 4 * tid + 1; cnt = c1;
                    if (r >= c1) { r -= c1; bin = 4 * tid + 2; cnt = c2;
                        if (r >= c2) { r -= c2; bin = 4 * tid + 3; cnt = c3; } } }
                s_sel[0] = prefix | ((unsigned long long)bin << shift);
                s_cnt[0] = r; s_cnt[1] = cnt;
            }
        }
        __syncthreads();
        prefix = s_sel[0]; mask |= 0xffull << shift;
        k = s_cnt[0];
        const int cnt = s_cnt[1];
        __syncthreads();
        if (cnt == 1 && shift > 0) {    // one candidate left: fetch it
            for (int i = tid; i < N; i += EDS_LP_THREADS) {
                const unsigned long long q = key(i);
                if ((q & mask) == prefix) s_sel[0] = q;
            }
            __syncthreads();
            prefix = s_sel[0];
            __syncthreads();
            return prefix;
        }
    }
    return prefix;
}

// out[slot - first] = tau.  method: 1 MAD, 2 STD (EDS_LP_*)
__global__ __launch_bounds__(EDS_LP_THREADS) void k_loss_param(EdsArrays A, int first, int method, double* __restrict__ out) {
    const int slot = first + blockIdx.x, tid = threadIdx.x, nthr = EDS_LP_THREADS;
    const int N = (int)A.pose[(size_t)slot * EDS_POSE_STRIDE + EDS_PB_N];
    const float* __restrict__ r = A.r + (size_t)slot * A.Np;
    __shared__ double s_part[EDS_LP_THREADS / 64];
    __shared__ double s_bcast;
    __shared__ int s_hist[256], s_cnt[2];
    __shared__ unsigned long long s_sel[1];
    if (method == 2) {                  // mean_std_vector returns the VARIANCE (Utils.hpp:272-290)
        if (N == 1) { if (tid == 0) out[blockIdx.x] = 0.0; return; }
        double acc = 0.0;
        for (int i = tid; i < N; i += nthr) acc += (double)r[i];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((tid & 63) == 0) s_part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int w = 0; w < EDS_LP_THREADS / 64; ++w) t += s_part[w]; s_bcast = t / (double)N; }
        __syncthreads();
        const double mu = s_bcast;
        acc = 0.0;
        for (int i = tid; i < N; i += nthr) { const double d = (double)r[i] - mu; acc += d * d / (double)(N - 1); }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_part[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int w = 0; w < EDS_LP_THREADS / 64; ++w) t += s_part[w]; out[blockIdx.x] = 1.345 * t; }
        return;
    }
    // nth_element(N/2): the value at sorted position N/2 (Utils.hpp:315-320), then the same of the absolute deviations
    const double median = val_of(radix_select([&](int i) { return key_of((double)r[i]); }, N, N / 2, tid, s_hist, s_sel, s_cnt));
    const double mad = val_of(radix_select([&](int i) { return key_of(fabs((double)r[i] - median)); }, N, N / 2, tid, s_hist, s_sel, s_cnt));
    if (tid == 0) out[blockIdx.x] = 1.345 * (1.4826 * mad);
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
    if (pbuf->h_tau) hipHostFree(pbuf->h_tau);
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
        (!pb.h_tau && (hipHostMalloc((void**)&pb.h_tau, (size_t)h->B * 8, hipHostMallocMapped) != hipSuccess ||
                       hipHostGetDevicePointer((void**)&pb.d_tau, pb.h_tau, 0) != hipSuccess))) {
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

// the device loss scale selects by radix, straight from the residual plane: any number of points
bool eds_points_supported(const eds_trk* h, int first, int count) {
    for (int s = first; s < first + count; ++s)
        if (h->slots[s].N < 1) return false;
    return true;
}

// tau_out[count]; the residuals must be resident in HBM (res_on_device) for every slot of the range
int eds_points_loss_param(eds_trk* h, int first, int count, int method, double* tau_out) {
    int rc = ensure(h);
    if (rc) return rc;
    EdsPointBuffers& pb = h->point_ops;
    hipLaunchKernelGGL(k_loss_param, dim3(count), dim3(EDS_LP_THREADS), 0, h->st, h->arrays(), first, method, pb.d_tau);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    std::memcpy(tau_out, pb.h_tau, (size_t)count * 8);      // the kernel wrote the scales into mapped pinned memory: no copy call
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
