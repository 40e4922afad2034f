// Streaming kernels of the event-to-model tracker for gfx950 (MI355X, CDNA4).
//
//   eds_gram_kernel     once per keyframe: G_k = A_k^T A_k per residual block (fp64)
//   eds_model_kernel    once per velocity: mhat_i = a_i.v / n_block(i)
//   eds_resjac_kernel   the residual / Jacobian pass (reference hot loop
//                       PhotometricError.hpp:152-176 + the Jet<13> autodiff Ceres wraps around it,
//                       here in closed form): one lane per point, SoA in, SoA out
//   eds_reduce_kernel   tall-skinny J^T J / J^T r / sum r^2 reduction: wavefront reduce-scatter,
//                       LDS across the 4 wavefronts, one fp64 record per workgroup
//
// Both bound by HBM, not MFMA (arithmetic intensity ~2 flop/B).  Workgroup -> (slot, chunk)
// mapping is XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so slot = f(id % 8)
// keeps all chunks of one alignment (and its 1.2 MB frame) behind ONE XCD's L2.
#include <hip/hip_runtime.h>

#include "eds_device.hpp"
#include "eds_kernels.hpp"

using namespace edsd;

__device__ __forceinline__ FrameView frame_view(const EdsArrays& A, int slot) {
    return make_frame_view(A.frame, (int)A.pose[(size_t)slot * EDS_POSE_STRIDE + EDS_PB_FRAME], A.H, A.W, A.Hp, A.Wp, A.tiled);
}

// linear workgroup id -> (slot, chunk); all chunks of a slot share id % 8 (one XCD)
__device__ __forceinline__ bool decode_wg(int first, int count, int nchunk, int& slot, int& chunk) {
    const int L = blockIdx.x;
    const int xcd = L & 7, s = L >> 3;
    const int group = s / nchunk;
    chunk = s - group * nchunk;
    const int rel = group * 8 + xcd;
    slot = first + rel;
    return rel < count;
}

// ---------------------------------------------------------------------------------------
// new_rho (optional): the slot's inverse depths as the host just narrowed them into device-mapped pinned memory (set_idepth on the live
// path: Tracker.cpp:167 re-reads the depths on every optimize) — every block stores its own points' values into the rho plane on its
// way, block 0 the padding as well: the depth refresh is ONE launch, no copy call.
__global__ __launch_bounds__(EDS_TPB) void eds_gram_kernel(EdsArrays A, int slot, const float* __restrict__ new_rho) {
    // grid.x = residual block index k; fp64 accumulation of the 21 unique products
    const int k = blockIdx.x;
    const double* pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N], nb = (int)pb[EDS_PB_NB], ne = (int)pb[EDS_PB_NE];
    const int start = k * ne;
    const int n = ne + ((k + 1 == nb) ? (N - (k + 1) * ne) : 0);
    const size_t base = (size_t)slot * A.Np;
    double acc[21];
#pragma unroll
    for (int i = 0; i < 21; ++i) acc[i] = 0.0;
    float* __restrict__ rho_plane = const_cast<float*>(A.rho);
    if (new_rho && k == 0)
        for (int i = N + threadIdx.x; i < A.Np; i += EDS_TPB) rho_plane[base + i] = new_rho[i];       // the padding (1.0)
    for (int i = threadIdx.x; i < n; i += EDS_TPB) {
        const size_t o = base + start + i;
        float rho = A.rho[o];
        if (new_rho) { rho = new_rho[start + i]; rho_plane[o] = rho; }
        float a[6];
        model_row(A.x[o], A.y[o], rho, A.gx[o], A.gy[o], a);
        int c = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = p; q < 6; ++q) acc[c++] += (double)a[p] * (double)a[q];
    }
    __shared__ double sh[EDS_TPB][21];
#pragma unroll
    for (int i = 0; i < 21; ++i) sh[threadIdx.x][i] = acc[i];
    __syncthreads();
    for (int s = EDS_TPB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int i = 0; i < 21; ++i) sh[threadIdx.x][i] += sh[threadIdx.x + s][i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* G = A.G + ((size_t)slot * EDS_MAX_BLOCKS + k) * 36;
        int c = 0;
        for (int p = 0; p < 6; ++p)
            for (int q = p; q < 6; ++q) { G[6 * p + q] = sh[0][c]; G[6 * q + p] = sh[0][c]; ++c; }
    }
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(EDS_TPB) void eds_model_kernel(EdsArrays A, int first, int count, int nchunk) {
    int slot, chunk;
    if (!decode_wg(first, count, nchunk, slot, chunk)) return;
    const double* pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N], nb = (int)pb[EDS_PB_NB], ne = (int)pb[EDS_PB_NE];
    const int i = chunk * EDS_TPB + threadIdx.x;
    if (i >= N) return;
    const size_t o = (size_t)slot * A.Np + i;
    float a[6];
    model_row(A.x[o], A.y[o], A.rho[o], A.gx[o], A.gy[o], a);
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) m += a[k] * (float)pb[EDS_PB_V + k];
    const int blk = block_of(i, ne, nb);
    A.mhat[o] = m * (float)pb[EDS_PB_BLK + EDS_PB_BLK_STRIDE * blk];
}

// ---------------------------------------------------------------------------------------
// Residual + Jacobian pass.  NC = 6: SE(3) left-perturbation columns [d/d upsilon, d/d omega]
// (J = -w [gradE_P, P x gradE_P]; identical to DSO's row, reference CoarseTracker.cpp:311-321).
// NC = 12: Ceres-local columns of the reference problem [t | quaternion local | velocity local].
// NT (bicubic on the strip copies only): non-temporal row loads — see project_sample_quad_strips.
template <int SAMPLING, int NC, bool NT>
__global__ __launch_bounds__(EDS_TPB) void eds_resjac_kernel(EdsArrays A, int first, int count, int nchunk) {
    int slot, chunk;
    if (!decode_wg(first, count, nchunk, slot, chunk)) return;
    const double* __restrict__ pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N];
    const int i = chunk * EDS_TPB + threadIdx.x;
    // bicubic on the tiled frame: the quad-cooperative gather (eds_device.hpp) — four consecutive points per quad, every lane of a
    // quad takes part in the loads whether or not it has a point, so lanes beyond N stay until the sample is formed
    const bool quad = SAMPLING == 0 && A.tiled && A.H < 8000;
    if (i >= N && !(quad && (i & ~3) < N)) return;      // whole quads beyond N (and every lane beyond N without the quad gather) leave
    PoseF ps;
    load_pose(pb, ps);
    const size_t o = (size_t)slot * A.Np + i;           // (i < Np: the planes are padded, a lane beyond N reads padding)
    const FrameView frame = frame_view(A, slot);
    PointKf kf;
    kf.x = A.x[o]; kf.y = A.y[o]; kf.rhop = A.rho[o] + 1e-5f;   // rho' = idp + eps (PhotometricError.hpp:100,200)
    kf.f0x = A.f0x[o]; kf.f0y = A.f0y[o]; kf.cell0 = A.cell0[o];
    PointProj pp;
    if (quad) {
        const int fslot = __builtin_amdgcn_readfirstlane((int)pb[EDS_PB_FRAME]);
        if (A.strips) {                  // the strip copies of the frames are current for this range (the host checked): one load per patch row
            const unsigned copy_bytes = (unsigned)(eds_strips_copy_elems(A.Hp, A.Wp) * 4);
            const char* sbase = reinterpret_cast<const char*>(A.strips) + (size_t)(unsigned)fslot * ((size_t)(2 * A.strip_phases) * copy_bytes);
            project_sample_quad_strips<NT>(frame, sbase, A.Hp, copy_bytes, A.strip_phases, ps, kf, i < N, threadIdx.x & 63, pp);
        } else {
            project_sample_quad(frame, A.frame + (size_t)fslot * A.Hp * A.Wp, ps, kf, i < N, threadIdx.x & 63, pp);
        }
        if (i >= N) return;
    } else {
        project_sample<SAMPLING>(frame, ps, kf, pp);
    }
    const float w = A.w[o];
    const size_t plane = (size_t)A.B * A.Np;
    float* __restrict__ Jo = A.J + o;
    if (NC == 6) {
        A.r[o] = w * (A.mhat[o] - pp.E);
        Jo[0 * plane] = -w * pp.g0;
        Jo[1 * plane] = -w * pp.g1;
        Jo[2 * plane] = -w * pp.g2;
        Jo[3 * plane] = -w * (pp.Py * pp.g2 - pp.Pz * pp.g1);
        Jo[4 * plane] = -w * (pp.Pz * pp.g0 - pp.Px * pp.g2);
        Jo[5 * plane] = -w * (pp.Px * pp.g1 - pp.Py * pp.g0);
    } else {
        const int nb = (int)pb[EDS_PB_NB], ne = (int)pb[EDS_PB_NE];
        const double* __restrict__ bk = pb + EDS_PB_BLK + EDS_PB_BLK_STRIDE * block_of(i, ne, nb);
        float a[6];
        model_row(A.x[o], A.y[o], A.rho[o], A.gx[o], A.gy[o], a);
        float m = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) m += a[k] * (float)pb[EDS_PB_V + k];
        const float inv_n = (float)bk[0];
        // PhotometricErrorNC: the sampled brightness is divided by its block norm too, which is only known after
        // this pass.  Emit the un-normalised pieces (model part in r, E in the mhat plane - unused by the 12-column
        // path -, pose columns of -E without the weight); eds_nc_fix_kernel finishes them.
        const bool ncm = pb[EDS_PB_NCMODE] != 0.0;
        const float wp = ncm ? 1.0f : w;
        if (ncm) { A.r[o] = m * inv_n; A.mhat[o] = pp.E; }
        else A.r[o] = w * (m * inv_n - pp.E);
        Jo[0 * plane] = -wp * pp.g0;
        Jo[1 * plane] = -wp * pp.g1;
        Jo[2 * plane] = -wp * pp.g2;
        // quaternion local: -2 w (R X) x gradE_P, with R X = P - t
        const float rx = pp.Px - ps.t[0], ry = pp.Py - ps.t[1], rz = pp.Pz - ps.t[2];
        const float w2 = -2.0f * wp;
        Jo[3 * plane] = w2 * (ry * pp.g2 - rz * pp.g1);
        Jo[4 * plane] = w2 * (rz * pp.g0 - rx * pp.g2);
        Jo[5 * plane] = w2 * (rx * pp.g1 - ry * pp.g0);
        // velocity part w (a/n - m (G v)/n^3) WITHOUT the local-parameterisation factor (I - v v^T/|v|^2)/|v|: it is the same
        // for every point, so the host applies it in fp64 to the 12 x 12 sums (eds_capi_solve.hip gather12) and to the rows it hands out
#pragma unroll
        for (int k = 0; k < 6; ++k) Jo[(6 + k) * plane] = w * (a[k] * inv_n - m * (float)bk[1 + k]);
    }
}

// ---------------------------------------------------------------------------------------
// PhotometricErrorNC (reference PhotometricErrorNC.hpp:151-186): r_i = w_i (m_i/||m|| - E_i/||E||) with
// ||E||^2 = 1e-3 + sum_j E_j^2 over the residual block.  With J'_j = -dE_j/d(local pose) from the pass above,
//   d(-E_i/||E||) = J'_i/||E|| - E_i (sum_j E_j J'_j)/||E||^3.
// Step 1: one workgroup per (slot, block) forms 1/||E|| and c = sum_j E_j J'_j / ||E||^3 in fp64.
__global__ __launch_bounds__(EDS_TPB) void eds_nc_stat_kernel(EdsArrays A, int first, int count, int nb_grid) {
    int slot, k;
    if (!decode_wg(first, count, nb_grid, slot, k)) return;
    const double* pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N], nb = (int)pb[EDS_PB_NB], ne = (int)pb[EDS_PB_NE];
    if (k >= nb) return;
    const int start = k * ne;
    const int n = ne + ((k + 1 == nb) ? (N - (k + 1) * ne) : 0);
    const size_t plane = (size_t)A.B * A.Np;
    double acc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = 0.0;
    for (int i = threadIdx.x; i < n; i += EDS_TPB) {
        const size_t o = (size_t)slot * A.Np + start + i;
        const double E = (double)A.mhat[o];
        acc[0] += E * E;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[1 + c] += E * (double)A.J[o + c * plane];
    }
    __shared__ double sh[EDS_TPB][7];
#pragma unroll
    for (int i = 0; i < 7; ++i) sh[threadIdx.x][i] = acc[i];
    __syncthreads();
    for (int s = EDS_TPB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int i = 0; i < 7; ++i) sh[threadIdx.x][i] += sh[threadIdx.x + s][i];
        __syncthreads();
    }
    if (threadIdx.x < 7) {
        double* out = A.ncstat + ((size_t)slot * EDS_MAX_BLOCKS + k) * 8;
        const double S = 1e-3 + sh[0][0];                        // meas_norm_sq starts at 1e-3 (PhotometricErrorNC.hpp:151)
        const double inv = 1.0 / sqrt(S);
        out[threadIdx.x] = threadIdx.x == 0 ? inv : sh[0][threadIdx.x] * inv / S;
    }
}

// Step 2: per point, r = w (m/||m|| - E/||E||) and pose columns w (J'/||E|| - E c).
__global__ __launch_bounds__(EDS_TPB) void eds_nc_fix_kernel(EdsArrays A, int first, int count, int nchunk) {
    int slot, chunk;
    if (!decode_wg(first, count, nchunk, slot, chunk)) return;
    const double* __restrict__ pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N], nb = (int)pb[EDS_PB_NB], ne = (int)pb[EDS_PB_NE];
    const int i = chunk * EDS_TPB + threadIdx.x;
    if (i >= N) return;
    const size_t o = (size_t)slot * A.Np + i;
    const size_t plane = (size_t)A.B * A.Np;
    const double* __restrict__ st = A.ncstat + ((size_t)slot * EDS_MAX_BLOCKS + block_of(i, ne, nb)) * 8;
    const float inv = (float)st[0], w = A.w[o], E = A.mhat[o];
    A.r[o] = w * (A.r[o] - E * inv);
#pragma unroll
    for (int c = 0; c < 6; ++c) A.J[o + c * plane] = w * (A.J[o + c * plane] * inv - E * (float)st[1 + c]);
}

// ---------------------------------------------------------------------------------------
// Reduction pass.  Segment s of a slot covers residual block k = s / cpb, points
// [start_k + c*256*PPL, ...) with c = s % cpb, so records never straddle residual blocks
// (the reference applies its loss per block, Tracker.cpp:146-161,192).
// PPL = 4 (the 6-column pass, one residual block): a lane folds FOUR consecutive points — one 16-byte load per plane and lane, seven
// of them in flight before the first product — so the butterfly, the LDS hop and the fp64 record are paid once per 1 024 points instead
// of once per 256.  A pure 28 B/point stream: round 3's one-point-per-lane form ran at 3.9 TB/s.
template <int NC, int PPL>
__global__ __launch_bounds__(EDS_TPB) void eds_reduce_kernel(EdsArrays A, int first, int count, int nseg, int nb_red, int cpb) {
    constexpr int NV = NC * (NC + 1) / 2 + NC + 1;
    constexpr int K = (NC == 6) ? EDS_RED_K6 : EDS_RED_K12;
    constexpr int NOUT = (K >= 64) ? K / 64 : 1;
    int slot, seg;
    if (!decode_wg(first, count, nseg, slot, seg)) return;
    const double* __restrict__ pb = A.pose + (size_t)slot * EDS_POSE_STRIDE;
    const int N = (int)pb[EDS_PB_N];
    int start = 0, n = N;
    const int k = seg / cpb, c = seg - k * cpb;
    if (nb_red > 1) {
        const int ne = N / nb_red;
        start = k * ne;
        n = ne + ((k + 1 == nb_red) ? (N - (k + 1) * ne) : 0);
    }
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.0f;
    const size_t plane = (size_t)A.B * A.Np;
    const float tau = (float)pb[EDS_PB_HUBER];
    if constexpr (PPL == 1) {
        const int li = c * EDS_TPB + threadIdx.x;
        if (li < n) {
            const size_t o = (size_t)slot * A.Np + start + li;
            float J[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) J[j] = A.J[o + j * plane];
            const float r = A.r[o];
            float hw = 1.0f, ct = r * r;
            if (NC == 6 && tau > 0.0f) {           // per-point Huber (extension; cf. CoarseTracker.cpp:445)
                const float ar = fabsf(r);
                if (ar > tau) hw = tau / ar;
                ct = hw * r * r * (2.0f - hw);
            }
            accumulate_normal<NC>(acc, J, r, hw, ct);
        }
    } else {
        static_assert(PPL == 1 || PPL == 4 || PPL == 8, "one point per lane, or four / eight consecutive ones (16-byte loads)");
        // nb_red == 1 here (start = 0): element 4 * k of a plane is 16-byte aligned (Np is a multiple of 256), and a group of
        // four lies wholly inside or wholly outside the padded plane
        constexpr int G = PPL / 4;                   // 16-byte groups per lane and plane: all of them in flight before the first product
        const int li = (c * EDS_TPB + threadIdx.x) * PPL;
        if (li < n) {
            const size_t o = (size_t)slot * A.Np + li;
            float4 Jv[G][NC], rv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool in = li + 4 * g < A.Np;
#pragma unroll
                for (int j = 0; j < NC; ++j) Jv[g][j] = in ? *reinterpret_cast<const float4*>(A.J + o + 4 * g + j * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
                rv[g] = in ? *reinterpret_cast<const float4*>(A.r + o + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float r4[4] = {rv[g].x, rv[g].y, rv[g].z, rv[g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float J[NC];
#pragma unroll
                    for (int j = 0; j < NC; ++j) J[j] = e == 0 ? Jv[g][j].x : (e == 1 ? Jv[g][j].y : (e == 2 ? Jv[g][j].z : Jv[g][j].w));
                    const bool on = li + 4 * g + e < n;    // the padding behind N holds whatever was there
                    const float r = on ? r4[e] : 0.0f;
#pragma unroll
                    for (int j = 0; j < NC; ++j) J[j] = on ? J[j] : 0.0f;
                    float hw = 1.0f, ct = r * r;
                    if (NC == 6 && tau > 0.0f) {
                        const float ar = fabsf(r);
                        if (ar > tau) hw = tau / ar;
                        ct = hw * r * r * (2.0f - hw);
                    }
                    accumulate_normal<NC>(acc, J, r, hw, ct);
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_reduce_scatter<K>(acc, lane);
    __shared__ float sh[EDS_TPB / 64][K];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        if (K >= 64 || lane < 32) sh[wave][wave_red_index<K>(lane, j)] = acc[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < EDS_TPB / 64; ++wv) s += (double)sh[wv][threadIdx.x];
        A.part[((size_t)slot * A.max_seg + seg) * EDS_RED_K + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------
// launchers (host side, same translation unit so the templates are instantiated here)
static inline int grid_for(int count, int per_slot) { return ((count + 7) / 8) * 8 * per_slot; }

void eds_launch_gram(const EdsArrays& A, int slot, int nb, hipStream_t st, const float* new_rho) {
    hipLaunchKernelGGL(eds_gram_kernel, dim3(nb), dim3(EDS_TPB), 0, st, A, slot, new_rho);
}
void eds_launch_model(const EdsArrays& A, int first, int count, int nchunk, hipStream_t st) {
    hipLaunchKernelGGL(eds_model_kernel, dim3(grid_for(count, nchunk)), dim3(EDS_TPB), 0, st, A, first, count, nchunk);
}
void eds_launch_resjac(const EdsArrays& A, int sampling, int ncols, int first, int count, int nchunk, hipStream_t st) {
    const dim3 g(grid_for(count, nchunk)), b(EDS_TPB);
    // non-temporal row loads once the frames of the range cannot stay in the 256 MiB Infinity Cache between two passes anyway: a pass
    // touches ~128 B per point on the strip copies (eds_resjac_nt_rule)
    const bool nt = sampling == 0 && A.strips && eds_resjac_nt_rule(count, A.Np);
    if (sampling == 0 && ncols == 6 && nt) hipLaunchKernelGGL((eds_resjac_kernel<0, 6, true>), g, b, 0, st, A, first, count, nchunk);
    else if (sampling == 0 && ncols == 6) hipLaunchKernelGGL((eds_resjac_kernel<0, 6, false>), g, b, 0, st, A, first, count, nchunk);
    else if (sampling == 0 && nt) hipLaunchKernelGGL((eds_resjac_kernel<0, 12, true>), g, b, 0, st, A, first, count, nchunk);
    else if (sampling == 0) hipLaunchKernelGGL((eds_resjac_kernel<0, 12, false>), g, b, 0, st, A, first, count, nchunk);
    else if (ncols == 6) hipLaunchKernelGGL((eds_resjac_kernel<1, 6, false>), g, b, 0, st, A, first, count, nchunk);
    else hipLaunchKernelGGL((eds_resjac_kernel<1, 12, false>), g, b, 0, st, A, first, count, nchunk);
}
void eds_launch_nc_normalise(const EdsArrays& A, int first, int count, int nb, int nchunk, hipStream_t st) {
    hipLaunchKernelGGL(eds_nc_stat_kernel, dim3(grid_for(count, nb)), dim3(EDS_TPB), 0, st, A, first, count, nb);
    hipLaunchKernelGGL(eds_nc_fix_kernel, dim3(grid_for(count, nchunk)), dim3(EDS_TPB), 0, st, A, first, count, nchunk);
}
void eds_launch_reduce(const EdsArrays& A, int ncols, int first, int count, int nseg, int nb_red, int cpb, hipStream_t st, int ppl) {
    const dim3 g(grid_for(count, nseg)), b(EDS_TPB);
    // (the 6-column pass always reduces one block per slot: the four-points-per-lane form; eds_reduce_points_per_lane says so to the host)
    if (ncols == 6 && nb_red == 1 && ppl == 8) hipLaunchKernelGGL((eds_reduce_kernel<6, 8>), g, b, 0, st, A, first, count, nseg, nb_red, cpb);
    else if (ncols == 6 && nb_red == 1) hipLaunchKernelGGL((eds_reduce_kernel<6, 4>), g, b, 0, st, A, first, count, nseg, nb_red, cpb);
    else if (ncols == 6) hipLaunchKernelGGL((eds_reduce_kernel<6, 1>), g, b, 0, st, A, first, count, nseg, nb_red, cpb);
    else hipLaunchKernelGGL((eds_reduce_kernel<12, 1>), g, b, 0, st, A, first, count, nseg, nb_red, cpb);
}
