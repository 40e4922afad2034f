// Keyframe point set-up on the device for gfx950.
//
// What the reference does once per keyframe on the CPU with OpenCV (KeyFrame::create, KeyFrame.cpp:333-463):
//   image -> [0,1] -> log(img + 0.2)                          :363-374
//   Sobel 3x3 in x and y, gradient magnitude                  :384-401
//   candidatePoints: 20x20 cells; MAX = k strongest per cell, MEDIAN = everything above the cell median   :740-823
//   norm_coord = (coord - c)/f, grad = Sobel at the pixel     :413-430
//   setDepthMap: nearest depth-map point -> idp, distance -> weight in [0,1]     :1137-1198
//   cleanPoints(0.7): drop weight < 0.7, order preserved      :1566-1587
// Here: all of it in fp64 on the GPU, ending directly in the slot's SoA planes (no N x 6 upload), with the
// index-aligned fp64 arrays kept for the caller's KeyFrame container (eds_trk_get_keyframe_points).
//
// Bandwidth-shaped, integer/selection work — no MFMA.  Selection per cell is rank-by-counting in LDS (a 400-element
// cell needs 160 k comparisons; 768 cells per VGA image), which reproduces cv::minMaxLoc's first-in-row-major tie
// rule and std::nth_element's order statistic without sorting.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "eds_handle.hpp"

#pragma clang fp contract(off)      // sums are formed exactly as written (the oracle states the same association)

namespace {

constexpr int KF_T = 256;
constexpr int KF_MAX_CELL = 32;                     // cell^2 <= 1024 magnitudes in LDS
constexpr double KF_LOG_EPS = (double)0.2f;         // `static constexpr float log_eps = 0.2` (KeyFrame.hpp:54)

__device__ __forceinline__ double load_px(const void* img, int type, size_t i) {
    if (type == 0) return (double)static_cast<const uint8_t*>(img)[i];
    if (type == 1) return (double)static_cast<const float*>(img)[i];
    return static_cast<const double*>(img)[i];
}

// ---- image preparation: what KeyFrame::create does to the image before anything else (KeyFrame.cpp:352-362) ----------------------
// (1) `cv::resize(img, img, out_size, cv::INTER_CUBIC)` when out_scale != 1 — INTER_CUBIC lands in the `fx` parameter, so the
//     interpolation is OpenCV's default INTER_LINEAR, replaced by the 2x2 block mean when both scales are exactly 2;
// (2) `cv::cvtColor(img, img, cv::COLOR_RGB2GRAY)` for a colour image.
// Both restated per element type from OpenCV's published implementation (imgproc/resize.cpp, color_rgb.simd.hpp): uint8 runs in
// fixed point (11-bit interpolation weights, 14-bit luma coefficients 4899 / 9617 / 1868), float in fp32, double in fp64.
__device__ __forceinline__ void kf_resize_coord(int d, double scale, int n_src, int* s0, int* s1, float* f) {
    float fr = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(fr);
    fr -= (float)s;
    if (s < 0) { fr = 0.0f; s = 0; }
    if (s >= n_src - 1) { fr = 0.0f; s = n_src - 1; }
    *s0 = s; *s1 = s + 1 < n_src ? s + 1 : n_src - 1; *f = fr;
}
__device__ __forceinline__ int kf_round_short(float v) {               // saturate_cast<short>(float): round half to even, saturate
    const float r = rintf(v);
    return r > 32767.0f ? 32767 : (r < -32768.0f ? -32768 : (int)r);
}
// one channel `ch` of pixel (r, c) of the H x W image resized from src (sH x sW, `cn` interleaved channels)
template <class T>
__device__ __forceinline__ T kf_resized(const T* __restrict__ src, int sH, int sW, int cn, int ch, int H, int W, int r, int c);
template <>
__device__ __forceinline__ uint8_t kf_resized<uint8_t>(const uint8_t* __restrict__ src, int sH, int sW, int cn, int ch, int H, int W, int r, int c) {
    if (sH == H && sW == W) return src[((size_t)r * sW + c) * cn + ch];
    if (sH == 2 * H && sW == 2 * W) {
        const uint8_t* p = src + ((size_t)(2 * r) * sW + 2 * c) * cn + ch;
        return (uint8_t)((p[0] + p[cn] + p[(size_t)sW * cn] + p[(size_t)sW * cn + cn] + 2) >> 2);
    }
    int x0, x1, y0, y1; float fx, fy;
    kf_resize_coord(c, (double)sW / (double)W, sW, &x0, &x1, &fx);
    kf_resize_coord(r, (double)sH / (double)H, sH, &y0, &y1, &fy);
    const int a0 = kf_round_short((1.0f - fx) * 2048.0f), a1 = kf_round_short(fx * 2048.0f);
    const int b0 = kf_round_short((1.0f - fy) * 2048.0f), b1 = kf_round_short(fy * 2048.0f);
    const int S0 = src[((size_t)y0 * sW + x0) * cn + ch] * a0 + src[((size_t)y0 * sW + x1) * cn + ch] * a1;
    const int S1 = src[((size_t)y1 * sW + x0) * cn + ch] * a0 + src[((size_t)y1 * sW + x1) * cn + ch] * a1;
    return (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}
template <>
__device__ __forceinline__ float kf_resized<float>(const float* __restrict__ src, int sH, int sW, int cn, int ch, int H, int W, int r, int c) {
    if (sH == H && sW == W) return src[((size_t)r * sW + c) * cn + ch];
    if (sH == 2 * H && sW == 2 * W) {
        const float* p = src + ((size_t)(2 * r) * sW + 2 * c) * cn + ch;
        return ((((0.0f + p[0]) + p[cn]) + p[(size_t)sW * cn]) + p[(size_t)sW * cn + cn]) * 0.25f;
    }
    int x0, x1, y0, y1; float fx, fy;
    kf_resize_coord(c, (double)sW / (double)W, sW, &x0, &x1, &fx);
    kf_resize_coord(r, (double)sH / (double)H, sH, &y0, &y1, &fy);
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    const float S0 = src[((size_t)y0 * sW + x0) * cn + ch] * a0 + src[((size_t)y0 * sW + x1) * cn + ch] * a1;
    const float S1 = src[((size_t)y1 * sW + x0) * cn + ch] * a0 + src[((size_t)y1 * sW + x1) * cn + ch] * a1;
    return S0 * b0 + S1 * b1;
}
template <>
__device__ __forceinline__ double kf_resized<double>(const double* __restrict__ src, int sH, int sW, int cn, int ch, int H, int W, int r, int c) {
    if (sH == H && sW == W) return src[((size_t)r * sW + c) * cn + ch];
    if (sH == 2 * H && sW == 2 * W) {
        const double* p = src + ((size_t)(2 * r) * sW + 2 * c) * cn + ch;
        return ((((0.0 + p[0]) + p[cn]) + p[(size_t)sW * cn]) + p[(size_t)sW * cn + cn]) * 0.25;
    }
    int x0, x1, y0, y1; float fx, fy;
    kf_resize_coord(c, (double)sW / (double)W, sW, &x0, &x1, &fx);
    kf_resize_coord(r, (double)sH / (double)H, sH, &y0, &y1, &fy);
    const double a0 = (double)(1.0f - fx), a1 = (double)fx, b0 = (double)(1.0f - fy), b1 = (double)fy;
    const double S0 = src[((size_t)y0 * sW + x0) * cn + ch] * a0 + src[((size_t)y0 * sW + x1) * cn + ch] * a1;
    const double S1 = src[((size_t)y1 * sW + x0) * cn + ch] * a0 + src[((size_t)y1 * sW + x1) * cn + ch] * a1;
    return S0 * b0 + S1 * b1;
}
template <class T>
__global__ __launch_bounds__(KF_T) void k_prepare(const T* __restrict__ src, int sH, int sW, int cn, T* __restrict__ dst, int H, int W) {
    const int c = blockIdx.x * KF_T + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    T v;
    if (cn == 1) {
        v = kf_resized<T>(src, sH, sW, 1, 0, H, W, r, c);
    } else {                            // COLOR_RGB2GRAY on the resized pixel
        const T R = kf_resized<T>(src, sH, sW, cn, 0, H, W, r, c), G = kf_resized<T>(src, sH, sW, cn, 1, H, W, r, c),
                B = kf_resized<T>(src, sH, sW, cn, 2, H, W, r, c);
        if (sizeof(T) == 1) v = (T)(((int)R * 4899 + (int)G * 9617 + (int)B * 1868 + (1 << 13)) >> 14);
        else v = (T)((float)R * 0.299f + (float)G * 0.587f + (float)B * 0.114f);
    }
    dst[(size_t)r * W + c] = v;
}

// block-level min / max; result valid on thread 0
__device__ __forceinline__ void block_minmax(double& mn, double& mx) {
    __shared__ double s_mn[KF_T], s_mx[KF_T];
    s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
    __syncthreads();
    for (int s = KF_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_mn[threadIdx.x] = fmin(s_mn[threadIdx.x], s_mn[threadIdx.x + s]);
            s_mx[threadIdx.x] = fmax(s_mx[threadIdx.x], s_mx[threadIdx.x + s]);
        }
        __syncthreads();
    }
    mn = s_mn[0]; mx = s_mx[0];
    __syncthreads();
}

// partial[2 b], partial[2 b + 1] = min, max over block b's grid-stride share  (cv::minMaxLoc, KeyFrame.cpp:365)
__global__ __launch_bounds__(KF_T) void k_minmax(const void* img, int type, size_t n, double* partial) {
    double mn = INFINITY, mx = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * KF_T + threadIdx.x; i < n; i += (size_t)gridDim.x * KF_T) {
        const double v = load_px(img, type, i);
        mn = fmin(mn, v); mx = fmax(mx, v);
    }
    block_minmax(mn, mx);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = mn; partial[2 * blockIdx.x + 1] = mx; }
}

__device__ __forceinline__ void final_minmax(const double* partial, int nblocks, double& mn, double& mx) {
    mn = INFINITY; mx = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += KF_T) { mn = fmin(mn, partial[2 * b]); mx = fmax(mx, partial[2 * b + 1]); }
    block_minmax(mn, mx);
}

// L = log((img - min)/(max - min) + log_eps)   (KeyFrame.cpp:366,373-374)
__global__ __launch_bounds__(KF_T) void k_log(const void* img, int type, size_t n, const double* partial, int nblocks, double* L) {
    double mn, mx;
    final_minmax(partial, nblocks, mn, mx);
    const double range = mx - mn;
    for (size_t i = (size_t)blockIdx.x * KF_T + threadIdx.x; i < n; i += (size_t)gridDim.x * KF_T)
        L[i] = log((load_px(img, type, i) - mn) / range + KF_LOG_EPS);
}

__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// cv::Sobel(L, CV_64F, 1, 0, 3) / (0, 1, 3), BORDER_REFLECT_101, and cv::cartToPolar's magnitude  (KeyFrame.cpp:384-401)
__global__ __launch_bounds__(KF_T) void k_sobel(const double* __restrict__ L, int H, int W, double* __restrict__ gx,
                                                double* __restrict__ gy, double* __restrict__ mag) {
    const int c = blockIdx.x * KF_T + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const int r0 = reflect101(r - 1, H), r2 = reflect101(r + 1, H), c0 = reflect101(c - 1, W), c2 = reflect101(c + 1, W);
    const double* t = L + (size_t)r0 * W; const double* m = L + (size_t)r * W; const double* b = L + (size_t)r2 * W;
    const double x = ((t[c2] - t[c0]) + 2.0 * (m[c2] - m[c0])) + (b[c2] - b[c0]);
    const double y = ((b[c0] - t[c0]) + 2.0 * (b[c] - t[c])) + (b[c2] - t[c2]);
    const size_t o = (size_t)r * W + c;
    gx[o] = x; gy[o] = y; mag[o] = sqrt(x * x + y * y);
}

// cv::Sobel with aperture 7 (the KeyFrame constructor's, reference KeyFrame.cpp:239-240): separable kernels smooth = [1 6 15 20 15 6 1],
// derivative = [-1 -4 -5 0 5 4 1] (cv::getSobelKernels), no scale, reflect-101 border, CV_64F — the row pass first, then the column
// pass, each in the order OpenCV's symmetric / anti-symmetric filters add: centre term, then the pairs outwards.
__global__ __launch_bounds__(KF_T) void k_sobel7(const double* __restrict__ L, int H, int W, double* __restrict__ gx,
                                                 double* __restrict__ gy, double* __restrict__ mag) {
    const int c = blockIdx.x * KF_T + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    int cc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cc[j] = reflect101(c + j - 3, W);
    double rs[7], rd[7];                       // row pass of the seven rows around r: smoothed / differentiated along x
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double* p = L + (size_t)reflect101(r + k - 3, H) * W;
        rs[k] = ((20.0 * p[cc[3]] + 15.0 * (p[cc[4]] + p[cc[2]])) + 6.0 * (p[cc[5]] + p[cc[1]])) + (p[cc[6]] + p[cc[0]]);
        rd[k] = (5.0 * (p[cc[4]] - p[cc[2]]) + 4.0 * (p[cc[5]] - p[cc[1]])) + (p[cc[6]] - p[cc[0]]);
    }
    const double x = ((20.0 * rd[3] + 15.0 * (rd[4] + rd[2])) + 6.0 * (rd[5] + rd[1])) + (rd[6] + rd[0]);
    const double y = (5.0 * (rs[4] - rs[2]) + 4.0 * (rs[5] - rs[1])) + (rs[6] - rs[0]);
    const size_t o = (size_t)r * W + c;
    gx[o] = x; gy[o] = y; mag[o] = sqrt(x * x + y * y);
}

// One workgroup per cell.  cand[cellid][pos] = local index (row-major inside the cell) in the reference's push order.
// The cell's magnitudes (>= 0, so their bit patterns order like the values) are SORTED once — bitonic network in LDS over (magnitude
// descending, index ascending): position p then holds the element of descending rank p, exact ties resolved by index like the
// reference's repeated arg-max — instead of every element counting its rank against all others (O(n^2) fp64 compares: 44 us for a VGA
// image in 20 x 20 cells; this: ~12 us).
__global__ __launch_bounds__(KF_T) void k_select(const double* __restrict__ mag, int W, int cell, int ncx, int method, int k_per_cell,
                                                 int* __restrict__ cand, int* __restrict__ cnt) {
    constexpr int MAXN = KF_MAX_CELL * KF_MAX_CELL;
    __shared__ unsigned long long v[MAXN];          // bits of the magnitudes, row-major inside the cell
    __shared__ unsigned long long key[MAXN];        // ... sorted
    __shared__ unsigned short idx[MAXN];
    __shared__ int s_count, s_wave[KF_T / 64], s_base;
    const int n2 = cell * cell, tid = threadIdx.x;
    const int cy = blockIdx.x / ncx, cx = blockIdx.x - cy * ncx;
    const int x0 = cx * cell, y0 = cy * cell;
    int M = 2;
    while (M < n2) M <<= 1;
    for (int i = tid; i < M; i += KF_T) {
        const unsigned long long k = i < n2 ? (unsigned long long)__double_as_longlong(mag[(size_t)(y0 + i / cell) * W + x0 + i % cell]) : 0ull;
        if (i < n2) v[i] = k;
        key[i] = k; idx[i] = (unsigned short)i;     // padding: magnitude 0 with an index beyond the cell's: sorts behind every real element
    }
    if (tid == 0) { s_count = 0; s_base = 0; }
    __syncthreads();
    for (int k = 2; k <= M; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < M; i += KF_T) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long ka = key[i], kb = key[ixj];
                    const unsigned short ia = idx[i], ib = idx[ixj];
                    const bool b_first = (kb > ka) || (kb == ka && ib < ia);      // b belongs in front of a
                    if (b_first == ((i & k) == 0)) { key[i] = kb; key[ixj] = ka; idx[i] = ib; idx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
    int* out = cand + (size_t)blockIdx.x * n2;
    int mine = 0;
    if (method == 1) {                       // MEDIAN: every magnitude above the cell median, row-major order (:797-817)
        const unsigned long long med = key[n2 - 1 - n2 / 2];                      // nth_element(size / 2): ascending position n2 / 2  (Utils.cpp:497-498)
        for (int c0 = 0; c0 < n2; c0 += KF_T) {
            const int i = c0 + tid;
            const bool pick = i < n2 && v[i] > med;
            const unsigned long long bal = __ballot(pick);
            const int lane = tid & 63, wave = tid >> 6;
            if (lane == 0) s_wave[wave] = __popcll(bal);
            __syncthreads();
            int before = s_base;
            for (int w = 0; w < wave; ++w) before += s_wave[w];
            if (pick) { out[before + __popcll(bal & ((1ull << lane) - 1ull))] = i; ++mine; }
            __syncthreads();
            if (tid == 0) { int t = 0; for (int w = 0; w < KF_T / 64; ++w) t += s_wave[w]; s_base += t; }
            __syncthreads();
        }
    } else {                                 // MAX: k times arg-max-and-zero; stops once the rest is flat (:768-793)
        const bool flat = key[0] == key[n2 - 1];                                  // max == min: nothing to pick (:784)
        const int kk = k_per_cell < n2 ? k_per_cell : n2;
        for (int p = tid; p < kk; p += KF_T)
            if (!flat && key[p] != 0ull) { out[p] = idx[p]; ++mine; }            // magnitude > 0
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    if (tid == 0) cnt[blockIdx.x] = s_count;
}

// exclusive scan of the per-cell counts (single workgroup); off[ncell] = total
__global__ __launch_bounds__(KF_T) void k_scan_cells(const int* __restrict__ cnt, int ncell, int* __restrict__ off) {
    __shared__ int s[KF_T];
    __shared__ int s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += KF_T) {
        const int i = base + threadIdx.x;
        const int c = i < ncell ? cnt[i] : 0;
        s[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < KF_T; d <<= 1) {
            const int t = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < ncell) off[i] = s_run + s[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == KF_T - 1) s_run += s[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) off[ncell] = s_run;
}

// candidate pixel coordinates and their Sobel gradient, in push order  (KeyFrame.cpp:413-430)
__global__ __launch_bounds__(KF_T) void k_emit(const int* __restrict__ cand, const int* __restrict__ cnt, const int* __restrict__ off,
                                               int cell, int ncx, int W, const double* __restrict__ gx, const double* __restrict__ gy,
                                               double* __restrict__ coord, double* __restrict__ grad) {
    const int cid = blockIdx.x;
    const int n2 = cell * cell, c = cnt[cid], o = off[cid];
    const int cy = cid / ncx, cx = cid - cy * ncx;
    for (int p = threadIdx.x; p < c; p += KF_T) {
        const int li = cand[(size_t)cid * n2 + p];
        const int x = cx * cell + li % cell, y = cy * cell + li / cell;
        coord[2 * (size_t)(o + p)] = (double)x; coord[2 * (size_t)(o + p) + 1] = (double)y;
        grad[2 * (size_t)(o + p)] = gx[(size_t)y * W + x]; grad[2 * (size_t)(o + p) + 1] = gy[(size_t)y * W + x];
    }
}

// Nearest depth-map point of every candidate (KeyFrame.cpp:1137-1166: a brute-force search; lowest index wins exact ties).
// Two launches: the depth points are cut into gridDim.y chunks, workgroup (x, y) finds for its 64 candidates the nearest point of
// chunk y (points staged through LDS, fp64 like the reference); the merge walks the chunks in order with a strict <, so the
// winner is the one a single sequential scan would have found.  (One workgroup per 256 candidates scanning all m points: 95 us
// for 886 candidates x 3 000 points — 4 workgroups on a 256-CU chip.)
constexpr int NN_T = 64, NN_TILE = 512;
__global__ __launch_bounds__(NN_T) void k_nearest_part(const double* __restrict__ coord, int n, const double* __restrict__ dxy, int m, int chunk,
                                                       double* __restrict__ pd2, int* __restrict__ pidx) {
    __shared__ double sx[NN_TILE], sy[NN_TILE];
    const int i = blockIdx.x * NN_T + threadIdx.x;
    const int lo = blockIdx.y * chunk, hi = min(m, lo + chunk);
    const double qx = i < n ? coord[2 * (size_t)i] : 0.0, qy = i < n ? coord[2 * (size_t)i + 1] : 0.0;
    double best = INFINITY;
    int bi = 0;
    for (int base = lo; base < hi; base += NN_TILE) {
        const int len = min(NN_TILE, hi - base);
        __syncthreads();
        for (int j = threadIdx.x; j < len; j += NN_T) { sx[j] = dxy[2 * (size_t)(base + j)]; sy[j] = dxy[2 * (size_t)(base + j) + 1]; }
        __syncthreads();
        for (int j = 0; j < len; ++j) {
            const double dx = qx - sx[j], dy = qy - sy[j];
            const double d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; bi = base + j; }
        }
    }
    if (i < n) { pd2[(size_t)blockIdx.y * n + i] = best; pidx[(size_t)blockIdx.y * n + i] = bi; }
}
__global__ __launch_bounds__(KF_T) void k_nearest_merge(const double* __restrict__ coord, int n, const double* __restrict__ dxy,
                                                        const double* __restrict__ didp, int nchunk, const double* __restrict__ pd2,
                                                        const int* __restrict__ pidx, double* __restrict__ idp, double* __restrict__ dist) {
    const int i = blockIdx.x * KF_T + threadIdx.x;
    if (i >= n) return;
    double best = INFINITY;
    int bi = 0;
    for (int c = 0; c < nchunk; ++c) {
        const double d2 = pd2[(size_t)c * n + i];
        if (d2 < best) { best = d2; bi = pidx[(size_t)c * n + i]; }
    }
    const double qx = coord[2 * (size_t)i], qy = coord[2 * (size_t)i + 1];
    const double dx = dxy[2 * (size_t)bi] - qx, dy = dxy[2 * (size_t)bi + 1] - qy;     // cv::norm(dist)  (:1161-1162)
    idp[i] = didp[bi];
    dist[i] = sqrt(dx * dx + dy * dy);
}

// weights from the distances (:1168-1181), cleanPoints(thr) (:1566-1587): in-place, order-preserving compaction.
// Single workgroup; a chunk is read completely before it is written, and destinations never pass the read front.
__global__ __launch_bounds__(1024) void k_weights_clean(double* __restrict__ coord, double* __restrict__ grad, double* __restrict__ idp,
                                                        double* __restrict__ wd, int n, int has_depth, double const_idp,
                                                        const double* __restrict__ partial, int nblocks, double thr, int* __restrict__ summary) {
    __shared__ double s_mn[1024], s_mx[1024];
    __shared__ int s_wave[16];
    __shared__ int s_run;
    double mn = INFINITY, mx = -INFINITY;
    if (has_depth) {
        for (int b = threadIdx.x; b < nblocks; b += 1024) { mn = fmin(mn, partial[2 * b]); mx = fmax(mx, partial[2 * b + 1]); }
        s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                s_mn[threadIdx.x] = fmin(s_mn[threadIdx.x], s_mn[threadIdx.x + s]);
                s_mx[threadIdx.x] = fmax(s_mx[threadIdx.x], s_mx[threadIdx.x + s]);
            }
            __syncthreads();
        }
        mn = s_mn[0]; mx = s_mx[0];
    }
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        double w = 1.0, c0 = 0, c1 = 0, g0 = 0, g1 = 0, d = const_idp;
        bool keep = false;
        if (i < n) {
            if (has_depth) {
                if (mn != mx) w = 1.0 - ((wd[i] - mn) / (mx - mn));
                d = idp[i];
            }
            keep = !(w < thr);
            c0 = coord[2 * (size_t)i]; c1 = coord[2 * (size_t)i + 1]; g0 = grad[2 * (size_t)i]; g1 = grad[2 * (size_t)i + 1];
        }
        const unsigned long long bal = __ballot(keep);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < 16; ++k) { const int c = s_wave[k]; before += k < wave ? c : 0; total += c; }
        const int run = s_run;
        if (keep) {
            const size_t o = (size_t)(run + before + within);
            coord[2 * o] = c0; coord[2 * o + 1] = c1; grad[2 * o] = g0; grad[2 * o + 1] = g1; idp[o] = d; wd[o] = w;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { summary[0] = n; summary[1] = s_run; }
}

// the slot's fp32 planes from the cleaned fp64 arrays — the same conversion set_keyframe does on the host
__global__ __launch_bounds__(KF_T) void k_fill_slot(EdsArrays A, int slot, int N, double fx, double fy, double cx, double cy,
                                                    const double* __restrict__ coord, const double* __restrict__ grad,
                                                    const double* __restrict__ idp, const double* __restrict__ w) {
    const int i = blockIdx.x * KF_T + threadIdx.x;
    if (i >= A.Np) return;
    const size_t o = (size_t)slot * A.Np + i;
    const bool in = i < N;
    const double nx = in ? (coord[2 * (size_t)i] - cx) / fx : 0.0, ny = in ? (coord[2 * (size_t)i + 1] - cy) / fy : 0.0;   // :417-423
    const_cast<float*>(A.x)[o] = (float)nx;
    const_cast<float*>(A.y)[o] = (float)ny;
    const_cast<float*>(A.rho)[o] = in ? (float)idp[i] : 1.f;
    const_cast<float*>(A.gx)[o] = in ? (float)grad[2 * (size_t)i] : 0.f;
    const_cast<float*>(A.gy)[o] = in ? (float)grad[2 * (size_t)i + 1] : 0.f;
    const_cast<float*>(A.w)[o] = in ? (float)w[i] : 0.f;
    const double u0 = in ? fx * nx + cx : 0.0, v0 = in ? fy * ny + cy : 0.0;
    double cu = floor(u0), cv = floor(v0);
    if (!(cu > -32000.0)) cu = -32000.0; if (cu > 32000.0) cu = 32000.0;
    if (!(cv > -32000.0)) cv = -32000.0; if (cv > 32000.0) cv = 32000.0;
    const_cast<float*>(A.f0x)[o] = (float)(u0 - cu);
    const_cast<float*>(A.f0y)[o] = (float)(v0 - cv);
    const_cast<int*>(A.cell0)[o] = (int)(((unsigned)(int)cv << 16) | ((unsigned)(int)cu & 0xffffu));
}

}  // namespace

void eds_keyframe_free(EdsKeyframeBuffers* kb) {
    if (kb->d_src) hipFree(kb->d_src);
    void* d[] = {kb->d_raw, kb->d_log, kb->d_gx, kb->d_gy, kb->d_mag, kb->d_partial, kb->d_cand, kb->d_cnt, kb->d_off,
                 kb->d_coord, kb->d_grad, kb->d_idp, kb->d_w, kb->d_dxy, kb->d_didp, kb->d_summary};
    for (void* p : d) if (p) hipFree(p);
    *kb = EdsKeyframeBuffers();
}

static int ensure(eds_trk* h) {
    EdsKeyframeBuffers& kb = h->kf_build;
    if (kb.d_raw) return EDS_OK;
    const size_t n = (size_t)h->H * h->W;
    hipError_t e = hipMalloc(&kb.d_raw, n * 8);
    double** planes[] = {&kb.d_log, &kb.d_gx, &kb.d_gy, &kb.d_mag, &kb.d_idp, &kb.d_w};
    for (double** p : planes) if (e == hipSuccess) e = hipMalloc((void**)p, n * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_coord, n * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_grad, n * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_partial, 2 * 256 * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_cand, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_cnt, (n / 4 + 2) * 4);       // cells are at least 2 x 2
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_off, (n / 4 + 2) * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&kb.d_summary, 16);
    if (e != hipSuccess) { eds_keyframe_free(&kb); return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(keyframe set-up buffers)"); }
    return EDS_OK;
}

int eds_keyframe_build(eds_trk* h, int slot, int img_type, const void* img, int img_H, int img_W, int channels, const eds_kf_select* sel,
                       int n_depth, const double* depth_xy, const double* depth_idp, double fx, double fy, double cx, double cy, int* n_points) {
    const int H = h->H, W = h->W;
    const size_t n = (size_t)H * W;
    if (img_H <= 0 || img_W <= 0) { img_H = H; img_W = W; }
    if (channels != 1 && channels != 3) return eds_internal_fail(EDS_ERR_INVALID, "an image has 1 (grey) or 3 (RGB, interleaved) channels");
    if (channels == 3 && img_type == 2) return eds_internal_fail(EDS_ERR_INVALID, "cv::cvtColor takes 8-bit or float colour images, not CV_64F");
    if (img_H < 2 || img_W < 2) return eds_internal_fail(EDS_ERR_INVALID, "bad image size");
    const bool prepare = channels != 1 || img_H != H || img_W != W;
    const int cell = sel->cell;
    if (cell < 2 || cell > KF_MAX_CELL || cell > H || cell > W) return eds_internal_fail(EDS_ERR_INVALID, "cell size must be in [2, 32] and fit the image");
    if (sel->method != EDS_KF_MAX && sel->method != EDS_KF_MEDIAN) return eds_internal_fail(EDS_ERR_INVALID, "unknown point selection method");
    if (img_type < 0 || img_type > 2) return eds_internal_fail(EDS_ERR_INVALID, "img_type must be EDS_IMG_U8, EDS_IMG_F32 or EDS_IMG_F64");
    if (n_depth > 0 && (!depth_xy || !depth_idp)) return eds_internal_fail(EDS_ERR_INVALID, "null depth map");
    int rc = ensure(h);
    if (rc) return rc;
    EdsKeyframeBuffers& kb = h->kf_build;
    kb.last_slot = -1;
    if (n_depth > kb.cap_depth) {
        if (kb.d_dxy) { hipFree(kb.d_dxy); hipFree(kb.d_didp); kb.d_dxy = kb.d_didp = nullptr; }
        kb.cap_depth = n_depth + n_depth / 4 + 256;
        if (hipMalloc((void**)&kb.d_dxy, (size_t)kb.cap_depth * 16) != hipSuccess || hipMalloc((void**)&kb.d_didp, (size_t)kb.cap_depth * 8) != hipSuccess) {
            kb.cap_depth = 0;
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(depth map)");
        }
    }
    hipStream_t st = h->st;
    const size_t px = img_type == 0 ? 1 : (img_type == 1 ? 4 : 8);
    hipError_t e = hipSuccess;
    if (!prepare) {
        e = hipMemcpyAsync(kb.d_raw, img, n * px, hipMemcpyHostToDevice, st);
    } else {                            // resize and / or grey conversion on the device (KeyFrame.cpp:352-362), into the usual buffer
        const size_t src_bytes = (size_t)img_H * img_W * channels * px;
        if (src_bytes > kb.src_bytes) {
            if (kb.d_src) hipFree(kb.d_src);
            kb.d_src = nullptr; kb.src_bytes = 0;
            if (hipMalloc(&kb.d_src, src_bytes) != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(source image)");
            kb.src_bytes = src_bytes;
        }
        e = hipMemcpyAsync(kb.d_src, img, src_bytes, hipMemcpyHostToDevice, st);
        const dim3 g((W + KF_T - 1) / KF_T, H), b(KF_T);
        if (img_type == 0) hipLaunchKernelGGL(k_prepare<uint8_t>, g, b, 0, st, (const uint8_t*)kb.d_src, img_H, img_W, channels, (uint8_t*)kb.d_raw, H, W);
        else if (img_type == 1) hipLaunchKernelGGL(k_prepare<float>, g, b, 0, st, (const float*)kb.d_src, img_H, img_W, channels, (float*)kb.d_raw, H, W);
        else hipLaunchKernelGGL(k_prepare<double>, g, b, 0, st, (const double*)kb.d_src, img_H, img_W, channels, (double*)kb.d_raw, H, W);
    }
    if (e == hipSuccess && n_depth > 0) e = hipMemcpyAsync(kb.d_dxy, depth_xy, (size_t)n_depth * 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_depth > 0) e = hipMemcpyAsync(kb.d_didp, depth_idp, (size_t)n_depth * 8, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    const int NB = 256;
    hipLaunchKernelGGL(k_minmax, dim3(NB), dim3(KF_T), 0, st, kb.d_raw, img_type, n, kb.d_partial);
    hipLaunchKernelGGL(k_log, dim3(NB), dim3(KF_T), 0, st, kb.d_raw, img_type, n, kb.d_partial, NB, kb.d_log);
    if (sel->sobel_ksize == 7) hipLaunchKernelGGL(k_sobel7, dim3((W + KF_T - 1) / KF_T, H), dim3(KF_T), 0, st, kb.d_log, H, W, kb.d_gx, kb.d_gy, kb.d_mag);
    else hipLaunchKernelGGL(k_sobel, dim3((W + KF_T - 1) / KF_T, H), dim3(KF_T), 0, st, kb.d_log, H, W, kb.d_gx, kb.d_gy, kb.d_mag);
    const int ncx = W / cell, ncy = H / cell, ncell = ncx * ncy;       // only whole cells (KeyFrame.cpp:752-754)
    const int k_per_cell = sel->method == EDS_KF_MAX ? (sel->num_points > 0 ? sel->num_points / ncell : 0) : 0;
    hipLaunchKernelGGL(k_select, dim3(ncell), dim3(KF_T), 0, st, kb.d_mag, W, cell, ncx, (int)sel->method, k_per_cell, kb.d_cand, kb.d_cnt);
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(KF_T), 0, st, kb.d_cnt, ncell, kb.d_off);
    hipLaunchKernelGGL(k_emit, dim3(ncell), dim3(KF_T), 0, st, kb.d_cand, kb.d_cnt, kb.d_off, cell, ncx, W, kb.d_gx, kb.d_gy, kb.d_coord, kb.d_grad);
    int ncand = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&ncand, kb.d_off + ncell, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    if (ncand < 1) return eds_internal_fail(EDS_ERR_INVALID, "the selection produced no candidate point");
    const double const_idp = 1.0 / ((sel->max_depth - sel->min_depth) / 2.0);       // KeyFrame.cpp:1189
    if (n_depth > 0) {
        // per-chunk winners go to two planes the selection no longer needs (|grad| and the log image: n doubles each)
        int nchunk = (int)std::min<size_t>(16, n / (size_t)ncand);
        nchunk = std::max(1, std::min(nchunk, (n_depth + 63) / 64));
        const int chunk = (n_depth + nchunk - 1) / nchunk;
        hipLaunchKernelGGL(k_nearest_part, dim3((ncand + NN_T - 1) / NN_T, nchunk), dim3(NN_T), 0, st, kb.d_coord, ncand, kb.d_dxy, n_depth, chunk,
                           kb.d_mag, reinterpret_cast<int*>(kb.d_log));
        hipLaunchKernelGGL(k_nearest_merge, dim3((ncand + KF_T - 1) / KF_T), dim3(KF_T), 0, st, kb.d_coord, ncand, kb.d_dxy, kb.d_didp, nchunk,
                           kb.d_mag, reinterpret_cast<const int*>(kb.d_log), kb.d_idp, kb.d_w);
        hipLaunchKernelGGL(k_minmax, dim3(NB), dim3(KF_T), 0, st, (const void*)kb.d_w, 2, (size_t)ncand, kb.d_partial);
    }
    hipLaunchKernelGGL(k_weights_clean, dim3(1), dim3(1024), 0, st, kb.d_coord, kb.d_grad, kb.d_idp, kb.d_w, ncand, n_depth > 0 ? 1 : 0,
                       const_idp, kb.d_partial, NB, sel->weight_threshold, kb.d_summary);
    int summary[2] = {0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(summary, kb.d_summary, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    const int N = summary[1];
    if (n_points) *n_points = N;
    kb.last_N = N; kb.last_candidates = ncand;
    kb.K[0] = fx; kb.K[1] = fy; kb.K[2] = cx; kb.K[3] = cy;
    if (N < 1) return eds_internal_fail(EDS_ERR_INVALID, "no point survived the weight threshold");
    if (N > h->Nmax) return eds_internal_fail(EDS_ERR_INVALID, "the keyframe has more points than the handle's max_points");
    hipLaunchKernelGGL(k_fill_slot, dim3(h->Np / KF_T), dim3(KF_T), 0, st, h->arrays(), slot, N, fx, fy, cx, cy, kb.d_coord, kb.d_grad, kb.d_idp, kb.d_w);
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    Slot& s = h->slots[slot];
    s.N = N; s.K[0] = fx; s.K[1] = fy; s.K[2] = cx; s.K[3] = cy;
    if ((rc = eds_internal_refresh_gram(h, slot))) return rc;
    s.has_kf = true;
    s.residuals.clear();
    s.res_on_device = false; s.trace_on_device = false; s.ntrace = 0;     // as eds_trk_set_keyframe
    kb.last_slot = slot;
    return EDS_OK;
}

int eds_keyframe_get_points(eds_trk* h, int slot, double* coord_xy, double* norm_xy, double* grad_xy, double* idp, double* weights) {
    EdsKeyframeBuffers& kb = h->kf_build;
    if (kb.last_slot != slot || kb.last_N < 1) return eds_internal_fail(EDS_ERR_STATE, "this slot is not the one eds_trk_build_keyframe filled last");
    const size_t N = (size_t)kb.last_N;
    hipError_t e = hipStreamSynchronize(h->st);
    if (e == hipSuccess && (coord_xy || norm_xy)) {
        double* dst = coord_xy ? coord_xy : norm_xy;
        e = hipMemcpy(dst, kb.d_coord, N * 16, hipMemcpyDeviceToHost);
        if (e == hipSuccess && norm_xy) {
            for (size_t i = 0; i < N; ++i) {                      // KeyFrame.cpp:417-423
                const double x = dst[2 * i], y = dst[2 * i + 1];
                norm_xy[2 * i] = (x - kb.K[2]) / kb.K[0];
                norm_xy[2 * i + 1] = (y - kb.K[3]) / kb.K[1];
            }
        }
    }
    if (e == hipSuccess && grad_xy) e = hipMemcpy(grad_xy, kb.d_grad, N * 16, hipMemcpyDeviceToHost);
    if (e == hipSuccess && idp) e = hipMemcpy(idp, kb.d_idp, N * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && weights) e = hipMemcpy(weights, kb.d_w, N * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    return EDS_OK;
}
