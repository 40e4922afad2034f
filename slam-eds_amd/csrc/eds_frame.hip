// Event-frame construction on the GPU: events -> brightness-increment image -> the tiled
// fp32 frame the tracker samples, without the 2.46 MB host round trip per slice.
//
// Replaces, for out_scale == 1, reference EventFrame::create (src/tracking/EventFrame.cpp:302-389) and its helper
// eds::utils::drawValuesPoints(..., "bilinear", 0.5, true) (src/utils/Utils.cpp:50-122):
//   1. per event: undistort through the forward LUT (EventFrame.cpp:316-317), polarity +-1 (:318)
//   2. 4-tap bilinear vote with the exponential window weight expWeight(i/N, 1) (Utils.cpp:66,83-107; Utils.hpp:542-546)
//      — fp64 atomics, so the sum differs from the reference's sequential fp64 loop only by addition order
//   3. 3x3 Gaussian, sigma 0.5, reflect-101 borders (Utils.cpp:113-119: ksize = (int(1.25*240/100), int(1.7*180/100)))
//   4. optional morphological level i >= 1: dilate + erode with a (2i+1)^2 box (EventFrame.cpp:350-357)
//   5. divide by the Frobenius norm (EventFrame.cpp:359-383) and store as fp32 in the handle's frame layout
#include <hip/hip_runtime.h>

#include <climits>
#include <algorithm>
#include <cmath>
#include <cstring>

#include "eds_fused.hpp"
#include "eds_handle.hpp"

namespace {

// one event of a slice of n events (i: its position in the slice, for the window weight) into `img` (drawValuesPoints, Utils.cpp:50-122)
__device__ __forceinline__ void vote_one(int px, int py, int positive, int i, int n, const float* __restrict__ mapx, const float* __restrict__ mapy,
                                         int H, int W, int use_exp, double* __restrict__ img) {
    double ux = px, uy = py;
    if (mapx) {
        const int cx = px < W ? px : W - 1, cy = py < H ? py : H - 1;     // the LUT has the sensor's size
        ux = (double)mapx[(size_t)cy * W + cx];
        uy = (double)mapy[(size_t)cy * W + cx];
    }
    double weight = 1.0;
    if (use_exp) {                      // expWeight(idx / window_size, 1.0)
        const double value = ((double)i / (double)n - 0.5) / (1.0 / 6.0);
        weight = exp(-0.5 * value * value);
    }
    const double val = weight * (positive ? 1.0 : -1.0);
    int x0 = (int)floor(ux), y0 = (int)floor(uy);
    int x1 = x0 + 1, y1 = y0 + 1;
    // voting weights; 0 if the tap is outside the image (Utils.cpp:92-95)
    const double wa = (x0 < W && y0 < H && x0 >= 0 && y0 >= 0) ? (x1 - ux) * (y1 - uy) : 0.0;
    const double wb = (x0 < W && y1 < H && x0 >= 0 && y1 >= 0) ? (x1 - ux) * (uy - y0) : 0.0;
    const double wc = (x1 < W && y0 < H && x1 >= 0 && y0 >= 0) ? (ux - x0) * (y1 - uy) : 0.0;
    const double wd = (x1 < W && y1 < H && x1 >= 0 && y1 >= 0) ? (ux - x0) * (uy - y0) : 0.0;
    x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
    if (wa != 0.0) atomicAdd(&img[(size_t)y0 * W + x0], val * wa);
    if (wb != 0.0) atomicAdd(&img[(size_t)y1 * W + x0], val * wb);
    if (wc != 0.0) atomicAdd(&img[(size_t)y0 * W + x1], val * wc);
    if (wd != 0.0) atomicAdd(&img[(size_t)y1 * W + x1], val * wd);
}
__global__ void k_vote(const uint16_t* __restrict__ ex, const uint16_t* __restrict__ ey, const uint8_t* __restrict__ pol,
                       const float* __restrict__ mapx, const float* __restrict__ mapy, int n, int H, int W, int use_exp,
                       double* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vote_one(ex[i], ey[i], pol[i], i, n, mapx, mapy, H, W, use_exp, img);
}
// the same from an array of structs (x, y: uint16 fields, polarity: one byte, non-zero = positive)
__global__ void k_vote_aos(const uint8_t* __restrict__ ev, int stride, int ox, int oy, int op, const float* __restrict__ mapx,
                           const float* __restrict__ mapy, int n, int H, int W, int use_exp, double* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* e = ev + (size_t)i * stride;
    vote_one(*reinterpret_cast<const uint16_t*>(e + ox), *reinterpret_cast<const uint16_t*>(e + oy), e[op] != 0, i, n, mapx, mapy, H, W, use_exp, img);
}
// many slices at once: blockIdx.y = slice, its events are [offsets[y], offsets[y + 1]) of the concatenated arrays, its image img + y * H * W
__global__ void k_vote_batch(const uint16_t* __restrict__ ex, const uint16_t* __restrict__ ey, const uint8_t* __restrict__ pol,
                             const int* __restrict__ offsets, const float* __restrict__ mapx, const float* __restrict__ mapy, int H, int W,
                             int use_exp, double* __restrict__ img) {
    const int o = offsets[blockIdx.y], n = offsets[blockIdx.y + 1] - o;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vote_one(ex[o + i], ey[o + i], pol[o + i], i, n, mapx, mapy, H, W, use_exp, img + (size_t)blockIdx.y * H * W);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// 3x3 Gaussian in ONE pass with the arithmetic of cv::sepFilter2D's two (row filter, then column filter): the three row-filtered
// values a pixel's column filter needs are formed on the fly, each exactly as the row pass would round it
__device__ __forceinline__ double blur_row_at(const double* __restrict__ row, int c, int W, double k0, double k1) {
    return k0 * row[reflect101(c - 1, W)] + k1 * row[c] + k0 * row[reflect101(c + 1, W)];
}
#define EDS_SUMSQ_WAYS 64
// (SUMSQ: the blurred image IS level 0 — it goes straight to the level plane and its sum of squares is accumulated here, one launch
// and one pass over the image less than blur -> k_levels; blockIdx.z: image of a batch, 0 otherwise)
// RPT rows per thread (batches: a column strip slides down RPT rows, every row filter is formed once instead of three times and a
// workgroup adds one partial sum of squares for RPT rows; a single image keeps one row per workgroup, 1 440 of them, to fill the chip)
template <bool SUMSQ, int RPT>
__global__ void k_blur3(const double* __restrict__ src, double* __restrict__ dst, int H, int W, double k0, double k1, double* __restrict__ sumsq) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r0 = blockIdx.y * RPT;
    src += (size_t)blockIdx.z * H * W; dst += (size_t)blockIdx.z * H * W;
    double s = 0.0;
    if (c < W) {
        double a = blur_row_at(src + (size_t)reflect101(r0 - 1, H) * W, c, W, k0, k1);
        double b = blur_row_at(src + (size_t)r0 * W, c, W, k0, k1);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = r0 + k;
            if (r >= H) break;
            const double d = blur_row_at(src + (size_t)reflect101(r + 1, H) * W, c, W, k0, k1);
            const double v = k0 * a + k1 * b + k0 * d;
            dst[(size_t)r * W + c] = v;
            s += v * v;
            a = b; b = d;
        }
    }
    if (SUMSQ) {                        // as k_levels
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        __shared__ double sh[4];
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&sumsq[blockIdx.z * EDS_SUMSQ_WAYS + ((blockIdx.y * gridDim.x + blockIdx.x) & (EDS_SUMSQ_WAYS - 1))],
                                        sh[0] + sh[1] + sh[2] + sh[3]);
    }
}
// cv::resize(src, dst, out_size, cv::INTER_CUBIC) as the reference WRITES it (EventFrame.cpp:345, KeyFrame.cpp:355): the fourth
// positional parameter of cv::resize is `fx`, not the interpolation, so the call runs with the default INTER_LINEAR — and OpenCV
// turns INTER_LINEAR into the 2x2 block average (INTER_AREA fast path) when both scales are exactly 2.  Published OpenCV behaviour
// (imgproc/resize.cpp), restated: source coordinate fx = float((dx + 0.5) * scale - 0.5), sx = floor(fx), fractional part and
// the two weights in fp32, clamped to the first / last pixel; horizontal pass then vertical pass, in fp64 for CV_64F images.
__device__ __forceinline__ void resize_coord(int d, double scale, int n_src, int* s0, int* s1, double* w0, double* w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= n_src - 1) { f = 0.0f; s = n_src - 1; }
    *s0 = s; *s1 = s + 1 < n_src ? s + 1 : n_src - 1;
    *w0 = (double)(1.0f - f); *w1 = (double)f;
}
__global__ void k_resize(const double* __restrict__ src, int sH, int sW, double* __restrict__ dst, int H, int W) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    if (sH == 2 * H && sW == 2 * W) {                // INTER_AREA fast path: sum of the block in raster order, times 1/4
        const double* p = src + (size_t)(2 * r) * sW + 2 * c;
        dst[(size_t)r * W + c] = (((0.0 + p[0]) + p[1]) + p[sW] + p[sW + 1]) * 0.25;
        return;
    }
    int x0, x1, y0, y1;
    double a0, a1, b0, b1;
    resize_coord(c, (double)sW / (double)W, sW, &x0, &x1, &a0, &a1);
    resize_coord(r, (double)sH / (double)H, sH, &y0, &y1, &b0, &b1);
    const double h0 = src[(size_t)y0 * sW + x0] * a0 + src[(size_t)y0 * sW + x1] * a1;
    const double h1 = src[(size_t)y1 * sW + x0] * a0 + src[(size_t)y1 * sW + x1] * a1;
    dst[(size_t)r * W + c] = h0 * b0 + h1 * b1;
}
// Levels of EventFrame::create from ONE brightness image (EventFrame.cpp:348-357): level 0 is the image, level i >= 1 its
// dilation + erosion with a (2i+1)^2 box; pixels outside the image are ignored (cv::morphologyDefaultBorderValue).  blockIdx.z
// selects the level (level0 + z); each level's image goes to its own plane and its sum of squares is accumulated on the way.
// (batch != 0: blockIdx.z selects one of many IMAGES instead — src + z * H * W — all at level `level0`)
__global__ void k_levels(const double* __restrict__ src, double* __restrict__ planes, double* __restrict__ sumsq, int H, int W, int level0,
                         double* __restrict__ clear, int batch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, rad = level0 + (batch ? 0 : (int)blockIdx.z);
    if (batch) src += (size_t)blockIdx.z * H * W;
    double v = 0.0;
    if (clear && blockIdx.z == 0 && c < W) clear[(size_t)r * W + c] = 0.0;       // the vote image of the NEXT call (not `src`: the blur moved on)
    if (c < W) {
        if (rad == 0) {
            v = src[(size_t)r * W + c];
        } else {
            double mx = -1.7976931348623157e308, mn = 1.7976931348623157e308;
            for (int dr = -rad; dr <= rad; ++dr) {
                const int rr = r + dr;
                if (rr < 0 || rr >= H) continue;
                for (int dc = -rad; dc <= rad; ++dc) {
                    const int cc = c + dc;
                    if (cc < 0 || cc >= W) continue;
                    const double t = src[(size_t)rr * W + cc];
                    mx = t > mx ? t : mx;
                    mn = t < mn ? t : mn;
                }
            }
            v = mx + mn;
        }
        planes[(size_t)blockIdx.z * H * W + (size_t)r * W + c] = v;
    }
    double s = v * v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    // one atomic per workgroup, spread over EDS_SUMSQ_WAYS accumulators per level: 1 440 workgroups adding to ONE fp64 address
    // serialise (17 us for a VGA level, measured); k_store_levels adds the accumulators up
    if (threadIdx.x == 0) atomicAdd(&sumsq[blockIdx.z * EDS_SUMSQ_WAYS + ((blockIdx.y * gridDim.x + blockIdx.x) & (EDS_SUMSQ_WAYS - 1))],
                                    sh[0] + sh[1] + sh[2] + sh[3]);
}
// level / ||level||_F -> fp32 in the handle's layout, one slot per level (padding and margin filled with the nearest border pixel)
__global__ void k_store_levels(const double* __restrict__ planes, const double* __restrict__ sumsq, float* __restrict__ frames, int first_slot,
                               int H, int W, int Hp, int Wp, int tiled, int normalise, double* __restrict__ sumsq_next,
                               double* __restrict__ total_out, double* __restrict__ clear) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x) - EDS_FRAME_MARGIN, r = (int)blockIdx.y - EDS_FRAME_MARGIN;
    if (clear && blockIdx.z == 0 && r >= 0 && r < H && c >= 0 && c < W) clear[(size_t)r * W + c] = 0.0;   // the vote image of the NEXT call
    if (sumsq_next && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)     // ALL of the other accumulator set, for the next call
        for (int k = threadIdx.x; k < EDS_MAX_LEVELS * EDS_SUMSQ_WAYS; k += blockDim.x) sumsq_next[k] = 0.0;      // (which may build more levels)
    if (c >= Wp - EDS_FRAME_MARGIN) return;
    double ss = 0.0;
    for (int k = 0; k < EDS_SUMSQ_WAYS; ++k) ss += sumsq[blockIdx.z * EDS_SUMSQ_WAYS + k];      // wave-uniform, L2-resident
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) total_out[blockIdx.z] = ss;     // ||level||^2 to the host (mapped)
    const double inv = normalise ? 1.0 / sqrt(ss) : 1.0;         // PhotometricErrorNC wants the raw frame (EventFrame.cpp:278-281)
    const double v = planes[(size_t)blockIdx.z * H * W + (size_t)min(max(r, 0), H - 1) * W + min(max(c, 0), W - 1)] * inv;
    frames[(size_t)(first_slot + blockIdx.z) * Hp * Wp + eds_frame_index(r, c, Wp, tiled)] = (float)v;
}

// The same for the images of a batch in the tiled layout: one thread per (tile, row of the tile) — a 16-byte store, a quad of lanes
// writes one whole 64-byte tile, a wavefront 1 KB of consecutive tiles — and a workgroup walks many tile groups, so the accumulators
// are added up once per workgroup instead of once per 64 pixels (k_store_levels: 64 loads and adds, a square root and a division in
// front of ONE pixel per lane).  Same arithmetic per pixel.
__global__ void k_store_tiles_batch(const double* __restrict__ planes, const double* __restrict__ sumsq, float* __restrict__ frames, int first_slot,
                                    int H, int W, int Hp, int Wp, int normalise, double* __restrict__ total_out) {
    const int TW = Wp >> 2, ntiles = (Hp >> 2) * TW;
    double ss = 0.0;
    for (int k = 0; k < EDS_SUMSQ_WAYS; ++k) ss += sumsq[blockIdx.z * EDS_SUMSQ_WAYS + k];      // (k_store_levels' order)
    if (blockIdx.x == 0 && threadIdx.x == 0) total_out[blockIdx.z] = ss;
    const double inv = normalise ? 1.0 / sqrt(ss) : 1.0;
    const double* __restrict__ pl = planes + (size_t)blockIdx.z * H * W;
    float* __restrict__ fr = frames + (size_t)(first_slot + blockIdx.z) * Hp * Wp;
    const int j = threadIdx.x & 3, per = blockDim.x >> 2;
    for (int t = blockIdx.x * per + (threadIdx.x >> 2); t < ntiles; t += gridDim.x * per) {
        const int ty = t / TW, tx = t - ty * TW;
        const int r = min(max(ty * 4 + j - EDS_FRAME_MARGIN, 0), H - 1), c0 = tx * 4 - EDS_FRAME_MARGIN;
        const double* __restrict__ row = pl + (size_t)r * W;
        float4 o;
        o.x = (float)(row[min(max(c0, 0), W - 1)] * inv);
        o.y = (float)(row[min(max(c0 + 1, 0), W - 1)] * inv);
        o.z = (float)(row[min(max(c0 + 2, 0), W - 1)] * inv);
        o.w = (float)(row[min(max(c0 + 3, 0), W - 1)] * inv);
        *reinterpret_cast<float4*>(fr + (size_t)t * 16 + j * 4) = o;
    }
}

// row-major H x W fp32 image -> the handle's frame layout (tiles, padding and margin filled with the nearest border pixel): the
// destination rows [r_lo, r_lo + gridDim.y), logical coordinates.  `src` may be device-mapped pinned host memory (set_event_frame
// reads its staging buffer over PCIe, a band of rows per launch, while the host narrows the next band).
__global__ void k_store_rowmajor(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Hp, int Wp, int tiled, int r_lo) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x) - EDS_FRAME_MARGIN, r = (int)blockIdx.y + r_lo;
    if (c >= Wp - EDS_FRAME_MARGIN) return;
    dst[eds_frame_index(r, c, Wp, tiled)] = src[(size_t)min(max(r, 0), H - 1) * W + min(max(c, 0), W - 1)];
}

// The same for a frame the host is STILL NARROWING (round 3): one launch per frame instead of one per band.  Workgroup (band, part)
// waits until the host has published "rows [0, n) of upload `seq` are in the staging buffer" in a pinned word, then moves its part
// of the band.  A workgroup waits for the host only, never for another workgroup: no residency assumption.  The wait is
// bounded (EDS_FOLLOW_TIMEOUT_TICKS of the 100 MHz clock: a host that stops mid-frame for seconds); a workgroup that gives up
// marks prog[1] and the next wait on the handle reports it.
#define EDS_FOLLOW_PARTS 4
#define EDS_FOLLOW_THREADS 1024
#define EDS_FOLLOW_TIMEOUT_TICKS 200000000ull
__global__ __launch_bounds__(EDS_FOLLOW_THREADS) void k_store_follow(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Hp, int Wp,
                                                                  int tiled, unsigned* prog, unsigned seq, int rows_per, int nbands) {
    const int band = blockIdx.x / EDS_FOLLOW_PARTS, part = blockIdx.x - band * EDS_FOLLOW_PARTS;
    const int rb = rows_per * band, re = min(H, rows_per * (band + 1));
    const int lo = band == 0 ? -EDS_FRAME_MARGIN : rb, hi = band == nbands - 1 ? Hp - EDS_FRAME_MARGIN : re;
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        for (;;) {
            const unsigned v = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((v >> 20) == seq && (int)(v & 0xfffffu) >= re) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > EDS_FOLLOW_TIMEOUT_TICKS) {
                ok = 0;
                __hip_atomic_store(prog + 1, 0x80000000u | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            // (every poll is a PCIe read of the host's cache line: workgroups whose band is far away ask rarely)
            if ((v >> 20) == seq && (int)(v & 0xfffffu) + rows_per < rb) __builtin_amdgcn_s_sleep(100); else __builtin_amdgcn_s_sleep(6);
        }
        // the staging rows the host wrote before it published `re` must be the ones read below: acquire at system scope (invalidates this
        // CU's vector cache, which every wavefront of the workgroup shares) — the buffer is re-used for every frame, so "first touched
        // here" was an argument about THIS upload only
        if (ok) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    // four consecutive columns per lane: one 16-byte read of the staging buffer (plain: the lines of this band are first touched here,
    // after the host has written them — the caller cuts the bands at multiples of 32 rows, so no 128-byte line of the staging buffer
    // belongs to two bands) and one 16-byte write of a tile row
    const int Q = Wp >> 2, n = (hi - lo) * Q;
    const bool vec = (W & 3) == 0;
    constexpr int U = 4, STEP = EDS_FOLLOW_PARTS * EDS_FOLLOW_THREADS;
    for (int e0 = part * EDS_FOLLOW_THREADS + (int)threadIdx.x; e0 < n; e0 += U * STEP) {
        float4 v[U];
        int rr_[U], c_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                        // U reads over PCIe in flight per lane
            const int e = e0 + u * STEP;
            rr_[u] = INT_MIN;
            if (e < n) {
                const int rq = e / Q, r = lo + rq, c = 4 * (e - rq * Q) - EDS_FRAME_MARGIN;
                rr_[u] = r; c_[u] = c;
                const float* row = src + (size_t)min(max(r, 0), H - 1) * W;
                if (vec && c >= 0 && c + 3 < W) v[u] = *reinterpret_cast<const float4*>(row + c);
                else v[u] = make_float4(row[min(max(c, 0), W - 1)], row[min(max(c + 1, 0), W - 1)], row[min(max(c + 2, 0), W - 1)], row[min(max(c + 3, 0), W - 1)]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rr_[u] == INT_MIN) continue;
            float* d = dst + eds_frame_index(rr_[u], c_[u], Wp, tiled);      // c is a multiple of 4: the four pixels are contiguous in either layout
            *reinterpret_cast<float4*>(d) = v[u];
        }
    }
}

__global__ void k_mirror_rows(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += gridDim.x * blockDim.x) d4[i] = s4[i];
}

}  // namespace

bool eds_mirror_residuals(eds_trk* h, int first, int count) {
    if (!h->d_rmap || first + count > EDS_RHOST_SLOTS) return false;
    const int n = count * h->Np;                                  // Np is a multiple of 4 (EDS_POINT_ALIGN)
    hipLaunchKernelGGL(k_mirror_rows, dim3((n / 4 + 255) / 256), dim3(256), 0, h->st, h->dr + (size_t)first * h->Np, h->d_rmap + (size_t)first * h->Np, n);
    return true;
}

// source rows [row_b, row_e) (and, with them, the margin / padding rows that replicate row 0 or row H - 1)
void eds_frame_store_rowmajor(eds_trk* h, int slot, const float* d_src, int row_b, int row_e) {
    const int lo = row_b <= 0 ? -EDS_FRAME_MARGIN : row_b, hi = row_e >= h->H ? h->Hp - EDS_FRAME_MARGIN : row_e;
    if (hi <= lo) return;
    hipLaunchKernelGGL(k_store_rowmajor, dim3((h->Wp + 255) / 256, hi - lo), dim3(256), 0, h->st, d_src,
                       h->dframe + (size_t)slot * h->Hp * h->Wp, h->H, h->W, h->Hp, h->Wp, h->tiled, lo);
}

// A WHOLE frame, 16 bytes per lane (the batch upload, eds_trk_set_event_frames): thread (tile column, row) moves the four pixels of one
// tile row — one float4 read from the row-major source (pinned host memory read over PCIe: the wider the requests, the fewer of
// them are in flight per byte), one float4 store into the tile.  Pieces that touch the replicated margin or the padding go pixel by
// pixel through the clamp.  Tiled frames with W a multiple of 4 only (the caller falls back to k_store_rowmajor otherwise).
__global__ __launch_bounds__(256) void k_store_frame4(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Hp, int Wp) {
    const int tc = (int)(blockIdx.x * blockDim.x + threadIdx.x), r = (int)blockIdx.y - EDS_FRAME_MARGIN;
    if (tc >= (Wp >> 2)) return;
    const int c = 4 * tc - EDS_FRAME_MARGIN, rs = min(max(r, 0), H - 1);
    float4 v;
    if (c >= 0 && c + 3 < W) {
        v = *reinterpret_cast<const float4*>(src + (size_t)rs * W + c);
    } else {
        const float* row = src + (size_t)rs * W;
        v = make_float4(row[min(max(c, 0), W - 1)], row[min(max(c + 1, 0), W - 1)], row[min(max(c + 2, 0), W - 1)], row[min(max(c + 3, 0), W - 1)]);
    }
    *reinterpret_cast<float4*>(dst + eds_frame_index(r, c, Wp, 1)) = v;
}
void eds_frame_store_whole(eds_trk* h, int slot, const float* d_src, hipStream_t st) {
    float* dst = h->dframe + (size_t)slot * h->Hp * h->Wp;
    if (h->tiled && (h->W & 3) == 0)
        hipLaunchKernelGGL(k_store_frame4, dim3(((h->Wp >> 2) + 255) / 256, h->Hp), dim3(256), 0, st, d_src, dst, h->H, h->W, h->Hp, h->Wp);
    else
        hipLaunchKernelGGL(k_store_rowmajor, dim3((h->Wp + 255) / 256, h->Hp), dim3(256), 0, st, d_src, dst, h->H, h->W, h->Hp, h->Wp, h->tiled, -EDS_FRAME_MARGIN);
}

// One launch that follows the host through the staging buffer (k_store_follow); the caller publishes its progress in h->h_fprog[0].
void eds_frame_store_follow(eds_trk* h, int slot, unsigned seq, int rows_per) {
    const int nbands = (h->H + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(k_store_follow, dim3(nbands * EDS_FOLLOW_PARTS), dim3(EDS_FOLLOW_THREADS), 0, h->st, h->d_fstage,
                       h->dframe + (size_t)slot * h->Hp * h->Wp, h->H, h->W, h->Hp, h->Wp, h->tiled, h->d_fprog, seq, rows_per, nbands);
}

void eds_frame_free(EdsFrameBuffers* fb) {
    void* d[] = {fb->d_mapx, fb->d_mapy, fb->d_img, fb->d_tmp, fb->d_norm, fb->d_planes};
    for (void* p : d) if (p) hipFree(p);
    if (fb->h_events) hipHostFree(fb->h_events);       // d_ex, d_ey, d_pol are its device view
    if (fb->h_norm_out) hipHostFree(fb->h_norm_out);
    if (fb->h_aos) hipHostFree(fb->h_aos);
    void* bd[] = {fb->b_img, fb->b_tmp, fb->b_planes, fb->b_norm};
    for (void* q : bd) if (q) hipFree(q);
    if (fb->h_bmeta) hipHostFree(fb->h_bmeta);
    *fb = EdsFrameBuffers();
}

int eds_frame_set_map(eds_trk* h, const float* mapx, const float* mapy, int mH, int mW) {
    EdsFrameBuffers& fb = h->frame_build;
    if (!mapx || !mapy) {               // identity LUT
        if (fb.d_mapx) { hipFree(fb.d_mapx); hipFree(fb.d_mapy); fb.d_mapx = fb.d_mapy = nullptr; }
        fb.map_H = fb.map_W = 0;
        return EDS_OK;
    }
    const size_t n = (size_t)mH * mW;
    if (fb.d_mapx && (fb.map_H != mH || fb.map_W != mW)) { hipFree(fb.d_mapx); hipFree(fb.d_mapy); fb.d_mapx = fb.d_mapy = nullptr; }
    if (!fb.d_mapx) {
        if (hipMalloc((void**)&fb.d_mapx, n * 4) != hipSuccess || hipMalloc((void**)&fb.d_mapy, n * 4) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(undistortion map)");
    }
    fb.map_H = mH; fb.map_W = mW;
    if (hipMemcpy(fb.d_mapx, mapx, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(fb.d_mapy, mapy, n * 4, hipMemcpyHostToDevice) != hipSuccess)
        return eds_internal_fail(EDS_ERR_HIP, "hipMemcpy(undistortion map)");
    return EDS_OK;
}

// Levels level0 .. level0 + nlevels - 1 of EventFrame::create into slots first_slot .. first_slot + nlevels - 1 from ONE vote:
// events -> brightness image at the sensor's size (sH x sW) -> 3x3 Gaussian -> resize to the handle's H x W when they differ
// (out_scale != 1) -> every level + its Frobenius norm in one launch -> normalise + store every level in one launch.
int eds_frame_build_levels(eds_trk* h, int first_slot, int level0, int nlevels, int n_events, const uint16_t* ex, const uint16_t* ey,
                           const uint8_t* pol, int sH, int sW, double blur_sigma, int use_exp_weights, double* norms_out, const EdsEventAos* aos) {
    EdsFrameBuffers& fb = h->frame_build;
    const int H = h->H, W = h->W;
    const size_t n = (size_t)H * W, ns = (size_t)sH * sW;
    if (fb.d_mapx && (fb.map_H != sH || fb.map_W != sW)) return eds_internal_fail(EDS_ERR_INVALID, "undistortion map and sensor size differ");
    if (fb.img_elems < std::max(n, ns)) {
        if (fb.d_img) { hipFree(fb.d_img); hipFree(fb.d_tmp); fb.d_img = fb.d_tmp = nullptr; }
        fb.img_elems = std::max(n, ns);
        fb.img_clean = false;
        if (hipMalloc((void**)&fb.d_img, fb.img_elems * 8) != hipSuccess || hipMalloc((void**)&fb.d_tmp, fb.img_elems * 8) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(frame accumulation)");
    }
    constexpr size_t NORM_SET = (size_t)EDS_MAX_LEVELS * EDS_SUMSQ_WAYS;
    if (!fb.d_norm) {
        if (hipMalloc((void**)&fb.d_norm, 8 * 2 * NORM_SET) != hipSuccess || hipMemset(fb.d_norm, 0, 8 * 2 * NORM_SET) != hipSuccess ||
            hipHostMalloc((void**)&fb.h_norm_out, 8 * EDS_MAX_LEVELS, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&fb.d_norm_out, fb.h_norm_out, 0) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(norms)");
    }
    if (fb.plane_levels < nlevels) {
        if (fb.d_planes) hipFree(fb.d_planes);
        fb.d_planes = nullptr; fb.plane_levels = 0;
        if (hipMalloc((void**)&fb.d_planes, n * 8 * nlevels) != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(level planes)");
        fb.plane_levels = nlevels;
    }
    if (aos) {
        const size_t bytes = (size_t)n_events * aos->stride;
        if (bytes > fb.cap_aos) {
            if (fb.h_aos) hipHostFree(fb.h_aos);
            fb.h_aos = fb.d_aos = nullptr;
            fb.cap_aos = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void**)&fb.h_aos, fb.cap_aos, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&fb.d_aos, fb.h_aos, 0) != hipSuccess) {
                fb.cap_aos = 0;
                return eds_internal_fail(EDS_ERR_HIP, "hipHostMalloc(events)");
            }
        }
    } else
    if (n_events > fb.cap_events) {
        if (fb.h_events) hipHostFree(fb.h_events);
        fb.d_ex = fb.d_ey = nullptr; fb.d_pol = nullptr; fb.h_events = nullptr;
        fb.cap_events = (n_events + n_events / 4 + 1024 + 7) & ~7;
        uint8_t* dev = nullptr;
        if (hipHostMalloc((void**)&fb.h_events, (size_t)fb.cap_events * 5, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&dev, fb.h_events, 0) != hipSuccess) {
            fb.cap_events = 0;
            return eds_internal_fail(EDS_ERR_HIP, "hipHostMalloc(events)");
        }
        fb.d_ex = reinterpret_cast<uint16_t*>(dev);
        fb.d_ey = fb.d_ex + fb.cap_events;
        fb.d_pol = reinterpret_cast<uint8_t*>(fb.d_ey + fb.cap_events);
    }
    hipStream_t st = h->st;
    hipError_t e = hipSuccess;
    if (n_events > 0 && aos) {
        std::memcpy(fb.h_aos, aos->data, (size_t)n_events * aos->stride);     // as they are: k_vote_aos picks the fields
    } else if (n_events > 0) {          // pack into the pinned staging; k_vote reads it over PCIe (5 bytes per event)
        const size_t cap = (size_t)fb.cap_events;
        std::memcpy(fb.h_events, ex, (size_t)n_events * 2);
        std::memcpy(fb.h_events + cap * 2, ey, (size_t)n_events * 2);
        std::memcpy(fb.h_events + cap * 4, pol, (size_t)n_events);
    }
    if (!fb.img_clean) e = hipMemsetAsync(fb.d_img, 0, fb.img_elems * 8, st);     // first call, or the last one could not clear it
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.img_clean = false;
    double* norm_cur = fb.d_norm + (size_t)(fb.calls & 1) * NORM_SET;
    double* norm_next = fb.d_norm + (size_t)((fb.calls + 1) & 1) * NORM_SET;
    ++fb.calls;
    if (n_events > 0 && aos)
        hipLaunchKernelGGL(k_vote_aos, dim3((n_events + 255) / 256), dim3(256), 0, st, fb.d_aos, aos->stride, aos->off_x, aos->off_y, aos->off_pol,
                           fb.d_mapx, fb.d_mapy, n_events, sH, sW, use_exp_weights, fb.d_img);
    else if (n_events > 0)
        hipLaunchKernelGGL(k_vote, dim3((n_events + 255) / 256), dim3(256), 0, st, fb.d_ex, fb.d_ey, fb.d_pol, fb.d_mapx, fb.d_mapy,
                           n_events, sH, sW, use_exp_weights, fb.d_img);
    const dim3 b2(256);
    double* cur = fb.d_img;
    double* other = fb.d_tmp;
    // one plain level at the sensor's size: the blurred image is the level — blur, plane and sum of squares in one launch
    const bool fused0 = blur_sigma > 0.0 && nlevels == 1 && level0 == 0 && sH == H && sW == W;
    if (blur_sigma > 0.0) {             // cv::getGaussianKernel(3, sigma): exp(-x^2 / (2 sigma^2)), normalised
        const double t = std::exp(-0.5 / (blur_sigma * blur_sigma)), s = 1.0 + 2.0 * t;
        if (fused0) hipLaunchKernelGGL((k_blur3<true, 1>), dim3((sW + 255) / 256, sH), b2, 0, st, cur, fb.d_planes, sH, sW, t / s, 1.0 / s, norm_cur);
        else hipLaunchKernelGGL((k_blur3<false, 1>), dim3((sW + 255) / 256, sH), b2, 0, st, cur, other, sH, sW, t / s, 1.0 / s, (double*)nullptr);
        std::swap(cur, other);
    }
    if (sH != H || sW != W) {           // out_scale != 1 (EventFrame.cpp:342-346)
        hipLaunchKernelGGL(k_resize, dim3((W + 255) / 256, H), b2, 0, st, cur, sH, sW, other, H, W);
        std::swap(cur, other);
    }
    // the vote image can be cleared for the next call once the blur has read it (by k_levels, or by the store when k_levels is not
    // run), provided it covers exactly the frame (sensor size == frame size)
    double* clear = (blur_sigma > 0.0 && ns == n && fb.img_elems == n) ? fb.d_img : nullptr;
    if (!fused0) hipLaunchKernelGGL(k_levels, dim3((W + 255) / 256, H, nlevels), b2, 0, st, cur, fb.d_planes, norm_cur, H, W, level0, clear, 0);
    hipLaunchKernelGGL(k_store_levels, dim3((h->Wp + 255) / 256, h->Hp, nlevels), b2, 0, st, fb.d_planes, norm_cur, h->dframe, first_slot, H, W,
                       h->Hp, h->Wp, h->tiled, h->cfg.nc ? 0 : 1, norm_next, fb.d_norm_out, fused0 ? clear : (double*)nullptr);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    fb.img_clean = clear != nullptr;
    for (int i = 0; i < nlevels; ++i) {
        if (norms_out) norms_out[i] = std::sqrt(fb.h_norm_out[i]);     // the accumulators added up in k_store_levels' order
        h->slots[first_slot + i].has_frame = true;
        ++h->slots[first_slot + i].frame_version;
    }
    return EDS_OK;
}

// `count` independent event slices (one per alignment of a batch: BASELINE.json configs[4] has 64 (keyframe, event frame) pairs) into
// slots first_slot .. first_slot + count - 1, level `level` each, in chunks of up to EDS_FRAME_BATCH images per pass: one vote launch
// over all slices of a chunk, one blur, one level + sum-of-squares launch, one normalise + tile launch (blockIdx.z = image).  Same
// arithmetic per image as eds_frame_build_levels.
#define EDS_FRAME_BATCH 32
int eds_frame_build_batch(eds_trk* h, int first_slot, int count, const int* offsets, const uint16_t* ex, const uint16_t* ey, const uint8_t* pol,
                          int level, double blur_sigma, int use_exp_weights, double* norms_out) {
    EdsFrameBuffers& fb = h->frame_build;
    const int H = h->H, W = h->W;
    const size_t n = (size_t)H * W;
    if (fb.d_mapx && (fb.map_H != H || fb.map_W != W)) return eds_internal_fail(EDS_ERR_INVALID, "undistortion map and frame size differ");
    const int C = std::min(count, EDS_FRAME_BATCH);
    if (fb.batch_cap < C) {
        void* d[] = {fb.b_img, fb.b_tmp, fb.b_planes, fb.b_norm};
        for (void* q : d) if (q) hipFree(q);
        fb.b_img = fb.b_tmp = fb.b_planes = fb.b_norm = nullptr; fb.batch_cap = 0;
        if (hipMalloc((void**)&fb.b_img, C * n * 8) != hipSuccess || hipMalloc((void**)&fb.b_tmp, C * n * 8) != hipSuccess ||
            hipMalloc((void**)&fb.b_planes, C * n * 8) != hipSuccess || hipMalloc((void**)&fb.b_norm, (size_t)C * EDS_SUMSQ_WAYS * 8) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "allocation of the batched event-frame buffers failed");
        fb.batch_cap = C;
    }
    // A GROUP of chunks is queued without waiting in between: events of chunk k + 1 are copied into the pinned staging while the device
    // works on chunk k (every chunk has its own part of the staging, of the offsets and of the totals; the image buffers are reused
    // in stream order).  Groups are bounded by EDS_FRAME_GROUP_EVENTS so that the pinned staging stays small.
    constexpr long long EDS_FRAME_GROUP_EVENTS = 8ll << 20;
    hipStream_t st = h->st;
    const dim3 b2(256);
    int g0 = 0;
    while (g0 < count) {
        int g1 = g0;                                                // group = slices [g0, g1): whole chunks, at least one
        long long gev = 0;
        while (g1 < count) {
            const int c1 = std::min(count, g1 + fb.batch_cap);
            const long long ev = (long long)offsets[c1] - offsets[g1];
            if (ev < 0) return eds_internal_fail(EDS_ERR_INVALID, "offsets must not decrease");
            if (g1 > g0 && gev + ev > EDS_FRAME_GROUP_EVENTS) break;
            gev += ev; g1 = c1;
        }
        const int gn = g1 - g0;
        if (fb.meta_cap < gn) {
            if (fb.h_bmeta) hipHostFree(fb.h_bmeta);
            fb.h_bmeta = nullptr; fb.meta_cap = 0;
            char* dmeta = nullptr;
            const int cap = std::max(gn, 64);
            if (hipHostMalloc((void**)&fb.h_bmeta, (size_t)cap * 8 + (size_t)(cap + 1) * 4, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&dmeta, fb.h_bmeta, 0) != hipSuccess)
                return eds_internal_fail(EDS_ERR_HIP, "allocation of the batched event-frame buffers failed");
            fb.d_bmeta = dmeta; fb.meta_cap = cap;
        }
        double* h_tot = reinterpret_cast<double*>(fb.h_bmeta);                       // [meta_cap] totals out
        int* h_off = reinterpret_cast<int*>(fb.h_bmeta + (size_t)fb.meta_cap * 8);     // [meta_cap + 1] event offsets relative to the group
        double* d_tot = reinterpret_cast<double*>(fb.d_bmeta);
        int* d_off = reinterpret_cast<int*>(fb.d_bmeta + (size_t)fb.meta_cap * 8);
        if (gev > fb.cap_events) {           // the mapped staging of the single-slice builder, grown
            if (fb.h_events) hipHostFree(fb.h_events);
            fb.h_events = nullptr; fb.d_ex = fb.d_ey = nullptr; fb.d_pol = nullptr;
            fb.cap_events = (int)((gev + gev / 4 + 1024 + 7) & ~7ll);
            uint8_t* dev = nullptr;
            if (hipHostMalloc((void**)&fb.h_events, (size_t)fb.cap_events * 5, hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&dev, fb.h_events, 0) != hipSuccess) { fb.cap_events = 0; return eds_internal_fail(EDS_ERR_HIP, "hipHostMalloc(events)"); }
            fb.d_ex = reinterpret_cast<uint16_t*>(dev);
            fb.d_ey = fb.d_ex + fb.cap_events;
            fb.d_pol = reinterpret_cast<uint8_t*>(fb.d_ey + fb.cap_events);
        }
        const int ge0 = offsets[g0];
        for (int b = 0; b <= gn; ++b) h_off[b] = offsets[g0 + b] - ge0;
        for (int b = 0; b < gn; ++b) if (h_off[b + 1] < h_off[b]) return eds_internal_fail(EDS_ERR_INVALID, "offsets must not decrease");
        const size_t cap = (size_t)fb.cap_events;
        for (int c0 = 0; c0 < gn; c0 += fb.batch_cap) {             // chunks of the group
            const int cn = std::min(fb.batch_cap, gn - c0);
            const int e0 = h_off[c0], ne = h_off[c0 + cn] - e0;
            int maxn = 0;
            for (int b = 0; b < cn; ++b) maxn = std::max(maxn, h_off[c0 + b + 1] - h_off[c0 + b]);
            if (ne > 0) {
                std::memcpy(fb.h_events + (size_t)e0 * 2, ex + ge0 + e0, (size_t)ne * 2);
                std::memcpy(fb.h_events + cap * 2 + (size_t)e0 * 2, ey + ge0 + e0, (size_t)ne * 2);
                std::memcpy(fb.h_events + cap * 4 + (size_t)e0, pol + ge0 + e0, (size_t)ne);
            }
            hipError_t e = hipMemsetAsync(fb.b_img, 0, (size_t)cn * n * 8, st);
            if (e == hipSuccess) e = hipMemsetAsync(fb.b_norm, 0, (size_t)cn * EDS_SUMSQ_WAYS * 8, st);
            if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
            if (maxn > 0)
                hipLaunchKernelGGL(k_vote_batch, dim3((maxn + 255) / 256, cn), dim3(256), 0, st, fb.d_ex, fb.d_ey, fb.d_pol, d_off + c0, fb.d_mapx,
                                   fb.d_mapy, H, W, use_exp_weights, fb.b_img);
            double* cur = fb.b_img;
            const bool fused0 = blur_sigma > 0.0 && level == 0;       // the blurred image is the level: blur + plane + sum of squares in one launch
            if (blur_sigma > 0.0) {
                const double t = std::exp(-0.5 / (blur_sigma * blur_sigma)), sk = 1.0 + 2.0 * t;
                if (fused0) hipLaunchKernelGGL((k_blur3<true, 4>), dim3((W + 255) / 256, (H + 3) / 4, cn), b2, 0, st, cur, fb.b_planes, H, W, t / sk, 1.0 / sk, fb.b_norm);
                else hipLaunchKernelGGL((k_blur3<false, 4>), dim3((W + 255) / 256, (H + 3) / 4, cn), b2, 0, st, cur, fb.b_tmp, H, W, t / sk, 1.0 / sk, (double*)nullptr);
                cur = fb.b_tmp;
            }
            if (!fused0) hipLaunchKernelGGL(k_levels, dim3((W + 255) / 256, H, cn), b2, 0, st, cur, fb.b_planes, fb.b_norm, H, W, level, (double*)nullptr, 1);
            if (h->tiled) {
                const int ntiles = (h->Hp >> 2) * (h->Wp >> 2);
                hipLaunchKernelGGL(k_store_tiles_batch, dim3(std::min((ntiles + 63) / 64, 64), 1, cn), b2, 0, st, fb.b_planes, fb.b_norm, h->dframe,
                                   first_slot + g0 + c0, H, W, h->Hp, h->Wp, h->cfg.nc ? 0 : 1, d_tot + c0);
            } else {
                hipLaunchKernelGGL(k_store_levels, dim3((h->Wp + 255) / 256, h->Hp, cn), b2, 0, st, fb.b_planes, fb.b_norm, h->dframe, first_slot + g0 + c0, H, W,
                                   h->Hp, h->Wp, h->tiled, h->cfg.nc ? 0 : 1, (double*)nullptr, d_tot + c0, (double*)nullptr);
            }
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
        for (int b = 0; b < gn; ++b) {
            if (norms_out) norms_out[g0 + b] = std::sqrt(h_tot[b]);
            h->slots[first_slot + g0 + b].has_frame = true;
            ++h->slots[first_slot + g0 + b].frame_version;
        }
        g0 = g1;
    }
    return EDS_OK;
}
