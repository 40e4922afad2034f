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

#include <cmath>
#include <cstring>

#include "eds_fused.hpp"
#include "eds_handle.hpp"

namespace {

__global__ void k_vote(const uint16_t* __restrict__ ex, const uint16_t* __restrict__ ey, const uint8_t* __restrict__ pol,
                       const float* __restrict__ mapx, const float* __restrict__ mapy, int n, int H, int W, int use_exp,
                       double* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = ex[i], py = ey[i];
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
    const double val = weight * (pol[i] ? 1.0 : -1.0);
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

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
// separable 3-tap Gaussian, row filter then column filter like cv::sepFilter2D
__global__ void k_blur_rows(const double* __restrict__ src, double* __restrict__ dst, int H, int W, double k0, double k1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const double* row = src + (size_t)r * W;
    dst[(size_t)r * W + c] = k0 * row[reflect101(c - 1, W)] + k1 * row[c] + k0 * row[reflect101(c + 1, W)];
}
__global__ void k_blur_cols(const double* __restrict__ src, double* __restrict__ dst, int H, int W, double k0, double k1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    dst[(size_t)r * W + c] = k0 * src[(size_t)reflect101(r - 1, H) * W + c] + k1 * src[(size_t)r * W + c] +
                             k0 * src[(size_t)reflect101(r + 1, H) * W + c];
}
// dilate + erode with a (2 rad + 1)^2 box; pixels outside the image are ignored (cv::morphologyDefaultBorderValue)
__global__ void k_morph(const double* __restrict__ src, double* __restrict__ dst, int H, int W, int rad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    double mx = -1.7976931348623157e308, mn = 1.7976931348623157e308;
    for (int dr = -rad; dr <= rad; ++dr) {
        const int rr = r + dr;
        if (rr < 0 || rr >= H) continue;
        for (int dc = -rad; dc <= rad; ++dc) {
            const int cc = c + dc;
            if (cc < 0 || cc >= W) continue;
            const double v = src[(size_t)rr * W + cc];
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
    }
    dst[(size_t)r * W + c] = mx + mn;
}
__global__ void k_sumsq(const double* __restrict__ src, size_t n, double* __restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[i] * src[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}
// frame / ||frame||_F -> fp32 in the handle's layout (padding and margin filled with the nearest border pixel)
__global__ void k_store(const double* __restrict__ src, const double* __restrict__ sumsq, float* __restrict__ dst, int H, int W,
                        int Hp, int Wp, int tiled, int normalise) {
    const int c = (int)(blockIdx.x * blockDim.x + threadIdx.x) - EDS_FRAME_MARGIN, r = (int)blockIdx.y - EDS_FRAME_MARGIN;
    if (c >= Wp - EDS_FRAME_MARGIN) return;
    const double inv = normalise ? 1.0 / sqrt(*sumsq) : 1.0;     // PhotometricErrorNC wants the raw frame (EventFrame.cpp:278-281)
    const double v = src[(size_t)min(max(r, 0), H - 1) * W + min(max(c, 0), W - 1)] * inv;
    dst[eds_frame_index(r, c, Wp, tiled)] = (float)v;
}

}  // namespace

void eds_frame_free(EdsFrameBuffers* fb) {
    void* d[] = {fb->d_mapx, fb->d_mapy, fb->d_img, fb->d_tmp, fb->d_norm, fb->d_ex};     // d_ey, d_pol are slices of d_ex
    for (void* p : d) if (p) hipFree(p);
    if (fb->h_events) hipHostFree(fb->h_events);
    *fb = EdsFrameBuffers();
}

int eds_frame_set_map(eds_trk* h, const float* mapx, const float* mapy) {
    EdsFrameBuffers& fb = h->frame_build;
    const size_t n = (size_t)h->H * h->W;
    if (!mapx || !mapy) {               // identity LUT
        if (fb.d_mapx) { hipFree(fb.d_mapx); hipFree(fb.d_mapy); fb.d_mapx = fb.d_mapy = nullptr; }
        return EDS_OK;
    }
    if (!fb.d_mapx) {
        if (hipMalloc((void**)&fb.d_mapx, n * 4) != hipSuccess || hipMalloc((void**)&fb.d_mapy, n * 4) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(undistortion map)");
    }
    if (hipMemcpy(fb.d_mapx, mapx, n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(fb.d_mapy, mapy, n * 4, hipMemcpyHostToDevice) != hipSuccess)
        return eds_internal_fail(EDS_ERR_HIP, "hipMemcpy(undistortion map)");
    return EDS_OK;
}

int eds_frame_build(eds_trk* h, int slot, int n_events, const uint16_t* ex, const uint16_t* ey, const uint8_t* pol, int level,
                    double blur_sigma, int use_exp_weights, double* norm_out) {
    EdsFrameBuffers& fb = h->frame_build;
    const int H = h->H, W = h->W;
    const size_t n = (size_t)H * W;
    if (!fb.d_img) {
        if (hipMalloc((void**)&fb.d_img, n * 8) != hipSuccess || hipMalloc((void**)&fb.d_tmp, n * 8) != hipSuccess ||
            hipMalloc((void**)&fb.d_norm, 16) != hipSuccess)
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(frame accumulation)");
    }
    if (n_events > fb.cap_events) {
        if (fb.d_ex) hipFree(fb.d_ex);
        if (fb.h_events) hipHostFree(fb.h_events);
        fb.d_ex = fb.d_ey = nullptr; fb.d_pol = nullptr; fb.h_events = nullptr;
        fb.cap_events = (n_events + n_events / 4 + 1024 + 7) & ~7;
        const size_t bytes = (size_t)fb.cap_events * 5;
        if (hipMalloc((void**)&fb.d_ex, bytes) != hipSuccess || hipHostMalloc((void**)&fb.h_events, bytes, hipHostMallocDefault) != hipSuccess) {
            fb.cap_events = 0;
            return eds_internal_fail(EDS_ERR_HIP, "hipMalloc(events)");
        }
        fb.d_ey = fb.d_ex + fb.cap_events;
        fb.d_pol = reinterpret_cast<uint8_t*>(fb.d_ey + fb.cap_events);
    }
    hipStream_t st = h->st;
    hipError_t e = hipSuccess;
    if (n_events > 0) {                 // pack into the pinned staging, one asynchronous copy
        const size_t cap = (size_t)fb.cap_events;
        std::memcpy(fb.h_events, ex, (size_t)n_events * 2);
        std::memcpy(fb.h_events + cap * 2, ey, (size_t)n_events * 2);
        std::memcpy(fb.h_events + cap * 4, pol, (size_t)n_events);
        if ((size_t)n_events * 8 >= cap * 5)    // nearly full: one copy of everything; else three tight ones
            e = hipMemcpyAsync(fb.d_ex, fb.h_events, cap * 5, hipMemcpyHostToDevice, st);
        else {
            e = hipMemcpyAsync(fb.d_ex, fb.h_events, (size_t)n_events * 2, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(fb.d_ey, fb.h_events + cap * 2, (size_t)n_events * 2, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(fb.d_pol, fb.h_events + cap * 4, (size_t)n_events, hipMemcpyHostToDevice, st);
        }
    }
    if (e == hipSuccess) e = hipMemsetAsync(fb.d_img, 0, n * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(fb.d_norm, 0, 16, st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    if (n_events > 0)
        hipLaunchKernelGGL(k_vote, dim3((n_events + 255) / 256), dim3(256), 0, st, fb.d_ex, fb.d_ey, fb.d_pol, fb.d_mapx, fb.d_mapy,
                           n_events, H, W, use_exp_weights, fb.d_img);
    const dim3 g2((W + 255) / 256, H), b2(256);
    double* cur = fb.d_img;
    double* other = fb.d_tmp;
    if (blur_sigma > 0.0) {             // cv::getGaussianKernel(3, sigma): exp(-x^2 / (2 sigma^2)), normalised
        const double t = std::exp(-0.5 / (blur_sigma * blur_sigma)), s = 1.0 + 2.0 * t;
        hipLaunchKernelGGL(k_blur_rows, g2, b2, 0, st, cur, other, H, W, t / s, 1.0 / s);
        hipLaunchKernelGGL(k_blur_cols, g2, b2, 0, st, other, cur, H, W, t / s, 1.0 / s);
    }
    if (level > 0) {
        hipLaunchKernelGGL(k_morph, g2, b2, 0, st, cur, other, H, W, level);
        double* t = cur; cur = other; other = t;
    }
    hipLaunchKernelGGL(k_sumsq, dim3(256), dim3(256), 0, st, cur, n, fb.d_norm);
    const dim3 g3((h->Wp + 255) / 256, h->Hp);
    hipLaunchKernelGGL(k_store, g3, b2, 0, st, cur, fb.d_norm, h->dframe + (size_t)slot * h->Hp * h->Wp, H, W, h->Hp, h->Wp, h->tiled,
                       h->cfg.nc ? 0 : 1);
    e = hipGetLastError();
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    double ss = 0.0;
    e = hipMemcpyAsync(&ss, fb.d_norm, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return eds_internal_fail(EDS_ERR_HIP, hipGetErrorString(e));
    if (norm_out) *norm_out = std::sqrt(ss);
    h->slots[slot].has_frame = true;
    return EDS_OK;
}
