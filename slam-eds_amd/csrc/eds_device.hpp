// Device-side building blocks (gfx950 / CDNA4, wave64) shared by the streaming kernels
// (eds_kernels.hip) and the persistent per-alignment solver (eds_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "eds_layout.hpp"

namespace edsd {

// ---------------------------------------------------------------------------------------
// Frame sampling.  The frame is fp32 row-major; indices clamp to the border exactly like
// ceres::Grid2D::GetValue (reference use: PhotometricError.hpp:110-111).
// ---------------------------------------------------------------------------------------
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte load
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Catmull-Rom cubic Hermite through p1 (x=0), p2 (x=1): value and derivative
// (ceres CubicHermiteSpline; same spline restated at reference src/utils/globalFuncs.h:192-207).
__device__ __forceinline__ void hermite(float p0, float p1, float p2, float p3, float x, float& f, float& df) {
    const float a = 0.5f * (-p0 + 3.0f * p1 - 3.0f * p2 + p3);
    const float b = 0.5f * (2.0f * p0 - 5.0f * p1 + 4.0f * p2 - p3);
    const float c = 0.5f * (p2 - p0);
    f = p1 + x * (c + x * (b + x * a));
    df = c + x * (2.0f * b + 3.0f * a * x);
}

// Loads the 4x4 neighbourhood rows r0-1..r0+2, cols c0-1..c0+2 (clamped).
__device__ __forceinline__ void load_patch16(const float* __restrict__ frame, int H, int W, int r0, int c0, float (&p)[16]) {
    if (c0 >= 1 && c0 + 2 < W && r0 >= 1 && r0 + 2 < H) {           // interior: four 16-byte row segments
        const float* base = frame + (size_t)(r0 - 1) * W + (c0 - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4u v = *reinterpret_cast<const float4u*>(base + (size_t)k * W);
            p[4 * k + 0] = v.x; p[4 * k + 1] = v.y; p[4 * k + 2] = v.z; p[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* row = frame + (size_t)clampi(r0 - 1 + k, 0, H - 1) * W;
#pragma unroll
            for (int j = 0; j < 4; ++j) p[4 * k + j] = row[clampi(c0 - 1 + j, 0, W - 1)];
        }
    }
}

// Bicubic value and derivatives from a register-resident 4x4 patch (ay: row phase, ax: col phase).
__device__ __forceinline__ void bicubic_patch(const float (&p)[16], float ay, float ax, float& E, float& Erow, float& Ecol) {
    float f[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) hermite(p[4 * k], p[4 * k + 1], p[4 * k + 2], p[4 * k + 3], ax, f[k], d[k]);
    hermite(f[0], f[1], f[2], f[3], ay, E, Erow);
    float unused;
    hermite(d[0], d[1], d[2], d[3], ay, Ecol, unused);
}

__device__ __forceinline__ void load_patch4(const float* __restrict__ frame, int H, int W, int r0, int c0, float (&p)[4]) {
    if (c0 >= 0 && c0 + 1 < W && r0 >= 0 && r0 + 1 < H) {
        const float* base = frame + (size_t)r0 * W + c0;
        const float2u a = *reinterpret_cast<const float2u*>(base);
        const float2u b = *reinterpret_cast<const float2u*>(base + W);
        p[0] = a.x; p[1] = a.y; p[2] = b.x; p[3] = b.y;
    } else {
        const float* ra = frame + (size_t)clampi(r0, 0, H - 1) * W;
        const float* rb = frame + (size_t)clampi(r0 + 1, 0, H - 1) * W;
        const int ca = clampi(c0, 0, W - 1), cb = clampi(c0 + 1, 0, W - 1);
        p[0] = ra[ca]; p[1] = ra[cb]; p[2] = rb[ca]; p[3] = rb[cb];
    }
}
__device__ __forceinline__ void bilinear_patch(const float (&p)[4], float ay, float ax, float& E, float& Erow, float& Ecol) {
    const float top = p[0] + ax * (p[1] - p[0]), bot = p[2] + ax * (p[3] - p[2]);
    E = top + ay * (bot - top);
    Erow = bot - top;
    Ecol = (1.0f - ay) * (p[1] - p[0]) + ay * (p[3] - p[2]);
}

// Splits a projected pixel coordinate (fp64) into the integer cell and the fp32 phase,
// robust to NaN/inf/behind-camera projections (the reference has no in-bounds test,
// PhotometricError.hpp:157-172: Grid2D simply clamps).
__device__ __forceinline__ void split_coord(double u, int size, int& cell, float& phase) {
    const double lo = -8.0, hi = (double)size + 8.0;
    if (u >= lo && u <= hi) {
        const double fu = floor(u);
        cell = (int)fu;
        phase = (float)(u - fu);
    } else {                       // far outside (or NaN): every tap clamps to one border pixel
        cell = (u > hi) ? size + 8 : -8;
        phase = 0.0f;
    }
}

template <int SAMPLING>
__device__ __forceinline__ void sample_frame(const float* __restrict__ frame, int H, int W, double vrow, double ucol,
                                             float& E, float& Erow, float& Ecol) {
    int r0, c0;
    float ay, ax;
    split_coord(vrow, H, r0, ay);
    split_coord(ucol, W, c0, ax);
    if (SAMPLING == 0) {
        float p[16];
        load_patch16(frame, H, W, r0, c0, p);
        bicubic_patch(p, ay, ax, E, Erow, Ecol);
    } else {
        float p[4];
        load_patch4(frame, H, W, r0, c0, p);
        bilinear_patch(p, ay, ax, E, Erow, Ecol);
    }
}

// ---------------------------------------------------------------------------------------
// Per-point geometry.  P = R kp + t in fp64 (sub-pixel phase needs it), the rest in fp32.
// ---------------------------------------------------------------------------------------
struct PoseRT {                    // wave-uniform, lives in SGPRs
    double R[9], t[3], fx, fy, cx, cy;
};
__device__ __forceinline__ void load_pose(const double* __restrict__ pb, PoseRT& ps) {
#pragma unroll
    for (int i = 0; i < 9; ++i) ps.R[i] = pb[EDS_PB_R + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) ps.t[i] = pb[EDS_PB_T + i];
    ps.fx = pb[EDS_PB_K]; ps.fy = pb[EDS_PB_K + 1]; ps.cx = pb[EDS_PB_K + 2]; ps.cy = pb[EDS_PB_K + 3];
}

struct PointProj {
    float Px, Py, Pz;              // point in the event-frame camera
    float g0, g1, g2;              // dE/dP  (gradE_P of SURVEY §8a)
    float E;
};

// Projects, samples and forms dE/dP for one point.
template <int SAMPLING>
__device__ __forceinline__ void project_sample(const float* __restrict__ frame, int H, int W, const PoseRT& ps,
                                               double X, double Y, double Z, PointProj& o) {
    const double Px = ps.R[0] * X + ps.R[1] * Y + ps.R[2] * Z + ps.t[0];
    const double Py = ps.R[3] * X + ps.R[4] * Y + ps.R[5] * Z + ps.t[1];
    const double Pz = ps.R[6] * X + ps.R[7] * Y + ps.R[8] * Z + ps.t[2];
    const double iz = 1.0 / Pz;
    const double un = Px * iz, vn = Py * iz;
    const double u = ps.fx * un + ps.cx;       // column  (PhotometricError.hpp:167)
    const double v = ps.fy * vn + ps.cy;       // row     (PhotometricError.hpp:168)
    float E, Er, Ec;
    sample_frame<SAMPLING>(frame, H, W, v, u, E, Er, Ec);
    const float izf = (float)iz, unf = (float)un, vnf = (float)vn;
    const float dx = (float)ps.fx * Ec, dy = (float)ps.fy * Er;
    o.g0 = dx * izf;
    o.g1 = dy * izf;
    o.g2 = -(dx * unf + dy * vnf) * izf;
    o.Px = (float)Px; o.Py = (float)Py; o.Pz = (float)Pz;
    o.E = E;
}

// a_i = -(gx df0/dv + gy df1/dv): the row of the linear model m = A v
// (PhotometricError.hpp:114-122,136-143).
__device__ __forceinline__ void model_row(float x, float y, float rho, float gx, float gy, float (&a)[6]) {
    a[0] = gx * rho;
    a[1] = gy * rho;
    a[2] = -(gx * x + gy * y) * rho;
    a[3] = -(gx * x * y + gy * (1.0f + y * y));
    a[4] = gx * (1.0f + x * x) + gy * x * y;
    a[5] = -gx * y + gy * x;
}

// Residual block of point i for N points in nb blocks (Tracker.cpp:178-195).
__device__ __forceinline__ int block_of(int i, int ne, int nb) {
    if (ne <= 0) return nb - 1;
    const int k = i / ne;
    return k < nb ? k : nb - 1;
}

// ---------------------------------------------------------------------------------------
// Wavefront reduce-scatter: K (power of two, >= 32) partial sums per lane, 64 lanes.
// Each butterfly step halves the number of live values per lane, so the whole reduction
// costs K-1 (+1) cross-lane exchanges instead of 6*K.  No MFMA: this is a tall-skinny
// N x 7 -> 7 x 7 contraction.  On return lane l holds, in v[0 .. max(K/64,1)-1], the
// wavefront totals of value indices  wave_red_index<K>(l, j).
// ---------------------------------------------------------------------------------------
template <int K, int STEP>
struct WaveRedStep {
    static __device__ __forceinline__ void run(float* v, int lane) {
        constexpr int HALF = K >> (STEP + 1);
        if (HALF > 0) {
            const bool upper = (lane >> STEP) & 1;
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const float send = upper ? v[j] : v[j + HALF];
                const float keep = upper ? v[j + HALF] : v[j];
                v[j] = keep + __shfl_xor(send, 1 << STEP, 64);
            }
        } else {
            v[0] += __shfl_xor(v[0], 1 << STEP, 64);
        }
        WaveRedStep<K, STEP + 1>::run(v, lane);
    }
};
template <int K>
struct WaveRedStep<K, 6> {
    static __device__ __forceinline__ void run(float*, int) {}
};
template <int K>
__device__ __forceinline__ void wave_reduce_scatter(float* v, int lane) {
    WaveRedStep<K, 0>::run(v, lane);
}
template <int K>
__device__ __forceinline__ int wave_red_index(int lane, int j) {
    int idx = j;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int half = K >> (s + 1);
        if (half > 0) idx += ((lane >> s) & 1) * half;
    }
    return idx;
}

// Accumulates one point's contribution: upper triangle of J J^T (row-major pairs a<=b),
// then J r, then r^2 — the layout of the EDS_RED_* records.
template <int NC>
__device__ __forceinline__ void accumulate_normal(float* acc, const float (&J)[NC], float r, float hw, float cost_term) {
    int o = 0;
#pragma unroll
    for (int a = 0; a < NC; ++a) {
        const float ja = hw * J[a];
#pragma unroll
        for (int b = a; b < NC; ++b) acc[o++] += ja * J[b];
    }
#pragma unroll
    for (int a = 0; a < NC; ++a) acc[o++] += hw * J[a] * r;
    acc[o] += cost_term;
}

}  // namespace edsd
