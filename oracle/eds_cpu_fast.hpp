// Optimised CPU variant of the pose-only LM6 solve — the "what a tuned CPU port would do" baseline of SURVEY.md §8(d) /
// BASELINE.md §3, so that the GPU speed-up is not inflated by the oracle's forward-mode autodiff.
//
// BASELINE / TEST INFRASTRUCTURE ONLY, like everything under oracle/ (never linked into the product).  Same algorithm as
// eds_oracle::pose6_lm (damped Gauss-Newton, template reference src/tracking/CoarseTracker.cpp:545-664, on the residual of
// reference src/tracking/PhotometricError.hpp:124-182 with the velocity fixed), but the way a performance-minded CPU
// implementation would be written:
//   * structure-of-arrays fp32 point constants, built once per solve (back-projection and the normalised model hoisted out
//     of the iteration loop: PhotometricError.hpp:95-106,131-149),
//   * closed-form 1x6 SE(3) row instead of Jet<6> autodiff,
//   * fp32 frame with a replicated border (Grid2D's clamp without a branch per tap), fp32 bicubic,
//   * fp64 only for the projection (sub-pixel phase at u ~ 640), the 28 running sums and the 6x6 solve,
//   * round 5: the point loop VECTORISED over points (AVX2 + FMA, eight points per step: fp64 projection in two halves, the 16 taps by
//     vgatherdps, packed splines, the 28 products in fp32 lanes flushed to fp64 every 256 points) — evaluate_avx2(); the scalar
//     evaluate() stays as its checker (tests/test_oracle.py compares both with the oracle).
// Checked against the oracle in tests/test_oracle.py (pose within 1e-5 of pose6_lm, same accept pattern).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define EDS_CPU_FAST_AVX2 1
#endif

#include "eds_oracle.hpp"

namespace eds_cpu_fast {

struct Prepared {
    int N = 0, H = 0, W = 0, Wp = 0;
    std::vector<float> X, Y, Z, w, mhat;         // back-projected point, weight, normalised model
    std::vector<float> frame;                    // (H + 6) x (W + 6), 3 replicated pixels on every side
    double fx, fy, cx, cy;
};

inline void prepare(const eds_oracle::Problem& pb, const double* v, Prepared* P) {
    const int N = pb.N, M = 3;
    P->N = N; P->H = pb.H; P->W = pb.W; P->Wp = pb.W + 2 * M;
    P->fx = pb.fx; P->fy = pb.fy; P->cx = pb.cx; P->cy = pb.cy;
    const int N8 = (N + 7) & ~7;          // padded to whole vectors: points beyond N carry weight 0 (and sit in front of the camera)
    P->X.assign(N8, 0.0f); P->Y.assign(N8, 0.0f); P->Z.assign(N8, 1.0f); P->w.assign(N8, 0.0f); P->mhat.assign(N8, 0.0f);
    double S = 1e-3;
    std::vector<double> m(N);
    for (int i = 0; i < N; ++i) {
        const double x = pb.norm_coord[2 * i], y = pb.norm_coord[2 * i + 1], rho = pb.idp[i], gx = pb.grad[2 * i], gy = pb.grad[2 * i + 1];
        const double f0 = -rho * v[0] + x * rho * v[2] + x * y * v[3] - (1.0 + x * x) * v[4] + y * v[5];
        const double f1 = -rho * v[1] + y * rho * v[2] + (1.0 + y * y) * v[3] - x * y * v[4] - x * v[5];
        m[i] = -(gx * f0 + gy * f1);
        S += m[i] * m[i];
        const double z = 1.0 / (rho + 1e-5);
        P->X[i] = (float)(x * z); P->Y[i] = (float)(y * z); P->Z[i] = (float)z; P->w[i] = (float)pb.weights[i];
    }
    const double inv_n = 1.0 / std::sqrt(S);
    for (int i = 0; i < N; ++i) P->mhat[i] = (float)(m[i] * inv_n);
    P->frame.resize((size_t)(pb.H + 2 * M) * P->Wp);
    for (int r = -M; r < pb.H + M; ++r) {
        const double* src = pb.frame + (size_t)std::min(std::max(r, 0), pb.H - 1) * pb.W;
        float* dst = P->frame.data() + (size_t)(r + M) * P->Wp;
        for (int c = -M; c < pb.W + M; ++c) dst[c + M] = (float)src[std::min(std::max(c, 0), pb.W - 1)];
    }
}

inline void hermite(float p0, float p1, float p2, float p3, float x, float* f, float* df) {
    const float a = 0.5f * (-p0 + 3.0f * p1 - 3.0f * p2 + p3), b = 0.5f * (2.0f * p0 - 5.0f * p1 + 4.0f * p2 - p3), c = 0.5f * (p2 - p0);
    *f = p1 + x * (c + x * (b + x * a));
    *df = c + x * (2.0f * b + 3.0f * a * x);
}

struct Sums { double H[36], b[6], cost; };

inline void evaluate(const Prepared& P, const double* p, const double* q, Sums* out) {
    double R[9];
    eds_oracle::quat_to_R<double>(q, R);
    double acc[28];
    for (double& a : acc) a = 0.0;
    const float* fr = P.frame.data();
    const int Wp = P.Wp, H = P.H, W = P.W;
    for (int i = 0; i < P.N; ++i) {
        const double X = P.X[i], Y = P.Y[i], Z = P.Z[i];
        const double Px = R[0] * X + R[1] * Y + R[2] * Z + p[0], Py = R[3] * X + R[4] * Y + R[5] * Z + p[1], Pz = R[6] * X + R[7] * Y + R[8] * Z + p[2];
        const double iz = 1.0 / Pz, un = Px * iz, vn = Py * iz;
        const double u = P.fx * un + P.cx, v = P.fy * vn + P.cy;
        const double fu = std::floor(u), fv = std::floor(v);
        int c0 = (int)fu, r0 = (int)fv;
        const float ax = (float)(u - fu), ay = (float)(v - fv);
        c0 = std::min(std::max(c0, -2), W); r0 = std::min(std::max(r0, -2), H);       // clamped origin: replicated border = Grid2D clamp
        const float* t = fr + (size_t)(r0 - 1 + 3) * Wp + (c0 - 1 + 3);
        float f[4], d[4];
        for (int k = 0; k < 4; ++k) hermite(t[k * Wp], t[k * Wp + 1], t[k * Wp + 2], t[k * Wp + 3], ax, &f[k], &d[k]);
        float E, Er, Ec, unused;
        hermite(f[0], f[1], f[2], f[3], ay, &E, &Er);
        hermite(d[0], d[1], d[2], d[3], ay, &Ec, &unused);
        const float w = P.w[i], r = w * (P.mhat[i] - E);
        const float dx = (float)P.fx * Ec, dy = (float)P.fy * Er, izf = (float)iz;
        const float g0 = dx * izf, g1 = dy * izf, g2 = -(dx * (float)un + dy * (float)vn) * izf;
        const float pxf = (float)Px, pyf = (float)Py, pzf = (float)Pz;
        const float J[6] = {-w * g0, -w * g1, -w * g2, -w * (pyf * g2 - pzf * g1), -w * (pzf * g0 - pxf * g2), -w * (pxf * g1 - pyf * g0)};
        int o = 0;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) acc[o++] += (double)(J[a] * J[b]);
        for (int a = 0; a < 6; ++a) acc[o++] += (double)(J[a] * r);
        acc[o] += (double)(r * r);
    }
    int o = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { out->H[6 * a + b] = acc[o]; out->H[6 * b + a] = acc[o]; ++o; }
    for (int a = 0; a < 6; ++a) out->b[a] = acc[o++];
    out->cost = acc[o];
}

#ifdef EDS_CPU_FAST_AVX2
// The same sums, eight points per step.
inline void evaluate_avx2(const Prepared& P, const double* p, const double* q, Sums* out) {
    double R[9];
    eds_oracle::quat_to_R<double>(q, R);
    double acc[28];
    for (double& a : acc) a = 0.0;
    const float* fr = P.frame.data();
    const int Wp = P.Wp, N8 = (P.N + 7) & ~7;
    const __m256d r0 = _mm256_set1_pd(R[0]), r1 = _mm256_set1_pd(R[1]), r2 = _mm256_set1_pd(R[2]), r3 = _mm256_set1_pd(R[3]), r4 = _mm256_set1_pd(R[4]),
                  r5 = _mm256_set1_pd(R[5]), r6 = _mm256_set1_pd(R[6]), r7 = _mm256_set1_pd(R[7]), r8 = _mm256_set1_pd(R[8]);
    const __m256d t0 = _mm256_set1_pd(p[0]), t1 = _mm256_set1_pd(p[1]), t2 = _mm256_set1_pd(p[2]);
    const __m256d fxd = _mm256_set1_pd(P.fx), fyd = _mm256_set1_pd(P.fy), cxd = _mm256_set1_pd(P.cx), cyd = _mm256_set1_pd(P.cy), one = _mm256_set1_pd(1.0);
    const __m256 fxf = _mm256_set1_ps((float)P.fx), fyf = _mm256_set1_ps((float)P.fy), half = _mm256_set1_ps(0.5f);
    const __m256i wp = _mm256_set1_epi32(Wp), lo = _mm256_set1_epi32(-2), hiW = _mm256_set1_epi32(P.W), hiH = _mm256_set1_epi32(P.H), two = _mm256_set1_epi32(2);
    __m256 vacc[28];
    auto flush = [&]() {
        for (int k = 0; k < 28; ++k) {
            const __m256d a = _mm256_add_pd(_mm256_cvtps_pd(_mm256_castps256_ps128(vacc[k])), _mm256_cvtps_pd(_mm256_extractf128_ps(vacc[k], 1)));
            alignas(32) double t[4];
            _mm256_store_pd(t, a);
            acc[k] += (t[0] + t[1]) + (t[2] + t[3]);
            vacc[k] = _mm256_setzero_ps();
        }
    };
    for (int k = 0; k < 28; ++k) vacc[k] = _mm256_setzero_ps();
    auto hermite8 = [&](__m256 p0, __m256 p1, __m256 p2, __m256 p3, __m256 x, __m256* f, __m256* df) {
        const __m256 three = _mm256_set1_ps(3.0f), twof = _mm256_set1_ps(2.0f), five = _mm256_set1_ps(5.0f), four = _mm256_set1_ps(4.0f);
        const __m256 a = _mm256_mul_ps(half, _mm256_add_ps(_mm256_sub_ps(p3, p0), _mm256_mul_ps(three, _mm256_sub_ps(p1, p2))));
        const __m256 b = _mm256_mul_ps(half, _mm256_sub_ps(_mm256_add_ps(_mm256_mul_ps(twof, p0), _mm256_mul_ps(four, p2)), _mm256_add_ps(_mm256_mul_ps(five, p1), p3)));
        const __m256 c = _mm256_mul_ps(half, _mm256_sub_ps(p2, p0));
        *f = _mm256_fmadd_ps(x, _mm256_fmadd_ps(x, _mm256_fmadd_ps(x, a, b), c), p1);
        *df = _mm256_fmadd_ps(x, _mm256_fmadd_ps(_mm256_mul_ps(three, a), x, _mm256_mul_ps(twof, b)), c);
    };
    for (int i = 0; i < N8; i += 8) {
        // projection in fp64, two halves of four points
        __m256 axf, ayf, izf, unf, vnf, pxf, pyf, pzf;
        __m256i c0, rr0;
        {
            __m128 ax_[2], ay_[2], iz_[2], un_[2], vn_[2], px_[2], py_[2], pz_[2];
            __m128i c_[2], r_[2];
            for (int h = 0; h < 2; ++h) {
                const __m256d X = _mm256_cvtps_pd(_mm_loadu_ps(&P.X[i + 4 * h])), Y = _mm256_cvtps_pd(_mm_loadu_ps(&P.Y[i + 4 * h])), Z = _mm256_cvtps_pd(_mm_loadu_ps(&P.Z[i + 4 * h]));
                const __m256d Px = _mm256_fmadd_pd(r0, X, _mm256_fmadd_pd(r1, Y, _mm256_fmadd_pd(r2, Z, t0)));
                const __m256d Py = _mm256_fmadd_pd(r3, X, _mm256_fmadd_pd(r4, Y, _mm256_fmadd_pd(r5, Z, t1)));
                const __m256d Pz = _mm256_fmadd_pd(r6, X, _mm256_fmadd_pd(r7, Y, _mm256_fmadd_pd(r8, Z, t2)));
                const __m256d iz = _mm256_div_pd(one, Pz), un = _mm256_mul_pd(Px, iz), vn = _mm256_mul_pd(Py, iz);
                const __m256d u = _mm256_fmadd_pd(fxd, un, cxd), v = _mm256_fmadd_pd(fyd, vn, cyd);
                const __m256d fu = _mm256_floor_pd(u), fv = _mm256_floor_pd(v);
                ax_[h] = _mm256_cvtpd_ps(_mm256_sub_pd(u, fu)); ay_[h] = _mm256_cvtpd_ps(_mm256_sub_pd(v, fv));
                c_[h] = _mm256_cvttpd_epi32(fu); r_[h] = _mm256_cvttpd_epi32(fv);
                iz_[h] = _mm256_cvtpd_ps(iz); un_[h] = _mm256_cvtpd_ps(un); vn_[h] = _mm256_cvtpd_ps(vn);
                px_[h] = _mm256_cvtpd_ps(Px); py_[h] = _mm256_cvtpd_ps(Py); pz_[h] = _mm256_cvtpd_ps(Pz);
            }
            axf = _mm256_set_m128(ax_[1], ax_[0]); ayf = _mm256_set_m128(ay_[1], ay_[0]); izf = _mm256_set_m128(iz_[1], iz_[0]);
            unf = _mm256_set_m128(un_[1], un_[0]); vnf = _mm256_set_m128(vn_[1], vn_[0]);
            pxf = _mm256_set_m128(px_[1], px_[0]); pyf = _mm256_set_m128(py_[1], py_[0]); pzf = _mm256_set_m128(pz_[1], pz_[0]);
            c0 = _mm256_set_m128i(c_[1], c_[0]); rr0 = _mm256_set_m128i(r_[1], r_[0]);
        }
        c0 = _mm256_min_epi32(_mm256_max_epi32(c0, lo), hiW); rr0 = _mm256_min_epi32(_mm256_max_epi32(rr0, lo), hiH);      // clamped origin: replicated border = Grid2D clamp
        const __m256i base = _mm256_add_epi32(_mm256_mullo_epi32(_mm256_add_epi32(rr0, two), wp), _mm256_add_epi32(c0, two));   // (r0 - 1 + 3) * Wp + (c0 - 1 + 3)
        __m256 f[4], d[4];
        for (int k = 0; k < 4; ++k) {
            const __m256i o = _mm256_add_epi32(base, _mm256_set1_epi32(k * Wp));
            const __m256 p0 = _mm256_i32gather_ps(fr, o, 4), p1 = _mm256_i32gather_ps(fr + 1, o, 4), p2 = _mm256_i32gather_ps(fr + 2, o, 4), p3 = _mm256_i32gather_ps(fr + 3, o, 4);
            hermite8(p0, p1, p2, p3, axf, &f[k], &d[k]);
        }
        __m256 E, Er, Ec, unused;
        hermite8(f[0], f[1], f[2], f[3], ayf, &E, &Er);
        hermite8(d[0], d[1], d[2], d[3], ayf, &Ec, &unused);
        const __m256 w = _mm256_loadu_ps(&P.w[i]), r = _mm256_mul_ps(w, _mm256_sub_ps(_mm256_loadu_ps(&P.mhat[i]), E));
        const __m256 dx = _mm256_mul_ps(fxf, Ec), dy = _mm256_mul_ps(fyf, Er);
        const __m256 g0 = _mm256_mul_ps(dx, izf), g1 = _mm256_mul_ps(dy, izf);
        const __m256 g2 = _mm256_mul_ps(_mm256_sub_ps(_mm256_setzero_ps(), _mm256_fmadd_ps(dx, unf, _mm256_mul_ps(dy, vnf))), izf);
        const __m256 nw = _mm256_sub_ps(_mm256_setzero_ps(), w);
        const __m256 J[6] = {_mm256_mul_ps(nw, g0), _mm256_mul_ps(nw, g1), _mm256_mul_ps(nw, g2),
                             _mm256_mul_ps(nw, _mm256_fmsub_ps(pyf, g2, _mm256_mul_ps(pzf, g1))), _mm256_mul_ps(nw, _mm256_fmsub_ps(pzf, g0, _mm256_mul_ps(pxf, g2))),
                             _mm256_mul_ps(nw, _mm256_fmsub_ps(pxf, g1, _mm256_mul_ps(pyf, g0)))};
        int o = 0;
        for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { vacc[o] = _mm256_fmadd_ps(J[a], J[b], vacc[o]); ++o; }
        for (int a = 0; a < 6; ++a) { vacc[o] = _mm256_fmadd_ps(J[a], r, vacc[o]); ++o; }
        vacc[o] = _mm256_fmadd_ps(r, r, vacc[o]);
        if (((i + 8) & 255) == 0) flush();          // fp32 lanes hold 32 products each between flushes
    }
    flush();
    int o = 0;
    for (int a = 0; a < 6; ++a) for (int b = a; b < 6; ++b) { out->H[6 * a + b] = acc[o]; out->H[6 * b + a] = acc[o]; ++o; }
    for (int a = 0; a < 6; ++a) out->b[a] = acc[o++];
    out->cost = acc[o];
}
#endif

inline void evaluate_best(const Prepared& P, const double* p, const double* q, Sums* out) {
#ifdef EDS_CPU_FAST_AVX2
    evaluate_avx2(P, p, q, out);
#else
    evaluate(P, p, q, out);
#endif
}

// Returns the number of iterations executed; accepted[] like pose6_lm.
inline int lm6(const Prepared& P, double p[3], double q[4], int iters, double lambda0, int* accepted, bool scalar = false) {
    Sums cur, cand;
    auto eval = [&](const double* pp, const double* qq, Sums* o) { if (scalar) evaluate(P, pp, qq, o); else evaluate_best(P, pp, qq, o); };
    eval(p, q, &cur);
    double lambda = lambda0;
    int it = 0;
    for (; it < iters; ++it) {
        double Hl[36], nb[6], xi[6];
        std::memcpy(Hl, cur.H, sizeof(Hl));
        for (int i = 0; i < 6; ++i) { Hl[7 * i] *= (1.0 + lambda); nb[i] = -cur.b[i]; }
        if (!eds_oracle::cholesky_solve(6, Hl, nb, xi)) break;
        double pc[3], qc[4];
        std::memcpy(pc, p, sizeof(pc)); std::memcpy(qc, q, sizeof(qc));
        eds_oracle::se3_left_update(xi, pc, qc);
        eval(pc, qc, &cand);
        const bool ok = cand.cost < cur.cost;
        if (accepted) accepted[it] = ok ? 1 : 0;
        if (ok) { std::memcpy(p, pc, sizeof(pc)); std::memcpy(q, qc, sizeof(qc)); cur = cand; lambda *= 0.5; }
        else { lambda *= 4.0; if (lambda < 1e-6) lambda = 1e-6; }
    }
    return it;
}

}  // namespace eds_cpu_fast
