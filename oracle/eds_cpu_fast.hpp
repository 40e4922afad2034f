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
//   * fp64 only for the projection (sub-pixel phase at u ~ 640), the 28 running sums and the 6x6 solve.
// Checked against the oracle in tests/test_oracle.py (pose within 1e-5 of pose6_lm, same accept pattern).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

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
    P->X.resize(N); P->Y.resize(N); P->Z.resize(N); P->w.resize(N); P->mhat.resize(N);
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

// Returns the number of iterations executed; accepted[] like pose6_lm.
inline int lm6(const Prepared& P, double p[3], double q[4], int iters, double lambda0, int* accepted) {
    Sums cur, cand;
    evaluate(P, p, q, &cur);
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
        evaluate(P, pc, qc, &cand);
        const bool ok = cand.cost < cur.cost;
        if (accepted) accepted[it] = ok ? 1 : 0;
        if (ok) { std::memcpy(p, pc, sizeof(pc)); std::memcpy(q, qc, sizeof(qc)); cur = cand; lambda *= 0.5; }
        else { lambda *= 4.0; if (lambda < 1e-6) lambda = 1e-6; }
    }
    return it;
}

}  // namespace eds_cpu_fast
