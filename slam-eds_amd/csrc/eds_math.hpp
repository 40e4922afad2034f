// Small fp64 geometry / linear algebra used on BOTH sides of the PCIe bus: the host
// solver (EDS_EXEC_HOST) and the single "solver lane" of the persistent device kernel
// (EDS_EXEC_DEVICE) run exactly this code, so the two execution modes take identical steps.
//
// Semantics follow the types the reference keeps unchanged:
//   * Eigen::Quaterniond (x,y,z,w storage) -> toRotationMatrix  (PhotometricError.hpp:163)
//   * Sophus SE3 exp, tangent [upsilon; omega], epsilon 1e-10    (reference src/sophus/se3.hpp:406-428,
//     so3.hpp:343-369, sophus.hpp:45-47)
//   * ceres::EigenQuaternionParameterization / UnitNormVectorAddition Plus (Tracker.cpp:111-114;
//     PhotometricError.hpp:32-54)
#pragma once
#include <math.h>

#include "eds_layout.hpp"

#if defined(__HIPCC__)
#define EDS_HD __host__ __device__ inline
#else
#define EDS_HD inline
#endif

namespace edsm {

EDS_HD void quat_to_R(const double* q, double* R) {
    const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Hamilton product a*b, xyzw storage; out may alias a or b.
EDS_HD void quat_mul(const double* a, const double* b, double* out) {
    const double x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    out[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    out[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    out[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    out[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
}

// T <- exp(xi) * T with T = (t, q).  Sophus closed form incl. its small-angle branch.
EDS_HD void se3_left_update(const double* xi, double* t, double* q) {
    const double* ups = xi;
    const double* om = xi + 3;
    const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double th = sqrt(th2);
    double imag, real;
    if (th < 1e-10) {
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
    } else {
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
    }
    double dq[4] = {imag * om[0], imag * om[1], imag * om[2], real};
    double Rd[9];
    quat_to_R(dq, Rd);
    // V = I + c1 [om]x + c2 [om]x^2  (or R itself below epsilon)
    double V[9];
    if (th < 1e-10) {
        for (int i = 0; i < 9; ++i) V[i] = Rd[i];
    } else {
        const double c1 = (1.0 - cos(th)) / th2, c2 = (th - sin(th)) / (th2 * th);
        const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double o2 = 0;
                for (int k = 0; k < 3; ++k) o2 += O[3 * i + k] * O[3 * k + j];
                V[3 * i + j] = (i == j ? 1.0 : 0.0) + c1 * O[3 * i + j] + c2 * o2;
            }
    }
    double nt[3];
    for (int i = 0; i < 3; ++i)
        nt[i] = Rd[3 * i] * t[0] + Rd[3 * i + 1] * t[1] + Rd[3 * i + 2] * t[2] +
                V[3 * i] * ups[0] + V[3 * i + 1] * ups[1] + V[3 * i + 2] * ups[2];
    for (int i = 0; i < 3; ++i) t[i] = nt[i];
    double nq[4];
    quat_mul(dq, q, nq);
    const double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int i = 0; i < 4; ++i) q[i] = nq[i] / nn;
}

// Ceres Plus over (p additive | EigenQuaternionParameterization | UnitNormVectorAddition).
EDS_HD void state_plus12(const double* p, const double* q, const double* v, const double* d,
                         double* po, double* qo, double* vo) {
    for (int i = 0; i < 3; ++i) po[i] = p[i] + d[i];
    const double nd = sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]);
    if (nd > 0.0) {
        const double s = sin(nd) / nd;
        const double qd[4] = {s * d[3], s * d[4], s * d[5], cos(nd)};
        double nq[4];
        quat_mul(qd, q, nq);
        for (int i = 0; i < 4; ++i) qo[i] = nq[i];
    } else {
        for (int i = 0; i < 4; ++i) qo[i] = q[i];
    }
    double s2 = 0.0, tmp[6];
    for (int i = 0; i < 6; ++i) { tmp[i] = v[i] + d[6 + i]; s2 += tmp[i] * tmp[i]; }
    const double inv = 1.0 / sqrt(s2);
    for (int i = 0; i < 6; ++i) vo[i] = tmp[i] * inv;
}

// Dense SPD solve A x = b (n <= 12) by Cholesky; false if A is not numerically PD.
EDS_HD bool cholesky_solve(int n, const double* A, const double* b, double* x) {
    double L[144], y[12];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(s > 0.0) || !(s < 1e300)) return false;
                L[i * n + i] = sqrt(s);
            } else {
                L[i * n + j] = s / L[j * n + j];
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = 0; i < n; ++i)
        if (!(x[i] == x[i]) || !(fabs(x[i]) < 1e300)) return false;
    return true;
}

// Fills the per-pass constants of a slot's pose block from (p, q, v) and the per-block
// Gram matrices G (36 doubles each): rotation, per-block 1/n and G v / n^3
// (n^2 = v^T G v + 1e-3: PhotometricError.hpp:132-149 in closed form), and the local
// Jacobian of the unit-norm velocity plus.  The intrinsics/point-count entries are
// written by set_keyframe and left alone here.
EDS_HD void fill_pose_block(const double* p, const double* q, const double* v, const double* G, int nb, double* pb) {
    quat_to_R(q, pb + EDS_PB_R);
    for (int i = 0; i < 3; ++i) pb[EDS_PB_T + i] = p[i];
    for (int i = 0; i < 6; ++i) pb[EDS_PB_V + i] = v[i];
    for (int i = 0; i < 4; ++i) pb[EDS_PB_Q + i] = q[i];
    double vv = 0.0;
    for (int i = 0; i < 6; ++i) vv += v[i] * v[i];
    const double vn = sqrt(vv);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) pb[EDS_PB_PV + 6 * i + j] = ((i == j ? 1.0 : 0.0) - v[i] * v[j] / vv) / vn;
    for (int k = 0; k < nb; ++k) {
        const double* Gk = G + 36 * k;
        double Gv[6], S = 1e-3;
        for (int i = 0; i < 6; ++i) {
            double s = 0.0;
            for (int j = 0; j < 6; ++j) s += Gk[6 * i + j] * v[j];
            Gv[i] = s;
            S += v[i] * s;
        }
        const double n = sqrt(S);
        double* o = pb + EDS_PB_BLK + EDS_PB_BLK_STRIDE * k;
        o[0] = 1.0 / n;
        for (int i = 0; i < 6; ++i) o[1 + i] = Gv[i] / (n * n * n);
        o[7] = S;
    }
}

}  // namespace edsm
