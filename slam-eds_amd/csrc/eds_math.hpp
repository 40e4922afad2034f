// Small fp64 geometry / linear algebra used on BOTH sides of the PCIe bus: the host
// solver (EDS_EXEC_HOST) and the single "solver lane" of the persistent device kernel
// (EDS_EXEC_DEVICE) run exactly this code, so the two execution modes take identical steps.
// Everything is written with compile-time extents and fully unrolled loops so that on the GPU
// all temporaries live in VGPRs (no scratch): the solver lane is the serial section of every
// iteration and must stay a few microseconds.
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
#if defined(__clang__)
#define EDS_UNROLL _Pragma("unroll")
#else
#define EDS_UNROLL
#endif

namespace edsm {

// 1/sqrt(x).  On the GPU one v_rsq_f64 + refinement replaces a sqrt AND a division (each a
// 15-20 instruction fp64 sequence on the serial solver lane); the host spells it out.
EDS_HD double rsqrt_(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(x);
#else
    return 1.0 / sqrt(x);
#endif
}

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

// sin and cos of a half-angle.  On the GPU, where this sits on the serial path of every LM iteration, |x| <= 0.5 (rotation
// increments below 57 degrees: all of tracking) takes the Taylor polynomials to x^17 / x^18 (truncation below 1e-23, i.e. the
// same fp64 value as the library to the last bits) instead of the ~100-instruction library call with its argument reduction.
EDS_HD void sincos_half(double x, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (fabs(x) <= 0.5) {
        const double z = x * x;
        double ps = 1.0 / 355687428096000.0;         // 1/17!
        ps = fma(-ps, z, 1.0 / 1307674368000.0);     // 1/15! - z/17!
        ps = fma(-ps, z, 1.0 / 6227020800.0);        // 1/13! - ...
        ps = fma(-ps, z, 1.0 / 39916800.0);
        ps = fma(-ps, z, 1.0 / 362880.0);
        ps = fma(-ps, z, 1.0 / 5040.0);
        ps = fma(-ps, z, 1.0 / 120.0);
        ps = fma(-ps, z, 1.0 / 6.0);
        *s = fma(-x * z, ps, x);                     // x - x^3 (1/3! - z (1/5! - ...))
        double pc = 1.0 / 6402373705728000.0;        // 1/18!
        pc = fma(-pc, z, 1.0 / 20922789888000.0);    // 1/16! - z/18!
        pc = fma(-pc, z, 1.0 / 87178291200.0);
        pc = fma(-pc, z, 1.0 / 479001600.0);
        pc = fma(-pc, z, 1.0 / 3628800.0);
        pc = fma(-pc, z, 1.0 / 40320.0);
        pc = fma(-pc, z, 1.0 / 720.0);
        pc = fma(-pc, z, 1.0 / 24.0);
        pc = fma(-pc, z, 0.5);
        *c = fma(-z, pc, 1.0);                       // 1 - z (1/2! - z (1/4! - ...))
        return;
    }
#endif
    sincos(x, s, c);
}

// T <- exp(xi) * T with T = (t, q).  Sophus closed form incl. its small-angle branch.
EDS_HD void se3_left_update(const double* xi, double* t, double* q) {
    const double u0 = xi[0], u1 = xi[1], u2 = xi[2], o0 = xi[3], o1 = xi[4], o2 = xi[5];
    const double th2 = o0 * o0 + o1 * o1 + o2 * o2;
    const bool tiny = th2 < 1e-20;              // |omega| < 1e-10, Sophus' epsilon
    double imag, real, c1, c2;
    if (tiny) {
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
        c1 = 0.0; c2 = 0.0;
    } else {
        // one half-angle sincos feeds everything: sin(th) = 2 sh ch, 1 - cos(th) = 2 sh^2
        // (same cancellation in th - sin(th) as the Sophus expression; it only scales O(th^2) terms)
        const double inv_th = rsqrt_(th2), inv_th2 = inv_th * inv_th;
        const double th = th2 * inv_th;
        double sh, ch;
        sincos_half(0.5 * th, &sh, &ch);
        imag = sh * inv_th;
        real = ch;
        c1 = 2.0 * sh * sh * inv_th2;
        c2 = (th - 2.0 * sh * ch) * (inv_th2 * inv_th);
    }
    double dq[4] = {imag * o0, imag * o1, imag * o2, real};
    double Rd[9];
    quat_to_R(dq, Rd);
    // V = I + c1 [om]x + c2 [om]x^2  (Sophus uses R itself below epsilon)
    double V[9];
    if (tiny) {
        EDS_UNROLL
        for (int i = 0; i < 9; ++i) V[i] = Rd[i];
    } else {
        // [om]x^2 = om om^T - |om|^2 I
        V[0] = 1.0 + c2 * (o0 * o0 - th2); V[1] = -c1 * o2 + c2 * (o0 * o1);   V[2] = c1 * o1 + c2 * (o0 * o2);
        V[3] = c1 * o2 + c2 * (o1 * o0);   V[4] = 1.0 + c2 * (o1 * o1 - th2); V[5] = -c1 * o0 + c2 * (o1 * o2);
        V[6] = -c1 * o1 + c2 * (o2 * o0);  V[7] = c1 * o0 + c2 * (o2 * o1);   V[8] = 1.0 + c2 * (o2 * o2 - th2);
    }
    const double t0 = t[0], t1 = t[1], t2 = t[2];
    t[0] = Rd[0] * t0 + Rd[1] * t1 + Rd[2] * t2 + V[0] * u0 + V[1] * u1 + V[2] * u2;
    t[1] = Rd[3] * t0 + Rd[4] * t1 + Rd[5] * t2 + V[3] * u0 + V[4] * u1 + V[5] * u2;
    t[2] = Rd[6] * t0 + Rd[7] * t1 + Rd[8] * t2 + V[6] * u0 + V[7] * u1 + V[8] * u2;
    double nq[4];
    quat_mul(dq, q, nq);
    const double inv = rsqrt_(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    q[0] = nq[0] * inv; q[1] = nq[1] * inv; q[2] = nq[2] * inv; q[3] = nq[3] * inv;
}

// Ceres Plus over (p additive | EigenQuaternionParameterization | UnitNormVectorAddition).
EDS_HD void state_plus12(const double* p, const double* q, const double* v, const double* d,
                         double* po, double* qo, double* vo) {
    EDS_UNROLL
    for (int i = 0; i < 3; ++i) po[i] = p[i] + d[i];
    const double nd2 = d[3] * d[3] + d[4] * d[4] + d[5] * d[5];
    const double nd = sqrt(nd2);
    if (nd > 0.0) {
        double sn, cs, s;
#if defined(__HIP_DEVICE_COMPILE__)
        if (nd2 <= 0.25) {              // |delta| <= 0.5 rad: sin(nd)/nd and cos(nd) as polynomials in nd^2 (truncation < 1e-23): no
            const double z = nd2;       // library sincos, no division on the solver's critical path
            double ps = 1.0 / 355687428096000.0;
            ps = fma(-ps, z, 1.0 / 1307674368000.0); ps = fma(-ps, z, 1.0 / 6227020800.0); ps = fma(-ps, z, 1.0 / 39916800.0);
            ps = fma(-ps, z, 1.0 / 362880.0); ps = fma(-ps, z, 1.0 / 5040.0); ps = fma(-ps, z, 1.0 / 120.0); ps = fma(-ps, z, 1.0 / 6.0);
            s = fma(-z, ps, 1.0);
            double pc = 1.0 / 6402373705728000.0;
            pc = fma(-pc, z, 1.0 / 20922789888000.0); pc = fma(-pc, z, 1.0 / 87178291200.0); pc = fma(-pc, z, 1.0 / 479001600.0);
            pc = fma(-pc, z, 1.0 / 3628800.0); pc = fma(-pc, z, 1.0 / 40320.0); pc = fma(-pc, z, 1.0 / 720.0); pc = fma(-pc, z, 1.0 / 24.0);
            pc = fma(-pc, z, 0.5);
            cs = fma(-z, pc, 1.0);
            sn = s * nd;
        } else
#endif
        {
            sincos(nd, &sn, &cs);       // one range reduction for both (the serial solver lane pays for every instruction)
            s = sn / nd;
        }
        (void)sn;
        const double qd[4] = {s * d[3], s * d[4], s * d[5], cs};
        double nq[4];
        quat_mul(qd, q, nq);
        EDS_UNROLL
        for (int i = 0; i < 4; ++i) qo[i] = nq[i];
    } else {
        EDS_UNROLL
        for (int i = 0; i < 4; ++i) qo[i] = q[i];
    }
    double s2 = 0.0, tmp[6];
    EDS_UNROLL
    for (int i = 0; i < 6; ++i) { tmp[i] = v[i] + d[6 + i]; s2 += tmp[i] * tmp[i]; }
    const double inv = rsqrt_(s2);
    EDS_UNROLL
    for (int i = 0; i < 6; ++i) vo[i] = tmp[i] * inv;
}

// In-place Cholesky solve on a PACKED lower triangle (row i, col j <= i at i(i+1)/2 + j):
// on entry L holds the lower triangle of an SPD matrix and b the right-hand side, on exit L holds
// its Cholesky factor and b the solution.  Returns false if the matrix is not numerically PD.
#define EDS_TRI(i, j) ((i) * ((i) + 1) / 2 + (j))
template <int N>
EDS_HD bool chol_solve_packed(double* L, double* b) {
    bool ok = true;
    double id[N];                           // 1 / L_jj: fp64 divisions are ~20 instructions on the GPU, multiplies are one
    EDS_UNROLL
    for (int j = 0; j < N; ++j) {
        double d = L[EDS_TRI(j, j)];
        EDS_UNROLL
        for (int k = 0; k < j; ++k) d -= L[EDS_TRI(j, k)] * L[EDS_TRI(j, k)];
        ok = ok && (d > 0.0) && (d < 1e300);
        const double inv = rsqrt_(d);
        const double ljj = d * inv;
        id[j] = inv;
        L[EDS_TRI(j, j)] = ljj;
        EDS_UNROLL
        for (int i = j + 1; i < N; ++i) {
            double s = L[EDS_TRI(i, j)];
            EDS_UNROLL
            for (int k = 0; k < j; ++k) s -= L[EDS_TRI(i, k)] * L[EDS_TRI(j, k)];
            L[EDS_TRI(i, j)] = s * inv;
        }
    }
    if (!ok) return false;
    EDS_UNROLL
    for (int i = 0; i < N; ++i) {           // L y = b
        double s = b[i];
        EDS_UNROLL
        for (int k = 0; k < i; ++k) s -= L[EDS_TRI(i, k)] * b[k];
        b[i] = s * id[i];
    }
    EDS_UNROLL
    for (int i = N - 1; i >= 0; --i) {      // L^T x = y
        double s = b[i];
        EDS_UNROLL
        for (int k = i + 1; k < N; ++k) s -= L[EDS_TRI(k, i)] * b[k];
        b[i] = s * id[i];
    }
    double chk = 0.0;
    EDS_UNROLL
    for (int i = 0; i < N; ++i) chk += b[i];
    return (chk == chk) && (fabs(chk) < 1e300);
}

// Dense SPD solve A x = b for a full row-major A (n = 6 or 12 take the unrolled path).
template <int N>
EDS_HD bool cholesky_solve_n(const double* A, const double* b, double* x) {
    double L[N * (N + 1) / 2], y[N];
    EDS_UNROLL
    for (int i = 0; i < N; ++i) {
        EDS_UNROLL
        for (int j = 0; j <= i; ++j) L[EDS_TRI(i, j)] = A[i * N + j];
        y[i] = b[i];
    }
    const bool ok = chol_solve_packed<N>(L, y);
    EDS_UNROLL
    for (int i = 0; i < N; ++i) x[i] = y[i];
    return ok;
}
EDS_HD bool cholesky_solve(int n, const double* A, const double* b, double* x) {
    if (n == 6) return cholesky_solve_n<6>(A, b, x);
    if (n == 12) return cholesky_solve_n<12>(A, b, x);
    return false;
}

// Fills the per-pass constants of a slot's pose block from (p, q, v) and the per-block
// Gram matrices G (36 doubles each): rotation, per-block 1/n and G v / n^3
// (n^2 = v^T G v + 1e-3: PhotometricError.hpp:132-149 in closed form), and the local
// Jacobian of the unit-norm velocity plus.  The intrinsics/point-count entries are
// written by set_keyframe and left alone here.
// R - I straight from the quaternion (no 1 - (...) cancellation): exact to fp64 even for tiny rotations.
EDS_HD void quat_to_RmI(const double* q, double* D) {
    const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    D[0] = -(tyy + tzz); D[1] = txy - twz;    D[2] = txz + twy;
    D[3] = txy + twz;    D[4] = -(txx + tzz); D[5] = tyz - twx;
    D[6] = txz - twy;    D[7] = tyz + twx;    D[8] = -(txx + tyy);
}
// Rotation part of a pose block (R and R - I) + translation: what changes from pass to pass.
EDS_HD void fill_pose_rt(const double* p, const double* q, double* pb) {
    quat_to_R(q, pb + EDS_PB_R);
    quat_to_RmI(q, pb + EDS_PB_D);
    for (int i = 0; i < 3; ++i) pb[EDS_PB_T + i] = p[i];
}

EDS_HD void fill_pose_block(const double* p, const double* q, const double* v, const double* G, int nb, double* pb) {
    quat_to_R(q, pb + EDS_PB_R);
    quat_to_RmI(q, pb + EDS_PB_D);
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
        double S = 1e-3;
        double* o = pb + EDS_PB_BLK + EDS_PB_BLK_STRIDE * k;
        for (int i = 0; i < 6; ++i) {
            double s = 0.0;
            for (int j = 0; j < 6; ++j) s += Gk[6 * i + j] * v[j];
            o[1 + i] = s;
            S += v[i] * s;
        }
        const double n = sqrt(S);
        o[0] = 1.0 / n;
        for (int i = 0; i < 6; ++i) o[1 + i] = o[1 + i] / (n * n * n);
        o[7] = S;
    }
}

}  // namespace edsm
