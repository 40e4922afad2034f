// =============================================================================
// eds_oracle.hpp — CPU ORACLE for the EDS event-to-model photometric tracker.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may build, link or call it.
// The shipped path (slam-eds_amd/csrc, libeds_hip.so) never includes this file.
//
// PARITY UNPINNED: the reference (uzh-rpg/slam-eds) ships no tests, fixtures
// or golden vectors and its hot path cannot be compiled here
// (needs Ceres, Eigen, OpenCV, Boost, yaml-cpp, Rock base-types — all absent,
// SURVEY.md §8c).  This file is therefore a from-scratch fp64 restatement of
// the reference's *behaviour*, each function citing the reference file:line it
// follows, plus a restatement of the published algorithms of the third-party
// pieces the reference delegates to:
//   * Ceres Solver (<= 2.1; version not pinned by the reference, manifest.xml:12):
//     Jet forward-mode autodiff, Grid2D + BiCubicInterpolator
//     (ceres/cubic_interpolation.h), HuberLoss/CauchyLoss + Corrector,
//     EigenQuaternionParameterization, AutoDiffLocalParameterization,
//     TrustRegionMinimizer + LevenbergMarquardtStrategy with default options.
//   * Eigen3: Quaternion::toRotationMatrix.
//   * Sophus (vendored in the reference, src/sophus): SE3/SO3 exp/log.
// It is cross-checked by an independent numpy implementation
// (oracle/np_oracle.py: closed forms + central finite differences).
// =============================================================================
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace eds_oracle {

// -----------------------------------------------------------------------------
// Forward-mode dual number (what ceres::Jet<double,N> is).
// -----------------------------------------------------------------------------
template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
    Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) { Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
    Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
    for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv;
    return r;
}
template <int N> inline Jet<N> operator+(const Jet<N>& x, double s) { Jet<N> r = x; r.a += s; return r; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& x) { return x + s; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, double s) { Jet<N> r = x; r.a -= s; return r; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& x) { return (-x) + s; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, double s) { Jet<N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& x) { return x * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, double s) { return x * (1.0 / s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& x) { return Jet<N>(s) / x; }
template <int N> inline Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> inline Jet<N>& operator*=(Jet<N>& x, const Jet<N>& y) { x = x * y; return x; }
template <int N> inline Jet<N> sqrt(const Jet<N>& x) { Jet<N> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <int N> inline Jet<N> sq(const Jet<N>& x) { return x * x; }  // ceres::pow(x, 2)
inline double sq(double x) { return x * x; }
inline double sqrt(double x) { return std::sqrt(x); }
template <int N> inline double scalar(const Jet<N>& x) { return x.a; }
inline double scalar(double x) { return x; }

// -----------------------------------------------------------------------------
// Inputs of one alignment (what Tracker::optimize hands to PhotometricError::Create,
// reference src/tracking/Tracker.cpp:164-191).  All pointers are borrowed.
// -----------------------------------------------------------------------------
struct Problem {
    int N = 0;
    const double* grad = nullptr;        // N x 2 AoS (cv::Point2d), KeyFrame.hpp:80
    const double* norm_coord = nullptr;  // N x 2 AoS (cv::Point2d), KeyFrame.hpp:80
    const double* idp = nullptr;         // N, DepthPoints::getIDepth (DepthPoints.cpp:230-237)
    const double* weights = nullptr;     // N, KeyFrame.hpp:90
    const double* frame = nullptr;       // H*W row-major, EventFrame.hpp:59
    int H = 0, W = 0;
    double fx = 0, fy = 0, cx = 0, cy = 0;
};

// -----------------------------------------------------------------------------
// ceres::Grid2D<double,1>(data, 0, H, 0, W) + ceres::BiCubicInterpolator
// (upstream Ceres cubic_interpolation.h; reference use: PhotometricError.hpp:110-111,172).
// Grid2D::GetValue clamps row/col to [0,H-1] x [0,W-1].
// -----------------------------------------------------------------------------
inline double grid_value(const Problem& pb, int r, int c) {
    const int rr = std::min(std::max(0, r), pb.H - 1);
    const int cc = std::min(std::max(0, c), pb.W - 1);
    return pb.frame[static_cast<size_t>(rr) * pb.W + cc];
}
// Catmull-Rom cubic Hermite spline through p1 (x=0) and p2 (x=1).
inline void cubic_hermite(double p0, double p1, double p2, double p3, double x, double* f, double* dfdx) {
    const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
    const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
    const double c = 0.5 * (-p0 + p2);
    const double d = p1;
    if (f) *f = d + x * (c + x * (b + x * a));
    if (dfdx) *dfdx = c + x * (2.0 * b + 3.0 * a * x);
}
// Evaluate(r, c, f, dfdr, dfdc): row first (PhotometricError.hpp:172 passes (yp, xp)).
inline void bicubic(const Problem& pb, double r, double c, double* f, double* dfdr, double* dfdc) {
    const int row = static_cast<int>(std::floor(r));
    const int col = static_cast<int>(std::floor(c));
    double fr[4], dfr[4];
    for (int k = 0; k < 4; ++k) {
        const int rr = row - 1 + k;
        cubic_hermite(grid_value(pb, rr, col - 1), grid_value(pb, rr, col), grid_value(pb, rr, col + 1),
                      grid_value(pb, rr, col + 2), c - col, &fr[k], &dfr[k]);
    }
    cubic_hermite(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
    if (dfdc) cubic_hermite(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, nullptr);
}
inline void bicubic_eval(const Problem& pb, const double& r, const double& c, double* f) { bicubic(pb, r, c, f, nullptr, nullptr); }
template <int N> inline void bicubic_eval(const Problem& pb, const Jet<N>& r, const Jet<N>& c, Jet<N>* f) {
    double fa, dr, dc;
    bicubic(pb, r.a, c.a, &fa, &dr, &dc);
    f->a = fa;
    for (int i = 0; i < N; ++i) f->v[i] = dr * r.v[i] + dc * c.v[i];
}
// Bilinear sampler with the same clamping (north_star variant; DSO's sampler is
// src/utils/globalFuncs.h:78-92).  Not used by the reference tracker.
inline void bilinear(const Problem& pb, double r, double c, double* f, double* dfdr, double* dfdc) {
    const int row = static_cast<int>(std::floor(r));
    const int col = static_cast<int>(std::floor(c));
    const double dy = r - row, dx = c - col;
    const double tl = grid_value(pb, row, col), tr = grid_value(pb, row, col + 1);
    const double bl = grid_value(pb, row + 1, col), br = grid_value(pb, row + 1, col + 1);
    if (f) *f = (1 - dy) * ((1 - dx) * tl + dx * tr) + dy * ((1 - dx) * bl + dx * br);
    if (dfdc) *dfdc = (1 - dy) * (tr - tl) + dy * (br - bl);
    if (dfdr) *dfdr = (1 - dx) * (bl - tl) + dx * (br - tr);
}
inline void bilinear_eval(const Problem& pb, const double& r, const double& c, double* f) { bilinear(pb, r, c, f, nullptr, nullptr); }
template <int N> inline void bilinear_eval(const Problem& pb, const Jet<N>& r, const Jet<N>& c, Jet<N>* f) {
    double fa, dr, dc;
    bilinear(pb, r.a, c.a, &fa, &dr, &dc);
    f->a = fa;
    for (int i = 0; i < N; ++i) f->v[i] = dr * r.v[i] + dc * c.v[i];
}
enum Sampling { BICUBIC = 0, BILINEAR = 1 };
template <class T> inline void sample(const Problem& pb, int sampling, const T& r, const T& c, T* f) {
    if (sampling == BILINEAR) bilinear_eval(pb, r, c, f); else bicubic_eval(pb, r, c, f);
}

// -----------------------------------------------------------------------------
// Motion field of a static scene under camera velocity v = [lin(3), ang(3)]
// (PhotometricError::compute_flow, PhotometricError.hpp:114-122; scalar twin
// utils/Utils.hpp:165-173).  Uses the RAW inverse depth (no eps).
// -----------------------------------------------------------------------------
template <class T>
inline void compute_flow(double xp, double yp, const T* vx, double idp, T* out) {
    out[0] = (vx[0] * (-idp)) + (vx[2] * (xp * idp)) + (vx[3] * (xp * yp)) - (vx[4] * (1.0 + xp * xp)) + (vx[5] * yp);
    out[1] = (vx[1] * (-idp)) + (vx[2] * (yp * idp)) + (vx[3] * (1.0 + yp * yp)) - (vx[4] * (xp * yp)) - (vx[5] * xp);
}

// Eigen::Quaternion<T>::toRotationMatrix for coefficients stored (x,y,z,w);
// no normalisation (used at PhotometricError.hpp:163).
template <class T>
inline void quat_to_R(const T* q, T R[9]) {
    const T tx = q[0] * 2.0, ty = q[1] * 2.0, tz = q[2] * 2.0;
    const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

static constexpr double kEps = 1e-05;            // PhotometricError.hpp:200
static constexpr double kModelNormSq0 = 1e-03;   // PhotometricError.hpp:132

// -----------------------------------------------------------------------------
// One residual block = PhotometricError::operator() over points [start, start+n)
// (PhotometricError.hpp:124-182; constructor back-projection :95-106).
// nc = true switches to PhotometricErrorNC (PhotometricErrorNC.hpp:124-192).
// -----------------------------------------------------------------------------
template <class T>
inline void photometric_block(const Problem& pb, int sampling, bool nc, int start, int n,
                              const T* px, const T* qx, const T* vx, T* residual) {
    // pass 1: model and its squared norm over this block (:131-149)
    T model_norm_sq(kModelNormSq0);
    for (int i = 0; i < n; ++i) {
        const int idx = start + i;
        T flow[2];
        compute_flow<T>(pb.norm_coord[2 * idx], pb.norm_coord[2 * idx + 1], vx, pb.idp[idx], flow);
        residual[i] = -(flow[0] * pb.grad[2 * idx] + flow[1] * pb.grad[2 * idx + 1]);
        model_norm_sq += sq(residual[i]);
    }
    const T model_norm = sqrt(model_norm_sq);
    T R[9];
    quat_to_R<T>(qx, R);
    std::vector<T> bright(nc ? n : 0);
    T meas_norm_sq(kModelNormSq0);
    // pass 2: warp, project, sample (:152-176)
    for (int i = 0; i < n; ++i) {
        const int idx = start + i;
        const double z = 1.0 / (pb.idp[idx] + kEps);           // :100
        const double X = pb.norm_coord[2 * idx] * z;           // :101
        const double Y = pb.norm_coord[2 * idx + 1] * z;       // :102
        const T p0 = R[0] * X + R[1] * Y + R[2] * z + px[0];   // :163
        const T p1 = R[3] * X + R[4] * Y + R[5] * z + px[1];
        const T p2 = R[6] * X + R[7] * Y + R[8] * z + px[2];
        const T xp = (p0 / p2) * pb.fx + pb.cx;                // :167
        const T yp = (p1 / p2) * pb.fy + pb.cy;                // :168
        T e;
        sample<T>(pb, sampling, yp, xp, &e);                   // :172 (row = yp first)
        if (nc) { bright[i] = e; meas_norm_sq += sq(e); }
        else residual[i] = (residual[i] / model_norm - e) * pb.weights[idx];   // :173
    }
    if (nc) {
        const T meas_norm = sqrt(meas_norm_sq);
        for (int i = 0; i < n; ++i)
            residual[i] = (residual[i] / model_norm - bright[i] / meas_norm) * pb.weights[start + i];
    }
}

// Block partition of Tracker.cpp:178-195: n = N / T each, remainder to the last.
inline void block_range(int N, int num_blocks, int b, int* start, int* n) {
    const int ne = N / num_blocks;
    *start = b * ne;
    *n = ne + ((b + 1 == num_blocks) ? (N - (b + 1) * ne) : 0);
}

// -----------------------------------------------------------------------------
// Local parameterisations (Tracker.cpp:111-114,197-198).
// -----------------------------------------------------------------------------
// ceres::EigenQuaternionParameterization::Plus — q_delta (x) q, storage xyzw.
inline void quat_plus(const double* x, const double* d, double* out) {
    const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (nd > 0.0) {
        const double s = std::sin(nd) / nd;
        const double qd[4] = {s * d[0], s * d[1], s * d[2], std::cos(nd)};   // xyzw
        // Hamilton product qd * x
        const double x1 = qd[0], y1 = qd[1], z1 = qd[2], w1 = qd[3];
        const double x2 = x[0], y2 = x[1], z2 = x[2], w2 = x[3];
        out[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
        out[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
        out[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
        out[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    } else {
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
}
// ...::ComputeJacobian — 4x3 row-major, rows in storage order x,y,z,w.
inline void quat_plus_jacobian(const double* x, double J[12]) {
    J[0] = x[3];  J[1] = x[2];   J[2] = -x[1];
    J[3] = -x[2]; J[4] = x[3];   J[5] = x[0];
    J[6] = x[1];  J[7] = -x[0];  J[8] = x[3];
    J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
}
// UnitNormVectorAddition (PhotometricError.hpp:32-54): (x+d)/||x+d||.
template <class T>
inline void unit_plus(const T* x, const T* d, T* out) {
    T sum(0.0);
    for (int i = 0; i < 6; ++i) { const T s = x[i] + d[i]; sum += s * s; out[i] = s; }
    const T inv = T(1.0) / sqrt(sum);
    for (int i = 0; i < 6; ++i) out[i] = out[i] * inv;
}
// AutoDiffLocalParameterization<UnitNormVectorAddition,6,6>::ComputeJacobian:
// d Plus(x, d)/d d at d = 0, 6x6 row-major, by forward-mode autodiff.
inline void unit_plus_jacobian(const double* x, double J[36]) {
    Jet<6> xj[6], dj[6], out[6];
    for (int i = 0; i < 6; ++i) { xj[i] = Jet<6>(x[i]); dj[i] = Jet<6>(0.0, i); }
    unit_plus<Jet<6>>(xj, dj, out);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) J[6 * i + j] = out[i].v[j];
}

// -----------------------------------------------------------------------------
// Robust loss + Ceres corrector (Tracker.cpp:146-161; upstream loss_function.cc,
// corrector.cc).  Applied to s = ||r_block||^2, one scalar per block.
// -----------------------------------------------------------------------------
enum LossType { LOSS_NONE = 0, LOSS_HUBER = 1, LOSS_CAUCHY = 2 };   // tracking/Config.hpp:36
inline void loss_eval(int type, double a, double s, double rho[3]) {
    const double tiny = std::numeric_limits<double>::min();
    if (type == LOSS_HUBER) {
        const double b = a * a;
        if (s > b) {
            const double r = std::sqrt(s);
            rho[0] = 2.0 * a * r - b;
            rho[1] = std::max(tiny, a / r);
            rho[2] = -rho[1] / (2.0 * s);
        } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    } else if (type == LOSS_CAUCHY) {
        const double b = a * a, c = 1.0 / b;
        const double sum = 1.0 + s * c, inv = 1.0 / sum;
        rho[0] = b * std::log(sum);
        rho[1] = std::max(tiny, inv);
        rho[2] = -c * (inv * inv);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}
struct Corrector {
    double sqrt_rho1, residual_scaling, alpha_sq_norm;
    Corrector(double sq_norm, const double rho[3]) {
        sqrt_rho1 = std::sqrt(rho[1]);
        if (sq_norm == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; return; }
        const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
};

// -----------------------------------------------------------------------------
// Configuration of the reference solve (tracking/Config.hpp:40-58; Tracker.cpp:117-161).
// -----------------------------------------------------------------------------
struct SolveConfig {
    int sampling = BICUBIC;
    bool nc = false;
    int num_blocks = 1;                 // options.num_threads -> residual blocks
    int loss_type = LOSS_NONE;
    double loss_param = 1.0;            // config.loss_params[0]
    int max_num_iterations = 10;        // options.max_num_iterations[id]
    double function_tolerance = 1e-6;   // YAML; Ceres default
    double gradient_tolerance = 1e-8;   // Tracker.cpp:142
    double parameter_tolerance = 1e-6;  // Tracker.cpp:143
    int eval_threads = 1;               // host threads evaluating blocks (Ceres pool)
};

// Full evaluation of the 12-column Ceres problem at (p,q,v).
struct Evaluation {
    double cost = 0.0;                  // 1/2 sum_b rho(s_b)
    std::vector<double> residuals;      // N, loss-corrected
    std::vector<double> raw_residuals;  // N, before the corrector (what :223-230 stores)
    std::vector<double> jac_global;     // N x 13 row-major [p|q|v], before corrector
    std::vector<double> jac_local;      // N x 12 row-major, loss-corrected
    std::vector<double> jac_local_raw;  // N x 12, before the corrector
    double gradient[12];                // J_local^T r (corrected)
    bool ok = true;
};

inline void evaluate_block_jets(const Problem& pb, const SolveConfig& cfg, int start, int n,
                                const double* p, const double* q, const double* v,
                                double* r_out, double* jac_global /* n x 13 or null */) {
    if (n <= 0) return;
    if (!jac_global) {
        std::vector<double> r(n);
        photometric_block<double>(pb, cfg.sampling, cfg.nc, start, n, p, q, v, r.data());
        std::memcpy(r_out, r.data(), sizeof(double) * n);
        return;
    }
    typedef Jet<13> J13;
    J13 pj[3], qj[4], vj[6];
    for (int i = 0; i < 3; ++i) pj[i] = J13(p[i], i);
    for (int i = 0; i < 4; ++i) qj[i] = J13(q[i], 3 + i);
    for (int i = 0; i < 6; ++i) vj[i] = J13(v[i], 7 + i);
    std::vector<J13> r(n);
    photometric_block<J13>(pb, cfg.sampling, cfg.nc, start, n, pj, qj, vj, r.data());
    for (int i = 0; i < n; ++i) {
        r_out[i] = r[i].a;
        for (int k = 0; k < 13; ++k) jac_global[13 * i + k] = r[i].v[k];
    }
}

// Persistent worker pool of the block evaluations.  Ceres keeps a pool of options.num_threads workers for the whole solve
// (Tracker.cpp:138 sets it; the T residual blocks of Tracker.cpp:178-195 are evaluated on it); spawning T std::threads per
// evaluation — what this file did until round 4 — cost more than the evaluation itself from 8 threads up (REF12 on the CPU got
// SLOWER with threads).  One pool per process, grown on demand; the caller takes part in the work.
class EvalPool {
  public:
    static EvalPool& instance() { static EvalPool p; return p; }
    // runs f(b) for b in [0, n) on up to T threads (the caller is one of them)
    void run(int n, int T, const std::function<void(int)>& f) {
        const int helpers = std::min(T, n) - 1;
        if (helpers <= 0) { for (int b = 0; b < n; ++b) f(b); return; }
        std::lock_guard<std::mutex> serial(run_m_);          // one parallel section at a time
        {
            std::lock_guard<std::mutex> lk(m_);
            while ((int)workers_.size() < helpers) { const int id = (int)workers_.size(); workers_.emplace_back([this, id]() { worker(id); }); }
            job_ = &f; n_ = n; next_.store(0, std::memory_order_relaxed); want_.store(helpers, std::memory_order_relaxed);
            active_.store(helpers, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        for (int b; (b = next_.fetch_add(1, std::memory_order_relaxed)) < n;) f(b);
        for (int spin = 0; active_.load(std::memory_order_acquire) != 0; ++spin) { if (spin > 64) std::this_thread::yield(); }
    }
    ~EvalPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }

  private:
    // A worker SPINS for its next section for a while (the evaluations of a solve follow each other within a fraction of a millisecond:
    // a sleeping worker's wake-up costs as much as its share of the work) and only then blocks on the condition variable.
    void worker(int id) {
        unsigned long seen = 0;
        for (;;) {
            int spins = 0;
            while (!stop_.load(std::memory_order_relaxed) && !(gen_.load(std::memory_order_acquire) != seen && id < want_.load(std::memory_order_relaxed))) {
                if (++spins < 40000) { __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1, std::memory_order_release);
                cv_.wait(lk, [&] { return stop_.load() || (gen_.load() != seen && id < want_.load()); });
                sleepers_.fetch_sub(1, std::memory_order_release);
                break;
            }
            if (stop_.load()) return;
            const std::function<void(int)>* job; int n;
            { std::lock_guard<std::mutex> lk(m_); if (gen_.load() == seen || id >= want_.load()) continue; seen = gen_.load(); job = job_; n = n_; }
            for (int b; (b = next_.fetch_add(1, std::memory_order_relaxed)) < n;) (*job)(b);
            active_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_m_;
    std::condition_variable cv_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<int> next_{0}, active_{0}, sleepers_{0};
    std::atomic<unsigned long> gen_{0};
    std::atomic<bool> stop_{false};
    std::atomic<int> want_{0};
    int n_ = 0;
};

inline void evaluate(const Problem& pb, const SolveConfig& cfg, const double* p, const double* q,
                     const double* v, bool want_jacobian, Evaluation* ev) {
    const int N = pb.N, B = std::max(1, cfg.num_blocks);
    ev->residuals.assign(N, 0.0);
    ev->raw_residuals.assign(N, 0.0);
    if (want_jacobian) { ev->jac_global.assign(static_cast<size_t>(N) * 13, 0.0); ev->jac_local.assign(static_cast<size_t>(N) * 12, 0.0); }
    // everything that belongs to one residual block runs on the worker that evaluates the block — functor, chain rule through the local
    // parameterisations, loss and corrector — as Ceres' evaluator does it; only the block costs are added up afterwards, in block order
    double Jq[12], Jv[36];
    quat_plus_jacobian(q, Jq);
    unit_plus_jacobian(v, Jv);
    if (want_jacobian) ev->jac_local_raw.assign(static_cast<size_t>(N) * 12, 0.0);
    std::vector<double> block_cost(B, 0.0);
    std::vector<char> block_ok(B, 1);
    auto run_block = [&](int b) {
        int start, n; block_range(N, B, b, &start, &n);
        evaluate_block_jets(pb, cfg, start, n, p, q, v, ev->raw_residuals.data() + start,
                            want_jacobian ? ev->jac_global.data() + static_cast<size_t>(start) * 13 : nullptr);
        if (want_jacobian) {
            for (int i = start; i < start + n; ++i) {
                const double* g = &ev->jac_global[static_cast<size_t>(i) * 13];
                double* l = &ev->jac_local[static_cast<size_t>(i) * 12];
                for (int k = 0; k < 3; ++k) l[k] = g[k];
                for (int k = 0; k < 3; ++k) { double s = 0; for (int m = 0; m < 4; ++m) s += g[3 + m] * Jq[3 * m + k]; l[3 + k] = s; }
                for (int k = 0; k < 6; ++k) { double s = 0; for (int m = 0; m < 6; ++m) s += g[7 + m] * Jv[6 * m + k]; l[6 + k] = s; }
            }
            std::memcpy(&ev->jac_local_raw[static_cast<size_t>(start) * 12], &ev->jac_local[static_cast<size_t>(start) * 12], sizeof(double) * 12 * n);
        }
        double s = 0.0;
        for (int i = 0; i < n; ++i) { const double r = ev->raw_residuals[start + i]; ev->residuals[start + i] = r; s += r * r; if (!std::isfinite(r)) block_ok[b] = 0; }
        if (cfg.loss_type == LOSS_NONE) { block_cost[b] = 0.5 * s; return; }
        double rho[3];
        loss_eval(cfg.loss_type, cfg.loss_param, s, rho);
        block_cost[b] = 0.5 * rho[0];
        Corrector c(s, rho);
        if (want_jacobian) {
            for (int i = 0; i < n; ++i) {
                double* l = &ev->jac_local[static_cast<size_t>(start + i) * 12];
                if (c.alpha_sq_norm == 0.0) { for (int k = 0; k < 12; ++k) l[k] *= c.sqrt_rho1; }
            }
            if (c.alpha_sq_norm != 0.0) {   // J = sqrt_rho1 (J - alpha_sq_norm r r^T J)
                double rtJ[12] = {0};
                for (int i = 0; i < n; ++i) for (int k = 0; k < 12; ++k) rtJ[k] += ev->raw_residuals[start + i] * ev->jac_local[static_cast<size_t>(start + i) * 12 + k];
                for (int i = 0; i < n; ++i) for (int k = 0; k < 12; ++k) {
                    double& e = ev->jac_local[static_cast<size_t>(start + i) * 12 + k];
                    e = c.sqrt_rho1 * (e - c.alpha_sq_norm * ev->raw_residuals[start + i] * rtJ[k]);
                }
            }
        }
        for (int i = 0; i < n; ++i) ev->residuals[start + i] *= c.residual_scaling;
    };
    if (cfg.eval_threads > 1 && B > 1) {
        const std::function<void(int)> job = run_block;
        EvalPool::instance().run(B, std::min(cfg.eval_threads, B), job);
    } else {
        for (int b = 0; b < B; ++b) run_block(b);
    }
    ev->cost = 0.0;
    ev->ok = true;
    for (int b = 0; b < B; ++b) { ev->cost += block_cost[b]; if (!block_ok[b]) ev->ok = false; }
    if (want_jacobian) {
        for (int k = 0; k < 12; ++k) ev->gradient[k] = 0.0;
        for (int i = 0; i < N; ++i) for (int k = 0; k < 12; ++k) {
            ev->gradient[k] += ev->jac_local[static_cast<size_t>(i) * 12 + k] * ev->residuals[i];
            if (!std::isfinite(ev->jac_local[static_cast<size_t>(i) * 12 + k])) ev->ok = false;
        }
    }
}

// x (+) delta over the three parameter blocks: p additive, q EigenQuaternion, v UnitNorm.
inline void state_plus(const double x[13], const double d[12], double out[13]) {
    for (int i = 0; i < 3; ++i) out[i] = x[i] + d[i];
    quat_plus(x + 3, d + 3, out + 3);
    unit_plus<double>(x + 7, d + 6, out + 7);
}

// Dense symmetric positive-definite solve (Cholesky), n <= 12.  Returns false
// if the matrix is not numerically PD.
inline bool cholesky_solve(int n, const double* A, const double* b, double* x) {
    double L[144];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            if (i == j) { if (!(s > 0.0) || !std::isfinite(s)) return false; L[i * n + i] = std::sqrt(s); }
            else L[i * n + j] = s / L[j * n + j];
        }
    }
    double y[12];
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
    for (int i = 0; i < n; ++i) if (!std::isfinite(x[i])) return false;
    return true;
}

// -----------------------------------------------------------------------------
// ceres::Solve with the reference's options (Tracker.cpp:117-143,202): trust
// region, Levenberg-Marquardt, Jacobi scaling, monotonic steps, Ceres defaults
// initial_trust_region_radius 1e4, max 1e16, min 1e-32, min_relative_decrease
// 1e-3, min/max_lm_diagonal 1e-6/1e32, max_num_consecutive_invalid_steps 5.
// (Restated from upstream trust_region_minimizer.cc / levenberg_marquardt_strategy.cc.)
// -----------------------------------------------------------------------------
enum Termination { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };
struct SolveSummary {
    int termination = FAILURE;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    double initial_cost = 0, final_cost = 0;
    int num_residuals = 0;
    bool usable() const { return termination == CONVERGENCE || termination == NO_CONVERGENCE; }
    int num_iterations() const { return num_successful_steps + num_unsuccessful_steps; }   // Tracker.cpp:211
    std::vector<double> cost_trace;     // cost after each iteration (diagnostics)
};

inline void solve_lm(const Problem& pb, const SolveConfig& cfg, double p[3], double q[4], double v[6], SolveSummary* sum) {
    const int NP = 12;
    double x[13], cand[13], params[13];
    std::memcpy(x, p, 24); std::memcpy(x + 3, q, 32); std::memcpy(x + 7, v, 48);
    std::memcpy(params, x, sizeof(x));
    auto xnorm = [](const double* a) { double s = 0; for (int i = 0; i < 13; ++i) s += a[i] * a[i]; return std::sqrt(s); };
    *sum = SolveSummary();
    sum->num_residuals = pb.N;

    Evaluation ev, evc;
    std::vector<double> J;     // N x 12 scaled jacobian
    double scale[12], grad[12];
    double x_cost = 0, x_norm = xnorm(x);
    double grad_max_norm = 0;
    bool first = true;
    auto eval_grad_jac = [&]() -> bool {
        evaluate(pb, cfg, x, x + 3, x + 7, true, &ev);
        if (!ev.ok) return false;
        x_cost = ev.cost;
        J = ev.jac_local;
        if (first) {
            double ss[12] = {0};
            for (int i = 0; i < pb.N; ++i) for (int k = 0; k < NP; ++k) ss[k] += J[static_cast<size_t>(i) * NP + k] * J[static_cast<size_t>(i) * NP + k];
            for (int k = 0; k < NP; ++k) scale[k] = 1.0 / (1.0 + std::sqrt(ss[k]));
            first = false;
        }
        for (int i = 0; i < pb.N; ++i) for (int k = 0; k < NP; ++k) J[static_cast<size_t>(i) * NP + k] *= scale[k];
        for (int k = 0; k < NP; ++k) grad[k] = ev.gradient[k];
        double ng[12], proj[13];
        for (int k = 0; k < NP; ++k) ng[k] = -grad[k];
        state_plus(x, ng, proj);
        grad_max_norm = 0;
        for (int i = 0; i < 13; ++i) grad_max_norm = std::max(grad_max_norm, std::fabs(x[i] - proj[i]));
        return true;
    };

    // LM strategy state
    double radius = 1e4, decrease_factor = 2.0;
    const double max_radius = 1e16, min_radius = 1e-32, min_diag = 1e-6, max_diag = 1e32;
    bool reuse_diagonal = false;
    double diagonal[12];
    int consecutive_invalid = 0;

    // iteration zero
    if (!eval_grad_jac()) { sum->termination = FAILURE; return; }
    sum->initial_cost = x_cost;
    double minimum_cost = x_cost;
    int iteration = 0;
    bool step_successful = true;
    sum->termination = NO_CONVERGENCE;

    while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (step_successful) {
            ++sum->num_successful_steps;
            if (x_cost < minimum_cost || iteration == 0) { minimum_cost = x_cost; std::memcpy(params, x, sizeof(x)); }
        } else {
            ++sum->num_unsuccessful_steps;
        }
        sum->cost_trace.push_back(x_cost);
        if (iteration >= cfg.max_num_iterations) { sum->termination = NO_CONVERGENCE; break; }
        if (step_successful && grad_max_norm <= cfg.gradient_tolerance) { sum->termination = CONVERGENCE; break; }
        if (radius < min_radius) { sum->termination = CONVERGENCE; break; }

        ++iteration;
        step_successful = false;

        // ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
        if (!reuse_diagonal) {
            double ss[12] = {0};
            for (int i = 0; i < pb.N; ++i) for (int k = 0; k < NP; ++k) ss[k] += J[static_cast<size_t>(i) * NP + k] * J[static_cast<size_t>(i) * NP + k];
            for (int k = 0; k < NP; ++k) diagonal[k] = std::min(std::max(ss[k], min_diag), max_diag);
        }
        // normal equations J^T J, J^T r: one sweep over the rows of J (every entry is still the sum over the points in point order; the
        // column-strided form of round 1 walked the 192 KB of J 156 times and was most of a solve on one core)
        double A[144], g[12], step[12];
        for (int k = 0; k < 144; ++k) A[k] = 0.0;
        for (int k = 0; k < NP; ++k) g[k] = 0.0;
        for (int i = 0; i < pb.N; ++i) {
            const double* Ji = &J[static_cast<size_t>(i) * NP];
            const double ri = ev.residuals[i];
            for (int a = 0; a < NP; ++a) {
                const double ja = Ji[a];
                g[a] += ja * ri;
                for (int b = a; b < NP; ++b) A[a * NP + b] += ja * Ji[b];
            }
        }
        for (int a = 0; a < NP; ++a) {
            for (int b = 0; b < a; ++b) A[a * NP + b] = A[b * NP + a];
            A[a * NP + a] += diagonal[a] / radius;   // lm_diagonal^2
        }
        reuse_diagonal = true;
        bool valid = cholesky_solve(NP, A, g, step);
        double model_cost_change = 0;
        if (valid) {
            for (int k = 0; k < NP; ++k) step[k] = -step[k];
            for (int i = 0; i < pb.N; ++i) {
                double m = 0; for (int k = 0; k < NP; ++k) m += J[static_cast<size_t>(i) * NP + k] * step[k];
                model_cost_change += -m * (ev.residuals[i] + m / 2.0);
            }
            valid = model_cost_change > 0.0;
        }
        if (!valid) {   // HandleInvalidStep
            if (++consecutive_invalid >= 5) { sum->termination = FAILURE; break; }
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            continue;
        }
        consecutive_invalid = 0;
        double delta[12];
        for (int k = 0; k < NP; ++k) delta[k] = step[k] * scale[k];

        // ComputeCandidatePointAndEvaluateCost
        state_plus(x, delta, cand);
        evaluate(pb, cfg, cand, cand + 3, cand + 7, false, &evc);
        const double cand_cost = evc.ok ? evc.cost : std::numeric_limits<double>::max();

        // ParameterToleranceReached
        double step_norm = 0; for (int i = 0; i < 13; ++i) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
        step_norm = std::sqrt(step_norm);
        if (step_norm <= cfg.parameter_tolerance * (x_norm + cfg.parameter_tolerance)) { sum->termination = CONVERGENCE; break; }
        // FunctionToleranceReached
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= cfg.function_tolerance * x_cost) { sum->termination = CONVERGENCE; break; }
        // IsStepSuccessful
        const double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > 1e-3) {   // HandleSuccessfulStep
            std::memcpy(x, cand, sizeof(x));
            x_norm = xnorm(x);
            if (!eval_grad_jac()) { sum->termination = FAILURE; break; }
            step_successful = true;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
            radius = std::min(max_radius, radius);
            decrease_factor = 2.0;
            reuse_diagonal = false;
        } else {                          // HandleUnsuccessfulStep
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
        }
    }
    sum->final_cost = minimum_cost;
    if (sum->usable()) { std::memcpy(p, params, 24); std::memcpy(q, params + 3, 32); std::memcpy(v, params + 7, 48); }
}

// -----------------------------------------------------------------------------
// Adaptive loss scale (Tracker::getLossParams, Tracker.cpp:281-317) with the
// helpers' quirks: n_quantile_vector reorders its argument in place
// (utils/Utils.hpp:315-320); mean_std_vector returns the VARIANCE (:272-290).
// -----------------------------------------------------------------------------
enum LossParamMethod { LP_CONSTANT = 0, LP_MAD = 1, LP_STD = 2 };   // Tracker.hpp:34
inline double n_quantile(std::vector<double>& vec, int n) { std::nth_element(vec.begin(), vec.begin() + n, vec.end()); return vec[n]; }
inline double loss_param(std::vector<double>& residuals, int method, double current) {
    if (method == LP_MAD) {
        const double median = n_quantile(residuals, static_cast<int>(residuals.size() / 2));
        std::vector<double> abs_med; abs_med.reserve(residuals.size());
        for (double r : residuals) abs_med.push_back(std::fabs(r - median));
        const double mad = 1.4826 * n_quantile(abs_med, static_cast<int>(abs_med.size() / 2));
        return 1.345 * mad;
    }
    if (method == LP_STD) {
        const size_t sz = residuals.size();
        if (sz == 1) return 0.0;
        double mu = 0; for (double r : residuals) mu += r; mu /= sz;
        double var = 0; for (double r : residuals) var += (r - mu) * (r - mu) / (sz - 1);
        return 1.345 * var;
    }
    return current;
}

// -----------------------------------------------------------------------------
// Sophus-compatible SE3 exp / log (reference src/sophus/se3.hpp:406-428,559-586;
// so3.hpp:343-369,491-530; epsilon 1e-10, sophus.hpp:45-47).  Tangent = [upsilon; omega].
// Pose = translation t[3] + unit quaternion q (x,y,z,w).
// -----------------------------------------------------------------------------
static constexpr double kSophusEps = 1e-10;
inline void quat_mul(const double* a, const double* b, double* out) {   // xyzw Hamilton a*b
    const double x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    double o[4];
    o[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    o[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    o[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
    o[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    for (int i = 0; i < 4; ++i) out[i] = o[i];
}
inline void so3_exp(const double* omega, double* q, double* theta_out) {
    const double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
    const double theta = std::sqrt(theta_sq), half = 0.5 * theta;
    double imag, real;
    if (theta < kSophusEps) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * t4;
    } else { imag = std::sin(half) / theta; real = std::cos(half); }
    q[0] = imag * omega[0]; q[1] = imag * omega[1]; q[2] = imag * omega[2]; q[3] = real;
    if (theta_out) *theta_out = theta;
}
inline void mat3_mul_vec(const double* M, const double* v, double* o) { double t[3]; for (int i = 0; i < 3; ++i) t[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2]; for (int i = 0; i < 3; ++i) o[i] = t[i]; }
inline void hat(const double* w, double* O) { O[0] = 0; O[1] = -w[2]; O[2] = w[1]; O[3] = w[2]; O[4] = 0; O[5] = -w[0]; O[6] = -w[1]; O[7] = w[0]; O[8] = 0; }
inline void mat3_mul(const double* A, const double* B, double* C) { double t[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { t[3 * i + j] = 0; for (int k = 0; k < 3; ++k) t[3 * i + j] += A[3 * i + k] * B[3 * k + j]; } for (int i = 0; i < 9; ++i) C[i] = t[i]; }
inline void se3_exp(const double* xi, double* t, double* q) {
    const double* ups = xi; const double* omega = xi + 3;
    double theta; so3_exp(omega, q, &theta);
    double Om[9], Om2[9], V[9];
    hat(omega, Om); mat3_mul(Om, Om, Om2);
    if (theta < kSophusEps) { quat_to_R<double>(q, V); }
    else {
        const double tsq = theta * theta;
        const double c1 = (1.0 - std::cos(theta)) / tsq, c2 = (theta - std::sin(theta)) / (tsq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * Om[i] + c2 * Om2[i];
    }
    mat3_mul_vec(V, ups, t);
}
inline void so3_log(const double* q, double* omega, double* theta_out) {
    const double sn = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double n = std::sqrt(sn), w = q[3];
    double f;
    if (n < kSophusEps) { f = 2.0 / w - 2.0 * sn / (w * w * w); }
    else if (std::fabs(w) < kSophusEps) { f = (w > 0 ? M_PI : -M_PI) / n; }
    else { f = 2.0 * std::atan(n / w) / n; }
    if (theta_out) *theta_out = f * n;
    for (int i = 0; i < 3; ++i) omega[i] = f * q[i];
}
inline void se3_log(const double* t, const double* q, double* xi) {
    double theta; so3_log(q, xi + 3, &theta);
    double Om[9], Om2[9], Vi[9];
    hat(xi + 3, Om); mat3_mul(Om, Om, Om2);
    const double c = (std::fabs(theta) < kSophusEps) ? (1.0 / 12.0)
                   : (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
    for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * Om[i] + c * Om2[i];
    mat3_mul_vec(Vi, t, xi);
}
// T <- exp(xi) * T   (left multiplication, as DSO's CoarseTracker.cpp:594 does)
inline void se3_left_update(const double* xi, double* t, double* q) {
    double dt[3], dq[4], R[9], rt[3];
    se3_exp(xi, dt, dq);
    quat_to_R<double>(dq, R);
    mat3_mul_vec(R, t, rt);
    for (int i = 0; i < 3; ++i) t[i] = rt[i] + dt[i];
    quat_mul(dq, q, q);
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= nn;
}
// || log( A * B^-1 ) ||, the SE(3) discrepancy used by the parity statement.
inline double se3_distance(const double* ta, const double* qa, const double* tb, const double* qb) {
    // B^-1 = (R_b^T, -R_b^T t_b)
    double qbi[4] = {-qb[0], -qb[1], -qb[2], qb[3]}, Rbi[9], Ra[9], tmp[3], t[3], q[4], xi[6];
    quat_to_R<double>(qbi, Rbi); quat_to_R<double>(qa, Ra);
    mat3_mul_vec(Rbi, tb, tmp);              // R_b^T t_b
    mat3_mul_vec(Ra, tmp, t);                // R_a R_b^T t_b
    for (int i = 0; i < 3; ++i) t[i] = ta[i] - t[i];
    quat_mul(qa, qbi, q);
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    se3_log(t, q, xi);
    double s = 0; for (int i = 0; i < 6; ++i) s += xi[i] * xi[i];
    return std::sqrt(s);
}

// -----------------------------------------------------------------------------
// Pose-only (6-DoF) view of the same residual, the north_star sub-problem:
// velocity held fixed, T <- exp(xi) T.  The Jacobian row is obtained by
// forward-mode autodiff of projection + sampling w.r.t. xi at xi = 0, where
// P(xi) = P + upsilon + omega x P to first order (Sophus tangent order
// [upsilon; omega], se3.hpp:406-428).  Optional per-point Huber weight is an
// extension patterned on DSO's CoarseTracker.cpp:445 (hw = |r|<tau ? 1 : tau/|r|).
// -----------------------------------------------------------------------------
struct Pose6Eval {
    std::vector<double> r;       // N (includes the point weight w_i)
    std::vector<double> J;       // N x 6 row-major
    std::vector<double> hw;      // N Huber weights (1 if disabled)
    double H[36];                // sum hw J^T J
    double b[6];                 // sum hw J^T r
    double cost;                 // sum hw r^2 (2 - hw)   (= sum r^2 without Huber)
};
inline void pose6_eval(const Problem& pb, const SolveConfig& cfg, const double* p, const double* q,
                       const double* v, double huber_tau, Pose6Eval* out) {
    const int N = pb.N, B = std::max(1, cfg.num_blocks);
    out->r.assign(N, 0); out->J.assign(static_cast<size_t>(N) * 6, 0); out->hw.assign(N, 1.0);
    double R[9]; quat_to_R<double>(q, R);
    typedef Jet<6> J6;
    for (int b = 0; b < B; ++b) {
        int start, n; block_range(N, B, b, &start, &n);
        double S = kModelNormSq0;
        std::vector<double> m(n);
        for (int i = 0; i < n; ++i) {
            const int idx = start + i; double fl[2];
            compute_flow<double>(pb.norm_coord[2 * idx], pb.norm_coord[2 * idx + 1], v, pb.idp[idx], fl);
            m[i] = -(fl[0] * pb.grad[2 * idx] + fl[1] * pb.grad[2 * idx + 1]);
            S += m[i] * m[i];
        }
        const double nrm = std::sqrt(S);
        for (int i = 0; i < n; ++i) {
            const int idx = start + i;
            const double z = 1.0 / (pb.idp[idx] + kEps), X = pb.norm_coord[2 * idx] * z, Y = pb.norm_coord[2 * idx + 1] * z;
            const double P[3] = {R[0] * X + R[1] * Y + R[2] * z + p[0], R[3] * X + R[4] * Y + R[5] * z + p[1], R[6] * X + R[7] * Y + R[8] * z + p[2]};
            J6 Pj[3];
            for (int k = 0; k < 3; ++k) { Pj[k] = J6(P[k]); Pj[k].v[k] = 1.0; }
            // d(omega x P)/d omega = -[P]x
            Pj[0].v[4] = P[2];  Pj[0].v[5] = -P[1];
            Pj[1].v[3] = -P[2]; Pj[1].v[5] = P[0];
            Pj[2].v[3] = P[1];  Pj[2].v[4] = -P[0];
            const J6 xp = (Pj[0] / Pj[2]) * pb.fx + pb.cx;
            const J6 yp = (Pj[1] / Pj[2]) * pb.fy + pb.cy;
            J6 e; sample<J6>(pb, cfg.sampling, yp, xp, &e);
            const J6 res = (J6(m[i] / nrm) - e) * pb.weights[idx];
            out->r[idx] = res.a;
            for (int k = 0; k < 6; ++k) out->J[static_cast<size_t>(idx) * 6 + k] = res.v[k];
        }
    }
    for (int i = 0; i < 36; ++i) out->H[i] = 0; for (int i = 0; i < 6; ++i) out->b[i] = 0;
    out->cost = 0;
    for (int i = 0; i < N; ++i) {
        const double r = out->r[i];
        double hw = 1.0;
        if (huber_tau > 0.0 && std::fabs(r) > huber_tau) hw = huber_tau / std::fabs(r);
        out->hw[i] = hw;
        out->cost += hw * r * r * (2.0 - hw);
        const double* Ji = &out->J[static_cast<size_t>(i) * 6];
        for (int a = 0; a < 6; ++a) { out->b[a] += hw * Ji[a] * r; for (int c = 0; c < 6; ++c) out->H[6 * a + c] += hw * Ji[a] * Ji[c]; }
    }
}
// One Gauss-Newton increment xi = -H^-1 b (dense Cholesky in fp64).
inline bool pose6_step(const Pose6Eval& ev, double xi[6]) {
    double nb[6]; for (int i = 0; i < 6; ++i) nb[i] = -ev.b[i];
    return cholesky_solve(6, ev.H, nb, xi);
}
// `iters` Gauss-Newton iterations; records every increment (iters x 6) and the
// cost before each step (iters).  Returns the number of iterations executed.
inline int pose6_gauss_newton(const Problem& pb, const SolveConfig& cfg, double p[3], double q[4], const double v[6],
                              double huber_tau, int iters, double* increments, double* costs) {
    Pose6Eval ev;
    int it = 0;
    for (; it < iters; ++it) {
        pose6_eval(pb, cfg, p, q, v, huber_tau, &ev);
        double xi[6];
        if (costs) costs[it] = ev.cost;
        if (!pose6_step(ev, xi)) break;
        if (increments) std::memcpy(increments + 6 * it, xi, sizeof(xi));
        se3_left_update(xi, p, q);
    }
    return it;
}

// DSO-style damped Gauss-Newton (Levenberg-Marquardt) on the 6-DoF problem, the
// in-repo template being CoarseTracker::trackNewestCoarse (reference
// src/tracking/CoarseTracker.cpp:545-664): solve (H with diag *= 1+lambda) xi = -b,
// T' = exp(xi) T, accept iff the cost drops (then lambda *= 0.5, re-linearise at T'),
// else lambda *= 4 and retry from the same linearisation.  One residual/Jacobian
// pass per iteration.  Outputs per iteration: increment tried, cost at the
// candidate, accepted flag.  Returns iterations executed.
inline int pose6_lm(const Problem& pb, const SolveConfig& cfg, double p[3], double q[4], const double v[6],
                    double huber_tau, int iters, double lambda0, double* increments, double* costs, int* accepted,
                    double* initial_cost) {
    Pose6Eval cur, cand;
    pose6_eval(pb, cfg, p, q, v, huber_tau, &cur);
    if (initial_cost) *initial_cost = cur.cost;
    double lambda = lambda0;
    int it = 0;
    for (; it < iters; ++it) {
        double Hl[36], nb[6], xi[6];
        std::memcpy(Hl, cur.H, sizeof(Hl));
        for (int i = 0; i < 6; ++i) { Hl[7 * i] *= (1.0 + lambda); nb[i] = -cur.b[i]; }
        if (!cholesky_solve(6, Hl, nb, xi)) break;
        double pc[3], qc[4];
        std::memcpy(pc, p, sizeof(pc)); std::memcpy(qc, q, sizeof(qc));
        se3_left_update(xi, pc, qc);
        pose6_eval(pb, cfg, pc, qc, v, huber_tau, &cand);
        const bool ok = cand.cost < cur.cost;
        if (increments) std::memcpy(increments + 6 * it, xi, sizeof(xi));
        if (costs) costs[it] = cand.cost;
        if (accepted) accepted[it] = ok ? 1 : 0;
        if (ok) { std::memcpy(p, pc, sizeof(pc)); std::memcpy(q, qc, sizeof(qc)); std::swap(cur, cand); lambda *= 0.5; }
        else { lambda *= 4.0; if (lambda < 1e-6) lambda = 1e-6; }
    }
    return it;
}

}  // namespace eds_oracle
