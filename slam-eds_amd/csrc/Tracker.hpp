// C++ shim with the public surface of eds::tracking::Tracker (reference src/tracking/Tracker.hpp:36-114)
// for the event-to-model alignment path, implemented on the C ABI of libeds_hip.so (include/eds_hip.h).
//
// Inside EDS (Eigen, OpenCV and Rock base-types available) define EDS_HIP_WITH_EDS_TYPES before including
// this header: the shim then uses Eigen::Vector3d / Eigen::Quaterniond / Eigen::Matrix<double,6,1> /
// base::Transform3d / base::Vector6d / eds::tracking::KeyFrame / eds::tracking::Config directly, and an external
// component relinks unchanged.  Without it (this repository's own tests: none of those libraries
// exist in the build image) the minimal stand-ins below provide the same type NAMES, template shapes and memory layouts.
//
// Every mirrored member has EXACTLY the reference's signature (return type, parameter types, cv-qualification):
// tests/cpp/shim_eds_types_check.cpp takes each one as a pointer-to-member of the reference type, so drift breaks the build.
//   Tracker(kf, config) :62   Tracker(config) :65   reset(kf, px, qx, keep_velo) :67   reset(kf, px, qx, velo) :69
//   set :71   optimize x3 :73-81   getTransform() :83   getTransform(bool&) :85   getVelocity() -> Matrix<double,6,1>& :87
//   linearVelocity :89   angularVelocity :91   getLossParams :93   getCoord :96   getInfo :109   needNewKeyframe :113
//   public `config` :40; private kf, px, qx, vx, info, poses, squared_norm_flow :44-58
// Outside the hot path (reference Tracker.cpp:378-648): trackPoints, trackPointsPyr, trackPointsAlongEpiline, getEMatrix, getFMatrix and
// the public getFilteredPose.  They are KLT / epipolar helpers on cv::Mat and Sophus types and stay the REFERENCE's code:
//   * define EDS_HIP_REFERENCE_MEMBERS (with EDS_HIP_WITH_EDS_TYPES) and this class DECLARES the six with the reference's exact
//     signatures (Tracker.hpp:98-111), keeps `poses` as std::vector<eds::SE3> (:55) and lets getTransform(bool&) call the reference's
//     getFilteredPose (Tracker.cpp:251-260) — an EDS tree then compiles its own Tracker.cpp:378-648 definitions of those six in a
//     translation unit of its own against THIS header, unchanged (they only touch kf, px, qx, vx, poses, squared_norm_flow and public
//     members: all present under the reference's names), and every caller of the class relinks unchanged;
//     tests/cpp/shim_eds_types_check.cpp pins the six signatures and tests/cpp/shim_reference_members.cpp is such a translation unit
//     (against mock types);
//   * without it the six are absent and the mean filter that getTransform(bool&) needs is restated privately on plain arrays.
// Error convention: the reference path has no exceptions — optimize returns false, getCoord an empty vector; the shim does the same for
// every failure of the library underneath (no device, out of memory, a HIP error) and keeps the status and message for the caller who
// wants to know why: hipLastStatus() / hipLastError().
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/eds_hip.h"

#if defined(EDS_HIP_REFERENCE_MEMBERS) && !defined(EDS_HIP_WITH_EDS_TYPES)
#error "EDS_HIP_REFERENCE_MEMBERS declares members on cv::Mat and eds::SE3: it needs EDS_HIP_WITH_EDS_TYPES"
#endif
#ifdef EDS_HIP_WITH_EDS_TYPES
#include <eds/tracking/Config.hpp>
#include <eds/tracking/KeyFrame.hpp>
#include <eds/tracking/Types.hpp>
#else
// ---- minimal stand-ins (same names, same template shapes, same data layout as the reference types they replace) -------
namespace Eigen {
enum { DontAlign = 0x2 };
template <class T, int R, int C, int Options = 0>
struct Matrix {                              // column vector / small dense matrix, column-major like Eigen
    T v[R * C];
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    T* data() { return v; }
    const T* data() const { return v; }
    static Matrix Zero() { Matrix m; for (int i = 0; i < R * C; ++i) m.v[i] = T(0); return m; }
};
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
struct Quaterniond {                         // coeffs() order x,y,z,w like Eigen
    double c[4];
    static Quaterniond Identity() { return Quaterniond{{0, 0, 0, 1}}; }
    double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
    double* coeffs() { return c; } const double* coeffs() const { return c; }
};
}  // namespace Eigen
namespace base {
struct Time { int64_t microseconds = 0; };
typedef Eigen::Matrix<double, 6, 1, Eigen::DontAlign> Vector6d;         // Rock base/Eigen.hpp
// Eigen::Transform<double,3,Isometry> stand-in: column-major 4x4 like Eigen's matrix()
struct Transform3d {
    double m[16];
    static Transform3d Identity() { Transform3d t; for (int i = 0; i < 16; ++i) t.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return t; }
    double& operator()(int r, int c) { return m[4 * c + r]; }
    double operator()(int r, int c) const { return m[4 * c + r]; }
    Transform3d inverse() const {           // rigid inverse
        Transform3d o = Identity();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = (*this)(c, r);
        for (int r = 0; r < 3; ++r) o(r, 3) = -(o(r, 0) * (*this)(0, 3) + o(r, 1) * (*this)(1, 3) + o(r, 2) * (*this)(2, 3));
        return o;
    }
};
}  // namespace base
namespace cv { struct Point2d { double x, y; }; }
namespace eds { namespace tracking {
enum LOSS_FUNCTION { NONE, HUBER, CAUCHY };                                         // tracking/Config.hpp:36
enum LINEAR_SOLVER_TYPE { DENSE_QR, DENSE_SCHUR, SPARSE_SCHUR, SPARSE_NORMAL_CHOLESKY };
enum BOOTSTRAP_TYPE { EIGHT_POINTS, MiDAS };
struct SolverOptions {                                                              // tracking/Config.hpp:40-47
    LINEAR_SOLVER_TYPE linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    int num_threads = 1;
    std::vector<int> max_num_iterations{10};
    double function_tolerance = 1e-6;
    bool minimizer_progress_to_stdout = false;
};
struct Config {                                                                     // tracking/Config.hpp:49-58
    double percent_points = 0.0;
    std::string type = "ceres";
    LOSS_FUNCTION loss_type = NONE;
    std::vector<double> loss_params{1.0};
    SolverOptions options;
    BOOTSTRAP_TYPE bootstrap = EIGHT_POINTS;
};
struct TrackerInfo {                                                                // tracking/Config.hpp:60-68
    base::Time time; double meas_time_us = 0; uint32_t num_points = 0; int num_iterations = 0; double time_seconds = 0; uint8_t success = 0;
};
// the members of eds::tracking::KeyFrame the tracker touches (KeyFrame.hpp:60-96)
struct KeyFrame {
    std::vector<cv::Point2d> norm_coord, grad;
    std::vector<cv::Point2d> coord;         // pixel coordinates (KeyFrame.hpp:80); optional here, kept index-aligned if present
    std::vector<Eigen::Vector2d> tracks;    // KeyFrame.hpp:92; filled by Tracker::getCoord
    std::vector<double> weights, residuals;
    std::vector<double> inv_depth;          // stand-in for DepthPoints::getIDepth()
    double K_ref[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // row-major 3x3
    int rows = 0, cols = 0;                 // kf->img.rows / cols
};
}}  // namespace eds::tracking
#endif  // EDS_HIP_WITH_EDS_TYPES

namespace eds { namespace tracking {

enum LOSS_PARAM_METHOD { CONSTANT, MAD, STD };                                      // Tracker.hpp:34

/** Extra switches of the GPU tracker (not part of the reference Config). */
struct HipOptions {
    int device = 0;
    int solver = EDS_SOLVER_REF12;          // REF12 reproduces the reference problem; GN6 / LM6 are pose-only
    int sampling = EDS_SAMPLE_BICUBIC;
    int exec = EDS_EXEC_DEVICE;
    double huber_tau = 0.0, lambda0 = 0.01;
    bool nc = false;                        // PhotometricErrorNC instead of PhotometricError (Tracker.cpp:25-27 toggle)
    bool reuse_uploads = true;              // skip the keyframe upload when the KeyFrame vectors are byte-identical to the last call's
};

namespace hipshim {                          // Sophus-compatible SE(3) exp / log on plain arrays (reference src/sophus/se3.hpp:406-428,
                                             // 559-586, so3.hpp:343-369,430-466; epsilon 1e-10): only getTransform(bool&) needs them
inline void quat_mul(const double* a, const double* b, double* o) {                 // xyzw
    const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    const double z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3], w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
inline void so3_log(const double* q, double* om, double* theta) {                  // SO3Group::logAndTheta (so3.hpp:491-531)
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], n = std::sqrt(n2), w = q[3];
    double two_atan_by_n;
    if (n < 1e-10) two_atan_by_n = 2.0 / w - 2.0 * n2 / (w * w * w);
    else if (std::fabs(w) < 1e-10) two_atan_by_n = (w > 0 ? M_PI : -M_PI) / n;
    else two_atan_by_n = 2.0 * std::atan(n / w) / n;
    *theta = two_atan_by_n * n;
    for (int i = 0; i < 3; ++i) om[i] = two_atan_by_n * q[i];
}
inline void hat_sq(const double* o, double* W, double* W2) {
    const double w[9] = {0, -o[2], o[1], o[2], 0, -o[0], -o[1], o[0], 0};
    for (int i = 0; i < 9; ++i) W[i] = w[i];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += w[3 * r + k] * w[3 * k + c]; W2[3 * r + c] = s; }
}
inline void se3_log(const double* t, const double* q, double* xi) {                 // xi = [upsilon, omega]
    double om[3], th, W[9], W2[9], Vi[9];
    so3_log(q, om, &th);
    hat_sq(om, W, W2);
    if (std::fabs(th) < 1e-10) { for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * W[i] + (1.0 / 12.0) * W2[i]; }
    else {
        const double c = (1.0 - th / (2.0 * std::tan(0.5 * th))) / (th * th);
        for (int i = 0; i < 9; ++i) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * W[i] + c * W2[i];
    }
    for (int r = 0; r < 3; ++r) xi[r] = Vi[3 * r] * t[0] + Vi[3 * r + 1] * t[1] + Vi[3 * r + 2] * t[2];
    for (int i = 0; i < 3; ++i) xi[3 + i] = om[i];
}
inline void se3_exp(const double* xi, double* t, double* q) {
    const double* om = xi + 3;
    const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], th = std::sqrt(th2);
    double imag, real, W[9], W2[9], V[9];
    if (th < 1e-10) { const double th4 = th2 * th2; imag = 0.5 - th2 / 48.0 + th4 / 3840.0; real = 1.0 - th2 / 8.0 + th4 / 384.0; }
    else { imag = std::sin(0.5 * th) / th; real = std::cos(0.5 * th); }
    q[0] = imag * om[0]; q[1] = imag * om[1]; q[2] = imag * om[2]; q[3] = real;
    hat_sq(om, W, W2);
    if (th < 1e-10) {                        // Sophus: V = R below epsilon
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                             2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
        for (int i = 0; i < 9; ++i) V[i] = R[i];
    } else {
        const double a = (1.0 - std::cos(th)) / th2, b = (th - std::sin(th)) / (th2 * th);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * W[i] + b * W2[i];
    }
    for (int r = 0; r < 3; ++r) t[r] = V[3 * r] * xi[0] + V[3 * r + 1] * xi[1] + V[3 * r + 2] * xi[2];
}
}  // namespace hipshim

class Tracker {
  public:
    /** Configuration **/
    ::eds::tracking::Config config;         // public like the reference (Tracker.hpp:40)
    HipOptions hip;                         // extension

  private:
    std::shared_ptr<eds::tracking::KeyFrame> kf;                                    // Tracker.hpp:44
    Eigen::Vector3d px;                                                             // :47
    Eigen::Quaterniond qx;                                                          // :48
    Eigen::Matrix<double, 6, 1> vx;                                                 // :49
    eds::tracking::TrackerInfo info;                                                // :52
#ifdef EDS_HIP_REFERENCE_MEMBERS
    std::vector<eds::SE3> poses;                                                    // :55 — what the reference's getFilteredPose reads
#else
    std::vector<std::array<double, 7>> poses;   // :55 (eds::SE3 there): t[3], q xyzw[4] of every getTransform(bool&) call
#endif
    double squared_norm_flow = 0.0;                                                 // :58

    int last_status = EDS_OK;                   // of the last call that reached the library (hipLastStatus / hipLastError)
    std::string last_error;
    bool fail_(int rc, const char* where) { last_status = rc; last_error = std::string(where) + ": " + eds_last_error(); return false; }

    eds_trk* h = nullptr;
    int h_cap = 0, h_rows = 0, h_cols = 0;
    // what the device copy of the keyframe was made from (hip.reuse_uploads)
    std::vector<double> up_points, up_idp;      // norm_coord | grad | weights, then idp, as uploaded last
    double up_K[4] = {0, 0, 0, 0};
    bool device_kf_valid = false;

    void seed_velocity() { const double c = 1.0 / std::sqrt(6.0); for (int i = 0; i < 6; ++i) vx[i] = c; }   // Tracker.cpp:45-46
    eds_trk_cfg make_cfg() const {
        eds_trk_cfg c; eds_trk_cfg_default(&c);
        c.device = hip.device; c.solver = hip.solver; c.sampling = hip.sampling; c.exec = hip.exec;
        c.huber_tau = hip.huber_tau; c.lambda0 = hip.lambda0; c.nc = hip.nc ? 1 : 0;
        c.num_blocks = std::max(1, config.options.num_threads);
        c.loss_type = (int)config.loss_type;
        c.loss_param = config.loss_params.empty() ? 1.0 : config.loss_params[0];
        c.num_levels = (int)std::min<size_t>(EDS_MAX_LEVELS, config.options.max_num_iterations.size());
        for (int i = 0; i < EDS_MAX_LEVELS; ++i)
            c.max_num_iterations[i] = config.options.max_num_iterations.empty() ? 10
                : config.options.max_num_iterations[std::min<size_t>(i, config.options.max_num_iterations.size() - 1)];
        c.function_tolerance = config.options.function_tolerance;
        return c;
    }
    bool ensure_handle(int N, int rows, int cols) {
        eds_trk_cfg c = make_cfg();
        int rc;
        if (h && (h_cap < N || h_rows != rows || h_cols != cols)) { eds_trk_destroy(h); h = nullptr; }
        if (!h) {
            h_cap = std::max(N, 2048); h_rows = rows; h_cols = cols; device_kf_valid = false;
            if ((rc = eds_trk_create(&c, 1, h_cap, rows, cols, &h)) != EDS_OK) { h = nullptr; return fail_(rc, "eds_trk_create"); }
        } else {
            eds_trk_cfg cur;
            if (eds_trk_get_config(h, &cur) != EDS_OK || std::memcmp(&cur, &c, sizeof(c)) != 0)
                if ((rc = eds_trk_set_config(h, &c)) != EDS_OK) return fail_(rc, "eds_trk_set_config");
        }
        return true;
    }
    // The reference hands raw pointers to the functor on every call (Tracker.cpp:189-191) and re-reads the inverse depths
    // (Tracker.cpp:167).  Uploading 9 planes per call is the dominant cost of a live call, so the device copy is kept while the
    // KeyFrame vectors are byte-identical to what it was made from (one 100 KB memcmp); when only the inverse depths moved
    // (DepthPoints update between slices) only that plane goes up.
    bool upload_keyframe(int N, const std::vector<double>& idp, double fx, double fy, double cx, double cy) {
        const size_t n2 = 2 * (size_t)N;
        const double K[4] = {fx, fy, cx, cy};
        const double* nc = &kf->norm_coord[0].x; const double* gr = &kf->grad[0].x; const double* w = kf->weights.data();
        bool same_points = hip.reuse_uploads && device_kf_valid && up_points.size() == 5 * (size_t)N && std::memcmp(up_K, K, sizeof(K)) == 0 &&
                           std::memcmp(up_points.data(), nc, n2 * 8) == 0 && std::memcmp(up_points.data() + n2, gr, n2 * 8) == 0 &&
                           std::memcmp(up_points.data() + 2 * n2, w, (size_t)N * 8) == 0;
        int rc = EDS_OK;
        if (!same_points) {
            rc = eds_trk_set_keyframe(h, 0, N, nc, gr, idp.data(), w, fx, fy, cx, cy);
            up_points.resize(5 * (size_t)N);
            std::memcpy(up_points.data(), nc, n2 * 8); std::memcpy(up_points.data() + n2, gr, n2 * 8); std::memcpy(up_points.data() + 2 * n2, w, (size_t)N * 8);
            std::memcpy(up_K, K, sizeof(K));
            up_idp = idp;
        } else if (up_idp.size() != idp.size() || std::memcmp(up_idp.data(), idp.data(), idp.size() * 8) != 0) {
            rc = eds_trk_set_idepth(h, 0, N, idp.data());
            up_idp = idp;
        }
        device_kf_valid = (rc == EDS_OK);
        return rc == EDS_OK ? true : fail_(rc, "eds_trk_set_keyframe");
    }
    void current_pose(double* t, double* q) const {
        const double n = std::sqrt(qx.x() * qx.x() + qx.y() * qx.y() + qx.z() * qx.z() + qx.w() * qx.w());   // Sophus::SE3(q, t) normalises
        q[0] = qx.x() / n; q[1] = qx.y() / n; q[2] = qx.z() / n; q[3] = qx.w() / n;
        for (int i = 0; i < 3; ++i) t[i] = px[i];
    }
    static base::Transform3d to_transform(const double* t, const double* q) {
        base::Transform3d T = base::Transform3d::Identity();
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - z * w);     T(0, 2) = 2 * (x * z + y * w);
        T(1, 0) = 2 * (x * y + z * w);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - x * w);
        T(2, 0) = 2 * (x * z - y * w);     T(2, 1) = 2 * (y * z + x * w);     T(2, 2) = 1 - 2 * (x * x + y * y);
        for (int i = 0; i < 3; ++i) T(i, 3) = t[i];
        return T;
    }
#ifndef EDS_HIP_REFERENCE_MEMBERS
    // Tracker::getFilteredPose (Tracker.cpp:592-648) on the pose history, mean_filter_size = 3 (its default): the mean of the last
    // poses in the Lie algebra relative to the oldest rotation.  The reference accumulates into a function-static Vector6d that
    // is never reset (Tracker.cpp:611) — every call adds to what all earlier calls (of every Tracker) left there; the same
    // accumulator semantics are kept here so that a relinked component sees the same numbers.
    static double* filter_accumulator() { static double P[6] = {0, 0, 0, 0, 0, 0}; return P; }
    bool filtered_pose(double* t, double* q, size_t mean_filter_size = 3) {
        if (mean_filter_size < 2) { std::memcpy(t, poses.back().data(), 24); std::memcpy(q, poses.back().data() + 3, 32); return true; }
        if (poses.size() < mean_filter_size) return false;
        const size_t n = std::min(poses.size(), mean_filter_size), first = poses.size() - n;
        double* P = filter_accumulator();
        const double* q0 = poses[first].data() + 3;
        const double q0_inv[4] = {-q0[0], -q0[1], -q0[2], q0[3]};
        for (size_t i = first; i != poses.size(); ++i) {
            double q_inc[4], xi[6];
            hipshim::quat_mul(q0_inv, poses[i].data() + 3, q_inc);
            hipshim::se3_log(poses[i].data(), q_inc, xi);
            for (int k = 0; k < 6; ++k) P[k] += xi[k];
        }
        for (int k = 0; k < 6; ++k) P[k] /= (double)n;
        double tq[4];
        hipshim::se3_exp(P, t, tq);
        hipshim::quat_mul(q0, tq, q);
        return true;
    }
#endif

  public:
    /** @brief Default constructor */
    Tracker(std::shared_ptr<eds::tracking::KeyFrame> kf, const eds::tracking::Config& config) : Tracker(config) { this->kf = kf; }   // Tracker.hpp:62
    /** @brief Default constructor */
    Tracker(const eds::tracking::Config& config) : config(config) {                                                               // Tracker.hpp:65
        for (int i = 0; i < 3; ++i) px[i] = 0.0;
        qx = Eigen::Quaterniond::Identity(); seed_velocity();
    }
    ~Tracker() { if (h) eds_trk_destroy(h); }
    Tracker(const Tracker&) = delete;
    Tracker& operator=(const Tracker&) = delete;

    void reset(std::shared_ptr<eds::tracking::KeyFrame> kf, const Eigen::Vector3d& px, const Eigen::Quaterniond& qx, const bool& keep_velo = true) {
        this->kf = kf; this->px = px; this->qx = qx; if (!keep_velo) seed_velocity();                                             // Tracker.cpp:49-64
    }
    void reset(std::shared_ptr<eds::tracking::KeyFrame> kf, const Eigen::Vector3d& px, const Eigen::Quaterniond& qx, const base::Vector6d& velo) {
        this->kf = kf; this->px = px; this->qx = qx; for (int i = 0; i < 6; ++i) vx[i] = velo[i];                                 // Tracker.cpp:66-72
    }
    /** Stores the INVERSE of T_kf_ef (Tracker.cpp:74-79). */
    void set(const base::Transform3d& T_kf_ef) {
        const base::Transform3d Ti = T_kf_ef.inverse();
        for (int i = 0; i < 3; ++i) px[i] = Ti(i, 3);
        // Eigen::Quaterniond(Matrix3d)
        const double tr = Ti(0, 0) + Ti(1, 1) + Ti(2, 2);
        double q[4];
        if (tr > 0) { double s = std::sqrt(tr + 1.0) * 2; q[3] = 0.25 * s; q[0] = (Ti(2, 1) - Ti(1, 2)) / s; q[1] = (Ti(0, 2) - Ti(2, 0)) / s; q[2] = (Ti(1, 0) - Ti(0, 1)) / s; }
        else {
            int i = 0; if (Ti(1, 1) > Ti(0, 0)) i = 1; if (Ti(2, 2) > Ti(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (i + 2) % 3;
            double s = std::sqrt(Ti(i, i) - Ti(j, j) - Ti(k, k) + 1.0) * 2;
            q[i] = 0.25 * s; q[j] = (Ti(j, i) + Ti(i, j)) / s; q[k] = (Ti(k, i) + Ti(i, k)) / s; q[3] = (Ti(k, j) - Ti(j, k)) / s;
        }
        for (int i = 0; i < 4; ++i) qx.coeffs()[i] = q[i];
    }
    void optimize(const int& id, const std::vector<double>* event_frame, ::base::Transform3d& T_kf_ef, const Eigen::Vector3d& px,
                  const Eigen::Quaterniond& qx, const eds::tracking::LOSS_PARAM_METHOD loss_param_method) {                       // Tracker.cpp:81-91
        this->px = px; this->qx = qx; optimize(id, event_frame, T_kf_ef, loss_param_method);
    }
    void optimize(const int& id, const std::vector<double>* event_frame, ::base::Transform3d& T_kf_ef, const Eigen::Matrix<double, 6, 1>& vx,
                  const eds::tracking::LOSS_PARAM_METHOD loss_param_method) {                                                    // Tracker.cpp:93-102
        this->vx = vx; optimize(id, event_frame, T_kf_ef, loss_param_method);
    }
    /** Tracker::optimize (Tracker.cpp:104-241).  false: nothing was updated. */
    bool optimize(const int& id, const std::vector<double>* event_frame, ::base::Transform3d& T_kf_ef,
                  const eds::tracking::LOSS_PARAM_METHOD loss_param_method = eds::tracking::LOSS_PARAM_METHOD::MAD) {
        if (!kf || !event_frame) return false;
        const int N = (int)kf->norm_coord.size();
#ifdef EDS_HIP_WITH_EDS_TYPES
        const int rows = kf->img.rows, cols = kf->img.cols;
        const double fx = kf->K_ref.template at<double>(0, 0), fy = kf->K_ref.template at<double>(1, 1), cx = kf->K_ref.template at<double>(0, 2),
                     cy = kf->K_ref.template at<double>(1, 2);
        std::vector<double> idp; kf->inv_depth.getIDepth(idp);                                                                   // Tracker.cpp:167
#else
        const int rows = kf->rows, cols = kf->cols;
        const double fx = kf->K_ref[0], fy = kf->K_ref[4], cx = kf->K_ref[2], cy = kf->K_ref[5];
        const std::vector<double>& idp = kf->inv_depth;
#endif
        if (N < 1 || (int)idp.size() != N || (int)kf->grad.size() != N || (int)kf->weights.size() != N ||
            event_frame->size() != (size_t)rows * cols) return false;                                                           // asserts at PhotometricError.hpp:70-73
        // (a failure of the library underneath — no device, no memory, a HIP error — is reported the way the reference reports an
        // unusable solution: false, nothing updated; hipLastStatus() / hipLastError() say what it was)
        last_status = EDS_OK; last_error.clear();
        if (!ensure_handle(N, rows, cols) || !upload_keyframe(N, idp, fx, fy, cx, cy)) return false;
        int rc;
        if ((rc = eds_trk_set_event_frame(h, 0, event_frame->data())) != EDS_OK) return fail_(rc, "eds_trk_set_event_frame");
        double p[3] = {px[0], px[1], px[2]}, q[4] = {qx.x(), qx.y(), qx.z(), qx.w()}, v[6];
        for (int i = 0; i < 6; ++i) v[i] = vx[i];
        eds_trk_info ti;
        std::memset(&ti, 0, sizeof(ti));
        rc = eds_trk_optimize(h, 0, id, p, q, v, &ti);
        info.meas_time_us = ti.meas_time_us; info.num_points = ti.num_points; info.num_iterations = ti.num_iterations;           // Tracker.cpp:209-213
        info.time_seconds = ti.time_seconds; info.success = ti.success;
        if (rc == EDS_ERR_NOT_USABLE) { last_status = rc; last_error = "solution not usable"; return false; }                   // Tracker.cpp:236-239
        if (rc != EDS_OK) return fail_(rc, "eds_trk_optimize");
        for (int i = 0; i < 3; ++i) px[i] = p[i];
        for (int i = 0; i < 4; ++i) qx.coeffs()[i] = q[i];
        for (int i = 0; i < 6; ++i) vx[i] = v[i];
        T_kf_ef = getTransform().inverse();                                                                                     // Tracker.cpp:220
        kf->residuals.resize(N);
        // Tracker.cpp:223-233 — kf->residuals filled, then config.loss_params = getLossParams(method), whose MAD selection reorders
        // kf->residuals in place — as ONE call and one read-back (round 2: get_residuals -> loss_param -> get_residuals)
        double tau = config.loss_params.empty() ? 0.0 : config.loss_params[0];
        if (eds_trk_residuals_and_loss(h, 0, (int)loss_param_method, kf->residuals.data(), &tau) == EDS_OK && loss_param_method != CONSTANT)
            config.loss_params = std::vector<double>{tau};
        return true;
    }
    /** T_ef_kf = SE3(qx, px) (Tracker.cpp:243-249). */
    ::base::Transform3d getTransform() {
        double t[3], q[4];
        current_pose(t, q);
        return to_transform(t, q);
    }
    /** Tracker.cpp:251-260: records the pose, returns the mean-filtered one once 3 poses are in the history, identity before. */
#ifdef EDS_HIP_REFERENCE_MEMBERS
    ::base::Transform3d getTransform(bool& result) {        // behaviour of Tracker.cpp:251-260, written on this shim's helpers
        // the history (the reference's std::vector<eds::SE3>, which the EDS tree's own getFilteredPose reads) gains the current
        // pose; a copy of it goes through that filter, which replaces it by the mean once enough poses exist
        poses.emplace_back(qx, px);
        ::eds::SE3 filtered = poses.back();
        result = getFilteredPose(filtered);
        if (!result) return base::Transform3d::Identity();
        const Eigen::Quaterniond& fq = filtered.unit_quaternion();
        const double t[3] = {filtered.translation()[0], filtered.translation()[1], filtered.translation()[2]};
        const double q[4] = {fq.x(), fq.y(), fq.z(), fq.w()};
        return to_transform(t, q);
    }
    // Declared here with the reference's signatures (Tracker.hpp:98-111), DEFINED by the EDS tree's own Tracker.cpp:378-648
    // (header comment; tests/cpp/shim_reference_members.cpp is the compile-checked example of such a translation unit)
    void trackPoints(const cv::Mat& event_frame, const uint16_t& patch_radius = 7);                                              // :98
    void trackPointsPyr(const cv::Mat& event_frame, const size_t num_level = 3);                                                 // :100
    std::vector<cv::Point2d> trackPointsAlongEpiline(const cv::Mat& event_frame, const uint16_t& patch_radius = 7,
                                                     const int& border_type = cv::BORDER_DEFAULT, const uint8_t& border_value = 255);   // :102-103
    cv::Mat getEMatrix();                                                                                                        // :105
    cv::Mat getFMatrix();                                                                                                        // :107
    bool getFilteredPose(eds::SE3& pose, const size_t& mean_filter_size = 3);                                                    // :111
#else
    ::base::Transform3d getTransform(bool& result) {
        std::array<double, 7> cur;
        current_pose(cur.data(), cur.data() + 3);
        poses.push_back(cur);
        double t[3], q[4];
        result = filtered_pose(t, q);
        return result ? to_transform(t, q) : base::Transform3d::Identity();
    }
#endif
    Eigen::Matrix<double, 6, 1>& getVelocity() { return vx; }                                                                    // Tracker.cpp:262-265
    const Eigen::Vector3d linearVelocity() { Eigen::Vector3d o; for (int i = 0; i < 3; ++i) o[i] = vx[i]; return o; }
    const Eigen::Vector3d angularVelocity() { Eigen::Vector3d o; for (int i = 0; i < 3; ++i) o[i] = vx[3 + i]; return o; }
    /** Tracker::getLossParams (Tracker.cpp:281-317); MAD reorders kf->residuals in place like the reference. */
    std::vector<double> getLossParams(eds::tracking::LOSS_PARAM_METHOD method = CONSTANT) {
        if (method == CONSTANT || !h) return config.loss_params;
        double tau = config.loss_params.empty() ? 0.0 : config.loss_params[0];
        if (eds_trk_loss_param(h, 0, (int)method, &tau) != EDS_OK) return config.loss_params;
        if (kf) { kf->residuals.resize(kf->norm_coord.size()); eds_trk_get_residuals(h, 0, kf->residuals.data()); }
        return std::vector<double>{tau};
    }
    ::eds::tracking::TrackerInfo getInfo() { return info; }

    /** Tracker::getCoord (Tracker.cpp:319-376): the points re-projected under the current pose; with delete_out_point the
     *  ones that left the frame are erased from the keyframe (all index-aligned vectors) and from the device copy.  Like the
     *  reference, which reads kf's vectors at call time, the device copy is refreshed first when the KeyFrame changed. */
    std::vector<cv::Point2d> getCoord(const bool& delete_out_point = false) {
        std::vector<cv::Point2d> coord;
        if (!kf) return coord;
        const int N = (int)kf->norm_coord.size();
        if (N < 1) return coord;
#ifdef EDS_HIP_WITH_EDS_TYPES
        const int rows = kf->img.rows, cols = kf->img.cols;
        const double fx = kf->K_ref.template at<double>(0, 0), fy = kf->K_ref.template at<double>(1, 1), cx = kf->K_ref.template at<double>(0, 2),
                     cy = kf->K_ref.template at<double>(1, 2);
        std::vector<double> idp; kf->inv_depth.getIDepth(idp);
#else
        const int rows = kf->rows, cols = kf->cols;
        const double fx = kf->K_ref[0], fy = kf->K_ref[4], cx = kf->K_ref[2], cy = kf->K_ref[5];
        const std::vector<double>& idp = kf->inv_depth;
#endif
        if ((int)idp.size() != N || (int)kf->grad.size() != N || (int)kf->weights.size() != N) return coord;
        last_status = EDS_OK; last_error.clear();
        if (!ensure_handle(N, rows, cols) || !upload_keyframe(N, idp, fx, fy, cx, cy)) return coord;     // (empty: see the header's error convention)
        double p[3] = {px[0], px[1], px[2]}, q[4] = {qx.x(), qx.y(), qx.z(), qx.w()}, v[6];
        for (int i = 0; i < 6; ++i) v[i] = vx[i];
        int rc;
        if ((rc = eds_trk_set_state(h, 0, p, q, v)) != EDS_OK) { fail_(rc, "eds_trk_set_state"); return coord; }
        coord.resize(N);
        std::vector<double> tracks(2 * (size_t)N);
        std::vector<int32_t> kept(N);
        int n = 0;
        if ((rc = eds_trk_update_points(h, 0, delete_out_point ? 1 : 0, &coord[0].x, tracks.data(), kept.data(), &n, &squared_norm_flow)) != EDS_OK) {
            fail_(rc, "eds_trk_update_points"); coord.clear(); return coord;
        }
        coord.resize(n);
        if (n != N) {                        // what repeated KeyFrame::erasePoint does (KeyFrame.cpp:1060-1106), in one sweep
#ifdef EDS_HIP_WITH_EDS_TYPES
            std::vector<char> keep(N, 0);
            for (int k = 0; k < n; ++k) keep[kept[k]] = 1;
            for (int i = N - 1; i >= 0; --i) if (!keep[i]) kf->erasePoint(i);
#else
            auto compact = [&](auto& vec) {
                if ((int)vec.size() != N) return;
                for (int k = 0; k < n; ++k) vec[k] = vec[kept[k]];
                vec.resize(n);
            };
            compact(kf->norm_coord); compact(kf->grad); compact(kf->coord); compact(kf->weights); compact(kf->residuals); compact(kf->inv_depth);
            kf->tracks.resize(N); compact(kf->tracks);
#endif
            // the device planes were compacted the same way: keep the shadow copy in step so the next optimize does not re-upload
            auto compact_shadow = [&](double* base, int width) {
                for (int k = 0; k < n; ++k) for (int c = 0; c < width; ++c) base[(size_t)width * k + c] = base[(size_t)width * kept[k] + c];
            };
            if (device_kf_valid && up_points.size() == 5 * (size_t)N && up_idp.size() == (size_t)N) {
                std::vector<double> np_(5 * (size_t)n);
                compact_shadow(up_points.data(), 2); compact_shadow(up_points.data() + 2 * (size_t)N, 2); compact_shadow(up_points.data() + 4 * (size_t)N, 1);
                std::memcpy(np_.data(), up_points.data(), 2 * (size_t)n * 8);
                std::memcpy(np_.data() + 2 * (size_t)n, up_points.data() + 2 * (size_t)N, 2 * (size_t)n * 8);
                std::memcpy(np_.data() + 4 * (size_t)n, up_points.data() + 4 * (size_t)N, (size_t)n * 8);
                up_points.swap(np_);
                compact_shadow(up_idp.data(), 1); up_idp.resize(n);
            } else {
                device_kf_valid = false;
            }
        }
        kf->tracks.resize(n);
        for (int k = 0; k < n; ++k) { kf->tracks[k][0] = tracks[2 * k]; kf->tracks[k][1] = tracks[2 * k + 1]; }                 // Tracker.cpp:364-366
        return coord;
    }
    /** Tracker::needNewKeyframe (Tracker.cpp:650-654). */
    bool needNewKeyframe(const double& weight_factor = 0.03) {
#ifdef EDS_HIP_WITH_EDS_TYPES
        const int rows = kf->img.rows, cols = kf->img.cols;
#else
        const int rows = kf->rows, cols = kf->cols;
#endif
        const double image_weight = (cols + rows) * weight_factor;
        return (image_weight * std::sqrt((float)squared_norm_flow) / (cols + rows)) > 1;
    }
    /** Extensions: why the last optimize / getCoord returned false / nothing (EDS_OK and "" after a success). */
    int hipLastStatus() const { return last_status; }
    const std::string& hipLastError() const { return last_error; }
    /** Extension (the reference keeps it private, Tracker.hpp:58): the mean squared flow of the last getCoord. */
    double hipSquaredNormFlow() const { return squared_norm_flow; }
};

}}  // namespace eds::tracking
