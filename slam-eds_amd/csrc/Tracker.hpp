// C++ shim with the public surface of eds::tracking::Tracker (reference src/tracking/Tracker.hpp:36-114)
// for the event-to-model alignment path, implemented on the C ABI of libeds_hip.so (include/eds_hip.h).
//
// Inside EDS (Eigen, OpenCV and Rock base-types available) define EDS_HIP_WITH_EDS_TYPES before including
// this header: the shim then uses Eigen::Vector3d / Eigen::Quaterniond / base::Transform3d /
// eds::tracking::KeyFrame / eds::tracking::Config directly, and an external component relinks unchanged
// (INTEGRATION.md).  Without it (this repository's own tests: none of those libraries exist in the build
// image) the minimal stand-ins below provide the same member names and memory layouts.
//
// Mirrored members:   Tracker(kf, config), Tracker(config), reset (both overloads), set, optimize (all three
// overloads), getTransform, getVelocity, linearVelocity, angularVelocity, getLossParams, getInfo, getCoord,
// needNewKeyframe, public config.
// Not mirrored (outside the hot path, reference Tracker.cpp:378-648): trackPoints*, getEMatrix, getFMatrix,
// getFilteredPose — keep the reference implementation for those.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eds_hip.h"

#ifdef EDS_HIP_WITH_EDS_TYPES
#include <eds/tracking/Config.hpp>
#include <eds/tracking/KeyFrame.hpp>
#include <eds/tracking/Types.hpp>
#else
// ---- minimal stand-ins (same names, same data layout as the reference types they replace) ----------------
namespace base {
struct Time { int64_t microseconds = 0; };
typedef std::array<double, 6> Vector6d;
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
namespace Eigen {
struct Vector3d { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; }
                  static Vector3d Zero() { return Vector3d{{0, 0, 0}}; } double* data() { return v; } const double* data() const { return v; } };
struct Quaterniond {                        // coeffs() order x,y,z,w like Eigen
    double c[4];
    static Quaterniond Identity() { return Quaterniond{{0, 0, 0, 1}}; }
    double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
    double* coeffs() { return c; } const double* coeffs() const { return c; }
};
}  // namespace Eigen
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
    std::vector<std::array<double, 2>> tracks;   // Eigen::Vector2d per point (KeyFrame.hpp:92); filled by Tracker::getCoord
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
};

class Tracker {
  public:
    ::eds::tracking::Config config;         // public like the reference (Tracker.hpp:40)
    HipOptions hip;

  private:
    std::shared_ptr<eds::tracking::KeyFrame> kf;
    Eigen::Vector3d px;
    Eigen::Quaterniond qx;
    double vx_[6];
    eds::tracking::TrackerInfo info;
    eds_trk* h = nullptr;
    int h_cap = 0, h_rows = 0, h_cols = 0;

    static void seed_velocity(double* v) { const double c = 1.0 / std::sqrt(6.0); for (int i = 0; i < 6; ++i) v[i] = c; }   // Tracker.cpp:45-46
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
    void ensure_handle(int N, int rows, int cols) {
        eds_trk_cfg c = make_cfg();
        if (h && (h_cap < N || h_rows != rows || h_cols != cols)) { eds_trk_destroy(h); h = nullptr; }
        if (!h) {
            h_cap = std::max(N, 2048); h_rows = rows; h_cols = cols;
            if (eds_trk_create(&c, 1, h_cap, rows, cols, &h) != EDS_OK) throw std::runtime_error(std::string("eds_trk_create: ") + eds_last_error());
        } else if (eds_trk_set_config(h, &c) != EDS_OK) {
            throw std::runtime_error(std::string("eds_trk_set_config: ") + eds_last_error());
        }
    }

  public:
    Tracker(std::shared_ptr<eds::tracking::KeyFrame> kf_, const eds::tracking::Config& config_) : Tracker(config_) { kf = kf_; }   // Tracker.hpp:62
    explicit Tracker(const eds::tracking::Config& config_) : config(config_) {                                                   // Tracker.hpp:65
        px = Eigen::Vector3d::Zero(); qx = Eigen::Quaterniond::Identity(); seed_velocity(vx_);
    }
    ~Tracker() { if (h) eds_trk_destroy(h); }
    Tracker(const Tracker&) = delete;
    Tracker& operator=(const Tracker&) = delete;

    void reset(std::shared_ptr<eds::tracking::KeyFrame> kf_, const Eigen::Vector3d& px_, const Eigen::Quaterniond& qx_, const bool& keep_velo = true) {
        kf = kf_; px = px_; qx = qx_; if (!keep_velo) seed_velocity(vx_);                                                        // Tracker.cpp:49-64
    }
    void reset(std::shared_ptr<eds::tracking::KeyFrame> kf_, const Eigen::Vector3d& px_, const Eigen::Quaterniond& qx_, const base::Vector6d& velo) {
        kf = kf_; px = px_; qx = qx_; for (int i = 0; i < 6; ++i) vx_[i] = velo[i];                                              // Tracker.cpp:66-72
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
    void optimize(const int& id, const std::vector<double>* event_frame, base::Transform3d& T_kf_ef, const Eigen::Vector3d& px_,
                  const Eigen::Quaterniond& qx_, const LOSS_PARAM_METHOD loss_param_method) {                                    // Tracker.cpp:81-91
        px = px_; qx = qx_; optimize(id, event_frame, T_kf_ef, loss_param_method);
    }
    void optimize(const int& id, const std::vector<double>* event_frame, base::Transform3d& T_kf_ef, const base::Vector6d& vx_in,
                  const LOSS_PARAM_METHOD loss_param_method) {                                                                   // Tracker.cpp:93-102
        for (int i = 0; i < 6; ++i) vx_[i] = vx_in[i]; optimize(id, event_frame, T_kf_ef, loss_param_method);
    }
    /** Tracker::optimize (Tracker.cpp:104-241).  false: nothing was updated. */
    bool optimize(const int& id, const std::vector<double>* event_frame, base::Transform3d& T_kf_ef,
                  const LOSS_PARAM_METHOD loss_param_method = MAD) {
        if (!kf || !event_frame) return false;
        const int N = (int)kf->norm_coord.size();
#ifdef EDS_HIP_WITH_EDS_TYPES
        const int rows = kf->img.rows, cols = kf->img.cols;
        const double fx = kf->K_ref.at<double>(0, 0), fy = kf->K_ref.at<double>(1, 1), cx = kf->K_ref.at<double>(0, 2), cy = kf->K_ref.at<double>(1, 2);
        std::vector<double> idp; kf->inv_depth.getIDepth(idp);                                                                   // Tracker.cpp:167
#else
        const int rows = kf->rows, cols = kf->cols;
        const double fx = kf->K_ref[0], fy = kf->K_ref[4], cx = kf->K_ref[2], cy = kf->K_ref[5];
        const std::vector<double>& idp = kf->inv_depth;
#endif
        if (N < 1 || (int)idp.size() != N || (int)kf->grad.size() != N || (int)kf->weights.size() != N ||
            event_frame->size() != (size_t)rows * cols) return false;                                                           // asserts at PhotometricError.hpp:70-73
        ensure_handle(N, rows, cols);
        // the reference hands raw pointers to the functor on every call (Tracker.cpp:189-191): upload on every call
        if (eds_trk_set_keyframe(h, 0, N, &kf->norm_coord[0].x, &kf->grad[0].x, idp.data(), kf->weights.data(), fx, fy, cx, cy) != EDS_OK ||
            eds_trk_set_event_frame(h, 0, event_frame->data()) != EDS_OK)
            throw std::runtime_error(std::string("libeds_hip: ") + eds_last_error());
        double p[3] = {px[0], px[1], px[2]}, q[4] = {qx.x(), qx.y(), qx.z(), qx.w()}, v[6];
        for (int i = 0; i < 6; ++i) v[i] = vx_[i];
        eds_trk_info ti;
        const int rc = eds_trk_optimize(h, 0, id, p, q, v, &ti);
        info.meas_time_us = ti.meas_time_us; info.num_points = ti.num_points; info.num_iterations = ti.num_iterations;           // Tracker.cpp:209-213
        info.time_seconds = ti.time_seconds; info.success = ti.success;
        if (rc == EDS_ERR_NOT_USABLE) return false;                                                                             // Tracker.cpp:236-239
        if (rc != EDS_OK) throw std::runtime_error(std::string("eds_trk_optimize: ") + eds_last_error());
        for (int i = 0; i < 3; ++i) px[i] = p[i];
        for (int i = 0; i < 4; ++i) qx.coeffs()[i] = q[i];
        for (int i = 0; i < 6; ++i) vx_[i] = v[i];
        T_kf_ef = getTransform().inverse();                                                                                     // Tracker.cpp:220
        kf->residuals.resize(N);
        eds_trk_get_residuals(h, 0, kf->residuals.data());                                                                      // Tracker.cpp:223-230
        config.loss_params = getLossParams(loss_param_method);                                                                  // Tracker.cpp:233
        return true;
    }
    /** T_ef_kf = SE3(qx, px) (Tracker.cpp:243-249). */
    base::Transform3d getTransform() const {
        base::Transform3d T = base::Transform3d::Identity();
        const double n = std::sqrt(qx.x() * qx.x() + qx.y() * qx.y() + qx.z() * qx.z() + qx.w() * qx.w());
        const double x = qx.x() / n, y = qx.y() / n, z = qx.z() / n, w = qx.w() / n;
        T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - z * w);     T(0, 2) = 2 * (x * z + y * w);
        T(1, 0) = 2 * (x * y + z * w);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - x * w);
        T(2, 0) = 2 * (x * z - y * w);     T(2, 1) = 2 * (y * z + x * w);     T(2, 2) = 1 - 2 * (x * x + y * y);
        for (int i = 0; i < 3; ++i) T(i, 3) = px[i];
        return T;
    }
    base::Vector6d getVelocity() const { base::Vector6d o; for (int i = 0; i < 6; ++i) o[i] = vx_[i]; return o; }
    const Eigen::Vector3d linearVelocity() const { Eigen::Vector3d o; for (int i = 0; i < 3; ++i) o[i] = vx_[i]; return o; }
    const Eigen::Vector3d angularVelocity() const { Eigen::Vector3d o; for (int i = 0; i < 3; ++i) o[i] = vx_[3 + i]; return o; }
    /** Tracker::getLossParams (Tracker.cpp:281-317); MAD reorders kf->residuals in place like the reference. */
    std::vector<double> getLossParams(LOSS_PARAM_METHOD method = CONSTANT) {
        if (method == CONSTANT || !h) return config.loss_params;
        double tau = config.loss_params.empty() ? 0.0 : config.loss_params[0];
        if (eds_trk_loss_param(h, 0, (int)method, &tau) != EDS_OK) return config.loss_params;
        if (kf) { kf->residuals.resize(kf->norm_coord.size()); eds_trk_get_residuals(h, 0, kf->residuals.data()); }
        return std::vector<double>{tau};
    }
    eds::tracking::TrackerInfo getInfo() const { return info; }

    double squared_norm_flow = 0.0;                                                                                             // Tracker.hpp:58
    /** Tracker::getCoord (Tracker.cpp:319-376): the points re-projected under the current pose; with delete_out_point the
     *  ones that left the frame are erased from the keyframe (all index-aligned vectors) and from the device copy. */
    std::vector<cv::Point2d> getCoord(const bool& delete_out_point = false) {
        std::vector<cv::Point2d> coord;
        if (!kf || !h) return coord;
        const int N = (int)kf->norm_coord.size();
        double p[3] = {px[0], px[1], px[2]}, q[4] = {qx.x(), qx.y(), qx.z(), qx.w()}, v[6];
        for (int i = 0; i < 6; ++i) v[i] = vx_[i];
        if (eds_trk_set_state(h, 0, p, q, v) != EDS_OK) throw std::runtime_error(std::string("eds_trk_set_state: ") + eds_last_error());
        coord.resize(N);
        std::vector<double> tracks(2 * (size_t)N);
        std::vector<int32_t> kept(N);
        int n = 0;
        if (eds_trk_update_points(h, 0, delete_out_point ? 1 : 0, &coord[0].x, tracks.data(), kept.data(), &n, &squared_norm_flow) != EDS_OK)
            throw std::runtime_error(std::string("eds_trk_update_points: ") + eds_last_error());
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
        }
        kf->tracks.resize(n);
        for (int k = 0; k < n; ++k) { kf->tracks[k][0] = tracks[2 * k]; kf->tracks[k][1] = tracks[2 * k + 1]; }                 // Tracker.cpp:364-366
        return coord;
    }
    /** Tracker::needNewKeyframe (Tracker.cpp:650-654). */
    bool needNewKeyframe(const double& weight_factor = 0.03) const {
#ifdef EDS_HIP_WITH_EDS_TYPES
        const int rows = kf->img.rows, cols = kf->img.cols;
#else
        const int rows = kf->rows, cols = kf->cols;
#endif
        const double image_weight = (cols + rows) * weight_factor;
        return (image_weight * std::sqrt((float)squared_norm_flow) / (cols + rows)) > 1;
    }
};

}}  // namespace eds::tracking
