// oracle/_ref driver: the REAL reference functor — /root/reference/src/tracking/PhotometricError.hpp, included unmodified where it lies —
// solved the way eds::tracking::Tracker::optimize solves it (Tracker.cpp:104-241 replayed: options :117-143, loss :146-161, block
// partition :178-195, parameterisations :111-114,197-198, Solve :202, residuals at the solution :223-230).  TEST INFRASTRUCTURE: it
// exists to pin oracle/eds_oracle.hpp (tests/test_ref_pin.py); nothing under slam-eds_amd/ may use it.  It needs Ceres <= 2.1
// (LocalParameterization), Eigen, OpenCV, yaml-cpp and Rock base-types, none of which this image has: oracle/ref/Makefile probes for
// them and builds this file only when they are all there (no stand-in headers).  Until then: PARITY UNPINNED.
//   ref_driver case.bin out.bin
// case.bin: int32 {N, H, W, num_threads, loss (0 none | 1 Huber | 2 Cauchy), max_num_iterations}, double {fx, fy, cx, cy, loss_param,
//           function_tolerance}, double norm_coord[N][2], grad[N][2], idp[N], weights[N], frame[H*W], px[3], qx[4] (x y z w), vx[6]
// out.bin:  double px[3], qx[4], vx[6], {usable, successful, unsuccessful, termination_type, initial_cost, final_cost}, residuals[N]
#include <eds/tracking/PhotometricError.hpp>

#include <cstdint>
#include <cstdio>
#include <vector>

template <class T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hd[6];
    double k[6];
    if (!rd(f, hd, 6) || !rd(f, k, 6)) return 3;
    const int N = hd[0], H = hd[1], W = hd[2], T = hd[3], loss = hd[4], iters = hd[5];
    std::vector<cv::Point2d> norm_coord(N), grad(N);
    std::vector<double> idp(N), weights(N), frame((size_t)H * W), residuals(N);
    Eigen::Vector3d px;
    Eigen::Quaterniond qx;
    Eigen::Matrix<double, 6, 1> vx;
    static_assert(sizeof(cv::Point2d) == 16, "cv::Point2d is two doubles");
    if (!rd(f, norm_coord.data(), N) || !rd(f, grad.data(), N) || !rd(f, idp.data(), N) || !rd(f, weights.data(), N) ||
        !rd(f, frame.data(), frame.size()) || !rd(f, px.data(), 3) || !rd(f, qx.coeffs().data(), 4) || !rd(f, vx.data(), 6)) return 3;
    fclose(f);

    ceres::Problem problem;                                                            // Tracker.cpp:108-114
    ceres::Solver::Options options;
    ceres::LocalParameterization* quaternion_local_parameterization = new ceres::EigenQuaternionParameterization;
    ceres::LocalParameterization* velocity_local_parameterization =
        new ceres::AutoDiffLocalParameterization<eds::tracking::UnitNormVectorAddition, 6, 6>;
    options.linear_solver_type = ceres::DENSE_QR;                                      // :117-134 (the YAML's choice; dense 12 columns)
    options.num_threads = T;                                                           // :138-143
    options.max_num_iterations = iters;
    options.function_tolerance = k[5];
    options.minimizer_progress_to_stdout = false;
    options.gradient_tolerance = 1e-08;
    options.parameter_tolerance = 1e-06;
    ceres::LossFunction* loss_function = loss == 1 ? (ceres::LossFunction*)new ceres::HuberLoss(k[4])      // :146-161
                                        : loss == 2 ? (ceres::LossFunction*)new ceres::CauchyLoss(k[4]) : NULL;
    std::vector<ceres::CostFunction*> blocks;
    const int num_elements = N / options.num_threads;                                   // :178-195
    for (int i = 0; i < options.num_threads; ++i) {
        const int extra = (i + 1 == options.num_threads) ? N - (i + 1) * num_elements : 0;
        ceres::CostFunction* c = eds::tracking::PhotometricError::Create(&grad, &norm_coord, &idp, &weights, &frame, H, W, k[0], k[1], k[2], k[3],
                                                                         i * num_elements, num_elements + extra);
        problem.AddResidualBlock(c, loss_function, px.data(), qx.coeffs().data(), vx.data());
        blocks.push_back(c);
    }
    problem.SetParameterization(qx.coeffs().data(), quaternion_local_parameterization);   // :197-198
    problem.SetParameterization(vx.data(), velocity_local_parameterization);
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);                                          // :202
    if (summary.IsSolutionUsable()) {                                                   // :217-230
        double* params[3] = {px.data(), qx.coeffs().data(), vx.data()};
        for (int i = 0; i < options.num_threads; ++i) blocks[i]->Evaluate(params, &residuals[i * num_elements], nullptr);
    }
    const double s[6] = {summary.IsSolutionUsable() ? 1.0 : 0.0, (double)summary.num_successful_steps, (double)summary.num_unsuccessful_steps,
                         (double)summary.termination_type, summary.initial_cost, summary.final_cost};
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 4;
    fwrite(px.data(), 8, 3, o); fwrite(qx.coeffs().data(), 8, 4, o); fwrite(vx.data(), 8, 6, o); fwrite(s, 8, 6, o);
    fwrite(residuals.data(), 8, N, o);
    fclose(o);
    return 0;
}
