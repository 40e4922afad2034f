// TEST-ONLY mock of eds/tracking/Config.hpp (reference tracking/Config.hpp:36-68): see Types.hpp in this directory.
#pragma once
#include <string>
#include <vector>
#include "Types.hpp"
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
}}  // namespace eds::tracking
