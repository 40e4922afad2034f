// Probe of oracle/ref/Makefile: compiles only where everything the reference's PhotometricError.hpp pulls in is installed — Ceres with
// the LocalParameterization API (<= 2.1: removed in 2.2, used at Tracker.cpp:111-114), Eigen, OpenCV, yaml-cpp, Rock base-types.
#include <ceres/ceres.h>
#include <ceres/version.h>
#if CERES_VERSION_MAJOR > 2 || (CERES_VERSION_MAJOR == 2 && CERES_VERSION_MINOR > 1)
#error "Ceres > 2.1: ceres::LocalParameterization / EigenQuaternionParameterization / AutoDiffLocalParameterization are gone"
#endif
#include <ceres/cubic_interpolation.h>
#include <Eigen/Dense>
#include <opencv2/opencv.hpp>
#include <yaml-cpp/yaml.h>
#include <base/Float.hpp>
#include <base/Time.hpp>
#include <base/samples/DistanceImage.hpp>
#include <base/samples/Pointcloud.hpp>
int main() { ceres::LocalParameterization* p = new ceres::EigenQuaternionParameterization; delete p; return 0; }
