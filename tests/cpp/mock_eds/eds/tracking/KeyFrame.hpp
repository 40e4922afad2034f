// TEST-ONLY mock of eds/tracking/KeyFrame.hpp: the members of eds::tracking::KeyFrame that Tracker touches, with the
// reference's names and signatures (KeyFrame.hpp:60-96,154; mapping/DepthPoints.hpp:87).  See Types.hpp in this directory.
#pragma once
#include <vector>
#include "Types.hpp"
namespace eds { namespace mapping {
struct DepthPoints {
    std::vector<double> mu;
    void getIDepth(std::vector<double>& x) { x = mu; }                      // DepthPoints.hpp:87
};
}}
namespace eds { namespace tracking {
struct KFPointIterators { int dummy; };
struct KeyFrame {
    cv::Mat img, K_ref;                                                      // KeyFrame.hpp:68,76
    std::vector<cv::Point2d> coord, norm_coord, grad;                        // :80
    std::vector<double> weights, residuals;                                  // :88,90
    std::vector<Eigen::Vector2d> tracks;                                     // :92
    eds::mapping::DepthPoints inv_depth;                                     // :96
    KFPointIterators erasePoint(const int& idx) {                           // :154 (KeyFrame.cpp:1060-1106)
        auto er = [&](auto& v) { if ((int)v.size() > idx) v.erase(v.begin() + idx); };
        er(coord); er(norm_coord); er(grad); er(weights); er(residuals); er(tracks); er(inv_depth.mu);
        return KFPointIterators{0};
    }
};
}}
