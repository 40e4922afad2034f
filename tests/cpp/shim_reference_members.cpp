// TEST-ONLY example of the translation unit an EDS tree adds next to slam-eds_amd/csrc/Tracker.hpp: the DEFINITIONS of the six members
// that stay the reference's (its Tracker.cpp:378-648 — trackPoints, trackPointsPyr, trackPointsAlongEpiline, getEMatrix, getFMatrix,
// getFilteredPose), compiled against the shim's class with EDS_HIP_REFERENCE_MEMBERS.  The bodies HERE are stand-ins written for this
// check (the reference's need OpenCV and Sophus, absent from the image, and are not copied): each touches exactly the private and public
// members the reference's body touches — this->kf, px, qx, vx, poses, squared_norm_flow, getCoord, linearVelocity / angularVelocity,
// getTransform, getFMatrix — so that a shim that renames, retypes or hides one of them stops this file from compiling.  Linked with
// shim_eds_types_check.cpp into one object set by tests/test_cpp_shim_gpu.py (no duplicate or missing symbol among the members).
#define EDS_HIP_WITH_EDS_TYPES
#define EDS_HIP_REFERENCE_MEMBERS
#include "../../slam-eds_amd/csrc/Tracker.hpp"

namespace eds { namespace tracking {

void Tracker::trackPoints(const cv::Mat& event_frame, const uint16_t& patch_radius) {      // Tracker.cpp:378-434: getCoord(true), kf->tracks, erasePoint
    std::vector<cv::Point2d> coord = this->getCoord(true);
    for (size_t i = 0; i < coord.size() && i < this->kf->tracks.size(); ++i) this->kf->tracks[i][0] += 0.0 * (event_frame.cols + patch_radius);
    if (!coord.empty() && this->vx[0] != this->vx[0]) this->kf->erasePoint(0);
}
void Tracker::trackPointsPyr(const cv::Mat& event_frame, const size_t num_level) {         // :436-488
    std::vector<cv::Point2d> coord = this->getCoord(true);
    for (size_t i = 0; i < coord.size() && i < this->kf->tracks.size(); ++i) this->kf->tracks[i][1] += 0.0 * (event_frame.rows + (int)num_level);
}
std::vector<cv::Point2d> Tracker::trackPointsAlongEpiline(const cv::Mat& event_frame, const uint16_t& patch_radius, const int& border_type,
                                                          const uint8_t& border_value) {   // :490-553: kf->coord / norm_coord / inv_depth, velocities, getFMatrix
    const Eigen::Vector3d lv = this->linearVelocity(), av = this->angularVelocity();
    cv::Mat F = this->getFMatrix();
    std::vector<cv::Point2d> out(this->kf->coord);
    if (out.size() != this->kf->norm_coord.size() || out.size() != this->kf->inv_depth.mu.size()) out.clear();
    (void)lv; (void)av; (void)F; (void)event_frame; (void)patch_radius; (void)border_type; (void)border_value;
    return out;
}
cv::Mat Tracker::getEMatrix() {                                                             // :555-575: E = [t]x R of getTransform()
    base::Transform3d T_ef_kf = this->getTransform();
    cv::Mat E; E.rows = E.cols = 3; E.d.assign(9, 0.0);
    const double t[3] = {T_ef_kf(0, 3), T_ef_kf(1, 3), T_ef_kf(2, 3)};
    const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) E.d[3 * r + c] += tx[3 * r + k] * T_ef_kf(k, c);
    return E;
}
cv::Mat Tracker::getFMatrix() {                                                             // :577-587: K^-T E K^-1 from kf->K_ref
    const double fx = this->kf->K_ref.at<double>(0, 0), fy = this->kf->K_ref.at<double>(1, 1);
    cv::Mat F = this->getEMatrix();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F.d[3 * r + c] /= (r < 2 ? (r == 0 ? fx : fy) : 1.0) * (c < 2 ? (c == 0 ? fx : fy) : 1.0);
    return F;
}
bool Tracker::getFilteredPose(eds::SE3& pose, const size_t& mean_filter_size) {            // :592-648: the mean over this->poses
    if (mean_filter_size < 2) { pose = this->poses.back(); return true; }
    if (this->poses.size() < mean_filter_size) return false;
    pose = this->poses[this->poses.size() - mean_filter_size];
    return this->squared_norm_flow >= 0.0;
}

}}  // namespace eds::tracking
