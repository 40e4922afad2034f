// Compile check of the EDS_HIP_WITH_EDS_TYPES branch of slam-eds_amd/csrc/Tracker.hpp against TEST-ONLY mocks of the EDS /
// Eigen / OpenCV / Rock types (tests/cpp/mock_eds).  Never run.
//
// Every mirrored member is taken as a pointer-to-member of EXACTLY the type the reference declares
// (/root/reference/src/tracking/Tracker.hpp:62-113): return type, parameter types and cv-qualification.  A shim member whose
// signature drifts (value instead of reference, a const member, base::Vector6d where the reference has Eigen::Matrix<double,6,1>)
// no longer converts and the build breaks.
#define EDS_HIP_WITH_EDS_TYPES
#define EDS_HIP_REFERENCE_MEMBERS          // the six out-of-path members declared with the reference's signatures (Tracker.hpp:98-111)
#include <type_traits>

#include "../../slam-eds_amd/csrc/Tracker.hpp"

namespace {
using T = eds::tracking::Tracker;
using KF = std::shared_ptr<eds::tracking::KeyFrame>;
using eds::tracking::LOSS_PARAM_METHOD;

// Tracker.hpp:62,65 — constructible from (kf, config) and from (config), the latter implicitly like the reference's non-explicit ctor
static_assert(std::is_constructible<T, KF, const eds::tracking::Config&>::value, "Tracker(kf, config)");
static_assert(std::is_convertible<const eds::tracking::Config&, T>::value || std::is_constructible<T, const eds::tracking::Config&>::value, "Tracker(config)");
// :40 public config of the reference type
static_assert(std::is_same<decltype(T::config), ::eds::tracking::Config>::value, "public Config config");

// :67
void (T::*p_reset1)(KF, const Eigen::Vector3d&, const Eigen::Quaterniond&, const bool&) = &T::reset;
// :69
void (T::*p_reset2)(KF, const Eigen::Vector3d&, const Eigen::Quaterniond&, const base::Vector6d&) = &T::reset;
// :71
void (T::*p_set)(const base::Transform3d&) = &T::set;
// :73-75
void (T::*p_opt1)(const int&, const std::vector<double>*, ::base::Transform3d&, const Eigen::Vector3d&, const Eigen::Quaterniond&,
                  const LOSS_PARAM_METHOD) = &T::optimize;
// :77-78
void (T::*p_opt2)(const int&, const std::vector<double>*, ::base::Transform3d&, const Eigen::Matrix<double, 6, 1>&, const LOSS_PARAM_METHOD) = &T::optimize;
// :80-81
bool (T::*p_opt3)(const int&, const std::vector<double>*, ::base::Transform3d&, const LOSS_PARAM_METHOD) = &T::optimize;
// :83
::base::Transform3d (T::*p_gt0)() = &T::getTransform;
// :85
::base::Transform3d (T::*p_gt1)(bool&) = &T::getTransform;
// :87 — by REFERENCE: callers write through it
Eigen::Matrix<double, 6, 1>& (T::*p_vel)() = &T::getVelocity;
// :89,91
const Eigen::Vector3d (T::*p_lin)() = &T::linearVelocity;
const Eigen::Vector3d (T::*p_ang)() = &T::angularVelocity;
// :93
std::vector<double> (T::*p_lp)(LOSS_PARAM_METHOD) = &T::getLossParams;
// :96
std::vector<cv::Point2d> (T::*p_coord)(const bool&) = &T::getCoord;
// :109
::eds::tracking::TrackerInfo (T::*p_info)() = &T::getInfo;
// :113
bool (T::*p_need)(const double&) = &T::needNewKeyframe;

// :98-111 — the members OUTSIDE the hot path, declared by the shim so that the reference's own definitions (Tracker.cpp:378-648) compile
// against it unchanged and callers relink unchanged; tests/cpp/shim_reference_members.cpp is such a translation unit
void (T::*p_track)(const cv::Mat&, const uint16_t&) = &T::trackPoints;                                                     // :98
void (T::*p_track_pyr)(const cv::Mat&, const size_t) = &T::trackPointsPyr;                                                 // :100
std::vector<cv::Point2d> (T::*p_track_epi)(const cv::Mat&, const uint16_t&, const int&, const uint8_t&) = &T::trackPointsAlongEpiline;   // :102-103
cv::Mat (T::*p_emat)() = &T::getEMatrix;                                                                                   // :105
cv::Mat (T::*p_fmat)() = &T::getFMatrix;                                                                                   // :107
bool (T::*p_filt)(eds::SE3&, const size_t&) = &T::getFilteredPose;                                                         // :111

// :58 squared_norm_flow is private in the reference: it must not be reachable as a public data member here either
template <class U, class = void> struct has_public_sq_flow : std::false_type {};
template <class U> struct has_public_sq_flow<U, decltype(void(std::declval<U&>().squared_norm_flow))> : std::true_type {};
static_assert(!has_public_sq_flow<T>::value, "squared_norm_flow is private (Tracker.hpp:58)");
}  // namespace

int shim_eds_types_check(std::shared_ptr<eds::tracking::KeyFrame> kf, const std::vector<double>* frame) {
    eds::tracking::Config cfg;
    eds::tracking::Tracker a(kf, cfg), b(cfg);
    base::Transform3d Tm = base::Transform3d::Identity();
    base::Vector6d velo = base::Vector6d::Zero();
    Eigen::Matrix<double, 6, 1> v6 = Eigen::Matrix<double, 6, 1>::Zero();
    a.reset(kf, Eigen::Vector3d::Zero(), Eigen::Quaterniond::Identity());          // default keep_velo = true (:67)
    a.reset(kf, Eigen::Vector3d::Zero(), Eigen::Quaterniond::Identity(), velo);
    a.set(Tm);
    bool ok = a.optimize(0, frame, Tm);                                             // default method = MAD (:81)
    a.optimize(0, frame, Tm, Eigen::Vector3d::Zero(), Eigen::Quaterniond::Identity(), eds::tracking::MAD);
    a.optimize(0, frame, Tm, v6, eds::tracking::STD);
    Tm = a.getTransform();
    bool filtered = false;
    Tm = a.getTransform(filtered);
    a.getVelocity()[0] = 1.0;                                                       // writable through the reference
    const Eigen::Vector3d lv = a.linearVelocity(), av = a.angularVelocity();
    const std::vector<double> lp = a.getLossParams();                               // default CONSTANT (:93)
    const std::vector<cv::Point2d> c = a.getCoord();                                // default false (:96)
    const eds::tracking::TrackerInfo info = a.getInfo();
    (void)p_reset1; (void)p_reset2; (void)p_set; (void)p_opt1; (void)p_opt2; (void)p_opt3; (void)p_gt0; (void)p_gt1; (void)p_vel; (void)p_lin;
    (void)p_ang; (void)p_lp; (void)p_coord; (void)p_info; (void)p_need;
    (void)p_track; (void)p_track_pyr; (void)p_track_epi; (void)p_emat; (void)p_fmat; (void)p_filt;
    cv::Mat ef;
    a.trackPoints(ef); a.trackPointsPyr(ef); (void)a.trackPointsAlongEpiline(ef);       // the reference's default arguments (:98-103)
    eds::SE3 fp;
    filtered = a.getFilteredPose(fp) || filtered;                                       // default mean_filter_size = 3 (:111)
    (void)a.getEMatrix(); (void)a.getFMatrix();
    return (int)ok + (int)filtered + (int)a.needNewKeyframe() + (int)lp.size() + (int)c.size() + info.num_iterations + (int)(lv[0] + av[0]);
}
