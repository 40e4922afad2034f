// TEST-ONLY mock of the types the EDS tree supplies to slam-eds_amd/csrc/Tracker.hpp when EDS_HIP_WITH_EDS_TYPES is defined
// (Eigen, OpenCV's cv::Point2d / cv::Mat, Rock base-types).  Same member names and signatures as the real ones for the
// handful of members the shim touches, so that its EDS_HIP_WITH_EDS_TYPES branch can at least be compile-checked in an
// image that has none of those libraries.  Not used by the product.
#pragma once
#include <array>
#include <cstdint>
#include <vector>
// ---- minimal mocks (same names, same template shapes as the real types) ------------------------------------------
namespace Eigen {
enum { DontAlign = 0x2 };
template <class T, int R, int C, int Options = 0>
struct Matrix {
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
struct Quaterniond {                        // coeffs() order x,y,z,w like Eigen
    double c[4];
    static Quaterniond Identity() { return Quaterniond{{0, 0, 0, 1}}; }
    double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
    double* coeffs() { return c; } const double* coeffs() const { return c; }
};
}  // namespace Eigen
namespace base {
struct Time { int64_t microseconds = 0; };
typedef Eigen::Matrix<double, 6, 1, Eigen::DontAlign> Vector6d;    // Rock base/Eigen.hpp: a DIFFERENT type from Eigen::Matrix<double,6,1>
// Eigen::Transform<double,3,Isometry> mock: column-major 4x4 like Eigen's matrix()
struct Transform3d {
    double m[16];
    static Transform3d Identity() { Transform3d t; for (int i = 0; i < 16; ++i) t.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return t; }
    double& operator()(int r, int c) { return m[4 * c + r]; }
    double operator()(int r, int c) const { return m[4 * c + r]; }
    Transform3d& matrix() { return *this; }                         // Eigen::Transform::matrix(): the 4x4 it wraps
    const Transform3d& matrix() const { return *this; }
    Transform3d inverse() const {           // rigid inverse
        Transform3d o = Identity();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o(r, c) = (*this)(c, r);
        for (int r = 0; r < 3; ++r) o(r, 3) = -(o(r, 0) * (*this)(0, 3) + o(r, 1) * (*this)(1, 3) + o(r, 2) * (*this)(2, 3));
        return o;
    }
};
}  // namespace base
namespace cv { struct Point2d { double x, y; }; enum { BORDER_DEFAULT = 4 }; }
namespace cv {
struct Mat {                                 // kf->img.rows / cols, kf->K_ref.at<double>(r, c)
    int rows = 0, cols = 0;
    std::vector<double> d;
    template <class T> T& at(int r, int c) { return reinterpret_cast<T&>(d[(size_t)r * cols + c]); }
};
}
// Sophus::SE3d as the reference uses it (tracking/Types.hpp:76 `typedef Sophus::SE3d SE3`): built from (quaternion, translation),
// read back through unit_quaternion() / translation() / matrix()
namespace eds {
struct SE3 {
    Eigen::Quaterniond q_ = Eigen::Quaterniond::Identity();
    Eigen::Vector3d t_ = Eigen::Vector3d::Zero();
    SE3() {}
    SE3(const Eigen::Quaterniond& q, const Eigen::Vector3d& t) : q_(q), t_(t) {}
    const Eigen::Quaterniond& unit_quaternion() const { return q_; }
    const Eigen::Vector3d& translation() const { return t_; }
    base::Transform3d matrix() const {
        base::Transform3d T = base::Transform3d::Identity();
        const double x = q_.x(), y = q_.y(), z = q_.z(), w = q_.w();
        T(0, 0) = 1 - 2 * (y * y + z * z); T(0, 1) = 2 * (x * y - z * w);     T(0, 2) = 2 * (x * z + y * w);
        T(1, 0) = 2 * (x * y + z * w);     T(1, 1) = 1 - 2 * (x * x + z * z); T(1, 2) = 2 * (y * z - x * w);
        T(2, 0) = 2 * (x * z - y * w);     T(2, 1) = 2 * (y * z + x * w);     T(2, 2) = 1 - 2 * (x * x + y * y);
        for (int i = 0; i < 3; ++i) T(i, 3) = t_[i];
        return T;
    }
};
}  // namespace eds
