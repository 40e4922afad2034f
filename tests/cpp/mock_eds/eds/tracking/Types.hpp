// TEST-ONLY mock of the types the EDS tree supplies to slam-eds_amd/csrc/Tracker.hpp when EDS_HIP_WITH_EDS_TYPES is defined
// (Eigen, OpenCV's cv::Point2d / cv::Mat, Rock base-types).  Same member names and signatures as the real ones for the
// handful of members the shim touches, so that its EDS_HIP_WITH_EDS_TYPES branch can at least be compile-checked in an
// image that has none of those libraries.  Not used by the product.
#pragma once
#include <array>
#include <cstdint>
#include <vector>
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
namespace Eigen { struct Vector2d { double v[2]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } }; }
namespace cv {
struct Mat {                                 // kf->img.rows / cols, kf->K_ref.at<double>(r, c)
    int rows = 0, cols = 0;
    std::vector<double> d;
    template <class T> T& at(int r, int c) { return reinterpret_cast<T&>(d[(size_t)r * cols + c]); }
};
}
