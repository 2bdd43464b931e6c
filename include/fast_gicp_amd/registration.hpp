// Host-side mirror of the reference's registration classes on top of the C ABI (fast_vgicp_hip.h).
//
//   fast_gicp::LsqRegistration   <- include/fast_gicp/gicp/lsq_registration.hpp + impl/lsq_registration_impl.hpp
//   fast_gicp::FastVGICPCuda     <- include/fast_gicp/gicp/fast_vgicp_cuda.hpp  + impl/fast_vgicp_cuda_impl.hpp
//   fast_gicp::NDTCuda           <- include/fast_gicp/ndt/ndt_cuda.hpp          + impl/ndt_cuda_impl.hpp
//
// Same class / method / enum names, argument meaning and defaults as the reference, so code written
// against it (gicp_align, gicp_test, pygicp) reads the same. PCL and Eigen are not available in this
// build environment (the reference does not vendor them), so the PCL base class is replaced by the
// few members of pcl::Registration the reference's callers use, on a minimal PointCloud / Matrix4
// type. INTEGRATION.md shows the 1:1 PCL-templated shim for a tree that has PCL.
//
// Two ways to run the optimiser:
//   * setUseDeviceLM(true)  (default): fvh_*_align -- the whole LM loop on the GPU, one D2H.
//   * setUseDeviceLM(false): the reference's host loop (step_lm below) calling the virtuals
//     linearize() / compute_error() -> fvh_*_update_correspondences / fvh_*_compute_error,
//     i.e. exactly the reference's call sequence. Both give the same result to fp64 rounding.
#pragma once
#if defined(__SSE2__)
#include <immintrin.h>
#endif
#include <array>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../fast_vgicp_hip.h"
#include "kdtree.hpp"

namespace fast_gicp {

// gicp_settings.hpp:7-11, ndt_settings.hpp:6, fast_vgicp_cuda.hpp:18 (same ordinals)
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS };
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };
enum class NDTDistanceMode { P2D, D2D };
enum class NearestNeighborMethod { CPU_PARALLEL_KDTREE, GPU_BRUTEFORCE, GPU_RBF_KERNEL };
enum class LSQ_OPTIMIZER_TYPE { GaussNewton, LevenbergMarquardt };

struct PointXYZ {
  float x = 0, y = 0, z = 0;
};

template <typename PointT>
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  std::vector<PointT> points;
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  const PointT& at(size_t i) const { return points.at(i); }
  void resize(size_t n) { points.resize(n); }
};

struct Matrix4f {  // row-major 4x4 (Eigen::Matrix4f stand-in)
  float m[16];
  static Matrix4f Identity() { Matrix4f I; std::memset(I.m, 0, sizeof(I.m)); I.m[0] = I.m[5] = I.m[10] = I.m[15] = 1.f; return I; }
  float& operator()(int r, int c) { return m[r * 4 + c]; }
  float operator()(int r, int c) const { return m[r * 4 + c]; }
};

using Matrix6d = std::array<double, 36>;  // row-major (symmetric)
using Vector6d = std::array<double, 6>;

struct Isometry3d {  // row-major R | t, double
  double R[9];
  double t[3];
  static Isometry3d Identity() { Isometry3d T; std::memset(&T, 0, sizeof(T)); T.R[0] = T.R[4] = T.R[8] = 1.0; return T; }
  static Isometry3d from(const Matrix4f& M) {
    Isometry3d T;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[i * 3 + j] = M(i, j); T.t[i] = M(i, 3); }
    return T;
  }
  Matrix4f cast_float() const {
    Matrix4f M = Matrix4f::Identity();
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M(i, j) = (float)R[i * 3 + j]; M(i, 3) = (float)t[i]; }
    return M;
  }
  void to_colmajor16(double* T16) const {
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T16[j * 4 + i] = R[i * 3 + j]; T16[12 + i] = t[i]; T16[i * 4 + 3] = 0.0; }
    T16[15] = 1.0;
  }
  static Isometry3d from_colmajor16(const double* T16) {
    Isometry3d T;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[i * 3 + j] = T16[j * 4 + i]; T.t[i] = T16[12 + i]; }
    return T;
  }
  Isometry3d operator*(const Isometry3d& B) const {
    Isometry3d C;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) C.R[i * 3 + j] = R[i * 3] * B.R[j] + R[i * 3 + 1] * B.R[3 + j] + R[i * 3 + 2] * B.R[6 + j];
      C.t[i] = R[i * 3] * B.t[0] + R[i * 3 + 1] * B.t[1] + R[i * 3 + 2] * B.t[2] + t[i];
    }
    return C;
  }
};

// so3.hpp:58-104 (rotation first)
inline Isometry3d se3_exp(const Vector6d& a) {
  const double ox = a[0], oy = a[1], oz = a[2];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - theta_sq / 48.0 + tq / 3840.0;
    real = 1.0 - theta_sq / 8.0 + tq / 384.0;
  } else {
    const double th = std::sqrt(theta_sq);
    imag = std::sin(0.5 * th) / th;
    real = std::cos(0.5 * th);
  }
  const double qw = real, qx = imag * ox, qy = imag * oy, qz = imag * oz;
  Isometry3d T;
  T.R[0] = 1 - 2 * (qy * qy + qz * qz); T.R[1] = 2 * (qx * qy - qz * qw);     T.R[2] = 2 * (qx * qz + qy * qw);
  T.R[3] = 2 * (qx * qy + qz * qw);     T.R[4] = 1 - 2 * (qx * qx + qz * qz); T.R[5] = 2 * (qy * qz - qx * qw);
  T.R[6] = 2 * (qx * qz - qy * qw);     T.R[7] = 2 * (qy * qz + qx * qw);     T.R[8] = 1 - 2 * (qx * qx + qy * qy);
  const double theta = std::sqrt(theta_sq);
  double V[9];
  if (theta < 1e-10) {
    std::memcpy(V, T.R, sizeof(V));
  } else {
    const double A = (1.0 - std::cos(theta)) / theta_sq, B = (theta - std::sin(theta)) / (theta_sq * theta);
    const double O[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double o2 = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
        V[i * 3 + j] = (i == j ? 1.0 : 0.0) + A * O[i * 3 + j] + B * o2;
      }
  }
  for (int i = 0; i < 3; i++) T.t[i] = V[i * 3] * a[3] + V[i * 3 + 1] * a[4] + V[i * 3 + 2] * a[5];
  return T;
}

namespace detail {
// A pivot with |d| <= DBL_MIN is handled like Eigen::LDLT (column left unscaled, pseudo-inverse of D in the solve): an
// all-zero system (no correspondences) gives d = 0 and the optimiser returns the guess, converged, as the reference does.
inline void ldlt6_solve(const Matrix6d& A, const Vector6d& rhs, Vector6d& x) {
  double L[36] = {0}, D[6], y[6];
  constexpr double tiny = 2.2250738585072014e-308;
  for (int j = 0; j < 6; j++) {
    double d = A[j * 6 + j];
    for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k] * D[k];
    D[j] = d;
    L[j * 6 + j] = 1.0;
    for (int i = j + 1; i < 6; i++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k] * D[k];
      L[i * 6 + j] = std::fabs(d) > tiny ? s / d : s;
    }
  }
  for (int i = 0; i < 6; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] = std::fabs(D[i]) > tiny ? y[i] / D[i] : 0.0;
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s; }
}
/// the table LsqRegistration::step_lm prints with setDebugPrint(true) (lsq_registration_impl.hpp:143-149), from the rows the
/// device LM recorded ({i, y0, yi, rho, lambda, |delta|} per trial step)
inline void print_lm_trace(const std::vector<double>& rows) {
  for (size_t r = 0; r + 5 < rows.size(); r += 6) {
    if ((int)rows[r] == 0) std::printf("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n", "i", "y0", "yi", "rho", "lambda", "|delta|", "dec");
    std::printf("%5d %15g %15g %15g %15g %15g %5c\n", (int)rows[r], rows[r + 1], rows[r + 2], rows[r + 3], rows[r + 4], rows[r + 5], rows[r + 3] > 0.0 ? 'x' : ' ');
  }
  std::fflush(stdout);  // (std::endl in the reference)
}
/// xyz of a cloud as the C ABI wants it: when the point type starts with three packed floats and is 12 or 16 bytes (PointXYZ
/// here, pcl::PointXYZ upstream) the cloud's own storage is handed over with its stride -- no 200 KB repack per setInputSource;
/// any other point type is packed into `scratch` (stride 3)
template <typename PointT>
struct XyzView {
  const float* data;
  int stride;
  XyzView(const PointCloud<PointT>& c, std::vector<float>& scratch) {
    constexpr bool direct = (sizeof(PointT) == 12 || sizeof(PointT) == 16) && offsetof(PointT, x) == 0 && offsetof(PointT, y) == 4 && offsetof(PointT, z) == 8;
    if (direct) {
      data = c.size() ? &c.points[0].x : nullptr;
      stride = (int)(sizeof(PointT) / sizeof(float));
    } else {
      scratch.resize(c.size() * 3);
      for (size_t i = 0; i < c.size(); i++) { scratch[3 * i] = c.points[i].x; scratch[3 * i + 1] = c.points[i].y; scratch[3 * i + 2] = c.points[i].z; }
      data = scratch.data();
      stride = 3;
    }
  }
};
// out[i] = in[i] with (x, y, z) <- M (x, y, z, 1): pcl::transformPointCloud for the output cloud of align(). It runs between the
// end of one registration and the first launch of the next with the GPU idle, so packed xyz / xyzw points (12- or 16-byte structs)
// go four at a time through SSE (the scalar loop: 50 us for 17k points, ~15 % of a registration). Every lane evaluates
// ((m0 x + m1 y) + m2 z) + m3 in that order without fusing: bit for bit the scalar loop's result.
template <typename PointT>
inline void transform_points_scalar(const PointT* in, PointT* out, size_t begin, size_t n, const float* m) {
  for (size_t i = begin; i < n; i++) {
    PointT q = in[i];
    const float x = q.x, y = q.y, z = q.z;
    q.x = m[0] * x + m[1] * y + m[2] * z + m[3];
    q.y = m[4] * x + m[5] * y + m[6] * z + m[7];
    q.z = m[8] * x + m[9] * y + m[10] * z + m[11];
    out[i] = q;
  }
}
#if defined(__SSE2__)
inline void fvh_sse_rows(const float* m, __m128 X, __m128 Y, __m128 Z, __m128& ox, __m128& oy, __m128& oz) {
  auto row = [&](int r) {
    const __m128 a = _mm_mul_ps(_mm_set1_ps(m[4 * r]), X), b = _mm_mul_ps(_mm_set1_ps(m[4 * r + 1]), Y), c = _mm_mul_ps(_mm_set1_ps(m[4 * r + 2]), Z);
    return _mm_add_ps(_mm_add_ps(_mm_add_ps(a, b), c), _mm_set1_ps(m[4 * r + 3]));
  };
  ox = row(0); oy = row(1); oz = row(2);
}
#endif
template <typename PointT>
inline void transform_points(const PointT* in, PointT* out, size_t n, const float* m) {
  size_t done = 0;
#if defined(__SSE2__)
  constexpr bool xyz_first = offsetof(PointT, x) == 0 && offsetof(PointT, y) == 4 && offsetof(PointT, z) == 8;
  if constexpr (xyz_first && sizeof(PointT) == 16) {  // x y z w: a 4 x 4 transpose each way
    const float* src = reinterpret_cast<const float*>(in);
    float* dst = reinterpret_cast<float*>(out);
    for (; done + 4 <= n; done += 4) {
      __m128 p0 = _mm_loadu_ps(src + 4 * done), p1 = _mm_loadu_ps(src + 4 * done + 4), p2 = _mm_loadu_ps(src + 4 * done + 8), p3 = _mm_loadu_ps(src + 4 * done + 12);
      _MM_TRANSPOSE4_PS(p0, p1, p2, p3);  // p0 = x's, p1 = y's, p2 = z's, p3 = w's
      __m128 ox, oy, oz;
      fvh_sse_rows(m, p0, p1, p2, ox, oy, oz);
      _MM_TRANSPOSE4_PS(ox, oy, oz, p3);
      _mm_storeu_ps(dst + 4 * done, ox); _mm_storeu_ps(dst + 4 * done + 4, oy); _mm_storeu_ps(dst + 4 * done + 8, oz); _mm_storeu_ps(dst + 4 * done + 12, p3);
    }
  } else if constexpr (xyz_first && sizeof(PointT) == 12) {  // x y z packed: 4 points = 3 vectors
    const float* src = reinterpret_cast<const float*>(in);
    float* dst = reinterpret_cast<float*>(out);
    for (; done + 4 <= n; done += 4) {
      const __m128 a = _mm_loadu_ps(src + 3 * done);      // x0 y0 z0 x1
      const __m128 b = _mm_loadu_ps(src + 3 * done + 4);  // y1 z1 x2 y2
      const __m128 c = _mm_loadu_ps(src + 3 * done + 8);  // z2 x3 y3 z3
      const __m128 x2y2x3y3 = _mm_shuffle_ps(b, c, _MM_SHUFFLE(2, 1, 3, 2));
      const __m128 y0z0y1z1 = _mm_shuffle_ps(a, b, _MM_SHUFFLE(1, 0, 2, 1));
      const __m128 X = _mm_shuffle_ps(a, x2y2x3y3, _MM_SHUFFLE(2, 0, 3, 0));         // x0 x1 x2 x3
      const __m128 Y = _mm_shuffle_ps(y0z0y1z1, x2y2x3y3, _MM_SHUFFLE(3, 1, 2, 0));  // y0 y1 y2 y3
      const __m128 Z = _mm_shuffle_ps(y0z0y1z1, c, _MM_SHUFFLE(3, 0, 3, 1));         // z0 z1 z2 z3
      __m128 ox, oy, oz;
      fvh_sse_rows(m, X, Y, Z, ox, oy, oz);
      // back to x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
      const __m128 x0x2y0y2 = _mm_shuffle_ps(ox, oy, _MM_SHUFFLE(2, 0, 2, 0));
      const __m128 y1y3z1z3 = _mm_shuffle_ps(oy, oz, _MM_SHUFFLE(3, 1, 3, 1));
      const __m128 z0z2x1x3 = _mm_shuffle_ps(oz, ox, _MM_SHUFFLE(3, 1, 2, 0));
      const __m128 ra = _mm_shuffle_ps(x0x2y0y2, z0z2x1x3, _MM_SHUFFLE(2, 0, 2, 0));  // x0 y0 z0 x1
      const __m128 rb = _mm_shuffle_ps(y1y3z1z3, x0x2y0y2, _MM_SHUFFLE(3, 1, 2, 0));  // y1 z1 x2 y2
      const __m128 rc = _mm_shuffle_ps(z0z2x1x3, y1y3z1z3, _MM_SHUFFLE(3, 1, 3, 1));  // z2 x3 y3 z3
      _mm_storeu_ps(dst + 3 * done, ra); _mm_storeu_ps(dst + 3 * done + 4, rb); _mm_storeu_ps(dst + 3 * done + 8, rc);
    }
  }
#endif
  transform_points_scalar(in, out, done, n, m);
}

template <typename PointT>
inline std::vector<float> pack_xyz(const PointCloud<PointT>& c) {
  std::vector<float> xyz(c.size() * 3);
  for (size_t i = 0; i < c.size(); i++) { xyz[3 * i] = c.points[i].x; xyz[3 * i + 1] = c.points[i].y; xyz[3 * i + 2] = c.points[i].z; }
  return xyz;
}
/// where align() spends its host time (apps/gicp_align, GICP_ALIGN_BREAKDOWN): off unless a caller switches it on
struct HostTiming { bool on = false; double optimize_us = 0, transform_us = 0; };
inline HostTiming& host_timing() { static HostTiming t; return t; }
}  // namespace detail

/// LsqRegistration (lsq_registration.hpp:14-83) minus the PCL base: the members of pcl::Registration its callers use.
template <typename PointSource, typename PointTarget>
class LsqRegistration {
public:
  using PointCloudSource = PointCloud<PointSource>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTarget = PointCloud<PointTarget>;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;

  LsqRegistration() { final_hessian_.fill(0.0); for (int i = 0; i < 6; i++) final_hessian_[i * 7] = 1.0; }  // lsq_registration_impl.hpp:9-22
  virtual ~LsqRegistration() {}

  void setRotationEpsilon(double eps) { rotation_epsilon_ = eps; }
  void setTransformationEpsilon(double eps) { transformation_epsilon_ = eps; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setInitialLambdaFactor(double f) { lm_init_lambda_factor_ = f; }
  void setDebugPrint(bool p) { lm_debug_print_ = p; }
  void setUseDeviceLM(bool on) { use_device_lm_ = on; }
  /// lsq_optimizer_type_ (lsq_registration.hpp:78: protected there, set by subclasses; LevenbergMarquardt by default, :15). Gauss-Newton
  /// (step_gn, :108-121) is device-resident too (fvh_lm_params::optimizer); setUseDeviceLM(false) runs the reference's host loop on the device's linearize().
  void setLSQType(LSQ_OPTIMIZER_TYPE type) { lsq_optimizer_type_ = type; }
  const Matrix6d& getFinalHessian() const { return final_hessian_; }
  const Matrix4f& getFinalTransformation() const { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  int getNumIterations() const { return nr_iterations_; }

  double evaluateCost(const Matrix4f& relative_pose, Matrix6d* H = nullptr, Vector6d* b = nullptr) { return this->linearize(Isometry3d::from(relative_pose), H, b); }

  virtual void swapSourceAndTarget() {}
  virtual void clearSource() {}
  virtual void clearTarget() {}
  virtual void setInputSource(const PointCloudSourceConstPtr& cloud) { input_ = cloud; }
  virtual void setInputTarget(const PointCloudTargetConstPtr& cloud) { target_ = cloud; }
  PointCloudSourceConstPtr getInputSource() const { return input_; }
  PointCloudTargetConstPtr getInputTarget() const { return target_; }

  /// pcl::Registration::align
  void align(PointCloudSource& output, const Matrix4f& guess = Matrix4f::Identity()) {
    if (!input_ || !target_) throw std::invalid_argument("align: source / target cloud not set");
    computeTransformation(output, guess);
  }
  /// pcl::Registration::getFitnessScore(max_range): mean squared exact-NN distance of T*source to the target
  virtual double getFitnessScore(double max_range = std::numeric_limits<double>::max()) = 0;

protected:
  virtual void computeTransformation(PointCloudSource& output, const Matrix4f& guess) {  // lsq_registration_impl.hpp:53-79
    Isometry3d x0 = Isometry3d::from(guess);
    lm_lambda_ = -1.0;
    converged_ = false;
    detail::HostTiming& ht = detail::host_timing();
    std::chrono::steady_clock::time_point ht0, ht1;
    if (ht.on) ht0 = std::chrono::steady_clock::now();
    if (use_device_lm_ && device_align(x0)) {  // (both optimiser types: Levenberg-Marquardt and, since round 5, Gauss-Newton run inside the kernel)
      // whole loop ran on the GPU
    } else {
      for (int i = 0; i < max_iterations_ && !converged_; i++) {
        nr_iterations_ = i;
        Isometry3d delta;
        if (!step_optimize(x0, delta)) {
          std::fprintf(stderr, "lm not converged!!\n");
          break;
        }
        converged_ = is_converged(delta);
      }
    }
    final_transformation_ = x0.cast_float();
    if (ht.on) ht1 = std::chrono::steady_clock::now();
    output.points.resize(input_->size());  // pcl::transformPointCloud(*input_, output, final_transformation_)
    detail::transform_points(input_->points.data(), output.points.data(), input_->size(), final_transformation_.m);
    if (ht.on) {
      ht.optimize_us += std::chrono::duration<double, std::micro>(ht1 - ht0).count();
      ht.transform_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht1).count();
    }
  }

  bool is_converged(const Isometry3d& delta) const {  // :82-91
    double rmax = 0, tmax = 0;
    for (int i = 0; i < 9; i++) rmax = std::max(rmax, std::fabs(delta.R[i] - (i % 4 == 0 ? 1.0 : 0.0)) / rotation_epsilon_);
    for (int i = 0; i < 3; i++) tmax = std::max(tmax, std::fabs(delta.t[i]) / transformation_epsilon_);
    return std::max(rmax, tmax) < 1;
  }

  virtual double linearize(const Isometry3d& trans, Matrix6d* H = nullptr, Vector6d* b = nullptr) = 0;
  virtual double compute_error(const Isometry3d& trans) = 0;
  /// subclasses with a device-resident LM return true after updating x0 / converged_ / nr_iterations_ / final_hessian_
  virtual bool device_align(Isometry3d& x0) { (void)x0; return false; }

  bool step_optimize(Isometry3d& x0, Isometry3d& delta) {  // :94-104
    return lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? step_gn(x0, delta) : step_lm(x0, delta);
  }
  bool step_gn(Isometry3d& x0, Isometry3d& delta) {  // :108-121
    Matrix6d H;
    Vector6d b, nb, d;
    linearize(x0, &H, &b);
    for (int j = 0; j < 6; j++) nb[j] = -b[j];
    detail::ldlt6_solve(H, nb, d);
    delta = se3_exp(d);
    x0 = delta * x0;
    final_hessian_ = H;
    return true;
  }
  bool step_lm(Isometry3d& x0, Isometry3d& delta) {  // :123-168
    Matrix6d H;
    Vector6d b;
    const double y0 = linearize(x0, &H, &b);
    if (lm_lambda_ < 0.0) {
      double mx = 0;
      for (int i = 0; i < 6; i++) mx = std::max(mx, std::fabs(H[i * 7]));
      lm_lambda_ = lm_init_lambda_factor_ * mx;
    }
    double nu = 2.0;
    for (int i = 0; i < lm_max_iterations_; i++) {
      Matrix6d A = H;
      Vector6d nb, d;
      for (int j = 0; j < 6; j++) { A[j * 7] += lm_lambda_; nb[j] = -b[j]; }
      detail::ldlt6_solve(A, nb, d);
      delta = se3_exp(d);
      const Isometry3d xi = delta * x0;
      const double yi = compute_error(xi);
      double denom = 0;
      for (int j = 0; j < 6; j++) denom += d[j] * (lm_lambda_ * d[j] - b[j]);
      const double rho = (y0 - yi) / denom;
      if (lm_debug_print_) {
        double dn = 0;
        for (int j = 0; j < 6; j++) dn += d[j] * d[j];
        if (i == 0) std::printf("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n", "i", "y0", "yi", "rho", "lambda", "|delta|", "dec");
        std::printf("%5d %15g %15g %15g %15g %15g %5c\n", i, y0, yi, rho, lm_lambda_, std::sqrt(dn), rho > 0.0 ? 'x' : ' ');
        std::fflush(stdout);
      }
      if (rho < 0) {
        if (is_converged(delta)) return true;
        lm_lambda_ = nu * lm_lambda_;
        nu = 2 * nu;
        continue;
      }
      x0 = xi;
      lm_lambda_ = lm_lambda_ * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
      final_hessian_ = H;
      return true;
    }
    return false;
  }

protected:
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Matrix4f final_transformation_ = Matrix4f::Identity();
  int nr_iterations_ = 0;
  int max_iterations_ = 64;               // :11
  double transformation_epsilon_ = 5e-4;  // :13
  bool converged_ = false;
  double rotation_epsilon_ = 2e-3;        // :12
  LSQ_OPTIMIZER_TYPE lsq_optimizer_type_ = LSQ_OPTIMIZER_TYPE::LevenbergMarquardt;
  int lm_max_iterations_ = 10;            // :17
  double lm_init_lambda_factor_ = 1e-9;   // :18
  double lm_lambda_ = -1.0;
  bool lm_debug_print_ = false;
  bool use_device_lm_ = true;
  Matrix6d final_hessian_;
};

namespace detail {
inline void check(int rc, const char* what, const char* err) {
  if (rc != 0) throw std::runtime_error(std::string(what) + " failed (status " + std::to_string(rc) + "): " + (err ? err : ""));
}
}  // namespace detail

/// FastVGICPCuda (fast_vgicp_cuda.hpp:24-85, impl/fast_vgicp_cuda_impl.hpp)
template <typename PointSource, typename PointTarget>
class FastVGICPCuda : public LsqRegistration<PointSource, PointTarget> {
  using Base = LsqRegistration<PointSource, PointTarget>;
  using Base::input_;
  using Base::target_;

public:
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;

  explicit FastVGICPCuda(int device = 0) {  // fast_vgicp_cuda_impl.hpp:21-32
    detail::check(fvh_vgicp_create(device, &core_), "fvh_vgicp_create", "cannot create the HIP engine (no GPU? there is no CPU fallback)");
    call(fvh_vgicp_set_resolution(core_, voxel_resolution_), "set_resolution");
    call(fvh_vgicp_set_kernel_params(core_, 0.5, 3.0), "set_kernel_params");
  }
  ~FastVGICPCuda() override { if (core_) fvh_vgicp_destroy(core_); }
  FastVGICPCuda(const FastVGICPCuda&) = delete;
  FastVGICPCuda& operator=(const FastVGICPCuda&) = delete;

  void setCorrespondenceRandomness(int) {}  // empty in the reference too (:38): k stays 20
  void setResolution(double resolution) { call(fvh_vgicp_set_resolution(core_, resolution), "set_resolution"); }
  /// FastVGICP::setVoxelAccumulationMode (fast_vgicp_impl.hpp:41-43; CPU class only in the reference -- the CUDA core has the
  /// additive voxel alone): takes effect when the target voxel map is next built (setInputTarget / swapSourceAndTarget)
  void setVoxelAccumulationMode(VoxelAccumulationMode mode) {
    call(fvh_vgicp_set_voxel_accumulation_mode(core_, static_cast<int>(mode)), "set_voxel_accumulation_mode");
    if (this->target_) call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");  // (setInputTarget left cloud + covariances on the device)
  }
  void setKernelWidth(double kernel_width, double max_dist = -1.0) {  // :45-51
    if (max_dist <= 0.0) max_dist = kernel_width * 5.0;
    call(fvh_vgicp_set_kernel_params(core_, kernel_width, max_dist), "set_kernel_params");
  }
  void setRegularizationMethod(RegularizationMethod method) { regularization_method_ = method; }
  void setNeighborSearchMethod(NeighborSearchMethod method, double radius = -1.0) { call(fvh_vgicp_set_neighbor_search_method(core_, (int)method, radius), "set_neighbor_search_method"); }
  void setNearestNeighborSearchMethod(NearestNeighborMethod method) { neighbor_search_method_ = method; }
  /// NearestNeighborMethod::CPU_PARALLEL_KDTREE -- the reference's DEFAULT (fast_vgicp_cuda_impl.hpp:27,152-167) -- asks for EXACT k-NN lists from a
  /// host kd-tree. The engine's device search returns the same lists (fp32 distances, ties to the lower index: equal to the kd-tree's at 17k / 100k /
  /// 1M points, tests/test_gpu_parity*.py, test_neighbor_methods_agree) in 35 us instead of the ~4 ms a host tree + a PCIe round trip of N x k
  /// indices costs per cloud, so by default that enum value is SERVED BY THE DEVICE. setHostKdTree(true) (or FVH_HOST_KDTREE=1 in the
  /// environment) restores the host tree of kdtree.hpp for callers who want the reference's data flow.
  void setHostKdTree(bool on) { host_kdtree_ = on; }
  bool getHostKdTree() const { return host_kdtree_; }
  void setComputePrecision(int fvh_precision) { call(fvh_vgicp_set_precision(core_, fvh_precision), "set_precision"); }

  void swapSourceAndTarget() override {  // :69-72
    call(fvh_vgicp_swap_source_and_target(core_), "swap_source_and_target");
    input_.swap(target_);
  }
  void clearSource() override { input_.reset(); }
  void clearTarget() override { target_.reset(); }

  void setInputSource(const PointCloudSourceConstPtr& cloud) override {  // :85-111
    if (cloud == input_) return;
    input_ = cloud;
    const detail::XyzView<PointSource> view(*cloud, scratch_xyz_);
    call(fvh_vgicp_set_source_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_source_cloud");
    switch (effective_neighbor_method(cloud->size())) {
      case NearestNeighborMethod::CPU_PARALLEL_KDTREE: {
        const std::vector<float> xyz = detail::pack_xyz(*cloud);
        const std::vector<int> nb = find_neighbors_parallel_kdtree(k_correspondences_, xyz);
        call(fvh_vgicp_set_source_neighbors(core_, k_correspondences_, nb.data()), "set_source_neighbors");
        call(fvh_vgicp_calculate_source_covariances(core_, (int)regularization_method_), "calculate_source_covariances");
      } break;
      case NearestNeighborMethod::GPU_BRUTEFORCE:
        call(fvh_vgicp_find_source_neighbors(core_, k_correspondences_), "find_source_neighbors");
        call(fvh_vgicp_calculate_source_covariances(core_, (int)regularization_method_), "calculate_source_covariances");
        break;
      case NearestNeighborMethod::GPU_RBF_KERNEL:
        call(fvh_vgicp_calculate_source_covariances_rbf(core_, (int)regularization_method_), "calculate_source_covariances_rbf");
        break;
    }
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {  // :114-141
    if (cloud == target_) return;
    target_ = cloud;
    const detail::XyzView<PointTarget> view(*cloud, scratch_xyz_);
    call(fvh_vgicp_set_target_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_target_cloud");
    switch (effective_neighbor_method(cloud->size())) {
      case NearestNeighborMethod::CPU_PARALLEL_KDTREE: {
        const std::vector<float> xyz = detail::pack_xyz(*cloud);
        const std::vector<int> nb = find_neighbors_parallel_kdtree(k_correspondences_, xyz);
        call(fvh_vgicp_set_target_neighbors(core_, k_correspondences_, nb.data()), "set_target_neighbors");
        call(fvh_vgicp_calculate_target_covariances(core_, (int)regularization_method_), "calculate_target_covariances");
      } break;
      case NearestNeighborMethod::GPU_BRUTEFORCE:
        call(fvh_vgicp_find_target_neighbors(core_, k_correspondences_), "find_target_neighbors");
        call(fvh_vgicp_calculate_target_covariances(core_, (int)regularization_method_), "calculate_target_covariances");
        break;
      case NearestNeighborMethod::GPU_RBF_KERNEL:
        call(fvh_vgicp_calculate_target_covariances_rbf(core_, (int)regularization_method_), "calculate_target_covariances_rbf");
        break;
    }
    call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");
  }

  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) override {
    double T16[16], score = 0;
    Isometry3d::from(this->final_transformation_).to_colmajor16(T16);
    call(fvh_vgicp_fitness_score(core_, T16, max_range, &score), "fitness_score");
    return score;
  }
  fvh_vgicp* core() { return core_; }

  // ---- scan streams as a two-stage pipeline (no reference counterpart; C ABI: fvh_vgicp_align_async / _wait, fvh_vgicp_prepare_source_device /
  // _adopt_prepared_source). kitti.cpp:95-128 with the preparation of scan k+1 hidden under the registration of scan k:
  //     alignAsync(); prepareNextSourceDevice(next scan); alignWait(); swapSourceAndTarget(); adoptPreparedSource();
  // The prepared source lives on the device only: getInputSource() is null for it and alignWait() returns the pose without transforming a host
  // cloud. Neighbours and covariances follow setNearestNeighborSearchMethod (the host kd-tree cannot be served here: device search instead,
  // identical lists) and setRegularizationMethod. `stages`: see fvh_vgicp_prepare_source_device (2: order, neighbours, covariances).
  void prepareNextSourceDevice(const float* d_xyz, int n, int stride_floats = 3, int stages = 2) {
    const int rbf = neighbor_search_method_ == NearestNeighborMethod::GPU_RBF_KERNEL ? 1 : 0;
    prepared_stages_ = stages;
    prepared_cloud_.reset();
    call(fvh_vgicp_prepare_source_device(core_, d_xyz, n, stride_floats, k_correspondences_, (int)regularization_method_, rbf, stages), "prepare_source_device");
  }
  /// the same for a host cloud (consumed before the call returns); adoptPreparedSource() then makes it getInputSource()
  void prepareNextSource(const PointCloudSourceConstPtr& cloud, int stages = 2) {
    const int rbf = neighbor_search_method_ == NearestNeighborMethod::GPU_RBF_KERNEL ? 1 : 0;
    prepared_stages_ = stages;
    const detail::XyzView<PointSource> view(*cloud, scratch_xyz_);
    call(fvh_vgicp_prepare_source(core_, view.data, (int)cloud->size(), view.stride, k_correspondences_, (int)regularization_method_, rbf, stages), "prepare_source");
    prepared_cloud_ = cloud;
  }
  void adoptPreparedSource() {
    call(fvh_vgicp_adopt_prepared_source(core_), "adopt_prepared_source");
    input_ = prepared_cloud_;
    prepared_cloud_.reset();
    if (prepared_stages_ < 2) {
      if (neighbor_search_method_ == NearestNeighborMethod::GPU_RBF_KERNEL) call(fvh_vgicp_calculate_source_covariances_rbf(core_, (int)regularization_method_), "calculate_source_covariances_rbf");
      else call(fvh_vgicp_calculate_source_covariances(core_, (int)regularization_method_), "calculate_source_covariances");
    }
  }
  void alignAsync(const Matrix4f& guess = Matrix4f::Identity()) {
    double g16[16];
    Isometry3d::from(guess).to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    call(fvh_vgicp_align_async(core_, g16, &p), "align_async");
  }
  const Matrix4f& alignWait() {
    fvh_lm_result r;
    call(fvh_vgicp_align_wait(core_, &r), "align_wait");
    this->final_transformation_ = Isometry3d::from_colmajor16(r.T).cast_float();
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return this->final_transformation_;
  }

protected:
  double linearize(const Isometry3d& trans, Matrix6d* H, Vector6d* b) override {  // :170-173
    double T16[16], err = 0, Hc[36];
    trans.to_colmajor16(T16);
    call(fvh_vgicp_update_correspondences(core_, T16), "update_correspondences");
    call(fvh_vgicp_compute_error(core_, T16, (H && b) ? Hc : nullptr, (H && b) ? b->data() : nullptr, &err), "compute_error");
    if (H && b) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) (*H)[i * 6 + j] = Hc[j * 6 + i];
    return err;
  }
  double compute_error(const Isometry3d& trans) override {  // :176-178
    double T16[16], err = 0;
    trans.to_colmajor16(T16);
    call(fvh_vgicp_compute_error(core_, T16, nullptr, nullptr, &err), "compute_error");
    return err;
  }
  bool device_align(Isometry3d& x0) override {
    double g16[16];
    x0.to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    fvh_lm_result r;
    call(fvh_vgicp_set_lm_trace(core_, this->lm_debug_print_ ? 1 : 0), "set_lm_trace");
    call(fvh_vgicp_align(core_, g16, &p, &r), "align");
    if (this->lm_debug_print_) {
      int n = 0;
      call(fvh_vgicp_get_lm_trace(core_, &n, nullptr), "get_lm_trace");
      std::vector<double> rows(6 * (size_t)n);
      if (n) call(fvh_vgicp_get_lm_trace(core_, &n, rows.data()), "get_lm_trace");
      detail::print_lm_trace(rows);
    }
    x0 = Isometry3d::from_colmajor16(r.T);
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return true;
  }
  /// the enum value the switch statements act on: the default one is served by the device search unless the host tree was asked for
  /// (a cloud with fewer points than k keeps the host path: the reference pads those lists with index 0, :155,162)
  NearestNeighborMethod effective_neighbor_method(size_t n_points) const {
    if (neighbor_search_method_ != NearestNeighborMethod::CPU_PARALLEL_KDTREE || host_kdtree_ || n_points < (size_t)k_correspondences_) return neighbor_search_method_;
    return NearestNeighborMethod::GPU_BRUTEFORCE;
  }
  static bool host_kdtree_default() {
    static const bool on = [] { const char* v = std::getenv("FVH_HOST_KDTREE"); return v && std::atoi(v) != 0; }();
    return on;
  }
  /// find_neighbors_parallel_kdtree (:152-167): host kd-tree + OpenMP
  std::vector<int> find_neighbors_parallel_kdtree(int k, const std::vector<float>& xyz) const {
    const int n = (int)(xyz.size() / 3);
    host::KdTree tree(xyz.data(), n);
    std::vector<int> neighbors((size_t)n * k);
    const int threads = host::omp_threads_for(n);  // hundreds of threads on a 17k-point loop only add contention
#pragma omp parallel for schedule(guided, 8) num_threads(threads)
    for (int i = 0; i < n; i++) {
      int* row = &neighbors[(size_t)i * k];
      tree.knn(&xyz[3 * (size_t)i], k, row);
      // a cloud with fewer than k points: the reference's zero-initialised vector keeps index 0 in the unfilled entries (:155,162)
      for (int j = 0; j < k; j++) if (row[j] < 0) row[j] = 0;
    }
    return neighbors;
  }
  void call(int rc, const char* what) const { detail::check(rc, what, fvh_vgicp_last_error(core_)); }

private:
  int k_correspondences_ = 20;                                                                   // :24
  double voxel_resolution_ = 1.0;                                                                // :25
  RegularizationMethod regularization_method_ = RegularizationMethod::PLANE;                     // :26
  NearestNeighborMethod neighbor_search_method_ = NearestNeighborMethod::CPU_PARALLEL_KDTREE;    // :27
  bool host_kdtree_ = host_kdtree_default();                                                     // setHostKdTree
  int prepared_stages_ = 2;                                                                         // prepareNextSourceDevice
  PointCloudSourceConstPtr prepared_cloud_;                                                         // prepareNextSource (host cloud): becomes input_ on adoption
  fvh_vgicp* core_ = nullptr;
  std::vector<float> scratch_xyz_;  // only used for point types that are not 12 / 16 bytes of packed xyz
};

/// FastGICP (gicp/fast_gicp.hpp:24-98, impl/fast_gicp_impl.hpp) on the HIP engine: the reference class is CPU/OpenMP only;
/// here the covariances (exact k-NN + regularisation), the nearest-target-point correspondences and the cost sums run on
/// the device; the LM recursion runs on the device too (setUseDeviceLM(false): the reference's host loop, LsqRegistration::step_lm).
template <typename PointSource, typename PointTarget>
class FastGICP : public LsqRegistration<PointSource, PointTarget> {
  using Base = LsqRegistration<PointSource, PointTarget>;

protected:
  using Base::input_;
  using Base::target_;

public:
  using PointCloudSource = typename Base::PointCloudSource;
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;

  explicit FastGICP(int device = 0) {  // fast_gicp_impl.hpp:9-23
    detail::check(fvh_vgicp_create(device, &core_), "fvh_vgicp_create", "cannot create the HIP engine (no GPU? there is no CPU fallback)");
  }
  ~FastGICP() override { if (core_) fvh_vgicp_destroy(core_); }
  FastGICP(const FastGICP&) = delete;
  FastGICP& operator=(const FastGICP&) = delete;

  void setNumThreads(int) {}  // :29-38: OpenMP threads of the CPU class; the device needs none
  void setCorrespondenceRandomness(int k) { k_correspondences_ = k; }  // :41-43
  void setRegularizationMethod(RegularizationMethod method) { regularization_method_ = method; }  // :46-48
  void setMaxCorrespondenceDistance(double d) { call(fvh_vgicp_gicp_set_max_correspondence_distance(core_, d), "gicp_set_max_correspondence_distance"); }  // pcl::Registration

  void swapSourceAndTarget() override {  // :51-62
    call(fvh_vgicp_gicp_swap_source_and_target(core_), "gicp_swap_source_and_target");
    input_.swap(target_);
  }
  void clearSource() override { input_.reset(); }  // :65-68
  void clearTarget() override { target_.reset(); }  // :71-74

  void setInputSource(const PointCloudSourceConstPtr& cloud) override {  // :77-85 (covariances: :103-112, computed eagerly here)
    if (cloud == input_) return;
    input_ = cloud;
    const detail::XyzView<PointSource> view(*cloud, scratch_xyz_);
    call(fvh_vgicp_set_source_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_source_cloud");
    estimate_covariances(*cloud, true);
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {  // :88-95
    if (cloud == target_) return;
    target_ = cloud;
    const detail::XyzView<PointTarget> view(*cloud, scratch_xyz_);
    call(fvh_vgicp_set_target_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_target_cloud");
    estimate_covariances(*cloud, false);
  }
  /// gicp/fast_gicp.hpp:60-70: covariances computed elsewhere / read back. The reference carries them as 4x4 doubles with a
  /// zero last row and column; here a covariance is its 3x3 block, 9 doubles per point (symmetric, so any major).
  using Covariances = std::vector<std::array<double, 9>>;
  void setSourceCovariances(const Covariances& covs) {
    if (!input_ || covs.size() != input_->size()) throw std::invalid_argument("setSourceCovariances: one covariance per source point, after setInputSource");
    call(fvh_vgicp_set_source_covariances(core_, covs.empty() ? nullptr : covs[0].data()), "set_source_covariances");
  }
  virtual void setTargetCovariances(const Covariances& covs) {  // (virtual: FastVGICP's target voxel map is made of them)
    if (!target_ || covs.size() != target_->size()) throw std::invalid_argument("setTargetCovariances: one covariance per target point, after setInputTarget");
    call(fvh_vgicp_set_target_covariances(core_, covs.empty() ? nullptr : covs[0].data()), "set_target_covariances");
  }
  Covariances getSourceCovariances() const { return get_covariances(input_ ? input_->size() : 0, true); }
  Covariances getTargetCovariances() const { return get_covariances(target_ ? target_->size() : 0, false); }

  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) override {
    double T16[16], score = 0;
    Isometry3d::from(this->final_transformation_).to_colmajor16(T16);
    call(fvh_vgicp_fitness_score(core_, T16, max_range, &score), "fitness_score");
    return score;
  }
  fvh_vgicp* core() { return core_; }

protected:
  Covariances get_covariances(size_t n, bool source) const {
    std::vector<float> f(9 * n);
    if (n) call(source ? fvh_vgicp_get_source_covariances(core_, f.data()) : fvh_vgicp_get_target_covariances(core_, f.data()), "get_covariances");
    Covariances out(n);
    for (size_t i = 0; i < n; i++) for (int j = 0; j < 9; j++) out[i][j] = f[9 * i + j];
    return out;
  }
  /// the whole LM loop on the device (fvh_vgicp_gicp_align): one nearest-point search + one cost launch per LM transition, no host round trip
  bool device_align(Isometry3d& x0) override {
    double g16[16];
    x0.to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    fvh_lm_result r;
    call(fvh_vgicp_set_lm_trace(core_, this->lm_debug_print_ ? 1 : 0), "set_lm_trace");
    call(fvh_vgicp_gicp_align(core_, g16, &p, &r), "gicp_align");
    if (this->lm_debug_print_) {
      int n = 0;
      call(fvh_vgicp_get_lm_trace(core_, &n, nullptr), "get_lm_trace");
      std::vector<double> rows(6 * (size_t)n);
      if (n) call(fvh_vgicp_get_lm_trace(core_, &n, rows.data()), "get_lm_trace");
      detail::print_lm_trace(rows);
    }
    x0 = Isometry3d::from_colmajor16(r.T);
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return true;
  }
  double linearize(const Isometry3d& trans, Matrix6d* H, Vector6d* b) override {  // :159-213
    double T16[16], err = 0, Hc[36];
    trans.to_colmajor16(T16);
    call(fvh_vgicp_gicp_update_correspondences(core_, T16), "gicp_update_correspondences");
    call(fvh_vgicp_gicp_compute_error(core_, T16, (H && b) ? Hc : nullptr, (H && b) ? b->data() : nullptr, &err), "gicp_compute_error");
    if (H && b) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) (*H)[i * 6 + j] = Hc[j * 6 + i];
    return err;
  }
  double compute_error(const Isometry3d& trans) override {  // :216-240
    double T16[16], err = 0;
    trans.to_colmajor16(T16);
    call(fvh_vgicp_gicp_compute_error(core_, T16, nullptr, nullptr, &err), "gicp_compute_error");
    return err;
  }
  void call(int rc, const char* what) const { detail::check(rc, what, fvh_vgicp_last_error(core_)); }

  /// calculate_covariances (fast_gicp_impl.hpp:244-301): k nearest neighbours of every point + regularisation. The device's exact
  /// search serves k <= 64 on clouds of at least k points; what the reference's kd-tree also accepts -- larger k, or fewer points than
  /// k (nearestKSearch then returns them all; the unfilled entries of its index vector stay 0) -- goes through the host kd-tree.
  template <typename CloudT>
  void estimate_covariances(const CloudT& cloud, bool source) {
    const int n = (int)cloud.size(), k = k_correspondences_;
    if (k < 1) throw std::invalid_argument("setCorrespondenceRandomness: k must be >= 1");
    if (n == 0) return;
    if (k <= 64 && n >= k) {
      call(source ? fvh_vgicp_find_source_neighbors(core_, k) : fvh_vgicp_find_target_neighbors(core_, k), "find_neighbors");
    } else {
      if (k > 64) throw std::invalid_argument("setCorrespondenceRandomness: more than 64 neighbours per point are not supported by the covariance kernel");
      const std::vector<float> xyz = detail::pack_xyz(cloud);
      host::KdTree tree(xyz.data(), n);
      std::vector<int> nb((size_t)n * k);
      for (int i = 0; i < n; i++) {
        int* row = &nb[(size_t)i * k];
        tree.knn(&xyz[3 * (size_t)i], k, row);
        for (int j = 0; j < k; j++) if (row[j] < 0) row[j] = 0;
      }
      call(source ? fvh_vgicp_set_source_neighbors(core_, k, nb.data()) : fvh_vgicp_set_target_neighbors(core_, k, nb.data()), "set_neighbors");
    }
    call(source ? fvh_vgicp_calculate_source_covariances(core_, (int)regularization_method_) : fvh_vgicp_calculate_target_covariances(core_, (int)regularization_method_), "calculate_covariances");
  }

protected:
  int k_correspondences_ = 20;                                                // :17
  RegularizationMethod regularization_method_ = RegularizationMethod::PLANE;  // :21
  fvh_vgicp* core_ = nullptr;
  std::vector<float> scratch_xyz_;  // only used for point types that are not 12 / 16 bytes of packed xyz
};

/// FastVGICP -- the reference's CPU / OpenMP class (gicp/fast_vgicp.hpp:28-78, impl/fast_vgicp_impl.hpp) -- served by the same HIP
/// engine in its fp64 arithmetic. As in the reference it EXTENDS FastGICP (fast_vgicp.hpp:28): the cloud / covariance side --
/// `setCorrespondenceRandomness(k)` honoured (FastGICP::k_correspondences_, fast_gicp_impl.hpp:41-43,253-265; the CUDA class ignores
/// it, fast_vgicp_cuda_impl.hpp:38), covariances from EXACT k nearest neighbours (the reference's kd-tree; here the device's exact
/// search: identical lists), `setNumThreads` (a no-op: there is no OpenMP team to size), the covariance accessors -- is FastGICP's;
/// what it replaces is the correspondence model: voxels of the target instead of its nearest points (fast_vgicp_impl.hpp:73-204).
/// The neighbour search takes no radius (DIRECT1 / 7 / 27 only, fast_vgicp_voxel.hpp:16-43).
template <typename PointSource, typename PointTarget>
class FastVGICP : public FastGICP<PointSource, PointTarget> {
  using Base = FastGICP<PointSource, PointTarget>;
  using Base::core_;
  using Base::input_;
  using Base::target_;

public:
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;

  explicit FastVGICP(int device = 0) : Base(device) {  // fast_vgicp_impl.hpp:19-25
    this->call(fvh_vgicp_set_resolution(core_, 1.0), "set_resolution");
    this->call(fvh_vgicp_set_neighbor_search_method(core_, (int)NeighborSearchMethod::DIRECT1, -1.0), "set_neighbor_search_method");
  }
  // The reference builds the voxel map lazily, inside the first linearisation of every align (fast_vgicp_impl.hpp:120-123), from
  // whatever resolution and target covariances are current then; this class builds it eagerly -- so every setter the map depends on
  // rebuilds it when a target is set (setResolution, setTargetCovariances, setVoxelAccumulationMode).
  void setResolution(double resolution) {  // :36-38
    this->call(fvh_vgicp_set_resolution(core_, resolution), "set_resolution");
    if (target_) this->call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");
  }
  void setTargetCovariances(const typename Base::Covariances& covs) override {  // gicp/fast_gicp.hpp:66-68, consumed by :129-156 of the impl
    Base::setTargetCovariances(covs);
    this->call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");
  }
  void setNeighborSearchMethod(NeighborSearchMethod method) {  // :46-48
    if (method == NeighborSearchMethod::DIRECT_RADIUS) detail::check(FVH_ERR_INVALID_ARGUMENT, "setNeighborSearchMethod", "FastVGICP has no DIRECT_RADIUS (fast_vgicp_voxel.hpp:16-43): use FastVGICPCuda");
    this->call(fvh_vgicp_set_neighbor_search_method(core_, (int)method, -1.0), "set_neighbor_search_method");
  }
  void setVoxelAccumulationMode(VoxelAccumulationMode mode) {  // :41-43; takes effect when the target voxel map is next built
    this->call(fvh_vgicp_set_voxel_accumulation_mode(core_, static_cast<int>(mode)), "set_voxel_accumulation_mode");
    if (target_) this->call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");
  }
  void setMaxCorrespondenceDistance(double) {}  // inherited from pcl::Registration; FastVGICP never reads it (voxel correspondences)

  void swapSourceAndTarget() override {  // fast_vgicp_impl.hpp:46-53: swaps clouds and covariances, drops the voxel map (rebuilt here, not lazily)
    this->call(fvh_vgicp_swap_source_and_target(core_), "swap_source_and_target");
    input_.swap(target_);
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {  // :56-63
    if (cloud == target_) return;
    Base::setInputTarget(cloud);
    this->call(fvh_vgicp_create_target_voxelmap(core_), "create_target_voxelmap");
  }

protected:
  double linearize(const Isometry3d& trans, Matrix6d* H, Vector6d* b) override {  // :119-178
    double T16[16], err = 0, Hc[36];
    trans.to_colmajor16(T16);
    this->call(fvh_vgicp_update_correspondences(core_, T16), "update_correspondences");
    this->call(fvh_vgicp_compute_error(core_, T16, (H && b) ? Hc : nullptr, (H && b) ? b->data() : nullptr, &err), "compute_error");
    if (H && b) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) (*H)[i * 6 + j] = Hc[j * 6 + i];
    return err;
  }
  double compute_error(const Isometry3d& trans) override {  // :181-204
    double T16[16], err = 0;
    trans.to_colmajor16(T16);
    this->call(fvh_vgicp_compute_error(core_, T16, nullptr, nullptr, &err), "compute_error");
    return err;
  }
  bool device_align(Isometry3d& x0) override {
    double g16[16];
    x0.to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    fvh_lm_result r;
    this->call(fvh_vgicp_set_lm_trace(core_, this->lm_debug_print_ ? 1 : 0), "set_lm_trace");
    this->call(fvh_vgicp_align(core_, g16, &p, &r), "align");
    if (this->lm_debug_print_) {
      int n = 0;
      this->call(fvh_vgicp_get_lm_trace(core_, &n, nullptr), "get_lm_trace");
      std::vector<double> rows(6 * (size_t)n);
      if (n) this->call(fvh_vgicp_get_lm_trace(core_, &n, rows.data()), "get_lm_trace");
      detail::print_lm_trace(rows);
    }
    x0 = Isometry3d::from_colmajor16(r.T);
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return true;
  }
};

/// NDTCuda (ndt_cuda.hpp:23-69, impl/ndt_cuda_impl.hpp)
template <typename PointSource, typename PointTarget>
class NDTCuda : public LsqRegistration<PointSource, PointTarget> {
  using Base = LsqRegistration<PointSource, PointTarget>;
  using Base::input_;
  using Base::target_;

public:
  using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
  using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;

  explicit NDTCuda(int device = 0) { detail::check(fvh_ndt_create(device, &core_), "fvh_ndt_create", "cannot create the HIP engine (no GPU? there is no CPU fallback)"); }
  ~NDTCuda() override { if (core_) fvh_ndt_destroy(core_); }
  NDTCuda(const NDTCuda&) = delete;
  NDTCuda& operator=(const NDTCuda&) = delete;

  void setDistanceMode(NDTDistanceMode mode) { call(fvh_ndt_set_distance_mode(core_, (int)mode), "set_distance_mode"); }
  void setResolution(double resolution) { call(fvh_ndt_set_resolution(core_, resolution), "set_resolution"); }
  void setNeighborSearchMethod(NeighborSearchMethod method, double radius = -1.0) { call(fvh_ndt_set_neighbor_search_method(core_, (int)method, radius), "set_neighbor_search_method"); }

  void swapSourceAndTarget() override { call(fvh_ndt_swap_source_and_target(core_), "swap_source_and_target"); input_.swap(target_); }
  void clearSource() override { input_.reset(); }
  void clearTarget() override { target_.reset(); }
  void setInputSource(const PointCloudSourceConstPtr& cloud) override {  // ndt_cuda_impl.hpp:52-60
    if (cloud == input_) return;
    input_ = cloud;
    const detail::XyzView<PointSource> view(*cloud, scratch_xyz_);
    call(fvh_ndt_set_source_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_source_cloud");
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) override {  // :63-73
    if (cloud == target_) return;
    target_ = cloud;
    const detail::XyzView<PointTarget> view(*cloud, scratch_xyz_);
    call(fvh_ndt_set_target_cloud_strided(core_, view.data, (int)cloud->size(), view.stride), "set_target_cloud");
  }
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) override {
    double T16[16], score = 0;
    Isometry3d::from(this->final_transformation_).to_colmajor16(T16);
    call(fvh_ndt_fitness_score(core_, T16, max_range, &score), "fitness_score");
    return score;
  }
  fvh_ndt* core() { return core_; }

  // ---- frame streams as a two-stage pipeline (no reference counterpart; C ABI: fvh_ndt_align_async / _wait, fvh_ndt_prepare_source_device /
  // _adopt_prepared_source). kitti.cpp:95-128 with the preparation of frame k+1 hidden under the registration of frame k:
  //     adoptPreparedSource(); alignAsync(); [filter frame k+1 on the prepare stream; prepareNextSourceDevice()]; alignWait(); swapSourceAndTarget();
  // The prepared source lives on the device only: getInputSource() is null for it and alignWait() returns the pose without
  // transforming a host cloud. Same kernels on the same data as the sequential calls: same registration.
  void prepareNextSourceDevice(const float* d_xyz, int n, int stride_floats = 3) { prepared_cloud_.reset(); call(fvh_ndt_prepare_source_device(core_, d_xyz, n, stride_floats), "prepare_source_device"); }
  /// the output of the filter's last ApproximateVoxelGrid call becomes the prepared source without a copy (fvh_ndt_prepare_source_from_voxelgrid)
  void prepareNextSourceFromFilter(fvh_voxelgrid* filter) { prepared_cloud_.reset(); call(fvh_ndt_prepare_source_from_voxelgrid(core_, filter), "prepare_source_from_voxelgrid"); }
  /// the same for a host cloud (consumed before the call returns); adoptPreparedSource() then makes it getInputSource()
  void prepareNextSource(const PointCloudSourceConstPtr& cloud) {
    const detail::XyzView<PointSource> view(*cloud, scratch_xyz_);
    call(fvh_ndt_prepare_source(core_, view.data, (int)cloud->size(), view.stride), "prepare_source");
    prepared_cloud_ = cloud;
  }
  void adoptPreparedSource() { call(fvh_ndt_adopt_prepared_source(core_), "adopt_prepared_source"); input_ = prepared_cloud_; prepared_cloud_.reset(); }
  void alignAsync(const Matrix4f& guess = Matrix4f::Identity()) {
    double g16[16];
    Isometry3d::from(guess).to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    call(fvh_ndt_align_async(core_, g16, &p), "align_async");
  }
  const Matrix4f& alignWait() {
    fvh_lm_result r;
    call(fvh_ndt_align_wait(core_, &r), "align_wait");
    this->final_transformation_ = Isometry3d::from_colmajor16(r.T).cast_float();
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return this->final_transformation_;
  }

protected:
  void computeTransformation(typename Base::PointCloudSource& output, const Matrix4f& guess) override {  // :76-79
    call(fvh_ndt_create_voxelmaps(core_), "create_voxelmaps");
    Base::computeTransformation(output, guess);
  }
  double linearize(const Isometry3d& trans, Matrix6d* H, Vector6d* b) override {
    double T16[16], err = 0, Hc[36];
    trans.to_colmajor16(T16);
    call(fvh_ndt_update_correspondences(core_, T16), "update_correspondences");
    call(fvh_ndt_compute_error(core_, T16, (H && b) ? Hc : nullptr, (H && b) ? b->data() : nullptr, &err), "compute_error");
    if (H && b) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) (*H)[i * 6 + j] = Hc[j * 6 + i];
    return err;
  }
  double compute_error(const Isometry3d& trans) override {
    double T16[16], err = 0;
    trans.to_colmajor16(T16);
    call(fvh_ndt_compute_error(core_, T16, nullptr, nullptr, &err), "compute_error");
    return err;
  }
  bool device_align(Isometry3d& x0) override {
    double g16[16];
    x0.to_colmajor16(g16);
    fvh_lm_params p{this->max_iterations_, this->rotation_epsilon_, this->transformation_epsilon_, this->lm_max_iterations_, this->lm_init_lambda_factor_,
                    this->lsq_optimizer_type_ == LSQ_OPTIMIZER_TYPE::GaussNewton ? 1 : 0};
    fvh_lm_result r;
    call(fvh_ndt_set_lm_trace(core_, this->lm_debug_print_ ? 1 : 0), "set_lm_trace");
    call(fvh_ndt_align(core_, g16, &p, &r), "align");
    if (this->lm_debug_print_) {
      int n = 0;
      call(fvh_ndt_get_lm_trace(core_, &n, nullptr), "get_lm_trace");
      std::vector<double> rows(6 * (size_t)n);
      if (n) call(fvh_ndt_get_lm_trace(core_, &n, rows.data()), "get_lm_trace");
      detail::print_lm_trace(rows);
    }
    x0 = Isometry3d::from_colmajor16(r.T);
    this->converged_ = r.converged != 0;
    this->nr_iterations_ = r.nr_iterations;
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) this->final_hessian_[i * 6 + j] = r.H[j * 6 + i];
    if (r.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    return true;
  }
  void call(int rc, const char* what) const { detail::check(rc, what, fvh_ndt_last_error(core_)); }

private:
  fvh_ndt* core_ = nullptr;
  std::vector<float> scratch_xyz_;
  PointCloudSourceConstPtr prepared_cloud_;  // prepareNextSource (host cloud): becomes input_ on adoption
};

}  // namespace fast_gicp
