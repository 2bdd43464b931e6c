// ORACLE (test infrastructure, NOT product code).
// Small fixed-size fp64 linear algebra used by the CPU restatement of the
// reference's VGICP/NDT hot path.  The reference uses Eigen (not available
// here, thirdparty/Eigen is an empty submodule); these helpers restate the
// handful of 3x3 / 6x6 operations it calls.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace orc {

struct V3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

struct M3 {  // row-major 3x3
  double m[9];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
};

inline M3 m3_zero() { M3 a; std::memset(a.m, 0, sizeof(a.m)); return a; }
inline M3 m3_identity() { M3 a = m3_zero(); a(0, 0) = a(1, 1) = a(2, 2) = 1.0; return a; }
inline M3 m3_add(const M3& a, const M3& b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i]; return c; }
inline M3 m3_scale(const M3& a, double s) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] * s; return c; }
inline M3 m3_mul(const M3& a, const M3& b) {
  M3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a(i, k) * b(k, j);
      c(i, j) = s;
    }
  return c;
}
inline M3 m3_transpose(const M3& a) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c(i, j) = a(j, i); return c; }
inline V3 m3_mulv(const M3& a, const V3& x) {
  V3 y;
  for (int i = 0; i < 3; i++) y[i] = a(i, 0) * x[0] + a(i, 1) * x[1] + a(i, 2) * x[2];
  return y;
}
inline double m3_det(const M3& a) {
  return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
}
// Eigen's fixed-size 3x3 .inverse() is the cofactor (adjugate / det) formula.
inline M3 m3_inverse(const M3& a) {
  M3 c;
  c(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  c(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  c(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  c(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  c(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  c(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  c(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  c(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  c(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  double det = a(0, 0) * c(0, 0) + a(0, 1) * c(1, 0) + a(0, 2) * c(2, 0);
  double inv = 1.0 / det;
  for (int i = 0; i < 9; i++) c.m[i] *= inv;
  return c;
}
inline double m3_frobenius(const M3& a) { double s = 0; for (int i = 0; i < 9; i++) s += a.m[i] * a.m[i]; return std::sqrt(s); }

// skewd(), reference include/fast_gicp/so3/so3.hpp:21-31
inline M3 skew(const V3& x) {
  M3 s = m3_zero();
  s(0, 1) = -x[2]; s(0, 2) = x[1];
  s(1, 0) = x[2];  s(1, 2) = -x[0];
  s(2, 0) = -x[1]; s(2, 1) = x[0];
  return s;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (fp64).
// Stands in for Eigen::JacobiSVD (fast_gicp_impl.hpp:276) and
// SelfAdjointEigenSolver::computeDirect (covariance_regularization.cu:20,58,86):
// for a symmetric PSD matrix the SVD is U = V = eigenvectors, singular values =
// eigenvalues.  Output: eigenvalues ASCENDING in w, eigenvectors in columns of V.
inline void sym_eig3(const M3& A_in, double w[3], M3& V) {
  M3 A = A_in;
  V = m3_identity();
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
    double diag = A(0, 0) * A(0, 0) + A(1, 1) * A(1, 1) + A(2, 2) * A(2, 2);
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A(p, q);
        if (apq == 0.0) continue;
        double theta = (A(q, q) - A(p, p)) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        // A <- J^T A J
        for (int k = 0; k < 3; k++) {
          double akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq;
          A(k, q) = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk;
          A(q, k) = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V(k, p), vkq = V(k, q);
          V(k, p) = c * vkp - s * vkq;
          V(k, q) = s * vkp + c * vkq;
        }
      }
  }
  double ev[3] = {A(0, 0), A(1, 1), A(2, 2)};
  int idx[3] = {0, 1, 2};
  std::sort(idx, idx + 3, [&](int a, int b) { return ev[a] < ev[b]; });
  M3 Vs;
  for (int j = 0; j < 3; j++) {
    w[j] = ev[idx[j]];
    for (int i = 0; i < 3; i++) Vs(i, j) = V(i, idx[j]);
  }
  V = Vs;
}

// V * diag(d) * V^T
inline M3 m3_vdvt(const M3& V, const double d[3]) {
  M3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c(i, j) = V(i, 0) * d[0] * V(j, 0) + V(i, 1) * d[1] * V(j, 1) + V(i, 2) * d[2] * V(j, 2);
  return c;
}

// 4x4 rigid transform, row-major (Eigen::Isometry3d restated)
struct Iso3 {
  double R[9];
  double t[3];
};
inline Iso3 iso_identity() { Iso3 T; std::memset(&T, 0, sizeof(T)); T.R[0] = T.R[4] = T.R[8] = 1; return T; }
inline Iso3 iso_from_rowmajor16(const double* m) {
  Iso3 T;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.R[i * 3 + j] = m[i * 4 + j]; T.t[i] = m[i * 4 + 3]; }
  return T;
}
inline void iso_to_rowmajor16(const Iso3& T, double* m) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m[i * 4 + j] = T.R[i * 3 + j]; m[i * 4 + 3] = T.t[i]; }
  m[12] = m[13] = m[14] = 0; m[15] = 1;
}
inline Iso3 iso_mul(const Iso3& A, const Iso3& B) {  // A * B
  Iso3 C;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) C.R[i * 3 + j] = A.R[i * 3 + 0] * B.R[0 * 3 + j] + A.R[i * 3 + 1] * B.R[1 * 3 + j] + A.R[i * 3 + 2] * B.R[2 * 3 + j];
    C.t[i] = A.R[i * 3 + 0] * B.t[0] + A.R[i * 3 + 1] * B.t[1] + A.R[i * 3 + 2] * B.t[2] + A.t[i];
  }
  return C;
}
inline V3 iso_apply(const Iso3& T, const V3& p) {
  V3 q;
  for (int i = 0; i < 3; i++) q[i] = T.R[i * 3 + 0] * p[0] + T.R[i * 3 + 1] * p[1] + T.R[i * 3 + 2] * p[2] + T.t[i];
  return q;
}
inline M3 iso_rot(const Iso3& T) { M3 R; std::memcpy(R.m, T.R, sizeof(R.m)); return R; }

// Sensitivity probe (tests only): ORC_LM_ARITH_VARIANT bit 0 = one reciprocal per LDLT pivot instead of a division per
// entry, bit 1 = full-angle terms of se3_exp from the half-angle sincos (double-angle identities). Both variants are
// rounding-level rearrangements of the same formulas; tests use them to measure how far a 1-ulp change in the LM step
// moves the converged pose (it is NOT the reference arithmetic, which is the default 0).
inline int lm_arith_variant() {
  static const int v = [] { const char* e = std::getenv("ORC_LM_ARITH_VARIANT"); return e ? std::atoi(e) : 0; }();
  return v;
}

// se3_exp / so3_exp, reference include/fast_gicp/so3/so3.hpp:58-104 (rotation first).
inline Iso3 se3_exp(const double a[6]) {
  const double ox = a[0], oy = a[1], oz = a[2];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  // so3_exp -> quaternion (w, x, y, z)
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double th = std::sqrt(theta_sq);
    double half = 0.5 * th;
    imag_factor = std::sin(half) / th;
    real_factor = std::cos(half);
  }
  const double qw = real_factor, qx = imag_factor * ox, qy = imag_factor * oy, qz = imag_factor * oz;
  // Eigen::Quaterniond::toRotationMatrix() (no normalisation)
  M3 R;
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
  }
  const double theta = std::sqrt(theta_sq);
  V3 om = {{ox, oy, oz}};
  M3 Om = skew(om);
  M3 Om2 = m3_mul(Om, Om);
  M3 V;
  if (theta < 1e-10) {
    V = R;  // so3.matrix()
  } else {
    const double tsq = theta * theta;
    if (lm_arith_variant() & 2) {
      const double sh = std::sin(0.5 * theta), ch = std::cos(0.5 * theta), inv = 1.0 / theta_sq;
      V = m3_add(m3_identity(), m3_add(m3_scale(Om, 2.0 * sh * sh * inv), m3_scale(Om2, (theta - 2.0 * sh * ch) * inv / theta)));
    } else
    V = m3_add(m3_identity(), m3_add(m3_scale(Om, (1.0 - std::cos(theta)) / tsq), m3_scale(Om2, (theta - std::sin(theta)) / (tsq * theta))));
  }
  Iso3 T;
  std::memcpy(T.R, R.m, sizeof(T.R));
  V3 tv = {{a[3], a[4], a[5]}};
  V3 tt = m3_mulv(V, tv);
  T.t[0] = tt[0]; T.t[1] = tt[1]; T.t[2] = tt[2];
  return T;
}

// Solve (A) x = rhs for symmetric 6x6 A via LDL^T (no pivoting). Stands in for
// Eigen::LDLT<Matrix6d>::solve (lsq_registration_impl.hpp:111,134). Zero pivots as Eigen treats them: the column is not
// scaled (ldlt_inplace: "if (rs > 0 && pivot_is_valid) A21 /= realAkk") and the solve applies the pseudo-inverse of D
// (|D_ii| <= numeric_limits::min() -> 0), so H = 0 (no correspondences) yields d = 0, delta = I: guess returned, converged.
inline void ldlt6_solve(const double A_in[36], const double rhs[6], double x[6]) {
  double L[36] = {0}, D[6];
  for (int j = 0; j < 6; j++) {
    double d = A_in[j * 6 + j];
    for (int k = 0; k < j; k++) d -= L[j * 6 + k] * L[j * 6 + k] * D[k];
    D[j] = d;
    L[j * 6 + j] = 1.0;
    for (int i = j + 1; i < 6; i++) {
      double s = A_in[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k] * D[k];
      L[i * 6 + j] = !(std::fabs(d) > 2.2250738585072014e-308) ? s : ((lm_arith_variant() & 1) ? s * (1.0 / d) : s / d);
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] = !(std::fabs(D[i]) > 2.2250738585072014e-308) ? 0.0 : ((lm_arith_variant() & 1) ? y[i] * (1.0 / D[i]) : y[i] / D[i]);
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s; }
}

}  // namespace orc
