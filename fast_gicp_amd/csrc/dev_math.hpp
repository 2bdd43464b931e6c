// Register-resident small-matrix math for the gfx950 kernels. Everything is 3x3 / 6x6 and stays
// in VGPRs (no MFMA, no LDS): symmetric matrices are 6 scalars (xx xy xz yy yz zz).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Contraction only inside one source expression (not across statements, which is hipcc's default "fast"): whether a
// multiply and an add fuse must not depend on what the optimiser happens to see around them -- the persistent and the
// per-transition instantiations of the cost kernel have to produce bit-identical sums.
#pragma clang fp contract(on)

namespace fvh {

template <typename T>
struct Sym3 {  // symmetric 3x3
  T xx, xy, xz, yy, yz, zz;
};

template <typename T>
struct Vec3 {
  T x, y, z;
};

// rigid pose: row-major rotation + translation
template <typename T>
struct Pose {
  T r[9];
  T t[3];
};

struct PoseD {  // kernel-argument form (always double; cast inside)
  double r[9];
  double t[3];
};

template <typename T>
__device__ __forceinline__ Pose<T> pose_cast(const PoseD& p) {
  Pose<T> q;
#pragma unroll
  for (int i = 0; i < 9; i++) q.r[i] = (T)p.r[i];
#pragma unroll
  for (int i = 0; i < 3; i++) q.t[i] = (T)p.t[i];
  return q;
}

template <typename T>
__device__ __forceinline__ Vec3<T> transform(const Pose<T>& P, const Vec3<T>& a) {
  Vec3<T> q;
  q.x = P.r[0] * a.x + P.r[1] * a.y + P.r[2] * a.z + P.t[0];
  q.y = P.r[3] * a.x + P.r[4] * a.y + P.r[5] * a.z + P.t[1];
  q.z = P.r[6] * a.x + P.r[7] * a.y + P.r[8] * a.z + P.t[2];
  return q;
}

// R C R^T for symmetric C
template <typename T>
__device__ __forceinline__ Sym3<T> rotate_cov(const T* R, const Sym3<T>& C) {
  // M = R C
  T m00 = R[0] * C.xx + R[1] * C.xy + R[2] * C.xz, m01 = R[0] * C.xy + R[1] * C.yy + R[2] * C.yz, m02 = R[0] * C.xz + R[1] * C.yz + R[2] * C.zz;
  T m10 = R[3] * C.xx + R[4] * C.xy + R[5] * C.xz, m11 = R[3] * C.xy + R[4] * C.yy + R[5] * C.yz, m12 = R[3] * C.xz + R[4] * C.yz + R[5] * C.zz;
  T m20 = R[6] * C.xx + R[7] * C.xy + R[8] * C.xz, m21 = R[6] * C.xy + R[7] * C.yy + R[8] * C.yz, m22 = R[6] * C.xz + R[7] * C.yz + R[8] * C.zz;
  Sym3<T> S;
  S.xx = m00 * R[0] + m01 * R[1] + m02 * R[2];
  S.xy = m00 * R[3] + m01 * R[4] + m02 * R[5];
  S.xz = m00 * R[6] + m01 * R[7] + m02 * R[8];
  S.yy = m10 * R[3] + m11 * R[4] + m12 * R[5];
  S.yz = m10 * R[6] + m11 * R[7] + m12 * R[8];
  S.zz = m20 * R[6] + m21 * R[7] + m22 * R[8];
  return S;
}

// adjugate (transposed cofactors) and determinant of a symmetric 3x3: A^-1 = adj(A) / det
template <typename T>
__device__ __forceinline__ Sym3<T> adjugate(const Sym3<T>& A, T& det) {
  Sym3<T> C;
  C.xx = A.yy * A.zz - A.yz * A.yz;
  C.xy = A.xz * A.yz - A.xy * A.zz;
  C.xz = A.xy * A.yz - A.xz * A.yy;
  C.yy = A.xx * A.zz - A.xz * A.xz;
  C.yz = A.xy * A.xz - A.xx * A.yz;
  C.zz = A.xx * A.yy - A.xy * A.xy;
  det = A.xx * C.xx + A.xy * C.xy + A.xz * C.xz;
  return C;
}

// w / d without the range handling of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup: 12 instructions): hardware
// reciprocal seed + two Newton steps, ~1 ulp. For determinants of covariance sums -- normal, far from the exponent limits; a
// zero determinant gives NaN instead of inf, and both poison the sums the same way.
__device__ __forceinline__ double fast_div(double w, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return w * r;
}
__device__ __forceinline__ float fast_div(float w, float d) { return w / d; }
// a / b, correctly rounded, from the correctly rounded y = 1 / b (Markstein: q0 = RN(a y), r = a - b q0 exact in the fma,
// RN(q0 + r y) is the rounded quotient) -- three instructions per quotient for a divisor shared by the whole launch
__device__ __forceinline__ double div_by(double a, double b, double y) {
  const double q0 = a * y;
  return fma(fma(-q0, b, a), y, q0);
}
__device__ __forceinline__ float div_by(float a, float b, float) { return a / b; }

// inverse of a symmetric 3x3 by cofactors (what Eigen's fixed-size inverse does)
template <typename T>
__device__ __forceinline__ Sym3<T> inverse(const Sym3<T>& A) {
  Sym3<T> C;
  C.xx = A.yy * A.zz - A.yz * A.yz;
  C.xy = A.xz * A.yz - A.xy * A.zz;
  C.xz = A.xy * A.yz - A.xz * A.yy;
  C.yy = A.xx * A.zz - A.xz * A.xz;
  C.yz = A.xy * A.xz - A.xx * A.yz;
  C.zz = A.xx * A.yy - A.xy * A.xy;
  T det = A.xx * C.xx + A.xy * C.xy + A.xz * C.xz;
  T inv = (T)1 / det;
  C.xx *= inv; C.xy *= inv; C.xz *= inv; C.yy *= inv; C.yz *= inv; C.zz *= inv;
  return C;
}

template <typename T>
__device__ __forceinline__ Vec3<T> mul(const Sym3<T>& M, const Vec3<T>& v) {
  Vec3<T> r;
  r.x = M.xx * v.x + M.xy * v.y + M.xz * v.z;
  r.y = M.xy * v.x + M.yy * v.y + M.yz * v.z;
  r.z = M.xz * v.x + M.yz * v.y + M.zz * v.z;
  return r;
}

template <typename T>
__device__ __forceinline__ Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
  Vec3<T> c;
  c.x = a.y * b.z - a.z * b.y;
  c.y = a.z * b.x - a.x * b.z;
  c.z = a.x * b.y - a.y * b.x;
  return c;
}

// 1 / d, sqrt(x), 1 / sqrt(x) for NORMAL, finite arguments far from the exponent limits: hardware seed + two Newton steps (~1 ulp), without
// the scaling / fix-up sequences of the IEEE operations (12-20 instructions each). The Jacobi rotations below are a chain of ~2,000
// DEPENDENT fp64 instructions per matrix (each rotation: three divisions, two square roots); that chain IS the covariance kernel of a
// 17k-point cloud (9.4 us) and the NDT voxel finalisation (10.7 us) -- these cut it by ~40 %.
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
__device__ __forceinline__ double rsq_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double sqrt_nr(double x) {
  const double y = rsq_nr(x);
  const double s = x * y;
  return fma(fma(-s, s, x), 0.5 * y, s);
}

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi, fp64. Eigenvalues ascending in w,
// eigenvectors in the COLUMNS of V (row-major V[r*3+c]).
// A rotation is skipped for |apq| <= 1e-290 (physically zero; keeps the reciprocal seed finite); |theta| > 1e100 -- where
// sqrt(theta^2 + 1) == |theta| exactly and theta^2 may overflow -- takes t = 1 / (2 theta) through an IEEE division (rare branch).
__device__ inline void sym_eig3(const Sym3<double>& S, double w[3], double V[9]) {
  double a00 = S.xx, a01 = S.xy, a02 = S.xz, a11 = S.yy, a12 = S.yz, a22 = S.zz;
  double v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#define FVH_JACOBI_ROT(app, aqq, apq, arp, arq, p, q)                                   \
  if (fabs(apq) > 1e-290) {                                                              \
    const double theta = (aqq - app) * rcp_nr(2.0 * apq);                                \
    double t;                                                                            \
    if (!(fabs(theta) <= 1e100)) t = 0.5 / theta;                                        \
    else { const double rr = rcp_nr(fabs(theta) + sqrt_nr(fma(theta, theta, 1.0))); t = theta >= 0 ? rr : -rr; } \
    const double c = rsq_nr(fma(t, t, 1.0)), s = t * c;                                  \
    app -= t * apq;                                                                      \
    aqq += t * apq;                                                                      \
    apq = 0.0;                                                                           \
    double rp = arp, rq = arq;                                                           \
    arp = c * rp - s * rq;                                                               \
    arq = s * rp + c * rq;                                                               \
    _Pragma("unroll") for (int k = 0; k < 3; k++) {                                      \
      double vp = v[k * 3 + p], vq = v[k * 3 + q];                                       \
      v[k * 3 + p] = c * vp - s * vq;                                                    \
      v[k * 3 + q] = s * vp + c * vq;                                                    \
    }                                                                                    \
  }
  for (int sweep = 0; sweep < 12; sweep++) {
    double off = a01 * a01 + a02 * a02 + a12 * a12;
    double diag = a00 * a00 + a11 * a11 + a22 * a22;
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    FVH_JACOBI_ROT(a00, a11, a01, a02, a12, 0, 1)  // (p,q)=(0,1), r=2
    FVH_JACOBI_ROT(a00, a22, a02, a01, a12, 0, 2)  // (0,2), r=1
    FVH_JACOBI_ROT(a11, a22, a12, a01, a02, 1, 2)  // (1,2), r=0
  }
#undef FVH_JACOBI_ROT
  // sort ascending (3-element network), permuting columns of v
  double e0 = a00, e1 = a11, e2 = a22;
  int i0 = 0, i1 = 1, i2 = 2;
  if (e0 > e1) { double t = e0; e0 = e1; e1 = t; int ti = i0; i0 = i1; i1 = ti; }
  if (e1 > e2) { double t = e1; e1 = e2; e2 = t; int ti = i1; i1 = i2; i2 = ti; }
  if (e0 > e1) { double t = e0; e0 = e1; e1 = t; int ti = i0; i0 = i1; i1 = ti; }
  w[0] = e0; w[1] = e1; w[2] = e2;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double c0 = v[k * 3 + 0], c1 = v[k * 3 + 1], c2 = v[k * 3 + 2];
    V[k * 3 + 0] = i0 == 0 ? c0 : (i0 == 1 ? c1 : c2);
    V[k * 3 + 1] = i1 == 0 ? c0 : (i1 == 1 ? c1 : c2);
    V[k * 3 + 2] = i2 == 0 ? c0 : (i2 == 1 ? c1 : c2);
  }
}

// V diag(d) V^T
__device__ __forceinline__ Sym3<double> vdvt(const double V[9], const double d[3]) {
  Sym3<double> C;
  C.xx = V[0] * d[0] * V[0] + V[1] * d[1] * V[1] + V[2] * d[2] * V[2];
  C.xy = V[0] * d[0] * V[3] + V[1] * d[1] * V[4] + V[2] * d[2] * V[5];
  C.xz = V[0] * d[0] * V[6] + V[1] * d[1] * V[7] + V[2] * d[2] * V[8];
  C.yy = V[3] * d[0] * V[3] + V[4] * d[1] * V[4] + V[5] * d[2] * V[5];
  C.yz = V[3] * d[0] * V[6] + V[4] * d[1] * V[7] + V[5] * d[2] * V[8];
  C.zz = V[6] * d[0] * V[6] + V[7] * d[1] * V[7] + V[8] * d[2] * V[8];
  return C;
}

// RegularizationMethod (ordinals of gicp_settings.hpp:7): reference CPU fast_gicp_impl.hpp:267-297,
// GPU covariance_regularization.cu:34-124. NONE and NORMALIZED_MIN_EIG are unimplemented on the
// reference GPU; implemented here with the CPU semantics.
__device__ inline Sym3<double> regularize_cov(const Sym3<double>& C, int method) {
  if (method == 0) return C;
  if (method == 4) {  // FROBENIUS
    Sym3<double> A = C;
    A.xx += 1e-3; A.yy += 1e-3; A.zz += 1e-3;
    Sym3<double> Ai = inverse(A);
    double nrm = sqrt(Ai.xx * Ai.xx + Ai.yy * Ai.yy + Ai.zz * Ai.zz + 2.0 * (Ai.xy * Ai.xy + Ai.xz * Ai.xz + Ai.yz * Ai.yz));
    double s = 1.0 / nrm;
    Ai.xx *= s; Ai.xy *= s; Ai.xz *= s; Ai.yy *= s; Ai.yz *= s; Ai.zz *= s;
    return inverse(Ai);
  }
  double w[3], V[9], d[3];
  sym_eig3(C, w, V);
  if (method == 3) { d[0] = 1e-3; d[1] = 1.0; d[2] = 1.0; }                                            // PLANE
  else if (method == 1) { d[0] = fmax(w[0], 1e-3); d[1] = fmax(w[1], 1e-3); d[2] = fmax(w[2], 1e-3); }  // MIN_EIG
  else { d[0] = fmax(w[0] / w[2], 1e-3); d[1] = fmax(w[1] / w[2], 1e-3); d[2] = fmax(w[2] / w[2], 1e-3); }  // NORMALIZED_MIN_EIG
  return vdvt(V, d);
}

// ---- voxel keys -------------------------------------------------------------------------------
// coord = floor(p / res - 0.5)  (fast_vgicp_voxel.hpp:158-160 / vector3_hash.cuh:35-38), packed 21
// bits per axis into one 64-bit key so bucket claims are a single 64-bit CAS.
#define FVH_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define FVH_COORD_BIAS (1 << 20)

__device__ __forceinline__ unsigned long long pack_key(int x, int y, int z) {
  return (unsigned long long)(unsigned)(x + FVH_COORD_BIAS) | ((unsigned long long)(unsigned)(y + FVH_COORD_BIAS) << 21) |
         ((unsigned long long)(unsigned)(z + FVH_COORD_BIAS) << 42);
}
__host__ __device__ __forceinline__ void unpack_key(unsigned long long k, int& x, int& y, int& z) {
  x = (int)(k & 0x1FFFFF) - FVH_COORD_BIAS;
  y = (int)((k >> 21) & 0x1FFFFF) - FVH_COORD_BIAS;
  z = (int)((k >> 42) & 0x1FFFFF) - FVH_COORD_BIAS;
}
__device__ __forceinline__ bool coord_in_range(int x, int y, int z) {
  return (unsigned)(x + FVH_COORD_BIAS) < (1u << 21) && (unsigned)(y + FVH_COORD_BIAS) < (1u << 21) && (unsigned)(z + FVH_COORD_BIAS) < (1u << 21);
}
// (int)floor(v) is undefined for NaN/inf/huge v (a lidar return at infinity, a 1e9 outlier): test the floating value first.
// |f| < 2^20 - 2^12 leaves room for any neighbour offset the engine accepts and keeps the key inside its 21 bits.
template <typename T>
__device__ __forceinline__ bool voxel_index_ok(T fx, T fy, T fz) {
  const T lim = (T)(FVH_COORD_BIAS - 4096);
  return fabs(fx) < lim && fabs(fy) < lim && fabs(fz) < lim;  // false for NaN
}
// 64-bit mix (murmur3 finaliser); the bucket layout is private so the hash is ours to choose
// Multiplicative (Fibonacci) hash of the packed voxel key: two 32-bit multiplies. The GOOD bits are the high ones -- every
// input bit reaches them -- so tables index with hash_slot() (top log2(capacity) bits), never with the low bits.
// (Round 1 used the 64-bit murmur finaliser: two 64-bit multiplies = eight quarter-rate 32-bit multiplies per lookup, ~5 % of
// the VALU time of the LM kernel's main loop, for a table whose keys are small consecutive integers.)
__device__ __forceinline__ unsigned hash_key(unsigned long long k) {
  return (unsigned)k * 0x9E3779B1u + (unsigned)(k >> 32) * 0x85EBCA77u;
}
__device__ __forceinline__ unsigned hash_slot(unsigned long long k, unsigned mask /* capacity - 1, capacity a power of two */) {
  return mask ? hash_key(k) >> __builtin_clz(mask) : 0u;
}

// order-preserving float <-> uint mapping for atomicMin/Max
__device__ __forceinline__ unsigned float_to_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

}  // namespace fvh
