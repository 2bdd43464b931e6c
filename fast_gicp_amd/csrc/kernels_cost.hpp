// The GN/LM hot loop for gfx950: ONE kernel per linearize() / compute_error().
//
// Replaces SURVEY 2.2 K18-K23 + X2: find_voxel_correspondences (N_off launches + remove_if,
// find_voxel_correspondences.cu:16-111), compute_derivatives (transform_reduce of 43-float tuples,
// compute_derivatives.cu:18-184), the NDT variants (ndt_compute_derivatives.cu:33-231), the three
// tiny H2D copies and the blocking D2H per evaluation, and -- in device-LM mode -- the host side
// of LsqRegistration::step_lm (lsq_registration_impl.hpp:123-168).
//
// Work item = (source element i, offset group g). Adjacent lanes share the source element
// (broadcast loads). Per item: transform, voxel coordinate (fp64 like the CPU reference or fp32 like
// the CUDA one), probe the 64-B bucket table, Mahalanobis M = (C_B + R C_A R^T)^-1 with R of the
// LINEARISATION pose (fast_vgicp_impl.hpp:101-115 caches it; compute_derivatives.cu:71-72), residual
// and the 28 unique values {err, b(6), H_rr(6), H_rt(9), H_tt(6)} accumulated per thread in fp64
// registers. 3x3/6x6 math stays in VGPRs (no MFMA). Reduction: wave shuffles -> LDS across the 4
// waves -> one 28-double partial per workgroup written through to L2 (sc1) -> the LAST workgroup
// (atomic ticket) sums the partials in a fixed order and, in device-LM mode, runs the LM step
// (6x6 LDL^T, se3_exp, rho test, lambda schedule, convergence) so the next launch finds the new
// pose in HBM. No host round trip inside the loop.
#pragma once
#include <cstddef>

#include "dev_math.hpp"

// Contraction only inside one source expression (not across statements, which is hipcc's default "fast"): whether a
// multiply and an add fuse must not depend on what the optimiser happens to see around them -- the persistent and the
// per-transition instantiations of the cost kernel have to produce bit-identical sums.
#pragma clang fp contract(on)

namespace fvh {

constexpr int NSUM = 28;       // err(1) b(6) Hrr(6) Hrt(9) Htt(6)
constexpr int PART_STRIDE = 32;
constexpr int MAX_PARTIAL_ROWS = 512;  // workgroup rows; 8 group rows follow

enum CostMode { MODE_VGICP = 0, MODE_NDT_P2D = 1, MODE_NDT_D2D = 2 };
enum Phase { PH_LINEARIZE = 0, PH_TRIAL = 1, PH_DONE = 2, PH_FIND_ONLY = 3, PH_EVAL_DERIV = 4, PH_EVAL_ERROR = 5 };
// PH_TRIAL in device-LM mode is the FUSED launch: trial error with the old correspondences + speculative linearisation at the trial pose

struct LmState {
  PoseD x0;        // current estimate == linearisation pose
  PoseD xi;        // trial pose
  PoseD delta;     // se3_exp(d) of the last step
  PoseD x_lin;     // pose at which the CURRENT correspondence buffer was computed (reference: linearized_x)
  double H[36], b[6], d[6];
  double y0, lambda, nu;
  double final_H[36];
  double sums[PART_STRIDE];  // reduced {err, b, H...} of the last evaluation (all-reduced in multi-GPU mode)
  // parameters
  double rotation_epsilon, transformation_epsilon, lm_init_lambda_factor;
  int max_iterations, lm_max_iterations;
  // status
  int phase, outer_iter, inner_iter, converged, lm_failed, num_linearize, num_error_evals, nr_iterations;
  int corr_cur;       // which of the two correspondence buffers is current (device-LM mode flips it on accept)
  int vm_num_voxels;  // copied from the voxel map's counters by the last workgroup: capacity hint for the next build
  int vm_dropped;     // > 0: the hint-sized table overflowed -> host rebuilds at the safe size and re-runs
  int pad_;
  // LAST 8 bytes are never covered by the state write-back: `aborted` is raised by the persistent kernel's barrier
  // watchdog (the host zeroes the word before the launch and falls back to one launch per transition if it is set)
  unsigned gen, aborted;
};
static_assert(sizeof(LmState) % 8 == 0 && offsetof(LmState, gen) == sizeof(LmState) - 8, "barrier word must be the last 64-bit word of LmState");

struct CostParams {
  const float4* src_pts;
  const float4* src_cov;      // null for P2D
  const int* d_n_src;         // device-side count (D2D source voxels) or null
  const int* order;           // optional Morton permutation of the source: work item w handles element order[w] (coherent lookups)
  int n_src;
  const uint4* table;                // voxel records (64 B per bucket)
  const unsigned long long* keys;    // voxel keys of the same buckets, dense (kernels_voxelmap.hpp)
  unsigned mask;
  double res;
  const int* offsets;         // n_off x 3
  int n_off;
  int group;                  // offsets per work item
  int groups_per_src;         // ceil(n_off / group)
  int* corr;                  // 2 x [n_src][n_off] bucket index or -1 (double buffered for the speculative linearisation)
  size_t corr_stride;         // elements per buffer
  int host_corr_sel;          // host-mode launches: buffer to use
  LmState* st;
  double* partials;           // [gridDim.x][PART_STRIDE]
  unsigned* ticket;
  const int* vm_counters;     // target map {num_voxels, dropped}
  const int* vm_counters2;    // source map (D2D NDT) or null
  int host_phase;             // -1: device-LM mode (phase from st); else PH_FIND_ONLY / PH_EVAL_*
  int defer_lm;               // 1: multi-GPU -- only publish st->sums, LM step runs after the all-reduce
  PoseD lin, ev;              // host mode poses; device-LM first launch (init = 1): lin = initial guess
  int init;                   // 1: this is the first launch of an align -- start from P.lin and (re)initialise the LM state
  // persistent kernel: values of the 8 group counters + the top counter when this launch starts (they are not cleared between
  // launches). Separate scalars, not an array: an indexed kernel-argument array is copied to scratch memory.
  unsigned tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7, tb_top;
  double* bcast;              // persistent kernel: [PERSIST_REPLICAS][BCAST_SLOTS] broadcast rows
  unsigned long long launch_tag;  // persistent kernel: sequence number of this launch (tags of older launches never match)
  unsigned long long* result_host;  // persistent kernel: mapped pinned host memory, [sizeof(LmState)/8 words of state][sequence word] (null: not used)
  unsigned long long watchdog_ticks;  // persistent kernel: 100 MHz ticks a workgroup may wait at the barrier before it aborts the launch
  int max_iterations, lm_max_iterations;
  double rotation_epsilon, transformation_epsilon, lm_init_lambda_factor;
};

// ------------------------------------------------------------------------------------------------
// LM step on one thread (lsq_registration_impl.hpp:82-91,123-168; so3.hpp:58-104)
// ------------------------------------------------------------------------------------------------
__device__ inline void dev_se3_exp(const double a[6], PoseD& T) {
  // This runs on ONE lane between two evaluations of the cost, so its instruction count is latency on the critical path
  // of every LM transition (measured: four libm calls + 21 divisions made the persistent kernel 36 us slower per align).
  // One sincos of the half angle; the full-angle terms of the V matrix follow from the double-angle identities
  //   1 - cos(t) = 2 sin^2(t/2),   sin(t) = 2 sin(t/2) cos(t/2)
  // (so3.hpp:58-104 calls sin/cos four times; oracle probe ORC_LM_ARITH_VARIANT=2: converged poses agree to 5e-15).
  const double ox = a[0], oy = a[1], oz = a[2];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  const double theta = sqrt(theta_sq);
  double sh = 0.0, ch = 1.0;
  if (theta >= 1e-10) sincos(0.5 * theta, &sh, &ch);  // needed by the V matrix even when the quaternion takes its Taylor branch
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    imag = sh / theta;
    real = ch;
  }
  const double qw = real, qx = imag * ox, qy = imag * oy, qz = imag * oz;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T.r[0] = 1 - (tyy + tzz); T.r[1] = txy - twz;       T.r[2] = txz + twy;
  T.r[3] = txy + twz;       T.r[4] = 1 - (txx + tzz); T.r[5] = tyz - twx;
  T.r[6] = txz - twy;       T.r[7] = tyz + twx;       T.r[8] = 1 - (txx + tyy);
  double V[9];
  if (theta < 1e-10) {
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = T.r[i];
  } else {
    const double inv_tsq = 1.0 / theta_sq;
    const double A = 2.0 * sh * sh * inv_tsq, B = (theta - 2.0 * sh * ch) * inv_tsq / theta;
    // Omega = skew(omega), Omega^2
    const double O[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double O2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3 + 0] * O[0 * 3 + j] + O[i * 3 + 1] * O[1 * 3 + j] + O[i * 3 + 2] * O[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + A * O[i] + B * O2[i];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) T.t[i] = V[i * 3 + 0] * a[3] + V[i * 3 + 1] * a[4] + V[i * 3 + 2] * a[5];
}

__device__ inline void dev_pose_mul(const PoseD& A, const PoseD& B, PoseD& C) {  // C = A * B
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) C.r[i * 3 + j] = A.r[i * 3 + 0] * B.r[0 * 3 + j] + A.r[i * 3 + 1] * B.r[1 * 3 + j] + A.r[i * 3 + 2] * B.r[2 * 3 + j];
    C.t[i] = A.r[i * 3 + 0] * B.t[0] + A.r[i * 3 + 1] * B.t[1] + A.r[i * 3 + 2] * B.t[2] + A.t[i];
  }
}

__device__ inline void dev_ldlt6_solve(const double* A, const double* rhs, double* x) {
  // one reciprocal per pivot (6 divisions instead of 21: each fp64 division is ~12 dependent instructions on the one
  // lane that runs this; oracle probe ORC_LM_ARITH_VARIANT=1: converged poses agree to 2e-17); only the strictly
  // lower part of L is used. A pivot with |d| <= DBL_MIN is treated the way Eigen::LDLT does (the column stays unscaled,
  // the solve uses the pseudo-inverse of D): with no correspondences at all H = 0, lambda = 0 and the step is d = 0, so the
  // reference returns the initial guess flagged converged (lsq_registration_impl.hpp:111-168) instead of a NaN pose.
  double L[36], D[6], Dinv[6], y[6];
  for (int j = 0; j < 6; j++) {
    double dj = A[j * 6 + j];
    for (int k = 0; k < j; k++) dj -= L[j * 6 + k] * L[j * 6 + k] * D[k];
    D[j] = dj;
    const bool pivot_ok = fabs(dj) > 2.2250738585072014e-308;
    Dinv[j] = pivot_ok ? 1.0 / dj : 0.0;
    for (int i = j + 1; i < 6; i++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k] * D[k];
      L[i * 6 + j] = pivot_ok ? s * Dinv[j] : s;
    }
  }
  for (int i = 0; i < 6; i++) { double s = rhs[i]; for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k]; y[i] = s; }
  for (int i = 0; i < 6; i++) y[i] *= Dinv[i];
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k]; x[i] = s; }
}

__device__ inline bool dev_is_converged(const LmState* st, const PoseD& delta) {
  // lsq_registration_impl.hpp:82-91: (|R - I|.max / rot_eps, |t|.max / trans_eps).max < 1 -- the max first, two divisions
  double rmax = 0, tmax = 0;
  for (int i = 0; i < 9; i++) rmax = fmax(rmax, fabs(delta.r[i] - ((i % 4 == 0) ? 1.0 : 0.0)));
  for (int i = 0; i < 3; i++) tmax = fmax(tmax, fabs(delta.t[i]));
  return fmax(rmax / st->rotation_epsilon, tmax / st->transformation_epsilon) < 1;
}

// sums -> symmetric 6x6 H (row-major) and b
__device__ __host__ inline void unpack_sums(const double* s, double* H, double* b) {
  for (int i = 0; i < 6; i++) b[i] = s[1 + i];
  const double* rr = s + 7;   // xx xy xz yy yz zz
  const double* rt = s + 13;  // 3x3 row-major
  const double* tt = s + 22;
  const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      H[i * 6 + j] = rr[sym[i][j]];
      H[i * 6 + 3 + j] = rt[i * 3 + j];
      H[(3 + j) * 6 + i] = rt[i * 3 + j];
      H[(3 + i) * 6 + 3 + j] = tt[sym[i][j]];
    }
}

__device__ inline void dev_lm_propose(LmState* st) {  // d = (H + lambda I)^-1 (-b); xi = exp(d) * x0
  double A[36], nb[6];
  for (int i = 0; i < 36; i++) A[i] = st->H[i];
  for (int j = 0; j < 6; j++) { A[j * 6 + j] += st->lambda; nb[j] = -st->b[j]; }
  double d[6];
  dev_ldlt6_solve(A, nb, d);
  for (int j = 0; j < 6; j++) st->d[j] = d[j];
  PoseD delta;
  dev_se3_exp(d, delta);
  st->delta = delta;
  dev_pose_mul(delta, st->x0, st->xi);
}

// One transition of the {linearize -> trial* -> accept} machine; exactly the control flow of
// LsqRegistration::computeTransformation + step_lm.  sums[0..27] = {err, b, H} of the linearisation this
// launch computed (at x0 for PH_LINEARIZE, speculatively at xi for the fused PH_TRIAL launch), sums[28] =
// trial error y_i at xi with the OLD correspondences (fused launch only).
__device__ inline void dev_lm_consume_linearization(LmState* st, const double* sums) {
  st->y0 = sums[0];
  unpack_sums(sums, st->H, st->b);
  st->num_linearize++;
  st->nr_iterations = st->outer_iter;
  if (st->lambda < 0.0) {
    double mx = 0;
    for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(st->H[i * 6 + i]));
    st->lambda = st->lm_init_lambda_factor * mx;
  }
  st->nu = 2.0;
  st->inner_iter = 0;
}

__device__ inline void dev_lm_step(LmState* st, const double* sums) {
  if (st->phase == PH_LINEARIZE) {
    st->x_lin = st->x0;
    dev_lm_consume_linearization(st, sums);
    if (st->lm_max_iterations <= 0) { st->lm_failed = 1; st->phase = PH_DONE; return; }
    dev_lm_propose(st);
    st->phase = PH_TRIAL;
    return;
  }
  // PH_TRIAL (fused)
  const double yi = sums[28];
  st->num_error_evals++;
  double denom = 0;
  for (int j = 0; j < 6; j++) denom += st->d[j] * (st->lambda * st->d[j] - st->b[j]);
  const double rho = (st->y0 - yi) / denom;
  if (rho < 0) {
    if (dev_is_converged(st, st->delta)) {  // step_lm returns true with x0 unchanged -> converged_ = true
      st->converged = 1;
      st->outer_iter++;
      st->phase = PH_DONE;
      return;
    }
    st->lambda = st->nu * st->lambda;
    st->nu = 2 * st->nu;
    st->inner_iter++;
    if (st->inner_iter >= st->lm_max_iterations) { st->lm_failed = 1; st->phase = PH_DONE; return; }  // "lm not converged!!"
    dev_lm_propose(st);  // new trial from the SAME (H, b); the speculative linearisation of this launch is discarded
    return;
  }
  // accepted
  st->x0 = st->xi;
  { const double u = 2 * rho - 1; st->lambda = st->lambda * fmax(1.0 / 3.0, 1 - u * u * u); }
  for (int i = 0; i < 36; i++) st->final_H[i] = st->H[i];
  st->converged = dev_is_converged(st, st->delta) ? 1 : 0;
  st->outer_iter++;
  if (st->converged || st->outer_iter >= st->max_iterations) { st->phase = PH_DONE; return; }
  // the speculative linearisation at xi (== the new x0) is exactly the next step_lm's linearize()
  st->corr_cur ^= 1;
  st->x_lin = st->x0;
  dev_lm_consume_linearization(st, sums);
  dev_lm_propose(st);
}

// tiny kernels for the multi-GPU path and for (re)initialising the state
__global__ void lm_init_kernel(LmState* st, PoseD guess, double rot_eps, double trans_eps, double lambda_factor, int max_iter, int lm_max_iter, unsigned* ticket) {
  if (blockIdx.x == 0 && threadIdx.x <= 8) ticket[threadIdx.x] = 0;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->x0 = guess; st->xi = guess;
  st->rotation_epsilon = rot_eps; st->transformation_epsilon = trans_eps; st->lm_init_lambda_factor = lambda_factor;
  st->max_iterations = max_iter; st->lm_max_iterations = lm_max_iter;
  st->lambda = -1.0; st->nu = 2.0; st->y0 = 0.0;
  st->phase = max_iter > 0 ? PH_LINEARIZE : PH_DONE;
  st->corr_cur = 0; st->x_lin = guess;
  st->outer_iter = 0; st->inner_iter = 0; st->converged = 0; st->lm_failed = 0; st->num_linearize = 0; st->num_error_evals = 0; st->nr_iterations = 0;
  for (int i = 0; i < 36; i++) st->final_H[i] = (i % 7 == 0) ? 1.0 : 0.0;
}
__global__ void lm_update_kernel(LmState* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (st->phase == PH_DONE) return;
  dev_lm_step(st, st->sums);
}

// ------------------------------------------------------------------------------------------------
// per-correspondence terms
// ------------------------------------------------------------------------------------------------
template <typename Real>
__device__ __forceinline__ void accumulate_term(double* acc, const Vec3<Real>& q, const Vec3<Real>& mu, const Sym3<Real>& M, Real w, bool deriv) {
  const Vec3<Real> e = {mu.x - q.x, mu.y - q.y, mu.z - q.z};
  const Vec3<Real> Me = mul(M, e);
  acc[0] += (double)(w * (e.x * Me.x + e.y * Me.y + e.z * Me.z));
  if (!deriv) return;
  // J = [skew(q), -I]:  b = w J^T M e = w [Me x q ; -Me]
  const Vec3<Real> bq = cross(Me, q);
  acc[1] += (double)(w * bq.x); acc[2] += (double)(w * bq.y); acc[3] += (double)(w * bq.z);
  acc[4] -= (double)(w * Me.x); acc[5] -= (double)(w * Me.y); acc[6] -= (double)(w * Me.z);
  // P = skew(q) M (columns q x M_col) ; H = w [[P S^T, P], [P^T, M]]
  const Vec3<Real> c0 = {M.xx, M.xy, M.xz}, c1 = {M.xy, M.yy, M.yz}, c2 = {M.xz, M.yz, M.zz};
  const Vec3<Real> p0 = cross(q, c0), p1 = cross(q, c1), p2 = cross(q, c2);  // P_ij = p_j[i]
  const Vec3<Real> r0 = {p0.x, p1.x, p2.x}, r1 = {p0.y, p1.y, p2.y}, r2 = {p0.z, p1.z, p2.z};  // rows of P
  const Vec3<Real> h0 = cross(q, r0), h1 = cross(q, r1), h2 = cross(q, r2);  // rows of H_rr
  acc[7] += (double)(w * h0.x); acc[8] += (double)(w * h0.y); acc[9] += (double)(w * h0.z);
  acc[10] += (double)(w * h1.y); acc[11] += (double)(w * h1.z); acc[12] += (double)(w * h2.z);
  acc[13] += (double)(w * r0.x); acc[14] += (double)(w * r0.y); acc[15] += (double)(w * r0.z);
  acc[16] += (double)(w * r1.x); acc[17] += (double)(w * r1.y); acc[18] += (double)(w * r1.z);
  acc[19] += (double)(w * r2.x); acc[20] += (double)(w * r2.y); acc[21] += (double)(w * r2.z);
  acc[22] += (double)(w * M.xx); acc[23] += (double)(w * M.xy); acc[24] += (double)(w * M.xz);
  acc[25] += (double)(w * M.yy); acc[26] += (double)(w * M.yz); acc[27] += (double)(w * M.zz);
}

// Continue a linear probe from `slot` (the first bucket has already been inspected).
__device__ __forceinline__ int probe_continue(const unsigned long long* __restrict__ keys, unsigned mask, unsigned long long key, unsigned slot) {
  for (unsigned it = 0; it < mask; it++) {
    slot = (slot + 1) & mask;
    const unsigned long long k = keys[slot];
    if (k == key) return (int)slot;
    if (k == FVH_EMPTY_KEY) return -1;  // first empty bucket ends the probe (find_voxel_correspondences.cu:46-48)
  }
  return -1;
}

constexpr int COST_CH = 4;     // voxel lookups a thread keeps in flight at once
constexpr int PERSIST_TICKET_BYTES = 9 * 128;  // persistent kernel: 8 group counters + top counter, one 128-B line each
constexpr int PERSIST_REPLICAS = 32;           // copies of the broadcast row; workgroup b polls copy b % PERSIST_REPLICAS
constexpr int BCAST_SLOTS = 40;                // 5 segments of 64 B = 7 sums + 1 tag each (35 >= 29 sums)
constexpr int TICKET_GROUPS = 8;  // hierarchical arrival counters (one per XCD-sized group of workgroups) + 1 top counter

// PERSIST = true: ONE launch runs the whole LM loop. Every trip of the outer loop is what one launch of the
// non-persistent kernel does; instead of exiting, the workgroups wait at a barrier (monotonic arrival counters polled
// with agent-scope loads), then every workgroup finishes the reduction and runs the LM step itself on its own LDS
// copy of the state, and goes again until the state says PH_DONE. This removes the per-launch
// dispatch + ramp (~4.5 us of an 18 us launch at 17k points, FVH_COST_TIMING) and the speculative no-op launches.
// All workgroups must be co-resident (the host clamps the grid to the occupancy limit); a watchdog turns a stuck
// barrier into an abort flag + fallback to the multi-launch path instead of a hang.
__device__ __forceinline__ double uniform_f64(double x) {  // wave-uniform value -> SGPR pair
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#ifdef FVH_COST_TIMING
// Debug build only (-DFVH_COST_TIMING): 100 MHz wall-clock stamps of the LAST workgroup's walk through the epilogue
// (slot 0 = earliest workgroup start of the launch, slot 9 = the last workgroup's own start). Read with
// fvh_debug_cost_timing(); tools/cost_timing.py prints the breakdown.
__device__ unsigned long long g_cost_timing[16];
__device__ unsigned long long g_ptime[16][512][12];  // persistent kernel, per trip and workgroup: {start, main end, arrival, open (opener only), seen, LM done}; plain stores, no shared address
#define FVH_STAMP(i) do { if (threadIdx.x == 0) stamp[i] = wall_clock64(); } while (0)
#define FVH_PT_MIN(trip, k) do { if (threadIdx.x == 0 && (trip) < 16 && blockIdx.x < 512) g_ptime[trip][blockIdx.x][k] = wall_clock64(); } while (0)
#define FVH_PT_MAX(trip, k) FVH_PT_MIN(trip, k)
#else
#define FVH_STAMP(i) do { } while (0)
#define FVH_PT_MIN(trip, k) do { } while (0)
#define FVH_PT_MAX(trip, k) do { } while (0)
#endif

// 2 workgroups per CU are part of the design (the persistent grid must be co-resident, 66 KB of LDS each): tell the
// register allocator, which otherwise drifts over 256 VGPRs + AGPRs with small code changes and halves the grid.
#define FVH_COST_BOUNDS __launch_bounds__(256, 2)
template <typename Real, int MODE, bool PERSIST>
__global__ FVH_COST_BOUNDS void cost_kernel(CostParams P) {
#ifdef FVH_COST_TIMING
  __shared__ unsigned long long stamp[12];  // LDS, not registers: must not change the kernel being measured
  if (threadIdx.x == 0) { stamp[0] = wall_clock64(); atomicMin(&g_cost_timing[0], stamp[0]); }
#endif
  __shared__ double red[4][PART_STRIDE];
  __shared__ double fin[8][PART_STRIDE];
  __shared__ int s_last;
  __shared__ LmState s_st;
  static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as 64-bit words");
  constexpr int ST_WORDS = sizeof(LmState) / 8 - 1;  // without the barrier word
  static_assert(ST_WORDS <= 256, "one state word per thread after the barrier");
  LmState* st = P.st;
  unsigned long long* st_words = reinterpret_cast<unsigned long long*>(st);
  unsigned gen = 0;  // PERSIST: barrier generations this workgroup has passed
  int phase, corr_sel;
  PoseD lin_d, ev_d;
  if (P.host_phase >= 0) {
    phase = P.host_phase;
    lin_d = P.lin;
    ev_d = P.ev;
    corr_sel = P.host_corr_sel;
  } else if (P.init) {  // first launch of an align: the state in memory is stale, everything comes from the kernel arguments
    phase = PH_LINEARIZE;
    lin_d = P.lin;
    ev_d = P.lin;
    corr_sel = 0;
  } else {
    phase = st->phase;
    if (phase == PH_DONE) return;
    lin_d = st->x_lin;  // == x0 for PH_LINEARIZE
    ev_d = (phase == PH_LINEARIZE) ? st->x0 : st->xi;
    corr_sel = st->corr_cur;
  }
  __shared__ Pose<Real> s_pose[2];
  if (threadIdx.x == 0) { s_pose[0] = pose_cast<Real>(lin_d); s_pose[1] = pose_cast<Real>(ev_d); }
  __syncthreads();
  for (;;) {  // PERSIST: one trip per LM transition; otherwise exactly one trip
  if (PERSIST) FVH_PT_MIN(gen, 0);
  const bool fused = (P.host_phase < 0) && (phase == PH_TRIAL);  // trial error (old ids) + speculative linearisation at xi (new ids)
  const bool do_find = (phase == PH_LINEARIZE) || (phase == PH_FIND_ONLY) || fused;
  const bool do_cost = (phase != PH_FIND_ONLY);
  const bool do_deriv = (phase == PH_LINEARIZE) || (phase == PH_EVAL_DERIV) || fused;
  // The two poses are wave-uniform: as scalars they cost 48 SGPRs for the whole main loop, and this kernel already
  // spills hundreds of SGPRs into VGPR lanes (446 in the persistent variant, which pushed it one register past the 256
  // VGPRs two workgroups per CU allow). They live in LDS instead and are read per element, where their registers die
  // before the lookups start.
  // (s_pose[0] = lin: rotation used by the cached Mahalanobis of the OLD ids; s_pose[1] = ev: evaluation pose, also the
  // linearisation pose of the NEW ids; filled before the first trip and by the barrier code of every persistent trip)
  const Real res = (Real)P.res;
  const int n_src = P.d_n_src ? *P.d_n_src : P.n_src;
  const int n_items = n_src * P.groups_per_src;
  int* corr_old = P.corr + (size_t)corr_sel * P.corr_stride;                      // read (stored ids)
  int* corr_new = fused ? P.corr + (size_t)(corr_sel ^ 1) * P.corr_stride : corr_old;  // written by the find

  double acc[NSUM];
#pragma unroll
  for (int v = 0; v < NSUM; v++) acc[v] = 0.0;
  double acc_y = 0.0;  // fused: trial error with the old ids

  const float4* tf = reinterpret_cast<const float4*>(P.table);
  // (consecutive threads take consecutive items on purpose: spreading a workgroup's items over the cloud made the launch
  // 24 % slower -- the loop is sensitive to how many distinct cache lines a wave touches)
  for (int w = blockIdx.x * 256 + threadIdx.x; w < n_items; w += gridDim.x * 256) {
    const int i0 = w / P.groups_per_src;
    const int g = w - i0 * P.groups_per_src;
    const int i = P.order ? P.order[i0] : i0;
    // ---- round trip 1: the source element ----
    const float4 a4 = P.src_pts[i];
    float4 c0 = make_float4(0, 0, 0, 0), c1 = c0;
    if (MODE != MODE_NDT_P2D && do_cost) { c0 = P.src_cov[2 * i]; c1 = P.src_cov[2 * i + 1]; }
    const Vec3<Real> a = {(Real)a4.x, (Real)a4.y, (Real)a4.z};
    const Pose<Real>* pose_ptr = s_pose;
    asm volatile("" : "+v"(pose_ptr));  // opaque: the loads below stay inside the iteration instead of becoming 48 loop-invariant registers
    const Pose<Real> lin = pose_ptr[0], ev = pose_ptr[1];
    Sym3<Real> RCR = {0, 0, 0, 0, 0, 0}, RCR_old = {0, 0, 0, 0, 0, 0};
    if (MODE != MODE_NDT_P2D && do_cost) {
      const Sym3<Real> CA = {(Real)c0.x, (Real)c0.y, (Real)c0.z, (Real)c0.w, (Real)c1.x, (Real)c1.y};
      // the ids found by THIS launch are linearised at `ev` when fused, at `lin` otherwise
      RCR = rotate_cov(fused ? ev.r : lin.r, CA);
      if (fused) RCR_old = rotate_cov(lin.r, CA);
    }
    const Vec3<Real> q = transform(ev, a);
    int cx = 0, cy = 0, cz = 0;
    bool coord_ok = true;  // false: non-finite / out-of-range source point -> no correspondences (and no (int)floor(NaN))
    if (do_find) {
      const Vec3<Real> ql = fused ? q : transform(lin, a);
      const Real fx = floor(ql.x / res - (Real)0.5), fy = floor(ql.y / res - (Real)0.5), fz = floor(ql.z / res - (Real)0.5);
      coord_ok = voxel_index_ok(fx, fy, fz);
      cx = coord_ok ? (int)fx : 0;
      cy = coord_ok ? (int)fy : 0;
      cz = coord_ok ? (int)fz : 0;
    }
    const int o_begin = g * P.group, o_end = min(P.n_off, o_begin + P.group);
    for (int oc = o_begin; oc < o_end; oc += COST_CH) {
      int b[COST_CH], bo[COST_CH];
      // ---- round trip 2: COST_CH independent lookups in flight (first probe of each, and/or the stored ids) ----
      if (fused) {
#pragma unroll
        for (int c = 0; c < COST_CH; c++) bo[c] = (oc + c < o_end) ? corr_old[(size_t)i * P.n_off + oc + c] : -1;
      }
      if (do_find) {
        unsigned long long key[COST_CH];
        unsigned slot[COST_CH];
        unsigned long long k0[COST_CH];
        bool live[COST_CH];
#pragma unroll
        for (int c = 0; c < COST_CH; c++) {
          const int o = min(oc + c, o_end - 1);
          const int x = cx + P.offsets[3 * o], y = cy + P.offsets[3 * o + 1], z = cz + P.offsets[3 * o + 2];
          live[c] = (oc + c < o_end) && coord_ok && coord_in_range(x, y, z);
          key[c] = pack_key(x, y, z);
          slot[c] = hash_key(key[c]) & P.mask;
          k0[c] = P.keys[slot[c]];
        }
#pragma unroll
        for (int c = 0; c < COST_CH; c++) {
          const unsigned long long k = k0[c];
          int r = -1;
          if (live[c]) {
            if (k == key[c]) r = (int)slot[c];
            else if (k != FVH_EMPTY_KEY) r = probe_continue(P.keys, P.mask, key[c], slot[c]);  // rare at load <= 0.25
          }
          b[c] = r;
          if (oc + c < o_end) corr_new[(size_t)i * P.n_off + oc + c] = r;
        }
      } else {
#pragma unroll
        for (int c = 0; c < COST_CH; c++) b[c] = (oc + c < o_end) ? corr_old[(size_t)i * P.n_off + oc + c] : -1;
      }
      if (!do_cost) continue;
      // ---- round trip 3: the voxel records of all hits, unconditional loads (bucket 0 for misses) ----
      float4 q1[COST_CH], q2[COST_CH], q3[COST_CH];
#pragma unroll
      for (int c = 0; c < COST_CH; c++) {
        const size_t base = (size_t)max(b[c], 0) * 4;
        q1[c] = tf[base + 1]; q2[c] = tf[base + 2]; q3[c] = tf[base + 3];
      }
      if (fused) {  // trial error with the OLD ids (usually the same buckets -> the lines are already on their way)
#pragma unroll
        for (int c = 0; c < COST_CH; c++) {
          if (bo[c] < 0) continue;
          const size_t base = (size_t)bo[c] * 4;
          const float4 o1 = tf[base + 1], o2 = tf[base + 2], o3 = tf[base + 3];
          const int npts = (int)o1.w;
          const Vec3<Real> mu = {(Real)o1.x, (Real)o1.y, (Real)o1.z};
          const Sym3<Real> A = {(Real)o2.x + RCR_old.xx, (Real)o2.y + RCR_old.xy, (Real)o2.z + RCR_old.xz, (Real)o2.w + RCR_old.yy, (Real)o3.x + RCR_old.yz, (Real)o3.y + RCR_old.zz};
          const Vec3<Real> e = {mu.x - q.x, mu.y - q.y, mu.z - q.z};
          Real wgt;
          if (MODE == MODE_VGICP) {
            if (npts <= 0) continue;
            wgt = sqrt((Real)npts);
          } else {
            if (npts <= 6) continue;
            const Real ksq = res * res;
            wgt = ksq / (ksq + (e.x * e.x + e.y * e.y + e.z * e.z));
          }
          const Sym3<Real> M = inverse(A);
          const Vec3<Real> Me = mul(M, e);
          acc_y += (double)(wgt * (e.x * Me.x + e.y * Me.y + e.z * Me.z));
        }
      }
#pragma unroll
      for (int c = 0; c < COST_CH; c++) {
        if (b[c] < 0) continue;
        const int npts = (int)q1[c].w;
        const Vec3<Real> mu = {(Real)q1[c].x, (Real)q1[c].y, (Real)q1[c].z};
        const Sym3<Real> A = {(Real)q2[c].x + RCR.xx, (Real)q2[c].y + RCR.xy, (Real)q2[c].z + RCR.xz, (Real)q2[c].w + RCR.yy, (Real)q3[c].x + RCR.yz, (Real)q3[c].y + RCR.zz};
        Real wgt;
        if (MODE == MODE_VGICP) {
          if (npts <= 0) continue;
          wgt = sqrt((Real)npts);  // fast_vgicp_impl.hpp:149, compute_derivatives.cu:78
        } else {
          if (npts <= 6) continue;  // ndt_compute_derivatives.cu:61,133
          const Real ex = mu.x - q.x, ey = mu.y - q.y, ez = mu.z - q.z;
          const Real ksq = res * res;
          wgt = ksq / (ksq + (ex * ex + ey * ey + ez * ez));  // cauchy(resolution, |e|) :15-18
        }
        const Sym3<Real> M = inverse(A);
        accumulate_term<Real>(acc, q, mu, M, wgt, do_deriv);
      }
    }
  }

  // ---- workgroup reduction through an LDS transpose ------------------------------------------
  // Every thread stores its values to tile[v][thread] (row stride 264 doubles: conflict-free for
  // both phases), then thread (v = t/8, part = t%8) sums the 32 entries part, part+8, ... of row v
  // and 3 shuffle steps combine the 8 parts. ~66 LDS operations per thread instead of the 336
  // ds_bpermute of a per-value wave-shuffle tree (measured: the (H,b) launch cost 7 us more than
  // the error-only launch, almost all of it crossbar traffic).
  if (!do_cost) return;  // host-mode PH_FIND_ONLY (never persistent)
  FVH_STAMP(1);
  if (PERSIST) FVH_PT_MAX(gen, 1);
  constexpr int RED_ROWS = NSUM + 1, RED_STRIDE = 264;
  __shared__ double tile[RED_ROWS * RED_STRIDE];
  const int nsum = do_deriv ? NSUM : 1;
  const int t = threadIdx.x;
#pragma unroll
  for (int v = 0; v < NSUM; v++)
    if (v < nsum) tile[v * RED_STRIDE + t] = acc[v];  // static indexing keeps acc[] in VGPRs
  if (fused) tile[NSUM * RED_STRIDE + t] = acc_y;
  __syncthreads();
  {
    const int v = t >> 3, part = t & 7;
    const bool live = (v < nsum) || (fused && v == NSUM);
    double x = 0.0;
    if (v < RED_ROWS && live) {
      const double* row = tile + v * RED_STRIDE + part;
#pragma unroll 8
      for (int j = 0; j < 32; j++) x += row[8 * j];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) x += __shfl_xor(x, off);
    // write-through (sc1) so another workgroup can read it from L2 without a release fence;
    // row slot v: sums 0..27, fused trial error at 28 (== NSUM), 29..31 zero
    if (part == 0 && v < PART_STRIDE) {
      __hip_atomic_store(&P.partials[(size_t)blockIdx.x * PART_STRIDE + v], (v < RED_ROWS && live) ? x : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  FVH_STAMP(2);

  // ---- two-level arrival + two-level reduction ----------------------------------------------
  // Workgroup b belongs to group b % 8 (its XCD under the observed dispatch order). The last arriver
  // of a group sums that group's partial rows with all 256 threads (<= 8 independent sc1 loads per
  // thread, fixed order), publishes one group row and arrives at the top counter; the last group
  // sums the <= 8 group rows and runs the LM step. No address sees more than gridDim/8 + 8 atomics
  // and no thread walks a long chain of dependent L2 round trips.
  const unsigned grp = blockIdx.x % TICKET_GROUPS;
  const unsigned ngroups = min((unsigned)TICKET_GROUPS, gridDim.x);
  const unsigned gsize = (gridDim.x - grp + TICKET_GROUPS - 1) / TICKET_GROUPS;
  // sum of this group's partial rows -> fin[chunk][v] -> one group row (fixed order: deterministic)
  auto reduce_group_rows = [&](size_t out_row) {
    {
      const int v = threadIdx.x & 31, chunk = threadIdx.x >> 5;  // 8 chunks x 32 values
      double s = 0.0;
      for (unsigned j0 = chunk; j0 < gsize; j0 += 8 * 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const unsigned j = j0 + 8 * u;
          t[u] = (j < gsize) ? __hip_atomic_load(&P.partials[(size_t)(grp + j * TICKET_GROUPS) * PART_STRIDE + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) s += t[u];
      }
      fin[chunk][v] = s;
    }
    __syncthreads();
    if (threadIdx.x < PART_STRIDE) {
      const int v = threadIdx.x;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; c++) s += fin[c][v];
      __hip_atomic_store(&P.partials[out_row * PART_STRIDE + v], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  // sum of the <= 8 group rows in group order -> red[0][v]
  auto reduce_final = [&](size_t first_row) {
    {
      const int v = threadIdx.x & 31;
      const unsigned g = threadIdx.x >> 5;
      fin[g][v] = (g < ngroups) ? __hip_atomic_load(&P.partials[(first_row + g) * PART_STRIDE + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < PART_STRIDE) {
      const int v = threadIdx.x;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; c++) s += fin[c][v];
      red[0][v] = s;
    }
  };
  // what lm_init_kernel would have written (first evaluation of an align), on the LDS copy of the state
  auto init_state = [&]() {
    s_st.x0 = P.lin; s_st.xi = P.lin; s_st.x_lin = P.lin;
    s_st.rotation_epsilon = P.rotation_epsilon; s_st.transformation_epsilon = P.transformation_epsilon; s_st.lm_init_lambda_factor = P.lm_init_lambda_factor;
    s_st.max_iterations = P.max_iterations; s_st.lm_max_iterations = P.lm_max_iterations;
    s_st.lambda = -1.0; s_st.nu = 2.0; s_st.y0 = 0.0;
    s_st.phase = PH_LINEARIZE; s_st.corr_cur = 0;
    s_st.outer_iter = 0; s_st.inner_iter = 0; s_st.converged = 0; s_st.lm_failed = 0; s_st.num_linearize = 0; s_st.num_error_evals = 0; s_st.nr_iterations = 0;
    for (int i = 0; i < 36; i++) s_st.final_H[i] = (i % 7 == 0) ? 1.0 : 0.0;
  };

  if constexpr (!PERSIST) {
    // ---- two-level arrival + two-level reduction --------------------------------------------
    // Workgroup b belongs to group b % 8 (its XCD under the observed dispatch order). The last arriver
    // of a group sums that group's partial rows with all 256 threads (<= 8 independent sc1 loads per
    // thread, fixed order), publishes one group row and arrives at the top counter; the last group
    // sums the <= 8 group rows and runs the LM step. No address sees more than gridDim/8 + 8 atomics
    // and no thread walks a long chain of dependent L2 round trips.
    if (threadIdx.x == 0) s_last = (atomicAdd(&P.ticket[grp], 1u) == gsize - 1);
    __syncthreads();
    if (!s_last) return;
    FVH_STAMP(3);
    reduce_group_rows((size_t)MAX_PARTIAL_ROWS + grp);
    if (threadIdx.x == 0) s_last = (atomicAdd(&P.ticket[TICKET_GROUPS], 1u) == ngroups - 1);
    __syncthreads();
    if (!s_last) return;
    FVH_STAMP(4);

    // ---- the very last workgroup: sum the group rows in group order (deterministic), LM step ----
    reduce_final((size_t)MAX_PARTIAL_ROWS);
    if (threadIdx.x <= TICKET_GROUPS) P.ticket[threadIdx.x] = 0;  // re-arm for the next launch
    // The LM step is one thread of dependent fp64 math; run it on an LDS copy of the state (a global
    // round trip per st-> access would cost more than the arithmetic) and write the state back with all lanes.
    for (int i = threadIdx.x; i < ST_WORDS; i += 256) reinterpret_cast<unsigned long long*>(&s_st)[i] = st_words[i];
    int vm_nv = 0, vm_dr = 0;
    if (threadIdx.x == 0) {
      vm_nv = P.vm_counters[0];
      vm_dr = P.vm_counters[1] + (P.vm_counters2 ? P.vm_counters2[1] : 0);
    }
    __syncthreads();
    FVH_STAMP(5);
    if (threadIdx.x == 0) {
      for (int v = 0; v < PART_STRIDE; v++) s_st.sums[v] = red[0][v];
      s_st.vm_num_voxels = vm_nv;
      s_st.vm_dropped = vm_dr;
      if (P.host_phase < 0 && P.init) init_state();
      if (P.host_phase < 0 && !P.defer_lm) dev_lm_step(&s_st, red[0]);
    }
    FVH_STAMP(6);
    __syncthreads();
    for (int i = threadIdx.x; i < ST_WORDS; i += 256) st_words[i] = reinterpret_cast<const unsigned long long*>(&s_st)[i];
#ifdef FVH_COST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      stamp[7] = wall_clock64();
      for (int i = 1; i <= 7; i++) g_cost_timing[i] = stamp[i];
      g_cost_timing[8] = gridDim.x;
      g_cost_timing[9] = stamp[0];
    }
#endif
    return;
  } else {
    // ---- persistent trip: the same two-level arrival, but nobody leaves -------------------------
    // Counters are monotonic -- over the launch and across launches (the host passes their starting values instead
    // of clearing them: a memset is a 3.5 us operation on the stream): in trip t the last arriver of a group draws
    // tbase + gsize * (t + 1) - 1 and, after reducing its group's rows, bumps the top counter. Workgroup 0 (the "opener")
    // waits for tbase + ngroups * (t + 1) there, sums the <= 8 group rows, runs the LM step and BROADCASTS
    // what the next trip needs -- phase, correspondence buffer, the two poses: 26 values -- as PERSIST_REPLICAS copies
    // of a 40-double row in which every 64-byte segment is 7 values + a tag (launch sequence, trip), written by 8
    // adjacent lanes of one store instruction. Workgroup b polls copy b % PERSIST_REPLICAS with ONE 40-lane load per
    // poll: when all 5 tags match, the values in the same segments are this trip's -- barrier and payload in a single
    // memory round trip, and no address is read by more than ~8 workgroups.
    // Dead ends measured on the way (474 workgroups, 17k points): one barrier word on the line of the arrival counters
    // (+10 us per trip); one barrier word + every workgroup reloading the state (21.6 us per trip: ~500 readers of the
    // same lines queue at their memory channel); every workgroup running the LM step redundantly on its own copy
    // (22.7 us per trip: the step takes 5 us instead of 1.5 when ~500 waves fetch its code at once).
    // Group rows alternate by trip parity; only the opener reads them, and trip t + 1's rows are written after every
    // workgroup -- the opener included -- has arrived at trip t + 1.
    const unsigned trip = gen;
    const double want_tag = (double)(P.launch_tag * 4096ull + trip + 1);
    const double abort_tag = -(double)(P.launch_tag * 4096ull);  // launch-specific: a poisoned row of an older launch means nothing
    const size_t grow0 = (size_t)MAX_PARTIAL_ROWS + (size_t)(trip & 1u) * TICKET_GROUPS;
    __shared__ double bc[BCAST_SLOTS];  // payload of the broadcast row as seen by this workgroup
    static_assert(TICKET_GROUPS == 8, "tb0..tb7");
    const unsigned tb = grp == 0 ? P.tb0 : grp == 1 ? P.tb1 : grp == 2 ? P.tb2 : grp == 3 ? P.tb3 : grp == 4 ? P.tb4 : grp == 5 ? P.tb5 : grp == 6 ? P.tb6 : P.tb7;
    if (threadIdx.x == 0) s_last = (atomicAdd(&P.ticket[grp * 32], 1u) == tb + gsize * (trip + 1) - 1);
    FVH_PT_MAX(trip, 2);
    __syncthreads();
    if (s_last) {
      reduce_group_rows(grow0 + grp);
      FVH_PT_MAX(trip, 5);
      if (threadIdx.x == 0) atomicAdd(&P.ticket[TICKET_GROUPS * 32], 1u);
      __syncthreads();
    }
    // The opener is always workgroup 0 (not whoever arrives last): the LM step is ~20 KB of code that runs once per
    // trip -- on a random CU it is fetched cold every time; on a fixed CU it stays in the instruction cache, and the LM
    // state stays in this workgroup's LDS for the whole launch instead of travelling through memory each trip.
    const bool opener = (blockIdx.x == 0);
    if (opener) {
      if (threadIdx.x == 0) {  // wait for the last group (its own arrival included)
        const unsigned want = P.tb_top + ngroups * (trip + 1);
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(&P.ticket[TICKET_GROUPS * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (wall_clock64() - t0 > P.watchdog_ticks) { ok = 0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        s_last = ok;
      }
      __syncthreads();
      if (!s_last) {  // not every workgroup is resident / something is stuck: never hang the GPU -- poison every tag and leave
        for (int idx = threadIdx.x; idx < PERSIST_REPLICAS * BCAST_SLOTS / 8; idx += 256) __hip_atomic_store(&P.bcast[idx * 8 + 7], abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) __hip_atomic_store(&st->aborted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      FVH_PT_MAX(trip, 6);
      reduce_final(grow0);
      __syncthreads();
      FVH_PT_MAX(trip, 7);
      if (trip == 0 && threadIdx.x == 0) {
        init_state();
        s_st.vm_num_voxels = P.vm_counters[0];
        s_st.vm_dropped = P.vm_counters[1] + (P.vm_counters2 ? P.vm_counters2[1] : 0);
      }
      if (threadIdx.x >= 64 && threadIdx.x < 64 + PART_STRIDE) s_st.sums[threadIdx.x - 64] = red[0][threadIdx.x - 64];  // (a lane each, not 32 round trips of lane 0)
      __syncthreads();
      FVH_PT_MAX(trip, 8);
      if (threadIdx.x == 0) dev_lm_step(&s_st, red[0]);
      FVH_PT_MAX(trip, 9);
      __syncthreads();
      if (threadIdx.x < 26) {  // payload: phase, correspondence buffer, x_lin, evaluation pose of the next trip
        const int ph = s_st.phase;
        const PoseD& pe = (ph == PH_LINEARIZE) ? s_st.x0 : s_st.xi;
        const int d = threadIdx.x;
        double v;
        if (d == 0) v = (double)ph;
        else if (d == 1) v = (double)s_st.corr_cur;
        else if (d < 11) v = s_st.x_lin.r[d - 2];
        else if (d < 14) v = s_st.x_lin.t[d - 11];
        else if (d < 23) v = pe.r[d - 14];
        else v = pe.t[d - 23];
        bc[d] = v;
      }
      __syncthreads();
      FVH_PT_MAX(trip, 10);
      for (int idx = threadIdx.x; idx < PERSIST_REPLICAS * BCAST_SLOTS; idx += 256) {
        const int slot = idx % BCAST_SLOTS, seg = slot >> 3, k = slot & 7, d = seg * 7 + k;
        const double val = (k == 7) ? want_tag : (d < 26 ? bc[d] : 0.0);
        __hip_atomic_store(&P.bcast[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      FVH_PT_MAX(trip, 3);
      if (s_st.phase == PH_DONE) {  // the state leaves the LDS once, at the end (the kernel boundary makes it visible)
        for (int i = threadIdx.x; i < ST_WORDS; i += 256) st_words[i] = reinterpret_cast<const unsigned long long*>(&s_st)[i];
        if (P.result_host) {
          // ... and goes straight to the host through mapped pinned memory: the caller spins on the sequence word instead of
          // paying a device-to-host copy kernel and a stream synchronisation (~8 us of a 300 us registration)
          for (int i = threadIdx.x; i < ST_WORDS; i += 256)
            __hip_atomic_store(&P.result_host[i], reinterpret_cast<const unsigned long long*>(&s_st)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (threadIdx.x == 0) __hip_atomic_store(&P.result_host[ST_WORDS + 1], P.launch_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (!opener) {
      if (threadIdx.x < 64) {  // wave 0 polls this workgroup's copy
        const double* rep = P.bcast + (size_t)(blockIdx.x % PERSIST_REPLICAS) * BCAST_SLOTS;
        const int lane = threadIdx.x;
        const bool is_slot = lane < BCAST_SLOTS, is_tag = is_slot && ((lane & 7) == 7);
        const unsigned long long t0 = wall_clock64();
        int ok = 0;
        for (;;) {
          const double v = is_slot ? __hip_atomic_load(&rep[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
          if (__any(is_tag && v == abort_tag)) break;  // another workgroup's watchdog aborted THIS launch
          if (__all(!is_tag || v == want_tag)) {
            const int d = (lane >> 3) * 7 + (lane & 7);
            if (is_slot && !is_tag && d < 26) bc[d] = v;
            ok = 1;
            break;
          }
          if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > P.watchdog_ticks))) {
            // not every workgroup is resident / something is stuck: never hang the GPU -- poison every tag and leave
            for (int idx = lane; idx < PERSIST_REPLICAS * BCAST_SLOTS / 8; idx += 64) __hip_atomic_store(&P.bcast[idx * 8 + 7], abort_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store(&st->aborted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) s_last = ok;
      }
      __syncthreads();
      if (!s_last) return;
    }
    FVH_PT_MAX(trip, 4);
    gen++;
    // LDS loads land in VGPRs; these values are wave-uniform, so move them to SGPRs (two PoseD in VGPRs cost the
    // kernel its second wave per SIMD, i.e. half of the co-resident workgroups the barrier needs)
    phase = (int)uniform_f64(bc[0]);
    if (phase == PH_DONE) return;
    corr_sel = (int)uniform_f64(bc[1]);
    // payload 2..13 = x_lin (r[9], t[3]), 14..25 = evaluation pose -> s_pose[0], s_pose[1] (every reader of the previous
    // poses is past the barrier above)
    if (threadIdx.x < 24) {
      const int which = threadIdx.x / 12, k = threadIdx.x % 12;
      const Real v = (Real)bc[2 + threadIdx.x];
      if (k < 9) s_pose[which].r[k] = v; else s_pose[which].t[k - 9] = v;
    }
    __syncthreads();  // bc[] is rewritten after the next barrier; s_pose is read by the next trip
  }
  }  // trips
}

}  // namespace fvh
