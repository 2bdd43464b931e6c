// The GN/LM hot loop for gfx950: ONE kernel for the whole LsqRegistration::computeTransformation loop (PERSIST = true: one
// launch per align(), one "trip" per LM transition), or one launch per linearize() / compute_error() (PERSIST = false: the
// host-driven calls of the C ABI and the fallback route of align()).
//
// Replaces SURVEY 2.2 K18-K23 + X2: find_voxel_correspondences (N_off launches + remove_if,
// find_voxel_correspondences.cu:16-111), compute_derivatives (transform_reduce of 43-float tuples,
// compute_derivatives.cu:18-184), the NDT variants (ndt_compute_derivatives.cu:33-231), the three tiny H2D copies and the
// blocking D2H per evaluation, and the host side of LsqRegistration::step_lm (lsq_registration_impl.hpp:123-168).
//
// Work item = (source element i, offset group g). Adjacent lanes share the source element (broadcast loads). Per item:
// transform, voxel coordinate (fp64 like the CPU reference or fp32 like the CUDA one), probe the dense key table, Mahalanobis
// M = (C_B + R C_A R^T)^-1 with R of the LINEARISATION pose (fast_vgicp_impl.hpp:101-115 caches it; compute_derivatives.cu:71-72),
// residual; per hit only S += w M, g += w M e, err are accumulated -- b = J^T g and H = J^T S J follow once per item (J depends
// on the element alone). 3x3 / 6x6 math stays in VGPRs (no MFMA).
//
// Reduction and hand-offs. Per item the wave reduces its 64 items' 29 sums with a transposing butterfly (v_permlane32/16_swap)
// into ONE fp64 accumulator per lane; after the main loop 4 waves x 32 slots meet in 1 KB of LDS -> one row of 32 sums per
// workgroup. Workgroup b belongs to group b % ng (ng = 1 for small grids, else 8: the group IS the workgroup's XCD).
//   * PERSIST = false: rows are written through (sc1), the last arriver of a group (atomic ticket) adds the group's rows in a fixed
//     order into a group row, the last group adds the <= 8 group rows in group order and runs the LM step; the state goes to HBM.
//   * PERSIST = true: nothing takes a ticket and nothing waits for a store. Everything that crosses workgroups travels as
//     {value, tag} PAIRS (one aligned 16-byte store per lane; tag = (launch sequence, trip): a pair is its own arrival signal,
//     nothing is ever cleared). The first workgroup of each group (the COLLECTOR) polls the pairs of its group's rows, adds
//     them in the ticket route's order and publishes a tagged group row; EVERY collector then polls all group rows, adds them in
//     group order and runs the LM step redundantly on its own LDS copy of the state (identical inputs, identical arithmetic:
//     bit-identical on all eight) and broadcasts the 26 values the next trip needs (phase, correspondence buffer, two poses) to
//     the workgroups of ITS OWN group only. Of the three hand-offs per trip only the group rows cross XCDs: rows and broadcast
//     stay inside one XCD, where a plain store is visible to an L1-bypassing (sc1) load through the XCD's own L2 in ~0.2 us
//     instead of a ~1 us memory-side round trip (`xcd_local`; every workgroup checks HW_REG_XCC_ID against the XCD its group
//     stands for before it relies on that). Both routes add the same numbers in the same order: bit-identical results.
//     (Round 4 could also confine a launch to a subset of the XCDs; measured no better for small grids and worse for concurrent aligns,
//     profiles/r04_small_grid_layouts.txt / r04_concurrency.txt: removed.)
#pragma once
#include <cstddef>
#include <type_traits>

#include "dev_math.hpp"
#include "kernels_peer.hpp"
#include "kernels_voxelmap.hpp"

// Contraction only inside one source expression (not across statements, which is hipcc's default "fast"): whether a
// multiply and an add fuse must not depend on what the optimiser happens to see around them -- the persistent and the
// per-transition instantiations of the cost kernel have to produce bit-identical sums.
#pragma clang fp contract(on)

namespace fvh {

constexpr int NSUM = 28;       // err(1) b(6) Hrr(6) Hrt(9) Htt(6)
constexpr int PART_STRIDE = 32;
constexpr int MAX_PARTIAL_ROWS = 1024;  // workgroup rows (4 workgroups per CU x 256 CUs); the group rows follow

enum CostMode { MODE_VGICP = 0, MODE_NDT_P2D = 1, MODE_NDT_D2D = 2 };
enum Phase { PH_LINEARIZE = 0, PH_TRIAL = 1, PH_DONE = 2, PH_FIND_ONLY = 3, PH_EVAL_DERIV = 4, PH_EVAL_ERROR = 5,
             // a trial whose speculative linearisation can never be used -- the proposed step is already below the convergence
             // thresholds, or an accepted step ends the outer loop: the trip evaluates the trial error of the stored ids only
             PH_TRIAL_FINAL = 6 };
// PH_TRIAL in device-LM mode is the FUSED launch: trial error with the old correspondences + speculative linearisation at the trial pose

struct LmState {
  PoseD x0;        // current estimate == linearisation pose
  PoseD xi;        // trial pose
  PoseD x_lin;     // pose at which the CURRENT correspondence buffer was computed (reference: linearized_x)
  double H[36], b[6], d[6];
  double y0, lambda, nu;
  double final_H[36];
  double sums[PART_STRIDE];  // reduced {err, b, H...} of the last evaluation (all-reduced in multi-GPU mode)
  // parameters
  double rotation_epsilon, transformation_epsilon, lm_init_lambda_factor;
  int max_iterations, lm_max_iterations;
  int optimizer, pad_;  // 0 Levenberg-Marquardt (step_lm), 1 Gauss-Newton (step_gn, lsq_registration_impl.hpp:108-121: every transition is a linearisation, the step is always taken)
  // status
  int phase, outer_iter, inner_iter, converged, lm_failed, num_linearize, num_error_evals, nr_iterations;
  int corr_cur;       // which of the two correspondence buffers is current (device-LM mode flips it on accept)
  int vm_num_voxels;  // copied from the voxel map's counters by the last workgroup: capacity hint for the next build
  int vm_dropped;     // > 0: the hint-sized table overflowed -> host rebuilds at the safe size and re-runs
  int vm_num_voxels2; // the same for the source voxel map (NDT D2D): shapes the grid of the next align of a frame stream
  int delta_converged;  // is_converged(delta) of the step proposed last (computed when the step is proposed, consumed by the next trial)
  int halo_exceeded;    // sharded target map (VmRegion): some evaluation of this align had a source element outside the shard's inner box (summed over all ranks)
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
  const float4* src_cov_sorted;  // with `src_sorted`, optional: the covariances in Morton order (element order[w]: entries 2 w, 2 w + 1)
  const float4* src_sorted;   // with `order`, optional: the Morton-ordered copy of src_pts (.w = original index): the point of element order[w] is src_sorted[w] -- a coalesced load
  int n_src;
  const uint4* table;                // voxel records (64 B per bucket)
  const unsigned long long* keys;    // voxel keys of the same buckets, dense (kernels_voxelmap.hpp)
  unsigned mask;
  const unsigned long long* bitmap;  // occupancy bitmap of a large map (kernels_voxelmap.hpp: VmGrid) or null: misses answered without touching the key table
  const VmGrid* grid;
  const VmRegion* region;            // the target map is a rank's shard (kernels_voxelmap.hpp: VmRegion) or null: source elements that leave its inner box raise `exceeded`
  double res, inv_res;        // voxel resolution and its correctly rounded reciprocal (host)
  const int* offsets;         // n_off x 3
  const int* offsets_packed;  // n_off x (dx + 512) | (dy + 512) << 10 | (dz + 512) << 20
  int n_off;
  int group;                  // offsets per work item
  int groups_per_src;         // ceil(n_off / group)
  unsigned gps_magic;         // ceil(2^32 / groups_per_src) when item / groups_per_src == mulhi(item, magic) for every item of this launch, else 0
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
  double* bcast;              // persistent kernel: [PERSIST_REPLICAS][BCAST_PAIRS] {value, tag} pairs
  unsigned long long launch_tag;  // persistent kernel: sequence number of this launch (tags of older launches never match)
  unsigned long long* result_host;  // persistent kernel: mapped pinned host memory, [sizeof(LmState)/8 words of state][sequence word] (null: not used)
  unsigned long long watchdog_ticks;  // persistent kernel: 100 MHz ticks a workgroup may wait at the barrier before it aborts the launch
  int max_iterations, lm_max_iterations;
  int optimizer;       // fvh_lm_params::optimizer (first launch of an align)
  double rotation_epsilon, transformation_epsilon, lm_init_lambda_factor;
  // multi-GPU (kernels_peer.hpp): this rank walks the source elements [item_lo, item_hi) of the (Morton) order -- its spatial
  // tile -- and the reduced sums are exchanged with the peers inside the kernel; peer.n <= 1: single GPU, whole cloud
  int item_lo, item_hi;
  // ... the same for a source whose element count lives on the device (NDT D2D: the voxels of the source map, walked in the canonical --
  // key-sorted -- order `order` so that every rank cuts the same list): this rank takes chunk tile_rank of tile_n equal chunks (tile_n <= 1: everything)
  int tile_rank, tile_n;
  // NDT device LM on small grids: a workgroup takes 128 items, waves 0-1 evaluate the trial error of the stored ids, waves 2-3 find and
  // linearise the new ones (see the main loop); 0: every wave does both for its own 64 items
  int split;
  double* lm_trace;   // setDebugPrint on the device LM: 6 doubles per trial {i, y0, yi, rho, lambda, |d|} (lsq_registration_impl.hpp:143-149), or null
  int corr_by_position;  // the correspondence rows are indexed by the element's POSITION in the walk order (Morton), not by its original index
  int external_find;  // FastGICP on the device: the correspondences of every linearisation were found by nn1_corr_kernel right before this launch (nothing to probe here)
  PeerView peer;
  unsigned long long peer_watchdog_ticks;
  // grid layout (both routes take the same (nb, ng): the same partition of the items and the same order of the sums)
  int ng;              // reduction groups: workgroup b belongs to group b % ng (1: single level; 8: one group per XCD)
  int prio_mode;       // wave priority in the main loop (FVH_COST_PRIO). 0: none; 1: s_setprio 1 for the SECOND workgroup of a CU (lb >= prio_from).
                       // (Round 4 also measured: the first half of the loop only, the FIRST workgroup instead, every workgroup, graded by age --
                       // all worse or equal, profiles/r04_priority_ab.txt; they are no longer in the kernel.)
  int prio_from;       // ... "second" = workgroups from this one on (the host passes the number of CUs)
  int xcd_local;       // persistent kernel: every member of a group sits on the group's XCD (checked per workgroup): rows and broadcast
                       // travel through that XCD's L2 (plain stores) instead of write-through + memory-side polls
  int lm_everywhere;   // persistent kernel, two levels: EVERY workgroup polls the group rows and runs the LM step on its own copy of the
                       // state -- no broadcast hand-off (small grids: one workgroup per CU, the step's code stays in its instruction cache)
};

// ------------------------------------------------------------------------------------------------
// LM step (lsq_registration_impl.hpp:82-91,123-168; so3.hpp:58-104)
// ------------------------------------------------------------------------------------------------
// This code runs on ONE wave between two evaluations of the cost while every other SIMD of the chip waits for its result:
// its dependent-instruction chain is latency on the critical path of every LM transition. Round 2's version (6x6 one element
// per lane, pivots through lane shuffles, IEEE divisions, libm sincos) was a chain of ~6,500 cycles = 2.8 us of every ~17 us
// trip: ten fp64 divisions (~300 dependent cycles each: v_div_scale / v_rcp / Newton / v_div_fmas / v_div_fixup) and twelve
// LDS-crossbar shuffles sat on it. Here every lane runs the WHOLE step redundantly in registers (wave-uniform values, no
// shuffles), reciprocals are a hardware seed + two Newton steps (fast_rcp: ~1 ulp, operands are pivots / norms far from the
// exponent limits; the exact-zero pivot keeps Eigen's pseudo-inverse semantics), and the half-angle sine / cosine of a
// step below one radian come from their Taylor polynomials.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
// sin(x), cos(x) for |x| <= 0.5: Taylor to x^17 / x^18 (truncation < 4e-23 relative), Horner in x^2
__device__ __forceinline__ void sincos_small(double x, double* sn, double* cs) {
  const double z = x * x;
  double ps = 1.0 / 355687428096000.0;
  ps = fma(ps, z, -1.0 / 1307674368000.0);
  ps = fma(ps, z, 1.0 / 6227020800.0);
  ps = fma(ps, z, -1.0 / 39916800.0);
  ps = fma(ps, z, 1.0 / 362880.0);
  ps = fma(ps, z, -1.0 / 5040.0);
  ps = fma(ps, z, 1.0 / 120.0);
  ps = fma(ps, z, -1.0 / 6.0);
  *sn = fma(x * z, ps, x);
  double pc = -1.0 / 6402373705728000.0;
  pc = fma(pc, z, 1.0 / 20922789888000.0);
  pc = fma(pc, z, -1.0 / 87178291200.0);
  pc = fma(pc, z, 1.0 / 479001600.0);
  pc = fma(pc, z, -1.0 / 3628800.0);
  pc = fma(pc, z, 1.0 / 40320.0);
  pc = fma(pc, z, -1.0 / 720.0);
  pc = fma(pc, z, 1.0 / 24.0);
  *cs = fma(z * z, pc, fma(z, -0.5, 1.0));
}
__device__ inline void dev_se3_exp(const double a[6], PoseD& T) {
  // One sincos of the half angle; the full-angle terms of the V matrix follow from the double-angle identities
  //   1 - cos(t) = 2 sin^2(t/2),   sin(t) = 2 sin(t/2) cos(t/2)
  // (so3.hpp:58-104 calls sin/cos four times; oracle probe ORC_LM_ARITH_VARIANT=2: converged poses agree to 5e-15).
  const double ox = a[0], oy = a[1], oz = a[2];
  const double theta_sq = ox * ox + oy * oy + oz * oz;
  const double theta = sqrt(theta_sq);
  double sh = 0.0, ch = 1.0;
  if (theta >= 1e-10) {  // needed by the V matrix even when the quaternion takes its Taylor branch
    if (__builtin_amdgcn_ballot_w64(theta > 1.0) == 0ull) sincos_small(0.5 * theta, &sh, &ch); else sincos(0.5 * theta, &sh, &ch);  // (wave-uniform callers: a scalar branch)
  }
  const double inv_theta = theta >= 1e-10 ? fast_rcp(theta) : 0.0;
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    imag = sh * inv_theta;
    real = ch;
  }
  const double qw = real, qx = imag * ox, qy = imag * oy, qz = imag * oz;
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  T.r[0] = 1 - (tyy + tzz); T.r[1] = txy - twz;       T.r[2] = txz + twy;
  T.r[3] = txy + twz;       T.r[4] = 1 - (txx + tzz); T.r[5] = tyz - twx;
  T.r[6] = txz - twy;       T.r[7] = tyz + twx;       T.r[8] = 1 - (txx + tyy);
  // t = V v,  V = I + A Omega + B Omega^2 (so3.hpp:91-101; V = R below 1e-10): Omega v = omega x v and
  // Omega^2 v = omega (omega . v) - theta^2 v, so the 3x3 products are never formed
  const double vx = a[3], vy = a[4], vz = a[5];
  if (__builtin_amdgcn_ballot_w64(theta < 1e-10) != 0ull) {
    T.t[0] = T.r[0] * vx + T.r[1] * vy + T.r[2] * vz;
    T.t[1] = T.r[3] * vx + T.r[4] * vy + T.r[5] * vz;
    T.t[2] = T.r[6] * vx + T.r[7] * vy + T.r[8] * vz;
  } else {
    const double inv_tsq = inv_theta * inv_theta;
    const double A = 2.0 * sh * sh * inv_tsq, B = (theta - 2.0 * sh * ch) * inv_tsq * inv_theta;
    const double cx = oy * vz - oz * vy, cy = oz * vx - ox * vz, cz = ox * vy - oy * vx;
    const double ov = ox * vx + oy * vy + oz * vz;
    T.t[0] = vx + A * cx + B * (ox * ov - theta_sq * vx);
    T.t[1] = vy + A * cy + B * (oy * ov - theta_sq * vy);
    T.t[2] = vz + A * cz + B * (oz * ov - theta_sq * vz);
  }
}

// lsq_registration_impl.hpp:82-91: (|R - I|.max / rot_eps, |t|.max / trans_eps).max < 1. For positive thresholds x / eps < 1 is
// exactly x < eps in IEEE arithmetic (a quotient of x < eps never rounds up to 1), so the two divisions are only formed for the
// nonsensical thresholds (zero, negative, NaN) where the comparison form would answer differently.
__device__ __forceinline__ bool dev_is_converged(double rot_eps, double trans_eps, const PoseD& delta) {
  if (__builtin_amdgcn_ballot_w64(rot_eps > 0 && trans_eps > 0) != 0ull) {  // max_i x_i < eps  <=>  every x_i < eps: twelve compares, no max chain
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 9; i++) ok &= (fabs(delta.r[i] - ((i % 4 == 0) ? 1.0 : 0.0)) < rot_eps);  // (&=, not &&: no short-circuit branches)
#pragma unroll
    for (int i = 0; i < 3; i++) ok &= (fabs(delta.t[i]) < trans_eps);
    return ok;
  }
  double rmax = 0, tmax = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) rmax = fmax(rmax, fabs(delta.r[i] - ((i % 4 == 0) ? 1.0 : 0.0)));
#pragma unroll
  for (int i = 0; i < 3; i++) tmax = fmax(tmax, fabs(delta.t[i]));
  return fmax(rmax / rot_eps, tmax / trans_eps) < 1;
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
// where element (i, j) of H sits in the sums (the layout unpack_sums spells out)
__device__ __host__ constexpr int sums_index_of_H(int i, int j) {
  const int a = i < 3 ? i : i - 3, c = j < 3 ? j : j - 3;
  const int lo = a < c ? a : c, hi = a < c ? c : a;
  const int sym = lo * 3 - (lo * (lo - 1)) / 2 + (hi - lo);
  return (i < 3 && j < 3) ? 7 + sym : (i >= 3 && j >= 3) ? 22 + sym : (i < 3) ? 13 + i * 3 + (j - 3) : 13 + j * 3 + (i - 3);
}

// ------------------------------------------------------------------------------------------------
// One transition of the {linearize -> trial* -> accept} machine; exactly the control flow of LsqRegistration::
// computeTransformation + step_lm. sums[0..27] = {err, b, H} of the linearisation this evaluation computed (at x0 for
// PH_LINEARIZE, speculatively at xi for the fused PH_TRIAL evaluation), sums[28] = trial error y_i at xi with the OLD
// correspondences (fused evaluation only; PH_TRIAL_FINAL: y_i at sums[0]).
// Executed by ONE FULL WAVE, every lane running the same instruction stream on the same (wave-uniform) values: the state is
// read from LDS once at the top, the 6x6 LDL^T, the triangular solves, se3_exp and the tests live in registers, lanes 0..35 /
// 0..11 / 0 write the results back. `st` and `sums` must be LDS.
// ------------------------------------------------------------------------------------------------
#ifdef FVH_LM_STEP_NOINLINE  // A/B switch: a real function (called once per trip through generic pointers) instead of inlined code on LDS
#define FVH_LM_STEP_ATTR __noinline__
#else
#define FVH_LM_STEP_ATTR __forceinline__
#endif
// Every value below is wave-uniform, but the compiler cannot know (the state comes out of LDS, i.e. out of VGPRs): written
// naively every `if` becomes exec-mask bookkeeping (s_and_saveexec / s_cbranch_execz, ~70 of them) and every merge a
// v_cndmask. A condition that went through a ballot IS uniform to the compiler: plain scalar branches, no selects.
#define FVH_UNI(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#ifdef FVH_COST_TIMING  // shader-cycle stamps inside the LM step (tools/persist_timing.py): [call][stage]; intrusive (every stamp drains the
__device__ unsigned long long g_lmtime[16][8];  // memory pipeline, the previous stamp's store included): only with -DFVH_LM_STAGES
__device__ unsigned g_lmcall;
#endif
#if defined(FVH_COST_TIMING) && defined(FVH_LM_STAGES)
#define FVH_LM_T(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0 && lm_call < 16) g_lmtime[lm_call][k] = __builtin_readcyclecounter(); } while (0)
#else
#define FVH_LM_T(k) do { } while (0)
#endif
// GN (compile time): LsqRegistration::step_gn instead of step_lm. A template parameter, not a branch on the state: the Levenberg-Marquardt
// instantiations of the LM kernel sit on a register cliff and must not pay for the other optimiser's code.
template <bool GN = false>
__device__ FVH_LM_STEP_ATTR void dev_lm_step_wave(LmState* st, const double* sums, const int lane, double* trace = nullptr) {
  const bool in36 = lane < 36, in12 = lane < 12, in6 = lane < 6;
  double* x0p = reinterpret_cast<double*>(&st->x0);
  double* xip = reinterpret_cast<double*>(&st->xi);
  double* xlp = reinterpret_cast<double*>(&st->x_lin);
#if defined(FVH_COST_TIMING) && defined(FVH_LM_STAGES)
  const unsigned lm_call = __builtin_amdgcn_readfirstlane(g_lmcall);
  if (lane == 0) g_lmcall = lm_call + 1;
#endif
  FVH_LM_T(0);
  // ---- everything the decision needs, requested together (one LDS round trip) ----
  const int phase0 = __builtin_amdgcn_readfirstlane(st->phase);
  int phase = phase0;
  double lambda = st->lambda, nu = st->nu, y0 = st->y0;
  int outer_iter = __builtin_amdgcn_readfirstlane(st->outer_iter), inner_iter = __builtin_amdgcn_readfirstlane(st->inner_iter);
  int converged = __builtin_amdgcn_readfirstlane(st->converged), lm_failed = __builtin_amdgcn_readfirstlane(st->lm_failed);
  int num_linearize = __builtin_amdgcn_readfirstlane(st->num_linearize), num_error_evals = __builtin_amdgcn_readfirstlane(st->num_error_evals);
  int nr_iterations = __builtin_amdgcn_readfirstlane(st->nr_iterations), corr_cur = __builtin_amdgcn_readfirstlane(st->corr_cur);
  const int max_iterations = __builtin_amdgcn_readfirstlane(st->max_iterations), lm_max_iterations = __builtin_amdgcn_readfirstlane(st->lm_max_iterations);
  constexpr bool gauss_newton = GN;  // step_gn: H d = -b undamped, x0 = exp(d) x0 at once (no trial evaluation)
  const double rot_eps = st->rotation_epsilon, trans_eps = st->transformation_epsilon, lambda_factor = st->lm_init_lambda_factor;
  double dprev[6], bprev[6];
#pragma unroll
  for (int j = 0; j < 6; j++) { dprev[j] = st->d[j]; bprev[j] = st->b[j]; }
  const int delta_conv_prev = __builtin_amdgcn_readfirstlane(st->delta_converged);

  const double s0 = sums[0], s28 = sums[28];
  if (lane == 0 && sums[30] > 0.0) st->halo_exceeded = 1;  // (sharded target map: a source element left some rank's shard in this evaluation)

  bool accepted = false, consume = false, done = false;
  if (phase0 == PH_LINEARIZE) {
    consume = true;
  } else {  // PH_TRIAL (fused): trial error of this evaluation at sums[28]; PH_TRIAL_FINAL (error only): at sums[0]
    const double yi = phase0 == PH_TRIAL_FINAL ? s0 : s28;
    num_error_evals++;
    double denom = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) denom += dprev[j] * (lambda * dprev[j] - bprev[j]);
    // rho = (y0 - yi) / denom: a degenerate denominator (zero / subnormal: d = 0, no correspondences) keeps the IEEE quotient
    double rho;
    if (FVH_UNI(fabs(denom) > 1e-290)) rho = (y0 - yi) * fast_rcp(denom); else rho = (y0 - yi) / denom;
    if (trace) {  // the line LsqRegistration prints per trial when setDebugPrint(true) (lsq_registration_impl.hpp:143-149)
      double dn = 0;
#pragma unroll
      for (int j = 0; j < 6; j++) dn += dprev[j] * dprev[j];
      if (lane == 0) {
        double* row = trace + 6 * (size_t)(num_error_evals - 1);
        row[0] = (double)inner_iter; row[1] = y0; row[2] = yi; row[3] = rho; row[4] = lambda; row[5] = sqrt(dn);
      }
    }
    const bool conv = delta_conv_prev != 0;  // is_converged(delta) of the step whose trial this is (lsq_registration_impl.hpp:151,160)
    if (FVH_UNI(rho < 0)) {
      if (conv) {  // step_lm returns true with x0 unchanged -> converged_ = true
        converged = 1; outer_iter++; phase = PH_DONE; done = true;
      } else {
        lambda = nu * lambda;
        nu = 2 * nu;
        inner_iter++;
        if (inner_iter >= lm_max_iterations) { lm_failed = 1; phase = PH_DONE; done = true; }  // "lm not converged!!"
        // otherwise: a new trial from the SAME (H, b); the speculative linearisation of this evaluation is discarded
      }
    } else {  // accepted: x0 = xi, final_H = H (both done in the tail below)
      accepted = true;
      { const double u = 2 * rho - 1; lambda = lambda * fmax(1.0 / 3.0, 1 - u * u * u); }
      converged = conv ? 1 : 0;
      outer_iter++;
      if (converged || outer_iter >= max_iterations) {
        phase = PH_DONE; done = true;
      } else {  // the speculative linearisation at xi (== the new x0) is exactly the next step_lm's linearize()
        corr_cur ^= 1;
        consume = true;
      }
    }
  }
  FVH_LM_T(1);
  // (H, b) in registers on every lane: lower triangle row by row -- Hl[i (i + 1) / 2 + j], j <= i
  double Hl[21], bv[6];
  if (!done) {
    if (consume) {  // dev_lm_consume_linearization: y0, H, b of the evaluation just reduced
      y0 = s0;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        bv[i] = sums[1 + i];
#pragma unroll
        for (int j = 0; j <= i; j++) Hl[i * (i + 1) / 2 + j] = sums[sums_index_of_H(i, j)];
      }
      num_linearize++;
      nr_iterations = outer_iter;
      if (FVH_UNI(lambda < 0.0)) {
        double mx = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(Hl[i * (i + 1) / 2 + i]));
        lambda = lambda_factor * mx;
      }
      nu = 2.0;
      inner_iter = 0;
      if (phase0 == PH_LINEARIZE && !gauss_newton) {
        if (lm_max_iterations <= 0) { lm_failed = 1; phase = PH_DONE; done = true; } else phase = PH_TRIAL;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; i++) {
        bv[i] = bprev[i];
#pragma unroll
        for (int j = 0; j <= i; j++) Hl[i * (i + 1) / 2 + j] = st->H[i * 6 + j];
      }
    }
  }
  if (lane == 0) { st->lambda = lambda; st->nu = nu; st->y0 = y0; }  // final from here on (stored now: not held in registers across the solve)
  FVH_LM_T(2);
  PoseD delta;
  double d[6];
  bool gn_step = false;
  int delta_conv = delta_conv_prev;
  if (!done) {
    // ---- d = (H + lambda I)^-1 (-b): LDL^T in registers (Eigen::LDLT semantics for a vanishing pivot: the column stays
    // unscaled and the solve uses the pseudo-inverse of D -- with no correspondences at all H = 0, lambda = 0, d = 0 and the
    // reference returns the initial guess flagged converged instead of a NaN pose, lsq_registration_impl.hpp:111-168) ----
    double L[15], D[6], Dinv[6];  // strictly lower part row by row: L[i (i - 1) / 2 + j], j < i
    const double lam_solve = gauss_newton ? 0.0 : lambda;  // (H + 0 is H: Gauss-Newton factorises H itself, lsq_registration_impl.hpp:111)
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double dj = Hl[j * (j + 1) / 2 + j] + lam_solve;
#pragma unroll
      for (int k = 0; k < j; k++) dj -= L[j * (j - 1) / 2 + k] * L[j * (j - 1) / 2 + k] * D[k];
      D[j] = dj;
      const double r = fast_rcp(dj);
      const bool pivot_ok = fabs(dj) > 2.2250738585072014e-308;
      Dinv[j] = pivot_ok ? r : 0.0;
      const double scale = pivot_ok ? r : 1.0;
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        double sacc = Hl[i * (i + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; k++) sacc -= L[i * (i - 1) / 2 + k] * L[j * (j - 1) / 2 + k] * D[k];
        L[i * (i - 1) / 2 + j] = sacc * scale;
      }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double sacc = -bv[i];
#pragma unroll
      for (int k = 0; k < i; k++) sacc -= L[i * (i - 1) / 2 + k] * y[k];
      y[i] = sacc;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= Dinv[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      double sacc = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) sacc -= L[k * (k - 1) / 2 + i] * d[k];
      d[i] = sacc;
    }
    FVH_LM_T(3);
    dev_se3_exp(d, delta);
    FVH_LM_T(4);
    // the trial of this proposal is the last evaluation of the align if the step is already converged (then accepted or not, the
    // loop ends: lsq_registration_impl.hpp:57-66,150-165) or if accepting it exhausts max_iterations -- its speculative
    // linearisation would be thrown away
    delta_conv = FVH_UNI(dev_is_converged(rot_eps, trans_eps, delta)) ? 1 : 0;
    if (gauss_newton) {  // the step is taken: converged_ = is_converged(delta), next outer iteration (lsq_registration_impl.hpp:57-66)
      gn_step = true;
      converged = delta_conv;
      outer_iter++;
      phase = (converged || outer_iter >= max_iterations) ? PH_DONE : PH_LINEARIZE;
    } else {
      phase = (delta_conv || outer_iter + 1 >= max_iterations) ? PH_TRIAL_FINAL : PH_TRIAL;
    }
  }
  FVH_LM_T(5);
  // ---- tail: the state in LDS. Everything comes out of registers (lane 0 stores the poses, the step and the scalars as 16-byte
  // writes; the 6x6 copies go one element per lane): no LDS read-modify-write chains between the stages ----
  double x0r[12], xir[12];  // the (new) current estimate: x0 = xi when the trial was accepted; read here, not at the top (24 doubles held across the
  {                         // solve spilled); one LDS round trip
    const double* src = accepted ? xip : x0p;
#pragma unroll
    for (int k = 0; k < 12; k++) x0r[k] = src[k];
  }
  if (accepted && in36) st->final_H[lane] = st->H[lane];  // final_hessian_ = H (before H is replaced below)
  if (consume) {
    // the state's full row-major copy of H (the next trials' source, final_H, the host's getFinalHessian): one element per lane
    const int li = lane / 6, lj = lane - li * 6;
    const int a = li < 3 ? li : li - 3, c = lj < 3 ? lj : lj - 3;
    const int lo = a < c ? a : c, hi = a < c ? c : a;
    const int sym = lo * 3 - (lo * (lo - 1)) / 2 + (hi - lo);
    int hidx;
    if (li < 3 && lj < 3) hidx = 7 + sym;
    else if (li >= 3 && lj >= 3) hidx = 22 + sym;
    else if (li < 3) hidx = 13 + li * 3 + (lj - 3);
    else hidx = 13 + lj * 3 + (li - 3);
    if (in36) st->H[lane] = sums[hidx];
    if (in6) st->b[lane] = sums[1 + lane];
    if (gn_step && in36) st->final_H[lane] = sums[hidx];  // step_gn: final_hessian_ = H
  }
  if (!done) {  // xi = delta * x0 (dev_pose_mul)
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) xir[i * 3 + j] = delta.r[i * 3 + 0] * x0r[0 * 3 + j] + delta.r[i * 3 + 1] * x0r[1 * 3 + j] + delta.r[i * 3 + 2] * x0r[2 * 3 + j];
      xir[9 + i] = delta.r[i * 3 + 0] * x0r[9] + delta.r[i * 3 + 1] * x0r[10] + delta.r[i * 3 + 2] * x0r[11] + delta.t[i];
    }
  }
  if (lane == 0) {
    if (accepted) {
#pragma unroll
      for (int k = 0; k < 12; k++) x0p[k] = x0r[k];
    }
    if (consume) {
#pragma unroll
      // x_lin = x0 (Gauss-Newton: the NEW x0, where the next trip linearises -- unless this step ends the loop: the stored correspondences then
      // stay those of the linearisation just consumed, at the OLD x0, which is what a later compute_error() must rotate the covariances by)
      for (int k = 0; k < 12; k++) xlp[k] = (gn_step && phase != PH_DONE) ? xir[k] : x0r[k];
    }
    if (gn_step) {
#pragma unroll
      for (int k = 0; k < 12; k++) x0p[k] = xir[k];  // x0 = delta * x0
    }
    if (!done) {
#pragma unroll
      for (int k = 0; k < 12; k++) xip[k] = xir[k];
#pragma unroll
      for (int j = 0; j < 6; j++) st->d[j] = d[j];
    }
    st->phase = phase;
    st->outer_iter = outer_iter; st->inner_iter = inner_iter; st->converged = converged; st->lm_failed = lm_failed;
    st->num_linearize = num_linearize; st->num_error_evals = num_error_evals; st->nr_iterations = nr_iterations; st->corr_cur = corr_cur;
    st->delta_converged = delta_conv;
  }
  FVH_LM_T(6);
}

// tiny kernels for the multi-GPU path and for (re)initialising the state
__global__ void lm_init_kernel(LmState* st, PoseD guess, double rot_eps, double trans_eps, double lambda_factor, int max_iter, int lm_max_iter, unsigned* ticket, int optimizer = 0) {
  if (blockIdx.x == 0 && threadIdx.x <= 8) ticket[threadIdx.x] = 0;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->x0 = guess; st->xi = guess;
  st->rotation_epsilon = rot_eps; st->transformation_epsilon = trans_eps; st->lm_init_lambda_factor = lambda_factor;
  st->max_iterations = max_iter; st->lm_max_iterations = lm_max_iter; st->optimizer = optimizer; st->pad_ = 0;
  st->lambda = -1.0; st->nu = 2.0; st->y0 = 0.0;
  st->phase = max_iter > 0 ? PH_LINEARIZE : PH_DONE;
  st->corr_cur = 0; st->x_lin = guess; st->delta_converged = 0; st->halo_exceeded = 0;
  st->outer_iter = 0; st->inner_iter = 0; st->converged = 0; st->lm_failed = 0; st->num_linearize = 0; st->num_error_evals = 0; st->nr_iterations = 0;
  for (int i = 0; i < 36; i++) st->final_H[i] = (i % 7 == 0) ? 1.0 : 0.0;
}
template <bool GN = false>
__global__ __launch_bounds__(64) void lm_update_kernel(LmState* st) {  // <<<1, 64>>>: one wave, state staged through LDS
  __shared__ LmState s;
  constexpr int WORDS = sizeof(LmState) / 8 - 1;  // without the barrier word
  unsigned long long* g = reinterpret_cast<unsigned long long*>(st);
  unsigned long long* l = reinterpret_cast<unsigned long long*>(&s);
  for (int i = threadIdx.x; i < WORDS; i += 64) l[i] = g[i];
  __syncthreads();
  if (s.phase == PH_DONE) return;
  dev_lm_step_wave<GN>(&s, s.sums, threadIdx.x);
  __syncthreads();
  for (int i = threadIdx.x; i < WORDS; i += 64) g[i] = l[i];
}

// ------------------------------------------------------------------------------------------------
// per-correspondence terms
// ------------------------------------------------------------------------------------------------
// All correspondences of ONE source element share its transformed point q, and the Jacobian J = [skew(q), -I] depends on q
// only (fast_vgicp_impl.hpp:152-160, compute_derivatives.cu:80-91). So per hit only
//     S += w M  (6),   g += w M e  (3),   err += w e^T M e
// is accumulated, and the 27 derivative sums of the element follow ONCE from  b = J^T g,  H = J^T S J  (item_sums below) --
// instead of ~100 fp64 operations per hit.
template <typename Real>
struct ItemAcc {
  Sym3<Real> S;
  Vec3<Real> g;
  Real err;
};
// One correspondence with combined covariance A = C_B + R C_A R^T and weight w: the Mahalanobis matrix w A^-1 is applied as
// (w / det A) adj(A) -- one division, and the adjugate is never scaled on its own.
template <typename Real>
__device__ __forceinline__ void hit_term(ItemAcc<Real>& it, const Vec3<Real>& q, const Vec3<Real>& mu, const Sym3<Real>& A, Real w, bool deriv) {
  const Vec3<Real> e = {mu.x - q.x, mu.y - q.y, mu.z - q.z};
  Real det;
  const Sym3<Real> C = adjugate(A, det);
  const Real s = fast_div(w, det);
  const Vec3<Real> Ce = mul(C, e);
  it.err += s * (e.x * Ce.x + e.y * Ce.y + e.z * Ce.z);
  if (!deriv) return;
  it.g.x += s * Ce.x; it.g.y += s * Ce.y; it.g.z += s * Ce.z;
  it.S.xx += s * C.xx; it.S.xy += s * C.xy; it.S.xz += s * C.xz; it.S.yy += s * C.yy; it.S.yz += s * C.yz; it.S.zz += s * C.zz;
}
// the weight sqrt(n) stored with a voxel record (vm_finalize_kernel / gicp_records_kernel): a double in q3.zw
template <typename Real>
__device__ __forceinline__ Real record_weight(const float4& q3) { return (Real)__hiloint2double(__float_as_int(q3.w), __float_as_int(q3.z)); }
template <typename Real>
__device__ __forceinline__ Real record_weight(const float2&) { return (Real)1; }  // (NDT records carry none: never called)
// v[0] = err, v[1..6] = b, v[7..12] = H_rr (xx xy xz yy yz zz), v[13..21] = H_rt (row-major), v[22..27] = H_tt
template <typename Real>
__device__ __forceinline__ void item_sums(double* v, const ItemAcc<Real>& it, const Vec3<Real>& q) {
  v[0] = (double)it.err;
  // b = J^T g = [g x q ; -g]
  const Vec3<Real> bq = cross(it.g, q);
  v[1] = (double)bq.x; v[2] = (double)bq.y; v[3] = (double)bq.z;
  v[4] = -(double)it.g.x; v[5] = -(double)it.g.y; v[6] = -(double)it.g.z;
  // P = skew(q) S (columns q x S_col) ; H = [[P skew(q)^T, P], [P^T, S]]
  const Sym3<Real>& S = it.S;
  const Vec3<Real> c0 = {S.xx, S.xy, S.xz}, c1 = {S.xy, S.yy, S.yz}, c2 = {S.xz, S.yz, S.zz};
  const Vec3<Real> p0 = cross(q, c0), p1 = cross(q, c1), p2 = cross(q, c2);  // P_ij = p_j[i]
  const Vec3<Real> r0 = {p0.x, p1.x, p2.x}, r1 = {p0.y, p1.y, p2.y}, r2 = {p0.z, p1.z, p2.z};  // rows of P
  const Vec3<Real> h0 = cross(q, r0), h1 = cross(q, r1), h2 = cross(q, r2);  // rows of H_rr
  v[7] = (double)h0.x; v[8] = (double)h0.y; v[9] = (double)h0.z; v[10] = (double)h1.y; v[11] = (double)h1.z; v[12] = (double)h2.z;
  v[13] = (double)r0.x; v[14] = (double)r0.y; v[15] = (double)r0.z;
  v[16] = (double)r1.x; v[17] = (double)r1.y; v[18] = (double)r1.z;
  v[19] = (double)r2.x; v[20] = (double)r2.y; v[21] = (double)r2.z;
  v[22] = (double)S.xx; v[23] = (double)S.xy; v[24] = (double)S.xz; v[25] = (double)S.yy; v[26] = (double)S.yz; v[27] = (double)S.zz;
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
#ifndef FVH_PERSIST_REPLICAS
#define FVH_PERSIST_REPLICAS 32
#endif
constexpr int PERSIST_REPLICAS = FVH_PERSIST_REPLICAS;  // copies of the broadcast row (even); workgroup b polls copy b % PERSIST_REPLICAS
constexpr int BCAST_PAIRS = 32;                // {value, tag} pairs per copy (26 used)
constexpr int BCAST_VALUES = 26;               // phase, correspondence buffer, two poses
constexpr int TICKET_GROUPS = 8;  // two-level reduction: workgroup b belongs to group b % 8 (its XCD under the observed dispatch order)
#ifndef FVH_SINGLE_LEVEL_MAX
#define FVH_SINGLE_LEVEL_MAX 128
#endif
constexpr int SINGLE_LEVEL_MAX_BLOCKS = FVH_SINGLE_LEVEL_MAX;  // grids up to this size MAY reduce in one level (one group: every thread of the reducing workgroup polls <= 16 rows, 8 at a time); the host picks (fvh_capi.hip: default_groups)
// Everything that crosses workgroups inside the persistent kernel travels as {value, tag} PAIRS: one naturally aligned 16-byte
// agent-scope (sc1, write-through) store per lane, polled with 16-byte sc1 loads. The tag names (launch, trip), so a pair is its
// own arrival signal -- no counter, no fence, no "wait for the store, then raise a flag" (MI355X_MICROARCH.md: data-tagged
// granules are the cheapest hand-off, ~1 us; a returning device-scope atomic plus a dependent load is two round trips).
typedef double pair_t __attribute__((ext_vector_type(2)));
constexpr size_t TAGGED_ROWS_OFFSET = (size_t)PART_STRIDE * (MAX_PARTIAL_ROWS + 2 * TICKET_GROUPS);  // doubles into CostParams::partials
constexpr int GLOBAL_ROW = TICKET_GROUPS;                                                          // multi-GPU: the all-reduced sums, published by workgroup 0 to the other collectors
constexpr size_t TAGGED_ROWS_DOUBLES = 2 * (TICKET_GROUPS + 1) * PART_STRIDE * 2;                  // group rows: [parity][group | global][32] pairs
constexpr size_t WG_ROWS_OFFSET = TAGGED_ROWS_OFFSET + TAGGED_ROWS_DOUBLES;                         // workgroup rows of the persistent kernel: [MAX_PARTIAL_ROWS][32] pairs
constexpr size_t WG_ROWS_DOUBLES = (size_t)MAX_PARTIAL_ROWS * PART_STRIDE * 2;
constexpr size_t PARTIALS_DOUBLES = WG_ROWS_OFFSET + WG_ROWS_DOUBLES;
__device__ __forceinline__ void store_pair_agent(pair_t* p, pair_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// XCD-local flavour: a plain store stays in (and is served from) the L2 of the writer's XCD; readers ON THAT XCD see it with the same
// L1-bypassing sc1 loads, as an L2 hit (MI355X_MICROARCH.md: same-XCD hand-off with plain producer stores). Not visible to other XCDs.
__device__ __forceinline__ void store_pair_xcd(pair_t* p, pair_t v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned xcc_id() {  // the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v;
}
__device__ __forceinline__ pair_t load_pair_agent(const pair_t* p) {
  pair_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// eight pairs in flight together, ONE wait (the loads and the wait sit in one asm block: nothing can touch a destination
// register before its data has landed)
__device__ __forceinline__ void load_pairs8_agent(pair_t (&v)[8], const pair_t* const (&p)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}

// PERSIST = true: ONE launch runs the whole LM loop. Every trip of the outer loop is what one launch of the non-persistent
// kernel does; instead of exiting, the workgroups hand their sums to their group's collector and poll its broadcast (the
// protocol is described where it is implemented, at the end of the kernel). This removes the per-launch dispatch + ramp
// (~4.5 us of an 18 us launch at 17k points, FVH_COST_TIMING) and the speculative no-op launches. All workgroups must be
// co-resident (the host clamps the grid to the occupancy limit); a watchdog on every poll turns a stuck hand-off into an abort
// flag + fallback to the multi-launch path instead of a hang.
__device__ __forceinline__ double uniform_f64(double x) {  // wave-uniform value -> SGPR pair
  const unsigned long long u = (unsigned long long)__double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// gfx950 v_permlane{32,16}_swap on a double (both dwords): afterwards the lanes whose bit W is clear hold (own a, partner's a)
// and the lanes whose bit W is set hold (partner's b, own b) in (a, b) -- partner = lane ^ W -- so that a + b is the pair
// sum of `a` on the lower lane and of `b` on the upper lane.
template <int W>
__device__ __forceinline__ void swap_halves(double& a, double& b) {
  static_assert(W == 32 || W == 16, "v_permlane32_swap / v_permlane16_swap");
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  unsigned alo = (unsigned)ua, ahi = (unsigned)(ua >> 32), blo = (unsigned)ub, bhi = (unsigned)(ub >> 32);
  if constexpr (W == 32) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    alo = r0[0]; blo = r0[1]; ahi = r1[0]; bhi = r1[1];
  } else {
    const auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    alo = r0[0]; blo = r0[1]; ahi = r1[0]; bhi = r1[1];
  }
  a = __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo));
  b = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}

#ifdef FVH_COST_TIMING
// Debug build only (-DFVH_COST_TIMING): 100 MHz wall-clock stamps of the LAST workgroup's walk through the epilogue
// (slot 0 = earliest workgroup start of the launch, slot 9 = the last workgroup's own start). Read with
// fvh_debug_cost_timing(); tools/cost_timing.py prints the breakdown.
__device__ unsigned long long g_cost_timing[16];
__device__ unsigned long long g_ptime[16][512][12];  // persistent kernel, per trip and workgroup: {start, main end, row published, broadcast issued (collectors), broadcast seen, group row published, collector holds all sums, ... LM done}; plain stores, no shared address
#define FVH_STAMP(i) do { if (threadIdx.x == 0) stamp[i] = wall_clock64(); } while (0)
#define FVH_PT_MIN(trip, k) do { if (threadIdx.x == 0 && (trip) < 16 && lb < 512) g_ptime[trip][lb][k] = wall_clock64(); } while (0)
#define FVH_PT_MAX(trip, k) FVH_PT_MIN(trip, k)
// main-loop timeline of wave 0 (tools/main_timing.py): waits for everything in flight, then stamps -- it serialises what the
// scheduler would overlap, so the segments are upper bounds
__device__ unsigned long long g_mtime[16][512][12];
#define FVH_MT(trip, k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0 && (trip) < 16 && lb < 512) g_mtime[trip][lb][k] = wall_clock64(); } while (0)
#else
#define FVH_STAMP(i) do { } while (0)
#define FVH_PT_MIN(trip, k) do { } while (0)
#define FVH_PT_MAX(trip, k) do { } while (0)
#ifdef FVH_ASM_MARKS  // static instruction counts per section: hipcc -S -DFVH_ASM_MARKS, then tools/count_isa.py
#define FVH_MT(trip, k) asm volatile("; FVH_MARK " #k)
#define FVH_MARK(k) asm volatile("; FVH_MARK " #k)
#else
#define FVH_MT(trip, k) do { } while (0)
#endif
#endif
#ifndef FVH_MARK
#define FVH_MARK(k) do { } while (0)
#endif

// Workgroups per CU the register allocator must make room for (the persistent grid has to be co-resident, so this is part
// of the design, not a hint): 3 -> at most 168 VGPRs. LDS (5 KB) no longer limits it.
#ifndef FVH_COST_WG_PER_CU
#define FVH_COST_WG_PER_CU 3
#endif
#define FVH_COST_BOUNDS __launch_bounds__(256, FVH_COST_WG_PER_CU)
// CH: voxel lookups per work item the code is unrolled for -- COST_CH, or 1 for launches whose items hold a single offset (NDT D2D over a
// few thousand source voxels, DIRECT1): the four-chunk code executed its three dead chunks masked, ~40 % of the main loop's instructions
// on a grid with one wave per SIMD, where nothing hides an instruction (tools/count_isa.py: 2,200 -> see profiles/r05_isa_counts.txt).
// a pointer into LDS that stays one (a 32-bit offset; reads are ds_read, waited for with lgkmcnt only)
template <typename Real> using LdsPose = const Pose<Real> __attribute__((address_space(3)))*;
typedef const int __attribute__((address_space(3)))* LdsInt;
template <typename Real>
__device__ __forceinline__ Pose<Real> lds_pose(LdsPose<Real> p, int which) {
  Pose<Real> o;
#pragma unroll
  for (int k = 0; k < 9; k++) o.r[k] = p[which].r[k];
#pragma unroll
  for (int k = 0; k < 3; k++) o.t[k] = p[which].t[k];
  return o;
}

template <typename Real, int MODE, bool PERSIST, int CH = COST_CH, bool GN = false>
__global__ FVH_COST_BOUNDS void cost_kernel(CostParams P) {
  static_assert(CH == 1 || CH == COST_CH, "one or COST_CH lookups per item");
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
  // Workgroup id / count. The dispatcher hands consecutive blocks to consecutive XCDs (block b runs on XCD (b + c) % 8 with one c per
  // launch: tools/probes/probe_xcc.hip). That is an observation, not a contract: every workgroup publishes the c it sees with its sums and
  // the collectors end the launch if they ever differ (`xcd_local` below).
  const unsigned lb = blockIdx.x, nb = gridDim.x;
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
  // The occupancy grid of a large map and the box of a sharded one are constant during a launch: staged in LDS once. Read per item
  // through their device pointers they were dependent FLAT round trips in front of the probes -- `enabled`, then the rest of the grid,
  // then the four occupancy words one after the other (each behind the full wait of the one before); six short-circuited loads for the box.
  __shared__ int s_grid[8];    // b0[3], nb[3], enabled
  __shared__ int s_region[8];  // inner_lo[3], inner_hi[3]
  constexpr int OFFP_LDS = 64;
  __shared__ int s_offp[OFFP_LDS];  // the packed neighbour offsets (DIRECT1 / 7 / 27 and small RADIUS sets): four of an item's eleven first loads
  if (threadIdx.x >= 64 && threadIdx.x < 64 + OFFP_LDS) s_offp[threadIdx.x - 64] = ((int)threadIdx.x - 64 < P.n_off) ? P.offsets_packed[threadIdx.x - 64] : 0;
  if (threadIdx.x < 8) {
    const int t = threadIdx.x;
    int gv = 0, rv = 0;
    if (P.bitmap && P.grid) gv = t < 3 ? P.grid->b0[t] : (t < 6 ? P.grid->nb[t - 3] : (t == 6 ? P.grid->enabled : 0));
    if (P.region) rv = t < 3 ? P.region->inner_lo[t] : (t < 6 ? P.region->inner_hi[t - 3] : 0);
    s_grid[t] = gv; s_region[t] = rv;
  }
  __syncthreads();
  // The number of source elements of an NDT D2D launch lives on the device (the source map's voxel counter) and does not change during the
  // launch: read once here. (Read per trip it was a dependent global round trip in front of every trip's first loads.)
  const int n_src_launch = P.d_n_src ? *P.d_n_src : P.n_src;
  // Sticky items (persistent launches whose every thread has at most ONE item: the 17k headline, NDT frames, DIRECT1 / 7 on small clouds):
  // a thread's item is the same on every trip, so what does not depend on the pose -- the element's index, point and covariance, and the
  // voxel ids it found on earlier trips (both correspondence buffers) -- is kept in LDS after the first trip: the first round trip of
  // every later trip (source element + stored ids, ~0.3 us from L2) becomes an LDS read. 12 + 1 + 2 CH KB per workgroup; each thread
  // touches its own slots only (no barrier). The ids still go to the global buffers too (getters, the per-transition route after an abort).
  constexpr int STICKY_T = PERSIST ? 256 : 1;
  __shared__ float4 s_src[3][STICKY_T];
  // the element index of a thread's FIRST ELEM_ITS items (a thread's items are the same on every trip of a launch): on clouds walked in
  // Morton order `order[i0]` is a dependent round trip in front of everything else the item loads -- after the first trip it is an LDS read
  constexpr int ELEM_ITS = 2;
  __shared__ int s_elem[PERSIST ? ELEM_ITS : 1][STICKY_T];
  __shared__ int s_ids[2][CH][STICKY_T];
  __shared__ int s_ofp[CH][STICKY_T];  // the item's packed neighbour offsets
  bool a_primed = false;  // wave roles: waves 0-1 skip the plain linearisations, so they fill their sticky slots on their FIRST fused trip, whichever that is
  for (;;) {  // PERSIST: one trip per LM transition; otherwise exactly one trip
  if (PERSIST) FVH_PT_MIN(gen, 0);
  const bool fused = (P.host_phase < 0) && (phase == PH_TRIAL);  // trial error (old ids) + speculative linearisation at xi (new ids)
  // (external find -- FastGICP's device LM -- exists in the per-transition VGICP instantiation only: it is never persistent)
  constexpr bool EXT_OK = !PERSIST && MODE == MODE_VGICP;
  const bool external = EXT_OK && P.external_find != 0;
  const bool do_find = !external && ((phase == PH_LINEARIZE) || (phase == PH_FIND_ONLY) || fused);
  const bool do_cost = (phase != PH_FIND_ONLY);
  const bool do_deriv = (phase == PH_LINEARIZE) || (phase == PH_EVAL_DERIV) || fused;
  // The two poses are wave-uniform: as scalars they cost 48 SGPRs for the whole main loop, and this kernel already
  // spills hundreds of SGPRs into VGPR lanes (446 in the persistent variant, which pushed it one register past the 256
  // VGPRs two workgroups per CU allow). They live in LDS instead and are read per element, where their registers die
  // before the lookups start.
  // (s_pose[0] = lin: rotation used by the cached Mahalanobis of the OLD ids; s_pose[1] = ev: evaluation pose, also the
  // linearisation pose of the NEW ids; filled before the first trip and by the barrier code of every persistent trip)
  const Real res = (Real)P.res, inv_res = (Real)P.inv_res;
  const int n_src = n_src_launch;
  int w_lo = (P.item_hi > 0 ? min(P.item_lo, n_src) : 0) * P.groups_per_src;           // this rank's tile of the item list
  int n_items = (P.item_hi > 0 ? min(P.item_hi, n_src) : n_src) * P.groups_per_src;     // (end of the range)
  if (P.tile_n > 1) {  // (kernel argument: uniform) device-side element count: the tile is cut here
    const int chunk = (n_src + P.tile_n - 1) / P.tile_n, lo = min(P.tile_rank * chunk, n_src);
    w_lo = lo * P.groups_per_src;
    n_items = min(lo + chunk, n_src) * P.groups_per_src;
  }
  int* corr_old = P.corr + (size_t)corr_sel * P.corr_stride;                      // read (stored ids)
  int* corr_new = fused ? P.corr + (size_t)(corr_sel ^ 1) * P.corr_stride : corr_old;  // written by the find
  // Wave roles (NDT instantiations with one lookup per item, device LM, grids of at most one workgroup per CU -- `P.split`): a fused trip
  // is two independent pieces of work per item -- (A) the trial error of the STORED id at the trial pose (old record, R_lin C R_lin^T,
  // one Mahalanobis term) and (B) the new linearisation (voxel coordinate, probe, record, R_ev C R_ev^T, hit term, 28 item sums, the
  // butterfly). With one wave per SIMD nothing overlaps them and every instruction is ~6 cycles of latency, while 3/4 of the chip's
  // SIMDs hold no wave at all: so a workgroup takes 128 items instead of 256, waves 0-1 run (A) and waves 2-3 run (B) for the same
  // items. Roles are wave-uniform (scalar branches). The critical wave's instruction stream shrinks to (B). (A)'s waves read the
  // stored ids from (B)'s sticky slots (thread + 128) and contribute slot 28 (the trial error) only.
  constexpr bool SPLIT_OK = (MODE != MODE_VGICP) && CH == 1;
  const bool split = SPLIT_OK && P.split != 0;
  const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool role_a = split && wave_id < 2;   // trial error of the stored ids only
  const bool role_b = split && wave_id >= 2;  // everything else
  const int wstride = split ? 128 : 256;
  const bool sticky = PERSIST && ((long long)(n_items - w_lo) <= (long long)nb * wstride);  // (uniform, the same on every trip of a launch)
  const bool cached = sticky && (role_a ? a_primed : gen > 0);
  const int sel_new = fused ? (corr_sel ^ 1) : corr_sel;

  // Lane-distributed wave accumulator: after every work item the wave reduces the 29 sums of its 64 items with a
  // transposing butterfly (below) that leaves the wave total of slot L >> 1 in lane L -- so the running sums of the whole
  // main loop live in ONE fp64 register per lane instead of 29 (58 VGPRs of a kernel that sat at the 256-VGPR limit).
  // At 17k points a thread has a single item per trip, i.e. the butterfly runs once, exactly like a reduction after the loop.
  double wacc = 0.0;
  const int lane = threadIdx.x & 63;

  const float4* tf = reinterpret_cast<const float4*>(P.table);
  // (consecutive threads take consecutive items on purpose: spreading a workgroup's items over the cloud made the launch
  // 24 % slower -- the loop is sensitive to how many distinct cache lines a wave touches)
  // The loop bound is wave-uniform (the butterfly needs all 64 lanes); lanes past the end contribute zeros.
  int item_no = -1;  // (wave-uniform) which of this thread's items the loop is at
  for (int wbase = w_lo + (int)lb * wstride + (split ? ((wave_id & 1) << 6) : (int)(threadIdx.x & 192)); wbase < n_items; wbase += (int)nb * wstride) {
    item_no++;
    if (role_a && !fused) continue;  // (a plain linearisation / an error-only evaluation has no second piece of work: waves 0-1 contribute zeros)
    const int w = wbase + lane;
    ItemAcc<Real> it = {{0, 0, 0, 0, 0, 0}, {0, 0, 0}, 0};
    Real acc_y = 0;  // fused: trial error with the old ids
    if (PERSIST && P.prio_mode && lb >= (unsigned)P.prio_from) __builtin_amdgcn_s_setprio(1);  // (the dispatcher fills the CUs once before any gets a second workgroup)
    Vec3<Real> q = {0, 0, 0};
    bool any_hit = false;  // an element without correspondences contributes exact zeros (its q may be non-finite: 0 * NaN)
    bool left_shard = false;  // sharded target map: this element's neighbourhood is not wholly inside the rank's shard
    if (PERSIST) FVH_MT(gen, 0);
    if (w < n_items) {
    const int i0 = P.gps_magic ? (int)__umulhi((unsigned)w, P.gps_magic) : w / P.groups_per_src;  // (a 32-bit division is ~30 instructions)
    const int g = w - i0 * P.groups_per_src;
    const int st = PERSIST ? (int)threadIdx.x : 0;  // this thread's sticky slot
    int i;
    // (sticky launches: `cached`, slot 0; others: the first ELEM_ITS items of a thread from the second trip on -- wave roles only exist on sticky launches)
    const bool elem_cached = PERSIST && P.order != nullptr && (cached || (!split && gen > 0 && item_no < ELEM_ITS));
    if (cached) i = s_elem[0][st];
    else if (elem_cached) i = s_elem[item_no][st];
    else {
      i = P.order ? P.order[i0] : i0;
      if (PERSIST && !sticky && P.order != nullptr && item_no < ELEM_ITS) s_elem[item_no][st] = i;
    }
    const float4* a4_src = P.src_sorted ? P.src_sorted + i0 : P.src_pts + i;  // (uniform choice) in Morton order the sorted copy is read contiguously
    const int o_begin = g * P.group, o_end = min(P.n_off, o_begin + P.group);
    // The loop is a chain of dependent memory round trips (0.3 us each when coalesced, 0.5-0.8 us when scattered; measured
    // with tools/main_timing.py) with ~4 us of fp64 arithmetic between them, on 2 waves per SIMD -- nothing hides a round trip
    // unless the SAME wave has independent work for it. So every load is issued as soon as its address is known:
    //   round trip 1: the source element, and -- their addresses depend on (i, g) only -- the stored ids and the neighbour
    //                 offsets of the first chunk;
    //   round trip 2: the records of the stored ids (fused trip), in flight during q, the voxel coordinate and R_lin C R_lin^T;
    //   round trip 3: the first key probes, in flight during the trial error of the stored ids;
    //   round trip 4: only for slots whose voxel changed (and rare probe continuations), in flight during R_ev C R_ev^T.
    // The order of the arithmetic is chosen for registers as much as for latency: 40 VGPRs of records are live from round
    // trip 2 on, so the two rotations of C_A are never live together (a first version that was, spilled 100 registers and ran
    // 40 % slower). Measured effect of the early issue at 17k points: none (8.3 us per trip before and after) -- with two
    // waves per SIMD the loop is bound by VALU issue (~1,650 VALU instructions per wave and trip, ~4.25 cycles each, PMC), and
    // the other wave already filled the round trips; the younger workgroup of a CU is the one that finishes late.
    // ---- round trip 1 ----
    // (Written as ONE branch on `cached` with straight-line loads inside: as per-value selects "cached ? LDS : global" inside the
    // slot loop every slot's loads ended at a join with a full wait -- the stored ids and offsets of the four slots were four dependent
    // round trips behind the source element's, on every item of a cloud too large for sticky items. tools/scan_serial_loads.py, round 5.)
    float4 a4, c0 = make_float4(0, 0, 0, 0), c1 = c0;
    const bool ext_fused = EXT_OK && external && fused;
    int b[CH], bo[CH];
    int ofp[CH];  // packed neighbour offsets
#pragma unroll
    for (int c = 0; c < CH; c++) { bo[c] = -1; b[c] = -1; ofp[c] = 0; }
    const bool want_bo = fused && !role_b, want_ofp = do_find && !role_a, want_b = !role_a && !do_find;
    if (cached) {
      a4 = s_src[0][st];
      if (MODE != MODE_NDT_P2D) { c0 = s_src[1][st]; c1 = s_src[2][st]; }
      if (want_bo) {
#pragma unroll
        for (int c = 0; c < CH; c++) bo[c] = s_ids[corr_sel][c][role_a ? st + 128 : st];
      }
      if (want_ofp) {
#pragma unroll
        for (int c = 0; c < CH; c++) ofp[c] = s_ofp[c][st];
      } else if (want_b) {  // (an error-only evaluation of a persistent launch: the stored ids of the current buffer)
#pragma unroll
        for (int c = 0; c < CH; c++) b[c] = s_ids[corr_sel][c][st];
      }
    } else {
      const size_t row = (size_t)(P.corr_by_position ? i0 : i) * P.n_off + o_begin;
      a4 = *a4_src;
      if (MODE != MODE_NDT_P2D && do_cost) {
        const float4* cv = P.src_cov_sorted ? P.src_cov_sorted + 2 * (size_t)i0 : P.src_cov + 2 * (size_t)i;  // (uniform choice)
        c0 = cv[0]; c1 = cv[1];
      }
      // (slots past the item's end read the item's last slot again and are masked BEHIND the branch: a select on a loaded value inside its
      // group would be a use of the load right behind its issue, and the next group's loads would wait for it)
      if (want_bo) {
#pragma unroll
        for (int c = 0; c < CH; c++) bo[c] = corr_old[row + min(c, o_end - 1 - o_begin)];
      }
      if (want_ofp) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          // (the table through a pointer TYPED as LDS: as two plain loads the compiler selected between the two ADDRESSES and issued one FLAT load)
          const int oi = min(o_begin + c, o_end - 1);
          ofp[c] = (P.n_off <= OFFP_LDS) ? ((LdsInt)s_offp)[oi] : P.offsets_packed[oi];
        }
      } else if (want_b) {
        const int* src_ids = ext_fused ? corr_new : corr_old;
#pragma unroll
        for (int c = 0; c < CH; c++) b[c] = src_ids[row + min(c, o_end - 1 - o_begin)];
      }
      if (sticky) {  // (trip 0 of a persistent launch: a linearisation, do_cost and do_find hold)
        s_elem[0][st] = i; s_src[0][st] = a4;
        if (MODE != MODE_NDT_P2D) { s_src[1][st] = c0; s_src[2][st] = c1; }
        if (want_ofp) {
#pragma unroll
          for (int c = 0; c < CH; c++) s_ofp[c][st] = ofp[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {  // slots past the item's end hold no id (the sticky copies say so already)
      const bool in = o_begin + c < o_end;
      bo[c] = in ? bo[c] : -1; b[c] = in ? b[c] : -1;
    }
    float4 q1[CH], q2[CH];
    // voxel records: of the old ids first (fused), then of the ids of this evaluation. q3 = {c_yz, c_zz, weight sqrt(n) as a double};
    // NDT does not use the weight and loads 8 bytes only
    using Q3 = typename std::conditional<MODE == MODE_VGICP, float4, float2>::type;
    Q3 q3[CH];
    // (a work item is at most CH offsets -- cost_shape() -- so this is the whole item: no chunk loop)
    // (the source point is consumed HERE, in front of round trip 2: its load is older than the ids', so this costs no wait of its own --
    // consumed behind the record loads, the join of "fused" and "not fused" needs a full wait for it, which drains those loads too)
    Vec3<Real> a = {(Real)a4.x, (Real)a4.y, (Real)a4.z};
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z));
    if (PERSIST) FVH_MT(gen, 1);
    // ---- round trip 2 (fused): issued before the arithmetic below, which does not need it ----
    if (fused && !role_b) {  // the records of the stored ids (bucket 0 for "none")
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const size_t base = (size_t)max(bo[c], 0) * 4;
        q1[c] = tf[base + 1]; q2[c] = tf[base + 2]; q3[c] = *reinterpret_cast<const Q3*>(tf + base + 3);
      }
    }
    // opaque: the pose loads stay inside the iteration instead of becoming 48 loop-invariant registers. The pointer keeps its LDS address
    // space (a 32-bit offset): as a generic pointer the reads were FLAT loads, whose wait is vmcnt(0) AND lgkmcnt(0) -- every read of the
    // pose drained the record loads that were meant to stay in flight behind the arithmetic (tools/scan_serial_loads.py, round 5)
    LdsPose<Real> pose_ptr = (LdsPose<Real>)s_pose;
    asm volatile("" : "+v"(pose_ptr));
    Sym3<Real> RCR = {0, 0, 0, 0, 0, 0}, RCR_old = {0, 0, 0, 0, 0, 0};
    int cx = 0, cy = 0, cz = 0;
    bool coord_ok = true;  // false: non-finite / out-of-range source point -> no correspondences (and no (int)floor(NaN))
    {
      const Pose<Real> ev = lds_pose(pose_ptr, 1);
      q = transform(ev, a);
      if (do_find && !role_a) {
        Vec3<Real> ql = q;
        if (!fused) { const Pose<Real> lin = lds_pose(pose_ptr, 0); ql = transform(lin, a); }
        const Real fx = floor(div_by(ql.x, res, inv_res) - (Real)0.5), fy = floor(div_by(ql.y, res, inv_res) - (Real)0.5), fz = floor(div_by(ql.z, res, inv_res) - (Real)0.5);
        coord_ok = voxel_index_ok(fx, fy, fz);
        cx = coord_ok ? (int)fx : 0;
        cy = coord_ok ? (int)fy : 0;
        cz = coord_ok ? (int)fz : 0;
        if (P.region) {  // (kernel argument: uniform) sharded target map: is every voxel this element can reach in the shard?
          LdsInt rg = (LdsInt)s_region;
          asm volatile("" : "+v"(rg));  // re-read per item (LDS): six values must not live across the main loop
          const int l0 = rg[0], l1 = rg[1], l2 = rg[2], h0 = rg[3], h1 = rg[4], h2 = rg[5];
          left_shard = coord_ok && ((cx < l0) | (cx > h0) | (cy < l1) | (cy > h1) | (cz < l2) | (cz > h2));
        }
      }
    }
    // the ids found by THIS launch are linearised at `ev` when fused (R_ev C R_ev^T is formed after the probes, below), at
    // `lin` otherwise; the stored ids of a fused trip use `lin`
    if (MODE != MODE_NDT_P2D && do_cost) {
      const Sym3<Real> CA = {(Real)c0.x, (Real)c0.y, (Real)c0.z, (Real)c0.w, (Real)c1.x, (Real)c1.y};
      Real Rl[9];
#pragma unroll
      for (int k = 0; k < 9; k++) Rl[k] = pose_ptr[0].r[k];
      if (fused) { if (!role_b) RCR_old = rotate_cov(Rl, CA); } else RCR = rotate_cov(Rl, CA);
    }
    if (PERSIST) FVH_MT(gen, 2);
    {
      unsigned long long key[CH];
      unsigned slot[CH];
      unsigned long long k0[CH];
      float4 sq1 = make_float4(0, 0, 0, 0), sq2 = sq1;  // CH == 1: the speculatively loaded record of the home slot
      Q3 sq3 = {};
      constexpr unsigned long long DEAD_KEY = FVH_EMPTY_KEY - 1;  // no voxel has it (keys use 63 bits): "this lookup does not exist" without a flag register
      // ---- round trip 3: CH independent first probes in flight ----
      if (do_find && !role_a) {
        bool live[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
          const int x = cx + (int)(ofp[c] & 1023) - 512, y = cy + (int)((ofp[c] >> 10) & 1023) - 512, z = cz + (int)((ofp[c] >> 20) & 1023) - 512;
          live[c] = (o_begin + c < o_end) && coord_ok && coord_in_range(x, y, z);
          key[c] = live[c] ? pack_key(x, y, z) : DEAD_KEY;
        }
        bool filtered = false;
        if (P.bitmap) {  // (kernel argument: uniform) large map: the cache-resident occupancy bits answer the misses -- only guaranteed hits go to the table
          LdsInt gp = (LdsInt)s_grid;
          asm volatile("" : "+v"(gp));  // re-read per item (LDS): seven values must not live across the main loop
          const int gb0 = gp[0], gb1 = gp[1], gb2 = gp[2], gn0 = gp[3], gn1 = gp[4], gn2 = gp[5];
          if (gp[6]) {  // (enabled; 0: the map's box did not fit the bitmap budget -- every lookup goes to the table as before)
            filtered = true;
            unsigned long long w[CH];
            unsigned word[CH], bit[CH];  // (the bitmap budget is 32 MB = 4 M words: 32 bits index it)
#pragma unroll
            for (int c = 0; c < CH; c++) {  // (vm_grid_locate on the staged values)
              const unsigned ux = (unsigned)(key[c] & 0x1FFFFF), uy = (unsigned)((key[c] >> 21) & 0x1FFFFF), uz = (unsigned)((key[c] >> 42) & 0x1FFFFF);
              const int bx = (int)(ux >> 2) - gb0, by = (int)(uy >> 2) - gb1, bz = (int)(uz >> 2) - gb2;
              live[c] = live[c] && (unsigned)bx < (unsigned)gn0 && (unsigned)by < (unsigned)gn1 && (unsigned)bz < (unsigned)gn2;
              word[c] = live[c] ? ((unsigned)bz * (unsigned)gn1 + (unsigned)by) * (unsigned)gn0 + (unsigned)bx : 0u;
              bit[c] = (ux & 3u) | ((uy & 3u) << 2) | ((uz & 3u) << 4);
            }
#pragma unroll
            for (int c = 0; c < CH; c++) w[c] = P.bitmap[word[c]];  // unconditional (dead lookups read word 0): CH independent loads, one round trip
#pragma unroll
            for (int c = 0; c < CH; c++) {
              live[c] = live[c] && ((w[c] >> bit[c]) & 1ull);
              if (!live[c]) key[c] = DEAD_KEY;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
          slot[c] = hash_slot(key[c], P.mask);
          // (filtered: a lookup the bitmap answered reads slot 0 -- one line for all of them -- and is not looked at; unfiltered: the rare dead
          // lookup reads a valid slot. Either way the CH loads are unconditional and independent: one round trip)
          // Whatever slot 0 holds cannot match DEAD_KEY, and a dead lookup never continues a probe: no select on the loaded value (it would be
          // a use of the load right behind its issue -- the probes are meant to be in flight during the trial error below).
          k0[c] = P.keys[(filtered && !live[c]) ? 0u : slot[c]];
        }
        // One-lookup items (CH == 1: 122 VGPRs, far from the cliff): the RECORD of the home slot is requested together with its key. At a
        // load factor <= 0.25 the home slot is the answer of nearly every hit, so the dependent round trip "key -> record" of a slot
        // whose voxel changed (every slot of a plain linearisation) disappears; a miss or a continued probe simply does not use it.
        if constexpr (CH == 1) {
          if (do_cost && !(fused && (int)slot[0] == bo[0])) {  // (fused and unchanged: the old record in q1..q3 IS the record)
            const size_t sb = (size_t)slot[0] * 4;
            sq1 = tf[sb + 1]; sq2 = tf[sb + 2]; sq3 = *reinterpret_cast<const Q3*>(tf + sb + 3);
          }
        }
      }
      if (PERSIST) FVH_MT(gen, 3);
      // ---- fused trip: trial error with the OLD ids while the probes are in flight ----
      if (fused && do_cost && !role_b) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          if (bo[c] < 0) continue;
          const float4 o1 = q1[c], o2 = q2[c];
          const Q3 o3 = q3[c];
          const int npts = (int)o1.w;
          const Vec3<Real> mu = {(Real)o1.x, (Real)o1.y, (Real)o1.z};
          const Sym3<Real> A = {(Real)o2.x + RCR_old.xx, (Real)o2.y + RCR_old.xy, (Real)o2.z + RCR_old.xz, (Real)o2.w + RCR_old.yy, (Real)o3.x + RCR_old.yz, (Real)o3.y + RCR_old.zz};
          const Vec3<Real> e = {mu.x - q.x, mu.y - q.y, mu.z - q.z};
          Real wgt;
          if constexpr (MODE == MODE_VGICP) {
            if (npts <= 0) continue;
            wgt = record_weight<Real>(o3);
          } else {
            if (npts <= 6) continue;
            const Real ksq = res * res;
            wgt = ksq / (ksq + (e.x * e.x + e.y * e.y + e.z * e.z));
          }
          // w e^T A^-1 e = (w / det A) e^T adj(A) e: the adjugate is never scaled
          Real det;
          const Sym3<Real> Adj = adjugate(A, det);
          const Vec3<Real> Ce = mul(Adj, e);
          acc_y += fast_div(wgt, det) * (e.x * Ce.x + e.y * Ce.y + e.z * Ce.z);
        }
      }
      if (PERSIST) FVH_MT(gen, 4);
      if (do_find && !role_a) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          const unsigned long long k = k0[c];
          int r = -1;
          if (k == key[c]) r = (int)slot[c];
          else if (k != FVH_EMPTY_KEY && key[c] != DEAD_KEY) r = probe_continue(P.keys, P.mask, key[c], slot[c]);  // rare at load <= 0.25
          b[c] = r;
          if (o_begin + c < o_end) corr_new[(size_t)(P.corr_by_position ? i0 : i) * P.n_off + o_begin + c] = r;
          if (sticky) s_ids[sel_new][c][st] = (o_begin + c < o_end) ? r : -1;
        }
      }
      if (do_cost && !role_a) {
      if (PERSIST) FVH_MT(gen, 5);
      // ---- round trip 4: the voxel records of the ids of this evaluation (bucket 0 for misses). Fused trip: near convergence
      // the new id of a slot IS its old id -- then the record is already in registers; only slots whose voxel changed load.
      if (fused) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          if (b[c] >= 0 && b[c] != bo[c]) {  // the voxel of this slot changed: fetch its record now
            if (CH == 1 && do_find && b[c] == (int)slot[c]) { q1[c] = sq1; q2[c] = sq2; q3[c] = sq3; continue; }  // ... unless it came with the key
            const size_t base = (size_t)b[c] * 4;
            q1[c] = tf[base + 1]; q2[c] = tf[base + 2]; q3[c] = *reinterpret_cast<const Q3*>(tf + base + 3);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          if (CH == 1 && do_find && b[c] == (int)slot[c]) { q1[c] = sq1; q2[c] = sq2; q3[c] = sq3; continue; }
          const size_t base = (size_t)max(b[c], 0) * 4;
          q1[c] = tf[base + 1]; q2[c] = tf[base + 2]; q3[c] = *reinterpret_cast<const Q3*>(tf + base + 3);
        }
      }
      if (fused && MODE != MODE_NDT_P2D) {  // R_ev C_A R_ev^T, while the records of changed ids are in flight
        LdsPose<Real> pose_ptr2 = (LdsPose<Real>)s_pose;
        asm volatile("" : "+v"(pose_ptr2) : : "memory");  // not before this point (see the register note above)
        const Sym3<Real> CA = {(Real)c0.x, (Real)c0.y, (Real)c0.z, (Real)c0.w, (Real)c1.x, (Real)c1.y};
        Real Re[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Re[k] = pose_ptr2[1].r[k];
        RCR = rotate_cov(Re, CA);
      }
      if (PERSIST) FVH_MT(gen, 6);
#pragma unroll
      for (int c = 0; c < CH; c++) {
        if (b[c] < 0) continue;
        const int npts = (int)q1[c].w;
        const Vec3<Real> mu = {(Real)q1[c].x, (Real)q1[c].y, (Real)q1[c].z};
        const Sym3<Real> A = {(Real)q2[c].x + RCR.xx, (Real)q2[c].y + RCR.xy, (Real)q2[c].z + RCR.xz, (Real)q2[c].w + RCR.yy, (Real)q3[c].x + RCR.yz, (Real)q3[c].y + RCR.zz};
        Real wgt;
        if constexpr (MODE == MODE_VGICP) {
          if (npts <= 0) continue;
          wgt = record_weight<Real>(q3[c]);  // sqrt(n): fast_vgicp_impl.hpp:149, compute_derivatives.cu:78 (stored with the voxel)
        } else {
          if (npts <= 6) continue;  // ndt_compute_derivatives.cu:61,133
          const Real ex = mu.x - q.x, ey = mu.y - q.y, ez = mu.z - q.z;
          const Real ksq = res * res;
          wgt = ksq / (ksq + (ex * ex + ey * ey + ez * ez));  // cauchy(resolution, |e|) :15-18
        }
        hit_term<Real>(it, q, mu, A, wgt, do_deriv);
        any_hit = true;
      }
      }  // do_cost
    }
    }  // w < n_items
    if (PERSIST && P.prio_mode) __builtin_amdgcn_s_setprio(0);
    if (!do_cost) continue;  // host-mode PH_FIND_ONLY
    if (PERSIST) FVH_MT(gen, 7);
    if (role_a) {  // one number per item: the wave's total goes where the butterfly would have left slot 28 (lanes 56, 57)
      double tot = (double)acc_y;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
      if ((lane >> 1) == NSUM) wacc += tot;
      continue;
    }

    // ---- per-item wave reduction: transposing butterfly over 32 slots (28 sums, the fused trial error, 3 zeros) ----
    // Step m = 32, 16, 8, 4, 2: the lane pair (L, L ^ m) splits its current slots in halves, each lane keeps one half and
    // receives the partner's values of that half -- 16 + 8 + 4 + 2 + 1 exchanges instead of 29 x 6 for a per-value tree --
    // and after the last step (m = 1) lane L holds the WAVE total of slot L >> 1. The two big steps use gfx950's
    // v_permlane32_swap / v_permlane16_swap, which exchange "my upper half" against "your lower half" in place: no select,
    // no LDS crossbar, 2 instructions per double. (Round 1 reduced through a 61 KB tile[29][264] LDS transpose, which
    // together with 256 VGPRs pinned the kernel at two workgroups per CU.) Every slot takes the same summation tree: the
    // trial error is the same number whether it travels in slot 28 (fused) or in slot 0 (host-mode error evaluation).
    {
      double v[32];
      if (do_deriv) {
        const Vec3<Real> qz = {any_hit ? q.x : (Real)0, any_hit ? q.y : (Real)0, any_hit ? q.z : (Real)0};
        item_sums<Real>(v, it, qz);
      } else {
        v[0] = (double)it.err;
#pragma unroll
        for (int j = 1; j < NSUM; j++) v[j] = 0.0;
      }
      v[28] = (double)acc_y;
      v[29] = v[31] = 0.0;
      v[30] = left_shard ? 1.0 : 0.0;  // travels with the sums -- through the workgroup rows, the group rows and the exchange between the ranks -- so that every rank learns it
#pragma unroll
      for (int j = 0; j < 16; j++) { double a = v[j], b = v[j + 16]; swap_halves<32>(a, b); v[j] = a + b; }
#pragma unroll
      for (int j = 0; j < 8; j++) { double a = v[j], b = v[j + 8]; swap_halves<16>(a, b); v[j] = a + b; }
#pragma unroll
      for (int m = 8; m >= 2; m >>= 1) {
        const bool up = (lane & m) != 0;
        const int half = m >> 1;  // slots kept: 4, 2, 1
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (j < half) {
            const double lo = v[j], hi = v[j + half];
            const double recv = __shfl_xor(up ? lo : hi, m);
            v[j] = (up ? hi : lo) + recv;
          }
        }
      }
      v[0] += __shfl_xor(v[0], 1);
      wacc += v[0];
    }
    if (PERSIST) FVH_MT(gen, 8);
  }

  if (role_a && fused) a_primed = true;  // (their sticky slots are filled now)
  // ---- workgroup reduction: 4 waves x 32 slots through 1 KB of LDS ----------------------------------------
  if (!do_cost) return;  // host-mode PH_FIND_ONLY (never persistent)
  // Everything from here to the end of the trip is written in terms of `tid`, an OPAQUE copy of threadIdx.x made per trip:
  // in the persistent instantiation the epilogue sits inside the trip loop, and the optimiser otherwise hoists its ~30
  // thread-index-derived addresses and flags into the kernel prologue, where they stay live across the main loop
  // (213 instead of 163 VGPRs).
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  FVH_STAMP(1);
  if (PERSIST) FVH_PT_MAX(gen, 1);
  if ((tid & 1) == 0) red[tid >> 6][(tid & 63) >> 1] = wacc;  // lane L (even) publishes slot L >> 1 of its wave
  __syncthreads();

  // ---- two-level reduction (one level for small grids) ------------------------------------------
  // Workgroup b belongs to group b % NG (NG = 8: its XCD under the observed dispatch order). A group's rows are summed by
  // 256 threads -- thread (value v, chunk c) takes the rows c, c + 8, ... (<= 8 independent loads per batch, fixed order) --
  // into one group row; the group rows are summed in group order. Grids of <= 64 workgroups are ONE group: a single level.
  // The per-transition kernel finds "the last arriver" with atomic tickets; the persistent kernel has no tickets at all
  // (designated collectors poll tagged rows, below). Both take the SAME summation order: bit-identical sums on both routes.
  const unsigned NG = (unsigned)P.ng;  // (host, default_groups(): 8 = one group per XCD, 1 for tiny grids)
  const unsigned grp = lb % NG;
  const unsigned ngroups = NG;
  const unsigned gsize = (nb - grp + NG - 1) / NG;
  if constexpr (!PERSIST) {
    if (tid < PART_STRIDE) {
      const int vv = tid;
      const double x = (red[0][vv] + red[1][vv]) + (red[2][vv] + red[3][vv]);
      // write-through (sc1) so another workgroup can read it from L2 without a release fence;
      // row slot v: sums 0..27, fused trial error at 28 (== NSUM), 29..31 zero
      __hip_atomic_store(&P.partials[(size_t)lb * PART_STRIDE + vv], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  FVH_STAMP(2);
  // sum of this group's partial rows -> fin[chunk][v] -> one group row (fixed order: deterministic)
  auto reduce_group_rows = [&](size_t out_row) {
    {
      const int v = tid & 31, chunk = tid >> 5;  // 8 chunks x 32 values
      double s = 0.0;
      for (unsigned j0 = chunk; j0 < gsize; j0 += 8 * 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const unsigned j = j0 + 8 * u;
          t[u] = (j < gsize) ? __hip_atomic_load(&P.partials[(size_t)(grp + j * NG) * PART_STRIDE + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) s += t[u];
      }
      fin[chunk][v] = s;
    }
    __syncthreads();
    if (tid < PART_STRIDE) {
      const int v = tid;
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
      const int v = tid & 31;
      const unsigned g = tid >> 5;
      fin[g][v] = (g < ngroups) ? __hip_atomic_load(&P.partials[(first_row + g) * PART_STRIDE + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    __syncthreads();
    if (tid < PART_STRIDE) {
      const int v = tid;
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
    s_st.max_iterations = P.max_iterations; s_st.lm_max_iterations = P.lm_max_iterations; s_st.optimizer = P.optimizer; s_st.pad_ = 0;
    s_st.lambda = -1.0; s_st.nu = 2.0; s_st.y0 = 0.0;
    s_st.phase = PH_LINEARIZE; s_st.corr_cur = 0; s_st.delta_converged = 0; s_st.halo_exceeded = 0;
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
    if (tid == 0) s_last = (atomicAdd(&P.ticket[grp], 1u) == gsize - 1);
    __syncthreads();
    if (!s_last) return;
    FVH_STAMP(3);
    reduce_group_rows((size_t)MAX_PARTIAL_ROWS + grp);
    if (tid == 0) s_last = (atomicAdd(&P.ticket[TICKET_GROUPS], 1u) == ngroups - 1);
    __syncthreads();
    if (!s_last) return;
    FVH_STAMP(4);

    // ---- the very last workgroup: sum the group rows in group order (deterministic), LM step ----
    reduce_final((size_t)MAX_PARTIAL_ROWS);
    if (tid <= TICKET_GROUPS) P.ticket[tid] = 0;  // re-arm for the next launch
    if (MODE == MODE_VGICP && P.peer.n > 1) {  // multi-GPU: sum the blocks of all ranks (rank order -> bit-identical on every rank); one workgroup per rank is in here
      __syncthreads();
      if (!peer_exchange_sums(P.peer, red[0], P.peer.xbase, P.peer_watchdog_ticks, tid, 256, &s_last)) {
        if (tid == 0) __hip_atomic_store(&st->aborted, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a peer did not deliver: the host reports FVH_ERR_COMM
        return;
      }
    }
    // The LM step is one thread of dependent fp64 math; run it on an LDS copy of the state (a global
    // round trip per st-> access would cost more than the arithmetic) and write the state back with all lanes.
    for (int i = tid; i < ST_WORDS; i += 256) reinterpret_cast<unsigned long long*>(&s_st)[i] = st_words[i];
    int vm_nv = 0, vm_dr = 0, vm_nv2 = 0;
    if (tid == 0) {
      vm_nv = P.vm_counters[0];
      vm_nv2 = P.vm_counters2 ? P.vm_counters2[0] : 0;
      vm_dr = P.vm_counters[1] + (P.vm_counters2 ? P.vm_counters2[1] : 0);
    }
    __syncthreads();
    FVH_STAMP(5);
    if (tid == 0) {
      for (int v = 0; v < PART_STRIDE; v++) s_st.sums[v] = red[0][v];
      s_st.vm_num_voxels = vm_nv;
      s_st.vm_num_voxels2 = vm_nv2;
      s_st.vm_dropped = vm_dr;
      if (P.host_phase < 0 && P.init) init_state();
    }
    __syncthreads();
    if (P.host_phase < 0 && !P.defer_lm && tid < 64) dev_lm_step_wave<GN>(&s_st, red[0], tid, P.lm_trace);
    FVH_STAMP(6);
    __syncthreads();
    for (int i = tid; i < ST_WORDS; i += 256) st_words[i] = reinterpret_cast<const unsigned long long*>(&s_st)[i];
#ifdef FVH_COST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      stamp[7] = wall_clock64();
      for (int i = 1; i <= 7; i++) g_cost_timing[i] = stamp[i];
      g_cost_timing[8] = nb;
      g_cost_timing[9] = stamp[0];
    }
#endif
    return;
  } else {
    // ---- persistent trip: nobody leaves, and nobody takes a ticket --------------------------------
    // Every workgroup publishes its 32 sums as {sum, tag} pairs -- ONE 16-byte store per lane, no wait -- and goes straight to
    // polling the broadcast of its group. The first workgroup of each group (the COLLECTOR, lb < NG) polls the pairs of its
    // group's rows, eight in flight per thread, adds them in the fixed order of reduce_group_rows as soon as the last one has
    // landed, and publishes a tagged group row (write-through: this is the one hand-off that crosses XCDs). Then EVERY collector
    // polls all NG group rows on one wave, adds them in group order -- reduce_final's order --, runs the LM step in registers
    // on ITS OWN LDS copy of the state (the eight copies see identical sums and stay bit-identical; round 2 measured ~500
    // redundant steps, which thrash the instruction caches: eight do not) and broadcasts what the next trip needs -- phase,
    // correspondence buffer, two poses: 26 values -- as `reps` copies of 26 tagged pairs to the workgroups of its own group.
    // Round 3 had ONE workgroup run the step and broadcast to the whole chip: rows -> collector, group rows -> opener, broadcast ->
    // everybody were three memory-side round trips of ~1 us each. Now rows and broadcast stay inside the group = inside one XCD
    // (xcd_local: plain stores, served to the polling sc1 loads by that XCD's L2), and only the group rows pay the ~1 us.
    // lm_everywhere (grids of <= 2 workgroups per CU, single GPU): the group rows are polled by EVERY workgroup, each runs the step on
    // its own LDS state and nobody broadcasts -- the third hand-off is gone (17k: 126.1 -> 123.5 us; NDT frames 81.8 -> 79.0). The
    // step itself is ~0.3 us slower there (it shares its CU with computing or polling waves), so the gain is ~0.4 us per trip, and
    // with three workgroups per CU it turns into a loss (768 workgroups: +4-6 us per launch).
    // Multi-GPU: workgroup 0 alone meets the peers (kernels_peer.hpp) and hands the all-reduced sums to the other collectors as
    // one more tagged row. Workgroup 0 also owns what leaves the launch: the LM trace, the final state, the result word.
    // Dead ends measured in round 1/2 (474 workgroups, 17k points): one barrier word on the line of the arrival counters
    // (+10 us per trip); one barrier word + every workgroup reloading the state (21.6 us per trip: ~500 readers of the
    // same lines queue at their memory channel); EVERY workgroup running the LM step on a state RELOADED from memory each trip (22.7 us per trip; with the state resident in LDS and
    // the group rows as input it is the lm_everywhere flavour above).
    // Rows are single-buffered: a workgroup writes its trip t + 1 row only after it has seen its group's broadcast of trip t, which
    // the collector sends after every row of trip t has been consumed; group rows alternate between two parities (a collector can
    // be at most one trip ahead of the slowest one: it needs that one's next group row). Tags embed the launch sequence: rows of
    // older launches never match.
    const unsigned trip = gen;
    unsigned long long ltag = P.launch_tag;
    asm volatile("" : "+s"(ltag));  // opaque per trip: otherwise the two conversions below are hoisted out of the trip loop and held in 4 VGPRs across the main loop
    // (formed where they are used, from scalars: held as doubles they were four VGPRs live across the inlined LM step)
    auto want_tag_of = [&]() { unsigned long long t = ltag * 4096ull + trip + 1; asm volatile("" : "+s"(t)); return (double)t; };
    auto abort_tag_of = [&]() { unsigned long long t = ltag * 4096ull; asm volatile("" : "+s"(t)); return -(double)t; };  // launch-specific: a poisoned row of an older launch means nothing
    __shared__ double bc[BCAST_PAIRS];  // payload of the broadcast row as seen by this workgroup
    __shared__ unsigned s_abort;
    pair_t* wrows = reinterpret_cast<pair_t*>(P.partials + WG_ROWS_OFFSET);
    pair_t* trows = reinterpret_cast<pair_t*>(P.partials + TAGGED_ROWS_OFFSET) + (size_t)(trip & 1u) * (TICKET_GROUPS + 1) * PART_STRIDE;
    pair_t* bcast = reinterpret_cast<pair_t*>(P.bcast);
    const bool local = P.xcd_local != 0;  // (kernel argument: uniform)
    // copies of the broadcast row a group owns: all of them for a single group, PERSIST_REPLICAS / 8 each otherwise; workgroup lb
    // polls copy (lb / NG) % reps of its group -- no address is read by more than ~15 workgroups
    const unsigned reps = NG == 1 ? (unsigned)PERSIST_REPLICAS : (unsigned)(PERSIST_REPLICAS / TICKET_GROUPS);
    static_assert(PERSIST_REPLICAS % (2 * TICKET_GROUPS) == 0, "an even number of broadcast copies per group");
    pair_t* my_bcast = bcast + (size_t)(NG == 1 ? 0u : grp * reps) * BCAST_PAIRS;
    auto poison = [&](int first, int stride) {  // never hang the GPU: every tag of this launch's broadcast becomes the abort tag (write-through: best effort across XCDs, every poll also has its own watchdog)
      pair_t pv;
      pv.x = 0.0; pv.y = abort_tag_of();
      for (int idx = first; idx < PERSIST_REPLICAS * BCAST_PAIRS; idx += stride) store_pair_agent(bcast + idx, pv);
    };
    // The XCD-local flavour is only sound if every member of a group really runs on ONE XCD (HIP promises no placement). Members of a
    // group share blockIdx % 8, so it is enough that the dispatcher's rotation
    // c = (XCC_ID - blockIdx) mod 8 is the same for every workgroup of the launch: each workgroup publishes the c it sees in the unused
    // slot 31 of its row, the collectors compare (rows against their own, then the group rows among each other) and end the launch
    // with code 3 if anything differs -- the host redoes the align with one launch per transition and, after a few of those, stops
    // asking for the local flavour.
    const double my_rot = local ? (double)((__builtin_amdgcn_readfirstlane(xcc_id()) - blockIdx.x) & 7u) : 0.0;
    if (tid < PART_STRIDE) {
      pair_t pv;
      pv.x = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);  // row slot v: sums 0..27, fused trial error at 28 (== NSUM), 29..30 zero
      if (tid == PART_STRIDE - 1) pv.x = my_rot;                         // slot 31: the dispatcher's rotation as this workgroup sees it
      pv.y = want_tag_of();
      if (local) store_pair_xcd(wrows + (size_t)lb * PART_STRIDE + tid, pv); else store_pair_agent(wrows + (size_t)lb * PART_STRIDE + tid, pv);
    }
    FVH_PT_MAX(trip, 2);
    const bool collector = lb < NG;  // == the first workgroup of group `grp`
    const bool multi_gpu = MODE == MODE_VGICP && P.peer.n > 1;  // (kernel argument: uniform)
    const bool everywhere = P.lm_everywhere != 0 && !multi_gpu;  // (kernel arguments: uniform)
    const bool adds_rows = collector;  // (round 4 also let EVERY workgroup of a single-level grid add all rows itself: one hand-off fewer, paid back by the wider poll -- removed)
    if (collector || everywhere) {
      if (tid == 0) { s_last = 1; s_abort = 0u; }
      __syncthreads();
      if (adds_rows) {
        const int v = tid & 31, chunk = tid >> 5;
        const double want = want_tag_of();
        double s = 0.0;
        for (unsigned j0 = chunk; j0 < gsize; j0 += 8 * 8) {
          const pair_t* src[8];
          double t[8];
          unsigned pend = 0;
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const unsigned j = j0 + 8 * u;
            const bool in = j < gsize;
            src[u] = wrows + (size_t)(grp + (in ? j : j0) * NG) * PART_STRIDE + v;  // (rows past the end: a valid address, never looked at)
            t[u] = 0.0;
            pend |= in ? (1u << u) : 0u;
          }
          const unsigned long long t0 = wall_clock64();
          for (;;) {
            pair_t pv[8];
            load_pairs8_agent(pv, src);
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (((pend >> u) & 1u) && pv[u].y == want) { t[u] = pv[u].x; pend &= ~(1u << u); }
            if (!pend) break;
            if (wall_clock64() - t0 > P.watchdog_ticks) { s_last = 0; break; }  // a row never came: not every workgroup is resident / something is stuck
            __builtin_amdgcn_s_sleep(1);
          }
          if (v == PART_STRIDE - 1) {  // slot 31 is not a sum: count the rows whose rotation is not ours
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = (((j0 + 8 * u) < gsize) && t[u] != my_rot) ? 1.0 : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; u++) s += t[u];
        }
        fin[chunk][v] = s;
      }
      if (adds_rows) __syncthreads();  // (uniform per workgroup)
      if (!s_last) {
        poison(tid, 256);
        if (tid == 0) __hip_atomic_store(&st->aborted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      if (adds_rows && tid < PART_STRIDE) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 8; c++) s += fin[c][tid];
        if (tid == PART_STRIDE - 1) s = (s == 0.0) ? my_rot : -1.0;  // this group's verdict: its rotation, or "members disagree"
        if (NG > 1) {
          pair_t pv;
          pv.x = s; pv.y = want_tag_of();
          store_pair_agent(trows + (size_t)grp * PART_STRIDE + tid, pv);  // (crosses XCDs: always write-through)
        }
        red[1][tid] = s;  // single level: this IS the only group row
      }
      FVH_PT_MAX(trip, 5);
      // From here to the broadcast ONE wave does everything (the other three wait at the barrier below): lanes 0..31 poll the
      // pairs of the <= 8 group rows (eight in flight per lane) and add them in group order, the sums go to LDS, the LM step runs
      // in registers, lanes 0..25 read the payload back and store it to the group's copies. No __syncthreads, no LDS hop between
      // the stages. (red[1] above was written by this same wave: LDS operations of a wave complete in order.)
      if (tid < 64) {
        const int lane = tid;
        double val = 0.0;
        bool ok = true;
        if (NG == 1) {
          if (lane < PART_STRIDE) val = red[1][lane];
        } else if (!multi_gpu || lb == 0) {
          const pair_t* src[8];
          double t[8];
          unsigned pend = 0;
#pragma unroll
          for (int g = 0; g < 8; g++) {
            const bool in = (unsigned)g < NG;
            src[g] = trows + (size_t)(in ? g : 0) * PART_STRIDE + (lane & 31);
            t[g] = 0.0;
            pend |= in ? (1u << g) : 0u;
          }
          if (lane >= PART_STRIDE) pend = 0;
          const double want = want_tag_of();
          const unsigned long long t0 = wall_clock64();
          // (lm_everywhere: the non-collectors start polling at once; a delayed first look and spaced polls were measured and bought nothing,
          // profiles/r04_lm_everywhere.txt)
          while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
            pair_t pv[8];
            load_pairs8_agent(pv, src);
#pragma unroll
            for (int g = 0; g < 8; g++)
              if (((pend >> g) & 1u) && pv[g].y == want) { t[g] = pv[g].x; pend &= ~(1u << g); }
            if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > P.watchdog_ticks))) { ok = false; break; }
          }
          if (lane == PART_STRIDE - 1) {  // slot 31: every group must have seen our rotation
#pragma unroll
            for (int g = 0; g < 8; g++) t[g] = ((unsigned)g < NG && t[g] != my_rot) ? 1.0 : 0.0;
          }
#pragma unroll
          for (int g = 0; g < 8; g++) val += t[g];  // group order
        } else {  // multi-GPU, collectors 1..NG-1: the all-reduced sums come from workgroup 0
          const pair_t* src = trows + (size_t)GLOBAL_ROW * PART_STRIDE + (lane & 31);
          const double want = want_tag_of();
          const unsigned long long t0 = wall_clock64();
          bool pend = lane < PART_STRIDE;
          while (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
            const pair_t pv = load_pair_agent(src);
            if (pend && pv.y == want) { val = pv.x; pend = false; }
            if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > P.watchdog_ticks))) { ok = false; break; }
          }
        }
        FVH_PT_MAX(trip, 6);
        // slot 31 now holds: single level -- our rotation or -1; two levels -- the number of groups that disagree with us (the
        // multi-GPU collectors 1.. read workgroup 0's verdict instead, which is 0 or it would not have been published)
        if (lane == PART_STRIDE - 1) {
          const bool placement_ok = !local || (NG == 1 ? val == my_rot : val == 0.0);
          if (ok && !placement_ok) s_abort = 3u;  // 3: the workgroups of a group do not share an XCD
          val = 0.0;
        }
        if (lane < PART_STRIDE) red[0][lane] = val;
        if (!ok && lane == 0) s_abort = 1u;  // 1: not every workgroup is resident / something is stuck
      }
      if (multi_gpu && lb == 0) {
        // multi-GPU: workgroup 0 of every rank meets the others in each other's mailboxes (kernels_peer.hpp, all 256 threads); every rank
        // then runs the same LM step (VGICP handles only: the NDT handles shard through RCCL between launches)
        __syncthreads();
        if constexpr (MODE == MODE_VGICP) {
          if (!s_abort && !peer_exchange_sums(P.peer, red[0], P.peer.xbase + trip, P.peer_watchdog_ticks, tid, 256, &s_last)) {
            if (tid == 0) s_abort = 2u;  // 2: a peer did not deliver
          }
        }
        __syncthreads();
        if (!s_abort && NG > 1 && tid < PART_STRIDE) {
          pair_t pv;
          pv.x = red[0][tid]; pv.y = want_tag_of();
          store_pair_agent(trows + (size_t)GLOBAL_ROW * PART_STRIDE + tid, pv);
        }
      }
      if (tid < 64) {
        const int lane = tid;
        if (__builtin_amdgcn_ballot_w64(s_abort != 0u) == 0ull) {  // (written by this wave, or behind the barriers of the peer exchange)
          FVH_PT_MAX(trip, 7);
          if (trip == 0 && lane == 0) {
            init_state();
            s_st.vm_num_voxels = P.vm_counters[0];
            s_st.vm_num_voxels2 = P.vm_counters2 ? P.vm_counters2[0] : 0;
            s_st.vm_dropped = P.vm_counters[1] + (P.vm_counters2 ? P.vm_counters2[1] : 0);
          }
          if (lane < PART_STRIDE) s_st.sums[lane] = red[0][lane];
          FVH_PT_MAX(trip, 8);
          FVH_MARK(20);
#ifdef FVH_COST_TIMING
          const unsigned long long lm_c0 = __builtin_readcyclecounter();
#endif
          // The step is ~10 KB of code that runs once per trip -- always on the same few CUs (the collectors), so it stays in their
          // instruction caches, and the state stays in the collector's LDS for the whole launch instead of travelling through memory
          dev_lm_step_wave<GN>(&s_st, red[0], lane, lb == 0 ? P.lm_trace : nullptr);
#ifdef FVH_COST_TIMING
          if (threadIdx.x == 0 && trip < 16 && lb == 0) g_ptime[trip][0][11] = __builtin_readcyclecounter() - lm_c0;  // shader cycles of the LM step (next to its wall-clock stamps 8 -> 9)
#endif
          FVH_MARK(21);
          FVH_PT_MAX(trip, 9);
          // payload: phase, correspondence buffer, x_lin, evaluation pose of the next trip -- one value per lane, read back from the state
          const int ph = __builtin_amdgcn_readfirstlane(s_st.phase);
          double v = 0.0;
          const int d = lane & 31;  // both halves of the wave hold the payload: each store instruction fills TWO copies
          if (d < BCAST_VALUES) {
            const double* xl = reinterpret_cast<const double*>(&s_st.x_lin);
            const double* pe = reinterpret_cast<const double*>(ph == PH_LINEARIZE ? &s_st.x0 : &s_st.xi);
            if (d == 0) v = (double)ph;
            else if (d == 1) v = (double)s_st.corr_cur;
            else if (d < 14) v = xl[d - 2];
            else v = pe[d - 14];
            if (lane < 32) bc[d] = v;  // (this workgroup's own copy for the next trip)
            pair_t pv;
            pv.x = v; pv.y = want_tag_of();
            pair_t* dst = my_bcast + lane;  // lanes 32..63: the next copy (BCAST_PAIRS == 32)
            static_assert(BCAST_PAIRS == 32, "two copies per store instruction");
            if (everywhere) {  // (nobody polls a broadcast: every workgroup has just computed the payload itself)
            } else if (local) {
              for (unsigned r = 0; r < reps; r += 2) store_pair_xcd(dst + (size_t)r * BCAST_PAIRS, pv);
            } else {
              for (unsigned r = 0; r < reps; r += 2) store_pair_agent(dst + (size_t)r * BCAST_PAIRS, pv);
            }
          }
          FVH_PT_MAX(trip, 3);
        }
      }
      __syncthreads();
      if (s_abort) {  // never hang the GPU -- poison every tag of this launch and leave; the host takes it from `aborted`
        poison(tid, 256);
        if (tid == 0) __hip_atomic_store(&st->aborted, s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      if (lb == 0 && s_st.phase == PH_DONE) {  // the state leaves the LDS once, at the end (the kernel boundary makes it visible)
        for (int i = tid; i < ST_WORDS; i += 256) st_words[i] = reinterpret_cast<const unsigned long long*>(&s_st)[i];
        if (P.result_host) {
          // ... and goes straight to the host through mapped pinned memory: the caller spins on the sequence word instead of
          // paying a device-to-host copy kernel and a stream synchronisation (~8 us of a 300 us registration)
          for (int i = tid; i < ST_WORDS; i += 256)
            __hip_atomic_store(&P.result_host[i], reinterpret_cast<const unsigned long long*>(&s_st)[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) __hip_atomic_store(&P.result_host[ST_WORDS + 1], P.launch_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    } else {
      if (tid < 64) {  // wave 0 polls this workgroup's copy of its group's broadcast
        const pair_t* rep = my_bcast + (size_t)((lb / NG) % reps) * BCAST_PAIRS;
        const int lane = tid;
        const bool is_val = lane < BCAST_VALUES;
        const unsigned long long t0 = wall_clock64();
        const double want_tag = want_tag_of(), abort_tag = abort_tag_of();
        int ok = 0;
        for (;;) {
          pair_t pv;
          pv.x = 0.0; pv.y = want_tag;
          if (is_val) pv = load_pair_agent(rep + lane);
          if (__any(is_val && pv.y == abort_tag)) break;  // another workgroup's watchdog aborted THIS launch
          if (__all(pv.y == want_tag)) {
            if (is_val) bc[lane] = pv.x;
            ok = 1;
            break;
          }
          if (__builtin_amdgcn_readfirstlane((int)(wall_clock64() - t0 > P.watchdog_ticks))) {
            // not every workgroup is resident / something is stuck: never hang the GPU -- poison every tag and leave
            poison(lane, 64);
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
    if (tid < 24) {
      const int which = tid / 12, k = tid % 12;
      const Real v = (Real)bc[2 + tid];
      if (k < 9) s_pose[which].r[k] = v; else s_pose[which].t[k - 9] = v;
    }
    __syncthreads();  // bc[] is rewritten after the next barrier; s_pose is read by the next trip
  }
  }  // trips
}

}  // namespace fvh
