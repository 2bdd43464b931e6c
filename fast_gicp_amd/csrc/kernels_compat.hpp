// FVH_COMPUTE_CUDA_COMPAT, the part that is not the k-NN covariance (kernels_cov.hpp: cov_from_neighbors_cuda_compat_kernel): the
// RBF covariance estimator and the voxel sums of the reference's DEVICE classes in THEIR arithmetic -- float, uncentred, every float
// operation in the association of the statement it restates, no fma contraction -- so that the engine can be held to
// oracle/cuda_compat.cpp (the float restatement of FastVGICPCuda / NDTCuda) instead of to the fp64 CPU class.
//
// Float sums depend on their ORDER. The reference does not fix one (gaussian_voxelmap.cu:89-148 accumulates with atomicAdd in arrival
// order; covariance_estimation_rbf.cu walks blocks of 512 candidates on one thread each and adds the block partials in block order, which
// IS fixed). This mode takes the order of the oracle leg, which is the order of a sequential run of the reference's kernels: candidates and
// points in INDEX order. A different order moves the result by what tests/test_oracle.py::test_cuda_compat_order_spread
// measures on the oracle leg itself.
//
// This mode is about parity, not speed: one thread per (query, block phase) / per voxel, sequential float sums.
#pragma once
#include "kernels_cov.hpp"
#include "kernels_voxelmap.hpp"

namespace fvh {

// ------------------------------------------------------------------------------------------------
// covariance_estimation_rbf.cu:40-109,120-150. For query x: for every block of 512 candidates (index order) a partial
//   {sum w, sum w p, sum (w p) p^T}, w = expf(-kernel_width * |x - p|^2) for |x - p|^2 <= max_dist^2   (:67-85, accumulate :40-44)
// starting from zero, candidates in index order; the partials are added in block order (finalization_kernel :98-104), then
//   mean = sum_p / sum_w;  cov = (sum_pp - mean sum_p^T) / sum_w                                    (finalize :47-52)
// and the regularisation of covariance_regularization.cu. WITHOUT the reference's padding points at the origin (:130-133: the last
// block is padded with (0,0,0) "points" that are not masked -- a query within max_dist of the origin counts them; oracle/cuda_compat.cpp
// leaves them out as well and says so).
// A workgroup = 64 queries x 4 block phases: wave s takes the blocks s, s + 4, ...; the 512 candidates of a wave's block are staged in
// LDS (every lane reads the same candidate: a broadcast read), the four partials of a round meet in LDS and wave 0 adds them in block
// order. expf: the correctly rounded single-precision exponential ((float)exp((double)x): glibc's expf, which the oracle calls, is
// correctly rounded in all but ~0.1 % of its arguments; CUDA's expf is a 2-ulp function, so the real device path is not reproducible
// to the bit by anybody).
// ------------------------------------------------------------------------------------------------
constexpr int RBFC_BLOCK = 512;
__global__ __launch_bounds__(256) void cov_rbf_cuda_compat_kernel(const float4* __restrict__ pts, int n, float exp_factor, float max_dist, int method, float4* __restrict__ cov,
                                                                  const int* __restrict__ subset /* a rank's tile of the Morton order, or null */, int m /* queries of this launch */) {
#pragma clang fp contract(off)
  __shared__ float4 s_cand[4][RBFC_BLOCK];
  __shared__ float s_part[4][13][64];
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int t = blockIdx.x * 64 + lane;
  const int qi = subset ? subset[min(t, m - 1)] : min(t, m - 1);
  const float4 x4 = pts[qi];
  const float x[3] = {x4.x, x4.y, x4.z};
  const float max_dist_sq = max_dist * max_dist;
  const int nblocks = (n + RBFC_BLOCK - 1) / RBFC_BLOCK;
  float tot[13];
#pragma unroll
  for (int q = 0; q < 13; q++) tot[q] = 0.f;
  for (int b0 = 0; b0 < nblocks; b0 += 4) {
    const int b = b0 + s;
    float part[13];
#pragma unroll
    for (int q = 0; q < 13; q++) part[q] = 0.f;
    const int begin = b * RBFC_BLOCK, cnt = b < nblocks ? min(n - begin, RBFC_BLOCK) : 0;
    __syncthreads();  // (the previous round's partials and candidates have been consumed)
    for (int j = lane; j < cnt; j += 64) s_cand[s][j] = pts[begin + j];
    __syncthreads();
    {
      for (int j = 0; j < cnt; j++) {
        const float4 p4 = s_cand[s][j];
        const float p[3] = {p4.x, p4.y, p4.z};
        const float dx = x[0] - p[0], dy = x[1] - p[1], dz = x[2] - p[2];
        const float sq = dx * dx + dy * dy + dz * dz;
        if (sq > max_dist_sq) continue;
        const float w = (float)exp((double)(-exp_factor * sq));
        part[0] += w;
#pragma unroll
        for (int a = 0; a < 3; a++) part[1 + a] += w * p[a];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c = 0; c < 3; c++) part[4 + a * 3 + c] += (w * p[a]) * p[c];  // w * x * x^T, left to right
      }
    }
#pragma unroll
    for (int q = 0; q < 13; q++) s_part[s][q][lane] = part[q];
    __syncthreads();
    if (s == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (b0 + u >= nblocks) break;
#pragma unroll
        for (int q = 0; q < 13; q++) {
          const float v = s_part[u][q][lane];
          tot[q] = (b0 + u == 0) ? v : tot[q] + v;  // (sum = dists[index]; sum += dists[...] in block order)
        }
      }
    }
  }
  if (s != 0 || t >= m) return;
  const float sw = tot[0];
  float mean[3];
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] = tot[1 + a] / sw;
  M3f C;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) C.m[a * 3 + c] = (tot[4 + a * 3 + c] - mean[a] * tot[1 + c]) / sw;
  const M3f R = regularize_cov_cuda_compat(C, method);
  cov[2 * (size_t)qi] = make_float4(R.m[0], R.m[1], R.m[2], R.m[4]);
  cov[2 * (size_t)qi + 1] = make_float4(R.m[5], R.m[8], 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// Voxel sums of the device classes (gaussian_voxelmap.cu:89-148 accumulate, :158-198 finalize; ndt_cuda.cu:128,139 MIN_EIG): float sums
// in POINT INDEX order per voxel. The map itself -- which voxels exist, their buckets, their point counts -- is built by the engine's own
// kernels (kernels_voxelmap.hpp) with the voxel coordinate computed in float (vector3_hash.cuh:35-38); this pass then REPLACES the
// mean / covariance of every voxel record by the float result:
//   1. vmc_point_bucket_kernel: the bucket of every point (probe of the finished key table) as a sort key, original index as payload;
//   2. a stable LSD radix sort of (bucket, index) -- radix_sort_pairs of host_downsample.inc.hpp -- groups a voxel's points, index order kept;
//   3. vmc_segment_heads_kernel: where each bucket's run starts;
//   4. vmc_finalize_kernel: one thread per voxel walks its run and accumulates in float, then finalises as the reference does.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool vmc_coord(const float4& p, float res, int& cx, int& cy, int& cz) {
#pragma clang fp contract(off)
  const float fx = floorf(p.x / res - 0.5f), fy = floorf(p.y / res - 0.5f), fz = floorf(p.z / res - 0.5f);  // (x.array() / resolution - 0.5).floor()
  const bool ok = voxel_index_ok(fx, fy, fz);
  cx = ok ? (int)fx : 0; cy = ok ? (int)fy : 0; cz = ok ? (int)fz : 0;
  return ok;
}
__global__ __launch_bounds__(256) void vmc_point_bucket_kernel(const float4* __restrict__ pts, int n, float res, const unsigned long long* __restrict__ keys, unsigned mask,
                                                               unsigned* __restrict__ out_key, int* __restrict__ out_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  unsigned bucket = mask + 1u;  // "no voxel": sorts behind every bucket (non-finite point, or outside a sharded map's box)
  if (vmc_coord(pts[i], res, cx, cy, cz)) {
    const unsigned long long key = pack_key(cx, cy, cz);
    unsigned slot = hash_slot(key, mask);
    for (unsigned it = 0; it <= mask; it++) {
      const unsigned long long k = keys[slot];
      if (k == key) { bucket = slot; break; }
      if (k == FVH_EMPTY_KEY) break;
      slot = (slot + 1) & mask;
    }
  }
  out_key[i] = bucket;
  out_idx[i] = i;
}
__global__ __launch_bounds__(256) void vmc_segment_heads_kernel(const unsigned* __restrict__ sorted_key, int n, int* __restrict__ seg_start) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const unsigned k = sorted_key[p];
  if (p == 0 || sorted_key[p - 1] != k) seg_start[k] = p;
}
// MODE 0: VGICP (mean of the points, mean of the point covariances); MODE 1: NDT (sample covariance from uncentred sums, MIN_EIG)
template <int MODE>
__global__ __launch_bounds__(64) void vmc_finalize_kernel(const float4* __restrict__ pts, const float4* __restrict__ cov, const int* __restrict__ sorted_idx, const int* __restrict__ seg_start,
                                                          const int* __restrict__ occupied, const int* __restrict__ counters, uint4* __restrict__ table,
                                                          float4* __restrict__ compact_pts, float4* __restrict__ compact_cov) {
#pragma clang fp contract(off)
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= counters[0]) return;
  const int b = occupied[id];
  const int cnt = (int)table[(size_t)b * 4].z;
  const int* run = sorted_idx + seg_start[b];
  float sum[3] = {0.f, 0.f, 0.f};
  float S[9];
#pragma unroll
  for (int q = 0; q < 9; q++) S[q] = 0.f;
  for (int j = 0; j < cnt; j++) {
    const int i = run[j];
    const float4 p4 = pts[i];
    const float p[3] = {p4.x, p4.y, p4.z};
#pragma unroll
    for (int a = 0; a < 3; a++) sum[a] += p[a];
    if (MODE == 0) {  // the engine holds a point covariance as its upper triangle (kernels_cov.hpp)
      const float4 c0 = cov[2 * (size_t)i], c1 = cov[2 * (size_t)i + 1];
      S[0] += c0.x; S[1] += c0.y; S[2] += c0.z; S[4] += c0.w; S[5] += c1.x; S[8] += c1.y;
    } else {
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) S[a * 3 + c] += p[a] * p[c];
    }
  }
  const float nf = (float)cnt;
  float mean[3];
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] = sum[a] / nf;
  float4 r2, r3;
  float4* tf = reinterpret_cast<float4*>(table);
  const float4 old3 = tf[(size_t)b * 4 + 3];  // .zw: the weight sqrt(n) as a double (vm_finalize_kernel) -- kept
  if (MODE == 0) {
    r2 = make_float4(S[0] / nf, S[1] / nf, S[2] / nf, S[4] / nf);
    r3 = make_float4(S[5] / nf, S[8] / nf, old3.z, old3.w);
  } else {
    M3f C;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) C.m[a * 3 + c] = (S[a * 3 + c] - mean[a] * sum[c]) / nf;  // (cov - mean * sum^T) / n, gaussian_voxelmap.cu:184-193
    const M3f R = regularize_cov_cuda_compat(C, 1 /* MIN_EIG */);
    r2 = make_float4(R.m[0], R.m[1], R.m[2], R.m[4]);
    r3 = make_float4(R.m[5], R.m[8], old3.z, old3.w);
  }
  tf[(size_t)b * 4 + 1] = make_float4(mean[0], mean[1], mean[2], nf);
  tf[(size_t)b * 4 + 2] = r2;
  tf[(size_t)b * 4 + 3] = r3;
  if (compact_pts) {
    compact_pts[id] = make_float4(mean[0], mean[1], mean[2], 0.f);
    compact_cov[2 * (size_t)id] = r2;
    compact_cov[2 * (size_t)id + 1] = r3;
  }
}

}  // namespace fvh
