// Voxel-grid downsampling on gfx950 -- the step right before the registration path in every caller of the
// reference (src/align.cpp:136-147, src/kitti.cpp:80-82,117-119, src/python/main.cpp:46-62,80-92), which uses
// pcl::ApproximateVoxelGrid (benchmarks, pygicp) or pcl::VoxelGrid (src/test/gicp_test.cpp:55-65).  Both are
// third-party (PCL) algorithms; their restatements live in oracle/vgicp_oracle.cpp and these kernels reproduce
// them BIT-EXACTLY, including the fp32 summation order and the output order:
//
//  * VoxelGrid: key = linear voxel index relative to the cloud's min corner; stable radix sort of (key, index);
//    every run of equal keys is one output point, emitted in key order, summed in input order.
//  * ApproximateVoxelGrid looks inherently sequential (a 512-slot history that flushes a centroid whenever a
//    point hashes to a slot held by another voxel) but decomposes: a slot only ever sees the points that hash
//    to it, in input order.  So: ONE stable 9-bit radix pass by slot; inside a slot every maximal run of the same
//    voxel is one output centroid (summed in input order); the run is flushed by the first point of the next run
//    in that slot, so its output position is the rank of that point among all flush-triggering points (an
//    exclusive scan over ORIGINAL indices); the last run of every slot is flushed at the end in slot order.
//
// All work is integer/byte movement plus a handful of fp32 adds per point: HBM/latency bound, no LDS tiling needed.
#pragma once
#include "dev_math.hpp"
#include "kernels_sort.hpp"

namespace fvh {

constexpr int SCAN_BLOCK_ITEMS = 1024;  // 256 threads x 4 contiguous items
constexpr int AVG_SLOTS = 512;          // pcl::ApproximateVoxelGrid histsize_

// ---- generic exclusive scan of n unsigned values: block sums -> radix_scan_kernel on the sums -> apply ----
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const unsigned* __restrict__ data, int n, unsigned* __restrict__ block_sums) {
  __shared__ unsigned ws[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) block_sums[gridDim.x] = 0u;  // the spare entry that receives the grand total in the scan
  const int base = blockIdx.x * SCAN_BLOCK_ITEMS + threadIdx.x * 4;
  unsigned s = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) s += (base + u < n) ? data[base + u] : 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// out[i] = block_base[block] + exclusive prefix inside the block (out may alias data)
__global__ __launch_bounds__(256) void scan_apply_kernel(const unsigned* data, int n, const unsigned* __restrict__ block_base, unsigned* out) {
  __shared__ unsigned ws[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int base = blockIdx.x * SCAN_BLOCK_ITEMS + threadIdx.x * 4;
  unsigned v[4], s = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) { v[u] = (base + u < n) ? data[base + u] : 0u; s += v[u]; }
  unsigned x = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
  if (lane == 63) ws[wv] = x;
  __syncthreads();
  unsigned run = block_base[blockIdx.x] + x - s;
  for (int w = 0; w < wv; w++) run += ws[w];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (base + u < n) out[base + u] = run;
    run += v[u];
  }
}

// ---- keys ----
struct VgGrid {   // pcl::VoxelGrid: min corner / divisions of the voxel lattice over the cloud's bounding box
  int minb[3];
  int divb[3];
};

__device__ __forceinline__ bool finite3(const float4 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }

// the oracle's (and PCL's) fp32 arithmetic: floor(p * inv) as float, minus float(minb), truncated to int
__global__ __launch_bounds__(256) void vg_keys_exact_kernel(const float4* __restrict__ pts, int n, float inv, VgGrid g, unsigned* __restrict__ keys, int* __restrict__ idx,
                                                            unsigned* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  if (!finite3(p)) *bad = 1u;  // benign race: every writer stores the same value
  const int i0 = (int)(floorf(p.x * inv) - (float)g.minb[0]);
  const int i1 = (int)(floorf(p.y * inv) - (float)g.minb[1]);
  const int i2 = (int)(floorf(p.z * inv) - (float)g.minb[2]);
  keys[i] = (unsigned)i0 + (unsigned)i1 * (unsigned)g.divb[0] + (unsigned)i2 * (unsigned)g.divb[0] * (unsigned)g.divb[1];
  idx[i] = i;
}

__device__ __forceinline__ void avg_voxel(const float4 p, float inv, int& ix, int& iy, int& iz) {
  ix = (int)floorf(p.x * inv);
  iy = (int)floorf(p.y * inv);
  iz = (int)floorf(p.z * inv);
}

__global__ __launch_bounds__(256) void vg_keys_approx_kernel(const float4* __restrict__ pts, int n, float inv, unsigned* __restrict__ keys, int* __restrict__ idx,
                                                             unsigned* __restrict__ bad, unsigned* __restrict__ trig /* cleared here: saves a memset */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  trig[i] = 0u;
  int ix, iy, iz;
  if (!finite3(pts[i])) *bad = 1u;
  avg_voxel(pts[i], inv, ix, iy, iz);
  // (ix*7171 + iy*3079 + iz*4231) & 511 -- only the low 9 bits matter, so unsigned wrap-around is the same value
  keys[i] = ((unsigned)ix * 7171u + (unsigned)iy * 3079u + (unsigned)iz * 4231u) & (AVG_SLOTS - 1);
  idx[i] = i;
}

// ---- run heads ----
// exact: head[j] = first element of a run of equal keys.
__global__ __launch_bounds__(256) void vg_mark_exact_kernel(const unsigned* __restrict__ keys, int n, unsigned* __restrict__ head) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}

// approx: head[j] = slot start or a different voxel than the previous element of the slot; such a non-slot-start
// head is the point that flushes the previous run -> trig[its original index] = 1. slot_used[slot] = 1.
__global__ __launch_bounds__(256) void vg_mark_approx_kernel(const unsigned* __restrict__ keys, const int* __restrict__ idx, const float4* __restrict__ pts, int n, float inv,
                                                             unsigned* __restrict__ head, unsigned* __restrict__ trig, unsigned* __restrict__ slot_used) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const unsigned slot = keys[j];
  const bool slot_start = (j == 0) || (keys[j - 1] != slot);
  bool h = slot_start;
  const int id = idx[j];
  if (!slot_start) {
    int ax, ay, az, bx, by, bz;
    avg_voxel(pts[id], inv, ax, ay, az);
    avg_voxel(pts[idx[j - 1]], inv, bx, by, bz);
    h = (ax != bx) || (ay != by) || (az != bz);
    if (h) trig[id] = 1u;
  } else {
    slot_used[slot] = 1u;
  }
  head[j] = h ? 1u : 0u;
}

// ---- centroids ----
// One thread per run head: fp32 sums in input order (adds only, so nothing for the compiler to contract), then the
// correctly rounded division by the count -- the arithmetic of the PCL filters.
template <bool APPROX>
__global__ __launch_bounds__(256) void vg_emit_kernel(const unsigned* __restrict__ keys, const int* __restrict__ idx, const float4* __restrict__ pts, int n,
                                                      const unsigned* __restrict__ head, const unsigned* __restrict__ pos_scan /* exact: scan(head) by j; approx: scan(trig) by original index */,
                                                      const unsigned* __restrict__ slot_rank /* approx: exclusive scan of slot_used, [AVG_SLOTS] = #used */,
                                                      const unsigned* __restrict__ trig_total /* approx: number of flush-triggering points */, float* __restrict__ out,
                                                      const unsigned* __restrict__ bad, unsigned* __restrict__ result /* approx: {trigger count, used slots, bad} for ONE host copy */) {
  if (APPROX && result && blockIdx.x == 0 && threadIdx.x == 0) { result[0] = *trig_total; result[1] = slot_rank[AVG_SLOTS]; result[2] = *bad; }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || !head[j]) return;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int e = j;
  do {
    const float4 p = pts[idx[e]];
    cx += p.x; cy += p.y; cz += p.z;
    e++;
  } while (e < n && !head[e]);
  const float cnt = (float)(e - j);
  unsigned pos;
  if (APPROX) {
    const unsigned slot = keys[j];
    if (e < n && keys[e] == slot) pos = pos_scan[idx[e]];
    else pos = *trig_total + slot_rank[slot];
  } else {
    pos = pos_scan[j];
  }
  out[3 * (size_t)pos] = cx / cnt;
  out[3 * (size_t)pos + 1] = cy / cnt;
  out[3 * (size_t)pos + 2] = cz / cnt;
}

}  // namespace fvh
