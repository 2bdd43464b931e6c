// Spatial (Morton) ordering of a cloud on gfx950: bounding box -> 27-bit Morton keys (9 bits per
// axis) -> stable LSD radix sort, 3 passes of 9 bits -> sorted float4 copy with the ORIGINAL point
// index carried in .w.  The reference has no counterpart (its O(N^2) ops sweep everything); here the
// order is what makes the exact k-NN / RBF / fitness sweeps cullable by tile bounding boxes, and a
// contiguous range of the sorted cloud is a spatial tile for the multi-GPU shard.
//
// Per pass: radix_hist (per-wave digit histograms in LDS -> hist[bin][wave]), radix_scan (one
// workgroup, exclusive scan over bins x waves = scatter bases), radix_scatter (each wave walks its
// contiguous range 64 items at a time; same-digit lanes find each other with 9 ballots, the lowest
// lane of a digit group advances the wave's cursor, rank = popcount of lower peers -> stable).
#pragma once
#include "dev_math.hpp"

namespace fvh {

constexpr int RADIX_BITS = 9;
constexpr int RADIX_BINS = 1 << RADIX_BITS;
constexpr int RADIX_PASSES = 3;
constexpr int SORT_ITEMS_PER_WAVE = 1024;  // contiguous items a wave owns per pass (16 steps of 64)

// order-preserving float <-> uint mapping for atomicMin/Max
__device__ __forceinline__ unsigned float_to_ordered(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// box[0..2] = min xyz, box[3..5] = max xyz (ordered-uint encoded); host pre-sets min = 0xFFFFFFFF, max = 0
__global__ __launch_bounds__(256) void cloud_bbox_kernel(const float4* __restrict__ pts, int n, unsigned* __restrict__ box) {
  __shared__ unsigned s[6];
  if (threadIdx.x < 3) s[threadIdx.x] = 0xFFFFFFFFu;
  else if (threadIdx.x < 6) s[threadIdx.x] = 0u;
  __syncthreads();
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float4 p = pts[i];
    lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
    hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&s[a], float_to_ordered(lo[a]));
      atomicMax(&s[3 + a], float_to_ordered(hi[a]));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&box[threadIdx.x], s[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&box[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ unsigned spread3_9(unsigned v) {  // 9 bits -> every third bit
  v &= 0x1FF;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

__global__ __launch_bounds__(256) void morton_keys_kernel(const float4* __restrict__ pts, int n, const unsigned* __restrict__ box, unsigned* __restrict__ keys,
                                                          int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lx = ordered_to_float(box[0]), ly = ordered_to_float(box[1]), lz = ordered_to_float(box[2]);
  const float ex = ordered_to_float(box[3]) - lx, ey = ordered_to_float(box[4]) - ly, ez = ordered_to_float(box[5]) - lz;
  const float extent = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-6f));
  const float scale = 511.999f / extent;  // cubic cells so the key is isotropic
  const float4 p = pts[i];
  const unsigned ix = (unsigned)fminf(511.f, fmaxf(0.f, (p.x - lx) * scale));
  const unsigned iy = (unsigned)fminf(511.f, fmaxf(0.f, (p.y - ly) * scale));
  const unsigned iz = (unsigned)fminf(511.f, fmaxf(0.f, (p.z - lz) * scale));
  keys[i] = spread3_9(ix) | (spread3_9(iy) << 1) | (spread3_9(iz) << 2);
  idx[i] = i;
}

// hist[bin * nwaves + wave]
__global__ __launch_bounds__(256) void radix_hist_kernel(const unsigned* __restrict__ keys, int n, int shift, int nwaves, unsigned* __restrict__ hist) {
  __shared__ unsigned h[4][RADIX_BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  for (int b = lane; b < RADIX_BINS; b += 64) h[wv][b] = 0;
  __syncthreads();
  if (wave < nwaves) {
    const int begin = wave * SORT_ITEMS_PER_WAVE, end = min(n, begin + SORT_ITEMS_PER_WAVE);
    for (int i = begin + lane; i < end; i += 64) atomicAdd(&h[wv][(keys[i] >> shift) & (RADIX_BINS - 1)], 1u);
  }
  __syncthreads();
  if (wave < nwaves)
    for (int b = lane; b < RADIX_BINS; b += 64) hist[(size_t)b * nwaves + wave] = h[wv][b];
}

// exclusive scan of `count` unsigned values in place, one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void radix_scan_kernel(unsigned* __restrict__ data, int count) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  const int per = (count + 1023) / 1024;  // contiguous items per thread
  const int begin = tid * per, end = min(count, begin + per);
  unsigned local = 0;
  for (int i = begin; i < end; i++) local += data[i];
  // inclusive scan of `local` across the workgroup
  unsigned x = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(x, off);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  unsigned base = 0;
  for (int w = 0; w < wv; w++) base += wsum[w];
  unsigned run = base + x - local;  // exclusive prefix of this thread's first item
  for (int i = begin; i < end; i++) {
    const unsigned v = data[i];
    data[i] = run;
    run += v;
  }
}

// stable scatter of (key, idx); on the last pass also gathers the point into the sorted cloud with
// its original index in .w
__global__ __launch_bounds__(256) void radix_scatter_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ idx_in, int n, int shift, int nwaves,
                                                            const unsigned* __restrict__ offsets /* scanned hist */, unsigned* __restrict__ keys_out, int* __restrict__ idx_out,
                                                            const float4* __restrict__ pts, float4* __restrict__ sorted_pts) {
  __shared__ unsigned cur[4][RADIX_BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  if (wave >= nwaves) return;
  for (int b = lane; b < RADIX_BINS; b += 64) cur[wv][b] = offsets[(size_t)b * nwaves + wave];
  // the cursors are private to this wave: wave-level ordering is enough (no workgroup barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int begin = wave * SORT_ITEMS_PER_WAVE, end = min(n, begin + SORT_ITEMS_PER_WAVE);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int base = begin; base < end; base += 64) {
    const int i = base + lane;
    const bool valid = i < end;
    const unsigned key = valid ? keys_in[i] : 0xFFFFFFFFu;
    const int id = valid ? idx_in[i] : -1;
    const unsigned d = (key >> shift) & (RADIX_BINS - 1);
    // lanes holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < RADIX_BITS; bit++) {
      const unsigned long long m = __ballot((d >> bit) & 1);
      peers &= ((d >> bit) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int cnt = __popcll(peers);
    const int leader = __ffsll((long long)peers) - 1;
    unsigned dst_base = 0;
    if (valid && lane == leader) {
      dst_base = cur[wv][d];
      cur[wv][d] = dst_base + cnt;
    }
    dst_base = __shfl(dst_base, leader);
    if (valid) {
      const unsigned dst = dst_base + rank;
      keys_out[dst] = key;
      idx_out[dst] = id;
      if (sorted_pts) {
        float4 p = pts[id];
        p.w = __int_as_float(id);
        sorted_pts[dst] = p;
      }
    }
  }
}

}  // namespace fvh
