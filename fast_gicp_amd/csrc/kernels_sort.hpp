// Spatial (Morton) ordering of a cloud on gfx950: bounding box -> 27-bit Morton keys (9 bits per
// axis) -> stable LSD radix sort, 3 passes of 9 bits -> sorted float4 copy with the ORIGINAL point
// index carried in .w.  The reference has no counterpart (its O(N^2) ops sweep everything); here the
// order is what makes the exact k-NN / RBF / fitness sweeps cullable by tile bounding boxes, and a
// contiguous range of the sorted cloud is a spatial tile for the multi-GPU shard.
//
// Per pass: radix_hist (per-wave digit histograms in LDS -> hist[bin][wave]), radix_binscan (one wave per
// bin: prefixes over the waves + bin total), radix_scan (512 bin totals), radix_scatter (each wave walks its
// contiguous range 64 items at a time; same-digit lanes find each other with 9 ballots, the lowest
// lane of a digit group advances the wave's cursor, rank = popcount of lower peers -> stable).
#pragma once
#include <cstddef>
#include "dev_math.hpp"

namespace fvh {

constexpr int RADIX_BITS = 9;
constexpr int RADIX_BINS = 1 << RADIX_BITS;
constexpr int RADIX_PASSES = 3;
constexpr int SORT_ITEMS_MAX = 1024;  // contiguous items a wave owns per pass (runtime: 256 for small clouds -> more, shorter waves)


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

// 27-bit Morton key of a point inside the cloud's bounding cube (cubic cells: the key is isotropic).
// min_inverted: the box comes from pack_points_kernel (box[0..2] = ~ordered(min)), not from cloud_bbox_kernel
struct MortonFrame { float lx, ly, lz, scale; };
__device__ __forceinline__ MortonFrame morton_frame(const unsigned* __restrict__ box, int min_inverted) {
  const unsigned flip = min_inverted ? 0xFFFFFFFFu : 0u;
  MortonFrame f;
  f.lx = ordered_to_float(box[0] ^ flip); f.ly = ordered_to_float(box[1] ^ flip); f.lz = ordered_to_float(box[2] ^ flip);
  const float ex = ordered_to_float(box[3]) - f.lx, ey = ordered_to_float(box[4]) - f.ly, ez = ordered_to_float(box[5]) - f.lz;
  f.scale = 511.999f / fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-6f));
  return f;
}
__device__ __forceinline__ unsigned morton27(const MortonFrame& f, const float4& p) {
  const unsigned ix = (unsigned)fminf(511.f, fmaxf(0.f, (p.x - f.lx) * f.scale));
  const unsigned iy = (unsigned)fminf(511.f, fmaxf(0.f, (p.y - f.ly) * f.scale));
  const unsigned iz = (unsigned)fminf(511.f, fmaxf(0.f, (p.z - f.lz) * f.scale));
  return spread3_9(ix) | (spread3_9(iy) << 1) | (spread3_9(iz) << 2);
}
__global__ __launch_bounds__(256) void morton_keys_kernel(const float4* __restrict__ pts, int n, const unsigned* __restrict__ box, unsigned* __restrict__ keys,
                                                          int* __restrict__ idx, int min_inverted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = morton27(morton_frame(box, min_inverted), pts[i]);
  idx[i] = i;
}

// hist[bin * nwaves + wave]. With `pts` the keys do not exist yet: this first stage of the sort computes them on the way
// (and stores them for its scatter) -- one dependent stage less than a key kernel in front of it (a stage of this chain is ~5 us
// whatever it does).
template <int BITS>
__global__ __launch_bounds__(256) void radix_hist_kernel(unsigned* __restrict__ keys, int n, int shift, int nwaves, int items, unsigned* __restrict__ hist,
                                                         const float4* __restrict__ pts = nullptr, const unsigned* __restrict__ box = nullptr, int min_inverted = 0) {
  constexpr int BINS = 1 << BITS;
  __shared__ unsigned h[4][BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  for (int b = lane; b < BINS; b += 64) h[wv][b] = 0;
  __syncthreads();
  if (wave < nwaves) {
    const int begin = wave * items, end = min(n, begin + items);
    if (pts) {
      const MortonFrame f = morton_frame(box, min_inverted);
      for (int i = begin + lane; i < end; i += 64) {
        const unsigned k = morton27(f, pts[i]);
        keys[i] = k;
        atomicAdd(&h[wv][(k >> shift) & (BINS - 1)], 1u);
      }
    } else {
      for (int i = begin + lane; i < end; i += 64) atomicAdd(&h[wv][(keys[i] >> shift) & (BINS - 1)], 1u);
    }
  }
  __syncthreads();
  if (wave < nwaves)
    for (int b = lane; b < BINS; b += 64) hist[(size_t)b * nwaves + wave] = h[wv][b];
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

// Scatter bases of a pass without scanning the whole bins x waves matrix on one workgroup (that single-workgroup scan
// was 32 us at 118k keys and ~300 us at 1M -- more than the rest of the pass): one wave per bin turns the bin's row of
// per-wave counts into exclusive prefixes and emits the bin total; radix_scan_kernel then scans the 512 totals, and
// the scatter adds bin_base[bin] to the in-bin prefix.
__global__ __launch_bounds__(256) void radix_binscan_kernel(unsigned* __restrict__ hist /* [bins][nwaves] */, int nwaves, unsigned* __restrict__ totals /* [bins] */, int bins = RADIX_BINS) {
  const int lane = threadIdx.x & 63;
  const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bin >= bins) return;
  unsigned* row = hist + (size_t)bin * nwaves;
  unsigned run = 0;
  for (int base = 0; base < nwaves; base += 64) {
    const int w = base + lane;
    const unsigned v = (w < nwaves) ? row[w] : 0u;
    unsigned x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (w < nwaves) row[w] = run + x - v;
    run += __shfl(x, 63);
  }
  if (lane == 0) totals[bin] = run;
}

// stable scatter of (key, idx); on the last pass also gathers the point into the sorted cloud with
// its original index in .w
template <int BITS>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ idx_in /* null: the identity (first pass) */, int n, int shift, int nwaves, int items,
                                                            const unsigned* __restrict__ offsets /* in-bin prefixes */, const unsigned* __restrict__ bin_base /* scanned bin totals */,
                                                            unsigned* __restrict__ keys_out, int* __restrict__ idx_out,
                                                            const float4* __restrict__ pts, float4* __restrict__ sorted_pts) {
  constexpr int BINS = 1 << BITS;
  __shared__ unsigned cur[4][BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  if (wave >= nwaves) return;
  for (int b = lane; b < BINS; b += 64) cur[wv][b] = offsets[(size_t)b * nwaves + wave] + bin_base[b];
  // the cursors are private to this wave: wave-level ordering is enough (no workgroup barrier)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int begin = wave * items, end = min(n, begin + items);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int base = begin; base < end; base += 64) {
    const int i = base + lane;
    const bool valid = i < end;
    const unsigned key = valid ? keys_in[i] : 0xFFFFFFFFu;
    const int id = valid ? (idx_in ? idx_in[i] : i) : -1;
    const unsigned d = (key >> shift) & (BINS - 1);
    // lanes holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < BITS; bit++) {
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


// ------------------------------------------------------------------------------------------------
// Mid-size clouds (SORT_SMALL_MAX < n <= SORT_FUSED_MAX, e.g. the 100k-point scans of configs[2] / [4]): TWO launches per pass.
// The four-launch pass above spends two launches (radix_binscan, radix_scan: ~7 + ~5 us + their gaps) on a few microseconds of
// prefix arithmetic whose input is small: with 2,048 points per workgroup a 100k-point cloud is 49 workgroups, i.e. 49 rows of
// per-workgroup digit counts. The histogram kernel writes those rows ([workgroup][bin], next to the per-wave rows [wave][bin]) and
// every workgroup of the scatter derives its own cursors from them -- bin totals -> exclusive scan in LDS, + the rows of the
// earlier workgroups, + its own earlier waves -- exactly as the ApproximateVoxelGrid chain does (kernels_downsample.hpp, round 4).
// No in-kernel hand-off, no co-residency requirement. Two passes over the top 2 x BITS bits of the 27-bit key.
// ------------------------------------------------------------------------------------------------
constexpr int SORT_FUSED_MAX = 262144;
#ifndef FVH_SORT_FUSED_ITEMS
#define FVH_SORT_FUSED_ITEMS 512  // (256 / 512 / 1024 measured: sort stage 47.7 / 44.5 / 51.5 us at 100k points)
#endif
constexpr int SORT_FUSED_ITEMS = FVH_SORT_FUSED_ITEMS, SORT_FUSED_ROUNDS = SORT_FUSED_ITEMS / 64;  // points per wave (2,048 per workgroup)
constexpr int SORT_FUSED_MAX_WGS = SORT_FUSED_MAX / (4 * SORT_FUSED_ITEMS);        // 128 rows at most

template <int BITS>
__global__ __launch_bounds__(256) void radix_hist_fused_kernel(unsigned* __restrict__ keys, int n, int shift, int nwaves, unsigned* __restrict__ hist /* [nwaves][BINS] */,
                                                               unsigned* __restrict__ hist_wg /* [workgroups][BINS] */, const float4* __restrict__ pts = nullptr,
                                                               const unsigned* __restrict__ box = nullptr, int min_inverted = 0) {
  constexpr int BINS = 1 << BITS;
  __shared__ unsigned h[4][BINS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int begin = wave * SORT_FUSED_ITEMS, end = min(n, begin + SORT_FUSED_ITEMS);
  // (a launch of ~200 waves on 256 CUs: one wave's chain of load -> LDS round trips IS the kernel, so all of its keys are fetched first)
  unsigned k[SORT_FUSED_ROUNDS];
  if (wave < nwaves) {
    if (pts) {
      const MortonFrame f = morton_frame(box, min_inverted);
      float4 p[SORT_FUSED_ROUNDS];
#pragma unroll
      for (int u = 0; u < SORT_FUSED_ROUNDS; u++) p[u] = pts[min(begin + u * 64 + lane, end - 1)];
#pragma unroll
      for (int u = 0; u < SORT_FUSED_ROUNDS; u++) {
        k[u] = morton27(f, p[u]);
        const int i = begin + u * 64 + lane;
        if (i < end) keys[i] = k[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < SORT_FUSED_ROUNDS; u++) k[u] = keys[min(begin + u * 64 + lane, end - 1)];
    }
  }
  for (int b = lane; b < BINS; b += 64) h[wv][b] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (wave < nwaves) {
#pragma unroll
    for (int u = 0; u < SORT_FUSED_ROUNDS; u++)
      if (begin + u * 64 + lane < end) atomicAdd(&h[wv][(k[u] >> shift) & (BINS - 1)], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int b = lane; b < BINS; b += 64) hist[(size_t)wave * BINS + b] = h[wv][b];
  }
  __syncthreads();
  for (int b = threadIdx.x; b < BINS; b += 256) hist_wg[(size_t)blockIdx.x * BINS + b] = (h[0][b] + h[1][b]) + (h[2][b] + h[3][b]);
}

template <int BITS>
__global__ __launch_bounds__(256) void radix_scatter_fused_kernel(const unsigned* __restrict__ keys_in, const int* __restrict__ idx_in /* null: the identity (first pass) */, int n, int shift, int nwaves,
                                                                  const unsigned* __restrict__ hist /* [nwaves][BINS] */, const unsigned* __restrict__ hist_wg /* [workgroups][BINS] */,
                                                                  unsigned* __restrict__ keys_out, int* __restrict__ idx_out, const float4* __restrict__ pts, float4* __restrict__ sorted_pts) {
  constexpr int BINS = 1 << BITS;
  constexpr int PER = BINS / 64;       // consecutive bins a lane takes of a row
#ifndef FVH_SORT_FUSED_BATCH_VALUES
#define FVH_SORT_FUSED_BATCH_VALUES 64  // (32 / 64 values per lane in flight: sort stage 47.0 / 44.4 us at 100k points)
#endif
  constexpr int BATCH = FVH_SORT_FUSED_BATCH_VALUES / PER;  // rows a wave keeps in flight
  constexpr int PSTRIDE = 65;          // partial sums of (lane, q) live at q * 65 + lane: conflict-free for the lanes' writes AND the per-bin reads
  auto pidx = [](int b) { return (b % PER) * PSTRIDE + b / PER; };
  static_assert(PER % 4 == 0 && BATCH >= 1, "a lane reads whole 16-byte groups of a row");
  __shared__ unsigned cur[4][BINS];
  __shared__ unsigned s_pt[4][PER * 65], s_pb[4][PER * 65], s_tot[BINS], s_base[BINS], s_w[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, t = threadIdx.x;
  const int wave = blockIdx.x * 4 + wv;
  const int begin = wave * SORT_FUSED_ITEMS, end = min(n, begin + SORT_FUSED_ITEMS);
  // this wave's keys and indices, in flight while the cursors are derived
  unsigned key[SORT_FUSED_ROUNDS];
  int id[SORT_FUSED_ROUNDS];
  if (wave < nwaves) {
#pragma unroll
    for (int u = 0; u < SORT_FUSED_ROUNDS; u++) {
      const int i = min(begin + u * 64 + lane, end - 1);
      key[u] = keys_in[i];
      id[u] = idx_in ? idx_in[i] : i;
    }
  }
  float4 gp[SORT_FUSED_ROUNDS];  // last pass: the points themselves (scattered 16-byte reads), in flight beside the rows below
  if (sorted_pts && wave < nwaves) {
#pragma unroll
    for (int u = 0; u < SORT_FUSED_ROUNDS; u++) { const float4 q = pts[id[u]]; gp[u] = make_float4(q.x, q.y, q.z, __int_as_float(id[u])); }  // (built whole: a later `.w =` pins the array in memory)
  }
  {
    const int nwg = (int)gridDim.x, me = (int)blockIdx.x;
    unsigned tot[PER], bef[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) { tot[q] = 0; bef[q] = 0; }
    for (int g0 = wv; g0 < nwg; g0 += 4 * BATCH) {  // the rows of all workgroups, dealt to the four waves
      uint4 a[BATCH][PER / 4];
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int g = min(g0 + 4 * u, nwg - 1);
        const uint4* row = reinterpret_cast<const uint4*>(hist_wg + (size_t)g * BINS) + lane * (PER / 4);
#pragma unroll
        for (int q = 0; q < PER / 4; q++) a[u][q] = row[q];
      }
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int g = g0 + 4 * u;
        if (g < nwg) {
          const bool before = g < me;
#pragma unroll
          for (int q = 0; q < PER / 4; q++) {
            const unsigned v[4] = {a[u][q].x, a[u][q].y, a[u][q].z, a[u][q].w};
#pragma unroll
            for (int c = 0; c < 4; c++) { tot[4 * q + c] += v[c]; bef[4 * q + c] += before ? v[c] : 0u; }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < PER; q++) { s_pt[wv][q * PSTRIDE + lane] = tot[q]; s_pb[wv][q * PSTRIDE + lane] = bef[q]; }
    __syncthreads();
    for (int b = t; b < BINS; b += 256) { const int j = pidx(b); s_tot[b] = (s_pt[0][j] + s_pt[1][j]) + (s_pt[2][j] + s_pt[3][j]); }
    __syncthreads();
    {  // exclusive scan of the BINS totals: every wave scans a quarter, the quarters meet through s_w
      constexpr int QPER = BINS / 256;  // consecutive totals per thread
      unsigned v[QPER], sum = 0;
#pragma unroll
      for (int q = 0; q < QPER; q++) { v[q] = s_tot[t * QPER + q]; sum += v[q]; }
      unsigned x = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
      if (lane == 63) s_w[wv] = x;
      __syncthreads();
      unsigned run = x - sum;
      for (int w = 0; w < wv; w++) run += s_w[w];
#pragma unroll
      for (int q = 0; q < QPER; q++) { s_base[t * QPER + q] = run; run += v[q]; }
    }
    __syncthreads();
    for (int b = t; b < BINS; b += 256) {
      const int j = pidx(b);
      unsigned c = s_base[b] + ((s_pb[0][j] + s_pb[1][j]) + (s_pb[2][j] + s_pb[3][j]));
#pragma unroll
      for (int w = 0; w < 4; w++) {
        cur[w][b] = c;
        const int gw = me * 4 + w;
        if (gw < nwaves) c += hist[(size_t)gw * BINS + b];
      }
    }
    __syncthreads();
  }
  if (wave >= nwaves) return;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int u = 0; u < SORT_FUSED_ROUNDS; u++) {  // (no early exit: a `break` keeps the loop rolled and sends key / id / gp to memory)
    const bool valid = begin + u * 64 + lane < end;
    const unsigned d = (key[u] >> shift) & (BINS - 1);
    unsigned long long peers = __ballot(valid);
    if (peers) {  // wave-uniform
#pragma unroll
      for (int bit = 0; bit < BITS; bit++) {
        const unsigned long long m = __ballot((d >> bit) & 1);
        peers &= ((d >> bit) & 1) ? m : ~m;
      }
      const int rank = __popcll(peers & lt_mask);
      const int leader = __ffsll((long long)peers) - 1;
      unsigned dst_base = 0;
      if (valid && lane == leader) {
        dst_base = cur[wv][d];
        cur[wv][d] = dst_base + (unsigned)__popcll(peers);
      }
      dst_base = __shfl(dst_base, valid ? leader : 0);
      if (valid) {
        const unsigned dst = dst_base + rank;
        keys_out[dst] = key[u];
        idx_out[dst] = id[u];
        if (sorted_pts) sorted_pts[dst] = gp[u];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Small clouds (n <= SORT_SMALL_MAX): the whole ordering in ONE launch of ONE 1024-thread workgroup.
// The multi-kernel path above costs ~105 us at 17k points -- nine dependent launches of a handful of
// waves each, plus a 512 x n_waves histogram matrix to scan -- where the actual work is a few
// microseconds. Here: bounding box -> 24-bit Morton keys (8 bits/axis) -> three stable 8-bit passes
// with per-wave LDS histograms (16 waves x 256 bins) -> sorted float4 cloud (.w = original index) ->
// boxes of its 64-point tiles -> boxes of 64 tiles. (key, idx) ping-pong through global memory (L2
// resident); all hand-offs are workgroup-local, ordered by __syncthreads().
// ------------------------------------------------------------------------------------------------
constexpr int SORT_SMALL_MAX = 32768;
constexpr int SMALL_BITS = 9, SMALL_BINS = 512, SMALL_WAVES = 16, SMALL_PASSES = 2, SMALL_AXIS_BITS = 6;  // 18-bit Morton: ~1.3 m cells on an 84 m cloud

__device__ __forceinline__ unsigned spread3_8(unsigned v) {  // up to 8 bits -> every third bit
  v &= 0xFF;
  v = (v | (v << 8)) & 0x00F00F;
  v = (v | (v << 4)) & 0x0C30C3;
  v = (v | (v << 2)) & 0x249249;
  return v;
}

// Tail of the single-workgroup fallback (sort_small_kernel with `sorted` given): sorted copy and both box levels
// by the same 1024-thread workgroup.
__device__ inline void sort_small_tail(const float4* __restrict__ pts, const int* order, int n, float4* __restrict__ sorted, float4* bbox1, float4* __restrict__ bbox2) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntiles = (n + 63) >> 6, nsuper = (ntiles + 63) >> 6;
  for (int t = wv; t < ntiles; t += 16) {
    const int j = t * 64 + lane;
    const int src = order[min(j, n - 1)];
    float4 q = pts[src];
    q.w = __int_as_float(src);
    if (j < n) sorted[j] = q;
    float l3[3] = {q.x, q.y, q.z}, h3[3] = {q.x, q.y, q.z};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        l3[a] = fminf(l3[a], __shfl_xor(l3[a], off));
        h3[a] = fmaxf(h3[a], __shfl_xor(h3[a], off));
      }
    if (lane == 0) {
      bbox1[2 * t] = make_float4(l3[0], l3[1], l3[2], 0.f);
      bbox1[2 * t + 1] = make_float4(h3[0], h3[1], h3[2], 0.f);
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int s2 = wv; s2 < nsuper; s2 += 16) {
    const int t = min(s2 * 64 + lane, ntiles - 1);
    const float4 l = bbox1[2 * t], h = bbox1[2 * t + 1];
    float l3[3] = {l.x, l.y, l.z}, h3[3] = {h.x, h.y, h.z};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        l3[a] = fminf(l3[a], __shfl_xor(l3[a], off));
        h3[a] = fmaxf(h3[a], __shfl_xor(h3[a], off));
      }
    if (lane == 0) {
      bbox2[2 * s2] = make_float4(l3[0], l3[1], l3[2], 0.f);
      bbox2[2 * s2 + 1] = make_float4(h3[0], h3[1], h3[2], 0.f);
    }
  }
}

__device__ inline void sort_small_impl(const float4* __restrict__ pts, int n, unsigned* keysA, int* idxA, unsigned* keysB, int* idxB) {
  __shared__ unsigned hist[SMALL_WAVES][SMALL_BINS];  // per-wave digit counts, then per-wave scatter cursors
  __shared__ unsigned bin_total[SMALL_BINS];
  __shared__ float s_lo[SMALL_WAVES][3], s_hi[SMALL_WAVES][3];
  __shared__ float s_box[6];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // ---- bounding cube ----
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  for (int i = tid; i < n; i += 1024) {
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
    if (lane == 0) { s_lo[wv][a] = lo[a]; s_hi[wv][a] = hi[a]; }
  }
  __syncthreads();
  if (tid < 3) {
    float l = s_lo[0][tid], h = s_hi[0][tid];
    for (int w = 1; w < SMALL_WAVES; w++) { l = fminf(l, s_lo[w][tid]); h = fmaxf(h, s_hi[w][tid]); }
    s_box[tid] = l;
    s_box[3 + tid] = h;
  }
  __syncthreads();
  const float lx = s_box[0], ly = s_box[1], lz = s_box[2];
  const float extent = fmaxf(fmaxf(s_box[3] - lx, s_box[4] - ly), fmaxf(s_box[5] - lz, 1e-6f));
  const float qmax = (float)((1 << SMALL_AXIS_BITS) - 1);
  const float scale = (qmax + 0.999f) / extent;

  // each wave owns a contiguous chunk (multiple of 64) -> stable order = (wave, step, lane)
  const int chunk = (((n + SMALL_WAVES - 1) / SMALL_WAVES) + 63) & ~63;
  const int begin = wv * chunk, end = min(n, begin + chunk);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  constexpr int HALF = 16;                                              // 64-item steps held in registers at once
  constexpr int NHALF = SORT_SMALL_MAX / (SMALL_WAVES * 64 * HALF);      // 2
  for (int pass = 0; pass < SMALL_PASSES; pass++) {
    // pass 0 computes keys and writes B, pass 1 reads B and writes A: the final order is always in idxA
    const unsigned* kin = keysB;
    const int* iin = idxB;
    unsigned* kout = (pass == SMALL_PASSES - 1) ? keysA : keysB;
    int* iout = (pass == SMALL_PASSES - 1) ? idxA : idxB;
    const int shift = pass * SMALL_BITS;
    // A half chunk (16 steps x 64 items) goes to registers at once: 16 independent loads in flight instead of
    // one dependent L2 round trip per step (that chain, not the arithmetic, was the first version's 35 us/pass).
    auto load_half = [&](int h, unsigned (&key)[HALF], int (&id)[HALF]) {
#pragma unroll
      for (int u = 0; u < HALF; u++) {
        const int i = begin + (h * HALF + u) * 64 + lane;
        key[u] = 0xFFFFFFFFu;
        id[u] = -1;
        if (i < end) {
          if (pass == 0) {
            const float4 p = pts[i];
            const unsigned ix = (unsigned)fminf(qmax, fmaxf(0.f, (p.x - lx) * scale));
            const unsigned iy = (unsigned)fminf(qmax, fmaxf(0.f, (p.y - ly) * scale));
            const unsigned iz = (unsigned)fminf(qmax, fmaxf(0.f, (p.z - lz) * scale));
            key[u] = spread3_8(ix) | (spread3_8(iy) << 1) | (spread3_8(iz) << 2);
            id[u] = i;
          } else {
            key[u] = kin[i];
            id[u] = iin[i];
          }
        }
      }
    };
    for (int b = lane; b < SMALL_BINS; b += 64) hist[wv][b] = 0;
    __syncthreads();
#pragma unroll 1
    for (int h = 0; h < NHALF; h++) {
      if (begin + h * HALF * 64 >= end) break;
      unsigned key[HALF];
      int id[HALF];
      load_half(h, key, id);
#pragma unroll
      for (int u = 0; u < HALF; u++)
        if (begin + (h * HALF + u) * 64 + lane < end) atomicAdd(&hist[wv][(key[u] >> shift) & (SMALL_BINS - 1)], 1u);
    }
    __syncthreads();
    if (tid < SMALL_BINS) {  // per bin: exclusive prefix over the waves, and the bin total
      unsigned run = 0;
      for (int w = 0; w < SMALL_WAVES; w++) { const unsigned c = hist[w][tid]; hist[w][tid] = run; run += c; }
      bin_total[tid] = run;
    }
    __syncthreads();
    if (wv == 0) {  // exclusive scan of the bin totals: BINS/64 per lane
      constexpr int PER = SMALL_BINS / 64;
      unsigned v[PER], sum = 0;
#pragma unroll
      for (int j = 0; j < PER; j++) { v[j] = bin_total[lane * PER + j]; sum += v[j]; }
      unsigned x = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
      unsigned run = x - sum;
#pragma unroll
      for (int j = 0; j < PER; j++) { bin_total[lane * PER + j] = run; run += v[j]; }
    }
    __syncthreads();
    for (int b = lane; b < SMALL_BINS; b += 64) hist[wv][b] += bin_total[b];  // this wave's scatter cursors
    // (own row only: LDS operations of one wave complete in order, no barrier needed)
#pragma unroll 1
    for (int h = 0; h < NHALF; h++) {
      if (begin + h * HALF * 64 >= end) break;
      unsigned key[HALF];
      int id[HALF];
      load_half(h, key, id);
#pragma unroll
      for (int u = 0; u < HALF; u++) {
        const int i0 = begin + (h * HALF + u) * 64;
        if (i0 >= end) break;  // wave-uniform
        const bool valid = (i0 + lane) < end;
        const unsigned d = (key[u] >> shift) & (SMALL_BINS - 1);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < SMALL_BITS; bit++) {
          const unsigned long long m = __ballot((d >> bit) & 1);
          peers &= ((d >> bit) & 1) ? m : ~m;
        }
        const int rank = __popcll(peers & lt_mask);
        const int leader = __ffsll((long long)peers) - 1;
        unsigned dst_base = 0;
        if (valid && lane == leader) {
          dst_base = hist[wv][d];
          hist[wv][d] = dst_base + (unsigned)__popcll(peers);
        }
        dst_base = __shfl(dst_base, leader);
        if (valid) {
          kout[dst_base + rank] = key[u];
          iout[dst_base + rank] = id[u];
        }
      }
    }
    __syncthreads();  // orders this pass's global writes before the next pass's reads (same workgroup)
  }
}

__global__ __launch_bounds__(1024) void sort_small_kernel(const float4* __restrict__ pts, int n, unsigned* keysA, int* idxA, unsigned* keysB, int* idxB) {
  sort_small_impl(pts, n, keysA, idxA, keysB, idxB);
}

// ------------------------------------------------------------------------------------------------
// Small clouds, cooperative version: the same 18-bit Morton order as sort_small_kernel (identical output), but on
// COOP_WGS co-resident workgroups instead of one workgroup doing everything (62 us at 17k points on one CU, + 10 us for the
// tile / super boxes). Phases: keys + digit-0 histograms | scatter 0 | digit-1 histograms | scatter 1 + gather (bounding cube:
// pack_points_kernel before; tile and super boxes: sort_coop_finish_kernel after -- a kernel boundary costs what a hand-off costs, ~3 us,
// and that kernel exists anyway: it is the fallback). What one phase hands to the next crosses workgroups through tagged words the
// consumer polls (see sort_coop_kernel) -- no grid barrier. (Round 5 also built the widening of the caller's array + the bounding cube
// INTO this kernel, for one more hand-off: 22.4 us against pack 4.8 + sort 19.4, and the registration rate did not move. Not kept.)
// Each wave owns a contiguous chunk of <= 128 keys (2 steps of 64), so stable order = (workgroup, wave, step, lane).
// The per-pass scan is done redundantly by every workgroup from the 512 x COOP_WGS matrix of workgroup totals.
// A poll that never succeeds (workgroups not co-resident) trips a watchdog: not every workgroup reports "done" and
// sort_coop_finish_kernel, launched right behind, does the whole job on one workgroup instead.
// ------------------------------------------------------------------------------------------------
constexpr int COOP_WGS = 32, COOP_THREADS = 512, COOP_WAVES = COOP_THREADS / 64, COOP_STEPS = 2;
static_assert(COOP_WGS * COOP_WAVES * COOP_STEPS * 64 >= SORT_SMALL_MAX, "every key needs a slot");
static_assert(COOP_THREADS == SMALL_BINS, "one thread per bin in the scans");
static_assert(COOP_THREADS == COOP_WGS * 16 && SMALL_BINS == 16 * 32, "a thread polls 32 consecutive words of one row of the matrix");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct SortCoopState {     // zeroed once, when allocated. Every word is compared with the launch's sequence number (>= 1): nothing is ever reset
  unsigned abort_seq;              // == seq: a workgroup of launch `seq` gave up (watchdog, or the test hook)
  unsigned pad[15];
  unsigned done_seq[COOP_WGS];     // [w] == seq: workgroup w of launch `seq` ran to the end; all of them = the cooperative kernel did the whole job
};

template <typename T> __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // write-through (sc1)
template <typename T> __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // bypasses the non-coherent L2

// Hand-offs between workgroups carry their own arrival signal (round 3, as in the LM kernel): a histogram entry is
// {count:11, tag:21}, a scattered element {tag:31, key:18, index:15} in ONE 64-bit word, an entry of the final order {tag:32,
// index:32} -- tag = this launch's sequence number (+ the pass). A consumer polls the words it needs with L2-bypassing loads
// until their tags are current; nobody waits for anybody else: no arrival counter, no "wait for my stores, add 1, poll the
// counter" (four grid barriers of ~3 us each in the first version of this kernel). Buffers are zeroed
// when allocated, sequence numbers start at 1 and every complete launch rewrites every word it will poll next time.
// What a hand-off costs (round 5, tools/sort_timing.py + rocprofv3 A/B, tools/ab_sort.sh): an sc1 load past the caches is ~1 us for one
// word per lane and ~2 us for the 64 KB of a pass's histogram matrix per workgroup; a store is visible ~1 us after it was issued. Two things
// were slower than that and are gone: (a) the matrix as [bin][workgroup] -- a workgroup published 512 scattered 4-byte words -- read by 32
// separate agent-scope loads per thread, ten of which the compiler serialised behind their own s_waitcnt (now [workgroup][bin]: 2 KB of whole
// lines per workgroup, read as eight 16-byte loads behind ONE wait and transposed through LDS); (b) a failed poll that consulted the watchdog
// -- a clock read and another load past the caches -- before every retry (now every 16th). 27.5 -> 22.2 us at 17k points. With cheap
// retries the delay before the first look no longer matters (0.1 ... 0.9 us measure the same; polling with no delay at all costs 3 us).
#ifndef FVH_COOP_POLL_SLEEP
#define FVH_COOP_POLL_SLEEP 8
#endif
#ifndef FVH_COOP_FIRST_SLEEP
#define FVH_COOP_FIRST_SLEEP 8
#endif
constexpr int COOP_MATRIX_WORDS = 2 * SMALL_BINS * COOP_WGS;                                    // u32 {count, tag}
constexpr int COOP_HTAG_BITS = 21;
constexpr unsigned COOP_HTAG_MASK = (1u << COOP_HTAG_BITS) - 1u;
static_assert(COOP_THREADS * COOP_STEPS < (1 << (32 - COOP_HTAG_BITS)), "a workgroup's count of one bin must fit the bits above the tag");
constexpr size_t COOP_ELEM_OFFSET = sizeof(SortCoopState) + sizeof(unsigned) * COOP_MATRIX_WORDS;  // u64 x SORT_SMALL_MAX: pass-0 output
constexpr size_t COOP_STATE_BYTES = COOP_ELEM_OFFSET + sizeof(unsigned long long) * SORT_SMALL_MAX;
static_assert(sizeof(SortCoopState) % 16 == 0, "the matrix behind it is read in 16-byte words");
static_assert(SORT_SMALL_MAX <= (1 << 15) && SMALL_BITS * SMALL_PASSES <= 18, "element packing: 15 index bits, 18 key bits");

// eight 16-byte agent-scope loads of 128 consecutive bytes in flight together, one wait (loads and wait in one asm block: nothing touches a
// destination register before its data has landed)
__device__ __forceinline__ void load8x16_agent(u32x4 (&v)[8], const u32x4* p) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
      "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
      "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
      "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
      "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p)
      : "memory");
}

__device__ __forceinline__ bool coop_timed_out(SortCoopState* st, unsigned seq, unsigned long long t0, unsigned long long watchdog_ticks) {
  if (__hip_atomic_load(&st->abort_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq || wall_clock64() - t0 > watchdog_ticks) {
    __hip_atomic_store(&st->abort_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
#ifdef FVH_SORT_TIMING  // debug build only (tools/sort_timing.py): 100 MHz wall-clock stamps of every workgroup's walk through the phases
__device__ unsigned long long g_sort_time[COOP_WGS][16];
#define FVH_ST(k) do { if (threadIdx.x == 0) g_sort_time[wg][k] = wall_clock64(); } while (0)
#else
#define FVH_ST(k) do { } while (0)
#endif

__global__ __launch_bounds__(COOP_THREADS) void sort_coop_kernel(const float4* __restrict__ pts, int n, int* order, float4* sorted,
                                                                const unsigned* __restrict__ box /* {~ordered(min) x3, ordered(max) x3} */, unsigned* hist /* [2][COOP_WGS][SMALL_BINS] */,
                                                                unsigned long long* elem, SortCoopState* st, unsigned seq, unsigned long long watchdog_ticks) {
  __shared__ unsigned wh[COOP_WAVES][SMALL_BINS];  // per-wave digit counts -> exclusive prefix over the waves of this workgroup -> scatter cursors
  __shared__ unsigned wsum[COOP_WAVES];
  __shared__ unsigned short s_cnt[COOP_WGS][SMALL_BINS];  // the pass's matrix of workgroup totals (counts <= 1,024), on its way from "thread = 32 words of a row" to "thread = bin"
  const int wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int gw = wg * COOP_WAVES + wv;  // global wave index
  const int chunk = ((((n + COOP_WGS * COOP_WAVES - 1) / (COOP_WGS * COOP_WAVES)) + 63) & ~63);
  const int begin = gw * chunk, end = min(n, begin + chunk);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const unsigned long long t_start = wall_clock64();
  const unsigned long long etag = (unsigned long long)(seq & 0x7fffffffu);
  unsigned spins = 0;  // failed polls: the watchdog (a clock read + a load past the caches) is consulted every 16th
  FVH_ST(0);
  if (watchdog_ticks == 0) {  // test hook: the fallback does the whole job
    if (tid == 0) __hip_atomic_store(&st->abort_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  float4 p[COOP_STEPS];
#pragma unroll
  for (int u = 0; u < COOP_STEPS; u++) {
    const int i = begin + u * 64 + lane;
    p[u] = (i < end) ? pts[i] : make_float4(0, 0, 0, 0);
  }
  unsigned bx[6];
#pragma unroll
  for (int a = 0; a < 6; a++) bx[a] = box[a];  // (the bounding cube was reduced by pack_points_kernel when the cloud was set)

  // ---- keys (the arithmetic of sort_small_kernel) ----
  const float lx = ordered_to_float(~bx[0]), ly = ordered_to_float(~bx[1]), lz = ordered_to_float(~bx[2]);
  const float hx = ordered_to_float(bx[3]), hy = ordered_to_float(bx[4]), hz = ordered_to_float(bx[5]);
  const float extent = fmaxf(fmaxf(hx - lx, hy - ly), fmaxf(hz - lz, 1e-6f));
  const float qmax = (float)((1 << SMALL_AXIS_BITS) - 1);
  const float scale = (qmax + 0.999f) / extent;
  unsigned key[COOP_STEPS];
  int id[COOP_STEPS];
#pragma unroll
  for (int u = 0; u < COOP_STEPS; u++) {
    const int i = begin + u * 64 + lane;
    const unsigned ix = (unsigned)fminf(qmax, fmaxf(0.f, (p[u].x - lx) * scale));
    const unsigned iy = (unsigned)fminf(qmax, fmaxf(0.f, (p[u].y - ly) * scale));
    const unsigned iz = (unsigned)fminf(qmax, fmaxf(0.f, (p[u].z - lz) * scale));
    key[u] = (i < end) ? (spread3_8(ix) | (spread3_8(iy) << 1) | (spread3_8(iz) << 2)) : 0xFFFFFFFFu;
    id[u] = (i < end) ? i : -1;
  }

  FVH_ST(1);  // keys computed
  for (int pass = 0; pass < SMALL_PASSES; pass++) {
    const int shift = pass * SMALL_BITS;
    unsigned* gh = hist + (size_t)pass * SMALL_BINS * COOP_WGS;
    // {count:11, tag:21}: a workgroup holds at most 1,024 keys, so 11 bits carry its count and the tag wraps every 2^20 sorts of an engine
    // (16 tag bits wrapped every 32,768: a word a watchdog-aborted launch never rewrote could have matched a later launch)
    const unsigned htag = ((seq << 1) | (unsigned)pass) & COOP_HTAG_MASK;
    int failed = 0;
    if (pass > 0) {  // this wave's chunk in the order pass 0 produced: poll until every element of the chunk has landed
      if (FVH_COOP_FIRST_SLEEP) __builtin_amdgcn_s_sleep(FVH_COOP_FIRST_SLEEP);
#pragma unroll
      for (int u = 0; u < COOP_STEPS; u++) {
        const int i = begin + u * 64 + lane;
        unsigned long long v = 0;
        if (i < end) {
          while (true) {
            v = ld_agent(&elem[i]);
            if ((v >> 33) == etag) break;
            if ((++spins & 15u) == 0u && coop_timed_out(st, seq, t_start, watchdog_ticks)) { failed = 1; break; }
            __builtin_amdgcn_s_sleep(FVH_COOP_POLL_SLEEP);
          }
        }
        key[u] = (i < end) ? (unsigned)(v >> 15) & 0x3FFFFu : 0xFFFFFFFFu;
        id[u] = (i < end) ? (int)(v & 0x7FFFu) : -1;
      }
    }
    FVH_ST(2 + 5 * pass);  // (pass 1: this workgroup's elements of pass 0 have landed)
    // per-wave digit histogram
    for (int b = lane; b < SMALL_BINS; b += 64) wh[wv][b] = 0;
#pragma unroll
    for (int u = 0; u < COOP_STEPS; u++)
      if (begin + u * 64 + lane < end) atomicAdd(&wh[wv][(key[u] >> shift) & (SMALL_BINS - 1)], 1u);
    if (__syncthreads_or(failed)) return;
    {  // thread = bin: exclusive prefix over this workgroup's waves; workgroup total -> global, tagged
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < COOP_WAVES; w++) { const unsigned c = wh[w][tid]; wh[w][tid] = run; run += c; }
      st_agent(&gh[(size_t)wg * SMALL_BINS + tid], (run << COOP_HTAG_BITS) | htag);  // [workgroup][bin]: a workgroup publishes 2 KB of whole lines
    }
    FVH_ST(3 + 5 * pass);  // workgroup totals published
    {  // thread = bin: total over all workgroups and the part before this workgroup (polled until all 32 entries are current); then the bins are scanned
      // The matrix is [workgroup][bin]: thread t takes 32 consecutive words of row t / 16 -- eight 16-byte loads in flight together, ONE wait
      // (32 separate agent-scope loads were compiled into 22 batched + 10 serialised round trips: 3.5 of this kernel's 26 us per pass) --
      // checks their tags, and the counts cross to "thread = bin" through LDS.
      const u32x4* src = reinterpret_cast<const u32x4*>(gh + (size_t)(tid >> 4) * SMALL_BINS + (size_t)(tid & 15) * 32);
      u32x4 q[8];
      if (FVH_COOP_FIRST_SLEEP) __builtin_amdgcn_s_sleep(FVH_COOP_FIRST_SLEEP);
      while (true) {
        load8x16_agent(q, src);
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) bad |= ((q[j].x & COOP_HTAG_MASK) ^ htag) | ((q[j].y & COOP_HTAG_MASK) ^ htag) | ((q[j].z & COOP_HTAG_MASK) ^ htag) | ((q[j].w & COOP_HTAG_MASK) ^ htag);
        if (!bad) break;
        if ((++spins & 15u) == 0u && coop_timed_out(st, seq, t_start, watchdog_ticks)) { failed = 1; break; }
        __builtin_amdgcn_s_sleep(FVH_COOP_POLL_SLEEP);
      }
      FVH_ST(4 + 5 * pass);  // this thread's part of the matrix is current
      {
        uint2* dst = reinterpret_cast<uint2*>(&s_cnt[tid >> 4][(tid & 15) * 32]);
#pragma unroll
        for (int j = 0; j < 8; j++)
          dst[j] = make_uint2((q[j].x >> COOP_HTAG_BITS) | ((q[j].y >> COOP_HTAG_BITS) << 16), (q[j].z >> COOP_HTAG_BITS) | ((q[j].w >> COOP_HTAG_BITS) << 16));
      }
      if (__syncthreads_or(failed)) return;
#ifdef FVH_SORT_TIMING
      if (pass == 0) { FVH_ST(14); if (threadIdx.x == 0) g_sort_time[wg][15] = spins; }
#endif
      unsigned total = 0, before = 0;
#pragma unroll
      for (int j = 0; j < COOP_WGS; j++) {
        const unsigned a = s_cnt[j][tid];
        total += a;
        before += (j < wg) ? a : 0u;
      }
      unsigned x = total;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
      if (lane == 63) wsum[wv] = x;
      if (__syncthreads_or(failed)) return;
      unsigned base = x - total + before;
      for (int w = 0; w < wv; w++) base += wsum[w];
#pragma unroll
      for (int w = 0; w < COOP_WAVES; w++) wh[w][tid] += base;  // scatter cursor of wave w for this bin
    }
    __syncthreads();
    FVH_ST(5 + 5 * pass);  // scatter cursors ready
    const bool last = (pass == SMALL_PASSES - 1);
#pragma unroll
    for (int u = 0; u < COOP_STEPS; u++) {
      const int i0 = begin + u * 64;
      if (i0 >= end) break;  // wave-uniform
      const bool valid = (i0 + lane) < end;
      const unsigned d = (key[u] >> shift) & (SMALL_BINS - 1);
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < SMALL_BITS; bit++) {
        const unsigned long long m = __ballot((d >> bit) & 1);
        peers &= ((d >> bit) & 1) ? m : ~m;
      }
      const int rank = __popcll(peers & lt_mask);
      const int leader = __ffsll((long long)peers) - 1;
      unsigned dst_base = 0;
      if (valid && lane == leader) {
        dst_base = wh[wv][d];
        wh[wv][d] = dst_base + (unsigned)__popcll(peers);
      }
      dst_base = __shfl(dst_base, leader);
      if (valid) {
        const unsigned dst = dst_base + rank;
        if (!last) {
          st_agent(&elem[dst], (etag << 33) | ((unsigned long long)key[u] << 15) | (unsigned long long)(unsigned)id[u]);
        } else {
          order[dst] = id[u];
          float4 q = pts[id[u]];
          q.w = __int_as_float(id[u]);
          sorted[dst] = q;
        }
      }
    }
    FVH_ST(6 + 5 * pass);  // scatter issued
  }

  __syncthreads();
  FVH_ST(13);
  if (tid == 0) st->done_seq[wg] = seq;  // (read by the finish kernel, behind the kernel boundary)
}

// Runs right behind sort_coop_kernel: the boxes of the 64-point tiles from the sorted copy (coalesced; inside the cooperative kernel they cost
// one more hand-off) by the first `tile_wgs` workgroups, one tile per wave, and the super boxes (64 tiles = 4,096 points) straight from the
// points by one workgroup each -- min / max are exact, so "from the points" equals "from the tile boxes", and no workgroup waits for another.
// (One workgroup per super tile doing its 64 tiles took 8.6 us: 144 dependent wave shuffles per wave through ONE CU's LDS crossbar.)
// When the cooperative kernel did not finish (watchdog), workgroup 0 does the whole job: same order, sorted copy and both box levels.
__global__ __launch_bounds__(1024) void sort_coop_finish_kernel(const float4* __restrict__ pts, int n, unsigned* keysA, int* order, unsigned* keysB, int* idxB, float4* sorted,
                                                                float4* bbox1, float4* bbox2, const SortCoopState* __restrict__ st, unsigned* box, unsigned seq, int tile_wgs) {
  __shared__ int s_redo;
  __shared__ float s_wl[16][3], s_wh[16][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv == 0) {  // (every workgroup reads the same words; nobody writes them here)
    const bool bad = lane < COOP_WGS && st->done_seq[lane] != seq;
    const unsigned long long any = __ballot(bad);
    if (lane == 0) s_redo = (any != 0ull || st->abort_seq == seq) ? 1 : 0;
  }
  __syncthreads();
  // the cloud's bounding-cube accumulators (pack_points_kernel's atomics) cleared for the next cloud: saves a memset on the stream
  if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 70) box[threadIdx.x - 64] = 0u;
  if (s_redo) {
    if (blockIdx.x != 0) return;
    sort_small_impl(pts, n, keysA, order, keysB, idxB);
    sort_small_tail(pts, order, n, sorted, bbox1, bbox2);
    return;
  }
  const int ntiles = (n + 63) >> 6;
  const bool super = (int)blockIdx.x >= tile_wgs;
  float l3[3], h3[3];
  if (!super) {  // one tile per wave
    const int t = blockIdx.x * 16 + wv;
    if (t >= ntiles) return;  // (wave-uniform; no barrier below on this side)
    const float4 q = sorted[min(t * 64 + lane, n - 1)];
    l3[0] = h3[0] = q.x; l3[1] = h3[1] = q.y; l3[2] = h3[2] = q.z;
  } else {       // 4,096 points per workgroup, four per thread (positions past the end repeat the last point, as the tiles past the end repeat the last tile)
    const int s2 = blockIdx.x - tile_wgs;
    float4 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = sorted[min(s2 * 4096 + u * 1024 + (int)threadIdx.x, n - 1)];
    l3[0] = fminf(fminf(q[0].x, q[1].x), fminf(q[2].x, q[3].x)); h3[0] = fmaxf(fmaxf(q[0].x, q[1].x), fmaxf(q[2].x, q[3].x));
    l3[1] = fminf(fminf(q[0].y, q[1].y), fminf(q[2].y, q[3].y)); h3[1] = fmaxf(fmaxf(q[0].y, q[1].y), fmaxf(q[2].y, q[3].y));
    l3[2] = fminf(fminf(q[0].z, q[1].z), fminf(q[2].z, q[3].z)); h3[2] = fmaxf(fmaxf(q[0].z, q[1].z), fmaxf(q[2].z, q[3].z));
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      l3[a] = fminf(l3[a], __shfl_xor(l3[a], off));
      h3[a] = fmaxf(h3[a], __shfl_xor(h3[a], off));
    }
  if (!super) {
    if (lane == 0) {
      const int t = blockIdx.x * 16 + wv;
      bbox1[2 * t] = make_float4(l3[0], l3[1], l3[2], 0.f);
      bbox1[2 * t + 1] = make_float4(h3[0], h3[1], h3[2], 0.f);
    }
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { s_wl[wv][a] = l3[a]; s_wh[wv][a] = h3[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      l[a] = s_wl[0][a]; h[a] = s_wh[0][a];
      for (int w = 1; w < 16; w++) { l[a] = fminf(l[a], s_wl[w][a]); h[a] = fmaxf(h[a], s_wh[w][a]); }
    }
    const int s2 = blockIdx.x - tile_wgs;
    bbox2[2 * s2] = make_float4(l[0], l[1], l[2], 0.f);
    bbox2[2 * s2 + 1] = make_float4(h[0], h[1], h[2], 0.f);
  }
}

// One wave per 64-point tile: gather the tile's points in sorted order (.w = original index) and
// box them. Runs on the whole GPU (N/64 independent waves) right after the single-workgroup sort.
__global__ __launch_bounds__(256) void gather_tiles_kernel(const float4* __restrict__ pts, const int* __restrict__ order, int n, float4* __restrict__ sorted,
                                                           float4* __restrict__ bbox1) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t * 64 >= n) return;
  const int j = t * 64 + lane;
  const int src = order[min(j, n - 1)];
  float4 p = pts[src];
  p.w = __int_as_float(src);
  if (j < n) sorted[j] = p;
  float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if (lane == 0) {
    bbox1[2 * t] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bbox1[2 * t + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

}  // namespace fvh
