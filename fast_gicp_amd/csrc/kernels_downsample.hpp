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
constexpr int EMIT_BATCH = 8;
template <bool APPROX>
__global__ __launch_bounds__(256) void vg_emit_kernel(const unsigned* __restrict__ keys, const int* __restrict__ idx, const float4* __restrict__ pts, int n,
                                                      const unsigned* __restrict__ head, const unsigned* __restrict__ pos_scan /* exact: scan(head) by j; approx: scan(trig) by original index */,
                                                      const unsigned* __restrict__ slot_rank /* approx: exclusive scan of slot_used, [AVG_SLOTS] = #used */,
                                                      const unsigned* __restrict__ trig_total /* approx: number of flush-triggering points */, float* __restrict__ out,
                                                      const unsigned* __restrict__ bad, unsigned* __restrict__ result /* approx: {trigger count, used slots, bad} for ONE host copy */) {
  if (APPROX && result && blockIdx.x == 0 && threadIdx.x == 0) { result[0] = *trig_total; result[1] = slot_rank[AVG_SLOTS]; result[2] = *bad; }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || !head[j]) return;
  // The sum is sequential by definition (fp32, input order: the arithmetic of the PCL filter) but its LOADS are not: the next
  // EMIT_BATCH indices, head flags and points are requested together, then consumed in order -- a run of r points costs
  // ~2 r / EMIT_BATCH dependent round trips instead of 2 r (runs of 15-100 points made this the longest kernel of the filter).
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int e = j;
  float4 p = pts[idx[j]];
  bool more = true;
  while (more) {
    int ib[EMIT_BATCH];
    unsigned hb[EMIT_BATCH];
    float4 pb[EMIT_BATCH];
#pragma unroll
    for (int u = 0; u < EMIT_BATCH; u++) { const int k = e + 1 + u; hb[u] = (k < n) ? head[k] : 1u; ib[u] = idx[min(k, n - 1)]; }
#pragma unroll
    for (int u = 0; u < EMIT_BATCH; u++) pb[u] = pts[ib[u]];
#pragma unroll
    for (int u = 0; u < EMIT_BATCH; u++) {
      if (more) {
        cx += p.x; cy += p.y; cz += p.z;
        e++;
        if (hb[u]) more = false; else p = pb[u];
      }
    }
  }
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

// =====================================================================================================================
// pcl::ApproximateVoxelGrid, fused chain (round 2): SIX launches, no memset, no key / index arrays, the count goes to the
// host through mapped memory. Round 1 ran the same decomposition as 9 kernels + 2 memsets + a D2H copy (87 us per 118k-point
// frame, a third of the LiDAR-stream step):
//   avg_keys_hist    slot of every point (straight from the caller's xyz, any stride) + per-wave slot histograms; clears the
//                    small state of the later kernels
//   avg_binscan      per-slot prefix over the waves; the last workgroup also scans the 512 slot totals
//   avg_scatter      stable counting sort by slot: writes the POINTS in slot order (.w = original index) -- nothing else
//   avg_mark         run heads inside a slot (voxel of a point != voxel of its predecessor); a head that is not a slot start is
//                    the point whose arrival flushes the previous run: one bit in `trigbits` at its ORIGINAL index
//   avg_scan         one workgroup: prefix of the trigger bits per 1024 indices, rank of the used slots, the output count
//   avg_emit         one thread per run: centroid (fp32 sums in input order, as PCL), output position = rank of the flushing
//                    point among all triggers (block prefix + popcounts), or behind them in slot order for the final flushes
// Output points and order are bit-identical to the oracle's restatement of the PCL filter (tests/test_gpu_downsample.py).
// =====================================================================================================================
#ifndef FVH_AVG_ITEMS
#define FVH_AVG_ITEMS 512
#endif
constexpr int AVG_ITEMS = FVH_AVG_ITEMS;  // points per wave in the counting sort (512: the fused chain's scatter re-reads one histogram row per WORKGROUP of the sort -- 58 rows at 118k points instead of 116; measured 256 / 512 / 1024: chain 43.4 / 40.4 / 44.7 us)

struct AvgState {            // small device state, cleared by avg_keys_hist for the NEXT stages of the same call
  unsigned slot_used[AVG_SLOTS];
  unsigned slot_rank[AVG_SLOTS + 1];
  unsigned bin_base[AVG_SLOTS];
  unsigned ticket, bad, trig_total, used_slots, emit_ticket;
};

__device__ __forceinline__ unsigned avg_slot(int ix, int iy, int iz) { return ((unsigned)ix * 7171u + (unsigned)iy * 3079u + (unsigned)iz * 4231u) & (AVG_SLOTS - 1); }
__device__ __forceinline__ float4 load_xyz(const float* __restrict__ xyz, int i, int stride) {
  const float* q = xyz + (size_t)i * stride;
  return make_float4(q[0], q[1], q[2], 0.f);
}

// Frames of up to AVG_FUSED_MAX_POINTS points (a LiDAR frame is 118k) take a FOUR-launch chain: the slot x wave prefix of the counting sort
// and the prefix of the trigger bits are small enough to be recomputed by every workgroup that needs them (from 237 KB of per-workgroup
// histograms, 15 KB of trigger words), which removes the two scan launches -- each a global dependency of ~5-9 us for ~1 us of work --
// without any in-kernel hand-off (round 2's six launches: 47 us per 118k-point frame). Larger inputs keep the scan kernels.
constexpr int AVG_ROUNDS = AVG_ITEMS / 64;
constexpr int AVG_FUSED_MAX_POINTS = 262144;
constexpr int AVG_FUSED_MAX_WGS = AVG_FUSED_MAX_POINTS / (4 * AVG_ITEMS);      // workgroups of the counting sort (4 waves x 256 points)
constexpr int AVG_FUSED_MAX_BLOCKS = AVG_FUSED_MAX_POINTS / 1024;              // trigger-bit blocks of 1,024 indices

__global__ __launch_bounds__(256) void avg_keys_hist_kernel(const float* __restrict__ xyz, int n, int stride, float inv, int nwaves, unsigned* __restrict__ hist /* [slots][nwaves], or (fused) [nwaves][slots] */,
                                                            unsigned* __restrict__ trigbits, int nwords, AvgState* __restrict__ st, unsigned* __restrict__ hist_wg = nullptr /* fused chain: [workgroups][slots] */) {
  __shared__ unsigned h[4][AVG_SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  for (int w = blockIdx.x * 256 + threadIdx.x; w < ((nwords + 31) & ~31); w += gridDim.x * 256) trigbits[w] = 0u;  // (whole blocks of 32 words: the buffer is sized so)
  if (blockIdx.x == 0) {
    for (int s = threadIdx.x; s < AVG_SLOTS; s += 256) st->slot_used[s] = 0u;
    // (st->bad is NOT reset here: other workgroups of this very launch raise it with an atomic, and nothing orders a plain store
    // of workgroup 0 against them -- the flag is cleared by whoever consumed it: the last workgroup of avg_emit_kernel, or the host)
  }
  // (the launch is ~230 waves on 256 CUs: one wave's chain of load -> LDS round trips IS the kernel, so all of its points are fetched first)
  float4 pts[AVG_ROUNDS];
  const int begin = wave * AVG_ITEMS, end = min(n, begin + AVG_ITEMS);
  if (wave < nwaves) {
#pragma unroll
    for (int u = 0; u < AVG_ROUNDS; u++) { const int i = begin + u * 64 + lane; pts[u] = load_xyz(xyz, min(i, end - 1), stride); }
  }
  for (int b = lane; b < AVG_SLOTS; b += 64) h[wv][b] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if (wave < nwaves) {
    bool bad = false;
#pragma unroll
    for (int u = 0; u < AVG_ROUNDS; u++) {
      const int i = begin + u * 64 + lane;
      if (i < end) {
        const float4 p = pts[u];
        bad |= !finite3(p);
        int ix, iy, iz;
        avg_voxel(p, inv, ix, iy, iz);
        atomicAdd(&h[wv][avg_slot(ix, iy, iz)], 1u);
      }
    }
    if (__any(bad) && lane == 0) atomicOr(&st->bad, 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (hist_wg) { for (int b = lane; b < AVG_SLOTS; b += 64) hist[(size_t)wave * AVG_SLOTS + b] = h[wv][b]; }
    else { for (int b = lane; b < AVG_SLOTS; b += 64) hist[(size_t)b * nwaves + wave] = h[wv][b]; }
  }
  if (hist_wg) {  // this workgroup's row: the sum over its four waves
    __syncthreads();
    for (int b = threadIdx.x; b < AVG_SLOTS; b += 256) hist_wg[(size_t)blockIdx.x * AVG_SLOTS + b] = (h[0][b] + h[1][b]) + (h[2][b] + h[3][b]);
  }
}

// one wave per slot: exclusive prefix over the waves (radix_binscan_kernel), slot total; the LAST workgroup scans the totals
__global__ __launch_bounds__(256) void avg_binscan_kernel(unsigned* __restrict__ hist, int nwaves, unsigned* __restrict__ totals /* [slots] scratch */, AvgState* __restrict__ st) {
  __shared__ int s_last;
  __shared__ unsigned s_tot[AVG_SLOTS];
  const int lane = threadIdx.x & 63;
  const int bin = blockIdx.x * 4 + (threadIdx.x >> 6);
  {
    unsigned* row = hist + (size_t)bin * nwaves;
    unsigned run = 0;
    for (int base = 0; base < nwaves; base += 64) {
      const int w = base + lane;
      const unsigned v = (w < nwaves) ? row[w] : 0u;
      unsigned x = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
      if (w < nwaves) row[w] = run + x - v;
      run += __shfl(x, 63);
    }
    if (lane == 0) __hip_atomic_store(&totals[bin], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (no __threadfence(): an agent-scope release writes back the whole L2 of the XCD, once per WORKGROUP -- that was 12 of this
  // kernel's 17 us. The only values another workgroup of this launch reads are the totals, stored write-through above; waiting
  // for the store is all the release they need.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(&st->ticket, 1u) == gridDim.x - 1);
  __syncthreads();
  if (!s_last) return;
  for (int b = threadIdx.x; b < AVG_SLOTS; b += 256) s_tot[b] = __hip_atomic_load(&totals[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x < 64) {  // 512 totals: 8 per lane
    unsigned v[8], sum = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) { v[u] = s_tot[threadIdx.x * 8 + u]; sum += v[u]; }
    unsigned x = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    unsigned run = x - sum;
#pragma unroll
    for (int u = 0; u < 8; u++) { st->bin_base[threadIdx.x * 8 + u] = run; run += v[u]; }
    if (threadIdx.x == 0) st->ticket = 0u;  // re-armed for the next call
  }
}

// stable scatter by slot: the points themselves, in slot order, original index in .w
// `hist_wg` != null: the fused chain -- offsets = per-wave counts [nwaves][slots]; this workgroup derives its scatter cursors itself:
// slot base (exclusive scan of the slot totals) + the counts of all earlier workgroups + those of its own earlier waves.
__global__ __launch_bounds__(256) void avg_scatter_kernel(const float* __restrict__ xyz, int n, int stride, float inv, int nwaves, const unsigned* __restrict__ offsets,
                                                          const AvgState* __restrict__ st, float4* __restrict__ sorted_pts, const unsigned* __restrict__ hist_wg = nullptr) {
  __shared__ unsigned cur[4][AVG_SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int begin = wave * AVG_ITEMS, end = min(n, begin + AVG_ITEMS);
  float4 pts[AVG_ROUNDS];  // this wave's points, in flight while the cursors are derived (see avg_keys_hist_kernel)
  if (wave < nwaves) {
#pragma unroll
    for (int u = 0; u < AVG_ROUNDS; u++) { const int i = begin + u * 64 + lane; pts[u] = load_xyz(xyz, min(i, end - 1), stride); }
  }
  if (hist_wg) {
    __shared__ unsigned s_tot[AVG_SLOTS], s_base[AVG_SLOTS];
    __shared__ unsigned s_pt[4][AVG_SLOTS], s_pb[4][AVG_SLOTS];
    const int nwg = (int)gridDim.x, me = (int)blockIdx.x, t = threadIdx.x;
    // the rows of all workgroups, dealt to the four waves; a lane takes eight consecutive slots of a row as two 16-byte loads
    // (one 4-byte load per slot and row: 116 dependent-issue loads per thread, ~6 of this kernel's 14 us)
    static_assert(AVG_SLOTS == 512, "a row is 64 lanes x 8 slots");
    unsigned tot[8], bef[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { tot[k] = 0; bef[k] = 0; }
    for (int g0 = wv; g0 < nwg; g0 += 32) {
      uint4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int g = min(g0 + 4 * u, nwg - 1);
        const uint4* row = reinterpret_cast<const uint4*>(hist_wg + (size_t)g * AVG_SLOTS) + lane * 2;
        a[u] = row[0]; b[u] = row[1];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int g = g0 + 4 * u;
        if (g < nwg) {
          const unsigned v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
          const bool before = g < me;
#pragma unroll
          for (int k = 0; k < 8; k++) { tot[k] += v[k]; bef[k] += before ? v[k] : 0u; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { s_pt[wv][lane * 8 + k] = tot[k]; s_pb[wv][lane * 8 + k] = bef[k]; }
    __syncthreads();
    const unsigned tot0 = (s_pt[0][t] + s_pt[1][t]) + (s_pt[2][t] + s_pt[3][t]), tot1 = (s_pt[0][256 + t] + s_pt[1][256 + t]) + (s_pt[2][256 + t] + s_pt[3][256 + t]);
    const unsigned bef0 = (s_pb[0][t] + s_pb[1][t]) + (s_pb[2][t] + s_pb[3][t]), bef1 = (s_pb[0][256 + t] + s_pb[1][256 + t]) + (s_pb[2][256 + t] + s_pb[3][256 + t]);
    s_tot[t] = tot0; s_tot[256 + t] = tot1;
    __syncthreads();
    if (t < 64) {  // 512 totals: 8 per lane, exclusive scan
      unsigned v[8], sum = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) { v[u] = s_tot[t * 8 + u]; sum += v[u]; }
      unsigned x = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
      unsigned run = x - sum;
#pragma unroll
      for (int u = 0; u < 8; u++) { s_base[t * 8 + u] = run; run += v[u]; }
    }
    __syncthreads();
    {
      unsigned c0 = s_base[t] + bef0, c1 = s_base[256 + t] + bef1;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        cur[w][t] = c0; cur[w][256 + t] = c1;
        const int gw = me * 4 + w;
        if (gw < nwaves) { c0 += offsets[(size_t)gw * AVG_SLOTS + t]; c1 += offsets[(size_t)gw * AVG_SLOTS + 256 + t]; }
      }
    }
    __syncthreads();
    if (wave >= nwaves) return;
  } else {
    if (wave >= nwaves) return;
    for (int b = lane; b < AVG_SLOTS; b += 64) cur[wv][b] = offsets[(size_t)b * nwaves + wave] + st->bin_base[b];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  }
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int u = 0; u < AVG_ROUNDS; u++) {
    if (begin + u * 64 >= end) break;
    const int i = begin + u * 64 + lane;
    const bool valid = i < end;
    float4 p = pts[u];
    int ix, iy, iz;
    avg_voxel(p, inv, ix, iy, iz);
    const unsigned d = avg_slot(ix, iy, iz);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 9; bit++) {
      const unsigned long long m = __ballot((d >> bit) & 1);
      peers &= ((d >> bit) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int cnt = __popcll(peers);
    const int leader = __ffsll((long long)peers) - 1;
    unsigned dst_base = 0;
    if (valid && lane == leader) { dst_base = cur[wv][d]; cur[wv][d] = dst_base + cnt; }
    dst_base = __shfl(dst_base, leader);
    if (valid) {
      p.w = __int_as_float(i);
      sorted_pts[dst_base + rank] = p;
    }
  }
}

__global__ __launch_bounds__(256) void avg_mark_kernel(const float4* __restrict__ sorted_pts, int n, float inv, unsigned char* __restrict__ head, unsigned* __restrict__ trigbits,
                                                       AvgState* __restrict__ st) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float4 p = sorted_pts[j];
  int ax, ay, az;
  avg_voxel(p, inv, ax, ay, az);
  const unsigned slot = avg_slot(ax, ay, az);
  bool slot_start = (j == 0), h = true;
  if (j > 0) {
    int bx, by, bz;
    avg_voxel(sorted_pts[j - 1], inv, bx, by, bz);
    slot_start = avg_slot(bx, by, bz) != slot;
    h = slot_start || (ax != bx) || (ay != by) || (az != bz);
  }
  if (slot_start) st->slot_used[slot] = 1u;
  else if (h) { const int id = __float_as_int(p.w); atomicOr(&trigbits[id >> 5], 1u << (id & 31)); }
  head[j] = h ? 1 : 0;
}

// ONE workgroup of 1024 threads: block_base[b] = triggers with original index < 1024 b; slot_rank = exclusive scan of slot_used;
// the count of output points goes to the device state (avg_emit hands it to the host).
// `early_result` (mapped host memory, or null): the count is known HERE, one kernel before the centroids exist -- a caller that
// consumes the output on the same stream (fvh_voxelgrid_filter_device_async) gets it now and queues its next stage behind the emit
// kernel while that kernel runs; the emit kernel then reports nothing.
__global__ __launch_bounds__(1024) void avg_scan_kernel(const unsigned* __restrict__ trigbits, int nwords, unsigned* __restrict__ block_base /* [nblocks + 1] */, AvgState* __restrict__ st,
                                                        unsigned long long* __restrict__ early_result = nullptr /* {count, bad, seq} */, unsigned long long seq = 0) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nblocks = (nwords + 31) / 32;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {  // exclusive scan of the per-block popcounts, 1024 blocks (= 1M indices) per round
    const int b = base + tid;
    unsigned c = 0;
    if (b < nblocks) {
      const int w0 = b * 32, w1 = min(nwords, w0 + 32);
      for (int w = w0; w < w1; w++) c += __popc(trigbits[w]);
    }
    unsigned x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    unsigned pre = carry;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    if (b < nblocks) block_base[b] = pre + x - c;
    __syncthreads();
    if (tid == 1023) carry = pre + x;
    __syncthreads();
  }
  if (tid == 0) block_base[nblocks] = carry;
  // slot ranks (512 flags: first 8 waves, one flag per lane)
  unsigned f = 0, x = 0;
  if (tid < AVG_SLOTS) { f = st->slot_used[tid]; x = f; }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
  __syncthreads();
  if (lane == 63 && wv < 8) wsum[wv] = x;
  __syncthreads();
  if (tid < AVG_SLOTS) {
    unsigned pre = 0;
    for (int w = 0; w < wv; w++) pre += wsum[w];
    st->slot_rank[tid] = pre + x - f;
    if (tid == AVG_SLOTS - 1) {
      const unsigned used = pre + x, trig = carry;
      st->slot_rank[AVG_SLOTS] = used;
      st->used_slots = used;
      st->trig_total = trig;
      if (early_result) {  // (every setter of `bad` ran in the first kernel of the chain)
        __hip_atomic_store(&early_result[0], (unsigned long long)(trig + used), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&early_result[1], (unsigned long long)st->bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        st->bad = 0u;  // consumed: re-armed for the next call
        __threadfence_system();
        __hip_atomic_store(&early_result[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// LiDAR runs are long (118k-point frame, leaf 0.5: mean 14 points, p99 134, max 237) and the sum of a run is sequential by
// definition: a thread that fetched its run from L2 eight points at a time paid ~30 dependent round trips for the longest run,
// and the kernel took as long as that run. The workgroup therefore stages a WINDOW of AVG_WINDOW consecutive points (its own 256 and
// the 768 that follow, with their head flags) in LDS with coalesced loads; the walks read LDS (a batch is ~100 cycles instead of
// ~1,500) and fall back to global batches only beyond the window.
constexpr int AVG_WINDOW = 1024;
// the window into LDS (no barrier here: the caller's next __syncthreads() publishes it)
__device__ __forceinline__ void avg_emit_stage(const float4* __restrict__ sorted_pts, const unsigned char* __restrict__ head, int n, float4* s_pts, unsigned char* s_head) {
  const int w0 = blockIdx.x * 256;
#pragma unroll
  for (int u = 0; u < AVG_WINDOW / 256; u++) {
    const int k = w0 + u * 256 + (int)threadIdx.x;
    s_pts[u * 256 + threadIdx.x] = sorted_pts[min(k, n - 1)];
    s_head[u * 256 + threadIdx.x] = (k < n) ? head[k] : (unsigned char)1;
  }
}
__device__ __forceinline__ void avg_emit_points(const float4* __restrict__ sorted_pts, const unsigned char* __restrict__ head, int n, float inv, const unsigned* __restrict__ trigbits,
                                                const unsigned* block_base, const unsigned* slot_rank, unsigned trig_total, float* __restrict__ out,
                                                const float4* s_pts, const unsigned char* s_head, const unsigned short* word_prefix = nullptr /* (LDS) triggers before each word, within its block */,
                                                float* __restrict__ out4 = nullptr /* the same centroids as float4 {x, y, z, 0}: the layout a registration handle keeps its clouds in */) {
  const int w0 = blockIdx.x * 256;
  const int j = w0 + threadIdx.x;
  if (j >= n || !s_head[threadIdx.x]) return;
  float4 p = s_pts[threadIdx.x];
  int ix, iy, iz;
  avg_voxel(p, inv, ix, iy, iz);
  const unsigned slot = avg_slot(ix, iy, iz);
  float cx = 0.f, cy = 0.f, cz = 0.f;
  int e = j;
  float4 q = p;  // the first point of the NEXT run (it flushes this one)
  bool more = true;
  while (more) {  // batched loads, sequential sum (see vg_emit_kernel)
    unsigned char hb[EMIT_BATCH];
    float4 pb[EMIT_BATCH];
    if (e + EMIT_BATCH < w0 + AVG_WINDOW) {  // the whole batch lies in the staged window
#pragma unroll
      for (int u = 0; u < EMIT_BATCH; u++) { const int k = e + 1 + u - w0; hb[u] = s_head[k]; pb[u] = s_pts[k]; }
    } else {
#pragma unroll
      for (int u = 0; u < EMIT_BATCH; u++) { const int k = e + 1 + u; hb[u] = (k < n) ? head[k] : (unsigned char)1; pb[u] = sorted_pts[min(k, n - 1)]; }
    }
#pragma unroll
    for (int u = 0; u < EMIT_BATCH; u++) {
      if (more) {
        cx += p.x; cy += p.y; cz += p.z;
        e++;
        if (hb[u]) { more = false; q = pb[u]; } else p = pb[u];
      }
    }
  }
  const float cnt = (float)(e - j);
  unsigned pos = trig_total + slot_rank[slot];  // last run of its slot: flushed at the end, in slot order
  if (e < n) {
    int qx, qy, qz;
    avg_voxel(q, inv, qx, qy, qz);
    if (avg_slot(qx, qy, qz) == slot) {  // flushed by the arrival of point q: its rank among all flush-triggering points (by original index)
      const int id = __float_as_int(q.w);
      const int w = id >> 5;
      unsigned r = block_base[id >> 10] + __popc(trigbits[w] & ((1u << (id & 31)) - 1u));
      if (word_prefix) r += word_prefix[w];
      else for (int k = (id >> 10) * 32; k < w; k++) r += __popc(trigbits[k]);  // (up to 31 dependent L2 round trips)
      pos = r;
    }
  }
  // write-through: the host may hand `out` to a kernel on ANOTHER stream as soon as it sees the sequence word of the last
  // workgroup (below), i.e. before this kernel's end-of-kernel release
  __hip_atomic_store(&out[3 * (size_t)pos], cx / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&out[3 * (size_t)pos + 1], cy / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&out[3 * (size_t)pos + 2], cz / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (out4) {  // (fvh_ndt_*_from_voxelgrid takes this buffer over as the cloud: no widening kernel, one launch less between the filter and the voxel map)
    __hip_atomic_store(&out4[4 * (size_t)pos], cx / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&out4[4 * (size_t)pos + 1], cy / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&out4[4 * (size_t)pos + 2], cz / cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&out4[4 * (size_t)pos + 3], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The LAST workgroup to finish writes {count, bad flag, sequence} to mapped host memory: when the host sees the sequence
// word, every centroid is in `out` (no copy kernel, no stream synchronisation on the way to the next stage).
// `block_base` == null: the fused chain (no avg_scan_kernel ran) -- every workgroup scans the trigger words and the slot flags itself
// (AVG_FUSED_MAX_BLOCKS blocks of 32 words at most); workgroup 0 also leaves the totals in the device state and, for a caller that consumes
// the output on the same stream (`early_result`), reports the count NOW, before the centroids exist.
__global__ __launch_bounds__(256) void avg_emit_kernel(const float4* __restrict__ sorted_pts, const unsigned char* __restrict__ head, int n, float inv, const unsigned* __restrict__ trigbits,
                                                       const unsigned* __restrict__ block_base, AvgState* __restrict__ st, float* __restrict__ out,
                                                       unsigned long long* __restrict__ host_result /* {count, bad, seq} mapped, or null */, unsigned long long seq,
                                                       int nwords = 0, unsigned long long* __restrict__ early_result = nullptr, float* __restrict__ out4 = nullptr) {
  __shared__ unsigned s_bb[AVG_FUSED_MAX_BLOCKS + 1], s_rank[AVG_SLOTS + 1], s_w[4];
  __shared__ float4 s_pts[AVG_WINDOW];
  __shared__ unsigned char s_head[AVG_WINDOW];
  __shared__ unsigned short s_wp[AVG_FUSED_MAX_BLOCKS * 32];
  unsigned trig_total, used_slots;
  if (block_base) {
    trig_total = st->trig_total; used_slots = st->used_slots;
    avg_emit_stage(sorted_pts, head, n, s_pts, s_head);
    __syncthreads();
    avg_emit_points(sorted_pts, head, n, inv, trigbits, block_base, st->slot_rank, trig_total, out, s_pts, s_head, nullptr, out4);
  } else {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nblocks = (nwords + 31) / 32;
    // (counting the triggers per block with atomics in avg_mark_kernel instead: ~20k adds on ~116 addresses, that kernel 5.5 -> 29 us)
    unsigned c = 0;
    if (t < nblocks) {
      const uint4* q = reinterpret_cast<const uint4*>(trigbits + t * 32);  // (whole blocks: zeroed to the block boundary by avg_keys_hist_kernel)
#pragma unroll
      for (int w = 0; w < 8; w++) {  // (and the triggers before each word of the block: a rank is then one LDS read, not a walk over up to 31 words)
        const uint4 v = q[w];
        unsigned short* wp = s_wp + t * 32 + w * 4;
        wp[0] = (unsigned short)c; c += __popc(v.x);
        wp[1] = (unsigned short)c; c += __popc(v.y);
        wp[2] = (unsigned short)c; c += __popc(v.z);
        wp[3] = (unsigned short)c; c += __popc(v.w);
      }
    }
    const unsigned f0 = st->slot_used[2 * t], f1 = st->slot_used[2 * t + 1];
    avg_emit_stage(sorted_pts, head, n, s_pts, s_head);  // (its loads fly with the ones above; the barriers below publish it)
    unsigned x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    unsigned pre = 0;
    for (int w = 0; w < wv; w++) pre += s_w[w];
    s_bb[t] = pre + x - c;  // (AVG_FUSED_MAX_BLOCKS == 256: one block per thread)
    trig_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (t == 0) s_bb[AVG_FUSED_MAX_BLOCKS] = trig_total;
    __syncthreads();
    // slot ranks: exclusive scan of the 512 "slot used" flags, two per thread
    unsigned y2 = f0 + f1, z = y2;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(z, off); if (lane >= off) z += y; }
    if (lane == 63) s_w[wv] = z;
    __syncthreads();
    unsigned pre2 = 0;
    for (int w = 0; w < wv; w++) pre2 += s_w[w];
    s_rank[2 * t] = pre2 + z - y2; s_rank[2 * t + 1] = pre2 + z - y2 + f0;
    used_slots = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (t == 0) s_rank[AVG_SLOTS] = used_slots;
    __syncthreads();
    if (blockIdx.x == 0 && t == 0) {
      st->trig_total = trig_total; st->used_slots = used_slots;
      if (early_result) {  // (every setter of `bad` ran in the first kernel of the chain)
        __hip_atomic_store(&early_result[0], (unsigned long long)(trig_total + used_slots), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&early_result[1], (unsigned long long)st->bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        st->bad = 0u;  // consumed: re-armed for the next call
        __threadfence_system();
        __hip_atomic_store(&early_result[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    avg_emit_points(sorted_pts, head, n, inv, trigbits, s_bb, s_rank, trig_total, out, s_pts, s_head, s_wp, out4);
  }
  if (!host_result) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's centroids are in memory (write-through stores); no L2 write-back (see avg_binscan_kernel)
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&st->emit_ticket, 1u) == gridDim.x - 1) {
    st->emit_ticket = 0u;
    __hip_atomic_store(&host_result[0], (unsigned long long)(trig_total + used_slots), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host_result[1], (unsigned long long)st->bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    st->bad = 0u;  // consumed: re-armed for the next call (like the ticket), long after every setter of this call has finished
    __threadfence_system();
    __hip_atomic_store(&host_result[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace fvh
