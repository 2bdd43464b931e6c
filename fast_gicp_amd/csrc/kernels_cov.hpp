// Per-point covariance estimation for gfx950 (wave64).
//
// Replaces SURVEY 2.2 K1/K3 (brute_force_knn.cu:16-108: one thread per query, heap in global
// memory), K4 (covariance_estimation.cu:16-51), K5/K6 (covariance_estimation_rbf.cu:11-151: one
// launch per 512-point block + N x ceil(N/512) x 52 B scratch) and K7-K9
// (covariance_regularization.cu:15-125).
//
// Brute-force k-NN: a WAVE owns Q queries; the 64 lanes sweep the candidates 64 at a time
// (coalesced float4 loads, every candidate register reused by the Q queries). The running top-k of
// a query lives ACROSS the lanes of the wave (lane j = j-th smallest so far) and the k-th distance
// is a wave-uniform scalar threshold: a batch whose ballot(d < tau) is empty costs ~9 instructions
// for 64 candidates; an accepted candidate is inserted with one ballot + one wave shift.
// Distances are fp32 ((dx*dx + dy*dy) + dz*dz, no FMA) and ties go to the lower index, so the
// neighbour SETS are bit-reproducible and equal to the oracle's kd-tree result.
#pragma once
#include "dev_math.hpp"

namespace fvh {

// where nn1_corr_kernel finds the pose / the output buffer when it runs inside the device-resident LM loop of FastGICP
// (raw pointers into the LmState on the device: r[9] t[3] doubles per pose)
struct LmLink {
  const int* phase;
  const int* corr_cur;
  const double* x0;
  const double* xi;
  size_t corr_stride;
};

// fp32 squared distance with a fixed association and NO fma contraction (hipcc's __fmul_rn is a
// plain '*' that -ffp-contract=fast would fuse): bit-identical to the oracle's sqdist_f32.
__device__ __forceinline__ float sqdist_nofma(const float4& p, float qx, float qy, float qz) {
#pragma clang fp contract(off)
  const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
  return (dx * dx + dy * dy) + dz * dz;
}
// (x*m0 + y*m1) + (z*m2 + m3): pcl::transformPointCloud's SSE association, no contraction
__device__ __forceinline__ float transform_row_nofma(const float4& p, const float* m) {
#pragma clang fp contract(off)
  return (p.x * m[0] + p.y * m[1]) + (p.z * m[2] + m[3]);
}

constexpr int KNN_Q = 8;  // queries per wave

// lane i <- lane i-1 (lane 0 keeps `fill`): one v_mov_b32_dpp wave_shr:1, no LDS crossbar
__device__ __forceinline__ float wave_shr1(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false); }
// value of a wave-uniform lane -> SGPR (v_readlane_b32); __shfl() would go through the LDS crossbar
__device__ __forceinline__ float read_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// ------------------------------------------------------------------------------------------------
// LDS-tiled candidate sweep shared by the O(N^2) kernels (k-NN, RBF, fitness).
// A workgroup (4 waves x Q queries) walks the candidate cloud in tiles of 512 points staged in LDS
// (8 KB, double buffered): the global loads of tile s+1 are issued before the compute on tile s, so
// the L2/HBM latency hides behind ~2.5k VALU instructions, and each candidate is fetched from L2
// once per workgroup instead of once per wave. Tiles are visited own-tile-first, then zig-zag
// outwards (t0, t0+1, t0-1, ...): LiDAR clouds are scan-ordered, so the k-NN threshold is tight
// after the first tile. body(c, p): c = candidate index of this lane (may be >= n: masked by the
// caller), p = the candidate. Slots past the end of the cloud hold a far-away point (distance ~1e37,
// finite) so the per-query code needs no range check and stays branch-free.
// ------------------------------------------------------------------------------------------------
constexpr int SWEEP_TILE = 512;

__device__ __forceinline__ float4 load_candidate(const float4* __restrict__ pts, int i, int n) {
  const float4 p = pts[min(i, n - 1)];
  const float far = 3.0e18f;
  return i < n ? p : make_float4(far, far, far, 0.f);
}

template <typename F>
__device__ __forceinline__ void sweep_candidates(const float4* __restrict__ pts, int n, int first_tile, float4 (*tile)[SWEEP_TILE], F&& body) {
  const int nt = (n + SWEEP_TILE - 1) / SWEEP_TILE;
  const int tid = threadIdx.x, lane = threadIdx.x & 63;
  auto tile_of = [&](int s) {
    int t = (s & 1) ? first_tile + ((s + 1) >> 1) : first_tile - (s >> 1);
    return t < 0 ? t + nt : (t >= nt ? t - nt : t);
  };
  {
    const int base = tile_of(0) * SWEEP_TILE;
    tile[0][tid] = load_candidate(pts, base + tid, n);
    tile[0][tid + 256] = load_candidate(pts, base + tid + 256, n);
  }
  __syncthreads();
  for (int s = 0; s < nt; s++) {
    float4 n0, n1;
    const bool more = (s + 1 < nt);
    if (more) {
      const int nbase = tile_of(s + 1) * SWEEP_TILE;
      n0 = load_candidate(pts, nbase + tid, n);
      n1 = load_candidate(pts, nbase + tid + 256, n);
    }
    const int base = tile_of(s) * SWEEP_TILE;
    const float4* cur = tile[s & 1];
#pragma unroll 2
    for (int bb = 0; bb < SWEEP_TILE / 64; bb++) body(base + bb * 64 + lane, cur[bb * 64 + lane]);
    if (more) {
      tile[(s + 1) & 1][tid] = n0;
      tile[(s + 1) & 1][tid + 256] = n1;
    }
    __syncthreads();
  }
}

// out_idx: [n][k] neighbour indices, ascending (distance, index); k <= 64.
// The running top-k list is ordered by the total order (distance, index), so the result does not
// depend on the visiting order of the tiles.
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void knn_bruteforce_kernel(const float4* __restrict__ pts, int n, int k, int* __restrict__ out_idx) {
  __shared__ float4 tile[2][SWEEP_TILE];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * KNN_Q;
  float qx[KNN_Q], qy[KNN_Q], qz[KNN_Q];
  float ld[KNN_Q];   // lane j: j-th smallest (distance, index) so far
  int li[KNN_Q];
  float tau_d[KNN_Q];  // wave-uniform: the current k-th entry
  int tau_i[KNN_Q];
#pragma unroll
  for (int j = 0; j < KNN_Q; j++) {
    const float4 q = pts[min(q_base + j, n - 1)];
    qx[j] = q.x; qy[j] = q.y; qz[j] = q.z;
    ld[j] = __builtin_inff(); li[j] = 0x7fffffff; tau_d[j] = __builtin_inff(); tau_i[j] = 0x7fffffff;
  }
  sweep_candidates(pts, n, min(blockIdx.x * 4 * KNN_Q, n - 1) / SWEEP_TILE, tile, [&](int c, const float4& p) {
#pragma unroll
    for (int j = 0; j < KNN_Q; j++) {
      const float d = sqdist_nofma(p, qx[j], qy[j], qz[j]);
      unsigned long long mask = __ballot(d <= tau_d[j]);  // cheap superset test; exact test below
      while (mask) {  // wave-uniform
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float cd = read_lane(d, src);
        const int ci = c - lane + src;
        if (!(cd < tau_d[j] || (cd == tau_d[j] && ci < tau_i[j]))) continue;
        const int pos = __popcll(__ballot(ld[j] < cd || (ld[j] == cd && li[j] < ci)));
        const float sd = wave_shr1(ld[j], ld[j]);
        const int si = wave_shr1(li[j], li[j]);
        if (lane > pos) { ld[j] = sd; li[j] = si; }
        else if (lane == pos) { ld[j] = cd; li[j] = ci; }
        tau_d[j] = read_lane(ld[j], k - 1);
        tau_i[j] = read_lane(li[j], k - 1);
      }
    }
  });
#pragma unroll
  for (int j = 0; j < KNN_Q; j++)
    if (q_base + j < n && lane < k) out_idx[(size_t)(q_base + j) * k + lane] = li[j];
}
#endif

// ------------------------------------------------------------------------------------------------
// Tile-culled exact k-NN. Points are taken in the order given (LiDAR clouds are scan-ordered, so 64
// consecutive points form a spatially compact tile). tile_bbox_kernel stores each tile's bounding
// box; a query wave (Q consecutive queries) seeds its threshold from its own tile and the two
// adjacent ones, then scans the boxes 64 at a time (lane = tile) and only sweeps tiles whose
// box-to-box distance can still beat the current k-th distance. Same list machinery, same total
// order (distance, index): the result is identical to the full sweep for ANY point order; only the
// amount of skipped work depends on how coherent the order is.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_bbox_kernel(const float4* __restrict__ pts, int n, float4* __restrict__ bbox) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t * 64 >= n) return;
  const float4 p = pts[min(t * 64 + lane, n - 1)];
  float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if (lane == 0) {
    bbox[2 * t] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bbox[2 * t + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

// Both box levels in ONE launch (a dependent stage less than tile boxes -> boxes of 64 tile boxes): the first `tile_wgs`
// workgroups box the 64-point tiles (one wave each), the others box 4,096 consecutive points each -- the same min / max as the
// box of the 64 tile boxes, without waiting for them.
__global__ __launch_bounds__(256) void tile_super_bbox_kernel(const float4* __restrict__ pts, int n, float4* __restrict__ bbox1, float4* __restrict__ bbox2, int tile_wgs,
                                                              unsigned* __restrict__ clear_box /* the cloud's bounding cube, consumed earlier in the chain: left zeroed for the next upload */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (clear_box && blockIdx.x == 0 && threadIdx.x < 6) clear_box[threadIdx.x] = 0u;
  float lo[3], hi[3];
  if ((int)blockIdx.x < tile_wgs) {
    const int t = blockIdx.x * 4 + wv;
    if (t * 64 >= n) return;
    const float4 p = pts[min(t * 64 + lane, n - 1)];
    lo[0] = hi[0] = p.x; lo[1] = hi[1] = p.y; lo[2] = hi[2] = p.z;
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
    return;
  }
  __shared__ float s_lo[4][3], s_hi[4][3];
  const int s = blockIdx.x - tile_wgs;
  const int first = s * 4096, last = min(n, first + 4096) - 1;  // (a partial tile pads with its last point, like the tile boxes: no effect on min / max)
  lo[0] = lo[1] = lo[2] = 3e38f; hi[0] = hi[1] = hi[2] = -3e38f;
#pragma unroll 4
  for (int u = 0; u < 16; u++) {
    const float4 p = pts[min(first + u * 256 + (int)threadIdx.x, last)];
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
  if (threadIdx.x == 0) {
    float l[3], h[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      l[a] = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
      h[a] = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
    }
    bbox2[2 * s] = make_float4(l[0], l[1], l[2], 0.f);
    bbox2[2 * s + 1] = make_float4(h[0], h[1], h[2], 0.f);
  }
}

// Wave-wide bitonic sort of one (distance, index) pair per lane, ascending in the total order
// (distance, index): 21 compare-exchange stages. Seeds a query's top-k list from a whole 64-point tile
// at once -- ~250 instructions instead of ~43 serial insertions of ~50 dependent instructions each.
__device__ __forceinline__ void wave_bitonic_sort(float& d, int& i, int lane) {
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      const float od = __shfl_xor(d, j2);
      const int oi = __shfl_xor(i, j2);
      const bool want_min = (((lane & j2) == 0) == ((lane & k2) == 0));
      const bool less = (od < d) || (od == d && oi < i);  // partner < me (indices are unique: no ties)
      const bool take = want_min ? less : !less;
      d = take ? od : d;
      i = take ? oi : i;
    }
  }
}

// box-to-box lower bound of the squared distance, shrunk so that it never exceeds the fp32-rounded
// exact distance of any point pair inside the boxes
__device__ __forceinline__ float box_gap_sq(const float4& bl, const float4& bh, const float* gmin, const float* gmax) {
  const float gx = fmaxf(0.f, fmaxf(bl.x - gmax[0], gmin[0] - bh.x));
  const float gy = fmaxf(0.f, fmaxf(bl.y - gmax[1], gmin[1] - bh.y));
  const float gz = fmaxf(0.f, fmaxf(bl.z - gmax[2], gmin[2] - bh.z));
  return (gx * gx + gy * gy + gz * gz) * 0.99999f;
}

// boxes of 64 consecutive level-1 boxes (one wave per super tile)
__global__ __launch_bounds__(256) void super_bbox_kernel(const float4* __restrict__ bbox1, int ntiles, float4* __restrict__ bbox2,
                                                         unsigned* __restrict__ clear_box = nullptr /* the cloud's bounding cube, consumed earlier in the chain: left zeroed for the next upload */) {
  if (clear_box && blockIdx.x == 0 && threadIdx.x < 6) clear_box[threadIdx.x] = 0u;
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s * 64 >= ntiles) return;
  const int t = min(s * 64 + lane, ntiles - 1);
  const float4 l = bbox1[2 * t], h = bbox1[2 * t + 1];
  float lo[3] = {l.x, l.y, l.z}, hi[3] = {h.x, h.y, h.z};
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if (lane == 0) {
    bbox2[2 * s] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bbox2[2 * s + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

__device__ __forceinline__ float point_box_sq(const float4& bl, const float4& bh, float x, float y, float z) {
  const float gx = fmaxf(0.f, fmaxf(bl.x - x, x - bh.x));
  const float gy = fmaxf(0.f, fmaxf(bl.y - y, y - bh.y));
  const float gz = fmaxf(0.f, fmaxf(bl.z - z, z - bh.z));
  return (gx * gx + gy * gy + gz * gz) * 0.99999f;  // never above the fp32-rounded exact distance
}

// (distance, index) as ONE 64-bit key: squared distances are non-negative floats, whose bit patterns order like unsigned
// integers (+inf, the poison of out-of-range candidates, above every finite value), so the lexicographic
// (distance, original index) order of the exact k-NN is a single u64 compare. The two-float-compare form compiled to
// short-circuit branches on the exec mask: the 21-stage sort alone was ~470 instructions of a 999-instruction kernel
// that is VALU-issue bound (~2,400 issued instructions per query).
__device__ __forceinline__ unsigned long long knn_key(float d, int idx) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)idx; }
__device__ __forceinline__ unsigned long long read_lane64(unsigned long long v, int l) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ unsigned long long wave_shr1_64(unsigned long long v) {
  return ((unsigned long long)(unsigned)wave_shr1((int)(v >> 32), (int)(v >> 32)) << 32) | (unsigned)wave_shr1((int)v, (int)v);
}
__device__ __forceinline__ unsigned long long shfl_xor64(unsigned long long v, int m) {
  return ((unsigned long long)(unsigned)__shfl_xor((int)(v >> 32), m) << 32) | (unsigned)__shfl_xor((int)v, m);
}
// lane ^ 1 / lane ^ 2 through DPP quad permutes (VALU moves, no trip through the LDS crossbar): 11 of the 21 stages of the
// 64-lane bitonic sort
template <int XOR>
__device__ __forceinline__ unsigned long long quad_xor64(unsigned long long v) {
  constexpr int ctrl = (XOR == 1) ? 0xB1 /* quad_perm [1,0,3,2] */ : 0x4E /* quad_perm [2,3,0,1] */;
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), ctrl, 0xF, 0xF, false);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ void wave_bitonic_sort64(unsigned long long& key, int lane) {
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      const unsigned long long other = (j2 == 1) ? quad_xor64<1>(key) : (j2 == 2) ? quad_xor64<2>(key) : shfl_xor64(key, j2);
      const bool want_min = (((lane & j2) == 0) == ((lane & k2) == 0));
      const bool less = other < key;  // keys are unique (indices are)
      key = (want_min == less) ? other : key;
    }
  }
}

// Minimum of a u32 over the wave (every lane gets it): an inclusive min-scan inside the rows of 16 lanes (DPP row shifts; lanes
// without a source keep the identity), the row results passed on with the two row broadcasts, the total read from lane 63.
// Six dependent VALU ops and one readlane -- no trip through the LDS crossbar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_step(unsigned x) {
  return min(x, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = dpp_min_step<0x111, 0xF>(v);  // row_shr:1
  v = dpp_min_step<0x112, 0xF>(v);  // row_shr:2
  v = dpp_min_step<0x114, 0xF>(v);  // row_shr:4
  v = dpp_min_step<0x118, 0xF>(v);  // row_shr:8 -> lane 15 of a row: the row's minimum
  v = dpp_min_step<0x142, 0xA>(v);  // row_bcast:15 into rows 1, 3
  v = dpp_min_step<0x143, 0xC>(v);  // row_bcast:31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ONE query per wave (the regime is latency-chain bound, not throughput bound: measured time was
// proportional to the queries per wave). Seed = bitonic sort of the own tile; two-level box culling
// (super tiles of 64 tiles, then tiles); the next surviving tile is prefetched while the current one
// is merged, so the L2 round trips overlap with the insertion chain.
#ifdef FVH_KNN_TIMING
// debug build: what the culled searches actually evaluate, summed over every launch since the last reset -- [0] k-NN candidate points,
// [1] k-NN box tests, [2] RBF candidate points, [3] RBF box tests, [4] k-NN queries, [5] RBF queries (tools/pair_counts.py: the flop
// fraction of the VALU rooflines = these x 8 flop / launch time / 157.3 TFLOP/s)
__device__ unsigned long long g_pair_counts[8];
#define PAIR_COUNT(slot, n) do { if (lane == 0) atomicAdd(&g_pair_counts[slot], (unsigned long long)(n)); } while (0)
__device__ unsigned long long g_knn_time[32768][8];  // debug build: per query {start, seeds loaded, sorted, neighbours merged, culled sweep done, #tiles swept, #insertions}
#define KNN_STAMP(i) do { if (lane == 0 && q < 32768) g_knn_time[q][i] = wall_clock64(); } while (0)
#else
#define KNN_STAMP(i) do { } while (0)
#define PAIR_COUNT(slot, n) do { } while (0)
#endif
template <bool NEAREST_FIRST>
__global__ __launch_bounds__(256) void knn_tiled1_kernel(const float4* __restrict__ spts, const float4* __restrict__ bbox1, const float4* __restrict__ bbox2, int n, int k,
                                                         int* __restrict__ out_idx, int q_begin = 0, int q_end = 0x7fffffff /* queries = this range of the Morton order (a rank's tile) */) {
  const int lane = threadIdx.x & 63;
  const int q = q_begin + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // (launched with 64, 128 or 256 threads: 1, 2 or 4 queries per workgroup)
  if (q >= min(n, q_end)) return;
  KNN_STAMP(0);
  int dbg_tiles = 0, dbg_ins = 0, dbg_boxes = 0;
  const int ntiles = (n + 63) >> 6, nsuper = (ntiles + 63) >> 6;
  const float4 qv = spts[q];
  const float qx = read_lane(qv.x, 0), qy = read_lane(qv.y, 0), qz = read_lane(qv.z, 0);
  const int t0 = q >> 6;
  // issue the three seed loads together
  const float4 p0 = load_candidate(spts, (t0 << 6) + lane, n);
  const float4 pa = load_candidate(spts, (min(t0 + 1, ntiles - 1) << 6) + lane, n);
  const float4 pb = load_candidate(spts, (max(t0 - 1, 0) << 6) + lane, n);
  unsigned long long lk = knn_key(sqdist_nofma(p0, qx, qy, qz), (((t0 << 6) + lane) < n) ? __float_as_int(p0.w) : 0x7fffffff);
#ifdef FVH_KNN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  KNN_STAMP(1);
  wave_bitonic_sort64(lk, lane);
  KNN_STAMP(2);
  unsigned long long tk = read_lane64(lk, k - 1);  // the current k-th entry: the acceptance threshold
  float td = __uint_as_float((unsigned)(tk >> 32));

  auto merge_tile = [&](const float4& p) __attribute__((always_inline)) {
    const unsigned long long ck = knn_key(sqdist_nofma(p, qx, qy, qz), __float_as_int(p.w));
    // exact acceptance test per lane: (d, idx) below the current k-th entry in the total order
    unsigned long long mask = __ballot(ck < tk);
    dbg_tiles++;
    while (mask) {
      dbg_ins++;
      const int c = __ffsll((long long)mask) - 1;
      const unsigned long long cc = read_lane64(ck, c);
      const int pos = __popcll(__ballot(lk < cc));
      const unsigned long long sh = wave_shr1_64(lk);
      lk = (lane > pos) ? sh : ((lane == pos) ? cc : lk);
      tk = read_lane64(lk, k - 1);
      // the threshold tightened: candidates that no longer qualify leave the mask here instead of costing an iteration each
      mask &= __ballot(ck < tk) & ~(1ull << c);
    }
    td = __uint_as_float((unsigned)(tk >> 32));
  };
  if (t0 + 1 < ntiles) merge_tile(pa);
  if (t0 > 0) merge_tile(pb);
  KNN_STAMP(3);

  if constexpr (!NEAREST_FIRST) {
    // boxes in index order: scalar mask arithmetic, a find-first-set per box -- the cheapest walk per box
    for (int sc = 0; sc < nsuper; sc += 64) {
      const int s = sc + lane;
      const float lb2 = (s < nsuper) ? point_box_sq(bbox2[2 * s], bbox2[2 * s + 1], qx, qy, qz) : __builtin_inff();
      dbg_boxes += 64;
      unsigned long long smask = __ballot(lb2 <= td);
      while (smask) {
        const int ssrc = __ffsll((long long)smask) - 1;
        smask &= smask - 1;
        if (read_lane(lb2, ssrc) > td) continue;
        const int t = ((sc + ssrc) << 6) + lane;
        const float lb = (t < ntiles) ? point_box_sq(bbox1[2 * t], bbox1[2 * t + 1], qx, qy, qz) : __builtin_inff();
        dbg_boxes += 64;
        unsigned long long tmask = __ballot(lb <= td && (t < t0 - 1 || t > t0 + 1));
        if (!tmask) continue;
        // software pipeline: the load of the next surviving tile is in flight while this one is merged
        int cur = __ffsll((long long)tmask) - 1;
        tmask &= tmask - 1;
        float4 pcur = load_candidate(spts, ((((sc + ssrc) << 6) + cur) << 6) + lane, n);
        while (true) {
          int nxt = -1;
          float4 pnxt = pcur;
          if (tmask) {
            nxt = __ffsll((long long)tmask) - 1;
            tmask &= tmask - 1;
            pnxt = load_candidate(spts, ((((sc + ssrc) << 6) + nxt) << 6) + lane, n);
          }
          if (read_lane(lb, cur) <= td) merge_tile(pcur);
          if (nxt < 0) break;
          cur = nxt;
          pcur = pnxt;
        }
      }
    }
  } else {
    // Nearest box first, at both levels (wave_min_u32 per box, the keys in a VGPR: ~35 more instructions per box). The order does
    // not change the result -- the k smallest keys of a total order -- only how many candidates pass the threshold on the way. A
    // query whose own tile straddles a jump of the Morton curve meets its true neighbours late in index order: 7 such queries of
    // the bundled target cloud took 400 insertions / 43 us each and alone stretched the kernel from 34 to 58 us
    // (tools/knn_timing.py). Ascending boxes also end the walk at the first one beyond the threshold (it only shrinks). Small
    // clouds take this walk: their kernel lasts as long as its slowest queries (17k points: 49 -> 39 us on average over the two
    // bundled clouds). The throughput-bound sizes hide a slow query behind the others and pay for the extra instructions
    // instead (100k points: 159 -> 169 us), so they keep the index order. (One kernel that switches per query after N
    // insertions was measured too: its index-order part compiled 10-17 % slower than the plain one.)
    for (int sc = 0; sc < nsuper; sc += 64) {
      const int s = sc + lane;
      const float lb2 = (s < nsuper) ? point_box_sq(bbox2[2 * s], bbox2[2 * s + 1], qx, qy, qz) : __builtin_inff();
      dbg_boxes += 64;
      unsigned k2 = (lb2 <= td) ? ((s == (t0 >> 6)) ? 0u : __float_as_uint(lb2) + 1u) : ~0u;  // (the own super tile before all others)
      while (true) {
        const unsigned m2 = wave_min_u32(k2);
        if (m2 == ~0u) break;
        const int ssrc = __ffsll((long long)__ballot(k2 == m2)) - 1;
        k2 = (lane == ssrc) ? ~0u : k2;
        if (read_lane(lb2, ssrc) > td) break;  // ascending: none of the remaining super tiles can qualify either
        const int t = ((sc + ssrc) << 6) + lane;
        const float lb = (t < ntiles) ? point_box_sq(bbox1[2 * t], bbox1[2 * t + 1], qx, qy, qz) : __builtin_inff();
        dbg_boxes += 64;
        unsigned k1 = (lb <= td && (t < t0 - 1 || t > t0 + 1)) ? __float_as_uint(lb) : ~0u;
        unsigned m1 = wave_min_u32(k1);
        if (m1 == ~0u) continue;
        // software pipeline: the load of the next surviving tile is in flight while this one is merged
        int cur = __ffsll((long long)__ballot(k1 == m1)) - 1;
        k1 = (lane == cur) ? ~0u : k1;
        float4 pcur = load_candidate(spts, ((((sc + ssrc) << 6) + cur) << 6) + lane, n);
        while (true) {
          int nxt = -1;
          float4 pnxt = pcur;
          m1 = wave_min_u32(k1);
          if (m1 != ~0u && __uint_as_float(m1) <= td) {
            nxt = __ffsll((long long)__ballot(k1 == m1)) - 1;
            k1 = (lane == nxt) ? ~0u : k1;
            pnxt = load_candidate(spts, ((((sc + ssrc) << 6) + nxt) << 6) + lane, n);
          }
          if (read_lane(lb, cur) <= td) merge_tile(pcur);
          if (nxt < 0) break;
          cur = nxt;
          pcur = pnxt;
        }
      }
    }
  }
  if (lane < k) out_idx[(size_t)__float_as_int(qv.w) * k + lane] = (int)(unsigned)lk;
  KNN_STAMP(4);
#ifdef FVH_KNN_TIMING
  if (lane == 0 && q < 32768) { g_knn_time[q][5] = dbg_tiles; g_knn_time[q][6] = dbg_ins; }
#endif
  PAIR_COUNT(0, 64 * (1 + dbg_tiles)); PAIR_COUNT(1, dbg_boxes); PAIR_COUNT(4, 1);  // (the own tile + every merged tile: 64 candidate distances each)
}

__device__ __forceinline__ void store_cov(float4* __restrict__ cov, int i, const Sym3<double>& C) {
  cov[2 * i] = make_float4((float)C.xx, (float)C.xy, (float)C.xz, (float)C.yy);
  cov[2 * i + 1] = make_float4((float)C.yz, (float)C.zz, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// FastVGICPCuda arithmetic (fvh_vgicp_set_precision(FVH_COMPUTE_CUDA_COMPAT)): the covariance of a point as the reference's DEVICE
// path computes it -- covariance_estimation.cu:20-35: UNCENTRED float sums over the k neighbours in list order, mean = sum / k,
// C = sum p p^T / k - mean mean^T; covariance_regularization.cu:15-125: Eigen::SelfAdjointEigenSolver<Matrix3f>::computeDirect (closed
// form, trigonometric roots, eigenvalues ascending), C <- V diag V^-1 with the cofactor inverse (PLANE: diag(1e-3, 1, 1); MIN_EIG:
// max(lambda, 1e-3)), FROBENIUS through two cofactor inverses; NONE / NORMALIZED_MIN_EIG leave the matrix alone (:122-124).
// At 50 m from the sensor the uncentred float sums carry |p|^2 x 6e-8 ~ 2e-4 m^2 of rounding noise -- the size of a plane's smallest
// eigenvalue -- so the result depends on the ORDER of the float operations: every expression below has the association of the
// statement it restates and NO fma contraction (as oracle/cuda_compat.cpp, the judge of this mode, which is built with
// -ffp-contract=off; nvcc contracts at its discretion, so the real device code differs from both in the last bits).
// One thread per point: 20 independent gathers, ~400 float operations; this mode is about parity, not speed.
// ------------------------------------------------------------------------------------------------
struct M3f { float m[9]; };
__device__ __forceinline__ M3f m3f_mul(const M3f& a, const M3f& b) {
#pragma clang fp contract(off)
  M3f c;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ M3f m3f_inverse(const M3f& a) {  // Eigen's fixed-size 3x3 inverse: cofactors / determinant
#pragma clang fp contract(off)
  M3f c;
  c.m[0] = a.m[4] * a.m[8] - a.m[5] * a.m[7];
  c.m[1] = a.m[2] * a.m[7] - a.m[1] * a.m[8];
  c.m[2] = a.m[1] * a.m[5] - a.m[2] * a.m[4];
  c.m[3] = a.m[5] * a.m[6] - a.m[3] * a.m[8];
  c.m[4] = a.m[0] * a.m[8] - a.m[2] * a.m[6];
  c.m[5] = a.m[2] * a.m[3] - a.m[0] * a.m[5];
  c.m[6] = a.m[3] * a.m[7] - a.m[4] * a.m[6];
  c.m[7] = a.m[1] * a.m[6] - a.m[0] * a.m[7];
  c.m[8] = a.m[0] * a.m[4] - a.m[1] * a.m[3];
  const float det = a.m[0] * c.m[0] + a.m[1] * c.m[3] + a.m[2] * c.m[6];
  const float inv = 1.0f / det;
#pragma unroll
  for (int i = 0; i < 9; i++) c.m[i] *= inv;
  return c;
}
struct V3f { float v[3]; };
__device__ __forceinline__ V3f v3f_cross(const V3f& a, const V3f& b) {
#pragma clang fp contract(off)
  return V3f{{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
__device__ __forceinline__ float v3f_dot(const V3f& a, const V3f& b) {
#pragma clang fp contract(off)
  return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2];
}
__device__ __forceinline__ V3f m3f_col(const M3f& a, int j) { return V3f{{a.m[j], a.m[3 + j], a.m[6 + j]}}; }
// direct_selfadjoint_eigenvalues<..., 3, false>::extractKernel
__device__ __forceinline__ void eig3f_extract_kernel(const M3f& mat, V3f& res, V3f& representative) {
#pragma clang fp contract(off)
  int i0 = 0;
  if (fabsf(mat.m[4]) > fabsf(mat.m[0])) i0 = 1;
  if (fabsf(mat.m[8]) > fabsf(mat.m[i0 * 4])) i0 = 2;
  representative = m3f_col(mat, i0);
  const V3f c0 = v3f_cross(representative, m3f_col(mat, (i0 + 1) % 3)), c1 = v3f_cross(representative, m3f_col(mat, (i0 + 2) % 3));
  const float n0 = v3f_dot(c0, c0), n1 = v3f_dot(c1, c1);
  if (n0 > n1) { const float s = sqrtf(n0); res = V3f{{c0.v[0] / s, c0.v[1] / s, c0.v[2] / s}}; }
  else { const float s = sqrtf(n1); res = V3f{{c1.v[0] / s, c1.v[1] / s, c1.v[2] / s}}; }
}
// SelfAdjointEigenSolver<Matrix3f>::computeDirect: vals ascending, eigenvectors in the COLUMNS of vecs; reads the lower triangle
__device__ inline void eig3f_direct(const M3f& A, float vals[3], M3f& vecs) {
#pragma clang fp contract(off)
  const float shift = (A.m[0] + A.m[4] + A.m[8]) / 3.0f;
  M3f S;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) S.m[i * 3 + j] = i >= j ? A.m[i * 3 + j] : A.m[j * 3 + i];
  S.m[0] -= shift; S.m[4] -= shift; S.m[8] -= shift;
  float scale = 0.0f;
#pragma unroll
  for (int i = 0; i < 9; i++) scale = fmaxf(scale, fabsf(S.m[i]));
  if (scale > 0.0f) {
#pragma unroll
    for (int i = 0; i < 9; i++) S.m[i] /= scale;
  }
  {  // computeRoots: x^3 - c2 x^2 + c1 x - c0 = 0
    const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
    const float c0 = S.m[0] * S.m[4] * S.m[8] + 2.0f * S.m[3] * S.m[6] * S.m[7] - S.m[0] * S.m[7] * S.m[7] - S.m[4] * S.m[6] * S.m[6] - S.m[8] * S.m[3] * S.m[3];
    const float c1 = S.m[0] * S.m[4] - S.m[3] * S.m[3] + S.m[0] * S.m[8] - S.m[6] * S.m[6] + S.m[4] * S.m[8] - S.m[7] * S.m[7];
    const float c2 = S.m[0] + S.m[4] + S.m[8];
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = fmaxf(a_over_3, 0.0f);
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = fmaxf(q, 0.0f);
    const float rho = sqrtf(a_over_3);
    const float theta = atan2f(sqrtf(q), half_b) * s_inv3;
    const float cos_theta = cosf(theta), sin_theta = sinf(theta);
    vals[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    vals[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    vals[2] = c2_over_3 + 2.0f * rho * cos_theta;
  }
  const float eps = 1.1920929e-07f;  // NumTraits<float>::epsilon()
  V3f e0 = {{1.f, 0.f, 0.f}}, e1 = {{0.f, 1.f, 0.f}}, e2 = {{0.f, 0.f, 1.f}};
  if (!((vals[2] - vals[0]) <= eps)) {
    M3f tmp = S;
    float d0 = vals[2] - vals[1];
    const float d1 = vals[1] - vals[0];
    int k = 0, l = 2;
    if (d0 > d1) { k = 2; l = 0; d0 = d1; }
    V3f vk, vl;
    tmp.m[0] -= vals[k]; tmp.m[4] -= vals[k]; tmp.m[8] -= vals[k];
    eig3f_extract_kernel(tmp, vk, vl);  // vl = the saved representative column (nearly orthogonal to vk)
    if (d0 <= 2.0f * eps * d1) {
      const float pr = v3f_dot(vk, vl);
#pragma unroll
      for (int i = 0; i < 3; i++) vl.v[i] -= pr * vl.v[i];  // (as written in Eigen: col(l) -= col(k).dot(col(l)) * col(l))
      const float nn = sqrtf(v3f_dot(vl, vl));
#pragma unroll
      for (int i = 0; i < 3; i++) vl.v[i] /= nn;
    } else {
      tmp = S;
      tmp.m[0] -= vals[l]; tmp.m[4] -= vals[l]; tmp.m[8] -= vals[l];
      V3f dummy;
      eig3f_extract_kernel(tmp, vl, dummy);
    }
    if (k == 0) { e0 = vk; e2 = vl; } else { e2 = vk; e0 = vl; }
    V3f v1 = v3f_cross(e2, e0);
    const float nn = sqrtf(v3f_dot(v1, v1));
#pragma unroll
    for (int i = 0; i < 3; i++) v1.v[i] /= nn;
    e1 = v1;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) { vecs.m[i * 3] = e0.v[i]; vecs.m[i * 3 + 1] = e1.v[i]; vecs.m[i * 3 + 2] = e2.v[i]; }
#pragma unroll
  for (int i = 0; i < 3; i++) vals[i] = vals[i] * scale + shift;
}
__device__ inline M3f regularize_cov_cuda_compat(const M3f& cov, int method) {
#pragma clang fp contract(off)
  if (method == 4) {  // FROBENIUS (:74-82)
    M3f C = cov;
    C.m[0] += 1e-3f; C.m[4] += 1e-3f; C.m[8] += 1e-3f;
    M3f Ci = m3f_inverse(C);
    float nrm = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; i++) nrm += Ci.m[i] * Ci.m[i];
    nrm = sqrtf(nrm);
#pragma unroll
    for (int i = 0; i < 9; i++) Ci.m[i] /= nrm;
    return m3f_inverse(Ci);
  }
  if (method != 3 && method != 1) return cov;  // NONE / NORMALIZED_MIN_EIG: ":122-124 unimplemented", the matrix is left alone
  float vals[3];
  M3f V;
  eig3f_direct(cov, vals, V);
  M3f D;
#pragma unroll
  for (int i = 0; i < 9; i++) D.m[i] = 0.0f;
  if (method == 3) { D.m[0] = 1e-3f; D.m[4] = 1.0f; D.m[8] = 1.0f; }  // PLANE (:34-52,112)
  else { D.m[0] = fmaxf(1e-3f, vals[0]); D.m[4] = fmaxf(1e-3f, vals[1]); D.m[8] = fmaxf(1e-3f, vals[2]); }  // MIN_EIG (:84-101)
  return m3f_mul(m3f_mul(V, D), m3f_inverse(V));
}
__global__ __launch_bounds__(256) void cov_from_neighbors_cuda_compat_kernel(const float4* __restrict__ pts, int n, int k, const int* __restrict__ nbr, int method,
                                                                             float4* __restrict__ cov, const int* __restrict__ subset /* a rank's tile of the Morton order, or null */) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int i = subset ? subset[t] : t;
  float mean[3] = {0.f, 0.f, 0.f};
  M3f C;
#pragma unroll
  for (int q = 0; q < 9; q++) C.m[q] = 0.f;
  for (int j = 0; j < k; j++) {
    const float4 p4 = pts[nbr[(size_t)i * k + j]];
    const float p[3] = {p4.x, p4.y, p4.z};
#pragma unroll
    for (int a = 0; a < 3; a++) mean[a] += p[a];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) C.m[a * 3 + b] += p[a] * p[b];
  }
  const float kf = (float)k;
#pragma unroll
  for (int a = 0; a < 3; a++) mean[a] /= kf;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) C.m[a * 3 + b] = C.m[a * 3 + b] / kf - mean[a] * mean[b];
  const M3f R = regularize_cov_cuda_compat(C, method);
  // the engine stores a covariance as its upper triangle (fvh_vgicp_set_*_covariances reads the same six entries of a 3x3)
  cov[2 * (size_t)i] = make_float4(R.m[0], R.m[1], R.m[2], R.m[4]);
  cov[2 * (size_t)i + 1] = make_float4(R.m[5], R.m[8], 0.f, 0.f);
}

// K4 + K7/8/9 fused: centred fp64 covariance of the k neighbours (CPU semantics,
// fast_gicp_impl.hpp:259-265; the CUDA path's uncentred fp32 sum loses ~2 digits at 50 m range),
// then regularisation. FOUR lanes per point: with one thread per point a 17k-point cloud is 272 waves on 1,024 SIMDs,
// each walking 2 x 20 dependent gathers (21 us); here a lane gathers k/4 neighbours once, keeps them in registers for
// the second (centred) pass, and the partial sums meet through two xor-shuffles in a fixed order.
constexpr int COV_LANES = 4, COV_MAX_PER_LANE = 16;  // k <= 64
// The 64 points of a 256-thread workgroup (4 lanes each) hand their raw covariances to the FIRST wave through LDS, which
// regularises one point per lane: the eigen-decomposition (~1,000 dependent fp64 instructions) runs once per workgroup instead of
// once per wave with every value replicated four times. Called by all 256 threads.
__device__ __forceinline__ void cov_regularize_dense(const Sym3<double>& C, int method, float4* __restrict__ cov, int i, bool owner /* one lane per valid point */,
                                                     float4* __restrict__ cov_sorted = nullptr /* optional second copy, at position pos0 + (point's number in this launch) */, int pos0 = 0) {
  __shared__ double s_c[64][6];
  __shared__ int s_i[64];
  const int p = threadIdx.x / COV_LANES;
  if ((threadIdx.x % COV_LANES) == 0) {
    s_c[p][0] = C.xx; s_c[p][1] = C.xy; s_c[p][2] = C.xz; s_c[p][3] = C.yy; s_c[p][4] = C.yz; s_c[p][5] = C.zz;
    s_i[p] = owner ? i : -1;
  }
  __syncthreads();
  if (threadIdx.x < 64 && s_i[threadIdx.x] >= 0) {
    const int t = threadIdx.x;
    const Sym3<double> R = regularize_cov(Sym3<double>{s_c[t][0], s_c[t][1], s_c[t][2], s_c[t][3], s_c[t][4], s_c[t][5]}, method);
    store_cov(cov, s_i[t], R);
    if (cov_sorted) store_cov(cov_sorted, pos0 + (int)blockIdx.x * 64 + t, R);  // (64 points per workgroup, in launch order)
  }
}
template <int PER_LANE>  // neighbours a lane holds: k <= 4 * PER_LANE (5 for the reference's k = 20)
__global__ __launch_bounds__(256) void cov_from_neighbors_kernel(const float4* __restrict__ pts, int n, int k, const int* __restrict__ nbr, int method,
                                                                 float4* __restrict__ cov, const int* __restrict__ subset = nullptr /* n point indices (a rank's tile), or all */,
                                                                 float4* __restrict__ cov_sorted = nullptr /* subset = the Morton order: the covariances again, in that order */) {
  const int gt = blockIdx.x * 256 + threadIdx.x;
  const int sub = gt % COV_LANES;
  const int i = subset ? subset[min(gt / COV_LANES, n - 1)] : min(gt / COV_LANES, n - 1);
  const int* nb = nbr + (size_t)i * k;
  float px[PER_LANE], py[PER_LANE], pz[PER_LANE];
  double mx = 0, my = 0, mz = 0;
  // Two round trips for the whole gather: all indices, then all points (slots past k read a valid neighbour again and are masked). Written
  // as "if (j < k) { p = pts[nb[j]]; ... }" the lane-dependent branch kept every load behind its own wait: ten dependent round trips,
  // ~4 of this kernel's 8.9 us at 17k points (tools/scan_serial_loads.py).
  int idx[PER_LANE];
#pragma unroll
  for (int u = 0; u < PER_LANE; u++) idx[u] = nb[min(sub + u * COV_LANES, k - 1)];
  float4 pq[PER_LANE];
#pragma unroll
  for (int u = 0; u < PER_LANE; u++) pq[u] = pts[idx[u]];
#pragma unroll
  for (int u = 0; u < PER_LANE; u++) {
    const bool in = sub + u * COV_LANES < k;
    px[u] = in ? pq[u].x : 0.f; py[u] = in ? pq[u].y : 0.f; pz[u] = in ? pq[u].z : 0.f;
    mx = in ? mx + (double)pq[u].x : mx; my = in ? my + (double)pq[u].y : my; mz = in ? mz + (double)pq[u].z : mz;
  }
#pragma unroll
  for (int off = 1; off < COV_LANES; off <<= 1) { mx += __shfl_xor(mx, off); my += __shfl_xor(my, off); mz += __shfl_xor(mz, off); }
  mx /= k; my /= k; mz /= k;
  Sym3<double> C = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < PER_LANE; u++) {
    if (sub + u * COV_LANES < k) {
      const double dx = (double)px[u] - mx, dy = (double)py[u] - my, dz = (double)pz[u] - mz;
      C.xx += dx * dx; C.xy += dx * dy; C.xz += dx * dz; C.yy += dy * dy; C.yz += dy * dz; C.zz += dz * dz;
    }
  }
#pragma unroll
  for (int off = 1; off < COV_LANES; off <<= 1) {
    C.xx += __shfl_xor(C.xx, off); C.xy += __shfl_xor(C.xy, off); C.xz += __shfl_xor(C.xz, off);
    C.yy += __shfl_xor(C.yy, off); C.yz += __shfl_xor(C.yz, off); C.zz += __shfl_xor(C.zz, off);
  }
  const double inv = 1.0 / k;
  C.xx *= inv; C.xy *= inv; C.xz *= inv; C.yy *= inv; C.yz *= inv; C.zz *= inv;
  cov_regularize_dense(C, method, cov, i, sub == 0 && gt / COV_LANES < n, cov_sorted, 0);
}

// k > 32 (up to 64): the same four-lanes-per-point scheme, but the neighbours are gathered again for the centred pass instead
// of being held in registers (the 16-per-lane instantiation of the kernel above needed 252 VGPRs + 928 spilled ones); the
// second gather hits the lines the first one just brought in.
__global__ __launch_bounds__(256) void cov_from_neighbors_regather_kernel(const float4* __restrict__ pts, int n, int k, const int* __restrict__ nbr, int method,
                                                                          float4* __restrict__ cov, const int* __restrict__ subset = nullptr, float4* __restrict__ cov_sorted = nullptr) {
  const int gt = blockIdx.x * 256 + threadIdx.x;
  const int sub = gt % COV_LANES;
  const int i = subset ? subset[min(gt / COV_LANES, n - 1)] : min(gt / COV_LANES, n - 1);
  const int* nb = nbr + (size_t)i * k;
  double mx = 0, my = 0, mz = 0;
#pragma unroll 4
  for (int j = sub; j < k; j += COV_LANES) {
    const float4 p = pts[nb[j]];
    mx += (double)p.x; my += (double)p.y; mz += (double)p.z;
  }
#pragma unroll
  for (int off = 1; off < COV_LANES; off <<= 1) { mx += __shfl_xor(mx, off); my += __shfl_xor(my, off); mz += __shfl_xor(mz, off); }
  mx /= k; my /= k; mz /= k;
  Sym3<double> C = {0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int j = sub; j < k; j += COV_LANES) {
    const float4 p = pts[nb[j]];
    const double dx = (double)p.x - mx, dy = (double)p.y - my, dz = (double)p.z - mz;
    C.xx += dx * dx; C.xy += dx * dy; C.xz += dx * dz; C.yy += dy * dy; C.yz += dy * dz; C.zz += dz * dz;
  }
#pragma unroll
  for (int off = 1; off < COV_LANES; off <<= 1) {
    C.xx += __shfl_xor(C.xx, off); C.xy += __shfl_xor(C.xy, off); C.xz += __shfl_xor(C.xz, off);
    C.yy += __shfl_xor(C.yy, off); C.yz += __shfl_xor(C.yz, off); C.zz += __shfl_xor(C.zz, off);
  }
  const double inv = 1.0 / k;
  C.xx *= inv; C.xy *= inv; C.xz *= inv; C.yy *= inv; C.yz *= inv; C.zz *= inv;
  cov_regularize_dense(C, method, cov, i, sub == 0 && gt / COV_LANES < n, cov_sorted, 0);
}

__global__ __launch_bounds__(256) void regularize_kernel(float4* __restrict__ cov, int n, int method) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = cov[2 * i], b = cov[2 * i + 1];
  Sym3<double> C = {a.x, a.y, a.z, a.w, b.x, b.y};
  store_cov(cov, i, regularize_cov(C, method));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int RBF_Q = 8;

// K5 + K6 + K7 fused, single pass, no scratch: w = exp(-kernel_width * d^2) for d^2 <= max_dist^2
// (covariance_estimation_rbf.cu:67-85), weighted mean/cov (:40-52), evaluated centred on the query
// (identical maths, fp32-safe), wave-reduced in fp64, regularised. The reference's unmasked
// zero-padding of the last 512-block (:127-129) is NOT replicated.
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void cov_rbf_kernel(const float4* __restrict__ pts, int n, float kernel_width, float max_dist_sq, int method,
                                                      float4* __restrict__ cov) {
  __shared__ float4 tile[2][SWEEP_TILE];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * RBF_Q;
  float qx[RBF_Q], qy[RBF_Q], qz[RBF_Q];
  float sw[RBF_Q], sx[RBF_Q], sy[RBF_Q], sz[RBF_Q], sxx[RBF_Q], sxy[RBF_Q], sxz[RBF_Q], syy[RBF_Q], syz[RBF_Q], szz[RBF_Q];
#pragma unroll
  for (int j = 0; j < RBF_Q; j++) {
    const float4 q = pts[min(q_base + j, n - 1)];
    qx[j] = q.x; qy[j] = q.y; qz[j] = q.z;
    sw[j] = sx[j] = sy[j] = sz[j] = sxx[j] = sxy[j] = sxz[j] = syy[j] = syz[j] = szz[j] = 0.f;
  }
  sweep_candidates(pts, n, 0, tile, [&](int c, const float4& p) {
#pragma unroll
    for (int j = 0; j < RBF_Q; j++) {
      const float dx = p.x - qx[j], dy = p.y - qy[j], dz = p.z - qz[j];
      const float sq = sqdist_nofma(p, qx[j], qy[j], qz[j]);
      const float w = (sq > max_dist_sq) ? 0.f : __expf(-kernel_width * sq);
      sw[j] += w;
      const float wx = w * dx, wy = w * dy, wz = w * dz;
      sx[j] += wx; sy[j] += wy; sz[j] += wz;
      sxx[j] += wx * dx; sxy[j] += wx * dy; sxz[j] += wx * dz; syy[j] += wy * dy; syz[j] += wy * dz; szz[j] += wz * dz;
    }
  });
#pragma unroll
  for (int j = 0; j < RBF_Q; j++) {
    const double W = wave_sum((double)sw[j]);
    const double X = wave_sum((double)sx[j]), Y = wave_sum((double)sy[j]), Z = wave_sum((double)sz[j]);
    const double XX = wave_sum((double)sxx[j]), XY = wave_sum((double)sxy[j]), XZ = wave_sum((double)sxz[j]);
    const double YY = wave_sum((double)syy[j]), YZ = wave_sum((double)syz[j]), ZZ = wave_sum((double)szz[j]);
    if (lane == 0 && q_base + j < n) {
      const double iw = 1.0 / W;
      const double mx = X * iw, my = Y * iw, mz = Z * iw;
      Sym3<double> C;
      C.xx = XX * iw - mx * mx; C.xy = XY * iw - mx * my; C.xz = XZ * iw - mx * mz;
      C.yy = YY * iw - my * my; C.yz = YZ * iw - my * mz; C.zz = ZZ * iw - mz * mz;
      store_cov(cov, q_base + j, regularize_cov(C, method));
    }
  }
}
#endif

constexpr int FIT_Q = 8;

// pcl::Registration::getFitnessScore restated for the device: transform the source by the FLOAT
// pose (final_transformation_ is float), exact 1-NN by tiled brute force, sum d^2 <= max_range.
// out[0] += sum, out[1] += count (fp64 atomics).
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void fitness_kernel(const float4* __restrict__ src, int ns, const float4* __restrict__ tgt, int nt, const float* __restrict__ T12 /* row-major 3x4 */,
                                                      double max_range, double* __restrict__ out) {
  __shared__ float4 tile[2][SWEEP_TILE];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * FIT_Q;
  float qx[FIT_Q], qy[FIT_Q], qz[FIT_Q], best[FIT_Q];
#pragma unroll
  for (int j = 0; j < FIT_Q; j++) {
    const float4 p = src[min(q_base + j, ns - 1)];
    qx[j] = transform_row_nofma(p, T12 + 0);
    qy[j] = transform_row_nofma(p, T12 + 4);
    qz[j] = transform_row_nofma(p, T12 + 8);
    best[j] = __builtin_inff();
  }
  sweep_candidates(tgt, nt, 0, tile, [&](int c, const float4& p) {
#pragma unroll
    for (int j = 0; j < FIT_Q; j++) {
      best[j] = fminf(best[j], sqdist_nofma(p, qx[j], qy[j], qz[j]));
    }
  });
  double sum = 0.0, cnt = 0.0;
#pragma unroll
  for (int j = 0; j < FIT_Q; j++) {
    float b = best[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) b = fminf(b, __shfl_xor(b, off));
    if (q_base + j < ns && (double)b <= max_range) { sum += (double)b; cnt += 1.0; }
  }
  if (lane == 0 && (sum != 0.0 || cnt != 0.0)) {
    atomicAdd(&out[0], sum);
    atomicAdd(&out[1], cnt);
  }
}
#endif

// RBF covariances on the Morton-sorted cloud: only tiles whose box is within max_dist of the wave's
// query box are swept (fixed-radius culling). cov is indexed by ORIGINAL point index.
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void cov_rbf_tiled_kernel(const float4* __restrict__ spts, const float4* __restrict__ bbox, int n, float kernel_width, float max_dist_sq,
                                                            int method, float4* __restrict__ cov) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * RBF_Q;
  if (q_base >= n) return;
  const int ntiles = (n + 63) >> 6;
  float qx[RBF_Q], qy[RBF_Q], qz[RBF_Q];
  int qo[RBF_Q];
  float sw[RBF_Q], sx[RBF_Q], sy[RBF_Q], sz[RBF_Q], sxx[RBF_Q], sxy[RBF_Q], sxz[RBF_Q], syy[RBF_Q], syz[RBF_Q], szz[RBF_Q];
  float gmin[3] = {3e38f, 3e38f, 3e38f}, gmax[3] = {-3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int j = 0; j < RBF_Q; j++) {
    const float4 q = spts[min(q_base + j, n - 1)];
    qx[j] = q.x; qy[j] = q.y; qz[j] = q.z; qo[j] = __float_as_int(q.w);
    gmin[0] = fminf(gmin[0], q.x); gmin[1] = fminf(gmin[1], q.y); gmin[2] = fminf(gmin[2], q.z);
    gmax[0] = fmaxf(gmax[0], q.x); gmax[1] = fmaxf(gmax[1], q.y); gmax[2] = fmaxf(gmax[2], q.z);
    sw[j] = sx[j] = sy[j] = sz[j] = sxx[j] = sxy[j] = sxz[j] = syy[j] = syz[j] = szz[j] = 0.f;
  }
  for (int chunk = 0; chunk < ntiles; chunk += 64) {
    const int t = chunk + lane;
    const float lb = box_gap_sq(bbox[2 * min(t, ntiles - 1)], bbox[2 * min(t, ntiles - 1) + 1], gmin, gmax);
    unsigned long long mask = __ballot(t < ntiles && lb <= max_dist_sq);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float4 p = load_candidate(spts, ((chunk + src) << 6) + lane, n);
#pragma unroll
      for (int j = 0; j < RBF_Q; j++) {
        const float dx = p.x - qx[j], dy = p.y - qy[j], dz = p.z - qz[j];
        const float sq = sqdist_nofma(p, qx[j], qy[j], qz[j]);
        const float w = (sq > max_dist_sq) ? 0.f : __expf(-kernel_width * sq);
        sw[j] += w;
        const float wx = w * dx, wy = w * dy, wz = w * dz;
        sx[j] += wx; sy[j] += wy; sz[j] += wz;
        sxx[j] += wx * dx; sxy[j] += wx * dy; sxz[j] += wx * dz; syy[j] += wy * dy; syz[j] += wy * dz; szz[j] += wz * dz;
      }
    }
  }
  for (int j = 0; j < RBF_Q; j++) {
    const double W = wave_sum((double)sw[j]);
    const double X = wave_sum((double)sx[j]), Y = wave_sum((double)sy[j]), Z = wave_sum((double)sz[j]);
    const double XX = wave_sum((double)sxx[j]), XY = wave_sum((double)sxy[j]), XZ = wave_sum((double)sxz[j]);
    const double YY = wave_sum((double)syy[j]), YZ = wave_sum((double)syz[j]), ZZ = wave_sum((double)szz[j]);
    if (lane == 0 && q_base + j < n) {
      const double iw = 1.0 / W;
      const double mx = X * iw, my = Y * iw, mz = Z * iw;
      Sym3<double> C;
      C.xx = XX * iw - mx * mx; C.xy = XY * iw - mx * my; C.xz = XZ * iw - mx * mz;
      C.yy = YY * iw - my * my; C.yz = YZ * iw - my * mz; C.zz = ZZ * iw - mz * mz;
      store_cov(cov, qo[j], regularize_cov(C, method));
    }
  }
}
#endif

// Same sums, ONE query per wave with the two box levels of the k-NN kernel (the 8-queries-per-wave version tests every
// tile box against the group's box and ends with 80 wave reductions: 420 us at 100k points against 190 us for k-NN +
// covariance). Lane l accumulates the candidates it sees at position l of every tile within max_dist, in ascending
// tile order -- the arithmetic of cov_rbf_tiled_kernel, term by term (tiles it skips would have contributed weight 0).
// The wave of a query ends with ten wave totals {W, X, Y, Z, XX, ..., ZZ}. They go to `sums` (SoA: sums[k * n + q]) and
// cov_rbf_finish_kernel turns them into the regularised covariance with one THREAD per query: the eigen-decomposition of the
// PLANE regularisation is ~1,000 dependent fp64 instructions, and run by lane 0 of a one-query wave it cost a full wave's issue
// slots per query -- 180 of this kernel's 300 us at 100k points.
__device__ __forceinline__ void rbf_cov_from_sums(double W, double X, double Y, double Z, double XX, double XY, double XZ, double YY, double YZ, double ZZ, int method,
                                                  float4* __restrict__ cov, int index, float4* __restrict__ cov_sorted = nullptr, int pos = 0) {
  const double iw = 1.0 / W;
  const double mx = X * iw, my = Y * iw, mz = Z * iw;
  Sym3<double> C;
  C.xx = XX * iw - mx * mx; C.xy = XY * iw - mx * my; C.xz = XZ * iw - mx * mz;
  C.yy = YY * iw - my * my; C.yz = YZ * iw - my * mz; C.zz = ZZ * iw - mz * mz;
  const Sym3<double> R = regularize_cov(C, method);
  store_cov(cov, index, R);
  if (cov_sorted) store_cov(cov_sorted, pos, R);  // the same record at the point's place along the Morton curve (CloudDev::cov_sorted)
}
__global__ __launch_bounds__(256) void cov_rbf_finish_kernel(const double* __restrict__ sums, const float4* __restrict__ spts, int n, int method, float4* __restrict__ cov, int q_begin, int q_end, float4* __restrict__ cov_sorted = nullptr) {
  const int q = q_begin + blockIdx.x * 256 + threadIdx.x;
  if (q >= min(n, q_end)) return;
  const size_t N = (size_t)n;
  rbf_cov_from_sums(sums[q], sums[N + q], sums[2 * N + q], sums[3 * N + q], sums[4 * N + q], sums[5 * N + q], sums[6 * N + q], sums[7 * N + q], sums[8 * N + q], sums[9 * N + q], method,
                    cov, __float_as_int(spts[q].w), cov_sorted, q);
}
__global__ __launch_bounds__(256) void cov_rbf1_kernel(const float4* __restrict__ spts, const float4* __restrict__ bbox1, const float4* __restrict__ bbox2, int n, float kernel_width,
                                                       float max_dist_sq, int method, float4* __restrict__ cov, int q_begin = 0, int q_end = 0x7fffffff,
                                                       double* __restrict__ sums = nullptr /* [10][n]: leave the regularisation to cov_rbf_finish_kernel */) {
  const int lane = threadIdx.x & 63;
  const int q = q_begin + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (q >= min(n, q_end)) return;
  const int ntiles = (n + 63) >> 6, nsuper = (ntiles + 63) >> 6;
  const float4 qv = spts[q];
  const float qx = read_lane(qv.x, 0), qy = read_lane(qv.y, 0), qz = read_lane(qv.z, 0);
  int dbg_tiles = 0, dbg_boxes = 0;
  float sw = 0.f, sx = 0.f, sy = 0.f, sz = 0.f, sxx = 0.f, sxy = 0.f, sxz = 0.f, syy = 0.f, syz = 0.f, szz = 0.f;
  auto sweep = [&](const float4& p) __attribute__((always_inline)) {
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    const float sq = sqdist_nofma(p, qx, qy, qz);
    const float w = (sq > max_dist_sq) ? 0.f : __expf(-kernel_width * sq);
    dbg_tiles++;
    sw += w;
    const float wx = w * dx, wy = w * dy, wz = w * dz;
    sx += wx; sy += wy; sz += wz;
    sxx += wx * dx; sxy += wx * dy; sxz += wx * dz; syy += wy * dy; syz += wy * dz; szz += wz * dz;
  };
  for (int sc = 0; sc < nsuper; sc += 64) {
    const int s = sc + lane;
    const float lb2 = (s < nsuper) ? point_box_sq(bbox2[2 * s], bbox2[2 * s + 1], qx, qy, qz) : __builtin_inff();
    dbg_boxes += 64;
    unsigned long long smask = __ballot(lb2 <= max_dist_sq);
    while (smask) {
      const int ssrc = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      const int t = ((sc + ssrc) << 6) + lane;
      const float lb = (t < ntiles) ? point_box_sq(bbox1[2 * t], bbox1[2 * t + 1], qx, qy, qz) : __builtin_inff();
      dbg_boxes += 64;
      unsigned long long tmask = __ballot(lb <= max_dist_sq);
      if (!tmask) continue;
      int cur = __ffsll((long long)tmask) - 1;
      tmask &= tmask - 1;
      float4 pcur = load_candidate(spts, ((((sc + ssrc) << 6) + cur) << 6) + lane, n);
      while (true) {  // the next tile is in flight while this one is accumulated
        int nxt = -1;
        float4 pnxt = pcur;
        if (tmask) {
          nxt = __ffsll((long long)tmask) - 1;
          tmask &= tmask - 1;
          pnxt = load_candidate(spts, ((((sc + ssrc) << 6) + nxt) << 6) + lane, n);
        }
        sweep(pcur);
        if (nxt < 0) break;
        pcur = pnxt;
      }
    }
  }
  PAIR_COUNT(2, 64 * dbg_tiles); PAIR_COUNT(3, dbg_boxes); PAIR_COUNT(5, 1);
  (void)dbg_tiles; (void)dbg_boxes;
  const double W = wave_sum((double)sw);
  const double X = wave_sum((double)sx), Y = wave_sum((double)sy), Z = wave_sum((double)sz);
  const double XX = wave_sum((double)sxx), XY = wave_sum((double)sxy), XZ = wave_sum((double)sxz);
  const double YY = wave_sum((double)syy), YZ = wave_sum((double)syz), ZZ = wave_sum((double)szz);
  if (sums) {
    if (lane < 10) {  // every lane holds all ten totals: lane k stores total k
      double v = W;
      v = lane == 1 ? X : v; v = lane == 2 ? Y : v; v = lane == 3 ? Z : v; v = lane == 4 ? XX : v; v = lane == 5 ? XY : v;
      v = lane == 6 ? XZ : v; v = lane == 7 ? YY : v; v = lane == 8 ? YZ : v; v = lane == 9 ? ZZ : v;
      sums[(size_t)lane * n + q] = v;
    }
    return;
  }
  if (lane == 0) rbf_cov_from_sums(W, X, Y, Z, XX, XY, XZ, YY, YZ, ZZ, method, cov, __float_as_int(qv.w));
}

// getFitnessScore on the Morton-sorted clouds: the wave first sweeps the target tile whose box is
// nearest to its (transformed) query box, then every tile that can still beat the current minima.
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void fitness_tiled_kernel(const float4* __restrict__ ssrc, int ns, const float4* __restrict__ stgt, const float4* __restrict__ tbox, int nt,
                                                            const float* __restrict__ T12, double max_range, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * FIT_Q;
  if (q_base >= ns) return;
  const int ntiles = (nt + 63) >> 6;
  float qx[FIT_Q], qy[FIT_Q], qz[FIT_Q], best[FIT_Q];
  float gmin[3] = {3e38f, 3e38f, 3e38f}, gmax[3] = {-3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int j = 0; j < FIT_Q; j++) {
    const float4 p = ssrc[min(q_base + j, ns - 1)];
    qx[j] = transform_row_nofma(p, T12 + 0);
    qy[j] = transform_row_nofma(p, T12 + 4);
    qz[j] = transform_row_nofma(p, T12 + 8);
    gmin[0] = fminf(gmin[0], qx[j]); gmin[1] = fminf(gmin[1], qy[j]); gmin[2] = fminf(gmin[2], qz[j]);
    gmax[0] = fmaxf(gmax[0], qx[j]); gmax[1] = fmaxf(gmax[1], qy[j]); gmax[2] = fmaxf(gmax[2], qz[j]);
    best[j] = __builtin_inff();
  }
  auto sweep_tile = [&](int t) {
    const float4 p = load_candidate(stgt, (t << 6) + lane, nt);
#pragma unroll
    for (int j = 0; j < FIT_Q; j++) {
      float d = sqdist_nofma(p, qx[j], qy[j], qz[j]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) d = fminf(d, __shfl_xor(d, off));
      best[j] = fminf(best[j], d);  // wave-uniform
    }
  };
  // pass 1: nearest box first
  float lb_min = __builtin_inff();
  int t_min = 0;
  for (int chunk = 0; chunk < ntiles; chunk += 64) {
    const int t = chunk + lane;
    const float lb = (t < ntiles) ? box_gap_sq(tbox[2 * t], tbox[2 * t + 1], gmin, gmax) : __builtin_inff();
    if (lb < lb_min) { lb_min = lb; t_min = t; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ol = __shfl_xor(lb_min, off);
    const int ot = __shfl_xor(t_min, off);
    if (ol < lb_min || (ol == lb_min && ot < t_min)) { lb_min = ol; t_min = ot; }
  }
  sweep_tile(t_min);
  // pass 2: everything that can still improve one of the minima
  for (int chunk = 0; chunk < ntiles; chunk += 64) {
    const int t = chunk + lane;
    const float lb = box_gap_sq(tbox[2 * min(t, ntiles - 1)], tbox[2 * min(t, ntiles - 1) + 1], gmin, gmax);
    float bmax = best[0];
#pragma unroll
    for (int j = 1; j < FIT_Q; j++) bmax = fmaxf(bmax, best[j]);
    unsigned long long mask = __ballot(t < ntiles && t != t_min && lb <= bmax);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      bmax = best[0];
#pragma unroll
      for (int j = 1; j < FIT_Q; j++) bmax = fmaxf(bmax, best[j]);
      if (read_lane(lb, src) > bmax) continue;
      sweep_tile(chunk + src);
    }
  }
  double sum = 0.0, cnt = 0.0;
#pragma unroll
  for (int j = 0; j < FIT_Q; j++)
    if (q_base + j < ns && (double)best[j] <= max_range) { sum += (double)best[j]; cnt += 1.0; }
  if (lane == 0) {
    atomicAdd(&out[0], sum);
    atomicAdd(&out[1], cnt);
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// FastGICP correspondences (fast_gicp_impl.hpp:118-156): for every source point the exact nearest
// TARGET POINT of its transformed position (fp32 transform and distance, as the reference's trans.cast<float>() and
// the kd-tree), kept when the squared distance < threshold^2. Same sweep as fitness_tiled_kernel, carrying the
// winner's ORIGINAL index; equal distances -> lower index (the oracle's kd-tree order). corr[original source index] =
// original target index or -1.
// ------------------------------------------------------------------------------------------------
#ifdef FVH_TEST_KERNELS  // superseded kernel kept as a cross-check of the culled one: only in the test build (fast_gicp_amd/build.py: build_test_kernels_lib)
__global__ __launch_bounds__(256) void nn_corr_tiled_kernel(const float4* __restrict__ ssrc, int ns, const float4* __restrict__ stgt, const float4* __restrict__ tbox, int nt,
                                                            const float* __restrict__ T12, double thr_sq, int* __restrict__ corr) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int q_base = wave * FIT_Q;
  if (q_base >= ns) return;
  const int ntiles = (nt + 63) >> 6;
  float qx[FIT_Q], qy[FIT_Q], qz[FIT_Q], best[FIT_Q];
  int besti[FIT_Q], qid[FIT_Q];
  float gmin[3] = {3e38f, 3e38f, 3e38f}, gmax[3] = {-3e38f, -3e38f, -3e38f};
#pragma unroll
  for (int j = 0; j < FIT_Q; j++) {
    const float4 p = ssrc[min(q_base + j, ns - 1)];
    qid[j] = __float_as_int(p.w);
    qx[j] = transform_row_nofma(p, T12 + 0);
    qy[j] = transform_row_nofma(p, T12 + 4);
    qz[j] = transform_row_nofma(p, T12 + 8);
    gmin[0] = fminf(gmin[0], qx[j]); gmin[1] = fminf(gmin[1], qy[j]); gmin[2] = fminf(gmin[2], qz[j]);
    gmax[0] = fmaxf(gmax[0], qx[j]); gmax[1] = fmaxf(gmax[1], qy[j]); gmax[2] = fmaxf(gmax[2], qz[j]);
    best[j] = __builtin_inff();
    besti[j] = 0x7fffffff;
  }
  auto sweep_tile = [&](int t) {
    const float4 p = load_candidate(stgt, (t << 6) + lane, nt);
    const int pid = ((t << 6) + lane < nt) ? __float_as_int(p.w) : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < FIT_Q; j++) {
      const float d = sqdist_nofma(p, qx[j], qy[j], qz[j]);
      float dm = d;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dm = fminf(dm, __shfl_xor(dm, off));
      if (dm > best[j]) continue;  // wave-uniform
      int im = (d == dm) ? pid : 0x7fffffff;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) im = min(im, __shfl_xor(im, off));
      if (dm < best[j] || im < besti[j]) { best[j] = dm; besti[j] = im; }
    }
  };
  float lb_min = __builtin_inff();
  int t_min = 0;
  for (int chunk = 0; chunk < ntiles; chunk += 64) {
    const int t = chunk + lane;
    const float lb = (t < ntiles) ? box_gap_sq(tbox[2 * t], tbox[2 * t + 1], gmin, gmax) : __builtin_inff();
    if (lb < lb_min) { lb_min = lb; t_min = t; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ol = __shfl_xor(lb_min, off);
    const int ot = __shfl_xor(t_min, off);
    if (ol < lb_min || (ol == lb_min && ot < t_min)) { lb_min = ol; t_min = ot; }
  }
  sweep_tile(t_min);
  for (int chunk = 0; chunk < ntiles; chunk += 64) {
    const int t = chunk + lane;
    const float lb = box_gap_sq(tbox[2 * min(t, ntiles - 1)], tbox[2 * min(t, ntiles - 1) + 1], gmin, gmax);
    float bmax = best[0];
#pragma unroll
    for (int j = 1; j < FIT_Q; j++) bmax = fmaxf(bmax, best[j]);
    unsigned long long mask = __ballot(t < ntiles && t != t_min && lb <= bmax);  // <=: an equally near point with a lower index may live there
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      bmax = best[0];
#pragma unroll
      for (int j = 1; j < FIT_Q; j++) bmax = fmaxf(bmax, best[j]);
      if (read_lane(lb, src) > bmax) continue;
      sweep_tile(chunk + src);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < FIT_Q; j++)
      if (q_base + j < ns) corr[qid[j]] = ((double)best[j] < thr_sq) ? besti[j] : -1;
  }
}
#endif

// Same search, ONE query per wave (rounds 2-5; superseded by nn1_group_kernel below and kept in the test build as its cross-check:
// FVH_FIT_MODE=3 / FVH_GICP_NN_MODE=3) -- the 8-queries-per-wave sweep above took 430 us per search at 17k x 17k, this one 98.
// Seed = nearest tile by box distance through the two box levels, then every tile whose box can still hold a point at
// least as near (ties resolve to the lower original index).
#ifdef FVH_TEST_KERNELS
__global__ __launch_bounds__(256) void nn1_corr_kernel(const float4* __restrict__ ssrc, int ns, const float4* __restrict__ stgt, const float4* __restrict__ bbox1,
                                                       const float4* __restrict__ bbox2, int nt, const float* __restrict__ T12, double thr_sq, int* __restrict__ corr,
                                                       float* __restrict__ best_out = nullptr /* getFitnessScore: squared NN distance per query, in the order of ssrc */,
                                                       LmLink lm = LmLink{nullptr, nullptr, nullptr, nullptr, 0} /* device LM (FastGICP): pose and output buffer follow the LM state on the device */) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // (one query per wave; launched with one wave per workgroup, see find_neighbors)
  if (q >= ns) return;
  const int ntiles = (nt + 63) >> 6, nsuper = (ntiles + 63) >> 6;
  const float4 qv = ssrc[q];
  float Tl[12];
  if (lm.phase) {
    // device-resident LM loop: this search belongs to the linearisation the NEXT cost launch will run -- at x0 into the current
    // correspondence buffer (PH_LINEARIZE), or speculatively at the trial pose xi into the other one (fused PH_TRIAL)
    const int phase = *lm.phase;
    if (phase == 2 /* PH_DONE */) return;
    const double* pose = (phase == 0 /* PH_LINEARIZE */) ? lm.x0 : lm.xi;
    const int sel = (phase == 1 /* PH_TRIAL */) ? (*lm.corr_cur ^ 1) : *lm.corr_cur;
    corr += (size_t)sel * lm.corr_stride;
#pragma unroll
    for (int r = 0; r < 3; r++) {  // trans.cast<float>() (fast_gicp_impl.hpp:121)
      Tl[4 * r] = (float)pose[3 * r]; Tl[4 * r + 1] = (float)pose[3 * r + 1]; Tl[4 * r + 2] = (float)pose[3 * r + 2]; Tl[4 * r + 3] = (float)pose[9 + r];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 12; j++) Tl[j] = T12[j];
  }
  const float qx = transform_row_nofma(qv, Tl + 0), qy = transform_row_nofma(qv, Tl + 4), qz = transform_row_nofma(qv, Tl + 8);
  float best = __builtin_inff();
  int besti = 0x7fffffff;
  auto sweep = [&](const float4& p, int base) __attribute__((always_inline)) {
    const float d = sqdist_nofma(p, qx, qy, qz);
    float dm = d;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dm = fminf(dm, __shfl_xor(dm, off));
    if (dm > best) return;  // wave-uniform
    int im = (d == dm && base + lane < nt) ? __float_as_int(p.w) : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) im = min(im, __shfl_xor(im, off));
    if (dm < best || im < besti) { best = dm; besti = im; }
  };
  // wave argmin of (value, index): value ties -> lower index
  auto argmin = [&](float v, int i, float& vo, int& io) __attribute__((always_inline)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off);
      const int oi = __shfl_xor(i, off);
      if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    vo = v; io = i;
  };
  // ---- seed: the tile whose box is nearest ----
  float sv = __builtin_inff();
  int si = 0;
  for (int sc = 0; sc < nsuper; sc += 64) {
    const int s = sc + lane;
    const float lb2 = (s < nsuper) ? point_box_sq(bbox2[2 * s], bbox2[2 * s + 1], qx, qy, qz) : __builtin_inff();
    float v; int i;
    argmin(lb2, s, v, i);
    if (v < sv) { sv = v; si = i; }
  }
  int t_seed;
  {
    const int t = (si << 6) + lane;
    const float lb = (t < ntiles) ? point_box_sq(bbox1[2 * t], bbox1[2 * t + 1], qx, qy, qz) : __builtin_inff();
    float v;
    argmin(lb, t, v, t_seed);
  }
  sweep(load_candidate(stgt, (t_seed << 6) + lane, nt), t_seed << 6);
  // ---- everything that can still be at least as near ----
  for (int sc = 0; sc < nsuper; sc += 64) {
    const int s = sc + lane;
    const float lb2 = (s < nsuper) ? point_box_sq(bbox2[2 * s], bbox2[2 * s + 1], qx, qy, qz) : __builtin_inff();
    unsigned long long smask = __ballot(lb2 <= best);
    while (smask) {
      const int ssrc_l = __ffsll((long long)smask) - 1;
      smask &= smask - 1;
      if (read_lane(lb2, ssrc_l) > best) continue;
      const int t = ((sc + ssrc_l) << 6) + lane;
      const float lb = (t < ntiles) ? point_box_sq(bbox1[2 * t], bbox1[2 * t + 1], qx, qy, qz) : __builtin_inff();
      unsigned long long tmask = __ballot(lb <= best && t != t_seed);
      if (!tmask) continue;
      int cur = __ffsll((long long)tmask) - 1;
      tmask &= tmask - 1;
      float4 pcur = load_candidate(stgt, ((((sc + ssrc_l) << 6) + cur) << 6) + lane, nt);
      while (true) {  // the next surviving tile is in flight while this one is reduced
        int nxt = -1;
        float4 pnxt = pcur;
        if (tmask) {
          nxt = __ffsll((long long)tmask) - 1;
          tmask &= tmask - 1;
          pnxt = load_candidate(stgt, ((((sc + ssrc_l) << 6) + nxt) << 6) + lane, nt);
        }
        if (read_lane(lb, cur) <= best) sweep(pcur, (((sc + ssrc_l) << 6) + cur) << 6);
        if (nxt < 0) break;
        cur = nxt;
        pcur = pnxt;
      }
    }
  }
  if (lane == 0) {
    if (corr) corr[__float_as_int(qv.w)] = ((double)best < thr_sq) ? besti : -1;
    if (best_out) best_out[q] = best;
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// Exact 1-NN, FOUR queries per wave: one query per ROW of 16 lanes (DPP row operations are VALU moves -- a row all-reduce is four
// instructions, where a 64-lane reduction through the LDS crossbar is six dependent ~100-cycle hops). Each row walks the target's two box
// levels 16 boxes at a time and sweeps a surviving tile 16 points at a time; rows diverge freely (a row is wholly active or wholly masked,
// and nothing in here crosses rows: no ballot, no readlane). Per (query, tile) ~16 instructions instead of ~60 + 12 crossbar hops.
// Replaces the one-query-per-wave search above for getFitnessScore and FastGICP's correspondences (98 -> 20 us for 17k queries in 17k points,
// 789 -> 165 us for 100k queries in the 1M-point map). Also tried in round 6 and dropped (HISTORY.md): 64 queries per wave, one per lane, tiles
// broadcast through LDS -- the least work per distance, and 102 us anyway: the 64 queries of a Morton tile together need ~130 target tiles, each
// of which is then swept for all 64 lanes.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned row_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
// all-reduce over the 16 lanes of a row: lane ^ 1, lane ^ 2 (quad permutes), then the mirrors of 8 and of 16 lanes
#define FVH_ROW_ALLREDUCE(x, OP)                                  \
  do {                                                            \
    x = OP(x, row_dpp<0xB1>(x));  /* quad_perm [1,0,3,2] */       \
    x = OP(x, row_dpp<0x4E>(x));  /* quad_perm [2,3,0,1] */       \
    x = OP(x, row_dpp<0x141>(x)); /* row_half_mirror */           \
    x = OP(x, row_dpp<0x140>(x)); /* row_mirror */                \
  } while (0)
__device__ __forceinline__ unsigned u32_min(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned u32_or(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned row_min_u32(unsigned x) { FVH_ROW_ALLREDUCE(x, u32_min); return x; }
__device__ __forceinline__ unsigned row_or_u32(unsigned x) { FVH_ROW_ALLREDUCE(x, u32_or); return x; }
template <int CTRL>
__device__ __forceinline__ unsigned long long row_min_step64(unsigned long long k) {
  const unsigned long long o = ((unsigned long long)row_dpp<CTRL>((unsigned)(k >> 32)) << 32) | row_dpp<CTRL>((unsigned)k);
  return o < k ? o : k;
}
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long k) {
  k = row_min_step64<0xB1>(k); k = row_min_step64<0x4E>(k); k = row_min_step64<0x141>(k); k = row_min_step64<0x140>(k);
  return k;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void nn1_rows_kernel(const float4* __restrict__ ssrc, int ns, const float4* __restrict__ stgt, const float4* __restrict__ bbox1,
                                                       const float4* __restrict__ bbox2, int nt, const float* __restrict__ T12, double thr_sq, int* __restrict__ corr,
                                                       float* __restrict__ best_out = nullptr /* getFitnessScore: squared NN distance per query, in the order of ssrc */,
                                                       LmLink lm = LmLink{nullptr, nullptr, nullptr, nullptr, 0} /* device LM (FastGICP): pose and output buffer follow the LM state on the device */,
                                                       const float4* __restrict__ tpts = nullptr /* the target in ORIGINAL order: the id this row found last time seeds the bound (see below) */) {
  const int rl = threadIdx.x & 15;                                  // lane within the row
  const int q = (blockIdx.x * 256 + (int)threadIdx.x) >> 4;         // this row's query (position in the source's Morton order)
  if (q >= ns) return;                                              // (whole rows)
  const int ntiles = (nt + 63) >> 6, nsuper = (ntiles + 63) >> 6;
  const float4 qv = ssrc[q];
  float Tl[12];
  const int* seed_ids = corr;  // what this buffer held before this search (the host-driven route: the correspondences of the last call)
  if (lm.phase) {
    // device-resident LM loop: this search belongs to the linearisation the NEXT cost launch will run -- at x0 into the current
    // correspondence buffer (PH_LINEARIZE), or speculatively at the trial pose xi into the other one (fused PH_TRIAL)
    const int phase = *lm.phase;
    if (phase == 2 /* PH_DONE */) return;
    const double* pose = (phase == 0 /* PH_LINEARIZE */) ? lm.x0 : lm.xi;
    const int sel = (phase == 1 /* PH_TRIAL */) ? (*lm.corr_cur ^ 1) : *lm.corr_cur;
    seed_ids = corr + (size_t)*lm.corr_cur * lm.corr_stride;  // the ids of the last accepted linearisation
    corr += (size_t)sel * lm.corr_stride;
#pragma unroll
    for (int r = 0; r < 3; r++) {  // trans.cast<float>() (fast_gicp_impl.hpp:121)
      Tl[4 * r] = (float)pose[3 * r]; Tl[4 * r + 1] = (float)pose[3 * r + 1]; Tl[4 * r + 2] = (float)pose[3 * r + 2]; Tl[4 * r + 3] = (float)pose[9 + r];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 12; j++) Tl[j] = T12[j];
  }
  const float qx = transform_row_nofma(qv, Tl + 0), qy = transform_row_nofma(qv, Tl + 4), qz = transform_row_nofma(qv, Tl + 8);
  unsigned long long best = 0x7f800000ffffffffull;  // (distance bits, original index): +inf and the largest index -- above every candidate, below every NaN distance (never taken)
  // Seed: between two LM transitions the pose moves by millimetres and nearly every point keeps its neighbour. The point this row found LAST
  // time is a candidate like any other -- its key under the NEW pose is an upper bound from the start, and the walk below visits only what
  // can still beat (or tie) it. Whatever the buffer holds (another pair's ids after a swap, nothing at all) is harmless: any index inside the
  // target is a real candidate, anything else is ignored. The result is the minimum over the same total order: identical ids.
  if (tpts && seed_ids) {
    const int si = seed_ids[__float_as_int(qv.w)];
    if ((unsigned)si < (unsigned)nt) {
      const unsigned long long sk = knn_key(sqdist_nofma(tpts[si], qx, qy, qz), si);
      best = sk < best ? sk : best;
    }
  }
  // the 64 points of a tile, 16 per step, against the row's query; afterwards every lane of the row holds the row's minimum
  auto sweep = [&](int tile) __attribute__((always_inline)) {
    const int base = tile << 6;
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; u++) p[u] = stgt[min(base + 16 * u + rl, nt - 1)];
    unsigned long long m = best;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned long long key = knn_key(sqdist_nofma(p[u], qx, qy, qz), __float_as_int(p[u].w));
      m = (base + 16 * u + rl < nt && key < m) ? key : m;
    }
    best = row_min_u64(m);
  };
  // Boxes are visited NEAREST FIRST at both levels. In index order the walk sweeps whatever passes the bound of the moment: measured on the
  // bundled pair, 44 tiles per query pass the bound the first tile leaves (90th percentile: 156) where 5 pass the final one -- a query inside
  // a 2 m tile box is often far from every point of it. A bound travels as its float bits (non-negative floats order like unsigned integers;
  // a NaN bound -- a non-finite query -- lies above +inf and is never a candidate) with the box number in its low bits, so that a row
  // minimum is an arg-min; the truncated mantissa bits only ever LOWER a bound (a box may be visited needlessly, never skipped wrongly).
  const auto best_bits = [&]() __attribute__((always_inline)) { return (unsigned)(best >> 32); };
  // one super tile: its 64 tile boxes, four per lane (eight loads, one round trip), then tiles in ascending bound order until the
  // nearest remaining one lies beyond the row's minimum
  auto process_super = [&](int sup) __attribute__((always_inline)) {
    unsigned key[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int t = (sup << 6) + 16 * u + rl;
      const int tc = min(t, ntiles - 1);
      const float4 l4 = bbox1[2 * tc], h4 = bbox1[2 * tc + 1];
      key[u] = (t < ntiles) ? ((__float_as_uint(point_box_sq(l4, h4, qx, qy, qz)) & ~63u) | (unsigned)(16 * u + rl)) : ~0u;
    }
    while (true) {
      const unsigned m = row_min_u32(min(min(key[0], key[1]), min(key[2], key[3])));
      if ((m & ~63u) > best_bits()) break;  // (also the end of the list: ~0u; and a NaN bound: above +inf)
      const int tl = (int)(m & 63u);
      sweep((sup << 6) + tl);
#pragma unroll
      for (int u = 0; u < 4; u++) key[u] = (16 * u + rl == tl) ? ~0u : key[u];
    }
  };
  // super tiles in chunks of 256 (a 1M-point map is one chunk): sixteen bounds per lane, all loads of a chunk in flight together
  constexpr int CHUNK = 256;
  unsigned k2[16];
  auto load_chunk = [&](int cb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      k2[u] = ~0u;
      if (cb + 16 * u < nsuper) {  // (uniform)
        const int s = cb + 16 * u + rl;
        const int sc = min(s, nsuper - 1);
        const float4 l4 = bbox2[2 * sc], h4 = bbox2[2 * sc + 1];
        k2[u] = (s < nsuper) ? ((__float_as_uint(point_box_sq(l4, h4, qx, qy, qz)) & ~255u) | (unsigned)(16 * u + rl)) : ~0u;
      }
    }
  };
  auto chunk_min = [&]() __attribute__((always_inline)) {
    unsigned m = k2[0];
#pragma unroll
    for (int u = 1; u < 16; u++) m = min(m, k2[u]);
    return row_min_u32(m);
  };
  // ---- the super tile nearest to the query first: whatever is near is most likely in there, and the bound is tight from the start ----
  int s_first = 0;
  {
    unsigned bk = ~0u;
    for (int cb = 0; cb < nsuper; cb += CHUNK) {
      load_chunk(cb);
      const unsigned m = chunk_min();
      if (m < bk) { bk = m; s_first = cb + (int)(m & 255u); }
    }
    s_first = min(s_first, nsuper - 1);  // (a non-finite query: every bound is NaN or the list is empty -- any super tile will do)
  }
  process_super(s_first);
  // ---- then every other super tile that can still matter, nearest first within a chunk ----
  for (int cb = 0; cb < nsuper; cb += CHUNK) {
    if (nsuper > CHUNK) load_chunk(cb);  // (a single chunk is still in registers)
#pragma unroll
    for (int u = 0; u < 16; u++) k2[u] = (cb + 16 * u + rl == s_first) ? ~0u : k2[u];
    while (true) {
      const unsigned m = chunk_min();
      if ((m & ~255u) > best_bits()) break;
      const int sl = (int)(m & 255u);
      process_super(cb + sl);
#pragma unroll
      for (int u = 0; u < 16; u++) k2[u] = (16 * u + rl == sl) ? ~0u : k2[u];
    }
  }
  if (rl == 0) {
    const float bd = __uint_as_float((unsigned)(best >> 32));
    if (corr) corr[__float_as_int(qv.w)] = ((double)bd < thr_sq) ? (int)(unsigned)best : -1;
    if (best_out) best_out[q] = bd;
  }
}

// pcl::Registration::getFitnessScore: mean of the squared nearest-neighbour distances not above max_range. One workgroup,
// fixed summation order (thread t sums entries t, t + 1024, ... in fp64, then a fixed tree): bit-reproducible, unlike
// per-wave atomics. out[0] = sum, out[1] = count.
__global__ __launch_bounds__(1024) void fitness_reduce_kernel(const float* __restrict__ best, int n, double max_range, double* __restrict__ out) {
  __shared__ double s_sum[16], s_cnt[16];
  // thread t: entries t, t + 1024, ... in four interleaved partial sums (a fixed order; the loads of a round are independent -- as one
  // dependent chain a 17k-point cloud was 17 round trips on the only workgroup of the launch)
  double ps[4] = {0.0, 0.0, 0.0, 0.0}, pc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i0 = threadIdx.x; i0 < n; i0 += 4096) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = best[min(i0 + 1024 * u, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double d = (double)v[u];
      if (i0 + 1024 * u < n && d <= max_range) { ps[u] += d; pc[u] += 1.0; }
    }
  }
  double sum = (ps[0] + ps[1]) + (ps[2] + ps[3]), cnt = (pc[0] + pc[1]) + (pc[2] + pc[3]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_xor(sum, off); cnt += __shfl_xor(cnt, off); }
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int w = 0; w < 16; w++) { a += s_sum[w]; c += s_cnt[w]; }
    out[0] = a;
    out[1] = c;
  }
}

// One 64-byte record per target point in the layout of a voxel bucket ({key (unused), q1 = point + count 1, q2/q3 =
// covariance}), so cost_kernel<Real, MODE_VGICP> evaluates FastGICP's cost unchanged: weight sqrt(1) = 1, mean = point.
__global__ __launch_bounds__(256) void gicp_records_kernel(const float4* __restrict__ pts, const float4* __restrict__ cov, int n, float4* __restrict__ table) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  table[4 * (size_t)i + 0] = make_float4(0.f, 0.f, __int_as_float(1), 0.f);
  table[4 * (size_t)i + 1] = make_float4(p.x, p.y, p.z, 1.0f);
  table[4 * (size_t)i + 2] = cov[2 * (size_t)i];
  const float4 c1 = cov[2 * (size_t)i + 1];
  table[4 * (size_t)i + 3] = make_float4(c1.x, c1.y, __int_as_float(__double2loint(1.0)), __int_as_float(__double2hiint(1.0)));  // .zw = weight sqrt(1) as a double
}

// float xyz (stride 3 or 4) -> float4 (w = 0); optionally reduces the cloud's bounding cube into box[0..2] = ~ordered(min),
// box[3..5] = ordered(max) (zero-initialised by the host; both are atomicMax) for the cooperative sort
inline int pack_grid(int n, bool with_box) { const int b = (n + 255) / 256; return with_box ? (b < 128 ? b : 128) : b; }
__global__ __launch_bounds__(256) void pack_points_kernel(const float* __restrict__ xyz, int n, int stride, float4* __restrict__ out, unsigned* __restrict__ box) {
  __shared__ float s_lo[4][3], s_hi[4][3];
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  // grid-stride: with a bounding cube to reduce the grid is capped (pack_grid()) -- every workgroup ends with six atomics on the
  // same six words, ~12 ns apart at the memory-side atomic unit: 391 workgroups of a 100k-point cloud spent 5 of 11 us queueing there
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = xyz[(size_t)i * stride], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
    out[i] = make_float4(x, y, z, 0.f);
    lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
    hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
  }
  if (!box) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
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
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const float l = fminf(fminf(s_lo[0][a], s_lo[1][a]), fminf(s_lo[2][a], s_lo[3][a]));
    const float h = fmaxf(fmaxf(s_hi[0][a], s_hi[1][a]), fmaxf(s_hi[2][a], s_hi[3][a]));
    if (l <= h) {
      atomicMax(&box[a], ~float_to_ordered(l));
      atomicMax(&box[3 + a], float_to_ordered(h));
    }
  }
}

}  // namespace fvh
