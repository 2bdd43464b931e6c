// Gaussian voxel map build for gfx950: spatial-hash insert + LDS-staged bucket accumulation.
//
// Replaces the reference's K10-K17 chain: voxel_coord_kernel,
// voxel_bucket_assignment_kernel (+ host retry loop that drops up to 1 % of the points),
// voxel_coord_select_kernel, accumulate_points_kernel (13 global float atomics per point),
// finalize_voxels_kernel / ndt_finalize_voxels_kernel  (src/fast_gicp/cuda/gaussian_voxelmap.cu:9-289).
//
// HBM layout
//   keys  : capacity x u64 voxel key (EMPTY = ~0). capacity = pow2 >= 4 x (voxel count of the previous
//           build on this handle); first build / overflow fallback: pow2 >= 2 * N_t, which can never
//           overflow. No point is ever dropped: an exhausted probe budget is reported and the host rebuilds.
//           The keys are their own dense array because most probes of the cost kernel MISS (DIRECT27 on
//           a 100k-point scene: 65 %): a miss touches 8 B of a 2 MB array that stays in the XCD's L2
//           instead of a 64-B line of a 16 MB table.
//   table : capacity x 64-byte voxel record, bucket index == key slot (no id indirection):
//           q0 = {key_lo, key_hi, num_points, 0}   q1 = {mean.xyz, (float)num_points}
//           q2 = {c_xx, c_xy, c_xz, c_yy}          q3 = {c_yz, c_zz, 0, 0}
//           Only occupied buckets are ever written or (meaningfully) read: the table is never cleared.
//   acc   : capacity x 10 doubles scratch {sum p (3), sum C or sum pp^T (6), count}; fp64 so the
//           result is independent of the atomic arrival order to ~1e-16.
//   The keys (and the two counters) are double buffered and the finalize pass of build N clears the
//   buffers build N+1 will fill and zeroes the accumulators it consumed, so a steady-state rebuild
//   (swapSourceAndTarget in the reference's loop) is two launches; vm_clear_kernel only runs for the
//   first build at a capacity.
//
// Pass 1 (vm_accumulate): one point per thread. Each workgroup first aggregates its 256 points in
// an LDS mini hash table (ds_cmpst_rtn_b64 claim + ds_add_f64), then flushes each occupied LDS slot
// with ONE global claim (global_atomic_cmpswap_x2) and 10 global_atomic_add_f64 -- instead of 13
// global atomics per point. Scan-ordered clouds put 10-40 points of a block in the same voxel.
// Pass 2 (vm_finalize): one thread per bucket: mean/cov from the sums (+ MIN_EIG for NDT), written
// into the bucket; occupied buckets are also listed compactly (getters, D2D source iteration).
#pragma once
#include "dev_math.hpp"

namespace fvh {

constexpr int VM_LDS_SLOTS = 512;
constexpr int VM_LDS_PROBES = 8;
constexpr int VM_ACC_STRIDE = 10;
// Two accumulator sets per bucket: the workgroup whose CAS CREATED the bucket is its owner and writes its partial sums with
// plain stores; every other contributor adds into the visitor set with memory-side fp64 atomics. In Morton order nearly every
// voxel has one contributing workgroup, so most of the ~10 atomics per (workgroup, voxel) -- 21 of this pass's 33 us at 100k
// points -- become stores. vm_finalize_kernel adds the two sets.
constexpr int VM_ACC_BUCKET = 2 * VM_ACC_STRIDE;
constexpr unsigned VM_MAX_PROBE = 255;

// Claim (or find) the bucket of `key`, linear probing from `slot`. CAS first: an atomic goes to the memory side (the
// XCDs' L2s are not coherent), ~2 us per round trip, and the CAS alone answers both "free" and "already this key".
// `created`: this call's CAS turned an empty bucket into the key's (exactly one caller per bucket and build sees that).
__device__ __forceinline__ unsigned global_claim_from(unsigned long long* keys, unsigned mask, unsigned long long key, unsigned slot, unsigned probes_done, bool& created) {
  const unsigned max_probe = mask < VM_MAX_PROBE ? mask : VM_MAX_PROBE;
  created = false;
  for (unsigned it = probes_done; it <= max_probe; it++) {
    const unsigned long long old = atomicCAS(keys + slot, FVH_EMPTY_KEY, key);
    if (old == FVH_EMPTY_KEY) { created = true; return slot; }
    if (old == key) return slot;
    slot = (slot + 1) & mask;
  }
  return 0xFFFFFFFFu;  // probe budget exhausted: the caller counts it in `dropped` and the host rebuilds at the safe size
}
__device__ __forceinline__ unsigned global_claim(unsigned long long* keys, unsigned mask, unsigned long long key) {
  bool created;
  return global_claim_from(keys, mask, key, hash_slot(key, mask), 0, created);
}

// first build at a capacity (afterwards vm_finalize_kernel leaves everything clean): keys -> EMPTY, accumulators and counters -> 0
__global__ __launch_bounds__(256) void vm_clear_kernel(unsigned long long* __restrict__ keys, double* __restrict__ acc, unsigned capacity, int* __restrict__ counters) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < capacity) keys[i] = FVH_EMPTY_KEY;
  if (i < capacity * 10) reinterpret_cast<uint4*>(acc)[i] = make_uint4(0, 0, 0, 0);  // 2 x 10 doubles = 10 quads per bucket
  if (i < 16) counters[i] = 0;
}

// ---- target map sharded by the ranks' spatial tiles (multi-GPU) --------------------------------------------------
// A rank that walks one spatial tile of the source only ever looks up the voxels around T * tile: its map may hold just those -- the
// voxels of the tile's bounding box (in voxel coordinates, at the initial guess) widened by a halo = the reach of the neighbour offsets +
// a margin for the motion of the pose during the align. A voxel inside the box receives ALL its points (the box is voxel-aligned), so
// its record is the replicated map's. The cost kernel reports (in a spare slot of its sums, so that every rank learns it) if a source
// element ever leaves the inner box -- then some of its neighbours may be missing from the shard -- and the host redoes the align on the
// full map.
struct VmRegion {
  int lo[3], hi[3];              // voxel coordinates this shard holds, inclusive; lo > hi: nothing
  int inner_lo[3], inner_hi[3];  // base voxels whose whole neighbourhood lies inside
  int pad[2];
};
__global__ __launch_bounds__(1024) void vm_region_kernel(const float4* __restrict__ sorted, int lo, int hi, PoseD T, double res, int halo, int reach, VmRegion* __restrict__ out) {
  __shared__ int smin[3][16], smax[3][16];
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff - 1, -0x7fffffff - 1, -0x7fffffff - 1};
  for (int j = lo + (int)threadIdx.x; j < hi; j += 1024) {
    const float4 p = sorted[j];
    const double q[3] = {T.r[0] * p.x + T.r[1] * p.y + T.r[2] * p.z + T.t[0], T.r[3] * p.x + T.r[4] * p.y + T.r[5] * p.z + T.t[1], T.r[6] * p.x + T.r[7] * p.y + T.r[8] * p.z + T.t[2]};
    const double f[3] = {floor(q[0] / res - 0.5), floor(q[1] / res - 0.5), floor(q[2] / res - 0.5)};
    if (!voxel_index_ok(f[0], f[1], f[2])) continue;
#pragma unroll
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], (int)f[a]); mx[a] = max(mx[a], (int)f[a]); }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int o = 32; o >= 1; o >>= 1) { mn[a] = min(mn[a], __shfl_xor(mn[a], o)); mx[a] = max(mx[a], __shfl_xor(mx[a], o)); }
    if ((threadIdx.x & 63) == 0) { smin[a][threadIdx.x >> 6] = mn[a]; smax[a][threadIdx.x >> 6] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int a = 0; a < 3; a++) {
      int l = smin[a][0], h = smax[a][0];
      for (int w = 1; w < 16; w++) { l = min(l, smin[a][w]); h = max(h, smax[a][w]); }
      const bool any = l <= h;
      out->lo[a] = any ? l - halo : 1; out->hi[a] = any ? h + halo : 0;
      out->inner_lo[a] = any ? l - halo + reach : 1; out->inner_hi[a] = any ? h + halo - reach : 0;
    }
    out->pad[0] = out->pad[1] = 0;
  }
}

// MODE 0: VGICP additive (sum of points and of point covariances); MODE 1: NDT (sum of points and p p^T);
// MODE 2: VGICP multiplicative (MultiplicativeGaussianVoxel::append, fast_vgicp_voxel.hpp:86-94: sum of C^-1 p and of C^-1)
template <int MODE>
__global__ __launch_bounds__(256) void vm_accumulate_kernel(const float4* __restrict__ pts, const float4* __restrict__ cov, int n, double res,
                                                            unsigned long long* __restrict__ table_keys, unsigned mask, double* __restrict__ acc,
                                                            int* __restrict__ dropped, const int* __restrict__ order, const VmRegion* __restrict__ region = nullptr,
                                                            int float_coord = 0 /* FVH_COMPUTE_CUDA_COMPAT: the coordinate in float, as vector3_hash.cuh:35-38 */) {
  __shared__ unsigned long long lkey[VM_LDS_SLOTS];
  __shared__ double lacc[VM_LDS_SLOTS * VM_ACC_STRIDE];
  const int tid = threadIdx.x;
  for (int s = tid; s < VM_LDS_SLOTS; s += 256) lkey[s] = FVH_EMPTY_KEY;
  for (int s = tid; s < VM_LDS_SLOTS * VM_ACC_STRIDE; s += 256) lacc[s] = 0.0;
  __syncthreads();

  const int i0 = blockIdx.x * 256 + tid;
  if (i0 < n) {
    const int i = order ? order[i0] : i0;  // Morton order: a workgroup's 256 points share a handful of voxels -> the LDS stage absorbs them
    const float4 p = pts[i];
    // fp64 coordinate, as the CPU reference (fast_vgicp_voxel.hpp:158-160); float_coord: (x.array() / resolution - 0.5).floor() in float like the
    // CUDA class -- the two differ for points within float rounding of a voxel face when the resolution is not a power of two
    double fx, fy, fz;
    if (float_coord) {
      const float rf = (float)res;
      fx = (double)floorf(__fsub_rn(__fdiv_rn(p.x, rf), 0.5f)); fy = (double)floorf(__fsub_rn(__fdiv_rn(p.y, rf), 0.5f)); fz = (double)floorf(__fsub_rn(__fdiv_rn(p.z, rf), 0.5f));
    } else {
      fx = floor((double)p.x / res - 0.5); fy = floor((double)p.y / res - 0.5); fz = floor((double)p.z / res - 0.5);
    }
    const bool ok = voxel_index_ok(fx, fy, fz);
    const int cx = ok ? (int)fx : 0, cy = ok ? (int)fy : 0, cz = ok ? (int)fz : 0;
    bool outside = false;  // a sharded map (VmRegion) holds the voxels of its box only: other points are somebody else's
    if (region && ok) outside = cx < region->lo[0] || cx > region->hi[0] || cy < region->lo[1] || cy > region->hi[1] || cz < region->lo[2] || cz > region->hi[2];
    if (outside) {
    } else if (!ok) {
      // non-finite or absurdly far point: it belongs to no voxel. Counted apart from `dropped` (table overflow, which the
      // host answers with a rebuild at the safe size) so that one lidar NaN cannot fail an align.
      atomicAdd(dropped + 1, 1);
    } else {
      const unsigned long long key = pack_key(cx, cy, cz);
      double v[VM_ACC_STRIDE];
      v[0] = p.x; v[1] = p.y; v[2] = p.z;
      if (MODE == 0) {
        const float4 c0 = cov[2 * i], c1 = cov[2 * i + 1];
        v[3] = c0.x; v[4] = c0.y; v[5] = c0.z; v[6] = c0.w; v[7] = c1.x; v[8] = c1.y;
      } else if (MODE == 2) {
        const float4 c0 = cov[2 * i], c1 = cov[2 * i + 1];
        const Sym3<double> Ci = inverse(Sym3<double>{(double)c0.x, (double)c0.y, (double)c0.z, (double)c0.w, (double)c1.x, (double)c1.y});
        const Vec3<double> cp = mul(Ci, Vec3<double>{(double)p.x, (double)p.y, (double)p.z});
        v[0] = cp.x; v[1] = cp.y; v[2] = cp.z;
        v[3] = Ci.xx; v[4] = Ci.xy; v[5] = Ci.xz; v[6] = Ci.yy; v[7] = Ci.yz; v[8] = Ci.zz;
      } else {
        const double x = p.x, y = p.y, z = p.z;
        v[3] = x * x; v[4] = x * y; v[5] = x * z; v[6] = y * y; v[7] = y * z; v[8] = z * z;
      }
      v[9] = 1.0;
      // stage in the workgroup's LDS mini table
      unsigned slot = hash_slot(key, VM_LDS_SLOTS - 1);
      bool staged = false;
#pragma unroll 1
      for (int pr = 0; pr < VM_LDS_PROBES; pr++) {
        unsigned long long old = atomicCAS(&lkey[slot], FVH_EMPTY_KEY, key);
        if (old == FVH_EMPTY_KEY || old == key) { staged = true; break; }
        slot = (slot + 1) & (VM_LDS_SLOTS - 1);
      }
      if (staged) {
#pragma unroll
        for (int j = 0; j < VM_ACC_STRIDE; j++) atomicAdd(&lacc[slot * VM_ACC_STRIDE + j], v[j]);
      } else {  // LDS table crowded: go straight to the global table
        unsigned b = global_claim(table_keys, mask, key);
        if (b != 0xFFFFFFFFu) {
#pragma unroll
          for (int j = 0; j < VM_ACC_STRIDE; j++) atomicAdd(&acc[(size_t)b * VM_ACC_BUCKET + VM_ACC_STRIDE + j], v[j]);  // (a lone point: always a visitor)
        } else {
          atomicAdd(dropped, 1);
        }
      }
    }
  }
  __syncthreads();
  // flush: one global claim + 10 fp64 atomics per (workgroup, voxel). A thread owns two LDS slots: the first probes of
  // both are in flight together (each is a memory-side round trip).
  static_assert(VM_LDS_SLOTS == 512, "two LDS slots per thread of a 256-thread workgroup");
  unsigned long long fk[2], fold[2];
  unsigned fslot[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    fk[u] = lkey[tid + 256 * u];
    fslot[u] = hash_slot(fk[u], mask);
    fold[u] = FVH_EMPTY_KEY;
    if (fk[u] != FVH_EMPTY_KEY) fold[u] = atomicCAS(table_keys + fslot[u], FVH_EMPTY_KEY, fk[u]);
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
    if (fk[u] == FVH_EMPTY_KEY) continue;
    const int s = tid + 256 * u;
    unsigned b = fslot[u];
    bool created = fold[u] == FVH_EMPTY_KEY;
    if (fold[u] != FVH_EMPTY_KEY && fold[u] != fk[u]) b = global_claim_from(table_keys, mask, fk[u], (fslot[u] + 1) & mask, 1, created);
    if (b == 0xFFFFFFFFu) { atomicAdd(dropped, 1); continue; }
    if (created) {  // owner: nobody else writes this half
      uint4* dst = reinterpret_cast<uint4*>(acc + (size_t)b * VM_ACC_BUCKET);
#pragma unroll
      for (int j = 0; j < VM_ACC_STRIDE / 2; j++) {
        const double lo = lacc[s * VM_ACC_STRIDE + 2 * j], hi = lacc[s * VM_ACC_STRIDE + 2 * j + 1];
        dst[j] = make_uint4((unsigned)__double2loint(lo), (unsigned)__double2hiint(lo), (unsigned)__double2loint(hi), (unsigned)__double2hiint(hi));
      }
    } else {
#pragma unroll
      for (int j = 0; j < VM_ACC_STRIDE; j++) atomicAdd(&acc[(size_t)b * VM_ACC_BUCKET + VM_ACC_STRIDE + j], lacc[s * VM_ACC_STRIDE + j]);
    }
  }
}

// One thread per bucket. MODE 0: AdditiveGaussianVoxel::finalize (fast_vgicp_voxel.hpp:118-121,
// gaussian_voxelmap.cu:164-171). MODE 1: ndt_finalize_voxels_kernel (gaussian_voxelmap.cu:184-193)
// + MIN_EIG regularisation (ndt_cuda.cu:128,139).
constexpr int VM_FIN_THREADS = 1024;  // buckets per workgroup = counter atomics saved
template <int MODE>
__global__ __launch_bounds__(VM_FIN_THREADS) void vm_finalize_kernel(const unsigned long long* __restrict__ keys, uint4* __restrict__ table, unsigned capacity, double* __restrict__ acc,
                                                          int* __restrict__ num_voxels, int* __restrict__ occupied, float4* __restrict__ compact_pts, float4* __restrict__ compact_cov,
                                                          unsigned long long* __restrict__ next_keys, int* __restrict__ next_counters) {
  // NDT (MODE 1) regularises every voxel covariance through an eigen-decomposition (~1,000 dependent fp64 instructions). At a
  // load factor of 0.25 a wave of buckets holds ~16 voxels, i.e. it would pay the decomposition for a quarter-full wave: the
  // workgroup's voxels are compacted through LDS first and regularised by the first ceil(count / 64) waves, one voxel per lane.
  constexpr bool DENSE = (MODE == 1);
  __shared__ int s_wave_cnt[VM_FIN_THREADS / 64], s_base;
  __shared__ double s_cov[DENSE ? VM_FIN_THREADS : 1][7];  // raw covariance + the weight sqrt(n)
  __shared__ unsigned s_bucket[DENSE ? VM_FIN_THREADS : 1];
  const unsigned b = blockIdx.x * VM_FIN_THREADS + threadIdx.x;
  bool live = false;
  float mxf = 0.f, myf = 0.f, mzf = 0.f;
  float4 q2 = make_float4(0, 0, 0, 0), q3 = q2;
  Sym3<double> C = {0, 0, 0, 0, 0, 0};
  double wn = 0.0;
  unsigned long long key = FVH_EMPTY_KEY;
  if (b < capacity) {
    next_keys[b] = FVH_EMPTY_KEY;  // the buffers of the NEXT build (the map before this one is dead)
    if (b < 16) next_counters[b] = 0;
    key = keys[b];
  }
  if (key != FVH_EMPTY_KEY) {
    uint4 q0 = make_uint4((unsigned)key, (unsigned)(key >> 32), 0u, 0u);
    double a[VM_ACC_STRIDE];
    {
      uint4* aq = reinterpret_cast<uint4*>(acc + (size_t)b * VM_ACC_BUCKET);  // 2 x 80 B per bucket (owner sums, visitor sums), 16-B aligned
#pragma unroll
      for (int j = 0; j < VM_ACC_STRIDE / 2; j++) {
        const uint4 v = aq[j], w = aq[VM_ACC_STRIDE / 2 + j];
        a[2 * j] = __hiloint2double((int)v.y, (int)v.x) + __hiloint2double((int)w.y, (int)w.x);
        a[2 * j + 1] = __hiloint2double((int)v.w, (int)v.z) + __hiloint2double((int)w.w, (int)w.z);
        aq[j] = make_uint4(0, 0, 0, 0);  // consumed: clean for the next build
        aq[VM_ACC_STRIDE / 2 + j] = make_uint4(0, 0, 0, 0);
      }
    }
    const double cnt = a[9];
    const double inv = 1.0 / cnt;
    double mx = a[0] * inv, my = a[1] * inv, mz = a[2] * inv;
    if (MODE == 0) {
      C.xx = a[3] * inv; C.xy = a[4] * inv; C.xz = a[5] * inv; C.yy = a[6] * inv; C.yz = a[7] * inv; C.zz = a[8] * inv;
    } else if (MODE == 2) {  // MultiplicativeGaussianVoxel::finalize (fast_vgicp_voxel.hpp:96-102): cov = (sum C^-1)^-1, mean = cov * sum C^-1 p
      C = inverse(Sym3<double>{a[3], a[4], a[5], a[6], a[7], a[8]});
      const Vec3<double> m = mul(C, Vec3<double>{a[0], a[1], a[2]});
      mx = m.x; my = m.y; mz = m.z;
    } else {  // (regularised below, on dense waves)
      C.xx = (a[3] - mx * a[0]) * inv; C.xy = (a[4] - mx * a[1]) * inv; C.xz = (a[5] - mx * a[2]) * inv;
      C.yy = (a[6] - my * a[1]) * inv; C.yz = (a[7] - my * a[2]) * inv; C.zz = (a[8] - mz * a[2]) * inv;
    }
    const int n = (int)cnt;
    q0.z = (unsigned)n;
    q0.w = 0;
    // q3.zw: the GICP weight sqrt(n) of the voxel as a double (fast_vgicp_impl.hpp:149) -- once per voxel here instead of once per
    // correspondence and evaluation in the LM kernel
    wn = sqrt((double)n);
    mxf = (float)mx; myf = (float)my; mzf = (float)mz;
    table[(size_t)b * 4] = q0;
    reinterpret_cast<float4*>(table)[(size_t)b * 4 + 1] = make_float4(mxf, myf, mzf, (float)n);
    live = true;
  }  // occupied bucket
  // Compact list of the occupied buckets: ONE atomic per 1024-bucket WORKGROUP on the voxel counter. (Round 1 had one per wave: 4,096
  // same-address atomics at 100k points / 262k buckets -- the memory-side atomic unit retires them one after the other, ~12 ns
  // each, which was the 50 us this kernel took; per voxel it had been 60k of them.)
  const int wv = threadIdx.x >> 6;
  const unsigned long long mask = __ballot(live);
  const int slot_in_wave = __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
  if ((threadIdx.x & 63) == 0) s_wave_cnt[wv] = __popcll(mask);
  __syncthreads();
  int total = 0, local = slot_in_wave;  // voxels of this workgroup; this voxel's position among them
  for (int w = 0; w < VM_FIN_THREADS / 64; w++) { const int c = s_wave_cnt[w]; total += c; local += (w < wv) ? c : 0; }
  if (threadIdx.x == 0) s_base = total ? atomicAdd(num_voxels, total) : 0;
  if (DENSE && live) {
    s_cov[local][0] = C.xx; s_cov[local][1] = C.xy; s_cov[local][2] = C.xz; s_cov[local][3] = C.yy; s_cov[local][4] = C.yz; s_cov[local][5] = C.zz; s_cov[local][6] = wn;
    s_bucket[local] = b;
  }
  __syncthreads();
  float4* tf = reinterpret_cast<float4*>(table);
  if (DENSE) {
    // one voxel per lane of the first waves: regularise (ndt_cuda.cu:128,139: MIN_EIG), write the covariance half of the record
    if ((int)threadIdx.x < total) {
      const int t = threadIdx.x;
      const Sym3<double> R = regularize_cov(Sym3<double>{s_cov[t][0], s_cov[t][1], s_cov[t][2], s_cov[t][3], s_cov[t][4], s_cov[t][5]}, 1 /* MIN_EIG */);
      const unsigned bt = s_bucket[t];
      const float4 r2 = make_float4((float)R.xx, (float)R.xy, (float)R.xz, (float)R.yy);
      const double w = s_cov[t][6];
      const float4 r3 = make_float4((float)R.yz, (float)R.zz, __int_as_float(__double2loint(w)), __int_as_float(__double2hiint(w)));
      tf[(size_t)bt * 4 + 2] = r2;
      tf[(size_t)bt * 4 + 3] = r3;
      if (compact_pts) {  // D2D NDT: the source voxels are the "source cloud"
        const int id = s_base + t;
        compact_cov[2 * id] = r2;
        compact_cov[2 * id + 1] = r3;
      }
    }
    if (!live) return;
    const int id = s_base + local;
    occupied[id] = (int)b;
    if (compact_pts) compact_pts[id] = make_float4(mxf, myf, mzf, 0.f);
    return;
  }
  if (!live) return;
  q2 = make_float4((float)C.xx, (float)C.xy, (float)C.xz, (float)C.yy);
  q3 = make_float4((float)C.yz, (float)C.zz, __int_as_float(__double2loint(wn)), __int_as_float(__double2hiint(wn)));
  tf[(size_t)b * 4 + 2] = q2;
  tf[(size_t)b * 4 + 3] = q3;
  const int id = s_base + local;
  occupied[id] = (int)b;
  if (compact_pts) {
    compact_pts[id] = make_float4(mxf, myf, mzf, 0.f);
    compact_cov[2 * id] = q2;
    compact_cov[2 * id + 1] = q3;
  }
}

// ---- canonical order of a map's voxel list (multi-GPU NDT D2D) -----------------------------------------------------------------
// The compact list of a map (`occupied`, compact_pts / compact_cov) is in the order in which the finalize pass's workgroups took their
// ranges from an atomic counter, and a voxel's bucket depends on the order of the build's CAS races: two ranks that build the same map
// hold the same voxels in different orders. Ranks that share a D2D registration must cut the SAME list (north_star: shard by spatial
// tile), so the voxels are ranked by their key -- z-major, then y, then x: contiguous ranges are slabs of space -- and walked in that
// order: order[rank of voxel i] = i. n_v^2 key compares out of LDS (a few thousand voxels: microseconds), one thread per voxel.
__global__ __launch_bounds__(256) void vm_canonical_order_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ occupied, const int* __restrict__ counters,
                                                                 int* __restrict__ order) {
  __shared__ unsigned long long s_keys[1024];
  const int nv = counters[0];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= nv) return;  // (whole workgroup: uniform)
  const unsigned long long mine = i < nv ? keys[occupied[i]] : 0ull;
  int rank = 0;
  for (int base = 0; base < nv; base += 1024) {
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += 256) s_keys[j] = (base + j < nv) ? keys[occupied[base + j]] : FVH_EMPTY_KEY;  // (EMPTY = ~0: never below a voxel key)
    __syncthreads();
    const int m = min(1024, nv - base);
    for (int j = 0; j < m; j++) rank += (s_keys[j] < mine) ? 1 : 0;
  }
  if (i < nv) order[rank] = i;
}

// ---- occupancy bitmap of large maps ------------------------------------------------------------------------------------------
// 65 % of the DIRECT7 / DIRECT27 probes of a registration MISS (the neighbour voxel does not exist), and on a map that no longer
// fits the L2s every miss still pulls a 64-byte sector of the key table out of HBM for an 8-byte compare: at 1M points the LM
// kernel moved 1.46x its algorithmic bytes that way (PMC, round 2). A dense bit per voxel over the map's bounding box -- 4 x 4 x 4
// voxels per 64-bit word, ~1 MB for a 300 m x 300 m map at 0.5 m -- stays cache resident and answers the misses without touching
// the table; only probes whose bit is set (guaranteed hits) go on to the keys and records. Built on the device after the finalize
// pass (bounds of the occupied voxels -> grid -> clear -> set), for maps of FVH_BITMAP_MIN_POINTS points and up (builds of that
// size are rare; small maps are cache resident anyway).
struct VmGrid {
  unsigned umin[3], umax[3];   // bounds of the occupied voxels, biased coordinates (coord + FVH_COORD_BIAS)
  int b0[3], nb[3];            // first block (biased coordinate >> 2) and number of blocks per axis
  unsigned long long nwords;   // nb[0] * nb[1] * nb[2]
  int enabled;                 // 0: the box needs more words than the budget -- lookups go to the table as before
  int pad_;
};
__global__ void vm_grid_init_kernel(VmGrid* __restrict__ g) {
  if (threadIdx.x < 3) { g->umin[threadIdx.x] = 0xFFFFFFFFu; g->umax[threadIdx.x] = 0u; }
  if (threadIdx.x == 0) { g->enabled = 0; g->nwords = 0; }
}
__global__ __launch_bounds__(256) void vm_grid_bounds_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ occupied, const int* __restrict__ num_voxels,
                                                             VmGrid* __restrict__ g) {
  __shared__ unsigned s_min[3], s_max[3];
  if (threadIdx.x < 3) { s_min[threadIdx.x] = 0xFFFFFFFFu; s_max[threadIdx.x] = 0u; }
  __syncthreads();
  const int nv = *num_voxels;
  unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
    const unsigned long long k = keys[occupied[i]];
    const unsigned c[3] = {(unsigned)(k & 0x1FFFFF), (unsigned)((k >> 21) & 0x1FFFFF), (unsigned)((k >> 42) & 0x1FFFFF)};
#pragma unroll
    for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], c[a]); mx[a] = max(mx[a], c[a]); }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) { atomicMin(&s_min[a], mn[a]); atomicMax(&s_max[a], mx[a]); }
  __syncthreads();
  if (threadIdx.x < 3 && s_min[threadIdx.x] != 0xFFFFFFFFu) { atomicMin(&g->umin[threadIdx.x], s_min[threadIdx.x]); atomicMax(&g->umax[threadIdx.x], s_max[threadIdx.x]); }
}
__global__ void vm_grid_setup_kernel(VmGrid* __restrict__ g, unsigned long long budget_words) {
  if (threadIdx.x != 0) return;
  if (g->umin[0] == 0xFFFFFFFFu) { g->enabled = 0; g->nwords = 0; return; }  // empty map
  unsigned long long n = 1;
  for (int a = 0; a < 3; a++) {
    g->b0[a] = (int)(g->umin[a] >> 2);
    g->nb[a] = (int)(g->umax[a] >> 2) - g->b0[a] + 1;
    n *= (unsigned long long)g->nb[a];
  }
  g->nwords = n;
  g->enabled = n <= budget_words ? 1 : 0;
}
__global__ __launch_bounds__(256) void vm_grid_clear_kernel(unsigned long long* __restrict__ bitmap, const VmGrid* __restrict__ g) {
  if (!g->enabled) return;
  const unsigned long long n = g->nwords;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) bitmap[i] = 0ull;
}
__device__ __forceinline__ bool vm_grid_locate(const VmGrid& g, unsigned ux, unsigned uy, unsigned uz, unsigned long long& word, unsigned& bit) {
  const int bx = (int)(ux >> 2) - g.b0[0], by = (int)(uy >> 2) - g.b0[1], bz = (int)(uz >> 2) - g.b0[2];
  if ((unsigned)bx >= (unsigned)g.nb[0] || (unsigned)by >= (unsigned)g.nb[1] || (unsigned)bz >= (unsigned)g.nb[2]) return false;
  word = ((unsigned long long)bz * (unsigned)g.nb[1] + (unsigned)by) * (unsigned)g.nb[0] + (unsigned)bx;
  bit = (ux & 3u) | ((uy & 3u) << 2) | ((uz & 3u) << 4);
  return true;
}
__global__ __launch_bounds__(256) void vm_grid_set_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ occupied, const int* __restrict__ num_voxels,
                                                          const VmGrid* __restrict__ gp, unsigned long long* __restrict__ bitmap) {
  if (!gp->enabled) return;
  const VmGrid g = *gp;
  const int nv = *num_voxels;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
    const unsigned long long k = keys[occupied[i]];
    unsigned long long word;
    unsigned bit;
    if (vm_grid_locate(g, (unsigned)(k & 0x1FFFFF), (unsigned)((k >> 21) & 0x1FFFFF), (unsigned)((k >> 42) & 0x1FFFFF), word, bit)) atomicOr(&bitmap[word], 1ull << bit);
  }
}

}  // namespace fvh
