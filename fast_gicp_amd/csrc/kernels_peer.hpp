// Multi-GPU exchange through peer-mapped memory: one process per GPU, every rank owns an "exchange
// region" of fine-grained device memory that all other ranks of the node map (hipIpc handles between processes, plain
// pointers between handles of one process). Nothing here calls RCCL: the two exchanges of the registration path are
//   (1) the 32-double normal-equation block {err, b, H} once per cost evaluation -- latency bound (256 B): every rank
//       WRITES its block + a tag straight into every peer's mailbox (xGMI is point to point: one hop, no ring) and polls its
//       own; the sum is taken in rank order, so all ranks hold bit-identical sums and run the LM step redundantly (no
//       broadcast). It is a device function called from INSIDE the cost kernel, so a sharded align stays one persistent
//       launch per rank.  Measured (tools/probes/probe_ipc_mailbox.hip, two processes on one MI355X): 1.4 us per exchange.
//   (2) the covariances of a cloud after the sharded k-NN / RBF estimation -- every rank computed its Morton tile; a rank
//       packs its tile into its staging area, signals a generation number to all peers, and a gather kernel reads the other
//       tiles directly from the peers' staging areas.
// Region layout (bytes): [0, 5120) mailbox = 2 parities x 8 ranks x 40 doubles (32 sums, tag, pad)
//                        [5120, 5184) stage generation per source rank (u64)     [8192, ...) staging, two halves
#pragma once
#include "dev_math.hpp"

namespace fvh {

constexpr int FVH_MAX_PEERS = 8;
constexpr int MAIL_SLOT = 40;  // doubles
constexpr size_t PEER_MAIL_BYTES = 2 * FVH_MAX_PEERS * MAIL_SLOT * sizeof(double);
constexpr size_t PEER_SIG_OFFSET = PEER_MAIL_BYTES;
constexpr size_t PEER_CHECK_OFFSET = PEER_SIG_OFFSET + 64;  // self-check nonce per source rank (u64 x 8), see peer_check_*_kernel
constexpr size_t PEER_STAGE_OFFSET = 8192;
static_assert(PEER_CHECK_OFFSET + 64 <= PEER_STAGE_OFFSET, "exchange region header");

struct PeerView {  // kernel argument (by value)
  int n, rank;
  unsigned long long xbase;  // exchange counter of the first exchange of this launch; exchange x uses tag x and parity x & 1
  char* region[FVH_MAX_PEERS];
};

// region pointer of rank p WITHOUT indexing the by-value struct dynamically (an indexed kernel-argument array is copied to
// scratch memory: 8 selects instead)
__device__ __forceinline__ char* peer_region(const PeerView& pv, int p) {
  char* r = pv.region[0];
#pragma unroll
  for (int i = 1; i < FVH_MAX_PEERS; i++) r = (p == i) ? pv.region[i] : r;
  return r;
}
__device__ __forceinline__ double* peer_mail(char* region, unsigned parity, int src_rank) {
  return reinterpret_cast<double*>(region) + ((size_t)parity * FVH_MAX_PEERS + src_rank) * MAIL_SLOT;
}

// All-reduce (sum, rank order) of the 32 doubles in LDS `row` over the ranks of `pv`; called by ALL threads of one workgroup
// (>= 64 threads) with uniform arguments. Returns false if a peer did not deliver within `watchdog` ticks of the 100 MHz clock.
__device__ inline bool peer_exchange_sums(const PeerView& pv, double* row, unsigned long long x, unsigned long long watchdog, int tid, int nthreads, int* s_flag) {
  const unsigned parity = (unsigned)(x & 1ull);
  const double tag = (double)x;
  for (int idx = tid; idx < pv.n * 32; idx += nthreads) {
    const int p = idx >> 5, v = idx & 31;
    __hip_atomic_store(&peer_mail(peer_region(pv, p), parity, pv.rank)[v], row[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  if (tid == 0) *s_flag = 1;
  __syncthreads();
  if (tid < pv.n) {
    __hip_atomic_store(&peer_mail(peer_region(pv, tid), parity, pv.rank)[32], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const double* mine = peer_mail(peer_region(pv, pv.rank), parity, tid);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&mine[32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
      if (wall_clock64() - t0 > watchdog) { *s_flag = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  const bool ok = *s_flag != 0;
  if (ok && tid < 32) {
    double s = 0.0;
    char* own = peer_region(pv, pv.rank);
    for (int r = 0; r < pv.n; r++) s += __hip_atomic_load(&peer_mail(own, parity, r)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    row[tid] = s;
  }
  __syncthreads();
  return ok;
}

// ---- self-check: does a store of every rank reach every rank? (run once after attaching, before anything depends on it) ----
// every rank writes `nonce` into slot [its rank] of EVERY region (its own included) ...
__global__ void peer_check_write_kernel(PeerView pv, unsigned long long nonce) {
  if ((int)threadIdx.x < pv.n) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(peer_region(pv, threadIdx.x) + PEER_CHECK_OFFSET) + pv.rank, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
  }
}
// ... and waits until all n slots of its OWN region hold it; missing[0] = bit mask of the ranks whose store never came
__global__ void peer_check_wait_kernel(PeerView pv, unsigned long long nonce, unsigned long long watchdog, unsigned* __restrict__ missing) {
  if ((int)threadIdx.x < pv.n) {
    const unsigned long long* slot = reinterpret_cast<const unsigned long long*>(peer_region(pv, pv.rank) + PEER_CHECK_OFFSET) + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    // the slot is a monotonic generation (a fixed prefix + the number of self-checks so far): a rank that is already one round ahead
    // has overwritten it with a LARGER value of the same prefix -- that is still "its store reached us"
    for (;;) {
      const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((v >> 32) == (nonce >> 32) && v >= nonce) break;
      if (wall_clock64() - t0 > watchdog) { atomicOr(missing, 1u << threadIdx.x); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// ---- covariance all-gather --------------------------------------------------------------------------------------------
// tile [lo, hi) of the Morton order -> this rank's staging half (two float4 per point)
__global__ __launch_bounds__(256) void peer_pack_cov_kernel(const float4* __restrict__ cov, const int* __restrict__ order, int lo, int hi, float4* __restrict__ stage) {
  const int j = lo + blockIdx.x * 256 + threadIdx.x;
  if (j >= hi) return;
  const int i = order[j];
  stage[2 * (size_t)(j - lo)] = cov[2 * (size_t)i];
  stage[2 * (size_t)(j - lo) + 1] = cov[2 * (size_t)i + 1];
}
// (a kernel of its own: the kernel boundary after peer_pack_cov_kernel makes the staging visible before any rank sees the generation)
__global__ void peer_signal_kernel(PeerView pv, unsigned long long gen) {
  __threadfence_system();
  if ((int)threadIdx.x < pv.n)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(peer_region(pv, threadIdx.x) + PEER_SIG_OFFSET) + pv.rank, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ONE workgroup waits until every owner has published generation `gen` (a kernel of its own, in stream order before the
// gather: if the thousands of gather workgroups polled themselves they would fill the GPU with spinning waves and starve the
// kernels of a slower rank that shares it -- measured: 2 ranks on one GPU, 1M points, the late rank never got a CU)
__global__ void peer_wait_kernel(PeerView pv, unsigned long long gen, unsigned long long watchdog, int* __restrict__ err) {
  if ((int)threadIdx.x < pv.n) {
    const unsigned long long* sig = reinterpret_cast<const unsigned long long*>(peer_region(pv, pv.rank) + PEER_SIG_OFFSET) + threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(sig, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < gen) {
      if (wall_clock64() - t0 > watchdog) { atomicAdd(err, 1); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
// every point outside this rank's tile: cov[order[j]] <- the owner's staging (read over xGMI / the local fabric)
__global__ __launch_bounds__(256) void peer_gather_cov_kernel(PeerView pv, size_t stage_offset, float4* __restrict__ cov, const int* __restrict__ order, int n, int chunk,
                                                              const int* __restrict__ err) {
  if (*err) return;  // a peer never published: the host reports it
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int owner = j / chunk;
  if (owner == pv.rank) return;
  const float4* stage = reinterpret_cast<const float4*>(peer_region(pv, owner) + stage_offset);
  const size_t k = (size_t)(j - owner * chunk);
  const int i = order[j];
  cov[2 * (size_t)i] = stage[2 * k];
  cov[2 * (size_t)i + 1] = stage[2 * k + 1];
}

}  // namespace fvh
