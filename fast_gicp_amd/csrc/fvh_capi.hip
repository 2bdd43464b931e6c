// libfast_vgicp_hip.so -- host side of the engine + the extern "C" boundary declared in
// include/fast_vgicp_hip.h. Mirrors the sequencing of the reference's FastVGICPCudaCore
// (src/fast_gicp/cuda/fast_vgicp_cuda.cu) and NDTCudaCore (src/fast_gicp/cuda/ndt_cuda.cu)
// on top of the gfx950 kernels in kernels_*.hpp. No Thrust, no torch types, one HIP stream per handle.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fast_vgicp_hip.h"
#include "kernels_compat.hpp"
#include "kernels_cost.hpp"
#include "kernels_cov.hpp"
#include "kernels_downsample.hpp"
#include "kernels_peer.hpp"
#include "kernels_sort.hpp"
#include "kernels_voxelmap.hpp"

using namespace fvh;

namespace {

constexpr int MAX_COST_BLOCKS = MAX_PARTIAL_ROWS;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct CloudDev {
  int n = 0, k = 0;
  DevBuf box;                          // bounding cube {~ordered(min) x3, ordered(max) x3}, reduced by pack_points_kernel
  DevBuf bbox2;                        // boxes of 64 consecutive tile boxes
  DevBuf order;                        // Morton permutation: order[j] = original index of the j-th point along the curve
  DevBuf pts, cov, nbr, bbox, sorted;  // sorted: Morton-ordered copy, .w = original index; bbox: boxes of its 64-point tiles
  bool has_pts = false, has_cov = false, has_nbr = false, has_sorted = false;
  DevBuf cov_sorted;                   // the covariances again, in Morton order (clouds the LM loop walks in that order: coalesced instead of gathered); has_cov_sorted: of the current cov
  bool has_cov_sorted = false;
  bool nbr_tile_only = false;          // multi-GPU: the neighbour lists exist for this rank's tile only
  bool has_box = false;                // box holds the bounding cube of the CURRENT points (uploads that skip it: NDT, downsampler)
  bool box_dirty = false;              // box holds the cube of a cloud (cleared again by the cooperative sort that consumes it)
  void swap(CloudDev& o) { std::swap(*this, o); }
  void release() { box.release(); pts.release(); cov.release(); nbr.release(); bbox.release(); bbox2.release(); sorted.release(); order.release(); cov_sorted.release(); }
};

// Clouds of this size and up are walked in Morton order (PMC: 2.8x HBM over-fetch on a randomly ordered 100k scan
// against a 1M-point map). Below it everything is L2-resident and the extra index load is not worth it.
constexpr int COHERENT_MIN_POINTS = 32768;
inline const int* coherent_order(const CloudDev& c) {
  static const int min_pts = [] { const char* v = getenv("FVH_COHERENT_MIN_POINTS"); return v ? atoi(v) : COHERENT_MIN_POINTS; }();
  return (c.has_sorted && c.n >= min_pts) ? c.order.as<int>() : nullptr;
}

struct VoxelMapDev {
  double res = 1.0;
  unsigned capacity = 0;
  DevBuf table, acc, occupied, compact_pts, compact_cov;
  DevBuf bitmap, grid;    // occupancy bitmap of a large map + its VmGrid (kernels_voxelmap.hpp); has_bitmap: built for the live map
  bool has_bitmap = false;
  DevBuf canon;           // canonical (key-sorted) order of the compact voxel list (multi-GPU NDT D2D: every rank cuts the same list); has_canon: of the live map
  bool has_canon = false;
  DevBuf compat_keys, compat_idx, compat_seg, compat_hist;  // FVH_COMPUTE_CUDA_COMPAT: (bucket, point index) pairs x 2, run starts per bucket, radix histograms (kernels_compat.hpp)
  DevBuf region;          // VmRegion of a map that holds one rank's shard only (multi-GPU, fvh_vgicp_set_target_map_sharding)
  bool is_shard = false;  // the live map was built through `region`
  DevBuf keys[2];   // voxel keys, double buffered: keys[cur] belongs to the live map, the other one is what the next build fills
  DevBuf counters;  // 2 sets of 16 ints, [0] num_voxels [1] dropped; set `cur` belongs to the live map
  int cur = 0;
  unsigned clean_cap = 0;  // keys[cur ^ 1], counter set cur ^ 1 and acc are clean (EMPTY / 0) over this capacity; 0 = unknown
  int* counters_cur() const { return counters.as<int>() + 16 * cur; }
  const unsigned long long* keys_cur() const { return keys[cur].as<unsigned long long>(); }
  bool valid = false;
  int nv_hint = -1;      // voxel count of the last build seen through a readback; sizes the next table
  int num_skipped = 0;   // points of the last fetched build that belong to no voxel (non-finite / out of the 21-bit range)
  // lazily fetched host copies (getters only)
  bool host_valid = false;
  std::vector<uint4> h_table;
  std::vector<int> h_occupied;
  std::unordered_map<int, int> bucket_to_index;
  void invalidate() { valid = false; host_valid = false; has_bitmap = false; is_shard = false; has_canon = false; }
  void release() { table.release(); acc.release(); occupied.release(); compact_pts.release(); compact_cov.release(); counters.release(); keys[0].release(); keys[1].release(); bitmap.release(); grid.release(); region.release(); canon.release(); compat_keys.release(); compat_idx.release(); compat_seg.release(); compat_hist.release(); clean_cap = 0; has_bitmap = false; is_shard = false; has_canon = false; }
};

struct Profiler {
  bool on = false;
  bool cost_only = false;  // level 2: only the LM / cost launches are bracketed (two events per registration instead of twelve)
  struct Rec { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; };
  std::map<std::string, Rec> recs;
  hipEvent_t begin(const char* cls, hipStream_t s, hipEvent_t* stop_out) {
    Rec& r = recs[cls];
    if (r.used == r.ev.size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      r.ev.push_back({a, b});
    }
    auto& pr = r.ev[r.used++];
    (void)hipEventRecord(pr.first, s);
    *stop_out = pr.second;
    return pr.first;
  }
  void reset() { for (auto& kv : recs) kv.second.used = 0; }
  void destroy() { for (auto& kv : recs) for (auto& pr : kv.second.ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); } recs.clear(); }
};

// RCCL is dlopen'ed on first use so single-GPU users never load it.
struct Rccl {
  struct UID { char b[128]; };  // ncclUniqueId (passed by value)
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, UID, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  bool load() {
    if (lib) return true;
    // reuse the RCCL the process already has (e.g. the one torch.distributed loaded) before loading another copy
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    // RTLD_LOCAL: a process may also hold torch's bundled RCCL; two copies with globally visible symbols
    // interpose each other and corrupt the heap at exit
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    return GetUniqueId && CommInitRank && CommDestroy && AllReduce && AllGather;
  }
};
Rccl g_rccl;
// Gang kernels (the cooperative sort, the persistent LM kernel: grids whose workgroups wait for each other) of two handles could starve
// each other of CU slots (the watchdogs + fall-backs recover, slowly). The persistent LM launches split the slots through the SlotPool
// below; the cooperative sort is used only while no OTHER handle has a gang kernel IN FLIGHT. "In flight" is tracked, not guessed: a
// handle marks itself under the registry's lock before it launches one (check and mark are one atomic step across host threads), records
// an event behind it, and is in flight until that event has fired or the handle has seen its own result (align returned / synchronize).
// A second handle that merely exists (the reference's align.cpp keeps its NDT object alive while the VGICP rows run), or one that the
// same thread uses in turn, costs nothing. (Round 4 used a 20 ms wall-clock window over 64 hashed slots here.)
struct Engine;
struct GangRegistry {
  std::mutex mu;
  std::vector<Engine*> engines;  // registration handles alive in this process
} g_gangs;
std::atomic<int> g_sort_routes[4];  // sorts queued by this process: cooperative kernel, one workgroup, two-launch passes, four-launch passes (fvh_debug_sort_routes)
inline long long steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Co-resident workgroup slots of a device, shared by the persistent LM launches of this process. A persistent grid must be
// resident as a whole, so concurrent aligns (several handles driven by several host threads) SPLIT the slots instead of
// one taking the device and the others falling back to one launch per LM transition: a launch is granted
// min(what it wants, slots / recent concurrency, what is free). Other PROCESSES on the same GPU are invisible here: that
// case is caught by the barrier watchdog and answered with a back-off (Engine::persist_backoff).
struct SlotPool {
  std::mutex mu;
  int reserved[16] = {0}, active[16] = {0}, recent[16] = {0}, calm[16] = {0};
  std::chrono::steady_clock::time_point last_contention[16];
  static constexpr int DECAY_AFTER = 32;       // releases in a row that saw less concurrency than the estimate before the estimate drops by one
  static constexpr int QUIET_RESET_MS = 20;    // no overlapping align / refusal for this long: the burst is over, the estimate starts again from what is active now
  struct Grant { int n = 0; };  // n workgroups, spread over the chip (block b runs on XCD b % 8)
  // (Round 4 could also confine a grant to a subset of the XCDs -- K concurrent aligns sharing the chip XCD by XCD, or a small grid on ONE
  // XCD. Measured, profiles/r04_concurrency.txt / r04_small_grid_layouts.txt: no better than chip-wide grids of cap / K workgroups -- a
  // hand-off costs ~1 us whether or not it crosses XCDs, and a confined launch has to leave room for the blocks that only pass through.
  // Removed from the pool and from the kernel.)
  Grant acquire(int dev, int cap, int want) {  // -> n == 0: use the multi-launch route
    std::lock_guard<std::mutex> lk(mu);
    dev &= 15;
    const auto now = std::chrono::steady_clock::now();
    active[dev]++;
    // A burst of concurrent aligns (a 4-stream leg of a benchmark, a batch of parallel requests) must not throttle the lone aligns that
    // follow it: once nothing has overlapped for QUIET_RESET_MS the estimate is what is active right now. (Round 2 never forgot --
    // a leak; decaying only per calm release kept a lone handle at cap / 4 for its next ~100 aligns.)
    if (active[dev] > 1) last_contention[dev] = now;
    else if (recent[dev] > 1 && now - last_contention[dev] > std::chrono::milliseconds(QUIET_RESET_MS)) { recent[dev] = 1; calm[dev] = 0; }
    recent[dev] = std::max(recent[dev], active[dev]);
    static const int max_split = [] { const char* v = getenv("FVH_SLOT_MAX_SPLIT"); return v ? std::max(1, atoi(v)) : 4; }();
    // (FVH_CONTENDED_SLOT_PCT: the part of the device the concurrent aligns may hold between them -- the rest stays free for the other
    // streams' neighbour searches and sorts, which cannot start on a CU whose register file three resident LM workgroups fill)
    static const int contended_pct = [] { const char* v = getenv("FVH_CONTENDED_SLOT_PCT"); return v ? std::min(100, std::max(10, atoi(v))) : 100; }();
    const bool contended = recent[dev] > 1;
    const int pool = contended ? cap * contended_pct / 100 : cap;
    const int share = std::max(1, pool / std::max(1, std::min(recent[dev], max_split)));
    Grant g;
    g.n = std::min(std::min(want, share), std::max(0, cap - reserved[dev]));
    if (g.n < std::min(want, 32)) {  // too little left to be worth a gang launch
      // a refused request holds nothing and is never released: it must not stay counted in `active` (round 2 leaked it here, and
      // every later persistent launch of the process got cap / min(recent, 4) workgroups for good). The concurrency ESTIMATE keeps
      // the bump: the next grants shrink so that this caller gets its share on the retry; it decays slowly in release().
      active[dev]--;
      calm[dev] = 0;
      last_contention[dev] = now;
      return Grant{};
    }
    reserved[dev] += g.n;
    return g;
  }
  void snapshot(int dev, int* res, int* act, int* rec) {
    std::lock_guard<std::mutex> lk(mu);
    dev &= 15;
    *res = reserved[dev]; *act = active[dev]; *rec = recent[dev];
  }
  void release(int dev, const Grant& g) {
    std::lock_guard<std::mutex> lk(mu);
    dev &= 15;
    reserved[dev] -= g.n;
    active[dev]--;
    // The estimate of the concurrency decays slowly: host threads spend half their time between aligns, so `active` at a release
    // under-reads the contention. (Dropping it at every calm release made four 474-workgroup aligns oscillate: shares grew back to
    // cap / 2, the third thread was refused, and 40 % of its aligns took the multi-launch route.) Under SUSTAINED but lower concurrency
    // the estimate comes down one step per DECAY_AFTER calm releases; once nothing overlaps at all, acquire() resets it (QUIET_RESET_MS).
    if (recent[dev] > active[dev] + 1) {
      if (++calm[dev] >= DECAY_AFTER) { recent[dev]--; calm[dev] = 0; }
    } else {
      calm[dev] = 0;
    }
  }
};
SlotPool g_slots;
// XCD-local hand-offs of the persistent LM kernel (kernels_cost.hpp: xcd_local). On by default; every launch checks that the
// dispatcher placed the members of each group on one XCD (abort code 3 otherwise: the block -> XCD mapping is an observation, not a
// contract), and after XCD_LOCAL_MAX_STRIKES such aborts the process stops asking for it. FVH_XCD_LOCAL=0 never asks for it.
constexpr int XCD_LOCAL_MAX_STRIKES = 3;
std::atomic<int> g_xcd_local_strikes{0};
inline bool xcd_local_wanted() {
  static const bool env_on = [] { const char* v = getenv("FVH_XCD_LOCAL"); return !v || atoi(v) != 0; }();
  return env_on && g_xcd_local_strikes.load() < XCD_LOCAL_MAX_STRIKES;
}
// the layout of one cost launch: both routes of an align take the same (nb, ng), i.e. the same partition and summation order
struct GridPlan { int nb = 0; int ng = 0; int local = 0; };
// Grids of up to SINGLE_LEVEL_MAX_BLOCKS workgroups (NDT D2D over a few thousand source voxels, DIRECT1 at 17k points):
//   FVH_SMALL_GRID_LAYOUT=2 (default)  chip-wide in EIGHT groups like the large grids (XCD-local rows and broadcast, one cross-XCD hand-off);
//                        =0            chip-wide, ONE group: rows -> workgroup 0 -> broadcast, write-through hand-offs (round 3; kept for A/B runs
//                                      and as the second layout the bit-identity tests walk). (=1, one XCD only, was measured worse and is gone.)
// The group count is a function of the grid size alone, so that every route of an align adds the sums in the same order.
inline int small_grid_layout() {
  static const int v = [] { const char* e = getenv("FVH_SMALL_GRID_LAYOUT"); return (e && atoi(e) == 0) ? 0 : 2; }();
  return v;
}
inline int default_groups(int nb) {
  if (nb > SINGLE_LEVEL_MAX_BLOCKS) return TICKET_GROUPS;
  return (small_grid_layout() == 2 && nb >= 2 * TICKET_GROUPS) ? TICKET_GROUPS : 1;
}

struct Engine {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int align_optimizer = 0;  // optimiser of the launches being queued: 0 LM (and every host-driven evaluation), 1 Gauss-Newton (fvh_lm_params::optimizer)
  int precision = FVH_COMPUTE_FP64;
  bool float_cost() const { return precision != FVH_COMPUTE_FP64; }  // FP32 and CUDA_COMPAT: the per-correspondence arithmetic in float (sums in fp64)
  std::vector<int> offsets_host{0, 0, 0};
  int n_off = 1;
  DevBuf fit_best;   // squared nearest-neighbour distance per source point (fitness score)
  DevBuf sort_coop;  // SortCoopState + tagged histograms / elements / final order of the cooperative small sort (kernels_sort.hpp)
  unsigned sort_seq = 0;  // its launch sequence number: the tag of everything one launch hands between workgroups
  DevBuf rbf_sums;   // [10][n] wave totals of the RBF covariance sweep
  bool abort_word_dirty = false;
  int last_persist_blocks = 0;
  bool zero_copy_armed = false;  // the last persistent launch writes its result to result_host
  DevBuf bcast;    // its broadcast rows (tagged with persist_seq, never cleared)
  unsigned long long persist_seq = 0;
  DevBuf offsets_dev, state, partials, ticket, corr, misc, fit, staging, sort_keys, sort_idx, sort_hist;
  void* pinned = nullptr;  // sizeof(LmState) + slack
  // pinned staging of host clouds (upload_cloud): TWO slots used in turn, each with its own "consumed" event -- a target and a
  // source handed over back to back do not wait for each other's consumer
  void* upload_pinned = nullptr;
  size_t upload_pinned_cap = 0;  // bytes per slot
  hipEvent_t upload_done[2] = {nullptr, nullptr};
  bool upload_busy[2] = {false, false};
  int upload_slot = 0;
  bool ensure_upload_pinned(size_t bytes) {
    if (bytes <= upload_pinned_cap) return true;
    for (int s = 0; s < 2; s++) {
      if (upload_busy[s]) { (void)hipEventSynchronize(upload_done[s]); upload_busy[s] = false; }
      if (!upload_done[s] && hipEventCreateWithFlags(&upload_done[s], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    if (upload_pinned) { (void)hipHostFree(upload_pinned); upload_pinned = nullptr; upload_pinned_cap = 0; }
    const size_t want = (bytes + bytes / 4 + 4096 + 255) & ~(size_t)255;
    if (hipHostMalloc(&upload_pinned, 2 * want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); upload_pinned = nullptr; return false; }
    upload_pinned_cap = want;
    return true;
  }
  void* result_host = nullptr;            // mapped pinned memory the persistent kernel writes the final state + a sequence word to
  unsigned long long* result_dev = nullptr;  // its device address
  PoseD lin;               // pose of the last update_correspondences()
  bool has_corr = false;
  int corr_kind = 0;       // 0: voxel correspondences (VGICP / NDT), 1: nearest-point correspondences (GICP)
  int corr_n_src = 0;
  int corr_sel = 0;        // which of the two correspondence buffers the host-mode calls use
  // The rows of the stored correspondences are indexed by the element's POSITION in the walk order (clouds walked in Morton order: a wave's 64
  // items then read and write 64 CONSECUTIVE rows instead of 64 scattered ones -- 1M-map LM launch 186 -> 169 us), else by its original index.
  // Decided by the launch that FINDS them and kept for every later evaluation of the same list (a sort queued in between -- fitness_score,
  // find_neighbors -- may make an order appear: the rows then stay indexed the way they were written); the getters map back to point indices.
  bool corr_by_position = false;
  int last_steps = 0, prev_steps = 0;  // launches the last two aligns needed (odometry loops alternate directions)
  int persist_aborts = 0;               // persistent launches the watchdog turned into multi-launch retries
  int persist_backoff = 0, persist_skip = 0;  // after an abort: aligns to run on the multi-launch route before the next persistent attempt (doubles per consecutive abort)
  Profiler prof;
  bool lm_trace_on = false;  // setDebugPrint: the device LM records one row per trial step
  DevBuf lm_trace;
  int lm_trace_rows = 0;
  void* comm = nullptr;
  int nranks = 1, rank = 0;
  // peer-mapped exchange (kernels_peer.hpp): replaces RCCL for the two exchanges of a sharded registration
  struct PeerComm {
    int n = 1, rank = 0, ranks_on_device = 1;
    char* region = nullptr;            // this handle's exchange region (fine-grained device memory, exported to the peers)
    size_t region_bytes = 0, stage_half_bytes = 0;
    char* peer_region[FVH_MAX_PEERS] = {nullptr};
    bool ipc_opened[FVH_MAX_PEERS] = {false};
    unsigned long long check_gen = 0;  // self-checks so far (fvh_vgicp_peer_selfcheck: collective, same count on every rank)
    int peer_device[FVH_MAX_PEERS] = {-1, -1, -1, -1, -1, -1, -1, -1};  // device an attached region lives on (hipPointerGetAttributes), -1 unknown
    unsigned long long x = 2;          // exchange counter: the next sums exchange uses tag x / parity x & 1 (same on every rank: collective call discipline)
    unsigned long long stage_gen = 0;  // covariance all-gathers so far
    DevBuf err;                        // gather watchdog flag
    bool attached() const { return n > 1; }
    PeerView view(unsigned long long xbase) const {
      PeerView v;
      v.n = n; v.rank = rank; v.xbase = xbase;
      for (int i = 0; i < FVH_MAX_PEERS; i++) v.region[i] = peer_region[i];
      return v;
    }
  } peer;

  // Side stream: swap_source_and_target() rebuilds the target map from a cloud whose data is complete when the stream is known
  // drained (`quiet`: the last call was an align that returned). The two build kernels are then independent of what the caller
  // does next -- upload the new source, sort it, search its neighbours, compute its covariances -- and run beside that chain on
  // `side`; the first call that is not part of that chain (align above all) makes the main stream wait for `side_done`.
  // multi-GPU, either exchange: the engine shards a VGICP cloud internally -- every rank makes the same calls on the same full clouds and
  // works on its spatial tile (a range of the Morton order); the peer path exchanges inside the kernels, the RCCL path between them
  // (NDT handles: the caller names the tile itself -- fvh_ndt_set_source_tile -- with or without a communicator)
  int tile_rank = 0, tile_n = 1;
  bool sharded() const { return peer.attached() || comm != nullptr || tile_n > 1; }
  int shard_ranks() const { return peer.attached() ? peer.n : (tile_n > 1 ? tile_n : (comm ? nranks : 1)); }
  int shard_rank() const { return peer.attached() ? peer.rank : (tile_n > 1 ? tile_rank : (comm ? rank : 0)); }
  DevBuf gather_stage;  // RCCL route: covariances of the whole cloud in Morton order (ncclAllGather in place)
  hipStream_t side = nullptr;
  hipEvent_t side_done = nullptr;
  bool side_pending = false;       // a build on `side` the main stream has not been ordered after yet
  bool quiet = false, was_quiet = false;  // the main stream had drained when the current API call began (set by align on return)
  hipStream_t side_stream() {
    static const bool on = [] { const char* v = getenv("FVH_SIDE_STREAM"); return !v || atoi(v) != 0; }();
    if (!on || comm || peer.attached()) return nullptr;
    if (!side) {
      if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side = nullptr; return nullptr; }
      if (hipEventCreateWithFlags(&side_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(side); side = nullptr; return nullptr; }
    }
    return side;
  }
  int join_side() {
    if (!side_pending) return FVH_OK;
    side_pending = false;
    // As a rule the build ends long before the source chain on the main stream does, and the host -- which runs ahead of the
    // GPU here -- can watch for that at no cost: a wait packet on the main stream instead puts ~6 us of queue processing in
    // front of the LM kernel (kernel trace, profiles/r03_side_stream_knn.md). Bounded: then the packet after all.
    for (int spins = 0; spins < 400; spins++) {
      const hipError_t q = hipEventQuery(side_done);
      if (q == hipSuccess) {  // (kernels launched from here on see the finished map)
        if (spins) (void)hipGetLastError();  // "not ready" must not be what the next launch check reads
        return FVH_OK;
      }
      if (q != hipErrorNotReady) break;
    }
    (void)hipGetLastError();
    hipError_t e = hipStreamWaitEvent(stream, side_done, 0);
    return e == hipSuccess ? FVH_OK : hipfail(e, "hipStreamWaitEvent");
  }
  // The build is not queued by swap itself, nor behind the upload of the next source: at that point of a registration the GPU is
  // waiting for the host (kernel trace: 12 us of idle main stream while the host queued the side work). It is kept here and
  // queued behind the launches of the neighbour search or the covariance pass (after_source_chain_call), which give the host
  // ~100 us of slack; any call outside the source chain runs it on the main stream, in order (settle).
  std::function<int(hipStream_t)> deferred;
  int settle() {
    if (deferred) {
      auto f = std::move(deferred);
      deferred = nullptr;
      const int rc = f(nullptr);
      if (rc) return rc;
    }
    return join_side();
  }
  int after_source_chain_call(int rc) {
    if (!deferred) return rc;
    auto f = std::move(deferred);
    deferred = nullptr;
    int rc2 = f(rc == FVH_OK ? side_stream() : nullptr);
    if (rc2 != FVH_OK && rc == FVH_OK) {
      // the build could not be queued beside the source chain (allocation / launch error): once more, in order on the main stream, so
      // that a failure is at least the build's own and not a side effect of where it was queued; whatever remains is reported as what
      // it is -- the map build swap_source_and_target() deferred -- by the call that had to run it
      (void)hipGetLastError();
      side_pending = false;
      rc2 = f(nullptr);
      if (rc2 != FVH_OK) err = "target voxel map build deferred by swap_source_and_target: " + err;
    }
    return rc ? rc : rc2;
  }
  // A voxel-grid filter that borrowed this engine's stream (fvh_voxelgrid_share_stream_with_*) queues work the engine does not see:
  // the stream is then no longer known to be drained
  Engine* stream_owner = nullptr;
  // the caller has used this handle's DEVICE neighbour search / fitness / RBF covariances before: uploads queue the Morton sort at once
  // (a caller that brings its own neighbour lists -- FastVGICPCuda's default host kd-tree -- never consumes it on small clouds)
  bool device_search_seen = false;

  int fail(int code, const std::string& m) { err = m; return code; }
  int hipfail(hipError_t e, const char* what) { err = std::string(what) + ": " + hipGetErrorString(e); return FVH_ERR_HIP; }

  int init(int dev, bool gang_kernels = true /* false: this engine never launches a persistent / cooperative kernel (voxel-grid filter) */) {
    device = dev;
    hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) return hipfail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) return hipfail(e, "hipStreamCreate");
    owned_stream = stream;
    if ((e = hipHostMalloc(&pinned, sizeof(LmState) + 1024, hipHostMallocDefault)) != hipSuccess) return hipfail(e, "hipHostMalloc");
    if (hipHostMalloc(&result_host, sizeof(LmState) + 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      std::memset(result_host, 0, sizeof(LmState) + 64);
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, result_host, 0) == hipSuccess) result_dev = reinterpret_cast<unsigned long long*>(dp);
    }
    (void)hipGetLastError();  // zero-copy results are optional
    if ((e = state.ensure(sizeof(LmState))) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = partials.ensure(sizeof(double) * PARTIALS_DOUBLES)) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = ticket.ensure(64)) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = bcast.ensure(sizeof(pair_t) * PERSIST_REPLICAS * BCAST_PAIRS)) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = hipMemsetAsync(bcast.p, 0, sizeof(pair_t) * PERSIST_REPLICAS * BCAST_PAIRS, stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(partials.p, 0, sizeof(double) * PARTIALS_DOUBLES, stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");  // tagged rows: tags of no launch
    if ((e = misc.ensure(256)) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = fit.ensure(64)) != hipSuccess) return hipfail(e, "hipMalloc");
    if ((e = hipMemsetAsync(state.p, 0, sizeof(LmState), stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(ticket.p, 0, 64, stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(misc.p, 0, 256, stream)) != hipSuccess) return hipfail(e, "hipMemsetAsync");
    std::memset(&lin, 0, sizeof(lin));
    lin.r[0] = lin.r[4] = lin.r[8] = 1.0;
    int rc = upload_offsets();
    if (rc) return rc;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return hipfail(e, "hipStreamSynchronize");  // "warming up GPU" (fast_vgicp_cuda.cu:19-20)
    if (gang_kernels) {
      if ((e = hipEventCreateWithFlags(&gang_done, hipEventDisableTiming)) != hipSuccess) return hipfail(e, "hipEventCreate");
      std::lock_guard<std::mutex> lk(g_gangs.mu);
      g_gangs.engines.push_back(this);
      counted = true;
    }
    return FVH_OK;
  }
  bool counted = false;             // registered in g_gangs (registration handles; a voxel-grid filter never launches a gang kernel)
  hipEvent_t gang_done = nullptr;   // recorded behind the last gang kernel this handle queued
  std::atomic<int> gang_launching{0};  // > 0: between "marked" and "event recorded"
  std::atomic<bool> gang_open{false};  // a gang kernel was queued and the handle has not seen its result yet (the event says whether it still runs)
  bool gang_in_flight() const {  // (called by OTHER handles under g_gangs.mu)
    if (gang_launching.load(std::memory_order_acquire) > 0) return true;
    if (!gang_open.load(std::memory_order_acquire)) return false;
    const hipError_t q = hipEventQuery(gang_done);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return true; }
    return false;
  }
  // mark this handle as launching a gang kernel; `exclusive`: only if no other handle has one in flight (false = refused, nothing marked)
  bool gang_begin(bool exclusive) {
    if (!counted) return !exclusive;
    std::lock_guard<std::mutex> lk(g_gangs.mu);
    if (exclusive)
      for (Engine* o : g_gangs.engines) if (o != this && o->device == device && o->gang_in_flight()) return false;  // (another device's kernels take none of this one's CU slots)
    gang_launching.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  void gang_end() {  // the gang kernel (and whatever must follow it) is queued
    if (!counted) return;
    (void)hipEventRecord(gang_done, stream);
    gang_open.store(true, std::memory_order_release);
    gang_launching.fetch_sub(1, std::memory_order_acq_rel);
  }
  void gang_clear() { if (counted) gang_open.store(false, std::memory_order_release); }  // the caller holds the result of an align / drained the stream: nothing of this handle is in flight
  hipStream_t owned_stream = nullptr;  // the stream init() made; `stream` may be another handle's (fvh_voxelgrid_share_stream_*)
  void shutdown() {
    if (stream != owned_stream) stream = owned_stream;  // a borrowed stream is its owner's to drain and destroy
    if (counted) {
      std::lock_guard<std::mutex> lk(g_gangs.mu);
      g_gangs.engines.erase(std::remove(g_gangs.engines.begin(), g_gangs.engines.end(), this), g_gangs.engines.end());
      counted = false;
    }
    (void)hipSetDevice(device);
    if (side) (void)hipStreamSynchronize(side);
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
    comm = nullptr;
    peer_detach();
    if (peer.region) { (void)hipFree(peer.region); peer.region = nullptr; }
    peer.err.release();
    prof.destroy();
    gather_stage.release(); lm_trace.release(); fit_best.release(); sort_coop.release(); rbf_sums.release(); bcast.release(); offsets_dev.release(); state.release(); partials.release(); ticket.release(); corr.release(); misc.release(); fit.release(); staging.release(); sort_keys.release(); sort_idx.release(); sort_hist.release();
    if (pinned) (void)hipHostFree(pinned);
    if (upload_pinned) (void)hipHostFree(upload_pinned);
    for (int s = 0; s < 2; s++) if (upload_done[s]) (void)hipEventDestroy(upload_done[s]);
    if (result_host) (void)hipHostFree(result_host);
    if (stream) (void)hipStreamDestroy(stream);
    if (gang_done) (void)hipEventDestroy(gang_done);
    if (side_done) (void)hipEventDestroy(side_done);
    if (side) (void)hipStreamDestroy(side);
  }
  void peer_detach() {
    for (int i = 0; i < FVH_MAX_PEERS; i++) {
      if (peer.ipc_opened[i] && peer.peer_region[i]) (void)hipIpcCloseMemHandle(peer.peer_region[i]);
      peer.peer_region[i] = nullptr; peer.ipc_opened[i] = false;
    }
    peer.n = 1; peer.rank = 0; peer.ranks_on_device = 1;
  }
  int upload_offsets() {
    // device layout: n_off x {dx, dy, dz}, then n_off packed triples (10 bits each, biased by 512): one register per offset in the cost kernel
    std::vector<int> dev(offsets_host);
    for (int o = 0; o < n_off; o++) {
      for (int a = 0; a < 3; a++)
        if (std::abs(offsets_host[3 * o + a]) > 511) return fail(FVH_ERR_INVALID_ARGUMENT, "neighbor offsets beyond +-511 voxels are not supported");
      dev.push_back((offsets_host[3 * o] + 512) | ((offsets_host[3 * o + 1] + 512) << 10) | ((offsets_host[3 * o + 2] + 512) << 20));
    }
    hipError_t e = offsets_dev.ensure(dev.size() * sizeof(int));
    if (e != hipSuccess) return hipfail(e, "hipMalloc");
    e = hipMemcpyAsync(offsets_dev.p, dev.data(), dev.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hipfail(e, "hipMemcpyAsync");
    e = hipStreamSynchronize(stream);  // offsets_host may be reassigned by the caller right after
    if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize");
    has_corr = false;
    return FVH_OK;
  }
  // FastVGICPCudaCore::set_neighbor_search_method (fast_vgicp_cuda.cu:41-94)
  int set_offsets(int method, double radius) {
    std::vector<int> o;
    switch (method) {
      case FVH_DIRECT1: o = {0, 0, 0}; break;
      case FVH_DIRECT7: o = {0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 1, 0, 0, -1}; break;
      case FVH_DIRECT27:
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { o.push_back(i - 1); o.push_back(j - 1); o.push_back(k - 1); }
        break;
      case FVH_DIRECT_RADIUS: {
        if (!(radius >= 0.0) || radius > 511.0) return fail(FVH_ERR_INVALID_ARGUMENT, "DIRECT_RADIUS: radius must be within [0, 511] voxels");  // (NaN too; offsets travel packed, 10 bits per axis)
        int range = (int)std::ceil(radius);
        for (int i = -range; i <= range; i++) for (int j = -range; j <= range; j++) for (int k = -range; k <= range; k++)
          if (std::sqrt((double)(i * i + j * j + k * k)) <= radius + 1e-3) { o.push_back(i); o.push_back(j); o.push_back(k); }
        if (o.empty()) return fail(FVH_ERR_INVALID_ARGUMENT, "DIRECT_RADIUS: no offsets within radius");
      } break;
      default: return fail(FVH_ERR_INVALID_ARGUMENT, "unknown neighbor search method");
    }
    offsets_host = o;
    n_off = (int)o.size() / 3;
    return upload_offsets();
  }
};

#define HIP_OR_FAIL(E, expr)                                   \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) return (E)->hipfail(_e, #expr);      \
  } while (0)

struct ProfScope {
  Engine* e; hipEvent_t stop = nullptr; hipStream_t st;
  ProfScope(Engine* e_, const char* cls, hipStream_t on = nullptr) : e(e_), st(on ? on : e_->stream) {
    if (e->prof.on && (!e->prof.cost_only || std::strcmp(cls, "cost") == 0)) e->prof.begin(cls, st, &stop);
  }
  ~ProfScope() { if (stop) (void)hipEventRecord(stop, st); }
};

inline PoseD pose_from_colmajor16(const double* T) {
  PoseD p;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) p.r[i * 3 + j] = T[j * 4 + i]; p.t[i] = T[12 + i]; }
  return p;
}
inline void pose_to_colmajor16(const PoseD& p, double* T) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[j * 4 + i] = p.r[i * 3 + j]; T[12 + i] = p.t[i]; T[i * 4 + 3] = 0.0; }
  T[15] = 1.0;
}

// ---------------------------------------------------------------------------------------------
// shared building blocks
// ---------------------------------------------------------------------------------------------
int upload_cloud(Engine* e, CloudDev& c, const float* xyz, int n, int stride, bool on_device, bool want_box = true /* the cooperative sort's bounding cube (VGICP clouds) */,
                 hipStream_t on = nullptr /* another stream than the handle's (the prepared-source slot of an NDT handle) */) {
  hipStream_t const st = on ? on : e->stream;
  if (n < 0 || (n > 0 && !xyz)) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_cloud: null points");
  if (stride != 3 && stride != 4) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_cloud: stride must be 3 or 4 floats");
  HIP_OR_FAIL(e, c.pts.ensure(sizeof(float4) * (size_t)std::max(n, 1)));
  c.n = n;
  c.has_pts = true;
  c.has_sorted = false;
  if (n == 0) return FVH_OK;
  unsigned* boxp = nullptr;
  c.has_box = want_box;
  if (want_box) {
    const bool fresh_box = c.box.p == nullptr;
    HIP_OR_FAIL(e, c.box.ensure(64));
    if (fresh_box || c.box_dirty) HIP_OR_FAIL(e, hipMemsetAsync(c.box.p, 0, 64, st));  // (the cooperative sort's finish kernel leaves it cleared)
    c.box_dirty = true;
    boxp = c.box.as<unsigned>();
  }
  // widen `srcp` to float4 (+ bounding cube); `slot`: the pinned upload slot that kernel reads (its "free again" event follows it)
  auto pack = [&](const float* srcp, int slot) -> int {
    pack_points_kernel<<<pack_grid(n, boxp != nullptr), 256, 0, st>>>(srcp, n, stride, c.pts.as<float4>(), boxp);
    HIP_OR_FAIL(e, hipGetLastError());
    if (slot >= 0) { HIP_OR_FAIL(e, hipEventRecord(e->upload_done[slot], st)); e->upload_busy[slot] = true; }
    return FVH_OK;
  };
  if (on_device) {
    int rc = pack(xyz, -1);
    if (rc) return rc;
  } else {
    // H2D the xyz (stride 3) / xyzi (stride 4, e.g. a KITTI .bin buffer) array into a staging buffer, then widen to float4 on device
    const size_t bytes = sizeof(float) * stride * (size_t)n;
    HIP_OR_FAIL(e, e->staging.ensure(bytes));
    static const size_t pinned_max = [] { const char* v = getenv("FVH_PINNED_UPLOAD_MAX"); return v ? (size_t)atoll(v) : (size_t)(8u << 20); }();
    if (bytes <= pinned_max && e->ensure_upload_pinned(bytes)) {
      // the caller's (pageable) buffer is consumed by a plain memcpy into pinned memory of the handle; the copy to the device and
      // everything after it is then truly asynchronous -- no stream synchronisation before returning (the reference's loop hands
      // over a host cloud per registration: this took the PCIe-inclusive rate from 3,220 to the rate below)
      const int us = (e->upload_slot ^= 1);
      char* slot = static_cast<char*>(e->upload_pinned) + (size_t)us * e->upload_pinned_cap;
      if (e->upload_busy[us]) { HIP_OR_FAIL(e, hipEventSynchronize(e->upload_done[us])); e->upload_busy[us] = false; }  // the upload before the last still reading this slot (normally long finished)
      std::memcpy(slot, xyz, bytes);
      // Small clouds: the widening kernel reads the pinned buffer itself, over PCIe (a 17k-point cloud is 0.2-0.3 MB: a few microseconds)
      // -- a copy-engine transfer in front of it costs its own start-up plus a hand-over between the copy and the compute queue,
      // ~20 us of a 250 us registration. Large clouds keep the copy engine (the kernel's PCIe reads would be the slower transfer).
      static const size_t zero_copy_max = [] { const char* v = getenv("FVH_ZEROCOPY_UPLOAD_MAX"); return v ? (size_t)atoll(v) : (size_t)(1u << 20); }();
      void* pinned_dev = nullptr;
      if (bytes <= zero_copy_max && hipHostGetDevicePointer(&pinned_dev, slot, 0) == hipSuccess && pinned_dev) {
        int rc = pack(static_cast<const float*>(pinned_dev), us);
        if (rc) return rc;
      } else {
        (void)hipGetLastError();
        HIP_OR_FAIL(e, hipMemcpyAsync(e->staging.p, slot, bytes, hipMemcpyHostToDevice, st));
        HIP_OR_FAIL(e, hipEventRecord(e->upload_done[us], st));
        e->upload_busy[us] = true;
        int rc = pack(e->staging.as<float>(), -1);
        if (rc) return rc;
      }
    } else {
      HIP_OR_FAIL(e, hipMemcpyAsync(e->staging.p, xyz, bytes, hipMemcpyHostToDevice, st));
      pack_points_kernel<<<pack_grid(n, boxp != nullptr), 256, 0, st>>>(e->staging.as<float>(), n, stride, c.pts.as<float4>(), boxp);
      HIP_OR_FAIL(e, hipGetLastError());
      HIP_OR_FAIL(e, hipStreamSynchronize(st));  // caller may free xyz on return (reference copies too)
    }
  }
  return FVH_OK;
}

int set_neighbors(Engine* e, CloudDev& c, int k, const int* idx) {
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "set_neighbors: cloud not set");
  if (k <= 0 || !idx) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_neighbors: bad k / null");
  // the covariance kernel gathers pts[idx]: an index outside [0, n) (e.g. a -1 pad of a k-NN on fewer than k points) must not reach it
  for (size_t j = 0, m = (size_t)c.n * k; j < m; j++)
    if ((unsigned)idx[j] >= (unsigned)c.n) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_neighbors: neighbour index " + std::to_string(idx[j]) + " outside [0, " + std::to_string(c.n) + ")");
  HIP_OR_FAIL(e, c.nbr.ensure(sizeof(int) * (size_t)c.n * k));
  HIP_OR_FAIL(e, hipMemcpyAsync(c.nbr.p, idx, sizeof(int) * (size_t)c.n * k, hipMemcpyHostToDevice, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  c.k = k;
  c.has_nbr = true;
  return FVH_OK;
}

// Morton-sort the cloud (kernels_sort.hpp) and box its 64-point tiles; cached until the cloud changes.
int ensure_sorted(Engine* e, CloudDev& c) {
  e->device_search_seen = true;
  if (c.has_sorted) return FVH_OK;
  const int n = c.n;
  static const int items_env = [] { const char* v = getenv("FVH_SORT_ITEMS"); return v ? atoi(v) : 0; }();
  const int items = items_env > 0 ? items_env : (n <= 262144 ? 256 : (n <= 1048576 ? 512 : SORT_ITEMS_MAX));  // more, shorter waves for small clouds (latency-bound)
  const int nwaves = (n + items - 1) / items;
  const int ntiles = (n + 63) / 64;
  HIP_OR_FAIL(e, c.sorted.ensure(sizeof(float4) * (size_t)n));
  HIP_OR_FAIL(e, c.bbox.ensure(sizeof(float4) * 2 * (size_t)ntiles));
  HIP_OR_FAIL(e, e->sort_keys.ensure(sizeof(unsigned) * 2 * (size_t)n + 64));
  HIP_OR_FAIL(e, e->sort_idx.ensure(sizeof(int) * (size_t)n));
  HIP_OR_FAIL(e, c.order.ensure(sizeof(int) * (size_t)n));
  HIP_OR_FAIL(e, e->sort_hist.ensure(sizeof(unsigned) * (size_t)RADIX_BINS * (nwaves + 1)));
  const int nsuper_small = (ntiles + 63) / 64;
  HIP_OR_FAIL(e, c.bbox2.ensure(sizeof(float4) * 2 * (size_t)nsuper_small));
  ProfScope ps(e, "sort");
  static const int sort_mode = [] { const char* v = getenv("FVH_SORT_MODE"); return v ? atoi(v) : 2; }();  // 0: multi-kernel radix, 1: single workgroup, 2: cooperative (single-engine processes), 3: cooperative always
  if (sort_mode >= 1 && n <= SORT_SMALL_MAX) {
    // cooperative kernel (32 workgroups meeting at grid barriers) while no OTHER handle has a gang kernel in flight (GangRegistry above:
    // two gang kernels from two streams could starve each other of CU slots; the watchdog + fallback would recover, slowly)
    const bool coop = c.has_box && (sort_mode == 3 ? e->gang_begin(false) : (sort_mode == 2 && e->gang_begin(true)));  // (mode 1: never -- other PROCESSES' gang kernels on a shared GPU are invisible to the registry)
    struct GangEnd { Engine* e; bool on; ~GangEnd() { if (on) e->gang_end(); } } gang_end{e, coop};  // (on every way out: the event behind whatever was queued)
    g_sort_routes[coop ? 0 : 1].fetch_add(1, std::memory_order_relaxed);
    if (coop) {
      const bool fresh = e->sort_coop.p == nullptr;
      HIP_OR_FAIL(e, e->sort_coop.ensure(COOP_STATE_BYTES));
      SortCoopState* cs = e->sort_coop.as<SortCoopState>();
      char* base = reinterpret_cast<char*>(cs);
      unsigned* chist = reinterpret_cast<unsigned*>(cs + 1);
      // once: tags of no launch everywhere (afterwards every launch rewrites the tagged words, and the state words are compared with the launch's number)
      if (fresh) HIP_OR_FAIL(e, hipMemsetAsync(cs, 0, COOP_STATE_BYTES, e->stream));
      if ((++e->sort_seq & (COOP_HTAG_MASK >> 1)) == 0) ++e->sort_seq;  // (a histogram tag of 0 is what fresh memory holds)
      unsigned long long wd = 2'000'000ull;  // 20 ms
      { const char* v = getenv("FVH_SORT_COOP_WATCHDOG_TICKS"); if (v) wd = strtoull(v, nullptr, 10); }  // test hook: 0 forces the fallback
      unsigned long long* celem = reinterpret_cast<unsigned long long*>(base + COOP_ELEM_OFFSET);
      sort_coop_kernel<<<COOP_WGS, COOP_THREADS, 0, e->stream>>>(c.pts.as<float4>(), n, c.order.as<int>(), c.sorted.as<float4>(), c.box.as<unsigned>(), chist, celem, cs, e->sort_seq, wd);
      // tile boxes (one tile per wave) + super boxes (one workgroup each); when the cooperative kernel did not finish, workgroup 0 redoes everything
      const int fin_tile_wgs = (ntiles + 15) / 16;
      sort_coop_finish_kernel<<<fin_tile_wgs + nsuper_small, 1024, 0, e->stream>>>(c.pts.as<float4>(), n, e->sort_keys.as<unsigned>(), c.order.as<int>(), e->sort_keys.as<unsigned>() + n,
                                                                                   e->sort_idx.as<int>(), c.sorted.as<float4>(), c.bbox.as<float4>(), c.bbox2.as<float4>(), cs, c.box.as<unsigned>(),
                                                                                   e->sort_seq, fin_tile_wgs);
      c.box_dirty = false;  // consumed and cleared by the finish kernel
    } else {
      sort_small_kernel<<<1, 1024, 0, e->stream>>>(c.pts.as<float4>(), n, e->sort_keys.as<unsigned>(), c.order.as<int>(), e->sort_keys.as<unsigned>() + n, e->sort_idx.as<int>());
      gather_tiles_kernel<<<(ntiles + 3) / 4, 256, 0, e->stream>>>(c.pts.as<float4>(), c.order.as<int>(), n, c.sorted.as<float4>(), c.bbox.as<float4>());
      super_bbox_kernel<<<(nsuper_small + 3) / 4, 256, 0, e->stream>>>(c.bbox.as<float4>(), ntiles, c.bbox2.as<float4>());
    }
    HIP_OR_FAIL(e, hipGetLastError());
    c.has_sorted = true;
    return FVH_OK;
  }
  // Large clouds: 27-bit Morton keys, stable LSD radix sort. Every kernel of this chain is a dependent stage of >= 5 us whatever it
  // does (a 100k-point cloud is 400 KB of keys): the first histogram kernel computes the keys itself and both box levels come out
  // of one launch (two stages less); clouds up to 256k points are ordered by the top 22 key bits in TWO 11-bit passes (cells of
  // 4 x 4 x 2 fine cells: with a few points per cell the tiles are as compact as with the full key) instead of three 9-bit ones.
  // Measured at 100k points: 75 us (15 stages) -> 73 (13) -> 71 (9): a 2,048-bin pass costs 34 us against 25 for a 512-bin one
  // (the histogram's transposed [bin][wave] write), so the stage count alone buys little.
  unsigned* keys[2] = {e->sort_keys.as<unsigned>(), e->sort_keys.as<unsigned>() + n};
  const bool packed_box = c.has_box;  // the upload already reduced the bounding cube (pack_points_kernel): no memsets, no extra pass over the cloud
  unsigned* box = packed_box ? c.box.as<unsigned>() : reinterpret_cast<unsigned*>(e->sort_keys.as<unsigned>() + 2 * (size_t)n);
  if (!packed_box) {
    HIP_OR_FAIL(e, hipMemsetAsync(box, 0xFF, 12, e->stream));
    HIP_OR_FAIL(e, hipMemsetAsync(box + 3, 0, 12, e->stream));
    cloud_bbox_kernel<<<std::min(256, (n + 255) / 256), 256, 0, e->stream>>>(c.pts.as<float4>(), n, box);
  }
  // Up to SORT_FUSED_MAX points: two launches per pass (kernels_sort.hpp: the scatter derives its cursors from per-workgroup digit counts),
  // two passes over the top 2 x FVH_SORT_FUSED_BITS bits of the key. 100k points: 71 us in nine launches -> see profiles/r04_sort_fused.txt.
  static const int fused_bits = [] { const char* v = getenv("FVH_SORT_FUSED_BITS"); const int b = v ? atoi(v) : 10; return (b == 9 || b == 10) ? b : 0; }();  // 0: the four-launch passes below. (9: the sort is 8 us shorter and the exact k-NN behind it 14 us longer -- coarser cells, looser tiles)
  g_sort_routes[(fused_bits && n <= SORT_FUSED_MAX) ? 2 : 3].fetch_add(1, std::memory_order_relaxed);
  if (fused_bits && n <= SORT_FUSED_MAX) {
    const int fwaves = (n + SORT_FUSED_ITEMS - 1) / SORT_FUSED_ITEMS, fwgs = (fwaves + 3) / 4, fbins = 1 << fused_bits;
    HIP_OR_FAIL(e, e->sort_hist.ensure(sizeof(unsigned) * (size_t)fbins * (size_t)(fwaves + fwgs)));
    unsigned* fhist = e->sort_hist.as<unsigned>();
    unsigned* fhist_wg = fhist + (size_t)fbins * fwaves;
    for (int pass = 0; pass < 2; pass++) {
      const int in = pass & 1, out = in ^ 1;
      const int shift = 27 - (2 - pass) * fused_bits;
      const bool first = pass == 0, last = pass == 1;
      const float4* kp = first ? c.pts.as<float4>() : nullptr;  // first stage: keys computed on the way
      const int* iin = first ? nullptr : e->sort_idx.as<int>();              // (first pass: the identity)
      int* iout = first ? e->sort_idx.as<int>() : c.order.as<int>();         // the final permutation lands in the cloud's own buffer
      const float4* gp = last ? c.pts.as<float4>() : nullptr;
      float4* sp = last ? c.sorted.as<float4>() : nullptr;
      if (fused_bits == 9) {
        radix_hist_fused_kernel<9><<<fwgs, 256, 0, e->stream>>>(keys[in], n, shift, fwaves, fhist, fhist_wg, kp, box, packed_box ? 1 : 0);
        radix_scatter_fused_kernel<9><<<fwgs, 256, 0, e->stream>>>(keys[in], iin, n, shift, fwaves, fhist, fhist_wg, keys[out], iout, gp, sp);
      } else {
        radix_hist_fused_kernel<10><<<fwgs, 256, 0, e->stream>>>(keys[in], n, shift, fwaves, fhist, fhist_wg, kp, box, packed_box ? 1 : 0);
        radix_scatter_fused_kernel<10><<<fwgs, 256, 0, e->stream>>>(keys[in], iin, n, shift, fwaves, fhist, fhist_wg, keys[out], iout, gp, sp);
      }
    }
    const int nsuper = (ntiles + 63) / 64;
    HIP_OR_FAIL(e, c.bbox2.ensure(sizeof(float4) * 2 * (size_t)nsuper));
    const int tile_wgs = (ntiles + 3) / 4;
    tile_super_bbox_kernel<<<tile_wgs + nsuper, 256, 0, e->stream>>>(c.sorted.as<float4>(), n, c.bbox.as<float4>(), c.bbox2.as<float4>(), tile_wgs, packed_box ? c.box.as<unsigned>() : nullptr);
    HIP_OR_FAIL(e, hipGetLastError());
    if (packed_box) { c.has_box = false; c.box_dirty = false; }
    c.has_sorted = true;
    return FVH_OK;
  }
  static const int two_pass_max = [] { const char* v = getenv("FVH_SORT_TWO_PASS_MAX"); return v ? atoi(v) : 262144; }();
  const bool two_pass = n <= two_pass_max;
  const int passes = two_pass ? 2 : RADIX_PASSES, bits = two_pass ? 11 : RADIX_BITS, bins = 1 << bits;
  HIP_OR_FAIL(e, e->sort_hist.ensure(sizeof(unsigned) * (size_t)bins * (nwaves + 1)));
  // the final permutation must land in the cloud's own buffer
  int* idx[2];
  idx[passes & 1] = c.order.as<int>();
  idx[(passes & 1) ^ 1] = e->sort_idx.as<int>();
  const int wblocks = (nwaves + 3) / 4;
  unsigned* hist = e->sort_hist.as<unsigned>();
  unsigned* bin_tot = hist + (size_t)bins * nwaves;
  for (int pass = 0; pass < passes; pass++) {
    const int in = pass & 1, out = in ^ 1;
    const int shift = two_pass ? (pass == 0 ? 5 : 16) : pass * RADIX_BITS;
    const bool first = pass == 0, last = pass == passes - 1;
    const float4* kp = first ? c.pts.as<float4>() : nullptr;  // first stage: keys computed on the way
    if (two_pass) radix_hist_kernel<11><<<wblocks, 256, 0, e->stream>>>(keys[in], n, shift, nwaves, items, hist, kp, box, packed_box ? 1 : 0);
    else radix_hist_kernel<RADIX_BITS><<<wblocks, 256, 0, e->stream>>>(keys[in], n, shift, nwaves, items, hist, kp, box, packed_box ? 1 : 0);
    radix_binscan_kernel<<<bins / 4, 256, 0, e->stream>>>(hist, nwaves, bin_tot, bins);
    radix_scan_kernel<<<1, 1024, 0, e->stream>>>(bin_tot, bins);
    const int* iin = first ? nullptr : idx[in];
    if (two_pass) radix_scatter_kernel<11><<<wblocks, 256, 0, e->stream>>>(keys[in], iin, n, shift, nwaves, items, hist, bin_tot, keys[out], idx[out], last ? c.pts.as<float4>() : nullptr, last ? c.sorted.as<float4>() : nullptr);
    else radix_scatter_kernel<RADIX_BITS><<<wblocks, 256, 0, e->stream>>>(keys[in], iin, n, shift, nwaves, items, hist, bin_tot, keys[out], idx[out], last ? c.pts.as<float4>() : nullptr, last ? c.sorted.as<float4>() : nullptr);
  }
  const int nsuper = (ntiles + 63) / 64;
  HIP_OR_FAIL(e, c.bbox2.ensure(sizeof(float4) * 2 * (size_t)nsuper));
  const int tile_wgs = (ntiles + 3) / 4;
  tile_super_bbox_kernel<<<tile_wgs + nsuper, 256, 0, e->stream>>>(c.sorted.as<float4>(), n, c.bbox.as<float4>(), c.bbox2.as<float4>(), tile_wgs, packed_box ? c.box.as<unsigned>() : nullptr);
  HIP_OR_FAIL(e, hipGetLastError());
  if (packed_box) { c.has_box = false; c.box_dirty = false; }  // consumed by the key kernel, zeroed again by the last kernel of the chain
  c.has_sorted = true;
  return FVH_OK;
}

// ---- multi-GPU: this rank's tile of a cloud = the range [lo, hi) of its Morton order (chunks of equal size, rank order) ----
struct Tile { int lo, hi, chunk; };
inline Tile peer_tile(const Engine* e, int n) {
  if (!e->sharded()) return Tile{0, n, n};
  const int nr = std::max(1, e->shard_ranks());
  const int chunk = ((n + nr - 1) / nr + 63) & ~63;  // whole 64-point tiles of the sorted order
  const int lo = std::min(n, e->shard_rank() * chunk);
  return Tile{lo, std::min(n, lo + chunk), std::max(chunk, 1)};
}
constexpr unsigned long long PEER_WATCHDOG_TICKS = 200'000'000ull;  // 2 s of the 100 MHz clock: a peer may still be uploading / sorting its copy

// After a sharded covariance estimation every rank holds its tile only: pack it into this rank's staging half, publish the
// generation to all peers, and read the other tiles straight out of the peers' staging areas (kernels_peer.hpp).
int peer_allgather_cov(Engine* e, CloudDev& c) {
  Engine::PeerComm& pc = e->peer;
  const Tile t = peer_tile(e, c.n);
  if ((size_t)t.chunk * 32 > pc.stage_half_bytes) return e->fail(FVH_ERR_INVALID_ARGUMENT, "peer exchange: cloud larger than the max_points given to peer_export");
  const unsigned long long gen = ++pc.stage_gen;
  const size_t off = PEER_STAGE_OFFSET + (size_t)(gen & 1ull) * pc.stage_half_bytes;
  const PeerView pv = pc.view(0);
  ProfScope ps(e, "peer_gather");
  if (t.hi > t.lo)
    peer_pack_cov_kernel<<<(t.hi - t.lo + 255) / 256, 256, 0, e->stream>>>(c.cov.as<float4>(), c.order.as<int>(), t.lo, t.hi, reinterpret_cast<float4*>(pc.region + off));
  peer_signal_kernel<<<1, 64, 0, e->stream>>>(pv, gen);
  HIP_OR_FAIL(e, pc.err.ensure(64));
  HIP_OR_FAIL(e, hipMemsetAsync(pc.err.p, 0, 4, e->stream));
  peer_wait_kernel<<<1, 64, 0, e->stream>>>(pv, gen, PEER_WATCHDOG_TICKS, pc.err.as<int>());
  peer_gather_cov_kernel<<<(c.n + 255) / 256, 256, 0, e->stream>>>(pv, off, c.cov.as<float4>(), c.order.as<int>(), c.n, t.chunk, pc.err.as<int>());
  HIP_OR_FAIL(e, hipGetLastError());
  int* h_err = reinterpret_cast<int*>(e->pinned);
  HIP_OR_FAIL(e, hipMemcpyAsync(h_err, pc.err.p, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  if (*h_err) return e->fail(FVH_ERR_COMM, "peer exchange: a rank did not publish its covariance tile in time (every rank must make the same sequence of calls)");
  return FVH_OK;
}

// The same all-gather on the RCCL route (fvh_vgicp_comm_init): every rank packs the covariances of its tile into its slot of a buffer that
// holds the whole cloud in Morton order -- tile r is the range [r chunk, (r + 1) chunk) of it --, ncclAllGather fills the other slots in
// place (32 B per point over xGMI), and one kernel scatters the buffer back to the original point order.
__global__ __launch_bounds__(256) void scatter_sorted_cov_kernel(const float4* __restrict__ stage, float4* __restrict__ cov, const int* __restrict__ order, int n, int lo, int hi) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n || (j >= lo && j < hi)) return;  // (this rank's own tile is already in place)
  const int i = order[j];
  cov[2 * (size_t)i] = stage[2 * (size_t)j];
  cov[2 * (size_t)i + 1] = stage[2 * (size_t)j + 1];
}
int rccl_allgather_cov(Engine* e, CloudDev& c) {
  const Tile t = peer_tile(e, c.n);
  const int nr = std::max(1, e->shard_ranks());
  // With a communicator attached every rank uploads the SAME full cloud (the engine shards internally); a caller still handing each rank
  // its own tile (the contract before round 4) would get mismatched all-gather counts -- a hang or a corrupted collective. Checked
  // per call: max over the ranks of (n, -n) must be (n, -n) everywhere.
  {  // (on EVERY call: gated on this rank's own last size, a rank whose size had not changed skipped the collective the others issued -- a hang)
    int* d = e->misc.as<int>() + 32;
    int* hh = reinterpret_cast<int*>(e->pinned) + 8;
    hh[0] = c.n; hh[1] = -c.n;
    HIP_OR_FAIL(e, hipMemcpyAsync(d, hh, 8, hipMemcpyHostToDevice, e->stream));
    const int rc0 = g_rccl.AllReduce(d, d, 2, /*ncclInt32*/ 2, /*ncclMax*/ 2, e->comm, e->stream);
    if (rc0 != 0) return e->fail(FVH_ERR_COMM, "ncclAllReduce failed with code " + std::to_string(rc0));
    HIP_OR_FAIL(e, hipMemcpyAsync(hh, d, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    if (hh[0] != c.n || hh[1] != -c.n)
      return e->fail(FVH_ERR_COMM, "the ranks hold clouds of different sizes (" + std::to_string(-hh[1]) + " .. " + std::to_string(hh[0]) + " points): with a communicator attached every rank "
                     "uploads the same FULL cloud and the engine shards it internally (include/fast_vgicp_hip.h: fvh_vgicp_comm_init)");
  }
  HIP_OR_FAIL(e, e->gather_stage.ensure(sizeof(float4) * 2 * (size_t)t.chunk * nr));
  float4* stage = e->gather_stage.as<float4>();
  ProfScope ps(e, "peer_gather");
  if (t.hi > t.lo)
    peer_pack_cov_kernel<<<(t.hi - t.lo + 255) / 256, 256, 0, e->stream>>>(c.cov.as<float4>(), c.order.as<int>(), t.lo, t.hi, stage + 2 * (size_t)t.lo);
  HIP_OR_FAIL(e, hipGetLastError());
  const int rc = g_rccl.AllGather(stage + 2 * (size_t)e->shard_rank() * t.chunk, stage, (size_t)t.chunk * 8, /*ncclFloat*/ 7, e->comm, e->stream);
  if (rc != 0) return e->fail(FVH_ERR_COMM, "ncclAllGather failed with code " + std::to_string(rc));
  scatter_sorted_cov_kernel<<<(c.n + 255) / 256, 256, 0, e->stream>>>(stage, c.cov.as<float4>(), c.order.as<int>(), c.n, t.lo, t.hi);
  HIP_OR_FAIL(e, hipGetLastError());
  return FVH_OK;
}
inline int allgather_cov(Engine* e, CloudDev& c) { return e->peer.attached() ? peer_allgather_cov(e, c) : rccl_allgather_cov(e, c); }

int find_neighbors(Engine* e, CloudDev& c, int k) {
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "find_neighbors: cloud not set");
  if (k <= 0 || k > 64) return e->fail(FVH_ERR_INVALID_ARGUMENT, "find_neighbors: k must be in [1, 64]");
  if (c.n < k) return e->fail(FVH_ERR_INVALID_ARGUMENT, "find_neighbors: fewer points than k");
  HIP_OR_FAIL(e, c.nbr.ensure(sizeof(int) * (size_t)c.n * k));
#ifdef FVH_TEST_KERNELS  // test build only: FVH_KNN_MODE=0 selects the superseded full LDS-tiled sweep as a cross-check of the culled search
  static const int knn_mode = [] { const char* v = getenv("FVH_KNN_MODE"); return v ? atoi(v) : 1; }();
  if (knn_mode == 0 && !e->sharded()) {
    const int waves = (c.n + KNN_Q - 1) / KNN_Q;
    ProfScope ps(e, "knn");
    knn_bruteforce_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(c.pts.as<float4>(), c.n, k, c.nbr.as<int>());
  } else
#endif
  {
    int rc = ensure_sorted(e, c);
    if (rc) return rc;
    const Tile t = peer_tile(e, c.n);  // multi-GPU: the queries of this rank's tile only (the whole sorted cloud is the candidate set: an exact, implicit halo)
    ProfScope ps(e, "knn");
    // small clouds: the kernel lasts as long as its slowest queries, and those are the ones the nearest-first walk shortens;
    // the throughput-bound sizes hide them behind the other queries and keep the cheaper index-order walk (kernels_cov.hpp)
    static const int nf_max = [] { const char* v = getenv("FVH_KNN_NEAREST_FIRST_MAX_POINTS"); return v ? atoi(v) : 65536; }();
    // one query = one wave = one WORKGROUP: a 4-wave workgroup holds its four slots until its slowest query is done (queries take
    // 8 us on average, 12.6 at the 90th percentile), single-wave workgroups hand each slot back at once: 39.7 -> 37 us at 17k points,
    // 161 -> 154 us at 100k (FVH_KNN_BLOCK=256 / 128: the old shapes, for A/B runs)
    static const int knn_block = [] { const char* v = getenv("FVH_KNN_BLOCK"); const int b = v ? atoi(v) : 64; return (b == 256 || b == 128) ? b : 64; }();
    const int per_block = knn_block / 64;
    if (t.hi > t.lo) {
      if (c.n <= nf_max)
        knn_tiled1_kernel<true><<<(t.hi - t.lo + per_block - 1) / per_block, knn_block, 0, e->stream>>>(c.sorted.as<float4>(), c.bbox.as<float4>(), c.bbox2.as<float4>(), c.n, k, c.nbr.as<int>(), t.lo, t.hi);
      else
        knn_tiled1_kernel<false><<<(t.hi - t.lo + per_block - 1) / per_block, knn_block, 0, e->stream>>>(c.sorted.as<float4>(), c.bbox.as<float4>(), c.bbox2.as<float4>(), c.n, k, c.nbr.as<int>(), t.lo, t.hi);
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  c.k = k;
  c.has_nbr = true;
  c.nbr_tile_only = e->sharded();
  return FVH_OK;
}

int calc_cov_knn(Engine* e, CloudDev& c, int method) {
  if (!c.has_pts || !c.has_nbr) return e->fail(FVH_ERR_BAD_STATE, "calculate_covariances: cloud or neighbours not set");
  if (method < 0 || method > 4) return e->fail(FVH_ERR_INVALID_ARGUMENT, "unknown regularization method");
  if (c.k > COV_LANES * COV_MAX_PER_LANE) return e->fail(FVH_ERR_UNSUPPORTED, "calculate_covariances: more than 64 neighbours per point");
  HIP_OR_FAIL(e, c.cov.ensure(sizeof(float4) * 2 * (size_t)std::max(c.n, 1)));
  const bool sharded = e->sharded();
  if (sharded) { int rc = ensure_sorted(e, c); if (rc) return rc; }
  if (c.n) {
    const Tile t = peer_tile(e, c.n);
    const int m = sharded ? (t.hi - t.lo) : c.n;                      // points this rank computes
    const int* subset = sharded ? c.order.as<int>() + t.lo : nullptr;  // ... its tile of the Morton order
    // clouds the LM loop walks in Morton order: computed in that order too, and left a second time at the points' places along the curve
    float4* cov_sorted = nullptr;
    c.has_cov_sorted = false;
    if (!sharded && e->precision != FVH_COMPUTE_CUDA_COMPAT && coherent_order(c)) {
      HIP_OR_FAIL(e, c.cov_sorted.ensure(sizeof(float4) * 2 * (size_t)c.n));
      subset = c.order.as<int>();
      cov_sorted = c.cov_sorted.as<float4>();
    }
    ProfScope ps(e, "cov");
    const int blocks = (int)(((long long)m * COV_LANES + 255) / 256);
    if (m > 0 && e->precision == FVH_COMPUTE_CUDA_COMPAT) {
      // FastVGICPCuda's own arithmetic: uncentred float sums in list order + Eigen's closed-form float eigen solver (kernels_cov.hpp)
      cov_from_neighbors_cuda_compat_kernel<<<(m + 255) / 256, 256, 0, e->stream>>>(c.pts.as<float4>(), m, c.k, c.nbr.as<int>(), method, c.cov.as<float4>(), subset);
    } else if (m > 0) {
      if (c.k <= 20) cov_from_neighbors_kernel<5><<<blocks, 256, 0, e->stream>>>(c.pts.as<float4>(), m, c.k, c.nbr.as<int>(), method, c.cov.as<float4>(), subset, cov_sorted);
      else if (c.k <= 32) cov_from_neighbors_kernel<8><<<blocks, 256, 0, e->stream>>>(c.pts.as<float4>(), m, c.k, c.nbr.as<int>(), method, c.cov.as<float4>(), subset, cov_sorted);
      else cov_from_neighbors_regather_kernel<<<blocks, 256, 0, e->stream>>>(c.pts.as<float4>(), m, c.k, c.nbr.as<int>(), method, c.cov.as<float4>(), subset, cov_sorted);
      c.has_cov_sorted = cov_sorted != nullptr;
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  if (sharded && c.n) { int rc = allgather_cov(e, c); if (rc) return rc; }
  c.has_cov = true;
  return FVH_OK;
}

int calc_cov_rbf(Engine* e, CloudDev& c, double kernel_width, double max_dist, int method) {
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "calculate_covariances_rbf: cloud not set");
  if (method < 0 || method > 4) return e->fail(FVH_ERR_INVALID_ARGUMENT, "unknown regularization method");
  HIP_OR_FAIL(e, c.cov.ensure(sizeof(float4) * 2 * (size_t)std::max(c.n, 1)));
  const bool sharded = e->sharded();
  c.has_cov_sorted = false;
  if (c.n) {
    const float md = (float)max_dist;
#ifdef FVH_TEST_KERNELS  // test build only: FVH_RBF_MODE=0 full sweep, 2: eight queries per wave (both superseded by the one-query-per-wave sweep)
    static const int rbf_mode = [] { const char* v = getenv("FVH_RBF_MODE"); return v ? atoi(v) : 1; }();
    const int waves = (c.n + RBF_Q - 1) / RBF_Q;
    if (rbf_mode == 0 && !sharded) {
      ProfScope ps(e, "rbf");
      cov_rbf_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(c.pts.as<float4>(), c.n, (float)kernel_width, md * md, method, c.cov.as<float4>());
    } else
#endif
    {
      if (e->precision == FVH_COMPUTE_CUDA_COMPAT) {
        // FastVGICPCuda's own arithmetic: float sums per block of 512 candidates in index order, blocks added in order (kernels_compat.hpp)
        const int* subset = nullptr;
        int m = c.n;
        if (sharded) {
          int rc = ensure_sorted(e, c);
          if (rc) return rc;
          const Tile t = peer_tile(e, c.n);
          subset = c.order.as<int>() + t.lo; m = t.hi - t.lo;
        }
        ProfScope ps(e, "rbf");
        if (m > 0) cov_rbf_cuda_compat_kernel<<<(m + 63) / 64, 256, 0, e->stream>>>(c.pts.as<float4>(), c.n, (float)kernel_width, md, method, c.cov.as<float4>(), subset, m);
        HIP_OR_FAIL(e, hipGetLastError());
        if (sharded) { int rc = allgather_cov(e, c); if (rc) return rc; }
        c.has_cov = true;
        return FVH_OK;
      }
      int rc = ensure_sorted(e, c);
      if (rc) return rc;
      const Tile t = peer_tile(e, c.n);
      ProfScope ps(e, "rbf");
#ifdef FVH_TEST_KERNELS
      if (rbf_mode == 2 && !sharded) cov_rbf_tiled_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(c.sorted.as<float4>(), c.bbox.as<float4>(), c.n, (float)kernel_width, md * md, method, c.cov.as<float4>());
      else
#endif
      if (t.hi > t.lo) {
        // sweep (one query per wave) -> ten totals per query; regularisation with one thread per query
        HIP_OR_FAIL(e, e->rbf_sums.ensure(sizeof(double) * 10 * (size_t)c.n));
        // (single-wave workgroups, which shortened the k-NN kernel, change nothing here: 152 us either way -- this sweep keeps the VALU pipes 95 % busy)
        cov_rbf1_kernel<<<(t.hi - t.lo + 3) / 4, 256, 0, e->stream>>>(c.sorted.as<float4>(), c.bbox.as<float4>(), c.bbox2.as<float4>(), c.n, (float)kernel_width, md * md, method, c.cov.as<float4>(), t.lo, t.hi,
                                                                      e->rbf_sums.as<double>());
        float4* cov_sorted = nullptr;
        if (!sharded && coherent_order(c)) { HIP_OR_FAIL(e, c.cov_sorted.ensure(sizeof(float4) * 2 * (size_t)c.n)); cov_sorted = c.cov_sorted.as<float4>(); }
        cov_rbf_finish_kernel<<<(t.hi - t.lo + 255) / 256, 256, 0, e->stream>>>(e->rbf_sums.as<double>(), c.sorted.as<float4>(), c.n, method, c.cov.as<float4>(), t.lo, t.hi, cov_sorted);
        c.has_cov_sorted = cov_sorted != nullptr;
      }
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  if (sharded && c.n) { int rc = allgather_cov(e, c); if (rc) return rc; }
  c.has_cov = true;
  return FVH_OK;
}

int set_cov_host(Engine* e, CloudDev& c, const double* covs9) {
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "set_covariances: cloud not set");
  if (!covs9) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_covariances: null");
  std::vector<float4> h(2 * (size_t)c.n);
  for (int i = 0; i < c.n; i++) {
    const double* m = covs9 + 9 * (size_t)i;
    h[2 * i] = make_float4((float)m[0], (float)m[1], (float)m[2], (float)m[4]);
    h[2 * i + 1] = make_float4((float)m[5], (float)m[8], 0.f, 0.f);
  }
  HIP_OR_FAIL(e, c.cov.ensure(sizeof(float4) * 2 * (size_t)std::max(c.n, 1)));
  HIP_OR_FAIL(e, hipMemcpyAsync(c.cov.p, h.data(), sizeof(float4) * h.size(), hipMemcpyHostToDevice, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  c.has_cov = true;
  c.has_cov_sorted = false;
  return FVH_OK;
}

int get_cov_host(Engine* e, CloudDev& c, float* covs9) {
  if (!c.has_cov) return e->fail(FVH_ERR_BAD_STATE, "get_covariances: covariances not computed");
  std::vector<float4> h(2 * (size_t)c.n);
  HIP_OR_FAIL(e, hipMemcpyAsync(h.data(), c.cov.p, sizeof(float4) * h.size(), hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  for (int i = 0; i < c.n; i++) {
    const float4 a = h[2 * i], b = h[2 * i + 1];
    float* m = covs9 + 9 * (size_t)i;
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.y; m[4] = a.w; m[5] = b.x; m[6] = a.z; m[7] = b.x; m[8] = b.y;
  }
  return FVH_OK;
}

int get_nbr_host(Engine* e, CloudDev& c, int* k, int* out) {
  if (!c.has_nbr) return e->fail(FVH_ERR_BAD_STATE, "get_neighbors: neighbours not set");
  if (k) *k = c.k;
  if (out) {
    HIP_OR_FAIL(e, hipMemcpyAsync(out, c.nbr.p, sizeof(int) * (size_t)c.n * c.k, hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  }
  return FVH_OK;
}

int radix_sort_pairs(Engine* e, unsigned* keys[2], int* idx[2], int n, int bits, int* result, hipStream_t on = nullptr, DevBuf* hist_buf = nullptr);

// GaussianVoxelMap::create_voxelmap (gaussian_voxelmap.cu:208-257) -- two kernels, no retry loop
template <int MODE>
int build_voxelmap(Engine* e, const CloudDev& c, VoxelMapDev& vm, double res, bool want_compact, bool force_safe = false, hipStream_t on_side = nullptr,
                   bool shard = false /* only the voxels of vm.region (already computed on this stream) */,
                   bool detached = false /* a map that is not the live one yet (prepared-source slot): the caller orders the main stream after `on_side` itself, correspondences stay valid */) {
  hipStream_t const st = on_side ? on_side : e->stream;
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "create_voxelmap: cloud not set");
  if (MODE != 1 && !c.has_cov) return e->fail(FVH_ERR_BAD_STATE, "create_voxelmap: covariances not computed");
  if (!(res > 0)) return e->fail(FVH_ERR_INVALID_ARGUMENT, "create_voxelmap: resolution must be > 0");
  unsigned safe = 1024;
  while (safe < 2u * (unsigned)std::max(c.n, 1)) safe <<= 1;
  unsigned cap = safe;  // can never overflow
  if (!force_safe && vm.nv_hint >= 0) {  // keep the table L2-resident: 4x the last voxel count
    cap = 1024;
    while (cap < 4u * (unsigned)vm.nv_hint) cap <<= 1;
    cap = std::min(cap, safe);
  }
  vm.res = res;
  vm.capacity = cap;
  vm.invalidate();
  {  // a reallocation hands back dirty memory
    void* before[4] = {vm.keys[0].p, vm.keys[1].p, vm.acc.p, vm.counters.p};
    HIP_OR_FAIL(e, vm.table.ensure((size_t)cap * 64));
    HIP_OR_FAIL(e, vm.keys[0].ensure((size_t)cap * 8));
    HIP_OR_FAIL(e, vm.keys[1].ensure((size_t)cap * 8));
    HIP_OR_FAIL(e, vm.acc.ensure((size_t)cap * VM_ACC_BUCKET * sizeof(double)));
    HIP_OR_FAIL(e, vm.counters.ensure(2 * 16 * sizeof(int)));
    if (before[0] != vm.keys[0].p || before[1] != vm.keys[1].p || before[2] != vm.acc.p || before[3] != vm.counters.p) vm.clean_cap = 0;
  }
  HIP_OR_FAIL(e, vm.occupied.ensure(sizeof(int) * (size_t)std::max(c.n, 1)));
  // FVH_COMPUTE_CUDA_COMPAT: voxel coordinates in float and, behind the build, the float voxel sums of the CUDA classes (kernels_compat.hpp);
  // the multiplicative voxels have no device counterpart in the reference and keep the fp64 path
  const bool compat = e->precision == FVH_COMPUTE_CUDA_COMPAT && MODE != 2;
  if (want_compact) {
    HIP_OR_FAIL(e, vm.compact_pts.ensure(sizeof(float4) * (size_t)std::max(c.n, 1)));
    HIP_OR_FAIL(e, vm.compact_cov.ensure(sizeof(float4) * 2 * (size_t)std::max(c.n, 1)));
  }
  {
    ProfScope ps(e, "voxelmap", st);
    const int fill = vm.cur ^ 1;
    unsigned long long* keys = vm.keys[fill].as<unsigned long long>();
    int* counters = vm.counters.as<int>() + 16 * fill;
    if (vm.clean_cap != cap) vm_clear_kernel<<<(cap * 10 + 255) / 256, 256, 0, st>>>(keys, vm.acc.as<double>(), cap, counters);
    vm.clean_cap = 0;  // keys[fill] is in use from here on; the finalize pass below makes the OTHER pair clean
    if (c.n) {
      vm_accumulate_kernel<MODE><<<(c.n + 255) / 256, 256, 0, st>>>(c.pts.as<float4>(), c.cov.as<float4>(), c.n, res, keys, cap - 1, vm.acc.as<double>(), counters + 1,
                                                                           coherent_order(c), shard ? vm.region.as<VmRegion>() : nullptr, compat ? 1 : 0);
      vm_finalize_kernel<MODE><<<(cap + VM_FIN_THREADS - 1) / VM_FIN_THREADS, VM_FIN_THREADS, 0, st>>>(keys, vm.table.as<uint4>(), cap, vm.acc.as<double>(), counters, vm.occupied.as<int>(),
                                                                        want_compact ? vm.compact_pts.as<float4>() : nullptr, want_compact ? vm.compact_cov.as<float4>() : nullptr,
                                                                        vm.keys[vm.cur].as<unsigned long long>(), vm.counters.as<int>() + 16 * vm.cur);
      vm.clean_cap = cap;
      if (compat) {
        const int n = c.n;
        HIP_OR_FAIL(e, vm.compat_keys.ensure(sizeof(unsigned) * 2 * (size_t)n));
        HIP_OR_FAIL(e, vm.compat_idx.ensure(sizeof(int) * 2 * (size_t)n));
        HIP_OR_FAIL(e, vm.compat_seg.ensure(sizeof(int) * ((size_t)cap + 2)));
        unsigned* ck[2] = {vm.compat_keys.as<unsigned>(), vm.compat_keys.as<unsigned>() + n};
        int* ci[2] = {vm.compat_idx.as<int>(), vm.compat_idx.as<int>() + n};
        vmc_point_bucket_kernel<<<(n + 255) / 256, 256, 0, st>>>(c.pts.as<float4>(), n, (float)res, keys, cap - 1, ck[0], ci[0]);
        int bits = 1;
        while ((1u << bits) <= cap) bits++;  // buckets 0 .. cap - 1 and `cap` itself ("no voxel")
        int sorted = 0;
        int rc = radix_sort_pairs(e, ck, ci, n, bits, &sorted, st, &vm.compat_hist);
        if (rc) return rc;
        vmc_segment_heads_kernel<<<(n + 255) / 256, 256, 0, st>>>(ck[sorted], n, vm.compat_seg.as<int>());
        // (an upper bound of the voxel count sizes the grid: the exact one is on the device)
        const int max_voxels = (int)std::min<long long>(n, cap);
        vmc_finalize_kernel<MODE><<<(max_voxels + 63) / 64, 64, 0, st>>>(c.pts.as<float4>(), MODE == 0 ? c.cov.as<float4>() : nullptr, ci[sorted], vm.compat_seg.as<int>(), vm.occupied.as<int>(),
                                                                     counters, vm.table.as<uint4>(), want_compact ? vm.compact_pts.as<float4>() : nullptr,
                                                                     want_compact ? vm.compact_cov.as<float4>() : nullptr);
      }
      // large map: occupancy bitmap over the bounding box of its voxels (kernels_voxelmap.hpp) -- the LM kernel answers its misses
      // from these cache-resident bits instead of a 64-byte HBM sector per probe. Four small launches after the finalize pass; maps
      // of this size are built once per localisation run, not once per registration.
      static const int bitmap_min = [] { const char* v = getenv("FVH_BITMAP_MIN_POINTS"); return v ? atoi(v) : 300000; }();
      static const size_t bitmap_bytes = [] { const char* v = getenv("FVH_BITMAP_MAX_BYTES"); return v ? std::min((size_t)atoll(v), (size_t)16 << 30) : (size_t)(32u << 20); }();  // (the LM kernel indexes the words with 32 bits)
      if (c.n >= bitmap_min && bitmap_bytes >= 8 && !shard) {  // (a shard is a fraction of the map: its keys stay cache-resident)
        HIP_OR_FAIL(e, vm.bitmap.ensure(bitmap_bytes));
        HIP_OR_FAIL(e, vm.grid.ensure(sizeof(VmGrid)));
        VmGrid* g = vm.grid.as<VmGrid>();
        vm_grid_init_kernel<<<1, 64, 0, st>>>(g);
        vm_grid_bounds_kernel<<<64, 256, 0, st>>>(keys, vm.occupied.as<int>(), counters, g);
        vm_grid_setup_kernel<<<1, 64, 0, st>>>(g, (unsigned long long)(bitmap_bytes / 8));
        vm_grid_clear_kernel<<<512, 256, 0, st>>>(vm.bitmap.as<unsigned long long>(), g);
        vm_grid_set_kernel<<<256, 256, 0, st>>>(keys, vm.occupied.as<int>(), counters, g, vm.bitmap.as<unsigned long long>());
        vm.has_bitmap = true;
      }
    }
    vm.cur = fill;
  }
  HIP_OR_FAIL(e, hipGetLastError());
  if (on_side && !detached) {
    HIP_OR_FAIL(e, hipEventRecord(e->side_done, on_side));
    e->side_pending = true;
  }
  vm.valid = true;
  vm.is_shard = shard;
  if (!detached) e->has_corr = false;
  return FVH_OK;
}

using Rebuild = std::function<int()>;

// `rebuild_safe`: what to do when the hint-sized table of this map overflowed (counter [1]): rebuild at the safe size and
// read again, as align / compute_error do -- the getters must never hand out a silently truncated map.
int fetch_voxelmap_host(Engine* e, VoxelMapDev& vm, const Rebuild* rebuild_safe = nullptr) {
  if (!vm.valid) return e->fail(FVH_ERR_BAD_STATE, "voxel map not built");
  if (vm.host_valid) return FVH_OK;
  int counters[3] = {0, 0, 0};
  for (int attempt = 0;; attempt++) {
    HIP_OR_FAIL(e, hipMemcpyAsync(counters, vm.counters_cur(), sizeof(counters), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    if (counters[1] == 0) break;
    if (attempt == 1 || !rebuild_safe) return e->fail(FVH_ERR_BAD_STATE, "voxel map table overflowed (" + std::to_string(counters[1]) + " entries dropped)");
    int rc = (*rebuild_safe)();
    if (rc) return rc;
  }
  vm.nv_hint = counters[0];
  vm.num_skipped = counters[2];
  vm.h_occupied.resize(counters[0]);
  vm.h_table.resize((size_t)vm.capacity * 4);
  if (counters[0]) HIP_OR_FAIL(e, hipMemcpyAsync(vm.h_occupied.data(), vm.occupied.p, sizeof(int) * counters[0], hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipMemcpyAsync(vm.h_table.data(), vm.table.p, (size_t)vm.capacity * 64, hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  vm.bucket_to_index.clear();
  for (int i = 0; i < counters[0]; i++) vm.bucket_to_index[vm.h_occupied[i]] = i;
  vm.host_valid = true;
  return FVH_OK;
}

int get_voxels_host(Engine* e, VoxelMapDev& vm, int* coords3, int* num_points, float* means3, float* covs9, const Rebuild* rebuild_safe = nullptr) {
  int rc = fetch_voxelmap_host(e, vm, rebuild_safe);
  if (rc) return rc;
  for (size_t i = 0; i < vm.h_occupied.size(); i++) {
    const uint4* q = &vm.h_table[(size_t)vm.h_occupied[i] * 4];
    if (coords3) {
      unsigned long long key = (unsigned long long)q[0].x | ((unsigned long long)q[0].y << 32);
      unpack_key(key, coords3[3 * i], coords3[3 * i + 1], coords3[3 * i + 2]);
    }
    if (num_points) num_points[i] = (int)q[0].z;
    const float* f1 = reinterpret_cast<const float*>(&q[1]);
    const float* f2 = reinterpret_cast<const float*>(&q[2]);
    const float* f3 = reinterpret_cast<const float*>(&q[3]);
    if (means3) { means3[3 * i] = f1[0]; means3[3 * i + 1] = f1[1]; means3[3 * i + 2] = f1[2]; }
    if (covs9) {
      float* m = covs9 + 9 * i;
      m[0] = f2[0]; m[1] = f2[1]; m[2] = f2[2]; m[3] = f2[1]; m[4] = f2[3]; m[5] = f3[0]; m[6] = f2[2]; m[7] = f3[0]; m[8] = f3[1];
    }
  }
  return FVH_OK;
}

struct CostSource {
  const float4* pts; const float4* cov; const int* d_n; int n_upper;
  const int* counters2;  // source voxel map counters (D2D) or null
  const int* order;      // Morton permutation of the source (large clouds) or null
  const float4* sorted = nullptr;  // with `order`: the cloud's Morton-ordered copy (.w = original index) -- element order[j] is sorted[j]
  const float4* cov_sorted = nullptr;  // with `sorted`, optional: the covariances in the same order
  int n_off_override = 0;  // > 0: correspondences per source element regardless of the handle's offset list (GICP: 1)
  bool shardable = false;  // the source elements are the points of a cloud with a Morton order: with peers attached each rank walks its tile
  bool external_find = false;  // FastGICP device LM: nn1_corr_kernel fills the correspondence buffers between the cost launches
  int n_shape = 0;             // > 0: expected number of source elements when the exact one lives on the device (NDT D2D: source voxels, from the
                               // last build of that map): shapes the grid and the offsets per item; the kernel is grid-stride, any value is correct
  VoxelMapDev* source_map = nullptr;  // NDT D2D: where align() leaves the source voxel count it saw
  bool device_tile = false;           // NDT D2D with a tile set: the kernel cuts this rank's chunk of the (canonically ordered) element list from the device-side count
  // NDT D2D: the elements ARE the source map's compact voxel list. A rebuild of that map (table overflow -> safe size) flips its counter set and
  // refills the list: whoever retries with a CostSource made before the rebuild re-reads the map's addresses first (round 6: the stale counter
  // set -- zeroed by the rebuild's finalize pass -- made the retry evaluate an EMPTY source)
  void refresh() {
    if (!source_map) return;
    const VoxelMapDev& m = *source_map;
    pts = m.compact_pts.as<float4>(); cov = m.compact_cov.as<float4>();
    d_n = counters2 = m.counters_cur();
    if (device_tile) order = m.has_canon ? m.canon.as<int>() : nullptr;
  }
};

// Persistent LM kernel (kernels_cost.hpp, PERSIST): co-resident workgroup capacity of the device for this instantiation.
constexpr unsigned long long PERSIST_WATCHDOG_TICKS = 5'000'000ull;  // 50 ms of the 100 MHz wall clock
constexpr long long PERSIST_MAX_ITEMS = 4'000'000;                   // beyond this a trip is no longer latency-bound: multi-launch path
template <int MODE>
int persistent_capacity(Engine* e) {
  static std::mutex mu;
  static int cap[16][2];  // [device][precision]; 0 = not queried yet
  std::lock_guard<std::mutex> lk(mu);
  const int pi = e->float_cost() ? 1 : 0, dev = e->device & 15;
  if (cap[dev][pi] <= 0) {
    int per_cu = 0, cus = 0;
    hipError_t r = pi ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cost_kernel<float, MODE, true>, 256, 0)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cost_kernel<double, MODE, true>, 256, 0);
    if (r != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) return 0;
    cap[dev][pi] = per_cu * cus;
  }
  return cap[dev][pi];
}

// A work item is (source element, group of offsets). Small clouds: enough items to cover the chip (target_items). Large
// clouds: still at most COST_CH..group_max offsets per item -- one thread walking all 27 offsets of its point left 24 %
// of the resident threads without work at 100k points and made the launch 18 % slower than 4 offsets per item
// (measured at 100k x DIRECT27: group 27: 423 us, 14: 426, 9: 361, 7: 386, 6: 368, 5: 423, 4: 358, 3: 355, 2: 459, 1: 615).
struct CostShape { int group, groups_per_src; long long n_walk; int blocks; int split; };
// `mode`, `device_lm`: NDT launches of the device-resident optimiser loop whose items hold one offset and whose grid stays within one
// workgroup per CU take the wave-role layout (kernels_cost.hpp: `split` -- 128 items per workgroup, waves 0-1 the trial error of the
// stored ids, waves 2-3 the new linearisation). Both routes of an align call this with the same arguments: the same layout.
inline CostShape cost_shape(const Engine* e, const CostSource& src, int mode = MODE_VGICP, bool device_lm = false) {
  static const long long target_items = [] { const char* v = getenv("FVH_COST_TARGET_ITEMS"); return v ? atoll(v) : 256LL * 256 * 2; }();
  static const int max_blocks = [] { const char* v = getenv("FVH_COST_MAX_BLOCKS"); int b = v ? atoi(v) : MAX_COST_BLOCKS; return b < 1 ? 1 : (b > MAX_COST_BLOCKS ? MAX_COST_BLOCKS : b); }();
  static const int group_max = [] { const char* v = getenv("FVH_COST_GROUP_MAX"); return v ? std::min(std::max(1, atoi(v)), COST_CH) : COST_CH; }();  // the kernel keeps one item's lookups in flight together: at most COST_CH
  const int n_off = src.n_off_override > 0 ? src.n_off_override : e->n_off;
  CostShape s;
  const int n_expected = src.n_shape > 0 ? std::min(src.n_shape, src.n_upper) : src.n_upper;
  const int groups = (int)std::min<long long>(n_off, std::max<long long>((n_off + group_max - 1) / group_max, target_items / std::max(n_expected, 1)));
  s.group = (n_off + groups - 1) / groups;
  s.groups_per_src = (n_off + s.group - 1) / s.group;
  s.n_walk = n_expected;
  if (e->sharded() && src.shardable) { const Tile t = peer_tile(e, src.n_upper); s.n_walk = std::max(t.hi - t.lo, 0); }  // multi-GPU: this rank's tile
  if (src.device_tile) s.n_walk = (n_expected + e->shard_ranks() - 1) / std::max(1, e->shard_ranks());
  s.blocks = (int)std::max<long long>(1, std::min<long long>(max_blocks, (s.n_walk * s.groups_per_src + 255) / 256));
  s.split = 0;
  static const int split_on = [] { const char* v = getenv("FVH_COST_SPLIT"); return v ? atoi(v) : 1; }();
  if (split_on && device_lm && mode != MODE_VGICP && s.group == 1) {
    static int cus[16] = {0};
    const int dev = e->device & 15;
    if (cus[dev] <= 0 && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) cus[dev] = 0;
    const long long wgs = (s.n_walk * s.groups_per_src + 127) / 128;
    if (cus[dev] > 0 && wgs <= cus[dev]) { s.split = 1; s.blocks = (int)std::max<long long>(1, wgs); }
  }
  return s;
}

template <int MODE>
int launch_cost(Engine* e, const CostSource& src, const VoxelMapDev& vm, int host_phase, const PoseD* lin, const PoseD* ev, const fvh_lm_params* init = nullptr,
                bool persistent = false, unsigned long long peer_xbase = 0 /* multi-GPU: exchange counter of this launch's first sums exchange */,
                const GridPlan* plan = nullptr /* align(): the layout both routes take (workgroups granted, groups, XCD confinement) */) {
  CostParams P;
  std::memset(&P, 0, sizeof(P));
  P.src_pts = src.pts; P.src_cov = src.cov; P.d_n_src = src.d_n; P.n_src = src.n_upper; P.order = src.order; P.src_sorted = src.order ? src.sorted : nullptr; P.src_cov_sorted = (src.order && src.sorted) ? src.cov_sorted : nullptr;
  P.table = vm.table.as<uint4>(); P.keys = vm.keys_cur(); P.mask = vm.capacity - 1; P.res = vm.res; P.inv_res = 1.0 / vm.res;
  P.bitmap = vm.has_bitmap ? vm.bitmap.as<unsigned long long>() : nullptr;
  P.grid = vm.has_bitmap ? vm.grid.as<VmGrid>() : nullptr;
  P.region = vm.is_shard ? vm.region.as<VmRegion>() : nullptr;
  const int n_off = src.n_off_override > 0 ? src.n_off_override : e->n_off;
  P.offsets = e->offsets_dev.as<int>(); P.offsets_packed = e->offsets_dev.as<int>() + 3 * (size_t)e->n_off; P.n_off = n_off;
  const CostShape shape = cost_shape(e, src, MODE, host_phase < 0);
  P.group = shape.group;
  P.split = shape.split;
  P.groups_per_src = shape.groups_per_src;
  {  // w / d == mulhi(w, ceil(2^32 / d)) for all w with w * d < 2^32 (d = 1: no shift-free magic, plain division is free there)
    const unsigned long long d = (unsigned long long)shape.groups_per_src, items = (unsigned long long)std::max(src.n_upper, 1) * d;
    P.gps_magic = (d > 1 && items * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
  }
  P.corr = e->corr.as<int>();
  P.corr_stride = (size_t)std::max(src.n_upper, 1) * n_off;
  P.host_corr_sel = e->corr_sel;
  P.st = e->state.as<LmState>(); P.partials = e->partials.as<double>(); P.ticket = e->ticket.as<unsigned>();
  P.vm_counters = vm.counters_cur();
  P.vm_counters2 = src.counters2;
  P.host_phase = host_phase;
  P.external_find = src.external_find ? 1 : 0;
  {
    const bool finds = host_phase < 0 || host_phase == PH_FIND_ONLY;  // (the device-resident loop finds its own lists; PH_EVAL_* read a stored one)
    if (finds) e->corr_by_position = src.order != nullptr && !src.external_find && !src.device_tile;
    if (e->corr_by_position && !src.order) return e->fail(FVH_ERR_BAD_STATE, "compute_error: the stored correspondences were found in the cloud's spatial order, which is gone; call update_correspondences again");
    P.corr_by_position = e->corr_by_position ? 1 : 0;
  }
  P.lm_trace = (e->lm_trace_on && host_phase < 0) ? e->lm_trace.as<double>() : nullptr;
  P.defer_lm = (e->comm != nullptr) ? 1 : 0;
  if (lin) P.lin = *lin;
  if (ev) P.ev = *ev;
  if (init) {
    P.init = 1;
    P.max_iterations = init->max_iterations; P.lm_max_iterations = init->lm_max_iterations;
    P.rotation_epsilon = init->rotation_epsilon; P.transformation_epsilon = init->transformation_epsilon; P.lm_init_lambda_factor = init->lm_init_lambda_factor;
    P.optimizer = init->optimizer != 0 ? 1 : 0;
  }
  long long n_walk = src.n_upper;
  P.item_lo = 0; P.item_hi = 0;
  P.peer.n = 1; P.peer.rank = 0; P.peer.xbase = 0;
  for (int i = 0; i < FVH_MAX_PEERS; i++) P.peer.region[i] = nullptr;
  P.tile_rank = 0; P.tile_n = 1;
  if (MODE == MODE_NDT_D2D && src.device_tile) { P.tile_rank = e->shard_rank(); P.tile_n = e->shard_ranks(); }
  if ((MODE == MODE_VGICP || MODE == MODE_NDT_P2D) && e->sharded() && src.shardable) {
    // multi-GPU: this rank's spatial tile of the source (a range of its Morton order) and -- peer route -- the mailboxes of all ranks
    // (RCCL route: the sums meet between the launches, allreduce_sums)
    const Tile t = peer_tile(e, src.n_upper);
    P.item_lo = t.lo; P.item_hi = std::max(t.hi, 1);  // (item_hi == 0 means "everything")
    if (t.hi <= t.lo) { P.item_lo = 0; P.item_hi = 1; n_walk = 0; P.n_src = 0; } else n_walk = t.hi - t.lo;
    if (MODE == MODE_VGICP && e->peer.attached()) P.peer = e->peer.view(peer_xbase);
    static const unsigned long long wd = [] { const char* v = getenv("FVH_PEER_WATCHDOG_TICKS"); return v ? strtoull(v, nullptr, 10) : PEER_WATCHDOG_TICKS; }();
    P.peer_watchdog_ticks = wd;
  }
  int blocks = shape.blocks;
  (void)n_walk;
  {
    // The persistent kernel needs every workgroup resident at once: its grid is clamped to what the device can hold (the
    // kernel is grid-stride). The per-transition launches take the SAME grid, so that both routes partition the items --
    // and therefore order the sums -- identically (bit-identical results whichever route an align takes).
    int cap = persistent_capacity<MODE>(e);
    if (cap <= 0) return e->fail(FVH_ERR_HIP, "cost kernel: occupancy query failed");
    if (e->peer.attached()) cap = std::max(1, cap / std::max(1, e->peer.ranks_on_device));  // ranks sharing one GPU share its co-resident slots
    blocks = std::min(blocks, cap);
    if (plan && plan->nb > 0) blocks = std::min(blocks, plan->nb);
  }
  P.ng = (plan && plan->ng > 0) ? plan->ng : default_groups(blocks);
  if (P.ng > blocks) P.ng = 1;  // (every group needs its first workgroup)
  P.xcd_local = 0; P.lm_everywhere = 0;
  {
    // Grids of two workgroups per CU (257 .. 512 workgroups: the 17k-point headline has 474): the second workgroup of a CU loses VALU
    // arbitration to the first (older waves win), finishes its main loop ~2 us later and keeps the whole trip waiting; s_setprio 1 for it
    // makes the pair finish together: LM launch 136 -> 127 us (profiles/r04_priority_ab.txt). With three workgroups per CU the same
    // priority costs 7 % (100k x 100k DIRECT27), for everybody at once it changes nothing: only this shape gets it.
    static const int prio = [] { const char* v = getenv("FVH_COST_PRIO"); return v ? (atoi(v) != 0 ? 1 : 0) : -1; }();  // A/B knob: 0 never, 1 always (default: by the rule below)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
    P.prio_from = cus;
    P.prio_mode = prio >= 0 ? prio : ((persistent && blocks > cus && blocks <= 2 * cus) ? 1 : 0);
    // The LM step on EVERY workgroup (each polls the group rows itself: no broadcast hand-off) for grids of at most two workgroups per
    // CU: 17k headline 126.1 -> 123.5 us, NDT LiDAR frames 81.8 -> 79.0 us. With three per CU (100k / 1M points: 768 workgroups) the
    // redundant steps cost more than the hand-off they replace (210.5 -> 216 us, 223 -> 227 us): those keep the collectors' broadcast.
    // FVH_LM_EVERYWHERE: 0 never, 1 (default) by this rule, 2 always. (profiles/r04_lm_everywhere.txt)
    static const int everywhere = [] { const char* v = getenv("FVH_LM_EVERYWHERE"); return v ? atoi(v) : 1; }();
    P.lm_everywhere = (persistent && P.ng > 1 && (everywhere == 2 || (everywhere == 1 && blocks <= 2 * cus))) ? 1 : 0;
  }
  const int launch_blocks = blocks;
  if (persistent && plan) P.xcd_local = plan->local;
  // abort word = 0 (the last 8 bytes of the state; never covered by the state write-back)
  if (e->abort_word_dirty) {
    HIP_OR_FAIL(e, hipMemsetAsync(reinterpret_cast<char*>(e->state.p) + sizeof(LmState) - 8, 0, 8, e->stream));
    e->abort_word_dirty = false;
  }
  if (persistent) {
    { const char* v = getenv("FVH_PERSIST_WATCHDOG_TICKS"); P.watchdog_ticks = v ? strtoull(v, nullptr, 10) : PERSIST_WATCHDOG_TICKS; }  // test hook: 0 forces the abort + fallback path
    // multi-GPU: workgroup 0 may legitimately wait for a late peer (up to the peer watchdog); the collectors waiting for its all-reduced
    // row and the workgroups waiting for their broadcast must outlast that, or a 50 ms skew between ranks would look like a stuck local barrier
    if (P.peer.n > 1 && P.watchdog_ticks) P.watchdog_ticks = std::max(P.watchdog_ticks, 2 * P.peer_watchdog_ticks);
    static const int zc = [] { const char* v = getenv("FVH_ZEROCOPY_RESULT"); return v ? atoi(v) : 1; }();
    P.result_host = (zc && (!e->prof.on || e->prof.cost_only)) ? e->result_dev : nullptr;  // (full stage profiling drains the stream per call anyway; the two events of level 2 do not need it)
    e->zero_copy_armed = P.result_host != nullptr;
    P.bcast = e->bcast.as<double>();
    P.launch_tag = ++e->persist_seq;
    e->last_persist_blocks = blocks;
    (void)e->gang_begin(false);  // (the persistent launches share the chip through the SlotPool; other handles' cooperative sorts stay away while this runs)
    {
      ProfScope ps(e, "cost");
      // (items of ONE offset take the instantiation unrolled for one lookup; both routes of an align pick by the same shape.
      // Gauss-Newton aligns take their own instantiations: the Levenberg-Marquardt ones do not carry the other optimiser's code)
#define FVH_LAUNCH_COST(PERS, GRID)                                                                                                   \
  do {                                                                                                                                \
    const bool f32 = e->float_cost();                                                                                                 \
    if ((host_phase < 0 ? e->align_optimizer : 0) == 0) {                                                                             \
      if (P.group == 1) { if (f32) cost_kernel<float, MODE, PERS, 1><<<GRID, 256, 0, e->stream>>>(P); else cost_kernel<double, MODE, PERS, 1><<<GRID, 256, 0, e->stream>>>(P); } \
      else { if (f32) cost_kernel<float, MODE, PERS><<<GRID, 256, 0, e->stream>>>(P); else cost_kernel<double, MODE, PERS><<<GRID, 256, 0, e->stream>>>(P); }                    \
    } else {                                                                                                                          \
      if (P.group == 1) { if (f32) cost_kernel<float, MODE, PERS, 1, true><<<GRID, 256, 0, e->stream>>>(P); else cost_kernel<double, MODE, PERS, 1, true><<<GRID, 256, 0, e->stream>>>(P); } \
      else { if (f32) cost_kernel<float, MODE, PERS, COST_CH, true><<<GRID, 256, 0, e->stream>>>(P); else cost_kernel<double, MODE, PERS, COST_CH, true><<<GRID, 256, 0, e->stream>>>(P); } \
    }                                                                                                                                 \
  } while (0)
      FVH_LAUNCH_COST(true, launch_blocks);
    }
    e->gang_end();
  } else {
    ProfScope ps(e, "cost");
    FVH_LAUNCH_COST(false, blocks);
  }
  HIP_OR_FAIL(e, hipGetLastError());
  return FVH_OK;
}

int allreduce_sums(Engine* e) {
  LmState* st = e->state.as<LmState>();
  int rc = g_rccl.AllReduce(st->sums, st->sums, PART_STRIDE, /*ncclDouble*/ 8, /*ncclSum*/ 0, e->comm, e->stream);
  if (rc != 0) return e->fail(FVH_ERR_COMM, "ncclAllReduce failed with code " + std::to_string(rc));
  return FVH_OK;
}

template <int MODE>
int do_update_correspondences(Engine* e, const CostSource& src, VoxelMapDev& vm, const double* T16) {
  if (!T16) return e->fail(FVH_ERR_INVALID_ARGUMENT, "update_correspondences: null pose");
  if (!vm.valid) return e->fail(FVH_ERR_BAD_STATE, "update_correspondences: target voxel map not built");
  HIP_OR_FAIL(e, e->corr.ensure(2 * sizeof(int) * (size_t)std::max(src.n_upper, 1) * e->n_off));
  e->lin = pose_from_colmajor16(T16);
  e->corr_sel = 0;
  int rc = launch_cost<MODE>(e, src, vm, PH_FIND_ONLY, &e->lin, &e->lin);
  if (rc) return rc;
  e->has_corr = true;
  e->corr_kind = 0;
  e->corr_n_src = src.n_upper;
  return FVH_OK;
}

template <int MODE>
int do_compute_error(Engine* e, const CostSource& src_in, VoxelMapDev& vm, const double* T16, double* H36, double* b6, double* error, const Rebuild& rebuild_safe) {
  CostSource src = src_in;
  if (!T16 || !error) return e->fail(FVH_ERR_INVALID_ARGUMENT, "compute_error: null argument");
  if (!e->has_corr) return e->fail(FVH_ERR_BAD_STATE, "compute_error: call update_correspondences first");
  if (!vm.valid) return e->fail(FVH_ERR_BAD_STATE, "compute_error: the target voxel map / target records were invalidated (target cloud replaced); rebuild and call update_correspondences");
  const bool deriv = (H36 != nullptr && b6 != nullptr);
  PoseD ev = pose_from_colmajor16(T16);
  LmState* h = reinterpret_cast<LmState*>(e->pinned);
  for (int attempt = 0; attempt < 2; attempt++) {
    const bool sharded = MODE == MODE_VGICP && e->peer.attached() && src.shardable;
    int rc = launch_cost<MODE>(e, src, vm, deriv ? PH_EVAL_DERIV : PH_EVAL_ERROR, &e->lin, &ev, nullptr, false, e->peer.x);
    if (rc) return rc;
    if (sharded) e->peer.x++;  // one sums exchange per evaluation, on every rank
    if (e->comm) { rc = allreduce_sums(e); if (rc) return rc; }
    HIP_OR_FAIL(e, hipMemcpyAsync(h, e->state.p, sizeof(LmState), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    if (h->aborted) {
      e->abort_word_dirty = true;
      e->peer.x = (e->peer.x + 8192) & ~1ull;
      return e->fail(FVH_ERR_COMM, "compute_error: a peer rank did not deliver its sums (every rank must make the same sequence of calls)");
    }
    vm.nv_hint = h->vm_num_voxels;
    if (h->vm_dropped == 0 || attempt == 1) break;
    // the hint-sized table overflowed: rebuild at the safe size, redo the correspondences, evaluate again
    rc = rebuild_safe();
    if (rc) return rc;
    src.refresh();
    rc = launch_cost<MODE>(e, src, vm, PH_FIND_ONLY, &e->lin, &e->lin);
    if (rc) return rc;
    e->has_corr = true;
  }
  if (h->vm_dropped) return e->fail(FVH_ERR_BAD_STATE, "voxel map overflow persists after safe rebuild");
  *error = h->sums[0];
  if (deriv) {
    double Hr[36];
    unpack_sums(h->sums, Hr, b6);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) H36[j * 6 + i] = Hr[i * 6 + j];  // column-major (symmetric)
  }
  return FVH_OK;
}

// What align_begin() leaves for align_finish(): an align is a launch (the persistent LM kernel) and a wait for its result; the C ABI
// offers the two halves separately (fvh_ndt_align_async / _wait) so that the host can queue the NEXT frame's preparation on the
// handle's second stream while the LM kernel runs.
struct AlignCtx {
  bool active = false;
  fvh_lm_params p;
  double guess16[16];
  bool degenerate = false, persistent = false, sharded = false, no_persist = false, retried = false, forced = false;
  long long budget = 0;
  GridPlan plan;
  int grant_dev = 0;
  SlotPool::Grant grant;
  void release_slots() { if (grant.n > 0) g_slots.release(grant_dev, grant); grant = SlotPool::Grant{}; }
};

template <int MODE>
int do_align(Engine* e, const CostSource& src, VoxelMapDev& vm, const double* guess16, const fvh_lm_params* params, fvh_lm_result* result, const Rebuild& rebuild_safe,
             bool retried = false, bool no_persist = false, const GridPlan* forced_plan = nullptr);

// first half: validate, pick the route and the grid, launch the persistent LM kernel (the multi-launch route queues nothing here)
template <int MODE>
int align_begin(Engine* e, AlignCtx& c, const CostSource& src, VoxelMapDev& vm, const double* guess16, const fvh_lm_params* params,
                bool retried = false, bool no_persist = false, const GridPlan* forced_plan = nullptr /* multi-launch retry of an aborted persistent launch: its layout */) {
  if (!guess16) return e->fail(FVH_ERR_INVALID_ARGUMENT, "align: null argument");
  if (!vm.valid) return e->fail(FVH_ERR_BAD_STATE, "align: target voxel map not built");
  c = AlignCtx{};
  fvh_lm_params& p = c.p;
  if (params) p = *params; else fvh_default_lm_params(&p);
  std::memcpy(c.guess16, guess16, sizeof(c.guess16));
  e->align_optimizer = p.optimizer != 0 ? 1 : 0;  // (every launch of this align -- begin, finish, fall-backs -- takes that optimiser's instantiation)
  c.retried = retried; c.no_persist = no_persist; c.grant_dev = e->device;
  HIP_OR_FAIL(e, e->corr.ensure(2 * sizeof(int) * (size_t)std::max(src.n_upper, 1) * e->n_off));
  LmState* st = e->state.as<LmState>();
  const PoseD guess = pose_from_colmajor16(guess16);
  c.degenerate = p.max_iterations <= 0;  // nothing to launch: only the state has to say "done"
  if (c.degenerate) {
    lm_init_kernel<<<1, 64, 0, e->stream>>>(st, guess, p.rotation_epsilon, p.transformation_epsilon, p.lm_init_lambda_factor, p.max_iterations, p.lm_max_iterations, e->ticket.as<unsigned>(), p.optimizer != 0 ? 1 : 0);
    HIP_OR_FAIL(e, hipGetLastError());
  }
  c.budget = (long long)std::max(p.max_iterations, 0) * (1 + (long long)std::max(p.lm_max_iterations, 0)) + 1;
  if (e->lm_trace_on) HIP_OR_FAIL(e, e->lm_trace.ensure(sizeof(double) * 6 * (size_t)std::max<long long>(c.budget, 1)));
  e->lm_trace_rows = 0;
  // One persistent launch for the whole LM loop when the problem is in the latency-bound regime and there is no RCCL
  // collective between evaluations. Concurrent aligns of this process (several handles, several host threads) split the
  // device's co-resident workgroup slots (SlotPool); after a watchdog abort -- typically ANOTHER PROCESS on the same GPU, which
  // the pool cannot see -- the handle backs off: it skips the persistent route for 1, 2, 4, ... 64 aligns before trying again,
  // so a shared GPU costs one 50 ms stall now and then instead of one per registration.
  static const int persist_env = [] { const char* v = getenv("FVH_PERSISTENT"); return v ? atoi(v) : 1; }();
  c.sharded = MODE == MODE_VGICP && e->peer.attached() && src.shardable;
  c.persistent = persist_env != 0 && !c.degenerate && !e->comm && !no_persist && c.budget < 4000 && (long long)src.n_upper * e->n_off <= PERSIST_MAX_ITEMS;
  if (c.persistent && !c.sharded && e->persist_skip > 0) { e->persist_skip--; c.persistent = false; }  // backing off (a sharded align must take the same route on every rank)
  if (forced_plan) { c.plan = *forced_plan; c.forced = true; }
  if (c.persistent) {
    int cap = persistent_capacity<MODE>(e);
    if (e->peer.attached()) cap = std::max(1, cap / std::max(1, e->peer.ranks_on_device));
    const int want = std::min(cost_shape(e, src, MODE, true).blocks, std::max(cap, 1));
    // Layouts (kernels_cost.hpp): chip-wide grids reduce per XCD (ng = 8; small ones: default_groups())
    const bool local_ok = xcd_local_wanted();
    c.grant = g_slots.acquire(e->device, std::max(cap, 1), want);
    int granted = c.grant.n;
    if (granted <= 0) {
      if (c.sharded) granted = want;  // ranks must not diverge: take the slots anyway (the watchdog covers the rare collision)
      else c.persistent = false;
    }
    c.plan = GridPlan{};
    c.plan.nb = granted;
    c.plan.ng = default_groups(granted);
    c.plan.local = (c.plan.ng == TICKET_GROUPS && local_ok) ? 1 : 0;  // (one chip-wide group spans XCDs: write-through)
  }
  if (c.persistent) {
    int rc = launch_cost<MODE>(e, src, vm, -1, &guess, nullptr, &p, true, e->peer.x, &c.plan);
    if (rc) { c.release_slots(); return rc; }
  }
  c.active = true;
  return FVH_OK;
}

// second half: wait for the persistent kernel's result (or run the multi-launch loop), answer aborts and table overflows, fill `result`
template <int MODE>
int align_finish(Engine* e, AlignCtx& c, const CostSource& src, VoxelMapDev& vm, fvh_lm_result* result, const Rebuild& rebuild_safe) {
  if (!c.active) return e->fail(FVH_ERR_BAD_STATE, "align: nothing in flight");
  struct Done { AlignCtx& c; ~Done() { c.release_slots(); c.active = false; } } done{c};
  if (!result) return e->fail(FVH_ERR_INVALID_ARGUMENT, "align: null argument");
  const fvh_lm_params& p = c.p;
  e->align_optimizer = p.optimizer != 0 ? 1 : 0;
  const bool persistent = c.persistent, sharded = c.sharded, degenerate = c.degenerate;
  const long long budget = c.budget;
  const PoseD guess = pose_from_colmajor16(c.guess16);
  LmState* st = e->state.as<LmState>();
  long long launched = 0;
  int batch = e->last_steps > 0 ? std::max(e->last_steps, e->prev_steps) + 1 : 8;
  LmState* h = reinterpret_cast<LmState*>(e->pinned);
  if (persistent) {
    bool have_result = false;
    if (e->result_dev && e->zero_copy_armed) {
      // spin on the sequence word the kernel writes after the state (mapped pinned memory); if the stream drains without it
      // (watchdog abort) fall through to the copy
      volatile unsigned long long* seq = reinterpret_cast<volatile unsigned long long*>(e->result_host) + sizeof(LmState) / 8;
      static const bool block = [] { const char* v = getenv("FVH_HOST_WAIT"); return v && std::string(v) == "block"; }();  // FVH_HOST_WAIT=block: sleep in hipStreamSynchronize instead of spinning a core on the result word
      if (block) (void)hipStreamSynchronize(e->stream);
      // (the stream is only asked now and then -- it answers "drained" when a launch ended without its result word, i.e. aborted: every query
      // takes the runtime's lock, which concurrent aligns of other host threads also need for their launches)
      static const unsigned long long query_mask = [] { const char* v = getenv("FVH_RESULT_QUERY_SPINS"); unsigned long long n = v ? strtoull(v, nullptr, 10) : 1024ull; unsigned long long m = 1; while (m < n) m <<= 1; return m - 1; }();
      for (unsigned long long spins = 0;; spins++) {
        if (*seq == e->persist_seq) { have_result = true; break; }
        if ((spins & query_mask) == query_mask && hipStreamQuery(e->stream) != hipErrorNotReady) { have_result = (*seq == e->persist_seq); break; }  // drained (or failed: the copy below reports it)
      }
      if (have_result) {
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(h, e->result_host, sizeof(LmState) - 8);
        h->gen = 0; h->aborted = 0;
      }
    }
    if (!have_result) {
      HIP_OR_FAIL(e, hipMemcpyAsync(h, st, sizeof(LmState), hipMemcpyDeviceToHost, e->stream));
      HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    }
    if (h->aborted || h->phase != PH_DONE) {  // the barrier watchdog fired (workgroups not co-resident): redo with one launch per transition
      e->persist_aborts++;
      e->abort_word_dirty = true;
      e->persist_backoff = std::min(std::max(2 * e->persist_backoff, 1), 64);
      e->persist_skip = e->persist_backoff;
      // multi-GPU: an abort on ANY rank reaches every rank within a watchdog period (its mailbox stays empty), so all ranks
      // arrive here and restart together; the exchange counter jumps past whatever this launch may have used
      if (sharded) e->peer.x = (e->peer.x + 8192) & ~1ull;
      if (h->aborted == 3u) g_xcd_local_strikes.fetch_add(1);  // the members of a group did not share an XCD: a few of these and the XCD-local flavour is off for good
      const GridPlan plan = c.plan;
      const bool retried = c.retried;
      c.release_slots();
      return do_align<MODE>(e, src, vm, c.guess16, &p, result, rebuild_safe, retried, true, &plan);  // the same layout: the same partition of the items, the same sums
    }
    launched = 1;
    e->persist_backoff = 0;  // a clean persistent run: the device is ours again
    if (sharded) e->peer.x += (unsigned long long)(p.optimizer ? h->num_linearize : 1 + h->num_error_evals);  // one exchange per trip
  }
  while (!persistent) {
    for (int s = 0; s < batch; s++) {
      // the first launch carries the initial guess and the LM parameters and (re)initialises the device state
      const bool first = (launched == 0 && s == 0 && !degenerate);
      int rc = launch_cost<MODE>(e, src, vm, -1, first ? &guess : nullptr, nullptr, first ? &p : nullptr, false, e->peer.x + (unsigned long long)(launched + s), c.forced ? &c.plan : nullptr);
      if (rc) return rc;
      if (e->comm) {
        rc = allreduce_sums(e);
        if (rc) return rc;
        if (p.optimizer) lm_update_kernel<true><<<1, 64, 0, e->stream>>>(st); else lm_update_kernel<false><<<1, 64, 0, e->stream>>>(st);
      }
    }
    launched += batch;
    HIP_OR_FAIL(e, hipMemcpyAsync(h, st, sizeof(LmState), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    if (h->aborted) {  // (only the peer exchange raises it on this route)
      e->abort_word_dirty = true;
      e->peer.x = (e->peer.x + 8192) & ~1ull;
      return e->fail(FVH_ERR_COMM, "align: a peer rank did not deliver its sums (every rank must make the same sequence of calls)");
    }
    if (h->phase == PH_DONE || launched >= budget) break;
    batch = 3;
  }
  if (!persistent && sharded && h->num_linearize > 0) e->peer.x += (unsigned long long)(p.optimizer ? h->num_linearize : 1 + h->num_error_evals);  // launches after PH_DONE leave before the exchange
  vm.nv_hint = h->vm_num_voxels;
  if (src.source_map) src.source_map->nv_hint = h->vm_num_voxels2;
  if (h->vm_dropped > 0) {  // hint-sized table overflowed: rebuild at the safe size and run again (rare)
    if (c.retried) return e->fail(FVH_ERR_BAD_STATE, "voxel map overflow persists after safe rebuild");
    const bool no_persist = c.no_persist, forced = c.forced;
    const GridPlan plan = c.plan;
    c.release_slots();
    int rc = rebuild_safe();
    if (rc) return rc;
    CostSource fresh = src;
    fresh.refresh();
    return do_align<MODE>(e, fresh, vm, c.guess16, &p, result, rebuild_safe, true, no_persist, forced ? &plan : nullptr);
  }
  e->gang_clear();
  e->prev_steps = e->last_steps;
  e->last_steps = p.optimizer ? std::max(1, (int)h->num_linearize) : 1 + h->num_error_evals;  // launches this align needed: the first linearize + one fused launch per trial (Gauss-Newton: one per linearisation)
  e->lin = h->x_lin;
  e->corr_sel = h->corr_cur;
  e->has_corr = true;  // correspondences of the last consumed linearisation stay valid for compute_error()
  e->corr_kind = 0;    // voxel-bucket ids (a nearest-point list of an earlier gicp_update_correspondences is gone)
  e->corr_n_src = src.n_upper;
  pose_to_colmajor16(h->x0, result->T);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) result->H[j * 6 + i] = h->final_H[i * 6 + j];
  result->final_error = h->y0;
  result->converged = h->converged;
  result->nr_iterations = h->nr_iterations;
  result->num_linearize = h->num_linearize;
  result->num_error_evals = h->num_error_evals;
  result->lm_failed = h->lm_failed;
  result->num_launches = (int)launched;
  e->lm_trace_rows = e->lm_trace_on ? h->num_error_evals : 0;
  return FVH_OK;
}

template <int MODE>
int do_align(Engine* e, const CostSource& src, VoxelMapDev& vm, const double* guess16, const fvh_lm_params* params, fvh_lm_result* result, const Rebuild& rebuild_safe,
             bool retried, bool no_persist, const GridPlan* forced_plan) {
  if (!result) return e->fail(FVH_ERR_INVALID_ARGUMENT, "align: null argument");
  AlignCtx c;
  int rc = align_begin<MODE>(e, c, src, vm, guess16, params, retried, no_persist, forced_plan);
  if (rc) return rc;
  return align_finish<MODE>(e, c, src, vm, result, rebuild_safe);
}

// exact 1-NN of every (transformed) source point in the target (kernels_cov.hpp: nn1_rows_kernel -- four queries per wave, one per 16-lane row)
void launch_nn1(Engine* e, const CloudDev& src, const CloudDev& tgt, const float* T12, double thr_sq, int* corr, float* best_out, const LmLink& lm) {
  nn1_rows_kernel<<<(src.n + 15) / 16, 256, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.bbox2.as<float4>(), tgt.n, T12, thr_sq, corr, best_out, lm);
}

int do_fitness(Engine* e, CloudDev& src, CloudDev& tgt, const double* T16, double max_range, double* score) {
  if (!T16 || !score) return e->fail(FVH_ERR_INVALID_ARGUMENT, "fitness_score: null argument");
  if (!src.has_pts || !tgt.has_pts || src.n == 0 || tgt.n == 0) return e->fail(FVH_ERR_BAD_STATE, "fitness_score: clouds not set");
  float T12[12];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T12[i * 4 + j] = (float)T16[j * 4 + i]; T12[i * 4 + 3] = (float)T16[12 + i]; }
  char* base = (char*)e->fit.p;
  HIP_OR_FAIL(e, hipMemsetAsync(base, 0, 16, e->stream));
  HIP_OR_FAIL(e, hipMemcpyAsync(base + 16, T12, sizeof(T12), hipMemcpyHostToDevice, e->stream));
#ifdef FVH_TEST_KERNELS  // test build only: FVH_FIT_MODE=0 full sweep, 2: eight queries per wave
  static const int fit_mode = [] { const char* v = getenv("FVH_FIT_MODE"); return v ? atoi(v) : 1; }();
#else
  constexpr int fit_mode = 1;
#endif
  if (fit_mode != 0) {
    int rc = ensure_sorted(e, src);
    if (!rc) rc = ensure_sorted(e, tgt);
    if (rc) return rc;
  }
  {
    ProfScope ps(e, "fitness");
#ifdef FVH_TEST_KERNELS
    const int waves = (src.n + FIT_Q - 1) / FIT_Q;
    if (fit_mode == 0) {
      fitness_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(src.pts.as<float4>(), src.n, tgt.pts.as<float4>(), tgt.n, (const float*)(base + 16), max_range, (double*)base);
    } else if (fit_mode == 2) {
      fitness_tiled_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.n, (const float*)(base + 16), max_range,
                                                                    (double*)base);
    } else
#endif
    {  // the exact 1-NN search of the GICP path (64 queries per wave), then a fixed-order reduction
      HIP_OR_FAIL(e, e->fit_best.ensure(sizeof(float) * (size_t)src.n));
#ifdef FVH_TEST_KERNELS
      if (fit_mode == 3)
        nn1_corr_kernel<<<src.n, 64, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.bbox2.as<float4>(), tgt.n,
                                                                (const float*)(base + 16), 0.0, nullptr, e->fit_best.as<float>());
      else
#endif
      launch_nn1(e, src, tgt, (const float*)(base + 16), 0.0, nullptr, e->fit_best.as<float>(), LmLink{nullptr, nullptr, nullptr, nullptr, 0});
      fitness_reduce_kernel<<<1, 1024, 0, e->stream>>>(e->fit_best.as<float>(), src.n, max_range, (double*)base);
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  double out[2];
  HIP_OR_FAIL(e, hipMemcpyAsync(out, base, 16, hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  *score = out[1] > 0 ? out[0] / out[1] : 1.7976931348623157e308;
  return FVH_OK;
}

int comm_init(Engine* e, const void* id128, int nranks, int rank) {
  if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return e->fail(FVH_ERR_INVALID_ARGUMENT, "comm_init: bad arguments");
  if (!g_rccl.load()) return e->fail(FVH_ERR_COMM, std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "missing symbols"));
  if (e->comm) { g_rccl.CommDestroy(e->comm); e->comm = nullptr; }
  Rccl::UID uid;
  std::memcpy(uid.b, id128, 128);
  HIP_OR_FAIL(e, hipSetDevice(e->device));
  int rc = g_rccl.CommInitRank(&e->comm, nranks, uid, rank);
  if (rc != 0) { e->comm = nullptr; return e->fail(FVH_ERR_COMM, "ncclCommInitRank failed with code " + std::to_string(rc)); }
  e->nranks = nranks; e->rank = rank;
  return FVH_OK;
}

int profile_get(Engine* e, const char* cls, double* total_ms, int* launches) {
  if (!cls) return e->fail(FVH_ERR_INVALID_ARGUMENT, "profile_get: null class");
  if (e->side) HIP_OR_FAIL(e, hipStreamSynchronize(e->side));  // (the map build's events may live there)
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  double tot = 0; int n = 0;
  auto it = e->prof.recs.find(cls);
  if (it != e->prof.recs.end())
    for (size_t i = 0; i < it->second.used; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, it->second.ev[i].first, it->second.ev[i].second) == hipSuccess) { tot += ms; n++; }
    }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return FVH_OK;
}

// ---------------------------------------------------------------------------------------------
// FastGICP on the device: nearest-target-point correspondences + the VGICP cost kernel on per-point records
// ---------------------------------------------------------------------------------------------
// sorted clouds + per-target-point records in the voxel-bucket layout (1 MB at 17k points: rebuilt every time rather than tracked)
int gicp_prepare(Engine* e, CloudDev& src, CloudDev& tgt, VoxelMapDev& records, double max_dist, const char* who) {
  if (!src.has_pts || !tgt.has_pts || src.n == 0 || tgt.n == 0) return e->fail(FVH_ERR_BAD_STATE, std::string(who) + ": clouds not set");
  if (!src.has_cov || !tgt.has_cov) return e->fail(FVH_ERR_BAD_STATE, std::string(who) + ": covariances not set");
  if (!(max_dist > 0)) return e->fail(FVH_ERR_INVALID_ARGUMENT, std::string(who) + ": max correspondence distance must be > 0");
  int rc = ensure_sorted(e, src);
  if (!rc) rc = ensure_sorted(e, tgt);
  if (rc) return rc;
  HIP_OR_FAIL(e, records.table.ensure(sizeof(float4) * 4 * (size_t)tgt.n));
  HIP_OR_FAIL(e, records.counters.ensure(2 * 16 * sizeof(int)));
  HIP_OR_FAIL(e, hipMemsetAsync(records.counters.p, 0, 2 * 16 * sizeof(int), e->stream));
  gicp_records_kernel<<<(tgt.n + 255) / 256, 256, 0, e->stream>>>(tgt.pts.as<float4>(), tgt.cov.as<float4>(), tgt.n, records.table.as<float4>());
  records.capacity = 1; records.res = 1.0; records.valid = true;
  HIP_OR_FAIL(e, e->corr.ensure(2 * sizeof(int) * (size_t)src.n));
  return FVH_OK;
}

// FastGICP::computeTransformation with the whole LM loop on the device (fast_gicp_impl.hpp:118-240 driven by
// lsq_registration_impl.hpp:53-168). Per LM transition TWO launches and no host round trip: nn1_corr_kernel searches the
// nearest target point of every source point at the pose the LM state on the device says comes next (x0 for a linearisation,
// the trial pose for the fused trial + speculative linearisation) and the cost kernel consumes those ids (external_find).
// Round 1 drove this from the host: two blocking round trips per iteration.
int gicp_align(Engine* e, CloudDev& src, CloudDev& tgt, VoxelMapDev& records, const CostSource& cs, double max_dist, const double* guess16, const fvh_lm_params* params,
               fvh_lm_result* result) {
  if (!guess16 || !result) return e->fail(FVH_ERR_INVALID_ARGUMENT, "gicp_align: null argument");
  int rc = gicp_prepare(e, src, tgt, records, max_dist, "gicp_align");
  if (rc) return rc;
  fvh_lm_params p;
  if (params) p = *params; else fvh_default_lm_params(&p);
  e->align_optimizer = p.optimizer != 0 ? 1 : 0;
  LmState* st = e->state.as<LmState>();
  const PoseD guess = pose_from_colmajor16(guess16);
  float T12[12];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T12[i * 4 + j] = (float)guess16[j * 4 + i]; T12[i * 4 + 3] = (float)guess16[12 + i]; }
  char* base = (char*)e->fit.p;
  HIP_OR_FAIL(e, hipMemcpyAsync(base + 16, T12, sizeof(T12), hipMemcpyHostToDevice, e->stream));
  const double thr = std::min(max_dist, 1.8446743e19);
  const LmLink link{&st->phase, &st->corr_cur, st->x0.r, st->xi.r, (size_t)src.n};
  const long long budget = (long long)std::max(p.max_iterations, 0) * (1 + (long long)std::max(p.lm_max_iterations, 0)) + 1;
  if (e->lm_trace_on) HIP_OR_FAIL(e, e->lm_trace.ensure(sizeof(double) * 6 * (size_t)std::max<long long>(budget, 1)));
  e->lm_trace_rows = 0;
  LmState* h = reinterpret_cast<LmState*>(e->pinned);
  if (p.max_iterations <= 0) {
    lm_init_kernel<<<1, 64, 0, e->stream>>>(st, guess, p.rotation_epsilon, p.transformation_epsilon, p.lm_init_lambda_factor, p.max_iterations, p.lm_max_iterations, e->ticket.as<unsigned>(), p.optimizer != 0 ? 1 : 0);
    HIP_OR_FAIL(e, hipGetLastError());
  }
  long long launched = 0;
  int batch = e->last_steps > 0 ? std::max(e->last_steps, e->prev_steps) + 1 : 8;
  for (;;) {
    for (int s = 0; s < batch && p.max_iterations > 0; s++) {
      const bool first = (launched == 0 && s == 0);
      {
        ProfScope ps(e, "gicp_nn");
        launch_nn1(e, src, tgt, reinterpret_cast<const float*>(base + 16), thr * thr, e->corr.as<int>(), nullptr, first ? LmLink{nullptr, nullptr, nullptr, nullptr, 0} : link);
      }
      rc = launch_cost<MODE_VGICP>(e, cs, records, -1, first ? &guess : nullptr, nullptr, first ? &p : nullptr);
      if (rc) return rc;
    }
    launched += batch;
    HIP_OR_FAIL(e, hipMemcpyAsync(h, st, sizeof(LmState), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    if (h->phase == PH_DONE || launched >= budget || p.max_iterations <= 0) break;
    batch = 3;
  }
  e->prev_steps = e->last_steps;
  e->last_steps = p.optimizer ? std::max(1, (int)h->num_linearize) : 1 + h->num_error_evals;
  e->lin = h->x_lin;
  e->corr_sel = h->corr_cur;
  e->has_corr = true;
  e->corr_kind = 1;
  e->corr_n_src = src.n;
  pose_to_colmajor16(h->x0, result->T);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) result->H[j * 6 + i] = h->final_H[i * 6 + j];
  result->final_error = h->y0;
  result->converged = h->converged;
  result->nr_iterations = h->nr_iterations;
  result->num_linearize = h->num_linearize;
  result->num_error_evals = h->num_error_evals;
  result->lm_failed = h->lm_failed;
  result->num_launches = (int)(2 * launched);
  e->lm_trace_rows = e->lm_trace_on ? h->num_error_evals : 0;
  return FVH_OK;
}

int gicp_update_correspondences(Engine* e, CloudDev& src, CloudDev& tgt, VoxelMapDev& records, const double* T16, double max_dist) {
  if (!T16) return e->fail(FVH_ERR_INVALID_ARGUMENT, "gicp_update_correspondences: null pose");
  int rc = gicp_prepare(e, src, tgt, records, max_dist, "gicp_update_correspondences");
  if (rc) return rc;
  float T12[12];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T12[i * 4 + j] = (float)T16[j * 4 + i]; T12[i * 4 + 3] = (float)T16[12 + i]; }  // trans.cast<float>()
  char* base = (char*)e->fit.p;
  HIP_OR_FAIL(e, hipMemcpyAsync(base + 16, T12, sizeof(T12), hipMemcpyHostToDevice, e->stream));
  const double thr = std::min(max_dist, 1.8446743e19);  // threshold^2 must stay finite in fp64 (reference default: float max)
  {
    ProfScope ps(e, "gicp_nn");
#ifdef FVH_TEST_KERNELS  // test build only: FVH_GICP_NN_MODE=0 selects the superseded eight-queries-per-wave search
    static const int nn_mode = [] { const char* v = getenv("FVH_GICP_NN_MODE"); return v ? atoi(v) : 1; }();
    if (nn_mode == 3) {
      nn1_corr_kernel<<<src.n, 64, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.bbox2.as<float4>(), tgt.n,
                                                              reinterpret_cast<const float*>(base + 16), thr * thr, e->corr.as<int>());
    } else if (nn_mode != 1) {
      const int waves = (src.n + FIT_Q - 1) / FIT_Q;
      nn_corr_tiled_kernel<<<(waves + 3) / 4, 256, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.n, reinterpret_cast<const float*>(base + 16), thr * thr,
                                                                   e->corr.as<int>());
    } else
#endif
    {
      launch_nn1(e, src, tgt, reinterpret_cast<const float*>(base + 16), thr * thr, e->corr.as<int>(), nullptr, LmLink{nullptr, nullptr, nullptr, nullptr, 0});
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));  // T12 is a stack buffer
  e->lin = pose_from_colmajor16(T16);
  e->corr_sel = 0;
  e->has_corr = true;
  e->corr_kind = 1;
  e->corr_by_position = false;  // (nn1_rows_kernel writes the row of the ORIGINAL index)
  e->corr_n_src = src.n;
  return FVH_OK;
}

// ---------------------------------------------------------------------------------------------
// voxel-grid downsampling (kernels_downsample.hpp)
// ---------------------------------------------------------------------------------------------
struct DownsampleDev {
  CloudDev cloud;                 // the input, widened to float4
  DevBuf keys, idx, head, trig, scan, bsums, slots, out;
  int out_n = 0;
  void release() { cloud.release(); keys.release(); idx.release(); head.release(); trig.release(); scan.release(); bsums.release(); slots.release(); out.release(); }
};

// exclusive scan of n unsigned values (in -> out, may alias); the grand total lands in bsums[nb]
int device_scan(Engine* e, DevBuf& bsums, const unsigned* in, int n, unsigned* out, const unsigned** total) {
  const int nb = (n + SCAN_BLOCK_ITEMS - 1) / SCAN_BLOCK_ITEMS;
  HIP_OR_FAIL(e, bsums.ensure(sizeof(unsigned) * (size_t)(nb + 1)));
  scan_block_sums_kernel<<<nb, 256, 0, e->stream>>>(in, n, bsums.as<unsigned>());
  radix_scan_kernel<<<1, 1024, 0, e->stream>>>(bsums.as<unsigned>(), nb + 1);
  scan_apply_kernel<<<nb, 256, 0, e->stream>>>(in, n, bsums.as<unsigned>(), out);
  HIP_OR_FAIL(e, hipGetLastError());
  *total = bsums.as<unsigned>() + nb;
  return FVH_OK;
}

// stable LSD radix sort of (key, idx) pairs on `bits` key bits; returns the index (0/1) of the buffer pair holding the result
int radix_sort_pairs(Engine* e, unsigned* keys[2], int* idx[2], int n, int bits, int* result, hipStream_t on, DevBuf* hist_buf) {
  hipStream_t const st = on ? on : e->stream;
  DevBuf& hb = hist_buf ? *hist_buf : e->sort_hist;  // (a caller on another stream than the handle's brings its own histograms)
  const int items = n <= 262144 ? 256 : (n <= 1048576 ? 512 : SORT_ITEMS_MAX);
  const int nwaves = (n + items - 1) / items;
  HIP_OR_FAIL(e, hb.ensure(sizeof(unsigned) * (size_t)RADIX_BINS * (nwaves + 1)));
  unsigned* bin_tot = hb.as<unsigned>() + (size_t)RADIX_BINS * nwaves;
  const int wblocks = (nwaves + 3) / 4;
  const int passes = std::max(1, (bits + RADIX_BITS - 1) / RADIX_BITS);
  for (int pass = 0; pass < passes; pass++) {
    const int in = pass & 1, out = in ^ 1, shift = pass * RADIX_BITS;
    radix_hist_kernel<RADIX_BITS><<<wblocks, 256, 0, st>>>(keys[in], n, shift, nwaves, items, hb.as<unsigned>());
    radix_binscan_kernel<<<RADIX_BINS / 4, 256, 0, st>>>(hb.as<unsigned>(), nwaves, bin_tot);
    radix_scan_kernel<<<1, 1024, 0, st>>>(bin_tot, RADIX_BINS);
    radix_scatter_kernel<RADIX_BITS><<<wblocks, 256, 0, st>>>(keys[in], idx[in], n, shift, nwaves, items, hb.as<unsigned>(), bin_tot, keys[out], idx[out], nullptr, nullptr);
  }
  HIP_OR_FAIL(e, hipGetLastError());
  *result = passes & 1;
  return FVH_OK;
}

inline float host_ordered_to_float(unsigned u) {
  const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  std::memcpy(&f, &v, 4);
  return f;
}

// pcl::ApproximateVoxelGrid: the fused six-launch chain of kernels_downsample.hpp (no memset, no key / index arrays; the count
// comes back through mapped host memory instead of a copy kernel + stream synchronisation)
int downsample_approx(Engine* e, DownsampleDev& d, const float* xyz, int n, int stride, bool on_device, float leaf, int* out_n, bool early = false) {
  if (n < 0 || (n > 0 && !xyz)) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: null points");
  if (stride != 3 && stride != 4) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: stride must be 3 or 4 floats");
  if (n == 0) return FVH_OK;
  const float* d_xyz = xyz;
  if (!on_device) {
    HIP_OR_FAIL(e, e->staging.ensure(sizeof(float) * stride * (size_t)n));
    HIP_OR_FAIL(e, hipMemcpyAsync(e->staging.p, xyz, sizeof(float) * stride * (size_t)n, hipMemcpyHostToDevice, e->stream));
    d_xyz = e->staging.as<float>();
  }
  const int nwaves = (n + AVG_ITEMS - 1) / AVG_ITEMS, nwords = (n + 31) / 32, nblocks = (nwords + 31) / 32;
  const int wblocks = (nwaves + 3) / 4;
  HIP_OR_FAIL(e, d.keys.ensure(sizeof(unsigned) * ((size_t)AVG_SLOTS * nwaves + AVG_SLOTS + (size_t)AVG_SLOTS * wblocks)));  // slot x wave histogram + slot totals + per-workgroup histograms (fused chain)
  HIP_OR_FAIL(e, d.idx.ensure(sizeof(float4) * (size_t)n));                                     // the points in slot order
  HIP_OR_FAIL(e, d.head.ensure((size_t)n));                                                     // run heads (bytes)
  HIP_OR_FAIL(e, d.trig.ensure(sizeof(unsigned) * (size_t)nblocks * 32));                       // trigger bits by original index (whole blocks of 32 words)
  HIP_OR_FAIL(e, d.scan.ensure(sizeof(unsigned) * (size_t)(nblocks + 1)));                      // trigger prefix per 1024 indices
  HIP_OR_FAIL(e, d.out.ensure(sizeof(float) * 3 * (size_t)n));
  const bool fresh = d.slots.p == nullptr;
  HIP_OR_FAIL(e, d.slots.ensure(sizeof(AvgState)));
  if (fresh) HIP_OR_FAIL(e, hipMemsetAsync(d.slots.p, 0, sizeof(AvgState), e->stream));  // (the ticket re-arms itself afterwards)
  unsigned* hist = d.keys.as<unsigned>();
  unsigned* totals = hist + (size_t)AVG_SLOTS * nwaves;
  unsigned* hist_wg = totals + AVG_SLOTS;
  // up to AVG_FUSED_MAX_POINTS points: four launches -- the scatter and the emit kernel recompute the two small prefix sums themselves
  // (kernels_downsample.hpp); FVH_AVG_FUSED=0 keeps round 2's six for A/B runs
  static const bool fused_on = [] { const char* v = getenv("FVH_AVG_FUSED"); return !v || atoi(v) != 0; }();
  const bool fused = fused_on && n <= AVG_FUSED_MAX_POINTS;
  float4* sorted = d.idx.as<float4>();
  unsigned char* head = d.head.as<unsigned char>();
  AvgState* st = d.slots.as<AvgState>();
  const float inv = 1.0f / leaf;
  const unsigned long long seq = ++e->persist_seq;
  volatile unsigned long long* hres = reinterpret_cast<volatile unsigned long long*>(e->result_host);
  {
    ProfScope ps(e, "downsample");
    const int pblocks = (n + 255) / 256;
    // early: the count travels to the host before the centroids exist and the call returns while the emit kernel runs (same-stream consumers only)
    early = early && on_device && e->result_dev && !e->prof.on;
    unsigned long long* final_result = (e->prof.on || early) ? nullptr : e->result_dev;
    if (fused) {
      avg_keys_hist_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, d.trig.as<unsigned>(), nwords, st, hist_wg);
      avg_scatter_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, st, sorted, hist_wg);
      avg_mark_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, n, inv, head, d.trig.as<unsigned>(), st);
      avg_emit_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, head, n, inv, d.trig.as<unsigned>(), nullptr, st, d.out.as<float>(), final_result, seq, nwords, early ? e->result_dev : nullptr);
    } else {
      avg_keys_hist_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, d.trig.as<unsigned>(), nwords, st);
      avg_binscan_kernel<<<AVG_SLOTS / 4, 256, 0, e->stream>>>(hist, nwaves, totals, st);
      avg_scatter_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, st, sorted);
      avg_mark_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, n, inv, head, d.trig.as<unsigned>(), st);
      avg_scan_kernel<<<1, 1024, 0, e->stream>>>(d.trig.as<unsigned>(), nwords, d.scan.as<unsigned>(), st, early ? e->result_dev : nullptr, seq);
      avg_emit_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, head, n, inv, d.trig.as<unsigned>(), d.scan.as<unsigned>(), st, d.out.as<float>(), final_result, seq);
    }
  }
  HIP_OR_FAIL(e, hipGetLastError());
  unsigned long long count = 0, bad = 0;
  bool have = false;
  if (e->result_dev && !e->prof.on) {  // spin on the sequence word the scan kernel writes after the count (mapped pinned memory)
    for (unsigned long long spins = 0;; spins++) {
      if (hres[2] == seq) { have = true; break; }
      if ((spins & 0x3ff) == 0x3ff && hipStreamQuery(e->stream) != hipErrorNotReady) { have = (hres[2] == seq); break; }
    }
    if (have) { std::atomic_thread_fence(std::memory_order_acquire); count = hres[0]; bad = hres[1]; }
  }
  if (!have) {
    unsigned h[2] = {0, 0};
    HIP_OR_FAIL(e, hipMemcpyAsync(h, &st->trig_total, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));  // trig_total, used_slots are adjacent
    unsigned hb = 0;
    HIP_OR_FAIL(e, hipMemcpyAsync(&hb, &st->bad, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    count = (unsigned long long)h[0] + h[1];
    bad = hb;
    if (!(e->result_dev && !e->prof.on) || hb) HIP_OR_FAIL(e, hipMemsetAsync(&st->bad, 0, sizeof(unsigned), e->stream));  // the emit kernel only re-arms the flag when it reports through mapped memory
  } else if (!on_device) {
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));  // the caller may free its host buffer on return; (the staging copy is long done, this only drains the emit kernel)
  }
  if (bad) { d.out_n = 0; return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: non-finite coordinates in the input"); }
  d.out_n = (int)count;
  *out_n = d.out_n;
  return FVH_OK;
}

int downsample(Engine* e, DownsampleDev& d, int method, const float* xyz, int n, int stride, bool on_device, float leaf, int* out_n, bool early = false) {
  if (e->stream_owner) e->stream_owner->quiet = false;  // work on a borrowed stream: its owner can no longer assume the stream has drained (Engine::quiet)
  if (!out_n) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: null out_n");
  if (method != FVH_VOXELGRID_EXACT && method != FVH_VOXELGRID_APPROXIMATE) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: unknown method");
  if (!(leaf > 0.f)) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: leaf size must be > 0");
  int rc = FVH_OK;
  d.out_n = 0;
  *out_n = 0;
  if (method == FVH_VOXELGRID_APPROXIMATE) return downsample_approx(e, d, xyz, n, stride, on_device, leaf, out_n, early);
  rc = upload_cloud(e, d.cloud, xyz, n, stride, on_device, false);
  if (rc) return rc;
  if (n == 0) return FVH_OK;
  ProfScope ps(e, "downsample");
  const float inv = 1.0f / leaf;
  const float4* pts = d.cloud.pts.as<float4>();
  HIP_OR_FAIL(e, d.keys.ensure(sizeof(unsigned) * 2 * (size_t)n + 128));
  HIP_OR_FAIL(e, d.idx.ensure(sizeof(int) * 2 * (size_t)n));
  HIP_OR_FAIL(e, d.head.ensure(sizeof(unsigned) * (size_t)n));
  HIP_OR_FAIL(e, d.scan.ensure(sizeof(unsigned) * (size_t)n));
  HIP_OR_FAIL(e, d.out.ensure(sizeof(float) * 3 * (size_t)n));
  unsigned* keys[2] = {d.keys.as<unsigned>(), d.keys.as<unsigned>() + n};
  int* idx[2] = {d.idx.as<int>(), d.idx.as<int>() + n};
  const int blocks = (n + 255) / 256;
  unsigned* h_total = reinterpret_cast<unsigned*>(e->pinned);
  unsigned* bad = d.keys.as<unsigned>() + 2 * (size_t)n + 8;  // set by the key kernels on a non-finite coordinate
  int sorted = 0;
  if (method == FVH_VOXELGRID_EXACT) {
    HIP_OR_FAIL(e, hipMemsetAsync(bad, 0, sizeof(unsigned), e->stream));
    // pcl::VoxelGrid: lattice over the bounding box (getMinMax3D), linear voxel index, points grouped by index
    unsigned* box = d.keys.as<unsigned>() + 2 * (size_t)n;
    HIP_OR_FAIL(e, hipMemsetAsync(box, 0xFF, 12, e->stream));
    HIP_OR_FAIL(e, hipMemsetAsync(box + 3, 0, 12, e->stream));
    cloud_bbox_kernel<<<std::min(256, blocks), 256, 0, e->stream>>>(pts, n, box);
    HIP_OR_FAIL(e, hipMemcpyAsync(h_total, box, 24, hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    VgGrid g;
    long long total = 1;
    for (int a = 0; a < 3; a++) {
      const float mn = host_ordered_to_float(h_total[a]), mx = host_ordered_to_float(h_total[3 + a]);
      if (!std::isfinite(mn) || !std::isfinite(mx)) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: non-finite coordinates");
      const float lo = std::floor(mn * inv), hi = std::floor(mx * inv);
      if (std::fabs(lo) > 2.0e9f || std::fabs(hi) > 2.0e9f) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: leaf size too small for the input (voxel index overflow)");
      g.minb[a] = (int)lo;
      g.divb[a] = (int)hi - g.minb[a] + 1;
      total *= g.divb[a];
      if (total > 0x7fffffffLL) return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: leaf size too small for the input (voxel index overflow)");  // PCL refuses too
    }
    int bits = 1;
    while (bits < 31 && (1LL << bits) < total) bits++;
    vg_keys_exact_kernel<<<blocks, 256, 0, e->stream>>>(pts, n, inv, g, keys[0], idx[0], bad);
    if ((rc = radix_sort_pairs(e, keys, idx, n, bits, &sorted))) return rc;
    vg_mark_exact_kernel<<<blocks, 256, 0, e->stream>>>(keys[sorted], n, d.head.as<unsigned>());
    const unsigned* total_dev = nullptr;
    if ((rc = device_scan(e, d.bsums, d.head.as<unsigned>(), n, d.scan.as<unsigned>(), &total_dev))) return rc;
    vg_emit_kernel<false><<<blocks, 256, 0, e->stream>>>(keys[sorted], idx[sorted], pts, n, d.head.as<unsigned>(), d.scan.as<unsigned>(), nullptr, nullptr, d.out.as<float>(), nullptr, nullptr);
    HIP_OR_FAIL(e, hipGetLastError());
    HIP_OR_FAIL(e, hipMemcpyAsync(h_total, total_dev, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipMemcpyAsync(h_total + 2, bad, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    d.out_n = (int)h_total[0];
  } else {
    // pcl::ApproximateVoxelGrid, slot-parallel (see kernels_downsample.hpp). Housekeeping on the stream is kept to ONE small
    // memset and ONE 16-byte copy: the key kernel clears the trigger flags, `bad` sits behind the slot flags, the scan's
    // spare entry is cleared by its first kernel, and the emit kernel gathers the three numbers the host needs.
    HIP_OR_FAIL(e, d.trig.ensure(sizeof(unsigned) * (size_t)n));
    HIP_OR_FAIL(e, d.slots.ensure(sizeof(unsigned) * (AVG_SLOTS + 1 + 1 + 4)));
    unsigned* bad2 = d.slots.as<unsigned>() + AVG_SLOTS + 1;
    unsigned* result = bad2 + 1;  // {trigger count, used slots, bad}
    HIP_OR_FAIL(e, hipMemsetAsync(d.slots.p, 0, sizeof(unsigned) * (AVG_SLOTS + 2), e->stream));
    vg_keys_approx_kernel<<<blocks, 256, 0, e->stream>>>(pts, n, inv, keys[0], idx[0], bad2, d.trig.as<unsigned>());
    if ((rc = radix_sort_pairs(e, keys, idx, n, RADIX_BITS, &sorted))) return rc;
    vg_mark_approx_kernel<<<blocks, 256, 0, e->stream>>>(keys[sorted], idx[sorted], pts, n, inv, d.head.as<unsigned>(), d.trig.as<unsigned>(), d.slots.as<unsigned>());
    const unsigned* trig_total = nullptr;
    if ((rc = device_scan(e, d.bsums, d.trig.as<unsigned>(), n, d.scan.as<unsigned>(), &trig_total))) return rc;
    radix_scan_kernel<<<1, 1024, 0, e->stream>>>(d.slots.as<unsigned>(), AVG_SLOTS + 1);
    vg_emit_kernel<true><<<blocks, 256, 0, e->stream>>>(keys[sorted], idx[sorted], pts, n, d.head.as<unsigned>(), d.scan.as<unsigned>(), d.slots.as<unsigned>(), trig_total, d.out.as<float>(), bad2, result);
    HIP_OR_FAIL(e, hipGetLastError());
    HIP_OR_FAIL(e, hipMemcpyAsync(h_total, result, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
    d.out_n = (int)(h_total[0] + h_total[1]);
  }
  if (h_total[2]) { d.out_n = 0; return e->fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: non-finite coordinates in the input"); }
  *out_n = d.out_n;
  return FVH_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// opaque handles
// ---------------------------------------------------------------------------------------------
struct fvh_vgicp {
  Engine e;
  double resolution = 1.0, kernel_width = 0.25, kernel_max_dist = 3.0;  // fast_vgicp_cuda.cu:22-26
  CloudDev source, target;
  VoxelMapDev voxelmap;
  VoxelMapDev gicp_records;  // per-target-point records for the nearest-point (GICP) cost
  double gicp_max_dist = 3.4028234663852886e38;
  CostSource cost_source() const {
    // multi-GPU: the tiles are ranges of the Morton order whatever the size of the cloud (spatially compact shards)
    const int* order = e.sharded() ? (source.has_sorted ? source.order.as<int>() : nullptr) : coherent_order(source);
    CostSource c{source.pts.as<float4>(), source.cov.as<float4>(), nullptr, source.n, nullptr, order};
    if (order) { c.sorted = source.sorted.as<float4>(); if (source.has_cov_sorted) c.cov_sorted = source.cov_sorted.as<float4>(); }
    c.shardable = true;
    return c;
  }
  CostSource gicp_cost_source(bool external_find = false) const {
    CostSource c{source.pts.as<float4>(), source.cov.as<float4>(), nullptr, source.n, nullptr, nullptr};
    c.n_off_override = 1;
    c.external_find = external_find;
    return c;
  }
  int voxel_mode = 0;        // VoxelAccumulationMode ordinal: 0 ADDITIVE, 1 ADDITIVE_WEIGHTED (same voxel type in the reference), 2 MULTIPLICATIVE
  // multi-GPU: the target map sharded by the ranks' spatial tiles + halo (fvh_vgicp_set_target_map_sharding)
  bool shard_map = false;
  int shard_margin = 2;      // voxels of pose motion the halo allows for on top of the reach of the neighbour offsets
  int shard_fallbacks = 0;   // aligns redone on the full map because a source element left the shard's inner box
  int build_map(double res, bool force_safe = false, hipStream_t on_side = nullptr, bool shard = false) {
    return voxel_mode == 2 ? build_voxelmap<2>(&e, target, voxelmap, res, false, force_safe, on_side, shard) : build_voxelmap<0>(&e, target, voxelmap, res, false, force_safe, on_side, shard);
  }
  Rebuild rebuild_safe() { return [this] { return build_map(voxelmap.res, true, nullptr, voxelmap.is_shard); }; }
  // A sharded align leaves this rank's SHARD behind as the live map; everything host-driven (update_correspondences, compute_error,
  // the voxel getters) works on the whole map: rebuilt here, at the resolution the live map was built with. (The correspondences of
  // the shard die with it: compute_error then asks for update_correspondences instead of reading the wrong buckets.)
  int whole_map() { return (voxelmap.valid && voxelmap.is_shard) ? build_map(voxelmap.res) : (int)FVH_OK; }
};

struct fvh_ndt {
  Engine e;
  double resolution = 1.0;          // ndt_cuda.cu:15
  int distance_mode = FVH_NDT_D2D;  // ndt_cuda.cu:21
  CloudDev source, target;
  VoxelMapDev source_vm, target_vm;
  // Pipelined frame streams (fvh_ndt_prepare_source_device / _adopt_prepared_source): the NEXT source cloud and its voxel map are built
  // in this slot on the handle's second stream (Engine::side) while the LM kernel of the current frame runs on the main one
  CloudDev next_source;
  VoxelMapDev next_vm;
  bool next_ready = false;
  hipEvent_t prep_done = nullptr;
  AlignCtx pending;  // fvh_ndt_align_async .. fvh_ndt_align_wait
  CostSource cost_source() const {
    const bool tiled = e.tile_n > 1;  // fvh_ndt_set_source_tile: this handle evaluates one spatial tile of the source
    if (distance_mode == FVH_NDT_P2D) {
      // P2D shards like VGICP: source POINTS by Morton tile (the order exists once the caller's align / update_correspondences sorted the cloud)
      CostSource cs{source.pts.as<float4>(), nullptr, nullptr, source.n, nullptr, tiled ? (source.has_sorted ? source.order.as<int>() : nullptr) : coherent_order(source)};
      cs.shardable = tiled;
      return cs;
    }
    CostSource cs{source_vm.compact_pts.as<float4>(), source_vm.compact_cov.as<float4>(), source_vm.counters_cur(), source.n, source_vm.counters_cur(), nullptr};
    if (tiled) { cs.order = source_vm.has_canon ? source_vm.canon.as<int>() : nullptr; cs.device_tile = true; }  // D2D: source VOXELS in canonical order
    // the source elements are the voxels of the source map; their number is on the device. A frame stream rebuilds that map per
    // frame with nearly the same voxel count: the count the last align saw (+ 25 %) sizes the grid -- a few thousand voxels instead
    // of the cloud's ~25k points: 7 one-offset items per voxel instead of 3 of <= 3 offsets (a shorter chain per trip), a smaller
    // grid at the barrier (<= 128 workgroups reduce in one level)
    if (source_vm.nv_hint > 0) cs.n_shape = (int)std::min<long long>(source.n, (long long)source_vm.nv_hint + source_vm.nv_hint / 4 + 256);
    cs.source_map = const_cast<VoxelMapDev*>(&source_vm);
    return cs;
  }
  // the source voxels ranked by key (kernels_voxelmap.hpp: vm_canonical_order_kernel): the list a tiled D2D handle cuts its chunk from
  int ensure_canon() {
    if (source_vm.has_canon) return FVH_OK;
    HIP_OR_FAIL(&e, source_vm.canon.ensure(sizeof(int) * (size_t)std::max(source.n, 1)));
    vm_canonical_order_kernel<<<(std::max(source.n, 1) + 255) / 256, 256, 0, e.stream>>>(source_vm.keys_cur(), source_vm.occupied.as<int>(), source_vm.counters_cur(), source_vm.canon.as<int>());
    HIP_OR_FAIL(&e, hipGetLastError());
    source_vm.has_canon = true;
    return FVH_OK;
  }
  Rebuild rebuild_safe() {
    return [this] {
      int rc = build_voxelmap<1>(&e, target, target_vm, target_vm.res, true, true);
      if (!rc && distance_mode == FVH_NDT_D2D) {
        rc = build_voxelmap<1>(&e, source, source_vm, source_vm.res, true, true);
        // A tiled handle walks the source voxels in canonical order, and the rebuild has just reordered the compact list that order indexes:
        // rank them again, into the SAME buffer (source.n has not changed: no reallocation), so that the CostSource the caller retries with
        // -- which holds that buffer's address -- cuts a partition of the NEW list (a stale permutation counted voxels twice / never)
        if (!rc && e.tile_n > 1) rc = ensure_canon();
      }
      return rc;
    };
  }
};

struct fvh_voxelgrid {
  Engine e;
  DownsampleDev d;
};

// CHECK_HANDLE_SOURCE_CHAIN: the calls that only touch the SOURCE cloud -- they may run beside a target-map build on the side
// stream (Engine::side); every other call orders the main stream after that build first.
#define CHECK_HANDLE_SOURCE_CHAIN(h) \
  if (!(h)) return FVH_ERR_INVALID_ARGUMENT; \
  { hipError_t _e = hipSetDevice((h)->e.device); if (_e != hipSuccess) return (h)->e.hipfail(_e, "hipSetDevice"); } \
  (h)->e.was_quiet = (h)->e.quiet; (h)->e.quiet = false;
// CHECK_HANDLE_HOST_ONLY: plain host-side setters that queue nothing: they leave the engine's stream bookkeeping alone.
#define CHECK_HANDLE_HOST_ONLY(h) \
  if (!(h)) return FVH_ERR_INVALID_ARGUMENT; \
  { hipError_t _e = hipSetDevice((h)->e.device); if (_e != hipSuccess) return (h)->e.hipfail(_e, "hipSetDevice"); }
#define CHECK_HANDLE(h) \
  CHECK_HANDLE_SOURCE_CHAIN(h) \
  { int _j = (h)->e.settle(); if (_j) return _j; }

extern "C" {

void fvh_default_lm_params(fvh_lm_params* p) {
  if (!p) return;
  p->max_iterations = 64; p->rotation_epsilon = 2e-3; p->transformation_epsilon = 5e-4; p->lm_max_iterations = 10; p->lm_init_lambda_factor = 1e-9; p->optimizer = 0;
}
int fvh_device_count(int* count) {
  if (!count) return FVH_ERR_INVALID_ARGUMENT;
  return hipGetDeviceCount(count) == hipSuccess ? FVH_OK : FVH_ERR_HIP;
}

// ---- VGICP -------------------------------------------------------------------------------------
int fvh_vgicp_create(int device, fvh_vgicp** out) {
  if (!out) return FVH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  fvh_vgicp* h = new (std::nothrow) fvh_vgicp();
  if (!h) return FVH_ERR_HIP;
  int rc = h->e.init(device);
  if (rc) { std::fprintf(stderr, "fvh_vgicp_create: %s\n", h->e.err.c_str()); h->e.shutdown(); delete h; return rc; }
  *out = h;
  return FVH_OK;
}
int fvh_vgicp_destroy(fvh_vgicp* h) {
  if (!h) return FVH_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(h->e.device);
  h->e.deferred = nullptr;
  if (h->e.side) (void)hipStreamSynchronize(h->e.side);
  if (h->e.stream) (void)hipStreamSynchronize(h->e.stream);
  h->source.release(); h->target.release(); h->voxelmap.release(); h->gicp_records.release();
  h->e.shutdown();
  delete h;
  return FVH_OK;
}
const char* fvh_vgicp_last_error(const fvh_vgicp* h) { return h ? h->e.err.c_str() : "null handle"; }
int fvh_vgicp_set_resolution(fvh_vgicp* h, double r) { CHECK_HANDLE_HOST_ONLY(h); if (!(r > 0)) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "resolution must be > 0"); h->resolution = r; return FVH_OK; }
int fvh_vgicp_set_kernel_params(fvh_vgicp* h, double w, double d) { CHECK_HANDLE_HOST_ONLY(h); h->kernel_width = w; h->kernel_max_dist = d; return FVH_OK; }
int fvh_vgicp_set_neighbor_search_method(fvh_vgicp* h, int m, double radius) { CHECK_HANDLE(h); return h->e.set_offsets(m, radius); }
int fvh_vgicp_set_precision(fvh_vgicp* h, int p) {
  CHECK_HANDLE(h);
  if (p != FVH_COMPUTE_FP64 && p != FVH_COMPUTE_FP32 && p != FVH_COMPUTE_CUDA_COMPAT) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "bad precision");
  if ((p == FVH_COMPUTE_CUDA_COMPAT) != (h->e.precision == FVH_COMPUTE_CUDA_COMPAT)) { h->voxelmap.invalidate(); h->e.has_corr = false; }  // the voxel records of the two arithmetics differ (kernels_compat.hpp)
  h->e.precision = p;
  return FVH_OK;
}

int fvh_vgicp_create_target_voxelmap(fvh_vgicp* h) { CHECK_HANDLE(h); return h->build_map(h->resolution); }
int fvh_vgicp_set_voxel_accumulation_mode(fvh_vgicp* h, int mode) {
  CHECK_HANDLE(h);
  if (mode < 0 || mode > 2) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "unknown voxel accumulation mode");
  if (mode != h->voxel_mode) { h->voxelmap.invalidate(); h->e.has_corr = false; }
  h->voxel_mode = mode;
  return FVH_OK;
}
// multi-GPU (with peers attached or a communicator): keep only this rank's shard of the target voxel map (see fvh_vgicp_align)
int fvh_vgicp_set_target_map_sharding(fvh_vgicp* h, int on, int margin_voxels) {
  CHECK_HANDLE(h);
  if (margin_voxels < 0 || margin_voxels > 64) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "set_target_map_sharding: margin must be within [0, 64] voxels");
  if ((on != 0) != h->shard_map || margin_voxels != h->shard_margin) { h->voxelmap.invalidate(); h->e.has_corr = false; }
  h->shard_map = on != 0;
  h->shard_margin = margin_voxels;
  return FVH_OK;
}
int fvh_vgicp_debug_get_live_map_voxels(fvh_vgicp* h, int* n) {  // voxel count of the map the last align used, as it is (a shard stays a shard: no rebuild)
  CHECK_HANDLE(h);
  if (!n) return FVH_ERR_INVALID_ARGUMENT;
  if (!h->voxelmap.valid) return h->e.fail(FVH_ERR_BAD_STATE, "voxel map not built");
  HIP_OR_FAIL(&h->e, hipMemcpyAsync(n, h->voxelmap.counters_cur(), sizeof(int), hipMemcpyDeviceToHost, h->e.stream));
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  return FVH_OK;
}
int fvh_vgicp_debug_get_map_shard(fvh_vgicp* h, int* is_shard, int* fallbacks) {
  CHECK_HANDLE(h);
  if (is_shard) *is_shard = h->voxelmap.valid && h->voxelmap.is_shard ? 1 : 0;
  if (fallbacks) *fallbacks = h->shard_fallbacks;
  return FVH_OK;
}
int fvh_vgicp_debug_get_spatial_order(fvh_vgicp* h, int which, int* order, float* tile_boxes) {
  CHECK_HANDLE(h);
  Engine* e = &h->e;
  CloudDev& c = which ? h->target : h->source;
  if (!c.has_pts) return e->fail(FVH_ERR_BAD_STATE, "debug_get_spatial_order: cloud not set");
  int rc = ensure_sorted(e, c);
  if (rc) return rc;
  if (c.n == 0) return FVH_OK;
  if (order) HIP_OR_FAIL(e, hipMemcpyAsync(order, c.order.p, sizeof(int) * (size_t)c.n, hipMemcpyDeviceToHost, e->stream));
  if (tile_boxes) HIP_OR_FAIL(e, hipMemcpyAsync(tile_boxes, c.bbox.p, sizeof(float4) * 2 * (size_t)((c.n + 63) / 64), hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  return FVH_OK;
}
int fvh_vgicp_swap_source_and_target(fvh_vgicp* h) {
  CHECK_HANDLE(h);
  h->source.swap(h->target);
  h->e.has_corr = false;
  if (!h->target.has_pts || !h->target.has_cov) { h->voxelmap.invalidate(); return FVH_OK; }  // fast_vgicp_cuda.cu:102-104
  // the stream had drained (the caller holds the result of an align): the new target's points and covariances are complete, and
  // the build can run on the side stream while the caller prepares the next source cloud on the main one
  if (h->e.was_quiet && h->e.side_stream()) {
    const double res = h->resolution;
    h->voxelmap.invalidate();
    h->e.deferred = [h, res](hipStream_t on_side) { return h->build_map(res, false, on_side); };
    return FVH_OK;
  }
  return h->build_map(h->resolution);
}
int fvh_vgicp_gicp_swap_source_and_target(fvh_vgicp* h) {  // FastGICP::swapSourceAndTarget (fast_gicp_impl.hpp:56-62): no voxel map to rebuild
  CHECK_HANDLE(h);
  h->source.swap(h->target);
  h->e.has_corr = false;
  h->voxelmap.invalidate();
  return FVH_OK;
}
static void cloud_replaced(CloudDev& c) { c.has_cov = false; c.has_nbr = false; c.has_cov_sorted = false; }
// The Morton order of a VGICP cloud is queued right behind its upload: every neighbour search needs it (and large clouds walk the LM
// loop in it), and the caller's next call -- find_*_neighbors as a rule -- would launch it a few microseconds of host time later
// with the stream idle in between (kernel trace: 4 us between the pack kernel and the sort).
static int uploaded(Engine* e, CloudDev& c, int rc) {
  if (rc || c.n == 0) return rc;
  if (!e->device_search_seen && !e->sharded() && c.n < COHERENT_MIN_POINTS) return FVH_OK;  // nobody may ever need the order: it stays lazy (ensure_sorted where it is consumed)
  return ensure_sorted(e, c);
}
int fvh_vgicp_set_source_cloud(fvh_vgicp* h, const float* xyz, int n) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, xyz, n, 3, false)); }
int fvh_vgicp_set_target_cloud(fvh_vgicp* h, const float* xyz, int n) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, xyz, n, 3, false)); }
int fvh_vgicp_set_source_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, xyz, n, stride, false)); }
int fvh_vgicp_set_target_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, xyz, n, stride, false)); }
int fvh_vgicp_set_source_cloud_device(fvh_vgicp* h, const float* d, int n, int stride) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, d, n, stride, true)); }
int fvh_vgicp_set_target_cloud_device(fvh_vgicp* h, const float* d, int n, int stride) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, d, n, stride, true)); }
int fvh_vgicp_set_source_neighbors(fvh_vgicp* h, int k, const int* idx) { CHECK_HANDLE(h); return set_neighbors(&h->e, h->source, k, idx); }
int fvh_vgicp_set_target_neighbors(fvh_vgicp* h, int k, const int* idx) { CHECK_HANDLE(h); return set_neighbors(&h->e, h->target, k, idx); }
int fvh_vgicp_find_source_neighbors(fvh_vgicp* h, int k) { CHECK_HANDLE_SOURCE_CHAIN(h); return h->e.after_source_chain_call(find_neighbors(&h->e, h->source, k)); }
int fvh_vgicp_find_target_neighbors(fvh_vgicp* h, int k) { CHECK_HANDLE(h); return find_neighbors(&h->e, h->target, k); }
int fvh_vgicp_calculate_source_covariances(fvh_vgicp* h, int m) { CHECK_HANDLE_SOURCE_CHAIN(h); return h->e.after_source_chain_call(calc_cov_knn(&h->e, h->source, m)); }
int fvh_vgicp_calculate_target_covariances(fvh_vgicp* h, int m) { CHECK_HANDLE(h); return calc_cov_knn(&h->e, h->target, m); }
int fvh_vgicp_calculate_source_covariances_rbf(fvh_vgicp* h, int m) { CHECK_HANDLE_SOURCE_CHAIN(h); return h->e.after_source_chain_call(calc_cov_rbf(&h->e, h->source, h->kernel_width, h->kernel_max_dist, m)); }
int fvh_vgicp_calculate_target_covariances_rbf(fvh_vgicp* h, int m) { CHECK_HANDLE(h); return calc_cov_rbf(&h->e, h->target, h->kernel_width, h->kernel_max_dist, m); }
int fvh_vgicp_set_source_covariances(fvh_vgicp* h, const double* c) { CHECK_HANDLE(h); return set_cov_host(&h->e, h->source, c); }
int fvh_vgicp_set_target_covariances(fvh_vgicp* h, const double* c) { CHECK_HANDLE(h); return set_cov_host(&h->e, h->target, c); }

int fvh_vgicp_get_num_source_points(const fvh_vgicp* h, int* n) { if (!h || !n) return FVH_ERR_INVALID_ARGUMENT; *n = h->source.has_pts ? h->source.n : 0; return FVH_OK; }
int fvh_vgicp_get_num_target_points(const fvh_vgicp* h, int* n) { if (!h || !n) return FVH_ERR_INVALID_ARGUMENT; *n = h->target.has_pts ? h->target.n : 0; return FVH_OK; }
int fvh_vgicp_get_source_neighbors(fvh_vgicp* h, int* k, int* out) { CHECK_HANDLE(h); return get_nbr_host(&h->e, h->source, k, out); }
int fvh_vgicp_get_target_neighbors(fvh_vgicp* h, int* k, int* out) { CHECK_HANDLE(h); return get_nbr_host(&h->e, h->target, k, out); }
int fvh_vgicp_get_source_covariances(fvh_vgicp* h, float* c) { CHECK_HANDLE(h); return get_cov_host(&h->e, h->source, c); }
int fvh_vgicp_get_target_covariances(fvh_vgicp* h, float* c) { CHECK_HANDLE(h); return get_cov_host(&h->e, h->target, c); }
int fvh_vgicp_get_num_voxels(fvh_vgicp* h, int* n) { CHECK_HANDLE(h); { int _w = h->whole_map(); if (_w) return _w; } if (!n) return FVH_ERR_INVALID_ARGUMENT; const Rebuild rb = h->rebuild_safe(); int rc = fetch_voxelmap_host(&h->e, h->voxelmap, &rb); if (rc) return rc; *n = (int)h->voxelmap.h_occupied.size(); return FVH_OK; }
int fvh_vgicp_get_voxel_num_points(fvh_vgicp* h, int* o) { CHECK_HANDLE(h); { int _w = h->whole_map(); if (_w) return _w; } const Rebuild rb = h->rebuild_safe(); return get_voxels_host(&h->e, h->voxelmap, nullptr, o, nullptr, nullptr, &rb); }
int fvh_vgicp_get_voxel_means(fvh_vgicp* h, float* o) { CHECK_HANDLE(h); { int _w = h->whole_map(); if (_w) return _w; } const Rebuild rb = h->rebuild_safe(); return get_voxels_host(&h->e, h->voxelmap, nullptr, nullptr, o, nullptr, &rb); }
int fvh_vgicp_get_voxel_covs(fvh_vgicp* h, float* o) { CHECK_HANDLE(h); { int _w = h->whole_map(); if (_w) return _w; } const Rebuild rb = h->rebuild_safe(); return get_voxels_host(&h->e, h->voxelmap, nullptr, nullptr, nullptr, o, &rb); }
int fvh_vgicp_get_voxel_coords(fvh_vgicp* h, int* o) { CHECK_HANDLE(h); { int _w = h->whole_map(); if (_w) return _w; } const Rebuild rb = h->rebuild_safe(); return get_voxels_host(&h->e, h->voxelmap, o, nullptr, nullptr, nullptr, &rb); }

// row of source point i in the stored correspondence buffer (Engine::corr_by_position: its position in the cloud's Morton order)
static int fetch_row_of_point(Engine* e, const CloudDev& c, int n_src, std::vector<int>& row_of) {
  row_of.resize((size_t)n_src);
  if (!e->corr_by_position) { for (int i = 0; i < n_src; i++) row_of[(size_t)i] = i; return FVH_OK; }
  if (!c.has_sorted || c.n != n_src) return e->fail(FVH_ERR_BAD_STATE, "correspondence getters: the spatial order the stored list was found in is gone; call update_correspondences again");
  std::vector<int> order((size_t)n_src);
  if (n_src) {
    HIP_OR_FAIL(e, hipMemcpyAsync(order.data(), c.order.p, sizeof(int) * (size_t)n_src, hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  }
  for (int j = 0; j < n_src; j++) row_of[(size_t)order[(size_t)j]] = j;
  return FVH_OK;
}
static int fetch_corr(Engine* e, int n_src, std::vector<int>& corr) {
  if (!e->has_corr) return e->fail(FVH_ERR_BAD_STATE, "no correspondences: call update_correspondences first");
  corr.resize((size_t)n_src * (e->corr_kind == 1 ? 1 : e->n_off));
  if (corr.empty()) return FVH_OK;
  HIP_OR_FAIL(e, hipMemcpyAsync(corr.data(), e->corr.as<int>() + (size_t)e->corr_sel * corr.size(), sizeof(int) * corr.size(), hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  return FVH_OK;
}
int fvh_vgicp_get_num_correspondences(fvh_vgicp* h, int* n) {
  CHECK_HANDLE(h);
  if (!n) return FVH_ERR_INVALID_ARGUMENT;
  std::vector<int> corr;
  int rc = fetch_corr(&h->e, h->e.corr_n_src, corr);
  if (rc) return rc;
  int c = 0;
  if (h->e.sharded() && h->e.corr_kind == 0 && h->source.has_sorted && h->e.corr_n_src == h->source.n) {
    // multi-GPU: only this rank's tile of the source was evaluated (the other rows of the buffer were never written)
    const Tile t = peer_tile(&h->e, h->source.n);
    std::vector<int> order((size_t)std::max(t.hi - t.lo, 0));
    if (!order.empty()) {
      HIP_OR_FAIL(&h->e, hipMemcpyAsync(order.data(), h->source.order.as<int>() + t.lo, sizeof(int) * order.size(), hipMemcpyDeviceToHost, h->e.stream));
      HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
    }
    if (h->e.corr_by_position) { for (int j = t.lo; j < t.hi; j++) for (int o = 0; o < h->e.n_off; o++) c += (corr[(size_t)j * h->e.n_off + o] >= 0); }  // (rows by position: the tile IS a range of rows)
    else for (int i : order) for (int o = 0; o < h->e.n_off; o++) c += (corr[(size_t)i * h->e.n_off + o] >= 0);
  } else {
    for (int v : corr) c += (v >= 0);
  }
  *n = c;
  return FVH_OK;
}
// offset-major then source index, invalid pairs removed: the order of find_voxel_correspondences.cu:93-110
int fvh_vgicp_get_voxel_correspondences(fvh_vgicp* h, int* pairs) {
  CHECK_HANDLE(h);
  if (!pairs) return FVH_ERR_INVALID_ARGUMENT;
  // the map first: if its hint-sized table overflowed it is rebuilt here, which drops the correspondences (BAD_STATE below)
  const Rebuild rb = h->rebuild_safe();
  int rc = fetch_voxelmap_host(&h->e, h->voxelmap, &rb);
  if (rc) return rc;
  std::vector<int> corr;
  rc = fetch_corr(&h->e, h->e.corr_n_src, corr);
  if (rc) return rc;
  std::vector<int> row_of;
  rc = fetch_row_of_point(&h->e, h->source, h->e.corr_n_src, row_of);
  if (rc) return rc;
  size_t w = 0;
  for (int o = 0; o < h->e.n_off; o++)
    for (int i = 0; i < h->e.corr_n_src; i++) {
      int b = corr[(size_t)row_of[(size_t)i] * h->e.n_off + o];
      if (b < 0) continue;
      pairs[2 * w] = i;
      pairs[2 * w + 1] = h->voxelmap.bucket_to_index[b];
      w++;
    }
  return FVH_OK;
}

int fvh_vgicp_update_correspondences(fvh_vgicp* h, const double* T) {
  CHECK_HANDLE(h);
  if (!h->source.has_pts || !h->source.has_cov) return h->e.fail(FVH_ERR_BAD_STATE, "update_correspondences: source cloud/covariances not set");
  if (h->e.sharded() && h->source.n) { int rc = ensure_sorted(&h->e, h->source); if (rc) return rc; }
  h->e.corr_kind = 0;
  { int rc = h->whole_map(); if (rc) return rc; }  // host-driven evaluations use the whole map (a sharded align left its shard behind)
  return do_update_correspondences<MODE_VGICP>(&h->e, h->cost_source(), h->voxelmap, T);
}
// ---- FastGICP (nearest target point) on the same handle: fast_gicp_impl.hpp:118-240 ----
int fvh_vgicp_gicp_set_max_correspondence_distance(fvh_vgicp* h, double d) {
  CHECK_HANDLE(h);
  if (!(d > 0)) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "max correspondence distance must be > 0");
  h->gicp_max_dist = d;
  return FVH_OK;
}
int fvh_vgicp_gicp_update_correspondences(fvh_vgicp* h, const double* T) {
  CHECK_HANDLE(h);
  return gicp_update_correspondences(&h->e, h->source, h->target, h->gicp_records, T, h->gicp_max_dist);
}
int fvh_vgicp_gicp_align(fvh_vgicp* h, const double* guess, const fvh_lm_params* p, fvh_lm_result* r) {
  CHECK_HANDLE(h);
  const int rc = gicp_align(&h->e, h->source, h->target, h->gicp_records, h->gicp_cost_source(true), h->gicp_max_dist, guess, p, r);
  if (rc == FVH_OK) h->e.gang_clear();  // (the result came back through a stream synchronisation)
  return rc;
}
int fvh_vgicp_gicp_compute_error(fvh_vgicp* h, const double* T, double* H, double* b, double* err) {
  CHECK_HANDLE(h);
  if (!h->e.has_corr || h->e.corr_kind != 1) return h->e.fail(FVH_ERR_BAD_STATE, "gicp_compute_error: call gicp_update_correspondences first");
  return do_compute_error<MODE_VGICP>(&h->e, h->gicp_cost_source(), h->gicp_records, T, H, b, err, [] { return (int)FVH_OK; });
}
int fvh_vgicp_gicp_get_correspondences(fvh_vgicp* h, int* target_index_per_source_point) {
  CHECK_HANDLE(h);
  if (!target_index_per_source_point) return FVH_ERR_INVALID_ARGUMENT;
  if (!h->e.has_corr || h->e.corr_kind != 1) return h->e.fail(FVH_ERR_BAD_STATE, "gicp_get_correspondences: call gicp_update_correspondences first");
  std::vector<int> corr;
  int rc = fetch_corr(&h->e, h->e.corr_n_src, corr);
  if (rc) return rc;
  std::memcpy(target_index_per_source_point, corr.data(), sizeof(int) * corr.size());
  return FVH_OK;
}
int fvh_vgicp_compute_error(fvh_vgicp* h, const double* T, double* H, double* b, double* err) {
  CHECK_HANDLE(h);
  if (h->e.has_corr && h->e.corr_kind != 0) return h->e.fail(FVH_ERR_BAD_STATE, "compute_error: the stored correspondences are nearest-point (GICP) ones; call update_correspondences first");
  { int rc = h->whole_map(); if (rc) return rc; }
  return do_compute_error<MODE_VGICP>(&h->e, h->cost_source(), h->voxelmap, T, H, b, err, h->rebuild_safe());
}
int fvh_vgicp_align(fvh_vgicp* h, const double* guess, const fvh_lm_params* p, fvh_lm_result* r) {
  CHECK_HANDLE(h);
  if (!h->source.has_pts || !h->source.has_cov) return h->e.fail(FVH_ERR_BAD_STATE, "align: source cloud/covariances not set");
  if (h->e.sharded() && h->source.n) { int rc = ensure_sorted(&h->e, h->source); if (rc) return rc; }
  if (h->shard_map && h->e.sharded() && h->e.shard_ranks() > 1 && guess) {
    // Sharded target map: this rank's map holds the voxels around T_guess * (its tile of the source) only -- the tile's bounding box in
    // voxel coordinates, widened by the reach of the neighbour offsets + shard_margin voxels of pose motion. Built per align (it depends
    // on the guess); if a source element leaves the inner box during the LM loop the align is redone on the full map.
    Engine* e = &h->e;
    if (!h->target.has_pts || !h->target.has_cov) return e->fail(FVH_ERR_BAD_STATE, "align: target cloud / covariances not set");
    HIP_OR_FAIL(e, h->voxelmap.region.ensure(sizeof(VmRegion)));
    int reach = 0;
    for (int v : e->offsets_host) reach = std::max(reach, std::abs(v));
    const Tile t = peer_tile(e, h->source.n);
    vm_region_kernel<<<1, 1024, 0, e->stream>>>(h->source.sorted.as<float4>(), t.lo, t.hi, pose_from_colmajor16(guess), h->resolution, reach + h->shard_margin, reach, h->voxelmap.region.as<VmRegion>());
    HIP_OR_FAIL(e, hipGetLastError());
    int rc = h->build_map(h->resolution, false, nullptr, true);
    if (rc) return rc;
    rc = do_align<MODE_VGICP>(e, h->cost_source(), h->voxelmap, guess, p, r, h->rebuild_safe());
    if (rc) return rc;
    // (the flag travelled with the sums of every evaluation: all ranks hold the same value and fall back together -- the redo is a collective align like any other)
    if (!reinterpret_cast<const LmState*>(e->pinned)->halo_exceeded) { h->e.quiet = true; return FVH_OK; }
    h->shard_fallbacks++;
    rc = h->build_map(h->resolution);
    if (rc) return rc;
  }
  const int rc = do_align<MODE_VGICP>(&h->e, h->cost_source(), h->voxelmap, guess, p, r, h->rebuild_safe());
  // the result is on the host: everything this handle queued has run (of a persistent launch only the workgroups' exit remains,
  // and they touch neither clouds nor the map any more)
  if (rc == FVH_OK) h->e.quiet = true;
  return rc;
}
static int get_lm_trace(Engine* e, int* n, double* rows6) {
  if (!n) return e->fail(FVH_ERR_INVALID_ARGUMENT, "get_lm_trace: null count");
  *n = e->lm_trace_rows;
  if (rows6 && e->lm_trace_rows > 0) {
    HIP_OR_FAIL(e, hipMemcpyAsync(rows6, e->lm_trace.p, sizeof(double) * 6 * (size_t)e->lm_trace_rows, hipMemcpyDeviceToHost, e->stream));
    HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  }
  return FVH_OK;
}
int fvh_vgicp_set_lm_trace(fvh_vgicp* h, int on) { CHECK_HANDLE(h); h->e.lm_trace_on = on != 0; return FVH_OK; }
int fvh_vgicp_get_lm_trace(fvh_vgicp* h, int* n, double* rows6) { CHECK_HANDLE(h); return get_lm_trace(&h->e, n, rows6); }
int fvh_vgicp_fitness_score(fvh_vgicp* h, const double* T, double max_range, double* score) { CHECK_HANDLE(h); return do_fitness(&h->e, h->source, h->target, T, max_range, score); }
int fvh_vgicp_profile_enable(fvh_vgicp* h, int on) { CHECK_HANDLE_HOST_ONLY(h); h->e.prof.on = on != 0; h->e.prof.cost_only = on == 2; return FVH_OK; }
int fvh_vgicp_profile_reset(fvh_vgicp* h) { CHECK_HANDLE(h); HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream)); h->e.prof.reset(); return FVH_OK; }
int fvh_vgicp_profile_get(fvh_vgicp* h, const char* cls, double* ms, int* n) { CHECK_HANDLE(h); return profile_get(&h->e, cls, ms, n); }
int fvh_vgicp_debug_set_voxel_hint(fvh_vgicp* h, int num_voxels) { CHECK_HANDLE(h); h->voxelmap.nv_hint = num_voxels; return FVH_OK; }
int fvh_vgicp_debug_get_persist_aborts(fvh_vgicp* h, int* n) { CHECK_HANDLE(h); if (!n) return FVH_ERR_INVALID_ARGUMENT; *n = h->e.persist_aborts; return FVH_OK; }
int fvh_vgicp_debug_get_persist_grid(fvh_vgicp* h, int* blocks, int* capacity) {
  CHECK_HANDLE(h);
  if (blocks) *blocks = h->e.last_persist_blocks;
  if (capacity) *capacity = persistent_capacity<MODE_VGICP>(&h->e);
  return FVH_OK;
}
int fvh_debug_sort_routes(int* counts4) {
  if (!counts4) return FVH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < 4; i++) counts4[i] = g_sort_routes[i].load();
  return FVH_OK;
}
int fvh_debug_xcd_local(int* wanted, int* placement_aborts) {
  if (wanted) *wanted = xcd_local_wanted() ? 1 : 0;
  if (placement_aborts) *placement_aborts = g_xcd_local_strikes.load();
  return FVH_OK;
}
int fvh_debug_slot_pool(int device, int* reserved, int* active, int* recent) {
  int r = 0, a = 0, c = 0;
  g_slots.snapshot(device, &r, &a, &c);
  if (reserved) *reserved = r;
  if (active) *active = a;
  if (recent) *recent = c;
  return FVH_OK;
}
int fvh_vgicp_debug_get_table_capacity(fvh_vgicp* h, int* capacity) { CHECK_HANDLE(h); if (!capacity) return FVH_ERR_INVALID_ARGUMENT; *capacity = (int)h->voxelmap.capacity; return FVH_OK; }
int fvh_vgicp_debug_get_skipped_points(fvh_vgicp* h, int* n) {
  CHECK_HANDLE(h);
  if (!n) return FVH_ERR_INVALID_ARGUMENT;
  if (!h->voxelmap.valid) return h->e.fail(FVH_ERR_BAD_STATE, "voxel map not built");
  HIP_OR_FAIL(&h->e, hipMemcpyAsync(n, h->voxelmap.counters_cur() + 2, sizeof(int), hipMemcpyDeviceToHost, h->e.stream));
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  return FVH_OK;
}
int fvh_vgicp_synchronize(fvh_vgicp* h) { CHECK_HANDLE(h); HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream)); h->e.gang_clear(); return FVH_OK; }

#ifdef FVH_KNN_TIMING
int fvh_debug_pair_counts(unsigned long long* out8, int reset) {  // candidate points / box tests the culled k-NN and RBF searches evaluated since the last reset
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pair_counts), sizeof(unsigned long long) * 8) != hipSuccess) return FVH_ERR_HIP;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pair_counts), z, sizeof(z)) != hipSuccess) return FVH_ERR_HIP; }
  return FVH_OK;
}
int fvh_debug_knn_timing(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_time), sizeof(unsigned long long) * 32768 * 8) == hipSuccess ? FVH_OK : FVH_ERR_HIP; }
#endif
#ifdef FVH_SORT_TIMING
int fvh_debug_sort_timing(unsigned long long* out /* [COOP_WGS][16] */) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_time), sizeof(unsigned long long) * COOP_WGS * 16) == hipSuccess ? FVH_OK : FVH_ERR_HIP; }
#endif
#ifdef FVH_COST_TIMING
// debug build only (not declared in the public header): out[0] = earliest workgroup start, out[1..7] = epilogue stamps of the last workgroup, 100 MHz ticks
int fvh_debug_lm_timing(unsigned long long* out /* [16][8] shader-cycle stamps of the last 16 LM steps since the reset */, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lmtime), sizeof(unsigned long long) * 16 * 8) != hipSuccess) return FVH_ERR_HIP;
  if (reset) {
    unsigned zero = 0;
    void* p = nullptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lmcall), &zero, sizeof(zero)) != hipSuccess) return FVH_ERR_HIP;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_lmtime)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 16 * 8) != hipSuccess) return FVH_ERR_HIP;
  }
  return FVH_OK;
}
int fvh_debug_persist_timing(unsigned long long* out, int reset) {  // out: [16][512][12]
  const size_t bytes = sizeof(unsigned long long) * 16 * 512 * 12;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ptime), bytes) != hipSuccess) return FVH_ERR_HIP;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ptime)) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return FVH_ERR_HIP;
  }
  return FVH_OK;
}
int fvh_debug_main_timing(unsigned long long* out, int reset) {  // out: [16][512][12]
  const size_t bytes = sizeof(unsigned long long) * 16 * 512 * 12;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mtime), bytes) != hipSuccess) return FVH_ERR_HIP;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_mtime)) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return FVH_ERR_HIP;
  }
  return FVH_OK;
}
int fvh_debug_cost_timing(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cost_timing), sizeof(unsigned long long) * 16) != hipSuccess) return FVH_ERR_HIP;
  if (reset) { unsigned long long init[16]; for (auto& v : init) v = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cost_timing), init, sizeof(init)) != hipSuccess) return FVH_ERR_HIP; }
  return FVH_OK;
}
#endif
int fvh_comm_unique_id(void* id128) {
  if (!id128) return FVH_ERR_INVALID_ARGUMENT;
  if (!g_rccl.load()) return FVH_ERR_COMM;
  return g_rccl.GetUniqueId(id128) == 0 ? FVH_OK : FVH_ERR_COMM;
}
int fvh_vgicp_comm_init(fvh_vgicp* h, const void* id, int nranks, int rank) { CHECK_HANDLE(h); return comm_init(&h->e, id, nranks, rank); }
// ---- peer-mapped exchange (kernels_peer.hpp) ----
int fvh_vgicp_peer_export(fvh_vgicp* h, int max_points, void* ipc_handle64, unsigned long long* process_local_ptr) {
  CHECK_HANDLE(h);
  Engine* e = &h->e;
  if (max_points <= 0 || !ipc_handle64) return e->fail(FVH_ERR_INVALID_ARGUMENT, "peer_export: bad arguments");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI hands the IPC handle over as 64 opaque bytes");
  if (e->peer.attached()) return e->fail(FVH_ERR_BAD_STATE, "peer_export: detach first");
  const size_t half = (((size_t)max_points + 63) & ~(size_t)63) * 32;  // two float4 per point; a single-rank "tile" is the whole cloud
  const size_t bytes = PEER_STAGE_OFFSET + 2 * half;
  if (!e->peer.region || e->peer.region_bytes < bytes) {
    if (e->peer.region) { HIP_OR_FAIL(e, hipFree(e->peer.region)); e->peer.region = nullptr; }
    HIP_OR_FAIL(e, hipExtMallocWithFlags(reinterpret_cast<void**>(&e->peer.region), bytes, hipDeviceMallocFinegrained));
    e->peer.region_bytes = bytes;
  }
  e->peer.stage_half_bytes = half;
  HIP_OR_FAIL(e, hipMemsetAsync(e->peer.region, 0, PEER_STAGE_OFFSET, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  e->peer.x = 2; e->peer.stage_gen = 0;
  hipIpcMemHandle_t hm;
  HIP_OR_FAIL(e, hipIpcGetMemHandle(&hm, e->peer.region));
  std::memcpy(ipc_handle64, &hm, 64);
  if (process_local_ptr) *process_local_ptr = (unsigned long long)(uintptr_t)e->peer.region;
  return FVH_OK;
}
int fvh_vgicp_peer_attach(fvh_vgicp* h, int nranks, int rank, int ranks_on_this_device, const void* ipc_handles, const unsigned long long* process_local_ptrs) {
  CHECK_HANDLE(h);
  Engine* e = &h->e;
  if (nranks < 1 || nranks > FVH_MAX_PEERS || rank < 0 || rank >= nranks || ranks_on_this_device < 1 || (!ipc_handles && !process_local_ptrs))
    return e->fail(FVH_ERR_INVALID_ARGUMENT, "peer_attach: bad arguments (at most 8 ranks)");
  if (!e->peer.region) return e->fail(FVH_ERR_BAD_STATE, "peer_attach: call peer_export first");
  if (e->comm) return e->fail(FVH_ERR_BAD_STATE, "peer_attach: an RCCL communicator is attached (use one or the other)");
  e->peer_detach();
  for (int p = 0; p < nranks; p++) {
    if (p == rank) { e->peer.peer_region[p] = e->peer.region; continue; }
    if (process_local_ptrs && process_local_ptrs[p]) { e->peer.peer_region[p] = reinterpret_cast<char*>((uintptr_t)process_local_ptrs[p]); continue; }  // a handle of THIS process
    if (!ipc_handles) { e->peer_detach(); return e->fail(FVH_ERR_INVALID_ARGUMENT, "peer_attach: no IPC handle for a peer of another process"); }
    hipIpcMemHandle_t hm;
    std::memcpy(&hm, static_cast<const char*>(ipc_handles) + 64 * (size_t)p, 64);
    void* ptr = nullptr;
    hipError_t rc = hipIpcOpenMemHandle(&ptr, hm, hipIpcMemLazyEnablePeerAccess);
    if (rc != hipSuccess) { e->peer_detach(); return e->hipfail(rc, "hipIpcOpenMemHandle (peer exchange region)"); }
    e->peer.peer_region[p] = static_cast<char*>(ptr);
    e->peer.ipc_opened[p] = true;
  }
  // where does every region live, and can this device reach it? (a clear error now instead of a 2 s exchange time-out later)
  for (int p = 0; p < nranks; p++) {
    e->peer.peer_device[p] = -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, e->peer.peer_region[p]) == hipSuccess) e->peer.peer_device[p] = attr.device;
    else (void)hipGetLastError();
    const int pd = e->peer.peer_device[p];
    if (pd >= 0 && pd != e->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, e->device, pd) == hipSuccess && !can) {
        e->peer_detach();
        return e->fail(FVH_ERR_COMM, "peer_attach: device " + std::to_string(e->device) + " cannot access device " + std::to_string(pd) + " (rank " + std::to_string(p) +
                                       "): no peer-to-peer path between them (hipDeviceCanAccessPeer); use the RCCL route (fvh_vgicp_comm_init)");
      }
    }
  }
  e->peer.n = nranks; e->peer.rank = rank; e->peer.ranks_on_device = ranks_on_this_device;
  e->has_corr = false;
  return FVH_OK;
}
// Collective (every rank calls it, in the same order with respect to other collective calls): every rank stores a nonce into every
// region and waits for everybody's nonce in its own -- the path the in-kernel mailboxes take, checked once up front.
int fvh_vgicp_peer_selfcheck(fvh_vgicp* h, double timeout_seconds, int* missing_rank_mask) {
  CHECK_HANDLE(h);
  Engine* e = &h->e;
  if (missing_rank_mask) *missing_rank_mask = 0;
  if (!e->peer.attached()) return e->fail(FVH_ERR_BAD_STATE, "peer_selfcheck: no peers attached");
  if (!(timeout_seconds > 0)) timeout_seconds = 5.0;
  const unsigned long long nonce = 0x5e1fc4ec00000000ull + (++e->peer.check_gen);
  const PeerView pv = e->peer.view(0);
  HIP_OR_FAIL(e, e->peer.err.ensure(64));
  HIP_OR_FAIL(e, hipMemsetAsync(e->peer.err.p, 0, 4, e->stream));
  peer_check_write_kernel<<<1, 64, 0, e->stream>>>(pv, nonce);
  peer_check_wait_kernel<<<1, 64, 0, e->stream>>>(pv, nonce, (unsigned long long)(timeout_seconds * 1e8), e->peer.err.as<unsigned>());
  HIP_OR_FAIL(e, hipGetLastError());
  unsigned* h_missing = reinterpret_cast<unsigned*>(e->pinned);
  HIP_OR_FAIL(e, hipMemcpyAsync(h_missing, e->peer.err.p, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_OR_FAIL(e, hipStreamSynchronize(e->stream));
  if (*h_missing) {
    if (missing_rank_mask) *missing_rank_mask = (int)*h_missing;
    std::string who;
    for (int p = 0; p < e->peer.n; p++)
      if (*h_missing & (1u << p)) who += (who.empty() ? "" : ", ") + std::to_string(p) + " (device " + std::to_string(e->peer.peer_device[p]) + ")";
    return e->fail(FVH_ERR_COMM, "peer_selfcheck: rank " + std::to_string(e->peer.rank) + " (device " + std::to_string(e->device) + ") never saw the store of rank(s) " + who +
                                   " within " + std::to_string(timeout_seconds) + " s -- that rank did not call the self-check, or its stores do not reach this device's memory "
                                   "(IPC mapping / xGMI peer access); use the RCCL route (fvh_vgicp_comm_init)");
  }
  return FVH_OK;
}
int fvh_vgicp_peer_detach(fvh_vgicp* h) { CHECK_HANDLE(h); h->e.peer_detach(); h->e.has_corr = false; return FVH_OK; }
int fvh_vgicp_comm_destroy(fvh_vgicp* h) { CHECK_HANDLE(h); if (h->e.comm) { g_rccl.CommDestroy(h->e.comm); h->e.comm = nullptr; } h->e.nranks = 1; h->e.rank = 0; return FVH_OK; }

// ---- NDT ---------------------------------------------------------------------------------------
// between fvh_ndt_align_async and fvh_ndt_align_wait the handle's clouds, maps and state belong to the running LM kernel
#define NDT_NOT_PENDING(h) if ((h)->pending.active) return (h)->e.fail(FVH_ERR_BAD_STATE, "an align_async is in flight: call fvh_ndt_align_wait first");
int fvh_ndt_create(int device, fvh_ndt** out) {
  if (!out) return FVH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  fvh_ndt* h = new (std::nothrow) fvh_ndt();
  if (!h) return FVH_ERR_HIP;
  int rc = h->e.init(device);
  if (!rc) rc = h->e.set_offsets(FVH_DIRECT7, 0.0);  // ndt_cuda.cu:22
  if (rc) { std::fprintf(stderr, "fvh_ndt_create: %s\n", h->e.err.c_str()); h->e.shutdown(); delete h; return rc; }
  *out = h;
  return FVH_OK;
}
int fvh_ndt_destroy(fvh_ndt* h) {
  if (!h) return FVH_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(h->e.device);
  if (h->e.stream) (void)hipStreamSynchronize(h->e.stream);
  if (h->e.side) (void)hipStreamSynchronize(h->e.side);
  if (h->pending.active) { h->pending.release_slots(); h->pending.active = false; }
  h->source.release(); h->target.release(); h->source_vm.release(); h->target_vm.release(); h->next_source.release(); h->next_vm.release();
  if (h->prep_done) (void)hipEventDestroy(h->prep_done);
  h->e.shutdown();
  delete h;
  return FVH_OK;
}
const char* fvh_ndt_last_error(const fvh_ndt* h) { return h ? h->e.err.c_str() : "null handle"; }
int fvh_ndt_set_distance_mode(fvh_ndt* h, int m) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); if (m != FVH_NDT_P2D && m != FVH_NDT_D2D) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "bad distance mode"); h->distance_mode = m; h->e.has_corr = false; return FVH_OK; }
int fvh_ndt_set_resolution(fvh_ndt* h, double r) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); if (!(r > 0)) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "resolution must be > 0"); h->resolution = r; return FVH_OK; }
int fvh_ndt_set_neighbor_search_method(fvh_ndt* h, int m, double radius) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); return h->e.set_offsets(m, radius); }
int fvh_ndt_set_precision(fvh_ndt* h, int p) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (p != FVH_COMPUTE_FP64 && p != FVH_COMPUTE_FP32 && p != FVH_COMPUTE_CUDA_COMPAT) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "bad precision");
  if ((p == FVH_COMPUTE_CUDA_COMPAT) != (h->e.precision == FVH_COMPUTE_CUDA_COMPAT)) { h->source_vm.invalidate(); h->target_vm.invalidate(); h->e.has_corr = false; }  // the voxel records of the two arithmetics differ
  h->e.precision = p;
  return FVH_OK;
}
int fvh_ndt_swap_source_and_target(fvh_ndt* h) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  h->source.swap(h->target);
  std::swap(h->source_vm, h->target_vm);
  h->e.has_corr = false;
  return FVH_OK;
}
int fvh_ndt_set_source_cloud(fvh_ndt* h, const float* xyz, int n) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->source_vm.invalidate(); return upload_cloud(&h->e, h->source, xyz, n, 3, false, false); }
int fvh_ndt_set_target_cloud(fvh_ndt* h, const float* xyz, int n) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->target_vm.invalidate(); return upload_cloud(&h->e, h->target, xyz, n, 3, false, false); }
int fvh_ndt_set_source_cloud_strided(fvh_ndt* h, const float* xyz, int n, int s) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->source_vm.invalidate(); return upload_cloud(&h->e, h->source, xyz, n, s, false, false); }
int fvh_ndt_set_target_cloud_strided(fvh_ndt* h, const float* xyz, int n, int s) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->target_vm.invalidate(); return upload_cloud(&h->e, h->target, xyz, n, s, false, false); }
int fvh_ndt_set_source_cloud_device(fvh_ndt* h, const float* d, int n, int s) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->source_vm.invalidate(); return upload_cloud(&h->e, h->source, d, n, s, true, false); }
int fvh_ndt_set_target_cloud_device(fvh_ndt* h, const float* d, int n, int s) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->target_vm.invalidate(); return upload_cloud(&h->e, h->target, d, n, s, true, false); }
int fvh_ndt_create_source_voxelmap(fvh_ndt* h) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  // a swapped-in target map has no compact arrays: rebuild in that case
  if (h->distance_mode == FVH_NDT_P2D) return FVH_OK;  // ndt_cuda.cu:122
  if (h->source_vm.valid && h->source_vm.compact_pts.p) return FVH_OK;
  return build_voxelmap<1>(&h->e, h->source, h->source_vm, h->resolution, true);
}
int fvh_ndt_create_target_voxelmap(fvh_ndt* h) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (h->target_vm.valid) return FVH_OK;  // ndt_cuda.cu:133-135
  return build_voxelmap<1>(&h->e, h->target, h->target_vm, h->resolution, true);
}
int fvh_ndt_create_voxelmaps(fvh_ndt* h) { int rc = fvh_ndt_create_source_voxelmap(h); if (rc) return rc; return fvh_ndt_create_target_voxelmap(h); }
static int ndt_ready(fvh_ndt* h) {
  if (!h->source.has_pts) return h->e.fail(FVH_ERR_BAD_STATE, "source cloud not set");
  if (!h->target_vm.valid) return h->e.fail(FVH_ERR_BAD_STATE, "target voxel map not built (create_voxelmaps)");
  if (h->distance_mode == FVH_NDT_D2D && !(h->source_vm.valid && h->source_vm.compact_pts.p)) return h->e.fail(FVH_ERR_BAD_STATE, "source voxel map not built (create_voxelmaps)");
  if (h->e.tile_n > 1) {  // a tiled handle walks a range of an ORDER every rank agrees on
    Engine* e = &h->e;
    if (h->distance_mode == FVH_NDT_P2D) {  // the points' Morton order (a function of the cloud alone)
      if (h->source.n) { int rc = ensure_sorted(e, h->source); if (rc) return rc; }
    } else {   // the source voxels ranked by key
      int rc = h->ensure_canon(); if (rc) return rc;
    }
  }
  return FVH_OK;
}
// north_star: "large scans shard by spatial tile ... all-reduce of the normal equations per iteration", for NDT. A tiled handle holds the
// FULL clouds (the target map is replicated) and evaluates ONE spatial tile of the source: P2D -- chunk `rank` of `nranks` equal chunks of
// the source points' Morton order, exactly as a sharded VGICP handle; D2D -- the same chunk of the source map's voxels ranked by key
// (the compact list of a map is ordered by atomics and differs between two builds of the same map: ndt_cuda.cu:142-161 walks it as it
// comes; the canonical order makes every rank cut the same list). update_correspondences / compute_error then return the tile's PARTIAL
// sums (err, H, b) -- the caller adds them over the ranks (fast_gicp_amd/distributed.py: ShardedNDT, host route); with a communicator
// attached (fvh_ndt_comm_init with the same nranks) align() all-reduces them on the device between the launches of the LM loop.
int fvh_ndt_set_source_tile(fvh_ndt* h, int rank, int nranks) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (nranks < 1 || rank < 0 || rank >= nranks) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "set_source_tile: need 0 <= rank < nranks");
  if (h->e.comm && nranks > 1 && (nranks != h->e.nranks || rank != h->e.rank)) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "set_source_tile: rank / nranks differ from the attached communicator's");
  h->e.tile_rank = nranks > 1 ? rank : 0;
  h->e.tile_n = nranks;
  h->e.has_corr = false;
  return FVH_OK;
}
int fvh_ndt_update_correspondences(fvh_ndt* h, const double* T) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  int rc = ndt_ready(h); if (rc) return rc;
  if (h->distance_mode == FVH_NDT_P2D) return do_update_correspondences<MODE_NDT_P2D>(&h->e, h->cost_source(), h->target_vm, T);
  return do_update_correspondences<MODE_NDT_D2D>(&h->e, h->cost_source(), h->target_vm, T);
}
int fvh_ndt_compute_error(fvh_ndt* h, const double* T, double* H, double* b, double* err) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  int rc = ndt_ready(h); if (rc) return rc;
  if (h->distance_mode == FVH_NDT_P2D) return do_compute_error<MODE_NDT_P2D>(&h->e, h->cost_source(), h->target_vm, T, H, b, err, h->rebuild_safe());
  return do_compute_error<MODE_NDT_D2D>(&h->e, h->cost_source(), h->target_vm, T, H, b, err, h->rebuild_safe());
}
int fvh_ndt_align(fvh_ndt* h, const double* guess, const fvh_lm_params* p, fvh_lm_result* r) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (h->e.tile_n > 1 && !h->e.comm) return h->e.fail(FVH_ERR_BAD_STATE, "align: this handle evaluates one tile of the source (fvh_ndt_set_source_tile) and has no communicator: its sums are "
                                                                        "partial -- all-reduce update_correspondences / compute_error over the ranks, or attach one (fvh_ndt_comm_init)");
  int rc = fvh_ndt_create_voxelmaps(h);  // NDTCuda::computeTransformation (ndt_cuda_impl.hpp:76-79)
  if (rc) return rc;
  rc = ndt_ready(h); if (rc) return rc;
  if (h->distance_mode == FVH_NDT_P2D) return do_align<MODE_NDT_P2D>(&h->e, h->cost_source(), h->target_vm, guess, p, r, h->rebuild_safe());
  return do_align<MODE_NDT_D2D>(&h->e, h->cost_source(), h->target_vm, guess, p, r, h->rebuild_safe());
}
// ---- pipelined frame streams: align = launch + wait; the next source is prepared beside the running LM kernel ----
int fvh_ndt_align_async(fvh_ndt* h, const double* guess, const fvh_lm_params* p) {
  CHECK_HANDLE(h);
  if (h->e.tile_n > 1) return h->e.fail(FVH_ERR_UNSUPPORTED, "align_async: not on a tiled handle (fvh_ndt_set_source_tile)");
  if (h->pending.active) return h->e.fail(FVH_ERR_BAD_STATE, "align_async: the previous align_async has not been waited for");
  int rc = fvh_ndt_create_voxelmaps(h);
  if (rc) return rc;
  rc = ndt_ready(h); if (rc) return rc;
  if (h->distance_mode == FVH_NDT_P2D) return align_begin<MODE_NDT_P2D>(&h->e, h->pending, h->cost_source(), h->target_vm, guess, p);
  return align_begin<MODE_NDT_D2D>(&h->e, h->pending, h->cost_source(), h->target_vm, guess, p);
}
int fvh_ndt_align_wait(fvh_ndt* h, fvh_lm_result* r) {
  CHECK_HANDLE_HOST_ONLY(h);
  if (!h->pending.active) return h->e.fail(FVH_ERR_BAD_STATE, "align_wait: no align_async in flight");
  if (h->distance_mode == FVH_NDT_P2D) return align_finish<MODE_NDT_P2D>(&h->e, h->pending, h->cost_source(), h->target_vm, r, h->rebuild_safe());
  return align_finish<MODE_NDT_D2D>(&h->e, h->pending, h->cost_source(), h->target_vm, r, h->rebuild_safe());
}
int fvh_ndt_prepare_source_device(fvh_ndt* h, const float* d_xyz, int n, int stride) {
  CHECK_HANDLE_HOST_ONLY(h);  // (touches the prepared slot and the second stream only: legal between align_async and align_wait)
  Engine* e = &h->e;
  hipStream_t ps = e->side_stream();  // null (multi-GPU handle, FVH_SIDE_STREAM=0): in order on the main stream -- correct, nothing overlaps
  if (ps == nullptr && h->pending.active) return e->fail(FVH_ERR_BAD_STATE, "prepare_source: this handle has no second stream; call it outside align_async .. align_wait");
  if (!h->prep_done) HIP_OR_FAIL(e, hipEventCreateWithFlags(&h->prep_done, hipEventDisableTiming));
  h->next_ready = false;
  h->next_vm.invalidate();
  int rc = upload_cloud(e, h->next_source, d_xyz, n, stride, true, false, ps);
  if (rc) return rc;
  {
    // D2D registers the map itself; in both modes it is the TARGET map of the frame after (swap_source_and_target): built here it is
    // hidden too. Its table is sized like the maps of the frames before it (the voxel count the last align saw).
    if (h->next_vm.nv_hint < 0) h->next_vm.nv_hint = std::max(h->source_vm.nv_hint, h->target_vm.nv_hint);
    rc = build_voxelmap<1>(e, h->next_source, h->next_vm, h->resolution, true, false, ps, false, /*detached=*/true);
    if (rc) return rc;
  }
  HIP_OR_FAIL(e, hipEventRecord(h->prep_done, ps ? ps : e->stream));
  h->next_ready = true;
  return FVH_OK;
}
int fvh_ndt_adopt_prepared_source(fvh_ndt* h) {
  CHECK_HANDLE(h);
  Engine* e = &h->e;
  if (h->pending.active) return e->fail(FVH_ERR_BAD_STATE, "adopt_prepared_source: an align_async is in flight");
  if (!h->next_ready) return e->fail(FVH_ERR_BAD_STATE, "adopt_prepared_source: nothing prepared (fvh_ndt_prepare_source_device)");
  h->next_ready = false;
  h->source.swap(h->next_source);
  std::swap(h->source_vm, h->next_vm);
  e->has_corr = false;
  // As a rule the preparation ended while the last align was still running: the host sees that at no cost. Otherwise the main stream waits.
  if (e->side) {
    hipError_t q = hipErrorNotReady;
    for (int spins = 0; spins < 64 && q == hipErrorNotReady; spins++) q = hipEventQuery(h->prep_done);
    if (q != hipSuccess) {
      (void)hipGetLastError();
      HIP_OR_FAIL(e, hipStreamWaitEvent(e->stream, h->prep_done, 0));
    }
  }
  return FVH_OK;
}
int fvh_ndt_debug_set_voxel_hint(fvh_ndt* h, int which, int num_voxels) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); (which ? h->target_vm : h->source_vm).nv_hint = num_voxels; return FVH_OK; }
int fvh_ndt_set_lm_trace(fvh_ndt* h, int on) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.lm_trace_on = on != 0; return FVH_OK; }
int fvh_ndt_get_lm_trace(fvh_ndt* h, int* n, double* rows6) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); return get_lm_trace(&h->e, n, rows6); }
int fvh_ndt_fitness_score(fvh_ndt* h, const double* T, double max_range, double* score) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); return do_fitness(&h->e, h->source, h->target, T, max_range, score); }
int fvh_ndt_get_num_voxels(fvh_ndt* h, int which, int* n) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (!n) return FVH_ERR_INVALID_ARGUMENT;
  VoxelMapDev& vm = which ? h->target_vm : h->source_vm;
  const Rebuild rb = h->rebuild_safe();
  int rc = fetch_voxelmap_host(&h->e, vm, &rb); if (rc) return rc;
  *n = (int)vm.h_occupied.size();
  return FVH_OK;
}
int fvh_ndt_get_voxels(fvh_ndt* h, int which, int* coords3, int* num_points, float* means3, float* covs9) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  const Rebuild rb = h->rebuild_safe();
  return get_voxels_host(&h->e, which ? h->target_vm : h->source_vm, coords3, num_points, means3, covs9, &rb);
}
int fvh_ndt_get_num_correspondences(fvh_ndt* h, int* n) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (h->e.tile_n > 1) return h->e.fail(FVH_ERR_UNSUPPORTED, "correspondence getters: not on a tiled handle (only its tile's rows of the list exist)");
  if (!n) return FVH_ERR_INVALID_ARGUMENT;
  std::vector<int> corr;
  int rc = fetch_corr(&h->e, h->e.corr_n_src, corr);
  if (rc) return rc;
  int nsrc = h->e.corr_n_src;
  if (h->distance_mode == FVH_NDT_D2D) {  // only the first num_source_voxels rows are live
    rc = fetch_voxelmap_host(&h->e, h->source_vm);  // (an overflow of either map was already answered by the evaluation that made `corr`)
    if (rc) return rc;
    nsrc = (int)h->source_vm.h_occupied.size();
  }
  int c = 0;
  for (size_t i = 0; i < (size_t)nsrc * h->e.n_off; i++) c += (corr[i] >= 0);
  *n = c;
  return FVH_OK;
}
// offset-major then source element, invalid pairs removed (ndt_cuda.cu:142-161 builds the list with the functor of find_voxel_correspondences.cu:84-111)
int fvh_ndt_get_voxel_correspondences(fvh_ndt* h, int* pairs) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (h->e.tile_n > 1) return h->e.fail(FVH_ERR_UNSUPPORTED, "correspondence getters: not on a tiled handle (only its tile's rows of the list exist)");
  if (!pairs) return FVH_ERR_INVALID_ARGUMENT;
  const Rebuild rb = h->rebuild_safe();
  int rc = fetch_voxelmap_host(&h->e, h->target_vm, &rb);
  if (rc) return rc;
  std::vector<int> corr;
  rc = fetch_corr(&h->e, h->e.corr_n_src, corr);
  if (rc) return rc;
  int nsrc = h->e.corr_n_src;
  if (h->distance_mode == FVH_NDT_D2D) {  // only the first num_source_voxels rows are live; row i = source voxel i of the getter
    rc = fetch_voxelmap_host(&h->e, h->source_vm);
    if (rc) return rc;
    nsrc = (int)h->source_vm.h_occupied.size();
  }
  std::vector<int> row_of;  // (P2D on a large cloud: rows by Morton position; D2D: by source-voxel index)
  rc = fetch_row_of_point(&h->e, h->source, nsrc, row_of);
  if (rc) return rc;
  size_t w = 0;
  for (int o = 0; o < h->e.n_off; o++)
    for (int i = 0; i < nsrc; i++) {
      const int b = corr[(size_t)row_of[(size_t)i] * h->e.n_off + o];
      if (b < 0) continue;
      pairs[2 * w] = i;
      pairs[2 * w + 1] = h->target_vm.bucket_to_index[b];
      w++;
    }
  return FVH_OK;
}
int fvh_ndt_profile_enable(fvh_ndt* h, int on) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.prof.on = on != 0; h->e.prof.cost_only = on == 2; return FVH_OK; }
int fvh_ndt_profile_reset(fvh_ndt* h) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream)); h->e.prof.reset(); return FVH_OK; }
int fvh_ndt_profile_get(fvh_ndt* h, const char* cls, double* ms, int* n) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); return profile_get(&h->e, cls, ms, n); }
int fvh_ndt_synchronize(fvh_ndt* h) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream)); h->e.gang_clear(); return FVH_OK; }
int fvh_ndt_comm_init(fvh_ndt* h, const void* id, int nranks, int rank) {
  CHECK_HANDLE(h); NDT_NOT_PENDING(h);
  if (h->e.tile_n > 1 && (h->e.tile_n != nranks || h->e.tile_rank != rank)) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "comm_init: rank / nranks differ from fvh_ndt_set_source_tile's");
  return comm_init(&h->e, id, nranks, rank);
}
int fvh_ndt_comm_destroy(fvh_ndt* h) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); if (h->e.comm) { g_rccl.CommDestroy(h->e.comm); h->e.comm = nullptr; } h->e.nranks = 1; h->e.rank = 0; return FVH_OK; }

// ---- voxel-grid downsampling ----
int fvh_voxelgrid_create(int device, fvh_voxelgrid** out) {
  if (!out) return FVH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  auto* h = new (std::nothrow) fvh_voxelgrid();
  if (!h) return FVH_ERR_HIP;
  int rc = h->e.init(device, false);
  if (rc) { fprintf(stderr, "fvh_voxelgrid_create: %s\n", h->e.err.c_str()); h->e.shutdown(); delete h; return rc; }
  *out = h;
  return FVH_OK;
}
int fvh_voxelgrid_destroy(fvh_voxelgrid* h) {
  if (!h) return FVH_ERR_INVALID_ARGUMENT;
  (void)hipSetDevice(h->e.device);
  if (h->e.stream) (void)hipStreamSynchronize(h->e.stream);
  h->d.release();
  h->e.shutdown();
  delete h;
  return FVH_OK;
}
const char* fvh_voxelgrid_last_error(const fvh_voxelgrid* h) { return h ? h->e.err.c_str() : "null handle"; }
int fvh_voxelgrid_filter(fvh_voxelgrid* h, int method, const float* xyz, int n, float leaf, int* out_n) { CHECK_HANDLE(h); return downsample(&h->e, h->d, method, xyz, n, 3, false, leaf, out_n); }
int fvh_voxelgrid_filter_strided(fvh_voxelgrid* h, int method, const float* xyz, int n, int stride, float leaf, int* out_n) { CHECK_HANDLE(h); return downsample(&h->e, h->d, method, xyz, n, stride, false, leaf, out_n); }
int fvh_voxelgrid_filter_device(fvh_voxelgrid* h, int method, const float* d_xyz, int n, int stride, float leaf, int* out_n) { CHECK_HANDLE(h); return downsample(&h->e, h->d, method, d_xyz, n, stride, true, leaf, out_n); }
// Extension for a device-resident pipeline (no PCL counterpart): the filter runs on the registration handle's stream, so what it
// writes is ordered before whatever that handle queues next, and the _async call returns as soon as the COUNT is known (it comes
// from the scan kernel, one kernel before the centroids exist): set_*_cloud_device + align are queued behind the emit kernel
// while it runs -- no idle stream between the filter and the registration. The output buffer is complete in stream order only:
// without a shared stream the call is the synchronous one.
int fvh_voxelgrid_share_stream_with_ndt(fvh_voxelgrid* h, fvh_ndt* other) {
  CHECK_HANDLE(h);
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  if (other && other->e.device != h->e.device) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "share_stream: the two handles live on different devices");
  h->e.stream = other ? other->e.stream : h->e.owned_stream;  // null: back to the filter's own stream
  h->e.stream_owner = other ? &other->e : nullptr;
  return FVH_OK;
}
int fvh_voxelgrid_share_prepare_stream_with_ndt(fvh_voxelgrid* h, fvh_ndt* other) {
  CHECK_HANDLE(h);
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  if (!other) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "share_prepare_stream: null registration handle");
  if (other->e.device != h->e.device) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "share_stream: the two handles live on different devices");
  hipStream_t ps = other->e.side_stream();
  h->e.stream = ps ? ps : other->e.stream;
  h->e.stream_owner = &other->e;
  return FVH_OK;
}
int fvh_voxelgrid_share_stream_with_vgicp(fvh_voxelgrid* h, fvh_vgicp* other) {
  CHECK_HANDLE(h);
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  if (other && other->e.device != h->e.device) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "share_stream: the two handles live on different devices");
  h->e.stream = other ? other->e.stream : h->e.owned_stream;
  h->e.stream_owner = other ? &other->e : nullptr;
  return FVH_OK;
}
int fvh_voxelgrid_filter_device_async(fvh_voxelgrid* h, int method, const float* d_xyz, int n, int stride, float leaf, int* out_n) {
  CHECK_HANDLE(h);
  return downsample(&h->e, h->d, method, d_xyz, n, stride, true, leaf, out_n, /*early=*/h->e.stream != h->e.owned_stream);
}
int fvh_voxelgrid_get_points(fvh_voxelgrid* h, float* out_xyz) {
  CHECK_HANDLE(h);
  if (h->d.out_n == 0) return FVH_OK;
  if (!out_xyz) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: null output");
  HIP_OR_FAIL(&h->e, hipMemcpyAsync(out_xyz, h->d.out.p, sizeof(float) * 3 * (size_t)h->d.out_n, hipMemcpyDefault, h->e.stream));
  HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream));
  return FVH_OK;
}
int fvh_voxelgrid_device_points(fvh_voxelgrid* h, const float** d_xyz, int* n) {
  CHECK_HANDLE(h);
  if (!d_xyz || !n) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: null output");
  *d_xyz = h->d.out_n ? h->d.out.as<float>() : nullptr;
  *n = h->d.out_n;
  return FVH_OK;
}
int fvh_voxelgrid_profile_enable(fvh_voxelgrid* h, int on) { CHECK_HANDLE(h); h->e.prof.on = on != 0; h->e.prof.cost_only = on == 2; return FVH_OK; }
int fvh_voxelgrid_profile_reset(fvh_voxelgrid* h) { CHECK_HANDLE(h); HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream)); h->e.prof.reset(); return FVH_OK; }
int fvh_voxelgrid_profile_get(fvh_voxelgrid* h, double* ms, int* n) { CHECK_HANDLE(h); return profile_get(&h->e, "downsample", ms, n); }

}  // extern "C"
