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
#include "host_runtime.inc.hpp"

struct Engine {
  int device = 0;
  fvh_engine_params params = env_engine_defaults();  // this handle's routes / thresholds / watchdogs (fvh_*_set_engine_params)
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
  bool async_in_flight = false;    // fvh_vgicp_align_async .. _align_wait: the handle's clouds, map and LM state belong to the running kernel
  // A persistent LM grid that takes more than 3/4 of the device's co-resident workgroup slots must not share the chip with another stream's
  // kernels while it is being DISPATCHED: with every XCD nearly full the dispatcher places workgroups wherever room appears, the
  // block -> XCD pattern the XCD-local hand-offs rely on (kernels_cost.hpp) no longer holds, and the launch ends in its watchdog (measured with
  // the pipelined loops: 0 aborts in 400 aligns up to 82 % of the slots, 1 at 93 %, 130 at 100 % -- 50 ms each). While such a grid is in
  // flight (`lm_crowds_chip`) the prepared-source calls and a voxel-grid filter on the prepare stream queue on the MAIN stream, behind
  // the kernel; and such a launch first waits for a preparation that is still running (`crowd_fence`).
  bool lm_crowds_chip = false;
  hipEvent_t crowd_fence = nullptr;  // set by the handle for one align: the event behind a preparation it has not adopted yet
  static bool crowds(int blocks, int capacity) { return (long long)blocks * 4 > (long long)capacity * 3; }
  bool feeder_on_main = false;       // where the last preparation / prepare-stream filter was queued
  hipEvent_t flip_ev = nullptr;
  hipEvent_t take_ev = nullptr;      // take_filter_output: orders the consuming stream after the filter's
  // the stream a preparation (or a filter that feeds one) is queued on: the second stream (*out = that stream), or -- no second stream, or a
  // crowding grid in flight -- the main one (*out = null). When the choice flips, the new stream is ordered after what the old one holds.
  int feeder_stream(hipStream_t* out) {
    hipStream_t s2 = side_stream();
    *out = nullptr;
    if (!s2) return FVH_OK;
    const bool want_main = lm_crowds_chip;
    if (want_main != feeder_on_main) {
      hipError_t r;
      if (!flip_ev && (r = hipEventCreateWithFlags(&flip_ev, hipEventDisableTiming)) != hipSuccess) return hipfail(r, "hipEventCreate");
      if ((r = hipEventRecord(flip_ev, feeder_on_main ? stream : s2)) != hipSuccess) return hipfail(r, "hipEventRecord");
      if ((r = hipStreamWaitEvent(want_main ? stream : s2, flip_ev, 0)) != hipSuccess) return hipfail(r, "hipStreamWaitEvent");
      feeder_on_main = want_main;
    }
    if (!want_main) *out = s2;
    return FVH_OK;
  }
  bool side_pending = false;       // a build on `side` the main stream has not been ordered after yet
  bool quiet = false, was_quiet = false;  // the main stream had drained when the current API call began (set by align on return)
  hipStream_t side_stream() {
    if (!params.side_stream || comm || peer.attached()) return nullptr;
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
      // (The second stream stays lazy -- made by the first call that uses it, ~9 ms once. Made here, with the handle, it took that out of a
      // caller's first loop iteration, but every stream is a hardware queue: eight processes sharing one GPU no longer met in the mailbox
      // self-check within 5 s, and four align-only handles of one process ran 8 x slower -- queues time-sliced under spinning kernels.)
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
    if (flip_ev) { (void)hipEventDestroy(flip_ev); flip_ev = nullptr; }
    if (take_ev) { (void)hipEventDestroy(take_ev); take_ev = nullptr; }
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

#include "host_stages.inc.hpp"
#include "host_gicp.inc.hpp"
#include "host_downsample.inc.hpp"

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
  // Pipelined scan streams (fvh_vgicp_prepare_source_device / _adopt_prepared_source / _align_async / _align_wait): the NEXT source cloud -- its
  // Morton order, neighbour lists, covariances AND its own voxel map, which swap_source_and_target() needs one registration later -- is
  // prepared in `next_source` / `next_map` on the handle's second stream (Engine::side) while the LM kernel of the current pair runs on the
  // main one. `source_map` is the map that came with the adopted source: swap_source_and_target() then swaps maps instead of building one.
  CloudDev next_source;
  VoxelMapDev next_map, source_map;
  bool next_ready = false;
  hipEvent_t prep_done = nullptr;
  AlignCtx pending;
  void source_changed() { source_map.invalidate(); }
  bool live_map_stale = false;  // the target's covariances changed after `voxelmap` was built (legal: the map stays as built until create_target_voxelmap -- but it must not be carried over a swap)
  void map_rules_changed() { source_map.invalidate(); next_map.invalidate(); next_ready = false; }  // accumulation mode / precision: prepared maps were built under the old rule
  CostSource cost_source() const {
    // multi-GPU: the tiles are ranges of the Morton order whatever the size of the cloud (spatially compact shards)
    const int* order = e.sharded() ? (source.has_sorted ? source.order.as<int>() : nullptr) : coherent_order(source, e.params.coherent_min_points);
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
    live_map_stale = false;
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
      CostSource cs{source.pts.as<float4>(), nullptr, nullptr, source.n, nullptr, tiled ? (source.has_sorted ? source.order.as<int>() : nullptr) : coherent_order(source, e.params.coherent_min_points)};
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
  if ((h)->e.async_in_flight) return (h)->e.fail(FVH_ERR_BAD_STATE, "an align_async is in flight: call fvh_vgicp_align_wait first"); \
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
void fvh_default_engine_params(fvh_engine_params* p) { if (p) *p = env_engine_defaults(); }
static int get_engine_params(Engine* e, fvh_engine_params* out) { if (!out) return e->fail(FVH_ERR_INVALID_ARGUMENT, "get_engine_params: null"); *out = e->params; return FVH_OK; }
static int set_engine_params(Engine* e, const fvh_engine_params* p) {
  if (!p) return e->fail(FVH_ERR_INVALID_ARGUMENT, "set_engine_params: null");
  if (const char* why = check_engine_params(*p)) return e->fail(FVH_ERR_INVALID_ARGUMENT, why);
  e->params = *p;
  return FVH_OK;
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
  if (h->pending.active) { h->pending.release_slots(); h->pending.active = false; }
  h->source.release(); h->target.release(); h->voxelmap.release(); h->gicp_records.release(); h->next_source.release(); h->next_map.release(); h->source_map.release();
  if (h->prep_done) (void)hipEventDestroy(h->prep_done);
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
  if ((p == FVH_COMPUTE_CUDA_COMPAT) != (h->e.precision == FVH_COMPUTE_CUDA_COMPAT)) { h->voxelmap.invalidate(); h->map_rules_changed(); h->e.has_corr = false; }  // the voxel records of the two arithmetics differ (kernels_compat.hpp)
  h->e.precision = p;
  return FVH_OK;
}

int fvh_vgicp_get_engine_params(fvh_vgicp* h, fvh_engine_params* out) { CHECK_HANDLE_HOST_ONLY(h); return get_engine_params(&h->e, out); }
int fvh_vgicp_set_engine_params(fvh_vgicp* h, const fvh_engine_params* p) { CHECK_HANDLE(h); return set_engine_params(&h->e, p); }

int fvh_vgicp_create_target_voxelmap(fvh_vgicp* h) { CHECK_HANDLE(h); return h->build_map(h->resolution); }
int fvh_vgicp_set_voxel_accumulation_mode(fvh_vgicp* h, int mode) {
  CHECK_HANDLE(h);
  if (mode < 0 || mode > 2) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "unknown voxel accumulation mode");
  if (mode != h->voxel_mode) { h->voxelmap.invalidate(); h->map_rules_changed(); h->e.has_corr = false; }
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
  std::swap(h->voxelmap, h->source_map);   // (the map of the old target stays with its cloud, now the source ...
  if (h->live_map_stale) h->source_map.invalidate();  // ... unless that cloud's covariances changed after the map was built)
  h->live_map_stale = false;
  if (!h->target.has_pts || !h->target.has_cov) { h->voxelmap.invalidate(); return FVH_OK; }  // fast_vgicp_cuda.cu:102-104
  // the new target came through the prepared-source slot with its own voxel map (built beside an earlier LM kernel): nothing to build
  if (h->voxelmap.valid && !h->voxelmap.is_shard && h->voxelmap.res == h->resolution) return FVH_OK;
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
  h->source_map.invalidate();
  return FVH_OK;
}
static void cloud_replaced(CloudDev& c) { c.has_cov = false; c.has_nbr = false; c.has_cov_sorted = false; }
// The Morton order of a VGICP cloud is queued right behind its upload: every neighbour search needs it (and large clouds walk the LM
// loop in it), and the caller's next call -- find_*_neighbors as a rule -- would launch it a few microseconds of host time later
// with the stream idle in between (kernel trace: 4 us between the pack kernel and the sort).
static int uploaded(Engine* e, CloudDev& c, int rc) {
  if (rc || c.n == 0) return rc;
  if (!e->device_search_seen && !e->sharded() && c.n < e->params.coherent_min_points) return FVH_OK;  // nobody may ever need the order: it stays lazy (ensure_sorted where it is consumed)
  return ensure_sorted(e, c);
}
int fvh_vgicp_set_source_cloud(fvh_vgicp* h, const float* xyz, int n) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); h->source_changed(); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, xyz, n, 3, false)); }
int fvh_vgicp_set_target_cloud(fvh_vgicp* h, const float* xyz, int n) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, xyz, n, 3, false)); }
int fvh_vgicp_set_source_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); h->source_changed(); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, xyz, n, stride, false)); }
int fvh_vgicp_set_target_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, xyz, n, stride, false)); }
int fvh_vgicp_set_source_cloud_device(fvh_vgicp* h, const float* d, int n, int stride) { CHECK_HANDLE_SOURCE_CHAIN(h); h->e.has_corr = false; cloud_replaced(h->source); h->source_changed(); return uploaded(&h->e, h->source, upload_cloud(&h->e, h->source, d, n, stride, true)); }
int fvh_vgicp_set_target_cloud_device(fvh_vgicp* h, const float* d, int n, int stride) { CHECK_HANDLE(h); h->e.has_corr = false; h->voxelmap.invalidate(); h->gicp_records.invalidate(); cloud_replaced(h->target); return uploaded(&h->e, h->target, upload_cloud(&h->e, h->target, d, n, stride, true)); }
int fvh_vgicp_set_source_neighbors(fvh_vgicp* h, int k, const int* idx) { CHECK_HANDLE(h); return set_neighbors(&h->e, h->source, k, idx); }
int fvh_vgicp_set_target_neighbors(fvh_vgicp* h, int k, const int* idx) { CHECK_HANDLE(h); return set_neighbors(&h->e, h->target, k, idx); }
int fvh_vgicp_find_source_neighbors(fvh_vgicp* h, int k) { CHECK_HANDLE_SOURCE_CHAIN(h); return h->e.after_source_chain_call(find_neighbors(&h->e, h->source, k)); }
int fvh_vgicp_find_target_neighbors(fvh_vgicp* h, int k) { CHECK_HANDLE(h); return find_neighbors(&h->e, h->target, k); }
int fvh_vgicp_calculate_source_covariances(fvh_vgicp* h, int m) { CHECK_HANDLE_SOURCE_CHAIN(h); h->source_changed(); return h->e.after_source_chain_call(calc_cov_knn(&h->e, h->source, m)); }
int fvh_vgicp_calculate_target_covariances(fvh_vgicp* h, int m) { CHECK_HANDLE(h); h->live_map_stale = true; return calc_cov_knn(&h->e, h->target, m); }
int fvh_vgicp_calculate_source_covariances_rbf(fvh_vgicp* h, int m) { CHECK_HANDLE_SOURCE_CHAIN(h); h->source_changed(); return h->e.after_source_chain_call(calc_cov_rbf(&h->e, h->source, h->kernel_width, h->kernel_max_dist, m)); }
int fvh_vgicp_calculate_target_covariances_rbf(fvh_vgicp* h, int m) { CHECK_HANDLE(h); h->live_map_stale = true; return calc_cov_rbf(&h->e, h->target, h->kernel_width, h->kernel_max_dist, m); }
int fvh_vgicp_set_source_covariances(fvh_vgicp* h, const double* c) { CHECK_HANDLE(h); h->source_changed(); return set_cov_host(&h->e, h->source, c); }
int fvh_vgicp_set_target_covariances(fvh_vgicp* h, const double* c) { CHECK_HANDLE(h); h->live_map_stale = true; return set_cov_host(&h->e, h->target, c); }

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
  h->e.crowd_fence = h->next_ready ? h->prep_done : nullptr;  // (a preparation still running on the second stream: a grid that crowds the chip waits for it)
  const int rc = do_align<MODE_VGICP>(&h->e, h->cost_source(), h->voxelmap, guess, p, r, h->rebuild_safe());
  // the result is on the host: everything this handle queued has run (of a persistent launch only the workgroups' exit remains,
  // and they touch neither clouds nor the map any more)
  if (rc == FVH_OK) h->e.quiet = true;
  return rc;
}
// ---- pipelined scan streams (scan-to-scan odometry: src/kitti.cpp's loop with the next scan prepared beside the running LM kernel) ----
//   prepare_source_device(scan k+1)  [second stream: pack, Morton sort, exact k-NN, covariances, the scan's own voxel map]
//   align_async(guess) .. align_wait(result)   [main stream: the persistent LM kernel of the pair (k, k-1)]
//   swap_source_and_target()         [the map that came with scan k becomes the live one: nothing is built]
//   adopt_prepared_source()          [scan k+1 becomes the source]
// Results are those of the sequential calls (same kernels on the same data; voxel sums differ in the order of their fp64 atomics only).
int fvh_vgicp_align_async(fvh_vgicp* h, const double* guess, const fvh_lm_params* p) {
  CHECK_HANDLE(h);
  if (h->e.sharded()) return h->e.fail(FVH_ERR_UNSUPPORTED, "align_async: not on a multi-GPU handle");
  if (!h->source.has_pts || !h->source.has_cov) return h->e.fail(FVH_ERR_BAD_STATE, "align: source cloud/covariances not set");
  h->e.crowd_fence = h->next_ready ? h->prep_done : nullptr;
  const int rc = align_begin<MODE_VGICP>(&h->e, h->pending, h->cost_source(), h->voxelmap, guess, p);
  if (rc == FVH_OK) h->e.async_in_flight = true;
  return rc;
}
int fvh_vgicp_align_wait(fvh_vgicp* h, fvh_lm_result* r) {
  CHECK_HANDLE_HOST_ONLY(h);
  if (!h->pending.active) return h->e.fail(FVH_ERR_BAD_STATE, "align_wait: no align_async in flight");
  h->e.async_in_flight = false;
  const int rc = align_finish<MODE_VGICP>(&h->e, h->pending, h->cost_source(), h->voxelmap, r, h->rebuild_safe());
  if (rc == FVH_OK) h->e.quiet = true;
  return rc;
}
static int vgicp_prepare(fvh_vgicp* h, const float* d_xyz, int n, int stride, bool on_device, int k, int regularization, int rbf, int stages) {
  CHECK_HANDLE_HOST_ONLY(h);  // (touches the prepared slot and the second stream only: legal between align_async and align_wait)
  Engine* e = &h->e;
  if (stages < 1 || stages > 3) return e->fail(FVH_ERR_INVALID_ARGUMENT, "prepare_source: stages must be 1 (order + neighbours), 2 (+ covariances) or 3 (+ voxel map)");
  if (e->sharded()) return e->fail(FVH_ERR_UNSUPPORTED, "prepare_source: not on a multi-GPU handle");
  hipStream_t ps = e->side_stream();  // null (FVH_SIDE_STREAM=0): in order on the main stream -- correct, nothing overlaps
  if (ps == nullptr && h->pending.active) return e->fail(FVH_ERR_BAD_STATE, "prepare_source: this handle has no second stream; call it outside align_async .. align_wait");
  if (!h->prep_done) HIP_OR_FAIL(e, hipEventCreateWithFlags(&h->prep_done, hipEventDisableTiming));
  // a target-map build that swap_source_and_target() deferred goes first (same stream: in order)
  if (e->deferred) { const int rc = e->after_source_chain_call(FVH_OK); if (rc) return rc; }
  { const int rc = e->feeder_stream(&ps); if (rc) return rc; }
  h->next_ready = false;
  h->next_map.invalidate();
  cloud_replaced(h->next_source);
  // The stages below are the ones the sequential calls run; they launch on the handle's current stream: for the length of this call that is
  // the second one. The scratch they use (sort keys / histograms, the RBF sums) is not touched by a running LM kernel.
  struct StreamSwap { Engine* e; hipStream_t main; ~StreamSwap() { e->stream = main; } } swap{e, e->stream};
  if (ps) e->stream = ps;
  // Beside a running LM kernel the cloud is sorted by the multi-kernel radix passes, not by the cooperative kernel: that kernel's finish /
  // fall-back kernel is a 1,024-thread workgroup at 120 VGPRs (it holds the whole one-workgroup sort), which fits on no CU that hosts LM
  // workgroups -- it sat in the queue until the LM kernel ended (kernel trace: 97 us instead of 4.3), and the neighbour search and the
  // covariances behind it ran AFTER the LM kernel instead of beside it: 5,760 registrations/s against 6,990 with the passes.
  struct SortMode { Engine* e; int saved; ~SortMode() { e->params.sort_mode = saved; } } sort_mode{e, e->params.sort_mode};
  if (ps != nullptr) e->params.sort_mode = 0;  // (also when the align is launched right AFTER this call: the chain then runs beside it all the same)
  int rc = upload_cloud(e, h->next_source, d_xyz, n, stride, on_device, true, ps);
  if (rc || n == 0) return rc ? rc : e->fail(FVH_ERR_INVALID_ARGUMENT, "prepare_source: empty cloud");
  rc = ensure_sorted(e, h->next_source);
  if (rc) return rc;
  if (rbf) { if (stages >= 2) rc = calc_cov_rbf(e, h->next_source, h->kernel_width, h->kernel_max_dist, regularization); }
  else {
    rc = find_neighbors(e, h->next_source, k);
    if (!rc && stages >= 2) rc = calc_cov_knn(e, h->next_source, regularization);
  }
  if (rc) return rc;
  if (stages >= 3) {
    if (h->next_map.nv_hint < 0) h->next_map.nv_hint = std::max(h->voxelmap.nv_hint, h->source_map.nv_hint);
    rc = h->voxel_mode == 2 ? build_voxelmap<2>(e, h->next_source, h->next_map, h->resolution, false, false, ps, false, /*detached=*/true)
                            : build_voxelmap<0>(e, h->next_source, h->next_map, h->resolution, false, false, ps, false, /*detached=*/true);
    if (rc) return rc;
  }
  HIP_OR_FAIL(e, hipEventRecord(h->prep_done, e->stream));
  h->next_ready = true;
  return FVH_OK;
}
int fvh_vgicp_prepare_source_device(fvh_vgicp* h, const float* d_xyz, int n, int stride, int k, int regularization, int rbf, int stages) { return vgicp_prepare(h, d_xyz, n, stride, true, k, regularization, rbf, stages); }
// the same for a HOST cloud (what the reference's callers hold): consumed before the call returns (a memcpy into the handle's pinned staging
// slot, as fvh_vgicp_set_source_cloud_strided), everything else queued on the second stream
int fvh_vgicp_prepare_source(fvh_vgicp* h, const float* xyz, int n, int stride, int k, int regularization, int rbf, int stages) { return vgicp_prepare(h, xyz, n, stride, false, k, regularization, rbf, stages); }
int fvh_vgicp_adopt_prepared_source(fvh_vgicp* h) {
  CHECK_HANDLE(h);  // (a map build swap_source_and_target() deferred runs here, in order on the main stream: queued on the second one it reached the LM kernel 10 us later -- measured)
  Engine* e = &h->e;
  if (!h->next_ready) return e->fail(FVH_ERR_BAD_STATE, "adopt_prepared_source: nothing prepared (fvh_vgicp_prepare_source_device)");
  h->next_ready = false;
  h->source.swap(h->next_source);
  std::swap(h->source_map, h->next_map);
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
int fvh_ndt_get_engine_params(fvh_ndt* h, fvh_engine_params* out) { CHECK_HANDLE_HOST_ONLY(h); return get_engine_params(&h->e, out); }
int fvh_ndt_set_engine_params(fvh_ndt* h, const fvh_engine_params* p) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); return set_engine_params(&h->e, p); }
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
  h->e.crowd_fence = h->next_ready ? h->prep_done : nullptr;  // (a preparation still running on the second stream: a grid that crowds the chip waits for it)
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
  h->e.crowd_fence = h->next_ready ? h->prep_done : nullptr;
  if (h->distance_mode == FVH_NDT_P2D) return align_begin<MODE_NDT_P2D>(&h->e, h->pending, h->cost_source(), h->target_vm, guess, p);
  return align_begin<MODE_NDT_D2D>(&h->e, h->pending, h->cost_source(), h->target_vm, guess, p);
}
int fvh_ndt_align_wait(fvh_ndt* h, fvh_lm_result* r) {
  CHECK_HANDLE_HOST_ONLY(h);
  if (!h->pending.active) return h->e.fail(FVH_ERR_BAD_STATE, "align_wait: no align_async in flight");
  if (h->distance_mode == FVH_NDT_P2D) return align_finish<MODE_NDT_P2D>(&h->e, h->pending, h->cost_source(), h->target_vm, r, h->rebuild_safe());
  return align_finish<MODE_NDT_D2D>(&h->e, h->pending, h->cost_source(), h->target_vm, r, h->rebuild_safe());
}
// The output of the voxel-grid filter's last ApproximateVoxelGrid call becomes cloud `c` of this handle WITHOUT a copy: the emit kernel wrote
// it as float4 too (kernels_downsample.hpp), and that buffer is swapped with the cloud's (the filter gets the cloud's old buffer as its next
// output buffer: nothing in flight reads it -- it belonged to a cloud that was replaced or swapped out). `consumer`: the stream the
// handle will read the cloud on; ordered after the filter's last kernel unless the host has already seen that kernel finish.
static int take_filter_output(Engine* e, fvh_voxelgrid* vg, CloudDev& c, hipStream_t consumer) {
  if (!vg) return e->fail(FVH_ERR_INVALID_ARGUMENT, "from_voxelgrid: null filter handle");
  if (vg->e.device != e->device) return e->fail(FVH_ERR_INVALID_ARGUMENT, "from_voxelgrid: the two handles live on different devices");
  if (!vg->d.out4_valid) return e->fail(FVH_ERR_BAD_STATE, "from_voxelgrid: no ApproximateVoxelGrid output to take (fvh_voxelgrid_filter* with FVH_VOXELGRID_APPROXIMATE first; an output can be taken once)");
  if (!vg->d.out_complete && vg->d.out_stream != consumer) {
    if (!e->take_ev) HIP_OR_FAIL(e, hipEventCreateWithFlags(&e->take_ev, hipEventDisableTiming));
    HIP_OR_FAIL(e, hipEventRecord(e->take_ev, vg->d.out_stream));
    HIP_OR_FAIL(e, hipStreamWaitEvent(consumer, e->take_ev, 0));
  }
  std::swap(c.pts, vg->d.out4);
  vg->d.out4_valid = false;
  c.n = vg->d.out_n;
  c.has_pts = true;
  c.has_sorted = false;
  c.has_box = false;
  return FVH_OK;
}
static int ndt_prepare(fvh_ndt* h, const float* d_xyz, int n, int stride, bool on_device, fvh_voxelgrid* from_filter = nullptr) {
  CHECK_HANDLE_HOST_ONLY(h);  // (touches the prepared slot and the second stream only: legal between align_async and align_wait)
  Engine* e = &h->e;
  hipStream_t ps = e->side_stream();  // null (multi-GPU handle, FVH_SIDE_STREAM=0): in order on the main stream -- correct, nothing overlaps
  if (ps == nullptr && h->pending.active) return e->fail(FVH_ERR_BAD_STATE, "prepare_source: this handle has no second stream; call it outside align_async .. align_wait");
  if (!h->prep_done) HIP_OR_FAIL(e, hipEventCreateWithFlags(&h->prep_done, hipEventDisableTiming));
  { const int rc0 = e->feeder_stream(&ps); if (rc0) return rc0; }  // (the main stream instead, behind an LM grid that crowds the chip: Engine::lm_crowds_chip)
  h->next_ready = false;
  h->next_vm.invalidate();
  int rc = from_filter ? take_filter_output(e, from_filter, h->next_source, ps ? ps : e->stream) : upload_cloud(e, h->next_source, d_xyz, n, stride, on_device, false, ps);
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
int fvh_ndt_prepare_source_device(fvh_ndt* h, const float* d_xyz, int n, int stride) { return ndt_prepare(h, d_xyz, n, stride, true); }
int fvh_ndt_prepare_source(fvh_ndt* h, const float* xyz, int n, int stride) { return ndt_prepare(h, xyz, n, stride, false); }  // a HOST cloud, consumed before the call returns
int fvh_ndt_prepare_source_from_voxelgrid(fvh_ndt* h, fvh_voxelgrid* vg) {
  if (h && !vg) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "from_voxelgrid: null filter handle");
  return ndt_prepare(h, nullptr, 0, 3, true, vg);
}
int fvh_ndt_set_source_cloud_from_voxelgrid(fvh_ndt* h, fvh_voxelgrid* vg) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->source_vm.invalidate(); return take_filter_output(&h->e, vg, h->source, h->e.stream); }
int fvh_ndt_set_target_cloud_from_voxelgrid(fvh_ndt* h, fvh_voxelgrid* vg) { CHECK_HANDLE(h); NDT_NOT_PENDING(h); h->e.has_corr = false; h->target_vm.invalidate(); return take_filter_output(&h->e, vg, h->target, h->e.stream); }
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
int fvh_voxelgrid_get_engine_params(fvh_voxelgrid* h, fvh_engine_params* out) { CHECK_HANDLE_HOST_ONLY(h); return get_engine_params(&h->e, out); }
int fvh_voxelgrid_set_engine_params(fvh_voxelgrid* h, const fvh_engine_params* p) { CHECK_HANDLE(h); return set_engine_params(&h->e, p); }
// A filter that shares a registration handle's PREPARE stream goes where that handle's preparations go (Engine::feeder_stream): the main
// stream while an LM grid that crowds the chip is in flight. For the length of the call only: the sharing itself stays.
struct FilterStream {
  Engine* e; hipStream_t saved; int rc = FVH_OK;
  explicit FilterStream(Engine* e_) : e(e_), saved(e_->stream) {
    Engine* o = e->stream_owner;
    if (!o || !o->side || e->stream != o->side) return;
    hipStream_t s = nullptr;
    rc = o->feeder_stream(&s);
    if (rc) e->err = o->err;
    else e->stream = s ? s : o->stream;
  }
  ~FilterStream() { e->stream = saved; }
};
int fvh_voxelgrid_filter(fvh_voxelgrid* h, int method, const float* xyz, int n, float leaf, int* out_n) { CHECK_HANDLE(h); FilterStream fs(&h->e); if (fs.rc) return fs.rc; return downsample(&h->e, h->d, method, xyz, n, 3, false, leaf, out_n); }
int fvh_voxelgrid_filter_strided(fvh_voxelgrid* h, int method, const float* xyz, int n, int stride, float leaf, int* out_n) { CHECK_HANDLE(h); FilterStream fs(&h->e); if (fs.rc) return fs.rc; return downsample(&h->e, h->d, method, xyz, n, stride, false, leaf, out_n); }
int fvh_voxelgrid_filter_device(fvh_voxelgrid* h, int method, const float* d_xyz, int n, int stride, float leaf, int* out_n) { CHECK_HANDLE(h); FilterStream fs(&h->e); if (fs.rc) return fs.rc; return downsample(&h->e, h->d, method, d_xyz, n, stride, true, leaf, out_n); }
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
int fvh_voxelgrid_share_prepare_stream_with_vgicp(fvh_voxelgrid* h, fvh_vgicp* other) {
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
  const bool shared = h->e.stream != h->e.owned_stream;
  FilterStream fs(&h->e); if (fs.rc) return fs.rc;
  return downsample(&h->e, h->d, method, d_xyz, n, stride, true, leaf, out_n, /*early=*/shared);
}
int fvh_voxelgrid_get_points(fvh_voxelgrid* h, float* out_xyz) {
  CHECK_HANDLE(h);
  if (h->d.out_n == 0) return FVH_OK;
  if (!out_xyz) return h->e.fail(FVH_ERR_INVALID_ARGUMENT, "voxelgrid: null output");
  if (h->e.stream_owner && h->e.stream_owner->feeder_on_main && h->e.stream == h->e.stream_owner->side) HIP_OR_FAIL(&h->e, hipStreamSynchronize(h->e.stream_owner->stream));  // (an _async filter that was queued there)
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
