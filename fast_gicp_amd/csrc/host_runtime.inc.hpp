// Host runtime of the engine: device buffers, engine parameters (the one place the environment is read), cloud / voxel-map state, HIP-event profiler, RCCL loader, gang registry and slot pool of the persistent kernels.
// (a section of the host translation unit: included by fvh_capi.hip inside its anonymous namespace, after the sections it builds on;
//  split out in round 6 -- the single file had grown to 3,100 lines)


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

// ---- engine parameters (include/fast_vgicp_hip.h: fvh_engine_params) ----
// The ONE place the library reads the process environment: every FVH_* knob of INTEGRATION.md passes through fvh_env(). The defaults of a new
// handle are the built-in values with the environment applied once per process; the process-wide knobs (slot pool, XCD-local hand-offs, small-grid
// layout: every handle must agree on them) live beside them.
inline const char* fvh_env(const char* name) { return getenv(name); }
inline long long fvh_env_ll(const char* name, long long dflt) { const char* v = fvh_env(name); return v ? atoll(v) : dflt; }
inline unsigned long long fvh_env_ull(const char* name, unsigned long long dflt) { const char* v = fvh_env(name); return v ? strtoull(v, nullptr, 10) : dflt; }
constexpr int COHERENT_MIN_POINTS = 32768;  // clouds of this size and up are walked in Morton order (PMC: 2.8x HBM over-fetch on a randomly ordered 100k scan against a 1M-point map)
struct ProcessParams { int slot_max_split, contended_slot_pct, xcd_local, small_grid_layout; };
inline const ProcessParams& process_params() {
  static const ProcessParams p = [] {
    ProcessParams q;
    q.slot_max_split = (int)std::max(1LL, fvh_env_ll("FVH_SLOT_MAX_SPLIT", 4));
    q.contended_slot_pct = (int)std::min(100LL, std::max(10LL, fvh_env_ll("FVH_CONTENDED_SLOT_PCT", 100)));
    q.xcd_local = fvh_env_ll("FVH_XCD_LOCAL", 1) != 0 ? 1 : 0;
    q.small_grid_layout = fvh_env_ll("FVH_SMALL_GRID_LAYOUT", 2) == 0 ? 0 : 2;
    return q;
  }();
  return p;
}
inline const fvh_engine_params& env_engine_defaults() {
  static const fvh_engine_params d = [] {
    fvh_engine_params p;
    std::memset(&p, 0, sizeof(p));
    p.struct_size = (int)sizeof(fvh_engine_params);
    p.sort_mode = (int)fvh_env_ll("FVH_SORT_MODE", 2);
    p.sort_items = (int)fvh_env_ll("FVH_SORT_ITEMS", 0);
    { const int b = (int)fvh_env_ll("FVH_SORT_FUSED_BITS", 10); p.sort_fused_bits = (b == 9 || b == 10) ? b : 0; }
    p.sort_two_pass_max = (int)fvh_env_ll("FVH_SORT_TWO_PASS_MAX", 262144);
    p.sort_coop_watchdog_ticks = 2'000'000ull;  // 20 ms (FVH_SORT_COOP_WATCHDOG_TICKS overrides it per call: test hook)
    p.knn_nearest_first_max_points = (int)fvh_env_ll("FVH_KNN_NEAREST_FIRST_MAX_POINTS", 65536);
    { const int b = (int)fvh_env_ll("FVH_KNN_BLOCK", 64); p.knn_block = (b == 256 || b == 128) ? b : 64; }
    p.coherent_min_points = (int)fvh_env_ll("FVH_COHERENT_MIN_POINTS", COHERENT_MIN_POINTS);
    p.bitmap_min_points = (int)fvh_env_ll("FVH_BITMAP_MIN_POINTS", 300000);
    p.bitmap_max_bytes = std::min(fvh_env_ull("FVH_BITMAP_MAX_BYTES", 32ull << 20), 16ull << 30);  // (the LM kernel indexes the words with 32 bits)
    p.persistent = (int)fvh_env_ll("FVH_PERSISTENT", 1);
    p.persist_watchdog_ticks = 5'000'000ull;    // 50 ms of the 100 MHz wall clock (FVH_PERSIST_WATCHDOG_TICKS overrides it per call: test hook)
    p.peer_watchdog_ticks = fvh_env_ull("FVH_PEER_WATCHDOG_TICKS", 200'000'000ull);  // 2 s: a peer may still be uploading / sorting its copy
    p.lm_everywhere = (int)fvh_env_ll("FVH_LM_EVERYWHERE", 1);
    { const char* v = fvh_env("FVH_COST_PRIO"); p.cost_prio = v ? (atoi(v) != 0 ? 1 : 0) : -1; }
    p.cost_split = (int)fvh_env_ll("FVH_COST_SPLIT", 1);
    p.cost_group_max = (int)std::min<long long>(std::max(1LL, fvh_env_ll("FVH_COST_GROUP_MAX", COST_CH)), COST_CH);  // the kernel keeps one item's lookups in flight together: at most COST_CH
    { const long long b = fvh_env_ll("FVH_COST_MAX_BLOCKS", MAX_PARTIAL_ROWS); p.cost_max_blocks = (int)(b < 1 ? 1 : (b > MAX_PARTIAL_ROWS ? MAX_PARTIAL_ROWS : b)); }
    p.cost_target_items = fvh_env_ll("FVH_COST_TARGET_ITEMS", 256LL * 256 * 2);
    p.zerocopy_result = (int)fvh_env_ll("FVH_ZEROCOPY_RESULT", 1);
    { const char* v = fvh_env("FVH_HOST_WAIT"); p.host_wait_block = (v && std::string(v) == "block") ? 1 : 0; }
    p.result_query_spins = fvh_env_ull("FVH_RESULT_QUERY_SPINS", 1024ull);
    p.side_stream = fvh_env_ll("FVH_SIDE_STREAM", 1) != 0 ? 1 : 0;
    p.pinned_upload_max = fvh_env_ull("FVH_PINNED_UPLOAD_MAX", 8ull << 20);
    p.zerocopy_upload_max = fvh_env_ull("FVH_ZEROCOPY_UPLOAD_MAX", 1ull << 20);
    p.avg_fused = fvh_env_ll("FVH_AVG_FUSED", 1) != 0 ? 1 : 0;
    p.nn1_seed = fvh_env_ll("FVH_NN1_SEED", 1) != 0 ? 1 : 0;
    return p;
  }();
  return d;
}
// the setters' sanity checks (a handle must never hold a value its kernels cannot take)
inline const char* check_engine_params(const fvh_engine_params& p) {
  if (p.struct_size != (int)sizeof(fvh_engine_params)) return "engine params: struct_size does not match this library's fvh_engine_params (start from fvh_default_engine_params / fvh_*_get_engine_params)";
  if (p.sort_mode < 0 || p.sort_mode > 3) return "engine params: sort_mode must be 0..3";
  if (p.sort_fused_bits != 0 && p.sort_fused_bits != 9 && p.sort_fused_bits != 10) return "engine params: sort_fused_bits must be 0, 9 or 10";
  if (p.knn_block != 64 && p.knn_block != 128 && p.knn_block != 256) return "engine params: knn_block must be 64, 128 or 256";
  if (p.cost_group_max < 1 || p.cost_group_max > COST_CH) return "engine params: cost_group_max must be 1..4";
  if (p.cost_max_blocks < 1 || p.cost_max_blocks > MAX_PARTIAL_ROWS) return "engine params: cost_max_blocks must be 1..1024";
  if (p.cost_target_items < 1) return "engine params: cost_target_items must be positive";
  if (p.lm_everywhere < 0 || p.lm_everywhere > 2 || p.cost_prio < -1 || p.cost_prio > 1) return "engine params: lm_everywhere must be 0..2, cost_prio -1..1";
  if (p.bitmap_max_bytes > (16ull << 30)) return "engine params: bitmap_max_bytes beyond 16 GiB (the LM kernel indexes its words with 32 bits)";
  if (p.sort_items < 0 || p.sort_items > SORT_ITEMS_MAX || (p.sort_items & 63)) return "engine params: sort_items must be 0 or a multiple of 64 up to 1024";
  return nullptr;
}

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

// Clouds of fvh_engine_params::coherent_min_points and up are walked in Morton order. Below it everything is L2-resident and the extra
// index load is not worth it.
inline const int* coherent_order(const CloudDev& c, int min_pts) { return (c.has_sorted && c.n >= min_pts) ? c.order.as<int>() : nullptr; }

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
    const int max_split = process_params().slot_max_split;
    // (FVH_CONTENDED_SLOT_PCT: the part of the device the concurrent aligns may hold between them -- the rest stays free for the other
    // streams' neighbour searches and sorts, which cannot start on a CU whose register file three resident LM workgroups fill)
    const int contended_pct = process_params().contended_slot_pct;
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
  return process_params().xcd_local && g_xcd_local_strikes.load() < XCD_LOCAL_MAX_STRIKES;
}
// the layout of one cost launch: both routes of an align take the same (nb, ng), i.e. the same partition and summation order
struct GridPlan { int nb = 0; int ng = 0; int local = 0; };
// Grids of up to SINGLE_LEVEL_MAX_BLOCKS workgroups (NDT D2D over a few thousand source voxels, DIRECT1 at 17k points):
//   FVH_SMALL_GRID_LAYOUT=2 (default)  chip-wide in EIGHT groups like the large grids (XCD-local rows and broadcast, one cross-XCD hand-off);
//                        =0            chip-wide, ONE group: rows -> workgroup 0 -> broadcast, write-through hand-offs (round 3; kept for A/B runs
//                                      and as the second layout the bit-identity tests walk). (=1, one XCD only, was measured worse and is gone.)
// The group count is a function of the grid size alone, so that every route of an align adds the sums in the same order.
inline int small_grid_layout() {
  return process_params().small_grid_layout;
}
inline int default_groups(int nb) {
  if (nb > SINGLE_LEVEL_MAX_BLOCKS) return TICKET_GROUPS;
  return (small_grid_layout() == 2 && nb >= 2 * TICKET_GROUPS) ? TICKET_GROUPS : 1;
}

