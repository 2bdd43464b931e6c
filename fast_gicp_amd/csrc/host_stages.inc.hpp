// The stages of a registration on one Engine: upload, Morton sort, neighbour search, covariances, voxel-map build, cost launches, the align state machine, fitness, communicator set-up, profiling.
// (a section of the host translation unit: included by fvh_capi.hip inside its anonymous namespace, after the sections it builds on;
//  split out in round 6 -- the single file had grown to 3,100 lines)

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
    const size_t pinned_max = (size_t)e->params.pinned_upload_max;
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
      const size_t zero_copy_max = (size_t)e->params.zerocopy_upload_max;
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
  const int items_env = e->params.sort_items;
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
  const int sort_mode = e->params.sort_mode;  // 0: multi-kernel radix, 1: single workgroup, 2: cooperative (single-engine processes), 3: cooperative always
  if (sort_mode >= 1 && n <= SORT_SMALL_MAX) {
    // cooperative kernel (32 workgroups meeting at grid barriers) while no OTHER handle has a gang kernel in flight (GangRegistry above:
    // two gang kernels from two streams could starve each other of CU slots; the watchdog + fallback would recover, slowly)
    const bool coop = c.has_box && (sort_mode == 3 ? e->gang_begin(false) : (sort_mode == 2 && e->gang_begin(true)));  // (mode 1: never -- other PROCESSES' gang kernels on a shared GPU are invisible to the registry)
    struct GangEnd { Engine* e; bool on; ~GangEnd() { if (on) e->gang_end(); } } gang_end{e, coop};  // (on every way out: the event behind whatever was queued)
    // Mode 2, refused (another handle's gang kernel holds CU slots right now): NOT the one-workgroup sort -- 1,024 threads x 120 VGPRs fit on no CU
    // that hosts LM workgroups, it waits in the queue until one of those kernels ends (kernel trace, HISTORY.md) -- but the radix passes below,
    // whose 256-thread workgroups are placed anywhere (four concurrent handles: 5,710 -> 6,310 registrations/s). Not on a multi-GPU handle: the
    // ranks cut the SAME Morton order into tiles, and the passes sort by a finer key than the cooperative / one-workgroup sorts (two handles
    // of one process, one refused and one not, would disagree about the tiles).
    if (coop || sort_mode != 2 || e->sharded()) {
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
      const unsigned long long wd = fvh_env_ull("FVH_SORT_COOP_WATCHDOG_TICKS", e->params.sort_coop_watchdog_ticks);  // (the environment overrides the handle's value per call: test hook, 0 forces the fallback)
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
  const int fused_bits = e->params.sort_fused_bits;  // 0: the four-launch passes below. (9: the sort is 8 us shorter and the exact k-NN behind it 14 us longer -- coarser cells, looser tiles)
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
  const int two_pass_max = e->params.sort_two_pass_max;
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
  peer_wait_kernel<<<1, 64, 0, e->stream>>>(pv, gen, e->params.peer_watchdog_ticks, pc.err.as<int>());
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
  static const int knn_mode = (int)fvh_env_ll("FVH_KNN_MODE", 1);
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
    const int nf_max = e->params.knn_nearest_first_max_points;
    // one query = one wave = one WORKGROUP: a 4-wave workgroup holds its four slots until its slowest query is done (queries take
    // 8 us on average, 12.6 at the 90th percentile), single-wave workgroups hand each slot back at once: 39.7 -> 37 us at 17k points,
    // 161 -> 154 us at 100k (FVH_KNN_BLOCK=256 / 128: the old shapes, for A/B runs)
    const int knn_block = e->params.knn_block;
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
    if (!sharded && e->precision != FVH_COMPUTE_CUDA_COMPAT && coherent_order(c, e->params.coherent_min_points)) {
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
    static const int rbf_mode = (int)fvh_env_ll("FVH_RBF_MODE", 1);
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
        if (!sharded && coherent_order(c, e->params.coherent_min_points)) { HIP_OR_FAIL(e, c.cov_sorted.ensure(sizeof(float4) * 2 * (size_t)c.n)); cov_sorted = c.cov_sorted.as<float4>(); }
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
                                                                           coherent_order(c, e->params.coherent_min_points), shard ? vm.region.as<VmRegion>() : nullptr, compat ? 1 : 0);
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
      const int bitmap_min = e->params.bitmap_min_points;
      const size_t bitmap_bytes = (size_t)e->params.bitmap_max_bytes;
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
  const long long target_items = e->params.cost_target_items;
  const int max_blocks = e->params.cost_max_blocks;
  const int group_max = e->params.cost_group_max;  // the kernel keeps one item's lookups in flight together: at most COST_CH
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
  const int split_on = e->params.cost_split;
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
    P.peer_watchdog_ticks = e->params.peer_watchdog_ticks;
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
    const int prio = e->params.cost_prio;  // A/B knob: 0 never, 1 always (default -1: by the rule below)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device);
    P.prio_from = cus;
    P.prio_mode = prio >= 0 ? prio : ((persistent && blocks > cus && blocks <= 2 * cus) ? 1 : 0);
    // The LM step on EVERY workgroup (each polls the group rows itself: no broadcast hand-off) for grids of at most two workgroups per
    // CU: 17k headline 126.1 -> 123.5 us, NDT LiDAR frames 81.8 -> 79.0 us. With three per CU (100k / 1M points: 768 workgroups) the
    // redundant steps cost more than the hand-off they replace (210.5 -> 216 us, 223 -> 227 us): those keep the collectors' broadcast.
    // FVH_LM_EVERYWHERE: 0 never, 1 (default) by this rule, 2 always. (profiles/r04_lm_everywhere.txt)
    const int everywhere = e->params.lm_everywhere;
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
    P.watchdog_ticks = fvh_env_ull("FVH_PERSIST_WATCHDOG_TICKS", e->params.persist_watchdog_ticks);  // (the environment overrides the handle's value per call: test hook, 0 forces the abort + fallback path)
    // multi-GPU: workgroup 0 may legitimately wait for a late peer (up to the peer watchdog); the collectors waiting for its all-reduced
    // row and the workgroups waiting for their broadcast must outlast that, or a 50 ms skew between ranks would look like a stuck local barrier
    if (P.peer.n > 1 && P.watchdog_ticks) P.watchdog_ticks = std::max(P.watchdog_ticks, 2 * P.peer_watchdog_ticks);
    const int zc = e->params.zerocopy_result;
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
  hipEvent_t const fence = e->crowd_fence;  // (valid for this align only, whatever becomes of it)
  e->crowd_fence = nullptr;
  e->lm_crowds_chip = false;
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
  const int persist_env = e->params.persistent;
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
    e->lm_crowds_chip = c.persistent && Engine::crowds(c.plan.nb, std::max(cap, 1));
  }
  {
    if (c.persistent && e->lm_crowds_chip && fence && hipEventQuery(fence) != hipSuccess) {  // a preparation still runs on the second stream: this grid waits for it (Engine::lm_crowds_chip)
      (void)hipGetLastError();
      HIP_OR_FAIL(e, hipStreamWaitEvent(e->stream, fence, 0));
    }
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
  struct Done { AlignCtx& c; Engine* e; ~Done() { c.release_slots(); c.active = false; e->lm_crowds_chip = false; } } done{c, e};
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
      const bool block = e->params.host_wait_block != 0;  // sleep in hipStreamSynchronize instead of spinning a core on the result word
      if (block) (void)hipStreamSynchronize(e->stream);
      // (the stream is only asked now and then -- it answers "drained" when a launch ended without its result word, i.e. aborted: every query
      // takes the runtime's lock, which concurrent aligns of other host threads also need for their launches)
      const unsigned long long query_mask = [&] { unsigned long long m = 1; while (m < e->params.result_query_spins) m <<= 1; return m - 1; }();
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
  nn1_rows_kernel<<<(src.n + 15) / 16, 256, 0, e->stream>>>(src.sorted.as<float4>(), src.n, tgt.sorted.as<float4>(), tgt.bbox.as<float4>(), tgt.bbox2.as<float4>(), tgt.n, T12, thr_sq, corr, best_out, lm,
                                                            e->params.nn1_seed ? tgt.pts.as<float4>() : nullptr);
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
  static const int fit_mode = (int)fvh_env_ll("FVH_FIT_MODE", 1);
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

