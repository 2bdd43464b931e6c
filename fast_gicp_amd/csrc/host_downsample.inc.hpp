// Voxel-grid downsampling (pcl::VoxelGrid / pcl::ApproximateVoxelGrid) on the engine.
// (a section of the host translation unit: included by fvh_capi.hip inside its anonymous namespace, after the sections it builds on;
//  split out in round 6 -- the single file had grown to 3,100 lines)

// ---------------------------------------------------------------------------------------------
// voxel-grid downsampling (kernels_downsample.hpp)
// ---------------------------------------------------------------------------------------------
struct DownsampleDev {
  CloudDev cloud;                 // the input, widened to float4
  DevBuf keys, idx, head, trig, scan, bsums, slots, out;
  DevBuf out4;                    // ApproximateVoxelGrid: the output again as float4 {x, y, z, 0}; fvh_ndt_*_from_voxelgrid SWAPS this buffer with the handle's cloud
  bool out4_valid = false;        // holds the output of the last call (not yet taken)
  hipStream_t out_stream = nullptr;  // the stream the last call ran on ...
  bool out_complete = true;          // ... and whether the host saw its LAST kernel finish (false: an _async call returned with the count only)
  int out_n = 0;
  void release() { cloud.release(); keys.release(); idx.release(); head.release(); trig.release(); scan.release(); bsums.release(); slots.release(); out.release(); out4.release(); }
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
  HIP_OR_FAIL(e, d.out4.ensure(sizeof(float4) * (size_t)n));  // (after a swap this is a cloud buffer of the registration handle: grown here if the frame is larger)
  d.out4_valid = false;
  const bool fresh = d.slots.p == nullptr;
  HIP_OR_FAIL(e, d.slots.ensure(sizeof(AvgState)));
  if (fresh) HIP_OR_FAIL(e, hipMemsetAsync(d.slots.p, 0, sizeof(AvgState), e->stream));  // (the ticket re-arms itself afterwards)
  unsigned* hist = d.keys.as<unsigned>();
  unsigned* totals = hist + (size_t)AVG_SLOTS * nwaves;
  unsigned* hist_wg = totals + AVG_SLOTS;
  // up to AVG_FUSED_MAX_POINTS points: four launches -- the scatter and the emit kernel recompute the two small prefix sums themselves
  // (kernels_downsample.hpp); FVH_AVG_FUSED=0 keeps round 2's six for A/B runs
  const bool fused_on = e->params.avg_fused != 0;
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
      avg_emit_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, head, n, inv, d.trig.as<unsigned>(), nullptr, st, d.out.as<float>(), final_result, seq, nwords, early ? e->result_dev : nullptr, d.out4.as<float>());
    } else {
      avg_keys_hist_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, d.trig.as<unsigned>(), nwords, st);
      avg_binscan_kernel<<<AVG_SLOTS / 4, 256, 0, e->stream>>>(hist, nwaves, totals, st);
      avg_scatter_kernel<<<wblocks, 256, 0, e->stream>>>(d_xyz, n, stride, inv, nwaves, hist, st, sorted);
      avg_mark_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, n, inv, head, d.trig.as<unsigned>(), st);
      avg_scan_kernel<<<1, 1024, 0, e->stream>>>(d.trig.as<unsigned>(), nwords, d.scan.as<unsigned>(), st, early ? e->result_dev : nullptr, seq);
      avg_emit_kernel<<<pblocks, 256, 0, e->stream>>>(sorted, head, n, inv, d.trig.as<unsigned>(), d.scan.as<unsigned>(), st, d.out.as<float>(), final_result, seq, 0, nullptr, d.out4.as<float>());
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
  d.out4_valid = d.out_n > 0;
  d.out_stream = e->stream;
  d.out_complete = !early;
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
  d.out4_valid = false;
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

