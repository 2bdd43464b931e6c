// FastGICP (nearest-target-point correspondences) on the engine.
// (a section of the host translation unit: included by fvh_capi.hip inside its anonymous namespace, after the sections it builds on;
//  split out in round 6 -- the single file had grown to 3,100 lines)

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
    static const int nn_mode = (int)fvh_env_ll("FVH_GICP_NN_MODE", 1);
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

