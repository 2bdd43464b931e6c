// ORACLE (test infrastructure, NOT product code): plain C ABI over vgicp_oracle.hpp for
// ctypes (oracle/oracle.py).  Used only by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.
#include <cstring>
#include <chrono>

#include "vgicp_oracle.hpp"

using namespace orc;

namespace {
std::shared_ptr<Cloud> make_cloud(const float* xyz, int n) {
  auto c = std::make_shared<Cloud>();
  c->xyz.assign(xyz, xyz + (size_t)3 * n);
  return c;
}
void copy_m3(const std::vector<M3>& v, double* out) { for (size_t i = 0; i < v.size(); i++) std::memcpy(out + 9 * i, v[i].m, 9 * sizeof(double)); }
std::vector<M3> load_m3(const double* in, size_t n) { std::vector<M3> v(n); for (size_t i = 0; i < n; i++) std::memcpy(v[i].m, in + 9 * i, 9 * sizeof(double)); return v; }
int dump_voxelmap(const VoxelMap& vm, int* coords, int* num, double* means, double* covs) {
  for (size_t i = 0; i < vm.voxels.size(); i++) {
    coords[3 * i] = vm.coords[i].x; coords[3 * i + 1] = vm.coords[i].y; coords[3 * i + 2] = vm.coords[i].z;
    num[i] = vm.voxels[i].num_points;
    for (int a = 0; a < 3; a++) means[3 * i + a] = vm.voxels[i].mean[a];
    std::memcpy(covs + 9 * i, vm.voxels[i].cov.m, 9 * sizeof(double));
  }
  return (int)vm.voxels.size();
}
// test leg "oracle fed the engine's own fp32 voxel data": replaces a voxel map by the given records (insertion order = given order)
void load_voxelmap(VoxelMap& vm, int nv, const int* coords, const int* num, const double* means, const double* covs) {
  vm.index.clear(); vm.coords.clear(); vm.voxels.clear();
  for (int i = 0; i < nv; i++) {
    VoxelKey k{coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
    Voxel v;
    v.num_points = num[i];
    for (int a = 0; a < 3; a++) v.mean[a] = means[3 * i + a];
    std::memcpy(v.cov.m, covs + 9 * i, 9 * sizeof(double));
    vm.index[k] = (int)vm.voxels.size();
    vm.coords.push_back(k);
    vm.voxels.push_back(v);
  }
}
}  // namespace

extern "C" {

struct orc_result {
  double T[16];
  double H[36];
  int converged, nr_iterations, num_linearize, num_error_evals;
};

int orc_load_pcd(const char* path, float* out_xyz, int max_n) {
  Cloud c;
  if (!load_pcd(path, c)) return -1;
  int n = (int)c.size();
  if (out_xyz && n <= max_n) std::memcpy(out_xyz, c.xyz.data(), c.xyz.size() * sizeof(float));
  return n;
}
int orc_remove_origin(float* xyz, int n) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  remove_origin_points(c);
  std::memcpy(xyz, c.xyz.data(), c.xyz.size() * sizeof(float));
  return (int)c.size();
}
int orc_approx_voxelgrid(const float* xyz, int n, float leaf, float* out) {
  Cloud c, o; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  approximate_voxel_grid(c, leaf, o);
  std::memcpy(out, o.xyz.data(), o.xyz.size() * sizeof(float));
  return (int)o.size();
}
int orc_voxelgrid(const float* xyz, int n, float leaf, float* out) {
  Cloud c, o; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  voxel_grid(c, leaf, o);
  std::memcpy(out, o.xyz.data(), o.xyz.size() * sizeof(float));
  return (int)o.size();
}
void orc_knn(const float* xyz, int n, int k, int threads, int* idx) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  std::vector<int> v;
  knn_all(c, k, threads, v);
  std::memcpy(idx, v.data(), v.size() * sizeof(int));
}
void orc_knn_query(const float* xyz, int n, const float* q, int nq, int k, int* idx, float* sq) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  KdTree t(c);
#pragma omp parallel for schedule(guided, 8)
  for (int i = 0; i < nq; i++) t.knn(q + 3 * i, k, idx + (size_t)i * k, sq ? sq + (size_t)i * k : nullptr);
}
void orc_covariances_knn(const float* xyz, int n, int k, const int* idx, int reg, int threads, double* covs9) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  std::vector<int> v;
  if (idx) v.assign(idx, idx + (size_t)n * k); else knn_all(c, k, threads, v);
  std::vector<M3> covs;
  covariances_from_neighbors(c, k, v, (RegularizationMethod)reg, threads, covs);
  copy_m3(covs, covs9);
}
void orc_covariances_rbf(const float* xyz, int n, double w, double maxd, int reg, int threads, double* covs9) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  std::vector<M3> covs;
  covariances_rbf(c, w, maxd, (RegularizationMethod)reg, threads, covs);
  copy_m3(covs, covs9);
}
void orc_regularize(const double* cov9, int reg, double* out9) {
  M3 a; std::memcpy(a.m, cov9, sizeof(a.m));
  M3 r = regularize(a, (RegularizationMethod)reg);
  std::memcpy(out9, r.m, sizeof(r.m));
}
int orc_voxelmap_vgicp(const float* xyz, const double* covs9, int n, double res, int* coords, int* num, double* means, double* covs) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  VoxelMap vm(res);
  vm.create_vgicp(c, load_m3(covs9, n), 0);
  return dump_voxelmap(vm, coords, num, means, covs);
}
int orc_voxelmap_ndt(const float* xyz, int n, double res, int* coords, int* num, double* means, double* covs) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  VoxelMap vm(res);
  vm.create_ndt(c);
  return dump_voxelmap(vm, coords, num, means, covs);
}
int orc_neighbor_offsets(int method, double radius, int* out, int max_n) {
  auto o = neighbor_offsets((NeighborSearchMethod)method, radius);
  for (size_t i = 0; i < o.size() && (int)i < max_n; i++) { out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z; }
  return (int)o.size();
}
void orc_se3_exp(const double* a, double* T16) { iso_to_rowmajor16(se3_exp(a), T16); }
double orc_fitness(const float* src, int ns, const float* tgt, int nt, const double* T16) {
  Cloud s, t; s.xyz.assign(src, src + (size_t)3 * ns); t.xyz.assign(tgt, tgt + (size_t)3 * nt);
  KdTree tree(t);
  return fitness_score(s, tree, iso_from_rowmajor16(T16));
}

// ---------------- FastVGICP handle ----------------
void* orc_vgicp_create() { return new FastVGICP(); }
void orc_vgicp_destroy(void* h) { delete (FastVGICP*)h; }
void orc_vgicp_set_gicp_mode(void* h, int on, double max_corr_dist) {
  auto* g = static_cast<FastVGICP*>(h);
  g->gicp_mode = on != 0;
  g->max_correspondence_distance = max_corr_dist;
}
void orc_vgicp_set_params(void* h, int threads, int k, int reg, double res, int search, int cov_mode, double kw, double kmax, int round_fp32) {
  auto* g = (FastVGICP*)h;
  if (threads > 0) g->num_threads = threads;
  g->k_correspondences = k; g->regularization = (RegularizationMethod)reg; g->voxel_resolution = res;
  g->search_method = (NeighborSearchMethod)search; g->cov_mode = cov_mode; g->kernel_width = kw; g->kernel_max_dist = kmax;
  g->round_storage_fp32 = round_fp32 != 0;
}
void orc_vgicp_set_lm(void* h, int max_iter, double rot_eps, double trans_eps, int lm_max_iter, double init_lambda_factor, int debug) {
  auto* g = (FastVGICP*)h;
  g->max_iterations = max_iter; g->rotation_epsilon = rot_eps; g->transformation_epsilon = trans_eps;
  g->lm_max_iterations = lm_max_iter; g->lm_init_lambda_factor = init_lambda_factor; g->lm_debug_print = debug != 0;
}
void orc_vgicp_set_optimizer(void* h, int gauss_newton) { ((FastVGICP*)h)->gauss_newton = gauss_newton != 0; }
void orc_ndt_set_optimizer(void* h, int gauss_newton) { ((NDT*)h)->gauss_newton = gauss_newton != 0; }
void orc_vgicp_set_target(void* h, const float* xyz, int n) { ((FastVGICP*)h)->setInputTarget(make_cloud(xyz, n)); }
void orc_vgicp_set_source(void* h, const float* xyz, int n) { ((FastVGICP*)h)->setInputSource(make_cloud(xyz, n)); }
void orc_vgicp_set_target_covs(void* h, const double* c9) { auto* g = (FastVGICP*)h; g->target_covs = load_m3(c9, g->target->size()); }
void orc_vgicp_set_source_covs(void* h, const double* c9) { auto* g = (FastVGICP*)h; g->source_covs = load_m3(c9, g->input->size()); }
void orc_vgicp_swap(void* h) { ((FastVGICP*)h)->swapSourceAndTarget(); }
void orc_vgicp_clear_source(void* h) { ((FastVGICP*)h)->clearSource(); }
void orc_vgicp_clear_target(void* h) { ((FastVGICP*)h)->clearTarget(); }
// what align() does ahead of the optimiser, so linearize/compute_error can be probed at fixed poses
void orc_vgicp_prepare(void* h) {
  auto* g = (FastVGICP*)h;
  if (g->target_cloud_updated) { g->pcl_tree.reset(new KdTree(*g->target)); g->target_cloud_updated = false; }
  g->voxelmap.reset();
  if (g->source_covs.size() != g->input->size()) g->calculate_covariances(g->input, *g->search_source, g->source_covs);
  if (g->target_covs.size() != g->target->size()) g->calculate_covariances(g->target, *g->search_target, g->target_covs);
}
double orc_vgicp_linearize(void* h, const double* T16, double* H36, double* b6) { return ((FastVGICP*)h)->linearize(iso_from_rowmajor16(T16), H36, b6); }
double orc_vgicp_compute_error(void* h, const double* T16) { return ((FastVGICP*)h)->compute_error(iso_from_rowmajor16(T16)); }
int orc_vgicp_num_correspondences(void* h) { return (int)((FastVGICP*)h)->voxel_correspondences.size(); }
// index-level parity (tests): every correspondence as {source index, 0, 0, 0, target voxel x, y, z} -- voxels by COORDINATE, so that
// the list does not depend on anybody's voxel numbering; FastGICP mode: {source index, 0, 0, 0, target point index, 0, 0}
int orc_vgicp_get_correspondences(void* h, int* out7) {
  auto* g = (FastVGICP*)h;
  for (size_t n = 0; n < g->voxel_correspondences.size(); n++) {
    const auto& c = g->voxel_correspondences[n];
    int* o = out7 + 7 * n;
    o[0] = c.first; o[1] = o[2] = o[3] = 0;
    if (g->gicp_mode) { o[4] = c.second; o[5] = o[6] = 0; }
    else { const VoxelKey& k = g->voxelmap->coords[c.second]; o[4] = k.x; o[5] = k.y; o[6] = k.z; }
  }
  return (int)g->voxel_correspondences.size();
}
int orc_vgicp_num_voxels(void* h) { auto* g = (FastVGICP*)h; return g->voxelmap ? (int)g->voxelmap->voxels.size() : -1; }
void orc_vgicp_get_covs(void* h, int which, double* out) { auto* g = (FastVGICP*)h; copy_m3(which ? g->target_covs : g->source_covs, out); }
int orc_vgicp_get_voxelmap(void* h, int* coords, int* num, double* means, double* covs) { return dump_voxelmap(*((FastVGICP*)h)->voxelmap, coords, num, means, covs); }
int orc_voxelmap_vgicp_mode(const float* xyz, const double* covs9, int n, double res, int mode, int* coords, int* num, double* means, double* covs) {
  Cloud c; c.xyz.assign(xyz, xyz + (size_t)3 * n);
  VoxelMap vm(res);
  vm.create_vgicp(c, load_m3(covs9, n), mode);
  return dump_voxelmap(vm, coords, num, means, covs);
}
double orc_vgicp_cuda_compat_sums(void* h, const double* T16, double* H36, double* b6) { return ((FastVGICP*)h)->cuda_compat_sums(iso_from_rowmajor16(T16), H36, b6); }
void orc_vgicp_set_voxel_mode(void* h, int mode) { auto* g = (FastVGICP*)h; g->voxel_mode = mode; g->voxelmap.reset(); }
void orc_vgicp_set_voxelmap(void* h, int nv, const int* coords, const int* num, const double* means, const double* covs) {
  auto* g = (FastVGICP*)h;
  g->voxelmap.reset(new VoxelMap(g->voxel_resolution));  // (align() resets it again: this is for linearize / compute_error at fixed poses)
  load_voxelmap(*g->voxelmap, nv, coords, num, means, covs);
}
void orc_vgicp_align(void* h, const double* guess16, orc_result* r) {
  auto* g = (FastVGICP*)h;
  g->align(iso_from_rowmajor16(guess16));
  iso_to_rowmajor16(g->final_transformation, r->T);
  std::memcpy(r->H, g->final_hessian, sizeof(r->H));
  r->converged = g->converged; r->nr_iterations = g->nr_iterations; r->num_linearize = g->num_linearize; r->num_error_evals = g->num_error_evals;
}
double orc_vgicp_fitness(void* h) { return ((FastVGICP*)h)->getFitnessScore(); }

// The metric loops of src/align.cpp:51-104 on the oracle (the "reference OpenMP FastVGICP"
// stand-in timed as cpu_baseline).  mode: 0 = single, 1 = N-times, 2 = N-times reuse.
// Returns wall milliseconds.
double orc_vgicp_bench(void* h, const float* tgt, int nt, const float* src, int ns, int mode, int loops, double* fitness) {
  auto* g = (FastVGICP*)h;
  CloudPtr target = make_cloud(tgt, nt), source = make_cloud(src, ns);
  Iso3 I = iso_identity();
  auto t1 = std::chrono::steady_clock::now();
  if (mode == 0) {
    g->clearTarget(); g->clearSource(); g->setInputTarget(target); g->setInputSource(source); g->align(I);
  } else if (mode == 1) {
    for (int i = 0; i < loops; i++) { g->clearTarget(); g->clearSource(); g->setInputTarget(target); g->setInputSource(source); g->align(I); }
  } else {
    CloudPtr t_ = target, s_ = source;
    for (int i = 0; i < loops; i++) {
      g->swapSourceAndTarget(); g->clearSource();
      g->setInputTarget(t_); g->setInputSource(s_); g->align(I);
      t_.swap(s_);
    }
  }
  auto t2 = std::chrono::steady_clock::now();
  if (fitness) *fitness = (mode == 0) ? g->getFitnessScore() : 0.0;
  return std::chrono::duration<double, std::milli>(t2 - t1).count();
}

// ---------------- NDT handle ----------------
void* orc_ndt_create() { return new NDT(); }
void orc_ndt_destroy(void* h) { delete (NDT*)h; }
void orc_ndt_set_params(void* h, int threads, double res, int mode, int search, double radius) {
  auto* g = (NDT*)h;
  if (threads > 0) g->num_threads = threads;
  g->resolution = res; g->distance_mode = (NDTDistanceMode)mode; g->search_method = (NeighborSearchMethod)search; g->search_radius = radius;
}
void orc_ndt_set_voxelmap(void* h, int which, int nv, const int* coords, const int* num, const double* means, const double* covs) {
  auto* g = (NDT*)h;
  auto& slot = which ? g->target_voxelmap : g->source_voxelmap;
  slot.reset(new VoxelMap(g->resolution));
  load_voxelmap(*slot, nv, coords, num, means, covs);
}
void orc_ndt_set_round_fp32(void* h, int on) { ((NDT*)h)->round_storage_fp32 = on != 0; }
void orc_ndt_set_lm(void* h, int max_iter, double rot_eps, double trans_eps, int lm_max_iter, double init_lambda_factor) {
  auto* g = (NDT*)h;
  g->max_iterations = max_iter; g->rotation_epsilon = rot_eps; g->transformation_epsilon = trans_eps;
  g->lm_max_iterations = lm_max_iter; g->lm_init_lambda_factor = init_lambda_factor;
}
void orc_ndt_set_target(void* h, const float* xyz, int n) { ((NDT*)h)->setInputTarget(make_cloud(xyz, n)); }
void orc_ndt_set_source(void* h, const float* xyz, int n) { ((NDT*)h)->setInputSource(make_cloud(xyz, n)); }
void orc_ndt_swap(void* h) { ((NDT*)h)->swapSourceAndTarget(); }
void orc_ndt_prepare(void* h) { auto* g = (NDT*)h; if (g->target_cloud_updated) { g->pcl_tree.reset(new KdTree(*g->target)); g->target_cloud_updated = false; } g->create_voxelmaps(); }
double orc_ndt_linearize(void* h, const double* T16, double* H36, double* b6) { return ((NDT*)h)->linearize(iso_from_rowmajor16(T16), H36, b6); }
double orc_ndt_compute_error(void* h, const double* T16) { return ((NDT*)h)->compute_error(iso_from_rowmajor16(T16)); }
int orc_ndt_num_correspondences(void* h) { return (int)((NDT*)h)->correspondences.size(); }
// {source element, source voxel x, y, z (D2D; P2D: zeros), target voxel x, y, z}
int orc_ndt_get_correspondences(void* h, int* out7) {
  auto* g = (NDT*)h;
  for (size_t n = 0; n < g->correspondences.size(); n++) {
    const auto& c = g->correspondences[n];
    int* o = out7 + 7 * n;
    o[0] = c.first; o[1] = o[2] = o[3] = 0;
    if (g->distance_mode == D2D) { const VoxelKey& s = g->source_voxelmap->coords[c.first]; o[1] = s.x; o[2] = s.y; o[3] = s.z; }
    const VoxelKey& k = g->target_voxelmap->coords[c.second];
    o[4] = k.x; o[5] = k.y; o[6] = k.z;
  }
  return (int)g->correspondences.size();
}
int orc_ndt_get_voxelmap(void* h, int which, int* coords, int* num, double* means, double* covs) {
  auto* g = (NDT*)h;
  return dump_voxelmap(which ? *g->target_voxelmap : *g->source_voxelmap, coords, num, means, covs);
}
void orc_ndt_align(void* h, const double* guess16, orc_result* r) {
  auto* g = (NDT*)h;
  g->align(iso_from_rowmajor16(guess16));
  iso_to_rowmajor16(g->final_transformation, r->T);
  std::memcpy(r->H, g->final_hessian, sizeof(r->H));
  r->converged = g->converged; r->nr_iterations = g->nr_iterations; r->num_linearize = g->num_linearize; r->num_error_evals = g->num_error_evals;
}
double orc_ndt_fitness(void* h) { return ((NDT*)h)->getFitnessScore(); }

}  // extern "C"
