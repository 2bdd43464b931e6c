// The binding a maintainer of koide3/fast_gicp adds to run FastVGICPCuda / NDTCuda on the MI355X engine:
// fast_gicp::cuda::FastVGICPCudaCore and fast_gicp::cuda::NDTCudaCore implemented on the C ABI of fast_vgicp_hip.h.
// Replaces src/fast_gicp/cuda/*.cu (fast_vgicp_cuda.cu, ndt_cuda.cu, brute_force_knn.cu, covariance_estimation*.cu,
// covariance_regularization.cu, gaussian_voxelmap.cu, find_voxel_correspondences.cu, compute_derivatives.cu,
// compute_mahalanobis.cu, ndt_compute_derivatives.cu); built with g++, links libfast_vgicp_hip.so.
// Eigen layouts the ABI was designed for: Vector3f = 3 packed floats, Matrix3f = 9 floats column-major,
// Isometry3d::data() = 4x4 double column-major, Matrix<double,6,6>::data() column-major (symmetric), Matrix<double,6,1>.
// Checked in this repository by tests/test_integration_shim_cpu.py (syntax + every call against the real ABI header).
#include <stdexcept>
#include <string>

#include <fast_gicp/cuda/fast_vgicp_cuda.cuh>
#include <fast_gicp/cuda/ndt_cuda.cuh>
#include <fast_vgicp_hip.h>

namespace fast_gicp {
namespace cuda {

namespace {
inline void ok(const fvh_vgicp* h, int rc, const char* what) {
  if (rc != FVH_OK) throw std::runtime_error(std::string(what) + ": " + fvh_vgicp_last_error(h));
}
inline void ok(const fvh_ndt* h, int rc, const char* what) {
  if (rc != FVH_OK) throw std::runtime_error(std::string(what) + ": " + fvh_ndt_last_error(h));
}
static_assert(sizeof(Eigen::Vector3f) == 3 * sizeof(float) && sizeof(Eigen::Matrix3f) == 9 * sizeof(float), "packed Eigen layouts");
// the reference's enum class ordinals ARE the ABI's integers (gicp_settings.hpp:7-11, ndt_settings.hpp:6)
static_assert((int)RegularizationMethod::PLANE == FVH_REG_PLANE && (int)RegularizationMethod::FROBENIUS == FVH_REG_FROBENIUS && (int)RegularizationMethod::NONE == FVH_REG_NONE, "regularisation ordinals");
static_assert((int)NeighborSearchMethod::DIRECT27 == FVH_DIRECT27 && (int)NeighborSearchMethod::DIRECT1 == FVH_DIRECT1 && (int)NeighborSearchMethod::DIRECT_RADIUS == FVH_DIRECT_RADIUS, "search ordinals");
static_assert((int)NDTDistanceMode::P2D == FVH_NDT_P2D && (int)NDTDistanceMode::D2D == FVH_NDT_D2D, "NDT ordinals");
}  // namespace

// ---------------------------------------------------------------- FastVGICPCudaCore (fast_vgicp_cuda.cu)
FastVGICPCudaCore::FastVGICPCudaCore() {
  if (fvh_vgicp_create(0, &h_) != FVH_OK) throw std::runtime_error("FastVGICPCudaCore: no MI355X / engine could not be created");
}
FastVGICPCudaCore::~FastVGICPCudaCore() { fvh_vgicp_destroy(h_); }

void FastVGICPCudaCore::set_resolution(double resolution) { ok(h_, fvh_vgicp_set_resolution(h_, resolution), "set_resolution"); }
void FastVGICPCudaCore::set_kernel_params(double kernel_width, double kernel_max_dist) { ok(h_, fvh_vgicp_set_kernel_params(h_, kernel_width, kernel_max_dist), "set_kernel_params"); }
void FastVGICPCudaCore::set_neighbor_search_method(fast_gicp::NeighborSearchMethod method, double radius) {
  ok(h_, fvh_vgicp_set_neighbor_search_method(h_, static_cast<int>(method), radius), "set_neighbor_search_method");
}
void FastVGICPCudaCore::swap_source_and_target() { ok(h_, fvh_vgicp_swap_source_and_target(h_), "swap_source_and_target"); }
void FastVGICPCudaCore::set_source_cloud(const CloudF& cloud) { ok(h_, fvh_vgicp_set_source_cloud(h_, cloud.empty() ? nullptr : cloud[0].data(), (int)cloud.size()), "set_source_cloud"); }
void FastVGICPCudaCore::set_target_cloud(const CloudF& cloud) { ok(h_, fvh_vgicp_set_target_cloud(h_, cloud.empty() ? nullptr : cloud[0].data(), (int)cloud.size()), "set_target_cloud"); }
void FastVGICPCudaCore::set_source_neighbors(int k, const std::vector<int>& neighbors) { ok(h_, fvh_vgicp_set_source_neighbors(h_, k, neighbors.data()), "set_source_neighbors"); }
void FastVGICPCudaCore::set_target_neighbors(int k, const std::vector<int>& neighbors) { ok(h_, fvh_vgicp_set_target_neighbors(h_, k, neighbors.data()), "set_target_neighbors"); }
void FastVGICPCudaCore::find_source_neighbors(int k) { ok(h_, fvh_vgicp_find_source_neighbors(h_, k), "find_source_neighbors"); }
void FastVGICPCudaCore::find_target_neighbors(int k) { ok(h_, fvh_vgicp_find_target_neighbors(h_, k), "find_target_neighbors"); }
void FastVGICPCudaCore::calculate_source_covariances(RegularizationMethod method) { ok(h_, fvh_vgicp_calculate_source_covariances(h_, static_cast<int>(method)), "calculate_source_covariances"); }
void FastVGICPCudaCore::calculate_target_covariances(RegularizationMethod method) { ok(h_, fvh_vgicp_calculate_target_covariances(h_, static_cast<int>(method)), "calculate_target_covariances"); }
void FastVGICPCudaCore::calculate_source_covariances_rbf(RegularizationMethod method) { ok(h_, fvh_vgicp_calculate_source_covariances_rbf(h_, static_cast<int>(method)), "calculate_source_covariances_rbf"); }
void FastVGICPCudaCore::calculate_target_covariances_rbf(RegularizationMethod method) { ok(h_, fvh_vgicp_calculate_target_covariances_rbf(h_, static_cast<int>(method)), "calculate_target_covariances_rbf"); }

void FastVGICPCudaCore::get_source_covariances(CovsF& covs) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_source_points(h_, &n), "get_num_source_points");
  covs.resize(n);
  if (n) ok(h_, fvh_vgicp_get_source_covariances(h_, covs[0].data()), "get_source_covariances");
}
void FastVGICPCudaCore::get_target_covariances(CovsF& covs) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_target_points(h_, &n), "get_num_target_points");
  covs.resize(n);
  if (n) ok(h_, fvh_vgicp_get_target_covariances(h_, covs[0].data()), "get_target_covariances");
}
void FastVGICPCudaCore::get_voxel_num_points(std::vector<int>& num_points) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_voxels(h_, &n), "get_num_voxels");
  num_points.resize(n);
  if (n) ok(h_, fvh_vgicp_get_voxel_num_points(h_, num_points.data()), "get_voxel_num_points");
}
void FastVGICPCudaCore::get_voxel_means(CloudF& means) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_voxels(h_, &n), "get_num_voxels");
  means.resize(n);
  if (n) ok(h_, fvh_vgicp_get_voxel_means(h_, means[0].data()), "get_voxel_means");
}
void FastVGICPCudaCore::get_voxel_covs(CovsF& covs) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_voxels(h_, &n), "get_num_voxels");
  covs.resize(n);
  if (n) ok(h_, fvh_vgicp_get_voxel_covs(h_, covs[0].data()), "get_voxel_covs");
}
void FastVGICPCudaCore::get_voxel_correspondences(std::vector<std::pair<int, int>>& correspondences) const {
  int n = 0;
  ok(h_, fvh_vgicp_get_num_correspondences(h_, &n), "get_num_correspondences");
  correspondences.resize(n);
  static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int), "pair<int,int> is two packed ints");
  if (n) ok(h_, fvh_vgicp_get_voxel_correspondences(h_, &correspondences[0].first), "get_voxel_correspondences");
}
void FastVGICPCudaCore::create_target_voxelmap() { ok(h_, fvh_vgicp_create_target_voxelmap(h_), "create_target_voxelmap"); }
void FastVGICPCudaCore::update_correspondences(const Eigen::Isometry3d& trans) { ok(h_, fvh_vgicp_update_correspondences(h_, trans.data()), "update_correspondences"); }
double FastVGICPCudaCore::compute_error(const Eigen::Isometry3d& trans, Hessian* H, Gradient* b) const {
  double e = 0.0;
  const bool deriv = H && b;  // the reference passes both or neither (fast_vgicp_cuda_impl.hpp:170-178)
  ok(h_, fvh_vgicp_compute_error(h_, trans.data(), deriv ? H->data() : nullptr, deriv ? b->data() : nullptr, &e), "compute_error");
  return e;
}

// ---------------------------------------------------------------- NDTCudaCore (ndt_cuda.cu)
NDTCudaCore::NDTCudaCore() {
  if (fvh_ndt_create(0, &h_) != FVH_OK) throw std::runtime_error("NDTCudaCore: no MI355X / engine could not be created");
}
NDTCudaCore::~NDTCudaCore() { fvh_ndt_destroy(h_); }
void NDTCudaCore::set_distance_mode(fast_gicp::NDTDistanceMode mode) { ok(h_, fvh_ndt_set_distance_mode(h_, static_cast<int>(mode)), "set_distance_mode"); }
void NDTCudaCore::set_resolution(double resolution) { ok(h_, fvh_ndt_set_resolution(h_, resolution), "set_resolution"); }
void NDTCudaCore::set_neighbor_search_method(fast_gicp::NeighborSearchMethod method, double radius) {
  ok(h_, fvh_ndt_set_neighbor_search_method(h_, static_cast<int>(method), radius), "set_neighbor_search_method");
}
void NDTCudaCore::swap_source_and_target() { ok(h_, fvh_ndt_swap_source_and_target(h_), "swap_source_and_target"); }
void NDTCudaCore::set_source_cloud(const CloudF& cloud) { ok(h_, fvh_ndt_set_source_cloud(h_, cloud.empty() ? nullptr : cloud[0].data(), (int)cloud.size()), "set_source_cloud"); }
void NDTCudaCore::set_target_cloud(const CloudF& cloud) { ok(h_, fvh_ndt_set_target_cloud(h_, cloud.empty() ? nullptr : cloud[0].data(), (int)cloud.size()), "set_target_cloud"); }
void NDTCudaCore::create_voxelmaps() { ok(h_, fvh_ndt_create_voxelmaps(h_), "create_voxelmaps"); }
void NDTCudaCore::create_target_voxelmap() { ok(h_, fvh_ndt_create_target_voxelmap(h_), "create_target_voxelmap"); }
void NDTCudaCore::create_source_voxelmap() { ok(h_, fvh_ndt_create_source_voxelmap(h_), "create_source_voxelmap"); }
void NDTCudaCore::update_correspondences(const Eigen::Isometry3d& trans) { ok(h_, fvh_ndt_update_correspondences(h_, trans.data()), "update_correspondences"); }
double NDTCudaCore::compute_error(const Eigen::Isometry3d& trans, Hessian* H, Gradient* b) const {
  double e = 0.0;
  const bool deriv = H && b;
  ok(h_, fvh_ndt_compute_error(h_, trans.data(), deriv ? H->data() : nullptr, deriv ? b->data() : nullptr, &e), "compute_error");
  return e;
}

}  // namespace cuda
}  // namespace fast_gicp
