// Test scaffolding: what the two seam classes look like once the maintainer has replaced their Thrust members by the
// engine handle (the public methods are the reference's: cuda/fast_vgicp_cuda.cuh:37-71, cuda/ndt_cuda.cuh:34-53).
#pragma once
#include <utility>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <fast_gicp/gicp/gicp_settings.hpp>
#include <fast_gicp/ndt/ndt_settings.hpp>

struct fvh_vgicp;
struct fvh_ndt;

namespace fast_gicp {
namespace cuda {

using CloudF = std::vector<Eigen::Vector3f, Eigen::aligned_allocator<Eigen::Vector3f>>;
using CovsF = std::vector<Eigen::Matrix3f, Eigen::aligned_allocator<Eigen::Matrix3f>>;
using Hessian = Eigen::Matrix<double, 6, 6>;
using Gradient = Eigen::Matrix<double, 6, 1>;

class FastVGICPCudaCore {
public:
  FastVGICPCudaCore();
  ~FastVGICPCudaCore();
  void set_resolution(double resolution);
  void set_kernel_params(double kernel_width, double kernel_max_dist);
  void set_neighbor_search_method(fast_gicp::NeighborSearchMethod method, double radius);
  void swap_source_and_target();
  void set_source_cloud(const CloudF& cloud);
  void set_target_cloud(const CloudF& cloud);
  void set_source_neighbors(int k, const std::vector<int>& neighbors);
  void set_target_neighbors(int k, const std::vector<int>& neighbors);
  void find_source_neighbors(int k);
  void find_target_neighbors(int k);
  void calculate_source_covariances(RegularizationMethod method);
  void calculate_target_covariances(RegularizationMethod method);
  void calculate_source_covariances_rbf(RegularizationMethod method);
  void calculate_target_covariances_rbf(RegularizationMethod method);
  void get_source_covariances(CovsF& covs) const;
  void get_target_covariances(CovsF& covs) const;
  void get_voxel_num_points(std::vector<int>& num_points) const;
  void get_voxel_means(CloudF& means) const;
  void get_voxel_covs(CovsF& covs) const;
  void get_voxel_correspondences(std::vector<std::pair<int, int>>& correspondences) const;
  void create_target_voxelmap();
  void update_correspondences(const Eigen::Isometry3d& trans);
  double compute_error(const Eigen::Isometry3d& trans, Hessian* H, Gradient* b) const;
  fvh_vgicp* handle() const { return h_; }  // new: for fvh_vgicp_align ("Optional fast path")

private:
  fvh_vgicp* h_ = nullptr;  // was: resolution, kernel params, offsets, eight thrust::device_vector members, GaussianVoxelMap
};

class NDTCudaCore {
public:
  NDTCudaCore();
  ~NDTCudaCore();
  void set_distance_mode(fast_gicp::NDTDistanceMode mode);
  void set_resolution(double resolution);
  void set_neighbor_search_method(fast_gicp::NeighborSearchMethod method, double radius);
  void swap_source_and_target();
  void set_source_cloud(const CloudF& cloud);
  void set_target_cloud(const CloudF& cloud);
  void create_voxelmaps();
  void create_target_voxelmap();
  void create_source_voxelmap();
  void update_correspondences(const Eigen::Isometry3d& trans);
  double compute_error(const Eigen::Isometry3d& trans, Hessian* H, Gradient* b) const;
  fvh_ndt* handle() const { return h_; }

private:
  fvh_ndt* h_ = nullptr;
};

}  // namespace cuda
}  // namespace fast_gicp
