// gicp_align on the MI355X engine -- the reference's benchmark driver (src/align.cpp:51-215) for the
// methods this engine provides (the fgicp, ndt_cuda and vgicp_cuda rows), same call sequence, same
// "single / 100times / 100times_reuse / fitness_score" output line (README.md:118-134).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "fast_gicp_amd/pcd_io.hpp"
#include "fast_gicp_amd/registration.hpp"
#include "fast_gicp_amd/voxelgrid.hpp"

using namespace fast_gicp;
using Cloud = PointCloud<PointXYZ>;

template <typename Registration>
void test(Registration& reg, const Cloud::ConstPtr& target, const Cloud::ConstPtr& source) {  // align.cpp:51-104
  Cloud aligned;
  auto t1 = std::chrono::high_resolution_clock::now();
  reg.clearTarget();
  reg.clearSource();
  reg.setInputTarget(target);
  reg.setInputSource(source);
  reg.align(aligned);
  auto t2 = std::chrono::high_resolution_clock::now();
  const double fitness_score = reg.getFitnessScore();
  std::cout << "single:" << std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() / 1e6 << "[msec] " << std::flush;

  t1 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 100; i++) {
    reg.clearTarget();
    reg.clearSource();
    reg.setInputTarget(target);
    reg.setInputSource(source);
    reg.align(aligned);
  }
  t2 = std::chrono::high_resolution_clock::now();
  std::cout << "100times:" << std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() / 1e6 << "[msec] " << std::flush;

  t1 = std::chrono::high_resolution_clock::now();
  Cloud::ConstPtr target_ = target, source_ = source;
  std::vector<double> per_iter;
  for (int i = 0; i < 100; i++) {
    const auto ti = std::chrono::high_resolution_clock::now();
    reg.swapSourceAndTarget();
    reg.clearSource();
    reg.setInputTarget(target_);
    reg.setInputSource(source_);
    reg.align(aligned);
    target_.swap(source_);
    per_iter.push_back(std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - ti).count());
  }
  t2 = std::chrono::high_resolution_clock::now();
  if (std::getenv("GICP_ALIGN_PER_ITER")) { for (int i = 0; i < 100; i += 1) std::cout << (int)per_iter[i] << " "; std::cout << std::endl; }
  std::cout << "100times_reuse:" << std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() / 1e6 << "[msec] fitness_score:" << fitness_score << std::endl;

  if (std::getenv("GICP_ALIGN_BREAKDOWN")) {  // where a registration of the reuse loop spends its host time (not part of the reference's output)
    using clk = std::chrono::steady_clock;
    double t_swap = 0, t_src = 0, t_align = 0;
    auto& ht = fast_gicp::detail::host_timing();
    ht = {};
    ht.on = true;
    for (int i = 0; i < 100; i++) {
      auto a = clk::now();
      reg.swapSourceAndTarget();
      reg.clearSource();
      reg.setInputTarget(target_);
      auto b = clk::now();
      reg.setInputSource(source_);
      auto c = clk::now();
      reg.align(aligned);
      auto d = clk::now();
      target_.swap(source_);
      t_swap += std::chrono::duration<double, std::micro>(b - a).count();
      t_src += std::chrono::duration<double, std::micro>(c - b).count();
      t_align += std::chrono::duration<double, std::micro>(d - c).count();
    }
    std::cout << "  breakdown per registration [us]: swap + clearSource + setInputTarget " << t_swap / 100 << ", setInputSource (upload, k-NN / covariance launches) " << t_src / 100
              << ", align (waits for the GPU, transforms the output cloud) " << t_align / 100 << " (of it: optimisation " << ht.optimize_us / 100 << ", output cloud " << ht.transform_us / 100 << ")" << std::endl;
    ht.on = false;
  }
}

// Not in the reference's output: the same 100 registrations of the reuse loop through the class's pipeline calls -- the next scan (a host
// cloud, as everywhere in this driver) is handed over, sorted, searched and its covariances computed while the LM kernel of the current pair runs.
template <typename Registration>
void test_pipelined(Registration& reg, const Cloud::ConstPtr& target, const Cloud::ConstPtr& source) {
  reg.clearTarget();
  reg.clearSource();
  reg.setInputTarget(target);
  reg.setInputSource(source);
  Cloud aligned;
  reg.align(aligned);
  Cloud::ConstPtr next = target;  // the cloud that becomes the source next = the target of now
  Cloud::ConstPtr other = source;
  for (int warm = 0; warm < 2; warm++) {  // (two rounds so that both clouds have been through the prepared slot)
    reg.alignAsync();
    reg.prepareNextSource(next);
    reg.alignWait();
    reg.swapSourceAndTarget();
    reg.adoptPreparedSource();
    next.swap(other);
  }
  const auto t1 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 100; i++) {
    reg.alignAsync();
    reg.prepareNextSource(next);
    reg.alignWait();
    reg.swapSourceAndTarget();
    reg.adoptPreparedSource();
    next.swap(other);
  }
  const auto t2 = std::chrono::high_resolution_clock::now();
  std::cout << "100times_reuse_pipelined:" << std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() / 1e6 << "[msec] converged:" << reg.hasConverged() << std::endl;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cout << "usage: gicp_align target_pcd source_pcd" << std::endl;
    return 0;
  }
  auto target_cloud = std::make_shared<Cloud>(), source_cloud = std::make_shared<Cloud>();
  if (loadPCDFile(argv[1], *target_cloud)) { std::cerr << "failed to open " << argv[1] << std::endl; return 1; }
  if (loadPCDFile(argv[2], *source_cloud)) { std::cerr << "failed to open " << argv[2] << std::endl; return 1; }
  // remove invalid points around origin (align.cpp:127-133)
  auto near_origin = [](const PointXYZ& p) { return (double)(p.x * p.x + p.y * p.y + p.z * p.z) < 1e-3; };
  source_cloud->points.erase(std::remove_if(source_cloud->points.begin(), source_cloud->points.end(), near_origin), source_cloud->points.end());
  target_cloud->points.erase(std::remove_if(target_cloud->points.begin(), target_cloud->points.end(), near_origin), target_cloud->points.end());
  // downsampling (align.cpp:136-147)
  auto ft = std::make_shared<Cloud>(), fs = std::make_shared<Cloud>();
  approximate_voxel_grid(*target_cloud, 0.1f, *ft);
  approximate_voxel_grid(*source_cloud, 0.1f, *fs);
  std::cout << "target:" << ft->size() << "[pts] source:" << fs->size() << "[pts]" << std::endl;

  std::cout << "--- fgicp_hip ---" << std::endl;  // align.cpp:171-178 (fgicp_st / fgicp_mt): nearest-point GICP, correspondences and sums on the device
  {
    FastGICP<PointXYZ, PointXYZ> fgicp;
    test(fgicp, ft, fs);
  }
  std::cout << "--- ndt_hip (P2D) ---" << std::endl;
  NDTCuda<PointXYZ, PointXYZ> ndt;
  ndt.setResolution(1.0);
  ndt.setDistanceMode(NDTDistanceMode::P2D);
  test(ndt, ft, fs);
  std::cout << "--- ndt_hip (D2D) ---" << std::endl;
  ndt.setDistanceMode(NDTDistanceMode::D2D);
  test(ndt, ft, fs);

  std::cout << "--- vgicp_hip (parallel_kdtree) ---" << std::endl;
  FastVGICPCuda<PointXYZ, PointXYZ> vgicp;
  vgicp.setResolution(1.0);
  test(vgicp, ft, fs);
  std::cout << "--- vgicp_hip (gpu_bruteforce) ---" << std::endl;
  vgicp.setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_BRUTEFORCE);
  test(vgicp, ft, fs);
  std::cout << "--- vgicp_hip (gpu_rbf_kernel) ---" << std::endl;
  vgicp.setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_RBF_KERNEL);
  vgicp.setKernelWidth(0.5);
  test(vgicp, ft, fs);
  std::cout << "--- vgicp_hip (gpu_bruteforce, DIRECT27) ---" << std::endl;
  vgicp.setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_BRUTEFORCE);
  vgicp.setNeighborSearchMethod(NeighborSearchMethod::DIRECT27);
  test(vgicp, ft, fs);
  test_pipelined(vgicp, ft, fs);
  std::cout << "--- vgicp_hip (gpu_rbf_kernel, DIRECT27) ---" << std::endl;
  vgicp.setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_RBF_KERNEL);
  test(vgicp, ft, fs);
  test_pipelined(vgicp, ft, fs);
  return 0;
}
