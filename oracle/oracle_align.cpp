// ORACLE (test infrastructure, NOT product code): the `gicp_align` benchmark driver of the
// reference (src/align.cpp:51-104,107-215) run on the CPU restatement, printing the same
// "single / 100times / 100times_reuse / fitness_score" line format as README.md:118-134.
#include <omp.h>

#include <chrono>
#include <cstring>
#include <iostream>

#include "vgicp_oracle.hpp"

using namespace orc;

template <typename Reg>
static void test(Reg& reg, const CloudPtr& target, const CloudPtr& source, int loops) {
  Iso3 I = iso_identity();
  auto t1 = std::chrono::steady_clock::now();
  reg.clearTarget(); reg.clearSource(); reg.setInputTarget(target); reg.setInputSource(source); reg.align(I);
  auto t2 = std::chrono::steady_clock::now();
  double fitness = reg.getFitnessScore();
  std::cout << "single:" << std::chrono::duration<double, std::milli>(t2 - t1).count() << "[msec] " << std::flush;
  t1 = std::chrono::steady_clock::now();
  for (int i = 0; i < loops; i++) { reg.clearTarget(); reg.clearSource(); reg.setInputTarget(target); reg.setInputSource(source); reg.align(I); }
  t2 = std::chrono::steady_clock::now();
  std::cout << loops << "times:" << std::chrono::duration<double, std::milli>(t2 - t1).count() << "[msec] " << std::flush;
  t1 = std::chrono::steady_clock::now();
  CloudPtr t_ = target, s_ = source;
  for (int i = 0; i < loops; i++) {
    reg.swapSourceAndTarget(); reg.clearSource();
    reg.setInputTarget(t_); reg.setInputSource(s_); reg.align(I);
    t_.swap(s_);
  }
  t2 = std::chrono::steady_clock::now();
  std::cout << loops << "times_reuse:" << std::chrono::duration<double, std::milli>(t2 - t1).count() << "[msec] fitness_score:" << fitness << std::endl;
  std::cout << "  iterations:" << reg.nr_iterations + 1 << " linearize:" << reg.num_linearize << " error_evals:" << reg.num_error_evals << " converged:" << reg.converged << std::endl;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::cout << "usage: oracle_align target_pcd source_pcd [loops=100] [--no-origin-filter]" << std::endl; return 0; }
  int loops = argc > 3 ? std::atoi(argv[3]) : 100;
  bool origin_filter = !(argc > 4 && std::strcmp(argv[4], "--no-origin-filter") == 0);
  auto target = std::make_shared<Cloud>(), source = std::make_shared<Cloud>();
  if (!load_pcd(argv[1], *target)) { std::cerr << "failed to open " << argv[1] << std::endl; return 1; }
  if (!load_pcd(argv[2], *source)) { std::cerr << "failed to open " << argv[2] << std::endl; return 1; }
  if (origin_filter) { remove_origin_points(*source); remove_origin_points(*target); }
  auto ft = std::make_shared<Cloud>(), fs = std::make_shared<Cloud>();
  approximate_voxel_grid(*target, 0.1f, *ft);
  approximate_voxel_grid(*source, 0.1f, *fs);
  std::cout << "target:" << ft->size() << "[pts] source:" << fs->size() << "[pts]" << std::endl;

  std::cout << "--- vgicp_st ---" << std::endl;
  FastVGICP vgicp;
  vgicp.voxel_resolution = 1.0;
  vgicp.num_threads = 1;
  test(vgicp, ft, fs, loops);
  std::cout << "--- vgicp_mt (" << omp_get_max_threads() << " threads) ---" << std::endl;
  vgicp.num_threads = omp_get_max_threads();
  test(vgicp, ft, fs, loops);
  std::cout << "--- vgicp_mt DIRECT27 ---" << std::endl;
  vgicp.search_method = DIRECT27;
  test(vgicp, ft, fs, loops);
  std::cout << "--- ndt (D2D, fp64 restatement of NDTCuda) ---" << std::endl;
  NDT ndt;
  test(ndt, ft, fs, loops);
  std::cout << "--- ndt (P2D) ---" << std::endl;
  ndt.distance_mode = P2D;
  test(ndt, ft, fs, loops);
  return 0;
}
