// gicp_kitti on the MI355X engine -- the reference's odometry driver (src/kitti.cpp:70-153) without the PCL visualiser:
// frames %06d.bin (x, y, z, intensity as float32, kitti.cpp:22-69) -> ApproximateVoxelGrid 0.25 ON THE DEVICE (the raw
// xyzi buffer goes to the GPU as it is) -> setInputSource -> align -> swapSourceAndTarget -> pose accumulation; prints the
// running frame rate and writes the trajectory in KITTI format (12 values per line, kitti.cpp:141-153).
//   usage: gicp_kitti /path/to/sequences/00/velodyne [ndt|ndt_pipelined|vgicp|vgicp_pipelined|gicp] [trajectory.txt]
// ndt_pipelined: the same loop as a two-stage pipeline -- while the LM kernel of frame k runs, frame k+1 is read, filtered on the handle's
// second stream and its voxel map built there (NDTCuda::alignAsync / prepareNextSourceDevice / adoptPreparedSource / alignWait).
#include <chrono>
#include <cstdio>
#include <deque>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "fast_gicp_amd/registration.hpp"

using namespace fast_gicp;
using Cloud = PointCloud<PointXYZ>;

class KittiLoader {  // kitti.cpp:22-69
public:
  explicit KittiLoader(const std::string& dataset_path) : dataset_path_(dataset_path) {
    for (num_frames_ = 0;; num_frames_++) {
      FILE* f = std::fopen(filename(num_frames_).c_str(), "rb");
      if (!f) break;
      std::fclose(f);
    }
    if (num_frames_ == 0) std::cerr << "error: no files in " << dataset_path << std::endl;
  }
  size_t size() const { return num_frames_; }
  // raw xyzi floats of frame i (the reference reads at most 1,000,000 floats per frame)
  std::vector<float> frame(size_t i) const {
    std::vector<float> buffer(1000000);
    FILE* file = std::fopen(filename(i).c_str(), "rb");
    if (!file) { std::cerr << "error: failed to load " << filename(i) << std::endl; return {}; }
    const size_t n = std::fread(buffer.data(), sizeof(float), buffer.size(), file) / 4;
    std::fclose(file);
    buffer.resize(n * 4);
    return buffer;
  }

private:
  std::string filename(size_t i) const { char name[32]; std::snprintf(name, sizeof(name), "/%06zu.bin", i); return dataset_path_ + name; }
  size_t num_frames_ = 0;
  std::string dataset_path_;
};

static Cloud::Ptr downsample_on_device(fvh_voxelgrid* vg, const std::vector<float>& xyzi, float leaf) {
  int n = 0;
  detail::check(fvh_voxelgrid_filter_strided(vg, FVH_VOXELGRID_APPROXIMATE, xyzi.data(), (int)(xyzi.size() / 4), 4, leaf, &n), "fvh_voxelgrid_filter_strided", fvh_voxelgrid_last_error(vg));
  std::vector<float> xyz((size_t)3 * n);
  detail::check(fvh_voxelgrid_get_points(vg, xyz.data()), "fvh_voxelgrid_get_points", fvh_voxelgrid_last_error(vg));
  auto cloud = std::make_shared<Cloud>();
  cloud->points.resize(n);
  for (int i = 0; i < n; i++) { cloud->points[i].x = xyz[3 * i]; cloud->points[i].y = xyz[3 * i + 1]; cloud->points[i].z = xyz[3 * i + 2]; }
  return cloud;
}

static void write_trajectory(const std::string& traj_path, const std::vector<Isometry3d>& poses) {  // kitti.cpp:141-153
  std::ofstream ofs(traj_path);
  ofs.precision(9);
  for (const auto& pose : poses) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++) {
        if (i || j) ofs << " ";
        ofs << (j < 3 ? pose.R[i * 3 + j] : pose.t[i]);
      }
    ofs << std::endl;
  }
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "usage: gicp_kitti /your/kitti/path/sequences/00/velodyne [ndt|ndt_pipelined|vgicp|vgicp_pipelined|gicp] [trajectory.txt]" << std::endl;
    return 0;
  }
  const std::string method = argc > 2 ? argv[2] : "gicp";  // the reference's default is FastGICP (kitti.cpp:85)
  const std::string traj_path = argc > 3 ? argv[3] : "/tmp/traj.txt";
  KittiLoader kitti(argv[1]);
  if (kitti.size() == 0) return 1;

  const float downsample_resolution = 0.25f;  // kitti.cpp:79
  fvh_voxelgrid* vg = nullptr;
  detail::check(fvh_voxelgrid_create(0, &vg), "fvh_voxelgrid_create", "cannot create the HIP engine (no GPU? there is no CPU fallback)");

  std::shared_ptr<LsqRegistration<PointXYZ, PointXYZ>> reg;
  if (method == "ndt") {
    auto ndt = std::make_shared<NDTCuda<PointXYZ, PointXYZ>>();
    ndt->setResolution(1.0);  // kitti.cpp:89-90
    reg = ndt;
  } else if (method == "vgicp") {
    auto vgicp = std::make_shared<FastVGICPCuda<PointXYZ, PointXYZ>>();
    vgicp->setResolution(1.0);
    vgicp->setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_BRUTEFORCE);
    reg = vgicp;
  } else {
    auto gicp = std::make_shared<FastGICP<PointXYZ, PointXYZ>>();
    gicp->setMaxCorrespondenceDistance(1.0);  // kitti.cpp:92
    reg = gicp;
  }

  if (method == "ndt_pipelined") {
    NDTCuda<PointXYZ, PointXYZ> ndt;
    ndt.setResolution(1.0);
    ndt.setInputTarget(downsample_on_device(vg, kitti.frame(0), downsample_resolution));
    detail::check(fvh_voxelgrid_share_prepare_stream_with_ndt(vg, ndt.core()), "share_prepare_stream", fvh_voxelgrid_last_error(vg));
    auto prepare = [&](size_t i) {  // read + filter + widen + voxel map of frame i, all on the handle's second stream
      const std::vector<float> xyzi = kitti.frame(i);
      int n = 0;
      const float* d_xyz = nullptr;
      detail::check(fvh_voxelgrid_filter_strided(vg, FVH_VOXELGRID_APPROXIMATE, xyzi.data(), (int)(xyzi.size() / 4), 4, downsample_resolution, &n), "fvh_voxelgrid_filter_strided", fvh_voxelgrid_last_error(vg));
      (void)d_xyz;
      ndt.prepareNextSourceFromFilter(vg);  // (the filter's float4 output becomes the cloud: no widening kernel)
    };
    std::vector<Isometry3d> poses(kitti.size());
    poses[0] = Isometry3d::Identity();
    std::deque<std::chrono::high_resolution_clock::time_point> stamps;
    stamps.push_back(std::chrono::high_resolution_clock::now());
    if (kitti.size() > 1) prepare(1);
    for (size_t i = 1; i < kitti.size(); i++) {
      ndt.adoptPreparedSource();
      ndt.alignAsync();
      if (i + 1 < kitti.size()) prepare(i + 1);  // beside the running LM kernel
      poses[i] = poses[i - 1] * Isometry3d::from(ndt.alignWait());
      ndt.swapSourceAndTarget();
      stamps.push_back(std::chrono::high_resolution_clock::now());
      if (stamps.size() > 30) stamps.pop_front();
      std::cout << stamps.size() / (std::chrono::duration_cast<std::chrono::nanoseconds>(stamps.back() - stamps.front()).count() / 1e9) << "fps" << std::endl;
    }
    detail::check(fvh_voxelgrid_share_stream_with_ndt(vg, nullptr), "share_stream", fvh_voxelgrid_last_error(vg));
    fvh_voxelgrid_destroy(vg);
    write_trajectory(traj_path, poses);
    return 0;
  }

  if (method == "vgicp_pipelined") {  // the same loop with FastVGICPCuda (kitti.cpp:88, the commented alternative): order, neighbours, covariances of scan k+1 beside the LM kernel of scan k
    FastVGICPCuda<PointXYZ, PointXYZ> vgicp;
    vgicp.setResolution(1.0);
    vgicp.setNearestNeighborSearchMethod(NearestNeighborMethod::GPU_BRUTEFORCE);
    vgicp.setInputTarget(downsample_on_device(vg, kitti.frame(0), downsample_resolution));
    detail::check(fvh_voxelgrid_share_prepare_stream_with_vgicp(vg, vgicp.core()), "share_prepare_stream", fvh_voxelgrid_last_error(vg));
    auto prepare = [&](size_t i) {
      const std::vector<float> xyzi = kitti.frame(i);
      int n = 0;
      const float* d_xyz = nullptr;
      detail::check(fvh_voxelgrid_filter_strided(vg, FVH_VOXELGRID_APPROXIMATE, xyzi.data(), (int)(xyzi.size() / 4), 4, downsample_resolution, &n), "fvh_voxelgrid_filter_strided", fvh_voxelgrid_last_error(vg));
      detail::check(fvh_voxelgrid_device_points(vg, &d_xyz, &n), "fvh_voxelgrid_device_points", fvh_voxelgrid_last_error(vg));
      vgicp.prepareNextSourceDevice(d_xyz, n, 3);
    };
    std::vector<Isometry3d> poses(kitti.size());
    poses[0] = Isometry3d::Identity();
    std::deque<std::chrono::high_resolution_clock::time_point> stamps;
    stamps.push_back(std::chrono::high_resolution_clock::now());
    if (kitti.size() > 1) { prepare(1); vgicp.adoptPreparedSource(); }
    for (size_t i = 1; i < kitti.size(); i++) {
      vgicp.alignAsync();
      if (i + 1 < kitti.size()) prepare(i + 1);  // beside the running LM kernel
      poses[i] = poses[i - 1] * Isometry3d::from(vgicp.alignWait());
      vgicp.swapSourceAndTarget();
      if (i + 1 < kitti.size()) vgicp.adoptPreparedSource();
      stamps.push_back(std::chrono::high_resolution_clock::now());
      if (stamps.size() > 30) stamps.pop_front();
      std::cout << stamps.size() / (std::chrono::duration_cast<std::chrono::nanoseconds>(stamps.back() - stamps.front()).count() / 1e9) << "fps" << std::endl;
    }
    detail::check(fvh_voxelgrid_share_stream_with_vgicp(vg, nullptr), "share_stream", fvh_voxelgrid_last_error(vg));
    fvh_voxelgrid_destroy(vg);
    write_trajectory(traj_path, poses);
    return 0;
  }

  // set initial frame as target (kitti.cpp:95-99)
  reg->setInputTarget(downsample_on_device(vg, kitti.frame(0), downsample_resolution));
  std::vector<Isometry3d> poses(kitti.size());
  poses[0] = Isometry3d::Identity();
  std::deque<std::chrono::high_resolution_clock::time_point> stamps;  // boost::circular_buffer(30) in the reference
  stamps.push_back(std::chrono::high_resolution_clock::now());

  for (size_t i = 1; i < kitti.size(); i++) {
    // set the current frame as source, align, swap for the next registration (kitti.cpp:115-125)
    reg->setInputSource(downsample_on_device(vg, kitti.frame(i), downsample_resolution));
    Cloud aligned;
    reg->align(aligned);
    reg->swapSourceAndTarget();
    poses[i] = poses[i - 1] * Isometry3d::from(reg->getFinalTransformation());  // accumulate pose (kitti.cpp:128)
    stamps.push_back(std::chrono::high_resolution_clock::now());
    if (stamps.size() > 30) stamps.pop_front();
    std::cout << stamps.size() / (std::chrono::duration_cast<std::chrono::nanoseconds>(stamps.back() - stamps.front()).count() / 1e9) << "fps" << std::endl;
  }
  fvh_voxelgrid_destroy(vg);

  write_trajectory(traj_path, poses);
  return 0;
}
