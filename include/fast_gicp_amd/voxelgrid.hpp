// pcl::ApproximateVoxelGrid<PointXYZ> restated for the host layer (the reference's callers downsample
// with it: src/align.cpp:136-147, src/python/main.cpp:46-62,80-92). PCL is third party and not
// available here; this follows its published algorithm: a 512-slot history hashed by
// (ix*7171 + iy*3079 + iz*4231) & 511, a slot holding a different voxel is flushed (fp32 centroid
// emitted) before reuse, the remaining slots are flushed in slot order. Pinned by README.md:116.
#pragma once
#include <cmath>
#include <vector>

#include "registration.hpp"

namespace fast_gicp {

template <typename PointT>
inline void approximate_voxel_grid(const PointCloud<PointT>& in, float leaf, PointCloud<PointT>& out) {
  constexpr int kSlots = 512;
  struct Slot { int ix = 0, iy = 0, iz = 0, count = 0; float sx = 0, sy = 0, sz = 0; };
  std::vector<Slot> slots(kSlots);
  const float inv = 1.0f / leaf;
  out.points.clear();
  out.points.reserve(in.size());
  auto flush = [&](Slot& s) {
    PointT p{};
    const float n = static_cast<float>(s.count);
    p.x = s.sx / n; p.y = s.sy / n; p.z = s.sz / n;
    out.points.push_back(p);
    s.count = 0; s.sx = s.sy = s.sz = 0.f;
  };
  for (const auto& p : in.points) {
    const int ix = static_cast<int>(std::floor(p.x * inv)), iy = static_cast<int>(std::floor(p.y * inv)), iz = static_cast<int>(std::floor(p.z * inv));
    Slot& s = slots[static_cast<unsigned>((ix * 7171 + iy * 3079 + iz * 4231) & (kSlots - 1))];
    if (s.count && (s.ix != ix || s.iy != iy || s.iz != iz)) flush(s);
    s.ix = ix; s.iy = iy; s.iz = iz;
    s.count++;
    s.sx += p.x; s.sy += p.y; s.sz += p.z;
  }
  for (auto& s : slots)
    if (s.count) flush(s);
}

}  // namespace fast_gicp
