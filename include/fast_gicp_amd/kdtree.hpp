// Exact k-NN on the host for NearestNeighborMethod::CPU_PARALLEL_KDTREE -- stands in for
// pcl::search::KdTree / FLANN in FastVGICPCuda::find_neighbors_parallel_kdtree
// (impl/fast_vgicp_cuda_impl.hpp:152-167). Node-less median kd-tree: the points' permutation IS the
// tree (the median of every index range is its node), only the split axis per node is stored. The
// points are copied once in tree order so that a leaf scan reads consecutive memory, and a query
// keeps its candidate list on the stack (a heap allocation per query serialised the OpenMP loop).
// Distances are fp32 ((dx*dx + dy*dy) + dz*dz) with ties to the lower index, the same total order the
// device k-NN uses, so both neighbour methods yield identical index sets.
#pragma once
#include <algorithm>
#include <cfloat>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace fast_gicp {
namespace host {

/// at most one thread per ~512 queries, never more than 32 (or the OpenMP default)
inline int omp_threads_for(int n) {
#ifdef _OPENMP
  return std::max(1, std::min(std::min(omp_get_max_threads(), 32), n / 512));
#else
  (void)n;
  return 1;
#endif
}

class KdTree {
public:
  KdTree(const float* xyz, int n) : xyz_(xyz), n_(n), perm_(n), axis_(n, 0), sorted_(3 * (size_t)n) {
    std::iota(perm_.begin(), perm_.end(), 0);
    // the two halves of a node are independent: OpenMP tasks down to kTaskMin points (round 5 built the tree on one thread: 2-3 of the 4.7 ms
    // a 17k-point cloud spent in this class)
    if (n > kTaskMin && omp_threads_for(n) > 1) {
#pragma omp parallel num_threads(omp_threads_for(n))
#pragma omp single nowait
      build(0, n);
    } else if (n > 0) {
      build(0, n);
    }
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) sorted_[3 * (size_t)i + a] = xyz[3 * (size_t)perm_[i] + a];
  }

  /// k nearest neighbours of q (ascending (distance, index)); out has k entries (-1 padded if n < k)
  void knn(const float* q, int k, int* out) const {
    constexpr int kStack = 64;
    float dstack[kStack];
    int istack[kStack];
    std::vector<float> dheap;
    std::vector<int> iheap;
    float* d = dstack;
    int* id = istack;
    if (k > kStack) { dheap.resize(k); iheap.resize(k); d = dheap.data(); id = iheap.data(); }
    Best best{k, 0, d, id};
    if (n_ > 0) search(0, n_, q, best);
    for (int j = 0; j < k; j++) out[j] = j < best.count ? best.i[j] : -1;
  }

private:
  struct Best {
    int k, count;
    float* d;
    int* i;
    bool full() const { return count == k; }
    float worst() const { return d[count - 1]; }
    void offer(float dist, int idx) {  // sorted insertion, k is small (20)
      if (count == k && !(dist < d[k - 1] || (dist == d[k - 1] && idx < i[k - 1]))) return;
      int pos = count < k ? count : k - 1;
      while (pos > 0 && (dist < d[pos - 1] || (dist == d[pos - 1] && idx < i[pos - 1]))) {
        d[pos] = d[pos - 1];
        i[pos] = i[pos - 1];
        pos--;
      }
      d[pos] = dist;
      i[pos] = idx;
      if (count < k) count++;
    }
  };
  static float sqdist(const float* a, const float* b) {
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
  }

  void build(int lo, int hi) {
    if (hi - lo <= kLeaf) return;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; i++)
      for (int a = 0; a < 3; a++) {
        const float v = xyz_[3 * (size_t)perm_[i] + a];
        mn[a] = std::min(mn[a], v);
        mx[a] = std::max(mx[a], v);
      }
    int ax = 0;
    if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
    if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
    const int mid = lo + (hi - lo) / 2;
    std::nth_element(perm_.begin() + lo, perm_.begin() + mid, perm_.begin() + hi, [&](int a, int b) { return xyz_[3 * (size_t)a + ax] < xyz_[3 * (size_t)b + ax]; });
    axis_[mid] = (unsigned char)ax;
    if (hi - lo > kTaskMin) {
#pragma omp task default(shared) firstprivate(lo, mid)
      build(lo, mid);
      build(mid + 1, hi);
#pragma omp taskwait
    } else {
      build(lo, mid);
      build(mid + 1, hi);
    }
  }

  void search(int lo, int hi, const float* q, Best& best) const {
    if (hi - lo <= kLeaf) {
      for (int i = lo; i < hi; i++) best.offer(sqdist(&sorted_[3 * (size_t)i], q), perm_[i]);
      return;
    }
    const int mid = lo + (hi - lo) / 2;
    const int ax = axis_[mid];
    const float* pmid = &sorted_[3 * (size_t)mid];
    best.offer(sqdist(pmid, q), perm_[mid]);
    const float diff = q[ax] - pmid[ax];
    if (diff < 0) {
      search(lo, mid, q, best);
      if (!best.full() || diff * diff <= best.worst()) search(mid + 1, hi, q, best);
    } else {
      search(mid + 1, hi, q, best);
      if (!best.full() || diff * diff <= best.worst()) search(lo, mid, q, best);
    }
  }

  static constexpr int kLeaf = 8;
  static constexpr int kTaskMin = 2048;  // ranges below this are built by the thread that reached them
  const float* xyz_;
  int n_;
  std::vector<int> perm_;
  std::vector<unsigned char> axis_;
  std::vector<float> sorted_;  // the points in tree order (xyz of perm_[i] at 3 i)
};

}  // namespace host
}  // namespace fast_gicp
