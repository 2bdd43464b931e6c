// ORACLE (test infrastructure, NOT product code) -- see vgicp_oracle.hpp for the contract.
#include "vgicp_oracle.hpp"

#include <omp.h>

#include <array>
#include <cassert>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>

namespace orc {

// =====================================================================================
// I/O + preprocessing
// =====================================================================================

// Minimal PCD reader (pcl::io::loadPCDFile stand-in; align.cpp:118-125). Reads fields x,y,z
// (TYPE F SIZE 4) from DATA ascii|binary files.
bool load_pcd(const std::string& path, Cloud& out) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs) return false;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  size_t npoints = 0;
  std::string data_mode, line;
  while (std::getline(ifs, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS") { std::string f; while (ss >> f) fields.push_back(f); }
    else if (key == "SIZE") { int s; while (ss >> s) sizes.push_back(s); }
    else if (key == "COUNT") { int c; while (ss >> c) counts.push_back(c); }
    else if (key == "POINTS") { ss >> npoints; }
    else if (key == "DATA") { ss >> data_mode; break; }
  }
  if (fields.empty() || sizes.size() != fields.size()) return false;
  if (counts.empty()) counts.assign(fields.size(), 1);
  int off[3] = {-1, -1, -1}, fidx[3] = {-1, -1, -1};
  int stride = 0;
  for (size_t i = 0; i < fields.size(); i++) {
    for (int a = 0; a < 3; a++)
      if (fields[i] == std::string(1, "xyz"[a])) { off[a] = stride; fidx[a] = (int)i; }
    stride += sizes[i] * counts[i];
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) return false;
  out.xyz.resize(npoints * 3);
  if (data_mode == "binary") {
    std::vector<char> buf(npoints * stride);
    ifs.read(buf.data(), buf.size());
    if ((size_t)ifs.gcount() != buf.size()) return false;
    for (size_t i = 0; i < npoints; i++)
      for (int a = 0; a < 3; a++) std::memcpy(&out.xyz[3 * i + a], &buf[i * stride + off[a]], 4);
  } else if (data_mode == "ascii") {
    for (size_t i = 0; i < npoints; i++) {
      std::getline(ifs, line);
      std::istringstream ss(line);
      std::vector<double> vals;
      double v;
      while (ss >> v) vals.push_back(v);
      int col = 0;
      for (size_t f = 0; f < fields.size(); f++) {
        for (int a = 0; a < 3; a++)
          if ((int)f == fidx[a]) out.xyz[3 * i + a] = (float)vals[col];
        col += counts[f];
      }
    }
  } else {
    return false;
  }
  return true;
}

// align.cpp:127-133: erase points with squaredNorm() < 1e-3 (fp32 norm vs double literal)
void remove_origin_points(Cloud& c) {
  size_t w = 0;
  for (size_t i = 0; i < c.size(); i++) {
    const float* p = c.pt(i);
    float sq = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if ((double)sq < 1e-3) continue;
    if (w != i) std::memcpy(&c.xyz[3 * w], p, 12);
    w++;
  }
  c.xyz.resize(3 * w);
}

// pcl::ApproximateVoxelGrid<PointXYZ>::applyFilter restated (third-party; algorithm as
// published in PCL filters/impl/approximate_voxel_grid.hpp): 512-slot history hashed by
// (ix*7171 + iy*3079 + iz*4231) & 511; a slot holding a different voxel is flushed
// (centroid emitted) before reuse; remaining slots are flushed in slot order at the end.
// Pinned by README.md:116 (17,249 / 17,518 points on the bundled pair, no origin filter).
void approximate_voxel_grid(const Cloud& in, float leaf, Cloud& out) {
  const int histsize = 512;
  struct He { int ix, iy, iz, count; float c[3]; };
  std::vector<He> hist(histsize);
  for (auto& h : hist) { h.count = 0; h.c[0] = h.c[1] = h.c[2] = 0.f; h.ix = h.iy = h.iz = 0; }
  const float inv = 1.0f / leaf;
  out.xyz.clear();
  out.xyz.reserve(in.xyz.size());
  auto flush = [&](He& h) {
    float n = static_cast<float>(h.count);
    out.xyz.push_back(h.c[0] / n);
    out.xyz.push_back(h.c[1] / n);
    out.xyz.push_back(h.c[2] / n);
  };
  for (size_t i = 0; i < in.size(); i++) {
    const float* p = in.pt(i);
    int ix = static_cast<int>(std::floor(p[0] * inv));
    int iy = static_cast<int>(std::floor(p[1] * inv));
    int iz = static_cast<int>(std::floor(p[2] * inv));
    unsigned int hash = static_cast<unsigned int>((ix * 7171 + iy * 3079 + iz * 4231) & (histsize - 1));
    He& h = hist[hash];
    if (h.count && (ix != h.ix || iy != h.iy || iz != h.iz)) {
      flush(h);
      h.count = 0;
      h.c[0] = h.c[1] = h.c[2] = 0.f;
    }
    h.ix = ix; h.iy = iy; h.iz = iz;
    h.count++;
    h.c[0] += p[0]; h.c[1] += p[1]; h.c[2] += p[2];
  }
  for (int i = 0; i < histsize; i++)
    if (hist[i].count) flush(hist[i]);
}

// pcl::VoxelGrid<PointXYZ>::applyFilter restated (third-party): voxel index relative to the
// cloud's min corner, points grouped by index (ascending), centroid per group in fp32.
void voxel_grid(const Cloud& in, float leaf, Cloud& out) {
  out.xyz.clear();
  if (in.size() == 0) return;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = 0; i < in.size(); i++)
    for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], in.pt(i)[a]); mx[a] = std::max(mx[a], in.pt(i)[a]); }
  int64_t minb[3], maxb[3], divb[3];
  for (int a = 0; a < 3; a++) {
    minb[a] = static_cast<int>(std::floor(mn[a] * inv));
    maxb[a] = static_cast<int>(std::floor(mx[a] * inv));
    divb[a] = maxb[a] - minb[a] + 1;
  }
  std::vector<std::pair<int64_t, int>> iv(in.size());
  for (size_t i = 0; i < in.size(); i++) {
    const float* p = in.pt(i);
    int64_t i0 = static_cast<int>(std::floor(p[0] * inv) - static_cast<float>(minb[0]));
    int64_t i1 = static_cast<int>(std::floor(p[1] * inv) - static_cast<float>(minb[1]));
    int64_t i2 = static_cast<int>(std::floor(p[2] * inv) - static_cast<float>(minb[2]));
    iv[i] = {i0 + i1 * divb[0] + i2 * divb[0] * divb[1], (int)i};
  }
  std::stable_sort(iv.begin(), iv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  size_t s = 0;
  while (s < iv.size()) {
    size_t e = s;
    float c[3] = {0, 0, 0};
    while (e < iv.size() && iv[e].first == iv[s].first) {
      const float* p = in.pt(iv[e].second);
      c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
      e++;
    }
    float n = static_cast<float>(e - s);
    out.xyz.push_back(c[0] / n); out.xyz.push_back(c[1] / n); out.xyz.push_back(c[2] / n);
    s = e;
  }
}

// =====================================================================================
// kd-tree (exact k-NN)
// =====================================================================================
static inline float sqdist_f32(const float* a, const float* b) {
  // fp32, fixed association, no FMA (compiled with -ffp-contract=off): bit-identical to
  // the HIP brute-force kernel so neighbour SETS can be compared exactly.
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

KdTree::KdTree(const Cloud& c) : cloud(c) {
  order.resize(c.size());
  std::iota(order.begin(), order.end(), 0);
  nodes.reserve(c.size() / 4 + 16);
  if (c.size()) build(0, (int)c.size());
}

int KdTree::build(int begin, int end) {
  int id = (int)nodes.size();
  nodes.push_back(Node{-1, -1, begin, end, -1, 0.f});
  const int leaf_size = 10;
  if (end - begin <= leaf_size) return id;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = begin; i < end; i++)
    for (int a = 0; a < 3; a++) { float v = cloud.pt(order[i])[a]; mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v); }
  int dim = 0;
  for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[dim] - mn[dim]) dim = a;
  if (mx[dim] == mn[dim]) return id;  // all identical points: keep as a (large) leaf
  int mid = (begin + end) / 2;
  std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                   [&](int a, int b) { return cloud.pt(a)[dim] < cloud.pt(b)[dim]; });
  float split = cloud.pt(order[mid])[dim];
  nodes[id].dim = dim;
  nodes[id].split = split;
  int l = build(begin, mid);
  int r = build(mid, end);
  nodes[id].left = l;
  nodes[id].right = r;
  return id;
}

namespace {
struct Cand { float d; int i; };
inline bool cand_less(const Cand& a, const Cand& b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }
struct KnnHeap {  // bounded max-heap on (d, i)
  Cand* h; int k, n = 0;
  void push(Cand c) {
    if (n < k) {
      h[n++] = c;
      std::push_heap(h, h + n, cand_less);
    } else if (cand_less(c, h[0])) {
      std::pop_heap(h, h + n, cand_less);
      h[n - 1] = c;
      std::push_heap(h, h + n, cand_less);
    }
  }
  bool full() const { return n == k; }
  float worst() const { return h[0].d; }
};
void knn_rec(const KdTree& t, int node, const float* q, KnnHeap& heap) {
  const KdTree::Node& nd = t.nodes[node];
  if (nd.dim < 0) {
    for (int i = nd.begin; i < nd.end; i++) {
      int pi = t.order[i];
      heap.push(Cand{sqdist_f32(t.cloud.pt(pi), q), pi});
    }
    return;
  }
  float diff = q[nd.dim] - nd.split;
  int near = diff < 0 ? nd.left : nd.right;
  int far = diff < 0 ? nd.right : nd.left;
  knn_rec(t, near, q, heap);
  if (!heap.full() || diff * diff <= heap.worst()) knn_rec(t, far, q, heap);
}
}  // namespace

void KdTree::knn(const float* q, int k, int* out_idx, float* out_sqdist) const {
  std::vector<Cand> buf(k);
  KnnHeap heap{buf.data(), k};
  if (!nodes.empty()) knn_rec(*this, 0, q, heap);
  std::sort(buf.begin(), buf.begin() + heap.n, cand_less);
  for (int i = 0; i < k; i++) {
    out_idx[i] = i < heap.n ? buf[i].i : -1;
    if (out_sqdist) out_sqdist[i] = i < heap.n ? buf[i].d : FLT_MAX;
  }
}

// FastVGICPCuda::find_neighbors_parallel_kdtree (fast_vgicp_cuda_impl.hpp:152-167)
void knn_all(const Cloud& c, int k, int num_threads, std::vector<int>& idx) {
  KdTree tree(c);
  idx.resize(c.size() * k);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < (int)c.size(); i++) tree.knn(c.pt(i), k, &idx[(size_t)i * k], nullptr);
}

// =====================================================================================
// covariances
// =====================================================================================

// fast_gicp_impl.hpp:267-297 (CPU: JacobiSVD, singular values descending) and
// covariance_regularization.cu:34-124 (GPU: symmetric eigen, ascending). For a symmetric PSD
// input both are V diag(f(lambda)) V^T; zero/rank-deficient inputs are arbitrary in the reference.
M3 regularize(const M3& cov, RegularizationMethod m) {
  if (m == NONE) return cov;
  if (m == FROBENIUS) {
    const double lambda = 1e-3;
    M3 C = m3_add(cov, m3_scale(m3_identity(), lambda));
    M3 Ci = m3_inverse(C);
    return m3_inverse(m3_scale(Ci, 1.0 / m3_frobenius(Ci)));
  }
  double w[3];
  M3 V;
  sym_eig3(cov, w, V);  // ascending: w[0] is the smallest (the PLANE normal direction)
  double d[3];
  switch (m) {
    case PLANE: d[0] = 1e-3; d[1] = 1.0; d[2] = 1.0; break;
    case MIN_EIG: for (int i = 0; i < 3; i++) d[i] = std::max(w[i], 1e-3); break;
    case NORMALIZED_MIN_EIG: for (int i = 0; i < 3; i++) d[i] = std::max(w[i] / w[2], 1e-3); break;
    default: std::cerr << "here must not be reached" << std::endl; abort();
  }
  return m3_vdvt(V, d);
}

// FastGICP::calculate_covariances body (fast_gicp_impl.hpp:253-297) with the neighbour
// indices given: centred fp64 covariance, divisor k.
void covariances_from_neighbors(const Cloud& c, int k, const std::vector<int>& idx, RegularizationMethod m, int num_threads, std::vector<M3>& covs) {
  covs.resize(c.size());
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < (int)c.size(); i++) {
    const int* nb = &idx[(size_t)i * k];
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < k; j++) for (int a = 0; a < 3; a++) mean[a] += (double)c.pt(nb[j])[a];
    for (int a = 0; a < 3; a++) mean[a] /= k;
    M3 cov = m3_zero();
    for (int j = 0; j < k; j++) {
      double d[3];
      for (int a = 0; a < 3; a++) d[a] = (double)c.pt(nb[j])[a] - mean[a];
      for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) cov(r, s) += d[r] * d[s];
    }
    cov = m3_scale(cov, 1.0 / k);
    covs[i] = regularize(cov, m);
  }
}

// covariance_estimation_rbf.cu:40-52,67-85,98-109 (CUDA only) restated in fp64 WITHOUT the
// reference's zero-padding bug (padded (0,0,0) points are not masked there).
// mean = S_p/S_w ; cov = (S_pp - mean S_p^T)/S_w, evaluated in the numerically equivalent
// centred-on-query form.
void covariances_rbf(const Cloud& c, double kernel_width, double max_dist, RegularizationMethod m, int num_threads, std::vector<M3>& covs) {
  covs.resize(c.size());
  const float max_dist_sq = (float)max_dist * (float)max_dist;  // :71-72 (fp32 constants)
  const int n = (int)c.size();
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const float* x = c.pt(i);
    double sw = 0, sd[3] = {0, 0, 0};
    M3 sdd = m3_zero();
    for (int j = 0; j < n; j++) {
      const float* p = c.pt(j);
      float sq = sqdist_f32(x, p);
      if (sq > max_dist_sq) continue;
      double w = std::exp(-kernel_width * (double)sq);
      double d[3] = {(double)p[0] - x[0], (double)p[1] - x[1], (double)p[2] - x[2]};
      sw += w;
      for (int a = 0; a < 3; a++) sd[a] += w * d[a];
      for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) sdd(r, s) += w * d[r] * d[s];
    }
    M3 cov;
    for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) cov(r, s) = sdd(r, s) / sw - (sd[r] / sw) * (sd[s] / sw);
    covs[i] = regularize(cov, m);
  }
}

// =====================================================================================
// voxel map
// =====================================================================================
size_t VoxelKeyHash::operator()(const VoxelKey& k) const {
  // boost::hash_combine x3 (fast_vgicp_voxel.hpp:46-55); only affects iteration order
  size_t seed = 0;
  auto comb = [&](int v) { seed ^= std::hash<int>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2); };
  comb(k.x); comb(k.y); comb(k.z);
  return seed;
}

VoxelKey VoxelMap::coord(const V3& x) const {
  return VoxelKey{(int)std::floor(x[0] / resolution - 0.5), (int)std::floor(x[1] / resolution - 0.5), (int)std::floor(x[2] / resolution - 0.5)};
}

int VoxelMap::lookup(const VoxelKey& k) const {
  auto it = index.find(k);
  return it == index.end() ? -1 : it->second;
}

void VoxelMap::create_vgicp(const Cloud& c, const std::vector<M3>& covs, int mode) {
  // mode: VoxelAccumulationMode ordinal (gicp_settings.hpp:10). ADDITIVE (0) and ADDITIVE_WEIGHTED (1) both create an
  // AdditiveGaussianVoxel (fast_vgicp_voxel.hpp:137-141); MULTIPLICATIVE (2) a MultiplicativeGaussianVoxel (:142-144).
  const bool multiplicative = (mode == 2);
  index.clear(); coords.clear(); voxels.clear();
  for (size_t i = 0; i < c.size(); i++) {
    V3 p = {{(double)c.pt(i)[0], (double)c.pt(i)[1], (double)c.pt(i)[2]}};
    VoxelKey key = coord(p);
    auto it = index.find(key);
    if (it == index.end()) {
      it = index.emplace(key, (int)voxels.size()).first;
      coords.push_back(key);
      voxels.emplace_back();
    }
    Voxel& v = voxels[it->second];
    v.num_points++;
    if (!multiplicative) {                            // AdditiveGaussianVoxel::append :112-116
      for (int a = 0; a < 3; a++) v.mean[a] += p[a];
      v.cov = m3_add(v.cov, covs[i]);
    } else {                                          // MultiplicativeGaussianVoxel::append :86-94 (the 4x4 with (3,3) = 1 is block diagonal)
      const M3 ci = m3_inverse(covs[i]);
      v.cov = m3_add(v.cov, ci);
      const V3 cp = m3_mulv(ci, p);
      for (int a = 0; a < 3; a++) v.mean[a] += cp[a];
    }
  }
  for (auto& v : voxels) {
    if (!multiplicative) {                            // finalize :118-121
      for (int a = 0; a < 3; a++) v.mean[a] /= v.num_points;
      v.cov = m3_scale(v.cov, 1.0 / v.num_points);
    } else {                                          // finalize :96-102: cov = (sum C^-1)^-1, mean = cov * sum C^-1 p
      v.cov = m3_inverse(v.cov);
      v.mean = m3_mulv(v.cov, v.mean);
    }
  }
}

void VoxelMap::create_ndt(const Cloud& c) {
  index.clear(); coords.clear(); voxels.clear();
  for (size_t i = 0; i < c.size(); i++) {
    V3 p = {{(double)c.pt(i)[0], (double)c.pt(i)[1], (double)c.pt(i)[2]}};
    VoxelKey key = coord(p);
    auto it = index.find(key);
    if (it == index.end()) {
      it = index.emplace(key, (int)voxels.size()).first;
      coords.push_back(key);
      voxels.emplace_back();
    }
    Voxel& v = voxels[it->second];
    v.num_points++;                                   // gaussian_voxelmap.cu:136-145
    for (int a = 0; a < 3; a++) v.mean[a] += p[a];
    for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) v.cov(r, s) += p[r] * p[s];
  }
  for (auto& v : voxels) {                            // ndt_finalize_voxels_kernel :184-193
    double sum[3] = {v.mean[0], v.mean[1], v.mean[2]};
    for (int a = 0; a < 3; a++) v.mean[a] /= v.num_points;
    for (int r = 0; r < 3; r++) for (int s = 0; s < 3; s++) v.cov(r, s) = (v.cov(r, s) - v.mean[r] * sum[s]) / v.num_points;
    // exact symmetrisation of rounding noise, then MIN_EIG (ndt_cuda.cu:128,139)
    for (int r = 0; r < 3; r++) for (int s = r + 1; s < 3; s++) { double a = 0.5 * (v.cov(r, s) + v.cov(s, r)); v.cov(r, s) = v.cov(s, r) = a; }
    v.cov = regularize(v.cov, MIN_EIG);
  }
}

std::vector<VoxelKey> neighbor_offsets(NeighborSearchMethod m, double radius) {
  std::vector<VoxelKey> o;
  switch (m) {
    case DIRECT1: o.push_back({0, 0, 0}); break;
    case DIRECT7:
      o = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
      break;
    case DIRECT27:
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) o.push_back({i - 1, j - 1, k - 1});
      break;
    case DIRECT_RADIUS: {  // fast_vgicp_cuda.cu:77-91
      int range = (int)std::ceil(radius);
      for (int i = -range; i <= range; i++) for (int j = -range; j <= range; j++) for (int k = -range; k <= range; k++)
        if (std::sqrt((double)(i * i + j * j + k * k)) <= radius + 1e-3) o.push_back({i, j, k});
    } break;
    default: std::cerr << "unsupported neighbor search method" << std::endl; abort();
  }
  return o;
}

// =====================================================================================
// LM optimiser (lsq_registration_impl.hpp)
// =====================================================================================
bool LsqBase::is_converged(const Iso3& delta) const {
  double rmax = 0, tmax = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rmax = std::max(rmax, std::fabs(delta.R[i * 3 + j] - (i == j ? 1.0 : 0.0)) / rotation_epsilon);
  for (int i = 0; i < 3; i++) tmax = std::max(tmax, std::fabs(delta.t[i]) / transformation_epsilon);
  return std::max(rmax, tmax) < 1;
}

// lsq_registration_impl.hpp:108-121
bool LsqBase::step_gn(Iso3& x0, Iso3& delta) {
  double H[36], b[6], nb[6], d[6];
  linearize(x0, H, b);
  num_linearize++;
  for (int j = 0; j < 6; j++) nb[j] = -b[j];
  ldlt6_solve(H, nb, d);
  delta = se3_exp(d);
  x0 = iso_mul(delta, x0);
  std::memcpy(final_hessian, H, sizeof(H));
  return true;
}

bool LsqBase::step_lm(Iso3& x0, Iso3& delta) {
  double H[36], b[6];
  double y0 = linearize(x0, H, b);
  num_linearize++;
  if (lm_lambda < 0.0) {
    double mx = 0;
    for (int i = 0; i < 6; i++) mx = std::max(mx, std::fabs(H[i * 6 + i]));
    lm_lambda = lm_init_lambda_factor * mx;
  }
  double nu = 2.0;
  for (int i = 0; i < lm_max_iterations; i++) {
    double A[36], nb[6], d[6];
    std::memcpy(A, H, sizeof(A));
    for (int j = 0; j < 6; j++) { A[j * 6 + j] += lm_lambda; nb[j] = -b[j]; }
    ldlt6_solve(A, nb, d);
    delta = se3_exp(d);
    Iso3 xi = iso_mul(delta, x0);
    double yi = compute_error(xi);
    num_error_evals++;
    double denom = 0;
    for (int j = 0; j < 6; j++) denom += d[j] * (lm_lambda * d[j] - b[j]);
    double rho = (y0 - yi) / denom;
    if (lm_debug_print) {
      double dn = 0; for (int j = 0; j < 6; j++) dn += d[j] * d[j];
      std::printf("%5d %15g %15g %15g %15g %15g %5c\n", i, y0, yi, rho, lm_lambda, std::sqrt(dn), rho > 0.0 ? 'x' : ' ');
    }
    if (rho < 0) {
      if (is_converged(delta)) return true;
      lm_lambda = nu * lm_lambda;
      nu = 2 * nu;
      continue;
    }
    x0 = xi;
    lm_lambda = lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    std::memcpy(final_hessian, H, sizeof(H));
    return true;
  }
  return false;
}

void LsqBase::optimize(const Iso3& guess) {
  Iso3 x0 = guess;
  lm_lambda = -1.0;
  converged = false;
  num_linearize = num_error_evals = 0;
  for (int i = 0; i < max_iterations && !converged; i++) {
    nr_iterations = i;
    Iso3 delta;
    if (!(gauss_newton ? step_gn(x0, delta) : step_lm(x0, delta))) {  // step_optimize, :94-104
      std::cerr << "lm not converged!!" << std::endl;
      break;
    }
    converged = is_converged(delta);
  }
  final_transformation = x0;
}

// =====================================================================================
// fitness (pcl::Registration::getFitnessScore, max_range = inf)
// =====================================================================================
double fitness_score(const Cloud& source, const KdTree& target_tree, const Iso3& T) {
  float m[12];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m[i * 4 + j] = (float)T.R[i * 3 + j]; m[i * 4 + 3] = (float)T.t[i]; }
  double sum = 0;
  const int n = (int)source.size();
#pragma omp parallel for reduction(+ : sum) schedule(static)
  for (int i = 0; i < n; i++) {
    const float* p = source.pt(i);
    float q[3];
    for (int r = 0; r < 3; r++) q[r] = (p[0] * m[r * 4 + 0] + p[1] * m[r * 4 + 1]) + (p[2] * m[r * 4 + 2] + m[r * 4 + 3]);
    int idx; float d;
    target_tree.knn(q, 1, &idx, &d);
    sum += (double)d;
  }
  return n > 0 ? sum / n : DBL_MAX;
}

// =====================================================================================
// FastVGICP
// =====================================================================================
FastVGICP::FastVGICP() { num_threads = omp_get_max_threads(); for (int i = 0; i < 36; i++) final_hessian[i] = (i % 7 == 0) ? 1.0 : 0.0; }

void FastVGICP::setInputTarget(const CloudPtr& c) {
  if (target == c) return;
  target = c;
  target_cloud_updated = true;                 // pcl::Registration::setInputTarget
  search_target.reset(new KdTree(*c));         // search_target_->setInputCloud
  target_covs.clear();
  voxelmap.reset();
}
void FastVGICP::setInputSource(const CloudPtr& c) {
  if (input == c) return;
  input = c;
  search_source.reset(new KdTree(*c));
  source_covs.clear();
}
void FastVGICP::clearSource() { input.reset(); source_covs.clear(); }
void FastVGICP::clearTarget() { target.reset(); target_covs.clear(); }
void FastVGICP::swapSourceAndTarget() {
  input.swap(target);
  search_source.swap(search_target);
  source_covs.swap(target_covs);
  voxelmap.reset();
  voxel_correspondences.clear();
  voxel_mahalanobis.clear();
}

void FastVGICP::calculate_covariances(const CloudPtr& c, const KdTree& tree, std::vector<M3>& covs) {
  if (cov_mode == 1) {
    covariances_rbf(*c, kernel_width, kernel_max_dist, regularization, num_threads, covs);
  } else {
    const int k = k_correspondences;
    std::vector<int> idx(c->size() * k);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < (int)c->size(); i++) tree.knn(c->pt(i), k, &idx[(size_t)i * k], nullptr);
    covariances_from_neighbors(*c, k, idx, regularization, num_threads, covs);
  }
  if (round_storage_fp32)
    for (auto& m : covs) for (int i = 0; i < 9; i++) m.m[i] = (double)(float)m.m[i];
}

void FastVGICP::align(const Iso3& guess) {
  // pcl::Registration::align -> initCompute(): (re)build tree_ on a new target
  if (target_cloud_updated) { pcl_tree.reset(new KdTree(*target)); target_cloud_updated = false; }
  voxelmap.reset();                                                          // fast_vgicp_impl.hpp:67
  if (source_covs.size() != input->size()) calculate_covariances(input, *search_source, source_covs);   // fast_gicp_impl.hpp:107-112
  if (target_covs.size() != target->size()) calculate_covariances(target, *search_target, target_covs);
  optimize(guess);
}

double FastVGICP::getFitnessScore() const {
  // final_transformation_ = x0.cast<float>() (lsq_registration_impl.hpp:77)
  return fitness_score(*input, *pcl_tree, final_transformation);
}

void FastVGICP::update_correspondences(const Iso3& T) {
  lin_pose = T;  // (kept for the cuda-compat leg, which re-forms R_eval C_A R_eval^T in float)
  voxel_correspondences.clear();
  if (gicp_mode) {
    // FastGICP::update_correspondences (fast_gicp_impl.hpp:118-156): query = trans.cast<float>() * point (fp32), exact 1-NN
    // in the target, kept when the fp32 squared distance < threshold^2; mahalanobis = (C_B + T C_A T^T)^-1 in fp64.
    float m[12];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m[i * 4 + j] = (float)T.R[i * 3 + j]; m[i * 4 + 3] = (float)T.t[i]; }
    const int n = (int)input->size();
    std::vector<int> nn(n);
    const double thr_sq = max_correspondence_distance * max_correspondence_distance;
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < n; i++) {
      const float* p = input->pt(i);
      float q[3];
      for (int r = 0; r < 3; r++) q[r] = (p[0] * m[r * 4 + 0] + p[1] * m[r * 4 + 1]) + (p[2] * m[r * 4 + 2] + m[r * 4 + 3]);
      int idx; float d;
      search_target->knn(q, 1, &idx, &d);
      nn[i] = ((double)d < thr_sq) ? idx : -1;
    }
    for (int i = 0; i < n; i++) if (nn[i] >= 0) voxel_correspondences.push_back({i, nn[i]});
    voxel_mahalanobis.resize(voxel_correspondences.size());
    const M3 R = iso_rot(T), Rt = m3_transpose(R);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
    for (int i = 0; i < (int)voxel_correspondences.size(); i++) {
      const auto& corr = voxel_correspondences[i];
      voxel_mahalanobis[i] = m3_inverse(m3_add(target_covs[corr.second], m3_mul(m3_mul(R, source_covs[corr.first]), Rt)));
    }
    return;
  }
  auto offsets = neighbor_offsets(search_method);
  std::vector<std::vector<std::pair<int, int>>> corrs(num_threads);
  for (auto& c : corrs) c.reserve((input->size() * offsets.size()) / num_threads);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < (int)input->size(); i++) {
    V3 a = {{(double)input->pt(i)[0], (double)input->pt(i)[1], (double)input->pt(i)[2]}};
    V3 q = iso_apply(T, a);
    VoxelKey coord = voxelmap->coord(q);
    for (const auto& o : offsets) {
      int v = voxelmap->lookup(VoxelKey{coord.x + o.x, coord.y + o.y, coord.z + o.z});
      if (v >= 0) corrs[omp_get_thread_num()].push_back({i, v});
    }
  }
  voxel_correspondences.reserve(input->size() * offsets.size());
  for (const auto& c : corrs) voxel_correspondences.insert(voxel_correspondences.end(), c.begin(), c.end());

  voxel_mahalanobis.resize(voxel_correspondences.size());
  const M3 R = iso_rot(T), Rt = m3_transpose(R);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 8)
  for (int i = 0; i < (int)voxel_correspondences.size(); i++) {
    const auto& corr = voxel_correspondences[i];
    M3 RCR = m3_add(voxelmap->voxels[corr.second].cov, m3_mul(m3_mul(R, source_covs[corr.first]), Rt));
    voxel_mahalanobis[i] = m3_inverse(RCR);   // 4x4 with (3,3)=1 then zeroed == 3x3 inverse (:110-114)
  }
}

double FastVGICP::linearize(const Iso3& T, double* H, double* b) {
  if (!gicp_mode && !voxelmap) {
    voxelmap.reset(new VoxelMap(voxel_resolution));
    voxelmap->create_vgicp(*target, target_covs, voxel_mode);
    if (round_storage_fp32)
      for (auto& v : voxelmap->voxels) { for (int a = 0; a < 3; a++) v.mean[a] = (double)(float)v.mean[a]; for (int i = 0; i < 9; i++) v.cov.m[i] = (double)(float)v.cov.m[i]; }
  }
  update_correspondences(T);

  double sum_errors = 0.0;
  std::vector<std::array<double, 42>> part(num_threads);
  for (auto& p : part) p.fill(0.0);
#pragma omp parallel for num_threads(num_threads) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < (int)voxel_correspondences.size(); i++) {
    const auto& corr = voxel_correspondences[i];
    double mu[3], w;
    if (gicp_mode) { for (int a = 0; a < 3; a++) mu[a] = (double)target->pt(corr.second)[a]; w = 1.0; }   // fast_gicp_impl.hpp:176-183
    else { const Voxel& vox = voxelmap->voxels[corr.second]; for (int a = 0; a < 3; a++) mu[a] = vox.mean[a]; w = std::sqrt((double)vox.num_points); }
    V3 a = {{(double)input->pt(corr.first)[0], (double)input->pt(corr.first)[1], (double)input->pt(corr.first)[2]}};
    V3 q = iso_apply(T, a);
    V3 e = {{mu[0] - q[0], mu[1] - q[1], mu[2] - q[2]}};
    const M3& M = voxel_mahalanobis[i];
    V3 Me = m3_mulv(M, e);
    sum_errors += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
    if (H == nullptr || b == nullptr) continue;
    // J = [skew(q), -I] (3x6); Hi = w J^T M J ; bi = w J^T M e
    double J[3][6];
    M3 S = skew(q);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J[r][c] = S(r, c); J[r][3 + c] = (r == c) ? -1.0 : 0.0; }
    double MJ[3][6];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 6; c++) MJ[r][c] = M(r, 0) * J[0][c] + M(r, 1) * J[1][c] + M(r, 2) * J[2][c];
    auto& P = part[omp_get_thread_num()];
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) P[r * 6 + c] += w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
      P[36 + r] += w * (J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2]);
    }
  }
  if (H && b) {
    std::memset(H, 0, 36 * sizeof(double));
    std::memset(b, 0, 6 * sizeof(double));
    for (int t = 0; t < num_threads; t++) { for (int i = 0; i < 36; i++) H[i] += part[t][i]; for (int i = 0; i < 6; i++) b[i] += part[t][36 + i]; }
  }
  return sum_errors;
}

double FastVGICP::compute_error(const Iso3& T) {
  double sum_errors = 0.0;
#pragma omp parallel for num_threads(num_threads) reduction(+ : sum_errors)
  for (int i = 0; i < (int)voxel_correspondences.size(); i++) {
    const auto& corr = voxel_correspondences[i];
    double mu[3], w;
    if (gicp_mode) { for (int a = 0; a < 3; a++) mu[a] = (double)target->pt(corr.second)[a]; w = 1.0; }   // fast_gicp_impl.hpp:216-238
    else { const Voxel& vox = voxelmap->voxels[corr.second]; for (int a = 0; a < 3; a++) mu[a] = vox.mean[a]; w = std::sqrt((double)vox.num_points); }
    V3 a = {{(double)input->pt(corr.first)[0], (double)input->pt(corr.first)[1], (double)input->pt(corr.first)[2]}};
    V3 q = iso_apply(T, a);
    V3 e = {{mu[0] - q[0], mu[1] - q[1], mu[2] - q[2]}};
    V3 Me = m3_mulv(voxel_mahalanobis[i], e);
    sum_errors += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
  }
  return sum_errors;
}

// "cuda-compat" leg: the per-correspondence arithmetic of the reference's DEVICE path in float, term by term as
// compute_derivatives.cu:50-103 writes it (pose cast to Isometry3f, float covariances and voxel records,
// RCR = (R_eval C_A) R_eval^T, cofactor inverse, w = sqrtf(n), H = ((w J^T) M) J, b = ((w J^T) M) e, err = ((w e^T) M) e),
// over the correspondences and the linearisation pose of the last linearize(). The reference reduces the float terms with a
// Thrust tree of unspecified order; here (and in the engine's FVH_COMPUTE_FP32 mode) the terms are summed in double.
double FastVGICP::cuda_compat_sums(const Iso3& T, double* H, double* b) const {
  float Re[9], R[9], t[3];
  for (int i = 0; i < 9; i++) { Re[i] = (float)lin_pose.R[i]; R[i] = (float)T.R[i]; }
  for (int i = 0; i < 3; i++) t[i] = (float)T.t[i];
  double sum_e = 0.0, Hs[36] = {0}, bs[6] = {0};
  for (size_t n = 0; n < voxel_correspondences.size(); n++) {
    const auto& corr = voxel_correspondences[n];
    const Voxel& vox = voxelmap->voxels[corr.second];
    if (vox.num_points <= 0) continue;
    float a[3], mu[3], CA[9], CB[9];
    for (int i = 0; i < 3; i++) { a[i] = input->pt(corr.first)[i]; mu[i] = (float)vox.mean[i]; }
    for (int i = 0; i < 9; i++) { CA[i] = (float)source_covs[corr.first].m[i]; CB[i] = (float)vox.cov.m[i]; }
    float q[3];
    for (int i = 0; i < 3; i++) q[i] = (R[i * 3] * a[0] + R[i * 3 + 1] * a[1] + R[i * 3 + 2] * a[2]) + t[i];
    float RC[9], A[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RC[i * 3 + j] = Re[i * 3] * CA[j] + Re[i * 3 + 1] * CA[3 + j] + Re[i * 3 + 2] * CA[6 + j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i * 3 + j] = CB[i * 3 + j] + (RC[i * 3] * Re[j * 3] + RC[i * 3 + 1] * Re[j * 3 + 1] + RC[i * 3 + 2] * Re[j * 3 + 2]);
    float c[9];
    c[0] = A[4] * A[8] - A[5] * A[7]; c[1] = A[2] * A[7] - A[1] * A[8]; c[2] = A[1] * A[5] - A[2] * A[4];
    c[3] = A[5] * A[6] - A[3] * A[8]; c[4] = A[0] * A[8] - A[2] * A[6]; c[5] = A[2] * A[3] - A[0] * A[5];
    c[6] = A[3] * A[7] - A[4] * A[6]; c[7] = A[1] * A[6] - A[0] * A[7]; c[8] = A[0] * A[4] - A[1] * A[3];
    const float det = A[0] * c[0] + A[1] * c[3] + A[2] * c[6], inv = 1.0f / det;
    float M[9];
    for (int i = 0; i < 9; i++) M[i] = c[i] * inv;
    const float w = std::sqrt((float)vox.num_points);
    const float e[3] = {mu[0] - q[0], mu[1] - q[1], mu[2] - q[2]};
    float J[3][6] = {{0, -q[2], q[1], -1, 0, 0}, {q[2], 0, -q[0], 0, -1, 0}, {-q[1], q[0], 0, 0, 0, -1}};
    float wJtM[6][3];  // (w J^T) M
    for (int r = 0; r < 6; r++) for (int k = 0; k < 3; k++) wJtM[r][k] = (w * J[0][r]) * M[k] + (w * J[1][r]) * M[3 + k] + (w * J[2][r]) * M[6 + k];
    float weM[3];
    for (int k = 0; k < 3; k++) weM[k] = (w * e[0]) * M[k] + (w * e[1]) * M[3 + k] + (w * e[2]) * M[6 + k];
    sum_e += (double)(weM[0] * e[0] + weM[1] * e[1] + weM[2] * e[2]);
    if (!H || !b) continue;
    for (int r = 0; r < 6; r++) {
      for (int cc = 0; cc < 6; cc++) Hs[r * 6 + cc] += (double)(wJtM[r][0] * J[0][cc] + wJtM[r][1] * J[1][cc] + wJtM[r][2] * J[2][cc]);
      bs[r] += (double)(wJtM[r][0] * e[0] + wJtM[r][1] * e[1] + wJtM[r][2] * e[2]);
    }
  }
  if (H && b) { std::memcpy(H, Hs, sizeof(Hs)); std::memcpy(b, bs, sizeof(bs)); }
  return sum_e;
}

// =====================================================================================
// NDT (CUDA-only in the reference; fp64 restatement of the same formulas)
// =====================================================================================
NDT::NDT() { num_threads = omp_get_max_threads(); for (int i = 0; i < 36; i++) final_hessian[i] = (i % 7 == 0) ? 1.0 : 0.0; }

void NDT::setInputTarget(const CloudPtr& c) {
  if (c == target) return;
  target = c;
  target_cloud_updated = true;
  target_voxelmap.reset();
}
void NDT::setInputSource(const CloudPtr& c) {
  if (c == input) return;
  input = c;
  source_voxelmap.reset();
}
void NDT::swapSourceAndTarget() {
  source_voxelmap.swap(target_voxelmap);
  input.swap(target);
}
void NDT::create_voxelmaps() {
  // round_storage_fp32: the voxel means / covariances go through float like the device tables (the reference's are float
  // too: gaussian_voxelmap.cuh:30-32) -- the parity leg that isolates the cost arithmetic from the storage rounding
  auto round_map = [&](VoxelMap& m) {
    if (!round_storage_fp32) return;
    for (auto& v : m.voxels) { for (int a = 0; a < 3; a++) v.mean[a] = (double)(float)v.mean[a]; for (int i = 0; i < 9; i++) v.cov.m[i] = (double)(float)v.cov.m[i]; }
  };
  if (!source_voxelmap && distance_mode != P2D) { source_voxelmap.reset(new VoxelMap(resolution)); source_voxelmap->create_ndt(*input); round_map(*source_voxelmap); }
  if (!target_voxelmap) { target_voxelmap.reset(new VoxelMap(resolution)); target_voxelmap->create_ndt(*target); round_map(*target_voxelmap); }
}
void NDT::align(const Iso3& guess) {
  if (target_cloud_updated) { pcl_tree.reset(new KdTree(*target)); target_cloud_updated = false; }
  create_voxelmaps();
  optimize(guess);
}
double NDT::getFitnessScore() const { return fitness_score(*input, *pcl_tree, final_transformation); }

double NDT::linearize(const Iso3& T, double* H, double* b) {
  // NDTCudaCore::update_correspondences (ndt_cuda.cu:142-161): P2D over source points,
  // D2D over source-voxel means; offset-major order, invalid pairs dropped.
  linearized_x = T;
  correspondences.clear();
  auto offsets = neighbor_offsets(search_method, search_radius);
  const int ns = distance_mode == P2D ? (int)input->size() : (int)source_voxelmap->voxels.size();
  for (const auto& o : offsets)
    for (int i = 0; i < ns; i++) {
      V3 a;
      if (distance_mode == P2D) a = V3{{(double)input->pt(i)[0], (double)input->pt(i)[1], (double)input->pt(i)[2]}};
      else a = source_voxelmap->voxels[i].mean;
      VoxelKey c = target_voxelmap->coord(iso_apply(T, a));
      int v = target_voxelmap->lookup(VoxelKey{c.x + o.x, c.y + o.y, c.z + o.z});
      if (v >= 0) correspondences.push_back({i, v});
    }
  return cost(T, H, b);
}
double NDT::compute_error(const Iso3& T) { return cost(T, nullptr, nullptr); }

double NDT::cost(const Iso3& T, double* H, double* b) {
  const M3 Re = iso_rot(linearized_x), Ret = m3_transpose(Re);
  double sum_errors = 0.0;
  double acc[42];
  std::memset(acc, 0, sizeof(acc));
  for (size_t i = 0; i < correspondences.size(); i++) {
    const auto& corr = correspondences[i];
    const Voxel& vox = target_voxelmap->voxels[corr.second];
    if (vox.num_points <= 6) continue;   // ndt_compute_derivatives.cu:61,133
    V3 a; M3 M;
    if (distance_mode == P2D) {
      a = V3{{(double)input->pt(corr.first)[0], (double)input->pt(corr.first)[1], (double)input->pt(corr.first)[2]}};
      M = m3_inverse(vox.cov);
    } else {
      const Voxel& sv = source_voxelmap->voxels[corr.first];
      a = sv.mean;
      M = m3_inverse(m3_add(vox.cov, m3_mul(m3_mul(Re, sv.cov), Ret)));
    }
    V3 q = iso_apply(T, a);
    V3 e = {{vox.mean[0] - q[0], vox.mean[1] - q[1], vox.mean[2] - q[2]}};
    double en = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    double ksq = resolution * resolution;
    double w = ksq / (ksq + en * en);   // cauchy(resolution, |e|) :15-18
    V3 Me = m3_mulv(M, e);
    sum_errors += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
    if (!H || !b) continue;
    double J[3][6];
    M3 S = skew(q);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J[r][c] = S(r, c); J[r][3 + c] = (r == c) ? -1.0 : 0.0; }
    double MJ[3][6];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 6; c++) MJ[r][c] = M(r, 0) * J[0][c] + M(r, 1) * J[1][c] + M(r, 2) * J[2][c];
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) acc[r * 6 + c] += w * (J[0][r] * MJ[0][c] + J[1][r] * MJ[1][c] + J[2][r] * MJ[2][c]);
      acc[36 + r] += w * (J[0][r] * Me[0] + J[1][r] * Me[1] + J[2][r] * Me[2]);
    }
  }
  if (H && b) { std::memcpy(H, acc, 36 * sizeof(double)); std::memcpy(b, acc + 36, 6 * sizeof(double)); }
  return sum_errors;
}

}  // namespace orc
