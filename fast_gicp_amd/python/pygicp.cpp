// pygicp -- Python bindings with the surface of the reference's src/python/main.cpp:152-223, on the
// host-side C++ mirror (include/fast_gicp_amd/registration.hpp) -> C ABI -> HIP engine.
// Eigen is not available here, so Nx3 / 4x4 arguments are numpy arrays (same shapes and dtypes the
// reference's pybind11/eigen.h conversions accept and return).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <iostream>

#include "fast_gicp_amd/registration.hpp"
#include "fast_gicp_amd/voxelgrid.hpp"

namespace py = pybind11;
using namespace fast_gicp;

using Cloud = PointCloud<PointXYZ>;
using Lsq = LsqRegistration<PointXYZ, PointXYZ>;
using VGICPCuda = FastVGICPCuda<PointXYZ, PointXYZ>;
using VGICP = FastVGICP<PointXYZ, PointXYZ>;
using GICP = FastGICP<PointXYZ, PointXYZ>;
using NDT = NDTCuda<PointXYZ, PointXYZ>;
using Points = py::array_t<double, py::array::c_style | py::array::forcecast>;
using Mat4 = py::array_t<double, py::array::c_style | py::array::forcecast>;

static NeighborSearchMethod search_method(const std::string& s) {  // main.cpp:21-34
  if (s == "DIRECT1") return NeighborSearchMethod::DIRECT1;
  if (s == "DIRECT7") return NeighborSearchMethod::DIRECT7;
  if (s == "DIRECT27") return NeighborSearchMethod::DIRECT27;
  if (s == "DIRECT_RADIUS") return NeighborSearchMethod::DIRECT_RADIUS;
  std::cerr << "error: unknown neighbor search method " << s << std::endl;
  return NeighborSearchMethod::DIRECT1;
}
static RegularizationMethod regularization_method(const std::string& s) {
  if (s == "NONE") return RegularizationMethod::NONE;
  if (s == "MIN_EIG") return RegularizationMethod::MIN_EIG;
  if (s == "NORMALIZED_MIN_EIG") return RegularizationMethod::NORMALIZED_MIN_EIG;
  if (s == "PLANE") return RegularizationMethod::PLANE;
  if (s == "FROBENIUS") return RegularizationMethod::FROBENIUS;
  throw std::invalid_argument("unknown regularization method " + s);
}
static NearestNeighborMethod nn_method(const std::string& s) {
  if (s == "CPU_PARALLEL_KDTREE") return NearestNeighborMethod::CPU_PARALLEL_KDTREE;
  if (s == "GPU_BRUTEFORCE") return NearestNeighborMethod::GPU_BRUTEFORCE;
  if (s == "GPU_RBF_KERNEL") return NearestNeighborMethod::GPU_RBF_KERNEL;
  throw std::invalid_argument("unknown nearest neighbor method " + s);
}

static Cloud::Ptr numpy2cloud(const Points& pts) {  // eigen2pcl, main.cpp:36-44
  if (pts.ndim() != 2 || pts.shape(1) != 3) throw std::invalid_argument("points must be an (N, 3) array");
  auto cloud = std::make_shared<Cloud>();
  cloud->resize(pts.shape(0));
  auto r = pts.unchecked<2>();
  for (py::ssize_t i = 0; i < pts.shape(0); i++) {
    cloud->points[i].x = (float)r(i, 0); cloud->points[i].y = (float)r(i, 1); cloud->points[i].z = (float)r(i, 2);
  }
  return cloud;
}
static py::array_t<double> cloud2numpy(const Cloud& c) {
  py::array_t<double> out({(py::ssize_t)c.size(), (py::ssize_t)3});
  auto w = out.mutable_unchecked<2>();
  for (size_t i = 0; i < c.size(); i++) { w(i, 0) = c.points[i].x; w(i, 1) = c.points[i].y; w(i, 2) = c.points[i].z; }
  return out;
}
static Matrix4f numpy2mat4(const Mat4& m) {
  if (m.ndim() != 2 || m.shape(0) != 4 || m.shape(1) != 4) throw std::invalid_argument("pose must be a (4, 4) array");
  Matrix4f M;
  auto r = m.unchecked<2>();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) M(i, j) = (float)r(i, j);
  return M;
}
static py::array_t<float> mat4_to_numpy(const Matrix4f& M) {
  py::array_t<float> out({4, 4});
  auto w = out.mutable_unchecked<2>();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) w(i, j) = M(i, j);
  return out;
}
static py::array_t<double> identity4() {
  py::array_t<double> I({4, 4});
  auto w = I.mutable_unchecked<2>();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) w(i, j) = i == j ? 1.0 : 0.0;
  return I;
}

static py::array_t<double> downsample(const Points& points, double resolution) {  // main.cpp:46-62
  Cloud filtered;
  approximate_voxel_grid(*numpy2cloud(points), (float)resolution, filtered);
  return cloud2numpy(filtered);
}

// downsample on the device: the same filter (pcl::ApproximateVoxelGrid semantics, or pcl::VoxelGrid with exact=True), bit-identical
// output in the same order, computed by fvh_voxelgrid_* (kernels_downsample.hpp)
static py::array_t<double> downsample_device(const Points& points, double resolution, bool exact, int device) {
  auto cloud = numpy2cloud(points);
  const std::vector<float> xyz = detail::pack_xyz(*cloud);
  fvh_voxelgrid* vg = nullptr;
  detail::check(fvh_voxelgrid_create(device, &vg), "fvh_voxelgrid_create", "cannot create the HIP engine (no GPU? there is no CPU fallback)");
  int n = 0;
  int rc = fvh_voxelgrid_filter(vg, exact ? FVH_VOXELGRID_EXACT : FVH_VOXELGRID_APPROXIMATE, xyz.data(), (int)cloud->size(), (float)resolution, &n);
  std::vector<float> out((size_t)3 * n);
  if (!rc) rc = fvh_voxelgrid_get_points(vg, out.data());
  const std::string err = rc ? fvh_voxelgrid_last_error(vg) : "";
  fvh_voxelgrid_destroy(vg);
  detail::check(rc, "fvh_voxelgrid_filter", err.c_str());
  py::array_t<double> res({(py::ssize_t)n, (py::ssize_t)3});
  auto w = res.mutable_unchecked<2>();
  for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) w(i, a) = out[3 * (size_t)i + a];
  return res;
}

// align_points, main.cpp:64-150. "VGICP" (the CPU FastVGICP in the reference, main.cpp:102-108) runs on the same GPU engine in
// its fp64 arithmetic WITH k_correspondences honoured (class FastVGICP of registration.hpp; "VGICP_CUDA" ignores it like the
// reference's FastVGICPCuda: k = 20); "GICP" (nearest-point correspondences, no voxels) runs on the device too.
static py::array_t<double> align_points(const Points& target, const Points& source, const std::string& method, double downsample_resolution, int k_correspondences,
                                        double max_correspondence_distance, double voxel_resolution, int /*num_threads*/, const std::string& neighbor_search_method,
                                        double neighbor_search_radius, const Mat4& initial_guess) {
  Cloud::Ptr target_cloud = numpy2cloud(target), source_cloud = numpy2cloud(source);
  if (downsample_resolution > 0.0) {
    auto ft = std::make_shared<Cloud>(), fs = std::make_shared<Cloud>();
    approximate_voxel_grid(*target_cloud, (float)downsample_resolution, *ft);
    approximate_voxel_grid(*source_cloud, (float)downsample_resolution, *fs);
    target_cloud = ft; source_cloud = fs;
  }
  std::shared_ptr<Lsq> reg;
  if (method == "VGICP") {  // main.cpp:102-108
    auto vgicp = std::make_shared<VGICP>();
    vgicp->setCorrespondenceRandomness(k_correspondences);
    vgicp->setResolution(voxel_resolution);
    if (search_method(neighbor_search_method) == fast_gicp::NeighborSearchMethod::DIRECT_RADIUS) {
      std::cerr << "error: FastVGICP has no DIRECT_RADIUS neighbor search (use VGICP_CUDA)" << std::endl;
      return identity4();
    }
    vgicp->setNeighborSearchMethod(search_method(neighbor_search_method));
    vgicp->setNumThreads(0);
    reg = vgicp;
  } else if (method == "VGICP_CUDA") {
    auto vgicp = std::make_shared<VGICPCuda>();
    vgicp->setCorrespondenceRandomness(k_correspondences);  // a no-op, as in the reference (k = 20)
    vgicp->setNeighborSearchMethod(search_method(neighbor_search_method), neighbor_search_radius);
    vgicp->setResolution(voxel_resolution);
    reg = vgicp;
  } else if (method == "NDT_CUDA") {
    auto ndt = std::make_shared<NDT>();
    ndt->setResolution(voxel_resolution);
    ndt->setNeighborSearchMethod(search_method(neighbor_search_method), neighbor_search_radius);
    reg = ndt;
  } else if (method == "GICP") {  // main.cpp:96-101
    auto gicp = std::make_shared<GICP>();
    gicp->setMaxCorrespondenceDistance(max_correspondence_distance);
    gicp->setCorrespondenceRandomness(k_correspondences);
    reg = gicp;
  } else {
    std::cerr << "error: registration method " << method << " is not provided by the MI355X engine (GICP, VGICP, VGICP_CUDA, NDT_CUDA)" << std::endl;
    return identity4();
  }
  reg->setInputTarget(target_cloud);
  reg->setInputSource(source_cloud);
  Cloud aligned;
  reg->align(aligned, numpy2mat4(initial_guess));
  py::array_t<double> out({4, 4});
  auto w = out.mutable_unchecked<2>();
  const Matrix4f& M = reg->getFinalTransformation();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) w(i, j) = M(i, j);
  return out;
}

static py::array_t<double> covs_to_numpy(const GICP::Covariances& covs) {
  py::array_t<double> out({(py::ssize_t)covs.size(), (py::ssize_t)3, (py::ssize_t)3});
  double* o = out.mutable_data();
  for (size_t i = 0; i < covs.size(); i++) std::memcpy(o + 9 * i, covs[i].data(), 9 * sizeof(double));
  return out;
}
static GICP::Covariances numpy_to_covs(const py::array_t<double, py::array::c_style | py::array::forcecast>& c) {
  if (c.ndim() != 3 || c.shape(1) != 3 || c.shape(2) != 3) throw std::invalid_argument("covariances must be an (N, 3, 3) array");
  GICP::Covariances covs((size_t)c.shape(0));
  for (size_t i = 0; i < covs.size(); i++) std::memcpy(covs[i].data(), c.data() + 9 * i, 9 * sizeof(double));
  return covs;
}

PYBIND11_MODULE(pygicp, m) {
  m.doc() = "pygicp on the MI355X HIP engine (surface of koide3/fast_gicp src/python/main.cpp)";
  m.def("downsample", &downsample, "downsample points", py::arg("points"), py::arg("downsample_resolution"));
  m.def("downsample_device", &downsample_device, "downsample points on the GPU (bit-identical to downsample; exact=True: pcl::VoxelGrid)", py::arg("points"),
        py::arg("downsample_resolution"), py::arg("exact") = false, py::arg("device") = 0);
  m.def("align_points", &align_points, "align two point sets", py::arg("target"), py::arg("source"), py::arg("method") = "VGICP_CUDA", py::arg("downsample_resolution") = -1.0,
        py::arg("k_correspondences") = 15, py::arg("max_correspondence_distance") = std::numeric_limits<double>::max(), py::arg("voxel_resolution") = 1.0,
        py::arg("num_threads") = 0, py::arg("neighbor_search_method") = "DIRECT1", py::arg("neighbor_search_radius") = 1.5, py::arg("initial_guess") = identity4());

  py::class_<Lsq, std::shared_ptr<Lsq>>(m, "LsqRegistration")
      .def("set_input_target", [](Lsq& reg, const Points& p) { reg.setInputTarget(numpy2cloud(p)); })
      .def("set_input_source", [](Lsq& reg, const Points& p) { reg.setInputSource(numpy2cloud(p)); })
      .def("swap_source_and_target", &Lsq::swapSourceAndTarget)
      .def("clear_source", &Lsq::clearSource)
      .def("clear_target", &Lsq::clearTarget)
      .def("get_final_hessian", [](Lsq& reg) {
        py::array_t<double> H({6, 6});
        std::memcpy(H.mutable_data(), reg.getFinalHessian().data(), 36 * sizeof(double));
        return H;
      })
      .def("get_final_transformation", [](Lsq& reg) { return mat4_to_numpy(reg.getFinalTransformation()); })
      .def("get_fitness_score", [](Lsq& reg, double max_range) { return reg.getFitnessScore(max_range); }, py::arg("max_range") = std::numeric_limits<double>::max())
      .def("has_converged", &Lsq::hasConverged)
      .def("set_maximum_iterations", &Lsq::setMaximumIterations)
      .def("set_transformation_epsilon", &Lsq::setTransformationEpsilon)
      .def("set_rotation_epsilon", &Lsq::setRotationEpsilon)
      .def("set_initial_lambda_factor", &Lsq::setInitialLambdaFactor)
      .def("set_debug_print", &Lsq::setDebugPrint)
      .def("set_use_device_lm", &Lsq::setUseDeviceLM)
      .def("set_lsq_type", [](Lsq& reg, const std::string& t) {  // lsq_optimizer_type_ (lsq_registration.hpp:13,78): "LM" (default) or "GN"
        if (t == "LM" || t == "LevenbergMarquardt") reg.setLSQType(fast_gicp::LSQ_OPTIMIZER_TYPE::LevenbergMarquardt);
        else if (t == "GN" || t == "GaussNewton") reg.setLSQType(fast_gicp::LSQ_OPTIMIZER_TYPE::GaussNewton);
        else throw std::invalid_argument("unknown optimizer type " + t + " (LM, GN)");
      })
      .def("evaluate_cost", [](Lsq& reg, const Mat4& pose) {
        Matrix6d H; Vector6d b;
        const double e = reg.evaluateCost(numpy2mat4(pose), &H, &b);
        py::array_t<double> Hn({6, 6}), bn(6);
        std::memcpy(Hn.mutable_data(), H.data(), 36 * sizeof(double));
        std::memcpy(bn.mutable_data(), b.data(), 6 * sizeof(double));
        return py::make_tuple(e, Hn, bn);
      })
      .def("align", [](Lsq& reg, const Mat4& initial_guess) {
        Cloud aligned;
        reg.align(aligned, numpy2mat4(initial_guess));
        return mat4_to_numpy(reg.getFinalTransformation());
      }, py::arg("initial_guess") = identity4());

  py::class_<VGICPCuda, Lsq, std::shared_ptr<VGICPCuda>>(m, "FastVGICPCuda")
      .def(py::init([](int device) { return std::make_shared<VGICPCuda>(device); }), py::arg("device") = 0)
      .def("set_resolution", &VGICPCuda::setResolution)
      .def("set_voxel_accumulation_mode", [](VGICPCuda& v, const std::string& mode) {  // FastVGICP::setVoxelAccumulationMode (fast_vgicp_impl.hpp:41-43)
        if (mode == "ADDITIVE") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::ADDITIVE);
        else if (mode == "ADDITIVE_WEIGHTED") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::ADDITIVE_WEIGHTED);
        else if (mode == "MULTIPLICATIVE") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::MULTIPLICATIVE);
        else throw std::invalid_argument("unknown voxel accumulation mode: " + mode);
      })
      .def("set_neighbor_search_method", [](VGICPCuda& v, const std::string& method, double radius) { v.setNeighborSearchMethod(search_method(method), radius); },
           py::arg("method") = "DIRECT1", py::arg("radius") = 1.5)
      .def("set_correspondence_randomness", &VGICPCuda::setCorrespondenceRandomness)
      .def("set_kernel_width", &VGICPCuda::setKernelWidth, py::arg("kernel_width"), py::arg("max_dist") = -1.0)
      .def("set_regularization_method", [](VGICPCuda& v, const std::string& s) { v.setRegularizationMethod(regularization_method(s)); })
      .def("set_nearest_neighbor_search_method", [](VGICPCuda& v, const std::string& s) { v.setNearestNeighborSearchMethod(nn_method(s)); })
      // not in the reference: CPU_PARALLEL_KDTREE (the default enum value) is served by the device's exact search unless the host tree is asked for
      .def("set_host_kdtree", &VGICPCuda::setHostKdTree, py::arg("on") = true)
      // not in the reference: the scan stream as a two-stage pipeline (include/fast_vgicp_hip.h: fvh_vgicp_prepare_source / _align_async ...).
      //   reg.align_async(); reg.prepare_next_source(next_scan); T = reg.align_wait(); reg.swap_source_and_target(); reg.adopt_prepared_source()
      // The numpy conversion, the staging copy and the device's sort / k-NN / covariances of the next scan all run while the LM kernel of the current pair does.
      .def("prepare_next_source", [](VGICPCuda& v, const Points& p, int stages) { v.prepareNextSource(numpy2cloud(p), stages); }, py::arg("points"), py::arg("stages") = 2)
      .def("adopt_prepared_source", &VGICPCuda::adoptPreparedSource)
      .def("align_async", [](VGICPCuda& v, const Mat4& initial_guess) { v.alignAsync(numpy2mat4(initial_guess)); }, py::arg("initial_guess") = identity4())
      .def("align_wait", [](VGICPCuda& v) { return mat4_to_numpy(v.alignWait()); });

  py::class_<GICP, Lsq, std::shared_ptr<GICP>>(m, "FastGICP")  // main.cpp:183-190
      .def(py::init([](int device) { return std::make_shared<GICP>(device); }), py::arg("device") = 0)
      .def("set_num_threads", &GICP::setNumThreads)
      .def("set_correspondence_randomness", &GICP::setCorrespondenceRandomness)
      .def("set_max_correspondence_distance", &GICP::setMaxCorrespondenceDistance)
      .def("set_regularization_method", [](GICP& g, const std::string& s) { g.setRegularizationMethod(regularization_method(s)); })
      // not in the reference's bindings (its C++ class has them: gicp/fast_gicp.hpp:60-70): (N, 3, 3) arrays
      .def("get_source_covariances", [](GICP& g) { return covs_to_numpy(g.getSourceCovariances()); })
      .def("get_target_covariances", [](GICP& g) { return covs_to_numpy(g.getTargetCovariances()); })
      .def("set_source_covariances", [](GICP& g, const py::array_t<double, py::array::c_style | py::array::forcecast>& c) { g.setSourceCovariances(numpy_to_covs(c)); })
      .def("set_target_covariances", [](GICP& g, const py::array_t<double, py::array::c_style | py::array::forcecast>& c) { g.setTargetCovariances(numpy_to_covs(c)); });

  // The reference's CPU class (main.cpp:192-196; as there it derives from FastGICP: set_num_threads, set_correspondence_randomness,
  // set_max_correspondence_distance, the covariance accessors), served by the GPU engine in its fp64 arithmetic; k IS honoured.
  py::class_<VGICP, GICP, std::shared_ptr<VGICP>>(m, "FastVGICP")
      .def(py::init([](int device) { return std::make_shared<VGICP>(device); }), py::arg("device") = 0)
      .def("set_max_correspondence_distance", &VGICP::setMaxCorrespondenceDistance)  // (FastVGICP never reads it: voxel correspondences)
      .def("set_resolution", &VGICP::setResolution)
      .def("set_voxel_accumulation_mode", [](VGICP& v, const std::string& mode) {  // FastVGICP::setVoxelAccumulationMode (fast_vgicp_impl.hpp:41-43)
        if (mode == "ADDITIVE") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::ADDITIVE);
        else if (mode == "ADDITIVE_WEIGHTED") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::ADDITIVE_WEIGHTED);
        else if (mode == "MULTIPLICATIVE") v.setVoxelAccumulationMode(fast_gicp::VoxelAccumulationMode::MULTIPLICATIVE);
        else throw std::invalid_argument("unknown voxel accumulation mode: " + mode);
      })
      .def("set_neighbor_search_method", [](VGICP& v, const std::string& method) { v.setNeighborSearchMethod(search_method(method)); }, py::arg("method") = "DIRECT1");

  py::class_<NDT, Lsq, std::shared_ptr<NDT>>(m, "NDTCuda")
      .def(py::init([](int device) { return std::make_shared<NDT>(device); }), py::arg("device") = 0)
      .def("set_neighbor_search_method", [](NDT& n, const std::string& method, double radius) { n.setNeighborSearchMethod(search_method(method), radius); },
           py::arg("method") = "DIRECT1", py::arg("radius") = 1.5)
      .def("set_resolution", &NDT::setResolution)
      .def("set_distance_mode", [](NDT& n, const std::string& s) {
        if (s == "P2D") n.setDistanceMode(NDTDistanceMode::P2D);
        else if (s == "D2D") n.setDistanceMode(NDTDistanceMode::D2D);
        else throw std::invalid_argument("unknown NDT distance mode " + s);
      })
      // not in the reference: the frame stream as a two-stage pipeline (as FastVGICPCuda's)
      .def("prepare_next_source", [](NDT& n, const Points& p) { n.prepareNextSource(numpy2cloud(p)); }, py::arg("points"))
      .def("adopt_prepared_source", &NDT::adoptPreparedSource)
      .def("align_async", [](NDT& n, const Mat4& initial_guess) { n.alignAsync(numpy2mat4(initial_guess)); }, py::arg("initial_guess") = identity4())
      .def("align_wait", [](NDT& n) { return mat4_to_numpy(n.alignWait()); });

  // testing hook: the host kd-tree behind NearestNeighborMethod::CPU_PARALLEL_KDTREE
  m.def("_kdtree_knn", [](const Points& points, int k) {
    auto cloud = numpy2cloud(points);
    const std::vector<float> xyz = detail::pack_xyz(*cloud);
    const int n = (int)cloud->size();
    host::KdTree tree(xyz.data(), n);
    py::array_t<int> out({(py::ssize_t)n, (py::ssize_t)k});
    int* o = out.mutable_data();
#pragma omp parallel for schedule(guided, 8) num_threads(host::omp_threads_for(n))
    for (int i = 0; i < n; i++) tree.knn(&xyz[3 * (size_t)i], k, o + (size_t)i * k);
    return out;
  });
  m.attr("__version__") = "dev";
}
