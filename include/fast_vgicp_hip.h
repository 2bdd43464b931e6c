/*
 * fast_vgicp_hip.h -- C ABI of libfast_vgicp_hip.so, the MI355X (gfx950) engine that drops in where
 * the reference's nvcc-built libfast_vgicp_cuda.so sits (the pimpl classes
 * fast_gicp::cuda::FastVGICPCudaCore and fast_gicp::cuda::NDTCudaCore).
 *
 * Reference files cited below are relative to koide3/fast_gicp @ 2024-10-22:
 *   [VC]  include/fast_gicp/cuda/fast_vgicp_cuda.cuh   (class FastVGICPCudaCore, lines 28-92)
 *   [VCU] src/fast_gicp/cuda/fast_vgicp_cuda.cu
 *   [NC]  include/fast_gicp/cuda/ndt_cuda.cuh          (class NDTCudaCore, lines 28-68)
 *   [NCU] src/fast_gicp/cuda/ndt_cuda.cu
 *
 * Conventions
 *   - Every function returns an int status (FVH_OK == 0); fvh_*_last_error() gives the message.
 *     Nothing throws across the ABI.  (The reference has no error reporting: asserts/abort.)
 *   - The library owns all device memory behind the opaque handle; inputs are copied
 *     (caller keeps its buffers), outputs go to caller-allocated buffers -- same ownership
 *     as the reference's std::vector-by-const-ref / out-param style.
 *   - A handle is used from one host thread at a time; distinct handles may be used
 *     concurrently.  Each handle owns one HIP stream; calls are synchronous unless noted.
 *   - Poses are 4x4 double, COLUMN-MAJOR, i.e. exactly Eigen::Isometry3d::data(); H is 6x6
 *     double column-major (Eigen::Matrix<double,6,6>::data(), it is symmetric), b is 6 doubles.
 *     Twist order is (rot xyz, trans xyz) as in the reference.
 *   - Point clouds are packed float xyz (Eigen::Vector3f layout, 12 B/point).
 *   - 3x3 covariances returned by getters are 9 floats column-major (Eigen::Matrix3f layout).
 *   - Enum values equal the reference's enum class ordinals (gicp_settings.hpp:7-11,
 *     ndt_settings.hpp:6) so a static_cast<int>() in the shim is enough.
 */
#ifndef FAST_VGICP_HIP_H
#define FAST_VGICP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fvh_vgicp fvh_vgicp; /* replaces fast_gicp::cuda::FastVGICPCudaCore */
typedef struct fvh_ndt fvh_ndt;     /* replaces fast_gicp::cuda::NDTCudaCore */

enum fvh_status { FVH_OK = 0, FVH_ERR_INVALID_ARGUMENT = 1, FVH_ERR_BAD_STATE = 2, FVH_ERR_HIP = 3, FVH_ERR_UNSUPPORTED = 4, FVH_ERR_COMM = 5 };

/* fast_gicp::RegularizationMethod (gicp_settings.hpp:7) */
enum fvh_regularization { FVH_REG_NONE = 0, FVH_REG_MIN_EIG = 1, FVH_REG_NORMALIZED_MIN_EIG = 2, FVH_REG_PLANE = 3, FVH_REG_FROBENIUS = 4 };
/* fast_gicp::NeighborSearchMethod (gicp_settings.hpp:9) */
enum fvh_neighbor_search { FVH_DIRECT27 = 0, FVH_DIRECT7 = 1, FVH_DIRECT1 = 2, FVH_DIRECT_RADIUS = 3 };
/* fast_gicp::NDTDistanceMode (ndt_settings.hpp:6) */
enum fvh_ndt_distance_mode { FVH_NDT_P2D = 0, FVH_NDT_D2D = 1 };
/* arithmetic: FP64 matches the CPU FastVGICP (default); FP32 runs the per-correspondence cost math in float like the CUDA path (sums are
 * still accumulated in fp64) on the engine's fp64-centred covariances; CUDA_COMPAT is FastVGICPCuda's arithmetic end to end -- FP32's cost
 * math + the k-NN covariances as covariance_estimation.cu:20-35 / covariance_regularization.cu:15-125 compute them (uncentred float sums
 * in neighbour-list order, Eigen's closed-form float eigen solver). VGICP handles only; RBF covariances and the voxel sums stay fp64. */
enum fvh_precision { FVH_COMPUTE_FP64 = 0, FVH_COMPUTE_FP32 = 1, FVH_COMPUTE_CUDA_COMPAT = 2 };

/* LsqRegistration parameters (lsq_registration_impl.hpp:9-22 defaults) for the device-resident optimiser loop */
typedef struct fvh_lm_params {
  int max_iterations;            /* 64   pcl max_iterations_ */
  double rotation_epsilon;       /* 2e-3 */
  double transformation_epsilon; /* 5e-4 */
  int lm_max_iterations;         /* 10 */
  double lm_init_lambda_factor;  /* 1e-9 */
  int optimizer;                 /* 0 LSQ_OPTIMIZER_TYPE::LevenbergMarquardt (step_lm, default), 1 GaussNewton (step_gn, lsq_registration_impl.hpp:108-121) */
} fvh_lm_params;

typedef struct fvh_lm_result {
  double T[16];        /* final pose x0, column-major double (reference casts it to float) */
  double H[36];        /* final_hessian_ (last accepted, undamped) */
  double final_error;  /* y0 of the last linearisation */
  int converged;       /* converged_ */
  int nr_iterations;   /* nr_iterations_ (index of the last outer iteration) */
  int num_linearize;   /* calls to linearize() */
  int num_error_evals; /* calls to compute_error() */
  int lm_failed;       /* 1 = "lm not converged!!" (lsq_registration_impl.hpp:69-72) */
  int num_launches;    /* kernel launches used by this align */
} fvh_lm_result;

void fvh_default_lm_params(fvh_lm_params* p);
int fvh_device_count(int* count);

/* Engine parameters of a handle (round 6): the routes, thresholds and watchdogs that rounds 1-5 read from the process environment each time
 * they were consulted. They are per-handle state now: a new handle starts from fvh_default_engine_params() -- the built-in defaults with
 * the FVH_* environment variables of INTEGRATION.md applied ONCE per process -- and fvh_*_set_engine_params() changes one handle only.
 * (Process-wide by nature and therefore NOT in here: how concurrent aligns split the device's co-resident workgroup slots, the XCD-local
 * hand-offs and the small-grid layout, which every handle of a process must agree on: FVH_SLOT_MAX_SPLIT, FVH_CONTENDED_SLOT_PCT,
 * FVH_XCD_LOCAL, FVH_SMALL_GRID_LAYOUT.) No reference counterpart. */
typedef struct fvh_engine_params {
  int struct_size;                       /* sizeof(fvh_engine_params) of the caller's header: set by fvh_default_engine_params, checked by the setters */
  /* Morton sort of a cloud */
  int sort_mode;                         /* 0 multi-kernel radix, 1 one workgroup, 2 cooperative kernel while no other handle has a gang kernel in flight, else the radix passes (default), 3 cooperative always */
  int sort_items;                        /* points per wave of the radix passes; 0 = by cloud size */
  int sort_fused_bits;                   /* two-launch passes (32k..256k points): 10 (default) or 9 key bits per pass, 0 = four-launch passes */
  int sort_two_pass_max;                 /* clouds up to this size take two 11-bit four-launch passes (262,144) */
  unsigned long long sort_coop_watchdog_ticks; /* 100 MHz ticks before the cooperative sort gives up and the one-workgroup sort redoes it (2,000,000; 0 = at once: test hook) */
  /* neighbour search */
  int knn_nearest_first_max_points;      /* clouds up to this size walk the boxes nearest first (65,536) */
  int knn_block;                         /* threads per workgroup of the k-NN kernel: 64 (default), 128, 256 */
  /* voxel map */
  int coherent_min_points;               /* clouds of this size and up are walked in Morton order (32,768) */
  int bitmap_min_points;                 /* maps of this size and up get an occupancy bitmap (300,000) */
  unsigned long long bitmap_max_bytes;   /* its budget (32 MiB) */
  /* the LM kernel */
  int persistent;                        /* 1 (default): one launch per align while the problem is latency-bound; 0: one launch per LM transition */
  unsigned long long persist_watchdog_ticks; /* 100 MHz ticks a hand-off may take before the launch aborts and the align is redone per transition (5,000,000; 0: test hook) */
  unsigned long long peer_watchdog_ticks;    /* ... for a peer rank's sums (200,000,000) */
  int lm_everywhere;                     /* 0 never, 1 (default) on grids of at most two workgroups per CU, 2 always: every workgroup runs the LM step itself */
  int cost_prio;                         /* -1 (default) by grid shape, 0 never, 1 always: s_setprio for the second workgroup of a CU */
  int cost_split;                        /* 1 (default): wave roles for NDT launches of at most one workgroup per CU */
  int cost_group_max;                    /* voxel lookups per work item, 1..4 (4) */
  int cost_max_blocks;                   /* grid cap of a cost launch (1,024) */
  long long cost_target_items;           /* work items a small problem is split into (131,072) */
  int zerocopy_result;                   /* 1 (default): the final LM state travels through mapped host memory */
  int host_wait_block;                   /* 0 (default): the host spins on the result word; 1: it sleeps in hipStreamSynchronize */
  unsigned long long result_query_spins; /* spins between two hipStreamQuery calls while waiting (1,024) */
  /* streams and uploads */
  int side_stream;                       /* 1 (default): swap_source_and_target rebuilds the target map on a second stream */
  unsigned long long pinned_upload_max;  /* host clouds up to this many bytes go through the handle's pinned staging buffer (8 MiB) */
  unsigned long long zerocopy_upload_max;/* ... and up to this many are read by the widening kernel straight over PCIe (1 MiB) */
  /* ApproximateVoxelGrid */
  int avg_fused;                         /* 1 (default): four launches up to 262,144 points; 0: round 2's six */
  /* exact 1-NN search (FastGICP correspondences) */
  int nn1_seed;                          /* 1 (default): the id a point found last time seeds the bound of its next search (identical ids); 0: every search from scratch */
} fvh_engine_params;
void fvh_default_engine_params(fvh_engine_params* p);

/* ---------------------------------------------------------------------------------------------
 * FastVGICPCudaCore
 * ------------------------------------------------------------------------------------------- */
/* [VC]:37 ctor ([VCU]:18-30: res 1.0, kernel 0.25/3.0, offsets {0,0,0}) / [VC]:38 dtor */
int fvh_vgicp_create(int device, fvh_vgicp** out);
int fvh_vgicp_destroy(fvh_vgicp* h);
const char* fvh_vgicp_last_error(const fvh_vgicp* h);

int fvh_vgicp_set_resolution(fvh_vgicp* h, double resolution);                               /* [VC]:40 */
int fvh_vgicp_set_kernel_params(fvh_vgicp* h, double kernel_width, double kernel_max_dist);  /* [VC]:41 */
/* DIRECT_RADIUS: offsets up to +-511 voxels per axis (they travel packed, 10 bits each); a larger radius is FVH_ERR_INVALID_ARGUMENT */
int fvh_vgicp_set_neighbor_search_method(fvh_vgicp* h, int method, double radius);           /* [VC]:42, [VCU]:41-94 */
int fvh_vgicp_set_precision(fvh_vgicp* h, int precision);                                    /* new */
int fvh_vgicp_get_engine_params(fvh_vgicp* h, fvh_engine_params* out);                       /* new: see fvh_engine_params */
int fvh_vgicp_set_engine_params(fvh_vgicp* h, const fvh_engine_params* p);
/* fast_gicp::VoxelAccumulationMode (gicp_settings.hpp:10) of the CPU FastVGICP (setVoxelAccumulationMode, fast_vgicp_impl.hpp:41-43;
 * the CUDA core only has the additive voxel): ADDITIVE / ADDITIVE_WEIGHTED -> AdditiveGaussianVoxel (fast_vgicp_voxel.hpp:105-122),
 * MULTIPLICATIVE -> MultiplicativeGaussianVoxel (:79-103: sum of C^-1 and C^-1 p, inverted at finalize). Takes effect at the next map build. */
enum fvh_voxel_accumulation_mode { FVH_VOXEL_ADDITIVE = 0, FVH_VOXEL_ADDITIVE_WEIGHTED = 1, FVH_VOXEL_MULTIPLICATIVE = 2 };
int fvh_vgicp_set_voxel_accumulation_mode(fvh_vgicp* h, int mode);

int fvh_vgicp_swap_source_and_target(fvh_vgicp* h);                                          /* [VC]:44, [VCU]:97-107 */
int fvh_vgicp_set_source_cloud(fvh_vgicp* h, const float* xyz, int n);                       /* [VC]:45 */
int fvh_vgicp_set_target_cloud(fvh_vgicp* h, const float* xyz, int n);                       /* [VC]:46 */
/* same from a HOST array of points `stride_floats` (3 or 4) floats apart -- e.g. &cloud->points[0].x of a pcl::PointCloud<pcl::PointXYZ>
 * (16-byte points): the PointT -> Vector3f repack of fast_vgicp_cuda_impl.hpp:87-90,116-119 disappears */
int fvh_vgicp_set_source_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride_floats);
int fvh_vgicp_set_target_cloud_strided(fvh_vgicp* h, const float* xyz, int n, int stride_floats);
/* same, but the cloud is already in device memory (stride_floats = 3 or 4); D2D copy on the handle's stream */
int fvh_vgicp_set_source_cloud_device(fvh_vgicp* h, const float* d_xyz, int n, int stride_floats);
int fvh_vgicp_set_target_cloud_device(fvh_vgicp* h, const float* d_xyz, int n, int stride_floats);

int fvh_vgicp_set_source_neighbors(fvh_vgicp* h, int k, const int* neighbors /* n*k */);     /* [VC]:48 */
int fvh_vgicp_set_target_neighbors(fvh_vgicp* h, int k, const int* neighbors);               /* [VC]:49 */
int fvh_vgicp_find_source_neighbors(fvh_vgicp* h, int k);                                    /* [VC]:50 brute-force k-NN incl. self */
int fvh_vgicp_find_target_neighbors(fvh_vgicp* h, int k);                                    /* [VC]:51 */

int fvh_vgicp_calculate_source_covariances(fvh_vgicp* h, int regularization);                /* [VC]:53 */
int fvh_vgicp_calculate_target_covariances(fvh_vgicp* h, int regularization);                /* [VC]:54 */
int fvh_vgicp_calculate_source_covariances_rbf(fvh_vgicp* h, int regularization);            /* [VC]:56 */
int fvh_vgicp_calculate_target_covariances_rbf(fvh_vgicp* h, int regularization);            /* [VC]:57 */
/* new: inject covariances computed elsewhere (9 doubles per point, symmetric, any major) */
int fvh_vgicp_set_source_covariances(fvh_vgicp* h, const double* covs9);
int fvh_vgicp_set_target_covariances(fvh_vgicp* h, const double* covs9);

int fvh_vgicp_get_num_source_points(const fvh_vgicp* h, int* n);
int fvh_vgicp_get_num_target_points(const fvh_vgicp* h, int* n);
int fvh_vgicp_get_source_neighbors(fvh_vgicp* h, int* k, int* neighbors /* n*k, may be NULL to query k */);
int fvh_vgicp_get_target_neighbors(fvh_vgicp* h, int* k, int* neighbors);
int fvh_vgicp_get_source_covariances(fvh_vgicp* h, float* covs9);                            /* [VC]:59 */
int fvh_vgicp_get_target_covariances(fvh_vgicp* h, float* covs9);                            /* [VC]:60 */
int fvh_vgicp_get_num_voxels(fvh_vgicp* h, int* num_voxels);
int fvh_vgicp_get_voxel_num_points(fvh_vgicp* h, int* num_points);                           /* [VC]:62 */
int fvh_vgicp_get_voxel_means(fvh_vgicp* h, float* means3);                                  /* [VC]:63 */
int fvh_vgicp_get_voxel_covs(fvh_vgicp* h, float* covs9);                                    /* [VC]:64 */
int fvh_vgicp_get_voxel_coords(fvh_vgicp* h, int* coords3);                                  /* new (same order as the three above) */
int fvh_vgicp_get_num_correspondences(fvh_vgicp* h, int* n);
int fvh_vgicp_get_voxel_correspondences(fvh_vgicp* h, int* pairs /* n*2: (source index, voxel index in getter order) */); /* [VC]:65 */

int fvh_vgicp_create_target_voxelmap(fvh_vgicp* h);                                          /* [VC]:67 */
int fvh_vgicp_update_correspondences(fvh_vgicp* h, const double* T16);                       /* [VC]:69 */
/* [VC]:71 -- H36/b6 may both be NULL (error only).  Reuses the correspondences and the
 * linearisation rotation of the last update_correspondences(). */
int fvh_vgicp_compute_error(fvh_vgicp* h, const double* T16, double* H36, double* b6, double* error);

/* new: the whole LsqRegistration::computeTransformation loop (lsq_registration_impl.hpp:53-168) on the device.
 * Normally ONE persistent kernel launch per call (every LM transition is a trip through an in-kernel barrier; the
 * result comes back through mapped pinned memory, result->num_launches == 1); one launch per LM transition when a
 * RCCL communicator is attached, when another handle of the process is aligning at the same moment, for > 4 M voxel
 * lookups per evaluation, or after the barrier watchdog aborted a persistent launch (FVH_PERSISTENT=0 forces it).
 * All routes produce bit-identical results. */
int fvh_vgicp_align(fvh_vgicp* h, const double* guess16, const fvh_lm_params* params, fvh_lm_result* result);
/* new: a scan stream as a two-stage pipeline (scan-to-scan odometry, src/kitti.cpp:95-128 / src/align.cpp:87-101 with the preparation of
 * scan k+1 hidden under the registration of scan k). The handle owns a SECOND stream and a prepared-source slot:
 *   fvh_vgicp_prepare_source_device  packs a device cloud into the slot and queues, on the second stream, the first `stages` stages of what
 *                                    the sequential loop runs for a new source: 1 = Morton order + exact k-NN (k), 2 = + covariances
 *                                    (rbf != 0: the RBF covariances instead of both; regularization as in
 *                                    fvh_vgicp_calculate_source_covariances), 3 = + the scan's own voxel map at the handle's resolution.
 *                                    Returns when everything is queued.
 *   fvh_vgicp_align_async / _wait    launch the LM kernel on the main stream / collect its result (together: fvh_vgicp_align). Between the
 *                                    two only fvh_vgicp_prepare_source_device may be called on this handle (anything else: FVH_ERR_BAD_STATE).
 *   fvh_vgicp_adopt_prepared_source  the slot becomes the source (stages = 1: the caller then calls fvh_vgicp_calculate_source_covariances);
 *                                    a map prepared with it travels with the cloud, and the next fvh_vgicp_swap_source_and_target makes
 *                                    that map the live one instead of building it ([VCU]:102-104's rebuild, already done).
 * The loop:  align_async; prepare_source_device(next scan); align_wait; swap_source_and_target; adopt_prepared_source.
 * The results are the sequential calls': with k-NN covariances bit for bit; RBF covariances are float sums in the cloud's spatial order, and a
 * cloud prepared beside a running LM kernel is ordered by the radix passes (a finer Morton key than the cooperative sort's, whose 1,024-thread
 * finish kernel fits on no CU that hosts LM workgroups): their last bits differ, poses agree to 1e-9. The bundled 17k pair: stages 2 is
 * best (+37 % registrations/s over the sequential loop), 3 and 1 within 4 % of it.
 * swap_source_and_target() on ANY handle now keeps the old target's map with its cloud: swapping back costs nothing, unless that cloud's
 * covariances changed after the map was built (then it is rebuilt, as the reference would).
 * Not on a multi-GPU handle (FVH_ERR_UNSUPPORTED). */
int fvh_vgicp_align_async(fvh_vgicp* h, const double* guess16, const fvh_lm_params* params);
int fvh_vgicp_align_wait(fvh_vgicp* h, fvh_lm_result* result);
int fvh_vgicp_prepare_source_device(fvh_vgicp* h, const float* d_xyz, int n, int stride_floats, int k, int regularization, int rbf, int stages);
int fvh_vgicp_prepare_source(fvh_vgicp* h, const float* xyz, int n, int stride_floats, int k, int regularization, int rbf, int stages); /* a HOST cloud, consumed before the call returns */
int fvh_vgicp_adopt_prepared_source(fvh_vgicp* h);
/* setDebugPrint(true) on the device LM: with the trace on, an align records one row per trial step -- {inner iteration i, y0,
 * yi, rho, lambda, |d|}, the columns LsqRegistration prints (lsq_registration_impl.hpp:143-149) -- fetched afterwards
 * (rows6 may be NULL to query the count). */
int fvh_vgicp_set_lm_trace(fvh_vgicp* h, int on);
int fvh_vgicp_get_lm_trace(fvh_vgicp* h, int* num_rows, double* rows6);
/* new: pcl::Registration::getFitnessScore(max_range) -- mean squared exact-NN distance of
 * (float)T * source to the target cloud (exact tile-culled 1-NN on device, fixed-order fp64 sum: bit-reproducible). */
int fvh_vgicp_fitness_score(fvh_vgicp* h, const double* T16, double max_range, double* score);

/* ---- FastGICP on the device: nearest-target-point correspondences instead of voxels, on the same handle.
 * Reference: fast_gicp::FastGICP (CPU/OpenMP only), include/fast_gicp/gicp/impl/fast_gicp_impl.hpp:
 *   update_correspondences :118-156 (query = trans.cast<float>() * p, exact 1-NN in the target, kept when the squared
 *   distance < corr_dist_threshold_^2; mahalanobis = (C_B + T C_A T^T)^-1), linearize :159-213, compute_error :216-240,
 *   setMaxCorrespondenceDistance = pcl::Registration (default float max, :18).
 * Needs both clouds and both covariance sets (calculate_*_covariances or set_*_covariances); no voxel map. ---- */
int fvh_vgicp_gicp_set_max_correspondence_distance(fvh_vgicp* h, double max_distance);
int fvh_vgicp_gicp_swap_source_and_target(fvh_vgicp* h);                                    /* fast_gicp_impl.hpp:56-62: swaps clouds, neighbours, covariances; no voxel map */
int fvh_vgicp_gicp_update_correspondences(fvh_vgicp* h, const double* T16);
int fvh_vgicp_gicp_compute_error(fvh_vgicp* h, const double* T16, double* H36, double* b6, double* error);
/* the whole FastGICP LM loop on the device: per LM transition one nearest-point search + one cost launch, no host round trip
 * (the search reads the pose of the next linearisation from the LM state on the device); same result struct as fvh_vgicp_align */
int fvh_vgicp_gicp_align(fvh_vgicp* h, const double* guess16, const fvh_lm_params* params, fvh_lm_result* result);
int fvh_vgicp_gicp_get_correspondences(fvh_vgicp* h, int* target_index_per_source_point /* num_source_points ints, -1 = none */);

/* new: per-kernel-class HIP-event timing on the handle's stream (for bench.py's roofline leg) */
int fvh_vgicp_profile_enable(fvh_vgicp* h, int on /* 0 off; 1 every kernel class; 2 the "cost" class only: two events per registration, the result still travels through mapped memory */);
int fvh_vgicp_profile_reset(fvh_vgicp* h);
/* kernel_class in {"cost", "knn", "cov", "rbf", "voxelmap", "fitness"}; sums over launches since reset */
int fvh_vgicp_profile_get(fvh_vgicp* h, const char* kernel_class, double* total_ms, int* launches);
int fvh_vgicp_synchronize(fvh_vgicp* h);
/* testing hooks: the bucket table is sized from the previous build's voxel count (4x, pow2); a wrong
 * hint must only cost a rebuild at the safe size (2 x N_t), never points. */
int fvh_vgicp_debug_set_voxel_hint(fvh_vgicp* h, int num_voxels);
int fvh_vgicp_debug_get_table_capacity(fvh_vgicp* h, int* capacity);
int fvh_vgicp_debug_get_skipped_points(fvh_vgicp* h, int* n);  /* target points of the current voxel map that belong to no voxel: non-finite or |coord| >= 2^20 voxels (they are skipped, never fatal) */
int fvh_vgicp_debug_get_persist_aborts(fvh_vgicp* h, int* n);  /* persistent-LM launches whose barrier watchdog fired (each was redone with one launch per LM transition) */
int fvh_vgicp_debug_get_persist_grid(fvh_vgicp* h, int* blocks, int* capacity);  /* workgroups of the last persistent-LM launch / co-resident workgroup capacity of the device for that kernel */
int fvh_debug_slot_pool(int device, int* reserved, int* active, int* recent);    /* the process-wide pool that splits those workgroup slots between concurrent aligns: all zero when nothing is in flight */
/* persistent LM kernel, XCD-local hand-offs (kernels_cost.hpp): *wanted = 1 while the process still asks for them, *placement_aborts =
 * launches that ended because the dispatcher had not placed the workgroups of a group on one XCD (3 of those switch the flavour off) */
int fvh_debug_xcd_local(int* wanted, int* placement_aborts);
/* test hook: how many Morton sorts this process has queued on each route -- {cooperative kernel, one workgroup, two-launch radix passes,
 * four-launch radix passes}. (The cooperative kernel is a gang kernel: it is not used while ANOTHER registration handle has one in flight.) */
int fvh_debug_sort_routes(int* counts4);

/* new: multi-GPU over RCCL (one process per GPU). With a communicator attached a VGICP handle shards INTERNALLY, like the peer path below:
 * every rank makes the same calls on the same FULL clouds; neighbour search and covariance estimation run on the rank's spatial tile (a
 * range of the cloud's Morton order) and the 32 B / point covariances are all-gathered (ncclAllGather); every cost evaluation walks the
 * rank's tile of the source and the 32-double normal-equation block (err, b, H, trial error) is all-reduced (ncclAllReduce) on the
 * handle's stream; the LM step then runs redundantly on every rank. The target voxel map is replicated. (NDT handles: either the caller
 * hands each rank its shard of the source and only the all-reduce is the library's, or -- fvh_ndt_set_source_tile -- every rank holds the
 * full source and the engine walks the rank's spatial tile.) */
int fvh_comm_unique_id(void* id128 /* 128 bytes out */);
int fvh_vgicp_comm_init(fvh_vgicp* h, const void* id128, int nranks, int rank);
int fvh_vgicp_comm_destroy(fvh_vgicp* h);

/* new: multi-GPU without RCCL -- peer-mapped exchange regions (one process per GPU on one node; also handles of one process).
 * north_star: "large scans shard by spatial tile across up to 8 GPUs ... all-reduce of the 6x6/6x1 normal equations per
 * iteration".  With peers attached, every rank makes the SAME sequence of calls on the SAME full clouds and the engine shards
 * internally by spatial tile (= a range of the cloud's Morton order):
 *   find_*_neighbors / calculate_*_covariances[_rbf]: each rank computes its tile (the whole sorted cloud is the candidate set:
 *     an exact, implicit halo), then the tiles are all-gathered through the peers' staging areas -> every rank holds exactly
 *     the covariances one GPU would have computed (voxel map: built from them on every rank, replicated);
 *   update_correspondences / compute_error / align: each rank walks its tile of the source; the 32-double block {err, b, H} is
 *     exchanged through the peers' mailboxes INSIDE the cost kernel (every rank writes its block into every peer's mailbox over
 *     xGMI and sums in rank order -> bit-identical sums, the LM step runs redundantly, no broadcast): a sharded align is still
 *     ONE persistent launch per rank.  Measured mailbox round trip (two processes, one MI355X): 1.4 us.
 * peer_export: allocates the region (fine-grained device memory; staging for clouds of up to max_points) and returns its
 *   hipIpcMemHandle_t as 64 opaque bytes (+ the raw device pointer for peers living in the same process).
 * peer_attach: handles of all ranks in rank order (the own entry is ignored); process_local_ptrs[p] != 0 marks a peer of THIS
 *   process (its pointer is used directly, IPC cannot map one's own allocation); ranks_on_this_device: how many of the ranks
 *   share this GPU (their persistent grids share its co-resident workgroup slots).
 * A rank that does not arrive within the watchdog (2 s, FVH_PEER_WATCHDOG_TICKS) fails the call with FVH_ERR_COMM on every rank. */
int fvh_vgicp_peer_export(fvh_vgicp* h, int max_points, void* ipc_handle64 /* out */, unsigned long long* process_local_ptr /* out, may be NULL */);
int fvh_vgicp_peer_attach(fvh_vgicp* h, int nranks, int rank, int ranks_on_this_device, const void* ipc_handles /* nranks x 64 B, may be NULL if all peers are local */,
                          const unsigned long long* process_local_ptrs /* nranks entries or NULL */);
int fvh_vgicp_peer_detach(fvh_vgicp* h);
/* peer_attach refuses (FVH_ERR_COMM) a peer region on a device this one cannot reach (hipDeviceCanAccessPeer).
 * peer_selfcheck: COLLECTIVE -- every rank stores a nonce into every attached region and waits until every rank's nonce has arrived
 *   in its own (the path the in-kernel mailboxes take, over xGMI between devices); FVH_ERR_COMM + the bit mask of the ranks whose
 *   store never came after timeout_seconds (<= 0: 5 s). Call it once after attaching, before the first sharded registration. */
int fvh_vgicp_peer_selfcheck(fvh_vgicp* h, double timeout_seconds, int* missing_rank_mask /* out, may be NULL */);
/* Multi-GPU, either exchange (peers attached or a communicator): shard the TARGET voxel map too ("sharded by the same tiles +
 * halo"). With `on`, fvh_vgicp_align builds this rank's map from the target points around T_guess * (its spatial tile of the source) only:
 * the tile's bounding box in voxel coordinates, widened by the reach of the neighbour offsets (1 voxel for DIRECT7 / 27, ceil(r) for
 * DIRECT_RADIUS) + `margin_voxels` for the motion of the pose during the align. A voxel inside the box receives all its points: its record
 * is the replicated map's. If a source element leaves the inner box during the LM loop (the pose moved further than the margin allows)
 * the rank redoes that align on the full map. The map is rebuilt per align (it depends on the guess): meant for maps that should not be
 * replicated, not for speed. The host-driven calls (update_correspondences / compute_error / the voxel getters) use the replicated map:
 * after a sharded align they rebuild it (and the shard's correspondences die with it). */
int fvh_vgicp_set_target_map_sharding(fvh_vgicp* h, int on, int margin_voxels);
int fvh_vgicp_debug_get_map_shard(fvh_vgicp* h, int* live_map_is_a_shard, int* aligns_redone_on_the_full_map);
/* voxel count of the live map as the last align left it (a shard stays a shard). The voxel GETTERS (fvh_vgicp_get_num_voxels, get_voxel_*),
 * update_correspondences and compute_error always work on the WHOLE map: after a sharded align they rebuild it first. */
int fvh_vgicp_debug_get_live_map_voxels(fvh_vgicp* h, int* num_voxels);
/* test hook: the spatial (Morton) order of a cloud -- order[j] = original index of the j-th point of the sorted cloud (n ints) -- and the
 * boxes of its 64-point tiles (8 floats per tile: min xyz 0, max xyz 0). which: 0 source, 1 target. Sorts the cloud if it is not sorted yet.
 * No reference counterpart (the order is this engine's own device for culling the exact searches and for the multi-GPU tiles). */
int fvh_vgicp_debug_get_spatial_order(fvh_vgicp* h, int which, int* order, float* tile_boxes);

/* ---------------------------------------------------------------------------------------------
 * NDTCudaCore
 * ------------------------------------------------------------------------------------------- */
int fvh_ndt_create(int device, fvh_ndt** out);                                               /* [NC]:34-35, [NCU]:13-23 (D2D, DIRECT7, res 1.0) */
int fvh_ndt_destroy(fvh_ndt* h);
const char* fvh_ndt_last_error(const fvh_ndt* h);
int fvh_ndt_set_distance_mode(fvh_ndt* h, int mode);                                         /* [NC]:37 */
int fvh_ndt_set_resolution(fvh_ndt* h, double resolution);                                   /* [NC]:38 */
int fvh_ndt_set_neighbor_search_method(fvh_ndt* h, int method, double radius);               /* [NC]:39 */
int fvh_ndt_set_precision(fvh_ndt* h, int precision);
int fvh_ndt_get_engine_params(fvh_ndt* h, fvh_engine_params* out);
int fvh_ndt_set_engine_params(fvh_ndt* h, const fvh_engine_params* p);
int fvh_ndt_swap_source_and_target(fvh_ndt* h);                                              /* [NC]:41, [NCU]:90-93 */
int fvh_ndt_set_source_cloud(fvh_ndt* h, const float* xyz, int n);                           /* [NC]:42 (invalidates the source voxel map) */
int fvh_ndt_set_target_cloud(fvh_ndt* h, const float* xyz, int n);                           /* [NC]:43 */
int fvh_ndt_set_source_cloud_strided(fvh_ndt* h, const float* xyz, int n, int stride_floats);
int fvh_ndt_set_target_cloud_strided(fvh_ndt* h, const float* xyz, int n, int stride_floats);
int fvh_ndt_set_source_cloud_device(fvh_ndt* h, const float* d_xyz, int n, int stride_floats);
int fvh_ndt_set_target_cloud_device(fvh_ndt* h, const float* d_xyz, int n, int stride_floats);
int fvh_ndt_create_voxelmaps(fvh_ndt* h);                                                    /* [NC]:45 (lazy; source skipped in P2D) */
int fvh_ndt_create_target_voxelmap(fvh_ndt* h);                                              /* [NC]:46 */
int fvh_ndt_create_source_voxelmap(fvh_ndt* h);                                              /* [NC]:47 */
int fvh_ndt_update_correspondences(fvh_ndt* h, const double* T16);                           /* [NC]:49 */
int fvh_ndt_compute_error(fvh_ndt* h, const double* T16, double* H36, double* b6, double* error); /* [NC]:50 */
int fvh_ndt_align(fvh_ndt* h, const double* guess16, const fvh_lm_params* params, fvh_lm_result* result);
/* new: a frame stream as a two-stage pipeline (src/kitti.cpp:95-128 with the preparation of frame k+1 hidden under the registration of
 * frame k). The handle owns a SECOND stream and a prepared-source slot:
 *   fvh_ndt_align_async          launches the LM kernel on the main stream and returns; fvh_ndt_align_wait collects its result
 *                                (together they are fvh_ndt_align, bit for bit). Between the two, only fvh_ndt_prepare_source_device
 *                                (and a voxel-grid filter sharing the prepare stream) may be called on this handle.
 *   fvh_ndt_prepare_source_device widens a device cloud into the slot and (D2D) builds its voxel map there, all on the second stream;
 *                                returns when everything is queued.
 *   fvh_ndt_adopt_prepared_source the slot becomes the source (what set_source_cloud + the lazy map build of the next align would have
 *                                made: same kernels, same data, bit-identical maps); the main stream is ordered after the preparation.
 * The loop:  adopt; align_async; [filter next frame on the prepare stream; prepare_source_device]; align_wait; swap_source_and_target. */
int fvh_ndt_align_async(fvh_ndt* h, const double* guess16, const fvh_lm_params* params);
int fvh_ndt_align_wait(fvh_ndt* h, fvh_lm_result* result);
int fvh_ndt_prepare_source_device(fvh_ndt* h, const float* d_xyz, int n, int stride_floats);
int fvh_ndt_prepare_source(fvh_ndt* h, const float* xyz, int n, int stride_floats); /* the same for a HOST cloud, consumed before the call returns */
int fvh_ndt_adopt_prepared_source(fvh_ndt* h);
/* new (round 6): the output of the filter's last ApproximateVoxelGrid call becomes the cloud WITHOUT a copy -- the filter's emit kernel wrote it as
 * float4 too, and that buffer is swapped with the cloud's (what fvh_voxelgrid_device_points + fvh_ndt_set_source_cloud_device /
 * _prepare_source_device do, minus the widening kernel and its launch: 8 of a LiDAR frame's 141 us). An output can be taken once; the filter's
 * packed xyz (fvh_voxelgrid_get_points / _device_points) stays valid. The streams of the two handles are ordered if they differ. */
struct fvh_voxelgrid;
int fvh_ndt_set_source_cloud_from_voxelgrid(fvh_ndt* h, struct fvh_voxelgrid* filter);
int fvh_ndt_set_target_cloud_from_voxelgrid(fvh_ndt* h, struct fvh_voxelgrid* filter);
int fvh_ndt_prepare_source_from_voxelgrid(fvh_ndt* h, struct fvh_voxelgrid* filter);
int fvh_ndt_fitness_score(fvh_ndt* h, const double* T16, double max_range, double* score);
/* testing hook (as fvh_vgicp_debug_set_voxel_hint): table size hint of the NEXT build of the source (which = 0) / target (1) voxel map */
int fvh_ndt_debug_set_voxel_hint(fvh_ndt* h, int which, int num_voxels);
int fvh_ndt_set_lm_trace(fvh_ndt* h, int on);
int fvh_ndt_get_lm_trace(fvh_ndt* h, int* num_rows, double* rows6);
int fvh_ndt_get_num_voxels(fvh_ndt* h, int which /* 0 source, 1 target */, int* num_voxels);
int fvh_ndt_get_voxels(fvh_ndt* h, int which, int* coords3, int* num_points, float* means3, float* covs9);
int fvh_ndt_get_num_correspondences(fvh_ndt* h, int* n);
/* debug getter of the list behind [NC]:67 (`correspondences`, ndt_cuda.cu:142-161), offset-major, invalid pairs removed:
 * (source element, target voxel) -- the source element is a point index (P2D) or an index into fvh_ndt_get_voxels(h, 0, ...)
 * (D2D), the target voxel an index into fvh_ndt_get_voxels(h, 1, ...). 2 * fvh_ndt_get_num_correspondences ints. */
int fvh_ndt_get_voxel_correspondences(fvh_ndt* h, int* pairs);
int fvh_ndt_profile_enable(fvh_ndt* h, int on);                                              /* HIP-event timing per kernel class, as fvh_vgicp_profile_* */
int fvh_ndt_profile_reset(fvh_ndt* h);
int fvh_ndt_profile_get(fvh_ndt* h, const char* kernel_class, double* total_ms, int* launches);
int fvh_ndt_synchronize(fvh_ndt* h);
int fvh_ndt_comm_init(fvh_ndt* h, const void* id128, int nranks, int rank);
/* new: multi-GPU NDT by spatial tile. The handle keeps the FULL clouds (target map replicated) and evaluates tile `rank` of `nranks` of the
 * source: P2D -- that chunk of the source points' Morton order (as a sharded VGICP handle); D2D -- that chunk of the source map's voxels
 * ranked by voxel key (a canonical order: the compact list [NCU]:142-161 walks is ordered by atomics and differs between two builds).
 * fvh_ndt_update_correspondences / fvh_ndt_compute_error then return the tile's PARTIAL err / H / b, to be added over the ranks by the
 * caller; with a communicator of the same nranks attached fvh_ndt_align all-reduces them on the device (without one it refuses).
 * nranks = 1 switches the tile off. The correspondence getters and fvh_ndt_align_async are not available on a tiled handle. */
int fvh_ndt_set_source_tile(fvh_ndt* h, int rank, int nranks);
int fvh_ndt_comm_destroy(fvh_ndt* h);

/* ---------------------------------------------------------------------------------------------------
 * Voxel-grid downsampling on device -- the filter every caller of the reference runs on the raw scan
 * right before the registration path: pcl::ApproximateVoxelGrid in src/align.cpp:136-147, src/kitti.cpp:80-82,117-119
 * and pygicp's downsample()/align_points() (src/python/main.cpp:46-62,80-92); pcl::VoxelGrid in
 * src/test/gicp_test.cpp:55-65.  Output points AND their order are bit-identical to those filters on PointXYZ
 * (fp32 sums in input order).  Non-finite coordinates are rejected (FVH_ERR_INVALID_ARGUMENT).
 * The output stays on the device (packed xyz, 3 floats per point) until the next filter call, so it can be handed
 * straight to fvh_*_set_*_cloud_device(ptr, n, 3).
 * --------------------------------------------------------------------------------------------------- */
typedef struct fvh_voxelgrid fvh_voxelgrid;
enum fvh_voxelgrid_method { FVH_VOXELGRID_EXACT = 0 /* pcl::VoxelGrid */, FVH_VOXELGRID_APPROXIMATE = 1 /* pcl::ApproximateVoxelGrid */ };
int fvh_voxelgrid_create(int device, fvh_voxelgrid** out);
int fvh_voxelgrid_get_engine_params(fvh_voxelgrid* h, fvh_engine_params* out);
int fvh_voxelgrid_set_engine_params(fvh_voxelgrid* h, const fvh_engine_params* p);
int fvh_voxelgrid_destroy(fvh_voxelgrid* h);
const char* fvh_voxelgrid_last_error(const fvh_voxelgrid* h);
int fvh_voxelgrid_filter(fvh_voxelgrid* h, int method, const float* xyz, int n, float leaf, int* out_n);                       /* setLeafSize(l,l,l); setInputCloud; filter */
int fvh_voxelgrid_filter_strided(fvh_voxelgrid* h, int method, const float* xyz, int n, int stride_floats /* 3, or 4 for xyzi (KITTI .bin, kitti.cpp:48-60) */, float leaf, int* out_n);
int fvh_voxelgrid_filter_device(fvh_voxelgrid* h, int method, const float* d_xyz, int n, int stride_floats, float leaf, int* out_n);
/* Device-resident pipeline (no PCL counterpart; kitti.cpp:95-128 with every stage on the GPU): run the filter on the registration
 * handle's stream (null: back to its own) -- its output is then ordered before whatever that handle queues next -- and let
 * _async return as soon as the output COUNT is known, while the last kernel of the filter still runs: the caller hands
 * fvh_voxelgrid_device_points() to fvh_*_set_*_cloud_device() of THAT handle. Without a shared stream _async is the synchronous call.
 * The registration handle must outlive the sharing (un-share or destroy the filter first). */
int fvh_voxelgrid_share_stream_with_ndt(fvh_voxelgrid* h, fvh_ndt* registration);
int fvh_voxelgrid_share_stream_with_vgicp(fvh_voxelgrid* h, fvh_vgicp* registration);
/* same, on the registration handle's SECOND stream (the one fvh_ndt_prepare_source_device works on): the filter of the next frame runs
 * beside the LM kernel of the current one, its output is ordered before the preparation that consumes it */
int fvh_voxelgrid_share_prepare_stream_with_ndt(fvh_voxelgrid* h, fvh_ndt* registration);
int fvh_voxelgrid_share_prepare_stream_with_vgicp(fvh_voxelgrid* h, fvh_vgicp* registration); /* ... the one fvh_vgicp_prepare_source_device works on */
int fvh_voxelgrid_filter_device_async(fvh_voxelgrid* h, int method, const float* d_xyz, int n, int stride_floats, float leaf, int* out_n);
int fvh_voxelgrid_get_points(fvh_voxelgrid* h, float* out_xyz /* host or device, 3*out_n floats */);
int fvh_voxelgrid_device_points(fvh_voxelgrid* h, const float** d_xyz, int* n);
int fvh_voxelgrid_profile_enable(fvh_voxelgrid* h, int on);
int fvh_voxelgrid_profile_reset(fvh_voxelgrid* h);
int fvh_voxelgrid_profile_get(fvh_voxelgrid* h, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* FAST_VGICP_HIP_H */
