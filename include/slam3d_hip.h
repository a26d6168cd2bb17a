/* slam3d_hip.h — C ABI of the MI355X registration back-end (libslam3d_hip.so).
 *
 * The reference (dfki-ric/slam3d) has no FFI: its plugin contract is the C++
 * virtual interface slam3d::Sensor / ScanSensor / PointCloudSensor
 * (slam3d/core/ScanSensor.hpp:122-125, slam3d/sensor/pcl/PointCloudSensor.hpp:106-243).
 * This header is the boundary a maintainer binds instead of PCL: every entry
 * point names the reference code it replaces.  Plain pointers and sizes only;
 * host pointers are borrowed for the duration of the call, results are returned
 * by value into caller buffers, no device pointer crosses the ABI except through
 * the explicit s3d_cloud handle.  The C++ mirror of the reference classes that
 * sits on top of this ABI is cpp/slam3d/sensor/pcl/PointCloudSensor.hpp; the
 * binding stub is shown in INTEGRATION.md.
 *
 * There is no CPU fallback: every call fails with S3D_STATUS_BACKEND_ERROR when
 * no HIP device is usable.
 *
 * Matrices: 4x4 double, COLUMN-major (the memory layout of slam3d::Transform =
 * Eigen::Transform<double,3,Isometry>, slam3d/core/Types.hpp:53).
 * Point clouds: float32, `stride` floats per point (3 = packed xyz,
 * 4 = pcl::PointXYZ / KITTI layout), xyz first.
 */
#ifndef SLAM3D_HIP_H
#define SLAM3D_HIP_H

#include "slam3d_registration_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct s3d_context s3d_context; /* one HIP device + stream + workspace; calls on one context are serialised */
typedef struct s3d_cloud   s3d_cloud;   /* a device-resident point cloud (immutable, like a slam3d Measurement)   */

/* knobs that are not part of slam3d::RegistrationParameters */
typedef struct s3d_exec_options {
  int force_iterations;     /* != 0: run exactly maximum_iterations outer iterations (bench mode, no early exit) */
  int check_interval;       /* 0 (default): the device reports its progress, the host never waits inside the loop;
                               N > 0: the host polls "all pairs converged" every N outer iterations instead       */
  int grid_cells_per_point; /* search-grid budget, cells per input point (0 = default 2)                         */
  int profile;              /* != 0: record per-stage HIP-event timings, read with s3d_last_profile();
                               >= 2: also count the searched queries per NN launch (slows the first passes)      */
  int cache_prepass;        /* 0 (default): like the reference, every call voxel-filters, grids and runs the k-NN
                               pre-pass of both clouds again (PointCloudSensor.cpp:127-131, "no caching").
                               1: keep those per-cloud products in HBM, keyed by (s3d_cloud, point_cloud_density,
                               grid budget, correspondence_randomness), and reuse them in later calls on the same
                               context - the mapper pattern (ScanSensor.cpp:113 links every new scan to the previous
                               one, :179-201 to its neighbours) then pays the pre-pass once per scan.  Results are
                               bit-identical either way.  Entries die with s3d_cloud_release(ctx, cloud); see
                               s3d_context_cache_control.  A cloud must not be modified while it is cached.       */
  int omp_unavailable;      /* GICP_OMP / NDT_OMP.  The reference has two behaviours, chosen when IT is built
                               (PointCloudSensor.cpp:149-162): with the external pclomp package the two enumerators
                               run pclomp's multi-threaded GICP / NDT - GICP_OMP the same objective as GICP, NDT_OMP
                               the NDT objective over pclomp's DIRECT7 neighbourhood - and
                               without it align() throws std::runtime_error("OMP is not available, ...").
                               0 (default): the pclomp build - GICP_OMP is served by the GICP device code, NDT_OMP by the
                               NDT code with the seven-voxel neighbourhood.
                               1: the build without pclomp - S3D_STATUS_OMP_UNAVAILABLE after the voxel filter and
                               the 100-point gate (the reference's order), which the C++ mirror re-raises.           */
  unsigned int debug_flags; /* S3D_DBG_* bits below; 0 = the product's behaviour.  Every bit switches ONE fast path off
                               (or forces one of two equivalent forms) and must not change a bit of any result - that is
                               what the tests use them for.  Read once per call from this struct: the library reads no
                               environment variable on any registration path.                                         */
  int debug_accum_blocks;   /* 0 = automatic; 1, 2, 4 ... 32 (a divisor of the 32 VIRTUAL blocks the sums of a pair are
                               defined over): REAL blocks per pair in the accumulate kernels - speed only, the
                               invariance test sets it.  Any other value: S3D_STATUS_INVALID_ARGUMENT                   */
} s3d_exec_options;

/* The S3D_DBG_* bits of s3d_exec_options.debug_flags are test / measurement switches, not API: they live in
 * slam3d_hip_debug.h, which this header does not include (the library and the tests do). */

/* The structs above grow at the END only.  A binding compiled against another revision of this header must not be
 * used: compare S3D_ABI_VERSION with s3d_abi_version() once after loading the library (the Python binding and the
 * C++ mirror do). */
#define S3D_ABI_VERSION 4
int  s3d_abi_version(void);
/* sha256 (hex) of the sources this binary was compiled from (csrc/Makefile: s3d_api.hip, its headers and the three
 * public headers, in that order).  A binding that sits next to the sources compares and refuses a stale binary
 * (slam3d_amd/api.py); measurements attached to a build (profiles/nn_traffic.json) are keyed by it. */
const char* s3d_source_hash(void);

typedef struct s3d_align_info {   /* diagnostics of one align() */
  int    n_source_filtered, n_target_filtered;   /* points after the voxel filter            */
  int    iterations, converged, correspondences; /* outer iterations, pcl hasConverged()      */
  double fitness;                                /* pcl getFitnessScore(max_corr_distance)    */
  int    inner_iterations, evaluations;          /* BFGS steps / objective evaluations (GICP) */
} s3d_align_info;

typedef struct s3d_profile {      /* milliseconds, HIP events on the context's stream */
  double voxel_ms, grid_ms, normals_ms, icp_ms, fitness_ms, total_ms;
  double nn_ms;       /* sum over the NN-search kernel launches of the ICP loop */
  int    nn_launches;
  long long nn_queries, nn_targets; /* summed over launches: queries searched, target points indexed */
  float  nn_launch_ms[64];          /* duration of the first 64 NN launches of the ICP loop, in order */
  int    nn_searched[64];           /* profile >= 2: queries of launch i that needed a grid search (not re-validated) */
  int    nn_unseeded[64];           /* ... of which without a usable previous neighbour (wide search)            */
  int    nn_records[64];            /* profile >= 2, settled passes: 64-query records tested by launch i ...         */
  int    nn_records_searched[64];   /* ... of which failed the record-level proof and ran query by query             */
} s3d_profile;

/* ---- context ------------------------------------------------------------------ */
/* hip_stream: a hipStream_t to run on (e.g. torch's current stream) or NULL for a private stream. */
int  s3d_context_create(int device, void* hip_stream, s3d_context** out);
/* A context on a private stream of the given priority class: 0 = default, 1 = the device's highest.  The reference is
 * entered from two threads at once (ScanSensor.cpp:209-210: the application thread registers every new scan against
 * the previous one, a detached thread runs linkToNeighbors), and one s3d_context serialises its callers: the
 * latency-critical sequential registration belongs on a context of its own, not on the one a loop-closure batch runs
 * on (measured: 6.2 ms instead of 15.7 ms next to a 128-pair batch, 1.6 ms idle).  Whether the device's dispatcher
 * honours the stream priority on top of that is up to the driver - on the MI355X pool this was built on it made no
 * measurable difference (tests/test_gpu_sweep.py::test_sequential_registration_next_to_a_batch).  Results do not
 * depend on it. */
int  s3d_context_create_priority(int device, int priority_class, s3d_context** out);
/* A context on a private stream that may only use the compute units whose bit is set in cu_mask (n_words x 32 bits,
 * bit i of word w = CU 32 w + i in the driver's enumeration; hipExtStreamCreateWithCUMask): the way to RESERVE part of
 * the GPU for the sequential registration while a loop-closure sweep runs on the complementary mask
 * (s3d_sweep_create_cu_mask).  Measured on an MI355X with 32 of the 256 CUs reserved: one 100 k-point pair next to a
 * running 128-pair batch 2.1 ms instead of 5.8 ms (1.4 ms on the idle GPU, 1.85 ms alone on its 32 CUs), the batch
 * 17.0 instead of 15.6 ms.  Results do not depend on masks.
 * s3d_cu_masks fills the two complementary masks for `reserved_cus` compute units (the first ones of the driver's
 * enumeration) and returns the number of words, or -1 (no such device, max_words too small, 0 < reserved < CUs
 * violated). */
int  s3d_cu_masks(int device, int reserved_cus, uint32_t* reserved_mask, uint32_t* rest_mask, int max_words);
int  s3d_context_create_cu_mask(int device, const uint32_t* cu_mask, int n_words, s3d_context** out);
void s3d_context_destroy(s3d_context* ctx);
const char* s3d_last_error(const s3d_context* ctx);
/* fills "name|gcnArch|CUs|HBM bytes"; returns S3D_STATUS_BACKEND_ERROR without a device */
int  s3d_backend_info(int device, char* buf, int len);
int  s3d_last_profile(const s3d_context* ctx, s3d_profile* out);
/* the cross-call pre-pass cache of a context (s3d_exec_options.cache_prepass): limit_bytes > 0 sets the HBM budget
 * (default 16 GiB, least-recently-used entries are dropped first), clear != 0 drops every entry.  stats (may be
 * NULL): entries, bytes, hits, misses (per cloud and call, since the context was created). */
typedef struct s3d_cache_stats { long long entries, bytes, hits, misses; } s3d_cache_stats;
int  s3d_context_cache_control(s3d_context* ctx, long long limit_bytes, int clear, s3d_cache_stats* stats);
/* Checkpoints.  The reference saves a graph as slam3d_graph.yml + one <index>.s3dm archive per vertex
 * (GraphSerialization.cpp:14-66) and reloads it with fromFolder (:68-135); a reloaded PointCloudMeasurement is a new
 * object, so its device copy and its cached pre-pass products are gone.  s3d_cloud_cache_export writes every cache
 * entry of `cloud` (filtered points, cell-sorted copies, cell table, normals, grid / voxel parameters) into a
 * caller-provided host buffer that can be stored next to the .s3dm file; it returns the byte size needed (0: nothing
 * is cached for this cloud, < 0: a status code) and writes only when `capacity` suffices.  s3d_cloud_cache_import
 * installs such a blob as the cache entries of `cloud` - a cloud with the same points, which is checked (point count
 * and a hash of the coordinates; S3D_STATUS_INVALID_ARGUMENT otherwise, nothing installed) - so that the first
 * registration after a reload starts from the cached products.  Results are bit-identical with or without. */
long long s3d_cloud_cache_export(s3d_context* ctx, const s3d_cloud* cloud, void* buffer, long long capacity);
int  s3d_cloud_cache_import(s3d_context* ctx, const s3d_cloud* cloud, const void* blob, long long size);
void s3d_default_params(s3d_reg_params* p);          /* RegistrationParameters.hpp:36-97 defaults */

/* ---- device-resident clouds ----------------------------------------------------- */
int  s3d_cloud_upload(s3d_context* ctx, const float* xyz, int n, int stride, s3d_cloud** out);
/* Bulk hand-over: n_clouds host arrays (xyz[i]: n[i] points of `stride` floats, x y z first) -> n_clouds clouds in ONE
 * device allocation, which lives until the last of them is released (any order).  Host threads copy the arrays into
 * pinned memory while the float4 expansion of the previous ones reads it over PCIe: the archive of a reloaded graph
 * (GraphSerialization.cpp:68-135 rebuilds every measurement) or the scans of a loop-closure sweep are handed over at
 * the link's rate instead of one allocation + staged copy + wait per scan.  out[i] are ordinary clouds. */
int  s3d_cloud_upload_many(s3d_context* ctx, int n_clouds, const float* const* xyz, const int* n, int stride, s3d_cloud** out);
/* host threads s3d_cloud_upload_many may use (default 0 = up to 8): a process that runs one rank per GPU on a host whose
 * CPU quota is shared by the ranks (ScanSensor.cpp:179-201 as an 8-rank sweep: 16 CPUs / 8 ranks) gives each context its
 * share, so that the hand-over threads of the ranks do not oversubscribe it.  n < 0: INVALID_ARGUMENT. */
int  s3d_context_set_upload_threads(s3d_context* ctx, int n);
/* wrap n float4 (x,y,z,*) already in HBM (e.g. a torch tensor); not copied, not freed */
int  s3d_cloud_wrap_device(s3d_context* ctx, const void* device_float4, int n, s3d_cloud** out);
int  s3d_cloud_size(const s3d_cloud* c);
void s3d_cloud_release(s3d_context* ctx, s3d_cloud* c);

/* ---- A3  PointCloudSensor::downsample (PointCloudSensor.cpp:190-201, pcl::VoxelGrid) ----
 * out_xyz: capacity 3*n floats (packed).  Empty input -> *n_out = 0. */
int  s3d_voxel_downsample(s3d_context* ctx, const float* xyz, int n, int stride, double leaf_size,
                          float* out_xyz, int* n_out);

/* ---- A7  exact 1-NN (pcl::search::KdTree::nearestKSearch(q, 1), FLANN, inside align and
 *          getFitnessScore, call sites PointCloudSensor.cpp:70, :73).  Neighbours farther than
 *          max_distance need not be found (idx = -1 / a farther point).  Ties: lowest index. */
int  s3d_nn_search(s3d_context* ctx, const float* target_xyz, int n, int stride_t, const float* query_xyz, int m,
                   int stride_q, double max_distance, int* idx, float* d2);

/* ---- A6  GICP computeCovariances pre-pass (PointCloudSensor.cpp:63 setCorrespondenceRandomness):
 *          unit eigenvector of the smallest eigenvalue of the k-NN covariance, packed xyz.
 *          The regularised covariance PCL builds is C = I - (1 - 0.001) n n^T.
 *          Limit of the back-end: 1 <= k <= 64 (PCL accepts any k <= cloud size); k outside that range, here and as
 *          correspondence_randomness of a registration, returns S3D_STATUS_INVALID_ARGUMENT. */
int  s3d_knn_normals(s3d_context* ctx, const float* xyz, int n, int stride, int k, float* normals_xyz);

/* ---- A2  align() (PointCloudSensor.cpp:119-174): downsample both, 100-point gate, doICP
 *          (A4: :52-82), fitness gate, distance-from-guess gate.  Returns enum s3d_status;
 *          result is written for every status that got as far as the ICP. */
int  s3d_align(s3d_context* ctx, const float* source_xyz, int n_source, int stride_source,
               const float* target_xyz, int n_target, int stride_target, const double guess[16],
               const s3d_reg_params* params, const s3d_exec_options* opts, double result[16],
               s3d_align_info* info);

/* ---- A2 batched: many independent pairs (ScanSensor::linkToNeighbors candidates,
 *          ScanSensor.cpp:179-201) in lock-step launches.  The same cloud handle may appear in
 *          many pairs; its voxel filter / grid / normals are then computed once.
 *          guesses: 16*n_pairs doubles.  records: n_pairs (status inside).  infos: optional.
 *          Returns S3D_STATUS_OK unless the batch itself could not run. */
int  s3d_align_batch(s3d_context* ctx, int n_pairs, s3d_cloud* const* sources, s3d_cloud* const* targets,
                     const double* guesses, const s3d_reg_params* params, const s3d_exec_options* opts,
                     s3d_edge_record* records, s3d_align_info* infos);

/* ---- A1  PointCloudSensor::createConstraint (PointCloudSensor.cpp:269-299): frame algebra,
 *          optional coarse align (loop closures), fine align, (I * covariance_scale)^-1.
 *          information: 6x6 row-major. */
int  s3d_create_constraint(s3d_context* ctx, const float* source_xyz, int n_source, int stride_source,
                           const double source_sensor_pose[16], const float* target_xyz, int n_target,
                           int stride_target, const double target_sensor_pose[16], const double odometry[16],
                           int loop, const s3d_reg_params* fine, const s3d_reg_params* coarse,
                           double covariance_scale, const s3d_exec_options* opts, double relative_pose[16],
                           double information[36], s3d_align_info* info);

/* ==== the callers either side of the registration path ===================== */

/* copy a device-resident cloud back to the host: n * stride floats, xyz first (stride 4: w = 1) */
int  s3d_cloud_download(s3d_context* ctx, const s3d_cloud* c, float* xyz, int stride);

/* ---- B1  PointCloudSensor::getAccumulatedCloud (PointCloudSensor.cpp:235-256): every cloud transformed by
 *          its pose (correctedPose * sensorPose, formed by the caller; PointCloudSensor::transform, :228-233 =
 *          pcl::transformPointCloud with a Matrix4d) and appended in list order.  frame != NULL adds
 *          createCombinedMeasurement (:258-266): the accumulated cloud shifted by frame.inverse().  The result
 *          stays in HBM and is a valid source/target for s3d_align_batch / s3d_create_constraint_clouds, so a
 *          loop-closure patch (ScanSensor.cpp:150-151) never visits the host.
 *          poses: 16 * n_clouds doubles, column-major.  *out is owned by the caller (s3d_cloud_release). */
int  s3d_cloud_accumulate(s3d_context* ctx, int n_clouds, s3d_cloud* const* clouds, const double* poses,
                          const double frame[16], s3d_cloud** out);

/* ---- A2 / A1 on device-resident clouds (scans uploaded once, patches built by s3d_cloud_accumulate): same
 *          contracts as s3d_align / s3d_create_constraint. */
int  s3d_align_clouds(s3d_context* ctx, s3d_cloud* source, s3d_cloud* target, const double guess[16],
                      const s3d_reg_params* params, const s3d_exec_options* opts, double result[16],
                      s3d_align_info* info);
int  s3d_create_constraint_clouds(s3d_context* ctx, s3d_cloud* source, const double source_sensor_pose[16],
                                  s3d_cloud* target, const double target_sensor_pose[16], const double odometry[16],
                                  int loop, const s3d_reg_params* fine, const s3d_reg_params* coarse,
                                  double covariance_scale, const s3d_exec_options* opts, double relative_pose[16],
                                  double information[36], s3d_align_info* info);

/* ---- B2  PointCloudSensor::removeOutliers (:211-226, pcl::RadiusOutlierRemoval): keeps, in input order, the
 *          points with at least min_neighbors OTHER points within `radius` (float d2 <= r*r as PCL compares
 *          it).  radius <= 0 or min_neighbors == 0: the input comes back unchanged (:214).
 *          out_xyz: capacity 3*n floats (packed). */
int  s3d_remove_outliers(s3d_context* ctx, const float* xyz, int n, int stride, double radius, unsigned min_neighbors,
                         float* out_xyz, int* n_out);
int  s3d_remove_outliers_cloud(s3d_context* ctx, const s3d_cloud* in, double radius, unsigned min_neighbors,
                               s3d_cloud** out);
/* A3 on a device-resident cloud */
int  s3d_voxel_downsample_cloud(s3d_context* ctx, const s3d_cloud* in, double leaf_size, s3d_cloud** out);

/* ---- B3  PointCloudSensor::buildMap (:301-318): accumulate, removeOutliers(outlier_radius,
 *          outlier_neighbors), downsample(map_resolution) — all in HBM; *out_map is device-resident. */
typedef struct s3d_map_profile {   /* milliseconds, HIP events; counts of points between the stages */
  double accumulate_ms, grid_ms, count_ms, compact_ms, voxel_ms, total_ms;
  long long n_accumulated, n_kept, n_map;
} s3d_map_profile;
int  s3d_build_map(s3d_context* ctx, int n_clouds, s3d_cloud* const* clouds, const double* poses,
                   double outlier_radius, unsigned outlier_neighbors, double map_resolution, s3d_cloud** out_map);
int  s3d_last_map_profile(const s3d_context* ctx, s3d_map_profile* out);

/* ---- B4  PointCloudSensor::fillGroundPlane (:362-388).  s3d_fit_plane is the
 *          pcl::RandomSampleConsensus<pcl::SampleConsensusModelPlane>::computeModel call of :364-368 (the model's
 *          fixed seed, so the result is deterministic as in the reference); the inlier counting of every
 *          hypothesis runs on the device.  s3d_fill_ground_plane runs it with the reference's settings
 *          (threshold 0.01, PCL defaults 1000 iterations / probability 0.99) and returns the ring points of
 *          :370-387 (packed xyz): *n_out is their number, the first min(*n_out, out_capacity) are written -
 *          call with out_capacity 0 to size the buffer.  The caller appends them to its cloud (cloud->push_back).
 *          No plane (fewer than 3 points, every sample collinear): S3D_STATUS_TOO_FEW_POINTS, found = 0. */
typedef struct s3d_plane_fit {
  float coefficients[4];    /* a, b, c, d of a x + b y + c z + d = 0, (a,b,c) normalised */
  int   found;
  int   n_inliers;          /* points within the threshold of the returned plane */
  int   iterations;         /* RANSAC iterations the sequential algorithm ran */
  int   hypotheses_scored;  /* planes actually scored on the device (>= iterations: batches of 32) */
} s3d_plane_fit;
int  s3d_fit_plane(s3d_context* ctx, const float* xyz, int n, int stride, double threshold, int max_iterations,
                   double probability, s3d_plane_fit* out);
int  s3d_fill_ground_plane(s3d_context* ctx, const float* xyz, int n, int stride, double radius, double map_resolution,
                           float* out_xyz, int out_capacity, int* n_out, s3d_plane_fit* fit /* may be NULL */);

/* ==== C1  loop-closure sweep sharded over the GPUs of one node ==========================
 * Replaces the serial candidate loop of ScanSensor::linkToNeighbors (slam3d/core/ScanSensor.cpp:170-202, also
 * run from the detached link thread, :209-210), which calls createConstraint once per candidate on the CPU.
 * One process, one rank (s3d_context + host thread) per device: the pair list is cut into contiguous blocks
 * (s3d_sweep_shard_range), every rank uploads the clouds its block references on first use (they stay resident
 * for later sweeps) and runs one s3d_align_batch; then ONE all-gather of the 128-byte s3d_edge_record's
 * (RCCL ncclAllGather over xGMI, communicators from ncclCommInitAll) leaves every edge in every GPU's HBM in
 * pair order, and `records` is read back from rank 0's gathered buffer.  A pair's record does not depend on the
 * block it lands in: the result equals s3d_align_batch on one context bit for bit.
 *
 * devices: the HIP device of every rank (NULL: devices 0 .. n_devices-1; n_devices 0: every visible device).
 * RCCL admits one rank per device; a list that names a device twice (several contexts on one GPU) gathers
 * with device-to-device copies instead - s3d_sweep_collective() returns "rccl" or "copy".  More than one
 * distinct device without a usable librccl: S3D_STATUS_BACKEND_ERROR. */
typedef struct s3d_sweep       s3d_sweep;
typedef struct s3d_sweep_cloud s3d_sweep_cloud;   /* a host cloud + its lazily created per-rank device copies */
int  s3d_sweep_create(int n_devices, const int* devices, s3d_sweep** out);
/* the same with every rank's stream restricted to the compute units of cu_mask (s3d_context_create_cu_mask) */
int  s3d_sweep_create_cu_mask(int n_devices, const int* devices, const uint32_t* cu_mask, int n_words, s3d_sweep** out);
void s3d_sweep_destroy(s3d_sweep* sw);
int  s3d_sweep_ranks(const s3d_sweep* sw);
const char* s3d_sweep_collective(const s3d_sweep* sw);
const char* s3d_sweep_last_error(const s3d_sweep* sw);
s3d_context* s3d_sweep_context(s3d_sweep* sw, int rank);          /* borrowed; owned by the sweep */
/* the host data is copied: xyz need not outlive the call */
int  s3d_sweep_cloud_create(s3d_sweep* sw, const float* xyz, int n, int stride, s3d_sweep_cloud** out);
void s3d_sweep_cloud_release(s3d_sweep* sw, s3d_sweep_cloud* c);
/* block [lo, hi) of `rank`: ceil(n_pairs / n_ranks) pairs per rank, the last ranks may be short or empty */
void s3d_sweep_shard_range(int n_pairs, int n_ranks, int rank, int* lo, int* hi);
/* same contract as s3d_align_batch; records: n_pairs, pair order */
int  s3d_align_batch_multi(s3d_sweep* sw, int n_pairs, s3d_sweep_cloud* const* sources,
                           s3d_sweep_cloud* const* targets, const double* guesses, const s3d_reg_params* params,
                           const s3d_exec_options* opts, s3d_edge_record* records);
/* the records of the last sweep as rank `rank` holds them in HBM after the all-gather (n_pairs, pair order) */
int  s3d_sweep_gathered_records(s3d_sweep* sw, int rank, int n_pairs, s3d_edge_record* records);

/* ---- candidate generation for a sweep (host only, no GPU): the part of ScanSensor::linkToNeighbors
 *          (ScanSensor.cpp:170-202) that runs BEFORE the registration, on a plain description of the pose graph, so that a
 *          C++ caller can fill the pair list of s3d_align_batch_multi / createConstraints without the Python helper.
 *          Restated: Graph::getNearbyVertices (Graph.cpp:240-261: linear scan over the vertices of the link sensors
 *          in insertion order, double norm of the translation difference < radius), the reverse walk over those
 *          neighbours, "skip the vertex itself", "skip when an edge vertex -> neighbour of this sensor exists"
 *          (BoostGraph::getEdge :156-176 on the out-edges as stored: addEdge stores an edge in both directions,
 *          removeEdge drops one), BoostGraph::calculateGraphDistance (:301-324: Dijkstra over all stored edges, weight 1
 *          for an SE(3) edge and 10000 for any other, float; an unreachable vertex keeps FLT_MAX), "skip when
 *          dist <= 2 * patch_building_range or dist < min_loop_length", stop after max_neighbor_links candidates.
 *          The candidates are the calls link(neighbour, vertex) of :198 in their order: out_sources[k] -> vertex,
 *          under the assumption that every listed registration succeeds (s3d_link_policy.static_graph = 0).
 *          Vertices are named by their insertion index 0 .. n_vertices-1 (boost vecS descriptors). */
typedef struct s3d_graph_edge {
  int source, target;       /* an out-edge as stored (both directions of an edge are two entries) */
  int se3;                  /* != 0: an SE(3) constraint (weight 1 in the graph distance), 0: anything else (10000) */
  int own_sensor;           /* != 0: the edge carries the linking sensor's name (getEdge(vertex, index, mName)) */
} s3d_graph_edge;
typedef struct s3d_link_policy {   /* ScanSensor's knobs, constructor defaults ScanSensor.cpp:34-41 */
  float    neighbor_radius;        /* mNeighborRadius       (1.0) */
  int      max_neighbor_links;     /* mMaxNeighorLinks      (1)   */
  unsigned min_loop_length;        /* mMinLoopLength        (10)  */
  unsigned patch_building_range;   /* mPatchBuildingRange   (0)   */
  int      static_graph;           /* 0: link(neighbour, vertex) of :198 adds its SE(3) edge before the next neighbour is
                                      examined, as in the reference when the registration succeeds (:143, :157-158) - with
                                      max_neighbor_links > 1 the next neighbour of the same cluster is then a hop or two
                                      away and fails min_loop_length: about one link per cluster.  != 0: every neighbour
                                      is judged on the graph as given (more candidates than the reference links; for a
                                      caller that re-filters after each registration).  Same result at the default of 1. */
} s3d_link_policy;
/* positions: 3 doubles per vertex (translation of correctedPose); linkable: per vertex != 0 when its sensor is one of
 * mLinkSensors (NULL: every vertex).  Writes at most `capacity` sources, *n_out = how many there are.
 * Returns S3D_STATUS_OK or S3D_STATUS_INVALID_ARGUMENT (bad indices / NULL arguments). */
int  s3d_link_candidates(int n_vertices, const double* positions, const unsigned char* linkable, int n_edges,
                         const s3d_graph_edge* edges, int vertex, const s3d_link_policy* policy, int* out_sources,
                         int capacity, int* n_out);

#ifdef __cplusplus
}
#endif
#endif
