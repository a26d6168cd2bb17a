/* s3d_oracle.h — CPU restatement of slam3d's point-cloud registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under slam3d_amd/ or cpp/ may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in PCL (>= 1.8.1, nominal
 * 1.12.1) + FLANN 1.9.1 + Eigen 3.4, none of which is vendored under
 * /root/reference or installed here, and no reference test holds an expected
 * value for this path.  The slam3d-side logic follows the
 * in-tree sources line by line; the PCL-side logic restates PCL's published
 * algorithm (gicp.hpp, bfgs.h, voxel_grid.hpp, registration.hpp) and is
 * anchored on the reference's call sites.
 */
#ifndef S3D_ORACLE_H
#define S3D_ORACLE_H

#include "../include/slam3d_registration_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A3: PointCloudSensor::downsample -> pcl::VoxelGrid (PointCloudSensor.cpp:190-201) */
typedef struct s3o_voxel_info {
  int   min_b[3], max_b[3], div_b[3];
  int   passthrough;      /* 1: dx*dy*dz > INT_MAX -> PCL warns and copies the input */
  float min_p[3], max_p[3];
} s3o_voxel_info;

/* xyz: n points, `stride` floats apart (3 for packed xyz, 4 for PCL/KITTI layout).
 * out: capacity 3*n floats (packed xyz).  Returns number of output points. */
int s3o_voxel_downsample(const float* xyz, int n, int stride, double leaf_size,
                         float* out, s3o_voxel_info* info);

/* ---- A7: exact 1-NN / k-NN (FLANN KDTreeSingleIndex restated: exact, L2, leaf 15).
 * Ties are broken by (d2, index) lexicographic order. */
typedef struct s3o_kdtree s3o_kdtree;
s3o_kdtree* s3o_kdtree_build(const float* xyz /*packed*/, int n);
void        s3o_kdtree_free(s3o_kdtree*);
void        s3o_kdtree_nn1(const s3o_kdtree*, const float q[3], int* idx, float* d2);
/* writes k (idx,d2) sorted ascending; returns number found (= min(k,n)) */
int         s3o_kdtree_knn(const s3o_kdtree*, const float q[3], int k, int* idx, float* d2);
/* batch helper used by tests */
void s3o_nn_search(const float* tgt, int n, const float* qry, int m, int* idx, float* d2);
void s3o_nn_search_brute(const float* tgt, int n, const float* qry, int m, int* idx, float* d2);

/* ---- A6: GICP::computeCovariances.  cov: n * 9 doubles (row-major 3x3).
 * normals (optional, may be NULL): n*3 doubles, unit eigenvector of the smallest eigenvalue.
 * returns 0 ok, -1 if k > n (PCL: error, covariances unusable). */
int s3o_gicp_covariances(const float* xyz, int n, int k, double gicp_epsilon,
                         double* cov, double* normals);

/* ---- A4/A5/A8/A9: doICP<GICP>: align + getFitnessScore.
 * pcl_source = slam3d TARGET cloud (queries), pcl_target = slam3d SOURCE cloud (kd-tree),
 * exactly the swap at PointCloudSensor.cpp:68-69.  guess/final: 4x4 float, column-major. */
typedef struct s3o_icp_result {
  float  final_transformation[16]; /* column-major */
  int    converged;                /* pcl hasConverged() */
  int    iterations;               /* nr_iterations_ */
  int    correspondences;          /* in the last outer iteration */
  double fitness;                  /* getFitnessScore(max_correspondence_distance) */
  int    inner_iterations_total;   /* sum of BFGS steps (diagnostic) */
  int    evaluations_total;        /* functor passes over the correspondences (diagnostic) */
} s3o_icp_result;

/* force_iterations != 0: ignore the `delta < 1` exit and run exactly
 * maximum_iterations outer iterations (bench mode). */
int s3o_gicp(const float* pcl_source, int m, const float* pcl_target, int n,
             const float guess[16], const s3d_reg_params* cfg, int force_iterations,
             s3o_icp_result* out);

/* point-to-plane Gauss-Newton ICP for the (reference-unhandled) `ICP` enumerator. */
int s3o_icp_point_to_plane(const float* pcl_source, int m, const float* pcl_target, int n,
                           const float guess[16], const s3d_reg_params* cfg, int force_iterations,
                           s3o_icp_result* out);

/* doNDT (PCS.cpp:84-117) -> pcl::NormalDistributionsTransform: Newton steps with More-Thuente line search on
 * the NDT score of the voxelised target (resolution, step_size, outlier_ratio of cfg).  `correspondences`
 * returns the number of NDT cells, `evaluations_total` the derivative passes. */
int s3o_ndt(const float* pcl_source, int m, const float* pcl_target, int n, const float guess[16],
            const s3d_reg_params* cfg, s3o_icp_result* out);

/* GICP objective (mean Mahalanobis residual) of a candidate final transformation F */
double s3o_gicp_cost(const float* pcl_source, int m, const float* pcl_target, int n, const float F[16],
                     const s3d_reg_params* cfg, int* n_corr);

double s3o_fitness_score(const float* pcl_source, int m, const float* pcl_target, int n,
                         const float final_transformation[16], double max_range);

/* ---- A2: align() (PointCloudSensor.cpp:119-174).  Clouds given with stride.
 * guess/result: 4x4 double column-major (slam3d::Transform).  Returns s3d_status. */
typedef struct s3o_align_info {
  int    n_source_filtered, n_target_filtered;
  int    iterations, converged, correspondences;
  double fitness;
} s3o_align_info;
int s3o_align(const float* source, int n_source, int stride_source,
              const float* target, int n_target, int stride_target,
              const double guess[16], const s3d_reg_params* cfg, int force_iterations,
              double result[16], s3o_align_info* info);

/* ---- A1: createConstraint (PointCloudSensor.cpp:269-299).
 * sensor poses / odometry: 4x4 double column-major.  information: 6x6 row-major. */
int s3o_create_constraint(const float* source, int n_source, int stride_source, const double source_sensor_pose[16],
                          const float* target, int n_target, int stride_target, const double target_sensor_pose[16],
                          const double odometry[16], int loop,
                          const s3d_reg_params* fine, const s3d_reg_params* coarse,
                          double covariance_scale,
                          double relative_pose[16], double information[36], s3o_align_info* info);

/* ---- B1: PointCloudSensor::transform (PointCloudSensor.cpp:228-233),
 * getAccumulatedCloud (:235-256), createCombinedMeasurement (:258-266).
 * tf / poses / frame: 4x4 double column-major.  out: packed xyz. */
void s3o_transform_cloud(const float* xyz, int n, int stride, const double tf[16], float* out);
int  s3o_accumulate_clouds(const float* const* clouds, const int* sizes, const int* strides, int n_clouds,
                           const double* poses, const double* frame /* NULL: no re-framing */, float* out);
/* ---- B2 (rank 2): removeOutliers (:211-226) -> pcl::RadiusOutlierRemoval; B3: buildMap (:301-318) */
int  s3o_remove_outliers(const float* xyz, int n, int stride, double radius, unsigned min_neighbors, float* out);
int  s3o_build_map(const float* const* clouds, const int* sizes, const int* strides, int n_clouds, const double* poses,
                   double outlier_radius, unsigned outlier_neighbors, double map_resolution, float* out);

/* ---- fillGroundPlane (:362-388): pcl::RandomSampleConsensus<SampleConsensusModelPlane> (threshold 0.01 there,
 * PCL defaults max_iterations 1000, probability 0.99) and the ring points appended afterwards */
int  s3o_fit_plane_ransac(const float* xyz, int n, int stride, double threshold, int max_iterations, double probability,
                          float coeffs[4], int* n_inliers, int* iterations);
int  s3o_fill_ground_points(const float coeffs[4], double radius, double map_resolution, float* out, int cap);
void s3o_mt19937_outputs(unsigned seed, int n, unsigned* out);   /* test hook for the sampler's generator */

/* ---- variants / diagnostics (process-global; defaults = PCL-literal behaviour)
 * eval_precision 0: the BFGS functor transforms points with a float 4x4 in float, as PCL does
 *                   (f(x) is then piecewise constant at the 1e-7 level: "float staircase");
 *                1: same float matrix, arithmetic carried in double;
 *                2: the matrix [R(x)|t(x)] itself kept in double (smooth f) — this is the
 *                   function the device path minimises (quadratic form).
 * debug_perturbation: multiplies every Mahalanobis entry by (1 + rel*U(-.5,.5)); used by
 *                   tests/test_conditioning.py to measure how well-defined the reference result is. */
void s3o_set_eval_precision(int mode);
void s3o_set_omp_available(int on);   /* 0: a reference built without pclomp (GICP_OMP / NDT_OMP throw, PCS.cpp:159-161) */
void s3o_set_debug_float_normals(int on);   /* diagnostic: covariances from float-rounded normals, as the device stores them */
void s3o_set_debug_perturbation(double rel);
void s3o_set_debug_perturbation_seed(unsigned long long seed);   /* the noise is a pure function of (seed, iteration, correspondence, entry) */
void s3o_set_trace(int on);

/* small helpers exposed for tests */
void s3o_default_params(s3d_reg_params* p);
void s3o_sym_eig3(const double a[9], double evals_desc[3], double evecs_cols[9]);
void s3o_mat4d_inverse_isometry(const double a[16], double out[16]);
void s3o_mat4d_mul(const double a[16], const double b[16], double out[16]);
double s3o_rotation_angle(const double m[16]);

#ifdef __cplusplus
}
#endif
#endif
