/* slam3d_registration_types.h — plain-C value types shared by the C-ABI
 * (include/slam3d_hip.h), the CPU oracle (oracle/) and the C++ host mirror
 * (cpp/slam3d/...).  No functions, no device types.
 *
 * Every type cites the reference declaration it mirrors
 * (paths relative to the dfki-ric/slam3d checkout).
 */
#ifndef SLAM3D_REGISTRATION_TYPES_H
#define SLAM3D_REGISTRATION_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* slam3d/sensor/pcl/RegistrationParameters.hpp:30
 *   enum RegistrationAlgorithm {ICP, GICP, GICP_OMP, NDT, NDT_OMP};
 * Same enumerator order, so the integer values are interchangeable. */
enum s3d_registration_algorithm {
  S3D_ALG_ICP      = 0, /* reference: enumerator exists, no `case` -> runtime_error
                           (PointCloudSensor.cpp:163-164).  Here: point-to-plane
                           Gauss-Newton ICP (the north-star's 6x6 reduction). */
  S3D_ALG_GICP     = 1, /* reference default: pcl::GeneralizedIterativeClosestPoint */
  S3D_ALG_GICP_OMP = 2, /* reference: pclomp variant, same arithmetic as GICP */
  S3D_ALG_NDT      = 3, /* doNDT (:84-117): voxel statistics + derivative passes on the device, Newton /
                           More-Thuente state machines on the host, batches advance in lock-step rounds */
  S3D_ALG_NDT_OMP  = 4  /* reference: pclomp::NormalDistributionsTransform - the NDT optimiser over pclomp's default
                           neighbour search DIRECT7 (the voxel holding the transformed point + its six face neighbours,
                           VoxelGridCovariance::getNeighborhoodAtPoint7) instead of PCL's kd-tree radius query */
};

/* slam3d/sensor/pcl/RegistrationParameters.hpp:36-97 — field-for-field, same
 * order, same types (enum == int), so a `slam3d::RegistrationParameters` can be
 * passed as `const s3d_reg_params*` (static_assert'ed in the C++ mirror). */
typedef struct s3d_reg_params {
  int    registration_algorithm;        /* :42  = GICP  */
  double point_cloud_density;           /* :45  = 0.2   */
  double max_fitness_score;             /* :49  = 2.0   */
  double max_translation;               /* :52  = 1.0   */
  double max_rotation;                  /* :55  = 1.0   */
  double euclidean_fitness_epsilon;     /* :61  = 1.0   (forwarded, unused by GICP) */
  double transformation_epsilon;        /* :64  = 1e-5  */
  double max_correspondence_distance;   /* :68  = 2.5   */
  int    maximum_iterations;            /* :71  = 50    */
  double rotation_epsilon;              /* :78  = 2e-3  */
  int    correspondence_randomness;     /* :81  = 20    */
  int    maximum_optimizer_iterations;  /* :84  = 20    */
  float  resolution;                    /* :90  = 1.0   (NDT) */
  double step_size;                     /* :93  = 0.05  (NDT) */
  double outlier_ratio;                 /* :96  = 0.35  (NDT) */
} s3d_reg_params;

/* Status of one align()-equivalent.  The reference signals these with C++
 * exceptions (PointCloudSensor.cpp:76, :135, :161, :164, :171); the C++ mirror
 * re-raises the same exception types with the same messages. */
enum s3d_status {
  S3D_STATUS_OK                    = 0,
  S3D_STATUS_TOO_FEW_POINTS        = 1, /* :134-135 NoMatch("Too few points after filtering, ...") */
  S3D_STATUS_NOT_CONVERGED         = 2, /* :74 !hasConverged()  -> NoMatch("ICP failed with Fitness-Score ...") */
  S3D_STATUS_FITNESS_EXCEEDED      = 3, /* :74 score > max_fitness_score -> same NoMatch */
  S3D_STATUS_TOO_FAR_FROM_GUESS    = 4, /* :169-172 NoMatch("ICP result is to far away from guess") */
  S3D_STATUS_UNKNOWN_ALGORITHM     = 5, /* :163-164 std::runtime_error("Unknown registration algorithm specified.") */
  S3D_STATUS_UNSUPPORTED_ALGORITHM = 6, /* reserved (NDT / NDT_OMP returned it before they were implemented) */
  S3D_STATUS_INVALID_ARGUMENT      = 7,
  S3D_STATUS_BACKEND_ERROR         = 8, /* HIP runtime error; message via s3d_last_error() */
  S3D_STATUS_OMP_UNAVAILABLE       = 9  /* :159-161 std::runtime_error("OMP is not available, ...") - a reference built
                                           without pclomp; only with s3d_exec_options.omp_unavailable (slam3d_hip.h) */
};

/* One registration result ("edge record").  16 doubles = 128 bytes: the unit
 * that is all-gathered over RCCL in the multi-GPU sweep.
 * transform[] is the 3x4 top of the 4x4 result, COLUMN-major (Eigen order):
 * transform[0..2] = first column of R, ..., transform[9..11] = translation. */
typedef struct s3d_edge_record {
  double transform[12];
  double fitness;          /* pcl getFitnessScore(max_correspondence_distance) equivalent */
  double iterations;       /* outer ICP iterations executed (fine stage)      */
  double correspondences;  /* #pairs under the distance gate in the last iteration */
  double status;           /* enum s3d_status                                  */
} s3d_edge_record;

#ifdef __cplusplus
}
#endif
#endif
