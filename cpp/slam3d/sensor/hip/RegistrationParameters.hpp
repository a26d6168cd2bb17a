// Field-for-field mirror of slam3d/sensor/pcl/RegistrationParameters.hpp:30-97 (same enum, same
// member names, order, types and in-class defaults), so application code that assigns members keeps
// compiling; layout-compatible with the C ABI's s3d_reg_params (static_assert in PointCloudSensor.hpp).
#pragma once

namespace slam3d {

enum RegistrationAlgorithm { ICP, GICP, GICP_OMP, NDT, NDT_OMP };

struct RegistrationParameters {
  RegistrationAlgorithm registration_algorithm = GICP;
  double point_cloud_density = 0.2;
  double max_fitness_score = 2.0;
  double max_translation = 1.0;
  double max_rotation = 1.0;
  double euclidean_fitness_epsilon = 1.0;
  double transformation_epsilon = 1e-5;
  double max_correspondence_distance = 2.5;
  int maximum_iterations = 50;
  double rotation_epsilon = 2e-3;
  int correspondence_randomness = 20;
  int maximum_optimizer_iterations = 20;
  float resolution = 1.0;
  double step_size = 0.05;
  double outlier_ratio = 0.35;
};

}  // namespace slam3d
