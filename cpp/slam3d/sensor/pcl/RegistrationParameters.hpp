// slam3d::RegistrationParameters for the MI355X build.  The reference declares the fifteen tuning members one by
// one (slam3d/sensor/pcl/RegistrationParameters.hpp:30-97); here the type IS the C ABI's s3d_reg_params
// (include/slam3d_registration_types.h — same member names, order and types, which is what application code that
// assigns `config.maximum_iterations = ...` relies on) with the reference's defaults filled in by the library.
#pragma once

#include "../../../../include/slam3d_hip.h"

namespace slam3d {

// same enumerators and values as the reference; the C struct stores the value as an int
enum RegistrationAlgorithm { ICP = S3D_ALG_ICP, GICP = S3D_ALG_GICP, GICP_OMP = S3D_ALG_GICP_OMP, NDT = S3D_ALG_NDT,
                             NDT_OMP = S3D_ALG_NDT_OMP };

struct RegistrationParameters : s3d_reg_params {
  RegistrationParameters() { s3d_default_params(this); }   // RegistrationParameters.hpp:36-97 defaults
};

}  // namespace slam3d
