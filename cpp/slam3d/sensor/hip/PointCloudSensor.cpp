// Host side of slam3d::PointCloudSensor on the MI355X back-end — the counterpart of the
// reference's slam3d/sensor/pcl/PointCloudSensor.cpp.  Everything numerical happens behind
// include/slam3d_hip.h; status codes are re-raised as the reference's exceptions with the
// reference's messages.
#include "PointCloudSensor.hpp"

#include <cstdio>
#include <iostream>

namespace slam3d {

void Logger::message(LogLevel lvl, const std::string& msg) {
  if (lvl < mLogLevel) return;
  static const char* names[] = {"[DEBUG] ", "[INFO] ", "[WARN] ", "[ERROR] ", "[FATAL] "};
  std::cerr << names[lvl] << msg << std::endl;
}

namespace {
std::string fmt(const char* f, double a, double b) {
  char buf[256];
  std::snprintf(buf, sizeof buf, f, a, b);
  return buf;
}
s3d_context* shared_context_for_static_calls() {
  static s3d_context* ctx = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    if (s3d_context_create(0, nullptr, &ctx) != S3D_STATUS_OK) ctx = nullptr;
  });
  if (!ctx) throw std::runtime_error("slam3d (MI355X build): no usable HIP device, and there is no CPU fallback");
  return ctx;
}
}  // namespace

PointCloudSensor::PointCloudSensor(const std::string& n, Logger* l, int device) : ScanSensor(n, l), mContext(nullptr) {
  mScanResolution = 0.1;      // PointCloudSensor.cpp:179-182
  mMapResolution = 0.1;
  mMapOutlierRadius = 0.2;
  mMapOutlierNeighbors = 3;
  if (s3d_context_create(device, nullptr, &mContext) != S3D_STATUS_OK)
    throw std::runtime_error("slam3d (MI355X build): no usable HIP device, and there is no CPU fallback");
}

PointCloudSensor::~PointCloudSensor() { s3d_context_destroy(mContext); }

PointCloud::Ptr PointCloudSensor::downsample(PointCloud::Ptr in, double leaf_size) {
  PointCloud::Ptr out(new PointCloud);
  if (in->size() > 0) {  // PointCloudSensor.cpp:193
    std::vector<float> packed(in->size() * 3);
    int n_out = 0;
    const int st = s3d_voxel_downsample(shared_context_for_static_calls(), &in->points[0].x, (int)in->size(), 4,
                                        leaf_size, packed.data(), &n_out);
    if (st != S3D_STATUS_OK) throw std::runtime_error("s3d_voxel_downsample failed");
    out->points.resize(n_out);
    for (int i = 0; i < n_out; ++i) out->points[i] = PointType{packed[3 * i], packed[3 * i + 1], packed[3 * i + 2], 1.f};
    out->width = (uint32_t)n_out;
    out->height = 1;
    out->is_dense = true;
  }
  return out;
}

PointCloud::Ptr PointCloudSensor::downsampleScan(PointCloud::Ptr source) {
  return mScanResolution > 0 ? downsample(source, mScanResolution) : source;  // :203-209
}

PointCloud::Ptr PointCloudSensor::transform(PointCloud::ConstPtr source, const Transform tf) const {
  PointCloud::Ptr out(new PointCloud(*source));  // pcl::transformPointCloud with a double matrix (:228-233)
  for (PointType& p : out->points) {
    const double x = p.x, y = p.y, z = p.z;
    p.x = (float)(tf(0, 0) * x + tf(0, 1) * y + tf(0, 2) * z + tf(0, 3));
    p.y = (float)(tf(1, 0) * x + tf(1, 1) * y + tf(1, 2) * z + tf(1, 3));
    p.z = (float)(tf(2, 0) * x + tf(2, 1) * y + tf(2, 2) * z + tf(2, 3));
  }
  return out;
}

Transform PointCloudSensor::align(const PointCloudMeasurement::Ptr& source, const PointCloudMeasurement::Ptr& target,
                                  const Transform& guess, const RegistrationParameters& config) {
  const PointCloud::Ptr s = source->getPointCloud(), t = target->getPointCloud();
  Transform result;
  s3d_align_info info;
  static const float dummy[4] = {0, 0, 0, 0};
  const int st = s3d_align(mContext, s->size() ? &s->points[0].x : dummy, (int)s->size(), 4,
                           t->size() ? &t->points[0].x : dummy, (int)t->size(), 4, guess.data(),
                           reinterpret_cast<const s3d_reg_params*>(&config), nullptr, result.data(), &info);
  switch (st) {
    case S3D_STATUS_OK: return result;
    case S3D_STATUS_TOO_FEW_POINTS:   // PointCloudSensor.cpp:135
      throw NoMatch("Too few points after filtering, you may have to decrease 'point_cloud_density'.");
    case S3D_STATUS_NOT_CONVERGED:
    case S3D_STATUS_FITNESS_EXCEEDED:  // :76
      throw NoMatch(fmt("ICP failed with Fitness-Score %g > %g", info.fitness, config.max_fitness_score));
    case S3D_STATUS_TOO_FAR_FROM_GUESS:  // :171
      throw NoMatch("ICP result is to far away from guess");
    case S3D_STATUS_UNSUPPORTED_ALGORITHM:  // NDT: not on the accelerated path (cf. :161)
      throw std::runtime_error("NDT is not available in the MI355X build, use GICP or ICP.");
    case S3D_STATUS_UNKNOWN_ALGORITHM:  // :164
      throw std::runtime_error("Unknown registration algorithm specified.");
    default:
      throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  }
}

Constraint::Ptr PointCloudSensor::createConstraint(const Measurement::Ptr& source, const Measurement::Ptr& target,
                                                   const Transform& odometry, bool loop) {
  // PointCloudSensor.cpp:274
  Transform guess = source->getInverseSensorPose() * odometry * target->getSensorPose();
  PointCloudMeasurement::Ptr sourceCloud = std::dynamic_pointer_cast<PointCloudMeasurement>(source);
  PointCloudMeasurement::Ptr targetCloud = std::dynamic_pointer_cast<PointCloudMeasurement>(target);
  if (!sourceCloud || !targetCloud) {  // :279-283
    mLogger->message(ERROR, "Measurement given to createConstraint() is not a PointCloud!");
    throw BadMeasurementType();
  }
  if (loop) guess = align(sourceCloud, targetCloud, guess, mCoarseConfiguration);  // :286-289
  Transform icp_result = align(sourceCloud, targetCloud, guess, mFineConfiguration);  // :292
  Transform transform = source->getSensorPose() * icp_result * target->getInverseSensorPose();  // :295
  Covariance<6> information = Covariance<6>::Identity();
  for (unsigned i = 0; i < 6; ++i) information(i, i) = 1.0 / mCovarianceScale;  // (I * scale)^-1, :296-298
  return Constraint::Ptr(new SE3Constraint(mName, transform, information));
}

void PointCloudSensor::setRegistrationParameters(const RegistrationParameters& conf, bool coarse) {
  if (coarse) { mLogger->message(INFO, " = RegistrationParameters (Coarse) ="); mCoarseConfiguration = conf; }
  else { mLogger->message(INFO, " = RegistrationParameters (Fine) ="); mFineConfiguration = conf; }
  std::ostringstream os;
  os << "correspondence_randomness:    " << conf.correspondence_randomness << "\nmax_correspondence_distance:  "
     << conf.max_correspondence_distance << "\nmaximum_iterations:           " << conf.maximum_iterations
     << "\npoint_cloud_density:          " << conf.point_cloud_density;
  mLogger->message(INFO, os.str());
}
void PointCloudSensor::setScanResolution(double r) { mScanResolution = r; }
void PointCloudSensor::setMapResolution(double r) { mMapResolution = r; }
void PointCloudSensor::setMapOutlierRemoval(double r, unsigned n) { mMapOutlierRadius = r; mMapOutlierNeighbors = n; }

}  // namespace slam3d
