// Drives the C++ mirror of the reference plugin API on two KITTI-layout .bin scans
// (float32 x,y,z,intensity):   example_create_constraint cloud1.bin cloud2.bin [loop] [ICP|GICP|NDT]
// Prints the SE(3) edge (row-major 4x4) or the exception the reference would have thrown.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "slam3d/sensor/pcl/PointCloudSensor.hpp"

using namespace slam3d;

static PointCloud::Ptr load_bin(const char* path) {
  PointCloud::Ptr c(new PointCloud);
  std::ifstream f(path, std::ios::binary);
  float v[4];
  while (f.read(reinterpret_cast<char*>(v), sizeof v)) c->push_back(makePoint(v[0], v[1], v[2]));
  return c;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s source.bin target.bin [loop] [ICP|GICP|NDT]\n", argv[0]); return 2; }
  Logger logger;
  logger.setLogLevel(WARNING);
  try {
    PointCloudSensor sensor("velodyne", &logger);
    RegistrationParameters fine, coarse;
    coarse.point_cloud_density = 0.5;
    coarse.max_correspondence_distance = 5.0;
    coarse.max_translation = 3.0;
    bool loop = false;
    for (int i = 3; i < argc; ++i) {
      if (!std::strcmp(argv[i], "loop")) loop = true;
      if (!std::strcmp(argv[i], "ICP")) fine.registration_algorithm = coarse.registration_algorithm = ICP;
      if (!std::strcmp(argv[i], "NDT")) fine.registration_algorithm = NDT;
    }
    sensor.setRegistrationParameters(fine, false);
    sensor.setRegistrationParameters(coarse, true);
    sensor.setCovarianceScale(4.0);
    Measurement::Ptr m1(new PointCloudMeasurement(load_bin(argv[1]), "robot", sensor.getName(), Transform::Identity()));
    Measurement::Ptr m2(new PointCloudMeasurement(load_bin(argv[2]), "robot", sensor.getName(), Transform::Identity()));
    Constraint::Ptr c = sensor.createConstraint(m1, m2, Transform::Identity(), loop);
    SE3Constraint::Ptr se3 = ptr::dynamic_pointer_cast<SE3Constraint>(c);
    std::printf("OK %s\n", se3->getTypeName());
    for (int r = 0; r < 4; ++r)
      std::printf("%.12g %.12g %.12g %.12g\n", se3->getRelativePose()(r, 0), se3->getRelativePose()(r, 1),
                  se3->getRelativePose()(r, 2), se3->getRelativePose()(r, 3));
    std::printf("information00 %.6g\n", se3->getInformation()(0, 0));
  } catch (const NoMatch& e) {
    std::printf("NoMatch %s\n", e.what());
  } catch (const BadMeasurementType& e) {
    std::printf("BadMeasurementType %s\n", e.what());
  } catch (const std::exception& e) {
    std::printf("runtime_error %s\n", e.what());
  }
  return 0;
}
