// Drives the patch / map half of the C++ mirror the way ScanSensor::link and the map publisher of the
// reference's applications do:   example_build_map poses.txt scan0.bin scan1.bin ...
// poses.txt: one corrected vertex pose per scan, 16 doubles row-major per line.  Scans: KITTI-layout .bin.
// Prints the map size + an order-independent checksum, then (>= 4 scans) registers the patch around
// vertex 0 (scans 0,1) against the patch around vertex 2 (scans 2,3) as a loop closure.
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
  if (argc < 3) { std::fprintf(stderr, "usage: %s poses.txt scan0.bin [scan1.bin ...]\n", argv[0]); return 2; }
  Logger logger;
  logger.setLogLevel(WARNING);
  try {
    PointCloudSensor sensor("velodyne", &logger);
    MeasurementStorage storage;
    sensor.setMeasurementStorage(&storage);
    std::ifstream pf(argv[1]);
    VertexObjectList vertices;
    for (int i = 2; i < argc; ++i) {
      Measurement::Ptr m(new PointCloudMeasurement(load_bin(argv[i]), "robot", sensor.getName(), Transform::Identity()));
      storage.add(m);
      VertexObject v;
      v.init(m, (IdType)(i - 2));
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) pf >> v.correctedPose(r, c);
      vertices.push_back(v);
    }
    PointCloud::Ptr map = sensor.buildMap(vertices);
    double sum[3] = {0, 0, 0};
    for (const PointType& p : map->points) { sum[0] += p.x; sum[1] += p.y; sum[2] += p.z; }
    std::printf("MAP %zu %.9g %.9g %.9g\n", map->size(), sum[0], sum[1], sum[2]);
    PointCloud::Ptr accu = sensor.getAccumulatedCloud(vertices);
    std::printf("ACCU %zu\n", accu->size());
    if (vertices.size() >= 4) {
      VertexObjectList a(vertices.begin(), vertices.begin() + 2), b(vertices.begin() + 2, vertices.begin() + 4);
      Measurement::Ptr pa = sensor.createCombinedMeasurement(a, vertices[0].correctedPose);   // ScanSensor.cpp:269
      Measurement::Ptr pb = sensor.createCombinedMeasurement(b, vertices[2].correctedPose);
      RegistrationParameters fine, coarse;
      fine.registration_algorithm = coarse.registration_algorithm = ICP;
      coarse.point_cloud_density = 0.5;
      coarse.max_correspondence_distance = 5.0;
      sensor.setRegistrationParameters(fine, false);
      sensor.setRegistrationParameters(coarse, true);
      Transform guess = vertices[0].correctedPose.inverse() * vertices[2].correctedPose;      // Graph::getTransform
      std::printf("PATCH %zu %zu\n", ptr::dynamic_pointer_cast<PointCloudMeasurement>(pa)->getPointCloud()->size(),
                  ptr::dynamic_pointer_cast<PointCloudMeasurement>(pb)->getPointCloud()->size());
      try {
        Constraint::Ptr c = sensor.createConstraint(pa, pb, guess, true);
        SE3Constraint::Ptr se3 = ptr::dynamic_pointer_cast<SE3Constraint>(c);
        std::printf("OK %s\n", se3->getTypeName());
        for (int r = 0; r < 4; ++r)
          std::printf("%.12g %.12g %.12g %.12g\n", se3->getRelativePose()(r, 0), se3->getRelativePose()(r, 1),
                      se3->getRelativePose()(r, 2), se3->getRelativePose()(r, 3));
      } catch (const NoMatch& e) {
        std::printf("NoMatch %s\n", e.what());
      }
    }
  } catch (const BadMeasurementType& e) {
    std::printf("BadMeasurementType %s\n", e.what());
  } catch (const std::exception& e) {
    std::printf("runtime_error %s\n", e.what());
  }
  return 0;
}
