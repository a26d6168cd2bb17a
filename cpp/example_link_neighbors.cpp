// The multi-GPU candidate sweep from C++:   example_link_neighbors <devices|all> scan1.bin scan2.bin ... (KITTI layout)
// Every scan is paired with its two successors (the candidate list of a ScanSensor::linkToNeighbors call,
// ScanSensor.cpp:179-201) and the candidates are registered twice: one blocking createConstraint after the other, as
// the reference does, and as ONE sweep over the given devices ("0,0" = two ranks on GPU 0, "all" = every GPU) through
// PointCloudSensor::createConstraints -> s3d_align_batch_multi (RCCL all-gather of the edges).  Prints both edge lists;
// they are identical bit for bit.  Then the same with loop = true (coarse registration first, ScanSensor::link's call,
// ScanSensor.cpp:156), the number of sweep clouds kept before / after a sweep over short-lived patch measurements, and
// the exception a reference built without pclomp throws for GICP_OMP.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "slam3d/sensor/pcl/PointCloudSensor.hpp"

using namespace slam3d;

static PointCloud::Ptr load_bin(const char* path) {
  PointCloud::Ptr c(new PointCloud);
  std::ifstream f(path, std::ios::binary);
  float v[4];
  while (f.read(reinterpret_cast<char*>(v), sizeof v)) c->push_back(makePoint(v[0], v[1], v[2]));
  return c;
}

static void print_edge(const char* tag, size_t i, const Constraint::Ptr& c) {
  if (!c) { std::printf("%s %zu NoMatch\n", tag, i); return; }
  SE3Constraint::Ptr se3 = ptr::dynamic_pointer_cast<SE3Constraint>(c);
  std::printf("%s %zu", tag, i);
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) std::printf(" %a", se3->getRelativePose()(r, col));
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <devices|all> scan1.bin scan2.bin [...]\n", argv[0]); return 2; }
  std::vector<int> devices;
  if (std::strcmp(argv[1], "all") != 0)
    for (char* tok = std::strtok(argv[1], ","); tok; tok = std::strtok(nullptr, ",")) devices.push_back(std::atoi(tok));
  Logger logger;
  logger.setLogLevel(ERROR);
  try {
    PointCloudSensor sensor("velodyne", &logger);
    std::vector<Measurement::Ptr> scans;
    for (int i = 2; i < argc; ++i)
      scans.push_back(Measurement::Ptr(new PointCloudMeasurement(load_bin(argv[i]), "robot", sensor.getName(), Transform::Identity())));
    std::vector<Measurement::Ptr> src, tgt;
    std::vector<Transform> odo;
    for (size_t i = 0; i < scans.size(); ++i)
      for (size_t j = i + 1; j < scans.size() && j <= i + 2; ++j) { src.push_back(scans[i]); tgt.push_back(scans[j]); odo.push_back(Transform::Identity()); }
    for (size_t i = 0; i < src.size(); ++i) {   // the reference's way: one candidate at a time
      Constraint::Ptr c;
      try { c = sensor.createConstraint(src[i], tgt[i], odo[i], false); } catch (const NoMatch&) {}
      print_edge("sequential", i, c);
    }
    const std::vector<Constraint::Ptr> sweep = sensor.createConstraints(src, tgt, odo, devices);
    for (size_t i = 0; i < sweep.size(); ++i) print_edge("sweep", i, sweep[i]);
    // loop closures: coarse then fine (PointCloudSensor.cpp:286-292)
    RegistrationParameters coarse;
    coarse.point_cloud_density = 0.5;
    coarse.max_correspondence_distance = 5.0;
    coarse.maximum_iterations = 10;
    coarse.max_translation = 3.0;
    sensor.setRegistrationParameters(coarse, true);
    for (size_t i = 0; i < src.size(); ++i) {
      Constraint::Ptr c;
      try { c = sensor.createConstraint(src[i], tgt[i], odo[i], true); } catch (const NoMatch&) {}
      print_edge("sequential_loop", i, c);
    }
    const std::vector<Constraint::Ptr> sweep_loop = sensor.createConstraints(src, tgt, odo, devices, true);
    for (size_t i = 0; i < sweep_loop.size(); ++i) print_edge("sweep_loop", i, sweep_loop[i]);
    // patches (ScanSensor::buildPatch) are new measurements with fresh uuids for every candidate: their sweep clouds
    // must go when they do
    const size_t kept_before = sensor.getSweepCloudCount();
    {
      std::vector<Measurement::Ptr> ps, pt;
      std::vector<Transform> po;
      for (size_t i = 0; i + 1 < scans.size(); ++i) {
        PointCloudMeasurement::Ptr a = ptr::dynamic_pointer_cast<PointCloudMeasurement>(scans[i]);
        PointCloudMeasurement::Ptr b = ptr::dynamic_pointer_cast<PointCloudMeasurement>(scans[i + 1]);
        ps.push_back(Measurement::Ptr(new PointCloudMeasurement(a->getPointCloud(), "robot", sensor.getName(), Transform::Identity())));
        pt.push_back(Measurement::Ptr(new PointCloudMeasurement(b->getPointCloud(), "robot", sensor.getName(), Transform::Identity())));
        po.push_back(Transform::Identity());
      }
      sensor.createConstraints(ps, pt, po, devices);
      std::printf("sweep_clouds_with_patches %zu\n", sensor.getSweepCloudCount());
    }
    sensor.createConstraints(std::vector<Measurement::Ptr>(1, src[0]), std::vector<Measurement::Ptr>(1, tgt[0]),
                             std::vector<Transform>(1, odo[0]), devices);
    std::printf("sweep_clouds %zu %zu\n", kept_before, sensor.getSweepCloudCount());
    // GICP_OMP in a reference built without pclomp (PointCloudSensor.cpp:159-161)
    RegistrationParameters omp;
    omp.registration_algorithm = GICP_OMP;
    sensor.setRegistrationParameters(omp, false);
    sensor.setOmpAvailable(false);
    try { sensor.createConstraint(src[0], tgt[0], odo[0], false); std::printf("omp served\n"); }
    catch (const NoMatch&) { std::printf("omp nomatch\n"); }
    catch (const std::runtime_error& e) { std::printf("omp %s\n", e.what()); }
  } catch (const std::exception& e) {
    std::printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
