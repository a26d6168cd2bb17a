// The multi-GPU candidate sweep from C++:   example_link_neighbors <devices|all> scan1.bin scan2.bin ... (KITTI layout)
// Every scan is paired with its two successors (the candidate list of a ScanSensor::linkToNeighbors call,
// ScanSensor.cpp:179-201) and the candidates are registered twice: one blocking createConstraint after the other, as
// the reference does, and as ONE sweep over the given devices ("0,0" = two ranks on GPU 0, "all" = every GPU) through
// PointCloudSensor::createConstraints -> s3d_align_batch_multi (RCCL all-gather of the edges).  Prints both edge lists;
// they are identical bit for bit.
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
  } catch (const std::exception& e) {
    std::printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
