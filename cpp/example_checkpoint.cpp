// The device cache across a checkpoint:   example_checkpoint <folder> scan1.bin scan2.bin   (KITTI layout)
// Registers scan2 against scan1, writes what the library cached for the two measurements to <folder>/<i>.s3dc - the
// files a GraphSerialization::toFolder would put next to its <index>.s3dm archives (GraphSerialization.cpp:40-47) -
// then plays fromFolder (:68-135): NEW measurement objects with the stored uuids, the cache files handed back, the
// same registration again.  Prints both edges (identical bit for bit) and the cache counters around the second one
// (no misses: nothing was recomputed).  A cache file offered to the wrong scan is refused.
#include <cstdio>
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

static void print_edge(const char* tag, const Constraint::Ptr& c) {
  SE3Constraint::Ptr se3 = ptr::dynamic_pointer_cast<SE3Constraint>(c);
  std::printf("%s", tag);
  for (int r = 0; r < 3; ++r)
    for (int col = 0; col < 4; ++col) std::printf(" %a", se3->getRelativePose()(r, col));
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s <folder> scan1.bin scan2.bin\n", argv[0]); return 2; }
  const std::string folder = argv[1];
  Logger logger;
  logger.setLogLevel(ERROR);
  try {
    PointCloudSensor sensor("velodyne", &logger);
    sensor.setPrepassCache(true);   // (off by default, like the reference: nothing is kept between calls unless asked for)
    Uuid id[2];
    {
      PointCloudMeasurement::Ptr m[2];
      for (int i = 0; i < 2; ++i) {
        m[i].reset(new PointCloudMeasurement(load_bin(argv[2 + i]), "robot", sensor.getName(), Transform::Identity()));
        id[i] = m[i]->getUniqueId();
      }
      std::printf("save before any registration %d\n", (int)sensor.saveDeviceCache(m[0], folder + "/0.s3dc"));
      print_edge("first", sensor.createConstraint(m[0], m[1], Transform::Identity(), false));
      for (int i = 0; i < 2; ++i)
        std::printf("save %d %d\n", i, (int)sensor.saveDeviceCache(m[i], folder + "/" + std::to_string(i) + ".s3dc"));
    }   // the measurements, their device copies and their cache entries are gone
    std::printf("entries after release %lld\n", sensor.getCacheStats().entries);
    PointCloudMeasurement::Ptr r[2];
    for (int i = 0; i < 2; ++i) {
      r[i].reset(new PointCloudMeasurement(load_bin(argv[2 + i]), "robot", sensor.getName(), Transform::Identity(), id[i]));
      std::printf("uuid kept %d\n", (int)(r[i]->getUniqueId() == id[i]));
    }
    // the reloaded measurements go to the device in one bulk hand-over; a second call finds nothing left to upload
    const size_t pre1 = sensor.preloadDeviceClouds({r[0], r[1], r[0]});
    const size_t pre2 = sensor.preloadDeviceClouds({r[0], r[1]});
    std::printf("preload %d %d\n", (int)pre1, (int)pre2);
    std::printf("load wrong scan %d\n", (int)sensor.loadDeviceCache(r[0], folder + "/1.s3dc"));
    std::printf("load missing file %d\n", (int)sensor.loadDeviceCache(r[0], folder + "/7.s3dc"));
    for (int i = 0; i < 2; ++i)
      std::printf("load %d %d\n", i, (int)sensor.loadDeviceCache(r[i], folder + "/" + std::to_string(i) + ".s3dc"));
    const s3d_cache_stats before = sensor.getCacheStats();
    print_edge("again", sensor.createConstraint(r[0], r[1], Transform::Identity(), false));
    const s3d_cache_stats after = sensor.getCacheStats();
    std::printf("entries %lld new hits %lld new misses %lld\n", after.entries, after.hits - before.hits, after.misses - before.misses);
  } catch (const std::exception& e) {
    std::printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
