// fillGroundPlane / loadPLY through the C++ mirror (reference PointCloudSensor.hpp:227, :234).
//   example_ground_plane scan.bin radius ring_out.bin [map.ply]
// scan.bin: float32 x,y,z,intensity per point.  Writes the appended ring points (packed float32 xyz) to
// ring_out.bin and prints "FILLED <n_before> <n_after>"; with a PLY file also "PLY <n_points> <tx> <ty> <tz>".
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slam3d/sensor/pcl/PointCloudSensor.hpp"

using namespace slam3d;

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s scan.bin radius ring_out.bin [map.ply]\n", argv[0]); return 2; }
  Logger logger;
  PointCloudSensor sensor("velodyne", &logger);
  PointCloud::Ptr cloud(new PointCloud);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  float p[4];
  while (std::fread(p, sizeof(float), 4, f) == 4) cloud->push_back(makePoint(p[0], p[1], p[2]));
  std::fclose(f);
  const size_t before = cloud->size();
  sensor.fillGroundPlane(cloud, std::atof(argv[2]));
  std::printf("FILLED %zu %zu\n", before, cloud->size());
  FILE* o = std::fopen(argv[3], "wb");
  if (!o) return 2;
  for (size_t i = before; i < cloud->size(); ++i) std::fwrite(&cloud->points[i].x, sizeof(float), 3, o);
  std::fclose(o);
  if (argc > 4) {
    MeasurementStorage storage;
    sensor.setMeasurementStorage(&storage);
    sensor.loadPLY(argv[4], "robot");
    PointCloudMeasurement::Ptr m = sensor.getInitialMap();
    if (!m) { std::printf("PLY failed\n"); return 1; }
    const Position t = m->getSensorPose().translation();
    std::printf("PLY %zu %g %g %g %d\n", m->getPointCloud()->size(), t[0], t[1], t[2], (int)storage.contains(m->getUniqueId()));
  }
  return 0;
}
