// Host-only probe of slam3d::PointCloudSensor::readPLY (no context, no GPU): prints what loadPLY would wrap.
#include <cstdio>

#include "slam3d/sensor/pcl/PointCloudSensor.hpp"

using namespace slam3d;

int main(int argc, char** argv) {
  PointCloud c;
  Transform T;
  const int r = PointCloudSensor::readPLY(argv[1], c, T);
  std::printf("%d %zu\n", r, c.size());
  for (const PointType& p : c.points) std::printf("%a %a %a\n", p.x, p.y, p.z);
  for (int i = 0; i < 3; ++i) std::printf("%g %g %g %g\n", T(i, 0), T(i, 1), T(i, 2), T(i, 3));
  return 0;
}
