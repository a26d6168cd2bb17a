// pcl_gicp.cpp — OPTIONAL pin of the oracle and of the CPU baseline against the real PCL (test infrastructure).
//
// The image this repository is built in has no PCL / Eigen / Boost / FLANN, so the oracle (oracle/s3d_oracle.c) is a
// restatement that nothing reference-held pins ("parity unpinned").  On any host that has libpcl-dev
// (PCL >= 1.8, the reference's own requirement: slam3d-dependencies.cmake:22) this program runs the reference's
// registration call exactly as slam3d/sensor/pcl/PointCloudSensor.cpp does it - pcl::VoxelGrid of both clouds
// (:190-201), the 100-point gate (:134-135), pcl::GeneralizedIterativeClosestPoint with the seven setters of
// :59-65, source/target swapped as in :68-69, align(guess) (:70), getFitnessScore(max_correspondence_distance)
// (:73) - and prints one JSON record per case.  tests/test_pcl_pin.py compares the oracle with these records when
// tests/golden/pcl_golden.json exists (written by `make -C oracle/pcl golden`) and skips otherwise.
//
//   pcl_gicp golden <cloud1.bin> <cloud2.bin> <cloud3.bin> <cloud4.bin>      five fixture cases -> JSON array
//   pcl_gicp bench  <source.xyz.bin> <target.xyz.bin> <leaf> <iters> <reps>  timing of one pair (packed xyz floats)
//
// It could not be compiled in the build environment; it uses only PCL's documented public API.
#include <pcl/filters/voxel_grid.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/registration/gicp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

typedef pcl::PointXYZ PointType;
typedef pcl::PointCloud<PointType> PointCloud;

struct Params {   // slam3d::RegistrationParameters defaults (RegistrationParameters.hpp:36-97)
  double point_cloud_density = 0.2, max_fitness_score = 2.0, max_translation = 1.0, max_rotation = 1.0;
  double euclidean_fitness_epsilon = 1.0, transformation_epsilon = 1e-5, max_correspondence_distance = 2.5;
  int maximum_iterations = 50;
  double rotation_epsilon = 2e-3;
  int correspondence_randomness = 20, maximum_optimizer_iterations = 20;
};

static PointCloud::Ptr load(const std::string& path, int stride) {
  PointCloud::Ptr c(new PointCloud);
  std::ifstream f(path, std::ios::binary);
  std::vector<float> v((size_t)stride);
  while (f.read(reinterpret_cast<char*>(v.data()), sizeof(float) * stride)) c->push_back(PointType(v[0], v[1], v[2]));
  return c;
}

static PointCloud::Ptr downsample(PointCloud::Ptr in, double leaf) {   // PointCloudSensor.cpp:190-201
  PointCloud::Ptr out(new PointCloud);
  if (in->size() > 0) {
    pcl::VoxelGrid<PointType> grid;
    grid.setLeafSize(leaf, leaf, leaf);
    grid.setInputCloud(in);
    grid.filter(*out);
  }
  return out;
}

struct Result { int status; Eigen::Matrix4f T; double fitness; bool converged; size_t ns, nt; double seconds; };

// align() + doICP<GICP> of the reference (PointCloudSensor.cpp:52-82, :119-174); status as enum s3d_status
static Result align(PointCloud::Ptr source, PointCloud::Ptr target, const Eigen::Matrix4f& guess, const Params& p) {
  Result r{0, Eigen::Matrix4f::Identity(), 0.0, false, 0, 0, 0.0};
  const auto t0 = std::chrono::steady_clock::now();
  PointCloud::Ptr fs = source, ft = target;
  if (p.point_cloud_density > 0) { fs = downsample(source, p.point_cloud_density); ft = downsample(target, p.point_cloud_density); }
  r.ns = fs->size(); r.nt = ft->size();
  if (ft->size() < 100 || fs->size() < 100) { r.status = 1; return r; }
  pcl::GeneralizedIterativeClosestPoint<PointType, PointType> icp;
  icp.setMaxCorrespondenceDistance(p.max_correspondence_distance);
  icp.setMaximumIterations(p.maximum_iterations);
  icp.setTransformationEpsilon(p.transformation_epsilon);
  icp.setEuclideanFitnessEpsilon(p.euclidean_fitness_epsilon);
  icp.setCorrespondenceRandomness(p.correspondence_randomness);
  icp.setMaximumOptimizerIterations(p.maximum_optimizer_iterations);
  icp.setRotationEpsilon(p.rotation_epsilon);
  icp.setInputSource(ft);   // :68-69: source and target are swapped on purpose
  icp.setInputTarget(fs);
  PointCloud result;
  icp.align(result, guess);
  r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  r.converged = icp.hasConverged();
  r.fitness = icp.getFitnessScore(p.max_correspondence_distance);
  r.T = icp.getFinalTransformation();
  if (!r.converged) { r.status = 2; return r; }
  if (r.fitness > p.max_fitness_score) { r.status = 3; return r; }
  const Eigen::Isometry3d Td(Eigen::Isometry3f(r.T).cast<double>()), Gd(Eigen::Isometry3f(guess).cast<double>());
  const Eigen::Isometry3d delta = Gd.inverse() * Td;                                   // :167
  if (delta.translation().norm() > p.max_translation || Eigen::AngleAxisd(delta.linear()).angle() > p.max_rotation)
    r.status = 4;
  return r;
}

static void print(const char* name, const Result& r, bool last) {
  std::printf(" {\"case\": \"%s\", \"status\": %d, \"converged\": %d, \"fitness\": %.17g, \"n_source_filtered\": %zu, "
              "\"n_target_filtered\": %zu, \"seconds\": %.6f, \"T\": [", name, r.status, (int)r.converged, r.fitness, r.ns, r.nt,
              r.seconds);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) std::printf("%.9g%s", (double)r.T(i, j), (i == 3 && j == 3) ? "" : ", ");   // row-major
  std::printf("]}%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
  if (argc >= 6 && std::string(argv[1]) == "golden") {
    PointCloud::Ptr c[4];
    for (int i = 0; i < 4; ++i) c[i] = load(argv[2 + i], 4);   // KITTI layout: x, y, z, intensity
    Params p;
    Eigen::Matrix4f g2 = Eigen::Matrix4f::Identity();
    g2(0, 3) = 2.0f;
    std::printf("[\n");
    print("1->2", align(c[0], c[1], Eigen::Matrix4f::Identity(), p), false);
    print("2->3", align(c[1], c[2], Eigen::Matrix4f::Identity(), p), false);
    print("3->4", align(c[2], c[3], Eigen::Matrix4f::Identity(), p), false);
    print("1->4 guess x=2", align(c[0], c[3], g2, p), false);
    print("1->4 identity", align(c[0], c[3], Eigen::Matrix4f::Identity(), p), true);   // must fail the distance gate
    std::printf("]\n");
    return 0;
  }
  if (argc >= 7 && std::string(argv[1]) == "bench") {
    PointCloud::Ptr s = load(argv[2], 3), t = load(argv[3], 3);
    Params p;
    p.point_cloud_density = std::atof(argv[4]);
    p.maximum_iterations = std::atoi(argv[5]);
    const int reps = std::atoi(argv[6]);
    double total = 0;
    Result r{};
    for (int i = 0; i < reps; ++i) { r = align(s, t, Eigen::Matrix4f::Identity(), p); total += r.seconds; }
    std::printf("[\n");
    print("bench", r, true);
    std::printf("]\n");
    std::fprintf(stderr, "{\"registrations_per_s\": %.6f, \"cores\": 1, \"kind\": \"reference\", \"reps\": %d}\n", reps / total, reps);
    return 0;
  }
  std::fprintf(stderr, "usage: pcl_gicp golden c1.bin c2.bin c3.bin c4.bin | pcl_gicp bench src.xyz.bin tgt.xyz.bin leaf iters reps\n");
  return 2;
}
