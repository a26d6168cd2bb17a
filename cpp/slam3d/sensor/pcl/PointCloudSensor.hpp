// slam3d::PointCloudSensor on the MI355X back-end — include path, class names, method names, argument meaning and
// exceptions of the reference's slam3d/sensor/pcl/PointCloudSensor.hpp:43-243:
//   createConstraint (PointCloudSensor.cpp:269-299), createCombinedMeasurement (:258-266), getAccumulatedCloud
//   (:235-256), buildMap (:301-318), removeOutliers (:211-226), fillGroundPlane (:362-388), loadPLY (:390-417),
//   setRegistrationParameters (:320-340), downsample (:190-201), downsampleScan (:203-209), transform (:228-233) and
//   the scalar setters.
// The arithmetic runs in libslam3d_hip.so through the C ABI (include/slam3d_hip.h); this header only does what the
// reference's own lines around PCL do: casts, exceptions, logging.  Not mirrored: the ScanSensor front-end policy
// (addMeasurement, link, linkToNeighbors) and the Mapper / Graph of slam3d core it talks to (out of
// scope; the candidate rules are restated in slam3d_amd/posegraph.py); the measurements of a patch therefore come
// from a MeasurementStorage handed in directly instead of mMapper->getGraph().
//
// pcl::PointXYZ / pcl::PointCloud are used where PCL's headers are installed (PointCloudSensor.hpp:43-44); this
// build environment has no PCL, and the stand-ins below carry the members the path touches.
#pragma once

#include <cstdint>
#include <mutex>
#include <sstream>

#include "../../../../include/slam3d_hip.h"
#include "../../core/Types.hpp"
#include "RegistrationParameters.hpp"

namespace slam3d {

static_assert(sizeof(RegistrationParameters) == sizeof(s3d_reg_params), "RegistrationParameters is the C ABI struct");

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>) && !defined(S3D_MIRROR_NO_PCL)
#define S3D_MIRROR_HAVE_PCL 1
#endif
#endif

#if defined(S3D_MIRROR_HAVE_PCL)
}  // namespace slam3d
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace slam3d {
typedef pcl::PointXYZ PointType;                 // PointCloudSensor.hpp:43
typedef pcl::PointCloud<PointType> PointCloud;   // PointCloudSensor.hpp:44
inline PointType makePoint(float x, float y, float z) { return PointType(x, y, z); }
#else
// pcl::PointXYZ: 16 bytes (x, y, z, padding)
struct PointType { float x, y, z, data_w; };
inline PointType makePoint(float x, float y, float z) { return PointType{x, y, z, 1.f}; }
struct PointCloudHeader { uint64_t stamp = 0; uint32_t seq = 0; std::string frame_id; };

// the members of pcl::PointCloud<pcl::PointXYZ> the path touches
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud> Ptr;
  typedef std::shared_ptr<const PointCloud> ConstPtr;
  std::vector<PointType> points;
  PointCloudHeader header;
  uint32_t width = 0, height = 1;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  void push_back(const PointType& p) { points.push_back(p); width = (uint32_t)points.size(); }
  PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = (uint32_t)points.size(); return *this; }
};
#endif  // S3D_MIRROR_HAVE_PCL

// PointCloudSensor.hpp:50-100
// The HBM copy of a measurement's cloud.  Not in the reference: it is what lets a scan be uploaded once and
// then registered against every neighbour (ScanSensor::linkToNeighbors), and a loop-closure patch go from
// createCombinedMeasurement into createConstraint without visiting the host.
struct ContextHolder {   // the sensor's s3d_context; device clouds refer to it weakly (a measurement may outlive its sensor)
  explicit ContextHolder(s3d_context* c) : ctx(c) {}
  ~ContextHolder() { s3d_context_destroy(ctx); }
  ContextHolder(const ContextHolder&) = delete;
  ContextHolder& operator=(const ContextHolder&) = delete;
  s3d_context* ctx;
};
struct DeviceCloud {
  DeviceCloud(const std::shared_ptr<ContextHolder>& owner, s3d_cloud* c) : context(owner), cloud(c) {}
  // released through its context while that is alive: the context then also drops the cloud's cached pre-pass products
  ~DeviceCloud() { std::shared_ptr<ContextHolder> h = context.lock(); s3d_cloud_release(h ? h->ctx : nullptr, cloud); }
  DeviceCloud(const DeviceCloud&) = delete;
  DeviceCloud& operator=(const DeviceCloud&) = delete;
  std::weak_ptr<ContextHolder> context;
  s3d_cloud* cloud;
};

class PointCloudMeasurement : public Measurement {
 public:
  typedef ptr::shared_ptr<PointCloudMeasurement> Ptr;   // boost::shared_ptr where Boost is installed
  PointCloudMeasurement(const PointCloud::Ptr& cloud, const std::string& r, const std::string& s, const Transform& p,
                        const Uuid& id = Uuid())
      : Measurement(r, s, p, id), mPointCloud(cloud) {}
  const PointCloud::Ptr getPointCloud() const { return mPointCloud; }
  const char* getTypeName() const override { return "slam3d::PointCloudMeasurement"; }
  // device-resident copy, attached lazily by the sensor (the host cloud must not be modified afterwards)
  std::shared_ptr<DeviceCloud> getDeviceCloud() const { std::lock_guard<std::mutex> l(mDeviceMutex); return mDeviceCloud; }
  void setDeviceCloud(std::shared_ptr<DeviceCloud> d) const { std::lock_guard<std::mutex> l(mDeviceMutex); mDeviceCloud = d; }
 protected:
  PointCloud::Ptr mPointCloud;
  mutable std::mutex mDeviceMutex;
  mutable std::shared_ptr<DeviceCloud> mDeviceCloud;
};

// the part of Sensor / ScanSensor (Sensor.hpp:84-168, ScanSensor.hpp:35-158) the path needs.  Inside a slam3d source
// tree the real classes are used instead (define S3D_MIRROR_USE_SLAM3D_CORE, or let __has_include find
// <slam3d/core/ScanSensor.hpp>): this header next to the real one would otherwise redefine them.  [That branch has
// not been through a compiler here - slam3d's headers need Boost and Eigen, which this image lacks.]
#if !defined(S3D_MIRROR_STANDALONE) && (defined(S3D_MIRROR_USE_SLAM3D_CORE) || \
    (defined(__has_include) && __has_include(<slam3d/core/ScanSensor.hpp>)))
#include <slam3d/core/ScanSensor.hpp>
#else
class Sensor {
 public:
  Sensor(const std::string& n, Logger* l) : mLogger(l), mName(n), mCovarianceScale(1.0) {}
  virtual ~Sensor() {}
  std::string getName() const { return mName; }
  void setCovarianceScale(ScalarType s) { mCovarianceScale = s; }
 protected:
  Logger* mLogger;
  std::string mName;
  ScalarType mCovarianceScale;
};

class ScanSensor : public Sensor {
 public:
  ScanSensor(const std::string& n, Logger* l) : Sensor(n, l) {}
  // ScanSensor.hpp:105 — builds the patch a loop-closure link is registered against (ScanSensor.cpp:269)
  virtual Measurement::Ptr createCombinedMeasurement(const VertexObjectList& vertices, Transform pose) const = 0;
  // ScanSensor.hpp:122-125 — THE plugin hook of the hot path
  virtual Constraint::Ptr createConstraint(const Measurement::Ptr& source, const Measurement::Ptr& target,
                                           const Transform& odometry, bool loop) = 0;
};
#endif

class PointCloudSensor : public ScanSensor {
 public:
  // reserved_cus > 0: that many compute units of the device are set aside for createConstraint (the application
  // thread's blocking call per new scan) and the sweeps of createConstraints run on the others
  // (s3d_context_create_cu_mask / s3d_sweep_create_cu_mask): with 32 of an MI355X's 256 a registration issued while
  // a sweep is running takes 2.1 ms instead of 5.8 (1.4 ms on the idle GPU), the sweep 9 % longer
  PointCloudSensor(const std::string& n, Logger* l, int device = 0, int reserved_cus = 0);
  ~PointCloudSensor();

  Constraint::Ptr createConstraint(const Measurement::Ptr& source, const Measurement::Ptr& target,
                                   const Transform& odometry, bool loop) override;
  void setRegistrationParameters(const RegistrationParameters& param, bool coarse);
  void setScanResolution(double r);
  void setMapResolution(double r);
  void setMapOutlierRemoval(double r, unsigned n);
  static PointCloud::Ptr downsample(PointCloud::Ptr source, double resolution);
  PointCloud::Ptr downsampleScan(PointCloud::Ptr source);
  PointCloud::Ptr transform(PointCloud::ConstPtr source, const Transform tf) const;

  // ---- patches and maps (reference: PointCloudSensor.hpp:127, :200-216).  The reference reaches the
  // measurements through mMapper->getGraph(); the mirror has no Mapper/Graph, the storage is handed in directly.
  void setMeasurementStorage(MeasurementStorage* s) { mStorage = s; }
  // Not in the reference (PointCloudSensor.cpp:127-131 re-filters both clouds in every align()): keep the voxel
  // filter / search grid / k-NN normals of every measurement's device cloud in HBM between createConstraint calls
  // (s3d_exec_options.cache_prepass).  Off by default, like the C ABI and the reference; worth switching on in a
  // mapper, which links every scan to several others (ScanSensor.cpp:113, :179-201).  Results are bit-identical either way.
  void setPrepassCache(bool on) { mPrepassCache = on; }
  PointCloud::Ptr removeOutliers(PointCloud::Ptr source, double radius, unsigned min_neighbors) const;
  PointCloud::Ptr getAccumulatedCloud(const VertexObjectList& vertices) const;
  Measurement::Ptr createCombinedMeasurement(const VertexObjectList& vertices, Transform pose) const override;
  PointCloud::Ptr buildMap(const VertexObjectList& vertices) const;
  // reference PointCloudSensor.hpp:227: fit the ground plane (RANSAC, scored on the device) and append rings of
  // points on it out to `radius` at the map resolution
  void fillGroundPlane(PointCloud::Ptr cloud, ScalarType radius);
  // reference PointCloudSensor.hpp:234: read a PLY file as the initial map.  The reference adds it to the pose
  // graph as vertex 0 with an identity PoseConstraint (PointCloudSensor.cpp:401-405); the mirror has no Mapper /
  // Graph: the measurement goes to the MeasurementStorage (if set) and is returned by getInitialMap().
  void loadPLY(const std::string& path, const std::string& robot);
  PointCloudMeasurement::Ptr getInitialMap() const { return mInitialMap; }
  // what pcl::PLYReader::read gives loadPLY: the vertex x/y/z and the sensor pose of the `camera` element.
  // Returns 0 on success (like PLYReader::read), -1 otherwise.
  static int readPLY(const std::string& path, PointCloud& cloud, Transform& sensor_pose);

  // Not in the reference: the candidate pairs of one ScanSensor::linkToNeighbors call (ScanSensor.cpp:179-201, which
  // links them one blocking createConstraint at a time) registered as ONE sweep over the GPUs of the node
  // (include/slam3d_hip.h, C1: one rank per device, contiguous blocks, RCCL all-gather of the edges).  Entry i is the
  // constraint createConstraint(sources[i], targets[i], odometry[i], false) would return - bit for bit, whatever the
  // number of GPUs - or null where that call would have thrown NoMatch (logged as ScanSensor.cpp:159-162 does).
  // devices: HIP device of every rank; empty = every visible device.
  // loop = true is what ScanSensor::link passes (ScanSensor.cpp:156: createConstraint(source_m, target_m, guess, true)):
  // a first sweep with the COARSE configuration whose results are the guesses of the fine sweep
  // (PointCloudSensor.cpp:286-292); a candidate whose coarse registration fails is a NoMatch, as in the reference.
  // Sweep clouds are kept per measurement (uploaded once, pre-pass cached per rank) while the measurement is alive
  // and recently used: entries of measurements that no longer exist (the patches ScanSensor::buildPatch makes for
  // every candidate have a fresh uuid each) are released at the next call, and at most getSweepCloudLimit() are kept.
  std::vector<Constraint::Ptr> createConstraints(const std::vector<Measurement::Ptr>& sources,
                                                 const std::vector<Measurement::Ptr>& targets,
                                                 const std::vector<Transform>& odometry,
                                                 const std::vector<int>& devices = std::vector<int>(), bool loop = false);
  void setSweepCloudLimit(size_t n) { mSweepCloudLimit = n; }
  size_t getSweepCloudLimit() const { return mSweepCloudLimit; }
  size_t getSweepCloudCount() const { return mSweepClouds.size(); }
  // GICP_OMP / NDT_OMP (PointCloudSensor.cpp:149-162).  true (default): like a reference built WITH pclomp, the two
  // enumerators run (GICP_OMP: the GICP device code; NDT_OMP: the NDT code over pclomp's DIRECT7 neighbourhood).  false: like a reference built without it,
  // align() throws std::runtime_error("OMP is not available, ...") (s3d_exec_options.omp_unavailable).
  void setOmpAvailable(bool on) { mOmpAvailable = on; }

  // Not in the reference: checkpoints.  GraphSerialization::toFolder writes one <index>.s3dm archive per vertex
  // (GraphSerialization.cpp:40-47) and fromFolder builds NEW measurement objects from them (:68-135): their device
  // copies and cached pre-pass products (voxel filter, search grid, k-NN normals per registration configuration) would
  // have to be computed again.  saveDeviceCache writes what the library holds for `m` into `file` (e.g.
  // <index>.s3dc next to the archive; false and no file when nothing is cached), loadDeviceCache uploads the
  // reloaded measurement and installs the file's content for it (false when the file is absent, damaged or was
  // made from other points - the first registration then simply recomputes).  Results do not depend on either.
  bool saveDeviceCache(const PointCloudMeasurement::Ptr& m, const std::string& file) const;
  bool loadDeviceCache(const PointCloudMeasurement::Ptr& m, const std::string& file);
  // Not in the reference: the device copies of MANY measurements in one bulk hand-over (s3d_cloud_upload_many: one
  // device allocation, pinned staging by host threads, 40 GB/s instead of one allocation + staged copy + wait per
  // scan) - the measurements of a graph that fromFolder has just rebuilt, or the scans of a first loop-closure sweep.
  // Measurements that have a device copy already are left alone; returns how many were uploaded.  A measurement that
  // is not preloaded is uploaded on its first use as before.
  size_t preloadDeviceClouds(const std::vector<PointCloudMeasurement::Ptr>& measurements) const;
  // entries / bytes / hits / misses of the context's pre-pass cache (s3d_context_cache_control)
  s3d_cache_stats getCacheStats() const { s3d_cache_stats st = {0, 0, 0, 0}; s3d_context_cache_control(mContext, 0, 0, &st); return st; }

  // the same align() the reference keeps file-local (PointCloudSensor.cpp:119-174), exposed for tests
  Transform align(const PointCloudMeasurement::Ptr& source, const PointCloudMeasurement::Ptr& target,
                  const Transform& guess, const RegistrationParameters& config);

 protected:
  RegistrationParameters mFineConfiguration;
  RegistrationParameters mCoarseConfiguration;
  double mScanResolution;
  double mMapResolution;
  double mMapOutlierRadius;
  unsigned mMapOutlierNeighbors;

 private:
  // upload-once cache: the device copy of a measurement's cloud
  std::shared_ptr<DeviceCloud> deviceCloudOf(const PointCloudMeasurement::Ptr& m) const;
  // clouds + poses (correctedPose * sensorPose) of the vertices, as the C ABI takes them
  void gather(const VertexObjectList& vertices, std::vector<std::shared_ptr<DeviceCloud>>& keep,
              std::vector<s3d_cloud*>& clouds, std::vector<double>& poses) const;
  PointCloud::Ptr download(s3d_cloud* c) const;
  MeasurementStorage* mStorage = nullptr;
  bool mPrepassCache = false;
  PointCloudMeasurement::Ptr mInitialMap;
  // the sweep (ranks, communicators) of the last device list and the sweep clouds of the measurements it has seen
  struct SweepEntry { s3d_sweep_cloud* cloud; ptr::weak_ptr<Measurement> owner; unsigned long long last_use; };
  std::vector<uint32_t> mSweepCuMask;    // empty: the sweeps use the whole device
  s3d_sweep* mSweep = nullptr;
  std::vector<int> mSweepDevices;
  std::map<Uuid, SweepEntry> mSweepClouds;
  size_t mSweepCloudLimit = 1024;
  unsigned long long mSweepClock = 0;
  std::mutex mSweepMutex;
  bool mOmpAvailable = true;
  void pruneSweepClouds();
  std::vector<s3d_edge_record> runSweep(const std::vector<s3d_sweep_cloud*>& src, const std::vector<s3d_sweep_cloud*>& tgt,
                                        const std::vector<double>& guesses, const RegistrationParameters& config);
  void releaseSweep();
  std::shared_ptr<ContextHolder> mContextHolder;
  s3d_context* mContext;   // == mContextHolder->ctx: one HIP device + stream; calls are serialised inside the library,
                           // so createConstraint may be entered from the link thread (ScanSensor.cpp:210)
};

}  // namespace slam3d
