// Host side of slam3d::PointCloudSensor on the MI355X back-end — the counterpart of the
// reference's slam3d/sensor/pcl/PointCloudSensor.cpp.  Everything numerical happens behind
// include/slam3d_hip.h; status codes are re-raised as the reference's exceptions with the
// reference's messages.
#include "PointCloudSensor.hpp"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <iostream>

namespace slam3d {

void Logger::message(LogLevel lvl, const std::string& msg) {
  if (lvl < mLogLevel) return;
  static const char* names[] = {"[DEBUG] ", "[INFO] ", "[WARN] ", "[ERROR] ", "[FATAL] "};
  std::cerr << names[lvl] << msg << std::endl;
}

Uuid generateUuid() {
  static std::atomic<unsigned long long> counter{0};
  char buf[48];
  std::snprintf(buf, sizeof buf, "s3d-%016llx", counter.fetch_add(1) + 1);
  return buf;
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

PointCloudSensor::PointCloudSensor(const std::string& n, Logger* l, int device, int reserved_cus)
    : ScanSensor(n, l), mContext(nullptr) {
  mScanResolution = 0.1;      // PointCloudSensor.cpp:179-182
  mMapResolution = 0.1;
  mMapOutlierRadius = 0.2;
  mMapOutlierNeighbors = 3;
  if (s3d_abi_version() != S3D_ABI_VERSION)
    throw std::runtime_error("slam3d (MI355X build): libslam3d_hip.so does not match the headers this mirror was built with");
  // createConstraint - the application thread's blocking call per new scan (ScanSensor.cpp:113) - runs on a HIGH-priority
  // context: the detached linkToNeighbors thread (:209-210) sweeps its candidates on the sweep's own contexts
  // (createConstraints), and the dispatcher takes this context's kernels before their queued blocks
  int st;
  if (reserved_cus > 0) {
    uint32_t mine[32], rest[32];
    const int words = s3d_cu_masks(device, reserved_cus, mine, rest, 32);
    if (words <= 0) throw std::invalid_argument("PointCloudSensor: reserved_cus must be below the device's compute units");
    mSweepCuMask.assign(rest, rest + words);
    st = s3d_context_create_cu_mask(device, mine, words, &mContext);
  } else {
    st = s3d_context_create_priority(device, 1, &mContext);
  }
  if (st != S3D_STATUS_OK)
    throw std::runtime_error("slam3d (MI355X build): no usable HIP device, and there is no CPU fallback");
  mContextHolder = std::make_shared<ContextHolder>(mContext);
}

PointCloudSensor::~PointCloudSensor() { releaseSweep(); }   // (the context goes with mContextHolder)

void PointCloudSensor::releaseSweep() {
  if (!mSweep) return;
  for (auto& kv : mSweepClouds) s3d_sweep_cloud_release(mSweep, kv.second.cloud);
  mSweepClouds.clear();
  s3d_sweep_destroy(mSweep);
  mSweep = nullptr;
}

// sweep clouds whose measurement is gone, then the least recently used ones beyond the limit (never one of the
// current call: those carry the current clock)
void PointCloudSensor::pruneSweepClouds() {
  for (auto it = mSweepClouds.begin(); it != mSweepClouds.end();) {
    if (it->second.owner.expired()) {
      s3d_sweep_cloud_release(mSweep, it->second.cloud);
      it = mSweepClouds.erase(it);
    } else {
      ++it;
    }
  }
  while (mSweepClouds.size() > mSweepCloudLimit) {
    auto lru = mSweepClouds.end();
    for (auto it = mSweepClouds.begin(); it != mSweepClouds.end(); ++it)
      if (it->second.last_use != mSweepClock && (lru == mSweepClouds.end() || it->second.last_use < lru->second.last_use)) lru = it;
    if (lru == mSweepClouds.end()) break;
    s3d_sweep_cloud_release(mSweep, lru->second.cloud);
    mSweepClouds.erase(lru);
  }
}

std::vector<s3d_edge_record> PointCloudSensor::runSweep(const std::vector<s3d_sweep_cloud*>& src,
                                                        const std::vector<s3d_sweep_cloud*>& tgt,
                                                        const std::vector<double>& guesses,
                                                        const RegistrationParameters& config) {
  std::vector<s3d_edge_record> rec(src.size());
  if (src.empty()) return rec;
  s3d_exec_options opts;
  std::memset(&opts, 0, sizeof opts);
  opts.cache_prepass = mPrepassCache ? 1 : 0;
  opts.omp_unavailable = mOmpAvailable ? 0 : 1;
  const int st = s3d_align_batch_multi(mSweep, (int)src.size(), src.data(), tgt.data(), guesses.data(),
                                       static_cast<const s3d_reg_params*>(&config), &opts, rec.data());
  if (st == S3D_STATUS_BACKEND_ERROR) throw std::runtime_error(std::string("HIP back-end error: ") + s3d_sweep_last_error(mSweep));
  if (st == S3D_STATUS_UNKNOWN_ALGORITHM) throw std::runtime_error("Unknown registration algorithm specified.");   // :164
  if (st == S3D_STATUS_OMP_UNAVAILABLE)   // :161
    throw std::runtime_error("OMP is not available, you need to rebuild SLAM3D with OMP or use another matching algorithm.");
  return rec;
}

std::vector<Constraint::Ptr> PointCloudSensor::createConstraints(const std::vector<Measurement::Ptr>& sources,
                                                                 const std::vector<Measurement::Ptr>& targets,
                                                                 const std::vector<Transform>& odometry,
                                                                 const std::vector<int>& devices, bool loop) {
  if (sources.size() != targets.size() || sources.size() != odometry.size())
    throw std::invalid_argument("createConstraints: sources, targets and odometry must have one entry per candidate");
  const size_t n = sources.size();
  std::vector<Constraint::Ptr> out(n);
  if (n == 0) return out;
  std::lock_guard<std::mutex> lock(mSweepMutex);
  if (!mSweep || devices != mSweepDevices) {
    releaseSweep();
    const int sst = mSweepCuMask.empty()
                        ? s3d_sweep_create((int)devices.size(), devices.empty() ? nullptr : devices.data(), &mSweep)
                        : s3d_sweep_create_cu_mask((int)devices.size(), devices.empty() ? nullptr : devices.data(),
                                                   mSweepCuMask.data(), (int)mSweepCuMask.size(), &mSweep);
    if (sst != S3D_STATUS_OK)
      throw std::runtime_error("slam3d (MI355X build): no usable HIP devices / RCCL for the sweep, and there is no CPU fallback");
    mSweepDevices = devices;
  }
  ++mSweepClock;
  std::vector<s3d_sweep_cloud*> src(n), tgt(n);
  std::vector<double> guesses(16 * n);
  for (size_t i = 0; i < n; ++i) {
    const Measurement::Ptr pair_m[2] = {sources[i], targets[i]};
    s3d_sweep_cloud* pair_c[2];
    for (int k = 0; k < 2; ++k) {
      PointCloudMeasurement::Ptr pc = ptr::dynamic_pointer_cast<PointCloudMeasurement>(pair_m[k]);
      if (!pc) {   // PointCloudSensor.cpp:279-283
        mLogger->message(ERROR, "Measurement given to createConstraint() is not a PointCloud!");
        throw BadMeasurementType();
      }
      auto it = mSweepClouds.find(pc->getUniqueId());
      if (it == mSweepClouds.end()) {
        const PointCloud::Ptr c = pc->getPointCloud();
        static const float dummy[4] = {0, 0, 0, 0};
        s3d_sweep_cloud* sc = nullptr;
        if (s3d_sweep_cloud_create(mSweep, c->size() ? &c->points[0].x : dummy, (int)c->size(), 4, &sc) != S3D_STATUS_OK)
          throw std::runtime_error("s3d_sweep_cloud_create failed");
        SweepEntry e;
        e.cloud = sc; e.owner = pair_m[k]; e.last_use = mSweepClock;
        it = mSweepClouds.emplace(pc->getUniqueId(), e).first;
      }
      it->second.last_use = mSweepClock;
      pair_c[k] = it->second.cloud;
    }
    src[i] = pair_c[0]; tgt[i] = pair_c[1];
    const Transform guess = sources[i]->getInverseSensorPose() * odometry[i] * targets[i]->getSensorPose();   // :274
    std::memcpy(&guesses[16 * i], guess.data(), 16 * sizeof(double));
  }
  pruneSweepClouds();
  // loop: the coarse sweep first, its transforms are the guesses of the fine one (:286-292)
  std::vector<size_t> live(n);
  for (size_t i = 0; i < n; ++i) live[i] = i;
  if (loop) {
    const std::vector<s3d_edge_record> coarse = runSweep(src, tgt, guesses, mCoarseConfiguration);
    std::vector<size_t> ok;
    for (size_t i = 0; i < n; ++i) {
      if ((int)coarse[i].status != S3D_STATUS_OK) {   // createConstraint would have thrown NoMatch out of the coarse align()
        mLogger->message(WARNING, "Failed to match candidate " + std::to_string(i) + " (coarse, status " + std::to_string((int)coarse[i].status) + ")");
        continue;
      }
      double* g = &guesses[16 * ok.size()];
      for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 3; ++r) g[c * 4 + r] = coarse[i].transform[c * 3 + r];
        g[c * 4 + 3] = c == 3 ? 1.0 : 0.0;
      }
      src[ok.size()] = src[i]; tgt[ok.size()] = tgt[i];
      ok.push_back(i);
    }
    live.swap(ok);
    src.resize(live.size()); tgt.resize(live.size()); guesses.resize(16 * live.size());
  }
  const std::vector<s3d_edge_record> rec = runSweep(src, tgt, guesses, mFineConfiguration);
  for (size_t k = 0; k < live.size(); ++k) {
    const size_t i = live[k];
    if ((int)rec[k].status != S3D_STATUS_OK) {   // NoMatch: ScanSensor::link logs a warning and goes on (ScanSensor.cpp:159-162)
      mLogger->message(WARNING, "Failed to match candidate " + std::to_string(i) + " (status " + std::to_string((int)rec[k].status) + ")");
      continue;
    }
    Transform icp = Transform::Identity();
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r) icp(r, c) = rec[k].transform[c * 3 + r];
    const Transform transform = sources[i]->getSensorPose() * icp * targets[i]->getInverseSensorPose();   // :295
    Covariance<6> information = Covariance<6>::Identity();
    for (unsigned d = 0; d < 6; ++d) information(d, d) = 1.0 / mCovarianceScale;
    out[i] = Constraint::Ptr(new SE3Constraint(mName, transform, information));
  }
  return out;
}

PointCloud::Ptr PointCloudSensor::downsample(PointCloud::Ptr in, double leaf_size) {
  PointCloud::Ptr out(new PointCloud);
  if (in->size() > 0) {  // PointCloudSensor.cpp:193
    std::vector<float> packed(in->size() * 3);
    int n_out = 0;
    const int st = s3d_voxel_downsample(shared_context_for_static_calls(), &in->points[0].x, (int)in->size(), 4,
                                        leaf_size, packed.data(), &n_out);
    if (st != S3D_STATUS_OK) throw std::runtime_error("s3d_voxel_downsample failed");
    out->points.resize(n_out);
    for (int i = 0; i < n_out; ++i) out->points[i] = makePoint(packed[3 * i], packed[3 * i + 1], packed[3 * i + 2]);
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
  // pcl::transformPointCloud with a Matrix4d (:228-233): double product, one rounding per coordinate, in the
  // summation order the device kernel (s3d_cloud_accumulate) uses
  PointCloud::Ptr out(new PointCloud(*source));
  for (PointType& p : out->points) {
    const double x = p.x, y = p.y, z = p.z;
    p.x = (float)((x * tf(0, 0) + y * tf(0, 1)) + (z * tf(0, 2) + tf(0, 3)));
    p.y = (float)((x * tf(1, 0) + y * tf(1, 1)) + (z * tf(1, 2) + tf(1, 3)));
    p.z = (float)((x * tf(2, 0) + y * tf(2, 1)) + (z * tf(2, 2) + tf(2, 3)));
  }
  return out;
}

std::shared_ptr<DeviceCloud> PointCloudSensor::deviceCloudOf(const PointCloudMeasurement::Ptr& m) const {
  std::shared_ptr<DeviceCloud> d = m->getDeviceCloud();
  if (d) return d;
  const PointCloud::Ptr c = m->getPointCloud();
  static const float dummy[4] = {0, 0, 0, 0};
  s3d_cloud* h = nullptr;
  if (s3d_cloud_upload(mContext, c->size() ? &c->points[0].x : dummy, (int)c->size(), 4, &h) != S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  d = std::make_shared<DeviceCloud>(mContextHolder, h);
  m->setDeviceCloud(d);
  return d;
}

size_t PointCloudSensor::preloadDeviceClouds(const std::vector<PointCloudMeasurement::Ptr>& measurements) const {
  std::vector<PointCloudMeasurement::Ptr> todo;
  std::vector<const float*> ptrs;
  std::vector<int> counts;
  static const float dummy[4] = {0, 0, 0, 0};
  for (const PointCloudMeasurement::Ptr& m : measurements) {
    if (!m || m->getDeviceCloud()) continue;
    bool seen = false;
    for (const PointCloudMeasurement::Ptr& t : todo) seen = seen || t == m;
    if (seen) continue;
    const PointCloud::Ptr c = m->getPointCloud();
    todo.push_back(m);
    ptrs.push_back(c->size() ? &c->points[0].x : dummy);
    counts.push_back((int)c->size());
  }
  if (todo.empty()) return 0;
  std::vector<s3d_cloud*> out(todo.size(), nullptr);
  if (s3d_cloud_upload_many(mContext, (int)todo.size(), ptrs.data(), counts.data(), 4, out.data()) != S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  for (size_t i = 0; i < todo.size(); ++i) todo[i]->setDeviceCloud(std::make_shared<DeviceCloud>(mContextHolder, out[i]));
  return todo.size();
}

bool PointCloudSensor::saveDeviceCache(const PointCloudMeasurement::Ptr& m, const std::string& file) const {
  std::shared_ptr<DeviceCloud> d = m->getDeviceCloud();
  if (!d) return false;                                   // never registered on this device: nothing to keep
  const long long need = s3d_cloud_cache_export(mContext, d->cloud, nullptr, 0);
  if (need <= 0) return false;
  std::vector<char> blob((size_t)need);
  if (s3d_cloud_cache_export(mContext, d->cloud, blob.data(), need) != need) return false;
  std::ofstream f(file, std::ios::binary | std::ios::trunc);
  f.write(blob.data(), (std::streamsize)blob.size());
  return (bool)f;
}

bool PointCloudSensor::loadDeviceCache(const PointCloudMeasurement::Ptr& m, const std::string& file) {
  std::ifstream f(file, std::ios::binary | std::ios::ate);
  if (!f) return false;
  const std::streamsize size = f.tellg();
  if (size <= 0) return false;
  std::vector<char> blob((size_t)size);
  f.seekg(0);
  if (!f.read(blob.data(), size)) return false;
  std::shared_ptr<DeviceCloud> d = deviceCloudOf(m);
  const int st = s3d_cloud_cache_import(mContext, d->cloud, blob.data(), (long long)blob.size());
  if (st == S3D_STATUS_BACKEND_ERROR) throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  if (st != S3D_STATUS_OK) mLogger->message(WARNING, "Device cache file " + file + " not used: " + s3d_last_error(mContext));
  return st == S3D_STATUS_OK;
}

PointCloud::Ptr PointCloudSensor::download(s3d_cloud* c) const {
  PointCloud::Ptr out(new PointCloud);
  const int n = s3d_cloud_size(c);
  out->points.resize(n);
  if (n > 0 && s3d_cloud_download(mContext, c, &out->points[0].x, 4) != S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  out->width = (uint32_t)n;
  out->height = 1;
  out->is_dense = true;
  return out;
}

PointCloud::Ptr PointCloudSensor::removeOutliers(PointCloud::Ptr in, double radius, unsigned min_neighbors) const {
  if (!(in->size() > 0 && radius > 0 && min_neighbors > 0)) return in;   // PointCloudSensor.cpp:214, :224
  std::vector<float> packed(in->size() * 3);
  int n_out = 0;
  if (s3d_remove_outliers(mContext, &in->points[0].x, (int)in->size(), 4, radius, min_neighbors, packed.data(), &n_out) !=
      S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  PointCloud::Ptr out(new PointCloud);
  out->points.resize(n_out);
  for (int i = 0; i < n_out; ++i) out->points[i] = makePoint(packed[3 * i], packed[3 * i + 1], packed[3 * i + 2]);
  out->width = (uint32_t)n_out;
  return out;
}

void PointCloudSensor::gather(const VertexObjectList& vertices, std::vector<std::shared_ptr<DeviceCloud>>& keep,
                              std::vector<s3d_cloud*>& clouds, std::vector<double>& poses) const {
  if (!mStorage) throw std::runtime_error("PointCloudSensor: no MeasurementStorage set (setMeasurementStorage)");
  for (const VertexObject& v : vertices) {
    Measurement::Ptr m = mStorage->get(v.measurementUuid);   // reference: mMapper->getGraph()->getMeasurement(uuid)
    PointCloudMeasurement::Ptr pcl = ptr::dynamic_pointer_cast<PointCloudMeasurement>(m);
    if (!pcl) {   // PointCloudSensor.cpp:243-247
      mLogger->message(ERROR, "Measurement in getAccumulatedCloud() is not a point cloud!");
      throw BadMeasurementType();
    }
    keep.push_back(deviceCloudOf(pcl));
    clouds.push_back(keep.back()->cloud);
    const Transform tf = v.correctedPose * pcl->getSensorPose();   // :249
    poses.insert(poses.end(), tf.data(), tf.data() + 16);
  }
}

PointCloud::Ptr PointCloudSensor::getAccumulatedCloud(const VertexObjectList& vertices) const {
  std::vector<std::shared_ptr<DeviceCloud>> keep;
  std::vector<s3d_cloud*> clouds;
  std::vector<double> poses;
  gather(vertices, keep, clouds, poses);
  s3d_cloud* accu = nullptr;
  if (s3d_cloud_accumulate(mContext, (int)clouds.size(), clouds.data(), poses.data(), nullptr, &accu) != S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  DeviceCloud guard(mContextHolder, accu);
  return download(accu);
}

Measurement::Ptr PointCloudSensor::createCombinedMeasurement(const VertexObjectList& vertices, Transform pose) const {
  std::vector<std::shared_ptr<DeviceCloud>> keep;
  std::vector<s3d_cloud*> clouds;
  std::vector<double> poses;
  gather(vertices, keep, clouds, poses);
  s3d_cloud* shifted = nullptr;   // accumulate + pose.inverse() in one pass over the points (:260-262)
  if (s3d_cloud_accumulate(mContext, (int)clouds.size(), clouds.data(), poses.data(), pose.data(), &shifted) !=
      S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  std::shared_ptr<DeviceCloud> dev = std::make_shared<DeviceCloud>(mContextHolder, shifted);
  PointCloud::Ptr cloud = download(shifted);
  mLogger->message(DEBUG, "Patch pointcloud has " + std::to_string(cloud->size()) + " points.");
  PointCloudMeasurement::Ptr m(new PointCloudMeasurement(cloud, "AccumulatedPointcloud", mName, Transform::Identity()));
  m->setDeviceCloud(dev);   // the patch stays in HBM for the createConstraint that follows (ScanSensor.cpp:150-156)
  return m;
}

PointCloud::Ptr PointCloudSensor::buildMap(const VertexObjectList& vertices) const {
  std::vector<std::shared_ptr<DeviceCloud>> keep;
  std::vector<s3d_cloud*> clouds;
  std::vector<double> poses;
  gather(vertices, keep, clouds, poses);
  s3d_cloud* map = nullptr;   // :305-309: accumulate, removeOutliers, downsample — all in HBM
  if (s3d_build_map(mContext, (int)clouds.size(), clouds.data(), poses.data(), mMapOutlierRadius, mMapOutlierNeighbors,
                    mMapResolution, &map) != S3D_STATUS_OK)
    throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  DeviceCloud guard(mContextHolder, map);
  mLogger->message(INFO, "Generated Pointcloud from " + std::to_string(vertices.size()) + " scans.");
  return download(map);
}

void PointCloudSensor::fillGroundPlane(PointCloud::Ptr cloud, ScalarType radius) {
  // PointCloudSensor.cpp:364-368 (RANSAC plane, threshold 0.01) and :370-387 (the ring points) in one call
  const int n = (int)cloud->size();
  const float* xyz = n > 0 ? &cloud->points[0].x : nullptr;
  int n_ring = 0;
  // at most (radius / res + 1) rings of (2 pi radius / res + 2) points
  const double rings = radius / mMapResolution + 2, per_ring = 2 * 3.141592654 * radius / mMapResolution + 3;   // the reference's PI (:48)
  std::vector<float> ring((size_t)(rings * per_ring) * 3 + 3);
  const int st = s3d_fill_ground_plane(mContext, xyz, n, 4, radius, mMapResolution, ring.data(), (int)(ring.size() / 3),
                                       &n_ring, nullptr);
  if (st == S3D_STATUS_TOO_FEW_POINTS) {   // the reference reads the coefficients of a failed fit (undefined)
    mLogger->message(ERROR, "Could not fit a ground plane.");
    return;
  }
  if (st != S3D_STATUS_OK) throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  for (int i = 0; i < n_ring; ++i) cloud->push_back(makePoint(ring[3 * i], ring[3 * i + 1], ring[3 * i + 2]));
}

namespace {

// PLY scalar types (name, aliases) -> size; value read as double
int plyTypeSize(const std::string& t) {
  if (t == "char" || t == "int8" || t == "uchar" || t == "uint8") return 1;
  if (t == "short" || t == "int16" || t == "ushort" || t == "uint16") return 2;
  if (t == "int" || t == "int32" || t == "uint" || t == "uint32" || t == "float" || t == "float32") return 4;
  if (t == "double" || t == "float64") return 8;
  return 0;
}

double plyDecode(const std::string& t, const unsigned char* b, bool swap) {
  unsigned char tmp[8];
  const int sz = plyTypeSize(t);
  for (int i = 0; i < sz; ++i) tmp[i] = b[swap ? sz - 1 - i : i];
  if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, tmp, 1); return v; }
  if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, tmp, 1); return v; }
  if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, tmp, 2); return v; }
  if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, tmp, 2); return v; }
  if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, tmp, 4); return v; }
  if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, tmp, 4); return v; }
  if (t == "float" || t == "float32") { float v; std::memcpy(&v, tmp, 4); return v; }
  double v; std::memcpy(&v, tmp, 8); return v;
}

struct PlyProperty { std::string name, type, count_type; bool list = false; };
struct PlyElement { std::string name; size_t count = 0; std::vector<PlyProperty> props; };

}  // namespace

int PointCloudSensor::readPLY(const std::string& path, PointCloud& cloud, Transform& sensor_pose) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return -1;
  std::string line;
  if (!std::getline(f, line) || line.compare(0, 3, "ply") != 0) return -1;
  int format = -1;   // 0 ascii, 1 binary little endian, 2 binary big endian
  std::vector<PlyElement> elements;
  bool header_done = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string word;
    ls >> word;
    if (word == "format") {
      std::string fmt_name;
      ls >> fmt_name;
      format = fmt_name == "ascii" ? 0 : fmt_name == "binary_little_endian" ? 1 : fmt_name == "binary_big_endian" ? 2 : -1;
    } else if (word == "element") {
      PlyElement e;
      ls >> e.name >> e.count;
      elements.push_back(e);
    } else if (word == "property" && !elements.empty()) {
      PlyProperty p;
      ls >> p.type;
      if (p.type == "list") { p.list = true; ls >> p.count_type >> p.type; }
      ls >> p.name;
      if (!plyTypeSize(p.type) || (p.list && !plyTypeSize(p.count_type))) return -1;
      elements.back().props.push_back(p);
    } else if (word == "end_header") {
      header_done = true;
      break;
    }
  }
  if (!header_done || format < 0) return -1;
  uint16_t probe = 1;
  const bool host_little = *reinterpret_cast<unsigned char*>(&probe) == 1;
  const bool swap = (format == 1 && !host_little) || (format == 2 && host_little);
  auto next = [&](const std::string& type, double& v) -> bool {   // one scalar of the body
    if (format == 0) return bool(f >> v);
    unsigned char b[8];
    if (!f.read(reinterpret_cast<char*>(b), plyTypeSize(type))) return false;
    v = plyDecode(type, b, swap);
    return true;
  };
  cloud.points.clear();
  float origin[3] = {0, 0, 0};
  float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // rows = the camera's x / y / z axis (PLYReader's orientation_)
  for (const PlyElement& e : elements) {
    for (size_t i = 0; i < e.count; ++i) {
      PointType p{0.f, 0.f, 0.f, 1.f};
      for (const PlyProperty& pr : e.props) {
        double v = 0;
        if (pr.list) {
          double cnt = 0;
          if (!next(pr.count_type, cnt)) return -1;
          for (int k = 0; k < (int)cnt; ++k)
            if (!next(pr.type, v)) return -1;
          continue;
        }
        if (!next(pr.type, v)) return -1;
        if (e.name == "vertex") {
          if (pr.name == "x") p.x = (float)v; else if (pr.name == "y") p.y = (float)v; else if (pr.name == "z") p.z = (float)v;
        } else if (e.name == "camera") {
          static const char* axes[3] = {"x_axis", "y_axis", "z_axis"};
          if (pr.name == "view_px") origin[0] = (float)v;
          else if (pr.name == "view_py") origin[1] = (float)v;
          else if (pr.name == "view_pz") origin[2] = (float)v;
          else
            for (int a = 0; a < 3; ++a)
              for (int c = 0; c < 3; ++c)
                if (pr.name == std::string(axes[a]) + "xyz"[c]) R[a][c] = (float)v;
        }
      }
      if (e.name == "vertex") cloud.points.push_back(p);
    }
  }
  cloud.width = (uint32_t)cloud.points.size();
  cloud.height = 1;
  cloud.is_dense = true;
  sensor_pose = Transform::Identity();   // :396-397: rotation = sensor_orientation_, translation = sensor_origin_
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) sensor_pose(r, c) = R[r][c];
    sensor_pose(r, 3) = origin[r];
  }
  return 0;
}

void PointCloudSensor::loadPLY(const std::string& path, const std::string& robot) {
  PointCloud::Ptr pcl_cloud(new PointCloud());
  Transform pc_tr;
  if (readPLY(path, *pcl_cloud, pc_tr) == 0) {   // PointCloudSensor.cpp:394
    mInitialMap.reset(new PointCloudMeasurement(pcl_cloud, robot, mName, pc_tr));
    if (mStorage) mStorage->add(mInitialMap);   // reference: graph vertex + identity PoseConstraint (:401-405)
    mLogger->message(INFO, "Successfully loaded initial map.");
  } else {
    mLogger->message(ERROR, "Could not load initial map.");
  }
}

Transform PointCloudSensor::align(const PointCloudMeasurement::Ptr& source, const PointCloudMeasurement::Ptr& target,
                                  const Transform& guess, const RegistrationParameters& config) {
  // each cloud is uploaded once per measurement and reused by every later align() it takes part in
  std::shared_ptr<DeviceCloud> s = deviceCloudOf(source), t = deviceCloudOf(target);
  Transform result;
  s3d_align_info info;
  s3d_exec_options opts;
  std::memset(&opts, 0, sizeof opts);
  opts.cache_prepass = mPrepassCache ? 1 : 0;
  opts.omp_unavailable = mOmpAvailable ? 0 : 1;
  const int st = s3d_align_clouds(mContext, s->cloud, t->cloud, guess.data(),
                                  static_cast<const s3d_reg_params*>(&config), &opts, result.data(), &info);
  switch (st) {
    case S3D_STATUS_OK: return result;
    case S3D_STATUS_TOO_FEW_POINTS:   // PointCloudSensor.cpp:135
      throw NoMatch("Too few points after filtering, you may have to decrease 'point_cloud_density'.");
    case S3D_STATUS_NOT_CONVERGED:
    case S3D_STATUS_FITNESS_EXCEEDED:  // :76 (doICP) / :110 (doNDT)
      throw NoMatch(fmt(config.registration_algorithm == NDT || config.registration_algorithm == NDT_OMP
                            ? "NDT failed with Fitness-Score %g > %g"
                            : "ICP failed with Fitness-Score %g > %g",
                        info.fitness, config.max_fitness_score));
    case S3D_STATUS_TOO_FAR_FROM_GUESS:  // :171
      throw NoMatch("ICP result is to far away from guess");
    case S3D_STATUS_UNSUPPORTED_ALGORITHM:  // (no enumerator maps to this any more; kept for older libraries)
      throw std::runtime_error("Registration algorithm not available in this build.");
    case S3D_STATUS_UNKNOWN_ALGORITHM:  // :164
      throw std::runtime_error("Unknown registration algorithm specified.");
    case S3D_STATUS_OMP_UNAVAILABLE:  // :161 (a reference built without pclomp; setOmpAvailable(false))
      throw std::runtime_error("OMP is not available, you need to rebuild SLAM3D with OMP or use another matching algorithm.");
    case S3D_STATUS_INVALID_ARGUMENT:
      // PCL accepts these and fails later ("Number of points in cloud is less than k", an empty NDT grid); the
      // back-end refuses them up front: correspondence_randomness outside 1..64 or above the filtered cloud size,
      // NDT resolution <= 0
      throw std::runtime_error("Registration parameters not supported by the MI355X back-end (correspondence_randomness "
                               "must be 1..64 and at most the filtered cloud size; NDT resolution must be positive).");
    default:
      throw std::runtime_error(std::string("HIP back-end error: ") + s3d_last_error(mContext));
  }
}

Constraint::Ptr PointCloudSensor::createConstraint(const Measurement::Ptr& source, const Measurement::Ptr& target,
                                                   const Transform& odometry, bool loop) {
  // PointCloudSensor.cpp:274
  Transform guess = source->getInverseSensorPose() * odometry * target->getSensorPose();
  PointCloudMeasurement::Ptr sourceCloud = ptr::dynamic_pointer_cast<PointCloudMeasurement>(source);
  PointCloudMeasurement::Ptr targetCloud = ptr::dynamic_pointer_cast<PointCloudMeasurement>(target);
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
