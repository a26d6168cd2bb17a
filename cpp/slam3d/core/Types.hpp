// slam3d/core/Types.hpp (MI355X build) — the boundary value types of the registration path.
//
// Mirrors the names and members the hot path and its callers use from the reference's
// slam3d/core/Types.hpp:46-187 and slam3d/core/Sensor.hpp:44-72, so that code written against the reference
// (createConstraint callers, tests) reads the same.  The reference builds these on Eigen and Boost
// (Types.hpp:30-39): where those headers are installed they are used - Transform, Position and Covariance<N>
// are then the reference's own Eigen typedefs (Types.hpp:49-54) and the smart pointers are boost::shared_ptr
// (Types.hpp:30) - and only where they are missing (this build environment has neither) the dependency-free
// stand-ins below take their place, with the same spelling.  -DS3D_MIRROR_NO_EIGEN / -DS3D_MIRROR_NO_BOOST force
// the stand-ins.  On a machine that has the whole slam3d core, use its headers and the binding of INTEGRATION.md.
#pragma once

#include <cmath>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Geometry>) && !defined(S3D_MIRROR_NO_EIGEN)
#define S3D_MIRROR_HAVE_EIGEN 1
#endif
#if __has_include(<boost/shared_ptr.hpp>) && __has_include(<boost/make_shared.hpp>) && !defined(S3D_MIRROR_NO_BOOST)
#define S3D_MIRROR_HAVE_BOOST 1
#endif
#endif

#if defined(S3D_MIRROR_HAVE_BOOST)
#include <boost/make_shared.hpp>
#include <boost/pointer_cast.hpp>
#include <boost/shared_ptr.hpp>
#include <boost/weak_ptr.hpp>
namespace slam3d { namespace ptr { using boost::shared_ptr; using boost::weak_ptr; using boost::make_shared; using boost::dynamic_pointer_cast; } }
#else
namespace slam3d { namespace ptr { using std::shared_ptr; using std::weak_ptr; using std::make_shared; using std::dynamic_pointer_cast; } }
#endif

#if defined(S3D_MIRROR_HAVE_EIGEN)
#include <Eigen/Geometry>
#endif

namespace slam3d {

typedef unsigned IdType;
typedef double ScalarType;

#if defined(S3D_MIRROR_HAVE_EIGEN)
// slam3d/core/Types.hpp:49-54, verbatim semantics: the mirror's code only uses operator(), data(), inverse(),
// operator*, Identity() and translation(), which Eigen provides with the same meaning (column-major 4x4 double)
typedef Eigen::Matrix<ScalarType, 3, 1> Position;
typedef Eigen::Transform<ScalarType, 3, Eigen::Isometry> Transform;
template <unsigned N> using Covariance = Eigen::Matrix<ScalarType, N, N>;
#else
struct Position {  // Eigen::Matrix<double,3,1>
  double v[3] = {0, 0, 0};
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};

// Types.hpp:53  typedef Eigen::Transform<ScalarType,3,Eigen::Isometry> Transform;
// 4x4 double, column-major storage like Eigen.
class Transform {
 public:
  Transform() { setIdentity(); }
  static Transform Identity() { return Transform(); }
  void setIdentity() { for (int i = 0; i < 16; ++i) m_[i] = (i % 5 == 0) ? 1.0 : 0.0; }
  double& operator()(int r, int c) { return m_[c * 4 + r]; }
  double operator()(int r, int c) const { return m_[c * 4 + r]; }
  double* data() { return m_; }             // == matrix().data()
  const double* data() const { return m_; }
  Position translation() const { Position p; p[0] = (*this)(0, 3); p[1] = (*this)(1, 3); p[2] = (*this)(2, 3); return p; }
  Transform inverse() const {               // Isometry: R^T, -R^T t
    Transform o;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) o(r, c) = (*this)(c, r);
    for (int r = 0; r < 3; ++r) o(r, 3) = -(o(r, 0) * (*this)(0, 3) + o(r, 1) * (*this)(1, 3) + o(r, 2) * (*this)(2, 3));
    return o;
  }
  Transform operator*(const Transform& b) const {
    Transform o;
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 4; ++r) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += (*this)(r, k) * b(k, c);
        o(r, c) = s;
      }
    return o;
  }
 private:
  double m_[16];
};

template <unsigned N>
struct Covariance {  // Eigen::Matrix<double,N,N>, row/col symmetric use only
  double v[N * N];
  static Covariance Identity() { Covariance c; for (unsigned i = 0; i < N * N; ++i) c.v[i] = (i % (N + 1) == 0) ? 1.0 : 0.0; return c; }
  double& operator()(unsigned r, unsigned c) { return v[r * N + c]; }
  double operator()(unsigned r, unsigned c) const { return v[r * N + c]; }
};

#endif  // S3D_MIRROR_HAVE_EIGEN

// Types.hpp:108-135
typedef std::string Uuid;  // reference: boost::uuids::uuid (random); here a process-unique string
Uuid generateUuid();        // defined in sensor/pcl/PointCloudSensor.cpp

class Measurement {
 public:
  typedef ptr::shared_ptr<Measurement> Ptr;  // boost::shared_ptr where Boost is installed (Types.hpp:30)
  // id: the identifier of a reloaded measurement (Types.hpp:114-115; nil = draw a new one)
  Measurement(const std::string& r, const std::string& s, const Transform& p, const Uuid& id = Uuid())
      : mRobotName(r), mSensorName(s), mSensorPose(p), mInverseSensorPose(p.inverse()),
        mUniqueId(id.empty() ? generateUuid() : id) {}
  virtual ~Measurement() {}
  Uuid getUniqueId() const { return mUniqueId; }
  std::string getRobotName() const { return mRobotName; }
  std::string getSensorName() const { return mSensorName; }
  Transform getSensorPose() const { return mSensorPose; }
  Transform getInverseSensorPose() const { return mInverseSensorPose; }
  virtual const char* getTypeName() const = 0;
 protected:
  std::string mRobotName, mSensorName;
  Transform mSensorPose, mInverseSensorPose;
  Uuid mUniqueId;
};

// slam3d/core/Types.hpp:305-330: what the pose graph attaches to a vertex
struct VertexObject {
  void init(const Measurement::Ptr m, IdType i) {
    index = i;
    robotName = m->getRobotName();
    sensorName = m->getSensorName();
    typeName = m->getTypeName();
    measurementUuid = m->getUniqueId();
    label = robotName + ":" + sensorName + "(" + std::to_string(index) + ")";
  }
  IdType index = 0;
  std::string label, robotName, sensorName, typeName;
  Transform correctedPose;
  Uuid measurementUuid;
};
typedef std::vector<VertexObject> VertexObjectList;

// slam3d/core/MeasurementStorage.hpp:16-53
class MeasurementStorage {
 public:
  virtual ~MeasurementStorage() {}
  virtual void add(Measurement::Ptr m) { mMeasurements[m->getUniqueId()] = m; }
  virtual Measurement::Ptr get(const Uuid& key) { return mMeasurements.at(key); }   // std::out_of_range if absent
  virtual bool contains(const Uuid& key) { return mMeasurements.count(key) != 0; }
 private:
  std::map<Uuid, Measurement::Ptr> mMeasurements;
};

enum ConstraintType { TENTATIVE, SE3, GRAVITY, POSITION, ORIENTATION, POSE };

// Types.hpp:137-187
class Constraint {
 public:
  typedef ptr::shared_ptr<Constraint> Ptr;
  explicit Constraint(const std::string& s) : mSensorName(s) {}
  virtual ~Constraint() {}
  virtual ConstraintType getType() = 0;
  virtual const char* getTypeName() = 0;
  const std::string& getSensorName() const { return mSensorName; }
 protected:
  std::string mSensorName;
};

class SE3Constraint : public Constraint {
 public:
  typedef ptr::shared_ptr<SE3Constraint> Ptr;
  SE3Constraint(const std::string& s, const Transform& t, const Covariance<6>& i) : Constraint(s), mRelativePose(t), mInformation(i) {}
  ConstraintType getType() override { return SE3; }
  const char* getTypeName() override { return "SE(3)"; }
  const Transform& getRelativePose() const { return mRelativePose; }
  const Covariance<6>& getInformation() const { return mInformation; }
 protected:
  Transform mRelativePose;
  Covariance<6> mInformation;
};

// Sensor.hpp:44-72
class BadMeasurementType : public std::exception {
 public:
  const char* what() const throw() override { return "Measurement type does not match sensor type!"; }
};
class NoMatch : public std::exception {
 public:
  explicit NoMatch(const std::string& msg) : message(msg) {}
  const char* what() const throw() override { return message.c_str(); }
  std::string message;
};

// Logger.hpp:47-107 (levels and message()); the default implementation writes to stderr
enum LogLevel { DEBUG = 0, INFO = 1, WARNING = 2, ERROR = 3, FATAL = 4 };
class Logger {
 public:
  virtual ~Logger() {}
  void setLogLevel(LogLevel lvl) { mLogLevel = lvl; }
  virtual void message(LogLevel lvl, const std::string& msg);
 protected:
  LogLevel mLogLevel = INFO;
};

}  // namespace slam3d
