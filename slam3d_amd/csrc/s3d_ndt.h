// s3d_ndt.h — host side of the NDT registration: the scalar optimiser of
// pcl::NormalDistributionsTransform::computeTransformation (PCL 1.12 ndt.hpp) — Newton direction from the 6x6
// Hessian, More-Thuente line search [More, Thuente 1994] as PCL codes it (trialValueSelectionMT,
// updateIntervalMT, computeStepLengthMT) — around a callback that evaluates score / gradient / Hessian.  The
// callback is the device pass (s3d_ndt_derivatives_kernel); nothing here touches point data.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>

namespace s3d {
namespace ndt {

struct Eval {   // one derivative pass: parameters p (tx ty tz rx ry rz), transform T (column-major 4x4 float)
  double score;
  double g[6];
  double H[36];
};
using EvalFn = std::function<void(const float T[16], const double p[6], bool want_hessian, Eval& out)>;

inline double dot6(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }
inline double norm6(const double* a) { return std::sqrt(dot6(a, a)); }

// ndt.hpp init(): the Gaussian fitting constants of Eq. 6.8 [Magnusson 2009]
inline void gauss_constants(double outlier_ratio, double resolution, double* d1, double* d2) {
  const double c1 = 10.0 * (1.0 - outlier_ratio), c2 = outlier_ratio / std::pow(resolution, 3), d3 = -std::log(c2);
  *d1 = -std::log(c1 + c2) - d3;
  *d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / *d1);
}

// convertTransform: (Translation * AngleAxis(rx, X) * AngleAxis(ry, Y) * AngleAxis(rz, Z)).matrix(), float
inline void convert_transform(const double p[6], float T[16]) {
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
  const float Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
  const float Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}};
  const float Rz[3][3] = {{cc, -sc, 0}, {sc, cc, 0}, {0, 0, 1}};
  float t[3][3], R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[i][j] = (Rx[i][0] * Ry[0][j] + Rx[i][1] * Ry[1][j]) + Rx[i][2] * Ry[2][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = (t[i][0] * Rz[0][j] + t[i][1] * Rz[1][j]) + t[i][2] * Rz[2][j];
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[j * 4 + i] = R[i][j];
    T[12 + i] = (float)p[i];
  }
}

// Eigen Matrix3f::eulerAngles(0, 1, 2) of the guess' rotation block (column-major 4x4)
inline void euler_xyz(const float T[16], float res[3]) {
  auto R = [&](int r, int c) { return T[c * 4 + r]; };
  res[0] = std::atan2(R(1, 2), R(2, 2));
  const float c2 = std::sqrt(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
  if (res[0] > 0.f) {
    res[0] -= 3.14159265358979323846f;
    res[1] = std::atan2(-R(0, 2), -c2);
  } else {
    res[1] = std::atan2(-R(0, 2), c2);
  }
  const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
  res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

// dR[k] = dR/dangle_k, d2R[kl] (kl = 00 01 02 11 12 22) for R = Rx Ry Rz, row-major 3x3; angles below 10e-5
// are snapped to cos = 1, sin = 0 as computeAngleDerivatives does
inline void angle_derivatives(const double p[6], double dR[3][9], double d2R[6][9]) {
  double E[3][3][9];   // E[axis][derivative order]
  for (int ax = 0; ax < 3; ++ax) {
    double c, s;
    if (std::fabs(p[3 + ax]) < 10e-5) { c = 1.0; s = 0.0; }
    else { c = std::cos(p[3 + ax]); s = std::sin(p[3 + ax]); }
    const int u = (ax + 1) % 3, v = (ax + 2) % 3;
    for (int o = 0; o < 3; ++o) for (int i = 0; i < 9; ++i) E[ax][o][i] = 0.0;
    E[ax][0][ax * 3 + ax] = 1.0;
    E[ax][0][u * 3 + u] = c;  E[ax][0][u * 3 + v] = -s; E[ax][0][v * 3 + u] = s;  E[ax][0][v * 3 + v] = c;
    E[ax][1][u * 3 + u] = -s; E[ax][1][u * 3 + v] = -c; E[ax][1][v * 3 + u] = c;  E[ax][1][v * 3 + v] = -s;
    E[ax][2][u * 3 + u] = -c; E[ax][2][u * 3 + v] = s;  E[ax][2][v * 3 + u] = -s; E[ax][2][v * 3 + v] = -c;
  }
  auto mul = [](const double* a, const double* b, double* o) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
  };
  double t[9];
  for (int k = 0; k < 3; ++k) {
    int o[3] = {0, 0, 0};
    o[k] = 1;
    mul(E[0][o[0]], E[1][o[1]], t);
    mul(t, E[2][o[2]], dR[k]);
  }
  int kl = 0;
  for (int k = 0; k < 3; ++k)
    for (int l = k; l < 3; ++l, ++kl) {
      int o[3] = {0, 0, 0};
      o[k] += 1; o[l] += 1;
      mul(E[0][o[0]], E[1][o[1]], t);
      mul(t, E[2][o[2]], d2R[kl]);
    }
}

// least-squares solution of H d = b for the symmetric 6x6 H (what JacobiSVD::solve returns): cyclic Jacobi
// eigen-decomposition, eigenvalues below Eigen's default threshold dropped
inline void solve6(const double Hin[36], const double b[6], double d[6]) {
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { A[i][j] = 0.5 * (Hin[i * 6 + j] + Hin[j * 6 + i]); V[i][j] = i == j ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) (i != j ? off : diag) += A[i][j] * A[i][j];
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) { const double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s * y; A[k][q] = s * x + c * y; }
        for (int k = 0; k < 6; ++k) { const double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s * y; A[q][k] = s * x + c * y; }
        for (int k = 0; k < 6; ++k) { const double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s * y; V[k][q] = s * x + c * y; }
      }
  }
  double lmax = 0;
  for (int i = 0; i < 6; ++i) lmax = std::fmax(lmax, std::fabs(A[i][i]));
  const double thr = 6 * DBL_EPSILON * lmax;
  for (int i = 0; i < 6; ++i) d[i] = 0;
  for (int k = 0; k < 6; ++k) {
    if (!(std::fabs(A[k][k]) > thr)) continue;
    double vb = 0;
    for (int i = 0; i < 6; ++i) vb += V[i][k] * b[i];
    vb /= A[k][k];
    for (int i = 0; i < 6; ++i) d[i] += V[i][k] * vb;
  }
}

inline double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t,
                          double g_t) {
  auto cubic = [](double a0, double f0, double g0, double a1, double f1, double g1) {   // Eq. 2.4.52 / 2.4.56 [Sun, Yuan 2006]
    const double z = 3 * (f1 - f0) / (a1 - a0) - g1 - g0, w = std::sqrt(z * z - g1 * g0);
    return a0 + (a1 - a0) * (w - g0 - z) / (g1 - g0 + 2 * w);
  };
  if (f_t > f_l) {                                                       // case 1
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return std::fabs(a_c - a_l) < std::fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {                                                   // case 2
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return std::fabs(a_c - a_t) >= std::fabs(a_s - a_t) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {                                // case 3
    const double a_c = cubic(a_l, f_l, g_l, a_t, f_t, g_t);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_n = std::fabs(a_c - a_t) < std::fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? std::fmin(a_t + 0.66 * (a_u - a_t), a_n) : std::fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  return cubic(a_u, f_u, g_u, a_t, f_t, g_t);                            // case 4
}

inline bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u, double& g_u, double a_t,
                            double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) { a_u = a_l; f_u = f_l; g_u = g_l; a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  return true;
}

struct Result { float T[16]; int converged, iterations, evaluations; };

// computeTransformation + computeStepLengthMT as a resumable state machine: it advances until it needs a
// derivative pass (request()), is fed the result (feed()) and continues.  One instance per scan pair; the batched
// entry point advances many of them in lock step so that every round is ONE kernel launch over all pairs.
// The rotation epsilon is never set by slam3d (0): the PCL 1.12 stopping test reduces to the iteration cap or the
// squared translation of the step <= transformation_epsilon.
class Solver {
 public:
  Solver(const float guess[16], double step_size, double transformation_epsilon, int maximum_iterations)
      : step_max_(step_size), step_min_(transformation_epsilon / 2), eps_(transformation_epsilon),
        max_iter_(maximum_iterations) {
    std::memcpy(res_.T, guess, sizeof res_.T);
    res_.converged = 0; res_.iterations = 0; res_.evaluations = 0;
    float e[3];
    euler_xyz(guess, e);
    for (int i = 0; i < 3; ++i) { p_[i] = (double)guess[12 + i]; p_[3 + i] = (double)e[i]; }
    std::memcpy(req_T_, guess, sizeof req_T_);
    std::memcpy(req_p_, p_, sizeof req_p_);
    req_h_ = true;
    phase_ = INIT;
  }
  bool pending() const { return phase_ != DONE; }
  const float* request_T() const { return req_T_; }
  const double* request_p() const { return req_p_; }
  bool request_hessian() const { return req_h_; }
  const Result& result() const { return res_; }

  void feed(const Eval& e) {
    ++res_.evaluations;
    switch (phase_) {
      case INIT:
        ev_ = e;
        newton();
        break;
      case LS_FIRST:
        ev_ = e;
        std::memcpy(res_.T, req_T_, sizeof res_.T);
        ls_measure();
        ls_check();
        break;
      case LS_LOOP:
        ev_.score = e.score;
        std::memcpy(ev_.g, e.g, sizeof ev_.g);
        std::memcpy(res_.T, req_T_, sizeof res_.T);
        ls_measure();
        if (open_ && (psi_t_ <= 0 && d_psi_t_ >= 0)) {
          open_ = false;
          f_l_ += phi_0_ - mu_ * d_phi_0_ * a_l_;
          g_l_ += mu_ * d_phi_0_;
          f_u_ += phi_0_ - mu_ * d_phi_0_ * a_u_;
          g_u_ += mu_ * d_phi_0_;
        }
        conv_ = open_ ? update_interval(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, psi_t_, d_psi_t_)
                      : update_interval(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, phi_t_, d_phi_t_);
        ++step_iterations_;
        ls_check();
        break;
      case LS_HESS:
        std::memcpy(ev_.H, e.H, sizeof ev_.H);
        after_line_search();
        break;
      case DONE:
        break;
    }
  }

 private:
  enum Phase { INIT, LS_FIRST, LS_LOOP, LS_HESS, DONE };

  void newton() {   // Newton direction, then the first trial of computeStepLengthMT
    double mg[6];
    for (int i = 0; i < 6; ++i) mg[i] = -ev_.g[i];
    solve6(ev_.H, mg, dir_);
    const double dn = norm6(dir_);
    if (dn == 0 || dn != dn) { res_.converged = dn == 0; phase_ = DONE; return; }
    for (int i = 0; i < 6; ++i) dir_[i] /= dn;
    phi_0_ = -ev_.score;
    d_phi_0_ = -dot6(ev_.g, dir_);
    if (d_phi_0_ >= 0) {
      if (d_phi_0_ == 0) { a_t_ = 0; after_line_search(); return; }
      d_phi_0_ = -d_phi_0_;
      for (int i = 0; i < 6; ++i) dir_[i] = -dir_[i];
    }
    step_iterations_ = 0;
    a_l_ = 0; a_u_ = 0; f_l_ = 0; g_l_ = d_phi_0_ - mu_ * d_phi_0_; f_u_ = 0; g_u_ = g_l_;
    conv_ = (step_max_ - step_min_) < 0;
    open_ = true;
    a_t_ = std::fmax(std::fmin(dn, step_max_), step_min_);
    ask(true, LS_FIRST);
  }
  void ask(bool hessian, Phase next) {
    for (int i = 0; i < 6; ++i) req_p_[i] = p_[i] + dir_[i] * a_t_;
    convert_transform(req_p_, req_T_);
    req_h_ = hessian;
    phase_ = next;
  }
  void ls_measure() {
    phi_t_ = -ev_.score;
    d_phi_t_ = -dot6(ev_.g, dir_);
    psi_t_ = phi_t_ - phi_0_ - mu_ * d_phi_0_ * a_t_;
    d_psi_t_ = d_phi_t_ - mu_ * d_phi_0_;
  }
  void ls_check() {
    if (!conv_ && step_iterations_ < 10 && !(psi_t_ <= 0 && d_phi_t_ <= -nu_ * d_phi_0_)) {
      a_t_ = open_ ? trial_value(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, psi_t_, d_psi_t_)
                   : trial_value(a_l_, f_l_, g_l_, a_u_, f_u_, g_u_, a_t_, phi_t_, d_phi_t_);
      a_t_ = std::fmax(std::fmin(a_t_, step_max_), step_min_);
      ask(false, LS_LOOP);
    } else if (step_iterations_) {
      req_h_ = true;          // computeHessian at the accepted point (same T / p as the last trial)
      phase_ = LS_HESS;
    } else {
      after_line_search();
    }
  }
  void after_line_search() {
    double delta[6];
    for (int i = 0; i < 6; ++i) { delta[i] = dir_[i] * a_t_; p_[i] += delta[i]; }
    float Tstep[16];
    convert_transform(delta, Tstep);
    const double tsq = (double)Tstep[12] * Tstep[12] + (double)Tstep[13] * Tstep[13] + (double)Tstep[14] * Tstep[14];
    ++res_.iterations;
    if (res_.iterations >= max_iter_ || (eps_ > 0 && tsq <= eps_)) { res_.converged = 1; phase_ = DONE; return; }
    newton();
  }

  static constexpr double mu_ = 1.e-4, nu_ = 0.9;
  double step_max_, step_min_, eps_;
  int max_iter_;
  Result res_;
  Phase phase_;
  Eval ev_;
  double p_[6], dir_[6];
  float req_T_[16];
  double req_p_[6];
  bool req_h_;
  // line-search state
  double phi_0_ = 0, d_phi_0_ = 0, a_t_ = 0, a_l_ = 0, a_u_ = 0, f_l_ = 0, g_l_ = 0, f_u_ = 0, g_u_ = 0;
  double phi_t_ = 0, d_phi_t_ = 0, psi_t_ = 0, d_psi_t_ = 0;
  bool conv_ = false, open_ = true;
  int step_iterations_ = 0;
};

// the same thing driven by a callback (one pair)
inline Result run(const EvalFn& eval, const float guess[16], double step_size, double transformation_epsilon,
                  int maximum_iterations) {
  Solver s(guess, step_size, transformation_epsilon, maximum_iterations);
  while (s.pending()) {
    Eval e;
    eval(s.request_T(), s.request_p(), s.request_hessian(), e);
    s.feed(e);
  }
  return s.result();
}

}  // namespace ndt
}  // namespace s3d
