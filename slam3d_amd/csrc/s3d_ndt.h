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

// computeStepLengthMT: on return `ev` / `T` describe the accepted trial point
inline double step_length(const EvalFn& eval, const double x[6], double dir[6], double step_init, double step_max,
                          double step_min, Eval& ev, float T[16]) {
  const double phi_0 = -ev.score;
  double d_phi_0 = -dot6(ev.g, dir);
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 = -d_phi_0;
    for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0, f_l = 0, g_l = d_phi_0 - mu * d_phi_0, f_u = 0, g_u = g_l;
  bool interval_converged = (step_max - step_min) < 0, open_interval = true;
  double a_t = std::fmax(std::fmin(step_init, step_max), step_min);
  double x_t[6];
  for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
  convert_transform(x_t, T);
  eval(T, x_t, true, ev);
  double phi_t = -ev.score, d_phi_t = -dot6(ev.g, dir);
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                        : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = std::fmax(std::fmin(a_t, step_max), step_min);
    for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
    convert_transform(x_t, T);
    Eval trial;
    eval(T, x_t, false, trial);
    ev.score = trial.score;
    std::memcpy(ev.g, trial.g, sizeof ev.g);
    phi_t = -ev.score;
    d_phi_t = -dot6(ev.g, dir);
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
    d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = false;
      f_l += phi_0 - mu * d_phi_0 * a_l;
      g_l += mu * d_phi_0;
      f_u += phi_0 - mu * d_phi_0 * a_u;
      g_u += mu * d_phi_0;
    }
    interval_converged = open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                                       : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    ++step_iterations;
  }
  if (step_iterations) {   // computeHessian at the accepted point
    Eval at;
    eval(T, x_t, true, at);
    std::memcpy(ev.H, at.H, sizeof ev.H);
  }
  return a_t;
}

struct Result { float T[16]; int converged, iterations, evaluations; };

// computeTransformation.  rotation epsilon is never set by slam3d (0): the PCL 1.12 stopping test reduces to the
// iteration cap or the squared translation of the step <= transformation_epsilon.
inline Result run(const EvalFn& eval_in, const float guess[16], double step_size, double transformation_epsilon,
                  int maximum_iterations) {
  Result R;
  std::memcpy(R.T, guess, sizeof R.T);
  R.converged = 0; R.iterations = 0; R.evaluations = 0;
  EvalFn eval = [&](const float T[16], const double p[6], bool h, Eval& o) { ++R.evaluations; eval_in(T, p, h, o); };
  double p[6];
  {
    float e[3];
    euler_xyz(guess, e);
    for (int i = 0; i < 3; ++i) { p[i] = (double)guess[12 + i]; p[3 + i] = (double)e[i]; }
  }
  Eval ev;
  eval(R.T, p, true, ev);
  while (!R.converged) {
    double mg[6], delta[6];
    for (int i = 0; i < 6; ++i) mg[i] = -ev.g[i];
    solve6(ev.H, mg, delta);
    double dn = norm6(delta);
    if (dn == 0 || dn != dn) { R.converged = dn == 0; break; }
    for (int i = 0; i < 6; ++i) delta[i] /= dn;
    dn = step_length(eval, p, delta, dn, step_size, transformation_epsilon / 2, ev, R.T);
    for (int i = 0; i < 6; ++i) { delta[i] *= dn; p[i] += delta[i]; }
    float Tstep[16];
    convert_transform(delta, Tstep);
    const double tsq = (double)Tstep[12] * Tstep[12] + (double)Tstep[13] * Tstep[13] + (double)Tstep[14] * Tstep[14];
    ++R.iterations;
    if (R.iterations >= maximum_iterations || (transformation_epsilon > 0 && tsq <= transformation_epsilon)) R.converged = 1;
  }
  return R;
}

}  // namespace ndt
}  // namespace s3d
