/* s3d_oracle.c — CPU restatement (plain C99, no third-party deps) of the
 * registration hot path of dfki-ric/slam3d.  See s3d_oracle.h for the status
 * of this file (test infrastructure, parity unpinned).
 *
 * Compile with -ffp-contract=off: float expressions are written in the order
 * the reference stack evaluates them and must not be fused.
 *
 * Citations "PCS.cpp:N" = /root/reference/slam3d/sensor/pcl/PointCloudSensor.cpp:N.
 * "PCL <file>" = PCL 1.12.1 source file restated from its published algorithm
 * (PCL is an un-vendored dependency: slam3d-dependencies.cmake:22).
 */
#include "s3d_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------ utils */

void s3o_default_params(s3d_reg_params* p) {
  /* RegistrationParameters.hpp:36-97 in-class defaults */
  p->registration_algorithm = S3D_ALG_GICP;
  p->point_cloud_density = 0.2;
  p->max_fitness_score = 2.0;
  p->max_translation = 1.0;
  p->max_rotation = 1.0;
  p->euclidean_fitness_epsilon = 1.0;
  p->transformation_epsilon = 1e-5;
  p->max_correspondence_distance = 2.5;
  p->maximum_iterations = 50;
  p->rotation_epsilon = 2e-3;
  p->correspondence_randomness = 20;
  p->maximum_optimizer_iterations = 20;
  p->resolution = 1.0f;
  p->step_size = 0.05;
  p->outlier_ratio = 0.35;
}

/* column-major 4x4 helpers: element (r,c) = m[c*4+r] */
#define M4(m, r, c) ((m)[(c) * 4 + (r)])

void s3o_mat4d_mul(const double a[16], const double b[16], double out[16]) {
  double t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += M4(a, r, k) * M4(b, k, c);
      t[c * 4 + r] = s;
    }
  memcpy(out, t, sizeof t);
}

/* Eigen::Transform<double,3,Isometry>::inverse(): R^T, -R^T t */
void s3o_mat4d_inverse_isometry(const double a[16], double out[16]) {
  double t[16] = {0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M4(t, r, c) = M4(a, c, r);
  for (int r = 0; r < 3; ++r) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += M4(t, r, k) * M4(a, k, 3);
    M4(t, r, 3) = -s;
  }
  M4(t, 3, 3) = 1.0;
  memcpy(out, t, sizeof t);
}

/* Eigen::AngleAxisd(R).angle(): via quaternion, angle = 2*atan2(|vec|, |w|) in [0,pi] */
double s3o_rotation_angle(const double m[16]) {
  /* rotation matrix -> quaternion (Eigen's Shepperd branch selection) */
  double m00 = M4(m, 0, 0), m11 = M4(m, 1, 1), m22 = M4(m, 2, 2);
  double t = m00 + m11 + m22, w, x, y, z;
  if (t > 0) {
    t = sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    x = (M4(m, 2, 1) - M4(m, 1, 2)) * t;
    y = (M4(m, 0, 2) - M4(m, 2, 0)) * t;
    z = (M4(m, 1, 0) - M4(m, 0, 1)) * t;
  } else {
    int i = 0;
    if (m11 > m00) i = 1;
    if (m22 > M4(m, i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double q[3];
    t = sqrt(M4(m, i, i) - M4(m, j, j) - M4(m, k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (M4(m, k, j) - M4(m, j, k)) * t;
    q[j] = (M4(m, j, i) + M4(m, i, j)) * t;
    q[k] = (M4(m, k, i) + M4(m, i, k)) * t;
    x = q[0]; y = q[1]; z = q[2];
  }
  double n = sqrt(x * x + y * y + z * z);
  return 2.0 * atan2(n, fabs(w));
}

/* cyclic Jacobi for a symmetric 3x3 (row-major a); eigenvalues descending,
 * eigenvectors in the COLUMNS of v (row-major 3x3).  Stands in for
 * Eigen::JacobiSVD<Matrix3d>(cov, ComputeFullU) on a symmetric PSD matrix
 * (PCL gicp.hpp computeCovariances): singular values == eigenvalues, U == V. */
void s3o_sym_eig3(const double a_in[9], double ev[3], double v[9]) {
  double a[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = a_in[i * 3 + j];
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A J */
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- J^T A */
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (d[order[j]] > d[order[i]]) { int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int k = 0; k < 3; ++k) {
    ev[k] = d[order[k]];
    for (int r = 0; r < 3; ++r) v[r * 3 + k] = V[r][order[k]];
  }
}

/* Eigen 3x3 inverse (compute_inverse_size3_helper): cofactors / determinant */
static void mat3_inverse(const double m[3][3], double inv[3][3]) {
  double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  double c10 = m[1][2] * m[2][0] - m[1][0] * m[2][2]; /* cofactor(1,0) of transpose layout */
  double c20 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  double det = m[0][0] * c00 + m[0][1] * c10 + m[0][2] * c20;
  double id = 1.0 / det;
  inv[0][0] = c00 * id;
  inv[1][0] = c10 * id;
  inv[2][0] = c20 * id;
  inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
  inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
  inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
  inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

/* pcl::transformPointCloud(cloud, cloud, Matrix4f): pcl::detail::Transformer::se3
 * (PCL common/impl/transforms.hpp): p0 + (p1 + (p2 + c3)) */
static inline void xf_pcl(const float m[16], const float p[3], float o[3]) {
  for (int r = 0; r < 3; ++r) {
    float p0 = M4(m, r, 0) * p[0], p1 = M4(m, r, 1) * p[1], p2 = M4(m, r, 2) * p[2];
    o[r] = p0 + (p1 + (p2 + M4(m, r, 3)));
  }
}
/* Eigen Matrix4f * Vector4f (w = 1): ((c0*x + c1*y) + c2*z) + c3*w */
static inline void xf_eigen(const float m[16], const float p[3], float o[3]) {
  for (int r = 0; r < 3; ++r)
    o[r] = ((M4(m, r, 0) * p[0] + M4(m, r, 1) * p[1]) + M4(m, r, 2) * p[2]) + M4(m, r, 3);
}
/* Eigen Matrix4f * Matrix4f, coefficient order k = 0..3 */
static void mat4f_mul(const float a[16], const float b[16], float out[16]) {
  float t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      t[c * 4 + r] = ((M4(a, r, 0) * M4(b, 0, c) + M4(a, r, 1) * M4(b, 1, c)) + M4(a, r, 2) * M4(b, 2, c)) +
                     M4(a, r, 3) * M4(b, 3, c);
  memcpy(out, t, sizeof t);
}

/* ------------------------------------------------------------------ voxel grid (A3) */

typedef struct { unsigned key; int idx; } key_idx;
static int cmp_key_idx(const void* a, const void* b) {
  const key_idx *x = (const key_idx*)a, *y = (const key_idx*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int s3o_voxel_downsample(const float* xyz, int n, int stride, double leaf_size, float* out,
                         s3o_voxel_info* info) {
  s3o_voxel_info li;
  memset(&li, 0, sizeof li);
  if (n <= 0) { /* PCS.cpp:193: empty in -> empty out */
    if (info) *info = li;
    return 0;
  }
  /* PCS.cpp:196 setLeafSize(double->float); PCL voxel_grid.h: inverse = 1/leaf (float) */
  const float leaf = (float)leaf_size;
  const float inv = 1.0f / leaf;
  /* PCL common getMinMax3D */
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int n_finite = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * stride;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    ++n_finite;
    for (int a = 0; a < 3; ++a) {
      if (p[a] < mn[a]) mn[a] = p[a];
      if (p[a] > mx[a]) mx[a] = p[a];
    }
  }
  if (n_finite == 0) {
    if (info) *info = li;
    return 0;
  }
  for (int a = 0; a < 3; ++a) { li.min_p[a] = mn[a]; li.max_p[a] = mx[a]; }
  /* PCL voxel_grid.hpp applyFilter: overflow check */
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1;
  int64_t dy = (int64_t)((mx[1] - mn[1]) * inv) + 1;
  int64_t dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT_MAX) {
    li.passthrough = 1; /* "Leaf size is too small": output = input */
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) out[(size_t)i * 3 + a] = xyz[(size_t)i * stride + a];
    if (info) *info = li;
    return n;
  }
  for (int a = 0; a < 3; ++a) {
    li.min_b[a] = (int)floorf(mn[a] * inv);
    li.max_b[a] = (int)floorf(mx[a] * inv);
    li.div_b[a] = li.max_b[a] - li.min_b[a] + 1;
  }
  const int mul[3] = {1, li.div_b[0], li.div_b[0] * li.div_b[1]};
  key_idx* kv = (key_idx*)malloc(sizeof(key_idx) * (size_t)n);
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * stride;
    if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
    int ijk0 = (int)(floorf(p[0] * inv) - (float)li.min_b[0]);
    int ijk1 = (int)(floorf(p[1] * inv) - (float)li.min_b[1]);
    int ijk2 = (int)(floorf(p[2] * inv) - (float)li.min_b[2]);
    kv[cnt].key = (unsigned)(ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2]);
    kv[cnt].idx = i;
    ++cnt;
  }
  /* PCL sorts by voxel index only (order inside a voxel unspecified); the
   * restatement fixes it to ascending point index. */
  qsort(kv, (size_t)cnt, sizeof(key_idx), cmp_key_idx);
  int total = 0;
  for (int i = 0; i < cnt;) {
    int j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f; /* AccumulatorXYZ: Eigen::Vector3f sum */
    while (j < cnt && kv[j].key == kv[i].key) {
      const float* p = xyz + (size_t)kv[j].idx * stride;
      sx += p[0]; sy += p[1]; sz += p[2];
      ++j;
    }
    const float c = (float)(j - i);
    out[(size_t)total * 3 + 0] = sx / c;
    out[(size_t)total * 3 + 1] = sy / c;
    out[(size_t)total * 3 + 2] = sz / c;
    ++total;
    i = j;
  }
  free(kv);
  if (info) *info = li;
  return total;
}

/* ------------------------------------------------------------------ kd-tree (A7) */

typedef struct {
  int   left, right;   /* children, -1 for leaf */
  int   lo, hi;        /* leaf: range in perm[] */
  int   dim;
  float divlow, divhigh;
} kd_node;

struct s3o_kdtree {
  const float* pts; /* packed xyz, not owned */
  int      n;
  int*     perm;
  kd_node* nodes;
  int      n_nodes, cap_nodes;
  float    bbmin[3], bbmax[3];
};

#define KD_LEAF 15 /* FLANN KDTreeSingleIndexParams default leaf_max_size (PCL kdtree_flann.hpp) */

static int kd_new_node(s3o_kdtree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}

static void kd_select(s3o_kdtree* t, int lo, int hi, int k, int dim) {
  /* quickselect on perm[lo..hi) by coordinate dim (ties by index for determinism) */
  int* p = t->perm;
  const float* x = t->pts;
  while (hi - lo > 1) {
    int mid = lo + (hi - lo) / 2;
    float pv = x[(size_t)p[mid] * 3 + dim];
    int pi = p[mid];
    int i = lo, j = hi - 1;
    while (i <= j) {
      while (x[(size_t)p[i] * 3 + dim] < pv || (x[(size_t)p[i] * 3 + dim] == pv && p[i] < pi)) ++i;
      while (x[(size_t)p[j] * 3 + dim] > pv || (x[(size_t)p[j] * 3 + dim] == pv && p[j] > pi)) --j;
      if (i <= j) { int tmp = p[i]; p[i] = p[j]; p[j] = tmp; ++i; --j; }
    }
    if (k <= j) hi = j + 1;
    else if (k >= i) lo = i;
    else return;
  }
}

static int kd_build_rec(s3o_kdtree* t, int lo, int hi) {
  int id = kd_new_node(t);
  kd_node nd;
  nd.left = nd.right = -1; nd.lo = lo; nd.hi = hi; nd.dim = 0; nd.divlow = nd.divhigh = 0;
  if (hi - lo <= KD_LEAF) { t->nodes[id] = nd; return id; }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = lo; i < hi; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = t->pts[(size_t)t->perm[i] * 3 + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int dim = 0;
  if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
  if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
  int mid = lo + (hi - lo) / 2;
  kd_select(t, lo, hi, mid, dim);
  float dl = -FLT_MAX, dh = FLT_MAX;
  for (int i = lo; i < mid; ++i) { float v = t->pts[(size_t)t->perm[i] * 3 + dim]; if (v > dl) dl = v; }
  dh = FLT_MAX;
  for (int i = mid; i < hi; ++i) { float v = t->pts[(size_t)t->perm[i] * 3 + dim]; if (v < dh) dh = v; }
  nd.dim = dim; nd.divlow = dl; nd.divhigh = dh;
  t->nodes[id] = nd;
  int l = kd_build_rec(t, lo, mid);
  int r = kd_build_rec(t, mid, hi);
  t->nodes[id].left = l;
  t->nodes[id].right = r;
  return id;
}

s3o_kdtree* s3o_kdtree_build(const float* xyz, int n) {
  s3o_kdtree* t = (s3o_kdtree*)calloc(1, sizeof *t);
  t->pts = xyz; t->n = n;
  t->perm = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) t->perm[i] = i;
  for (int a = 0; a < 3; ++a) { t->bbmin[a] = FLT_MAX; t->bbmax[a] = -FLT_MAX; }
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      float v = xyz[(size_t)i * 3 + a];
      if (v < t->bbmin[a]) t->bbmin[a] = v;
      if (v > t->bbmax[a]) t->bbmax[a] = v;
    }
  if (n > 0) kd_build_rec(t, 0, n);
  return t;
}
void s3o_kdtree_free(s3o_kdtree* t) {
  if (!t) return;
  free(t->perm); free(t->nodes); free(t);
}

/* FLANN L2_Simple: result += diff*diff over dims, float */
static inline float dist2f(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (dx * dx + dy * dy) + dz * dz;
}

typedef struct {
  const s3o_kdtree* t;
  const float* q;
  int k, cnt;
  int* idx;   /* sorted ascending by (d2, idx) */
  float* d2;
} kd_query;

static inline int lex_less(float d2a, int ia, float d2b, int ib) {
  return d2a < d2b || (d2a == d2b && ia < ib);
}
static void kq_insert(kd_query* s, float d2, int i) {
  if (s->cnt == s->k && !lex_less(d2, i, s->d2[s->k - 1], s->idx[s->k - 1])) return;
  int pos = s->cnt < s->k ? s->cnt++ : s->k - 1;
  while (pos > 0 && lex_less(d2, i, s->d2[pos - 1], s->idx[pos - 1])) {
    s->d2[pos] = s->d2[pos - 1]; s->idx[pos] = s->idx[pos - 1]; --pos;
  }
  s->d2[pos] = d2; s->idx[pos] = i;
}
static void kd_search_rec(kd_query* s, int id, double mind2, double dists[3]) {
  const kd_node* nd = &s->t->nodes[id];
  if (nd->left < 0) {
    for (int i = nd->lo; i < nd->hi; ++i) {
      int pi = s->t->perm[i];
      kq_insert(s, dist2f(s->q, s->t->pts + (size_t)pi * 3), pi);
    }
    return;
  }
  int dim = nd->dim;
  double val = s->q[dim];
  double diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
  int best, other; double cut;
  if (diff1 + diff2 < 0) { best = nd->left; other = nd->right; cut = diff2 * diff2; }
  else { best = nd->right; other = nd->left; cut = diff1 * diff1; }
  kd_search_rec(s, best, mind2, dists);
  double dst = dists[dim];
  double m2 = mind2 + cut - dst;
  dists[dim] = cut;
  /* exact search (eps = 0); small relative margin so that float rounding of the
   * point distances can never prune a lexicographically better candidate */
  if (s->cnt < s->k || m2 * (1.0 - 1e-6) <= (double)s->d2[s->k - 1]) kd_search_rec(s, other, m2, dists);
  dists[dim] = dst;
}
int s3o_kdtree_knn(const s3o_kdtree* t, const float q[3], int k, int* idx, float* d2) {
  if (t->n == 0 || k <= 0) return 0;
  kd_query s = {t, q, k < t->n ? k : t->n, 0, idx, d2};
  double dists[3] = {0, 0, 0}, m = 0;
  for (int a = 0; a < 3; ++a) {
    if (q[a] < t->bbmin[a]) { double d = (double)q[a] - t->bbmin[a]; dists[a] = d * d; }
    if (q[a] > t->bbmax[a]) { double d = (double)q[a] - t->bbmax[a]; dists[a] = d * d; }
    m += dists[a];
  }
  kd_search_rec(&s, 0, m, dists);
  return s.cnt;
}
void s3o_kdtree_nn1(const s3o_kdtree* t, const float q[3], int* idx, float* d2) {
  *idx = -1; *d2 = FLT_MAX;
  s3o_kdtree_knn(t, q, 1, idx, d2);
}
void s3o_nn_search(const float* tgt, int n, const float* qry, int m, int* idx, float* d2) {
  s3o_kdtree* t = s3o_kdtree_build(tgt, n);
  for (int i = 0; i < m; ++i) s3o_kdtree_nn1(t, qry + (size_t)i * 3, &idx[i], &d2[i]);
  s3o_kdtree_free(t);
}
void s3o_nn_search_brute(const float* tgt, int n, const float* qry, int m, int* idx, float* d2) {
  for (int i = 0; i < m; ++i) {
    int bi = -1; float bd = FLT_MAX;
    for (int j = 0; j < n; ++j) {
      float d = dist2f(qry + (size_t)i * 3, tgt + (size_t)j * 3);
      if (d < bd) { bd = d; bi = j; }
    }
    idx[i] = bi; d2[i] = bd;
  }
}

/* ------------------------------------------------------------------ covariances (A6) */

/* diagnostic: build every covariance as I - (1 - eps) n n^T from the FLOAT-rounded unit normal, which is how the device
 * stores it (12 bytes per point instead of 72); default off = PCL's U diag(1, 1, eps) U^T in double */
static int g_float_normals = 0;
void s3o_set_debug_float_normals(int on) { g_float_normals = on; }

static int gicp_covariances_tree(const s3o_kdtree* tree, const float* xyz, int n, int k, double eps,
                                 double* cov, double* normals) {
  if (k > n) return -1; /* PCL gicp.hpp: "Number of points in cloud is less than k_correspondences_" */
  int* nn_i = (int*)malloc(sizeof(int) * (size_t)k);
  float* nn_d = (float*)malloc(sizeof(float) * (size_t)k);
  for (int i = 0; i < n; ++i) {
    double c[3][3] = {{0}}, mean[3] = {0, 0, 0};
    s3o_kdtree_knn(tree, xyz + (size_t)i * 3, k, nn_i, nn_d);
    for (int j = 0; j < k; ++j) {
      const float* pt = xyz + (size_t)nn_i[j] * 3;
      mean[0] += pt[0]; mean[1] += pt[1]; mean[2] += pt[2];
      /* PCL: cov(0,0) += pt.x * pt.x;  — float product, double accumulation */
      c[0][0] += pt[0] * pt[0];
      c[1][0] += pt[1] * pt[0];
      c[1][1] += pt[1] * pt[1];
      c[2][0] += pt[2] * pt[0];
      c[2][1] += pt[2] * pt[1];
      c[2][2] += pt[2] * pt[2];
    }
    for (int a = 0; a < 3; ++a) mean[a] /= (double)k;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b <= a; ++b) {
        c[a][b] /= (double)k;
        c[a][b] -= mean[a] * mean[b];
        c[b][a] = c[a][b];
      }
    double ev[3], U[9];
    s3o_sym_eig3((const double*)c, ev, U);
    /* cov = sum_k v_k u_k u_k^T, v = (1, 1, gicp_epsilon) */
    double* out = cov + (size_t)i * 9;
    for (int a = 0; a < 9; ++a) out[a] = 0;
    for (int kk = 0; kk < 3; ++kk) {
      double v = kk == 2 ? eps : 1.0;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) out[a * 3 + b] += v * U[a * 3 + kk] * U[b * 3 + kk];
    }
    if (g_float_normals) { /* diagnostic (s3o_set_debug_float_normals): the covariance as the device stores it */
      const double nf[3] = {(double)(float)U[0 * 3 + 2], (double)(float)U[1 * 3 + 2], (double)(float)U[2 * 3 + 2]};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) out[a * 3 + b] = (a == b ? 1.0 : 0.0) - (1.0 - eps) * nf[a] * nf[b];
    }
    if (normals) for (int a = 0; a < 3; ++a) normals[(size_t)i * 3 + a] = U[a * 3 + 2];
  }
  free(nn_i); free(nn_d);
  return 0;
}
int s3o_gicp_covariances(const float* xyz, int n, int k, double eps, double* cov, double* normals) {
  s3o_kdtree* t = s3o_kdtree_build(xyz, n);
  int rc = gicp_covariances_tree(t, xyz, n, k, eps, cov, normals);
  s3o_kdtree_free(t);
  return rc;
}

/* ------------------------------------------------------------------ GICP (A4, A5, A8) */
static int g_eval_double; static double g_perturb; static int g_trace;

typedef struct {
  const float* src;      /* `output` cloud: pcl source transformed by guess, packed xyz */
  const float* tgt;      /* pcl target */
  const int*   idx_src;
  const int*   idx_tgt;
  const double* mahal;   /* per pcl-source point 3x3 row-major */
  int m;
  int evals;
} gicp_functor;

/* PCL gicp.hpp applyState on base_transformation_ = Identity:
 * R = Rz(x5) Ry(x4) Rx(x3) as float, t = float(x0..2).  (Eigen composes
 * AngleAxisf quaternions in float; here R is formed in double and rounded.) */
static void apply_state(const double x[6], float t[16]) {
  double cphi = cos(x[3]), sphi = sin(x[3]);
  double cth = cos(x[4]), sth = sin(x[4]);
  double cpsi = cos(x[5]), spsi = sin(x[5]);
  double R[3][3] = {{cpsi * cth, cpsi * sth * sphi - spsi * cphi, cpsi * sth * cphi + spsi * sphi},
                    {spsi * cth, spsi * sth * sphi + cpsi * cphi, spsi * sth * cphi - cpsi * sphi},
                    {-sth, cth * sphi, cth * cphi}};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) M4(t, r, c) = (float)R[r][c];
    M4(t, r, 3) = (float)x[r];
    M4(t, 3, r) = 0.f;
  }
  M4(t, 3, 3) = 1.f;
}

/* PCL OptimizationFunctorWithIndices::fdf — f and gradient in one pass */
static void gicp_fdf(gicp_functor* F, const double x[6], double* f_out, double g[6], int want_f, int want_g) {
  float T[16];
  apply_state(x, T);
  double Td[3][4];
  {
    double cphi = cos(x[3]), sphi = sin(x[3]), cth = cos(x[4]), sth = sin(x[4]), cpsi = cos(x[5]), spsi = sin(x[5]);
    Td[0][0] = cpsi * cth; Td[0][1] = cpsi * sth * sphi - spsi * cphi; Td[0][2] = cpsi * sth * cphi + spsi * sphi;
    Td[1][0] = spsi * cth; Td[1][1] = spsi * sth * sphi + cpsi * cphi; Td[1][2] = spsi * sth * cphi - cpsi * sphi;
    Td[2][0] = -sth; Td[2][1] = cth * sphi; Td[2][2] = cth * cphi;
    Td[0][3] = x[0]; Td[1][3] = x[1]; Td[2][3] = x[2];
  }
  double f = 0, gt[3] = {0, 0, 0}, Rs[3][3] = {{0}};
  F->evals++;
  for (int i = 0; i < F->m; ++i) {
    const float* ps = F->src + (size_t)F->idx_src[i] * 3;
    const float* pt = F->tgt + (size_t)F->idx_tgt[i] * 3;
    double res[3];
    if (!g_eval_double) {
      float pp[3];
      xf_eigen(T, ps, pp); /* Eigen::Vector4f pp(transformation_matrix * p_src) */
      res[0] = (double)(pp[0] - pt[0]); res[1] = (double)(pp[1] - pt[1]); res[2] = (double)(pp[2] - pt[2]);
    } else if (g_eval_double == 1) { /* variant 1: same float matrix, products and sums carried in double */
      for (int a = 0; a < 3; ++a)
        res[a] = ((double)M4(T, a, 0) * ps[0] + (double)M4(T, a, 1) * ps[1] + (double)M4(T, a, 2) * ps[2] +
                  (double)M4(T, a, 3)) - (double)pt[a];
    } else { /* variant 2: the matrix itself is kept in double (smooth objective) */
      for (int a = 0; a < 3; ++a)
        res[a] = (Td[a][0] * ps[0] + Td[a][1] * ps[1] + Td[a][2] * ps[2] + Td[a][3]) - (double)pt[a];
    }
    const double* M = F->mahal + (size_t)F->idx_src[i] * 9;
    double tmp[3];
    for (int a = 0; a < 3; ++a) tmp[a] = M[a * 3 + 0] * res[0] + M[a * 3 + 1] * res[1] + M[a * 3 + 2] * res[2];
    if (want_f) f += res[0] * tmp[0] + res[1] * tmp[1] + res[2] * tmp[2];
    if (want_g) {
      for (int a = 0; a < 3; ++a) gt[a] += tmp[a];
      /* pp = base_transformation_ (Identity) * p_src */
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rs[a][b] += (double)ps[a] * tmp[b];
    }
  }
  if (want_f) *f_out = f / (double)F->m;
  if (want_g) {
    double s = 2.0 / (double)F->m;
    for (int a = 0; a < 3; ++a) g[a] = gt[a] * s;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Rs[a][b] *= s;
    /* computeRDerivative: g[3+k] = sum_ij dR_k(j,i) * Rs(i,j) */
    double cphi = cos(x[3]), sphi = sin(x[3]);
    double cth = cos(x[4]), sth = sin(x[4]);
    double cpsi = cos(x[5]), spsi = sin(x[5]);
    double dPhi[3][3] = {{0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth},
                         {0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth},
                         {0, cphi * cth, -cth * sphi}};
    double dTh[3][3] = {{-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth},
                        {-spsi * sth, cth * sphi * spsi, cphi * cth * spsi},
                        {-cth, -sphi * sth, -cphi * sth}};
    double dPsi[3][3] = {{-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth},
                         {cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth},
                         {0, 0, 0}};
    g[3] = g[4] = g[5] = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        g[3] += dPhi[j][i] * Rs[i][j];
        g[4] += dTh[j][i] * Rs[i][j];
        g[5] += dPsi[j][i] * Rs[i][j];
      }
  }
}

/* diagnostics / variants (all default to the PCL-literal behaviour) */
static int    g_eval_double = 0; /* see s3d_oracle.h: 0 PCL-literal, 1 double arithmetic, 2 double matrix */
static double g_perturb = 0.0;   /* relative noise injected into the Mahalanobis matrices */
static unsigned long long g_perturb_seed = 1;
static int    g_trace = 0;
void s3o_set_eval_precision(int mode) { g_eval_double = mode; }
/* PCS.cpp:149-162: the reference is built with pclomp (1, default: GICP_OMP / NDT_OMP run) or without (0: they throw) */
static int g_omp_available = 1;
void s3o_set_omp_available(int on) { g_omp_available = on; }
void s3o_set_debug_perturbation(double rel) { g_perturb = rel; }
void s3o_set_debug_perturbation_seed(unsigned long long seed) { g_perturb_seed = seed; }
/* the conditioning probe's noise: a pure function of (seed, outer iteration, correspondence, matrix entry), so that the
 * experiment is reproducible and independent of the thread / call order (splitmix64 finaliser -> [0, 1)) */
static double perturb_unit(unsigned long long seed, int iter, int i, int entry) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)(unsigned)iter << 40) +
                         ((unsigned long long)(unsigned)i << 4) + (unsigned long long)entry;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
void s3o_set_trace(int on) { g_trace = on; }

/* ---- PCL registration/bfgs.h (a port of GSL vector_bfgs2 + Fletcher line search) */
enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };

typedef struct {
  gicp_functor* F;
  double f, gradient[6];
  double x0[6], g0[6], p[6], dx0[6], dg0[6];
  double g0norm, pnorm, fp0, delta_f;
  /* line-search cache ("wrapper" of GSL) */
  double x_alpha[6], g_alpha[6], f_alpha, df_alpha;
  double f_cache_key, df_cache_key, x_cache_key, g_cache_key;
} bfgs_t;

static double vnorm6(const double* v) { double s = 0; for (int i = 0; i < 6; ++i) s += v[i] * v[i]; return sqrt(s); }
static double vdot6(const double* a, const double* b) { double s = 0; for (int i = 0; i < 6; ++i) s += a[i] * b[i]; return s; }

static void bfgs_move_to(bfgs_t* b, double alpha) {
  if (alpha == b->x_cache_key) return;
  for (int i = 0; i < 6; ++i) b->x_alpha[i] = b->x0[i] + alpha * b->p[i];
  b->x_cache_key = alpha;
}
static double bfgs_slope(bfgs_t* b) { return vdot6(b->g_alpha, b->p); }
static double bfgs_apply_f(bfgs_t* b, double alpha) {
  if (alpha == b->f_cache_key) return b->f_alpha;
  bfgs_move_to(b, alpha);
  gicp_fdf(b->F, b->x_alpha, &b->f_alpha, NULL, 1, 0);
  b->f_cache_key = alpha;
  return b->f_alpha;
}
static double bfgs_apply_df(bfgs_t* b, double alpha) {
  if (alpha == b->df_cache_key) return b->df_alpha;
  bfgs_move_to(b, alpha);
  if (alpha != b->g_cache_key) {
    gicp_fdf(b->F, b->x_alpha, NULL, b->g_alpha, 0, 1);
    b->g_cache_key = alpha;
  }
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = alpha;
  return b->df_alpha;
}
static void bfgs_apply_fdf(bfgs_t* b, double alpha, double* f, double* df) {
  if (alpha == b->f_cache_key && alpha == b->df_cache_key) { *f = b->f_alpha; *df = b->df_alpha; return; }
  if (alpha == b->f_cache_key || alpha == b->df_cache_key) {
    *f = bfgs_apply_f(b, alpha);
    *df = bfgs_apply_df(b, alpha);
    return;
  }
  bfgs_move_to(b, alpha);
  gicp_fdf(b->F, b->x_alpha, &b->f_alpha, b->g_alpha, 1, 1);
  b->f_cache_key = alpha;
  b->g_cache_key = alpha;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = alpha;
  *f = b->f_alpha; *df = b->df_alpha;
}
static void bfgs_update_position(bfgs_t* b, double alpha, double x[6], double* f, double g[6]) {
  double fa, dfa;
  bfgs_apply_fdf(b, alpha, &fa, &dfa);
  *f = fa;
  memcpy(x, b->x_alpha, sizeof b->x_alpha);
  memcpy(g, b->g_alpha, sizeof b->g_alpha);
}
static void bfgs_change_direction(bfgs_t* b) {
  memcpy(b->x_alpha, b->x0, sizeof b->x0);
  b->x_cache_key = 0.0;
  b->f_cache_key = 0.0;
  memcpy(b->g_alpha, b->g0, sizeof b->g0);
  b->g_cache_key = 0.0;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = 0.0;
}

static double poly3(const double c[4], double z) { return c[0] + z * (c[1] + z * (c[2] + z * c[3])); }
static void check_extremum(const double c[4], double z, double* zmin, double* fmin) {
  double y = poly3(c, z);
  if (y < *fmin) { *zmin = z; *fmin = y; }
}
/* PCL BFGS::interpolate (GSL interpolate / interp_cubic / interp_quad) */
static double bfgs_interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin,
                               double xmax, int order) {
  double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
  if (ymin > ymax) { double t = ymin; ymin = ymax; ymax = t; }
  if (order > 2 && !(fpb != fpb) && fpb != INFINITY) {
    fpa = fpa * (b - a);
    fpb = fpb * (b - a);
    double eta = 3 * (fb - fa) - 2 * fpa - fpb;
    double xi = fpa + fpb - 2 * (fb - fa);
    double c[4] = {fa, fpa, eta, xi};
    y = ymin;
    double fmin = poly3(c, ymin);
    check_extremum(c, ymax, &y, &fmin);
    /* roots of c1 + 2 c2 z + 3 c3 z^2 */
    double A = 3 * c[3], B = 2 * c[2], C = c[1];
    if (A == 0) {
      if (B != 0) {
        double y0 = -C / B;
        if (y0 > ymin && y0 < ymax) check_extremum(c, y0, &y, &fmin);
      }
    } else {
      double disc = B * B - 4 * A * C;
      if (disc > 0) {
        double sq = sqrt(disc);
        double tq = -0.5 * (B + (B > 0 ? sq : -sq));
        double y0 = tq / A, y1 = (tq != 0) ? C / tq : y0;
        if (y0 > y1) { double t = y0; y0 = y1; y1 = t; }
        if (y0 > ymin && y0 < ymax) check_extremum(c, y0, &y, &fmin);
        if (y1 > ymin && y1 < ymax) check_extremum(c, y1, &y, &fmin);
      } else if (disc == 0) {
        double y0 = -0.5 * B / A;
        if (y0 > ymin && y0 < ymax) check_extremum(c, y0, &y, &fmin);
      }
    }
  } else {
    fpa = fpa * (b - a);
    double fl = fa + ymin * (fpa + ymin * (fb - fa - fpa));
    double fh = fa + ymax * (fpa + ymax * (fb - fa - fpa));
    double c = 2 * (fb - fa - fpa); /* curvature */
    y = ymin;
    double fmin = fl;
    if (fh < fmin) { y = ymax; fmin = fh; }
    /* PCL bfgs.h writes `if (c > a)` where GSL has `c > 0`; a == 0 in the
     * common first-bracket case.  Restated as PCL has it. */
    if (c > a) {
      double z = -fpa / c;
      if (z > ymin && z < ymax) {
        double f = fa + z * (fpa + z * (fb - fa - fpa));
        if (f < fmin) { y = z; fmin = f; }
      }
    }
  }
  return a + y * (b - a);
}

static int bfgs_line_search(bfgs_t* B, double rho, double sigma, double tau1, double tau2, double tau3, int order,
                            double alpha1, double* alpha_new) {
  const int bracket_iters = 100, section_iters = 100;
  double f0, fp0, falpha, falpha_prev, fpalpha, fpalpha_prev, delta, alpha_next;
  double alpha = alpha1, alpha_prev = 0.0;
  double a, b, fa, fb, fpa, fpb;
  int i = 0;
  bfgs_apply_fdf(B, 0.0, &f0, &fp0);
  falpha_prev = f0; fpalpha_prev = fp0;
  a = 0.0; b = alpha; fa = f0; fb = 0.0; fpa = fp0; fpb = 0.0;
  while (i++ < bracket_iters) {
    falpha = bfgs_apply_f(B, alpha);
    if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
      a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
      b = alpha; fb = falpha; fpb = NAN;
      break;
    }
    fpalpha = bfgs_apply_df(B, alpha);
    if (fabs(fpalpha) <= -sigma * fp0) { *alpha_new = alpha; return BFGS_SUCCESS; }
    if (fpalpha >= 0) {
      a = alpha; fa = falpha; fpa = fpalpha;
      b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
      break;
    }
    delta = alpha - alpha_prev;
    {
      double lower = alpha + delta, upper = alpha + tau1 * delta;
      alpha_next = bfgs_interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, lower, upper, order);
    }
    alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
  }
  while (i++ < section_iters) {
    delta = b - a;
    {
      double lower = a + tau2 * delta, upper = b - tau3 * delta;
      alpha = bfgs_interpolate(a, fa, fpa, b, fb, fpb, lower, upper, order);
    }
    falpha = bfgs_apply_f(B, alpha);
    if ((a - alpha) * fpa <= DBL_EPSILON) return BFGS_NOPROGRESS; /* roundoff prevents progress */
    if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
      b = alpha; fb = falpha; fpb = NAN;
    } else {
      fpalpha = bfgs_apply_df(B, alpha);
      if (fabs(fpalpha) <= -sigma * fp0) { *alpha_new = alpha; return BFGS_SUCCESS; }
      if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
        b = a; fb = fa; fpb = fpa;
        a = alpha; fa = falpha; fpa = fpalpha;
      } else {
        a = alpha; fa = falpha; fpa = fpalpha;
      }
    }
  }
  return BFGS_SUCCESS;
}

static void bfgs_init(bfgs_t* b, gicp_functor* F, const double x[6]) {
  memset(b, 0, sizeof *b);
  b->F = F;
  b->delta_f = 0;
  gicp_fdf(F, x, &b->f, b->gradient, 1, 1);
  memcpy(b->x0, x, sizeof b->x0);
  memcpy(b->g0, b->gradient, sizeof b->g0);
  b->g0norm = vnorm6(b->g0);
  for (int i = 0; i < 6; ++i) b->p[i] = b->gradient[i] * -1 / b->g0norm;
  b->pnorm = vnorm6(b->p);
  b->fp0 = -b->g0norm;
  memcpy(b->x_alpha, b->x0, sizeof b->x0);
  b->x_cache_key = 0;
  b->f_alpha = b->f; b->f_cache_key = 0;
  memcpy(b->g_alpha, b->g0, sizeof b->g0);
  b->g_cache_key = 0;
  b->df_alpha = bfgs_slope(b);
  b->df_cache_key = 0;
}

static int bfgs_one_step(bfgs_t* b, double x[6]) {
  /* parameters set at PCL gicp.hpp estimateRigidTransformationBFGS */
  const double sigma = 0.01, rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1.0;
  const int order = 3;
  double alpha = 0.0, alpha1;
  double f0 = b->f;
  if (b->pnorm == 0.0 || b->g0norm == 0.0 || b->fp0 == 0) return BFGS_NOPROGRESS;
  if (b->delta_f < 0) {
    double del = fmax(-b->delta_f, 10 * DBL_EPSILON * fabs(f0));
    alpha1 = fmin(1.0, 2.0 * del / (-b->fp0));
  } else {
    alpha1 = fabs(step_size);
  }
  int status = bfgs_line_search(b, rho, sigma, tau1, tau2, tau3, order, alpha1, &alpha);
  if (status != BFGS_SUCCESS) return status;
  bfgs_update_position(b, alpha, x, &b->f, b->gradient);
  b->delta_f = b->f - f0;
  {
    double dxg, dgg, dxdg, dgnorm, A, Bc;
    for (int i = 0; i < 6; ++i) { b->dx0[i] = x[i] - b->x0[i]; b->dg0[i] = b->gradient[i] - b->g0[i]; }
    dxg = vdot6(b->dx0, b->gradient);
    dgg = vdot6(b->dg0, b->gradient);
    dxdg = vdot6(b->dx0, b->dg0);
    dgnorm = vnorm6(b->dg0);
    if (dxdg != 0) {
      Bc = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
    } else {
      Bc = 0; A = 0;
    }
    for (int i = 0; i < 6; ++i) b->p[i] = -A * b->dx0[i] + b->gradient[i] - Bc * b->dg0[i];
  }
  memcpy(b->g0, b->gradient, sizeof b->g0);
  memcpy(b->x0, x, sizeof b->x0);
  b->g0norm = vnorm6(b->g0);
  b->pnorm = vnorm6(b->p);
  double dir = (vdot6(b->p, b->gradient) > 0) ? -1.0 : 1.0;
  for (int i = 0; i < 6; ++i) b->p[i] *= dir / b->pnorm;
  b->pnorm = vnorm6(b->p);
  b->fp0 = vdot6(b->p, b->g0);
  bfgs_change_direction(b);
  return BFGS_SUCCESS;
}

/* PCL gicp.hpp estimateRigidTransformationBFGS.  returns 0 ok, -1 = exception */
static int gicp_estimate_bfgs(gicp_functor* F, int max_inner, float T[16], int* inner_out) {
  if (F->m < 4) return -1; /* NotEnoughPointsException */
  double x[6];
  x[0] = M4(T, 0, 3); x[1] = M4(T, 1, 3); x[2] = M4(T, 2, 3);
  x[3] = atan2((double)M4(T, 2, 1), (double)M4(T, 2, 2));
  x[4] = asin(-(double)M4(T, 2, 0));
  x[5] = atan2((double)M4(T, 1, 0), (double)M4(T, 0, 0));
  const double gradient_tol = 1e-2;
  bfgs_t b;
  bfgs_init(&b, F, x);
  int inner = 0, result = BFGS_RUNNING;
  do {
    inner++;
    result = bfgs_one_step(&b, x);
    if (result) break;
    result = vnorm6(b.gradient) < gradient_tol ? BFGS_SUCCESS : BFGS_RUNNING; /* testGradient */
  } while (result == BFGS_RUNNING && inner < max_inner);
  *inner_out = inner;
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == max_inner) {
    apply_state(x, T); /* setIdentity(); applyState() */
    return 0;
  }
  return -1; /* SolverDidntConvergeException */
}

static const float IDENT4F[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

/* Registration::getFitnessScore (PCL registration.hpp), called at PCS.cpp:73 with
 * max_range = max_correspondence_distance (compared against SQUARED distances). */
static double fitness_tree(const s3o_kdtree* tree, const float* pcl_source, int m, const float final_tf[16],
                           double max_range) {
  double score = 0;
  int nr = 0;
  for (int i = 0; i < m; ++i) {
    float q[3];
    xf_pcl(final_tf, pcl_source + (size_t)i * 3, q);
    int j; float d2;
    s3o_kdtree_nn1(tree, q, &j, &d2);
    if ((double)d2 <= max_range) { score += d2; nr++; }
  }
  return nr > 0 ? score / nr : DBL_MAX;
}
double s3o_fitness_score(const float* pcl_source, int m, const float* pcl_target, int n, const float final_tf[16],
                         double max_range) {
  s3o_kdtree* t = s3o_kdtree_build(pcl_target, n);
  double f = fitness_tree(t, pcl_source, m, final_tf, max_range);
  s3o_kdtree_free(t);
  return f;
}

/* PCL gicp.hpp computeTransformation + Registration::align + PCS.cpp:52-82 */
int s3o_gicp(const float* input, int m, const float* target, int n, const float guess[16],
             const s3d_reg_params* cfg, int force_iterations, s3o_icp_result* out) {
  memset(out, 0, sizeof *out);
  const int k = cfg->correspondence_randomness;
  const double gicp_epsilon = 0.001; /* PCL default gicp_epsilon_ */
  s3o_kdtree* tree = s3o_kdtree_build(target, n);       /* tree_            */
  s3o_kdtree* tree_r = s3o_kdtree_build(input, m);      /* tree_reciprocal_ */
  double* cov_t = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  double* cov_i = (double*)malloc(sizeof(double) * 9 * (size_t)m);
  double* mahal = (double*)malloc(sizeof(double) * 9 * (size_t)m);
  float* output = (float*)malloc(sizeof(float) * 3 * (size_t)m);
  int* si = (int*)malloc(sizeof(int) * (size_t)m);
  int* ti = (int*)malloc(sizeof(int) * (size_t)m);
  int rc = 0;
  if (gicp_covariances_tree(tree, target, n, k, gicp_epsilon, cov_t, NULL) ||
      gicp_covariances_tree(tree_r, input, m, k, gicp_epsilon, cov_i, NULL)) {
    rc = -1;
    goto done;
  }
  for (int i = 0; i < m; ++i)
    for (int a = 0; a < 9; ++a) mahal[(size_t)i * 9 + a] = (a % 4 == 0) ? 1.0 : 0.0;
  float transformation[16], previous[16];
  memcpy(transformation, IDENT4F, sizeof IDENT4F);
  memcpy(previous, IDENT4F, sizeof IDENT4F);
  int nr_iterations = 0, converged = 0;
  const double dist_threshold = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
  /* pcl::transformPointCloud(output, output, guess) */
  for (int i = 0; i < m; ++i) xf_pcl(guess, input + (size_t)i * 3, output + (size_t)i * 3);
  int cnt = 0;
  while (!converged) {
    cnt = 0;
    double tR[4][4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int kk = 0; kk < 4; ++kk) s += (double)M4(transformation, i, kk) * (double)M4(guess, kk, j);
        tR[i][j] = s;
      }
    for (int i = 0; i < m; ++i) {
      float q[3];
      xf_eigen(transformation, output + (size_t)i * 3, q);
      int j; float d2;
      s3o_kdtree_nn1(tree, q, &j, &d2);
      if ((double)d2 < dist_threshold) {
        const double* C1 = cov_i + (size_t)i * 9;
        const double* C2 = cov_t + (size_t)j * 9;
        double Mt[3][3], tmp[3][3], inv[3][3];
        for (int a = 0; a < 3; ++a) /* M = R*C1 */
          for (int b = 0; b < 3; ++b) Mt[a][b] = tR[a][0] * C1[0 * 3 + b] + tR[a][1] * C1[1 * 3 + b] + tR[a][2] * C1[2 * 3 + b];
        for (int a = 0; a < 3; ++a) /* temp = M*R' + C2 */
          for (int b = 0; b < 3; ++b)
            tmp[a][b] = (Mt[a][0] * tR[b][0] + Mt[a][1] * tR[b][1] + Mt[a][2] * tR[b][2]) + C2[a * 3 + b];
        mat3_inverse(tmp, inv);
        if (g_perturb != 0.0) /* conditioning probe, see s3o_set_debug_perturbation */
          for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b) {
              double r = 1.0 + g_perturb * (perturb_unit(g_perturb_seed, nr_iterations, i, a * 3 + b) - 0.5);
              inv[a][b] *= r; inv[b][a] = inv[a][b];
            }
        memcpy(mahal + (size_t)i * 9, inv, sizeof inv);
        si[cnt] = i; ti[cnt] = j; cnt++;
      }
    }
    memcpy(previous, transformation, sizeof previous);
    gicp_functor F = {output, target, si, ti, mahal, cnt, 0};
    int inner = 0;
    if (gicp_estimate_bfgs(&F, cfg->maximum_optimizer_iterations, transformation, &inner)) {
      out->evaluations_total += F.evals;
      break; /* PCLException: converged_ stays false */
    }
    out->inner_iterations_total += inner;
    out->evaluations_total += F.evals;
    if (g_trace) fprintf(stderr, "[oracle] it %d cnt %d inner %d evals %d t=(%.9g %.9g %.9g) r21 %.9g r10 %.9g\n", nr_iterations, cnt, inner, F.evals, M4(transformation,0,3), M4(transformation,1,3), M4(transformation,2,3), M4(transformation,2,1), M4(transformation,1,0));
    double delta = 0;
    for (int kk = 0; kk < 4; ++kk)
      for (int l = 0; l < 4; ++l) {
        double ratio = (kk < 3 && l < 3) ? 1.0 / cfg->rotation_epsilon : 1.0 / cfg->transformation_epsilon;
        double c_delta = ratio * fabs((double)M4(previous, kk, l) - (double)M4(transformation, kk, l));
        if (c_delta > delta) delta = c_delta;
      }
    nr_iterations++;
    if (nr_iterations >= cfg->maximum_iterations || (!force_iterations && delta < 1)) {
      converged = 1;
      memcpy(previous, transformation, sizeof previous);
    }
  }
  mat4f_mul(previous, guess, out->final_transformation);
  out->converged = converged;
  out->iterations = nr_iterations;
  out->correspondences = cnt;
  out->fitness = fitness_tree(tree, input, m, out->final_transformation, cfg->max_correspondence_distance);
done:
  free(cov_t); free(cov_i); free(mahal); free(output); free(si); free(ti);
  s3o_kdtree_free(tree); s3o_kdtree_free(tree_r);
  return rc;
}

/* GICP objective of a candidate result F (= final_transformation_): correspondences and
 * Mahalanobis matrices are rebuilt at F exactly as one outer iteration of
 * computeTransformation would (guess = F, transformation_ = I), and the functor's f is
 * returned.  Lets tests rank two candidate results by the reference's own objective. */
double s3o_gicp_cost(const float* input, int m, const float* target, int n, const float F[16],
                     const s3d_reg_params* cfg, int* n_corr) {
  const int k = cfg->correspondence_randomness;
  s3o_kdtree* tree = s3o_kdtree_build(target, n);
  s3o_kdtree* tree_r = s3o_kdtree_build(input, m);
  double* cov_t = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  double* cov_i = (double*)malloc(sizeof(double) * 9 * (size_t)m);
  double cost = DBL_MAX;
  int cnt = 0;
  if (!gicp_covariances_tree(tree, target, n, k, 0.001, cov_t, NULL) &&
      !gicp_covariances_tree(tree_r, input, m, k, 0.001, cov_i, NULL)) {
    const double thr = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
    double f = 0;
    for (int i = 0; i < m; ++i) {
      float q[3];
      xf_pcl(F, input + (size_t)i * 3, q);
      int j; float d2;
      s3o_kdtree_nn1(tree, q, &j, &d2);
      if (!((double)d2 < thr)) continue;
      const double* C1 = cov_i + (size_t)i * 9;
      const double* C2 = cov_t + (size_t)j * 9;
      double Mt[3][3], tmp[3][3], inv[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          Mt[a][b] = (double)M4(F, a, 0) * C1[0 * 3 + b] + (double)M4(F, a, 1) * C1[1 * 3 + b] + (double)M4(F, a, 2) * C1[2 * 3 + b];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          tmp[a][b] = (Mt[a][0] * (double)M4(F, b, 0) + Mt[a][1] * (double)M4(F, b, 1) + Mt[a][2] * (double)M4(F, b, 2)) + C2[a * 3 + b];
      mat3_inverse(tmp, inv);
      double res[3] = {(double)q[0] - target[(size_t)j * 3], (double)q[1] - target[(size_t)j * 3 + 1], (double)q[2] - target[(size_t)j * 3 + 2]};
      for (int a = 0; a < 3; ++a)
        f += res[a] * (inv[a][0] * res[0] + inv[a][1] * res[1] + inv[a][2] * res[2]);
      cnt++;
    }
    if (cnt > 0) cost = f / cnt;
  }
  if (n_corr) *n_corr = cnt;
  free(cov_t); free(cov_i);
  s3o_kdtree_free(tree); s3o_kdtree_free(tree_r);
  return cost;
}

/* ------------------------------------------------------------------ point-to-plane ICP
 * Not in the reference (enumerator ICP has no `case`, PCS.cpp:139-165).  Defined here
 * (and in DESIGN.md) so that the HIP path for that enumerator has a CPU statement:
 *   normals  = smallest-eigenvector of the k-NN covariance of the pcl TARGET cloud
 *              (the same pre-pass GICP runs, k = correspondence_randomness);
 *   outer loop, distance gate, delta stopping rule, fitness: as GICP above;
 *   step: Gauss-Newton on sum (n_j . (p' + w x p' + dt - q_j))^2, 6x6 normal equations
 *         solved by Cholesky in double, update T <- Exp(w, dt) * T kept in float. */
static int chol6_solve(double A[6][6], double b[6], double x[6]) {
  double L[6][6] = {{0}};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) { if (s <= 0) return -1; L[i][i] = sqrt(s); }
      else L[i][j] = s / L[j][j];
    }
  double y[6];
  for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
  return 0;
}
static void rodrigues(const double w[3], double R[3][3]) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2), a, b;
  if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double kk = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
      R[i][j] = (i == j) + a * K[i][j] + b * kk;
    }
}

int s3o_icp_point_to_plane(const float* input, int m, const float* target, int n, const float guess[16],
                           const s3d_reg_params* cfg, int force_iterations, s3o_icp_result* out) {
  memset(out, 0, sizeof *out);
  const int k = cfg->correspondence_randomness;
  s3o_kdtree* tree = s3o_kdtree_build(target, n);
  double* cov_t = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  double* nrm = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  float* output = (float*)malloc(sizeof(float) * 3 * (size_t)m);
  int rc = 0;
  if (gicp_covariances_tree(tree, target, n, k, 0.001, cov_t, nrm)) { rc = -1; goto done; }
  float transformation[16], previous[16];
  memcpy(transformation, IDENT4F, sizeof IDENT4F);
  memcpy(previous, IDENT4F, sizeof IDENT4F);
  const double dist_threshold = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
  for (int i = 0; i < m; ++i) xf_pcl(guess, input + (size_t)i * 3, output + (size_t)i * 3);
  int nr_iterations = 0, converged = 0, cnt = 0;
  while (!converged) {
    double A[6][6] = {{0}}, b[6] = {0};
    cnt = 0;
    for (int i = 0; i < m; ++i) {
      float q[3];
      xf_eigen(transformation, output + (size_t)i * 3, q);
      int j; float d2;
      s3o_kdtree_nn1(tree, q, &j, &d2);
      if (!((double)d2 < dist_threshold)) continue;
      /* the normal is rounded to float: the definition of this mode stores xyz + normal as float4 */
      const double nx = (double)(float)nrm[(size_t)j * 3 + 0], ny = (double)(float)nrm[(size_t)j * 3 + 1],
                   nz = (double)(float)nrm[(size_t)j * 3 + 2];
      const double px = q[0], py = q[1], pz = q[2];
      const float* t = target + (size_t)j * 3;
      double r = nx * (px - t[0]) + ny * (py - t[1]) + nz * (pz - t[2]);
      double J[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
      for (int a = 0; a < 6; ++a) {
        b[a] -= J[a] * r;
        for (int c = 0; c <= a; ++c) A[a][c] += J[a] * J[c];
      }
      cnt++;
    }
    for (int a = 0; a < 6; ++a)
      for (int c = a + 1; c < 6; ++c) A[a][c] = A[c][a];
    memcpy(previous, transformation, sizeof previous);
    double xi[6];
    if (cnt < 6 || chol6_solve(A, b, xi)) break; /* degenerate: converged_ stays false */
    double R[3][3];
    rodrigues(xi, R);
    float nt[16];
    memcpy(nt, IDENT4F, sizeof nt);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 4; ++c) {
        double s = 0;
        for (int kk = 0; kk < 3; ++kk) s += R[r][kk] * (double)M4(transformation, kk, c);
        if (c == 3) s += xi[3 + r];
        M4(nt, r, c) = (float)s;
      }
    }
    memcpy(transformation, nt, sizeof nt);
    double delta = 0;
    for (int kk = 0; kk < 4; ++kk)
      for (int l = 0; l < 4; ++l) {
        double ratio = (kk < 3 && l < 3) ? 1.0 / cfg->rotation_epsilon : 1.0 / cfg->transformation_epsilon;
        double c_delta = ratio * fabs((double)M4(previous, kk, l) - (double)M4(transformation, kk, l));
        if (c_delta > delta) delta = c_delta;
      }
    nr_iterations++;
    if (nr_iterations >= cfg->maximum_iterations || (!force_iterations && delta < 1)) {
      converged = 1;
      memcpy(previous, transformation, sizeof previous);
    }
  }
  mat4f_mul(previous, guess, out->final_transformation);
  out->converged = converged;
  out->iterations = nr_iterations;
  out->correspondences = cnt;
  out->fitness = fitness_tree(tree, input, m, out->final_transformation, cfg->max_correspondence_distance);
done:
  free(cov_t); free(nrm); free(output);
  s3o_kdtree_free(tree);
  return rc;
}

/* ------------------------------------------------------------------ NDT
 * doNDT (PCS.cpp:84-117) -> pcl::NormalDistributionsTransform (PCL 1.12 ndt.hpp, voxel_grid_covariance.hpp),
 * restated from the published algorithm (Magnusson 2009, Algorithm 2 + More-Thuente 1994 line search as PCL
 * codes them).  PARITY UNPINNED like the rest of this file.  Deliberate simplifications, all below the 1e-4 bar:
 *   - the voxel centroid used for the radius query is the (double) mean rounded to float, PCL accumulates a
 *     separate float centroid;
 *   - voxels whose covariance cannot be inverted are dropped (PCL keeps them in the kd-tree);
 *   - Eigen's Transform::rotation() (an SVD clean-up) is taken as the 3x3 block of the guess;
 *   - the angle-derivative vectors j_ang / h_ang are formed as products of the elementary rotation matrices and
 *     their derivatives (the same numbers as PCL's closed forms up to rounding);
 *   - NDT_OMP is pclomp::NormalDistributionsTransform (koide3/ndt_omp, not in the image, pinned by nothing here): the
 *     same objective and optimiser with its default neighbour search DIRECT7 - the voxel holding the transformed point,
 *     floor(x / leaf), and its six face neighbours in the order 0 +x -x +y -y +z -z, every one with >= 6 points
 *     (VoxelGridCovariance::getNeighborhoodAtPoint7) - instead of the kd-tree radius query; its float point
 *     derivatives are NOT restated (double here as in PCL). */

typedef struct {
  int n;            /* valid cells */
  double* mean;     /* n * 3 */
  double* icov;     /* n * 9 */
  float* centroid;  /* n * 3 */
  s3o_kdtree* tree; /* over centroid */
  unsigned* key;    /* n, ascending: the voxel index of a cell (DIRECT7 look-up) */
  float leaf;
  int min_b[3], div_b[3];
} ndt_cells;

static void ndt_cells_free(ndt_cells* c) {
  free(c->mean); free(c->icov); free(c->centroid); free(c->key);
  if (c->tree) s3o_kdtree_free(c->tree);
  memset(c, 0, sizeof *c);
}

/* VoxelGridCovariance::applyFilter: leaf layout as pcl::VoxelGrid, >= 6 points per voxel, unbiased covariance,
 * eigenvalues below 0.01 * the largest are raised to it, inverse covariance */
static void ndt_build_cells(const float* xyz, int n, double resolution, ndt_cells* out) {
  memset(out, 0, sizeof *out);
  if (n <= 0) return;
  const float leaf = (float)resolution, inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const float v = xyz[(size_t)i * 3 + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; ++a) {
    min_b[a] = (int)floorf(mn[a] * inv);
    max_b[a] = (int)floorf(mx[a] * inv);
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  key_idx* kv = (key_idx*)malloc(sizeof(key_idx) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * 3;
    const int ijk0 = (int)floorf(p[0] * inv) - min_b[0], ijk1 = (int)floorf(p[1] * inv) - min_b[1],
              ijk2 = (int)floorf(p[2] * inv) - min_b[2];
    kv[i].key = (unsigned)(ijk0 + ijk1 * div_b[0] + ijk2 * div_b[0] * div_b[1]);
    kv[i].idx = i;
  }
  qsort(kv, (size_t)n, sizeof(key_idx), cmp_key_idx);
  out->mean = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  out->icov = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  out->centroid = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  out->key = (unsigned*)malloc(sizeof(unsigned) * (size_t)n);
  out->leaf = leaf;
  for (int a = 0; a < 3; ++a) { out->min_b[a] = min_b[a]; out->div_b[a] = div_b[a]; }
  int m = 0;
  for (int i = 0; i < n;) {
    int j = i;
    const unsigned cell_key = kv[i].key;
    double s[3] = {0, 0, 0}, c[3][3] = {{0}};
    while (j < n && kv[j].key == kv[i].key) {
      const float* p = xyz + (size_t)kv[j].idx * 3;
      const double pd[3] = {p[0], p[1], p[2]};
      for (int a = 0; a < 3; ++a) {
        s[a] += pd[a];
        for (int b = 0; b < 3; ++b) c[a][b] += pd[a] * pd[b];
      }
      ++j;
    }
    const int np = j - i;
    i = j;
    if (np < 6) continue;                                   /* min_points_per_voxel_ */
    double mean[3] = {s[0] / np, s[1] / np, s[2] / np};
    double cov[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) cov[a * 3 + b] = (c[a][b] - s[a] * mean[b]) / (np - 1.0);
    double ev[3], V[9];                                     /* descending, eigenvectors in columns */
    s3o_sym_eig3(cov, ev, V);
    if (ev[2] < -1e-12 || ev[1] < -1e-12 || ev[0] <= 0) continue;
    const double floor_ev = 0.01 * ev[0];                   /* min_covar_eigvalue_mult_ */
    if (ev[2] < floor_ev) {
      ev[2] = floor_ev;
      if (ev[1] < floor_ev) ev[1] = floor_ev;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          double v = 0;
          for (int k = 0; k < 3; ++k) v += V[a * 3 + k] * ev[k] * V[b * 3 + k];
          cov[a * 3 + b] = v;
        }
    }
    double cm[3][3], ci[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cm[a][b] = cov[a * 3 + b];
    mat3_inverse(cm, ci);
    int bad = 0;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) if (!isfinite(ci[a][b])) bad = 1;
    if (bad) continue;
    for (int a = 0; a < 3; ++a) {
      out->mean[(size_t)m * 3 + a] = mean[a];
      out->centroid[(size_t)m * 3 + a] = (float)mean[a];
      for (int b = 0; b < 3; ++b) out->icov[(size_t)m * 9 + a * 3 + b] = ci[a][b];
    }
    out->key[m] = cell_key;
    ++m;
  }
  free(kv);
  out->n = m;
  out->tree = s3o_kdtree_build(out->centroid, m);
}

typedef struct {
  const float* input;   /* the registration's source cloud (pcl input_), packed */
  int m;
  const ndt_cells* cells;
  double resolution, d1, d2;
  int evaluations;
  int direct7;          /* NDT_OMP: pclomp's default neighbour search */
} ndt_problem;

/* VoxelGridCovariance::getNeighborhoodAtPoint7 (pclomp): cell ids of the voxel of xt and its six face neighbours */
static int ndt_neighbourhood7(const ndt_cells* C, const float xt[3], int* out) {
  static const int rel[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
  int ijk[3], k = 0;
  for (int a = 0; a < 3; ++a) ijk[a] = (int)floorf(xt[a] / C->leaf) - C->min_b[a];
  for (int r = 0; r < 7; ++r) {
    const int v0 = ijk[0] + rel[r][0], v1 = ijk[1] + rel[r][1], v2 = ijk[2] + rel[r][2];
    if (v0 < 0 || v1 < 0 || v2 < 0 || v0 >= C->div_b[0] || v1 >= C->div_b[1] || v2 >= C->div_b[2]) continue;
    const unsigned key = (unsigned)(v0 + v1 * C->div_b[0] + v2 * C->div_b[0] * C->div_b[1]);
    int lo = 0, hi = C->n - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) / 2;
      if (C->key[mid] == key) { out[k++] = mid; break; }
      if (C->key[mid] < key) lo = mid + 1; else hi = mid - 1;
    }
  }
  return k;
}

static void m3mul(const double a[3][3], const double b[3][3], double o[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

/* point_gradient_ columns 3..5 = dR[k] x, point_hessian_ block (k, l) = d2R[k][l] x with R = Rx Ry Rz;
 * computeAngleDerivatives snaps angles below 10e-5 to cos = 1, sin = 0 */
static void ndt_angle_derivatives(const double p[6], double dR[3][3][3], double d2R[3][3][3][3]) {
  double cs[3], sn[3];
  for (int k = 0; k < 3; ++k) {
    if (fabs(p[3 + k]) < 10e-5) { cs[k] = 1.0; sn[k] = 0.0; }
    else { cs[k] = cos(p[3 + k]); sn[k] = sin(p[3 + k]); }
  }
  double E[3][3][3][3];   /* E[axis][order 0..2] */
  for (int ax = 0; ax < 3; ++ax) {
    const double c = cs[ax], s = sn[ax];
    const int u = (ax + 1) % 3, v = (ax + 2) % 3;
    for (int o = 0; o < 3; ++o) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[ax][o][i][j] = 0;
    E[ax][0][ax][ax] = 1;
    E[ax][0][u][u] = c;  E[ax][0][u][v] = -s; E[ax][0][v][u] = s;  E[ax][0][v][v] = c;
    E[ax][1][u][u] = -s; E[ax][1][u][v] = -c; E[ax][1][v][u] = c;  E[ax][1][v][v] = -s;
    E[ax][2][u][u] = -c; E[ax][2][u][v] = s;  E[ax][2][v][u] = -s; E[ax][2][v][v] = -c;
  }
  for (int k = 0; k < 3; ++k) {
    int o[3] = {0, 0, 0};
    o[k] = 1;
    double t[3][3];
    m3mul(E[0][o[0]], E[1][o[1]], t);
    m3mul(t, E[2][o[2]], dR[k]);
    for (int l = 0; l < 3; ++l) {
      int q[3] = {0, 0, 0};
      q[k] += 1; q[l] += 1;
      m3mul(E[0][q[0]], E[1][q[1]], t);
      m3mul(t, E[2][q[2]], d2R[k][l]);
    }
  }
}

/* convertTransform: (Translation * AngleAxis(x) * AngleAxis(y) * AngleAxis(z)).matrix() in float */
static void ndt_convert_transform(const double p[6], float T[16]) {
  const float a = (float)p[3], b = (float)p[4], c = (float)p[5];
  const float ca = cosf(a), sa = sinf(a), cb = cosf(b), sb = sinf(b), cc = cosf(c), sc = sinf(c);
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

/* computeDerivatives / computeHessian / updateDerivatives.  Returns the score. */
static double ndt_derivatives(ndt_problem* P, const float T[16], const double p[6], double g[6], double H[36],
                              int want_hessian) {
  double dR[3][3][3], d2R[3][3][3][3];
  ndt_angle_derivatives(p, dR, d2R);
  for (int i = 0; i < 6; ++i) g[i] = 0;
  for (int i = 0; i < 36; ++i) H[i] = 0;
  double score = 0;
  P->evaluations++;
  int nn_i[27];
  float nn_d[27];
  const float r2 = (float)(P->resolution * P->resolution);
  for (int i = 0; i < P->m; ++i) {
    const float* xf = P->input + (size_t)i * 3;
    float xt[3];
    xf_pcl(T, xf, xt);
    const int k = P->direct7 ? ndt_neighbourhood7(P->cells, xt, nn_i) : s3o_kdtree_knn(P->cells->tree, xt, 27, nn_i, nn_d);
    const double x[3] = {xf[0], xf[1], xf[2]};
    double J[3][6];   /* point_gradient_ */
    int have_j = 0;
    for (int c = 0; c < k; ++c) {
      if (!P->direct7 && !(nn_d[c] < r2)) break;             /* sorted ascending; FLANN radius search is strict */
      if (!have_j) {
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) J[a][b] = a == b ? 1.0 : 0.0;
          for (int kk = 0; kk < 3; ++kk) J[a][3 + kk] = dR[kk][a][0] * x[0] + dR[kk][a][1] * x[1] + dR[kk][a][2] * x[2];
        }
        have_j = 1;
      }
      const double* mu = P->cells->mean + (size_t)nn_i[c] * 3;
      const double* Ci = P->cells->icov + (size_t)nn_i[c] * 9;
      const double xq[3] = {(double)xt[0] - mu[0], (double)xt[1] - mu[1], (double)xt[2] - mu[2]};
      double Cx[3];
      for (int a = 0; a < 3; ++a) Cx[a] = Ci[a * 3] * xq[0] + Ci[a * 3 + 1] * xq[1] + Ci[a * 3 + 2] * xq[2];
      double e = exp(-P->d2 * (xq[0] * Cx[0] + xq[1] * Cx[1] + xq[2] * Cx[2]) / 2);
      const double score_inc = -P->d1 * e;
      e = P->d2 * e;
      if (e > 1 || e < 0 || e != e) continue;                /* updateDerivatives returns 0 */
      score += score_inc;
      e *= P->d1;
      double CJ[3][6], xCJ[6];
      for (int col = 0; col < 6; ++col) {
        for (int a = 0; a < 3; ++a) CJ[a][col] = Ci[a * 3] * J[0][col] + Ci[a * 3 + 1] * J[1][col] + Ci[a * 3 + 2] * J[2][col];
        xCJ[col] = xq[0] * CJ[0][col] + xq[1] * CJ[1][col] + xq[2] * CJ[2][col];
        g[col] += xCJ[col] * e;
      }
      if (!want_hessian) continue;
      for (int ii = 0; ii < 6; ++ii)
        for (int jj = 0; jj < 6; ++jj) {
          double hx = 0;                                     /* x^T C^-1 (d2 x / dp_i dp_j) */
          if (ii >= 3 && jj >= 3) {
            double hv[3];
            for (int a = 0; a < 3; ++a)
              hv[a] = d2R[ii - 3][jj - 3][a][0] * x[0] + d2R[ii - 3][jj - 3][a][1] * x[1] + d2R[ii - 3][jj - 3][a][2] * x[2];
            hx = Cx[0] * hv[0] + Cx[1] * hv[1] + Cx[2] * hv[2];
          }
          const double jcj = J[0][jj] * CJ[0][ii] + J[1][jj] * CJ[1][ii] + J[2][jj] * CJ[2][ii];
          H[ii * 6 + jj] += e * (-P->d2 * xCJ[ii] * xCJ[jj] + hx + jcj);
        }
    }
  }
  return score;
}

/* symmetric 6x6 Jacobi eigen-decomposition; solve H d = b in the least-squares sense (= JacobiSVD::solve) */
static void ndt_solve6(const double Hin[36], const double b[6], double d[6]) {
  double A[6][6], V[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { A[i][j] = 0.5 * (Hin[i * 6 + j] + Hin[j * 6 + i]); V[i][j] = i == j; }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { if (i != j) off += A[i][j] * A[i][j]; else diag += A[i][i] * A[i][i]; }
    if (off <= 1e-300 || off <= 1e-32 * diag) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 6; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 6; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  double lmax = 0;
  for (int i = 0; i < 6; ++i) if (fabs(A[i][i]) > lmax) lmax = fabs(A[i][i]);
  const double thr = 6 * DBL_EPSILON * lmax;                 /* Eigen's default SVD threshold: eps * max(rows, cols) */
  for (int i = 0; i < 6; ++i) d[i] = 0;
  for (int k = 0; k < 6; ++k) {
    if (!(fabs(A[k][k]) > thr)) continue;
    double vb = 0;
    for (int i = 0; i < 6; ++i) vb += V[i][k] * b[i];
    vb /= A[k][k];
    for (int i = 0; i < 6; ++i) d[i] += V[i][k] * vb;
  }
}

static double ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t,
                              double f_t, double g_t) {
  if (f_t > f_l) {                                           /* case 1 */
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {                                       /* case 2 */
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
  }
  if (fabs(g_t) <= fabs(g_l)) {                              /* case 3 */
    const double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l, w = sqrt(z * z - g_t * g_l);
    const double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
    const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    const double a_n = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
    return a_t > a_l ? fmin(a_t + 0.66 * (a_u - a_t), a_n) : fmax(a_t + 0.66 * (a_u - a_t), a_n);
  }
  {                                                          /* case 4 */
    const double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u, w = sqrt(z * z - g_t * g_u);
    return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
  }
}

static int ndt_update_interval(double* a_l, double* f_l, double* g_l, double* a_u, double* f_u, double* g_u, double a_t,
                               double f_t, double g_t) {
  if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
  if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  if (g_t * (*a_l - a_t) < 0) { *a_u = *a_l; *f_u = *f_l; *g_u = *g_l; *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
  return 1;
}

/* computeStepLengthMT.  x: current parameters; dir: unit direction (may be flipped); on return T / score / g / H
 * describe the accepted trial point. */
static double ndt_step_length(ndt_problem* P, const double x[6], double dir[6], double step_init, double step_max,
                              double step_min, double* score, double g[6], double H[36], float T[16]) {
  const double phi_0 = -*score;
  double d_phi_0 = -vdot6(g, dir);
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) return 0;
    d_phi_0 = -d_phi_0;
    for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1.e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
  double f_l = 0 /* psi(0) */, g_l = d_phi_0 - mu * d_phi_0, f_u = f_l, g_u = g_l;
  int interval_converged = (step_max - step_min) < 0, open_interval = 1;
  double a_t = step_init;
  a_t = fmin(a_t, step_max);
  a_t = fmax(a_t, step_min);
  double x_t[6];
  for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
  ndt_convert_transform(x_t, T);
  *score = ndt_derivatives(P, T, x_t, g, H, 1);
  double phi_t = -*score, d_phi_t = -vdot6(g, dir);
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t, d_psi_t = d_phi_t - mu * d_phi_0;
  while (!interval_converged && step_iterations < max_step_iterations && !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    if (open_interval) a_t = ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
    else a_t = ndt_trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = fmin(a_t, step_max);
    a_t = fmax(a_t, step_min);
    for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
    ndt_convert_transform(x_t, T);
    double Htmp[36];
    *score = ndt_derivatives(P, T, x_t, g, Htmp, 0);
    phi_t = -*score;
    d_phi_t = -vdot6(g, dir);
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
    d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = 0;
      f_l += phi_0 - mu * d_phi_0 * a_l;
      g_l += mu * d_phi_0;
      f_u += phi_0 - mu * d_phi_0 * a_u;
      g_u += mu * d_phi_0;
    }
    if (open_interval) interval_converged = ndt_update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
    else interval_converged = ndt_update_interval(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
    step_iterations++;
  }
  if (step_iterations) {                                     /* computeHessian at the accepted point */
    double gtmp[6];
    ndt_derivatives(P, T, x_t, gtmp, H, 1);
  }
  return a_t;
}

/* Eigen Matrix3f::eulerAngles(0, 1, 2) */
static void ndt_euler_xyz(const float R[3][3], float res[3]) {
  res[0] = atan2f(R[1][2], R[2][2]);
  const float c2 = sqrtf(R[0][0] * R[0][0] + R[0][1] * R[0][1]);
  if (res[0] > 0.f) {
    res[0] -= 3.14159265358979323846f;
    res[1] = atan2f(-R[0][2], -c2);
  } else {
    res[1] = atan2f(-R[0][2], c2);
  }
  const float s1 = sinf(res[0]), c1 = cosf(res[0]);
  res[2] = atan2f(s1 * R[2][0] - c1 * R[1][0], c1 * R[1][1] - s1 * R[2][1]);
  res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

int s3o_ndt(const float* input, int m, const float* target, int n, const float guess[16], const s3d_reg_params* cfg,
            s3o_icp_result* out) {
  memset(out, 0, sizeof *out);
  memcpy(out->final_transformation, guess, sizeof(float) * 16);
  ndt_cells cells;
  ndt_build_cells(target, n, (double)cfg->resolution, &cells);
  ndt_problem P;
  memset(&P, 0, sizeof P);
  P.input = input; P.m = m; P.cells = &cells; P.resolution = (double)cfg->resolution;
  P.direct7 = cfg->registration_algorithm == S3D_ALG_NDT_OMP;
  {                                                          /* init(): Eq. 6.8 [Magnusson 2009] */
    const double c1 = 10 * (1 - cfg->outlier_ratio), c2 = cfg->outlier_ratio / pow(P.resolution, 3), d3 = -log(c2);
    P.d1 = -log(c1 + c2) - d3;
    P.d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / P.d1);
  }
  float T[16];
  memcpy(T, guess, sizeof T);
  double p[6];
  {
    float R[3][3], e[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = guess[j * 4 + i];
    ndt_euler_xyz(R, e);
    for (int i = 0; i < 3; ++i) { p[i] = (double)guess[12 + i]; p[3 + i] = (double)e[i]; }
  }
  double g[6], H[36];
  double score = cells.n > 0 ? ndt_derivatives(&P, T, p, g, H, 1) : 0.0;
  int converged = 0, iterations = 0;
  while (!converged && cells.n > 0) {
    double mg[6], delta[6];
    for (int i = 0; i < 6; ++i) mg[i] = -g[i];
    ndt_solve6(H, mg, delta);
    double dn = vnorm6(delta);
    if (dn == 0 || dn != dn) { converged = dn == 0; break; }
    for (int i = 0; i < 6; ++i) delta[i] /= dn;
    dn = ndt_step_length(&P, p, delta, dn, cfg->step_size, cfg->transformation_epsilon / 2, &score, g, H, T);
    for (int i = 0; i < 6; ++i) { delta[i] *= dn; p[i] += delta[i]; }
    float Tstep[16];
    ndt_convert_transform(delta, Tstep);                     /* transformation_ of this iteration */
    const double translation_sqr = (double)Tstep[12] * Tstep[12] + (double)Tstep[13] * Tstep[13] + (double)Tstep[14] * Tstep[14];
    iterations++;
    /* transformation_rotation_epsilon_ is 0 (never set by slam3d): the 1.12 test reduces to this */
    if (iterations >= cfg->maximum_iterations ||
        (cfg->transformation_epsilon > 0 && translation_sqr <= cfg->transformation_epsilon))
      converged = 1;
  }
  memcpy(out->final_transformation, T, sizeof T);
  out->converged = converged;
  out->iterations = iterations;
  out->evaluations_total = P.evaluations;
  out->correspondences = cells.n;
  out->fitness = s3o_fitness_score(input, m, target, n, T, cfg->max_correspondence_distance);   /* PCS.cpp:107 */
  ndt_cells_free(&cells);
  return 0;
}

/* ------------------------------------------------------------------ align (A2) */

int s3o_align(const float* source, int n_source, int stride_source, const float* target, int n_target,
              int stride_target, const double guess[16], const s3d_reg_params* cfg, int force_iterations,
              double result[16], s3o_align_info* info) {
  s3o_align_info li;
  memset(&li, 0, sizeof li);
  int status = S3D_STATUS_OK;
  float* fs = (float*)malloc(sizeof(float) * 3 * (size_t)(n_source > 0 ? n_source : 1));
  float* ft = (float*)malloc(sizeof(float) * 3 * (size_t)(n_target > 0 ? n_target : 1));
  int ns, nt;
  if (cfg->point_cloud_density > 0) { /* PCS.cpp:127-131 */
    ns = s3o_voxel_downsample(source, n_source, stride_source, cfg->point_cloud_density, fs, NULL);
    nt = s3o_voxel_downsample(target, n_target, stride_target, cfg->point_cloud_density, ft, NULL);
  } else {
    ns = n_source; nt = n_target;
    for (int i = 0; i < ns; ++i) for (int a = 0; a < 3; ++a) fs[(size_t)i * 3 + a] = source[(size_t)i * stride_source + a];
    for (int i = 0; i < nt; ++i) for (int a = 0; a < 3; ++a) ft[(size_t)i * 3 + a] = target[(size_t)i * stride_target + a];
  }
  li.n_source_filtered = ns; li.n_target_filtered = nt;
  for (int i = 0; i < 16; ++i) result[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (nt < 100 || ns < 100) { status = S3D_STATUS_TOO_FEW_POINTS; goto done; } /* PCS.cpp:134-135 */
  float guess_f[16];
  for (int i = 0; i < 16; ++i) guess_f[i] = (float)guess[i]; /* PCS.cpp:70 guess.matrix().cast<float>() */
  s3o_icp_result r;
  int rc;
  switch (cfg->registration_algorithm) { /* PCS.cpp:139-165 */
    case S3D_ALG_GICP_OMP:
      if (!g_omp_available) { status = S3D_STATUS_OMP_UNAVAILABLE; goto done; }   /* PCS.cpp:159-161 */
      /* pclomp's GICP minimises the same objective */
      /* fall through */
    case S3D_ALG_GICP:
      /* setInputSource(target); setInputTarget(source)  PCS.cpp:68-69 */
      rc = s3o_gicp(ft, nt, fs, ns, guess_f, cfg, force_iterations, &r);
      break;
    case S3D_ALG_ICP:
      rc = s3o_icp_point_to_plane(ft, nt, fs, ns, guess_f, cfg, force_iterations, &r);
      break;
    case S3D_ALG_NDT_OMP:
      if (!g_omp_available) { status = S3D_STATUS_OMP_UNAVAILABLE; goto done; }   /* PCS.cpp:159-161 */
      /* fall through */
    case S3D_ALG_NDT:
      /* doNDT (PCS.cpp:84-117): same source/target swap */
      rc = s3o_ndt(ft, nt, fs, ns, guess_f, cfg, &r);
      break;
    default:
      status = S3D_STATUS_UNKNOWN_ALGORITHM; goto done;
  }
  if (rc) { status = S3D_STATUS_INVALID_ARGUMENT; goto done; }
  li.iterations = r.iterations; li.converged = r.converged; li.correspondences = r.correspondences;
  li.fitness = r.fitness;
  /* PCS.cpp:80: Transform(Eigen::Isometry3f(getFinalTransformation())) — widen, no re-orthonormalisation */
  for (int i = 0; i < 16; ++i) result[i] = (double)r.final_transformation[i];
  result[3] = result[7] = result[11] = 0.0; result[15] = 1.0;
  if (!r.converged) { status = S3D_STATUS_NOT_CONVERGED; goto done; }          /* PCS.cpp:74 */
  if (r.fitness > cfg->max_fitness_score) { status = S3D_STATUS_FITNESS_EXCEEDED; goto done; }
  {
    /* PCS.cpp:167-172 */
    double ginv[16], delta[16];
    s3o_mat4d_inverse_isometry(guess, ginv);
    s3o_mat4d_mul(ginv, result, delta);
    double tn = sqrt(M4(delta, 0, 3) * M4(delta, 0, 3) + M4(delta, 1, 3) * M4(delta, 1, 3) + M4(delta, 2, 3) * M4(delta, 2, 3));
    double ang = s3o_rotation_angle(delta);
    if (tn > cfg->max_translation || ang > cfg->max_rotation) status = S3D_STATUS_TOO_FAR_FROM_GUESS;
  }
done:
  free(fs); free(ft);
  if (info) *info = li;
  return status;
}

/* ------------------------------------------------------------------ createConstraint (A1) */

int s3o_create_constraint(const float* source, int n_source, int stride_source, const double source_sensor_pose[16],
                          const float* target, int n_target, int stride_target, const double target_sensor_pose[16],
                          const double odometry[16], int loop, const s3d_reg_params* fine,
                          const s3d_reg_params* coarse, double covariance_scale, double relative_pose[16],
                          double information[36], s3o_align_info* info) {
  double sinv[16], tinv[16], guess[16], tmp[16];
  s3o_mat4d_inverse_isometry(source_sensor_pose, sinv);
  s3o_mat4d_inverse_isometry(target_sensor_pose, tinv);
  /* PCS.cpp:274 */
  s3o_mat4d_mul(sinv, odometry, tmp);
  s3o_mat4d_mul(tmp, target_sensor_pose, guess);
  int st;
  if (loop) { /* PCS.cpp:286-289 */
    double coarse_result[16];
    st = s3o_align(source, n_source, stride_source, target, n_target, stride_target, guess, coarse, 0, coarse_result, info);
    if (st != S3D_STATUS_OK) return st;
    memcpy(guess, coarse_result, sizeof guess);
  }
  double icp_result[16];
  st = s3o_align(source, n_source, stride_source, target, n_target, stride_target, guess, fine, 0, icp_result, info); /* :292 */
  if (st != S3D_STATUS_OK) return st;
  s3o_mat4d_mul(source_sensor_pose, icp_result, tmp); /* :295 */
  s3o_mat4d_mul(tmp, tinv, relative_pose);
  for (int i = 0; i < 36; ++i) information[i] = 0;
  for (int i = 0; i < 6; ++i) information[i * 6 + i] = 1.0 / covariance_scale; /* :296-298 (I*scale)^-1 */
  return S3D_STATUS_OK;
}

/* ------------------------------------------------------------------ patch accumulation (B1) and map building (B2, B3)
 * the callers either side of the registration path. */

/* PointCloudSensor::transform (PCS.cpp:228-233) = pcl::transformPointCloud(in, out, tf.matrix()) with a
 * Matrix4d: pcl::detail::Transformer<double>::se3, carried in double and rounded to float once per
 * coordinate.  Summation order of the SSE2/AVX specialisation PCL compiles on x86-64:
 * (x*c0 + y*c1) + (z*c2 + c3).  tf: 4x4 column-major. */
void s3o_transform_cloud(const float* xyz, int n, int stride, const double tf[16], float* out) {
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * stride;
    const double x = p[0], y = p[1], z = p[2];
    for (int r = 0; r < 3; ++r)
      out[(size_t)i * 3 + r] = (float)((x * tf[0 * 4 + r] + y * tf[1 * 4 + r]) + (z * tf[2 * 4 + r] + tf[3 * 4 + r]));
  }
}

/* getAccumulatedCloud (PCS.cpp:235-256): every cloud transformed by its pose (= correctedPose * sensorPose,
 * formed by the caller) and appended; the reference appends in OpenMP completion order, restated here in
 * vertex order (the single-thread order).  frame != NULL adds createCombinedMeasurement (PCS.cpp:258-266):
 * the accumulated cloud — already rounded to float — transformed again by frame.inverse().
 * out: capacity 3 * sum(sizes).  Returns the number of points. */
int s3o_accumulate_clouds(const float* const* clouds, const int* sizes, const int* strides, int n_clouds,
                          const double* poses, const double* frame, float* out) {
  size_t off = 0;
  for (int c = 0; c < n_clouds; ++c) {
    s3o_transform_cloud(clouds[c], sizes[c], strides[c], poses + (size_t)c * 16, out + off * 3);
    off += (size_t)sizes[c];
  }
  if (frame) {
    double inv[16];
    s3o_mat4d_inverse_isometry(frame, inv);
    s3o_transform_cloud(out, (int)off, 3, inv, out);
  }
  return (int)off;
}

/* removeOutliers (PCS.cpp:211-226) -> pcl::RadiusOutlierRemoval::applyFilterIndices, dense-cloud branch
 * (PCL 1.12 radius_outlier_removal.hpp): k = min_neighbors + 1 nearest neighbours of the point (the point
 * itself included, FLANN float distances); the point stays iff all k exist and r*r >= d2[k-1].
 * radius <= 0 or min_neighbors == 0 or an empty cloud: the input is returned unchanged (:214).
 * out: capacity 3*n.  Returns the number of points kept, in input order. */
int s3o_remove_outliers(const float* xyz, int n, int stride, double radius, unsigned min_neighbors, float* out) {
  if (n <= 0) return 0;
  float* packed = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) packed[(size_t)i * 3 + a] = xyz[(size_t)i * stride + a];
  if (!(radius > 0) || min_neighbors == 0) {
    memcpy(out, packed, sizeof(float) * 3 * (size_t)n);
    free(packed);
    return n;
  }
  const int mean_k = (int)min_neighbors + 1;
  const double nn_dists_max = radius * radius;
  s3o_kdtree* t = s3o_kdtree_build(packed, n);
  int* idx = (int*)malloc(sizeof(int) * (size_t)mean_k);
  float* d2 = (float*)malloc(sizeof(float) * (size_t)mean_k);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const int k = s3o_kdtree_knn(t, packed + (size_t)i * 3, mean_k, idx, d2);
    int keep = 1;
    if (k == mean_k) {
      if (nn_dists_max < (double)d2[k - 1]) keep = 0;
    } else {
      keep = 0;
    }
    if (!keep) continue;
    for (int a = 0; a < 3; ++a) out[(size_t)m * 3 + a] = packed[(size_t)i * 3 + a];
    ++m;
  }
  free(idx); free(d2);
  s3o_kdtree_free(t);
  free(packed);
  return m;
}

/* buildMap (PCS.cpp:301-318): accumulate, removeOutliers(mMapOutlierRadius, mMapOutlierNeighbors),
 * downsample(mMapResolution).  out: capacity 3 * sum(sizes). */
int s3o_build_map(const float* const* clouds, const int* sizes, const int* strides, int n_clouds, const double* poses,
                  double outlier_radius, unsigned outlier_neighbors, double map_resolution, float* out) {
  size_t total = 0;
  for (int c = 0; c < n_clouds; ++c) total += (size_t)sizes[c];
  if (total == 0) return 0;
  float* accu = (float*)malloc(sizeof(float) * 3 * total);
  float* kept = (float*)malloc(sizeof(float) * 3 * total);
  const int n = s3o_accumulate_clouds(clouds, sizes, strides, n_clouds, poses, NULL, accu);
  const int m = s3o_remove_outliers(accu, n, 3, outlier_radius, outlier_neighbors, kept);
  const int r = s3o_voxel_downsample(kept, m, 3, map_resolution, out, NULL);
  free(accu); free(kept);
  return r;
}

/* ---- fillGroundPlane (PCS.cpp:362-388): pcl::RandomSampleConsensus over pcl::SampleConsensusModelPlane ----
 * Restated from PCL 1.12 (sample_consensus/ransac.hpp computeModel, sac_model.h getSamples/drawIndexSample,
 * sac_model_plane.hpp isSampleGood / computeModelCoefficients / countWithinDistance):
 *  - the model is constructed with random = false: boost::mt19937 seeded 12345, rnd() = uniform_int(0, INT_MAX)
 *    of it = engine() / 2 (boost's bucket division for a 2^31 range over a 2^32 engine);
 *  - drawIndexSample: partial Fisher-Yates on a persistent index permutation, swap(perm[i], perm[i + rnd() % (N-i)]);
 *  - a sample is bad when the component-wise ratios (p1-p0)/(p2-p0) are all equal (collinear), <= 1000 redraws;
 *  - plane = normalised (p1-p0) x (p2-p0), d = -n.p0; Eigen packet order for the 4-float sums;
 *  - an inlier has |(a x + b y) + (c z + d)| < (float)threshold (the SSE form of countWithinDistance; the
 *    scalar tail of PCL's loop sums in Eigen's order instead - a last-ulp ambiguity this restatement ignores);
 *  - k = log(1 - 0.99) / log(1 - w^3) after every improvement, stop at iterations >= k or > 1000.
 * Returns 1 and the best model, 0 when no model could be drawn (fewer than 3 points / all samples collinear). */
typedef struct { uint32_t mt[624]; int idx; } s3o_mt19937;
static void s3o_mt_seed(s3o_mt19937* g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t s3o_mt_next(s3o_mt19937* g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static int s3o_plane_from_sample(const float* xyz, int stride, const int* s, float mc[4]) {
  const float* p0 = xyz + (size_t)s[0] * stride;
  const float* p1 = xyz + (size_t)s[1] * stride;
  const float* p2 = xyz + (size_t)s[2] * stride;
  float a[3], b[3], r[3];
  for (int i = 0; i < 3; ++i) { a[i] = p1[i] - p0[i]; b[i] = p2[i] - p0[i]; r[i] = a[i] / b[i]; }
  if (r[0] == r[1] && r[2] == r[1]) return 0;
  mc[0] = a[1] * b[2] - a[2] * b[1];
  mc[1] = a[2] * b[0] - a[0] * b[2];
  mc[2] = a[0] * b[1] - a[1] * b[0];
  const float z = (mc[0] * mc[0] + mc[2] * mc[2]) + (mc[1] * mc[1] + 0.0f);
  if (z > 0.f) { const float nrm = sqrtf(z); mc[0] /= nrm; mc[1] /= nrm; mc[2] /= nrm; }
  mc[3] = -1.0f * ((mc[0] * p0[0] + mc[2] * p0[2]) + (mc[1] * p0[1] + 0.0f));
  return 1;
}
int s3o_fit_plane_ransac(const float* xyz, int n, int stride, double threshold, int max_iterations, double probability,
                         float coeffs[4], int* n_inliers, int* iterations) {
  if (n_inliers) *n_inliers = 0;
  if (iterations) *iterations = 0;
  if (n < 3) return 0;
  s3o_mt19937 g;
  s3o_mt_seed(&g, 12345u);
  int* perm = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  const float thr = (float)threshold;
  const double log_probability = log(1.0 - probability), one_over_indices = 1.0 / (double)n;
  int iters = 0, best = -2147483647, found = 0, skipped = 0;
  const int max_skip = max_iterations * 10;
  double k = 1.0;
  while ((double)iters < k && skipped < max_skip) {
    int s[3], good = 0;
    for (int check = 0; check < 1000 && !good; ++check) {
      for (int i = 0; i < 3; ++i) {
        const int j = i + (int)((s3o_mt_next(&g) >> 1) % (uint32_t)(n - i));
        const int tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
      }
      s[0] = perm[0]; s[1] = perm[1]; s[2] = perm[2];
      float r[3];
      const float* p0 = xyz + (size_t)s[0] * stride;
      const float* p1 = xyz + (size_t)s[1] * stride;
      const float* p2 = xyz + (size_t)s[2] * stride;
      for (int i = 0; i < 3; ++i) r[i] = (p1[i] - p0[i]) / (p2[i] - p0[i]);
      good = (r[0] != r[1]) || (r[2] != r[1]);
    }
    if (!good) break;
    float mc[4];
    if (!s3o_plane_from_sample(xyz, stride, s, mc)) { ++skipped; continue; }
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
      const float* p = xyz + (size_t)i * stride;
      const float v = (mc[0] * p[0] + mc[1] * p[1]) + (mc[2] * p[2] + mc[3]);
      if (fabsf(v) < thr) ++cnt;
    }
    if (cnt > best) {
      best = cnt; found = 1;
      memcpy(coeffs, mc, sizeof(float) * 4);
      const double w = (double)best * one_over_indices;
      double p_no_outliers = 1.0 - pow(w, 3.0);
      if (p_no_outliers < 2.220446049250313e-16) p_no_outliers = 2.220446049250313e-16;
      if (p_no_outliers > 1.0 - 2.220446049250313e-16) p_no_outliers = 1.0 - 2.220446049250313e-16;
      k = log_probability / log(p_no_outliers);
    }
    ++iters;
    if (iters > max_iterations) break;
  }
  free(perm);
  if (n_inliers) *n_inliers = found ? best : 0;
  if (iterations) *iterations = iters;
  return found;
}

/* The points fillGroundPlane appends (PCS.cpp:370-387): for r = res, 2 res, ... <= radius the projection of
 * (r,0,0) onto the plane, rotated about the plane normal (through the origin) in steps of res/radius.
 * Eigen::Hyperplane(normal, d).projection(p) = p - (n.p + d) n;  AngleAxis(angle, n).toRotationMatrix() (Rodrigues,
 * Eigen's evaluation order).  coeffs: the float RANSAC model.  out: capacity from s3o_fill_ground_count. */
int s3o_fill_ground_points(const float coeffs[4], double radius, double map_resolution, float* out, int cap) {
  const double n[3] = {(double)coeffs[0], (double)coeffs[1], (double)coeffs[2]}, d = (double)coeffs[3];
  const double angle_inc = map_resolution / radius;
  int m = 0;
  if (!(map_resolution > 0)) return 0;
  for (double r = map_resolution; r <= radius; r += map_resolution) {
    const double sd = (n[0] * r + n[1] * 0.0 + n[2] * 0.0) + d;
    const double sp[3] = {r - sd * n[0], 0.0 - sd * n[1], 0.0 - sd * n[2]};
    for (double angle = 0; angle < 2 * 3.141592654 /* #define PI 3.141592654, PointCloudSensor.cpp:48, :376 */; angle += angle_inc) {
      const double s = sin(angle), c = cos(angle);
      const double sa[3] = {s * n[0], s * n[1], s * n[2]};
      const double ca[3] = {(1.0 - c) * n[0], (1.0 - c) * n[1], (1.0 - c) * n[2]};
      double R[3][3], tmp;
      tmp = ca[0] * n[1]; R[0][1] = tmp - sa[2]; R[1][0] = tmp + sa[2];
      tmp = ca[0] * n[2]; R[0][2] = tmp + sa[1]; R[2][0] = tmp - sa[1];
      tmp = ca[1] * n[2]; R[1][2] = tmp - sa[0]; R[2][1] = tmp + sa[0];
      for (int i = 0; i < 3; ++i) R[i][i] = ca[i] * n[i] + c;
      if (m < cap)
        for (int i = 0; i < 3; ++i) out[(size_t)m * 3 + i] = (float)(R[i][0] * sp[0] + R[i][1] * sp[1] + R[i][2] * sp[2]);
      ++m;
    }
  }
  return m;
}

/* test hook: the first n outputs of the mt19937 above (checked against numpy's legacy RandomState in tests) */
void s3o_mt19937_outputs(unsigned seed, int n, unsigned* out) {
  s3o_mt19937 g;
  s3o_mt_seed(&g, (uint32_t)seed);
  for (int i = 0; i < n; ++i) out[i] = s3o_mt_next(&g);
}
