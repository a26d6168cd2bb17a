// s3d_core.h — per-thread math of the registration path, shared by the HIP
// kernels (s3d_kernels.hip) and by the CPU emulation harness under tests/emu/
// (g++; debugging aid only — the product library has no CPU back-end).
//
// Everything here is written MI355X-first: per-point covariances are stored as
// a unit normal (C = I - (1-eps) n n^T), the GICP objective is accumulated ONCE
// per outer iteration as a quadratic form in the 12 entries of [R|t] (73
// doubles), and the BFGS inner loop then runs on that form without touching
// the point data again.  The reference's (PCL's) per-evaluation loops over all
// correspondences are restated by the CPU test oracle; parity between the two
// formulations is what tests/ check.
//
// Compile with -ffp-contract=off: float expressions marked "order matters"
// must evaluate exactly as the oracle's; fma() is used explicitly elsewhere.
#pragma once

// candidate loads issued per step of the nearest-neighbour row scans.  Measured together with the kernel's
// register cap (S3D_NN_WAVES, s3d_kernels.h): 6 loads under a 72-VGPR cap (7 waves per SIMD) beat 8 loads at the
// compiler's free choice of 92 VGPRs (5 waves) by 8 % on the default batch; 8 waves (64 VGPRs) spill.
#ifndef S3D_NN_BATCH
#define S3D_NN_BATCH 6
#endif
#ifndef S3D_KNN_TWOPHASE
#define S3D_KNN_TWOPHASE 1
#endif
#ifndef S3D_KNN_PREFETCH
#define S3D_KNN_PREFETCH 1
#endif
#ifndef S3D_KNN3_DEPTH
#define S3D_KNN3_DEPTH 2      // candidate loads in flight per lane in the round-3 k-NN scan (2 or 4)
#endif

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define S3D_HD __host__ __device__ __forceinline__
#else
#define S3D_HD inline
#endif

namespace s3d {

// ------------------------------------------------------------------ small types

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

// 4x4 float, column-major: (r,c) = m[c*4+r]  (Eigen::Matrix4f layout)
struct Mat4f { float m[16]; };
#define S3D_M(mat, r, c) ((mat).m[(c) * 4 + (r)])

S3D_HD Mat4f mat4f_identity() {
  Mat4f I;
  for (int i = 0; i < 16; ++i) I.m[i] = (i % 5 == 0) ? 1.f : 0.f;
  return I;
}

// instrumentation hook of the CPU emulation harness (tests/emu): compiled out everywhere else
#if defined(S3D_EMU_COUNTERS)
extern long long g_s3d_counters[8];
#define S3D_COUNT(slot, n) (g_s3d_counters[slot] += (n))
#else
#define S3D_COUNT(slot, n) ((void)0)
#endif

// pcl::transformPointCloud(Matrix4f): p0 + (p1 + (p2 + c3))   [order matters]
S3D_HD F3 xf_pcl(const Mat4f& T, float x, float y, float z) {
  F3 o;
  o.x = S3D_M(T, 0, 0) * x + (S3D_M(T, 0, 1) * y + (S3D_M(T, 0, 2) * z + S3D_M(T, 0, 3)));
  o.y = S3D_M(T, 1, 0) * x + (S3D_M(T, 1, 1) * y + (S3D_M(T, 1, 2) * z + S3D_M(T, 1, 3)));
  o.z = S3D_M(T, 2, 0) * x + (S3D_M(T, 2, 1) * y + (S3D_M(T, 2, 2) * z + S3D_M(T, 2, 3)));
  return o;
}
// Eigen Matrix4f * Vector4f (w = 1): ((c0 x + c1 y) + c2 z) + c3   [order matters]
S3D_HD F3 xf_eigen(const Mat4f& T, float x, float y, float z) {
  F3 o;
  o.x = ((S3D_M(T, 0, 0) * x + S3D_M(T, 0, 1) * y) + S3D_M(T, 0, 2) * z) + S3D_M(T, 0, 3);
  o.y = ((S3D_M(T, 1, 0) * x + S3D_M(T, 1, 1) * y) + S3D_M(T, 1, 2) * z) + S3D_M(T, 1, 3);
  o.z = ((S3D_M(T, 2, 0) * x + S3D_M(T, 2, 1) * y) + S3D_M(T, 2, 2) * z) + S3D_M(T, 2, 3);
  return o;
}
// Eigen Matrix4f * Matrix4f, k = 0..3 in order   [order matters]
S3D_HD Mat4f mat4f_mul(const Mat4f& a, const Mat4f& b) {
  Mat4f t;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      S3D_M(t, r, c) = ((S3D_M(a, r, 0) * S3D_M(b, 0, c) + S3D_M(a, r, 1) * S3D_M(b, 1, c)) +
                        S3D_M(a, r, 2) * S3D_M(b, 2, c)) + S3D_M(a, r, 3) * S3D_M(b, 3, c);
  return t;
}

// FLANN L2_Simple: (dx^2 + dy^2) + dz^2 in float   [order matters]
S3D_HD float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return (dx * dx + dy * dy) + dz * dz;
}
// the same value with the x / y halves as ONE packed subtraction and ONE packed multiplication (v_pk_add_f32,
// v_pk_mul_f32: IEEE per component, so bit-identical); p: a point whose x, y sit in an even-aligned register pair,
// which is what a 16-byte load delivers
template <typename F4T>
S3D_HD float dist2_xy(float ax, float ay, float az, const F4T& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a = {ax, ay}, b = {p.x, p.y};
  v2f d = a - b;
  d = d * d;
  const float dz = az - p.z;
  return (d.x + d.y) + dz * dz;
#else
  return dist2(ax, ay, az, p.x, p.y, p.z);
#endif
}

// pcl::transformPointCloud with a Matrix4d (pcl::detail::Transformer<double>::se3): the product is carried in
// double and rounded to float once per coordinate, (x c0 + y c1) + (z c2 + c3).  T: 3x4 ROW-major.
S3D_HD F3 xf_pcl_d(const double* T, float xf, float yf, float zf) {
  const double x = xf, y = yf, z = zf;
  F3 o;
  o.x = (float)((x * T[0] + y * T[1]) + (z * T[2] + T[3]));
  o.y = (float)((x * T[4] + y * T[5]) + (z * T[6] + T[7]));
  o.z = (float)((x * T[8] + y * T[9]) + (z * T[10] + T[11]));
  return o;
}

// Square root for CONSERVATIVE bounds only - how far along x a ball still reaches into a row of cells - where the caller
// widens the result by 1e-4 relative plus a rounding margin anyway: the bare v_sqrt_f32 (1 ulp) instead of the correctly
// rounded library sequence, which is 18 instructions around that same instruction (scaling for denormals, two Newton
// corrections, class fix-ups) and was a quarter of the per-row cost of every pruned scan.  A clipped x-range that
// differs by one cell at a boundary changes what is examined, never what is found (the dropped cell lies beyond the
// ball).  The host (emulation, oracle-side tests) keeps sqrtf.
S3D_HD float sqrt_bound(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}

S3D_HD bool lex_less(float d2a, int ia, float d2b, int ib) { return d2a < d2b || (d2a == d2b && ia < ib); }

S3D_HD int imin(int a, int b) { return a < b ? a : b; }
S3D_HD int imax(int a, int b) { return a > b ? a : b; }

// ------------------------------------------------------------------ voxel grid (K1)
// pcl::VoxelGrid key of one point (voxel_grid.hpp applyFilter first pass)

struct VoxelParams {
  float inv_leaf;
  int   min_b[3];
  int   div_b[3];
  int   passthrough;  // dx*dy*dz > INT_MAX: every point is its own voxel (output = input)
};

S3D_HD VoxelParams voxel_params_from_bbox(const float mn[3], const float mx[3], float leaf) {
  VoxelParams vp;
  vp.inv_leaf = 1.0f / leaf;
  int64_t d[3];
  // PCL's conversions (static_cast<std::int64_t> / static_cast<int> of float products) are undefined beyond the
  // integer's range - a leaf tens of orders of magnitude below the cloud's extent.  Each d is >= 1, so ONE axis beyond
  // INT_MAX already means dx*dy*dz > INT_MAX: such a cloud is PCL's "leaf size is too small" case (passthrough)
  // without the out-of-range conversion ever being made; wherever PCL's arithmetic is defined this is PCL's result.
  bool over = false;
  for (int a = 0; a < 3; ++a) {
    const float e = (mx[a] - mn[a]) * vp.inv_leaf, lo = floorf(mn[a] * vp.inv_leaf), hi = floorf(mx[a] * vp.inv_leaf);
    const bool in_range = e < 2147483648.0f && fabsf(lo) < 2147483648.0f && fabsf(hi) < 2147483648.0f;   // (false for NaN)
    over = over || !in_range;
    d[a] = in_range ? (int64_t)e + 1 : 1;
    vp.min_b[a] = in_range ? (int)lo : 0;
    const int max_b = in_range ? (int)hi : 0;
    vp.div_b[a] = max_b - vp.min_b[a] + 1;
  }
  const int64_t d01 = d[0] * d[1];                                   // (<= 2^62)
  over = over || d01 > (int64_t)2147483647 || d01 * d[2] > (int64_t)2147483647;
  vp.passthrough = over ? 1 : 0;
  return vp;
}

S3D_HD uint32_t voxel_key(const VoxelParams& vp, float x, float y, float z) {
  int i0 = (int)(floorf(x * vp.inv_leaf) - (float)vp.min_b[0]);
  int i1 = (int)(floorf(y * vp.inv_leaf) - (float)vp.min_b[1]);
  int i2 = (int)(floorf(z * vp.inv_leaf) - (float)vp.min_b[2]);
  return (uint32_t)(i0 + i1 * vp.div_b[0] + i2 * vp.div_b[0] * vp.div_b[1]);
}

// ------------------------------------------------------------------ search grid (K3)
// Dense uniform grid over one cloud; points are stored cell-sorted (x fastest)
// so that the cells (ix-r .. ix+r, iy, iz) of one row are ONE contiguous run.

struct GridParams {
  float origin[3];
  float h, inv_h;
  int   dim[3];
  int   ncells;
};

// h0: wanted cell edge; the edge is grown (x 2^(1/6)) until the cell count fits `cap`
S3D_HD GridParams grid_params_from_bbox(const float mn[3], const float mx[3], float h0, int cap) {
  GridParams g;
  float h = h0;
  // (the count is compared axis by axis: every dim is >= 1, so a partial product beyond the budget decides - and an
  // edge so small that an axis alone has more cells than an int holds never reaches the int conversion.  400 steps of
  // 2^(1/6) cover 20 orders of magnitude; a wanted edge farther than that below the cloud's extent goes on by doubling)
  bool fits = false;
  for (int it = 0; it < 700 && !fits; ++it) {
    if (it > 0) h *= it <= 400 ? 1.1224620f : 2.0f;   // 2^(1/6): the cell count lands within sqrt(2) of the budget
    int64_t nc = 1;
    fits = h > 0.f;
    for (int a = 0; a < 3; ++a) {
      const float q = floorf((mx[a] - mn[a]) / h);
      if (!(q < 2.0e9f)) { fits = false; g.dim[a] = 1; continue; }      // (also NaN)
      g.dim[a] = (int)q + 1;
      if (fits) { nc *= g.dim[a]; fits = nc <= (int64_t)cap; }
    }
  }
  if (!fits) {   // no finite edge serves (an extent beyond float's range): one cell
    h = 3.0e38f;
    g.dim[0] = g.dim[1] = g.dim[2] = 1;
  }
  g.h = h;
  g.inv_h = 1.0f / h;
  for (int a = 0; a < 3; ++a) g.origin[a] = mn[a];
  g.ncells = g.dim[0] * g.dim[1] * g.dim[2];
  return g;
}

S3D_HD int grid_coord(const GridParams& g, int axis, float v) {
  // clamp in float first: a far-away query must not overflow the int conversion
  float f = floorf((v - g.origin[axis]) * g.inv_h);
  f = fminf(fmaxf(f, -1.0e6f), 1.0e6f);
  return (int)f;
}
S3D_HD int grid_cell_of_point(const GridParams& g, float x, float y, float z) {
  int ix = imin(imax(grid_coord(g, 0, x), 0), g.dim[0] - 1);
  int iy = imin(imax(grid_coord(g, 1, y), 0), g.dim[1] - 1);
  int iz = imin(imax(grid_coord(g, 2, z), 0), g.dim[2] - 1);
  return ix + g.dim[0] * (iy + g.dim[1] * iz);
}

// ------------------------------------------------------------------ K2 + K3 in one sort (round 5): the fused pre-pass
// pcl::VoxelGrid sorts the points by the voxel key (iz, iy, ix); the search grid then sorted the centroids again by the
// cell id (cz, cy, cx).  Both orders are z-major / x-fastest: with the search cell an INTEGER number m of voxels per
// edge and the grid's origin on the voxel lattice, a voxel lies in exactly one cell and ONE radix sort on the
// mixed-radix key
//     key = cell * msub + sub,   cell = cx + dim0 (cy + dim1 cz),   sub = sx + sub0 (sy + sub1 sz),
//     (cx, sx) = (ix / m, ix mod m) ...,   sub_a = min(m, div_b[a]),   msub = sub0 sub1 sub2
// leaves the raw points in cell order with voxel order inside a cell - the order the two sorts produced.  The centroid
// kernel then writes the cell-sorted arrays and the cell table directly: no second key pass, no second sort, no gather
// pass.  Tie-breaking id of a centroid (the `.w` of a cell-sorted point): PCL's voxel key, which orders the centroids as
// their index in pcl::VoxelGrid's output does.
// The grid is laid over the VOXEL lattice of the raw cloud's bounding box (known before the sort), not over the box of
// the centroids: m is the smallest edge >= 2 voxels (the old h0 = 2 leaf) whose cell count fits the budget.
// A slot the scheme cannot serve - PCL's own index overflow (passthrough), a mixed key beyond 32 bits, a centroid whose
// float sums put it outside its cell by more than the searches' rounding margin - says so in FusedGrid::ok, and the
// host runs the batch again on the two-sort path (s3d_api.hip Batch::run_all): never silently inexact.
struct FusedGrid {
  int m;            // cell edge in voxels
  int sub[3];       // sub-voxel radix per axis
  uint32_t msub;    // sub[0] * sub[1] * sub[2]
  int ok;           // 1: served; 0: not attempted; < 0: this slot cannot use the fused path (see above)
  unsigned long long msub_magic;   // ceil(2^64 / msub) (msub >= 2): key / msub = high 64 bits of key * magic, exactly
};
constexpr float kFusedCellTol = 1.0e-3f;   // a stored point may lie this far (in cells) outside its cell's box: half of
                                           // the 2e-3 margin every search adds for the rounding of cell assignments

S3D_HD bool fused_grid_from_voxels(const VoxelParams& vp, int cap, GridParams& g, FusedGrid& f) {
  f.m = 1; f.sub[0] = f.sub[1] = f.sub[2] = 1; f.msub = 1; f.ok = -1; f.msub_magic = 0ull;
  g.h = 1.f; g.inv_h = 1.f; g.ncells = 1;
  for (int a = 0; a < 3; ++a) { g.origin[a] = 0.f; g.dim[a] = 1; }
  if (vp.passthrough || vp.div_b[0] < 1 || vp.div_b[1] < 1 || vp.div_b[2] < 1) return false;
  const double vol = (double)vp.div_b[0] * (double)vp.div_b[1] * (double)vp.div_b[2];
  if (vol > 2147483647.0) return false;   // (the voxel keys are the tie-breaking ids: non-negative ints)
  // the cell count prod ceil(div_b / m) does not grow with m: the smallest m >= 2 that fits the budget by bisection
  auto cells = [&](int mm) {
    int64_t nc_ = 1;
    for (int a = 0; a < 3; ++a) nc_ *= (int64_t)((vp.div_b[a] + mm - 1) / mm);
    return nc_;
  };
  const int64_t budget = cap > 0 ? (int64_t)cap : 1;
  int lo = 2, hi = 2;                        // invariant below: cells(hi) <= budget, cells(lo - 1) > budget or lo == 2
  while (cells(hi) > budget) {
    if (hi >= (1 << 30)) return false;
    lo = hi + 1;
    hi *= 2;
  }
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (cells(mid) <= budget) hi = mid; else lo = mid + 1;
  }
  const int m = hi;
  const int64_t nc = cells(m);
  uint64_t msub = 1;
  for (int a = 0; a < 3; ++a) {
    g.dim[a] = (vp.div_b[a] + m - 1) / m;
    f.sub[a] = vp.div_b[a] < m ? vp.div_b[a] : m;
    msub *= (uint64_t)f.sub[a];
  }
  if (msub > 0xFFFFFFFFull || (uint64_t)nc * msub > 0xFFFFFFFEull) return false;   // (0xFFFFFFFF = kInvalidKey)
  f.m = m; f.msub = (uint32_t)msub;
  f.msub_magic = msub >= 2 ? (0xFFFFFFFFFFFFFFFFull / msub) + 1ull : 0ull;
  g.ncells = (int)nc;
  // voxel i of axis a covers [(min_b + i) / inv_leaf, (min_b + i + 1) / inv_leaf) up to the rounding of PCL's float product
  g.h = (float)((double)m / (double)vp.inv_leaf);
  g.inv_h = 1.0f / g.h;
  for (int a = 0; a < 3; ++a) g.origin[a] = (float)((double)vp.min_b[a] / (double)vp.inv_leaf);
  f.ok = 1;
  return true;
}

// the key of one raw point (the float operations of voxel_key, i.e. of pcl::VoxelGrid)
S3D_HD uint32_t fused_key(const VoxelParams& vp, const GridParams& g, const FusedGrid& f, float x, float y, float z) {
  const int i0 = (int)(floorf(x * vp.inv_leaf) - (float)vp.min_b[0]);
  const int i1 = (int)(floorf(y * vp.inv_leaf) - (float)vp.min_b[1]);
  const int i2 = (int)(floorf(z * vp.inv_leaf) - (float)vp.min_b[2]);
  const int c0 = i0 / f.m, c1 = i1 / f.m, c2 = i2 / f.m;
  const uint32_t cell = (uint32_t)(c0 + g.dim[0] * (c1 + g.dim[1] * c2));
  const uint32_t sub = (uint32_t)((i0 - c0 * f.m) + f.sub[0] * ((i1 - c1 * f.m) + f.sub[1] * (i2 - c2 * f.m)));
  return cell * f.msub + sub;
}

// key -> cell id without a division: for key < 2^32 and msub < 2^32 the high 64 bits of key * ceil(2^64 / msub) are
// floor(key / msub) exactly (the error term key * e / (msub 2^64), e < msub, is below 2^-32 < 1 / msub)
S3D_HD uint32_t fused_cell_of_key(const FusedGrid& f, uint32_t key) {
  if (f.msub < 2u) return key;
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__umul64hi((unsigned long long)key, f.msub_magic);
#else
  return (uint32_t)(((unsigned __int128)key * (unsigned __int128)f.msub_magic) >> 64);
#endif
}

// PCL's voxel key (the tie-breaking id) of a voxel from one of its raw points: pcl::VoxelGrid's own float operations
S3D_HD uint32_t fused_voxel_of_point(const VoxelParams& vp, float x, float y, float z) { return voxel_key(vp, x, y, z); }

// does the point lie in cell `cell` as the searches see it?  Fast path: the cell its position falls into IS the cell;
// otherwise (a centroid within rounding of a cell face, or outside its voxel by the rounding of its float sums) the
// tolerance test against the decoded cell coordinates
S3D_HD bool fused_point_in_cell(const VoxelParams& vp, const GridParams& g, const FusedGrid& f, uint32_t key, int cell,
                                float x, float y, float z);

// key -> cell id, cell coordinates and PCL's voxel key (the tie-breaking id)
S3D_HD void fused_decode(const VoxelParams& vp, const GridParams& g, const FusedGrid& f, uint32_t key, int* cell, int c[3],
                         uint32_t* voxel) {
  const uint32_t ce = key / f.msub;
  uint32_t su = key - ce * f.msub;
  const int s0 = (int)(su % (uint32_t)f.sub[0]); su /= (uint32_t)f.sub[0];
  const int s1 = (int)(su % (uint32_t)f.sub[1]);
  const int s2 = (int)(su / (uint32_t)f.sub[1]);
  uint32_t r = ce;
  c[0] = (int)(r % (uint32_t)g.dim[0]); r /= (uint32_t)g.dim[0];
  c[1] = (int)(r % (uint32_t)g.dim[1]);
  c[2] = (int)(r / (uint32_t)g.dim[1]);
  *cell = (int)ce;
  *voxel = (uint32_t)((c[0] * f.m + s0) + (c[1] * f.m + s1) * vp.div_b[0] + (c[2] * f.m + s2) * vp.div_b[0] * vp.div_b[1]);
}

// does the point lie in the box of cell c (+- kFusedCellTol cells)?  What the searches assume of a stored point.
S3D_HD bool fused_inside(const GridParams& g, const int c[3], float x, float y, float z) {
  const float p[3] = {x, y, z};
  bool ok = true;
  for (int a = 0; a < 3; ++a) {
    const float fcell = (p[a] - g.origin[a]) * g.inv_h - (float)c[a];
    ok = ok && fcell >= -kFusedCellTol && fcell <= 1.0f + kFusedCellTol;
  }
  return ok;
}

S3D_HD bool fused_point_in_cell(const VoxelParams& vp, const GridParams& g, const FusedGrid& f, uint32_t key, int cell,
                                float x, float y, float z) {
  const int ix = grid_coord(g, 0, x), iy = grid_coord(g, 1, y), iz = grid_coord(g, 2, z);
  if (ix >= 0 && ix < g.dim[0] && iy >= 0 && iy < g.dim[1] && iz >= 0 && iz < g.dim[2] &&
      ix + g.dim[0] * (iy + g.dim[1] * iz) == cell)
    return true;
  int c2, cc[3];
  uint32_t voxel;
  fused_decode(vp, g, f, key, &c2, cc, &voxel);
  return fused_inside(g, cc, x, y, z);
}

// pcl::RadiusOutlierRemoval on the search grid: how many points p of the cloud (the query itself included)
// have float d2(q, p) <= r2f; the scan stops as soon as `need` are found.  reach >= sqrt(r2f) * (1 + 1e-5): the
// box of cells that can hold such a point.
template <typename F4T>
S3D_HD int grid_radius_count(const GridParams& g, const uint32_t* __restrict__ cell_start, const F4T* __restrict__ pts,
                             float qx, float qy, float qz, float reach, float r2f, int need) {
  const int x0 = imin(imax(grid_coord(g, 0, qx - reach), 0), g.dim[0] - 1);
  const int x1 = imin(imax(grid_coord(g, 0, qx + reach), 0), g.dim[0] - 1);
  const int y0 = imin(imax(grid_coord(g, 1, qy - reach), 0), g.dim[1] - 1);
  const int y1 = imin(imax(grid_coord(g, 1, qy + reach), 0), g.dim[1] - 1);
  const int z0 = imin(imax(grid_coord(g, 2, qz - reach), 0), g.dim[2] - 1);
  const int z1 = imin(imax(grid_coord(g, 2, qz + reach), 0), g.dim[2] - 1);
  int count = 0;
  for (int z = z0; z <= z1; ++z)
    for (int y = y0; y <= y1; ++y) {
      const int row = g.dim[0] * (y + g.dim[1] * z);
      const int b = (int)cell_start[row + x0], e = (int)cell_start[row + x1 + 1];
      for (int j = b; j < e; j += 4) {      // four loads in flight per step
        F4T pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pp[u] = pts[j + u < e ? j + u : e - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u < e && dist2(qx, qy, qz, pp[u].x, pp[u].y, pp[u].z) <= r2f && ++count >= need) return count;
      }
    }
  return count;
}

// Exact 1-NN by ring expansion.  pts: cell-sorted float4 (xyz, w = bit-cast index
// in the un-sorted cloud); cell_start: ncells+1 entries.  Ties: lowest index.
// max_d: only neighbours closer than this matter (search stops beyond it).
struct NNResult {
  int idx; float d2; int pos;
  float second_d2;   // smallest d2 among the OTHER candidates examined (3e38 if none)
  float radius;      // grid_nn1_box: every point within this distance of the query was examined
};

template <typename F4T>
S3D_HD NNResult grid_nn1(const GridParams& g, const uint32_t* __restrict__ cell_start,
                         const F4T* __restrict__ pts, float qx, float qy, float qz, float max_d) {
  NNResult best;
  best.idx = -1; best.d2 = 3.0e38f; best.pos = -1; best.second_d2 = 3.0e38f; best.radius = 0.f;
  const float fx = (qx - g.origin[0]) * g.inv_h, fy = (qy - g.origin[1]) * g.inv_h,
              fz = (qz - g.origin[2]) * g.inv_h;
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  // distance (in cells) from the query to the nearest face of its own cell, shrunk by a
  // safety margin that covers float rounding of cell assignment and of the bound itself
  float ox = fx - (float)ix, oy = fy - (float)iy, oz = fz - (float)iz;
  float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
  face = fmaxf(face - 2.0e-3f, 0.f);
  const int rmax = (int)ceilf(max_d * g.inv_h) + 1;
  // rings that cannot touch the grid are skipped at once
  int r0 = 0;
  r0 = imax(r0, imax(-ix, ix - (g.dim[0] - 1)));
  r0 = imax(r0, imax(-iy, iy - (g.dim[1] - 1)));
  r0 = imax(r0, imax(-iz, iz - (g.dim[2] - 1)));
  for (int r = r0; r <= rmax; ++r) {
    const int z0 = imax(iz - r, 0), z1 = imin(iz + r, g.dim[2] - 1);
    const int y0 = imax(iy - r, 0), y1 = imin(iy + r, g.dim[1] - 1);
    const int xl = ix - r, xh = ix + r;
    for (int cz = z0; cz <= z1; ++cz) {
      const bool zface = (cz == iz - r) || (cz == iz + r);
      for (int cy = y0; cy <= y1; ++cy) {
        const bool full = zface || (cy == iy - r) || (cy == iy + r);
        const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
        // shell cells of this row: the whole run [xl, xh] on a face, else the two end cells
        for (int part = 0; part < 2; ++part) {
          int xa, xb;
          if (full) {
            if (part) break;
            xa = imax(xl, 0); xb = imin(xh, g.dim[0] - 1);
          } else {
            if (r == 0) { if (part) break; xa = xb = ix; }
            else { xa = xb = part ? xh : xl; }
            if (xa < 0 || xa >= g.dim[0]) continue;
          }
          if (xa > xb) continue;
          const uint32_t s = cell_start[rowbase + xa], e = cell_start[rowbase + xb + 1];
          for (uint32_t k = s; k < e; ++k) {
            const F4T p = pts[k];
            const float d2 = dist2(qx, qy, qz, p.x, p.y, p.z);
            const int pi = __builtin_bit_cast(int, p.w);
            if (lex_less(d2, pi, best.d2, best.idx < 0 ? 2147483647 : best.idx)) {
              best.d2 = d2; best.idx = pi; best.pos = (int)k;
            }
          }
        }
      }
    }
    // every point outside the cube of rings <= r is farther than (r + face) * h
    const float bound = ((float)r + face) * g.h;
    if (best.idx >= 0 && best.d2 <= bound * bound) break;
    if (bound > max_d) break;
  }
  return best;
}

// Exact 1-NN by BOX search with a radius hint (the kernel's hot variant).
// All points within distance d of the query lie in the cells that intersect the axis-aligned
// box [q-d, q+d]; per (y,z) row those cells are one contiguous run.  Scan the box for the hint
// radius; if the best candidate found is provably the nearest (best <= d) stop, otherwise the
// nearest neighbour is at most as far as that candidate: rescan with exactly that radius (or
// double the radius when nothing was found).  With the previous ICP iteration's distance as the
// hint almost every query resolves in one tight scan.  Same result as grid_nn1 (ties: lowest index).
// v_min_f64 / v_max_f64 on packed keys held as doubles (every key here is a normal, finite double)
S3D_HD double f64_min_raw(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? a : b;
#endif
}
S3D_HD double f64_max_raw(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return a < b ? b : a;
#endif
}

// FAST (the first pass of a registration, which needs neither the runner-up nor a re-scan of the incumbent): the best
// candidate is ONE packed key, (d2 bits + 2^23) << 32 | index, held as a double - v_min_f64 keeps the smaller, the
// same compare moves the position along: 4 instructions after the distance instead of ~15.  Same order as lex_less
// (distance, then index); best.d2 / best.idx are refreshed from the key by nn1_fast_sync where the search reads them.
constexpr unsigned long long kNNFastNone = 0x7FDFFFFFFFFFFFFFull;   // (low word = -1: "no index")
S3D_HD void nn1_fast_sync(NNResult& best, double bkey) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, bkey);
  best.d2 = __builtin_bit_cast(float, (uint32_t)(b >> 32) - 0x00800000u);
  best.idx = (int)(uint32_t)(b & 0xFFFFFFFFull);
}
template <bool FAST = false, typename F4T>
S3D_HD void nn1_consider(NNResult& best, double& bkey, const F4T& p, uint32_t k, float qx, float qy, float qz) {
  const float d2 = FAST ? dist2_xy(qx, qy, qz, p) : dist2(qx, qy, qz, p.x, p.y, p.z);   // (the same value, see dist2_xy)
  if (FAST) {
    const unsigned long long kb = ((unsigned long long)(__builtin_bit_cast(uint32_t, d2) + 0x00800000u) << 32) |
                                  (unsigned long long)__builtin_bit_cast(uint32_t, p.w);
    const double c = __builtin_bit_cast(double, kb);
    best.pos = c < bkey ? (int)k : best.pos;
    bkey = f64_min_raw(bkey, c);
    return;
  }
  const int pi = __builtin_bit_cast(int, p.w);
  if (pi == best.idx) return;  // (a rescan meets the incumbent again)
  if (lex_less(d2, pi, best.d2, best.idx < 0 ? 2147483647 : best.idx)) {
    best.second_d2 = fminf(best.second_d2, best.d2);
    best.d2 = d2; best.idx = pi; best.pos = (int)k;
  } else {
    best.second_d2 = fminf(best.second_d2, d2);
  }
}

// scan the (<= NR x NR) rows of a tight box with the memory accesses issued as three independent
// BATCHES (all row ranges, then the first point of every row, then the rest) instead of one dependent
// load per step: the kernel is bound by memory latency x chain length and by the L1 access rate.
template <int NR, bool FAST = false, typename F4T>
S3D_HD void nn1_scan_rows(NNResult& best, double& bkey, const GridParams& g, const uint32_t* __restrict__ cell_start,
                          const F4T* __restrict__ pts, float qx, float qy, float qz, int x0, int x1, int y0, int ny,
                          int z0, int nz) {
  uint32_t rs[NR * NR], re[NR * NR];
#pragma unroll
  for (int r = 0; r < NR * NR; ++r) {
    const int jy = r % NR, jz = r / NR;
    const bool ok = jy < ny && jz < nz;
    const int rowbase = g.dim[0] * ((y0 + (ok ? jy : 0)) + g.dim[1] * (z0 + (ok ? jz : 0)));
    const uint32_t a = cell_start[rowbase + x0], b = cell_start[rowbase + x1 + 1];
    rs[r] = a; re[r] = ok ? b : a;
  }
  F4T first[NR * NR];
#pragma unroll
  for (int r = 0; r < NR * NR; ++r) first[r] = pts[rs[r] < re[r] ? rs[r] : 0];
#pragma unroll
  for (int r = 0; r < NR * NR; ++r)
    if (rs[r] < re[r]) nn1_consider<FAST>(best, bkey, first[r], rs[r], qx, qy, qz);
#pragma unroll
  for (int r = 0; r < NR * NR; ++r) {
    for (uint32_t k = rs[r] + 1; k < re[r]; k += S3D_NN_BATCH) {      // S3D_NN_BATCH loads in flight per step
      const uint32_t e = re[r], last = e - 1;
      F4T pp[S3D_NN_BATCH];
#pragma unroll
      for (int u = 0; u < S3D_NN_BATCH; ++u) pp[u] = pts[k + u < e ? k + u : last];
#pragma unroll
      for (int u = 0; u < S3D_NN_BATCH; ++u)
        if (k + u < e) nn1_consider<FAST>(best, bkey, pp[u], k + u, qx, qy, qz);
    }
  }
  if (FAST) nn1_fast_sync(best, bkey);
}

// seed_pos >= 0: position (in pts) of a point known to be near the query — the previous ICP
// iteration's neighbour.  Its distance is then an exact upper bound of the NN distance: the box of
// that radius is scanned once and the result is proven (the seed itself lies in the box).
constexpr float kNNRevalSlack = 0.25f;   // in cells

// seed_trusted: the seed is (very probably) still the nearest neighbour although it is far away — a query in a
// part of the scan the other cloud does not cover, re-searched after a small transform update.  The box is then
// sized by the seed's distance without the one-cell cap, and a wide scan prunes against best + shell instead of
// best, so that the examined radius exceeds the neighbour's distance and the NEXT pass can re-validate the
// correspondence without a search (nn_still_nearest).  Without it every such query walks its whole ball again in
// every ICP iteration: on the reference's fixture scans those 2 % of the queries were 75 % of the step time.
template <bool FAST = false, typename F4T>
S3D_HD NNResult grid_nn1_box(const GridParams& g, const uint32_t* __restrict__ cell_start,
                             const F4T* __restrict__ pts, float qx, float qy, float qz, float max_d, float d_hint,
                             int seed_pos = -1, bool seed_trusted = false, float seed_d2 = -1.f) {
  // seed_d2 >= 0 (with seed_pos < 0): the squared distance of a point KNOWN to exist, e.g. the previous neighbour seen
  // through the copy that travels with the correspondence - it sizes the first box exactly as a seed does, without
  // the load of the seed itself (the scan meets the point anyway): one link less in a chain of dependent loads
  NNResult best;
  best.idx = -1; best.d2 = 3.0e38f; best.pos = -1; best.second_d2 = 3.0e38f; best.radius = 0.f;
  double bkey = __builtin_bit_cast(double, kNNFastNone);   // (FAST only)
  // the last attempt looks one shell beyond max_d: a query with NO neighbour in range then carries a radius
  // > max_d, which is what lets the next pass prove "still none" without a search
  const float cap = max_d + kNNRevalSlack * g.h;
  const float shell = seed_trusted ? kNNRevalSlack * g.h : 0.f;
  float d = fminf(fmaxf(d_hint, 0.25f * g.h), cap);
  if (seed_pos >= 0) {
    nn1_consider<FAST>(best, bkey, pts[seed_pos], (uint32_t)seed_pos, qx, qy, qz);
    if (FAST) nn1_fast_sync(best, bkey);
    // examine a little more than the seed's distance: the extra shell is what later lets
    // nn_still_nearest() prove the correspondence without a search (lower bound of the other points).
    // (an untrusted seed that ended up far away — the transform just moved — must not blow the box up)
    d = sqrtf(best.d2) * 1.0001f + 1.0e-6f + kNNRevalSlack * g.h;
    d = fminf(seed_trusted ? d : fminf(d, g.h), cap);
  } else if (seed_d2 >= 0.f) {
    d = sqrtf(seed_d2) * 1.0001f + 1.0e-6f + kNNRevalSlack * g.h;
    d = fminf(seed_trusted ? d : fminf(d, g.h), cap);
  }
  for (int attempt = 0; attempt < 64; ++attempt) {
    S3D_COUNT(3, 1);
    // margin: float rounding of the cell assignment of the points and of the box corners
    const float m = d * 1.0001f + 2.0e-3f * g.h;
    const int x0 = imax(grid_coord(g, 0, qx - m), 0), x1 = imin(grid_coord(g, 0, qx + m), g.dim[0] - 1);
    const int y0 = imax(grid_coord(g, 1, qy - m), 0), y1 = imin(grid_coord(g, 1, qy + m), g.dim[1] - 1);
    const int z0 = imax(grid_coord(g, 2, qz - m), 0), z1 = imin(grid_coord(g, 2, qz + m), g.dim[2] - 1);
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    bool pruned = false;
    if (x0 <= x1 && ny > 0 && nz > 0) {
      if (ny <= 2 && nz <= 2) {
        nn1_scan_rows<2, FAST>(best, bkey, g, cell_start, pts, qx, qy, qz, x0, x1, y0, ny, z0, nz);
      } else if (ny <= 3 && nz <= 3) {
        nn1_scan_rows<3, FAST>(best, bkey, g, cell_start, pts, qx, qy, qz, x0, x1, y0, ny, z0, nz);
      } else {
        // wide box (badly aligned clouds, first passes): shrinking-ball scan.  Rows are visited from the
        // query's own row outwards; a row whose slab is farther than the best distance so far (+ shell) is
        // skipped, and inside a row only the cells within the remaining radius are read.  (Bounds are
        // shrunk by 2e-3 cell; ties are never pruned: rows are skipped on strictly-greater only.)
        const int cy0 = imin(imax(grid_coord(g, 1, qy), y0), y1), cz0 = imin(imax(grid_coord(g, 2, qz), z0), z1);
        const float eps = 2.0e-3f * g.h;
        pruned = true;
        // squared pruning limit: the best distance itself, or (best + shell)^2 rounded up
        auto limit2 = [&]() {
          if (shell <= 0.f) return best.d2;
          const float l = sqrtf(best.d2) * 1.0001f + shell;
          return best.d2 < 1.0e30f ? l * l : 3.0e38f;
        };
        float lim2 = limit2();
        // (layers / rows are visited centre-out, alternating sides: once BOTH sides have produced a slab beyond
        // the limit, every later one is beyond it too and the loop ends)
        int zfar = 0;
        for (int oz = 0; oz <= 2 * (z1 - z0) + 1 && zfar != 3; ++oz) {
          const int cz = cz0 + ((oz & 1) ? -((oz + 1) >> 1) : (oz >> 1));
          if (cz < z0 || cz > z1) { zfar |= (oz & 1) ? 1 : 2; continue; }
          const float zlo = g.origin[2] + (float)cz * g.h, zhi = zlo + g.h;
          const float dz = fmaxf(fmaxf(zlo - qz, qz - zhi) - eps, 0.f);
          if (dz * dz > lim2) { zfar |= (oz & 1) ? 1 : 2; continue; }
          int yfar = 0;
          for (int oy = 0; oy <= 2 * (y1 - y0) + 1 && yfar != 3; ++oy) {
            const int cy = cy0 + ((oy & 1) ? -((oy + 1) >> 1) : (oy >> 1));
            if (cy < y0 || cy > y1) { yfar |= (oy & 1) ? 1 : 2; continue; }
            const float ylo = g.origin[1] + (float)cy * g.h, yhi = ylo + g.h;
            const float dy = fmaxf(fmaxf(ylo - qy, qy - yhi) - eps, 0.f);
            const float rowd2 = dy * dy + dz * dz;
            if (rowd2 > lim2) { yfar |= (oy & 1) ? 1 : 2; continue; }
            int xa = x0, xb = x1;
            if (best.idx >= 0) {   // only the cells within the remaining radius
              const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
              xa = imax(x0, grid_coord(g, 0, qx - rx));
              xb = imin(x1, grid_coord(g, 0, qx + rx));
              if (xa > xb) continue;
            }
            const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
            const uint32_t s = cell_start[rowbase + xa], e = cell_start[rowbase + xb + 1];
            S3D_COUNT(0, 1); S3D_COUNT(1, s == e); S3D_COUNT(2, (long long)(e - s));
            for (uint32_t k = s; k < e; k += S3D_NN_BATCH) {      // S3D_NN_BATCH loads in flight per step (a row holds ~13 points here)
              const uint32_t last = e - 1;
              F4T pp[S3D_NN_BATCH];
#pragma unroll
              for (int u = 0; u < S3D_NN_BATCH; ++u) pp[u] = pts[k + u < e ? k + u : last];
#pragma unroll
              for (int u = 0; u < S3D_NN_BATCH; ++u)
                if (k + u < e) nn1_consider<FAST>(best, bkey, pp[u], k + u, qx, qy, qz);
            }
            if (FAST) nn1_fast_sync(best, bkey);
            lim2 = limit2();
          }
        }
      }
    }
    // everything within `radius` of the query has been examined.  After a pruned scan that is the best
    // distance (+ the shell of a trusted seed, shaved by the rounding guard of limit2): without a shell the
    // re-validation fails once and the next, seeded search re-establishes a proper bound.
    best.radius = (pruned && best.idx >= 0) ? fminf(d, sqrtf(best.d2) + 0.999f * shell) : d;
    if (best.idx >= 0 && best.d2 <= d * d) break;  // nothing outside the box can be closer
    if (d >= cap) break;                           // neighbours beyond max_d (+ shell) do not matter
    d = best.idx >= 0 ? fminf(sqrtf(best.d2) * 1.0001f + 1.0e-6f, cap) : fminf(2.0f * d, cap);
  }
  return best;
}

// ---- temporal coherence of ICP correspondences, with a proof ----
// After a search at query position q_old we know the neighbour p* and a lower bound L on the distance
// of EVERY OTHER point to q_old (the smaller of the runner-up's distance and the examined radius).
// When the query moves to q_new, every other point is still farther than L - |q_new - q_old|
// (triangle inequality), so p* is certainly still the unique nearest neighbour if
//     |p* - q_new| + |q_new - q_old|  <  L (1 - 1e-5) - 1e-6.
// The inequality is exact for the float-valued positions; the margin covers the rounding of the
// three computed distances (coordinate differences are exact or 6e-8-relative, d2 and sqrt add
// ~3e-7 relative: 30x below the margin).
// Once ICP has converged the queries barely move and almost every correspondence is re-validated
// without touching the grid; the result is the same point and the same float d2 as a full search.
S3D_HD float nn_lower_bound_others(const NNResult& r) {
  return fminf(sqrtf(r.second_d2), r.radius);
}
S3D_HD bool nn_still_nearest(float d_new, float move, float lb_others) {
  return d_new + move < lb_others * 0.99999f - 1.0e-6f;
}

// ---- the same proof for 64 queries at once (round 4: the settled passes of a registration) ----
// A RECORD of 64 consecutive queries of a pair (cell order: spatial neighbours) keeps the box of their positions
// (centre c, half extents e: the guess-transformed points, which do not change during a registration), the pass of
// its last full evaluation (`touch`: the bounds corr_lb of its queries are relative to the positions under THAT
// pass's transformation_) and the smallest MARGIN of its queries at that pass,
//     margin_i = (lb_i (1 - 1e-5) - 1e-6 - d_i) / 2      (a neighbour at distance d_i, every other point beyond lb_i)
//     margin_i =  lb_i (1 - 1e-5) - 1e-6 - max_d         (no point within lb_i > max_d).
// Under the current transformation_ T a query has moved by |fl(T p) - fl(T_touch p)| <= nn_record_move_bound(...)
// =: b from where it stood at the touch, so its neighbour is now at most d_i + b away and every other point still
// farther than lb_i - b: with b < margin_i the inequality of nn_still_nearest holds for the query (with room to
// spare), and with b < min_i margin_i the WHOLE record is re-validated without loading a single query.  Nothing is
// rewritten then: the bounds stay relative to the touch pass, which is why the displacement is taken from the touch
// transform itself and not summed over the passes in between.  A record that fails runs the per-query test against
// T_touch, searches what fails that, and is touched anew.
struct WaveRec { float c[3]; float e[3]; float margin; int touch; };   // 32 bytes

S3D_HD float nn_margin(bool has_neighbour, float lb_others, float d, float max_d) {
  const float room = lb_others * 0.99999f - 1.0e-6f;
  return has_neighbour ? 0.5f * (room - d) : room - max_d;
}

// upper bound of |fl(T p) - fl(Tt p)| (xf_eigen, float) over the points p of the box (c, e); double arithmetic on
// the exact float entries.  |(T - Tt)(c, 1)| + sum_j |(T - Tt) column j| e_j bounds the exact displacement.  The
// rounding of a float transform is bounded per coordinate r by gamma_4 (|T_r0 x| + |T_r1 y| + |T_r2 z| + |t_r|),
// gamma_4 <= 4.0000008 u, u = 2^-24 (((a x + b y) + c z) + t: no term passes more than four roundings); both
// transforms together, per coordinate, then as a vector.  It dominates once a registration has settled - at 40 m
// from the origin a float coordinate is quantised to 3.8e-6 m and the bound is 1.9e-5 m - so it is taken per
// coordinate from the box's own largest |x|, |y|, |z|, not from one norm for all three.
// sqrt rounded UP, through the float unit (a double sqrt is ~50 instructions on the device): (float) s is within 2^-24
// of s, sqrtf within an ulp; an argument below the float range (< 1e-38: a displacement < 1e-19 m) may come out as 0,
// which the constant term of the bound absorbs
S3D_HD double sqrt_up(double s) { return (double)sqrtf((float)s) * 1.000001; }
S3D_HD double nn_record_move_bound(const Mat4f& T, const Mat4f& Tt, const float c[3], const float e[3]) {
  double d[3][4];
  for (int r = 0; r < 3; ++r)
    for (int a = 0; a < 4; ++a) d[r][a] = (double)S3D_M(T, r, a) - (double)S3D_M(Tt, r, a);
  double v2 = 0.0;
  for (int r = 0; r < 3; ++r) {
    const double v = fma(d[r][0], (double)c[0], fma(d[r][1], (double)c[1], fma(d[r][2], (double)c[2], d[r][3])));
    v2 = fma(v, v, v2);
  }
  double b = sqrt_up(v2);
  for (int a = 0; a < 3; ++a)
    b += sqrt_up(fma(d[0][a], d[0][a], fma(d[1][a], d[1][a], d[2][a] * d[2][a]))) * (double)e[a];
  const double pmax[3] = {fabs((double)c[0]) + (double)e[0], fabs((double)c[1]) + (double)e[1],
                          fabs((double)c[2]) + (double)e[2]};
  double g2 = 0.0;
  for (int r = 0; r < 3; ++r) {
    double g = fabs((double)S3D_M(T, r, 3)) + fabs((double)S3D_M(Tt, r, 3));
    for (int a = 0; a < 3; ++a) g = fma(fabs((double)S3D_M(T, r, a)) + fabs((double)S3D_M(Tt, r, a)), pmax[a], g);
    g2 = fma(g, g, g2);
  }
  return b * 1.000001 + 2.385e-7 * sqrt_up(g2) + 1.0e-9;   // 4 u = 2.3842e-7
}

// Exact k-NN of a point among its own cloud by ring expansion.  The k best are
// kept in caller-provided storage addressed as d2s[j*stride], idxs[j*stride]
// (LDS columns on the GPU).  Order of the result is unspecified.
template <typename F4T>
S3D_HD int grid_knn(const GridParams& g, const uint32_t* __restrict__ cell_start, const F4T* __restrict__ pts,
                    float qx, float qy, float qz, int k, float* d2s, int* idxs, int stride) {
  int cnt = 0, maxslot = 0;
  float maxd = -1.f; int maxi = -1;
  const float fx = (qx - g.origin[0]) * g.inv_h, fy = (qy - g.origin[1]) * g.inv_h,
              fz = (qz - g.origin[2]) * g.inv_h;
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  float ox = fx - (float)ix, oy = fy - (float)iy, oz = fz - (float)iz;
  float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
  face = fmaxf(face - 2.0e-3f, 0.f);
  const int rmax = imax(imax(g.dim[0], g.dim[1]), g.dim[2]);
  for (int r = 0; r <= rmax; ++r) {
    const int z0 = imax(iz - r, 0), z1 = imin(iz + r, g.dim[2] - 1);
    const int y0 = imax(iy - r, 0), y1 = imin(iy + r, g.dim[1] - 1);
    const int xl = ix - r, xh = ix + r;
    for (int cz = z0; cz <= z1; ++cz) {
      const bool zface = (cz == iz - r) || (cz == iz + r);
      for (int cy = y0; cy <= y1; ++cy) {
        const bool full = zface || (cy == iy - r) || (cy == iy + r);
        const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
        for (int part = 0; part < 2; ++part) {
          int xa, xb;
          if (full) {
            if (part) break;
            xa = imax(xl, 0); xb = imin(xh, g.dim[0] - 1);
          } else {
            if (r == 0) { if (part) break; xa = xb = ix; }
            else { xa = xb = part ? xh : xl; }
            if (xa < 0 || xa >= g.dim[0]) continue;
          }
          if (xa > xb) continue;
          const uint32_t s = cell_start[rowbase + xa], e = cell_start[rowbase + xb + 1];
          for (uint32_t kk = s; kk < e; ++kk) {
            const F4T p = pts[kk];
            const float d2 = dist2(qx, qy, qz, p.x, p.y, p.z);
            const int pi = __builtin_bit_cast(int, p.w);
            if (cnt < k) {
              d2s[cnt * stride] = d2; idxs[cnt * stride] = pi;
              if (cnt == 0 || lex_less(maxd, maxi, d2, pi)) { maxd = d2; maxi = pi; maxslot = cnt; }
              ++cnt;
            } else if (lex_less(d2, pi, maxd, maxi)) {
              d2s[maxslot * stride] = d2; idxs[maxslot * stride] = pi;
              maxd = d2; maxi = pi;  // provisional; rescan for the true maximum
              for (int j = 0; j < k; ++j) {
                const float dj = d2s[j * stride]; const int ij = idxs[j * stride];
                if (lex_less(maxd, maxi, dj, ij)) { maxd = dj; maxi = ij; maxslot = j; }
              }
            }
          }
        }
      }
    }
    const float bound = ((float)r + face) * g.h;
    if (cnt >= k && maxd <= bound * bound) break;
  }
  return cnt;
}

// Register-resident variant (the one the GPU runs for k <= 32): the k best are kept as a SORTED
// list of packed 64-bit keys ((float bits of d2 + 2^23) << 32 | index; d2 >= 0 so the integer order
// is the lexicographic (d2, index) order).  The keys are held as DOUBLES: for positive, finite,
// normal doubles the IEEE order equals the integer order of the bit patterns, so one chain step is
// v_min_f64 + v_max_f64 instead of a 64-bit compare and four selects (the +2^23 keeps every key a
// normal number, the sentinel is the largest finite pattern below it).  An insertion is one unrolled
// min/max chain — no LDS, no re-scan — and the list comes out in ascending distance, the order in
// which PCL sums the neighbours.  Returns the number of valid entries (<= k).
// one chain step on double-held keys.  On the GPU the two instructions are emitted directly:
// fmin()/fmax() would add a canonicalising v_max_f64 per operand (sNaN quieting) that the keys, which
// are never NaN, do not need.
S3D_HD void knn_chain_step(double& slot, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double hi;
  asm("v_max_f64 %0, %1, %2" : "=&v"(hi) : "v"(slot), "v"(c));
  asm("v_min_f64 %0, %0, %1" : "+v"(slot) : "v"(c));   // in place: no register copies around the chain
  c = hi;
#else
  const double lo = slot < c ? slot : c;
  c = slot < c ? c : slot;
  slot = lo;
#endif
}

// the whole 20-step chain as ONE asm block (k <= 20 is the reference default): between separate asm
// statements hipcc pads an s_nop it cannot prove unnecessary, i.e. 20 wasted issue slots per insertion
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void knn_chain20(double (&k)[20], double& c) {
  double t;
  asm("v_max_f64 %21, %0, %20\n\tv_min_f64 %0, %0, %20\n\tv_max_f64 %20, %1, %21\n\tv_min_f64 %1, %1, %21\n\tv_max_f64 %21, %2, %20\n\tv_min_f64 %2, %2, %20\n\tv_max_f64 %20, %3, %21\n\tv_min_f64 %3, %3, %21\n\tv_max_f64 %21, %4, %20\n\tv_min_f64 %4, %4, %20\n\tv_max_f64 %20, %5, %21\n\tv_min_f64 %5, %5, %21\n\tv_max_f64 %21, %6, %20\n\tv_min_f64 %6, %6, %20\n\tv_max_f64 %20, %7, %21\n\tv_min_f64 %7, %7, %21\n\tv_max_f64 %21, %8, %20\n\tv_min_f64 %8, %8, %20\n\tv_max_f64 %20, %9, %21\n\tv_min_f64 %9, %9, %21\n\tv_max_f64 %21, %10, %20\n\tv_min_f64 %10, %10, %20\n\tv_max_f64 %20, %11, %21\n\tv_min_f64 %11, %11, %21\n\tv_max_f64 %21, %12, %20\n\tv_min_f64 %12, %12, %20\n\tv_max_f64 %20, %13, %21\n\tv_min_f64 %13, %13, %21\n\tv_max_f64 %21, %14, %20\n\tv_min_f64 %14, %14, %20\n\tv_max_f64 %20, %15, %21\n\tv_min_f64 %15, %15, %21\n\tv_max_f64 %21, %16, %20\n\tv_min_f64 %16, %16, %20\n\tv_max_f64 %20, %17, %21\n\tv_min_f64 %17, %17, %21\n\tv_max_f64 %21, %18, %20\n\tv_min_f64 %18, %18, %20\n\tv_max_f64 %20, %19, %21\n\tv_min_f64 %19, %19, %21"
      : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(k[5]), "+v"(k[6]), "+v"(k[7]), "+v"(k[8]),
        "+v"(k[9]), "+v"(k[10]), "+v"(k[11]), "+v"(k[12]), "+v"(k[13]), "+v"(k[14]), "+v"(k[15]), "+v"(k[16]),
        "+v"(k[17]), "+v"(k[18]), "+v"(k[19]), "+v"(c), "=&v"(t));
}
#endif
template <int KMAX>
S3D_HD void knn_chain(double (&keys)[KMAX], double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (KMAX == 20) {
    knn_chain20(keys, c);
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < KMAX; ++j) knn_chain_step(keys[j], c);
}

S3D_HD float knn_key_d2(double key) {
  return __builtin_bit_cast(float, (uint32_t)(__builtin_bit_cast(unsigned long long, key) >> 32) - 0x00800000u);
}

// FULL: k == KMAX is known at compile time (the default k = 20 on the <20> instantiation): `worst` is then simply
// the last slot; with a run-time k the compiler evaluates the 20-way select chain after EVERY insertion
// (80 v_cndmask per insertion, a fifth of the kernel's instructions).
// BYPOS (the fused pre-pass, whose points carry PCL's voxel key in .w instead of an index): the low word of a key is the
// candidate's POSITION in pts, which is what the caller gathers the neighbours by; equal distances then order by
// position (cell, then voxel order) instead of by index.
template <int KMAX, bool FULL = false, bool BYPOS = false, typename F4T = void>
S3D_HD int grid_knn_sorted(const GridParams& g, const uint32_t* __restrict__ cell_start,
                           const F4T* __restrict__ pts, float qx, float qy, float qz, int k,
                           unsigned long long (&keys_out)[KMAX]) {
  const double kInf = __builtin_bit_cast(double, 0x7FDFFFFFFFFFFFFFull);
  double keys[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) keys[j] = kInf;
  double worst = kInf;  // == keys[k-1]
  int cnt = 0;
  const float fx = (qx - g.origin[0]) * g.inv_h, fy = (qy - g.origin[1]) * g.inv_h,
              fz = (qz - g.origin[2]) * g.inv_h;
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  float ox = fx - (float)ix, oy = fy - (float)iy, oz = fz - (float)iz;
  float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
  face = fmaxf(face - 2.0e-3f, 0.f);
  const int rmax = imax(imax(g.dim[0], g.dim[1]), g.dim[2]);
  // one candidate -> sorted list (unrolled min/max chain); `worst` mirrors keys[k-1]
#define S3D_KNN_INSERT(P_, POS_, D2_)                                                                   \
  {                                                                                                \
    double c = __builtin_bit_cast(double,                                                          \
        ((unsigned long long)(__builtin_bit_cast(uint32_t, (D2_)) + 0x00800000u) << 32) |          \
        (unsigned long long)(BYPOS ? (uint32_t)(POS_) : __builtin_bit_cast(uint32_t, (P_).w)));    \
    if (c < worst) {                                                                               \
      knn_chain<KMAX>(keys, c);                                                                    \
      if (FULL || k == KMAX) {                                                                     \
        worst = keys[KMAX - 1];                                                                    \
      } else {                                                                                     \
        worst = kInf;                                                                              \
        _Pragma("unroll") for (int j = 0; j < KMAX; ++j) worst = (j == k - 1) ? keys[j] : worst;   \
      }                                                                                            \
      ++cnt;                                                                                       \
    }                                                                                              \
  }
  // ---- rings 0 and 1 together = the 3x3x3 cells around the point: nine row ranges fetched as ONE
  // batch, then the points two at a time (the search is bound by dependent-load latency)
  {
    const int cix = imin(imax(ix, -1), g.dim[0]);            // (a query outside the grid: clamp, rows below are masked)
    const int xa = imax(cix - 1, 0), xb = imin(cix + 1, g.dim[0] - 1);
    uint32_t rs[9], re[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      // the point's own row first, then the four rows sharing a face with it, then the corner rows: the list
      // fills with near points early and most later candidates fail the one-compare pre-check instead of
      // running the 40-instruction insertion chain (the result does not depend on the order)
      const int rr = r == 0 ? 4 : (r <= 4 ? 2 * r - 1 : (r == 5 ? 0 : (r == 6 ? 2 : (r == 7 ? 6 : 8))));
      const int cy = iy + (rr % 3) - 1, cz = iz + (rr / 3) - 1;
      const bool ok = xa <= xb && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
      const int rowbase = ok ? g.dim[0] * (cy + g.dim[1] * cz) : 0;
      const uint32_t a = cell_start[rowbase + (ok ? xa : 0)], b = cell_start[rowbase + (ok ? xb + 1 : 0)];
      rs[r] = a; re[r] = ok ? b : a;
    }
#if S3D_KNN_TWOPHASE
    // Flattened loops over the candidates of several rows: with a loop per row the wave would run max-over-lanes
    // steps for every row (and pay the insertion chain on each); flattened it runs max-over-lanes of the TOTAL.
    // Two of them: the point's own row + the four face rows, then the four corner rows - the select chain that
    // maps a flat index to its row is then 4 or 3 compare/select pairs long instead of 8 (it is a quarter of the
    // loop's instructions), and a corner row whose slab lies beyond the k-th distance reached so far is dropped.
#define S3D_KNN_PHASE(R0_, NR_)                                                                                  \
    {                                                                                                            \
      uint32_t cum[NR_ + 1];                                                                                     \
      cum[0] = 0;                                                                                                \
      _Pragma("unroll") for (int r = 0; r < NR_; ++r) cum[r + 1] = cum[r] + (re[R0_ + r] - rs[R0_ + r]);         \
      const uint32_t total = cum[NR_];                                                                           \
      auto flatpos = [&](uint32_t t) {                                                                           \
        uint32_t b_ = rs[R0_];                                                                                   \
        _Pragma("unroll") for (int r = 1; r < NR_; ++r) b_ = t >= cum[r] ? rs[R0_ + r] - cum[r] : b_;            \
        return t + b_;                                                                                           \
      };                                                                                                         \
      F4T na = pts[0], nb = pts[0];                                                                              \
      uint32_t ia = 0, ib = 0;                                                                                   \
      if (total > 0) { ia = flatpos(0u); ib = flatpos(total > 1 ? 1u : 0u); na = pts[ia]; nb = pts[ib]; }        \
      for (uint32_t t = 0; t < total; t += 2) {                                                                  \
        const F4T pa = na, pb = nb;                                                                              \
        const uint32_t ja = ia, jb = ib;                                                                         \
        const bool two = t + 1 < total;                                                                          \
        if (t + 2 < total) {                                                                                     \
          ia = flatpos(t + 2);                                                                                   \
          ib = flatpos(t + 3 < total ? t + 3 : t + 2);                                                           \
          na = pts[ia];                                                                                          \
          nb = pts[ib];                                                                                          \
        }                                                                                                        \
        const float da = dist2(qx, qy, qz, pa.x, pa.y, pa.z);                                                    \
        const float db = dist2(qx, qy, qz, pb.x, pb.y, pb.z);                                                    \
        S3D_KNN_INSERT(pa, ja, da)                                                                               \
        if (two) S3D_KNN_INSERT(pb, jb, db)                                                                      \
      }                                                                                                          \
    }
    S3D_KNN_PHASE(0, 5)
    if (worst != kInf) {   // the list is full: corner rows out of reach of the k-th distance hold nothing
      const float lim2 = knn_key_d2(worst), eps = 2.0e-3f * g.h;
      const float ylo = g.origin[1] + (float)iy * g.h, zlo = g.origin[2] + (float)iz * g.h;
      const float dym = fmaxf(qy - ylo - eps, 0.f), dyp = fmaxf(ylo + g.h - qy - eps, 0.f);   // to the rows at y-1 / y+1
      const float dzm = fmaxf(qz - zlo - eps, 0.f), dzp = fmaxf(zlo + g.h - qz - eps, 0.f);
      if (dym * dym + dzm * dzm > lim2) re[5] = rs[5];
      if (dyp * dyp + dzm * dzm > lim2) re[6] = rs[6];
      if (dym * dym + dzp * dzp > lim2) re[7] = rs[7];
      if (dyp * dyp + dzp * dzp > lim2) re[8] = rs[8];
    }
    S3D_KNN_PHASE(5, 4)
#undef S3D_KNN_PHASE
#else
    // ONE flattened loop over the candidates of all nine rows: with a loop per row the wave would run
    // max-over-lanes steps for every row (and pay the insertion chain on each); flattened it runs
    // max-over-lanes of the TOTAL.  cum[r] = candidates before row r, off[r] maps a flat index into it.
    uint32_t cum[10];
    cum[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) cum[r + 1] = cum[r] + (re[r] - rs[r]);
    const uint32_t total = cum[9];
    // flat index -> position in pts (select chain over the nine row offsets, no register indexing)
#define S3D_KNN_FLATPOS(T_, OUT_)                                         \
    {                                                                     \
      uint32_t b_ = rs[0];                                                \
      _Pragma("unroll") for (int r = 1; r < 9; ++r)                       \
        b_ = (T_) >= cum[r] ? rs[r] - cum[r] : b_;                        \
      OUT_ = (T_) + b_;                                                   \
    }
    // software-pipelined: the two points of step t+2 are requested before the insertion chains of
    // step t run, so the chains (~200 instructions) cover the load latency
    F4T na = pts[0], nb = pts[0];
    uint32_t ia = 0, ib = 0;
    if (total > 0) {
      S3D_KNN_FLATPOS(0u, ia)
      S3D_KNN_FLATPOS((total > 1 ? 1u : 0u), ib)
      na = pts[ia]; nb = pts[ib];
    }
    for (uint32_t t = 0; t < total; t += 2) {
      F4T pa = na, pb = nb;
      uint32_t ja = ia, jb = ib;
      const bool two = t + 1 < total;
      if (!S3D_KNN_PREFETCH && t > 0) {
        S3D_KNN_FLATPOS(t, ja)
        S3D_KNN_FLATPOS((two ? t + 1 : t), jb)
        pa = pts[ja]; pb = pts[jb];
      }
      if (S3D_KNN_PREFETCH && t + 2 < total) {
        const uint32_t t3 = t + 3 < total ? t + 3 : t + 2;
        S3D_KNN_FLATPOS(t + 2, ia)
        S3D_KNN_FLATPOS(t3, ib)
        na = pts[ia]; nb = pts[ib];
      }
      const float da = dist2(qx, qy, qz, pa.x, pa.y, pa.z);
      const float db = dist2(qx, qy, qz, pb.x, pb.y, pb.z);
      S3D_KNN_INSERT(pa, ja, da)
      if (two) S3D_KNN_INSERT(pb, jb, db)
    }
#undef S3D_KNN_FLATPOS
#endif
    const float bound = (1.0f + face) * g.h;
    if (worst != kInf && knn_key_d2(worst) <= bound * bound) {
#pragma unroll
      for (int j = 0; j < KMAX; ++j) keys_out[j] = __builtin_bit_cast(unsigned long long, keys[j]);
      return cnt < k ? cnt : k;
    }
  }
  // ---- ring 2, pruned: the list is full, so the k-th distance so far bounds the true one.  If that ball lies
  // inside the 5x5x5 cells around the point, only the ring-2 cells the ball reaches can still contribute:
  // rows are slab-tested against the (shrinking) k-th distance and cut to the x-range the ball covers, the
  // 3x3x3 cells already examined are skipped.  On surface-like clouds this reads a handful of cells instead of
  // the 98 of the full shell (the un-pruned ring loop below was most of the kernel's time).
  if (worst != kInf) {
    const float b2 = (2.0f + face) * g.h;
    float lim2 = knn_key_d2(worst);
    if (lim2 <= b2 * b2) {
      const float eps = 2.0e-3f * g.h;
      for (int dz = -2; dz <= 2; ++dz) {
        const int cz = iz + dz;
        if (cz < 0 || cz >= g.dim[2]) continue;
        const float zlo = g.origin[2] + (float)cz * g.h;
        const float fz2 = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
        if (fz2 * fz2 > lim2) continue;
        for (int dy = -2; dy <= 2; ++dy) {
          const int cy = iy + dy;
          if (cy < 0 || cy >= g.dim[1]) continue;
          const float ylo = g.origin[1] + (float)cy * g.h;
          const float fy2 = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
          const float rowd2 = fy2 * fy2 + fz2 * fz2;
          if (rowd2 > lim2) continue;
          const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
          const int xa = imax(imax(ix - 2, grid_coord(g, 0, qx - rx)), 0);
          const int xb = imin(imin(ix + 2, grid_coord(g, 0, qx + rx)), g.dim[0] - 1);
          const bool inner = dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
          const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
          for (int part = 0; part < 2; ++part) {
            int sa, sb;
            if (!inner) { if (part) break; sa = xa; sb = xb; }
            else if (part == 0) { sa = xa; sb = imin(xb, ix - 2); }    // left of the examined cells
            else { sa = imax(xa, ix + 2); sb = xb; }                  // right of them
            if (sa > sb) continue;
            const uint32_t s = cell_start[rowbase + sa], e = cell_start[rowbase + sb + 1];
            for (uint32_t kk = s; kk < e; kk += 4) {      // four loads in flight per step
              const uint32_t last = e - 1;
              F4T pp[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) pp[u] = pts[kk + u < e ? kk + u : last];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if (kk + u < e) {
                  const float d2 = dist2(qx, qy, qz, pp[u].x, pp[u].y, pp[u].z);
                  S3D_KNN_INSERT(pp[u], kk + u, d2)
                }
              }
            }
          }
          lim2 = fminf(lim2, knn_key_d2(worst));
        }
      }
#pragma unroll
      for (int j = 0; j < KMAX; ++j) keys_out[j] = __builtin_bit_cast(unsigned long long, keys[j]);
      return cnt < k ? cnt : k;
    }
  }
  // ---- beyond: (round 5) whole rings only while the list is not full - a point in a sparse part of the cloud.  Once it
  // is, its k-th distance bounds the true one and ONE pruned box finishes the search: the rows of the box of that
  // (shrinking) radius, slab-tested and cut to the chord of the ball, the cube of rings already examined left out.
  // (Until round 4 the rings went on un-pruned until the bound proved the list: (2r + 1)^2 row look-ups per ring, nearly
  // all of them empty - on the reference's scans, whose far field is rings of points metres apart, 13 % of the points
  // took this path and it was most of the pre-pass.)
  int rex = 1;                       // the cube of rings <= rex has been examined
  if (worst == kInf) {
    for (int r = 2; r <= rmax; ++r) {
      const int z0 = imax(iz - r, 0), z1 = imin(iz + r, g.dim[2] - 1);
      const int y0 = imax(iy - r, 0), y1 = imin(iy + r, g.dim[1] - 1);
      const int xl = ix - r, xh = ix + r;
      for (int cz = z0; cz <= z1; ++cz) {
        const bool zface = (cz == iz - r) || (cz == iz + r);
        for (int cy = y0; cy <= y1; ++cy) {
          const bool full = zface || (cy == iy - r) || (cy == iy + r);
          const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
          for (int part = 0; part < 2; ++part) {
            int xa, xb;
            if (full) {
              if (part) break;
              xa = imax(xl, 0); xb = imin(xh, g.dim[0] - 1);
            } else {
              xa = xb = part ? xh : xl;
              if (xa < 0 || xa >= g.dim[0]) continue;
            }
            if (xa > xb) continue;
            const uint32_t s = cell_start[rowbase + xa], e = cell_start[rowbase + xb + 1];
            for (uint32_t kk = s; kk < e; ++kk) {
              const F4T p = pts[kk];
              const float d2 = dist2(qx, qy, qz, p.x, p.y, p.z);
              S3D_KNN_INSERT(p, kk, d2)
            }
          }
        }
      }
      rex = r;
      if (worst != kInf) break;
    }
  }
  if (worst != kInf) {
    const float bound = ((float)rex + face) * g.h;
    float lim2 = knn_key_d2(worst);
    if (lim2 > bound * bound) {
      const float eps = 2.0e-3f * g.h;
      const float R = sqrt_bound(lim2) * 1.0001f + eps;
      const int z0 = imax(grid_coord(g, 2, qz - R), 0), z1 = imin(grid_coord(g, 2, qz + R), g.dim[2] - 1);
      const int y0 = imax(grid_coord(g, 1, qy - R), 0), y1 = imin(grid_coord(g, 1, qy + R), g.dim[1] - 1);
      for (int cz = z0; cz <= z1; ++cz) {
        const float zlo = g.origin[2] + (float)cz * g.h;
        const float fz2 = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
        if (fz2 * fz2 > lim2) continue;
        for (int cy = y0; cy <= y1; ++cy) {
          const float ylo = g.origin[1] + (float)cy * g.h;
          const float fy2 = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
          const float rowd2 = fy2 * fy2 + fz2 * fz2;
          if (rowd2 > lim2) continue;
          const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
          const int xa = imax(grid_coord(g, 0, qx - rx), 0);
          const int xb = imin(grid_coord(g, 0, qx + rx), g.dim[0] - 1);
          const bool inner = cy >= iy - rex && cy <= iy + rex && cz >= iz - rex && cz <= iz + rex;
          const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
          for (int part = 0; part < 2; ++part) {
            int sa, sb;
            if (!inner) { if (part) break; sa = xa; sb = xb; }
            else if (part == 0) { sa = xa; sb = imin(xb, ix - rex - 1); }    // left of the examined cube
            else { sa = imax(xa, ix + rex + 1); sb = xb; }                   // right of it
            if (sa > sb) continue;
            const uint32_t s = cell_start[rowbase + sa], e = cell_start[rowbase + sb + 1];
            for (uint32_t kk = s; kk < e; kk += 4) {      // four loads in flight per step
              const uint32_t last = e - 1;
              F4T pp[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) pp[u] = pts[kk + u < e ? kk + u : last];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if (kk + u < e) {
                  const float d2 = dist2(qx, qy, qz, pp[u].x, pp[u].y, pp[u].z);
                  S3D_KNN_INSERT(pp[u], kk + u, d2)
                }
              }
            }
          }
          lim2 = fminf(lim2, knn_key_d2(worst));
        }
      }
    }
  }
#undef S3D_KNN_INSERT
#pragma unroll
  for (int j = 0; j < KMAX; ++j) keys_out[j] = __builtin_bit_cast(unsigned long long, keys[j]);
  return cnt < k ? cnt : k;
}

// ------------------------------------------------------------------ K4, round 3: 32-bit keys, med3 insertion
//
// The list above pays 2 x 20 VALU instructions per insertion (64-bit keys: v_min_f64 + v_max_f64 per slot) and runs
// the chain whenever ANY lane of the wave accepts a candidate.  This variant keeps the k + 1 best as 32-bit keys
//     key = (float bits of d2 with the low 11 mantissa bits cleared) | (segment << 7) | offset
// i.e. the distance truncated to 12 mantissa bits with the candidate's PLACE in the low bits: (segment, offset)
// addresses a per-lane table of the row segments the search visits (LDS on the GPU), so neither the index nor
// the position travels through the list.  On 32-bit keys an insertion into a sorted list is ONE instruction per
// slot, new[j] = med3(old[j-1], old[j], c): the slots are independent of each other (no carried value), the chain
// runs unconditionally (a candidate that is too far, or a masked tail slot carrying the sentinel, leaves the list
// as it is) and the loop has no per-candidate branch.
//
// Exactness.  Truncation is monotone, so the k smallest keys are the k nearest points unless the k-th and the
// (k+1)-th key agree in their distance bits - which slot k (the extra one) shows.  Such a query (about 20 x 2^-12 of
// them), one whose k-th distance reaches beyond the examined cells, one with more row segments than the table holds
// and one with fewer than k candidates are NOT answered here: the function returns false and the caller hands the
// point to the exact 64-bit search above (grid_knn_sorted).  Everything answered here is the exact k-NN set with
// the (d2, index) tie rule; the neighbours come out in ascending TRUNCATED distance, i.e. PCL's summation order up
// to swaps inside a 2^-12 band.
constexpr int kKnn3Segs = 16;         // table entries per lane
constexpr int kKnn3OffBits = 7;       // a table entry covers at most 128 consecutive points
constexpr uint32_t kKnn3OffMask = (1u << kKnn3OffBits) - 1u;
constexpr int kKnn3IdBits = kKnn3OffBits + 4;
constexpr uint32_t kKnn3IdMask = (1u << kKnn3IdBits) - 1u, kKnn3Sentinel = 0xFFFFFFFFu;
constexpr int kKnn3MaxPoints = 1 << (32 - kKnn3OffBits);   // positions must fit a table entry
#ifndef S3D_KNN3_FAR_RINGS
#define S3D_KNN3_FAR_RINGS 4
#endif
constexpr int kKnn3FarRings = S3D_KNN3_FAR_RINGS;          // a declined query whose K-th neighbour may lie farther than this many cells is "far"

S3D_HD uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
#endif
}

// one candidate into the sorted list of KL keys (ascending)
template <int KL>
S3D_HD void knn3_insert(uint32_t (&keys)[KL], uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (KL == 21) {   // k = 20 (the reference default): the whole chain as ONE asm block, top slot first
    asm("v_med3_u32 %20, %19, %20, %21\n\tv_med3_u32 %19, %18, %19, %21\n\tv_med3_u32 %18, %17, %18, %21\n\t"
        "v_med3_u32 %17, %16, %17, %21\n\tv_med3_u32 %16, %15, %16, %21\n\tv_med3_u32 %15, %14, %15, %21\n\t"
        "v_med3_u32 %14, %13, %14, %21\n\tv_med3_u32 %13, %12, %13, %21\n\tv_med3_u32 %12, %11, %12, %21\n\t"
        "v_med3_u32 %11, %10, %11, %21\n\tv_med3_u32 %10, %9, %10, %21\n\tv_med3_u32 %9, %8, %9, %21\n\t"
        "v_med3_u32 %8, %7, %8, %21\n\tv_med3_u32 %7, %6, %7, %21\n\tv_med3_u32 %6, %5, %6, %21\n\t"
        "v_med3_u32 %5, %4, %5, %21\n\tv_med3_u32 %4, %3, %4, %21\n\tv_med3_u32 %3, %2, %3, %21\n\t"
        "v_med3_u32 %2, %1, %2, %21\n\tv_med3_u32 %1, %0, %1, %21\n\tv_min_u32 %0, %0, %21"
        : "+v"(keys[0]), "+v"(keys[1]), "+v"(keys[2]), "+v"(keys[3]), "+v"(keys[4]), "+v"(keys[5]), "+v"(keys[6]),
          "+v"(keys[7]), "+v"(keys[8]), "+v"(keys[9]), "+v"(keys[10]), "+v"(keys[11]), "+v"(keys[12]), "+v"(keys[13]),
          "+v"(keys[14]), "+v"(keys[15]), "+v"(keys[16]), "+v"(keys[17]), "+v"(keys[18]), "+v"(keys[19]), "+v"(keys[20])
        : "v"(c));
    return;
  }
#endif
#pragma unroll
  for (int j = KL - 1; j >= 1; --j) keys[j] = umed3(keys[j - 1], keys[j], c);
  keys[0] = keys[0] < c ? keys[0] : c;
}

// a row range [s, e) of the cell-sorted cloud -> one table entry, tab[j * tstride] = (start << 7) | (len - 1).
// false: the table is full or the range is longer than an entry can say (the caller gives the query up).
template <int SEGBITS = 4>
S3D_HD bool knn3_push(uint32_t* tab, int tstride, int& nseg, uint32_t s, uint32_t e) {
  if (s >= e) return true;
  constexpr uint32_t kMax = kKnn3OffMask + 1u;
  constexpr int kKnn3Segs = 1 << SEGBITS;      // (4: the 16-entry table of the fast path, == s3d::kKnn3Segs)
  if (e - s > kMax) {          // a long range (coarse grids): two entries, beyond that the exact search
    if (nseg >= kKnn3Segs || e - s > 2u * kMax) return false;
    tab[nseg * tstride] = (s << kKnn3OffBits) | kKnn3OffMask;
    ++nseg;
    s += kMax;
  }
  if (nseg >= kKnn3Segs) return false;
  tab[nseg * tstride] = (s << kKnn3OffBits) | (e - s - 1u);
  ++nseg;
  return true;
}

// scan the table entries [e0, nseg) of this lane into the list: ONE flat loop over all their points (a wave runs
// max-over-lanes of the TOTAL, not of every row), two points per trip with the loads of the next two in flight.
// A lane that has run out carries the invalid id: its key is the sentinel and the insertion leaves its list alone.
template <int KL, int SEGBITS = 4, typename F4T>
S3D_HD void knn3_scan(uint32_t (&keys)[KL], const uint32_t* tab, int tstride, int e0, int nseg,
                      const F4T* __restrict__ pts, float qx, float qy, float qz) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  constexpr uint32_t kKnn3IdMask = (1u << (kKnn3OffBits + SEGBITS)) - 1u;   // (SEGBITS = 4: s3d::kKnn3IdMask)
  int e = e0;
  uint32_t pos = 0, end = 0, idelta = 0;
  // [pos, end): what is left of the current entry; id of a point = pos + idelta.  The load itself is unconditional
  // (a lane without a next point re-reads position 0) so that it stays in flight across the insertion below.
#define S3D_KNN3_NEXT(P_, ID_)                                                                         \
  {                                                                                                    \
    if (pos == end && e < nseg) {                                                                      \
      const uint32_t ent = tab[e * tstride];                                                           \
      pos = ent >> kKnn3OffBits; end = pos + (ent & kKnn3OffMask) + 1u;                                \
      idelta = ((uint32_t)e << kKnn3OffBits) - pos;                                                    \
      ++e;                                                                                             \
    }                                                                                                  \
    const bool has_ = pos != end;                                                                      \
    P_ = pts[has_ ? pos : 0u];                                                                         \
    ID_ = has_ ? pos + idelta : kNone;                                                                 \
    pos += has_ ? 1u : 0u;                                                                             \
  }
#define S3D_KNN3_USE(P_, ID_)                                                                          \
  {                                                                                                    \
    const float d2_ = dist2_xy(qx, qy, qz, P_);                                                        \
    const uint32_t key_ = (__builtin_bit_cast(uint32_t, d2_) & ~kKnn3IdMask) | ID_;   /* kNone: the sentinel */ \
    if (key_ < keys[KL - 1]) knn3_insert<KL>(keys, key_);                                              \
  }
#if S3D_KNN3_DEPTH == 4
  F4T pa, pb, pc, pd;
  uint32_t ia, ib, ic, id_;
  S3D_KNN3_NEXT(pa, ia)
  S3D_KNN3_NEXT(pb, ib)
  S3D_KNN3_NEXT(pc, ic)
  S3D_KNN3_NEXT(pd, id_)
  while (ia != kNone) {
    S3D_KNN3_USE(pa, ia)
    S3D_KNN3_NEXT(pa, ia)
    S3D_KNN3_USE(pb, ib)
    S3D_KNN3_NEXT(pb, ib)
    S3D_KNN3_USE(pc, ic)
    S3D_KNN3_NEXT(pc, ic)
    S3D_KNN3_USE(pd, id_)
    S3D_KNN3_NEXT(pd, id_)
  }
#else
  F4T pa, pb;
  uint32_t ia, ib;
  S3D_KNN3_NEXT(pa, ia)
  S3D_KNN3_NEXT(pb, ib)
  while (ia != kNone) {
    S3D_KNN3_USE(pa, ia)
    S3D_KNN3_NEXT(pa, ia)
    S3D_KNN3_USE(pb, ib)
    S3D_KNN3_NEXT(pb, ib)
  }
#endif
#undef S3D_KNN3_NEXT
#undef S3D_KNN3_USE
}

// the largest d2 a key can stand for
template <int SEGBITS = 4>
S3D_HD float knn3_key_d2_upper(uint32_t key) { return __builtin_bit_cast(float, key | ((1u << (kKnn3OffBits + SEGBITS)) - 1u)); }

// ---- the stages of the search (grid_knn_med3 below runs them back to back for one query; the block kernel
// s3d_knn3_moments_kernel re-deals the queries of a block between them)

// distance of the point to the nearest face of its cell, in cells, less the rounding margin
S3D_HD float knn3_face(const GridParams& g, float qx, float qy, float qz) {
  const float fx = (qx - g.origin[0]) * g.inv_h, fy = (qy - g.origin[1]) * g.inv_h,
              fz = (qz - g.origin[2]) * g.inv_h;
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  const float ox = fx - (float)ix, oy = fy - (float)iy, oz = fz - (float)iz;
  const float face = fminf(fminf(fminf(ox, 1.f - ox), fminf(oy, 1.f - oy)), fminf(oz, 1.f - oz));
  return fmaxf(face - 2.0e-3f, 0.f);
}

// stage 1: the row segments of the 3x3x3 cells around the point -> table entries [0, nseg); total = their points.
// false: a range the table cannot hold.
template <int SEGBITS = 4>
S3D_HD bool knn3_build27(const GridParams& g, const uint32_t* __restrict__ cell_start, float qx, float qy, float qz,
                         uint32_t* tab, int tstride, int& nseg, uint32_t& total) {
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  const int cix = imin(imax(ix, -1), g.dim[0]);
  const int xa = imax(cix - 1, 0), xb = imin(cix + 1, g.dim[0] - 1);
  uint32_t rs[9], re[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {          // nine row ranges fetched as one batch
    const int cy = iy + (r % 3) - 1, cz = iz + (r / 3) - 1;
    const bool in = xa <= xb && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
    const int rowbase = in ? g.dim[0] * (cy + g.dim[1] * cz) : 0;
    const uint32_t a = cell_start[rowbase + (in ? xa : 0)], b = cell_start[rowbase + (in ? xb + 1 : 0)];
    rs[r] = a; re[r] = in ? b : a;
  }
  bool ok = true;
  nseg = 0; total = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    ok = knn3_push<SEGBITS>(tab, tstride, nseg, rs[r], re[r]) && ok;
    total += re[r] - rs[r];
  }
  return ok;
}

// after stage 1's scan: 0 = the K nearest are found, 1 = the pruned 5x5x5 shell has to be looked at (lim2: the ball
// that bounds it), 2 = not answerable here (fewer than K candidates, or the ball leaves the 5x5x5 cells)
template <int KL>
S3D_HD int knn3_after27(const GridParams& g, float face, const uint32_t (&keys)[KL], float& lim2) {
  constexpr int K = KL - 1;
  if (keys[K - 1] == kKnn3Sentinel) return 2;
  lim2 = knn3_key_d2_upper(keys[K - 1]);
  const float b1 = (1.0f + face) * g.h;
  if (lim2 <= b1 * b1) return 0;
  const float b2 = (2.0f + face) * g.h;
  return lim2 <= b2 * b2 ? 1 : 2;
}

// stage 2: the row segments of the 5x5x5 shell that the ball of squared radius lim2 reaches, appended to the table
// ([nseg on entry, nseg on return)); total = their points.  false: the table is full.
S3D_HD bool knn3_build_shell(const GridParams& g, const uint32_t* __restrict__ cell_start, float qx, float qy, float qz,
                             float lim2, uint32_t* tab, int tstride, int& nseg, uint32_t& total) {
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  const float eps = 2.0e-3f * g.h;
  bool ok = true;
  total = 0;
  for (int dz = -2; dz <= 2; ++dz) {
    const int cz = iz + dz;
    if (cz < 0 || cz >= g.dim[2]) continue;
    const float zlo = g.origin[2] + (float)cz * g.h;
    const float fz2 = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
    if (fz2 * fz2 > lim2) continue;
    for (int dy = -2; dy <= 2; ++dy) {
      const int cy = iy + dy;
      if (cy < 0 || cy >= g.dim[1]) continue;
      const float ylo = g.origin[1] + (float)cy * g.h;
      const float fy2 = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
      const float rowd2 = fy2 * fy2 + fz2 * fz2;
      if (rowd2 > lim2) continue;
      const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
      const int xa = imax(imax(ix - 2, grid_coord(g, 0, qx - rx)), 0);
      const int xb = imin(imin(ix + 2, grid_coord(g, 0, qx + rx)), g.dim[0] - 1);
      const bool inner = dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
      const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
      if (!inner) {
        if (xa <= xb) {
          const uint32_t a = cell_start[rowbase + xa], b = cell_start[rowbase + xb + 1];
          ok = knn3_push(tab, tstride, nseg, a, b) && ok; total += b - a;
        }
      } else {
        const int lb = imin(xb, ix - 2), ra = imax(xa, ix + 2);   // left / right of the cells already examined
        if (xa <= lb) {
          const uint32_t a = cell_start[rowbase + xa], b = cell_start[rowbase + lb + 1];
          ok = knn3_push(tab, tstride, nseg, a, b) && ok; total += b - a;
        }
        if (ra <= xb) {
          const uint32_t a = cell_start[rowbase + ra], b = cell_start[rowbase + xb + 1];
          ok = knn3_push(tab, tstride, nseg, a, b) && ok; total += b - a;
        }
      }
    }
  }
  return ok;
}

// the K-th and the (K+1)-th key in one distance band: the truncated order may not be the exact one
template <int KL, int SEGBITS = 4>
S3D_HD bool knn3_unambiguous(const uint32_t (&keys)[KL]) {
  return (keys[KL - 1] >> (kKnn3OffBits + SEGBITS)) != (keys[KL - 2] >> (kKnn3OffBits + SEGBITS));
}

// Exact K-NN of a point of the cloud among the cloud (K = KL - 1).  true: keys[0..K-1] hold the K nearest in
// ascending truncated distance, (key >> 7) & 15 = table entry, key & 127 = offset (knn3_position); false: not
// answered (see above).  tab: kKnn3Segs entries of this lane, stride tstride.
template <int KL, typename F4T>
// far (may be null): set when the query is declined because its K-th neighbour lies beyond the 5x5x5 cells (or the cells
// hold fewer than K points) - a point in a sparse part of the cloud, whose exact search is a long walk (the caller
// serves those wave-cooperatively, s3d_knn_moments_far_kernel); left alone otherwise.
// (round 6: *far = 2 for such a query, 1 for one declined at this point - the 27 cells searched, the K-th neighbour
// beyond the 5x5x5 proof - that is not "far": a large batch hands both kinds to the ring search, grid_knn_med3_rings below)
S3D_HD bool grid_knn_med3(const GridParams& g, const uint32_t* __restrict__ cell_start, const F4T* __restrict__ pts,
                          float qx, float qy, float qz, uint32_t* tab, int tstride, uint32_t (&keys)[KL],
                          int* far = nullptr) {
#pragma unroll
  for (int j = 0; j < KL; ++j) keys[j] = kKnn3Sentinel;
  int nseg;
  uint32_t total;
  if (!knn3_build27(g, cell_start, qx, qy, qz, tab, tstride, nseg, total)) return false;
  knn3_scan<KL>(keys, tab, tstride, 0, nseg, pts, qx, qy, qz);
  float lim2 = 0.f;
  const int st = knn3_after27<KL>(g, knn3_face(g, qx, qy, qz), keys, lim2);
  if (st == 2) {
    // how far?  An upper bound of the K-th distance is known (the K-th of the 27 cells' points, if they hold K at all).
    // Up to kKnn3FarRings cells the per-lane ring search is a fair walk; beyond, or with fewer than K points in the 27
    // cells, it is a long one
    if (far) {
      const float reach = (float)kKnn3FarRings * g.h;
      *far = (keys[KL - 2] == kKnn3Sentinel || knn3_key_d2_upper(keys[KL - 2]) > reach * reach) ? 2 : 1;
    }
    return false;
  }
  if (st == 1) {
    const int first = nseg;
    if (!knn3_build_shell(g, cell_start, qx, qy, qz, lim2, tab, tstride, nseg, total)) return false;
    knn3_scan<KL>(keys, tab, tstride, first, nseg, pts, qx, qy, qz);
  }
  return knn3_unambiguous<KL>(keys);
}

// position (in the cell-sorted cloud) of the neighbour a key stands for
template <int SEGBITS = 4>
S3D_HD uint32_t knn3_position(uint32_t key, const uint32_t* tab, int tstride) {
  return (tab[((key >> kKnn3OffBits) & ((1u << SEGBITS) - 1u)) * tstride] >> kKnn3OffBits) + (key & kKnn3OffMask);
}

// ------------------------------------------------------------------ K4, round 6: the SPARSE parts of a cloud, ring by ring
//
// What the fast path above declines on a real lidar scan is not what it declines on a dense cloud.  Measured on the
// reference's scans (tools_dev/knn3_declines.cpp; 0.2 m voxels, 1.0-1.2 m cells of which 98 % are empty): 5-7 % of the
// points have FEWER THAN K points in their 27 cells - the far field, ring lines metres apart - against 0.2 % whose
// K-th neighbour lies beyond the 5x5x5 proof, 0.2 % ties and 0.05 % table overflows; their true K-th neighbour lies in
// ring 2 (44 %), 3 (37 %), 4 (11 %), 5 (8 %).  The exact per-lane search (grid_knn_sorted) serves them ring after ring
// through nested per-row loops: on a batch of such scans it took 2.8 of the pre-pass's 4.4 ms for 13 % of the points.
// Here the same machinery that makes the fast path fast - row segments into a per-lane LDS table by batches of
// look-ups, ONE flat candidate loop per stage, 32-bit keys, v_med3 insertion - goes on beyond the 5x5x5 cells:
//   stage A  while the list holds fewer than K keys: whole rings, several at a time (rings 2 ... 3, then 4 ... 6: the
//            rows outside the examined square as one x-range each, the cells left and right of it for the rows
//            inside) - nearly all of it empty cells, so few segments;
//   stage B  the list is full, its K-th key bounds the true K-th distance: proven if that ball lies inside the cube of
//            rings examined so far, otherwise ONE pruned box finishes - the rows of the ball's box, slab-tested and cut to
//            the chord, the examined cube left out (what grid_knn_sorted does at its end).
// A table with 2^SEGBITS entries per lane; a key keeps 32 - 7 - SEGBITS distance bits (SEGBITS = 5: 11 mantissa bits,
// a tie band of 2^-11 at the K-th place: ~1 % of these queries are ambiguous and, like everything else this path cannot
// answer - table full, K-th neighbour beyond `rmax` rings - go on to the exact 64-bit search).  Same K-NN set, same tie
// rule as every other route (the neighbours' moments are sums of floats in double: exact, so their order is free).
#ifndef S3D_KNN_RING_BATCH
#define S3D_KNN_RING_BATCH 4
#endif
constexpr int kKnn3RingBatch = S3D_KNN_RING_BATCH;   // rows whose (up to four) table look-ups are in flight together
constexpr int kKnn3RingSegBits = 5;    // the ring search's table: 32 entries per lane (s3d_knn3_rings_kernel) - on the reference's
                                       // scans 64 entries answer FEWER queries (a key keeps one distance bit less: more ties)
#ifndef S3D_KNN_RING_MAX
#define S3D_KNN_RING_MAX 6
#endif
#ifndef S3D_KNN_RING_STEP
#define S3D_KNN_RING_STEP 3
#endif
constexpr int kKnn3RingMax = S3D_KNN_RING_MAX;     // ... how many rings it looks at before it hands the query on
constexpr int kKnn3RingStep = S3D_KNN_RING_STEP;   // ... in two steps while the list is short: rings 2 ... Step, then Step + 1 ... Max

// t / d for 0 <= t < 4096, 1 <= d <= 64, without an integer division (a multiply by ceil(2^18 / d) is exact there:
// the error t (ceil(2^18 / d) - 2^18 / d) < 4096 stays below 2^18 / d)
S3D_HD int small_div(int t, int magic) { return (int)(((unsigned)t * (unsigned)magic) >> 18); }
S3D_HD int small_div_magic(int d) { return ((1 << 18) + d - 1) / d; }

template <int SEGBITS>
S3D_HD bool knn3_build_ring(const GridParams& g, const uint32_t* __restrict__ cell_start, int ix, int iy, int iz, int rex,
                            int r, uint32_t* tab, int tstride, int& nseg) {
  // the cube of rings <= r less the cube of rings <= rex (examined): rows (cy, cz) of the square +-r in batches of
  // kKnn3RingBatch, their (up to four) table look-ups in flight together.  A row outside the examined square: ONE range
  // [ix - r, ix + r]; inside it: the cells left and right of the examined ones.  (Several rings in one step - rex + 1 < r
  // - cost far fewer look-ups than ring by ring: an x-range is one look-up however long, so the rows of rings 4, 5, 6
  // together are 218 look-ups instead of 130 + 202 + 290.)
  const int side = 2 * r + 1, nrows = side * side, magic = small_div_magic(side);
  bool ok = true;
  for (int t0 = 0; t0 < nrows; t0 += kKnn3RingBatch) {
    uint32_t a[kKnn3RingBatch][2], b[kKnn3RingBatch][2];
#pragma unroll
    for (int u = 0; u < kKnn3RingBatch; ++u) {
      const int t = t0 + u;
      const int tz = small_div(t, magic);
      const int dy = t - tz * side - r, dz = tz - r;
      const int cy = iy + dy, cz = iz + dz;
      const bool in = t < nrows && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
      const bool outer = dy < -rex || dy > rex || dz < -rex || dz > rex;
      const int rowbase = in ? g.dim[0] * (cy + g.dim[1] * cz) : 0;
      const int xl = imax(ix - r, 0), xh = imin(ix + r, g.dim[0] - 1);
      const int s0b = outer ? xh : imin(ix - rex - 1, g.dim[0] - 1);   // first range: [xl, s0b]
      const int s1a = imax(ix + rex + 1, 0);                          // second range (inner rows): [s1a, xh]
      const bool v0 = in && xl <= s0b;
      const bool v1 = in && !outer && s1a <= xh;
      a[u][0] = cell_start[rowbase + (v0 ? xl : 0)];  b[u][0] = cell_start[rowbase + (v0 ? s0b + 1 : 0)];
      a[u][1] = cell_start[rowbase + (v1 ? s1a : 0)]; b[u][1] = cell_start[rowbase + (v1 ? xh + 1 : 0)];
      if (!v0) b[u][0] = a[u][0];
      if (!v1) b[u][1] = a[u][1];
    }
#pragma unroll
    for (int u = 0; u < kKnn3RingBatch; ++u) {
      ok = knn3_push<SEGBITS>(tab, tstride, nseg, a[u][0], b[u][0]) && ok;
      ok = knn3_push<SEGBITS>(tab, tstride, nseg, a[u][1], b[u][1]) && ok;
    }
  }
  return ok;
}

// stage B: the rows of the box of the ball (q, sqrt(lim2)) outside the cube of rings <= rex, slab-tested and cut to
// the chord, appended to the table.  false: the table is full.
template <int SEGBITS>
S3D_HD bool knn3_build_box(const GridParams& g, const uint32_t* __restrict__ cell_start, float qx, float qy, float qz,
                           int ix, int iy, int iz, float lim2, int rex, uint32_t* tab, int tstride, int& nseg) {
  const float eps = 2.0e-3f * g.h;
  const float R = sqrt_bound(lim2) * 1.0001f + eps;
  const int z0 = imax(grid_coord(g, 2, qz - R), 0), z1 = imin(grid_coord(g, 2, qz + R), g.dim[2] - 1);
  const int y0 = imax(grid_coord(g, 1, qy - R), 0), y1 = imin(grid_coord(g, 1, qy + R), g.dim[1] - 1);
  const int ny = y1 - y0 + 1, nrows = (y0 <= y1 && z0 <= z1) ? ny * (z1 - z0 + 1) : 0;
  if (ny > 64 || nrows > 4096) return false;      // (the caller bounds the ball to a few rings: never on that path)
  const int magic = small_div_magic(imax(ny, 1));
  bool ok = true;
  for (int t0 = 0; t0 < nrows; t0 += kKnn3RingBatch) {
    uint32_t a[kKnn3RingBatch][2], b[kKnn3RingBatch][2];
#pragma unroll
    for (int u = 0; u < kKnn3RingBatch; ++u) {
      const int t = t0 + u;
      const int tz = small_div(t, magic);
      const int cy = y0 + (t - tz * ny), cz = z0 + tz;
      const float zlo = g.origin[2] + (float)cz * g.h, ylo = g.origin[1] + (float)cy * g.h;
      const float fz2 = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
      const float fy2 = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
      const float rowd2 = fy2 * fy2 + fz2 * fz2;
      const bool in = t < nrows && rowd2 <= lim2;
      const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
      const int xa = imax(grid_coord(g, 0, qx - rx), 0), xb = imin(grid_coord(g, 0, qx + rx), g.dim[0] - 1);
      const bool inner = cy >= iy - rex && cy <= iy + rex && cz >= iz - rex && cz <= iz + rex;
      // outside the examined cube: [xa, xb]; inside its (y, z) shadow: what lies left and right of it
      const int s0b = inner ? imin(xb, ix - rex - 1) : xb;
      const int s1a = imax(xa, ix + rex + 1);
      const bool v0 = in && xa <= s0b, v1 = in && inner && s1a <= xb;
      const int rowbase = in ? g.dim[0] * (cy + g.dim[1] * cz) : 0;
      a[u][0] = cell_start[rowbase + (v0 ? xa : 0)];  b[u][0] = cell_start[rowbase + (v0 ? s0b + 1 : 0)];
      a[u][1] = cell_start[rowbase + (v1 ? s1a : 0)]; b[u][1] = cell_start[rowbase + (v1 ? xb + 1 : 0)];
      if (!v0) b[u][0] = a[u][0];
      if (!v1) b[u][1] = a[u][1];
    }
#pragma unroll
    for (int u = 0; u < kKnn3RingBatch; ++u) {
      ok = knn3_push<SEGBITS>(tab, tstride, nseg, a[u][0], b[u][0]) && ok;
      ok = knn3_push<SEGBITS>(tab, tstride, nseg, a[u][1], b[u][1]) && ok;
    }
  }
  return ok;
}

// 0: keys[0..K-1] hold the K nearest (K = KL - 1), places through knn3_position<SEGBITS>; not answered: 1 = a tie band
// at the K-th place or a full table (an ordinary point: the exact per-lane search is quick), 2 = the K-th neighbour lies
// beyond `rmax` rings (an ISOLATED point, whose exact search is a walk over thousands of rows: wave-cooperatively).
// tab: 2^SEGBITS entries of this lane, stride tstride.
template <int KL, int SEGBITS, typename F4T>
S3D_HD int grid_knn_med3_rings(const GridParams& g, const uint32_t* __restrict__ cell_start, const F4T* __restrict__ pts,
                               float qx, float qy, float qz, uint32_t* tab, int tstride, uint32_t (&keys)[KL], int rmax) {
  constexpr int K = KL - 1;
#pragma unroll
  for (int j = 0; j < KL; ++j) keys[j] = kKnn3Sentinel;
  int nseg;
  uint32_t total;
  if (!knn3_build27<SEGBITS>(g, cell_start, qx, qy, qz, tab, tstride, nseg, total)) return 1;
  knn3_scan<KL, SEGBITS>(keys, tab, tstride, 0, nseg, pts, qx, qy, qz);
  const float face = knn3_face(g, qx, qy, qz);
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  int rex = 1;                                   // the cube of rings <= rex has been examined
  while (keys[K - 1] == kKnn3Sentinel) {         // stage A: rings rex + 1 ... rnext together, rnext = kKnn3RingStep, then rmax
    if (rex >= rmax) return 2;
    const int rnext = rex < kKnn3RingStep && kKnn3RingStep < rmax ? kKnn3RingStep : rmax;
    const int first = nseg;
    if (!knn3_build_ring<SEGBITS>(g, cell_start, ix, iy, iz, rex, rnext, tab, tstride, nseg)) return 1;
    knn3_scan<KL, SEGBITS>(keys, tab, tstride, first, nseg, pts, qx, qy, qz);
    rex = rnext;
  }
  const float lim2 = knn3_key_d2_upper<SEGBITS>(keys[K - 1]);
  const float proven = ((float)rex + face) * g.h;
  if (lim2 > proven * proven) {                  // stage B
    const float reach = ((float)rmax + face) * g.h;
    if (lim2 > reach * reach) return 2;
    const int first = nseg;
    if (!knn3_build_box<SEGBITS>(g, cell_start, qx, qy, qz, ix, iy, iz, lim2, rex, tab, tstride, nseg)) return 1;
    knn3_scan<KL, SEGBITS>(keys, tab, tstride, first, nseg, pts, qx, qy, qz);
  }
  return knn3_unambiguous<KL, SEGBITS>(keys) ? 0 : 1;
}

// ------------------------------------------------------------------ K5, round 3: 1-NN by a flat scan of the 27 cells
//
// The second and third pass of a registration search (nearly) every query again - the first transform update has
// moved them all - but with the clouds roughly aligned by then the neighbour lies within one cell: the seeded box
// search above visits the same ~50 candidates of the 27 cells around the query through per-row loops (a wave runs
// max-over-lanes of EVERY row) and a 15-instruction consider() per candidate.  Here the nine row segments go into the
// per-lane LDS table of the k-NN pre-pass (knn3_build27) and ONE flat loop scans them (max-over-lanes of the total),
// keeping the best and the runner-up as packed 64-bit keys ((d2 bits + 2^23) << 32 | position, held as doubles:
// v_min_f64 / v_max_f64; the low word is the candidate's position) - no seed, no radius, no retry.  The result is proven when the neighbour is nearer than
// the 27 cells' guaranteed reach (1 + face) h, which is then also the radius the re-validation of later passes builds
// on; anything else (a query outside the grid, empty cells, a farther neighbour, a row range beyond the table) returns
// false and goes through grid_nn1_box.  Same neighbour, same float d2, same tie rule (lowest index).
// seed_d2 (3e38: none): the squared distance of a point KNOWN to exist - the previous pass's neighbour under the new
// transform, read from the copy that travels with the correspondence, no gather.  The nine rows are then cut to the
// ball of that radius (+ the re-validation shell, as grid_nn1_box does): a slab test per row, the x-range of its
// chord - a stale neighbour half a cell away leaves ~8 of the 27 cells.
// PRESCAN (no seed known - the second pass): the query's own ROW (three cells, a third of the candidates on the
// surfaces a scan consists of) is scanned first, and its best point is the seed that cuts the other eight rows.
template <int PRESCAN = 0, typename F4T>
S3D_HD bool grid_nn1_scan27(const GridParams& g, const uint32_t* __restrict__ cell_start, const F4T* __restrict__ pts,
                            float qx, float qy, float qz, uint32_t* tab, int tstride, NNResult& r,
                            float seed_d2 = 3.0e38f) {
  r.idx = -1; r.d2 = 3.0e38f; r.pos = -1; r.second_d2 = 3.0e38f; r.radius = 0.f;
  const int ix = grid_coord(g, 0, qx), iy = grid_coord(g, 1, qy), iz = grid_coord(g, 2, qz);
  if (ix < 0 || ix >= g.dim[0] || iy < 0 || iy >= g.dim[1] || iz < 0 || iz >= g.dim[2]) return false;
  const double kInf = __builtin_bit_cast(double, 0x7FDFFFFFFFFFFFFFull);
  double best = kInf, second = kInf;
#define S3D_NN27_USE(P_, K_, V_)                                                                       \
  {                                                                                                    \
    const float d2_ = dist2_xy(qx, qy, qz, P_);                                                        \
    const unsigned long long kb_ = ((unsigned long long)(__builtin_bit_cast(uint32_t, d2_) + 0x00800000u) << 32) | \
                                   (unsigned long long)K_;                                             \
    const double c_ = V_ ? __builtin_bit_cast(double, kb_) : kInf;                                     \
    second = f64_min_raw(second, f64_max_raw(c_, best));                                               \
    best = f64_min_raw(best, c_);                                                                      \
  }
  bool prescanned = false;
  const int base_own = g.dim[0] * (iy + g.dim[1] * iz);          // the query's own row
  if (PRESCAN == 2 || (PRESCAN == 1 && seed_d2 >= 1.0e30f)) {   // (2: also next to a seed, whichever is nearer)
    const int rowbase = base_own;
    const uint32_t a = cell_start[rowbase + imax(ix - 1, 0)], b = cell_start[rowbase + imin(ix + 1, g.dim[0] - 1) + 1];
    for (uint32_t k = a; k < b; k += 2) {          // two loads in flight
      const bool v1 = k + 1 < b;
      const F4T p0 = pts[k], p1 = pts[v1 ? k + 1 : k];
      const uint32_t k1 = k + 1;
      S3D_NN27_USE(p0, k, true)
      S3D_NN27_USE(p1, k1, v1)
    }
    prescanned = true;
    if (best != kInf) seed_d2 = fminf(seed_d2, knn_key_d2(best));
  }
  int nseg = 0;
  uint32_t total = 0;
  float reach = 3.0e38f;            // every point nearer than this lies in an examined cell (if it lies in the 27 at all)
  {
    const bool cut = seed_d2 < 1.0e30f;
    const float br = cut ? sqrtf(seed_d2) * 1.0001f + 1.0e-6f + kNNRevalSlack * g.h : 0.f;
    const float b2 = cut ? br * br : 3.0e38f;
    if (cut) reach = br * 0.9999f;
    const float eps = 2.0e-3f * g.h;
    // (round 5: the kernels of passes 2-3 run at 60-80 % of the VALU issue rate and this per-query set-up is more than half
    // of their instructions - the slab distances of the three y and the three z layers are computed once instead of per
    // row, the row bases by additions from the own row's instead of a 64-bit multiply-add and a 32-bit multiply, both
    // quarter-rate, per row)
    float fy2[3] = {0.f, 0.f, 0.f}, fz2[3] = {0.f, 0.f, 0.f};
    if (cut) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float ylo = g.origin[1] + (float)(iy + j - 1) * g.h, zlo = g.origin[2] + (float)(iz + j - 1) * g.h;
        const float fy = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
        const float fz = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
        fy2[j] = fy * fy; fz2[j] = fz * fz;
      }
    }
    const int xa0 = imax(ix - 1, 0), xb0 = imin(ix + 1, g.dim[0] - 1);
    const int stride_y = g.dim[0], stride_z = g.dim[0] * g.dim[1];
    uint32_t rs[9], re[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {   // nine row ranges fetched as one batch
      const int cy = iy + (k % 3) - 1, cz = iz + (k / 3) - 1;
      bool in = cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
      int xa = xa0, xb = xb0;
      if (cut) {
        const float rowd2 = fy2[k % 3] + fz2[k / 3];
        in = in && rowd2 <= b2;
        const float rx = sqrt_bound(fmaxf(b2 - rowd2, 0.f)) * 1.0001f + eps;
        xa = imax(xa, grid_coord(g, 0, qx - rx));
        xb = imin(xb, grid_coord(g, 0, qx + rx));
      }
      in = in && xa <= xb;
      if (PRESCAN && k == 4) in = in && !prescanned;          // (the own row is done)
      const int rowbase = in ? base_own + ((k % 3) - 1) * stride_y + ((k / 3) - 1) * stride_z : 0;
      const uint32_t a = cell_start[rowbase + (in ? xa : 0)], b = cell_start[rowbase + (in ? xb + 1 : 0)];
      rs[k] = a; re[k] = in ? b : a;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      ok = knn3_push(tab, tstride, nseg, rs[k], re[k]) && ok;
      total += re[k] - rs[k];
    }
    if (!ok || (total == 0 && best == kInf)) return false;
  }
  // the iterator of knn3_scan without ids: [pos, end) = what is left of the current entry
  int e = 0;
  uint32_t pos = 0, end = 0;
#define S3D_NN27_NEXT(P_, K_, V_)                                                                      \
  {                                                                                                    \
    if (pos == end && e < nseg) {                                                                      \
      const uint32_t ent = tab[e * tstride];                                                           \
      pos = ent >> kKnn3OffBits; end = pos + (ent & kKnn3OffMask) + 1u;                                \
      ++e;                                                                                             \
    }                                                                                                  \
    V_ = pos != end;                                                                                   \
    K_ = V_ ? pos : 0u;                                                                                \
    P_ = pts[K_];                                                                                      \
    pos += V_ ? 1u : 0u;                                                                               \
  }
  F4T pa, pb;
  uint32_t ka, kb;
  bool va, vb;
  S3D_NN27_NEXT(pa, ka, va)
  S3D_NN27_NEXT(pb, kb, vb)
  while (va) {
    S3D_NN27_USE(pa, ka, va)
    S3D_NN27_NEXT(pa, ka, va)
    S3D_NN27_USE(pb, kb, vb)
    S3D_NN27_NEXT(pb, kb, vb)
  }
#undef S3D_NN27_NEXT
#undef S3D_NN27_USE
  // the low word of a key is the candidate's POSITION (what the loop has at hand), while the tie rule of the search is
  // the lowest point INDEX: two candidates at the same distance are therefore not decided here (declined)
  const unsigned long long bb = __builtin_bit_cast(unsigned long long, best);
  r.pos = (int)(uint32_t)(bb & 0xFFFFFFFFull);
  r.idx = __builtin_bit_cast(int, pts[r.pos].w);
  r.d2 = knn_key_d2(best);
  if (second != kInf) r.second_d2 = knn_key_d2(second);
  const float r1 = fminf((1.0f + knn3_face(g, qx, qy, qz)) * g.h, reach);
  r.radius = r1;
  return r.d2 <= r1 * r1 && r.second_d2 != r.d2;
}

// ------------------------------------------------------------------ covariance -> normal (K4)

// cyclic Jacobi on a symmetric 3x3; returns the unit eigenvector of the SMALLEST
// eigenvalue (= third column of U in PCL's JacobiSVD of the covariance)
S3D_HD void sym3_smallest_eigvec(double a00, double a01, double a02, double a11, double a12, double a22,
                                 double n[3]) {
  double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
#define S3D_JROT(p, q)                                                                       \
  if (a[p][q] != 0.0) {                                                                      \
    double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);                                    \
    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));        \
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                           \
    for (int k = 0; k < 3; ++k) {                                                            \
      double akp = a[k][p], akq = a[k][q];                                                   \
      a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq;                              \
    }                                                                                        \
    for (int k = 0; k < 3; ++k) {                                                            \
      double apk = a[p][k], aqk = a[q][k];                                                   \
      a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk;                              \
    }                                                                                        \
    for (int k = 0; k < 3; ++k) {                                                            \
      double vkp = V[k][p], vkq = V[k][q];                                                   \
      V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;                              \
    }                                                                                        \
  }
    S3D_JROT(0, 1)
    S3D_JROT(0, 2)
    S3D_JROT(1, 2)
#undef S3D_JROT
  }
  const double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
  // same selection as a descending sort that keeps the earlier column on ties
  int m = 0;
  double dm = d0;
  if (d1 <= dm) { m = 1; dm = d1; }
  if (d2 <= dm) { m = 2; dm = d2; }
  // (ties: the LAST minimal column, matching the oracle's stable descending order)
  n[0] = m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
  n[1] = m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
  n[2] = m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

// full decomposition by the same cyclic Jacobi: eigenvalues descending, eigenvectors in the columns of V
// (row-major 3x3).  Used for the NDT voxel covariances (pcl::VoxelGridCovariance runs a SelfAdjointEigenSolver).
S3D_HD void sym3_eig_desc(double a00, double a01, double a02, double a11, double a12, double a22, double ev[3],
                          double Vout[9]) {
  double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-300 || off <= 1e-34 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - sn * akq; a[k][q] = sn * akp + c * akq; }
        for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - sn * aqk; a[q][k] = sn * apk + c * aqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
      }
  }
  int o0 = 0, o1 = 1, o2 = 2;
  const double d[3] = {a[0][0], a[1][1], a[2][2]};
  if (d[o1] > d[o0]) { const int t = o0; o0 = o1; o1 = t; }
  if (d[o2] > d[o0]) { const int t = o0; o0 = o2; o2 = t; }
  if (d[o2] > d[o1]) { const int t = o1; o1 = o2; o2 = t; }
  ev[0] = d[o0]; ev[1] = d[o1]; ev[2] = d[o2];
  for (int r = 0; r < 3; ++r) { Vout[r * 3] = V[r][o0]; Vout[r * 3 + 1] = V[r][o1]; Vout[r * 3 + 2] = V[r][o2]; }
}

// Closed-form variant for the common, well-conditioned case (a surface patch: lambda_min well separated):
// eigenvalues by the trigonometric solution of the characteristic cubic, eigenvector = the largest cross
// product of two rows of (A - lambda_min I).  ~4x fewer operations than the Jacobi sweeps and no loop.
// Returns false — caller falls back to the Jacobi iteration, whose tie rules the oracle shares — when the
// matrix is (nearly) isotropic or lambda_min is not separated from the middle eigenvalue by 1e-6 of the
// spread: there the closed form loses digits and the choice of vector is a convention, not a number.
S3D_HD bool sym3_smallest_eigvec_direct(double a00, double a01, double a02, double a11, double a12, double a22,
                                        double n[3]) {
  const double scale = fmax(fmax(fabs(a00), fabs(a11)), fmax(fmax(fabs(a22), fabs(a01)), fmax(fabs(a02), fabs(a12))));
  if (!(scale > 1e-300)) return false;
  const double is = 1.0 / scale;
  a00 *= is; a01 *= is; a02 *= is; a11 *= is; a12 *= is; a22 *= is;
  const double q = (a00 + a11 + a22) / 3.0;
  const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
  const double pp = sqrt(p2 / 6.0);
  if (!(pp > 1e-9)) return false;
  const double ip = 1.0 / pp;
  const double c00 = b00 * ip, c01 = a01 * ip, c02 = a02 * ip, c11 = b11 * ip, c12 = a12 * ip, c22 = b22 * ip;
  double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
  r = fmin(fmax(r, -1.0), 1.0);
  // phi = acos(r) / 3 lies in [0, pi / 3]: cos(phi + 2 pi / 3) = -cos(phi) / 2 - sin(phi) sqrt(3) / 2 with
  // sin(phi) = sqrt(1 - cos^2(phi)) >= 0 - one cosine and a square root instead of two cosines
  const double phi = acos(r) * 0.33333333333333333;
  const double cphi = cos(phi);
  const double sphi = sqrt(fmax(fma(-cphi, cphi, 1.0), 0.0));
  const double emax = fma(2.0 * pp, cphi, q);
  const double emin = fma(2.0 * pp, fma(-0.8660254037844386, sphi, -0.5 * cphi), q);
  const double emid = 3.0 * q - emax - emin;
  if (!(emid - emin > 1e-6 * (emax - emin))) return false;
  // rows of A - emin I
  const double r0[3] = {a00 - emin, a01, a02}, r1[3] = {a01, a11 - emin, a12}, r2[3] = {a02, a12, a22 - emin};
  const double x0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
  const double x1[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
  const double x2[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  const double m0 = x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2];
  const double m1 = x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2];
  const double m2 = x2[0] * x2[0] + x2[1] * x2[1] + x2[2] * x2[2];
  double v0 = x0[0], v1 = x0[1], v2 = x0[2], mm = m0;
  if (m1 > mm) { v0 = x1[0]; v1 = x1[1]; v2 = x1[2]; mm = m1; }
  if (m2 > mm) { v0 = x2[0]; v1 = x2[1]; v2 = x2[2]; mm = m2; }
  if (!(mm > 1e-24)) return false;
  const double inv = 1.0 / sqrt(mm);
  n[0] = v0 * inv; n[1] = v1 * inv; n[2] = v2 * inv;
  return true;
}

// ---- the stored form of a unit normal: 16 bytes ----
// Covariances are stored as one unit normal per point (C = I - (1 - eps) n n^T).  Three floats lose the direction at
// the 6e-8 level, and PCL's GICP notices: the oracle run on float-rounded normals differs from the oracle on double
// normals by up to 1.6e-4 m / one outer iteration on the reference's fixtures once correspondence_randomness is 40
// (tools_dev/kdens.py; 2e-8 m at the default k = 20).  The record therefore carries, next to the float components,
// the remainders n - float(n) in units of 2^-34 as three signed 10-bit fields of a fourth word: |remainder| <= 2^-25,
// so |q| <= 512 (a +512 is stored as 511), and a component is restored to 2^-34 = 6e-11 with one convert and one fma.  16 bytes are also the friendlier
// access (one dwordx4) than the 12-byte record they replace.
struct NormalRec { float x, y, z; uint32_t lo; };

S3D_HD NormalRec normal_encode(const double n[3]) {
  NormalRec r;
  float* f = &r.x;
  uint32_t lo = 0;
  for (int a = 0; a < 3; ++a) {
    f[a] = (float)n[a];
    double q = (n[a] - (double)f[a]) * 17179869184.0;   // 2^34
    int qi = (int)(q < 0 ? q - 0.5 : q + 0.5);
    qi = qi < -512 ? -512 : (qi > 511 ? 511 : qi);
    lo |= ((uint32_t)qi & 0x3FFu) << (10 * a);
  }
  r.lo = lo;
  return r;
}
S3D_HD void normal_decode(const NormalRec& r, double n[3]) {
  const float* f = &r.x;
  for (int a = 0; a < 3; ++a) {
    const int qi = ((int)(r.lo << (22 - 10 * a))) >> 22;   // sign-extended 10-bit field
    n[a] = fma((double)qi, 5.8207660913467407e-11 /* 2^-34 */, (double)f[a]);
  }
}

// PCL computeCovariances moments of the k neighbours: float products, double sums
struct Moments { double mean[3]; double c00, c10, c11, c20, c21, c22; };
S3D_HD void moments_init(Moments& m) {
  m.mean[0] = m.mean[1] = m.mean[2] = 0; m.c00 = m.c10 = m.c11 = m.c20 = m.c21 = m.c22 = 0;
}
S3D_HD void moments_add(Moments& m, float x, float y, float z) {
  m.mean[0] += x; m.mean[1] += y; m.mean[2] += z;
  m.c00 += x * x; m.c10 += y * x; m.c11 += y * y; m.c20 += z * x; m.c21 += z * y; m.c22 += z * z;
}
// covariance of the k neighbours from their sums: c = {c00, c10, c11, c20, c21, c22}
S3D_HD void moments_covariance(const Moments& m, int k, double c[6]) {
  const double kd = (double)k;
  const double mx = m.mean[0] / kd, my = m.mean[1] / kd, mz = m.mean[2] / kd;
  c[0] = m.c00 / kd - mx * mx; c[1] = m.c10 / kd - my * mx; c[2] = m.c11 / kd - my * my;
  c[3] = m.c20 / kd - mz * mx; c[4] = m.c21 / kd - mz * my; c[5] = m.c22 / kd - mz * mz;
}
// the closed form alone (what the k-NN kernel runs in place; false: the caller hands the point to moments_normal)
S3D_HD bool moments_normal_direct(const Moments& m, int k, double n[3]) {
#if defined(S3D_EIG_JACOBI_ONLY)
  return false;
#else
  double c[6];
  moments_covariance(m, k, c);
  return sym3_smallest_eigvec_direct(c[0], c[1], c[3], c[2], c[4], c[5], n);
#endif
}
S3D_HD void moments_normal(const Moments& m, int k, double n[3]) {
  double c[6];
  moments_covariance(m, k, c);
#if defined(S3D_EIG_JACOBI_ONLY)
  sym3_smallest_eigvec(c[0], c[1], c[3], c[2], c[4], c[5], n);
#else
  if (!sym3_smallest_eigvec_direct(c[0], c[1], c[3], c[2], c[4], c[5], n))
    sym3_smallest_eigvec(c[0], c[1], c[3], c[2], c[4], c[5], n);
#endif
}

// ------------------------------------------------------------------ GICP quadratic form (K6/K7)
//
//   f(x) * m = sum_i (Theta P_i - q_i)^T M_i (Theta P_i - q_i),  Theta = [R(x) | t(x)],  P = (p,1)
//            = sum_{a,b,c,d} Theta_ca Theta_db A[ab][cd] - 2 sum_{c,a} Theta_ca B[c][a] + C0
//   A[ab][cd] = sum_i P_a P_b M_cd   (10 x 6),  B[c][a] = sum_i (M q)_c P_a  (3 x 4),  C0 = sum q^T M q
//
// Layout of the 76-double accumulator record:
//   [0..59]  A[pair(a,b)][sym(c,d)]   pair order (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
//                                     sym  order 00 01 02 11 12 22
//   [60..71] B[c*4+a]   [72] C0   [73] number of correspondences   [74..75] spare
enum { GQ_NACC = 76, GQ_B = 60, GQ_C0 = 72, GQ_CNT = 73 };

S3D_HD int gq_pair(int a, int b) {  // a <= b, a,b in 0..3
  return a * 4 - (a * (a - 1)) / 2 + (b - a);
}
S3D_HD int gq_sym(int c, int d) {  // c <= d in 0..2
  return c * 3 - (c * (c - 1)) / 2 + (d - c);
}

// Mahalanobis matrix of one correspondence (PCL gicp.hpp computeTransformation):
//   M = (R C1 R^T + C2)^-1,  C = I - (1-eps) n n^T  (== U diag(1,1,eps) U^T)
// S = R R^T (sym, 6 values: 00 01 02 11 12 22), n1r = R n1 (already rotated), n2 unit.
// (Products and sums are written as explicit fma(): the library is compiled with -ffp-contract=off for the float code
// that has to round like the reference stack; this double arithmetic has no reference bit pattern - PCL's depends on
// Eigen's vectorisation - and un-fused it was 47 of the accumulate kernel's 213 double-precision instructions.)
S3D_HD void gicp_mahalanobis(const double S[6], const double n1r[3], const double n2[3], double eps, double M[6]) {
  const double w = 1.0 - eps;
  const double t00 = fma(-w, fma(n2[0], n2[0], n1r[0] * n1r[0]), S[0] + 1.0);
  const double t01 = fma(-w, fma(n2[0], n2[1], n1r[0] * n1r[1]), S[1]);
  const double t02 = fma(-w, fma(n2[0], n2[2], n1r[0] * n1r[2]), S[2]);
  const double t11 = fma(-w, fma(n2[1], n2[1], n1r[1] * n1r[1]), S[3] + 1.0);
  const double t12 = fma(-w, fma(n2[1], n2[2], n1r[1] * n1r[2]), S[4]);
  const double t22 = fma(-w, fma(n2[2], n2[2], n1r[2] * n1r[2]), S[5] + 1.0);
  // symmetric cofactor inverse
  const double c00 = fma(t11, t22, -(t12 * t12));
  const double c01 = fma(t02, t12, -(t01 * t22));
  const double c02 = fma(t01, t12, -(t02 * t11));
  const double det = fma(t00, c00, fma(t01, c01, t02 * c02));
  const double id = 1.0 / det;
  M[0] = c00 * id; M[1] = c01 * id; M[2] = c02 * id;
  M[3] = fma(t00, t22, -(t02 * t02)) * id;
  M[4] = fma(t01, t02, -(t00 * t12)) * id;
  M[5] = fma(t00, t11, -(t01 * t01)) * id;
}

// add one correspondence (p = guess-transformed query, q = matched target, M sym) to acc[76].
// The form is expanded ABOUT THE CURRENT TRANSFORM Th0 (= double(transformation_) at the start of the
// outer iteration): with r = Th0 P - q and Theta = Th0 + dTheta,
//   m f = sum_{abcd} dTheta_ca dTheta_db A[ab][cd] + 2 sum_{ca} dTheta_ca B[c][a] + C0,
//   B[c][a] = sum (M r)_c P_a,  C0 = sum r^T M r.
// Expanding about the origin instead (r = -q) is the same algebra but loses ~7 digits to
// cancellation (|A Theta^2| ~ 1e11 against m f ~ 1e3), enough to disturb PCL's line search.
S3D_HD void gq_accumulate(double* acc, const double p[3], const double qt[3], const double M[6], const double* Th0) {
  const double P[4] = {p[0], p[1], p[2], 1.0};
  double q[3];  // q := -(r) so that the code below accumulates with the residual: B <- -(M r) P, see gq_eval
  for (int c = 0; c < 3; ++c)
    q[c] = qt[c] - fma(Th0[c * 4 + 0], P[0], fma(Th0[c * 4 + 1], P[1], fma(Th0[c * 4 + 2], P[2], Th0[c * 4 + 3])));
  int o = 0;
  for (int a = 0; a < 4; ++a)
    for (int b = a; b < 4; ++b) {
      const double pp = P[a] * P[b];
      for (int s = 0; s < 6; ++s) acc[o + s] = fma(pp, M[s], acc[o + s]);
      o += 6;
    }
  const double Mq0 = fma(M[0], q[0], fma(M[1], q[1], M[2] * q[2]));
  const double Mq1 = fma(M[1], q[0], fma(M[3], q[1], M[4] * q[2]));
  const double Mq2 = fma(M[2], q[0], fma(M[4], q[1], M[5] * q[2]));
  for (int a = 0; a < 4; ++a) {
    acc[GQ_B + 0 * 4 + a] = fma(Mq0, P[a], acc[GQ_B + 0 * 4 + a]);
    acc[GQ_B + 1 * 4 + a] = fma(Mq1, P[a], acc[GQ_B + 1 * 4 + a]);
    acc[GQ_B + 2 * 4 + a] = fma(Mq2, P[a], acc[GQ_B + 2 * 4 + a]);
  }
  acc[GQ_C0] = fma(q[0], Mq0, fma(q[1], Mq1, fma(q[2], Mq2, acc[GQ_C0])));
  acc[GQ_CNT] += 1.0;
}

// PCL applyState on an identity base: Theta = [float(Rz Ry Rx) | float(x0..2)]
S3D_HD void gicp_apply_state(const double x[6], Mat4f& T) {
  const double cphi = cos(x[3]), sphi = sin(x[3]);
  const double cth = cos(x[4]), sth = sin(x[4]);
  const double cpsi = cos(x[5]), spsi = sin(x[5]);
  T = mat4f_identity();
  S3D_M(T, 0, 0) = (float)(cpsi * cth);
  S3D_M(T, 0, 1) = (float)(cpsi * sth * sphi - spsi * cphi);
  S3D_M(T, 0, 2) = (float)(cpsi * sth * cphi + spsi * sphi);
  S3D_M(T, 1, 0) = (float)(spsi * cth);
  S3D_M(T, 1, 1) = (float)(spsi * sth * sphi + cpsi * cphi);
  S3D_M(T, 1, 2) = (float)(spsi * sth * cphi - cpsi * sphi);
  S3D_M(T, 2, 0) = (float)(-sth);
  S3D_M(T, 2, 1) = (float)(cth * sphi);
  S3D_M(T, 2, 2) = (float)(cth * cphi);
  S3D_M(T, 0, 3) = (float)x[0];
  S3D_M(T, 1, 3) = (float)x[1];
  S3D_M(T, 2, 3) = (float)x[2];
}

// sin and cos of an Euler angle of the BFGS state.  The state is a correction about the current
// transform, so |x| is far below pi/4 in every evaluation that matters: there the fdlibm kernel
// polynomials (|error| < 1 ulp on [-pi/4, pi/4], no argument reduction) replace the library calls,
// which are ~10x longer dependent chains for the single wave that runs the optimiser.
S3D_HD void gq_sincos(double x, double* s, double* c) {
  if (fabs(x) > 0.78) { *s = sin(x); *c = cos(x); return; }
  const double z = x * x;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                               2.75573137070700676789e-06), -1.98412698298579493134e-04),
                             8.33333333332248946124e-03), -1.66666666666666324348e-01);
  *s = fma(x * z, ps, x);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                               -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                             -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  *c = fma(z * z, pc, fma(z, -0.5, 1.0));
}

// f(x) and gradient from the quadratic form (PCL OptimizationFunctorWithIndices::fdf)
// Th0: the expansion point used by gq_accumulate (3x4 row-major).
// On the GPU this is called by ALL 64 lanes of the controller's wave with identical arguments (the
// BFGS around it runs redundantly on every lane, so control flow stays uniform): the three sincos and
// the twelve 12-term dot products of G are then spread over the lanes and exchanged by shuffles,
// which makes one evaluation ~3x shorter than the scalar form the CPU emulation uses.
#if defined(__HIP_DEVICE_COMPILE__)
// the value of lane `LANE` (a compile-time constant) in every lane: two v_readlane_b32 - scalar moves, no trip through
// the LDS crossbar and no wait for it, which is what a __shfl costs the one wave that evaluates the form
template <int LANE>
__device__ __forceinline__ double gq_bcast(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
  return __hiloint2double(hi, lo);
}
#endif
S3D_HD void gq_eval(const double* acc, const double* Th0, const double x[6], double* f, double g[6]) {
  double cphi, sphi, cth, sth, cpsi, spsi;
#if defined(__HIP_DEVICE_COMPILE__)
  const int wl = (int)(threadIdx.x & 63);
  {
    const int j = wl % 3;
    const double ang = j == 0 ? x[3] : (j == 1 ? x[4] : x[5]);
    double sv, cv;
    gq_sincos(ang, &sv, &cv);
    sphi = gq_bcast<0>(sv); cphi = gq_bcast<0>(cv);
    sth = gq_bcast<1>(sv);  cth = gq_bcast<1>(cv);
    spsi = gq_bcast<2>(sv); cpsi = gq_bcast<2>(cv);
  }
#else
  gq_sincos(x[3], &sphi, &cphi); gq_sincos(x[4], &sth, &cth); gq_sincos(x[5], &spsi, &cpsi);
#endif
  // Theta = [Rz(x5) Ry(x4) Rx(x3) | x0..2] carried in DOUBLE.  PCL's applyState rounds
  // Theta to float before every evaluation, which turns f(x) into a 1e-7-level staircase
  // and makes the line search terminate on rounding noise ("conditioning");
  // the float rounding is applied once per outer iteration instead (gicp_apply_state),
  // where PCL stores transformation_ as Matrix4f.
  double Th[3][4];
  Th[0][0] = cpsi * cth; Th[0][1] = fma(cpsi * sth, sphi, -(spsi * cphi)); Th[0][2] = fma(cpsi * sth, cphi, spsi * sphi);
  Th[1][0] = spsi * cth; Th[1][1] = fma(spsi * sth, sphi, cpsi * cphi); Th[1][2] = fma(spsi * sth, cphi, -(cpsi * sphi));
  Th[2][0] = -sth; Th[2][1] = cth * sphi; Th[2][2] = cth * cphi;
  Th[0][3] = x[0]; Th[1][3] = x[1]; Th[2][3] = x[2];
  for (int c = 0; c < 3; ++c)
    for (int a = 0; a < 4; ++a) Th[c][a] -= Th0[c * 4 + a];  // dTheta about the expansion point
  // G[a][c] = sum_{d,b} Th[d][b] A[ab][cd] - B[c][a]   ( = sum_i P_a (M res_i)_c )
  double G[4][3];
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int l12 = wl < 12 ? wl : 0, a = l12 / 3, c = l12 % 3;
    double sgc = 0.0;
    for (int b = 0; b < 4; ++b) {
      const int pr = a <= b ? gq_pair(a, b) : gq_pair(b, a);
      for (int d = 0; d < 3; ++d) {
        const int sm = c <= d ? gq_sym(c, d) : gq_sym(d, c);
        // Th[d][b] with lane-dependent (d, b) only through the loop counters: static indices
        sgc = fma(Th[d][b], acc[pr * 6 + sm], sgc);
      }
    }
    sgc -= acc[GQ_B + c * 4 + a];
    G[0][0] = gq_bcast<0>(sgc); G[0][1] = gq_bcast<1>(sgc); G[0][2] = gq_bcast<2>(sgc);
    G[1][0] = gq_bcast<3>(sgc); G[1][1] = gq_bcast<4>(sgc); G[1][2] = gq_bcast<5>(sgc);
    G[2][0] = gq_bcast<6>(sgc); G[2][1] = gq_bcast<7>(sgc); G[2][2] = gq_bcast<8>(sgc);
    G[3][0] = gq_bcast<9>(sgc); G[3][1] = gq_bcast<10>(sgc); G[3][2] = gq_bcast<11>(sgc);
  }
#else
  for (int a = 0; a < 4; ++a)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int b = 0; b < 4; ++b) {
        const int pr = a <= b ? gq_pair(a, b) : gq_pair(b, a);
        for (int d = 0; d < 3; ++d) {
          const int sm = c <= d ? gq_sym(c, d) : gq_sym(d, c);
          s = fma(Th[d][b], acc[pr * 6 + sm], s);
        }
      }
      G[a][c] = s - acc[GQ_B + c * 4 + a];
    }
#endif
  double fsum = 0.0;
  for (int a = 0; a < 4; ++a)
    for (int c = 0; c < 3; ++c) fsum = fma(Th[c][a], G[a][c] - acc[GQ_B + c * 4 + a], fsum);
  const double m = acc[GQ_CNT];
  *f = (fsum + acc[GQ_C0]) / m;
  const double sc = 2.0 / m;
  g[0] = G[3][0] * sc; g[1] = G[3][1] * sc; g[2] = G[3][2] * sc;
  double Rs[3][3];
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) Rs[a][c] = G[a][c] * sc;
  // PCL computeRDerivative: g[3+k] = sum_ij dR_k(j,i) Rs(i,j)
  const double dPhi[3][3] = {{0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth},
                             {0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth},
                             {0, cphi * cth, -cth * sphi}};
  const double dTh[3][3] = {{-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth},
                            {-spsi * sth, cth * sphi * spsi, cphi * cth * spsi},
                            {-cth, -sphi * sth, -cphi * sth}};
  const double dPsi[3][3] = {{-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth},
                             {cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth},
                             {0, 0, 0}};
  double g3 = 0, g4 = 0, g5 = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      g3 = fma(dPhi[j][i], Rs[i][j], g3);
      g4 = fma(dTh[j][i], Rs[i][j], g4);
      g5 = fma(dPsi[j][i], Rs[i][j], g5);
    }
  g[3] = g3; g[4] = g4; g[5] = g5;
}

// ---- BFGS (PCL registration/bfgs.h == GSL vector_bfgs2 + Fletcher line search)
// on the quadratic form.  Every evaluation yields f and the gradient together
// (they are O(1) here), which returns the same values PCL's cached
// applyF/applyDF/applyFDF would.
struct Bfgs {
  const double* acc;
  const double* th0;
  double f, gradient[6];
  double x0[6], g0[6], p[6];
  double g0norm, pnorm, fp0, delta_f;
  // cache of the last evaluated line-search point
  double c_alpha, c_f, c_df, c_x[6], c_g[6];
  int evals;
};
enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };

S3D_HD double v6norm(const double* v) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s = fma(v[i], v[i], s);
  return sqrt(s);
}
S3D_HD double v6dot(const double* a, const double* b) {
  double s = 0;
  for (int i = 0; i < 6; ++i) s = fma(a[i], b[i], s);
  return s;
}
S3D_HD void bfgs_eval(Bfgs& b, double alpha) {
  if (alpha == b.c_alpha) return;
  for (int i = 0; i < 6; ++i) b.c_x[i] = fma(alpha, b.p[i], b.x0[i]);
  gq_eval(b.acc, b.th0, b.c_x, &b.c_f, b.c_g);
  b.c_df = v6dot(b.c_g, b.p);
  b.c_alpha = alpha;
  b.evals++;
}

S3D_HD double poly3(const double c[4], double z) { return c[0] + z * (c[1] + z * (c[2] + z * c[3])); }
S3D_HD void check_extremum(const double c[4], double z, double* zmin, double* fmin) {
  double y = poly3(c, z);
  if (y < *fmin) { *zmin = z; *fmin = y; }
}
S3D_HD double bfgs_interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin,
                               double xmax, int order) {
  double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a);
  if (ymin > ymax) { double t = ymin; ymin = ymax; ymax = t; }
  if (order > 2 && !(fpb != fpb) && fpb != INFINITY) {
    fpa = fpa * (b - a);
    fpb = fpb * (b - a);
    const double eta = 3 * (fb - fa) - 2 * fpa - fpb;
    const double xi = fpa + fpb - 2 * (fb - fa);
    const double c[4] = {fa, fpa, eta, xi};
    y = ymin;
    double fmin = poly3(c, ymin);
    check_extremum(c, ymax, &y, &fmin);
    const double A = 3 * c[3], B = 2 * c[2], C = c[1];
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
    double c = 2 * (fb - fa - fpa);
    y = ymin;
    double fmin = fl;
    if (fh < fmin) { y = ymax; fmin = fh; }
    if (c > a) {  // sic: PCL bfgs.h (GSL: c > 0)
      double z = -fpa / c;
      if (z > ymin && z < ymax) {
        double f = fa + z * (fpa + z * (fb - fa - fpa));
        if (f < fmin) { y = z; fmin = f; }
      }
    }
  }
  return a + y * (b - a);
}

// Fletcher's line search (GSL / PCL bfgs.h: a bracketing loop, then a sectioning loop, at most 100 iterations between
// them).  Written as ONE loop with ONE evaluation site: f and the gradient come out of the same gq_eval, so PCL's
// applyF(alpha) followed by applyDF(alpha) is one evaluation here (its cache makes it one there), and the two phases
// differ only in where alpha comes from and what is done with the result.  (Round 4: as two loops with bfgs_f / bfgs_df
// calls the inlined controller held six copies of gq_eval - 9 200 instructions, more than the instruction cache of a
// compute unit pair - and the one wave that runs it spent most of an evaluation's 1.8 us waiting for code.)
// Same sequence of trial points, same decisions, same evaluation count as the two-loop form.
S3D_HD int bfgs_line_search(Bfgs& B, double rho, double sigma, double tau1, double tau2, double tau3, int order,
                            double alpha1, double* alpha_new) {
  const int bracket_iters = 100, section_iters = 100;
  const double f0 = B.f, fp0 = B.fp0;   // == applyFDF(0): position 0 is x0 with cached f, slope
  double falpha_prev = f0, fpalpha_prev = fp0;
  double alpha = alpha1, alpha_prev = 0.0;
  double a = 0.0, b = alpha, fa = f0, fb = 0.0, fpa = fp0, fpb = 0.0;
  int i = 0;
  bool section = false;
  for (;;) {
    if (!section) {
      if (!(i++ < bracket_iters)) { section = true; continue; }
    } else {
      if (!(i++ < section_iters)) return BFGS_SUCCESS;
      const double delta = b - a;
      const double lower = a + tau2 * delta, upper = b - tau3 * delta;
      alpha = bfgs_interpolate(a, fa, fpa, b, fb, fpb, lower, upper, order);
    }
    bfgs_eval(B, alpha);                       // the one evaluation site
    const double falpha = B.c_f, fpalpha = B.c_df;
    if (!section) {
      if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = NAN;
        section = true;
        continue;
      }
      if (fabs(fpalpha) <= -sigma * fp0) { *alpha_new = alpha; return BFGS_SUCCESS; }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        section = true;
        continue;
      }
      const double delta = alpha - alpha_prev;
      const double lower = alpha + delta, upper = alpha + tau1 * delta;
      const double alpha_next = bfgs_interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, lower, upper, order);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha; alpha = alpha_next;
    } else {
      if ((a - alpha) * fpa <= 2.220446049250313e-16) return BFGS_NOPROGRESS;
      if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
        b = alpha; fb = falpha; fpb = NAN;
      } else {
        if (fabs(fpalpha) <= -sigma * fp0) { *alpha_new = alpha; return BFGS_SUCCESS; }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
          a = alpha; fa = falpha; fpa = fpalpha;
        } else {
          a = alpha; fa = falpha; fpa = fpalpha;
        }
      }
    }
  }
}

S3D_HD void bfgs_init(Bfgs& b, const double* acc, const double* th0, const double x[6]) {
  b.acc = acc;
  b.th0 = th0;
  b.delta_f = 0;
  b.evals = 1;
  gq_eval(acc, th0, x, &b.f, b.gradient);
  for (int i = 0; i < 6; ++i) { b.x0[i] = x[i]; b.g0[i] = b.gradient[i]; }
  b.g0norm = v6norm(b.g0);
  for (int i = 0; i < 6; ++i) b.p[i] = b.gradient[i] * -1 / b.g0norm;
  b.pnorm = v6norm(b.p);
  b.fp0 = -b.g0norm;
  b.c_alpha = 0.0; b.c_f = b.f; b.c_df = v6dot(b.g0, b.p);
  for (int i = 0; i < 6; ++i) { b.c_x[i] = b.x0[i]; b.c_g[i] = b.g0[i]; }
}

S3D_HD int bfgs_one_step(Bfgs& b, double x[6]) {
  const double sigma = 0.01, rho = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1.0;
  const int order = 3;
  double alpha = 0.0, alpha1;
  const double f0 = b.f;
  if (b.pnorm == 0.0 || b.g0norm == 0.0 || b.fp0 == 0) return BFGS_NOPROGRESS;
  if (b.delta_f < 0) {
    double del = fmax(-b.delta_f, 10 * 2.220446049250313e-16 * fabs(f0));
    alpha1 = fmin(1.0, 2.0 * del / (-b.fp0));
  } else {
    alpha1 = fabs(step_size);
  }
  int status = bfgs_line_search(b, rho, sigma, tau1, tau2, tau3, order, alpha1, &alpha);
  if (status != BFGS_SUCCESS) return status;
  // updatePosition(alpha)
  if (alpha == 0.0) {  // (section loop ran out without setting alpha: position unchanged)
    for (int i = 0; i < 6; ++i) { x[i] = b.x0[i]; b.gradient[i] = b.g0[i]; }
  } else {
    // (alpha is the trial point the line search evaluated last - it returns an alpha only right after evaluating it -
    // so PCL's updatePosition(alpha) finds it in the cache: no evaluation site here)
    b.f = b.c_f;
    for (int i = 0; i < 6; ++i) { x[i] = b.c_x[i]; b.gradient[i] = b.c_g[i]; }
  }
  b.delta_f = b.f - f0;
  double dx0[6], dg0[6];
  for (int i = 0; i < 6; ++i) { dx0[i] = x[i] - b.x0[i]; dg0[i] = b.gradient[i] - b.g0[i]; }
  const double dxg = v6dot(dx0, b.gradient), dgg = v6dot(dg0, b.gradient), dxdg = v6dot(dx0, dg0);
  const double dgnorm = v6norm(dg0);
  double A, Bc;
  if (dxdg != 0) {
    Bc = dxg / dxdg;
    A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
  } else {
    Bc = 0; A = 0;
  }
  for (int i = 0; i < 6; ++i) b.p[i] = -A * dx0[i] + b.gradient[i] - Bc * dg0[i];
  for (int i = 0; i < 6; ++i) { b.g0[i] = b.gradient[i]; b.x0[i] = x[i]; }
  b.g0norm = v6norm(b.g0);
  b.pnorm = v6norm(b.p);
  const double dir = (v6dot(b.p, b.gradient) > 0) ? -1.0 : 1.0;
  for (int i = 0; i < 6; ++i) b.p[i] *= dir / b.pnorm;
  b.pnorm = v6norm(b.p);
  b.fp0 = v6dot(b.p, b.g0);
  // changeDirection(): cache is position 0 of the new line
  b.c_alpha = 0.0; b.c_f = b.f; b.c_df = b.fp0;
  for (int i = 0; i < 6; ++i) { b.c_x[i] = b.x0[i]; b.c_g[i] = b.g0[i]; }
  return BFGS_SUCCESS;
}

// PCL estimateRigidTransformationBFGS on the quadratic form.
// returns 0 and updates T, or -1 (the PCLException path: < 4 pairs / solver failure)
S3D_HD int gicp_estimate_bfgs(const double* acc, int max_inner, Mat4f& T, int* inner_out, int* evals_out) {
  *inner_out = 0; *evals_out = 0;
  if (acc[GQ_CNT] < 4.0) return -1;
  double x[6];
  x[0] = S3D_M(T, 0, 3); x[1] = S3D_M(T, 1, 3); x[2] = S3D_M(T, 2, 3);
  x[3] = atan2((double)S3D_M(T, 2, 1), (double)S3D_M(T, 2, 2));
  x[4] = asin(-(double)S3D_M(T, 2, 0));
  x[5] = atan2((double)S3D_M(T, 1, 0), (double)S3D_M(T, 0, 0));
  const double gradient_tol = 1e-2;
  double th0[12];  // expansion point of the form = the transform the accumulate pass used
  for (int c = 0; c < 3; ++c)
    for (int a = 0; a < 4; ++a) th0[c * 4 + a] = (double)S3D_M(T, c, a);
  Bfgs b;
  bfgs_init(b, acc, th0, x);
  int inner = 0, result = BFGS_RUNNING;
  do {
    inner++;
    result = bfgs_one_step(b, x);
    if (result) break;
    result = v6norm(b.gradient) < gradient_tol ? BFGS_SUCCESS : BFGS_RUNNING;
  } while (result == BFGS_RUNNING && inner < max_inner);
  *inner_out = inner; *evals_out = b.evals;
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == max_inner) {
    gicp_apply_state(x, T);
    return 0;
  }
  return -1;
}

// PCL GICP outer-loop stopping quantity
S3D_HD double icp_delta(const Mat4f& prev, const Mat4f& cur, double rotation_epsilon, double transformation_epsilon) {
  double delta = 0;
  for (int k = 0; k < 4; ++k)
    for (int l = 0; l < 4; ++l) {
      const double ratio = (k < 3 && l < 3) ? 1.0 / rotation_epsilon : 1.0 / transformation_epsilon;
      const double c = ratio * fabs((double)S3D_M(prev, k, l) - (double)S3D_M(cur, k, l));
      if (c > delta) delta = c;
    }
  return delta;
}

// transform_R = double(transformation_) * double(guess); returns R (3x3 row-major) and S = R R^T (sym 6)
S3D_HD void gicp_rotation(const Mat4f& T, const Mat4f& guess, double R[9], double S[6]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += (double)S3D_M(T, i, k) * (double)S3D_M(guess, k, j);
      R[i * 3 + j] = s;
    }
  int o = 0;
  for (int a = 0; a < 3; ++a)
    for (int b = a; b < 3; ++b) S[o++] = R[a * 3 + 0] * R[b * 3 + 0] + R[a * 3 + 1] * R[b * 3 + 1] + R[a * 3 + 2] * R[b * 3 + 2];
}

// ------------------------------------------------------------------ point-to-plane (ICP enumerator)
// accumulator record: [0..20] lower triangle of J^T J (row-major a>=c), [21..26] -J^T r,
// [27] sum r^2, [28] count   (PP_NACC = 32 with padding)
enum { PP_NACC = 32, PP_B = 21, PP_R2 = 27, PP_CNT = 28 };

S3D_HD void pp_accumulate(double* acc, const double p[3], const double q[3], const double n[3]) {
  const double r = fma(n[0], p[0] - q[0], fma(n[1], p[1] - q[1], n[2] * (p[2] - q[2])));
  const double J[6] = {fma(p[1], n[2], -(p[2] * n[1])), fma(p[2], n[0], -(p[0] * n[2])), fma(p[0], n[1], -(p[1] * n[0])),
                       n[0], n[1], n[2]};
  int o = 0;
  for (int a = 0; a < 6; ++a) {
    for (int c = 0; c <= a; ++c) { acc[o] = fma(J[a], J[c], acc[o]); ++o; }
    acc[PP_B + a] = fma(-J[a], r, acc[PP_B + a]);
  }
  acc[PP_R2] = fma(r, r, acc[PP_R2]);
  acc[PP_CNT] += 1.0;
}

S3D_HD int chol6_solve(const double* accA /*lower tri, 21*/, const double* b, double x[6]) {
  double L[6][6];
  int o = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = accA[o++];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0)) return -1;
        L[i][i] = sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return 0;
}

// T <- Exp(w, dt) * T, kept in float.  returns -1 when the system is degenerate
S3D_HD int pp_update(const double* acc, Mat4f& T) {
  if (acc[PP_CNT] < 6.0) return -1;
  double xi[6];
  if (chol6_solve(acc, acc + PP_B, xi)) return -1;
  const double th2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2], th = sqrt(th2);
  double a, b;
  if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  const double K[3][3] = {{0, -xi[2], xi[1]}, {xi[2], 0, -xi[0]}, {-xi[1], xi[0], 0}};
  double R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double kk = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
      R[i][j] = (i == j ? 1.0 : 0.0) + a * K[i][j] + b * kk;
    }
  Mat4f nt = mat4f_identity();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += R[r][k] * (double)S3D_M(T, k, c);
      if (c == 3) s += xi[3 + r];
      S3D_M(nt, r, c) = (float)s;
    }
  T = nt;
  return 0;
}

}  // namespace s3d
