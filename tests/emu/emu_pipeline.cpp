// emu_pipeline.cpp — TEST-ONLY sequential emulation of the device pipeline.
//
// Runs the exact per-thread functions of slam3d_amd/csrc/s3d_core.h (the ones the
// HIP kernels call) on the CPU, with std::stable_sort standing in for the radix
// sort and sequential sums for the wave/block reductions.  It exists so that the
// GPU-side reformulations (normal-only covariances, quadratic-form GICP, BFGS on
// the form, grid NN/k-NN) can be checked against oracle/ in this GPU-less
// container.  It is NOT a back-end of the product library and is never linked
// into it.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include "../../include/slam3d_registration_types.h"
#include "../../slam3d_amd/csrc/s3d_core.h"

using namespace s3d;

static double g_emu_perturb = 0.0;

static long long g_reval_hits = 0, g_reval_misses = 0, g_reval_mismatch = 0, g_far_seeded = 0;
extern "C" void emu_reval_stats(long long* out) { out[0] = g_reval_hits; out[1] = g_reval_misses; out[2] = g_reval_mismatch; out[3] = g_far_seeded; }
extern "C" void emu_set_perturb(double e) { g_emu_perturb = e; }
// record-level re-validation of the settled passes (s3d_nn_settled_kernel): from which outer iteration on (-1: off)
static int g_emu_records_from = -1;
static long long g_rec_tested = 0, g_rec_skipped = 0, g_rec_queries_skipped = 0, g_rec_pass[64][2];
extern "C" void emu_record_pass_stats(long long* out) { std::memcpy(out, g_rec_pass, sizeof g_rec_pass); std::memset(g_rec_pass, 0, sizeof g_rec_pass); }
extern "C" void emu_set_records_from(int it) { g_emu_records_from = it; g_rec_tested = g_rec_skipped = g_rec_queries_skipped = 0; }
extern "C" void emu_record_stats(long long* out) { out[0] = g_rec_tested; out[1] = g_rec_skipped; out[2] = g_rec_queries_skipped; }
namespace {

struct Cloud {
  std::vector<F4> pts;  // filtered order, w = 1
};
struct Grid {
  GridParams g;
  std::vector<uint32_t> cell_start;
  std::vector<F4> sorted;  // w = index bits
};

void bbox(const std::vector<F4>& p, float mn[3], float mx[3]) {
  for (int a = 0; a < 3; ++a) { mn[a] = FLT_MAX; mx[a] = -FLT_MAX; }
  for (const F4& q : p) {
    const float v[3] = {q.x, q.y, q.z};
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); }
  }
}

Cloud voxel(const float* xyz, int n, int stride, double leaf) {
  Cloud out;
  std::vector<F4> raw;
  raw.reserve(n);
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * stride;
    if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) raw.push_back({p[0], p[1], p[2], 1.f});
  }
  if (leaf <= 0 || raw.empty()) { out.pts = raw; return out; }
  float mn[3], mx[3];
  bbox(raw, mn, mx);
  VoxelParams vp = voxel_params_from_bbox(mn, mx, (float)leaf);
  std::vector<std::pair<uint32_t, int>> kv(raw.size());
  for (size_t i = 0; i < raw.size(); ++i)
    kv[i] = {vp.passthrough ? (uint32_t)i : voxel_key(vp, raw[i].x, raw[i].y, raw[i].z), (int)i};
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  for (size_t i = 0; i < kv.size();) {
    size_t j = i;
    float sx = 0, sy = 0, sz = 0;
    while (j < kv.size() && kv[j].first == kv[i].first) {
      const F4& p = raw[kv[j].second];
      sx += p.x; sy += p.y; sz += p.z;
      ++j;
    }
    const float c = (float)(j - i);
    out.pts.push_back({sx / c, sy / c, sz / c, 1.f});
    i = j;
  }
  return out;
}

Grid build_grid(const Cloud& c, float h0, int cells_per_point) {
  Grid G;
  float mn[3], mx[3];
  bbox(c.pts, mn, mx);
  int cap = (int)std::min<int64_t>((int64_t)cells_per_point * (int64_t)c.pts.size(), 1 << 24);
  G.g = grid_params_from_bbox(mn, mx, h0, std::max(cap, 64));
  std::vector<std::pair<uint32_t, int>> kv(c.pts.size());
  for (size_t i = 0; i < c.pts.size(); ++i)
    kv[i] = {(uint32_t)grid_cell_of_point(G.g, c.pts[i].x, c.pts[i].y, c.pts[i].z), (int)i};
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  G.cell_start.assign(G.g.ncells + 1, 0);
  G.sorted.resize(c.pts.size());
  size_t k = 0;
  for (int cell = 0; cell <= G.g.ncells; ++cell) {
    while (k < kv.size() && kv[k].first < (uint32_t)cell) ++k;
    G.cell_start[cell] = (uint32_t)k;
  }
  for (size_t i = 0; i < kv.size(); ++i) {
    const F4& p = c.pts[kv[i].second];
    G.sorted[i] = {p.x, p.y, p.z, __builtin_bit_cast(float, kv[i].second)};
  }
  return G;
}

// The fused pre-pass (round 5, s3d_core.h "K2 + K3 in one sort") as the device runs it: bbox of the finite raw points,
// pcl::VoxelGrid's lattice, the search grid over that lattice, ONE stable sort of the raw points by (cell, voxel) key,
// centroids by float sums in index order, written in cell order with PCL's voxel key as the tie-breaking id, and the
// cell table.  ok = FusedGrid::ok as the device would report it (< 0: the host falls back to the two-sort path).
struct Fused {
  Grid G;              // sorted[].w = voxel key bits
  VoxelParams vp;
  FusedGrid fz;
  std::vector<int> cell;   // cell id of every centroid
};
Fused build_fused(const float* xyz, int n, int stride, double leaf, int cells_per_point) {
  Fused F;
  std::vector<F4> raw;
  raw.reserve(n);
  for (int i = 0; i < n; ++i) {
    const float* p = xyz + (size_t)i * stride;
    raw.push_back({p[0], p[1], p[2], 1.f});
  }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool any = false;
  for (const F4& q : raw) {
    if (!(std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z))) continue;
    any = true;
    const float v[3] = {q.x, q.y, q.z};
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], v[a]); mx[a] = std::max(mx[a], v[a]); }
  }
  if (!any) {
    F.vp.inv_leaf = 1.f; F.vp.passthrough = 0;
    for (int a = 0; a < 3; ++a) { F.vp.min_b[a] = 0; F.vp.div_b[a] = 1; }
  } else {
    F.vp = voxel_params_from_bbox(mn, mx, (float)leaf);
  }
  const int cap = (int)std::max<int64_t>(std::min<int64_t>((int64_t)cells_per_point * (int64_t)n, 1 << 24), 64);
  fused_grid_from_voxels(F.vp, cap, F.G.g, F.fz);
  std::vector<std::pair<uint32_t, int>> kv;
  for (size_t i = 0; i < raw.size(); ++i) {
    const F4& q = raw[i];
    if (F.fz.ok > 0 && std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z))
      kv.push_back({fused_key(F.vp, F.G.g, F.fz, q.x, q.y, q.z), (int)i});
  }
  std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
  F.G.cell_start.assign((size_t)F.G.g.ncells + 1, 0);
  int prev_cell = -1;
  for (size_t i = 0; i < kv.size();) {
    size_t j = i;
    float sx = 0, sy = 0, sz = 0;
    while (j < kv.size() && kv[j].first == kv[i].first) {
      const F4& p = raw[kv[j].second];
      sx += p.x; sy += p.y; sz += p.z;
      ++j;
    }
    const float c = (float)(j - i);
    const float qx = sx / c, qy = sy / c, qz = sz / c;
    // as k_centroids_fused: the cell by the multiply-high, the id from the run's first raw point, the check from the position;
    // the decode (the definition of both) must agree
    const int cell = (int)fused_cell_of_key(F.fz, kv[i].first);
    const F4& pf = raw[kv[i].second];
    const uint32_t voxel = fused_voxel_of_point(F.vp, pf.x, pf.y, pf.z);
    {
      int cell2, cc[3];
      uint32_t voxel2;
      fused_decode(F.vp, F.G.g, F.fz, kv[i].first, &cell2, cc, &voxel2);
      if (cell2 != cell || voxel2 != voxel) { std::fprintf(stderr, "emu: fused decode mismatch\n"); std::abort(); }
    }
    if (!fused_point_in_cell(F.vp, F.G.g, F.fz, kv[i].first, cell, qx, qy, qz)) F.fz.ok = -2;
    const uint32_t pos = (uint32_t)F.G.sorted.size();
    for (int ce = prev_cell + 1; ce <= cell; ++ce) F.G.cell_start[ce] = pos;     // the gap fill of k_centroids_fused
    prev_cell = cell;
    F.G.sorted.push_back({qx, qy, qz, __builtin_bit_cast(float, voxel)});
    F.cell.push_back(cell);
    i = j;
  }
  for (int ce = prev_cell + 1; ce <= F.G.g.ncells; ++ce) F.G.cell_start[ce] = (uint32_t)F.G.sorted.size();
  return F;
}

struct D3 { double x, y, z; };
std::vector<D3> normals(const Cloud& c, const Grid& G, int k) {
  std::vector<D3> out(c.pts.size());
  std::vector<float> d2(k);
  std::vector<int> idx(k);
  for (size_t i = 0; i < c.pts.size(); ++i) {
    const F4& q = c.pts[i];
    unsigned long long keys[32];
    int cnt = grid_knn_sorted<32>(G.g, G.cell_start.data(), G.sorted.data(), q.x, q.y, q.z, k, keys);
    for (int j = 0; j < cnt; ++j) idx[j] = (int)(uint32_t)(keys[j] & 0xFFFFFFFFull);
    Moments m;
    moments_init(m);
    for (int j = 0; j < cnt; ++j) { const F4& p = c.pts[idx[j]]; moments_add(m, p.x, p.y, p.z); }
    double n[3];
    moments_normal(m, k, n);
    double nd[3];
    normal_decode(normal_encode(n), nd);   // device storage: the 16-byte NormalRec (float xyz + 10-bit remainders)
    out[i] = {nd[0], nd[1], nd[2]};
  }
  return out;
}

float h0_for(double density) { return density > 0 ? (float)(2.0 * density) : 0.25f; }

}  // namespace

extern "C" {

struct emu_info {
  int n_source_filtered, n_target_filtered, iterations, converged, correspondences;
  double fitness;
  int inner_total, evals_total;
};

// stage exports for unit tests -------------------------------------------------
int emu_voxel(const float* xyz, int n, int stride, double leaf, float* out) {
  Cloud c = voxel(xyz, n, stride, leaf);
  for (size_t i = 0; i < c.pts.size(); ++i) { out[i * 3] = c.pts[i].x; out[i * 3 + 1] = c.pts[i].y; out[i * 3 + 2] = c.pts[i].z; }
  return (int)c.pts.size();
}
void emu_nn(const float* tgt, int n, const float* qry, int m, float h0, int cpp, float max_d, int* idx, float* d2) {
  Cloud c = voxel(tgt, n, 3, 0.0);
  Grid G = build_grid(c, h0, cpp);
  for (int i = 0; i < m; ++i) {
    NNResult r = grid_nn1(G.g, G.cell_start.data(), G.sorted.data(), qry[i * 3], qry[i * 3 + 1], qry[i * 3 + 2], max_d);
    idx[i] = r.idx; d2[i] = r.d2;
  }
}
// fast != 0: the first-pass variant (nn1_consider<FAST>: one packed key, no runner-up); pos_ok (may be null) counts the
// results whose position names the returned index
void emu_nn_box(const float* tgt, int n, const float* qry, int m, float h0, int cpp, float max_d, const float* hint,
                int* idx, float* d2, int fast, int* pos_ok) {
  Cloud c = voxel(tgt, n, 3, 0.0);
  Grid G = build_grid(c, h0, cpp);
  int ok = 0;
  for (int i = 0; i < m; ++i) {
    NNResult r = fast ? grid_nn1_box<true>(G.g, G.cell_start.data(), G.sorted.data(), qry[i * 3], qry[i * 3 + 1],
                                           qry[i * 3 + 2], max_d, hint[i])
                      : grid_nn1_box(G.g, G.cell_start.data(), G.sorted.data(), qry[i * 3], qry[i * 3 + 1],
                                     qry[i * 3 + 2], max_d, hint[i]);
    idx[i] = r.idx; d2[i] = r.d2;
    if (r.idx < 0 ? r.pos < 0 : (r.pos >= 0 && __builtin_bit_cast(int, G.sorted[r.pos].w) == r.idx)) ++ok;
  }
  if (pos_ok) *pos_ok = ok;
}
void emu_normals(const float* xyz, int n, int k, float h0, int cpp, float* out) {
  Cloud c = voxel(xyz, n, 3, 0.0);
  Grid G = build_grid(c, h0, cpp);
  std::vector<D3> nr = normals(c, G, k);
  for (int i = 0; i < n; ++i) { out[i * 3] = (float)nr[i].x; out[i * 3 + 1] = (float)nr[i].y; out[i * 3 + 2] = (float)nr[i].z; }
}

// round-3 1-NN by a flat scan of the 27 cells (grid_nn1_scan27): idx / d2 of the answered queries (flag 1), -1 / 0 flag 0
// seed (may be null): per query the index of a target point whose distance is handed in as the known upper bound
void emu_nn_scan27(const float* tgt, int n, const float* qry, int m, float h0, int cpp, int* idx, float* d2, int* answered,
                   float* lower_bound, const int* seed, int prescan) {
  Cloud c = voxel(tgt, n, 3, 0.0);
  Grid G = build_grid(c, h0, cpp);
  for (int i = 0; i < m; ++i) {
    uint32_t tab[kKnn3Segs];
    NNResult r;
    float sd2 = 3.0e38f;
    if (seed && seed[i] >= 0) sd2 = dist2(qry[i * 3], qry[i * 3 + 1], qry[i * 3 + 2], tgt[seed[i] * 3], tgt[seed[i] * 3 + 1], tgt[seed[i] * 3 + 2]);
    const bool ok = prescan ? grid_nn1_scan27<1>(G.g, G.cell_start.data(), G.sorted.data(), qry[i * 3], qry[i * 3 + 1], qry[i * 3 + 2], tab, 1, r, sd2)
                            : grid_nn1_scan27(G.g, G.cell_start.data(), G.sorted.data(), qry[i * 3], qry[i * 3 + 1], qry[i * 3 + 2], tab, 1, r, sd2);
    answered[i] = ok ? 1 : 0;
    idx[i] = ok ? r.idx : -1; d2[i] = ok ? r.d2 : 0.f;
    lower_bound[i] = ok ? nn_lower_bound_others(r) : 0.f;
  }
}

// round-3 k-NN (grid_knn_med3: 32-bit keys, med3 insertion) against the exact 64-bit search on every point of a cloud.
// out[0] = points, out[1] = points the fast path declines (served by the exact search on the device), out[2] = answered
// points whose 20-neighbour SET differs from the exact one (must be 0), out[3] = answered points whose order differs
void emu_knn3_check(const float* xyz, int n, float h0, int cpp, long long* out) {
  Cloud c = voxel(xyz, n, 3, 0.0);
  Grid G = build_grid(c, h0, cpp);
  out[0] = n; out[1] = out[2] = out[3] = 0;
  for (int i = 0; i < n; ++i) {
    const F4& q = G.sorted[i];
    uint32_t tab[kKnn3Segs], keys[21];
    if (!grid_knn_med3<21>(G.g, G.cell_start.data(), G.sorted.data(), q.x, q.y, q.z, tab, 1, keys)) { ++out[1]; continue; }
    unsigned long long ref[20];
    const int cnt = grid_knn_sorted<20, true>(G.g, G.cell_start.data(), G.sorted.data(), q.x, q.y, q.z, 20, ref);
    std::vector<uint32_t> a, b;
    bool same_order = cnt == 20;
    for (int j = 0; j < 20; ++j) {
      const uint32_t idx = __builtin_bit_cast(uint32_t, G.sorted[knn3_position(keys[j], tab, 1)].w);
      a.push_back(idx);
      if (j < cnt) { b.push_back((uint32_t)(ref[j] & 0xFFFFFFFFull)); same_order = same_order && idx == b.back(); }
    }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    if (a != b) ++out[2];
    if (!same_order) ++out[3];
  }
}

// ---- the fused pre-pass (round 5) ----------------------------------------------------------------------------------
// centroids in cell order with their voxel keys and cells; info = {ok, m, dim0, dim1, dim2, ncells, msub, structural
// errors (cell table not monotone / a point not in the cell range the table gives / cells or voxel keys not ascending)}
int emu_fused_voxel(const float* xyz, int n, int stride, double leaf, int cpp, float* out_xyz, unsigned* out_voxel,
                    int* out_cell, long long* info, float* grid) {
  Fused F = build_fused(xyz, n, stride, leaf, cpp);
  long long bad = 0;
  const size_t m = F.G.sorted.size();
  for (size_t i = 0; i < m; ++i) {
    out_xyz[i * 3] = F.G.sorted[i].x; out_xyz[i * 3 + 1] = F.G.sorted[i].y; out_xyz[i * 3 + 2] = F.G.sorted[i].z;
    out_voxel[i] = __builtin_bit_cast(uint32_t, F.G.sorted[i].w);
    out_cell[i] = F.cell[i];
    const uint32_t a = F.G.cell_start[F.cell[i]], b = F.G.cell_start[F.cell[i] + 1];
    if (!(a <= i && i < b)) ++bad;
    if (i > 0 && (F.cell[i] < F.cell[i - 1] || (F.cell[i] == F.cell[i - 1] && out_voxel[i] <= out_voxel[i - 1]))) ++bad;
  }
  for (int c = 0; c < F.G.g.ncells; ++c) if (F.G.cell_start[c] > F.G.cell_start[c + 1]) ++bad;
  if (F.G.cell_start[0] != 0 || F.G.cell_start[F.G.g.ncells] != m) ++bad;
  info[0] = F.fz.ok; info[1] = F.fz.m; info[2] = F.G.g.dim[0]; info[3] = F.G.g.dim[1]; info[4] = F.G.g.dim[2];
  info[5] = F.G.g.ncells; info[6] = F.fz.msub; info[7] = bad;
  if (grid) { grid[0] = F.G.g.origin[0]; grid[1] = F.G.g.origin[1]; grid[2] = F.G.g.origin[2]; grid[3] = F.G.g.h; }
  return (int)m;
}

// every search of the registration on a fused grid: queries qry (m x 3) against the filtered cloud of `src`.
// mode 0: grid_nn1_box with the given hints, 1: its first-pass (FAST) form, 2: grid_nn1_scan27, 3: scan27 PRESCAN.
// Outputs per query: voxel key of the neighbour (0xFFFFFFFF: none / declined), float d2, answered flag.
void emu_fused_nn(const float* src, int n, int stride, double leaf, int cpp, const float* qry, int m, float max_d,
                  const float* hint, int mode, unsigned* nn_voxel, float* d2, int* answered) {
  Fused F = build_fused(src, n, stride, leaf, cpp);
  const GridParams& g = F.G.g;
  const uint32_t* cs = F.G.cell_start.data();
  const F4* pts = F.G.sorted.data();
  for (int i = 0; i < m; ++i) {
    const float x = qry[i * 3], y = qry[i * 3 + 1], z = qry[i * 3 + 2];
    NNResult r;
    bool ok = true;
    if (mode == 0) r = grid_nn1_box(g, cs, pts, x, y, z, max_d, hint[i]);
    else if (mode == 1) r = grid_nn1_box<true>(g, cs, pts, x, y, z, max_d, hint[i]);
    else {
      uint32_t tab[kKnn3Segs];
      ok = mode == 3 ? grid_nn1_scan27<1>(g, cs, pts, x, y, z, tab, 1, r, 3.0e38f)
                     : grid_nn1_scan27(g, cs, pts, x, y, z, tab, 1, r, 3.0e38f);
    }
    answered[i] = ok ? 1 : 0;
    const bool have = ok && r.pos >= 0;
    nn_voxel[i] = have ? __builtin_bit_cast(uint32_t, pts[r.pos].w) : 0xFFFFFFFFu;
    d2[i] = have ? r.d2 : 3.0e38f;
    if (have && r.idx != __builtin_bit_cast(int, pts[r.pos].w)) answered[i] = -1;   // (position and id must name one point)
  }
}

// k-NN on a fused grid: the med3 pre-pass against the exact search BY POSITION (grid_knn_sorted<.., BYPOS>), and the
// exact search by position against a brute-force scan.  out[0] = points, [1] = declined by med3, [2] = answered with a
// different 20-neighbour set, [3] = exact searches (sampled) whose set differs from brute force
void emu_fused_knn_check(const float* xyz, int n, int stride, double leaf, int cpp, long long* out) {
  Fused F = build_fused(xyz, n, stride, leaf, cpp);
  const GridParams& g = F.G.g;
  const int np_ = (int)F.G.sorted.size();
  out[0] = np_; out[1] = out[2] = out[3] = 0;
  for (int i = 0; i < np_; ++i) {
    const F4& q = F.G.sorted[i];
    unsigned long long ref[20];
    const int cnt = grid_knn_sorted<20, true, true>(g, F.G.cell_start.data(), F.G.sorted.data(), q.x, q.y, q.z, 20, ref);
    std::vector<uint32_t> b;
    for (int j = 0; j < cnt; ++j) b.push_back((uint32_t)(ref[j] & 0xFFFFFFFFull));
    std::sort(b.begin(), b.end());
    if (i % 37 == 0) {   // brute force: the 20 smallest (d2, position)
      std::vector<std::pair<float, uint32_t>> all;
      for (int j = 0; j < np_; ++j) all.push_back({dist2(q.x, q.y, q.z, F.G.sorted[j].x, F.G.sorted[j].y, F.G.sorted[j].z), (uint32_t)j});
      std::partial_sort(all.begin(), all.begin() + std::min(20, np_), all.end());
      std::vector<uint32_t> c;
      for (int j = 0; j < std::min(20, np_); ++j) c.push_back(all[j].second);
      std::sort(c.begin(), c.end());
      if (c != b) ++out[3];
    }
    uint32_t tab[kKnn3Segs], keys[21];
    if (!grid_knn_med3<21>(g, F.G.cell_start.data(), F.G.sorted.data(), q.x, q.y, q.z, tab, 1, keys)) { ++out[1]; continue; }
    std::vector<uint32_t> a;
    for (int j = 0; j < 20; ++j) a.push_back(knn3_position(keys[j], tab, 1));
    std::sort(a.begin(), a.end());
    if (a != b) ++out[2];
  }
}

// the ring-by-ring med3 search (round 6: grid_knn_med3_rings, what serves the sparse parts of a scan) on a fused grid, for
// EVERY point or only those the fast path declines (only_declined), against the exact search by position.
// out[0] = points tried, [1] = not answered, [2] = answered with a different K-NN set, [3] = declined by the fast path
}  // extern "C"
template <int SB>
static void fused_knn_rings_check(const float* xyz, int n, int stride, double leaf, int cpp, int rmax, int only_declined,
                                  long long* out) {
  Fused F = build_fused(xyz, n, stride, leaf, cpp);
  const GridParams& g = F.G.g;
  const int np_ = (int)F.G.sorted.size();
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int i = 0; i < np_; ++i) {
    const F4& q = F.G.sorted[i];
    uint32_t tab[1 << SB], keys[21];
    {
      uint32_t t16[kKnn3Segs], k16[21];
      const bool fast = grid_knn_med3<21>(g, F.G.cell_start.data(), F.G.sorted.data(), q.x, q.y, q.z, t16, 1, k16);
      if (!fast) ++out[3];
      if (fast && only_declined) continue;
    }
    ++out[0];
    if (grid_knn_med3_rings<21, SB>(g, F.G.cell_start.data(), F.G.sorted.data(), q.x, q.y, q.z, tab, 1, keys, rmax) != 0) { ++out[1]; continue; }
    unsigned long long ref[20];
    const int cnt = grid_knn_sorted<20, true, true>(g, F.G.cell_start.data(), F.G.sorted.data(), q.x, q.y, q.z, 20, ref);
    std::vector<uint32_t> a, b;
    for (int j = 0; j < cnt; ++j) b.push_back((uint32_t)(ref[j] & 0xFFFFFFFFull));
    for (int j = 0; j < 20; ++j) a.push_back(knn3_position<SB>(keys[j], tab, 1));
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    if (a != b) ++out[2];
  }
}
extern "C" {
// rmax + 100 * segbits (segbits 0: the device's table size)
void emu_fused_knn_rings_check(const float* xyz, int n, int stride, double leaf, int cpp, int rmax, int only_declined,
                               long long* out) {
  const int sb = rmax / 100;
  rmax %= 100;
  if (sb == 6) fused_knn_rings_check<6>(xyz, n, stride, leaf, cpp, rmax, only_declined, out);
  else fused_knn_rings_check<kKnn3RingSegBits>(xyz, n, stride, leaf, cpp, rmax, only_declined, out);
}

// B1/B2 (patch accumulation, radius outlier removal) as the kernels run them ---------------------
void emu_transform(const float* xyz, int n, const double* tf_colmajor, float* out) {
  double T[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) T[r * 4 + c] = tf_colmajor[c * 4 + r];
  for (int i = 0; i < n; ++i) {
    F3 q = xf_pcl_d(T, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
    out[i * 3] = q.x; out[i * 3 + 1] = q.y; out[i * 3 + 2] = q.z;
  }
}
int emu_remove_outliers(const float* xyz, int n, double radius, unsigned min_neighbors, int cpp, float* out) {
  Cloud c = voxel(xyz, n, 3, 0.0);
  Grid G = build_grid(c, (float)radius, cpp);
  const double r2 = radius * radius;
  float r2f = (float)r2;
  if ((double)r2f > r2) r2f = std::nextafterf(r2f, 0.f);
  const float reach = (float)(radius * 1.00001) + 1e-30f;
  const int need = (int)min_neighbors + 1;
  std::vector<char> keep(c.pts.size(), 0);
  for (size_t i = 0; i < G.sorted.size(); ++i) {
    const F4& q = G.sorted[i];
    const int cnt = grid_radius_count(G.g, G.cell_start.data(), G.sorted.data(), q.x, q.y, q.z, reach, r2f, need);
    keep[__builtin_bit_cast(int, q.w)] = cnt >= need;
  }
  int m = 0;
  for (size_t i = 0; i < c.pts.size(); ++i)
    if (keep[i]) { out[m * 3] = c.pts[i].x; out[m * 3 + 1] = c.pts[i].y; out[m * 3 + 2] = c.pts[i].z; ++m; }
  return m;
}

// align() as the device pipeline runs it -------------------------------------------
int emu_align(const float* source, int n_source, int stride_source, const float* target, int n_target,
              int stride_target, const double guess_d[16], const s3d_reg_params* cfg, int force_iterations,
              int cells_per_point, double result[16], emu_info* info) {
  emu_info li;
  std::memset(&li, 0, sizeof li);
  for (int i = 0; i < 16; ++i) result[i] = (i % 5 == 0) ? 1.0 : 0.0;
  const int alg = cfg->registration_algorithm;
  if (alg == S3D_ALG_NDT || alg == S3D_ALG_NDT_OMP) return S3D_STATUS_UNSUPPORTED_ALGORITHM;
  if (alg < 0 || alg > 4) return S3D_STATUS_UNKNOWN_ALGORITHM;
  Cloud S = voxel(source, n_source, stride_source, cfg->point_cloud_density);  // pcl TARGET (kd-tree side)
  Cloud T = voxel(target, n_target, stride_target, cfg->point_cloud_density);  // pcl SOURCE (queries)
  li.n_source_filtered = (int)S.pts.size();
  li.n_target_filtered = (int)T.pts.size();
  if (S.pts.size() < 100 || T.pts.size() < 100) { if (info) *info = li; return S3D_STATUS_TOO_FEW_POINTS; }
  const int k = cfg->correspondence_randomness;
  if (k > (int)S.pts.size() || k > (int)T.pts.size()) return S3D_STATUS_INVALID_ARGUMENT;
  const float h0 = h0_for(cfg->point_cloud_density);
  Grid GS = build_grid(S, h0, cells_per_point);
  std::vector<D3> NS = normals(S, GS, k), NT;
  const bool gicp = (alg == S3D_ALG_GICP || alg == S3D_ALG_GICP_OMP);
  if (gicp) { Grid GT = build_grid(T, h0, cells_per_point); NT = normals(T, GT, k); }

  Mat4f guess;
  for (int i = 0; i < 16; ++i) guess.m[i] = (float)guess_d[i];
  Mat4f Tr = mat4f_identity(), prev = mat4f_identity();
  const double thr = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
  const float max_d_search = (float)(cfg->max_correspondence_distance * 1.0001);   // what Batch::stage_icp passes to K5
  int nr = 0, converged = 0, cnt = 0;
  std::vector<float> hints(T.pts.size(), -1.f), lbs(T.pts.size(), 0.f);
  std::vector<int> seeds(T.pts.size(), -1);
  Mat4f T_nn = mat4f_identity();
  // 64-query records of the settled passes (s3d_nn_settled_kernel): box, margin, touch pass; transformation_ per pass
  const size_t nrec = (T.pts.size() + 63) / 64;
  std::vector<WaveRec> recs(nrec);
  for (WaveRec& w : recs) { w.touch = -1; w.margin = -1.f; }
  std::vector<Mat4f> T_hist;
  std::vector<char> rec_pass(nrec, 0);
  while (!converged) {
    double R[9], SS[6], Th0[12];
    gicp_rotation(Tr, guess, R, SS);
    for (int c = 0; c < 3; ++c)
      for (int a = 0; a < 4; ++a) Th0[c * 4 + a] = (double)S3D_M(Tr, c, a);
    double acc[GQ_NACC] = {0};
    const bool records_on = g_emu_records_from >= 0 && nr >= g_emu_records_from;
    for (size_t i = 0; i < T.pts.size(); ++i) {
      const F4& p0 = T.pts[i];
      const F3 p = xf_pcl(guess, p0.x, p0.y, p0.z);
      const F3 q = xf_eigen(Tr, p.x, p.y, p.z);
      WaveRec& W = recs[i / 64];
      bool rec_skip = false;
      if (records_on) {
        if (i % 64 == 0) {   // the record-level proof, once per record
          ++g_rec_tested;
          rec_pass[i / 64] = W.touch >= 0 && W.margin > 0.f &&
                             nn_record_move_bound(Tr, T_hist[(size_t)W.touch], W.c, W.e) < (double)W.margin;
          if (getenv("EMU_REC_DEBUG") && nr == atoi(getenv("EMU_REC_DEBUG")))
            fprintf(stderr, "rec %zu touch %d margin %.3e bound %.3e e %.2f %.2f %.2f\n", i / 64, W.touch, W.margin,
                    W.touch >= 0 ? nn_record_move_bound(Tr, T_hist[(size_t)W.touch], W.c, W.e) : -1.0, W.e[0], W.e[1], W.e[2]);
          if (rec_pass[i / 64]) ++g_rec_skipped;
          if (nr < 64) { ++g_rec_pass[nr][0]; g_rec_pass[nr][1] += rec_pass[i / 64] ? 1 : 0; }
        }
        rec_skip = rec_pass[i / 64] != 0;
      }
      // the same decisions as s3d_nn_search_kernel<0> (hints[i]: previous d2, 3e38 = searched but none, -1 = never)
      const float prevd = hints[i];
      const bool has_prev = prevd >= 0.f && prevd < 1.0e30f;
      NNResult r;
      r.idx = -1; r.d2 = 3.0e38f; r.pos = -1;
      bool revalidated = false;
      float move = 3.0e38f;
      if (rec_skip) {
        // nothing of this query is read by the kernel: the stored correspondence stands.  Check it against a full search.
        ++g_rec_queries_skipped;
        if (has_prev) {
          const F4& ps = GS.sorted[seeds[i]];
          r.idx = __builtin_bit_cast(int, ps.w); r.d2 = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z); r.pos = seeds[i];
        }
        NNResult full = grid_nn1_box(GS.g, GS.cell_start.data(), GS.sorted.data(), q.x, q.y, q.z, max_d_search, GS.g.h);
        const bool full_in = full.idx >= 0 && full.d2 <= max_d_search * max_d_search;
        const bool r_in = r.idx >= 0 && r.d2 <= max_d_search * max_d_search;
        if (full_in != r_in || (full_in && (full.idx != r.idx || full.d2 != r.d2))) ++g_reval_mismatch;
        revalidated = true;
      }
      const Mat4f& Tref = (records_on && W.touch >= 0) ? T_hist[(size_t)W.touch] : T_nn;
      if (!rec_skip && lbs[i] > 0.f && prevd >= 0.f) {   // the kernel's triangle-inequality shortcut
        const F3 qo = xf_eigen(Tref, p.x, p.y, p.z);
        move = std::sqrt(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
        if (has_prev) {
          const F4& ps = GS.sorted[seeds[i]];
          const float d2n = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
          if (nn_still_nearest(std::sqrt(d2n), move, lbs[i])) {
            r.idx = __builtin_bit_cast(int, ps.w); r.d2 = d2n; r.pos = seeds[i];
            revalidated = true;
          }
        } else if (nn_still_nearest(max_d_search, move, lbs[i])) {
          revalidated = true;                 // still no point within max_d
        }
        if (revalidated) {
          lbs[i] -= move;
          ++g_reval_hits;
          // the shortcut must agree with a full, unseeded search, bit for bit (within max_d)
          NNResult full = grid_nn1_box(GS.g, GS.cell_start.data(), GS.sorted.data(), q.x, q.y, q.z, max_d_search, GS.g.h);
          const bool full_in = full.idx >= 0 && full.d2 <= max_d_search * max_d_search;
          const bool r_in = r.idx >= 0 && r.d2 <= max_d_search * max_d_search;
          if (full_in != r_in || (full_in && (full.idx != r.idx || full.d2 != r.d2))) ++g_reval_mismatch;
        }
      }
      if (!revalidated) {
        const bool near_seed = has_prev && prevd < GS.g.h * GS.g.h;
        const bool far_seed = has_prev && !near_seed && move < kNNRevalSlack * GS.g.h;
        const int seed = (near_seed || far_seed) ? seeds[i] : -1;
        const float hint = has_prev ? std::fmin(std::sqrt(prevd) * 1.25f + 0.05f * GS.g.h, GS.g.h) : 3.0f * GS.g.h;
        r = grid_nn1_box(GS.g, GS.cell_start.data(), GS.sorted.data(), q.x, q.y, q.z, max_d_search, hint, seed, far_seed);
        lbs[i] = nn_lower_bound_others(r);
        ++g_reval_misses;
        if (far_seed) {   // a trusted far seed must give what an unseeded search gives
          ++g_far_seeded;
          NNResult full = grid_nn1_box(GS.g, GS.cell_start.data(), GS.sorted.data(), q.x, q.y, q.z, max_d_search, GS.g.h);
          const bool full_in = full.idx >= 0 && full.d2 <= max_d_search * max_d_search;
          const bool r_in = r.idx >= 0 && r.d2 <= max_d_search * max_d_search;
          if (full_in != r_in || (full_in && (full.idx != r.idx || full.d2 != r.d2))) ++g_reval_mismatch;
        }
      }
      if (!rec_skip) {
        hints[i] = r.d2;
        seeds[i] = r.pos;
      }
      if (records_on && !rec_skip) {
        // the record is being evaluated in full: its queries' margins for the next record-level proof, its box at
        // the first evaluation, the touch pass when its last query is through
        const bool has = r.idx >= 0 && r.d2 < 1.0e30f;
        const float m = nn_margin(has, lbs[i], has ? std::sqrt(r.d2) : 0.f, max_d_search);
        const size_t first = (i / 64) * 64, last = std::min(first + 64, T.pts.size()) - 1;
        static float run_min, mn[3], mx[3];
        if (i == first) { run_min = 3.0e38f; for (int a = 0; a < 3; ++a) { mn[a] = 3.0e38f; mx[a] = -3.0e38f; } }
        // (a record with a searched query is evaluated again in the next pass: its searches run in a kernel of their own,
        // after the record has been written - s3d_nn_record_touch_kernel)
        run_min = std::fmin(run_min, revalidated ? m : -1.f);
        const float pv[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; ++a) { mn[a] = std::fmin(mn[a], pv[a]); mx[a] = std::fmax(mx[a], pv[a]); }
        if (i == last) {
          for (int a = 0; a < 3; ++a) {
            W.c[a] = 0.5f * mn[a] + 0.5f * mx[a];
            W.e[a] = std::fmax(mx[a] - W.c[a], W.c[a] - mn[a]) * 1.000001f + 1.0e-30f;
          }
          W.margin = run_min;
          W.touch = nr;
        }
      }
      if (r.idx < 0 || !((double)r.d2 < thr)) continue;
      const F4& t = S.pts[r.idx];
      const double td[3] = {t.x, t.y, t.z};
      const double n2[3] = {NS[r.idx].x, NS[r.idx].y, NS[r.idx].z};
      if (gicp) {
        const double n1[3] = {NT[i].x, NT[i].y, NT[i].z};
        double n1r[3], M[6];
        for (int a = 0; a < 3; ++a) n1r[a] = R[a * 3] * n1[0] + R[a * 3 + 1] * n1[1] + R[a * 3 + 2] * n1[2];
        gicp_mahalanobis(SS, n1r, n2, 0.001, M);
        if (g_emu_perturb != 0.0) for (int a = 0; a < 6; ++a) M[a] *= 1.0 + g_emu_perturb * ((double)rand() / RAND_MAX - 0.5);
        const double pd[3] = {p.x, p.y, p.z};
        gq_accumulate(acc, pd, td, M, Th0);
      } else {
        const double qd[3] = {q.x, q.y, q.z};
        // (the point-to-plane kernel reads the float part of the record)
        const double nf[3] = {(double)(float)NS[r.idx].x, (double)(float)NS[r.idx].y, (double)(float)NS[r.idx].z};
        pp_accumulate(acc, qd, td, nf);
      }
    }
    prev = Tr;
    T_nn = Tr;
    T_hist.push_back(Tr);   // (pass index nr: what the controller stores)
    int rc;
    if (gicp) {
      cnt = (int)acc[GQ_CNT];
      int inner = 0, evals = 0;
      rc = gicp_estimate_bfgs(acc, cfg->maximum_optimizer_iterations, Tr, &inner, &evals);
      li.inner_total += inner; li.evals_total += evals;
      if (getenv("S3O_DEBUG")) fprintf(stderr, "[emu]    it %d cnt %d inner %d evals %d t=(%.9g %.9g %.9g) r21 %.9g r10 %.9g\n", nr, cnt, inner, evals, S3D_M(Tr,0,3), S3D_M(Tr,1,3), S3D_M(Tr,2,3), S3D_M(Tr,2,1), S3D_M(Tr,1,0));
    } else {
      cnt = (int)acc[PP_CNT];
      rc = pp_update(acc, Tr);
    }
    if (rc) break;
    const double delta = icp_delta(prev, Tr, cfg->rotation_epsilon, cfg->transformation_epsilon);
    nr++;
    if (nr >= cfg->maximum_iterations || (!force_iterations && delta < 1)) { converged = 1; prev = Tr; }
  }
  const Mat4f fin = mat4f_mul(prev, guess);
  // fitness (PointCloudSensor.cpp:73): squared distance compared against the un-squared range
  const double max_range = cfg->max_correspondence_distance;
  const float fit_d = (float)(std::sqrt(max_range) * 1.0001);
  double fsum = 0; int fnr = 0;
  for (size_t i = 0; i < T.pts.size(); ++i) {
    const F3 q = xf_pcl(fin, T.pts[i].x, T.pts[i].y, T.pts[i].z);
    NNResult r = grid_nn1(GS.g, GS.cell_start.data(), GS.sorted.data(), q.x, q.y, q.z, fit_d);
    if (r.idx >= 0 && (double)r.d2 <= max_range) { fsum += r.d2; fnr++; }
  }
  li.fitness = fnr > 0 ? fsum / fnr : DBL_MAX;
  li.iterations = nr; li.converged = converged; li.correspondences = cnt;
  for (int i = 0; i < 16; ++i) result[i] = (double)fin.m[i];
  result[3] = result[7] = result[11] = 0.0; result[15] = 1.0;
  if (info) *info = li;
  if (!converged) return S3D_STATUS_NOT_CONVERGED;
  if (li.fitness > cfg->max_fitness_score) return S3D_STATUS_FITNESS_EXCEEDED;
  return S3D_STATUS_OK;  // (delta-from-guess gate is host code shared with the product; tested there)
}
}

// ---- unit hooks for the quadratic-form algebra (tests/test_core_math.py)
extern "C" {
void emu_gq_build(const double* p, const double* q, const double* M6, int m, const double* th0, double* acc /*GQ_NACC*/) {
  for (int i = 0; i < GQ_NACC; ++i) acc[i] = 0;
  for (int i = 0; i < m; ++i) gq_accumulate(acc, p + 3 * i, q + 3 * i, M6 + 6 * i, th0);
}
void emu_gq_eval(const double* acc, const double* th0, const double* x, double* f, double* g) { gq_eval(acc, th0, x, f, g); }
void emu_apply_state(const double* x, float* T16) { Mat4f T; gicp_apply_state(x, T); for (int i = 0; i < 16; ++i) T16[i] = T.m[i]; }
int emu_bfgs(const double* acc, int max_inner, float* T16, int* inner, int* evals) {
  Mat4f T; for (int i = 0; i < 16; ++i) T.m[i] = T16[i];
  int rc = gicp_estimate_bfgs(acc, max_inner, T, inner, evals);
  for (int i = 0; i < 16; ++i) T16[i] = T.m[i];
  return rc;
}
// the largest |fl(T p) - fl(Tt p)| over m points of the box (c, e) against nn_record_move_bound (must not exceed it)
double emu_record_move_bound(const float* T16, const float* Tt16, const float* c, const float* e, const float* pts, int m,
                             double* worst_ratio) {
  Mat4f T, Tt;
  for (int i = 0; i < 16; ++i) { T.m[i] = T16[i]; Tt.m[i] = Tt16[i]; }
  const double bound = nn_record_move_bound(T, Tt, c, e);
  double worst = 0.0;
  for (int i = 0; i < m; ++i) {
    const F3 a = xf_eigen(T, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    const F3 b = xf_eigen(Tt, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    const double dx = (double)a.x - b.x, dy = (double)a.y - b.y, dz = (double)a.z - b.z;
    worst = std::fmax(worst, std::sqrt(dx * dx + dy * dy + dz * dz));
  }
  *worst_ratio = bound > 0 ? worst / bound : 0.0;
  return bound;
}
void emu_mahalanobis(const double* S6, const double* n1r, const double* n2, double eps, double* M6) { gicp_mahalanobis(S6, n1r, n2, eps, M6); }
void emu_normal_roundtrip(const double* n, double* out, float* fpart) {
  const NormalRec r = normal_encode(n);
  normal_decode(r, out);
  fpart[0] = r.x; fpart[1] = r.y; fpart[2] = r.z;
}
}
