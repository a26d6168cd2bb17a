// s3d_api.hip — host orchestration + C ABI (include/slam3d_hip.h) of the MI355X
// registration back-end.  One translation unit: kernels in s3d_kernels.h.
//
// A call never routes to a CPU implementation: without a usable HIP device every
// entry point returns S3D_STATUS_BACKEND_ERROR.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <exception>
#include <map>
#include <random>
#include <set>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/slam3d_hip.h"
#include "../../include/slam3d_hip_debug.h"
#include "s3d_kernels.h"
#include "s3d_ndt.h"

using namespace s3d;

// ------------------------------------------------------------------ host structs

// one device allocation shared by the clouds of a bulk hand-over (s3d_cloud_upload_many): freed with the last of them
struct S3dDevBlock {
  void* p = nullptr;
  ~S3dDevBlock() { if (p) (void)hipFree(p); }
};
struct s3d_cloud {
  float4* d = nullptr;
  int n = 0;
  bool owned = false;
  unsigned long long uid = 0;   // never reused: the key of the pre-pass cache (an address could be recycled)
  std::shared_ptr<S3dDevBlock> block;   // set: d points into this block (owned = false)
};

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct HipError {
  hipError_t e;
  const char* what;
  int line;
  int status = S3D_STATUS_BACKEND_ERROR;   // what the entry point returns (a bad option found late: INVALID_ARGUMENT)
};
#define HIPCHK(expr)                                            \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) throw HipError{_e, #expr, __LINE__};  \
  } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline void cpu_relax() {   // the body of a host spin loop
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

// ---- host-side 4x4 double algebra for the acceptance gate (PointCloudSensor.cpp:167-172)
#define HM(m, r, c) ((m)[(c) * 4 + (r)])
void mat4d_mul(const double a[16], const double b[16], double out[16]) {
  double t[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += HM(a, r, k) * HM(b, k, c);
      t[c * 4 + r] = s;
    }
  std::memcpy(out, t, sizeof t);
}
void mat4d_inverse_isometry(const double a[16], double out[16]) {  // Eigen Isometry inverse: R^T, -R^T t
  double t[16] = {0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) HM(t, r, c) = HM(a, c, r);
  for (int r = 0; r < 3; ++r) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += HM(t, r, k) * HM(a, k, 3);
    HM(t, r, 3) = -s;
  }
  HM(t, 3, 3) = 1.0;
  std::memcpy(out, t, sizeof t);
}
double rotation_angle(const double m[16]) {  // Eigen::AngleAxisd(R).angle() via the quaternion
  const double m00 = HM(m, 0, 0), m11 = HM(m, 1, 1), m22 = HM(m, 2, 2);
  double t = m00 + m11 + m22, w, x, y, z;
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    w = 0.5 * t;
    t = 0.5 / t;
    x = (HM(m, 2, 1) - HM(m, 1, 2)) * t;
    y = (HM(m, 0, 2) - HM(m, 2, 0)) * t;
    z = (HM(m, 1, 0) - HM(m, 0, 1)) * t;
  } else {
    int i = 0;
    if (m11 > m00) i = 1;
    if (m22 > HM(m, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double q[3];
    t = std::sqrt(HM(m, i, i) - HM(m, j, j) - HM(m, k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    w = (HM(m, k, j) - HM(m, j, k)) * t;
    q[j] = (HM(m, j, i) + HM(m, i, j)) * t;
    q[k] = (HM(m, k, i) + HM(m, i, k)) * t;
    x = q[0]; y = q[1]; z = q[2];
  }
  return 2.0 * std::atan2(std::sqrt(x * x + y * y + z * z), std::fabs(w));
}

}  // namespace

namespace {
std::atomic<unsigned long long> g_cloud_uid{1};

// One cloud's pre-pass products (cross-call cache, s3d_exec_options.cache_prepass): the voxel-filtered points, the
// cell-sorted copies, the cell table, the normals (if a GICP / point-to-plane call computed them) and the
// device-computed part of its SlotDev.  ONE device allocation per entry.
struct CacheKey {
  unsigned long long uid; uint32_t leaf_bits, h0_bits; int cell_cap;
  int layout;       // 0: two sorts (filt in voxel order + cell-sorted copies); 1: the fused pre-pass (no filt, the grid on
                    // the voxel lattice, voxel keys as ids) - a registration and an NDT call on one cloud keep separate entries
  bool operator<(const CacheKey& o) const {
    return std::tie(uid, leaf_bits, h0_bits, cell_cap, layout) < std::tie(o.uid, o.leaf_bits, o.h0_bits, o.cell_cap, o.layout);
  }
};
struct CacheEntry {
  char* block = nullptr;
  size_t bytes = 0;
  size_t o_filt = 0, o_sorted = 0, o_sorted3 = 0, o_normals = 0, o_cells = 0, o_slot = 0;
  int cap_pts = 0, cap_cells = 0;
  int k_normals = 0;              // correspondence_randomness the cached normals were computed with (0: none)
  bool has_sorted3 = false;
  unsigned long long last_use = 0;
  s3d::SlotDev snap;              // host copy of the slot record; its device-computed part (n, bbox, vp, g) is reused
};
}  // namespace

struct s3d_context {
  std::map<CacheKey, CacheEntry> cache;
  size_t cache_bytes = 0, cache_limit = (size_t)16 << 30;   // (context_create lowers it to a quarter of the free HBM if that is less)
  unsigned long long cache_clock = 0;
  long long cache_hits = 0, cache_misses = 0;
  long long fused_reruns = 0;   // batches the fused pre-pass could not serve and ran again on the two-sort path
  // (cloud uid, voxel size as float bits) that made a batch run again: such batches start on the two-sort path
  std::set<std::pair<unsigned long long, uint32_t>> fused_unservable;
  void cache_drop(std::map<CacheKey, CacheEntry>::iterator it) {
    if (it->second.block) (void)hipFree(it->second.block);
    cache_bytes -= it->second.bytes;
    cache.erase(it);
  }
  void cache_clear() { while (!cache.empty()) cache_drop(cache.begin()); }
  void cache_forget_cloud(unsigned long long uid) {
    for (auto it = cache.lower_bound(CacheKey{uid, 0, 0, 0}); it != cache.end() && it->first.uid == uid;) cache_drop(it++);
    fused_unservable.erase(fused_unservable.lower_bound({uid, 0u}), fused_unservable.upper_bound({uid, 0xFFFFFFFFu}));
  }
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  std::mutex mtx;
  s3d_profile prof{};
  s3d_map_profile map_prof{};
  // workspace (grown on demand, reused across calls)
  DevBuf slots, pairs, keysA, keysB, valsA, valsB, filt, sorted, sorted3, normals, moments, cell_start, counts, digit_tot, blockcnt, blockbb,
      corr_idx, corr_d2, corr_lb, corr_q, corr_n, partials, n_active, knn_list, knn_fallback, knn_redo, wave_recs, t_hist, worklist, rec_list, rec_counts, search_list;
  int* h_active = nullptr;  // pinned: [0] the polled active-pair counter, [4], [5] the ICP loop's progress words (stage_icp)
  int* h_active_dev = nullptr;   // the same words as the device addresses them
  unsigned icp_tag_counter = 0;   // (wraps; 0 is skipped: the words' idle value)
  // pinned staging of the slot / pair records (up and down): a copy from or to pageable memory stalls the stream for
  // tens of microseconds, which a single-pair registration of ~1.5 ms notices
  char* h_stage = nullptr;
  size_t h_stage_cap = 0;
  char* stage_host(size_t bytes) {
    if (bytes > h_stage_cap) {
      if (h_stage) HIPCHK(hipHostFree(h_stage));
      h_stage = nullptr; h_stage_cap = 0;
      const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
      HIPCHK(hipHostMalloc((void**)&h_stage, want));
      h_stage_cap = want;
    }
    return h_stage;
  }
  // descriptors of k_copy_many (cache restore / store): their own pinned and device buffers, grown on demand
  char* h_copy = nullptr; size_t h_copy_cap = 0;
  DevBuf d_copy;
  void copy_many(const std::vector<s3d::CopyDesc>& descs) {
    if (descs.empty()) return;
    const size_t bytes = sizeof(s3d::CopyDesc) * descs.size();
    if (bytes > h_copy_cap) {
      HIPCHK(hipStreamSynchronize(stream));               // (an earlier upload may still read the old buffer)
      if (h_copy) HIPCHK(hipHostFree(h_copy));
      if (d_copy.p) HIPCHK(hipFree(d_copy.p));
      h_copy = nullptr; d_copy.p = nullptr; h_copy_cap = 0;
      const size_t want = std::max<size_t>(2 * bytes, 1 << 14);
      HIPCHK(hipHostMalloc((void**)&h_copy, want));
      HIPCHK(hipMalloc(&d_copy.p, want));
      h_copy_cap = d_copy.cap = want;
    }
    std::memcpy(h_copy, descs.data(), bytes);
    HIPCHK(hipMemcpyAsync(d_copy.p, h_copy, bytes, hipMemcpyHostToDevice, stream));
    unsigned long long most = 0;
    for (const s3d::CopyDesc& d : descs) most = std::max(most, d.bytes);
    const unsigned bx = (unsigned)std::min<unsigned long long>(std::max<unsigned long long>(most / (16ull * s3d::kBlock * 4), 1), 256);
    for (size_t first = 0; first < descs.size(); first += 32768) {   // (gridDim.y is limited to 65535)
      const unsigned ny = (unsigned)std::min<size_t>(descs.size() - first, 32768);
      s3d::k_copy_many<<<dim3(bx, ny), s3d::kBlock, 0, stream>>>((const s3d::CopyDesc*)d_copy.p + first);
    }
  }
  hipEvent_t ev[8] = {};
  // (round 6) a second stream for the k-NN pre-pass of a SMALL batch, which does not fill the chip: K4 runs there while
  // the first correspondence pass - which needs the grid, not the normals - runs on the context's stream (Batch::run_all).
  // Only for a context that owns a plain stream (no caller stream, CU mask or priority to carry over).
  bool plain_stream = false;
  // a private second context on the same device (workspace and stream of its own), created on first use:
  // create_constraint_impl runs the FINE registration's pre-pass there while the coarse one registers
  s3d_context* twin = nullptr;
  hipEvent_t twin_ev = nullptr;
  hipStream_t side_stream = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  bool ensure_side_stream() {
    if (!plain_stream) return false;
    if (!side_stream) {
      if (hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking) != hipSuccess) { side_stream = nullptr; return false; }
      for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&side_ev[i], hipEventDisableTiming) != hipSuccess) {
          for (int j = 0; j < 2; ++j) if (side_ev[j]) { (void)hipEventDestroy(side_ev[j]); side_ev[j] = nullptr; }
          (void)hipStreamDestroy(side_stream); side_stream = nullptr;
          return false;
        }
    }
    return true;
  }
  std::vector<hipEvent_t> nn_ev;
  // s3d_cloud_upload_many: per worker thread a stream and two pinned slots (grown on demand, kept)
  struct UploadLane {
    hipStream_t st = nullptr;
    char* pinned[2] = {nullptr, nullptr};
    size_t cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
  };
  std::vector<UploadLane> upload_lanes;
  int upload_threads_cap = 0;       // s3d_context_set_upload_threads (0: up to 8)
  void release_upload_lanes() {
    for (UploadLane& L : upload_lanes) {
      if (L.st) { (void)hipStreamSynchronize(L.st); (void)hipStreamDestroy(L.st); }
      for (int k = 0; k < 2; ++k) {
        if (L.pinned[k]) (void)hipHostFree(L.pinned[k]);
        if (L.ev[k]) (void)hipEventDestroy(L.ev[k]);
      }
    }
    upload_lanes.clear();
  }

  // ONE device allocation for the whole workspace, carved at 2 MiB boundaries: a few hundred MB per
  // array in separate hipMallocs left some processes with 3x slower streaming kernels on this pool
  // (fragmented page mappings); one large arena gets the driver's largest fragments.
  DevBuf arena;
  DevBuf staging;   // raw host floats of an upload with stride != 4 (grown on demand)
  void carve(std::initializer_list<std::pair<DevBuf*, size_t>> reqs) {
    const size_t gran = (size_t)2 << 20;
    size_t total = 0;
    for (auto& r : reqs) total += (r.second + gran - 1) / gran * gran;
    if (total > arena.cap) {
      if (arena.p) HIPCHK(hipFree(arena.p));
      arena.p = nullptr; arena.cap = 0;
      const size_t want = total + total / 8;
      HIPCHK(hipMalloc(&arena.p, want));
      arena.cap = want;
      static const bool print_arena = getenv("S3D_DBG_ARENA") != nullptr;   // (the library's only environment variable: a debug print, read once)
      if (print_arena) std::fprintf(stderr, "[s3d] arena %p + %zu MiB\n", arena.p, want >> 20);
    }
    size_t off = 0;
    for (auto& r : reqs) {
      r.first->p = (char*)arena.p + off;
      r.first->cap = r.second;
      off += (r.second + gran - 1) / gran * gran;
    }
  }
  void release_all() {
    cache_clear();
    if (arena.p) { (void)hipFree(arena.p); arena.p = nullptr; arena.cap = 0; }
    if (staging.p) { (void)hipFree(staging.p); staging.p = nullptr; staging.cap = 0; }
  }
};

namespace {

int fail(s3d_context* ctx, const HipError& e) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "HIP error %d (%s) at s3d_api.hip:%d: %s", (int)e.e, hipGetErrorString(e.e), e.line,
                e.what);
  if (ctx) ctx->err = buf;
  return e.status;
}

// The boundary never lets an exception out (an extern "C" function left by unwinding is std::terminate in the host
// application; the reference's callers catch std::exception and log: ScanSensor.cpp:74-77, :124-127, :159-166).  Called
// from a catch (...) block of an entry point: classifies the exception in flight, records the message, returns the status.
int fail_current(s3d_context* ctx, std::string* err_out = nullptr) noexcept {
  int status = S3D_STATUS_BACKEND_ERROR;
  const char* msg = "unknown exception";
  char buf[256];
  try {
    throw;
  } catch (const HipError& e) {
    try { const int st = fail(ctx, e); if (err_out && ctx) *err_out = ctx->err; return st; } catch (...) { return e.status; }
  } catch (const std::bad_alloc&) {
    msg = "out of host memory (std::bad_alloc)";
  } catch (const std::length_error& e) {
    std::snprintf(buf, sizeof buf, "size beyond what a host container can hold (std::length_error: %s)", e.what());
    msg = buf;
    status = S3D_STATUS_INVALID_ARGUMENT;
  } catch (const std::exception& e) {
    std::snprintf(buf, sizeof buf, "host exception: %s", e.what());
    msg = buf;
  } catch (...) {
  }
  try {
    if (ctx) ctx->err = msg;
    if (err_out) *err_out = msg;
  } catch (...) {
  }
  return status;
}

// device -> host copy ON THE CONTEXT'S STREAM, then a wait for that stream: a plain hipMemcpy runs on the legacy null
// stream, which every blocking stream of the process (a CU-masked context's, an application's default stream) waits
// for and is waited for by - a 'reserved' context would serialise with the sweep it is meant to run beside
static void copy_to_host(s3d_context* ctx, void* dst, const void* src, size_t bytes);

// ------------------------------------------------------------------ one batch of align() jobs

struct Batch {
  s3d_context* ctx;
  RunParams rp{};
  s3d_exec_options opts{};
  std::vector<const s3d_cloud*> slot_clouds;
  std::vector<SlotDev> h_slots;
  std::vector<int> knn_slots;   // slots that get the k-NN pre-pass (stage_normals)
  std::vector<PairDev> h_pairs;
  int max_n = 0, max_n_t = 0, nb_sort = 0, nb_head = 0, accum_blocks = 1;
  // cross-call pre-pass cache (s3d_exec_options.cache_prepass): the slots are ordered so that [0, Cu) are the clouds
  // this call has to filter / grid itself and [Cu, C()) are restored from the context's cache
  bool use_cache = false;
  int Cu = 0;
  std::vector<CacheEntry*> slot_entry;   // per slot: the cache entry it is restored from (nullptr: computed here)
  std::vector<char> slot_has_normals;    // per slot: its normals came with the entry
  long long cell_cap_max = 1ll << 24;  // 24-bit cell ids sort in 3 radix passes; map jobs raise it (4 passes)
  long long max_cell_cap = 0;
  size_t total_pts = 0, total_cells = 0, total_corr = 0;

  SlotDev* d_slots() { return (SlotDev*)ctx->slots.p; }
  PairDev* d_pairs() { return (PairDev*)ctx->pairs.p; }
  uint32_t* kA() { return (uint32_t*)ctx->keysA.p; }
  uint32_t* kB() { return (uint32_t*)ctx->keysB.p; }
  uint32_t* vA() { return (uint32_t*)ctx->valsA.p; }
  uint32_t* vB() { return (uint32_t*)ctx->valsB.p; }
  float4* filt() { return (float4*)ctx->filt.p; }
  float4* sorted() { return (float4*)ctx->sorted.p; }
  CorrVec* sorted3() { return has_sorted3 ? (CorrVec*)ctx->sorted3.p : nullptr; }
  bool has_sorted3 = false;
  NormalRec* normals() { return (NormalRec*)ctx->normals.p; }
  uint32_t* cells() { return (uint32_t*)ctx->cell_start.p; }
  int C() const { return (int)h_slots.size(); }
  int P() const { return (int)h_pairs.size(); }

  void set_params(const s3d_reg_params* p, const s3d_exec_options* o) {
    if (o) opts = *o;
    dbg_nn = (int)(opts.debug_flags & 0x001FFFFFu);
    if (opts.grid_cells_per_point <= 0) opts.grid_cells_per_point = 2;
    rp.algorithm = p->registration_algorithm == S3D_ALG_ICP ? 0 : 1;
    rp.k = p->correspondence_randomness;
    rp.max_iterations = p->maximum_iterations;
    rp.max_inner = p->maximum_optimizer_iterations;
    rp.force_iterations = opts.force_iterations;
    rp.max_corr = p->max_correspondence_distance;
    rp.dist_threshold = p->max_correspondence_distance * p->max_correspondence_distance;
    rp.rotation_epsilon = p->rotation_epsilon;
    rp.transformation_epsilon = p->transformation_epsilon;
    rp.fit_range = p->max_correspondence_distance;  // PointCloudSensor.cpp:73 (un-squared)
    rp.gicp_epsilon = 0.001;                        // PCL default gicp_epsilon_
    rp.leaf = (float)p->point_cloud_density;        // PointCloudSensor.cpp:196 setLeafSize(double -> float)
    rp.h0 = p->point_cloud_density > 0 ? (float)(2.0 * p->point_cloud_density) : 0.25f;
  }

  int add_slot(const s3d_cloud* c, std::map<const s3d_cloud*, int>& index) {
    auto it = index.find(c);
    if (it != index.end()) return it->second;
    SlotDev s;
    std::memset(&s, 0, sizeof s);
    s.raw = c->d;
    s.n_raw = c->n;
    s.off = (int)total_pts;   // slots start on multiples of 64 points: a wave's 12-byte and 4-byte records begin on a cache line
    total_pts += (size_t)((c->n + 63) & ~63);
    const long long cap =
        std::min<long long>(std::max<long long>((long long)opts.grid_cells_per_point * c->n, 64), cell_cap_max);
    s.cell_cap = (int)cap;
    max_cell_cap = std::max(max_cell_cap, cap);
    s.cell_off = (int)total_cells;
    total_cells += (size_t)cap + 1;
    max_n = std::max(max_n, c->n);
    const int id = (int)h_slots.size();
    h_slots.push_back(s);
    slot_clouds.push_back(c);
    index[c] = id;
    return id;
  }

  void add_pairs(int n_pairs, s3d_cloud* const* sources, s3d_cloud* const* targets, const double* guesses) {
    std::map<const s3d_cloud*, int> index;
    for (int p = 0; p < n_pairs; ++p) {
      PairDev P;
      std::memset(&P, 0, sizeof P);
      P.slot_s = add_slot(sources[p], index);
      P.slot_t = add_slot(targets[p], index);
      P.corr_off = (int)total_corr;
      total_corr += (size_t)((targets[p]->n + 63) & ~63);
      max_n_t = std::max(max_n_t, targets[p]->n);
      for (int i = 0; i < 16; ++i) P.guess.m[i] = (float)guesses[(size_t)p * 16 + i];  // :70 cast<float>()
      h_pairs.push_back(P);
    }
  }

  // icp_buffers = false: voxel filter / search grid only (map building), no normals or correspondences
  CacheKey cache_key(int slot) const {
    CacheKey k;
    k.uid = slot_clouds[(size_t)slot]->uid;
    std::memcpy(&k.leaf_bits, &rp.leaf, 4);
    std::memcpy(&k.h0_bits, &rp.h0, 4);
    k.cell_cap = h_slots[(size_t)slot].cell_cap;
    k.layout = fused ? 1 : 0;
    return k;
  }

  // cached clouds to the back of the slot list (the staging kernels then cover the first Cu slots only)
  void order_slots_for_cache(bool need_sorted3) {
    const int n = C();
    Cu = n;
    slot_entry.assign((size_t)n, nullptr);
    slot_has_normals.assign((size_t)n, 0);
    if (!use_cache || n == 0) return;
    ++ctx->cache_clock;
    std::vector<CacheEntry*> ent((size_t)n, nullptr);
    int hits = 0;
    for (int i = 0; i < n; ++i) {
      auto it = ctx->cache.find(cache_key(i));
      // (an entry made by a map job has no xyz-only copy: a registration treats it as a miss and replaces it)
      if (it != ctx->cache.end() && (it->second.has_sorted3 || !need_sorted3)) { ent[i] = &it->second; it->second.last_use = ctx->cache_clock; ++hits; }
    }
    ctx->cache_hits += hits; ctx->cache_misses += n - hits;
    if (hits == 0) return;
    std::vector<int> order, remap((size_t)n);
    for (int i = 0; i < n; ++i) if (!ent[i]) order.push_back(i);
    Cu = (int)order.size();
    for (int i = 0; i < n; ++i) if (ent[i]) order.push_back(i);
    std::vector<SlotDev> hs((size_t)n);
    std::vector<const s3d_cloud*> sc((size_t)n);
    total_pts = 0; total_cells = 0;
    for (int j = 0; j < n; ++j) {
      const int i = order[(size_t)j];
      remap[(size_t)i] = j;
      hs[j] = h_slots[(size_t)i];
      sc[j] = slot_clouds[(size_t)i];
      slot_entry[j] = ent[i];
      hs[j].off = (int)total_pts;
      total_pts += (size_t)((hs[j].n_raw + 63) & ~63);
      hs[j].cell_off = (int)total_cells;
      total_cells += (size_t)hs[j].cell_cap + 1;
    }
    h_slots.swap(hs);
    slot_clouds.swap(sc);
    for (PairDev& pr : h_pairs) { pr.slot_s = remap[(size_t)pr.slot_s]; pr.slot_t = remap[(size_t)pr.slot_t]; }
  }

  // restore the cached clouds into this batch's arrays (after allocate() has carved them)
  void restore_from_cache() {
    std::vector<CopyDesc> cp;
    for (int j = Cu; j < C(); ++j) {
      const CacheEntry& e = *slot_entry[(size_t)j];
      const SlotDev& sl = h_slots[(size_t)j];
      const size_t n = (size_t)sl.n;
      if (n) {
        if (!fused) cp.push_back({e.block + e.o_filt, filt() + sl.off, 16 * n});
        cp.push_back({e.block + e.o_sorted, sorted() + sl.off, 16 * n});
        if (has_sorted3) cp.push_back({e.block + e.o_sorted3, sorted3() + sl.off, sizeof(CorrVec) * n});
        if (slot_has_normals[(size_t)j]) cp.push_back({e.block + e.o_normals, normals() + sl.off, sizeof(NormalRec) * n});
      }
      cp.push_back({e.block + e.o_cells, cells() + sl.cell_off, 4 * ((size_t)sl.g.ncells + 1)});
    }
    ctx->copy_many(cp);     // one launch for all of them
  }

  // after download(): keep the products of the clouds this call computed (and normals a cached cloud lacked)
  void store_to_cache(bool have_normals) {
    if (!use_cache) return;
    hipStream_t st = ctx->stream;
    const int k = (have_normals && rp.k >= 1 && rp.k <= 64) ? rp.k : 0;
    std::vector<CopyDesc> cp;    // every copy of this call as one launch (after the loop)
    for (int j = 0; j < C(); ++j) {
      const SlotDev& sl = h_slots[(size_t)j];
      const size_t n = (size_t)sl.n;
      const bool normals_now = k != 0 && sl.want_normals;      // K4 ran on this slot in this call
      if (j >= Cu) {   // restored from the cache: at most the normals are new
        CacheEntry& e = *slot_entry[(size_t)j];
        if (normals_now && n) {
          cp.push_back({normals() + sl.off, e.block + e.o_normals, sizeof(NormalRec) * n});
          e.k_normals = k;
        }
        continue;
      }
      {   // an entry this call could not use (no xyz-only copy: made by a map job) is replaced
        auto stale = ctx->cache.find(cache_key(j));
        if (stale != ctx->cache.end()) { HIPCHK(hipStreamSynchronize(st)); ctx->cache_drop(stale); }
      }
      CacheEntry e;
      const size_t cells_n = (size_t)sl.g.ncells + 1;
      auto place = [&](size_t bytes) { const size_t o = e.bytes; e.bytes += (bytes + 255) & ~(size_t)255; return o; };
      e.o_filt = place(fused ? 0 : 16 * n); e.o_sorted = place(16 * n); e.o_sorted3 = place(sizeof(CorrVec) * n);
      e.o_normals = place(sizeof(NormalRec) * n); e.o_cells = place(4 * cells_n);
      e.bytes = std::max<size_t>(e.bytes, 256);
      // make room: least-recently-used entries that this call does not use
      while (ctx->cache_bytes + e.bytes > ctx->cache_limit) {
        auto victim = ctx->cache.end();
        for (auto it = ctx->cache.begin(); it != ctx->cache.end(); ++it)
          if (it->second.last_use < ctx->cache_clock && (victim == ctx->cache.end() || it->second.last_use < victim->second.last_use))
            victim = it;
        if (victim == ctx->cache.end()) break;
        HIPCHK(hipStreamSynchronize(st));
        ctx->cache_drop(victim);
      }
      if (ctx->cache_bytes + e.bytes > ctx->cache_limit) continue;   // does not fit: stay uncached
      // best effort: the registration results are already on the host - a full (or shared) GPU must not turn an
      // optional optimisation into a failed call
      if (hipMalloc((void**)&e.block, e.bytes) != hipSuccess) { (void)hipGetLastError(); e.block = nullptr; continue; }
      if (n) {
        if (!fused) cp.push_back({filt() + sl.off, e.block + e.o_filt, 16 * n});
        cp.push_back({sorted() + sl.off, e.block + e.o_sorted, 16 * n});
        if (has_sorted3) cp.push_back({sorted3() + sl.off, e.block + e.o_sorted3, sizeof(CorrVec) * n});
        if (normals_now) cp.push_back({normals() + sl.off, e.block + e.o_normals, sizeof(NormalRec) * n});
      }
      cp.push_back({cells() + sl.cell_off, e.block + e.o_cells, 4 * cells_n});
      e.k_normals = normals_now ? k : 0;
      e.snap = sl;
      e.has_sorted3 = has_sorted3;
      e.last_use = ctx->cache_clock;
      ctx->cache_bytes += e.bytes;
      ctx->cache[cache_key(j)] = e;
    }
    ctx->copy_many(cp);
    HIPCHK(hipStreamSynchronize(st));   // the arena may be re-carved by the next call on another stream order
  }

  uint32_t leaf_bits() const { uint32_t b; std::memcpy(&b, &rp.leaf, 4); return b; }
  bool registration_batch = false;   // set by the callers of run_all(): GICP / point-to-plane on this batch's pairs
  void assign_want_normals() {
    // which clouds need the k-NN pre-pass: GICP uses the covariances of both clouds of a pair, point-to-plane only
    // the normals of the searched one (PCL target = slam3d source); a batch without pairs is s3d_knn_normals
    for (SlotDev& sl : h_slots) sl.want_normals = h_pairs.empty() ? 1 : 0;
    for (const PairDev& pr : h_pairs) {
      h_slots[pr.slot_s].want_normals = 1;
      if (rp.algorithm != 0) h_slots[pr.slot_t].want_normals = 1;
    }
  }
  void allocate(bool icp_buffers = true) {
    has_sorted3 = icp_buffers;
    fused = registration_batch && fused_wanted();    // (before the cache look-up: the layout is part of an entry's key)
    // a caller-held cloud that the fused pre-pass could not serve once (a property of the cloud and the voxel size: PCL's
    // index overflow, a key beyond 32 bits, a centroid outside its cell) would double the cost of every batch it is
    // part of: remembered per (cloud, voxel size), such batches start on the two-sort path
    if (fused && !ctx->fused_unservable.empty())
      for (const s3d_cloud* c : slot_clouds) fused = fused && !ctx->fused_unservable.count({c->uid, leaf_bits()});
    order_slots_for_cache(icp_buffers);
    if (total_pts > (size_t)0x7FFFFFF0 || total_cells > (size_t)0x7FFFFFF0 || total_corr > (size_t)0x7FFFFFF0)
      throw HipError{hipErrorInvalidValue, "batch too large for 32-bit offsets", __LINE__};
    nb_sort = std::max(1, cdiv(max_n, kSortTile));
    nb_head = std::max(1, cdiv(max_n, kBlock));
    // REAL blocks per pair in the accumulate kernels: enough to fill the chip for small batches, few long-running
    // ones for large batches.  Speed only: the sums are defined over kAccumVB virtual blocks per pair whatever
    // this number is (a divisor of kAccumVB), so a pair's result does not depend on the batch it is part of.
    accum_blocks = kAccumVB;
    while (accum_blocks > 4 && (long long)accum_blocks * std::max(1, P()) > 1024) accum_blocks >>= 1;
    while (accum_blocks > 1 && accum_blocks * kBlock * 2 > std::max(max_n_t, 1)) accum_blocks >>= 1;
    {   // A/B and the invariance test (s3d_exec_options.debug_accum_blocks): any divisor of kAccumVB
      const int v = opts.debug_accum_blocks;
      if (v != 0 && !(v >= 1 && v <= kAccumVB && kAccumVB % v == 0))
        throw HipError{hipErrorInvalidValue, "debug_accum_blocks must be 0 or a divisor of the 32 virtual blocks", __LINE__,
                       S3D_STATUS_INVALID_ARGUMENT};
      if (v != 0) accum_blocks = v;
    }
    const size_t np = std::max<size_t>(total_pts, 4);
    const size_t nc = std::max<size_t>(total_corr, 4);
    const size_t npi = icp_buffers ? np : 4;
    has_sorted3 = icp_buffers;
    // the batch's records live in ONE region - slots | pairs | a 16-byte tail (the sort's error word) | the list of the
    // slots K4 runs on - so that they go up in one copy and come back in one (round 5: six 4-us copies and a fill in front
    // of and behind a lone registration were three)
    ctx->carve({{&ctx->slots, records_bytes() + 64},
                {&ctx->keysA, 4 * np}, {&ctx->keysB, 4 * np}, {&ctx->valsA, 4 * np}, {&ctx->valsB, 4 * np},
                {&ctx->filt, 16 * np}, {&ctx->sorted, 16 * np}, {&ctx->sorted3, 12 * npi}, {&ctx->normals, sizeof(NormalRec) * npi}, {&ctx->moments, 72 * npi},
                {&ctx->cell_start, 4 * std::max<size_t>(total_cells, 4)},
                {&ctx->counts, 4 * (size_t)std::max(1, C()) * (1 << kSortMaxBits) * nb_sort},
                {&ctx->digit_tot, 4 * (size_t)std::max(1, C()) * (1 << kSortMaxBits) * kSortPlaces},
                {&ctx->blockcnt, 4 * (size_t)std::max(1, C()) * nb_head},
                {&ctx->blockbb, 24 * (size_t)std::max(1, C()) * nb_head},
                {&ctx->corr_idx, 4 * nc}, {&ctx->corr_d2, 4 * nc}, {&ctx->corr_lb, 4 * nc},
                {&ctx->corr_q, 16 * nc}, {&ctx->corr_n, 16 * nc},
                {&ctx->partials, 8 * (size_t)std::max(1, P()) * kAccumVB * GQ_NACC},
                {&ctx->n_active, 64 + 4 * 64 * sizeof(int)},
                // the queries a scan27 pass declines (pair, index): up to every query of every pair - NOT bounded by the
                // point count of the batch, whose clouds are shared by the pairs of a sweep
                {&ctx->worklist, icp_buffers ? 8 * nc : 8},
                {&ctx->wave_recs, sizeof(WaveRec) * (icp_buffers ? nc / kWave + 1 : 1)},
                {&ctx->rec_list, sizeof(uint4) * (icp_buffers ? rec_list_entries() : 1)},
                {&ctx->rec_counts, sizeof(int) * 2 * (kNNRecSublists + kNNSearchSublists)},
                {&ctx->search_list, sizeof(uint4) * (icp_buffers && settled_wanted() ? (size_t)kNNSearchSublists * (size_t)search_sub_cap() : 1)},
                {&ctx->t_hist, sizeof(Mat4f) * (icp_buffers ? (size_t)std::max(1, P()) * (size_t)hist_stride() : 1)},
                {&ctx->knn_fallback, sizeof(int) * npi},
                // four lists, a quarter each: near declines from the front; from the back of the second, third and fourth
                // quarter the ring search's leftovers, the far list and the far entries that need more rings (what a launch
                // of the ring search hands on is appended to other lists while it reads its own)
                {&ctx->knn_redo, sizeof(int2) * 4 * npi}});
    ctx->pairs.p = (char*)ctx->slots.p + slots_bytes();
    ctx->pairs.cap = sizeof(PairDev) * (size_t)P();
    ctx->knn_list.p = (char*)ctx->slots.p + slots_bytes() + pairs_bytes() + 16;
    ctx->knn_list.cap = sizeof(int) * (size_t)C();
    if (!ctx->h_active) {
      // (mapped + coherent asked for by name: the device's stores to the progress words must be visible to the spinning
      // host thread without a stream wait, whatever the runtime's default for plain pinned memory is)
      if (hipHostMalloc((void**)&ctx->h_active, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        ctx->h_active = nullptr;
        HIPCHK(hipHostMalloc((void**)&ctx->h_active, 64));
      }
      std::memset(ctx->h_active, 0, 64);
      if (hipHostGetDevicePointer((void**)&ctx->h_active_dev, ctx->h_active, 0) != hipSuccess) ctx->h_active_dev = nullptr;
    }

    assign_want_normals();
    for (SlotDev& sl : h_slots)
      for (int a = 0; a < 3; ++a) { sl.bb[a] = 0xFFFFFFFFu; sl.bb[3 + a] = 0u; }   // empty bbox: k_bbox<0> starts from it
    for (int j = Cu; j < C(); ++j) {   // cached clouds: the device-computed part of the slot record, normals if they match
      const CacheEntry& e = *slot_entry[(size_t)j];
      SlotDev& sl = h_slots[(size_t)j];
      sl.n = e.snap.n; sl.n_sort = 0;
      std::memcpy(sl.bb, e.snap.bb, sizeof sl.bb);
      sl.vp = e.snap.vp; sl.g = e.snap.g; sl.fz = e.snap.fz;
      if (sl.want_normals && icp_buffers && e.k_normals != 0 && e.k_normals == rp.k && (e.has_sorted3 || !icp_buffers)) {
        slot_has_normals[(size_t)j] = 1;
        sl.want_normals = 0;
      }
    }
    h_slots0 = h_slots;      // (what run_all() starts a second attempt from)
    h_pairs0 = h_pairs;
    upload_records();
    restore_from_cache();
  }
  std::vector<SlotDev> h_slots0;
  std::vector<PairDev> h_pairs0;
  size_t slots_bytes() const { return sizeof(SlotDev) * (size_t)C(); }
  size_t pairs_bytes() const { return sizeof(PairDev) * (size_t)P(); }
  size_t records_bytes() const { return slots_bytes() + pairs_bytes() + 16 + sizeof(int) * (size_t)C(); }
  int* sort_err_dev() { return (int*)((char*)ctx->slots.p + slots_bytes() + pairs_bytes()); }
  void upload_records() {
    hipStream_t st = ctx->stream;
    const size_t bs = slots_bytes(), bp = pairs_bytes();
    knn_slots.clear();               // the slots K4 runs on (stage_normals)
    for (int c = 0; c < C(); ++c)
      if (h_slots[(size_t)c].want_normals) knn_slots.push_back(c);
    char* stage = ctx->stage_host(records_bytes());
    if (bs) std::memcpy(stage, h_slots.data(), bs);
    if (bp) std::memcpy(stage + bs, h_pairs.data(), bp);
    std::memset(stage + bs + bp, 0, 16);           // (the sort's error word starts at zero)
    if (!knn_slots.empty()) std::memcpy(stage + bs + bp + 16, knn_slots.data(), sizeof(int) * knn_slots.size());
    HIPCHK(hipMemcpyAsync(ctx->slots.p, stage, bs + bp + 16 + sizeof(int) * knn_slots.size(), hipMemcpyHostToDevice, st));
  }

  // segmented LSD radix sort of (keys, vals) of every slot: `passes` digits of `bits` bits (8, 9 or 10).
  // input in A; result in A for even `passes`, in B for odd.
  // One sweep per pass (k_sort_onesweep, decoupled look-back); hist_done: the kernel that produced the keys has
  // counted the digit totals of all passes (k_keys_hist with sweep_passes) and zeroed the look-back rows.
  // The three-kernels-per-pass form (k_sort_hist / scan / scatter) reads the keys twice but never waits for another
  // tile: it is the faster one once the batch is large (256 pairs of 100 k points: voxel + grid 4.22 -> 4.07 ms,
  // equal at 128 pairs, 0.03 ms slower at 32), the one-sweep form - 5 launches instead of 11 per sort - for a small
  // batch and a lone pair (-25 us).  S3D_DBG_SORT_CLASSIC / _ONESWEEP force one or the other (A/B).
  bool sort_classic = false;
  void sort_choose() {
    sort_classic = (opts.debug_flags & S3D_DBG_SORT_CLASSIC) ? true : (opts.debug_flags & S3D_DBG_SORT_ONESWEEP) ? false
                   : (long long)Cu * (long long)max_n >= 40000000ll;
    if (max_n >= (1 << 28)) sort_classic = true;   // the one-sweep state words hold a 28-bit count next to their tag
  }
  bool sort_used = false;
  void sort_prepare(int nslots) {   // before the kernel that counts the digit totals
    if (!sort_classic && nslots > 0)
      HIPCHK(hipMemsetAsync(ctx->digit_tot.p, 0, sizeof(uint32_t) * (size_t)nslots * kSortPlaces * (1 << kSortMaxBits), ctx->stream));
  }
  // the digits of a sort from the number of key bits: 8-bit digits unless a ninth (then a tenth) bit per digit removes
  // a pass (18 bits: 2 x 9 instead of 3 x 8; 24: 3 x 8; 30: 4 x 8 - wider digits cost more per pass than they save,
  // see s3d_kernels.h K2a); S3D_DBG_SORT_FULL_KEYS: always 8-bit digits
  struct SortPlan { int passes, bits; };
  SortPlan sort_plan(int key_bits) const {
    const int p8 = std::max(1, cdiv(key_bits, 8));
    if (opts.debug_flags & S3D_DBG_SORT_FULL_KEYS) return {p8, 8};
    if (cdiv(key_bits, 9) < p8) return {cdiv(key_bits, 9), 9};
    if (cdiv(key_bits, 10) < p8) return {cdiv(key_bits, 10), 10};
    return {p8, 8};
  }
  static int bits_for(long long max_value) { int b = 1; while (b < 32 && (max_value >> b) != 0) ++b; return b; }
  // (measured, round 4: the clouds of a batch sorted in GROUPS of 16 ... 128, all passes of a group back to back so
  // that its 1.6 MB per cloud stay in the 256 MB memory-side cache between passes - voxel + grid 4.0 -> 7.3 / 5.7 / 4.7 /
  // 4.2 ms for groups of 16 / 32 / 64 / 128 of the 512 clouds: the smaller launches lose more than the cache gives)
  void sort(SortPlan plan, int nslots, bool hist_done = false) {
    switch (plan.bits) {
      case 8: sort_bits<8>(plan.passes, nslots, hist_done); break;
      case 9: sort_bits<9>(plan.passes, nslots, hist_done); break;
      default: sort_bits<10>(plan.passes, nslots, hist_done); break;
    }
  }
  template <int BITS>
  void sort_bits(int passes, int nslots, bool hist_done) {
    hipStream_t st = ctx->stream;
    if (nslots <= 0) return;
    constexpr int NB = 1 << BITS;
    uint32_t *ki = kA(), *vi = vA(), *ko = kB(), *vo = vB();
    uint32_t* cnt = (uint32_t*)ctx->counts.p;
    uint32_t* dtot = (uint32_t*)ctx->digit_tot.p;
    SlotDev* gslots = d_slots();
    const unsigned blocks = (unsigned)((nslots >= 8 ? cdiv(nslots, 8) * 8 : nslots) * nb_sort);
    if (!sort_classic) {
      sort_used = true;     // (the error word was zeroed with the records' upload)
      if (!hist_done) {
        sort_prepare(nslots);
        k_sort_hist_all<BITS><<<dim3(nb_sort, nslots), kBlock, 0, st>>>(gslots, ki, dtot, cnt, passes, nb_sort);
      }
      int* err = sort_err_dev();
      for (int p = 0; p < passes; ++p) {
        // (hist_done: the keys come from k_keys_hist, whose values are the identity and are not stored: the first pass
        // takes an element's index for its value)
        if (p == 0 && hist_done)
          k_sort_onesweep<BITS, true><<<blocks, kBlock, 0, st>>>(gslots, ki, vi, ko, vo, cnt, dtot, p, nb_sort, nslots, err);
        else
          k_sort_onesweep<BITS, false><<<blocks, kBlock, 0, st>>>(gslots, ki, vi, ko, vo, cnt, dtot, p, nb_sort, nslots, err);
        std::swap(ki, ko);
        std::swap(vi, vo);
      }
      return;
    }
    for (int p = 0; p < passes; ++p) {
      const int shift = BITS * p;
      if (p > 0 || !hist_done) k_sort_hist<BITS><<<dim3(nb_sort, nslots), kBlock, 0, st>>>(gslots, ki, cnt, shift, nb_sort);
      if (nb_sort <= kSortTileMajor) k_sort_scan_tiles<BITS><<<nslots, 256, 0, st>>>(gslots, cnt, dtot, nb_sort);
      else k_sort_scan_rows<BITS><<<dim3(NB / (kBlock / kWave), nslots), kBlock, 0, st>>>(gslots, cnt, dtot, nb_sort);
      if (p == 0 && hist_done)
        k_sort_scatter<BITS, true><<<blocks, kBlock, 0, st>>>(gslots, ki, vi, ko, vo, cnt, dtot, shift, nb_sort, nslots);
      else
        k_sort_scatter<BITS, false><<<blocks, kBlock, 0, st>>>(gslots, ki, vi, ko, vo, cnt, dtot, shift, nb_sort, nslots);
      std::swap(ki, ko);
      std::swap(vi, vo);
    }
  }

  // K1 + K2: pcl::VoxelGrid of every slot that is not restored from the cache (or a plain copy when leaf <= 0)
  void stage_voxel() {
    hipStream_t st = ctx->stream;
    const int NS = Cu;
    if (NS == 0) return;
    sort_choose();
    if (rp.leaf > 0.f) {
      // voxel keys are up to 31 bits wide and the host does not know how wide (the bounding boxes live on the device)
      k_bbox<0><<<dim3(cdiv(std::max(max_n, 1), kBlock * 4), NS), kBlock, 0, st>>>(d_slots(), filt());
      k_voxel_params<<<cdiv(NS, 64), 64, 0, st>>>(d_slots(), rp, NS);
      sort_prepare(NS);
      k_keys_hist<0, 8><<<dim3(nb_sort, NS), kBlock, 0, st>>>(d_slots(), filt(), kA(), vA(), (uint32_t*)ctx->counts.p, nb_sort,
                                                                sort_classic ? 0 : 4, (uint32_t*)ctx->digit_tot.p);
      sort(SortPlan{4, 8}, NS, true);
      uint32_t* bc = (uint32_t*)ctx->blockcnt.p;
      k_heads_count<<<dim3(cdiv(nb_head, kHeadsChunksPerBlock), NS), kBlock, 0, st>>>(d_slots(), kA(), bc, nb_head);
      k_heads_scan<<<NS, kBlock, 0, st>>>(d_slots(), bc, nb_head);
      k_centroids<<<(unsigned)((NS >= 8 ? cdiv(NS, 8) * 8 : NS) * nb_head), kBlock, 0, st>>>(d_slots(), kA(), vA(), bc, filt(),
                                                                                               (unsigned int*)ctx->blockbb.p, nb_head, NS);
    } else {
      k_copy_raw<<<dim3(nb_head, NS), kBlock, 0, st>>>(d_slots(), filt());
    }
  }

  // K1 + K2 + K3 of a registration batch in ONE sort (round 5; s3d_core.h "K2 + K3 in one sort", k_centroids_fused): the
  // raw points sorted by (search cell, voxel) - the centroids then come out in cell order and the centroid kernel writes
  // the cell-sorted arrays and the cell table itself.  Against stage_voxel + stage_grid: no k_keys_hist<1>, no second
  // sort (two 9-bit passes at the benchmark), no k_grid_finalize, no `filt`.  Wanted by run_all() for GICP / point-to-plane
  // with a voxel filter and k <= 32 (cache entries of this layout hold no `filt` and are keyed apart); a slot it cannot
  // serve reports so (FusedGrid::ok < 0) and run_all() runs the batch again on the two-sort path.
  bool fused = false;
  bool k4_counters_zeroed = false;   // the pre-pass's first kernel has cleared K4's list counters (stage_normals need not)
  bool fused_wanted() const {
    return rp.leaf > 0.f && rp.k <= 32 && has_sorted3 && !(opts.debug_flags & S3D_DBG_NO_FUSED_PREPASS);
  }
  void stage_prepass_fused() {
    hipStream_t st = ctx->stream;
    const int NS = Cu;
    if (NS == 0) return;
    sort_choose();
    // (the first kernel also clears the sort's digit totals and K4's list counters: two fill launches less)
    k_bbox<0><<<dim3(cdiv(std::max(max_n, 1), kBlock * 4), NS), kBlock, 0, st>>>(
        d_slots(), filt(), sort_classic ? nullptr : (uint32_t*)ctx->digit_tot.p, (int*)ctx->n_active.p + 4);
    k4_counters_zeroed = true;
    k_voxel_params<<<cdiv(NS, 64), 64, 0, st>>>(d_slots(), rp, NS);
    k_fused_grid_params<<<cdiv(NS, 64), 64, 0, st>>>(d_slots(), NS);
    k_keys_hist<2, 8><<<dim3(nb_sort, NS), kBlock, 0, st>>>(d_slots(), filt(), kA(), vA(), (uint32_t*)ctx->counts.p, nb_sort,
                                                              sort_classic ? 0 : 4, (uint32_t*)ctx->digit_tot.p);
    sort(SortPlan{4, 8}, NS, true);      // (32-bit keys: the host does not know how many bits a slot's keys use)
    uint32_t* bc = (uint32_t*)ctx->blockcnt.p;
    k_heads_count<<<dim3(cdiv(nb_head, kHeadsChunksPerBlock), NS), kBlock, 0, st>>>(d_slots(), kA(), bc, nb_head);
    k_heads_scan<<<NS, kBlock, 0, st>>>(d_slots(), bc, nb_head);
    k_centroids_fused<<<(unsigned)((NS >= 8 ? cdiv(NS, 8) * 8 : NS) * nb_head), kBlock, 0, st>>>(
        d_slots(), kA(), vA(), bc, sorted(), sorted3(), cells(), nb_head, NS);
  }

  // K3: dense search grid + cell-sorted copy of every slot that is not restored from the cache
  void stage_grid() {
    hipStream_t st = ctx->stream;
    const int NS = Cu;
    if (NS == 0) return;
    const int from_centroids = rp.leaf > 0.f ? 1 : 0;   // stage_voxel's k_centroids gathered the bbox already
    if (!from_centroids) {
      k_slot_reset_bbox<<<NS, 64, 0, st>>>(d_slots());
      k_bbox<1><<<dim3(cdiv(std::max(max_n, 1), kBlock * 4), NS), kBlock, 0, st>>>(d_slots(), filt());
    }
    k_grid_params<<<NS, kWave, 0, st>>>(d_slots(), rp, from_centroids ? (const unsigned int*)ctx->blockbb.p : nullptr, nb_head);
    // cell ids are below the largest cell budget of the batch, which the host knows
    const SortPlan plan = sort_plan(bits_for(std::max<long long>(max_cell_cap, 2) - 1));
    sort_prepare(NS);
    uint32_t* cnt = (uint32_t*)ctx->counts.p;
    uint32_t* dtot = (uint32_t*)ctx->digit_tot.p;
    const int sweep = sort_classic ? 0 : plan.passes;
    switch (plan.bits) {
      case 8: k_keys_hist<1, 8><<<dim3(nb_sort, NS), kBlock, 0, st>>>(d_slots(), filt(), kA(), vA(), cnt, nb_sort, sweep, dtot); break;
      case 9: k_keys_hist<1, 9><<<dim3(nb_sort, NS), kBlock, 0, st>>>(d_slots(), filt(), kA(), vA(), cnt, nb_sort, sweep, dtot); break;
      default: k_keys_hist<1, 10><<<dim3(nb_sort, NS), kBlock, 0, st>>>(d_slots(), filt(), kA(), vA(), cnt, nb_sort, sweep, dtot); break;
    }
    sort(plan, NS, true);
    const bool in_a = plan.passes % 2 == 0;
    // (a slot -> XCD block map as in k_centroids gains nothing here: vals[] runs nearly in step with the cell order)
    k_grid_finalize<<<dim3(cdiv(max_n + 1, kBlock), NS), kBlock, 0, st>>>(d_slots(), filt(), in_a ? kA() : kB(),
                                                                          in_a ? vA() : vB(), sorted(), sorted3(), cells());
  }

  // run_all in two halves (create_constraint_impl): phase 1 = pre-pass + k-NN only, nothing waited for; phase 2 = the
  // same batch goes on from there (ICP loop, fitness, download; the two-sort rerun of a fused failure as ever)
  int phase = 0;
  void set_guess(int p, const double g[16]) {     // a guess that is known only after phase 1 (it is not read before the ICP loop)
    for (int i = 0; i < 16; ++i) h_pairs[(size_t)p].guess.m[i] = (float)g[i];
    if ((size_t)p < h_pairs0.size()) h_pairs0[(size_t)p].guess = h_pairs[(size_t)p].guess;
    HIPCHK(hipMemcpyAsync(&d_pairs()[p].guess, &h_pairs[(size_t)p].guess, sizeof(Mat4f), hipMemcpyHostToDevice, ctx->stream));
  }
  // K4.  overlap_k4: on the context's side stream, next to the first correspondence pass (run_all)
  bool overlap_k4 = false, k4_pending = false;
  void join_k4() {           // everything after this on the context's stream sees the normals
    if (!k4_pending) return;
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->side_ev[1], 0));
    k4_pending = false;
  }
  void stage_normals() {
    hipStream_t st = overlap_k4 ? ctx->side_stream : ctx->stream;
    if (C() == 0) return;
    const int k = std::max(1, std::min(rp.k, 64));
    if (k > 32) {
      k_normals<<<dim3(nb_head, C()), kBlock, (size_t)k * kBlock * 8, st>>>(d_slots(), filt(), sorted(), cells(), normals(), k);
      return;
    }
    double* mom = (double*)ctx->moments.p;
    const size_t mom_plane = std::max<size_t>(total_pts, 4);   // nine planes, one double per point each
    const int NL = (int)knn_slots.size();   // (the list went up with the records: upload_records)
    if (NL == 0) return;
    int* d_list = (int*)ctx->knn_list.p;
    const int slots8 = NL >= 8 ? cdiv(NL, 8) * 8 : NL;
    dim3 grid((unsigned)(slots8 * nb_head));
    // the points whose normal the closed form declines (s3d_kernels.h): a device-side list, counted in n_active[4]
    int* fb_count = (int*)ctx->n_active.p + 4;
    int* fb_list = (int*)ctx->knn_fallback.p;
    if (!k4_counters_zeroed)
      HIPCHK(hipMemsetAsync(fb_count, 0, 5 * sizeof(int), st));   // eigen fallback, near redo, far list, ring-search leftovers, deep entries
    k4_counters_zeroed = false;
    // k = 20 (the reference default): 32-bit keys + med3 insertion; what it does not answer goes through the exact
    // 64-bit search (redo list, counted in n_active[5]).  S3D_DBG_KNN_EXACT64: the 64-bit search for every point.
    const bool exact64 = (opts.debug_flags & S3D_DBG_KNN_EXACT64) != 0;
    if (k == 20 && !exact64 && max_n < kKnn3MaxPoints) {
      int* redo_count = fb_count + 1;
      int2* redo_list = (int2*)ctx->knn_redo.p;
      // the FAR declines (the 20th neighbour more than kKnn3FarRings cells away: sparse parts of a real scan) of a SMALL
      // batch on a list of their own, served wave-cooperatively (s3d_knn_moments_far_kernel): a lone registration of two
      // of the reference's scans waits 0.55 ms less for the slowest lanes of the per-lane search (2.05 -> 1.50 ms); a large
      // batch is a matter of throughput, where 64 per-lane searches per wave win (96 pairs of those scans: normals 4.2
      // against 5.3 ms; the synthetic benchmark 7.35 against 7.8).  S3D_DBG_KNN_NO_FAR_COOP / _FORCE_FAR_COOP: A/B
      const bool small_batch = (long long)NL * max_n <= 2000000ll;
      const bool coop = (opts.debug_flags & S3D_DBG_KNN_NO_FAR_COOP) ? false
                        : (opts.debug_flags & S3D_DBG_KNN_FORCE_FAR_COOP) ? true : small_batch;
      // round 6: a LARGE batch puts every query whose 27 cells hold fewer than 20 points, or whose 20th neighbour lies beyond
      // the 5x5x5 proof, on the far list and serves it ring by ring (s3d_knn3_rings_kernel: the fast path's table + flat scan
      // carried on over rings 2 ... 6; on the reference's scans 93 % of all declines are of that kind, and the per-lane exact
      // search took 2.8 of the pre-pass's 4.4 ms for them).  S3D_DBG_KNN_NO_RINGS: the exact search for all of them.
      const bool rings = !coop && !(opts.debug_flags & S3D_DBG_KNN_NO_RINGS) && total_pts <= (size_t)0x1FFFFFF0;
      int* far_count = (coop || rings) ? fb_count + 2 : nullptr;
      const size_t redo_half = std::min<size_t>(std::max<size_t>(total_pts, 4), 0x1FFFFFF0);
      const int redo_cap = (int)(3 * redo_half);      // (far entries count down from the end of the buffer)
      const int iso_cap = (int)(2 * redo_half);       // (isolated points - the ring search's - from the end of the middle third)
      int* iso_count = fb_count + 3;
      // the far / near policy of a large batch on the DEVICE: a far list shorter than 1 % of the points is handed on to the
      // exact search as it is (its length in n_active[8], which the ICP stage zeroes and reuses afterwards)
      int* handed_on = rings ? fb_count + 4 : nullptr;
      const int rings_min = (opts.debug_flags & S3D_DBG_KNN_FORCE_RINGS) ? 0 : (int)std::min<size_t>(total_pts / 100, 0x7FFFFFFF);
      // (a small batch hands EVERY decline to the cooperative kernel - also the near ones, ties and table overflows, whose
      // per-lane search keeps a wave busy with one lane: far_all = 2; the ring search takes every "27 cells are not enough": 1)
      const int far_all = coop && small_batch ? 2 : (rings ? 1 : 0);
      s3d_knn3_moments_kernel<20><<<grid, kBlock, 0, st>>>(d_slots(), sorted(), cells(), mom, mom_plane, nb_head, d_list, NL, normals(), fb_count, fb_list, redo_count, redo_list, far_count, redo_cap, far_all);
      if (rings) {
        const int rblocks = (int)std::min<long long>(std::max<long long>((long long)NL * max_n / (16 * kBlock), 64), 1280);
        s3d_knn3_rings_kernel<20, kKnn3RingMax><<<rblocks, kBlock, 0, st>>>(d_slots(), sorted(), cells(), mom, mom_plane, normals(), fb_count, fb_list, redo_list, far_count, redo_cap, iso_count, iso_cap, handed_on, rings_min);
        // the isolated points among them (the 20th neighbour more than kKnn3RingMax rings away), a wave per query
        const int fblocks = (int)std::min<long long>(std::max<long long>((long long)NL * max_n / 512, 256), 8192);
        if (fused)
          s3d_knn_moments_far_kernel<20, true><<<fblocks, kWave, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, iso_count, redo_list, iso_cap, 3.0f);
        else
          s3d_knn_moments_far_kernel<20, false><<<fblocks, kWave, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, iso_count, redo_list, iso_cap, 3.0f);
      } else if (far_count) {
        const int fblocks = (int)std::min<long long>(std::max<long long>((long long)NL * max_n / 64, 256), 16384);
        if (fused)
          s3d_knn_moments_far_kernel<20, true><<<fblocks, kWave, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, far_count, redo_list, redo_cap, 3.0f);
        else
          s3d_knn_moments_far_kernel<20, false><<<fblocks, kWave, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, far_count, redo_list, redo_cap, 3.0f);
      }
      const bool thin = small_batch;   // a few clouds: the redo list's latency counts (see the kernel)
      if (far_all == 2) {
        // nothing on the near list
      } else if (thin && !fused)
        s3d_knn_moments_redo_kernel<20, true, true><<<1024, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, redo_count, redo_list, handed_on, redo_cap);
      else if (!fused)
        s3d_knn_moments_redo_kernel<20, true, false><<<1024, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, redo_count, redo_list, handed_on, redo_cap);
      else if (thin)
        s3d_knn_moments_redo_kernel<20, true, true, true><<<1024, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, redo_count, redo_list, handed_on, redo_cap);
      else
        s3d_knn_moments_redo_kernel<20, true, false, true><<<1024, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, normals(), fb_count, fb_list, redo_count, redo_list, handed_on, redo_cap);
    } else if (fused) {    // (the positions in `sorted` name the neighbours: BYPOS)
      if (k <= 8)
        s3d_knn_moments_kernel<8, false, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
      else if (k <= 16)
        s3d_knn_moments_kernel<16, false, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
      else if (k == 20)
        s3d_knn_moments_kernel<20, true, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
      else if (k < 20)
        s3d_knn_moments_kernel<20, false, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
      else
        s3d_knn_moments_kernel<32, false, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    } else if (k <= 8)
      s3d_knn_moments_kernel<8><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    else if (k <= 16)
      s3d_knn_moments_kernel<16><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    else if (k == 20)   // the reference default (correspondence_randomness = 20): list length known at compile time
      s3d_knn_moments_kernel<20, true><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    else if (k < 20)
      s3d_knn_moments_kernel<20><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    else
      s3d_knn_moments_kernel<32><<<grid, kBlock, 0, st>>>(d_slots(), filt(), sorted(), cells(), mom, mom_plane, k, nb_head, d_list, NL, normals(), fb_count, fb_list);
    s3d_normals_fallback_kernel<<<256, kBlock, 0, st>>>(fb_count, fb_list, mom, mom_plane, normals(), k);
    if (opts.debug_flags & S3D_DBG_PRINT_KNN) {   // dev aid: how many points took the eigen fallback / the exact-search redo
      int cnt[5];
      HIPCHK(hipMemcpyAsync(cnt, fb_count, sizeof cnt, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      std::fprintf(stderr, "[s3d] k-NN pre-pass: %zu points, eigen fallback %d, exact-search redo %d, far list (ring search / cooperative) %d, of those handed on to the exact search (a short list) %d, ring-search leftovers (cooperative) %d\n",
                   total_pts, cnt[0], cnt[1], cnt[2], cnt[4], cnt[3]);
    }
  }

  // per pair and outer iteration: the transformation_ that iteration's correspondence pass ran with (written by the
  // controller, read by the record-level re-validation of the settled passes and by the fitness pass)
  int hist_stride() const { return std::max(1, std::min(rp.max_iterations, 4096)); }
  WaveRec* wave_recs() { return (WaveRec*)ctx->wave_recs.p; }
  Mat4f* t_hist() { return (Mat4f*)ctx->t_hist.p; }
  int dbg_nn = 0;   // the S3D_DBG_NN_* bits of s3d_exec_options.debug_flags (set_params)
  // prof_slot >= 0: count searched / unseeded queries of this launch into the profile counters.
  // compact: the block-compacting mode of the kernel (see s3d_nn_search_kernel).
  NNArrays nn_arrays() {
    NNArrays A;
    A.sorted = sorted(); A.sorted3 = sorted3(); A.cell_start = cells(); A.normals = normals();
    A.corr_idx = (int*)ctx->corr_idx.p; A.corr_d2 = (float*)ctx->corr_d2.p; A.corr_lb = (float*)ctx->corr_lb.p;
    A.corr_q = (CorrVec*)ctx->corr_q.p; A.corr_n = (NormalRec*)ctx->corr_n.p;
    return A;
  }
  // it: outer iteration (0-based) of an ICP-loop pass, -1 otherwise.  The first pass has a kernel of its own
  // (s3d_nn_first_kernel); S3D_DBG_NN_NO_FIRST_KERNEL = the one kernel for every pass (A/B; the bits that switch the
  // re-validation or the seeds off imply it).
  void launch_nn(int mode, float max_d, int prof_slot = -1, bool compact = false, int it = -1) {
    hipStream_t st = ctx->stream;
    const int chunks = cdiv(std::max(max_n_t, 1), kBlock);
    const int pairs8 = P() >= 8 ? cdiv(P(), 8) * 8 : P();
    dim3 grid((unsigned)(pairs8 * chunks));
    NNArrays A = nn_arrays();
    int* pc = (prof_slot >= 0 && prof_slot < 64) ? (int*)ctx->n_active.p + 16 + 4 * prof_slot : nullptr;
    const bool family = mode == 0 && it >= 0 && !(dbg_nn & (262144 | 64));
    if (family && it == 0) {
      if (k4_pending) {
        // K4 is still running on the side stream: the pass leaves the copies of the neighbours' normals out (it needs the
        // grid only), and they are filled in once K4 is done - before the first accumulate launch reads them
        NNArrays A0 = A;
        A0.normals = nullptr;
        s3d_nn_first_kernel<<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A0, max_d, chunks, P(), dbg_nn, pc);
        join_k4();
        k_fill_corr_normals<<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, chunks, P());
        return;
      }
      s3d_nn_first_kernel<<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), dbg_nn, pc);
      return;
    }
    join_k4();        // (any other first pass reads the normals itself)
    // passes 2 and 3: the flat 27-cell scan + a worklist for what it declines (S3D_DBG_NN_NO_SCAN27 = off; the bits
    // that switch the re-validation off imply it)
    // (measured per 128 pairs: pass 2 1.47 -> 0.99 ms, pass 3 0.90 -> 0.80 ms with the compacting form; passes 4 and 5,
    // where 2 % and 0.1 % of the queries search, are slower this way: 0.39 -> 0.55, 0.15 -> 0.19 ms)
    // (round 5) a batch that is NOT served record-wise keeps the flat scan for passes 4-5 as well: on the reference's scans
    // 27 % / 7 % of the queries still search there, and the general kernel's seeded box search is ~8x slower per searched
    // query than the flat scan (one real pair: passes 4-5 0.10 + 0.11 ms); on a synthetic pair, where 2 % search, it costs
    // a few microseconds per pass (a block scan and the worklist launch)
    const int scan27_passes = settled_wanted() ? 2 : 4;
    if (family && it >= 1 && it <= scan27_passes && !(dbg_nn & (524288 | 128 | 2048)) && max_n < kKnn3MaxPoints) {
      int* wc = (int*)ctx->n_active.p + 8;              // eight counters, used in turn (zeroed by stage_icp / the drain)
      const bool compact27 = !(opts.debug_flags & S3D_DBG_SCAN27_NO_COMPACT);
      uint32_t* wl_pair = (uint32_t*)ctx->worklist.p;
      uint32_t* wl_index = wl_pair + std::max<size_t>(total_corr, 4);
      if (it == 1)
        s3d_nn_scan27_kernel<false, false><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), wc + (it & 7), wl_pair, wl_index, pc);
      else if (compact27)
        s3d_nn_scan27_kernel<true, true><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), wc + (it & 7), wl_pair, wl_index, pc);
      else
        s3d_nn_scan27_kernel<true, false><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), wc + (it & 7), wl_pair, wl_index, pc);
      s3d_nn_worklist_kernel<<<S3D_WL_BLOCKS, kWave, 0, st>>>(d_pairs(), d_slots(), A, max_d, dbg_nn, wc + (it & 7), wl_pair, wl_index,
                                                      wc + ((it + 1) & 7));
      return;
    }
    // passes 6 ...: record-level re-validation (s3d_nn_record_*_kernel); S3D_DBG_NN_NO_SETTLED = query by query
    if (settled_on() && mode == 0 && it >= kSettledFrom) {
      // two sets of list counters, used in turn (zeroed by stage_icp / by the consuming kernel of the pass before):
      // [set][0 .. 63] record lists, [set][64 .. 64 + 1023] search lists
      constexpr int kSet = kNNRecSublists + kNNSearchSublists;
      int* sc = (int*)ctx->rec_counts.p;
      int* cnt = sc + (it & 1) * kSet;
      int* cnt_next = sc + ((it + 1) & 1) * kSet;
      uint4* rl = (uint4*)ctx->rec_list.p;
      uint4* sl = (uint4*)ctx->search_list.p;
      const int scap = search_sub_cap();
      if (it == kSettledFrom) {                     // every record is evaluated (and gets its box): no test, no list
        s3d_nn_record_touch_kernel<false><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), wave_recs(),
                                                                   t_hist(), hist_stride(), nullptr, 1, 0, nullptr, nullptr,
                                                                   cnt + kNNRecSublists, scap, sl, pc);
      } else {
        const int nrec = cdiv(std::max(max_n_t, 1), kWave);
        const int rpt = rec_per_thread();
        const int bpp = cdiv(nrec, kBlock * rpt);
        const int nblocks = pairs8 * bpp;
        const int nsub = std::min(kNNRecSublists, nblocks);
        const int sub_cap = cdiv(nblocks, nsub) * kBlock * rpt;          // (every record of every block of a list failing)
        if (rpt == kNNRecPerThread)
          s3d_nn_record_test_kernel<kNNRecPerThread><<<(unsigned)nblocks, kBlock, 0, st>>>(
              d_pairs(), d_slots(), bpp, P(), wave_recs(), t_hist(), hist_stride(), cnt, nsub, sub_cap, rl, pc);
        else
          s3d_nn_record_test_kernel<1><<<(unsigned)nblocks, kBlock, 0, st>>>(
              d_pairs(), d_slots(), bpp, P(), wave_recs(), t_hist(), hist_stride(), cnt, nsub, sub_cap, rl, pc);
        // one wave per failing record while at most an eighth of the records fail (the waves without an entry leave
        // after one load); beyond that the waves loop.  (A quarter until round 5: a settled pass fails ~10 %, and the
        // dispatch of the empty waves was 5 of its 63 us; a sixteenth costs pass 5, which fails 18 %, more than it saves.)
        const long long waves = std::max<long long>(std::min<long long>((long long)P() * nrec, 7168), (long long)P() * nrec / 8);
        const int tblocks = std::max(cdiv(cdiv((int)waves, kBlock / kWave), nsub), 1) * nsub;   // a multiple of nsub blocks
        s3d_nn_record_touch_kernel<true><<<(unsigned)tblocks, kBlock, 0, st>>>(
            d_pairs(), d_slots(), A, max_d, chunks, P(), wave_recs(), t_hist(), hist_stride(), cnt, nsub, sub_cap, rl, cnt_next,
            cnt + kNNRecSublists, scap, sl, pc);
      }
      // the queries that failed their re-validation: one per wave and trip (a settled pass of 256 pairs lists ~4 000)
      s3d_nn_record_search_kernel<<<(unsigned)(search_parts() * kNNSearchSublists), kWave, 0, st>>>(d_pairs(), d_slots(), A, max_d, dbg_nn,
                                                                         cnt + kNNRecSublists, scap, sl,
                                                                         cnt_next + kNNRecSublists);
      return;
    }
    const int cmp = (compact && mode == 0 && !(dbg_nn & 65536)) ? 1 : 0;
    if (mode == 0)
      s3d_nn_search_kernel<0><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), dbg_nn, pc, cmp, nullptr,
                                                       nullptr, 0);
    else
      s3d_nn_search_kernel<1><<<grid, kBlock, 0, st>>>(d_pairs(), d_slots(), A, max_d, chunks, P(), dbg_nn, pc,
                                                       (dbg_nn & 65536) ? 0 : 1,   // (block compaction: 1.5 % of the queries search, scattered over all waves - 0.355 -> 0.29 ms at 256 pairs)
                                                       settled_used ? wave_recs() : nullptr, t_hist(), hist_stride());
  }
  // first outer iteration (0-based) that runs record-wise: the pass after the two flat-scan passes.  Pass 4 still
  // searches 2 % of its queries (580 000 at 256 pairs): the touch kernel streams all records and lists them, the search
  // kernel serves the lists 64 queries per wave - 0.50 ms against 0.76 for the block-compacting per-query kernel (with
  // one-query-per-wave searches only, before the throughput mode existed: 2.1 ms).  Passes 5 / 6: 0.247 / 0.098 ms.
  static constexpr int kSettledFrom = 3;
  // records per thread of the test kernel: four for a large batch (few blocks, one list append each), one for a small one
  int rec_per_thread() const { return (long long)P() * cdiv(std::max(max_n_t, 1), kWave) >= 65536 ? kNNRecPerThread : 1; }
  // capacity of one of the kNNSearchSublists search lists: every query of the records that map to it
  int search_sub_cap() const {
    const long long recs = (long long)(std::max<size_t>(total_corr, 4) / kWave + 1);
    return (int)std::min<long long>((recs / kNNSearchSublists + 1) * kWave, 0x7FFFFFF0);
  }
  // waves per search list.  Always 16 (round 5): a surplus wave leaves after one load, and on real scans, whose settled
  // passes list ten times the searches of the synthetic ones, a short list shared by 16 waves is served in a third of the
  // time (until then: one wave per ~3 000 queries of the batch, 1 ... 16 - 1 for 96 pairs of the reference's scans, 9
  // for the benchmark)
  int search_parts() const { return 16; }
  int rec_blocks() const {
    const int nrec = cdiv(std::max(max_n_t, 1), kWave), rpt = rec_per_thread();
    const int pairs8 = P() >= 8 ? cdiv(P(), 8) * 8 : std::max(P(), 1);
    return pairs8 * cdiv(nrec, kBlock * rpt);
  }
  size_t rec_list_entries() const {
    const int nrec = cdiv(std::max(max_n_t, 1), kWave), rpt = rec_per_thread();
    const int pairs8 = P() >= 8 ? cdiv(P(), 8) * 8 : std::max(P(), 1);
    const int nblocks = pairs8 * cdiv(nrec, kBlock * rpt);
    const int nsub = std::min(kNNRecSublists, nblocks);
    return (size_t)nsub * (size_t)(cdiv(nblocks, nsub) * kBlock * rpt);
  }
  bool settled_used = false;               // stage_icp: this batch's records / transform history are initialised
  // (a small batch keeps the per-query kernel: a record-wise pass is two launches whose second one ends with the few
  // searches of the pass, which the per-query kernel hides among its 1 500 waves per pair.  Measured, settled pass per
  // query / per record: 1 pair 11 / 16 us, 32 pairs equal, 128 pairs 0.100 / 0.055 ms, 256 pairs 0.196 / 0.089 ms)
  bool settled_wanted() const {
    return !(dbg_nn & (S3D_DBG_NN_NO_SETTLED | 262144 | 64)) &&
           ((long long)P() * cdiv(std::max(max_n_t, 1), kWave) >= 65536 || (opts.debug_flags & S3D_DBG_NN_FORCE_SETTLED));
  }
  bool settled_on() const { return settled_used && settled_wanted(); }

  // one outer iteration: correspondences (K5), accumulate (K6), controller (K7)
  void launch_iteration(int it, float max_d, int prof_slot) {
    // (the block-compacting variant of the general kernel served passes 3-5 until round 4, when those passes of a large batch
    // got kernels of their own; for the small batches that still come here it is neutral on synthetic pairs - 1, 16, 40
    // pairs: 1.277 / 2.92 / 5.57 ms with, 1.264 / 2.97 / 5.56 without - and costs one registration of two of the reference's
    // scans 40 us: packed into a few waves, its ~600 queries without a neighbour walk their 50-row balls lane by lane,
    // while spread one or two to a wave they are served cooperatively.  Off for the ICP passes; the fitness pass keeps it.)
    launch_nn(0, max_d, prof_slot, false, it);   // (the counters cost two atomics per searching wave)
    launch_iteration_after_nn();
  }
  void launch_iteration_after_nn() {
    hipStream_t st = ctx->stream;
    double* part = (double*)ctx->partials.p;
    if (rp.algorithm)
      s3d_gicp_accumulate_kernel<<<dim3(accum_blocks, P()), kBlock, 0, st>>>(
          d_pairs(), d_slots(), sorted3(), normals(), (CorrVec*)ctx->corr_q.p, (NormalRec*)ctx->corr_n.p, part, rp);
    if (!rp.algorithm)
      s3d_p2plane_accumulate_kernel<<<dim3(accum_blocks, P()), kBlock, 0, st>>>(
          d_pairs(), d_slots(), sorted3(), (CorrVec*)ctx->corr_q.p, (NormalRec*)ctx->corr_n.p, part, rp);
    s3d_icp_control_kernel<<<P(), kCtrlThreads, 0, st>>>(d_pairs(), part, rp, (int*)ctx->n_active.p, t_hist(), hist_stride(),
                                                         icp_host_word, icp_launch++, icp_tag);
  }
  // the progress word of this batch's ICP loop (stage_icp; null: nobody listens), the index of the next controller launch
  int* icp_host_word = nullptr;
  int icp_launch = 0, icp_tag = 0;

  // K5-K7 loop.  Converged pairs stop on the device at once (every kernel of an iteration leaves when its pair is
  // inactive); what the host has to learn is when to stop LAUNCHING.  Default (check_interval = 0): the controller kernel
  // reports through two words of pinned host memory - the index of the newest iteration whose controller has started, and
  // a tag once the last active pair has stopped - and the host launches iteration `it` as soon as iteration it - kIcpAhead
  // is reported, never waiting for the stream: no copy, no drained queue (the stream wait + copy of a poll left the device
  // idle for ~28 us, twice in a registration of two of the reference's scans), at most kIcpAhead empty iterations after
  // the last pair has stopped.  check_interval = N > 0: the host copies the active-pair counter every N iterations and
  // waits for it (the form of rounds 1-4).  Forced iterations: everything is launched at once.
#ifndef S3D_ICP_AHEAD
#define S3D_ICP_AHEAD 2
#endif
  static constexpr int kIcpAhead = S3D_ICP_AHEAD;
  void stage_icp() {
    hipStream_t st = ctx->stream;
    if (P() == 0) return;
    int* d_active = (int*)ctx->n_active.p;
    {   // pair state, "no radius hint" / "nothing known" for every correspondence, the 64-query records "never evaluated",
        // the list counters: one launch (k_icp_reset)
      const size_t ncorr = std::max<size_t>(total_corr, 4);
      const size_t nrec_words = (ncorr / kWave + 1) * (sizeof(WaveRec) / 4);
      const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>(cdiv((int)std::min<size_t>(ncorr, 0x7FFFFFFF), kBlock * 4), 1), 8192);
      k_icp_reset<<<blocks, kBlock, 0, st>>>(d_pairs(), P(), d_active, (uint32_t*)ctx->corr_d2.p, (uint32_t*)ctx->corr_lb.p, ncorr,
                                              (uint32_t*)ctx->wave_recs.p, nrec_words, (int*)ctx->rec_counts.p,
                                              2 * (kNNRecSublists + kNNSearchSublists));
    }

    settled_used = true;
    const float max_d = (float)(rp.max_corr * 1.0001);
    const bool prof = opts.profile != 0;
    if (prof) HIPCHK(hipMemsetAsync((int*)ctx->n_active.p + 16, 0, 4 * 64 * sizeof(int), st));
    auto launch_one = [&](int it) {
      if (prof) {
        if ((int)ctx->nn_ev.size() < 2 * (it + 1)) {
          hipEvent_t a, b;
          HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
          ctx->nn_ev.push_back(a); ctx->nn_ev.push_back(b);
        }
        HIPCHK(hipEventRecord(ctx->nn_ev[2 * it], st));
        launch_nn(0, max_d, opts.profile >= 2 ? it : -1, false, it);
        HIPCHK(hipEventRecord(ctx->nn_ev[2 * it + 1], st));
        launch_iteration_after_nn();
      } else {
        launch_iteration(it, max_d, -1);
      }
      ctx->prof.nn_launches = it + 1;
    };
    icp_host_word = nullptr; icp_launch = 0; icp_tag = 0;
    ctx->prof.nn_launches = 0;
    if (!rp.force_iterations && opts.check_interval <= 0 && ctx->h_active_dev) {
      // (every earlier call of this context has waited for its stream: nothing on the device still writes these words)
      volatile int* hw = ctx->h_active + 4;
      if (++ctx->icp_tag_counter == 0u) ++ctx->icp_tag_counter;
      icp_tag = (int)ctx->icp_tag_counter;
      hw[0] = -1; hw[1] = 0;
      __atomic_thread_fence(__ATOMIC_SEQ_CST);
      icp_host_word = ctx->h_active_dev + 4;
      try {
      for (int it = 0; it < rp.max_iterations; ++it) {
        bool done = false;
        for (unsigned spins = 1;; ++spins) {
          if (__atomic_load_n((const int*)&hw[1], __ATOMIC_ACQUIRE) == icp_tag) { done = true; break; }
          if (__atomic_load_n((const int*)&hw[0], __ATOMIC_RELAXED) >= it - kIcpAhead) break;
          if ((spins & 1023u) == 0) {
            // a failed launch or a drained stream must not leave the host spinning
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {
              done = __atomic_load_n((const int*)&hw[1], __ATOMIC_ACQUIRE) == icp_tag;
              break;
            }
            if (q != hipErrorNotReady) HIPCHK(q);
          }
          cpu_relax();
        }
        if (done) break;
        launch_one(it);
      }
      } catch (...) {
        // controller launches still in flight would write this call's words into the next call's freshly reset ones
        // (harmless for termination - the tag - but the throttle could run further ahead than kIcpAhead)
        (void)hipStreamSynchronize(st);
        icp_host_word = nullptr;
        throw;
      }
      icp_host_word = nullptr;
      return;
    }
    // segments of iterations between two polls of the active-pair counter (all of them when the count is forced)
    // (check_interval <= 0 here: the progress words could not be mapped - the poll of rounds 1-4, every 4 iterations)
    const int seg = rp.force_iterations ? std::max(rp.max_iterations, 1) : (opts.check_interval > 0 ? opts.check_interval : 4);
    for (int it0 = 0; it0 < rp.max_iterations; it0 += seg) {
      const int it1 = std::min(it0 + seg, rp.max_iterations);
      for (int it = it0; it < it1; ++it) launch_one(it);
      if (!rp.force_iterations && it1 < rp.max_iterations) {
        HIPCHK(hipMemcpyAsync(ctx->h_active, d_active, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (*ctx->h_active <= 0) break;
      }
    }
  }

  // A9: final_transformation_, one more NN pass, masked mean
  void stage_fitness() {
    hipStream_t st = ctx->stream;
    join_k4();         // (no correspondence pass has run - no pairs, no iterations: the normals are awaited here)
    if (P() == 0) return;
    k_pair_finalize<<<cdiv(P(), 64), 64, 0, st>>>(d_pairs(), P());
    launch_nn(1, (float)(std::sqrt(std::max(rp.fit_range, 0.0)) * 1.0001));
    double* part = (double*)ctx->partials.p;
    s3d_fitness_partial_kernel<<<dim3(accum_blocks, P()), kBlock, 0, st>>>(d_pairs(), d_slots(), (float*)ctx->corr_d2.p,
                                                                           part, rp);
    k_fitness_final<<<cdiv(P(), 64), 64, 0, st>>>(d_pairs(), part, P());
  }

  void download() {
    hipStream_t st = ctx->stream;
    join_k4();
    const size_t bs = slots_bytes(), bp = pairs_bytes();
    char* stage = ctx->stage_host(records_bytes());   // (the uploads of this call have completed by now: stream order)
    // slots, pairs and the sort's error word in one copy (the one-sweep sort's look-back gives up after ~4 M polls instead
    // of hanging the device: that must not pass silently)
    int* sort_err = (int*)(stage + bs + bp);
    HIPCHK(hipMemcpyAsync(stage, ctx->slots.p, bs + bp + 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!sort_used) *sort_err = 0;
    HIPCHK(hipGetLastError());
    if (*sort_err) throw HipError{hipErrorLaunchFailure, "radix sort: a tile waited for its predecessor beyond the poll limit", __LINE__};
    if (bs) std::memcpy(h_slots.data(), stage, bs);
    if (bp) std::memcpy(h_pairs.data(), stage + bs, bp);
  }

  void run_all() {
    hipStream_t st = ctx->stream;
    const bool prof = opts.profile != 0;
    auto mark = [&](int i) {
      if (!prof) return;
      if (!ctx->ev[i]) HIPCHK(hipEventCreate(&ctx->ev[i]));
      HIPCHK(hipEventRecord(ctx->ev[i], st));
    };
    ctx->prof = s3d_profile{};
    struct SideGuard {      // a call that unwinds must not leave K4 running into the next call's pre-pass
      Batch* b;
      ~SideGuard() { if (b->k4_pending && b->ctx->side_stream) (void)hipStreamSynchronize(b->ctx->side_stream); b->k4_pending = false; }
    } side_guard{this};
    for (int attempt = 0; attempt < 2; ++attempt) {
      const bool resume = phase == 2 && attempt == 0;     // (pre-pass and k-NN of this attempt ran in the phase-1 call)
      if (!resume) {
      mark(0);
      if (fused) stage_prepass_fused(); else stage_voxel();
      mark(1);
      if (!fused) stage_grid();
      mark(2);
      // (round 6) a SMALL registration batch - the reference's own call is ONE pair - does not fill the chip: its k-NN
      // pre-pass runs on a second stream next to the first correspondence pass (which needs the grid, not the normals;
      // launch_nn fills the copies of the neighbours' normals in afterwards): one registration of two of the reference's
      // scans waits ~60 us less.  Not with the K4 debug switches (their lists share counters with the ICP stage), not
      // under the profile (its stage split is one stream's), not beyond ~two clouds (400 k raw points: the mapper's one
      // scan against eight already keeps the chip busy - 1.50 -> 1.52 ms with the pre-pass beside the pass).
      overlap_k4 = false;
      {
        const unsigned k4_dbg = S3D_DBG_KNN_NO_FAR_COOP | S3D_DBG_KNN_FORCE_FAR_COOP | S3D_DBG_KNN_NO_RINGS |
                                S3D_DBG_KNN_FORCE_RINGS | S3D_DBG_KNN_EXACT64 | S3D_DBG_PRINT_KNN | S3D_DBG_NO_K4_OVERLAP;
        const bool first_kernel = !(dbg_nn & (262144 | 64));
        if (phase == 0 && registration_batch && !prof && P() > 0 && rp.max_iterations >= 1 && first_kernel && !(opts.debug_flags & k4_dbg) &&
            !knn_slots.empty() && (long long)knn_slots.size() * max_n <= 400000ll && ctx->ensure_side_stream()) {
          HIPCHK(hipEventRecord(ctx->side_ev[0], st));
          HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_ev[0], 0));
          overlap_k4 = true;
        }
      }
      stage_normals();
      if (overlap_k4) {
        HIPCHK(hipEventRecord(ctx->side_ev[1], ctx->side_stream));
        k4_pending = true;
        overlap_k4 = false;
      }
      mark(3);
      }
      if (phase == 1) return;
      stage_icp();
      mark(4);
      stage_fitness();
      mark(5);
      download();
      if (!fused) break;
      bool served = true;
      for (int j = 0; j < Cu; ++j) served = served && h_slots[(size_t)j].fz.ok > 0;
      if (served) break;
      // a slot the fused pre-pass cannot serve (PCL's own index overflow, a mixed key beyond 32 bits, a centroid outside
      // its cell by more than the searches' margin): the whole batch again on the two-sort path, from the records as
      // allocate() uploaded them
      fused = false;
      ++ctx->fused_reruns;
      if (use_cache || ctx->fused_unservable.size() < 4096)     // (bounded: handles of the host-buffer entry points die with the call)
        for (int j = 0; j < Cu; ++j)
          if (h_slots[(size_t)j].fz.ok <= 0) ctx->fused_unservable.insert({slot_clouds[(size_t)j]->uid, leaf_bits()});
      h_slots = h_slots0;
      h_pairs = h_pairs0;
      if (Cu < C()) {   // clouds restored from fused-layout cache entries: computed again like the others
        Cu = C();
        slot_entry.assign((size_t)C(), nullptr);
        slot_has_normals.assign((size_t)C(), 0);
        assign_want_normals();
      }
      upload_records();
    }
    store_to_cache(true);
    if (prof) {
      float ms = 0;
      auto el = [&](int a, int b) { HIPCHK(hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b])); return (double)ms; };
      ctx->prof.voxel_ms = el(0, 1); ctx->prof.grid_ms = el(1, 2); ctx->prof.normals_ms = el(2, 3);
      ctx->prof.icp_ms = el(3, 4); ctx->prof.fitness_ms = el(4, 5); ctx->prof.total_ms = el(0, 5);
      long long nq = 0, nt = 0;
      for (const PairDev& P : h_pairs) { nq += h_slots[P.slot_t].n; nt += h_slots[P.slot_s].n; }
      for (int i = 0; i < ctx->prof.nn_launches; ++i) {
        HIPCHK(hipEventElapsedTime(&ms, ctx->nn_ev[2 * i], ctx->nn_ev[2 * i + 1]));
        ctx->prof.nn_ms += ms;
        if (i < 64) ctx->prof.nn_launch_ms[i] = ms;
      }
      int counts[256];
      copy_to_host(ctx, counts, (int*)ctx->n_active.p + 16, sizeof counts);
      for (int i = 0; i < 64; ++i) {
        ctx->prof.nn_searched[i] = counts[4 * i]; ctx->prof.nn_unseeded[i] = counts[4 * i + 1];
        ctx->prof.nn_records[i] = counts[4 * i + 2]; ctx->prof.nn_records_searched[i] = counts[4 * i + 3];
      }
      ctx->prof.nn_queries = nq * ctx->prof.nn_launches;
      ctx->prof.nn_targets = nt * ctx->prof.nn_launches;
    }
  }

  // gates of doICP / align() applied on the host to the downloaded pair state
  int finish_pair(int p, const s3d_reg_params* params, const double guess[16], double result[16],
                  s3d_align_info* info) const {
    const PairDev& P = h_pairs[p];
    const SlotDev& Ss = h_slots[P.slot_s];
    const SlotDev& St = h_slots[P.slot_t];
    if (info) {
      info->n_source_filtered = Ss.n; info->n_target_filtered = St.n;
      info->iterations = P.iterations; info->converged = P.converged; info->correspondences = P.correspondences;
      info->fitness = P.fitness; info->inner_iterations = P.inner_total; info->evaluations = P.evals_total;
    }
    for (int i = 0; i < 16; ++i) result[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (St.n < 100 || Ss.n < 100) return S3D_STATUS_TOO_FEW_POINTS;  // PointCloudSensor.cpp:134-135
    if (params->correspondence_randomness > Ss.n || params->correspondence_randomness > St.n ||
        params->correspondence_randomness > 64 || params->correspondence_randomness < 1)
      return S3D_STATUS_INVALID_ARGUMENT;
    // :80-81 Transform(Eigen::Isometry3f(getFinalTransformation())): widen, no re-orthonormalisation
    for (int i = 0; i < 16; ++i) result[i] = (double)P.final_T.m[i];
    result[3] = result[7] = result[11] = 0.0;
    result[15] = 1.0;
    if (!P.converged) return S3D_STATUS_NOT_CONVERGED;                                   // :74
    if (P.fitness > params->max_fitness_score) return S3D_STATUS_FITNESS_EXCEEDED;       // :74
    double ginv[16], delta[16];
    mat4d_inverse_isometry(guess, ginv);
    mat4d_mul(ginv, result, delta);                                                      // :167
    const double tn = std::sqrt(HM(delta, 0, 3) * HM(delta, 0, 3) + HM(delta, 1, 3) * HM(delta, 1, 3) +
                                HM(delta, 2, 3) * HM(delta, 2, 3));
    if (tn > params->max_translation || rotation_angle(delta) > params->max_rotation)    // :169
      return S3D_STATUS_TOO_FAR_FROM_GUESS;
    return S3D_STATUS_OK;
  }
};

static void copy_to_host(s3d_context* ctx, void* dst, const void* src, size_t bytes) {
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
}

int check_algorithm(const s3d_reg_params* p, const s3d_exec_options* o) {
  switch (p->registration_algorithm) {  // PointCloudSensor.cpp:139-165
    case S3D_ALG_ICP: case S3D_ALG_GICP: case S3D_ALG_NDT: return S3D_STATUS_OK;   // (NDT: host-driven, see align_ndt())
    case S3D_ALG_GICP_OMP: case S3D_ALG_NDT_OMP:      // :149-162: pclomp build (served by the same code) or not
      return (o && o->omp_unavailable) ? S3D_STATUS_OMP_UNAVAILABLE : S3D_STATUS_OK;
    default: return S3D_STATUS_UNKNOWN_ALGORITHM;
  }
}

int upload_cloud(s3d_context* ctx, const float* xyz, int n, int stride, s3d_cloud* c) {
  c->n = n; c->owned = true; c->d = nullptr; c->uid = g_cloud_uid++;
  HIPCHK(hipMalloc((void**)&c->d, sizeof(float4) * (size_t)std::max(n, 1)));
  if (n > 0) {
    if (stride == 4) {
      HIPCHK(hipMemcpyAsync(c->d, xyz, sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    } else if (stride > 6) {   // wide records: pack on the host rather than ship the unused fields
      std::vector<float4> tmp((size_t)n);
      for (int i = 0; i < n; ++i) {
        const float* p = xyz + (size_t)i * stride;
        tmp[i] = make_float4(p[0], p[1], p[2], 1.f);
      }
      HIPCHK(hipMemcpyAsync(c->d, tmp.data(), sizeof(float4) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    } else {
      // raw floats to a staging buffer of the context, float4 expansion on the device
      const size_t bytes = sizeof(float) * ((size_t)(n - 1) * stride + 3);
      if (ctx->staging.cap < bytes) {
        if (ctx->staging.p) HIPCHK(hipFree(ctx->staging.p));
        ctx->staging.p = nullptr; ctx->staging.cap = 0;
        HIPCHK(hipMalloc(&ctx->staging.p, bytes + bytes / 4));
        ctx->staging.cap = bytes + bytes / 4;
      }
      HIPCHK(hipMemcpyAsync(ctx->staging.p, xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
      k_expand_points<<<cdiv(n, kBlock), kBlock, 0, ctx->stream>>>((const float*)ctx->staging.p, n, stride, c->d);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  return S3D_STATUS_OK;
}
void free_cloud(s3d_cloud* c) {
  if (c->owned && c->d) (void)hipFree(c->d);
  c->d = nullptr;
  c->block.reset();       // (a bulk hand-over's allocation goes with its last cloud)
}

// Bulk hand-over (s3d_cloud_upload_many).  One s3d_cloud_upload is a hipMalloc, a copy from pageable memory (which the
// runtime stages in small pieces: ~12 GB/s) and a stream wait: ~100 us for a 100 k-point scan, 50 ms for the 512 clouds
// of a sweep.  Here all clouds share ONE device allocation, and `lanes` host threads - each with a stream and two pinned
// slots of its own - copy the callers' arrays into pinned memory (wide records packed to xyz on the way) while the
// expansion kernels of the previous slots read that memory over PCIe and write the float4 layout straight into HBM: no
// staging copy on the device, no per-cloud allocation, no per-cloud wait.
void upload_many(s3d_context* ctx, int n_clouds, const float* const* xyz, const int* n, int stride, s3d_cloud** out) {
  size_t total = 0, most = 0;
  std::vector<size_t> off((size_t)n_clouds);
  for (int i = 0; i < n_clouds; ++i) {
    off[(size_t)i] = total;
    total += ((size_t)std::max(n[i], 1) + 63) & ~(size_t)63;      // (clouds start on 1 KiB boundaries)
    most = std::max(most, (size_t)std::max(n[i], 0));
  }
  auto block = std::make_shared<S3dDevBlock>();
  HIPCHK(hipMalloc(&block->p, sizeof(float4) * std::max<size_t>(total, 64)));
  const int hs = stride > 6 ? 3 : stride;                            // floats per point in the pinned slot
  const size_t slot_bytes = std::max<size_t>(sizeof(float) * most * (size_t)hs, 256);
  // up to eight worker threads, fewer for huge clouds: the pinned slots (two per thread) stay below 512 MiB
  const int lanes = (int)std::max<size_t>(1, std::min<size_t>({(size_t)n_clouds, (size_t)(ctx->upload_threads_cap > 0 ? std::min(ctx->upload_threads_cap, 8) : 8),
                                                                 (size_t)std::max(1u, std::thread::hardware_concurrency()),
                                                                 ((size_t)512 << 20) / (2 * (slot_bytes + slot_bytes / 4))}));
  if ((int)ctx->upload_lanes.size() < lanes) ctx->upload_lanes.resize((size_t)lanes);
  for (int t = 0; t < lanes; ++t) {
    s3d_context::UploadLane& L = ctx->upload_lanes[(size_t)t];
    if (!L.st) HIPCHK(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) if (!L.ev[k]) HIPCHK(hipEventCreateWithFlags(&L.ev[k], hipEventDisableTiming));
    if (L.cap < slot_bytes) {
      HIPCHK(hipStreamSynchronize(L.st));
      L.cap = 0;     // (a failed allocation below leaves the lane without slots: the next call must allocate again)
      for (int k = 0; k < 2; ++k) {
        if (L.pinned[k]) HIPCHK(hipHostFree(L.pinned[k]));
        L.pinned[k] = nullptr;
        HIPCHK(hipHostMalloc((void**)&L.pinned[k], slot_bytes + slot_bytes / 4));
      }
      L.cap = slot_bytes + slot_bytes / 4;
    }
  }
  std::atomic<int> next{0};
  std::vector<std::exception_ptr> errs;   // (any exception: one that left a std::thread's function would be std::terminate)
  std::mutex errs_mtx;
  auto work = [&](int t) {
    try {
      HIPCHK(hipSetDevice(ctx->device));
      s3d_context::UploadLane& L = ctx->upload_lanes[(size_t)t];
      bool used[2] = {false, false};
      int k = 0;
      for (int i = next.fetch_add(1); i < n_clouds; i = next.fetch_add(1)) {
        const int ni = n[i];
        if (ni <= 0) continue;
        if (used[k]) HIPCHK(hipEventSynchronize(L.ev[k]));          // the kernel that read this slot has finished
        float* dst = (float*)L.pinned[k];
        if (hs == stride) {
          std::memcpy(dst, xyz[i], sizeof(float) * ((size_t)(ni - 1) * stride + 3));
        } else {
          const float* src = xyz[i];
          for (int j = 0; j < ni; ++j) { dst[3 * j] = src[(size_t)j * stride]; dst[3 * j + 1] = src[(size_t)j * stride + 1]; dst[3 * j + 2] = src[(size_t)j * stride + 2]; }
        }
        k_expand_points<<<cdiv(ni, kBlock), kBlock, 0, L.st>>>(dst, ni, hs, (float4*)block->p + off[(size_t)i]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(L.ev[k], L.st));
        used[k] = true;
        k ^= 1;
      }
      HIPCHK(hipStreamSynchronize(L.st));
    } catch (...) {
      std::lock_guard<std::mutex> lock(errs_mtx);
      errs.push_back(std::current_exception());
    }
  };
  std::vector<std::thread> th;
  bool spawn_failed = false;
  try {
    th.reserve((size_t)lanes);
    for (int t = 1; t < lanes; ++t) th.emplace_back(work, t);
  } catch (const std::exception&) {   // std::system_error / bad_alloc: the lanes that did start (and lane 0) take all clouds
    spawn_failed = true;
  }
  work(0);
  for (std::thread& x : th) x.join();
  (void)spawn_failed;                 // (not an error: the clouds are dealt by the shared counter `next`)
  if (!errs.empty()) {
    // (kernels of the lanes that did not fail may still be writing into the block the caller is about to drop)
    for (int t = 0; t < lanes; ++t) (void)hipStreamSynchronize(ctx->upload_lanes[(size_t)t].st);
    std::rethrow_exception(errs[0]);
  }
  for (int i = 0; i < n_clouds; ++i) {
    s3d_cloud* c = out[i];
    c->n = std::max(n[i], 0); c->owned = false; c->uid = g_cloud_uid++;
    c->d = (float4*)block->p + off[(size_t)i];
    c->block = block;
  }
}

// ---- device-resident helpers shared by the map / patch entry points --------------------------------

void alloc_cloud(s3d_cloud* c, int n) {
  c->n = n; c->owned = true; c->d = nullptr; c->uid = g_cloud_uid++;
  HIPCHK(hipMalloc((void**)&c->d, sizeof(float4) * (size_t)std::max(n, 1)));
}

void rowmajor3x4(const double colmajor[16], double out[12]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) out[r * 4 + c] = colmajor[c * 4 + r];
}

// B1: transform + concatenate into a new owned cloud
void accumulate_dev(s3d_context* ctx, int n_clouds, s3d_cloud* const* clouds, const double* poses, const double* frame,
                    s3d_cloud* out) {
  size_t total = 0;
  std::vector<XformJob> jobs((size_t)std::max(n_clouds, 1));
  int max_n = 0;
  for (int i = 0; i < n_clouds; ++i) {
    XformJob& J = jobs[i];
    J.src = clouds[i]->d; J.n = clouds[i]->n; J.out_off = (int)total;
    rowmajor3x4(poses + (size_t)i * 16, J.T);
    total += (size_t)clouds[i]->n;
    max_n = std::max(max_n, clouds[i]->n);
  }
  if (total > (size_t)0x7FFFFFF0) throw HipError{hipErrorInvalidValue, "accumulated cloud too large for 32-bit offsets", __LINE__};
  alloc_cloud(out, (int)total);
  if (total == 0) return;
  Xf3x4d fr{};
  if (frame) {
    double inv[16];
    mat4d_inverse_isometry(frame, inv);   // pose.inverse() (PointCloudSensor.cpp:262)
    rowmajor3x4(inv, fr.m);
    fr.enabled = 1;
  }
  struct DevJobs {   // freed on every exit path
    XformJob* p = nullptr;
    ~DevJobs() { if (p) (void)hipFree(p); }
  } d_jobs;
  HIPCHK(hipMalloc((void**)&d_jobs.p, sizeof(XformJob) * jobs.size()));
  HIPCHK(hipMemcpyAsync(d_jobs.p, jobs.data(), sizeof(XformJob) * jobs.size(), hipMemcpyHostToDevice, ctx->stream));
  // enough blocks per cloud to fill the chip even for a two-cloud patch; grid-stride inside
  const int bx = std::max(1, std::min(cdiv(max_n, kBlock), std::max(8, cdiv(4096, n_clouds))));
  s3d_transform_concat_kernel<<<dim3(bx, n_clouds), kBlock, 0, ctx->stream>>>(d_jobs.p, out->d, fr);
  HIPCHK(hipStreamSynchronize(ctx->stream));   // jobs[] (pageable) must outlive the copy, d_jobs the kernel
  HIPCHK(hipGetLastError());
}

void copy_cloud_dev(s3d_context* ctx, const s3d_cloud* in, s3d_cloud* out) {
  alloc_cloud(out, in->n);
  if (in->n > 0)
    HIPCHK(hipMemcpyAsync(out->d, in->d, sizeof(float4) * (size_t)in->n, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
}

struct StageTimer {   // optional HIP-event stopwatch on the context's stream
  s3d_context* ctx; hipEvent_t a = nullptr, b = nullptr;
  explicit StageTimer(s3d_context* c) : ctx(c) { HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); }
  ~StageTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  void start() { HIPCHK(hipEventRecord(a, ctx->stream)); }
  double stop() {
    HIPCHK(hipEventRecord(b, ctx->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    return (double)ms;
  }
};

// B2: pcl::RadiusOutlierRemoval on the search grid
void remove_outliers_dev(s3d_context* ctx, const s3d_cloud* in, double radius, unsigned min_neighbors, s3d_cloud* out,
                         s3d_map_profile* mp) {
  if (in->n == 0 || !(radius > 0) || min_neighbors == 0) {   // PointCloudSensor.cpp:214: returns `in`
    copy_cloud_dev(ctx, in, out);
    return;
  }
  StageTimer tm(ctx);
  Batch b;
  b.ctx = ctx;
  b.opts.grid_cells_per_point = 2;
  b.opts.check_interval = 4;
  b.cell_cap_max = 1ll << 30;
  b.rp.leaf = 0.f;                 // search the cloud as given
  b.rp.h0 = (float)radius;         // the ball of one query then spans at most 3 cells per axis
  std::map<const s3d_cloud*, int> index;
  b.add_slot(in, index);
  b.allocate(false);
  hipStream_t st = ctx->stream;
  tm.start();
  b.stage_voxel();
  b.stage_grid();
  if (mp) mp->grid_ms = tm.stop();
  // PCL compares double(r*r) < double(d2f): the largest float not above r*r gives the same verdicts in float
  const double r2 = radius * radius;
  float r2f = (float)r2;
  if ((double)r2f > r2) r2f = std::nextafterf(r2f, 0.f);
  const float reach = (float)(radius * 1.00001) + 1e-30f;
  const long long need_ll = (long long)min_neighbors + 1;   // the query itself is one of the k neighbours
  const int need = (int)std::min<long long>(need_ll, 0x7FFFFFFF);
  uint32_t* flags = b.vA();   // free after the grid build unless the wide sort left its result there
  if (b.max_cell_cap > (1ll << 24)) flags = b.vB();
  tm.start();
  s3d_radius_count_kernel<<<dim3(b.nb_head, 1), kBlock, 0, st>>>(b.d_slots(), b.sorted(), b.cells(), flags, reach, r2f, need);
  if (mp) mp->count_ms = tm.stop();
  alloc_cloud(out, in->n);
  uint32_t* bc = (uint32_t*)ctx->blockcnt.p;
  tm.start();
  k_flags_count<<<dim3(b.nb_head, 1), kBlock, 0, st>>>(b.d_slots(), flags, bc, b.nb_head);
  k_heads_scan<<<1, kBlock, 0, st>>>(b.d_slots(), bc, b.nb_head);   // slot.n = number kept
  k_flags_compact<<<dim3(b.nb_head, 1), kBlock, 0, st>>>(b.d_slots(), flags, bc, b.filt(), out->d, b.nb_head);
  if (mp) mp->compact_ms = tm.stop();
  b.download();
  out->n = b.h_slots[0].n;
}

// A3 on a device-resident cloud
void voxel_dev(s3d_context* ctx, const s3d_cloud* in, double leaf_size, s3d_cloud* out, s3d_map_profile* mp) {
  if (in->n == 0 || !(leaf_size > 0)) {
    copy_cloud_dev(ctx, in, out);
    return;
  }
  StageTimer tm(ctx);
  Batch b;
  b.ctx = ctx;
  b.opts.grid_cells_per_point = 2;
  b.opts.check_interval = 4;
  b.rp.leaf = (float)leaf_size;   // PointCloudSensor.cpp:196 setLeafSize(double -> float)
  b.rp.h0 = 0.25f;
  std::map<const s3d_cloud*, int> index;
  b.add_slot(in, index);
  b.allocate(false);
  tm.start();
  b.stage_voxel();
  if (mp) mp->voxel_ms = tm.stop();
  b.download();
  const int m = b.h_slots[0].n;
  alloc_cloud(out, m);
  if (m > 0)
    HIPCHK(hipMemcpyAsync(out->d, b.filt() + b.h_slots[0].off, sizeof(float4) * (size_t)m, hipMemcpyDeviceToDevice,
                          ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
}

void download_packed(s3d_context* ctx, const s3d_cloud* c, float* xyz, int stride) {
  const int n = c->n;
  if (n <= 0) return;
  std::vector<float4> tmp((size_t)n);
  HIPCHK(hipMemcpyAsync(tmp.data(), c->d, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i) {
    float* o = xyz + (size_t)i * stride;
    o[0] = tmp[i].x; o[1] = tmp[i].y; o[2] = tmp[i].z;
    if (stride >= 4) o[3] = 1.f;
  }
}

bool is_ndt(const s3d_reg_params* p) {
  return p->registration_algorithm == S3D_ALG_NDT || p->registration_algorithm == S3D_ALG_NDT_OMP;
}

float host_ord2f(unsigned int u) {
  const unsigned int v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
  std::memcpy(&f, &v, sizeof f);
  return f;
}

constexpr int kNdtBlocks = 64;   // blocks per derivative pass and pair (fixed: the sum order must not depend on the batch)

// doNDT (PointCloudSensor.cpp:84-117) for every pair of `b` (voxel filter and grid already staged and
// downloaded).  The voxel statistics and every derivative pass run on the device — one launch per ROUND over
// all pairs that still iterate — and the Newton / More-Thuente state machines (s3d_ndt.h, one per pair) run on
// the host between the rounds (one 224-byte result per pair and round comes back).  NDT_OMP is the same optimiser over
// pclomp's default neighbourhood (DIRECT7: the voxel of the point and its six face neighbours, s3d_ndt_derivatives_kernel<true>).
// statuses / results / infos: one per pair.
void align_ndt_pairs(Batch& b, const s3d_reg_params* params, const double* guesses, std::vector<int>& statuses,
                     std::vector<std::array<double, 16>>& results, std::vector<s3d_align_info>& infos) {
  s3d_context* ctx = b.ctx;
  hipStream_t st = ctx->stream;
  const int NP = b.P();
  statuses.assign(NP, S3D_STATUS_OK);
  results.assign(NP, std::array<double, 16>{});
  infos.assign(NP, s3d_align_info{});
  struct Scratch {   // released on every exit path
    std::vector<void*> ptrs;
    void* get(size_t bytes) { void* p = nullptr; HIPCHK(hipMalloc(&p, std::max<size_t>(bytes, 16))); ptrs.push_back(p); return p; }
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  } S;
  // ---- gates that precede the registration, voxel layouts of the distinct target clouds (= slam3d sources)
  std::vector<NdtGrid> grids;
  std::map<int, int> grid_of_slot;
  std::vector<int> pair_grid(NP, -1);
  for (int p = 0; p < NP; ++p) {
    const PairDev& P = b.h_pairs[p];
    const SlotDev& Ss = b.h_slots[P.slot_s];
    const SlotDev& St = b.h_slots[P.slot_t];
    for (int i = 0; i < 16; ++i) results[p][i] = (i % 5 == 0) ? 1.0 : 0.0;
    infos[p].n_source_filtered = Ss.n; infos[p].n_target_filtered = St.n;
    if (St.n < 100 || Ss.n < 100) { statuses[p] = S3D_STATUS_TOO_FEW_POINTS; continue; }     // :134-135
    if (!(params->resolution > 0.f)) { statuses[p] = S3D_STATUS_INVALID_ARGUMENT; continue; }
    auto it = grid_of_slot.find(P.slot_s);
    if (it == grid_of_slot.end()) {
      float mn[3], mx[3];
      for (int a = 0; a < 3; ++a) { mn[a] = host_ord2f(Ss.bb[a]); mx[a] = host_ord2f(Ss.bb[3 + a]); }
      NdtGrid G;
      std::memset(&G, 0, sizeof G);
      G.slot = P.slot_s; G.n = Ss.n;
      G.vp = voxel_params_from_bbox(mn, mx, params->resolution);
      const long long cells = (long long)G.vp.div_b[0] * G.vp.div_b[1] * G.vp.div_b[2];
      if (G.vp.passthrough || cells > (1ll << 27)) { statuses[p] = S3D_STATUS_INVALID_ARGUMENT; continue; }
      G.table = (int*)S.get(sizeof(int) * (size_t)cells);
      G.cells = (double*)S.get(sizeof(double) * kNdtCellDoubles * (size_t)(Ss.n / 6 + 1));
      G.counter = (int*)S.get(sizeof(int));
      HIPCHK(hipMemsetAsync(G.table, 0xFF, sizeof(int) * (size_t)cells, st));
      HIPCHK(hipMemsetAsync(G.counter, 0, sizeof(int), st));
      it = grid_of_slot.emplace(P.slot_s, (int)grids.size()).first;
      grids.push_back(G);
    }
    pair_grid[p] = it->second;
  }
  const int NG = (int)grids.size();
  std::vector<int> n_cells(std::max(NG, 1), 0);
  NdtGrid* d_grids = nullptr;
  if (NG > 0) {
    d_grids = (NdtGrid*)S.get(sizeof(NdtGrid) * NG);
    HIPCHK(hipMemcpyAsync(d_grids, grids.data(), sizeof(NdtGrid) * NG, hipMemcpyHostToDevice, st));
    int max_n = 1;
    for (const NdtGrid& G : grids) max_n = std::max(max_n, G.n);
    k_ndt_keys<<<dim3(cdiv(max_n, kBlock), NG), kBlock, 0, st>>>(b.d_slots(), d_grids, b.filt(), b.kA(), b.vA());
    k_ndt_select_slots<<<cdiv(b.C(), 64), 64, 0, st>>>(b.d_slots(), b.C(), d_grids, NG);
    b.sort(Batch::SortPlan{4, 8}, b.C());   // result back in A
    k_ndt_cells<<<dim3(cdiv(max_n, kBlock), NG), kBlock, 0, st>>>(b.d_slots(), d_grids, b.filt(), b.kA(), b.vA());
    for (int g = 0; g < NG; ++g)
      HIPCHK(hipMemcpyAsync(&n_cells[g], grids[g].counter, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  // ---- the optimisers, advanced in lock step: one derivative launch per round over the pairs still iterating
  double d1, d2;
  ndt::gauss_constants(params->outlier_ratio, (double)params->resolution, &d1, &d2);
  const float r2 = params->resolution * params->resolution;
  // NDT_OMP = pclomp's transform, whose default neighbour search is DIRECT7 (the voxel of the point + its six face
  // neighbours) where PCL's NDT asks a kd-tree for the cells within `resolution`
  const bool direct7 = params->registration_algorithm == S3D_ALG_NDT_OMP;
  std::vector<std::unique_ptr<ndt::Solver>> solver(NP);
  std::vector<ndt::Result> res(NP);
  for (int p = 0; p < NP; ++p) {
    float guess_f[16];
    for (int i = 0; i < 16; ++i) guess_f[i] = (float)guesses[(size_t)p * 16 + i];   // :104 guess.matrix().cast<float>()
    std::memcpy(res[p].T, guess_f, sizeof guess_f);
    res[p].converged = 0; res[p].iterations = 0; res[p].evaluations = 0;
    if (statuses[p] == S3D_STATUS_OK && n_cells[pair_grid[p]] > 0)
      solver[p].reset(new ndt::Solver(guess_f, params->step_size, params->transformation_epsilon,
                                      params->maximum_iterations));
  }
  NdtJob* d_jobs = (NdtJob*)S.get(sizeof(NdtJob) * std::max(NP, 1));
  double* d_part = (double*)S.get(sizeof(double) * NDT_NACC * kNdtBlocks * (size_t)std::max(NP, 1));
  double* d_out = (double*)S.get(sizeof(double) * NDT_NACC * (size_t)std::max(NP, 1));
  std::vector<NdtJob> jobs;
  std::vector<int> job_pair;
  std::vector<double> h_out;
  for (;;) {
    jobs.clear(); job_pair.clear();
    for (int p = 0; p < NP; ++p) {
      if (!solver[p] || !solver[p]->pending()) continue;
      NdtJob J;
      std::memset(&J, 0, sizeof J);
      J.slot_in = b.h_pairs[p].slot_t;
      J.m = b.h_slots[J.slot_in].n;
      J.grid = pair_grid[p];
      J.want_hessian = solver[p]->request_hessian() ? 1 : 0;
      std::memcpy(J.T.m, solver[p]->request_T(), sizeof J.T.m);
      ndt::angle_derivatives(solver[p]->request_p(), J.ang.dR, J.ang.d2R);
      jobs.push_back(J);
      job_pair.push_back(p);
    }
    const int NJ = (int)jobs.size();
    if (NJ == 0) break;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs.data(), sizeof(NdtJob) * NJ, hipMemcpyHostToDevice, st));
    if (direct7)
      s3d_ndt_derivatives_kernel<true><<<dim3(kNdtBlocks, NJ), kBlock, 0, st>>>(b.d_slots(), d_grids, d_jobs, b.filt(), r2,
                                                                                params->resolution, d1, d2, d_part);
    else
      s3d_ndt_derivatives_kernel<false><<<dim3(kNdtBlocks, NJ), kBlock, 0, st>>>(b.d_slots(), d_grids, d_jobs, b.filt(), r2,
                                                                                 params->resolution, d1, d2, d_part);
    k_ndt_reduce<<<NJ, 64, 0, st>>>(d_part, kNdtBlocks, d_out);
    h_out.resize((size_t)NJ * NDT_NACC);
    HIPCHK(hipMemcpyAsync(h_out.data(), d_out, sizeof(double) * NDT_NACC * (size_t)NJ, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < NJ; ++j) {
      const double* h = h_out.data() + (size_t)j * NDT_NACC;
      ndt::Eval e;
      e.score = h[0];
      for (int i = 0; i < 6; ++i) e.g[i] = h[1 + i];
      int k = 7;
      for (int i = 0; i < 6; ++i)
        for (int jj = i; jj < 6; ++jj, ++k) e.H[i * 6 + jj] = e.H[jj * 6 + i] = jobs[j].want_hessian ? h[k] : 0.0;
      solver[job_pair[j]]->feed(e);
    }
  }
  for (int p = 0; p < NP; ++p)
    if (solver[p]) res[p] = solver[p]->result();
  // ---- getFitnessScore(max_correspondence_distance) (:107): the K8 path on the final transformations
  for (int p = 0; p < NP; ++p) {
    Mat4f Tf;
    std::memcpy(Tf.m, res[p].T, sizeof Tf.m);
    b.h_pairs[p].final_T = Tf;
    HIPCHK(hipMemcpyAsync(&b.d_pairs()[p].final_T, &b.h_pairs[p].final_T, sizeof(Mat4f), hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemsetAsync(ctx->corr_d2.p, 0xFF, 4 * std::max<size_t>(b.total_corr, 4), st));
  HIPCHK(hipMemsetAsync(ctx->corr_lb.p, 0, 4 * std::max<size_t>(b.total_corr, 4), st));
  b.launch_nn(1, (float)(std::sqrt(std::max(b.rp.fit_range, 0.0)) * 1.0001));
  double* part = (double*)ctx->partials.p;
  s3d_fitness_partial_kernel<<<dim3(b.accum_blocks, NP), kBlock, 0, st>>>(b.d_pairs(), b.d_slots(), (float*)ctx->corr_d2.p,
                                                                          part, b.rp);
  k_fitness_final<<<cdiv(NP, 64), 64, 0, st>>>(b.d_pairs(), part, NP);
  std::vector<Mat4f> finals(NP);
  for (int p = 0; p < NP; ++p) finals[p] = b.h_pairs[p].final_T;
  b.download();
  for (int p = 0; p < NP; ++p) {
    if (statuses[p] != S3D_STATUS_OK) continue;
    PairDev& Pd = b.h_pairs[p];
    Pd.converged = res[p].converged;
    Pd.iterations = res[p].iterations;
    Pd.correspondences = n_cells[pair_grid[p]];
    Pd.evals_total = res[p].evaluations;
    Pd.inner_total = 0;
    Pd.final_T = finals[p];
    statuses[p] = b.finish_pair(p, params, guesses + (size_t)p * 16, results[p].data(), &infos[p]);
  }
}

// align() of two device-resident clouds (PointCloudSensor.cpp:119-174)
// persistent: the clouds are caller-held handles (the pre-pass cache may keep their products); false for the
// host-buffer entry points, whose device copies die with the call
int align_dev(s3d_context* ctx, s3d_cloud* ps, s3d_cloud* pt, const double guess[16], const s3d_reg_params* params,
              const s3d_exec_options* opts, double result[16], s3d_align_info* info, bool persistent) {
  const int alg = check_algorithm(params, opts);
  Batch b;
  b.ctx = ctx;
  b.use_cache = persistent && opts && opts->cache_prepass != 0 && alg == S3D_STATUS_OK;
  int status;
  if (alg != S3D_STATUS_OK) {
    // the reference downsamples and applies the 100-point gate before it dispatches (:127-135)
    s3d_reg_params tmp = *params;
    tmp.registration_algorithm = S3D_ALG_GICP;
    tmp.maximum_iterations = 0;
    b.set_params(&tmp, opts);
    b.add_pairs(1, &ps, &pt, guess);
    b.allocate();
    b.stage_voxel();
    b.download();
    status = (b.h_slots[0].n < 100 || b.h_slots[b.h_pairs[0].slot_t].n < 100) ? S3D_STATUS_TOO_FEW_POINTS : alg;
    if (info) { info->n_source_filtered = b.h_slots[0].n; info->n_target_filtered = b.h_slots[b.h_pairs[0].slot_t].n; }
  } else if (is_ndt(params)) {
    b.set_params(params, opts);
    b.add_pairs(1, &ps, &pt, guess);
    b.allocate();
    b.stage_voxel();
    b.stage_grid();
    b.download();
    b.store_to_cache(false);
    std::vector<int> sts;
    std::vector<std::array<double, 16>> rs;
    std::vector<s3d_align_info> is;
    align_ndt_pairs(b, params, guess, sts, rs, is);
    status = sts[0];
    std::memcpy(result, rs[0].data(), sizeof(double) * 16);
    if (info) *info = is[0];
  } else {
    b.set_params(params, opts);
    b.add_pairs(1, &ps, &pt, guess);
    b.registration_batch = true;
    b.allocate();
    b.run_all();
    status = b.finish_pair(0, params, guess, result, info);
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return status;
}

struct ScopedDevice {
  explicit ScopedDevice(s3d_context* ctx) : lock(ctx->mtx) { HIPCHK(hipSetDevice(ctx->device)); }
  std::lock_guard<std::mutex> lock;
};

}  // namespace

// ------------------------------------------------------------------ C ABI

extern "C" {

static int create_constraint_impl(s3d_context* ctx, s3d_cloud* source, const double source_sensor_pose[16],
                                  s3d_cloud* target, const double target_sensor_pose[16], const double odometry[16],
                                  int loop, const s3d_reg_params* fine, const s3d_reg_params* coarse,
                                  double covariance_scale, const s3d_exec_options* opts, double relative_pose[16],
                                  double information[36], s3d_align_info* info, bool persistent);

void s3d_default_params(s3d_reg_params* p) try {
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
} catch (...) {}   // (a destructor-like entry point has no status to return)

int s3d_abi_version(void) { return S3D_ABI_VERSION; }
#ifndef S3D_SOURCE_HASH
#define S3D_SOURCE_HASH "unhashed build (compiled without csrc/Makefile)"
#endif
static const char kSourceHashTag[] = "S3D_SOURCE_HASH=" S3D_SOURCE_HASH;   // (findable in the file without loading it)
const char* s3d_source_hash(void) { return kSourceHashTag + 16; }

int s3d_backend_info(int device, char* buf, int len) try {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    if (buf && len > 0) std::snprintf(buf, len, "no HIP device");
    return S3D_STATUS_BACKEND_ERROR;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return S3D_STATUS_BACKEND_ERROR;
  if (buf && len > 0)
    std::snprintf(buf, len, "%s|%s|%d|%zu", prop.name, prop.gcnArchName, prop.multiProcessorCount,
                  (size_t)prop.totalGlobalMem);
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr); }

static int context_create(int device, void* hip_stream, int priority_class, s3d_context** out,
                          const uint32_t* cu_mask = nullptr, int cu_words = 0);
int s3d_context_create(int device, void* hip_stream, s3d_context** out) { return context_create(device, hip_stream, 0, out); }
int s3d_cu_masks(int device, int reserved_cus, uint32_t* reserved_mask, uint32_t* rest_mask, int max_words) try {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
  const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
  if (!reserved_mask || !rest_mask || words > max_words || reserved_cus <= 0 || reserved_cus >= cus) return -1;
  for (int w = 0; w < words; ++w) { reserved_mask[w] = 0u; rest_mask[w] = 0u; }
  for (int c = 0; c < cus; ++c) (c < reserved_cus ? reserved_mask : rest_mask)[c / 32] |= 1u << (c % 32);
  return words;
} catch (...) { return fail_current(nullptr); }
int s3d_context_create_cu_mask(int device, const uint32_t* cu_mask, int n_words, s3d_context** out) try {
  if (!cu_mask || n_words <= 0 || n_words > 32) return S3D_STATUS_INVALID_ARGUMENT;
  bool any = false;
  for (int i = 0; i < n_words; ++i) any = any || cu_mask[i] != 0u;
  if (!any) return S3D_STATUS_INVALID_ARGUMENT;
  return context_create(device, nullptr, 0, out, cu_mask, n_words);
} catch (...) { return fail_current(nullptr); }
int s3d_context_create_priority(int device, int priority_class, s3d_context** out) try {
  if (priority_class < 0 || priority_class > 1) return S3D_STATUS_INVALID_ARGUMENT;
  return context_create(device, nullptr, priority_class, out);
} catch (...) { return fail_current(nullptr); }
static int context_create(int device, void* hip_stream, int priority_class, s3d_context** out, const uint32_t* cu_mask,
                          int cu_words) try {
  if (!out) return S3D_STATUS_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return S3D_STATUS_BACKEND_ERROR;  // no CPU fallback, by design
  s3d_context* ctx = new s3d_context();
  ctx->device = device;
  try {
    HIPCHK(hipSetDevice(device));
    if (hip_stream) {
      ctx->stream = (hipStream_t)hip_stream;
    } else {
      if (cu_mask) {
        // (HIP has no flag argument here: a CU-masked stream is a BLOCKING stream, i.e. it synchronises with the legacy
        // null stream of the process.  The library itself issues nothing on the null stream - copy_to_host - but an
        // application that does, e.g. through plain hipMemcpy, serialises such a context with that work.)
        HIPCHK(hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)cu_words, cu_mask));
      } else if (priority_class > 0) {
        int least = 0, greatest = 0;     // (numerically lower = higher priority)
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest));
      } else {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->plain_stream = true;
      }
      ctx->own_stream = true;
    }
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 4 < ctx->cache_limit) ctx->cache_limit = free_b / 4;
    }
  } catch (...) {
    fail_current(ctx);
    delete ctx;
    return S3D_STATUS_BACKEND_ERROR;
  }
  *out = ctx;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr); }

void s3d_context_destroy(s3d_context* ctx) try {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  ctx->release_all();
  if (ctx->h_active) (void)hipHostFree(ctx->h_active);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_copy) (void)hipHostFree(ctx->h_copy);
  if (ctx->d_copy.p) (void)hipFree(ctx->d_copy.p);
  for (hipEvent_t e : ctx->ev) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->nn_ev) (void)hipEventDestroy(e);
  ctx->release_upload_lanes();
  if (ctx->twin) { s3d_context_destroy(ctx->twin); ctx->twin = nullptr; }
  if (ctx->twin_ev) (void)hipEventDestroy(ctx->twin_ev);
  if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
  for (hipEvent_t e : ctx->side_ev) if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
} catch (...) {}   // (a destructor-like entry point has no status to return)

const char* s3d_last_error(const s3d_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int s3d_last_profile(const s3d_context* ctx, s3d_profile* out) try {
  if (!ctx || !out) return S3D_STATUS_INVALID_ARGUMENT;
  *out = ctx->prof;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr); }

int s3d_context_cache_control(s3d_context* ctx, long long limit_bytes, int clear, s3d_cache_stats* stats) try {
  if (!ctx) return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    if (ctx->stream) HIPCHK(hipStreamSynchronize(ctx->stream));
    if (limit_bytes > 0) ctx->cache_limit = (size_t)limit_bytes;
    if (clear) ctx->cache_clear();
    while (ctx->cache_bytes > ctx->cache_limit && !ctx->cache.empty()) {   // a lowered budget: oldest first
      auto victim = ctx->cache.begin();
      for (auto it = ctx->cache.begin(); it != ctx->cache.end(); ++it)
        if (it->second.last_use < victim->second.last_use) victim = it;
      ctx->cache_drop(victim);
    }
  } catch (...) {
    return fail_current(ctx);
  }
  if (stats) {
    stats->entries = (long long)ctx->cache.size(); stats->bytes = (long long)ctx->cache_bytes;
    stats->hits = ctx->cache_hits; stats->misses = ctx->cache_misses;
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

// ---- the cached pre-pass products of one cloud as a host blob (checkpoints: GraphSerialization.cpp:14-66 writes one
// <index>.s3dm per vertex; a caller can put this blob next to it and hand it back after fromFolder, :68-135)
namespace {
constexpr uint32_t kBlobMagic = 0x42443353u, kBlobEntryMagic = 0x45443353u, kBlobVersion = 3u;   // "S3DB", "S3DE"; 2: payload hash per entry; 3: layout + fused grid
struct BlobHeader { uint32_t magic, version, entries, n_raw; unsigned long long points_hash; };
struct BlobEntry {
  uint32_t magic, leaf_bits, h0_bits; int cell_cap;
  int n, ncells, k_normals, has_sorted3;
  unsigned int bb[6];
  s3d::VoxelParams vp;
  s3d::GridParams g;
  int layout;                         // CacheKey::layout (1: the fused pre-pass - no filt, sorted.w = voxel keys)
  s3d::FusedGrid fz;
  unsigned long long payload_bytes;   // filt 16 n (layout 0 only) | sorted 16 n | sorted3 sizeof(CorrVec) n | normals 16 n | cells 4 (ncells + 1)
  unsigned long long payload_hash;    // FNV-1a of those bytes: a damaged checkpoint is refused, not installed
};
// version 2 (the revision before the fused pre-pass): the same entry without `layout` and `fz` - still accepted by
// s3d_cloud_cache_import as layout 0 (a checkpoint written by that revision stays valid); export writes version 3
struct BlobEntryV2 {
  uint32_t magic, leaf_bits, h0_bits; int cell_cap;
  int n, ncells, k_normals, has_sorted3;
  unsigned int bb[6];
  s3d::VoxelParams vp;
  s3d::GridParams g;
  unsigned long long payload_bytes, payload_hash;
};
// the entry header at `at` of a blob of `version` as a version-3 entry; returns the header's size in the blob (0: truncated)
size_t blob_read_entry(uint32_t version, const char* at, const char* end, BlobEntry* E) {
  if (version >= 3u) {
    if (end - at < (long long)sizeof *E) return 0;
    std::memcpy(E, at, sizeof *E);
    return sizeof *E;
  }
  BlobEntryV2 o;
  if (end - at < (long long)sizeof o) return 0;
  std::memcpy(&o, at, sizeof o);
  std::memset(E, 0, sizeof *E);
  E->magic = o.magic; E->leaf_bits = o.leaf_bits; E->h0_bits = o.h0_bits; E->cell_cap = o.cell_cap;
  E->n = o.n; E->ncells = o.ncells; E->k_normals = o.k_normals; E->has_sorted3 = o.has_sorted3;
  std::memcpy(E->bb, o.bb, sizeof o.bb);
  E->vp = o.vp; E->g = o.g; E->layout = 0;
  E->payload_bytes = o.payload_bytes; E->payload_hash = o.payload_hash;
  return sizeof o;
}
unsigned long long fnv1a64(const void* data, size_t bytes) {
  const unsigned char* p = (const unsigned char*)data;
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
// hash of the cloud's points as they lie on the device (x, y, z of every point; the fourth float is not data)
unsigned long long cloud_points_hash(s3d_context* ctx, const s3d_cloud* c) {
  std::vector<float> h((size_t)c->n * 4);
  if (c->n) {
    HIPCHK(hipMemcpyAsync(h.data(), c->d, 16 * (size_t)c->n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  std::vector<float> xyz((size_t)c->n * 3);
  for (int i = 0; i < c->n; ++i) for (int a = 0; a < 3; ++a) xyz[(size_t)i * 3 + a] = h[(size_t)i * 4 + a];
  return fnv1a64(xyz.data(), xyz.size() * 4);
}
size_t blob_payload_bytes(int n, int ncells, int layout) {
  return (size_t)n * ((layout ? 0 : 16) + 16 + sizeof(CorrVec) + sizeof(NormalRec)) + 4 * ((size_t)ncells + 1);
}
}  // namespace

long long s3d_cloud_cache_export(s3d_context* ctx, const s3d_cloud* cloud, void* buffer, long long capacity) try {
  if (!ctx || !cloud || capacity < 0) return -(long long)S3D_STATUS_INVALID_ARGUMENT;   // (< 0: a status, as the header says)
  try {
    ScopedDevice sd(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    size_t need = sizeof(BlobHeader);
    uint32_t entries = 0;
    auto first = ctx->cache.lower_bound(CacheKey{cloud->uid, 0, 0, 0});
    for (auto it = first; it != ctx->cache.end() && it->first.uid == cloud->uid; ++it) {
      need += sizeof(BlobEntry) + blob_payload_bytes(it->second.snap.n, it->second.snap.g.ncells, it->first.layout);
      ++entries;
    }
    if (entries == 0) return 0;                                   // nothing cached for this cloud
    if (!buffer || capacity < (long long)need) return (long long)need;
    char* out = (char*)buffer;
    BlobHeader H;
    H.magic = kBlobMagic; H.version = kBlobVersion; H.entries = entries; H.n_raw = (uint32_t)cloud->n;
    H.points_hash = cloud_points_hash(ctx, cloud);
    std::memcpy(out, &H, sizeof H); out += sizeof H;
    std::vector<char*> entry_at;
    for (auto it = first; it != ctx->cache.end() && it->first.uid == cloud->uid; ++it) {
      const CacheEntry& e = it->second;
      entry_at.push_back(out);
      const size_t n = (size_t)e.snap.n, cells_n = (size_t)e.snap.g.ncells + 1;
      BlobEntry E;
      std::memset(&E, 0, sizeof E);
      E.magic = kBlobEntryMagic; E.leaf_bits = it->first.leaf_bits; E.h0_bits = it->first.h0_bits; E.cell_cap = it->first.cell_cap;
      E.n = e.snap.n; E.ncells = e.snap.g.ncells; E.k_normals = e.k_normals; E.has_sorted3 = e.has_sorted3 ? 1 : 0;
      std::memcpy(E.bb, e.snap.bb, sizeof E.bb);
      E.vp = e.snap.vp; E.g = e.snap.g; E.layout = it->first.layout; E.fz = e.snap.fz;
      E.payload_bytes = blob_payload_bytes(e.snap.n, e.snap.g.ncells, E.layout);
      std::memcpy(out, &E, sizeof E); out += sizeof E;
      const size_t parts[5][2] = {{e.o_filt, E.layout ? 0 : 16 * n}, {e.o_sorted, 16 * n}, {e.o_sorted3, sizeof(CorrVec) * n},
                                  {e.o_normals, sizeof(NormalRec) * n}, {e.o_cells, 4 * cells_n}};
      for (int a = 0; a < 5; ++a) {
        const bool valid = !((a == 2 && !e.has_sorted3) || (a == 3 && e.k_normals == 0));   // never written: zeros
        if (parts[a][1] && valid) HIPCHK(hipMemcpyAsync(out, e.block + parts[a][0], parts[a][1], hipMemcpyDeviceToHost, ctx->stream));
        else std::memset(out, 0, parts[a][1]);
        out += parts[a][1];
      }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (char* at : entry_at) {     // the payloads have arrived: their hashes into the entry headers
      BlobEntry E;
      std::memcpy(&E, at, sizeof E);
      E.payload_hash = fnv1a64(at + sizeof E, (size_t)E.payload_bytes);
      std::memcpy(at, &E, sizeof E);
    }
    return (long long)need;
  } catch (...) {
    return -(long long)fail_current(ctx);
  }
} catch (...) { return -(long long)fail_current(ctx); }

int s3d_cloud_cache_import(s3d_context* ctx, const s3d_cloud* cloud, const void* blob, long long size) try {
  if (!ctx || !cloud || !blob || size < (long long)sizeof(BlobHeader)) return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    const char* in = (const char*)blob;
    const char* const end = in + size;
    BlobHeader H;
    std::memcpy(&H, in, sizeof H); in += sizeof H;
    if (H.magic != kBlobMagic || (H.version != kBlobVersion && H.version != 2u)) { ctx->err = "cache blob: bad magic / version"; return S3D_STATUS_INVALID_ARGUMENT; }
    if ((int)H.n_raw != cloud->n || H.points_hash != cloud_points_hash(ctx, cloud)) {
      ctx->err = "cache blob: made from a different point cloud";
      return S3D_STATUS_INVALID_ARGUMENT;
    }
    // validate everything before anything is installed
    const char* scan = in;
    for (uint32_t i = 0; i < H.entries; ++i) {
      BlobEntry E;
      const size_t head = blob_read_entry(H.version, scan, end, &E);
      if (!head) { ctx->err = "cache blob: truncated"; return S3D_STATUS_INVALID_ARGUMENT; }
      scan += head;
      if (E.magic != kBlobEntryMagic || E.n < 0 || E.n > cloud->n || E.ncells < 0 || E.ncells != E.g.ncells || E.ncells > E.cell_cap ||
          (E.layout != 0 && E.layout != 1) || E.payload_bytes != blob_payload_bytes(E.n, E.ncells, E.layout) ||
          (unsigned long long)(end - scan) < E.payload_bytes) {
        ctx->err = "cache blob: corrupt entry";
        return S3D_STATUS_INVALID_ARGUMENT;
      }
      // the payload is what the kernels will index with: its checksum, then the invariants they rely on (a cell
      // table that is monotone and ends at n, grid dimensions that multiply to the cell count, sorted positions that
      // name points of the cloud) - a blob that fails any of them is refused as a whole
      bool ok = fnv1a64(scan, (size_t)E.payload_bytes) == E.payload_hash &&
                (long long)E.g.dim[0] * E.g.dim[1] * E.g.dim[2] == (long long)E.ncells && E.g.dim[0] > 0 && E.g.dim[1] > 0 &&
                E.g.dim[2] > 0;
      if (ok) {
        const size_t n = (size_t)E.n;
        const char* sorted_at = scan + (E.layout ? 0 : 16 * n);
        const char* cells_at = scan + n * ((E.layout ? 0 : 16) + 16 + sizeof(CorrVec) + sizeof(NormalRec));
        uint32_t prev = 0;
        for (size_t c = 0; ok && c <= (size_t)E.ncells; ++c) {
          uint32_t v;
          std::memcpy(&v, cells_at + 4 * c, 4);
          ok = v >= prev && v <= (uint32_t)E.n && (c > 0 || v == 0);
          prev = v;
        }
        ok = ok && prev == (uint32_t)E.n;
        for (size_t k = 0; ok && k < n; ++k) {
          uint32_t w;
          std::memcpy(&w, sorted_at + 16 * k + 12, 4);
          ok = E.layout ? w < 0x80000000u : w < (uint32_t)E.n;   // (an index into filt, or PCL's voxel key: a non-negative int)
        }
        if (E.layout) ok = ok && E.fz.ok == 1 && E.fz.m >= 2 && E.fz.msub >= 1;
      }
      if (!ok) { ctx->err = "cache blob: damaged payload (checksum / cell table / indices)"; return S3D_STATUS_INVALID_ARGUMENT; }
      scan += E.payload_bytes;
    }
    ++ctx->cache_clock;
    for (uint32_t i = 0; i < H.entries; ++i) {
      BlobEntry E;
      in += blob_read_entry(H.version, in, end, &E);
      const CacheKey key{cloud->uid, E.leaf_bits, E.h0_bits, E.cell_cap, E.layout};
      auto old = ctx->cache.find(key);
      if (old != ctx->cache.end()) { HIPCHK(hipStreamSynchronize(ctx->stream)); ctx->cache_drop(old); }
      CacheEntry e;
      const size_t n = (size_t)E.n, cells_n = (size_t)E.ncells + 1;
      auto place = [&](size_t bytes) { const size_t o = e.bytes; e.bytes += (bytes + 255) & ~(size_t)255; return o; };
      e.o_filt = place(E.layout ? 0 : 16 * n); e.o_sorted = place(16 * n); e.o_sorted3 = place(sizeof(CorrVec) * n);
      e.o_normals = place(sizeof(NormalRec) * n); e.o_cells = place(4 * cells_n);
      e.bytes = std::max<size_t>(e.bytes, 256);
      if (ctx->cache_bytes + e.bytes > ctx->cache_limit) { in += E.payload_bytes; continue; }   // over the budget: stays uncached
      if (hipMalloc((void**)&e.block, e.bytes) != hipSuccess) { (void)hipGetLastError(); in += E.payload_bytes; continue; }
      const size_t parts[5][2] = {{e.o_filt, E.layout ? 0 : 16 * n}, {e.o_sorted, 16 * n}, {e.o_sorted3, sizeof(CorrVec) * n},
                                  {e.o_normals, sizeof(NormalRec) * n}, {e.o_cells, 4 * cells_n}};
      for (int a = 0; a < 5; ++a) {
        if (parts[a][1]) HIPCHK(hipMemcpyAsync(e.block + parts[a][0], in, parts[a][1], hipMemcpyHostToDevice, ctx->stream));
        in += parts[a][1];
      }
      HIPCHK(hipStreamSynchronize(ctx->stream));      // (the caller's buffer is pageable and may go away)
      std::memset(&e.snap, 0, sizeof e.snap);
      e.snap.n_raw = cloud->n; e.snap.cell_cap = E.cell_cap; e.snap.n = E.n;
      std::memcpy(e.snap.bb, E.bb, sizeof E.bb);
      e.snap.vp = E.vp; e.snap.g = E.g; e.snap.fz = E.fz;
      e.k_normals = E.k_normals; e.has_sorted3 = E.has_sorted3 != 0;
      e.last_use = ctx->cache_clock;
      ctx->cache_bytes += e.bytes;
      ctx->cache[key] = e;
    }
    return S3D_STATUS_OK;
  } catch (...) {
    return fail_current(ctx);
  }
} catch (...) { return fail_current(ctx); }

int s3d_cloud_upload(s3d_context* ctx, const float* xyz, int n, int stride, s3d_cloud** out) try {
  if (!ctx || !out || n < 0 || stride < 3 || (n > 0 && !xyz)) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud* c = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, xyz, n, stride, c);
  } catch (...) {
    free_cloud(c);
    delete c;
    return fail_current(ctx);
  }
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_cloud_upload_many(s3d_context* ctx, int n_clouds, const float* const* xyz, const int* n, int stride, s3d_cloud** out) try {
  if (!ctx || n_clouds < 0 || stride < 3 || (n_clouds > 0 && (!xyz || !n || !out))) return S3D_STATUS_INVALID_ARGUMENT;
  for (int i = 0; i < n_clouds; ++i)
    if (n[i] < 0 || (n[i] > 0 && !xyz[i])) return S3D_STATUS_INVALID_ARGUMENT;
  if (n_clouds == 0) return S3D_STATUS_OK;
  std::vector<s3d_cloud*> made((size_t)n_clouds, nullptr);
  for (int i = 0; i < n_clouds; ++i) made[(size_t)i] = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    upload_many(ctx, n_clouds, xyz, n, stride, made.data());
  } catch (...) {
    for (s3d_cloud* c : made) delete c;
    return fail_current(ctx);
  }
  for (int i = 0; i < n_clouds; ++i) out[i] = made[(size_t)i];
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_context_set_upload_threads(s3d_context* ctx, int n) try {
  if (!ctx || n < 0) return S3D_STATUS_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(ctx->mtx);
  ctx->upload_threads_cap = n;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_cloud_wrap_device(s3d_context* ctx, const void* device_float4, int n, s3d_cloud** out) try {
  if (!ctx || !out || n < 0 || (n > 0 && !device_float4)) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud* c = new s3d_cloud();
  c->d = (float4*)device_float4;
  c->n = n;
  c->owned = false;
  c->uid = g_cloud_uid++;
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_cloud_size(const s3d_cloud* c) { return c ? c->n : 0; }

void s3d_cloud_release(s3d_context* ctx, s3d_cloud* c) try {
  if (!c) return;
  if (ctx) {
    std::lock_guard<std::mutex> lock(ctx->mtx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->cache_forget_cloud(c->uid);   // eviction: the cached pre-pass products die with the cloud
    free_cloud(c);
  } else {
    free_cloud(c);
  }
  delete c;
} catch (...) {}   // (a destructor-like entry point has no status to return)

int s3d_align_batch(s3d_context* ctx, int n_pairs, s3d_cloud* const* sources, s3d_cloud* const* targets,
                    const double* guesses, const s3d_reg_params* params, const s3d_exec_options* opts,
                    s3d_edge_record* records, s3d_align_info* infos) try {
  if (!ctx || n_pairs < 0 || !params || (n_pairs > 0 && (!sources || !targets || !guesses || !records)))
    return S3D_STATUS_INVALID_ARGUMENT;
  const int alg = check_algorithm(params, opts);
  if (alg != S3D_STATUS_OK) {
    for (int p = 0; p < n_pairs; ++p) {
      std::memset(&records[p], 0, sizeof records[p]);
      records[p].transform[0] = records[p].transform[4] = records[p].transform[8] = 1.0;
      records[p].status = alg;
    }
    return alg;
  }
  try {
    ScopedDevice sd(ctx);
    if (is_ndt(params)) {   // NDT: device passes for all pairs per round, host state machines in between
      Batch b;
      b.ctx = ctx;
      b.use_cache = opts && opts->cache_prepass != 0;
      b.set_params(params, opts);
      b.add_pairs(n_pairs, sources, targets, guesses);
      b.allocate();
      b.stage_voxel();
      b.stage_grid();
      b.download();
      b.store_to_cache(false);
      std::vector<int> sts;
      std::vector<std::array<double, 16>> rs;
      std::vector<s3d_align_info> is;
      align_ndt_pairs(b, params, guesses, sts, rs, is);
      for (int p = 0; p < n_pairs; ++p) {
        s3d_edge_record& r = records[p];
        for (int c = 0; c < 4; ++c)
          for (int rr = 0; rr < 3; ++rr) r.transform[c * 3 + rr] = rs[p][c * 4 + rr];
        r.fitness = is[p].fitness;
        r.iterations = is[p].iterations;
        r.correspondences = is[p].correspondences;
        r.status = sts[p];
        if (infos) infos[p] = is[p];
      }
      return S3D_STATUS_OK;
    }
    Batch b;
    b.ctx = ctx;
    b.use_cache = opts && opts->cache_prepass != 0;
    b.set_params(params, opts);
    b.add_pairs(n_pairs, sources, targets, guesses);
    b.registration_batch = true;
    b.allocate();
    b.run_all();
    for (int p = 0; p < n_pairs; ++p) {
      double result[16];
      s3d_align_info info{};
      const int st = b.finish_pair(p, params, guesses + (size_t)p * 16, result, &info);
      s3d_edge_record& r = records[p];
      for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 3; ++rr) r.transform[c * 3 + rr] = result[c * 4 + rr];
      r.fitness = info.fitness;
      r.iterations = info.iterations;
      r.correspondences = info.correspondences;
      r.status = st;
      if (infos) infos[p] = info;
    }
  } catch (...) {
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_align(s3d_context* ctx, const float* source_xyz, int n_source, int stride_source, const float* target_xyz,
              int n_target, int stride_target, const double guess[16], const s3d_reg_params* params,
              const s3d_exec_options* opts, double result[16], s3d_align_info* info) try {
  if (!ctx || !guess || !params || !result || n_source < 0 || n_target < 0 || stride_source < 3 || stride_target < 3)
    return S3D_STATUS_INVALID_ARGUMENT;
  for (int i = 0; i < 16; ++i) result[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (info) std::memset(info, 0, sizeof *info);
  s3d_cloud cs, ct;
  int status = S3D_STATUS_OK;
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, source_xyz, n_source, stride_source, &cs);
    upload_cloud(ctx, target_xyz, n_target, stride_target, &ct);
    status = align_dev(ctx, &cs, &ct, guess, params, opts, result, info, false);
    free_cloud(&cs);
    free_cloud(&ct);
  } catch (...) {
    free_cloud(&cs);
    free_cloud(&ct);
    return fail_current(ctx);
  }
  return status;
} catch (...) { return fail_current(ctx); }

int s3d_create_constraint(s3d_context* ctx, const float* source_xyz, int n_source, int stride_source,
                          const double source_sensor_pose[16], const float* target_xyz, int n_target,
                          int stride_target, const double target_sensor_pose[16], const double odometry[16], int loop,
                          const s3d_reg_params* fine, const s3d_reg_params* coarse, double covariance_scale,
                          const s3d_exec_options* opts, double relative_pose[16], double information[36],
                          s3d_align_info* info) try {
  if (!ctx || !source_sensor_pose || !target_sensor_pose || !odometry || !fine || (loop && !coarse) ||
      !relative_pose || !information || n_source < 0 || n_target < 0 || stride_source < 3 || stride_target < 3)
    return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud cs, ct;   // both clouds go up once, the coarse and the fine align() share them
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, source_xyz, n_source, stride_source, &cs);
    upload_cloud(ctx, target_xyz, n_target, stride_target, &ct);
  } catch (...) {
    free_cloud(&cs);
    free_cloud(&ct);
    return fail_current(ctx);
  }
  const int st = create_constraint_impl(ctx, &cs, source_sensor_pose, &ct, target_sensor_pose, odometry, loop, fine,
                                        coarse, covariance_scale, opts, relative_pose, information, info, false);
  {
    std::lock_guard<std::mutex> lock(ctx->mtx);
    (void)hipSetDevice(ctx->device);
    free_cloud(&cs);
    free_cloud(&ct);
  }
  return st;
} catch (...) { return fail_current(ctx); }

int s3d_voxel_downsample(s3d_context* ctx, const float* xyz, int n, int stride, double leaf_size, float* out_xyz,
                         int* n_out) try {
  if (!ctx || !n_out || n < 0 || stride < 3 || (n > 0 && (!xyz || !out_xyz))) return S3D_STATUS_INVALID_ARGUMENT;
  *n_out = 0;
  if (n == 0) return S3D_STATUS_OK;  // PointCloudSensor.cpp:193
  s3d_cloud c;
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, xyz, n, stride, &c);
    s3d_reg_params p;
    s3d_default_params(&p);
    p.point_cloud_density = leaf_size;
    Batch b;
    b.ctx = ctx;
    b.set_params(&p, nullptr);
    std::map<const s3d_cloud*, int> index;
    b.add_slot(&c, index);
    b.allocate();
    b.stage_voxel();
    b.download();
    const int m = b.h_slots[0].n;
    std::vector<float4> tmp((size_t)std::max(m, 1));
    if (m > 0) copy_to_host(ctx, tmp.data(), b.filt() + b.h_slots[0].off, sizeof(float4) * (size_t)m);
    for (int i = 0; i < m; ++i) {
      out_xyz[(size_t)i * 3 + 0] = tmp[i].x; out_xyz[(size_t)i * 3 + 1] = tmp[i].y; out_xyz[(size_t)i * 3 + 2] = tmp[i].z;
    }
    *n_out = m;
    free_cloud(&c);
  } catch (...) {
    free_cloud(&c);
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_nn_search(s3d_context* ctx, const float* target_xyz, int n, int stride_t, const float* query_xyz, int m,
                  int stride_q, double max_distance, int* idx, float* d2) try {
  if (!ctx || n < 0 || m < 0 || stride_t < 3 || stride_q < 3 || (m > 0 && (!idx || !d2 || !query_xyz)) ||
      (n > 0 && !target_xyz))
    return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud ct, cq;
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, target_xyz, n, stride_t, &ct);
    upload_cloud(ctx, query_xyz, m, stride_q, &cq);
    s3d_reg_params p;
    s3d_default_params(&p);
    p.point_cloud_density = 0.0;  // search the clouds as given
    p.max_correspondence_distance = max_distance;
    Batch b;
    b.ctx = ctx;
    b.set_params(&p, nullptr);
    s3d_cloud* ps = &ct; s3d_cloud* pq = &cq;
    const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    b.add_pairs(1, &ps, &pq, ident);
    b.allocate();
    b.stage_voxel();
    b.stage_grid();
    k_pair_init<<<1, 64, 0, ctx->stream>>>(b.d_pairs(), 1, (int*)ctx->n_active.p);  // final_T = guess = I
    HIPCHK(hipMemsetAsync(ctx->corr_d2.p, 0xFF, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->corr_lb.p, 0, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
    b.launch_nn(1, (float)(max_distance * 1.0001));
    // back to the caller's query order / target indices (keysA, keysB are free at this point)
    k_export_corr<<<dim3(cdiv(std::max(m, 1), kBlock), 1), kBlock, 0, ctx->stream>>>(
        b.d_pairs(), b.d_slots(), b.sorted(), (int*)ctx->corr_idx.p, (float*)ctx->corr_d2.p, (int*)b.kA(), (float*)b.kB());
    b.download();
    if (m > 0) {
      copy_to_host(ctx, idx, (int*)b.kA() + b.h_pairs[0].corr_off, sizeof(int) * (size_t)m);
      copy_to_host(ctx, d2, (float*)b.kB() + b.h_pairs[0].corr_off, sizeof(float) * (size_t)m);
    }
    free_cloud(&ct);
    free_cloud(&cq);
  } catch (...) {
    free_cloud(&ct);
    free_cloud(&cq);
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_knn_normals(s3d_context* ctx, const float* xyz, int n, int stride, int k, float* normals_xyz) try {
  if (!ctx || n < 0 || stride < 3 || k < 1 || k > 64 || (n > 0 && (!xyz || !normals_xyz)))
    return S3D_STATUS_INVALID_ARGUMENT;
  if (k > n) return S3D_STATUS_INVALID_ARGUMENT;  // PCL: "Number of points in cloud is less than k"
  s3d_cloud c;
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, xyz, n, stride, &c);
    s3d_reg_params p;
    s3d_default_params(&p);
    p.point_cloud_density = 0.0;
    p.correspondence_randomness = k;
    Batch b;
    b.ctx = ctx;
    b.set_params(&p, nullptr);
    std::map<const s3d_cloud*, int> index;
    b.add_slot(&c, index);
    b.allocate();
    b.stage_voxel();
    b.stage_grid();
    b.stage_normals();
    k_export_normals<<<dim3(cdiv(std::max(n, 1), kBlock), 1), kBlock, 0, ctx->stream>>>(b.d_slots(), b.sorted(), b.normals(),
                                                                                         b.filt());  // filt is free now
    b.download();
    std::vector<float4> tmp((size_t)std::max(n, 1));
    copy_to_host(ctx, tmp.data(), b.filt() + b.h_slots[0].off, sizeof(float4) * (size_t)n);
    for (int i = 0; i < n; ++i) {
      normals_xyz[(size_t)i * 3 + 0] = tmp[i].x; normals_xyz[(size_t)i * 3 + 1] = tmp[i].y;
      normals_xyz[(size_t)i * 3 + 2] = tmp[i].z;
    }
    free_cloud(&c);
  } catch (...) {
    free_cloud(&c);
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

// ---- test hooks (include/slam3d_hip_debug.h): not part of the drop-in API
int s3d_debug_raise(s3d_context* ctx, int kind) try {
  if (!ctx) return S3D_STATUS_INVALID_ARGUMENT;
  ScopedDevice sd(ctx);
  switch (kind) {
    case 0: HIPCHK(hipErrorInvalidValue); break;
    case 1: throw std::bad_alloc();
    case 2: throw std::length_error("vector::reserve");
    case 3: throw std::runtime_error("raised on request");
    case 4: throw 42;
    case 5: {
      std::exception_ptr err;
      std::thread t([&] { try { throw std::bad_alloc(); } catch (...) { err = std::current_exception(); } });
      t.join();
      if (err) std::rethrow_exception(err);
      break;
    }
    default: return S3D_STATUS_INVALID_ARGUMENT;
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

long long s3d_debug_fused_reruns(s3d_context* ctx) { return ctx ? ctx->fused_reruns : -1; }

// The pre-pass of a registration (voxel filter + search grid, fused or as two sorts) and ONE nearest-neighbour pass of
// the filtered target cloud's points against the filtered source cloud, everything as the registration lays it out:
// the cell-sorted points of both clouds (w = the tie-breaking id: index in pcl::VoxelGrid's output order, or PCL's
// voxel key on the fused path - the same order) and, per cell-sorted target point, the POSITION of its neighbour in
// the source's array and the float d2.
int s3d_debug_filtered_nn(s3d_context* ctx, s3d_cloud* source, s3d_cloud* target, double leaf, int fused,
                          double max_distance, int capacity, float* source_sorted_xyzw, int* n_source,
                          float* target_sorted_xyzw, int* n_target, int* corr_pos, float* corr_d2, int* fused_ok) try {
  if (!ctx || !source || !target || !(leaf > 0.0) || capacity < 0 || !n_source || !n_target || !fused_ok)
    return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    s3d_reg_params p;
    s3d_default_params(&p);
    p.point_cloud_density = leaf;
    p.max_correspondence_distance = max_distance;
    Batch b;
    b.ctx = ctx;
    b.set_params(&p, nullptr);
    const double ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    b.add_pairs(1, &source, &target, ident);
    b.allocate();
    b.fused = fused != 0;
    if (b.fused) b.stage_prepass_fused();
    else { b.stage_voxel(); b.stage_grid(); }
    k_pair_init<<<1, 64, 0, ctx->stream>>>(b.d_pairs(), 1, (int*)ctx->n_active.p);
    HIPCHK(hipMemsetAsync(ctx->corr_d2.p, 0xFF, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->corr_lb.p, 0, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
    b.launch_nn(1, (float)(max_distance * 1.0001));
    b.download();
    const PairDev& P = b.h_pairs[0];
    const SlotDev& Ss = b.h_slots[P.slot_s];
    const SlotDev& St = b.h_slots[P.slot_t];
    *n_source = Ss.n; *n_target = St.n;
    *fused_ok = b.fused ? std::min(Ss.fz.ok, St.fz.ok) : 0;
    if (Ss.n > capacity || St.n > capacity) return S3D_STATUS_INVALID_ARGUMENT;
    if (Ss.n > 0) copy_to_host(ctx, source_sorted_xyzw, b.sorted() + Ss.off, sizeof(float4) * (size_t)Ss.n);
    if (St.n > 0) {
      copy_to_host(ctx, target_sorted_xyzw, b.sorted() + St.off, sizeof(float4) * (size_t)St.n);
      copy_to_host(ctx, corr_pos, (int*)ctx->corr_idx.p + P.corr_off, sizeof(int) * (size_t)St.n);
      copy_to_host(ctx, corr_d2, (float*)ctx->corr_d2.p + P.corr_off, sizeof(float) * (size_t)St.n);
    }
  } catch (...) {
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_cloud_download(s3d_context* ctx, const s3d_cloud* c, float* xyz, int stride) try {
  if (!ctx || !c || stride < 3 || (c->n > 0 && !xyz)) return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    download_packed(ctx, c, xyz, stride);
  } catch (...) {
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_cloud_accumulate(s3d_context* ctx, int n_clouds, s3d_cloud* const* clouds, const double* poses,
                         const double frame[16], s3d_cloud** out) try {
  if (!ctx || !out || n_clouds < 0 || (n_clouds > 0 && (!clouds || !poses))) return S3D_STATUS_INVALID_ARGUMENT;
  for (int i = 0; i < n_clouds; ++i)
    if (!clouds[i]) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud* c = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    accumulate_dev(ctx, n_clouds, clouds, poses, frame, c);
  } catch (...) {
    free_cloud(c);
    delete c;
    return fail_current(ctx);
  }
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_align_clouds(s3d_context* ctx, s3d_cloud* source, s3d_cloud* target, const double guess[16],
                     const s3d_reg_params* params, const s3d_exec_options* opts, double result[16],
                     s3d_align_info* info) try {
  if (!ctx || !source || !target || !guess || !params || !result) return S3D_STATUS_INVALID_ARGUMENT;
  for (int i = 0; i < 16; ++i) result[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (info) std::memset(info, 0, sizeof *info);
  try {
    ScopedDevice sd(ctx);
    return align_dev(ctx, source, target, guess, params, opts, result, info, true);
  } catch (...) {
    return fail_current(ctx);
  }
} catch (...) { return fail_current(ctx); }

int s3d_create_constraint_clouds(s3d_context* ctx, s3d_cloud* source, const double source_sensor_pose[16],
                                 s3d_cloud* target, const double target_sensor_pose[16], const double odometry[16],
                                 int loop, const s3d_reg_params* fine, const s3d_reg_params* coarse,
                                 double covariance_scale, const s3d_exec_options* opts, double relative_pose[16],
                                 double information[36], s3d_align_info* info) try {
  return create_constraint_impl(ctx, source, source_sensor_pose, target, target_sensor_pose, odometry, loop, fine, coarse,
                                covariance_scale, opts, relative_pose, information, info, true);
} catch (...) { return fail_current(ctx); }

static int create_constraint_impl(s3d_context* ctx, s3d_cloud* source, const double source_sensor_pose[16],
                                  s3d_cloud* target, const double target_sensor_pose[16], const double odometry[16],
                                  int loop, const s3d_reg_params* fine, const s3d_reg_params* coarse,
                                  double covariance_scale, const s3d_exec_options* opts, double relative_pose[16],
                                  double information[36], s3d_align_info* info, bool persistent) try {
  if (!ctx || !source || !target || !source_sensor_pose || !target_sensor_pose || !odometry || !fine ||
      (loop && !coarse) || !relative_pose || !information)
    return S3D_STATUS_INVALID_ARGUMENT;
  if (info) std::memset(info, 0, sizeof *info);
  double sinv[16], tinv[16], guess[16], tmp[16];
  mat4d_inverse_isometry(source_sensor_pose, sinv);
  mat4d_inverse_isometry(target_sensor_pose, tinv);
  mat4d_mul(sinv, odometry, tmp);              // PointCloudSensor.cpp:274
  mat4d_mul(tmp, target_sensor_pose, guess);
  try {
    ScopedDevice sd(ctx);
    int st;
    double result[16];
    // (round 6) coarse + fine: the FINE registration's pre-pass (voxel filter at its own density, grid, k-NN normals - a
    // third of a registration of two scans) depends on the clouds only, not on the coarse result: it runs on a private
    // second context (workspace and stream of its own) while the coarse registration runs here, and the fine ICP loop
    // goes on there once its guess is known.  Same records as one after the other (S3D_DBG_NO_K4_OVERLAP).  Not with the
    // pre-pass cache (its entries belong to this context), the profile, NDT, or a context on a caller's / masked stream.
    const bool caching = persistent && opts && opts->cache_prepass != 0;
    const bool two = loop && ctx->plain_stream && !caching && !(opts && (opts->profile != 0 || (opts->debug_flags & S3D_DBG_NO_K4_OVERLAP))) &&
                     check_algorithm(fine, opts) == S3D_STATUS_OK && !is_ndt(fine) &&
                     (long long)source->n + (long long)target->n <= 400000ll;
    if (two && !ctx->twin) {
      s3d_context* tw = nullptr;
      if (context_create(ctx->device, nullptr, 0, &tw) == S3D_STATUS_OK) ctx->twin = tw;
      if (ctx->twin && hipEventCreateWithFlags(&ctx->twin_ev, hipEventDisableTiming) != hipSuccess) ctx->twin_ev = nullptr;
    }
    if (two && ctx->twin && ctx->twin_ev) {
      s3d_context* tw = ctx->twin;
      struct Drain {          // nothing of this call may still run on the twin when it returns or unwinds
        s3d_context* t;
        ~Drain() { (void)hipStreamSynchronize(t->stream); if (t->side_stream) (void)hipStreamSynchronize(t->side_stream); }
      } drain{tw};
      // (the clouds may have been uploaded on this context's stream a moment ago)
      HIPCHK(hipEventRecord(ctx->twin_ev, ctx->stream));
      HIPCHK(hipStreamWaitEvent(tw->stream, ctx->twin_ev, 0));
      Batch f;
      bool ahead = false;
      try {
        f.ctx = tw;
        f.use_cache = false;
        f.set_params(fine, opts);
        f.add_pairs(1, &source, &target, guess);      // (the guess is replaced below: nothing reads it before the ICP loop)
        f.registration_batch = true;
        f.allocate();
        f.phase = 1;
        f.run_all();
        ahead = true;
      } catch (...) {
        // whatever is wrong with the fine stage is reported when its turn comes - after the coarse stage, as in the
        // reference's order (a failing coarse registration is what the caller then sees, :288)
        (void)hipStreamSynchronize(tw->stream);
      }
      st = align_dev(ctx, source, target, guess, coarse, opts, result, info, persistent);
      if (st != S3D_STATUS_OK) return st;
      std::memcpy(guess, result, sizeof guess);
      if (ahead) {
        f.set_guess(0, guess);
        f.phase = 2;
        f.run_all();
        st = f.finish_pair(0, fine, guess, result, info);
        HIPCHK(hipStreamSynchronize(tw->stream));
      } else {
        st = align_dev(ctx, source, target, guess, fine, opts, result, info, persistent);
      }
    } else {
    if (loop) {                                // :286-289
      st = align_dev(ctx, source, target, guess, coarse, opts, result, info, persistent);
      if (st != S3D_STATUS_OK) return st;
      std::memcpy(guess, result, sizeof guess);
    }
    st = align_dev(ctx, source, target, guess, fine, opts, result, info, persistent);   // :292
    }
    if (st != S3D_STATUS_OK) return st;
    mat4d_mul(source_sensor_pose, result, tmp);   // :295
    mat4d_mul(tmp, tinv, relative_pose);
  } catch (...) {
    return fail_current(ctx);
  }
  for (int i = 0; i < 36; ++i) information[i] = 0.0;
  for (int i = 0; i < 6; ++i) information[i * 6 + i] = 1.0 / covariance_scale;  // :296-298
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_remove_outliers_cloud(s3d_context* ctx, const s3d_cloud* in, double radius, unsigned min_neighbors,
                              s3d_cloud** out) try {
  if (!ctx || !in || !out) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud* c = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    remove_outliers_dev(ctx, in, radius, min_neighbors, c, nullptr);
  } catch (...) {
    free_cloud(c);
    delete c;
    return fail_current(ctx);
  }
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_remove_outliers(s3d_context* ctx, const float* xyz, int n, int stride, double radius, unsigned min_neighbors,
                        float* out_xyz, int* n_out) try {
  if (!ctx || !n_out || n < 0 || stride < 3 || (n > 0 && (!xyz || !out_xyz))) return S3D_STATUS_INVALID_ARGUMENT;
  *n_out = 0;
  if (n == 0) return S3D_STATUS_OK;
  s3d_cloud in, kept;
  try {
    ScopedDevice sd(ctx);
    upload_cloud(ctx, xyz, n, stride, &in);
    remove_outliers_dev(ctx, &in, radius, min_neighbors, &kept, nullptr);
    download_packed(ctx, &kept, out_xyz, 3);
    *n_out = kept.n;
    free_cloud(&in);
    free_cloud(&kept);
  } catch (...) {
    free_cloud(&in);
    free_cloud(&kept);
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

// ---- B4: fillGroundPlane (PointCloudSensor.cpp:362-388) ------------------------------------------------
// pcl::RandomSampleConsensus<SampleConsensusModelPlane>::computeModel with the model's fixed seed.  The sample
// stream (boost::mt19937(12345), partial Fisher-Yates, collinearity re-draws) and the stopping rule are scalar
// and stay on the host; scoring - the pass over all points per hypothesis, which is all of the time - runs on
// the device for kPlaneBatch hypotheses per pass.  The hypotheses of a batch are then replayed in order, so
// the result is the one the sequential loop reaches (hypotheses scored beyond its stopping point are ignored).
namespace {

struct PlaneSampler {
  std::mt19937 eng{12345u};
  std::vector<int> perm;
  const float* xyz; int stride;
  PlaneSampler(const float* x, int n, int st) : perm((size_t)n), xyz(x), stride(st) {
    for (int i = 0; i < n; ++i) perm[(size_t)i] = i;
  }
  const float* pt(int i) const { return xyz + (size_t)i * stride; }
  // sac_model.h getSamples: <= 1000 draws until isSampleGood
  bool draw(int s[3]) {
    const uint32_t n = (uint32_t)perm.size();
    for (int check = 0; check < 1000; ++check) {
      for (uint32_t i = 0; i < 3; ++i) std::swap(perm[i], perm[i + ((uint32_t)eng() >> 1) % (n - i)]);
      for (int i = 0; i < 3; ++i) s[i] = perm[(size_t)i];
      const float *p0 = pt(s[0]), *p1 = pt(s[1]), *p2 = pt(s[2]);
      float r[3];
      for (int i = 0; i < 3; ++i) r[i] = (p1[i] - p0[i]) / (p2[i] - p0[i]);
      if (r[0] != r[1] || r[2] != r[1]) return true;
    }
    return false;
  }
  // sac_model_plane.hpp computeModelCoefficients (Eigen's packet order for the two 4-float sums)
  bool model(const int s[3], float mc[4]) const {
    const float *p0 = pt(s[0]), *p1 = pt(s[1]), *p2 = pt(s[2]);
    float a[3], b[3], r[3];
    for (int i = 0; i < 3; ++i) { a[i] = p1[i] - p0[i]; b[i] = p2[i] - p0[i]; r[i] = a[i] / b[i]; }
    if (r[0] == r[1] && r[2] == r[1]) return false;
    mc[0] = a[1] * b[2] - a[2] * b[1];
    mc[1] = a[2] * b[0] - a[0] * b[2];
    mc[2] = a[0] * b[1] - a[1] * b[0];
    const float z = (mc[0] * mc[0] + mc[2] * mc[2]) + (mc[1] * mc[1] + 0.0f);
    if (z > 0.f) { const float nrm = std::sqrt(z); mc[0] /= nrm; mc[1] /= nrm; mc[2] /= nrm; }
    mc[3] = -1.0f * ((mc[0] * p0[0] + mc[2] * p0[2]) + (mc[1] * p0[1] + 0.0f));
    return true;
  }
};

void fit_plane_dev(s3d_context* ctx, const float* xyz, int n, int stride, double threshold, int max_iterations,
                   double probability, s3d_plane_fit* out) {
  std::memset(out, 0, sizeof *out);
  if (n < 3) return;
  s3d_cloud in;
  int* d_counts = nullptr;
  try {
    upload_cloud(ctx, xyz, n, stride, &in);
    HIPCHK(hipMalloc((void**)&d_counts, sizeof(int) * kPlaneBatch));
    PlaneSampler sampler(xyz, n, stride);
    const float thr = (float)threshold;
    const double log_probability = std::log(1.0 - probability), one_over_indices = 1.0 / (double)n;
    const int max_skip = max_iterations * 10;
    const int blocks = std::max(1, std::min(cdiv(n, kBlock), 2048));
    int iters = 0, best = -0x7FFFFFFF, skipped = 0;
    double k = 1.0;
    bool done = false;
    while (!done) {
      // the next events of the sequential loop: a hypothesis to score, a degenerate sample (skip), or no sample
      struct Event { int kind; int h; };   // 0 hypothesis h of the batch, 1 skipped, 2 no sample could be drawn
      std::vector<Event> events;
      PlaneBatch B;
      int nh = 0, batch_skips = 0;
      while (nh < kPlaneBatch && skipped + batch_skips < max_skip) {
        int s[3];
        if (!sampler.draw(s)) { events.push_back({2, 0}); break; }
        float mc[4];
        if (!sampler.model(s, mc)) { events.push_back({1, 0}); ++batch_skips; continue; }
        B.pl[nh] = make_float4(mc[0], mc[1], mc[2], mc[3]);
        events.push_back({0, nh});
        ++nh;
      }
      for (int h = nh; h < kPlaneBatch; ++h) B.pl[h] = make_float4(0.f, 0.f, 0.f, 3.0e38f);
      int counts[kPlaneBatch] = {};
      if (nh > 0) {
        HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(int) * kPlaneBatch, ctx->stream));
        s3d_plane_count_kernel<<<blocks, kBlock, 0, ctx->stream>>>(in.d, n, B, thr, d_counts);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        out->hypotheses_scored += nh;
      }
      if (events.empty()) break;   // max_skip reached
      for (const Event& e : events) {
        if (!((double)iters < k && skipped < max_skip)) { done = true; break; }
        if (e.kind == 2) { done = true; break; }
        if (e.kind == 1) { ++skipped; continue; }
        const int cnt = counts[e.h];
        if (cnt > best) {
          best = cnt;
          out->found = 1;
          out->coefficients[0] = B.pl[e.h].x; out->coefficients[1] = B.pl[e.h].y;
          out->coefficients[2] = B.pl[e.h].z; out->coefficients[3] = B.pl[e.h].w;
          const double w = (double)best * one_over_indices;
          double p_no = 1.0 - std::pow(w, 3.0);
          p_no = std::max(std::numeric_limits<double>::epsilon(), p_no);
          p_no = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no);
          k = log_probability / std::log(p_no);
        }
        ++iters;
        if (iters > max_iterations) { done = true; break; }
      }
      if (!((double)iters < k && skipped < max_skip)) done = true;
    }
    out->n_inliers = out->found ? best : 0;
    out->iterations = iters;
    (void)hipFree(d_counts);
    free_cloud(&in);
  } catch (...) {
    if (d_counts) (void)hipFree(d_counts);
    free_cloud(&in);
    throw;
  }
}

// the ring points of PointCloudSensor.cpp:370-387 (Eigen::Hyperplane::projection, AngleAxis::toRotationMatrix)
int fill_ground_points(const float coeffs[4], double radius, double map_resolution, float* out, int cap) {
  const double n[3] = {(double)coeffs[0], (double)coeffs[1], (double)coeffs[2]}, d = (double)coeffs[3];
  if (!(map_resolution > 0)) return 0;
  const double angle_inc = map_resolution / radius;
  const double two_pi = 2 * 3.141592654;   // the reference's own macro: #define PI 3.141592654 (PointCloudSensor.cpp:48, :376)
  int m = 0;
  for (double r = map_resolution; r <= radius; r += map_resolution) {
    const double sd = (n[0] * r + n[1] * 0.0 + n[2] * 0.0) + d;
    const double sp[3] = {r - sd * n[0], 0.0 - sd * n[1], 0.0 - sd * n[2]};
    for (double angle = 0; angle < two_pi; angle += angle_inc) {
      const double sn = std::sin(angle), c = std::cos(angle);
      const double sa[3] = {sn * n[0], sn * n[1], sn * n[2]};
      const double ca[3] = {(1.0 - c) * n[0], (1.0 - c) * n[1], (1.0 - c) * n[2]};
      double R[3][3], t;
      t = ca[0] * n[1]; R[0][1] = t - sa[2]; R[1][0] = t + sa[2];
      t = ca[0] * n[2]; R[0][2] = t + sa[1]; R[2][0] = t - sa[1];
      t = ca[1] * n[2]; R[1][2] = t - sa[0]; R[2][1] = t + sa[0];
      for (int i = 0; i < 3; ++i) R[i][i] = ca[i] * n[i] + c;
      if (m < cap)
        for (int i = 0; i < 3; ++i) out[(size_t)m * 3 + i] = (float)(R[i][0] * sp[0] + R[i][1] * sp[1] + R[i][2] * sp[2]);
      if (m == 0x7FFFFFFF) return m;
      ++m;
    }
  }
  return m;
}

}  // namespace

int s3d_fit_plane(s3d_context* ctx, const float* xyz, int n, int stride, double threshold, int max_iterations,
                  double probability, s3d_plane_fit* out) try {
  if (!ctx || !out || n < 0 || stride < 3 || (n > 0 && !xyz) || !(threshold >= 0) || max_iterations < 0 ||
      !(probability > 0 && probability < 1))
    return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    fit_plane_dev(ctx, xyz, n, stride, threshold, max_iterations, probability, out);
  } catch (...) {
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_fill_ground_plane(s3d_context* ctx, const float* xyz, int n, int stride, double radius, double map_resolution,
                          float* out_xyz, int out_capacity, int* n_out, s3d_plane_fit* fit) try {
  if (!ctx || !n_out || out_capacity < 0 || (out_capacity > 0 && !out_xyz) || !(radius > 0))
    return S3D_STATUS_INVALID_ARGUMENT;
  *n_out = 0;
  s3d_plane_fit f;
  // PointCloudSensor.cpp:366: setDistanceThreshold(.01); PCL defaults: 1000 iterations, probability 0.99
  const int st = s3d_fit_plane(ctx, xyz, n, stride, 0.01, 1000, 0.99, &f);
  if (fit) *fit = f;
  if (st != S3D_STATUS_OK) return st;
  if (!f.found) return S3D_STATUS_TOO_FEW_POINTS;
  *n_out = fill_ground_points(f.coefficients, radius, map_resolution, out_xyz, out_capacity);
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_voxel_downsample_cloud(s3d_context* ctx, const s3d_cloud* in, double leaf_size, s3d_cloud** out) try {
  if (!ctx || !in || !out) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud* c = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    voxel_dev(ctx, in, leaf_size, c, nullptr);
  } catch (...) {
    free_cloud(c);
    delete c;
    return fail_current(ctx);
  }
  *out = c;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_build_map(s3d_context* ctx, int n_clouds, s3d_cloud* const* clouds, const double* poses, double outlier_radius,
                  unsigned outlier_neighbors, double map_resolution, s3d_cloud** out_map) try {
  if (!ctx || !out_map || n_clouds < 0 || (n_clouds > 0 && (!clouds || !poses))) return S3D_STATUS_INVALID_ARGUMENT;
  for (int i = 0; i < n_clouds; ++i)
    if (!clouds[i]) return S3D_STATUS_INVALID_ARGUMENT;
  s3d_cloud accu, kept;
  s3d_cloud* map = new s3d_cloud();
  try {
    ScopedDevice sd(ctx);
    s3d_map_profile mp{};
    const auto t0 = std::chrono::steady_clock::now();
    {
      StageTimer tm(ctx);
      tm.start();
      accumulate_dev(ctx, n_clouds, clouds, poses, nullptr, &accu);    // PointCloudSensor.cpp:305
      mp.accumulate_ms = tm.stop();
    }
    mp.n_accumulated = accu.n;
    remove_outliers_dev(ctx, &accu, outlier_radius, outlier_neighbors, &kept, &mp);   // :308
    free_cloud(&accu);
    mp.n_kept = kept.n;
    voxel_dev(ctx, &kept, map_resolution, map, &mp);                   // :309
    free_cloud(&kept);
    mp.n_map = map->n;
    mp.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->map_prof = mp;
  } catch (...) {
    free_cloud(&accu);
    free_cloud(&kept);
    free_cloud(map);
    delete map;
    return fail_current(ctx);
  }
  *out_map = map;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

int s3d_last_map_profile(const s3d_context* ctx, s3d_map_profile* out) try {
  if (!ctx || !out) return S3D_STATUS_INVALID_ARGUMENT;
  *out = ctx->map_prof;
  return S3D_STATUS_OK;
} catch (...) { return fail_current(nullptr); }

int s3d_profile_nn_kernel(s3d_context* ctx, int n_pairs, s3d_cloud* const* sources, s3d_cloud* const* targets,
                          const double* guesses, const s3d_reg_params* params, int reps, double* avg_ms,
                          long long* n_queries, long long* n_targets) try {
  if (!ctx || n_pairs <= 0 || !sources || !targets || !guesses || !params || reps <= 0 || !avg_ms)
    return S3D_STATUS_INVALID_ARGUMENT;
  try {
    ScopedDevice sd(ctx);
    Batch b;
    b.ctx = ctx;
    b.set_params(params, nullptr);
    b.add_pairs(n_pairs, sources, targets, guesses);
    b.allocate();
    b.stage_voxel();
    b.stage_grid();
    k_pair_init<<<cdiv(n_pairs, 64), 64, 0, ctx->stream>>>(b.d_pairs(), n_pairs, (int*)ctx->n_active.p);
    const float max_d = (float)(b.rp.max_corr * 1.0001);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    double total_ms = 0.0;
    for (int r = -1; r < reps; ++r) {   // r = -1: warm-up
      // a FIRST pass every time: no radius hints, no re-validation bounds
      HIPCHK(hipMemsetAsync(ctx->corr_d2.p, 0xFF, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
      HIPCHK(hipMemsetAsync(ctx->corr_lb.p, 0, 4 * std::max<size_t>(b.total_corr, 4), ctx->stream));
      HIPCHK(hipEventRecord(e0, ctx->stream));
      b.launch_nn(0, max_d);
      HIPCHK(hipEventRecord(e1, ctx->stream));
      HIPCHK(hipEventSynchronize(e1));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) total_ms += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const float ms = (float)total_ms;
    b.download();
    long long nq = 0, nt = 0;
    for (const PairDev& P : b.h_pairs) { nq += b.h_slots[P.slot_t].n; nt += b.h_slots[P.slot_s].n; }
    *avg_ms = (double)ms / reps;
    if (n_queries) *n_queries = nq;
    if (n_targets) *n_targets = nt;
  } catch (...) {
    return fail_current(ctx);
  }
  return S3D_STATUS_OK;
} catch (...) { return fail_current(ctx); }

}  // extern "C"

#include "s3d_sweep.h"
#include "s3d_candidates.h"
