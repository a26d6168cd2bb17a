// s3d_kernels.h — HIP kernels (gfx950 / CDNA4, wave64) of the registration path.
// Included once by s3d_api.hip.  Kernel ids K1..K8 follow SURVEY.md §8a.
//
// Launch geometry: every per-cloud kernel runs on a 2-D grid (chunk, slot) and every
// per-pair kernel on (chunk, pair); the real element counts live in device memory
// (the host never learns N' after the voxel filter), so blocks whose chunk lies beyond
// the device-side count exit at once.  Nothing in the per-iteration loop syncs with
// the host.
//
// Reductions are deterministic: wave shuffle tree -> LDS -> fixed-order sum over
// blocks in the controller kernel.  No float atomics anywhere; the only atomics are
// integer min/max (bbox, order-independent) and the active-pair counter.
#pragma once

#include <hip/hip_runtime.h>

#include "s3d_core.h"

namespace s3d {

constexpr int kBlock = 256;        // 4 waves
constexpr int kWave = 64;
constexpr int kSortTile = 4096;    // elements per block and pass in the radix sort (16 per thread: digit runs of a
                                   // tile are then ~64 B long; 1024 -> 4096 took 1.3 ms off the 256-pair step)
#ifndef S3D_ACCUM_VB
#define S3D_ACCUM_VB 32
#endif
constexpr int kAccumVB = S3D_ACCUM_VB;   // VIRTUAL blocks per pair in the accumulate / fitness kernels: the unit of the
                                   // fixed summation tree (block_reduce_store_fixed); a launch runs them on 1..32 real blocks.
                                   // Round 4: 64 -> 32.  Every virtual block ends in a 73-value butterfly over its 256
                                   // virtual threads, and at 100 k points a virtual block of 64 folds only six
                                   // correspondences per thread before it: a quarter of the kernel's instructions.
                                   // 256 pairs: 6.3 -> 5.65 ms per step (16: 5.3); the price is the lone pair, whose
                                   // accumulate launch has 32 blocks instead of 64: 1.43 -> 1.48 ms (16: 1.59)
constexpr uint32_t kInvalidKey = 0xFFFFFFFFu;

// ------------------------------------------------------------------ device-side records

struct SlotDev {          // one cloud of the batch
  const float4* raw;      // uploaded points (x, y, z, *)
  int   n_raw;
  int   off;              // offset of this slot in the per-point work arrays (capacity n_raw)
  int   cell_off;         // offset of this slot's cell_start[] (capacity cell_cap + 1)
  int   cell_cap;
  int   want_normals;     // K4 runs on this cloud (point-to-plane: only the searched side of a pair needs normals)
  // device-computed (from `n` to the end: what the cross-call pre-pass cache keeps and restores, s3d_api.hip)
  int   n;                // points after the voxel filter
  int   n_sort;           // element count of the sort in flight
  unsigned int bb[6];     // bbox as order-preserving uint (min xyz, max xyz), atomics
  VoxelParams vp;
  GridParams  g;
  FusedGrid   fz;         // the fused pre-pass (s3d_core.h "K2 + K3 in one sort"): ok = 1 served, < 0 the host must re-run
};

struct PairDev {          // one align() job
  int   slot_s, slot_t;   // slam3d source (kd-tree side) / target (query side) slots
  int   corr_off;         // offset into corr arrays (capacity n_raw of slot_t)
  int   active, converged, iterations, correspondences;
  int   inner_total, evals_total;
  Mat4f guess, T, prev, final_T;
  Mat4f T_nn;             // transformation_ the last correspondence pass ran with (see nn_still_nearest)
  double fitness;
  int   fit_count;
  int   pad;
};

struct RunParams {        // per batch, passed by value
  int    algorithm;       // S3D_ALG_*
  int    k;               // correspondence_randomness
  int    max_iterations, max_inner, force_iterations;
  double max_corr, dist_threshold;   // distance and its square
  double rotation_epsilon, transformation_epsilon;
  double fit_range;       // getFitnessScore max_range (compared with SQUARED distances)
  double gicp_epsilon;
  float  leaf;            // voxel leaf (<= 0: no filter)
  float  h0;              // wanted grid cell edge
};

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ unsigned int f2ord(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// a value known to be identical in every lane, moved to scalar registers
__device__ __forceinline__ double wave_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}

// exclusive prefix over the 256 threads of a block of a 0/1 flag; returns also the block total
// unit normals, and the matched point + normal that travel with each correspondence, are stored as xyz only
// (12-byte records, dwordx3 accesses): K6 and the re-validating NN passes stream them
#ifndef S3D_CORR_PACKED
#define S3D_CORR_PACKED 1
#endif
#if S3D_CORR_PACKED
struct CorrVec { float x, y, z; };
__device__ __forceinline__ CorrVec corr_vec(const float4& v) { CorrVec c; c.x = v.x; c.y = v.y; c.z = v.z; return c; }
#else
typedef float4 CorrVec;
__device__ __forceinline__ CorrVec corr_vec(const float4& v) { return v; }
#endif

// sum of n block partials (src[b * stride]) in ascending b: the loads are issued eight at a time, the additions
// keep their order (a plain loop waits out one L2 round trip per partial: 32 of them cost the single-pair
// controller ~20 us per outer iteration)
__device__ __forceinline__ double ordered_partial_sum(const double* __restrict__ src, int n, size_t stride) {
  double v = 0.0;
  int b = 0;
  for (; b + 8 <= n; b += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(b + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; b < n; ++b) v += src[(size_t)b * stride];
  return v;
}

__device__ __forceinline__ int block_excl_flag(bool flag, int* total, int* lds4 /*4 ints*/) {
  const unsigned long long m = __ballot(flag);
  const int lane = lane_id(), w = wave_id();
  const int inwave = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) lds4[w] = __popcll(m);
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i) {
    const int c = lds4[i];
    if (i < w) before += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return before + inwave;
}

// XCD-aware block -> (pair, chunk) map shared by the NN kernels: consecutive block ids are dealt
// round-robin to the 8 XCDs, so every XCD gets whole pairs (pair % 8 == block % 8) and a pair's cell
// table and points stay in ONE 4 MiB L2 instead of being pulled into all eight.  Speed only: any
// bijective map is correct, and nothing depends on the placement actually happening.
__device__ __forceinline__ void nn_block_map(int chunks_per_pair, int npairs, int* pair, int* chunk) {
  const int b = blockIdx.x;
  if (npairs >= 8) {
    const int xcd = b & 7, slot = b >> 3;
    *pair = (slot / chunks_per_pair) * 8 + xcd;
    *chunk = slot % chunks_per_pair;
  } else {
    *pair = b / chunks_per_pair;
    *chunk = b % chunks_per_pair;
  }
}

// ------------------------------------------------------------------ K1: bbox + voxel keys

__global__ void __launch_bounds__(kBlock) k_slot_reset_bbox(SlotDev* slots) {
  SlotDev& s = slots[blockIdx.x];
  if (threadIdx.x < 3) s.bb[threadIdx.x] = 0xFFFFFFFFu;
  else if (threadIdx.x < 6) s.bb[threadIdx.x] = 0u;
}

// per-thread ordered-uint min / max -> the slot's bbox words: wave shuffle, LDS, then one candidate per block and axis,
// and only if it would move the bound (a lone 10^7-point slot otherwise queues ~2*10^5 atomics on the same six
// words: 2.5 ms).  Every thread of the block must call it.
template <bool ATOMIC = true>
__device__ __forceinline__ void block_bbox_merge(unsigned int (&mn)[3], unsigned int (&mx)[3], unsigned int* bb) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = min(mn[a], (unsigned int)__shfl_down((int)mn[a], o, kWave));
      mx[a] = max(mx[a], (unsigned int)__shfl_down((int)mx[a], o, kWave));
    }
  }
  __shared__ unsigned int red[kBlock / kWave][6];
  if (lane_id() == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[wave_id()][a] = mn[a]; red[wave_id()][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    unsigned int v = red[0][a];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) v = a < 3 ? min(v, red[w][a]) : max(v, red[w][a]);
    if (!ATOMIC) { bb[a] = v; return; }   // (bb: this block's own six words)
    const unsigned int cur = __atomic_load_n(&bb[a], __ATOMIC_RELAXED);
    if (a < 3) { if (v < cur) atomicMin(&bb[a], v); }
    else       { if (v > cur) atomicMax(&bb[a], v); }
  }
}

// which = 0: raw points (n_raw); 1: filtered points (n)
// zero_digit_tot / zero_counters (may be null): what the stages after this first kernel of the pre-pass want cleared - the
// digit totals of the one-sweep sort (kSortPlaces << kSortMaxBits words per slot) and the five K4 list counters - so
// that a lone registration does not pay two more 4-us fill launches
template <int WHICH>
__global__ void __launch_bounds__(kBlock) k_bbox(SlotDev* slots, const float4* __restrict__ filt,
                                                  uint32_t* __restrict__ zero_digit_tot = nullptr,
                                                  int* __restrict__ zero_counters = nullptr) {
  if (blockIdx.x == 0) {
    if (zero_digit_tot)
      for (int d = threadIdx.x; d < (4 << 10); d += kBlock) zero_digit_tot[(size_t)blockIdx.y * (4 << 10) + d] = 0u;
    if (zero_counters && blockIdx.y == 0 && threadIdx.x < 5) zero_counters[threadIdx.x] = 0;
  }
  SlotDev& s = slots[blockIdx.y];
  const int n = WHICH == 0 ? s.n_raw : s.n;
  const float4* __restrict__ p = WHICH == 0 ? s.raw : filt + s.off;
  const int base = blockIdx.x * (kBlock * 4);
  if (base >= n) return;
  unsigned int mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  // (issuing the four loads before the first use was measured: 0.21 -> 0.17 ms on 512 clouds of 100 k points, but
  // 0.10 -> 0.24 ms per call on ONE cloud of 10 M - the blocks then reach the merge of the six bound words together and
  // queue on them; the map build matters more)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = base + r * kBlock + threadIdx.x;
    if (i < n) {
      const float4 v = p[i];
      if (finite3(v.x, v.y, v.z)) {
        const unsigned int a = f2ord(v.x), b = f2ord(v.y), c = f2ord(v.z);
        mn[0] = min(mn[0], a); mn[1] = min(mn[1], b); mn[2] = min(mn[2], c);
        mx[0] = max(mx[0], a); mx[1] = max(mx[1], b); mx[2] = max(mx[2], c);
      }
    }
  }
  block_bbox_merge(mn, mx, s.bb);
}

__global__ void k_voxel_params(SlotDev* slots, RunParams rp, int nslots) {
  const int si = blockIdx.x * blockDim.x + threadIdx.x;
  if (si >= nslots) return;
  SlotDev& s = slots[si];
  if (s.n_raw <= 0 || s.bb[0] == 0xFFFFFFFFu) {  // empty or all non-finite
    s.vp.inv_leaf = 1.f; s.vp.passthrough = 1;
    for (int a = 0; a < 3; ++a) { s.vp.min_b[a] = 0; s.vp.div_b[a] = 1; }
  } else {
    float mn[3], mx[3];
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(s.bb[a]); mx[a] = ord2f(s.bb[3 + a]); }
    s.vp = voxel_params_from_bbox(mn, mx, rp.leaf);
  }
  s.n_sort = s.n_raw;
  s.fz.ok = 0;
}

// the fused pre-pass, after k_voxel_params: the search grid laid over the voxel lattice (fused_grid_from_voxels) - the
// one sort then orders the raw points by (cell, voxel).  An empty slot: a one-cell grid, nothing to sort (every key is
// invalid), nothing that could go wrong; a slot the scheme cannot serve: ok < 0, a one-cell grid, every key invalid.
// (A kernel of its own: inlined into k_voxel_params the two bodies crash hipcc 7.2's instruction selection.)
__global__ void k_fused_grid_params(SlotDev* slots, int nslots) {
  const int si = blockIdx.x * blockDim.x + threadIdx.x;
  if (si >= nslots) return;
  SlotDev& s = slots[si];
  VoxelParams v = s.vp;
  if (s.n_raw <= 0 || s.bb[0] == 0xFFFFFFFFu) v.passthrough = 0;
  GridParams g;
  FusedGrid f;
  fused_grid_from_voxels(v, s.cell_cap, g, f);
  s.g = g;
  s.fz = f;
}

// leaf <= 0: the filtered cloud is the raw cloud (PointCloudSensor.cpp:125-131)
__global__ void __launch_bounds__(kBlock) k_copy_raw(SlotDev* slots, float4* __restrict__ filt) {
  SlotDev& s = slots[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) s.n = s.n_raw;
  if (i >= s.n_raw) return;
  float4 v = s.raw[i];
  v.w = 1.f;
  filt[s.off + i] = v;
}

// Many device-to-device copies as ONE launch (the pre-pass cache moves five arrays per cloud: as hipMemcpyAsync calls
// they cost a launch each, which is what a one-new-scan call is bound by).  blockIdx.y = copy, grid-stride in x.
struct CopyDesc { const void* src; void* dst; unsigned long long bytes; };   // bytes: a multiple of 4
__global__ void __launch_bounds__(kBlock) k_copy_many(const CopyDesc* __restrict__ descs) {
  const CopyDesc d = descs[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * kBlock, t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if ((((size_t)d.src | (size_t)d.dst | (size_t)d.bytes) & 15) == 0) {
    const uint4* __restrict__ s4 = (const uint4*)d.src;
    uint4* __restrict__ d4 = (uint4*)d.dst;
    for (size_t i = t; i < d.bytes / 16; i += stride) d4[i] = s4[i];
  } else {
    const uint32_t* __restrict__ s1 = (const uint32_t*)d.src;
    uint32_t* __restrict__ d1 = (uint32_t*)d.dst;
    for (size_t i = t; i < d.bytes / 4; i += stride) d1[i] = s1[i];
  }
}

// ------------------------------------------------------------------ K2a: segmented stable LSD radix sort
// (key, value) pairs, one segment per slot, BITS-bit digits (round 4: 8, 9 or 10 - the pass count follows the key
// range instead of being 4 + 3 passes of 8 bits whatever the keys are: the voxel keys of the benchmark are 30 bits =
// 3 x 10, its grid-cell ids 18 bits = 2 x 9).  NB = 2^BITS bins.  The tile histograms `counts` come in two layouts:
//   nb_max <= kSortTileMajor (clouds up to 262 k points): counts[(slot * nb_max + tile) * NB + digit] - the NB
//     counters a histogram kernel writes and a scatter tile reads are contiguous, and the scan over the tiles of a
//     digit is a loop of a thread per digit (k_sort_scan_tiles);
//   larger slots (map building: ~10^4 tiles): counts[(slot * NB + digit) * nb_max + tile], a row per digit that a
//     wave scans 256 tiles at a time (k_sort_scan_rows).
// Measured (256 pairs x 100 k points, scatter kernel per pass): 8 bits 0.208 ms, 9 bits 0.224 ms, 10 bits 0.380 ms - a
// tile of 4096 elements leaves digit runs of 16 / 8 / 4 elements, and below 32 bytes a run no longer fills the sectors
// it is written to (and the kernel's LDS allows three blocks per compute unit instead of four).  So a digit gets a
// ninth bit where that removes a pass (grid-cell ids of up to 18 bits: 2 x 9 instead of 3 x 8) and a tenth only where
// 9 bits do not (19-20 bits); the 30-bit voxel keys of the benchmark stay at 4 x 8 (3 x 10 + a gated fourth pass for
// wider keys was built: voxel stage 2.35 -> 2.71 ms).
constexpr int kSortTileMajor = 64;
constexpr int kSortMaxBits = 10;
template <int NB>
__device__ __forceinline__ size_t sort_count_index(int slot, int digit, int tile, int nb_max) {
  return nb_max <= kSortTileMajor ? ((size_t)slot * nb_max + tile) * NB + digit
                                  : ((size_t)slot * NB + digit) * nb_max + tile;
}
// exclusive scan over the NB = D * 256 digits of a block, thread t holding the values of digits t * D ... t * D + D - 1
// (ex[k]: sum of all digits before t * D + k).  Two independent scans share the barrier.  wave_tot: [2][kBlock / kWave].
template <int D>
__device__ __forceinline__ void block_excl_scan2(const unsigned int (&a)[D], const unsigned int (&b)[D], unsigned int (&exa)[D],
                                                 unsigned int (&exb)[D], unsigned int (*wave_tot)[kBlock / kWave]) {
  const int lane = lane_id(), w = wave_id();
  unsigned int sa = 0, sb = 0;
#pragma unroll
  for (int k = 0; k < D; ++k) { sa += a[k]; sb += b[k]; }
  unsigned int ia = sa, ib = sb;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const unsigned int ta = __shfl_up(ia, o, kWave), tb = __shfl_up(ib, o, kWave);
    if (lane >= o) { ia += ta; ib += tb; }
  }
  if (lane == kWave - 1) { wave_tot[0][w] = ia; wave_tot[1][w] = ib; }
  __syncthreads();
  unsigned int ba = ia - sa, bb = ib - sb;
#pragma unroll
  for (int ww = 0; ww < kBlock / kWave; ++ww) { ba += ww < w ? wave_tot[0][ww] : 0u; bb += ww < w ? wave_tot[1][ww] : 0u; }
#pragma unroll
  for (int k = 0; k < D; ++k) { exa[k] = ba; exb[k] = bb; ba += a[k]; bb += b[k]; }
}

template <int BITS>
__global__ void __launch_bounds__(kBlock) k_sort_hist(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ counts, int shift, int nb_max) {
  constexpr int NB = 1 << BITS;
  __shared__ unsigned int hist[NB];
  const SlotDev& s = slots[blockIdx.y];
  const int n = s.n_sort;
  const int nb = (n + kSortTile - 1) / kSortTile;
  if ((int)blockIdx.x >= nb) return;
  for (int d = threadIdx.x; d < NB; d += kBlock) hist[d] = 0;
  __syncthreads();
  const int base = blockIdx.x * kSortTile;
#pragma unroll
  for (int r = 0; r < kSortTile / kBlock; ++r) {
    const int i = base + r * kBlock + threadIdx.x;
    if (i < n) atomicAdd(&hist[(keys[s.off + i] >> shift) & (uint32_t)(NB - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += kBlock) counts[sort_count_index<NB>(blockIdx.y, d, blockIdx.x, nb_max)] = hist[d];
}

// The keys of a sort and the tile histogram of its FIRST pass in one kernel (WHICH = 0: PCL voxel keys of the raw
// points, 1: grid-cell ids of the filtered points): the keys are produced a 4096-element tile at a time and counted
// as they are written, instead of being read back by k_sort_hist (one pass over the keys and one launch less per
// sort: 0.2 ms of the 256-pair step).
// sweep_passes > 0 (the one-sweep sort below): the digit totals of that many BITS-bit places go to digit_tot_all instead,
// and `counts` is the look-back state, whose row of this tile is zeroed.
constexpr int kSortPlaces = 4;
static_assert(kSortPlaces == 4 && kSortMaxBits == 10, "k_bbox clears (4 << 10) digit totals per slot");
template <int WHICH, int BITS>
__global__ void __launch_bounds__(kBlock) k_keys_hist(const SlotDev* __restrict__ slots, const float4* __restrict__ filt,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ counts, int nb_max, int sweep_passes,
                                                       uint32_t* __restrict__ digit_tot_all) {
  constexpr int NB = 1 << BITS;
  __shared__ unsigned int hist[kSortPlaces][NB];
  const SlotDev& s = slots[blockIdx.y];
  const int n = WHICH == 1 ? s.n : s.n_raw;
  const int nb = (n + kSortTile - 1) / kSortTile;
  if ((int)blockIdx.x >= nb) return;
  const int places = sweep_passes > 0 ? sweep_passes : 1;
  for (int p = 0; p < places; ++p)
    for (int d = threadIdx.x; d < NB; d += kBlock) hist[p][d] = 0;
  __syncthreads();
  const int base = blockIdx.x * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortTile / kBlock; ++r) {
    const int i = base + r * kBlock + threadIdx.x;
    if (i < n) {
      uint32_t key;
      if (WHICH == 0) {
        const float4 v = s.raw[i];
        key = kInvalidKey;
        if (finite3(v.x, v.y, v.z)) key = s.vp.passthrough ? (uint32_t)i : voxel_key(s.vp, v.x, v.y, v.z);
      } else if (WHICH == 2) {   // the fused pre-pass: (cell, voxel) keys of the raw points
        const float4 v = s.raw[i];
        key = kInvalidKey;
        if (s.fz.ok > 0 && finite3(v.x, v.y, v.z)) key = fused_key(s.vp, s.g, s.fz, v.x, v.y, v.z);
      } else {
        const float4 p = filt[s.off + i];
        key = (uint32_t)grid_cell_of_point(s.g, p.x, p.y, p.z);
      }
      keys[s.off + i] = key;
      // (no values are written: they are the identity, which the first pass of the sort supplies itself - 4 bytes per
      // element less to write here and to read there)
      atomicAdd(&hist[0][key & (uint32_t)(NB - 1)], 1u);
      for (int p = 1; p < sweep_passes; ++p) atomicAdd(&hist[p][(key >> (BITS * p)) & (uint32_t)(NB - 1)], 1u);
    }
  }
  __syncthreads();
  if (sweep_passes > 0) {
    for (int p = 0; p < sweep_passes; ++p)
      for (int d = threadIdx.x; d < NB; d += kBlock) {
        const unsigned int v = hist[p][d];
        if (v) atomicAdd(&digit_tot_all[((size_t)blockIdx.y * kSortPlaces + p) * NB + d], v);
      }
    for (int d = threadIdx.x; d < NB; d += kBlock) counts[((size_t)blockIdx.y * nb_max + blockIdx.x) * NB + d] = 0u;
    return;
  }
  for (int d = threadIdx.x; d < NB; d += kBlock) counts[sort_count_index<NB>(blockIdx.y, d, blockIdx.x, nb_max)] = hist[0][d];
}

// Offsets of one pass in two steps.  (1) one WAVE per (slot, digit) row of tile counts: exclusive scan of the row
// in place, 64 tiles per step (four steps' loads in flight), row total -> digit_tot.  (2) the exclusive scan of the
// slot's NB row totals is redone by every scatter tile (k_sort_scatter), which adds it to its tile's row offset.  A lone 10^7-point
// slot (map building) has ~10^4 tiles per row: the first version walked each row with one thread and took 5 ms
// per pass there.
template <int BITS>
__global__ void __launch_bounds__(kBlock) k_sort_scan_rows(const SlotDev* __restrict__ slots, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ digit_tot, int nb_max) {
  constexpr int NB = 1 << BITS;
  const SlotDev& s = slots[blockIdx.y];
  const int nb = (s.n_sort + kSortTile - 1) / kSortTile;
  const int digit = blockIdx.x * (kBlock / kWave) + wave_id();
  const int lane = lane_id();
  uint32_t* __restrict__ c = counts + ((size_t)blockIdx.y * NB + digit) * nb_max;
  unsigned int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 4 * kWave) {
    unsigned int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = b0 + j * kWave + lane;
      v[j] = b < nb ? c[b] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned int incl = v[j];
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const unsigned int t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
      }
      const int b = b0 + j * kWave + lane;
      if (b < nb) c[b] = carry + incl - v[j];
      carry += __shfl(incl, kWave - 1, kWave);
    }
  }
  if (lane == 0) digit_tot[(size_t)blockIdx.y * NB + digit] = carry;
}

// nb_max <= kSortTileMajor: one block per slot, a thread per digit walks the tiles (coalesced over the digits); all
// loads of a chunk of 16 tiles are issued before the first addition
template <int BITS>
__global__ void __launch_bounds__(256) k_sort_scan_tiles(const SlotDev* __restrict__ slots, uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ digit_tot, int nb_max) {
  constexpr int NB = 1 << BITS;
  const SlotDev& s = slots[blockIdx.x];
  const int nb = (s.n_sort + kSortTile - 1) / kSortTile;
  for (int d = threadIdx.x; d < NB; d += 256) {
    uint32_t* __restrict__ c = counts + (size_t)blockIdx.x * nb_max * NB + d;
    unsigned int carry = 0;
    for (int t0 = 0; t0 < nb; t0 += 16) {
      unsigned int v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = t0 + j < nb ? c[(size_t)(t0 + j) * NB] : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (t0 + j < nb) c[(size_t)(t0 + j) * NB] = carry;
        carry += v[j];
      }
    }
    digit_tot[(size_t)blockIdx.x * NB + d] = carry;
  }
}

// the ranks of a wave's 16 rounds of 64 elements among the elements of the same digit in the wave's quarter of the
// tile (a wave owns a CONTIGUOUS quarter, so the output order (wave, round, lane) is the input order and no block
// barrier is needed per round): lanes holding the same digit find each other with BITS ballots, the first of them
// bumps the wave's private LDS counter.
template <int BITS, int ROUNDS>
__device__ __forceinline__ void sort_rank_rounds(const uint32_t (&key)[ROUNDS], unsigned int (&rank)[ROUNDS], int base, int n,
                                                 int shift, unsigned short* __restrict__ my_cnt /* [NB] of this wave */) {
  constexpr uint32_t kMask = (1u << BITS) - 1u;
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int i = base + r * kWave + lane;
    const bool act = i < n;
    const unsigned int d = (key[r] >> shift) & kMask;
    unsigned long long peers = __ballot(act);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const unsigned long long m = __ballot(act && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const int in_round = __popcll(peers & ((1ull << lane) - 1ull));
    unsigned int before = 0;
    if (act) before = my_cnt[d];                            // same value for all peers
    __builtin_amdgcn_wave_barrier();
    if (act && in_round == 0) my_cnt[d] = (unsigned short)(before + (unsigned int)__popcll(peers));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    rank[r] = before + (unsigned int)in_round;
  }
}

// Stable scatter of one 4096-element tile.  Two block barriers around the digit scans, two around the LDS staging:
// the tile is first put in digit order in LDS, then written out with consecutive threads on consecutive addresses of
// each digit run (a direct scatter issues 64 unrelated 4-byte stores per wave and array).
template <int BITS, bool IDENT = false>
__global__ void __launch_bounds__(kBlock) k_sort_scatter(const SlotDev* __restrict__ slots,
                                                          const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                          const uint32_t* __restrict__ counts,
                                                          const uint32_t* __restrict__ digit_tot, int shift, int nb_max,
                                                          int nslots) {
  constexpr int NB = 1 << BITS, D = NB / kBlock;
  constexpr uint32_t kMask = (uint32_t)(NB - 1);
  __shared__ unsigned short wave_cnt[kBlock / kWave][NB];   // per wave: elements of each digit seen so far
  __shared__ unsigned short dig_local[NB];
  __shared__ unsigned int dig_global[NB], wave_tot[2][kBlock / kWave];
  __shared__ uint32_t lkey[kSortTile], lval[kSortTile];
  // 1-D grid, slot -> XCD affinity (nn_block_map): the digit runs that the tiles of one cloud write next to each other
  // meet in ONE L2
  int slot_i, tile_i;
  nn_block_map(nb_max, nslots, &slot_i, &tile_i);
  if (slot_i >= nslots) return;
  const SlotDev& s = slots[slot_i];
  const int n = s.n_sort;
  const int nb = (n + kSortTile - 1) / kSortTile;
  if (tile_i >= nb) return;
  const int lane = lane_id(), w = wave_id();
  constexpr int kRounds = kSortTile / kBlock;               // 16
  for (int d = threadIdx.x; d < NB; d += kBlock) {
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave; ++ww) wave_cnt[ww][d] = 0;
  }
  __syncthreads();
  const int base = tile_i * kSortTile + w * (kSortTile / (kBlock / kWave));
  uint32_t key[kRounds], val[kRounds];
  unsigned int rank[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {                       // all loads up front (independent)
    const int i = base + r * kWave + lane;
    key[r] = 0; val[r] = 0;
    if (i < n) { key[r] = keys_in[s.off + i]; val[r] = IDENT ? (uint32_t)i : vals_in[s.off + i]; }   // (IDENT: the values are the identity, not stored)
  }
  sort_rank_rounds<BITS, kRounds>(key, rank, base, n, shift, wave_cnt[w]);
  __syncthreads();
  {
    // digits threadIdx.x * D ... + D - 1: exclusive prefix over the waves (tile-local), the digit's first position
    // inside the tile (exclusive scan of the tile histogram over the digits) and in the output (this tile's row
    // offset + the digit base of the pass = the exclusive scan of the slot's digit totals, redone by every tile
    // instead of a kernel of its own between the row scan and the scatter)
    unsigned int run[D], dtot[D], ex_run[D], ex_tot[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int d = (int)threadIdx.x * D + k;
      unsigned int acc = 0;
#pragma unroll
      for (int ww = 0; ww < kBlock / kWave; ++ww) {
        const unsigned int c = wave_cnt[ww][d];
        wave_cnt[ww][d] = (unsigned short)acc;      // elements of this digit in earlier waves of the tile
        acc += c;
      }
      run[k] = acc;
      dtot[k] = digit_tot[(size_t)slot_i * NB + d];
    }
    block_excl_scan2<D>(run, dtot, ex_run, ex_tot, wave_tot);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int d = (int)threadIdx.x * D + k;
      dig_local[d] = (unsigned short)ex_run[k];
      dig_global[d] = counts[sort_count_index<NB>(slot_i, d, tile_i, nb_max)] + ex_tot[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int i = base + r * kWave + lane;
    if (i < n) {
      const unsigned int d = (key[r] >> shift) & kMask;
      const unsigned int lp = (unsigned int)dig_local[d] + wave_cnt[w][d] + rank[r];
      lkey[lp] = key[r];
      lval[lp] = val[r];
    }
  }
  __syncthreads();
  const int tile_n = min(kSortTile, n - tile_i * kSortTile);
  for (int j = threadIdx.x; j < tile_n; j += kBlock) {
    const uint32_t kk = lkey[j];
    const unsigned int d = (kk >> shift) & kMask;
    const unsigned int pos = dig_global[d] + ((unsigned int)j - dig_local[d]);
    keys_out[s.off + pos] = kk;
    vals_out[s.off + pos] = lval[j];
  }
}

// ------------------------------------------------------------------ K2a', round 3: the same sort in ONE sweep per pass
// (decoupled look-back: Merrill & Garland's chained scan, as in Adinets & Merrill's Onesweep).  Above, a pass is
// three kernels - tile histograms, their scan over the tiles, the scatter - and the keys are read twice.  Here the
// digit totals of ALL passes of a sort are counted once up front (k_sort_hist_all, or the kernel that produces the
// keys), and a pass is one kernel: a tile ranks its elements, publishes its NB digit counts (AGGREGATE), walks back
// over the tiles before it adding their counts until it meets one that has published its INCLUSIVE prefix, publishes
// its own, and scatters - one read and one write of keys and values per pass.  A state word is (tag << 28 | count):
// tag = 2 * pass + 1 (aggregate) / + 2 (inclusive), so a
// word left by an earlier pass reads as "not there yet"; the rows are zeroed once per sort by the histogram kernel.
// Tiles wait only for tiles with a smaller block index of the same launch (nn_block_map keeps a cloud's tiles in
// ascending block order), which the dispatcher has started before them.
constexpr uint32_t kSweepValMask = 0x0FFFFFFFu;

template <int BITS>
__global__ void __launch_bounds__(kBlock) k_sort_hist_all(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ digit_tot_all, uint32_t* __restrict__ state,
                                                           int passes, int nb_max) {
  constexpr int NB = 1 << BITS;
  __shared__ unsigned int hist[kSortPlaces][NB];
  const SlotDev& s = slots[blockIdx.y];
  const int n = s.n_sort;
  const int nb = (n + kSortTile - 1) / kSortTile;
  if ((int)blockIdx.x >= nb) return;
  for (int p = 0; p < passes; ++p)
    for (int d = threadIdx.x; d < NB; d += kBlock) hist[p][d] = 0;
  __syncthreads();
  const int base = blockIdx.x * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortTile / kBlock; ++r) {
    const int i = base + r * kBlock + threadIdx.x;
    if (i < n) {
      const uint32_t key = keys[s.off + i];
      for (int p = 0; p < passes; ++p) atomicAdd(&hist[p][(key >> (BITS * p)) & (uint32_t)(NB - 1)], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < passes; ++p)
    for (int d = threadIdx.x; d < NB; d += kBlock) {
      const unsigned int v = hist[p][d];
      if (v) atomicAdd(&digit_tot_all[((size_t)blockIdx.y * kSortPlaces + p) * NB + d], v);
    }
  for (int d = threadIdx.x; d < NB; d += kBlock) state[((size_t)blockIdx.y * nb_max + blockIdx.x) * NB + d] = 0u;
}

template <int BITS, bool IDENT = false>
__global__ void __launch_bounds__(kBlock) k_sort_onesweep(const SlotDev* __restrict__ slots,
                                                           const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                           uint32_t* __restrict__ state,
                                                           const uint32_t* __restrict__ digit_tot_all, int pass, int nb_max,
                                                           int nslots, int* __restrict__ error_flag) {
  constexpr int NB = 1 << BITS, D = NB / kBlock;
  constexpr uint32_t kMask = (uint32_t)(NB - 1);
  __shared__ unsigned short wave_cnt[kBlock / kWave][NB];   // per wave: elements of each digit seen so far
  __shared__ unsigned short dig_local[NB];
  __shared__ unsigned int dig_global[NB], wave_tot[2][kBlock / kWave];
  __shared__ uint32_t lkey[kSortTile], lval[kSortTile];
  int slot_i, tile_i;
  nn_block_map(nb_max, nslots, &slot_i, &tile_i);
  if (slot_i >= nslots) return;
  const SlotDev& s = slots[slot_i];
  const int n = s.n_sort;
  const int nb = (n + kSortTile - 1) / kSortTile;
  if (tile_i >= nb) return;
  const int shift = BITS * pass;
  const int lane = lane_id(), w = wave_id();
  constexpr int kRounds = kSortTile / kBlock;               // 16
  for (int d = threadIdx.x; d < NB; d += kBlock) {
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave; ++ww) wave_cnt[ww][d] = 0;
  }
  __syncthreads();
  const int base = tile_i * kSortTile + w * (kSortTile / (kBlock / kWave));
  uint32_t key[kRounds], val[kRounds];
  unsigned int rank[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {                       // all loads up front (independent)
    const int i = base + r * kWave + lane;
    key[r] = 0; val[r] = 0;
    if (i < n) { key[r] = keys_in[s.off + i]; val[r] = IDENT ? (uint32_t)i : vals_in[s.off + i]; }   // (IDENT: the values are the identity, not stored)
  }
  sort_rank_rounds<BITS, kRounds>(key, rank, base, n, shift, wave_cnt[w]);
  __syncthreads();
  {
    // digits threadIdx.x * D ...: this tile's counts, published at once; then the look-back over the earlier tiles
    unsigned int run[D], dtot[D], excl[D], ex_run[D], ex_tot[D];
    const uint32_t tag_agg = (uint32_t)(2 * pass + 1) << 28, tag_inc = (uint32_t)(2 * pass + 2) << 28;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int d = (int)threadIdx.x * D + k;
      unsigned int acc = 0;
#pragma unroll
      for (int ww = 0; ww < kBlock / kWave; ++ww) {
        const unsigned int c = wave_cnt[ww][d];
        wave_cnt[ww][d] = (unsigned short)acc;      // elements of this digit in earlier waves of the tile
        acc += c;
      }
      run[k] = acc;
      uint32_t* __restrict__ row = state + ((size_t)slot_i * nb_max) * NB + d;     // [tile * NB]
      __hip_atomic_store(row + (size_t)tile_i * NB, (tile_i == 0 ? tag_inc : tag_agg) | acc, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int d = (int)threadIdx.x * D + k;
      uint32_t* __restrict__ row = state + ((size_t)slot_i * nb_max) * NB + d;
      unsigned int ex = 0;
      int t = tile_i - 1;                    // (tile 0 has published an inclusive prefix: the walk ends there at the latest)
      unsigned int spins = 0;
      bool found = t < 0;
      while (!found) {
        // four earlier tiles per round trip, used in order until an inclusive prefix or a word that is not there yet
        uint32_t wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          wv[u] = __hip_atomic_load(row + (size_t)(t - u >= 0 ? t - u : 0) * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int consumed = 0;
        bool stop = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (stop || t - u < 0) continue;
          const uint32_t tg = wv[u] & ~kSweepValMask;
          if (tg == tag_inc) { ex += wv[u] & kSweepValMask; found = true; stop = true; }
          else if (tg == tag_agg) { ex += wv[u] & kSweepValMask; ++consumed; }
          else stop = true;
        }
        t -= consumed;
        if (!found && consumed == 0) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { if (error_flag) atomicOr(error_flag, 1); break; }   // (never hang the device)
        }
      }
      // (a tile that gave up publishes its prefix all the same - its successors must not wait for it - and the host
      // fails the call in Batch::download(); what this sort leaves in the output is not used)
      if (tile_i != 0)
        __hip_atomic_store(row + (size_t)tile_i * NB, tag_inc | (ex + run[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      excl[k] = ex;
      dtot[k] = digit_tot_all[((size_t)slot_i * kSortPlaces + pass) * NB + d];
    }
    block_excl_scan2<D>(run, dtot, ex_run, ex_tot, wave_tot);
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const int d = (int)threadIdx.x * D + k;
      dig_local[d] = (unsigned short)ex_run[k];
      dig_global[d] = excl[k] + ex_tot[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int i = base + r * kWave + lane;
    if (i < n) {
      const unsigned int d = (key[r] >> shift) & kMask;
      const unsigned int lp = (unsigned int)dig_local[d] + wave_cnt[w][d] + rank[r];
      lkey[lp] = key[r];
      lval[lp] = val[r];
    }
  }
  __syncthreads();
  const int tile_n = min(kSortTile, n - tile_i * kSortTile);
  for (int j = threadIdx.x; j < tile_n; j += kBlock) {
    const uint32_t kk = lkey[j];
    const unsigned int d = (kk >> shift) & kMask;
    const unsigned int pos = dig_global[d] + ((unsigned int)j - dig_local[d]);
    keys_out[s.off + pos] = kk;
    vals_out[s.off + pos] = lval[j];
  }
}

// ------------------------------------------------------------------ K2b: segmented centroid

__device__ __forceinline__ bool voxel_head(const uint32_t* __restrict__ keys, int i) {
  const uint32_t k = keys[i];
  return k != kInvalidKey && (i == 0 || keys[i - 1] != k);
}

// heads per 256-element chunk (what k_centroids' block scan starts from).  A wave takes four chunks, no barrier: as one
// block per chunk with a block reduction the kernel ran at 1.5 TB/s on its 4 bytes per element (0.135 ms at 256 pairs).
constexpr int kHeadsChunksPerBlock = 4 * (kBlock / kWave);
__global__ void __launch_bounds__(kBlock) k_heads_count(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ blockcnt, int nb_max) {
  const SlotDev& s = slots[blockIdx.y];
  const uint32_t* __restrict__ k = keys + s.off;
  const int lane = lane_id();
  const int chunk0 = ((int)blockIdx.x * (kBlock / kWave) + wave_id()) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int chunk = chunk0 + c, base = chunk * kBlock;
    if (base >= s.n_raw) return;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < kBlock / kWave; ++r) {
      const int i = base + r * kWave + lane;
      const bool head = i < s.n_raw && voxel_head(k, i);
      cnt += (int)__popcll(__ballot(head));
    }
    if (lane == 0) blockcnt[(size_t)blockIdx.y * nb_max + chunk] = (uint32_t)cnt;
  }
}

__global__ void __launch_bounds__(kBlock) k_heads_scan(SlotDev* slots, uint32_t* __restrict__ blockcnt, int nb_max) {
  __shared__ unsigned int part[kBlock];
  SlotDev& s = slots[blockIdx.x];
  const int nb = (s.n_raw + kBlock - 1) / kBlock;
  uint32_t* c = blockcnt + (size_t)blockIdx.x * nb_max;
  // each thread owns a contiguous run of blocks
  const int per = (nb + kBlock - 1) / kBlock;
  const int b0 = min((int)threadIdx.x * per, nb), b1 = min(b0 + per, nb);
  unsigned int sum = 0;
  for (int b = b0; b < b1; ++b) sum += c[b];
  part[threadIdx.x] = sum;
  __syncthreads();
  unsigned int v = sum;
  for (int o = 1; o < kBlock; o <<= 1) {
    unsigned int t = (threadIdx.x >= (unsigned)o) ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    v += t;
    part[threadIdx.x] = v;
    __syncthreads();
  }
  unsigned int run = v - sum;
  for (int b = b0; b < b1; ++b) {
    const unsigned int t = c[b];
    c[b] = run;
    run += t;
  }
  if (threadIdx.x == kBlock - 1) s.n = (int)v;
}

// (round 6) the run sums of a 256-element chunk of the sorted raw points, for both centroid kernels.  Until round 6 a head
// walked its run through global memory - key, index, gather: two dependent round trips per point, on a quarter of the
// lanes - which a lidar scan punishes: 0.2 m voxels next to the sensor hold tens of returns each, and a wave lasted as long
// as its longest run (the reference's scans: 37 ps per raw point against 20 on the synthetic cloud, whose voxels hold one
// point).  Here EVERY lane gathers its own point at once (one round trip for the block), the points and keys go through
// LDS, and a head adds its run up from there in the same order - ((0 + p0) + p1) + ... as pcl::VoxelGrid does: the same
// floats bit for bit.  A run that leaves the chunk goes on through global memory as before; a lane whose run started in an
// earlier chunk (its key equals the key before the chunk) gathers nothing.
struct RunSum { float sx, sy, sz; int end; float4 first; uint32_t key; };   // end: global index one past the run
__device__ __forceinline__ bool block_run_sums(const SlotDev& s, const uint32_t* __restrict__ k, const uint32_t* __restrict__ v,
                                               int base, RunSum& out) {
  __shared__ float lx[kBlock], ly[kBlock], lz[kBlock];
  __shared__ uint32_t lk[kBlock];
  const int t = (int)threadIdx.x, i = base + t;
  const bool in = i < s.n_raw;
  const uint32_t key = in ? k[i] : kInvalidKey;
  const uint32_t kb = base > 0 ? k[base - 1] : kInvalidKey;      // (uniform: the key before the chunk)
  const uint32_t idx = in ? v[i] : 0u;
  const bool mine = in && key != kInvalidKey && !(base > 0 && key == kb);
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (mine) p = s.raw[idx];
  lk[t] = key; lx[t] = p.x; ly[t] = p.y; lz[t] = p.z;
  __syncthreads();
  const bool head = mine && (t == 0 || lk[t - 1] != key);
  if (head) {
    float sx = 0.f + p.x, sy = 0.f + p.y, sz = 0.f + p.z;   // (0 + x: the sums start as they always did)
    int j = t + 1;
    for (; j < kBlock && lk[j] == key; ++j) { sx += lx[j]; sy += ly[j]; sz += lz[j]; }
    int jg = base + j;
    if (j == kBlock)
      for (; jg < s.n_raw && k[jg] == key; ++jg) {
        const float4 q = s.raw[v[jg]];
        sx += q.x; sy += q.y; sz += q.z;
      }
    out.sx = sx; out.sy = sy; out.sz = sz; out.end = jg; out.first = p; out.key = key;
  }
  return head;
}

// one thread per sorted element; heads walk their run and emit the centroid
// (pcl::VoxelGrid fourth pass: float sum in order, divided by float count)
__global__ void __launch_bounds__(kBlock) k_centroids(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ vals, const uint32_t* __restrict__ blockcnt,
                                                       float4* __restrict__ filt, unsigned int* __restrict__ blockbb, int nb_max,
                                                       int nslots) {
  __shared__ int lds4[4];
  // (slot, chunk) from a 1-D grid with slot -> XCD affinity (nn_block_map): the gathers raw[v[j]] of one cloud then
  // hit ONE L2 instead of pulling the cloud into all eight
  int slot_i, chunk_i;
  nn_block_map(nb_max, nslots, &slot_i, &chunk_i);
  if (slot_i >= nslots) return;
  const SlotDev& s = slots[slot_i];
  const int base = chunk_i * kBlock;
  if (base >= s.n_raw) return;
  const int i = base + threadIdx.x;
  const uint32_t* __restrict__ k = keys + s.off;
  const uint32_t* __restrict__ v = vals + s.off;
  RunSum rs;
  const bool head = block_run_sums(s, k, v, base, rs);
  int total;
  const int pos = block_excl_flag(head, &total, lds4) + (int)blockcnt[(size_t)slot_i * nb_max + chunk_i];
  // the bounding box of the centroids (what the search grid is laid over) is gathered here, where they are written (a
  // separate pass over the filtered cloud was 0.22 ms of the 256-pair step): six words per block, no atomics - with
  // a block per 256 points, merging into the slot's words directly queued 1.2 M same-address accesses and cost
  // 1.8 ms; k_grid_params reduces the block words
  unsigned int mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  if (head) {
    const float c = (float)(rs.end - i);
    const float4 q = make_float4(rs.sx / c, rs.sy / c, rs.sz / c, 1.f);
    filt[s.off + pos] = q;
    if (finite3(q.x, q.y, q.z)) {
      const unsigned int a = f2ord(q.x), b = f2ord(q.y), cc = f2ord(q.z);
      mn[0] = a; mn[1] = b; mn[2] = cc; mx[0] = a; mx[1] = b; mx[2] = cc;
    }
  }
  block_bbox_merge<false>(mn, mx, blockbb + ((size_t)slot_i * nb_max + chunk_i) * 6);
}

// The fused pre-pass (s3d_core.h "K2 + K3 in one sort"): the raw points are sorted by (cell, voxel), so the centroids
// come out in cell order and this kernel writes what k_centroids + k_keys_hist<1> + the grid sort + k_grid_finalize
// produced: the cell-sorted points (w = PCL's voxel key: the tie-breaking id), their xyz-only copy and the cell table.
// Cell table by gap fill as in k_grid_finalize: the head at rank `pos` owns the cells (cell of the previous head, own
// cell]; the previous head's cell is read off the key of the element before the run (the last element of the previous
// voxel).  The LAST head also owns the tail (own cell, ncells] = pos + 1 = the number of centroids; a cloud without a
// valid point gets its all-zero table from the first thread of its first block.
// A centroid is checked against the box of its cell (fused_inside): float sums of many far-away points can leave it
// outside its voxel by more than the searches allow for - the slot then reports ok = -2 and the host re-runs the batch
// on the two-sort path.
__device__ __forceinline__ void cell_gap_fill(uint32_t* __restrict__ cs, int lo, int hi, uint32_t val) {
  // cells lo..hi (inclusive; none when hi < lo) get `val`: short gaps by the owning lane, long ones by the whole wave
  const int gap = hi - lo + 1;
  if (gap > 0 && gap <= 8)
    for (int c = lo; c <= hi; ++c) cs[c] = val;
  unsigned long long big = __ballot(gap > 8);
  const int lane = lane_id();
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const int l = __shfl(lo, src, kWave), h = __shfl(hi, src, kWave);
    const uint32_t v = (uint32_t)__shfl((int)val, src, kWave);
    for (int c = l + lane; c <= h; c += kWave) cs[c] = v;
  }
}

__global__ void __launch_bounds__(kBlock) k_centroids_fused(SlotDev* __restrict__ slots, const uint32_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ vals,
                                                             const uint32_t* __restrict__ blockcnt,
                                                             float4* __restrict__ sorted, CorrVec* __restrict__ sorted3,
                                                             uint32_t* __restrict__ cell_start, int nb_max, int nslots) {
  __shared__ int lds4[4];
  int slot_i, chunk_i;
  nn_block_map(nb_max, nslots, &slot_i, &chunk_i);
  if (slot_i >= nslots) return;
  SlotDev& s = slots[slot_i];
  const int base = chunk_i * kBlock;
  if (base >= s.n_raw && chunk_i != 0) return;          // (block 0 stays: an empty cloud still needs its cell table)
  const int i = base + threadIdx.x;
  const uint32_t* __restrict__ k = keys + s.off;
  const uint32_t* __restrict__ v = vals + s.off;
  uint32_t* __restrict__ cs = cell_start + s.cell_off;
  RunSum rs;
  const bool head = block_run_sums(s, k, v, base, rs);   // (pcl::VoxelGrid: float sums in index order)
  int total;
  const int pos = block_excl_flag(head, &total, lds4) + (int)blockcnt[(size_t)slot_i * nb_max + chunk_i];
  int lo = 0, hi = -1, lo2 = 0, hi2 = -1;
  uint32_t val = 0, val2 = 0;
  if (head) {
    const uint32_t key = rs.key;
    const float4 p_first = rs.first;
    const int j = rs.end;
    const float c = (float)(j - i);
    const float qx = rs.sx / c, qy = rs.sy / c, qz = rs.sz / c;
    // (no division: the cell by a multiply-high, the id from the run's first raw point with pcl::VoxelGrid's own float
    // operations - every point of the run has that voxel -, the cell check from the centroid's position; the decode with
    // its six divisions only for a centroid that is not plainly inside its cell: 0.98 -> 0.8x ms at 256 pairs)
    const int cell = (int)fused_cell_of_key(s.fz, key);
    const uint32_t voxel = fused_voxel_of_point(s.vp, p_first.x, p_first.y, p_first.z);
    sorted[s.off + pos] = make_float4(qx, qy, qz, __uint_as_float(voxel));
    if (sorted3) { CorrVec q3; q3.x = qx; q3.y = qy; q3.z = qz; sorted3[s.off + pos] = q3; }
    if (!fused_point_in_cell(s.vp, s.g, s.fz, key, cell, qx, qy, qz)) s.fz.ok = -2;   // (any number of threads may store the same value)
    lo = i == 0 ? 0 : (int)fused_cell_of_key(s.fz, k[i - 1]) + 1;
    hi = cell; val = (uint32_t)pos;
    if (j >= s.n_raw || k[j] == kInvalidKey) { lo2 = cell + 1; hi2 = s.g.ncells; val2 = (uint32_t)pos + 1u; }
  }
  if (i == 0 && (s.n_raw <= 0 || k[0] == kInvalidKey)) { lo2 = 0; hi2 = s.g.ncells; val2 = 0u; }
  cell_gap_fill(cs, lo, hi, val);
  cell_gap_fill(cs, lo2, hi2, val2);
}

// ------------------------------------------------------------------ K3: search grid

// One wave per slot.  blockbb != nullptr: the bbox of the filtered cloud comes as six words per 256-point block of
// k_centroids and is reduced here; otherwise k_bbox<1> has left it in s.bb.
__global__ void __launch_bounds__(kWave) k_grid_params(SlotDev* slots, RunParams rp, const unsigned int* __restrict__ blockbb,
                                                       int nb_max) {
  SlotDev& s = slots[blockIdx.x];
  if (blockbb) {
    const int nb = (s.n_raw + kBlock - 1) / kBlock;
    unsigned int mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    for (int b = threadIdx.x; b < nb; b += kWave) {
      const unsigned int* w = blockbb + ((size_t)blockIdx.x * nb_max + b) * 6;
#pragma unroll
      for (int a = 0; a < 3; ++a) { mn[a] = min(mn[a], w[a]); mx[a] = max(mx[a], w[3 + a]); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mn[a] = min(mn[a], (unsigned int)__shfl_xor((int)mn[a], o, kWave));
        mx[a] = max(mx[a], (unsigned int)__shfl_xor((int)mx[a], o, kWave));
      }
    }
    if (threadIdx.x == 0)
      for (int a = 0; a < 3; ++a) { s.bb[a] = mn[a]; s.bb[3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  if (s.n > 0 && s.bb[0] != 0xFFFFFFFFu)
    for (int a = 0; a < 3; ++a) { mn[a] = ord2f(s.bb[a]); mx[a] = ord2f(s.bb[3 + a]); }
  s.g = grid_params_from_bbox(mn, mx, rp.h0, s.cell_cap);
  s.n_sort = s.n;
}

// cell-sorted copy of the points (w = index in filtered order) and cell_start[] by gap fill:
// sorted position i owns the cells (key[i-1], key[i]]; position n owns the tail up to ncells.
__global__ void __launch_bounds__(kBlock) k_grid_finalize(const SlotDev* __restrict__ slots, const float4* __restrict__ filt,
                                                           const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                           float4* __restrict__ sorted, CorrVec* __restrict__ sorted3,
                                                           uint32_t* __restrict__ cell_start) {
  // sorted: xyz + original index (the search reads it); sorted3 (registration only, may be null): the same points as
  // 12-byte xyz records for the kernels that stream a cloud in cell order (queries of K5, K6)
  const SlotDev& s = slots[blockIdx.y];
  const int n = s.n;
  const int base = blockIdx.x * kBlock;
  if (base > n) return;
  const int i = base + threadIdx.x;
  const uint32_t* __restrict__ k = keys + s.off;
  uint32_t* __restrict__ cs = cell_start + s.cell_off;
  int lo = 0, hi = -1;  // cells lo..hi (inclusive) get value i
  if (i <= n) {
    lo = (i == 0) ? 0 : (int)k[i - 1] + 1;
    hi = (i == n) ? s.g.ncells : (int)k[i];
    if (i < n) {
      const uint32_t src = vals[s.off + i];
      const float4 p = filt[s.off + src];
      sorted[s.off + i] = make_float4(p.x, p.y, p.z, __uint_as_float(src));
      if (sorted3) sorted3[s.off + i] = corr_vec(p);
    }
  }
  const int gap = hi - lo + 1;
  // short gaps: own lane; long gaps: the whole wave fills them one after the other
  if (gap > 0 && gap <= 8)
    for (int c = lo; c <= hi; ++c) cs[c] = (uint32_t)i;
  unsigned long long big = __ballot(gap > 8);
  const int lane = lane_id();
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const int l = __shfl(lo, src, kWave), h = __shfl(hi, src, kWave), v = __shfl(i, src, kWave);
    for (int c = l + lane; c <= h; c += kWave) cs[c] = (uint32_t)v;
  }
}

// ------------------------------------------------------------------ K4: k-NN -> covariance -> normal

// generic fallback (k > 32): top-k in LDS columns
__global__ void __launch_bounds__(kBlock) k_normals(const SlotDev* __restrict__ slots, const float4* __restrict__ filt,
                                                     const float4* __restrict__ sorted, const uint32_t* __restrict__ cell_start,
                                                     NormalRec* __restrict__ normals, int k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* d2s = reinterpret_cast<float*>(smem);               // [k][kBlock]
  int* idxs = reinterpret_cast<int*>(smem) + k * kBlock;      // [k][kBlock]
  const SlotDev& s = slots[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= s.n || !s.want_normals) return;
  const float4* __restrict__ P = filt + s.off;
  const float4 q = sorted[s.off + i];
  const int cnt = grid_knn(s.g, cell_start + s.cell_off, sorted + s.off, q.x, q.y, q.z, k, d2s + threadIdx.x,
                           idxs + threadIdx.x, kBlock);
  // PCL sums the neighbours in the order nearestKSearch returns them - ascending distance (ties: index, as everywhere
  // here): selection sort of this thread's LDS column, one neighbour folded per round (k <= 64, a rarely used path)
  float* d2c = d2s + threadIdx.x;
  int* idc = idxs + threadIdx.x;
  Moments m;
  moments_init(m);
  for (int j = 0; j < cnt; ++j) {
    int best = j;
    float bd = d2c[j * kBlock];
    int bi = idc[j * kBlock];
    for (int l = j + 1; l < cnt; ++l) {
      const float dl = d2c[l * kBlock];
      const int il = idc[l * kBlock];
      if (lex_less(dl, il, bd, bi)) { best = l; bd = dl; bi = il; }
    }
    d2c[best * kBlock] = d2c[j * kBlock]; idc[best * kBlock] = idc[j * kBlock];   // (slot j is consumed: no need to store the winner)
    const float4 p = P[bi];
    moments_add(m, p.x, p.y, p.z);
  }
  double n[3];
  moments_normal(m, k, n);
  normals[s.off + i] = normal_encode(n);  // CELL-SORTED order
}

// k <= KMAX: the k best live in registers as sorted packed keys (s3d_core.h grid_knn_sorted).
// Threads walk the cloud in CELL-SORTED order: the 64 lanes of a wave sit in adjacent grid cells,
// so their cell_start / candidate loads fall into a handful of cache lines.  The kernel sums the neighbours' moments
// (9 doubles in registers) and turns them into the unit normal in place with the closed-form eigenvector
// (moments_normal_direct: no register beyond the search's 94) - 16 bytes written per point.  The few points the
// closed form declines (two smallest eigenvalues closer than 1e-6 of the spread, degenerate neighbourhoods) put their
// moments into the nine planes of `moments` and their index on a list; s3d_normals_fallback_kernel runs the Jacobi
// iteration on those: its registers would otherwise halve the occupancy of the search.
// Slot -> XCD affinity as in nn_block_map: blocks of one cloud share one L2.
#ifndef S3D_KNN_WAVES
#define S3D_KNN_WAVES 4
#endif
// one point of slot `s` (position i of its cell-sorted order) through the exact 64-bit search: neighbours, PCL
// moments, closed-form normal; a point the closed form declines goes to the eigen fallback list
// BYPOS (fused pre-pass): the points carry no index into `filt` (which does not exist then) - the keys name positions in
// `sorted`, and the neighbours are gathered from there (the same xyz)
template <int KMAX, bool FULL, bool BYPOS = false>
__device__ __forceinline__ void knn_moments_point(const SlotDev& s, int i, const float4* __restrict__ filt,
                                                  const float4* __restrict__ sorted,
                                                  const uint32_t* __restrict__ cell_start, double* __restrict__ moments,
                                                  size_t plane, int k, NormalRec* __restrict__ normals,
                                                  int* __restrict__ fallback_count, int* __restrict__ fallback_list) {
  const float4* __restrict__ P = (BYPOS ? sorted : filt) + s.off;
  const float4 q = sorted[s.off + i];
  unsigned long long keys[KMAX];
  const int cnt = grid_knn_sorted<KMAX, FULL, BYPOS>(s.g, cell_start + s.cell_off, sorted + s.off, q.x, q.y, q.z, k, keys);
  Moments m;
  moments_init(m);
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < cnt) {
      const float4 p = P[(int)(uint32_t)(keys[j] & 0xFFFFFFFFull)];
      moments_add(m, p.x, p.y, p.z);
    }
  }
  double n[3];
  if (moments_normal_direct(m, k, n)) {
    normals[s.off + i] = normal_encode(n);  // CELL-SORTED order
    return;
  }
  // nine planes of `plane` doubles (the batch's point count), and the point's place in them on the list
  double* o = moments + (size_t)(s.off + i);
  o[0] = m.mean[0]; o[plane] = m.mean[1]; o[2 * plane] = m.mean[2];
  o[3 * plane] = m.c00; o[4 * plane] = m.c10; o[5 * plane] = m.c11;
  o[6 * plane] = m.c20; o[7 * plane] = m.c21; o[8 * plane] = m.c22;
  fallback_list[atomicAdd(fallback_count, 1)] = s.off + i;
}

template <int KMAX, bool FULL = false, bool BYPOS = false>
__global__ void __launch_bounds__(kBlock, S3D_KNN_WAVES) s3d_knn_moments_kernel(const SlotDev* __restrict__ slots,
                                                                  const float4* __restrict__ filt,
                                                                  const float4* __restrict__ sorted,
                                                                  const uint32_t* __restrict__ cell_start,
                                                                  double* __restrict__ moments, size_t plane, int k,
                                                                  int chunks_per_slot, const int* __restrict__ slot_list,
                                                                  int nslots, NormalRec* __restrict__ normals,
                                                                  int* __restrict__ fallback_count,
                                                                  int* __restrict__ fallback_list) {
  // slot_list: the clouds that need normals (all of them for GICP, the searched side of each pair for
  // point-to-plane), so that the block -> XCD map spreads exactly those over the chip
  int li, chunk;
  nn_block_map(chunks_per_slot, nslots, &li, &chunk);
  if (li >= nslots) return;
  const SlotDev& s = slots[slot_list[li]];
  const int i = chunk * kBlock + threadIdx.x;
  if (i >= s.n) return;
  knn_moments_point<KMAX, FULL, BYPOS>(s, i, filt, sorted, cell_start, moments, plane, k, normals, fallback_count, fallback_list);
}

// ---- K4, round 3 (k = K known at compile time): 32-bit keys + v_med3_u32 insertion + per-query segment table in LDS
// (grid_knn_med3, s3d_core.h "K4, round 3").  One thread per query in the cloud's cell order: the 64 lanes of a wave are
// spatial neighbours and their candidate loads fall into a handful of cache lines.  A query the fast path does not
// answer (the k-th distance beyond the 5x5x5 cells, a tie of the truncated distance at the k-th place, a row range
// longer than two table entries) is appended to `redo_list` as (slot, position) and served by
// s3d_knn_moments_redo_kernel with the exact 64-bit search: 0.3 % of the benchmark's points.
// Measured and dropped: dealing the 256 queries of a block to its threads by candidate count
// (balances the scan loops, but the gathers of a wave no longer coalesce: slower), and the shell candidates of a wave
// / block as one evenly shared pool with per-query inboxes in LDS (fewer instructions, slower).
#ifndef S3D_KNN3_WAVES
#define S3D_KNN3_WAVES 8
#endif
template <int K>
__global__ void __launch_bounds__(kBlock, S3D_KNN3_WAVES) s3d_knn3_moments_kernel(const SlotDev* __restrict__ slots,
                                                                   const float4* __restrict__ sorted,
                                                                   const uint32_t* __restrict__ cell_start,
                                                                   double* __restrict__ moments, size_t plane,
                                                                   int chunks_per_slot, const int* __restrict__ slot_list,
                                                                   int nslots, NormalRec* __restrict__ normals,
                                                                   int* __restrict__ fallback_count,
                                                                   int* __restrict__ fallback_list,
                                                                   int* __restrict__ redo_count, int2* __restrict__ redo_list,
                                                                   int* __restrict__ far_count, int redo_cap, int far_all) {
  constexpr int KL = K + 1;
  __shared__ uint32_t tab[kKnn3Segs * kBlock];   // entry j of thread t at tab[j * kBlock + t]: conflict-free columns
  int li, chunk;
  nn_block_map(chunks_per_slot, nslots, &li, &chunk);
  if (li >= nslots) return;
  const int slot = slot_list[li];
  const SlotDev& s = slots[slot];
  const int i = chunk * kBlock + threadIdx.x;
  if (i >= s.n) return;
  const float4* __restrict__ pts = sorted + s.off;
  const float4 q = pts[i];
  uint32_t keys[KL];
  uint32_t* ctab = tab + threadIdx.x;
  int far = 0;
  if (!grid_knn_med3<KL>(s.g, cell_start + s.cell_off, pts, q.x, q.y, q.z, ctab, kBlock, keys, far_count ? &far : nullptr)) {
    // two lists in one buffer: the near declines (ties, table overflow) grow from the front, the FAR ones from the back.
    // far = 2: fewer than K points in the 27 cells, or the K-th neighbour more than kKnn3FarRings cells away; 1: the K-th
    // beyond the 5x5x5 proof but not that far; 0: a tie, a full table.  far_all = 0: the far list takes far = 2 (the
    // wave-cooperative kernel forced on a large batch); 1: far >= 1 (a large batch: s3d_knn3_rings_kernel serves them ring
    // by ring); 2: every decline (a small batch: all of them wave-cooperatively, s3d_knn_moments_far_kernel)
    if (far + far_all >= 2) redo_list[redo_cap - 1 - atomicAdd(far_count, 1)] = make_int2(slot, i);
    else redo_list[atomicAdd(redo_count, 1)] = make_int2(slot, i);
    return;
  }
  Moments mo;
  moments_init(mo);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const float4 p = pts[knn3_position(keys[j], ctab, kBlock)];
    moments_add(mo, p.x, p.y, p.z);
  }
  double n[3];
  if (moments_normal_direct(mo, K, n)) {
    normals[s.off + i] = normal_encode(n);  // CELL-SORTED order
    return;
  }
  double* o = moments + (size_t)(s.off + i);
  o[0] = mo.mean[0]; o[plane] = mo.mean[1]; o[2 * plane] = mo.mean[2];
  o[3 * plane] = mo.c00; o[4 * plane] = mo.c10; o[5 * plane] = mo.c11;
  o[6 * plane] = mo.c20; o[7 * plane] = mo.c21; o[8 * plane] = mo.c22;
  fallback_list[atomicAdd(fallback_count, 1)] = s.off + i;
}

// ---- K4, round 6: the sparse parts of a cloud, ring by ring (grid_knn_med3_rings, s3d_core.h "K4, round 6").
// The far list of s3d_knn3_moments_kernel (far_all = 2: fewer than K points in the 27 cells, or the K-th beyond the 5x5x5
// proof - 5-13 % of the points of a real lidar scan) through the fast path's own machinery carried on beyond the 5x5x5
// cells: 32 table entries per lane in LDS, whole rings while the list is short, then one pruned box.  What it cannot
// answer (~5 % of its entries: the K-th neighbour beyond kKnn3RingMax rings, a tie band at the K-th place, a full table)
// goes on to a list of its own for s3d_knn_moments_far_kernel, a wave per query.
// A fixed grid strides over the list; its length is read on the device.
#ifndef S3D_KNN_RINGS_WAVES
#define S3D_KNN_RINGS_WAVES 4
#endif
// (Tried: the list in TWO launches - rings up to 3 first, the entries beyond them in waves of their own, since a wave runs
// as long as its deepest lane and the depths are skewed: ring 2 / 3 / 4 / 5+ for 44 / 37 / 11 / 8 % of the entries; and both
// phases in one kernel with the deep entries of a block packed into its first lanes through LDS.  Neither was better than
// the one launch - two launches: normals of a batch of real scans 2.95 -> 3.3 ms, the second launch starts every entry
// from its 27 cells again at a fifth of a wave's lanes: EXPERIMENTS.md Round 6 (v).)
template <int K, int RMAX>
__global__ void __launch_bounds__(kBlock, S3D_KNN_RINGS_WAVES) s3d_knn3_rings_kernel(const SlotDev* __restrict__ slots,
                                                                  const float4* __restrict__ sorted,
                                                                  const uint32_t* __restrict__ cell_start,
                                                                  double* __restrict__ moments, size_t plane,
                                                                  NormalRec* __restrict__ normals,
                                                                  int* __restrict__ fallback_count,
                                                                  int* __restrict__ fallback_list,
                                                                  int2* __restrict__ redo_list,
                                                                  const int* __restrict__ in_count, int in_cap,
                                                                  int* __restrict__ out_count, int out_cap,
                                                                  int* __restrict__ handed_on, int min_count) {
  constexpr int KL = K + 1, SB = kKnn3RingSegBits;
  __shared__ uint32_t tab[(1 << SB) * kBlock];   // entry j of thread t at tab[j * kBlock + t]
  const int count = *in_count;
  uint32_t* ctab = tab + threadIdx.x;
  // The policy decided ON THE DEVICE, from the list's length (the host cannot know it): a list of fewer than `min_count`
  // entries - a cloud of even density, whose few declines have their K-th neighbour just beyond the 27 cells: 0.16 % of the
  // synthetic benchmark's points against 13 % of a lidar scan's - is left to the exact search, which reads it where it
  // is (`handed_on` = its length: s3d_knn_moments_redo_kernel).  That kernel absorbs 80 000 such entries in 0.1 ms; this
  // one, a fifth of its lanes busy at half the occupancy, took 0.24.
  if (handed_on) {
    const bool hand_on = count < min_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *handed_on = hand_on ? count : 0;
    if (hand_on) return;
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < count; j += gridDim.x * kBlock) {
    const int2 e = redo_list[in_cap - 1 - j];
    const SlotDev& s = slots[e.x];
    const float4* __restrict__ pts = sorted + s.off;
    const float4 q = pts[e.y];
    uint32_t keys[KL];
    const int why = grid_knn_med3_rings<KL, SB>(s.g, cell_start + s.cell_off, pts, q.x, q.y, q.z, ctab, kBlock, keys, RMAX);
    if (why != 0) {
      // not answered (the 20th neighbour beyond this launch's rings, a tie band at the K-th place, a full table): on to
      // the list of the next stage - the launch with more rings, then the wave-cooperative kernel.  (The first launch hands
      // its ties on with the rest: telling them apart here - two destinations - makes hipcc spill 470 registers instead of
      // 40.)  NOT to the per-lane exact search: these are points of the sparse parts, where that search walks ring after
      // ring of empty rows - one such lane keeps its wave for hundreds of microseconds, and the launch lasts as long as
      // the slowest of them (measured on a batch of real scans: 3.0 ms for 0.4 % of the points, 1.2 ms for the ties alone)
      redo_list[out_cap - 1 - atomicAdd(out_count, 1)] = e;
      continue;
    }
    Moments mo;
    moments_init(mo);
#pragma unroll
    for (int n = 0; n < K; ++n) {
      const float4 p = pts[knn3_position<SB>(keys[n], ctab, kBlock)];
      moments_add(mo, p.x, p.y, p.z);
    }
    double nrm[3];
    if (moments_normal_direct(mo, K, nrm)) {
      normals[s.off + e.y] = normal_encode(nrm);
      continue;
    }
    double* o = moments + (size_t)(s.off + e.y);
    o[0] = mo.mean[0]; o[plane] = mo.mean[1]; o[2 * plane] = mo.mean[2];
    o[3 * plane] = mo.c00; o[4 * plane] = mo.c10; o[5 * plane] = mo.c11;
    o[6 * plane] = mo.c20; o[7 * plane] = mo.c21; o[8 * plane] = mo.c22;
    fallback_list[atomicAdd(fallback_count, 1)] = s.off + e.y;
  }
}

// (Round 4, measured and dropped: the SHELL stage compacted over the block.  Half of the benchmark's queries need the
// 5x5x5 shell to prove their 20 neighbours and two thirds of those find nothing in it, yet every wave pays the stage -
// 2 040 of its 5 920 instructions - because every wave has such a lane.  A kernel that lets the 256 queries of a block
// change hands between the stages (key lists through LDS, 21 words per query; the shell queries packed into the first
// waves, only those waves run the stage) returns the same normals bit for bit and takes 10.7 ms instead of 7.4: the
// exchange buffer makes it 40 KB of LDS per block = 4 waves per SIMD, and THIS kernel needs its waves - 11.1 ms at 4
// per SIMD, 7.7 at 6, 7.4 at 8.  At equal occupancy the compaction is worth 4 %, not the 14 % of the instruction
// count: the two waves of a block that have no shell query finish early, but their LDS stays with the block.)
// the queries s3d_knn3_moments_kernel listed, through the exact search; a fixed grid strides over the list.
// THIN (the host asks for it when the batch is small): the list is dealt one entry per wave while the waves last, then
// eight - an exact search is a long chain of dependent steps and the lanes of a wave serialise their different paths,
// so the list's LATENCY is what a lone registration waits for (600 entries of one pair: 161 -> 40 us).  A large
// batch's list (150 k entries) is a throughput matter: full waves.
#ifndef S3D_KNN_REDO_WAVES
#define S3D_KNN_REDO_WAVES 3
#endif
template <int KMAX, bool FULL, bool THIN, bool BYPOS = false>
__global__ void __launch_bounds__(kBlock, S3D_KNN_REDO_WAVES) s3d_knn_moments_redo_kernel(const SlotDev* __restrict__ slots,
                                                                       const float4* __restrict__ filt,
                                                                       const float4* __restrict__ sorted,
                                                                       const uint32_t* __restrict__ cell_start,
                                                                       double* __restrict__ moments, size_t plane, int k,
                                                                       NormalRec* __restrict__ normals,
                                                                       int* __restrict__ fallback_count,
                                                                       int* __restrict__ fallback_list,
                                                                       const int* __restrict__ redo_count,
                                                                       const int2* __restrict__ redo_list,
                                                                       const int* __restrict__ handed_on = nullptr,
                                                                       int far_cap = 0) {
  // (round 6) + the far list the ring search handed on (a short one: see s3d_knn3_rings_kernel), read where it lies -
  // entry j of it at redo_list[far_cap - 1 - j]
  const int near = *redo_count;
  const int count = near + (handed_on ? *handed_on : 0);
  int first = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
  if (THIN) {
    const int waves = (int)gridDim.x * (kBlock / kWave);
    const int per = count <= waves ? 1 : count <= 8 * waves ? 8 : kWave;
    const int wave = (int)blockIdx.x * (kBlock / kWave) + wave_id();
    first = lane_id() < per ? wave * per + lane_id() : count;
    stride = waves * per;
  }
  for (int j = first; j < count; j += stride) {
    const int2 e = redo_list[j < near ? j : far_cap - 1 - (j - near)];
    knn_moments_point<KMAX, FULL, BYPOS>(slots[e.x], e.y, filt, sorted, cell_start, moments, plane, k, normals, fallback_count,
                                         fallback_list);
  }
}

// ---- K4, round 5: the FAR declines of the fast path, wave-cooperatively.
// A point in a sparse part of a scan (the far field of a lidar: ring spacings of metres against a cell edge sized for
// the dense near field) has its 20th neighbour many cells away: the exact per-lane search walks ring after ring, (2r + 1)^2
// dependent row look-ups each - on the reference's scans 13 % of the points, and ONE registration of two such scans
// spent 0.86 of its 2.1 ms waiting for the slowest lane of that kernel.  Here a wave serves one such point at a time, as
// wave_nn1_coop does for the 1-NN: the rows of the box [q - d, q + d] one per lane (range look-up), their points dealt
// over the lanes, the packed keys ((d2 bits + 2^23) << 32 | id, the keys of grid_knn_sorted) collected in LDS and the K
// smallest extracted by K wave-wide minimum reductions; d doubles while the box holds fewer than K points and is then
// set to the K-th distance found - two or three attempts of two memory round trips each.  Same K keys in the same
// order as the per-lane search (same ids, same tie rule), so the same moments and normal bit for bit.
constexpr int kFarRowBatches = 8;    // batches of 64 rows whose range loads are in flight together
constexpr int kKnnFarCap = 64;       // candidate keys held in LDS before the K smallest are selected (the K best so far
                                     // included): with 64 more appended at most 128, i.e. always the rank form of the selection
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = (unsigned long long)__shfl_xor((long long)v, o, kWave);
    v = t < v ? t : v;
  }
  return v;
}
// the K smallest of buf[0, n) into buf[0, min(K, n)) in ascending order (keys are distinct and > 0); returns their number.
// Up to 128 keys - the usual case: the box is grown until it holds K points - by RANK: every lane counts the keys below
// its own against broadcast reads of the buffer (no cross-lane traffic: the K wave-wide minimum reductions of the general
// form are 12 LDS-crossbar permutes each, 10 us per selection, most of a far query's time).
template <int K>
__device__ __forceinline__ int wave_select_smallest(unsigned long long* buf, unsigned long long* sel, int n) {
  const int lane = lane_id();
  const int m = n < K ? n : K;
  if (n <= 2 * kWave) {
    const unsigned long long a = lane < n ? buf[lane] : ~0ull, b = lane + kWave < n ? buf[lane + kWave] : ~0ull;
    int ra = 0, rb = 0;
    for (int j = 0; j < n; ++j) {
      const unsigned long long kj = buf[j];       // (every lane the same address: a broadcast read)
      ra += kj < a ? 1 : 0;
      rb += kj < b ? 1 : 0;
    }
    if (lane < n && ra < K) sel[ra] = a;
    if (lane + kWave < n && rb < K) sel[rb] = b;
  } else {
    unsigned long long last = 0ull;
    for (int j = 0; j < m; ++j) {
      unsigned long long mine = ~0ull;
      for (int i = lane; i < n; i += kWave) {
        const unsigned long long v = buf[i];
        mine = (v > last && v < mine) ? v : mine;
      }
      last = wave_min_u64(mine);
      if (lane == 0) sel[j] = last;
    }
  }
  __syncthreads();   // (the block is ONE wave: the LDS stores above before the loads below)
  for (int i = lane; i < m; i += kWave) buf[i] = sel[i];
  __syncthreads();
  return m;
}

template <int K, bool BYPOS>
__global__ void __launch_bounds__(kWave) s3d_knn_moments_far_kernel(const SlotDev* __restrict__ slots,
                                                                     const float4* __restrict__ filt,
                                                                     const float4* __restrict__ sorted,
                                                                     const uint32_t* __restrict__ cell_start,
                                                                     double* __restrict__ moments, size_t plane, int k,
                                                                     NormalRec* __restrict__ normals,
                                                                     int* __restrict__ fallback_count,
                                                                     int* __restrict__ fallback_list,
                                                                     const int* __restrict__ far_count,
                                                                     const int2* __restrict__ redo_list, int redo_cap,
                                                                     float first_box_cells) {
  __shared__ unsigned long long buf[kKnnFarCap + kWave];
  __shared__ unsigned long long sel[K];
  const int count = *far_count;
  const int lane = lane_id();
  for (int j = blockIdx.x; j < count; j += gridDim.x) {
    const int2 e = redo_list[redo_cap - 1 - j];
    const int slot = __builtin_amdgcn_readfirstlane(e.x), i = __builtin_amdgcn_readfirstlane(e.y);
    const SlotDev& s = slots[slot];
    const GridParams& g = s.g;
    const uint32_t* __restrict__ cs = cell_start + s.cell_off;
    const float4* __restrict__ pts = sorted + s.off;
    const float4 q = pts[i];
    int nbest = 0;
    float d = first_box_cells * g.h;   // (3 cells for the fast path's declines; the ring search's leftovers start beyond its rings)
    for (int attempt = 0; attempt < 48; ++attempt) {
      const float m = d * 1.0001f + 2.0e-3f * g.h;
      const int x0 = imax(grid_coord(g, 0, q.x - m), 0), x1 = imin(grid_coord(g, 0, q.x + m), g.dim[0] - 1);
      const int y0 = imax(grid_coord(g, 1, q.y - m), 0), y1 = imin(grid_coord(g, 1, q.y + m), g.dim[1] - 1);
      const int z0 = imax(grid_coord(g, 2, q.z - m), 0), z1 = imin(grid_coord(g, 2, q.z + m), g.dim[2] - 1);
      const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1;
      const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
      const int nrows = (x0 <= x1 && ny > 0 && nz > 0) ? ny * nz : 0;
      // ONE pass over the rows of the box, kFarRowBatches batches of 64 rows at a time with their range loads in flight together: an
      // isolated point - the 20th neighbour tens of metres away - grows its box to ~10 000 rows, nearly all empty, and
      // the launch lasts as long as that walk.  The few points of a box that turns out too small are evaluated for
      // nothing; the rows are what costs.
      uint32_t total = 0;
      nbest = 0;
      int nbuf = 0;
      // a box that reaches into a dense part of the cloud holds thousands of points (an isolated point 10 m from the rest:
      // 6 840): once K keys are selected, only candidates below the K-th enter the buffer
      unsigned long long thr = ~0ull;
      for (int base = 0; base < nrows; base += kFarRowBatches * kWave) {
        uint32_t rs4[kFarRowBatches], len4[kFarRowBatches];
#pragma unroll
        for (int u = 0; u < kFarRowBatches; ++u) {
          const int r = base + u * kWave + lane;
          rs4[u] = 0; len4[u] = 0;
          if (r < nrows) {
            const int rowbase = g.dim[0] * ((y0 + r % ny) + g.dim[1] * (z0 + r / ny));
            rs4[u] = cs[rowbase + x0];
            len4[u] = cs[rowbase + x1 + 1] - rs4[u];
          }
        }
#pragma unroll
        for (int u = 0; u < kFarRowBatches; ++u) {
          const uint32_t rs = rs4[u], len = len4[u];
          if (__ballot(len != 0u) == 0ull) continue;         // (an empty batch of rows: the usual case far out)
          uint32_t incl = len;
#pragma unroll
          for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, o, kWave);
            if (lane >= o) incl += t;
          }
          const uint32_t tb = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
          total += tb;
          for (uint32_t t0 = 0; t0 < tb; t0 += kWave) {
            if (nbuf > kKnnFarCap) {                   // (room for 64 more keys below)
              nbuf = wave_select_smallest<K>(buf, sel, nbuf);
              if (nbuf == K) thr = sel[K - 1];
            }
            const uint32_t t = t0 + (uint32_t)lane;
            int lo = 0, hi = kWave - 1;      // first lane whose inclusive count exceeds t
#pragma unroll
            for (int step = 0; step < 6; ++step) {
              const int mid = (lo + hi) >> 1;
              const uint32_t v = (uint32_t)__shfl((int)incl, mid, kWave);
              if (t >= v) lo = mid + 1; else hi = mid;
            }
            const uint32_t row_incl = (uint32_t)__shfl((int)incl, lo, kWave), row_len = (uint32_t)__shfl((int)len, lo, kWave);
            const uint32_t row_rs = (uint32_t)__shfl((int)rs, lo, kWave);
            const bool in_box = t < tb;
            const uint32_t pos = in_box ? row_rs + (t - (row_incl - row_len)) : 0u;
            const float4 p = pts[pos];
            const float d2 = dist2(q.x, q.y, q.z, p.x, p.y, p.z);
            const unsigned long long key = ((unsigned long long)(__float_as_uint(d2) + 0x00800000u) << 32) |
                                           (unsigned long long)(BYPOS ? pos : __float_as_uint(p.w));
            const bool have = in_box && key < thr;
            const unsigned long long hm = __ballot(have);
            if (hm == 0ull) continue;
            if (have) buf[nbuf + (int)__popcll(hm & ((1ull << lane) - 1ull))] = key;
            nbuf += (int)__popcll(hm);
            __syncthreads();
          }
        }
      }
      if (total < (uint32_t)K && !whole) { d *= 2.0f; continue; }
      nbest = wave_select_smallest<K>(buf, sel, nbuf);
      if (nbest < K) break;                       // (only when the box is the whole grid: the cloud has fewer than K points)
      const float d2k = __uint_as_float((uint32_t)(sel[K - 1] >> 32) - 0x00800000u);
      if (d2k <= d * d) break;                    // nothing outside the box can be nearer than the K-th
      d = sqrtf(d2k) * 1.0001f + 1.0e-6f;         // the exact radius: the next attempt proves
    }
    // the neighbours' moments in ascending (d2, id): PCL's summation order - lane j fetches neighbour j, then every
    // lane runs the same sums (uniform values through the shuffles)
    const float4* __restrict__ P = (BYPOS ? sorted : filt) + s.off;
    const int cnt = nbest < k ? nbest : k;
    float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < cnt) pn = P[(uint32_t)(sel[lane] & 0xFFFFFFFFull)];
    Moments mo;
    moments_init(mo);
    for (int n = 0; n < cnt; ++n) moments_add(mo, __shfl(pn.x, n, kWave), __shfl(pn.y, n, kWave), __shfl(pn.z, n, kWave));
    double nrm[3];
    const bool direct = moments_normal_direct(mo, k, nrm);
    if (lane == 0) {
      if (direct) {
        normals[s.off + i] = normal_encode(nrm);
      } else {
        double* o = moments + (size_t)(s.off + i);
        o[0] = mo.mean[0]; o[plane] = mo.mean[1]; o[2 * plane] = mo.mean[2];
        o[3 * plane] = mo.c00; o[4 * plane] = mo.c10; o[5 * plane] = mo.c11;
        o[6 * plane] = mo.c20; o[7 * plane] = mo.c21; o[8 * plane] = mo.c22;
        fallback_list[atomicAdd(fallback_count, 1)] = s.off + i;
      }
    }
    __syncthreads();
  }
}

// the points s3d_knn_moments_kernel listed: full moments_normal (closed form, then cyclic Jacobi).  A fixed small
// grid strides over the list; its length is read on the device (no host round trip).
__global__ void __launch_bounds__(kBlock) s3d_normals_fallback_kernel(const int* __restrict__ fallback_count,
                                                                       const int* __restrict__ fallback_list,
                                                                       const double* __restrict__ moments, size_t plane,
                                                                       NormalRec* __restrict__ normals, int k) {
  const int count = *fallback_count;
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < count; j += gridDim.x * kBlock) {
    const int gi = fallback_list[j];
    const double* o = moments + (size_t)gi;
    Moments m;
    m.mean[0] = o[0]; m.mean[1] = o[plane]; m.mean[2] = o[2 * plane];
    m.c00 = o[3 * plane]; m.c10 = o[4 * plane]; m.c11 = o[5 * plane];
    m.c20 = o[6 * plane]; m.c21 = o[7 * plane]; m.c22 = o[8 * plane];
    double n[3];
    moments_normal(m, k, n);
    normals[gi] = normal_encode(n);
  }
}

// ------------------------------------------------------------------ pair state

__global__ void k_pair_init(PairDev* pairs, int npairs, int* n_active) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p == 0) *n_active = npairs;
  if (p >= npairs) return;
  PairDev& P = pairs[p];
  P.active = 1; P.converged = 0; P.iterations = 0; P.correspondences = 0;
  P.inner_total = 0; P.evals_total = 0;
  P.T = mat4f_identity(); P.prev = mat4f_identity(); P.final_T = P.guess; P.T_nn = mat4f_identity();
  P.fitness = 0.0; P.fit_count = 0;
}

// Everything the ICP loop wants cleared before its first pass, in ONE launch (round 5; six hipMemsetAsync calls + k_pair_init
// were seven launches of ~5 us each in front of a lone registration): the pair records; corr_d2 = NaN ("no radius hint"),
// corr_lb = 0 ("nothing known"); the 64-query records = 0xFF.. ("never evaluated record-wise"); the record / search list
// counters and the scan27 worklist counters.
__global__ void __launch_bounds__(kBlock) k_icp_reset(PairDev* pairs, int npairs, int* n_active, uint32_t* __restrict__ corr_d2,
                                                       uint32_t* __restrict__ corr_lb, size_t ncorr,
                                                       uint32_t* __restrict__ rec_words, size_t nrec_words,
                                                       int* __restrict__ rec_counts, int nrec_counts) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
  if (t == 0) *n_active = npairs;
  if (t < 8) n_active[8 + t] = 0;
  for (size_t p = t; p < (size_t)npairs; p += stride) {
    PairDev& P = pairs[p];
    P.active = 1; P.converged = 0; P.iterations = 0; P.correspondences = 0;
    P.inner_total = 0; P.evals_total = 0;
    P.T = mat4f_identity(); P.prev = mat4f_identity(); P.final_T = P.guess; P.T_nn = mat4f_identity();
    P.fitness = 0.0; P.fit_count = 0;
  }
  for (size_t i = t; i < (size_t)nrec_counts; i += stride) rec_counts[i] = 0;
  for (size_t i = t; i < nrec_words; i += stride) rec_words[i] = 0xFFFFFFFFu;
  for (size_t i = t; i < ncorr; i += stride) { corr_d2[i] = 0xFFFFFFFFu; corr_lb[i] = 0u; }
}

// ------------------------------------------------------------------ K5: transform + exact 1-NN
// ---- wave-cooperative wide search -------------------------------------------------------------------
// A query without a near neighbour (a part of the scan the other cloud does not cover, a badly aligned first
// guess) has to examine a ball of many cells: one lane walking its rows one dependent load after the other is a
// chain of hundreds of memory latencies, and the 63 other lanes of its wave wait for it.  When only a few lanes
// of a wave are in that situation the wave serves them one at a time instead: every lane takes one ROW of the
// query's box (slab test, row range look-up, scan of the row's points), a lexicographic (d2, index) wave
// reduction picks the winner, and the loop of grid_nn1_box (exact radius / doubling) runs on wave-uniform
// values.  Same neighbour, same float d2 and the same tie rule as the per-lane search; the lower bound it
// reports for the re-validation is computed from what was examined and may differ (both are valid bounds).
constexpr int kCoopMaxLanes = 6;    // more wide lanes than this: the per-lane search is the faster one
constexpr int kCoopAllLanes = 3;    // this few searching lanes in a wave: serve all of them cooperatively

#ifndef S3D_COOP_INLINE
#define S3D_COOP_INLINE __forceinline__
#endif
__device__ S3D_COOP_INLINE NNResult wave_nn1_coop(const GridParams& g, const uint32_t* __restrict__ cell_start,
                                                  const float4* __restrict__ pts, float qx, float qy, float qz,
                                                  float max_d, float d_hint, int seed_pos, bool seed_trusted,
                                                  float seed_d2 = -1.f) {
  // all arguments are wave-uniform.  seed_d2 >= 0 (with seed_pos < 0): the squared distance of a point known to exist
  // sizes the first box as a seed does, without the load of the seed (grid_nn1_box)
  const int lane = lane_id();
  NNResult best;
  best.idx = -1; best.d2 = 3.0e38f; best.pos = -1; best.second_d2 = 3.0e38f; best.radius = 0.f;
  const float cap = max_d + kNNRevalSlack * g.h;
  // every seeded search examines a shell beyond the neighbour: that is what the next pass re-validates against
  const float shell = (seed_pos >= 0 || seed_d2 >= 0.f) ? kNNRevalSlack * g.h : 0.f;
  float d = fminf(fmaxf(d_hint, 0.25f * g.h), cap);
  double unused_key = 0.0;   // (nn1_consider's packed key: the FAST variant only)
  if (seed_pos >= 0) {
    nn1_consider(best, unused_key, pts[seed_pos], (uint32_t)seed_pos, qx, qy, qz);
    d = sqrtf(best.d2) * 1.0001f + 1.0e-6f + kNNRevalSlack * g.h;
    d = fminf(seed_trusted ? d : fminf(d, g.h), cap);
  } else if (seed_d2 >= 0.f) {
    d = sqrtf(seed_d2) * 1.0001f + 1.0e-6f + kNNRevalSlack * g.h;
    d = fminf(seed_trusted ? d : fminf(d, g.h), cap);
  }
  const float eps = 2.0e-3f * g.h;
  for (int attempt = 0; attempt < 64; ++attempt) {
    const float m = d * 1.0001f + 2.0e-3f * g.h;
    const int x0 = imax(grid_coord(g, 0, qx - m), 0), x1 = imin(grid_coord(g, 0, qx + m), g.dim[0] - 1);
    const int y0 = imax(grid_coord(g, 1, qy - m), 0), y1 = imin(grid_coord(g, 1, qy + m), g.dim[1] - 1);
    const int z0 = imax(grid_coord(g, 2, qz - m), 0), z1 = imin(grid_coord(g, 2, qz + m), g.dim[2] - 1);
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    // rows farther than the best so far (+ shell) cannot matter; the limit is fixed for the whole round
    float lim2 = 3.0e38f;
    if (best.idx >= 0) {
      const float l = sqrtf(best.d2) * 1.0001f + shell;
      lim2 = shell > 0.f ? l * l : best.d2;
    }
    NNResult mine;   // this lane's share of the candidates
    mine.idx = -1; mine.d2 = 3.0e38f; mine.pos = -1; mine.second_d2 = 3.0e38f; mine.radius = 0.f;
    if (x0 <= x1 && ny > 0 && nz > 0) {
      const int nrows = ny * nz;
      for (int base = 0; base < nrows; base += kWave) {
        // step 1: one ROW per lane — slab test and range look-up (one memory latency for up to 64 rows)
        const int r = base + lane;
        uint32_t rs = 0, len = 0;
        if (r < nrows) {
          const int cy = y0 + r % ny, cz = z0 + r / ny;
          const float ylo = g.origin[1] + (float)cy * g.h, zlo = g.origin[2] + (float)cz * g.h;
          const float dy = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)) - eps, 0.f);
          const float dz = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)) - eps, 0.f);
          const float rowd2 = dy * dy + dz * dz;
          if (rowd2 <= lim2) {
            int xa = x0, xb = x1;
            if (lim2 < 1.0e30f) {
              const float rx = sqrt_bound(fmaxf(lim2 - rowd2, 0.f)) * 1.0001f + eps;
              xa = imax(x0, grid_coord(g, 0, qx - rx));
              xb = imin(x1, grid_coord(g, 0, qx + rx));
            }
            if (xa <= xb) {
              const int rowbase = g.dim[0] * (cy + g.dim[1] * cz);
              rs = cell_start[rowbase + xa];
              len = cell_start[rowbase + xb + 1] - rs;
            }
          }
        }
        // step 2: the candidates of all those rows dealt out one per lane and round (second latency)
        uint32_t incl = len;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const uint32_t t = __shfl_up(incl, o, kWave);
          if (lane >= o) incl += t;
        }
        const uint32_t total = __shfl(incl, kWave - 1, kWave);
        for (uint32_t t0 = 0; t0 < total; t0 += kWave) {
          const uint32_t t = t0 + lane;
          int lo = 0, hi = kWave - 1;      // first lane whose inclusive count exceeds t
#pragma unroll
          for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi) >> 1;
            const uint32_t v = __shfl(incl, mid, kWave);
            if (t >= v) lo = mid + 1; else hi = mid;
          }
          const uint32_t row_incl = __shfl(incl, lo, kWave), row_len = __shfl(len, lo, kWave);
          const uint32_t row_rs = __shfl(rs, lo, kWave);
          if (t < total) {
            const uint32_t k = row_rs + (t - (row_incl - row_len));
            nn1_consider(mine, unused_key, pts[k], k, qx, qy, qz);
          }
        }
      }
    }
    // merge: the incumbent (wave-uniform) competes as lane 0's extra candidate
    if (lane == 0 && best.idx >= 0) {
      if (mine.idx == best.idx) { /* met again */ }
      else if (lex_less(best.d2, best.idx, mine.d2, mine.idx < 0 ? 2147483647 : mine.idx)) {
        mine.second_d2 = fminf(mine.second_d2, mine.d2);
        mine.d2 = best.d2; mine.idx = best.idx; mine.pos = best.pos;
      } else {
        mine.second_d2 = fminf(mine.second_d2, best.d2);
      }
      mine.second_d2 = fminf(mine.second_d2, best.second_d2);
    }
    // lexicographic (d2, idx) minimum over the wave
    float wd2 = mine.d2; int widx = mine.idx < 0 ? 2147483647 : mine.idx; int wpos = mine.pos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od2 = __shfl_xor(wd2, o, kWave);
      const int oidx = __shfl_xor(widx, o, kWave);
      const int opos = __shfl_xor(wpos, o, kWave);
      if (lex_less(od2, oidx, wd2, widx)) { wd2 = od2; widx = oidx; wpos = opos; }
    }
    // runner-up: every lane's second, and the best of every lane that did not win
    float sec = mine.second_d2;
    if (mine.idx >= 0 && mine.idx != widx) sec = fminf(sec, mine.d2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sec = fminf(sec, __shfl_xor(sec, o, kWave));
    if (widx != 2147483647) { best.idx = widx; best.d2 = wd2; best.pos = wpos; }
    best.second_d2 = fminf(best.second_d2, sec);
    best.radius = lim2 < 1.0e30f ? fminf(d, sqrtf(lim2) * 0.9999f) : d;
    if (best.idx >= 0 && best.d2 <= d * d) break;   // nothing outside the box can be closer
    if (d >= cap) break;
    d = best.idx >= 0 ? fminf(sqrtf(best.d2) * 1.0001f + 1.0e-6f + shell, cap) : fminf(2.0f * d, cap);
  }
  return best;
}

// MODE 0: ICP iteration  q = transformation_ * (guess * p)   (Eigen product of the PCL-transformed point)
// MODE 1: fitness pass   q = final_transformation * p        (pcl::transformPointCloud)
// One query of the correspondence pass.  PHASE 0: everything in one go (s3d_nn_search_kernel).  PHASE 3:
// re-validate and classify only (0: still needs a search and has a near seed, 1: will walk a wide box, 2: done);
// PHASE 2: search a query that PHASE 3 left open (the compact mode of s3d_nn_search_kernel packs those to the front of the
// block in between).
struct NNArrays {
  const float4* __restrict__ sorted;
  const CorrVec* __restrict__ sorted3;   // the same points, xyz only (the query side streams these)
  const uint32_t* __restrict__ cell_start;
  const NormalRec* __restrict__ normals;
  int* __restrict__ corr_idx;
  float* __restrict__ corr_d2;
  float* __restrict__ corr_lb;
  CorrVec* __restrict__ corr_q;
  NormalRec* __restrict__ corr_n;
};

// Tref: the transformation_ of the pass the stored bounds corr_lb refer to - the previous pass in the ICP loop
// (P.T_nn); in the fitness pass, after record-wise settled passes, that of the record's last full evaluation.
template <int MODE, int PHASE>
__device__ __forceinline__ void nn_query(const PairDev& P, const SlotDev& St, const SlotDev& Ss, int pair, int i,
                                         bool need, const NNArrays& A, float max_d, int dbg,
                                         int* __restrict__ prof_counts, int* out_class, const Mat4f& Tref) {
  const int ci = P.corr_off + (need ? i : 0);
  // queries are taken in the CELL-SORTED order of their own cloud (spatially coherent waves)
  const CorrVec p0 = A.sorted3[St.off + (need ? i : 0)];
  const F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
  F3 q;
  if (MODE == 0) q = xf_eigen(P.T, pg.x, pg.y, pg.z);
  else q = xf_pcl(P.final_T, p0.x, p0.y, p0.z);
  // corr_lb: the lower bound of all OTHER points at the previous position, its sign saying whether the previous pass
  // found a neighbour (> 0) or none within max_d (< 0); 0: nothing is known.  The re-validation needs nothing else of
  // the history - the previous distance (the radius hint of a search) is loaded only by the lanes that still search,
  // four bytes per query less on the streaming passes.
  // PHASE 5 (the first pass of a registration, s3d_nn_first_kernel): nothing is known, nothing is loaded
  const float lbs = PHASE == 5 ? 0.f : A.corr_lb[ci];
  const float lb = fabsf(lbs);
  float move = 3.0e38f;                          // how far this query moved since the previous pass (if known)
  if (need && lbs != 0.f && !(dbg & 64)) {
    // re-validate the previous result by the triangle inequality (s3d_core.h nn_still_nearest)
    const F3 qo = xf_eigen(Tref, pg.x, pg.y, pg.z);     // where this query stood at the pass the stored bounds refer to
    move = sqrtf(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
    if (PHASE != 2) {                                    // (a PHASE 2 query has failed this test already)
      if (lbs > 0.f) {
        const CorrVec ps = A.corr_q[ci];                  // the neighbour itself travels with the correspondence
        const float d2n = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
        if (nn_still_nearest(sqrtf(d2n), move, lb)) {
          // same point, its exact new distance: the accumulate kernels recompute it from the copy of the neighbour
          // (bit for bit: the same float operations), only the fitness kernel reads the stored value
          if (MODE != 0) A.corr_d2[ci] = d2n;
          A.corr_lb[ci] = lb - move;                     // still a lower bound for the others (> 0: the test above)
          need = false;
        }
      } else if (nn_still_nearest(max_d, move, lb)) {
        // no point at all within lb of the previous position, lb > max_d: still none within max_d
        A.corr_lb[ci] = move - lb;                       // (< 0)
        need = false;
      }
    }
  }
  // radius hint: this query's distance in the previous pass (NaN-filled before the first one)
  const float prev = PHASE == 5 ? __int_as_float(0x7FC00000) : (need ? A.corr_d2[ci] : 0.f);
  if (PHASE == 3) {   // classify only: 0 = near seed, 1 = wide, 2 = nothing to do (block-level compaction follows)
    const bool near_c = need && prev >= 0.f && prev < 1.0e30f && prev < Ss.g.h * Ss.g.h;
    *out_class = need ? (near_c ? 0 : 1) : 2;
    return;
  }
  if (__ballot(need) == 0ull) return;
  // seed: the neighbour found by the previous pass (its distance under the new transform bounds the
  // search radius exactly).  A NEAR neighbour (< one cell) is always used.  A FAR one — a query in a part of
  // the scan the other cloud does not cover — is trusted only when the query has barely moved since the
  // previous pass: right after a large transform update a one-cell box is the better first guess, once the
  // registration settles the old neighbour is still the nearest and the search must cover its ball anyway.
  const bool has_prev = prev >= 0.f && prev < 1.0e30f;
  const bool near_seed = has_prev && prev < Ss.g.h * Ss.g.h;
  const bool far_seed = has_prev && !near_seed && move < kNNRevalSlack * Ss.g.h && !(dbg & 128);
  const int seed = (near_seed || far_seed) ? A.corr_idx[ci] : -1;
  // first pass (nothing known yet): a generous three-cell box — the shrinking-ball scan makes a large
  // initial radius cheap, while a small one costs a second scan for every badly aligned query
  const float first = 3.0f * Ss.g.h;   // (1 / 1.5 / 2 / 4 / 6 cells were measured: DESIGN.md 6a)
  const float hint = has_prev ? fminf(sqrtf(prev) * 1.25f + 0.05f * Ss.g.h, Ss.g.h) : first;
  if (prof_counts) {   // profile >= 2 only: how many queries search, how many of them without a near seed
    const unsigned long long all = __ballot(need), un = __ballot(need && !near_seed);
    if (lane_id() == 0) {
      atomicAdd(&prof_counts[0], (int)__popcll(all));
      atomicAdd(&prof_counts[1], (int)__popcll(un));
    }
  }
  NNResult r;
  r.idx = -1; r.d2 = 3.0e38f; r.pos = -1; r.second_d2 = 3.0e38f; r.radius = 0.f;
  const uint32_t* __restrict__ cs = A.cell_start + Ss.cell_off;
  const float4* __restrict__ tp = A.sorted + Ss.off;
  {
    // served by the whole wave, one query after the other: the queries that will walk a wide box when they are
    // few, and every searching query when the wave has only a handful (a converging registration: most lanes
    // were re-validated, the wave would otherwise idle through the ~10 dependent loads of one lane's box scan)
    const unsigned long long nmask = __ballot(need);
    const bool all_coop = __popcll(nmask) <= kCoopAllLanes && !(dbg & 2048);
    const bool wide = need && (all_coop || !near_seed);
    unsigned long long wmask = __ballot(wide);
    const bool coop = wmask != 0ull && __popcll(wmask) <= kCoopMaxLanes && !(dbg & 2048);
    // (the first pass keeps its best candidate as one packed key and no runner-up: nn1_consider<FAST>)
    if (need && !(coop && wide)) r = grid_nn1_box<PHASE == 5>(Ss.g, cs, tp, q.x, q.y, q.z, max_d, hint, seed, far_seed);
    if (coop) {
      while (wmask) {
        const int src = __ffsll((long long)wmask) - 1;
        wmask &= wmask - 1ull;
        const NNResult w = wave_nn1_coop(Ss.g, cs, tp, __shfl(q.x, src, kWave), __shfl(q.y, src, kWave),
                                         __shfl(q.z, src, kWave), max_d, __shfl(hint, src, kWave),
                                         __shfl(seed, src, kWave), __shfl((int)far_seed, src, kWave) != 0);
        if (lane_id() == src) r = w;
      }
    }
  }
  if (!need) return;
  // results are kept in the query cloud's cell-sorted order and name the neighbour by its POSITION in
  // the target's cell-sorted array: every later access (K6, fitness) is then coalesced or a local gather
  A.corr_idx[ci] = r.pos;
  A.corr_d2[ci] = r.d2;
  const float lbv = nn_lower_bound_others(r);   // no neighbour at all: the scanned radius
  // (first pass: the runner-up is not tracked - "nothing known" about the others; the second pass searches every
  // query again anyway, the first transform update has moved them all)
  A.corr_lb[ci] = r.pos >= 0 ? (PHASE == 5 ? 0.f : lbv) : -lbv;
  if (r.pos >= 0) {
    // a copy of the matched point and of its normal is kept with the correspondence: the re-validation
    // above and the accumulate kernel then stream them instead of gathering by index
    A.corr_q[ci] = corr_vec(A.sorted[Ss.off + r.pos]);
    // (PHASE 5 without normals: the k-NN pre-pass of a small batch is still running on a second stream;
    // k_fill_corr_normals fills this copy in before anything reads it)
    if (PHASE != 5 || A.normals) A.corr_n[ci] = A.normals[Ss.off + r.pos];
  } else if (MODE == 0) {
    // no neighbour within max_d: a neighbour at infinity, so that the distance the accumulate kernels compute from
    // this copy fails their threshold like the stored 3e38 did
    CorrVec none;
    none.x = none.y = none.z = __int_as_float(0x7F800000);
    A.corr_q[ci] = none;
  }
}

// One thread per query of the pair.  compact = 0: every lane re-validates and, if it must, searches its own query.
// compact = 1, block-level compaction: the 256 queries of a block are re-validated and classified, the ones that
// still need a search are packed to the front of the block — near-seeded first, wide ones after them — and
// searched by the first waves; the others leave.  In the third to fifth pass of a registration a third of the
// lanes search while the rest are done, scattered over all waves: without compaction a wave pays for the search
// with most of its lanes masked off.  Neighbouring queries stay together (a block is 256 consecutive points of
// the cell order), no atomics are involved.  Measured per pass on 256 x 100k pairs (plain -> compacted):
// 3.39 -> 3.49, 2.77 -> 2.17, 1.06 -> 0.94, then 0.25 -> 0.29 ms once nearly every query re-validates (five block
// barriers on a streaming kernel): the host asks for it in passes 3 to 5 only.  A global worklist (atomics, second
// kernel) loses the spatial order of the queries and was 2x slower.  Same results either way, bit for bit.
// The `dbg` switches (s3d_exec_options.debug_flags: S3D_DBG_NN_NO_REVALIDATE, _NO_FAR_SEED, _NO_COOP, _NO_COMPACT - what
// test_nn_revalidation_shortcut_is_bitwise_neutral toggles) are read here.  The A/B switches of round 1 (ring search, plain
// block map, no seeds, first-pass box size) stayed in this kernel until round 4 because compiling them out re-rolled its
// register allocation (12.4 -> 13.5 ms of NN per step in round 2); since the settled passes have kernels of their own
// this one serves pass 4 and small batches, and without the switches pass 4 takes 0.763 instead of 0.729 ms: removed.
#ifndef S3D_NN27_WAVES
#define S3D_NN27_WAVES 8    // s3d_nn_scan27_kernel
#endif
#ifndef S3D_NN27_PRESCAN
#define S3D_NN27_PRESCAN 1  // pass 2 (no seed): the own row first, its best point cuts the other eight (grid_nn1_scan27<PRESCAN>)
#endif
#ifndef S3D_NN27_PRESCAN_REVAL
#define S3D_NN27_PRESCAN_REVAL 0
#endif
#ifndef S3D_NN27_SEED
#define S3D_NN27_SEED 1     // s3d_nn_scan27_kernel cuts the 27 cells to the ball of the previous neighbour's new distance
#endif
#ifndef S3D_NN_WAVES
#define S3D_NN_WAVES 7     // waves per SIMD the register allocation is capped for (see S3D_NN_BATCH, s3d_core.h)
#endif
template <int MODE>
__global__ void __launch_bounds__(kBlock, S3D_NN_WAVES) s3d_nn_search_kernel(const PairDev* __restrict__ pairs,
                                                                const SlotDev* __restrict__ slots, NNArrays A,
                                                                float max_d, int chunks_per_pair, int npairs, int dbg,
                                                                int* __restrict__ prof_counts, int compact,
                                                                const WaveRec* __restrict__ recs,
                                                                const Mat4f* __restrict__ T_hist, int hist_stride) {
  __shared__ int order[kBlock];
  __shared__ int lds4[4];
  int pair, chunk;
  nn_block_map(chunks_per_pair, npairs, &pair, &chunk);
  if (pair >= npairs) return;
  const PairDev& P = pairs[pair];
  if (MODE == 0 && !P.active) return;
  const SlotDev& St = slots[P.slot_t];
  const int i = chunk * kBlock + threadIdx.x;
  if (chunk * kBlock >= St.n) return;
  const SlotDev& Ss = slots[P.slot_s];
  // the positions the stored bounds refer to: the previous pass in the ICP loop; in the fitness pass (MODE 1) those of
  // the record's last full evaluation, when the settled passes ran record-wise (s3d_nn_settled_kernel)
  const Mat4f* tref = &P.T_nn;       // (a wave-uniform pointer: the matrix comes through scalar loads either way)
  if (MODE == 1 && recs) {
    const int rec = chunk * (kBlock / kWave) + wave_id();
    if (rec * kWave < St.n) {          // (a wave past the end of the cloud owns no record: what lies there is stale)
      const int touch = __builtin_amdgcn_readfirstlane(recs[(P.corr_off >> 6) + rec].touch);
      if (touch >= 0 && touch < hist_stride) tref = T_hist + (size_t)pair * hist_stride + touch;
    }
  }
  const Mat4f& Tref = *tref;
  // lanes past the end of the cloud stay in the wave (the cooperative search needs all of them) but own no query
  if (!compact) {
    nn_query<MODE, 0>(P, St, Ss, pair, i, i < St.n, A, max_d, dbg, prof_counts, nullptr, Tref);
    return;
  }
  int cls = 2;
  nn_query<MODE, 3>(P, St, Ss, pair, i, i < St.n, A, max_d, dbg, nullptr, &cls, Tref);
  int n0, n1;
  const int p0 = block_excl_flag(cls == 0, &n0, lds4);
  const int p1 = block_excl_flag(cls == 1, &n1, lds4);
  if (n0 + n1 == 0) return;
  if (cls == 0) order[p0] = (int)threadIdx.x;
  if (cls == 1) order[n0 + p1] = (int)threadIdx.x;
  __syncthreads();
  if ((int)(threadIdx.x & ~(kWave - 1)) >= n0 + n1) return;        // whole wave without work
  const bool need = (int)threadIdx.x < n0 + n1;
  const int j = chunk * kBlock + (need ? order[threadIdx.x] : 0);
  // (query j may belong to ANOTHER wave's record; Tref stays this wave's.  A PHASE 2 query has failed its proof already
  // and nn_query uses Tref from here on only for `move`, which decides whether a far previous neighbour is tried as a
  // seed first: a heuristic for where the search starts - the search itself is exact whatever it is told.  In the ICP
  // passes every record of a block has the same Tref; in the fitness pass after record-wise passes a neighbouring
  // record's touch pass may differ, and the seed decision is then made against that record's displacement.)
  nn_query<MODE, 2>(P, St, Ss, pair, j, need, A, max_d, dbg, prof_counts, nullptr, Tref);
}

// ---- the SETTLED passes of a registration, record-wise (round 4; nn_record_move_bound, s3d_core.h).
// Once a registration has settled, >= 99.9 % of its queries do nothing in a correspondence pass but prove "unchanged":
// 28 bytes read and 4 written per query, fifteen times per registration of the benchmark.  Here a pass is two
// launches.  s3d_nn_record_test_kernel: one THREAD per record of 64 consecutive queries (32 bytes: the box of their
// positions, their smallest margin, the pass of their last full evaluation) tests the record against the displacement
// between the current transformation_ and that pass's - kept per pair and pass by the controller (T_hist) - and a
// record that passes is left alone: nothing of its queries is read or written.  The records that fail are appended
// to a list.  s3d_nn_record_touch_kernel<true>: a fixed grid of waves walks the
// list, one record per wave and trip, through nn_query's re-validation (against the transformation_ of the touch
// pass; a query that fails it goes to a search list, see below), which re-establishes margin and touch pass.  The
// first record-wise pass of a registration evaluates every record: s3d_nn_record_touch_kernel<false>, the per-query
// kernel's launch geometry, no list.  (One kernel that tests, packs the failing records in LDS and serves them was
// built first: 0.064 ms per settled pass of 128 pairs against 0.098 query by query, with 88 % of the records skipped -
// every block pays the chain pair record -> record -> touch transform -> test -> barrier before its first failing
// record, and its waves then take their records one after the other.)
// The neighbours, distances and copies left in corr_* are those of the per-query kernel bit for bit - every skipped
// query is proven unchanged, by a weaker inequality than nn_still_nearest's - and so are the registrations
// (S3D_DBG_NN_NO_SETTLED in s3d_exec_options.debug_flags switches the record test off; tests/test_gpu_parity.py).
// A record that was never evaluated record-wise (touch < 0) always fails the test; it is evaluated against the
// previous pass and gets its box.
// The list is kNNRecSublists lists: block b of the test kernel appends to list b % nsub (ONE atomic per block and
// list counter - 3 000 wave-level appends to a single counter made the test kernel 45 us long, all of it waiting on
// that address), wave w of the touch kernel walks list w % nsub.  An entry carries the record's touch pass, so that
// the touch kernel fetches the pair record and the touch transform side by side.
// An entry also carries where the record's queries and correspondences start and how many of its 64 lanes hold a
// query: the touch kernel issues the loads of a record's data as soon as it has the entry, next to - not after - the
// loads of the pair record.
__device__ __forceinline__ uint4 nn_record_entry(int pair, int rec, int touch, int hist_stride, int corr_off, int query_off,
                                                 int n) {
  const int valid = imin(kWave, n - rec * kWave);
  // the touch pass travels in 16 bits; 0xFFFF = "no usable touch transform" (never evaluated record-wise, or a pass
  // beyond the transform history: hist_stride <= 4096 < 0xFFFF, so a pass the history holds always fits)
  const unsigned t16 = (touch < 0 || touch >= hist_stride) ? 0xFFFFu : (unsigned)touch;
  return make_uint4((unsigned)pair, (unsigned)(corr_off + rec * kWave), t16 | ((unsigned)valid << 16),
                    (unsigned)(query_off + rec * kWave));
}
constexpr int kNNRecSublists = 64;
// The SEARCH lists (the queries of touched records that fail their own proof) are kNNSearchSublists lists, chosen by
// the RECORD: record r (its index in the batch's record array) appends to list r % kNNSearchSublists, so a list can
// hold at most every query of the records that map to it - a capacity that does not depend on which wave touches
// what - and the one atomic per touching wave spreads over 1 024 counters.  (64 counters were enough for the settled
// benchmark, ~4 000 searches per pass; on 32 pairs of 1 M points, whose passes 5-12 still search in nearly every record,
// 500 000 appends to 64 counters made the touch kernel 0.18 ms long and the first record-wise pass 5.8 ms.)
constexpr int kNNSearchSublists = 1024;
constexpr int kNNRecPerThread = 4;     // records per thread of the test kernel (large batches; 1 for small ones)
template <int RPT>
__global__ void __launch_bounds__(kBlock) s3d_nn_record_test_kernel(const PairDev* __restrict__ pairs,
                                                                     const SlotDev* __restrict__ slots, int blocks_per_pair,
                                                                     int npairs, const WaveRec* __restrict__ recs,
                                                                     const Mat4f* __restrict__ T_hist, int hist_stride,
                                                                     int* __restrict__ list_counts, int nsub, int sub_cap,
                                                                     uint4* __restrict__ list, int* __restrict__ prof_counts) {
  __shared__ int wave_cnt[kBlock / kWave][RPT];
  __shared__ int block_base;
  int pair, chunk;
  nn_block_map(blocks_per_pair, npairs, &pair, &chunk);
  if (pair >= npairs) return;
  const PairDev& P = pairs[pair];
  if (!P.active) return;
  const SlotDev& St = slots[P.slot_t];
  const int nrec = (St.n + kWave - 1) / kWave;
  if (chunk * (kBlock * RPT) >= nrec) return;
  const WaveRec* __restrict__ prec = recs + (P.corr_off >> 6);
  const Mat4f* __restrict__ hist = T_hist + (size_t)pair * hist_stride;
  const int lane = lane_id(), w = wave_id();
  WaveRec W[RPT];
  int rec[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {       // (all record loads first, then all touch transforms: two round trips per thread)
    rec[j] = chunk * (kBlock * RPT) + j * kBlock + (int)threadIdx.x;
    W[j] = prec[rec[j] < nrec ? rec[j] : nrec - 1];
  }
  bool fail[RPT];
  int at[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const bool ok = rec[j] < nrec && W[j].touch >= 0 && W[j].touch < hist_stride && W[j].margin > 0.f;
    fail[j] = rec[j] < nrec;
    if (ok) fail[j] = !(nn_record_move_bound(P.T, hist[W[j].touch], W[j].c, W[j].e) < (double)W[j].margin);
    const unsigned long long m = __ballot(fail[j]);
    at[j] = (int)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[w][j] = (int)__popcll(m);
  }
  __syncthreads();
  // this thread's entries start at: the block's base + the failing records of (j', wave') before (j, wave)
  int before[RPT], total = 0;
#pragma unroll
  for (int j = 0; j < RPT; ++j)
#pragma unroll
    for (int ww = 0; ww < kBlock / kWave; ++ww) {
      if (ww == w) before[j] = total;
      total += wave_cnt[ww][j];
    }
  if (total == 0) return;
  const int sub = (int)(blockIdx.x % (unsigned)nsub);
  if (threadIdx.x == 0) {
    block_base = atomicAdd(&list_counts[sub], total);
    if (prof_counts) {   // profile >= 2: records tested / records that go through the per-query path
      atomicAdd(&prof_counts[2], imin(kBlock * RPT, nrec - chunk * (kBlock * RPT)));
      atomicAdd(&prof_counts[3], total);
    }
  }
  __syncthreads();
  uint4* __restrict__ out = list + (size_t)sub * sub_cap + block_base;
#pragma unroll
  for (int j = 0; j < RPT; ++j)
    if (fail[j]) out[before[j] + at[j]] = nn_record_entry(pair, rec[j], W[j].touch, hist_stride, P.corr_off, St.off, St.n);
}

// A query of a touched record that fails its own re-validation needs a search.  Not here: a search is a chain of a
// dozen dependent loads, a settled pass has a few thousand of them among 25 million queries, and in a kernel that only
// visits the failing records nothing hides them - the launch ended 30-60 us after its last re-validation, waiting
// for the waves that searched (inlined, or out of line: the same; the per-query kernel hides the same searches among
// its 400 000 waves).  The lanes are appended to kNNSearchSublists search lists instead ((pair, index) as in the lists
// of the scan27 passes; one atomic per wave that has any, on the counter of the wave's list) and
// s3d_nn_record_search_kernel serves them right after, eight queries per wave.  The record's margin is then -1: it is
// evaluated again in the next pass, when its searched queries carry fresh bounds.
template <bool LISTED>
__global__ void __launch_bounds__(kBlock, 8) s3d_nn_record_touch_kernel(const PairDev* __restrict__ pairs,
                                                           const SlotDev* __restrict__ slots, NNArrays A, float max_d,
                                                           int chunks_per_pair, int npairs, WaveRec* __restrict__ recs,
                                                           const Mat4f* __restrict__ T_hist, int hist_stride,
                                                           const int* __restrict__ list_counts, int nsub, int sub_cap,
                                                           const uint4* __restrict__ list,
                                                           int* __restrict__ list_counts_next,
                                                           int* __restrict__ search_counts, int search_cap,
                                                           uint4* __restrict__ search_list, int* __restrict__ prof_counts) {
  const int lane = lane_id();
  // LISTED: the four waves of a block take four consecutive entries of ONE list - the records a block of the test
  // kernel appended together belong to one pair, whose records then come through the scalar cache once per block
  const int sub = LISTED ? (int)(blockIdx.x % (unsigned)nsub) : 0;
  const int step = LISTED ? ((int)gridDim.x / nsub) * (kBlock / kWave) : 1;   // (the grid is a multiple of nsub blocks)
  const int count = LISTED ? list_counts[sub] : 1;
  if (LISTED && (int)blockIdx.x == 0 && (int)threadIdx.x < nsub) list_counts_next[threadIdx.x] = 0;   // the next pass's test appends here
  const uint4* __restrict__ mylist = list + (size_t)sub * sub_cap;
  const int entry0 = LISTED ? ((int)blockIdx.x / nsub) * (kBlock / kWave) + wave_id() : 0;
  // (the first entry is fetched next to the count, not after it: a position beyond the count holds stale data, which is
  // not used; sub_cap covers every position a wave of this grid can name, see Batch::launch_nn)
  uint4 e_first = make_uint4(0u, 0u, 0u, 0u);
  if (LISTED && entry0 < sub_cap) e_first = mylist[entry0];
  for (int entry = entry0; entry < count; entry += step) {
    int pair, cbase, qbase, nvalid, touch = -1;
    if (LISTED) {
      const uint4 e = entry == entry0 ? e_first : mylist[entry];
      pair = __builtin_amdgcn_readfirstlane((int)e.x);
      cbase = __builtin_amdgcn_readfirstlane((int)e.y);
      touch = __builtin_amdgcn_readfirstlane((int)(e.z & 0xFFFFu));
      touch = touch == 0xFFFF ? -1 : touch;
      nvalid = __builtin_amdgcn_readfirstlane((int)(e.z >> 16));
      qbase = __builtin_amdgcn_readfirstlane((int)e.w);
    } else {
      int chunk;
      nn_block_map(chunks_per_pair, npairs, &pair, &chunk);
      if (pair >= npairs) return;
      const PairDev& P0 = pairs[pair];
      if (!P0.active) return;
      const SlotDev& St = slots[P0.slot_t];
      const int rec = chunk * (kBlock / kWave) + wave_id();
      if (rec * kWave >= St.n) return;
      cbase = P0.corr_off + rec * kWave;
      qbase = St.off + rec * kWave;
      nvalid = imin(kWave, St.n - rec * kWave);
    }
    // the record's data: nothing here depends on the pair record, whose loads run next to these
    const bool valid = lane < nvalid;
    const int ci = cbase + (valid ? lane : 0);
    const CorrVec p0 = A.sorted3[qbase + (valid ? lane : 0)];
    const float lbs = A.corr_lb[ci];
    const CorrVec ps = A.corr_q[ci];
    const PairDev& P = pairs[pair];
    // LISTED = false is the first record-wise pass of a registration: no record has been evaluated yet
    const bool have = LISTED && touch >= 0 && touch < hist_stride;
    const Mat4f& Tref = have ? T_hist[(size_t)pair * hist_stride + touch] : P.T_nn;   // (never evaluated record-wise: the previous pass)
    // nn_query's re-validation, the same float operations
    const F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
    const F3 q = xf_eigen(P.T, pg.x, pg.y, pg.z);
    const float lb = fabsf(lbs);
    bool need = valid;
    float margin = 3.0e38f;
    if (valid && lbs != 0.f) {
      const F3 qo = xf_eigen(Tref, pg.x, pg.y, pg.z);
      const float move = sqrtf(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
      if (lbs > 0.f) {
        const float dn = sqrtf(dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z));
        if (nn_still_nearest(dn, move, lb)) {
          A.corr_lb[ci] = lb - move;
          margin = nn_margin(true, lb - move, dn, max_d);
          need = false;
        }
      } else if (nn_still_nearest(max_d, move, lb)) {
        A.corr_lb[ci] = move - lb;
        margin = nn_margin(false, lb - move, 0.f, max_d);
        need = false;
      }
    }
    const unsigned long long nmask = __ballot(need);
    if (nmask != 0ull) {      // its searches: to the search list of this RECORD
      const int ssub = (cbase >> 6) % kNNSearchSublists;
      int base = 0;
      if (lane == 0) base = atomicAdd(&search_counts[ssub], (int)__popcll(nmask));
      base = __shfl(base, 0, kWave);
      if (need) {
        const int k = base + (int)__popcll(nmask & ((1ull << lane) - 1ull));
        if (k < search_cap)   // (cannot overflow: a list holds every query of the records that map to it)
          search_list[(size_t)ssub * search_cap + k] = make_uint4((unsigned)pair, (unsigned)(qbase + lane), (unsigned)ci,
                                                                  (unsigned)P.slot_s);
      }
      margin = -1.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) margin = fminf(margin, __shfl_xor(margin, o, kWave));
    WaveRec* __restrict__ W = recs + (cbase >> 6);
    if (!have) {
      // first evaluation of this registration: the box of the record's positions (guess * p: fixed from here on)
      float mn[3] = {valid ? pg.x : 3.0e38f, valid ? pg.y : 3.0e38f, valid ? pg.z : 3.0e38f};
      float mx[3] = {valid ? pg.x : -3.0e38f, valid ? pg.y : -3.0e38f, valid ? pg.z : -3.0e38f};
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, kWave));
          mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, kWave));
        }
      }
      if (lane == 0) {
        WaveRec w;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          w.c[a] = 0.5f * mn[a] + 0.5f * mx[a];
          w.e[a] = fmaxf(mx[a] - w.c[a], w.c[a] - mn[a]) * 1.000001f + 1.0e-30f;   // (float rounding of the two differences)
        }
        w.margin = margin;
        w.touch = P.iterations;
        *W = w;
      }
    } else if (lane == 0) {
      W->margin = margin;
      W->touch = P.iterations;
    }
    if (prof_counts && nmask != 0ull && lane == 0) atomicAdd(&prof_counts[0], (int)__popcll(nmask));
    if (!LISTED) return;
  }
}

// ONE query served by the whole wave (wave_nn1_coop): everything known about it (point, bound, previous distance,
// previous neighbour's position and copy) fetched in one round trip - every lane the same addresses -, nn_query's seed
// rules, the stores of nn_query by one lane.  qi: the query's place in sorted3, ci: its correspondence's place.
__device__ __forceinline__ void nn_search_one_coop(const PairDev& P, const SlotDev& Ss, const NNArrays& A, float max_d, int dbg,
                                                   int qi, int ci) {
  const CorrVec p0 = A.sorted3[qi];
  const float lbs = A.corr_lb[ci];
  const float prev = A.corr_d2[ci];
  const int prev_pos = A.corr_idx[ci];
  const CorrVec ps = A.corr_q[ci];
  const F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
  const F3 q = xf_eigen(P.T, pg.x, pg.y, pg.z);
  // nn_query's seed rules (speed only)
  float move = 3.0e38f;
  if (lbs != 0.f && !(dbg & 64)) {
    const F3 qo = xf_eigen(P.T_nn, pg.x, pg.y, pg.z);
    move = sqrtf(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
  }
  const float h = Ss.g.h;
  const bool has_prev = prev >= 0.f && prev < 1.0e30f;
  const bool near_seed = has_prev && prev < h * h;
  const bool far_seed = has_prev && !near_seed && move < kNNRevalSlack * h && !(dbg & 128);
  const float hint = has_prev ? fminf(sqrtf(prev) * 1.25f + 0.05f * h, h) : 3.0f * h;
  const uint32_t* __restrict__ cs = A.cell_start + Ss.cell_off;
  const float4* __restrict__ tp = A.sorted + Ss.off;
  NNResult r;
  if (near_seed && lbs > 0.f)     // (lbs > 0: the copy of the previous neighbour is a real point)
    r = wave_nn1_coop(Ss.g, cs, tp, q.x, q.y, q.z, max_d, hint, -1, false, dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z));
  else
    r = wave_nn1_coop(Ss.g, cs, tp, q.x, q.y, q.z, max_d, hint, (near_seed || far_seed) ? prev_pos : -1, far_seed);
  // the stores of nn_query, by one lane
  if (lane_id() != 0) return;
  A.corr_idx[ci] = r.pos;
  A.corr_d2[ci] = r.d2;
  const float lbv = nn_lower_bound_others(r);
  A.corr_lb[ci] = r.pos >= 0 ? lbv : -lbv;
  if (r.pos >= 0) {
    if (!(lbs > 0.f && r.pos == prev_pos)) {     // (the previous neighbour confirmed: its copies are in place)
      A.corr_q[ci] = corr_vec(A.sorted[Ss.off + r.pos]);
      A.corr_n[ci] = A.normals[Ss.off + r.pos];
    }
  } else {
    CorrVec none;
    none.x = none.y = none.z = __int_as_float(0x7F800000);
    A.corr_q[ci] = none;
  }
}

// The queries the touch kernel listed: the general search of nn_query (PHASE 2: "has failed its re-validation
// already"), laid out for LATENCY.  A settled pass has a few thousand of them and the launch lasts as long as one of
// them does - a chain of dependent loads, each a cold miss somewhere in a 10 GB workspace: through nn_query 40 us.
// Here an entry carries the query's place, its correspondence's place and the slot of the searched cloud, so that
// everything known about the query (point, bound, previous distance, previous neighbour's position and copy) and both
// records (pair, slot) are fetched in ONE round trip after the entry; a near previous neighbour sizes the box through
// its copy instead of being fetched (grid_nn1_box seed_d2); and a search that confirms the previous neighbour - the
// usual outcome once a registration has settled - leaves the copies of point and normal where they are.  Same
// neighbours, same float d2 as nn_query (the box search is exact whatever sizes its first box); the lower bounds may
// differ (they depend on what was examined), which moves later search decisions, not results.
// kNNSearchSublists lists: block b serves list b % kNNSearchSublists.  One wave per block and ONE query per wave and trip,
// the wave-cooperative search (wave_nn1_coop: the rows of the box one per lane, their points dealt over the lanes - two
// round trips per attempt; eight queries per wave, each lane walking its own box row by row, took 38 us per launch).
// A LONG list - a registration that has not settled yet: every record has searching queries - is a matter of
// throughput, not latency: 64 queries per wave and trip, each lane its own search (nn_query PHASE 2, as the lists of
// the scan27 passes are served).
// (round 6) throughput mode: what the flat scan of a wave's 64 entries declines - queries without a near neighbour, a
// walk through a wide box each - is served wave-cooperatively, one after the other, while at most this many lanes of the
// wave are left (the per-lane walks of a few lanes kept the whole wave for the slowest of them: passes 4-6 of a 96-pair
// batch of the reference's scans 0.36 + 0.26 + 0.23 -> 0.32 + 0.23 + 0.19 ms; 8 or 16: the same)
#ifndef S3D_RS_COOP_LANES
#define S3D_RS_COOP_LANES 16
#endif
__global__ void __launch_bounds__(kWave) s3d_nn_record_search_kernel(const PairDev* __restrict__ pairs,
                                                                      const SlotDev* __restrict__ slots, NNArrays A,
                                                                      float max_d, int dbg,
                                                                      const int* __restrict__ search_counts, int search_cap,
                                                                      const uint4* __restrict__ search_list,
                                                                      int* __restrict__ search_counts_next) {
  const int sub = (int)(blockIdx.x % (unsigned)kNNSearchSublists), part = (int)blockIdx.x / kNNSearchSublists;
  const int parts = (int)gridDim.x / kNNSearchSublists;
  const int count = imin(search_counts[sub], search_cap);
  if ((int)blockIdx.x < kNNSearchSublists / kWave) search_counts_next[blockIdx.x * kWave + threadIdx.x] = 0;   // the next pass appends here
  const uint4* __restrict__ mylist = search_list + (size_t)sub * search_cap;
  if (count > 4 * parts) {
    // (round 5) a long list is first tried with the FLAT scan of the 27 cells, the previous neighbour's copy as the seed
    // that cuts them (grid_nn1_scan27, as passes 2-3 run it): per searched query the general box search is several times
    // slower, and nearly every listed query of passes 4-5 has its neighbour within a cell.  What the scan declines goes
    // through nn_query as before.
    __shared__ uint32_t tab[kKnn3Segs * kWave];
    for (int j0 = part * kWave; j0 < count; j0 += parts * kWave) {   // (whole waves stay: nn_query votes)
      const int j = j0 + (int)threadIdx.x;
      const bool need = j < count;
      const uint4 e = mylist[need ? j : 0];
      const PairDev& P = pairs[e.x];
      const SlotDev& Ss = slots[P.slot_s];
      bool done = false;
      if (!(dbg & 524288) && need && Ss.n < kKnn3MaxPoints) {      // (S3D_DBG_NN_NO_SCAN27 switches this off too)
        const int ci = (int)e.z;
        const CorrVec p0 = A.sorted3[e.y];
        const float lbs = A.corr_lb[ci];
        const F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
        const F3 q = xf_eigen(P.T, pg.x, pg.y, pg.z);
        float seed_d2 = 3.0e38f;
        int prev_pos = -1;
        if (lbs > 0.f) {
          const CorrVec ps = A.corr_q[ci];
          seed_d2 = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
          prev_pos = A.corr_idx[ci];
        }
        NNResult r;
        if (grid_nn1_scan27<0>(Ss.g, A.cell_start + Ss.cell_off, A.sorted + Ss.off, q.x, q.y, q.z, tab + threadIdx.x, kWave, r,
                               seed_d2)) {
          A.corr_idx[ci] = r.pos;
          A.corr_d2[ci] = r.d2;
          A.corr_lb[ci] = nn_lower_bound_others(r);
          if (r.pos != prev_pos) {                 // (the previous neighbour confirmed: its copies are in place)
            A.corr_q[ci] = corr_vec(A.sorted[Ss.off + r.pos]);
            A.corr_n[ci] = A.normals[Ss.off + r.pos];
          }
          done = true;
        }
      }
#if S3D_RS_COOP_LANES > 0
      {
        unsigned long long left = __ballot(need && !done);
        if (left != 0ull && __popcll(left) <= S3D_RS_COOP_LANES && !(dbg & 2048)) {
          while (left) {
            const int src = __ffsll((long long)left) - 1;
            left &= left - 1ull;
            const int pair = __shfl((int)e.x, src, kWave), qi = __shfl((int)e.y, src, kWave);
            const int ci = __shfl((int)e.z, src, kWave), ss = __shfl((int)e.w, src, kWave);
            nn_search_one_coop(pairs[__builtin_amdgcn_readfirstlane(pair)], slots[__builtin_amdgcn_readfirstlane(ss)], A, max_d, dbg,
                               __builtin_amdgcn_readfirstlane(qi), __builtin_amdgcn_readfirstlane(ci));
          }
          continue;
        }
      }
#endif
      nn_query<0, 2>(P, slots[P.slot_t], Ss, (int)e.x, (int)e.z - P.corr_off, need && !done, A, max_d, dbg | 2048, nullptr,
                     nullptr, P.T_nn);
    }
    return;
  }
  for (int j = part; j < count; j += parts) {
    const uint4 e = mylist[j];
    const int pair = __builtin_amdgcn_readfirstlane((int)e.x), qi = __builtin_amdgcn_readfirstlane((int)e.y);
    const int ci = __builtin_amdgcn_readfirstlane((int)e.z), ss = __builtin_amdgcn_readfirstlane((int)e.w);
    nn_search_one_coop(pairs[pair], slots[ss], A, max_d, dbg, qi, ci);
  }
}

// ---- the FIRST pass of a registration has a kernel of its own (round 3): no history to load, no re-validation, no
// seeds - nn_query<0, 5>.  Inside the one kernel above the same path cost every other pass a register re-roll
// (DESIGN.md 6a x); as a separate __global__ it takes 0.13 ms off the first pass of 128 pairs and touches nothing
// else.  S3D_DBG_NN_NO_FIRST_KERNEL switches it off (A/B).
// Measured and dropped: (a) for a one-pair batch, 16 queries per wave with EVERY search served by the whole wave
// (wave_nn1_coop: four times the waves, two latencies per query) - 35 us slower per registration than the per-lane
// search; (b) the settled passes as a 24-VGPR stream kernel of the re-validation alone plus a worklist
// kernel for the ~13 queries per pair whose proof fails in every pass - the stream is bound by its 32 bytes per query
// (4.1 TB/s with either kernel), and the second launch costs more than the search code in the stream did.
#ifndef S3D_NN_FIRST_WAVES
#define S3D_NN_FIRST_WAVES 8      // (8 waves, 64 registers: 1.58 -> 1.54 ms per 128 pairs against 7; 5 and 6 are slower)
#endif
__global__ void __launch_bounds__(kBlock, S3D_NN_FIRST_WAVES) s3d_nn_first_kernel(const PairDev* __restrict__ pairs,
                                                               const SlotDev* __restrict__ slots, NNArrays A,
                                                               float max_d, int chunks_per_pair, int npairs, int dbg,
                                                               int* __restrict__ prof_counts) {
  int pair, chunk;
  nn_block_map(chunks_per_pair, npairs, &pair, &chunk);
  if (pair >= npairs) return;
  const PairDev& P = pairs[pair];
  if (!P.active) return;
  const SlotDev& St = slots[P.slot_t];
  const int i = chunk * kBlock + threadIdx.x;
  if (chunk * kBlock >= St.n) return;
  nn_query<0, 5>(P, St, slots[P.slot_s], pair, i, i < St.n, A, max_d, dbg, prof_counts, nullptr, P.T_nn);
}

// the copies of the neighbours' normals a first pass without normals left out (see nn_query PHASE 5): the launch
// geometry of the pass itself
__global__ void __launch_bounds__(kBlock) k_fill_corr_normals(const PairDev* __restrict__ pairs, const SlotDev* __restrict__ slots,
                                                               NNArrays A, int chunks_per_pair, int npairs) {
  int pair, chunk;
  nn_block_map(chunks_per_pair, npairs, &pair, &chunk);
  if (pair >= npairs) return;
  const PairDev& P = pairs[pair];
  if (!P.active) return;
  const int i = chunk * kBlock + (int)threadIdx.x;
  if (i >= slots[P.slot_t].n) return;
  const int ci = P.corr_off + i;
  const int pos = A.corr_idx[ci];
  if (pos >= 0) A.corr_n[ci] = A.normals[slots[P.slot_s].off + pos];
}

// ---- passes 2 and 3 of the ICP loop: the flat 27-cell scan (grid_nn1_scan27, s3d_core.h "K5, round 3").
// REVAL (pass 3): a query first tries the re-validation of nn_query - two thirds of them pass by then.  A query the
// scan does not answer (its neighbour farther than the 27 cells reach, a query outside the target's grid, empty cells)
// is appended to a worklist - one atomic per wave that has any - which s3d_nn_worklist_kernel serves right after with
// the general search.  Same neighbours, same distances: bit-identical registrations (S3D_DBG_NN_NO_SCAN27 = off).
// COMPACT (pass 3, where a third of the queries still search, scattered over all waves): the queries of a block that
// failed their re-validation are packed to the front of the block (one block scan, as in s3d_nn_search_kernel's
// compact mode) and only the first waves scan - every lane of them busy.
template <bool REVAL, bool COMPACT>
__global__ void __launch_bounds__(kBlock, S3D_NN27_WAVES) s3d_nn_scan27_kernel(const PairDev* __restrict__ pairs,
                                                                  const SlotDev* __restrict__ slots, NNArrays A,
                                                                  float max_d, int chunks_per_pair, int npairs,
                                                                  int* __restrict__ work_count,
                                                                  uint32_t* __restrict__ work_pair,
                                                                  uint32_t* __restrict__ work_index,
                                                                  int* __restrict__ prof_counts) {
  __shared__ uint32_t tab[kKnn3Segs * kBlock];
  __shared__ int order[COMPACT ? kBlock : 1];
  __shared__ int lds4[4];
  int pair, chunk;
  nn_block_map(chunks_per_pair, npairs, &pair, &chunk);
  if (pair >= npairs) return;
  const PairDev& P = pairs[pair];
  if (!P.active) return;
  const SlotDev& St = slots[P.slot_t];
  if (chunk * kBlock >= St.n) return;                                         // (the whole block)
  int i = chunk * kBlock + threadIdx.x;
  if (!COMPACT && (chunk * kBlock + (int)(threadIdx.x & ~(kWave - 1))) >= St.n) return;   // (whole waves: the worklist append votes)
  const SlotDev& Ss = slots[P.slot_s];
  bool need = i < St.n;
  int ci = P.corr_off + (need ? i : 0);
  CorrVec p0 = A.sorted3[St.off + (need ? i : 0)];
  F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
  F3 q = xf_eigen(P.T, pg.x, pg.y, pg.z);
  // the previous pass's neighbour under the new transform: its distance is an upper bound that cuts the 27 cells to
  // the ball it spans (grid_nn1_scan27).  The copy of the neighbour travels with the correspondence: a streamed load.
  float seed_d2 = 3.0e38f;
  constexpr bool kSeed = S3D_NN27_SEED && REVAL;   // (pass 2's neighbours are a transform update away: the cut costs more than it saves)
  if (kSeed && need) {
    const float lbs = A.corr_lb[ci];
    if (lbs > 0.f) {
      const CorrVec ps = A.corr_q[ci];
      seed_d2 = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
    }
    if (REVAL && lbs != 0.f) {      // nn_query's re-validation, the same float operations
      const float lb = fabsf(lbs);
      const F3 qo = xf_eigen(P.T_nn, pg.x, pg.y, pg.z);
      const float move = sqrtf(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
      if (lbs > 0.f) {
        if (nn_still_nearest(sqrtf(seed_d2), move, lb)) { A.corr_lb[ci] = lb - move; need = false; }
      } else if (nn_still_nearest(max_d, move, lb)) {
        A.corr_lb[ci] = move - lb;
        need = false;
      }
    }
  }
  if (!kSeed && REVAL && need) {
    const float lbs = A.corr_lb[ci];
    if (lbs != 0.f) {               // nn_query's re-validation, the same float operations
      const float lb = fabsf(lbs);
      const F3 qo = xf_eigen(P.T_nn, pg.x, pg.y, pg.z);
      const float move = sqrtf(dist2(q.x, q.y, q.z, qo.x, qo.y, qo.z));
      if (lbs > 0.f) {
        const CorrVec ps = A.corr_q[ci];
        const float d2n = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
        if (nn_still_nearest(sqrtf(d2n), move, lb)) { A.corr_lb[ci] = lb - move; need = false; }
      } else if (nn_still_nearest(max_d, move, lb)) {
        A.corr_lb[ci] = move - lb;
        need = false;
      }
    }
  }
  if (COMPACT) {
    int total;
    const int at = block_excl_flag(need, &total, lds4);
    if (total == 0) return;
    if (need) order[at] = (int)threadIdx.x;
    __syncthreads();
    if ((int)(threadIdx.x & ~(kWave - 1)) >= total) return;                   // whole wave without work
    need = (int)threadIdx.x < total;
    i = chunk * kBlock + (need ? order[threadIdx.x] : 0);
    ci = P.corr_off + i;
    p0 = A.sorted3[St.off + i];
    pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
    q = xf_eigen(P.T, pg.x, pg.y, pg.z);
    seed_d2 = 3.0e38f;
    if (kSeed && need && A.corr_lb[ci] > 0.f) {    // (the query has changed hands: its own neighbour copy)
      const CorrVec ps = A.corr_q[ci];
      seed_d2 = dist2(q.x, q.y, q.z, ps.x, ps.y, ps.z);
    }
  }
  const unsigned long long nmask = __ballot(need);
  if (nmask == 0ull) return;
  if (prof_counts && lane_id() == 0) atomicAdd(&prof_counts[0], (int)__popcll(nmask));
  NNResult r;
  bool ok = false;
  if (need) ok = grid_nn1_scan27<REVAL ? S3D_NN27_PRESCAN_REVAL : S3D_NN27_PRESCAN>(Ss.g, A.cell_start + Ss.cell_off, A.sorted + Ss.off, q.x, q.y, q.z, tab + threadIdx.x, kBlock, r, seed_d2);
  if (need && ok) {                 // (the stores of nn_query; a scan27 answer always has a neighbour)
    A.corr_idx[ci] = r.pos;
    A.corr_d2[ci] = r.d2;
    A.corr_lb[ci] = nn_lower_bound_others(r);
    A.corr_q[ci] = corr_vec(A.sorted[Ss.off + r.pos]);
    A.corr_n[ci] = A.normals[Ss.off + r.pos];
  }
  const bool fail = need && !ok;
  const unsigned long long m = __ballot(fail);
  if (m == 0ull) return;
  int base = 0;
  if (lane_id() == 0) {
    base = atomicAdd(work_count, (int)__popcll(m));
    if (prof_counts) atomicAdd(&prof_counts[1], (int)__popcll(m));   // (profile >= 2: the second counter = declined here)
  }
  base = __shfl(base, 0, kWave);
  if (fail) {
    const int k = base + (int)__popcll(m & ((1ull << lane_id()) - 1ull));
    work_pair[k] = (uint32_t)pair;
    work_index[k] = (uint32_t)i;
  }
}

// the queries s3d_nn_scan27_kernel listed, through the general search (nn_query PHASE 2: "has failed its re-validation
// already").  One wave per block; a short list is dealt 8 entries per wave (their searches diverge, and a wave
// serialises its lanes' paths: the list's latency is what counts), a long one 64.  The lanes of a wave serve
// different pairs here: no wave-cooperative search, whose grid arguments are wave-uniform.
// (round 6) how long a list is still served one query per wave: 16 entries per block instead of 2.  What a scan27 pass
// declines on the reference's scans is ~500 queries per pair WITHOUT a neighbour in range, each a walk through the 5 x 5 rows
// of a 2.9 m ball in cells that hold up to 60 points: 64 of them per wave, lane by lane, made the two worklist launches of
// a 96-pair batch 250 + 190 us; wave-cooperatively 105 + 76 us (records bit-identical).  The synthetic benchmark's list
// (cheap declines: a neighbour just beyond the 27 cells' reach) now falls under the rule too and pays 20 us in pass 2.
// Tried: 8 / 64 entries per block, 8 192 / 16 384 blocks (the same within 0.03 ms), per-lane with ceil(count / blocks)
// lanes per wave (0.60 + 0.45 ms against 0.50 + 0.39).
#ifndef S3D_WL_BLOCKS
#define S3D_WL_BLOCKS 4096
#endif
#ifndef S3D_WL_COOP_FACTOR
#define S3D_WL_COOP_FACTOR 16
#endif
__global__ void __launch_bounds__(kWave) s3d_nn_worklist_kernel(const PairDev* __restrict__ pairs,
                                                                 const SlotDev* __restrict__ slots, NNArrays A,
                                                                 float max_d, int dbg, const int* __restrict__ work_count,
                                                                 const uint32_t* __restrict__ work_pair,
                                                                 const uint32_t* __restrict__ work_index,
                                                                 int* __restrict__ work_count_next) {
  const int count = *work_count;
  if (count <= S3D_WL_COOP_FACTOR * (int)gridDim.x && !(dbg & 2048)) {
    // (round 5) a SHORT list - a lone registration, a small batch: one query per wave and trip, wave-cooperative.  What a
    // scan27 pass declines is mostly queries without a near neighbour (a part of one scan the other does not cover), each
    // a walk through a wide box: eight of them per wave, each lane on its own, made the launch as long as the slowest
    // walk - 75 us per pass of one registration of two of the reference's scans, 2 x 75 us of its 1.5 ms.
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
      const int pair = __builtin_amdgcn_readfirstlane((int)work_pair[j]);
      const int i = __builtin_amdgcn_readfirstlane((int)work_index[j]);
      const PairDev& P = pairs[pair];
      nn_search_one_coop(P, slots[P.slot_s], A, max_d, dbg, slots[P.slot_t].off + i, P.corr_off + i);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *work_count_next = 0;
    return;
  }
  const int per = count <= 8 * (int)gridDim.x ? 8 : kWave;
  for (int j0 = blockIdx.x * per; j0 < count; j0 += gridDim.x * per) {   // (whole waves stay: nn_query votes)
    const int j = j0 + (int)threadIdx.x;
    const bool need = (int)threadIdx.x < per && j < count;
    const int pair = (int)work_pair[need ? j : 0];
    const int i = (int)work_index[need ? j : 0];
    const PairDev& P = pairs[pair];
    nn_query<0, 2>(P, slots[P.slot_t], slots[P.slot_s], pair, i, need, A, max_d, dbg | 2048, nullptr, nullptr, P.T_nn);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *work_count_next = 0;   // the counter the next scan27 pass appends to
}

// API export (s3d_nn_search / s3d_knn_normals): back from cell-sorted order to the caller's point order
__global__ void __launch_bounds__(kBlock) k_export_corr(const PairDev* __restrict__ pairs, const SlotDev* __restrict__ slots,
                                                         const float4* __restrict__ sorted, const int* __restrict__ corr_idx,
                                                         const float* __restrict__ corr_d2, int* __restrict__ out_idx,
                                                         float* __restrict__ out_d2) {
  const PairDev& P = pairs[blockIdx.y];
  const SlotDev& St = slots[P.slot_t];
  const SlotDev& Ss = slots[P.slot_s];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= St.n) return;
  const int orig = __float_as_int(sorted[St.off + i].w);
  const int pos = corr_idx[P.corr_off + i];
  out_idx[P.corr_off + orig] = pos >= 0 ? __float_as_int(sorted[Ss.off + pos].w) : -1;
  out_d2[P.corr_off + orig] = corr_d2[P.corr_off + i];
}
__global__ void __launch_bounds__(kBlock) k_export_normals(const SlotDev* __restrict__ slots, const float4* __restrict__ sorted,
                                                            const NormalRec* __restrict__ normals, float4* __restrict__ out) {
  const SlotDev& s = slots[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= s.n) return;
  const NormalRec nv = normals[s.off + i];
  out[s.off + __float_as_int(sorted[s.off + i].w)] = make_float4(nv.x, nv.y, nv.z, 0.f);
}

// ------------------------------------------------------------------ K6: per-correspondence terms + reduction

// ---- fixed summation tree of the accumulate kernels --------------------------------------------------------
// A pair's sums are defined over kAccumVB VIRTUAL blocks of 256 virtual threads: virtual block v owns the 256-element
// tiles [v n / 64, (v + 1) n / 64) of the pair's n tiles (vb_tile_begin) and its virtual thread t folds element t of
// each of them in ascending order; the 256 per-thread sums of a virtual
// block are combined by the butterfly below (64 lanes) and then over its four waves in ascending order; the
// controller adds the kAccumVB = 32 block sums in kCtrlGroups = 8 groups of 4 (ascending inside a group, then the
// groups in ascending order: icp_control_pair); k_fitness_final adds its two sums in 4 groups of 8 - a different, equally
// fixed root.  HOW MANY real blocks execute the virtual blocks (64 for
// a single pair, 4 for a 256-pair batch: few long-running blocks are faster there) is a launch parameter that
// cannot change a bit of the result: a pair registers to the same edge alone, in any batch and in any shard of a
// multi-GPU sweep.
//
// Wave reduction as a butterfly that HALVES the value set at every level: partners l and l ^ MASK split the N
// live values - the lane with the bit clear keeps the low half and sends the high half, its partner the other
// way round - so the six levels move N/2 + N/4 + ... ~ N values per lane instead of 6 N (76 accumulators: 77
// exchanges instead of 456; a real block of a large batch reduces 16 virtual blocks per launch).  Lane l ends up
// holding the wave totals of at most kSlots of the N values (wrs_index).
template <int N, int MASK>
__device__ __forceinline__ void wrs_level(double* v, int lane) {
  constexpr int H = (N + 1) / 2;
  if constexpr (MASK == 32 || MASK == 16) {
    // the two big levels (3/4 of all exchanges) as register-file swaps: v_permlane32_swap exchanges the upper half
    // of its first operand with the lower half of its second, v_permlane16_swap the odd 16-lane rows of the first with
    // the even rows of the second.  Applied to (low value, high value) every lane then holds the value it keeps in
    // one register and its partner's copy of it in the other: no select, no LDS permute, one addition.
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double lo_v = v[k];
      const double hi_v = (H + k < N) ? v[H + k] : 0.0;
      unsigned a0 = (unsigned)__double2loint(lo_v), a1 = (unsigned)__double2hiint(lo_v);
      unsigned b0 = (unsigned)__double2loint(hi_v), b1 = (unsigned)__double2hiint(hi_v);
      if constexpr (MASK == 32) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
      } else {
        const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        a0 = r0[0]; b0 = r0[1]; a1 = r1[0]; b1 = r1[1];
      }
      const double x = __hiloint2double((int)a1, (int)a0), y = __hiloint2double((int)b1, (int)b0);
      // lanes with the bit clear: x = own low value, y = the partner's; lanes with it set: y = own high value,
      // x = the partner's - the same two operands as "keep + received" of the generic form
      v[k] = x + y;
    }
  } else {
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double lo_v = v[k];
      const double hi_v = (H + k < N) ? v[H + k] : 0.0;
      const double send = hi ? lo_v : hi_v;
      const double keep = hi ? hi_v : lo_v;
      v[k] = keep + __shfl_xor(send, MASK, kWave);
    }
  }
}
// index (among the N inputs of the level with mask MASK) of the value whose wave total lane `lane` holds in
// slot `slot` after the last level; -1: that slot holds nothing
template <int N, int MASK>
__device__ __forceinline__ int wrs_index(int lane, int slot) {
  constexpr int H = (N + 1) / 2;
  int idx;
  if constexpr (MASK == 1) idx = slot;
  else idx = wrs_index<H, MASK / 2>(lane, slot);
  if (idx < 0 || idx >= H) return -1;
  idx = (lane & MASK) ? H + idx : idx;
  return idx < N ? idx : -1;
}
constexpr int wrs_half(int n, int levels) { return levels == 0 ? n : wrs_half((n + 1) / 2, levels - 1); }

// the 256-element tiles [first, last) of virtual block v: the ntiles tiles of a pair dealt out contiguously and as
// evenly as integers allow (6 or 7 tiles each for 100 k points), so that no virtual block runs masked steps
__device__ __forceinline__ int vb_tile_begin(int v, int ntiles) { return (int)(((long long)v * ntiles) / kAccumVB); }

// parity: 0 / 1 alternating between the consecutive virtual blocks of a real block.  The four wave results go through
// one of two LDS buffers, so ONE block barrier per virtual block is enough: a wave that writes buffer p again (two
// virtual blocks later) has passed the barrier in between, which waves 0-1 reach only after reading buffer p.
// (Two barriers, or none with per-wave partials and a combine kernel, measured the same: DESIGN.md 6a.)
template <int NACC>
__device__ __forceinline__ void block_reduce_store_fixed(double (&acc)[NACC], double* __restrict__ out, int parity) {
  __shared__ double red[2][kBlock / kWave][NACC];
  const int lane = lane_id(), w = wave_id();
  wrs_level<NACC, 32>(acc, lane);
  wrs_level<wrs_half(NACC, 1), 16>(acc, lane);
  wrs_level<wrs_half(NACC, 2), 8>(acc, lane);
  wrs_level<wrs_half(NACC, 3), 4>(acc, lane);
  wrs_level<wrs_half(NACC, 4), 2>(acc, lane);
  wrs_level<wrs_half(NACC, 5), 1>(acc, lane);
  constexpr int kSlots = wrs_half(NACC, 6);
#pragma unroll
  for (int sl = 0; sl < kSlots; ++sl) {
    const int idx = wrs_index<NACC, 32>(lane, sl);
    if (idx >= 0) red[parity][w][idx] = acc[sl];
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double v = red[parity][0][threadIdx.x];
#pragma unroll
    for (int ww = 1; ww < kBlock / kWave; ++ww) v += red[parity][ww][threadIdx.x];
    out[threadIdx.x] = v;
  }
}
template <int NACC>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NACC], double* __restrict__ out) {
  __shared__ double red[kBlock / kWave][NACC];
  const int lane = lane_id(), w = wave_id();
#pragma unroll
  for (int c = 0; c < NACC; ++c) {
    const double v = wave_sum(acc[c]);
    if (lane == 0) red[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double v = red[0][threadIdx.x];
#pragma unroll
    for (int ww = 1; ww < kBlock / kWave; ++ww) v += red[ww][threadIdx.x];
    out[threadIdx.x] = v;
  }
}

// ------------------------------------------------------------------ K7: per-pair controller
// fixed-order sum of the block partials, then the solver step and the PCL stopping rule.  Called by ALL threads of a
// block (kCtrlThreads of them) for one pair; the scalar solver runs on wave 0 with the record in LDS.
constexpr int kCtrlGroups = 8;      // the root of the fixed summation tree: 8 groups of kAccumVB / 8 = 4 virtual blocks
static_assert(kAccumVB % kCtrlGroups == 0 && kAccumVB % 8 == 0, "the tree roots deal the virtual blocks in whole groups");
constexpr int kCtrlThreads = 128;   // threads that load the tree root (the stand-alone kernel's block size: more
                                    // would cap the optimiser's registers below the 256 it uses)
// host_word (may be null; pinned host memory the device writes through): [0] the index of the newest outer iteration
// whose controller has started (written by the kernel, block 0), [1] = tag once the last active pair of the batch has
// stopped - what the host throttles and ends its launch loop by (Batch::stage_icp) without ever waiting for the stream.
__device__ __forceinline__ void icp_report_done(int* n_active, int* host_word, int tag) {
  if (atomicSub(n_active, 1) == 1 && host_word)
    __hip_atomic_store(host_word + 1, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void icp_control_pair(PairDev& P, const double* __restrict__ pair_partials, const RunParams& rp,
                                                 int* n_active, Mat4f* __restrict__ pair_hist, int hist_stride,
                                                 int* host_word = nullptr, int tag = 0) {
  __shared__ double grp[kCtrlGroups][GQ_NACC];
  __shared__ double acc[GQ_NACC];
  const bool gicp = rp.algorithm != 0;
  const int nacc = gicp ? GQ_NACC : PP_NACC;
  // root of the fixed tree: the kAccumVB virtual-block sums in 8 groups of 4 (ascending inside a group), then the
  // groups in ascending order.  Every thread takes up to five (group, accumulator) cells and issues all of their
  // loads before the first addition: one L2 round trip instead of eight on the critical path of every iteration.
  if (threadIdx.x < kCtrlThreads) {
    constexpr int kCells = (kCtrlGroups * GQ_NACC + kCtrlThreads - 1) / kCtrlThreads;   // 5
    constexpr int kPer = kAccumVB / kCtrlGroups;                                        // 4
    double t[kCells][kPer];
#pragma unroll
    for (int u = 0; u < kCells; ++u) {
      const int cell = (int)threadIdx.x + u * kCtrlThreads;
      const int g = cell / GQ_NACC, c = cell % GQ_NACC;
      const bool ok = g < kCtrlGroups && c < nacc;
      const double* src = pair_partials + ((size_t)(ok ? g : 0) * kPer) * GQ_NACC + (ok ? c : 0);
#pragma unroll
      for (int j = 0; j < kPer; ++j) t[u][j] = src[(size_t)j * GQ_NACC];
    }
#pragma unroll
    for (int u = 0; u < kCells; ++u) {
      const int cell = (int)threadIdx.x + u * kCtrlThreads;
      const int g = cell / GQ_NACC, c = cell % GQ_NACC;
      if (g < kCtrlGroups && c < nacc) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) v += t[u][j];
        grp[g][c] = v;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < nacc) {
    double v = grp[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < kCtrlGroups; ++g) v += grp[g][threadIdx.x];
    acc[threadIdx.x] = v;
  }
  __syncthreads();
  // Wave 0 runs the optimiser redundantly on all 64 lanes (identical inputs -> identical, uniform
  // control flow) so that gq_eval can spread its transcendental and dot-product work over the lanes;
  // lane 0 alone writes the pair state back.
  if (threadIdx.x >= kWave) return;
  const bool writer = threadIdx.x == 0;
  Mat4f T = P.T;
  const Mat4f prev = T;
  const int it = P.iterations + 1;
  int rc, inner = 0, evals = 0, corr;
  if (gicp) {
    corr = (int)acc[GQ_CNT];
    rc = gicp_estimate_bfgs(acc, rp.max_inner, T, &inner, &evals);
  } else {
    corr = (int)acc[PP_CNT];
    rc = pp_update(acc, T);
  }
  const double delta = rc ? 0.0 : icp_delta(prev, T, rp.rotation_epsilon, rp.transformation_epsilon);
  if (!writer) return;
  P.correspondences = corr;
  P.prev = prev;
  P.T_nn = prev;
  // the transformation_ this iteration's correspondence pass ran with, by pass index: what the record-level
  // re-validation of the settled passes measures displacements from (s3d_nn_settled_kernel)
  if (pair_hist && it - 1 < hist_stride) pair_hist[it - 1] = prev;
  if (rc) {  // PCLException path: loop breaks, converged_ stays false
    P.active = 0; P.converged = 0;
    icp_report_done(n_active, host_word, tag);
    return;
  }
  P.inner_total += inner; P.evals_total += evals;
  P.T = T;
  P.iterations = it;
  if (it >= rp.max_iterations || (!rp.force_iterations && delta < 1.0)) {
    P.converged = 1; P.active = 0; P.prev = T;
    icp_report_done(n_active, host_word, tag);
  }
}

// One block per pair.  (Measured dead end, round 2: running this body at the end of the accumulate kernels - the last
// block of a pair to finish, a counter per pair - saves the launch, but the optimiser's 308 registers become the
// accumulate kernel's: 240 -> 328, one wave per SIMD instead of two, which costs more than the launch.)
__global__ void __launch_bounds__(kCtrlThreads) s3d_icp_control_kernel(PairDev* pairs, const double* __restrict__ partials,
                                                                        RunParams rp, int* n_active,
                                                                        Mat4f* __restrict__ T_hist, int hist_stride,
                                                                        int* host_word, int launch, int tag) {
  if (host_word && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(host_word, launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  PairDev& P = pairs[blockIdx.x];
  if (!P.active) return;
  icp_control_pair(P, partials + (size_t)blockIdx.x * kAccumVB * GQ_NACC, rp, n_active,
                   T_hist ? T_hist + (size_t)blockIdx.x * hist_stride : nullptr, hist_stride, host_word, tag);
}

// GICP: Mahalanobis matrix + 73-term quadratic form (s3d_core.h "GICP quadratic form")
__global__ void __launch_bounds__(kBlock) s3d_gicp_accumulate_kernel(const PairDev* __restrict__ pairs,
                                                                      const SlotDev* __restrict__ slots,
                                                                      const CorrVec* __restrict__ sorted,
                                                                      const NormalRec* __restrict__ normals,
                                                                      const CorrVec* __restrict__ corr_q,
                                                                      const NormalRec* __restrict__ corr_n,
                                                                      double* __restrict__ partials, RunParams rp) {
  const PairDev& P = pairs[blockIdx.y];
  if (!P.active) return;
  const SlotDev& St = slots[P.slot_t];
  double R[9], S[6], Th0[12];
  gicp_rotation(P.T, P.guess, R, S);
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a) Th0[c * 4 + a] = (double)S3D_M(P.T, c, a);
  // R, S and Th0 are the same for every lane of the block (one pair) but come out of vector arithmetic: moved to
  // scalar registers they free 54 VGPRs, which takes the kernel from 262 to 232 and so from one to TWO waves
  // per SIMD - twice the loads in flight (0.39 -> 0.335 ms per launch; a second prefetch stage on top: nothing)
#pragma unroll
  for (int c = 0; c < 9; ++c) R[c] = wave_uniform(R[c]);
#pragma unroll
  for (int c = 0; c < 6; ++c) S[c] = wave_uniform(S[c]);
#pragma unroll
  for (int c = 0; c < 12; ++c) Th0[c] = wave_uniform(Th0[c]);
  const int M = St.n;
  // The virtual blocks v = blockIdx.x, blockIdx.x + gridDim.x, ... of this pair (see block_reduce_store_fixed), one
  // after the other, as ONE software-pipelined stream: the five loads of the next element - the first element of
  // the next virtual block included - are in flight while the current one is folded into the 73 accumulators (the
  // loads of one element per lane and wave do not cover the HBM latency-bandwidth product).
  const int ntiles = (M + kBlock - 1) / kBlock;
  constexpr int kAccOutMul = 1;
  double* __restrict__ out = partials + (size_t)blockIdx.y * kAccumVB * GQ_NACC;
  // next virtual block of this real block that owns at least one tile, starting at `from`; the empty ones on the
  // way (a cloud of fewer than 64 tiles) get their zero sums
  auto next_nonempty = [&](int from) {
    int u = from;
    for (; u < kAccumVB && vb_tile_begin(u, ntiles) == vb_tile_begin(u + 1, ntiles); u += gridDim.x)
      for (int t = threadIdx.x; t < GQ_NACC * kAccOutMul; t += kBlock) out[(size_t)u * GQ_NACC * kAccOutMul + t] = 0.0;
    return u;
  };
  int v = next_nonempty(blockIdx.x);
  if (v >= kAccumVB) return;
  int tile = vb_tile_begin(v, ntiles);
  int i = tile * kBlock + threadIdx.x;
  CorrVec p0 = corr_vec(make_float4(0.f, 0.f, 0.f, 0.f)), qf = p0;
  NormalRec na = {0.f, 0.f, 0.f, 0u}, nb = na;
  {
    const int j = i < M ? i : M - 1;
    p0 = sorted[St.off + j]; qf = corr_q[P.corr_off + j];
    na = normals[St.off + j]; nb = corr_n[P.corr_off + j];
  }
  int parity = 0;
  while (v < kAccumVB) {
    double acc[GQ_NACC];
#pragma unroll
    for (int c = 0; c < GQ_NACC; ++c) acc[c] = 0.0;
    const int vn = next_nonempty(v + gridDim.x);
    const int tile_end = vb_tile_begin(v + 1, ntiles);
    const int first_next = vn < kAccumVB ? vb_tile_begin(vn, ntiles) * kBlock + (int)threadIdx.x : i;
    for (; tile < tile_end; ++tile) {
      const int in = tile + 1 < tile_end ? i + kBlock : first_next;
      const int j = in < M ? in : M - 1;   // (the last tile may be partial: no branch around the loads)
      const CorrVec p0n = sorted[St.off + j], qfn = corr_q[P.corr_off + j];
      const NormalRec nan_ = normals[St.off + j], nbn = corr_n[P.corr_off + j];
      // the squared distance of the correspondence as the NN kernel computed it (the same float operations on the
      // same values; a query without a neighbour has one at infinity): not stored and read back, 4 bytes less
      const F3 pf = xf_pcl(P.guess, p0.x, p0.y, p0.z);
      const F3 qn = xf_eigen(P.T, pf.x, pf.y, pf.z);
      const float d2 = dist2(qn.x, qn.y, qn.z, qf.x, qf.y, qf.z);
      if (i < M && (double)d2 < rp.dist_threshold) {
        // unit normals to 6e-11 from their 16-byte records (s3d_core.h NormalRec)
        double n1[3], n2[3];
        normal_decode(na, n1);
        normal_decode(nb, n2);
        double n1r[3], Mm[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) n1r[a] = fma(R[a * 3], n1[0], fma(R[a * 3 + 1], n1[1], R[a * 3 + 2] * n1[2]));
        gicp_mahalanobis(S, n1r, n2, rp.gicp_epsilon, Mm);
        const double pd[3] = {pf.x, pf.y, pf.z};
        const double qd[3] = {qf.x, qf.y, qf.z};
        gq_accumulate(acc, pd, qd, Mm, Th0);
      }
      p0 = p0n; qf = qfn; na = nan_; nb = nbn;
      i = in;
    }
    block_reduce_store_fixed<GQ_NACC>(acc, out + (size_t)v * GQ_NACC * kAccOutMul, parity);
    parity ^= 1;
    v = vn;
    if (v < kAccumVB) tile = vb_tile_begin(v, ntiles);
  }
}


// point-to-plane: J^T J (21) + J^T r (6) + r^2 + count
__global__ void __launch_bounds__(kBlock) s3d_p2plane_accumulate_kernel(const PairDev* __restrict__ pairs,
                                                                         const SlotDev* __restrict__ slots,
                                                                         const CorrVec* __restrict__ sorted,
                                                                         const CorrVec* __restrict__ corr_q,
                                                                         const NormalRec* __restrict__ corr_n,
                                                                         double* __restrict__ partials, RunParams rp) {
  const PairDev& P = pairs[blockIdx.y];
  if (!P.active) return;
  const SlotDev& St = slots[P.slot_t];
  const int M = St.n;
  const int ntiles = (M + kBlock - 1) / kBlock;
  double* __restrict__ out = partials + (size_t)blockIdx.y * kAccumVB * GQ_NACC;
  // virtual blocks of the fixed summation tree (block_reduce_store_fixed), gridDim.x of them at a time
  int parity = 0;
  for (int v = blockIdx.x; v < kAccumVB; v += gridDim.x) {
    const int t0 = vb_tile_begin(v, ntiles), t1 = vb_tile_begin(v + 1, ntiles);
    if (t0 == t1) {   // (uniform over the block: no barrier is skipped by part of it)
      if (threadIdx.x < PP_NACC) out[(size_t)v * GQ_NACC + threadIdx.x] = 0.0;
      continue;
    }
    double acc[PP_NACC];
#pragma unroll
    for (int c = 0; c < PP_NACC; ++c) acc[c] = 0.0;
    for (int i = t0 * kBlock + (int)threadIdx.x; i < t1 * kBlock && i < M; i += kBlock) {
      const CorrVec p0 = sorted[St.off + i];
      const F3 pg = xf_pcl(P.guess, p0.x, p0.y, p0.z);
      const F3 pq = xf_eigen(P.T, pg.x, pg.y, pg.z);
      const CorrVec qf = corr_q[P.corr_off + i];
      const float d2 = dist2(pq.x, pq.y, pq.z, qf.x, qf.y, qf.z);   // as the NN kernel computed it (see the GICP kernel)
      if (!((double)d2 < rp.dist_threshold)) continue;
      const NormalRec nf = corr_n[P.corr_off + i];   // (point-to-plane keeps its float normals: its own design, its own oracle)
      const double pd[3] = {pq.x, pq.y, pq.z};
      const double qd[3] = {qf.x, qf.y, qf.z};
      const double nd[3] = {nf.x, nf.y, nf.z};
      pp_accumulate(acc, pd, qd, nd);
    }
    block_reduce_store_fixed<PP_NACC>(acc, out + (size_t)v * GQ_NACC, parity);
    parity ^= 1;
  }
}

// final_transformation_ = previous_transformation_ * guess
__global__ void k_pair_finalize(PairDev* pairs, int npairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  pairs[p].final_T = mat4f_mul(pairs[p].prev, pairs[p].guess);
}

// ------------------------------------------------------------------ K8: fitness (masked mean of d2)
__global__ void __launch_bounds__(kBlock) s3d_fitness_partial_kernel(const PairDev* __restrict__ pairs,
                                                                      const SlotDev* __restrict__ slots,
                                                                      const float* __restrict__ corr_d2,
                                                                      double* __restrict__ partials, RunParams rp) {
  const PairDev& P = pairs[blockIdx.y];
  const int M = slots[P.slot_t].n;
  double* __restrict__ out = partials + (size_t)blockIdx.y * kAccumVB * GQ_NACC;
  // the same virtual blocks as the accumulate kernels: the score does not depend on the launch geometry
  const int ntiles = (M + kBlock - 1) / kBlock;
  for (int v = blockIdx.x; v < kAccumVB; v += gridDim.x) {
    double acc[2] = {0.0, 0.0};
    const int t1 = vb_tile_begin(v + 1, ntiles);
    for (int i = vb_tile_begin(v, ntiles) * kBlock + (int)threadIdx.x; i < t1 * kBlock && i < M; i += kBlock) {
      const float d2 = corr_d2[P.corr_off + i];
      if ((double)d2 <= rp.fit_range) { acc[0] += (double)d2; acc[1] += 1.0; }   // (no neighbour at all: d2 = 3e38)
    }
    block_reduce_store_fixed<2>(acc, out + (size_t)v * GQ_NACC, (v / (int)gridDim.x) & 1);
  }
}

__global__ void k_fitness_final(PairDev* pairs, const double* __restrict__ partials, int npairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const double* in = partials + (size_t)p * kAccumVB * GQ_NACC;
  double s = 0.0, c = 0.0;   // fixed root: kAccumVB / 8 = 4 groups of 8, ascending (not the controller's 8 groups of 4)
  for (int g = 0; g < kAccumVB / 8; ++g) {
    const double* gi = in + (size_t)g * 8 * GQ_NACC;
    const double sg = ordered_partial_sum(gi, 8, GQ_NACC), cg = ordered_partial_sum(gi + 1, 8, GQ_NACC);
    s = g == 0 ? sg : s + sg;
    c = g == 0 ? cg : c + cg;
  }
  pairs[p].fitness = c > 0.0 ? s / c : 1.7976931348623157e308;
  pairs[p].fit_count = (int)c;
}

// host hand-over of a cloud with a stride other than 4 floats: the raw floats are copied as they are and
// expanded to the float4 layout on the device (no host-side conversion pass, 25 % fewer PCIe bytes for packed xyz)
__global__ void __launch_bounds__(kBlock) k_expand_points(const float* __restrict__ raw, int n, int stride,
                                                          float4* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float* p = raw + (size_t)i * stride;
  out[i] = make_float4(p[0], p[1], p[2], 1.f);
}

// ------------------------------------------------------------------ B1: patch accumulation
// PointCloudSensor::getAccumulatedCloud / createCombinedMeasurement (PointCloudSensor.cpp:235-266): every
// cloud of the patch through its own pose, appended in vertex order; with `frame` the float result goes
// through the second transform (pose.inverse()) in the same thread, rounded to float in between exactly
// as the reference's two transformPointCloud passes do.  16 B in, 16 B out per point.
struct XformJob {
  const float4* src;
  int n, out_off;
  double T[12];  // 3x4 row-major
};
struct Xf3x4d { double m[12]; int enabled; };

__global__ void __launch_bounds__(kBlock) s3d_transform_concat_kernel(const XformJob* __restrict__ jobs,
                                                                       float4* __restrict__ out, Xf3x4d frame) {
  const XformJob& J = jobs[blockIdx.y];
  const int n = J.n;
  const float4* __restrict__ src = J.src;
  float4* __restrict__ dst = out + J.out_off;
  double T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = J.T[i];
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = src[i];
    F3 q = xf_pcl_d(T, p.x, p.y, p.z);
    if (frame.enabled) q = xf_pcl_d(frame.m, q.x, q.y, q.z);
    dst[i] = make_float4(q.x, q.y, q.z, 1.f);
  }
}

// ------------------------------------------------------------------ B2: radius outlier removal
// One thread per point in CELL order (neighbouring lanes scan the same rows); the keep flag lands at the
// point's original index so that the compaction below preserves the input order like pcl::FilterIndices.
__global__ void __launch_bounds__(kBlock) s3d_radius_count_kernel(const SlotDev* __restrict__ slots,
                                                                   const float4* __restrict__ sorted,
                                                                   const uint32_t* __restrict__ cell_start,
                                                                   uint32_t* __restrict__ flags, float reach, float r2f,
                                                                   int need) {
  const SlotDev& s = slots[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= s.n) return;
  const float4 q = sorted[s.off + i];
  const int c = grid_radius_count(s.g, cell_start + s.cell_off, sorted + s.off, q.x, q.y, q.z, reach, r2f, need);
  flags[s.off + (int)__float_as_uint(q.w)] = c >= need ? 1u : 0u;
}

__global__ void __launch_bounds__(kBlock) k_flags_count(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ flags,
                                                         uint32_t* __restrict__ blockcnt, int nb_max) {
  __shared__ int lds4[4];
  const SlotDev& s = slots[blockIdx.y];
  const int base = blockIdx.x * kBlock;
  if (base >= s.n_raw) return;
  const int i = base + threadIdx.x;
  const bool keep = i < s.n_raw && flags[s.off + i] != 0u;
  int total;
  block_excl_flag(keep, &total, lds4);
  if (threadIdx.x == 0) blockcnt[(size_t)blockIdx.y * nb_max + blockIdx.x] = (uint32_t)total;
}

// after k_heads_scan: out[pos] = point i for every kept i, input order preserved
__global__ void __launch_bounds__(kBlock) k_flags_compact(const SlotDev* __restrict__ slots, const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ blockcnt,
                                                           const float4* __restrict__ filt, float4* __restrict__ out,
                                                           int nb_max) {
  __shared__ int lds4[4];
  const SlotDev& s = slots[blockIdx.y];
  const int base = blockIdx.x * kBlock;
  if (base >= s.n_raw) return;
  const int i = base + threadIdx.x;
  const bool keep = i < s.n_raw && flags[s.off + i] != 0u;
  int total;
  const int pos = block_excl_flag(keep, &total, lds4) + (int)blockcnt[(size_t)blockIdx.y * nb_max + blockIdx.x];
  if (keep) out[pos] = filt[s.off + i];
}

// ------------------------------------------------------------------ B4: plane RANSAC scoring (fillGroundPlane)
// fillGroundPlane (PointCloudSensor.cpp:362-388) fits the ground with pcl::RandomSampleConsensus over
// SampleConsensusModelPlane; all of its time is countWithinDistance - one pass over the whole map per hypothesis.
// The hypotheses do not depend on each other's scores (the sample stream is fixed by the seed), so kPlaneBatch of
// them are scored by ONE pass over the points: each point is read once and tested against every plane of the
// batch (the planes arrive as kernel arguments = scalar registers), the counts are reduced per wave and added
// with one atomic per wave and plane.  Inlier test = PCL's: |(a x + b y) + (c z + d)| < (float)threshold.
constexpr int kPlaneBatch = 32;
struct PlaneBatch { float4 pl[kPlaneBatch]; };

__global__ void __launch_bounds__(kBlock) s3d_plane_count_kernel(const float4* __restrict__ pts, int n, PlaneBatch B,
                                                                  float thr, int* __restrict__ counts) {
  int c[kPlaneBatch];
#pragma unroll
  for (int h = 0; h < kPlaneBatch; ++h) c[h] = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 p = pts[i];
#pragma unroll
    for (int h = 0; h < kPlaneBatch; ++h) {
      const float v = (B.pl[h].x * p.x + B.pl[h].y * p.y) + (B.pl[h].z * p.z + B.pl[h].w);
      c[h] += fabsf(v) < thr ? 1 : 0;
    }
  }
#pragma unroll
  for (int h = 0; h < kPlaneBatch; ++h) {
    int v = c[h];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    if (lane_id() == 0 && v) atomicAdd(&counts[h], v);
  }
}

// ------------------------------------------------------------------ K9: NDT
// doNDT (PointCloudSensor.cpp:84-117) -> pcl::NormalDistributionsTransform.  The data-parallel parts run here:
// the voxel statistics of the target (pcl::VoxelGridCovariance: >= 6 points per voxel of edge `resolution`,
// unbiased covariance, eigenvalues raised to 1 % of the largest, inverse) and the score / gradient / Hessian pass
// over the input points (ndt.hpp computeDerivatives + updateDerivatives).  The Newton step and the More-Thuente
// line search around them are scalar and run on the host (s3d_ndt.h), one derivative pass per trial step.
constexpr int NDT_NACC = 28;          // score, gradient (6), upper triangle of the Hessian (21)
constexpr int kNdtCellDoubles = 10;   // mean (3), inverse covariance xx xy xz yy yz zz (6), pad

struct NdtAngles {                    // R = Rx Ry Rz: first and second derivatives by the three angles (3x3 each)
  double dR[3][9];
  double d2R[6][9];                   // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
};

struct NdtGrid {                      // the voxelised target of one cloud (slot)
  int slot, n;                        // slot of the batch, its filtered point count
  VoxelParams vp;                     // voxel layout at `resolution`
  int* table;                         // dense voxel index -> cell id (-1: no cell)
  double* cells;                      // kNdtCellDoubles per cell
  int* counter;                       // number of cells
};

struct NdtJob {                       // one derivative pass of one pair
  int slot_in, m;                     // the input cloud (slot) and its point count
  int grid;                           // index of the target's NdtGrid
  int want_hessian;
  Mat4f T;
  NdtAngles ang;
};

// the segmented sort then sorts exactly the clouds that get an NDT grid
__global__ void k_ndt_select_slots(SlotDev* slots, int nslots, const NdtGrid* __restrict__ grids, int ngrids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  int n = 0;
  for (int g = 0; g < ngrids; ++g) n = grids[g].slot == i ? grids[g].n : n;
  slots[i].n_sort = n;
}

__global__ void __launch_bounds__(kBlock) k_ndt_keys(const SlotDev* __restrict__ slots, const NdtGrid* __restrict__ grids,
                                                      const float4* __restrict__ filt, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  const NdtGrid& G = grids[blockIdx.y];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= G.n) return;
  const int off = slots[G.slot].off;
  const float4 p = filt[off + i];
  keys[off + i] = voxel_key(G.vp, p.x, p.y, p.z);
  vals[off + i] = (uint32_t)i;
}

// one thread per sorted element; the head of a run of >= 6 equal keys builds the cell.  Sums run in ascending
// point index (the stable sort keeps that order), in double, as VoxelGridCovariance does.
__global__ void __launch_bounds__(kBlock) k_ndt_cells(const SlotDev* __restrict__ slots, const NdtGrid* __restrict__ grids,
                                                       const float4* __restrict__ filt, const uint32_t* __restrict__ keys_all,
                                                       const uint32_t* __restrict__ vals_all) {
  const NdtGrid& G = grids[blockIdx.y];
  const int n = G.n;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int off = slots[G.slot].off;
  const float4* __restrict__ pts = filt + off;
  const uint32_t* __restrict__ keys = keys_all + off;
  const uint32_t* __restrict__ vals = vals_all + off;
  int* __restrict__ table = G.table;
  double* __restrict__ cells = G.cells;
  int* __restrict__ counter = G.counter;
  const uint32_t key = keys[i];
  if (i > 0 && keys[i - 1] == key) return;
  double s[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  int j = i;
  for (; j < n && keys[j] == key; ++j) {
    const float4 p = pts[vals[j]];
    const double x = p.x, y = p.y, z = p.z;
    s[0] += x; s[1] += y; s[2] += z;
    c[0] += x * x; c[1] += x * y; c[2] += x * z; c[3] += y * y; c[4] += y * z; c[5] += z * z;
  }
  const int np = j - i;
  if (np < 6) return;
  const double inv_n = 1.0 / (double)np, inv_n1 = 1.0 / ((double)np - 1.0);
  const double m0 = s[0] * inv_n, m1 = s[1] * inv_n, m2 = s[2] * inv_n;
  double a00 = (c[0] - s[0] * m0) * inv_n1, a01 = (c[1] - s[0] * m1) * inv_n1, a02 = (c[2] - s[0] * m2) * inv_n1;
  double a11 = (c[3] - s[1] * m1) * inv_n1, a12 = (c[4] - s[1] * m2) * inv_n1, a22 = (c[5] - s[2] * m2) * inv_n1;
  double ev[3], V[9];
  sym3_eig_desc(a00, a01, a02, a11, a12, a22, ev, V);
  if (ev[2] < -1e-12 || ev[1] < -1e-12 || !(ev[0] > 0.0)) return;
  const double floor_ev = 0.01 * ev[0];
  if (ev[2] < floor_ev) {
    ev[2] = floor_ev;
    if (ev[1] < floor_ev) ev[1] = floor_ev;
    double r[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) r[a * 3 + b] = V[a * 3] * ev[0] * V[b * 3] + V[a * 3 + 1] * ev[1] * V[b * 3 + 1] + V[a * 3 + 2] * ev[2] * V[b * 3 + 2];
    a00 = r[0]; a01 = r[1]; a02 = r[2]; a11 = r[4]; a12 = r[5]; a22 = r[8];
  }
  // symmetric cofactor inverse
  const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double id = 1.0 / det;
  const double i00 = c00 * id, i01 = c01 * id, i02 = c02 * id;
  const double i11 = (a00 * a22 - a02 * a02) * id, i12 = (a01 * a02 - a00 * a12) * id, i22 = (a00 * a11 - a01 * a01) * id;
  if (!(isfinite(i00) && isfinite(i01) && isfinite(i02) && isfinite(i11) && isfinite(i12) && isfinite(i22))) return;
  const int id_cell = atomicAdd(counter, 1);   // (cell ids are arbitrary: look-ups go through the dense table)
  double* o = cells + (size_t)id_cell * kNdtCellDoubles;
  o[0] = m0; o[1] = m1; o[2] = m2;
  o[3] = i00; o[4] = i01; o[5] = i02; o[6] = i11; o[7] = i12; o[8] = i22; o[9] = 0.0;
  table[key] = id_cell;
}

// score, gradient and Hessian of the NDT objective at the transform T (parameters enter through the angle
// derivatives): every input point against the cells whose centroid lies within `resolution` of its image —
// the kd-tree radius query of PCL, answered here by the 27 voxels around the point in the dense table.
// DIRECT7 (NDT_OMP: pclomp::NormalDistributionsTransform with its default neighbour search,
// VoxelGridCovariance::getNeighborhoodAtPoint7): the voxel that holds the image of the point - floor(x / leaf), the
// division pclomp writes - and its six face neighbours, in pclomp's order, every one of them that is a cell; no radius.
template <bool DIRECT7>
__global__ void __launch_bounds__(kBlock) s3d_ndt_derivatives_kernel(const SlotDev* __restrict__ slots,
                                                                      const NdtGrid* __restrict__ grids,
                                                                      const NdtJob* __restrict__ jobs,
                                                                      const float4* __restrict__ filt, float r2, float leaf,
                                                                      double d1, double d2,
                                                                      double* __restrict__ partials) {
  const NdtJob& Jb = jobs[blockIdx.y];
  const NdtGrid& G = grids[Jb.grid];
  const float4* __restrict__ input = filt + slots[Jb.slot_in].off;
  const int m = Jb.m;
  const Mat4f T = Jb.T;
  const NdtAngles& ang = Jb.ang;
  const VoxelParams vp = G.vp;
  const int* __restrict__ table = G.table;
  const double* __restrict__ cells = G.cells;
  const int want_hessian = Jb.want_hessian;
  double acc[NDT_NACC];
#pragma unroll
  for (int c = 0; c < NDT_NACC; ++c) acc[c] = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
    const float4 pf = input[i];
    const F3 xt = xf_pcl(T, pf.x, pf.y, pf.z);
    const int c0 = DIRECT7 ? (int)floorf(xt.x / leaf) - vp.min_b[0] : (int)(floorf(xt.x * vp.inv_leaf) - (float)vp.min_b[0]);
    const int c1 = DIRECT7 ? (int)floorf(xt.y / leaf) - vp.min_b[1] : (int)(floorf(xt.y * vp.inv_leaf) - (float)vp.min_b[1]);
    const int c2 = DIRECT7 ? (int)floorf(xt.z / leaf) - vp.min_b[2] : (int)(floorf(xt.z * vp.inv_leaf) - (float)vp.min_b[2]);
    const double x[3] = {pf.x, pf.y, pf.z};
    double J[3][3];                       // columns 3..5 of point_gradient_
    double Hx[6][3];                      // d2R[kl] x
    bool have = false;
    for (int nb = 0; nb < (DIRECT7 ? 7 : 27); ++nb) {
        {
          // 27: dz outermost, dx innermost; 7: (0,0,0) (+x) (-x) (+y) (-y) (+z) (-z)
          const int dx = DIRECT7 ? (nb == 1 ? 1 : (nb == 2 ? -1 : 0)) : nb % 3 - 1;
          const int dy = DIRECT7 ? (nb == 3 ? 1 : (nb == 4 ? -1 : 0)) : (nb / 3) % 3 - 1;
          const int dz = DIRECT7 ? (nb == 5 ? 1 : (nb == 6 ? -1 : 0)) : nb / 9 - 1;
          const int v0 = c0 + dx, v1 = c1 + dy, v2 = c2 + dz;
          if (v0 < 0 || v1 < 0 || v2 < 0 || v0 >= vp.div_b[0] || v1 >= vp.div_b[1] || v2 >= vp.div_b[2]) continue;
          const int cid = table[v0 + v1 * vp.div_b[0] + v2 * vp.div_b[0] * vp.div_b[1]];
          if (cid < 0) continue;
          const double* __restrict__ cell = cells + (size_t)cid * kNdtCellDoubles;
          const double mu0 = cell[0], mu1 = cell[1], mu2 = cell[2];
          // radius test on the float centroid, float arithmetic, strict (FLANN radius search)
          if (!DIRECT7 && !(dist2(xt.x, xt.y, xt.z, (float)mu0, (float)mu1, (float)mu2) < r2)) continue;
          if (!have) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int a = 0; a < 3; ++a)
                J[a][k] = ang.dR[k][a * 3] * x[0] + ang.dR[k][a * 3 + 1] * x[1] + ang.dR[k][a * 3 + 2] * x[2];
            if (want_hessian) {
#pragma unroll
              for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a)
                  Hx[k][a] = ang.d2R[k][a * 3] * x[0] + ang.d2R[k][a * 3 + 1] * x[1] + ang.d2R[k][a * 3 + 2] * x[2];
            }
            have = true;
          }
          const double q0 = (double)xt.x - mu0, q1 = (double)xt.y - mu1, q2 = (double)xt.z - mu2;
          const double C00 = cell[3], C01 = cell[4], C02 = cell[5], C11 = cell[6], C12 = cell[7], C22 = cell[8];
          const double Cx0 = C00 * q0 + C01 * q1 + C02 * q2, Cx1 = C01 * q0 + C11 * q1 + C12 * q2,
                       Cx2 = C02 * q0 + C12 * q1 + C22 * q2;
          double e = exp(-d2 * (q0 * Cx0 + q1 * Cx1 + q2 * Cx2) / 2);
          const double score_inc = -d1 * e;
          e = d2 * e;
          if (e > 1 || e < 0 || e != e) continue;
          acc[0] += score_inc;
          e *= d1;
          // C J for the six parameter columns (the first three columns of J are the unit vectors)
          double CJ[6][3], xCJ[6];
          CJ[0][0] = C00; CJ[0][1] = C01; CJ[0][2] = C02;
          CJ[1][0] = C01; CJ[1][1] = C11; CJ[1][2] = C12;
          CJ[2][0] = C02; CJ[2][1] = C12; CJ[2][2] = C22;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            CJ[3 + k][0] = C00 * J[0][k] + C01 * J[1][k] + C02 * J[2][k];
            CJ[3 + k][1] = C01 * J[0][k] + C11 * J[1][k] + C12 * J[2][k];
            CJ[3 + k][2] = C02 * J[0][k] + C12 * J[1][k] + C22 * J[2][k];
          }
#pragma unroll
          for (int col = 0; col < 6; ++col) {
            xCJ[col] = q0 * CJ[col][0] + q1 * CJ[col][1] + q2 * CJ[col][2];
            acc[1 + col] += xCJ[col] * e;
          }
          if (!want_hessian) continue;
          int o = 7;
#pragma unroll
          for (int ii = 0; ii < 6; ++ii)
#pragma unroll
            for (int jj = ii; jj < 6; ++jj) {
              double hx = 0.0;
              if (ii >= 3) {   // (jj >= ii >= 3)
                const int a = ii - 3, b = jj - 3;
                const int kl = a == 0 ? b : (a == 1 ? 2 + b : 5);
                hx = Cx0 * Hx[kl][0] + Cx1 * Hx[kl][1] + Cx2 * Hx[kl][2];
              }
              // J_jj . (C J_ii)
              double jcj;
              if (jj < 3) jcj = CJ[ii][jj];
              else jcj = J[0][jj - 3] * CJ[ii][0] + J[1][jj - 3] * CJ[ii][1] + J[2][jj - 3] * CJ[ii][2];
              acc[o++] += e * (-d2 * xCJ[ii] * xCJ[jj] + hx + jcj);
            }
        }
    }
  }
  block_reduce_store<NDT_NACC>(acc, partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NDT_NACC);
}

// fixed-order sum of the block partials of every job
__global__ void k_ndt_reduce(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= NDT_NACC) return;
  out[(size_t)blockIdx.x * NDT_NACC + c] =
      ordered_partial_sum(partials + (size_t)blockIdx.x * nblocks * NDT_NACC + c, nblocks, NDT_NACC);
}

}  // namespace s3d
