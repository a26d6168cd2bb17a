// dev tool: what HBM streaming rate does this MI355X actually deliver?  read-only (float4 / float), read+write copy,
// and a K6-like mix (12 + 12 + 16 + 16 bytes per element from four arrays), one element per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct V3 { float x, y, z; };
__global__ void __launch_bounds__(256) k_read4(const float4* __restrict__ a, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  for (; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}
template <int U>
__global__ void __launch_bounds__(256) k_read4u(const float4* __restrict__ a, size_t n, float* out) {
  size_t i = ((size_t)blockIdx.x * 256) * U + threadIdx.x;
  float s = 0.f;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n) ? a[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_mix(const V3* __restrict__ a, const V3* __restrict__ b, const float4* __restrict__ c,
                                             const float4* __restrict__ d, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  V3 p = a[i], q = b[i]; float4 m = c[i], o = d[i];
  float s = p.x + p.y + p.z + q.x + q.y + q.z + m.x + m.y + m.z + m.w + o.x + o.y + o.z + o.w;
  if (s == 123.456f) out[0] = s;
}
// the settled K5 pass: 12 + 4 + 12 read, 4 written
__global__ void __launch_bounds__(256) k_reval(const V3* __restrict__ a, const V3* __restrict__ b, float* __restrict__ lb, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  V3 p = a[i], q = b[i]; float l = lb[i];
  lb[i] = l - (p.x - q.x) * (p.y - q.y) * (p.z - q.z);
}

struct M4 { float m[16]; };
struct PairB { int slot_s, slot_t, corr_off, active, converged, iterations, correspondences, inner, evals; M4 guess, T, prev, final_T, T_nn; double fitness; int fc, pad; };
struct SlotB { int off, n; float misc[30]; };
__device__ __forceinline__ void xf(const M4& m, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = m.m[0] * x + m.m[4] * y + m.m[8] * z + m.m[12];
  oy = m.m[1] * x + m.m[5] * y + m.m[9] * z + m.m[13];
  oz = m.m[2] * x + m.m[6] * y + m.m[10] * z + m.m[14];
}
// the settled pass as the product runs it: block -> (pair, chunk), pair / slot records by scalar loads, then the stream
template <int CH>
__global__ void __launch_bounds__(256) k_reval_pairs(const PairB* __restrict__ pairs, const SlotB* __restrict__ slots,
                                                     const V3* __restrict__ sorted3, const V3* __restrict__ corr_q,
                                                     float* __restrict__ lb, int chunks_per_pair, int npairs, int* fail) {
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int pair = (slot / chunks_per_pair) * 8 + xcd, chunk0 = (slot % chunks_per_pair) * CH;
  if (pair >= npairs) return;
  const PairB& P = pairs[pair];
  if (!P.active) return;
  const SlotB& St = slots[P.slot_t];
  V3 p0[CH], ps[CH]; float l[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = (chunk0 + c) * 256 + threadIdx.x;
    const int j = i < St.n ? i : 0;
    p0[c] = sorted3[St.off + j]; ps[c] = corr_q[P.corr_off + j]; l[c] = lb[P.corr_off + j];
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = (chunk0 + c) * 256 + threadIdx.x;
    if (i >= St.n) continue;
    float gx, gy, gz, qx, qy, qz, ox, oy, oz;
    xf(P.guess, p0[c].x, p0[c].y, p0[c].z, gx, gy, gz);
    xf(P.T, gx, gy, gz, qx, qy, qz);
    xf(P.T_nn, gx, gy, gz, ox, oy, oz);
    const float move = sqrtf((qx - ox) * (qx - ox) + (qy - oy) * (qy - oy) + (qz - oz) * (qz - oz));
    const float d = sqrtf((qx - ps[c].x) * (qx - ps[c].x) + (qy - ps[c].y) * (qy - ps[c].y) + (qz - ps[c].z) * (qz - ps[c].z));
    if (d + move < fabsf(l[c]) * 0.99999f - 1e-6f) lb[P.corr_off + i] = fabsf(l[c]) - move;
    else atomicAdd(fail, 1);
  }
}
int main() {
  const size_t n = 25600000;   // elements (one 256-pair pass)
  float4 *a, *b; V3 *c, *d; float* out; float* lb;
  CHK(hipMalloc(&a, n * 16)); CHK(hipMalloc(&b, n * 16)); CHK(hipMalloc(&c, n * 12)); CHK(hipMalloc(&d, n * 12));
  CHK(hipMalloc(&out, 16)); CHK(hipMalloc(&lb, n * 4));
  CHK(hipMemset(a, 0, n * 16)); CHK(hipMemset(b, 0, n * 16)); CHK(hipMemset(c, 0, n * 12)); CHK(hipMemset(d, 0, n * 12)); CHK(hipMemset(lb, 0, n * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms  %6.2f TB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
  };
  const unsigned nb = (unsigned)((n + 255) / 256);
  {  // a footprint far beyond the 256 MB of the memory-side cache: 4 GB read once per launch
    const size_t nbig = (size_t)256 << 20;
    float4* big; CHK(hipMalloc(&big, nbig * 16)); CHK(hipMemset(big, 0, nbig * 16));
    run("read float4, 4 GB footprint", nbig * 16.0, [&] { k_read4u<1><<<(unsigned)(nbig / 256), 256>>>(big, nbig, out); });
    run("read float4, 1 GB of it", (nbig / 4) * 16.0, [&] { k_read4u<1><<<(unsigned)(nbig / 4 / 256), 256>>>(big, nbig / 4, out); });
    run("read float4, 256 MB of it", (nbig / 16) * 16.0, [&] { k_read4u<1><<<(unsigned)(nbig / 16 / 256), 256>>>(big, nbig / 16, out); });
    run("read float4, 128 MB of it", (nbig / 32) * 16.0, [&] { k_read4u<1><<<(unsigned)(nbig / 32 / 256), 256>>>(big, nbig / 32, out); });
    CHK(hipFree(big));
  }
  run("read float4, 1 per thread", n * 16.0, [&] { k_read4<<<nb, 256>>>(a, n, out); });
  run("read float4, 2 per thread", n * 16.0, [&] { k_read4u<2><<<(nb + 1) / 2, 256>>>(a, n, out); });
  run("read float4, 4 per thread", n * 16.0, [&] { k_read4u<4><<<(nb + 3) / 4, 256>>>(a, n, out); });
  run("read float4, 8 per thread", n * 16.0, [&] { k_read4u<8><<<(nb + 7) / 8, 256>>>(a, n, out); });
  run("read float4, grid-stride 2048 blk", n * 16.0, [&] { k_read4<<<2048, 256>>>(a, n, out); });
  run("read float4, grid-stride 8192 blk", n * 16.0, [&] { k_read4<<<8192, 256>>>(a, n, out); });
  run("copy float4 (r+w)", n * 32.0, [&] { k_copy4<<<nb, 256>>>(a, b, n); });
  run("K6 mix 12+12+16+16 read", n * 56.0, [&] { k_mix<<<nb, 256>>>(c, d, a, b, n, out); });
  run("K5 settled 12+12+4 r, 4 w", n * 32.0, [&] { k_reval<<<nb, 256>>>(c, d, lb, n); });

  {
    const int NP = 256, NQ = 100000;
    std::vector<PairB> hp(NP); std::vector<SlotB> hs(2 * NP);
    for (int p = 0; p < NP; ++p) {
      PairB& P = hp[p]; memset(&P, 0, sizeof(P));
      P.slot_s = 2 * p; P.slot_t = 2 * p + 1; P.corr_off = p * NQ; P.active = 1;
      for (int k = 0; k < 4; ++k) { P.guess.m[5 * k] = 1.f; P.T.m[5 * k] = 1.f; P.T_nn.m[5 * k] = 1.f; }
      hs[2 * p].off = p * NQ; hs[2 * p].n = NQ; hs[2 * p + 1].off = p * NQ; hs[2 * p + 1].n = NQ;
    }
    PairB* dp; SlotB* ds; int* fail;
    CHK(hipMalloc(&dp, NP * sizeof(PairB))); CHK(hipMalloc(&ds, 2 * NP * sizeof(SlotB))); CHK(hipMalloc(&fail, 4));
    CHK(hipMemcpy(dp, hp.data(), NP * sizeof(PairB), hipMemcpyHostToDevice));
    CHK(hipMemcpy(ds, hs.data(), 2 * NP * sizeof(SlotB), hipMemcpyHostToDevice));
    std::vector<float> one(n, 1.0f); CHK(hipMemcpy(lb, one.data(), n * 4, hipMemcpyHostToDevice));
    const int chunks = (NQ + 255) / 256;
    run("settled pass, pairs, 1 chunk/blk", n * 32.0, [&] { k_reval_pairs<1><<<NP * chunks, 256>>>(dp, ds, c, d, lb, chunks, NP, fail); });
    {  // the same pass with 1.4 GB of other traffic between two launches (as the accumulate kernel does in the product)
      float4* fl; CHK(hipMalloc(&fl, (size_t)90000000 * 16)); CHK(hipMemset(fl, 0, (size_t)90000000 * 16));
      hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
      float tot = 0.f;
      for (int it = 0; it < 12; ++it) {
        k_read4u<1><<<(unsigned)(90000000 / 256), 256>>>(fl, 90000000, out);
        hipEventRecord(a0, 0);
        k_reval_pairs<1><<<NP * chunks, 256>>>(dp, ds, c, d, lb, chunks, NP, fail);
        hipEventRecord(a1, 0); hipEventSynchronize(a1);
        float ms; hipEventElapsedTime(&ms, a0, a1);
        if (it >= 2) tot += ms;
      }
      printf("%-34s %8.3f ms  %6.2f TB/s\n", "settled pass after 1.4 GB of reads", tot / 10, n * 32.0 / (tot / 10 * 1e-3) / 1e12);
      CHK(hipFree(fl));
    }
    {  // sustained: 3000 launches back to back (0.5 s), the mean of the last 1000
      for (int i = 0; i < 2000; ++i) k_reval_pairs<1><<<NP * chunks, 256>>>(dp, ds, c, d, lb, chunks, NP, fail);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 1000; ++i) k_reval_pairs<1><<<NP * chunks, 256>>>(dp, ds, c, d, lb, chunks, NP, fail);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-34s %8.3f ms  %6.2f TB/s\n", "settled pass, sustained 0.5 s", ms / 1000, n * 32.0 / (ms / 1000 * 1e-3) / 1e12);
    }
    run("settled pass, pairs, 2 chunk/blk", n * 32.0, [&] { k_reval_pairs<2><<<NP * ((chunks + 1) / 2), 256>>>(dp, ds, c, d, lb, (chunks + 1) / 2, NP, fail); });
    run("settled pass, pairs, 4 chunk/blk", n * 32.0, [&] { k_reval_pairs<4><<<NP * ((chunks + 3) / 4), 256>>>(dp, ds, c, d, lb, (chunks + 3) / 4, NP, fail); });
    int hf; CHK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost)); printf("fails %d\n", hf);
  }
  return 0;
}
