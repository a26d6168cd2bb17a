// dev check (round 5): v_cvt_flr_i32_f32 == (int)floorf(x) for every float the grid code can feed it, saturating outside int.
// build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt_flr tools_dev/ubench/cvt_flr.hip && /tmp/cvt_flr
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const float* a, int* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(a[i]));
  o[i] = r;
}
int main() {
  std::vector<float> h;
  // every float pattern with a stride (all exponents, both signs) + neighbourhoods of integers and half-integers
  for (uint64_t b = 0; b < (1ull << 32); b += 9973) { uint32_t u = (uint32_t)b; float f; memcpy(&f, &u, 4); h.push_back(f); }
  for (int i = -70000; i <= 70000; ++i)
    for (float d : {0.f, 0.5f, -0.5f}) {
      float f = (float)i + d; h.push_back(f); h.push_back(nextafterf(f, 1e30f)); h.push_back(nextafterf(f, -1e30f));
    }
  for (float f : {8388607.5f, 8388608.f, 16777216.f, 2147483520.f, 2147483648.f, -2147483648.f, -2147483904.f, 1e20f, -1e20f})
    h.push_back(f);
  const int n = (int)h.size();
  float* da; int* dout;
  hipMalloc(&da, 4 * (size_t)n); hipMalloc(&dout, 4 * (size_t)n);
  hipMemcpy(da, h.data(), 4 * (size_t)n, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>>(da, dout, n);
  std::vector<int> o(n);
  hipMemcpy(o.data(), dout, 4 * (size_t)n, hipMemcpyDeviceToHost);
  long long bad = 0, nan_nonzero = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    const float f = h[i];
    if (std::isnan(f)) { nan_nonzero += o[i] != 0; continue; }
    const double fl = std::floor((double)f);
    const long long want = fl >= 2147483647.0 ? 2147483647ll : (fl <= -2147483648.0 ? -2147483648ll : (long long)fl);
    if ((long long)o[i] != want) { ++bad; if (shown++ < 10) printf("x %.9g (%08x): got %d want %lld\n", f, *(const uint32_t*)&h[i], o[i], want); }
  }
  printf("%d values: %lld differ from saturating floor; NaN inputs giving non-zero: %lld\n", n, bad, nan_nonzero);
  return bad != 0;
}
