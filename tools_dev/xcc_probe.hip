// dev probe: which XCD does block b of a launch land on?  (HW_REG_XCC_ID, gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 15u);
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  int h[4096];
  for (int launch = 0; launch < 4; ++launch) {
    const int nb = launch == 2 ? 37 : 64;   // an odd-sized launch in between: does the start XCD rotate?
    probe<<<nb, 256>>>(d);
    hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
    printf("launch %d (%d blocks):", launch, nb);
    for (int b = 0; b < (nb < 24 ? nb : 24); ++b) printf(" %d", h[b]);
    printf("\n");
  }
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  probe<<<64, 256, 0, st>>>(d);
  hipStreamSynchronize(st);
  hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
  printf("own stream:");
  for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
  printf("\n");
  return 0;
}
