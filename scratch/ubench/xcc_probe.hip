// which XCD does workgroup b run on?  hipcc --offload-arch=gfx950 -O3 -o xcc_probe xcc_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) o[blockIdx.x] = (int)x;
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d);
  int h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 32; ++i) printf("%d:%x ", i, h[i]);
  int hist[16] = {0}, agree = 0;
  for (int i = 0; i < 1024; ++i) { hist[h[i] & 15]++; agree += (h[i] & 7) == (i & 7); }
  printf("\nhist:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\nblocks with xcc == blockIdx %% 8: %d of 1024\n", agree);
  return 0;
}
