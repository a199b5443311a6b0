// cost of the legacy v_mfma_f32_16x16x16_bf16 (K = 16) next to v_mfma_f32_16x16x32_bf16 on gfx950: is a K = 16 tail cheaper?
// hipcc --offload-arch=gfx950 -O3 -o mfma_k16 mfma_k16.hip && ./mfma_k16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K16>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8 a, b; s16x4 a4, b4;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
  for (int e = 0; e < 4; ++e) { a4[e] = (short)(threadIdx.x + e); b4[e] = (short)(threadIdx.x * 3 + e); }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (K16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
template <int K16>
static void run() {
  float* out; hipMalloc(&out, 4);
  const int iters = 1024, grid = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<K16>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k<K16>, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms / 200 * 1e-3 * 2.3e9 / (2.0 * iters * 8 * 4);    // two waves per SIMD
  printf("%s: %.1f us per launch, ~%.1f cycles per MFMA per SIMD at 2.3 GHz\n", K16 ? "16x16x16 bf16_1k" : "16x16x32 bf16", ms / 200 * 1e3, per);
  hipFree(out);
}
int main() { run<0>(); run<1>(); run<0>(); run<1>(); return 0; }
