// Micro-benchmark: what does a (taken / not taken) scalar branch cost a wavefront whose SIMD partner streams MFMAs?
//   waves 0-3: MFMA only (16x16x32 bf16, 4 accumulators);  waves 4-7 (same SIMDs): V VALU + one branch per group
// hipcc --offload-arch=gfx950 -O3 -o branch_cost branch_cost.hip && ./branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V, int BR, bool PARTNER_MFMA>
__global__ __launch_bounds__(512) void k(int iters, int zero, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)(lane * 8 + i);
  __syncthreads();
  const unsigned long long t0 = clock64();
  if (wave < 4) {
    if (PARTNER_MFMA) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
      }
    } else {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[(q + j * 4) & 7] = __builtin_fmaf(v[(q + j * 4) & 7], 1.0001f, 0.5f);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int q = 0; q < V; ++q) v[(q + j * V) & 7] = __builtin_fmaf(v[(q + j * V) & 7], 1.0001f, 0.5f);
        if (BR == 1) asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" ::"s"(zero) : "scc");        // taken
        if (BR == 2) asm volatile("s_cmp_eq_u32 %0, 1\n s_cbranch_scc1 1f\n s_nop 0\n1:" ::"s"(zero) : "scc");        // not taken
        if (BR == 3) asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n s_nop 0\n1: s_or_b64 exec, exec, s[20:21]" ::"v"(lane), "v"(lane) : "vcc", "s20", "s21");   // exec-mask if, not skipped
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V, int BR, bool PM>
static void run(const char* what) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 400;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<V, BR, PM>), dim3(256), dim3(512), 0, 0, iters, 0, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0, vv = 0;
  for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : vv) += (double)h[b * 8 + w];
  m /= 1024; vv /= 1024;
  printf("%-44s V=%2d: partner wave %7.1f cycles per 16-slot group | VALU wave %7.1f cycles per group of (%d VALU + branch)\n", what, V,
         m / (iters * 4.0), vv / (iters * 16.0), V);
  hipFree(out); hipFree(cyc);
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int sel = argc > 1 ? atoi(argv[1]) : -1;
  int id = 0;
#define RUN(...) do { if (sel < 0 || sel == id) { __VA_ARGS__; } ++id; } while (0)
  RUN(run<4, 0, true>("no branch, partner MFMA"));
  RUN(run<4, 1, true>("taken s_cbranch, partner MFMA"));
  RUN(run<4, 2, true>("not-taken s_cbranch, partner MFMA"));
  RUN(run<4, 3, true>("exec-mask if (not skipped), partner MFMA"));
  RUN(run<16, 0, true>("no branch, partner MFMA"));
  RUN(run<16, 1, true>("taken s_cbranch, partner MFMA"));
  RUN(run<16, 2, true>("not-taken s_cbranch, partner MFMA"));
  RUN(run<16, 3, true>("exec-mask if (not skipped), partner MFMA"));
  RUN(run<4, 0, false>("no branch, partner VALU"));
  RUN(run<4, 1, false>("taken s_cbranch, partner VALU"));
  RUN(run<4, 2, false>("not-taken s_cbranch, partner VALU"));
  RUN(run<16, 1, false>("taken s_cbranch, partner VALU"));
  return 0;
}
