// Micro-benchmark: issue-to-issue cycles of v_mfma_f32_16x16x32_bf16 when the same accumulator is reused every NACC-th MFMA
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 24; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j % NACC], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
extern "C" int ub_dep(int nacc, int iters, int blocks, float* out, unsigned long long* cyc, void* stream) {
  dim3 g(blocks), b(256);
  switch (nacc) {
    case 1: hipLaunchKernelGGL(k<1>, g, b, 0, (hipStream_t)stream, iters, out, cyc); break;
    case 2: hipLaunchKernelGGL(k<2>, g, b, 0, (hipStream_t)stream, iters, out, cyc); break;
    case 3: hipLaunchKernelGGL(k<3>, g, b, 0, (hipStream_t)stream, iters, out, cyc); break;
    case 4: hipLaunchKernelGGL(k<4>, g, b, 0, (hipStream_t)stream, iters, out, cyc); break;
    case 6: hipLaunchKernelGGL(k<6>, g, b, 0, (hipStream_t)stream, iters, out, cyc); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
