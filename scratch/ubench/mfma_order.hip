// Micro-benchmark: the bf16x6 inner block (3 A pieces x 3 NF x 3 B pieces -> 18 MFMAs into 3 accumulators) in two issue orders
//   order 0: product-major (as shipped until now): for product: for nf  -> an accumulator is reused every 3rd MFMA
//   order 1: accumulator-major: for nf: for product                    -> six consecutive MFMAs into one accumulator
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int ORDER, int MF>
__global__ __launch_bounds__(256) void k(int iters, float* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[MF][3];
  for (int m = 0; m < MF; ++m) for (int i = 0; i < 3; ++i) acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3], b[3][3];
  for (int q = 0; q < 3; ++q) for (int i = 0; i < 8; ++i) {
    a[q][i] = (__bf16)(float)(lane + i + q);
    for (int n = 0; n < 3; ++n) b[q][n][i] = (__bf16)(float)(lane - i + q * 3 + n);
  }
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#define MMA(qa, qb, nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa], b[qb][nf], acc[mf][nf], 0, 0, 0);
      if (ORDER == 0) {
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(2, 0, nf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(0, 2, nf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(1, 1, nf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(1, 0, nf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(0, 1, nf)
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) MMA(0, 0, nf)
      } else {
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) { MMA(2, 0, nf) MMA(0, 2, nf) MMA(1, 1, nf) MMA(1, 0, nf) MMA(0, 1, nf) MMA(0, 0, nf) }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int m = 0; m < MF; ++m) for (int i = 0; i < 3; ++i) s += acc[m][i][0] + acc[m][i][1] + acc[m][i][2] + acc[m][i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
extern "C" int ub_order(int order, int mf, int iters, int blocks, float* out, unsigned long long* cyc, void* stream) {
  dim3 g(blocks), b(256);
  if (mf == 4) { if (order) hipLaunchKernelGGL((k<1, 4>), g, b, 0, (hipStream_t)stream, iters, out, cyc); else hipLaunchKernelGGL((k<0, 4>), g, b, 0, (hipStream_t)stream, iters, out, cyc); }
  else { if (order) hipLaunchKernelGGL((k<1, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); else hipLaunchKernelGGL((k<0, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); }
  return (int)hipGetLastError();
}
