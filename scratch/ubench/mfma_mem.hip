// Micro-benchmark: can one wave issue LDS reads / global loads in the shadow of its own MFMAs?
// per iteration: 16 MFMAs (4 accumulators) + L loads, either bunched in front of the MFMAs or interleaved one per MFMA.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int KIND, int L, int INTER>   // KIND 0 none, 1 ds_read_b128, 2 global_load_dwordx4
__global__ __launch_bounds__(256) void k(int iters, const f32x4* __restrict__ g, float* out, unsigned long long* cyc) {
  __shared__ f32x4 sm[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = (f32x4){1.f, 2.f, 3.f, 4.f};
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  f32x4 v[L > 0 ? L : 1];
  f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int base = (it * 64 + lane) & 1023;
    if (!INTER) {
#pragma unroll
      for (int l = 0; l < L; ++l) v[l] = KIND == 1 ? sm[(base + l * 64) & 1023] : g[(size_t)wave * 65536 + ((base + l * 64) & 65535)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < L) v[j] = KIND == 1 ? sm[(base + j * 64) & 1023] : g[(size_t)wave * 65536 + ((base + j * 64) & 65535)];
        __builtin_amdgcn_sched_barrier(0);
        acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (KIND) {
#pragma unroll
      for (int l = 0; l < L; ++l) sum += v[l];
    }
  }
  const unsigned long long t1 = clock64();
  float s = sum[0] + sum[1] + sum[2] + sum[3];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
extern "C" int ub_mem(int kind, int L, int inter, int iters, int blocks, const void* g, float* out, unsigned long long* cyc, void* stream) {
  dim3 gr(blocks), b(256);
  hipStream_t st = (hipStream_t)stream;
#define RUN(K, LL, I) if (kind == K && L == LL && inter == I) { hipLaunchKernelGGL((k<K, LL, I>), gr, b, 0, st, iters, (const f32x4*)g, out, cyc); return (int)hipGetLastError(); }
  RUN(0, 0, 0)
  RUN(1, 4, 0) RUN(1, 4, 1) RUN(1, 8, 0) RUN(1, 8, 1) RUN(1, 12, 0) RUN(1, 12, 1)
  RUN(2, 4, 0) RUN(2, 4, 1) RUN(2, 8, 0) RUN(2, 8, 1) RUN(2, 12, 0) RUN(2, 12, 1)
  return -1;
}
