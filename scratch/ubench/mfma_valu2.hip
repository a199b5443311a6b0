// Micro-benchmark: does VALU work issue in the shadow of v_mfma_f32_16x16x32_bf16 on gfx950?
//   mode 0: MFMA only (4 independent accumulators, back to back)        mode 1: VALU only (same VALU count as mode 2)
//   mode 2: MFMA + V VALU ops per MFMA, same wave                       mode 3: wave parity: even waves MFMA only, odd waves VALU only
// one workgroup per CU, W waves per SIMD (block = 256 * W threads... waves w and w+4 share a SIMD)
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int V, int MODE, int BIG>
__global__ __launch_bounds__(512) void k(int iters, float* out, unsigned long long* cyc) {
  constexpr int mode = MODE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  f32x16 accb[4];
  for (int i = 0; i < 4; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 16; ++e) accb[i][e] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)(lane * 8 + i);
  const bool do_mfma = mode == 0 || mode == 2 || (mode == 3 && (wave >> 2) == 0);
  const bool do_valu = mode == 1 || mode == 2 || (mode == 3 && (wave >> 2) == 1);
  __syncthreads();
  const unsigned long long t0 = clock64();
  auto body = [&](bool m, bool vv) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (m) {
          if (BIG) accb[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accb[j & 3], 0, 0, 0);
          else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
        }
        if (vv) {
#pragma unroll
          for (int q = 0; q < V; ++q) v[(q + j * V) & 7] = __builtin_fmaf(v[(q + j * V) & 7], 1.0001f, 0.5f);
        }
      }
    }
  };
  if (do_mfma && do_valu) body(true, true);
  else if (do_mfma) body(true, false);
  else body(false, true);
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accb[i][0] + accb[i][7] + accb[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

extern "C" int ub_run(int big, int V, int mode, int waves_per_simd, int iters, int blocks, float* out, unsigned long long* cyc, void* stream) {
  dim3 g(blocks), b(256 * waves_per_simd);
  #define UB(vv) case vv: \
    if (big) { \
    if (mode == 0) hipLaunchKernelGGL((k<vv, 0, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else if (mode == 1) hipLaunchKernelGGL((k<vv, 1, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else if (mode == 2) hipLaunchKernelGGL((k<vv, 2, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else hipLaunchKernelGGL((k<vv, 3, 1>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    } else { \
    if (mode == 0) hipLaunchKernelGGL((k<vv, 0, 0>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else if (mode == 1) hipLaunchKernelGGL((k<vv, 1, 0>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else if (mode == 2) hipLaunchKernelGGL((k<vv, 2, 0>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    else hipLaunchKernelGGL((k<vv, 3, 0>), g, b, 0, (hipStream_t)stream, iters, out, cyc); \
    } \
    break;
  switch (V) {
    UB(0) UB(1) UB(2) UB(3) UB(4) UB(6) UB(8) UB(12) UB(16)
    default: return -1;
  }
  return (int)hipGetLastError();
}
