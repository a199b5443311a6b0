// LDS read throughput per CU: ds_read_b64_tr_b16 (the transpose read of the weight-gradient kernels) against ds_read_b64 and
// ds_read_b128, 8 waves per CU (two 256-thread workgroups), conflict-free addresses (row stride 288 B = 32 mod 64).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ubench/lds_tr_rate scratch/ubench/lds_tr_rate.hip && scratch/ubench/lds_tr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RS 288
#define ITER 2048

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 16 * 1024; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const int t16 = lane & 15, g = lane >> 4;
  const unsigned char* base = smem + wave * 64 * RS / 4 * 3 + (g * 4 + (t16 >> 2)) * RS + (t16 & 3) * 8;
  float acc = 0.f;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned char* p = base + ((it + u) & 7) * 32 + u * 16 * RS % (20 * RS);
      if (MODE == 0) {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
        acc += (float)v[0] + (float)v[3];
      } else if (MODE == 1) {
        f32x2 v = *reinterpret_cast<const f32x2*>(p);
        acc += v[0] + v[1];
      } else {
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + wave * 8192 + ((it + u) & 7) * 1024 + lane * 16);
        acc += v[0] + v[3];
      }
    }
  }
  if (acc == 12345.f) out[t] = acc;
}

template <int MODE>
static void run(const char* name, int bytes) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double total = 512.0 * 4 * ITER * 8 * 64 * bytes;          // bytes read
  printf("%-22s %.3f ms  %.1f B/clk/CU at 2.4 GHz (%.2f TB/s chip-wide)\n", name, ms, total / (ms * 1e-3) / 256 / 2.4e9, total / (ms * 1e-3) / 1e12);
}

int main() {
  run<0>("ds_read_b64_tr_b16", 8);
  run<1>("ds_read_b64", 8);
  run<2>("ds_read_b128", 16);
  return 0;
}
