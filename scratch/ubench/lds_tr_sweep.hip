// ds_read_b64_tr_b16 / ds_read_b64 rate against the row stride and the lane-group -> row assignment of a position-major tile
// (lane (t16, g) reads 8 bytes at row rowmap(g) + (t16 >> 2), byte (t16 & 3) * 8): clocks per wave instruction per CU, 8 waves.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ubench/lds_tr_sweep scratch/ubench/lds_tr_sweep.hip && scratch/ubench/lds_tr_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define ITER 2048

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int rs, int map, int colstep) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 16 * 1024; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const int t16 = lane & 15, g = lane >> 4;
  // map 0: groups take rows 0-3, 4-7, 8-11, 12-15;  1: 0-3, 4-7, 12-15, 8-11;  2: 0-3, 8-11, 4-7, 12-15;  3: rows g, g+4, g+8, g+12
  int row;
  if (map == 0) row = g * 4 + (t16 >> 2);
  else if (map == 1) row = (g == 2 ? 12 : g == 3 ? 8 : g * 4) + (t16 >> 2);
  else if (map == 2) row = (g == 1 ? 8 : g == 2 ? 4 : g * 4) + (t16 >> 2);
  else row = g + 4 * (t16 >> 2);
  const unsigned base = (unsigned)(size_t)smem + wave * 40 * rs / 4 * 0 + row * rs + (t16 & 3) * 8;
  unsigned acc = 0;
  for (int it = 0; it < ITER; ++it) {
    const unsigned a = base + (it & 7) * colstep;
    u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
    if (MODE == 0)
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n ds_read_b64_tr_b16 %1, %8 offset:32\n ds_read_b64_tr_b16 %2, %8 offset:64\n ds_read_b64_tr_b16 %3, %8 offset:96\n"
          "ds_read_b64_tr_b16 %4, %8 offset:128\n ds_read_b64_tr_b16 %5, %8 offset:160\n ds_read_b64_tr_b16 %6, %8 offset:192\n ds_read_b64_tr_b16 %7, %8 offset:224\n"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
    else
      asm volatile(
          "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:32\n ds_read_b64 %2, %8 offset:64\n ds_read_b64 %3, %8 offset:96\n"
          "ds_read_b64 %4, %8 offset:128\n ds_read_b64 %5, %8 offset:160\n ds_read_b64 %6, %8 offset:192\n ds_read_b64 %7, %8 offset:224\n"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
    acc += v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
  }
  if (acc == 0x12345u) out[t] = (float)acc;
}

template <int MODE>
static double run(int rs, int map) {
  static float* out = nullptr;
  if (!out) (void)hipMalloc(&out, 4096);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, out, rs, map, 0);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  const double instr = 512.0 * 4 * ITER * 8;
  return (ms * 1e-3) * 2.4e9 * 256 / instr;
}

int main() {
  printf("clocks per wave instruction per CU (2.4 GHz), 8 waves per CU; columns: row map 0 / 1 / 2 / 3\n");
  for (int rs : {288, 320, 352, 272, 304, 264, 280, 296, 312, 160, 224, 416, 544, 200, 328}) {
    printf("row stride %3d B (%3d mod 256): tr_b16", rs, rs % 256);
    for (int map = 0; map < 4; ++map) printf(" %5.2f", run<0>(rs, map));
    printf("   b64");
    for (int map = 0; map < 4; ++map) printf(" %5.2f", run<1>(rs, map));
    printf("\n");
  }
  return 0;
}
