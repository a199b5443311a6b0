// Does ds_read_b128 / ds_read_b64 serve addresses that are only 2- / 4- / 8-byte aligned (KFD sets SH_MEM_CONFIG to unaligned mode
// on gfx9), are the bytes right, and at what rate?  Channel-major operand image [channel][position] of bf16: lane (c = lane & 15,
// g = lane >> 4) reads positions p0 + 8g .. + 7 of channel c - a filter-tap shift of the 3x3 weight gradient would be p0 += s.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/ubench/lds_unaligned scratch/ubench/lds_unaligned.hip && scratch/ubench/lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define CS 1088          // bytes per channel row: 64 mod 1024 -> (c, g) -> c*64 + g*16: 64 distinct 16-byte windows
#define ITER 2048

// correctness: LDS holds the u16 index of every element; out[shift][lane][8] = what the lane got
template <int W>
__global__ void check(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[16 * CS];
  for (int i = threadIdx.x; i < 16 * CS / 2; i += 64) reinterpret_cast<unsigned short*>(smem)[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  for (int s = 0; s < 16; ++s) {
    const unsigned addr = (unsigned)(size_t)(smem) + c * CS + (s + (W / 2) * g) * 2;
    unsigned short* o = out + (s * 64 + lane) * 8;
    if (W == 16) {
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      for (int j = 0; j < 4; ++j) { o[2 * j] = v[j] & 0xffff; o[2 * j + 1] = v[j] >> 16; }
    } else {
      u32x2 v;
      asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      for (int j = 0; j < 2; ++j) { o[2 * j] = v[j] & 0xffff; o[2 * j + 1] = v[j] >> 16; }
    }
  }
}

template <int W>
__global__ __launch_bounds__(256, 2) void rate(float* out, int mis, int cs) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 16 * 1024; i += 256) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const int c = lane & 15, g = lane >> 4;
  const unsigned base = (unsigned)(size_t)(smem) + c * cs + g * W + mis;
  unsigned acc = 0;
  for (int it = 0; it < ITER; ++it) {
    const unsigned a = base + ((it & 7) * 64);
    if (W == 16) {
      u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:64\n ds_read_b128 %2, %8 offset:128\n ds_read_b128 %3, %8 offset:192\n"
          "ds_read_b128 %4, %8 offset:256\n ds_read_b128 %5, %8 offset:320\n ds_read_b128 %6, %8 offset:384\n ds_read_b128 %7, %8 offset:448\n"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
      acc += v0[0] ^ v1[1] ^ v2[2] ^ v3[3] ^ v4[0] ^ v5[1] ^ v6[2] ^ v7[3];
    } else {
      u32x2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:64\n ds_read_b64 %2, %8 offset:128\n ds_read_b64 %3, %8 offset:192\n"
          "ds_read_b64 %4, %8 offset:256\n ds_read_b64 %5, %8 offset:320\n ds_read_b64 %6, %8 offset:384\n ds_read_b64 %7, %8 offset:448\n"
          "s_waitcnt lgkmcnt(0)"
          : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a) : "memory");
      acc += v0[0] ^ v1[1] ^ v2[0] ^ v3[1] ^ v4[0] ^ v5[1] ^ v6[0] ^ v7[1];
    }
  }
  if (acc == 0x12345u) out[t] = (float)acc;
}

template <int W>
static void run_check() {
  unsigned short* out;
  hipMalloc(&out, 16 * 64 * 8 * 2);
  hipMemset(out, 0xff, 16 * 64 * 8 * 2);
  hipLaunchKernelGGL(check<W>, dim3(1), dim3(64), 0, 0, out);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("ds_read_b%d unaligned: kernel failed: %s\n", W * 8, hipGetErrorString(e)); exit(1); }
  unsigned short* h = (unsigned short*)malloc(16 * 64 * 8 * 2);
  hipMemcpy(h, out, 16 * 64 * 8 * 2, hipMemcpyDeviceToHost);
  for (int s = 0; s < 16; ++s) {
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < W / 2; ++j) {
        const int want = (lane & 15) * (CS / 2) + s + (W / 2) * (lane >> 4) + j;
        if (h[(s * 64 + lane) * 8 + j] != (unsigned short)want) ++bad;
      }
    printf("ds_read_b%d  shift %2d elements (%2d bytes): %s", W * 8, s, 2 * s, bad ? "WRONG" : "ok");
    if (bad) printf(" (%d mismatches; lane 0 got %u %u %u %u)", bad, h[s * 64 * 8], h[s * 64 * 8 + 1], h[s * 64 * 8 + 2], h[s * 64 * 8 + 3]);
    printf("\n");
  }
  free(h);
  hipFree(out);
}

template <int W>
static void run_rate(int mis, int cs) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL(rate<W>, dim3(512), dim3(256), 0, 0, out, mis, cs);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = 512.0 * 4 * ITER * 8;
  const double total = instr * 64 * W;
  printf("ds_read_b%-3d channel stride %4d B, misaligned by %2d B: %.3f ms  %.1f B/clk/CU, %.2f clk per wave instruction per CU (2.4 GHz)\n", W * 8, cs, mis, ms,
         total / (ms * 1e-3) / 256 / 2.4e9, (ms * 1e-3) * 2.4e9 * 256 / instr);
  hipFree(out);
}

int main() {
  run_check<16>();
  run_check<8>();
  // channel strides: 1088 = 68 x 16 (lanes (c, g) -> window 4c + g: 64 distinct windows of 1024 B), 400 / 432 / 144 = odd x 16
  // (window c * odd + g: distinct per g group), 392 / 456 = 8 mod 64 (rows not 16-byte aligned)
  for (int cs : {1088, 400, 432, 144, 392, 456})
    for (int mis : {0, 2, 4, 8, 6}) run_rate<16>(mis, cs);
  for (int cs : {1088, 400})
    for (int mis : {0, 2, 4}) run_rate<8>(mis, cs);
  return 0;
}
