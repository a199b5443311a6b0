// x6 planes (x6p.h): producers / readers that are not fused into another kernel.
//   buctd_x6p_from_nhwc : fp32 NHWC (optionally through the producer's BatchNorm(+ReLU), the expression of bn_apply_kernel)
//                         -> padded, pre-split planes, pad and guard rows zeroed.  HBM-bound: 4 B read + 6 B written / element.
//   buctd_x6p_to_nhwc   : planes -> fp32 NHWC (h + m + l, exact) - tests and debugging.
// The fused producers live next to the arithmetic they extend: bn_apply / bn_bwd_apply (bn.hip) write planes of their
// result when asked to.  Consumers: buctd_conv3x3_bf16x6_p (conv3x3.hip), buctd_conv3x3_wgrad_bf16x6_p (conv3x3_wgrad4.hip).
// Reference call sites served: the operands of every BasicBlock convolution, lib/models/pose_hrnet.py:28-57.
#include "x6p.h"
#include "../../include/buctd_hip.h"

struct X6pGeo {
  int N, H, W, C, SW, IB;
  long P;
  unsigned ib_mul, ib_sh, sw_mul, sw_sh, c8_mul, c8_sh;
};

__device__ __forceinline__ int x6p_fdiv(int n, unsigned mul, unsigned sh) { return (int)(__umulhi((unsigned)n, mul) >> sh); }

static void x6p_magic(unsigned d, unsigned* mul, unsigned* sh) {
  if (d == 1) { *mul = 0xFFFFFFFFu; *sh = 0; return; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  *mul = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  *sh = l - 1;
}

// element offset (channel 0) of padded position pp in the NHWC tensor, or -1 for pad / guard rows
__device__ __forceinline__ long x6p_pixel(const X6pGeo& g, long pp) {
  if (pp < 0 || pp >= g.P) return -1;
  const int n = x6p_fdiv((int)pp, g.ib_mul, g.ib_sh);
  const int rem = (int)pp - n * g.IB;
  const int yy = x6p_fdiv(rem, g.sw_mul, g.sw_sh);
  const int xx = rem - yy * g.SW;
  if (n >= g.N || yy < 1 || xx < 1 || xx > g.W) return -1;
  return ((long)(n * g.H + yy - 1) * g.W + xx - 1) * g.C;
}

__global__ __launch_bounds__(256) void x6p_from_nhwc_kernel(X6pGeo g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int relu, unsigned char* __restrict__ planes, long items) {
  const int c8n = g.C >> 3;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < items; idx += (long)gridDim.x * 256) {
    const int row = c8n == 1 ? (int)idx : x6p_fdiv((int)idx, g.c8_mul, g.c8_sh);
    const int c8 = (int)idx - row * c8n;
    const long pp = (long)row - X6P_GB;
    const long off = x6p_pixel(g, pp);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (off >= 0) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + off + c8 * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(x + off + c8 * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
      if (mean) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c8 * 8 + j;
          v[j] = (v[j] - mean[c]) * (invstd[c] * gamma[c]) + beta[c];
          if (relu) v[j] = fmaxf(v[j], 0.f);
        }
      }
    }
    x6p_u16x8 h, m, l;
    x6p_split8(v, h, m, l);
    x6p_store8(planes + (size_t)row * (size_t)(g.C * 6), c8, h, m, l);
  }
}

__global__ __launch_bounds__(256) void x6p_to_nhwc_kernel(X6pGeo g, const unsigned char* __restrict__ planes,
                                                          float* __restrict__ x, long items) {
  const int c8n = g.C >> 3;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < items; idx += (long)gridDim.x * 256) {
    const int row = c8n == 1 ? (int)idx : x6p_fdiv((int)idx, g.c8_mul, g.c8_sh);
    const int c8 = (int)idx - row * c8n;
    const long off = x6p_pixel(g, (long)row - X6P_GB);
    if (off < 0) continue;
    const unsigned char* q = planes + (size_t)row * (size_t)(g.C * 6) + (c8 >> 1) * 96 + (c8 & 1) * 16;
    const x6p_u16x8 h = *reinterpret_cast<const x6p_u16x8*>(q);
    const x6p_u16x8 m = *reinterpret_cast<const x6p_u16x8*>(q + 32);
    const x6p_u16x8 l = *reinterpret_cast<const x6p_u16x8*>(q + 64);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = (__uint_as_float((unsigned)h[j] << 16) + __uint_as_float((unsigned)m[j] << 16)) +
             __uint_as_float((unsigned)l[j] << 16);
    *reinterpret_cast<f32x4*>(x + off + c8 * 8) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(x + off + c8 * 8 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  }
}

static int x6p_geo(int N, int H, int W, int C, X6pGeo* g, const char* who) {
  BUCTD_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && C % 16 == 0, "%s: N%d H%d W%d C%d (C must be a multiple of 16)", who, N,
                  H, W, C);
  g->N = N; g->H = H; g->W = W; g->C = C; g->SW = W + 2; g->IB = (H + 1) * (W + 2);
  g->P = x6p_positions(N, H, W);
  const long rows = X6P_GB + g->P + X6P_GA;
  BUCTD_CHECK_ARG(rows * (C / 8) < 2147483647L && rows * (long)C * 6 < 2147483647L, "%s: tensor too large", who);
  x6p_magic((unsigned)g->IB, &g->ib_mul, &g->ib_sh);
  x6p_magic((unsigned)g->SW, &g->sw_mul, &g->sw_sh);
  x6p_magic((unsigned)(C / 8), &g->c8_mul, &g->c8_sh);
  return BUCTD_OK;
}

extern "C" size_t buctd_x6p_bytes(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 16 != 0) return 0;
  return x6p_total_bytes(N, H, W, C);
}

extern "C" int buctd_x6p_from_nhwc(int N, int H, int W, int C, const float* x, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, int relu, void* planes, void* stream) {
  X6pGeo g;
  BUCTD_CHECK_ARG(x && planes, "buctd_x6p_from_nhwc: null pointer");
  BUCTD_CHECK_ARG(!mean || (invstd && gamma && beta), "buctd_x6p_from_nhwc: the fused BatchNorm needs all four arrays");
  const int rc = x6p_geo(N, H, W, C, &g, "buctd_x6p_from_nhwc");
  if (rc) return rc;
  const long items = (X6P_GB + g.P + X6P_GA) * (long)(C / 8);
  long blocks = (items + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(x6p_from_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, x, mean, invstd,
                     gamma, beta, relu, (unsigned char*)planes, items);
  BUCTD_CHECK_LAUNCH("buctd_x6p_from_nhwc");
  return BUCTD_OK;
}

extern "C" int buctd_x6p_to_nhwc(int N, int H, int W, int C, const void* planes, float* x, void* stream) {
  X6pGeo g;
  BUCTD_CHECK_ARG(x && planes, "buctd_x6p_to_nhwc: null pointer");
  const int rc = x6p_geo(N, H, W, C, &g, "buctd_x6p_to_nhwc");
  if (rc) return rc;
  const long items = (X6P_GB + g.P + X6P_GA) * (long)(C / 8);
  long blocks = (items + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(x6p_to_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g,
                     (const unsigned char*)planes, x, items);
  BUCTD_CHECK_LAUNCH("buctd_x6p_to_nhwc");
  return BUCTD_OK;
}
