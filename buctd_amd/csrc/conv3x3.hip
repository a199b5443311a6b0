// 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3"):
// every fp32 operand is split as hi + lo (two bf16), and a*b ~= hi*hi + hi*lo + lo*hi is accumulated in fp32 by
// three v_mfma_f32_16x16x32_bf16 - ~5x the rate of v_mfma_f32_16x16x4_f32 at ~2^-16 relative product error
// (whole-network heat-map error 1.4e-4..3.5e-4 of full scale, measured; the fp32 engine in conv.hip stays the
// exact path and the default for parity tests).
//
// This is the workhorse of the path: BasicBlock convs of reference lib/models/pose_hrnet.py:28-57, 214 of the
// ~300 convolutions of CoAM-W48 and ~75 % of its FLOPs, forward and (with FLIP) data gradient.
//
// Structure - direct convolution with an LDS-resident input tile instead of 9 separate im2col gathers:
//   * pixels are addressed in a zero-padded flattened space  p = n*IB + (y+1)*SW + (x+1),  SW = W+2,
//     IB = (H+1)*SW : one zero column left/right of every row, one zero row between images.  A filter tap is then
//     a constant shift  (r-1)*SW + (s-1)  of p, valid across row and image boundaries alike;
//   * a workgroup owns BM consecutive p's and all BN output channels; per 32-channel chunk it stages the
//     BM + 2*SW + 2 input rows it needs ONCE (fp32 -> bf16 hi|lo, 160-byte rows: stride = 32 mod 64 bytes makes the
//     four 16-lane groups of ds_read_b128 conflict free) and all 9 taps read shifted rows of that tile;
//   * weights stream through a double-buffered 32-deep B stage (one (tap, chunk) per step);  the 16-channel tail of
//     C = 48 pairs two taps in one K = 32 MFMA (lanes 0-31 feed tap t, lanes 32-63 tap t+1), so no MFMA lanes are
//     wasted on padding;
//   * epilogue as in conv.hip: bias, Welford BN partials (+ per-group valid-row counts, since pad positions are
//     skipped), eval-BN scale/shift, residual, ReLU.
#include "common.h"
#include "../../include/buctd_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define ROWB 160           // bytes per LDS row: 32 bf16 hi | 32 bf16 lo | 32 B pad
#define CK 32              // channels per chunk

struct C3Args {
  const float* x;
  const float* w;
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  float* stats;
  int* counts;
  int N, H, W, Ci, Co;
  int SW, IB, P;       // padded row width, padded image block, total padded positions
  int relu, flip;
  unsigned ib_mul, ib_sh, sw_mul, sw_sh;   // n / d == mulhi(n, mul) >> sh for 0 <= n < 2^31 (Granlund-Montgomery)
};

__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(__umulhi((unsigned)n, mul) >> sh);
}

__device__ __forceinline__ void split_store(unsigned char* row, int c, f32x4 v) {
  // channels c..c+3 of one row: hi at byte 2c, lo at 64 + 2c
  u16x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    const __bf16 l = (__bf16)(v[j] - (float)h);
    hi[j] = __builtin_bit_cast(unsigned short, h);
    lo[j] = __builtin_bit_cast(unsigned short, l);
  }
  *reinterpret_cast<u16x4*>(row + 2 * c) = hi;
  *reinterpret_cast<u16x4*>(row + 64 + 2 * c) = lo;
}

#define MAX_SW 75          // W <= 73: the staged tile is at most BM + 152 rows

template <int MF, int NF, bool FLIP>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_kernel(C3Args p) {
  constexpr int BM = 4 * MF * 16, BN = NF * 16;
  constexpr int PA = (BM + 2 * MAX_SW + 2 + 31) / 32;   // float4 loads per thread for one A chunk (8 per row)
  constexpr int PB = (BN * 8 + 255) / 256;              // float4 loads per thread for one B step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int R = BM + 2 * p.SW + 2;               // staged input rows
  unsigned char* At = smem;                      // [R][ROWB]
  unsigned char* Bt = smem + (size_t)R * ROWB;   // [2][BN][ROWB]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int p0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int halo = p.SW + 1;
  const int c4 = (t & 7) * 4;                    // this thread's 4-channel slot inside a 32-channel chunk

  // global element offset (channel 0) of every staged row this thread fills; -1 = zero row
  int goff[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int row = (t >> 3) + 32 * q;
    const int pp = p0 - halo + row;
    int o = -1;
    if (row < R && pp >= 0 && pp < p.P) {
      const int n = fast_div(pp, p.ib_mul, p.ib_sh);
      const int rem = pp - n * p.IB;
      const int yy = fast_div(rem, p.sw_mul, p.sw_sh);   // 0 = pad row above the image
      const int xx = rem - yy * p.SW;                     // 0 and SW-1 = pad columns
      if (n < p.N && yy >= 1 && xx >= 1 && xx <= p.W) o = ((n * p.H + yy - 1) * p.W + xx - 1) * p.Ci;
    }
    goff[q] = o;
  }

  f32x4 areg[PA], breg[PB];
  auto load_a = [&](int c0, int cw) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      // unconditional load from a clamped (always valid) address + select: a branch around each load would
      // serialise them behind vmcnt(0) waits
      // (the zero-select for pad rows happens at store time so that nothing consumes the load result here)
      const bool ok = goff[q] >= 0 && c4 < cw * 16;
      areg[q] = *reinterpret_cast<const f32x4*>(p.x + (ok ? goff[q] + c0 + c4 : 0));
    }
  };
  auto store_a = [&](int cw) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int row = (t >> 3) + 32 * q;
      const bool ok = goff[q] >= 0 && c4 < cw * 16;
      if (row < R) split_store(At + (size_t)row * ROWB, c4, ok ? areg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
  };
  // B stage: step s of a chunk fills K = 32 reduction slots.  slot k4..k4+3 belongs to unit u = k4/16:
  //   full chunk (cw = 2): (tap s, channels c0 + k4);   tail chunk (cw = 1): (tap 2s + u, channels c0 + k4 % 16)
  auto load_b = [&](int c0, int cw, int s) {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int idx = t + 256 * q;
      const int nl = idx >> 3, n = n0 + nl;
      const int u = c4 >> 4;
      const int tap = cw == 2 ? s : 2 * s + u;
      const int cc = cw == 2 ? c4 : (c4 & 15);
      const bool ok = nl < BN && n < p.Co && tap < 9;
      f32x4 v;
      if (!FLIP) {
        v = *reinterpret_cast<const f32x4*>(p.w + (ok ? ((long)n * 9 + tap) * p.Ci + c0 + cc : 0));
      } else {
        // data gradient: B(n = ci_out, k = (tap, co)) = w[co][8 - tap][ci_out]; here p.Ci is the channel
        // count of the SOURCE tensor dy (= conv Co) and p.Co the output channels (= conv Ci)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p.w[ok ? ((long)(c0 + cc + j) * 9 + (8 - tap)) * p.Co + n : 0];
      }
      breg[q] = v;   // zero-select at store time
    }
  };
  auto store_b = [&](int cw, int s, int buf) {
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int nl = (t + 256 * q) >> 3;
      const int tap = cw == 2 ? s : 2 * s + (c4 >> 4);
      const bool ok = n0 + nl < p.Co && tap < 9;
      if (nl < BN) split_store(Bt + ((size_t)buf * BN + nl) * ROWB, c4, ok ? breg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (p.Ci + CK - 1) / CK;
  load_a(0, p.Ci >= CK ? 2 : 1);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * CK;
    const int cw = (p.Ci - c0 >= CK) ? 2 : 1;   // 16-channel groups in this chunk (Ci % 16 == 0)
    const int nsteps = cw == 2 ? 9 : 5;
    load_b(c0, cw, 0);
    __syncthreads();                             // everybody finished reading the previous A tile / B stages
    store_a(cw);
    store_b(cw, 0, 0);
    if (ch + 1 < nchunks) load_a(c0 + CK, (p.Ci - c0 - CK >= CK) ? 2 : 1);   // in flight during this chunk
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) load_b(c0, cw, s + 1);
      // A row shift of the two 16-lane-pair halves of the wave
      int tap0, tap1, cb0, cb1;
      if (cw == 2) { tap0 = tap1 = s; cb0 = 0; cb1 = 32; }
      else { tap0 = 2 * s; tap1 = 2 * s + 1 < 9 ? 2 * s + 1 : 2 * s; cb0 = cb1 = 0; }
      const int sh0 = (tap0 / 3) * p.SW + tap0 % 3, sh1 = (tap1 / 3) * p.SW + tap1 % 3;
      const int aoff = (g < 2 ? sh0 * ROWB + cb0 : sh1 * ROWB + cb1) + (g & 1) * 16;
      const unsigned char* bb = Bt + (size_t)(s & 1) * BN * ROWB + g * 16;
      bf16x8 bh[NF], bl[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const unsigned char* q = bb + (size_t)(nf * 16 + i16) * ROWB;
        bh[nf] = *reinterpret_cast<const bf16x8*>(q);
        bl[nf] = *reinterpret_cast<const bf16x8*>(q + 64);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const unsigned char* q = At + (size_t)(wave * MF * 16 + mf * 16 + i16) * ROWB + aoff;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(q);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(q + 64);
        // small terms first, then hi*hi; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nf], acc[mf][nf], 0, 0, 0);
      }
      if (s + 1 < nsteps) store_b(cw, s + 1, (s + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  // accumulator (mf, nf, reg): row = wave*MF*16 + mf*16 + (lane>>4)*4 + reg, col = nf*16 + (lane&15)
  long ooff[MF][4];
  bool ok[MF][4];
  int cnt = 0;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int pp = p0 + wave * MF * 16 + mf * 16 + g * 4 + rg;
      bool v = pp < p.P;
      long o = 0;
      if (v) {
        const int n = fast_div(pp, p.ib_mul, p.ib_sh);
        const int rem = pp - n * p.IB;
        const int yy = fast_div(rem, p.sw_mul, p.sw_sh), xx = rem - yy * p.SW;
        v = n < p.N && yy >= 1 && xx >= 1 && xx <= p.W;
        o = ((long)(n * p.H + yy - 1) * p.W + xx - 1) * p.Co;
      }
      ok[mf][rg] = v;
      ooff[mf][rg] = o;
      cnt += v ? 1 : 0;
    }
  // valid rows of this wave's row group (identical for the 16 lanes of a column group)
  cnt += __shfl_xor(cnt, 16, 64);
  cnt += __shfl_xor(cnt, 32, 64);
  const int grp = blockIdx.x * 4 + wave;
  if (p.stats && p.counts && blockIdx.y == 0 && lane == 0) p.counts[grp] = cnt;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n0 + nf * 16 + i16;
    const bool nok = n < p.Co;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    if (p.stats) {
      float s1 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (ok[mf][rg]) s1 += acc[mf][nf][rg] + bv;
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = cnt > 0 ? s1 / (float)cnt : 0.f;
      float s2 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (ok[mf][rg]) {
            const float d = acc[mf][nf][rg] + bv - mean;
            s2 += d * d;
          }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (nok && g == 0) {
        p.stats[((long)grp * p.Co + n) * 2 + 0] = mean;
        p.stats[((long)grp * p.Co + n) * 2 + 1] = s2;
      }
    }
    if (nok) {
      const float sc = p.scale ? p.scale[n] : 1.f;
      const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (ok[mf][rg]) {
            float v = (acc[mf][nf][rg] + bv) * sc + sh;
            const long o = ooff[mf][rg] + n;
            if (p.res) v += p.res[o];
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[o] = v;
          }
    }
  }
}

// ---------------------------------------------------------------------------------------------- host ----
struct C3Plan { int MF, NF, BM, BN; size_t lds; };

// magic for unsigned division of n < 2^31 by d (1 <= d < 2^31): q = mulhi(n, mul) >> sh, mul = ceil(2^(32+sh)/d)
// with sh = ceil(log2 d); mul < 2^33, so d >= 2 keeps it in 32 bits after the usual "sh - 1" adjustment below.
static void magic_u32(unsigned d, unsigned* mul, unsigned* sh) {
  if (d == 1) { *mul = 0xFFFFFFFFu; *sh = 0; return; }   // mulhi(n, 2^32-1) == n - 1 for n >= 1, 0 for 0: avoid, d>1 always here
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                  // l = ceil(log2 d)
  // n < 2^31: m = ceil(2^(31+l) / d) fits in 32 bits and q = (n*m) >> (31+l) is exact
  const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
  *mul = (unsigned)m;
  *sh = l - 1;                                  // mulhi already shifts by 32: total shift 31 + l
}

static bool c3_plan(int N, int H, int W, int Ci, int Co, C3Plan* pl) {
  if (Ci % 16 != 0 || Co % 16 != 0 || W + 2 > MAX_SW) return false;
  const long P = (long)N * (H + 1) * (W + 2) + (W + 2);
  int nf = Co % 96 == 0 ? 6 : (Co % 64 == 0 ? 4 : (Co % 48 == 0 ? 3 : (Co % 32 == 0 ? 2 : 0)));
  if (Co == 16) nf = 1;
  if (nf == 0) return false;
  // keep >= ~2 workgroups per CU when the image planes are small
  int mf = 4;
  if ((P / 256) * (Co / (nf * 16)) < 512) mf = 2;
  if (nf == 6) mf = 2;                       // 256x96 tiles would spill (24 accumulators + staging registers)
  if (mf == 2 && nf == 6 && (P / 128) * (Co / 96) < 512 && Co % 48 == 0) nf = 3;
  pl->MF = mf; pl->NF = nf; pl->BM = 64 * mf; pl->BN = 16 * nf;
  pl->lds = (size_t)(pl->BM + 2 * (W + 2) + 2) * ROWB + (size_t)2 * pl->BN * ROWB;
  return pl->lds <= 160 * 1024;
}

extern "C" int buctd_conv3x3_bf16x3_supported(int N, int H, int W, int Ci, int Co) {
  C3Plan pl;
  return c3_plan(N, H, W, Ci, Co, &pl) ? 1 : 0;
}

extern "C" int buctd_conv3x3_bf16x3_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups,
                                                 int* rows_per_group) {
  C3Plan pl;
  BUCTD_CHECK_ARG(ngroups && rows_per_group && c3_plan(N, H, W, Ci, Co, &pl),
                  "buctd_conv3x3_bf16x3_stats_groups: unsupported shape");
  const long P = (long)N * (H + 1) * (W + 2) + (W + 2);
  *ngroups = ceil_div(P, pl.BM) * 4;
  *rows_per_group = pl.MF * 16;
  return BUCTD_OK;
}

template <int MF, int NF, bool FLIP>
static int c3_launch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
  static bool attr_set = false;
  auto fn = conv3x3_bf16x3_kernel<MF, NF, FLIP>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      buctd_set_error("conv3x3_bf16x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return BUCTD_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(ceil_div(a.P, pl.BM), ceil_div(a.Co, pl.BN));
  hipLaunchKernelGGL(fn, grid, dim3(256), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3");
  return BUCTD_OK;
}

template <bool FLIP>
static int c3_dispatch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
#define C3_CASE(mf, nf) if (pl.MF == mf && pl.NF == nf) return c3_launch<mf, nf, FLIP>(a, pl, st);
  C3_CASE(4, 1) C3_CASE(4, 2) C3_CASE(4, 3) C3_CASE(4, 4) C3_CASE(4, 6)
  C3_CASE(2, 1) C3_CASE(2, 2) C3_CASE(2, 3) C3_CASE(2, 4) C3_CASE(2, 6)
#undef C3_CASE
  buctd_set_error("conv3x3_bf16x3: no kernel for MF=%d NF=%d", pl.MF, pl.NF);
  return BUCTD_EINVAL;
}

// x: [N][H][W][Ci] -> y: [N][H][W][Co]; w: [Cw_out][3][3][Cw_in] of the FORWARD convolution.
// flip == 0: forward (Cw_out = Co, Cw_in = Ci).  flip == 1: data gradient - x is dy ([N][H][W][Ci = Cw_out]),
// y is dx ([N][H][W][Co = Cw_in]).
extern "C" int buctd_conv3x3_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const float* w, int flip,
                                    const float* bias, const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, float* stats_partials, int* stats_counts, void* stream) {
  C3Plan pl;
  BUCTD_CHECK_ARG(x && w && y, "buctd_conv3x3_bf16x3: null tensor pointer");
  BUCTD_CHECK_ARG(c3_plan(N, H, W, Ci, Co, &pl), "buctd_conv3x3_bf16x3: unsupported shape N%d H%d W%d Ci%d Co%d", N, H,
                  W, Ci, Co);
  BUCTD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "buctd_conv3x3_bf16x3: scale and shift go together");
  BUCTD_CHECK_ARG((stats_partials == nullptr) == (stats_counts == nullptr),
                  "buctd_conv3x3_bf16x3: stats partials and counts go together");
  C3Args a;
  a.x = x; a.w = w; a.out = y; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual;
  a.stats = stats_partials; a.counts = stats_counts;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.SW = W + 2; a.IB = (H + 1) * (W + 2);
  const long P = (long)N * a.IB + a.SW;
  BUCTD_CHECK_ARG(P < 2147483647L, "buctd_conv3x3_bf16x3: tensor too large");
  a.P = (int)P;
  a.relu = relu; a.flip = flip;
  magic_u32((unsigned)a.IB, &a.ib_mul, &a.ib_sh);
  magic_u32((unsigned)a.SW, &a.sw_mul, &a.sw_sh);
  return flip ? c3_dispatch<true>(a, pl, (hipStream_t)stream) : c3_dispatch<false>(a, pl, (hipStream_t)stream);
}
