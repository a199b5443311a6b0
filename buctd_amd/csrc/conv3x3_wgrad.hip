// Weight gradient of the 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores ("bf16x3" split-fp32
// operands, fp32 accumulate) - the companion of conv3x3.hip for the BasicBlock convs of reference
// lib/models/pose_hrnet.py:28-57 (autograd of nn.Conv2d):
//
//     dW[co][tap][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci]          p over the zero-padded flattened positions
//
// GEMM view: M = co, N = (tap, ci), K = positions.  Both operands are staged position-major in LDS exactly like the
// forward kernel's input tile (rows = positions, bf16 hi | lo per row), so a filter tap is again a row shift; the
// K-contiguous MFMA fragments (8 consecutive positions of one channel per lane) come out of the position-major
// tiles through the gfx950 transpose read ds_read_b64_tr_b16 (4 rows x 16 channels -> lane c gets 4 positions of
// channel c; verified on hardware, scratch/tr_probe).  One workgroup owns a (co-chunk, ci-chunk) pair - CH = 48 or
// 32 channels each - for a range of positions and all 9 taps: 9*CH/16 n-fragments dealt round-robin to the 4 waves,
// CH/16 m-fragments each.  Position ranges are split over the grid; per-split slabs are summed by wg3_reduce.
// The X tile is a 256-row ring indexed by (position - first staged position) & 255: consecutive stages of a workgroup
// overlap in all but WG_KB rows, so after the first stage only the WG_KB new rows are fetched and split (the halo of
// 2*SW+2 rows used to be re-staged every stage: 2.6x the traffic and VALU work at W = 72).
#include "common.h"
#include "../../include/buctd_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define WG_KB 96          // positions per LDS stage (3 MFMA k-steps of 32)
#define WG_MAX_SW 75
#define WG_RING 256        // X ring rows (power of two >= WG_KB + 2*WG_MAX_SW + 2)

struct WG3Args {
  const float* x;
  const float* dy;
  float* part;          // [nsplit][Co][9][Ci]
  int N, H, W, Ci, Co;
  int SW, IB, P;
  int pos_per_split;    // multiple of WG_KB
  unsigned ib_mul, ib_sh, sw_mul, sw_sh;
};

__device__ __forceinline__ int wg_fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(__umulhi((unsigned)n, mul) >> sh);
}

// channels c..c+3 of one row -> hi at byte 2c, lo at LO + 2c
template <int LO>
__device__ __forceinline__ void wg_split_store(unsigned char* row, int c, f32x4 v) {
  u16x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    const __bf16 l = (__bf16)(v[j] - (float)h);
    hi[j] = __builtin_bit_cast(unsigned short, h);
    lo[j] = __builtin_bit_cast(unsigned short, l);
  }
  *reinterpret_cast<u16x4*>(row + 2 * c) = hi;
  *reinterpret_cast<u16x4*>(row + LO + 2 * c) = lo;
}

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int row_bytes) {
  // two transpose reads: positions +0..3 and +4..7 of this lane's channel
  // (the v4i16 form + per-element bit_cast to __bf16 is miscompiled by ROCm 7.2's hipcc - every element became
  //  element 0 - so use the v4bf16 form and a shufflevector; checked by scratch/tr_probe2)
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * row_bytes));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* p, const unsigned char* q) {
  // as tr_frag with the two 4-row groups addressed separately (the ring may wrap between them)
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CF>   // channel fragments (of 16) per chunk: 3 -> 48 channels, 2 -> 32
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_bf16x3_kernel(WG3Args p) {
  constexpr int CH = CF * 16;
  constexpr int LO = CH * 2;                 // byte offset of the lo half inside a row
  constexpr int RS = CH * 4 + 32;            // row stride: 224 (CH 48) / 160 (CH 32), = 32 mod 64
  constexpr int C4 = CH / 4;                 // float4 per row
  constexpr int XR = WG_KB + 2 * WG_MAX_SW + 2;
  constexpr int PD = (WG_KB * C4 + 255) / 256;
  constexpr int PX = (WG_KB * C4 + 255) / 256;   // X rows travel WG_KB at a time (the first stage takes several rounds)
  constexpr int NFR = 9 * CF;                // n-fragments (tap, ci16)
  constexpr int NW = (NFR + 3) / 4;          // n-fragments per wave

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int R = WG_KB + 2 * p.SW + 2;
  unsigned char* Dt = smem;                          // dY tile [WG_KB][RS]
  unsigned char* Xt = smem + (size_t)WG_KB * RS;     // X  ring [WG_RING][RS]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int t16 = lane & 15, g = lane >> 4;
  const int co0 = blockIdx.x * CH, ci0 = blockIdx.y * CH;
  const int k_begin = blockIdx.z * p.pos_per_split;
  int k_end = k_begin + p.pos_per_split;
  if (k_end > p.P) k_end = p.P;
  const int halo = p.SW + 1;

  auto pos_offset = [&](int pp, int C) -> int {   // element offset of pixel pp in an [N][H][W][C] tensor, -1 = pad
    if (pp < 0 || pp >= p.P) return -1;
    const int n = wg_fast_div(pp, p.ib_mul, p.ib_sh);
    const int rem = pp - n * p.IB;
    const int yy = wg_fast_div(rem, p.sw_mul, p.sw_sh);
    const int xx = rem - yy * p.SW;
    if (n >= p.N || yy < 1 || xx < 1 || xx > p.W) return -1;
    return ((n * p.H + yy - 1) * p.W + xx - 1) * C;
  };

  f32x4 dreg[PD], xreg[PX];
  unsigned dmask = 0, xmask = 0;             // bit q: the row loaded in pass q is a real pixel (else zero row)
  auto load_d = [&](int k0) {
    dmask = 0;
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      const int idx = t + 256 * q;
      const int row = idx / C4, c4 = (idx - row * C4) * 4;
      const int pp = k0 + row;
      const int o = (row < WG_KB && pp < k_end) ? pos_offset(pp, p.Co) : -1;
      dmask |= (o >= 0 ? 1u : 0u) << q;
      dreg[q] = *reinterpret_cast<const f32x4*>(p.dy + (o >= 0 ? o + co0 + c4 : 0));
    }
  };
  // X rows rel0 .. rel0 + nrows - 1, counted from position k_begin - halo (ring slot = rel & 255)
  auto load_x = [&](int rel0, int nrows) {
    xmask = 0;
#pragma unroll
    for (int q = 0; q < PX; ++q)
      if (256 * q < nrows * C4) {
        const int idx = t + 256 * q;
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        const int o = row < nrows ? pos_offset(k_begin - halo + rel0 + row, p.Ci) : -1;
        xmask |= (o >= 0 ? 1u : 0u) << q;
        xreg[q] = *reinterpret_cast<const f32x4*>(p.x + (o >= 0 ? o + ci0 + c4 : 0));
      }
  };
  auto store_d = [&]() {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      const int idx = t + 256 * q;
      const int row = idx / C4, c4 = (idx - row * C4) * 4;
      if (row < WG_KB)
        wg_split_store<LO>(Dt + (size_t)row * RS, c4, ((dmask >> q) & 1u) ? dreg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
    }
  };
  auto store_x = [&](int rel0, int nrows) {
#pragma unroll
    for (int q = 0; q < PX; ++q)
      if (256 * q < nrows * C4) {
        const int idx = t + 256 * q;
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        if (row < nrows)
          wg_split_store<LO>(Xt + (size_t)((rel0 + row) & (WG_RING - 1)) * RS, c4,
                             ((xmask >> q) & 1u) ? xreg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
      }
  };

  f32x4 acc[CF][NW];
#pragma unroll
  for (int mf = 0; mf < CF; ++mf)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[mf][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read lane addressing: lane t16 of group g points at row g*8 + (t16>>2), channels 4*(t16&3)..+3
  const int lane_row = g * 8 + (t16 >> 2), lane_col = (t16 & 3) * 8;
  const int lane_off = lane_row * RS + lane_col;

  // first stage: the halo rows go in synchronously, WG_KB at a time through the same registers as the steady state
  int rel0 = 0, nrows = R < WG_KB ? R : WG_KB;
  if (k_begin < k_end) {
    dmask = 0;
    while (rel0 + WG_KB < R) {
      load_x(rel0, WG_KB);
      store_x(rel0, WG_KB);
      rel0 += WG_KB;
    }
    nrows = R - rel0;
    load_d(k_begin);
    load_x(rel0, nrows);
  }
  int si = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += WG_KB, ++si) {
    __syncthreads();                       // previous stage fully consumed
    store_d();
    store_x(rel0, nrows);
    if (k0 + WG_KB < k_end) {              // in flight during the MFMAs below
      rel0 = R + si * WG_KB;
      nrows = WG_KB;
      load_d(k0 + WG_KB);
      load_x(rel0, nrows);
    }
    __syncthreads();
    const int xbase = si * WG_KB + lane_row;   // ring row (before wrapping) of this lane's first position, tap shift 0
#pragma unroll
    for (int ks = 0; ks < WG_KB / 32; ++ks) {
      bf16x8 ah[CF], al[CF];
#pragma unroll
      for (int mf = 0; mf < CF; ++mf) {
        const unsigned char* q = Dt + (size_t)ks * 32 * RS + lane_off + mf * 32;
        ah[mf] = tr_frag(q, RS);
        al[mf] = tr_frag(q + LO, RS);
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int nf = wave + 4 * j;         // (tap, ci16) fragment of this wave
        if (nf < NFR) {
          const int tap = nf / CF, cf = nf - tap * CF;
          const int shift = (tap / 3) * p.SW + tap % 3;      // X row of position k + shift(tap) (the ring starts at -halo)
          const int r0 = (xbase + ks * 32 + shift) & (WG_RING - 1), r1 = (r0 + 4) & (WG_RING - 1);
          const unsigned char* q0 = Xt + (size_t)r0 * RS + lane_col + cf * 32;
          const unsigned char* q1 = Xt + (size_t)r1 * RS + lane_col + cf * 32;
          const bf16x8 bh = tr_frag2(q0, q1);
          const bf16x8 bl = tr_frag2(q0 + LO, q1 + LO);
#pragma unroll
          for (int mf = 0; mf < CF; ++mf) acc[mf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mf], bh, acc[mf][j], 0, 0, 0);
#pragma unroll
          for (int mf = 0; mf < CF; ++mf) acc[mf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mf], bl, acc[mf][j], 0, 0, 0);
#pragma unroll
          for (int mf = 0; mf < CF; ++mf) acc[mf][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mf], bh, acc[mf][j], 0, 0, 0);
        }
      }
    }
  }

  // partial slab: [split][co][tap][ci]; accumulator (mf, j, reg): co = co0 + mf*16 + g*4 + reg, ci = ci0 + cf*16 + t16
  float* outp = p.part + (size_t)blockIdx.z * p.Co * 9 * p.Ci;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int nf = wave + 4 * j;
    if (nf < NFR) {
      const int tap = nf / CF, cf = nf - tap * CF;
      const int ci = ci0 + cf * 16 + t16;
#pragma unroll
      for (int mf = 0; mf < CF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int co = co0 + mf * 16 + g * 4 + rg;
          outp[((size_t)co * 9 + tap) * p.Ci + ci] = acc[mf][j][rg];
        }
    }
  }
}

// slab reduction: 32 float4 columns x 8 split-lanes per workgroup, so even the 48x432 gradient (5184 float4)
// spreads over 162 workgroups and every split slab is read by 8 independent lanes
__global__ __launch_bounds__(256) void wg3_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int nsplit, int accumulate) {
  __shared__ f32x4 sm[8][32];
  const long n4 = n >> 2;
  const int col = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (long base = (long)blockIdx.x * 32; base < n4; base += (long)gridDim.x * 32) {
    const long i = base + col;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n4)
      for (int z = zl; z < nsplit; z += 8) s += reinterpret_cast<const f32x4*>(part + (long)z * n)[i];
    sm[zl][col] = s;
    __syncthreads();
    if (zl == 0 && i < n4) {
#pragma unroll
      for (int k = 1; k < 8; ++k) s += sm[k][col];
      if (accumulate) s += reinterpret_cast<const f32x4*>(out)[i];
      reinterpret_cast<f32x4*>(out)[i] = s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- host ----
struct WG3Plan { int CF, nsplit, pps; size_t lds; };

static void wg_magic(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  *mul = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  *sh = l - 1;
}

static bool wg3_plan(int N, int H, int W, int Ci, int Co, WG3Plan* pl) {
  if (W + 2 > WG_MAX_SW || H < 1 || W < 2) return false;
  int cf;
  if (Ci % 48 == 0 && Co % 48 == 0) cf = 3;
  else if (Ci % 32 == 0 && Co % 32 == 0) cf = 2;
  else return false;
  const int ch = cf * 16;
  const long P = (long)N * (H + 1) * (W + 2) + (W + 2);
  const long pairs = (long)(Co / ch) * (Ci / ch);
  long want = (384 + pairs - 1) / pairs;          // ~1.5 workgroups per CU (measured flat optimum 320..512 inside the
                                                  // train step): partial-slab traffic grows with the split
  const long stages = (P + WG_KB - 1) / WG_KB;
  if (want > stages) want = stages;
  if (want < 1) want = 1;
  const long per = (stages + want - 1) / want;    // stages per split
  pl->CF = cf;
  pl->pps = (int)(per * WG_KB);
  pl->nsplit = (int)((P + pl->pps - 1) / pl->pps);
  const int rs = ch * 4 + 32;
  pl->lds = (size_t)WG_KB * rs + (size_t)WG_RING * rs;
  return pl->lds <= 160 * 1024;
}

extern "C" int buctd_conv3x3_wgrad_bf16x3_supported(int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  return wg3_plan(N, H, W, Ci, Co, &pl) ? 1 : 0;
}

extern "C" size_t buctd_conv3x3_wgrad_bf16x3_workspace(int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  if (!wg3_plan(N, H, W, Ci, Co, &pl)) return 0;
  return (size_t)pl.nsplit * Co * 9 * Ci * sizeof(float);
}

template <int CF>
static int wg3_launch(const WG3Args& a, const WG3Plan& pl, hipStream_t st) {
  static bool attr_set = false;
  auto fn = conv3x3_wgrad_bf16x3_kernel<CF>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      buctd_set_error("conv3x3_wgrad_bf16x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return BUCTD_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(a.Co / (CF * 16), a.Ci / (CF * 16), pl.nsplit);
  hipLaunchKernelGGL(fn, grid, dim3(256), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x3");
  return BUCTD_OK;
}

extern "C" int buctd_conv3x3_wgrad_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const float* dy,
                                          float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  WG3Plan pl;
  BUCTD_CHECK_ARG(x && dy && dw, "buctd_conv3x3_wgrad_bf16x3: null tensor pointer");
  BUCTD_CHECK_ARG(wg3_plan(N, H, W, Ci, Co, &pl), "buctd_conv3x3_wgrad_bf16x3: unsupported shape N%d H%d W%d Ci%d Co%d",
                  N, H, W, Ci, Co);
  const size_t need = (size_t)pl.nsplit * Co * 9 * Ci * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_conv3x3_wgrad_bf16x3: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  WG3Args a;
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.SW = W + 2; a.IB = (H + 1) * (W + 2);
  const long P = (long)N * a.IB + a.SW;
  BUCTD_CHECK_ARG(P < 2147483647L, "buctd_conv3x3_wgrad_bf16x3: tensor too large");
  a.P = (int)P;
  a.pos_per_split = pl.pps;
  wg_magic((unsigned)a.IB, &a.ib_mul, &a.ib_sh);
  wg_magic((unsigned)a.SW, &a.sw_mul, &a.sw_sh);
  hipStream_t st = (hipStream_t)stream;
  int rc = pl.CF == 3 ? wg3_launch<3>(a, pl, st) : wg3_launch<2>(a, pl, st);
  if (rc) return rc;
  const long n = (long)Co * 9 * Ci;
  int blocks = ceil_div(n / 4, 32);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wg3_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, pl.nsplit,
                     accumulate);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x3(reduce)");
  return BUCTD_OK;
}
