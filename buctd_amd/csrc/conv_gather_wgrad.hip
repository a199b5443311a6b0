// Weight gradients of the 1x1 and stride-2 3x3 convolutions (HRNet fuse layers, transitions, layer1 Bottlenecks, stem:
// reference lib/models/pose_hrnet.py:60-108, 187-245, 338-372 - autograd of nn.Conv2d) in the bf16x6 arithmetic of
// conv3x3.hip (fp32 operands split exactly into three bf16 pieces, six v_mfma_f32_16x16x32_bf16 per product, fp32
// accumulate).  Until round 5 these ran on the exact-fp32 MFMA (conv.hip: 157 TFLOP/s peak, 20-46 TFLOP/s on these shapes).
//
//     dW[co][tap][ci] = sum over output pixels q of  dY[q][co] * X[src(q, tap)][ci]
//
// GEMM view: M = co, N = (tap, ci), K = output pixels.  1x1: one tap, src(q) = q.  Stride-2 3x3: src(q = (n, y, x), tap (r, s))
// = (n, 2y + r - 1, 2x + s - 1), zero outside the image - a GATHER per tap, so the X operand is not a shifted window of one
// staged tile as in conv3x3_wgrad.hip.
//
// Structure - every WAVEFRONT is autonomous:
//   * a workgroup owns a (co chunk, ci chunk) pair of CH = CF * 16 channels each, a tap group (1x1: the tap; stride 2: one
//     filter row = three taps) and a range of K; its four waves take the 32-pixel k-steps of that range in turn;
//   * per k-step a wave stages ITS 32 rows of dY and, tap after tap, the 32 gathered rows of X into its own 2 x 32-row LDS
//     slots (fp32 -> three bf16 pieces on the way, the row image of conv3x3_wgrad.hip), pulls the K-contiguous MFMA fragments
//     out of them with the gfx950 transpose read ds_read_b64_tr_b16 and multiplies: CF * CF * 6 MFMAs per tap.  No operand
//     is shared between waves, so the loop has NO workgroup barrier: LDS operations of one wave execute in order, and the
//     waves of a SIMD drift apart - one stages (global loads, VALU split, ds_write) while the other multiplies;
//   * the global loads of the next tile (next tap, or the next k-step's dY) are issued before the current tile is multiplied;
//   * every wave keeps its own accumulators (taps x CF x CF fragments); at the end the four sets meet in LDS, wave 0 writes
//     the workgroup's partial slab, and a reduction kernel adds the slabs in a fixed order (deterministic) into dW
//     ([Co][R][S][Ci]).
#include "c3_lean.h"

typedef __bf16 gw_bf16x4 __attribute__((ext_vector_type(4)));

struct GwArgs {
  const float* x;       // [N][H][W][Ci]
  const float* dy;      // [N][Ho][Wo][Co]
  float* part;          // slabs
  int N, H, W, Ho, Wo, Ci, Co;
  int kind;             // 1: 1x1 stride 1 (Ho = H, Wo = W); 2: 3x3 stride 2 pad 1
  int Q;                // N * Ho * Wo output pixels
  int ksteps;           // ceil(Q / 32)
  int split_q, split_rem;       // k-steps per split: q, the first split_rem splits q + 1
  int nsplit, ntg;      // K splits, tap groups (1 | 3)
  unsigned hw_mul, hw_sh, w_mul, w_sh;     // magic division by Ho * Wo and Wo
};

__device__ __forceinline__ bf16x8 gw_tr_frag(const unsigned char* p, int rs) {
  // positions {4g..4g+3} u {16+4g..16+4g+3} of this lane's channel (conv3x3_wgrad.hip: tr_frag)
  typedef __attribute__((address_space(3))) gw_bf16x4 lds_bf16x4;
  const gw_bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const gw_bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * rs));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CF> struct GwGeo {
  static constexpr int CH = CF * 16;
  static constexpr int LO = CH * 2;                                            // byte stride between the pieces of a row
  static constexpr int RS = (CH * 6) % 64 == 32 ? CH * 6 : CH * 6 + 32;        // row stride = 32 mod 64 bytes
  static constexpr int TILE = 32 * RS;                                         // one 32-row tile
  static constexpr int WAVE_LDS = 2 * TILE;                                    // dY tile | X tile
};

template <int CF, int TPG>      // channel fragments per chunk, taps per group
__global__ __launch_bounds__(256, 2) void gconv_wgrad_x6_kernel(GwArgs p) {
  using G = GwGeo<CF>;
  constexpr int CH = G::CH, LO = G::LO, RS = G::RS;
  constexpr int C4 = CH / 4;                       // float4 per row
  constexpr int PL = 32 * C4 / 64;                 // float4 per lane for one 32-row tile
  static_assert(32 * C4 % 64 == 0, "whole staging passes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int t16 = lane & 15, g = lane >> 4;
  unsigned char* Dt = smem + (size_t)wave * G::WAVE_LDS;
  unsigned char* Xt = Dt + G::TILE;
  const int co0 = blockIdx.x * CH, ci0 = blockIdx.y * CH;
  const int tg = blockIdx.z % p.ntg, z = blockIdx.z / p.ntg;
  const int ks_begin = z * p.split_q + (z < p.split_rem ? z : p.split_rem);
  const int ks_end = ks_begin + p.split_q + (z < p.split_rem ? 1 : 0);

  const int lane_off = (g * 4 + (t16 >> 2)) * RS + (t16 & 3) * 8;      // transpose-read addressing (conv3x3_wgrad.hip)

  f32x4 acc[TPG][CF][CF];
#pragma unroll
  for (int tp = 0; tp < TPG; ++tp)
#pragma unroll
    for (int mf = 0; mf < CF; ++mf)
#pragma unroll
      for (int nf = 0; nf < CF; ++nf) acc[tp][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per k-step and staged row: BYTE offset of the dY row (C3_OOB beyond the last pixel), byte offset of the source pixel of
  // tap offset (0, 0) of this tap group, and one validity bit per tap of the group (bit q * 4 + s).  Both tensors are read
  // through buffer descriptors: an out-of-range offset returns zeros - no select when the piece is split (c3_lean.h)
  const __amdgpu_buffer_rsrc_t r_x = c3_rsrc(p.x, (unsigned)p.N * p.H * p.W * p.Ci * 4u);
  const __amdgpu_buffer_rsrc_t r_d = c3_rsrc(p.dy, (unsigned)p.N * p.Ho * p.Wo * p.Co * 4u);
  unsigned dyoff[PL], xbase[PL];
  unsigned vmask = 0;
  const int ss = p.kind == 2 ? 2 : 1;
  const int dr = p.kind == 2 ? tg - 1 : 0;                       // the group's filter row
  int sc4[PL];
#pragma unroll
  for (int q = 0; q < PL; ++q) sc4[q] = ((lane + 64 * q) % C4) * 4;
  auto decode = [&](int ks) {
    vmask = 0;
#pragma unroll
    for (int q = 0; q < PL; ++q) {
      const int row = (lane + 64 * q) / C4;
      const int qq = ks * 32 + row;
      const bool ok = qq < p.Q;
      const int n = fast_div(qq, p.hw_mul, p.hw_sh);
      const int rem = qq - n * (p.Ho * p.Wo);
      const int y = fast_div(rem, p.w_mul, p.w_sh), x = rem - y * p.Wo;
      const int sy = y * ss + dr, sx = x * ss;
      dyoff[q] = ok ? (unsigned)(qq * p.Co + co0 + sc4[q]) * 4u : C3_OOB;
      xbase[q] = (unsigned)(((n * p.H + sy) * p.W + sx) * p.Ci + ci0 + sc4[q]) * 4u;
      const bool rok = ok & ((unsigned)sy < (unsigned)p.H);
#pragma unroll
      for (int sIdx = 0; sIdx < TPG; ++sIdx) {
        const int cx = sx + (p.kind == 2 ? sIdx - 1 : 0);
        vmask |= ((rok & ((unsigned)cx < (unsigned)p.W)) ? 1u : 0u) << (q * 4 + sIdx);
      }
    }
  };
  f32x4 dreg[PL], xreg[PL];
  auto load_d = [&]() {
#pragma unroll
    for (int q = 0; q < PL; ++q) dreg[q] = c3_bload(r_d, dyoff[q]);
  };
  auto load_x = [&](int sIdx) {      // tap sIdx of the group: source column offset sIdx - 1 (stride 2), 0 (1x1)
    const int delta = (p.kind == 2 ? sIdx - 1 : 0) * p.Ci * 4;
#pragma unroll
    for (int q = 0; q < PL; ++q) xreg[q] = c3_bload(r_x, ((vmask >> (q * 4 + sIdx)) & 1u) ? xbase[q] + (unsigned)delta : C3_OOB);
  };
  auto store_tile = [&](unsigned char* T, const f32x4 (&r)[PL]) {
#pragma unroll
    for (int q = 0; q < PL; ++q) {
      const int row = (lane + 64 * q) / C4;
      split_store_pk<3, LO>(T + row * RS, sc4[q], r[q]);
    }
  };

  int ks = ks_begin + wave;
  if (ks < ks_end) {
    decode(ks);
    load_d();
    load_x(0);
  }
  for (; ks < ks_end; ks += 4) {
    // dY tile of this k-step -> LDS -> A fragments (kept for all taps of the step)
    store_tile(Dt, dreg);
    bf16x8 a[3][CF];
#pragma unroll
    for (int mf = 0; mf < CF; ++mf)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) a[pc][mf] = gw_tr_frag(Dt + lane_off + mf * 32 + pc * LO, RS);
#pragma unroll
    for (int tp = 0; tp < TPG; ++tp) {
      store_tile(Xt, xreg);
      // the next tile's global loads travel under this tap's multiplications
      if (tp + 1 < TPG) load_x(tp + 1);
      else if (ks + 4 < ks_end) {
        decode(ks + 4);
        load_d();
        load_x(0);
      }
#pragma unroll
      for (int nf = 0; nf < CF; ++nf) {
        bf16x8 b[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[pc] = gw_tr_frag(Xt + lane_off + nf * 32 + pc * LO, RS);
#define GW_MMA(qa, qb)                                                                                      \
  _Pragma("unroll") for (int mf = 0; mf < CF; ++mf) acc[tp][mf][nf] =                                       \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa][mf], b[qb], acc[tp][mf][nf], 0, 0, 0);
        GW_MMA(2, 0) GW_MMA(0, 2) GW_MMA(1, 1) GW_MMA(1, 0) GW_MMA(0, 1) GW_MMA(0, 0)
#undef GW_MMA
      }
    }
  }

  // the four waves' accumulators meet in LDS (a wave's own staging region takes one tap group's fragments), wave 0 adds them
  // in wave order and writes the workgroup's slab in accumulator order: element ((tp * CF + mf) * CF + nf) * 64 + lane = the
  // register quad (co = co0 + mf * 16 + g * 4 + 0..3, ci = ci0 + nf * 16 + t16) of tap tg * TPG + tp
  {
    static_assert(CF * CF * 1024 <= G::WAVE_LDS, "a tap's fragments must fit the wave's staging region");
    const size_t slab4 = (size_t)TPG * CF * CF * 64;
    const size_t pair = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    f32x4* outp = reinterpret_cast<f32x4*>(p.part) + ((pair * p.ntg + tg) * p.nsplit + z) * slab4;
    f32x4* mine = reinterpret_cast<f32x4*>(Dt);
#pragma unroll
    for (int tp = 0; tp < TPG; ++tp) {
      __syncthreads();
      if (wave) {
#pragma unroll
        for (int mf = 0; mf < CF; ++mf)
#pragma unroll
          for (int nf = 0; nf < CF; ++nf) mine[(mf * CF + nf) * 64 + lane] = acc[tp][mf][nf];
      }
      __syncthreads();
      if (!wave) {
#pragma unroll
        for (int mf = 0; mf < CF; ++mf)
#pragma unroll
          for (int nf = 0; nf < CF; ++nf) {
            f32x4 v = acc[tp][mf][nf];
#pragma unroll
            for (int w = 1; w < 4; ++w)
              v += reinterpret_cast<const f32x4*>(smem + (size_t)w * G::WAVE_LDS)[(mf * CF + nf) * 64 + lane];
            outp[((tp * CF + mf) * CF + nf) * 64 + lane] = v;
          }
      }
    }
  }
}

// slab reduction: one thread per accumulator quad of a (pair, tap group), the nsplit slabs added in a fixed order
template <int CF, int TPG>
__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int Ci, int Co,
                                                                 int R, int ntg, int nslab, int accumulate) {
  constexpr int SLAB = TPG * CF * CF * 64;                 // float4 per slab
  __shared__ f32x4 sm[16][16];
  const int col = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int ptg = blockIdx.y;                              // pair * ntg + tg
  const int pair = ptg / ntg, tg = ptg - pair * ntg;
  const int nco = Co / (CF * 16);
  const int co0 = (pair % nco) * CF * 16, ci0 = (pair / nco) * CF * 16;
  const f32x4* base = reinterpret_cast<const f32x4*>(part) + (size_t)ptg * nslab * SLAB;
  for (int e0 = blockIdx.x * 16; e0 < SLAB; e0 += gridDim.x * 16) {
    const int e = e0 + col;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (e < SLAB) {
      const f32x4* src = base + e;
      int zz = zl;
      for (; zz + 16 < nslab; zz += 32) {
        s0 += src[(size_t)zz * SLAB];
        s1 += src[(size_t)(zz + 16) * SLAB];
      }
      for (; zz < nslab; zz += 16) s0 += src[(size_t)zz * SLAB];
    }
    sm[zl][col] = s0 + s1;
    __syncthreads();
    if (zl == 0 && e < SLAB) {
      f32x4 sv = sm[0][col];
#pragma unroll
      for (int k = 1; k < 16; ++k) sv += sm[k][col];
      const int lane = e & 63, r = e >> 6;
      const int nf = r % CF, mf = (r / CF) % CF, tp = r / (CF * CF);
      const int tap = tg * TPG + tp;
      const int ci = ci0 + nf * 16 + (lane & 15);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int co = co0 + mf * 16 + (lane >> 4) * 4 + rg;
        float* dst = dw + ((size_t)co * R * R + tap) * Ci + ci;
        *dst = accumulate ? *dst + sv[rg] : sv[rg];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- host ----
struct GwPlan { int CF, TPG, ntg, nsplit, q, rem, ksteps, Ho, Wo; size_t lds, ws; };

static bool gw_plan(int kind, int N, int H, int W, int Ci, int Co, GwPlan* pl) {
  if ((kind != 1 && kind != 2) || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 || Co % 16) return false;
  if (kind == 2 && ((H & 1) || (W & 1))) return false;
  int cf;
  // 64-channel chunks (one workgroup per CU: 106 KB of LDS) where the map is large enough to fill the chip that way: the
  // 64 <-> 256 layers of layer1 read dY once per 64 input channels instead of once per 32 (129 -> 104 us, 130 -> 98)
  if (kind == 1 && Ci % 64 == 0 && Co % 64 == 0 && (long)N * H * W >= 65536) cf = 4;
  else if (Ci % 48 == 0 && Co % 48 == 0) cf = 3;
  else if (Ci % 32 == 0 && Co % 32 == 0) cf = 2;
  else return false;
  pl->CF = cf;
  pl->TPG = kind == 2 ? 3 : 1;
  pl->ntg = kind == 2 ? 3 : 1;
  pl->Ho = kind == 2 ? H / 2 : H;
  pl->Wo = kind == 2 ? W / 2 : W;
  const long Q = (long)N * pl->Ho * pl->Wo;
  // (both tensors are addressed with 32-bit byte offsets below the out-of-range mark 2^31: c3_lean.h)
  if (pl->Wo < 2 || Q >= 2147483647L / 64 || (long)N * H * W * Ci * 4 >= 2147483647L || Q * Co * 4 >= 2147483647L) return false;
  pl->ksteps = (int)((Q + 31) / 32);
  const int ch = cf * 16;
  const long pairs = (long)(Co / ch) * (Ci / ch);
  // about 768 workgroups on the 512 slots; every workgroup at least sixteen k-steps (four per wave) to pay for its slab
  long want = (768 + pairs * pl->ntg - 1) / (pairs * pl->ntg);
  const long maxs = (pl->ksteps + 15) / 16;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  pl->nsplit = (int)want;
  pl->q = pl->ksteps / pl->nsplit;
  pl->rem = pl->ksteps % pl->nsplit;
  pl->lds = (size_t)4 * (cf == 4 ? GwGeo<4>::WAVE_LDS : cf == 3 ? GwGeo<3>::WAVE_LDS : GwGeo<2>::WAVE_LDS);
  pl->ws = (size_t)pairs * pl->ntg * pl->nsplit * ((size_t)pl->TPG * cf * cf * 64 * 16);
  return pl->lds <= 160 * 1024;
}

extern "C" int buctd_gconv_wgrad_x6_supported(int kind, int N, int H, int W, int Ci, int Co) {
  GwPlan pl;
  return gw_plan(kind, N, H, W, Ci, Co, &pl) ? 1 : 0;
}
extern "C" size_t buctd_gconv_wgrad_x6_workspace(int kind, int N, int H, int W, int Ci, int Co) {
  GwPlan pl;
  return gw_plan(kind, N, H, W, Ci, Co, &pl) ? pl.ws : 0;
}

template <int CF, int TPG>
static int gw_launch(const GwArgs& a, const GwPlan& pl, float* dw, int accumulate, hipStream_t st) {
  static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
  auto fn = gconv_wgrad_x6_kernel<CF, TPG>;
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done, "buctd_gconv_wgrad_x6")) return rc;
  const int ch = CF * 16;
  hipLaunchKernelGGL(fn, dim3(a.Co / ch, a.Ci / ch, pl.nsplit * pl.ntg), dim3(256), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_gconv_wgrad_x6");
  const int R = a.kind == 2 ? 3 : 1;
  const int pairs = (a.Co / ch) * (a.Ci / ch);
  const dim3 rgrid(ceil_div((long)TPG * CF * CF * 64, 16), pairs * pl.ntg);
  hipLaunchKernelGGL((gconv_wgrad_reduce_kernel<CF, TPG>), rgrid, dim3(256), 0, st, (const float*)a.part, dw, a.Ci, a.Co, R, pl.ntg,
                     pl.nsplit, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_gconv_wgrad_x6 (reduce)");
  return BUCTD_OK;
}

/* dw[Co][R][S][Ci] (+)= weight gradient of a 1x1 (kind 1) or stride-2 3x3 pad-1 (kind 2) convolution Ci -> Co on x
 * [N][H][W][Ci] with output gradient dy [N][Ho][Wo][Co] (Ho = H, or H / 2 for kind 2; H, W even). */
extern "C" int buctd_gconv_wgrad_x6(int kind, int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                                    int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  GwPlan pl;
  BUCTD_CHECK_ARG(x && dy && dw, "buctd_gconv_wgrad_x6: null tensor pointer");
  BUCTD_CHECK_ARG(gw_plan(kind, N, H, W, Ci, Co, &pl), "buctd_gconv_wgrad_x6: unsupported shape kind %d N%d H%d W%d Ci%d Co%d", kind, N,
                  H, W, Ci, Co);
  if (!workspace || workspace_bytes < pl.ws) {
    buctd_set_error("buctd_gconv_wgrad_x6: workspace %zu bytes < required %zu", workspace_bytes, pl.ws);
    return BUCTD_EWORKSPACE;
  }
  GwArgs a;
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.N = N; a.H = H; a.W = W; a.Ho = pl.Ho; a.Wo = pl.Wo; a.Ci = Ci; a.Co = Co; a.kind = kind;
  a.Q = N * pl.Ho * pl.Wo;
  a.ksteps = pl.ksteps; a.split_q = pl.q; a.split_rem = pl.rem; a.nsplit = pl.nsplit; a.ntg = pl.ntg;
  magic_u32((unsigned)(pl.Ho * pl.Wo), &a.hw_mul, &a.hw_sh);
  magic_u32((unsigned)pl.Wo, &a.w_mul, &a.w_sh);
  hipStream_t st = (hipStream_t)stream;
  if (kind == 1) {
    if (pl.CF == 4) return gw_launch<4, 1>(a, pl, dw, accumulate, st);
    if (pl.CF == 3) return gw_launch<3, 1>(a, pl, dw, accumulate, st);
    return gw_launch<2, 1>(a, pl, dw, accumulate, st);
  }
  if (pl.CF == 3) return gw_launch<3, 3>(a, pl, dw, accumulate, st);
  return gw_launch<2, 3>(a, pl, dw, accumulate, st);
}
