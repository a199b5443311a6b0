// NHWC implicit-GEMM convolution for gfx950: forward, data-gradient and
// weight-gradient, all on the fp32 MFMA tile engine of gemm_core.h.
//
// Replaces the cuDNN calls behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear in
//   reference lib/models/pose_hrnet.py:28-98 (BasicBlock / Bottleneck convs),
//   lib/models/pose_hrnet_coam.py:635-641,666-670 (CoAM convs),
//   lib/models/self_attention.py:31-37 (fc_q/k/v/o as 1x1 convs over tokens).
//
// Layouts: activations [N][H][W][C] fp32, weights [Co][R][S][Ci] fp32 (the
// physical layout of a torch channels_last OIHW tensor).
//
//   forward : out[m=(n,ho,wo)][co] = sum_{r,s,ci} x[n][ho*st-p+r][wo*st-p+s][ci] * w[co][r][s][ci]
//   dgrad   : dx[m=(n,hi,wi)][ci] = sum_{r,s,co} dy[n][(hi+p-r)/st][(wi+p-s)/st][co] * w[co][r][s][ci]
//   wgrad   : dw[co][(r,s,ci)]    = sum_{pix}    dy[pix][co] * x[src(pix,r,s)][ci]      (split over pixels)
//
// A operand = gathered rows of the source tensor (row image, 16 channels of one
// filter tap per stage); B operand = weights (row image for forward, col image
// for dgrad).  HBM->register loads for stage t+1 are issued before the MFMAs of
// stage t and written to the other LDS buffer afterwards (one barrier / stage).
#include "gemm_core.h"
#include <stdlib.h>
#include "../../include/buctd_hip.h"

struct ConvArgs {
  const float* src;
  const float* w;
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  float* stats;
  int N, SH, SW, SC;  // source tensor dims (fwd: x, dgrad: dy)
  int RH, RW;         // row space (fwd: Ho,Wo ; dgrad: H,W)
  int OC;             // output channels (fwd: Co ; dgrad: Ci)
  int R, S, stride, pad;
  int M, K;           // M = N*RH*RW, K = R*S*SC
  int wRSCi, wCi;     // weight strides: R*S*Ci and Ci
  int relu;
  int par;            // dgrad of a stride-2 conv split by output-pixel parity (blockIdx.z = 2*py + px): every class
                      // only visits the taps that reach it (1, 2, 2 or 4 of the 9) instead of testing all nine
};

template <class T, bool DGRAD, bool VEC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs p) {
  constexpr int BM = T::BM, BN = T::BN;
  constexpr bool BCOL = DGRAD;
  using AImg = OperandImage<BM, false>;
  using BImg = OperandImage<BN, BCOL>;
  constexpr int PA = BM / 64;
  constexpr int PB = (BN + 63) / 64;            // row-image passes for B
  constexpr int NB4 = BN / 4;
  constexpr int PBK = (GK * NB4 + 255) / 256;   // col-image passes for B
  static_assert(BM % 64 == 0, "BM multiple of 64");

  __shared__ __attribute__((aligned(16))) float lds[2 * (AImg::SIZE + BImg::SIZE)];
  constexpr int STAGE = AImg::SIZE + BImg::SIZE;


  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int arow = t >> 2, chunk = t & 3;

  // parity class (stride-2 data gradient only): rows of this launch slice are the pixels (2a+py, 2b+px)
  const bool par = DGRAD && p.par;
  const int py = par ? (int)(blockIdx.z >> 1) : 0, px = par ? (int)(blockIdx.z & 1) : 0;
  const int RHc = par ? (p.RH - py + 1) / 2 : p.RH, RWc = par ? (p.RW - px + 1) / 2 : p.RW;
  const int Mc = par ? p.N * RHc * RWc : p.M;
  if (m0 >= Mc) return;
  const int tr0 = par ? ((py + p.pad) & 1) : 0, ts0 = par ? ((px + p.pad) & 1) : 0;   // first reaching tap
  const int tns = par ? (p.S - ts0 + 1) / 2 : p.S;
  const int Kc = par ? ((p.R - tr0 + 1) / 2) * tns * p.SC : p.K;

  // per-pass row decode for the A gather
  int r_img[PA], r_a[PA], r_b[PA];
  bool r_ok[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int m = m0 + arow + 64 * q;
    r_ok[q] = m < Mc;
    const int mm = r_ok[q] ? m : 0;
    const int img = mm / (RHc * RWc);
    const int rem = mm - img * (RHc * RWc);
    r_a[q] = rem / RWc;
    r_b[q] = rem - r_a[q] * RWc;
    if (par) { r_a[q] = 2 * r_a[q] + py; r_b[q] = 2 * r_b[q] + px; }
    r_img[q] = img * p.SH * p.SW * p.SC;
  }

  auto src_offset = [&](int q, int r, int s, bool& ok) -> int {
    int sh, sw;
    if (!DGRAD) {
      sh = r_a[q] * p.stride - p.pad + r;
      sw = r_b[q] * p.stride - p.pad + s;
      ok = r_ok[q] && (unsigned)sh < (unsigned)p.SH && (unsigned)sw < (unsigned)p.SW;
    } else {
      const int th = r_a[q] + p.pad - r, tw = r_b[q] + p.pad - s;
      if (p.stride == 1) { sh = th; sw = tw; }
      else if (p.stride == 2) { sh = th >> 1; sw = tw >> 1; }
      else { sh = th / p.stride; sw = tw / p.stride; }
      ok = r_ok[q] && th >= 0 && tw >= 0 && sh * p.stride == th && sw * p.stride == tw && sh < p.SH && sw < p.SW;
    }
    return r_img[q] + (sh * p.SW + sw) * p.SC;
  };

  f32x4 areg[PA];
  f32x4 breg[BCOL ? PBK : PB];

  auto load_stage = [&](int k0) {
    // ---- A: gathered source rows --------------------------------------
    if (VEC) {
      const int tap = k0 / p.SC, c0 = k0 - tap * p.SC;
      int r = tap / tns, s = tap - r * tns;
      if (par) { r = tr0 + 2 * r; s = ts0 + 2 * s; }
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        bool ok;
        const int off = src_offset(q, r, s, ok);
        areg[q] = ok ? *reinterpret_cast<const f32x4*>(p.src + off + c0 + chunk * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + chunk * 4 + j;
          v[j] = 0.f;
          if (k < p.K) {
            const int tap = k / p.SC, c = k - tap * p.SC;
            const int r = tap / p.S, s = tap - r * p.S;
            bool ok;
            const int off = src_offset(q, r, s, ok);
            if (ok) v[j] = p.src[off + c];
          }
        }
        areg[q] = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
    // ---- B: weights ------------------------------------------------------
    if (!BCOL) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int nl = arow + 64 * q;
        const int n = n0 + nl;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (nl < BN && n < p.OC) {
          const float* wp = p.w + (long)n * p.K + k0 + chunk * 4;
          if (VEC) {
            v = *reinterpret_cast<const f32x4*>(wp);
          } else {
            const int kb = k0 + chunk * 4;
            if (kb + 0 < p.K) v.x = wp[0];
            if (kb + 1 < p.K) v.y = wp[1];
            if (kb + 2 < p.K) v.z = wp[2];
            if (kb + 3 < p.K) v.w = wp[3];
          }
        }
        breg[q] = v;
      }
    } else {
      // B(k=(tap,co), n=ci) = w[co][tap][ci]
#pragma unroll
      for (int q = 0; q < PBK; ++q) {
        const int idx = t + 256 * q;
        const int krow = idx / NB4, n4 = idx - krow * NB4;
        const int k = k0 + krow, n = n0 + n4 * 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (krow < GK && k < Kc && n < p.OC) {
          int tap = k / p.SC;
          const int co = k - tap * p.SC;
          if (par) { const int tr = tap / tns; tap = (tr0 + 2 * tr) * p.S + ts0 + 2 * (tap - tr * tns); }
          const float* wp = p.w + (long)co * p.wRSCi + tap * p.wCi + n;
          if (VEC) {
            v = *reinterpret_cast<const f32x4*>(wp);
          } else {
            v.x = wp[0];
            if (n + 1 < p.OC) v.y = wp[1];
            if (n + 2 < p.OC) v.z = wp[2];
            if (n + 3 < p.OC) v.w = wp[3];
          }
        }
        breg[q] = v;
      }
    }
  };

  auto store_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      *reinterpret_cast<f32x4*>((lds + buf * STAGE) + (arow + 64 * q) * AImg::LD + chunk * 4) = areg[q];
    if (!BCOL) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int nl = arow + 64 * q;
        if (nl < BN) *reinterpret_cast<f32x4*>((lds + buf * STAGE + AImg::SIZE) + nl * BImg::LD + chunk * 4) = breg[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < PBK; ++q) {
        const int idx = t + 256 * q;
        const int krow = idx / NB4, n4 = idx - krow * NB4;
        if (krow < GK) *reinterpret_cast<f32x4*>((lds + buf * STAGE + AImg::SIZE) + krow * BImg::LD + n4 * 4) = breg[q];
      }
    }
  };

  f32x4 acc[T::MF][T::NF];
  zero_acc<T>(acc);

  const int nk = (Kc + GK - 1) / GK;
  if (nk > 0) {
    load_stage(0);
    store_stage(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_stage((kt + 1) * GK);
    mma_stage<T, false, BCOL>(lds + cur * STAGE, lds + cur * STAGE + AImg::SIZE, acc, wm, wn, lane);
    if (kt + 1 < nk) store_stage(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------
  const bool do_stats = p.stats != nullptr;
#pragma unroll
  for (int nf = 0; nf < T::NF; ++nf) {
    const int n = n0 + acc_col<T>(wn, nf, lane);
    const bool nok = n < p.OC;
    const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
    if (do_stats) {
      // Welford partial of this wave's MF*16 rows for column n:
      // (mean, M2) over the valid rows; bn_finalize combines the groups.
      const int g = blockIdx.x * T::WM + wm;
      const int row0 = m0 + wm * T::MF * 16;
      int cnt = p.M - row0;
      cnt = cnt < 0 ? 0 : (cnt > T::MF * 16 ? T::MF * 16 : cnt);
      float s1 = 0.f;
#pragma unroll
      for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int m = m0 + acc_row<T>(wm, mf, lane, rg);
          if (m < p.M) s1 += acc[mf][nf][rg] + bv;
        }
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = cnt > 0 ? s1 / (float)cnt : 0.f;
      float s2 = 0.f;
#pragma unroll
      for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int m = m0 + acc_row<T>(wm, mf, lane, rg);
          if (m < p.M) {
            const float d = acc[mf][nf][rg] + bv - mean;
            s2 += d * d;
          }
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (nok && (lane >> 4) == 0) {
        p.stats[((long)g * p.OC + n) * 2 + 0] = mean;
        p.stats[((long)g * p.OC + n) * 2 + 1] = s2;
      }
    }
    if (nok) {
      const float sc = p.scale ? p.scale[n] : 1.f;
      const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
      for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int m = m0 + acc_row<T>(wm, mf, lane, rg);
          if (m < Mc) {
            float v = (acc[mf][nf][rg] + bv) * sc + sh;
            long orow = m;
            if (par) {
              const int img = m / (RHc * RWc), rem = m - img * (RHc * RWc);
              const int a = rem / RWc, b = rem - a * RWc;
              orow = ((long)img * p.RH + 2 * a + py) * p.RW + 2 * b + px;
            }
            const long o = orow * p.OC + n;
            if (p.res) v += p.res[o];
            if (p.relu) v = fmaxf(v, 0.f);
            p.out[o] = v;
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------
// wgrad: dw[co][(r,s,ci)] = sum_pix dy[pix][co] * x[src(pix,r,s)][ci]
// GEMM view: M' = Co (col image of dy), N' = R*S*Ci (col image gathered from
// x), K' = output pixels.  grid.z splits the pixel range; partial products go
// to a workspace slab per split and wgrad_reduce_kernel sums them.
// ---------------------------------------------------------------------------
struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;  // [nsplit][Co][R*S*Ci]
  int N, H, W, Ci, Ho, Wo, Co, R, S, stride, pad;
  int Mpix, Ncols, pix_per_split;
};

template <class T, bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int BM = T::BM, BN = T::BN;
  using AImg = OperandImage<BM, true>;
  using BImg = OperandImage<BN, true>;
  constexpr int NA4 = BM / 4, NB4 = BN / 4;
  constexpr int PAK = (GK * NA4 + 255) / 256;
  constexpr int PBK = (GK * NB4 + 255) / 256;

  __shared__ __attribute__((aligned(16))) float lds[2 * (AImg::SIZE + BImg::SIZE)];
  constexpr int STAGE = AImg::SIZE + BImg::SIZE;


  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int pix_begin = blockIdx.z * p.pix_per_split;
  int pix_end = pix_begin + p.pix_per_split;
  if (pix_end > p.Mpix) pix_end = p.Mpix;

  // fixed per-thread column decode for the gathered B operand
  int b_krow[PBK], b_col[PBK], b_r[PBK], b_s[PBK], b_ci[PBK];
#pragma unroll
  for (int q = 0; q < PBK; ++q) {
    const int idx = t + 256 * q;
    b_krow[q] = idx / NB4;
    const int n4 = idx - b_krow[q] * NB4;
    b_col[q] = n4 * 4;
    const int n = n0 + n4 * 4;
    const int tap = n / p.Ci;
    b_ci[q] = n - tap * p.Ci;
    b_r[q] = tap / p.S;
    b_s[q] = tap - b_r[q] * p.S;
  }

  f32x4 areg[PAK], breg[PBK];
  const int HoWo = p.Ho * p.Wo;

  auto load_stage = [&](int pix0) {
#pragma unroll
    for (int q = 0; q < PAK; ++q) {
      const int idx = t + 256 * q;
      const int krow = idx / NA4, m4 = idx - krow * NA4;
      const int pix = pix0 + krow, m = m0 + m4 * 4;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (krow < GK && pix < pix_end && m < p.Co) {
        const float* dp = p.dy + (long)pix * p.Co + m;
        if (VEC) {
          v = *reinterpret_cast<const f32x4*>(dp);
        } else {
          v.x = dp[0];
          if (m + 1 < p.Co) v.y = dp[1];
          if (m + 2 < p.Co) v.z = dp[2];
          if (m + 3 < p.Co) v.w = dp[3];
        }
      }
      areg[q] = v;
    }
#pragma unroll
    for (int q = 0; q < PBK; ++q) {
      const int pix = pix0 + b_krow[q];
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (b_krow[q] < GK && pix < pix_end) {
        const int img = pix / HoWo;
        const int rem = pix - img * HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        if (VEC) {
          const int hi = ho * p.stride - p.pad + b_r[q], wi = wo * p.stride - p.pad + b_s[q];
          if (n0 + b_col[q] < p.Ncols && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
            v = *reinterpret_cast<const f32x4*>(p.x + ((long)(img * p.H + hi) * p.W + wi) * p.Ci + b_ci[q]);
        } else {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            e[j] = 0.f;
            const int n = n0 + b_col[q] + j;
            if (n < p.Ncols) {
              const int tap = n / p.Ci, ci = n - tap * p.Ci;
              const int r = tap / p.S, s = tap - r * p.S;
              const int hi = ho * p.stride - p.pad + r, wi = wo * p.stride - p.pad + s;
              if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                e[j] = p.x[((long)(img * p.H + hi) * p.W + wi) * p.Ci + ci];
            }
          }
          v = (f32x4){e[0], e[1], e[2], e[3]};
        }
      }
      breg[q] = v;
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < PAK; ++q) {
      const int idx = t + 256 * q;
      const int krow = idx / NA4, m4 = idx - krow * NA4;
      if (krow < GK) *reinterpret_cast<f32x4*>((lds + buf * STAGE) + krow * AImg::LD + m4 * 4) = areg[q];
    }
#pragma unroll
    for (int q = 0; q < PBK; ++q)
      if (b_krow[q] < GK) *reinterpret_cast<f32x4*>((lds + buf * STAGE + AImg::SIZE) + b_krow[q] * BImg::LD + b_col[q]) = breg[q];
  };

  f32x4 acc[T::MF][T::NF];
  zero_acc<T>(acc);
  const int nk = (pix_end - pix_begin + GK - 1) / GK;
  if (nk > 0) {
    load_stage(pix_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_stage(pix_begin + (kt + 1) * GK);
    mma_stage<T, true, true>(lds + cur * STAGE, lds + cur * STAGE + AImg::SIZE, acc, wm, wn, lane);
    if (kt + 1 < nk) store_stage(cur ^ 1);
    __syncthreads();
  }
  float* outp = p.part + (long)blockIdx.z * p.Co * p.Ncols;
#pragma unroll
  for (int nf = 0; nf < T::NF; ++nf) {
    const int n = n0 + acc_col<T>(wn, nf, lane);
    if (n < p.Ncols) {
#pragma unroll
      for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int m = m0 + acc_row<T>(wm, mf, lane, rg);
          if (m < p.Co) outp[(long)m * p.Ncols + n] = acc[mf][nf][rg];
        }
    }
  }
}

// Slab reduction: a workgroup covers 32 output columns (float4 each when VEC) with 8 split-lanes per column, so even a
// small gradient with hundreds of slabs keeps every slab read independent and spreads over many workgroups (one
// thread per element with a serial loop over the slabs is latency-bound: 384 dependent loads).
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long n, int nsplit, int accumulate, float alpha) {
  __shared__ f32x4 sm[8][32];
  const long cols = VEC ? (n >> 2) : n;
  const int col = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (long base = (long)blockIdx.x * 32; base < cols; base += (long)gridDim.x * 32) {
    const long i = base + col;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < cols) {
      if (VEC) {
        for (int z = zl; z < nsplit; z += 8) s += reinterpret_cast<const f32x4*>(part + (long)z * n)[i];
      } else {
        for (int z = zl; z < nsplit; z += 8) s.x += part[(long)z * n + i];
      }
    }
    sm[zl][col] = s;
    __syncthreads();
    if (zl == 0 && i < cols) {
#pragma unroll
      for (int k = 1; k < 8; ++k) s += sm[k][col];
      if (VEC) {
        s *= alpha;
        if (accumulate) s += reinterpret_cast<const f32x4*>(out)[i];
        reinterpret_cast<f32x4*>(out)[i] = s;
      } else {
        const float v = s.x * alpha;
        out[i] = accumulate ? out[i] + v : v;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int check_desc(const buctd_conv_desc* d, const char* who) {
  BUCTD_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  BUCTD_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Ci > 0 && d->Co > 0 && d->R > 0 && d->S > 0,
                  "%s: non-positive dimension", who);
  BUCTD_CHECK_ARG(d->stride >= 1 && d->pad >= 0, "%s: bad stride/pad", who);
  BUCTD_CHECK_ARG(d->R * d->S <= 64, "%s: filter larger than 8x8 unsupported", who);
  const int ho = (d->H + 2 * d->pad - d->R) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
  BUCTD_CHECK_ARG(ho == d->Ho && wo == d->Wo, "%s: Ho/Wo (%d,%d) inconsistent with input (expect %d,%d)", who,
                  d->Ho, d->Wo, ho, wo);
  BUCTD_CHECK_ARG((long)d->N * d->H * d->W * d->Ci < 2147483647L && (long)d->N * d->Ho * d->Wo * d->Co < 2147483647L,
                  "%s: tensor exceeds 2^31 elements", who);
  return BUCTD_OK;
}

template <class T, bool DGRAD, bool VEC>
static void launch_conv(const ConvArgs& a, hipStream_t st) {
  dim3 grid(ceil_div(a.M, T::BM), ceil_div(a.OC, T::BN));
  if (DGRAD && a.par) {   // four parity classes; the (even, even) one is the largest
    const long mc = (long)a.N * ((a.RH + 1) / 2) * ((a.RW + 1) / 2);
    grid = dim3(ceil_div(mc, T::BM), ceil_div(a.OC, T::BN), 4);
  }
  hipLaunchKernelGGL((conv_gemm_kernel<T, DGRAD, VEC>), grid, dim3(256), 0, st, a);
}

// Tile choice: BN follows the output-channel count (48*2^b HRNet widths, 64 /
// 128 / 256 stem + W32 widths); BM drops to 64 when a 128-row grid would leave
// most of the 256 CUs idle (low-resolution branches).
struct ConvTileSel { int id, BM, WM, MF; };
static ConvTileSel conv_tile_select(int oc, long M, bool vec) {
  if (!vec) return {0, 128, 4, 2};                             // 128x64 generic
  const bool small_m = (long)ceil_div(M, 128) * ceil_div(oc, 96) < 384;
  if (oc <= 16) return {1, 256, 4, 4};                         // 256x16
  if (oc <= 32) return {2, 128, 4, 2};                         // 128x32
  if (oc <= 48) return small_m ? ConvTileSel{3, 64, 4, 1} : ConvTileSel{4, 128, 4, 2};    // 64x48 / 128x48
  if (oc <= 64) return small_m ? ConvTileSel{5, 64, 4, 1} : ConvTileSel{6, 128, 4, 2};    // 64x64 / 128x64
  if (oc % 96 == 0 || (oc % 128 != 0 && oc <= 96))
    return small_m ? ConvTileSel{7, 64, 2, 2} : ConvTileSel{8, 128, 2, 4};                // 64x96 / 128x96
  return small_m ? ConvTileSel{9, 64, 2, 2} : ConvTileSel{10, 128, 2, 4};                 // 64x128 / 128x128
}

template <bool DGRAD>
static void dispatch_conv(const ConvArgs& a, bool vec, hipStream_t st) {
  switch (conv_tile_select(a.OC, a.M, vec).id) {
    case 0: launch_conv<TileCfg<4, 1, 2, 4>, DGRAD, false>(a, st); break;
    case 1: launch_conv<TileCfg<4, 1, 4, 1>, DGRAD, true>(a, st); break;
    case 2: launch_conv<TileCfg<4, 1, 2, 2>, DGRAD, true>(a, st); break;
    case 3: launch_conv<TileCfg<4, 1, 1, 3>, DGRAD, true>(a, st); break;
    case 4: launch_conv<TileCfg<4, 1, 2, 3>, DGRAD, true>(a, st); break;
    case 5: launch_conv<TileCfg<4, 1, 1, 4>, DGRAD, true>(a, st); break;
    case 6: launch_conv<TileCfg<4, 1, 2, 4>, DGRAD, true>(a, st); break;
    case 7: launch_conv<TileCfg<2, 2, 2, 3>, DGRAD, true>(a, st); break;
    case 8: launch_conv<TileCfg<2, 2, 4, 3>, DGRAD, true>(a, st); break;
    case 9: launch_conv<TileCfg<2, 2, 2, 4>, DGRAD, true>(a, st); break;
    default: launch_conv<TileCfg<2, 2, 4, 4>, DGRAD, true>(a, st); break;
  }
}

static bool fwd_vec_ok(const buctd_conv_desc* d) { return d->Ci % 16 == 0; }
static bool dgrad_vec_ok(const buctd_conv_desc* d) { return d->Co % 16 == 0 && d->Ci % 4 == 0; }

constexpr int THIN_TH = 8, THIN_TW = 32, THIN_SPLITS = 1024;

// ---- thin forward ---------------------------------------------------------------
// Forward of the same preNet convolutions with at most 4 OUTPUT channels (64 -> 3 and 3 -> 3 7x7, stride 1, 'same'): the
// implicit-GEMM kernel computes a 16-wide column tile for 3 columns (7.8 ms at 384x288, N = 32).  One thread per output
// pixel of an 8 x 32 tile, three accumulators; the input tile (16 channels at a time, halo included) and the filter chunk
// sit in LDS, a tap costs one 16-byte read of the pixel's channels and broadcast reads of the filter: VALU bound.
struct ThinFwdArgs {
  const float* x;
  const float* w;      // [Co][R][R][Ci]
  const float* bias;
  float* y;
  int N, H, W, Ci, Co, tiles_y, tiles_x;
};

template <int R>
__global__ __launch_bounds__(256) void conv_fwd_thin_kernel(ThinFwdArgs p) {
  constexpr int PADR = R / 2, XH = THIN_TH + R - 1, XW = THIN_TW + R - 1;
  __shared__ __attribute__((aligned(16))) float xs[XH * XW * 16];
  __shared__ __attribute__((aligned(16))) float ws[R * R * 4 * 16];      // [tap][co (4)][16 channels]
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_y * p.tiles_x);
  const int rem = tile - n * p.tiles_y * p.tiles_x;
  const int y0 = (rem / p.tiles_x) * THIN_TH, x0 = (rem - (rem / p.tiles_x) * p.tiles_x) * THIN_TW;
  const int py = t / THIN_TW, px = t - py * THIN_TW;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int nchunks = (p.Ci + 15) / 16;
  for (int c = 0; c < nchunks; ++c) {
    const int c0 = c * 16;
    __syncthreads();
    for (int i = t; i < XH * XW * 4; i += 256) {
      const int pix = i >> 2, q = i & 3;
      const int yy = y0 + pix / XW - PADR, xx = x0 + pix % XW - PADR;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
        const float* src = p.x + ((long)(n * p.H + yy) * p.W + xx) * p.Ci + c0 + q * 4;
        if (c0 + q * 4 + 4 <= p.Ci && (p.Ci & 3) == 0) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + q * 4 + e < p.Ci) v[e] = src[e];
        }
      }
      *reinterpret_cast<f32x4*>(xs + pix * 16 + q * 4) = v;
    }
    for (int i = t; i < R * R * 4 * 16; i += 256) {
      const int ch = i & 15, co = (i >> 4) & 3, tap = i >> 6;
      ws[i] = (co < p.Co && c0 + ch < p.Ci) ? p.w[((long)co * R * R + tap) * p.Ci + c0 + ch] : 0.f;
    }
    __syncthreads();
    const int qn = (p.Ci - c0 >= 16) ? 4 : (p.Ci - c0 + 3) / 4;
    for (int r = 0; r < R; ++r) {
      const float* xr = xs + ((py + r) * XW + px) * 16;
      const float* wr = ws + r * R * 64;
#pragma unroll
      for (int sx = 0; sx < R; ++sx) {
        for (int q = 0; q < qn; ++q) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + sx * 16 + q * 4);
#pragma unroll
          for (int co = 0; co < 4; ++co) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + sx * 64 + co * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co] = __builtin_fmaf(xv[e], wv[e], acc[co]);
          }
        }
      }
    }
  }
  const int yy = y0 + py, xx = x0 + px;
  if (yy < p.H && xx < p.W) {
    float* o = p.y + ((long)(n * p.H + yy) * p.W + xx) * p.Co;
    for (int co = 0; co < p.Co; ++co) o[co] = acc[co] + (p.bias ? p.bias[co] : 0.f);
  }
}

// Wide-input variant (Ci a multiple of 8: the 64 -> 3 7x7 of the preNet).  The kernel above reads five LDS vectors per 16
// FMAs (one pixel per thread, four accumulators of which the preNet uses three) and ran at 19 % of the fp32 vector rate:
// 1.33 ms at 256x192, N = 32 - the largest kernel of the HRNet-W32 step.  Here a thread owns FOUR adjacent pixels of a
// 32 x 32 tile and Co <= 3 accumulators per pixel: the R + 3 input pixels of a filter row are read once for the four
// outputs (10 instead of 28 LDS vectors per row and 4 channels), the filter values are wave-uniform and come through the
// scalar cache (no LDS), and channel pairs go through v_pk_fma_f32 (even / odd channel partial sums, added at the end).
// The input tile is de-interleaved by column mod 4 so that the eight threads of a tile row read consecutive 32-byte
// pixels, rows 16 bytes apart in the bank map.
// CH = 4 serves the convolutions that are thin on BOTH sides (the 3 -> 3 7x7 of the condition branch: 441 FMAs per pixel, 0.48 ms
// at 384x288 on the kernel above, which pads 3 channels to 16 and 3 outputs to 4) and, with TR, their data gradient: the same
// correlation with the filter read transposed and flipped, kernel(out o, r, s, in i) = w[i][R-1-r][R-1-s][o].
constexpr int THIN4_T = 32, THIN4_CH = 8;
typedef float thin_f32x2 __attribute__((ext_vector_type(2)));
template <int R, int CO, int CH, bool TR>
__global__ __launch_bounds__(256) void conv_fwd_thin_px4_kernel(ThinFwdArgs p) {
  constexpr int PADR = R / 2, XH = THIN4_T + R - 1, XW = THIN4_T + R - 1, SL = (XW + 3) / 4;   // SL slots per column plane
  constexpr int Q = CH / 4;                                                                    // 16-byte pieces per pixel
  constexpr int ROWF = 4 * SL * CH + 4;                                                        // floats per tile row (+16 B)
  __shared__ __attribute__((aligned(16))) float xs[XH * ROWF];
  __shared__ __attribute__((aligned(16))) float wsm[CH == 4 ? CO * R * R * 4 : 4];   // CH = 4: the whole filter, [co][r][s][4 ci]
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_y * p.tiles_x);
  const int rem = tile - n * p.tiles_y * p.tiles_x;
  const int y0 = (rem / p.tiles_x) * THIN4_T, x0 = (rem - (rem / p.tiles_x) * p.tiles_x) * THIN4_T;
  const int ty = t >> 3, tx = t & 7;
  const bool vec = (p.Ci & 3) == 0;
  if constexpr (CH == 4) {
    // (one scalar load per filter value in the loop below paced the kernel: 441 dependent s_load_dword per thread)
    for (int i = t; i < CO * R * R * 4; i += 256) {
      const int e = i & 3, tap = (i >> 2) % (R * R), co = (i >> 2) / (R * R);
      const int r = tap / R, sx = tap - r * R;
      float w = 0.f;
      if (e < p.Ci)
        w = TR ? p.w[((long)(e * R + (R - 1 - r)) * R + (R - 1 - sx)) * p.Co + co] : p.w[((long)(co * R + r) * R + sx) * p.Ci + e];
      wsm[i] = w;
    }
  }
  thin_f32x2 acc[4][CO];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[j][co] = (thin_f32x2){0.f, 0.f};
  for (int c0 = 0; c0 < p.Ci; c0 += CH) {
    __syncthreads();
    for (int i = t; i < XH * XW * Q; i += 256) {
      const int pix = i / Q, q = i - pix * Q;
      const int ry = pix / XW, cx = pix - ry * XW;
      const int yy = y0 + ry - PADR, xx = x0 + cx - PADR;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
        const float* src = p.x + ((long)(n * p.H + yy) * p.W + xx) * p.Ci + c0 + q * 4;
        if (vec) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + q * 4 + e < p.Ci) v[e] = src[e];
        }
      }
      *reinterpret_cast<f32x4*>(xs + ry * ROWF + ((cx & 3) * SL + (cx >> 2)) * CH + q * 4) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      // the R + 3 pixels this thread's four outputs read in filter row r: column 4 tx + k, k = 0 .. R + 2
      f32x4 xw[R + 3][Q];
      const float* xr = xs + (ty + r) * ROWF + tx * CH;
#pragma unroll
      for (int k = 0; k < R + 3; ++k)
#pragma unroll
        for (int q = 0; q < Q; ++q)
          xw[k][q] = *reinterpret_cast<const f32x4*>(xr + ((k & 3) * SL + (k >> 2)) * CH + q * 4);
#pragma unroll
      for (int sx = 0; sx < R; ++sx)
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          // wave-uniform filter values: scalar loads
          f32x4 wv[Q];
          if constexpr (CH == 8) {
            const float* wp = p.w + ((long)(co * R + r) * R + sx) * p.Ci + c0;
            wv[0] = *reinterpret_cast<const f32x4*>(wp);
            wv[1] = *reinterpret_cast<const f32x4*>(wp + 4);
          } else {
            wv[0] = *reinterpret_cast<const f32x4*>(wsm + ((co * R + r) * R + sx) * 4);     // broadcast read (Ci <= 4: one chunk)
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            thin_f32x2 s = acc[j][co];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              const f32x4 a = xw[j + sx][q];
              s = __builtin_elementwise_fma((thin_f32x2){a.x, a.y}, (thin_f32x2){wv[q].x, wv[q].y}, s);
              s = __builtin_elementwise_fma((thin_f32x2){a.z, a.w}, (thin_f32x2){wv[q].z, wv[q].w}, s);
            }
            acc[j][co] = s;
          }
        }
    }
  }
  const int yy = y0 + ty;
  if (yy < p.H) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xx = x0 + 4 * tx + j;
      if (xx < p.W) {
        float* o = p.y + ((long)(n * p.H + yy) * p.W + xx) * p.Co;
#pragma unroll
        for (int co = 0; co < CO; ++co) o[co] = (acc[j][co].x + acc[j][co].y) + (p.bias ? p.bias[co] : 0.f);
      }
    }
  }
}

template <int CH, bool TR>
static void launch_thin_px4(const ThinFwdArgs& ta, dim3 grid, hipStream_t st) {
  if (ta.Co == 3) hipLaunchKernelGGL((conv_fwd_thin_px4_kernel<7, 3, CH, TR>), grid, dim3(256), 0, st, ta);
  else if (ta.Co == 2) hipLaunchKernelGGL((conv_fwd_thin_px4_kernel<7, 2, CH, TR>), grid, dim3(256), 0, st, ta);
  else hipLaunchKernelGGL((conv_fwd_thin_px4_kernel<7, 1, CH, TR>), grid, dim3(256), 0, st, ta);
}

// Data gradient of the same convolutions (<= 4 OUTPUT channels of the forward conv: dy is thin, dx wide): one thread per
// dx pixel, 16 input channels per pass; dy tile (halo) in LDS, filter chunk [tap][co][16 ci] read as broadcasts.
struct ThinDgradArgs {
  const float* dy;
  const float* w;      // [Co][R][R][Ci]
  float* dx;
  int N, H, W, Ci, Co, tiles_y, tiles_x;
};

template <int R>
__global__ __launch_bounds__(256) void conv_dgrad_thin_kernel(ThinDgradArgs p) {
  constexpr int PADR = R / 2, XH = THIN_TH + R - 1, XW = THIN_TW + R - 1;
  __shared__ __attribute__((aligned(16))) float ds[XH * XW * 4];
  __shared__ __attribute__((aligned(16))) float ws[R * R * 4 * 16];      // [tap][co (4)][16 input channels]
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_y * p.tiles_x);
  const int rem = tile - n * p.tiles_y * p.tiles_x;
  const int y0 = (rem / p.tiles_x) * THIN_TH, x0 = (rem - (rem / p.tiles_x) * p.tiles_x) * THIN_TW;
  const int py = t / THIN_TW, px = t - py * THIN_TW;
  for (int i = t; i < XH * XW; i += 256) {
    const int yy = y0 + i / XW - PADR, xx = x0 + i % XW - PADR;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
      const float* src = p.dy + ((long)(n * p.H + yy) * p.W + xx) * p.Co;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < p.Co) v[e] = src[e];
    }
    *reinterpret_cast<f32x4*>(ds + i * 4) = v;
  }
  const int nchunks = (p.Ci + 15) / 16;
  const int yy = y0 + py, xx = x0 + px;
  for (int c = 0; c < nchunks; ++c) {
    const int c0 = c * 16;
    __syncthreads();
    for (int i = t; i < R * R * 4 * 16; i += 256) {
      const int ch = i & 15, co = (i >> 4) & 3, tap = i >> 6;
      ws[i] = (co < p.Co && c0 + ch < p.Ci) ? p.w[((long)co * R * R + tap) * p.Ci + c0 + ch] : 0.f;
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
      const float* dr = ds + ((py + R - 1 - r) * XW + px + R - 1) * 4;
      const float* wr = ws + r * R * 64;
#pragma unroll
      for (int sx = 0; sx < R; ++sx) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dr - sx * 4);
#pragma unroll
        for (int co = 0; co < 3; ++co) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + sx * 64 + co * 16 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = __builtin_fmaf(dv[co], wv[e], acc[q][e]);
          }
        }
        if (p.Co > 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + sx * 64 + 48 + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = __builtin_fmaf(dv[3], wv[e], acc[q][e]);
          }
        }
      }
    }
    if (yy < p.H && xx < p.W) {
      float* o = p.dx + ((long)(n * p.H + yy) * p.W + xx) * p.Ci + c0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (c0 + q * 4 + 4 <= p.Ci && (p.Ci & 3) == 0) *reinterpret_cast<f32x4*>(o + q * 4) = acc[q];
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + q * 4 + e < p.Ci) o[q * 4 + e] = acc[q][e];
        }
      }
    }
  }
}

// The same data gradient with FOUR adjacent dx pixels per thread (whole 16-channel chunks, at most three dY channels: the
// 64 -> 3 convolution).  The kernel above reads 13 LDS vectors per tap for 48 FMAs and is paced by the LDS, not by the VALU
// (v_pk_fma_f32 alone changes nothing: 1.13 ms at 384x288); here the 12 broadcast filter vectors of a tap serve four pixels
// and the R + 3 dY pixels of a filter row are read once for them: 94 LDS reads against 672 packed FMAs per filter row.
template <int R>
__global__ __launch_bounds__(256) void conv_dgrad_thin_px4_kernel(ThinDgradArgs p) {
  constexpr int PADR = R / 2, XH = THIN4_T + R - 1, XW = THIN4_T + R - 1, SL = (XW + 3) / 4;
  constexpr int ROWF = 4 * SL * 4 + 4;                                       // floats per tile row: [column mod 4][slot][4]
  __shared__ __attribute__((aligned(16))) float ds[XH * ROWF];
  __shared__ __attribute__((aligned(16))) float ws[R * R * 3 * 16];          // [tap][co (3)][16 input channels]
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_y * p.tiles_x);
  const int rem = tile - n * p.tiles_y * p.tiles_x;
  const int y0 = (rem / p.tiles_x) * THIN4_T, x0 = (rem - (rem / p.tiles_x) * p.tiles_x) * THIN4_T;
  const int ty = t >> 3, tx = t & 7;
  for (int i = t; i < XH * XW; i += 256) {
    const int ry = i / XW, cx = i - ry * XW;
    const int yy = y0 + ry - PADR, xx = x0 + cx - PADR;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
      const float* src = p.dy + ((long)(n * p.H + yy) * p.W + xx) * p.Co;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < p.Co) v[e] = src[e];
    }
    *reinterpret_cast<f32x4*>(ds + ry * ROWF + ((cx & 3) * SL + (cx >> 2)) * 4) = v;
  }
  const int yy = y0 + ty;
  for (int c0 = 0; c0 < p.Ci; c0 += 16) {
    __syncthreads();
    for (int i = t; i < R * R * 3 * 16; i += 256) {
      const int ch = i & 15, co = (i >> 4) % 3, tap = i / 48;
      ws[i] = co < p.Co ? p.w[((long)co * R * R + tap) * p.Ci + c0 + ch] : 0.f;
    }
    __syncthreads();
    thin_f32x2 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[j][k] = (thin_f32x2){0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
      // dY pixels (ty + R - 1 - r, 4 tx + k), k = 0 .. R + 2, in tile coordinates: pixel j and tap sx read k = j + R - 1 - sx
      f32x4 dw[R + 3];
      const float* dr = ds + (ty + R - 1 - r) * ROWF + tx * 4;
#pragma unroll
      for (int k = 0; k < R + 3; ++k) dw[k] = *reinterpret_cast<const f32x4*>(dr + ((k & 3) * SL + (k >> 2)) * 4);
#pragma unroll
      for (int sx = 0; sx < R; ++sx)
#pragma unroll
        for (int co = 0; co < 3; ++co) {
          const float* wp = ws + ((r * R + sx) * 3 + co) * 16;
          f32x4 wv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const f32x4*>(wp + q * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = dw[j + R - 1 - sx][co];
            const thin_f32x2 dd = {d, d};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              acc[j][2 * q] = __builtin_elementwise_fma(dd, (thin_f32x2){wv[q].x, wv[q].y}, acc[j][2 * q]);
              acc[j][2 * q + 1] = __builtin_elementwise_fma(dd, (thin_f32x2){wv[q].z, wv[q].w}, acc[j][2 * q + 1]);
            }
          }
        }
    }
    if (yy < p.H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = x0 + 4 * tx + j;
        if (xx < p.W) {
          float* o = p.dx + ((long)(n * p.H + yy) * p.W + xx) * p.Ci + c0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(o + q * 4) = (f32x4){acc[j][2 * q].x, acc[j][2 * q].y, acc[j][2 * q + 1].x, acc[j][2 * q + 1].y};
        }
      }
    }
  }
}

// Data gradient of a stride-2 3x3 convolution with <= 4 INPUT channels (the HRNet stem conv1 behind a preNet, whose
// 3-channel output needs a gradient: pose_hrnet.py:287 fed by 452-458).  One thread per dx pixel; of the nine taps only
// those whose output coordinate is an integer contribute (2.25 on average); dy rows come through L1, the filter from LDS.
struct ThinS2Args {
  const float* dy;
  const float* w;      // [Co][3][3][Ci]
  float* dx;
  int N, H, W, Ci, Co, Ho, Wo;
};

__global__ __launch_bounds__(256) void conv_dgrad_thin_s2_kernel(ThinS2Args p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];      // [tap][co][4]
  for (int i = threadIdx.x; i < 9 * p.Co * 4; i += 256) {
    const int ci = i & 3, co = (i >> 2) % p.Co, tap = i / (4 * p.Co);
    wsm[i] = ci < p.Ci ? p.w[((long)co * 9 + tap) * p.Ci + ci] : 0.f;
  }
  __syncthreads();
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long)p.N * p.H * p.W) return;
  const int n = (int)(pix / ((long)p.H * p.W));
  const int rem = (int)(pix - (long)n * p.H * p.W);
  const int y = rem / p.W, x = rem - y * p.W;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < 3; ++r) {
    const int ty = y + 1 - r;
    if (ty < 0 || (ty & 1) || (ty >> 1) >= p.Ho) continue;
    for (int sx = 0; sx < 3; ++sx) {
      const int tx = x + 1 - sx;
      if (tx < 0 || (tx & 1) || (tx >> 1) >= p.Wo) continue;
      const float* dr = p.dy + (((long)n * p.Ho + (ty >> 1)) * p.Wo + (tx >> 1)) * p.Co;
      const float* wr = wsm + (r * 3 + sx) * p.Co * 4;
      for (int co = 0; co < p.Co; co += 4) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dr + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + (co + j) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(dv[j], wv[e], acc[e]);
        }
      }
    }
  }
  float* o = p.dx + pix * p.Ci;
  for (int e = 0; e < p.Ci; ++e) o[e] = acc[e];
}

// ---- thin INPUT, 64 outputs: the first convolution of the preNet (3 -> 64, 3x3, 'same') -----------------------------------
// The implicit-GEMM kernel pads the 27-deep reduction to 32 and reads the 3-channel pixels scalar: 0.59 ms at 384x288 (N = 32)
// with a second pass over the 906 MB output for the BatchNorm partials.  Output-bound work (27 FMAs and 256 bytes per pixel):
// lane = output channel with its 27 filter values in registers; a workgroup owns one image row, each of its four wavefronts a
// quarter of it (= one statistics group), four pixels per step: their 3 x 6 input pixels are wave-uniform and come out of the
// three zero-padded rows in LDS as aligned 16-byte broadcast reads, pixel pairs go through v_pk_fma_f32, every pixel leaves as
// one 256-byte store, and the Welford partial of the group (shifted sums: K = the group's first value) is formed on the way.
struct ThinInArgs {
  const float* x;
  const float* w;      // [64][3][3][Ci]
  const float* bias;
  float* y;
  float* part;         // [groups][64][2] (mean, M2) or null
  int N, H, W, Ci;     // input size
  int Ho, Wo;          // output size (stride ST: the stem's first convolution is the stride-2 case)
};
template <int CI, int ST>
__global__ __launch_bounds__(256) void conv3x3_thin_in_fwd_kernel(ThinInArgs p) {
  extern __shared__ __attribute__((aligned(16))) float xrows[];          // [3][(W + 2) * CI, padded to 16 B] + one zero row
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = blockIdx.x / p.Ho, oy = blockIdx.x - n * p.Ho;      // one OUTPUT row per workgroup
  const int rowf = ((p.W + 2) * CI + 3) & ~3;                            // floats per padded row
  for (int i = t; i < 3 * rowf; i += 256) {
    const int rr = i / rowf, c = i - rr * rowf;                          // c = (xx + 1) * CI + ci
    const int ry = ST * oy - 1 + rr, px = c / CI - 1;
    float v = 0.f;
    if (ry >= 0 && ry < p.H && px >= 0 && px < p.W) v = p.x[((long)(n * p.H + ry) * p.W) * CI + c - CI];
    xrows[i] = v;
  }
  float wr[9 * CI];
#pragma unroll
  for (int k = 0; k < 9 * CI; ++k) wr[k] = p.w[lane * 9 * CI + k];
  const float b = p.bias ? p.bias[lane] : 0.f;
  __syncthreads();
  const int q = p.Wo >> 2;                                               // output pixels per wavefront (a multiple of 4)
  const int xa = wave * q;
  float K = 0.f, s1 = 0.f, s2 = 0.f;
  float* yo = p.y + ((long)blockIdx.x * p.Wo + xa) * 64 + lane;
  for (int x0 = 0; x0 < q; x0 += 4) {
    // padded input pixels ST (xa + x0) .. + 3 ST + 2 of the three rows: (3 ST + 3) * CI floats from a 16-byte aligned offset
    constexpr int NV = ((3 * ST + 3) * CI + 3) / 4;
    thin_f32x2 acc[2] = {(thin_f32x2){b, b}, (thin_f32x2){b, b}};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float xv[NV * 4];
      const float* src = xrows + r * rowf + ST * (xa + x0) * CI;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(src + 4 * v);
        xv[4 * v] = u.x; xv[4 * v + 1] = u.y; xv[4 * v + 2] = u.z; xv[4 * v + 3] = u.w;
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const float w = wr[(r * 3 + dx) * CI + ci];
          const thin_f32x2 ww = {w, w};
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)      // pixels 2 pr, 2 pr + 1: padded columns 2 pr + dx, 2 pr + 1 + dx
            acc[pr] = __builtin_elementwise_fma((thin_f32x2){xv[(ST * 2 * pr + dx) * CI + ci], xv[(ST * (2 * pr + 1) + dx) * CI + ci]}, ww, acc[pr]);
        }
    }
    const float o[4] = {acc[0].x, acc[0].y, acc[1].x, acc[1].y};
    if (x0 == 0) K = o[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      yo[(long)(x0 + j) * 64] = o[j];
      const float d = o[j] - K;
      s1 += d;
      s2 = __builtin_fmaf(d, d, s2);
    }
  }
  if (p.part) {
    const float inv = 1.f / (float)q;
    float m2 = s2 - s1 * s1 * inv;
    if (m2 < 0.f) m2 = 0.f;
    float* o = p.part + (((long)blockIdx.x * 4 + wave) * 64 + lane) * 2;
    o[0] = K + s1 * inv;
    o[1] = m2;
  }
}
static bool fwd_thin_in_ok(const buctd_conv_desc* d) {
  if (!(d->R == 3 && d->S == 3 && d->pad == 1 && d->Co == 64 && d->Ci >= 1 && d->Ci <= 4)) return false;
  if (d->stride == 1) { if (d->Ho != d->H || d->Wo != d->W) return false; }
  else if (d->stride == 2) { if ((d->H & 1) || (d->W & 1) || d->Ho != d->H / 2 || d->Wo != d->W / 2) return false; }
  else return false;
  return d->Wo % 16 == 0 && d->Wo >= 64 && (size_t)3 * (d->W + 3) * d->Ci * sizeof(float) <= 60 * 1024 &&
         (long)d->N * d->Ho * d->Wo >= 4096;
}

// Weight gradient of the same convolution, same traversal: lane = output channel holds dW[co][3][3][Ci] (27 accumulators),
// a workgroup walks whole image rows (the three padded input rows in LDS, wave-uniform 16-byte broadcast reads, four pixels
// per step), each wavefront a quarter of the row: one coalesced 256-byte dY read and 27 FMAs per pixel.  The implicit-GEMM
// kernel took 0.78 ms at 384x288 (N = 32) for the one 906 MB pass over dY.  One slab [64][27] per wavefront, reduced by
// splitk_reduce_kernel (the usual [split][Co][R][S][Ci] layout).
constexpr int THIN_IN_WG = 1024;       // workgroups (each walks N * H / THIN_IN_WG image rows)
struct ThinInWgradArgs {
  const float* x;
  const float* dy;
  float* part;
  int N, H, W, Ci, Ho, Wo;
};
template <int CI, int ST>
__global__ __launch_bounds__(256) void conv3x3_thin_in_wgrad_kernel(ThinInWgradArgs p) {
  extern __shared__ __attribute__((aligned(16))) float xrows[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int rowf = ((p.W + 2) * CI + 3) & ~3;
  const int q = p.Wo >> 2, xa = wave * q;
  float acc[9 * CI];
#pragma unroll
  for (int k = 0; k < 9 * CI; ++k) acc[k] = 0.f;
  for (int row = blockIdx.x; row < p.N * p.Ho; row += gridDim.x) {      // output rows
    const int n = row / p.Ho, oy = row - n * p.Ho;
    __syncthreads();                 // the previous row's reads are done
    for (int i = t; i < 3 * rowf; i += 256) {
      const int rr = i / rowf, c = i - rr * rowf;
      const int ry = ST * oy - 1 + rr, px = c / CI - 1;
      float v = 0.f;
      if (ry >= 0 && ry < p.H && px >= 0 && px < p.W) v = p.x[((long)(n * p.H + ry) * p.W) * CI + c - CI];
      xrows[i] = v;
    }
    __syncthreads();
    const float* dyp = p.dy + ((long)row * p.Wo + xa) * 64 + lane;
    for (int x0 = 0; x0 < q; x0 += 4) {
      constexpr int NV = ((3 * ST + 3) * CI + 3) / 4;
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = dyp[(long)(x0 + j) * 64];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float xv[NV * 4];
        const float* src = xrows + r * rowf + ST * (xa + x0) * CI;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(src + 4 * v);
          xv[4 * v] = u.x; xv[4 * v + 1] = u.y; xv[4 * v + 2] = u.z; xv[4 * v + 3] = u.w;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[(r * 3 + dx) * CI + ci] = __builtin_fmaf(g[j], xv[(ST * j + dx) * CI + ci], acc[(r * 3 + dx) * CI + ci]);
      }
    }
  }
  // the four wavefronts' sums meet in LDS: one slab per workgroup
  __syncthreads();
  float* red = xrows;                                  // [4][9 * CI][64]
#pragma unroll
  for (int k = 0; k < 9 * CI; ++k) red[(wave * 9 * CI + k) * 64 + lane] = acc[k];
  __syncthreads();
  float* o = p.part + (size_t)blockIdx.x * 64 * (9 * CI);
  for (int i = t; i < 64 * 9 * CI; i += 256) {
    const int co = i / (9 * CI), k = i - co * (9 * CI);
    o[i] = (red[(0 * 9 * CI + k) * 64 + co] + red[(1 * 9 * CI + k) * 64 + co]) +
           (red[(2 * 9 * CI + k) * 64 + co] + red[(3 * 9 * CI + k) * 64 + co]);
  }
}
static size_t thin_in_wgrad_lds(const buctd_conv_desc* d) {
  const size_t rows = (size_t)3 * ((((size_t)d->W + 2) * d->Ci + 3) & ~(size_t)3) * sizeof(float);
  const size_t red = (size_t)4 * 9 * d->Ci * 64 * sizeof(float);
  return rows > red ? rows : red;
}
static int thin_in_wgs(const buctd_conv_desc* d) {
  const int rows = d->N * d->Ho;
  return rows < THIN_IN_WG ? rows : THIN_IN_WG;
}

static bool dgrad_thin_s2_ok(const buctd_conv_desc* d) {
  return d->stride == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && d->Ci <= 4 && d->Co % 4 == 0 && d->Co <= 256 &&
         (long)d->N * d->H * d->W >= 4096;
}

static bool fwd_thin_ok(const buctd_conv_desc* d) {
  return d->stride == 1 && d->R == 7 && d->S == 7 && d->pad == 3 && d->Co <= 4 && d->Ci <= 64 && d->Ho == d->H &&
         d->Wo == d->W && (long)d->N * d->H * d->W >= 4096;
}

extern "C" int buctd_conv2d_stats_groups(const buctd_conv_desc* d, int transposed, int* ngroups,
                                         int* rows_per_group) {
  int rc = check_desc(d, "buctd_conv2d_stats_groups");
  if (rc) return rc;
  BUCTD_CHECK_ARG(ngroups && rows_per_group, "buctd_conv2d_stats_groups: null output");
  const long M = transposed ? (long)d->N * d->H * d->W : (long)d->N * d->Ho * d->Wo;
  const int oc = transposed ? d->Ci : d->Co;
  if (!transposed && fwd_thin_in_ok(d)) {        // one group per wavefront of conv3x3_thin_in_fwd_kernel: a quarter image row
    *rows_per_group = d->Wo / 4;
    *ngroups = d->N * d->Ho * 4;
    return BUCTD_OK;
  }
  const ConvTileSel ts = conv_tile_select(oc, M, transposed ? dgrad_vec_ok(d) : fwd_vec_ok(d));
  *rows_per_group = ts.MF * 16;
  *ngroups = ceil_div(M, ts.BM) * ts.WM;
  return BUCTD_OK;
}

extern "C" int buctd_conv2d_fwd_thin(const buctd_conv_desc* d) {
  return (d && check_desc(d, "buctd_conv2d_fwd_thin") == BUCTD_OK && fwd_thin_ok(d)) ? 1 : 0;
}

extern "C" int buctd_conv2d_fwd(const buctd_conv_desc* d, const float* x, const float* w, const float* bias,
                                const float* scale, const float* shift, const float* residual, int relu, float* y,
                                float* stats_partials, void* stream) {
  int rc = check_desc(d, "buctd_conv2d_fwd");
  if (rc) return rc;
  BUCTD_CHECK_ARG(x && w && y, "buctd_conv2d_fwd: null tensor pointer");
  BUCTD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "buctd_conv2d_fwd: scale and shift go together");
  if (fwd_thin_ok(d) && !scale && !residual && !relu && !stats_partials) {     // plain conv (+ bias), <= 4 output channels
    ThinFwdArgs ta;
    ta.x = x; ta.w = w; ta.bias = bias; ta.y = y;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Co = d->Co;
    if ((d->Ci % THIN4_CH == 0 || d->Ci <= 4) && d->Co <= 3) {          // four pixels per thread (64 -> 3, 3 -> 3)
      ta.tiles_y = ceil_div(d->H, THIN4_T); ta.tiles_x = ceil_div(d->W, THIN4_T);
      const dim3 grid(d->N * ta.tiles_y * ta.tiles_x);
      if (d->Ci <= 4) launch_thin_px4<4, false>(ta, grid, (hipStream_t)stream);
      else launch_thin_px4<8, false>(ta, grid, (hipStream_t)stream);
      BUCTD_CHECK_LAUNCH("buctd_conv2d_fwd(thin, four pixels per thread)");
      return BUCTD_OK;
    }
    ta.tiles_y = ceil_div(d->H, THIN_TH); ta.tiles_x = ceil_div(d->W, THIN_TW);
    hipLaunchKernelGGL((conv_fwd_thin_kernel<7>), dim3(d->N * ta.tiles_y * ta.tiles_x), dim3(256), 0, (hipStream_t)stream, ta);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_fwd(thin)");
    return BUCTD_OK;
  }
  // buctd_conv2d_stats_groups reports the thin kernel's grouping for these shapes whatever the epilogue: statistics together
  // with an epilogue the thin kernel does not have would send the launch to the implicit-GEMM kernel, whose (more) groups
  // overrun the caller's partials buffer
  BUCTD_CHECK_ARG(!(fwd_thin_in_ok(d) && stats_partials && (scale || residual || relu)),
                  "buctd_conv2d_fwd: statistics of a thin-input convolution cannot be combined with scale / residual / relu");
  if (fwd_thin_in_ok(d) && !scale && !residual && !relu) {      // 3 -> 64 3x3 of the preNet: lane = output channel
    ThinInArgs ta;
    ta.x = x; ta.w = w; ta.bias = bias; ta.y = y; ta.part = stats_partials;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Ho = d->Ho; ta.Wo = d->Wo;
    const dim3 grid(d->N * d->Ho);
    const size_t lds = (size_t)3 * ((((size_t)d->W + 2) * d->Ci + 3) & ~(size_t)3) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define THIN_IN_FWD(ci) \
    if (d->Ci == ci) { \
      if (d->stride == 1) hipLaunchKernelGGL((conv3x3_thin_in_fwd_kernel<ci, 1>), grid, dim3(256), lds, st, ta); \
      else hipLaunchKernelGGL((conv3x3_thin_in_fwd_kernel<ci, 2>), grid, dim3(256), lds, st, ta); \
    }
    THIN_IN_FWD(1) THIN_IN_FWD(2) THIN_IN_FWD(3) THIN_IN_FWD(4)
#undef THIN_IN_FWD
    BUCTD_CHECK_LAUNCH("buctd_conv2d_fwd(thin input)");
    return BUCTD_OK;
  }
  ConvArgs a;
  a.src = x; a.w = w; a.out = y; a.bias = bias; a.scale = scale; a.shift = shift; a.res = residual;
  a.stats = stats_partials;
  a.N = d->N; a.SH = d->H; a.SW = d->W; a.SC = d->Ci;
  a.RH = d->Ho; a.RW = d->Wo; a.OC = d->Co;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad;
  a.M = d->N * d->Ho * d->Wo; a.K = d->R * d->S * d->Ci;
  a.wRSCi = d->R * d->S * d->Ci; a.wCi = d->Ci;
  a.relu = relu; a.par = 0;
  dispatch_conv<false>(a, fwd_vec_ok(d), (hipStream_t)stream);
  BUCTD_CHECK_LAUNCH("buctd_conv2d_fwd");
  return BUCTD_OK;
}

extern "C" int buctd_conv2d_dgrad(const buctd_conv_desc* d, const float* dy, const float* w, const float* bias,
                                  float* dx, float* stats_partials, void* stream) {
  int rc = check_desc(d, "buctd_conv2d_dgrad");
  if (rc) return rc;
  BUCTD_CHECK_ARG(dy && w && dx, "buctd_conv2d_dgrad: null tensor pointer");
  if (dgrad_thin_s2_ok(d) && !bias && !stats_partials) {  // stem conv1 behind a preNet: thin dx, stride 2
    ThinS2Args ta;
    ta.dy = dy; ta.w = w; ta.dx = dx;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Co = d->Co; ta.Ho = d->Ho; ta.Wo = d->Wo;
    const long px = (long)d->N * d->H * d->W;
    hipLaunchKernelGGL(conv_dgrad_thin_s2_kernel, dim3((unsigned)ceil_div(px, 256)), dim3(256), (size_t)9 * d->Co * 4 * sizeof(float),
                       (hipStream_t)stream, ta);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_dgrad(thin s2)");
    return BUCTD_OK;
  }
  if (fwd_thin_ok(d) && d->Ci <= 3 && !bias && !stats_partials) {      // thin on both sides (3 -> 3 7x7): the forward kernel, filter transposed
    ThinFwdArgs ta;
    ta.x = dy; ta.w = w; ta.bias = nullptr; ta.y = dx;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Co; ta.Co = d->Ci;
    ta.tiles_y = ceil_div(d->H, THIN4_T); ta.tiles_x = ceil_div(d->W, THIN4_T);
    launch_thin_px4<4, true>(ta, dim3(d->N * ta.tiles_y * ta.tiles_x), (hipStream_t)stream);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_dgrad(thin, four pixels per thread)");
    return BUCTD_OK;
  }
  if (fwd_thin_ok(d) && !bias && !stats_partials && d->Ci % 16 == 0 && d->Co <= 3) {      // 64 -> 3: four dx pixels per thread
    ThinDgradArgs ta;
    ta.dy = dy; ta.w = w; ta.dx = dx;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Co = d->Co;
    ta.tiles_y = ceil_div(d->H, THIN4_T); ta.tiles_x = ceil_div(d->W, THIN4_T);
    hipLaunchKernelGGL((conv_dgrad_thin_px4_kernel<7>), dim3(d->N * ta.tiles_y * ta.tiles_x), dim3(256), 0, (hipStream_t)stream, ta);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_dgrad(thin, four pixels per thread)");
    return BUCTD_OK;
  }
  if (fwd_thin_ok(d) && !bias && !stats_partials) {      // the preNet 7x7 with <= 4 output channels: thin dy, wide dx
    ThinDgradArgs ta;
    ta.dy = dy; ta.w = w; ta.dx = dx;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Co = d->Co;
    ta.tiles_y = ceil_div(d->H, THIN_TH); ta.tiles_x = ceil_div(d->W, THIN_TW);
    hipLaunchKernelGGL((conv_dgrad_thin_kernel<7>), dim3(d->N * ta.tiles_y * ta.tiles_x), dim3(256), 0, (hipStream_t)stream, ta);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_dgrad(thin)");
    return BUCTD_OK;
  }
  ConvArgs a;
  a.src = dy; a.w = w; a.out = dx; a.bias = bias; a.scale = nullptr; a.shift = nullptr; a.res = nullptr;
  a.stats = stats_partials;
  a.N = d->N; a.SH = d->Ho; a.SW = d->Wo; a.SC = d->Co;
  a.RH = d->H; a.RW = d->W; a.OC = d->Ci;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad;
  a.M = d->N * d->H * d->W; a.K = d->R * d->S * d->Co;
  a.wRSCi = d->R * d->S * d->Ci; a.wCi = d->Ci;
  a.relu = 0;
  a.par = (d->stride == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && !stats_partials && dgrad_vec_ok(d)) ? 1 : 0;
  dispatch_conv<true>(a, dgrad_vec_ok(d), (hipStream_t)stream);
  BUCTD_CHECK_LAUNCH("buctd_conv2d_dgrad");
  return BUCTD_OK;
}

// ---- thin weight gradient -----------------------------------------------------
// Weight gradient of the stride-1 'same' convolutions with three channels on one side - the full-resolution preNet of
// BUCTD-preNet (pose_hrnet.py:431-442: 3 -> 64 3x3, 64 -> 3 7x7, 3 -> 3 7x7 on the 384x288 crop).  The implicit-GEMM
// kernel pads the 3-channel side to a 64-wide tile and loads it scalar: 9.2 ms per launch, 27 ms of a 114 ms step.
// Here a thread owns work items (tap, group of 4 channels of the wide side) and a 4 x 4 block of dW per item and chunk;
// the workgroup walks 8 x 32 pixel tiles: the tile of the wide operand (16 channels at a time) and of the thin one sit
// in LDS, every pixel costs two 16-byte LDS reads (one of them a broadcast) and 16 FMAs per item.  VALU bound
// (64 FMA/clk/CU): ~2 ms for the 7x7 64 -> 3 filter at N = 32.  Partial sums per workgroup go to slabs of the usual
// [split][Co][R][S][Ci] layout, reduced by splitk_reduce_kernel.
struct ThinArgs {
  const float* x;
  const float* dy;
  float* part;
  int N, H, W, Ci, Co, tiles_y, tiles_x, ntiles;
};


template <int R, bool WIDE_DY>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(ThinArgs p) {
  constexpr int PADR = R / 2, XH = THIN_TH + R - 1, XW = THIN_TW + R - 1;
  constexpr int XC = WIDE_DY ? 4 : 16, DC = WIDE_DY ? 16 : 4;          // channels per LDS pixel of the two tiles
  __shared__ __attribute__((aligned(16))) float xs[XH * XW * XC];
  __shared__ __attribute__((aligned(16))) float ds[THIN_TH * THIN_TW * DC];
  const int t = threadIdx.x;
  const int wide = WIDE_DY ? p.Co : p.Ci;
  const int nchunks = (wide + 15) / 16;
  // item of this thread: tap (r, s) and 4-channel group g of the current 16-channel chunk
  const int groups = wide >= 16 ? 4 : (wide + 3) / 4;
  const int nitems = R * R * groups;
  const bool active = t < nitems;
  const int tap = active ? t / groups : 0, g = active ? t - tap * groups : 0;
  const int r = tap / R, sx = tap - r * R;
  const int xoff = (r * XW + sx) * XC + (WIDE_DY ? 0 : g * 4);
  const int doff = WIDE_DY ? g * 4 : 0;
  float acc[4][16];                        // [chunk][dy channel i (4)][x channel j (4)]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int n = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - n * p.tiles_y * p.tiles_x;
    const int y0 = (rem / p.tiles_x) * THIN_TH, x0 = (rem - (rem / p.tiles_x) * p.tiles_x) * THIN_TW;
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      __syncthreads();                     // previous chunk / tile consumed
      // stage the x tile (halo, zero outside the image) and the dy tile; the thin side once per tile (c == 0)
      if (!WIDE_DY || c == 0) {
        const int xc0 = WIDE_DY ? 0 : c * 16;
        for (int i = t; i < XH * XW * (XC / 4); i += 256) {
          const int pix = i / (XC / 4), q = i - pix * (XC / 4);
          const int yy = y0 + pix / XW - PADR, xx = x0 + pix % XW - PADR;
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
            const float* src = p.x + ((long)(n * p.H + yy) * p.W + xx) * p.Ci + xc0 + q * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (xc0 + q * 4 + e < p.Ci) v[e] = src[e];
          }
          *reinterpret_cast<f32x4*>(xs + pix * XC + q * 4) = v;
        }
      }
      if (WIDE_DY || c == 0) {
        const int dc0 = WIDE_DY ? c * 16 : 0;
        for (int i = t; i < THIN_TH * THIN_TW * (DC / 4); i += 256) {
          const int pix = i / (DC / 4), q = i - pix * (DC / 4);
          const int yy = y0 + pix / THIN_TW, xx = x0 + pix % THIN_TW;
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (yy < p.H && xx < p.W) {
            const float* src = p.dy + ((long)(n * p.H + yy) * p.W + xx) * p.Co + dc0 + q * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (dc0 + q * 4 + e < p.Co) v[e] = src[e];
          }
          *reinterpret_cast<f32x4*>(ds + pix * DC + q * 4) = v;
        }
      }
      __syncthreads();
      if (active && c * 16 + g * 4 < wide) {
        float a16[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a16[e] = acc[c][e];
        for (int py = 0; py < THIN_TH; ++py) {
          const float* xr = xs + xoff + py * XW * XC;
          const float* dr = ds + doff + py * THIN_TW * DC;
#pragma unroll 4
          for (int px = 0; px < THIN_TW; ++px) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + px * XC);
            const f32x4 dv = *reinterpret_cast<const f32x4*>(dr + px * DC);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) a16[i * 4 + j] = __builtin_fmaf(dv[i], xv[j], a16[i * 4 + j]);
          }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = a16[e];
      }
    }
  }
  // slab [split][Co][R][S][Ci]: accumulator (c, i, j) = (co, ci) pair of this item's tap
  if (active) {
    float* outp = p.part + (size_t)blockIdx.x * p.Co * R * R * p.Ci;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= nchunks) break;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = WIDE_DY ? c * 16 + g * 4 + i : i;
          const int ci = WIDE_DY ? j : c * 16 + g * 4 + j;
          if (co < p.Co && ci < p.Ci) outp[((size_t)co * R * R + tap) * p.Ci + ci] = acc[c][i * 4 + j];
        }
    }
  }
}

static bool wgrad_thin_ok(const buctd_conv_desc* d) {
  const int thin = d->Ci < d->Co ? d->Ci : d->Co, wide = d->Ci < d->Co ? d->Co : d->Ci;
  // 3x3 with the thin side on x (3 -> 64): the implicit-GEMM kernel is faster there (0.8 against 1.2 ms at 384x288)
  if (d->R == 3 && d->Co > d->Ci) return false;
  return d->stride == 1 && d->R == d->S && (d->R == 3 || d->R == 7) && d->pad == d->R / 2 && thin <= 4 && wide <= 64 &&
         d->Ho == d->H && d->Wo == d->W && (long)d->N * d->H * d->W >= 4096;
}

// 64 -> 3 7x7 ('same'): the weight gradient with a ROLLING window of input rows.  The tile kernel above reads one LDS vector
// of x for every 12 useful FMAs (a thread owns ONE tap: every x value it reads meets three dY channels) and is paced by the
// LDS at 38 TFLOP/s.  Here a thread owns a filter ROW r and a channel pair: the x value of tile column c serves the seven taps
// (r, s) of that row at once - 42 FMAs (21 v_pk_fma_f32) per 8-byte LDS read - against the dY row of the strip, held in
// registers.  A workgroup walks a 32-column strip of one image top to bottom; the R + 1 input rows it needs live in an LDS ring,
// one new row per output row (the next one is staged while the current one is multiplied).  7 x 32 = 224 of 256 threads
// multiply.  One slab [3][7][7][64] per workgroup, reduced by splitk_reduce_kernel.
constexpr int TR_W = 32, TR_RING = 8, TR_XW = TR_W + 6;
struct ThinRowsArgs {
  const float* x;      // [N][H][W][64]
  const float* dy;     // [N][H][W][Co], Co <= 3
  float* part;
  int N, H, W, Co, strips, segs, rows_per_seg;
};
__global__ __launch_bounds__(256) void conv_wgrad_thin_rows_kernel(ThinRowsArgs p) {
  constexpr int R = 7, CI = 64;
  extern __shared__ __attribute__((aligned(16))) float tr_smem[];
  float* xring = tr_smem;                                   // [TR_RING][TR_XW][64]
  float* dyrow = tr_smem + TR_RING * TR_XW * CI;            // [2][TR_W][4]
  const int t = threadIdx.x;
  int id = blockIdx.x;
  const int seg = id % p.segs; id /= p.segs;
  const int strip = id % p.strips, n = id / p.strips;
  const int x0 = strip * TR_W;
  const int ys = seg * p.rows_per_seg;
  int ye = ys + p.rows_per_seg;
  if (ye > p.H) ye = p.H;
  const int r = t >> 5, cg = t & 31;                        // filter row, channel pair (t < 224)
  thin_f32x2 acc[R][3];
#pragma unroll
  for (int sx = 0; sx < R; ++sx)
#pragma unroll
    for (int co = 0; co < 3; ++co) acc[sx][co] = (thin_f32x2){0.f, 0.f};
  // input row `row` of the image (any integer: rows outside are zeros) -> ring slot (row - (ys - 3)) % TR_RING
  auto stage_x = [&](int row) {
    float* dst = xring + ((row - (ys - 3)) & (TR_RING - 1)) * (TR_XW * CI);
    const bool rok = row >= 0 && row < p.H;
    for (int i = t; i < TR_XW * (CI / 4); i += 256) {
      const int c = i >> 4, q = i & 15;
      const int xx = x0 + c - 3;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rok && xx >= 0 && xx < p.W) v = *reinterpret_cast<const f32x4*>(p.x + ((long)(n * p.H + row) * p.W + xx) * CI + q * 4);
      *reinterpret_cast<f32x4*>(dst + c * CI + q * 4) = v;
    }
  };
  auto stage_dy = [&](int row, int buf) {
    if (t < TR_W) {
      const int xx = x0 + t;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row < ye && xx < p.W) {
        const float* src = p.dy + ((long)(n * p.H + row) * p.W + xx) * p.Co;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (e < p.Co) v[e] = src[e];
      }
      *reinterpret_cast<f32x4*>(dyrow + (buf * TR_W + t) * 4) = v;
    }
  };
  if (ys < ye) {
    for (int row = ys - 3; row <= ys + 3; ++row) stage_x(row);
    stage_dy(ys, 0);
  }
  __syncthreads();
  for (int py = ys; py < ye; ++py) {
    // the row the NEXT output row adds (its ring slot held row py - 4: nobody reads it now) and the next dY row travel through
    // registers: loaded in front of the multiplication, stored behind it
    const bool more = py + 1 < ye;
    constexpr int NPC = (TR_XW * (CI / 4) + 255) / 256;
    f32x4 nx[NPC], nd = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (more) {
      const int row = py + 4;
      const bool rok = row >= 0 && row < p.H;
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int i = t + 256 * u, c = i >> 4, q = i & 15;
        const int xx = x0 + c - 3;
        nx[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < TR_XW * (CI / 4) && rok && xx >= 0 && xx < p.W)
          nx[u] = *reinterpret_cast<const f32x4*>(p.x + ((long)(n * p.H + row) * p.W + xx) * CI + q * 4);
      }
      if (t < TR_W && x0 + t < p.W) {
        const float* src = p.dy + ((long)(n * p.H + py + 1) * p.W + x0 + t) * p.Co;
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (e < p.Co) nd[e] = src[e];
      }
    }
    if (t < R * 32) {
      const float* xr = xring + ((py + r - 3 - (ys - 3)) & (TR_RING - 1)) * (TR_XW * CI) + 2 * cg;
      const float* dr = dyrow + ((py - ys) & 1) * TR_W * 4;
      f32x4 dv[TR_W];
#pragma unroll
      for (int px = 0; px < TR_W; ++px) dv[px] = *reinterpret_cast<const f32x4*>(dr + px * 4);
#pragma unroll
      for (int c = 0; c < TR_XW; ++c) {
        const thin_f32x2 xv = *reinterpret_cast<const thin_f32x2*>(xr + c * CI);
#pragma unroll
        for (int sx = 0; sx < R; ++sx) {
          const int px = c - sx;                            // output pixel whose tap (r, sx) reads tile column c
          if (px >= 0 && px < TR_W) {
#pragma unroll
            for (int co = 0; co < 3; ++co) {
              const float d = dv[px][co];
              acc[sx][co] = __builtin_elementwise_fma(xv, (thin_f32x2){d, d}, acc[sx][co]);
            }
          }
        }
      }
    }
    if (more) {
      float* dst = xring + ((py + 4 - (ys - 3)) & (TR_RING - 1)) * (TR_XW * CI);
#pragma unroll
      for (int u = 0; u < NPC; ++u) {
        const int i = t + 256 * u;
        if (i < TR_XW * (CI / 4)) *reinterpret_cast<f32x4*>(dst + (i >> 4) * CI + (i & 15) * 4) = nx[u];
      }
      if (t < TR_W) *reinterpret_cast<f32x4*>(dyrow + (((py + 1 - ys) & 1) * TR_W + t) * 4) = nd;
    }
    __syncthreads();
  }
  if (t < R * 32) {
    float* outp = p.part + (size_t)blockIdx.x * p.Co * R * R * CI;
#pragma unroll
    for (int sx = 0; sx < R; ++sx)
#pragma unroll
      for (int co = 0; co < 3; ++co)
        if (co < p.Co) {
          float* o = outp + ((size_t)(co * R + r) * R + sx) * CI + 2 * cg;
          o[0] = acc[sx][co].x;
          o[1] = acc[sx][co].y;
        }
  }
}
static bool wgrad_thin_rows_ok(const buctd_conv_desc* d) {
  return d->stride == 1 && d->R == 7 && d->S == 7 && d->pad == 3 && d->Ci == 64 && d->Co >= 1 && d->Co <= 3 && d->Ho == d->H &&
         d->Wo == d->W && (long)d->N * d->H * d->W >= 4096;
}
static void thin_rows_geo(const buctd_conv_desc* d, int* strips, int* segs, int* rps) {
  *strips = ceil_div(d->W, TR_W);
  int sg = ceil_div(768, d->N * *strips);                // about three workgroups per CU ...
  const int maxsg = d->H / 16 > 0 ? d->H / 16 : 1;       // ... of at least 16 rows (each re-stages six halo rows)
  if (sg > maxsg) sg = maxsg;
  if (sg < 1) sg = 1;
  *rps = ceil_div(d->H, sg);
  *segs = ceil_div(d->H, *rps);
}
static int thin_rows_wgs(const buctd_conv_desc* d) {
  int st, sg, rps;
  thin_rows_geo(d, &st, &sg, &rps);
  return d->N * st * sg;
}

static int thin_splits(const buctd_conv_desc* d) {      // one slab per workgroup; small inputs get fewer
  const long tiles = (long)d->N * ceil_div(d->H, THIN_TH) * ceil_div(d->W, THIN_TW);
  return (int)(tiles < THIN_SPLITS ? tiles : THIN_SPLITS);
}

// ---- wgrad ------------------------------------------------------------------
static void wgrad_plan(const buctd_conv_desc* d, int* bm, int* bn, int* nsplit, int* pps) {
  const int co = d->Co;
  *bm = (co <= 48) ? 48 : (co % 96 == 0 ? 96 : (co <= 64 ? 64 : 128));
  *bn = 64;
  const int ncols = d->R * d->S * d->Ci;
  const long tiles = (long)ceil_div(co, *bm) * ceil_div(ncols, *bn);
  const long Mpix = (long)d->N * d->Ho * d->Wo;
  long want = (768 + tiles - 1) / tiles;   // ~3 workgroups per CU in total
  long maxsplit = (Mpix + 255) / 256;      // at least 256 pixels per split
  if (want > maxsplit) want = maxsplit;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  long per = (Mpix + want - 1) / want;
  per = ((per + GK - 1) / GK) * GK;
  *pps = (int)per;
  *nsplit = (int)((Mpix + per - 1) / per);
}

extern "C" size_t buctd_conv2d_wgrad_workspace(const buctd_conv_desc* d) {
  if (check_desc(d, "buctd_conv2d_wgrad_workspace")) return 0;
  if (wgrad_thin_rows_ok(d)) return (size_t)thin_rows_wgs(d) * d->Co * d->R * d->S * d->Ci * sizeof(float);
  if (wgrad_thin_ok(d)) return (size_t)thin_splits(d) * d->Co * d->R * d->S * d->Ci * sizeof(float);
  if (fwd_thin_in_ok(d)) return (size_t)thin_in_wgs(d) * d->Co * d->R * d->S * d->Ci * sizeof(float);
  int bm, bn, ns, pps;
  wgrad_plan(d, &bm, &bn, &ns, &pps);
  return (size_t)ns * d->Co * d->R * d->S * d->Ci * sizeof(float);
}

template <class T, bool VEC>
static void launch_wgrad(const WgradArgs& a, int nsplit, hipStream_t st) {
  dim3 grid(ceil_div(a.Co, T::BM), ceil_div(a.Ncols, T::BN), nsplit);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, VEC>), grid, dim3(256), 0, st, a);
}

extern "C" int buctd_conv2d_wgrad(const buctd_conv_desc* d, const float* x, const float* dy, float* dw,
                                  int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_desc(d, "buctd_conv2d_wgrad");
  if (rc) return rc;
  BUCTD_CHECK_ARG(x && dy && dw, "buctd_conv2d_wgrad: null tensor pointer");
  int bm, bn, ns, pps;
  wgrad_plan(d, &bm, &bn, &ns, &pps);
  const bool thin = wgrad_thin_ok(d);
  const bool thin_rows = thin && wgrad_thin_rows_ok(d);  // 64 -> 3 7x7: rolling rows
  const bool thin_in = !thin && fwd_thin_in_ok(d);       // 3 -> 64 3x3: lane = output channel
  if (thin) ns = thin_splits(d);
  if (thin_rows) ns = thin_rows_wgs(d);
  if (thin_in) ns = thin_in_wgs(d);
  const size_t need = (size_t)ns * d->Co * d->R * d->S * d->Ci * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) {
    buctd_set_error("buctd_conv2d_wgrad: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  if (thin_rows) {
    ThinRowsArgs ta;
    ta.x = x; ta.dy = dy; ta.part = (float*)workspace;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Co = d->Co;
    thin_rows_geo(d, &ta.strips, &ta.segs, &ta.rows_per_seg);
    const size_t lds = ((size_t)TR_RING * TR_XW * 64 + 2 * TR_W * 4) * sizeof(float);
    static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
    if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(conv_wgrad_thin_rows_kernel), (int)lds, attr_done,
                                             "buctd_conv2d_wgrad(thin rows)"))
      return rc;
    hipLaunchKernelGGL(conv_wgrad_thin_rows_kernel, dim3(ns), dim3(256), lds, (hipStream_t)stream, ta);
    BUCTD_CHECK_LAUNCH("buctd_conv2d_wgrad(thin rows)");
  } else
  if (thin) {
    ThinArgs ta;
    ta.x = x; ta.dy = dy; ta.part = (float*)workspace;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Co = d->Co;
    ta.tiles_y = ceil_div(d->H, THIN_TH); ta.tiles_x = ceil_div(d->W, THIN_TW);
    ta.ntiles = d->N * ta.tiles_y * ta.tiles_x;
    hipStream_t tst = (hipStream_t)stream;
    const bool wide_dy = d->Co > d->Ci;
    if (d->R == 7) {
      if (wide_dy) hipLaunchKernelGGL((conv_wgrad_thin_kernel<7, true>), dim3(ns), dim3(256), 0, tst, ta);
      else hipLaunchKernelGGL((conv_wgrad_thin_kernel<7, false>), dim3(ns), dim3(256), 0, tst, ta);
    } else {
      if (wide_dy) hipLaunchKernelGGL((conv_wgrad_thin_kernel<3, true>), dim3(ns), dim3(256), 0, tst, ta);
      else hipLaunchKernelGGL((conv_wgrad_thin_kernel<3, false>), dim3(ns), dim3(256), 0, tst, ta);
    }
    BUCTD_CHECK_LAUNCH("buctd_conv2d_wgrad(thin)");
  }
  WgradArgs a;
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
  a.R = d->R; a.S = d->S; a.stride = d->stride; a.pad = d->pad;
  a.Mpix = d->N * d->Ho * d->Wo; a.Ncols = d->R * d->S * d->Ci; a.pix_per_split = pps;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (d->Ci % 4 == 0) && (d->Co % 4 == 0);
  if (thin_in) {
    ThinInWgradArgs ta;
    ta.x = x; ta.dy = dy; ta.part = (float*)workspace;
    ta.N = d->N; ta.H = d->H; ta.W = d->W; ta.Ci = d->Ci; ta.Ho = d->Ho; ta.Wo = d->Wo;
    const dim3 grid(thin_in_wgs(d));
    const size_t lds = thin_in_wgrad_lds(d);
#define THIN_IN_WG(ci) \
    if (d->Ci == ci) { \
      if (d->stride == 1) hipLaunchKernelGGL((conv3x3_thin_in_wgrad_kernel<ci, 1>), grid, dim3(256), lds, st, ta); \
      else hipLaunchKernelGGL((conv3x3_thin_in_wgrad_kernel<ci, 2>), grid, dim3(256), lds, st, ta); \
    }
    THIN_IN_WG(1) THIN_IN_WG(2) THIN_IN_WG(3) THIN_IN_WG(4)
#undef THIN_IN_WG
  } else
  if (thin) {}                                                             // launched above
  else if (!vec) launch_wgrad<TileCfg<2, 2, 2, 2>, false>(a, ns, st);     // 64x64 generic
  else if (bm == 48) launch_wgrad<TileCfg<1, 4, 3, 1>, true>(a, ns, st);  // 48x64
  else if (bm == 96) launch_wgrad<TileCfg<2, 2, 3, 2>, true>(a, ns, st);  // 96x64
  else if (bm == 64) launch_wgrad<TileCfg<2, 2, 2, 2>, true>(a, ns, st);  // 64x64
  else launch_wgrad<TileCfg<2, 2, 4, 2>, true>(a, ns, st);                // 128x64
  BUCTD_CHECK_LAUNCH("buctd_conv2d_wgrad");
  const long n = (long)d->Co * a.Ncols;
  if (n % 4 == 0) {
    int blocks = ceil_div(n / 4, 32);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, ns,
                       accumulate, 1.0f);
  } else {
    int blocks = ceil_div(n, 32);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, ns,
                       accumulate, 1.0f);
  }
  BUCTD_CHECK_LAUNCH("buctd_conv2d_wgrad(reduce)");
  return BUCTD_OK;
}
