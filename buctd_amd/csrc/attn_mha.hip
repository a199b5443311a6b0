// Fused single-head self-attention forward for the TransPose encoder (reference lib/models/transpose_h.py:192-197:
// nn.MultiheadAttention(d_model = 96 + 16, 1 head) on T = 3072 tokens at 256x192; eval mode = BASELINE config C5's
// inference passes).  out = softmax(scale * Q K^T) V without the T x T matrix ever reaching HBM (the materialised path
// writes and reads it three times: 37.7 MB per image and layer at T = 3072).
//
// Flash-style, exact fp32: both products on v_mfma_f32_16x16x4_f32 (bitwise an fp32 FMA chain), online soft-max in fp32.
//   * a workgroup = 8 wavefronts = 128 query rows (16 per wavefront); keys/values stream through LDS 64 at a time, the
//     next block travelling through registers while the current one is multiplied;
//   * K is staged row-major [64][d + 8], V transposed [d][64 + 8], the wavefront's probabilities [16][64 + 8]: with
//     these strides (= 8 mod 16 floats) every ds_read_b128 of an MFMA operand is bank-conflict free, and the
//     contraction index is permuted (step kk of lane group g = element 16*(kk/4) + 4*g + kk%4) so that ONE 16-byte read
//     feeds four MFMA steps;
//   * Q lives in registers for the whole kernel (d / 4 floats per lane), O in d / 16 accumulators.
// HBM bytes per image and layer: Q, K, V read once per 128-query tile from L2 (K, V: 2.75 MB stay L2-resident), O written
// once: the kernel is bound by the fp32 MFMA rate (4 T^2 d FLOP at 157 TFLOP/s peak).
#include "common.h"
#include "../../include/buctd_hip.h"

#define MHA_BQ 128
#define MHA_BK 64
#define MHA_VPM 2      // soft-max VALU instructions slotted behind each MFMA of the next tile's logits (mha_fwd_x6q_kernel)

template <int DF>   // d / 16
__global__ __launch_bounds__(512, 1) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int T, int ldqk, int ldv,
                                                         float scale, float* __restrict__ out, float* __restrict__ lse) {
  constexpr int D = DF * 16, LDK = D + 8, LDV = MHA_BK + 8, LDP = MHA_BK + 8;
  constexpr int C4 = D / 4;                                  // float4 per token row
  constexpr int PL = (MHA_BK * C4 + 511) / 512;              // float4 per thread for one K (or V) block
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                                            // [64][LDK]
  float* Vt = Ks + MHA_BK * LDK;                             // [D][LDV]
  float* Ps = Vt + D * LDV;                                  // [8 waves][16][LDP]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * MHA_BQ + wave * 16;
  const float* qb = q + ((long)b * T) * ldqk;
  const float* kb = k + ((long)b * T) * ldqk;
  const float* vb = v + ((long)b * T) * ldv;
  float* Pw = Ps + wave * 16 * LDP;

  // Q fragment: lane (i16, g) holds Q[q0 + i16][16 m + 4 g + e], m < DF, e < 4  (pre-scaled)
  f32x4 qf[DF];
#pragma unroll
  for (int m = 0; m < DF; ++m) {
    qf[m] = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + i16) * ldqk + 16 * m + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) qf[m][e] *= scale;
  }
  f32x4 o[DF];
#pragma unroll
  for (int n = 0; n < DF; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrow[4], lrow[4];                                    // running max / sum of rows g*4 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

  f32x4 kreg[PL], vreg[PL];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const int id = t + 512 * p;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MHA_BK) {
        kreg[p] = *reinterpret_cast<const f32x4*>(kb + (long)(k0 + row) * ldqk + c4);
        vreg[p] = *reinterpret_cast<const f32x4*>(vb + (long)(k0 + row) * ldv + c4);
      }
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const int id = t + 512 * p;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MHA_BK) {
        *reinterpret_cast<f32x4*>(Ks + row * LDK + c4) = kreg[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) Vt[(c4 + e) * LDV + row] = vreg[p][e];
      }
    }
  };

  load_kv(0);
  for (int k0 = 0; k0 < T; k0 += MHA_BK) {
    __syncthreads();                      // everyone is done with the previous block
    store_kv();
    if (k0 + MHA_BK < T) load_kv(k0 + MHA_BK);
    __syncthreads();
    // S = (scale Q) K^T : 16 queries x 64 keys per wavefront, 4 key fragments
    f32x4 s[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      s[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* kp = Ks + (nb * 16 + i16) * LDK + 4 * g;
#pragma unroll
      for (int m = 0; m < DF; ++m) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + 16 * m);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[m][e], kv[e], s[nb], 0, 0, 0);
      }
    }
    // online soft-max: lane holds keys nb*16 + i16 of rows g*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float mnew = fmaxf(mrow[r], mx);
      const float corr = __expf(mrow[r] - mnew);     // 0 for the first block (exp(-inf))
      float ps = 0.f;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float pv = __expf(s[nb][r] - mnew);
        s[nb][r] = pv;
        ps += pv;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) ps += __shfl_xor(ps, off, 64);
      lrow[r] = lrow[r] * corr + ps;
      mrow[r] = mnew;
#pragma unroll
      for (int n = 0; n < DF; ++n) o[n][r] *= corr;
    }
    // P (C layout: row g*4 + r, key nb*16 + i16) -> LDS -> A layout (row i16, keys 16 m + 4 g + e)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Pw[(g * 4 + r) * LDP + nb * 16 + i16] = s[nb][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wavefront reads back its own tile, no barrier needed
    f32x4 pa[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pa[m] = *reinterpret_cast<const f32x4*>(Pw + i16 * LDP + 16 * m + 4 * g);
    // O += P V : B operand V^T[dcol = 16 n + i16][key 16 m + 4 g + e]
#pragma unroll
    for (int n = 0; n < DF; ++n) {
      const float* vp = Vt + (n * 16 + i16) * LDV + 4 * g;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + 16 * m);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[m][e], vv[e], o[n], 0, 0, 0);
      }
    }
  }
  // O / l : lane holds rows g*4 + r, column 16 n + i16
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv = 1.f / lrow[r];
    float* op = out + ((long)b * T + q0 + g * 4 + r) * (long)D;
#pragma unroll
    for (int n = 0; n < DF; ++n) op[16 * n + i16] = o[n][r] * inv;
    if (lse && i16 == 0) lse[(long)b * T + q0 + g * 4 + r] = mrow[r] + __logf(lrow[r]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same attention in the bf16x6 arithmetic of the convolutions (operands split exactly into three bf16 pieces, six
// v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate - fp32 class, conv3x3.hip): 417 TFLOP/s-equivalent of matrix
// peak instead of 157.
//   * S^T = K (scale Q)^T, not S: the accumulator of a 16-key block then gives lane (query i16, group g) the logits of
//     keys 4g..4g+3 of ITS query, so two blocks are - with the reduction slot e of group g enumerated as key
//     16 (e >> 2) + 4 g + (e & 3) - exactly the A operand of one K = 32 MFMA of O += P V: the probabilities never leave
//     the registers (the fp32 kernel sends them through LDS to change layout);
//   * K and V tiles are staged key-major as [64][d h | d m | d l] (row stride 6 d bytes = 32 mod 64: conflict-free
//     16-byte row reads for the A operand of S^T; the V fragments in the enumeration above come out of the same layout
//     through two ds_read_b64_tr_b16 transpose reads each);
//   * Q is split once into registers (B operand of S^T), pre-multiplied by scale * log2(e): soft-max in the exp2 domain.
typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mbf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short mu16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ mbf16x8 mha_tr_pair(const unsigned char* p, const unsigned char* q) {
  typedef __attribute__((address_space(3))) mbf16x4 lds_bf16x4;
  const mbf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const mbf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void mha_split8(const float (&x)[8], mbf16x8 (&pc)[3]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float r = x[e];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16 h = (__bf16)r;
      pc[q][e] = h;
      r -= (float)h;
    }
  }
}
__device__ __forceinline__ void mha_mma6(f32x4& acc, const mbf16x8 (&a)[3], const mbf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}

template <int DF>
__global__ __launch_bounds__(512, 1) void mha_fwd_x6_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, int T, int ldqk, int ldv,
                                                            float scale, float* __restrict__ out,
                                                            float* __restrict__ lse) {
  constexpr int D = DF * 16, LO = D * 2, RS = D * 6 + (((D * 6) % 64 == 32) ? 0 : 32), NK = (D + 31) / 32;
  constexpr int C4 = D / 4, PL = (MHA_BK * C4 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  unsigned char* kt = smx;                    // [64 keys][RS]
  unsigned char* vt = smx + MHA_BK * RS;      // [64 keys][RS]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * MHA_BQ + wave * 16;
  const float* qb = q + ((long)b * T) * ldqk;
  const float* kb = k + ((long)b * T) * ldqk;
  const float* vb = v + ((long)b * T) * ldv;

  // B operand of S^T: lane (query i16, group g) holds Q[q0 + i16][32 kk + 8 g .. +8] (zeros past d), pre-scaled
  mbf16x8 qq[NK][3];
  const float s2 = scale * 1.44269504088896340736f;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * g;
    float x[8];
    if (c0 < D) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + i16) * ldqk + c0);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + i16) * ldqk + c0 + 4);
      x[0] = a.x * s2; x[1] = a.y * s2; x[2] = a.z * s2; x[3] = a.w * s2;
      x[4] = c.x * s2; x[5] = c.y * s2; x[6] = c.z * s2; x[7] = c.w * s2;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
    mha_split8(x, qq[kk]);
  }
  f32x4 o[DF];
#pragma unroll
  for (int n = 0; n < DF; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;      // query i16: running maximum (same in its 4 lanes), this lane's share of the sum
  int koff[NK];                            // byte offset of this lane's 8 channels in a K tile row (clamped in the pad)
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * g;
    koff[kk] = (c0 < D ? c0 : 0) * 2;
  }
  const int tr_off = (4 * g + (i16 >> 2)) * RS + (i16 & 3) * 8;   // keys 4g .. 4g+3 of a 16-key block, this lane's column

  f32x4 kreg[PL], vreg[PL];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const int id = t + 512 * p;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MHA_BK) {
        kreg[p] = *reinterpret_cast<const f32x4*>(kb + (long)(k0 + row) * ldqk + c4);
        vreg[p] = *reinterpret_cast<const f32x4*>(vb + (long)(k0 + row) * ldv + c4);
      }
    }
  };
  auto split_row4 = [&](unsigned char* row, int c, f32x4 val) {
    mu16x4 pc[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r = val[j];
#pragma unroll
      for (int qp = 0; qp < 3; ++qp) {
        const __bf16 h = (__bf16)r;
        pc[qp][j] = __builtin_bit_cast(unsigned short, h);
        r -= (float)h;
      }
    }
#pragma unroll
    for (int qp = 0; qp < 3; ++qp) *reinterpret_cast<mu16x4*>(row + qp * LO + 2 * c) = pc[qp];
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const int id = t + 512 * p;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MHA_BK) {
        split_row4(kt + row * RS, c4, kreg[p]);
        split_row4(vt + row * RS, c4, vreg[p]);
      }
    }
  };

  load_kv(0);
  for (int k0 = 0; k0 < T; k0 += MHA_BK) {
    __syncthreads();                      // everyone is done with the previous block
    store_kv();
    if (k0 + MHA_BK < T) load_kv(k0 + MHA_BK);
    __syncthreads();
    // S^T blocks: st[f][rg] = logit of (key 16 f + 4 g + rg, query i16), exp2 domain
    f32x4 st[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      st[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* krow = kt + (16 * f + i16) * RS;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        mbf16x8 aa[3];
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
          aa[qp] = *reinterpret_cast<const mbf16x8*>(krow + koff[kk] + qp * LO);
          if (32 * kk + 8 * g >= D) {            // zero-padded tail of the channel contraction
#pragma unroll
            for (int e = 0; e < 8; ++e) aa[qp][e] = (__bf16)0.f;
          }
        }
        mha_mma6(st[f], aa, qq[kk]);
      }
    }
    // online soft-max of query i16 (its 64 logits of this tile sit in the four lanes i16 + 16 g)
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, st[f][rg]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(mrun, mx);
    if (__any(mnew > mrun)) {
      const float fac = __builtin_amdgcn_exp2f(mrun - mnew);   // 1 where the maximum stayed, 0 on the first tile
      lrun *= fac;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float fr = __shfl(fac, g * 4 + rg, 64);          // accumulator rows of O are queries g*4 + rg
#pragma unroll
        for (int n = 0; n < DF; ++n) o[n][rg] *= fr;
      }
      mrun = mnew;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {            // 32 keys per MFMA: blocks 2h, 2h + 1
      float pv[8];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float e2 = __builtin_amdgcn_exp2f(st[2 * h + fb][rg] - mrun);
          pv[4 * fb + rg] = e2;
          lrun += e2;
        }
      mbf16x8 pp[3];
      mha_split8(pv, pp);
      const unsigned char* vbase = vt + 32 * h * RS + tr_off;
#pragma unroll
      for (int n = 0; n < DF; ++n) {
        mbf16x8 bb[3];
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) bb[qp] = mha_tr_pair(vbase + n * 32 + qp * LO, vbase + n * 32 + qp * LO + 16 * RS);
        mha_mma6(o[n], pp, bb);
      }
    }
  }
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
  const float linv = 1.f / lrun;
  if (lse && g == 0) lse[(long)b * T + q0 + i16] = (mrun + __builtin_amdgcn_logf(lrun)) * 0.69314718055994530942f;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const float fl = __shfl(linv, g * 4 + rg, 64);
    float* op = out + ((long)b * T + q0 + g * 4 + rg) * (long)D;
#pragma unroll
    for (int n = 0; n < DF; ++n) op[16 * n + i16] = o[n][rg] * fl;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16x6 attention over PRE-SPLIT keys and values.  mha_fwd_x6_kernel splits every 64-key tile in each of the T / 128
// workgroups of an image (24 times at T = 3072) and its two barriers per tile line the eight wavefronts up in a "split
// (VALU), multiply (MFMA)" rhythm in which the two pipes take turns.  Here mha_kv_split_kernel writes the tiles once per
// layer in exactly the LDS layout ([key][d h | d m | d l | pad], RS bytes per key, 6 B per element) and the attention kernel
// moves them HBM/L2 -> LDS with the DMA path (global_load_lds_dwordx4: no registers, no VALU, no ds_write):
//   * K(j + 1) travels while O += P V(j) runs, V(j + 1) while S^T = K(j + 1) Q^T runs - one buffer each, two barriers per
//     tile as before, but nothing is waited for;
//   * workgroups of one image are dealt to ONE XCD (blockIdx -> (image, query tile) below), so its K/V image (4.3 MB at
//     T = 3072, d = 112) is fetched into that XCD's L2 once instead of once per XCD.
__global__ __launch_bounds__(256) void mha_kv_split_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                           long rows, int d, int ldk, int ldv, int rs,
                                                           unsigned char* __restrict__ k6, unsigned char* __restrict__ v6) {
  const int c4n = d >> 2;
  const float* src = blockIdx.y ? v : k;
  const int ld = blockIdx.y ? ldv : ldk;
  unsigned char* dst = blockIdx.y ? v6 : k6;
  const long items = rows * c4n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < items; idx += (long)gridDim.x * 256) {
    const long row = idx / c4n;
    const int c = (int)(idx - row * c4n) * 4;
    const f32x4 val = *reinterpret_cast<const f32x4*>(src + row * ld + c);
    mu16x4 pc[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r = val[j];
#pragma unroll
      for (int qp = 0; qp < 3; ++qp) {
        const __bf16 h = (__bf16)r;
        pc[qp][j] = __builtin_bit_cast(unsigned short, h);
        r -= (float)h;
      }
    }
    unsigned char* o = dst + row * rs + 2 * c;
#pragma unroll
    for (int qp = 0; qp < 3; ++qp) *reinterpret_cast<mu16x4*>(o + qp * 2 * d) = pc[qp];
  }
}

// one wave-instruction of the tile copy: 1 KB, LDS destination = M0 + lane * 16 (inline assembly for the reason given at
// w4_dma in conv3x3_wgrad4.hip: with the builtin the compiler drains vmcnt in front of the first LDS read)
__device__ __forceinline__ void mha_dma(const unsigned char* src_lane, unsigned char* lds_kb) {
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds_kb;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_addr) : "memory");
}

template <int DF>
__global__ __launch_bounds__(512, 1) void mha_fwd_x6p_kernel(const float* __restrict__ q,
                                                             const unsigned char* __restrict__ k6,
                                                             const unsigned char* __restrict__ v6, int B, int T, int ldq,
                                                             float scale, float* __restrict__ out,
                                                             float* __restrict__ lse) {
  constexpr int D = DF * 16, LO = D * 2, RS = D * 6 + (((D * 6) % 64 == 32) ? 0 : 32), NK = (D + 31) / 32;
  constexpr int TILE = MHA_BK * RS, NINS = TILE / 1024, NI = (NINS + 7) / 8;
  static_assert(TILE % 1024 == 0, "a 64-key tile is a whole number of 1 KB wave copies");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  unsigned char* kt = smx;                    // [64 keys][RS]
  unsigned char* vt = smx + TILE;             // [64 keys][RS]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  // workgroup -> (image, query tile): consecutive workgroup ids go round the 8 XCDs; give XCD x the images = x mod 8
  const int nq = T / MHA_BQ;
  int b, qt;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot / nq;
    b = grp * 8 + xcd;
    qt = slot - grp * nq;
  } else {
    b = blockIdx.x / nq;
    qt = blockIdx.x - b * nq;
  }
  const int q0 = qt * MHA_BQ + wave * 16;
  const float* qb = q + ((long)b * T) * ldq;
  const unsigned char* kimg = k6 + (long)b * T * RS + lane * 16;
  const unsigned char* vimg = v6 + (long)b * T * RS + lane * 16;
  auto dma_tile = [&](const unsigned char* img, int k0, unsigned char* dst) {
    const unsigned char* src = img + (long)k0 * RS;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = wave + 8 * i;
      if (u < NINS) mha_dma(src + u * 1024, dst + u * 1024);
    }
  };
  dma_tile(kimg, 0, kt);
  dma_tile(vimg, 0, vt);

  mbf16x8 qq[NK][3];
  const float s2 = scale * 1.44269504088896340736f;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * g;
    float x[8];
    if (c0 < D) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + i16) * ldq + c0);
      const f32x4 c = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + i16) * ldq + c0 + 4);
      x[0] = a.x * s2; x[1] = a.y * s2; x[2] = a.z * s2; x[3] = a.w * s2;
      x[4] = c.x * s2; x[5] = c.y * s2; x[6] = c.z * s2; x[7] = c.w * s2;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = 0.f;
    }
    mha_split8(x, qq[kk]);
  }
  f32x4 o[DF];
#pragma unroll
  for (int n = 0; n < DF; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;
  int koff[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * g;
    koff[kk] = (c0 < D ? c0 : 0) * 2;
  }
  const int tr_off = (4 * g + (i16 >> 2)) * RS + (i16 & 3) * 8;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k0 = 0; k0 < T; k0 += MHA_BK) {
    f32x4 st[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      st[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* krow = kt + (16 * f + i16) * RS;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        mbf16x8 aa[3];
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
          aa[qp] = *reinterpret_cast<const mbf16x8*>(krow + koff[kk] + qp * LO);
          if (32 * kk + 8 * g >= D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) aa[qp][e] = (__bf16)0.f;
          }
        }
        mha_mma6(st[f], aa, qq[kk]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // V(j) (sent during the previous P V phase) has landed
    __syncthreads();                                       // ... for everyone, and nobody reads K(j) any more
    if (k0 + MHA_BK < T) dma_tile(kimg, k0 + MHA_BK, kt);
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, st[f][rg]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(mrun, mx);
    if (__any(mnew > mrun)) {
      const float fac = __builtin_amdgcn_exp2f(mrun - mnew);
      lrun *= fac;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float fr = __shfl(fac, g * 4 + rg, 64);
#pragma unroll
        for (int n = 0; n < DF; ++n) o[n][rg] *= fr;
      }
      mrun = mnew;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float pv[8];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float e2 = __builtin_amdgcn_exp2f(st[2 * h + fb][rg] - mrun);
          pv[4 * fb + rg] = e2;
          lrun += e2;
        }
      mbf16x8 pp[3];
      mha_split8(pv, pp);
      const unsigned char* vbase = vt + 32 * h * RS + tr_off;
#pragma unroll
      for (int n = 0; n < DF; ++n) {
        mbf16x8 bb[3];
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) bb[qp] = mha_tr_pair(vbase + n * 32 + qp * LO, vbase + n * 32 + qp * LO + 16 * RS);
        mha_mma6(o[n], pp, bb);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(j + 1) has landed
    __syncthreads();                                       // ... for everyone, and nobody reads V(j) any more
    if (k0 + MHA_BK < T) dma_tile(vimg, k0 + MHA_BK, vt);
  }
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
  const float linv = 1.f / lrun;
  if (lse && g == 0) lse[(long)b * T + q0 + i16] = (mrun + __builtin_amdgcn_logf(lrun)) * 0.69314718055994530942f;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const float fl = __shfl(linv, g * 4 + rg, 64);
    float* op = out + ((long)b * T + q0 + g * 4 + rg) * (long)D;
#pragma unroll
    for (int n = 0; n < DF; ++n) op[16 * n + i16] = o[n][rg] * fl;
  }
}

// The same with 32 queries per wavefront.  With 16, every wavefront reads the whole K and V tile from LDS for 180 MFMAs:
// 8 wavefronts x 90 KB per 64 keys = 5760 LDS cycles per CU against 5760 MFMA cycles per SIMD - the LDS port is as busy as
// the matrix pipe and the two do not overlap perfectly (measured 0.45-0.5 of the MFMA roof).  Two query blocks per
// wavefront reuse every K fragment (A operand of S^T) and every V fragment (B operand of P V) twice: half the LDS bytes
// per MFMA.  So that the grid still fills the chip in whole rounds (32 images x 3072 queries / 32 per wavefront = 3072
// wavefronts on 2048 slots would be 1.5 rounds), a workgroup keeps 8 wavefronts and 128 queries and splits the KEYS:
// wavefronts 0-3 attend to the first half of the keys, 4-7 to the second (each half streams its own 32-key K and V tiles,
// 4 x 22 KB of LDS at d = 112), and the two partial soft-max states of a query (o, m, l) are merged through LDS at the end.
template <int DF>
__global__ __launch_bounds__(512, 1) void mha_fwd_x6q_kernel(const float* __restrict__ q,
                                                             const unsigned char* __restrict__ k6,
                                                             const unsigned char* __restrict__ v6, int B, int T, int ldq,
                                                             float scale, float* __restrict__ out,
                                                             float* __restrict__ lse) {
  constexpr int D = DF * 16, LO = D * 2, RS = D * 6 + (((D * 6) % 64 == 32) ? 0 : 32), NK = (D + 31) / 32;
  constexpr int BK = 32, TILE = BK * RS, NINS = TILE / 1024, NI = (NINS + 3) / 4;
  static_assert(TILE % 1024 == 0, "a 32-key tile is a whole number of 1 KB wave copies");
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qw = wave & 3, kh = wave >> 2;
  unsigned char* kt = smx + kh * 2 * TILE;    // this half's [32 keys][RS]
  unsigned char* vt = kt + TILE;
  const int i16 = lane & 15, g = lane >> 4;
  const int nq = T / MHA_BQ;
  int b, qt;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot / nq;
    b = grp * 8 + xcd;
    qt = slot - grp * nq;
  } else {
    b = blockIdx.x / nq;
    qt = blockIdx.x - b * nq;
  }
  const int q0 = qt * MHA_BQ + qw * 32;
  const int TH = T >> 1;                      // keys per half (a multiple of 64)
  const float* qb = q + ((long)b * T) * ldq;
  const unsigned char* kimg = k6 + ((long)b * T + (long)kh * TH) * RS + lane * 16;
  const unsigned char* vimg = v6 + ((long)b * T + (long)kh * TH) * RS + lane * 16;
  auto dma_tile = [&](const unsigned char* img, int k0, unsigned char* dst) {
    const unsigned char* src = img + (long)k0 * RS;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int u = qw + 4 * i;
      if (u < NINS) mha_dma(src + u * 1024, dst + u * 1024);
    }
  };
  dma_tile(kimg, 0, kt);
  dma_tile(vimg, 0, vt);

  mbf16x8 qq[2][NK][3];
  const float s2 = scale * 1.44269504088896340736f;
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const int c0 = 32 * kk + 8 * g;
      float x[8];
      if (c0 < D) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + 16 * w + i16) * ldq + c0);
        const f32x4 c = *reinterpret_cast<const f32x4*>(qb + (long)(q0 + 16 * w + i16) * ldq + c0 + 4);
        x[0] = a.x * s2; x[1] = a.y * s2; x[2] = a.z * s2; x[3] = a.w * s2;
        x[4] = c.x * s2; x[5] = c.y * s2; x[6] = c.z * s2; x[7] = c.w * s2;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      mha_split8(x, qq[w][kk]);
    }
  f32x4 o[2][DF];
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int n = 0; n < DF; ++n) o[w][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
  int koff[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * g;
    koff[kk] = (c0 < D ? c0 : 0) * 2;
  }
  const int tr_off = (4 * g + (i16 >> 2)) * RS + (i16 & 3) * 8;

  // S^T of one 32-key tile for both query blocks (K tile in kt)
  auto qk_tile = [&](f32x4 (&sx)[2][2]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      sx[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      sx[1][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* krow = kt + (16 * f + i16) * RS;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        mbf16x8 aa[3];
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
          aa[qp] = *reinterpret_cast<const mbf16x8*>(krow + koff[kk] + qp * LO);
          if (32 * kk + 8 * g >= D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) aa[qp][e] = (__bf16)0.f;
          }
        }
        mha_mma6(sx[0][f], aa, qq[0][kk]);
        mha_mma6(sx[1][f], aa, qq[1][kk]);
      }
    }
  };
  // Software pipeline over the tiles: the logits of tile j + 1 are multiplied (MFMA) in the same straight-line stretch in
  // which the probabilities of tile j are exponentiated and split (VALU) - the two pipes work side by side instead of
  // taking turns.  kt holds K(j + 1) while vt holds V(j); each is refilled by DMA during the phase that does not read it.
  const int ntile = TH / BK;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x4 st[2][2];
  qk_tile(st);
  __syncthreads();
  if (ntile > 1) dma_tile(kimg, BK, kt);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < ntile; ++j) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) mx = fmaxf(mx, st[w][f][rg]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[w], mx);
      if (__any(mnew > mrun[w])) {
        const float fac = __builtin_amdgcn_exp2f(mrun[w] - mnew);
        lrun[w] *= fac;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float fr = __shfl(fac, g * 4 + rg, 64);
#pragma unroll
          for (int n = 0; n < DF; ++n) o[w][n][rg] *= fr;
        }
        mrun[w] = mnew;
      }
    }
    // K(j + 1) Q^T (on the last tile: a dead product of the stale tile - no branch) with the soft-max of tile j dealt over
    // its 2 NK fragment slots: after the twelve MFMAs of a slot, the exp2 + three-way split of one or two PAIRS of
    // probabilities (a pair = one packed register per piece).  Two wavefronts of a SIMD running this stretch dovetail
    // (one's VALU share under the other's MFMAs) although the barriers keep them in step.
    f32x4 sn[2][2];
    unsigned ppu[2][3][4];
    constexpr int SL = 2 * NK;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int f = i / NK, kk = i - f * NK;
      if (kk == 0) {
        sn[0][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sn[1][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const unsigned char* krow = kt + (16 * f + i16) * RS;
      mbf16x8 aa[3];
#pragma unroll
      for (int qp = 0; qp < 3; ++qp) {
        aa[qp] = *reinterpret_cast<const mbf16x8*>(krow + koff[kk] + qp * LO);
        if (32 * kk + 8 * g >= D) {
#pragma unroll
          for (int e = 0; e < 8; ++e) aa[qp][e] = (__bf16)0.f;
        }
      }
      if (i > 0) {
        // order without a fence: the first K fragment of this slot passes through an empty asm together with the packed
        // probabilities of the previous slot's share - that share is then issued before these MFMAs, after the last ones
        typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));
        mu32x4 a0 = __builtin_bit_cast(mu32x4, aa[0]);
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
          if ((pr * SL) / 8 != i - 1) continue;
          const int w = pr >> 2, ix = 2 * ((pr >> 1) & 1) + (pr & 1);
          asm volatile("" : "+v"(a0), "+v"(ppu[w][0][ix]), "+v"(ppu[w][1][ix]), "+v"(ppu[w][2][ix]), "+v"(lrun[w]));
        }
        aa[0] = __builtin_bit_cast(mbf16x8, a0);
      }
      mha_mma6(sn[0][f], aa, qq[0][kk]);
      mha_mma6(sn[1][f], aa, qq[1][kk]);
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {
        if ((pr * SL) / 8 != i) continue;
        const int w = pr >> 2, fb = (pr >> 1) & 1, r0 = (pr & 1) * 2;
        float ra = __builtin_amdgcn_exp2f(st[w][fb][r0] - mrun[w]);
        float rb = __builtin_amdgcn_exp2f(st[w][fb][r0 + 1] - mrun[w]);
        lrun[w] += ra;
        lrun[w] += rb;
#pragma unroll
        for (int qp = 0; qp < 3; ++qp) {
          const __bf16 ha = (__bf16)ra, hb = (__bf16)rb;
          ppu[w][qp][2 * fb + (pr & 1)] = (unsigned)__builtin_bit_cast(unsigned short, ha) |
                                          ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
          ra -= (float)ha;
          rb -= (float)hb;
        }
      }
    }
    mbf16x8 pp[2][3];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int qp = 0; qp < 3; ++qp) {
        typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));
        mu32x4 r = {ppu[w][qp][0], ppu[w][qp][1], ppu[w][qp][2], ppu[w][qp][3]};
        asm volatile("" : "+v"(r));       // the probabilities stay on this side of the barrier
        pp[w][qp] = __builtin_bit_cast(mbf16x8, r);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // V(j) has landed
    __syncthreads();                                       // ... for everyone, and nobody reads K(j + 1) any more
    if (j + 2 < ntile) dma_tile(kimg, (j + 2) * BK, kt);
    const unsigned char* vbase = vt + tr_off;
#pragma unroll
    for (int n = 0; n < DF; ++n) {
      mbf16x8 bb[3];
#pragma unroll
      for (int qp = 0; qp < 3; ++qp)
        bb[qp] = mha_tr_pair(vbase + n * 32 + qp * LO, vbase + n * 32 + qp * LO + 16 * RS);
      mha_mma6(o[0][n], pp[0], bb);
      mha_mma6(o[1][n], pp[1], bb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // K(j + 2) has landed
    __syncthreads();                                       // ... for everyone, and nobody reads V(j) any more
    if (j + 1 < ntile) dma_tile(vimg, (j + 1) * BK, vt);
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int f = 0; f < 2; ++f) st[w][f] = sn[w][f];
  }
  // merge the two key halves (the tiles are dead after the loop's last barrier): wavefronts 4-7 park (o, m, l) in LDS
  float* mo = reinterpret_cast<float*>(smx) + (size_t)qw * 64 * (2 * DF * 4 + 4);
  float* mine = mo + lane * (2 * DF * 4 + 4);
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    lrun[w] += __shfl_xor(lrun[w], 16, 64);
    lrun[w] += __shfl_xor(lrun[w], 32, 64);
  }
  if (kh == 1) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
#pragma unroll
      for (int n = 0; n < DF; ++n) *reinterpret_cast<f32x4*>(mine + (w * DF + n) * 4) = o[w][n];
      mine[2 * DF * 4 + 2 * w] = mrun[w];
      mine[2 * DF * 4 + 2 * w + 1] = lrun[w];
    }
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const float m1 = mine[2 * DF * 4 + 2 * w], l1 = mine[2 * DF * 4 + 2 * w + 1];
      const float m = fmaxf(mrun[w], m1);
      const float f0 = __builtin_amdgcn_exp2f(mrun[w] - m), f1 = __builtin_amdgcn_exp2f(m1 - m);
      const float lr = lrun[w] * f0 + l1 * f1;
      const float linv = 1.f / lr;
      if (lse && g == 0) lse[(long)b * T + q0 + 16 * w + i16] = (m + __builtin_amdgcn_logf(lr)) * 0.69314718055994530942f;
      const float a0 = f0 * linv, a1 = f1 * linv;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float c0 = __shfl(a0, g * 4 + rg, 64), c1 = __shfl(a1, g * 4 + rg, 64);
        float* op = out + ((long)b * T + q0 + 16 * w + g * 4 + rg) * (long)D;
#pragma unroll
        for (int n = 0; n < DF; ++n) op[16 * n + i16] = o[w][n][rg] * c0 + mine[(w * DF + n) * 4 + rg] * c1;
      }
    }
  }
}

static size_t mha_x6_lds(int d) {
  const int rs = d * 6 + (((d * 6) % 64 == 32) ? 0 : 32);
  return (size_t)2 * MHA_BK * rs;
}

static size_t mha_lds(int d) {
  return ((size_t)MHA_BK * (d + 8) + (size_t)d * (MHA_BK + 8) + (size_t)8 * 16 * (MHA_BK + 8)) * sizeof(float);
}

extern "C" int buctd_mha_fwd_supported(int T, int d) {
  return (T > 0 && T % MHA_BQ == 0 && d >= 16 && d <= 128 && d % 16 == 0) ? 1 : 0;
}

template <int DF>
static int mha_launch(int B, int T, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                      float* out, float* lse, int x6, hipStream_t st) {
  static unsigned char attr_done[2][BUCTD_MAX_DEVICES] = {{0}};
  void (*fn)(const float*, const float*, const float*, int, int, int, float, float*, float*) =
      x6 ? mha_fwd_x6_kernel<DF> : mha_fwd_kernel<DF>;
  const size_t lds = x6 ? mha_x6_lds(DF * 16) : mha_lds(DF * 16);
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[x6 ? 1 : 0], "buctd_mha_fwd")) return rc;
  hipLaunchKernelGGL(fn, dim3(T / MHA_BQ, B), dim3(512), lds, st, q, k, v, T, ldqk, ldv, scale, out, lse);
  BUCTD_CHECK_LAUNCH("buctd_mha_fwd");
  return BUCTD_OK;
}

static int mha_run(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                   float* out, float* lse, int x6, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && B > 0, "buctd_mha_fwd: null pointer");
  BUCTD_CHECK_ARG(buctd_mha_fwd_supported(T, d), "buctd_mha_fwd: unsupported shape T%d d%d (T %% 128 == 0, d %% 16 == 0, d <= 128)",
                  T, d);
  BUCTD_CHECK_ARG(ldqk >= d && ldv >= d && ldqk % 4 == 0 && ldv % 4 == 0, "buctd_mha_fwd: row strides must be >= d and 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (d / 16) {
#define MHA_CASE(n) case n: return mha_launch<n>(B, T, q, k, v, ldqk, ldv, scale, out, lse, x6, st);
    MHA_CASE(1) MHA_CASE(2) MHA_CASE(3) MHA_CASE(4) MHA_CASE(5) MHA_CASE(6) MHA_CASE(7) MHA_CASE(8)
#undef MHA_CASE
  }
  buctd_set_error("buctd_mha_fwd: no kernel for d=%d", d);
  return BUCTD_EINVAL;
}

extern "C" int buctd_mha_fwd(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                             float scale, float* out, float* lse, void* stream) {
  return mha_run(B, T, d, q, k, v, ldqk, ldv, scale, out, lse, 0, stream);
}

extern "C" int buctd_mha_fwd_bf16x6(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk,
                                    int ldv, float scale, float* out, float* lse, void* stream) {
  return mha_run(B, T, d, q, k, v, ldqk, ldv, scale, out, lse, 1, stream);
}

static int mha_x6_rs(int d) { return d * 6 + (((d * 6) % 64 == 32) ? 0 : 32); }
static constexpr int mha_x6_rs_c(int d) { return d * 6 + (((d * 6) % 64 == 32) ? 0 : 32); }

extern "C" size_t buctd_mha_fwd_bf16x6_workspace(int B, int T, int d) {
  if (B <= 0 || !buctd_mha_fwd_supported(T, d)) return 0;
  return (size_t)2 * B * T * mha_x6_rs(d);
}

template <int DF>
static int mha_launch_p(int B, int T, const float* q, const unsigned char* k6, const unsigned char* v6, int ldq, float scale,
                        float* out, float* lse, hipStream_t st) {
  static unsigned char attr_done[2][BUCTD_MAX_DEVICES] = {{0}};
  constexpr bool wide = true;
  // wide (default): 32 queries per wavefront, keys split over the two wave quartets, 32-key tiles; BUCTD_MHA_Q16=1 keeps the
  // 16-query kernel (bit-identical to buctd_mha_fwd_bf16x6) for comparison
  void (*fn)(const float*, const unsigned char*, const unsigned char*, int, int, int, float, float*, float*) =
      wide ? mha_fwd_x6q_kernel<DF> : mha_fwd_x6p_kernel<DF>;
  const size_t lds = wide ? (size_t)4 * 32 * mha_x6_rs_c(DF * 16) : mha_x6_lds(DF * 16);
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[wide ? 1 : 0], "buctd_mha_fwd_bf16x6_ws"))
    return rc;
  hipLaunchKernelGGL(fn, dim3((unsigned)(B * (T / MHA_BQ))), dim3(512), lds, st, q, k6, v6, B, T, ldq, scale,
                     out, lse);
  BUCTD_CHECK_LAUNCH("buctd_mha_fwd_bf16x6_ws");
  return BUCTD_OK;
}

extern "C" int buctd_mha_fwd_bf16x6_ws(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk,
                                       int ldv, float scale, float* out, float* lse, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && B > 0, "buctd_mha_fwd_bf16x6_ws: null pointer");
  BUCTD_CHECK_ARG(buctd_mha_fwd_supported(T, d),
                  "buctd_mha_fwd_bf16x6_ws: unsupported shape T%d d%d (T %% 128 == 0, d %% 16 == 0, d <= 128)", T, d);
  BUCTD_CHECK_ARG(ldqk >= d && ldv >= d && ldqk % 4 == 0 && ldv % 4 == 0,
                  "buctd_mha_fwd_bf16x6_ws: row strides must be >= d and 16-byte aligned");
  const size_t need = buctd_mha_fwd_bf16x6_workspace(B, T, d);
  BUCTD_CHECK_ARG(workspace && workspace_bytes >= need, "buctd_mha_fwd_bf16x6_ws: workspace of %zu bytes needed, %zu given", need,
                  workspace_bytes);
  BUCTD_CHECK_ARG((long)B * T * mha_x6_rs(d) < (1L << 40), "buctd_mha_fwd_bf16x6_ws: tensor too large");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* k6 = (unsigned char*)workspace;
  unsigned char* v6 = k6 + need / 2;
  const long rows = (long)B * T;
  long blocks = (rows * (d / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mha_kv_split_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, st, k, v, rows, d, ldqk, ldv, mha_x6_rs(d),
                     k6, v6);
  BUCTD_CHECK_LAUNCH("buctd_mha_fwd_bf16x6_ws (split)");
  switch (d / 16) {
#define MHA_CASE(n) case n: return mha_launch_p<n>(B, T, q, k6, v6, ldqk, scale, out, lse, st);
    MHA_CASE(1) MHA_CASE(2) MHA_CASE(3) MHA_CASE(4) MHA_CASE(5) MHA_CASE(6) MHA_CASE(7) MHA_CASE(8)
#undef MHA_CASE
  }
  buctd_set_error("buctd_mha_fwd_bf16x6_ws: no kernel for d=%d", d);
  return BUCTD_EINVAL;
}

