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

static size_t mha_lds(int d) {
  return ((size_t)MHA_BK * (d + 8) + (size_t)d * (MHA_BK + 8) + (size_t)8 * 16 * (MHA_BK + 8)) * sizeof(float);
}

extern "C" int buctd_mha_fwd_supported(int T, int d) {
  return (T > 0 && T % MHA_BQ == 0 && d >= 16 && d <= 128 && d % 16 == 0) ? 1 : 0;
}

template <int DF>
static int mha_launch(int B, int T, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                      float* out, float* lse, hipStream_t st) {
  static bool attr_set = false;     // idempotent attribute call: a race at first use only repeats it
  auto fn = mha_fwd_kernel<DF>;
  const size_t lds = mha_lds(DF * 16);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      buctd_set_error("buctd_mha_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return BUCTD_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(fn, dim3(T / MHA_BQ, B), dim3(512), lds, st, q, k, v, T, ldqk, ldv, scale, out, lse);
  BUCTD_CHECK_LAUNCH("buctd_mha_fwd");
  return BUCTD_OK;
}

extern "C" int buctd_mha_fwd(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                             float scale, float* out, float* lse, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && B > 0, "buctd_mha_fwd: null pointer");
  BUCTD_CHECK_ARG(buctd_mha_fwd_supported(T, d), "buctd_mha_fwd: unsupported shape T%d d%d (T %% 128 == 0, d %% 16 == 0, d <= 128)",
                  T, d);
  BUCTD_CHECK_ARG(ldqk >= d && ldv >= d && ldqk % 4 == 0 && ldv % 4 == 0, "buctd_mha_fwd: row strides must be >= d and 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (d / 16) {
#define MHA_CASE(n) case n: return mha_launch<n>(B, T, q, k, v, ldqk, ldv, scale, out, lse, st);
    MHA_CASE(1) MHA_CASE(2) MHA_CASE(3) MHA_CASE(4) MHA_CASE(5) MHA_CASE(6) MHA_CASE(7) MHA_CASE(8)
#undef MHA_CASE
  }
  buctd_set_error("buctd_mha_fwd: no kernel for d=%d", d);
  return BUCTD_EINVAL;
}
