// Fused single-head self-attention for TRAINING the TransPose encoder (reference lib/models/transpose_h.py:168-213:
// nn.MultiheadAttention(d_model, 1 head, dropout 0.1) inside TransformerEncoderLayer; autograd of
// softmax(scale Q K^T) -> dropout -> . V): forward with in-kernel attention dropout + log-sum-exp, and a flash-style
// backward.  Nothing T x T ever reaches HBM (the materialised path wrote and re-read six T x T fp32 tensors per layer:
// 37.7 MB per image at T = 3072).  Exact fp32: every product on v_mfma_f32_16x16x4_f32 (bitwise an fp32 FMA chain).
//
// Dropout: the counter hash of common.h keyed by (seed, (b T + q) T + key) - the same function and index the materialised
// path (attention.hip) uses, so both paths drop the same elements for the same seed (tests compare them).
//
// Backward, with S = (scale Q) K^T, P = exp(S - lse), Z = keep / (1 - p), Pd = P o Z, O = Pd V, D_q = dO_q . O_q:
//     dV = Pd^T dO        dPd = dO V^T        dS = P o (Z o dPd - D)        dQ = scale dS K        dK = dS^T (scale Q)
// ONE kernel template, two roles.  A workgroup (8 wavefronts) OWNS 128 rows of one side - queries (role dQ) or keys (role
// dK/dV), 16 per wavefront, their operands X1 / X2 in registers for the whole kernel - and STREAMS the other side through
// LDS 64 rows at a time (Y1 / Y2, row-major, the next block travelling through registers):
//     role dQ  : own = queries: X1 = scale Q, X2 = dO;   stream = keys:    Y1 = K,       Y2 = V
//     role dKV : own = keys:    X1 = K,       X2 = V;    stream = queries: Y1 = scale Q, Y2 = dO
//   T1 = Y1 X1^T and T2 = Y2 X2^T are 16 x 16 tiles in the MFMA C layout: lane (i16, g) holds rows (stream) 4g..4g+3 of
//   column (own) i16.  That layout IS the A-operand layout of a product that contracts over the stream rows with the own
//   rows as output rows (A[i = own][k = g] at step r <-> stream row 4g + r), so E = dS (or Pd) feeds the accumulating
//   products   acc_own += E^T Y   straight from registers - no transposition through LDS - with B = Y[4g + r][16n + i16]
//   read as single floats from the row-major tile.
// Work per 16 x 16 tile pair: 84 (dQ) / 112 (dK, dV) MFMAs of 32 cycles; per image and layer (T / 16)^2 x 196 MFMAs.
#include "common.h"
#include "../../include/buctd_hip.h"

#define MT_BO 128        // rows owned by a workgroup (16 per wavefront)
#define MT_BS 64         // rows streamed per block

// ---------------------------------------------------------------------------------------------------------- forward ----
// mha_fwd_kernel of attn_mha.hip plus attention dropout on the probabilities that enter P V (the soft-max normaliser uses
// the undropped ones) and the log-sum-exp row statistic for the backward.
template <int DF>
__global__ __launch_bounds__(512, 1) void mha_fwd_train_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, int T, int ldqk, int ldv,
                                                               float scale, float p_drop, uint64_t seed,
                                                               float* __restrict__ out, float* __restrict__ lse) {
  constexpr int D = DF * 16, LDK = D + 8, LDV = MT_BS + 8, LDP = MT_BS + 8;
  constexpr int C4 = D / 4;
  constexpr int PL = (MT_BS * C4 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;                                            // [64][LDK]
  float* Vt = Ks + MT_BS * LDK;                              // [D][LDV]
  float* Ps = Vt + D * LDV;                                  // [8 waves][16][LDP]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, q0 = blockIdx.x * MT_BO + wave * 16;
  const float* qb = q + ((long)b * T) * ldqk;
  const float* kb = k + ((long)b * T) * ldqk;
  const float* vb = v + ((long)b * T) * ldv;
  float* Pw = Ps + wave * 16 * LDP;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;

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
  float mrow[4], lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

  f32x4 kreg[PL], vreg[PL];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      const int id = t + 512 * p;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MT_BS) {
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
      if (row < MT_BS) {
        *reinterpret_cast<f32x4*>(Ks + row * LDK + c4) = kreg[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) Vt[(c4 + e) * LDV + row] = vreg[p][e];
      }
    }
  };

  load_kv(0);
  for (int k0 = 0; k0 < T; k0 += MT_BS) {
    __syncthreads();
    store_kv();
    if (k0 + MT_BS < T) load_kv(k0 + MT_BS);
    __syncthreads();
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
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float mnew = fmaxf(mrow[r], mx);
      const float corr = __expf(mrow[r] - mnew);
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
    // dropout on what enters P V; C layout: row (query) g*4 + r, key nb*16 + i16
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = s[nb][r];
        if (p_drop > 0.f) {
          const uint64_t idx = ((uint64_t)b * T + (uint64_t)(q0 + g * 4 + r)) * (uint64_t)T + (uint64_t)(k0 + nb * 16 + i16);
          pv *= keep_scale(seed, idx, p_drop, inv_keep);
        }
        Pw[(g * 4 + r) * LDP + nb * 16 + i16] = pv;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wavefront reads back its own tile
    f32x4 pa[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) pa[m] = *reinterpret_cast<const f32x4*>(Pw + i16 * LDP + 16 * m + 4 * g);
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
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float inv = 1.f / lrow[r];
    float* op = out + ((long)b * T + q0 + g * 4 + r) * (long)D;
#pragma unroll
    for (int n = 0; n < DF; ++n) op[16 * n + i16] = o[n][r] * inv;
    if (i16 == 0) lse[(long)b * T + q0 + g * 4 + r] = mrow[r] + __logf(lrow[r]);
  }
}

// --------------------------------------------------------------------------------------------------------- backward ----
// D[row] = dO[row] . O[row]  (one wavefront per row of d <= 128 floats)
__global__ __launch_bounds__(256) void mha_rowdot_kernel(const float* __restrict__ a, const float* __restrict__ bmat, long rows,
                                                         int d, float* __restrict__ out) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += a[row * d + c] * bmat[row * d + c];
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

struct MhaBwdArgs {
  const float* q; const float* k; const float* v;     // [B][T] rows, strides ldqk (q, k) / ldv
  const float* dout;                                   // [B][T][d]
  const float* lse; const float* dvec;                 // [B][T]
  float* dq; float* dk; float* dv;                     // strides lddqk (dq, dk) / lddv
  int T, ldqk, ldv, lddqk, lddv;
  float scale, p_drop;
  uint64_t seed;
};

template <int DF, bool DKV>
__global__ __launch_bounds__(512, 1) void mha_bwd_kernel(MhaBwdArgs p) {
  constexpr int D = DF * 16, LDK = D + 8;
  constexpr int C4 = D / 4;
  constexpr int PL = (MT_BS * C4 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Y1s = sm;                       // [64][LDK]   dQ: K        dKV: scale Q
  float* Y2s = Y1s + MT_BS * LDK;        // [64][LDK]   dQ: V        dKV: dO
  float* St = Y2s + MT_BS * LDK;         // [64][2]     dKV: (lse, D) of the streamed queries
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, T = p.T;
  const int own0 = blockIdx.x * MT_BO + wave * 16;
  const long brow = (long)b * T;
  const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  // stream sources
  const float* y1 = DKV ? p.q : p.k;
  const float* y2 = DKV ? p.dout : p.v;
  const int ldy1 = p.ldqk, ldy2 = DKV ? D : p.ldv;
  const float y1scale = DKV ? p.scale : 1.f;

  // own operands in registers: lane (i16, g) holds X[own0 + i16][16 m + 4 g + e]
  f32x4 x1[DF], x2[DF];
  {
    const float* s1 = (DKV ? p.k : p.q) + (brow + own0 + i16) * (long)p.ldqk;
    const float* s2 = DKV ? p.v + (brow + own0 + i16) * (long)p.ldv : p.dout + (brow + own0 + i16) * (long)D;
#pragma unroll
    for (int m = 0; m < DF; ++m) {
      x1[m] = *reinterpret_cast<const f32x4*>(s1 + 16 * m + 4 * g);
      x2[m] = *reinterpret_cast<const f32x4*>(s2 + 16 * m + 4 * g);
      if (!DKV) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x1[m][e] *= p.scale;
      }
    }
  }
  // role dQ: the row statistics belong to the own column i16
  float lse_own = 0.f, d_own = 0.f;
  if (!DKV) {
    lse_own = p.lse[brow + own0 + i16];
    d_own = p.dvec[brow + own0 + i16];
  }
  f32x4 acc1[DF], acc2[DKV ? DF : 1];     // dQ: acc1 = dQ;  dKV: acc1 = dV, acc2 = dK
#pragma unroll
  for (int n = 0; n < DF; ++n) acc1[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < (DKV ? DF : 1); ++n) acc2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 r1[PL], r2[PL];
  float rst = 0.f;
  auto load_y = [&](int s0) {
#pragma unroll
    for (int q = 0; q < PL; ++q) {
      const int id = t + 512 * q;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MT_BS) {
        r1[q] = *reinterpret_cast<const f32x4*>(y1 + (brow + s0 + row) * (long)ldy1 + c4);
        r2[q] = *reinterpret_cast<const f32x4*>(y2 + (brow + s0 + row) * (long)ldy2 + c4);
      }
    }
    if (DKV && t < 2 * MT_BS) rst = (t & 1) ? p.dvec[brow + s0 + (t >> 1)] : p.lse[brow + s0 + (t >> 1)];
  };
  auto store_y = [&]() {
#pragma unroll
    for (int q = 0; q < PL; ++q) {
      const int id = t + 512 * q;
      const int row = id / C4, c4 = (id - row * C4) * 4;
      if (row < MT_BS) {
        f32x4 a = r1[q];
        if (DKV) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] *= y1scale;
        }
        *reinterpret_cast<f32x4*>(Y1s + row * LDK + c4) = a;
        *reinterpret_cast<f32x4*>(Y2s + row * LDK + c4) = r2[q];
      }
    }
    if (DKV && t < 2 * MT_BS) St[t] = rst;
  };

  load_y(0);
  for (int s0 = 0; s0 < T; s0 += MT_BS) {
    __syncthreads();
    store_y();
    if (s0 + MT_BS < T) load_y(s0 + MT_BS);
    __syncthreads();
#pragma unroll 1
    for (int tb = 0; tb < MT_BS / 16; ++tb) {
      // T1 = Y1 X1^T, T2 = Y2 X2^T : rows = stream rows tb*16 + 4g + r, column = own row i16
      f32x4 t1 = (f32x4){0.f, 0.f, 0.f, 0.f}, t2 = t1;
      const float* a1 = Y1s + (tb * 16 + i16) * LDK + 4 * g;
      const float* a2 = Y2s + (tb * 16 + i16) * LDK + 4 * g;
#pragma unroll
      for (int m = 0; m < DF; ++m) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(a1 + 16 * m);
        const f32x4 w = *reinterpret_cast<const f32x4*>(a2 + 16 * m);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(u[e], x1[m][e], t1, 0, 0, 0);
          t2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], x2[m][e], t2, 0, 0, 0);
        }
      }
      // element-wise: P = exp(S - lse_q), Pd = P Z, dS = P (Z dPd - D_q)
      f32x4 e_ds, e_pd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int srow = s0 + tb * 16 + 4 * g + r;            // stream row of this element
        const int qi = DKV ? srow : own0 + i16;
        const int ki = DKV ? own0 + i16 : srow;
        const float lq = DKV ? St[(tb * 16 + 4 * g + r) * 2] : lse_own;
        const float dq_ = DKV ? St[(tb * 16 + 4 * g + r) * 2 + 1] : d_own;
        const float pr = __expf(t1[r] - lq);
        float z = 1.f;
        if (p.p_drop > 0.f) z = keep_scale(p.seed, ((uint64_t)brow + (uint64_t)qi) * (uint64_t)T + (uint64_t)ki, p.p_drop, inv_keep);
        e_pd[r] = pr * z;
        e_ds[r] = pr * (z * t2[r] - dq_);
      }
      // acc_own += E^T Y : A = E (C layout -> A layout as is), B = Y[tb*16 + 4g + r][16 n + i16]
      const float* b1 = Y1s + (tb * 16 + 4 * g) * LDK + i16;
      const float* b2 = Y2s + (tb * 16 + 4 * g) * LDK + i16;
#pragma unroll
      for (int n = 0; n < DF; ++n) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (DKV) {
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(e_pd[r], b2[r * LDK + 16 * n], acc1[n], 0, 0, 0);   // dV += Pd^T dO
            acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(e_ds[r], b1[r * LDK + 16 * n], acc2[n], 0, 0, 0);   // dK += dS^T (scale Q)
          } else {
            acc1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(e_ds[r], b1[r * LDK + 16 * n], acc1[n], 0, 0, 0);   // dQ += dS K
          }
        }
      }
    }
  }
  // accumulators: lane holds own rows own0 + 4g + r, column 16 n + i16
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long row = brow + own0 + 4 * g + r;
    if (DKV) {
#pragma unroll
      for (int n = 0; n < DF; ++n) {
        p.dv[row * p.lddv + 16 * n + i16] = acc1[n][r];
        p.dk[row * p.lddqk + 16 * n + i16] = acc2[n][r];
      }
    } else {
#pragma unroll
      for (int n = 0; n < DF; ++n) p.dq[row * p.lddqk + 16 * n + i16] = acc1[n][r] * p.scale;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ host ----
static size_t mt_fwd_lds(int d) {
  return ((size_t)MT_BS * (d + 8) + (size_t)d * (MT_BS + 8) + (size_t)8 * 16 * (MT_BS + 8)) * sizeof(float);
}
static size_t mt_bwd_lds(int d) { return ((size_t)2 * MT_BS * (d + 8) + 2 * MT_BS) * sizeof(float); }

extern "C" int buctd_mha_train_supported(int T, int d) {
  return (T > 0 && T % MT_BO == 0 && d >= 16 && d <= 128 && d % 16 == 0) ? 1 : 0;
}

template <typename F>
static int mt_attr(F fn, unsigned char (&done)[BUCTD_MAX_DEVICES], const char* who) {
  return buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, done, who);
}

template <int DF>
static int mt_fwd_launch(int B, int T, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                         float p_drop, uint64_t seed, float* out, float* lse, hipStream_t st) {
  static unsigned char done[BUCTD_MAX_DEVICES] = {0};
  const int rc = mt_attr(mha_fwd_train_kernel<DF>, done, "buctd_mha_fwd_train");
  if (rc) return rc;
  hipLaunchKernelGGL(mha_fwd_train_kernel<DF>, dim3(T / MT_BO, B), dim3(512), mt_fwd_lds(DF * 16), st, q, k, v, T, ldqk, ldv,
                     scale, p_drop, seed, out, lse);
  BUCTD_CHECK_LAUNCH("buctd_mha_fwd_train");
  return BUCTD_OK;
}

template <int DF>
static int mt_bwd_launch(int B, const MhaBwdArgs& a, hipStream_t st) {
  static unsigned char done[2][BUCTD_MAX_DEVICES] = {{0}};
  int rc = mt_attr(mha_bwd_kernel<DF, false>, done[0], "buctd_mha_bwd");
  if (rc) return rc;
  rc = mt_attr(mha_bwd_kernel<DF, true>, done[1], "buctd_mha_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL((mha_bwd_kernel<DF, true>), dim3(a.T / MT_BO, B), dim3(512), mt_bwd_lds(DF * 16), st, a);
  BUCTD_CHECK_LAUNCH("buctd_mha_bwd (dK, dV)");
  hipLaunchKernelGGL((mha_bwd_kernel<DF, false>), dim3(a.T / MT_BO, B), dim3(512), mt_bwd_lds(DF * 16), st, a);
  BUCTD_CHECK_LAUNCH("buctd_mha_bwd (dQ)");
  return BUCTD_OK;
}

/* softmax(scale q k^T) -> dropout(p_drop, seed) -> . v for one head, fused (no T x T tensor), train mode: also writes the
 * row statistic lse[B][T] the backward needs.  q, k: rows of stride ldqk floats, v: ldv; out: [B][T][d] contiguous.
 * Reference: nn.MultiheadAttention in transpose_h.py:192-197 (forward of the training step). */
extern "C" int buctd_mha_fwd_train(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                                   float scale, float p_drop, uint64_t seed, float* out, float* lse, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && lse && B > 0, "buctd_mha_fwd_train: null pointer");
  BUCTD_CHECK_ARG(buctd_mha_train_supported(T, d), "buctd_mha_fwd_train: unsupported shape T%d d%d", T, d);
  BUCTD_CHECK_ARG(ldqk >= d && ldv >= d && ldqk % 4 == 0 && ldv % 4 == 0 && p_drop >= 0.f && p_drop < 1.f,
                  "buctd_mha_fwd_train: bad strides or dropout probability");
  hipStream_t st = (hipStream_t)stream;
  switch (d / 16) {
#define MT_CASE(n) case n: return mt_fwd_launch<n>(B, T, q, k, v, ldqk, ldv, scale, p_drop, seed, out, lse, st);
    MT_CASE(1) MT_CASE(2) MT_CASE(3) MT_CASE(4) MT_CASE(5) MT_CASE(6) MT_CASE(7) MT_CASE(8)
#undef MT_CASE
  }
  return BUCTD_EINVAL;
}

/* backward of buctd_mha_fwd_train: dq, dk (rows of stride lddqk), dv (stride lddv) from q, k, v, the forward's out and lse,
 * and dout [B][T][d].  workspace: B * T floats (the row dots dout . out). */
extern "C" size_t buctd_mha_bwd_workspace(int B, int T) { return (size_t)B * T * sizeof(float); }
extern "C" int buctd_mha_bwd(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                             const float* out, const float* dout, const float* lse, float scale, float p_drop, uint64_t seed,
                             float* dq, float* dk, int lddqk, float* dv, int lddv, void* workspace, size_t workspace_bytes,
                             void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && B > 0, "buctd_mha_bwd: null pointer");
  BUCTD_CHECK_ARG(buctd_mha_train_supported(T, d), "buctd_mha_bwd: unsupported shape T%d d%d", T, d);
  BUCTD_CHECK_ARG(ldqk >= d && ldv >= d && lddqk >= d && lddv >= d && ldqk % 4 == 0 && ldv % 4 == 0,
                  "buctd_mha_bwd: bad strides");
  if (!workspace || workspace_bytes < buctd_mha_bwd_workspace(B, T)) {
    buctd_set_error("buctd_mha_bwd: workspace %zu bytes < required %zu", workspace_bytes, buctd_mha_bwd_workspace(B, T));
    return BUCTD_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)B * T;
  hipLaunchKernelGGL(mha_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dout, out, rows, d, (float*)workspace);
  BUCTD_CHECK_LAUNCH("buctd_mha_bwd (row dots)");
  MhaBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.dout = dout; a.lse = lse; a.dvec = (const float*)workspace;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.T = T; a.ldqk = ldqk; a.ldv = ldv; a.lddqk = lddqk; a.lddv = lddv;
  a.scale = scale; a.p_drop = p_drop; a.seed = seed;
  switch (d / 16) {
#define MT_CASE(n) case n: return mt_bwd_launch<n>(B, a, st);
    MT_CASE(1) MT_CASE(2) MT_CASE(3) MT_CASE(4) MT_CASE(5) MT_CASE(6) MT_CASE(7) MT_CASE(8)
#undef MT_CASE
  }
  return BUCTD_EINVAL;
}
