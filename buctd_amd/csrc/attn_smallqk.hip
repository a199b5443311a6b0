// Fused (flash-style) attention for the CoAM position-attention core when the query/key contraction is narrow.
//
// Reference lib/models/self_attention.py:74-86 computes  softmax(fc_q(y_cond) fc_k(y)^T / sqrt(C)) -> dropout -> . V
// and materialises the T x T matrix three times (191 MB per image at T = 6912).  Because fc_q acts on only
// d_cond (+1 for its bias) input channels, the logits have rank R = d_cond + 1:
//     S = [y_cond, 1] ([Wq | bq]^T K^T)          (verified there to 3e-9)
// so the host (buctd_amd/models/self_attention.py) hands this kernel  q' = [y_cond, 1, 0-pad] (B,T,R4)  and
// k' = K [Wq | bq] (B,T,R4), and the logits cost R4 FMAs per element on the VALU.  Nothing T x T ever reaches HBM:
//   attn_stats : per query row, running max m and sum l of exp(s - m) over all keys (one pass, LDS-staged keys)
//   attn_fwd   : O = P V with P = exp(s - m)/l (x dropout) regenerated on the fly as the MFMA A operand
//   attn_bwd_q : dP = dO V^T on the MFMA, dS = P (g - D), D = dO.O ; accumulates dq' ; also writes D
//   attn_bwd_kv: per key block: dV += Pd^T dO and dk' += dS^T q' (P regenerated in both fragment layouts)
// All contractions over C or T run on v_mfma_f32_16x16x4_f32 (exact fp32).  Algorithmic HBM traffic per image is
// O(T (R4 + C)) instead of O(T^2); the dropout mask is a counter hash of (seed, b*T + i, j).
#include "gemm_core.h"
#include "../../include/buctd_hip.h"

// Dropout mask: keep(i, j) = fin(rowkey(i) + colkey(j)) >= p * 2^32.  The two keys are full lowbias32 hashes of
// (seed, index), computed once per row / column of a tile; the per-element finisher is one rotate-xor (it breaks the additive
// structure of the key sum - without it the four masks of a rectangle (i, j), (i, j'), (i', j), (i', j') are correlated at
// 0.07 - 0.33) and ONE 24-bit multiply: v_mul_u32_u24 is full rate where the 32-bit v_mul_lo_u32 takes four issue slots, its
// low 24 input bits see all 32 bits of the sum through the rotate, and the comparison reads the top bits of the product -
// the best-mixed ones.  (Rounds 2-5 used a 32-bit multiply + shift-xor behind the rotate: 12 issue slots per element against
// 7, in kernels that are VALU-paced.  Keep rate, neighbour / row-pair / column-pair correlations and the rectangle statistic of
// both finishers at T = 4096: indistinguishable from independent draws - tests/test_host.py holds the numbers.)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rowkey(uint32_t s0, uint32_t row) { return mix32(s0 ^ (row * 0x9E3779B1u)); }
__device__ __forceinline__ uint32_t colkey(uint32_t s1, uint32_t col) { return mix32(s1 + col * 0x85EBCA77u); }
__device__ __forceinline__ float keepf(uint32_t rk, uint32_t ck, uint32_t thr, float inv_keep) {
  uint32_t x = rk + ck;
  x ^= __builtin_amdgcn_alignbit(x, x, 21);     // rotate left by 11
  x = __umul24(x, 0x9E3779u);                   // low 32 bits of (x & 0xFFFFFF) * 0x9E3779
  return x >= thr ? inv_keep : 0.f;
}

struct AttnArgs {
  const float* q;     // [B][T][R4]
  const float* k;     // [B][T][R4]
  const float* v;     // [B][T][C]
  const float* o;     // [B][T][C]   (bwd)
  const float* dout;  // [B][T][C]   (bwd)
  float* m;           // [B][T] row max of the scaled logits
  float* linv;        // [B][T] 1 / sum exp(s - m)
  float* dvec;        // [B][T] D = dO . O
  float* out;         // fwd: O ; bwd_q: dq' ; bwd_kv: dk'
  float* out2;        // bwd_kv: dV
  int T, C;
  int b3;               // split-operand kernels: 0 none (fp32 MFMA), 2 = two bf16 pieces (bf16x3), 3 = three (bf16x6)
  float scale, scale2, p_drop, inv_keep;   // scale2 = scale * log2(e): logits are formed in the exp2 domain
  uint32_t s0, s1, thr;
};

template <int R4>
__device__ __forceinline__ float dotr(const float (&a)[R4], const float* b) {
  float s = a[0] * b[0];     // explicit FMAs: the file is compiled with -ffp-contract=off
#pragma unroll
  for (int r = 1; r < R4; ++r) s = __builtin_fmaf(a[r], b[r], s);
  return s;
}
template <int R4>
__device__ __forceinline__ void loadr(float (&a)[R4], const float* p) {
#pragma unroll
  for (int r = 0; r < R4; r += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + r);
    a[r] = t.x; a[r + 1] = t.y; a[r + 2] = t.z; a[r + 3] = t.w;
  }
}

// ------------------------------------------------------------------------------------------- stats ----
// one thread per query row, keys streamed through LDS in chunks of 512
template <int R4>
__global__ __launch_bounds__(256) void attn_stats_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) float ks[512 * R4];
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float qv[R4];
#pragma unroll
  for (int r = 0; r < R4; ++r) qv[r] = 0.f;
  if (i < p.T) loadr<R4>(qv, p.q + ((long)b * p.T + i) * R4);
#pragma unroll
  for (int r = 0; r < R4; ++r) qv[r] *= p.scale2;
  float mx = -INFINITY, sum = 0.f;
  for (int j0 = 0; j0 < p.T; j0 += 512) {
    const int nj = p.T - j0 < 512 ? p.T - j0 : 512;
    __syncthreads();
    for (int e = threadIdx.x; e < nj * R4 / 4; e += 256)
      reinterpret_cast<f32x4*>(ks)[e] = reinterpret_cast<const f32x4*>(p.k + ((long)b * p.T + j0) * R4)[e];
    __syncthreads();
    for (int j = 0; j < nj; ++j) {
      const float s = dotr<R4>(qv, ks + j * R4);
      if (s > mx) {
        sum = sum * __builtin_amdgcn_exp2f(mx - s) + 1.f;
        mx = s;
      } else {
        sum += __builtin_amdgcn_exp2f(s - mx);
      }
    }
  }
  if (i < p.T) {
    p.m[(long)b * p.T + i] = mx;
    p.linv[(long)b * p.T + i] = 1.f / sum;
  }
}

// --------------------------------------------------------------------------------------------- fwd ----
// workgroup = 64 query rows (16 per wave); key blocks of 64
template <int R4, int CF, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  // B-operand reads touch rows 4s+kq at columns nf*16 + i16: conflict-free when the row stride is 16 mod 64 floats
  constexpr int C = CF * 16, LDV = C + (80 - C % 64) % 64;
  __shared__ __attribute__((aligned(16))) float ks[64 * R4];
  __shared__ __attribute__((aligned(16))) float vs[64 * LDV];
  __shared__ uint32_t cks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const int i = blockIdx.x * 64 + wave * 16 + i16;          // A-layout row of this lane
  const long rowbase = (long)b * p.T;
  const uint32_t rk = rowkey(p.s0, (uint32_t)(rowbase + i));
  float qv[R4];
  loadr<R4>(qv, p.q + (rowbase + i) * R4);
#pragma unroll
  for (int r = 0; r < R4; ++r) qv[r] *= p.scale2;
  const float mi = p.m[rowbase + i], li = p.linv[rowbase + i];
  f32x4 acc[CF];
#pragma unroll
  for (int nf = 0; nf < CF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < p.T; j0 += 64) {
    __syncthreads();
    for (int e = t; e < 64 * R4 / 4; e += 256)
      reinterpret_cast<f32x4*>(ks)[e] = reinterpret_cast<const f32x4*>(p.k + (rowbase + j0) * R4)[e];
    for (int e = t; e < 64 * (C / 4); e += 256) {
      const int r = e / (C / 4), c4 = e - r * (C / 4);
      *reinterpret_cast<f32x4*>(vs + r * LDV + c4 * 4) = reinterpret_cast<const f32x4*>(p.v + (rowbase + j0 + r) * C)[c4];
    }
    if (DROP && t < 64) cks[t] = colkey(p.s1, (uint32_t)(j0 + t));
    __syncthreads();
    // four k-steps at a time: probabilities on the VALU, all twelve V fragments in flight, then the MFMAs back to back
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 4) {
      float pv[4], bv[4][CF];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * (s0 + u) + kq;
#pragma unroll
        for (int nf = 0; nf < CF; ++nf) bv[u][nf] = vs[j * LDV + nf * 16 + i16];
        pv[u] = __builtin_amdgcn_exp2f(dotr<R4>(qv, ks + j * R4) - mi) * li;
        if (DROP) pv[u] *= keepf(rk, cks[j], p.thr, p.inv_keep);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int nf = 0; nf < CF; ++nf)
          acc[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[u], bv[u][nf], acc[nf], 0, 0, 0);
    }
  }
  // C/D layout: row = kq*4 + reg, col = nf*16 + i16
#pragma unroll
  for (int nf = 0; nf < CF; ++nf)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      p.out[(rowbase + blockIdx.x * 64 + wave * 16 + kq * 4 + rg) * C + nf * 16 + i16] = acc[nf][rg];
}

// ------------------------------------------------------------------------- bf16x3 variants (math mode) ----
// Same algorithm with the T- and C-contractions on v_mfma_f32_16x16x32_bf16 and split-fp32 operands
// (x = hi + lo, a*b ~= al*bh + ah*bl + ah*bh, fp32 accumulate, ~2^-16 relative product error): the fp32 MFMA
// (32 cycles for 2 kFLOP) paces the kernels above at 40-50 % pipe occupancy; three bf16 MFMAs (17 cycles, 16 kFLOP
// each) do the same product in a fifth of the time, which leaves the kernels VALU-bound (exp2, hash, split).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// Two fp32 values -> NP packed bf16 pairs (exact residual chain): one v_cvt_pk_bf16_f32 per piece delivers the packed
// word that is stored, the two residuals come from its halves (shift / mask) - same roundings as the element-wise form,
// 5 instead of 8-9 VALU instructions per value and piece pair.
typedef float a_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 a_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned a_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned a_u32x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&w)[NP]) {
  a_f32x2 r = {x0, x1};
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    w[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, a_bf16x2));
    if (q + 1 < NP) r -= (a_f32x2){__uint_as_float(w[q] << 16), __uint_as_float(w[q] & 0xffff0000u)};
  }
}
// channels c..c+3 of one LDS row -> NP bf16 pieces, piece q at byte q * LO + 2c
template <int LO, int NP>
__device__ __forceinline__ void split_store4(unsigned char* row, int c, f32x4 v) {
  unsigned lo[NP], hi[NP];
  split_pair<NP>(v[0], v[1], lo);
  split_pair<NP>(v[2], v[3], hi);
#pragma unroll
  for (int q = 0; q < NP; ++q) *reinterpret_cast<a_u32x2*>(row + q * LO + 2 * c) = (a_u32x2){lo[q], hi[q]};
}
// gfx950 transpose read: the 16 lanes of a group address 4 rows x 16 bf16; lane c receives the 4 rows of column c.
// Two of them (rows +0..3 at p, rows +0..3 at q) give the 8 reduction slots of one MFMA operand.
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p, const unsigned char* q) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&pc)[NP]) {
  unsigned w[4][NP];
#pragma unroll
  for (int e = 0; e < 4; ++e) split_pair<NP>(x[2 * e], x[2 * e + 1], w[e]);
#pragma unroll
  for (int q = 0; q < NP; ++q) pc[q] = __builtin_bit_cast(bf16x8, (a_u32x4){w[0][q], w[1][q], w[2][q], w[3][q]});
}
// acc += a . b on split operands, smallest terms first.  NP = 2: the three terms of weight >= 2^-8 (bf16x3, product
// error ~2^-16); NP = 3: the six terms of weight >= 2^-16 (bf16x6, fp32 class - see conv3x3.hip)
template <int NP>
__device__ __forceinline__ void mfma_split(f32x4& acc, const bf16x8 (&a)[NP], const bf16x8 (&b)[NP]) {
  if constexpr (NP == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  }
}
// LDS tile row of C channels in NP bf16 pieces, padded to 32 mod 64 bytes (conflict-free 16-byte row reads)
template <int C, int NP>
struct SplitRow {
  static constexpr int LO = C * 2;
  static constexpr int RS = C * 2 * NP + (((C * 2 * NP) % 64 == 32) ? 0 : 32);
};

// forward: workgroup = 64 query rows (16 per wave); per 32-key step lane (row i16, g) generates the probabilities of
// keys 8g..8g+7 - exactly the A operand of the K = 32 MFMA - and the V fragments (lane = channel, 8 keys) come out of
// the key-major bf16 tile through the transpose read.  The soft-max statistics are computed on the fly (running row
// maximum, accumulators rescaled when it moves - rare after the first tiles), so the separate statistics pass of the
// fp32 path is gone; m and 1/l are written at the end for the backward kernels.
template <int R4, int CF, bool DROP, int NP>
__global__ __launch_bounds__(256) void attn_fwd_b3_kernel(AttnArgs p) {
  constexpr int C = CF * 16, LO = SplitRow<C, NP>::LO, RS = SplitRow<C, NP>::RS;   // V rows: NP pieces of C bf16 | pad
  __shared__ __attribute__((aligned(16))) float ks[64 * R4];
  __shared__ __attribute__((aligned(16))) unsigned char vt[64 * RS];
  __shared__ uint32_t cks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int i = blockIdx.x * 64 + wave * 16 + i16;
  const long rowbase = (long)b * p.T;
  const uint32_t rk = rowkey(p.s0, (uint32_t)(rowbase + i));
  float qv[R4];
  loadr<R4>(qv, p.q + (rowbase + i) * R4);
#pragma unroll
  for (int r = 0; r < R4; ++r) qv[r] *= p.scale2;
  float mrun = -INFINITY, lrun = 0.f;     // row i16: running maximum (same in its 4 lanes), this lane's share of the sum
  f32x4 acc[CF];
#pragma unroll
  for (int nf = 0; nf < CF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tr_off = (g * 8 + (i16 >> 2)) * RS + (i16 & 3) * 8;
  // next tile's k' and V rows travel through registers while the current tile is being multiplied
  f32x4 kreg = (f32x4){0.f, 0.f, 0.f, 0.f}, vreg[CF];
  auto load_tile = [&](int j0) {
    if (t < 64 * R4 / 4) kreg = reinterpret_cast<const f32x4*>(p.k + (rowbase + j0) * R4)[t];
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      vreg[q] = reinterpret_cast<const f32x4*>(p.v + (rowbase + j0 + r) * C)[c4];
    }
  };
  auto store_tile = [&](int j0) {
    if (t < 64 * R4 / 4) reinterpret_cast<f32x4*>(ks)[t] = kreg;
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      split_store4<LO, NP>(vt + r * RS, c4 * 4, vreg[q]);
    }
    if (DROP && t < 64) cks[t] = colkey(p.s1, (uint32_t)(j0 + t));
  };
  load_tile(0);
  for (int j0 = 0; j0 < p.T; j0 += 64) {
    __syncthreads();
    store_tile(j0);
    __syncthreads();
    if (j0 + 64 < p.T) load_tile(j0 + 64);
#pragma unroll
    for (int kstep = 0; kstep < 2; ++kstep) {
      float sv[8], pv[8];
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sv[e] = dotr<R4>(qv, ks + (32 * kstep + 8 * g + e) * R4);
        mx = fmaxf(mx, sv[e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun, mx);
      if (__any(mnew > mrun)) {
        const float fac = __builtin_amdgcn_exp2f(mrun - mnew);   // 1 where the maximum stayed, 0 on the first tile
        lrun *= fac;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float fr = __shfl(fac, g * 4 + rg, 64);          // accumulator rows are g*4 + rg
#pragma unroll
          for (int nf = 0; nf < CF; ++nf) acc[nf][rg] *= fr;
        }
        mrun = mnew;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = __builtin_amdgcn_exp2f(sv[e] - mrun);
        lrun += pv[e];
        if (DROP) pv[e] *= keepf(rk, cks[32 * kstep + 8 * g + e], p.thr, p.inv_keep);
      }
      bf16x8 pp[NP];
      split8<NP>(pv, pp);
      const unsigned char* vb = vt + 32 * kstep * RS + tr_off;
#pragma unroll
      for (int nf = 0; nf < CF; ++nf) {
        bf16x8 bb[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) bb[q] = tr_pair(vb + nf * 32 + q * LO, vb + nf * 32 + q * LO + 4 * RS);
        mfma_split<NP>(acc[nf], pp, bb);
      }
    }
  }
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
  const float linv = 1.f / lrun;
  if (g == 0) {
    p.m[rowbase + i] = mrun;
    p.linv[rowbase + i] = linv;
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const float fl = __shfl(linv, g * 4 + rg, 64);
#pragma unroll
    for (int nf = 0; nf < CF; ++nf)
      p.out[(rowbase + blockIdx.x * 64 + wave * 16 + g * 4 + rg) * C + nf * 16 + i16] = acc[nf][rg] * fl;
  }
}

// 8 consecutive channels [c0, c0+8) of one fp32 row as NP bf16-piece MFMA operands (zeros past channel C)
template <int NP>
__device__ __forceinline__ void load_split8(const float* row, int c0, int C, bf16x8 (&pc)[NP]) {
  float x[8];
  if (c0 < C) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(row + c0 + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.f;
  }
  split8<NP>(x, pc);
}

// dq' pass: dP = dO V^T with the channel contraction on the bf16 MFMA (dO fragments live in registers for the whole
// kernel, V rows come out of the bf16 tile with plain 16-byte reads); everything else as attn_bwd_q_kernel.
template <int R4, int CF, bool DROP, int NP>
__global__ __launch_bounds__(256) void attn_bwd_q_b3_kernel(AttnArgs p) {
  constexpr int C = CF * 16, CS = C / 4, LO = SplitRow<C, NP>::LO, RS = SplitRow<C, NP>::RS, NK = (C + 31) / 32;
  __shared__ __attribute__((aligned(16))) float ks[64 * R4];
  __shared__ __attribute__((aligned(16))) unsigned char vt[64 * RS];
  __shared__ uint32_t cks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const long rowbase = (long)b * p.T;
  const int r0 = blockIdx.x * 64 + wave * 16;
  // D = dO . O of row r0 + i16 (each lane sums a quarter of the channels)
  float dpart = 0.f;
  {
    const float* dp = p.dout + (rowbase + r0 + i16) * C + kq * CS;
    const float* op = p.o + (rowbase + r0 + i16) * C + kq * CS;
#pragma unroll
    for (int s = 0; s < CS; s += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dp + s);
      const f32x4 o = *reinterpret_cast<const f32x4*>(op + s);
      dpart += d.x * o.x + d.y * o.y + d.z * o.z + d.w * o.w;
    }
  }
  dpart += __shfl_xor(dpart, 16, 64);
  dpart += __shfl_xor(dpart, 32, 64);
  if (kq == 0) p.dvec[rowbase + r0 + i16] = dpart;
  // A operand: dO[row r0 + i16][32 kk + 8 kq .. +7]
  bf16x8 aa[NK][NP];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) load_split8<NP>(p.dout + (rowbase + r0 + i16) * C, 32 * kk + 8 * kq, C, aa[kk]);
  float qrow[4][R4], mrow[4], lrow[4], drow[4];
  uint32_t rkrow[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const long row = rowbase + r0 + kq * 4 + rg;
    rkrow[rg] = rowkey(p.s0, (uint32_t)row);
    loadr<R4>(qrow[rg], p.q + row * R4);
#pragma unroll
    for (int r = 0; r < R4; ++r) qrow[rg][r] *= p.scale2;
    mrow[rg] = p.m[row];
    lrow[rg] = p.linv[row];
    drow[rg] = __shfl(dpart, kq * 4 + rg, 64);
  }
  float dq[4][R4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int r = 0; r < R4; ++r) dq[rg][r] = 0.f;
  // byte offset of this lane's 8 channels inside a tile row (clamped for the zero-padded tail of the last step)
  int boff[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * kq;
    boff[kk] = (c0 < C ? c0 : 0) * 2;
  }

  f32x4 kreg = (f32x4){0.f, 0.f, 0.f, 0.f}, vreg[CF];
  auto load_tile = [&](int j0) {
    if (t < 64 * R4 / 4) kreg = reinterpret_cast<const f32x4*>(p.k + (rowbase + j0) * R4)[t];
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      vreg[q] = reinterpret_cast<const f32x4*>(p.v + (rowbase + j0 + r) * C)[c4];
    }
  };
  auto store_tile = [&](int j0) {
    if (t < 64 * R4 / 4) reinterpret_cast<f32x4*>(ks)[t] = kreg;
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      split_store4<LO, NP>(vt + r * RS, c4 * 4, vreg[q]);
    }
    if (DROP && t < 64) cks[t] = colkey(p.s1, (uint32_t)(j0 + t));
  };
  load_tile(0);
  for (int j0 = 0; j0 < p.T; j0 += 64) {
    __syncthreads();
    store_tile(j0);
    __syncthreads();
    if (j0 + 64 < p.T) load_tile(j0 + 64);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      f32x4 dp4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      const unsigned char* vrow = vt + (16 * f + i16) * RS;    // B(k = channel, n = key 16f + i16)
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        bf16x8 bb[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) bb[q] = *reinterpret_cast<const bf16x8*>(vrow + boff[kk] + q * LO);
        mfma_split<NP>(dp4, aa[kk], bb);
      }
      const int j = 16 * f + i16;
      float kv[R4];
      loadr<R4>(kv, ks + j * R4);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float pv = __builtin_amdgcn_exp2f(dotr<R4>(qrow[rg], kv) - mrow[rg]) * lrow[rg];
        float gg = dp4[rg];
        if (DROP) gg *= keepf(rkrow[rg], cks[j], p.thr, p.inv_keep);
        const float ds = pv * (gg - drow[rg]);
#pragma unroll
        for (int r = 0; r < R4; ++r) dq[rg][r] += ds * kv[r];
      }
    }
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      float v = dq[rg][r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (i16 == 0) p.out[(rowbase + r0 + kq * 4 + rg) * R4 + r] = v * p.scale;
    }
}

// dk' / dV pass.  dP (queries as rows) per 16 x 16 block as in attn_bwd_kv_kernel; two such blocks give, per lane,
// eight dropped probabilities of its key - the A operand of ONE K = 32 MFMA of dV += Pd^T dO when the reduction slot
// e of lane group g is enumerated as query 16 (e >> 2) + 4 g + (e & 3); the dO fragments in that enumeration come
// out of the query-major bf16 tile through two transpose reads.
template <int R4, int CF, bool DROP, int NP>
__global__ __launch_bounds__(256) void attn_bwd_kv_b3_kernel(AttnArgs p) {
  constexpr int C = CF * 16, LO = SplitRow<C, NP>::LO, RS = SplitRow<C, NP>::RS, NK = (C + 31) / 32;
  __shared__ __attribute__((aligned(16))) float qs[64 * R4];
  __shared__ __attribute__((aligned(16))) unsigned char dt[64 * RS];     // dO tile: [query][hi C | lo C]
  __shared__ float ms[64], ls[64], dsm[64];
  __shared__ uint32_t rks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const long rowbase = (long)b * p.T;
  const int j0 = blockIdx.x * 64 + wave * 16;
  const uint32_t ck = colkey(p.s1, (uint32_t)(j0 + i16));
  float ka[R4];
  loadr<R4>(ka, p.k + (rowbase + j0 + i16) * R4);
  // B operand of dP: V[key j0 + i16][32 kk + 8 kq .. +7]
  bf16x8 vv[NK][NP];
  int boff[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const int c0 = 32 * kk + 8 * kq;
    load_split8<NP>(p.v + (rowbase + j0 + i16) * C, c0, C, vv[kk]);
    boff[kk] = (c0 < C ? c0 : 0) * 2;
  }
  f32x4 dv[CF];
#pragma unroll
  for (int nf = 0; nf < CF; ++nf) dv[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dk[R4];
#pragma unroll
  for (int r = 0; r < R4; ++r) dk[r] = 0.f;
  const int tr_off = (4 * kq + (i16 >> 2)) * RS + (i16 & 3) * 8;   // rows 4 kq .. 4 kq + 3 of a 16-query block

  f32x4 qreg = (f32x4){0.f, 0.f, 0.f, 0.f}, dreg[CF];
  float mreg = 0.f, lreg = 0.f, sreg = 0.f;
  auto load_tile = [&](int i0) {
    if (t < 64 * R4 / 4) qreg = reinterpret_cast<const f32x4*>(p.q + (rowbase + i0) * R4)[t];
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      dreg[q] = reinterpret_cast<const f32x4*>(p.dout + (rowbase + i0 + r) * C)[c4];
    }
    if (t < 64) {
      mreg = p.m[rowbase + i0 + t];
      lreg = p.linv[rowbase + i0 + t];
      sreg = p.dvec[rowbase + i0 + t];
    }
  };
  auto store_tile = [&](int i0) {
    if (t < 64 * R4 / 4) reinterpret_cast<f32x4*>(qs)[t] = qreg * p.scale2;
#pragma unroll
    for (int q = 0; q < CF; ++q) {
      const int e = t + 256 * q, r = e / (C / 4), c4 = e - r * (C / 4);
      split_store4<LO, NP>(dt + r * RS, c4 * 4, dreg[q]);
    }
    if (t < 64) {
      ms[t] = mreg;
      ls[t] = lreg;
      dsm[t] = sreg;
      if (DROP) rks[t] = rowkey(p.s0, (uint32_t)(rowbase + i0 + t));
    }
  };
  load_tile(0);
  for (int i0 = 0; i0 < p.T; i0 += 64) {
    __syncthreads();
    store_tile(i0);
    __syncthreads();
    if (i0 + 64 < p.T) load_tile(i0 + 64);
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // 32 queries per step: blocks f = 2h, 2h + 1
      float pd[8];
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        const int f = 2 * h + fb;
        // dP[query 16f + m][key n]: A = dO rows of the tile, B = this lane's V fragments
        f32x4 dp4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned char* drow = dt + (16 * f + i16) * RS;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          bf16x8 aa[NP];
#pragma unroll
          // lanes of the zero-padded tail of the channel contraction (32 kk + 8 kq >= C) read the row's first channels
          // (boff is clamped): finite values against the zeros load_split8 put into vv - the product is 0 either way
          for (int q = 0; q < NP; ++q) aa[q] = *reinterpret_cast<const bf16x8*>(drow + boff[kk] + q * LO);
          mfma_split<NP>(dp4, aa, vv[kk]);
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int i = 16 * f + 4 * kq + rg;
          float qv[R4];
          loadr<R4>(qv, qs + i * R4);
          const float pv = __builtin_amdgcn_exp2f(dotr<R4>(ka, qv) - ms[i]) * ls[i];
          float keep = 1.f;
          if (DROP) keep = keepf(rks[i], ck, p.thr, p.inv_keep);
          const float ds = pv * (dp4[rg] * keep - dsm[i]);
#pragma unroll
          for (int r = 0; r < R4; ++r) dk[r] += ds * qv[r];
          pd[4 * fb + rg] = pv * keep;
        }
      }
      bf16x8 pp[NP];
      split8<NP>(pd, pp);
      const unsigned char* db = dt + 32 * h * RS + tr_off;
#pragma unroll
      for (int nf = 0; nf < CF; ++nf) {
        bf16x8 bb[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) bb[q] = tr_pair(db + nf * 32 + q * LO, db + nf * 32 + q * LO + 16 * RS);
        mfma_split<NP>(dv[nf], pp, bb);
      }
    }
  }
#pragma unroll
  for (int nf = 0; nf < CF; ++nf)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) p.out2[(rowbase + j0 + kq * 4 + rg) * C + nf * 16 + i16] = dv[nf][rg];
#pragma unroll
  for (int r = 0; r < R4; ++r) {
    float v = dk[r];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (kq == 0) p.out[(rowbase + j0 + i16) * R4 + r] = v * (p.scale / p.scale2);
  }
}

// ------------------------------------------------------------------------------------------- bwd_q ----
template <int R4, int CF, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs p) {
  constexpr int C = CF * 16, CS = C / 4, LDV = C + 8;  // b128 row reads: stride 32 mod 64 bytes
  __shared__ __attribute__((aligned(16))) float ks[64 * R4];
  __shared__ __attribute__((aligned(16))) float vs[64 * LDV];
  __shared__ uint32_t cks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const long rowbase = (long)b * p.T;
  const int r0 = blockIdx.x * 64 + wave * 16;
  // A operand: dO[row r0+i16][c = kq*CS + s]  (the reduction index c is permuted identically for A and B)
  float ado[CS];
  float dpart = 0.f;
  {
    const float* dp = p.dout + (rowbase + r0 + i16) * C + kq * CS;
    const float* op = p.o + (rowbase + r0 + i16) * C + kq * CS;
#pragma unroll
    for (int s = 0; s < CS; s += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dp + s);
      const f32x4 o = *reinterpret_cast<const f32x4*>(op + s);
      ado[s] = d.x; ado[s + 1] = d.y; ado[s + 2] = d.z; ado[s + 3] = d.w;
      dpart += d.x * o.x + d.y * o.y + d.z * o.z + d.w * o.w;
    }
  }
  dpart += __shfl_xor(dpart, 16, 64);
  dpart += __shfl_xor(dpart, 32, 64);            // D of row r0 + i16, in every lane of that column group
  if (kq == 0) p.dvec[rowbase + r0 + i16] = dpart;
  // C/D layout rows of this lane: r0 + kq*4 + rg
  float qrow[4][R4], mrow[4], lrow[4], drow[4];
  uint32_t rkrow[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const long row = rowbase + r0 + kq * 4 + rg;
    rkrow[rg] = rowkey(p.s0, (uint32_t)row);
    loadr<R4>(qrow[rg], p.q + row * R4);
#pragma unroll
    for (int r = 0; r < R4; ++r) qrow[rg][r] *= p.scale2;
    mrow[rg] = p.m[row];
    lrow[rg] = p.linv[row];
    drow[rg] = __shfl(dpart, kq * 4 + rg, 64);   // lane (kq*4+rg) has i16 == that row
  }
  float dq[4][R4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int r = 0; r < R4; ++r) dq[rg][r] = 0.f;

  for (int j0 = 0; j0 < p.T; j0 += 64) {
    __syncthreads();
    for (int e = t; e < 64 * R4 / 4; e += 256)
      reinterpret_cast<f32x4*>(ks)[e] = reinterpret_cast<const f32x4*>(p.k + (rowbase + j0) * R4)[e];
    for (int e = t; e < 64 * (C / 4); e += 256) {
      const int r = e / (C / 4), c4 = e - r * (C / 4);
      *reinterpret_cast<f32x4*>(vs + r * LDV + c4 * 4) = reinterpret_cast<const f32x4*>(p.v + (rowbase + j0 + r) * C)[c4];
    }
    if (DROP && t < 64) cks[t] = colkey(p.s1, (uint32_t)(j0 + t));
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      // dP[rows][keys 16f..16f+15] = dO V^T : B(k = c, n = key) = V[key][c]
      f32x4 dp4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* vrow = vs + (16 * f + i16) * LDV + kq * CS;
#pragma unroll
      for (int s = 0; s < CS; ++s) dp4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ado[s], vrow[s], dp4, 0, 0, 0);
      const int j = 16 * f + i16;                 // C/D column of this lane
      float kv[R4];
      loadr<R4>(kv, ks + j * R4);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float pv = __builtin_amdgcn_exp2f(dotr<R4>(qrow[rg], kv) - mrow[rg]) * lrow[rg];
        float g = dp4[rg];
        if (DROP) g *= keepf(rkrow[rg], cks[j], p.thr, p.inv_keep);
        const float ds = pv * (g - drow[rg]);
#pragma unroll
        for (int r = 0; r < R4; ++r) dq[rg][r] += ds * kv[r];
      }
    }
  }
  // reduce over the 16 column lanes, scale, store dq'
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      float v = dq[rg][r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (i16 == 0) p.out[(rowbase + r0 + kq * 4 + rg) * R4 + r] = v * p.scale;
    }
}

// ------------------------------------------------------------------------------------------ bwd_kv ----
// workgroup = 64 keys (16 per wave), loop over query blocks of 64.  Per 16 x 16 (query, key) block:
//   dP[i][j] = dO V^T on the MFMA with the QUERIES as rows: in the C/D layout lane (j = lane&15, kq) then holds queries
//   i = 4 kq + reg, which is exactly the A-operand layout of Pd^T for  dV += Pd^T dO  when the reduction index of that
//   MFMA is enumerated as i = 4 kq + s (legal: A and B use the same permutation).  So P, the dropout hash and dS are
//   computed ONCE per element and feed both the dk' accumulation (lane-local) and the dV MFMA.
template <int R4, int CF, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnArgs p) {
  constexpr int C = CF * 16, CS = C / 4, LDV = C + 8;  // b128 row reads: stride 32 mod 64 bytes
  __shared__ __attribute__((aligned(16))) float qs[64 * R4];
  __shared__ __attribute__((aligned(16))) float dos[64 * LDV];
  __shared__ float ms[64], ls[64], dsm[64];
  __shared__ uint32_t rks[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y;
  const long rowbase = (long)b * p.T;
  const int j0 = blockIdx.x * 64 + wave * 16;      // first key of this wave; this lane's key is j0 + i16
  const uint32_t ck = colkey(p.s1, (uint32_t)(j0 + i16));
  float ka[R4];
  loadr<R4>(ka, p.k + (rowbase + j0 + i16) * R4);
  // B operand of dP: V[key j0 + i16][c = kq*CS + s]
  float av[CS];
  {
    const float* vp = p.v + (rowbase + j0 + i16) * C + kq * CS;
#pragma unroll
    for (int s = 0; s < CS; s += 4) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(vp + s);
      av[s] = d.x; av[s + 1] = d.y; av[s + 2] = d.z; av[s + 3] = d.w;
    }
  }
  f32x4 dv[CF];
#pragma unroll
  for (int nf = 0; nf < CF; ++nf) dv[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dk[R4];
#pragma unroll
  for (int r = 0; r < R4; ++r) dk[r] = 0.f;

  for (int i0 = 0; i0 < p.T; i0 += 64) {
    __syncthreads();
    for (int e = t; e < 64 * R4; e += 256) qs[e] = p.q[(rowbase + i0) * R4 + e] * p.scale2;
    for (int e = t; e < 64 * (C / 4); e += 256) {
      const int r = e / (C / 4), c4 = e - r * (C / 4);
      *reinterpret_cast<f32x4*>(dos + r * LDV + c4 * 4) =
          reinterpret_cast<const f32x4*>(p.dout + (rowbase + i0 + r) * C)[c4];
    }
    if (t < 64) {
      ms[t] = p.m[rowbase + i0 + t];
      ls[t] = p.linv[rowbase + i0 + t];
      dsm[t] = p.dvec[rowbase + i0 + t];
      if (DROP) rks[t] = rowkey(p.s0, (uint32_t)(rowbase + i0 + t));
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      // dP[query 16f + m][key n]: A(m, k = c) = dO[16f + i16][kq*CS + s], B(k = c, n) = V[key i16][kq*CS + s]
      f32x4 dp4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* drow = dos + (16 * f + i16) * LDV + kq * CS;
#pragma unroll
      for (int s = 0; s < CS; ++s) dp4 = __builtin_amdgcn_mfma_f32_16x16x4f32(drow[s], av[s], dp4, 0, 0, 0);
      // C/D layout: lane (key i16, kq), reg rg <-> query 16f + 4kq + rg
      float pd[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = 16 * f + 4 * kq + rg;
        float qv[R4];
        loadr<R4>(qv, qs + i * R4);                 // carries scale * log2(e); undone at the final store
        const float pv = __builtin_amdgcn_exp2f(dotr<R4>(ka, qv) - ms[i]) * ls[i];
        float keep = 1.f;
        if (DROP) keep = keepf(rks[i], ck, p.thr, p.inv_keep);
        const float ds = pv * (dp4[rg] * keep - dsm[i]);
#pragma unroll
        for (int r = 0; r < R4; ++r) dk[r] += ds * qv[r];
        pd[rg] = pv * keep;
      }
      // dV[key m][c n] += sum_i Pd[i][key] dO[i][c]: A(m = key i16, k = i = 4kq + s) = pd[s],
      //                                              B(k = i, n = c) = dO[16f + 4kq + s][nf*16 + i16]
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nf = 0; nf < CF; ++nf)
          dv[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[s], dos[(16 * f + 4 * kq + s) * LDV + nf * 16 + i16], dv[nf],
                                                        0, 0, 0);
    }
  }
  // dV: C/D layout rows = keys j0 + kq*4 + rg, cols = nf*16 + i16
#pragma unroll
  for (int nf = 0; nf < CF; ++nf)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) p.out2[(rowbase + j0 + kq * 4 + rg) * C + nf * 16 + i16] = dv[nf][rg];
  // dk': this lane holds the partial of key j0 + i16 over its queries (4 of every 16): sum the 4 kq lanes
#pragma unroll
  for (int r = 0; r < R4; ++r) {
    float v = dk[r];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (kq == 0) p.out[(rowbase + j0 + i16) * R4 + r] = v * (p.scale / p.scale2);
  }
}

// -------------------------------------------------------------------------------------------- host ----
static bool attn_shape_ok(int T, int R4, int C) {
  return T > 0 && T % 64 == 0 && (R4 == 4 || R4 == 8 || R4 == 16 || R4 == 20) && (C == 48 || C == 96 || C == 192 ||
                                                                                   C == 16 || C == 32 || C == 64 ||
                                                                                   C == 128);
}

extern "C" int buctd_attn_smallqk_supported(int T, int R4, int C) { return attn_shape_ok(T, R4, C) ? 1 : 0; }

template <int R4, int CF>
static void attn_launch(int which, const AttnArgs& a, int B, hipStream_t st) {
  const dim3 grid(a.T / 64, B);
  const bool drop = a.p_drop > 0.f;
#define ATTN_SPLIT(kern)                                                                                         \
  do {                                                                                                           \
    if (a.b3 == 3) {                                                                                             \
      if (drop) hipLaunchKernelGGL((kern<R4 <= 8 ? R4 : 4, CF, true, 3>), grid, dim3(256), 0, st, a);            \
      else hipLaunchKernelGGL((kern<R4 <= 8 ? R4 : 4, CF, false, 3>), grid, dim3(256), 0, st, a);                \
    } else {                                                                                                     \
      if (drop) hipLaunchKernelGGL((kern<R4 <= 8 ? R4 : 4, CF, true, 2>), grid, dim3(256), 0, st, a);            \
      else hipLaunchKernelGGL((kern<R4 <= 8 ? R4 : 4, CF, false, 2>), grid, dim3(256), 0, st, a);                \
    }                                                                                                            \
  } while (0)
  if (which == 1 && a.b3 && R4 <= 8) {
    ATTN_SPLIT(attn_fwd_b3_kernel);
  } else if (which == 1) {
    if (drop) hipLaunchKernelGGL((attn_fwd_kernel<R4, CF, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<R4, CF, false>), grid, dim3(256), 0, st, a);
  } else if (which == 2 && a.b3 && R4 <= 8) {
    ATTN_SPLIT(attn_bwd_q_b3_kernel);
  } else if (which == 3 && a.b3 && R4 <= 8) {
    ATTN_SPLIT(attn_bwd_kv_b3_kernel);
#undef ATTN_SPLIT
  } else if (which == 2) {
    if (drop) hipLaunchKernelGGL((attn_bwd_q_kernel<R4, CF, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_bwd_q_kernel<R4, CF, false>), grid, dim3(256), 0, st, a);
  } else {
    if (drop) hipLaunchKernelGGL((attn_bwd_kv_kernel<R4, CF, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_bwd_kv_kernel<R4, CF, false>), grid, dim3(256), 0, st, a);
  }
}
template <int R4>
static void attn_dispatch_c(int which, const AttnArgs& a, int B, hipStream_t st) {
  switch (a.C) {
    case 16: attn_launch<R4, 1>(which, a, B, st); break;
    case 32: attn_launch<R4, 2>(which, a, B, st); break;
    case 48: attn_launch<R4, 3>(which, a, B, st); break;
    case 64: attn_launch<R4, 4>(which, a, B, st); break;
    case 96: attn_launch<R4, 6>(which, a, B, st); break;
    case 128: attn_launch<R4, 8>(which, a, B, st); break;
    default: attn_launch<R4, 12>(which, a, B, st); break;
  }
}
static void attn_dispatch(int which, const AttnArgs& a, int R4, int B, hipStream_t st) {
  if (which == 0) {
    const dim3 grid(ceil_div(a.T, 256), B);
    if (R4 == 4) hipLaunchKernelGGL((attn_stats_kernel<4>), grid, dim3(256), 0, st, a);
    else if (R4 == 8) hipLaunchKernelGGL((attn_stats_kernel<8>), grid, dim3(256), 0, st, a);
    else if (R4 == 16) hipLaunchKernelGGL((attn_stats_kernel<16>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_stats_kernel<20>), grid, dim3(256), 0, st, a);
    return;
  }
  if (R4 == 4) attn_dispatch_c<4>(which, a, B, st);
  else if (R4 == 8) attn_dispatch_c<8>(which, a, B, st);
  else if (R4 == 16) attn_dispatch_c<16>(which, a, B, st);
  else attn_dispatch_c<20>(which, a, B, st);
}

// flag of the C ABI -> pieces per operand: 0 fp32 MFMA kernels, 1 -> two bf16 pieces, 2 -> three (bf16x6; its 192-channel
// tile rows would need 77 KB of static LDS, so that width stays on the fp32 kernels)
static int attn_split_mode(int flag, int C) { return flag == 2 ? (C <= 128 ? 3 : 0) : (flag ? 2 : 0); }

static AttnArgs attn_args(int T, int C, float scale, float p_drop, uint64_t seed) {
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.T = T; a.C = C; a.scale = scale; a.scale2 = scale * 1.44269504088896340736f; a.p_drop = p_drop;
  a.inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  a.s0 = (uint32_t)seed; a.s1 = (uint32_t)(seed >> 32);
  const double th = (double)p_drop * 4294967296.0;
  a.thr = th >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)th;
  return a;
}

extern "C" int buctd_attn_smallqk_fwd(int B, int T, int R4, int C, const float* q, const float* k, const float* v,
                                      float scale, float p_drop, uint64_t seed, int bf16x3, float* out, float* m,
                                      float* linv, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && out && m && linv && B > 0, "buctd_attn_smallqk_fwd: null argument");
  BUCTD_CHECK_ARG(attn_shape_ok(T, R4, C), "buctd_attn_smallqk_fwd: unsupported T=%d R4=%d C=%d", T, R4, C);
  BUCTD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (long)B * T < 2147483647L, "buctd_attn_smallqk_fwd: bad p_drop / size");
  AttnArgs a = attn_args(T, C, scale, p_drop, seed);
  a.q = q; a.k = k; a.v = v; a.m = m; a.linv = linv; a.out = out; a.b3 = attn_split_mode(bf16x3, C);
  if (!(a.b3 && R4 <= 8)) {            // the bf16x3 forward kernel computes the soft-max statistics on the fly
    attn_dispatch(0, a, R4, B, (hipStream_t)stream);
    BUCTD_CHECK_LAUNCH("buctd_attn_smallqk_fwd(stats)");
  }
  attn_dispatch(1, a, R4, B, (hipStream_t)stream);
  BUCTD_CHECK_LAUNCH("buctd_attn_smallqk_fwd");
  return BUCTD_OK;
}

extern "C" int buctd_attn_smallqk_bwd(int B, int T, int R4, int C, const float* q, const float* k, const float* v,
                                      const float* o, const float* dout, const float* m, const float* linv,
                                      float scale, float p_drop, uint64_t seed, int bf16x3, float* dq, float* dk,
                                      float* dv, float* dvec_workspace, void* stream) {
  BUCTD_CHECK_ARG(q && k && v && o && dout && m && linv && dq && dk && dv && dvec_workspace && B > 0,
                  "buctd_attn_smallqk_bwd: null argument");
  BUCTD_CHECK_ARG(attn_shape_ok(T, R4, C), "buctd_attn_smallqk_bwd: unsupported T=%d R4=%d C=%d", T, R4, C);
  AttnArgs a = attn_args(T, C, scale, p_drop, seed);
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout;
  a.m = const_cast<float*>(m); a.linv = const_cast<float*>(linv); a.dvec = dvec_workspace; a.b3 = attn_split_mode(bf16x3, C);
  a.out = dq;
  attn_dispatch(2, a, R4, B, (hipStream_t)stream);
  BUCTD_CHECK_LAUNCH("buctd_attn_smallqk_bwd(q)");
  a.out = dk; a.out2 = dv;
  attn_dispatch(3, a, R4, B, (hipStream_t)stream);
  BUCTD_CHECK_LAUNCH("buctd_attn_smallqk_bwd(kv)");
  return BUCTD_OK;
}
