// Batched strided fp32 MFMA matmul for the CoAM attention contractions and the
// TransPose encoder (reference lib/models/self_attention.py:74-87,146-159 and
// the nn.MultiheadAttention calls of lib/models/transpose_h.py:192-213).
//
//   C[b](m, n) = alpha * sum_k A[b](m, k) * B[b](k, n)  (+ bias[n])
//
// Operand addressing (element offsets from the batch base):
//   A rows : m*lda + (k / Kc)*gsAk + k % Kc        (k contiguous inside a group)
//   A cols : k*lda + m                             (transposed operand)
//   B rows : n*ldb + (k / Kc)*gsBk + k % Kc
//   B cols : k*ldb + (n / Nc)*gsBn + n % Nc        (n contiguous inside a group)
//   C      : m*ldc + (n / Nc)*gsCn + n % Nc
// The k-groups let the reduction run over (batch, channel) pairs of an NHWC
// tensor (fc_o weight gradient); the n-groups let the columns run over (batch,
// channel) pairs (fc_o forward / data gradient), so a 6912x6912 weight is
// streamed once for the whole batch instead of once per image.
// Split-K (grid.z = batch*nsplit) writes slabs that splitk_reduce sums.
#include "gemm_core.h"
#include "../../include/buctd_hip.h"

struct MMArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  long sAb, sBb, sCb;
  int lda, ldb, ldc;
  int Kc;
  long gsAk, gsBk;
  int Nc;
  long gsBn, gsCn;
  float alpha;
  int bias_axis;
  int nsplit, k_per_split;
};

// row image: thread (arow, chunk) fetches 4 consecutive k of one row
template <int PASSES, int TILE_ROWS, bool VEC>
__device__ __forceinline__ void mm_load_rows(const float* __restrict__ base, int r0, int rmax, int ld, int Kc, long gs,
                                             int k0, int k_end, f32x4* reg, int arow, int chunk) {
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int rl = arow + 64 * q;
    const int r = r0 + rl;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (rl < TILE_ROWS && r < rmax) {
      const int kb = k0 + chunk * 4;
      if (VEC) {
        if (kb < k_end) {
          const int g = kb / Kc, c = kb - g * Kc;
          v = *reinterpret_cast<const f32x4*>(base + (long)r * ld + g * gs + c);
        }
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = kb + j;
          e[j] = 0.f;
          if (k < k_end) {
            const int g = k / Kc, c = k - g * Kc;
            e[j] = base[(long)r * ld + g * gs + c];
          }
        }
        v = (f32x4){e[0], e[1], e[2], e[3]};
      }
    }
    reg[q] = v;
  }
}

// col image: thread fetches 4 consecutive columns of one k row
template <int PASSES, int N4S, bool VEC>
__device__ __forceinline__ void mm_load_cols(const float* __restrict__ base, int c0, int cmax, int ld, int Gc, long gs,
                                             int k0, int k_end, f32x4* reg, int t) {
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int idx = t + 256 * q;
    const int krow = idx / N4S, c4 = idx - krow * N4S;
    const int k = k0 + krow, col = c0 + c4 * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (krow < GK && k < k_end && col < cmax) {
      if (VEC) {
        const int g = col / Gc, c = col - g * Gc;
        v = *reinterpret_cast<const f32x4*>(base + (long)k * ld + g * gs + c);
      } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[j] = 0.f;
          const int cc = col + j;
          if (cc < cmax) {
            const int g = cc / Gc, c = cc - g * Gc;
            e[j] = base[(long)k * ld + g * gs + c];
          }
        }
        v = (f32x4){e[0], e[1], e[2], e[3]};
      }
    }
    reg[q] = v;
  }
}

template <class T, bool ACOL, bool BCOL, bool VEC>
__global__ __launch_bounds__(256) void matmul_kernel(MMArgs p) {
  constexpr int BM = T::BM, BN = T::BN;
  using AImg = OperandImage<BM, ACOL>;
  using BImg = OperandImage<BN, BCOL>;
  constexpr int PA = (BM + 63) / 64, PB = (BN + 63) / 64;
  constexpr int NA4 = BM / 4, NB4 = BN / 4;
  constexpr int PAK = (GK * NA4 + 255) / 256, PBK = (GK * NB4 + 255) / 256;

  __shared__ __attribute__((aligned(16))) float lds[2 * (AImg::SIZE + BImg::SIZE)];
  constexpr int STAGE = AImg::SIZE + BImg::SIZE;


  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int batch = blockIdx.z / p.nsplit, split = blockIdx.z - batch * p.nsplit;
  const int k_begin = split * p.k_per_split;
  int k_end = k_begin + p.k_per_split;
  if (k_end > p.K) k_end = p.K;
  const float* Ab = p.A + (long)batch * p.sAb;
  const float* Bb = p.B + (long)batch * p.sBb;
  const int arow = t >> 2, chunk = t & 3;

  f32x4 areg[ACOL ? PAK : PA], breg[BCOL ? PBK : PB];

  auto load_stage = [&](int k0) {
    if (!ACOL) mm_load_rows<PA, BM, VEC>(Ab, m0, p.M, p.lda, p.Kc, p.gsAk, k0, k_end, areg, arow, chunk);
    else mm_load_cols<PAK, NA4, VEC>(Ab, m0, p.M, p.lda, 0x7fffffff, 0, k0, k_end, areg, t);
    if (!BCOL) mm_load_rows<PB, BN, VEC>(Bb, n0, p.N, p.ldb, p.Kc, p.gsBk, k0, k_end, breg, arow, chunk);
    else mm_load_cols<PBK, NB4, VEC>(Bb, n0, p.N, p.ldb, p.Nc, p.gsBn, k0, k_end, breg, t);
  };
  auto store_stage = [&](int buf) {
    if (!ACOL) {
#pragma unroll
      for (int q = 0; q < PA; ++q) {
        const int rl = arow + 64 * q;
        if (rl < BM) *reinterpret_cast<f32x4*>((lds + buf * STAGE) + rl * AImg::LD + chunk * 4) = areg[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < PAK; ++q) {
        const int idx = t + 256 * q;
        const int krow = idx / NA4, c4 = idx - krow * NA4;
        if (krow < GK) *reinterpret_cast<f32x4*>((lds + buf * STAGE) + krow * AImg::LD + c4 * 4) = areg[q];
      }
    }
    if (!BCOL) {
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const int rl = arow + 64 * q;
        if (rl < BN) *reinterpret_cast<f32x4*>((lds + buf * STAGE + AImg::SIZE) + rl * BImg::LD + chunk * 4) = breg[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < PBK; ++q) {
        const int idx = t + 256 * q;
        const int krow = idx / NB4, c4 = idx - krow * NB4;
        if (krow < GK) *reinterpret_cast<f32x4*>((lds + buf * STAGE + AImg::SIZE) + krow * BImg::LD + c4 * 4) = breg[q];
      }
    }
  };

  f32x4 acc[T::MF][T::NF];
  zero_acc<T>(acc);
  const int nk = (k_end - k_begin + GK - 1) / GK;
  if (nk > 0) {
    load_stage(k_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_stage(k_begin + (kt + 1) * GK);
    mma_stage<T, ACOL, BCOL>(lds + cur * STAGE, lds + cur * STAGE + AImg::SIZE, acc, wm, wn, lane);
    if (kt + 1 < nk) store_stage(cur ^ 1);
    __syncthreads();
  }

  // split-K: the host points C at the dense slab array [batch][split][M][N]
  float* Cb = p.C + (long)batch * p.sCb;
#pragma unroll
  for (int nf = 0; nf < T::NF; ++nf) {
    const int n = n0 + acc_col<T>(wn, nf, lane);
    if (n < p.N) {
      const int g = n / p.Nc, c = n - g * p.Nc;
      const long coff = (long)g * p.gsCn + c;
      const float bvn = (p.bias && p.bias_axis == 0) ? p.bias[n] : 0.f;
#pragma unroll
      for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int m = m0 + acc_row<T>(wm, mf, lane, rg);
          if (m < p.M) {
            if (p.nsplit > 1)
              Cb[((long)split * p.M + m) * p.N + n] = acc[mf][nf][rg];
            else
              Cb[(long)m * p.ldc + coff] = acc[mf][nf][rg] * p.alpha + ((p.bias && p.bias_axis == 1) ? p.bias[m] : bvn);
          }
        }
    }
  }
}

// reduce split-K slabs [batch][nsplit][M][N] into the strided C: 32 outputs x 8 split-lanes per workgroup (every slab
// read independent; a serial loop per output would chain hundreds of dependent loads for the tall-skinny products)
__global__ __launch_bounds__(256) void matmul_splitk_reduce(const float* __restrict__ part, MMArgs p, int nbatch) {
  __shared__ float sm[8][32];
  const long per = (long)p.M * p.N;
  const long total = per * nbatch;
  const int col = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (long base = (long)blockIdx.x * 32; base < total; base += (long)gridDim.x * 32) {
    const long i = base + col;
    float s = 0.f;
    int b = 0, m = 0, n = 0;
    if (i < total) {
      b = (int)(i / per);
      const long r = i - (long)b * per;
      m = (int)(r / p.N);
      n = (int)(r - (long)m * p.N);
      for (int z = zl; z < p.nsplit; z += 8) s += part[((long)(b * p.nsplit + z) * p.M + m) * p.N + n];
    }
    sm[zl][col] = s;
    __syncthreads();
    if (zl == 0 && i < total) {
#pragma unroll
      for (int k = 1; k < 8; ++k) s += sm[k][col];
      const int g = n / p.Nc, c = n - g * p.Nc;
      p.C[(long)b * p.sCb + (long)m * p.ldc + (long)g * p.gsCn + c] =
          s * p.alpha + (p.bias ? p.bias[p.bias_axis == 1 ? m : n] : 0.f);
    }
    __syncthreads();
  }
}

template <class T, bool ACOL, bool BCOL, bool VEC>
static void launch_mm(const MMArgs& a, int nbatch, hipStream_t st) {
  dim3 grid(ceil_div(a.M, T::BM), ceil_div(a.N, T::BN), nbatch * a.nsplit);
  hipLaunchKernelGGL((matmul_kernel<T, ACOL, BCOL, VEC>), grid, dim3(256), 0, st, a);
}

template <bool ACOL, bool BCOL>
static void dispatch_mm(const MMArgs& a, int nbatch, bool vec, hipStream_t st) {
  if (!vec) {
    launch_mm<TileCfg<4, 1, 2, 4>, ACOL, BCOL, false>(a, nbatch, st);  // 128x64 generic
  } else if (a.N <= 48) {
    launch_mm<TileCfg<4, 1, 2, 3>, ACOL, BCOL, true>(a, nbatch, st);   // 128x48
  } else if (a.N % 96 == 0 && a.N % 128 != 0) {
    launch_mm<TileCfg<2, 2, 4, 3>, ACOL, BCOL, true>(a, nbatch, st);   // 128x96
  } else {
    launch_mm<TileCfg<2, 2, 4, 4>, ACOL, BCOL, true>(a, nbatch, st);   // 128x128
  }
}

static void mm_split_plan(const buctd_matmul_desc* d, int* nsplit, int* kps) {
  const int tn = d->N <= 48 ? 48 : ((d->N % 96 == 0 && d->N % 128 != 0) ? 96 : 128);
  const long tiles = (long)ceil_div(d->M, 128) * ceil_div(d->N, tn) * d->batch;
  long want = 1;
  if (tiles < 128 && d->K >= 1024) {
    want = (1024 + tiles - 1) / tiles;
    const long maxsplit = d->K / 256;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
  }
  long per = (d->K + want - 1) / want;
  per = ((per + GK - 1) / GK) * GK;
  *kps = (int)per;
  *nsplit = (int)((d->K + per - 1) / per);
}

extern "C" size_t buctd_matmul_workspace(const buctd_matmul_desc* d) {
  if (!d) return 0;
  int ns, kps;
  mm_split_plan(d, &ns, &kps);
  if (ns <= 1) return 0;
  return (size_t)d->batch * ns * d->M * d->N * sizeof(float);
}

extern "C" int buctd_matmul(const buctd_matmul_desc* d, const float* A, const float* B, const float* bias, float* C,
                            void* workspace, size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(d && A && B && C, "buctd_matmul: null argument");
  BUCTD_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0, "buctd_matmul: non-positive dimension");
  BUCTD_CHECK_ARG(d->Kc > 0 && d->Nc > 0, "buctd_matmul: Kc/Nc must be positive (use K / N for a single group)");
  BUCTD_CHECK_ARG(d->a_layout == 0 || d->a_layout == 1, "buctd_matmul: a_layout must be 0 (rows) or 1 (cols)");
  BUCTD_CHECK_ARG(d->b_layout == 0 || d->b_layout == 1, "buctd_matmul: b_layout must be 0 (rows) or 1 (cols)");
  MMArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.sAb = d->stride_a; a.sBb = d->stride_b; a.sCb = d->stride_c;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.Kc = d->Kc; a.gsAk = d->group_stride_a; a.gsBk = d->group_stride_bk;
  a.Nc = d->Nc; a.gsBn = d->group_stride_bn; a.gsCn = d->group_stride_c;
  a.alpha = d->alpha;
  a.bias_axis = d->bias_axis;
  mm_split_plan(d, &a.nsplit, &a.k_per_split);
  hipStream_t st = (hipStream_t)stream;

  auto mult4 = [](long v) { return (v & 3) == 0; };
  bool vec = mult4(d->lda) && mult4(d->ldb) && mult4(d->stride_a) && mult4(d->stride_b) &&
             (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
  const bool rows_used = d->a_layout == 0 || d->b_layout == 0;
  if (rows_used) vec = vec && (d->Kc % 4 == 0) && mult4(d->group_stride_a) && mult4(d->group_stride_bk) && (d->K % 4 == 0);
  if (d->a_layout == 1) vec = vec && (d->M % 4 == 0);
  if (d->b_layout == 1) vec = vec && (d->Nc % 4 == 0) && mult4(d->group_stride_bn) && (d->N % 4 == 0);

  MMArgs k = a;
  if (a.nsplit > 1) {
    const size_t need = (size_t)d->batch * a.nsplit * d->M * d->N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      buctd_set_error("buctd_matmul: workspace %zu bytes < required %zu", workspace_bytes, need);
      return BUCTD_EWORKSPACE;
    }
    k.C = (float*)workspace;
    k.sCb = (long)a.nsplit * d->M * d->N;
  }
  if (d->a_layout == 0 && d->b_layout == 0) dispatch_mm<false, false>(k, d->batch, vec, st);
  else if (d->a_layout == 0 && d->b_layout == 1) dispatch_mm<false, true>(k, d->batch, vec, st);
  else if (d->a_layout == 1 && d->b_layout == 0) dispatch_mm<true, false>(k, d->batch, vec, st);
  else dispatch_mm<true, true>(k, d->batch, vec, st);
  BUCTD_CHECK_LAUNCH("buctd_matmul");
  if (a.nsplit > 1) {
    const long total = (long)d->batch * d->M * d->N;
    int blocks = ceil_div(total, 32);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(matmul_splitk_reduce, dim3(blocks), dim3(256), 0, st, (const float*)workspace, a, d->batch);
    BUCTD_CHECK_LAUNCH("buctd_matmul(reduce)");
  }
  return BUCTD_OK;
}
