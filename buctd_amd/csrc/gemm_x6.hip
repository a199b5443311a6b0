// bf16x6 GEMM: C = alpha * A * B (+ bias) with fp32 operands split exactly into three bf16 pieces and six
// v_mfma_f32_16x16x32_bf16 per product (the hl, lh, mm, mh, hm, hh terms, small to large) - the arithmetic of the
// bf16x6 3x3 convolutions (conv3x3.hip), for the large plain GEMMs of the path: fc_o = nn.Linear(T, T) of the CoAM
// channel attention (self_attention.py:150-159; T = 6912: 147 GFLOP per pass at N = 32, three passes per step).
//
// Both operands are consumed as prepared IMAGES in MFMA fragment order (x6_image_kernel, one memory-bound pass each):
//   block (vb, kb) = 16 vectors x 32 reduction slots = [3 pieces][64 lanes][8 bf16] = 3 KB; lane l of piece q holds
//   piece q of X[vb*16 + (l & 15)][kb*32 + (l >> 4)*8 .. +8].  Image = [Vpad/16][Kpad/32] blocks, k fastest.
// An A block (vectors = rows of C) and a B block (vectors = columns of C) have the same format, so one preparation
// kernel serves both, with a strided / grouped address function for the token-major activations.
//
// Kernel: 512 threads = 8 wavefronts (2 x 4), workgroup tile 128 x 192, wave tile 64 x 48 (MF = 4, NF = 3: the wave
// tile of the 48x36 convolution kernel).  The A blocks of two k-steps are copied global -> registers -> LDS as they are
// (no VALU; a wave's 64 lanes write one contiguous 1 KB piece, which is exactly the fragment a later ds_read_b128 picks
// up conflict-free), double buffered, one barrier per two steps; the B fragments come straight from L2 one step
// ahead, as in the convolution kernel.  LDS 96 KB -> one workgroup (two waves per SIMD) per CU.
#include "common.h"
#include <stdlib.h>
#include "../../include/buctd_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int GX_MF = 4, GX_NF = 3, GX_WM = 2, GX_WN = 4, GX_CH = 2;
constexpr int GX_BM = GX_WM * GX_MF * 16, GX_BN = GX_WN * GX_NF * 16;     // 128 x 192
constexpr int GX_KC = GX_CH * 32;                                          // reduction slots per chunk
constexpr int GX_KPAD = 4 * GX_KC;                                         // images are padded to four chunks
constexpr int GX_ABUF = (GX_BM / 16) * GX_CH * 3072;                       // one A buffer: 48 KB

struct GxArgs {
  const unsigned char* a;
  const unsigned char* b;
  float* c;
  const float* bias;
  int M, N, KB;          // KB = Kpad / 32 (steps)
  long ldc, gsc;
  int Nc;
  float alpha;
  int bias_axis;
  int row_major;
};

struct GxImg {
  const float* src;
  unsigned char* out;
  int V, K, Vpad, Kpad;
  int vg, kg;            // group sizes (0: none)
  long vgs, vs, kgs, ks; // address(v, k) = (v / vg) * vgs + (v % vg) * vs + (k / kg) * kgs + (k % kg) * ks
};

__global__ __launch_bounds__(256) void x6_image_kernel(GxImg p) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int kbn = p.Kpad / 32;
  const long nblk = (long)(p.Vpad / 16) * kbn;
  const long blk = gid >> 6;
  if (blk >= nblk) return;
  const int lane = (int)(gid & 63);
  const int vb = (int)(blk / kbn), kb = (int)(blk - (long)vb * kbn);
  const int v = vb * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 8;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = 0.f;
  if (v < p.V) {
    const long vo = p.vg ? (long)(v / p.vg) * p.vgs + (long)(v % p.vg) * p.vs : (long)v * p.vs;
    if (p.ks == 1 && !p.kg && k0 + 8 <= p.K && ((vo + k0) & 3) == 0) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.src + vo + k0);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.src + vo + k0 + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        if (k < p.K) x[e] = p.src[vo + (p.kg ? (long)(k / p.kg) * p.kgs + (long)(k % p.kg) * p.ks : (long)k * p.ks)];
      }
    }
  }
  u16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {          // exact three-way split: the residual subtractions are exact in fp32
    const __bf16 hh = (__bf16)x[e];
    const float r1 = x[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    const __bf16 ll = (__bf16)(r1 - (float)mm);
    h[e] = __builtin_bit_cast(unsigned short, hh);
    m[e] = __builtin_bit_cast(unsigned short, mm);
    l[e] = __builtin_bit_cast(unsigned short, ll);
  }
  unsigned char* o = p.out + blk * 3072 + lane * 16;
  *reinterpret_cast<u16x8*>(o) = h;
  *reinterpret_cast<u16x8*>(o + 1024) = m;
  *reinterpret_cast<u16x8*>(o + 2048) = l;
}

__global__ __launch_bounds__(512, 1) void x6_gemm_kernel(GxArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];    // [2][BM/16][CH][3][1024]
  constexpr int MF = GX_MF, NF = GX_NF, CH = GX_CH;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wave_m = wave % GX_WM, wave_n = wave / GX_WM;
  const int i16 = lane & 15, g = lane >> 4;
  // XCD-aware order: the workgroups of one XCD walk the tiles that share the LARGER operand's blocks back to back, so
  // that operand streams from HBM once and the smaller one is re-read from L2 / the Infinity Cache
  int bx, by;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = lin & 7, idx = lin >> 3, per = total >> 3, rem = total & 7;
    const unsigned L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    if (p.row_major) {        // consecutive workgroups share the ROW tile: the larger operand is A (fc_o: 287 MB of weights)
      bx = (int)(L / gy);
      by = (int)(L - (unsigned)bx * gy);
    } else {
      by = (int)(L / gx);
      bx = (int)(L - (unsigned)by * gx);
    }
  }
  const int mb0 = bx * (GX_BM / 16), nb0 = by * (GX_BN / 16);
  const int nchunks = p.KB / CH;

  // A staging: the chunk's 16 blocks x 3 pieces = 48 one-KB pieces; wave w copies pieces w, w + 8, ... through
  // registers.  (global_load_lds_dwordx4 would save the registers, but with LDS-DMA and ordinary loads pending on the
  // same counter the compiler gives up in-order counting and waits vmcnt(0) for every B fragment - measured in the ISA.)
  constexpr int NPC = (GX_BM / 16) * CH * 3 / 8;     // pieces per wave and chunk
  f32x4 areg[NPC];
  auto stage_load = [&](int c) {
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int piece = wave + 8 * j;                  // = (mb * CH + kk) * 3 + q
      const int blk = piece / 3, q = piece - blk * 3;
      const int mb = blk / CH, kk = blk - mb * CH;
      areg[j] = *reinterpret_cast<const f32x4*>(p.a + (((size_t)(mb0 + mb) * p.KB + (size_t)c * CH + kk) * 3 + q) * 1024 +
                                                lane * 16);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NPC; ++j)
      *reinterpret_cast<f32x4*>(smem + buf * GX_ABUF + (wave + 8 * j) * 1024 + lane * 16) = areg[j];
  };

  const unsigned char* bptr = p.b + ((size_t)(nb0 + wave_n * NF) * p.KB) * 3072 + lane * 16;
  bf16x8 bc[3][NF], bn[3][NF];
  auto load_b = [&](int kb, bf16x8 (&dst)[3][NF]) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        dst[q][nf] = *reinterpret_cast<const bf16x8*>(bptr + ((size_t)nf * p.KB + kb) * 3072 + q * 1024);
  };
  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int AD = 2;
  bf16x8 a[AD + 1][3];
  auto read_a = [&](const unsigned char* abase, int i, bf16x8 (&dst)[3]) {     // fragment i = (kk, mf) of the chunk
    const int kk = i / MF, mf = i % MF;
    const unsigned char* ap = abase + (((wave_m * MF + mf) * CH + kk) * 3) * 1024 + lane * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(ap + q * 1024);
  };

  stage_load(0);
  load_b(0, bn);
  stage_store(0);
  __syncthreads();
  int kb = 0;
  const int last = p.KB - 1;
  auto do_chunk = [&](int c, int buf) {          // buf = c & 1, as a constant
    if (c + 1 < nchunks) stage_load(c + 1);       // in front of this chunk's B fetches: landed when they are
    const unsigned char* abase = smem + buf * GX_ABUF;
#pragma unroll
    for (int i = 0; i < AD; ++i) read_a(abase, i, a[i]);
#pragma unroll
    for (int kk = 0; kk < CH; ++kk) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int q = 0; q < 3; ++q) bc[q][nf] = bn[q][nf];
      load_b(kb < last ? kb + 1 : last, bn);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int i = kk * MF + mf;
        if (i + AD < CH * MF) read_a(abase, i + AD, a[(i + AD) % (AD + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 (&ac)[3] = a[i % (AD + 1)];
#define GX_MMA(qa, qb) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[qa], bc[qb][nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) { GX_MMA(2, 0) GX_MMA(0, 2) GX_MMA(1, 1) GX_MMA(1, 0) GX_MMA(0, 1) GX_MMA(0, 0) }
#undef GX_MMA
        __builtin_amdgcn_sched_barrier(0);
      }
      ++kb;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nchunks) stage_store(buf ^ 1);    // that buffer was last read in chunk c - 1, a barrier ago
    __syncthreads();
  };
  // four chunks per trip (Kpad is a multiple of 4 chunks): the compiler resolves the B prefetch that crosses the loop
  // back-edge with a full vmcnt(0) in the last step of the body - one step in eight instead of one in two
  for (int c = 0; c < nchunks; c += 4) {
    do_chunk(c, 0);
    do_chunk(c + 1, 1);
    do_chunk(c + 2, 0);
    do_chunk(c + 3, 1);
  }

  // epilogue: lane (i16, g) holds C[m0 + g*4 + rg][n0 + i16] of every 16 x 16 fragment
  const int m_w = bx * GX_BM + wave_m * MF * 16, n_w = by * GX_BN + wave_n * NF * 16;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = n_w + nf * 16 + i16;
    if (n >= p.N) continue;
    const long co = (long)(n / p.Nc) * p.gsc + (n % p.Nc);
    const float bn_ = (p.bias && p.bias_axis == 0) ? p.bias[n] : 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int m = m_w + mf * 16 + g * 4 + rg;
        if (m >= p.M) continue;
        float v = acc[mf][nf][rg] * p.alpha + bn_;
        if (p.bias && p.bias_axis == 1) v += p.bias[m];
        p.c[(long)m * p.ldc + co] = v;
      }
  }
}

int pad_to(int x, int m) { return (x + m - 1) / m * m; }
}  // namespace

extern "C" int buctd_x6_image_dims(int V, int K, int role, int* Vpad, int* Kpad) {
  BUCTD_CHECK_ARG(V > 0 && K > 0 && (role == 0 || role == 1) && Vpad && Kpad, "buctd_x6_image_dims: bad argument");
  *Vpad = pad_to(V, role == 0 ? GX_BM : GX_BN);
  *Kpad = pad_to(K, GX_KPAD);
  return BUCTD_OK;
}

extern "C" size_t buctd_x6_image_bytes(int V, int K, int role) {
  if (V <= 0 || K <= 0 || (role != 0 && role != 1)) return 0;
  return (size_t)(pad_to(V, role == 0 ? GX_BM : GX_BN) / 16) * (pad_to(K, GX_KPAD) / 32) * 3072;
}

extern "C" int buctd_x6_image(const float* src, int V, int K, int vg, long vgs, long vs, int kg, long kgs, long ks, int role,
                              void* image, void* stream) {
  BUCTD_CHECK_ARG(src && image && V > 0 && K > 0 && (role == 0 || role == 1) && vg >= 0 && kg >= 0,
                  "buctd_x6_image: bad argument");
  GxImg p;
  p.src = src; p.out = (unsigned char*)image; p.V = V; p.K = K;
  p.Vpad = pad_to(V, role == 0 ? GX_BM : GX_BN); p.Kpad = pad_to(K, GX_KPAD);
  p.vg = vg; p.kg = kg; p.vgs = vgs; p.vs = vs; p.kgs = kgs; p.ks = ks;
  const long lanes = (long)(p.Vpad / 16) * (p.Kpad / 32) * 64;
  hipLaunchKernelGGL(x6_image_kernel, dim3((unsigned)ceil_div(lanes, 256)), dim3(256), 0, (hipStream_t)stream, p);
  BUCTD_CHECK_LAUNCH("buctd_x6_image");
  return BUCTD_OK;
}

extern "C" int buctd_x6_gemm(int M, int N, int K, const void* a_image, const void* b_image, const float* bias, int bias_axis,
                             float alpha, float* C, long ldc, int Nc, long gsc, void* stream) {
  BUCTD_CHECK_ARG(a_image && b_image && C && M > 0 && N > 0 && K > 0, "buctd_x6_gemm: bad argument");
  BUCTD_CHECK_ARG(bias_axis == 0 || bias_axis == 1, "buctd_x6_gemm: bias_axis must be 0 (per column) or 1 (per row)");
  if (Nc <= 0) Nc = N;
  static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(x6_gemm_kernel), 2 * GX_ABUF, attr_done, "buctd_x6_gemm")) return rc;
  GxArgs p;
  p.a = (const unsigned char*)a_image; p.b = (const unsigned char*)b_image; p.c = C; p.bias = bias;
  p.M = M; p.N = N; p.KB = pad_to(K, GX_KPAD) / 32; p.ldc = ldc; p.gsc = gsc; p.Nc = Nc; p.alpha = alpha;
  p.bias_axis = bias_axis;
  p.row_major = (size_t)pad_to(M, GX_BM) > (size_t)pad_to(N, GX_BN) ? 1 : 0;     // image bytes ~ V x Kpad
  dim3 grid(pad_to(M, GX_BM) / GX_BM, pad_to(N, GX_BN) / GX_BN);
  hipLaunchKernelGGL(x6_gemm_kernel, grid, dim3(512), 2 * GX_ABUF, (hipStream_t)stream, p);
  BUCTD_CHECK_LAUNCH("buctd_x6_gemm");
  return BUCTD_OK;
}
