// Weight gradient of the 3x3 / stride 1 / pad 1 convolution from x6 PLANES (x6p.h), bf16x6 arithmetic - the successor of
// conv3x3_wgrad.hip for 48-channel blocks (HRNet-W48: 48 / 96 / 192 / 384 channels), reference
// lib/models/pose_hrnet.py:28-57 (autograd of nn.Conv2d):
//
//     dW[co][tap][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci]          p over the zero-padded flattened positions
//
// What changed against conv3x3_wgrad.hip, and why (its PMC profile: 4.0 VALU instructions per MFMA, 2.13x the algorithmic
// HBM traffic, two barriers per 64 positions):
//   * both operands arrive pre-split and zero-padded (planes), so a stage is moved global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4): no staging registers, no split arithmetic, no per-row div/mod, no zero-selects, no
//     ds_write.  What is left on the VALU is the ring addressing of the transpose reads (~0.3 instructions per MFMA);
//   * ONE 512-thread workgroup per CU instead of two of 256: the two halves of the workgroup take the two 32-position
//     k-steps of a 64-position stage and share one X ring (one halo instead of two), their accumulators meet in LDS once at
//     the end: 256 partial slabs instead of 512 (slab traffic 42 -> 21 MB for the 48 -> 48 filter);
//   * stage s+1 lands while stage s is multiplied (dY double-buffered, X ring one 64-row block ahead): one barrier per
//     stage, and it only waits for DMA that has had a whole stage of MFMAs to arrive.
// GEMM view: M = co (3 fragments of 16), N = (tap, ci16) (27 fragments, dealt tap-wise to the four waves of a half: 7/7/7/6),
// K = positions.  Fragments come out of the position-major tiles through ds_read_b64_tr_b16 exactly as in conv3x3_wgrad.hip
// (a lane group's 8 positions are {4g..4g+3} u {16+4g..16+4g+3}; row stride 288 = 32 mod 64 bytes: conflict-free).
#include "common.h"
#include "x6p.h"
#include "../../include/buctd_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define W4_KB 64                      // positions per stage
#define W4_ROWB 288                   // bytes per LDS row: 48 channels x 3 pieces
#define W4_BLK (W4_KB * W4_ROWB)      // one 64-row block = 18432 bytes = 18 DMA wave-instructions of 1 KB
#define W4_NINS 18
#define W4_MAX_SW 75
#ifndef W4_ABL
#define W4_ABL 0     // experiment builds only (scratch/wg4_abl.sh): 1 no barrier, 2 no MFMA, 3 no DMA in the loop
#endif

struct WG4Args {
  const unsigned char* xp;   // X planes,  row 0
  const unsigned char* dp;   // dY planes, row 0
  float* part;               // [nsplit][Co][9][Ci]
  int Ci, Co, SW;
  int split_q, split_rem;    // stages per split: q, the first split_rem splits q + 1
  int nblk;                  // 64-row blocks a stage reads = ceil((64 + 2 * (SW + 1)) / 64)
  int NB;                    // ring size in blocks = nblk + 1
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 w4_tr(const unsigned char* p, const unsigned char* q) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one 1 KB piece of a 64-row block: LDS bytes [u*1024, u*1024 + 1024) of the block <- 64 lanes x 16 bytes of the planes.
// piece i = u*64 + lane = (row i / 18, 16-byte column i % 18) of the 48-channel sub-block starting at byte `col0` of a row
__device__ __forceinline__ void w4_dma(const unsigned char* plane_row0, int row_bytes, int col0, unsigned char* lds_block,
                                       int u, int lane) {
  const int i = u * 64 + lane;
  const int row = (i * 3641) >> 16;               // i / 18 for i < 1152
  const int j = i - row * 18;
  const unsigned char* src = plane_row0 + (long)row * row_bytes + col0 + j * 16;
  // Inline assembly on purpose: with the builtin (__builtin_amdgcn_global_load_lds) hipcc (ROCm 7.2) cannot tell the DMA's
  // LDS destination from the tiles the MFMA loop reads and drains the DMA counter (s_waitcnt vmcnt(0)) in front of the
  // first ds_read of the stage - the prefetch would not overlap anything.  The asm form is invisible to that pass; the
  // kernel orders the hand-over itself: w4_dma_wait() in front of the barrier that ends the stage.  M0 = LDS byte address
  // of the wave's 1 KB (the hardware adds lane * 16).
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds_block + u * 1024);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void w4_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad4_kernel(WG4Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Dt = smem;                      // dY: 2 buffers of one 64-row block
  unsigned char* Xt = smem + 2 * W4_BLK;         // X ring: NB blocks

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = wave >> 2, wq = wave & 3;
  const int t16 = lane & 15, g = lane >> 4;
  const int co0 = blockIdx.x * 48, ci0 = blockIdx.y * 48;
  const int z = blockIdx.z;
  const int s_begin = z * p.split_q + (z < p.split_rem ? z : p.split_rem);
  const int ns = p.split_q + (z < p.split_rem ? 1 : 0);
  const int halo = p.SW + 1;
  const int R = p.NB * W4_KB;
  const int drow = p.Co * 6, xrow = p.Ci * 6;
  const int dcol = (co0 >> 4) * 96, xcol = (ci0 >> 4) * 96;
  // stream row 0 of this split: dY position k_begin, X position k_begin - halo
  const unsigned char* dsrc = p.dp + (long)s_begin * W4_KB * drow;
  const unsigned char* xsrc = p.xp + ((long)s_begin * W4_KB - halo) * xrow;

  f32x4 acc[3][7];
#pragma unroll
  for (int mf = 0; mf < 3; ++mf)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[mf][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read lane addressing (conv3x3_wgrad.hip): lane t16 of group g points at row 4g + (t16 >> 2) (second read:
  // + 16), channels 4 (t16 & 3) .. + 3
  const int lane_row = g * 4 + (t16 >> 2), lane_col = (t16 & 3) * 8;

  // prologue: dY block 0 and X blocks 0 .. nblk-1 of the stream
  if (ns > 0 && W4_ABL != 6) {
    for (int v = wave; v < W4_NINS * (1 + p.nblk); v += 8) {
      const int b = v / W4_NINS, u = v - b * W4_NINS;
      if (b == 0) w4_dma(dsrc, drow, dcol, Dt, u, lane);
      else w4_dma(xsrc + (long)(b - 1) * W4_KB * xrow, xrow, xcol, Xt + (b - 1) * W4_BLK, u, lane);
    }
  }
  w4_dma_wait();
  __syncthreads();

  int slot_next = p.nblk % p.NB;      // ring slot of the block the next prefetch fills (stream block s + nblk)
  int sbase = 0;                      // ring row of stream row 64 s
  for (int s = 0; s < (W4_ABL == 4 ? 0 : ns); ++s) {
    if (s + 1 < ns) {
      // stage s + 1: its dY block into the other buffer (read during stage s - 1), stream block s + nblk into the slot of
      // block s - 1; 36 wave-instructions dealt to the 8 waves
      const unsigned char* dn = dsrc + (long)(s + 1) * W4_KB * drow;
      const unsigned char* xn = xsrc + (long)(s + p.nblk) * W4_KB * xrow;
      unsigned char* dl = Dt + ((s + 1) & 1) * W4_BLK;
      unsigned char* xl = Xt + slot_next * W4_BLK;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int v = wave + 8 * k;
        if (W4_ABL == 3) break;
        if (v < W4_NINS) w4_dma(dn, drow, dcol, dl, v, lane);
        else if (v < 2 * W4_NINS) w4_dma(xn, xrow, xcol, xl, v - W4_NINS, lane);
      }
      slot_next = slot_next + 1 == p.NB ? 0 : slot_next + 1;
    }

    // ---- this half's k-step: positions 64 s + 32 half + (0..31) ----
    // Software pipeline, pinned with scheduling fences: the six transpose reads of fragment f + 1 are issued in front of
    // the 18 MFMAs of fragment f (an in-order wave otherwise alternates "read, wait, multiply": measured, the reads and the
    // MFMAs of a stage simply added up: 16 + 25 us over the launch).
    const unsigned char* xq0[3];
    const unsigned char* xq1[3];
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      const int tap = jt < 2 ? 2 * wq + jt : 8;
      const int tr = tap / 3, tc = tap - tr * 3;
      int rs = sbase + 32 * half + tr * p.SW + tc;       // ring row of this k-step's first X row for the tap (scalar)
      if (rs >= R) rs -= R;
      int r0 = rs + lane_row;
      if (r0 >= R) r0 -= R;
      int r1 = r0 + 16;
      if (r1 >= R) r1 -= R;
      xq0[jt] = Xt + r0 * W4_ROWB + lane_col + (jt == 2 ? wq * 96 : 0);   // the shared tap 8: channel fragment cf = wq
      xq1[jt] = Xt + r1 * W4_ROWB + lane_col + (jt == 2 ? wq * 96 : 0);
    }
    const int nfr = wq == 3 ? 6 : 7;                     // wave 3 of a half owns taps 6, 7 only
    bf16x8 a[3][3], bb[2][3];
    auto load_b = [&](int f, bf16x8 (&dst)[3]) {         // f compile-time: fragment f = (tap slot f / 3, cf f % 3), 6 = tap 8
      const int jt = f / 3, cfo = (f - jt * 3) * 96;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) dst[pc] = w4_tr(xq0[jt] + cfo + pc * 32, xq1[jt] + cfo + pc * 32);
    };
    load_b(0, bb[0]);
    {
      const unsigned char* q = Dt + (s & 1) * W4_BLK + (32 * half + lane_row) * W4_ROWB + lane_col;
#pragma unroll
      for (int pc = 2; pc >= 0; --pc)                    // piece 2 first: the first MFMAs use it
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) a[pc][mf] = w4_tr(q + mf * 96 + pc * 32, q + mf * 96 + pc * 32 + 16 * W4_ROWB);
    }
#pragma unroll
    for (int f = 0; f < 7; ++f) {
      if (f == 6 && nfr == 6) break;
      if (f + 1 < 7 && (f + 1 < 6 || nfr == 7)) load_b(f + 1, bb[(f + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 (&b)[3] = bb[f & 1];
#if W4_ABL == 2
#define W4_MMA(qa, qb) _Pragma("unroll") for (int mf = 0; mf < 3; ++mf) asm volatile("" ::"v"(a[qa][mf]), "v"(b[qb]));
#else
#define W4_MMA(qa, qb)                                                                                      \
  _Pragma("unroll") for (int mf = 0; mf < 3; ++mf) acc[mf][f] =                                             \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa][mf], b[qb], acc[mf][f], 0, 0, 0);
#endif
      W4_MMA(2, 0) W4_MMA(0, 2) W4_MMA(1, 1) W4_MMA(1, 0) W4_MMA(0, 1) W4_MMA(0, 0)
#undef W4_MMA
      __builtin_amdgcn_sched_barrier(0);
    }
    sbase += W4_KB;
    if (sbase >= R) sbase -= R;
    w4_dma_wait();         // this wave's pieces of stage s + 1 have landed ...
    if (W4_ABL != 1) __syncthreads();       // ... everybody's have, and nobody reads stage s any more
  }

  // the two halves' accumulators meet in LDS (the tiles are dead): half 1 stores, half 0 adds and writes the slab
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  if (half == 1) {
#pragma unroll
    for (int mf = 0; mf < 3; ++mf)
#pragma unroll
      for (int j = 0; j < 7; ++j) red[((wq * 3 + mf) * 7 + j) * 64 + lane] = acc[mf][j];
  }
  __syncthreads();
  if (half == 0 && W4_ABL != 5) {
    float* outp = p.part + (size_t)blockIdx.z * p.Co * 9 * p.Ci;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (j == 6 && wq == 3) break;
      const int tap = j < 6 ? 2 * wq + j / 3 : 8;
      const int cf = j < 6 ? j % 3 : wq;
      const int ci = ci0 + cf * 16 + t16;
#pragma unroll
      for (int mf = 0; mf < 3; ++mf) {
        const f32x4 v = acc[mf][j] + red[((wq * 3 + mf) * 7 + j) * 64 + lane];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int co = co0 + mf * 16 + g * 4 + rg;
          outp[((size_t)co * 9 + tap) * p.Ci + ci] = v[rg];
        }
      }
    }
  }
}

// slab reduction (as wg3_reduce_kernel of conv3x3_wgrad.hip): 16 float4 columns x 16 split-lanes per workgroup
__global__ __launch_bounds__(256) void wg4_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long n,
                                                         int nsplit, int accumulate) {
  __shared__ f32x4 sm[16][16];
  const long n4 = n >> 2;
  const int col = threadIdx.x & 15, zl = threadIdx.x >> 4;
  for (long base = (long)blockIdx.x * 16; base < n4; base += (long)gridDim.x * 16) {
    const long i = base + col;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (i < n4) {
      const f32x4* src = reinterpret_cast<const f32x4*>(part) + i;
      int zz = zl;
      for (; zz + 48 < nsplit; zz += 64) {
        s0 += src[(long)zz * n4];
        s1 += src[(long)(zz + 16) * n4];
        s2 += src[(long)(zz + 32) * n4];
        s3 += src[(long)(zz + 48) * n4];
      }
      for (; zz < nsplit; zz += 16) s0 += src[(long)zz * n4];
    }
    sm[zl][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zl == 0 && i < n4) {
      f32x4 s = sm[0][col];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += sm[k][col];
      if (accumulate) s += reinterpret_cast<const f32x4*>(out)[i];
      reinterpret_cast<f32x4*>(out)[i] = s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- host ----
struct WG4Plan { int nsplit, q, rem, nblk, NB; size_t lds; };

static bool wg4_plan(int N, int H, int W, int Ci, int Co, WG4Plan* pl) {
  if (N <= 0 || H < 1 || W < 2 || W + 2 > W4_MAX_SW || Ci <= 0 || Co <= 0 || Ci % 48 != 0 || Co % 48 != 0) return false;
  const long P = x6p_positions(N, H, W);
  if ((P + X6P_GB + X6P_GA) * (long)(Ci > Co ? Ci : Co) * 6 >= 2147483647L) return false;
  const long pairs = (long)(Co / 48) * (Ci / 48);
  const long stages = (P + W4_KB - 1) / W4_KB;
  long want = (256 + pairs - 1) / pairs;          // one workgroup per CU
  if (want > stages) want = stages;
  if (want < 1) want = 1;
  pl->nsplit = (int)want;
  pl->q = (int)(stages / want);
  pl->rem = (int)(stages % want);
  pl->nblk = (W4_KB + 2 * (W + 3) + W4_KB - 1) / W4_KB;
  pl->NB = pl->nblk + 1;
  pl->lds = (size_t)(2 + pl->NB) * W4_BLK;
  if (pl->lds < (size_t)4 * 21 * 1024) pl->lds = (size_t)4 * 21 * 1024;      // the cross-half reduction buffer
  return pl->lds <= 160 * 1024;
}

extern "C" int buctd_conv3x3_wgrad_bf16x6_p_supported(int N, int H, int W, int Ci, int Co) {
  WG4Plan pl;
  return wg4_plan(N, H, W, Ci, Co, &pl) ? 1 : 0;
}
extern "C" size_t buctd_conv3x3_wgrad_bf16x6_p_workspace(int N, int H, int W, int Ci, int Co) {
  WG4Plan pl;
  if (!wg4_plan(N, H, W, Ci, Co, &pl)) return 0;
  return (size_t)pl.nsplit * Co * 9 * Ci * sizeof(float);
}

/* dw[Co][3][3][Ci] (+)= the weight gradient from the two operands as x6 planes (allocation bases, buctd_x6p_bytes):
 * x_planes of the convolution input [N][H][W][Ci], dy_planes of the output gradient [N][H][W][Co]. */
extern "C" __attribute__((visibility("default"))) int buctd_conv3x3_wgrad_bf16x6_p(int N, int H, int W, int Ci, int Co, const void* x_planes,
                                            const void* dy_planes, float* dw, int accumulate, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  WG4Plan pl;
  BUCTD_CHECK_ARG(x_planes && dy_planes && dw, "buctd_conv3x3_wgrad_bf16x6_p: null tensor pointer");
  BUCTD_CHECK_ARG(wg4_plan(N, H, W, Ci, Co, &pl), "buctd_conv3x3_wgrad_bf16x6_p: unsupported shape N%d H%d W%d Ci%d Co%d", N,
                  H, W, Ci, Co);
  const size_t need = (size_t)pl.nsplit * Co * 9 * Ci * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_conv3x3_wgrad_bf16x6_p: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  static bool attr_set = false;     // idempotent attribute call: a race at first use only repeats it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad4_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      buctd_set_error("buctd_conv3x3_wgrad_bf16x6_p: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return BUCTD_ELAUNCH;
    }
    attr_set = true;
  }
  WG4Args a;
  a.xp = (const unsigned char*)x_planes + x6p_row0(Ci);
  a.dp = (const unsigned char*)dy_planes + x6p_row0(Co);
  a.part = (float*)workspace;
  a.Ci = Ci; a.Co = Co; a.SW = W + 2;
  a.split_q = pl.q; a.split_rem = pl.rem; a.nblk = pl.nblk; a.NB = pl.NB;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_wgrad4_kernel, dim3(Co / 48, Ci / 48, pl.nsplit), dim3(512), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x6_p");
  const long n = (long)Co * 9 * Ci;
  int blocks = ceil_div(n / 4, 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wg4_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, dw, n, pl.nsplit, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x6_p (reduce)");
  return BUCTD_OK;
}
