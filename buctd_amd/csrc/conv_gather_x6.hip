// The convolutions around the BasicBlocks in the bf16x6 arithmetic of conv3x3.hip (fp32 operands split exactly into three
// bf16 pieces, six v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate): the stride-2 3x3 convolutions of the HRNet
// transitions and fuse layers (reference lib/models/pose_hrnet.py:187-245, 338-372), the 1x1 convolutions of the fuse
// layers and of the layer1 Bottlenecks (pose_hrnet.py:60-108), forward and data gradient.  Until round 3 these ran on the
// exact-fp32 MFMA (conv.hip, 157 TFLOP/s peak, 25-45 % of it on these shapes).
//
// A stride-2 window is not a shifted window of a dense tile, so this is a GATHERED implicit GEMM, one kernel for all of them:
//   * the OUTPUT pixels of a launch form a grid Hg x Wg per image, addressed in the zero-padded flattened position space of
//     conv3x3.hip (so that the epilogue - bias, Welford BatchNorm partials + valid-row counts, eval-BN, residual, ReLU,
//     coalesced stores - is literally c3_epilogue);
//   * the contraction runs over (tap, 16-channel chunk) "half-steps", two per K = 32 MFMA step (lanes 0-31 / 32-63 of the A
//     operand); tap t of grid pixel (y, x) reads source pixel (y * ss + dy_t, x * ss + dx_t), zero outside the source:
//       forward, stride 2 : grid = Ho x Wo, ss = 2, nine taps (r - 1, s - 1);
//       1x1               : grid = H x W, ss = 1, one tap (0, 0);
//       data gradient of a stride-2 conv: FOUR parity classes in blockIdx.z - class (a, b) owns the input pixels
//         (2u + a, 2v + b), grid = H/2 x W/2, source = dy, ss = 1, and only the taps that reach that parity (1, 2, 2 or 4 of
//         the nine: rows a = 0 -> r = 1 at dy row u; a = 1 -> r = 0 at u + 1 and r = 2 at u), written through the epilogue's
//         output map (pixel (y, x) of the class grid -> (2y + a, 2x + b));
//   * a workgroup = 256 threads owns BM consecutive positions x BN output channels; per step every thread gathers its 16-byte
//     pieces of the two half-step tiles into registers TWO steps ahead, splits them into LDS rows [16 h | 16 m | 16 l]
//     (the A-row format of conv3x3.hip) and the four waves multiply; the weight fragments come straight from the prepared
//     image in L2 ([step][output channel][32 h | 32 m | 32 l k-slots], built once per weight update by buctd_gconv_x6_prep),
//     one step ahead.
// VALU per MFMA: 16 gathered floats per thread and step x ~4 instructions against 72 MFMAs per wave (BN = 96) - under one.
#include "c3_lean.h"
#include <string.h>

#define GC_MAXT 9
// the train-mode epilogue's LDS (c3_lean.h): staging / reduction / exchange at the front of smem (C3_EPI_LDS), its column table behind
#define GC_TAB_OFF (56 * 1024)

struct GcClass {
  int ntaps, nhs, nsteps;            // taps, half-steps = ntaps * (SC / 16), steps = ceil(nhs / 2)
  int oy0, ox0;                      // output map offsets of the class
  long wp_off;                       // byte offset of the class's weight image
  unsigned pdy, pdx;                 // source offsets of the taps, 2 bits each: (offset + 1) << (2 * tap) - scalar arithmetic in
                                     // the kernel (an indexed kernarg array became a vector load + vmcnt(0) in the step loop)
};

struct GcArgs {
  C3Args e;                          // epilogue view: grid geometry (N, H = Hg, W = Wg, SW, IB, P), Co = output channels ...
  const float* src;                  // [N][SH][SWd][SC]
  const unsigned char* wp;
  int SH, SWd, SC, ss, cpt;          // cpt = SC / 16 chunks per tap
  int ncls;
  GcClass cls[4];
};

// MODE: -1 = the general epilogue (bias, eval-mode scale / shift, ReLU, partial-sum statistics), else an option set of the
// train step as a template argument (c3_lean.h: 0 plain, C3M_STATS forward with the statistics accumulator, C3M_RES data gradient
// with a skip gradient) - the straight-line epilogue of the 3x3 train-mode kernels.
template <int MF, int NF, int WM, int WN, int MODE>
__global__ __launch_bounds__(256, 2) void gconv_x6_kernel(GcArgs p) {
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16, ROWB = 96, PST = 32, BROW = 192;
  constexpr int QA = BM / 64;                           // rows per thread and half-step (thread = row t >> 2 + 64 q, float4 t & 3)
  static_assert(WM * WN == 4 && BM % 64 == 0, "four waves, BM a multiple of 64");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // A: [2 stages][2 half-steps][BM][96 B]; then the epilogue's
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int bx = blockIdx.x, by = blockIdx.y;
  const GcClass& c = p.cls[blockIdx.z];
  C3Args e = p.e;
  e.oy0 = c.oy0; e.ox0 = c.ox0;
  const int p0 = bx * BM, n0 = by * BN;

  // rows this thread gathers: grid pixel -> element offset of the source pixel of tap offset (0, 0), and ONE bit per tap:
  // that tap's source pixel exists.  The step loop then needs an add and a bit test per 16-byte piece (counters of the first
  // version: 6.3 VALU instructions per MFMA, most of them this address and bounds arithmetic redone for every piece).
  // WN == 1 (48- / 64-wide outputs): a wave multiplies ITS MF * 16 rows by all BN columns, so it stages exactly those rows
  // itself (16 per pass) and the step loop needs no workgroup barrier at all - LDS operations of one wave execute in order.
  // WN == 2: the two waves of a row share the rows; the workgroup stages them together (64 per pass) behind a barrier.
  constexpr bool PRIV = WN == 1;
  constexpr int RSTEP = PRIV ? 16 : 64;
  static_assert(!PRIV || MF * 16 / 16 == QA, "wave-private staging: MF passes of 16 rows == BM / 64");
  // (branch-free: every pass in ONE basic block - the divisions of the passes interleave; a row outside the grid has no taps)
  unsigned rowbase[QA];        // BYTE offset of the source pixel of tap offset (0, 0), this thread's channel slot included
  unsigned tapok[QA];
  const int arow = PRIV ? wave * (MF * 16) + (lane >> 2) : (t >> 2), c4 = t & 3;
  const __amdgpu_buffer_rsrc_t r_src = c3_rsrc(p.src, (unsigned)e.N * p.SH * p.SWd * p.SC * 4u);
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int pp = p0 + arow + RSTEP * q;
    const int n = fast_div(pp, e.ib_mul, e.ib_sh);
    const int rem = pp - n * e.IB;
    const int yy = fast_div(rem, e.sw_mul, e.sw_sh), xx = rem - yy * e.SW;
    const bool real = (pp < e.P) & (n < e.N) & (yy >= 1) & (xx >= 1) & (xx <= e.W);
    const int sy0 = (yy - 1) * p.ss, sx0 = (xx - 1) * p.ss;
    rowbase[q] = (unsigned)(((n * p.SH + sy0) * p.SWd + sx0) * p.SC + c4 * 4) * 4u;
    unsigned ok = 0u;
    for (int tp = 0; tp < c.ntaps; ++tp) {
      const int sy = sy0 + (int)((c.pdy >> (2 * tp)) & 3u) - 1, sx = sx0 + (int)((c.pdx >> (2 * tp)) & 3u) - 1;
      ok |= (((unsigned)sy < (unsigned)p.SH) & ((unsigned)sx < (unsigned)p.SWd) ? 1u : 0u) << tp;
    }
    tapok[q] = real ? ok : 0u;
  }
  const unsigned char* wbase = p.wp + c.wp_off + (size_t)(n0 + wave_n * NF * 16 + i16) * BROW + g * 16;
  const size_t wstep = (size_t)e.Co * BROW;

  // the gathered A pieces travel TWO steps ahead of their MFMAs through two register sets (a step's 72 MFMAs per wave cover
  // ~0.5 us, a gathered L2 / HBM round trip under load takes 1-2 us), the weight fragments (hot in L2) one step ahead
  f32x4 areg[2][2][QA];                     // [set][half-step][row]
  bf16x8 bnx[2][3][NF];                     // weight fragments: set st & 1 is multiplied, the other one travels
  int tap = 0, chunk = 0;                   // (tap, chunk) of the next half-step to load
  // byte offset of the current tap relative to tap offset (0, 0) - scalar
  auto tap_delta = [&](int tp) -> int {
    const int dy = (int)((c.pdy >> (2 * tp)) & 3u) - 1, dx = (int)((c.pdx >> (2 * tp)) & 3u) - 1;
    return (dy * p.SWd + dx) * p.SC * 4;
  };
  int cur_delta = tap_delta(0);
  // Every load is unconditional: a piece outside the source (or past the last half-step) carries an out-of-range offset, the
  // buffer load returns zeros for it - no select when the piece is split, no mask to carry, 32-bit address arithmetic
  auto load_a = [&](int st, f32x4 (&dst)[2][QA]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool live = 2 * st + h < c.nhs;
      const unsigned d = (unsigned)(cur_delta + chunk * 64);
      const unsigned tbit = live ? (1u << tap) : 0u;
#pragma unroll
      for (int q = 0; q < QA; ++q) dst[h][q] = c3_bload(r_src, (tapok[q] & tbit) ? rowbase[q] + d : C3_OOB);
      if (live && ++chunk == p.cpt) {
        chunk = 0;
        ++tap;
        cur_delta = tap_delta(tap);
      }
    }
  };
  auto load_b = [&](int st, bf16x8 (&dst)[3][NF]) {
    const unsigned char* wp = wbase + (size_t)st * wstep;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) dst[q][nf] = *reinterpret_cast<const bf16x8*>(wp + (size_t)nf * 16 * BROW + q * 64);
  };
  constexpr int ABUF = 2 * BM * ROWB;       // one A stage: two half-step tiles
  auto store_a = [&](const f32x4 (&src)[2][QA], int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < QA; ++q)
        split_store_pk<3, PST>(smem + (size_t)buf * ABUF + ((size_t)h * BM + arow + RSTEP * q) * ROWB, c4 * 4, src[h][q]);
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this lane's A fragment bytes: half-step g >> 1, channels 8 (g & 1) .. + 7 of row wave_m * MF * 16 + mf * 16 + i16
  const unsigned char* abase = smem + ((size_t)(g >> 1) * BM + wave_m * MF * 16 + i16) * ROWB + (g & 1) * 16;

  // LDS holds two A stages: while the four waves multiply stage st out of buffer st & 1, they split and store stage st + 1
  // (in registers since the previous step) into the other one - ONE barrier per step, and the store phase (VALU + ds_write)
  // has MFMAs beside it (with a single buffer and two barriers it took 40 % of the step, scratch/gc_abl.sh).
  auto step = [&](int st, auto set) {
    constexpr int SET = decltype(set)::value;
    bf16x8 (&bc)[3][NF] = bnx[SET];
    // unconditional on purpose (past the end: dummy pieces, the last weight step again): a skipped load is a merge of old
    // and new register contents, which the compiler resolves with copies that wait for the loads just issued
    load_a(st + 2, areg[SET]);        // this set's stage st went to LDS during the previous step
    load_b(st + 1 < c.nsteps ? st + 1 : st, bnx[1 - SET]);
    const unsigned char* ab = abase + SET * ABUF;
    bf16x8 a[2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[0][q] = *reinterpret_cast<const bf16x8*>(ab + q * PST);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (mf + 1 < MF) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
          a[(mf + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(ab + (size_t)(mf + 1) * 16 * ROWB + q * PST);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 (&ac)[3] = a[mf & 1];
#define GC_MMA(qa, qb) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[qa], bc[qb][nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        GC_MMA(2, 0) GC_MMA(0, 2) GC_MMA(1, 1) GC_MMA(1, 0) GC_MMA(0, 1) GC_MMA(0, 0)
      }
#undef GC_MMA
      // a quarter of the next stage's split + store behind each fragment's MFMAs
      if (mf < 2 * QA && mf < MF) {
        const int h = mf / QA, q = mf % QA;
        split_store_pk<3, PST>(smem + (size_t)(1 - SET) * ABUF + ((size_t)h * BM + arow + RSTEP * q) * ROWB, c4 * 4,
                               areg[1 - SET][h][q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (2 * QA > MF) {          // pieces that found no fragment slot
#pragma unroll
      for (int i = MF; i < 2 * QA; ++i) {
        const int h = i / QA, q = i % QA;
        split_store_pk<3, PST>(smem + (size_t)(1 - SET) * ABUF + ((size_t)h * BM + arow + RSTEP * q) * ROWB, c4 * 4,
                               areg[1 - SET][h][q]);
      }
    }
    if (!PRIV) __syncthreads();        // stage st + 1 is complete, stage st is read
  };

  load_a(0, areg[0]);
  load_b(0, bnx[0]);
  load_a(1, areg[1]);
  store_a(areg[0], 0);
  if (!PRIV) __syncthreads();
  for (int st = 0; st < c.nsteps; st += 2) {
    step(st, IC<0>{});
    if (st + 1 < c.nsteps) step(st + 1, IC<1>{});
  }
  __syncthreads();
  if constexpr (MODE < 0) c3_epilogue<MF, NF, WM, WN>(e, acc, smem, bx, by, p0, n0);
  else c3l_epilogue<MF, NF, WM, WN, MODE>(e, acc, smem, reinterpret_cast<float*>(smem + GC_TAB_OFF), p0, n0);
}

// ---- prepared weight images --------------------------------------------------------------------------------------------
// w is [Co][R][S][Ci] (the engine's filter layout).  dir 0 (forward): rows = Co, contraction = (tap, ci);
// dir 1 (data gradient): rows = Ci, contraction = (tap, co).
struct GcPrep {
  const float* w;
  unsigned char* out;
  int Co, Ci, R, S, dir;
  int ncls;
  int ntaps[4], nsteps[4];
  long off[4];                       // byte offsets of the class images; off[ncls] is not needed
  int tr[4][GC_MAXT], ts[4][GC_MAXT];
  long items[4];                     // cumulative (step, row, k-slot) counts
};

__device__ __forceinline__ void gc_prep_body(const GcPrep& p) {
  const int nrows = p.dir ? p.Ci : p.Co, kc = p.dir ? p.Co : p.Ci, cpt = kc / 16;
  const long total = p.items[p.ncls - 1];
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int cl = 0;
    while (idx >= p.items[cl]) ++cl;
    const long li = idx - (cl ? p.items[cl - 1] : 0);
    const int ks = (int)(li & 31);
    const long sr = li >> 5;
    const int row = (int)(sr % nrows), st = (int)(sr / nrows);
    const int hs = 2 * st + (ks >> 4);
    float v = 0.f;
    if (hs < p.ntaps[cl] * cpt) {
      const int tp = hs / cpt, k = (hs - tp * cpt) * 16 + (ks & 15);
      const int r = p.tr[cl][tp], s = p.ts[cl][tp];
      const int co = p.dir ? k : row, ci = p.dir ? row : k;
      v = p.w[(((long)co * p.R + r) * p.S + s) * p.Ci + ci];
    }
    unsigned char* o = p.out + p.off[cl] + ((size_t)st * nrows + row) * 192 + ks * 2;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __bf16 h = (__bf16)v;
      *reinterpret_cast<unsigned short*>(o + q * 64) = __builtin_bit_cast(unsigned short, h);
      v -= (float)h;
    }
  }
}

__global__ __launch_bounds__(256) void gconv_x6_prep_kernel(GcPrep p) { gc_prep_body(p); }

// every image of a model in one launch (after the optimizer step): blockIdx.y = item of a device table of GcPrep
__global__ __launch_bounds__(256) void gconv_x6_prep_batched_kernel(const GcPrep* __restrict__ items) {
  gc_prep_body(items[blockIdx.y]);
}

// kind 1: 1x1 stride 1 pad 0; kind 2: 3x3 stride 2 pad 1
static bool gc_kind_ok(int kind) { return kind == 1 || kind == 2; }

// tap lists of one (kind, dir, class): (r, s) of the filter and the source offset (dy, dx)
static int gc_taps(int kind, int dir, int cls, int* tr, int* ts, int* tdy, int* tdx) {
  if (kind == 1) { tr[0] = ts[0] = 0; tdy[0] = tdx[0] = 0; return 1; }
  if (!dir) {
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) { tr[r * 3 + s] = r; ts[r * 3 + s] = s; tdy[r * 3 + s] = r - 1; tdx[r * 3 + s] = s - 1; }
    return 9;
  }
  const int a = cls >> 1, b = cls & 1;
  // parity a = 0: filter row 1 at dy row u; a = 1: filter row 0 at u + 1, filter row 2 at u
  const int rl[2][2] = {{1, -1}, {0, 2}}, dl[2][2] = {{0, 0}, {1, 0}}, nl[2] = {1, 2};
  int n = 0;
  for (int i = 0; i < nl[a]; ++i)
    for (int j = 0; j < nl[b]; ++j) {
      tr[n] = rl[a][i]; ts[n] = rl[b][j]; tdy[n] = dl[a][i]; tdx[n] = dl[b][j];
      ++n;
    }
  return n;
}

static int gc_ncls(int kind, int dir) { return (kind == 2 && dir) ? 4 : 1; }

static size_t gc_class_bytes(int ntaps, int kc, int nrows) {
  const int nhs = ntaps * (kc / 16);
  return (size_t)((nhs + 1) / 2) * nrows * 192;
}

extern "C" size_t buctd_gconv_x6_prep_bytes(int kind, int Ci, int Co, int dir) {
  if (!gc_kind_ok(kind) || Ci <= 0 || Co <= 0 || Ci % 16 || Co % 16) return 0;
  const int kc = dir ? Co : Ci, nrows = dir ? Ci : Co;
  size_t total = 0;
  int tr[GC_MAXT], ts[GC_MAXT], ty[GC_MAXT], tx[GC_MAXT];
  for (int c = 0; c < gc_ncls(kind, dir); ++c) total += gc_class_bytes(gc_taps(kind, dir, c, tr, ts, ty, tx), kc, nrows);
  return total;
}

static int gc_prep_fill(int kind, int Ci, int Co, const float* w, int dir, void* wprep, GcPrep* p, long* items_out) {
  BUCTD_CHECK_ARG(w && wprep, "buctd_gconv_x6_prep: null pointer");
  BUCTD_CHECK_ARG(buctd_gconv_x6_prep_bytes(kind, Ci, Co, dir) > 0, "buctd_gconv_x6_prep: unsupported kind %d Ci%d Co%d", kind, Ci,
                  Co);
  memset(p, 0, sizeof(GcPrep));
  p->w = w; p->out = (unsigned char*)wprep; p->Co = Co; p->Ci = Ci; p->R = p->S = kind == 1 ? 1 : 3; p->dir = dir;
  p->ncls = gc_ncls(kind, dir);
  const int kc = dir ? Co : Ci, nrows = dir ? Ci : Co;
  long off = 0, items = 0;
  int ty[GC_MAXT], tx[GC_MAXT];
  for (int c = 0; c < p->ncls; ++c) {
    p->ntaps[c] = gc_taps(kind, dir, c, p->tr[c], p->ts[c], ty, tx);
    p->nsteps[c] = (p->ntaps[c] * (kc / 16) + 1) / 2;
    p->off[c] = off;
    off += (long)gc_class_bytes(p->ntaps[c], kc, nrows);
    items += (long)p->nsteps[c] * nrows * 32;
    p->items[c] = items;
  }
  *items_out = items;
  return BUCTD_OK;
}

extern "C" int buctd_gconv_x6_prep(int kind, int Ci, int Co, const float* w, int dir, void* wprep, void* stream) {
  GcPrep p;
  long items;
  const int rc = gc_prep_fill(kind, Ci, Co, w, dir, wprep, &p, &items);
  if (rc) return rc;
  long blocks = (items + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gconv_x6_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  BUCTD_CHECK_LAUNCH("buctd_gconv_x6_prep");
  return BUCTD_OK;
}

extern "C" size_t buctd_gconv_x6_prep_item_bytes(void) { return sizeof(GcPrep); }

extern "C" int buctd_gconv_x6_prep_item(int kind, int Ci, int Co, const float* w, int dir, void* wprep, void* item_host) {
  BUCTD_CHECK_ARG(item_host, "buctd_gconv_x6_prep_item: null pointer");
  long items;
  return gc_prep_fill(kind, Ci, Co, w, dir, wprep, (GcPrep*)item_host, &items);
}

extern "C" int buctd_gconv_x6_prep_batched(const void* items_device, int n, void* stream) {
  BUCTD_CHECK_ARG(items_device && n > 0 && n <= 65535, "buctd_gconv_x6_prep_batched: bad argument");
  hipLaunchKernelGGL(gconv_x6_prep_batched_kernel, dim3(48, (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                     (const GcPrep*)items_device);
  BUCTD_CHECK_LAUNCH("buctd_gconv_x6_prep_batched");
  return BUCTD_OK;
}

// ---- launch ----------------------------------------------------------------------------------------------------------------
struct GcPlan { int MF, NF, WM, WN, BM, BN; };

static bool gc_plan(int nout, long P, GcPlan* pl) {
  // 96-wide tiles gather every source piece once per 96 outputs; the barrier-free 48-wide ones would win only where the
  // contraction is long and the source tiny (192 -> 384 at 24 x 18: 79 -> 55 us) and lose elsewhere (48 -> 96 at 96 x 72: 43 -> 57)
  if (nout % 96 == 0) *pl = {4, 3, 2, 2, 128, 96};
  // 128-wide tiles where they still make a full round of workgroups (the 1x1 convolutions of layer1 on 256 channels: every
  // 64-wide column tile gathers and splits the same rows again - 64 -> 256 at 96 x 72: 146 -> 113 us, the data gradient of
  // 256 -> 64: 115 -> 94); small maps keep the 64-wide tiles (128 -> 256 at 16 x 12: 15 -> 17-21 us with the wide ones)
  else if (nout % 128 == 0 && ((P + 127) / 128) * (nout / 128) >= 512) *pl = {4, 4, 2, 2, 128, 128};
  else if (nout % 64 == 0) *pl = {2, 4, 4, 1, 128, 64};
  else if (nout % 48 == 0) *pl = {2, 3, 4, 1, 128, 48};
  else return false;
  return true;
}

// geometry of a launch: grid Hg x Wg, source SH x SWd, channels
static bool gc_geo(int kind, int dir, int N, int H, int W, int Ci, int Co, int* Hg, int* Wg, int* SH, int* SWd, int* SC, int* nout) {
  if (!gc_kind_ok(kind) || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 || Co % 16) return false;
  if (kind == 1) { *Hg = H; *Wg = W; *SH = H; *SWd = W; }
  else {
    if ((H & 1) || (W & 1)) return false;
    *Hg = H / 2; *Wg = W / 2;
    if (!dir) { *SH = H; *SWd = W; } else { *SH = H / 2; *SWd = W / 2; }
  }
  *SC = dir ? Co : Ci;
  *nout = dir ? Ci : Co;
  const long P = (long)N * (*Hg + 1) * (*Wg + 2) + *Wg + 2;
  const long big = (long)N * H * W * (Ci > Co ? Ci : Co);
  GcPlan pl;
  return gc_plan(*nout, P, &pl) && P < 2147483647L && big * 4 < 2147483647L;      // (32-bit byte offsets below the out-of-range mark 2^31)
}

extern "C" int buctd_gconv_x6_supported(int kind, int N, int H, int W, int Ci, int Co, int dir) {
  int a, b, c, d, e, f;
  return gc_geo(kind, dir, N, H, W, Ci, Co, &a, &b, &c, &d, &e, &f) ? 1 : 0;
}

extern "C" int buctd_gconv_x6_stats_groups(int kind, int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group) {
  int Hg, Wg, SH, SWd, SC, nout;
  BUCTD_CHECK_ARG(ngroups && rows_per_group && gc_geo(kind, 0, N, H, W, Ci, Co, &Hg, &Wg, &SH, &SWd, &SC, &nout),
                  "buctd_gconv_x6_stats_groups: unsupported shape");
  GcPlan pl;
  const long P = (long)N * (Hg + 1) * (Wg + 2) + Wg + 2;
  gc_plan(nout, P, &pl);
  *ngroups = (int)((P + pl.BM - 1) / pl.BM) * pl.WM;
  *rows_per_group = pl.MF * 16;
  return BUCTD_OK;
}

template <int MF, int NF, int WM, int WN>
static int gc_launch(const GcArgs& a, int tiles, int ncol, hipStream_t st, int mode) {
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16, LD = NF * 16 + 4, EP = MF >= 2 ? 2 : 1;
  constexpr size_t tile = (size_t)2 * 2 * BM * 96, epi = (size_t)4 * EP * 16 * LD * 4 + 4 * 128 * 4;
  constexpr size_t lds = tile > epi ? tile : epi;
  static_assert(lds <= GC_TAB_OFF && C3_EPI_LDS <= GC_TAB_OFF && GC_TAB_OFF + 4 * BN * 4 <= 64 * 1024, "gconv_x6: static LDS budget");
  const dim3 grid(tiles, ncol, a.ncls);
  if (mode == C3M_STATS) hipLaunchKernelGGL((gconv_x6_kernel<MF, NF, WM, WN, C3M_STATS>), grid, dim3(256), GC_TAB_OFF + 4 * BN * 4, st, a);
  else if (mode == C3M_RES) hipLaunchKernelGGL((gconv_x6_kernel<MF, NF, WM, WN, C3M_RES>), grid, dim3(256), GC_TAB_OFF + 4 * BN * 4, st, a);
  else if (mode == 0) hipLaunchKernelGGL((gconv_x6_kernel<MF, NF, WM, WN, 0>), grid, dim3(256), GC_TAB_OFF + 4 * BN * 4, st, a);
  else hipLaunchKernelGGL((gconv_x6_kernel<MF, NF, WM, WN, -1>), grid, dim3(256), lds, st, a);
  return BUCTD_OK;
}

// ---- 1x1 convolutions on large maps: a row-streaming kernel ------------------------------------------------------------------
// The 1x1 convolutions of layer1 (64 <-> 256 channels at 1/4 resolution: pose_hrnet.py:60-98) move 283 MB for 7 GFLOP - they are
// HBM-side (35 us), and the tiled kernel above spends its time in per-tile prologues and epilogues (K = 64: two steps per tile;
// 100-115 us).  Here a wavefront streams 32-row blocks of the [rows][Ci] input:
//   * its A fragments come STRAIGHT from global memory in MFMA layout (lane (i16, g) of a K = 32 step needs channels 8 g .. 8 g + 7
//     of row i16: 32 contiguous bytes), are split into the three bf16 pieces in registers - no LDS round trip for the input, every
//     element is read and split exactly once per launch whatever the number of output columns;
//   * the whole prepared filter image ([step][Co][32 h | 32 m | 32 l], <= 104 KB, rows padded to 208 B: conflict-free
//     ds_read_b128) sits in LDS for the lifetime of the persistent workgroup (8 waves, one workgroup per CU);
//   * all Co / 16 column fragments of a block are accumulated in registers (two 16-row fragments x Co / 16 <= 128 registers);
//     the next K-step's A loads travel under the current one's MFMAs;
//   * epilogue per block: bias, BatchNorm statistics as per-lane column sums (folded over the workgroup at the end into the
//     integer accumulator of bn_acc.h); the tile leaves through a per-wave LDS staging slice, 16 rows x 64 columns at a time,
//     so that the stores and the residual reads are 16-byte pieces of contiguous runs (scalar accesses in accumulator
//     layout: 143 / 495 us for the data gradients with a residual instead of 60 / 120).
struct R1Args {
  const float* x;            // [rows][Ci]
  const unsigned char* wp;   // gathered-kernel image of the direction: [Ci / 32 steps][Co][192 B]
  const float* bias;
  const float* res;          // [rows][Co] or null
  float* out;                // [rows][Co]
  long long* stats_acc;      // or null
  long rows;
  int Ci, Co, nblocks;       // nblocks = ceil(rows / 32)
};

#define R1_WROW 208          // LDS stride of an image row (192 B + 16: lanes i16 = 0..15 land in 16 distinct 16-byte bank groups)
#define R1_LD 68             // staging row stride in floats (64 columns + 4)
#define R1_STG (16 * R1_LD * 4)      // staging bytes per wave

template <int NFT>           // column fragments: Co / 16
__global__ __launch_bounds__(512, 1) void conv1x1_rows_x6_kernel(R1Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int KS = p.Ci / 32, Co = p.Co;
  // the filter image -> LDS (16-byte pieces; 12 per row)
  {
    const int pieces = KS * Co * 12;
    for (int i = t; i < pieces; i += 512) {
      const int row = i / 12, pc = i - row * 12;
      *reinterpret_cast<f32x4*>(smem + (size_t)row * R1_WROW + pc * 16) =
          *reinterpret_cast<const f32x4*>(p.wp + (size_t)row * 192 + pc * 16);
    }
  }
  __syncthreads();
  const unsigned char* bl = smem + (size_t)i16 * R1_WROW + g * 16;      // this lane's slot inside a 16-row group of the image
  float* stg = reinterpret_cast<float*>(smem + (size_t)KS * Co * R1_WROW + (size_t)wave * R1_STG);

  // BatchNorm statistics of a wave's rows as sums AROUND A PIVOT (the wave's first output row, column by column): a raw fp32
  // sum of squares loses |mean|^2 / var of its digits; converted to raw sums in fp64 at the end (c3_lean.h: c3l_epilogue)
  float s1[NFT], s2[NFT], piv[NFT];
#pragma unroll
  for (int nf = 0; nf < NFT; ++nf) s1[nf] = s2[nf] = piv[nf] = 0.f;
  int nrows = 0;                 // real rows this LANE has added (its four row slots of every fragment)
  bool first = true;

  const int nwaves = gridDim.x * 8;
  int blk = blockIdx.x * 8 + wave;
  // A of (block, k-step): rows r0 + mf * 16 + i16, channels 32 s + 8 g .. + 7
  f32x4 an[2][2];
  auto load_a = [&](int b, int sIdx) {
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      long row = (long)b * 32 + mf * 16 + i16;
      if (row >= p.rows) row = 0;                    // clamped: the values are not stored
      const float* src = p.x + row * p.Ci + sIdx * 32 + g * 8;
      an[mf][0] = *reinterpret_cast<const f32x4*>(src);
      an[mf][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  };
  if (blk < p.nblocks) load_a(blk, 0);
  for (; blk < p.nblocks; blk += nwaves) {
    f32x4 acc[2][NFT];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFT; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sIdx = 0; sIdx < KS; ++sIdx) {
      // this step's A: fp32 -> three bf16 pieces (exact residuals), then the next step's loads go out
      bf16x8 a[3][2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        c3_f32x2 v[4] = {{an[mf][0].x, an[mf][0].y}, {an[mf][0].z, an[mf][0].w}, {an[mf][1].x, an[mf][1].y}, {an[mf][1].z, an[mf][1].w}};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          unsigned w[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            w[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v[j], c3_bf16x2));
            if (q < 2) v[j] -= (c3_f32x2){__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
          }
          typedef unsigned r1_u32x4 __attribute__((ext_vector_type(4)));
          a[q][mf] = __builtin_bit_cast(bf16x8, (r1_u32x4){w[0], w[1], w[2], w[3]});
        }
      }
      if (sIdx + 1 < KS) load_a(blk, sIdx + 1);
      else if (blk + nwaves < p.nblocks) load_a(blk + nwaves, 0);
      const unsigned char* bs = bl + (size_t)sIdx * Co * R1_WROW;
#pragma unroll
      for (int nf = 0; nf < NFT; ++nf) {
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8*>(bs + (size_t)nf * 16 * R1_WROW + q * 64);
#define R1_MMA(qa, qb)                                                                                      \
  _Pragma("unroll") for (int mf = 0; mf < 2; ++mf) acc[mf][nf] =                                            \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa][mf], b[qb], acc[mf][nf], 0, 0, 0);
        R1_MMA(2, 0) R1_MMA(0, 2) R1_MMA(1, 1) R1_MMA(1, 0) R1_MMA(0, 1) R1_MMA(0, 0)
#undef R1_MMA
      }
    }
    // epilogue: accumulator (mf, nf, rg) = row blk * 32 + mf * 16 + g * 4 + rg, column nf * 16 + i16.  Statistics from the
    // registers; the tile leaves through this wave's LDS staging slice, 16 rows x 64 columns at a time, so that every lane
    // stores (and reads the residual as) 16-byte pieces of contiguous runs
    if (first && p.stats_acc) {          // the wave's first block: its first row (always a real one) is the pivot row
      first = false;
#pragma unroll
      for (int nf = 0; nf < NFT; ++nf) piv[nf] = __shfl(acc[0][nf][0] + (p.bias ? p.bias[nf * 16 + i16] : 0.f), i16, 64);
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) nrows += ((long)blk * 32 + mf * 16 + g * 4 + rg < p.rows) ? 1 : 0;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
#pragma unroll
      for (int cg = 0; cg < NFT / 4; ++cg) {
#pragma unroll
        for (int nl = 0; nl < 4; ++nl) {
          const int nf = cg * 4 + nl;
          const float bv = p.bias ? p.bias[nf * 16 + i16] : 0.f;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const float v = acc[mf][nf][rg] + bv;
            stg[(g * 4 + rg) * R1_LD + nl * 16 + i16] = v;
            if ((long)blk * 32 + mf * 16 + g * 4 + rg < p.rows) {
              const float d = v - piv[nf];
              s1[nf] += d;
              s2[nf] = __builtin_fmaf(d, d, s2[nf]);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int item = lane + 64 * k, row = item >> 4, c4 = item & 15;
          const long grow = (long)blk * 32 + mf * 16 + row;
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * R1_LD + c4 * 4);
          if (grow < p.rows) {
            const long o = grow * Co + cg * 64 + c4 * 4;
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
            *reinterpret_cast<f32x4*>(p.out + o) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (p.stats_acc) {
    // lanes -> columns (the four row groups of a fragment), waves -> workgroup (fixed order), one exact integer addition per
    // channel and sum (bn_acc.h); the exchange reuses the image area (nobody reads it any more)
    __syncthreads();
    double2* exch = reinterpret_cast<double2*>(smem);       // [8 waves][Co]
#pragma unroll
    for (int nf = 0; nf < NFT; ++nf) {
      // this lane's rows as raw sums in fp64: sum v = S1 + n pi, sum v^2 = S2 + 2 pi S1 + n pi^2
      const double pi = (double)piv[nf], nn = (double)nrows;
      double a1 = (double)s1[nf] + nn * pi, a2 = (double)s2[nf] + 2.0 * pi * (double)s1[nf] + nn * pi * pi;
      a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
      a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
      if (g == 0) exch[wave * Co + nf * 16 + i16] = make_double2(a1, a2);
    }
    __syncthreads();
    for (int c = t; c < Co; c += 512) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const double2 v = exch[w * Co + c];
        a1 += v.x;
        a2 += v.y;
      }
      bnacc_add(p.stats_acc, Co, bnacc_shard(), c, a1, a2);
    }
  }
}

// shapes the row-streaming kernel takes: 1x1, whole 32-channel steps, 4 / 8 / 16 column fragments, the image (+ the statistics
// exchange) within one CU's LDS, and enough rows to give every wave of the persistent grid a few blocks
static bool r1_ok(int kind, long rows, int cin, int cout) {
  if (kind != 1 || cin % 32 || (cout != 64 && cout != 128 && cout != 256)) return false;
  const size_t lds = (size_t)(cin / 32) * cout * R1_WROW + (size_t)8 * R1_STG;
  return lds <= 150 * 1024 && (size_t)8 * cout * 16 <= (size_t)(cin / 32) * cout * R1_WROW && rows >= 65536;
}

static int r1_run(long rows, int cin, int cout, const float* x, const void* wprep, const float* bias, const float* residual, float* out,
                  long long* stats_acc, hipStream_t st, const char* who) {
  R1Args a;
  a.x = x; a.wp = (const unsigned char*)wprep; a.bias = bias; a.res = residual; a.out = out; a.stats_acc = stats_acc;
  a.rows = rows; a.Ci = cin; a.Co = cout; a.nblocks = (int)((rows + 31) / 32);
  const size_t lds = (size_t)(cin / 32) * cout * R1_WROW + (size_t)8 * R1_STG;
  static unsigned char done[3][BUCTD_MAX_DEVICES] = {{0}};
  void (*fn)(R1Args) = cout == 64 ? conv1x1_rows_x6_kernel<4> : cout == 128 ? conv1x1_rows_x6_kernel<8> : conv1x1_rows_x6_kernel<16>;
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, done[cout == 64 ? 0 : cout == 128 ? 1 : 2], who))
    return rc;
  int grid = 256;
  if (grid * 8 > a.nblocks) grid = (a.nblocks + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, st, a);
  BUCTD_CHECK_LAUNCH(who);
  return BUCTD_OK;
}

static int gc_run(int kind, int dir, int N, int H, int W, int Ci, int Co, const float* src, const void* wprep, const float* bias,
                  const float* scale, const float* shift, const float* residual, int relu, float* out, float* stats_partials,
                  int* stats_counts, void* stream, const char* who, long long* stats_acc = nullptr) {
  int Hg, Wg, SH, SWd, SC, nout;
  BUCTD_CHECK_ARG(src && wprep && out, "%s: null tensor pointer", who);
  BUCTD_CHECK_ARG(gc_geo(kind, dir, N, H, W, Ci, Co, &Hg, &Wg, &SH, &SWd, &SC, &nout),
                  "%s: unsupported shape kind %d N%d H%d W%d Ci%d Co%d", who, kind, N, H, W, Ci, Co);
  BUCTD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "%s: scale and shift go together", who);
  BUCTD_CHECK_ARG((stats_partials == nullptr) == (stats_counts == nullptr), "%s: stats partials and counts go together", who);
  BUCTD_CHECK_ARG(!(dir && stats_partials), "%s: no statistics on the data gradient", who);
  if (!scale && !relu && !stats_partials && r1_ok(kind, (long)N * H * W, SC, nout))
    return r1_run((long)N * H * W, SC, nout, src, wprep, bias, residual, out, stats_acc, (hipStream_t)stream, who);
  GcPlan pl;
  gc_plan(nout, (long)N * (Hg + 1) * (Wg + 2) + Wg + 2, &pl);
  GcArgs a;
  C3Args& e = a.e;
  e.x = nullptr; e.wp = nullptr; e.out = out; e.bias = bias; e.scale = scale; e.shift = shift; e.res = residual;
  e.stats = stats_partials; e.counts = stats_counts;
  e.N = N; e.H = Hg; e.W = Wg; e.Ci = SC; e.Co = nout;
  e.SW = Wg + 2; e.IB = (Hg + 1) * (Wg + 2);
  e.P = (int)((long)N * e.IB + e.SW);
  e.relu = relu; e.na = 0;
  e.in_mean = e.in_invstd = e.in_gamma = e.in_beta = nullptr; e.in_relu = 0; e.col_major = 0;
  e.bs_z = e.bs_y = e.bs_mean = e.bs_invstd = e.bs_gamma = e.bs_beta = nullptr; e.bs_part = nullptr;
  e.stats_acc = stats_acc; e.bs_acc = nullptr;
  memset(&e.in_acc, 0, sizeof(e.in_acc));
  BUCTD_CHECK_ARG(!(dir && stats_acc), "%s: no statistics on the data gradient", who);
  magic_u32((unsigned)e.IB, &e.ib_mul, &e.ib_sh);
  magic_u32((unsigned)e.SW, &e.sw_mul, &e.sw_sh);
  const bool par = kind == 2 && dir;
  e.omap = par ? 1 : 0; e.oH = H; e.oW = W; e.ost = par ? 2 : 1; e.oy0 = e.ox0 = 0;
  a.src = src; a.wp = (const unsigned char*)wprep;
  a.SH = SH; a.SWd = SWd; a.SC = SC; a.ss = (kind == 2 && !dir) ? 2 : 1; a.cpt = SC / 16;
  a.ncls = gc_ncls(kind, dir);
  long off = 0;
  int tr[GC_MAXT], ts[GC_MAXT];
  for (int c = 0; c < 4; ++c) {
    GcClass& k = a.cls[c];
    k.ntaps = k.nhs = k.nsteps = 0; k.oy0 = k.ox0 = 0; k.wp_off = 0; k.pdy = k.pdx = 0u;
    if (c >= a.ncls) continue;
    int tdy[GC_MAXT], tdx[GC_MAXT];
    k.ntaps = gc_taps(kind, dir, c, tr, ts, tdy, tdx);
    k.pdy = k.pdx = 0u;
    for (int t = 0; t < k.ntaps; ++t) {
      k.pdy |= (unsigned)(tdy[t] + 1) << (2 * t);
      k.pdx |= (unsigned)(tdx[t] + 1) << (2 * t);
    }
    k.nhs = k.ntaps * a.cpt;
    k.nsteps = (k.nhs + 1) / 2;
    k.oy0 = par ? (c >> 1) : 0; k.ox0 = par ? (c & 1) : 0;
    k.wp_off = off;
    off += (long)gc_class_bytes(k.ntaps, SC, nout);
  }
  const int tiles = (e.P + pl.BM - 1) / pl.BM, ncol = nout / pl.BN;
  hipStream_t st = (hipStream_t)stream;
  // the option sets of the train step take the straight-line epilogue (tensors below 2 GB: 32-bit byte offsets)
  int mode = -1;
  if (!bias && !scale && !relu && !stats_partials && !(stats_acc && residual) &&
      (long)N * H * W * (Ci > Co ? Ci : Co) * 4 < 2147483648L)
    mode = stats_acc ? C3M_STATS : residual ? C3M_RES : 0;
  if (pl.NF == 3 && pl.WM == 2) gc_launch<4, 3, 2, 2>(a, tiles, ncol, st, mode);
  else if (pl.NF == 4 && pl.WM == 2) gc_launch<4, 4, 2, 2>(a, tiles, ncol, st, mode);
  else if (pl.NF == 4) gc_launch<2, 4, 4, 1>(a, tiles, ncol, st, mode);
  else gc_launch<2, 3, 4, 1>(a, tiles, ncol, st, mode);
  BUCTD_CHECK_LAUNCH(who);
  return BUCTD_OK;
}

extern "C" int buctd_gconv_x6_fwd(int kind, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                  const float* bias, const float* scale, const float* shift, const float* residual, int relu,
                                  float* y, float* stats_partials, int* stats_counts, void* stream) {
  return gc_run(kind, 0, N, H, W, Ci, Co, x, wprep, bias, scale, shift, residual, relu, y, stats_partials, stats_counts, stream,
                "buctd_gconv_x6_fwd");
}

/* forward with the output statistics as an accumulator (bn_acc.h; buctd_bn_acc_bytes(Co) zeroed bytes) instead of partials */
extern "C" int buctd_gconv_x6_fwd_acc(int kind, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                      const float* bias, float* y, void* stats_acc, void* stream) {
  BUCTD_CHECK_ARG(stats_acc, "buctd_gconv_x6_fwd_acc: null accumulator");
  return gc_run(kind, 0, N, H, W, Ci, Co, x, wprep, bias, nullptr, nullptr, nullptr, 0, y, nullptr, nullptr, stream,
                "buctd_gconv_x6_fwd_acc", (long long*)stats_acc);
}

extern "C" int buctd_gconv_x6_dgrad(int kind, int N, int H, int W, int Ci, int Co, const float* dy, const void* wprep,
                                    const float* residual, float* dx, void* stream) {
  return gc_run(kind, 1, N, H, W, Ci, Co, dy, wprep, nullptr, nullptr, nullptr, residual, 0, dx, nullptr, nullptr, stream,
                "buctd_gconv_x6_dgrad");
}
