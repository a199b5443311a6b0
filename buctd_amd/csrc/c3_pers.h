// The persistent form of the train-mode 3x3 convolution (c3_lean.h): a workgroup walks a LIST of tiles and overlaps them in
// itself (reference lib/models/pose_hrnet.py:28-57, 177-185: the BasicBlock convolutions of all branches of a module).
//
// Why.  Ablations of the one-tile-per-workgroup kernel (scratch/patches/r6_c3_lean_ablation_switches.patch, four copies of
// the 48 -> 48 @96x72 convolution in one grid = the sustained rate): 51.7 us per convolution; MFMAs alone 33.5 us (the chip
// clocks ~1.9 GHz under this load: 59.8 k MFMA cycles per SIMD); everything else alone 28 us; removed one at a time: epilogue
// -10.0 (its stores -4.8), input split + LDS stores -6.0, input loads -4.2, weight fragments -1.5, fragment reads -1.1.
// A tile's first loads, its split and its 86 KB of output leave in bursts during which the workgroup multiplies nothing, and
// the other workgroup of the CU covers only ~60 % of the matrix pipe on its own.  Here
//   * the (tile, chunk) sequence of a workgroup is ONE software pipeline: during chunk k the next chunk - of this tile or of
//     the NEXT tile - is split into the other A buffer, and the loads of the chunk after that are in flight; the first MFMA of
//     tile i+1 issues right behind the last output store of tile i, which drains under it;
//   * the epilogue stages through the A buffer the last chunk has just released (the other one already holds tile i+1);
//   * the weight fragments are prefetched across the tile boundary as well;
//   * per-member work (decoding the input BatchNorm from its accumulator) is done once per workgroup, not once per tile.
// Tiles, MFMA order and epilogue arithmetic are those of c3l_tile: bit-identical outputs and sums for the same tile plan.
#pragma once
#include "c3_lean.h"

__host__ __device__ static inline size_t c3p_abytes(int na) {
  const size_t b = (size_t)na * 32 * Geo<3>::ROWB;
  return b < (size_t)C3_EPI_LDS ? (size_t)C3_EPI_LDS : b;
}

// tile j of a member with T tiles -> (bx, by): the XCD digit of a workgroup's tile ids is fixed (its slot and the grid are
// multiples of 8 apart), so every XCD walks a contiguous run of position tiles (halo rows of neighbours meet in its L2)
__device__ __forceinline__ void c3p_tile_xy(const C3Args& p, unsigned j, unsigned T, int gx, int gy, int* bx, int* by) {
  const unsigned xcd = j & 7, idx = j >> 3, per = T >> 3, rem = T & 7;
  const unsigned L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
  if (p.col_major) {
    *by = (int)(L / (unsigned)gx);
    *bx = (int)(L - (unsigned)*by * gx);
  } else {
    *bx = (int)(L / (unsigned)gy);
    *by = (int)(L - (unsigned)*bx * gy);
  }
}

// tiles j0, j0 + G, ... < T of ONE convolution.  smem = [2][arows][ROWB] A buffers | [3][Ci] input-BatchNorm table | [4][BN]
// epilogue table.  Needs Ci >= 32 (two chunks: the pipeline looks two chunks ahead and at most one tile).
template <int MF, int NF, int WM, int WN, int MODE>
__device__ __forceinline__ void c3p_segment(const C3Args& p, unsigned char* smem, unsigned j0, unsigned T, unsigned G, int gx,
                                            int gy) {
  constexpr bool IN_BN = (MODE & C3M_IN_BN) != 0;
  constexpr bool BSR = (MODE & C3M_BS_REBUILD) != 0, BS = BSR || (MODE & C3M_BS_Y) != 0;
  constexpr int ROWB = Geo<3>::ROWB, PST = Geo<3>::PST, CPR = Geo<3>::CPR;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  const int arows = p.na * 32;
  // an A buffer also serves as the epilogue's staging area while the other one holds the next tile: at least C3_EPI_LDS bytes
  const size_t abytes = c3p_abytes(p.na);
  float* bntab = reinterpret_cast<float*>(smem + 2 * abytes);
  float* tab = bntab + 3 * p.Ci;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;

  C3Stager<BM, IN_BN> stg;
  stg.init(p);

  // B fragments: image [step][Co/16][3][64][16 B]; a tile's column offset selects the fragment rows
  const int blane = lane * 16;
  const size_t bstep = (size_t)(p.Co / 16) * 3 * 1024;
  auto b_base = [&](int by) -> const unsigned char* { return p.wp + ((size_t)((by * BN) / 16 + wave_n * NF) * 3) * 1024; };
  constexpr bool BPF2 = MF <= 2 && NF <= 3;     // small wave tiles: fragments two steps ahead (c3x6_tile)
  bf16x8 bc[3][NF], bn[3][NF];
  bf16x8 bn2[BPF2 ? 3 : 1][BPF2 ? NF : 1];
  auto load_b = [&](const unsigned char* src, bf16x8 (&dst)[3][NF]) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int q = 0; q < 3; ++q) dst[q][nf] = *reinterpret_cast<const bf16x8*>(src + (nf * 3 + q) * 1024 + blane);
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Ci / 16, nsteps = nchunks * 5;
  const size_t aoff = (size_t)(wave_m * MF * 16 + i16) * ROWB + (g & 1) * 16;
  const bool lowk = g < 2;
  constexpr int AD = 1;      // fragment prefetch distance (2 costs twelve more registers: spills)
  bf16x8 a[AD + 1][3];
  auto read_a = [&](const unsigned char* abase, int i, bf16x8 (&dst)[3]) {
    const int st = i / MF, mf = i % MF;
    const int tap0 = 2 * st, tap1 = 2 * st + 1 < 9 ? 2 * st + 1 : 2 * st;
    const int o0 = ((tap0 / 3) * p.SW + tap0 % 3) * ROWB, o1 = ((tap1 / 3) * p.SW + tap1 % 3) * ROWB;
    const unsigned char* ap = abase + (lowk ? o0 : o1) + mf * 16 * ROWB;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(ap + q * PST);
  };

  // ---- per-segment set-up: the input BatchNorm table, the first tile's first chunk
  int bx, by;
  c3p_tile_xy(p, j0, T, gx, gy, &bx, &by);
  stg.make(p, bx * BM);
  stg.load(0);
  const unsigned char* bptr = b_base(by);
  load_b(bptr, bn);
  if constexpr (BPF2) load_b(bptr + bstep, bn2);           // nsteps >= 10
  if constexpr (IN_BN) {
    for (int c = t; c < p.Ci; c += 256) {
      const BnFwdStat st = bnacc_fwd_stat(p.in_acc.acc, p.Ci, c, p.in_acc.rows, p.in_acc.eps);
      if (j0 == 0) {            // the workgroup of tile 0 leaves mean / invstd for the backward kernels, updates the running statistics
        p.in_acc.mean_out[c] = st.mean;
        p.in_acc.invstd_out[c] = st.invstd;
        if (p.in_acc.rmean) bnacc_running(st, p.in_acc.rows, p.in_acc.momentum, p.in_acc.rmean, p.in_acc.rvar, c);
      }
      bntab[c] = st.mean;
      bntab[p.Ci + c] = st.invstd * p.in_gamma[c];
      bntab[2 * p.Ci + c] = p.in_beta[c];
    }
    __syncthreads();
  }
  stg.store(p, smem, 0, bntab);
  stg.load(16);
  __syncthreads();

  unsigned kbuf = 0;                  // A buffer of the current chunk
  for (unsigned j = j0; j < T; j += G) {
    const bool has_next = j + G < T;
    const int p0 = bx * BM, n0 = by * BN;
    int bx_n = bx, by_n = by;
    if (has_next) c3p_tile_xy(p, j + G, T, gx, gy, &bx_n, &by_n);
    const unsigned char* bptr_n = b_base(by_n);
    if constexpr (BS) {
      // the column tables of THIS tile's output channels (the previous tile's epilogue ended with a barrier)
      for (int c = t; c < BN; c += 256) {
        const float is = p.bs_invstd[n0 + c];
        tab[c] = p.bs_mean[n0 + c];
        tab[BN + c] = is;
        if constexpr (BSR) {
          tab[2 * BN + c] = is * p.bs_gamma[n0 + c];
          tab[3 * BN + c] = p.bs_beta[n0 + c];
        }
      }
    }
    int gs = 0;
    // fragment address of step gs + d: this tile's image, the next tile's behind it, clamped at the end of the list
    auto b_src = [&](int s) -> const unsigned char* {
      if (s < nsteps) return bptr + (size_t)s * bstep;
      if (has_next) return bptr_n + (size_t)(s - nsteps) * bstep;
      return bptr + (size_t)(nsteps - 1) * bstep;
    };
    for (int ch = 0; ch < nchunks; ++ch) {
      const unsigned char* abase = smem + kbuf * abytes + aoff;
      unsigned char* other = smem + (kbuf ^ 1) * abytes;
      const bool last = ch + 1 == nchunks;
#pragma unroll
      for (int i = 0; i < AD; ++i) read_a(abase, i, a[i]);
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        if (s == 2 && (!last || has_next)) {    // the next chunk (of this tile, or chunk 0 of the next one): registers -> pieces
          stg.store(p, other, last ? 0 : (ch + 1) * 16, bntab);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            bc[q][nf] = bn[q][nf];
            if constexpr (BPF2) bn[q][nf] = bn2[q][nf];
          }
        if constexpr (BPF2) load_b(b_src(gs + 2), bn2);
        else load_b(b_src(gs + 1), bn);
        if (s == 2) {
          // the loads of the chunk after the next one, BEHIND this step's B prefetch (vector loads return in order)
          __builtin_amdgcn_sched_barrier(0);
          if (ch + 2 < nchunks) stg.load((ch + 2) * 16);
          else if (has_next) {
            // this tile's rows are all staged (its last chunk went out just above): the row offsets of the NEXT tile take
            // their place, its chunk 0 is fetched during this tile's last-but-one chunk, its chunk 1 during the last
            if (!last) stg.make(p, bx_n * BM);
            stg.load(last ? 16 : 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const int i = s * MF + mf;
          if (i + AD < 5 * MF) read_a(abase, i + AD, a[(i + AD) % (AD + 1)]);
          __builtin_amdgcn_sched_barrier(0);
          bf16x8 (&ac)[3] = a[i % (AD + 1)];
#define X6_MMA(qa, qb) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[qa], bc[qb][nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) { X6_MMA(2, 0) X6_MMA(0, 2) X6_MMA(1, 1) X6_MMA(1, 0) X6_MMA(0, 1) X6_MMA(0, 0) }
#undef X6_MMA
          __builtin_amdgcn_sched_barrier(0);
        }
        ++gs;
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();       // the next chunk is complete in the other buffer; nobody reads this one any more
      kbuf ^= 1;
    }
    // the tile leaves through the buffer its last chunk has released (kbuf now names the buffer of the next tile's chunk 0)
    c3l_epilogue<MF, NF, WM, WN, MODE>(p, acc, smem + (kbuf ^ 1) * abytes, tab, p0, n0);
    __syncthreads();         // staging, tables and the accumulator exchange are free again
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bx = bx_n;
    by = by_n;
    bptr = bptr_n;
  }
}
