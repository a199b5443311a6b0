// 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores with split-fp32 operands, fp32 accumulation.
// Two math modes, selected by the number of bf16 pieces NP an fp32 operand is split into:
//   NP = 3 ("bf16x6", fp32-class - the default of the engine): x = h + m + l with h = bf16(x), m = bf16(x - h),
//          l = bf16(x - h - m).  The split is EXACT (24 mantissa bits = 3 x 8) and  a*b = sum of the six piece
//          products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm), each one exact in the fp32 accumulator; the dropped
//          terms (ml, lm, ll) are <= 2^-24 |a b| - the size of ONE fp32 rounding, so the result is as accurate as an
//          fp32 FMA chain.  Six v_mfma_f32_16x16x32_bf16 per product = 2.5 PF / 6 = 417 TFLOP/s-equivalent, against
//          157 TFLOP/s of the exact fp32 MFMA (conv.hip).
//   NP = 2 ("bf16x3", optional, reduced precision ~2^-16 per product): hi*hi + hi*lo + lo*hi.
//
// This is the workhorse of the path: BasicBlock convs of reference lib/models/pose_hrnet.py:28-57, 214 of the
// ~300 convolutions of CoAM-W48 and ~75 % of its FLOPs, forward and (with FLIP) data gradient.
//
// Structure - direct convolution with an LDS-resident input tile instead of 9 separate im2col gathers:
//   * pixels are addressed in a zero-padded flattened space  p = n*IB + (y+1)*SW + (x+1),  SW = W+1,
//     IB = (H+1)*SW : one zero column between consecutive rows (c3_common.h: c3_row_width), one zero row between images.  A filter tap is then
//     a constant shift  (r-1)*SW + (s-1)  of p, valid across row and image boundaries alike;
//   * a workgroup owns BM consecutive p's and all BN output channels; per channel chunk it stages the
//     BM + 2*SW + 2 input rows it needs ONCE (fp32 -> bf16 pieces at store time) and all 9 taps read shifted rows of
//     that tile.  NP = 2: 32-channel chunks (+ one 16-channel tail), rows [32 hi | 32 lo | pad] = 160 B.
//     NP = 3: 16-channel chunks, rows [16 h | 16 m | 16 l] = 96 B, two taps per K = 32 MFMA (lanes 0-31 feed tap t,
//     lanes 32-63 tap t+1; the 10th half-step of a chunk has zero weights).  Both strides are 32 mod 64 bytes, which
//     makes the four 16-lane groups of ds_read_b128 conflict free (checked exhaustively);
//   * weights are split and re-ordered ONCE per weight update by buctd_conv3x3_*_prep into the exact stage image the
//     kernel consumes ([step][Co][NP x 32 k-slots] bf16), so the double-buffered B stage is a plain 16-byte copy;
//   * the 4 waves tile the workgroup 4x1 (BN = 48/64) or 2x2 (BN = 96/128): every wave issues its fragment reads up
//     front and then MF*NF*(3 | 6) back-to-back MFMAs, small terms first; the step loop is fully unrolled per chunk;
//   * workgroups are renumbered so that each XCD (own L2) walks a contiguous run of position tiles;
//   * epilogue as in conv.hip: bias, Welford BN partials (+ per-group valid-row counts, since pad positions are
//     skipped), eval-BN scale/shift, residual, ReLU.
#include "c3_common.h"
#include <string.h>

// conv3x3_lean.hip: the train-mode option sets as template arguments (c3_lean.h)
enum { C3M_IN_BN = 1, C3M_STATS = 2, C3M_RES = 4, C3M_BS_REBUILD = 8, C3M_BS_Y = 16 };
int c3_lean_launch(const C3Group& h, int fam, int mode, unsigned grid, size_t lds, hipStream_t st);
// conv3x3_pers.hip: the same option sets as a persistent grid of resident workgroups (c3_pers.h)
int c3_pers_launch(const C3Group& h, int fam, int mode, unsigned grid, size_t lds, hipStream_t st);
#define C3P_SLOTS 512      // resident workgroups of a persistent launch: two per CU
static inline size_t c3p_lds_bytes(int na, int Ci, int BN) {       // c3_pers.h: two A buffers (each at least the epilogue's LDS) + tables
  size_t b = (size_t)na * 32 * Geo<3>::ROWB;
  if (b < (size_t)C3_EPI_LDS) b = C3_EPI_LDS;
  return 2 * b + (size_t)3 * Ci * 4 + (size_t)4 * BN * 4;
}

template <int NP, int MF, int NF, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(C3Args p) {
  using G = Geo<NP>;
  constexpr int ROWB = G::ROWB, PST = G::PST, CPR = G::CPR, BROW = G::BROW, BLDS = G::BLDS;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int RPP = 256 / CPR;                                 // rows staged per pass (32 | 64)
  constexpr int PA = (BM + 2 * MAX_SW + 2 + RPP - 1) / RPP;      // float4 loads per thread for one A chunk, max
  constexpr int BPR = BROW / 16;                                 // 16-byte pieces per B row (8 | 12)
  constexpr int PB = (BN * BPR + 255) / 256;                     // 16-byte pieces per thread for one B step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int na = p.na;
  const int arows = na * 32;
  unsigned char* At = smem;                              // [na*32][ROWB]
  unsigned char* Bt = smem + (size_t)arows * ROWB;       // [2][BN][BLDS]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;

  // XCD-aware tile order: hardware workgroup id round-robins over the 8 XCDs; give every XCD a contiguous run of
  // tiles so that the halo rows shared by neighbouring position tiles (and the N tiles of one position tile) hit
  // the same L2.
  int bx, by;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = lin & 7, idx = lin >> 3, per = total >> 3, rem = total & 7;
    const unsigned L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    bx = (int)(L / gy);
    by = (int)(L - (unsigned)bx * gy);
  }
  const int p0 = bx * BM, n0 = by * BN;
  const int halo = p.SW + 1;
  const int c4 = (t % CPR) * 4;                  // this thread's 4-channel slot inside a chunk
  const int prow = t / CPR;                      // its row inside a staging pass

  // global element offset (channel 0) of every staged row this thread fills; -1 = zero row, -2 = beyond the tile
  int goff[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int row = prow + RPP * q;
    const int pp = p0 - halo + row;
    int o = row < arows ? -1 : -2;
    if (row < arows && pp >= 0 && pp < p.P) {
      const int n = fast_div(pp, p.ib_mul, p.ib_sh);
      const int rem = pp - n * p.IB;
      const int yy = fast_div(rem, p.sw_mul, p.sw_sh);   // 0 = pad row above the image
      const int xx = rem - yy * p.SW;                     // 0 and SW-1 = pad columns
      if (n < p.N && yy >= 1 && xx >= 1 && xx <= p.W) o = ((n * p.H + yy - 1) * p.W + xx - 1) * p.Ci;
    }
    goff[q] = o;
  }

  f32x4 areg[PA], breg[PB];
  // unconditional loads from clamped (always valid) addresses; the zero-select for pad rows happens at store time so
  // that nothing consumes the load result early (a branch around each load would serialise them behind vmcnt(0))
  auto load_a = [&](int c0, int cw) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (RPP * q < arows) {
        const bool ok = goff[q] >= 0 && c4 < cw * 16;
        areg[q] = *reinterpret_cast<const f32x4*>(p.x + (ok ? goff[q] + c0 + c4 : 0));
      }
  };
  auto store_a = [&](int cw) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (RPP * q < arows && goff[q] != -2) {
        const int row = prow + RPP * q;
        const bool ok = goff[q] >= 0 && c4 < cw * 16;
        split_store<NP, PST>(At + (size_t)row * ROWB, c4, ok ? areg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
      }
  };
  // B: one step image is [Co][BROW]; this thread copies 16-byte pieces (row, piece) = divmod(t + 256 q, BPR)
  int bsrc[PB], bdst[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int id = t + 256 * q;
    const int nl = id / BPR, pc = id - nl * BPR;
    const bool ok = nl < BN;
    bsrc[q] = ok ? (n0 + nl) * BROW + pc * 16 : 0;
    bdst[q] = ok ? nl * BLDS + pc * 16 : (nl & 15) * BLDS + BROW + (t & 1) * 16;   // idle threads: row pad bytes
  }
  const long step_bytes = (long)p.Co * BROW;
  auto load_b = [&](int gs) {
    const unsigned char* src = p.wp + gs * step_bytes;
#pragma unroll
    for (int q = 0; q < PB; ++q) breg[q] = *reinterpret_cast<const f32x4*>(src + bsrc[q]);
  };
  auto store_b = [&](int buf) {
    unsigned char* dst = Bt + (size_t)buf * BN * BLDS;
#pragma unroll
    for (int q = 0; q < PB; ++q) *reinterpret_cast<f32x4*>(dst + bdst[q]) = breg[q];
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // chunk plan: NP = 2: Ci/32 chunks of 32 channels (9 steps each) + a 16-channel tail (5 steps);
  //             NP = 3: Ci/16 chunks of 16 channels (5 steps each)
  const int nfull = NP == 2 ? p.Ci / CK : 0;
  const int ntail = NP == 2 ? ((p.Ci % CK) ? 1 : 0) : p.Ci / 16;
  const int nchunks = nfull + ntail, last_step = nfull * 9 + ntail * 5 - 1;
  const unsigned char* abase = At + (size_t)(wave_m * MF * 16 + i16) * ROWB + (g & 1) * 16;
  const unsigned char* bbase = Bt + (size_t)(wave_n * NF * 16 + i16) * BLDS + g * 16;
  const bool lowk = g < 2;     // lanes feeding reduction slots 0..15 of the K = 32 MFMA
  int gs = 0, buf = 0;

  // one chunk = NS fully unrolled steps; CW = 16-channel groups in the chunk (2: one tap per step, 1: two taps)
  auto run_chunk = [&](auto cw_tag) {
    constexpr int CW = decltype(cw_tag)::value;
    constexpr int NS = CW == 2 ? 9 : 5;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      load_b(gs < last_step ? gs + 1 : last_step);
      const int tap0 = CW == 2 ? s : 2 * s;
      const int tap1 = CW == 2 ? s : (2 * s + 1 < 9 ? 2 * s + 1 : 2 * s);   // tap 9 of a paired chunk: its weights are zero
      const int o0 = ((tap0 / 3) * p.SW + tap0 % 3) * ROWB;
      const int o1 = ((tap1 / 3) * p.SW + tap1 % 3) * ROWB + (CW == 2 ? 32 : 0);
      const unsigned char* ap = abase + (lowk ? o0 : o1);
      const unsigned char* bp = bbase + (size_t)buf * BN * BLDS;
      bf16x8 a[NP][MF], b[NP][NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int q = 0; q < NP; ++q) b[q][nf] = *reinterpret_cast<const bf16x8*>(bp + nf * 16 * BLDS + q * 64);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int q = 0; q < NP; ++q) a[q][mf] = *reinterpret_cast<const bf16x8*>(ap + mf * 16 * ROWB + q * PST);
      // small terms first, the leading product last; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
#define C3_MMA(qa, qb)                                                                                          \
  _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) acc[mf][nf] =                                                \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa][mf], b[qb][nf], acc[mf][nf], 0, 0, 0);
        if constexpr (NP == 3) {
          C3_MMA(2, 0) C3_MMA(0, 2) C3_MMA(1, 1) C3_MMA(1, 0) C3_MMA(0, 1) C3_MMA(0, 0)
        } else {
          C3_MMA(1, 0) C3_MMA(0, 1) C3_MMA(0, 0)
        }
#undef C3_MMA
      }
      store_b(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      ++gs;
    }
  };

  const int cw0 = nfull > 0 ? 2 : 1;
  load_a(0, cw0);
  load_b(0);
  store_a(cw0);
  store_b(0);
  if (nchunks > 1) load_a(cw0 * 16, nfull > 1 ? 2 : 1);   // in flight during the first chunk
  __syncthreads();
  int c_next = cw0 * 16;                          // first channel of the chunk whose loads are in flight
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cw = ch < nfull ? 2 : 1;
    if (ch > 0) {
      store_a(cw);                                   // everybody left the previous tile at the last step's barrier
      c_next += cw * 16;
      if (ch + 1 < nchunks) load_a(c_next, ch + 1 < nfull ? 2 : 1);
      __syncthreads();
    }
    if (cw == 2) run_chunk(IC<2>{});
    else run_chunk(IC<1>{});
  }

  c3_epilogue<MF, NF, WM, WN>(p, acc, smem, bx, by, p0, n0);
}

// ---- the fp32-class kernel (NP = 3, "bf16x6") ----------------------------------------------------------------------
// Same tiling and epilogue as conv3x3_split_kernel, different data movement:
//   * A: 16-channel chunks, rows [16 h | 16 m | 16 l] = 96 B, DOUBLE buffered: the next chunk is split and stored while
//     the current one is being multiplied (its global loads were issued a chunk earlier), one barrier per chunk;
//   * B: no LDS stage and no per-step barrier: the prepared image is laid out in MFMA fragment order
//     ([step][Co/16][piece][lane][16 B]), every wave fetches the 3*NF fragments of a step straight from L2/L1
//     (1 KB contiguous per fragment; all workgroups read the same few hundred KB) at the top of the step.
//     With six MFMAs per product the fragment stream needs ~31 B/clk/CU of the 64 B/clk L1 - half of what the
//     three-MFMA mode would need, which is why that mode keeps its LDS stage.
// Two taps per K = 32 MFMA (lanes 0-31 feed tap 2s, lanes 32-63 tap 2s+1; the 10th half-step has zero weights).
// One output tile (bx, by) of a launch described by p: the body of conv3x3_x6_kernel (one tile per workgroup) and of
// conv3x3_x6_group_kernel (a persistent workgroup walking a tile table over several convolutions).
// smem = [DBUF ? 2 : 1][arows][ROWB] A buffers | [3][Ci] floats: the input-BatchNorm table (mean, invstd * gamma, beta)
template <int MF, int NF, int WM, int WN, bool DBUF, bool BPF_>
__device__ __forceinline__ void c3x6_tile(const C3Args& p, unsigned char* smem, int bx, int by) {
  constexpr int ROWB = Geo<3>::ROWB, PST = Geo<3>::PST, CPR = Geo<3>::CPR;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int RPP = 256 / CPR;                                 // 64 rows staged per pass
  constexpr int PA = (BM + 2 * MAX_SW + 2 + RPP - 1) / RPP;
  const int arows = p.na * 32;
  const size_t abytes = DBUF ? (size_t)arows * ROWB : 0;         // one A buffer
  float* bntab = reinterpret_cast<float*>(smem + (size_t)(DBUF ? 2 : 1) * arows * ROWB);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // wave-uniform: keeps the B addressing scalar
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int p0 = bx * BM, n0 = by * BN;
  const bool in_bn = p.in_mean != nullptr || p.in_acc.acc != nullptr;
  const int halo = p.SW + 1;
  const int c4 = (t % CPR) * 4, prow = t / CPR;

  // global element offset (channel 0) of every staged row of this thread; -1 zero row, -2 beyond the tile
  int goff[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int row = prow + RPP * q;
    const int pp = p0 - halo + row;
    int o = row < arows ? -1 : -2;
    if (row < arows && pp >= 0 && pp < p.P) {
      const int n = fast_div(pp, p.ib_mul, p.ib_sh);
      const int rem = pp - n * p.IB;
      const int yy = fast_div(rem, p.sw_mul, p.sw_sh);
      const int xx = rem - yy * p.SW;
      if (n < p.N && yy >= 1 && xx >= 1 && xx <= p.W) o = ((n * p.H + yy - 1) * p.W + xx - 1) * p.Ci;
    }
    goff[q] = o;
  }
  f32x4 areg[PA];
  auto load_a = [&](int c0) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (RPP * q < arows) areg[q] = *reinterpret_cast<const f32x4*>(p.x + (goff[q] >= 0 ? goff[q] + c0 + c4 : 0));
  };
  auto store_a = [&](unsigned char* At, int c0) {
    f32x4 mu, sc, be;
    if (in_bn) {
      mu = *reinterpret_cast<const f32x4*>(bntab + c0 + c4);
      sc = *reinterpret_cast<const f32x4*>(bntab + p.Ci + c0 + c4);
      be = *reinterpret_cast<const f32x4*>(bntab + 2 * p.Ci + c0 + c4);
    }
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (RPP * q < arows && goff[q] != -2) {
        f32x4 v = goff[q] >= 0 ? areg[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (in_bn && goff[q] >= 0) {                 // zero padding stays zero: it pads the NORMALISED tensor
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = (v[j] - mu[j]) * sc[j] + be[j];
            if (p.in_relu) v[j] = fmaxf(v[j], 0.f);
          }
        }
        split_store<3, PST>(At + (size_t)(prow + RPP * q) * ROWB, c4, v);
      }
  };

  // B fragments of this lane: image [step][Co/16][3][64][16 B]
  const unsigned char* bptr = p.wp + ((size_t)(n0 / 16 + wave_n * NF) * 3) * 1024;   // scalar base
  const int blane = lane * 16;                                                        // the only per-lane part
  const size_t bstep = (size_t)(p.Co / 16) * 3 * 1024;
  constexpr bool BPF = BPF_;         // step-ahead B prefetch (36 more registers per step of distance)
  // small wave tiles (MF <= 2: at most 36 MFMAs = 576 cycles per step) cannot cover an L2 round trip (~1000 cycles, cycle
  // stamps on the 384-channel 12x9 maps) with one step of distance: their fragments travel two steps ahead
  constexpr bool BPF2 = BPF && MF <= 2 && NF <= 3;
  bf16x8 bc[3][NF], bn[BPF ? 3 : 1][BPF ? NF : 1];      // fragments of the current step / of the next one (in flight)
  bf16x8 bn2[BPF2 ? 3 : 1][BPF2 ? NF : 1];              // ... and of the one after
  auto load_b = [&](int gs, bf16x8 (&dst)[3][NF]) {
    const unsigned char* src = bptr + (size_t)gs * bstep;
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

  const int nchunks = p.Ci / 16, last_step = nchunks * 5 - 1;
  const size_t aoff = (size_t)(wave_m * MF * 16 + i16) * ROWB + (g & 1) * 16;
  const bool lowk = g < 2;
  int gs = 0;

  // one chunk = 5 unrolled steps.  The B fragments travel one step ahead: a step first takes over the set fetched
  // during the previous step (register moves - indexing two sets by step parity made the compiler keep copies of both
  // and spill), then issues the fetch of the next step's set, whose L2 latency the step's MFMAs cover
  constexpr int AD = MF >= 2 ? 2 : 1;      // fragment prefetch distance
  bf16x8 a[AD + 1][3];
  auto read_a = [&](const unsigned char* abase, int i, bf16x8 (&dst)[3]) {     // fragment i = (step, mf) of the chunk
    const int st = i / MF, mf = i % MF;
    const int tap0 = 2 * st, tap1 = 2 * st + 1 < 9 ? 2 * st + 1 : 2 * st;
    const int o0 = ((tap0 / 3) * p.SW + tap0 % 3) * ROWB, o1 = ((tap1 / 3) * p.SW + tap1 % 3) * ROWB;
    const unsigned char* ap = abase + (lowk ? o0 : o1) + mf * 16 * ROWB;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(ap + q * PST);
  };
  auto run_chunk = [&](int ch) {
    const unsigned char* abase = smem + (ch & 1) * abytes + aoff;
#pragma unroll
    for (int i = 0; i < AD; ++i) read_a(abase, i, a[i]);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (DBUF && s == 2 && ch + 1 < nchunks) {    // next chunk: registers -> pieces -> the other A buffer
        // done while the fewest registers are live (no B prefetch in flight, no A fragments): the split needs ~60
        store_a(smem + ((ch + 1) & 1) * abytes, (ch + 1) * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (BPF) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            bc[q][nf] = bn[q][nf];
            if constexpr (BPF2) bn[q][nf] = bn2[q][nf];
          }
        if constexpr (BPF2) load_b(gs + 2 < last_step ? gs + 2 : last_step, bn2);
        else load_b(gs < last_step ? gs + 1 : last_step, bn);
      } else {
        load_b(gs, bc);          // 144 MFMAs per step: the fetch latency is small against them, the registers are not
      }
      if (DBUF && s == 2 && ch + 2 < nchunks) {
        // the global loads of chunk ch+2 go out BEHIND this step's B prefetch: vector loads return in order, so the next
        // step's wait for its B fragments (older) leaves them in flight, and only the wait two steps on needs them -
        // issued in front of the prefetch they put their whole HBM latency into the very next step
        __builtin_amdgcn_sched_barrier(0);
        load_a((ch + 2) * 16);
        __builtin_amdgcn_sched_barrier(0);
      }
      // A fragments travel AD 16-row fragments ahead of their MFMAs through a ring of AD + 1 register sets, across the
      // step boundaries of the chunk (flat index i = s * MF + mf): with eight waves reading, a ds_read_b128 triple takes
      // longer than the 18 MFMAs of one fragment (cycle stamps: ~650 idle cycles per 72-MFMA step with AD = 1 and a
      // cold start at every step).  The scheduling fences keep the compiler from hoisting every read to the top.
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
      __builtin_amdgcn_sched_barrier(0);           // keep the steps apart: cross-step hoisting only costs registers
    }
  };

  load_a(0);
  if constexpr (BPF) load_b(0, bn);
  if constexpr (BPF2) load_b(last_step > 0 ? 1 : 0, bn2);
  if (in_bn) {
    // the producer's BatchNorm as (mean, invstd * gamma, beta) per input channel, under the first chunk's loads: from the
    // arrays a finalize launch left, or decoded from the producer's accumulator (bn_acc.h) - then tile (0, 0) also
    // leaves mean / invstd for the backward kernels and updates the running statistics
    for (int c = t; c < p.Ci; c += 256) {
      float m, is;
      if (p.in_acc.acc) {
        const BnFwdStat st = bnacc_fwd_stat(p.in_acc.acc, p.Ci, c, p.in_acc.rows, p.in_acc.eps);
        m = st.mean;
        is = st.invstd;
        if (bx == 0 && by == 0) {
          p.in_acc.mean_out[c] = m;
          p.in_acc.invstd_out[c] = is;
          if (p.in_acc.rmean) bnacc_running(st, p.in_acc.rows, p.in_acc.momentum, p.in_acc.rmean, p.in_acc.rvar, c);
        }
      } else {
        m = p.in_mean[c];
        is = p.in_invstd[c];
      }
      bntab[c] = m;
      bntab[p.Ci + c] = is * p.in_gamma[c];
      bntab[2 * p.Ci + c] = p.in_beta[c];
    }
    __syncthreads();
  }
  store_a(smem, 0);
  if (nchunks > 1) load_a(16);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    if (!DBUF && ch > 0) {   // single buffer (the largest position tiles): restage between two barriers
      store_a(smem, ch * 16);
      if (ch + 1 < nchunks) load_a((ch + 1) * 16);
      __syncthreads();
    }
    run_chunk(ch);
    __syncthreads();       // DBUF: chunk ch+1 is complete in its buffer; nobody reads this chunk's buffer any more
  }
  c3_epilogue<MF, NF, WM, WN>(p, acc, smem, bx, by, p0, n0);
}

template <int MF, int NF, int WM, int WN, bool DBUF, bool BPF_, int WPS>
__global__ __launch_bounds__(256, WPS) void conv3x3_x6_kernel(C3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int bx, by;
  {
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    const unsigned xcd = lin & 7, idx = lin >> 3, per = total >> 3, rem = total & 7;
    const unsigned L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    if (p.col_major) {
      // large filters (384 -> 384: 8 MB of prepared B, two L2s' worth): an XCD's run of workgroups walks the position
      // tiles of ONE column tile, so its L2 keeps that column tile's 1/gy of B for all of them; position-major order
      // streams the whole image through every L2 and each B fetch pays a MALL / HBM round trip
      by = (int)(L / gx);
      bx = (int)(L - (unsigned)by * gx);
    } else {
      bx = (int)(L / gy);
      by = (int)(L - (unsigned)bx * gy);
    }
  }
  c3x6_tile<MF, NF, WM, WN, DBUF, BPF_>(p, smem, bx, by);
}

// ---- several convolutions in ONE launch (the branches of a HighResolutionModule, pose_hrnet.py:177-185) -----------------
// A HighResolutionModule runs 2-4 independent branches whose k-th convolutions cost the same FLOPs on maps of different
// size.  As separate launches each of them is ONE round of workgroups on the 512 slots of the chip: all of them load their
// first chunk, multiply, and write their tile at the same time, so prologue and epilogue are exposed chip-wide and launches
// of other streams only get slots when the round retires - all at once.  Here the tiles of all branches form one grid:
// the dispatcher hands a new tile to a CU slot as soon as one retires, rounds de-phase, one workgroup's epilogue runs under
// its neighbour's main loop, and the tail of the grid is made of the cheapest tiles (convolutions ordered by tile cost).
// Every tile is computed by c3x6_tile with the argument block and the tile shape of ITS convolution - the arithmetic (and
// the accumulation order of every output and of the statistics) is that of the single launches: bit-identical results.
// (C3Group: c3_common.h)

// FAM 0: 48-channel multiples (NF = 3; HRNet-W48), FAM 1: 32-channel multiples (NF = 2 / 4; HRNet-W32)
template <int FAM>
__global__ __launch_bounds__(256, 2) void conv3x3_x6_group_kernel(C3Group g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the descriptor is read where it lies - in the kernel-argument segment (constant memory, scalar loads at a computed
  // offset): indexing the by-value copy with the convolution id made the compiler spill the whole block to scratch
  const C3Group& g = *(const C3Group*)__builtin_amdgcn_kernarg_segment_ptr();
  // workgroup -> (convolution, tile): hardware workgroup ids round-robin over the 8 XCDs; every XCD walks, convolution after
  // convolution, a contiguous eighth of that convolution's tile sequence (its L2 keeps the halo rows of neighbouring tiles)
  const unsigned lin = blockIdx.x, xcd = lin & 7;
  unsigned idx = lin >> 3;
  int c = -1;
  unsigned L = 0;
  for (int k = 0; k < g.nconv; ++k) {
    const unsigned total = (unsigned)g.tiles[k], per = total >> 3, rem = total & 7;
    const unsigned mine = per + (xcd < rem ? 1u : 0u);
    if (idx < mine) {
      L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
      c = k;
      break;
    }
    idx -= mine;
  }
  if (c < 0) return;
  const C3Args& p = g.conv[c];
  int bx, by;
  if (p.col_major) {
    by = (int)(L / (unsigned)g.gx[c]);
    bx = (int)(L - (unsigned)by * g.gx[c]);
  } else {
    bx = (int)(L / (unsigned)g.gy[c]);
    by = (int)(L - (unsigned)bx * g.gy[c]);
  }
  const int v = g.variant[c];
  if constexpr (FAM == 0) {
    switch (v) {
      case 0: c3x6_tile<7, 3, 4, 1, false, false>(p, smem, bx, by); break;
      case 1: c3x6_tile<4, 3, 2, 2, true, true>(p, smem, bx, by); break;
      case 2: c3x6_tile<2, 2, 4, 1, true, true>(p, smem, bx, by); break;
      case 3: c3x6_tile<4, 3, 4, 1, true, true>(p, smem, bx, by); break;
      default: c3x6_tile<4, 2, 4, 1, true, true>(p, smem, bx, by); break;
    }
  } else {
    switch (v) {
      case 0: c3x6_tile<4, 2, 4, 1, true, true>(p, smem, bx, by); break;
      case 1: c3x6_tile<1, 4, 4, 1, true, true>(p, smem, bx, by); break;
      case 2: c3x6_tile<2, 2, 4, 1, true, true>(p, smem, bx, by); break;
      case 3: c3x6_tile<1, 2, 4, 1, true, true>(p, smem, bx, by); break;
      case 4: c3x6_tile<2, 4, 4, 1, true, true>(p, smem, bx, by); break;
      case 5: c3x6_tile<2, 4, 2, 2, true, true>(p, smem, bx, by); break;
      case 6: c3x6_tile<1, 3, 4, 1, true, true>(p, smem, bx, by); break;
      default: c3x6_tile<2, 3, 4, 1, true, true>(p, smem, bx, by); break;
    }
  }
}

// ------------------------------------------------------------------------------------- weight preparation ----
// out[step][n][k]: step = (chunk, s); full chunk: (tap s, channel c0 + k); 16-channel tail: slots 0-15 = tap 2s,
// 16-31 = tap 2s + 1 (tap 9 = zero).  flip = 0: B(n, tap, c) = w[n][tap][c] (w = [Nc][9][Kc]);
// flip = 1 (data gradient): B(n, tap, c) = w[c][8 - tap][n] (w = [Kc][9][Nc]).
__global__ __launch_bounds__(256) void conv3x3_prep_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                           int Kc, int Nc, int flip, long pieces) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= pieces) return;
  const int k4 = (int)(idx & 7);
  const long rown = idx >> 3;
  const int n = (int)(rown % Nc), step = (int)(rown / Nc);
  const int nfull = Kc / CK;
  int c0, s, cw;
  if (step < nfull * 9) { c0 = (step / 9) * CK; s = step % 9; cw = 2; }
  else { c0 = nfull * CK; s = step - nfull * 9; cw = 1; }
  const int c4 = k4 * 4;
  const int tap = cw == 2 ? s : 2 * s + (c4 >> 4);
  const int cc = c0 + (cw == 2 ? c4 : (c4 & 15));
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tap < 9) {
    if (!flip) v = *reinterpret_cast<const f32x4*>(w + ((long)n * 9 + tap) * Kc + cc);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = w[((long)(cc + j) * 9 + (8 - tap)) * Nc + n];
    }
  }
  split_store<2, 64>(out + rown * 128, c4, v);
}

// the same for many filters in one launch (all prepared images of a model after an optimizer step): items live in
// device memory, thread -> item by binary search over the running piece count; item.reserved = 3 selects the
// fragment-order image of the six-MFMA mode (conv3x3_prep3 layout), anything else the two-piece stage image
__device__ __forceinline__ void prep3_piece(const float* __restrict__ w, unsigned char* __restrict__ out, int Kc, int Nc,
                                            int flip, long idx);
__global__ __launch_bounds__(256) void conv3x3_prep_batched_kernel(const buctd_c3_prep_item* __restrict__ items, int n,
                                                                   long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].piece_begin <= idx) lo = mid;
    else hi = mid - 1;
  }
  const buctd_c3_prep_item it = items[lo];
  const long local = idx - it.piece_begin;
  const int Kc = it.flip ? it.Co : it.Ci, Nc = it.flip ? it.Ci : it.Co;
  if (it.reserved == 3) {
    prep3_piece(it.w, reinterpret_cast<unsigned char*>(it.wprep), Kc, Nc, it.flip, local);
    return;
  }
  const int k4 = (int)(local & 7);
  const long rown = local >> 3;
  const int nn = (int)(rown % Nc), step = (int)(rown / Nc);
  const int nfull = Kc / CK;
  int c0, s, cw;
  if (step < nfull * 9) { c0 = (step / 9) * CK; s = step % 9; cw = 2; }
  else { c0 = nfull * CK; s = step - nfull * 9; cw = 1; }
  const int c4 = k4 * 4;
  const int tap = cw == 2 ? s : 2 * s + (c4 >> 4);
  const int cc = c0 + (cw == 2 ? c4 : (c4 & 15));
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tap < 9) {
    if (!it.flip) v = *reinterpret_cast<const f32x4*>(it.w + ((long)nn * 9 + tap) * Kc + cc);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = it.w[((long)(cc + j) * 9 + (8 - tap)) * Nc + nn];
    }
  }
  split_store<2, 64>(reinterpret_cast<unsigned char*>(it.wprep) + rown * 128, c4, v);
}

// NP = 3 image in MFMA fragment order: out[step][n/16][piece][lane][8 bf16], lane = 16*g + (n & 15) holding k-slots
// 8g..8g+7 of row n; step = (16-channel chunk c, s): k-slots 0-15 = (tap 2s, channels 16c..16c+15), 16-31 = (tap 2s+1,
// same channels; tap 9 = zero).  One thread = 4 k-slots of one row.
__device__ __forceinline__ void prep3_piece(const float* __restrict__ w, unsigned char* __restrict__ out, int Kc, int Nc,
                                            int flip, long idx) {
  const int k4 = (int)(idx & 7);               // k-slots 4*k4 .. 4*k4+3
  const long rown = idx >> 3;
  const int n = (int)(rown % Nc), step = (int)(rown / Nc);
  const int chunk = step / 5, s = step - chunk * 5;
  const int tap = 2 * s + (k4 >> 2);
  const int cc = chunk * 16 + (k4 & 3) * 4;
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tap < 9) {
    if (!flip) v = *reinterpret_cast<const f32x4*>(w + ((long)n * 9 + tap) * Kc + cc);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = w[((long)(cc + j) * 9 + (8 - tap)) * Nc + n];
    }
  }
  const int lane = (k4 >> 1) * 16 + (n & 15);
  unsigned char* dst = out + (((size_t)step * (Nc / 16) + (n >> 4)) * 3) * 1024 + lane * 16;
  // pieces 1 KB apart; the 4 k-slots occupy 8 bytes at (k4 & 1) * 8 inside the lane's 16
  split_store<3, 1024>(dst, (k4 & 1) * 4, v);
}
__global__ __launch_bounds__(256) void conv3x3_prep3_kernel(const float* __restrict__ w, unsigned char* __restrict__ out,
                                                            int Kc, int Nc, int flip, long items) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= items) return;
  prep3_piece(w, out, Kc, Nc, flip, idx);
}

// ---------------------------------------------------------------------------------------------- host ----
struct C3Plan { int MF, NF, WM, WN, BM, BN, na; size_t lds; };

// magic for unsigned division of n < 2^31 by d (2 <= d < 2^31): q = mulhi(n, mul) >> sh

static int c3_steps(int Kc, int np) { return np == 3 ? (Kc / 16) * 5 : (Kc / CK) * 9 + ((Kc % CK) ? 5 : 0); }

// pers: the plan of a persistent launch (c3_pers.h) - double-buffered tiles only
static bool c3_plan(int np, int N, int H, int W, int Ci, int Co, C3Plan* pl, bool pers = false) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0 || c3_row_width(W) > MAX_SW) return false;
  const int rowb = np == 3 ? Geo<3>::ROWB : Geo<2>::ROWB, blds = np == 3 ? Geo<3>::BLDS : Geo<2>::BLDS;
  const long P = (long)N * (H + 1) * c3_row_width(W) + c3_row_width(W);
  int nf, wn;
  if (Co % 96 == 0) { nf = 3; wn = 2; }
  else if (Co % 48 == 0) { nf = 3; wn = 1; }
  else if (Co % 128 == 0) { nf = 4; wn = 2; }
  else if (Co % 64 == 0) { nf = 4; wn = 1; }
  else if (Co % 32 == 0) { nf = 2; wn = 1; }
  else { nf = 1; wn = 1; }
  // wide column tiles (2x2 waves, BN = 96) only pay while they still fill the machine: the low-resolution branches
  // (192 ch @24x18, 384 ch @12x9) run faster as 4x1 waves with BN = 48 and twice as many workgroups (measured)
  if (np == 2 && Co % 96 == 0 && ((P + 127) / 128) * (Co / 96) < 320) { nf = 3; wn = 1; }
  // bf16x6 (B fragments per wave from L2): the 192-channel 24x18 maps keep the 128 x 96 workgroup tile (252 workgroups,
  // one per CU: 53.6 us against 57.2 for 504 half-width tiles); the smallest maps (384 ch @12x9, 36 tiles of 128
  // positions) go to 32-column tiles - 432 workgroups instead of 288 that load 32 CUs twice (71 us against 95)
  bool small = false;
  if (np == 3 && Co % 32 == 0 && ((P + 127) / 128) * ((Co + 95) / 96) < 200) { nf = 2; wn = 1; small = true; }
  int wm = 4 / wn, bn = wn * nf * 16;
  // largest position tile that still gives every CU work (256 CUs, 2 resident workgroups each)
  int mf = 1;
  const int cand[2] = {4, 2};
  for (int i = (nf == 4 ? 1 : 0); i < 2; ++i) {   // 64-row x 64-column wave tiles would spill
    const long blocks = ((P + wm * cand[i] * 16 - 1) / (wm * cand[i] * 16)) * (Co / bn);
    // bf16x6 fetches its B fragments per wave, one step ahead: a 16-row wave tile (MF = 1) leaves 18 MFMAs to cover an
    // L2 round trip, so the smallest maps (384 ch @12x9: 288 workgroups at MF = 2) prefer the larger tile (measured)
    // The 32-column tiles of the smallest maps take MF = 4 from 160 workgroups on: 16 B of B fragments per lane and 2 MF MFMAs
    // make MF = 2 wave tiles ask the CU's vector cache for 62 B / cycle at full matrix rate (it delivers 64), MF = 4 for half of
    // that.  HRNet-W48 branch 3 (384 ch @12x9, N = 32) as 204 tiles of 256 x 32 instead of 396 of 128 x 32: 78.3 -> 81.4 us
    // alone (204 workgroups leave a wave per SIMD without a partner), but the train step launches it with branch 2 only, and
    // there the other member fills the CUs: 121.9 -> 107.9 us (scratch/b3_ab.sh).  One plan per shape, grouped or not: a
    // group launch stays bit-identical to its members' own launches (the BatchNorm sums are formed per wave tile).
    const long need = np == 3 ? (small ? 160 : 224) : 320;
    if (blocks >= need) { mf = cand[i]; break; }   // (238 for the 192-channel 24x18 maps at N = 32)
  }
  bool single = false;
  if (np == 3 && nf == 3 && wn == 1 && Co == bn && !pers) {
    // 512-position tiles in ONE round of workgroups (2 resident per CU = 512 slots) instead of 1.75 rounds of
    // 256-position tiles: the 48-channel branch at N*H*W >= ~115k positions (single A buffer: 64.5 KB at W = 72)
    const long b8 = (P + 511) / 512;
    if (b8 > 256 && b8 <= 512) {
      mf = 8; single = true;
      // the launch is ONE round of workgroups on 512 slots and a CU's two workgroups share its matrix pipe, so the launch
      // lasts as long as a CU with two tiles: 448-position tiles (MF = 7) fill all 512 slots with 7/8 of the work each
      // where 512-position tiles leave 63 slots empty (N = 32 @ 96x72: 449 -> 512 workgroups)
      if ((P + 447) / 448 <= 512) mf = 7;
    }
  }
  pl->MF = mf; pl->NF = nf; pl->WM = wm; pl->WN = wn; pl->BM = wm * mf * 16; pl->BN = bn;
  pl->na = (pl->BM + 2 * c3_row_width(W) + 3 + 31) / 32;      // the staged rows + one spare row (c3_lean.h: C3Stager)
  size_t stage = (size_t)4 * (mf >= 2 ? 2 : 1) * 16 * (nf * 16 + 4) * 4 + 4 * 128 * 4;   // epilogue staging + row offsets
  if (np == 3 && stage < (size_t)C3_EPI_LDS) stage = C3_EPI_LDS;      // ... the bs reduction scratch and the accumulator exchange
  if (np == 3) {            // two A buffers (one for the 512-position tiles), no B stage; behind them the input-BatchNorm table
    const int nb = single ? 1 : 2;
    while ((size_t)nb * pl->na * 32 * rowb < stage) ++pl->na;
    pl->lds = (size_t)nb * pl->na * 32 * rowb + (size_t)3 * Ci * sizeof(float);
  } else {
    while ((size_t)pl->na * 32 * rowb < stage) ++pl->na;
    pl->lds = (size_t)pl->na * 32 * rowb + (size_t)2 * pl->BN * blds;
  }
  return pl->lds <= 160 * 1024;
}

static bool c3_np_ok(int np) { return np == 2 || np == 3; }

template <int NP, int MF, int NF, int WM, int WN>
static int c3_launch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
  static unsigned char attr_done[3][BUCTD_MAX_DEVICES] = {{0}};
  void (*fn)(C3Args);
  int variant = 0;
  if constexpr (NP == 3) {
    if (MF >= 7) fn = conv3x3_x6_kernel<MF, NF, WM, WN, false, false, 2>;
    else { fn = conv3x3_x6_kernel<MF, NF, WM, WN, true, true, 2>; variant = 2; }
  }
  else fn = conv3x3_split_kernel<NP, (MF > 4 ? 4 : MF), NF, WM, WN>;
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[variant], "conv3x3 (split bf16)"))
    return rc;
  dim3 grid(ceil_div(a.P, pl.BM), a.Co / pl.BN);
  hipLaunchKernelGGL(fn, grid, dim3(256), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3 (split bf16)");
  return BUCTD_OK;
}

template <int NP>
static int c3_dispatch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
#define C3_CASE(mf, nf, wm, wn) \
  if (pl.MF == mf && pl.NF == nf && pl.WN == wn) return c3_launch<NP, mf, nf, wm, wn>(a, pl, st);
#define C3_MF(nf, wm, wn) C3_CASE(4, nf, wm, wn) C3_CASE(2, nf, wm, wn) C3_CASE(1, nf, wm, wn)
  if constexpr (NP == 3) { C3_CASE(8, 3, 4, 1) C3_CASE(7, 3, 4, 1) C3_CASE(8, 3, 2, 2) }
  C3_MF(1, 4, 1) C3_MF(2, 4, 1) C3_MF(3, 4, 1) C3_MF(3, 2, 2)
  C3_CASE(2, 4, 4, 1) C3_CASE(1, 4, 4, 1) C3_CASE(2, 4, 2, 2) C3_CASE(1, 4, 2, 2)
#undef C3_MF
#undef C3_CASE
  buctd_set_error("conv3x3 (split bf16): no kernel for MF=%d NF=%d WN=%d", pl.MF, pl.NF, pl.WN);
  return BUCTD_EINVAL;
}

static int c3_supported(int np, int N, int H, int W, int Ci, int Co) {
  C3Plan pl;
  return c3_np_ok(np) && c3_plan(np, N, H, W, Ci, Co, &pl) ? 1 : 0;
}

static int c3_stats_groups(int np, int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group) {
  C3Plan pl;
  BUCTD_CHECK_ARG(ngroups && rows_per_group && c3_np_ok(np) && c3_plan(np, N, H, W, Ci, Co, &pl),
                  "buctd_conv3x3_*_stats_groups: unsupported shape");
  const long P = (long)N * (H + 1) * c3_row_width(W) + c3_row_width(W);
  *ngroups = ceil_div(P, pl.BM) * pl.WM;
  *rows_per_group = pl.MF * 16;
  return BUCTD_OK;
}

static size_t c3_prep_bytes(int np, int Ci, int Co, int flip) {
  if (!c3_np_ok(np) || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return 0;
  const int Kc = flip ? Co : Ci, Nc = flip ? Ci : Co;
  return (size_t)c3_steps(Kc, np) * Nc * (np == 3 ? Geo<3>::BROW : Geo<2>::BROW);
}

static int c3_prep(int np, int Ci, int Co, const float* w, int flip, void* wprep, void* stream) {
  BUCTD_CHECK_ARG(w && wprep && c3_np_ok(np), "buctd_conv3x3_*_prep: null pointer");
  BUCTD_CHECK_ARG(Ci > 0 && Co > 0 && Ci % 16 == 0 && Co % 16 == 0, "buctd_conv3x3_*_prep: Ci=%d Co=%d must be multiples of 16",
                  Ci, Co);
  const int Kc = flip ? Co : Ci, Nc = flip ? Ci : Co;
  const long pieces = (long)c3_steps(Kc, np) * Nc * 8;
  if (np == 3)
    hipLaunchKernelGGL(conv3x3_prep3_kernel, dim3(ceil_div(pieces, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned char*)wprep, Kc, Nc, flip ? 1 : 0, pieces);
  else
    hipLaunchKernelGGL(conv3x3_prep_kernel, dim3(ceil_div(pieces, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned char*)wprep, Kc, Nc, flip ? 1 : 0, pieces);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_*_prep");
  return BUCTD_OK;
}

// x: [N][H][W][Ci] -> y: [N][H][W][Co], wprep from the matching prep call.  Forward: prep(Ci, Co, w, 0).
// Data gradient of a forward conv (CiF -> CoF): x = dy ([N][H][W][CoF]), y = dx ([N][H][W][CiF]), i.e. this call's
// Ci = CoF, Co = CiF, and wprep = prep(CiF, CoF, w, 1).
struct C3InBn { const float* mean; const float* invstd; const float* gamma; const float* beta; int relu; };
// BatchNorm-backward reduction as a by-product of a data-gradient launch (C3Args::bs_*)
struct C3BwdStat { const float* z; const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta; float* part; long long* acc; };
// accumulator forms (bn_acc.h): the launch's own forward statistics, and the input BatchNorm's statistics decoded in the prologue
struct C3Acc { long long* stats_acc; const buctd_bn_acc_in* in; };

// builds the argument block and the tile plan of one convolution (no launch)
static int c3_fill(int np, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                   const float* scale, const float* shift, const float* residual, int relu, float* y,
                   float* stats_partials, int* stats_counts, const C3InBn* in_bn, const C3BwdStat* bst, const C3Acc* accs,
                   C3Args& a, C3Plan& pl) {
  BUCTD_CHECK_ARG(x && wprep && y, "buctd_conv3x3 (split bf16): null tensor pointer");
  BUCTD_CHECK_ARG(c3_np_ok(np) && c3_plan(np, N, H, W, Ci, Co, &pl),
                  "buctd_conv3x3 (split bf16): unsupported shape N%d H%d W%d Ci%d Co%d", N, H, W, Ci, Co);
  BUCTD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "buctd_conv3x3 (split bf16): scale and shift go together");
  BUCTD_CHECK_ARG((stats_partials == nullptr) == (stats_counts == nullptr),
                  "buctd_conv3x3 (split bf16): stats partials and counts go together");
  a.x = x; a.wp = (const unsigned char*)wprep; a.out = y; a.bias = bias; a.scale = scale; a.shift = shift;
  a.res = residual; a.stats = stats_partials; a.counts = stats_counts;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.SW = c3_row_width(W); a.IB = (H + 1) * a.SW;
  const long P = (long)N * a.IB + a.SW;
  BUCTD_CHECK_ARG(P < 2147483647L && (long)N * H * W * (Ci > Co ? Ci : Co) < 2147483647L,
                  "buctd_conv3x3 (split bf16): tensor too large");
  a.P = (int)P;
  a.relu = relu; a.na = pl.na;
  a.omap = 0; a.oH = H; a.oW = W; a.ost = 1; a.oy0 = a.ox0 = 0;
  a.in_mean = a.in_invstd = a.in_gamma = a.in_beta = nullptr;
  a.in_relu = 0;
  if (in_bn && in_bn->mean) {
    BUCTD_CHECK_ARG(np == 3 && in_bn->invstd && in_bn->gamma && in_bn->beta,
                    "buctd_conv3x3: fused input BatchNorm needs the bf16x6 kernel and all four statistics arrays");
    a.in_mean = in_bn->mean; a.in_invstd = in_bn->invstd; a.in_gamma = in_bn->gamma; a.in_beta = in_bn->beta;
    a.in_relu = in_bn->relu;
  }
  a.stats_acc = nullptr; a.bs_acc = nullptr;
  memset(&a.in_acc, 0, sizeof(a.in_acc));
  if (accs) {
    BUCTD_CHECK_ARG(np == 3, "buctd_conv3x3: statistics accumulators need the bf16x6 kernel");
    a.stats_acc = accs->stats_acc;
    if (accs->in && accs->in->acc) {
      const buctd_bn_acc_in& i = *accs->in;
      BUCTD_CHECK_ARG(in_bn && !in_bn->mean && in_bn->gamma && in_bn->beta && i.rows > 0 && i.mean_out && i.invstd_out &&
                          (i.running_mean == nullptr) == (i.running_var == nullptr),
                      "buctd_conv3x3: input BatchNorm from an accumulator needs gamma, beta, rows, mean_out and invstd_out");
      a.in_gamma = in_bn->gamma; a.in_beta = in_bn->beta; a.in_relu = in_bn->relu;
      a.in_acc.acc = (const long long*)i.acc; a.in_acc.rows = (double)i.rows; a.in_acc.eps = i.eps; a.in_acc.momentum = i.momentum;
      a.in_acc.mean_out = i.mean_out; a.in_acc.invstd_out = i.invstd_out; a.in_acc.rmean = i.running_mean; a.in_acc.rvar = i.running_var;
    }
  }
  a.bs_z = a.bs_y = a.bs_mean = a.bs_invstd = a.bs_gamma = a.bs_beta = nullptr;
  a.bs_part = nullptr;
  if (bst && (bst->part || bst->acc)) {
    BUCTD_CHECK_ARG(np == 3 && bst->z && bst->mean && bst->invstd && (bst->y || (bst->gamma && bst->beta)),
                    "buctd_conv3x3: the BatchNorm-backward by-product needs the bf16x6 kernel, z, mean, invstd and y or gamma + beta");
    a.bs_z = bst->z; a.bs_y = bst->y; a.bs_mean = bst->mean; a.bs_invstd = bst->invstd; a.bs_gamma = bst->gamma;
    a.bs_beta = bst->beta; a.bs_part = bst->part; a.bs_acc = bst->acc;
  }
  a.col_major = (np == 3 && Co / pl.BN >= 2 && (size_t)c3_steps(Ci, 3) * Co * Geo<3>::BROW > ((size_t)3 << 20)) ? 1 : 0;
  magic_u32((unsigned)a.IB, &a.ib_mul, &a.ib_sh);
  magic_u32((unsigned)a.SW, &a.sw_mul, &a.sw_sh);
  return BUCTD_OK;
}

// the train-mode option set of a launch (c3_lean.h), or -1: the general kernel
static int c3_lean_mode(const C3Args& a) {
  // (tensors are addressed through buffer descriptors with 32-bit byte offsets, pad rows by an offset beyond 2 GB)
  if ((long)a.N * a.H * a.W * (a.Ci > a.Co ? a.Ci : a.Co) * 4 >= 2147483648L) return -1;
  if (a.bias || a.scale || a.shift || a.relu || a.stats || a.counts || a.bs_part || a.omap || a.in_mean || a.in_invstd) return -1;
  if (a.in_acc.acc && (!a.in_gamma || !a.in_beta)) return -1;
  int m = 0;
  if (a.in_acc.acc) m |= C3M_IN_BN;
  if (a.stats_acc) m |= C3M_STATS;
  if (a.res) m |= C3M_RES;
  if (a.bs_acc) m |= a.bs_y ? C3M_BS_Y : C3M_BS_REBUILD;
  switch (m) {
    case C3M_STATS: case C3M_STATS | C3M_IN_BN: case C3M_BS_REBUILD: case C3M_RES | C3M_BS_Y: case C3M_RES: return m;
    default: return -1;
  }
}

// general: the launch runs conv3x3_x6_group_kernel (no train-mode option set) - its family 1 also holds the small 48-column
// tiles (MF = 1 / 2, NF = 3) that HRNet-W48 gets at a few persons per call, so that an eval-mode group of W48 branches
// (buctd_conv3x3_bf16x6_group_eval) shares a kernel at small batches too; the train-mode kernels (c3_lean.h) do not have them
static int c3_group_variant(const C3Plan& pl, int* fam, bool general = false) {
  struct V { int mf, nf, wm, wn; };
  static const V f0[] = {{7, 3, 4, 1}, {4, 3, 2, 2}, {2, 2, 4, 1}, {4, 3, 4, 1}, {4, 2, 4, 1}};
  static const V f1[] = {{4, 2, 4, 1}, {1, 4, 4, 1}, {2, 2, 4, 1}, {1, 2, 4, 1}, {2, 4, 4, 1}, {2, 4, 2, 2}, {1, 3, 4, 1}, {2, 3, 4, 1}};
  for (int f = 0; f < 2; ++f) {
    if (*fam >= 0 && *fam != f) continue;
    const V* l = f ? f1 : f0;
    const int n = f ? (general ? 8 : 6) : 5;
    for (int i = 0; i < n; ++i)
      if (l[i].mf == pl.MF && l[i].nf == pl.NF && l[i].wm == pl.WM && l[i].wn == pl.WN) {
        *fam = f;
        return i;
      }
  }
  return -1;
}

// The persistent form of a train-mode launch (c3_pers.h), taken when the tiles of the launch exceed the resident slots - and only
// when switched on (buctd_conv3x3_bf16x6_persistent): measured 10-18 % SLOWER than the dispatcher-scheduled one-tile
// workgroups on the HRNet-W48 shapes, it stays an opt-in for experiments.
static int c3_persistent_on = 0;
extern "C" int buctd_conv3x3_bf16x6_persistent(int on) {
  const int was = c3_persistent_on;
  if (on >= 0) c3_persistent_on = on ? 1 : 0;
  return was;
}
static int c3_pers_try(int n, const C3Args* a, int lean, hipStream_t stream, bool* done, int* dry_wgs) {
  if (!c3_persistent_on) return BUCTD_OK;
  C3Plan pp[C3G_MAX];
  int order[C3G_MAX], var[C3G_MAX];
  double cost[C3G_MAX];
  long tiles = 0;
  for (int k = 0; k < n; ++k) {
    if (a[k].Ci < 32 || !c3_plan(3, a[k].N, a[k].H, a[k].W, a[k].Ci, a[k].Co, &pp[k], true)) return BUCTD_OK;
    tiles += (long)ceil_div(a[k].P, pp[k].BM) * (a[k].Co / pp[k].BN);
    cost[k] = (double)pp[k].BM * pp[k].BN * a[k].Ci;
    order[k] = k;
  }
  if (tiles <= C3P_SLOTS) return BUCTD_OK;
  int fam = -1;
  bool ok = false;
  for (int f = 0; f < 2 && !ok; ++f) {
    ok = true;
    for (int k = 0; k < n && ok; ++k) {
      fam = f;
      var[k] = c3_group_variant(pp[k], &fam);
      ok = var[k] >= 0 && !(f == 0 && (var[k] == 0 || var[k] > 3));   // (no persistent form: the single-buffered 448-position tile, the tiles added after round 6's experiment)
    }
  }
  if (!ok) return BUCTD_OK;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  C3Group h;
  h.nconv = n;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    const int k = order[i];
    h.conv[i] = a[k];
    h.conv[i].na = pp[k].na;
    h.gx[i] = ceil_div(a[k].P, pp[k].BM);
    h.gy[i] = a[k].Co / pp[k].BN;
    h.tiles[i] = h.gx[i] * h.gy[i];
    h.variant[i] = var[k];
    const size_t l = c3p_lds_bytes(pp[k].na, a[k].Ci, pp[k].BN);
    if (l > lds) lds = l;
  }
  for (int i = n; i < C3G_MAX; ++i) h.tiles[i] = h.gx[i] = h.gy[i] = h.variant[i] = 0;
  if (lds > 160 * 1024) return BUCTD_OK;
  *done = true;
  if (dry_wgs) { *dry_wgs = C3P_SLOTS; return BUCTD_OK; }
  return c3_pers_launch(h, fam, lean, C3P_SLOTS, lds, stream);
}

// n convolutions (argument blocks a[], tile plans pl[]) as ONE launch: the train-mode kernel of their common option set, else
// the general group kernel (n > 1 only).  *done = false: the tile shapes have no common kernel family - nothing was launched.
static int c3_group_launch(int n, const C3Args* a, const C3Plan* pl, hipStream_t stream, bool* done, int* dry_wgs = nullptr) {
  *done = false;
  int order[C3G_MAX];
  double cost[C3G_MAX];
  int lean = c3_lean_mode(a[0]);
  for (int k = 0; k < n; ++k) {
    cost[k] = (double)pl[k].BM * pl[k].BN * a[k].Ci;
    order[k] = k;
    if (c3_lean_mode(a[k]) != lean) lean = -1;
  }
  if (n == 1 && lean < 0) return BUCTD_OK;
  if (lean >= 0) {
    // more than one round of workgroups: the persistent form (a slot walks its tiles as one software pipeline)
    const int rc = c3_pers_try(n, a, lean, stream, done, dry_wgs);
    if (rc || *done) return rc;
  }
  // the kernel family that holds the tile shapes of ALL members (128 x 32 tiles exist in both families)
  int fam = -1;
  bool ok = false;
  int var[C3G_MAX];
  for (int f = 0; f < 2 && !ok; ++f) {
    ok = true;
    for (int k = 0; k < n && ok; ++k) {
      fam = f;
      var[k] = c3_group_variant(pl[k], &fam, lean < 0);
      ok = var[k] >= 0;
    }
  }
  if (!ok) return BUCTD_OK;
  // costliest tiles first: the grid ends with the cheap ones
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  C3Group h;
  h.nconv = n;
  size_t lds = 0;
  unsigned per_xcd = 0;
  for (int i = 0; i < n; ++i) {
    const int k = order[i];
    h.conv[i] = a[k];
    h.gx[i] = ceil_div(h.conv[i].P, pl[k].BM);
    h.gy[i] = h.conv[i].Co / pl[k].BN;
    h.tiles[i] = h.gx[i] * h.gy[i];
    h.variant[i] = var[k];
    const size_t l = pl[k].lds + (lean >= 0 ? (size_t)4 * pl[k].BN * sizeof(float) : 0);   // + the epilogue table (c3_lean.h)
    if (l > lds) lds = l;
    per_xcd += ((unsigned)h.tiles[i] + 7) >> 3;
  }
  for (int i = n; i < C3G_MAX; ++i) h.tiles[i] = h.gx[i] = h.gy[i] = h.variant[i] = 0;
  if (lds > 160 * 1024) return BUCTD_OK;
  *done = true;
  if (dry_wgs) { *dry_wgs = (int)(per_xcd * 8); return BUCTD_OK; }
  if (lean >= 0) return c3_lean_launch(h, fam, lean, per_xcd * 8, lds, stream);
  static unsigned char attr_done[2][BUCTD_MAX_DEVICES] = {{0}};
  void (*fn)(C3Group) = fam ? conv3x3_x6_group_kernel<1> : conv3x3_x6_group_kernel<0>;
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[fam], "buctd_conv3x3_bf16x6_group"))
    return rc;
  hipLaunchKernelGGL(fn, dim3(per_xcd * 8), dim3(256), lds, stream, h);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x6_group");
  return BUCTD_OK;
}

static int c3_run(int np, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                  const float* scale, const float* shift, const float* residual, int relu, float* y,
                  float* stats_partials, int* stats_counts, void* stream, const C3InBn* in_bn = nullptr,
                  const C3BwdStat* bst = nullptr, const C3Acc* accs = nullptr) {
  C3Args a;
  C3Plan pl;
  const int rc = c3_fill(np, N, H, W, Ci, Co, x, wprep, bias, scale, shift, residual, relu, y, stats_partials, stats_counts,
                         in_bn, bst, accs, a, pl);
  if (rc) return rc;
  if (np == 3 && c3_lean_mode(a) >= 0) {       // a train-mode option set: the specialised kernel, if the tile shape has one
    bool done = false;
    const int rc2 = c3_group_launch(1, &a, &pl, (hipStream_t)stream, &done);
    if (rc2 || done) return rc2;
  }
  return np == 3 ? c3_dispatch<3>(a, pl, (hipStream_t)stream) : c3_dispatch<2>(a, pl, (hipStream_t)stream);
}

// ---- "bf16x3" (NP = 2) entry points --------------------------------------------------------------------------
extern "C" int buctd_conv3x3_bf16x3_supported(int N, int H, int W, int Ci, int Co) { return c3_supported(2, N, H, W, Ci, Co); }
extern "C" int buctd_conv3x3_bf16x3_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group) {
  return c3_stats_groups(2, N, H, W, Ci, Co, ngroups, rows_per_group);
}
extern "C" size_t buctd_conv3x3_bf16x3_prep_bytes(int Ci, int Co, int flip) { return c3_prep_bytes(2, Ci, Co, flip); }
extern "C" int buctd_conv3x3_bf16x3_prep(int Ci, int Co, const float* w, int flip, void* wprep, void* stream) {
  return c3_prep(2, Ci, Co, w, flip, wprep, stream);
}
extern "C" int buctd_conv3x3_bf16x3_prep_batched(const buctd_c3_prep_item* items_device, int n, long total_pieces,
                                                 void* stream) {
  BUCTD_CHECK_ARG(items_device && n > 0 && total_pieces > 0, "buctd_conv3x3_bf16x3_prep_batched: bad argument");
  hipLaunchKernelGGL(conv3x3_prep_batched_kernel, dim3(ceil_div(total_pieces, 256)), dim3(256), 0, (hipStream_t)stream,
                     items_device, n, total_pieces);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3_prep_batched");
  return BUCTD_OK;
}
extern "C" int buctd_conv3x3_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                    const float* bias, const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, float* stats_partials, int* stats_counts, void* stream) {
  return c3_run(2, N, H, W, Ci, Co, x, wprep, bias, scale, shift, residual, relu, y, stats_partials, stats_counts, stream);
}

// ---- "bf16x6" (NP = 3, fp32-class) entry points -----------------------------------------------------------------
extern "C" int buctd_conv3x3_bf16x6_supported(int N, int H, int W, int Ci, int Co) { return c3_supported(3, N, H, W, Ci, Co); }
extern "C" int buctd_conv3x3_bf16x6_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group) {
  return c3_stats_groups(3, N, H, W, Ci, Co, ngroups, rows_per_group);
}
extern "C" size_t buctd_conv3x3_bf16x6_prep_bytes(int Ci, int Co, int flip) { return c3_prep_bytes(3, Ci, Co, flip); }
extern "C" int buctd_conv3x3_bf16x6_prep(int Ci, int Co, const float* w, int flip, void* wprep, void* stream) {
  return c3_prep(3, Ci, Co, w, flip, wprep, stream);
}
extern "C" int buctd_conv3x3_bf16x6(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                    const float* bias, const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, float* stats_partials, int* stats_counts, void* stream) {
  return c3_run(3, N, H, W, Ci, Co, x, wprep, bias, scale, shift, residual, relu, y, stats_partials, stats_counts, stream);
}
extern "C" int buctd_conv3x3_bf16x6_bnin(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                         const float* bias, const float* scale, const float* shift,
                                         const float* residual, int relu, float* y, float* stats_partials,
                                         int* stats_counts, const float* in_mean, const float* in_invstd,
                                         const float* in_gamma, const float* in_beta, int in_relu, void* stream) {
  C3InBn b{in_mean, in_invstd, in_gamma, in_beta, in_relu};
  return c3_run(3, N, H, W, Ci, Co, x, wprep, bias, scale, shift, residual, relu, y, stats_partials, stats_counts, stream, &b);
}

/* Data gradient of a 3x3 convolution (buctd_conv3x3_bf16x6 on the flipped image, residual = the skip gradient) that also
 * forms the reduction pass of the BatchNorm backward consuming its output g = dx: per row group (the groups of
 * buctd_conv3x3_bf16x6_stats_groups(N, H, W, Ci, Co)) s1 = sum m g, s2 = sum m g zhat with m the ReLU mask of that
 * BatchNorm's forward output (bn_y > 0 where given, else rebuilt from bn_z with gamma / beta) - bn_part [groups][2][Co],
 * the input of buctd_bn_bwd_from_partials.  Replaces the separate pass over g, z and y (bn.hip: bn_bwd_reduce2_kernel). */
extern "C" int buctd_conv3x3_bf16x6_bnstat(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                           const float* residual, float* y, const float* bn_z, const float* bn_y,
                                           const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                           const float* bn_beta, float* bn_part, void* stream) {
  C3BwdStat b{bn_z, bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_part, nullptr};
  BUCTD_CHECK_ARG(bn_part, "buctd_conv3x3_bf16x6_bnstat: null partials pointer");
  return c3_run(3, N, H, W, Ci, Co, x, wprep, nullptr, nullptr, nullptr, residual, 0, y, nullptr, nullptr, stream, nullptr,
                &b);
}

/* The accumulator forms (include/buctd_hip.h: "BatchNorm statistics without finalize launches").
 * buctd_conv3x3_bf16x6_acc: forward convolution whose output statistics go to stats_acc (may be null) and whose input may
 * be the raw output of a producing convolution, normalised on the fly with that producer's statistics - decoded from ITS
 * accumulator (in_bn: acc + rows + eps, gamma, beta) in the prologue; tile (0, 0) writes mean / invstd out and updates the
 * running statistics.  buctd_conv3x3_bf16x6_bnstat_acc: the data gradient with the BatchNorm-backward sums as accumulator. */
extern "C" int buctd_conv3x3_bf16x6_acc(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                        const float* residual, int relu, float* y, void* stats_acc,
                                        const buctd_bn_acc_in* in_bn, const float* in_gamma, const float* in_beta, int in_relu,
                                        void* stream) {
  C3InBn b{nullptr, nullptr, in_gamma, in_beta, in_relu};
  C3Acc a{(long long*)stats_acc, in_bn};
  BUCTD_CHECK_ARG(!in_bn || in_bn->acc, "buctd_conv3x3_bf16x6_acc: in_bn without an accumulator");
  return c3_run(3, N, H, W, Ci, Co, x, wprep, nullptr, nullptr, nullptr, residual, relu, y, nullptr, nullptr, stream,
                in_bn ? &b : nullptr, nullptr, &a);
}
extern "C" int buctd_conv3x3_bf16x6_bnstat_acc(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                               const float* residual, float* y, const float* bn_z, const float* bn_y,
                                               const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                               const float* bn_beta, void* bn_acc, void* stream) {
  C3BwdStat b{bn_z, bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta, nullptr, (long long*)bn_acc};
  BUCTD_CHECK_ARG(bn_acc, "buctd_conv3x3_bf16x6_bnstat_acc: null accumulator");
  return c3_run(3, N, H, W, Ci, Co, x, wprep, nullptr, nullptr, nullptr, residual, 0, y, nullptr, nullptr, stream, nullptr,
                &b);
}

/* Several 3x3 convolutions in ONE launch (include/buctd_hip.h: buctd_c3_conv): the tiles of all of them in one grid, each
 * computed exactly as its own buctd_conv3x3_bf16x6_acc / _bnstat_acc launch would.  Shapes whose tile plans have no place in
 * a group kernel are launched one after the other instead - the results are the same either way. */
static int c3_group_fill(int n, const buctd_c3_conv* convs, C3Args* a, C3Plan* pl) {
  for (int k = 0; k < n; ++k) {
    const buctd_c3_conv& c = convs[k];
    C3InBn ib{nullptr, nullptr, c.in_gamma, c.in_beta, c.in_relu};
    C3BwdStat bs{c.bn_z, c.bn_y, c.bn_mean, c.bn_invstd, c.bn_gamma, c.bn_beta, nullptr, (long long*)c.bn_acc};
    C3Acc ac{(long long*)c.stats_acc, c.in_bn};
    BUCTD_CHECK_ARG(!c.in_bn || c.in_bn->acc, "buctd_conv3x3_bf16x6_group: in_bn without an accumulator");
    const int rc = c3_fill(3, c.N, c.H, c.W, c.Ci, c.Co, c.x, c.wprep, nullptr, nullptr, nullptr, c.residual, c.relu, c.y,
                           nullptr, nullptr, c.in_bn ? &ib : nullptr, c.bn_acc ? &bs : nullptr, &ac, a[k], pl[k]);
    if (rc) return rc;
  }
  return BUCTD_OK;
}

extern "C" int buctd_conv3x3_bf16x6_group(int n, const buctd_c3_conv* convs, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && n <= C3G_MAX && convs, "buctd_conv3x3_bf16x6_group: 1..%d convolutions", C3G_MAX);
  C3Args a[C3G_MAX];
  C3Plan pl[C3G_MAX];
  if (const int rc0 = c3_group_fill(n, convs, a, pl)) return rc0;
  bool done = false;
  const int rc = c3_group_launch(n, a, pl, (hipStream_t)stream, &done);
  if (rc || done) return rc;
  for (int k = 0; k < n; ++k) {       // no common kernel: one launch per member (each may still take its train-mode kernel)
    bool d1 = false;
    int rc1 = c3_group_launch(1, a + k, pl + k, (hipStream_t)stream, &d1);
    if (!rc1 && !d1) rc1 = c3_dispatch<3>(a[k], pl[k], (hipStream_t)stream);
    if (rc1) return rc1;
  }
  return BUCTD_OK;
}

/* The k-th convolutions of the branches of a HighResolutionModule in EVAL mode - folded BatchNorm (scale / shift), skip
 * connection and ReLU in the epilogue - as one launch: each entry is one buctd_conv3x3_bf16x6 call, bit-identical to it. */
extern "C" int buctd_conv3x3_bf16x6_group_eval(int n, const buctd_c3_conv_eval* convs, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && n <= C3G_MAX && convs, "buctd_conv3x3_bf16x6_group_eval: 1..%d convolutions", C3G_MAX);
  C3Args a[C3G_MAX];
  C3Plan pl[C3G_MAX];
  for (int k = 0; k < n; ++k) {
    const buctd_c3_conv_eval& c = convs[k];
    const int rc = c3_fill(3, c.N, c.H, c.W, c.Ci, c.Co, c.x, c.wprep, nullptr, c.scale, c.shift, c.residual, c.relu, c.y,
                           nullptr, nullptr, nullptr, nullptr, nullptr, a[k], pl[k]);
    if (rc) return rc;
  }
  bool done = false;
  const int rc = n > 1 ? c3_group_launch(n, a, pl, (hipStream_t)stream, &done) : BUCTD_OK;
  if (rc || done) return rc;
  for (int k = 0; k < n; ++k)
    if (const int rc1 = c3_dispatch<3>(a[k], pl[k], (hipStream_t)stream)) return rc1;
  return BUCTD_OK;
}

/* The number of workgroups buctd_conv3x3_bf16x6_group(n, convs) launches as ONE kernel (0: the members have no common kernel
 * and go out one launch each; < 0: error).  No launch - for tools that look a launch up in a kernel trace by its grid. */
extern "C" int buctd_conv3x3_bf16x6_group_workgroups(int n, const buctd_c3_conv* convs) {
  BUCTD_CHECK_ARG(n > 0 && n <= C3G_MAX && convs, "buctd_conv3x3_bf16x6_group_workgroups: 1..%d convolutions", C3G_MAX);
  C3Args a[C3G_MAX];
  C3Plan pl[C3G_MAX];
  if (const int rc0 = c3_group_fill(n, convs, a, pl)) return rc0;
  bool done = false;
  int wgs = 0;
  const int rc = c3_group_launch(n, a, pl, nullptr, &done, &wgs);
  if (rc) return rc;
  return done ? wgs : 0;
}
