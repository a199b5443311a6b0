// 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3"):
// every fp32 operand is split as hi + lo (two bf16), and a*b ~= hi*hi + hi*lo + lo*hi is accumulated in fp32 by
// three v_mfma_f32_16x16x32_bf16 - ~5x the rate of v_mfma_f32_16x16x4_f32 at ~2^-16 relative product error
// (whole-network heat-map error 1.4e-4..3.5e-4 of full scale, measured; the fp32 engine in conv.hip stays the
// exact path and the default for parity tests).
//
// This is the workhorse of the path: BasicBlock convs of reference lib/models/pose_hrnet.py:28-57, 214 of the
// ~300 convolutions of CoAM-W48 and ~75 % of its FLOPs, forward and (with FLIP) data gradient.
//
// Structure - direct convolution with an LDS-resident input tile instead of 9 separate im2col gathers:
//   * pixels are addressed in a zero-padded flattened space  p = n*IB + (y+1)*SW + (x+1),  SW = W+2,
//     IB = (H+1)*SW : one zero column left/right of every row, one zero row between images.  A filter tap is then
//     a constant shift  (r-1)*SW + (s-1)  of p, valid across row and image boundaries alike;
//   * a workgroup owns BM consecutive p's and all BN output channels; per 32-channel chunk it stages the
//     BM + 2*SW + 2 input rows it needs ONCE (fp32 -> bf16 hi|lo, 160-byte rows: stride = 32 mod 64 bytes makes the
//     four 16-lane groups of ds_read_b128 conflict free) and all 9 taps read shifted rows of that tile;
//   * weights are split and re-ordered ONCE per weight update by buctd_conv3x3_bf16x3_prep into the exact stage
//     image the kernel consumes ([step][Co][32 hi | 32 lo] bf16; one (tap, 32-channel chunk) per step), so the
//     double-buffered B stage is a plain 16-byte copy; the 16-channel tail of C = 48 pairs two taps in one K = 32
//     MFMA (lanes 0-31 feed tap t, lanes 32-63 tap t+1), so no MFMA lanes are wasted on padding;
//   * the 4 waves tile the workgroup 4x1 (BN = 48/64) or 2x2 (BN = 96/128): every wave issues its 2*(MF+NF)
//     ds_read_b128 up front and then MF*NF*3 back-to-back MFMAs; the step loop is fully unrolled per chunk;
//   * workgroups are renumbered so that each XCD (own L2) walks a contiguous run of position tiles;
//   * epilogue as in conv.hip: bias, Welford BN partials (+ per-group valid-row counts, since pad positions are
//     skipped), eval-BN scale/shift, residual, ReLU.
#include "common.h"
#include <hip/hip_ext.h>
#include <vector>
#include "../../include/buctd_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define ROWB 160           // bytes per LDS row: 32 bf16 hi | 32 bf16 lo | 32 B pad
#define CK 32              // channels per chunk

struct C3Args {
  const float* x;
  const unsigned char* wp;   // prepared weight image: [steps][Co][128 B = 32 bf16 hi | 32 bf16 lo]
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  float* stats;
  int* counts;
  int N, H, W, Ci, Co;
  int SW, IB, P;       // padded row width, padded image block, total padded positions
  int relu, na;        // na: 32-row staging passes per chunk = ceil((BM + 2*SW + 2) / 32)
  unsigned ib_mul, ib_sh, sw_mul, sw_sh;   // n / d == mulhi(n, mul) >> sh for 0 <= n < 2^31 (Granlund-Montgomery)
};

__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(__umulhi((unsigned)n, mul) >> sh);
}

__device__ __forceinline__ void split_store(unsigned char* row, int c, f32x4 v) {
  // channels c..c+3 of one row: hi at byte 2c, lo at 64 + 2c
  u16x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)v[j];
    const __bf16 l = (__bf16)(v[j] - (float)h);
    hi[j] = __builtin_bit_cast(unsigned short, h);
    lo[j] = __builtin_bit_cast(unsigned short, l);
  }
  *reinterpret_cast<u16x4*>(row + 2 * c) = hi;
  *reinterpret_cast<u16x4*>(row + 64 + 2 * c) = lo;
}

#define MAX_SW 75          // W <= 73: the staged tile is at most BM + 152 rows

template <int V>
struct IC { static constexpr int value = V; };

template <int MF, int NF, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16x3_kernel(C3Args p) {
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int PA = (BM + 2 * MAX_SW + 2 + 31) / 32;   // float4 loads per thread for one A chunk (8 per row), max
  constexpr int PB = (BN * 8 + 255) / 256;              // 16-byte pieces per thread for one B step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int na = p.na;
  unsigned char* At = smem;                              // [na*32][ROWB]
  unsigned char* Bt = smem + (size_t)na * 32 * ROWB;     // [2][BN][ROWB]

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
  const int c4 = (t & 7) * 4;                    // this thread's 4-channel slot inside a 32-channel chunk

  // global element offset (channel 0) of every staged row this thread fills; -1 = zero row
  int goff[PA];
#pragma unroll
  for (int q = 0; q < PA; ++q) {
    const int row = (t >> 3) + 32 * q;
    const int pp = p0 - halo + row;
    int o = -1;
    if (q < na && pp >= 0 && pp < p.P) {
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
      if (q < na) {
        const bool ok = goff[q] >= 0 && c4 < cw * 16;
        areg[q] = *reinterpret_cast<const f32x4*>(p.x + (ok ? goff[q] + c0 + c4 : 0));
      }
  };
  auto store_a = [&](int cw) {
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (q < na) {
        const int row = (t >> 3) + 32 * q;
        const bool ok = goff[q] >= 0 && c4 < cw * 16;
        split_store(At + (size_t)row * ROWB, c4, ok ? areg[q] : (f32x4){0.f, 0.f, 0.f, 0.f});
      }
  };
  // B: one step image is [Co][128 B]; this thread copies 16-byte piece (t & 7) of rows (t >> 3) + 32 q
  int bsrc[PB], bdst[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int nl = (t >> 3) + 32 * q;
    const bool ok = nl < BN;
    bsrc[q] = ok ? (n0 + nl) * 128 + (t & 7) * 16 : 0;
    bdst[q] = ok ? nl * ROWB + (t & 7) * 16 : (nl & 15) * ROWB + 128 + (t & 1) * 16;   // idle threads: row pad bytes
  }
  const long step_bytes = (long)p.Co * 128;
  auto load_b = [&](int gs) {
    const unsigned char* src = p.wp + gs * step_bytes;
#pragma unroll
    for (int q = 0; q < PB; ++q) breg[q] = *reinterpret_cast<const f32x4*>(src + bsrc[q]);
  };
  auto store_b = [&](int buf) {
    unsigned char* dst = Bt + (size_t)buf * BN * ROWB;
#pragma unroll
    for (int q = 0; q < PB; ++q) *reinterpret_cast<f32x4*>(dst + bdst[q]) = breg[q];
  };

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nfull = p.Ci / CK, tail = (p.Ci % CK) ? 1 : 0;
  const int nchunks = nfull + tail, last_step = nfull * 9 + tail * 5 - 1;
  const unsigned char* abase = At + (size_t)(wave_m * MF * 16 + i16) * ROWB + (g & 1) * 16;
  const unsigned char* bbase = Bt + (size_t)(wave_n * NF * 16 + i16) * ROWB + g * 16;
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
      const int tap1 = CW == 2 ? s : (2 * s + 1 < 9 ? 2 * s + 1 : 2 * s);   // tap 9 of the tail: its weights are zero
      const int o0 = ((tap0 / 3) * p.SW + tap0 % 3) * ROWB;
      const int o1 = ((tap1 / 3) * p.SW + tap1 % 3) * ROWB + (CW == 2 ? 32 : 0);
      const unsigned char* ap = abase + (lowk ? o0 : o1);
      const unsigned char* bp = bbase + (size_t)buf * BN * ROWB;
      bf16x8 ah[MF], al[MF], bh[NF], bl[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        bh[nf] = *reinterpret_cast<const bf16x8*>(bp + nf * 16 * ROWB);
        bl[nf] = *reinterpret_cast<const bf16x8*>(bp + nf * 16 * ROWB + 64);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        ah[mf] = *reinterpret_cast<const bf16x8*>(ap + mf * 16 * ROWB);
        al[mf] = *reinterpret_cast<const bf16x8*>(ap + mf * 16 * ROWB + 64);
      }
      // small terms first, then hi*hi; consecutive MFMAs hit different accumulators
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mf], bh[nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mf], bl[nf], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mf], bh[nf], acc[mf][nf], 0, 0, 0);
      }
      store_b(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      ++gs;
    }
  };

  load_a(0, nfull > 0 ? 2 : 1);
  load_b(0);
  store_a(nfull > 0 ? 2 : 1);
  store_b(0);
  if (nchunks > 1) load_a(CK, nfull > 1 ? 2 : 1);   // in flight during the first chunk
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int cw = ch < nfull ? 2 : 1;
    if (ch > 0) {
      store_a(cw);                                   // everybody left the previous tile at the last step's barrier
      if (ch + 1 < nchunks) load_a((ch + 1) * CK, ch + 1 < nfull ? 2 : 1);
      __syncthreads();
    }
    if (cw == 2) run_chunk(IC<2>{});
    else run_chunk(IC<1>{});
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  // accumulator (mf, nf, reg): row = wave_m*MR + mf*16 + (lane>>4)*4 + reg, col = wave_n*NF*16 + nf*16 + (lane&15).
  // The tile goes through LDS (the A tile is dead after the last step's barrier) so that every output row leaves as
  // 16-byte pieces of one contiguous run instead of 64-byte column slivers of four rows.
  constexpr int MR = MF * 16;              // rows of this wave's tile
  constexpr int LD = NF * 16 + 4;          // staging row stride in floats: 4*LD = 16 mod 64 keeps the writes conflict-free
  constexpr int EP = MF >= 2 ? 2 : 1;      // 16-row fragments staged per pass
  float* stg = reinterpret_cast<float*>(smem) + wave * (EP * 16 * LD);
  int* rowoff = reinterpret_cast<int*>(smem) + 4 * EP * 16 * LD + wave * 64;
  int myoff = -1;                          // lane r: element offset of output row r of this wave (-1: pad position)
  if (lane < MR) {
    const int pp = p0 + wave_m * MR + lane;
    if (pp < p.P) {
      const int n = fast_div(pp, p.ib_mul, p.ib_sh);
      const int rem = pp - n * p.IB;
      const int yy = fast_div(rem, p.sw_mul, p.sw_sh), xx = rem - yy * p.SW;
      if (n < p.N && yy >= 1 && xx >= 1 && xx <= p.W) myoff = ((n * p.H + yy - 1) * p.W + xx - 1) * p.Co;
    }
  }
  const unsigned long long vmask = __ballot(myoff >= 0);
  const int cnt = __popcll(vmask);
  rowoff[lane] = myoff;
  const int grp = bx * WM + wave_m;
  if (p.stats && p.counts && by == 0 && wave_n == 0 && lane == 0) p.counts[grp] = cnt;
  const int ncol0 = n0 + wave_n * NF * 16;
  float bv[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) bv[nf] = p.bias ? p.bias[ncol0 + nf * 16 + i16] : 0.f;
  if (p.stats) {
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float s1 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          s1 += ((vmask >> (mf * 16 + g * 4 + rg)) & 1) ? acc[mf][nf][rg] + bv[nf] : 0.f;
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = s1 * inv_cnt;
      float s2 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float d = acc[mf][nf][rg] + bv[nf] - mean;
          s2 += ((vmask >> (mf * 16 + g * 4 + rg)) & 1) ? d * d : 0.f;
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) {
        const int n = ncol0 + nf * 16 + i16;
        *reinterpret_cast<float2*>(p.stats + ((long)grp * p.Co + n) * 2) = make_float2(mean, s2);
      }
    }
  }
#pragma unroll
  for (int ps = 0; ps < MF / EP; ++ps) {
    if (ps) __syncthreads();
#pragma unroll
    for (int e = 0; e < EP; ++e)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          stg[(e * 16 + g * 4 + rg) * LD + nf * 16 + i16] = acc[ps * EP + e][nf][rg] + bv[nf];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EP * NF; ++k) {
      const int item = lane + 64 * k;
      const int row = item / (NF * 4), c4 = item - row * (NF * 4);
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * LD + c4 * 4);
      const int off = rowoff[ps * EP * 16 + row];
      const int n = ncol0 + c4 * 4;
      if (off >= 0) {
        if (p.scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[j] + sh[j];
        }
        if (p.res) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + off + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(p.out + off + n) = v;
      }
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
  split_store(out + rown * 128, c4, v);
}

// the same for many filters in one launch (all prepared images of a model after an optimizer step): items live in
// device memory, thread -> item by binary search over the running piece count
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
  split_store(reinterpret_cast<unsigned char*>(it.wprep) + rown * 128, c4, v);
}

// ---------------------------------------------------------------------------------------------- host ----
struct C3Plan { int MF, NF, WM, WN, BM, BN, na; size_t lds; };

// magic for unsigned division of n < 2^31 by d (2 <= d < 2^31): q = mulhi(n, mul) >> sh
static void magic_u32(unsigned d, unsigned* mul, unsigned* sh) {
  if (d == 1) { *mul = 0xFFFFFFFFu; *sh = 0; return; }   // never taken: IB and SW are >= 3
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                  // l = ceil(log2 d)
  // n < 2^31: m = ceil(2^(31+l) / d) fits in 32 bits and q = (n*m) >> (31+l) is exact
  const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
  *mul = (unsigned)m;
  *sh = l - 1;                                  // mulhi already shifts by 32: total shift 31 + l
}

static int c3_steps(int Kc) { return (Kc / CK) * 9 + ((Kc % CK) ? 5 : 0); }

static bool c3_plan(int N, int H, int W, int Ci, int Co, C3Plan* pl) {
  if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0 || W + 2 > MAX_SW) return false;
  const long P = (long)N * (H + 1) * (W + 2) + (W + 2);
  int nf, wn;
  if (Co % 96 == 0) { nf = 3; wn = 2; }
  else if (Co % 48 == 0) { nf = 3; wn = 1; }
  else if (Co % 128 == 0) { nf = 4; wn = 2; }
  else if (Co % 64 == 0) { nf = 4; wn = 1; }
  else if (Co % 32 == 0) { nf = 2; wn = 1; }
  else { nf = 1; wn = 1; }
  // wide column tiles (2x2 waves, BN = 96) only pay while they still fill the machine: the low-resolution branches
  // (192 ch @24x18, 384 ch @12x9) run faster as 4x1 waves with BN = 48 and twice as many workgroups (measured)
  if (Co % 96 == 0 && ((P + 127) / 128) * (Co / 96) < 320) { nf = 3; wn = 1; }
  int wm = 4 / wn, bn = wn * nf * 16;
  // largest position tile that still gives every CU work (256 CUs, 2 resident workgroups each)
  int mf = 1;
  const int cand[2] = {4, 2};
  for (int i = (nf == 4 ? 1 : 0); i < 2; ++i) {   // 64-row x 64-column wave tiles would spill
    const long blocks = ((P + wm * cand[i] * 16 - 1) / (wm * cand[i] * 16)) * (Co / bn);
    if (blocks >= 320) { mf = cand[i]; break; }
  }
  pl->MF = mf; pl->NF = nf; pl->WM = wm; pl->WN = wn; pl->BM = wm * mf * 16; pl->BN = bn;
  pl->na = (pl->BM + 2 * (W + 2) + 2 + 31) / 32;
  const size_t stage = (size_t)4 * (mf >= 2 ? 2 : 1) * 16 * (nf * 16 + 4) * 4 + 4 * 64 * 4;   // epilogue staging + row offsets
  while ((size_t)pl->na * 32 * ROWB < stage) ++pl->na;
  pl->lds = (size_t)pl->na * 32 * ROWB + (size_t)2 * pl->BN * ROWB;
  return pl->lds <= 160 * 1024;
}

extern "C" int buctd_conv3x3_bf16x3_supported(int N, int H, int W, int Ci, int Co) {
  C3Plan pl;
  return c3_plan(N, H, W, Ci, Co, &pl) ? 1 : 0;
}

extern "C" int buctd_conv3x3_bf16x3_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups,
                                                 int* rows_per_group) {
  C3Plan pl;
  BUCTD_CHECK_ARG(ngroups && rows_per_group && c3_plan(N, H, W, Ci, Co, &pl),
                  "buctd_conv3x3_bf16x3_stats_groups: unsupported shape");
  const long P = (long)N * (H + 1) * (W + 2) + (W + 2);
  *ngroups = ceil_div(P, pl.BM) * pl.WM;
  *rows_per_group = pl.MF * 16;
  return BUCTD_OK;
}

extern "C" size_t buctd_conv3x3_bf16x3_prep_bytes(int Ci, int Co, int flip) {
  if (Ci <= 0 || Co <= 0 || Ci % 16 != 0 || Co % 16 != 0) return 0;
  const int Kc = flip ? Co : Ci, Nc = flip ? Ci : Co;
  return (size_t)c3_steps(Kc) * Nc * 128;
}

extern "C" int buctd_conv3x3_bf16x3_prep(int Ci, int Co, const float* w, int flip, void* wprep, void* stream) {
  BUCTD_CHECK_ARG(w && wprep, "buctd_conv3x3_bf16x3_prep: null pointer");
  BUCTD_CHECK_ARG(Ci > 0 && Co > 0 && Ci % 16 == 0 && Co % 16 == 0, "buctd_conv3x3_bf16x3_prep: Ci=%d Co=%d must be multiples of 16",
                  Ci, Co);
  const int Kc = flip ? Co : Ci, Nc = flip ? Ci : Co;
  const long pieces = (long)c3_steps(Kc) * Nc * 8;
  hipLaunchKernelGGL(conv3x3_prep_kernel, dim3(ceil_div(pieces, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned char*)wprep, Kc, Nc, flip ? 1 : 0, pieces);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3_prep");
  return BUCTD_OK;
}

extern "C" int buctd_conv3x3_bf16x3_prep_batched(const buctd_c3_prep_item* items_device, int n, long total_pieces,
                                                 void* stream) {
  BUCTD_CHECK_ARG(items_device && n > 0 && total_pieces > 0, "buctd_conv3x3_bf16x3_prep_batched: bad argument");
  hipLaunchKernelGGL(conv3x3_prep_batched_kernel, dim3(ceil_div(total_pieces, 256)), dim3(256), 0, (hipStream_t)stream,
                     items_device, n, total_pieces);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3_prep_batched");
  return BUCTD_OK;
}

// Optional live timing of the launches of one shape (bench.py's roofline figure): HIP events attached to the dispatch
// itself (hipExtLaunchKernelGGL start/stop events) time the kernel exactly as a kernel trace does - events recorded
// around the launch on the stream would also count marker handling and whatever the stream waits for.
struct C3Timing {
  bool on = false;
  int N = 0, H = 0, W = 0, Ci = 0, Co = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};
static C3Timing g_c3_timing;

template <int MF, int NF, int WM, int WN>
static int c3_launch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
  static bool attr_set = false;
  auto fn = conv3x3_bf16x3_kernel<MF, NF, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      buctd_set_error("conv3x3_bf16x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      return BUCTD_ELAUNCH;
    }
    attr_set = true;
  }
  dim3 grid(ceil_div(a.P, pl.BM), a.Co / pl.BN);
  C3Timing& tm = g_c3_timing;
  if (tm.on && a.N == tm.N && a.H == tm.H && a.W == tm.W && a.Ci == tm.Ci && a.Co == tm.Co) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
      hipExtLaunchKernelGGL(fn, grid, dim3(256), (uint32_t)pl.lds, st, e0, e1, 0, a);
      tm.events.emplace_back(e0, e1);
      BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3");
      return BUCTD_OK;
    }
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_bf16x3");
  return BUCTD_OK;
}

extern "C" int buctd_conv3x3_bf16x3_timing_begin(int N, int H, int W, int Ci, int Co) {
  C3Timing& tm = g_c3_timing;
  for (auto& ev : tm.events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  tm.events.clear();
  tm.N = N; tm.H = H; tm.W = W; tm.Ci = Ci; tm.Co = Co;
  tm.on = true;
  return BUCTD_OK;
}

extern "C" int buctd_conv3x3_bf16x3_timing_end(double* total_us, int* launches) {
  BUCTD_CHECK_ARG(total_us && launches, "buctd_conv3x3_bf16x3_timing_end: null pointer");
  C3Timing& tm = g_c3_timing;
  tm.on = false;
  double tot = 0.0;
  int n = 0;
  for (auto& ev : tm.events) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      tot += (double)ms * 1e3;
      ++n;
    }
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  tm.events.clear();
  *total_us = tot;
  *launches = n;
  return BUCTD_OK;
}

static int c3_dispatch(const C3Args& a, const C3Plan& pl, hipStream_t st) {
#define C3_CASE(mf, nf, wm, wn) \
  if (pl.MF == mf && pl.NF == nf && pl.WN == wn) return c3_launch<mf, nf, wm, wn>(a, pl, st);
#define C3_MF(nf, wm, wn) C3_CASE(4, nf, wm, wn) C3_CASE(2, nf, wm, wn) C3_CASE(1, nf, wm, wn)
  C3_MF(1, 4, 1) C3_MF(2, 4, 1) C3_MF(3, 4, 1) C3_MF(3, 2, 2)
  C3_CASE(2, 4, 4, 1) C3_CASE(1, 4, 4, 1) C3_CASE(2, 4, 2, 2) C3_CASE(1, 4, 2, 2)
#undef C3_MF
#undef C3_CASE
  buctd_set_error("conv3x3_bf16x3: no kernel for MF=%d NF=%d WN=%d", pl.MF, pl.NF, pl.WN);
  return BUCTD_EINVAL;
}

// x: [N][H][W][Ci] -> y: [N][H][W][Co], wprep from buctd_conv3x3_bf16x3_prep.  Forward: prep(Ci, Co, w, 0).
// Data gradient of a forward conv (CiF -> CoF): x = dy ([N][H][W][CoF]), y = dx ([N][H][W][CiF]), i.e. this call's
// Ci = CoF, Co = CiF, and wprep = prep(CiF, CoF, w, 1).
extern "C" int buctd_conv3x3_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                    const float* bias, const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, float* stats_partials, int* stats_counts, void* stream) {
  C3Plan pl;
  BUCTD_CHECK_ARG(x && wprep && y, "buctd_conv3x3_bf16x3: null tensor pointer");
  BUCTD_CHECK_ARG(c3_plan(N, H, W, Ci, Co, &pl), "buctd_conv3x3_bf16x3: unsupported shape N%d H%d W%d Ci%d Co%d", N, H,
                  W, Ci, Co);
  BUCTD_CHECK_ARG((scale == nullptr) == (shift == nullptr), "buctd_conv3x3_bf16x3: scale and shift go together");
  BUCTD_CHECK_ARG((stats_partials == nullptr) == (stats_counts == nullptr),
                  "buctd_conv3x3_bf16x3: stats partials and counts go together");
  C3Args a;
  a.x = x; a.wp = (const unsigned char*)wprep; a.out = y; a.bias = bias; a.scale = scale; a.shift = shift;
  a.res = residual; a.stats = stats_partials; a.counts = stats_counts;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.SW = W + 2; a.IB = (H + 1) * (W + 2);
  const long P = (long)N * a.IB + a.SW;
  BUCTD_CHECK_ARG(P < 2147483647L && (long)N * H * W * (Ci > Co ? Ci : Co) < 2147483647L,
                  "buctd_conv3x3_bf16x3: tensor too large");
  a.P = (int)P;
  a.relu = relu; a.na = pl.na;
  magic_u32((unsigned)a.IB, &a.ib_mul, &a.ib_sh);
  magic_u32((unsigned)a.SW, &a.sw_mul, &a.sw_sh);
  return c3_dispatch(a, pl, (hipStream_t)stream);
}
