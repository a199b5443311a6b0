// Shared pieces of the bf16-split convolution kernels (conv3x3.hip: 3x3 stride 1 over an LDS-resident tile;
// conv_gather_x6.hip: stride-2 3x3 / 1x1 / data gradients as a gathered implicit GEMM): piece geometry, the argument block,
// the fp32 -> bf16-piece split, and the epilogue (bias, Welford BatchNorm partials, eval-BN, residual, ReLU, store).
#pragma once
#include "common.h"
#include <stdlib.h>
#include "../../include/buctd_hip.h"
#include "bn_acc.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define CK 32              // NP = 2: channels per full chunk

// per-mode geometry
template <int NP> struct Geo;
template <> struct Geo<2> {
  static constexpr int ROWB = 160;   // LDS bytes per A row: 32 hi | 32 lo | 32 B pad
  static constexpr int PST = 64;     // byte stride between the pieces of an A row
  static constexpr int CPR = 8;      // float4 per staged row (32 channels)
  static constexpr int BROW = 128;   // global bytes per (step, co) row of the prepared image
  static constexpr int BLDS = 160;   // LDS stride of a B row
};
template <> struct Geo<3> {
  static constexpr int ROWB = 96;    // 16 h | 16 m | 16 l
  static constexpr int PST = 32;
  static constexpr int CPR = 4;      // 16 channels
  static constexpr int BROW = 192;   // 32 h | 32 m | 32 l k-slots
  static constexpr int BLDS = 224;   // + 32 B pad: 224 = 32 mod 64
};

struct C3Args {
  const float* x;
  const unsigned char* wp;   // prepared weight image: [steps][Co][NP pieces x 32 bf16 k-slots]
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  float* stats;
  int* counts;
  int N, H, W, Ci, Co;
  int SW, IB, P;       // padded row width, padded image block, total padded positions
  int relu, na;        // na: 32-row groups of the staged tile = ceil((BM + 2*SW + 2) / 32)
  // optional BatchNorm(+ReLU) of the PRODUCER applied to the input while it is staged (bf16x6 kernel only): the input
  // tensor is the producer's raw convolution output z and the kernel consumes relu((z - mean) * (invstd * gamma) + beta),
  // the exact expression of bn_apply_kernel - the normalised tensor never exists in HBM
  const float* in_mean;
  const float* in_invstd;
  const float* in_gamma;
  const float* in_beta;
  int in_relu;
  int col_major;       // bf16x6 kernel: consecutive workgroups of an XCD share the COLUMN tile (see the kernel)
  unsigned ib_mul, ib_sh, sw_mul, sw_sh;   // n / d == mulhi(n, mul) >> sh for 0 <= n < 2^31 (Granlund-Montgomery)
  // output map of the gathered kernels (conv_gather_x6.hip; omap = 0: the H x W grid itself): grid pixel (y, x) is written
  // to pixel (y * ost + oy0, x * ost + ox0) of an oH x oW tensor - the parity classes of a stride-2 data gradient
  int omap, oH, oW, ost, oy0, ox0;
  // optional by-product of a DATA-GRADIENT launch (conv3x3.hip only): the reduction pass of the BatchNorm backward that
  // consumes this launch's output g = out (after the residual was added) - per row group the partial sums
  //   s1[c] = sum_rows m * g,   s2[c] = sum_rows m * g * (z - mean[c]) * invstd[c],   m = ReLU mask of the BatchNorm's
  // forward output (bs_y > 0 where given, else rebuilt as (z - mean) * (invstd * gamma) + beta > 0, bn_apply's expression)
  // - exactly the sums of bn_bwd_reduce2_kernel, formed while the tile is on its way out instead of by a kernel that
  // reads g, z and y again.  bs_part: [groups][2][Co] floats, groups as the forward statistics (bx * WM + wave_m).
  const float* bs_z;
  const float* bs_y;
  const float* bs_mean;
  const float* bs_invstd;
  const float* bs_gamma;
  const float* bs_beta;
  float* bs_part;
  // The same two by-products as exact integer accumulators (bn_acc.h) instead of per-group partials: the consumer of the
  // statistics decodes two numbers per channel in its prologue and no finalize launch sits between the two kernels.
  //   stats_acc: forward statistics (sum z, sum z^2) of this launch's output;  bs_acc: the BatchNorm-backward sums above.
  long long* stats_acc;
  long long* bs_acc;
  // in_acc.acc != null: the producer's statistics (in_mean / in_invstd) are decoded from ITS accumulator in this launch's
  // prologue; the workgroup of tile (0, 0) also writes them out (in_acc.mean_out / invstd_out, running statistics)
  BnAccFwd in_acc;
};

// several convolutions in ONE launch (conv3x3.hip: conv3x3_x6_group_kernel; conv3x3_lean.hip: the train-mode kernels)
#define C3G_MAX 4
struct C3Group {
  C3Args conv[C3G_MAX];
  int nconv;
  int tiles[C3G_MAX];      // workgroup tiles of convolution c
  int gx[C3G_MAX];         // ... position tiles
  int gy[C3G_MAX];         // ... column tiles
  int variant[C3G_MAX];    // index into the variant list of the kernel family
};

// LDS the epilogue needs at the front of `smem`: row staging + row offsets (<= 19.5 KB), the lane reduction of the
// BatchNorm-backward by-product (<= 24 KB, reuses the staging area), and behind them the cross-wave exchange of the
// accumulator paths (WM * BN <= 256 pairs of doubles)
#define C3_EPI_EXCH_OFF 24576
#define C3_EPI_LDS (C3_EPI_EXCH_OFF + 256 * 16)

__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(__umulhi((unsigned)n, mul) >> sh);
}

// channels c..c+3 of one row -> NP bf16 pieces, piece q at byte q*PST + 2c.  The residual subtractions are exact.
// The same pieces (same roundings, same exact residuals) formed two channels at a time: v_cvt_pk_bf16_f32 delivers a packed
// pair that IS the stored word, so the packing shifts / masks of the element-wise form disappear (22 instead of 28 VALU
// instructions per four channels with NP = 3).  Used by the gathered kernels; the 3x3 kernels keep the form they were tuned with.
typedef float c3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 c3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned c3_u32x2 __attribute__((ext_vector_type(2)));
template <int NP, int PST>
__device__ __forceinline__ void split_store_pk(unsigned char* row, int c, f32x4 v) {
  c3_f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, c3_bf16x2));
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, c3_bf16x2));
    *reinterpret_cast<c3_u32x2*>(row + q * PST + 2 * c) = (c3_u32x2){a, b};
    if (q + 1 < NP) {
      lo -= (c3_f32x2){__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u)};
      hi -= (c3_f32x2){__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
    }
  }
}

template <int NP, int PST>
__device__ __forceinline__ void split_store(unsigned char* row, int c, f32x4 v) {
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    u16x4 pc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16 h = (__bf16)v[j];
      pc[j] = __builtin_bit_cast(unsigned short, h);
      v[j] -= (float)h;
    }
    *reinterpret_cast<u16x4*>(row + q * PST + 2 * c) = pc;
  }
}

#define MAX_SW 75          // W <= 74: the staged tile is at most BM + 152 rows

// Row width of the zero-padded flattened position space  p = n*IB + (y+1)*SW + (x+1),  IB = (H+1)*SW,  P = N*IB + SW:
// ONE pad column per row (xx = 0).  In the flattened space the position right of a row's last pixel IS the next row's
// pad column, so it serves as the right pad of row y and the left pad of row y + 1 at once (a second pad column, as in the
// first rounds, only added 1/(W+2) of positions that are multiplied and thrown away - and at N = 32, 96x72 made the tile
// grid one tile too long for a balanced single round of workgroups).  Pixel (y, x) of image n: yy = y + 1 in 1..H,
// xx = x + 1 in 1..W; yy = 0 is the pad row above image n = the pad row below image n - 1.
static inline int c3_row_width(int W) { return W + 1; }

template <int V>
struct IC { static constexpr int value = V; };

// ---- epilogue (shared by both kernels) -----------------------------------------------------------------------
// accumulator (mf, nf, reg): row = wave_m*MR + mf*16 + (lane>>4)*4 + reg, col = wave_n*NF*16 + nf*16 + (lane&15).
// Called after a barrier that ends every read of the staged tiles: the tile area of `smem` is reused for staging.
template <int MF, int NF, int WM, int WN>
__device__ __forceinline__ void c3_epilogue(const C3Args& p, f32x4 (&acc)[MF][NF], unsigned char* smem, int bx, int by,
                                            int p0, int n0) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  // The tile goes through LDS (the A tile is dead by now) so that every output row leaves as 16-byte pieces of one
  // contiguous run instead of 64-byte column slivers of four rows.
  constexpr int MR = MF * 16;              // rows of this wave's tile (<= 128)
  constexpr int RH = (MR + 63) / 64;       // 64-row halves: lane r owns rows r and r + 64
  constexpr int LD = NF * 16 + 4;          // staging row stride in floats: 4*LD = 16 mod 64 keeps the writes conflict-free
  constexpr int EP = 1;                    // 16-row fragments staged per pass (one: the look-ahead registers of the
                                           // store loop scale with it, and a pass costs no workgroup barrier)
  static_assert(MR <= 128, "c3_epilogue: at most 128 rows per wave");
  float* stg = reinterpret_cast<float*>(smem) + wave * (EP * 16 * LD);
  int* rowoff = reinterpret_cast<int*>(smem) + 4 * EP * 16 * LD + wave * 128;
  // element offset of output row r of this wave (-1: pad position)
  unsigned long long vmask[RH];
  int cnt = 0;
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    int myoff = -1;
    const int r = lane + 64 * h;
    if (r < MR) {
      const int pp = p0 + wave_m * MR + r;
      if (pp < p.P) {
        const int n = fast_div(pp, p.ib_mul, p.ib_sh);
        const int rem = pp - n * p.IB;
        const int yy = fast_div(rem, p.sw_mul, p.sw_sh), xx = rem - yy * p.SW;
        if (n < p.N && yy >= 1 && xx >= 1 && xx <= p.W)
          myoff = p.omap ? ((n * p.oH + (yy - 1) * p.ost + p.oy0) * p.oW + (xx - 1) * p.ost + p.ox0) * p.Co
                         : ((n * p.H + yy - 1) * p.W + xx - 1) * p.Co;
      }
    }
    vmask[h] = __ballot(myoff >= 0);
    cnt += __popcll(vmask[h]);
    rowoff[r] = myoff;
  }
  auto valid = [&](int row) -> bool { return (vmask[row >> 6] >> (row & 63)) & 1ull; };
  const int grp = bx * WM + wave_m;
  if (p.stats && p.counts && by == 0 && wave_n == 0 && lane == 0) p.counts[grp] = cnt;
  const int ncol0 = n0 + wave_n * NF * 16;
  float bv[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) bv[nf] = p.bias ? p.bias[ncol0 + nf * 16 + i16] : 0.f;
  double2* exch = reinterpret_cast<double2*>(smem + C3_EPI_EXCH_OFF);      // [wave_m][BN]
  if (p.stats || p.stats_acc) {
    const float inv_cnt = cnt > 0 ? 1.f / (float)cnt : 0.f;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float s1 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) s1 += valid(mf * 16 + g * 4 + rg) ? acc[mf][nf][rg] + bv[nf] : 0.f;
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = s1 * inv_cnt;
      float s2 = 0.f;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const float d = acc[mf][nf][rg] + bv[nf] - mean;
          s2 += valid(mf * 16 + g * 4 + rg) ? d * d : 0.f;
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) {
        const int n = ncol0 + nf * 16 + i16;
        if (p.stats) *reinterpret_cast<float2*>(p.stats + ((long)grp * p.Co + n) * 2) = make_float2(mean, s2);
        if (p.stats_acc) {
          // this wave's rows as (sum, sum of squares) in fp64 - the terms bn_finalize_kernel forms from a Welford partial
          const double nn = (double)cnt, m = (double)mean;
          exch[wave_m * (WN * NF * 16) + wave_n * NF * 16 + nf * 16 + i16] = make_double2(nn * m, (double)s2 + nn * m * m);
        }
      }
    }
    if (p.stats_acc) {
      // the workgroup's waves in a fixed order, then ONE exact integer addition per channel and sum
      __syncthreads();
      if (t < WN * NF * 16) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          const double2 v = exch[w * (WN * NF * 16) + t];
          a1 += v.x;
          a2 += v.y;
        }
        bnacc_add(p.stats_acc, p.Co, bnacc_shard(), n0 + t, a1, a2);
      }
    }
  }
  // BatchNorm-backward partial sums of the outgoing tile (C3Args::bs_part).  Item k of a lane has column quad
  // c4 = (lane + 64 k) % (NF * 4): NACC distinct quads per lane, one accumulator pair each.
  constexpr int Q4 = NF * 4;
  constexpr int NACC = (64 % Q4 == 0) ? 1 : 3;       // NF = 3: 64 = 4 mod 12 -> three quads per lane; else one
  static_assert(NF <= 4 && (64 % Q4 == 0 || Q4 == 12), "c3_epilogue: unexpected column count for the bs reduction");
  f32x4 bs1[NACC], bs2[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) bs1[j] = bs2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool bs_on = p.bs_part != nullptr || p.bs_acc != nullptr;
  const bool bs_rebuild = bs_on && p.bs_y == nullptr;
  // Every wave stages through ITS OWN slice of LDS (stg, rowoff) and the LDS executes one wave's operations in order, so
  // the passes need no workgroup barrier - only the compiler must keep the stores in front of the loads (wave_barrier).
#define C3_EPI_SYNC() __builtin_amdgcn_wave_barrier()
  // The tensors the store loop reads beside the tile - the residual, and z / y of the BatchNorm-backward by-product - travel
  // ONE PASS AHEAD through two register sets: loaded inside the loop they cost a memory round trip per pass, and in a launch
  // that is one round of workgroups every CU sits in its epilogue at the same time, so nothing else hides it (measured at
  // 48 channels: + 6 us for the residual, + 13 / + 22 us for the by-product).  EP * NF items per lane and pass.
  constexpr int NI = EP * NF;
  constexpr int NPASS = MF / EP;
  f32x4 pre_r[2][NI], pre_z[2][NI], pre_y[2][NI];
  const bool has_res = p.res != nullptr;
  const bool has_y = bs_on && !bs_rebuild;
  C3_EPI_SYNC();     // rowoff is complete
  auto issue = [&](int ps, int buf) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const int item = lane + 64 * k;
      const int row = item / (NF * 4), c4 = item - row * (NF * 4);
      const int off = rowoff[ps * EP * 16 + row];
      const int o = off >= 0 ? off + ncol0 + c4 * 4 : 0;       // pad rows: any valid address, the value is not used
      if (has_res) pre_r[buf][k] = *reinterpret_cast<const f32x4*>(p.res + o);
      if (bs_on) pre_z[buf][k] = *reinterpret_cast<const f32x4*>(p.bs_z + o);
      if (has_y) pre_y[buf][k] = *reinterpret_cast<const f32x4*>(p.bs_y + o);
    }
  };
  if (has_res || bs_on) issue(0, 0);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    if (ps) C3_EPI_SYNC();
    if ((has_res || bs_on) && ps + 1 < NPASS) issue(ps + 1, (ps + 1) & 1);
#pragma unroll
    for (int e = 0; e < EP; ++e)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          stg[(e * 16 + g * 4 + rg) * LD + nf * 16 + i16] = acc[ps * EP + e][nf][rg] + bv[nf];
    C3_EPI_SYNC();
#pragma unroll
    for (int k = 0; k < NI; ++k) {
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
        if (has_res) {
          const f32x4 r = pre_r[ps & 1][k];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(p.out + off + n) = v;
        if (bs_on) {
          // the arithmetic of bn_bwd_reduce2_kernel's body, element for element
          const f32x4 zz = pre_z[ps & 1][k];
          const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bs_mean + n);
          const f32x4 is = *reinterpret_cast<const f32x4*>(p.bs_invstd + n);
          f32x4 yy;
          if (bs_rebuild) {
            const f32x4 sc = is * *reinterpret_cast<const f32x4*>(p.bs_gamma + n);
            yy = (zz - mu) * sc + *reinterpret_cast<const f32x4*>(p.bs_beta + n);
          } else {
            yy = pre_y[ps & 1][k];
          }
          f32x4 gm = v;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(yy[j] > 0.f)) gm[j] = 0.f;
          bs1[k % NACC] += gm;
          bs2[k % NACC] += gm * (zz - mu) * is;
        }
      }
    }
  }
  if (bs_on) {
    // lanes -> column quads in a fixed order (deterministic): every lane parks its NACC pairs in LDS (the staging area is
    // dead after one more barrier), lane c < NF * 4 adds the entries whose quad is c, lanes in ascending order
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem) + (size_t)wave * (NACC * 64 * 2);
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      red[(j * 64 + lane) * 2 + 0] = bs1[j];
      red[(j * 64 + lane) * 2 + 1] = bs2[j];
    }
    __syncthreads();
    if (lane < Q4) {
      f32x4 a1 = (f32x4){0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        // entries (j, l) with (l + 64 j) % Q4 == lane: l = first, first + Q4, ...
        const int first = ((lane - (64 * j) % Q4) % Q4 + Q4) % Q4;
        for (int l = first; l < 64; l += Q4) {
          a1 += red[(j * 64 + l) * 2 + 0];
          a2 += red[(j * 64 + l) * 2 + 1];
        }
      }
      if (p.bs_part) {
        float* dst = p.bs_part + ((long)grp * 2) * p.Co + ncol0 + lane * 4;
        *reinterpret_cast<f32x4*>(dst) = a1;
        *reinterpret_cast<f32x4*>(dst + p.Co) = a2;
      }
      if (p.bs_acc) {
        // [wave_m][BN] pairs: the wave's fp32 sums of its rows, channel by channel
#pragma unroll
        for (int j = 0; j < 4; ++j)
          exch[wave_m * (WN * NF * 16) + wave_n * NF * 16 + lane * 4 + j] = make_double2((double)a1[j], (double)a2[j]);
      }
    }
    if (p.bs_acc) {
      __syncthreads();
      if (t < WN * NF * 16) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          const double2 v = exch[w * (WN * NF * 16) + t];
          a1 += v.x;
          a2 += v.y;
        }
        bnacc_add(p.bs_acc, p.Co, bnacc_shard(), n0 + t, a1, a2);
      }
    }
  }
}


static inline void magic_u32(unsigned d, unsigned* mul, unsigned* sh) {
  if (d == 1) { *mul = 0xFFFFFFFFu; *sh = 0; return; }   // never taken: IB and SW are >= 3
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                  // l = ceil(log2 d)
  // n < 2^31: m = ceil(2^(31+l) / d) fits in 32 bits and q = (n*m) >> (31+l) is exact
  const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;
  *mul = (unsigned)m;
  *sh = l - 1;                                  // mulhi already shifts by 32: total shift 31 + l
}
