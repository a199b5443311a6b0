// The train-mode specialisations of the bf16x6 3x3 convolution (conv3x3.hip: c3x6_tile is the general form).
//
// Why a second form.  Round 5 measured the general kernel at 2.8 VALU instructions per MFMA (PMC), and this round's
// micro-benchmark (scratch/ubench/mfma_valu2.hip) says what that costs on gfx950: a wavefront CANNOT issue VALU work in the
// shadow of its own v_mfma_f32_16x16x32_bf16 (16.6 cycles per MFMA alone, 16.6 + 4.0 V with V VALU instructions per MFMA),
// only the OTHER wavefront of the SIMD can - and only while the two are in different phases.  5300 VALU instructions next to
// 1890 MFMAs per wavefront and tile therefore cost 20 k of the tile's 50 k cycles outright.  The ISA of the general kernel
// shows where they come from: an epilogue that carries every run-time option through every 16-byte item (bias, eval-BN scale /
// shift, residual, ReLU, two BatchNorm-backward forms, output maps: 55-60 VALU + up to 40 register moves per item, 21 items per
// lane), two masked passes over the accumulators for the BatchNorm statistics (750), the element-wise fp32 -> bf16 split (28 per
// four channels where the packed form needs 22) with the input-BatchNorm select in front of it (29).
//
// The train step uses FIVE option sets (block.hip), each a compile-time MODE here:
//   STATS            conv1 forward: raw output + its BatchNorm statistics (integer accumulator, bn_acc.h)
//   STATS | IN_BN    conv2 forward: the producer's BatchNorm + ReLU applied while the input is staged
//   BS_REBUILD       conv2 data gradient: output + the sums of bn1's backward, ReLU mask rebuilt from z1
//   RES | BS_Y       conv1 data gradient inside a chain: + skip gradient, sums of the previous block's bn2 backward (mask from y)
//   RES              conv1 data gradient of the first block of a chain
// Everything else (bias, eval-mode scale / shift, ReLU on the output, partial-sum statistics, output maps) stays with the
// general kernel.  Outputs are bit-identical to the general kernel (same tile plans, same MFMA order, same epilogue arithmetic
// for the output and for the BatchNorm-backward sums); the forward statistics are formed differently (see c3l_epilogue).
#pragma once
#include "c3_common.h"

enum { C3M_IN_BN = 1, C3M_STATS = 2, C3M_RES = 4, C3M_BS_REBUILD = 8, C3M_BS_Y = 16 };

// Tensors are addressed through buffer descriptors (32-bit byte offsets, hardware bounds check): a pad row carries the offset
// C3_OOB, its loads return zeros and its stores are dropped - no select, no branch, no 64-bit address arithmetic per row.
// (Tensors of the train-mode kernels are below 2 GB: conv3x3.hip checks.)
#define C3_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t c3_rsrc(const void* ptr, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 c3_bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void c3_bstore(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), r, off, 0, 0);
}

// LDS behind the A buffers: [3][Ci] floats input-BatchNorm table | [4][BN] floats epilogue table
//   BS modes: mean, invstd, invstd * gamma, beta of the workgroup's BN output columns;  STATS: one pivot per wave and column
static inline size_t c3l_tab_bytes(int Ci, int BN) { return (size_t)3 * Ci * 4 + (size_t)4 * BN * 4; }

// ---- epilogue --------------------------------------------------------------------------------------------------------
// accumulator (mf, nf, reg): row = wave_m*MR + mf*16 + (lane>>4)*4 + reg, col = wave_n*NF*16 + nf*16 + (lane&15).
// As c3_epilogue the tile leaves through a per-wave LDS staging slice as 16-byte pieces of contiguous rows; what differs:
//   * the option set is a template argument: an item is ds_read_b128 + row offset + (residual) + store + its sums;
//   * forward statistics are summed from the STAGED rows, where a lane holds four columns of one row and pad rows are simply
//     skipped (no mask arithmetic), around a per-wave, per-column pivot pi = the wave's first REAL output row (any real row
//     is within a few sigma of the channel mean, so sum (v - pi)^2 keeps the accuracy of a centred sum where a raw
//     sum v^2 loses |mean|^2 / var of it): per lane <= MF terms in fp32, everything after that in fp64 -
//         sum v = S1 + n pi,   sum v^2 = S2 + 2 pi S1 + n pi^2   - and one exact integer addition per workgroup and channel.
template <int MF, int NF, int WM, int WN, int MODE>
__device__ __forceinline__ void c3l_epilogue(const C3Args& p, f32x4 (&acc)[MF][NF], unsigned char* smem, float* tab, int p0,
                                             int n0) {
  constexpr bool STATS = (MODE & C3M_STATS) != 0, RES = (MODE & C3M_RES) != 0;
  constexpr bool BSR = (MODE & C3M_BS_REBUILD) != 0, BSY = (MODE & C3M_BS_Y) != 0, BS = BSR || BSY;
  static_assert(!(STATS && BS), "c3l_epilogue: forward statistics and backward sums never meet in one launch");
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  constexpr int MR = MF * 16;
  constexpr int RH = (MR + 63) / 64;
  constexpr int LD = NF * 16 + 4;
  constexpr int BN = WN * NF * 16;
  constexpr int Q4 = NF * 4;
  constexpr int NACC = (64 % Q4 == 0) ? 1 : 3;
  constexpr int NI = NF;                     // 16 rows x NF*4 quads = 64 * NF items per pass
  static_assert(MR <= 128 && NF <= 4 && (64 % Q4 == 0 || Q4 == 12), "c3l_epilogue: unexpected tile shape");
  float* stg = reinterpret_cast<float*>(smem) + wave * (16 * LD);
  int* rowoff = reinterpret_cast<int*>(smem) + 4 * 16 * LD + wave * 128;
  double2* exch = reinterpret_cast<double2*>(smem + C3_EPI_EXCH_OFF);

  // byte offset of output row r of this wave (C3_OOB: pad position)
  // output map (conv_gather_x6.hip; omap = 0: the H x W grid itself): grid pixel (y, x) -> pixel (y * ost + oy0, x * ost + ox0)
  // of an oH x oW tensor - the parity classes of a stride-2 data gradient
  const int oH = p.omap ? p.oH : p.H, oW = p.omap ? p.oW : p.W, ost = p.omap ? p.ost : 1;
  const int oy0 = p.omap ? p.oy0 : 0, ox0 = p.omap ? p.ox0 : 0;
  const unsigned obytes = (unsigned)p.N * oH * oW * p.Co * 4u;
  const __amdgpu_buffer_rsrc_t r_out = c3_rsrc(p.out, obytes);
  int cnt = 0;
  int r0 = -1;          // STATS: the wave's first REAL row (wave-uniform) - the pivot row
#pragma unroll
  for (int h = 0; h < RH; ++h) {
    unsigned myoff = C3_OOB;
    const int r = lane + 64 * h;
    const int pp = p0 + wave_m * MR + r;
    const int n = fast_div(pp, p.ib_mul, p.ib_sh);
    const int rem = pp - n * p.IB;
    const int yy = fast_div(rem, p.sw_mul, p.sw_sh), xx = rem - yy * p.SW;
    if ((r < MR) & (pp < p.P) & (n < p.N) & (yy >= 1) & (xx >= 1) & (xx <= p.W))
      myoff = (unsigned)(((n * oH + (yy - 1) * ost + oy0) * oW + (xx - 1) * ost + ox0) * p.Co) * 4u;
    if (STATS) {
      const unsigned long long vm = __ballot(myoff != C3_OOB);
      cnt += __popcll(vm);
      if (r0 < 0 && vm) r0 = 64 * h + __builtin_ctzll(vm);
    }
    if (r < MR) reinterpret_cast<unsigned*>(rowoff)[r] = myoff;
  }
  const int ncol0 = n0 + wave_n * NF * 16;
  float* piv = tab + wave * (NF * 16);                 // STATS: this wave's pivots
  const float* btab = tab + wave_n * NF * 16;          // BS: column tables of this wave, kinds BN floats apart
  if constexpr (STATS) {
    // pivot = the accumulator row of the wave's first real position (a pad position holds a partial sum - zero in a gathered
    // convolution - that can be far from the channel mean): accumulator (mf0, nf, rg0) of the lanes of group g0
    r0 = r0 < 0 ? 0 : r0;
    const int mf0 = r0 >> 4, g0 = (r0 >> 2) & 3, rg0 = r0 & 3;
    float pv[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) pv[nf] = 0.f;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
      if (mf == mf0) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          pv[nf] = rg0 == 0 ? acc[mf][nf][0] : rg0 == 1 ? acc[mf][nf][1] : rg0 == 2 ? acc[mf][nf][2] : acc[mf][nf][3];
      }
    if (g == g0) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) piv[nf * 16 + i16] = pv[nf];
    }
  }
  __builtin_amdgcn_wave_barrier();             // rowoff, piv: written and read by this wave only (the LDS keeps a wave's order)

  // a lane's items: item k = (row_k, quad c4_k) of every 16-row pass; everything that does not depend on the pass is formed
  // once: the staged piece, the row-offset slot, the output / side-tensor addresses up to the row offset
  const float* sp[NI];
  const unsigned* rp[NI];
  int c4_k[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int item = lane + 64 * k;
    const int row = item / Q4;
    c4_k[k] = item - row * Q4;
    sp[k] = stg + row * LD + c4_k[k] * 4;
    rp[k] = reinterpret_cast<const unsigned*>(rowoff) + row;
  }
  const unsigned colb = (unsigned)ncol0 * 4u;          // byte offset of this wave's first column
  f32x4 s1[NACC], s2[NACC], pq[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    s1[j] = s2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (STATS) pq[j] = *reinterpret_cast<const f32x4*>(piv + c4_k[j] * 4);
  }
  // tensors read beside the tile travel one pass ahead (c3_epilogue: a memory round trip per pass otherwise)
  [[maybe_unused]] __amdgpu_buffer_rsrc_t r_res, r_z, r_y;
  if constexpr (RES) r_res = c3_rsrc(p.res, obytes);
  if constexpr (BS) r_z = c3_rsrc(p.bs_z, obytes);
  if constexpr (BSY) r_y = c3_rsrc(p.bs_y, obytes);
  f32x4 pre_r[2][RES ? NI : 1], pre_z[2][BS ? NI : 1], pre_y[2][BSY ? NI : 1];
  auto issue = [&](int ps, int buf) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const unsigned o = rp[k][ps * 16] + colb + c4_k[k] * 16;      // pad rows: out of range, the load returns zeros
      if constexpr (RES) pre_r[buf][k] = c3_bload(r_res, o);
      if constexpr (BS) pre_z[buf][k] = c3_bload(r_z, o);
      if constexpr (BSY) pre_y[buf][k] = c3_bload(r_y, o);
    }
  };
  if constexpr (RES || BS) issue(0, 0);
#pragma unroll
  for (int ps = 0; ps < MF; ++ps) {
    if (ps) __builtin_amdgcn_wave_barrier();
    if constexpr (RES || BS) {
      if (ps + 1 < MF) issue(ps + 1, (ps + 1) & 1);
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) stg[(g * 4 + rg) * LD + nf * 16 + i16] = acc[ps][nf][rg];
    __builtin_amdgcn_wave_barrier();
    // The items of a pass are ONE basic block (no branch around the store: a pad row's store is dropped by the bounds check,
    // its sums add zeros): next to a workgroup that is streaming MFMAs every DEPENDENT instruction of this wave waits tens
    // of cycles for its issue slot, so the pass must offer the scheduler independent work - the items side by side.
    f32x4 v[NI];
    unsigned off[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      v[k] = *reinterpret_cast<const f32x4*>(sp[k]);
      off[k] = rp[k][ps * 16];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const bool ok = off[k] != C3_OOB;
      if constexpr (RES) v[k] += pre_r[ps & 1][k];
      c3_bstore(r_out, off[k] + colb + c4_k[k] * 16, v[k]);
      if constexpr (STATS) {
        f32x4 d = v[k] - pq[k % NACC];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = ok ? d[j] : 0.f;
        s1[k % NACC] += d;
        f32x4 q = s2[k % NACC];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = __builtin_fmaf(d[j], d[j], q[j]);
        s2[k % NACC] = q;
      }
      if constexpr (BS) {
        // the arithmetic of bn_bwd_reduce2_kernel's body, element for element (as c3_epilogue)
        const f32x4 zz = pre_z[ps & 1][k];
        const f32x4 mu = *reinterpret_cast<const f32x4*>(btab + c4_k[k] * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(btab + BN + c4_k[k] * 4);
        f32x4 yy;
        if constexpr (BSR) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(btab + 2 * BN + c4_k[k] * 4);
          yy = (zz - mu) * sc + *reinterpret_cast<const f32x4*>(btab + 3 * BN + c4_k[k] * 4);
        } else {
          yy = pre_y[ps & 1][k];
        }
        f32x4 gm;
#pragma unroll
        for (int j = 0; j < 4; ++j) gm[j] = (ok && yy[j] > 0.f) ? v[k][j] : 0.f;
        s1[k % NACC] += gm;
        f32x4 tm = gm * (zz - mu) * is;       // pad rows: gm = 0 and z = 0 (bounds check): the term is a zero
        s2[k % NACC] += tm;
      }
    }
    // keep a pass's arithmetic with the pass: left alone the optimiser sinks the sums of all passes behind the last store
    // and carries every staged piece there in registers (hundreds spilled in the backward modes)
    if constexpr (STATS || BS) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) asm volatile("" : "+v"(s1[j]), "+v"(s2[j]));
    }
  }
  if constexpr (STATS || BS) {
    // lanes -> column quads in a fixed order: every lane parks its NACC pairs in LDS (the staging area is dead after one more
    // barrier), lane c < Q4 adds the entries whose quad is c, lanes in ascending order
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem) + (size_t)wave * (NACC * 64 * 2);
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      red[(j * 64 + lane) * 2 + 0] = s1[j];
      red[(j * 64 + lane) * 2 + 1] = s2[j];
    }
    __builtin_amdgcn_wave_barrier();          // a wave reads back only what it wrote
    if (lane < Q4) {
      if constexpr (BS) {
        f32x4 a1 = (f32x4){0.f, 0.f, 0.f, 0.f}, a2 = a1;
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          const int first = ((lane - (64 * j) % Q4) % Q4 + Q4) % Q4;
          for (int l = first; l < 64; l += Q4) {
            a1 += red[(j * 64 + l) * 2 + 0];
            a2 += red[(j * 64 + l) * 2 + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          exch[wave_m * BN + wave_n * NF * 16 + lane * 4 + j] = make_double2((double)a1[j], (double)a2[j]);
      } else {
        double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          const int first = ((lane - (64 * j) % Q4) % Q4 + Q4) % Q4;
          for (int l = first; l < 64; l += Q4) {
            const f32x4 u1 = red[(j * 64 + l) * 2 + 0], u2 = red[(j * 64 + l) * 2 + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a1[e] += (double)u1[e];
              a2[e] += (double)u2[e];
            }
          }
        }
        const double nn = (double)cnt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double pi = (double)piv[lane * 4 + e];
          exch[wave_m * BN + wave_n * NF * 16 + lane * 4 + e] =
              make_double2(a1[e] + nn * pi, a2[e] + 2.0 * pi * a1[e] + nn * pi * pi);
        }
      }
    }
    // the workgroup's waves in a fixed order, then ONE exact integer addition per channel and sum
    __syncthreads();
    if (t < BN) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const double2 v = exch[w * BN + t];
        a1 += v.x;
        a2 += v.y;
      }
      bnacc_add(STATS ? p.stats_acc : p.bs_acc, p.Co, bnacc_shard(), n0 + t, a1, a2);
    }
  }
}

// ---- input staging ----------------------------------------------------------------------------------------------------------
// A thread's share of one 16-channel chunk of a tile's input: row prow + 64 q (q < PA) of the BM + 2 SW + 2 staged rows, four
// channels.  make(): the rows' byte offsets in x (C3_OOB for a pad position: the load returns zeros) and their LDS slots (rows
// beyond the tile - last pass only - all go to the spare row arows - 1, which no fragment reads); load(): fp32 -> registers;
// store(): (input BatchNorm,) split into three bf16 pieces, pieces -> LDS.
// Passes [0, P0) exist for every row width, the rest come in pairs under a wave-uniform test.  A group is ONE basic block: the
// split is a chain of dependent conversions and subtractions per value, and a wave whose SIMD partner streams MFMAs waits tens
// of cycles per dependent issue (cycle stamps: the same ten passes take 2 k cycles beside an idle partner, 9-10 k beside a
// multiplying one) - the passes must stand side by side for the scheduler to interleave them.
template <int BM, bool IN_BN>
struct C3Stager {
  static constexpr int ROWB = Geo<3>::ROWB, PST = Geo<3>::PST, CPR = Geo<3>::CPR;
  static constexpr int RPP = 256 / CPR;                                 // 64 rows staged per pass
  static constexpr int PA = (BM + 2 * MAX_SW + 2 + RPP - 1) / RPP;
  static constexpr int P0 = (BM + 2 * 3 + 2) / RPP < PA ? (BM + 2 * 3 + 2) / RPP : PA;
  unsigned goff[PA], lrow[PA];
  f32x4 areg[PA];
  int arows, c4, prow;
  __amdgpu_buffer_rsrc_t r_x;

  __device__ __forceinline__ void init(const C3Args& p) {
    const int t = threadIdx.x;
    arows = p.na * 32;
    c4 = (t % CPR) * 4;
    prow = t / CPR;
    r_x = c3_rsrc(p.x, (unsigned)p.N * p.H * p.W * p.Ci * 4u);
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int row = prow + RPP * q;
      lrow[q] = (unsigned)(row < arows ? row : arows - 1) * ROWB + 2 * c4;
    }
  }
  __device__ __forceinline__ void make(const C3Args& p, int p0) {
    const int halo = p.SW + 1;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int row = prow + RPP * q;
      const int pp = p0 - halo + row;
      const int n = fast_div(pp, p.ib_mul, p.ib_sh);
      const int rem = pp - n * p.IB;
      const int yy = fast_div(rem, p.sw_mul, p.sw_sh);
      const int xx = rem - yy * p.SW;
      // (bitwise: a short-circuit && would put every pass in its own basic blocks)
      const bool real = (row < arows) & (pp >= 0) & (pp < p.P) & (n < p.N) & (yy >= 1) & (xx >= 1) & (xx <= p.W);
      goff[q] = real ? (unsigned)(((n * p.H + yy - 1) * p.W + xx - 1) * p.Ci + c4) * 4u : C3_OOB;
    }
  }
  __device__ __forceinline__ void load(int c0) {
#pragma unroll
    for (int q = 0; q < P0; ++q) areg[q] = c3_bload(r_x, goff[q] + (unsigned)c0 * 4u);
#pragma unroll
    for (int q0 = P0; q0 < PA; q0 += 2)
      if (RPP * q0 < arows) {
        areg[q0] = c3_bload(r_x, goff[q0] + (unsigned)c0 * 4u);
        if (q0 + 1 < PA) areg[q0 + 1] = c3_bload(r_x, goff[q0 + 1] + (unsigned)c0 * 4u);
      }
  }
  __device__ __forceinline__ void store_pass(const C3Args& p, unsigned char* At, int q, const f32x4& mu, const f32x4& sc,
                                             const f32x4& be) {
    f32x4 v = areg[q];
    if constexpr (IN_BN) {
      // the exact expression of bn_apply_kernel; zero padding stays zero: it pads the NORMALISED tensor
      const bool real = goff[q] != C3_OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (v[j] - mu[j]) * sc[j] + be[j];
        const float w = p.in_relu ? fmaxf(u, 0.f) : u;
        v[j] = real ? w : 0.f;
      }
    }
    split_store_pk<3, PST>(At + lrow[q], 0, v);
  }
  // bntab: [3][Ci] floats (mean, invstd * gamma, beta) of the input BatchNorm
  __device__ __forceinline__ void store(const C3Args& p, unsigned char* At, int c0, const float* bntab) {
    f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f}, sc = mu, be = mu;
    if constexpr (IN_BN) {
      mu = *reinterpret_cast<const f32x4*>(bntab + c0 + c4);
      sc = *reinterpret_cast<const f32x4*>(bntab + p.Ci + c0 + c4);
      be = *reinterpret_cast<const f32x4*>(bntab + 2 * p.Ci + c0 + c4);
    }
#pragma unroll
    for (int q = 0; q < P0; ++q) store_pass(p, At, q, mu, sc, be);
#pragma unroll
    for (int q0 = P0; q0 < PA; q0 += 2)
      if (RPP * q0 < arows) {
        store_pass(p, At, q0, mu, sc, be);
        if (q0 + 1 < PA) store_pass(p, At, q0 + 1, mu, sc, be);       // (rows beyond the tile land in the spare row)
      }
  }
};

// ---- one output tile ---------------------------------------------------------------------------------------------------
// c3x6_tile (conv3x3.hip) with the option set as a template argument, the packed split, and the epilogue above.
// smem = [DBUF ? 2 : 1][arows][ROWB] A buffers | [3][Ci] input-BatchNorm table | [4][BN] epilogue table
template <int MF, int NF, int WM, int WN, bool DBUF, bool BPF_, int MODE>
__device__ __forceinline__ void c3l_tile(const C3Args& p, unsigned char* smem, int bx, int by) {
  static_assert(DBUF || !BPF_, "c3l_tile: the single-buffer form fetches its B fragments in place (the next chunk's rows ride behind them)");
  constexpr bool IN_BN = (MODE & C3M_IN_BN) != 0;
  constexpr bool BSR = (MODE & C3M_BS_REBUILD) != 0, BS = BSR || (MODE & C3M_BS_Y) != 0;
  constexpr int ROWB = Geo<3>::ROWB, PST = Geo<3>::PST, CPR = Geo<3>::CPR;
  constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
  constexpr int RPP = 256 / CPR;                                 // 64 rows staged per pass
  constexpr int PA = (BM + 2 * MAX_SW + 2 + RPP - 1) / RPP;
  const int arows = p.na * 32;
  const size_t abytes = DBUF ? (size_t)arows * ROWB : 0;         // one A buffer
  float* bntab = reinterpret_cast<float*>(smem + (size_t)(DBUF ? 2 : 1) * arows * ROWB);
  float* tab = bntab + 3 * p.Ci;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int p0 = bx * BM, n0 = by * BN;
  const int halo = p.SW + 1;
  const int c4 = (t % CPR) * 4, prow = t / CPR;

  C3Stager<BM, IN_BN> stg;
  stg.init(p);
  stg.make(p, p0);
  auto load_a = [&](int c0) { stg.load(c0); };
  // Single-buffer tiles (!DBUF) fetch the NEXT chunk's rows inside the current chunk, a quarter behind each of its first four
  // B loads: gfx9 counts vector-memory loads in ONE in-order counter, so a wait for a B fragment also waits for every row load
  // issued before it - issued in one batch in front of the chunk (as the double-buffered form can afford: its B fragments are
  // a step ahead) the first B wait of the chunk stood through a whole HBM round trip; now a B wait sees row loads that are at
  // least one step (126 MFMAs) old.  (Pad rows carry the out-of-range offset: no test per pass.)
  auto load_a_part = [&](int c0, int part) {
    constexpr int Q = (PA + 3) / 4;
#pragma unroll
    for (int q = part * Q; q < (part + 1) * Q; ++q)
      if (q < PA) stg.areg[q] = c3_bload(stg.r_x, stg.goff[q] + (unsigned)c0 * 4u);
  };
  auto store_a = [&](unsigned char* At, int c0) { stg.store(p, At, c0, bntab); };

  // B fragments of this lane: image [step][Co/16][3][64][16 B]
  const unsigned char* bptr = p.wp + ((size_t)(n0 / 16 + wave_n * NF) * 3) * 1024;   // scalar base
  const int blane = lane * 16;                                                        // the only per-lane part
  const size_t bstep = (size_t)(p.Co / 16) * 3 * 1024;
  constexpr bool BPF = BPF_;
  constexpr bool BPF2 = BPF && MF <= 2 && NF <= 3;
  bf16x8 bc[3][NF], bn[BPF ? 3 : 1][BPF ? NF : 1];
  bf16x8 bn2[BPF2 ? 3 : 1][BPF2 ? NF : 1];
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
        load_b(gs, bc);
        if (!DBUF && s < 4 && ch + 1 < nchunks) {
          __builtin_amdgcn_sched_barrier(0);
          load_a_part((ch + 1) * 16, s);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (DBUF && s == 2 && ch + 2 < nchunks) {
        __builtin_amdgcn_sched_barrier(0);
        load_a((ch + 2) * 16);
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
  };

  load_a(0);
  if constexpr (BPF) load_b(0, bn);
  if constexpr (BPF2) load_b(last_step > 0 ? 1 : 0, bn2);
  if constexpr (IN_BN) {
    // the producer's BatchNorm as (mean, invstd * gamma, beta) per input channel, decoded from the producer's accumulator
    // (bn_acc.h) under the first chunk's loads; tile (0, 0) also leaves mean / invstd for the backward kernels and updates
    // the running statistics
    for (int c = t; c < p.Ci; c += 256) {
      const BnFwdStat st = bnacc_fwd_stat(p.in_acc.acc, p.Ci, c, p.in_acc.rows, p.in_acc.eps);
      if (bx == 0 && by == 0) {
        p.in_acc.mean_out[c] = st.mean;
        p.in_acc.invstd_out[c] = st.invstd;
        if (p.in_acc.rmean) bnacc_running(st, p.in_acc.rows, p.in_acc.momentum, p.in_acc.rmean, p.in_acc.rvar, c);
      }
      bntab[c] = st.mean;
      bntab[p.Ci + c] = st.invstd * p.in_gamma[c];
      bntab[2 * p.Ci + c] = p.in_beta[c];
    }
  }
  if constexpr (BS) {
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
  if constexpr (IN_BN) __syncthreads();
  store_a(smem, 0);
  if (DBUF && nchunks > 1) load_a(16);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    if (!DBUF && ch > 0) {   // single buffer (the largest position tiles): restage between two barriers
      store_a(smem, ch * 16);
      __syncthreads();
    }
    run_chunk(ch);
    __syncthreads();       // DBUF: chunk ch+1 is complete in its buffer; nobody reads this chunk's buffer any more
  }
  c3l_epilogue<MF, NF, WM, WN, MODE>(p, acc, smem, tab, p0, n0);
}
