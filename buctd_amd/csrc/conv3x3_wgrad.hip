// Weight gradient of the 3x3 / stride 1 / pad 1 NHWC convolution on the bf16 matrix cores with split-fp32 operands,
// fp32 accumulate - the companion of conv3x3.hip for the BasicBlock convs of reference
// lib/models/pose_hrnet.py:28-57 (autograd of nn.Conv2d).  NP = 3 pieces per operand ("bf16x6", six MFMAs per
// product, fp32-class: see conv3x3.hip) or NP = 2 ("bf16x3", three MFMAs, ~2^-16):
//
//     dW[co][tap][ci] = sum_p dY[p][co] * X[p + shift(tap)][ci]          p over the zero-padded flattened positions
//
// GEMM view: M = co, N = (tap, ci), K = positions.  Both operands are staged position-major in LDS exactly like the
// forward kernel's input tile (rows = positions, NP bf16 pieces per row), so a filter tap is again a row shift; the
// K-contiguous MFMA fragments (8 consecutive positions of one channel per lane) come out of the position-major
// tiles through the gfx950 transpose read ds_read_b64_tr_b16 (4 rows x 16 channels -> lane c gets 4 positions of
// channel c; verified on hardware).  One workgroup owns a (co-chunk, ci-chunk) pair - CH = 48 or 32 channels each -
// for a range of positions and all 9 taps: 9*CH/16 n-fragments dealt round-robin to the 4 waves, CH/16 m-fragments
// each.  Position ranges are split over the grid; per-split slabs are summed by wg3_reduce.
// The X tile is a ring of RING rows indexed by (position - first staged position) mod RING: consecutive stages of a
// workgroup overlap in all but KB rows, so after the first stage only the KB new rows are fetched and split.
//   NP = 2: KB = 96 positions per stage;  NP = 3: KB = 64 (6 B per element instead of 4).  Ring size = the R rows of
//   one stage.  A lane group's 8 positions are {4g..4g+3} u {16+4g..16+4g+3} of the 32-position k-step (any assignment
//   is legal as long as both operands use the same one): the 32 lanes served in one LDS cycle then touch 8 consecutive
//   rows = 8 distinct 32-byte bank windows (row stride = 32 mod 64 bytes), where the natural {8g..8g+7} assignment made
//   rows r and r+8 collide (2-way conflicts on every transpose read: half of all LDS cycles, measured).
#include "c3_common.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define WG_MAX_SW 75
#define WG_PRO 96          // halo rows fetched per round of the first stage

template <int NP> struct WGeo;
template <> struct WGeo<2> { static constexpr int KB = 96; };
template <> struct WGeo<3> { static constexpr int KB = 64; };
// The X ring holds exactly the R = KB + 2*SW + 2 rows a stage needs (a runtime size: W = 72 -> 214 rows at KB = 64, which
// is what lets two workgroups of the six-byte-per-element mode share a CU's 160 KB).

struct WG3Args {
  const float* x;
  const float* dy;
  float* part;          // [nsplit][Co][9][Ci]
  int N, H, W, Ci, Co;
  int SW, IB, P;
  int split_q, split_rem;   // stages per split: q, the first split_rem splits q + 1
  // optional BatchNorm(+ReLU) of the producer applied to X while it is staged (X = the producer's raw conv output):
  // relu((x - mean) * (invstd * gamma) + beta), the expression of bn_apply_kernel
  const float* x_mean;
  const float* x_invstd;
  const float* x_gamma;
  const float* x_beta;
  int x_relu;
  unsigned ib_mul, ib_sh, sw_mul, sw_sh;
};

__device__ __forceinline__ int wg_fast_div(int n, unsigned mul, unsigned sh) {
  return (int)(__umulhi((unsigned)n, mul) >> sh);
}

// channels c..c+3 of one row -> NP bf16 pieces, piece q at byte q*LO + 2c (the residual subtractions are exact)
// Two channels at a time: v_cvt_pk_bf16_f32 delivers the packed pair that is stored, the residuals come from a shift / mask
// and one packed subtraction - the same pieces bit for bit with 22 instead of 28 VALU instructions per four channels.
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));
template <int NP, int LO>
__device__ __forceinline__ void wg_split_store(unsigned char* row, int c, f32x4 v) {
  wg_f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, wg_bf16x2));
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, wg_bf16x2));
    *reinterpret_cast<wg_u32x2*>(row + q * LO + 2 * c) = (wg_u32x2){a, b};
    if (q + 1 < NP) {
      lo -= (wg_f32x2){__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u)};
      hi -= (wg_f32x2){__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
    }
  }
}

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int row_bytes) {
  // two transpose reads: positions +0..3 and +4..7 of this lane's channel
  // (the v4i16 form + per-element bit_cast to __bf16 is miscompiled by ROCm 7.2's hipcc - every element became
  //  element 0 - so use the v4bf16 form and a shufflevector)
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * row_bytes));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* p, const unsigned char* q) {
  // as tr_frag with the two 4-row groups addressed separately (the ring may wrap between them)
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ int ring_slot(int r, int ring) {     // r in [0, 2*ring)
  return r - (r >= ring ? ring : 0);
}

// One workgroup's share of a weight gradient: chunk pair (bxc, byc) of ncx x ncy, position split z of nz.  The body of
// conv3x3_wgrad_split_kernel (one convolution per launch) and of conv3x3_wgrad_group_kernel (the branches of a
// HighResolutionModule in one launch).
// Round 6: the staging code is straight-line (cycle stamps of the forward kernel: a wave whose SIMD partner streams MFMAs
// waits tens of cycles per DEPENDENT instruction, so the passes of a stage must stand side by side in ONE basic block for the
// scheduler to interleave them) - tensors are read through buffer descriptors (a pad position is an out-of-range offset: the
// load returns zeros, no select and no branch per row), the pixel arithmetic is branch-free, the input BatchNorm comes from
// an LDS table instead of four global loads per staged piece; in the main loop the wave id is a scalar (tap and channel
// fragment of an n-fragment are SALU work) and the X fragment of the next n-fragment is read while the current one is
// multiplied.
#define WG_OOB 0x80000000u
template <int NP, int CF>   // channel fragments (of 16) per chunk: 3 -> 48 channels, 2 -> 32
__device__ __forceinline__ void wg3_tile(const WG3Args& p, unsigned char* smem, int bxc, int byc, int z, int ncx, int nz) {
  constexpr int KB = WGeo<NP>::KB;
  constexpr int CH = CF * 16;
  constexpr int LO = CH * 2;                 // byte stride between the pieces of a row
  constexpr int RS = (CH * 2 * NP) % 64 == 32 ? CH * 2 * NP : CH * 2 * NP + 32;   // row stride = 32 mod 64
  constexpr int C4 = CH / 4;                 // float4 per row
  constexpr int PD = (KB * C4 + 255) / 256;      // float4 per thread for one stage of KB rows
  constexpr int PP = (WG_PRO * C4 + 255) / 256;  // ... for one prologue round of WG_PRO rows
  constexpr int NFR = 9 * CF;                // n-fragments (tap, ci16)
  constexpr int NW = (NFR + 3) / 4;          // n-fragments per wave
  constexpr bool WHOLE = KB * C4 == PD * 256;       // a stage is a whole number of passes of the workgroup (NP = 3: always)

  const int R = KB + 2 * p.SW + 2;                   // rows of one stage = ring size
  unsigned char* Dt = smem;                          // dY tile [KB][RS]
  unsigned char* Xt = smem + (size_t)KB * RS;        // X  ring [R][RS]
  float* bntab = reinterpret_cast<float*>(Xt + (size_t)R * RS);     // [3][CH]: mean, invstd * gamma, beta of this ci chunk

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int t16 = lane & 15, g = lane >> 4;
  const int co0 = bxc * CH, ci0 = byc * CH;
  // splits get q or q+1 stages (the first `split_rem` ones one more): no split-count rounding loss
  const int k_begin = (z * p.split_q + (z < p.split_rem ? z : p.split_rem)) * KB;
  int k_end = k_begin + (p.split_q + (z < p.split_rem ? 1 : 0)) * KB;
  if (k_end > p.P) k_end = p.P;
  const int halo = p.SW + 1;
  const int ci_lim = p.Ci - ci0;             // input channels this chunk really has (< CH only in a ragged last chunk: zeros beyond)
  const bool x_bn = p.x_mean != nullptr;

  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)p.N * p.H * p.W * p.Ci * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (unsigned)p.N * p.H * p.W * p.Co * 4u, 0x00020000);
  typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
  auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  };
  // byte offset of pixel pp (channel 0) in an [N][H][W][C] tensor, WG_OOB for a pad position (or `ok` false)
  auto pos_offset = [&](int pp, int C, bool ok) -> unsigned {
    const int n = wg_fast_div(pp, p.ib_mul, p.ib_sh);
    const int rem = pp - n * p.IB;
    const int yy = wg_fast_div(rem, p.sw_mul, p.sw_sh);
    const int xx = rem - yy * p.SW;
    const bool real = ok & (pp >= 0) & (pp < p.P) & (n < p.N) & (yy >= 1) & (xx >= 1) & (xx <= p.W);
    return real ? (unsigned)(((n * p.H + yy - 1) * p.W + xx - 1) * C) * 4u : WG_OOB;
  };
  // a thread's pieces of a stage: piece q = (row srow[q], channels sc4[q] .. + 3) of the KB rows
  int srow[PD], sc4[PD];
#pragma unroll
  for (int q = 0; q < PD; ++q) {
    const int idx = t + 256 * q;
    srow[q] = idx / C4;
    sc4[q] = (idx - srow[q] * C4) * 4;
  }
  auto bn_in = [&](f32x4 v, int c, bool real) -> f32x4 {    // channels ci0 + c .. + 3; zero padding stays zero
    const f32x4 mu = *reinterpret_cast<const f32x4*>(bntab + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bntab + CH + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(bntab + 2 * CH + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = (v[j] - mu[j]) * sc[j] + be[j];
      const float w = p.x_relu ? fmaxf(u, 0.f) : u;
      v[j] = real ? w : 0.f;
    }
    return v;
  };

  f32x4 dreg[PD], xreg[PD];
  unsigned xoff[PD];                          // (the BatchNorm form needs to know which rows are padding)
  auto load_d = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      const int pp = k0 + srow[q];
      dreg[q] = bload(r_d, pos_offset(pp, p.Co, (WHOLE || srow[q] < KB) & (pp < k_end)) + (unsigned)(co0 + sc4[q]) * 4u);
    }
  };
  // X rows rel0 .. rel0 + KB - 1, counted from position k_begin - halo
  auto load_x = [&](int rel0) {
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      xoff[q] = pos_offset(k_begin - halo + rel0 + srow[q], p.Ci, (WHOLE || srow[q] < KB) & (sc4[q] < ci_lim));
      xreg[q] = bload(r_x, xoff[q] + (unsigned)(ci0 + sc4[q]) * 4u);
    }
  };
  auto store_d = [&]() {
#pragma unroll
    for (int q = 0; q < PD; ++q)
      if (WHOLE || srow[q] < KB) wg_split_store<NP, LO>(Dt + (size_t)srow[q] * RS, sc4[q], dreg[q]);
  };
  auto store_x = [&](int slot0) {            // slot0: ring slot of the first of the KB rows (already reduced mod RING)
#pragma unroll
    for (int q = 0; q < PD; ++q)
      if (WHOLE || srow[q] < KB)
        wg_split_store<NP, LO>(Xt + (size_t)ring_slot(slot0 + srow[q], R) * RS, sc4[q],
                               x_bn ? bn_in(xreg[q], sc4[q], xoff[q] != WG_OOB) : xreg[q]);
  };

  f32x4 acc[CF][NW];
#pragma unroll
  for (int mf = 0; mf < CF; ++mf)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[mf][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transpose-read lane addressing: lane t16 of group g points at row g*4 + (t16>>2) (second read: + 16), channels
  // 4*(t16&3)..+3
  const int lane_row = g * 4 + (t16 >> 2), lane_col = (t16 & 3) * 8;
  const int lane_off = lane_row * RS + lane_col;

  if (x_bn) {
    for (int c = t; c < CH; c += 256) {
      const bool in = c < ci_lim;
      bntab[c] = in ? p.x_mean[ci0 + c] : 0.f;
      bntab[CH + c] = in ? p.x_invstd[ci0 + c] * p.x_gamma[ci0 + c] : 0.f;
      bntab[2 * CH + c] = in ? p.x_beta[ci0 + c] : 0.f;
    }
    __syncthreads();
  }
  // first stage: the first R - KB rows (the halo) go in synchronously, WG_PRO rows per round with all loads of a
  // round in flight together; the last KB rows of the first stage travel through the steady-state registers
  if (k_begin < k_end) {
    const int pro = R - KB;                  // = 2*SW + 2 < R: slots 0 .. pro-1, no wrap
    for (int r0 = 0; r0 < pro; r0 += WG_PRO) {
      f32x4 preg[PP];
      unsigned poff[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        const int idx = t + 256 * q;
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        poff[q] = pos_offset(k_begin - halo + r0 + row, p.Ci, (row < WG_PRO) & (r0 + row < pro) & (c4 < ci_lim));
        preg[q] = bload(r_x, poff[q] + (unsigned)(ci0 + c4) * 4u);
      }
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        const int idx = t + 256 * q;
        const int row = idx / C4, c4 = (idx - row * C4) * 4;
        if (row < WG_PRO && r0 + row < pro)
          wg_split_store<NP, LO>(Xt + (size_t)(r0 + row) * RS, c4, x_bn ? bn_in(preg[q], c4, poff[q] != WG_OOB) : preg[q]);
      }
    }
    load_d(k_begin);
    load_x(pro);
  }
  int slot_new = R - KB;                     // ring slot receiving the first of the KB rows held in xreg
  int slot_base = 0;                         // ring slot of position (k0 - halo), tap shift 0
  int rel_next = R;                          // first row (from k_begin - halo) of the stage after this one
  const bool has_last = wave + 4 * (NW - 1) < NFR;     // (27 n-fragments on four waves: 7 + 7 + 7 + 6)
  for (int k0 = k_begin; k0 < k_end; k0 += KB) {
    __syncthreads();                       // previous stage fully consumed
    store_d();
    store_x(slot_new);
    if (k0 + KB < k_end) {                 // in flight during the MFMAs below
      load_d(k0 + KB);
      load_x(rel_next);
    }
    slot_new = ring_slot(slot_new + KB, R);
    rel_next += KB;
    __syncthreads();
    // X fragment of n-fragment j in k-step ks: tap, channel fragment and the ring slot of the window's first row are scalars (the
    // wave id is); the lane's part is ONE add and the wrap of its row, taken in byte space (a byte offset is beyond the ring
    // exactly when its row is: the offset inside a row stays below RS) as an unsigned minimum - o - ringb wraps to a huge
    // number unless o >= ringb.  Six VALU instructions per n-fragment where slot arithmetic per row + two 24-bit multiplies
    // took twelve to sixteen (what-if without any ring arithmetic: -8 us of the 107 of a {0,1} launch);
    // read one n-fragment ahead of its MFMAs, across the k-step boundary
    const unsigned ringb = (unsigned)R * RS;
    bf16x8 b[2][NP];
    auto read_b = [&](int ks, int j, bf16x8 (&dst)[NP]) {
      const int nf = wave + 4 * j;
      const int tap = nf / CF, cf = nf - tap * CF;
      const int shift = (tap / 3) * p.SW + tap % 3 + ks * 32;      // X row of position k + shift(tap) (the ring starts at -halo)
      const int s = ring_slot(slot_base + shift, R);
      unsigned o0 = (unsigned)lane_off + (unsigned)(s * RS + cf * 32);
      o0 = o0 - ringb < o0 ? o0 - ringb : o0;
      unsigned o1 = o0 + 16 * RS;
      o1 = o1 - ringb < o1 ? o1 - ringb : o1;
      const unsigned char* q0 = Xt + o0;
      const unsigned char* q1 = Xt + o1;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) dst[pc] = tr_frag2(q0 + pc * LO, q1 + pc * LO);
    };
    read_b(0, 0, b[0]);
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks) {
      bf16x8 a[NP][CF];
#pragma unroll
      for (int mf = 0; mf < CF; ++mf) {
        const unsigned char* q = Dt + (size_t)ks * 32 * RS + lane_off + mf * 32;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) a[pc][mf] = tr_frag(q + pc * LO, 4 * RS);   // second read: rows + 16
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int i = ks * NW + j;                         // flat index over the stage's n-fragments of this wave
        const bool more = i + 1 < (KB / 32) * NW;
        if (more) {
          const int jn = j + 1 < NW ? j + 1 : 0, ksn = j + 1 < NW ? ks : ks + 1;
          if (jn < NW - 1 || has_last) read_b(ksn, jn, b[(i + 1) & 1]);
        }
        if (j < NW - 1 || has_last) {
          bf16x8 (&bb)[NP] = b[i & 1];
#define WG_MMA(qa, qb)                                                                                      \
  _Pragma("unroll") for (int mf = 0; mf < CF; ++mf) acc[mf][j] =                                            \
      __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[qa][mf], bb[qb], acc[mf][j], 0, 0, 0);
          if constexpr (NP == 3) {
            WG_MMA(2, 0) WG_MMA(0, 2) WG_MMA(1, 1) WG_MMA(1, 0) WG_MMA(0, 1) WG_MMA(0, 0)
          } else {
            WG_MMA(1, 0) WG_MMA(0, 1) WG_MMA(0, 0)
          }
#undef WG_MMA
        }
      }
    }
    slot_base = ring_slot(slot_base + KB, R);
  }

  // partial slab in ACCUMULATOR order: element ((wave * NW + j) * CF + mf) * 64 + lane = the register quad (co = co0 + mf*16 +
  // g*4 + 0..3, tap / ci of fragment nf = wave + 4 j) - one 16-byte store per lane and quad, whole lines per wavefront,
  // instead of 84 scalar stores per lane into [co][tap][ci] rows; the reduction kernel sorts while it adds
  {
    const size_t slab4 = (size_t)4 * NW * CF * 64;
    const size_t pair = (size_t)byc * ncx + bxc;
    f32x4* outp = reinterpret_cast<f32x4*>(p.part) + (pair * nz + z) * slab4;
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int mf = 0; mf < CF; ++mf) {
        outp[((wave * NW + j) * CF + mf) * 64 + lane] = acc[mf][j];
      }
  }
}

template <int NP, int CF>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_split_kernel(WG3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wg3_tile<NP, CF>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.z);
}

// Several weight gradients in ONE launch: workgroups [first[k], first[k + 1]) belong to convolution k.  Inside a convolution
// the hardware's round-robin of workgroup ids over the 8 XCDs is undone so that every XCD walks a contiguous run of position
// splits (neighbouring splits share their 2 * SW + 2 halo rows: they meet in that XCD's L2), chunk pairs fastest.
#define WG3G_MAX 4
struct WG3Group {
  WG3Args conv[WG3G_MAX];
  int n;
  unsigned first[WG3G_MAX + 1];
  int ncx[WG3G_MAX], ncy[WG3G_MAX], nsplit[WG3G_MAX];
};
template <int NP, int CF>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_group_kernel(WG3Group g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WG3Group& g = *(const WG3Group*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < g.n && blockIdx.x >= g.first[k + 1]) ++k;
  const unsigned lid = blockIdx.x - g.first[k], total = g.first[k + 1] - g.first[k];
  const unsigned xcd = lid & 7, idx = lid >> 3, per = total >> 3, rem = total & 7;
  const unsigned L = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
  const unsigned pairs = (unsigned)(g.ncx[k] * g.ncy[k]);
  const unsigned z = L / pairs, pr = L - z * pairs;
  const unsigned byc = pr / (unsigned)g.ncx[k], bxc = pr - byc * (unsigned)g.ncx[k];
  wg3_tile<NP, CF>(g.conv[k], smem, (int)bxc, (int)byc, (int)z, g.ncx[k], g.nsplit[k]);
}

// slab reduction: thread = (element column, split lane); an element is one accumulator register quad (wave, j, mf, lane) =
// rows co0 + mf*16 + g*4 + 0..3 of column (tap, ci0 + cf*16 + t16), nf = wave + 4 j = tap*CF + cf.  16 elements x 16 split
// lanes per workgroup, four slab loads in flight per lane (the 512 slabs of the six-MFMA mode are 43 MB: at 8 split-lanes
// this pass took 27 us), fixed summation order: deterministic.
struct WG3Red { const float* part; float* out; int Ci, Co, nsplit, accumulate; };
template <int CF>
__device__ __forceinline__ void wg3_reduce_body(const WG3Red& a, f32x4 (&sm)[16][16], int bx, int nbx, int pair) {
  constexpr int NFR = 9 * CF;
  constexpr int NW = (NFR + 3) / 4;
  constexpr int SLAB = 4 * NW * CF * 64;              // float4 per slab
  const float* __restrict__ part = a.part;
  float* __restrict__ out = a.out;
  const int Ci = a.Ci, Co = a.Co, nsplit = a.nsplit, accumulate = a.accumulate;
  const int col = threadIdx.x & 15, zl = threadIdx.x >> 4;
  // pair = ci_chunk * (Co / CH) + co_chunk
  const int nco = Co / (CF * 16);
  const int co0 = (pair % nco) * CF * 16, ci0 = (pair / nco) * CF * 16;
  const f32x4* base = reinterpret_cast<const f32x4*>(part) + (size_t)pair * nsplit * SLAB;
  for (int e0 = bx * 16; e0 < SLAB; e0 += nbx * 16) {
    const int e = e0 + col;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (e < SLAB) {
      const f32x4* src = base + e;
      int zz = zl;
      for (; zz + 48 < nsplit; zz += 64) {
        s0 += src[(size_t)zz * SLAB];
        s1 += src[(size_t)(zz + 16) * SLAB];
        s2 += src[(size_t)(zz + 32) * SLAB];
        s3 += src[(size_t)(zz + 48) * SLAB];
      }
      for (; zz < nsplit; zz += 16) s0 += src[(size_t)zz * SLAB];
    }
    sm[zl][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zl == 0 && e < SLAB) {
      f32x4 sv = sm[0][col];
#pragma unroll
      for (int k = 1; k < 16; ++k) sv += sm[k][col];
      const int lane = e & 63, r = e >> 6;
      const int mf = r % CF, wj = r / CF, j = wj % NW, wv = wj / NW;
      const int nf = wv + 4 * j;
      if (nf < NFR) {
        const int tap = nf / CF, cf = nf - tap * CF;
        const int ci = ci0 + cf * 16 + (lane & 15);
        if (ci < Ci)                     // ragged last chunk: the columns beyond Ci were multiplied with zeros
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int co = co0 + mf * 16 + (lane >> 4) * 4 + rg;
          float* dst = out + ((size_t)co * 9 + tap) * Ci + ci;
          *dst = accumulate ? *dst + sv[rg] : sv[rg];
        }
      }
    }
    __syncthreads();
  }
}
template <int CF>
__global__ __launch_bounds__(256) void wg3_reduce_kernel(WG3Red a) {
  __shared__ f32x4 sm[16][16];
  wg3_reduce_body<CF>(a, sm, blockIdx.x, gridDim.x, blockIdx.y);
}
struct WG3RedGroup {
  WG3Red r[WG3G_MAX];
  int n;
  unsigned first[WG3G_MAX + 1];     // in units of blockIdx.y (chunk pairs)
};
template <int CF>
__global__ __launch_bounds__(256) void wg3_reduce_group_kernel(WG3RedGroup g_) {
  __shared__ f32x4 sm[16][16];
  const WG3RedGroup& g = *(const WG3RedGroup*)__builtin_amdgcn_kernarg_segment_ptr();
  int k = 0;
  while (k + 1 < g.n && blockIdx.y >= g.first[k + 1]) ++k;
  wg3_reduce_body<CF>(g.r[k], sm, blockIdx.x, gridDim.x, (int)(blockIdx.y - g.first[k]));
}

// ---------------------------------------------------------------------------------------------- host ----
struct WG3Plan { int CF, nsplit, q, rem; size_t lds; };

static void wg_magic(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  *mul = (unsigned)(((1ull << (31 + l)) + d - 1) / d);
  *sh = l - 1;
}

static bool wg3_plan(int np, int N, int H, int W, int Ci, int Co, WG3Plan* pl, int target = 0) {
  if ((np != 2 && np != 3) || c3_row_width(W) > WG_MAX_SW || H < 1 || W < 2) return false;
  // (x and dy are addressed with 32-bit byte offsets below the out-of-range mark 2^31: wg3_tile)
  if ((long)N * H * W * (Ci > Co ? Ci : Co) * 4 >= 2147483647L) return false;
  const int kb = np == 3 ? WGeo<3>::KB : WGeo<2>::KB;
  int cf;
  if (Ci % 48 == 0 && Co % 48 == 0) cf = 3;
  else if (Ci % 32 == 0 && Co % 32 == 0) cf = 2;
  // 48-column tiles over an input width that is only a multiple of 16 (the 256 -> 48 transition of HRNet-W48): the last
  // 48-channel chunk of the input is ragged, its missing channels are staged as zeros and dropped by the reduction
  else if (Co % 48 == 0 && Ci % 16 == 0 && Ci > 48) cf = 3;
  else return false;
  const int ch = cf * 16;
  const long P = (long)N * (H + 1) * c3_row_width(W) + c3_row_width(W);
  const long pairs = (long)(Co / ch) * ((Ci + ch - 1) / ch);
  // two resident workgroups per CU (512 in all) when one (co, ci) chunk pair exists; fewer splits per pair otherwise
  // (partial-slab traffic grows with the split).  Splits take q or q + 1 stages, so no rounding loss.
  // target > 0: the convolution shares its launch with others (wg3 group) and gets this many workgroups
  long want = ((target > 0 ? target : (np == 3 ? 512 : 384)) + pairs - 1) / pairs;
  const long stages = (P + kb - 1) / kb;
  if (want > stages) want = stages;
  if (want < 1) want = 1;
  pl->CF = cf;
  pl->nsplit = (int)want;
  pl->q = (int)(stages / want);
  pl->rem = (int)(stages % want);
  int rs = ch * 2 * np;
  if (rs % 64 != 32) rs += 32;
  pl->lds = (size_t)(2 * kb + 2 * c3_row_width(W) + 2) * rs + (size_t)3 * ch * sizeof(float);   // dY tile (KB rows) + X ring (KB + 2*SW + 2 rows) + input-BatchNorm table
  return pl->lds <= 160 * 1024;
}

static size_t wg3_slab_floats(int cf) {     // accumulator-order slab of one pair
  return (size_t)4 * ((9 * cf + 3) / 4) * cf * 64 * 4;
}
static size_t wg3_ws_bytes(const WG3Plan& pl, int Ci, int Co) {
  const size_t pairs = (size_t)(Co / (pl.CF * 16)) * ((Ci + pl.CF * 16 - 1) / (pl.CF * 16));
  return pairs * pl.nsplit * wg3_slab_floats(pl.CF) * sizeof(float);
}

template <int NP, int CF>
static int wg3_launch(const WG3Args& a, const WG3Plan& pl, hipStream_t st) {
  static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
  auto fn = conv3x3_wgrad_split_kernel<NP, CF>;
  const dim3 block(256);
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done, "conv3x3_wgrad (split bf16)"))
    return rc;
  dim3 grid(a.Co / (CF * 16), (a.Ci + CF * 16 - 1) / (CF * 16), pl.nsplit);
  hipLaunchKernelGGL(fn, grid, block, pl.lds, st, a);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad (split bf16)");
  return BUCTD_OK;
}

struct WG3InBn { const float* mean; const float* invstd; const float* gamma; const float* beta; int relu; };

// argument block + plan of one weight gradient (no launch); target: see wg3_plan
static int wg3_fill(int np, int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw, void* workspace,
                    size_t workspace_bytes, const WG3InBn* x_bn, int target, WG3Args& a, WG3Plan& pl) {
  BUCTD_CHECK_ARG(x && dy && dw, "buctd_conv3x3_wgrad (split bf16): null tensor pointer");
  BUCTD_CHECK_ARG(wg3_plan(np, N, H, W, Ci, Co, &pl, target),
                  "buctd_conv3x3_wgrad (split bf16): unsupported shape N%d H%d W%d Ci%d Co%d", N, H, W, Ci, Co);
  const size_t need = wg3_ws_bytes(pl, Ci, Co);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_conv3x3_wgrad (split bf16): workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.N = N; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
  a.SW = c3_row_width(W); a.IB = (H + 1) * a.SW;
  const long P = (long)N * a.IB + a.SW;
  BUCTD_CHECK_ARG(P < 2147483647L, "buctd_conv3x3_wgrad (split bf16): tensor too large");
  a.P = (int)P;
  a.split_q = pl.q; a.split_rem = pl.rem;
  a.x_mean = a.x_invstd = a.x_gamma = a.x_beta = nullptr;
  a.x_relu = 0;
  if (x_bn && x_bn->mean) {
    BUCTD_CHECK_ARG(Ci % (pl.CF * 16) == 0, "buctd_conv3x3_wgrad: the fused input BatchNorm needs whole %d-channel chunks (Ci = %d)",
                    pl.CF * 16, Ci);
    BUCTD_CHECK_ARG(x_bn->invstd && x_bn->gamma && x_bn->beta, "buctd_conv3x3_wgrad: fused input BatchNorm needs all four arrays");
    a.x_mean = x_bn->mean; a.x_invstd = x_bn->invstd; a.x_gamma = x_bn->gamma; a.x_beta = x_bn->beta;
    a.x_relu = x_bn->relu;
  }
  wg_magic((unsigned)a.IB, &a.ib_mul, &a.ib_sh);
  wg_magic((unsigned)a.SW, &a.sw_mul, &a.sw_sh);
  return BUCTD_OK;
}

static int wg3_run(int np, int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                   int accumulate, void* workspace, size_t workspace_bytes, void* stream, const WG3InBn* x_bn = nullptr) {
  WG3Plan pl;
  WG3Args a;
  int rc = wg3_fill(np, N, H, W, Ci, Co, x, dy, dw, workspace, workspace_bytes, x_bn, 0, a, pl);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (np == 3) rc = pl.CF == 3 ? wg3_launch<3, 3>(a, pl, st) : wg3_launch<3, 2>(a, pl, st);
  else rc = pl.CF == 3 ? wg3_launch<2, 3>(a, pl, st) : wg3_launch<2, 2>(a, pl, st);
  if (rc) return rc;
  const int pairs = (Co / (pl.CF * 16)) * ((Ci + pl.CF * 16 - 1) / (pl.CF * 16));
  const dim3 rgrid(ceil_div((long)(wg3_slab_floats(pl.CF) / 4), 16), pairs);
  const WG3Red ra{(const float*)workspace, dw, Ci, Co, pl.nsplit, accumulate};
  if (pl.CF == 3)
    hipLaunchKernelGGL(wg3_reduce_kernel<3>, rgrid, dim3(256), 0, st, ra);
  else
    hipLaunchKernelGGL(wg3_reduce_kernel<2>, rgrid, dim3(256), 0, st, ra);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad (split bf16, reduce)");
  return BUCTD_OK;
}

// ---- the weight gradients of several convolutions in one launch (+ one launch for their slab reductions) ----------------
// A convolution launched alone is cut into 512 position splits so that its ONE round of workgroups fills the chip; every
// split pays the 2 * SW + 2 halo rows of its first stage and writes (and the reduction re-reads) an 84 KB slab.  n
// convolutions in one grid fill the chip together with fewer splits each, i.e. less halo and slab traffic per convolution.
// 256 workgroups per member, whatever the number of members (round 6, after the kernel itself got faster the slab share
// decides: two members 107 us at 2 x 256 against 115 at 2 x 512 and 128 at 2 x 768; three members 165 at 3 x 256 against 191 at
// 3 x 341 and 217 at 3 x 170; four members 246 at 4 x 256; C4 +0.8 % in an interleaved A/B - scratch/time_group_wgrad.py)
#define WG3_GROUP_PER_MEMBER 256
static int wg3_group_target(int n) { (void)n; return WG3_GROUP_PER_MEMBER; }

extern "C" size_t buctd_conv3x3_wgrad_bf16x6_group_workspace(int n, int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  if (n < 1 || n > WG3G_MAX || !wg3_plan(3, N, H, W, Ci, Co, &pl, n > 1 ? wg3_group_target(n) : 0)) return 0;
  return wg3_ws_bytes(pl, Ci, Co);
}

extern "C" int buctd_conv3x3_wgrad_bf16x6_group(int n, const buctd_wg3_conv* convs, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && n <= WG3G_MAX && convs, "buctd_conv3x3_wgrad_bf16x6_group: 1..%d convolutions", WG3G_MAX);
  if (n == 1) {
    const buctd_wg3_conv& c = convs[0];
    WG3InBn b{c.x_mean, c.x_invstd, c.x_gamma, c.x_beta, c.x_relu};
    return wg3_run(3, c.N, c.H, c.W, c.Ci, c.Co, c.x, c.dy, c.dw, c.accumulate, c.workspace, c.workspace_bytes, stream, &b);
  }
  WG3Group g;
  WG3RedGroup r;
  WG3Plan pl[WG3G_MAX];
  g.n = r.n = n;
  g.first[0] = r.first[0] = 0;
  size_t lds = 0;
  const int target = wg3_group_target(n);
  for (int k = 0; k < n; ++k) {
    const buctd_wg3_conv& c = convs[k];
    WG3InBn b{c.x_mean, c.x_invstd, c.x_gamma, c.x_beta, c.x_relu};
    const int rc = wg3_fill(3, c.N, c.H, c.W, c.Ci, c.Co, c.x, c.dy, c.dw, c.workspace, c.workspace_bytes, &b, target, g.conv[k], pl[k]);
    if (rc) return rc;
    BUCTD_CHECK_ARG(pl[k].CF == pl[0].CF, "buctd_conv3x3_wgrad_bf16x6_group: the convolutions must share the chunk width (48 or 32 channels)");
    const int ch = pl[k].CF * 16;
    g.ncx[k] = c.Co / ch;
    g.ncy[k] = (c.Ci + ch - 1) / ch;
    g.nsplit[k] = pl[k].nsplit;
    g.first[k + 1] = g.first[k] + (unsigned)(g.ncx[k] * g.ncy[k] * pl[k].nsplit);
    r.r[k] = WG3Red{(const float*)c.workspace, c.dw, c.Ci, c.Co, pl[k].nsplit, c.accumulate};
    r.first[k + 1] = r.first[k] + (unsigned)(g.ncx[k] * g.ncy[k]);
    if (pl[k].lds > lds) lds = pl[k].lds;
  }
  for (int k = n; k < WG3G_MAX; ++k) { g.first[k + 1] = g.first[n]; r.first[k + 1] = r.first[n]; g.ncx[k] = g.ncy[k] = g.nsplit[k] = 0; }
  hipStream_t st = (hipStream_t)stream;
  static unsigned char attr_done[2][BUCTD_MAX_DEVICES] = {{0}};
  const int cf = pl[0].CF;
  void (*fn)(WG3Group) = cf == 3 ? conv3x3_wgrad_group_kernel<3, 3> : conv3x3_wgrad_group_kernel<3, 2>;
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[cf - 2], "buctd_conv3x3_wgrad_bf16x6_group"))
    return rc;
  hipLaunchKernelGGL(fn, dim3(g.first[n]), dim3(256), lds, st, g);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x6_group");
  const dim3 rgrid(ceil_div((long)(wg3_slab_floats(cf) / 4), 16), r.first[n]);
  if (cf == 3) hipLaunchKernelGGL(wg3_reduce_group_kernel<3>, rgrid, dim3(256), 0, st, r);
  else hipLaunchKernelGGL(wg3_reduce_group_kernel<2>, rgrid, dim3(256), 0, st, r);
  BUCTD_CHECK_LAUNCH("buctd_conv3x3_wgrad_bf16x6_group (reduce)");
  return BUCTD_OK;
}

/* The number of workgroups of the weight-gradient kernel buctd_conv3x3_wgrad_bf16x6_group(n, convs) launches (< 0: error; no
 * launch, no workspace needed) - for tools that look the launch up in a kernel trace by its grid (bench.py). */
extern "C" int buctd_conv3x3_wgrad_bf16x6_group_workgroups(int n, const buctd_wg3_conv* convs) {
  BUCTD_CHECK_ARG(n > 0 && n <= WG3G_MAX && convs, "buctd_conv3x3_wgrad_bf16x6_group_workgroups: 1..%d convolutions", WG3G_MAX);
  long total = 0;
  for (int k = 0; k < n; ++k) {
    const buctd_wg3_conv& c = convs[k];
    WG3Plan pl;
    BUCTD_CHECK_ARG(wg3_plan(3, c.N, c.H, c.W, c.Ci, c.Co, &pl, n > 1 ? wg3_group_target(n) : 0),
                    "buctd_conv3x3_wgrad_bf16x6_group_workgroups: unsupported shape");
    const int ch = pl.CF * 16;
    total += (long)(c.Co / ch) * ((c.Ci + ch - 1) / ch) * pl.nsplit;
  }
  return (int)total;
}

static size_t wg3_workspace(int np, int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  if (!wg3_plan(np, N, H, W, Ci, Co, &pl)) return 0;
  return wg3_ws_bytes(pl, Ci, Co);
}

extern "C" int buctd_conv3x3_wgrad_bf16x3_supported(int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  return wg3_plan(2, N, H, W, Ci, Co, &pl) ? 1 : 0;
}
extern "C" size_t buctd_conv3x3_wgrad_bf16x3_workspace(int N, int H, int W, int Ci, int Co) {
  return wg3_workspace(2, N, H, W, Ci, Co);
}
extern "C" int buctd_conv3x3_wgrad_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const float* dy,
                                          float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  return wg3_run(2, N, H, W, Ci, Co, x, dy, dw, accumulate, workspace, workspace_bytes, stream);
}
extern "C" int buctd_conv3x3_wgrad_bf16x6_supported(int N, int H, int W, int Ci, int Co) {
  WG3Plan pl;
  return wg3_plan(3, N, H, W, Ci, Co, &pl) ? 1 : 0;
}
extern "C" size_t buctd_conv3x3_wgrad_bf16x6_workspace(int N, int H, int W, int Ci, int Co) {
  return wg3_workspace(3, N, H, W, Ci, Co);
}
extern "C" int buctd_conv3x3_wgrad_bf16x6(int N, int H, int W, int Ci, int Co, const float* x, const float* dy,
                                          float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  return wg3_run(3, N, H, W, Ci, Co, x, dy, dw, accumulate, workspace, workspace_bytes, stream);
}
extern "C" int buctd_conv3x3_wgrad_bf16x6_bnin(int N, int H, int W, int Ci, int Co, const float* x, const float* dy,
                                               float* dw, int accumulate, const float* x_mean, const float* x_invstd,
                                               const float* x_gamma, const float* x_beta, int x_relu, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  WG3InBn b{x_mean, x_invstd, x_gamma, x_beta, x_relu};
  return wg3_run(3, N, H, W, Ci, Co, x, dy, dw, accumulate, workspace, workspace_bytes, stream, &b);
}
