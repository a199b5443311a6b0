// fp32 MFMA tile engine shared by the conv / wgrad / batched-matmul kernels.
//
// CDNA4 mapping: one workgroup = 4 wave64 waves arranged WM x WN; each wave
// owns an (MF*16) x (NF*16) output tile held in MF*NF v_mfma_f32_16x16x4_f32
// accumulators (exact fp32: the result is a k-ordered fmaf chain, so parity
// with the fp32 reference does not depend on a reduced-precision format).
//
// LDS operand images, BK = 16 reduction elements per stage:
//   "row image"  (k contiguous in HBM):  T[row][k], row stride LDR = 24 floats.
//       A lane (i = lane&15, kq = lane>>4) fetches k = 4kq..4kq+3 with ONE
//       ds_read_b128; stride 24 makes the 4 non-contiguous 16-lane groups of
//       ds_read_b128 hit 16 distinct 16-byte slots (conflict free).
//       MFMA step kk then consumes physical k = 4*kq + kk for both operands -
//       a permutation of the reduction index, legal because A and B use the
//       same one.
//   "col image"  (k strided, m/n contiguous in HBM): T[k][col], stride cols+4.
//       Fetched with 4 ds_read_b32 (k = 4kq+kk); (cols+4)*4 = 16 mod 32 banks
//       puts the two kq values of a 32-lane half on disjoint bank sets.
#pragma once
#include "common.h"

#define GK 16          // reduction elements per LDS stage
#define LDR 24         // row-image stride in floats

template <int WM_, int WN_, int MF_, int NF_>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, MF = MF_, NF = NF_;
  static constexpr int BM = WM_ * MF_ * 16;
  static constexpr int BN = WN_ * NF_ * 16;
  static_assert(WM_ * WN_ == 4, "256-thread workgroups");
};

template <int ROWS, bool COLIMG>
struct OperandImage {
  static constexpr int LD = COLIMG ? (ROWS + 4) : LDR;
  static constexpr int SIZE = COLIMG ? GK * (ROWS + 4) : ROWS * LDR;
};

// One BK=16 stage of MFMAs for this wave's tile.
template <class T, bool ACOL, bool BCOL>
__device__ __forceinline__ void mma_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                          f32x4 (&acc)[T::MF][T::NF], int wm, int wn, int lane) {
  constexpr int LDA = OperandImage<T::BM, ACOL>::LD;
  constexpr int LDB = OperandImage<T::BN, BCOL>::LD;
  const int i = lane & 15, kq = lane >> 4;
  float a[T::MF][4], b[T::NF][4];
#pragma unroll
  for (int mf = 0; mf < T::MF; ++mf) {
    const int row = (wm * T::MF + mf) * 16 + i;
    if (!ACOL) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(As + row * LDA + kq * 4);
      a[mf][0] = v.x; a[mf][1] = v.y; a[mf][2] = v.z; a[mf][3] = v.w;
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[mf][kk] = As[(kq * 4 + kk) * LDA + row];
    }
  }
#pragma unroll
  for (int nf = 0; nf < T::NF; ++nf) {
    const int col = (wn * T::NF + nf) * 16 + i;
    if (!BCOL) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Bs + col * LDB + kq * 4);
      b[nf][0] = v.x; b[nf][1] = v.y; b[nf][2] = v.z; b[nf][3] = v.w;
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b[nf][kk] = Bs[(kq * 4 + kk) * LDB + col];
    }
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < T::NF; ++nf)
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mf][kk], b[nf][kk], acc[mf][nf], 0, 0, 0);
}

template <class T>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[T::MF][T::NF]) {
#pragma unroll
  for (int mf = 0; mf < T::MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < T::NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// Accumulator element (mf, nf, reg) of lane `lane` in wave (wm, wn) is output
// row  (wm*MF+mf)*16 + (lane>>4)*4 + reg   and column (wn*NF+nf)*16 + (lane&15)
// (16x16 C/D map: col = lane&15, row = (lane>>4)*4 + reg).
template <class T>
__device__ __forceinline__ int acc_row(int wm, int mf, int lane, int reg) {
  return (wm * T::MF + mf) * 16 + (lane >> 4) * 4 + reg;
}
template <class T>
__device__ __forceinline__ int acc_col(int wn, int nf, int lane) {
  return (wn * T::NF + nf) * 16 + (lane & 15);
}
