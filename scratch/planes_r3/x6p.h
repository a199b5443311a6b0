// "x6 planes": an activation tensor stored PRE-SPLIT for the bf16x6 matrix-core kernels (conv3x3.hip, conv3x3_wgrad4.hip).
//
// Logical tensor [N][H][W][C] fp32 (C % 16 == 0).  Storage: one ROW per position of the zero-padded flattened pixel space
// the 3x3 kernels work in,
//       p = n*IB + (y+1)*SW + (x+1),   SW = W+2, IB = (H+1)*SW,   0 <= p < P = N*IB + SW
// (one zero column left and right of every image row, one zero row between images), plus X6P_GB guard rows before row 0
// and X6P_GA guard rows after row P-1.  Pad and guard rows hold ZEROS - the producer writes them - so a consumer stages a
// tile with a plain copy: no per-row div/mod, no validity masks, no clamping, and a filter tap is a constant row shift.
// A row is C/16 chunks of 96 bytes, chunk = [16 x bf16 h | 16 x bf16 m | 16 x bf16 l] with x = h + m + l EXACTLY
// (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); conv3x3.hip explains why six piece products are fp32-class).
// That is byte for byte the LDS row image of the forward / data-gradient kernel (per 16-channel chunk) and, for a
// 48-channel block, of the weight-gradient kernel's position-major tiles (row stride 288 = 32 mod 64: conflict-free
// transpose reads) - so both kernels move planes into LDS without touching a VALU.
// 6 bytes per element instead of 4: the producing element-wise kernel pays 2 (or 6, when it also keeps the fp32 tensor)
// extra bytes per element on a path that sits at 10-20 % of HBM bandwidth; the two consumers of every conv operand
// (forward or data gradient + weight gradient) each lose ~2 VALU instructions per MFMA.
#pragma once
#include "common.h"

#define X6P_GB 128    // guard rows in front of row 0   (>= SW + 1 for SW <= 75)
#define X6P_GA 768    // guard rows behind row P - 1    (>= largest position tile + SW + 1)

static inline long x6p_positions(int N, int H, int W) { return (long)N * (H + 1) * (W + 2) + (W + 2); }
static inline size_t x6p_row_bytes(int C) { return (size_t)C * 6; }
static inline size_t x6p_total_bytes(int N, int H, int W, int C) {
  return (size_t)(X6P_GB + x6p_positions(N, H, W) + X6P_GA) * x6p_row_bytes(C);
}
// byte offset of row 0 inside the allocation
static inline size_t x6p_row0(int C) { return (size_t)X6P_GB * x6p_row_bytes(C); }

typedef unsigned short x6p_u16x8 __attribute__((ext_vector_type(8)));

// eight consecutive channels -> the three 16-byte pieces (h, m, l); identical arithmetic to split_store of conv3x3.hip
__device__ __forceinline__ void x6p_split8(const float (&v)[8], x6p_u16x8& h, x6p_u16x8& m, x6p_u16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r = v[j];
    const __bf16 a = (__bf16)r;
    h[j] = __builtin_bit_cast(unsigned short, a);
    r -= (float)a;
    const __bf16 b = (__bf16)r;
    m[j] = __builtin_bit_cast(unsigned short, b);
    r -= (float)b;
    const __bf16 c = (__bf16)r;
    l[j] = __builtin_bit_cast(unsigned short, c);
  }
}

// store the pieces of channels c8*8 .. c8*8+7 of one row (row_ptr = first byte of the row)
__device__ __forceinline__ void x6p_store8(unsigned char* row_ptr, int c8, const x6p_u16x8& h, const x6p_u16x8& m,
                                           const x6p_u16x8& l) {
  unsigned char* q = row_ptr + (c8 >> 1) * 96 + (c8 & 1) * 16;
  *reinterpret_cast<x6p_u16x8*>(q) = h;
  *reinterpret_cast<x6p_u16x8*>(q + 32) = m;
  *reinterpret_cast<x6p_u16x8*>(q + 64) = l;
}
