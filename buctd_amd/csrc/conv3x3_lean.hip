// Train-mode launches of the bf16x6 3x3 convolution (reference lib/models/pose_hrnet.py:28-57: the BasicBlock convolutions,
// forward and data gradient): the group kernel of conv3x3.hip with the option set of the launch as a template argument
// (c3_lean.h).  One kernel per (channel family, option set); a workgroup finds (member, tile) from its id exactly as
// conv3x3_x6_group_kernel does, so tile plans, tile order and outputs are those of the general launches.
#include "c3_lean.h"

template <int FAM, int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_x6_lean_kernel(C3Group g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the descriptor is read where it lies - in the kernel-argument segment (see conv3x3_x6_group_kernel)
  const C3Group& g = *(const C3Group*)__builtin_amdgcn_kernarg_segment_ptr();
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
      case 0: c3l_tile<7, 3, 4, 1, false, false, MODE>(p, smem, bx, by); break;
      case 1: c3l_tile<4, 3, 2, 2, true, true, MODE>(p, smem, bx, by); break;
      case 2: c3l_tile<2, 2, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      case 3: c3l_tile<4, 3, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      default: c3l_tile<4, 2, 4, 1, true, true, MODE>(p, smem, bx, by); break;
    }
  } else {
    switch (v) {
      case 0: c3l_tile<4, 2, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      case 1: c3l_tile<1, 4, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      case 2: c3l_tile<2, 2, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      case 3: c3l_tile<1, 2, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      case 4: c3l_tile<2, 4, 4, 1, true, true, MODE>(p, smem, bx, by); break;
      default: c3l_tile<2, 4, 2, 2, true, true, MODE>(p, smem, bx, by); break;
    }
  }
}

// h: members in launch order with tiles / gx / gy / variant filled (conv3x3.hip: buctd_conv3x3_bf16x6_group);
// lds: the largest member's plan + its epilogue table.  Returns BUCTD_EINVAL for an option set that has no kernel here.
int c3_lean_launch(const C3Group& h, int fam, int mode, unsigned grid, size_t lds, hipStream_t st) {
  static unsigned char attr_done[2][5][BUCTD_MAX_DEVICES] = {{{0}}};
  void (*fn)(C3Group) = nullptr;
  int mi = -1;
#define C3L_PICK(i, m)                                                                               \
  if (mode == (m)) {                                                                                 \
    mi = i;                                                                                          \
    fn = fam ? conv3x3_x6_lean_kernel<1, (m)> : conv3x3_x6_lean_kernel<0, (m)>;                      \
  }
  C3L_PICK(0, C3M_STATS)
  C3L_PICK(1, C3M_STATS | C3M_IN_BN)
  C3L_PICK(2, C3M_BS_REBUILD)
  C3L_PICK(3, C3M_RES | C3M_BS_Y)
  C3L_PICK(4, C3M_RES)
#undef C3L_PICK
  if (!fn) {
    buctd_set_error("conv3x3 (bf16x6, train mode): no kernel for option set %d", mode);
    return BUCTD_EINVAL;
  }
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[fam ? 1 : 0][mi],
                                           "conv3x3 (bf16x6, train mode)"))
    return rc;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, h);
  BUCTD_CHECK_LAUNCH("conv3x3 (bf16x6, train mode)");
  return BUCTD_OK;
}
