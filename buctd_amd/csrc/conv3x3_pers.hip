// Persistent train-mode launches of the bf16x6 3x3 convolution (c3_pers.h): a grid of resident workgroups (two per CU), each
// walking its share of the tiles of every member of the launch (reference lib/models/pose_hrnet.py:28-57, 177-185).
// Tiles are dealt round-robin over the slots in ONE sequence over all members (costliest tiles first), so a slot's share is
// within one tile of every other's; a slot's tiles of a member are a grid-stride sequence, processed as one software
// pipeline (c3p_segment).  The assignment is a pure function of the launch geometry: results do not depend on timing, and
// the statistics travel through the order-independent integer accumulators (bn_acc.h) - run-to-run bit-identical.
#include "c3_pers.h"

template <int FAM, int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_x6_pers_kernel(C3Group g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const C3Group& g = *(const C3Group*)__builtin_amdgcn_kernarg_segment_ptr();
  const unsigned G = gridDim.x, s = blockIdx.x;
  unsigned off = 0;
  for (int k = 0; k < g.nconv; ++k) {
    const unsigned T = (unsigned)g.tiles[k];
    const unsigned j0 = (s + G - off % G) % G;        // first tile of member k whose place in the sequence is = s mod G
    off += T;
    if (j0 >= T) continue;
    const C3Args& p = g.conv[k];
    const int v = g.variant[k];
    if constexpr (FAM == 0) {
      switch (v) {
        case 1: c3p_segment<4, 3, 2, 2, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        case 2: c3p_segment<2, 2, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        default: c3p_segment<4, 3, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
      }
    } else {
      switch (v) {
        case 0: c3p_segment<4, 2, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        case 1: c3p_segment<1, 4, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        case 2: c3p_segment<2, 2, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        case 3: c3p_segment<1, 2, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        case 4: c3p_segment<2, 4, 4, 1, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
        default: c3p_segment<2, 4, 2, 2, MODE>(p, smem, j0, T, G, g.gx[k], g.gy[k]); break;
      }
    }
  }
}

// h: members in launch order (costliest tiles first), every member planned with double-buffered tiles (no variant 0 of
// family 0) and Ci >= 32; grid: resident slots, a multiple of 8
int c3_pers_launch(const C3Group& h, int fam, int mode, unsigned grid, size_t lds, hipStream_t st) {
  static unsigned char attr_done[2][5][BUCTD_MAX_DEVICES] = {{{0}}};
  void (*fn)(C3Group) = nullptr;
  int mi = -1;
#define C3P_PICK(i, m)                                                                               \
  if (mode == (m)) {                                                                                 \
    mi = i;                                                                                          \
    fn = fam ? conv3x3_x6_pers_kernel<1, (m)> : conv3x3_x6_pers_kernel<0, (m)>;                      \
  }
  C3P_PICK(0, C3M_STATS)
  C3P_PICK(1, C3M_STATS | C3M_IN_BN)
  C3P_PICK(2, C3M_BS_REBUILD)
  C3P_PICK(3, C3M_RES | C3M_BS_Y)
  C3P_PICK(4, C3M_RES)
#undef C3P_PICK
  if (!fn) {
    buctd_set_error("conv3x3 (bf16x6, persistent): no kernel for option set %d", mode);
    return BUCTD_EINVAL;
  }
  if (const int rc = buctd_raise_lds_limit(reinterpret_cast<const void*>(fn), 160 * 1024, attr_done[fam ? 1 : 0][mi],
                                           "conv3x3 (bf16x6, persistent)"))
    return rc;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, h);
  BUCTD_CHECK_LAUNCH("conv3x3 (bf16x6, persistent)");
  return BUCTD_OK;
}
