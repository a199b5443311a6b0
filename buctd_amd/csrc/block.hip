// Native launch sequences of one residual BasicBlock in train mode (reference lib/models/pose_hrnet.py:28-57, stride 1,
// no downsample, bf16x6 math):  y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x).
// Nothing is computed here: these entry points enqueue the kernels that the host mirror used to enqueue one ctypes call
// at a time (nine calls and a dozen small allocations per block and direction - ~95 us of Python per block forward,
// which made HRNet-W32 at 256x192 host-bound).  Same kernels, same order, same streams: results are bit-identical.
#include "common.h"
#include "../../include/buctd_hip.h"
#include <string.h>

#define BLK_TRY(call)      \
  do {                     \
    const int rc_ = (call); \
    if (rc_) return rc_;   \
  } while (0)

extern "C" int buctd_basic_block_fwd_train(const buctd_basic_block* b, void* stream) {
  BUCTD_CHECK_ARG(b && b->x && b->w1_fwd && b->w2_fwd && b->z1 && b->z2 && b->y && b->acc && b->stat,
                  "buctd_basic_block_fwd_train: null pointer");
  const int N = b->N, H = b->H, W = b->W, C = b->C;
  const long rows = (long)N * H * W;
  // the two statistics accumulators (bn_acc.h; zero on entry): conv1's output, conv2's output
  void* acc1 = b->acc;
  void* acc2 = (char*)b->acc + buctd_bn_acc_bytes(C);
  float *mean1 = b->stat, *invstd1 = b->stat + C, *mean2 = b->stat + 2 * C, *invstd2 = b->stat + 3 * C;
  BLK_TRY(buctd_conv3x3_bf16x6_acc(N, H, W, C, C, b->x, b->w1_fwd, nullptr, 0, b->z1, acc1, nullptr, nullptr, nullptr, 0, stream));
  // conv2 decodes bn1's statistics from acc1 in its prologue (its first tile leaves mean1 / invstd1 for the backward pass and
  // updates the running statistics) and applies bn1 + ReLU while it stages its input: no finalize launch, and
  // relu(bn1(z1)) never exists in memory
  const buctd_bn_acc_in st1 = {acc1, rows, b->eps1, b->momentum1, mean1, invstd1, b->running_mean1, b->running_var1};
  BLK_TRY(buctd_conv3x3_bf16x6_acc(N, H, W, C, C, b->z1, b->w2_fwd, nullptr, 0, b->z2, acc2, &st1, b->gamma1, b->beta1, 1, stream));
  const buctd_bn_acc_in st2 = {acc2, rows, b->eps2, b->momentum2, mean2, invstd2, b->running_mean2, b->running_var2};
  BLK_TRY(buctd_bn_apply_acc(b->z2, &st2, b->gamma2, b->beta2, b->x, 1, b->y, rows, C, stream));
  return BUCTD_OK;
}

// prev: the block in FRONT of b in a chain (its output is b's input), or NULL.  With prev the data gradient of conv1 - whose
// output is prev's upstream gradient - also forms the sums of prev's bn2 backward (into prev's accumulator).
// bn2_ready: the block behind did that for b, so b's bn2 backward needs no reduction pass.
static int block_bwd_impl(const buctd_basic_block* b, const buctd_basic_block_grads* g, const buctd_basic_block* prev,
                          const buctd_basic_block_grads* gprev, bool bn2_ready, void* stream, void* side_stream) {
  BUCTD_CHECK_ARG(b && g && b->x && b->w1_bwd && b->w2_bwd && b->z1 && b->z2 && b->y && b->stat && g->dy && g->dres &&
                      g->dy1 && g->dw1 && g->dw2 && g->bn_acc && g->wg_ws,
                  "buctd_basic_block_bwd: null pointer");
  BUCTD_CHECK_ARG(g->dz2 && g->dz1, "buctd_basic_block_bwd: dz2 / dz1 scratch missing");
  const int N = b->N, H = b->H, W = b->W, C = b->C;
  const long rows = (long)N * H * W;
  const float *mean1 = b->stat, *invstd1 = b->stat + C, *mean2 = b->stat + 2 * C, *invstd2 = b->stat + 3 * C;
  hipStream_t main_s = (hipStream_t)stream, side_s = side_stream ? (hipStream_t)side_stream : main_s;
  // weight gradients run on the side stream behind the kernel that produced their dY operand
  // one cached event per host thread AND device (an event belongs to the device it was created on)
  static thread_local hipEvent_t evs[16] = {nullptr};
  int devid = 0;
  if (side_s != main_s && (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16)) {
    buctd_set_error("buctd_basic_block_bwd: cannot identify the current device");
    return BUCTD_ELAUNCH;
  }
  hipEvent_t& ev = evs[devid];
  auto fork = [&]() -> int {
    if (side_s == main_s) return BUCTD_OK;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      buctd_set_error("buctd_basic_block_bwd: hipEventCreate failed");
      return BUCTD_ELAUNCH;
    }
    if (hipEventRecord(ev, main_s) != hipSuccess || hipStreamWaitEvent(side_s, ev, 0) != hipSuccess) {
      buctd_set_error("buctd_basic_block_bwd: stream fork failed");
      return BUCTD_ELAUNCH;
    }
    return BUCTD_OK;
  };
  // The sums of each BatchNorm backward (sum g, sum g zhat over the batch) are a by-product of the data gradient that
  // PRODUCES g (its epilogue has the tile in registers): bn1's of conv2's data gradient, bn2's - in a chain - of the data
  // gradient of the block behind.  They travel as integer accumulators (bn_acc.h): g->bn_acc = [bn1 | bn2], zero on entry.
  void* acc_bn1 = g->bn_acc;
  void* acc_bn2 = (char*)g->bn_acc + buctd_bn_acc_bytes(C);
  // conv2 / bn2 (+ skip): dres = masked upstream gradient
  BLK_TRY(buctd_bn_bwd_acc(g->dy, b->y, b->z2, mean2, invstd2, b->gamma2, nullptr, 1, rows, C, g->dz2, g->dres, g->dgamma2,
                           g->dbeta2, g->acc_bn2, acc_bn2, bn2_ready ? 1 : 0, stream));
  BLK_TRY(fork());
  BLK_TRY(buctd_conv3x3_wgrad_bf16x6_bnin(N, H, W, C, C, b->z1, g->dz2, g->dw2, g->acc_w2, mean1, invstd1, b->gamma1,
                                          b->beta1, 1, g->wg_ws, g->wg_ws_bytes, side_s));
  // conv2's data gradient dy1, and with it the sums of bn1's backward (ReLU mask rebuilt from z1)
  BLK_TRY(buctd_conv3x3_bf16x6_bnstat_acc(N, H, W, C, C, g->dz2, b->w2_bwd, nullptr, g->dy1, b->z1, nullptr, mean1, invstd1,
                                          b->gamma1, b->beta1, acc_bn1, stream));
  BLK_TRY(buctd_bn_bwd_acc(g->dy1, nullptr, b->z1, mean1, invstd1, b->gamma1, b->beta1, 1, rows, C, g->dz1, nullptr, g->dgamma1,
                           g->dbeta1, g->acc_bn1, acc_bn1, 1, stream));
  BLK_TRY(fork());
  BLK_TRY(buctd_conv3x3_wgrad_bf16x6(N, H, W, C, C, b->x, g->dz1, g->dw1, g->acc_w1, g->wg_ws, g->wg_ws_bytes, side_s));
  // conv1's data gradient; the skip gradient joins in its epilogue; in a chain its output is the upstream gradient of the
  // block in front, whose bn2 sums it forms on the way out
  if (g->dx) {
    if (prev)
      BLK_TRY(buctd_conv3x3_bf16x6_bnstat_acc(N, H, W, C, C, g->dz1, b->w1_bwd, g->dres, g->dx, prev->z2, prev->y, prev->stat + 2 * C,
                                              prev->stat + 3 * C, prev->gamma2, nullptr,
                                              (char*)gprev->bn_acc + buctd_bn_acc_bytes(C), stream));
    else
      BLK_TRY(buctd_conv3x3_bf16x6(N, H, W, C, C, g->dz1, b->w1_bwd, nullptr, nullptr, nullptr, g->dres, 0, g->dx, nullptr,
                                   nullptr, stream));
  }
  return BUCTD_OK;
}

extern "C" int buctd_basic_block_bwd(const buctd_basic_block* b, const buctd_basic_block_grads* g, void* stream,
                                     void* side_stream) {
  return block_bwd_impl(b, g, nullptr, nullptr, false, stream, side_stream);
}

// A residual CHAIN (the four BasicBlocks of an HRNet branch, pose_hrnet.py:165-185 _make_one_branch): the blocks' launch
// sequences behind ONE call per direction.  Block k's input is block k-1's output; in the backward block k's upstream
// gradient is block k+1's input gradient.  Same kernels, same order, same streams as n single calls (bit-identical); what
// it saves is host time - the HRNet-W32 step is bound by it, and the W48 step starves the GPU wherever the maps are small.
extern "C" int buctd_basic_chain_fwd_train(int n, const buctd_basic_block* blocks, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && blocks, "buctd_basic_chain_fwd_train: bad argument");
  for (int k = 0; k < n; ++k) BLK_TRY(buctd_basic_block_fwd_train(blocks + k, stream));
  return BUCTD_OK;
}
extern "C" int buctd_basic_chain_bwd(int n, const buctd_basic_block* blocks, const buctd_basic_block_grads* grads, void* stream,
                                     void* side_stream) {
  BUCTD_CHECK_ARG(n > 0 && blocks && grads, "buctd_basic_chain_bwd: bad argument");
  bool ready = false;      // block k's bn2 sums were formed by block k + 1's conv1 data gradient
  for (int k = n - 1; k >= 0; --k) {
    // block k - 1 can take its bn2 sums from this block's data gradient if that gradient IS its upstream gradient, the two
    // blocks have one shape and share the workspace the sums travel in
    const bool chain = k > 0 && grads[k].dx && grads[k].dx == grads[k - 1].dy && grads[k - 1].bn_acc &&
                       blocks[k - 1].y == blocks[k].x && blocks[k - 1].N == blocks[k].N && blocks[k - 1].H == blocks[k].H &&
                       blocks[k - 1].W == blocks[k].W && blocks[k - 1].C == blocks[k].C;
    BLK_TRY(block_bwd_impl(blocks + k, grads + k, chain ? blocks + k - 1 : nullptr, chain ? grads + k - 1 : nullptr, ready, stream,
                           side_stream));
    ready = chain;
  }
  return BUCTD_OK;
}

// ---- the branches of a HighResolutionModule (pose_hrnet.py:177-185, 247-249) ---------------------------------------------
// nb independent chains of n BasicBlocks each (blocks[b * n + k] = block k of branch b), advanced TOGETHER: the k-th
// convolutions of all branches are one launch (buctd_conv3x3_bf16x6_group: their tiles form one grid of several de-phased
// rounds instead of nb phase-locked single rounds on nb streams), and so are the BatchNorm applies, the BatchNorm backwards
// and the weight gradients.  3 launches per block step forward, 8 backward - whatever the number of branches.
// Forward and data gradients are bit-identical to the per-branch chains; the weight gradients use the group split
// (buctd_conv3x3_wgrad_bf16x6_group: fixed order, fp32-class).
#define BR_MAX 4

static void acc_in_of(const buctd_basic_block& b, int second, buctd_bn_acc_in* st) {
  const long rows = (long)b.N * b.H * b.W;
  const int C = b.C;
  st->acc = second ? (char*)b.acc + buctd_bn_acc_bytes(C) : b.acc;
  st->rows = rows;
  st->eps = second ? b.eps2 : b.eps1;
  st->momentum = second ? b.momentum2 : b.momentum1;
  st->mean_out = b.stat + (second ? 2 * C : 0);
  st->invstd_out = b.stat + (second ? 3 * C : C);
  st->running_mean = second ? b.running_mean2 : b.running_mean1;
  st->running_var = second ? b.running_var2 : b.running_var1;
}

extern "C" int buctd_basic_branches_fwd_train(int nb, int n, const buctd_basic_block* blocks, void* stream) {
  BUCTD_CHECK_ARG(nb > 0 && nb <= BR_MAX && n > 0 && blocks, "buctd_basic_branches_fwd_train: 1..%d branches", BR_MAX);
  for (int i = 0; i < nb * n; ++i)
    BUCTD_CHECK_ARG(blocks[i].x && blocks[i].w1_fwd && blocks[i].w2_fwd && blocks[i].z1 && blocks[i].z2 && blocks[i].y &&
                        blocks[i].acc && blocks[i].stat,
                    "buctd_basic_branches_fwd_train: null pointer in block %d", i);
  for (int k = 0; k < n; ++k) {
    buctd_c3_conv cv[BR_MAX];
    buctd_bn_acc_in st1[BR_MAX];
    buctd_bn_apply_item ap[BR_MAX];
    memset(cv, 0, sizeof(cv));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      buctd_c3_conv& c = cv[b];
      c.N = B.N; c.H = B.H; c.W = B.W; c.Ci = c.Co = B.C;
      c.x = B.x; c.wprep = B.w1_fwd; c.y = B.z1; c.stats_acc = B.acc;
    }
    BLK_TRY(buctd_conv3x3_bf16x6_group(nb, cv, stream));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      buctd_c3_conv& c = cv[b];
      acc_in_of(B, 0, &st1[b]);
      c.x = B.z1; c.wprep = B.w2_fwd; c.y = B.z2; c.stats_acc = (char*)B.acc + buctd_bn_acc_bytes(B.C);
      c.in_bn = &st1[b]; c.in_gamma = B.gamma1; c.in_beta = B.beta1; c.in_relu = 1;
    }
    BLK_TRY(buctd_conv3x3_bf16x6_group(nb, cv, stream));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      buctd_bn_apply_item& a = ap[b];
      a.z = B.z2;
      acc_in_of(B, 1, &a.st);
      a.gamma = B.gamma2; a.beta = B.beta2; a.residual = B.x; a.relu = 1; a.y = B.y;
      a.rows = (long)B.N * B.H * B.W; a.C = B.C;
    }
    BLK_TRY(buctd_bn_apply_acc_group(nb, ap, stream));
  }
  return BUCTD_OK;
}

extern "C" int buctd_basic_branches_bwd(int nb, int n, const buctd_basic_block* blocks, const buctd_basic_block_grads* grads,
                                        void* stream, void* side_stream) {
  BUCTD_CHECK_ARG(nb > 0 && nb <= BR_MAX && n > 0 && blocks && grads, "buctd_basic_branches_bwd: 1..%d branches", BR_MAX);
  for (int i = 0; i < nb * n; ++i) {
    const buctd_basic_block& b = blocks[i];
    const buctd_basic_block_grads& g = grads[i];
    BUCTD_CHECK_ARG(b.x && b.w1_bwd && b.w2_bwd && b.z1 && b.z2 && b.y && b.stat && g.dy && g.dres && g.dy1 && g.dw1 && g.dw2 &&
                        g.bn_acc && g.wg_ws && g.dz2 && g.dz1,
                    "buctd_basic_branches_bwd: null pointer in block %d", i);
  }
  hipStream_t main_s = (hipStream_t)stream, side_s = side_stream ? (hipStream_t)side_stream : main_s;
  static thread_local hipEvent_t evs[16] = {nullptr};
  int devid = 0;
  if (side_s != main_s && (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16)) {
    buctd_set_error("buctd_basic_branches_bwd: cannot identify the current device");
    return BUCTD_ELAUNCH;
  }
  hipEvent_t& ev = evs[devid];
  auto fork = [&]() -> int {
    if (side_s == main_s) return BUCTD_OK;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      buctd_set_error("buctd_basic_branches_bwd: hipEventCreate failed");
      return BUCTD_ELAUNCH;
    }
    if (hipEventRecord(ev, main_s) != hipSuccess || hipStreamWaitEvent(side_s, ev, 0) != hipSuccess) {
      buctd_set_error("buctd_basic_branches_bwd: stream fork failed");
      return BUCTD_ELAUNCH;
    }
    return BUCTD_OK;
  };
  bool ready[BR_MAX] = {false, false, false, false};     // block k's bn2 sums were formed by block k + 1's conv1 data gradient
  for (int k = n - 1; k >= 0; --k) {
    buctd_bn_bwd_item bi[BR_MAX];
    buctd_wg3_conv wg[BR_MAX];
    buctd_c3_conv cv[BR_MAX];
    bool chain[BR_MAX];
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      chain[b] = false;
      if (k > 0) {
        const buctd_basic_block& Bp = blocks[b * n + k - 1];
        const buctd_basic_block_grads& Gp = grads[b * n + k - 1];
        chain[b] = G.dx && G.dx == Gp.dy && Gp.bn_acc && Bp.y == B.x && Bp.N == B.N && Bp.H == B.H && Bp.W == B.W && Bp.C == C;
      }
      // conv2 / bn2 (+ skip): dres = masked upstream gradient
      bi[b] = buctd_bn_bwd_item{G.dy, B.y, B.z2, B.stat + 2 * C, B.stat + 3 * C, B.gamma2, nullptr, 1, (long)B.N * B.H * B.W, C,
                                G.dz2, G.dres, G.dgamma2, G.dbeta2, G.acc_bn2, (char*)G.bn_acc + buctd_bn_acc_bytes(C),
                                ready[b] ? 1 : 0};
    }
    BLK_TRY(buctd_bn_bwd_acc_group(nb, bi, stream));
    BLK_TRY(fork());
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      wg[b] = buctd_wg3_conv{B.N, B.H, B.W, C, C, B.z1, G.dz2, G.dw2, G.acc_w2, B.stat, B.stat + C, B.gamma1, B.beta1, 1,
                             G.wg_ws, G.wg_ws_bytes};
    }
    BLK_TRY(buctd_conv3x3_wgrad_bf16x6_group(nb, wg, side_s));
    // conv2's data gradients dy1, and with them the sums of bn1's backward (ReLU mask rebuilt from z1)
    memset(cv, 0, sizeof(cv));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      buctd_c3_conv& c = cv[b];
      c.N = B.N; c.H = B.H; c.W = B.W; c.Ci = c.Co = C;
      c.x = G.dz2; c.wprep = B.w2_bwd; c.y = G.dy1;
      c.bn_z = B.z1; c.bn_mean = B.stat; c.bn_invstd = B.stat + C; c.bn_gamma = B.gamma1; c.bn_beta = B.beta1; c.bn_acc = G.bn_acc;
    }
    BLK_TRY(buctd_conv3x3_bf16x6_group(nb, cv, stream));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      bi[b] = buctd_bn_bwd_item{G.dy1, nullptr, B.z1, B.stat, B.stat + C, B.gamma1, B.beta1, 1, (long)B.N * B.H * B.W, C,
                                G.dz1, nullptr, G.dgamma1, G.dbeta1, G.acc_bn1, G.bn_acc, 1};
    }
    BLK_TRY(buctd_bn_bwd_acc_group(nb, bi, stream));
    BLK_TRY(fork());
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      wg[b] = buctd_wg3_conv{B.N, B.H, B.W, C, C, B.x, G.dz1, G.dw1, G.acc_w1, nullptr, nullptr, nullptr, nullptr, 0,
                             G.wg_ws, G.wg_ws_bytes};
    }
    BLK_TRY(buctd_conv3x3_wgrad_bf16x6_group(nb, wg, side_s));
    // conv1's data gradients; the skip gradient joins in the epilogue; inside a chain the output is the upstream gradient of
    // the block in front, whose bn2 sums are formed on the way out
    int m = 0;
    memset(cv, 0, sizeof(cv));
    for (int b = 0; b < nb; ++b) {
      const buctd_basic_block& B = blocks[b * n + k];
      const buctd_basic_block_grads& G = grads[b * n + k];
      const int C = B.C;
      ready[b] = chain[b];
      if (!G.dx) continue;
      buctd_c3_conv& c = cv[m++];
      c.N = B.N; c.H = B.H; c.W = B.W; c.Ci = c.Co = C;
      c.x = G.dz1; c.wprep = B.w1_bwd; c.residual = G.dres; c.y = G.dx;
      if (chain[b]) {
        const buctd_basic_block& Bp = blocks[b * n + k - 1];
        const buctd_basic_block_grads& Gp = grads[b * n + k - 1];
        c.bn_z = Bp.z2; c.bn_y = Bp.y; c.bn_mean = Bp.stat + 2 * C; c.bn_invstd = Bp.stat + 3 * C; c.bn_gamma = Bp.gamma2;
        c.bn_acc = (char*)Gp.bn_acc + buctd_bn_acc_bytes(C);
      }
    }
    if (m) BLK_TRY(buctd_conv3x3_bf16x6_group(m, cv, stream));
  }
  return BUCTD_OK;
}

/* `to` waits for everything enqueued on `from` so far (event record + stream wait through one cached event per host thread
 * and device): the fork in front of a weight gradient launched on the side stream. */
extern "C" int buctd_stream_fork(void* from, void* to) {
  if (from == to) return BUCTD_OK;
  static thread_local hipEvent_t evs[16] = {nullptr};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 16) {
    buctd_set_error("buctd_stream_fork: cannot identify the current device");
    return BUCTD_ELAUNCH;
  }
  hipEvent_t& ev = evs[devid];
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    buctd_set_error("buctd_stream_fork: hipEventCreate failed");
    return BUCTD_ELAUNCH;
  }
  if (hipEventRecord(ev, (hipStream_t)from) != hipSuccess || hipStreamWaitEvent((hipStream_t)to, ev, 0) != hipSuccess) {
    buctd_set_error("buctd_stream_fork: stream fork failed");
    return BUCTD_ELAUNCH;
  }
  return BUCTD_OK;
}
