/* libbuctd_hip.so - C ABI of the MI355X (gfx950) BUCTD hot path.
 *
 * The reference (amathislab/BUCTD) has no FFI on this path: every device op is
 * a torch.nn module call that lands in cuDNN / cuBLAS / ATen.  Each entry point
 * below names the reference call site it stands in for (file:line relative to
 * the reference root).  INTEGRATION.md shows the ctypes binding the Python host
 * (buctd_amd/_C.py) uses.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (buctd_last_error());
 *  - no allocation, no ownership transfer, no synchronisation: the caller owns
 *    all buffers (device pointers), passes a hipStream_t as `void* stream`, and
 *    kernels are only enqueued on that stream;
 *  - activations are fp32 NHWC  [N][H][W][C]; conv weights are fp32
 *    [Co][R][S][Ci] (physical layout of a torch channels_last OIHW tensor);
 *    Linear weights are [out][in]; token tensors are [B][T][C];
 *  - "rows" = product of all leading dims, "C" = innermost (channel) dim.
 */
#ifndef BUCTD_HIP_H
#define BUCTD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int buctd_version(void);
const char* buctd_last_error(void);

/* ------------------------------------------------------------------ conv --- */
typedef struct {
  int N, H, W, Ci; /* input  [N][H][W][Ci]            */
  int Co, R, S;    /* filter [Co][R][S][Ci]           */
  int stride, pad; /* symmetric                       */
  int Ho, Wo;      /* output [N][Ho][Wo][Co]          */
} buctd_conv_desc;

/* y = conv(x,w) (+bias) ; then, in this order and each optional:
 *   stats_partials != NULL : per-(row-group, channel) Welford partials (mean, M2) of the
 *                            biased conv output for train-mode BatchNorm (see buctd_bn_finalize);
 *   scale/shift   != NULL  : y = y*scale[c] + shift[c]   (eval-mode BatchNorm folded);
 *   residual      != NULL  : y += residual ; relu != 0 : y = max(y,0).
 * Replaces nn.Conv2d(+BatchNorm2d+ReLU+residual): pose_hrnet.py:28-98, pose_hrnet_coam.py:44-60,
 * nn.Linear on token tensors (R=S=1): self_attention.py:74-76,87. */
/* 1 when buctd_conv2d_fwd runs this shape on the thin-output kernel (stride-1 'same' 7x7 with <= 4 output channels: the
 * full-resolution preNet convolutions, pose_hrnet.py:431-442) - it then takes no fused epilogue: call it with bias
 * only and run buctd_bn_stats on the (3-channel) result for a train-mode BatchNorm. */
int buctd_conv2d_fwd_thin(const buctd_conv_desc* d);
int buctd_conv2d_fwd(const buctd_conv_desc* d, const float* x, const float* w, const float* bias,
                     const float* scale, const float* shift, const float* residual, int relu, float* y,
                     float* stats_partials, void* stream);
/* dx = conv_transpose(dy, w) (+bias, + Welford partials over dx): the data gradient of
 * buctd_conv2d_fwd, and the forward of nn.ConvTranspose2d (pose_resnet.py:197-205). */
int buctd_conv2d_dgrad(const buctd_conv_desc* d, const float* dy, const float* w, const float* bias, float* dx,
                       float* stats_partials, void* stream);
/* number / height of the Welford row groups written by fwd (transposed=0, channels=Co, rows=N*Ho*Wo)
 * or dgrad (transposed=1, channels=Ci, rows=N*H*W); partial buffer = ngroups*channels*2 floats. */
int buctd_conv2d_stats_groups(const buctd_conv_desc* d, int transposed, int* ngroups, int* rows_per_group);
/* dw (+)= sum over pixels dy (x) x ; split over pixels through `workspace`. */
size_t buctd_conv2d_wgrad_workspace(const buctd_conv_desc* d);
int buctd_conv2d_wgrad(const buctd_conv_desc* d, const float* x, const float* dy, float* dw, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);

/* 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores with split-fp32 operands, fp32 accumulate
 * (csrc/conv3x3.hip).  Replaces the BasicBlock convs of pose_hrnet.py:28-57 (nn.Conv2d(k=3, s=1, p=1, bias=False)
 * forward and its autograd data gradient).  Same epilogue options as buctd_conv2d_fwd.  Two families of entry points:
 *   buctd_conv3x3_bf16x6_*  fp32-class (the engine's default): x = h + m + l exactly (three bf16 pieces), six MFMAs
 *                           per product; dropped piece products are <= 2^-24 of the product;
 *   buctd_conv3x3_bf16x3_*  optional reduced precision: two pieces, three MFMAs, ~2^-16 per product.
 *
 * The filter is consumed as a prepared image (bf16 pieces in the kernel's stage order), produced once per
 * weight update by the family's _prep from the forward filter w = [Co][3][3][Ci]:
 *   flip = 0: image for the forward convolution (Ci -> Co);
 *   flip = 1: image for its data gradient (the call below then takes x = dy [N][H][W][Co] with "Ci" = Co and
 *             produces y = dx [N][H][W][Ci] with "Co" = Ci).
 * A prepared image belongs to the family that made it.
 * stats_counts: ngroups ints (valid rows per group; pad positions of the flattened tile are skipped). */
int buctd_conv3x3_bf16x6_supported(int N, int H, int W, int Ci, int Co);
int buctd_conv3x3_bf16x6_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group);
size_t buctd_conv3x3_bf16x6_prep_bytes(int Ci, int Co, int flip);
int buctd_conv3x3_bf16x6_prep(int Ci, int Co, const float* w, int flip, void* wprep, void* stream);
int buctd_conv3x3_bf16x6(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                         const float* scale, const float* shift, const float* residual, int relu, float* y,
                         float* stats_partials, int* stats_counts, void* stream);
/* buctd_conv3x3_bf16x6 with the BatchNorm(+ReLU) of the PRODUCING layer applied to the input while it is staged: x is
 * the producer's raw convolution output z, the kernel convolves relu((z - mean) * (invstd * gamma) + beta) - bitwise
 * what buctd_bn_apply would have written - so the tensor between bn1/relu and conv2 of a BasicBlock
 * (pose_hrnet.py:44-49) never exists in HBM.  in_* are [Ci] arrays. */
int buctd_conv3x3_bf16x6_bnin(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                              const float* scale, const float* shift, const float* residual, int relu, float* y,
                              float* stats_partials, int* stats_counts, const float* in_mean, const float* in_invstd,
                              const float* in_gamma, const float* in_beta, int in_relu, void* stream);
/* Data gradient of a 3x3 convolution (buctd_conv3x3_bf16x6 on the flip = 1 image; residual = the skip gradient, NULL ok)
 * that also forms the REDUCTION pass of the BatchNorm backward consuming its output g = y: per row group (the groups of
 * buctd_conv3x3_bf16x6_stats_groups(N, H, W, Ci, Co)) s1 = sum m g, s2 = sum m g (z - mean) invstd, with m the ReLU mask of
 * that BatchNorm's forward output (bn_y > 0 where given, else rebuilt from bn_z as (z - mean)(invstd gamma) + beta > 0) -
 * bn_part [groups][2][Co] floats, the input of buctd_bn_bwd_from_partials.  What autograd does with three more passes over
 * g, z and y (the sum reductions of native_batch_norm_backward, pose_hrnet.py:41-57). */
int buctd_conv3x3_bf16x6_bnstat(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* residual,
                                float* y, const float* bn_z, const float* bn_y, const float* bn_mean, const float* bn_invstd,
                                const float* bn_gamma, const float* bn_beta, float* bn_part, void* stream);
/* The accumulator forms (see "BatchNorm statistics without finalize launches" below).
 * _acc: forward convolution whose output statistics are added to stats_acc (NULL: none) and whose input may be the raw
 * output of the producing convolution, normalised while it is staged with THAT layer's statistics decoded from its
 * accumulator (in_bn != NULL: with in_gamma / in_beta / in_relu; the launch's first tile writes mean / invstd out and updates
 * the running statistics).  _bnstat_acc: buctd_conv3x3_bf16x6_bnstat with the sums added to the accumulator bn_acc. */
struct buctd_bn_acc_in_;
int buctd_conv3x3_bf16x6_acc(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* residual,
                             int relu, float* y, void* stats_acc, const struct buctd_bn_acc_in_* in_bn, const float* in_gamma,
                             const float* in_beta, int in_relu, void* stream);
int buctd_conv3x3_bf16x6_bnstat_acc(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep,
                                    const float* residual, float* y, const float* bn_z, const float* bn_y,
                                    const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                    void* bn_acc, void* stream);
/* Several independent 3x3 convolutions in ONE launch - the k-th convolutions of the 2-4 branches of a HighResolutionModule
 * (pose_hrnet.py:177-185), which cost the same FLOPs on maps of different size.  The tiles of all of them form one grid
 * (costliest first), so the launch is several rounds of workgroups that de-phase instead of one phase-locked round per
 * convolution.  Each entry is one buctd_conv3x3_bf16x6_acc call (forward: stats_acc / in_bn ...) or one
 * buctd_conv3x3_bf16x6_bnstat_acc call (data gradient: bn_acc != NULL with bn_z ...; without bn_acc a plain data gradient);
 * results are bit-identical to those calls.  n <= 4. */
typedef struct {
  int N, H, W, Ci, Co;
  const float* x;
  const void* wprep;
  const float* residual;
  int relu;
  float* y;
  void* stats_acc;                         /* forward statistics of y (NULL: none) */
  const struct buctd_bn_acc_in_* in_bn;    /* input BatchNorm from the producer's accumulator (NULL: none) */
  const float *in_gamma, *in_beta;
  int in_relu;
  const float *bn_z, *bn_y, *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   /* BatchNorm-backward by-product ... */
  void* bn_acc;                                                          /* ... into this accumulator (NULL: none) */
} buctd_c3_conv;
int buctd_conv3x3_bf16x6_group(int n, const buctd_c3_conv* convs, void* stream);
/* The same for EVAL mode (validate(), lib/core/function.py:178-336; inference): the k-th convolutions of the branches with the
 * folded BatchNorm (y = conv * scale + shift), the skip connection and the ReLU in the epilogue - each entry is one
 * buctd_conv3x3_bf16x6(N, H, W, Ci, Co, x, wprep, NULL, scale, shift, residual, relu, y, NULL, NULL, stream) call, bit-identical
 * to it; members that share no kernel go out one launch each.  n <= 4. */
typedef struct {
  int N, H, W, Ci, Co;
  const float* x;
  const void* wprep;
  const float *scale, *shift;     /* both or neither */
  const float* residual;
  int relu;
  float* y;
} buctd_c3_conv_eval;
int buctd_conv3x3_bf16x6_group_eval(int n, const buctd_c3_conv_eval* convs, void* stream);
/* Train-mode launches (the option sets of block.hip: statistics accumulator, input BatchNorm from an accumulator, skip
 * gradient, BatchNorm-backward sums) run kernels specialised on the option set (csrc/conv3x3_lean.hip).  Their PERSISTENT
 * form (csrc/conv3x3_pers.hip: a grid of resident workgroups, each walking its share of the tiles as one software pipeline
 * across tile boundaries) is an opt-in: bit-identical results, measured slower than the dispatcher-scheduled launches on
 * HRNet-W48.  on = 1 / 0 switches it, on < 0 only queries; returns the previous setting.  Process-wide; call
 * it before launching from several threads. */
int buctd_conv3x3_bf16x6_persistent(int on);
/* The number of workgroups buctd_conv3x3_bf16x6_group(n, convs) launches as ONE kernel (0: the members share no kernel and go
 * out one launch each; < 0: error).  No launch - for tools that find a launch in a kernel trace by its grid (bench.py). */
int buctd_conv3x3_bf16x6_group_workgroups(int n, const buctd_c3_conv* convs);
int buctd_conv3x3_bf16x3_supported(int N, int H, int W, int Ci, int Co);
int buctd_conv3x3_bf16x3_stats_groups(int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group);
size_t buctd_conv3x3_bf16x3_prep_bytes(int Ci, int Co, int flip);
int buctd_conv3x3_bf16x3_prep(int Ci, int Co, const float* w, int flip, void* wprep, void* stream);
/* The same for n filters in ONE launch (every prepared image of a model after an optimizer step), either family:
 * item.reserved = 3 writes the bf16x6 image, anything else the bf16x3 one.  `items_device` is an array in device
 * memory; an item has steps * Nc * 8 pieces (= prep_bytes / 16 for bf16x3, / 24 for bf16x6); piece_begin is the running
 * sum over the preceding items and total_pieces the sum over all of them. */
typedef struct {
  const float* w;       /* forward filter [Co][3][3][Ci] */
  void* wprep;          /* destination image */
  int Ci, Co, flip, reserved;
  long piece_begin;
} buctd_c3_prep_item;
int buctd_conv3x3_bf16x3_prep_batched(const buctd_c3_prep_item* items_device, int n, long total_pieces, void* stream);
int buctd_conv3x3_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                         const float* scale, const float* shift, const float* residual, int relu, float* y,
                         float* stats_partials, int* stats_counts, void* stream);

/* weight gradient of the same convolution on the bf16 matrix cores (autograd of nn.Conv2d, pose_hrnet.py:28-57;
 * transpose-read fragments from position-major LDS tiles, split over positions through `workspace`):
 * dw (+)= sum_p dy[p] (x) x[p + tap].  dw: [Co][3][3][Ci].  Families as above. */
int buctd_conv3x3_wgrad_bf16x6_supported(int N, int H, int W, int Ci, int Co);
size_t buctd_conv3x3_wgrad_bf16x6_workspace(int N, int H, int W, int Ci, int Co);
int buctd_conv3x3_wgrad_bf16x6(int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                               int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* the weight gradient with the same on-the-fly BatchNorm(+ReLU) of its X operand (x = the producer's raw output) */
int buctd_conv3x3_wgrad_bf16x6_bnin(int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                                    int accumulate, const float* x_mean, const float* x_invstd, const float* x_gamma,
                                    const float* x_beta, int x_relu, void* workspace, size_t workspace_bytes,
                                    void* stream);
/* The weight gradients of the k-th convolutions of the 2-4 branches of a HighResolutionModule (pose_hrnet.py:177-185) in ONE
 * launch (+ one for their slab reductions).  Convolutions that share a launch fill the chip together, so each is cut into
 * fewer position splits than it would be alone: less halo re-reading, fewer slabs (a different, still fixed, summation order
 * than the single call: deterministic, fp32-class, not bit-identical to it).  Each item needs its OWN workspace of
 * buctd_conv3x3_wgrad_bf16x6_group_workspace(n, ...) bytes (n = convolutions in the launch); x_mean != NULL: the BatchNorm
 * (+ReLU) of the producer applied to x while it is staged, as in buctd_conv3x3_wgrad_bf16x6_bnin.  n <= 4. */
typedef struct {
  int N, H, W, Ci, Co;
  const float *x, *dy;
  float* dw;
  int accumulate;
  const float *x_mean, *x_invstd, *x_gamma, *x_beta;
  int x_relu;
  void* workspace;
  size_t workspace_bytes;
} buctd_wg3_conv;
size_t buctd_conv3x3_wgrad_bf16x6_group_workspace(int n, int N, int H, int W, int Ci, int Co);
int buctd_conv3x3_wgrad_bf16x6_group(int n, const buctd_wg3_conv* convs, void* stream);
/* workgroups of the weight-gradient kernel that call launches (no launch; tools that search a kernel trace by grid: bench.py) */
int buctd_conv3x3_wgrad_bf16x6_group_workgroups(int n, const buctd_wg3_conv* convs);
int buctd_conv3x3_wgrad_bf16x3_supported(int N, int H, int W, int Ci, int Co);
size_t buctd_conv3x3_wgrad_bf16x3_workspace(int N, int H, int W, int Ci, int Co);
int buctd_conv3x3_wgrad_bf16x3(int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                               int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- matmul --- */
typedef struct {
  int batch, M, N, K;
  int a_layout;            /* 0: A(m,k)=A[m*lda + (k/Kc)*group_stride_a + k%Kc] ; 1: A(m,k)=A[k*lda+m]   */
  int b_layout;            /* 0: B(k,n)=B[n*ldb + (k/Kc)*group_stride_bk + k%Kc] ;
                              1: B(k,n)=B[k*ldb + (n/Nc)*group_stride_bn + n%Nc]                          */
  int lda, ldb, ldc;       /* C(m,n)=C[m*ldc + (n/Nc)*group_stride_c + n%Nc]                              */
  long stride_a, stride_b, stride_c; /* per-batch element strides (0 = shared operand) */
  int Kc; long group_stride_a, group_stride_bk;
  int Nc; long group_stride_bn, group_stride_c;
  float alpha;
  int bias_axis;           /* 0: bias[n], 1: bias[m] (only read when bias != NULL) */
} buctd_matmul_desc;
/* C = alpha*A*B (+bias). Replaces torch.matmul / nn.Linear in self_attention.py:78,86,150,158-159. */
size_t buctd_matmul_workspace(const buctd_matmul_desc* d);
int buctd_matmul(const buctd_matmul_desc* d, const float* A, const float* B, const float* bias, float* C,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- batchnorm --- */
/* Combine Welford partials -> mean, invstd (biased var, eps) and update running stats
 * (momentum, unbiased var) exactly like nn.BatchNorm2d in train mode (pose_hrnet.py:37).
 * group_counts (NULL ok): valid rows of each group when they are not rows_per_group-regular. */
int buctd_bn_finalize(const float* partials, const int* group_counts, int ngroups, int rows_per_group, long rows,
                      int C, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                      float* running_var, void* stream);
/* Welford partials of a plain [rows][C] tensor (when the producer was not a conv epilogue). */
int buctd_bn_stats(const float* z, long rows, int C, float* partials, int* ngroups, int* rows_per_group,
                   void* stream);
int buctd_bn_stats_groups(long rows, int C, int* ngroups, int* rows_per_group);
/* y = act((z-mean)*invstd*gamma+beta (+residual)) */
int buctd_bn_apply(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   const float* residual, int relu, float* y, long rows, int C, void* stream);
/* Backward of bn_apply in train mode. y is the forward output (ReLU mask), may be NULL when relu==0.  With relu != 0
 * and y == NULL the mask is rebuilt bit-exactly from z as (z-mean)*(invstd*gamma)+beta > 0 - valid only when the
 * forward had no residual - which saves one full read of the activation tensor per pass (beta is read only then).
 * Writes dz; dres (NULL ok) receives the masked upstream gradient (gradient of the residual input);
 * dgamma/dbeta are overwritten (accumulate=0) or added to. workspace: buctd_bn_bwd_workspace bytes. */
size_t buctd_bn_bwd_workspace(long rows, int C);
int buctd_bn_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, int relu, long rows, int C, float* dz, float* dres,
                 float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The rest of buctd_bn_bwd when the reduction pass already happened (buctd_conv3x3_bf16x6_bnstat): part = [nparts][2][C]
 * partial (s1, s2) sums -> fp64 merge, dgamma / dbeta, then dz (and dres).  workspace: 2 * C floats. */
int buctd_bn_bwd_from_partials(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int relu, long rows, int C, const float* part,
                               int nparts, float* dz, float* dres, float* dgamma, float* dbeta, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream);
/* ---- BatchNorm statistics without finalize launches (csrc/bn_acc.h) ----
 * The kernel that produces a tensor reduces it to one pair of sums per workgroup and channel and adds the pair into an
 * ACCUMULATOR with integer atomics (fixed point, two 64-bit limbs per sum: order-independent, so bit-deterministic); the
 * kernel that consumes the statistics decodes two numbers per channel in its prologue.  Nothing is launched between the
 * two (the mean/var reduction of native_batch_norm and the sum reductions of native_batch_norm_backward,
 * pose_hrnet.py:41-57, as by-products of their neighbours).  An accumulator is buctd_bn_acc_bytes(C) bytes and must be ZERO
 * when its producer is launched.
 * buctd_bn_acc_in: how a consumer takes FORWARD statistics from an accumulator holding (sum z, sum z^2) over `rows` values per
 * channel: it derives mean / invstd (biased variance, eps) itself; ONE workgroup of the launch also writes them to
 * mean_out / invstd_out ([C] each: what the backward kernels read) and updates running_mean / running_var (NULL: not
 * tracked) with `momentum` and the unbiased variance, like nn.BatchNorm2d in train mode. */
size_t buctd_bn_acc_bytes(int C);
typedef struct buctd_bn_acc_in_ {
  const void* acc;
  long rows;
  float eps, momentum;
  float *mean_out, *invstd_out;
  float *running_mean, *running_var;
} buctd_bn_acc_in;
/* buctd_bn_apply with the statistics taken from an accumulator (C % 4 == 0) */
int buctd_bn_apply_acc(const float* z, const buctd_bn_acc_in* st, const float* gamma, const float* beta,
                       const float* residual, int relu, float* y, long rows, int C, void* stream);
/* buctd_bn_bwd on an accumulator of the backward sums (sum g, sum g zhat): acc_ready = 0 runs the streaming reduction into
 * `acc` (zero on entry) first; acc_ready = 1: the data gradient that produced dy already filled it
 * (buctd_conv3x3_bf16x6_bnstat_acc).  The apply kernel decodes the sums and writes dgamma / dbeta.  C % 4 == 0, C <= 1024. */
int buctd_bn_bwd_acc(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, long rows, int C, float* dz, float* dres,
                     float* dgamma, float* dbeta, int accumulate, void* acc, int acc_ready, void* stream);
/* The same for the 2-4 branch tensors of a HighResolutionModule (pose_hrnet.py:177-185) in ONE launch per kernel kind: every
 * tensor is processed exactly as by its own buctd_bn_apply_acc / buctd_bn_bwd_acc call (bit-identical results).  n <= 4. */
typedef struct {
  const float* z;
  buctd_bn_acc_in st;
  const float *gamma, *beta, *residual;
  int relu;
  float* y;
  long rows;
  int C;
} buctd_bn_apply_item;
int buctd_bn_apply_acc_group(int n, const buctd_bn_apply_item* items, void* stream);
typedef struct {
  const float *dy, *y, *z, *mean, *invstd, *gamma, *beta;
  int relu;
  long rows;
  int C;
  float *dz, *dres, *dgamma, *dbeta;
  int accumulate;
  void* acc;
  int acc_ready;
} buctd_bn_bwd_item;
int buctd_bn_bwd_acc_group(int n, const buctd_bn_bwd_item* items, void* stream);
/* Backward of a fuse row (lib/models/pose_hrnet.py:257-265, y = relu(sum_j upsample(term_j))) for the terms of one shift,
 * exactly buctd_fuse_sum_bwd (g [N][H>>shift][W>>shift][C] = window sum of dy * (y > 0); y may be NULL), plus the
 * BatchNorm-backward sums of nt <= 3 terms that are conv -> BatchNorm outputs of that resolution: sum(g) and
 * sum(g * (z_t - mean_t) * invstd_t) are added into acc[t] (buctd_bn_acc_bytes(C) each, zero on entry), so that
 * buctd_bn_bwd_acc(..., acc[t], acc_ready = 1) can follow without its reduction launch. */
int buctd_fuse_sum_bwd_bnstat(const float* dy, const float* y, int shift, int N, int H, int W, int C, float* g, int nt,
                              const float* const* z, const float* const* mean, const float* const* invstd,
                              void* const* acc, void* stream);
/* eval-mode helpers: scale = gamma/sqrt(var+eps), shift = beta - mean*scale */
int buctd_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                  float eps, int C, float* scale, float* shift, void* stream);

/* ----------------------------------------------------------- elementwise --- */
/* out[i] = a[i] + b[i] (b may be NULL -> copy); relu optional */
int buctd_add(const float* a, const float* b, float* out, long n, int relu, void* stream);
/* out = terms[0] + ... + terms[n-1] (2 <= n <= 4, host array of device pointers), summed left to right: the gradient
 * fan-in of a tensor with several consumers (HRNet fuse rows, pose_hrnet.py:257-265) in one pass */
int buctd_add_n(const float* const* terms, int n, float* out, long count, void* stream);
/* out[i] = a[i] * b[i]: DAModule with MODEL.ATT_CHANNEL_ONLY, `input * c_out` (pose_hrnet_coam.py:716-717) */
int buctd_mul(const float* a, const float* b, float* out, long n, void* stream);
/* out[i] = x[i] * alpha * (*dev_scalar) ; dev_scalar is a device pointer or NULL (chain rule through
 * the scalar loss without a host sync: loss.backward() hands d(loss) over as a device scalar) */
int buctd_scale(const float* x, const float* dev_scalar, float alpha, float* out, long n, void* stream);
/* dst[r][cd0+c] = src[r][cs0+c] for c < Cc: channel concat / split on [rows][C] tensors
 * (torch.cat((x, x_cond), dim=1), transpose_h.py:672) */
int buctd_copy_channels(const float* src, long rows, int Cs, int cs0, float* dst, int Cd, int cd0, int Cc,
                        void* stream);
/* out[b][i] = a[b][i] + v[i]: position embedding added to each image's tokens (transpose_h.py:189-192) */
int buctd_add_bcast(const float* a, const float* v, float* out, long batch, long n, void* stream);
/* dx = dy * (y > 0) */
int buctd_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream);
/* column sums of [rows][C] (bias gradients); db overwritten or accumulated */
size_t buctd_colsum_workspace(long rows, int C);
int buctd_colsum(const float* x, long rows, int C, float* out, int accumulate, void* workspace,
                 size_t workspace_bytes, void* stream);
/* NCHW channel slice [c0, c0+Cc) of x[N][Ctot][H][W]  ->  NHWC [N][H][W][Cc], and back (full tensor). */
int buctd_nchw_to_nhwc(const float* x, int N, int Ctot, int c0, int Cc, int H, int W, float* y, void* stream);
int buctd_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y, void* stream);
/* HighResolutionModule fuse (pose_hrnet.py:250-265): out = relu(sum_j upsample_nearest(term_j, 2^shift_j)).
 * terms[j] is [N][H>>shift_j][W>>shift_j][C]; up to 4 terms. */
int buctd_fuse_sum(const float* const* terms, const int* shifts, int nterms, int N, int H, int W, int C, int relu,
                   float* out, void* stream);
/* gradient of one fuse term: g[n][h][w][c] = sum over its 2^shift x 2^shift block of dy*(y>0) (y NULL: no mask) */
int buctd_fuse_sum_bwd(const float* dy, const float* y, int shift, int N, int H, int W, int C, float* g,
                       void* stream);
/* torchvision TF.resize (bilinear, align_corners=False, no antialias; pose_hrnet_coam.py:755) of the NCHW
 * channel slice [c0, c0+Cc) of x -> NHWC [N][Ho][Wo][Cc] */
int buctd_resize_bilinear(const float* x, int N, int Ctot, int c0, int Cc, int H, int W, int Ho, int Wo, float* y,
                          void* stream);
/* MaxPool2d(3, stride 2, pad 1) NHWC (pose_resnet.py:122) forward (argmax index saved) / backward */
int buctd_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int32_t* idx, void* stream);
int buctd_maxpool3x3s2_bwd(const float* dy, const int32_t* idx, int N, int H, int W, int C, float* dx,
                           void* stream);

/* ------------------------------------------------------------- attention --- */
/* Row softmax with fused scale and inverted dropout (self_attention.py:78-84):
 *   p = softmax(scale * s) over the last dim (length L);  pd = p * keep / (1-p_drop),
 *   keep drawn from a counter-based generator keyed by (seed, row, col) so the backward can rebuild it.
 * p (pre-dropout) and pd may alias when p_drop == 0. */
int buctd_softmax_dropout_fwd(const float* s, long rows, int L, float scale, float p_drop, uint64_t seed, float* p,
                              float* pd, void* stream);
/* ds = scale * p * (g - sum_j g_j p_j), g = dpd * keep/(1-p_drop) */
int buctd_softmax_dropout_bwd(const float* dpd, const float* p, long rows, int L, float scale, float p_drop,
                              uint64_t seed, float* ds, void* stream);
/* Fused position attention for narrow query/key contractions (self_attention.py:74-86 with fc_q folded into the
 * keys: logits = scale * q' k'^T with q' = [y_cond, 1, 0-pad] and k' = fc_k(y) [Wq | bq | 0], both [B][T][R4]).
 * out = dropout(softmax(logits)) v, v and out [B][T][C]; nothing of size T x T is written to memory.
 * m / linv [B][T] receive the row max and reciprocal row sum (saved for the backward).  The dropout mask is a
 * counter hash of (seed, b*T + i, j), rebuilt by the backward.  Shapes: T % 64 == 0, R4 in {4,8,16,20},
 * C in {16,32,48,64,96,128,192} (buctd_attn_smallqk_supported).  bf16x3 != 0 selects the variants whose T- and
 * C-contractions run on the bf16 matrix cores with split-fp32 operands: 1 = two pieces per operand (accuracy class of
 * buctd_conv3x3_bf16x3), 2 = three pieces, six MFMAs per product (fp32 class, as buctd_conv3x3_bf16x6; C <= 128).
 * R4 <= 8 only; wider contractions silently use the fp32 kernels. */
int buctd_attn_smallqk_supported(int T, int R4, int C);
int buctd_attn_smallqk_fwd(int B, int T, int R4, int C, const float* q, const float* k, const float* v, float scale,
                           float p_drop, uint64_t seed, int bf16x3, float* out, float* m, float* linv, void* stream);
/* dq/dk: [B][T][R4] gradients of q'/k'; dv: [B][T][C]; dvec_workspace: B*T floats. */
int buctd_attn_smallqk_bwd(int B, int T, int R4, int C, const float* q, const float* k, const float* v,
                           const float* o, const float* dout, const float* m, const float* linv, float scale,
                           float p_drop, uint64_t seed, int bf16x3, float* dq, float* dk, float* dv,
                           float* dvec_workspace, void* stream);
/* elementwise inverted dropout (transpose_h.py:180-183), mask rebuilt from (seed, index) */
int buctd_dropout(const float* x, float* y, long n, float p_drop, uint64_t seed, void* stream);
/* LayerNorm over the last dim (transpose_h.py:178-179) */
int buctd_layernorm_fwd(const float* x, const float* gamma, const float* beta, long rows, int C, float eps, float* y,
                        float* mean, float* invstd, void* stream);
/* LayerNorm(a + b) in one pass (the post-norm residual of the encoder layer, transpose_h.py:204-209); sum_out (NULL ok)
 * receives a + b - the tensor buctd_layernorm_bwd takes as x. */
int buctd_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, long rows, int C, float eps,
                            float* sum_out, float* y, float* mean, float* invstd, void* stream);
size_t buctd_layernorm_bwd_workspace(long rows, int C);
int buctd_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
                        long rows, int C, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------- loss / target / decode --- */
/* JointsMSELoss (core/loss.py:23-41): loss = 0.5/(K*N*HW) * sum w^2 (p-g)^2 ; grad (NULL ok) = dloss/dp * gscale.
 * pred/gt are [N][K][HW] (NCHW heatmaps), w is [N][K] (NULL: weights 1). workspace: N*K floats. */
int buctd_joints_mse(const float* pred, const float* gt, const float* w, int N, int K, int HW, float* loss,
                     float* grad, float gscale, void* workspace, size_t workspace_bytes, void* stream);
/* get_max_preds (core/inference.py:19-47): first-index argmax per [N*K] row of length H*W;
 * preds[row] = (x, y) zeroed where maxval <= 0. */
int buctd_argmax_decode(const float* hm, int rows, int H, int W, float* preds, float* maxvals, int32_t* idx,
                        void* stream);
/* the same plus the POST_PROCESS refinement of get_final_preds (core/inference.py:68-77): quarter[row] = the
 * (+-0.25 | 0, +-0.25 | 0) offset towards the higher neighbour for interior peaks, (0, 0) otherwise. */
int buctd_argmax_decode_refined(const float* hm, int rows, int H, int W, float* preds, float* maxvals, int32_t* idx,
                                float* quarter, void* stream);
/* generate_target (dataset/JointsDataset.py:397-453): joints [B][K][3] (crop px), vis [B][K] ->
 * target [B][K][Hh][Wh], weight [B][K]. */
int buctd_gaussian_target(const float* joints, const float* vis, int B, int K, int Hh, int Wh, float stride_x,
                          float stride_y, float sigma, float* target, float* weight, void* stream);
/* get_condition_image(_colored) (dataset/JointsDataset.py:500-543): joints [B][K][2+] (row stride js floats),
 * colors [K][Cc] (NULL: 255 mono) -> cond [B][Cc][H][W] in [0,255]; 15x15 Gaussian blur sigma 2.6 (reflect-101),
 * peak-normalised to 255 per image; truncate != 0 applies the mono path's .astype(int). workspace: B*Cc*H*W floats + B floats. */
size_t buctd_cond_render_workspace(int B, int Cc, int H, int W);
int buctd_cond_render(const float* joints, int js, const float* colors, int B, int K, int Cc, int H, int W,
                      int truncate, float* cond, void* workspace, size_t workspace_bytes, void* stream);
/* flip-test merge (core/function.py:226-236): out = 0.5*(a + shift(flip_back(b))) on [N][K][H][W];
 * perm[k] = source channel after the left/right swap; shift != 0 applies the 1-px SHIFT_HEATMAP. */
int buctd_flipback_avg(const float* a, const float* b, const int32_t* perm, int N, int K, int H, int W, int shift,
                       float* out, void* stream);

/* ------------------------------------------- gathered bf16x6 convolutions --- */
/* The convolutions around the BasicBlocks in the bf16x6 arithmetic (csrc/conv_gather_x6.hip): kind 1 = 1x1 / stride 1 /
 * pad 0 (fuse layers pose_hrnet.py:187-245, Bottlenecks :60-108), kind 2 = 3x3 / stride 2 / pad 1 (transitions :338-372,
 * fuse-layer downsampling), forward (dir 0) and data gradient (dir 1; a stride-2 data gradient runs its four output
 * parities as one launch).  (H, W) are the INPUT dims of the convolution, NHWC fp32 tensors, w is [Co][R][S][Ci];
 * Ci % 16 == 0, Co % 16 == 0, output channels of the launch a multiple of 48 or 64, H and W even for kind 2.
 * prep builds the weight image the kernels consume - once per weight update, per direction.  The forward takes the
 * epilogue options and Welford BatchNorm partials + counts of buctd_conv3x3_bf16x6 (groups from _stats_groups); the data
 * gradient adds `residual` (NULL ok) to dx.  Replace what cuDNN does for F.conv2d / its backward on those layers. */
int buctd_gconv_x6_supported(int kind, int N, int H, int W, int Ci, int Co, int dir);
size_t buctd_gconv_x6_prep_bytes(int kind, int Ci, int Co, int dir);
int buctd_gconv_x6_prep(int kind, int Ci, int Co, const float* w, int dir, void* wprep, void* stream);
/* all images of a model with ONE launch (after an optimizer step rewrote the filters in place): the caller fills one table
 * entry per (filter, direction) on the host with _prep_item (entry size _prep_item_bytes), uploads the table once, and
 * launches _prep_batched on it whenever the filters change. */
size_t buctd_gconv_x6_prep_item_bytes(void);
int buctd_gconv_x6_prep_item(int kind, int Ci, int Co, const float* w, int dir, void* wprep, void* item_host);
int buctd_gconv_x6_prep_batched(const void* items_device, int n, void* stream);
int buctd_gconv_x6_stats_groups(int kind, int N, int H, int W, int Ci, int Co, int* ngroups, int* rows_per_group);
int buctd_gconv_x6_fwd(int kind, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                       const float* scale, const float* shift, const float* residual, int relu, float* y,
                       float* stats_partials, int* stats_counts, void* stream);
/* forward with the output statistics added to an accumulator (buctd_bn_acc_bytes(Co), zero on entry) instead of partials */
int buctd_gconv_x6_fwd_acc(int kind, int N, int H, int W, int Ci, int Co, const float* x, const void* wprep, const float* bias,
                           float* y, void* stats_acc, void* stream);
int buctd_gconv_x6_dgrad(int kind, int N, int H, int W, int Ci, int Co, const float* dy, const void* wprep,
                         const float* residual, float* dx, void* stream);
/* Weight gradient of a 1x1 (kind 1) or stride-2 3x3 pad-1 (kind 2; H, W even) convolution Ci -> Co in the bf16x6 arithmetic
 * (csrc/conv_gather_wgrad.hip): dw [Co][R][S][Ci] (+)= sum over output pixels dy (x) gathered x, x = [N][H][W][Ci], dy =
 * [N][Ho][Wo][Co].  The autograd weight gradient of the fuse-layer / transition / Bottleneck convolutions
 * (pose_hrnet.py:60-108, 187-245, 338-372); both channel counts must be multiples of 48, or both of 32. */
int buctd_gconv_wgrad_x6_supported(int kind, int N, int H, int W, int Ci, int Co);
size_t buctd_gconv_wgrad_x6_workspace(int kind, int N, int H, int W, int Ci, int Co);
int buctd_gconv_wgrad_x6(int kind, int N, int H, int W, int Ci, int Co, const float* x, const float* dy, float* dw,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------- BasicBlock sequences --- */
/* The kernel sequence of one residual BasicBlock in train mode (pose_hrnet.py:28-57: stride 1, C -> C, no downsample,
 * bf16x6 math) behind ONE call per direction: conv1 (+ statistics), conv2 with bn1 + ReLU applied in its input staging
 * (+ statistics), bn2 + skip + ReLU - three launches, the BatchNorm statistics travel as accumulators (no finalize) - and
 * the mirrored backward (two BatchNorm-backward applies, two data gradients that also form the BatchNorm sums, two weight
 * gradients on `side_stream`; NULL: same stream).  Pure launch sequences of the entry points above (bit-identical
 * results); they exist because a dozen calls per block through a Python binding cost more host time than HRNet-W32 needs
 * GPU time.  acc: 2 * buctd_bn_acc_bytes(C) ZEROED bytes (forward statistics of conv1 | conv2); stat: 4 * C floats receiving
 * mean1, invstd1, mean2, invstd2 (saved for the backward).  running_* may be NULL. */
typedef struct {
  int N, H, W, C;
  const float* x;
  const void *w1_fwd, *w2_fwd;          /* prepared images, flip = 0 */
  const void *w1_bwd, *w2_bwd;          /* prepared images, flip = 1 (backward only) */
  const float *gamma1, *beta1, *gamma2, *beta2;
  float *running_mean1, *running_var1, *running_mean2, *running_var2;
  float eps1, momentum1, eps2, momentum2;
  float *z1, *z2, *y;
  void* acc;
  float* stat;
} buctd_basic_block;
typedef struct {
  const float* dy;                      /* gradient of the block output */
  float *dz2, *dres, *dy1, *dz1;        /* scratch tensors of the block's shape */
  float* dx;                            /* NULL: the block input needs no gradient */
  float *dw1, *dw2, *dgamma1, *dbeta1, *dgamma2, *dbeta2;
  int acc_w1, acc_w2, acc_bn1, acc_bn2; /* accumulate into (1) or overwrite (0) the gradient buffers */
  void* bn_acc;                         /* 2 * buctd_bn_acc_bytes(C) ZEROED bytes: backward sums of bn1 | bn2 */
  void* wg_ws; size_t wg_ws_bytes;      /* buctd_conv3x3_wgrad_bf16x6_workspace, used on `side_stream` */
} buctd_basic_block_grads;
int buctd_basic_block_fwd_train(const buctd_basic_block* b, void* stream);
int buctd_basic_block_bwd(const buctd_basic_block* b, const buctd_basic_block_grads* g, void* stream, void* side_stream);
/* n chained blocks (an HRNet branch, pose_hrnet.py:165-185) behind one call per direction: blocks[k].x = blocks[k-1].y,
 * grads[k].dy = grads[k+1].dx; the caller wires the pointers.  Same launches as n single calls. */
int buctd_basic_chain_fwd_train(int n, const buctd_basic_block* blocks, void* stream);
int buctd_basic_chain_bwd(int n, const buctd_basic_block* blocks, const buctd_basic_block_grads* grads, void* stream,
                          void* side_stream);

/* The branches of a HighResolutionModule (pose_hrnet.py:177-185, 247-249): nb independent chains of n blocks each, advanced
 * together - blocks[b * n + k] / grads[b * n + k] = block k of branch b, wired per branch as for buctd_basic_chain_*.  The
 * k-th convolutions of all branches are ONE launch (buctd_conv3x3_bf16x6_group), so are their BatchNorm applies, BatchNorm
 * backwards and weight gradients (buctd_conv3x3_wgrad_bf16x6_group on `side_stream`; each branch needs its own wg_ws of
 * buctd_conv3x3_wgrad_bf16x6_group_workspace(nb, ...) bytes): 3 launches per block step forward, 8 backward, whatever nb.
 * Activations, statistics and data gradients are bit-identical to nb buctd_basic_chain_* calls.  nb <= 4. */
int buctd_basic_branches_fwd_train(int nb, int n, const buctd_basic_block* blocks, void* stream);
int buctd_basic_branches_bwd(int nb, int n, const buctd_basic_block* blocks, const buctd_basic_block_grads* grads, void* stream,
                             void* side_stream);

/* Stream `to` waits for everything enqueued on stream `from` so far (one cached event per host thread and device): the fork in
 * front of work launched on a second stream, e.g. a weight gradient beside the data-gradient chain. */
int buctd_stream_fork(void* from, void* to);

/* ------------------------------------------------------------ bf16x6 GEMM --- */
/* C = alpha * A * B (+ bias) in the bf16x6 arithmetic of the 3x3 convolutions (fp32 operands split exactly into three
 * bf16 pieces, six bf16 MFMAs per product, fp32 accumulate) for the large plain GEMMs of the path - fc_o =
 * nn.Linear(T, T) of the CoAM channel attention (self_attention.py:150-159) forward, data and weight gradient.
 * Both operands are handed over as prepared images (fragment order, 6 bytes per element):
 *   image of a logical matrix X[v][k] (v: row of C for the A operand, column of C for the B operand; k: reduction
 *   index) whose element sits at src[(v / vg) * vgs + (v % vg) * vs + (k / kg) * kgs + (k % kg) * ks]
 *   (vg or kg = 0: no grouping, plain strides vs / ks).  role 0 = A operand, 1 = B operand (they differ in padding).
 * C(m, n) = C[m * ldc + (n / Nc) * gsc + n % Nc] (Nc <= 0: no grouping); bias_axis 0: bias[n], 1: bias[m]. */
int buctd_x6_image_dims(int V, int K, int role, int* Vpad, int* Kpad);
size_t buctd_x6_image_bytes(int V, int K, int role);
int buctd_x6_image(const float* src, int V, int K, int vg, long vgs, long vs, int kg, long kgs, long ks, int role,
                   void* image, void* stream);
int buctd_x6_gemm(int M, int N, int K, const void* a_image, const void* b_image, const float* bias, int bias_axis,
                  float alpha, float* C, long ldc, int Nc, long gsc, void* stream);

/* -------------------------------------------------------------- optimizer --- */
/* torch.optim.Adam step (utils/utils.py:268-272: betas .9/.999, eps 1e-8, no weight decay, no amsgrad)
 * on flat fp32 buffers; gscale multiplies the gradient first (1/world for averaged all-reduce). */
int buctd_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                    float eps, int step, float gscale, void* stream);
/* torch.optim.SGD(lr, momentum, dampening 0, weight_decay, nesterov) on a flat arena - the 'sgd' branch of get_optimizer
 * (lib/utils/utils.py:260-267).  first_step != 0: the momentum buffer is initialised with the gradient. */
int buctd_sgd_step(float* p, const float* g, float* momentum_buf, long n, float lr, float momentum, float weight_decay,
                   int nesterov, int first_step, float gscale, void* stream);

/* Fused single-head self-attention forward (flash style, exact fp32) - nn.MultiheadAttention of the TransPose encoder
 * layer, transpose_h.py:192-197, in eval mode / without attention dropout: out[b][i] = softmax_j(scale * q_i . k_j) v_j.
 * q, k: [B][T][.] rows of stride ldqk floats (q and k may be the two halves of one packed projection: k = q + d);
 * v: [B][T][.] stride ldv; out: [B][T][d] contiguous; lse (NULL ok): [B][T] log-sum-exp of the scaled logits.
 * T % 128 == 0, d % 16 == 0, d <= 128. */
int buctd_mha_fwd_supported(int T, int d);
int buctd_mha_fwd(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                  float* out, float* lse, void* stream);
/* the same in the bf16x6 arithmetic of the convolutions (three bf16 pieces per operand, six MFMAs per product, fp32
 * accumulate: fp32 class on the bf16 matrix cores); probabilities stay in registers (S^T formulation). */
int buctd_mha_fwd_bf16x6(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                         float scale, float* out, float* lse, void* stream);
/* bf16x6 attention over keys / values split ONCE per call into the workspace (6 B per element, the kernel's LDS tile layout)
 * and streamed into LDS by DMA - same results bit for bit as buctd_mha_fwd_bf16x6 (same pieces, same MFMA order), faster:
 * no per-workgroup re-split, K/V copies overlap the products, one image per XCD L2. */
size_t buctd_mha_fwd_bf16x6_workspace(int B, int T, int d);
int buctd_mha_fwd_bf16x6_ws(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv,
                            float scale, float* out, float* lse, void* workspace, size_t workspace_bytes, void* stream);
/* The same attention for TRAINING (transpose_h.py:168-213, autograd of softmax -> dropout -> . v), fused in both
 * directions - nothing T x T in HBM: forward with in-kernel attention dropout (counter hash keyed by (seed, (b T + q) T + key),
 * the function of buctd_softmax_dropout_fwd) and the row statistic lse [B][T]; flash-style backward (one kernel template in
 * two roles: dK/dV with the keys owned, dQ with the queries owned).  Exact fp32 MFMA.  T % 128 == 0, d % 16 == 0, d <= 128. */
int buctd_mha_train_supported(int T, int d);
int buctd_mha_fwd_train(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv, float scale,
                        float p_drop, uint64_t seed, float* out, float* lse, void* stream);
size_t buctd_mha_bwd_workspace(int B, int T);
int buctd_mha_bwd(int B, int T, int d, const float* q, const float* k, const float* v, int ldqk, int ldv, const float* out,
                  const float* dout, const float* lse, float scale, float p_drop, uint64_t seed, float* dq, float* dk,
                  int lddqk, float* dv, int lddv, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------- sample pipeline --- */
/* Person crop of the per-sample pipeline (dataset/JointsDataset.py:287-294): cv2.warpAffine(img_u8, M, (w, h),
 * flags=INTER_LINEAR) restated bit for bit (OpenCV's fixed-point bilinear, BORDER_CONSTANT 0), fused with
 * transforms.ToTensor + Normalize(mean, std); written into channels [0, 3) of the NCHW network input
 * (out + b * out_batch_stride).  flip: the source is mirrored horizontally first (JointsDataset.py:244);
 * rw > 0: source pixels outside the rectangle (rx, ry, rw, rh) read as zero (NEW_AUGMENTATION, 272-285).
 * crop_u8 (NULL ok): the 8-bit crop [B][h][w][3] (meta['input_img']).  items live in device memory. */
typedef struct {
  const unsigned char* src; /* uint8 [H][W][3], device memory */
  int H, W;
  int flip;
  int rx, ry, rw, rh;
  double m[6];              /* 2x3 forward matrix (source -> crop), as utils.transforms.get_affine_transform returns it */
} buctd_warp_item;
int buctd_warp_affine_norm(const buctd_warp_item* items_device, int B, int dst_h, int dst_w, const float* mean3,
                           const float* std3, float* out, long out_batch_stride, unsigned char* crop_u8, void* stream);
/* buctd_cond_render with a destination batch stride (in floats): renders straight into channels [3, 3+Cc) of the
 * network input. */
int buctd_cond_render_into(const float* joints, int js, const float* colors, int B, int K, int Cc, int H, int W,
                           int truncate, float* cond, long cond_batch_stride, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Generative pose synthesis (dataset/pose_synthesis.py:234-817, called from JointsDataset.py:202-215): for every person
 * and joint one of the error types jitter / miss / inversion / swap / good is drawn and a key point proposed
 * accordingly.  joints, estimated [B][K][3] and near_joints [B][M][K][3] (neighbours; visibility 0 = absent) are float64
 * device arrays, area [B] float64, num_overlap [B] int; out [B][K][3].  Randomness: a counter-based generator keyed by
 * (seed, person, joint) - pass a fresh seed per batch.  tables: per-dataset constants (host struct, copied). */
typedef struct {
  double sigmas[32];
  int pair[32];                /* symmetric partner of a joint, -1 = none (kps_symmetry) */
  int jitter_cls[32], miss_cls[32], inv_cls[32], swap_cls[32];
  double jitter_p[2][3];       /* [num_valid <= 10 | else][class] */
  double miss_p[3][3];         /* [num_valid <= 5 | <= 10 | else][class] */
  double inv_p[3];
  double swap_p[2][3];         /* [crowded | else][class] */
  double out_vis;              /* third column of a synthesized joint: 1 (coco) / 0 (crowdpose) */
} buctd_synth_tables;
int buctd_synthesize_pose(const buctd_synth_tables* tables, const double* joints, const double* estimated,
                          const double* near_joints, const double* area, const int* num_overlap, int B, int K, int M,
                          unsigned long long seed, double* out, void* stream);

/* ------------------------------------------------------------------- NMS --- */
/* Greedy box NMS - replaces _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
 * float nms_overlap_thresh, int device_id) of lib/nms/gpu_nms.hpp:1-2 / nms_kernel.cu:94-143 (kernel 33-77).
 * Differences of form, not of result: `boxes` ([boxes_num][boxes_dim], x1 y1 x2 y2 first, sorted by descending score
 * like gpu_nms.pyx:27-30 does before the call) is a DEVICE pointer, so are keep_out ([boxes_num] ints) and num_out;
 * the suppression mask lives in caller-provided workspace and the greedy sweep runs on the device as well.
 * A box is suppressed when IoU (with the +1 pixel convention) with an earlier kept box is > thresh. */
size_t buctd_nms_workspace(int boxes_num);
int buctd_nms(int* keep_out, int* num_out, const float* boxes, int boxes_num, int boxes_dim, float thresh,
              void* workspace, size_t workspace_bytes, void* stream);
/* cpu_nms(dets [n][5] float32, thresh) of lib/nms/cpu_nms.pyx:20-71 as plain host C++ (suppression at IoU >= thresh);
 * order = argsort of the scores, descending, computed by the caller like the .pyx does with numpy. */
int buctd_cpu_nms(const float* dets, int n, const int* order, float thresh, int* keep_out, int* num_out);

#ifdef __cplusplus
}
#endif
#endif
