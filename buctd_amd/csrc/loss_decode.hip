// Heat-map side of the path: JointsMSELoss (reference lib/core/loss.py:23-41), arg-max decode
// (lib/core/inference.py:19-47), Gaussian target rendering (lib/dataset/JointsDataset.py:397-453),
// condition heat-map rendering (JointsDataset.py:457-543), flip-test merge (lib/core/function.py:226-236),
// and the fused Adam update (lib/utils/utils.py:268-272).  All HBM-bound; one pass per tensor.
#include "common.h"
#include "../../include/buctd_hip.h"

// --------------------------------------------------------------- joints MSE ----
// one block per (n,k) heat-map: partial = sum w^2 (p-g)^2 ; grad written in the same pass
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ w, int HW, float gcoef,
                                                          float* __restrict__ part, float* __restrict__ grad) {
  __shared__ float sm[4];
  const long row = blockIdx.x;
  const float wt = w ? w[row] : 1.f;
  const float w2 = wt * wt;
  float s = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float d = pred[row * HW + i] - gt[row * HW + i];
    s += d * d;
    if (grad) grad[row * HW + i] = gcoef * w2 * d;
  }
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[row] = s * w2;
}
__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ part, int n, float coef,
                                                        float* __restrict__ loss) {
  __shared__ float sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  // fp64 partial per thread, fp32 across the block is enough for <= 256 addends
  float f = block_sum_256((float)s, sm);
  if (threadIdx.x == 0) loss[0] = f * coef;
}
extern "C" int buctd_joints_mse(const float* pred, const float* gt, const float* w, int N, int K, int HW, float* loss,
                                float* grad, float gscale, void* workspace, size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(pred && gt && loss && N > 0 && K > 0 && HW > 0, "buctd_joints_mse: bad argument");
  const size_t need = (size_t)N * K * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_joints_mse: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // L = 0.5/(K*N*HW) sum w^2 d^2 ; dL/dp = w^2 d / (K*N*HW)
  const float denom = (float)((double)K * (double)N * (double)HW);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(N * K), dim3(256), 0, st, pred, gt, w, HW, gscale / denom,
                     (float*)workspace, grad);
  BUCTD_CHECK_LAUNCH("buctd_joints_mse(partial)");
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, N * K, 0.5f / denom, loss);
  BUCTD_CHECK_LAUNCH("buctd_joints_mse(final)");
  return BUCTD_OK;
}

// ------------------------------------------------------------ argmax decode ----
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ hm, int HW, int W,
                                                     float* __restrict__ preds, float* __restrict__ maxvals,
                                                     int32_t* __restrict__ idx, float* __restrict__ quarter) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const long row = blockIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < HW; i += 256) {
    const float v = hm[row * HW + i];
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) {
        best = sv[k];
        bi = si[k];
      }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: numpy argmax would return the first NaN; not reachable here
    const float m = best > 0.f ? 1.f : 0.f;
    preds[row * 2 + 0] = (float)(bi % W) * m;
    preds[row * 2 + 1] = floorf((float)bi / (float)W) * m;
    maxvals[row] = best;
    if (idx) idx[row] = bi;
    if (quarter) {
      // reference core/inference.py:68-77 (POST_PROCESS): a quarter pixel towards the higher neighbour, only for
      // peaks with 1 < px < W-1 and 1 < py < H-1 (a masked peak decodes to (0,0) and never qualifies)
      const int H = HW / W;
      const int px = best > 0.f ? bi % W : 0, py = best > 0.f ? bi / W : 0;
      float qx = 0.f, qy = 0.f;
      if (px > 1 && px < W - 1 && py > 1 && py < H - 1) {
        const float* h = hm + row * HW;
        const float dx = h[py * W + px + 1] - h[py * W + px - 1];
        const float dy = h[(py + 1) * W + px] - h[(py - 1) * W + px];
        qx = dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f);
        qy = dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f);
      }
      quarter[row * 2 + 0] = qx;
      quarter[row * 2 + 1] = qy;
    }
  }
}
extern "C" int buctd_argmax_decode(const float* hm, int rows, int H, int W, float* preds, float* maxvals,
                                   int32_t* idx, void* stream) {
  BUCTD_CHECK_ARG(hm && preds && maxvals && rows > 0 && H > 0 && W > 0, "buctd_argmax_decode: bad argument");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, hm, H * W, W, preds, maxvals, idx,
                     (float*)nullptr);
  BUCTD_CHECK_LAUNCH("buctd_argmax_decode");
  return BUCTD_OK;
}
extern "C" int buctd_argmax_decode_refined(const float* hm, int rows, int H, int W, float* preds, float* maxvals,
                                           int32_t* idx, float* quarter, void* stream) {
  BUCTD_CHECK_ARG(hm && preds && maxvals && quarter && rows > 0 && H > 0 && W > 0,
                  "buctd_argmax_decode_refined: bad argument");
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, hm, H * W, W, preds, maxvals, idx,
                     quarter);
  BUCTD_CHECK_LAUNCH("buctd_argmax_decode_refined");
  return BUCTD_OK;
}

// ---------------------------------------------------------- gaussian target ----
// one block per (b,k) map
__global__ __launch_bounds__(256) void gaussian_target_kernel(const float* __restrict__ joints,
                                                              const float* __restrict__ vis, int Hh, int Wh,
                                                              float stride_x, float stride_y, float sigma,
                                                              float* __restrict__ target,
                                                              float* __restrict__ weight) {
  const long row = blockIdx.x;
  const float jx = joints[row * 3 + 0], jy = joints[row * 3 + 1];
  float wgt = vis[row];
  const int tmp = (int)(sigma * 3.f);
  // int() truncates toward zero, exactly like the reference
  const int mu_x = (int)(jx / stride_x + 0.5f), mu_y = (int)(jy / stride_y + 0.5f);
  const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
  const bool outside = ulx >= Wh || uly >= Hh || brx < 0 || bry < 0;
  if (outside) wgt = 0.f;
  const bool paint = !outside && wgt > 0.5f;
  const float inv2s2 = 1.f / (2.f * sigma * sigma);
  for (int i = threadIdx.x; i < Hh * Wh; i += 256) {
    const int y = i / Wh, x = i - y * Wh;
    float v = 0.f;
    if (paint && x >= ulx && x < brx && y >= uly && y < bry) {
      const float dx = (float)(x - mu_x), dy = (float)(y - mu_y);
      v = expf(-(dx * dx + dy * dy) * inv2s2);
    }
    target[row * Hh * Wh + i] = v;
  }
  if (threadIdx.x == 0) weight[row] = wgt;
}
extern "C" int buctd_gaussian_target(const float* joints, const float* vis, int B, int K, int Hh, int Wh,
                                     float stride_x, float stride_y, float sigma, float* target, float* weight,
                                     void* stream) {
  BUCTD_CHECK_ARG(joints && vis && target && weight && B > 0 && K > 0 && Hh > 0 && Wh > 0 && sigma > 0.f,
                  "buctd_gaussian_target: bad argument");
  hipLaunchKernelGGL(gaussian_target_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, joints, vis, Hh, Wh,
                     stride_x, stride_y, sigma, target, weight);
  BUCTD_CHECK_LAUNCH("buctd_gaussian_target");
  return BUCTD_OK;
}

// ---------------------------------------------------------- condition render ----
// cv2.GaussianBlur(ksize 15, sigma 0 -> 0.3*((15-1)*0.5-1)+0.8 = 2.6), BORDER_REFLECT_101, separable.
// The impulse image is K points, so instead of a dense blur each output pixel sums the K
// (reflected) point responses: out(y,x) = sum_k col_k * gy(y, py_k) * gx(x, px_k), where g?(.,.)
// accumulates the 1-D taps that land on the impulse after reflection.
struct CondTaps {
  float t[15];
};
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i;
}
__device__ __forceinline__ float tap_response(int out_pos, int src_pos, int n, const CondTaps& tp) {
  // sum of taps d in [-7,7] with reflect101(out_pos + d) == src_pos
  float s = 0.f;
#pragma unroll
  for (int d = -7; d <= 7; ++d)
    if (reflect101(out_pos + d, n) == src_pos) s += tp.t[d + 7];
  return s;
}
__global__ __launch_bounds__(256) void cond_render_kernel(const float* __restrict__ joints, int js,
                                                          const float* __restrict__ colors, int K, int Cc, int H,
                                                          int W, CondTaps tp, float* __restrict__ raw,
                                                          float* __restrict__ maxbuf) {
  // grid: (pixel blocks, B); impulses of image b are cached in LDS
  __shared__ int s_px[64], s_py[64];
  __shared__ float s_col[64][4];
  __shared__ int s_cnt;
  __shared__ float sm[4];
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    int cnt = 0;
    for (int k = 0; k < K && cnt < 64; ++k) {
      // np.array(kpts).astype(int): truncation toward zero
      const int kx = (int)joints[((long)b * K + k) * js + 0], ky = (int)joints[((long)b * K + k) * js + 1];
      if (0 < kx && kx < W && 0 < ky && ky < H) {
        // later keypoints overwrite earlier ones at the same pixel (zero_matrix[y][x] = color)
        int slot = cnt;
        for (int q = 0; q < cnt; ++q)
          if (s_px[q] == kx - 1 && s_py[q] == ky - 1) slot = q;
        s_px[slot] = kx - 1;
        s_py[slot] = ky - 1;
        for (int c = 0; c < Cc && c < 4; ++c) s_col[slot][c] = colors ? colors[k * Cc + c] : 255.f;
        if (slot == cnt) ++cnt;
      }
    }
    s_cnt = cnt;
  }
  __syncthreads();
  const int cnt = s_cnt;
  float lmax = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int y = i / W, x = i - y * W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < cnt; ++q) {
      const int dy = s_py[q] - y, dx = s_px[q] - x;
      // reflection can only matter within 7 px of a border; fast reject otherwise
      if ((dy > 7 || dy < -7) && y >= 7 && y < H - 7) continue;
      if ((dx > 7 || dx < -7) && x >= 7 && x < W - 7) continue;
      const float gy = tap_response(y, s_py[q], H, tp);
      if (gy == 0.f) continue;
      const float gx = tap_response(x, s_px[q], W, tp);
      const float g = gy * gx;
      for (int c = 0; c < Cc && c < 4; ++c) acc[c] += s_col[q][c] * g;
    }
    for (int c = 0; c < Cc && c < 4; ++c) {
      raw[(((long)b * Cc + c) * H) * W + i] = acc[c];
      lmax = fmaxf(lmax, acc[c]);
    }
  }
  lmax = block_max_256(lmax, sm);
  // values are >= 0, so the int ordering of the float bits matches the float ordering
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(maxbuf + b), __float_as_int(lmax));
}
__global__ __launch_bounds__(256) void cond_normalise_kernel(const float* __restrict__ raw,
                                                             const float* __restrict__ maxbuf, long per_img,
                                                             int truncate, float* __restrict__ cond, long cond_stride) {
  const int b = blockIdx.y;
  const float am = maxbuf[b];
  // heatmap /= am / 255  (skipped when the image is empty)
  const float div = am / 255.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
    const float v = raw[b * per_img + i];
    const float o = am == 0.f ? v : v / div;
    cond[b * cond_stride + i] = truncate ? truncf(o) : o;
  }
}
extern "C" size_t buctd_cond_render_workspace(int B, int Cc, int H, int W) {
  return ((size_t)B * Cc * H * W + (size_t)B) * sizeof(float);
}
extern "C" int buctd_cond_render(const float* joints, int js, const float* colors, int B, int K, int Cc, int H, int W,
                                 int truncate, float* cond, void* workspace, size_t workspace_bytes, void* stream) {
  return buctd_cond_render_into(joints, js, colors, B, K, Cc, H, W, truncate, cond, (long)Cc * H * W, workspace,
                                workspace_bytes, stream);
}
extern "C" int buctd_cond_render_into(const float* joints, int js, const float* colors, int B, int K, int Cc, int H,
                                      int W, int truncate, float* cond, long cond_batch_stride, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(cond_batch_stride >= (long)Cc * H * W, "buctd_cond_render_into: batch stride smaller than one image");
  BUCTD_CHECK_ARG(joints && cond && B > 0 && K > 0 && K <= 64 && Cc >= 1 && Cc <= 4 && H > 0 && W > 0 && js >= 2,
                  "buctd_cond_render: bad argument (K<=64, 1<=Cc<=4)");
  const size_t need = buctd_cond_render_workspace(B, Cc, H, W);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_cond_render: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  // cv2.getGaussianKernel(15, sigma<=0): sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8 ; taps normalised to sum 1
  CondTaps tp;
  const double sigma = 0.3 * ((15 - 1) * 0.5 - 1.0) + 0.8;
  double sum = 0.0, tv[15];
  for (int i = 0; i < 15; ++i) {
    const double x = (double)i - 7.0;
    tv[i] = exp(-(x * x) / (2.0 * sigma * sigma));
    sum += tv[i];
  }
  for (int i = 0; i < 15; ++i) tp.t[i] = (float)(tv[i] / sum);
  hipStream_t st = (hipStream_t)stream;
  float* raw = (float*)workspace;
  float* maxbuf = raw + (size_t)B * Cc * H * W;
  hipError_t e = hipMemsetAsync(maxbuf, 0, (size_t)B * sizeof(float), st);
  if (e != hipSuccess) {
    buctd_set_error("buctd_cond_render: memset failed: %s", hipGetErrorString(e));
    return BUCTD_ELAUNCH;
  }
  dim3 grid(ceil_div(H * W, 256 * 4), B);
  hipLaunchKernelGGL(cond_render_kernel, grid, dim3(256), 0, st, joints, js, colors, K, Cc, H, W, tp, raw, maxbuf);
  BUCTD_CHECK_LAUNCH("buctd_cond_render(render)");
  const long per_img = (long)Cc * H * W;
  dim3 grid2(ceil_div(per_img, 256 * 4), B);
  hipLaunchKernelGGL(cond_normalise_kernel, grid2, dim3(256), 0, st, (const float*)raw, (const float*)maxbuf, per_img,
                     truncate, cond, cond_batch_stride);
  BUCTD_CHECK_LAUNCH("buctd_cond_render(normalise)");
  return BUCTD_OK;
}

// --------------------------------------------------------------- flip-back ----
__global__ __launch_bounds__(256) void flipback_avg_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const int32_t* __restrict__ perm, int K, int H, int W,
                                                           int shift, long total, float* __restrict__ out) {
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int x = (int)(i % W);
    long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int k = (int)(r % K);
    const long n = r / K;
    // flipped[n][k][y][x'] = b[n][perm[k]][y][W-1-x'] ; shift: out x reads flipped x-1 for x >= 1, x = 0 keeps x = 0
    const int xs = (shift && x >= 1) ? x - 1 : x;
    const float fv = b[((n * K + perm[k]) * H + y) * W + (W - 1 - xs)];
    out[i] = (a[i] + fv) * 0.5f;
  }
}
extern "C" int buctd_flipback_avg(const float* a, const float* b, const int32_t* perm, int N, int K, int H, int W,
                                  int shift, float* out, void* stream) {
  BUCTD_CHECK_ARG(a && b && perm && out && N > 0 && K > 0 && H > 0 && W > 0, "buctd_flipback_avg: bad argument");
  const long total = (long)N * K * H * W;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flipback_avg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, perm, K, H,
                     W, shift, total, out);
  BUCTD_CHECK_LAUNCH("buctd_flipback_avg");
  return BUCTD_OK;
}

// --------------------------------------------------------------------- adam ----
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale) {
  // torch.optim.Adam (no amsgrad, no weight decay):
  //   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
  const float step_size = lr / bc1;
  const long n4 = n >> 2;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * gscale;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = b1 * mm[j] + (1.f - b1) * gg[j];
      vv[j] = b2 * vv[j] + (1.f - b2) * gg[j] * gg[j];
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] -= step_size * (mm[j] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    const float gg = g[i] * gscale;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}
extern "C" int buctd_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                               float beta2, float eps, int step, float gscale, void* stream) {
  BUCTD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "buctd_adam_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, (float)bc1, (float)sqrt(bc2), gscale);
  BUCTD_CHECK_LAUNCH("buctd_adam_step");
  return BUCTD_OK;
}

// ---------------------------------------------------------------------- sgd ----
// torch.optim.SGD(lr, momentum, dampening = 0, weight_decay, nesterov) on the flat arena - the 'sgd' branch of the
// reference's get_optimizer (lib/utils/utils.py:260-267):
//   g += wd p ;  buf = g (first step) | mu buf + g ;  g = nesterov ? g + mu buf : buf ;  p -= lr g        (mu = 0: no buffer)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long n, float lr, float mu, float wd, int nesterov, int first, float gscale) {
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    float pp = p[i];
    float gg = g[i] * gscale + wd * pp;
    if (mu != 0.f) {
      const float b = first ? gg : mu * buf[i] + gg;
      buf[i] = b;
      gg = nesterov ? gg + mu * b : b;
    }
    p[i] = pp - lr * gg;
  }
}
extern "C" int buctd_sgd_step(float* p, const float* g, float* momentum_buf, long n, float lr, float momentum,
                              float weight_decay, int nesterov, int first_step, float gscale, void* stream) {
  BUCTD_CHECK_ARG(p && g && n > 0 && (momentum == 0.f || momentum_buf), "buctd_sgd_step: bad argument");
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n, lr,
                     momentum, weight_decay, nesterov, first_step, gscale);
  BUCTD_CHECK_LAUNCH("buctd_sgd_step");
  return BUCTD_OK;
}
