// Row softmax (+scale, +inverted dropout), elementwise dropout and LayerNorm for the
// CoAM attention cores (reference lib/models/self_attention.py:78-86,150-158) and the
// TransPose encoder (lib/models/transpose_h.py:168-213).
//
// Dropout masks come from a counter-based hash of (seed, element index): nothing is stored,
// the backward regenerates the same mask.  (The reference draws from torch's Philox stream;
// the streams cannot match, so train-mode parity is defined with p_drop = 0 - SURVEY 8c.)
#include "common.h"
#include "../../include/buctd_hip.h"

// ------------------------------------------------------ softmax, one block per row ----
#define SM_NPT 32  // values cached per thread: rows up to 8192 columns stay in registers
__global__ __launch_bounds__(256) void softmax_fwd_block_kernel(const float* __restrict__ s, int L, float scale,
                                                                float p_drop, uint64_t seed, float* __restrict__ p,
                                                                float* __restrict__ pd) {
  __shared__ float sm[4];
  const long row = blockIdx.x;
  const float* sr = s + row * L;
  float v[SM_NPT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    v[j] = c < L ? sr[c] * scale : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = block_max_256(mx, sm);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    v[j] = c < L ? __expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = block_sum_256(sum, sm);
  const float inv = 1.f / sum;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    if (c < L) {
      const float pv = v[j] * inv;
      p[row * L + c] = pv;
      if (p_drop > 0.f) pd[row * L + c] = pv * keep_scale(seed, (uint64_t)(row * L + c), p_drop, inv_keep);
      else if (pd != p) pd[row * L + c] = pv;
    }
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_block_kernel(const float* __restrict__ dpd,
                                                                const float* __restrict__ p, int L, float scale,
                                                                float p_drop, uint64_t seed, float* __restrict__ ds) {
  __shared__ float sm[4];
  const long row = blockIdx.x;
  float g[SM_NPT], pv[SM_NPT];
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    g[j] = 0.f;
    pv[j] = 0.f;
    if (c < L) {
      pv[j] = p[row * L + c];
      g[j] = dpd[row * L + c];
      if (p_drop > 0.f) g[j] *= keep_scale(seed, (uint64_t)(row * L + c), p_drop, inv_keep);
      dot += g[j] * pv[j];
    }
  }
  dot = block_sum_256(dot, sm);
#pragma unroll
  for (int j = 0; j < SM_NPT; ++j) {
    const int c = threadIdx.x + 256 * j;
    if (c < L) ds[row * L + c] = scale * pv[j] * (g[j] - dot);
  }
}

// ------------------------------------------------------- softmax, one wave per row ----
#define SW_NPT 8  // rows up to 512 columns
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(const float* __restrict__ s, long rows, int L,
                                                               float scale, float p_drop, uint64_t seed,
                                                               float* __restrict__ p, float* __restrict__ pd) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[SW_NPT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SW_NPT; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < L ? s[row * L + c] * scale : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < SW_NPT; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < L ? __expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
#pragma unroll
  for (int j = 0; j < SW_NPT; ++j) {
    const int c = lane + 64 * j;
    if (c < L) {
      const float pv = v[j] * inv;
      p[row * L + c] = pv;
      if (p_drop > 0.f) pd[row * L + c] = pv * keep_scale(seed, (uint64_t)(row * L + c), p_drop, inv_keep);
      else if (pd != p) pd[row * L + c] = pv;
    }
  }
}
__global__ __launch_bounds__(256) void softmax_bwd_wave_kernel(const float* __restrict__ dpd,
                                                               const float* __restrict__ p, long rows, int L,
                                                               float scale, float p_drop, uint64_t seed,
                                                               float* __restrict__ ds) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float g[SW_NPT], pv[SW_NPT];
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < SW_NPT; ++j) {
    const int c = lane + 64 * j;
    g[j] = 0.f;
    pv[j] = 0.f;
    if (c < L) {
      pv[j] = p[row * L + c];
      g[j] = dpd[row * L + c];
      if (p_drop > 0.f) g[j] *= keep_scale(seed, (uint64_t)(row * L + c), p_drop, inv_keep);
      dot += g[j] * pv[j];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int j = 0; j < SW_NPT; ++j) {
    const int c = lane + 64 * j;
    if (c < L) ds[row * L + c] = scale * pv[j] * (g[j] - dot);
  }
}

extern "C" int buctd_softmax_dropout_fwd(const float* s, long rows, int L, float scale, float p_drop, uint64_t seed,
                                         float* p, float* pd, void* stream) {
  BUCTD_CHECK_ARG(s && p && pd && rows > 0 && L > 0, "buctd_softmax_dropout_fwd: bad argument");
  BUCTD_CHECK_ARG(L <= 256 * SM_NPT, "buctd_softmax_dropout_fwd: row length %d > %d unsupported", L, 256 * SM_NPT);
  BUCTD_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "buctd_softmax_dropout_fwd: p_drop out of range");
  BUCTD_CHECK_ARG(p_drop == 0.f || p != pd, "buctd_softmax_dropout_fwd: p and pd must differ when dropping");
  hipStream_t st = (hipStream_t)stream;
  if (L <= 64 * SW_NPT)
    hipLaunchKernelGGL(softmax_fwd_wave_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, s, rows, L, scale, p_drop,
                       seed, p, pd);
  else
    hipLaunchKernelGGL(softmax_fwd_block_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, L, scale, p_drop, seed, p,
                       pd);
  BUCTD_CHECK_LAUNCH("buctd_softmax_dropout_fwd");
  return BUCTD_OK;
}
extern "C" int buctd_softmax_dropout_bwd(const float* dpd, const float* p, long rows, int L, float scale,
                                         float p_drop, uint64_t seed, float* ds, void* stream) {
  BUCTD_CHECK_ARG(dpd && p && ds && rows > 0 && L > 0, "buctd_softmax_dropout_bwd: bad argument");
  BUCTD_CHECK_ARG(L <= 256 * SM_NPT, "buctd_softmax_dropout_bwd: row length %d > %d unsupported", L, 256 * SM_NPT);
  hipStream_t st = (hipStream_t)stream;
  if (L <= 64 * SW_NPT)
    hipLaunchKernelGGL(softmax_bwd_wave_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, dpd, p, rows, L, scale,
                       p_drop, seed, ds);
  else
    hipLaunchKernelGGL(softmax_bwd_block_kernel, dim3((unsigned)rows), dim3(256), 0, st, dpd, p, L, scale, p_drop, seed,
                       ds);
  BUCTD_CHECK_LAUNCH("buctd_softmax_dropout_bwd");
  return BUCTD_OK;
}

// ------------------------------------------------------------------- dropout ----
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                      float p_drop, uint64_t seed) {
  const float inv_keep = 1.f / (1.f - p_drop);
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step)
    y[i] = x[i] * keep_scale(seed, (uint64_t)i, p_drop, inv_keep);
}
extern "C" int buctd_dropout(const float* x, float* y, long n, float p_drop, uint64_t seed, void* stream) {
  BUCTD_CHECK_ARG(x && y && n > 0 && p_drop >= 0.f && p_drop < 1.f, "buctd_dropout: bad argument");
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, y, n, p_drop, seed);
  BUCTD_CHECK_LAUNCH("buctd_dropout");
  return BUCTD_OK;
}

// ----------------------------------------------------------------- layernorm ----
#define LN_NPT 8  // one wave per row, C <= 512
// x2 (NULL ok): the normalised value is x + x2 - the post-norm residual of the encoder layer, transpose_h.py:204-209, without
// a separate addition pass; sum_out (NULL ok) receives x + x2 (what the backward pass normalises again)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, long rows, int C,
                                                            float eps, float* __restrict__ sum_out, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ invstd) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[LN_NPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_NPT; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < C ? x[row * C + c] : 0.f;
    if (x2 && c < C) {
      v[j] += x2[row * C + c];
      if (sum_out) sum_out[row * C + c] = v[j];
    }
    s += v[j];
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_NPT; ++j) {
    const int c = lane + 64 * j;
    const float d = c < C ? v[j] - mu : 0.f;
    q += d * d;
  }
  const float is = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int j = 0; j < LN_NPT; ++j) {
    const int c = lane + 64 * j;
    if (c < C) y[row * C + c] = (v[j] - mu) * is * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    invstd[row] = is;
  }
}
// dx = invstd * (gy - mean(gy) - xhat*mean(gy*xhat)), gy = dy*gamma ; partial dgamma/dbeta per 4-row block
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, long rows, int C,
                                                            float* __restrict__ dx) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float mu = mean[row], is = invstd[row];
  float gy[LN_NPT], xh[LN_NPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < LN_NPT; ++j) {
    const int c = lane + 64 * j;
    gy[j] = 0.f;
    xh[j] = 0.f;
    if (c < C) {
      gy[j] = dy[row * C + c] * gamma[c];
      xh[j] = (x[row * C + c] - mu) * is;
      s1 += gy[j];
      s2 += gy[j] * xh[j];
    }
  }
  s1 = wave_sum(s1) / (float)C;
  s2 = wave_sum(s2) / (float)C;
#pragma unroll
  for (int j = 0; j < LN_NPT; ++j) {
    const int c = lane + 64 * j;
    if (c < C) dx[row * C + c] = is * (gy[j] - s1 - xh[j] * s2);
  }
}
// column partials of dy*xhat and dy over row chunks
#define LN_ROWS 64
__global__ __launch_bounds__(256) void layernorm_param_partial_kernel(const float* __restrict__ dy,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, long rows,
                                                                      int C, float* __restrict__ part) {
  const long r0 = (long)blockIdx.x * LN_ROWS;
  long r1 = r0 + LN_ROWS;
  if (r1 > rows) r1 = rows;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg = 0.f, sb = 0.f;
    for (long r = r0; r < r1; ++r) {
      const float d = dy[r * C + c];
      sg += d * (x[r * C + c] - mean[r]) * invstd[r];
      sb += d;
    }
    part[((long)blockIdx.x * 2 + 0) * C + c] = sg;
    part[((long)blockIdx.x * 2 + 1) * C + c] = sb;
  }
}
__global__ __launch_bounds__(256) void layernorm_param_final_kernel(const float* __restrict__ part, int nchunks, int C,
                                                                    float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double sg = 0.0, sb = 0.0;
  for (int k = 0; k < nchunks; ++k) {
    sg += (double)part[((long)k * 2 + 0) * C + c];
    sb += (double)part[((long)k * 2 + 1) * C + c];
  }
  dgamma[c] = accumulate ? dgamma[c] + (float)sg : (float)sg;
  dbeta[c] = accumulate ? dbeta[c] + (float)sb : (float)sb;
}

extern "C" int buctd_layernorm_fwd(const float* x, const float* gamma, const float* beta, long rows, int C, float eps,
                                   float* y, float* mean, float* invstd, void* stream) {
  BUCTD_CHECK_ARG(x && gamma && beta && y && mean && invstd && rows > 0 && C > 0, "buctd_layernorm_fwd: bad argument");
  BUCTD_CHECK_ARG(C <= 64 * LN_NPT, "buctd_layernorm_fwd: C %d > %d unsupported", C, 64 * LN_NPT);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (const float*)nullptr, gamma, beta, rows, C, eps, (float*)nullptr, y, mean, invstd);
  BUCTD_CHECK_LAUNCH("buctd_layernorm_fwd");
  return BUCTD_OK;
}
extern "C" int buctd_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, long rows, int C,
                                       float eps, float* sum_out, float* y, float* mean, float* invstd, void* stream) {
  BUCTD_CHECK_ARG(a && b && gamma && beta && y && mean && invstd && rows > 0 && C > 0, "buctd_add_layernorm_fwd: bad argument");
  BUCTD_CHECK_ARG(C <= 64 * LN_NPT, "buctd_add_layernorm_fwd: C %d > %d unsupported", C, 64 * LN_NPT);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b, gamma, beta,
                     rows, C, eps, sum_out, y, mean, invstd);
  BUCTD_CHECK_LAUNCH("buctd_add_layernorm_fwd");
  return BUCTD_OK;
}
extern "C" size_t buctd_layernorm_bwd_workspace(long rows, int C) {
  return (size_t)((rows + LN_ROWS - 1) / LN_ROWS) * 2 * C * sizeof(float);
}
extern "C" int buctd_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* invstd,
                                   const float* gamma, long rows, int C, float* dx, float* dgamma, float* dbeta,
                                   int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && rows > 0 && C > 0,
                  "buctd_layernorm_bwd: bad argument");
  BUCTD_CHECK_ARG(C <= 64 * LN_NPT, "buctd_layernorm_bwd: C %d > %d unsupported", C, 64 * LN_NPT);
  const size_t need = buctd_layernorm_bwd_workspace(rows, C);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_layernorm_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, dy, x, mean, invstd, gamma, rows,
                     C, dx);
  BUCTD_CHECK_LAUNCH("buctd_layernorm_bwd(dx)");
  const int nchunks = ceil_div(rows, LN_ROWS);
  hipLaunchKernelGGL(layernorm_param_partial_kernel, dim3(nchunks), dim3(256), 0, st, dy, x, mean, invstd, rows, C,
                     (float*)workspace);
  BUCTD_CHECK_LAUNCH("buctd_layernorm_bwd(partial)");
  hipLaunchKernelGGL(layernorm_param_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, (const float*)workspace,
                     nchunks, C, dgamma, dbeta, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_layernorm_bwd(final)");
  return BUCTD_OK;
}
