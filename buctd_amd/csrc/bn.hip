// Train / eval BatchNorm2d pieces for NHWC tensors viewed as [rows][C]
// (reference: nn.BatchNorm2d(momentum=0.1, eps=1e-5) in lib/models/pose_hrnet.py:37-56).
//
// Forward statistics come either from the conv epilogue (conv.hip) or from
// bn_stats_kernel as Welford partials (mean, M2) per (row group, channel);
// bn_finalize_kernel merges them in fp64 with Chan's formula - one wave64 per
// channel, shuffle tree - so the batch variance never suffers the E[x^2]-E[x]^2
// cancellation.  All streaming kernels are HBM-bound: float4 per lane, rows
// tiled so that a wave reads whole 64B+ segments.
#include "common.h"
#include "../../include/buctd_hip.h"
#include "bn_acc.h"

#define STAT_ROWS 64  // rows per Welford group for bn_stats_kernel

// ---------------------------------------------------------------- stats ----
// grid.x = row groups of STAT_ROWS rows; thread t handles channel c = t (looping by 256)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, long rows, int C,
                                                       float* __restrict__ part) {
  const long r0 = (long)blockIdx.x * STAT_ROWS;
  long r1 = r0 + STAT_ROWS;
  if (r1 > rows) r1 = rows;
  const int cnt = (int)(r1 - r0);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (long r = r0; r < r1; ++r) s += z[r * C + c];
    const float mean = cnt > 0 ? s / (float)cnt : 0.f;
    float m2 = 0.f;
    for (long r = r0; r < r1; ++r) {
      const float d = z[r * C + c] - mean;
      m2 += d * d;
    }
    part[((long)blockIdx.x * C + c) * 2 + 0] = mean;
    part[((long)blockIdx.x * C + c) * 2 + 1] = m2;
  }
}

// The same partials for tensors with at most 4 channels (the 3-channel outputs of the preNet at full resolution): the kernel
// above keeps one THREAD per channel busy - 3 of 256 - and took 225 us for the 42 MB of a 384x288 batch.  Here a wavefront
// owns a group: lane = row, C values per lane, sums over the 64 lanes by shuffles (two passes: mean, then squared distances).
template <int C>
__global__ __launch_bounds__(256) void bn_stats_thin_kernel(const float* __restrict__ z, long rows, float* __restrict__ part) {
  const long grp = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long r0 = grp * STAT_ROWS;
  if (r0 >= rows) return;
  const long left = rows - r0;
  const int cnt = left < STAT_ROWS ? (int)left : STAT_ROWS;
  float v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = lane < cnt ? z[(r0 + lane) * C + c] : 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float mean = wave_sum(v[c]) / (float)cnt;
    const float d = lane < cnt ? v[c] - mean : 0.f;
    const float m2 = wave_sum(d * d);
    if (lane == 0) {
      part[(grp * C + c) * 2 + 0] = mean;
      part[(grp * C + c) * 2 + 1] = m2;
    }
  }
}

struct Wf {
  double n, mean, m2;
};
__device__ __forceinline__ Wf wf_merge(Wf a, Wf b) {
  const double n = a.n + b.n;
  if (n == 0.0) return a;
  const double d = b.mean - a.mean;
  Wf r;
  r.n = n;
  r.mean = a.mean + d * (b.n / n);
  r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / n);
  return r;
}
__device__ __forceinline__ double shfl_xor_d(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return __hiloint2double(hi, lo);
}

// one 256-thread workgroup per channel; fp64 sums of the (L2-resident) partials:
//   mean = sum n_g mean_g / N ;  M2 = sum [ M2_g + n_g (mean_g - mean)^2 ]      (Chan et al., exact)
// the two sums of a finalize share ONE cross-wave exchange (two barriers instead of four): these kernels are pure latency -
// one workgroup per channel between two convolutions that wait for it
__device__ __forceinline__ void block_sum_d2(double& a, double& b, double (*sm)[2]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += shfl_xor_d(a, o);
    b += shfl_xor_d(b, o);
  }
  if ((threadIdx.x & 63) == 0) {
    sm[threadIdx.x >> 6][0] = a;
    sm[threadIdx.x >> 6][1] = b;
  }
  __syncthreads();
  a = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
  b = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part,
                                                          const int* __restrict__ counts, int ngroups, int rpg,
                                                          long rows, int C, float eps, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ rmean, float* __restrict__ rvar) {
  // One workgroup per channel, ONE pass: S1 = sum n_g mean_g and S2 = sum (M2_g + n_g mean_g^2) in fp64.  The products
  // of fp32 inputs are exact in fp64, so var = (S2 - S1^2/N)/N keeps ~16 - 2 log10(|mean|/std) digits - far beyond the
  // fp32 result for any activation statistics (|mean|/std < 1e4) - while the partials are gathered once, EIGHT
  // independent 8-byte loads (+ their counts) in flight per thread: up to 2048 groups cost one memory round trip.
  __shared__ double sm[4][2];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const float2* pp = reinterpret_cast<const float2*>(part) + c;
  for (int g0 = threadIdx.x; g0 < ngroups; g0 += 2048) {
    float2 v[8];
    int n[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = g0 + 256 * u;
      const bool ok = g < ngroups;
      const int gc = ok ? g : 0;
      v[u] = pp[(long)gc * C];
      long cnt = rows - (long)gc * rpg;
      cnt = cnt < 0 ? 0 : (cnt > rpg ? rpg : cnt);
      n[u] = ok ? (counts ? counts[gc] : (int)cnt) : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double m = (double)v[u].x, nn = (double)n[u];
      s1 += nn * m;
      if (n[u] > 0) s2 += (double)v[u].y + nn * m * m;
    }
  }
  block_sum_d2(s1, s2, sm);
  if (threadIdx.x == 0) {
    const double mu = s1 / (double)rows;
    double m2 = s2 - s1 * mu;
    if (m2 < 0.0) m2 = 0.0;
    const double var = m2 / (double)rows;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      const double unb = rows > 1 ? m2 / (double)(rows - 1) : var;
      rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
      rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
    }
  }
}

// ---------------------------------------------------------------- apply ----
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       int relu, float* __restrict__ y, long total, int C) {
  if (VEC) {
    const long n4 = total >> 2;
    const long step = (long)gridDim.x * 256;
    const int dc = (int)((step * 4) % C);
    int c = (int)((((long)blockIdx.x * 256 + threadIdx.x) * 4) % C) - dc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
      c += dc;
      if (c >= C) c -= C;
      const f32x4 v = reinterpret_cast<const f32x4*>(z)[i];
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = invstd[c + j] * gamma[c + j];
        o[j] = (v[j] - mean[c + j]) * sc + beta[c + j];
      }
      if (res) {
        const f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
        o += r;
      }
      if (relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
      }
      reinterpret_cast<f32x4*>(y)[i] = o;
    }
  } else {
    const long step = (long)gridDim.x * 256;
    const int dc = (int)(step % C);
    int c = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C) - dc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
      c += dc;
      if (c >= C) c -= C;
      float o = (z[i] - mean[c]) * (invstd[c] * gamma[c]) + beta[c];
      if (res) o += res[i];
      if (relu) o = fmaxf(o, 0.f);
      y[i] = o;
    }
  }
}

// -------------------------------------------------------------- backward ----
// pass 1: per row-chunk partial sums  s1 = sum g,  s2 = sum g * zhat   (g = dy * relu-mask)
// block = 256 threads laid out as (rows 256/CT) x (CT column threads), CT = min(C,64) rounded;
// simple layout: thread handles channel c = t % CW, row lane = t / CW.
#define BWD_ROWS 64
// Vector path (C % 4 == 0, C/4 <= 256): thread = (row lane, 4-channel group); 16-byte loads of dy, y, z.
__global__ __launch_bounds__(256) void bn_bwd_reduce_vec_kernel(const float* __restrict__ dy,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ z,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int relu, long rows,
                                                                int C, float* __restrict__ part) {
  __shared__ f32x4 sm[2][256];
  const int c4n = C >> 2;
  const int rl = 256 / c4n;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const long r0 = (long)blockIdx.x * BWD_ROWS;
  long r1 = r0 + BWD_ROWS;
  if (r1 > rows) r1 = rows;
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (tr < rl) {
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[tc];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[tc];
    f32x4 sc = is, be = is;
    if (relu && !y) {   // mask recomputed exactly as bn_apply formed its output (no residual in the forward)
      sc = is * reinterpret_cast<const f32x4*>(gamma)[tc];
      be = reinterpret_cast<const f32x4*>(beta)[tc];
    }
    for (long r = r0 + tr; r < r1; r += rl) {
      const long o = r * c4n + tc;
      f32x4 g = reinterpret_cast<const f32x4*>(dy)[o];
      const f32x4 zz = reinterpret_cast<const f32x4*>(z)[o];
      if (relu) {
        f32x4 yy;
        if (y) yy = reinterpret_cast<const f32x4*>(y)[o];
        else yy = (zz - mu) * sc + be;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yy[j] > 0.f)) g[j] = 0.f;
      }
      s1 += g;
      s2 += g * (zz - mu) * is;
    }
  }
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rl; ++k) {
      s1 += sm[0][k * c4n + tc];
      s2 += sm[1][k * c4n + tc];
    }
    reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 0) * C)[tc] = s1;
    reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 1) * C)[tc] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ z,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu, long rows,
                                                            int C, float* __restrict__ part) {
  __shared__ float sm[2][256];
  const int cw = C < 256 ? C : 256;        // channels handled per sweep
  const int rl = 256 / cw;                  // row lanes
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const long r0 = (long)blockIdx.x * BWD_ROWS;
  long r1 = r0 + BWD_ROWS;
  if (r1 > rows) r1 = rows;
  for (int cb = 0; cb < C; cb += cw) {
    const int c = cb + tc;
    float s1 = 0.f, s2 = 0.f;
    if (c < C && tr < rl) {
      const float mu = mean[c], is = invstd[c];
      const float sc = (relu && !y) ? is * gamma[c] : 0.f, be = (relu && !y) ? beta[c] : 0.f;
      for (long r = r0 + tr; r < r1; r += rl) {
        float g = dy[r * C + c];
        const float zz = z[r * C + c];
        if (relu) {
          const float yy = y ? y[r * C + c] : (zz - mu) * sc + be;
          if (!(yy > 0.f)) g = 0.f;
        }
        s1 += g;
        s2 += g * (zz - mu) * is;
      }
    }
    sm[0][threadIdx.x] = s1;
    sm[1][threadIdx.x] = s2;
    __syncthreads();
    if (tr == 0 && c < C) {
      for (int k = 1; k < rl; ++k) {
        s1 += sm[0][k * cw + tc];
        s2 += sm[1][k * cw + tc];
      }
      part[((long)blockIdx.x * 2 + 0) * C + c] = s1;
      part[((long)blockIdx.x * 2 + 1) * C + c] = s2;
    }
    __syncthreads();
  }
}

// pass 2: one workgroup per channel sums the chunk partials (fp64, four loads in flight), writes s[2][C] and
// dgamma/dbeta
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nchunks, int C,
                                                              float* __restrict__ s, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  __shared__ double sm[4][2];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k0 = threadIdx.x; k0 < nchunks; k0 += 2048) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 256 * u;
      const bool ok = k < nchunks;
      const long kc = ok ? k : 0;
      a[u] = part[(kc * 2 + 0) * C + c];
      b[u] = part[(kc * 2 + 1) * C + c];
      if (!ok) { a[u] = 0.f; b[u] = 0.f; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += (double)a[u];
      s2 += (double)b[u];
    }
  }
  block_sum_d2(s1, s2, sm);
  if (threadIdx.x == 0) {
    s[c] = (float)s1;
    s[C + c] = (float)s2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
  }
}

// pass 3: dz = gamma*invstd*(g - s1/M - zhat*s2/M);  dres = g
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ s, int relu, long total, int C,
                                                           float inv_rows, float* __restrict__ dz,
                                                           float* __restrict__ dres) {
  if (VEC) {
    const long n4 = total >> 2;
    const long step = (long)gridDim.x * 256;
    const int dc = (int)((step * 4) % C);
    int c = (int)((((long)blockIdx.x * 256 + threadIdx.x) * 4) % C) - dc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
      c += dc;
      if (c >= C) c -= C;
      f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
      const f32x4 zz = reinterpret_cast<const f32x4*>(z)[i];
      if (relu) {
        f32x4 yy;
        if (y) yy = reinterpret_cast<const f32x4*>(y)[i];
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float sc = invstd[c + j] * gamma[c + j];
            yy[j] = (zz[j] - mean[c + j]) * sc + beta[c + j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yy[j] > 0.f)) g[j] = 0.f;
      }
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float is = invstd[c + j];
        const float zh = (zz[j] - mean[c + j]) * is;
        o[j] = gamma[c + j] * is * (g[j] - s[c + j] * inv_rows - zh * s[C + c + j] * inv_rows);
      }
      reinterpret_cast<f32x4*>(dz)[i] = o;
      if (dres) reinterpret_cast<f32x4*>(dres)[i] = g;
    }
  } else {
    const long step = (long)gridDim.x * 256;
    const int dc = (int)(step % C);
    int c = (int)(((long)blockIdx.x * 256 + threadIdx.x) % C) - dc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
      c += dc;
      if (c >= C) c -= C;
      float g = dy[i];
      if (relu) {
        const float yy = y ? y[i] : (z[i] - mean[c]) * (invstd[c] * gamma[c]) + beta[c];
        if (!(yy > 0.f)) g = 0.f;
      }
      const float is = invstd[c];
      const float zh = (z[i] - mean[c]) * is;
      dz[i] = gamma[c] * is * (g - s[c] * inv_rows - zh * s[C + c] * inv_rows);
      if (dres) dres[i] = g;
    }
  }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               int C, float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

// ------------------------------------------------------------------ host ----
static int stream_grid(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int buctd_bn_stats_groups(long rows, int C, int* ngroups, int* rows_per_group) {
  BUCTD_CHECK_ARG(rows > 0 && C > 0 && ngroups && rows_per_group, "buctd_bn_stats_groups: bad argument");
  *rows_per_group = STAT_ROWS;
  *ngroups = ceil_div(rows, STAT_ROWS);
  return BUCTD_OK;
}

extern "C" int buctd_bn_stats(const float* z, long rows, int C, float* partials, int* ngroups, int* rows_per_group,
                              void* stream) {
  BUCTD_CHECK_ARG(z && partials && rows > 0 && C > 0, "buctd_bn_stats: bad argument");
  const int ng = ceil_div(rows, STAT_ROWS);
  if (ngroups) *ngroups = ng;
  if (rows_per_group) *rows_per_group = STAT_ROWS;
  const dim3 tgrid(ceil_div(ng, 4));
  if (C == 1) hipLaunchKernelGGL(bn_stats_thin_kernel<1>, tgrid, dim3(256), 0, (hipStream_t)stream, z, rows, partials);
  else if (C == 2) hipLaunchKernelGGL(bn_stats_thin_kernel<2>, tgrid, dim3(256), 0, (hipStream_t)stream, z, rows, partials);
  else if (C == 3) hipLaunchKernelGGL(bn_stats_thin_kernel<3>, tgrid, dim3(256), 0, (hipStream_t)stream, z, rows, partials);
  else if (C == 4) hipLaunchKernelGGL(bn_stats_thin_kernel<4>, tgrid, dim3(256), 0, (hipStream_t)stream, z, rows, partials);
  else hipLaunchKernelGGL(bn_stats_kernel, dim3(ng), dim3(256), 0, (hipStream_t)stream, z, rows, C, partials);
  BUCTD_CHECK_LAUNCH("buctd_bn_stats");
  return BUCTD_OK;
}

extern "C" int buctd_bn_finalize(const float* partials, const int* group_counts, int ngroups, int rows_per_group,
                                 long rows, int C, float eps,
                                 float momentum, float* mean, float* invstd, float* running_mean,
                                 float* running_var, void* stream) {
  BUCTD_CHECK_ARG(partials && mean && invstd && ngroups > 0 && rows_per_group > 0 && rows > 0 && C > 0,
                  "buctd_bn_finalize: bad argument");
  BUCTD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr),
                  "buctd_bn_finalize: running_mean/var go together");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, group_counts, ngroups,
                     rows_per_group, rows, C, eps, momentum, mean, invstd, running_mean, running_var);
  BUCTD_CHECK_LAUNCH("buctd_bn_finalize");
  return BUCTD_OK;
}

extern "C" int buctd_bn_apply(const float* z, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* residual, int relu, float* y, long rows, int C,
                              void* stream) {
  BUCTD_CHECK_ARG(z && mean && invstd && gamma && beta && y && rows > 0 && C > 0, "buctd_bn_apply: bad argument");
  const long total = rows * C;
  if (C % 4 == 0)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(stream_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, z, mean,
                       invstd, gamma, beta, residual, relu, y, total, C);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, z, mean,
                       invstd, gamma, beta, residual, relu, y, total, C);
  BUCTD_CHECK_LAUNCH("buctd_bn_apply");
  return BUCTD_OK;
}

__global__ void bn_bwd_reduce2_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int relu, long rows,
                                      int C, long rows_per_block, float* __restrict__ part);
static long bwd2_blocks(long rows, long* rows_per_block);

extern "C" size_t buctd_bn_bwd_workspace(long rows, int C) {
  const long nchunks = (rows + BWD_ROWS - 1) / BWD_ROWS;
  return (size_t)(nchunks * 2 * C + 2 * C) * sizeof(float);
}

extern "C" int buctd_bn_bwd(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int relu, long rows, int C, float* dz, float* dres,
                            float* dgamma, float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                            void* stream) {
  BUCTD_CHECK_ARG(dy && z && mean && invstd && gamma && dz && rows > 0 && C > 0, "buctd_bn_bwd: bad argument");
  BUCTD_CHECK_ARG(!relu || y || beta, "buctd_bn_bwd: relu backward needs the forward output, or beta to rebuild its sign");
  const size_t need = buctd_bn_bwd_workspace(rows, C);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_bn_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int nchunks = ceil_div(rows, BWD_ROWS);
  float* part = (float*)workspace;
  float* s = part + (long)nchunks * 2 * C;
  if (C % 4 == 0 && C / 4 <= 256) {
    // the fixed-grid reduction (see bn_bwd_reduce2_kernel below): at most 1024 partials, never more than rows / 64
    long rpb;
    const long nb = bwd2_blocks(rows, &rpb);
    hipLaunchKernelGGL(bn_bwd_reduce2_kernel, dim3((unsigned)nb), dim3(256), 0, st, dy, y, z, mean, invstd, gamma, beta, relu,
                       rows, C, rpb, part);
    nchunks = (int)nb;
  } else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nchunks), dim3(256), 0, st, dy, y, z, mean, invstd, gamma, beta, relu, rows,
                       C, part);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd(reduce)");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, (const float*)part, nchunks, C, s,
                     dgamma, dbeta, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd(finalize)");
  const long total = rows * C;
  const float inv_rows = 1.0f / (float)rows;
  if (C % 4 == 0)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(stream_grid(total / 4)), dim3(256), 0, st, dy, y, z, mean,
                       invstd, gamma, beta, (const float*)s, relu, total, C, inv_rows, dz, dres);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, st, dy, y, z, mean, invstd,
                       gamma, beta, (const float*)s, relu, total, C, inv_rows, dz, dres);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd(apply)");
  return BUCTD_OK;
}

/* The rest of buctd_bn_bwd when the reduction pass already happened elsewhere (buctd_conv3x3_bf16x6_bnstat: the data
 * gradient that produced dy formed the partial sums on its way out): part = [nparts][2][C] partial (s1, s2) sums.
 * workspace: 2 * C floats.  Finalize (fp64 merge, dgamma / dbeta) + apply (dz, optional dres), as buctd_bn_bwd. */
extern "C" int buctd_bn_bwd_from_partials(const float* dy, const float* y, const float* z, const float* mean,
                                          const float* invstd, const float* gamma, const float* beta, int relu, long rows,
                                          int C, const float* part, int nparts, float* dz, float* dres, float* dgamma,
                                          float* dbeta, int accumulate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  BUCTD_CHECK_ARG(dy && z && mean && invstd && gamma && dz && part && nparts > 0 && rows > 0 && C > 0,
                  "buctd_bn_bwd_from_partials: bad argument");
  BUCTD_CHECK_ARG(!relu || y || beta, "buctd_bn_bwd_from_partials: relu backward needs the forward output, or beta to rebuild its sign");
  if (!workspace || workspace_bytes < (size_t)2 * C * sizeof(float)) {
    buctd_set_error("buctd_bn_bwd_from_partials: workspace %zu bytes < required %zu", workspace_bytes, (size_t)2 * C * sizeof(float));
    return BUCTD_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* s = (float*)workspace;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, part, nparts, C, s, dgamma, dbeta, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd_from_partials(finalize)");
  const long total = rows * C;
  const float inv_rows = 1.0f / (float)rows;
  if (C % 4 == 0)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(stream_grid(total / 4)), dim3(256), 0, st, dy, y, z, mean,
                       invstd, gamma, beta, (const float*)s, relu, total, C, inv_rows, dz, dres);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, st, dy, y, z, mean, invstd,
                       gamma, beta, (const float*)s, relu, total, C, inv_rows, dz, dres);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd_from_partials(apply)");
  return BUCTD_OK;
}

extern "C" int buctd_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
  BUCTD_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0,
                  "buctd_bn_fold: bad argument");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  BUCTD_CHECK_LAUNCH("buctd_bn_fold");
  return BUCTD_OK;
}

// =====================================================================================================================
// BatchNorm backward, second generation:
//   * bn_bwd_reduce2_kernel: a FIXED grid of <= 1024 workgroups, each walking a contiguous range of rows with four
//     independent 16-byte loads per tensor in flight (the first generation launched one workgroup per 64 rows - 3456 tiny
//     workgroups for the 96x72 maps, three dependent-latency round trips each: 1.4-2.2 TB/s); the finalize then folds
//     <= 1024 partials per channel instead of rows / 64;
// Arithmetic of every element is the expression of the first-generation kernels, in the same order.

#define BWD2_MAX_BLOCKS 1024

__global__ __launch_bounds__(256) void bn_bwd_reduce2_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int relu, long rows, int C, long rows_per_block,
                                                             float* __restrict__ part) {
  __shared__ f32x4 sm[2][256];
  const int c4n = C >> 2;                 // C % 4 == 0, C / 4 <= 256
  const int rl = 256 / c4n;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (tr < rl) {
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[tc];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[tc];
    f32x4 sc = is, be = is;
    const bool rebuild = relu && !y;      // mask recomputed exactly as bn_apply formed its output
    if (rebuild) {
      sc = is * reinterpret_cast<const f32x4*>(gamma)[tc];
      be = reinterpret_cast<const f32x4*>(beta)[tc];
    }
    auto body = [&](f32x4 g, f32x4 zz, f32x4 yy) {
      if (relu) {
        if (rebuild) yy = (zz - mu) * sc + be;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yy[j] > 0.f)) g[j] = 0.f;
      }
      s1 += g;
      s2 += g * (zz - mu) * is;
    };
    long r = r0 + tr;
    const long step = rl;
    for (; r + 3 * step < r1; r += 4 * step) {
      f32x4 g[4], zz[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long o = (r + u * step) * c4n + tc;
        g[u] = reinterpret_cast<const f32x4*>(dy)[o];
        zz[u] = reinterpret_cast<const f32x4*>(z)[o];
        yy[u] = (relu && y) ? reinterpret_cast<const f32x4*>(y)[o] : zz[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) body(g[u], zz[u], yy[u]);
    }
    for (; r < r1; r += step) {
      const long o = r * c4n + tc;
      const f32x4 zz = reinterpret_cast<const f32x4*>(z)[o];
      body(reinterpret_cast<const f32x4*>(dy)[o], zz, (relu && y) ? reinterpret_cast<const f32x4*>(y)[o] : zz);
    }
  }
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rl; ++k) {
      s1 += sm[0][k * c4n + tc];
      s2 += sm[1][k * c4n + tc];
    }
    reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 0) * C)[tc] = s1;
    reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 2 + 1) * C)[tc] = s2;
  }
}

static long bwd2_blocks(long rows, long* rows_per_block) {
  long rpb = (rows + BWD2_MAX_BLOCKS - 1) / BWD2_MAX_BLOCKS;
  if (rpb < 64) rpb = 64;      // never more partials than the first generation's rows / 64 (its workspace size)
  *rows_per_block = rpb;
  return (rows + rpb - 1) / rpb;
}

// =====================================================================================================================
// BatchNorm without finalize launches (bn_acc.h): the statistics arrive as exact integer accumulators filled by the
// producing kernel; the streaming kernels below decode them in their prologue into an LDS table - every workgroup for
// itself (C x 256 B from L2), workgroup 0 also leaves them in the form later kernels read (mean / invstd for the backward
// pass, running statistics, dgamma / dbeta).  The per-element arithmetic is that of bn_apply_kernel / bn_bwd_apply_kernel.

// grid: every workgroup pays the prologue (32 C bytes of accumulator words from L2 per 256 threads, one round trip, a few fp64
// operations, two barriers), so at least 4 float4 per thread - and at most ONE round of 8 resident workgroups per CU
// (<= 64 registers), whose prologues overlap each other's streaming
static int acc_grid(long n4, int per_cu = 8) {
  long b = (n4 + 256 * 4 - 1) / (256 * 4);
  if (b > 256 * per_cu) b = 256 * per_cu;
  if (b < 1) b = 1;
  return (int)b;
}

// y = act((z - mean) * (invstd * gamma) + beta (+ residual)); LDS: [4][C] accumulator words | [3][C] floats
// One tensor = one BnApplyArgs; workgroup `bid` of `nblk` (a whole launch, or this tensor's share of a group launch).
struct BnApplyArgs {
  const float* z;
  BnAccFwd a;
  const float* gamma;
  const float* beta;
  const float* res;
  int relu;
  float* y;
  long total;
  int C;
};
__device__ __forceinline__ void bn_apply_acc_body(const BnApplyArgs& p, unsigned bid, unsigned nblk, unsigned char* lds_raw) {
  const float* __restrict__ z = p.z;
  const float* __restrict__ res = p.res;
  float* __restrict__ y = p.y;
  const BnAccFwd& a = p.a;
  const int C = p.C, relu = p.relu;
  long long* wsum = reinterpret_cast<long long*>(lds_raw);
  float* tab = reinterpret_cast<float*>(lds_raw + (size_t)BNACC_WORDS * C * sizeof(long long));
  const long n4 = p.total >> 2;
  const long step = (long)nblk * 256;
  // the first elements are on their way from HBM while the statistics are decoded
  long i = (long)bid * 256 + threadIdx.x;
  f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0, r0 = v0, r1 = v0;
  auto fetch = [&](long j) {
    if (j < n4) {
      v0 = reinterpret_cast<const f32x4*>(z)[j];
      if (res) r0 = reinterpret_cast<const f32x4*>(res)[j];
    }
    if (j + step < n4) {
      v1 = reinterpret_cast<const f32x4*>(z)[j + step];
      if (res) r1 = reinterpret_cast<const f32x4*>(res)[j + step];
    }
  };
  fetch(i);
  bnacc_gather_lds(a.acc, C, wsum);
  for (int c = threadIdx.x; c < C; c += 256) {
    double s1, s2;
    bnacc_read_lds(wsum, C, c, &s1, &s2);
    const BnFwdStat st = bnacc_fwd_stat_sums(s1, s2, a.rows, a.eps);
    tab[c] = st.mean;
    tab[C + c] = st.invstd * p.gamma[c];
    tab[2 * C + c] = p.beta[c];
    if (bid == 0) {
      a.mean_out[c] = st.mean;
      a.invstd_out[c] = st.invstd;
      if (a.rmean) bnacc_running(st, a.rows, a.momentum, a.rmean, a.rvar, c);
    }
  }
  __syncthreads();
  const int dc = (int)((step * 4) % C);
  int c = (int)((i * 4) % C) - dc;
  auto one = [&](long j, int cc, f32x4 v, f32x4 r) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + cc);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + C + cc);
    const f32x4 be = *reinterpret_cast<const f32x4*>(tab + 2 * C + cc);
    f32x4 o;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) o[j4] = (v[j4] - mu[j4]) * sc[j4] + be[j4];
    if (res) o += r;
    if (relu) {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) o[j4] = fmaxf(o[j4], 0.f);
    }
    reinterpret_cast<f32x4*>(y)[j] = o;
  };
  for (; i < n4; i += 2 * step) {       // two elements per trip: their loads are in flight together
    c += dc;
    if (c >= C) c -= C;
    one(i, c, v0, r0);
    c += dc;
    if (c >= C) c -= C;
    if (i + step < n4) one(i + step, c, v1, r1);
    fetch(i + 2 * step);
  }
}
__global__ __launch_bounds__(256, 8) void bn_apply_acc_kernel(BnApplyArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bn_apply_acc_body(p, blockIdx.x, gridDim.x, lds_raw);
}

// Several tensors in one launch (the branches of a HighResolutionModule): workgroups [first[k], first[k + 1]) work on tensor
// k exactly as its own launch of that many workgroups would.  The descriptor is read in place from the kernel-argument
// segment (scalar loads at a computed offset).
#define BNG_MAX 4
template <class A>
struct BnGroup {
  A s[BNG_MAX];
  int n;
  unsigned first[BNG_MAX + 1];
};
template <class A>
__device__ __forceinline__ int bn_group_find(const BnGroup<A>& g, unsigned* bid, unsigned* nblk) {
  int k = 0;
  while (k + 1 < g.n && blockIdx.x >= g.first[k + 1]) ++k;
  *bid = blockIdx.x - g.first[k];
  *nblk = g.first[k + 1] - g.first[k];
  return k;
}
__global__ __launch_bounds__(256, 8) void bn_apply_acc_group_kernel(BnGroup<BnApplyArgs> g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const BnGroup<BnApplyArgs>& g = *(const BnGroup<BnApplyArgs>*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned bid, nblk;
  const int k = bn_group_find(g, &bid, &nblk);
  bn_apply_acc_body(g.s[k], bid, nblk, lds_raw);
}

// 44 bytes of dynamic LDS per channel (four 8-byte accumulator words + mean, scale, shift): 3720 channels fill a CU's 160 KB
#define BN_APPLY_ACC_MAXC 3720
extern "C" size_t buctd_bn_acc_bytes(int C) { return C > 0 ? bnacc_bytes(C) : 0; }

static int acc_in_check(const buctd_bn_acc_in* st, const char* who) {
  BUCTD_CHECK_ARG(st && st->acc && st->rows > 0 && st->mean_out && st->invstd_out &&
                      (st->running_mean == nullptr) == (st->running_var == nullptr),
                  "%s: statistics accumulator, rows, mean_out and invstd_out are required; running_mean/var go together", who);
  return BUCTD_OK;
}
static BnAccFwd acc_in(const buctd_bn_acc_in* st) {
  BnAccFwd a;
  a.acc = (const long long*)st->acc; a.rows = (double)st->rows; a.eps = st->eps; a.momentum = st->momentum;
  a.mean_out = st->mean_out; a.invstd_out = st->invstd_out; a.rmean = st->running_mean; a.rvar = st->running_var;
  return a;
}

extern "C" int buctd_bn_apply_acc(const float* z, const buctd_bn_acc_in* st, const float* gamma, const float* beta,
                                  const float* residual, int relu, float* y, long rows, int C, void* stream) {
  BUCTD_CHECK_ARG(z && gamma && beta && y && rows > 0 && C > 0 && C % 4 == 0 && C <= BN_APPLY_ACC_MAXC,
                  "buctd_bn_apply_acc: bad argument (C must be a multiple of 4, <= %d)", BN_APPLY_ACC_MAXC);
  const int rc = acc_in_check(st, "buctd_bn_apply_acc");
  if (rc) return rc;
  BUCTD_CHECK_ARG(st->rows == rows, "buctd_bn_apply_acc: the statistics cover %ld rows, the tensor has %ld", st->rows, rows);
  const long total = rows * C;
  const BnApplyArgs a{z, acc_in(st), gamma, beta, residual, relu, y, total, C};
  if ((size_t)(BNACC_WORDS * 8 + 3 * 4) * C > 64 * 1024) {       // 44 bytes of LDS per channel: beyond 1489 channels above the default limit
    static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
    if (const int rc2 = buctd_raise_lds_limit(reinterpret_cast<const void*>(bn_apply_acc_kernel), 160 * 1024, attr_done, "buctd_bn_apply_acc"))
      return rc2;
  }
  hipLaunchKernelGGL(bn_apply_acc_kernel, dim3(acc_grid(total / 4)), dim3(256), (size_t)(BNACC_WORDS * 8 + 3 * 4) * C,
                     (hipStream_t)stream, a);
  BUCTD_CHECK_LAUNCH("buctd_bn_apply_acc");
  return BUCTD_OK;
}

extern "C" int buctd_bn_apply_acc_group(int n, const buctd_bn_apply_item* items, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && n <= BNG_MAX && items, "buctd_bn_apply_acc_group: 1..%d tensors", BNG_MAX);
  BnGroup<BnApplyArgs> g;
  g.n = n;
  g.first[0] = 0;
  size_t lds = 0;
  for (int k = 0; k < n; ++k) {
    const buctd_bn_apply_item& it = items[k];
    BUCTD_CHECK_ARG(it.z && it.gamma && it.beta && it.y && it.rows > 0 && it.C > 0 && it.C % 4 == 0 && it.C <= BN_APPLY_ACC_MAXC,
                    "buctd_bn_apply_acc_group: bad argument (C must be a multiple of 4, <= %d)", BN_APPLY_ACC_MAXC);
    const int rc = acc_in_check(&it.st, "buctd_bn_apply_acc_group");
    if (rc) return rc;
    BUCTD_CHECK_ARG(it.st.rows == it.rows, "buctd_bn_apply_acc_group: the statistics cover %ld rows, the tensor has %ld",
                    it.st.rows, it.rows);
    const long total = it.rows * it.C;
    g.s[k] = BnApplyArgs{it.z, acc_in(&it.st), it.gamma, it.beta, it.residual, it.relu, it.y, total, it.C};
    g.first[k + 1] = g.first[k] + (unsigned)acc_grid(total / 4);
    const size_t l = (size_t)(BNACC_WORDS * 8 + 3 * 4) * it.C;
    if (l > lds) lds = l;
  }
  for (int k = n; k < BNG_MAX; ++k) g.first[k + 1] = g.first[n];
  if (lds > 64 * 1024) {
    static unsigned char attr_done[BUCTD_MAX_DEVICES] = {0};
    if (const int rc2 = buctd_raise_lds_limit(reinterpret_cast<const void*>(bn_apply_acc_group_kernel), 160 * 1024, attr_done,
                                              "buctd_bn_apply_acc_group"))
      return rc2;
  }
  hipLaunchKernelGGL(bn_apply_acc_group_kernel, dim3(g.first[n]), dim3(256), lds, (hipStream_t)stream, g);
  BUCTD_CHECK_LAUNCH("buctd_bn_apply_acc_group");
  return BUCTD_OK;
}

// backward reduction into an accumulator: bn_bwd_reduce2_kernel with one exact integer addition per workgroup, channel and sum
struct BnBwdArgs {        // one BatchNorm backward (reduction and apply kernels)
  const float* dy;
  const float* y;
  const float* z;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  long long* acc;
  int relu;
  long rows;
  int C;
  long rows_per_block;     // reduction
  float inv_rows;          // apply ...
  float* dz;
  float* dres;
  float* dgamma;
  float* dbeta;
  int accumulate;
};
__device__ __forceinline__ void bn_bwd_reduce_acc_body(const BnBwdArgs& p, unsigned bid, f32x4 (&sm)[2][256]) {
  const float* __restrict__ dy = p.dy;
  const float* __restrict__ y = p.y;
  const float* __restrict__ z = p.z;
  const float* __restrict__ mean = p.mean;
  const float* __restrict__ invstd = p.invstd;
  const float* __restrict__ gamma = p.gamma;
  const float* __restrict__ beta = p.beta;
  long long* __restrict__ acc = p.acc;
  const int relu = p.relu, C = p.C;
  const long rows = p.rows, rows_per_block = p.rows_per_block;
  const int c4n = C >> 2;                 // C % 4 == 0, C / 4 <= 256
  const int rl = 256 / c4n;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const long r0 = (long)bid * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (tr < rl) {
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[tc];
    const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[tc];
    f32x4 sc = is, be = is;
    const bool rebuild = relu && !y;      // mask recomputed exactly as bn_apply formed its output
    if (rebuild) {
      sc = is * reinterpret_cast<const f32x4*>(gamma)[tc];
      be = reinterpret_cast<const f32x4*>(beta)[tc];
    }
    auto body = [&](f32x4 g, f32x4 zz, f32x4 yy) {
      if (relu) {
        if (rebuild) yy = (zz - mu) * sc + be;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yy[j] > 0.f)) g[j] = 0.f;
      }
      s1 += g;
      s2 += g * (zz - mu) * is;
    };
    long r = r0 + tr;
    const long step = rl;
    for (; r + 3 * step < r1; r += 4 * step) {
      f32x4 g[4], zz[4], yy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long o = (r + u * step) * c4n + tc;
        g[u] = reinterpret_cast<const f32x4*>(dy)[o];
        zz[u] = reinterpret_cast<const f32x4*>(z)[o];
        yy[u] = (relu && y) ? reinterpret_cast<const f32x4*>(y)[o] : zz[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) body(g[u], zz[u], yy[u]);
    }
    for (; r < r1; r += step) {
      const long o = r * c4n + tc;
      const f32x4 zz = reinterpret_cast<const f32x4*>(z)[o];
      body(reinterpret_cast<const f32x4*>(dy)[o], zz, (relu && y) ? reinterpret_cast<const f32x4*>(y)[o] : zz);
    }
  }
  sm[0][threadIdx.x] = s1;
  sm[1][threadIdx.x] = s2;
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rl; ++k) {
      s1 += sm[0][k * c4n + tc];
      s2 += sm[1][k * c4n + tc];
    }
    sm[0][tc] = s1;
    sm[1][tc] = s2;
  }
  __syncthreads();
  // lane = channel: one atomic instruction covers a contiguous run of words (bn_acc.h)
  const unsigned shard = bnacc_shard();
  for (int c = threadIdx.x; c < C; c += 256)
    bnacc_add(acc, C, shard, c, (double)reinterpret_cast<const float*>(&sm[0][0])[c], (double)reinterpret_cast<const float*>(&sm[1][0])[c]);
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_acc_kernel(BnBwdArgs p) {
  __shared__ f32x4 sm[2][256];
  bn_bwd_reduce_acc_body(p, blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_acc_group_kernel(BnGroup<BnBwdArgs> g_) {
  __shared__ f32x4 sm[2][256];
  const BnGroup<BnBwdArgs>& g = *(const BnGroup<BnBwdArgs>*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned bid, nblk;
  const int k = bn_group_find(g, &bid, &nblk);
  bn_bwd_reduce_acc_body(g.s[k], bid, sm);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of a fuse row (reference lib/models/pose_hrnet.py:257-265: y = relu(sum_j upsample(term_j))) together with the
// BatchNorm-backward sums of the terms that are conv -> BatchNorm outputs: the kernel that forms g = window sum of the masked
// upstream gradient also adds sum(g) and sum(g * zhat_t) of up to FSB_MAX terms into their accumulators (bn_acc.h), so that
// their BatchNorm backward starts at the apply kernel (acc_ready) - one reduction launch and one pass over g and z less per
// term.  g is written exactly as fuse_sum_bwd_kernel writes it; a workgroup owns a contiguous run of rows (fixed partition:
// the sums are reproducible run to run).
#define FSB_MAX 3
struct FuseBwdStatArgs {
  const float* dy;
  const float* y;        // forward output of the fuse row (ReLU mask), or nullptr
  float* g;
  int shift, N, H, W, C;
  long rows_per_block;
  int nt;
  const float* z[FSB_MAX];
  const float* mean[FSB_MAX];
  const float* invstd[FSB_MAX];
  long long* acc[FSB_MAX];
};
__global__ __launch_bounds__(256) void fuse_bwd_stats_kernel(FuseBwdStatArgs p) {
  __shared__ f32x4 sm[1 + FSB_MAX][256];
  const float* __restrict__ dy = p.dy;
  const float* __restrict__ y = p.y;
  const int C = p.C, c4n = C >> 2, rl = 256 / c4n;
  const int tc = threadIdx.x % c4n, tr = threadIdx.x / c4n;
  const int f = 1 << p.shift, hs = p.H >> p.shift, ws = p.W >> p.shift;
  const long rows = (long)p.N * hs * ws;
  const long r0 = (long)blockIdx.x * p.rows_per_block;
  long r1 = r0 + p.rows_per_block;
  if (r1 > rows) r1 = rows;
  f32x4 s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 s2[FSB_MAX];
  f32x4 mu[FSB_MAX], is[FSB_MAX];
#pragma unroll
  for (int t = 0; t < FSB_MAX; ++t) {
    s2[t] = s1;
    mu[t] = is[t] = s1;
    if (t < p.nt && tr < rl) {
      mu[t] = reinterpret_cast<const f32x4*>(p.mean[t])[tc];
      is[t] = reinterpret_cast<const f32x4*>(p.invstd[t])[tc];
    }
  }
  if (tr < rl) {
    auto window = [&](long r) -> f32x4 {      // g of low-resolution row r: the masked upstream gradient summed over its window
      long pix = r;
      const int w = (int)(pix % ws);
      pix /= ws;
      const int h = (int)(pix % hs);
      const int n = (int)(pix / hs);
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int dh = 0; dh < f; ++dh)
        for (int dw = 0; dw < f; ++dw) {
          const long o = (((long)n * p.H + h * f + dh) * p.W + w * f + dw) * c4n + tc;
          f32x4 d = reinterpret_cast<const f32x4*>(dy)[o];
          if (y) {
            const f32x4 yy = reinterpret_cast<const f32x4*>(y)[o];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!(yy[j] > 0.f)) d[j] = 0.f;
          }
          a += d;
        }
      return a;
    };
    auto tail = [&](long o, f32x4 a, const f32x4 (&zz)[FSB_MAX]) {
      reinterpret_cast<f32x4*>(p.g)[o] = a;
      s1 += a;
#pragma unroll
      for (int t = 0; t < FSB_MAX; ++t)
        if (t < p.nt) s2[t] += a * (zz[t] - mu[t]) * is[t];
    };
    long r = r0 + tr;
    const long step = rl;
    if (f == 1) {
      // same resolution (the down-sampling chains): four rows in flight per lane, as in bn_bwd_reduce_acc_body
      for (; r + 3 * step < r1; r += 4 * step) {
        f32x4 d[4], yy[4], zz[4][FSB_MAX];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long o = (r + u * step) * c4n + tc;
          d[u] = reinterpret_cast<const f32x4*>(dy)[o];
          yy[u] = y ? reinterpret_cast<const f32x4*>(y)[o] : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
          for (int t = 0; t < FSB_MAX; ++t)
            zz[u][t] = t < p.nt ? reinterpret_cast<const f32x4*>(p.z[t])[o] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(yy[u][j] > 0.f)) d[u][j] = 0.f;
          tail((r + u * step) * c4n + tc, d[u], zz[u]);
        }
      }
    }
    for (; r < r1; r += step) {
      const long o = r * c4n + tc;
      f32x4 zz[FSB_MAX];
#pragma unroll
      for (int t = 0; t < FSB_MAX; ++t)
        zz[t] = t < p.nt ? reinterpret_cast<const f32x4*>(p.z[t])[o] : (f32x4){0.f, 0.f, 0.f, 0.f};
      tail(o, window(r), zz);
    }
  }
  sm[0][threadIdx.x] = s1;
#pragma unroll
  for (int t = 0; t < FSB_MAX; ++t) sm[1 + t][threadIdx.x] = s2[t];
  __syncthreads();
  if (tr == 0) {
    for (int k = 1; k < rl; ++k) {
      s1 += sm[0][k * c4n + tc];
#pragma unroll
      for (int t = 0; t < FSB_MAX; ++t) s2[t] += sm[1 + t][k * c4n + tc];
    }
    sm[0][tc] = s1;
#pragma unroll
    for (int t = 0; t < FSB_MAX; ++t) sm[1 + t][tc] = s2[t];
  }
  __syncthreads();
  const unsigned shard = bnacc_shard();
  for (int c = threadIdx.x; c < C; c += 256) {
    const double v1 = (double)reinterpret_cast<const float*>(&sm[0][0])[c];
#pragma unroll
    for (int t = 0; t < FSB_MAX; ++t)
      if (t < p.nt) bnacc_add(p.acc[t], C, shard, c, v1, (double)reinterpret_cast<const float*>(&sm[1 + t][0])[c]);
  }
}

extern "C" int buctd_fuse_sum_bwd_bnstat(const float* dy, const float* y, int shift, int N, int H, int W, int C, float* g, int nt,
                                         const float* const* z, const float* const* mean, const float* const* invstd,
                                         void* const* acc, void* stream) {
  BUCTD_CHECK_ARG(dy && g && shift >= 0 && shift <= 5 && N > 0 && C > 0 && C % 4 == 0 && C / 4 <= 256,
                  "buctd_fuse_sum_bwd_bnstat: bad argument (C must be a multiple of 4, <= 1024)");
  BUCTD_CHECK_ARG((H >> shift) << shift == H && (W >> shift) << shift == W, "buctd_fuse_sum_bwd_bnstat: H/W not divisible by 2^shift");
  BUCTD_CHECK_ARG(nt >= 1 && nt <= FSB_MAX && z && mean && invstd && acc, "buctd_fuse_sum_bwd_bnstat: 1..%d BatchNorm terms", FSB_MAX);
  FuseBwdStatArgs a;
  a.dy = dy; a.y = y; a.g = g; a.shift = shift; a.N = N; a.H = H; a.W = W; a.C = C; a.nt = nt;
  for (int t = 0; t < FSB_MAX; ++t) {
    const int k = t < nt ? t : 0;
    BUCTD_CHECK_ARG(z[k] && mean[k] && invstd[k] && acc[k], "buctd_fuse_sum_bwd_bnstat: null pointer in term %d", k);
    a.z[t] = z[k]; a.mean[t] = mean[k]; a.invstd[t] = invstd[k]; a.acc[t] = (long long*)acc[k];
  }
  const long rows = (long)N * (H >> shift) * (W >> shift);
  const long nb = bwd2_blocks(rows, &a.rows_per_block);
  hipLaunchKernelGGL(fuse_bwd_stats_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
  BUCTD_CHECK_LAUNCH("buctd_fuse_sum_bwd_bnstat");
  return BUCTD_OK;
}

// dz = gamma * invstd * (g - s1 / M - zhat * s2 / M); dres = g.  LDS: [4][C] accumulator words | [6][C] floats (mean, invstd,
// gamma, beta, s1, s2)
__device__ __forceinline__ void bn_bwd_apply_acc_body(const BnBwdArgs& p, unsigned bid, unsigned nblk, unsigned char* lds_raw) {
  const float* __restrict__ dy = p.dy;
  const float* __restrict__ y = p.y;
  const float* __restrict__ z = p.z;
  const float* __restrict__ mean = p.mean;
  const float* __restrict__ invstd = p.invstd;
  const float* __restrict__ gamma = p.gamma;
  const float* __restrict__ beta = p.beta;
  const long long* __restrict__ acc = p.acc;
  float* __restrict__ dz = p.dz;
  float* __restrict__ dres = p.dres;
  float* __restrict__ dgamma = p.dgamma;
  float* __restrict__ dbeta = p.dbeta;
  const int relu = p.relu, C = p.C, accumulate = p.accumulate;
  const long total = p.rows * C;
  const float inv_rows = p.inv_rows;
  long long* wsum = reinterpret_cast<long long*>(lds_raw);
  float* tab = reinterpret_cast<float*>(lds_raw + (size_t)BNACC_WORDS * C * sizeof(long long));
  const long n4 = total >> 2;
  const long step = (long)nblk * 256;
  const bool ld_y = relu && y;
  long i = (long)bid * 256 + threadIdx.x;
  f32x4 g0 = (f32x4){0.f, 0.f, 0.f, 0.f}, g1 = g0, z0 = g0, z1 = g0, y0 = g0, y1 = g0;
  auto fetch = [&](long j) {
    if (j < n4) {
      g0 = reinterpret_cast<const f32x4*>(dy)[j];
      z0 = reinterpret_cast<const f32x4*>(z)[j];
      if (ld_y) y0 = reinterpret_cast<const f32x4*>(y)[j];
    }
    if (j + step < n4) {
      g1 = reinterpret_cast<const f32x4*>(dy)[j + step];
      z1 = reinterpret_cast<const f32x4*>(z)[j + step];
      if (ld_y) y1 = reinterpret_cast<const f32x4*>(y)[j + step];
    }
  };
  fetch(i);       // in flight while the sums are decoded
  bnacc_gather_lds(acc, C, wsum);
  for (int c = threadIdx.x; c < C; c += 256) {
    double s1, s2;
    bnacc_read_lds(wsum, C, c, &s1, &s2);
    tab[c] = mean[c];
    tab[C + c] = invstd[c];
    tab[2 * C + c] = gamma[c];
    tab[3 * C + c] = (relu && !y) ? beta[c] : 0.f;
    tab[4 * C + c] = (float)s1;
    tab[5 * C + c] = (float)s2;
    if (bid == 0) {
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    }
  }
  __syncthreads();
  const int dc = (int)((step * 4) % C);
  int c = (int)((i * 4) % C) - dc;
  auto one = [&](long j, int cc, f32x4 g, f32x4 zz, f32x4 yy) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(tab + cc);
    const f32x4 is = *reinterpret_cast<const f32x4*>(tab + C + cc);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(tab + 2 * C + cc);
    if (relu) {
      if (!y) {
        const f32x4 be = *reinterpret_cast<const f32x4*>(tab + 3 * C + cc);
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float sc = is[j4] * ga[j4];
          yy[j4] = (zz[j4] - mu[j4]) * sc + be[j4];
        }
      }
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
        if (!(yy[j4] > 0.f)) g[j4] = 0.f;
    }
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(tab + 4 * C + cc);
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(tab + 5 * C + cc);
    f32x4 o;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const float zh = (zz[j4] - mu[j4]) * is[j4];
      o[j4] = ga[j4] * is[j4] * (g[j4] - s1[j4] * inv_rows - zh * s2[j4] * inv_rows);
    }
    reinterpret_cast<f32x4*>(dz)[j] = o;
    if (dres) reinterpret_cast<f32x4*>(dres)[j] = g;
  };
  for (; i < n4; i += 2 * step) {       // two elements per trip: their loads are in flight together
    c += dc;
    if (c >= C) c -= C;
    one(i, c, g0, z0, y0);
    c += dc;
    if (c >= C) c -= C;
    if (i + step < n4) one(i + step, c, g1, z1, y1);
    fetch(i + 2 * step);
  }
}
__global__ __launch_bounds__(256, 6) void bn_bwd_apply_acc_kernel(BnBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bn_bwd_apply_acc_body(p, blockIdx.x, gridDim.x, lds_raw);
}
__global__ __launch_bounds__(256, 6) void bn_bwd_apply_acc_group_kernel(BnGroup<BnBwdArgs> g_) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const BnGroup<BnBwdArgs>& g = *(const BnGroup<BnBwdArgs>*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned bid, nblk;
  const int k = bn_group_find(g, &bid, &nblk);
  bn_bwd_apply_acc_body(g.s[k], bid, nblk, lds_raw);
}

/* buctd_bn_bwd on an accumulator (buctd_bn_acc_bytes(C), zero on entry unless acc_ready): acc_ready = 0 runs the
 * streaming reduction into it first; acc_ready = 1: the data gradient that produced dy already did
 * (buctd_conv3x3_bf16x6_bnstat_acc).  No finalize launch: the apply kernel decodes the sums itself and its first workgroup
 * writes dgamma / dbeta. */
static int bn_bwd_acc_fill(const buctd_bn_bwd_item& it, const char* who, BnBwdArgs& a, long* nb) {
  BUCTD_CHECK_ARG(it.dy && it.z && it.mean && it.invstd && it.gamma && it.dz && it.acc && it.rows > 0 && it.C > 0 && it.C % 4 == 0 &&
                      it.C / 4 <= 256,
                  "%s: bad argument (C must be a multiple of 4, <= 1024)", who);
  BUCTD_CHECK_ARG(!it.relu || it.y || it.beta, "%s: relu backward needs the forward output, or beta to rebuild its sign", who);
  a.dy = it.dy; a.y = it.y; a.z = it.z; a.mean = it.mean; a.invstd = it.invstd; a.gamma = it.gamma; a.beta = it.beta;
  a.acc = (long long*)it.acc; a.relu = it.relu; a.rows = it.rows; a.C = it.C;
  *nb = bwd2_blocks(it.rows, &a.rows_per_block);
  a.inv_rows = 1.0f / (float)it.rows;
  a.dz = it.dz; a.dres = it.dres; a.dgamma = it.dgamma; a.dbeta = it.dbeta; a.accumulate = it.accumulate;
  return BUCTD_OK;
}

extern "C" int buctd_bn_bwd_acc(const float* dy, const float* y, const float* z, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int relu, long rows, int C, float* dz, float* dres,
                                float* dgamma, float* dbeta, int accumulate, void* acc, int acc_ready, void* stream) {
  const buctd_bn_bwd_item it{dy, y, z, mean, invstd, gamma, beta, relu, rows, C, dz, dres, dgamma, dbeta, accumulate, acc, acc_ready};
  BnBwdArgs a;
  long nb;
  const int rc = bn_bwd_acc_fill(it, "buctd_bn_bwd_acc", a, &nb);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (!acc_ready) {
    hipLaunchKernelGGL(bn_bwd_reduce_acc_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    BUCTD_CHECK_LAUNCH("buctd_bn_bwd_acc(reduce)");
  }
  const long total = rows * C;
  hipLaunchKernelGGL(bn_bwd_apply_acc_kernel, dim3(acc_grid(total / 4, 6)), dim3(256), (size_t)(BNACC_WORDS * 8 + 6 * 4) * C, st, a);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd_acc(apply)");
  return BUCTD_OK;
}

/* n BatchNorm backwards in one launch each for the reductions that are still needed (items with acc_ready = 0) and the
 * applies: the branches of a HighResolutionModule.  Every tensor is processed exactly as by buctd_bn_bwd_acc. */
extern "C" int buctd_bn_bwd_acc_group(int n, const buctd_bn_bwd_item* items, void* stream) {
  BUCTD_CHECK_ARG(n > 0 && n <= BNG_MAX && items, "buctd_bn_bwd_acc_group: 1..%d tensors", BNG_MAX);
  BnGroup<BnBwdArgs> ga, gr;
  ga.n = n; gr.n = 0;
  ga.first[0] = gr.first[0] = 0;
  size_t lds = 0;
  for (int k = 0; k < n; ++k) {
    long nb;
    const int rc = bn_bwd_acc_fill(items[k], "buctd_bn_bwd_acc_group", ga.s[k], &nb);
    if (rc) return rc;
    const long total = items[k].rows * items[k].C;
    ga.first[k + 1] = ga.first[k] + (unsigned)acc_grid(total / 4, 6);
    const size_t l = (size_t)(BNACC_WORDS * 8 + 6 * 4) * items[k].C;
    if (l > lds) lds = l;
    if (!items[k].acc_ready) {
      gr.s[gr.n] = ga.s[k];
      gr.first[gr.n + 1] = gr.first[gr.n] + (unsigned)nb;
      ++gr.n;
    }
  }
  for (int k = n; k < BNG_MAX; ++k) ga.first[k + 1] = ga.first[n];
  for (int k = gr.n; k < BNG_MAX; ++k) gr.first[k + 1] = gr.first[gr.n];
  hipStream_t st = (hipStream_t)stream;
  if (gr.n) {
    hipLaunchKernelGGL(bn_bwd_reduce_acc_group_kernel, dim3(gr.first[gr.n]), dim3(256), 0, st, gr);
    BUCTD_CHECK_LAUNCH("buctd_bn_bwd_acc_group(reduce)");
  }
  hipLaunchKernelGGL(bn_bwd_apply_acc_group_kernel, dim3(ga.first[n]), dim3(256), lds, st, ga);
  BUCTD_CHECK_LAUNCH("buctd_bn_bwd_acc_group(apply)");
  return BUCTD_OK;
}
