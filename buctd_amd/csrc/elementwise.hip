// HBM-bound streaming kernels of the HRNet trunk: layout changes at the NCHW
// API boundary, the HighResolutionModule fuse (nearest up-sampling + sum + ReLU,
// reference lib/models/pose_hrnet.py:250-265), the bilinear condition resize
// (lib/models/pose_hrnet_coam.py:755), bias-gradient column sums, max-pooling.
// Every kernel moves each tensor once, 16 bytes per lane where the channel count
// allows it.
#include "common.h"
#include "../../include/buctd_hip.h"

static int stream_grid(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------ add / relu ----
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n, int relu) {
  const long n4 = n >> 2;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i];
    if (b) v += reinterpret_cast<const f32x4*>(b)[i];
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    float v = a[i] + (b ? b[i] : 0.f);
    out[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
  const long n4 = n >> 2;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(a)[i] * reinterpret_cast<const f32x4*>(b)[i];
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, long n) {
  const long n4 = n >> 2;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const f32x4 yy = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!(yy[j] > 0.f)) g[j] = 0.f;
    reinterpret_cast<f32x4*>(dx)[i] = g;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out = x * alpha * (*dev_scalar)   (dev_scalar may be NULL)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, const float* __restrict__ dev_scalar,
                                                    float alpha, float* __restrict__ out, long n) {
  const float s = alpha * (dev_scalar ? dev_scalar[0] : 1.f);
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) out[i] = x[i] * s;
}
extern "C" int buctd_scale(const float* x, const float* dev_scalar, float alpha, float* out, long n, void* stream) {
  BUCTD_CHECK_ARG(x && out && n > 0, "buctd_scale: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dev_scalar, alpha, out,
                     n);
  BUCTD_CHECK_LAUNCH("buctd_scale");
  return BUCTD_OK;
}

// dst[r][cd0 + c] = src[r][cs0 + c], c < Cc : channel concat / split of NHWC tensors
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, long rows, int Cs, int cs0,
                                                            float* __restrict__ dst, int Cd, int cd0, int Cc) {
  const long total = rows * Cc;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const long r = i / Cc;
    const int c = (int)(i - r * Cc);
    dst[r * Cd + cd0 + c] = src[r * Cs + cs0 + c];
  }
}
extern "C" int buctd_copy_channels(const float* src, long rows, int Cs, int cs0, float* dst, int Cd, int cd0, int Cc,
                                   void* stream) {
  BUCTD_CHECK_ARG(src && dst && rows > 0 && Cc > 0 && cs0 >= 0 && cd0 >= 0 && cs0 + Cc <= Cs && cd0 + Cc <= Cd,
                  "buctd_copy_channels: bad argument");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(stream_grid(rows * Cc)), dim3(256), 0, (hipStream_t)stream, src, rows,
                     Cs, cs0, dst, Cd, cd0, Cc);
  BUCTD_CHECK_LAUNCH("buctd_copy_channels");
  return BUCTD_OK;
}

// out[b][i] = a[b][i] + v[i]  (position embedding added to every image's token tensor)
__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ v,
                                                        float* __restrict__ out, long n, long total) {
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) out[i] = a[i] + v[i % n];
}
extern "C" int buctd_add_bcast(const float* a, const float* v, float* out, long batch, long n, void* stream) {
  BUCTD_CHECK_ARG(a && v && out && batch > 0 && n > 0, "buctd_add_bcast: bad argument");
  hipLaunchKernelGGL(add_bcast_kernel, dim3(stream_grid(batch * n)), dim3(256), 0, (hipStream_t)stream, a, v, out, n,
                     batch * n);
  BUCTD_CHECK_LAUNCH("buctd_add_bcast");
  return BUCTD_OK;
}

extern "C" int buctd_add(const float* a, const float* b, float* out, long n, int relu, void* stream) {
  BUCTD_CHECK_ARG(a && out && n > 0, "buctd_add: bad argument");
  hipLaunchKernelGGL(add_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, relu);
  BUCTD_CHECK_LAUNCH("buctd_add");
  return BUCTD_OK;
}
extern "C" int buctd_mul(const float* a, const float* b, float* out, long n, void* stream) {
  BUCTD_CHECK_ARG(a && b && out && n > 0, "buctd_mul: bad argument");
  hipLaunchKernelGGL(mul_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  BUCTD_CHECK_LAUNCH("buctd_mul");
  return BUCTD_OK;
}
extern "C" int buctd_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream) {
  BUCTD_CHECK_ARG(dy && y && dx && n > 0, "buctd_relu_bwd: bad argument");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  BUCTD_CHECK_LAUNCH("buctd_relu_bwd");
  return BUCTD_OK;
}

// ----------------------------------------------------------------- colsum ----
// rows per partial: 64, or more when that would leave the final pass more than 8192 partials per column to walk (the bias
// gradients of the preNet's full-resolution convolutions: 3.5 M rows -> 55 k partials, 332 us in ONE workgroup for 3 columns)
#define CS_ROWS 64
static long colsum_chunk_rows(long rows) {
  const long want = (rows + 8191) / 8192;
  return want > CS_ROWS ? want : CS_ROWS;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long rows, int C, long chunk_rows,
                                                             float* __restrict__ part) {
  __shared__ float sm[256];
  const int cw = C < 256 ? C : 256;
  const int rl = 256 / cw;
  const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
  const long r0 = (long)blockIdx.x * chunk_rows;
  long r1 = r0 + chunk_rows;
  if (r1 > rows) r1 = rows;
  for (int cb = 0; cb < C; cb += cw) {
    const int c = cb + tc;
    float s = 0.f;
    if (c < C && tr < rl)
      for (long r = r0 + tr; r < r1; r += rl) s += x[r * C + c];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0 && c < C) {
      for (int k = 1; k < rl; ++k) s += sm[k * cw + tc];
      part[(long)blockIdx.x * C + c] = s;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nchunks, int C,
                                                           float* __restrict__ out, int accumulate) {
  // one workgroup per column (a wavefront per column walked 55 k partials in 864 dependent steps)
  __shared__ double sm[4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int k = threadIdx.x; k < nchunks; k += 256) s += (double)part[(long)k * C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int lo = __double2loint(s), hi = __double2hiint(s);
    lo = __shfl_xor(lo, o, 64);
    hi = __shfl_xor(hi, o, 64);
    s += __hiloint2double(hi, lo);
  }
  if (lane == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    out[c] = accumulate ? out[c] + (float)s : (float)s;
  }
}
extern "C" size_t buctd_colsum_workspace(long rows, int C) {
  const long cr = colsum_chunk_rows(rows);
  return (size_t)((rows + cr - 1) / cr) * C * sizeof(float);
}
extern "C" int buctd_colsum(const float* x, long rows, int C, float* out, int accumulate, void* workspace,
                            size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(x && out && rows > 0 && C > 0, "buctd_colsum: bad argument");
  const size_t need = buctd_colsum_workspace(rows, C);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_colsum: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  const long cr = colsum_chunk_rows(rows);
  const int nchunks = ceil_div(rows, cr);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nchunks), dim3(256), 0, st, x, rows, C, cr, (float*)workspace);
  BUCTD_CHECK_LAUNCH("buctd_colsum(partial)");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(256), 0, st, (const float*)workspace, nchunks, C,
                     out, accumulate);
  BUCTD_CHECK_LAUNCH("buctd_colsum(final)");
  return BUCTD_OK;
}

// ------------------------------------------------------ NCHW <-> NHWC ----
// One workgroup transposes a [C-chunk <=32][64 pixels] tile through LDS so that both the
// NCHW side (pixels contiguous) and the NHWC side (channels contiguous) are coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int Ctot, int c0, int Cc,
                                                           int HW, float* __restrict__ y) {
  __shared__ float tile[32][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64, cb = blockIdx.y * 32;
  const float* xs = x + ((long)n * Ctot + c0) * HW;
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int c = i >> 6, p = i & 63;
    if (cb + c < Cc && p0 + p < HW) tile[c][p] = xs[(long)(cb + c) * HW + p0 + p];
  }
  __syncthreads();
  float* ys = y + (long)n * HW * Cc;
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int p = i >> 5, c = i & 31;
    if (cb + c < Cc && p0 + p < HW) ys[(long)(p0 + p) * Cc + cb + c] = tile[c][p];
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int C, int HW,
                                                           float* __restrict__ y) {
  __shared__ float tile[32][65];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 64, cb = blockIdx.y * 32;
  const float* xs = x + (long)n * HW * C;
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int p = i >> 5, c = i & 31;
    if (cb + c < C && p0 + p < HW) tile[c][p] = xs[(long)(p0 + p) * C + cb + c];
  }
  __syncthreads();
  float* ys = y + (long)n * C * HW;
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int c = i >> 6, p = i & 63;
    if (cb + c < C && p0 + p < HW) ys[(long)(cb + c) * HW + p0 + p] = tile[c][p];
  }
}
extern "C" int buctd_nchw_to_nhwc(const float* x, int N, int Ctot, int c0, int Cc, int H, int W, float* y,
                                  void* stream) {
  BUCTD_CHECK_ARG(x && y && N > 0 && Cc > 0 && c0 >= 0 && c0 + Cc <= Ctot && H > 0 && W > 0,
                  "buctd_nchw_to_nhwc: bad argument");
  BUCTD_CHECK_ARG(N <= 65535, "buctd_nchw_to_nhwc: batch too large");
  dim3 grid(ceil_div(H * W, 64), ceil_div(Cc, 32), N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, Ctot, c0, Cc, H * W, y);
  BUCTD_CHECK_LAUNCH("buctd_nchw_to_nhwc");
  return BUCTD_OK;
}
extern "C" int buctd_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y, void* stream) {
  BUCTD_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "buctd_nhwc_to_nchw: bad argument");
  BUCTD_CHECK_ARG(N <= 65535, "buctd_nhwc_to_nchw: batch too large");
  dim3 grid(ceil_div(H * W, 64), ceil_div(C, 32), N);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, C, H * W, y);
  BUCTD_CHECK_LAUNCH("buctd_nhwc_to_nchw");
  return BUCTD_OK;
}

// ------------------------------------------------------------ fuse sum ----
struct FuseArgs {
  const float* t[4];
  int shift[4];
  int nterms;
};
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseArgs a, int N, int H, int W, int C4, int relu,
                                                       float* __restrict__ out) {
  const long total = (long)N * H * W * C4;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c4 = (int)(i % C4);
    long pix = i / C4;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < a.nterms) {
        const int s = a.shift[j];
        const int hs = H >> s, ws = W >> s;
        const long o = (((long)n * hs + (h >> s)) * ws + (w >> s)) * C4 + c4;
        v += reinterpret_cast<const f32x4*>(a.t[j])[o];
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}
extern "C" int buctd_fuse_sum(const float* const* terms, const int* shifts, int nterms, int N, int H, int W, int C,
                              int relu, float* out, void* stream) {
  BUCTD_CHECK_ARG(terms && shifts && out && nterms >= 1 && nterms <= 4, "buctd_fuse_sum: 1..4 terms");
  BUCTD_CHECK_ARG(C % 4 == 0 && N > 0 && H > 0 && W > 0, "buctd_fuse_sum: C must be a multiple of 4");
  FuseArgs a;
  a.nterms = nterms;
  for (int j = 0; j < 4; ++j) {
    a.t[j] = j < nterms ? terms[j] : nullptr;
    a.shift[j] = j < nterms ? shifts[j] : 0;
    if (j < nterms) {
      BUCTD_CHECK_ARG(terms[j] != nullptr && shifts[j] >= 0 && shifts[j] <= 5, "buctd_fuse_sum: bad term %d", j);
      BUCTD_CHECK_ARG((H >> shifts[j]) << shifts[j] == H && (W >> shifts[j]) << shifts[j] == W,
                      "buctd_fuse_sum: H/W not divisible by 2^shift for term %d", j);
    }
  }
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(fuse_sum_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, a, N, H, W, C / 4,
                     relu, out);
  BUCTD_CHECK_LAUNCH("buctd_fuse_sum");
  return BUCTD_OK;
}

// g[n][hs][ws][c] = sum_{dh,dw < 2^s} dy[n][hs*2^s+dh][ws*2^s+dw][c] * (y > 0)
__global__ __launch_bounds__(256) void fuse_sum_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           int shift, int N, int H, int W, int C4,
                                                           float* __restrict__ g) {
  const int hs = H >> shift, ws = W >> shift, f = 1 << shift;
  const long total = (long)N * hs * ws * C4;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c4 = (int)(i % C4);
    long pix = i / C4;
    const int w = (int)(pix % ws);
    pix /= ws;
    const int h = (int)(pix % hs);
    const int n = (int)(pix / hs);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < f; ++dh)
      for (int dw = 0; dw < f; ++dw) {
        const long o = (((long)n * H + h * f + dh) * W + w * f + dw) * C4 + c4;
        f32x4 d = reinterpret_cast<const f32x4*>(dy)[o];
        if (y) {
          const f32x4 yy = reinterpret_cast<const f32x4*>(y)[o];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(yy[j] > 0.f)) d[j] = 0.f;
        }
        acc += d;
      }
    reinterpret_cast<f32x4*>(g)[i] = acc;
  }
}
extern "C" int buctd_fuse_sum_bwd(const float* dy, const float* y, int shift, int N, int H, int W, int C, float* g,
                                  void* stream) {
  BUCTD_CHECK_ARG(dy && g && shift >= 0 && shift <= 5 && C % 4 == 0 && N > 0, "buctd_fuse_sum_bwd: bad argument");
  BUCTD_CHECK_ARG((H >> shift) << shift == H && (W >> shift) << shift == W,
                  "buctd_fuse_sum_bwd: H/W not divisible by 2^shift");
  const long total = (long)N * (H >> shift) * (W >> shift) * (C / 4);
  hipLaunchKernelGGL(fuse_sum_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, y, shift, N,
                     H, W, C / 4, g);
  BUCTD_CHECK_LAUNCH("buctd_fuse_sum_bwd");
  return BUCTD_OK;
}

// ------------------------------------------------------- bilinear resize ----
// torch F.interpolate(mode='bilinear', align_corners=False, antialias=False):
//   src = max((dst + 0.5) * scale - 0.5, 0), scale = in/out ; i0 = floor(src), i1 = min(i0+1, in-1)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int Ctot, int c0, int Cc,
                                                              int H, int W, int Ho, int Wo, int N,
                                                              float* __restrict__ y) {
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  const long total = (long)N * Ho * Wo * Cc;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c = (int)(i % Cc);
    long pix = i / Cc;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float fy = ((float)ho + 0.5f) * sh - 0.5f;
    float fx = ((float)wo + 0.5f) * sw - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* xs = x + ((long)n * Ctot + c0 + c) * H * W;
    const float v = hy * (hx * xs[(long)y0 * W + x0] + lx * xs[(long)y0 * W + x1]) +
                    ly * (hx * xs[(long)y1 * W + x0] + lx * xs[(long)y1 * W + x1]);
    y[i] = v;
  }
}
extern "C" int buctd_resize_bilinear(const float* x, int N, int Ctot, int c0, int Cc, int H, int W, int Ho, int Wo,
                                     float* y, void* stream) {
  BUCTD_CHECK_ARG(x && y && N > 0 && Cc > 0 && c0 >= 0 && c0 + Cc <= Ctot && H > 0 && W > 0 && Ho > 0 && Wo > 0,
                  "buctd_resize_bilinear: bad argument");
  const long total = (long)N * Ho * Wo * Cc;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, Ctot, c0,
                     Cc, H, W, Ho, Wo, N, y);
  BUCTD_CHECK_LAUNCH("buctd_resize_bilinear");
  return BUCTD_OK;
}

// ---------------------------------------------------------------- maxpool ----
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                          int Ho, int Wo, float* __restrict__ y,
                                                          int32_t* __restrict__ idx) {
  const long total = (long)N * Ho * Wo * C;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c = (int)(i % C);
    long pix = i / C;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float best = -INFINITY;
    int bi = -1;
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s) {
        const int hi = ho * 2 - 1 + r, wi = wo * 2 - 1 + s;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
          const float v = x[(((long)n * H + hi) * W + wi) * C + c];
          if (bi < 0 || v > best) {  // first maximum wins
            best = v;
            bi = hi * W + wi;
          }
        }
      }
    y[i] = best;
    idx[i] = bi;
  }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                                          int N, int H, int W, int C, int Ho, int Wo,
                                                          float* __restrict__ dx) {
  // gather form: each input pixel sums the outputs whose argmax it is (<= 4 candidates)
  const long total = (long)N * H * W * C;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
    const int c = (int)(i % C);
    long pix = i / C;
    const int wi = (int)(pix % W);
    pix /= W;
    const int hi = (int)(pix % H);
    const int n = (int)(pix / H);
    float s = 0.f;
    const int self = hi * W + wi;
    for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {  // windows [2ho-1, 2ho+1] containing hi
      if (ho >= Ho) continue;
      for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        if (idx[o] == self) s += dy[o];
      }
    }
    dx[i] = s;
  }
}
extern "C" int buctd_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int32_t* idx,
                                      void* stream) {
  BUCTD_CHECK_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0, "buctd_maxpool3x3s2_fwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_grid((long)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x,
                     N, H, W, C, Ho, Wo, y, idx);
  BUCTD_CHECK_LAUNCH("buctd_maxpool3x3s2_fwd");
  return BUCTD_OK;
}
extern "C" int buctd_maxpool3x3s2_bwd(const float* dy, const int32_t* idx, int N, int H, int W, int C, float* dx,
                                      void* stream) {
  BUCTD_CHECK_ARG(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0, "buctd_maxpool3x3s2_bwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid((long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy,
                     idx, N, H, W, C, Ho, Wo, dx);
  BUCTD_CHECK_LAUNCH("buctd_maxpool3x3s2_bwd");
  return BUCTD_OK;
}

// out = a0 + a1 (+ a2 (+ a3)) : the gradient fan-in of a tensor with several consumers (an HRNet branch output feeds up to
// four fuse rows, pose_hrnet.py:257-265) in ONE pass instead of autograd's chain of two-operand adds.  Summation order
// a0 + a1 + a2 + a3, left to right (what the chain computes).
struct AddNArgs { const float* a[4]; };
__global__ __launch_bounds__(256) void add_n_kernel(AddNArgs p, int n, float* __restrict__ out, long count) {
  const long n4 = count >> 2;
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    f32x4 s = reinterpret_cast<const f32x4*>(p.a[0])[i] + reinterpret_cast<const f32x4*>(p.a[1])[i];
    if (n > 2) s += reinterpret_cast<const f32x4*>(p.a[2])[i];
    if (n > 3) s += reinterpret_cast<const f32x4*>(p.a[3])[i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < count; i += step) {
    float s = p.a[0][i] + p.a[1][i];
    if (n > 2) s += p.a[2][i];
    if (n > 3) s += p.a[3][i];
    out[i] = s;
  }
}
extern "C" int buctd_add_n(const float* const* terms, int n, float* out, long count, void* stream) {
  BUCTD_CHECK_ARG(terms && n >= 2 && n <= 4 && out && count > 0, "buctd_add_n: 2..4 terms");
  AddNArgs a;
  for (int k = 0; k < 4; ++k) a.a[k] = k < n ? terms[k] : nullptr;
  for (int k = 0; k < n; ++k) BUCTD_CHECK_ARG(a.a[k], "buctd_add_n: null term");
  hipLaunchKernelGGL(add_n_kernel, dim3(stream_grid((count + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, n, out, count);
  BUCTD_CHECK_LAUNCH("buctd_add_n");
  return BUCTD_OK;
}
