// Device side of the per-sample input pipeline (reference lib/dataset/JointsDataset.py:134-361):
// affine person crop + ToTensor + Normalize in one kernel, written straight into the channels [0, 3) of the NCHW network
// input.  The Gaussian target (buctd_gaussian_target) and the condition heat-map (buctd_cond_render_into, which fills
// channels [3, 3+Cc)) are the other two kernels of a batch.
//
// The crop restates cv2.warpAffine(src_u8, M, (w, h), flags=INTER_LINEAR) (JointsDataset.py:287-291) bit for bit as
// OpenCV computes it on 8-bit images (imgwarp.cpp): M inverted in double; per destination pixel the source coordinate
// in 1/1024 px (rounded half-to-even per term), + 16, >> 5 -> 1/32 px; the four neighbours weighted by
// (32-fy)(32-fx)*32 ... fy*fx*32 (sum 2^15, exact - OpenCV's table correction never fires for the bilinear table);
// (sum + 2^14) >> 15; BORDER_CONSTANT 0.  HBM-bound: one pass, ~4 source bytes read (L2-served neighbours) and 12 bytes
// written per destination pixel.
#include "common.h"
#include "../../include/buctd_hip.h"

struct WarpParams {
  const buctd_warp_item* items;
  int dh, dw;
  float mean[3], inv_std[3];
  float* out;            // [B][>=3][dh][dw] float32, normalised
  long out_batch_stride;
  unsigned char* crop;   // optional [B][dh][dw][3] uint8 (meta['input_img'])
};

__device__ __forceinline__ int sat_int(double v) {
  v = rint(v);                                   // cvRound: half to even
  v = fmin(fmax(v, -2147483648.0), 2147483647.0);
  return (int)v;
}

__global__ __launch_bounds__(256) void warp_affine_norm_kernel(WarpParams p) {
  const int b = blockIdx.y;
  const buctd_warp_item it = p.items[b];
  // invert M exactly like cv::warpAffine (double, this operation order)
  double m0 = it.m[0], m1 = it.m[1], m2 = it.m[2], m3 = it.m[3], m4 = it.m[4], m5 = it.m[5];
  double d = m0 * m4 - m1 * m3;
  d = d != 0.0 ? 1.0 / d : 0.0;
  const double a11 = m4 * d, a22 = m0 * d;
  m0 = a11; m1 *= -d; m3 *= -d; m4 = a22;
  const double b1 = -m0 * m2 - m1 * m5, b2 = -m3 * m2 - m4 * m5;
  m2 = b1; m5 = b2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.dh * p.dw; i += gridDim.x * 256) {
    const int y = i / p.dw, x = i - y * p.dw;
    const int adelta = sat_int(m0 * (double)x * 1024.0), bdelta = sat_int(m3 * (double)x * 1024.0);
    const int x0 = sat_int((m1 * (double)y + m2) * 1024.0) + 16, y0 = sat_int((m4 * (double)y + m5) * 1024.0) + 16;
    const int X = (x0 + adelta) >> 5, Y = (y0 + bdelta) >> 5;
    int sx = X >> 5, sy = Y >> 5;
    sx = min(max(sx, -32768), 32767);
    sy = min(max(sy, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    int acc[3] = {0, 0, 0};
    auto tap = [&](int yy, int xx, int wgt) {
      if (wgt == 0 || yy < 0 || yy >= it.H || xx < 0 || xx >= it.W) return;
      if (it.rw > 0 && (xx < it.rx || xx >= it.rx + it.rw || yy < it.ry || yy >= it.ry + it.rh)) return;
      const int xs = it.flip ? it.W - 1 - xx : xx;
      const unsigned char* s = it.src + ((long)yy * it.W + xs) * 3;
      acc[0] += wgt * (int)s[0];
      acc[1] += wgt * (int)s[1];
      acc[2] += wgt * (int)s[2];
    };
    tap(sy, sx, w00);
    tap(sy, sx + 1, w01);
    tap(sy + 1, sx, w10);
    tap(sy + 1, sx + 1, w11);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int v = min(max((acc[c] + (1 << 14)) >> 15, 0), 255);
      // ToTensor: v / 255 in float32; Normalize: (t - mean) / std in float32 (a division, like torchvision)
      const float t = (float)v / 255.f;
      p.out[(long)b * p.out_batch_stride + ((long)c * p.dh + y) * p.dw + x] = (t - p.mean[c]) / p.inv_std[c];
      if (p.crop) p.crop[(((long)b * p.dh + y) * p.dw + x) * 3 + c] = (unsigned char)v;
    }
  }
}

extern "C" int buctd_warp_affine_norm(const buctd_warp_item* items_device, int B, int dst_h, int dst_w,
                                      const float* mean3, const float* std3, float* out, long out_batch_stride,
                                      unsigned char* crop_u8, void* stream) {
  BUCTD_CHECK_ARG(items_device && out && mean3 && std3 && B > 0 && dst_h > 0 && dst_w > 0 &&
                      out_batch_stride >= 3L * dst_h * dst_w,
                  "buctd_warp_affine_norm: bad argument");
  WarpParams p;
  p.items = items_device; p.dh = dst_h; p.dw = dst_w; p.out = out; p.out_batch_stride = out_batch_stride; p.crop = crop_u8;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.inv_std[c] = std3[c]; }   // inv_std holds std: divided by
  dim3 grid(ceil_div((long)dst_h * dst_w, 256 * 2), B);
  hipLaunchKernelGGL(warp_affine_norm_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  BUCTD_CHECK_LAUNCH("buctd_warp_affine_norm");
  return BUCTD_OK;
}
