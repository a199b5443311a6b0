// Greedy box NMS for the detector-box evaluation path (reference lib/nms/nms_kernel.cu:23-77 device part, 94-143 host
// part; lib/nms/cpu_nms.pyx:20-71).  Never executed on the BUCTD path itself (keep = [] for every BU / GT-box run,
// dataloader.py:627-630) - SURVEY 8f row f4.
//
// MI355X shape of the problem: the reference's 64-thread block IS one wave64 wavefront and its suppression word one
// uint64 per (row, column block).  Here a wavefront keeps its 64 column boxes in REGISTERS (one per lane) and broadcasts
// them with v_readlane instead of staging them in LDS; the greedy sweep over the mask also runs on the device (one
// wavefront, lanes = column blocks), so neither boxes nor mask ever cross PCIe and nothing is allocated per call
// (the reference cudaMallocs, copies the boxes in and the mask out on every call).
#include "common.h"
#include "../../include/buctd_hip.h"

__device__ __forceinline__ float nms_iou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                                         float by2) {
  const float left = fmaxf(ax1, bx1), right = fminf(ax2, bx2);
  const float top = fmaxf(ay1, by1), bottom = fminf(ay2, by2);
  const float w = fmaxf(right - left + 1.f, 0.f), h = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = w * h;
  const float sa = (ax2 - ax1 + 1.f) * (ay2 - ay1 + 1.f);
  const float sb = (bx2 - bx1 + 1.f) * (by2 - by1 + 1.f);
  return inter / (sa + sb - inter);
}

__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// boxes [n][dim >= 4] sorted by descending score; mask [n][col_blocks]: bit i of mask[r][c] = box 64c+i overlaps box r
// by more than thresh (and comes after it).
__global__ __launch_bounds__(64) void nms_mask_kernel(int n, int dim, float thresh, const float* __restrict__ boxes,
                                                      unsigned long long* __restrict__ mask) {
  const int row_start = blockIdx.y, col_start = blockIdx.x, lane = threadIdx.x;
  const int col_blocks = gridDim.x;
  const int col_size = min(n - col_start * 64, 64), row_size = min(n - row_start * 64, 64);
  const int cb = col_start * 64 + lane, rb = row_start * 64 + lane;
  float cx1 = 0.f, cy1 = 0.f, cx2 = 0.f, cy2 = 0.f, rx1 = 0.f, ry1 = 0.f, rx2 = 0.f, ry2 = 0.f;
  if (lane < col_size) {
    cx1 = boxes[(long)cb * dim + 0]; cy1 = boxes[(long)cb * dim + 1];
    cx2 = boxes[(long)cb * dim + 2]; cy2 = boxes[(long)cb * dim + 3];
  }
  if (lane < row_size) {
    rx1 = boxes[(long)rb * dim + 0]; ry1 = boxes[(long)rb * dim + 1];
    rx2 = boxes[(long)rb * dim + 2]; ry2 = boxes[(long)rb * dim + 3];
  }
  unsigned long long t = 0;
  const int start = row_start == col_start ? lane + 1 : 0;
  for (int i = 0; i < col_size; ++i) {          // uniform trip count: readlane needs a uniform source lane
    const float bx1 = lane_bcast(cx1, i), by1 = lane_bcast(cy1, i), bx2 = lane_bcast(cx2, i), by2 = lane_bcast(cy2, i);
    if (i >= start && nms_iou(rx1, ry1, rx2, ry2, bx1, by1, bx2, by2) > thresh) t |= 1ull << i;
  }
  if (lane < row_size) mask[(long)rb * col_blocks + col_start] = t;
}

// greedy sweep (reference host loop nms_kernel.cu:123-139) by one wavefront: lane j owns removal word j, j + 64, ...
__global__ __launch_bounds__(64) void nms_sweep_kernel(int n, int col_blocks, const unsigned long long* __restrict__ mask,
                                                       int* __restrict__ keep, int* __restrict__ num_out) {
  extern __shared__ unsigned long long remv[];
  const int lane = threadIdx.x;
  for (int j = lane; j < col_blocks; j += 64) remv[j] = 0;
  __syncthreads();
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    const int nblock = i >> 6, inblock = i & 63;
    const bool alive = !((remv[nblock] >> inblock) & 1ull);     // same address in every lane: broadcast read
    if (alive) {
      if (lane == 0) keep[kept] = i;
      ++kept;
      for (int j = nblock + lane; j < col_blocks; j += 64) remv[j] |= mask[(long)i * col_blocks + j];
    }
    __syncthreads();
  }
  if (lane == 0) *num_out = kept;
}

extern "C" size_t buctd_nms_workspace(int boxes_num) {
  if (boxes_num <= 0) return 0;
  return (size_t)boxes_num * ceil_div(boxes_num, 64) * sizeof(unsigned long long);
}

extern "C" int buctd_nms(int* keep_out, int* num_out, const float* boxes, int boxes_num, int boxes_dim, float thresh,
                         void* workspace, size_t workspace_bytes, void* stream) {
  BUCTD_CHECK_ARG(keep_out && num_out && boxes && boxes_num > 0 && boxes_dim >= 4, "buctd_nms: bad argument");
  const size_t need = buctd_nms_workspace(boxes_num);
  if (!workspace || workspace_bytes < need) {
    buctd_set_error("buctd_nms: workspace %zu bytes < required %zu", workspace_bytes, need);
    return BUCTD_EWORKSPACE;
  }
  const int col_blocks = ceil_div(boxes_num, 64);
  BUCTD_CHECK_ARG((size_t)col_blocks * 8 <= 64 * 1024, "buctd_nms: more than 524288 boxes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, st, boxes_num, boxes_dim, thresh, boxes,
                     (unsigned long long*)workspace);
  BUCTD_CHECK_LAUNCH("buctd_nms(mask)");
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), (size_t)col_blocks * 8, st, boxes_num, col_blocks,
                     (const unsigned long long*)workspace, keep_out, num_out);
  BUCTD_CHECK_LAUNCH("buctd_nms(sweep)");
  return BUCTD_OK;
}

// cpu_nms (lib/nms/cpu_nms.pyx:20-71): host code, dets [n][5] = x1 y1 x2 y2 score in any order; a box is suppressed when
// its overlap with a kept, higher-scoring box is >= thresh (note: >=, where the device kernel and nms.py use >).
// order: indices by descending score, as the caller's argsort produced them.
extern "C" int buctd_cpu_nms(const float* dets, int n, const int* order, float thresh, int* keep_out, int* num_out) {
  BUCTD_CHECK_ARG(dets && order && keep_out && num_out && n >= 0, "buctd_cpu_nms: bad argument");
  unsigned char* dead = n > 0 ? new unsigned char[n]() : nullptr;
  int kept = 0;
  for (int a = 0; a < n; ++a) {
    const int i = order[a];
    if (dead[i]) continue;
    keep_out[kept++] = i;
    const float* bi = dets + (long)i * 5;
    const float area_i = (bi[2] - bi[0] + 1.f) * (bi[3] - bi[1] + 1.f);
    for (int b = a + 1; b < n; ++b) {
      const int j = order[b];
      if (dead[j]) continue;
      const float* bj = dets + (long)j * 5;
      const float x1 = bi[0] >= bj[0] ? bi[0] : bj[0], y1 = bi[1] >= bj[1] ? bi[1] : bj[1];
      const float x2 = bi[2] <= bj[2] ? bi[2] : bj[2], y2 = bi[3] <= bj[3] ? bi[3] : bj[3];
      const float w = x2 - x1 + 1.f > 0.f ? x2 - x1 + 1.f : 0.f, h = y2 - y1 + 1.f > 0.f ? y2 - y1 + 1.f : 0.f;
      const float inter = w * h;
      const float area_j = (bj[2] - bj[0] + 1.f) * (bj[3] - bj[1] + 1.f);
      if (inter / (area_i + area_j - inter) >= thresh) dead[j] = 1;
    }
  }
  delete[] dead;
  *num_out = kept;
  return BUCTD_OK;
}
