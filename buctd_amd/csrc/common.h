// Shared helpers for libbuctd_hip.so (gfx950 / MI355X only).
// Every exported entry point returns 0 on success and a negative code on
// failure; the message is retrievable through buctd_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define BUCTD_OK 0
#define BUCTD_EINVAL (-1)
#define BUCTD_ELAUNCH (-2)
#define BUCTD_EWORKSPACE (-3)

void buctd_set_error(const char* fmt, ...);

#define BUCTD_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      buctd_set_error(__VA_ARGS__);           \
      return BUCTD_EINVAL;                    \
    }                                         \
  } while (0)

#define BUCTD_CHECK_LAUNCH(name)                                                   \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      buctd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return BUCTD_ELAUNCH;                                                        \
    }                                                                              \
  } while (0)

// Raises the dynamic-LDS limit of kernel `fn` to `bytes` once per DEVICE: the attribute belongs to the device's copy of the
// kernel, so a process-wide flag would leave a second GPU of the same process at the 64 KB default.  `done` is the caller's
// static flag array (one per kernel); a race at first use only repeats the idempotent call.
#define BUCTD_MAX_DEVICES 16
static inline int buctd_raise_lds_limit(const void* fn, int bytes, unsigned char (&done)[BUCTD_MAX_DEVICES], const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BUCTD_MAX_DEVICES) {
    buctd_set_error("%s: cannot identify the current device", who);
    return BUCTD_ELAUNCH;
  }
  if (done[dev]) return BUCTD_OK;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    buctd_set_error("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
    return BUCTD_ELAUNCH;
  }
  done[dev] = 1;
  return BUCTD_OK;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for 256-thread blocks (4 waves); result valid in all threads.
__device__ __forceinline__ float block_sum_256(float v, float* sm /* >=4 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// Counter-based dropout: keep/scale factor of element `idx` under `seed` (shared by the fused softmax, the
// element-wise dropout and the fused attention kernels, so forward and backward regenerate the same mask).
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, float p_drop, float inv_keep) {
  // splitmix64 finalizer over a Weyl sequence keyed by the seed
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p_drop ? inv_keep : 0.f;
}

