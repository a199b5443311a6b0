// BatchNorm statistics without a finalize launch (reference: nn.BatchNorm2d in lib/models/pose_hrnet.py:41-57).
//
// The kernel that PRODUCES a tensor (a convolution epilogue, a data-gradient epilogue, a streaming reduction) reduces it to one
// pair of sums per workgroup and channel - forward (sum z, sum z^2), backward (sum g, sum g zhat) - and adds the pair into a
// small accumulator with integer atomics; the kernel that CONSUMES the statistics (the next convolution's input staging,
// bn_apply, bn_bwd_apply) decodes the 2 numbers per channel in its prologue.  No launch sits between producer and consumer
// (round 4: 584 finalize launches per CoAM-W48 step, 5 us of work each, 16-26 us each on the wall clock next to the
// one-round convolution kernels of the other streams).
//
// Why integers: integer addition is associative, so the accumulated value does not depend on the order in which the
// workgroups arrive - the step stays bit-deterministic (the 1-rank == 2-rank and streams-on == streams-off tests) where a
// floating-point atomicAdd would not.  A sum v (fp64) is added as two 64-bit limbs
//     hi = trunc(v)            (unit 1)         lo = rint((v - hi) * 2^48)   (unit 2^-48)
// - both conversions are exact functions of v, the decoded sum  hi + lo * 2^-48  is within 2^-49 per addend of the exact sum.
// Range: |v| < 2^46 per addend (a workgroup's sum of squares: rms activations up to ~4e5 over a 448-row tile; 2^13 addends
// per shard word stay below 2^59); beyond that, or for a NaN, the producer POISONS the word (atomic max with 2^61) and the
// consumer decodes NaN, so a broken activation shows up in the output as it does with floating-point statistics.  Resolution: 2^-48 absolute - far
// below eps = 1e-5 of the variance for any activation scale, relative 2^-24 of a workgroup's sum of squares down to
// rms ~1e-5.
//
// Contention: every workgroup of a launch adds into the same few words at the same time; one word takes ~20 ns per atomic
// (scratch/ubench/atomic_fanin.hip: 506 workgroups x 96 words = +10 us on one copy, +0.6 us on 8 copies).  BNACC_SHARDS
// copies, selected by the workgroup index, are summed (exactly) by the consumer.
//
// Layout: long long acc[BNACC_SHARDS][C][4] = {s1 lo, s1 hi, s2 lo, s2 hi}; must be zero before the producer launch
// (the host hands out slices of a zeroed pool, ops.py: _AccPool).
#pragma once
#include "common.h"

#define BNACC_SHARDS 8
#define BNACC_WORDS 4

__host__ __device__ static inline size_t bnacc_bytes(int C) { return (size_t)BNACC_SHARDS * C * BNACC_WORDS * sizeof(long long); }

// one sum into the pair of limbs at w (w[0] lo, w[1] hi)
__device__ __forceinline__ void bnacc_add1(long long* w, double v) {
  if (!(fabs(v) < 0x1p46)) {        // out of range or NaN: poison
    __hip_atomic_fetch_max(w + 1, (long long)1 << 61, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double h = trunc(v);
  const long long ih = (long long)h;
  const long long il = (long long)rint((v - h) * 0x1p48);     // v - h is exact, |.| < 1
  if (il) __hip_atomic_fetch_add(w, il, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ih) __hip_atomic_fetch_add(w + 1, ih, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroup `shard_id` (any integer: its low bits pick the copy) adds the pair (s1, s2) of channel c
__device__ __forceinline__ void bnacc_add(long long* acc, int C, unsigned shard_id, int c, double s1, double s2) {
  long long* w = acc + ((size_t)(shard_id % BNACC_SHARDS) * C + c) * BNACC_WORDS;
  bnacc_add1(w, s1);
  bnacc_add1(w + 2, s2);
}

__device__ __forceinline__ double bnacc_decode(long long lo, long long hi) {
  if (hi >= ((long long)1 << 60)) return __builtin_nan("");
  return (double)hi + (double)lo * 0x1p-48;
}

// the two sums of channel c
__device__ __forceinline__ void bnacc_read(const long long* __restrict__ acc, int C, int c, double* s1, double* s2) {
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  ll2 a = {0, 0}, b = {0, 0};
  bool bad = false;
#pragma unroll
  for (int sh = 0; sh < BNACC_SHARDS; ++sh) {
    const ll2* p = reinterpret_cast<const ll2*>(acc + ((size_t)sh * C + c) * BNACC_WORDS);
    const ll2 u = p[0], v = p[1];
    bad |= u.y >= ((long long)1 << 60) || v.y >= ((long long)1 << 60);
    a += u;
    b += v;
  }
  *s1 = bad ? __builtin_nan("") : bnacc_decode(a.x, a.y);
  *s2 = bad ? __builtin_nan("") : bnacc_decode(b.x, b.y);
}

// forward statistics of channel c from (sum z, sum z^2) over `rows` values: the arithmetic of bn_finalize_kernel
struct BnFwdStat { float mean, invstd; double mu, m2; };
__device__ __forceinline__ BnFwdStat bnacc_fwd_stat(const long long* __restrict__ acc, int C, int c, double rows, float eps) {
  double s1, s2;
  bnacc_read(acc, C, c, &s1, &s2);
  BnFwdStat r;
  r.mu = s1 / rows;
  double m2 = s2 - s1 * r.mu;
  if (m2 < 0.0) m2 = 0.0;        // (a NaN stays a NaN)
  r.m2 = m2;
  const double var = m2 / rows;
  r.mean = (float)r.mu;
  r.invstd = (float)(1.0 / sqrt(var + (double)eps));
  return r;
}

// running statistics, as bn_finalize_kernel updates them (unbiased variance)
__device__ __forceinline__ void bnacc_running(const BnFwdStat& s, double rows, float momentum, float* rmean, float* rvar, int c) {
  const double unb = rows > 1.0 ? s.m2 / (rows - 1.0) : s.m2 / rows;
  rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * s.mu);
  rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
}

// What a consumer of forward statistics needs besides the accumulator: written once per launch by ONE workgroup.
struct BnAccFwd {
  const long long* acc;      // null: the consumer takes mean / invstd from memory as before
  double rows;               // values per channel
  float eps, momentum;
  float* mean_out;           // [C] (+ invstd_out [C]): the statistics in the form the backward kernels read
  float* invstd_out;
  float* rmean;              // running statistics to update (null: not tracked)
  float* rvar;
};
