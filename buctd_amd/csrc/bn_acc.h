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
// (scratch/ubench/atomic_fanin.hip: 506 workgroups x 96 words = +10 us on one copy, +0.6 us on 8 copies, one per XCD).
// BNACC_SHARDS copies, selected by the XCD a workgroup runs on, are summed (exactly) by the consumer.
//
// Layout: long long acc[BNACC_SHARDS][4][C], word kinds {s1 lo, s1 hi, s2 lo, s2 hi}; must be zero before the producer launch
// (the host hands out slices of a zeroed pool, ops.py: _AccPool).
#pragma once
#include "common.h"

#define BNACC_SHARDS 8
#define BNACC_WORDS 4

__host__ __device__ static inline size_t bnacc_bytes(int C) { return (size_t)BNACC_SHARDS * C * BNACC_WORDS * sizeof(long long); }

// one sum into its pair of limbs (lo at w_lo, hi at w_hi)
__device__ __forceinline__ void bnacc_add1(long long* w_lo, long long* w_hi, double v) {
  if (!(fabs(v) < 0x1p46)) {        // out of range or NaN: poison
    __hip_atomic_fetch_max(w_hi, (long long)1 << 61, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double h = trunc(v);
  const long long ih = (long long)h;
  const long long il = (long long)rint((v - h) * 0x1p48);     // v - h is exact, |.| < 1
  if (il) __hip_atomic_fetch_add(w_lo, il, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ih) __hip_atomic_fetch_add(w_hi, ih, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The copy a workgroup adds into: the XCD it runs on.  Atomics of ONE XCD on a word are served at that XCD's L2 rate; words
// that workgroups of different XCDs hit at the same time are serialised across the fabric (measured in the train step: shards
// picked by tile index cost 14-24 us per convolution launch, by XCD nothing measurable).  Any choice is correct - the
// consumer sums all copies exactly - only the speed depends on it.
__device__ __forceinline__ unsigned bnacc_shard() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & (BNACC_SHARDS - 1);
}

// this workgroup adds the pair (s1, s2) of channel c.  Call it with CONSECUTIVE lanes on consecutive channels: the words of
// one kind are contiguous over the channels, so a wavefront's atomic instruction is a few cache-line requests (the L2 serves
// atomic requests to one line one after the other, ~10 ns each: with a lane's four words side by side every lane was its own
// request - 192 requests per workgroup of the BatchNorm-backward reduction, +20 us on a 25 us kernel).
__device__ __forceinline__ void bnacc_add(long long* acc, int C, unsigned shard, int c, double s1, double s2) {
  long long* w = acc + (size_t)shard * BNACC_WORDS * C + c;
  bnacc_add1(w, w + C, s1);
  bnacc_add1(w + 2 * C, w + 3 * C, s2);
}

__device__ __forceinline__ double bnacc_decode(long long lo, long long hi) {
  if (hi >= ((long long)1 << 60)) return __builtin_nan("");
  return (double)hi + (double)lo * 0x1p-48;
}

// the two sums of channel c
__device__ __forceinline__ void bnacc_read(const long long* __restrict__ acc, int C, int c, double* s1, double* s2) {
  long long w[BNACC_WORDS] = {0, 0, 0, 0};
  bool bad = false;
#pragma unroll
  for (int sh = 0; sh < BNACC_SHARDS; ++sh) {
    const long long* p = acc + (size_t)sh * BNACC_WORDS * C + c;
    long long u[BNACC_WORDS];
#pragma unroll
    for (int k = 0; k < BNACC_WORDS; ++k) u[k] = p[(size_t)k * C];
    bad |= u[1] >= ((long long)1 << 60) || u[3] >= ((long long)1 << 60);
#pragma unroll
    for (int k = 0; k < BNACC_WORDS; ++k) w[k] += u[k];
  }
  *s1 = bad ? __builtin_nan("") : bnacc_decode(w[0], w[1]);
  *s2 = bad ? __builtin_nan("") : bnacc_decode(w[2], w[3]);
}

struct BnFwdStat { float mean, invstd; double mu, m2; };

// The same for a whole workgroup: all 256 threads fetch the BNACC_SHARDS * 4 * C words (coalesced 8-byte loads, a handful per
// thread instead of 32 in the threads of C channels) and add them - exactly - into wsum[4][C] in LDS.  Ends with a barrier;
// bnacc_read_lds then decodes a channel.  Every copy is checked for the poison mark BEFORE it is added: a diverged
// activation poisons the copies of several XCDs at once, and 4 x 2^61 wraps to a negative (8 x 2^61 to zero) sum that would
// decode as a finite number.  A poisoned word is stored as the saturated mark 2^61 (hi words only carry the mark, so a lo
// word never trips the check).
__device__ __forceinline__ void bnacc_gather_lds(const long long* __restrict__ acc, int C, long long* wsum) {
  const int n = BNACC_WORDS * C;
  for (int k = threadIdx.x; k < n; k += 256) {
    long long v[BNACC_SHARDS];
#pragma unroll
    for (int sh = 0; sh < BNACC_SHARDS; ++sh) v[sh] = acc[(size_t)sh * n + k];
    const bool is_hi = ((k / C) & 1) != 0;          // word kinds {s1 lo, s1 hi, s2 lo, s2 hi}
    long long t = 0;
    bool bad = false;
#pragma unroll
    for (int sh = 0; sh < BNACC_SHARDS; ++sh) {
      bad |= is_hi && v[sh] >= ((long long)1 << 60);
      t += v[sh];
    }
    wsum[k] = bad ? ((long long)1 << 61) : t;
  }
  __syncthreads();
}
__device__ __forceinline__ void bnacc_read_lds(const long long* wsum, int C, int c, double* s1, double* s2) {
  *s1 = bnacc_decode(wsum[c], wsum[C + c]);
  *s2 = bnacc_decode(wsum[2 * C + c], wsum[3 * C + c]);
}

// forward statistics of channel c from (sum z, sum z^2) over `rows` values: the arithmetic of bn_finalize_kernel
__device__ __forceinline__ BnFwdStat bnacc_fwd_stat_sums(double s1, double s2, double rows, float eps);
__device__ __forceinline__ BnFwdStat bnacc_fwd_stat_sums(double s1, double s2, double rows, float eps) {
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

__device__ __forceinline__ BnFwdStat bnacc_fwd_stat(const long long* __restrict__ acc, int C, int c, double rows, float eps) {
  double s1, s2;
  bnacc_read(acc, C, c, &s1, &s2);
  return bnacc_fwd_stat_sums(s1, s2, rows, eps);
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
