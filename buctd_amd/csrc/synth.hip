// Generative pose synthesis on the device (reference lib/dataset/pose_synthesis.py:234-817: the
// training-time condition of every "generative sampling" recipe, ~34 ms of numpy per person on a CPU worker).
//
// One wavefront per (person, joint).  The five error types of the reference (jitter, miss, inversion, swap, good) each
// propose N candidates on a ring around a source key point and keep those far enough from the other sources; one
// survivor is drawn uniformly, then the type from the renormalised probability table.  "Uniform survivor of N iid
// candidates" is done in two passes over a counter-based generator: count the survivors (64 candidates per step,
// ballot + popcount), draw (source, rank), regenerate that source's candidates and take the survivor of that rank.
// float64 throughout, like the reference.  oracle/pose_synthesis.py is the CPU twin (same generator, same scheme).
//
// Known deviation (documented, shared by the twin): the reference updates synth_joints[j] IN PLACE inside its joint loop, so
// for the SECOND joint of a symmetric pair its inversion source synth_joints[pair] (pose_synthesis.py:271) - and the
// distance guards against it - is the first joint AFTER its own perturbation (or (0, 0) if that joint was dropped).  The
// joints here are synthesized in parallel, each from the UNPERTURBED pose.  The per-joint class frequencies measured
// against the imported reference (oracle/make_golden.py:pose_synthesis_case, 1500 runs, max gap 0.03) include that effect;
// sample-level equality is claimed against the twin only.
#include "common.h"
#include "../../include/buctd_hip.h"

#define SY_MAXSRC 64
#define SY_N 500

struct SynthArgs {
  const double* joints;     // [B][K][3]
  const double* estimated;  // [B][K][3]
  const double* near;       // [B][M][K][3]
  const double* area;       // [B]
  const int* num_overlap;   // [B]
  double* out;              // [B][K][3]
  int B, K, M;
  unsigned long long seed;
  buctd_synth_tables t;
};

__device__ __forceinline__ double sy_uniform(unsigned long long seed, int person, int joint, int stream,
                                             unsigned long long index) {
  const unsigned long long key = ((((unsigned long long)person * 64 + joint) * 64 + stream) << 24);
  unsigned long long z = (index + key) + 1ull;
  z = seed + z * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

struct SyCtx {
  unsigned long long seed;
  int b, j, ns, lane;
  const double* sx;
  const double* sy;
};

// candidate `idx` of (stream, source s): position on the ring [r_lo, r_hi]; returns whether it survives.
// guard_mode 0: farther than r from every source but s; 1: farther than thr from every source but s;
// 2: farther than r from the sources g0 and g1 (g1 < 0: none)
__device__ __forceinline__ bool sy_candidate(const SyCtx& c, int stream, int s, int idx, double r_lo, double r_hi,
                                             int guard_mode, double thr, int g0, int g1, double* px, double* py) {
  const double ang = sy_uniform(c.seed, c.b, c.j, stream, 2ull * idx) * 6.283185307179586;
  const double r = r_lo + (r_hi - r_lo) * sy_uniform(c.seed, c.b, c.j, stream, 2ull * idx + 1);
  const double x = c.sx[s] + r * cos(ang), y = c.sy[s] + r * sin(ang);
  *px = x;
  *py = y;
  bool ok = true;
  if (guard_mode == 2) {
    const double dx0 = c.sx[g0] - x, dy0 = c.sy[g0] - y;
    ok = sqrt(dx0 * dx0 + dy0 * dy0) > r;
    if (g1 >= 0) {
      const double dx1 = c.sx[g1] - x, dy1 = c.sy[g1] - y;
      ok = ok && sqrt(dx1 * dx1 + dy1 * dy1) > r;
    }
    return ok;
  }
  const double lim = guard_mode == 1 ? thr : r;
  for (int i = 0; i < c.ns; ++i) {
    if (i == s) continue;
    const double dx = c.sx[i] - x, dy = c.sy[i] - y;
    ok = ok && sqrt(dx * dx + dy * dy) > lim;
  }
  return ok;
}

__device__ __forceinline__ int sy_count(const SyCtx& c, int stream, int s, int n, double r_lo, double r_hi, int mode,
                                        double thr, int g0, int g1) {
  int cnt = 0;
  for (int base = 0; base < n; base += 64) {        // uniform trip count
    const int idx = base + c.lane;
    double x, y;
    const bool ok = idx < n && sy_candidate(c, stream, s, idx, r_lo, r_hi, mode, thr, g0, g1, &x, &y);
    cnt += __popcll(__ballot(ok));
  }
  return cnt;
}

// the survivor of rank k (0-based, k < number of survivors) of (stream, s): wave-uniform result
__device__ __forceinline__ void sy_select(const SyCtx& c, int stream, int s, int n, double r_lo, double r_hi, int mode,
                                          double thr, int g0, int g1, int k, double* ox, double* oy) {
  for (int base = 0; base < n; base += 64) {
    const int idx = base + c.lane;
    double x = 0.0, y = 0.0;
    const bool ok = idx < n && sy_candidate(c, stream, s, idx, r_lo, r_hi, mode, thr, g0, g1, &x, &y);
    const unsigned long long m = __ballot(ok);
    const int cnt = __popcll(m);
    if (k < cnt) {
      unsigned long long mm = m;
      for (int q = 0; q < k; ++q) mm &= mm - 1;      // drop the k lowest set bits
      const int src_lane = __ffsll((long long)mm) - 1;
      *ox = __shfl(x, src_lane, 64);
      *oy = __shfl(y, src_lane, 64);
      return;
    }
    k -= cnt;
  }
  *ox = 0.0;
  *oy = 0.0;
}

__global__ __launch_bounds__(64) void synth_pose_kernel(SynthArgs a) {
  __shared__ double sx[SY_MAXSRC], sy[SY_MAXSRC];
  __shared__ int cnt_s[SY_MAXSRC];
  const int b = blockIdx.x / a.K, j = blockIdx.x % a.K, lane = threadIdx.x;
  const double* J = a.joints + (long)b * a.K * 3;
  const double* E = a.estimated + (long)b * a.K * 3;
  const double* NR = a.near + (long)b * a.M * a.K * 3;
  double* O = a.out + ((long)b * a.K + j) * 3;
  int nv = 0;
  for (int q = 0; q < a.K; ++q) nv += J[q * 3 + 2] > 0.0 ? 1 : 0;
  auto synth_xy = [&](int q, double* x, double* y) {     // joints.copy(), un-annotated joints from the estimate
    const double* p = J[q * 3 + 2] == 0.0 ? E + q * 3 : J + q * 3;
    *x = p[0];
    *y = p[1];
  };
  const double area = a.area[b];
  const double var = (a.t.sigmas[j] * 2.0) * (a.t.sigmas[j] * 2.0);
  const double d10 = sqrt(-2.0 * area * var * log(0.10)), d50 = sqrt(-2.0 * area * var * log(0.50)),
               d85 = sqrt(-2.0 * area * var * log(0.85));
  const int pair = a.t.pair[j];
  // sources: synthesized joint | neighbours' joint j | inversion source | neighbours' paired joint
  int ns = 0, nswap = 0, nswapinv = 0;
  if (lane == 0) {
    synth_xy(j, &sx[0], &sy[0]);
    ns = 1;
    for (int m = 0; m < a.M && ns < SY_MAXSRC; ++m)
      if (NR[((long)m * a.K + j) * 3 + 2] > 0.0) { sx[ns] = NR[((long)m * a.K + j) * 3]; sy[ns] = NR[((long)m * a.K + j) * 3 + 1]; ++ns; ++nswap; }
    if (pair >= 0 && J[pair * 3 + 2] > 0.0 && ns < SY_MAXSRC) { synth_xy(pair, &sx[ns], &sy[ns]); ++ns; }
    if (pair >= 0)
      for (int m = 0; m < a.M && ns < SY_MAXSRC; ++m)
        if (NR[((long)m * a.K + pair) * 3 + 2] > 0.0) { sx[ns] = NR[((long)m * a.K + pair) * 3]; sy[ns] = NR[((long)m * a.K + pair) * 3 + 1]; ++ns; ++nswapinv; }
  }
  ns = __shfl(ns, 0, 64);
  nswap = __shfl(nswap, 0, 64);
  nswapinv = __shfl(nswapinv, 0, 64);
  __syncthreads();
  const bool has_inv = pair >= 0 && J[pair * 3 + 2] > 0.0;
  const int skip = 1 + nswap;       // 'the inversion source' of the reference, whether or not one exists
  SyCtx c;
  c.seed = a.seed; c.b = b; c.j = j; c.ns = ns; c.lane = lane; c.sx = sx; c.sy = sy;

  double cx[5], cy[5];
  bool have[5];
  // one-source types: count, draw a rank, select
  auto single = [&](int stream, int pick_stream, int s, int n, double lo, double hi, int mode, int g0, int g1, int t) {
    const int cnt = sy_count(c, stream, s, n, lo, hi, mode, 0.0, g0, g1);
    have[t] = cnt > 0;
    cx[t] = cy[t] = 0.0;
    if (cnt > 0) {
      (void)sy_uniform(a.seed, b, j, pick_stream, 0);                      // source draw (one source: unused)
      const int k = (int)(sy_uniform(a.seed, b, j, pick_stream, 1) * cnt);
      sy_select(c, stream, s, n, lo, hi, mode, 0.0, g0, g1, k, &cx[t], &cy[t]);
    }
  };
  single(0, 30, 0, SY_N, d85, d50, 0, 0, -1, 0);                            // jitter
  {                                                                        // miss
    long total = 0;
    for (int s = 0; s < ns; ++s) {
      const int n = sy_count(c, 1 + s, s, 4 * SY_N, d50, d10, 1, d50, 0, -1);
      const int wgt = s == 0 ? n : n / 4;
      if (lane == 0) cnt_s[s] = n;
      total += wgt;
    }
    __syncthreads();
    have[1] = total > 0;
    cx[1] = cy[1] = 0.0;
    if (total > 0) {
      long tt = (long)(sy_uniform(a.seed, b, j, 40, 0) * (double)total);
      for (int s = 0; s < ns; ++s) {
        const int n = cnt_s[s], wgt = s == 0 ? n : n / 4;
        if (tt < wgt) {
          const int k = (int)(sy_uniform(a.seed, b, j, 40, 1) * n);
          sy_select(c, 1 + s, s, 4 * SY_N, d50, d10, 1, d50, 0, -1, k, &cx[1], &cy[1]);
          break;
        }
        tt -= wgt;
      }
    }
    __syncthreads();
  }
  have[2] = false; cx[2] = cy[2] = 0.0;
  if (has_inv) single(41, 42, skip, SY_N, 0.0, d50, 0, 0, -1, 2);           // inversion
  have[3] = false; cx[3] = cy[3] = 0.0;
  if (nswap > 0 || nswapinv > 0) {                                         // swap
    const int g1 = skip < ns ? skip : -1;
    long total = 0;
    for (int s = 0; s < ns; ++s) {
      int n = 0;
      if (s != 0 && s != skip) n = sy_count(c, 43 + s, s, SY_N, 0.0, d50, 2, 0.0, 0, g1);
      if (lane == 0) cnt_s[s] = n;
      total += n;
    }
    __syncthreads();
    have[3] = total > 0;
    if (total > 0) {
      long tt = (long)(sy_uniform(a.seed, b, j, 60, 0) * (double)total);
      for (int s = 0; s < ns; ++s) {
        const int n = cnt_s[s];
        if (tt < n) {
          const int k = (int)(sy_uniform(a.seed, b, j, 60, 1) * n);
          sy_select(c, 43 + s, s, SY_N, 0.0, d50, 2, 0.0, 0, g1, k, &cx[3], &cy[3]);
          break;
        }
        tt -= n;
      }
    }
    __syncthreads();
  }
  single(61, 62, 0, SY_N / 4, 0.0, d85, 0, 0, -1, 4);                       // good

  const int ov = a.num_overlap[b];
  const double p_jit = a.t.jitter_p[nv <= 10 ? 0 : 1][a.t.jitter_cls[j]];
  const double p_miss = a.t.miss_p[nv <= 5 ? 0 : (nv <= 10 ? 1 : 2)][a.t.miss_cls[j]];
  const double p_inv = a.t.inv_p[a.t.inv_cls[j]];
  const bool crowded = (nv <= 10 && ov > 0) || (nv <= 15 && ov >= 3);
  const double p_swap = a.t.swap_p[crowded ? 0 : 1][a.t.swap_cls[j]];
  const double p_good = 1.0 - (p_jit + p_miss + p_inv + p_swap);
  double pr[5] = {have[0] ? p_jit : 0.0, have[1] ? p_miss : 0.0, have[2] ? p_inv : 0.0, have[3] ? p_swap : 0.0,
                  have[4] ? p_good : 0.0};
  const double norm = pr[0] + pr[1] + pr[2] + pr[3] + pr[4];
  if (lane == 0) {
    if (norm == 0.0) {
      O[0] = O[1] = O[2] = 0.0;
    } else {
      const double u = sy_uniform(a.seed, b, j, 63, 0) * norm;
      double acc = 0.0;
      int chosen = 4;
      for (int t = 0; t < 5; ++t) {
        acc += pr[t];
        if (u < acc) { chosen = t; break; }
      }
      while (!have[chosen]) --chosen;
      O[0] = cx[chosen];
      O[1] = cy[chosen];
      O[2] = a.t.out_vis;
    }
  }
}

extern "C" int buctd_synthesize_pose(const buctd_synth_tables* tables, const double* joints, const double* estimated,
                                     const double* near_joints, const double* area, const int* num_overlap, int B, int K,
                                     int M, unsigned long long seed, double* out, void* stream) {
  BUCTD_CHECK_ARG(tables && joints && estimated && area && num_overlap && out && B > 0 && K > 0 && K <= 32 && M >= 0 &&
                      (M == 0 || near_joints) && 2 + 2 * M <= SY_MAXSRC && B < (1 << 20),
                  "buctd_synthesize_pose: bad argument (K <= 32, M <= 31)");
  SynthArgs a;
  a.joints = joints; a.estimated = estimated; a.near = near_joints; a.area = area; a.num_overlap = num_overlap; a.out = out;
  a.B = B; a.K = K; a.M = M; a.seed = seed; a.t = *tables;
  hipLaunchKernelGGL(synth_pose_kernel, dim3(B * K), dim3(64), 0, (hipStream_t)stream, a);
  BUCTD_CHECK_LAUNCH("buctd_synthesize_pose");
  return BUCTD_OK;
}
