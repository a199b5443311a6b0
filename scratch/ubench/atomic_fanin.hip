// How expensive is "every workgroup of a one-round launch adds C x K 64-bit words into one small accumulator" on MI355X?
// (sizing of the BatchNorm statistics without finalize launches: 506 workgroups x 48 channels x 2 quantities x limbs)
// hipcc --offload-arch=gfx950 -O3 -o atomic_fanin atomic_fanin.hip && ./atomic_fanin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: plain stores of per-workgroup partials (today's scheme); 1: u64 atomic adds; 2: f64 atomic adds
// words = C * K per workgroup; stride_w = distance in 8-byte words between two accumulator words; shards: copies selected by
// blockIdx % shards (shard_w words apart)
__global__ __launch_bounds__(256) void fanin_kernel(unsigned long long* acc, int words, int stride_w, int shards, long shard_w,
                                                    int mode, int spin) {
  // a little unequal work in front so that arrivals are not perfectly simultaneous (spin = 0: all at once)
  if (spin) {
    const long long t0 = clock64();
    const long long wait = (long long)((blockIdx.x * 2654435761u) % (unsigned)spin);
    while (clock64() - t0 < wait) {}
  }
  const int t = threadIdx.x;
  if (mode == 0) {
    if (t < words) acc[(long)blockIdx.x * words + t] = (unsigned long long)t;
    return;
  }
  unsigned long long* base = acc + (long)(blockIdx.x % shards) * shard_w;
  for (int w = t; w < words; w += 256) {
    if (mode == 1)
      __hip_atomic_fetch_add(base + (long)w * stride_w, (unsigned long long)(w + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      unsafeAtomicAdd(reinterpret_cast<double*>(base + (long)w * stride_w), 1.0);
  }
}

int main() {
  const size_t bytes = 64u << 20;
  unsigned long long* acc;
  CHECK(hipMalloc(&acc, bytes));
  CHECK(hipMemset(acc, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, int G, int words, int stride_w, int shards, long shard_w, int mode, int spin) {
    const int reps = 200;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fanin_kernel, dim3(G), dim3(256), 0, 0, acc, words, stride_w, shards, shard_w, mode, spin);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fanin_kernel, dim3(G), dim3(256), 0, 0, acc, words, stride_w, shards, shard_w, mode, spin);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s G=%4d words=%4d stride=%4dB shards=%d spin=%5d : %7.2f us per launch (%.1f ns per atomic)\n", name, G, words,
           stride_w * 8, shards, spin, ms * 1000.f / reps, mode ? ms * 1e6f / reps / ((float)G * words) : 0.f);
  };
  for (int spin : {0, 4000}) {
    run("empty (words=0)", 506, 0, 1, 1, 0, 1, spin);
    run("plain partial stores", 506, 96, 1, 1, 0, 0, spin);
    for (int words : {96, 192, 288, 768}) {
      run("u64 atomics, contiguous", 506, words, 1, 1, 0, 1, spin);
      run("u64 atomics, 64 B apart", 506, words, 8, 1, 0, 1, spin);
      run("u64 atomics, 256 B apart", 506, words, 32, 1, 0, 1, spin);
      run("u64 atomics, 4 KB apart", 506, words, 512, 1, 0, 1, spin);
      run("u64 atomics, contiguous, 8 shards", 506, words, 1, 8, 1 << 16, 1, spin);
      run("u64 atomics, 256 B, 8 shards", 506, words, 32, 8, 1 << 16, 1, spin);
      run("f64 atomics, contiguous", 506, words, 1, 1, 0, 2, spin);
      run("f64 atomics, 256 B apart", 506, words, 32, 1, 0, 2, spin);
    }
    run("u64 atomics, contiguous, G=2048", 2048, 96, 1, 1, 0, 1, spin);
    run("u64 atomics, 256 B, G=2048", 2048, 96, 32, 1, 0, 1, spin);
  }
  // correctness: the sums
  CHECK(hipMemset(acc, 0, bytes));
  hipLaunchKernelGGL(fanin_kernel, dim3(506), dim3(256), 0, 0, acc, 96, 1, 1, 0, 1, 0);
  std::vector<unsigned long long> h(96);
  CHECK(hipMemcpy(h.data(), acc, 96 * 8, hipMemcpyDeviceToHost));
  bool ok = true;
  for (int w = 0; w < 96; ++w) ok &= h[w] == 506ull * (w + 1);
  printf("sums %s\n", ok ? "exact" : "WRONG");
  return 0;
}
