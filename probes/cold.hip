#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_stream(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) b[i] = a[i];
}
// each block loads NL x 4 KB (16 B per thread per load), all issued up front, then stamps
template <int NL>
__global__ void k_probe(const u32x4* __restrict__ w, unsigned long long* st, unsigned* sink) {
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  u32x4 r[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) r[i] = w[((size_t)blockIdx.x * NL + i) * 256 + threadIdx.x];
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc ^= r[i].x ^ r[i].w;
  asm volatile("" :: "v"(acc));
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0x12345u) sink[0] = acc;
  if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = t1; }
}
template <int NL> void run(const char* tag, const u32x4* w, const u32x4* big, u32x4* big2, size_t nbig, unsigned long long* st, unsigned* sink, int blocks, bool thrash) {
  std::vector<unsigned long long> h(2 * blocks);
  double sum = 0, mx = 0; int reps = 20;
  for (int r = 0; r < reps; ++r) {
    if (thrash) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, big, big2, nbig);
    hipLaunchKernelGGL((k_probe<NL>), dim3(blocks), dim3(256), 0, 0, w, st, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> d(blocks);
    for (int b = 0; b < blocks; ++b) d[b] = (h[2 * b + 1] - h[2 * b]) / 100.0;
    std::sort(d.begin(), d.end());
    sum += d[blocks / 2]; mx += d[blocks - 1];
  }
  printf("%-28s blocks %4d  %3d KB/block: median %.2f us, slowest block %.2f us\n", tag, blocks, NL * 4, sum / reps, mx / reps);
}
int main() {
  size_t nbig = (512ull << 20) / 16; u32x4 *big, *big2, *w; unsigned long long* st; unsigned* sink;
  hipMalloc(&big, nbig * 16); hipMalloc(&big2, nbig * 16); hipMalloc(&w, 64ull << 20); hipMalloc(&st, 1 << 20); hipMalloc(&sink, 64);
  hipMemset(big, 1, nbig * 16); hipMemset(w, 1, 64ull << 20);
  for (int thrash = 0; thrash < 2; ++thrash) {
    const char* tag = thrash ? "after 512 MB stream (cold)" : "back-to-back (warm MALL/L2?)";
    run<1>(tag, w, big, big2, nbig, st, sink, 64, thrash);
    run<4>(tag, w, big, big2, nbig, st, sink, 64, thrash);
    run<12>(tag, w, big, big2, nbig, st, sink, 64, thrash);
    run<24>(tag, w, big, big2, nbig, st, sink, 32, thrash);
    run<24>(tag, w, big, big2, nbig, st, sink, 256, thrash);
  }
  return 0;
}
