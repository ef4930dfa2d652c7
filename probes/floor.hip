#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_empty(float* p) {}
__global__ void k_ldst(const float* __restrict__ a, float* __restrict__ b) { int i = blockIdx.x * 256 + threadIdx.x; b[i] = a[i] + 1.f; }
__global__ void k_chain2(const float* __restrict__ a, const int* __restrict__ idx, float* __restrict__ b) {
  int i = blockIdx.x * 256 + threadIdx.x; int j = idx[i]; b[i] = a[j] + 1.f; }
__global__ void k_red(const float* __restrict__ a, float* __restrict__ b) {
  __shared__ float s[256]; int i = blockIdx.x * 256 + threadIdx.x; s[threadIdx.x] = a[i]; __syncthreads();
  float v = s[(threadIdx.x + 64) & 255]; __syncthreads(); s[threadIdx.x] = v; __syncthreads(); b[i] = s[(threadIdx.x + 1) & 255]; }
template <class F> double run(F f, int n) {
  for (int i = 0; i < 50; ++i) f();
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < n; ++i) f();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / n;
}
int main() {
  float *a, *b; int* idx; size_t n = 2048 * 256;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&idx, n * 4); hipMemset(a, 0, n * 4); hipMemset(idx, 0, n * 4);
  hipStream_t s; hipStreamCreate(&s);
  for (int blocks : {1, 64, 512, 2048}) {
    printf("blocks %4d: empty %.2f us | ld+st %.2f | 2-level chain %.2f | lds 3 barriers %.2f\n", blocks,
      run([&]{ hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, a); }, 2000),
      run([&]{ hipLaunchKernelGGL(k_ldst, dim3(blocks), dim3(256), 0, s, a, b); }, 2000),
      run([&]{ hipLaunchKernelGGL(k_chain2, dim3(blocks), dim3(256), 0, s, a, idx, b); }, 2000),
      run([&]{ hipLaunchKernelGGL(k_red, dim3(blocks), dim3(256), 0, s, a, b); }, 2000));
  }
  // same through a graph of 26 kernels
  hipGraph_t g; hipGraphExec_t ge; hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 26; ++i) hipLaunchKernelGGL(k_ldst, dim3(64), dim3(256), 0, s, a, b);
  hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  printf("graph of 26 ld+st kernels (64 blocks): %.2f us per kernel\n", run([&]{ hipGraphLaunch(ge, s); }, 300) / 26);
  return 0;
}
