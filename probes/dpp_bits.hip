// Is a reduction through DPP/permlane-swap bit-identical to the same partner pattern through ds_bpermute?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "common.h"
using namespace txo;
__device__ inline float sh(float v, int src) { return __shfl(v, src, 64); }
__global__ void k(const float* in, float* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, l = threadIdx.x & 63;
    const float v = in[g];
    float* o = out + (size_t)g * 8;
    o[0] = wave_sum(v);
    float e = v;
    e += sh(e, l ^ 1); e += sh(e, l ^ 2); e += sh(e, (l & ~7) | (7 - (l & 7))); e += sh(e, (l & ~15) | (15 - (l & 15)));
    { const float p = sh(e, l ^ 16); const bool odd = (l >> 4) & 1; const float a = odd ? p : e, b = odd ? e : p; e = a + b; }
    { const float p = sh(e, l ^ 32); const bool hi = (l >> 5) & 1; const float a = hi ? p : e, b = hi ? e : p; e = a + b; }
    o[1] = e;
    o[2] = v + dpp_mov<DPP_XOR1>(v); o[3] = v + sh(v, l ^ 1);
    o[4] = row16_sum(v);
    float r = v; r += sh(r, l ^ 1); r += sh(r, l ^ 2); r += sh(r, (l & ~7) | (7 - (l & 7))); r += sh(r, (l & ~15) | (15 - (l & 15)));
    o[5] = r;
    o[6] = grp4_sum(v);
    float q = v;
    { const float p = sh(q, l ^ 16); const bool odd = (l >> 4) & 1; const float a = odd ? p : q, b = odd ? q : p; q = a + b; }
    { const float p = sh(q, l ^ 32); const bool hi = (l >> 5) & 1; const float a = hi ? p : q, b = hi ? q : p; q = a + b; }
    o[7] = q;
}
int main() {
    const int n = 256 * 64;
    float* h = (float*)malloc(n * 4);
    srand(5); for (int i = 0; i < n; ++i) h[i] = (rand() % 200001 - 100000) / 31337.0f;
    float *din, *dout; hipMalloc(&din, n * 4); hipMalloc(&dout, n * 32);
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, din, dout);
    float* ho = (float*)malloc(n * 32); hipMemcpy(ho, dout, n * 32, hipMemcpyDeviceToHost);
    const char* nm[4] = {"wave_sum", "one xor1 step", "row16_sum", "grp4_sum"};
    for (int k2 = 0; k2 < 4; ++k2) {
        int nb = 0; for (int i = 0; i < n; ++i) nb += __builtin_memcmp(&ho[i * 8 + 2 * k2], &ho[i * 8 + 2 * k2 + 1], 4) != 0;
        printf("%-14s lanes whose bits differ from the bpermute emulation: %d of %d\n", nm[k2], nb, n);
    }
    return 0;
}
