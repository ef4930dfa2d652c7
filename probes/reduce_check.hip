// Checks the DPP / permlane reductions of csrc/common.h against __shfl_xor butterflies on random data.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I texocr_amd/csrc probes/reduce_check.hip -o probes/reduce_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "common.h"
using namespace txo;

__device__ inline float ref_sum(float v, int from, int to) { for (int o = from; o >= to; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ inline float ref_max(float v, int from, int to) { for (int o = from; o >= to; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

__global__ void check(const float* in, float* out, int* iout) {
    const int t = threadIdx.x, g = blockIdx.x * blockDim.x + t;
    const float v = in[g];
    float* o = out + (size_t)g * 16;
    o[0] = wave_sum(v);      o[1] = ref_sum(v, 32, 1);
    o[2] = wave_max(v);      o[3] = ref_max(v, 32, 1);
    o[4] = grp4_sum(v);      o[5] = ref_sum(v, 32, 16);
    o[6] = grp4_max(v);      o[7] = ref_max(v, 32, 16);
    o[8] = row16_sum(v);     o[9] = ref_sum(v, 8, 1);
    o[10] = row8_sum(v);     o[11] = ref_sum(v, 4, 1);
    o[12] = row16_max(v);    o[13] = ref_max(v, 8, 1);
    o[14] = dpp_mov<DPP_ROR8>(v); o[15] = __shfl_xor(v, 8, 64);
    int* io = iout + (size_t)g * 8;
    const int iv = (int)(v * 1000.f);
    io[0] = wave_sum(iv); { int s = iv; for (int x = 32; x > 0; x >>= 1) s += __shfl_xor(s, x, 64); io[1] = s; }
    float best = floorf(v * 4.f); int bi = g & 63;     // coarse values -> ties
    float rb = best; int ri = bi;
    wave_argmax(best, bi);
    for (int x = 32; x > 0; x >>= 1) { const float ov = __shfl_xor(rb, x, 64); const int oi = __shfl_xor(ri, x, 64); if (ov > rb || (ov == rb && oi < ri)) { rb = ov; ri = oi; } }
    io[2] = bi; io[3] = ri; io[4] = __float_as_int(best); io[5] = __float_as_int(rb);
    io[6] = __float_as_int(xor16(v)); io[7] = __float_as_int(__shfl_xor(v, 16, 64));
}

int main() {
    const int n = 256 * 8;
    float* h = (float*)malloc(n * 4);
    srand(3); for (int i = 0; i < n; ++i) h[i] = (rand() % 2001 - 1000) / 250.0f;
    float *din, *dout; int* diout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 64); hipMalloc(&diout, n * 32);
    hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(8), dim3(256), 0, 0, din, dout, diout);
    float* ho = (float*)malloc(n * 64); int* hi = (int*)malloc(n * 32);
    hipMemcpy(ho, dout, n * 64, hipMemcpyDeviceToHost); hipMemcpy(hi, diout, n * 32, hipMemcpyDeviceToHost);
    const char* names[8] = {"wave_sum", "wave_max", "grp4_sum", "grp4_max", "row16_sum", "row8_sum", "row16_max", "ror8"};
    int bad = 0;
    for (int k = 0; k < 8; ++k) {
        double worst = 0; int nb = 0;
        for (int i = 0; i < n; ++i) { const double d = fabs(ho[i * 16 + 2 * k] - ho[i * 16 + 2 * k + 1]); if (d > worst) worst = d; if (d > 1e-3) ++nb; }
        printf("%-10s max |dpp - shfl| = %.3g  bad %d\n", names[k], worst, nb); bad += nb;
    }
    int b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < n; ++i) { b1 += hi[i * 8] != hi[i * 8 + 1]; b2 += hi[i * 8 + 2] != hi[i * 8 + 3] || hi[i * 8 + 4] != hi[i * 8 + 5]; b3 += hi[i * 8 + 6] != hi[i * 8 + 7]; }
    printf("wave_sum(int) bad %d, wave_argmax bad %d, xor16 bad %d\n", b1, b2, b3);
    if (b2) for (int i = 0; i < 8; ++i) printf("  lane %d: dpp (%x, %d) shfl (%x, %d)\n", i, hi[i*8+4], hi[i*8+2], hi[i*8+5], hi[i*8+3]);
    return (bad + b1 + b2 + b3) ? 1 : 0;
}
