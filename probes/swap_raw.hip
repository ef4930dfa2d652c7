// What v_permlane16_swap / v_permlane32_swap actually return (input = lane id).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline unsigned opaque_copy(unsigned v) { asm("" : "+v"(v)); return v; }
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(l, opaque_copy(l), false, false);
    const auto q = __builtin_amdgcn_permlane32_swap(l, opaque_copy(l), false, false);
    unsigned a = l, b = l + 100;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    const auto d = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    out[l * 8 + 0] = r[0]; out[l * 8 + 1] = r[1]; out[l * 8 + 2] = q[0]; out[l * 8 + 3] = q[1];
    out[l * 8 + 4] = a; out[l * 8 + 5] = b; out[l * 8 + 6] = d[0]; out[l * 8 + 7] = d[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[512]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: swap16(l,l) = {%u, %u}  swap32(l,l) = {%u, %u}  asm16(l, l+100) = {%u, %u}  builtin16(l, l+100) = {%u, %u}\n",
        l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
    return 0;
}
