// Does an LDS-DMA (global_load_lds_dwordx4 issued from inline asm, M0 = wave-uniform destination) reach LDS addresses above
// 64 KB on gfx950 (160 KB LDS per workgroup)?  Fills 156 KB of dynamic LDS in 1-KB wave chunks and reads it back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned char* dst, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)lds;      // LDS byte address of the dynamic segment
    for (int c = wave; c < nchunk; c += 8) glds16(src + (size_t)c * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + c * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < nchunk * 64; i += 512) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(lds)[i];
}
int main() {
    const int nchunk = 156; const size_t n = (size_t)nchunk * 1024;
    std::vector<unsigned char> h(n); for (size_t i = 0; i < n; ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    unsigned char *s, *d; hipMalloc(&s, n); hipMalloc(&d, n); hipMemcpy(s, h.data(), n, hipMemcpyHostToDevice); hipMemset(d, 0, n);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)n);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), n, 0, s, d, nchunk);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned char> g(n); hipMemcpy(g.data(), d, n, hipMemcpyDeviceToHost);
    size_t bad = 0, first = n; for (size_t i = 0; i < n; ++i) if (g[i] != h[i]) { ++bad; if (first == n) first = i; }
    printf("%s: %zu of %zu bytes wrong (first at %zu)\n", hipGetErrorString(e), bad, n, first);
    return bad != 0;
}
