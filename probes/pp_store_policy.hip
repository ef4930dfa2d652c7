// does the cache policy of the epilogue's stores change what the 256x256 GEMM's epilogue costs?  (r04: the epilogue's 128 KB per tile leave a
// CU at ~16 B/clk with plain stores)  EpiStore (plain) vs non-temporal stores vs raw buffer stores with sc0|sc1 / nt aux bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../texocr_amd/csrc/gemm_pp.h"
using namespace txo;
template <int POLICY> struct EpiPol {
    bf16* out; int ldo; const float* bias;
    static constexpr bool PAIRED = false;
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int, float (&cb)[32]) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[e] = 0.f; }
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&)[32], const float (&)[10], bool valid) const {
        if (!valid) return;
        union { bf16 h[8]; u32x4 u; } c;
#pragma unroll
        for (int e = 0; e < 8; ++e) c.h[e] = __float2bfloat16(v[e]);
        bf16* p = out + (size_t)m * ldo + n;
        if constexpr (POLICY == 0) st16(p, c.u);
        else if constexpr (POLICY == 1) __builtin_nontemporal_store(c.u, reinterpret_cast<u32x4*>(p));
        else {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(c.u, r, (int)(((size_t)m * ldo + n) * 2), 0, POLICY == 2 ? 3 : (POLICY == 3 ? 17 : 2));   // aux: 3 = sc0|sc1?, 17 = sc1|..., 2 = nt (experiment)
        }
    }
};
template <int POLICY> static void run(const bf16* A, const bf16* W, bf16* C, int M, int N, int K, const char* name) {
    EpiPol<POLICY> epi{C, N, nullptr};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_gemm_pp(0, A, W, M, N, K, epi);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch_gemm_pp(0, A, W, M, N, K, epi);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s M=%d N=%d K=%d: %.3f ms = %6.0f TFLOP/s\n", name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    const int M = 150784, NMAX = 6144, KMAX = 3072;
    bf16 *A, *W, *C;
    hipMalloc(&A, (size_t)M * KMAX * 2); hipMalloc(&W, (size_t)NMAX * KMAX * 2); hipMalloc(&C, (size_t)M * NMAX * 2);
    std::vector<unsigned short> h((size_t)NMAX * KMAX);
    for (auto& x : h) x = 0x3c00 + (rand() & 0x3ff);
    hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)M * KMAX; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)M * KMAX - off) * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int N : {6144, 2304, 768}) {
            const int K = N == 768 ? 3072 : 768;
            run<0>(A, W, C, M, N, K, "plain stores");
            run<1>(A, W, C, M, N, K, "non-temporal stores");
            run<2>(A, W, C, M, N, K, "buffer stores aux=3");
            run<3>(A, W, C, M, N, K, "buffer stores aux=17");
            run<4>(A, W, C, M, N, K, "buffer stores aux=2");
        }
    return 0;
}
