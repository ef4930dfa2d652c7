// microbenchmark of gemm_pp_kernel: time vs K at fixed M, N  ->  per-tile overhead a and per-K-tile cost b
#define PP_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../texocr_amd/csrc/gemm_pp.h"
using namespace txo;
template <typename T> struct EpiNull {            // no stores (unless a value is an impossible one): what does the epilogue cost without them?
    T* out; int ldo; const float* bias;
    static constexpr bool PAIRED = false;
    static constexpr bool HAS_ROW = false;
    __device__ inline void cols(int n, float (&cb)[32]) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) cb[e] = 0.f; }
    __device__ inline void rowop(int, int, float (&)[10]) const {}
    __device__ inline void fin(int m, int n, float (&v)[8], const float (&cb)[32], const float (&)[10], bool valid) const {
        if (valid && v[0] == 12345.678f) store8<T>(out + (size_t)m * ldo + n, v);
    }
};
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 150784, N = argc > 2 ? atoi(argv[2]) : 6144;
    const int Ks[] = {256, 768, 1536, 3072, 6144};
    const int KMAX = 6144;
    bf16 *A, *W, *C;
    hipMalloc(&A, (size_t)M * KMAX * 2); hipMalloc(&W, (size_t)N * KMAX * 2); hipMalloc(&C, (size_t)M * N * 2);
    std::vector<unsigned short> h((size_t)N * KMAX);
    for (auto& x : h) x = 0x3c00 + (rand() & 0x3ff);       // random bf16 in [0.0078, 0.0156)-ish, not zeros
    hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)M * KMAX; off += h.size()) hipMemcpy(A + off, h.data(), std::min(h.size(), (size_t)M * KMAX - off) * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int K : Ks) {
        EpiStore<bf16> epi{C, N, nullptr};
        for (int i = 0; i < 2; ++i) launch_gemm_pp(0, A, W, M, N, K, epi);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int it = 5;
        for (int i = 0; i < it; ++i) launch_gemm_pp(0, A, W, M, N, K, epi);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        const double tiles = ((M + 255) / 256) * (double)(N / 256), rounds = tiles / 256.0;
        printf("M=%d N=%d K=%5d: %.3f ms = %6.0f TFLOP/s | %.2f us per tile-round, %.3f us per K tile\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9,
               ms * 1e3 / rounds, ms * 1e3 / rounds / (K / 64));
    }
    {
        EpiNull<bf16> epi{C, N, nullptr};
        launch_gemm_pp(0, A, W, M, N, 768, epi); hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) launch_gemm_pp(0, A, W, M, N, 768, epi);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("NO STORES K=768: %.3f ms = %.0f TFLOP/s\n", ms, 2.0 * M * N * 768 / ms / 1e9);
        static unsigned long long h[4096];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_dbg), sizeof(h));
        for (int seq = 2; seq < 4; ++seq) for (int wr = 0; wr < 2; ++wr) {
            const unsigned long long* d = h + (seq * 2 + wr) * 4;
            const unsigned long long* p = h + ((seq - 1) * 2 + wr) * 4;
            printf("  no-store tile %d group %d: since prev epilogue end %.2f us | loop %.2f | catch-up %.2f | epilogue %.2f\n", seq, wr,
                   (d[0] - p[3]) * 0.01, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, (d[3] - d[2]) * 0.01);
        }
    }
    {   // stamps of the last run (K = 6144) are overwritten: rerun K = 768 once and dump
        EpiStore<bf16> epi{C, N, nullptr};
        launch_gemm_pp(0, A, W, M, N, 768, epi); hipDeviceSynchronize();
        static unsigned long long h[4096];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_dbg), sizeof(h));
        for (int seq = 2; seq < 8; ++seq) for (int wr = 0; wr < 2; ++wr) {
            const unsigned long long* d = h + (seq * 2 + wr) * 4;
            const unsigned long long* p = h + ((seq - 1) * 2 + wr) * 4;
            printf("tile %d group %d: since prev epilogue end %.2f us | loop %.2f | catch-up %.2f | epilogue %.2f\n", seq, wr,
                   (d[0] - p[3]) * 0.01, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, (d[3] - d[2]) * 0.01);
        }
    }
    return 0;
}
