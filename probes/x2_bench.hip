// The ViT-Base encoder's four GEMM shapes (M = 256 images x 589 tokens) with the encoder's own epilogues on three kernels:
//   pp/staged  gemm_pp_kernel<Epi, false>  256x256 tile, one 8-wave workgroup per CU, accumulators staged through LDS for the epilogue
//   pp/direct  gemm_pp_kernel<Epi, true>   the same main loop on the transposed tile, epilogue straight from the accumulators
//   x2         probes/rejected/gemm_x2.h   two 4-wave workgroups per CU with 128x256 tiles (epilogue of one behind the MFMAs of the other)
// TFLOP/s of each, outputs compared bit for bit against pp/staged, per-tile stamps (pp: block 0; x2: two workgroups that share a CU).    hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=on [-DX2_MODE=1|2] -o /tmp/x2_bench probes/x2_bench.hip
#define X2_STAMPS 1
#define PP_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
#include <algorithm>
#include "rejected/gemm_x2.h"
using namespace txo;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned rng_state = 12345u;
static inline unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static void fill_bf16(bf16* d, size_t n, float scale) {          // random sign, magnitude ~ scale * [0.5, 2)
    std::vector<unsigned short> h(n);
    const unsigned short base = (unsigned short)(((127 + (int)std::lround(std::log2(scale)) - 1) & 0xff) << 7);
    for (auto& x : h) { const unsigned r = rnd(); x = (unsigned short)(((r & 1) << 15) | (base + (r >> 1 & 0xff))); }
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}
static void fill_f32(float* d, size_t n, float lo, float hi) {
    std::vector<float> h(n);
    for (auto& x : h) x = lo + (hi - lo) * (rnd() & 0xffff) / 65535.f;
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}
__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, size_t n, unsigned long long* cnt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}
static unsigned long long count_diff(const void* a, const void* b, size_t bytes) {
    static unsigned long long* d = nullptr;
    if (!d) CK(hipMalloc(&d, 8));
    CK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, bytes / 4, d);
    unsigned long long h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    return h;
}
template <class F> static float time_ms(F f, int it) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / it;
}
static void dump_stamps(int nk) {
    static unsigned long long h[1024 * 64];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_x2_dbg), sizeof(h)));
    std::map<unsigned long long, std::vector<int>> by_cu;
    for (int b = 0; b < 512; ++b) {
        const unsigned long long id = h[b * 64];
        const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32);
        // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
        by_cu[((unsigned long long)xcc << 16) | (hw & 0xff00)].push_back(b);
    }
    int pairs = 0, shown = 0;
    for (auto& kv : by_cu) pairs += kv.second.size() == 2;
    printf("  placement: %zu distinct CUs hold the 512 workgroups, %d of them exactly two\n", by_cu.size(), pairs);
    for (auto& kv : by_cu) {
        if (kv.second.size() != 2 || shown++ >= 2) continue;
        const unsigned long long t0 = std::min(h[kv.second[0] * 64 + 1], h[kv.second[1] * 64 + 1]);
        for (int b : kv.second) {
            printf("  wg %3d (cu key %llx):", b, kv.first);
            for (int seq = 0; seq < 6; ++seq) {
                const unsigned long long* d = h + b * 64 + 1 + seq * 3;
                printf(" | t=%6.2f loop %5.2f epi %5.2f", (d[0] - t0) * 0.01, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01);
            }
            printf("\n");
        }
    }
    (void)nk;
}


static void dump_pp_stamps(const char* what) {
    static unsigned long long h[4096];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_dbg), sizeof(h)));
    for (int seq = 2; seq < 5; ++seq) for (int wr = 0; wr < 2; ++wr) {
        const unsigned long long* d = h + (seq * 2 + wr) * 4;
        const unsigned long long* p = h + ((seq - 1) * 2 + wr) * 4;
        printf("    %s tile %d group %d: since prev epilogue end %.2f us | loop %.2f | catch-up %.2f | epilogue %.2f\n", what, seq, wr,
               (d[0] - p[3]) * 0.01, (d[1] - d[0]) * 0.01, (d[2] - d[1]) * 0.01, (d[3] - d[2]) * 0.01);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 150784;
    const int it = argc > 2 ? atoi(argv[2]) : 5;
    const bool with_x2 = argc > 3 ? atoi(argv[3]) != 0 : false;
    const int D = 768, I = 768, F = 3072, heads = 12, ntok = 589;
    bf16 *Az, *Ah, *W, *out[3];
    float *y[3], *yinit, *stats, *gb, *bias;
    CK(hipMalloc(&Az, (size_t)M * D * 2)); CK(hipMalloc(&Ah, (size_t)M * F * 2)); CK(hipMalloc(&W, (size_t)2 * F * D * 2));
    for (int v = 0; v < 3; ++v) { CK(hipMalloc(&out[v], (size_t)M * 2 * F * 2)); CK(hipMalloc(&y[v], (size_t)M * D * 4)); }
    CK(hipMalloc(&yinit, (size_t)M * D * 4));
    CK(hipMalloc(&stats, (size_t)M * 8)); CK(hipMalloc(&gb, 2 * D * 4)); CK(hipMalloc(&bias, 2 * F * 4));
    fill_bf16(Az, (size_t)M * D, 1.0f); fill_bf16(Ah, (size_t)M * F, 0.5f); fill_bf16(W, (size_t)2 * F * D, 0.03f);
    fill_f32(yinit, (size_t)M * D, -2.f, 2.f); fill_f32(stats, (size_t)M * 2, 0.5f, 1.5f); fill_f32(gb, 2 * D, 0.8f, 1.2f); fill_f32(bias, 2 * F, -0.1f, 0.1f);
    const int ctv = getenv("X2_CT") ? atoi(getenv("X2_CT")) : 0;
    const char* names[3] = {"pp/staged", "pp/direct", "x2"};
    // run(v, epi_for_variant): variant 0 / 1 / 2
    auto bench = [&](const char* what, const bf16* A, int N, int K, auto make_epi, void** bufs, size_t out_bytes, bool inplace) {
        const double fl = 2.0 * M * N * K;
        const int nv = with_x2 ? 3 : 2;
        auto launch = [&](int v) {
            auto e = make_epi(v);
            if (v == 0) launch_gemm_pp(0, A, W, M, N, K, e, 0);
            else if (v == 1) launch_gemm_pp(0, A, W, M, N, K, e, 1);
            else launch_gemm_x2(0, A, W, M, N, K, e, ctv);
        };
        unsigned long long nd[3] = {0, 0, 0};
        for (int v = 0; v < nv; ++v) {           // one launch each from identical inputs, compared with variant 0
            if (inplace) CK(hipMemcpy(bufs[v], yinit, out_bytes, hipMemcpyDeviceToDevice)); else CK(hipMemset(bufs[v], 0x5a + v, out_bytes));
            launch(v); CK(hipDeviceSynchronize());
            if (v) nd[v] = count_diff(bufs[0], bufs[v], out_bytes);
        }
        float ms[3] = {0, 0, 0};
        for (int rep = 0; rep < 2; ++rep)        // interleaved rounds in one process; keep the faster
            for (int v = 0; v < nv; ++v) { const float t = time_ms([&] { launch(v); }, it); ms[v] = (rep == 0 || t < ms[v]) ? t : ms[v]; }
        printf("%-20s N=%4d K=%4d |", what, N, K);
        for (int v = 0; v < nv; ++v) printf(" %s %.3f ms = %5.0f TF (diff %llu) |", names[v], ms[v], fl / ms[v] / 1e9, nd[v]);
        printf(" direct/staged time %.3f\n", ms[1] / ms[0]);
        launch(0); CK(hipDeviceSynchronize()); dump_pp_stamps(names[0]);
        launch(1); CK(hipDeviceSynchronize()); dump_pp_stamps(names[1]);
        fflush(stdout);
    };
    void* ob[3] = {out[0], out[1], out[2]};
    void* yb[3] = {y[0], y[1], y[2]};
    // FFN-in + GeGLU (attention.py:15-17)
    bench("FFN-in  EpiGeglu", Az, 2 * F, D, [&](int v) { return EpiGeglu<bf16>{out[v], bias, F, 0}; }, ob, (size_t)M * F * 2, false);
    // q/k/v + head-major scatter (attention.py:124-127)
    bench("q/k/v   EpiHeads", Az, 3 * I, D, [&](int v) { return EpiHeads<bf16>{out[v], (size_t)M * I, I, heads, ntok, 0}; }, ob, (size_t)3 * M * I * 2, false);
    // gated output projection + residual (attention.py:96-99,180)
    bench("out-proj EpiGluRes", Az, 2 * D, I, [&](int v) { return EpiGluRes<true>{y[v], ResidLN{y[v], stats, gb, D}, bias, 0}; }, yb, (size_t)M * D * 4, true);
    // FFN-out + residual (attention.py:63-67)
    bench("FFN-out EpiBiasRes", Ah, D, F, [&](int v) { return EpiBiasRes{y[v], ResidLN{y[v], stats, gb, D}, bias, 0}; }, yb, (size_t)M * D * 4, true);
    // plain store
    bench("plain   EpiStore", Az, 2 * F, D, [&](int v) { return EpiStore<bf16>{out[v], 2 * F, nullptr, 0}; }, ob, (size_t)M * 2 * F * 2, false);
    if (with_x2) dump_stamps(D / 32);
    return 0;
}
