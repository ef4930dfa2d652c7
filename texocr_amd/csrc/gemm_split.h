// fp32 GEMM on the bf16 matrix pipe by operand splitting ("bf16x3"):  C[M,N] = A[M,K] * W[N,K]^T, A and W fp32 in memory.
//
// Every operand value x is split into hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo + O(2^-17 |x|)), and a product is taken as
//     a * w  ~=  a_lo * w_hi  +  a_hi * w_lo  +  a_hi * w_hi          (the dropped a_lo * w_lo term is O(2^-16) of the product)
// -- three v_mfma_f32_16x16x32_bf16 per 32 k instead of eight v_mfma_f32_16x16x4_f32, i.e. 5.3x less matrix-pipe time at fp32 accumulation,
// with a relative error per product of ~2^-16 (measured against the exact-f32 kernel: tests/test_gpu_parity.py).
//
// Used for ONE thing: the ResNetV2 backbone of the default factory's hybrid embedder (reference model/resnet.py:38-66,143-149,200-254; conv.h)
// inside the bf16 engine.  That backbone cannot be stored or multiplied in bf16 -- with random weights a 2^-9 perturbation anywhere moves its
// output by 10-20 % (engine.hip: bk_fp32) -- and as exact-f32 MFMA GEMMs its 45 convolutions were 14 of the 21 ms of a 64-image encode.  The fp32
// parity engine keeps the exact-f32 kernel (gemm_big.h): its results are pinned bit for bit.
//
// Same tiling as gemm_big_kernel<float>: 128 x 128 block tile, 4 waves as 2 x 2, 64 x 64 per wave, K in stages of 32 elements (128 B of fp32 per
// row), register-staged with the next stage's loads in flight behind the MFMAs.  The split happens once per staged element, on its way into
// LDS: four planes per buffer (A_hi | A_lo | W_hi | W_lo) of [128 rows][64 B], 16-byte pieces XOR-swizzled with row bits 1-2.
// Same A loaders (LoadPlain / LoadConv / LoadStem of float) and epilogue functors as gemm_big.h.  Bound: MFMA bf16 / VALU (the splits).
#pragma once
#include "common.h"
#include "gemm_big.h"

namespace txo {

// 4 floats -> {4 x hi, 4 x lo} as packed bf16 pairs.  hi = round-to-nearest-even bf16 of x; lo = bf16 of the exact remainder x - hi.
__device__ inline void split4(const u32x4& x, uint2& hi, uint2& lo) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    auto pk = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v)); };
    const float x0 = __uint_as_float(x.x), x1 = __uint_as_float(x.y), x2 = __uint_as_float(x.z), x3 = __uint_as_float(x.w);
    hi.x = pk(x0, x1); hi.y = pk(x2, x3);
    const float h0 = __uint_as_float(hi.x << 16), h1 = __uint_as_float(hi.x & 0xffff0000u);
    const float h2 = __uint_as_float(hi.y << 16), h3 = __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pk(x0 - h0, x1 - h1); lo.y = pk(x2 - h2, x3 - h3);
}

// GroupNorm statistics from the epilogue (conv.h: every convolution of the backbone is followed by GroupNorm(32) over its output): while the
// output tile sits in LDS, the sum and the sum of squares of every (image, group) it covers are written to a slot of `part`
//     part[((tile_m * 2 + row half) * 2 + which) * 32 + group] = {sum, sum of squares}      which = 0: the image of the tile's first row, 1: the next
// (rows = pixels, a tile of 128 rows covers at most two images when HW >= 128).  Every slot is written by exactly one thread and
// gn_finish_tiles_kernel adds them in a fixed order: deterministic, no float atomics -- and the separate pass over the convolution's output
// (gn_partial_kernel: 2.4 of the backbone's 12.6 ms) is gone.  part == nullptr: off.
struct GnPart { float* part; int HW, cpg; };

template <class ALoad, class Epi>
__global__ __launch_bounds__(GB_THREADS) void gemm_split_kernel(ALoad aload, const float* __restrict__ W, int M, int N, int K, int tiles_n,
                                                                int n_tiles, Epi epi, GnPart gn) {
    constexpr int STAGE_K = 32;                                       // fp32 elements per row and stage = one 16x16x32 MFMA k-step
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][4][GB_BM * 64];   // [buf][A_hi | A_lo | W_hi | W_lo][row * 64 B]

    int bid = blockIdx.x;                                             // XCD-aware tile order, as gemm_big_kernel
    {
        int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * GB_BM, n0 = tile_n * GB_BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 15, lg = lane >> 4;

    // staging map: thread loads rows srow + 32 i (i < 4), floats spiece * 4 .. + 3 of the stage's 32
    const int srow = tid >> 3, spiece = tid & 7;
    u32x4 ra[4], rw[4];
    const u32x4 zero = {0u, 0u, 0u, 0u};

    auto load_stage = [&](int kt) {
        const int k = kt * STAGE_K + spiece * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + srow + 32 * i, n = n0 + srow + 32 * i;
            ra[i] = (m < M) ? aload(m, k) : zero;
            rw[i] = (n < N) ? ld16(W + (size_t)n * K + k) : zero;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i, off = row * 64 + (((spiece >> 1) ^ ((row >> 1) & 3)) << 4) + (spiece & 1) * 8;
            uint2 h, l;
            split4(ra[i], h, l);
            *reinterpret_cast<uint2*>(&lds[buf][0][off]) = h; *reinterpret_cast<uint2*>(&lds[buf][1][off]) = l;
            split4(rw[i], h, l);
            *reinterpret_cast<uint2*>(&lds[buf][2][off]) = h; *reinterpret_cast<uint2*>(&lds[buf][3][off]) = l;
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / STAGE_K;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_stage(kt + 1);
        u32x4 ah[4], al[4], wh[4], wl[4];                             // lane (lr, lg): row lr of the 16-row tile, k = 8 lg .. 8 lg + 7
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // 16-byte piece lg of a row sits at piece lg ^ ((row >> 1) & 3): the four 16-lane groups of a ds_read_b128 then hit 16 different
            // 16-byte bank slots each (rows are 64 B: without it rows r and r + 4 share their slots)
            const int sw = ((lg ^ ((lr >> 1) & 3)) << 4);
            const int ao = (wm + 16 * i + lr) * 64 + sw, wo = (wn + 16 * i + lr) * 64 + sw;
            ah[i] = ld16(&lds[buf][0][ao]); al[i] = ld16(&lds[buf][1][ao]);
            wh[i] = ld16(&lds[buf][2][wo]); wl[i] = ld16(&lds[buf][3][wo]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                             // the two small terms first, the large one last
                mma16<bf16>(acc[i][j], al[i], wh[j]);
                mma16<bf16>(acc[i][j], ah[i], wl[j]);
                mma16<bf16>(acc[i][j], ah[i], wh[j]);
            }
        if (kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: as gemm_big_kernel (accumulators -> LDS tile [128][128] f32 = the 64 KB of the staging buffers -> 8-column row segments)
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]);
    auto tidx = [](int row, int col) { return row * GB_BN + (col ^ (((row >> 2) & 1) << 4)); };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[tidx(wm + 16 * i + lg * 4 + r, wn + 16 * j + lr)] = acc[i][j][r];
    __syncthreads();
    static_assert(!Epi::PAIRED, "the backbone's epilogues are plain stores");
    if (gn.part) {
        // thread = (column c of the tile, row half rh): 64 rows of one output channel, split at the image boundary
        const int c = tid & 127, rh = tid >> 7;
        const int mlim = min(GB_BM, M - m0);                           // rows of the tile that exist
        const int split = min(mlim, (m0 / gn.HW + 1) * gn.HW - m0);    // rows [0, split): the first image; [split, mlim): the next
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        for (int r = rh * 64; r < rh * 64 + 64; ++r) {
            const float v = tile[tidx(r, c)];
            const bool in0 = r < split, in1 = r >= split && r < mlim;
            s0 += in0 ? v : 0.f; q0 = in0 ? fmaf(v, v, q0) : q0;
            s1 += in1 ? v : 0.f; q1 = in1 ? fmaf(v, v, q1) : q1;
        }
        // the cpg channels of a group are cpg adjacent lanes (cpg = 2 .. 32, a power of two)
        for (int o = 1; o < gn.cpg; o <<= 1) {
            s0 += __shfl_xor(s0, o, 64); q0 += __shfl_xor(q0, o, 64); s1 += __shfl_xor(s1, o, 64); q1 += __shfl_xor(q1, o, 64);
        }
        const int n = n0 + c;
        if ((c & (gn.cpg - 1)) == 0 && n < N) {
            float* dst = gn.part + ((((size_t)tile_m * 2 + rh) * 2) * 32 + n / gn.cpg) * 2;
            dst[0] = s0; dst[1] = q0; dst[64] = s1; dst[65] = q1;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + GB_THREADS * q, row = idx >> 4, cg = idx & 15;
        const int m = m0 + row, n = n0 + cg * 8;
        if (m < M && n + 7 < N) {
            float v[8];
            load8(&tile[tidx(row, cg * 8)], v);
            epi(m, n, v);
        }
    }
}

inline bool gemm_split_fits(int K) { return K % 32 == 0; }

// per (image, group): mean and 1 / sqrt(var + eps) from the tile slots above, added in tile order (fp64 combine, as conv.h: gn_finish_kernel)
__global__ void gn_finish_tiles_kernel(const float* __restrict__ part, float* __restrict__ stats, int HW, double count) {
    const int b = blockIdx.x, g = threadIdx.x;
    if (g >= 32) return;
    const int t0 = (int)(((long long)b * HW) / GB_BM), t1 = (int)((((long long)b + 1) * HW - 1) / GB_BM);
    double a = 0.0, d = 0.0;
    for (int t = t0; t <= t1; ++t) {
        const int which = ((long long)t * GB_BM) / HW == b ? 0 : 1;
        for (int rh = 0; rh < 2; ++rh) {
            const float* p = part + ((((size_t)t * 2 + rh) * 2 + which) * 32 + g) * 2;
            a += p[0]; d += p[1];
        }
    }
    const double mean = a / count;
    const double var = fmax(d / count - mean * mean, 0.0);
    stats[((size_t)b * 32 + g) * 2 + 0] = (float)mean;
    stats[((size_t)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}
inline bool gn_fusable(int HW, int C) { const int cpg = C / 32; return HW >= GB_BM && C % 32 == 0 && cpg >= 1 && cpg <= 32 && (cpg & (cpg - 1)) == 0; }

template <class ALoad, class Epi>
inline void launch_gemm_split(hipStream_t s, ALoad aload, const float* W, int M, int N, int K, Epi epi, GnPart gn = GnPart{nullptr, 1, 1}) {
    const int tiles_m = (M + GB_BM - 1) / GB_BM, tiles_n = (N + GB_BN - 1) / GB_BN;
    const int n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_split_kernel<ALoad, Epi>), dim3(n_tiles), dim3(GB_THREADS), 0, s, aload, W, M, N, K, tiles_n, n_tiles, epi, gn);
}

}  // namespace txo
