// bf16 GEMM for the large encoder-side projections:  C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), A and W plain row-major bf16.
//
// Same operations and epilogues as gemm_big.h (reference model/attention.py:124-127 q/k/v, :96-99,180 gated out-proj,
// :15-17 GeGLU FFN-in, :63-67 FFN-out, :125-126 cross K/V), restructured for the MFMA-bound regime (ViT-Base, BASELINE
// configs[3]) where the 128x128 register-staged kernel reaches ~0.6 PFLOP/s:
//   * 256 x 256 block tile, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles
//     (128 accumulator registers) -- twice the MFMA work per byte staged and per LDS fragment read;
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).  One wave
//     instruction writes 1 KiB = 8 rows x 128 B linearly; the XOR piece swizzle that keeps the ds_read_b128 fragment reads
//     conflict free is applied on the SOURCE side (lane (row, slot) fetches piece slot ^ (row & 7));
//   * K in 64-element (128-byte) tiles, two LDS buffers of 64 KiB managed as half-tiles; DMA runs 1 - 1.5 tiles ahead of the
//     MFMAs: counted s_waitcnt vmcnt + raw s_barrier (a __syncthreads() would drain the DMA queue);
//   * one block per CU (128 KiB LDS), two waves per SIMD, and the two wave groups run their phases one barrier apart: while
//     one wave of a SIMD issues its 16 MFMAs the other reads fragments and issues DMA (main-loop comment below);
//   * epilogue: each wave stages its own 64 x 64 accumulator quarter through LDS (no block barrier) and hands 8-column
//     row segments to the same epilogue functors as gemm_big.h.
// Accumulation order per output element is the same as gemm_big_kernel<bf16> (k ascending in 32-element MFMA chunks),
// so both kernels produce bit-identical results -- tests compare them.
// Requires K % 64 == 0 and N % 256 == 0 (every encoder GEMM of the reference configurations); rows are ragged (M is
// B * tokens): loads clamp to the last row, stores are masked.
// Bound: MFMA (bf16 dense peak 2.5 PFLOP/s).
#pragma once
#include "common.h"
#include "gemm_big.h"

namespace txo {

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_THREADS = 512;
constexpr int PP_TILE_BYTES = PP_BM * 128;            // one operand tile: 256 rows x 128 B
constexpr int PP_BUF_BYTES = 2 * PP_TILE_BYTES;       // A + W
constexpr int PP_LDS_BYTES = 2 * PP_BUF_BYTES;        // two buffers = 128 KiB

// LDS-DMA of 16 bytes per lane: LDS destination = lds_base (wave-uniform) + lane * 16
__device__ inline void dma16(const void* gsrc, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)gsrc,
                                     (__attribute__((address_space(3))) void*)(uint32_t)(uintptr_t)lds_base, 16, 0, 0);
}

template <class Epi>
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_pp_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                                int K, int tiles_n, int n_tiles, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];     // ONE array: [buf][A|W][row*128]

    // XCD-aware tile order (bijective): blocks that share an XCD walk consecutive tiles of one A row panel
    int bid = blockIdx.x;
    {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * PP_BM, n0 = tile_n * PP_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;              // waves w and w+4 share a SIMD: same columns, other row half
    const int lr = lane & 15, lg = lane >> 4;

    // ---- DMA source addresses.  Operand tiles are moved as half-tiles of 128 rows x 128 B (16 KiB): 2 wave-instructions
    // per wave, instruction j covers rows j*64 + wave*8 .. +7 of the half.  LDS slot (row, dslot) <- source piece dslot ^ (row & 7).
    const int drow = lane >> 3, dslot = lane & 7;
    const bf16* asrc[2][2]; const bf16* wsrc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = x * 128 + j * 64 + wave * 8 + drow;      // row inside the 256-row tile
            const int piece = dslot ^ (row & 7);
            asrc[x][j] = A + (size_t)min(m0 + row, M - 1) * K + piece * 8;
            wsrc[x][j] = W + (size_t)(n0 + row) * K + piece * 8;
        }
    // half-tile x of operand `op` (0 = A, 1 = W) of K tile t -> buffer buf
    auto issue_half = [&](int op, int x, int t, int buf) {
        unsigned char* base = lds + buf * PP_BUF_BYTES + op * PP_TILE_BYTES + x * (PP_TILE_BYTES / 2) + wave * 8 * 128;
        const bf16* s0 = op ? wsrc[x][0] : asrc[x][0];
        const bf16* s1 = op ? wsrc[x][1] : asrc[x][1];
        dma16(s0 + t * PP_BK, base);
        dma16(s1 + t * PP_BK, base + 64 * 128);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: four phases per K tile, one accumulator quadrant (64 rows x 32 columns x 64 k = 16 MFMAs) each.
    // A phase = memory part {fragment ds_reads, one half-tile of DMA, waits} | barrier | 16 MFMAs | barrier.  The two wave
    // groups (wr = 0 / 1; waves w and w+4 share a SIMD) run one barrier apart, so that on every SIMD one wave issues MFMAs
    // while the other fetches.
    //   reads per phase   P1: W[c0] + A[h0]   P2: W[c1]   P3: A[h1]   P4: none (W[c0] stays in registers)
    //   DMA per phase     P1: A0(t+1)  P2: A1(t+1)  P3: W0(t+2)  P4: W1(t+2)   -- A one tile ahead, W a tile and a half
    // Slot reuse (WAR): every memory part ends with lgkmcnt(0) BEFORE its barrier, so a slot's last reads (A: P3, W: P2) are
    // complete two barriers before the DMA that refills it is issued.  Arrival (RAW): the counted vmcnt at P4 retires
    // everything but the two W halves just issued; both groups have passed that wait before either reads tile t+1.
    const int nk = K / PP_BK;
    issue_half(1, 0, 0, 0); issue_half(1, 1, 0, 0); issue_half(0, 0, 0, 0); issue_half(0, 1, 0, 0);
    issue_half(1, 0, 1, 1); issue_half(1, 1, 1, 1);                       // nk >= 2 (gemm_pp_fits)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                            // stagger the second group by one barrier

    u32x4 fa[4][2], fb0[2][2], fb1[2][2];
    auto mfma_quadrant = [&](int h, const u32x4 (&fb)[2][2], int c) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16<bf16>(acc[h * 4 + i][c * 2 + j], fa[i][ks], fb[j][ks]);
        __builtin_amdgcn_s_setprio(0);
    };
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const unsigned char* la = lds + buf * PP_BUF_BYTES;
        const unsigned char* lw = la + PP_TILE_BYTES;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;
        // ---- P1: (h0, c0)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = ld16(lw + swz128(wc * 64 + j * 16 + lr, ks * 4 + lg));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ld16(la + swz128(wr * 128 + i * 16 + lr, ks * 4 + lg));
        if (more1) issue_half(0, 0, t + 1, buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(0, fb0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- P2: (h0, c1)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = ld16(lw + swz128(wc * 64 + 32 + j * 16 + lr, ks * 4 + lg));
        if (more1) issue_half(0, 1, t + 1, buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(0, fb1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- P3: (h1, c1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ld16(la + swz128(wr * 128 + 64 + i * 16 + lr, ks * 4 + lg));
        if (more2) issue_half(1, 0, t + 2, buf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(1, fb1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- P4: (h1, c0)
        if (more2) { issue_half(1, 1, t + 2, buf); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(1, fb0, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                            // the first group catches up: all LDS reads are done

    // ---- epilogue: per wave, two 64-row halves of its 128 x 64 accumulator tile staged through the wave's own 16 KiB of
    // LDS (no block barrier).  A lane keeps one column group for all its items, so the column operands (bias) are loaded
    // once; the row operands (residual / position rows) of four items are requested together ahead of the arithmetic --
    // one functor call per item would wait for its own loads sixteen times in a row.
    float* stage = reinterpret_cast<float*>(lds + wave * 16384);   // [64][64] f32, 16-column groups XOR-swizzled by row
    auto sidx = [](int row, int col) { return row * 64 + (col ^ (((row >> 2) & 1) << 4)); };
    const int nbase = n0 + wc * 64;
    float cb[16];
    int cv = 0, nv = 0;                                            // paired: value columns of this lane; plain: its 8 columns
    if constexpr (Epi::PAIRED) {
        const int vg = lane & 3;
        cv = (vg >> 1) * 32 + (vg & 1) * 8; nv = nbase + cv;
        epi.cols(nv, nv + 16, cb);
    } else {
        cv = (lane & 7) * 8; nv = nbase + cv;
        epi.cols(nv, cb);
    }
    const int jout = Epi::PAIRED ? (nv >> 5) * 16 + (nv & 15) : nv;
    constexpr int ITEMS = Epi::PAIRED ? 4 : 8, RSTEP = Epi::PAIRED ? 16 : 8, RSHIFT = Epi::PAIRED ? 2 : 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[sidx(i * 16 + lg * 4 + r, j * 16 + lr)] = acc[h * 4 + i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private region: no barrier needed
        const int mbase = m0 + wr * 128 + h * 64;
#pragma unroll
        for (int q0 = 0; q0 < ITEMS; q0 += 4) {
            float rr[4][8];
            if constexpr (Epi::HAS_ROW) {
#pragma unroll
                for (int q = 0; q < 4; ++q) epi.rowop(min(mbase + (lane >> RSHIFT) + RSTEP * (q0 + q), M - 1), jout, rr[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (lane >> RSHIFT) + RSTEP * (q0 + q), m = mbase + row;
                float v[8];
                load8(&stage[sidx(row, cv)], v);
                if constexpr (Epi::PAIRED) {
                    float g[8];
                    load8(&stage[sidx(row, cv + 16)], g);
                    epi.fin(min(m, M - 1), jout, v, g, cb, rr[q], m < M);
                } else {
                    epi.fin(min(m, M - 1), jout, v, cb, rr[q], m < M);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // reads done before the next half overwrites the region
    }
}

// true when the shape fits this kernel; otherwise the caller uses gemm_big_kernel
inline bool gemm_pp_fits(int M, int N, int K) { return K % PP_BK == 0 && K >= 2 * PP_BK && N % PP_BN == 0 && M >= PP_BM; }

template <class Epi>
inline void launch_gemm_pp(hipStream_t s, const bf16* A, const bf16* W, int M, int N, int K, Epi epi) {
    const int tiles_m = (M + PP_BM - 1) / PP_BM, tiles_n = N / PP_BN;
    const int n_tiles = tiles_m * tiles_n;
    static bool attr_set = false;                     // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PP_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp_kernel<Epi>), dim3(n_tiles), dim3(PP_THREADS), PP_LDS_BYTES, s, A, W, M, N, K, tiles_n, n_tiles, epi);
}

}  // namespace txo
