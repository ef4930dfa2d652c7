// bf16 GEMM for the encoder-side projections with the epilogue of one output tile hidden behind the matrix work of another:
//   C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), A and W plain row-major bf16 -- same operations and epilogue functors as
//   gemm_big.h / gemm_pp.h (reference model/attention.py:124-127 q/k/v, :96-99,180 gated out-proj, :15-17 GeGLU FFN-in,
//   :63-67 FFN-out).
//
// Why a third GEMM kernel.  gemm_pp.h (256 x 256 tile, ONE 8-wave workgroup per CU) reaches 1.38 PFLOP/s with its stores
// suppressed and 0.85-1.07 with them: at K = 768 a tile has 12 K steps of 64 and then ALL eight waves of the CU sit in the
// epilogue (bias / GLU / GeGLU / residual + LayerNorm rebuild, 64-256 KB through a store path of ~16 B/clk per CU, and the
// next tile's counted vmcnt waits queue behind those stores) while no MFMA runs: 7-12 us of a 25-30 us tile round.
// Here TWO independent 4-wave workgroups live on every CU (80 KiB of LDS and 256 registers each: one wave of each on every
// SIMD), each with its own 128 x 256 output tile and its own operand ring.  While one workgroup is in its epilogue the other
// owns the matrix pipe; when both are in their K loops the hardware arbiter interleaves them as it did the two wave groups
// of gemm_pp.  Price: a 128 x 256 tile stages (128 + 256) rows per K step for half the MFMA work of (256 + 256) rows, i.e.
// 1.5x the L2 -> LDS bytes per FLOP (DESIGN.md section 5 prices this against the measured L2 -> LDS rate).
//
// Structure of one workgroup:
//   * 256 threads = 4 waves side by side (1 x 4), wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles (128 accumulator registers),
//     exactly the wave tile of gemm_pp -- the epilogue functors see the same (row, 8-column) segments;
//   * K in 32-element stages (64-byte rows): one MFMA k-chunk per stage, 12 fragment reads and 32 MFMAs per wave and stage;
//     ring of THREE stages of 24 KiB (A 8 KiB + W 16 KiB); operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     one wave instruction = 16 rows x 64 B written linearly), two stages ahead of the MFMAs; counted s_waitcnt vmcnt(6) +
//     ONE raw s_barrier per stage (the barrier both publishes stage g and frees the slot of stage g - 1 for stage g + 2);
//   * 64-byte LDS rows: piece p of row r sits at 16-byte slot p ^ ((-(r >> 2)) & 3); the permutation is applied on the DMA's
//     SOURCE address; with it every ds_read_b128 lane group of a fragment read touches all 64 banks once (conflict free);
//   * persistent workgroups (two per CU); the stages of a workgroup's successive output tiles form one DMA stream; banded,
//     XCD-aware tile order as gemm_pp;
//   * epilogue: wave-private 2 KiB of LDS, a 16 x 64 accumulator row tile goes through it in two 32-column halves (lower
//     half-wave reads the first, upper half-wave the second), then all 64 lanes run the epilogue functor on 8-column segments.
// Accumulation order per output element: k ascending in 32-element MFMA chunks = gemm_big_kernel<bf16> and gemm_pp_kernel:
// bit-identical results (tests compare the three).
// Requires K % 32 == 0, K >= 96, N % 256 == 0, M >= 128; rows are ragged (loads clamp to the last row, stores are masked).
// Bound: MFMA (bf16 dense peak 2.5 PFLOP/s) -- or the L2 -> LDS path at 1.5x the bytes of the 256 x 256 tile.
#pragma once
#include "../../texocr_amd/csrc/common.h"
#include "../../texocr_amd/csrc/gemm_big.h"
#include "../../texocr_amd/csrc/gemm_pp.h"

namespace txo {

constexpr int X2_BM = 128, X2_BN = 256, X2_BK = 32, X2_THREADS = 256, X2_STAGES = 3;
constexpr int X2_A_BYTES = X2_BM * 64, X2_W_BYTES = X2_BN * 64;          // 64-byte rows
constexpr int X2_STAGE_BYTES = X2_A_BYTES + X2_W_BYTES;                  // 24 KiB
constexpr int X2_RING_BYTES = X2_STAGES * X2_STAGE_BYTES;                // 72 KiB
constexpr int X2_SMEM_BYTES = X2_RING_BYTES + 4 * 2048;                  // + 2 KiB of epilogue staging per wave = 80 KiB: two workgroups per CU
constexpr int X2_DMA_PER_STAGE = 6;                                      // wave instructions per wave and stage: 2 (A) + 4 (W)

#ifdef X2_STAMPS
// probes/x2_bench.hip: per workgroup {hw id, then per tile: loop start, loop end, epilogue end} (100 MHz real-time counter)
__device__ unsigned long long g_x2_dbg[1024 * 64];
#define X2_STAMP(slot) do { if (lane == 0 && wave == 0 && seq < 20) g_x2_dbg[blockIdx.x * 64 + 1 + seq * 3 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define X2_STAMP(slot) do {} while (0)
#endif

// X2_MODE (probe builds only): 0 = the kernel; 1 = no MFMAs (what the DMA stream + fragment reads alone take); 2 = no DMA after the prologue (MFMA + LDS reads alone)
#ifndef X2_MODE
#define X2_MODE 0
#endif

template <class Epi>
__global__ __launch_bounds__(X2_THREADS, 2) void gemm_x2_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                                int K, int tiles_n, int n_tiles, int ct, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];     // ONE array: [stage][A|W][row*64] + epilogue staging

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               // = column quarter of the tile
    const int lr = lane & 15, lg = lane >> 4;
#ifdef X2_STAMPS
    if (tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_x2_dbg[blockIdx.x * 64] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif

    // ---- persistent workgroups, banded XCD-aware tile order (gemm_pp.h): 8 * chunks workgroups; the chunks workgroups of one XCD
    // work on consecutive tiles of a band of `ct` column tiles, so the band's W panels stay in that XCD's L2 while A row panels stream
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, chunks = gridDim.x >> 3;      // gridDim is a multiple of 8
    auto tile_seq = [&](int seq) { return (seq * 8 + xcd) * chunks + slot; };
    int n_my = 0;
    while (tile_seq(n_my) < n_tiles) ++n_my;
    n_my = __builtin_amdgcn_readfirstlane(n_my);                                         // (keeps the loop control below in scalar registers)
    if (n_my == 0) return;
    const int tiles_m = n_tiles / tiles_n, band_tiles = tiles_m * ct;
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        const int L = tile_seq(seq);
        const int band = L / band_tiles, k = L - band * band_tiles;
        const int cols = min(ct, tiles_n - band * ct);                                   // the last band may be narrower
        const int row = k / cols, col = k - row * cols;
        m0 = row * X2_BM; n0 = (band * ct + col) * X2_BN;
    };

    // ---- DMA source offsets (elements).  One wave instruction covers 16 rows x 64 B: lane -> (row lane >> 2, slot lane & 3), and the
    // slot holds source piece slot ^ ((-(row >> 2)) & 3).  Wave w moves rows j * 64 + w * 16 .. + 15 of A (j = 0, 1) and of W (j = 0 .. 3).
    const int drow = lane >> 2, dswz = ((lane & 3) ^ ((-(lane >> 4)) & 3)) * 8;
    int aoff[2], woff0;
    auto set_off = [&](int seq) {
        int m0, n0; tile_origin(seq, m0, n0);
#pragma unroll
        for (int j = 0; j < 2; ++j) aoff[j] = min(m0 + j * 64 + wave * 16 + drow, M - 1) * K + dswz;
        woff0 = (n0 + wave * 16 + drow) * K + dswz;
    };
    auto issue_stage = [&](int kt, int sl) {
        unsigned char* base = lds + sl * X2_STAGE_BYTES + wave * 16 * 64;
        const int k0 = kt * X2_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(A + aoff[j] + k0, base + j * 64 * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(W + woff0 + j * 64 * K + k0, base + X2_A_BYTES + j * 64 * 64);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop over the stream of stages g = 0 .. n_my * nk - 1.  Per stage:
    //   s_waitcnt vmcnt(6)   this wave's DMA of stage g has landed (only stage g + 1's six are still out)
    //   s_barrier            everybody's has; everybody has finished reading stage g - 1 (reads are drained before the MFMAs)
    //   DMA of stage g + 2 into the slot of stage g - 1
    //   12 fragment reads of stage g, 32 MFMAs
    const int nk = K / X2_BK, total = n_my * nk;
    int dkt = 0, dseq = 0;                                                   // the stage the DMA stream issues next
    set_off(0);
    auto dma_next = [&](int g2) {
        issue_stage(dkt, g2 % X2_STAGES);
        if (++dkt == nk) { dkt = 0; ++dseq; if (dseq < n_my) set_off(dseq); }
    };
    dma_next(0);
    dma_next(1);                                                             // total >= nk >= 3

    // fragment read offsets: row-tile t of an operand at t * 1024 + foff
    const int foff = lr * 64 + ((lg ^ ((-(lr >> 2)) & 3)) << 4);
    float* stage = reinterpret_cast<float*>(lds + X2_RING_BYTES + wave * 2048);     // [16][32] f32, 16-column halves XOR-swizzled
    auto sidx = [](int row, int col) { return row * 32 + (col ^ (((row >> 2) & 1) << 4)); };

    int g = 0, sl = 0;
    for (int seq = 0; seq < n_my; ++seq) {
        X2_STAMP(0);
        for (int kt = 0; kt < nk; ++kt, ++g) {
            if (g + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#if X2_MODE != 2
            if (g + 2 < total) dma_next(g + 2);
#else
            if (g + 2 < total && g < 1) dma_next(g + 2);
#endif
            const unsigned char* la = lds + sl * X2_STAGE_BYTES + foff;
            const unsigned char* lw = la + X2_A_BYTES + wave * 64 * 64;
            u32x4 fa[8], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = ld16(lw + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = ld16(la + i * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the slot may be refilled after the next barrier
#if X2_MODE != 1
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16<bf16>(acc[i][j], fa[i], fb[j]);
            __builtin_amdgcn_s_setprio(0);
#else
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(fa[i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(fb[j]));
#endif
            sl = (sl + 1 == X2_STAGES) ? 0 : sl + 1;
        }
        X2_STAMP(1);

        // ---- epilogue of this workgroup's tile (the CU's other workgroup keeps the matrix pipe busy meanwhile)
        {
            int m0, n0; tile_origin(seq, m0, n0);
            const int nbase = n0 + wave * 64;
            const int half = lane >> 5, hl = lane & 31;                      // half-wave `half` takes the row tile's columns half * 32 .. + 31
            float cb[32];
            int cv, nv;                                                      // paired: value columns of this lane; plain: its 8 columns
            if constexpr (Epi::PAIRED) {
                cv = (hl & 1) * 8; nv = nbase + half * 32 + cv;
                epi.cols(nv, nv + 16, cb);
            } else {
                cv = (hl & 3) * 8; nv = nbase + half * 32 + cv;
                epi.cols(nv, cb);
            }
            const int jout = Epi::PAIRED ? (nv >> 5) * 16 + (nv & 15) : nv;
            constexpr int ITEMS = Epi::PAIRED ? 1 : 2, RSHIFT = Epi::PAIRED ? 1 : 2;
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                    // row tile i of the wave: 16 rows x 64 columns
                const int mbase = m0 + i * 16;
                float rr[ITEMS][10];
                if constexpr (Epi::HAS_ROW) {
#pragma unroll
                    for (int q = 0; q < ITEMS; ++q) epi.rowop(min(mbase + (hl >> RSHIFT) + 8 * q, M - 1), jout, rr[q]);
                }
                float v[ITEMS][8], gt[8];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stage[sidx(lg * 4 + r, jj * 16 + lr)] = acc[i][p * 2 + jj][r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private region: no barrier needed
                    if (half == p) {
                        if constexpr (Epi::PAIRED) {
                            load8(&stage[sidx(hl >> 1, cv)], v[0]);
                            load8(&stage[sidx(hl >> 1, cv + 16)], gt);
                        } else {
#pragma unroll
                            for (int q = 0; q < ITEMS; ++q) load8(&stage[sidx((hl >> 2) + 8 * q, cv)], v[q]);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the region is overwritten
                }
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const int m = mbase + (hl >> RSHIFT) + 8 * q;
                    if constexpr (Epi::PAIRED) epi.fin(min(m, M - 1), jout, v[q], gt, cb, rr[q], m < M);
                    else epi.fin(min(m, M - 1), jout, v[q], cb, rr[q], m < M);
                }
                if constexpr (Epi::HAS_ROW) __builtin_amdgcn_sched_barrier(0);   // keep the row tiles' residual loads from piling up (registers)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        X2_STAMP(2);
    }
}

inline bool gemm_x2_fits(int M, int N, int K) {
    return K % X2_BK == 0 && K >= 3 * X2_BK && N % X2_BN == 0 && M >= X2_BM && (long long)M * K < (1ll << 31) && (long long)N * K < (1ll << 31);
}

template <class Epi>
inline void launch_gemm_x2(hipStream_t s, const bf16* A, const bf16* W, int M, int N, int K, Epi epi, int ct_override = 0) {
    const int tiles_m = (M + X2_BM - 1) / X2_BM, tiles_n = N / X2_BN;
    const int n_tiles = tiles_m * tiles_n;
    static int cus_of[64] = {0}; static bool attr_of[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = (dev >= 0 && dev < 64) ? dev : 0;
    if (!cus_of[di] || di != dev) {
        hipDeviceProp_t prop;
        cus_of[di] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int cus = cus_of[di];
    if (!attr_of[di] || di != dev) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x2_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, X2_SMEM_BYTES);
        attr_of[di] = true;
    }
    // band width as gemm_pp: all column tiles when W stays in an XCD's L2 anyway (or K is long), else ~1.6 MB of W per band
    const long long w_bytes = (long long)N * K * 2, coltile_bytes = (long long)X2_BN * K * 2;
    int ct = tiles_n;
    if (w_bytes > (5ll << 19) && K < 2048) ct = (int)std::max<long long>(1, std::min<long long>(tiles_n, (13ll << 17) / coltile_bytes));
    if (ct_override > 0) ct = std::max(1, std::min(tiles_n, ct_override));
    const int grid = ((std::min(n_tiles, 2 * cus) + 7) / 8) * 8;             // two workgroups per CU
    hipLaunchKernelGGL((gemm_x2_kernel<Epi>), dim3(grid), dim3(X2_THREADS), X2_SMEM_BYTES, s, A, W, M, N, K, tiles_n, n_tiles, ct, epi);
}

}  // namespace txo
