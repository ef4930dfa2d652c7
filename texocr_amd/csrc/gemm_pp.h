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
//   * persistent blocks (one per CU): the K tiles of a block's successive output tiles form one DMA stream, so the next
//     tile's operands are already landing while the current tile is written out;
//   * two epilogue forms behind the same functors as gemm_big.h (8-column row segments): STAGED -- each wave moves its accumulators
//     through its own 4 KiB of LDS sixteen rows at a time (no block barrier) -- and DIRECT (template parameter TR, r05): the MFMAs
//     compute the transposed tile, the weight rows are permuted on their way into LDS, and a lane then holds 8 contiguous columns of
//     a row in registers: no LDS, no wait between row tiles.  Direct for the epilogues without row operands, staged for the
//     residual ones (comment at the epilogue);
//   * the seam between two tiles (r05): the next tile's second A tile is requested BEFORE the epilogue's stores and the first K
//     step's wait counts them (comment at `early_a`), so the store drain no longer stalls the first two K steps.
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
constexpr int PP_STAGE_BYTES = 8 * 4096;              // epilogue staging, 4 KiB per wave
constexpr int PP_SMEM_BYTES = PP_LDS_BYTES + PP_STAGE_BYTES;   // 160 KiB: the whole LDS of a CU
constexpr int PP_SB_MB = 64;                          // A bytes of a row super-block (tile walk below), MB

// LDS-DMA of 16 bytes per lane: LDS destination = lds_base (wave-uniform) + lane * 16
__device__ inline void dma16(const void* gsrc, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)gsrc,
                                     (__attribute__((address_space(3))) void*)(uint32_t)(uintptr_t)lds_base, 16, 0, 0);
}

// PP_SEAM = 0 builds the kernel without the early request / relaxed count at the tile seam (A/B in probes/x2_bench.hip)
#ifndef PP_SEAM
#define PP_SEAM 1
#endif
#ifdef PP_STAMPS
__device__ unsigned long long g_pp_dbg[4096];      // probes/pp_bench.hip: block 0, waves 0 and 4: {loop start, loop end, epilogue end} per tile
#define PP_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && seq < 64) g_pp_dbg[(seq * 2 + wr) * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_STAMP(slot) do {} while (0)
#endif

// TR: the MFMAs compute the TRANSPOSED tile (weight fragment as the A operand) and the epilogue runs straight from the accumulators,
// no LDS (comment at the epilogue below).  TR = false is the LDS-staged epilogue (kept for A/B: TXO_PP_TR=0, probes/pp_epi_bench.hip).
template <class Epi, bool TR>
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_pp_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N,
                                                                int K, int tiles_n, int n_tiles, int ct, int sbr, int rev, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];     // ONE array: [buf][A|W][row*128] + epilogue staging

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;              // waves w and w+4 share a SIMD: same columns, other row half
    const int lr = lane & 15, lg = lane >> 4;

    // ---- persistent blocks.  Tile order: the output is cut into bands of `ct` column tiles; inside a band tiles run
    // row-major, and 32 consecutive tiles (one per CU of an XCD: blocks b, b+8, ... share an XCD) form the working set of one
    // XCD at a time: `ct` column tiles of W stay in that XCD's 4 MiB L2 for the whole band while 32/ct row panels of A
    // stream through.  With the plain row-panel order an XCD needs ALL of W every round (9.4 MB for the ViT-Base FFN-in:
    // it comes from the Infinity Cache at ~35 GB/s per CU, which is what bounded the main loop at half the MFMA rate).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, chunks = gridDim.x >> 3;      // gridDim is a multiple of 8
    auto tile_seq = [&](int seq) { return (seq * 8 + xcd) * chunks + slot; };            // position in the banded order
    int n_my = 0;
    while (tile_seq(n_my) < n_tiles) ++n_my;
    if (n_my == 0) return;
    // Row super-blocks (r06): with more than one band every band re-reads ALL of A, and at batch 256 an A operand (231 MB at the ViT-Base
    // FFN-in) does not survive in the 256 MB Infinity Cache from one band to the next -- five of its six passes came from HBM.  The bands
    // therefore run inside super-blocks of `sbr` row panels (A of a super-block ~64 MB: the passes after the first are Infinity-Cache hits).
    // The persistent stream runs on across super-blocks, so unlike launching the GEMM per image chunk this costs no tail.
    const int tiles_m = n_tiles / tiles_n, sb_tiles = sbr * tiles_n;
    auto tile_origin = [&](int seq, int& m0, int& n0) {
        int L = tile_seq(seq);
        const int sb = L / sb_tiles; L -= sb * sb_tiles;
        const int band_tiles = min(sbr, tiles_m - sb * sbr) * ct;                        // the last super-block may be shorter
        const int band = L / band_tiles, k = L - band * band_tiles;
        const int cols = min(ct, tiles_n - band * ct);                                   // the last band may be narrower
        int row = k / cols; const int col = k - row * cols;
        row += sb * sbr;
        if (rev) row = tiles_m - 1 - row;                                                // the row panels top down (engine.hip: encode)
        m0 = row * PP_BM; n0 = (band * ct + col) * PP_BN;
    };

    // ---- DMA source offsets (elements).  Operand tiles move as half-tiles of 128 rows x 128 B (16 KiB): 2 wave-instructions
    // per wave, instruction j covers rows j*64 + wave*8 .. +7 of the half.  LDS slot (row, dslot) <- source piece dslot ^ (row & 7).
    // The A stream and the W stream run ahead of the MFMAs by different distances, so each keeps the offsets of the tile
    // it is currently fetching for.
    const int drow = lane >> 3, dslot = lane & 7;
    // The four instructions of a tile cover rows +0 / +64 / +128 / +192 from the lane's first row, and every one of those rows has
    // the same (row & 7), hence the same swizzled piece.  W needs ONE per-lane offset (the others are uniform multiples of K
    // further); A keeps four because its rows clamp to M - 1 in the last row panel.  (Eight precomputed offsets were long-lived
    // VGPRs in a kernel at the 256-register limit: with the larger residual epilogues hipcc spilled them and reloaded them from
    // scratch at every DMA issue -- vector-memory loads in the middle of the counted-vmcnt pipeline.)
    // (set_aoff / set_woff run once per output tile: they rebuild their lane constants from an opaque copy of the thread id instead of
    // keeping five of them live across a K loop that uses all 256 registers -- spilled, they came back through scratch loads that drain the
    // DMA queue)
    int aoff[2][2], woff0 = 0;
    auto set_aoff = [&](int seq) {
        int m0, n0; tile_origin(seq, m0, n0);
        int t = tid; asm volatile("" : "+v"(t));
        const int row0 = t >> 3, swz = ((t & 7) ^ ((t >> 3) & 7)) * 8;               // wave * 8 + drow == tid >> 3
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int j = 0; j < 2; ++j) aoff[x][j] = min(m0 + x * 128 + j * 64 + row0, M - 1) * K + swz;
    };
    // TR: LDS row `s` of a wave's 64-row strip of the W tile holds weight row strip + wperm(s): MFMA column tile (c, j) of the strip --
    // LDS rows c*32 + j*16 + q, read in natural order, conflict free -- then carries, in accumulator row q = 4*lg + r (the lane's r-th register),
    //   plain epilogues:  column c*32 + lg*8 + j*4 + r                 -> lane lg, tiles j = 0, 1 of half c: 8 CONTIGUOUS columns c*32 + lg*8 ..
    //   gated epilogues:  output o = lg*8 + c*4 + r of the strip's 32 (weight rows come as 16 value rows, then their 16 gate rows), its value
    //                     row (j = 0) or gate row (j = 1)              -> lane lg, halves c = 0, 1: outputs lg*8 .. +7, value AND gate
    // The permutation is free: the DMA's source address is per lane.  The four instructions of a tile cover LDS rows +0 / +64 / +128 / +192:
    // whole strips apart, so they still differ by uniform multiples of K.
    auto wperm = [](int s) -> int {
        if constexpr (!TR) return s;
        const int c = s >> 5, j = (s >> 4) & 1, q = s & 15;
        if constexpr (Epi::PAIRED) { const int o = (q >> 2) * 8 + c * 4 + (q & 3); return (o >> 4) * 32 + (o & 15) + j * 16; }
        else return c * 32 + (q >> 2) * 8 + j * 4 + (q & 3);
    };
    auto set_woff = [&](int seq) {
        int m0, n0; tile_origin(seq, m0, n0);
        int t = tid; asm volatile("" : "+v"(t));
        woff0 = (n0 + wperm(t >> 3)) * K + ((t & 7) ^ ((t >> 3) & 7)) * 8;
    };
    // half-tile x of operand `op` (0 = A, 1 = W), K tile kt of the stream's current tile -> buffer buf
    auto issue_half = [&](int op, int x, int kt, int buf) {
        unsigned char* base = lds + buf * PP_BUF_BYTES + op * PP_TILE_BYTES + x * (PP_TILE_BYTES / 2) + wave * 8 * 128;
        const bf16* src = op ? W : A;
        const int o0 = op ? woff0 + x * 128 * K : aoff[x][0], o1 = op ? woff0 + (x * 128 + 64) * K : aoff[x][1];
        dma16(src + o0 + kt * PP_BK, base);
        dma16(src + o1 + kt * PP_BK, base + 64 * 128);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: the K tiles of all this block's output tiles form ONE stream g = 0 .. n_my*nk-1 (the DMA of the next
    // output tile's first K tiles is in flight while the current tile is finished and written out).  Two phases per K
    // tile, half of the wave's accumulators (64 rows x 64 columns x 64 k = 32 MFMAs) each.  A phase = memory part {fragment
    // ds_reads, two half-tiles of DMA, waits} | barrier | 32 MFMAs | barrier.  The two wave groups (wr = 0 / 1; waves w and
    // w+4 share a SIMD) run one barrier apart, so that on every SIMD one wave issues MFMAs while the other fetches.
    //   reads per phase   a: W[c0], W[c1], A[h0]      b: A[h1]
    //   DMA per phase     a: A0(g+1), A1(g+1)         b: W0(g+2), W1(g+2)     -- A one tile ahead, W a tile and a half
    // Slot reuse (WAR): every memory part drains its ds_reads (lgkmcnt(0)) before its barrier; the W slots are refilled in
    // phase b, one barrier after the other group's phase-a reads, the A slots in phase a of the next tile.  Arrival (RAW):
    // the counted vmcnt in phase b retires everything but the two W halves just issued; both groups have passed that wait
    // before either reads tile g+1.  (Four phases of 16 MFMAs each -- twice the barriers -- measured 1.48 us per K tile.)
    const int nk = K / PP_BK, total = n_my * nk;
    set_aoff(0); set_woff(0);
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
                for (int j = 0; j < 2; ++j) {
                    if constexpr (TR) mma16<bf16>(acc[h * 4 + i][c * 2 + j], fb[j][ks], fa[i][ks]);   // lane (lr, lg), register r: row lr, column 4*lg + r of the tile
                    else mma16<bf16>(acc[h * 4 + i][c * 2 + j], fa[i][ks], fb[j][ks]);
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // epilogue staging: 4 KiB per wave behind the two buffers (the buffers hold the next tile's operands by then)
    float* stage = reinterpret_cast<float*>(lds + PP_LDS_BYTES + wave * 4096);   // [16][64] f32, 16-column groups XOR-swizzled
    auto sidx = [](int row, int col) { return row * 64 + (col ^ (((row >> 2) & 1) << 4)); };

    // The seam between two output tiles.  A tile's stores (64-256 KB per CU through a store path of ~16-20 B/clk) are still draining when the
    // next tile's K loop starts, and vector-memory operations retire in issue order: a counted wait for a DMA requested AFTER the stores is a
    // wait for the stores.  So (1) K tile 1 of the next tile's A operand is requested BEFORE the epilogue (its slot, A of the finished tile's
    // last K tile, is free once both groups are past their last reads: after the catch-up barrier) -- every operand of the next tile's first
    // TWO K steps is then older than the stores -- and (2) the first K step's wait counts the stores as well (vmcnt(4 + EPI_STORES): exact when
    // every row of the tile was valid, i.e. every store was executed; otherwise the plain count, which waits for everything).
    constexpr int EPI_STORES = 8 * (Epi::PAIRED ? 1 : 2) * Epi::ST;      // store instructions per wave and tile (8 row tiles x column halves x 16-byte pieces)
    static_assert(4 + EPI_STORES < 64, "vmcnt is a 6-bit counter");
    bool early_a = false, relax = false;

    int g = 0;
    for (int seq = 0; seq < n_my; ++seq) {
        PP_STAMP(0);
        for (int kt = 0; kt < nk; ++kt, ++g) {
            const int buf = g & 1;
            const unsigned char* la = lds + buf * PP_BUF_BYTES;
            const unsigned char* lw = la + PP_TILE_BYTES;
            const bool more1 = g + 1 < total, more2 = g + 2 < total;
            const int kta = (kt + 1 == nk) ? 0 : kt + 1;                  // K tile the A stream fetches now
            const int ktw = (kt + 2 >= nk) ? kt + 2 - nk : kt + 2;        // ... and the W stream
            // ---- phase a: rows h0 x all 64 columns (32 MFMAs)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    fb0[j][ks] = ld16(lw + swz128(wc * 64 + j * 16 + lr, ks * 4 + lg));
                    fb1[j][ks] = ld16(lw + swz128(wc * 64 + 32 + j * 16 + lr, ks * 4 + lg));
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ld16(la + swz128(wr * 128 + i * 16 + lr, ks * 4 + lg));
            if (kt + 1 == nk && more1) set_aoff(seq + 1);                 // the A stream moves on to the next output tile
            if (more1 && !(early_a && kt == 0)) {                         // (K tile 1 of a tile that follows another was requested before that tile's epilogue)
                issue_half(0, 0, kta, buf ^ 1);
                issue_half(0, 1, kta, buf ^ 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // W(g) is read: its slots may be refilled in phase b
            __builtin_amdgcn_s_barrier();
            mfma_quadrant(0, fb0, 0);
            mfma_quadrant(0, fb1, 1);
            __builtin_amdgcn_s_barrier();
            // ---- phase b: rows h1 x all 64 columns
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ld16(la + swz128(wr * 128 + 64 + i * 16 + lr, ks * 4 + lg));
            if (more2) {
                if (kt + 2 == nk) set_woff(seq + 1);                      // the W stream moves on to the next output tile
                issue_half(1, 0, ktw, buf);
                issue_half(1, 1, ktw, buf);
                // everything but the two W halves just issued -- and, in the first K step behind an epilogue, but that epilogue's stores: what
                // this wait is for (A and W of K tile 1) was requested BEFORE them, and vector-memory operations retire in issue order
                if (kt == 0 && relax) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + EPI_STORES) : "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // A(g) is read: refilled from phase a of the next tile on
            __builtin_amdgcn_s_barrier();
            mfma_quadrant(1, fb1, 1);
            mfma_quadrant(1, fb0, 0);
            __builtin_amdgcn_s_barrier();
        }

        PP_STAMP(1);
        if (wr == 0) __builtin_amdgcn_s_barrier();
        PP_STAMP(2);
        const bool next_tile = PP_SEAM && seq + 1 < n_my;
        auto request_early = [&]() {                                      // A, K tile 1 of the next tile (set_aoff(seq + 1) ran in the last K step)
            early_a = next_tile;
            if (next_tile) {
                issue_half(0, 0, 1, (g & 1) ^ 1);
                issue_half(0, 1, 1, (g & 1) ^ 1);
            }
        };
        int m0, n0; tile_origin(seq, m0, n0);
        const bool full = m0 + PP_BM <= M;                                // every row of the tile is valid: every store below is executed
        relax = next_tile && full;
        if constexpr (TR) {
            // ---- epilogue straight from the accumulators.  A lane holds row lr of each of the wave's 8 row tiles and, with the weight-row
            // permutation above, 8 contiguous columns per (row tile, column half c) -- for the gated epilogues value AND gate of 8 contiguous
            // outputs -- which go to the same functors as the staged form.  No LDS round trip, no wait between the items: the functor
            // arithmetic of all 16 items is one straight-line block.  The column operands (bias) are ordinary loads and are consumed
            // FIRST (asm below): the compiler's wait for an ordinary load is vmcnt(0) whenever an LDS-DMA is in flight, so it has to sit in
            // front of the early request.  The stores are never waited for; one store instruction covers 16 rows x 64 B (bf16).
            // This is the form for the epilogues WITHOUT row operands (q/k/v scatter, GeGLU; measured on the ViT-Base shapes, same box:
            // +7-9 % with the seam handling above).  The residual epilogues read and write 512 B / 1 KB per row of fp32 stream per tile --
            // with every CU in its epilogue at once that is an HBM-rate burst (134 MB in ~20 us at FFN-out) in either form, and hipcc
            // drains the store queue (vmcnt(0)) in front of every item's loads; they keep the staged form (launch_gemm_pp).
            __builtin_amdgcn_sched_barrier(0);                            // nothing of the epilogue moves up into the last K step
            const int strip = n0 + wc * 64;
            // (everything the epilogue derives from the lane id is computed HERE from a recomputed lane id, not hoisted above the K loop
            // where it would be live across a loop that uses nearly all 256 registers)
            int el;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
            const int elr = el & 15, elg = el >> 4;
            constexpr int NG = Epi::PAIRED ? 1 : 2, NT = NG * 8;
            float cb[NG][32];
            int jout[NG];
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                if constexpr (Epi::PAIRED) {
                    const int nv = strip + (elg >> 1) * 32 + (elg & 1) * 8;    // value rows of the lane's 8 outputs; their gate rows are nv + 16
                    epi.cols(nv, nv + 16, cb[c]);
                    jout[c] = (nv >> 5) * 16 + (nv & 15);
                } else {
                    jout[c] = strip + c * 32 + elg * 8;
                    epi.cols(jout[c], cb[c]);
                }
#pragma unroll
                for (int e = 0; e < Epi::NCB; ++e) asm volatile("" : "+v"(cb[c][e]));
            }
            const int mrow = m0 + wr * 128 + elr;                         // row of row tile 0
            request_early();
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) {                             // item (column half c, row tile i): row mrow + i*16, 8 columns from jout[c]
                    const int m = mrow + i * 16;
                    float rr[10], v[8];
                    if constexpr (Epi::HAS_ROW) epi.rowop(min(m, M - 1), jout[c], rr);
                    if constexpr (Epi::PAIRED) {
                        float gt[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][2][r]; gt[r] = acc[i][1][r]; gt[4 + r] = acc[i][3][r]; }
                        epi.fin(min(m, M - 1), jout[c], v, gt, cb[c], rr, m < M);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] = acc[i][c * 2][r]; v[4 + r] = acc[i][c * 2 + 1][r]; }
                        epi.fin(min(m, M - 1), jout[c], v, cb[c], rr, m < M);
                    }
                    if constexpr (Epi::HAS_ROW) __builtin_amdgcn_sched_barrier(0);     // (one item's row operands at a time: registers)
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            request_early();
            const int nbase = n0 + wc * 64;
            float cb[32];
            int cv, nv;                                                   // paired: value columns of this lane; plain: its 8 columns
            if constexpr (Epi::PAIRED) {
                const int vg = lane & 3;
                cv = (vg >> 1) * 32 + (vg & 1) * 8; nv = nbase + cv;
                epi.cols(nv, nv + 16, cb);
            } else {
                cv = (lane & 7) * 8; nv = nbase + cv;
                epi.cols(nv, cb);
            }
            const int jout = Epi::PAIRED ? (nv >> 5) * 16 + (nv & 15) : nv;
            constexpr int ITEMS = Epi::PAIRED ? 1 : 2, RSTEP = Epi::PAIRED ? 16 : 8, RSHIFT = Epi::PAIRED ? 2 : 3;
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                 // row tile i of the wave: 16 rows x 64 columns
                const int mbase = m0 + wr * 128 + i * 16;
                float rr[ITEMS][10];
                if constexpr (Epi::HAS_ROW) {
#pragma unroll
                    for (int q = 0; q < ITEMS; ++q) epi.rowop(min(mbase + (lane >> RSHIFT) + RSTEP * q, M - 1), jout, rr[q]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[sidx(lg * 4 + r, j * 16 + lr)] = acc[i][j][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private region: no barrier needed
#pragma unroll
                for (int q = 0; q < ITEMS; ++q) {
                    const int row = (lane >> RSHIFT) + RSTEP * q, m = mbase + row;
                    float v[8];
                    load8(&stage[sidx(row, cv)], v);
                    if constexpr (Epi::PAIRED) {
                        float gt[8];
                        load8(&stage[sidx(row, cv + 16)], gt);
                        epi.fin(min(m, M - 1), jout, v, gt, cb, rr[q], m < M);
                    } else {
                        epi.fin(min(m, M - 1), jout, v, cb, rr[q], m < M);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // reads done before the next row tile overwrites the region
                // keep the row tiles' residual loads from being hoisted over one another (8 x 20 registers: the kernel is at 256)
                if constexpr (Epi::HAS_ROW) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        PP_STAMP(3);
        if (wr == 1 && seq + 1 < n_my) __builtin_amdgcn_s_barrier();      // stagger again for the next tile
    }
}

// true when the shape fits this kernel; otherwise the caller uses gemm_big_kernel
inline bool gemm_pp_fits(int M, int N, int K) {
    return K % PP_BK == 0 && K >= 2 * PP_BK && N % PP_BN == 0 && M >= PP_BM && (long long)M * K < (1ll << 31) && (long long)N * K < (1ll << 31);
}

// tr: 1 = epilogue straight from the transposed accumulators, 0 = LDS-staged epilogue, -1 = by epilogue (direct unless it has row operands).
// The engine reads TXO_PP_TR once per engine and passes it here (tests build engines with each form and compare them bit for bit).

template <class Epi, bool TR>
inline void launch_gemm_pp_t(hipStream_t s, const bf16* A, const bf16* W, int M, int N, int K, Epi epi, int rev, int sb_mb, int ct_force) {
    const int tiles_m = (M + PP_BM - 1) / PP_BM, tiles_n = N / PP_BN;
    const int n_tiles = tiles_m * tiles_n;
    // persistent grid: one block per CU (128-160 KiB LDS each).  Both the CU count and the > 64 KiB dynamic-LDS opt-in are per
    // DEVICE (one process may drive several GPUs through several engines), so they are kept per device ordinal
    static int cus_of[64] = {0}; static bool attr_of[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = (dev >= 0 && dev < 64) ? dev : 0;
    if (!cus_of[di] || di != dev) {
        hipDeviceProp_t prop;
        cus_of[di] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int cus = cus_of[di];
    constexpr int smem = TR ? PP_LDS_BYTES : PP_SMEM_BYTES;
    if (!attr_of[di] || di != dev) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<Epi, TR>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_of[di] = true;
    }
    // band width: all column tiles when W is small enough to stay in an XCD's L2 anyway (or K is long: A panels are then
    // the larger operand and must not be re-streamed per band), else as many column tiles as ~1.6 MB of W
    const long long w_bytes = (long long)N * K * 2, coltile_bytes = (long long)PP_BN * K * 2;
    int ct = tiles_n;
    if (w_bytes > (5ll << 19) && K < 2048) ct = (int)std::max<long long>(1, std::min<long long>(tiles_n, (13ll << 17) / coltile_bytes));
    if (ct_force > 0 && ct < tiles_n) ct = std::max(1, std::min(tiles_n, ct_force));
    // narrower last band: when tiles_n % ct != 0 the row-major walk of the last band uses its own width (tile_origin)
    // row super-block (kernel comment): only where there is more than one band; sb_mb = MB of A per super-block (0 = one super-block)
    int sbr = tiles_m;
    if (ct < tiles_n && sb_mb > 0) sbr = (int)std::max<long long>(8, std::min<long long>(tiles_m, ((long long)sb_mb << 20) / ((long long)PP_BM * K * 2)));
    const int grid = ((std::min(n_tiles, cus) + 7) / 8) * 8;
    hipLaunchKernelGGL((gemm_pp_kernel<Epi, TR>), dim3(grid), dim3(PP_THREADS), smem, s, A, W, M, N, K, tiles_n, n_tiles, ct, sbr, rev, epi);
}
template <class Epi>
inline void launch_gemm_pp(hipStream_t s, const bf16* A, const bf16* W, int M, int N, int K, Epi epi, int tr = -1, int rev = 0, int sb_mb = PP_SB_MB, int ct_force = 0) {
    if (tr < 0) tr = Epi::HAS_ROW ? 0 : 1;
    if (tr) launch_gemm_pp_t<Epi, true>(s, A, W, M, N, K, epi, rev, sb_mb, ct_force); else launch_gemm_pp_t<Epi, false>(s, A, W, M, N, K, epi, rev, sb_mb, ct_force);
}

}  // namespace txo
