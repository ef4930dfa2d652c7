// Decode-step CROSS attention in latent form: scores and values are taken against the raw encoder rows, not against
// per-layer projected K / V panels.
//
// Reference (model/attention.py:114-127,148,166,172-173 with enc is not None, ctor :216, call :248): every decoder layer's
// cross attention projects k = enc Wk^T and v = enc Wv^T from the SAME raw encoder output `enc` (N rows of D), splits
// heads (f = head*64 + j), energy_h[n] = q_h . k_h[n] * 0.125, softmax over n, out_h = sum_n p_h[n] v_h[n].
// dim_head is fixed at 64, so the cached k and v rows of a key (2 * heads * 64 elements per layer) are linear images of
// ONE D-element row that is the same for every layer and every head:
//     energy_h[n] = (0.125 Wk_h^T q_h) . enc[n]      =: q'_h . enc[n]          (q'_h has D elements)
//     out_h       = Wv_h (sum_n p_h[n] enc[n])       =: Wv_h c_h               (c_h has D elements)
// The K/V form (dec_attn.h) streams 2 * N * heads * 64 elements per (image, layer, position); this form streams N * D
// per (image, layer, position), shared by all heads (config.yml dims: 1 206 272 -> 301 568 bytes in bf16; ViT-Base: 1 809 408 ->
// 904 704).  The projection of K / V at decode_begin disappears.
//
// One cross-attention sub-layer = five launches (engine.hip: launch_lat_cross):
//   1. dec_gemm<PRO_LN2, EPI_STORE_T>   x = LN(y) (residual), z = LN(x), q = z Wq^T                       rows x inner
//   2. grp_gemm  (per head, K = 64)      q'_h = q_h (0.125 Wk_h)   -- Wk stored transposed per head, scale folded in (exact)
//   3. lat_core                          c_h = softmax_n(q'_h . enc[n]) enc                 -- this file's tile, below
//   4. grp_gemm  (per head, K = D)       o_h = c_h Wv_h^T ; 'b h n d -> b n (h d)'
//   5. dec_gemm<PRO_NONE, EPI_GLU_RES>   gated output projection + residual (as in the K/V form)
// The projections run on the matrix pipe over all rows (weights read once per 16 rows).  (A first version kept them inside the
// attention tile, one row at a time on the VALU -- probes/rejected/lat_attn_in_tile_projections.h.txt: 192 KB of weights per (row,
// head pair) on top of the 301 KB of encoder rows; the tile then moved more bytes per CU than the K/V form and lost at every batch
// size below 129, also inside the persistent kernel: profiles/r04_latent_in_tile_*.)
//
// lat_core tile = (row, group of G <= LA_GMAX = 16 heads: the 16 columns of the MFMA tile) by NW waves (8 at D <= 256, 4 above):
//   keys are dealt to the waves in 16-key tiles.  Per tile: the encoder rows are requested as WHOLE rows (bf16, r05: 2 x 512 B per
//   wave-instruction; fragment-shaped requests -- 16 rows x 64 B -- are bound by the CU's address path, see `coal` below), written to a
//   wave-private LDS image, S^T[key][head] = enc_tile q'^T takes its A fragments from the image, and the image is read back TRANSPOSED
//   (bf16: ds_read_b64_tr_b16) as the A operand of c^T[d][head] += enc_tile^T P^T; the P accumulator tile in its C layout IS the B
//   operand (no lane movement).  (fp32, and the per-key slot tables of a beam search's self attention: the registers are the fragments.)  Online softmax per head = per lane column.  Heads sit
//   on the MFMA column: 16 columns, G of them used (a beam search packs an image's k x heads heads 16 to a tile).  The waves' (m, l, c)
//   are merged through LDS.
// A head's arithmetic does not depend on G or on which tile computed it.
// Algorithmic bytes per launch = rows * N * D * sizeof(T) (+ q' and c: 2 * rows * heads * D * sizeof(T)).  Bound: HBM.
// The output c is row-major [rows][heads * D], or -- for the folded output projection on the launch path -- in that GEMM's tiled A layout (c_hpr).
#pragma once
#include "common.h"
#include "dec_attn.h"
#include "dec_gemm.h"
#include "enc_attn.h"   // pack_bf16x2

namespace txo {

constexpr int LA_GMAX = 16;               // heads per tile = columns of the MFMA tile
constexpr int LA_PATH_TILES = 4;          // beam search, self attention: key tiles per wave whose slots are looked up (8 waves: 512 positions)

template <typename T> struct LatCoreArgs {
    const T* qp;                          // [rows][heads*D]   q' (already scaled)
    const T* enc;                         // [images][len][D]
    T* c;                                 // [rows][heads*D]   normalised sum_n p[n] enc[n]
    int rows, heads, G, ngrp;             // heads per tile, tiles per row = ceil(heads / G)
    int len;                              // keys: encoder tokens (cross) / decoded positions t + 1 (self, when the host knows t)
    int kv_div;                           // beam search: encoder image = row / kv_div
    // SELF attention in latent form (r05): `enc` is the history of normalised block inputs z, [rows][enc_rows][D] with enc_rows = the
    // positional table's length; keys 0 .. t.  t_ptr: the device-side position counter (graph replay: len = *t_ptr + 1), or null.
    // path: beam search -- position p of row r lives in slot path[r * path_stride + p] (the current position in the row's own slot), or null.
    int enc_rows;                         // rows per image of `enc` (0 = len)
    const int* t_ptr;
    const short* path; int path_stride;
    // c in the tiled layout of dec_gemm's A operand (DecGemmArgs::a_tiled): output row m = row * (heads / c_hpr) + head / c_hpr, column
    // k = (head % c_hpr) * D + d of a [.][c_hpr * D] matrix; c_hpr = 0: row-major [rows][heads * D]
    int c_hpr;
    unsigned long long* stamps;           // diagnostic: per tile {entry, last key tile done, exit}
};

template <int D_> constexpr int la_waves() { return D_ <= 256 ? 8 : 4; }
// NW_: waves per tile.  The 8-wave tile of the narrow widths leaves ONE tile per CU (72 KB of LDS, 240 registers), so nothing runs under
// a tile's q' prologue and its cross-wave merge; the 4-wave form (r05: 40 KB) lets two tiles share a CU -- one streams while the other
// merges -- and doubles the tile slots the head grouping can fill (engine.hip: latent_group).
template <typename T, int D_, int NW_ = la_waves<D_>()> constexpr size_t la_lds_bytes() {
    // stats | q' image [16][D] | per-wave transposition scratch (later: the waves' partial c)
    return (size_t)NW_ * 16 * 2 * 4 + (size_t)16 * D_ * sizeof(T) + (size_t)NW_ * 16 * D_ * sizeof(T);
}
// fp32 at 768 would need 196 KB of scratch: the K/V form serves that shape
template <typename T, int D_> constexpr bool la_supported() { return la_lds_bytes<T, D_>() <= 160 * 1024 - 4096; }

// byte offset of 16-byte chunk `ch` of row `key` in a wave's [16][D] image.  The swizzle works on 32-byte blocks (block ^ key
// bits) so that the 8 rows a 32-lane half reads TRANSPOSED (4 rows x 32 B per 16 lanes) fall into 8 different 32-byte bank
// groups, and swaps the two chunks of a block with key bit 2 so that the 8 keys of one ds_write_b128 lane group hit 8
// different 16-byte slots of the 128-byte write window.
template <int ROWB, int XM> __device__ inline int la_off(int key, int ch) {
    return key * ROWB + (((((ch >> 1) ^ (key & XM)) << 1) | ((ch & 1) ^ ((key >> 2) & 1))) << 4);
}

template <typename T, int D_, int NW_ = la_waves<D_>()>
__device__ __forceinline__ void lat_core_tile(const LatCoreArgs<T>& a, int tile, int tid, unsigned char* lds) {
    constexpr int NW = NW_, NT = NW * 64;
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK;
    constexpr int KC = D_ / KCH;                              // 64-byte k-chunks per encoder row
    constexpr int DT = D_ / 16;                               // 16-wide d tiles of c^T
    constexpr int ROWB = D_ * (int)sizeof(T);
    constexpr int NBLK = ROWB / 32;                           // 32-byte blocks per row
    constexpr int XM = (NBLK % 8 == 0) ? 7 : ((NBLK % 4 == 0) ? 3 : 1);
    constexpr bool FAST = sizeof(T) == 2;
    constexpr int QSM = (D_ / PER16 >= 16) ? 15 : 7;          // chunk swizzle of the q' image (a row has >= 8 chunks)
    typedef short s16x4 __attribute__((ext_vector_type(4)));

    const int img = tile / a.ngrp, hg = tile - img * a.ngrp;
    const int h0 = hg * a.G, nh = min(a.G, a.heads - h0);
    const int HD = a.heads * D_;
    const int lane = tid & 63, lc = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long ts0 = 0, ts1 = 0;
    if (a.stamps) ts0 = __builtin_amdgcn_s_memrealtime();

    float* st_m = reinterpret_cast<float*>(lds);              // [NW][16]
    float* st_l = st_m + NW * 16;                             // [NW][16]
    unsigned char* qc = reinterpret_cast<unsigned char*>(st_l + NW * 16);     // q' image [16][D_] of T (rows >= nh: zero)
    unsigned char* scr_all = qc + (size_t)16 * D_ * sizeof(T);
    unsigned char* scr = scr_all + (size_t)wave * 16 * ROWB;
    float* part = reinterpret_cast<float*>(scr_all);          // [NW][nh][D_] after the key loop

    // ---- the tile's q' rows first (the head of the dependency chain), then the first encoder tiles ----
    constexpr int QCH = D_ / PER16;                           // 16-byte chunks per q' row
    constexpr int QPT = (16 * QCH + NT - 1) / NT;             // chunks of the [16][D] image per thread
    u32x4 qreg[QPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int idx = tid + i * NT, h = idx / QCH, ch = idx - h * QCH;
        const int hc = min(h, nh - 1);                        // (rows >= nh of the image are zeroed below: no branch around the load)
        qreg[i] = ld16(a.qp + (size_t)img * HD + (size_t)(h0 + hc) * D_ + ch * PER16);
        if (h >= nh) qreg[i] = u32x4{0u, 0u, 0u, 0u};
    }
    const int kvimg = img / a.kv_div;
    const int len = a.t_ptr ? *a.t_ptr + 1 : a.len;
    const int erows = a.enc_rows ? a.enc_rows : len;
    const T* eb = a.enc + (size_t)kvimg * erows * D_;
    const int ntiles = (len + 15) >> 4, nt_w = (ntiles + NW - 1) / NW;
    const int last_key = len - 1;
    // beam search: the slots of this lane's keys (tile it of this wave: key 16 * (wave + it * NW) + lc), fetched before any encoder
    // row so that no row request waits behind a slot lookup; LA_PATH_TILES tiles per wave cover the positional table (engine.hip checks)
    short slot[LA_PATH_TILES];
    if (a.path) {
#pragma unroll
        for (int it = 0; it < LA_PATH_TILES; ++it) {
            const int key = min((wave + it * NW) * 16 + lc, last_key);
            slot[it] = key == last_key ? (short)img : a.path[(size_t)img * a.path_stride + key];
        }
    }
    // NF: tiles requested AHEAD of the one being consumed (registers: KC x 4 per tile), NF + 1 statically named register
    // sets used round-robin.  Every request is unconditional (a branch around the refill made hipcc wait vmcnt(0) before every
    // tile, so each tile paid the full latency of the refill issued just before it; rotating the sets through copies forces
    // the same wait at the copy): tiles past the end clamp to the last key's row -- one cache line per instruction.
    constexpr int NF = KC <= 8 ? 2 : (NW == 4 ? 1 : 0);       // (4-wave tiles run one wave per SIMD: 512 registers; f32 at 8 waves: one set)
    u32x4 cur[KC], a1[NF >= 1 ? KC : 1], a2[NF == 2 ? KC : 1];
    (void)a1; (void)a2;
    // Request shape (r05).  Fragment-shaped: instruction kc of a tile covers 16 rows x 64 B (lane = row lc, piece lg) -- the registers ARE the
    // MFMA fragments, but every instruction touches 16 half-used 128-byte lines, and the CU's address path, not HBM, then bounds the stream
    // (probes/xcd_bw.hip: 77 MB over 256 workgroups in 12.9 us with this shape against 9.8 us with whole lines).  Whole rows (COAL): instruction j
    // covers RPI consecutive rows completely (lane = row j * RPI + lane / CPR, piece lane % CPR); the tile goes to the wave's LDS image as before
    // and the score MFMAs take their fragments from the image.  Same values in the same MFMAs: the same bits.  The per-key slot tables of a beam
    // search's self attention (a.path) keep the fragment shape (a lane would need eight slots per tile).
    // (r06, measured and NOT kept: the same shape for width 768 -- a tile's 16 consecutive rows requested as 24 linear KiB, three (row, piece)
    // pairs per lane -- needs the fragments re-read from the image while both prefetch sets and 192 accumulator registers are live: the
    // kernel spills inside the tile loop, 55 -> 123 us per launch; with buffer loads whose rows past the last key read as zero, width 256
    // went 17.0 -> 18.2 us.  profiles/r06_cfg4_latent_form.txt)
    constexpr int CPR = ROWB / 16, RPI = CPR <= 64 ? 64 / CPR : 1;     // 16-byte pieces per row, rows per wave-instruction
    constexpr bool COAL_OK = FAST && CPR <= 64 && 64 % CPR == 0 && KC == 16 / RPI;
    const bool coal = COAL_OK && a.path == nullptr;
    const int c_row = lane / CPR, c_ch = lane % CPR;
    auto issue = [&](int it, u32x4 (&dst)[KC]) {              // `it`-th tile of this wave: keys 16 * (wave + it * NW) ..
        if (coal) {
            const int k0 = (wave + it * NW) * 16 + c_row;
#pragma unroll
            for (int j = 0; j < KC; ++j) dst[j] = ld16(eb + (size_t)min(k0 + j * RPI, last_key) * D_ + c_ch * PER16);
            return;
        }
        const int key = min((wave + it * NW) * 16 + lc, last_key);
        const T* p = eb + (size_t)key * D_ + lg * PER16;
        if (a.path) {                                         // (wave-uniform branch; the slot by a select chain: no run-time register indexing)
            int sl = img;                                     // tiles past the table clamp to the last key, which lives in the row's own slot
#pragma unroll
            for (int j = 0; j < LA_PATH_TILES; ++j) sl = it == j ? (int)slot[j] : sl;
            p = a.enc + ((size_t)sl * erows + key) * D_ + lg * PER16;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) dst[kc] = ld16(p + kc * KCH);
    };
    issue(0, cur);
    if constexpr (NF >= 1) issue(1, a1);

    // q' image: 16-byte chunks of a head's row XOR-swizzled with the head (conflict-free fragment reads)
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int idx = tid + i * NT, h = idx / QCH, ch = idx - h * QCH;
        if (idx < 16 * QCH) st16(qc + ((size_t)h * D_ + ((ch ^ (h & QSM)) * PER16)) * sizeof(T), qreg[i]);
    }
    __syncthreads();
    // B fragments of S^T = enc q'^T: lane (head = lc, group lg) holds q'[head][kc*KCH + lg*PER16 ..]; kept in registers where
    // they fit beside the tiles in flight, else re-read from LDS for every tile
    constexpr bool QF_REGS = KC <= 8;
    auto qfrag = [&](int kc) { return ld16(qc + ((size_t)lc * D_ + (((kc * 4 + lg) ^ (lc & QSM)) * PER16)) * sizeof(T)); };
    u32x4 qf[QF_REGS ? KC : 1];
    if constexpr (QF_REGS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) qf[kc] = qfrag(kc);
    }

    // ---- key tiles ----
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -3.0e38f, l_run = 0.f;
    // transposed reads: lane (lg, q = lc >> 2, pp = lc & 3) addresses key 4 lg + q, columns 16 dt + 4 pp ..
    const int tq = lc >> 2, tpp = lc & 3, tkey = lg * 4 + tq;
    const int tr_base = tkey * ROWB + ((((tpp >> 1) ^ ((tkey >> 2) & 1))) << 4) + ((tpp & 1) << 3);
    const int tr_kx = tkey & XM;
    auto process = [&](int it, u32x4 (&e)[KC]) {
        const int t0 = (wave + it * NW) * 16;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (coal) {
            // whole-row requests: registers -> image (row j * RPI + lane / CPR, piece lane % CPR), fragments <- image
#pragma unroll
            for (int j = 0; j < KC; ++j) st16(scr + la_off<ROWB, XM>(j * RPI + c_row, c_ch), e[j]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                if constexpr (QF_REGS) mma16<T>(s, ld16(scr + la_off<ROWB, XM>(lc, 4 * kc + lg)), qf[kc]);
                else mma16<T>(s, ld16(scr + la_off<ROWB, XM>(lc, 4 * kc + lg)), qfrag(kc));
            }
        } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if constexpr (QF_REGS) mma16<T>(s, e[kc], qf[kc]);
            else {
                mma16<T>(s, e[kc], qfrag(kc));
                if ((kc & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // q' fragments four at a time (all 24 at once: 96 registers)
            }
        }
        // the same registers -> the wave's LDS image (row = key lc, chunks 4 kc + lg)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) st16(scr + la_off<ROWB, XM>(lc, 4 * kc + lg), e[kc]);
        }
        asm volatile("" ::: "memory");
        // online softmax per head (lane column); this lane's keys: t0 + 4 lg + r
        float sv[4]; bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ok[r] = t0 + lg * 4 + r <= last_key; sv[r] = ok[r] ? s[r] : -3.0e38f; }
        const float mt = grp4_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
        if (__builtin_amdgcn_ballot_w64(mt > m_run) != 0ull) {
            asm volatile("; a running maximum grows" ::: "memory");
            const float m_new = fmaxf(m_run, mt);
            const float alpha = exp_sel<FAST>(m_run - m_new);  // first tile: exp(-huge) = 0 with acc = l = 0
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] *= alpha;
        }
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = ok[r] ? exp_sel<FAST>(sv[r] - m_run) : 0.f;
        if constexpr (sizeof(T) == 2) {
            const unsigned p01 = pack_bf16x2(p[0], p[1]), p23 = pack_bf16x2(p[2], p[3]);
            // the normaliser sums exactly the rounded weights that multiply the encoder rows
            l_run += (__uint_as_float(p01 << 16) + __uint_as_float(p01 & 0xffff0000u)) + (__uint_as_float(p23 << 16) + __uint_as_float(p23 & 0xffff0000u));
            const s16x4 pb = __builtin_bit_cast(s16x4, uint2{p01, p23});
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const unsigned char* va = scr + tr_base + ((dt ^ tr_kx) << 5);
                const s16x4 af = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uint32_t)(uintptr_t)va);
                acc[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, pb, acc[dt], 0, 0, 0);
            }
        } else {
            l_run += (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = lg * 4 + r, d = dt * 16 + lc;
                    const float af = *reinterpret_cast<const float*>(scr + la_off<ROWB, XM>(key, d >> 2) + ((d & 3) << 2));
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, p[r], acc[dt], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
    };
    // six tiles per trip, every exit a wave-uniform break: hipcc drains vmcnt(0) at a loop header (the back edge's pending
    // requests merge with the pre-header's), so the trip is made long enough that 589 keys never take the back edge
    if constexpr (NF == 2) {
        for (int it = 0; it < nt_w; it += 6) {
            issue(it + 2, a2); process(it, cur);
            if (it + 1 >= nt_w) break;
            issue(it + 3, cur); process(it + 1, a1);
            if (it + 2 >= nt_w) break;
            issue(it + 4, a1); process(it + 2, a2);
            if (it + 3 >= nt_w) break;
            issue(it + 5, a2); process(it + 3, cur);
            if (it + 4 >= nt_w) break;
            issue(it + 6, cur); process(it + 4, a1);
            if (it + 5 >= nt_w) break;
            issue(it + 7, a1); process(it + 5, a2);
        }
    } else if constexpr (NF == 0) {
        for (int it = 0; it < nt_w; ++it) { if (it > 0) issue(it, cur); process(it, cur); }
    } else {                                                  // two sets: a set is refilled as soon as its tile has been consumed
        for (int it = 0; it < nt_w; it += 6) {
            process(it, cur);
            if (it + 1 >= nt_w) break;
            issue(it + 2, cur); process(it + 1, a1);
            if (it + 2 >= nt_w) break;
            issue(it + 3, a1); process(it + 2, cur);
            if (it + 3 >= nt_w) break;
            issue(it + 4, cur); process(it + 3, a1);
            if (it + 4 >= nt_w) break;
            issue(it + 5, a1); process(it + 4, cur);
            if (it + 5 >= nt_w) break;
            issue(it + 6, cur); process(it + 5, a1);
            issue(it + 7, a1);
        }
    }
    if (a.stamps) { asm volatile("" :: "v"(acc[0][0])); ts1 = __builtin_amdgcn_s_memrealtime(); }

    // ---- merge the waves' (m, l, c^T), normalise, store c ----
    // the waves' partial c alias the transposition scratch, PCH heads at a time (16 keys x D of T per wave = PCH x D floats)
    constexpr int PCH = 4 * (int)sizeof(T);
    l_run = grp4_sum(l_run);
    if (lg == 0) { st_m[wave * 16 + lc] = m_run; st_l[wave * 16 + lc] = l_run; }
    T* crow = a.c + (size_t)img * HD + (size_t)h0 * D_;
    for (int hb = 0; hb < nh; hb += PCH) {
        const int nhb = min(PCH, nh - hb);
        __syncthreads();                                      // every wave is done with its transposition image / the previous chunk
        if (lc >= hb && lc < hb + nhb) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *reinterpret_cast<f32x4*>(&part[((size_t)wave * nhb + (lc - hb)) * D_ + dt * 16 + lg * 4]) = acc[dt];
        }
        __syncthreads();
        for (int idx = tid * 4; idx < nhb * D_; idx += NT * 4) {   // four consecutive d of one head per thread
            const int h = hb + idx / D_;
            float m = st_m[h];
#pragma unroll
            for (int w = 1; w < NW; ++w) m = fmaxf(m, st_m[w * 16 + h]);
            float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float e = exp_sel<FAST>(st_m[w * 16 + h] - m);
                const f32x4 pv = *reinterpret_cast<const f32x4*>(&part[((size_t)w * nhb) * D_ + idx]);
#pragma unroll
                for (int j = 0; j < 4; ++j) num[j] = fmaf(e, pv[j], num[j]);
                den = fmaf(e, st_l[w * 16 + h], den);
            }
            T* dst = crow + (size_t)hb * D_ + idx;
            if (a.c_hpr) {                                    // (kernel-uniform) tiled: 4 consecutive d stay inside one 64-byte chunk
                const int hh = h0 + h, m = img * (a.heads / a.c_hpr) + hh / a.c_hpr, k = (hh % a.c_hpr) * D_ + (idx % D_);
                constexpr int KCHT = 64 / (int)sizeof(T);
                dst = a.c + (((size_t)(m >> 4) * (a.c_hpr * D_ / KCHT) + (k / KCHT)) * 16 + (m & 15)) * KCHT + (k % KCHT);
            }
            if constexpr (sizeof(T) == 2) {
                uint2 o; o.x = pack_bf16x2(num[0] / den, num[1] / den); o.y = pack_bf16x2(num[2] / den, num[3] / den);
                *reinterpret_cast<uint2*>(dst) = o;
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(num[0] / den, num[1] / den, num[2] / den, num[3] / den);
            }
        }
    }
    if (a.stamps && tid == 0) {
        unsigned long long* d = a.stamps + 3 * (size_t)tile;
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_amdgcn_s_memrealtime();
    }
}

template <typename T, int D_, int NW_ = la_waves<D_>()>
__global__ __launch_bounds__(NW_ * 64) void lat_core_kernel(LatCoreArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char la_smem[];
    // workgroups b and b + 8 run on one XCD (round-robin dispatch): every tile that reads ONE image's encoder rows -- its head groups, and in
    // a beam search its kv_div beams -- goes to ONE XCD, whose L2 then fetches those rows once for all of them
    // (image = (slot / tpu) * 8 + xcd, tile within the image = slot % tpu: row = image * kv_div + w / ngrp, head group = w % ngrp)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tpu = a.kv_div * a.ngrp, unit = (slot / tpu) * 8 + xcd, w = slot - (slot / tpu) * tpu;
    const int row = unit * a.kv_div + w / a.ngrp, hg = w - (w / a.ngrp) * a.ngrp;
    if (row >= a.rows) return;
    lat_core_tile<T, D_, NW_>(a, row * a.ngrp + hg, threadIdx.x, la_smem);
}

// ---- grouped projection: out[r][g*NG + n] = sum_k A[r][g*KG + k] * W[g*NG + n][k]   (g = head) ---------------------------------
// The two per-head projections around lat_core: q'_h = q_h (0.125 Wk_h) with KG = 64, NG = D, and o_h = c_h Wv_h^T with KG = D,
// NG = 64.  One wave = one 16 (rows) x 16 (columns) tile with its whole K (64 ... 768) in the wave: the weight rows are the A
// operand (lane column = activation row), so a lane ends up with 4 consecutive output columns of one row -> 8 / 16-byte stores.
// Block = 4 waves = 64 columns of the same 16 rows.  Bound: latency (a few MFLOP; weights from L2).
template <typename T> struct GrpGemmArgs {
    const T* A; int lda;                  // [rows][lda]
    const T* W;                           // [N][KG]
    T* out; int ldo;                      // [rows][ldo]
    int rows, N, NG;
};
template <typename T, int KG>
__global__ __launch_bounds__(256) void grp_gemm_kernel(GrpGemmArgs<T> a) {
    constexpr int PER16 = Elem<T>::PER16, KCH = Elem<T>::KCHUNK, KC = KG / KCH;
    constexpr int GR = KC < 8 ? KC : 8;                       // k-chunks requested together
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 64 + wave * 16, m0 = blockIdx.y * 16;
    if (n0 >= a.N) return;                                    // wave-uniform
    const int g = n0 / a.NG;
    const T* wrow = a.W + (size_t)(n0 + lr) * KG + lg * PER16;
    const T* arow = a.A + (size_t)min(m0 + lr, a.rows - 1) * a.lda + (size_t)g * KG + lg * PER16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += GR) {
        u32x4 wf[GR], af[GR];
#pragma unroll
        for (int c = 0; c < GR; ++c) { wf[c] = ld16(wrow + (k0 + c) * KCH); af[c] = ld16(arow + (k0 + c) * KCH); }
        __builtin_amdgcn_sched_barrier(0);                    // every request ahead of the first MFMA (else hipcc interleaves them: one latency per pair)
#pragma unroll
        for (int c = 0; c < GR; ++c) mma16<T>(acc, wf[c], af[c]);
    }
    // C: column = lane & 15 = activation row, rows 4 lg + r = output columns n0 + 4 lg + r
    if (m0 + lr < a.rows) {
        T* dst = a.out + (size_t)(m0 + lr) * a.ldo + n0 + lg * 4;
        if constexpr (sizeof(T) == 2) {
            uint2 o; o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
            *reinterpret_cast<uint2*>(dst) = o;
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

}  // namespace txo
