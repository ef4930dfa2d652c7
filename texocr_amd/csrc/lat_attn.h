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
// per (image, position), shared by the heads of a tile and by all layers (config.yml dims: 1 206 272 -> 301 568 bytes in
// bf16, and the same bytes again for the next layer).  The projection of K / V at decode_begin disappears.
//
// Tile = (image row, group of G <= 8 heads) by NW waves (8 at D <= 256, 4 above):
//   0. weights of the two small projections are requested at entry (they depend on nothing), then the first two 16-key
//      tiles of `enc` per wave (they do not depend on the previous stage either);
//   1. wave 0: x = LN(y) (residual, written by head group 0), z = LN(x) -> LDS               (attention.py:257-259, :243)
//   2. q_h = Wq_h z (VALU, 4 threads per output), q'_h = 0.125 Wk_h^T q_h -> LDS as the B operand image [16][D] of T
//   3. keys are dealt to the waves in 16-key tiles.  Per tile: S^T[key][head] = enc_tile q'^T on the matrix pipe with the
//      enc fragments straight from global memory in registers (A operand: one 16-byte piece per lane is a fragment), the
//      same registers are written to a wave-private LDS image and read back TRANSPOSED (bf16: ds_read_b64_tr_b16) as the A
//      operand of c^T[d][head] += enc_tile^T P^T; the P accumulator tile in its C layout IS the B operand (no lane
//      movement).  Online softmax per head = per lane column.  Heads sit on the MFMA column: 16 columns, G of them used --
//      the matrix pipe is 16x faster than this needs, the point is that the VALU would be 10x too slow.
//   4. the waves' (m, l, c) are merged through LDS, o_h = Wv_h c_h (VALU), 'b h n d -> b n (h d)'.
// A head's arithmetic does not depend on G or on which tile computed it -> the launch path and the persistent kernel may
// pick different G and still agree bit for bit.
// Algorithmic bytes per (image, position) = N * D * sizeof(T) for all layers that find it in L2 / once per layer from HBM.
// Bound: L2 -> CU fill rate (66-135 GB/s per CU) at batch <= 64, HBM at batch 256.
#pragma once
#include "common.h"
#include "dec_attn.h"
#include "dec_gemm.h"
#include "enc_attn.h"   // pack_bf16x2

namespace txo {

constexpr int LA_GMAX = 8;                // heads per tile (the waves' partial c alias the transposition scratch)

template <typename T> struct LatAttnArgs {
    const float* y;                       // [rows][D] stream in front of the sub-layer
    float* x_out;                         // [rows][D] residual x = LN(y) (written by head group 0's tile)
    const float* gamma; const float* beta;
    int rows;                             // launch path: rows of this launch
    int heads, G, ngrp;                   // heads per tile, tiles per row = ceil(heads / G)
    int len;                              // encoder tokens
    const T* Wq;                          // [heads*64][D]
    const T* WkT;                         // [heads][D][64]   (Wk_h transposed: row d holds Wk[h*64 + 0..63][d])
    const T* Wv;                          // [heads*64][D]
    const T* enc;                         // [images][len][D]
    T* out;                               // [rows][heads*64]
    int kv_div;                           // beam search: encoder image = row / kv_div
    unsigned long long* stamps;           // diagnostic: per tile {entry, last key tile done (stamp_mode 1: first key tile about to start), exit}
    int stamp_mode;
};

template <int D_> constexpr int la_waves() { return D_ <= 256 ? 8 : 4; }
template <typename T, int D_> constexpr size_t la_lds_bytes() {
    // zs | qs | stats | q' image (later: merged c) | per-wave transposition scratch (later: the waves' partial c)
    return (size_t)D_ * 4 + LA_GMAX * 64 * 4 + (size_t)la_waves<D_>() * 16 * 2 * 4 + (size_t)16 * D_ * 4 +
           (size_t)la_waves<D_>() * 16 * D_ * sizeof(T);
}
// fp32 at 768 would need 196 KB of scratch: the K/V form serves that shape
template <typename T, int D_> constexpr bool la_supported() { return la_lds_bytes<T, D_>() <= 160 * 1024 - 8192; }

// byte offset of 16-byte chunk `ch` of row `key` in a wave's [16][D] image.  The swizzle works on 32-byte blocks (block ^ key
// bits) so that the 8 rows a 32-lane half reads TRANSPOSED (4 rows x 32 B per 16 lanes) fall into 8 different 32-byte bank
// groups, and swaps the two chunks of a block with key bit 2 so that the 8 keys of one ds_write_b128 lane group hit 8
// different 16-byte slots of the 128-byte write window.
template <int ROWB, int XM> __device__ inline int la_off(int key, int ch) {
    return key * ROWB + (((((ch >> 1) ^ (key & XM)) << 1) | ((ch & 1) ^ ((key >> 2) & 1))) << 4);
}

// The three small projections of a tile (q = Wq z, q' = Wk^T q, o = Wv c): rows of W (K elements each) . vec -> emit(row, value).
// Thread (pd = tid >> 2, prt = tid & 3) owns the 16-byte pieces prt, prt + 4, ... of a row (a quad reads 64 contiguous bytes) and
// the quad sums its partial dots.  A STEP = RB rounds of NT/4 rows whose fragments (<= 8 or one row's worth of registers per
// thread) are requested together; the next step's fragments are requested before the current step is consumed (these are
// dependent load -> use chains out of L2, ~0.7 us each: at G = 8 the q' projection alone is 16 rounds).
template <typename T, int K> struct LaRows {
    static constexpr int PER16 = Elem<T>::PER16, PIECES = K / PER16, PPT = PIECES / 4;   // pieces per thread per row
    static_assert(PIECES % 4 == 0, "row pieces must split over a quad");
    static constexpr int RB = PPT >= 8 ? 1 : 8 / PPT, NREG = RB * PPT;
};
template <typename T, int NT, int K>
__device__ __forceinline__ void la_rows_load(const T* W, int nrow, int r0, int tid, u32x4 (&f)[LaRows<T, K>::NREG]) {
    using R = LaRows<T, K>;
    const int pd = tid >> 2, prt = tid & 3;
#pragma unroll
    for (int rb = 0; rb < R::RB; ++rb) {
        const T* wrow = W + (size_t)min(r0 + rb * (NT / 4) + pd, nrow - 1) * K;   // clamped: no branch around a load
#pragma unroll
        for (int i = 0; i < R::PPT; ++i) f[rb * R::PPT + i] = ld16(wrow + (prt + 4 * i) * R::PER16);
    }
}
template <typename T, int NT, int K, class Vec, class Emit>
__device__ __forceinline__ void la_rows_dot(const T* W, int nrow, Vec&& vec, Emit&& emit, int tid, u32x4 (&cur)[LaRows<T, K>::NREG]) {
    using R = LaRows<T, K>;
    constexpr int PER16 = R::PER16, STEP = R::RB * (NT / 4);
    const int pd = tid >> 2, prt = tid & 3;
    auto consume = [&](int r0) {
#pragma unroll
        for (int rb = 0; rb < R::RB; ++rb) {
            const int row = r0 + rb * (NT / 4) + pd;
            const float* v = vec(min(row, nrow - 1));
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < R::PPT; ++i) {
                float wf[PER16];
                unpack16<T, PER16>(cur[rb * R::PPT + i], wf);
                const float* vp = v + (prt + 4 * i) * PER16;
#pragma unroll
                for (int e = 0; e < PER16; ++e) acc = fmaf(wf[e], vp[e], acc);
            }
            acc = quad_sum(acc);
            if (prt == 0 && row < nrow) emit(row, acc);
        }
    };
    int r0 = 0;
    for (; r0 + STEP < nrow; r0 += STEP) {                    // every step but the last requests its successor first
        u32x4 nxt[R::NREG];
        la_rows_load<T, NT, K>(W, nrow, r0 + STEP, tid, nxt);
        consume(r0);
#pragma unroll
        for (int i = 0; i < R::NREG; ++i) cur[i] = nxt[i];
    }
    consume(r0);
}

// One (row, head group) tile by NW * 64 threads (tid 0..NT-1; in the persistent kernel the whole 512-thread workgroup).
// COH / wait_prev: see dec_gemm.h.  poll_wave: wave 0 polls the team's flag line with VECTOR loads (its wait would also wait
// for anything it requested earlier), so it requests its first tiles behind the wait.
// NFQ: 16-key tiles a wave keeps in flight (2 or 4; 1 where a tile's registers do not allow more).
template <typename T, int D_, bool COH, int NFQ, class Wait>
__device__ __forceinline__ void lat_attn_tile(const LatAttnArgs<T>& a, int tile, int tid_in, unsigned char* lds, bool poll_wave,
                                              Wait&& wait_prev) {
    constexpr int NW = la_waves<D_>(), NT = NW * 64;
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
    const int inner = a.heads * DH;
    int tid = tid_in, lane = tid & 63, lc = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the thread index is made opaque again at every phase boundary: hipcc otherwise computes the per-thread addresses of ALL
    // phases at tile entry and carries them (64-bit row pointers, 40 LDS offsets) through the phases that need the registers
    auto phase = [&]() { asm volatile("" : "+v"(tid)); lane = tid & 63; lc = lane & 15; lg = lane >> 4; };
    unsigned long long ts0 = 0, ts1 = 0;
    if (a.stamps) ts0 = __builtin_amdgcn_s_memrealtime();

    float* zs = reinterpret_cast<float*>(lds);
    float* qs = zs + D_;
    float* st_m = qs + LA_GMAX * 64;                          // [NW][16]
    float* st_l = st_m + NW * 16;                             // [NW][16]
    unsigned char* qc = reinterpret_cast<unsigned char*>(st_l + NW * 16);     // q' image [16][D_] of T, later merged c [nh][D_] f32
    unsigned char* scr_all = qc + (size_t)16 * D_ * 4;
    unsigned char* scr = scr_all + (size_t)wave * 16 * ROWB;
    float* part = reinterpret_cast<float*>(scr_all);          // [NW][nh][D_] after the key loop

    // ---- 0. the row itself (launch path: nothing to wait for, so it is requested FIRST -- a wave's loads return in order and
    //         the LayerNorm below is the head of the tile's dependency chain), then the weights of the q and q' projections ----
    constexpr bool WIDE = D_ % 256 == 0;
    constexpr int NV = WIDE ? D_ / 256 : 1, NE = WIDE ? 1 : D_ / 64;
    float4 yv[NV]; float ye[NE];
    auto load_y = [&]() {
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) yv[i] = ldc_f4_at<COH>(a.y, (size_t)img * D_ + i * 256 + lane * 4);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) ye[i] = ldc_f32<COH>(a.y + (size_t)img * D_ + i * 64 + lane);
        }
    };
    if constexpr (!COH) { if (wave == 0) load_y(); }
    const T* wq_t = a.Wq + (size_t)h0 * DH * D_;
    const T* wk_t = a.WkT + (size_t)h0 * D_ * DH;
    const T* wv_t = a.Wv + (size_t)h0 * DH * D_;
    u32x4 pre_q[LaRows<T, D_>::NREG], pre_k[LaRows<T, DH>::NREG];
    la_rows_load<T, NT, D_>(wq_t, nh * DH, 0, tid, pre_q);
    la_rows_load<T, NT, DH>(wk_t, nh * D_, 0, tid, pre_k);

    // ---- encoder rows: two 16-key tiles per wave in flight; every wave runs the same number of tiles (masked past the end) ----
    const int kvimg = img / a.kv_div;
    const T* eb = a.enc + (size_t)kvimg * a.len * D_;
    const int ntiles = (a.len + 15) >> 4, nt_w = (ntiles + NW - 1) / NW;
    const int last_key = a.len - 1;
    // NF: tiles requested AHEAD of the one being consumed (registers: KC x 4 per tile), NF + 1 statically named register
    // sets used round-robin.  Every request is unconditional (a branch around the refill made hipcc wait vmcnt(0) before every
    // tile, so each tile paid the full latency of the refill issued just before it; rotating the sets through copies forces
    // the same wait at the copy): tiles past the end clamp to the last key's row -- one cache line per instruction.
    constexpr int NF = KC <= 8 ? NFQ : 0;                     // (wide rows / f32: a tile is 64-96 registers -- one set, no request ahead)
    u32x4 cur[KC], a1[NF >= 1 ? KC : 1], a2[NF == 2 ? KC : 1];
    auto issue = [&](int it, u32x4 (&dst)[KC]) {              // `it`-th tile of this wave: keys 16 * (wave + it * NW) ..
        const int key = min((wave + it * NW) * 16 + lc, last_key);
        const T* p = eb + (size_t)key * D_ + lg * PER16;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) dst[kc] = ld16(p + kc * KCH);
    };
    const bool late = poll_wave && wave == 0;
    auto issue_first = [&]() {
        issue(0, cur);
        if constexpr (NF == 2) issue(1, a1);
    };
    (void)a1; (void)a2;
    if (!late) issue_first();
    wait_prev();
    if (late) issue_first();

    // ---- 1. row prologue (wave 0): x = LN(y), z = LN(x) -> LDS ----
    if (wave == 0) {
        const float inv_d = 1.0f / D_;
        if constexpr (COH) load_y();
        if constexpr (WIDE) {
            float4 (&v)[NV] = yv; float4 g[NV], b[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                g[i] = *reinterpret_cast<const float4*>(a.gamma + c);
                b[i] = *reinterpret_cast<const float4*>(a.beta + c);
            }
            ln64<NV, FAST>(v, NV, g, b, inv_d);
            if (hg == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(a.x_out + (size_t)img * D_ + i * 256 + lane * 4) = v[i];
            }
            ln64<NV, FAST>(v, NV, g, b, inv_d);
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(&zs[i * 256 + lane * 4]) = v[i];
        } else {
            float (&vals)[NE] = ye;
            auto ln_narrow = [&]() {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < NE; ++i) s += vals[i];
                const float mean = wave_sum(s) * inv_d;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < NE; ++i) { const float dl = vals[i] - mean; q += dl * dl; }
                const float rstd = rsqrt_sel<FAST>(wave_sum(q) * inv_d + LN_EPS);
#pragma unroll
                for (int i = 0; i < NE; ++i) { const int c = i * 64 + lane; vals[i] = (vals[i] - mean) * rstd * a.gamma[c] + a.beta[c]; }
            };
            ln_narrow();
            if (hg == 0) {
#pragma unroll
                for (int i = 0; i < NE; ++i) a.x_out[(size_t)img * D_ + i * 64 + lane] = vals[i];
            }
            ln_narrow();
#pragma unroll
            for (int i = 0; i < NE; ++i) zs[i * 64 + lane] = vals[i];
        }
    }
    // the q' image's unused head rows must be finite: zero the whole image (the projection overwrites its rows behind the barrier)
    for (int i = tid; i < 16 * D_ * (int)sizeof(T) / 16; i += NT) st16(qc + (size_t)i * 16, u32x4{0u, 0u, 0u, 0u});
    __syncthreads();

    // ---- 2. q_h = 0.125 Wq_h z (0.125 is exact), q'_h = Wk_h^T q_h -> the B operand image ----
    la_rows_dot<T, NT, D_>(wq_t, nh * DH, [&](int) { return zs; }, [&](int row, float v) { qs[row] = v * ATTN_SCALE; }, tid, pre_q);
    __syncthreads();
    phase();
    la_rows_dot<T, NT, DH>(wk_t, nh * D_, [&](int row) { return qs + (row / D_) * DH; },
                                  [&](int row, float v) {      // 16-byte chunks of a head's row XOR-swizzled with the head: conflict-free fragment reads
                                      const int h = row / D_, d = row - h * D_, ch = (d / PER16) ^ (h & QSM);
                                      reinterpret_cast<T*>(qc)[(size_t)h * D_ + ch * PER16 + (d % PER16)] = Elem<T>::from_f32(v);
                                  }, tid, pre_k);
    __syncthreads();
    phase();
    // B fragments of S^T = enc q'^T: lane (head = lc, group lg) holds q'[head][kc*KCH + lg*PER16 ..]; kept in registers where
    // they fit beside two tiles in flight, else re-read from LDS for every tile
    constexpr bool QF_REGS = NF == 2 && !COH;               // (the persistent kernel has no registers to spare: LDS reads there)
    auto qfrag = [&](int kc) { return ld16(qc + ((size_t)lc * D_ + (((kc * 4 + lg) ^ (lc & QSM)) * PER16)) * sizeof(T)); };
    u32x4 qf[QF_REGS ? KC : 1];
    if constexpr (QF_REGS) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) qf[kc] = qfrag(kc);
    }

    if (a.stamps && a.stamp_mode == 1) ts1 = __builtin_amdgcn_s_memrealtime();
    // ---- 3. key tiles ----
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
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { if constexpr (QF_REGS) mma16<T>(s, e[kc], qf[kc]); else mma16<T>(s, e[kc], qfrag(kc)); }
        // the same registers -> the wave's LDS image (row = key lc, chunks 4 kc + lg)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) st16(scr + la_off<ROWB, XM>(lc, 4 * kc + lg), e[kc]);
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
    } else {
        for (int it = 0; it < nt_w; it += 6) {
            issue(it + 1, a1); process(it, cur);
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
        }
    }
    if (a.stamps && a.stamp_mode != 1) { asm volatile("" :: "v"(acc[0][0])); ts1 = __builtin_amdgcn_s_memrealtime(); }

    // ---- 4. merge the waves' (m, l, c^T); the value projection's first weights fly under it ----
    phase();
    u32x4 pre_v[LaRows<T, D_>::NREG];
    la_rows_load<T, NT, D_>(wv_t, nh * DH, 0, tid, pre_v);
    l_run = grp4_sum(l_run);
    if (lg == 0) { st_m[wave * 16 + lc] = m_run; st_l[wave * 16 + lc] = l_run; }
    __syncthreads();                                          // every wave is done with its transposition image
    if (lc < nh) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            *reinterpret_cast<f32x4*>(&part[((size_t)wave * nh + lc) * D_ + dt * 16 + lg * 4]) = acc[dt];
    }
    __syncthreads();
    float* cs = reinterpret_cast<float*>(qc);                 // merged, normalised c [nh][D_] (the q' image is dead)
    for (int idx = tid; idx < nh * D_; idx += NT) {
        const int h = idx / D_, d = idx - h * D_;
        float m = st_m[h];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, st_m[w * 16 + h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = exp_sel<FAST>(st_m[w * 16 + h] - m);
            num = fmaf(e, part[((size_t)w * nh + h) * D_ + d], num);
            den = fmaf(e, st_l[w * 16 + h], den);
        }
        cs[idx] = num / den;
    }
    __syncthreads();

    // ---- 5. o_h = Wv_h c_h ; 'b h n d -> b n (h d)' ----
    la_rows_dot<T, NT, D_>(wv_t, nh * DH, [&](int row) { return cs + (row / DH) * D_; },
                                  [&](int row, float v) { a.out[(size_t)img * inner + h0 * DH + row] = Elem<T>::from_f32(v); }, tid, pre_v);
    if (a.stamps && tid == 0) {
        unsigned long long* d = a.stamps + 3 * (size_t)tile;
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_amdgcn_s_memrealtime();
    }
}

template <typename T, int D_, int NFQ = 2>
__global__ __launch_bounds__(la_waves<D_>() * 64) void lat_attn_kernel(LatAttnArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char la_smem[];
    // workgroups b and b + 8 run on one XCD (round-robin dispatch): the head groups of ONE image go to ONE XCD, whose L2 then
    // fetches the image's encoder rows once for all of them (row = (slot / ngrp) * 8 + xcd, head group = slot % ngrp)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int row = (slot / a.ngrp) * 8 + xcd, hg = slot - (slot / a.ngrp) * a.ngrp;
    if (row >= a.rows) return;
    lat_attn_tile<T, D_, false, NFQ>(a, row * a.ngrp + hg, threadIdx.x, la_smem, false, NoWait{});
}

}  // namespace txo
