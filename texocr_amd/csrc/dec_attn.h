// Decode-step attention sub-layer front half: LayerNorm sandwich + q (cross) / q,k,v (self) projection +
// ONE query row per (image, head) against the cached K/V panel, in one launch.
//
// Reference (KV-cached form of AttentionLayers.forward + MultiHeadAttention.forward, model/attention.py):
//   x = LN(y) (next residual; or x = tok_emb + pos_emb for the first sub-layer, decoder.py:51-52)
//   z = LN(x)                                            (:243, same shared gamma/beta)
//   q = z Wq^T  [self: also k_t = z Wk^T, v_t = z Wv^T appended to the cache]   (:124-127, no bias)
//   cross : keys/values = per-layer projections of the raw encoder output, N rows, non-causal.  The reference
//           re-projects them from `enc` at every step (:124-126); here they are projected once per generate()
//           and re-read every step -> the HBM-bound kernel of the whole path.
//   self  : keys/values of positions 0..t (causal; with one query at the last position the triangular fill
//           of :158-163 masks nothing that is cached).
//   energy = q.k * 0.125 ; softmax (max-subtracted, fp32) ; out = sum_j p_j v_j ; 'b h n d -> b n (h d)'.
//
// Layout: K,V [images*heads][lmax][64] of T (f32: 256-byte rows, bf16: 128-byte rows).
//
// gfx950 mapping: one 256-thread block per (image, head).  The whole K and V panel of the block (up to
// NL wave-instructions of 1 KiB per wave each) is requested AT KERNEL ENTRY, 16 B per lane, straight into
// VGPRs: one HBM round trip for the launch instead of a load/compute ping-pong, ~300 KB in flight per CU.
// While it flies, wave 0 normalises the image's row and all threads project the head's 64 (or 3x64)
// outputs from L2-resident weights.  Scores stay in registers (each lane owns the keys it loaded); one
// block-wide max and one sum/accumulator reduction go through LDS.  Longer panels (N > 32*NL keys in bf16,
// 16*NL in f32) are processed in several passes with an online-softmax rescale.
// Algorithmic bytes per launch = images*heads*2*len*64*sizeof(T).   Bound: HBM (8 TB/s peak).
#pragma once
#include "common.h"
#include "dec_gemm.h"   // ldc_* coherent loads, NoWait

namespace txo {

enum { ATT_CROSS = 0, ATT_SELF = 1 };
// APRO_NONE: no prologue / projection in this launch -- q comes from a.qin and (self) row t is already in the cache
enum { APRO_EMBED = 0, APRO_LN2 = 1, APRO_NONE = 2 };

template <typename T> struct DecAttnArgs {
    // row prologue
    const float* y;                 // [images][D] (APRO_LN2)
    const int64_t* tok; const float* tok_emb; const float* pos_emb;   // APRO_EMBED
    float* x_out;                   // [images][D] residual (written by the head-0 block)
    const float* gamma; const float* beta;
    int D;
    const T* W;                     // cross: Wq [inner][D]; self: Wqkv [3*inner][D]
    const float* qin;               // APRO_NONE: [images][heads*64] fp32 query
    // attention
    T* K; T* V;                     // [images*heads][lmax][64]; self mode appends row t
    T* out;                         // [images][heads*64]
    int heads, lmax;
    int len;                        // cross: number of keys
    const int* t_ptr;               // decode position (device)
    int t_host;                     // >= 0: the position passed by the host (eager launches); -1: read *t_ptr (graph replay)
    // beam search: rows are (image, beam) slots.  Cross K/V are shared by the kv_div beams of an image; a beam's
    // self-attention history is scattered over slots: position p of row r lives in slot path[r*path_stride + p].
    int kv_div;                     // cross: K/V image = row / kv_div (1 without beams)
    const short* path; int path_stride;   // self (APRO_NONE): null without beams
    unsigned long long* stamps;     // diagnostic (TXO_STAMPS): per block {entry, K panel consumed, exit}
    // padding mask over the decoded positions (reference attention.py:130-155: energy filled with -FLT_MAX where the key is masked):
    // kmask[row * kmask_stride + position] == 0 -> that position is never attended by a later query (KMASK instantiations only)
    const unsigned char* kmask; int kmask_stride;
};

template <typename T, int PER16>
__device__ inline void unpack16(const u32x4& raw, float (&f)[PER16]) {
    if constexpr (PER16 == 4) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
        f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
    } else {   // bf16 -> f32 is a 16-bit shift
        f[0] = __uint_as_float(raw.x << 16); f[1] = __uint_as_float(raw.x & 0xffff0000u);
        f[2] = __uint_as_float(raw.y << 16); f[3] = __uint_as_float(raw.y & 0xffff0000u);
        f[4] = __uint_as_float(raw.z << 16); f[5] = __uint_as_float(raw.z & 0xffff0000u);
        f[6] = __uint_as_float(raw.w << 16); f[7] = __uint_as_float(raw.w & 0xffff0000u);
    }
}

template <int NVMAX, bool FAST>
__device__ inline void ln64(float4 (&v)[NVMAX], int nv, const float4 (&g)[NVMAX], const float4 (&b)[NVMAX], float inv_d) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrt_sel<FAST>(wave_sum(q) * inv_d + LN_EPS);
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) if (i < nv) {
        v[i].x = v[i].x * rstd * g[i].x + b[i].x; v[i].y = v[i].y * rstd * g[i].y + b[i].y;
        v[i].z = v[i].z * rstd * g[i].z + b[i].z; v[i].w = v[i].w * rstd * g[i].w + b[i].w;
    }
}

// NL: wave-instructions (1 KiB each) of K (then of V) a wave keeps in registers per pass.
// WB: projections whose weight rows are requested together at kernel entry (all of them when registers allow).
// NARROW: embed_dim is not a multiple of 256 (test-size models): scalar row prologue.
//
// The hot loops are written WITHOUT data-dependent branches (clamped unconditional loads, masked scores of
// -3e38 whose exp is exactly 0, selects): a guarded load becomes its own basic block and hipcc's waitcnt
// insertion then falls back to vmcnt(0) per iteration, which serialises the whole K/V stream.
// cross attention: wave-instructions of K (then V) per pass; the launch path and the persistent kernel must agree (same passes ->
// same bits).  10 = 320 keys (bf16) per pass, two passes at 589 keys.  One pass of 20 kept the whole panel in flight but held 160
// VGPRs of it: the persistent kernel (256-VGPR cap, both attention variants inlined) spilled around it and a CU ran two blocks of
// the launch kernel at most.  Measured (same box, ms per generate): batch 64 persistent 36.4 -> 33.5, batch 256 launches 101.4 -> 98.8;
// 7 and 5 are within noise of 10.
#ifndef TXO_NL_CROSS
#define TXO_NL_CROSS 10
#endif
constexpr int DA_NL_CROSS = TXO_NL_CROSS;

// LDS of one 256-thread group
template <bool BEAM> struct DecAttnLds {
    __attribute__((aligned(16))) float zs[768];   // normalised row
    __attribute__((aligned(16))) float qkv[3][DH];
    float part[4][DH];
    float stat[12];
    short pth[BEAM ? 1024 : 8];                   // beam: slot of every history position of this row
};

// One (image, head) pair `bh` by 256 threads (tid 0..255).  COH / wait_prev: see dec_gemm.h (persistent kernel); KV_EARLY:
// the K panel does not depend on the previous stage (cross attention), so every wave except the one that polls the team
// counter (tid < 64 of the workgroup's first group: a poll's wait would also wait for that wave's own panel) requests it
// BEFORE the wait.  valid = false: same barriers, clamped addresses, no stores.
// pf(): called once this tile's own loads have been issued (persistent kernel: requests the next stage's weights there).
template <typename T, int MODE, int APRO, int NL, int WB, bool NARROW, bool BEAM, bool COH, class Wait, class Pf = NoPf, bool KMASK = false>
__device__ __forceinline__ void dec_attn_tile(const DecAttnArgs<T>& a, int bh, int tid, DecAttnLds<BEAM>& L_, bool valid,
                                              bool poll_wave, Wait&& wait_prev, Pf&& pf = NoPf{}) {
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int LPR = DH / PER16;          // lanes per 64-element row: 16 (f32) / 8 (bf16)
    constexpr int KPI = 64 / LPR;            // keys per wave-instruction: 4 / 8
    constexpr int KPB = KPI * 4;             // keys per block-instruction (four waves)
    constexpr bool FUSED = APRO != APRO_NONE;
    constexpr int NP = MODE == ATT_SELF ? 3 : 1;
    constexpr int NVMAX = 3;                 // row prologue: D <= 768 (one wave, float4 per lane per 256)
    constexpr int WMAX = 8;                  // weight pieces per thread per projection per group (bf16: one group at D = 256)
    const int img = bh / a.heads, head = bh - img * a.heads;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane % LPR, kq = lane / LPR;
    const int D = a.D, inner = a.heads * DH;
    unsigned long long ts0 = 0, ts1 = 0;
    if (a.stamps) ts0 = __builtin_amdgcn_s_memrealtime();
    const int kvimg = MODE == ATT_CROSS ? img / a.kv_div : img;
    T* Kb = a.K + ((size_t)kvimg * a.heads + head) * a.lmax * DH;
    T* Vb = a.V + ((size_t)kvimg * a.heads + head) * a.lmax * DH;

    float (&zs)[768] = L_.zs; float (&qkv)[3][DH] = L_.qkv; float (&part)[4][DH] = L_.part; float (&stat)[12] = L_.stat;
    short* pth = L_.pth;

    // ---- 0. projection weights: thread (d = tid>>2, prt = tid&3) owns 16-byte pieces prt, prt+4, ... of row d.
    //         Requested first: they depend on nothing and are needed first. ----
    const int pd = tid >> 2, prt = tid & 3;
    const int pieces = (D * (int)sizeof(T)) >> 4;
    const int ngrp = (pieces + 4 * WMAX - 1) / (4 * WMAX);   // weight groups per row (1 at D = 256)
    u32x4 wreg[WB][WMAX];
    auto issue_w = [&](int p0, int g) {
#pragma unroll
        for (int p = 0; p < WB; ++p) {
            const int pp = min(p0 + p, NP - 1);
            const T* wrow = a.W + ((size_t)pp * inner + head * DH + pd) * D;
#pragma unroll
            for (int i = 0; i < WMAX; ++i) wreg[p][i] = ld16(wrow + min(prt + 4 * (g * WMAX + i), pieces - 1) * PER16);
        }
    };
    if constexpr (FUSED) issue_w(0, 0);

    // ---- 1. request the K panel of the first pass ----
    constexpr bool KV_EARLY = COH && MODE == ATT_CROSS && !BEAM;   // persistent kernel: the cross panel is older than the launch
    // persistent kernel, plain self attention: rows 0..t-1 of the cache are older than this position's hand-offs, so the
    // history is requested BEFORE the wait; only row t (appended by the stage before this one) is read behind it and takes
    // the place of its key by selects (same values in the same operations -> same bits).  A row is never requested before
    // its own position's hand-off (an early copy in this CU's L1 would outlive the append), hence nothing is early at t = 0.
    constexpr bool HIST_EARLY = COH && MODE == ATT_SELF && !FUSED && !BEAM;
    if constexpr (!KV_EARLY && !HIST_EARLY) wait_prev();
    int t = 0;
    if constexpr (MODE == ATT_SELF || APRO == APRO_EMBED) t = (COH || a.t_host >= 0) ? a.t_host : *a.t_ptr;
    // cached keys: fused self handles the new key t apart; plain self finds it in the cache already
    const int L = MODE == ATT_CROSS ? a.len : (FUSED ? t : t + 1);
    // valid = false (persistent kernel: a group without a tile keeps the workgroup's barriers): every panel address clamps
    // to row 0, one cache line per wave instead of the whole panel
    const int Lm1 = valid ? max(L - 1, 0) : 0;
    const int key0 = wave * KPI + kq;
    // byte-row address of key `key` in the K (or V) panel: own slot, or (beam) the slot recorded for that position
    const size_t slot_stride = (size_t)a.heads * a.lmax * DH;
    auto row_off = [&](int key) -> long long {
        if constexpr (BEAM) return (long long)((int)pth[key] - img) * (long long)slot_stride + (long long)key * DH;   // relative to the own slot
        else return (long long)key * DH;
    };
    if constexpr (BEAM) {
        for (int p = tid; p < L; p += 256) pth[p] = p == L - 1 ? (short)img : a.path[(size_t)img * a.path_stride + p];
        __syncthreads();
    }
    u32x4 rk[NL], rv[NL];
    // small panels (self attention) request V together with K; long ones (cross) request each V row as its K
    // registers are consumed, which halves the live panel (K and V of NL = 20 would not fit 256 VGPRs)
    constexpr bool V_EARLY = MODE == ATT_SELF;
    // cache policy: the cross panels (309 MB per step at batch 64) are streamed non-temporally so that they do not push
    // the decoder weights out of L2; the self-attention history (<= 67 MB over all layers) is re-read every step and
    // is read with the default policy (measured: self-attention launch 6.2 -> 5.4 us)
    // the self-attention cache is read with plain loads also in the persistent kernel: a row is final once its position's
    // hand-off has passed, and no CU reads it earlier (clamps below), so no L1 can hold an older copy of its lines
    auto ld_kv = [](const T* p) -> u32x4 { if constexpr (MODE == ATT_SELF && !(COH && TXO_PS_W_NT)) return ld16(p); else return ld16_stream(p); };
    int clamp_row = Lm1;
    auto issue_k = [&](int base) {
#pragma unroll
        for (int u = 0; u < NL; ++u) rk[u] = ld_kv(Kb + row_off(min(base + u * KPB + key0, clamp_row)) + sub * PER16);
        if constexpr (V_EARLY) {
#pragma unroll
            for (int u = 0; u < NL; ++u) rv[u] = ld_kv(Vb + row_off(min(base + u * KPB + key0, clamp_row)) + sub * PER16);
        }
    };
    [[maybe_unused]] u32x4 kt = {0u, 0u, 0u, 0u}, vt = {0u, 0u, 0u, 0u};
    if constexpr (KV_EARLY) {
        if (!poll_wave) issue_k(0);
        wait_prev();
        if (poll_wave) issue_k(0);
    } else if constexpr (HIST_EARLY) {
        const bool early = t > 0 && !poll_wave;                // the polling wave's wait would wait for its own panel too
        clamp_row = valid ? max(t - 1, 0) : 0;
        if (early) issue_k(0);
        wait_prev();
        if (!early) {
            if (t > 0) issue_k(0);
            else {
#pragma unroll
                for (int u = 0; u < NL; ++u) { rk[u] = kt; rv[u] = kt; }   // t = 0: no history; zeros (finite) under masked keys
            }
        }
        kt = ld16(Kb + (size_t)t * DH + sub * PER16);
        vt = ld16(Vb + (size_t)t * DH + sub * PER16);
        clamp_row = Lm1;                                       // later passes of a long history run behind the wait anyway
    } else {
        issue_k(0);
    }

    // ---- 2. row prologue (wave 0): x = LN(y) | emb ; z = LN(x) -> LDS ----
    if (FUSED && wave == 0) {
        const float inv_d = 1.0f / D;
        if constexpr (!NARROW) {
            const int nv = D >> 8;
            float4 v[NVMAX], g[NVMAX], b[NVMAX];
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv) {
                const int c = i * 256 + lane * 4;
                g[i] = *reinterpret_cast<const float4*>(a.gamma + c);
                b[i] = *reinterpret_cast<const float4*>(a.beta + c);
                if constexpr (APRO == APRO_EMBED) {
                    const float4 p = *reinterpret_cast<const float4*>(a.tok_emb + (size_t)ldc_i64<COH>(a.tok + img) * D + c);
                    const float4 q = *reinterpret_cast<const float4*>(a.pos_emb + (size_t)t * D + c);
                    v[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
                } else {
                    v[i] = ldc_f4_at<COH>(a.y, (size_t)img * D + c);
                }
            }
            if constexpr (APRO == APRO_LN2) ln64<NVMAX, sizeof(T) == 2>(v, nv, g, b, inv_d);
            if (head == 0 && valid) {
#pragma unroll
                for (int i = 0; i < NVMAX; ++i) if (i < nv)
                    *reinterpret_cast<float4*>(a.x_out + (size_t)img * D + i * 256 + lane * 4) = v[i];
            }
            ln64<NVMAX, sizeof(T) == 2>(v, nv, g, b, inv_d);
#pragma unroll
            for (int i = 0; i < NVMAX; ++i) if (i < nv) *reinterpret_cast<float4*>(&zs[i * 256 + lane * 4]) = v[i];
        } else {
            const int ne = (D + 63) >> 6;
            float vals[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) if (i < ne) {
                const int c = i * 64 + lane;
                vals[i] = 0.f;
                if (c < D) {
                    if constexpr (APRO == APRO_EMBED) vals[i] = a.tok_emb[(size_t)ldc_i64<COH>(a.tok + img) * D + c] + a.pos_emb[(size_t)t * D + c];
                    else vals[i] = ldc_f32<COH>(a.y + (size_t)img * D + c);
                }
            }
            auto ln_narrow = [&]() {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 12; ++i) if (i < ne) s += vals[i];
                const float mean = wave_sum(s) * inv_d;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < 12; ++i) if (i < ne) { const float dl = (i * 64 + lane < D) ? vals[i] - mean : 0.f; q += dl * dl; }
                const float rstd = rsqrt_sel<sizeof(T) == 2>(wave_sum(q) * inv_d + LN_EPS);
#pragma unroll
                for (int i = 0; i < 12; ++i) if (i < ne) {
                    const int c = i * 64 + lane;
                    vals[i] = c < D ? (vals[i] - mean) * rstd * a.gamma[c] + a.beta[c] : 0.f;
                }
            };
            if constexpr (APRO == APRO_LN2) ln_narrow();
            if (head == 0 && valid) {
#pragma unroll
                for (int i = 0; i < 12; ++i) if (i < ne) { const int c = i * 64 + lane; if (c < D) a.x_out[(size_t)img * D + c] = vals[i]; }
            }
            ln_narrow();
#pragma unroll
            for (int i = 0; i < 12; ++i) if (i < ne) { const int c = i * 64 + lane; if (c < D) zs[c] = vals[i]; }
        }
    }
    if constexpr (FUSED) __syncthreads();

    // ---- 3. projection from the weight pieces already in flight ----
#pragma unroll
    for (int p0 = 0; p0 < (FUSED ? NP : 0); p0 += WB) {
        float accp[WB];
#pragma unroll
        for (int p = 0; p < WB; ++p) accp[p] = 0.f;
        for (int g = 0; g < ngrp; ++g) {
            if (p0 > 0 || g > 0) issue_w(p0, g);
#pragma unroll
            for (int i = 0; i < WMAX; ++i) {
                const int piece = prt + 4 * (g * WMAX + i);
                const float keep = piece < pieces ? 1.f : 0.f;
                const float* zp = &zs[min(piece, pieces - 1) * PER16];
                float zf[PER16];
#pragma unroll
                for (int e = 0; e < PER16; ++e) zf[e] = zp[e] * keep;
#pragma unroll
                for (int p = 0; p < WB; ++p) {
                    float wf[PER16];
                    unpack16<T, PER16>(wreg[p][i], wf);
#pragma unroll
                    for (int e = 0; e < PER16; ++e) accp[p] = fmaf(wf[e], zf[e], accp[p]);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < WB; ++p) if (p0 + p < NP) {
            const float v = quad_sum(accp[p]);
            if (prt == 0) qkv[p0 + p][pd] = v;
        }
    }
    if constexpr (FUSED) __syncthreads();
    float s_new = 0.f;                                        // self: score of the new key t
    if constexpr (MODE == ATT_SELF && FUSED) {
        // append k_t, v_t (rounded to the cache type, exactly what later steps will read back)
        if (tid < 2 * DH && valid) {
            const int which = tid >> 6, dd = tid & 63;
            const T val = Elem<T>::from_f32(qkv[1 + which][dd]);
            (which ? Vb : Kb)[(size_t)t * DH + dd] = val;
            qkv[1 + which][dd] = Elem<T>::to_f32(val);
        }
        __syncthreads();
        s_new = wave_sum(qkv[0][lane] * ATTN_SCALE * qkv[1][lane]);
    }

    // this lane's 16-byte piece of the query, pre-scaled (0.125 is exact)
    float qv[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e)
        qv[e] = (FUSED ? qkv[0][sub * PER16 + e] : ldc_f32<COH>(a.qin + (size_t)img * inner + head * DH + sub * PER16 + e)) * ATTN_SCALE;

    pf();

    // ---- 4. passes over the panel (one pass when len <= NL*KPB) ----
    float m_run = (MODE == ATT_SELF && FUSED) ? s_new : -3.0e38f, l_run = 0.f;
    float acc[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) acc[e] = 0.f;
    const float count_me = sub == 0 ? 1.f : 0.f;              // each key's probability is summed once
    auto do_pass = [&](int base, int slot) {
        float sc[NL];
        float mx = -3.0e38f;
        // self attention: slots (KPB keys each) wholly past the history are skipped by a wave-uniform branch -- their scores
        // would be masked to -3e38 and their probabilities exactly 0, so nothing changes but the time (half the slots on average).
        // Only where K and V are both in registers already (V_EARLY): a branch around a load costs hipcc's counted waits.
        const int nslot = V_EARLY ? (L - base + KPB - 1) / KPB : NL;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if (V_EARLY && u >= nslot) { sc[u] = -3.0e38f; continue; }
            const int key = base + u * KPB + key0;
            float kf[PER16];
            if constexpr (HIST_EARLY) {
                const bool is_t = key == t;
                u32x4 kk = rk[u];
                kk.x = is_t ? kt.x : kk.x; kk.y = is_t ? kt.y : kk.y; kk.z = is_t ? kt.z : kk.z; kk.w = is_t ? kt.w : kk.w;
                unpack16<T, PER16>(kk, kf);
            } else {
                unpack16<T, PER16>(rk[u], kf);
            }
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < PER16; ++e) d = fmaf(qv[e], kf[e], d);
            // the K registers of this slot are dead now: request the matching V rows into their place
            if constexpr (!V_EARLY) rv[u] = ld_kv(Vb + row_off(min(key, Lm1)) + sub * PER16);
            d = LPR == 8 ? row8_sum(d) : row16_sum(d);                         // all LPR lanes of the key get the dot
            bool attend = key < L;
            if constexpr (KMASK) attend = attend && a.kmask[(size_t)img * a.kmask_stride + min(key, a.lmax - 1)] != 0;
            d = attend ? d : -3.0e38f;
            sc[u] = d;
            mx = fmaxf(mx, d);
            // keep the V requests in program order behind the K consumption (else both panels are live at once)
            if constexpr (!V_EARLY) __builtin_amdgcn_sched_barrier(0);
        }
        mx = wave_max(mx);
        if (lane == 0) stat[slot * 4 + wave] = mx;
        __syncthreads();
        const float pm = fmaxf(fmaxf(stat[slot * 4 + 0], stat[slot * 4 + 1]), fmaxf(stat[slot * 4 + 2], stat[slot * 4 + 3]));
        const float m_new = fmaxf(m_run, pm);
        const float alpha = exp_sel<sizeof(T) == 2>(m_run - m_new);   // first pass: exp(-huge) = 0 with acc = l = 0
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int e = 0; e < PER16; ++e) acc[e] *= alpha;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if (V_EARLY && u >= nslot) continue;
            const float p = exp_sel<sizeof(T) == 2>(sc[u] - m_new);   // masked keys: exp(-3e38 - m) == 0 exactly
            l_run = fmaf(p, count_me, l_run);
            float vf[PER16];
            if constexpr (HIST_EARLY) {
                const bool is_t = base + u * KPB + key0 == t;
                u32x4 vv = rv[u];
                vv.x = is_t ? vt.x : vv.x; vv.y = is_t ? vt.y : vv.y; vv.z = is_t ? vt.z : vv.z; vv.w = is_t ? vt.w : vv.w;
                unpack16<T, PER16>(vv, vf);
            } else {
                unpack16<T, PER16>(rv[u], vf);
            }
#pragma unroll
            for (int e = 0; e < PER16; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
        }
    };
    do_pass(0, 0);
    // (requesting the next pass's K panel as soon as this pass's K registers are consumed -- under its softmax and PV -- measured
    // slower: 34.2 vs 33.5 ms per generate at batch 64)
    for (int base = NL * KPB, slot = 1; base < L; base += NL * KPB, slot ^= 1) {   // panels longer than one pass
        issue_k(base);
        do_pass(base, slot);
    }

    if (a.stamps) { asm volatile("" :: "v"(acc[0])); ts1 = __builtin_amdgcn_s_memrealtime(); }
    // ---- 5. reduce over key groups (shuffles), waves (LDS), normalise, merge heads ----
#pragma unroll
    for (int e = 0; e < PER16; ++e) {
        if constexpr (LPR == 8) acc[e] = xor8_sum(acc[e]);                      // the other key of the row (lane ^ 8)
        acc[e] = grp4_sum(acc[e]);
    }
    l_run = wave_sum(l_run);
    if (kq == 0) {
#pragma unroll
        for (int e = 0; e < PER16; ++e) part[wave][sub * PER16 + e] = acc[e];
    }
    if (lane == 0) stat[8 + wave] = l_run;
    __syncthreads();
    if (tid < DH && valid) {
        float o = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        float l = (stat[8] + stat[9]) + (stat[10] + stat[11]);
        if constexpr (MODE == ATT_SELF && FUSED) {
            const float p_new = exp_sel<sizeof(T) == 2>(s_new - m_run);
            o = fmaf(p_new, qkv[2][tid], o);
            l += p_new;
        }
        a.out[(size_t)img * inner + head * DH + tid] = Elem<T>::from_f32(o / l);
    }
    if (a.stamps && tid == 0) {
        unsigned long long* d = a.stamps + 3 * (size_t)bh;
        d[0] = ts0; d[1] = ts1; d[2] = __builtin_amdgcn_s_memrealtime();
    }
}

// A group WITHOUT a tile in a round of the persistent kernel: the workgroup barriers of dec_attn_tile (not BEAM) for a panel of
// L keys and nothing else (see dec_gemm_idle)
template <typename T, int MODE, int APRO, int NL, class Wait>
__device__ __forceinline__ void dec_attn_idle(int L, Wait&& wait_prev) {
    constexpr int KEYS_PER_PASS = NL * (64 / (DH / Elem<T>::PER16)) * 4;
    wait_prev();
    if constexpr (APRO != APRO_NONE) { __syncthreads(); __syncthreads(); }   // normalised row written; projection written
    if constexpr (MODE == ATT_SELF && APRO != APRO_NONE) __syncthreads();   // k_t, v_t appended
    __syncthreads();                                                        // first pass: wave maxima
    for (int base = KEYS_PER_PASS; base < L; base += KEYS_PER_PASS) __syncthreads();
    __syncthreads();                                                        // partial outputs
}

template <typename T, int MODE, int APRO, int NL, int WB, bool NARROW, bool BEAM = false, bool KMASK = false>
__global__ __launch_bounds__(256, 2) void dec_attn_kernel(DecAttnArgs<T> a) {
    __shared__ DecAttnLds<BEAM> lds;
    dec_attn_tile<T, MODE, APRO, NL, WB, NARROW, BEAM, false, NoWait, NoPf, KMASK>(a, blockIdx.x, threadIdx.x, lds, true, false, NoWait{});
}

}  // namespace txo
